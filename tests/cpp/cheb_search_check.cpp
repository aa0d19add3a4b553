// CPU harness over gemma_amd/csrc/lmm_search.hip.h -- the very code the per-SNP kernel runs for the table-driven
// lambda search (Brent + Newton over Chebyshev-in-log(lambda) series), compiled with g++.  tests/test_cheb_search.py
// prepares the series with numpy, this program runs polish_bracket / cheb_deriv on them, and the test compares with the
// oracle's lambda-hat and with derivatives computed from exact sums.
//
//   cheb_search_check in.bin out.bin
//   in : doubles  [C, reml, n, L]  then L cases of
//          [lam_lo, lam_hi, d_lo, d_hi, l_min, l_max, mid, inv_half, probe_lambda, qform,
//           s0x (C + 2: sum x^2, sum x u_a), s0f ((C + 1)(C + 2)/2: sum u_a u_b),
//           snp row ((C + 2) * CHEB_N), fix row (((C + 1)(C + 2)/2 + 3) * CHEB_N: pairs, g, log|H| (unused here), sum (1-H)^2)]
//   out: per case [status, l, probe_ok, probe_dev1, probe_dev2]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../gemma_amd/csrc/lmm_search.hip.h"

using namespace gemma_hip;

template <int C, bool REML>
static void run(const double *in, size_t L, double n, double *out) {
  const size_t npair = (size_t)(C + 1) * (C + 2) / 2;
  const size_t nsnp = (size_t)(C + 2) * CHEB_N, nfix = (npair + 3) * CHEB_N;
  const size_t head = 10 + (C + 2) + npair;
  const size_t stride = head + nsnp + nfix;
  for (size_t s = 0; s < L; ++s) {
    const double *c = in + s * stride;
    ChebEvaluator<C, REML> ev;
    ev.cs.snp = c + head;
    ev.cs.sstride = 1;
    ev.cs.fix = c + head + nsnp;
    ev.cs.qform = c[9] != 0.0;
    for (int a = 0; a < C + 2; ++a) ev.cs.s0x[a] = c[10 + a];
    ev.cs.s0f = c + 10 + (C + 2);
    ev.cs.xa0 = CHEB_N;
    ev.cs.mid = c[6];
    ev.cs.inv_half = c[7];
    ev.cs.n = n;
    double l = 0.0, l_temp = 0.0;
    const int st = polish_bracket(ev, c[0], c[1], c[2], c[3], c[4], c[5], l, l_temp);
    double d1 = 0.0, d2 = 0.0;
    const bool ok = ev.dev12(c[8], d1, d2);
    double *o = out + s * 5;
    o[0] = st; o[1] = l; o[2] = ok ? 1.0 : 0.0; o[3] = d1; o[4] = d2;
  }
}

int main(int argc, char **argv) {
  if (argc != 3) return 2;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 3;
  double hdr[4];
  if (fread(hdr, 8, 4, f) != 4) return 3;
  const int C = (int)hdr[0];
  const bool reml = hdr[1] != 0.0;
  const double n = hdr[2];
  const size_t L = (size_t)hdr[3];
  const size_t npair = (size_t)(C + 1) * (C + 2) / 2;
  const size_t stride = 10 + (C + 2) + npair + (size_t)(C + 2) * CHEB_N + (npair + 3) * CHEB_N;
  std::vector<double> in(L * stride), out(L * 5);
  if (fread(in.data(), 8, in.size(), f) != in.size()) return 3;
  fclose(f);
  switch (C * 2 + (reml ? 1 : 0)) {
  case 2: run<1, false>(in.data(), L, n, out.data()); break;
  case 3: run<1, true>(in.data(), L, n, out.data()); break;
  case 4: run<2, false>(in.data(), L, n, out.data()); break;
  case 5: run<2, true>(in.data(), L, n, out.data()); break;
  case 6: run<3, false>(in.data(), L, n, out.data()); break;
  case 7: run<3, true>(in.data(), L, n, out.data()); break;
  case 8: run<4, false>(in.data(), L, n, out.data()); break;
  case 9: run<4, true>(in.data(), L, n, out.data()); break;
  default: return 4;
  }
  f = fopen(argv[2], "wb");
  fwrite(out.data(), 8, out.size(), f);
  fclose(f);
  return 0;
}
