// TEST INFRASTRUCTURE: runs gemma_amd/csrc/mvlmm.hip.h with one CPU "lane" so that the per-SNP logic of the HIP kernel
// can be compared with oracle/mvlmm_oracle.c where no GPU exists.  Not part of the shipped library.
#include <cstddef>
#include <cstring>
#include <vector>
#define MV_HD inline
#define MV_OUTLINE
#include "../../gemma_amd/csrc/mvlmm.hip.h"
using namespace gemma_hip;

struct HostLanes {
  static constexpr int N = 1;
  static int lane() { return 0; }
  static double sum(double v) { return v; }
};

template <int D, int C> static void run(const MvArgs &g) {
  static double scratch[MvNrScratch<D, C>::DOUBLES];
  MvNr<D, C, HostLanes> nr(g, scratch);
  for (long s = 0; s < g.l; ++s) {
    nr.x = g.UtX + s * g.ld;
    mv_one_snp<D, C, HostLanes>(g, s, nr);
  }
}

// the run-time instance (DT = CT = 0): d, c from the arguments; gxe: two SNP rows
static void run_rt(const MvArgs &g) {
  std::vector<double> scratch((size_t)MvNrLayout(g.d, g.c).DOUBLES);
  MvRt rt;
  rt.d = g.d;
  rt.c = g.c;
  for (long s = 0; s < g.l; ++s) {
    if (g.UtX2) {
      mv_one_snp_gxe<HostLanes>(g, s, scratch.data());
    } else {
      MvNr<0, 0, HostLanes> nr(g, scratch.data(), rt);
      nr.x = g.UtX + s * g.ld;
      mv_one_snp<0, 0, HostLanes>(g, s, nr);
    }
  }
}

// fixed != 0: the instance compiled for (d, c), 1 if there is none; fixed == 0: the run-time instance
extern "C" int mvh_batch2(int d, int c, const MvArgs *g, int fixed) {
  if (!fixed) {
    if (d < 1 || d > MV_DMAX || c < 2 || c > MV_CMAX) return 1;
    MvArgs a = *g;
    a.d = d;
    a.c = c;
    run_rt(a);
    return 0;
  }
#define CASE(DD, CC) if (d == DD && c == CC) { run<DD, CC>(*g); return 0; }
  CASE(1, 2) CASE(2, 2) CASE(3, 2) CASE(4, 2) CASE(5, 2)
  CASE(1, 3) CASE(2, 3) CASE(3, 3) CASE(4, 3) CASE(5, 3)
  CASE(1, 4) CASE(2, 4) CASE(3, 4) CASE(4, 4) CASE(5, 4)
  CASE(1, 5) CASE(2, 5) CASE(3, 5) CASE(1, 6) CASE(2, 6) CASE(3, 6) CASE(1, 7) CASE(2, 7) CASE(3, 7)
#undef CASE
  return 1;
}
extern "C" int mvh_batch(int d, int c, const MvArgs *g) { return mvh_batch2(d, c, g, 1); }

// the null-model fit (run-time instance): out as MvNullArgs::out
extern "C" int mvh_null_rt(const MvNullArgs *a) {
  std::vector<double> scratch((size_t)MvNrLayout(a->g.d, a->g.c).DOUBLES);
  mv_null_fit<0, 0, HostLanes>(*a, scratch.data());
  return 0;
}
extern "C" size_t mvh_args_size(void) { return sizeof(MvArgs); }
extern "C" size_t mvh_null_args_size(void) { return sizeof(MvNullArgs); }
