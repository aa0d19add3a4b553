// TEST INFRASTRUCTURE: runs gemma_amd/csrc/mvlmm.hip.h with one CPU "lane" so that the per-SNP logic of the HIP kernel
// can be compared with oracle/mvlmm_oracle.c where no GPU exists.  Not part of the shipped library.
#include <cstddef>
#include <cstring>
#define MV_HD inline
#include "../../gemma_amd/csrc/mvlmm.hip.h"
using namespace gemma_hip;

struct HostLanes {
  static constexpr int N = 1;
  static int lane() { return 0; }
  static double sum(double v) { return v; }
};

template <int D, int C> static void run(const MvArgs &g) {
  MvNr<D, C, HostLanes> nr{g, nullptr};
  static double scratch[MvNrScratch<D, C>::DOUBLES];
  nr.lds = scratch;
  for (long s = 0; s < g.l; ++s) {
    nr.x = g.UtX + s * g.ld;
    mv_one_snp<D, C, HostLanes>(g, s, nr);
  }
}

extern "C" int mvh_batch(int d, int c, const MvArgs *g) {
#define CASE(DD, CC) if (d == DD && c == CC) { run<DD, CC>(*g); return 0; }
  CASE(1, 2) CASE(2, 2) CASE(3, 2) CASE(4, 2) CASE(5, 2)
  CASE(1, 3) CASE(2, 3) CASE(3, 3) CASE(4, 3) CASE(5, 3)
  CASE(1, 4) CASE(2, 4) CASE(3, 4) CASE(4, 4) CASE(5, 4)
  CASE(1, 5) CASE(2, 5) CASE(3, 5) CASE(1, 6) CASE(2, 6) CASE(3, 6) CASE(1, 7) CASE(2, 7) CASE(3, 7)
#undef CASE
  return 1;
}
extern "C" size_t mvh_args_size(void) { return sizeof(MvArgs); }
