// Host-side logic of two round-4 kernels' launch plans, run on the CPU (hipcc compiles it, nothing is launched):
//   s2_build_raster   (i8gemm_sparse2.hip.h): every tile exactly once, every XCD exactly the number of workgroups the hardware
//                     hands it (b % 8), the eight XCDs inside one band of tile rows at the same sequence position;
//   gemm_kslice_plan  (dgemm_mfma.hip.h): slices are non-empty multiples of the K-tile that add up to the whole-tile part of K.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include "i8gemm_sparse2.hip.h"
using namespace gemma_hip;

int main() {
  int bad = 0;
  const int dims[][2] = {{79, 157}, {20, 40}, {1, 1}, {7, 3}, {16, 16}, {196, 391}, {3, 100}, {79, 40}, {40, 79}, {8, 4}, {9, 5}};
  for (auto &d : dims)
    for (int rb : {1, 2, 4, 8})
      for (int pr : {8, 16, 4}) {
        std::vector<int2> m;
        s2_build_raster(d[0], d[1], rb, m, pr);
        const long total = (long)d[0] * d[1];
        std::set<long> seen;
        for (auto &t : m) {
          if (t.x < 0 || t.x >= d[0] || t.y < 0 || t.y >= d[1]) ++bad;
          seen.insert((long)t.x * 1000000 + t.y);
        }
        if ((long)seen.size() != total || (long)m.size() != total) {
          ++bad;
          printf("coverage %dx%d rb=%d pr=%d: %zu distinct of %ld\n", d[0], d[1], rb, pr, seen.size(), total);
        }
      }
  {
    // the point of the raster: at the same sequence position the eight XCDs of the default order (rb = 1, 8 x 4 patches) sit in ONE band
    // of 8 tile rows, in eight different 4-column blocks -- checked on the headline shape away from the ragged ends
    std::vector<int2> m;
    s2_build_raster(79, 157, 1, m);
    for (int o = 0; o < 32 * 30; o += 32) {
      std::set<int> bands, cblocks;
      for (int x = 0; x < 8; ++x) {
        bands.insert(m[(size_t)o * 8 + x].x / 8);
        cblocks.insert(m[(size_t)o * 8 + x].y / 4);
      }
      if (bands.size() != 1 || cblocks.size() != 8) {
        ++bad;
        printf("position %d: %zu bands, %zu column blocks\n", o, bands.size(), cblocks.size());
      }
    }
  }
  long checked = 0;
  for (long K = 1; K <= 40000; ++K)
    for (int want = 1; want <= 8; ++want) {
      long K0, per;
      const int ns = gemm_kslice_plan(K, want, &K0, &per);
      ++checked;
      if (K0 != K / GEMM_BK * GEMM_BK || ns < 1 || ns > want) { ++bad; continue; }
      if (ns == 1) continue;
      long sum = 0;
      for (int ks = 0; ks < ns; ++ks) {
        const long k0 = ks * per, kn = (K0 - k0 < per) ? K0 - k0 : per;
        if (kn <= 0 || kn % GEMM_BK) ++bad;
        sum += kn;
      }
      if (sum != K0 || per < 4 * GEMM_BK) ++bad;
    }
  printf("raster and K-slice plans: %ld slice plans checked, bad = %d\n", checked, bad);
  return bad ? 1 : 0;
}
