// What does hipMalloc of the eigensolver's workspace cost (n = 50 000: ~30 buffers, 140 GB), and would ONE slab be cheaper?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipFree(0);
  const size_t GB = 1ull << 30;
  for (int rep = 0; rep < 2; ++rep) {
    { // seven n^2 buffers of 20 GB, one after the other
      std::vector<void *> p(7);
      double t0 = now();
      for (auto &q : p) if (hipMalloc(&q, 20 * GB) != hipSuccess) { printf("malloc failed\n"); return 1; }
      double t1 = now();
      for (auto &q : p) hipMemsetAsync(q, 0, 4096, 0);
      hipDeviceSynchronize();
      double t2 = now();
      for (auto &q : p) hipFree(q);
      double t3 = now();
      printf("rep %d: 7 x 20 GB: malloc %.3f s, first touch %.3f s, free %.3f s\n", rep, t1 - t0, t2 - t1, t3 - t2);
    }
    { // one slab of 140 GB
      void *q;
      double t0 = now();
      if (hipMalloc(&q, 140 * GB) != hipSuccess) { printf("slab failed\n"); return 1; }
      double t1 = now();
      hipMemsetAsync(q, 0, 4096, 0);
      hipDeviceSynchronize();
      double t2 = now();
      hipFree(q);
      double t3 = now();
      printf("rep %d: 1 x 140 GB: malloc %.3f s, first touch %.3f s, free %.3f s\n", rep, t1 - t0, t2 - t1, t3 - t2);
    }
    { // 70 x 2 GB
      std::vector<void *> p(70);
      double t0 = now();
      for (auto &q : p) if (hipMalloc(&q, 2 * GB) != hipSuccess) { printf("malloc failed\n"); return 1; }
      double t1 = now();
      for (auto &q : p) hipFree(q);
      double t2 = now();
      printf("rep %d: 70 x 2 GB: malloc %.3f s, free %.3f s\n", rep, t1 - t0, t2 - t1);
    }
    { // memset of a whole 20 GB buffer (what the solver does to VT first)
      void *q;
      hipMalloc(&q, 20 * GB);
      double t0 = now();
      hipMemsetAsync(q, 0, 20 * GB, 0);
      hipDeviceSynchronize();
      double t1 = now();
      hipMemsetAsync(q, 0, 20 * GB, 0);
      hipDeviceSynchronize();
      double t2 = now();
      hipFree(q);
      printf("rep %d: memset 20 GB: first %.3f s, second %.3f s\n", rep, t1 - t0, t2 - t1);
    }
  }
  return 0;
}
