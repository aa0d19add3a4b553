// Prototype (not part of the library) of the missing-mask product of the exact int8 U^T x on the 2:4 structured-sparse MFMA
// v_smfmac_i32_32x32x64_i8, with the data layout DESIGN.md 8 (item 1) proposes for round 3:
//   * the genotype product stays on the dense v_mfma_i32_32x32x32_i8 (two per 64 bytes of K and block pair);
//   * the mask travels as one 8-byte word per (SNP row, 32 individuals): the index nibbles of its 8 groups of four and 16
//     "kept" bits (bit 2g: at least one missing call in group g, bit 2g+1: at least two); the 16 kept bytes of the sparse A
//     operand are spread from those bits with one 24-bit multiply per 4 bytes;
//   * lane (row, half h) of the sparse instruction takes the word of K-step ks + h of the pair (ks, ks+1); the B operand is
//     the pair of dense B fragments the lane already holds for the two steps (the K order inside the 64 bytes is chosen so);
//   * a group with three or four missing calls cannot be expressed: the row is flagged and the surplus calls are listed for
//     an fp64 fix-up outside the product.
// One wavefront, operands straight from global memory: this checks the ENCODING against a CPU sum, not the speed.
//   hipcc --offload-arch=gfx950 -O2 scripts/i8_sparse_proto.hip -o /tmp/i8_sparse_proto && /tmp/i8_sparse_proto
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int R = 32, C = 32, K = 512; // one 32 x 32 output block, K a multiple of 64

struct Meta {
  unsigned idx, bits;
};

// A: R x K packed bytes (g | m << 4), Bt: C x K digit bytes, meta: R x (K / 32)
__global__ void proto_kernel(const signed char *A, const signed char *Bt, const Meta *meta, int *Cg, int *Cm) {
  const int l = threadIdx.x, r32 = l & 31, h = l >> 5;
  v16i accg = {0}, accm = {0};
  for (int k0 = 0; k0 < K; k0 += 64) {
    // dense fragments of the two K-steps of the pair: lane half h holds k = 16 h .. 16 h + 15 of each 32-byte step
    const v4i a0 = *reinterpret_cast<const v4i *>(A + r32 * K + k0 + 16 * h);
    const v4i a1 = *reinterpret_cast<const v4i *>(A + r32 * K + k0 + 32 + 16 * h);
    const v4i b0 = *reinterpret_cast<const v4i *>(Bt + r32 * K + k0 + 16 * h);
    const v4i b1 = *reinterpret_cast<const v4i *>(Bt + r32 * K + k0 + 32 + 16 * h);
    const v4i mg = {0x03030303, 0x03030303, 0x03030303, 0x03030303};
    accg = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0 & mg, b0, accg, 0, 0, 0);
    accg = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1 & mg, b1, accg, 0, 0, 0);
    // sparse mask operand: the word of step (k0 / 32) + h
    const Meta m = meta[r32 * (K / 32) + k0 / 32 + h];
    v4i ms;
#pragma unroll
    for (int d = 0; d < 4; ++d) ms[d] = (int)((((m.bits >> (4 * d)) & 0xFu) * 0x00204081u) & 0x01010101u);
    const v8i b8 = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    accm = __builtin_amdgcn_smfmac_i32_32x32x64_i8(ms, b8, accm, (int)m.idx, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    Cg[row * C + r32] = accg[r];
    Cm[row * C + r32] = accm[r];
  }
}

int main() {
  std::vector<signed char> A(R * K), Bt(C * K);
  std::vector<unsigned char> g(R * K), mk(R * K);
  std::vector<Meta> meta(R * (K / 32));
  srand(7);
  for (int i = 0; i < R * K; ++i) {
    mk[i] = (rand() % 100) < 12; // 12 % missing: groups with 3 or 4 missing calls do occur
    g[i] = mk[i] ? 0 : (unsigned char)(rand() % 3);
    A[i] = (signed char)(g[i] | (mk[i] << 4));
  }
  for (int i = 0; i < C * K; ++i) Bt[i] = (signed char)((rand() % 256) - 128);
  // ingest side: meta words and the surplus list
  struct Surplus {
    int row, k;
  };
  std::vector<Surplus> surplus;
  int groups_over = 0;
  for (int r = 0; r < R; ++r)
    for (int w = 0; w < K / 32; ++w) {
      Meta m{0, 0};
      for (int gq = 0; gq < 8; ++gq) {
        int pos[4], cnt = 0;
        for (int q = 0; q < 4; ++q)
          if (mk[r * K + 32 * w + 4 * gq + q]) pos[cnt++] = q;
        const int p0 = cnt >= 1 ? pos[0] : 0, p1 = cnt >= 2 ? pos[1] : (p0 == 3 ? 2 : 3);
        m.idx |= (unsigned)(p0 | (p1 << 2)) << (4 * gq);
        m.bits |= (unsigned)((cnt >= 1) | ((cnt >= 2) << 1)) << (2 * gq);
        if (cnt > 2) ++groups_over;
        for (int q = 2; q < cnt; ++q) surplus.push_back({r, 32 * w + 4 * gq + pos[q]});
      }
      meta[r * (K / 32) + w] = m;
    }
  signed char *dA, *dB;
  Meta *dM;
  int *dCg, *dCm;
  (void)hipMalloc(&dA, A.size());
  (void)hipMalloc(&dB, Bt.size());
  (void)hipMalloc(&dM, meta.size() * sizeof(Meta));
  (void)hipMalloc(&dCg, R * C * 4);
  (void)hipMalloc(&dCm, R * C * 4);
  (void)hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, Bt.data(), Bt.size(), hipMemcpyHostToDevice);
  (void)hipMemcpy(dM, meta.data(), meta.size() * sizeof(Meta), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(proto_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dM, dCg, dCm);
  std::vector<int> Cg(R * C), Cm(R * C);
  (void)hipMemcpy(Cg.data(), dCg, R * C * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(Cm.data(), dCm, R * C * 4, hipMemcpyDeviceToHost);
  // the fix-up the combine step would do: add the surplus calls
  for (const Surplus &s : surplus)
    for (int c = 0; c < C; ++c) Cm[s.row * C + c] += Bt[c * K + s.k];
  long bad_g = 0, bad_m = 0;
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c) {
      int sg = 0, sm = 0;
      for (int k = 0; k < K; ++k) {
        sg += (int)g[r * K + k] * Bt[c * K + k];
        sm += (int)mk[r * K + k] * Bt[c * K + k];
      }
      bad_g += sg != Cg[r * C + c];
      bad_m += sm != Cm[r * C + c];
    }
  printf("genotype product (dense): %ld of %d entries differ\n", bad_g, R * C);
  printf("mask product (sparse + %zu surplus calls from %d groups with more than two missing): %ld of %d entries differ\n",
         surplus.size(), groups_over, bad_m, R * C);
  return (bad_g || bad_m) ? 1 : 0;
}
