// Centred kinship of hard-call genotypes (-gk 1, PLINK 2-bit input) with the big product done in exact integers.
//
// BimbamKin / PlinkKin (GEMMA src/gemma_io.cpp:1487-1538 / :1651-1704) form, per SNP s over ALL individuals, the row
// c_s = g_s - mu_s with missing calls set to the mean mu_s (i.e. to 0 after centring), and accumulate K += C^T C.  Write
// g_si in {0,1,2} (0 where the call is missing), m_si the 0/1 missing mask, d_si = g_si - mu_s (so d = -mu_s at a missing
// call): c_si = d_si (1 - m_si), and
//
//     sum_s c_si c_sj = (G^T G)_ij - a_i - a_j + sum_s mu_s^2 + S_ij + S_ji,
//     a_i  = sum_s mu_s g_si,
//     S_ji = sum_{s : m_sj = 1} mu_s d'_si,      d'_si = g_si - mu_s (1 - m_si / 2)
//
// (the m/2 makes the pairs that are BOTH missing at s come out right under the symmetrisation; the identity is pinned
// against the reference restatement in tests/test_round2_groundwork.py).  G^T G is a product of 2-bit integers: exact in
// int32 on v_mfma_i32_32x32x32_i8 (i8gemm_packed_kernel with the transposed block as both operands), ~20x the rate of the
// fp64 SYRK it replaces.  S touches only the missing calls (1 % of the entries): one block per (individual j, range of
// i) lists the SNPs at which j is missing -- in SNP order, from j's row of the transposed block: deterministic, no atomics
// -- and adds mu_s d'_s over its range of i from the SNP-major packed rows.  Everything is accumulated in fp64 across the
// blocks of a run (the integer sums stay below 2^53) and folded into K at kin_end.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ingest.hip.h"

namespace gemma_hip {

constexpr int KI8_SEG = 4096;  // individuals i per block of the correction kernel (16 per thread)
constexpr int KI8_LIST = 4096; // SNPs listed per pass (LDS); a block walks the SNP axis in ranges of this many rows

// A (l x ldk, SNP-major, byte = g | m << 4) -> At (individual-major, byte as A) and Gt (individual-major, byte = g):
// rows i < n_rows_out (zero beyond n), ldl bytes per row (zero beyond l)
__global__ void kin_i8_transpose_kernel(const int8_t *__restrict__ A, long l, long ldk, long n, int8_t *__restrict__ At,
                                        int8_t *__restrict__ Gt, long ldl, long rows_out) {
  __shared__ int8_t tile[64][65];
  const long s0 = (long)blockIdx.x * 64, i0 = (long)blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6; // 64 x 4
  for (int r = ty; r < 64; r += 4) {
    const long s = s0 + r, i = i0 + tx;
    tile[r][tx] = (s < l && i < n) ? A[s * ldk + i] : (int8_t)0;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const long i = i0 + r, s = s0 + tx;
    if (i < rows_out && s < ldl) {
      const int8_t b = tile[tx][r];
      At[i * ldl + s] = b;
      Gt[i * ldl + s] = (int8_t)(b & 3);
    }
  }
}

// GtG_acc (n x n fp64) += C (int32, ldc), upper and lower alike (the product is symmetric)
__global__ void kin_i8_accum_kernel(const int *__restrict__ C, long ldc, long n, double *__restrict__ acc) {
  const long j = (long)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  for (long i = blockIdx.y; i < n; i += gridDim.y) acc[i * n + j] += (double)C[i * ldc + j];
}

struct KinCorrArgs {
  const int8_t *A;   // l x ldk, SNP-major packed (g | m << 4)
  const int8_t *At;  // individual-major packed, ldl bytes per row
  const double *mean; // l
  long l, ldk, ldl, n;
  double *S;     // n x n, row j: S[j][i]
  double *a;     // n
  double *smu2;  // 1
};

// grid (n individuals j, ceil(n / KI8_SEG) ranges of i), 256 threads
__global__ __launch_bounds__(256) void kin_i8_corr_kernel(KinCorrArgs g) {
  __shared__ int list[KI8_LIST];
  __shared__ int wcount[4];
  __shared__ double red[4];
  const long j = blockIdx.x;
  const long i0 = (long)blockIdx.y * KI8_SEG;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int8_t *rowj = g.At + j * g.ldl;
  double acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0;
  double a_part = 0.0, mu2_part = 0.0, cj = 0.0;
  for (long r0 = 0; r0 < g.l; r0 += KI8_LIST) {
    const long r1 = r0 + KI8_LIST < g.l ? r0 + KI8_LIST : g.l;
    // the SNPs of [r0, r1) at which j is missing, in SNP order
    int base = 0;
    for (long c0 = r0; c0 < r1; c0 += 256) {
      const long s = c0 + t;
      int8_t b = 0;
      double mu = 0.0;
      if (s < r1) {
        b = rowj[s];
        if (blockIdx.y == 0) {
          mu = g.mean[s];
          a_part += mu * (double)(b & 3);
          if (j == 0) mu2_part += mu * mu;
        }
      }
      const bool miss = (b >> 4) & 1;
      const unsigned long long bal = __ballot(miss);
      if (lane == 0) wcount[wave] = __popcll(bal);
      __syncthreads();
      int off = base;
      for (int w = 0; w < wave; ++w) off += wcount[w];
      if (miss) list[off + __popcll(bal & ((1ull << lane) - 1ull))] = (int)s;
      base += wcount[0] + wcount[1] + wcount[2] + wcount[3];
      __syncthreads();
    }
    // S[j][i] += mu_s d'_si over the listed SNPs, i = i0 + 16 t .. + 15 (one 16-byte load per thread and SNP: the
    // block reads 4 KiB of the SNP's packed row).  mu d' = mu g + (mu^2 / 2) m - mu^2: two bit-field extracts, two
    // conversions and two FMAs per call (the stage is VALU-bound -- a chain of 64-bit selects costs twice that); the
    // constant -sum mu_s^2 over the list is the same for every i and comes off at the end.
    const long ib = i0 + 16 * t;
    if (ib < g.ldk) { // ldk is a multiple of 128 >= n: a whole 16-byte group lies inside the zero-padded row
      const int8_t *colbase = g.A + ib;
#define KI8_APPLY(W4, MU)                                                                                          \
  do {                                                                                                            \
    const double h_ = 0.5 * (MU) * (MU);                                                                          \
    cj += (MU) * (MU);                                                                                            \
    const unsigned int ww_[4] = {(W4).x, (W4).y, (W4).z, (W4).w};                                                 \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) _Pragma("unroll") for (int k = 0; k < 4; ++k) {                 \
      const double gq_ = (double)__builtin_amdgcn_ubfe(ww_[q], 8 * k, 2);                                         \
      const double mq_ = (double)__builtin_amdgcn_ubfe(ww_[q], 8 * k + 4, 1);                                     \
      acc[4 * q + k] = fma((MU), gq_, acc[4 * q + k]);                                                            \
      acc[4 * q + k] = fma(h_, mq_, acc[4 * q + k]);                                                              \
    }                                                                                                             \
  } while (0)
      int e = 0;
      for (; e + 4 <= base; e += 4) { // four independent 16-byte loads in flight per thread
        const int s0 = list[e], s1 = list[e + 1], s2 = list[e + 2], s3 = list[e + 3];
        const uint4 w0 = *reinterpret_cast<const uint4 *>(colbase + (long)s0 * g.ldk);
        const uint4 w1 = *reinterpret_cast<const uint4 *>(colbase + (long)s1 * g.ldk);
        const uint4 w2 = *reinterpret_cast<const uint4 *>(colbase + (long)s2 * g.ldk);
        const uint4 w3 = *reinterpret_cast<const uint4 *>(colbase + (long)s3 * g.ldk);
        const double m0 = g.mean[s0], m1 = g.mean[s1], m2 = g.mean[s2], m3 = g.mean[s3];
        KI8_APPLY(w0, m0);
        KI8_APPLY(w1, m1);
        KI8_APPLY(w2, m2);
        KI8_APPLY(w3, m3);
      }
      for (; e < base; ++e) {
        const int s0 = list[e];
        const uint4 w0 = *reinterpret_cast<const uint4 *>(colbase + (long)s0 * g.ldk);
        const double m0 = g.mean[s0];
        KI8_APPLY(w0, m0);
      }
#undef KI8_APPLY
    }
    __syncthreads();
  }
  double *Sj = g.S + j * g.n;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const long i = i0 + 16 * t + q;
    if (i < g.n) Sj[i] += acc[q] - cj;
  }
  if (blockIdx.y == 0) {
    double v = wsum(a_part);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (t == 0) g.a[j] += ((red[0] + red[1]) + red[2]) + red[3];
    if (j == 0) {
      __syncthreads();
      v = wsum(mu2_part);
      if (lane == 0) red[wave] = v;
      __syncthreads();
      if (t == 0) g.smu2[0] += ((red[0] + red[1]) + red[2]) + red[3];
    }
  }
}

// K (upper triangle incl. diagonal, unscaled sums) += GtG - a_i - a_j + sum mu^2 + S_ij + S_ji
__global__ void kin_i8_fold_kernel(double *__restrict__ K, long n, const double *__restrict__ GtG,
                                   const double *__restrict__ S, const double *__restrict__ a,
                                   const double *__restrict__ smu2) {
  __shared__ double tile[32][33];
  const int bx = blockIdx.x, by = blockIdx.y; // upper-triangular 32 x 32 tile pairs
  if (bx < by) return;
  const int tx = threadIdx.x, ty = threadIdx.y; // 32 x 8
  for (int r = ty; r < 32; r += 8) { // S^T tile: S[j][i] for j in bx-range, i in by-range
    const long jj = (long)bx * 32 + r, ii = (long)by * 32 + tx;
    tile[r][tx] = (jj < n && ii < n) ? S[jj * n + ii] : 0.0;
  }
  __syncthreads();
  const double m2 = smu2[0];
  for (int r = ty; r < 32; r += 8) {
    const long i = (long)by * 32 + r, j = (long)bx * 32 + tx;
    if (i < n && j < n && j >= i) K[i * n + j] += GtG[i * n + j] - a[i] - a[j] + m2 + S[i * n + j] + tile[tx][r];
  }
}

} // namespace gemma_hip
