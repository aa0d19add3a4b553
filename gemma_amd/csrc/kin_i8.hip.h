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

// GtG_acc (n x n fp64) += C (int32, ldc)
// (rows i <= the last column of the block only: the product is formed for the tiles that meet the upper triangle, and
// kin_i8_fold_kernel reads nothing below the diagonal)
__global__ void kin_i8_accum_kernel(const int *__restrict__ C, long ldc, long n, double *__restrict__ acc) {
  const long j = (long)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const long iend = ((long)blockIdx.x + 1) * 256 < n ? ((long)blockIdx.x + 1) * 256 : n;
  for (long i = blockIdx.y; i < iend; i += gridDim.y) acc[i * n + j] += (double)C[i * ldc + j];
}

struct KinCorrArgs {
  const int8_t *A;   // l x ldk, SNP-major packed (g | m << 4)
  const int8_t *At;  // individual-major packed, ldl bytes per row
  const double *mean; // l
  long l, ldk, ldl, n;
  double *S;     // n x n, row j: S[j][i]
  double *a;     // n
  double *smu2;  // 1
  const int *lists_ok; // device flag of kin_i8_scan_kernel (nullptr: always run)
};

// grid (n individuals j, ceil(n / KI8_SEG) ranges of i), 256 threads
__global__ __launch_bounds__(256) void kin_i8_corr_kernel(KinCorrArgs g) {
  if (g.lists_ok && g.lists_ok[0]) return; // this block of SNPs went through kin_i8_corr2_kernel (lists of the missing calls)
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

// ---------------------------------------------------------------------------------------------------------------
// Round 3: the correction pass on LISTS of the missing calls.
//
// kin_i8_corr_kernel above spends six vector operations per (missing call of j, individual i) pair -- two bit-field
// extracts, two conversions, two FMAs -- and rebuilds j's list of SNPs in every block (one dependent byte load per 256 SNPs).
// Half of that arithmetic is the term (mu_s^2 / 2) m_si, which is non-zero only where i is missing at s as well: 1 % of
// the pairs.  Here the missing calls of a block are listed once, both ways (CSR):
//     listJ: per individual j the SNPs s with m_sj = 1, ascending       (kin_i8_count / scan / fill on At)
//     listS: per SNP s the individuals i with m_si = 1, ascending        (the same kernels on A), with the positions at
//            which each list crosses a multiple of KI8_SEG (kin_i8_sub_kernel)
// and the correction block (j, range of i) runs
//     * the genotype term mu_s g_si over listJ[j] from a 2-BIT copy of the block (16 individuals per dword: a quarter of the
//       bytes the packed rows cost -- 80 GB per 20 000 x 20 000 block otherwise): extract, convert, FMA = 3 operations a pair;
//     * the both-missing term as integer additions of round(2^44 mu_s^2 / 2) into an LDS row over listS[s] for s in listJ[j]
//       (order-independent, hence deterministic with LDS atomics; |error| <= 2^-45 per term against sums of magnitude p).
// a_j = sum_s mu_s g_sj and c_j = sum_{s in listJ[j]} mu_s^2 come out of the fill pass over At, sum_s mu_s^2 out of the scan.
// A block whose lists would not fit their buffers (more than 1/16 of the calls missing) keeps the kernel above: the scan
// kernel decides on the device (flag ok), both correction kernels are launched and one of them returns at once.

// A (l x ldk bytes, g | m << 4) -> A2 (l x 256 nseg dwords): dword 256 seg + t of a row holds the individuals
// KI8_SEG seg + 256 q + t, q = 0 .. 15, at bits 2 q .. 2 q + 1 -- thread t of the correction block (j, seg) owns exactly
// these, so that its sixteen accumulators go to memory as sixteen coalesced rows of 256 doubles
__global__ __launch_bounds__(256) void kin_i8_pack2_kernel(const int8_t *__restrict__ A, long l, long ldk, int nseg,
                                                           unsigned *__restrict__ A2) {
  const int t = threadIdx.x, seg = blockIdx.y;
  const long s = blockIdx.x;
  const int8_t *row = A + s * ldk + (long)seg * KI8_SEG + t;
  unsigned out = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const long i = (long)seg * KI8_SEG + 256 * q + t;
    const unsigned b = i < ldk ? (unsigned)(unsigned char)row[256 * q] : 0u;
    out |= (b & 3u) << (2 * q);
  }
  A2[s * (256L * nseg) + 256 * seg + t] = out;
}

// the sixteen "missing" bits of sixteen packed bytes
__device__ __forceinline__ unsigned ki8_mask16(const uint4 w) {
  const unsigned ww[4] = {w.x, w.y, w.z, w.w};
  unsigned mk = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned m = (ww[q] >> 4) & 0x01010101u;
    mk |= ((m | (m >> 7) | (m >> 14) | (m >> 21)) & 0xFu) << (4 * q);
  }
  return mk;
}

// cnt[r] = missing calls in row r of M (rows x ld bytes, len <= ld bytes looked at, len a multiple of 16); one wavefront per row
__global__ __launch_bounds__(256) void kin_i8_count_kernel(const int8_t *__restrict__ M, long rows, long len, long ld,
                                                           int *__restrict__ cnt) {
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  const uint4 *row = reinterpret_cast<const uint4 *>(M + r * ld);
  const long nq = len / 16;
  int c = 0;
  for (long q = lane; q < nq; q += 64) c += __popc(ki8_mask16(row[q]));
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
  if (lane == 0) cnt[r] = c;
}

// exclusive scans of the two count arrays (one block of 1024 threads), the decision whether the lists fit, sum mu^2
struct KinScanArgs {
  const int *cntS, *cntJ;
  int *offS, *offJ; // l + 1, n + 1
  long l, n;
  long cap;         // entries either list buffer holds
  int *ok;          // out: 1 = lists are valid
  const double *mean;
  double *smu2;     // += sum mu_s^2 when ok
};
__device__ __forceinline__ long ki8_block_scan_1024(long v, long *sh /* 16 */, long &total) {
  // exclusive prefix of v over the 1024 threads of the block; total = the sum
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  long inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const long o = __shfl_up(inc, off, 64);
    if (lane >= off) inc += o;
  }
  __syncthreads();
  if (lane == 63) sh[wave] = inc;
  __syncthreads();
  long base = 0;
  for (int w = 0; w < wave; ++w) base += sh[w];
  long tot = 0;
  for (int w = 0; w < 16; ++w) tot += sh[w];
  total = tot;
  return base + inc - v;
}
__global__ __launch_bounds__(1024) void kin_i8_scan_kernel(KinScanArgs g) {
  __shared__ long sh[16];
  __shared__ double shd[16];
  const int t = threadIdx.x;
  long totals[2];
  for (int which = 0; which < 2; ++which) {
    const int *cnt = which ? g.cntJ : g.cntS;
    int *off = which ? g.offJ : g.offS;
    const long N = which ? g.n : g.l;
    const long per = (N + 1023) / 1024, a0 = (long)t * per, a1 = a0 + per < N ? a0 + per : N;
    long mine = 0;
    for (long q = a0; q < a1; ++q) mine += cnt[q];
    long total;
    long run = ki8_block_scan_1024(mine, sh, total);
    totals[which] = total;
    if (total <= g.cap)
      for (long q = a0; q < a1; ++q) {
        off[q] = (int)run;
        run += cnt[q];
      }
    if (t == 0) off[N] = (int)(total <= g.cap ? total : 0);
    __syncthreads();
  }
  const bool ok = totals[0] <= g.cap && totals[1] <= g.cap;
  if (t == 0) g.ok[0] = ok ? 1 : 0;
  if (ok) {
    double v = 0.0;
    for (long q = t; q < g.l; q += 1024) v += g.mean[q] * g.mean[q];
    v = wsum(v);
    if ((t & 63) == 0) shd[t >> 6] = v;
    __syncthreads();
    if (t == 0) {
      double tot = 0.0;
      for (int w = 0; w < 16; ++w) tot += shd[w];
      g.smu2[0] += tot;
    }
  }
}

// list[off[r] ..] = positions of the missing calls of row r, ascending; BYIDV (rows = individuals of At): also
// a[r] += sum_s mu_s g_sr and cj[r] = sum_{s missing} mu_s^2 (mean has l entries)
template <bool BYIDV>
__global__ __launch_bounds__(256) void kin_i8_fill_kernel(const int8_t *__restrict__ M, long rows, long len, long ld,
                                                          const int *__restrict__ off, int *__restrict__ list,
                                                          const int *__restrict__ ok, const double *__restrict__ mean,
                                                          long l, double *__restrict__ a, double *__restrict__ cj) {
  if (!ok[0]) return;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  const uint4 *row = reinterpret_cast<const uint4 *>(M + r * ld);
  const long nq = len / 16;
  int base = off[r];
  double asum = 0.0, csum = 0.0;
  for (long q0 = 0; q0 < nq; q0 += 64) {
    const long q = q0 + lane;
    uint4 w = make_uint4(0, 0, 0, 0);
    if (q < nq) w = row[q];
    unsigned mk = ki8_mask16(w);
    const int c = __popc(mk);
    int inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(inc, o, 64);
      if (lane >= o) inc += v;
    }
    int pos = base + inc - c;
    const int p0 = (int)(q * 16);
    if (BYIDV) { // lane-strided bytes and means (coalesced; the row's bytes were just read as uint4: L1 hits)
      const int8_t *rb = M + r * ld;
#pragma unroll 4
      for (int e = 0; e < 16; ++e) {
        const long s = q0 * 16 + 64 * e + lane;
        if (s < l && s < len) {
          const unsigned b = (unsigned)(unsigned char)rb[s];
          const double mu = mean[s];
          asum = fma(mu, (double)(b & 3u), asum);
          csum = fma(mu * mu, (double)((b >> 4) & 1u), csum);
        }
      }
    }
    while (mk) {
      const int e = __ffs(mk) - 1;
      list[pos++] = p0 + e;
      mk &= mk - 1;
    }
    base += __shfl(inc, 63, 64);
  }
  if (BYIDV) {
    asum = wsum(asum);
    csum = wsum(csum);
    if (lane == 0) {
      a[r] += asum;
      cj[r] = csum;
    }
  }
}

// sub[s (nseg + 1) + b] = entries of listS[s] below b KI8_SEG (b = 0 .. nseg)
__global__ __launch_bounds__(256) void kin_i8_sub_kernel(const int *__restrict__ offS, const int *__restrict__ listS, long l,
                                                         int nseg, const int *__restrict__ ok, int *__restrict__ sub) {
  if (!ok[0]) return;
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= l * (nseg + 1)) return;
  const long s = id / (nseg + 1);
  const int b = (int)(id - s * (nseg + 1));
  const int *lst = listS + offS[s];
  const int cnt = offS[s + 1] - offS[s];
  const int bound = b * KI8_SEG;
  int lo = 0, hi = cnt; // first position with lst[pos] >= bound
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (lst[mid] < bound) lo = mid + 1;
    else hi = mid;
  }
  sub[id] = lo;
}

constexpr double KI8_FIX = 17592186044416.0; // 2^44: fixed point of the both-missing term
struct KinCorr2Args {
  const unsigned *A2; // l x ld2 dwords, ld2 = 256 nseg
  long ld2;
  const double *mean;
  long n;
  const int *offJ, *listJ, *offS, *listS, *sub;
  int nseg;
  const double *cj;
  double *S; // n x n, row j: S[j][i]
  const int *ok;
  int dbg_skip_pairs; // timing experiments only (GEMMA_HIP_KIN_DBG=1): the both-missing term left out, results wrong
  int dbg_skip_main;  // (GEMMA_HIP_KIN_DBG=2): the genotype term left out
};
// grid (n individuals j, ceil(n / KI8_SEG) ranges of i), 256 threads: thread t owns i = i0 + 256 q + t, q = 0 .. 15 (one dword of A2)
__global__ __launch_bounds__(256) void kin_i8_corr2_kernel(KinCorr2Args g) {
  __shared__ unsigned long long trow[KI8_SEG];
  if (!g.ok[0]) return;
  const long j = blockIdx.x;
  const int seg = blockIdx.y, t = threadIdx.x;
  const long i0 = (long)seg * KI8_SEG;
#pragma unroll
  for (int q = 0; q < KI8_SEG / 256; ++q) trow[t + 256 * q] = 0ull;
  const int lo = g.offJ[j], cnt = g.offJ[j + 1] - lo;
  const int *lst = g.listJ + lo;
  double acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0;
  const unsigned *col = g.A2 + 256 * seg + t; // individuals i0 + 256 q + t, q = 0 .. 15
#define KI8_APPLY2(W, MU)                                                                                          \
  do {                                                                                                            \
    _Pragma("unroll") for (int q = 0; q < 16; ++q)                                                                \
      acc[q] = fma((MU), (double)__builtin_amdgcn_ubfe((W), 2 * q, 2), acc[q]);                                   \
  } while (0)
  int e = g.dbg_skip_main ? cnt : 0;
  for (; e + 8 <= cnt; e += 8) { // eight independent loads in flight per thread; the list and the means are wave-uniform
    int sv[8];
    unsigned wv[8];
    double mv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) sv[u] = lst[e + u];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      wv[u] = col[(long)sv[u] * g.ld2];
      mv[u] = g.mean[sv[u]];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) KI8_APPLY2(wv[u], mv[u]);
  }
  for (; e < cnt; ++e) {
    const int s0 = lst[e];
    const unsigned w0 = col[(long)s0 * g.ld2];
    const double m0 = g.mean[s0];
    KI8_APPLY2(w0, m0);
  }
#undef KI8_APPLY2
  __syncthreads(); // trow is cleared
  // both missing: for s in listJ[j], the i of listS[s] inside this range get round(2^44 mu_s^2 / 2).  A wavefront takes 64
  // entries of listJ[j] at a time: one lane per entry fetches where listS[s] crosses this range (sub), then the wavefront walks
  // the entries one after the other with its lanes ALONG listS[s] (one coalesced load per SNP: a thread per SNP would pull a
  // cache line per call), four SNPs in flight
  if (!g.dbg_skip_pairs) {
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    for (int e0 = 64 * wave; e0 < cnt; e0 += 256) {
      const int e = e0 + lane;
      int b0 = 0, b1 = 0, off = 0;
      unsigned long long h = 0ull;
      if (e < cnt) {
        const int s = lst[e];
        const int *sb = g.sub + (long)s * (g.nseg + 1) + seg;
        b0 = sb[0];
        b1 = sb[1];
        off = g.offS[s];
        const double mu = g.mean[s];
        h = (unsigned long long)__double2ll_rn(0.5 * mu * mu * KI8_FIX);
      }
      const int cntw = cnt - e0 < 64 ? cnt - e0 : 64;
      for (int u = 0; u < cntw; u += 4) {
        int ii[4], ub1[4], uk[4], uoff[4];
        unsigned long long hh[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int uu = (u + q) & 63; // entries past cntw hold b0 = b1 = 0
          uoff[q] = __shfl(off, uu, 64);
          ub1[q] = __shfl(b1, uu, 64);
          uk[q] = __shfl(b0, uu, 64) + lane;
          hh[q] = (unsigned long long)__shfl((long long)h, uu, 64);
          ii[q] = uk[q] < ub1[q] ? g.listS[uoff[q] + uk[q]] : 0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (uk[q] < ub1[q]) atomicAdd(&trow[ii[q] - (int)i0], hh[q]);
#pragma unroll
        for (int q = 0; q < 4; ++q) // a SNP with more than 64 missing calls in this range: the rest of its list
          for (int k = uk[q] + 64; k < ub1[q]; k += 64) atomicAdd(&trow[g.listS[uoff[q] + k] - (int)i0], hh[q]);
      }
    }
  }
  __syncthreads();
  double *Sj = g.S + j * g.n;
  const double cj = g.cj[j];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const long i = i0 + 256 * q + t;
    if (i < g.n) Sj[i] += acc[q] + (double)trow[256 * q + t] * (1.0 / KI8_FIX) - cj;
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
