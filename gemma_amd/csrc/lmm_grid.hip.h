// Fixed-lambda evaluations of the per-SNP search as one skinny fp64 MFMA product per SNP batch.
//
// CalcLambda (GEMMA src/lmm.cpp:1950-2140) evaluates the likelihood derivative at the n_region + 1 grid values
// l_min * exp(i * log(l_max / l_min) / n_region) and the likelihood at l_min and l_max for EVERY SNP, and
// LogRL_f (:839-850) evaluates CalcPab with H = 1: 14 of the ~37 row passes per SNP of `-lmm 1` are at
// lambdas that do not depend on the SNP.  Everything those passes need from the SNP's row x = U^T x is
//     sum_i x_i^2 w_q(i),   sum_i x_i u_a(i) w_q(i)      (u_a: the U^T W columns and U^T y)
// for the SNP-independent weights w_0 = 1, w_{1+2g} = h_g, w_{2+2g} = h_g^2, h_g(i) = 1/(lambda_g delta_i + 1):
// one product [X.X | X] (l x n) by an n x (c+2)*NQ matrix of weights -- matrix-core work (0.8 ms of MFMA time at
// n = l = 20000, one read of UtX) instead of 14 streaming passes per SNP on the vector ALU.  The pairs that do
// not involve x (ww, wy, yy, traces) are computed once per lmm_setup.  The per-SNP kernel then assembles the
// CalcPab input of a grid evaluation from this table (FixedC::eval_grid) and runs the same recursion / formulas.
#pragma once
#include <hip/hip_runtime.h>
#include "dgemm_mfma.hip.h"
#include "lmm_assoc.hip.h"

namespace gemma_hip {

constexpr int GRID_FIX_LD = 16; // per weight: (c+1)(c+2)/2 <= 15 pair sums + the trace sum_i w_q(i)

struct GridGeom {
  int nq;    // weights: 1 + 2 * (n_region + 1)
  int nbx;   // 16-column MFMA blocks of the x.x group (nq columns)
  int nba;   // 16-column MFMA blocks of the x.u_a group ((c+1) * nq columns, a-major)
  int nc;    // 16-row K chunks: ceil(n / 16)
};

__device__ __forceinline__ double grid_weight(const double *__restrict__ eval, const double *lam_grid, int q, long k) {
  if (q == 0) return 1.0;
  const int gi = (q - 1) >> 1;
  const double h = recip(eval[k] * lam_grid[gi] + 1.0);
  return ((q - 1) & 1) ? h * h : h;
}

// Weight matrix in the MFMA B-operand order of grid_table_kernel:
//   Rp[chunk][cb][kq][col][j],  k = 16 * chunk + 4 * kq + j,  column cb * 16 + col  (zero beyond n / beyond the
//   used columns), so that lane (col, kq) of a wave reads its 4 K-steps of one chunk as 32 contiguous bytes.
__global__ void grid_weights_kernel(AssocArgs g, GridGeom gg, int c, double *__restrict__ Rp) {
  const long total = (long)gg.nc * (gg.nbx + gg.nba) * 256;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int j = (int)(e & 3), col = (int)((e >> 2) & 15), kq = (int)((e >> 6) & 3);
  const long r = e >> 8;
  const int nb = gg.nbx + gg.nba;
  const int cb = (int)(r % nb);
  const long chunk = r / nb;
  const long k = chunk * 16 + 4 * kq + j;
  double v = 0.0;
  if (k < g.n) {
    if (cb < gg.nbx) {
      const int q = cb * 16 + col;
      if (q < gg.nq) v = grid_weight(g.eval, g.lam_grid, q, k);
    } else {
      const int idx = (cb - gg.nbx) * 16 + col;
      const int a = idx / gg.nq, q = idx - a * gg.nq;
      if (a <= c) {
        const double u = (a < c) ? g.UtWt[(long)a * g.n + k] : g.Uty[k];
        v = u * grid_weight(g.eval, g.lam_grid, q, k);
      }
    }
  }
  Rp[e] = v;
}

// SNP-independent sums, one block per weight q:  F[q][pair(a, b)] = sum_i u_a u_b w_q  (a <= b over the c
// covariates then y, row-major upper triangle),  F[q][15] = sum_i w_q.
__global__ __launch_bounds__(256) void grid_fixed_kernel(AssocArgs g, int c, double *__restrict__ F) {
  const int q = blockIdx.x;
  const int nv = c + 1;
  double s[GRID_FIX_LD];
#pragma unroll
  for (int p = 0; p < GRID_FIX_LD; ++p) s[p] = 0.0;
  for (long i = threadIdx.x; i < g.n; i += 256) {
    const double w = grid_weight(g.eval, g.lam_grid, q, i);
    s[15] += w;
    int p = 0;
    for (int a = 0; a < nv; ++a) {
      const double ua = (a < c) ? g.UtWt[(long)a * g.n + i] : g.Uty[i];
      for (int b = a; b < nv; ++b) {
        const double ub = (b < c) ? g.UtWt[(long)b * g.n + i] : g.Uty[i];
        // static bound: p < 15 since nv <= 5 on this path
        if (p < 15) s[p] += ua * ub * w;
        ++p;
      }
    }
  }
  __shared__ double red[4][GRID_FIX_LD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int p = 0; p < GRID_FIX_LD; ++p) {
    const double v = wave_sum(s[p]);
    if (lane == 0) red[wave][p] = v;
  }
  __syncthreads();
  if (threadIdx.x < GRID_FIX_LD)
    F[(long)q * GRID_FIX_LD + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// T[s][cb * 16 + col] = sum_k A(s, k) * R(k, col),  A = x^2 for the first NBX column blocks, x for the rest.
// One block = 16 SNP rows; its 4 waves split K; v_mfma_f64_16x16x4_f64 with the A operand read straight from
// the UtX rows (lane (i, kq) takes x[s0 + i][16 chunk + 4 kq .. + 3]: a full 128-byte line per row per chunk;
// the k order inside a chunk is permuted identically on both operands).
// GATHER (the Chebyshev tables of the bracket intervals): blockIdx.y = interval k; the rows are the SNPs listed in
// list[k * cap ..] (count[k] of them, written by cheb_scan_kernel), the weights Rp + k * rp_stride, and the series of
// `slot` is column `slot` of the interval's plane: T[(k * NB * 16 + col) * cap + slot].  A SNP's sums depend on its own row only (fixed k order), so its series
// is the same bits whichever slot and whichever neighbours it gets.
struct TableGather {
  const int *list;
  const int *count;
  long cap;
  long rp_stride;
};
template <int NBX, int NBA, bool GATHER>
__global__ __launch_bounds__(256) void grid_table_kernel(const double *__restrict__ UtX, long ld, long l, int n,
                                                        int nc, const double *__restrict__ Rp,
                                                        double *__restrict__ T, TableGather tg) {
  constexpr int NB = NBX + NBA;
  __shared__ double red[3][NB][256];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int i = lane & 15, kq = lane >> 4;
  const long s0 = (long)blockIdx.x * 16;
  long row = s0 + i;
  if (GATHER) {
    const int kint = blockIdx.y;
    l = tg.count[kint];
    if (s0 >= l) return;
    if (row >= l) row = l - 1;
    row = tg.list[(long)kint * tg.cap + row];
    Rp += (long)kint * tg.rp_stride;
    T += (long)kint * tg.cap * (NB * 16); // this interval's [col][slot] plane
  } else {
    if (row >= l) row = l - 1;
  }
  const double *xr = UtX + row * ld + 4 * kq;
  const int c0 = (int)((long)nc * wave / 4), c1 = (int)((long)nc * (wave + 1) / 4);
  f64x4 acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) acc[b] = f64x4{0.0, 0.0, 0.0, 0.0};
  const double *rp = Rp + ((long)c0 * NB * 64 + lane) * 4;
  for (int ch = c0; ch < c1; ++ch) {
    const long k = (long)ch * 16 + 4 * kq;
    double xv[4];
    if (k + 3 < n) {
      const f64x2 lo = *reinterpret_cast<const f64x2 *>(xr + (long)ch * 16);
      const f64x2 hi = *reinterpret_cast<const f64x2 *>(xr + (long)ch * 16 + 2);
      xv[0] = lo.x; xv[1] = lo.y; xv[2] = hi.x; xv[3] = hi.y;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) xv[j] = (k + j < n) ? xr[(long)ch * 16 + j] : 0.0;
    }
    double rb[NB][4];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const f64x2 lo = *reinterpret_cast<const f64x2 *>(rp + (long)b * 256);
      const f64x2 hi = *reinterpret_cast<const f64x2 *>(rp + (long)b * 256 + 2);
      rb[b][0] = lo.x; rb[b][1] = lo.y; rb[b][2] = hi.x; rb[b][3] = hi.y;
    }
    rp += (long)NB * 256;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double xs = xv[j] * xv[j];
#pragma unroll
      for (int b = 0; b < NB; ++b)
        acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(b < NBX ? xs : xv[j], rb[b][j], acc[b], 0, 0, 0);
    }
  }
  // combine the four K slices in a fixed order (wave 0 + 1 + 2 + 3); accumulator r of lane: row kq + 4r, col i
  if (wave > 0) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave - 1][b][r * 64 + lane] = acc[b][r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double v = ((acc[b][r] + red[0][b][r * 64 + lane]) + red[1][b][r * 64 + lane]) + red[2][b][r * 64 + lane];
        const long srow = s0 + kq + 4 * r;
        if (srow < l) {
          if (GATHER)
            T[(long)(b * 16 + i) * tg.cap + srow] = v; // column-major over the slots: cheb_search_kernel reads coalesced
          else
            T[srow * (NB * 16) + b * 16 + i] = v;
        }
      }
  }
}

// ---- the same table product, tiled for reuse (default): one WAVE owns RG * 16 SNP rows and one slice of K, so that a
// 16-k chunk of the weight matrix (NB * 2 KiB) feeds RG * NB * 4 MFMAs instead of NB * 4; a block's four waves take four
// different row sets over the SAME K slice (their weight reads coincide in the L1), blockIdx.y picks the K slice.  The K
// slices leave partial sums P[ks][row][col]; table_reduce_kernel adds them in slice order (fixed order, no atomics: a
// SNP's sums do not depend on its neighbours, its slot or the batch size -- the number of slices depends on n alone).
struct TableV2 {
  const double *UtX;
  long ld;
  long l;        // rows (dense) -- GATHER: unused, count[kint] rules
  int n, nc;     // individuals, 16-k chunks
  int ksplit;    // K slices
  const double *Rp;
  double *P;     // partial sums: [(kint * ksplit + ks) * cap + row] * NB16 + col
  long cap;      // rows allocated per interval / batch
  TableGather tg;
};
template <int NBX, int NBA, int RG, bool GATHER, bool PF>
__global__ __launch_bounds__(256, PF ? 1 : 2) void table_v2_kernel(TableV2 a) {
  constexpr int NB = NBX + NBA;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, kq = lane >> 4;
  const int ks = blockIdx.y;
  const int kint = GATHER ? blockIdx.z : 0;
  long l = a.l;
  const double *Rp = a.Rp;
  if (GATHER) {
    l = a.tg.count[kint];
    Rp += (long)kint * a.tg.rp_stride;
  }
  const long s0 = ((long)blockIdx.x * 4 + wave) * (RG * 16);
  if (s0 >= l) return;
  const double *xr[RG];
#pragma unroll
  for (int g = 0; g < RG; ++g) {
    long row = s0 + g * 16 + i;
    if (row >= l) row = l - 1;
    if (GATHER) row = a.tg.list[(long)kint * a.tg.cap + row];
    xr[g] = a.UtX + row * a.ld + 4 * kq;
  }
  const int c0 = (int)((long)a.nc * ks / a.ksplit), c1 = (int)((long)a.nc * (ks + 1) / a.ksplit);
  f64x4 acc[RG][NB];
#pragma unroll
  for (int g = 0; g < RG; ++g)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[g][b] = f64x4{0.0, 0.0, 0.0, 0.0};
  const double *rp0 = Rp + (long)lane * 4;
  const int n = a.n;
  // The main loop runs over the chunks that lie entirely inside n and is branch-free: one block of vector loads, then
  // RG * NB * 4 MFMAs (a per-lane "is this k inside n" test inside the loop made the compiler serialise the row loads of
  // the row groups with a full s_waitcnt between them -- two memory latencies per chunk).  The one partial chunk of a
  // ragged n is done after it with masked loads.  PF: operands of chunk ch + 1 are requested before the MFMAs of chunk ch.
  const int nfull = n / 16;
  const int cfull = c1 < nfull ? c1 : nfull;
#define TV2_LOAD(RB, XV, CH)                                                                                      \
  do {                                                                                                            \
    const double *rp_ = rp0 + (long)(CH) * NB * 256;                                                              \
    _Pragma("unroll") for (int b = 0; b < NB; ++b) {                                                              \
      const f64x2 lo = *reinterpret_cast<const f64x2 *>(rp_ + (long)b * 256);                                     \
      const f64x2 hi = *reinterpret_cast<const f64x2 *>(rp_ + (long)b * 256 + 2);                                 \
      RB[b][0] = lo.x; RB[b][1] = lo.y; RB[b][2] = hi.x; RB[b][3] = hi.y;                                         \
    }                                                                                                             \
    _Pragma("unroll") for (int g = 0; g < RG; ++g) {                                                              \
      const f64x2 lo = *reinterpret_cast<const f64x2 *>(xr[g] + (long)(CH) * 16);                                 \
      const f64x2 hi = *reinterpret_cast<const f64x2 *>(xr[g] + (long)(CH) * 16 + 2);                             \
      XV[g][0] = lo.x; XV[g][1] = lo.y; XV[g][2] = hi.x; XV[g][3] = hi.y;                                         \
    }                                                                                                             \
  } while (0)
#define TV2_MFMA(RB, XV)                                                                                          \
  do {                                                                                                            \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                               \
      _Pragma("unroll") for (int g = 0; g < RG; ++g) {                                                            \
        const double xs = XV[g][j] * XV[g][j];                                                                    \
        _Pragma("unroll") for (int b = 0; b < NB; ++b)                                                            \
          acc[g][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(b < NBX ? xs : XV[g][j], RB[b][j], acc[g][b], 0, 0, 0); \
      }                                                                                                           \
    }                                                                                                             \
  } while (0)
  double rbA[NB][4], xvA[RG][4];
  if (PF) {
    double rbB[NB][4], xvB[RG][4];
    if (c0 < cfull) TV2_LOAD(rbA, xvA, c0);
    int ch = c0;
    for (; ch + 1 < cfull; ch += 2) {
      TV2_LOAD(rbB, xvB, ch + 1);
      __builtin_amdgcn_sched_barrier(0);
      TV2_MFMA(rbA, xvA);
      __builtin_amdgcn_sched_barrier(0);
      // unconditional (the last round re-requests chunk ch + 1): a conditional load makes the compiler wait for every
      // outstanding load at the merge point, which would undo the overlap for set B
      TV2_LOAD(rbA, xvA, (ch + 2 < cfull ? ch + 2 : ch + 1));
      __builtin_amdgcn_sched_barrier(0);
      TV2_MFMA(rbB, xvB);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ch < cfull) TV2_MFMA(rbA, xvA); // odd count: the last chunk's operands are already in set A
  } else {
    for (int ch = c0; ch < cfull; ++ch) {
      TV2_LOAD(rbA, xvA, ch);
      TV2_MFMA(rbA, xvA);
    }
  }
  if (c1 > nfull && c0 <= nfull) { // the partial chunk nfull of a ragged n belongs to this slice
    const int ch = nfull;
    const long k = (long)ch * 16 + 4 * kq;
    const double *rp_ = rp0 + (long)ch * NB * 256;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const f64x2 lo = *reinterpret_cast<const f64x2 *>(rp_ + (long)b * 256);
      const f64x2 hi = *reinterpret_cast<const f64x2 *>(rp_ + (long)b * 256 + 2);
      rbA[b][0] = lo.x; rbA[b][1] = lo.y; rbA[b][2] = hi.x; rbA[b][3] = hi.y;
    }
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) xvA[g][j] = (k + j < n) ? xr[g][(long)ch * 16 + j] : 0.0;
    TV2_MFMA(rbA, xvA);
  }
#undef TV2_LOAD
#undef TV2_MFMA
  // accumulator r of lane: row kq + 4r of the row group, column i of the block
  double *P = a.P + ((long)(kint * a.ksplit + ks) * a.cap) * (NB * 16);
#pragma unroll
  for (int g = 0; g < RG; ++g)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long srow = s0 + g * 16 + kq + 4 * r;
        if (srow < l) P[srow * (NB * 16) + b * 16 + i] = acc[g][b][r];
      }
}
// T = sum over the K slices, in slice order.  Dense: T[row][col]; GATHER: T[(kint * NB16 + col) * cap + slot]
template <bool GATHER>
__global__ __launch_bounds__(256) void table_reduce_kernel(const double *__restrict__ P, int ksplit, long cap, int nb16,
                                                          long l, const int *__restrict__ count, double *__restrict__ T) {
  const int kint = GATHER ? blockIdx.y : 0;
  if (GATHER) l = count[kint];
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= l * nb16) return;
  const long row = e / nb16;
  const int col = (int)(e - row * nb16);
  const double *p = P + ((long)kint * ksplit * cap + row) * nb16 + col;
  double v = 0.0;
  for (int ks = 0; ks < ksplit; ++ks) v += p[(long)ks * cap * nb16];
  if (GATHER)
    T[((long)kint * nb16 + col) * cap + row] = v;
  else
    T[row * nb16 + col] = v;
}

// ------------------------------------------------------------------ Chebyshev tables of the bracket intervals
// (lmm_search.hip.h).  Per lmm_setup and interval: c_k(delta_i), the series coefficients of t -> H_i(t) and of
// t -> 1 - H_i(t) (cheb_coeff_kernel), from them the weight matrix of the table product in MFMA operand order
// (cheb_weights_kernel; columns: [k] for x^2, [a * CHEB_N + k] for x u_a) and the SNP-independent series
// (cheb_fixed_kernel).  Per batch: cheb_scan_kernel finds every SNP's bracket intervals from the fixed-lambda table and
// hands out slots, grid_table_kernel<.., GATHER> computes the series of exactly those (SNP, interval) pairs.
constexpr double CHEB_MARGIN = 0.15; // of the interval's length, either side
constexpr double CHEB_MIN_LAMBDA = 1e-3; // intervals below are tabulated in Q form (lmm_search.hip.h): dS/dt is O(lambda) of S
                                         // there, and a series good to 1e-15 of S would carry 1e-13/lambda of relative error in it

struct ChebNodes {
  double lam[CHEB_N]; // exp(node m)
};

// Ck: series of H_i -- or, for an interval in Q form (lmm_search.hip.h: below lambda = 1e-3), of delta_i H_i --, Gk: of 1 - H_i,
// Lk: of log(lambda delta_i + 1), G2k: of (1 - H_i)^2  (each n x CHEB_N)
__global__ void cheb_coeff_kernel(const double *__restrict__ eval, int n, ChebNodes nd, const double *__restrict__ Dfit,
                                  int qform, double *__restrict__ Ck, double *__restrict__ Gk, double *__restrict__ Lk,
                                  double *__restrict__ G2k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double d = eval[i];
  double hm[CHEB_N], gm[CHEB_N], lm[CHEB_N];
#pragma unroll
  for (int m = 0; m < CHEB_N; ++m) {
    const double ld = nd.lam[m] * d;
    hm[m] = (qform ? d : 1.0) / (ld + 1.0);
    gm[m] = ld / (ld + 1.0);
    lm[m] = log(fabs(ld + 1.0)); // as the row passes: log|lambda delta + 1|
  }
  for (int k = 0; k < CHEB_N; ++k) {
    double sh = 0.0, sg = 0.0, sl = 0.0, s2 = 0.0;
#pragma unroll
    for (int m = 0; m < CHEB_N; ++m) {
      const double w = Dfit[k * CHEB_N + m];
      sh += hm[m] * w;
      sg += gm[m] * w;
      sl += lm[m] * w;
      s2 += gm[m] * gm[m] * w;
    }
    Ck[(long)i * CHEB_N + k] = sh;
    Gk[(long)i * CHEB_N + k] = sg;
    Lk[(long)i * CHEB_N + k] = sl;
    G2k[(long)i * CHEB_N + k] = s2;
  }
}

// weight matrix of one interval in the B-operand order of grid_table_kernel (see grid_weights_kernel)
__global__ void cheb_weights_kernel(AssocArgs g, GridGeom gg, int c, const double *__restrict__ Ck,
                                    double *__restrict__ Rp) {
  const long total = (long)gg.nc * (gg.nbx + gg.nba) * 256;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int j = (int)(e & 3), col = (int)((e >> 2) & 15), kq = (int)((e >> 6) & 3);
  const long r = e >> 8;
  const int nb = gg.nbx + gg.nba;
  const int cb = (int)(r % nb);
  const long chunk = r / nb;
  const long k = chunk * 16 + 4 * kq + j;
  double v = 0.0;
  if (k < g.n) {
    if (cb < gg.nbx) {
      const int q = cb * 16 + col;
      if (q < CHEB_N) v = Ck[k * CHEB_N + q];
    } else {
      const int idx = (cb - gg.nbx) * 16 + col;
      const int a = idx / CHEB_N, q = idx - a * CHEB_N;
      if (a <= c) {
        const double u = (a < c) ? g.UtWt[(long)a * g.n + k] : g.Uty[k];
        v = u * Ck[k * CHEB_N + q];
      }
    }
  }
  Rp[e] = v;
}

// SNP-independent series of one interval, one block per function: block b < npairs: the pair (a <= bb) among
// (w_1..w_c, y) in row-major upper-triangle order, a_k = sum_i u_a u_bb c_k(delta_i); block npairs: g, a_k = sum_i Gk[i][k];
// block npairs + 1: log|H| = sum_i log(lambda delta_i + 1), a_k = sum_i Lk[i][k]; block npairs + 2: sum_i (1 - H_i)^2
__global__ __launch_bounds__(256) void cheb_fixed_kernel(AssocArgs g, int c, const double *__restrict__ Ck,
                                                        const double *__restrict__ Gk, const double *__restrict__ Lk,
                                                        const double *__restrict__ G2k, double *__restrict__ F) {
  const int nv = c + 1, npairs = nv * (nv + 1) / 2;
  const int b = blockIdx.x;
  int pa = 0, pb = 0;
  if (b < npairs) {
    int p = 0;
    for (int a = 0; a < nv; ++a)
      for (int bb = a; bb < nv; ++bb, ++p)
        if (p == b) { pa = a; pb = bb; }
  }
  double s[CHEB_N];
#pragma unroll
  for (int k = 0; k < CHEB_N; ++k) s[k] = 0.0;
  for (long i = threadIdx.x; i < g.n; i += 256) {
    double w = 1.0;
    const double *src = (b == npairs) ? Gk : (b == npairs + 1) ? Lk : G2k;
    if (b < npairs) {
      const double ua = (pa < c) ? g.UtWt[(long)pa * g.n + i] : g.Uty[i];
      const double ub = (pb < c) ? g.UtWt[(long)pb * g.n + i] : g.Uty[i];
      w = ua * ub;
      src = Ck;
    }
#pragma unroll
    for (int k = 0; k < CHEB_N; ++k) s[k] += w * src[i * CHEB_N + k];
  }
  __shared__ double red[4][CHEB_N];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < CHEB_N; ++k) {
    const double v = wave_sum(s[k]);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < CHEB_N)
    F[(long)b * CHEB_N + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// Which grid intervals bracket a sign change of dev1 for this SNP (REML and / or ML search, as a_mode asks): the
// same dev1_grid values calc_lambda computes from the fixed-lambda table.  Every (SNP, tabulated interval) pair with a
// bracket gets a slot of that interval's table (slots[snp * nint + k], -1 = none) and the bracket's end values
// dends[(func * nint + k) * cap + slot] = {dev1(lambda_lo), dev1(lambda_hi)} (NaN = func has no bracket there).
struct ChebScanArgs {
  int *count;     // nint, zeroed before the launch
  int *list;      // nint x cap
  int *slots;     // l x nint
  double2 *dends; // 2 x nint x cap
  long cap;
};
template <int C, bool REML>
__device__ __forceinline__ void cheb_scan_func(const AssocArgs &g, const SnpCtx<FixedC<C>> &cx, const ChebScanArgs &sc,
                                               long snp, int lane) {
  const int func = REML ? 0 : 1;
  double d_lo = dev1_grid<FixedC<C>, REML>(cx, 0);
  for (int i = 0; i < g.n_region; ++i) {
    const double d_hi = dev1_grid<FixedC<C>, REML>(cx, i + 1);
    const int k = i - g.cheb_j0;
    if (d_lo * d_hi <= 0 && k >= 0 && k < g.cheb_nint && lane == 0) {
      int slot = sc.slots[snp * g.cheb_nint + k]; // this thread's own earlier write, if any
      if (slot < 0) {
        slot = atomicAdd(&sc.count[k], 1);
        sc.list[(long)k * sc.cap + slot] = (int)snp;
        sc.slots[snp * g.cheb_nint + k] = slot;
        sc.dends[((long)(1 - func) * g.cheb_nint + k) * sc.cap + slot] = make_double2(NAN, NAN);
      }
      sc.dends[((long)func * g.cheb_nint + k) * sc.cap + slot] = make_double2(d_lo, d_hi);
    }
    d_lo = d_hi;
  }
}
template <int C>
__global__ __launch_bounds__(256) void cheb_scan_kernel(AssocArgs g, ChebScanArgs sc) {
  const int lane = threadIdx.x & 63;
  const long snp = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (snp >= g.l) return;
  SnpCtx<FixedC<C>> cx;
  cx.g = &g;
  cx.x = g.UtX + snp * g.ld;
  cx.y = g.Uty;
  cx.lane = lane;
  cx.m = FixedC<C>();
  cx.logdet_iw = 0.0;
  cx.trow = g.grid_T + snp * g.grid_ld;
  cx.cslots = nullptr;
  if (lane == 0)
    for (int k = 0; k < g.cheb_nint; ++k) sc.slots[snp * g.cheb_nint + k] = -1;
  if (g.a_mode == 1 || g.a_mode == 4) cheb_scan_func<C, true>(g, cx, sc, snp, lane);
  if (g.a_mode == 2 || g.a_mode == 4 || g.a_mode == 9) cheb_scan_func<C, false>(g, cx, sc, snp, lane);
}

// One THREAD per (slot, interval, func): Brent + Newton of that bracket on the SNP's series (polish_bracket over
// ChebEvaluator, lmm_search.hip.h -- the code tests/cpp/cheb_search_check.cpp runs on the CPU).  The series are read
// column-major over the slots (coalesced across the wave), the SNP-independent series are the same addresses for every
// thread.  grid = (ceil(cap / 64), nint, 2).
struct ChebSearchArgs {
  const int *count;
  const int *list;            // [k * cap + slot] = SNP index (the constants of a Q-form interval come from its fixed-lambda row)
  unsigned long long qmask;   // bit k: interval k is in Q form
  const double2 *dends;
  ChebResult *res;
  double mid[ASSOC_MAX_REGION], inv_half[ASSOC_MAX_REGION];
};
template <int C, bool REML>
__device__ __forceinline__ void cheb_search_one(const AssocArgs &g, const ChebSearchArgs &sa, int k, long slot) {
  const int func = REML ? 0 : 1;
  const long e = ((long)func * g.cheb_nint + k) * g.cheb_cap + slot;
  const double2 d = sa.dends[e];
  ChebResult r;
  r.l = 0.0;
  r.status = CHEB_NONE;
  r.pad = 0;
  if (!(d.x != d.x)) { // this function has a bracket in the interval
    ChebEvaluator<C, REML> ev;
    ev.cs.snp = g.cheb_T + (long)k * g.cheb_cap * g.cheb_ld + slot;
    ev.cs.sstride = g.cheb_cap;
    ev.cs.fix = g.cheb_F + (long)k * g.cheb_fld;
    ev.cs.xa0 = g.cheb_xa0;
    ev.cs.mid = sa.mid[k];
    ev.cs.inv_half = sa.inv_half[k];
    ev.cs.n = (double)g.n;
    ev.cs.qform = (int)((sa.qmask >> k) & 1ull);
    ev.cs.s0f = g.grid_F; // weight q = 0 is 1: F[0][pair] = sum_i u_a u_b
#pragma unroll
    for (int a = 0; a < C + 2; ++a) ev.cs.s0x[a] = 0.0;
    if (ev.cs.qform) {
      const double *trow = g.grid_T + (long)sa.list[(long)k * g.cheb_cap + slot] * g.grid_ld;
      ev.cs.s0x[0] = trow[0];
#pragma unroll
      for (int a = 0; a <= C; ++a) ev.cs.s0x[1 + a] = trow[g.grid_xa0 + a * g.grid_nq];
    }
    double l = 0.0, l_temp = 0.0;
    const int j = g.cheb_j0 + k;
    r.status = polish_bracket(ev, g.lam_grid[j], g.lam_grid[j + 1], d.x, d.y, g.l_min, g.l_max, l, l_temp);
    r.l = l;
  }
  sa.res[e] = r;
}
template <int C>
__global__ __launch_bounds__(64) void cheb_search_kernel(AssocArgs g, ChebSearchArgs sa) {
  const int k = blockIdx.y;
  const long slot = (long)blockIdx.x * 64 + threadIdx.x;
  if (slot >= sa.count[k]) return;
  if (blockIdx.z == 0) {
    if (g.a_mode == 1 || g.a_mode == 4) cheb_search_one<C, true>(g, sa, k, slot);
  } else {
    if (g.a_mode == 2 || g.a_mode == 4 || g.a_mode == 9) cheb_search_one<C, false>(g, sa, k, slot);
  }
}

} // namespace gemma_hip
