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
template <int NBX, int NBA>
__global__ __launch_bounds__(256) void grid_table_kernel(const double *__restrict__ UtX, long ld, long l, int n,
                                                        int nc, const double *__restrict__ Rp,
                                                        double *__restrict__ T) {
  constexpr int NB = NBX + NBA;
  __shared__ double red[3][NB][256];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int i = lane & 15, kq = lane >> 4;
  const long s0 = (long)blockIdx.x * 16;
  long row = s0 + i;
  if (row >= l) row = l - 1;
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
        if (srow < l) T[srow * (NB * 16) + b * 16 + i] = v;
      }
  }
}

} // namespace gemma_hip
