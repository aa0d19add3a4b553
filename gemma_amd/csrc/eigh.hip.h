// Symmetric eigensolver for gfx950: all eigenpairs of a dense symmetric fp64 matrix.
//
// Replaces dsyevr_ behind lapack_eigen_symmv / EigenDecomp (GEMMA src/lapack.cpp:149-254).  Parity does
// not require LAPACK's vectors (every statistic is invariant under sign flips / rotations inside an
// eigenspace), only accurate eigenvalues and an orthogonal U.
//
//  1. Blocked Householder tridiagonalisation A = Q T Q^T (panel width 128).  Per column: one
//     single-workgroup kernel applies the pending panel updates to the column and builds the
//     reflector, a full-width SYMV streams the trailing matrix from HBM (the HBM-bound half of the
//     flops: sum_j 8 m^2 B = (8/3) n^3 B), a third kernel finishes w.  Per panel: the rank-2*128
//     trailing update A -= V W^T + W V^T runs on the fp64 MFMA GEMM.  Both triangles of A are kept,
//     so "column j" is read as the contiguous row j and nothing is ever accessed with stride n.
//  2. Divide and conquer on T (Cuppen; Gu-Eisenstat stabilisation): leaves <= 64 by implicit QL
//     (one wavefront per leaf, eigenvector rows in LDS); every merge = rank-one update
//     D + rho z z^T: deflation (O(k) scalar logic, on the host), secular equation (one wavefront
//     per root: bisection in pole-shifted coordinates to the last ulp), Loewner re-computation of z,
//     eigenvector matrix, and the k x k x n_sub fp64 MFMA GEMM that carries the flops.
//     Eigenvectors are stored as ROWS (Q^T) so Givens deflation, gathers and GEMM operands are
//     contiguous; eigenvalues are kept in physical row order with a logical permutation.
//  3. Back-transformation U^T = Z^T H_{n-3} ... H_0 by panels in compact WY form
//     (Z^T -= ((Z^T Y) T^T) Y^T: three MFMA GEMMs per panel), then one permuting transpose into
//     the row-major U with eigenvector k in column k, ascending eigenvalues.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <string>
#include <vector>

#include "dgemm_mfma.hip.h"
#include "eigh_tu.h" // EighShard

namespace gemma_hip {

constexpr int EIG_NB = 128;
constexpr int EIG_LEAF = 64;

__device__ __forceinline__ double eig_wsum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double eig_wprod(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v *= __shfl_xor(v, off, 64);
  return v;
}
// block-wide sum for 1024-thread blocks; result valid in every thread
__device__ __forceinline__ double eig_bsum1024(double v, double *red /* >= 17 doubles of LDS */) {
  v = eig_wsum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (unsigned w = 0; w < blockDim.x / 64; ++w) t += red[w];
    red[16] = t;
  }
  __syncthreads();
  return red[16];
}

// ---------------------------------------------------------------- 1. tridiagonalisation
// Column j of the current panel (panel starts at j0, k = j - j0 earlier columns in it), four launches,
// every one of them spread over the whole chip:
//   td_col   : x = A[j][j:] - sum_q ( V_q[j:] W_q[j] + W_q[j:] V_q[j] )           -> xcol, partial |x|^2
//   td_symv  : reflector scalars (LAPACK dlarfg: beta, tau, scale) from the partials, u_j -> row j of VT,
//              p = A[j+1:, j+1:] u  (one wavefront per row, 16-byte loads), W_q.u and V_q.u
//   td_w1    : w' = tau (p - V (W^T u) - W (V^T u))                                 -> wtmp, partial w'.u
//   td_w2    : w = w' - (tau/2)(w'.u) u                                              -> row k of WT
constexpr int TD_CHUNK = 256;

// deterministic sum of `cnt` partials, result in every thread of the block (first wavefront reduces)
__device__ __forceinline__ double td_sum_parts(const double *__restrict__ parts, int cnt, double *sh) {
  if (threadIdx.x < 64) {
    double s = 0.0;
    for (int i = threadIdx.x; i < cnt; i += 64) s += parts[i];
    s = eig_wsum(s);
    if (threadIdx.x == 0) *sh = s;
  }
  __syncthreads();
  return *sh;
}

struct TdScalars {
  double scale, tau, beta;
  bool has_e;
};
// LAPACK dlarfg on x[1:] (alpha = x[1], xnorm^2 given)
__device__ __forceinline__ TdScalars td_reflector(const double *__restrict__ xcol, long n, long j, double xnorm2) {
  TdScalars r;
  r.scale = 0.0;
  r.tau = 0.0;
  r.beta = 0.0;
  r.has_e = (j + 1 < n);
  if (r.has_e) {
    const double alpha = xcol[j + 1];
    if (xnorm2 == 0.0) {
      r.beta = alpha;
    } else {
      double beta = sqrt(alpha * alpha + xnorm2);
      if (alpha > 0.0) beta = -beta;
      r.tau = (beta - alpha) / beta;
      r.scale = 1.0 / (alpha - beta);
      r.beta = beta;
    }
  }
  return r;
}

// td_col and td_w1 are skinny panel products (m rows x 2k panel columns): 64 rows per block, the panel index q dealt
// over the block's four wavefronts (q = wave, wave + 4, ...) and combined through LDS in a fixed order -- four times the
// blocks and a quarter of the dependent-load chain of a thread-per-row loop.
constexpr int TD_ROWS = 64;
__global__ __launch_bounds__(256) void td_col_kernel(const double *__restrict__ A, long n, long j, long j0,
                                                     const double *__restrict__ VT, const double *__restrict__ WT,
                                                     double *__restrict__ xcol, double *__restrict__ ssbuf) {
  __shared__ double sv[EIG_NB], sw[EIG_NB];
  __shared__ double part[4][TD_ROWS];
  const int t = threadIdx.x, rl = t & 63, qg = t >> 6;
  const int k = (int)(j - j0);
  for (int q = t; q < k; q += 256) {
    sv[q] = VT[(j0 + q) * n + j];
    sw[q] = WT[(long)q * n + j];
  }
  __syncthreads();
  const long r = j + (long)blockIdx.x * TD_ROWS + rl;
  double acc = 0.0;
  if (r < n)
    for (int q = qg; q < k; q += 4) acc += VT[(j0 + q) * n + r] * sw[q] + WT[(long)q * n + r] * sv[q];
  part[qg][rl] = acc;
  __syncthreads();
  if (qg == 0) {
    double ss = 0.0;
    if (r < n) {
      const double x = A[j * n + r] - ((part[0][rl] + part[1][rl]) + (part[2][rl] + part[3][rl]));
      xcol[r] = x;
      if (r >= j + 2) ss = x * x;
    }
    ss = eig_wsum(ss);
    if (rl == 0) ssbuf[blockIdx.x] = ss;
  }
}

// The per-column work that only touches the panel: idx in [0, 2k) is one dot of the new reflector u_j with W_q (even)
// or u_q (odd), q = idx >> 1; idx >= 2k are the writers of VT row j (256 entries each; the first also stores d, e, tau).
struct TdPanel {
  long n, j, j0;
  double *VT;
  const double *WT;
  const double *xcol;
  double *ab;
  int k;
  double *d, *e, *tau;
  double *Spanel; // strict upper triangle of the panel's Y Y^T (ld EIG_NB) for the back-transformation, or nullptr
};

__device__ __forceinline__ void td_panel_block(const TdPanel &g, int idx, const TdScalars &sc, double *red) {
  const long n = g.n, j = g.j, j1 = g.j + 1;
  const double scale = sc.scale;
  const int t = threadIdx.x;
  if (idx < 2 * g.k) {
    const int q = idx >> 1;
    const double *__restrict__ vecp = (idx & 1) ? (g.VT + (g.j0 + q) * n) : (g.WT + (long)q * n);
    const double *__restrict__ xcol = g.xcol;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    long c = j1 + t;
    for (; c + 768 < n; c += 1024) { // four independent loads per operand in flight
      const double v0 = vecp[c], v1 = vecp[c + 256], v2 = vecp[c + 512], v3 = vecp[c + 768];
      const double x0 = xcol[c], x1 = xcol[c + 256], x2 = xcol[c + 512], x3 = xcol[c + 768];
      s0 += v0 * ((c == j1) ? 1.0 : x0 * scale);
      s1 += v1 * (x1 * scale);
      s2 += v2 * (x2 * scale);
      s3 += v3 * (x3 * scale);
    }
    for (; c < n; c += 256) s0 += vecp[c] * ((c == j1) ? 1.0 : xcol[c] * scale);
    const double s = eig_wsum((s0 + s1) + (s2 + s3));
    if ((t & 63) == 0) red[t >> 6] = s;
    __syncthreads();
    if (t == 0) {
      const double dot = (red[0] + red[1]) + (red[2] + red[3]);
      g.ab[idx] = dot;
      if ((idx & 1) && g.Spanel) g.Spanel[q * EIG_NB + g.k] = dot; // u_q . u_j
    }
  } else {
    const int w = idx - 2 * g.k;
    if (w == 0 && t == 0) {
      g.d[j] = g.xcol[j];
      if (sc.has_e) g.e[j] = sc.beta;
      g.tau[j] = sc.tau;
    }
    const long r = (long)w * 256 + t;
    if (r < n) {
      double v = 0.0;
      if (r == j1)
        v = 1.0;
      else if (r > j1)
        v = g.xcol[r] * scale;
      g.VT[j * n + r] = v;
    }
  }
}

// grid: [0, nsymv) SYMV rows (4 per block) | then the 2k + ceil(n/256) panel blocks (td_panel_block)
__global__ __launch_bounds__(256) void td_symv_kernel(const double *__restrict__ A, TdPanel pg,
                                                      const double *__restrict__ ssbuf, int nparts,
                                                      double *__restrict__ p, int nsymv) {
  const long n = pg.n, j = pg.j;
  const double *__restrict__ xcol = pg.xcol;
  __shared__ double sh;
  __shared__ double red[4];
  const double xnorm2 = td_sum_parts(ssbuf, nparts, &sh);
  const TdScalars sc = td_reflector(xcol, n, j, xnorm2);
  const double scale = sc.scale;
  const int lane = threadIdx.x & 63;
  const long j1 = j + 1;
  const int b = (int)blockIdx.x;
  if (b < nsymv) {
    const long r = j1 + (long)b * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const double *__restrict__ row = A + r * n;
    double s0 = 0.0, s1 = 0.0;
    long c = j1;
    const bool vec = ((n & 1) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(xcol) & 15) == 0);
    if (vec) {
      if (c & 1) { // leading odd element is u[j+1] == 1
        if (lane == 0) s0 += row[c];
        c++;
      }
      long cc = c + 2 * lane;
      for (; cc + 128 + 1 < n; cc += 256) {
        const f64x2 a0 = *reinterpret_cast<const f64x2 *>(row + cc);
        const f64x2 a1 = *reinterpret_cast<const f64x2 *>(row + cc + 128);
        const f64x2 x0 = *reinterpret_cast<const f64x2 *>(xcol + cc);
        const f64x2 x1 = *reinterpret_cast<const f64x2 *>(xcol + cc + 128);
        const double u00 = (cc == j1) ? 1.0 : x0.x * scale, u01 = x0.y * scale;
        s0 += a0.x * u00 + a0.y * u01;
        s1 += a1.x * (x1.x * scale) + a1.y * (x1.y * scale);
      }
      for (; cc + 1 < n; cc += 128) {
        const f64x2 a0 = *reinterpret_cast<const f64x2 *>(row + cc);
        const f64x2 x0 = *reinterpret_cast<const f64x2 *>(xcol + cc);
        const double u00 = (cc == j1) ? 1.0 : x0.x * scale, u01 = x0.y * scale;
        s0 += a0.x * u00 + a0.y * u01;
      }
    } else {
      for (long cc = c + lane; cc < n; cc += 64) {
        const double u = (cc == j1) ? 1.0 : xcol[cc] * scale;
        s0 += row[cc] * u;
      }
    }
    const double s = eig_wsum(s0 + s1);
    if (lane == 0) p[r] = s;
  } else {
    td_panel_block(pg, b - nsymv, sc, red);
  }
}

// Symmetric form of p = A[j+1:, j+1:] u: only the LOWER triangle is read (half the HBM traffic of the row-per-wave
// form above, which is what bounds the tridiagonalisation).  A block owns a strip of 64 rows (blockIdx.y) and one segment
// of 1024 columns (blockIdx.x) of the lower triangle, in sub-tiles of 64 x 64: thread (rg = t >> 5, cl = t & 31) loads
// A[64 I + 8 rg + k][c0 + 2 cl .. + 1], k = 0..7, as 16-byte pieces and feeds BOTH products of the element:
//   row part  rowacc[k]  += A[r][c] u[c]          (c <= r)   -> rowP[seg][r]   after a 32-lane butterfly at the end
//   col part  colacc[..] += A[r][c] u[r]          (c <  r)   -> colP[I][c]     after an 8-way LDS sum per sub-tile
// td_symv_reduce_kernel then forms p[r] = sum_seg rowP[seg][r] + sum_{I' >= strip(r)} colP[I'][r] in a fixed order (no
// atomics: the result does not depend on scheduling).  Needs n even (16-byte row alignment).
constexpr int TS_STRIP = 64, TS_SEG_MIN = 512, TS_WIDE = 256; // segment width: runtime, a multiple of TS_WIDE
__global__ __launch_bounds__(256) void td_symv_sym_kernel(const double *__restrict__ A, long n, long j,
                                                          const double *__restrict__ xcol,
                                                          const double *__restrict__ ssbuf, int nparts,
                                                          double *__restrict__ rowP, double *__restrict__ colP,
                                                          TdPanel pg, int ny_panel, int n_panel, int TS_SEG) {
  __shared__ double sh;
  __shared__ double cbuf[2][8][TS_WIDE];
  const long j1 = j + 1;
  if ((int)blockIdx.y < ny_panel) {
    // the panel dots and the VT row writers ride in the first grid rows: latency-bound reads of 2k panel rows that
    // overlap the HBM-bound strips instead of costing a launch of their own (~55 us per column at n = 20000)
    const int idx = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
    if (idx >= n_panel) return;
    const double xn2 = td_sum_parts(ssbuf, nparts, &sh);
    const TdScalars scp = td_reflector(xcol, n, j, xn2);
    td_panel_block(pg, idx, scp, &cbuf[0][0][0]);
    return;
  }
  const long I = (long)blockIdx.y - ny_panel + j1 / TS_STRIP; // strip index (absolute)
  const long seg = (long)blockIdx.x + j1 / TS_SEG;      // column segment (absolute)
  const long r0 = I * TS_STRIP;
  if (r0 >= n) return;
  const long cend_strip = std::min<long>(r0 + TS_STRIP, n); // columns <= last row of the strip
  const long cseg0 = seg * TS_SEG;
  if (cseg0 >= cend_strip) return;
  const double xnorm2 = td_sum_parts(ssbuf, nparts, &sh);
  const TdScalars sc = td_reflector(xcol, n, j, xnorm2);
  const double scale = sc.scale;
  const int t = threadIdx.x, cl = t & 31, rg = t >> 5;
  double ur[8], rowacc[8];
  long rr[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    rr[k] = r0 + rg * 8 + k;
    const bool ok = rr[k] >= j1 && rr[k] < n;
    ur[k] = ok ? ((rr[k] == j1) ? 1.0 : xcol[rr[k]] * scale) : 0.0;
    if (!ok) rr[k] = -1;
    rowacc[k] = 0.0;
  }
  const long c_lo = std::max<long>(cseg0, (j1 / TS_WIDE) * TS_WIDE), c_hi = std::min<long>(cseg0 + TS_SEG, cend_strip);
  int par = 0;
  for (long c0 = c_lo; c0 < c_hi; c0 += TS_WIDE, par ^= 1) {
    // 4 sub-tiles of 64 columns per pass: 32 independent 16-byte loads per thread in flight, one barrier per pass
    f64x2 a[4][8];
    const bool interior = (c0 + TS_WIDE <= r0) && (c0 >= j1); // strictly below the diagonal and inside the window
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long c = c0 + 64 * q + 2 * cl;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const long r = rr[k];
        a[q][k] = (r >= 0 && c < n && c <= r0 + 63) ? *reinterpret_cast<const f64x2 *>(A + r * n + c) : f64x2{0.0, 0.0};
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long c = c0 + 64 * q + 2 * cl;
      const double u0 = (c >= j1 && c < n) ? ((c == j1) ? 1.0 : xcol[c] * scale) : 0.0;
      const double u1 = (c + 1 >= j1 && c + 1 < n) ? ((c + 1 == j1) ? 1.0 : xcol[c + 1] * scale) : 0.0;
      double ca0 = 0.0, ca1 = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const long r = rr[k];
        if (interior) {
          rowacc[k] += a[q][k].x * u0 + a[q][k].y * u1;
          ca0 += a[q][k].x * ur[k];
          ca1 += a[q][k].y * ur[k];
        } else if (r >= 0) {
          const bool in0 = c >= j1, in1 = (c + 1 >= j1) && (c + 1 < n);
          if (in0 && c <= r) rowacc[k] += a[q][k].x * u0;
          if (in1 && c + 1 <= r) rowacc[k] += a[q][k].y * u1;
          if (in0 && c < r) ca0 += a[q][k].x * ur[k];
          if (in1 && c + 1 < r) ca1 += a[q][k].y * ur[k];
        }
      }
      cbuf[par][rg][64 * q + 2 * cl] = ca0;
      cbuf[par][rg][64 * q + 2 * cl + 1] = ca1;
    }
    __syncthreads();
    {
      const long cc = c0 + t;
      if (cc >= j1 && cc < cend_strip) {
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) v += cbuf[par][q][t];
        colP[I * n + cc] = v;
      }
    }
  }
  // row sums: butterfly over the 32 column lanes of each half-wave
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    double v = rowacc[k];
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (cl == 0 && rr[k] >= 0) rowP[seg * n + rr[k]] = v;
  }
}

// p[r] = sum_seg rowP[seg][r] + sum_{I' >= strip(r)} colP[I'][r], r in [j+1, n): 64 rows per block, four threads per
// row take every fourth partial (independent loads), fixed-order combine
__global__ __launch_bounds__(256) void td_symv_reduce_kernel(long n, long j, const double *__restrict__ rowP,
                                                             const double *__restrict__ colP, double *__restrict__ p,
                                                             int TS_SEG) {
  __shared__ double part[4][64];
  const long j1 = j + 1;
  const int rl = threadIdx.x & 63, pt = threadIdx.x >> 6;
  const long r = j1 + (long)blockIdx.x * 64 + rl;
  double s0 = 0.0, s1 = 0.0;
  if (r < n) {
    const long nstrip = (n + TS_STRIP - 1) / TS_STRIP;
    for (long sg = j1 / TS_SEG + pt; sg <= r / TS_SEG; sg += 4) s0 += rowP[sg * n + r];
    long I = r / TS_STRIP + pt;
    for (; I + 4 < nstrip; I += 8) {
      s0 += colP[I * n + r];
      s1 += colP[(I + 4) * n + r];
    }
    if (I < nstrip) s0 += colP[I * n + r];
  }
  part[pt][rl] = s0 + s1;
  __syncthreads();
  if (pt == 0 && r < n) p[r] = (part[0][rl] + part[1][rl]) + (part[2][rl] + part[3][rl]);
}

__global__ __launch_bounds__(256) void td_w1_kernel(long n, long j, long j0, const double *__restrict__ VT,
                                                    const double *__restrict__ WT, const double *__restrict__ p,
                                                    const double *__restrict__ ab, const double *tau,
                                                    double *__restrict__ wtmp, double *__restrict__ dotbuf) {
  __shared__ double sa[EIG_NB], sb[EIG_NB];
  __shared__ double part[4][TD_ROWS];
  const int t = threadIdx.x, rl = t & 63, qg = t >> 6;
  const int k = (int)(j - j0);
  for (int q = t; q < k; q += 256) {
    sa[q] = ab[2 * q];
    sb[q] = ab[2 * q + 1];
  }
  __syncthreads();
  const long r = j + 1 + (long)blockIdx.x * TD_ROWS + rl;
  double acc = 0.0;
  if (r < n)
    for (int q = qg; q < k; q += 4) acc += VT[(j0 + q) * n + r] * sa[q] + WT[(long)q * n + r] * sb[q];
  part[qg][rl] = acc;
  __syncthreads();
  if (qg == 0) {
    double dot = 0.0;
    if (r < n) {
      const double w = tau[j] * (p[r] - ((part[0][rl] + part[1][rl]) + (part[2][rl] + part[3][rl])));
      wtmp[r] = w;
      dot = w * VT[j * n + r];
    }
    dot = eig_wsum(dot);
    if (rl == 0) dotbuf[blockIdx.x] = dot;
  }
}

__global__ __launch_bounds__(TD_CHUNK) void td_w2_kernel(long n, long j, long j0, const double *__restrict__ VT,
                                                         double *__restrict__ WT, const double *__restrict__ wtmp,
                                                         const double *__restrict__ dotbuf, int nparts,
                                                         const double *tau) {
  __shared__ double sh;
  const double dot = td_sum_parts(dotbuf, nparts, &sh);
  const double alpha2 = -0.5 * tau[j] * dot;
  const long r = (long)blockIdx.x * TD_CHUNK + threadIdx.x;
  if (r < n) {
    double w = 0.0;
    if (r > j) w = wtmp[r] + alpha2 * VT[j * n + r];
    WT[(long)(j - j0) * n + r] = w;
  }
}

// ---------------------------------------------------------------- 2. divide and conquer
// Leaf: implicit QL with Wilkinson shift (EISPACK tql2).  One wavefront per leaf (size <= 64):
// lane k owns row k of the eigenvector matrix (LDS), the scalar recurrences run in every lane.
__global__ __launch_bounds__(64) void dc_leaf_kernel(const double *__restrict__ d, const double *__restrict__ e,
                                                     const int *__restrict__ leaf_lo,
                                                     const int *__restrict__ leaf_sz, double *__restrict__ dout,
                                                     double *__restrict__ QT, long n, int *info) {
  __shared__ double sd[EIG_LEAF], se[EIG_LEAF + 1];
  __shared__ double z[EIG_LEAF * (EIG_LEAF + 1)];
  const int lo = leaf_lo[blockIdx.x], s = leaf_sz[blockIdx.x];
  const int lane = threadIdx.x;
  if (lane < s) {
    sd[lane] = d[lo + lane];
    se[lane] = (lane + 1 < s) ? e[lo + lane] : 0.0;
    for (int c = 0; c < s; ++c) z[lane * (EIG_LEAF + 1) + c] = (lane == c) ? 1.0 : 0.0;
  }
  __syncthreads();
  const double eps = 2.220446049250313e-16;
  double f = 0.0, tst1 = 0.0;
  for (int l = 0; l < s; ++l) {
    tst1 = fmax(tst1, fabs(sd[l]) + fabs(se[l]));
    int m = l;
    while (m < s) {
      if (fabs(se[m]) <= eps * tst1) break;
      m++;
    }
    if (m > l) {
      int iter = 0;
      do {
        iter++;
        double g = sd[l];
        double p = (sd[l + 1] - g) / (2.0 * se[l]);
        double r = hypot(p, 1.0);
        if (p < 0) r = -r;
        const double dl0 = se[l] / (p + r);
        const double dl1 = se[l] * (p + r);
        double h = g - dl0;
        __syncthreads();
        if (lane == 0) {
          sd[l] = dl0;
          sd[l + 1] = dl1;
        }
        if (lane >= l + 2 && lane < s) sd[lane] -= h;
        __syncthreads();
        f += h;
        p = sd[m];
        double c = 1.0, c2 = c, c3 = c;
        const double el1 = se[l + 1];
        double sn = 0.0, s2 = 0.0;
        for (int i = m - 1; i >= l; --i) {
          c3 = c2;
          c2 = c;
          s2 = sn;
          const double ei = se[i], di = sd[i];
          g = c * ei;
          h = c * p;
          r = hypot(p, ei);
          const double e_ip1 = sn * r;
          sn = ei / r;
          c = p / r;
          p = c * di - sn * g;
          const double d_ip1 = h + sn * (c * g + sn * di);
          __syncthreads();
          if (lane == 0) {
            se[i + 1] = e_ip1;
            sd[i + 1] = d_ip1;
          }
          if (lane < s) {
            double *zr = z + lane * (EIG_LEAF + 1);
            const double hv = zr[i + 1];
            zr[i + 1] = sn * zr[i] + c * hv;
            zr[i] = c * zr[i] - sn * hv;
          }
          __syncthreads();
        }
        p = -sn * s2 * c3 * el1 * se[l] / dl1;
        __syncthreads();
        if (lane == 0) {
          se[l] = sn * p;
          sd[l] = c * p;
        }
        __syncthreads();
      } while (fabs(se[l]) > eps * tst1 && iter < 200);
      if (iter >= 200 && lane == 0) atomicAdd(info, 1);
    }
    __syncthreads();
    if (lane == 0) {
      sd[l] = sd[l] + f;
      se[l] = 0.0;
    }
    __syncthreads();
  }
  // eigenvector c -> row lo + c of QT, columns lo .. lo+s-1
  if (lane < s) dout[lo + lane] = sd[lane];
  for (int c = 0; c < s; ++c)
    if (lane < s) QT[(long)(lo + c) * n + lo + lane] = z[lane * (EIG_LEAF + 1) + c];
}

// z[i] = component of eigenvector row (lo+i) at the split: last of T1 (column mid-1) / first of T2
__global__ void dc_gather_z_kernel(const double *__restrict__ Q, long n, int lo, int mid, int ns,
                                   double *__restrict__ z) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns) return;
  const long r = lo + i;
  z[i] = Q[r * n + (r < mid ? mid - 1 : mid)];
}

struct GivensRot {
  int ra, rb;
  double c, s;
};
// deflation rotations, in order, on eigenvector rows (columns lo .. lo+ns-1)
__global__ void dc_givens_kernel(double *__restrict__ Q, long n, int lo, int ns,
                                 const GivensRot *__restrict__ rot, int nrot) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= ns) return;
  for (int t = 0; t < nrot; ++t) {
    const GivensRot g = rot[t];
    double *pa = Q + (long)(lo + g.ra) * n + lo + col;
    double *pb = Q + (long)(lo + g.rb) * n + lo + col;
    const double qa = *pa, qb = *pb;
    *pa = g.c * qa + g.s * qb;
    *pb = -g.s * qa + g.c * qb;
  }
}

// One wavefront per root j of  1 + rho * sum_i w_i^2 / (dl_i - lambda) = 0  (dl ascending, rho > 0,
// sum w^2 <= 1).  The root lies in (dl_j, dl_{j+1}) (last: (dl_k, dl_k + rho]); it is located by
// bisection on mu = lambda - dl_org with the origin at the nearer pole, so that the differences
// dl_i - lambda_j (row j of Delta) keep high relative accuracy -- what the Loewner step needs.
__global__ __launch_bounds__(256) void dc_secular_kernel(const double *__restrict__ dl,
                                                         const double *__restrict__ w, double rho, int k,
                                                         double *__restrict__ lam, double *__restrict__ Delta) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= k) return;
  int org = j;
  double lo_, hi_;
  auto fsec = [&](int o, double mu) {
    const double dorg = dl[o];
    double s = 0.0;
    for (int i = lane; i < k; i += 64) {
      const double wi = w[i];
      s += wi * wi / ((dl[i] - dorg) - mu);
    }
    return 1.0 + rho * eig_wsum(s);
  };
  if (j < k - 1) {
    const double mid = 0.5 * (dl[j + 1] - dl[j]);
    const double fm = fsec(j, mid);
    if (fm > 0.0) {
      org = j; lo_ = 0.0; hi_ = mid;
    } else {
      org = j + 1; lo_ = -mid; hi_ = 0.0;
    }
  } else {
    org = j; lo_ = 0.0; hi_ = rho;
  }
  double mu = 0.5 * (lo_ + hi_);
  for (int it = 0; it < 1200; ++it) {
    mu = 0.5 * (lo_ + hi_);
    if (mu == lo_ || mu == hi_) break;
    const double f = fsec(org, mu);
    if (f > 0.0) hi_ = mu; else lo_ = mu;
  }
  mu = 0.5 * (lo_ + hi_);
  if (mu == 0.0) mu = (org == j) ? hi_ : lo_;
  if (lane == 0) lam[j] = dl[org] + mu;
  const double dorg = dl[org];
  double *row = Delta + (long)j * k;
  for (int i = lane; i < k; i += 64) row[i] = (dl[i] - dorg) - mu;
}

// Loewner / Gu-Eisenstat: zhat_i = sign(w_i) sqrt| (dl_i - lam_i) prod_{j != i} (dl_i - lam_j)/(dl_i - dl_j) |
__global__ __launch_bounds__(256) void dc_zhat_kernel(const double *__restrict__ dl, const double *__restrict__ w,
                                                      const double *__restrict__ Delta, int k,
                                                      double *__restrict__ zhat) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= k) return;
  const double di = dl[i];
  double pr = 1.0;
  for (int j = lane; j < k; j += 64) {
    const double num = Delta[(long)j * k + i];
    pr *= (j == i) ? num : num / (di - dl[j]);
  }
  pr = eig_wprod(pr);
  if (lane == 0) zhat[i] = copysign(sqrt(fabs(pr)), w[i]);
}

// row j of Delta -> unit eigenvector of D + rho z z^T: u_j[i] = zhat_i / (dl_i - lam_j)
__global__ __launch_bounds__(256) void dc_eigvec_kernel(double *__restrict__ Delta,
                                                        const double *__restrict__ zhat, int k) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= k) return;
  double *row = Delta + (long)j * k;
  double ss = 0.0;
  for (int i = lane; i < k; i += 64) {
    const double v = zhat[i] / row[i];
    row[i] = v;
    ss += v * v;
  }
  const double inv = 1.0 / sqrt(eig_wsum(ss));
  for (int i = lane; i < k; i += 64) row[i] *= inv;
}

// dst[t][c] = src[lo + rows[t]][col0 + c]
__global__ void dc_gather_rows_kernel(const double *__restrict__ src, long n, int lo, const int *__restrict__ rows,
                                      int count, int col0, int ncols, double *__restrict__ dst, long ld_dst) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  if (c >= ncols || t >= count) return;
  dst[(long)t * ld_dst + c] = src[(long)(lo + rows[t]) * n + col0 + c];
}

// dst[j][p] = src[j][cols[p]] (src k x k with leading dimension k, dst k x count)
__global__ void dc_gather_cols_kernel(const double *__restrict__ src, int k, const int *__restrict__ cols, int count,
                                      double *__restrict__ dst, long ld_dst) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y;
  if (p >= ld_dst || j >= k) return;
  dst[(long)j * ld_dst + p] = p < count ? src[(long)j * k + cols[p]] : 0.0; // the padding column of an odd count
}

// ---------------------------------------------------------------- 3. back-transformation
// forward compact-WY factor (LAPACK dlarft, columnwise): T(i,i) = tau_i,
// T(0:i,i) = -tau_i T(0:i,0:i) S(0:i,i) with S = Y^T Y.  One workgroup per panel (blockIdx.x): the panels' factors are
// independent, so all of them are formed by ONE launch after the tridiagonalisation (the recurrence is a serial chain
// of kp steps, ~1.2 ms on its own).  S has leading dimension lds, T is kp x kp upper triangular with leading dimension kp.
__global__ __launch_bounds__(256) void bt_tfactor_kernel(const double *__restrict__ Sbase, long s_stride, int lds,
                                                         const double *__restrict__ tau_base, long n,
                                                         double *__restrict__ Tbase, long t_stride) {
  const long pnl = blockIdx.x;
  const double *__restrict__ S = Sbase + pnl * s_stride;
  const double *__restrict__ tau = tau_base + pnl * EIG_NB;
  double *__restrict__ T = Tbase + pnl * t_stride;
  const long rem = n - pnl * EIG_NB;
  const int kp = (int)(rem < EIG_NB ? rem : EIG_NB);
  const int t = threadIdx.x;
  for (int idx = t; idx < kp * kp; idx += 256) T[idx] = 0.0;
  __syncthreads();
  for (int i = 0; i < kp; ++i) {
    const double ti = tau[i];
    if (t < i) {
      double acc = 0.0;
      for (int c = t; c < i; ++c) acc += T[t * kp + c] * S[c * lds + i];
      T[t * kp + i] = -ti * acc;
    }
    if (t == 0) T[i * kp + i] = ti;
    __syncthreads();
  }
}

// U[r][t] = ZT[perm[t]][r];  eval[t] = dphys[perm[t]] * scale
__global__ void bt_transpose_perm_kernel(const double *__restrict__ ZT, long n, const int *__restrict__ perm,
                                         const double *__restrict__ dphys, double scale, double *__restrict__ U,
                                         double *__restrict__ eval) {
  __shared__ double tile[32][33];
  const long t0 = (long)blockIdx.x * 32, r0 = (long)blockIdx.y * 32;
  const int tx = threadIdx.x, ty = threadIdx.y; // 32 x 8
  for (int a = ty; a < 32; a += 8) {
    const long t = t0 + a, r = r0 + tx;
    tile[a][tx] = (t < n && r < n) ? ZT[(long)perm[t] * n + r] : 0.0;
  }
  __syncthreads();
  for (int a = ty; a < 32; a += 8) {
    const long r = r0 + a, t = t0 + tx;
    if (r < n && t < n) U[r * n + t] = tile[tx][a];
  }
  if (blockIdx.y == 0 && ty == 0) {
    const long t = t0 + tx;
    if (t < n) eval[t] = dphys[perm[t]] * scale;
  }
}

__global__ void eig_scale_kernel(double *v, long n, double s) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] *= s;
}

// ---------------------------------------------------------------- host orchestration
// The workspace of a solve is ~5 n^2 doubles in thirty-odd buffers, and hipMalloc / hipFree of such sizes cost 25-50 ms per GB
// (profiles/r05_alloc_probe.txt): 0.1-0.2 s of a 1.9 s solve at n = 20 000, 1.0-2.2 s of 19 s at n = 50 000 -- per call, because the
// buffers were freed on the way out.  Round 6: the buffers of one solve can stay in a pool between calls.
// Off unless asked for -- gemma_hip_eigh_reserve(n) (allocates the workspace of order n ahead of the solve: a caller that knows its n
// does this while it reads files) or GEMMA_HIP_EIGH_CACHE=1 (every solve leaves its buffers behind) -- and gemma_hip_eigh_release()
// hands the memory back; lmm_setup* drops a pool that holds more than a quarter of the device.  A request is served by the smallest
// idle block of at least its size and at most 9/8 of it (the same n again: exact fits in the order of allocation).
struct EigPool {
  struct Blk { void *p; size_t bytes; bool used; };
  std::vector<Blk> blks;
  bool keep = false;
  static bool env_keep() {
    const char *e = getenv("GEMMA_HIP_EIGH_CACHE");
    return e && e[0] == '1';
  }
  void *take(size_t bytes) {
    long best = -1;
    for (size_t i = 0; i < blks.size(); ++i)
      if (!blks[i].used && blks[i].bytes >= bytes && blks[i].bytes <= bytes + bytes / 8 + 4096 &&
          (best < 0 || blks[i].bytes < blks[(size_t)best].bytes))
        best = (long)i;
    if (best >= 0) {
      blks[(size_t)best].used = true;
      // A recycled block carries the previous solve's bytes.  The solver reads nothing it has not written: GEMMA_HIP_EIGH_POISON=1
      // (diagnostics) fills every block, fresh ones too, with NaN bit patterns, and every stage returns the same bits
      // (scripts/exp/r6_dbg2.py: one- and two-stage paths, the stage diagnostics).
      if (poison() && hipMemset(blks[(size_t)best].p, 0xFF, blks[(size_t)best].bytes) != hipSuccess) (void)hipGetLastError();
      return blks[(size_t)best].p;
    }
    void *q = nullptr;
    if (hipMalloc(&q, bytes) != hipSuccess) {
      (void)hipGetLastError();
      // idle blocks of other sizes may be what stands in the way: hand them back and try once more
      if (drop_idle() == 0) return nullptr;
      if (hipMalloc(&q, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
      }
    }
    if (poison() && hipMemset(q, 0xFF, bytes) != hipSuccess) (void)hipGetLastError();
    blks.push_back({q, bytes, true});
    return q;
  }
  static bool poison() {
    const char *e = getenv("GEMMA_HIP_EIGH_POISON");
    return e && e[0] == '1';
  }
  void give(void *q) {
    for (size_t i = 0; i < blks.size(); ++i)
      if (blks[i].p == q) {
        if (keep || env_keep()) {
          blks[i].used = false;
        } else {
          (void)hipFree(q);
          blks.erase(blks.begin() + (long)i);
        }
        return;
      }
    (void)hipFree(q); // not ours (cannot happen): still the caller's to free
  }
  size_t idle_bytes() const {
    size_t b = 0;
    for (const Blk &k : blks)
      if (!k.used) b += k.bytes;
    return b;
  }
  size_t drop_idle() {
    size_t freed = 0;
    for (size_t i = blks.size(); i-- > 0;)
      if (!blks[i].used) {
        (void)hipFree(blks[i].p);
        freed += blks[i].bytes;
        blks.erase(blks.begin() + (long)i);
      }
    return freed;
  }
};
static inline EigPool &eig_pool() {
  static EigPool pool;
  return pool;
}

struct EigWs {
  long n = 0;
  double *VT = nullptr, *WT = nullptr, *xcol = nullptr, *p = nullptr, *ab = nullptr;
  double *ssbuf = nullptr, *dotbuf = nullptr, *wtmp = nullptr;
  int segw = 1024;                         // column-segment width of the symmetric SYMV (512 or 1024)
  double *rowP = nullptr, *colP = nullptr; // symmetric SYMV partials (nullptr: row-per-wave SYMV)
  double *d = nullptr, *e = nullptr, *tau = nullptr;
  double *Delta = nullptr, *Wk = nullptr, *QB = nullptr;
  double *P = nullptr, *P2 = nullptr, *S = nullptr, *T = nullptr;
  double *Tall = nullptr; // per panel: compact-WY factor (kp x kp, ld kp), formed in one launch
  double *Sall = nullptr; // per panel: strict upper triangle of Y Y^T, collected by the tridiagonalisation (ld EIG_NB)
  double *zbuf = nullptr, *dl = nullptr, *w = nullptr, *lam = nullptr, *zhat = nullptr, *dphys = nullptr;
  int *ibuf = nullptr, *info = nullptr;
  int *ibuf2 = nullptr; // structured merges of the divide & conquer: row / column lists of the two children (4 n ints); optional
  GivensRot *rot = nullptr;
  std::vector<void *> owned;
  template <class Tp> bool get(Tp *&ptr, size_t count) {
    void *q = eig_pool().take(std::max<size_t>(count, 1) * sizeof(Tp));
    if (!q) return false;
    owned.push_back(q);
    ptr = reinterpret_cast<Tp *>(q);
    return true;
  }
  void release() {
    for (void *q : owned) eig_pool().give(q);
    owned.clear();
  }
};

#define EIG_HIP(expr)                                                                  \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) {                                                            \
      msg = std::string(#expr) + ": " + hipGetErrorString(e_);                         \
      return 4;                                                                        \
    }                                                                                  \
  } while (0)

// A (n x n, both triangles) -> d, e, tau, reflectors in ws.VT (row j = u_j).  A is destroyed.
static inline int eig_tridiagonalize(double *A, long n, EigWs &ws, hipStream_t s, std::string &msg) {
  EIG_HIP(hipMemsetAsync(ws.VT, 0, (size_t)n * n * 8, s));
  // symmetric SYMV for trailing sizes >= sym_min: below that the row-per-wave form has less overhead
  // (GEMMA_HIP_EIGH_SYMV_MIN overrides the switch-over; read per call so that tests can force the path)
  const char *esm = getenv("GEMMA_HIP_EIGH_SYMV_MIN");
  const long sym_min = esm ? atol(esm) : 8192;
  const char *esg = getenv("GEMMA_HIP_EIGH_SEG");
  ws.segw = (esg && atoi(esg) == 512) ? 512 : 1024;
  for (long j0 = 0; j0 < n; j0 += EIG_NB) {
    const long kp = std::min<long>(EIG_NB, n - j0);
    for (long k = 0; k < kp; ++k) {
      const long j = j0 + k;
      const long m = n - j - 1;
      const int nparts = (int)((n - j + TD_ROWS - 1) / TD_ROWS);
      hipLaunchKernelGGL(td_col_kernel, dim3(nparts), dim3(256), 0, s, A, n, j, j0, ws.VT, ws.WT, ws.xcol,
                         ws.ssbuf);
      const bool sym = ws.rowP != nullptr && m >= sym_min;
      const int nsymv = sym ? 0 : (int)((m + 3) / 4);
      const int nrow = (int)((n + 255) / 256);
      TdPanel pg{n, j, j0, ws.VT, ws.WT, ws.xcol, ws.ab, (int)k, ws.d, ws.e, ws.tau,
                 ws.Sall ? ws.Sall + (j0 / EIG_NB) * EIG_NB * EIG_NB : nullptr};
      const int n_panel = 2 * (int)k + nrow;
      if (sym) {
        const long j1 = j + 1;
        const int TS_SEG = ws.segw;
        const unsigned gx = (unsigned)((n - 1) / TS_SEG - j1 / TS_SEG + 1);
        const unsigned gy = (unsigned)((n - 1) / TS_STRIP - j1 / TS_STRIP + 1);
        const unsigned nyp = ((unsigned)n_panel + gx - 1) / gx;
        hipLaunchKernelGGL(td_symv_sym_kernel, dim3(gx, gy + nyp), dim3(256), 0, s, A, n, j, ws.xcol, ws.ssbuf, nparts,
                           ws.rowP, ws.colP, pg, (int)nyp, n_panel, TS_SEG);
        hipLaunchKernelGGL(td_symv_reduce_kernel, dim3((unsigned)((m + 63) / 64)), dim3(256), 0, s, n, j, ws.rowP,
                           ws.colP, ws.p, TS_SEG);
      } else {
        hipLaunchKernelGGL(td_symv_kernel, dim3(nsymv + n_panel), dim3(256), 0, s, A, pg, ws.ssbuf, nparts, ws.p,
                           nsymv);
      }
      const int nparts2 = (int)((m + TD_ROWS - 1) / TD_ROWS);
      if (m > 0)
        hipLaunchKernelGGL(td_w1_kernel, dim3(nparts2), dim3(256), 0, s, n, j, j0, ws.VT, ws.WT, ws.p, ws.ab,
                           ws.tau, ws.wtmp, ws.dotbuf);
      hipLaunchKernelGGL(td_w2_kernel, dim3((unsigned)((n + TD_CHUNK - 1) / TD_CHUNK)), dim3(TD_CHUNK), 0, s, n, j,
                         j0, ws.VT, ws.WT, ws.wtmp, ws.dotbuf, nparts2, ws.tau);
    }
    EIG_HIP(hipGetLastError());
    const long t0 = j0 + kp;
    if (t0 < n) {
      const long M = n - t0;
      // A22 -= V W^T + W V^T  (operands stored [k = q][m], i.e. 'T','N')
      EIG_HIP(launch_dgemm('T', 'N', M, M, kp, -1.0, ws.VT + j0 * n + t0, n, ws.WT + t0, n, 1.0, A + t0 * n + t0,
                           n, false, false, s));
      EIG_HIP(launch_dgemm('T', 'N', M, M, kp, -1.0, ws.WT + t0, n, ws.VT + j0 * n + t0, n, 1.0, A + t0 * n + t0,
                           n, false, false, s));
    }
  }
  return 0;
}

// Tridiagonal (hd, he: host copies, length n / n-1) -> eigenvectors as rows of *Zout (one of QA/QB),
// eigenvalues hd_phys in physical row order.  QA and QB are n x n device buffers.
static inline int eig_stedc(long n, std::vector<double> &hd, std::vector<double> &he, double *QA, double *QB,
                            EigWs &ws, hipStream_t s, double **Zout, std::vector<double> &dphys,
                            std::string &msg) {
  // static tree: split every block `levels` times so that all leaves sit at the same depth
  int levels = 0;
  while (((n + (1L << levels) - 1) >> levels) > EIG_LEAF) ++levels;
  std::vector<std::vector<int>> bounds(levels + 1);
  bounds[0] = {0, (int)n};
  for (int L = 0; L < levels; ++L) {
    std::vector<int> nb;
    for (size_t b = 0; b + 1 < bounds[L].size(); ++b) {
      const int lo = bounds[L][b], hi = bounds[L][b + 1];
      nb.push_back(lo);
      nb.push_back(lo + (hi - lo) / 2);
    }
    nb.push_back((int)n);
    bounds[L + 1] = nb;
  }
  // rank-one tearing at every internal boundary (Cuppen): d[m-1] -= |rho|, d[m] -= |rho|, rho = e[m-1]
  const std::vector<int> &leafb = bounds[levels];
  for (size_t b = 1; b + 1 < leafb.size(); ++b) {
    const int m = leafb[b];
    const double rho = he[m - 1];
    hd[m - 1] -= std::fabs(rho);
    hd[m] -= std::fabs(rho);
  }
  EIG_HIP(hipMemcpyAsync(ws.d, hd.data(), n * 8, hipMemcpyHostToDevice, s));
  EIG_HIP(hipMemcpyAsync(ws.e, he.data(), (n - 1) * 8, hipMemcpyHostToDevice, s));
  const int nleaf = (int)leafb.size() - 1;
  std::vector<int> hl(2 * nleaf);
  for (int b = 0; b < nleaf; ++b) {
    hl[b] = leafb[b];
    hl[nleaf + b] = leafb[b + 1] - leafb[b];
    if (hl[nleaf + b] > EIG_LEAF || hl[nleaf + b] < 1) {
      msg = "internal: bad leaf size";
      return 4;
    }
  }
  EIG_HIP(hipMemcpyAsync(ws.ibuf, hl.data(), hl.size() * sizeof(int), hipMemcpyHostToDevice, s));
  EIG_HIP(hipMemsetAsync(QA, 0, (size_t)n * n * 8, s));
  EIG_HIP(hipMemsetAsync(ws.info, 0, sizeof(int), s));
  hipLaunchKernelGGL(dc_leaf_kernel, dim3(nleaf), dim3(64), 0, s, ws.d, ws.e, ws.ibuf, ws.ibuf + nleaf, ws.dphys,
                     QA, n, ws.info);
  EIG_HIP(hipGetLastError());
  dphys.resize(n);
  int hinfo = 0;
  EIG_HIP(hipMemcpyAsync(dphys.data(), ws.dphys, n * 8, hipMemcpyDeviceToHost, s));
  EIG_HIP(hipMemcpyAsync(&hinfo, ws.info, sizeof(int), hipMemcpyDeviceToHost, s));
  EIG_HIP(hipStreamSynchronize(s));
  if (hinfo != 0) {
    msg = "QL iteration did not converge in a leaf";
    return 6;
  }

  double *Qc = QA, *Qn = QB;
  const double EPS = 2.220446049250313e-16;
  std::vector<double> z, dl, wv, lam;
  std::vector<int> order, keep, defl, ctype, l13, l23, hb;
  std::vector<GivensRot> rots;
  for (int L = levels; L >= 1; --L) {
    const std::vector<int> &bl = bounds[L];
    EIG_HIP(hipMemsetAsync(Qn, 0, (size_t)n * n * 8, s));
    for (size_t b = 0; b + 2 < bl.size() + 0; b += 2) {
      const int lo = bl[b], mid = bl[b + 1], hi = bl[b + 2];
      const int n1 = mid - lo, ns = hi - lo;
      double rho = he[mid - 1];
      z.resize(ns);
      hipLaunchKernelGGL(dc_gather_z_kernel, dim3((ns + 255) / 256), dim3(256), 0, s, Qc, n, lo, mid, ns, ws.zbuf);
      EIG_HIP(hipMemcpyAsync(z.data(), ws.zbuf, ns * 8, hipMemcpyDeviceToHost, s));
      EIG_HIP(hipStreamSynchronize(s));
      double *dd = dphys.data() + lo; // physical-order eigenvalues of the two children
      if (rho < 0) {
        for (int i = n1; i < ns; ++i) z[i] = -z[i];
        rho = -rho;
      }
      const double rs2 = 1.0 / std::sqrt(2.0);
      for (int i = 0; i < ns; ++i) z[i] *= rs2;
      rho *= 2.0;
      order.resize(ns);
      for (int i = 0; i < ns; ++i) order[i] = i;
      std::stable_sort(order.begin(), order.end(), [&](int a, int c) { return dd[a] < dd[c]; });
      double dmax = 0.0, zmax = 0.0;
      for (int i = 0; i < ns; ++i) {
        dmax = std::max(dmax, std::fabs(dd[i]));
        zmax = std::max(zmax, std::fabs(z[i]));
      }
      const double tol = 8.0 * EPS * std::max(dmax, zmax);
      keep.clear();
      defl.clear();
      rots.clear();
      // which child's columns a row of the children's eigenvector matrix occupies: 1, 2, or 3 = both (a Givens rotation of the
      // deflation below has mixed a row of each child) -- LAPACK's dlaed2 column types; used by the structured product below
      ctype.assign(ns, 1);
      for (int i = n1; i < ns; ++i) ctype[i] = 2;
      if (rho * zmax <= tol) {
        for (int i = 0; i < ns; ++i) defl.push_back(order[i]);
      } else {
        int pj = -1;
        for (int oi = 0; oi < ns; ++oi) {
          const int i = order[oi];
          if (rho * std::fabs(z[i]) <= tol) {
            defl.push_back(i);
            continue;
          }
          if (pj < 0) {
            pj = i;
            continue;
          }
          double sn = z[pj], cs = z[i];
          const double tau = std::hypot(cs, sn);
          const double t = dd[i] - dd[pj];
          cs /= tau;
          sn = -sn / tau;
          if (std::fabs(t * cs * sn) <= tol) {
            z[i] = tau;
            z[pj] = 0.0;
            rots.push_back(GivensRot{pj, i, cs, sn});
            if (ctype[pj] != ctype[i]) ctype[pj] = ctype[i] = 3;
            const double tt = dd[pj] * cs * cs + dd[i] * sn * sn;
            dd[i] = dd[pj] * sn * sn + dd[i] * cs * cs;
            dd[pj] = tt;
            defl.push_back(pj);
            pj = i;
          } else {
            keep.push_back(pj);
            pj = i;
          }
        }
        if (pj >= 0) keep.push_back(pj);
      }
      const int k = (int)keep.size();
      if (!rots.empty()) {
        EIG_HIP(hipMemcpyAsync(ws.rot, rots.data(), rots.size() * sizeof(GivensRot), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(dc_givens_kernel, dim3((ns + 255) / 256), dim3(256), 0, s, Qc, n, lo, ns, ws.rot,
                           (int)rots.size());
      }
      std::vector<double> dnew(ns);
      if (k > 0) {
        std::stable_sort(keep.begin(), keep.end(), [&](int a, int c) { return dd[a] < dd[c]; });
        dl.resize(k);
        wv.resize(k);
        for (int t = 0; t < k; ++t) {
          dl[t] = dd[keep[t]];
          wv[t] = z[keep[t]];
        }
        EIG_HIP(hipMemcpyAsync(ws.dl, dl.data(), k * 8, hipMemcpyHostToDevice, s));
        EIG_HIP(hipMemcpyAsync(ws.w, wv.data(), k * 8, hipMemcpyHostToDevice, s));
        EIG_HIP(hipMemcpyAsync(ws.ibuf, keep.data(), k * sizeof(int), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(dc_secular_kernel, dim3((k + 3) / 4), dim3(256), 0, s, ws.dl, ws.w, rho, k, ws.lam,
                           ws.Delta);
        hipLaunchKernelGGL(dc_zhat_kernel, dim3((k + 3) / 4), dim3(256), 0, s, ws.dl, ws.w, ws.Delta, k, ws.zhat);
        hipLaunchKernelGGL(dc_eigvec_kernel, dim3((k + 3) / 4), dim3(256), 0, s, ws.Delta, ws.zhat, k);
        // new eigenvector rows lo .. lo+k-1 of Qn:  R = Uk (k x k, row j = vector j) * Wk (k x ns), Wk = the kept rows of the
        // children's matrix.  A row of child 1 is zero in child 2's columns and vice versa, so the product is two of half the size
        // (round 4; LAPACK's dlaed3 does the same): R[:, 0:n1] = Uk[:, types 1,3] * W[types 1,3][0:n1], R[:, n1:] likewise.  The
        // column subsets of Uk and the row subsets of W are gathered into Wk (order kept: the sums run over the same non-zero terms
        // in the same order).  Small merges, merges whose gathers would not fit Wk, and GEMMA_HIP_EIGH_DC_STRUCT=0 take the dense form.
        static const bool dc_struct = !(getenv("GEMMA_HIP_EIGH_DC_STRUCT") && getenv("GEMMA_HIP_EIGH_DC_STRUCT")[0] == '0');
        l13.clear();
        l23.clear();
        for (int t = 0; t < k; ++t) {
          if (ctype[keep[t]] != 2) l13.push_back(t);
          if (ctype[keep[t]] != 1) l23.push_back(t);
        }
        const int k13 = (int)l13.size(), k23 = (int)l23.size(), n2 = ns - n1;
        // even leading dimensions (the GEMM's aligned path); the padding column of Uk multiplies nothing (K = the true count)
        const long ld13 = k13 + (k13 & 1), ld23 = k23 + (k23 & 1), ldn1 = n1 + (n1 & 1), ldn2 = n2 + (n2 & 1);
        const size_t need = std::max((size_t)k * ld13 + (size_t)k13 * ldn1, (size_t)k * ld23 + (size_t)k23 * ldn2);
        if (dc_struct && ws.ibuf2 && k >= 256 && need <= (size_t)n * n && (size_t)2 * k + 2 * k <= 4 * (size_t)n) {
          // ibuf2: [0, k13) positions t of types 1,3; [k, k + k13) their rows keep[t]; [2k, ..) and [3k, ..) the same for types 2,3
          hb.assign((size_t)4 * k, 0); // lives until the stream synchronisation at the end of this merge
          for (int p = 0; p < k13; ++p) { hb[p] = l13[p]; hb[(size_t)k + p] = keep[l13[p]]; }
          for (int p = 0; p < k23; ++p) { hb[(size_t)2 * k + p] = l23[p]; hb[(size_t)3 * k + p] = keep[l23[p]]; }
          EIG_HIP(hipMemcpyAsync(ws.ibuf2, hb.data(), hb.size() * sizeof(int), hipMemcpyHostToDevice, s));
          for (int half = 0; half < 2; ++half) {
            const int kk = half ? k23 : k13, nc = half ? n2 : n1, c0 = half ? n1 : 0;
            const long ldu = half ? ld23 : ld13, ldw = half ? ldn2 : ldn1;
            if (kk == 0 || nc == 0) continue;
            double *Uc = ws.Wk, *Wc = ws.Wk + (size_t)k * ldu;
            hipLaunchKernelGGL(dc_gather_cols_kernel, dim3((unsigned)((ldu + 255) / 256), k), dim3(256), 0, s, ws.Delta, k,
                               ws.ibuf2 + (size_t)(2 * half) * k, kk, Uc, ldu);
            hipLaunchKernelGGL(dc_gather_rows_kernel, dim3((nc + 255) / 256, kk), dim3(256), 0, s, Qc, n, lo,
                               ws.ibuf2 + (size_t)(2 * half + 1) * k, kk, lo + c0, nc, Wc, ldw);
            EIG_HIP(hipGetLastError());
            EIG_HIP(launch_dgemm('N', 'N', k, nc, kk, 1.0, Uc, ldu, Wc, ldw, 0.0, Qn + (long)lo * n + lo + c0, n, false, false, s));
          }
        } else {
          hipLaunchKernelGGL(dc_gather_rows_kernel, dim3((ns + 255) / 256, k), dim3(256), 0, s, Qc, n, lo, ws.ibuf,
                             k, lo, ns, ws.Wk, (long)ns);
          EIG_HIP(hipGetLastError());
          EIG_HIP(launch_dgemm('N', 'N', k, ns, k, 1.0, ws.Delta, k, ws.Wk, ns, 0.0, Qn + (long)lo * n + lo, n, false,
                               false, s));
        }
        lam.resize(k);
        EIG_HIP(hipMemcpyAsync(lam.data(), ws.lam, k * 8, hipMemcpyDeviceToHost, s));
      }
      const int nd = (int)defl.size();
      if (nd > 0) {
        EIG_HIP(hipMemcpyAsync(ws.ibuf + ns, defl.data(), nd * sizeof(int), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(dc_gather_rows_kernel, dim3((ns + 255) / 256, nd), dim3(256), 0, s, Qc, n, lo,
                           ws.ibuf + ns, nd, lo, ns, Qn + (long)(lo + k) * n + lo, n);
        EIG_HIP(hipGetLastError());
      }
      EIG_HIP(hipStreamSynchronize(s)); // lam on the host; ibuf/rot/dl reusable
      for (int t = 0; t < k; ++t) dnew[t] = lam[t];
      for (int t = 0; t < nd; ++t) dnew[k + t] = dd[defl[t]];
      for (int t = 0; t < ns; ++t) dd[t] = dnew[t];
      for (int t = 0; t < ns; ++t)
        if (!std::isfinite(dd[t])) {
          msg = "non-finite eigenvalue in a divide-and-conquer merge";
          return 6;
        }
    }
    std::swap(Qc, Qn);
  }
  *Zout = Qc;
  return 0;
}

// ZT (rows = eigenvectors of T) <- ZT * H_{n-3} ... H_0, panel by panel from the last one.
// Y = VT[j0 : j0+kp, j0:] -- reflector j0+q is zero in columns <= j0+q, so starting at column j0 (not j0+1) only adds
// a zero column and keeps every operand 16-byte aligned with K a multiple of the GEMM's K tile when n is.
static inline int eig_backtransform(double *ZT, long n, EigWs &ws, hipStream_t s, std::string &msg) {
  const long npan = (n + EIG_NB - 1) / EIG_NB;
  const long nb2 = (long)EIG_NB * EIG_NB;
  if (ws.Sall) {
    // Y Y^T's strict upper triangle (all dlarft reads) was stored by the tridiagonalisation's panel dots
    hipLaunchKernelGGL(bt_tfactor_kernel, dim3((unsigned)npan), dim3(256), 0, s, ws.Sall, nb2, EIG_NB, ws.tau, n,
                       ws.Tall, nb2);
    EIG_HIP(hipGetLastError());
  }
  for (long pnl = npan - 1; pnl >= 0; --pnl) {
    const long j0 = pnl * EIG_NB;
    const long kp = std::min<long>(EIG_NB, n - j0);
    const long c0 = j0;
    const long Kc = n - c0;
    const double *Y = ws.VT + j0 * n + c0; // kp x Kc, ld n
    const double *T = ws.T;
    if (ws.Sall) {
      T = ws.Tall + pnl * nb2;
    } else {
      // S = Y Y^T as a product: a 128 x 128 x Kc GEMM runs on ONE workgroup (~2 ms per panel at n = 20000)
      EIG_HIP(launch_dgemm('N', 'T', kp, kp, Kc, 1.0, Y, n, Y, n, 0.0, ws.S, kp, false, false, s));
      hipLaunchKernelGGL(bt_tfactor_kernel, dim3(1), dim3(256), 0, s, ws.S, 0L, (int)kp, ws.tau + j0, kp, ws.T, 0L);
    }
    // P = ZT[:, c0:] * Y^T  (n x kp)
    EIG_HIP(launch_dgemm('N', 'T', n, kp, Kc, 1.0, ZT + c0, n, Y, n, 0.0, ws.P, kp, false, false, s));
    // P2 = P * T^T
    EIG_HIP(launch_dgemm('N', 'T', n, kp, kp, 1.0, ws.P, kp, T, kp, 0.0, ws.P2, kp, false, false, s));
    // ZT[:, c0:] -= P2 * Y
    EIG_HIP(launch_dgemm('N', 'N', n, Kc, kp, -1.0, ws.P2, kp, Y, n, 1.0, ZT + c0, n, false, false, s));
  }
  return 0;
}

} // namespace gemma_hip
#include "eigh2.hip.h"
namespace gemma_hip {

// GEMMA_HIP_EIGH_STAGES: 1 = one-stage tridiagonalisation (HBM-bound SYMV per column), 2 = two-stage (dense -> band ->
// tridiagonal, eigh2.hip.h) whenever the matrix has at least one stage-1 panel; unset: two-stage from n = 8000 (rounds 2-3: 14000; measured then:
// n = 8192 0.50 s one-stage / 0.70 s two-stage, n = 20000 4.46 / 3.09 s -- the n sequential panel launches and the 2 n
// dependent chase steps are latency, the SYMV they replace is bandwidth: a n + b n^3 fits through the two sizes cross at 14000).  Odd n stays on the one-stage path (the panel
// GEMMs want even leading dimensions).
static inline bool eig_two_stage(long n) {
  const char *e = getenv("GEMMA_HIP_EIGH_STAGES");
  if (e && e[0] == '1') return false;
  if (n < 3 * E2_B || (n & 1) || (n - 2 + E2_NB - 1) / E2_NB > Q2_MAXJ) return false; // q2_apply_kernel's LDS table: n <= 65 536
  if (e && e[0] == '2') return true;
  return n >= 8000; // round 4: with the panel in one launch and the pipelined chase n = 8192 takes 0.48 s two-stage, 0.52 s one-stage
}

// The order the core runs at: an odd n >= 192 is embedded in n + 1 (eigh_device) unless GEMMA_HIP_EIGH_PAD=0.
static inline long eig_effective_n(long n) {
  const char *ep = getenv("GEMMA_HIP_EIGH_PAD");
  return ((n & 1) == 0 || n < 192 || (ep && ep[0] == '0')) ? n : n + 1;
}
// a rank of a collective solve that cannot even start -- it failed to allocate its copy of the matrix, its padded copy,
// its slot of the kept (U, eval) -- must not leave the others waiting in the core's first all-reduce.  It takes part in exactly
// that agreement (eigh_device_core: anyone_failed(!ok), one all-reduce of one double) with "failed", so that every rank returns an
// error.  n = the caller's order; no-op where the solve at that order exchanges nothing (one-stage: replicas).
static inline void eigh_collective_abort(long n, hipStream_t s, const EighShard *sh) {
  if (!(sh && sh->world > 1 && sh->bcast && sh->allreduce_sum) || !eig_two_stage(eig_effective_n(n))) return;
  double *d = nullptr;
  if (hipMalloc(reinterpret_cast<void **>(&d), 16) != hipSuccess) {
    (void)hipGetLastError();
    return; // cannot take part (16 bytes refused): the others' collective times out -- the end anyway
  }
  const double v = 1.0;
  if (hipMemcpyAsync(d, &v, 8, hipMemcpyHostToDevice, s) == hipSuccess) (void)sh->allreduce_sum(sh->ctx, d, 1, s);
  (void)hipStreamSynchronize(s);
  (void)hipFree(d);
}

// Several ranks, one decomposition (round 4).  The eigenvectors are independent through both back-transformations
// (a row of Z^T never meets another row), and those are 0.8 of 2.3 s at n = 20 000 and 10 of 19 s at n = 50 000.  Every rank
// runs the reduction and the divide & conquer on its own copy of the matrix -- the same code on the same bits: the results
// agree bit for bit, which is CHECKED (a hash of the tridiagonal matrix and of the eigenvalues is compared with rank 0's; on
// any difference rank 0 alone finishes and broadcasts U) --, applies Q2 and Q1 to its own slice of Z^T (whole 64-row blocks),
// and the slices travel once (one broadcast per rank: an all-gather on the library's two collectives).  One-stage solves
// (n < 8 000) are replicated whole: nothing to send.
static double g_eig_last[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // stage seconds of the last solve with GEMMA_HIP_EIGH_TIMING=1
static inline unsigned long long eig_fnv(unsigned long long h, const void *p, size_t bytes) {
  const unsigned char *c = static_cast<const unsigned char *>(p);
  for (size_t i = 0; i < bytes; ++i) {
    h ^= c[i];
    h *= 1099511628211ULL;
  }
  return h;
}

// every buffer a solve of order n takes (eigh_device_core; gemma_hip_eigh_reserve runs this alone and hands the blocks to the pool)
static inline bool eig_alloc_all(long n, EigWs &ws, Eig2Ws &w2) {
  const size_t nn = (size_t)n * n;
  bool ok = ws.get(ws.VT, nn) && ws.get(ws.WT, (size_t)EIG_NB * n) && ws.get(ws.xcol, n + 2) && ws.get(ws.p, n) &&
            ws.get(ws.ab, 2 * EIG_NB) && ws.get(ws.ssbuf, n / TD_ROWS + 2) && ws.get(ws.dotbuf, n / TD_ROWS + 2) &&
            ws.get(ws.wtmp, n) && ws.get(ws.d, n) && ws.get(ws.e, n) && ws.get(ws.tau, n) &&
            ws.get(ws.Delta, nn) && ws.get(ws.Wk, nn) && ws.get(ws.P, (size_t)n * EIG_NB) &&
            ws.get(ws.P2, (size_t)n * EIG_NB) && ws.get(ws.S, (size_t)EIG_NB * EIG_NB) &&
            ws.get(ws.T, (size_t)EIG_NB * EIG_NB) && ws.get(ws.zbuf, n) && ws.get(ws.dl, n) && ws.get(ws.w, n) &&
            ws.get(ws.lam, n) && ws.get(ws.zhat, n) && ws.get(ws.dphys, n) && ws.get(ws.ibuf, 2 * (size_t)n + 64) && ws.get(ws.ibuf2, 4 * (size_t)n + 64) &&
            ws.get(ws.info, 1) && ws.get(ws.rot, n);
  {
    // panel Gram matrices from the tridiagonalisation; GEMMA_HIP_EIGH_PANEL_S=0 recomputes them as GEMMs
    const char *e = getenv("GEMMA_HIP_EIGH_PANEL_S");
    if (ok && !(e && e[0] == '0'))
      ok = ws.get(ws.Sall, (size_t)((n + EIG_NB - 1) / EIG_NB) * EIG_NB * EIG_NB) &&
           ws.get(ws.Tall, (size_t)((n + EIG_NB - 1) / EIG_NB) * EIG_NB * EIG_NB);
  }
  {
    // symmetric (lower-triangle) SYMV partials; GEMMA_HIP_EIGH_SYMV=0 keeps the row-per-wave form
    const char *e = getenv("GEMMA_HIP_EIGH_SYMV");
    if (ok && (n & 1) == 0 && !(e && e[0] == '0')) {
      const size_t nseg = (size_t)(n + TS_SEG_MIN - 1) / TS_SEG_MIN, nstrip = (size_t)(n + TS_STRIP - 1) / TS_STRIP;
      ok = ws.get(ws.rowP, nseg * (size_t)n) && ws.get(ws.colP, nstrip * (size_t)n);
    }
  }
  if (ok && eig_two_stage(n)) {
    if (!ws.Tall) ok = ws.get(ws.Tall, (size_t)((n + EIG_NB - 1) / EIG_NB) * EIG_NB * EIG_NB);
    ok = ok && eig2_alloc(n, ws, w2);
  }
  return ok;
}

// G (n x n symmetric, device, destroyed) -> U (row-major, eigenvector k in column k), eval ascending.
static inline int eigh_device_core(double *G, long n, double *U, double *eval, hipStream_t s, std::string &msg,
                                   const EighShard *sh = nullptr) {
  EigWs ws;
  ws.n = n;
  const double t_enter = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  const size_t nn = (size_t)n * n;
  Eig2Ws w2;
  const bool two = eig_two_stage(n);
  bool ok = eig_alloc_all(n, ws, w2);
  // Collective runs: a rank that fails on its own (allocation, a non-finite entry, a leaf that does not converge) must not leave
  // the others waiting inside a collective it never reaches -- before the first exchange the ranks agree, with one all-reduce of
  // a status word in a buffer of its own, that every one of them got that far; if any did not, all of them return an error.
  const bool coll = two && sh && sh->world > 1 && sh->bcast && sh->allreduce_sum;
  double *agree_d = nullptr;
  if (coll && hipMalloc(reinterpret_cast<void **>(&agree_d), 16) != hipSuccess) {
    (void)hipGetLastError();
    agree_d = nullptr; // the agreement itself then fails on this rank: it reports 'bad' through the host value below
  }
  auto anyone_failed = [&](bool mine) -> bool {
    if (!coll) return mine;
    double v = mine ? 1.0 : 0.0;
    if (!agree_d) return true; // cannot take part: the others time out in the collective -- an allocation of 16 bytes failing is the end anyway
    if (hipMemcpyAsync(agree_d, &v, 8, hipMemcpyHostToDevice, s) != hipSuccess || sh->allreduce_sum(sh->ctx, agree_d, 1, s) ||
        hipMemcpyAsync(&v, agree_d, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
      return true;
    return v != 0.0;
  };
  if (anyone_failed(!ok)) {
    ws.release();
    if (agree_d) (void)hipFree(agree_d);
    msg = ok ? "another rank could not allocate its eigensolver workspace" : "cannot allocate the eigensolver workspace (about 5 n^2 doubles)";
    return 3;
  }
  int rc = 0;
  bool reached_agreement = false;
  bool fallback_pending = false; // root-only fall-back: the agreement before its broadcasts is still owed by this rank
  std::vector<double> hd(n), he(std::max<long>(n - 1, 1)), dphys;
  double *Z = nullptr;
  const char *tenv = getenv("GEMMA_HIP_EIGH_TIMING");
  const bool timing = tenv && tenv[0] == '1';
  auto now = [&]() {
    (void)hipStreamSynchronize(s);
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  };
  double t0 = timing ? now() : 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0, t1a = 0.0, t2a = 0.0;
  do {
    if (n == 1) {
      hipError_t e1 = hipMemcpyAsync(eval, G, 8, hipMemcpyDeviceToDevice, s);
      const double one = 1.0;
      hipError_t e2 = hipMemcpyAsync(U, &one, 8, hipMemcpyHostToDevice, s);
      hipError_t e3 = hipStreamSynchronize(s);
      if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) { msg = "copy failed"; rc = 4; }
      break;
    }
    if (two) {
      rc = eig2_sy2sb(G, n, ws, w2, s, msg);
      if (timing) t1a = now();
      if (!rc) rc = eig2_sb2st(n, ws, w2, s, msg);
    } else {
      rc = eig_tridiagonalize(G, n, ws, s, msg);
    }
    if (rc) break;
    if (hipMemcpyAsync(hd.data(), ws.d, n * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(he.data(), ws.e, (n - 1) * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
      msg = std::string("tridiagonalisation failed: ") + hipGetErrorString(hipGetLastError());
      rc = 4;
      break;
    }
    double tnorm = 0.0;
    bool finite = true;
    for (long i = 0; i < n; ++i) {
      tnorm = std::max(tnorm, std::fabs(hd[i]));
      finite = finite && std::isfinite(hd[i]);
    }
    for (long i = 0; i + 1 < n; ++i) {
      tnorm = std::max(tnorm, std::fabs(he[i]));
      finite = finite && std::isfinite(he[i]);
    }
    if (!finite) {
      msg = "matrix contains NaN/Inf";
      rc = 1;
      break;
    }
    if (timing) t1 = now();
    const double scale = (tnorm > 0.0) ? tnorm : 1.0;
    for (long i = 0; i < n; ++i) hd[i] /= scale;
    for (long i = 0; i + 1 < n; ++i) he[i] /= scale;
    // G (dead after the reduction) and U serve as the two eigenvector-row buffers
    rc = eig_stedc(n, hd, he, G, U, ws, s, &Z, dphys, msg);
    if (coll) reached_agreement = true;
    if (anyone_failed(rc != 0)) {
      if (!rc) { msg = "another rank failed in the reduction or the divide & conquer"; rc = 4; }
      break;
    }
    if (timing) t2 = now();
    long row0 = 0, rows = n;
    bool sharded = false, root_only = false;
    if (two && sh && sh->world > 1 && sh->bcast && sh->allreduce_sum) {
      // do all ranks hold the same tridiagonal matrix and the same eigenvalues, bit for bit?
      unsigned long long hsh = 1469598103934665603ULL;
      hsh = eig_fnv(hsh, hd.data(), hd.size() * 8);
      hsh = eig_fnv(hsh, he.data(), he.size() * 8);
      hsh = eig_fnv(hsh, dphys.data(), dphys.size() * 8);
      double hv[2] = {(double)(hsh & 0xffffffffULL), (double)(hsh >> 32)}, h0[2] = {0.0, 0.0};
      if (hipMemcpyAsync(ws.zbuf, hv, 16, hipMemcpyHostToDevice, s) != hipSuccess || sh->bcast(sh->ctx, ws.zbuf, 16, 0, s) ||
          hipMemcpyAsync(h0, ws.zbuf, 16, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        msg = "sharded back-transformation: agreement broadcast failed";
        rc = 4;
        break;
      }
      double differ = (h0[0] != hv[0] || h0[1] != hv[1]) ? 1.0 : 0.0;
      if (hipMemcpyAsync(ws.zbuf, &differ, 8, hipMemcpyHostToDevice, s) != hipSuccess || sh->allreduce_sum(sh->ctx, ws.zbuf, 1, s) ||
          hipMemcpyAsync(&differ, ws.zbuf, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        msg = "sharded back-transformation: agreement all-reduce failed";
        rc = 4;
        break;
      }
      const char *efd = getenv("GEMMA_HIP_EIGH_SHARD_FORCE_DIFFER"); // tests: take the fall-back branch
      if (efd && efd[0] == '1') differ = 1.0;
      if (differ == 0.0) {
        const long nrb = (n + 63) / 64;
        const long b0 = nrb * sh->rank / sh->world, b1 = nrb * (sh->rank + 1) / sh->world;
        row0 = 64 * b0;
        rows = std::min<long>(n, 64 * b1) - row0;
        sharded = true;
      } else {
        root_only = true; // rank 0 finishes alone and broadcasts (U, eval)
        fallback_pending = true;
      }
    }
    if (two) {
      if (!(root_only && sh->rank != 0) && rows > 0) {
        rc = eig2_apply_q2(Z + row0 * n, n, rows, w2, s, msg);
        if (timing) t2a = now();
        if (!rc) rc = eig2_apply_q1(Z + row0 * n, n, rows, ws, w2, s, msg);
      }
    } else {
      rc = eig_backtransform(Z, n, ws, s, msg);
    }
    if (root_only && sh->rank == 0) {
      const char *eff = getenv("GEMMA_HIP_EIGH_FAIL_FALLBACK"); // tests: rank 0 fails in the fall-back it runs alone
      if (eff && eff[0] == '1') { msg = "failure injected into the root-only fall-back (GEMMA_HIP_EIGH_FAIL_FALLBACK)"; rc = 4; }
    }
    if (sharded) {
      // every rank tells the others how its slice went before anybody waits for it
      double bad = rc ? 1.0 : 0.0;
      if (hipMemcpyAsync(ws.zbuf, &bad, 8, hipMemcpyHostToDevice, s) != hipSuccess || sh->allreduce_sum(sh->ctx, ws.zbuf, 1, s) ||
          hipMemcpyAsync(&bad, ws.zbuf, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        if (!rc) { msg = "sharded back-transformation: status all-reduce failed"; rc = 4; }
        break;
      }
      if (bad != 0.0) {
        if (!rc) { msg = "sharded back-transformation: another rank failed"; rc = 4; }
        break;
      }
      const long nrb = (n + 63) / 64;
      for (int r = 0; r < sh->world && !rc; ++r) {
        const long b0 = nrb * r / sh->world, b1 = nrb * (r + 1) / sh->world;
        const long r0 = 64 * b0, rr = std::min<long>(n, 64 * b1) - r0;
        if (rr > 0 && sh->bcast(sh->ctx, Z + r0 * n, (size_t)rr * n * 8, r, s)) {
          msg = "sharded back-transformation: slice broadcast failed";
          rc = 4;
        }
      }
    }
    if (rc) break;
    if (timing) t3 = now();
    if (timing && sharded)
      fprintf(stderr, "gemma_hip_eigh n=%ld: rank %d of %d back-transformed rows %ld .. %ld of Z^T\n", n, sh->rank, sh->world, row0,
              row0 + rows);
    if (root_only && sh->rank != 0) {
      // rank 0's result arrives as it is (U, then eval) -- once rank 0 has said that it HAS one (a rank 0 that failed in its
      // back-transformation left the loop before its broadcasts and the others waited for ever)
      fallback_pending = false;
      if (anyone_failed(false)) {
        msg = "sharded back-transformation: rank 0 failed in the fall-back";
        rc = 4;
        break;
      }
      if (sh->bcast(sh->ctx, U, nn * 8, 0, s) || sh->bcast(sh->ctx, eval, (size_t)n * 8, 0, s) || hipStreamSynchronize(s) != hipSuccess) {
        msg = "sharded back-transformation: fall-back broadcast failed";
        rc = 4;
      }
      break;
    }
    std::vector<int> perm(n);
    for (long i = 0; i < n; ++i) perm[i] = (int)i;
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int c) { return dphys[a] < dphys[c]; });
    double *Zfinal = Z;
    if (Z == U) { // the transpose cannot run in place: park Z^T in the (free) Delta buffer
      if (hipMemcpyAsync(ws.Delta, Z, nn * 8, hipMemcpyDeviceToDevice, s) != hipSuccess) { msg = "copy failed"; rc = 4; break; }
      Zfinal = ws.Delta;
    }
    if (hipMemcpyAsync(ws.ibuf, perm.data(), n * sizeof(int), hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(ws.dphys, dphys.data(), n * 8, hipMemcpyHostToDevice, s) != hipSuccess) {
      msg = "copy failed";
      rc = 4;
      break;
    }
    const unsigned nb32 = (unsigned)((n + 31) / 32);
    hipLaunchKernelGGL(bt_transpose_perm_kernel, dim3(nb32, nb32), dim3(32, 8), 0, s, Zfinal, n, ws.ibuf, ws.dphys,
                       scale, U, eval);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
      msg = "final transpose failed";
      rc = 4;
    }
    if (root_only) {
      fallback_pending = false;
      if (anyone_failed(rc != 0)) {
        if (!rc) { msg = "sharded back-transformation: fall-back agreement failed"; rc = 4; }
        break;
      }
    }
    if (!rc && root_only && (sh->bcast(sh->ctx, U, nn * 8, 0, s) || sh->bcast(sh->ctx, eval, (size_t)n * 8, 0, s) ||
                             hipStreamSynchronize(s) != hipSuccess)) {
      msg = "sharded back-transformation: fall-back broadcast failed";
      rc = 4;
    }
  } while (0);
  if (coll && !reached_agreement) (void)anyone_failed(true); // left the loop before the agreement: tell the others
  if (coll && fallback_pending) (void)anyone_failed(true);   // rank 0 left the fall-back before its broadcasts: the others are told
  if (agree_d) (void)hipFree(agree_d);
  (void)hipStreamSynchronize(s);
  if (timing && rc == 0 && n > 1) {
    // kept for gemma_hip_dbg_eigh_last: {reduction to band / tridiagonal, bulge chase, divide & conquer, Q2, Q1 (one-stage: the
    // whole back-transformation), sort + transpose}
    const double tend = now();
    g_eig_last[0] = two ? t1a - t0 : t1 - t0; g_eig_last[1] = two ? t1 - t1a : 0.0; g_eig_last[2] = t2 - t1;
    g_eig_last[3] = two ? t2a - t2 : 0.0; g_eig_last[4] = two ? t3 - t2a : t3 - t2; g_eig_last[5] = tend - t3;
    g_eig_last[6] = (double)n; g_eig_last[7] = two ? 2.0 : 1.0;
    if (two)
      fprintf(stderr, "gemma_hip_eigh n=%ld (two-stage): dense->band %.3f s, band->tridiagonal %.3f s, divide&conquer %.3f s, "
                      "back-transform Q2 %.3f s, Q1 %.3f s, sort+transpose %.3f s\n", n, t1a - t0, t1 - t1a, t2 - t1,
              t2a - t2, t3 - t2a, now() - t3);
    else
      fprintf(stderr, "gemma_hip_eigh n=%ld: tridiagonalisation %.3f s, divide&conquer %.3f s, back-transform %.3f s, "
                      "sort+transpose %.3f s\n", n, t1 - t0, t2 - t1, t3 - t2, now() - t3);
  }
  const double t_rel = timing ? now() : 0.0;
  ws.release();
  if (timing && rc == 0 && n > 1)
    fprintf(stderr, "gemma_hip_eigh n=%ld: workspace allocation %.3f s, release %.3f s\n", n, t0 - t_enter, now() - t_rel);
  return rc;
}

// Odd n: every GEMM of the solver would leave the aligned fast path (odd leading dimension), the symmetric SYMV and the
// two-stage reduction need even n.  The matrix is embedded in an (n+1) x (n+1) one whose extra row / column is zero except
// for the diagonal entry sigma = trace / n: the extra eigenpair (sigma, e_{n+1}) decouples exactly (no reflector ever touches
// a zero column, the divide & conquer splits at the zero off-diagonal), is recognised by its eigenvector and dropped.
__global__ void eig_pad_kernel(const double *__restrict__ G, long n, double *__restrict__ Gp, double sigma) {
  const long i = blockIdx.x;
  for (long j = threadIdx.x; j <= n; j += blockDim.x)
    Gp[i * (n + 1) + j] = (i < n && j < n) ? G[i * n + j] : ((i == n && j == n) ? sigma : 0.0);
}
__global__ void eig_unpad_kernel(const double *__restrict__ Up, const double *__restrict__ evp, long n, long kdrop,
                                 double *__restrict__ U, double *__restrict__ eval) {
  const long i = blockIdx.x;
  for (long k = threadIdx.x; k < n; k += blockDim.x) {
    const long ks = k + (k >= kdrop ? 1 : 0);
    U[i * n + k] = Up[i * (n + 1) + ks];
    if (i == 0) eval[k] = evp[ks];
  }
}
static inline int eigh_device(double *G, long n, double *U, double *eval, hipStream_t s, std::string &msg,
                              const EighShard *sh = nullptr) {
  const char *ep = getenv("GEMMA_HIP_EIGH_PAD"); // 0: odd n on the unaligned paths (tests)
  if ((n & 1) == 0 || n < 192 || (ep && ep[0] == '0')) return eigh_device_core(G, n, U, eval, s, msg, sh);
  const long m = n + 1;
  double *Gp = nullptr, *Up = nullptr, *evp = nullptr;
  auto cleanup = [&]() {
    if (Gp) (void)hipFree(Gp);
    if (Up) (void)hipFree(Up);
    if (evp) (void)hipFree(evp);
  };
  if (hipMalloc(reinterpret_cast<void **>(&Gp), (size_t)m * m * 8) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&Up), (size_t)m * m * 8) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&evp), (size_t)m * 8) != hipSuccess) {
    (void)hipGetLastError();
    cleanup();
    if (sh && sh->world > 1 && eig_two_stage(m)) {
      // a collective solve: the other ranks run the padded, two-stage, exchanging form -- this rank cannot quietly take the
      // replicated odd-n path instead.  It reports the failure through the core's first agreement.
      eigh_collective_abort(n, s, sh);
      msg = "cannot allocate the padded copy of an odd-order matrix in a collective solve";
      return 3;
    }
    return eigh_device_core(G, n, U, eval, s, msg, sh); // not enough room for the padded copy: the slower unaligned path
  }
  std::vector<double> hdiag((size_t)n);
  if (hipMemcpy2DAsync(hdiag.data(), 8, G, (size_t)(n + 1) * 8, 8, (size_t)n, hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess) {
    cleanup();
    msg = "copy failed";
    return 4;
  }
  double sigma = 0.0;
  for (long i = 0; i < n; ++i) sigma += hdiag[i];
  sigma /= (double)n;
  hipLaunchKernelGGL(eig_pad_kernel, dim3((unsigned)m), dim3(256), 0, s, G, n, Gp, sigma);
  int rc = eigh_device_core(Gp, m, Up, evp, s, msg, sh);
  if (rc) {
    cleanup();
    return rc;
  }
  std::vector<double> last((size_t)m);
  if (hipMemcpyAsync(last.data(), Up + (size_t)n * m, (size_t)m * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess) {
    cleanup();
    msg = "copy failed";
    return 4;
  }
  long kdrop = 0;
  for (long k = 1; k < m; ++k)
    if (std::fabs(last[k]) > std::fabs(last[kdrop])) kdrop = k;
  if (!(std::fabs(last[kdrop]) > 1.0 - 1e-9)) { // cannot happen for an exactly decoupled entry; never guess
    cleanup();
    return eigh_device_core(G, n, U, eval, s, msg, sh);
  }
  hipLaunchKernelGGL(eig_unpad_kernel, dim3((unsigned)n), dim3(256), 0, s, Up, evp, n, kdrop, U, eval);
  const bool ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
  cleanup();
  if (!ok) {
    msg = "un-padding failed";
    return 4;
  }
  return 0;
}

} // namespace gemma_hip
