// fp64 MFMA GEMM for gfx950 (MI355X): C(MxN) = alpha * op(A) * op(B) + beta * C, row-major.
//
// Replaces the cblas_dgemm behind fast_dgemm / fast_eigen_dgemm (GEMMA src/fastblas.cpp:175-209).
// Hot call sites: U^T X per SNP batch (src/lmm.cpp:1521,1847), K += Xb Xb^T (src/gemma_io.cpp:1554,
// 1711), CalcUtX (src/mathfunc.cpp:504-506).
//
// Design (CDNA4), details in DESIGN.md 3.1:
//  * v_mfma_f64_16x16x4_f64: one 16x16 output block per instruction, K = 4, 64 cycles/SIMD
//    (78.6 TFLOP/s chip peak).  A/B operands are ONE f64 per lane (A[i=l&15][k=l>>4],
//    B[k=l>>4][j=l&15]); the 16x16 f64 result is 4 f64 per lane: col = l&15, row = (l>>4)+4*r.
//  * 128x128 block tile, 256 threads = 4 wavefronts in a 2x2 grid, each wave owns 64x64
//    (4x4 MFMA blocks = 64 f64 accumulators per lane = 128 VGPR), BK = 16, 2 blocks/CU.
//  * Three interior kernels, selected by GEMMA_HIP_GEMM_PIPE:
//      dgemm_mfma_glds_kernel (default): global -> LDS by global_load_lds_dwordx4, pinned issue order
//        (one ds_read / LDS-DMA piece behind each MFMA), 218 ms at M = N = K = 20000 (73 TFLOP/s);
//      dgemm_mfma_pipe_kernel: register-staged, fragments software-pipelined one K-step ahead (238 ms);
//      dgemm_mfma_kernel<.., FULL>: register-staged, compiler-scheduled (243 ms); its FULL = false form is
//        the bounds-checked kernel for ragged strips (when beta != 0), K tails and unaligned views.
//  * blockIdx -> tile: XCD-aware (block b runs on XCD b % 8): every XCD gets a contiguous run of
//    tiles, rastered in groups of 4 tile-rows so the ~64 blocks an XCD runs at once form a
//    4x16 patch sharing A/B panels in that XCD's 4 MiB L2 (80 % TCC hit rate at n = 20000).
//  * SYRK mode (kinship): only tiles with tile_n >= tile_m are launched (triangular grid).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace gemma_hip {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_BK = 16;
constexpr int GEMM_LD_KM = 144; // [k][m] image: row stride in doubles
constexpr int GEMM_LD_MK = 18;  // [m][k] image: row stride in doubles
constexpr int GEMM_TILE_DOUBLES = 2304; // 16*144 == 128*18
constexpr int GEMM_THREADS = 256;

struct GemmArgs {
  const double *A;
  const double *B;
  double *C;
  long M, N, K;
  long lda, ldb, ldc;
  double alpha, beta;
  int tiles_m, tiles_n; // extent of the launched tile sub-grid
  int tm0, tn0;         // its origin in the full tile grid
  int syrk_upper;  // 1: launch only tiles with tn >= tm (sub-grid must start at (0,0))
  int square_a;    // 1: use A*A elementwise as the A operand (grid-lambda x^2 sums)
  int gm;          // tile rows per raster group (0 = default 8)
  int clamp;       // glds kernel: shift ragged last tiles back inside the matrix (needs beta == 0)
  int ablate;      // timing experiments only (GEMMA_HIP_GEMM_ABLATE): 1 = no global loads / LDS stores after
                   // the first K-tile, 2 = no barrier, 4 = fragments read once (results are then wrong)
  // K slices in ONE launch (round 4; launch_dgemm_ksliced): blockIdx.y = slice, slice ks multiplies K range
  // [ks * kslice, min(K, (ks + 1) * kslice)) into C + ks * cslice.  Skinny products (128 x m x m: m / 128 output tiles for 256
  // CUs) fill the chip this way without extra streams -- whose number of hardware queues the library does not control.
  int kslices = 1;
  long kslice = 0, cslice = 0;
  // round 5: every off-diagonal tile ALSO stores its transpose at C[col][row] (symmetric results formed on the upper-triangle tiles
  // only -- the eigensolver's rank-256 update of the trailing matrix: the separate mirror pass, a read and a write of half the
  // matrix per panel, goes away).  Lane (l15, l4) holds rows l4 + 4 r of column l15 of a 16 x 16 block: for one r the four lane
  // groups write four consecutive doubles of a transposed row, the four r together its whole 128-byte line.
  int mirror = 0;
};
template <bool A_KM, bool B_KN>
__device__ __forceinline__ void gemm_take_slice(GemmArgs &g) {
  if (g.kslices > 1) {
    const long ks = blockIdx.y, k0 = ks * g.kslice;
    const long kn = (g.K - k0 < g.kslice) ? g.K - k0 : g.kslice;
    g.A += A_KM ? k0 * g.lda : k0;
    g.B += B_KN ? k0 * g.ldb : k0;
    g.C += ks * g.cslice;
    g.K = kn;
  }
}

// 16-byte global load of two consecutive doubles with element-wise bounds; `vec_ok` says the
// address is 16-byte aligned and both elements are in range.
__device__ __forceinline__ f64x2 ld2(const double *p, bool ok0, bool ok1, bool vec_ok) {
  f64x2 v;
  if (vec_ok) {
    v = *reinterpret_cast<const f64x2 *>(p);
  } else {
    v.x = ok0 ? p[0] : 0.0;
    v.y = ok1 ? p[1] : 0.0;
  }
  return v;
}

// Operand tile loader.  KM = true: operand stored [k][m] (m contiguous, leading dim ld);
// KM = false: stored [m][k] (k contiguous).  Loads the BK x 128 tile at (k0, m0) into 4 f64x2.
template <bool KM, int NT>
__device__ __forceinline__ void load_tile(const double *__restrict__ P, long ld, long m0, long k0,
                                          long Mdim, long Kdim, bool aligned, int t, f64x2 *r,
                                          bool full) {
  constexpr int NLD = 1024 / NT;  // 16-byte loads per thread per operand tile
  constexpr int KSTEP = NT / 64;  // [k][m] image: k rows covered per pass
  constexpr int MSTEP = NT / 8;   // [m][k] image: m rows covered per pass
  if (full) { // interior tile, aligned operand: no predicates at all (block-uniform branch)
    if (KM) {
      const double *p = P + (k0 + (t >> 6)) * ld + m0 + 2 * (t & 63);
#pragma unroll
      for (int j = 0; j < NLD; ++j) r[j] = *reinterpret_cast<const f64x2 *>(p + (long)(KSTEP * j) * ld);
    } else {
      const double *p = P + (m0 + (t >> 3)) * ld + k0 + 2 * (t & 7);
#pragma unroll
      for (int j = 0; j < NLD; ++j) r[j] = *reinterpret_cast<const f64x2 *>(p + (long)(MSTEP * j) * ld);
    }
    return;
  }
  if (KM) {
    const int mm = 2 * (t & 63);
    const int kb = t >> 6;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const long k = k0 + kb + KSTEP * j;
      const long m = m0 + mm;
      const bool kin = k < Kdim;
      const bool ok0 = kin && (m < Mdim), ok1 = kin && (m + 1 < Mdim);
      const double *p = P + k * ld + m;
      if (ok0 || ok1)
        r[j] = ld2(p, ok0, ok1, aligned && ok1);
      else
        r[j] = f64x2{0.0, 0.0};
    }
  } else {
    const int kk = 2 * (t & 7);
    const int mb = t >> 3;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const long m = m0 + mb + MSTEP * j;
      const long k = k0 + kk;
      const bool min_ = m < Mdim;
      const bool ok0 = min_ && (k < Kdim), ok1 = min_ && (k + 1 < Kdim);
      const double *p = P + m * ld + k;
      if (ok0 || ok1)
        r[j] = ld2(p, ok0, ok1, aligned && ok1);
      else
        r[j] = f64x2{0.0, 0.0};
    }
  }
}

template <bool KM, int NT>
__device__ __forceinline__ void store_tile(double *__restrict__ S, int t, const f64x2 *r) {
  constexpr int NLD = 1024 / NT;
  if (KM) {
    const int mm = 2 * (t & 63);
    const int kb = t >> 6;
#pragma unroll
    for (int j = 0; j < NLD; ++j)
      *reinterpret_cast<f64x2 *>(S + (kb + (NT / 64) * j) * GEMM_LD_KM + mm) = r[j];
  } else {
    const int kk = 2 * (t & 7);
    const int mb = t >> 3;
#pragma unroll
    for (int j = 0; j < NLD; ++j)
      *reinterpret_cast<f64x2 *>(S + (mb + (NT / 8) * j) * GEMM_LD_MK + kk) = r[j];
  }
}

template <bool KM>
__device__ __forceinline__ double frag(const double *__restrict__ S, int m, int k) {
  return KM ? S[k * GEMM_LD_KM + m] : S[m * GEMM_LD_MK + k];
}

// block id -> (tm, tn)
__device__ __forceinline__ void tile_of_block(const GemmArgs &g, int &tm, int &tn) {
  const int nwg = gridDim.x;
  const int b = blockIdx.x;
  // bijective XCD remap: XCD x (= b % 8) gets a contiguous logical range
  const int q = nwg >> 3, r = nwg & 7, x = b & 7, o = b >> 3;
  const int L = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
  if (g.syrk_upper) {
    // logical index L over the upper triangle, row-major: row tm has (T - tm) tiles
    const int T = g.tiles_m;
    const double Td = (double)(2 * T + 1);
    int i = (int)((Td - sqrt(Td * Td - 8.0 * (double)L)) * 0.5);
    if (i < 0) i = 0;
    if (i > T - 1) i = T - 1;
    // fix up float error
    while (i > 0 && (long)i * (2 * T - i + 1) / 2 > L) --i;
    while ((long)(i + 1) * (2 * T - i) / 2 <= L) ++i;
    tm = i;
    tn = i + (L - (int)((long)i * (2 * T - i + 1) / 2));
  } else {
    const int GM = g.gm > 0 ? g.gm : 4; // 4 rows x 16 columns in flight per XCD: best L2 hit rate measured
    const int per_group = GM * g.tiles_n;
    const int grp = L / per_group;
    const int first_m = grp * GM;
    const int gsz = min(g.tiles_m - first_m, GM);
    const int in = L - grp * per_group;
    tm = first_m + in % gsz;
    tn = in / gsz;
  }
  tm += g.tm0;
  tn += g.tn0;
}

// FULL = true: every launched tile is a complete 128x128 tile of 16-byte aligned operands and K is a
// multiple of 16 -- no predicate anywhere (the hot instantiation).  FULL = false: ragged tiles / K tail /
// unaligned views, element-wise bounds on every access.
template <bool A_KM, bool B_KN, int NW, bool FULL>
__global__ __launch_bounds__(NW * 64, (NW == 8 ? 4 : 2)) void dgemm_mfma_kernel(GemmArgs g) {
  gemm_take_slice<A_KM, B_KN>(g);
  constexpr int NT = NW * 64;
  constexpr int NLD = 1024 / NT;
  constexpr int WN = NW / 2;        // waves along N (2 along M)
  constexpr int TN = 4 / (WN / 2);  // 16-wide MFMA blocks per wave along N: 4 (64 cols) or 2 (32 cols)
  __shared__ __attribute__((aligned(16))) double lds[4 * GEMM_TILE_DOUBLES];
  double *As0 = lds, *As1 = lds + GEMM_TILE_DOUBLES;
  double *Bs0 = lds + 2 * GEMM_TILE_DOUBLES, *Bs1 = lds + 3 * GEMM_TILE_DOUBLES;

  int tm, tn;
  tile_of_block(g, tm, tn);
  const long m0 = (long)tm * GEMM_BM, n0 = (long)tn * GEMM_BN;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int wcol = wn * (16 * TN);

  const bool a_al = FULL || (((reinterpret_cast<uintptr_t>(g.A) & 15) == 0) && ((g.lda & 1) == 0));
  const bool b_al = FULL || (((reinterpret_cast<uintptr_t>(g.B) & 15) == 0) && ((g.ldb & 1) == 0));

  f64x4 acc[4][TN];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};

  f64x2 ra[NLD], rb[NLD];
  const long nk = (g.K + GEMM_BK - 1) / GEMM_BK;
  const long nk_full = g.K / GEMM_BK; // K-tiles without a ragged tail
  const bool a_full = FULL || (a_al && (m0 + GEMM_BM <= g.M));
  const bool b_full = FULL || (b_al && (n0 + GEMM_BN <= g.N));

  load_tile<A_KM, NT>(g.A, g.lda, m0, 0, g.M, g.K, a_al, t, ra, a_full && nk_full > 0);
  load_tile<B_KN, NT>(g.B, g.ldb, n0, 0, g.N, g.K, b_al, t, rb, b_full && nk_full > 0);
  if (g.square_a) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) ra[j] = ra[j] * ra[j];
  }
  store_tile<A_KM, NT>(As0, t, ra);
  store_tile<B_KN, NT>(Bs0, t, rb);
  __syncthreads();

  for (long kt = 0; kt < nk; ++kt) {
    const double *As = (kt & 1) ? As1 : As0;
    const double *Bs = (kt & 1) ? Bs1 : Bs0;
    const bool more = ((kt + 1) < nk) && !(g.ablate & 1);
    if (more) {
      const bool kfull = FULL || ((kt + 1) < nk_full);
      load_tile<A_KM, NT>(g.A, g.lda, m0, (kt + 1) * GEMM_BK, g.M, g.K, a_al, t, ra, a_full && kfull);
      load_tile<B_KN, NT>(g.B, g.ldb, n0, (kt + 1) * GEMM_BK, g.N, g.K, b_al, t, rb, b_full && kfull);
    }
#pragma unroll
    for (int kk = 0; kk < GEMM_BK / 4; ++kk) {
      double a[4], b[TN];
      const int k = kk * 4 + l4;
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = frag<A_KM>(As, wm * 64 + i * 16 + l15, k);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = frag<B_KN>(Bs, wcol + j * 16 + l15, k);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
      if (kk == GEMM_BK / 4 - 2 && more) {
        // the next K-tile (issued before this tile's MFMAs) goes to the other LDS buffer while the last
        // 2 x 16 MFMAs of this tile are still in flight: only the barrier is left at the tile boundary
        if (g.square_a) {
#pragma unroll
          for (int j = 0; j < NLD; ++j) ra[j] = ra[j] * ra[j];
        }
        store_tile<A_KM, NT>((kt & 1) ? As0 : As1, t, ra);
        store_tile<B_KN, NT>((kt & 1) ? Bs0 : Bs1, t, rb);
      }
    }
    if (!(g.ablate & 2)) __syncthreads();
  }

  // epilogue: lane holds rows (l>>4)+4r, col l&15 of each 16x16 block
  const double alpha = g.alpha, beta = g.beta;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const long col = n0 + wcol + j * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = m0 + wm * 64 + i * 16 + l4 + 4 * r;
        if (FULL || (row < g.M && col < g.N)) {
          double *c = g.C + row * g.ldc + col;
          double v = alpha * acc[i][j][r];
          if (beta != 0.0) v += beta * (*c);
          *c = v;
          if (g.mirror && tm != tn) g.C[col * g.ldc + row] = v;
        }
      }
    }
  }
}

// Software-pipelined form of the FULL 4-wave kernel (complete, aligned tiles only).
//
// Why: with 2 blocks per CU the two wavefronts sharing a SIMD arbitrate fairly for the MFMA pipe, so the one that is
// behind runs alone while the leader waits on LDS and the pair converges to lock-step -- after that both wait on their
// ds_reads at the same time and the pipe drains once per K-step (measured: 11 % of the MFMA cycles idle with global
// loads and the barrier ablated away).  Here a wave never waits on LDS: the fragments of K-step s+1 (8 ds_read_b64,
// two register sets) are issued before the 16 MFMAs of step s, the staged tile t+1 is written to the other LDS buffer
// in the middle of step 2, the barrier sits between steps 2 and 3, and step 3 already reads tile t+1's first
// fragments.  The global loads of tile t+2 are issued right after that barrier, 3.5 K-steps (>= 3600 cycles of this
// wave's own MFMA issue) ahead of the ds_write that consumes them.
#define GEMMA_SB() __builtin_amdgcn_sched_barrier(0)
template <bool A_KM, bool B_KN>
__global__ __launch_bounds__(256, 2) void dgemm_mfma_pipe_kernel(GemmArgs g) {
  gemm_take_slice<A_KM, B_KN>(g);
  constexpr int NT = 256;
  constexpr int NLD = 4;
  __shared__ __attribute__((aligned(16))) double lds[4 * GEMM_TILE_DOUBLES];
  double *const As0 = lds, *const As1 = lds + GEMM_TILE_DOUBLES;
  double *const Bs0 = lds + 2 * GEMM_TILE_DOUBLES, *const Bs1 = lds + 3 * GEMM_TILE_DOUBLES;

  int tm, tn;
  tile_of_block(g, tm, tn);
  const long m0 = (long)tm * GEMM_BM, n0 = (long)tn * GEMM_BN;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, l4 = lane >> 4;

  // per-lane fragment offsets (doubles) for K-step 0, block i = 0 / j = 0
  const int a_off = A_KM ? l4 * GEMM_LD_KM + wm * 64 + l15 : (wm * 64 + l15) * GEMM_LD_MK + l4;
  const int b_off = B_KN ? l4 * GEMM_LD_KM + wn * 64 + l15 : (wn * 64 + l15) * GEMM_LD_MK + l4;
  constexpr int a_i = A_KM ? 16 : 16 * GEMM_LD_MK, a_kk = A_KM ? 4 * GEMM_LD_KM : 4;
  constexpr int b_j = B_KN ? 16 : 16 * GEMM_LD_MK, b_kk = B_KN ? 4 * GEMM_LD_KM : 4;

  f64x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};

  f64x2 ra[NLD], rb[NLD];
  const long nk = g.K / GEMM_BK;
  const bool sq = g.square_a != 0;
  // timing experiments (GEMMA_HIP_GEMM_ABLATE, results wrong): 1 no global loads, 16 no LDS stores, 2 no barrier
  const bool ab_ld = (g.ablate & 1) != 0, ab_st = (g.ablate & 16) != 0, ab_bar = (g.ablate & 2) != 0;

#define GEMMA_GLOAD(KT)                                                                         \
  do {                                                                                          \
    load_tile<A_KM, NT>(g.A, g.lda, m0, (long)(KT) * GEMM_BK, g.M, g.K, true, t, ra, true);     \
    load_tile<B_KN, NT>(g.B, g.ldb, n0, (long)(KT) * GEMM_BK, g.N, g.K, true, t, rb, true);     \
  } while (0)
#define GEMMA_LSTORE(AD, BD)                                                                    \
  do {                                                                                          \
    if (sq) {                                                                                   \
      _Pragma("unroll") for (int j_ = 0; j_ < NLD; ++j_) ra[j_] = ra[j_] * ra[j_];              \
    }                                                                                           \
    store_tile<A_KM, NT>(AD, t, ra);                                                            \
    store_tile<B_KN, NT>(BD, t, rb);                                                            \
  } while (0)
#define GEMMA_FRAGS(AS, BS, KK, FA, FB)                                                         \
  do {                                                                                          \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) FA[i_] = (AS)[a_off + i_ * a_i + (KK) * a_kk]; \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) FB[j_] = (BS)[b_off + j_ * b_j + (KK) * b_kk]; \
  } while (0)
#define GEMMA_MFMA_ROWS(FA, FB, I0, I1)                                                         \
  do {                                                                                          \
    _Pragma("unroll") for (int i_ = (I0); i_ < (I1); ++i_)                                      \
      _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                          \
        acc[i_][j_] = __builtin_amdgcn_mfma_f64_16x16x4f64(FA[i_], FB[j_], acc[i_][j_], 0, 0, 0); \
  } while (0)

  GEMMA_GLOAD(0);
  GEMMA_LSTORE(As0, Bs0);
  __syncthreads();
  if (nk > 1 && !ab_ld) GEMMA_GLOAD(1);

  double xa[4], xb[4], ya[4], yb[4];
  GEMMA_FRAGS(As0, Bs0, 0, xa, xb);

  for (long kt = 0; kt < nk; ++kt) {
    const bool odd = (kt & 1) != 0;
    const double *Ac = odd ? As1 : As0, *Bc = odd ? Bs1 : Bs0;
    double *An = odd ? As0 : As1, *Bn = odd ? Bs0 : Bs1;
    const bool more = (kt + 1) < nk;
    // K-step 0
    GEMMA_FRAGS(Ac, Bc, 1, ya, yb);
    GEMMA_SB();
    GEMMA_MFMA_ROWS(xa, xb, 0, 4);
    GEMMA_SB();
    // K-step 1
    GEMMA_FRAGS(Ac, Bc, 2, xa, xb);
    GEMMA_SB();
    GEMMA_MFMA_ROWS(ya, yb, 0, 4);
    GEMMA_SB();
    // K-step 2: tile t+1 goes to the other LDS buffer behind the first 8 MFMAs; barrier behind the last 8
    GEMMA_FRAGS(Ac, Bc, 3, ya, yb);
    GEMMA_SB();
    GEMMA_MFMA_ROWS(xa, xb, 0, 2);
    GEMMA_SB();
    if (more && !ab_st) GEMMA_LSTORE(An, Bn);
    GEMMA_SB();
    GEMMA_MFMA_ROWS(xa, xb, 2, 4);
    GEMMA_SB();
    if (!ab_bar) __syncthreads();
    if ((kt + 2) < nk && !ab_ld) GEMMA_GLOAD(kt + 2);
    // K-step 3: first fragments of tile t+1 behind this tile's last 16 MFMAs
    if (more) GEMMA_FRAGS(An, Bn, 0, xa, xb);
    GEMMA_SB();
    GEMMA_MFMA_ROWS(ya, yb, 0, 4);
    GEMMA_SB();
  }
#undef GEMMA_GLOAD
#undef GEMMA_LSTORE
#undef GEMMA_FRAGS
#undef GEMMA_MFMA_ROWS

  const double alpha = g.alpha, beta = g.beta;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long col = n0 + wn * 64 + j * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = m0 + wm * 64 + i * 16 + l4 + 4 * r;
        double *c = g.C + row * g.ldc + col;
        double v = alpha * acc[i][j][r];
        if (beta != 0.0) v += beta * (*c);
        *c = v;
        if (g.mirror && tm != tn) g.C[col * g.ldc + row] = v;
      }
    }
  }
}

// Direct-to-LDS, issue-interleaved form of the interior kernel (complete, aligned tiles only).
//
// Issue is in order per wave, so every instruction that sits between two MFMAs of one wave delays the second MFMA
// once the run of such instructions outlasts the 64-cycle shadow of the first (measured: one wave per SIMD reaches
// only 88 % of the MFMA rate on a clustered schedule even with loads and barrier ablated away, and each removed
// group of loads / LDS stores buys 2-3 %).  So:
//  * operands go global -> LDS with global_load_lds_dwordx4 (1 KiB per wave instruction): no staging VGPRs, no
//    ds_write pass, 8 instructions per wave per K-tile;
//  * the instruction order is pinned (sched_barrier after every MFMA): at most one ds_read and one LDS-DMA issue
//    follow any MFMA.  MFMA slots of one K-tile (M0..M63):
//      M0-7   ds_read fragments of K-step 1     M16-23 K-step 2     M32-39 K-step 3
//      after M51: s_waitcnt vmcnt(0) (tile t+1 has landed, issued >= 56 MFMAs earlier) + barrier
//      M52-59 ds_read the first fragments of tile t+1, issue the 8 LDS-DMA pieces of tile t+2
//  * LDS images are lane-linear per wave instruction, as LDS-DMA requires:
//      [k][m] operands: one k-row (128 doubles = 1 KiB) per instruction, rows still 144 doubles apart;
//      [m][k] operands: 8 rows x 16 doubles per instruction, unpadded, with the 16-byte chunk index XOR-ed by
//      (m >> 1) & 7 on the SOURCE address and on the fragment read (conflict-free ds_read_b64 half-waves).
typedef const __attribute__((address_space(1))) void *gemma_gptr_t;
typedef __attribute__((address_space(3))) void *gemma_lptr_t;

template <bool A_KM, bool B_KN>
__global__ __launch_bounds__(256, 2) void dgemm_mfma_glds_kernel(GemmArgs g) {
  gemm_take_slice<A_KM, B_KN>(g);
  __shared__ __attribute__((aligned(1024))) double lds[4 * GEMM_TILE_DOUBLES];
  int tm, tn;
  tile_of_block(g, tm, tn);
  long m0 = (long)tm * GEMM_BM, n0 = (long)tn * GEMM_BN;
  if (g.clamp) { // ragged last tile row / column: shift the tile back inside (beta == 0: the overlap is rewritten
                 // with bit-identical values, every element sums its k products in the same order in any tile)
    m0 = min(m0, g.M - GEMM_BM);
    n0 = min(n0, g.N - GEMM_BN);
  }
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, l4 = lane >> 4;

  // fragment offsets (doubles) inside an image, per K-step; block i / j adds a_i / b_j
  int a_l[4], b_l[4];
  constexpr int a_i = A_KM ? 16 : 256, b_j = B_KN ? 16 : 256;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int swz = ((((kk << 1) | (l4 >> 1)) ^ (l15 >> 1)) << 1) | (l4 & 1); // swizzled k position, [m][k] images
    a_l[kk] = A_KM ? (kk * 4 + l4) * GEMM_LD_KM + wm * 64 + l15 : (wm * 64 + l15) * 16 + swz;
    b_l[kk] = B_KN ? (kk * 4 + l4) * GEMM_LD_KM + wn * 64 + l15 : (wn * 64 + l15) * 16 + swz;
  }

  // LDS-DMA pieces: wave w moves pieces p = 4w + j (j = 0..3) of each operand tile.
  //   [k][m]: piece p = k-row p;  source row stride ld, lane offset 16 B * lane;  LDS p * 144 doubles
  //   [m][k]: piece p = rows 8p..8p+7;  lane -> row 8p + (lane >> 3), chunk (lane & 7) ^ ((row >> 1) & 7);  LDS p * 128
  // One running per-lane source pointer per piece (16 VGPRs), advanced one per MFMA slot.
  const char *pA[4], *pB[4];
  constexpr int pa = A_KM ? GEMM_LD_KM : 128, pb = B_KN ? GEMM_LD_KM : 128; // LDS doubles per piece
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long p = 4 * wave + j;
    const int chunk = (lane & 7) ^ (4 * (j & 1) + (lane >> 4));
    pA[j] = reinterpret_cast<const char *>(A_KM ? g.A + p * g.lda + m0 + 2 * lane
                                                : g.A + (m0 + 8 * p + (lane >> 3)) * g.lda + 2 * chunk);
    pB[j] = reinterpret_cast<const char *>(B_KN ? g.B + p * g.ldb + n0 + 2 * lane
                                                : g.B + (n0 + 8 * p + (lane >> 3)) * g.ldb + 2 * chunk);
  }
  const long da = (A_KM ? (long)GEMM_BK * g.lda : (long)GEMM_BK) * 8; // source byte advance per K-tile
  const long db = (B_KN ? (long)GEMM_BK * g.ldb : (long)GEMM_BK) * 8;

  f64x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};

  const long nk = g.K / GEMM_BK;
  double xa[4], xb[4], ya[4], yb[4];

// one LDS-DMA piece q (0..3: A pieces, 4..7: B pieces) of the tile the source pointers stand on, into (AD, BD)
#define GEMMA_DMA(q, AD, BD)                                                                                   \
  do {                                                                                                         \
    if ((q) < 4)                                                                                               \
      __builtin_amdgcn_global_load_lds((gemma_gptr_t)pA[(q)&3],                                                \
                                       (gemma_lptr_t)((AD) + (4 * wave + ((q)&3)) * pa), 16, 0, 0);            \
    else                                                                                                       \
      __builtin_amdgcn_global_load_lds((gemma_gptr_t)pB[(q)&3],                                                \
                                       (gemma_lptr_t)((BD) + (4 * wave + ((q)&3)) * pb), 16, 0, 0);            \
  } while (0)
#define GEMMA_ADV(q)                                                                                           \
  do {                                                                                                         \
    if ((q) < 4) pA[(q)&3] += da;                                                                              \
    else pB[(q)&3] += db;                                                                                      \
  } while (0)
// fragment pair q (0,1: A blocks 0-1 / 2-3;  2,3: B blocks 0-1 / 2-3) of K-step KK: one ds_read2[st64]_b64
#define GEMMA_FRAG2(q, AS, BS, KK, FA, FB)                                                                     \
  do {                                                                                                         \
    if ((q) < 2) {                                                                                             \
      FA[2 * (q)] = (AS)[a_l[KK] + (2 * (q)) * a_i];                                                           \
      FA[2 * (q) + 1] = (AS)[a_l[KK] + (2 * (q) + 1) * a_i];                                                   \
    } else {                                                                                                   \
      FB[2 * ((q)-2)] = (BS)[b_l[KK] + (2 * ((q)-2)) * b_j];                                                   \
      FB[2 * ((q)-2) + 1] = (BS)[b_l[KK] + (2 * ((q)-2) + 1) * b_j];                                           \
    }                                                                                                          \
  } while (0)
#define GEMMA_MF(q, FA, FB)                                                                                    \
  acc[(q) >> 2][(q)&3] = __builtin_amdgcn_mfma_f64_16x16x4f64(FA[(q) >> 2], FB[(q)&3], acc[(q) >> 2][(q)&3], 0, 0, 0)
// 16 MFMAs of one K-step on (FA, FB); the fragments of the next K-step behind MFMAs 0-3; ADV: pointer advances
// behind MFMAs 8-15
#define GEMMA_STEP(FA, FB, AS, BS, KK, GA, GB, ADV)                                                            \
  do {                                                                                                         \
    _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) {                                                        \
      GEMMA_MF(q_, FA, FB);                                                                                    \
      if (q_ < 4) GEMMA_FRAG2(q_, AS, BS, KK, GA, GB);                                                         \
      if ((ADV) && q_ >= 8) GEMMA_ADV(q_ - 8);                                                                 \
      GEMMA_SB();                                                                                              \
    }                                                                                                          \
  } while (0)
// one K-tile; MORE: tile t+1 exists (its first fragments are read at the end); LOAD2: tile t+2 exists (DMA it)
#define GEMMA_KTILE(AC, BC, AN, BN, MORE, LOAD2)                                                               \
  do {                                                                                                         \
    GEMMA_STEP(xa, xb, AC, BC, 1, ya, yb, false);                                                              \
    GEMMA_STEP(ya, yb, AC, BC, 2, xa, xb, false);                                                              \
    GEMMA_STEP(xa, xb, AC, BC, 3, ya, yb, LOAD2);                                                              \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                         \
      GEMMA_MF(q_, ya, yb);                                                                                    \
      GEMMA_SB();                                                                                              \
    }                                                                                                          \
    __syncthreads();                                                                                           \
    GEMMA_SB();                                                                                                \
    _Pragma("unroll") for (int q_ = 4; q_ < 16; ++q_) {                                                        \
      GEMMA_MF(q_, ya, yb);                                                                                    \
      if (q_ < 8) {                                                                                            \
        if (MORE) GEMMA_FRAG2(q_ - 4, AN, BN, 0, xa, xb);                                                      \
      } else {                                                                                                 \
        if (LOAD2) GEMMA_DMA(q_ - 8, AC, BC);                                                                  \
      }                                                                                                        \
      GEMMA_SB();                                                                                              \
    }                                                                                                          \
  } while (0)

  double *const As0 = lds, *const As1 = lds + GEMM_TILE_DOUBLES;
  double *const Bs0 = lds + 2 * GEMM_TILE_DOUBLES, *const Bs1 = lds + 3 * GEMM_TILE_DOUBLES;
#pragma unroll
  for (int q = 0; q < 8; ++q) GEMMA_DMA(q, As0, Bs0);
  if (nk > 1) {
#pragma unroll
    for (int q = 0; q < 8; ++q) GEMMA_ADV(q);
#pragma unroll
    for (int q = 0; q < 8; ++q) GEMMA_DMA(q, As1, Bs1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); // tile 0 has landed, tile 1 may still be in flight
    __builtin_amdgcn_s_barrier();
  } else {
    __syncthreads();
  }
  GEMMA_SB();
#pragma unroll
  for (int q = 0; q < 4; ++q) GEMMA_FRAG2(q, As0, Bs0, 0, xa, xb);

  // the source pointers stand on tile kt+1 when K-tile kt starts; its third K-step moves them to tile kt+2
  long kt = 0;
  for (; kt + 2 < nk; ++kt) { // steady state
    const bool odd = (kt & 1) != 0;
    double *Ac = odd ? As1 : As0, *Bc = odd ? Bs1 : Bs0;
    double *An = odd ? As0 : As1, *Bn = odd ? Bs0 : Bs1;
    GEMMA_KTILE(Ac, Bc, An, Bn, true, true);
  }
  if (nk >= 2) { // K-tile nk-2: nothing left to load
    const bool odd = (nk & 1) != 0;
    double *Ac = odd ? As1 : As0, *Bc = odd ? Bs1 : Bs0;
    double *An = odd ? As0 : As1, *Bn = odd ? Bs0 : Bs1;
    GEMMA_KTILE(Ac, Bc, An, Bn, true, false);
  }
  { // K-tile nk-1
    const bool odd = (nk & 1) == 0;
    double *Ac = odd ? As1 : As0, *Bc = odd ? Bs1 : Bs0;
    double *An = odd ? As0 : As1, *Bn = odd ? Bs0 : Bs1;
    GEMMA_KTILE(Ac, Bc, An, Bn, false, false);
  }
#undef GEMMA_DMA
#undef GEMMA_ADV
#undef GEMMA_FRAG2
#undef GEMMA_MF
#undef GEMMA_STEP
#undef GEMMA_KTILE

  const double alpha = g.alpha, beta = g.beta;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long col = n0 + wn * 64 + j * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = m0 + wm * 64 + i * 16 + l4 + 4 * r;
        double *c = g.C + row * g.ldc + col;
        double v = alpha * acc[i][j][r];
        if (beta != 0.0) v += beta * (*c);
        *c = v;
        if (g.mirror && tm != tn) g.C[col * g.ldc + row] = v;
      }
    }
  }
}

// mirror the strict upper triangle into the lower one and scale everything (kinship epilogue:
// K *= 1/ns_test, GEMMA src/gemma_io.cpp:1570, and the symmetric fill of :1724-1729)
static __global__ void symm_fill_scale_kernel(double *K, long n, long ld, double scale) {
  __shared__ double tile[32][33];
  // blocks cover the upper-triangular 32x32 tile pairs (bx >= by)
  const int bx = blockIdx.x, by = blockIdx.y;
  if (bx < by) return;
  const int tx = threadIdx.x, ty = threadIdx.y; // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const long i = (long)by * 32 + r, j = (long)bx * 32 + tx;
    double v = 0.0;
    if (i < n && j < n) {
      v = K[i * ld + j] * scale;
      if (j >= i) K[i * ld + j] = v;
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    // element (j, i) of the lower triangle <- tile[i_local][j_local]
    const long j = (long)bx * 32 + r, i = (long)by * 32 + tx;
    if (i < n && j < n && j > i) K[j * ld + i] = tile[tx][r];
  }
}

// Wavefronts per 128x128 block: 4 (2x2 waves of 64x64, 2 waves/SIMD at 2 blocks/CU; default) or 8 (2x4 waves
// of 64x32, 4 waves/SIMD; GEMMA_HIP_GEMM_WAVES=8).  Measured at M=N=K=20000 (profiles/r01_gemm_variants.txt):
// the 8-wave form is 1.7 % faster (238.8 vs 242.6 ms) but its blocks drift apart inside an XCD and the L2 hit
// rate drops from 82 % to 17 % (fabric traffic x4); the 4-wave form keeps the co-scheduled tiles in step.
static inline int gemm_waves() {
  static int nw = 0;
  if (nw == 0) {
    const char *e = getenv("GEMMA_HIP_GEMM_WAVES");
    nw = (e && e[0] == '8') ? 8 : 4;
  }
  return nw;
}

// GEMMA_HIP_GEMM_PIPE=0/1/2: interior kernel = register-staged / + software-pipelined fragments / direct-to-LDS with
// pinned interleaved issue (default; 243 -> 238 -> 219 ms at M = N = K = 20000)
static inline int gemm_pipe() {
  static int v = -1;
  if (v < 0) {
    const char *e = getenv("GEMMA_HIP_GEMM_PIPE");
    v = e ? atoi(e) : 2;
  }
  return v;
}

static inline bool gemm_clamp() { // GEMMA_HIP_GEMM_CLAMP=0: ragged strips through the bounds-checked kernel
  static int v = -1;
  if (v < 0) {
    const char *e = getenv("GEMMA_HIP_GEMM_CLAMP");
    v = (e && e[0] == '0') ? 0 : 1;
  }
  return v == 1;
}

template <bool A_KM, bool B_KN, bool FULL>
static inline hipError_t launch_dgemm_grid(const GemmArgs &g, hipStream_t s) {
  int nblocks;
  if (g.syrk_upper)
    nblocks = g.tiles_m * (g.tiles_m + 1) / 2;
  else
    nblocks = g.tiles_m * g.tiles_n;
  if (nblocks <= 0) return hipSuccess;
  if (FULL && gemm_pipe() == 2 && !g.square_a)
    hipLaunchKernelGGL((dgemm_mfma_glds_kernel<A_KM, B_KN>), dim3(nblocks, g.kslices), dim3(256), 0, s, g);
  else if (FULL && gemm_pipe())
    hipLaunchKernelGGL((dgemm_mfma_pipe_kernel<A_KM, B_KN>), dim3(nblocks, g.kslices), dim3(256), 0, s, g);
  else if (gemm_waves() == 8)
    hipLaunchKernelGGL((dgemm_mfma_kernel<A_KM, B_KN, 8, FULL>), dim3(nblocks, g.kslices), dim3(512), 0, s, g);
  else
    hipLaunchKernelGGL((dgemm_mfma_kernel<A_KM, B_KN, 4, FULL>), dim3(nblocks, g.kslices), dim3(256), 0, s, g);
  return hipGetLastError();
}

// Optional side stream for the ragged edge strips: they are tiny launches (<= 2 x 157 blocks at n = 20000)
// that would otherwise run alone after the main grid; on a second stream they fill the main grid's tail.
struct GemmAux {
  hipStream_t stream = nullptr;
  hipEvent_t ready = nullptr, done = nullptr;
};
static GemmAux g_gemm_aux;
static inline void gemm_aux_init() {
  if (g_gemm_aux.stream) return;
  const char *e = getenv("GEMMA_HIP_GEMM_SIDE_STREAM"); // "0": keep everything on the caller's stream
  if (e && e[0] == '0') return;
  if (hipStreamCreateWithFlags(&g_gemm_aux.stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&g_gemm_aux.ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&g_gemm_aux.done, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    g_gemm_aux = GemmAux();
  }
}
static inline void gemm_aux_destroy() {
  if (g_gemm_aux.stream) {
    (void)hipStreamDestroy(g_gemm_aux.stream);
    (void)hipEventDestroy(g_gemm_aux.ready);
    (void)hipEventDestroy(g_gemm_aux.done);
  }
  g_gemm_aux = GemmAux();
}

template <bool A_KM, bool B_KN>
static inline hipError_t launch_dgemm_t(GemmArgs g, hipStream_t s) {
  const int Tm = (int)((g.M + GEMM_BM - 1) / GEMM_BM), Tn = (int)((g.N + GEMM_BN - 1) / GEMM_BN);
  const int Fm = (int)(g.M / GEMM_BM), Fn = (int)(g.N / GEMM_BN); // complete tiles per dimension
  const bool aligned = ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0) && ((g.lda & 1) == 0) &&
                       ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0) && ((g.ldb & 1) == 0);
  hipError_t e;
  // K tail: the first floor(K / 16) * 16 of K through the fast kernels, the remaining < 16 accumulated on top by the
  // bounds-checked one (the eigensolver's divide & conquer merges have arbitrary K)
  if (aligned && g.K > 4 * GEMM_BK && (g.K % GEMM_BK) != 0 && Fm > 0 && Fn > 0 && !g.square_a) {
    const long K0 = (g.K / GEMM_BK) * GEMM_BK, K1 = g.K - K0;
    GemmArgs g0 = g;
    g0.K = K0;
    if ((e = launch_dgemm_t<A_KM, B_KN>(g0, s)) != hipSuccess) return e;
    GemmArgs g1 = g;
    g1.K = K1;
    g1.A = A_KM ? g.A + K0 * g.lda : g.A + K0;
    g1.B = B_KN ? g.B + K0 * g.ldb : g.B + K0;
    g1.beta = 1.0;
    g1.clamp = 0;
    g1.tiles_m = Tm; g1.tiles_n = Tn; g1.tm0 = 0; g1.tn0 = 0;
    if (g.syrk_upper) g1.tiles_n = Tm;
    return launch_dgemm_grid<A_KM, B_KN, false>(g1, s);
  }
  const bool fast = aligned && (g.K % GEMM_BK == 0) && g.K > 0 && Fm > 0 && Fn > 0;
  if (!fast) { // everything through the bounds-checked instantiation
    g.tiles_m = Tm; g.tiles_n = Tn; g.tm0 = 0; g.tn0 = 0;
    return launch_dgemm_grid<A_KM, B_KN, false>(g, s);
  }
  const int syrk = g.syrk_upper;
  g.clamp = 0;
  // beta == 0: ragged last tile rows / columns are shifted back inside the matrix and ride in the same launch
  // (bit-identical rewrites of the overlap); the [k][m] operand of a shifted tile must stay 16-byte aligned
  if (gemm_pipe() >= 2 && !syrk && !g.square_a && g.beta == 0.0 && (Tm > Fm || Tn > Fn) && gemm_clamp() &&
      (!A_KM || (g.M & 1) == 0) && (!B_KN || (g.N & 1) == 0)) {
    g.clamp = 1;
    g.tiles_m = Tm; g.tiles_n = Tn; g.tm0 = 0; g.tn0 = 0;
    return launch_dgemm_grid<A_KM, B_KN, true>(g, s);
  }
  const bool strips = (Tn > Fn) || (Tm > Fm && !syrk);
  const bool side = strips && g_gemm_aux.stream != nullptr && Fm * Fn >= 512;
  hipStream_t es = side ? g_gemm_aux.stream : s; // stream of the edge strips
  if (side) {
    if ((e = hipEventRecord(g_gemm_aux.ready, s)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(es, g_gemm_aux.ready, 0)) != hipSuccess) return e;
  }
  GemmArgs ge = g;
  ge.syrk_upper = 0;
  if (side) { // strips first on the side stream, the big grid on the caller's stream
    if (Tn > Fn) { // ragged right strip: all tile rows (SYRK: tm <= Tn-1 is every row)
      ge.tiles_m = Tm; ge.tiles_n = 1; ge.tm0 = 0; ge.tn0 = Fn;
      if ((e = launch_dgemm_grid<A_KM, B_KN, false>(ge, es)) != hipSuccess) return e;
    }
    if (Tm > Fm && !syrk) { // ragged bottom strip (complete columns only; the corner went with the right strip)
      ge.tiles_m = 1; ge.tiles_n = Fn; ge.tm0 = Fm; ge.tn0 = 0;
      if ((e = launch_dgemm_grid<A_KM, B_KN, false>(ge, es)) != hipSuccess) return e;
    }
  }
  // complete tiles: predicate-free kernel (SYRK: the upper triangle of the Fm x Fm complete tiles)
  g.tiles_m = Fm; g.tiles_n = Fn; g.tm0 = 0; g.tn0 = 0;
  e = launch_dgemm_grid<A_KM, B_KN, true>(g, s);
  if (e != hipSuccess) return e;
  if (side) {
    if ((e = hipEventRecord(g_gemm_aux.done, es)) != hipSuccess) return e;
    return hipStreamWaitEvent(s, g_gemm_aux.done, 0);
  }
  if (Tn > Fn) {
    ge.tiles_m = Tm; ge.tiles_n = 1; ge.tm0 = 0; ge.tn0 = Fn;
    if ((e = launch_dgemm_grid<A_KM, B_KN, false>(ge, s)) != hipSuccess) return e;
  }
  if (Tm > Fm && !syrk) {
    ge.tiles_m = 1; ge.tiles_n = Fn; ge.tm0 = Fm; ge.tn0 = 0;
    if ((e = launch_dgemm_grid<A_KM, B_KN, false>(ge, s)) != hipSuccess) return e;
  }
  return hipSuccess;
}

// ta/tb in {'N','T'} with the cblas row-major meaning; mirror (with syrk_upper, M == N): the lower triangle is written too, as the
// transpose of the upper tiles (GemmArgs::mirror)
static inline hipError_t launch_dgemm(char ta, char tb, long M, long N, long K, double alpha,
                                      const double *A, long lda, const double *B, long ldb,
                                      double beta, double *C, long ldc, bool syrk_upper,
                                      bool square_a, hipStream_t s, bool mirror = false) {
  GemmArgs g;
  g.mirror = (mirror && (syrk_upper ? M == N : M <= N)) ? 1 : 0; // without syrk_upper: a row strip of such an update (tiles (0, j) of C: the
                                                                 // eigensolver's look-ahead); every tile off the diagonal is also written transposed
  g.A = A; g.B = B; g.C = C;
  g.M = M; g.N = N; g.K = K;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.alpha = alpha; g.beta = beta;
  g.tiles_m = g.tiles_n = 0;
  g.tm0 = g.tn0 = 0;
  g.syrk_upper = syrk_upper ? 1 : 0;
  g.clamp = 0;
  g.square_a = square_a ? 1 : 0;
  {
    static int abl = -1;
    if (abl < 0) {
      const char *e = getenv("GEMMA_HIP_GEMM_ABLATE");
      abl = e ? atoi(e) : 0;
    }
    g.ablate = abl;
    static int gm = -1;
    if (gm < 0) {
      const char *e = getenv("GEMMA_HIP_GEMM_GM");
      gm = e ? atoi(e) : 0;
    }
    g.gm = gm;
  }
  const bool tA = (ta == 'T' || ta == 't'), tB = (tb == 'T' || tb == 't');
  // op(A) = A^T  <=> A stored [k][m]  (KM image);  op(B) = B <=> B stored [k][n] (KN image)
  if (tA && !tB) return launch_dgemm_t<true, true>(g, s);
  if (tA && tB) return launch_dgemm_t<true, false>(g, s);
  if (!tA && !tB) return launch_dgemm_t<false, true>(g, s);
  return launch_dgemm_t<false, false>(g, s);
}

// How K is cut for launch_dgemm_ksliced: K0 = the part that is whole K-tiles, `per` = length of a slice (a multiple of the K-tile),
// return = number of slices: every slice is non-empty ((ns - 1) per < K0 <= ns per) and -- when more than one -- at least four K-tiles
// long except possibly the last.  1 means "do not slice".  (tests/cpp/raster_slices_check.hip runs this for every K <= 40 000.)
static inline int gemm_kslice_plan(long K, int want, long *K0, long *per) {
  *K0 = (K / GEMM_BK) * GEMM_BK;
  *per = *K0;
  int ns = want < 1 ? 1 : want;
  while (ns > 1 && *K0 / ns < 4 * GEMM_BK) --ns;
  if (ns <= 1 || *K0 == 0) return 1;
  const long kt = *K0 / GEMM_BK, pt = (kt + ns - 1) / ns; // K-tiles in all, per slice
  ns = (int)((kt + pt - 1) / pt);
  *per = pt * GEMM_BK;
  return ns;
}

// C_ks (ks = 0 .. *nslices - 1, at C + ks * cslice) = alpha op(A) op(B) over the K range of slice ks, ONE launch per kernel
// variant (interior tiles / edge strips) with the slice in blockIdx.y; the caller adds the slices.  want = slices asked for; the
// number used (returned in *nslices) keeps every slice at least four K-tiles long.  A K that is not a multiple of the K-tile
// leaves its last < 16 columns to one extra bounds-checked launch that accumulates into slice 0.
static inline hipError_t launch_dgemm_ksliced(char ta, char tb, long M, long N, long K, double alpha, const double *A, long lda,
                                              const double *B, long ldb, double *C, long ldc, long cslice, int want,
                                              int *nslices, hipStream_t s) {
  const bool tA = (ta == 'T' || ta == 't'), tB = (tb == 'T' || tb == 't');
  long K0, per;
  const int ns = gemm_kslice_plan(K, want, &K0, &per);
  const long K1 = K - K0;
  *nslices = ns;
  if (ns <= 1) return launch_dgemm(ta, tb, M, N, K, alpha, A, lda, B, ldb, 0.0, C, ldc, false, false, s);
  GemmArgs g;
  g.A = A; g.B = B; g.C = C;
  g.M = M; g.N = N; g.K = K0;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.alpha = alpha; g.beta = 0.0;
  g.tiles_m = g.tiles_n = 0;
  g.tm0 = g.tn0 = 0;
  g.syrk_upper = 0; g.clamp = 0; g.square_a = 0; g.ablate = 0; g.gm = 0;
  g.kslices = ns; g.kslice = per; g.cslice = cslice;
  hipError_t e;
  // op(A) = A^T  <=> A stored [k][m]  (KM image);  op(B) = B <=> B stored [k][n] (KN image)
  if (tA && !tB) e = launch_dgemm_t<true, true>(g, s);
  else if (tA && tB) e = launch_dgemm_t<true, false>(g, s);
  else if (!tA && !tB) e = launch_dgemm_t<false, true>(g, s);
  else e = launch_dgemm_t<false, false>(g, s);
  if (e != hipSuccess || K1 == 0) return e;
  const double *A1 = tA ? A + K0 * lda : A + K0, *B1 = tB ? B + K0 : B + K0 * ldb;
  return launch_dgemm(ta, tb, M, N, K1, alpha, A1, lda, B1, ldb, 1.0, C, ldc, false, false, s);
}

} // namespace gemma_hip
