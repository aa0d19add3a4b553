// Exact integer form of UtX = X * U for hard-call genotypes (PLINK .bed): int8 MFMA, int32 accumulation.
//
// A PLINK genotype row after GEMMA's mean imputation (src/lmm.cpp:1797-1827) is  x_s = g_s + mean_s * m_s  with
// g in {0,1,2} (0 where the call is missing) and m the 0/1 missing mask, so
//     (U^T x_s)[j] = sum_k g_sk U[k][j] + mean_s * sum_k m_sk U[k][j].
// Both sums have an EXACTLY representable small-integer left factor.  U's column j is scaled by a power of two and
// rounded to a 55-bit integer (|V| <= 2^54: every entry in the column's top binade is exact, absolute error elsewhere
// <= 2^-55 of the column maximum), written in balanced base 256: V = sum_d 256^d D_d, D_d in [-128,127], d = 0..6.
// Each digit product  [G; M] (int8) x D_d (int8)  accumulates exactly in int32 (|sum| <= 20000*2*128 < 2^23), i.e.
// v_mfma_i32_32x32x32_i8 work at ~64x the fp64 MFMA rate; 7 digits x 2 left factors = 14 products replace the one
// fp64 product.  The fp64 result is assembled once per element (Horner over the digits, <= 2 roundings), which is
// closer to the exact dot product than an fp64 GEMM's 20000-term rounding chain.
//
// Kernel: 256 x 256 x 128-byte tiles, 512 threads = 8 wavefronts (2 x 4), wave tile 128 x 64 = 4 x 2 MFMA blocks
// (128 int32 accumulators), operands global -> LDS by global_load_lds_dwordx4 into two 64 KiB stages; both LDS
// images are [row][128 bytes of K] with the 16-byte chunk index XOR-ed by (row >> 1) & 7 on the source address and
// on the ds_read_b128 fragment reads (conflict-free).  One block per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>
#include "dgemm_mfma.hip.h"
#include "ingest.hip.h"

namespace gemma_hip {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

constexpr int I8_BM = 256, I8_BN = 256, I8_BK = 128;
constexpr int I8_DIGITS = 7;
constexpr int I8_SCALE_BITS = 54;

struct I8GemmArgs {
  const int8_t *A;   // M x ldk, K contiguous (rows: SNPs g, then SNP masks m)
  const int8_t *Bt;  // digit d: N x ldk, K contiguous (row j = column j of U), digits strideB bytes apart
  int *C;            // digit d: M x ldc int32, digits strideC elements apart
  long ldk, ldc;
  long strideB, strideC;
  int tiles_m, tiles_n;
  int nk;            // K tiles of 128 bytes
  int gm;            // raster group height (tile rows)
};

__device__ __forceinline__ void i8_tile_of_block(const I8GemmArgs &g, int &tm, int &tn) {
  const int nwg = gridDim.x, b = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, x = b & 7, o = b >> 3;
  const int L = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o; // XCD x gets a contiguous range
  const int GM = g.gm > 0 ? g.gm : 4;
  const int per_group = GM * g.tiles_n;
  const int grp = L / per_group;
  const int first_m = grp * GM;
  const int gsz = min(g.tiles_m - first_m, GM);
  const int in = L - grp * per_group;
  tm = first_m + in % gsz;
  tn = in / gsz;
}

__global__ __launch_bounds__(512, 2) void i8gemm_kernel(I8GemmArgs g) {
  extern __shared__ __attribute__((aligned(1024))) int8_t i8lds[]; // 2 stages x (A 32 KiB + B 32 KiB)
  int tm, tn;
  i8_tile_of_block(g, tm, tn);
  const int digit = blockIdx.y;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3; // 2 x 4 waves: rows wm*128, cols wn*64
  const int r32 = lane & 31, h = lane >> 5;

  const int8_t *Ag = g.A + (long)tm * I8_BM * g.ldk;
  const int8_t *Bg = g.Bt + (long)digit * g.strideB + (long)tn * I8_BN * g.ldk;
  int *Cg = g.C + (long)digit * g.strideC;

  // LDS-DMA: 32 pieces of 1 KiB per operand tile (piece p = rows 8p..8p+7); wave w moves pieces 4w..4w+3 of each.
  // lane -> row 8p + (lane >> 3), physical chunk lane & 7 holds logical chunk (lane & 7) ^ ((row >> 1) & 7)
  const int8_t *pA[4], *pB[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = 4 * wave + j;
    const int row = 8 * p + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    pA[j] = Ag + (long)row * g.ldk + 16 * chunk;
    pB[j] = Bg + (long)row * g.ldk + 16 * chunk;
  }
  // fragment byte offsets inside an operand image: block i of K-step ks: row = base + 32 i + r32, logical chunk 2 ks + h
  int fa[4], fb[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int sw = ((2 * ks + h) ^ ((r32 >> 1) & 7)) << 4;
    fa[ks] = (wm * 128 + r32) * 128 + sw;
    fb[ks] = (wn * 64 + r32) * 128 + sw;
  }

  i32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

#define GEMMA_I8_DMA(STAGE)                                                                                  \
  do {                                                                                                       \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                       \
      __builtin_amdgcn_global_load_lds((gemma_gptr_t)pA[j_],                                                 \
                                       (gemma_lptr_t)(i8lds + (STAGE)*65536 + (4 * wave + j_) * 1024), 16, 0, 0); \
      __builtin_amdgcn_global_load_lds((gemma_gptr_t)pB[j_],                                                 \
                                       (gemma_lptr_t)(i8lds + (STAGE)*65536 + 32768 + (4 * wave + j_) * 1024), 16, 0, 0); \
      pA[j_] += I8_BK;                                                                                       \
      pB[j_] += I8_BK;                                                                                       \
    }                                                                                                        \
  } while (0)

  GEMMA_I8_DMA(0);
  __syncthreads();
  for (int kt = 0; kt < g.nk; ++kt) {
    const int st = kt & 1;
    if (kt + 1 < g.nk) {
      if (st) GEMMA_I8_DMA(0); else GEMMA_I8_DMA(1);
    }
    const int8_t *As = i8lds + st * 65536, *Bs = As + 32768;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      i32x4 a[4], b[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const i32x4 *>(As + fa[ks] + i * 32 * 128);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const i32x4 *>(Bs + fb[ks] + j * 32 * 128);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
#undef GEMMA_I8_DMA

  // C/D map of the 32x32 forms: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long col = (long)tn * I8_BN + wn * 64 + j * 32 + r32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long row = (long)tm * I8_BM + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        Cg[row * g.ldc + col] = acc[i][j][r];
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// U -> per-column exponent and the 7 balanced base-256 digit matrices, transposed ([j][k], K contiguous)
__global__ __launch_bounds__(256) void u_colmax_kernel(const double *__restrict__ U, long n, long ld,
                                                       unsigned long long *__restrict__ colmax_bits) {
  // |u| as an integer key: IEEE doubles order like their bit patterns once the sign is cleared
  const long j = (long)blockIdx.x * 256 + threadIdx.x;
  const long k0 = (long)blockIdx.y * 1024, k1 = min(n, k0 + 1024);
  if (j >= n) return;
  unsigned long long m = 0;
  for (long k = k0; k < k1; ++k) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(fabs(U[k * ld + j]));
    m = b > m ? b : m;
  }
  atomicMax(colmax_bits + j, m);
}

// colmax_bits[j] -> e_j with max|u| < 2^e_j (0 for an all-zero column); in place as int
__global__ void u_exponent_kernel(const unsigned long long *__restrict__ colmax_bits, long n, int *__restrict__ ej) {
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const double m = __longlong_as_double((long long)colmax_bits[j]);
  int e = 0;
  if (m > 0.0 && m <= DBL_MAX) (void)frexp(m, &e);
  ej[j] = e;
}

// 32 x 32 tile of U (rows k, cols j) -> digit tiles [j][k]
__global__ __launch_bounds__(256) void u_digits_kernel(const double *__restrict__ U, long n, long ld,
                                                       const int *__restrict__ ej, int8_t *__restrict__ Bt, long ldk,
                                                       long strideB) {
  __shared__ long long tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
  const long k0 = (long)blockIdx.y * 32, j0 = (long)blockIdx.x * 32;
  for (int r = ty; r < 32; r += 8) {
    const long k = k0 + r, j = j0 + tx;
    long long v = 0;
    if (k < n && j < n) v = llrint(ldexp(U[k * ld + j], I8_SCALE_BITS - ej[j]));
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const long j = j0 + r, k = k0 + tx; // write row j, column k (k contiguous across tx)
    long long v = tile[tx][r];
#pragma unroll
    for (int d = 0; d < I8_DIGITS; ++d) {
      const int dig = (int)(int8_t)(unsigned char)(v & 0xff); // low byte as a signed digit
      v = (v - dig) >> 8;                                      // exact: v - dig is a multiple of 256
      Bt[(long)d * strideB + j * ldk + k] = (int8_t)dig;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// PLINK rows -> G rows [0, l), M rows [m_row0, m_row0 + l) of the int8 left factor, and mean_s
struct IngestI8Args {
  const unsigned char *src;
  long ld, l;
  const int *idx_map;
  int n;
  int8_t *A;
  long ldk;
  long m_row0;
  double *mean;
};
__global__ __launch_bounds__(256) void ingest_i8_kernel(IngestI8Args g) {
  const int lane = threadIdx.x & 63;
  const long s = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= g.l) return;
  const unsigned char *bs = g.src + s * g.ld;
  int8_t *gr = g.A + s * g.ldk, *mr = g.A + (g.m_row0 + s) * g.ldk;
  double tot = 0.0, cnt = 0.0;
  for (int i = lane; i < g.n; i += 64) {
    const int p = g.idx_map ? g.idx_map[i] : i;
    bool miss;
    const double v = plink_value((bs[p >> 2] >> (2 * (p & 3))) & 3u, miss);
    gr[i] = miss ? (int8_t)0 : (int8_t)(int)v;
    mr[i] = miss ? (int8_t)1 : (int8_t)0;
    if (!miss) { tot += v; cnt += 1.0; }
  }
  for (long i = g.n + lane; i < g.ldk; i += 64) { gr[i] = 0; mr[i] = 0; } // K padding
  tot = wsum(tot);
  cnt = wsum(cnt);
  if (lane == 0) g.mean[s] = tot / cnt; // x_total / (ni_test - n_miss), as ingest_lmm_kernel
}

// UtX[s][j] = 2^(e_j - 54) * sum_d 256^d (CG_d[s][j] + mean_s * CM_d[s][j])
__global__ __launch_bounds__(256) void i8_combine_kernel(const int *__restrict__ C, long ldc, long strideC, long m_row0,
                                                         const double *__restrict__ mean, const int *__restrict__ ej,
                                                         long l, long n, double *__restrict__ UtX, long ldx) {
  const long j = (long)blockIdx.x * 256 + threadIdx.x;
  const long s = blockIdx.y;
  if (j >= n || s >= l) return;
  double tg = 0.0, tmk = 0.0;
#pragma unroll
  for (int d = I8_DIGITS - 1; d >= 0; --d) {
    tg = tg * 256.0 + (double)C[(long)d * strideC + s * ldc + j];
    tmk = tmk * 256.0 + (double)C[(long)d * strideC + (m_row0 + s) * ldc + j];
  }
  UtX[s * ldx + j] = ldexp(fma(mean[s], tmk, tg), ej[j] - I8_SCALE_BITS);
}

} // namespace gemma_hip
