// Exact integer form of UtX = X * U for hard-call genotypes (PLINK .bed): int8 MFMA, int32 accumulation.
//
// A PLINK genotype row after GEMMA's mean imputation (src/lmm.cpp:1797-1827) is  x_s = g_s + mean_s * m_s  with
// g in {0,1,2} (0 where the call is missing) and m the 0/1 missing mask, so
//     (U^T x_s)[j] = sum_k g_sk U[k][j] + mean_s * sum_k m_sk U[k][j].
// Both sums have an EXACTLY representable small-integer left factor.  U's column j is scaled (round 6: by 0.99 * 2^(8 D - 1) over its
// exact maximum, u_scale_kernel below; rounds 1-5: by a power of two) and rounded to an integer of D balanced base-256 digits:
// V = sum_d 256^d D_d, D_d in [-128,127], d = 0..D-1 (absolute error <= 1.01 * 2^(-8 D) of the column maximum).
// Each digit product  [G; M] (int8) x D_d (int8)  accumulates exactly in int32 (|sum| <= 20000*2*128 < 2^23), i.e.
// v_mfma_i32_32x32x32_i8 work at ~64x the fp64 MFMA rate; 7 digits x 2 left factors = 14 products replace the one
// fp64 product.  The fp64 result is assembled once per element (Horner over the digits, <= 2 roundings), which is
// closer to the exact dot product than an fp64 GEMM's 20000-term rounding chain.
//
// Kernel (i8gemm_packed_kernel_t<true> below): 128 SNP rows x 256 columns x 128 K-bytes per tile, 512 threads = 8 wavefronts,
// operands global -> LDS by global_load_lds_dwordx4 into three 48 KiB stages; both LDS images are [row][128 bytes of K]
// with the 16-byte chunk index XOR-ed by (row >> 1) & 7 on the source address and on the ds_read_b128 fragment reads
// (conflict-free).  One block per CU.  History and ablations: profiles/r01_i8gemm_variants.txt.
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
constexpr int I8_DIGITS = 7;      // most digits a build handles (buffers are sized for it)
constexpr int SUR_MAX = 16;       // calls per row the sparse mask operand may drop before the row goes to the fp64 fix-up: the stride of
                                  // the dropped-call lists, shared by the list kernel (i8gemm_sparse.hip.h), the combine below and the host
constexpr int I8_SCALE_BITS = 54;  // with 7 digits; D digits scale to 8 D - 2 bits (|V| <= 2^(8D-2) < 128 * 256^(D-1))
// Digits actually used (host: i8_digits_for): 7 round U at 2^-56 of each column's maximum.  From n = 16384 up 6 digits are used: U is
// then rounded at 2^-48 of the column maximum, which perturbs a dot product by cmax |x|_2 / (0.99 * 2^47 * sqrt(12)) -- measured against
// long-double products 1.66e-16 rms in units of sum_k |x_k||u_k| (the fp64 MFMA GEMM: 3.6e-17), 4.6 x the GEMM's, for 6/7 of the
// matrix-pipe cycles; GEMMA_HIP_I8_FORM=7g6m (seven digits for the genotype product, six for the mask product) is below the GEMM's
// (tests/test_gpu_at_size.py::test_six_digit_rounding_of_U_is_what_the_model_says).  i8_scale_bits: the power-of-two scale of rounds 1-5.
__host__ __device__ inline int i8_scale_bits(int digits) { return 8 * digits - 2; }

// ---------------------------------------------------------------------------------------------------------------
// ONE packed left factor per SNP row, byte = g | (m << 4).  A wave reads an A fragment from LDS once and
// masks it into the genotype operand (a & 0x03) and the missing-mask operand ((a >> 4) & 0x01) -- the G and M
// products share every global / LDS byte of both operands, so LDS traffic per MFMA drops by a third and the left
// factor is read once.  Two digits of U are fused per output plane where 256 * C_{d+1} + C_d still fits int32
// (n <= 32640): the K loop runs for the upper digit, the accumulators are shifted left by 8, and it runs again for
// the lower digit -- 4 int32 planes leave the kernel instead of 7.
//   tile 128 SNP rows x 256 columns x 128 K bytes; 8 wavefronts (2 x 4), wave tile 64 x 64 for G and for M
//   (2 x 2 x 2 blocks of 32 x 32 = 128 int32 accumulators); three LDS stages of 48 KiB (A 16 KiB + B 32 KiB), LDS-DMA
//   two K-tiles ahead with a counted vmcnt and a raw s_barrier; pinned issue order: one ds_read_b128 / LDS-DMA piece /
//   pair of v_and behind each MFMA, fragments one K-step ahead, the barrier after MFMA 25 of 32 so that the next
//   tile's first fragments are already there when this tile ends.
struct I8PackArgs {
  const int8_t *A;   // lpad x ldk packed bytes
  const int8_t *Bt;  // digit d: N x ldk
  int *C;            // plane q: (2 lpad) x ldc; rows [0, lpad) = G products, [lpad, 2 lpad) = M products
  long ldk, ldc;
  long strideB, strideC;
  long m_row0;       // lpad
  int tiles_m, tiles_n;
  int nk;
  int gm;
  int fuse;          // 1: two digits per output plane (needs n * 2 * 128 * 257 < 2^31): 7 digits -> planes {0}, {2,1},
                     // {4,3}, {6,5}; 6 digits -> {1,0}, {3,2}, {5,4};  0: one plane per digit
  int digits;        // 6 or 7
  const int *tile_map = nullptr; // (tile_m, tile_n) per linear tile index when only some tiles are wanted (the symmetric
                                 // product of kin_i8.hip.h: tiles that meet the upper triangle); the grid has that many blocks
};
constexpr int I8P_BM = 128;
constexpr int I8P_STAGE = 49152;

// WITH_M = false: the genotype product alone (kin_i8.hip.h: G^T G needs no mask product) -- the same schedule with the mask
// MFMAs, their operand masks and their epilogue left out
// RAW = true (dosage planes, below): the A bytes are signed int8 values used as they are (no genotype mask)
template <bool WITH_M, bool RAW = false>
__global__ __launch_bounds__(512, 2) void i8gemm_packed_kernel_t(I8PackArgs g) {
  extern __shared__ __attribute__((aligned(1024))) int8_t i8lds[];
  int tm, tn;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, x = b & 7, o = b >> 3;
    const int L = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + o;
    const int GM = g.gm > 0 ? g.gm : 8;
    const int per_group = GM * g.tiles_n;
    const int grp = L / per_group;
    const int first_m = grp * GM;
    const int gsz = min(g.tiles_m - first_m, GM);
    const int in = L - grp * per_group;
    tm = first_m + in % gsz;
    tn = in / gsz;
    if (g.tile_map) {
      tm = g.tile_map[2 * L];
      tn = g.tile_map[2 * L + 1];
    }
  }
  // output plane q: fused pairs of digits when g.fuse (256 * C_{d+1} + C_d still fits int32): planes
  // {0}, {2,1}, {4,3}, {6,5}; otherwise one digit per plane
  const int plane = blockIdx.y;
  const int odd = g.digits & 1;
  // most significant digit of the plane and the number of digits it carries
  const int d_first = g.fuse ? (odd ? (plane == 0 ? 0 : 2 * plane) : 2 * plane + 1) : plane;
  const int nd = (g.fuse && !(odd && plane == 0)) ? 2 : 1;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3; // rows wm*64, cols wn*64
  const int r32 = lane & 31, h = lane >> 5;

  // LDS-DMA: a stage is 48 pieces of 1 KiB (0-15: A rows 8p.., 16-47: B rows 8(p-16)..); wave w moves pieces 6w..6w+5
  const int8_t *src[6];
  int dst[6];
#define I8P_INIT_SRC(DIGIT)                                                                                       \
  do {                                                                                                            \
    _Pragma("unroll") for (int j = 0; j < 6; ++j) {                                                               \
      const int p = 6 * wave + j;                                                                                 \
      const bool isA = p < 16;                                                                                    \
      const int row = 8 * (isA ? p : p - 16) + (lane >> 3);                                                       \
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);                                                            \
      const int8_t *base = isA ? g.A + ((long)tm * I8P_BM + row) * g.ldk                                          \
                               : g.Bt + (long)(DIGIT) * g.strideB + ((long)tn * I8_BN + row) * g.ldk;             \
      src[j] = base + 16 * chunk;                                                                                 \
      dst[j] = p * 1024;                                                                                          \
    }                                                                                                             \
  } while (0)
  // fragment byte offsets inside a stage, K-step ks: logical chunk 2 ks + h
  int fa[4], fb[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int sw = ((2 * ks + h) ^ ((r32 >> 1) & 7)) << 4;
    fa[ks] = (wm * 64 + r32) * 128 + sw;
    fb[ks] = 16384 + (wn * 64 + r32) * 128 + sw;
  }

  i32x16 accg[2][2], accm[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accg[i][j][r] = 0; accm[i][j][r] = 0; }

  // two fragment register sets (X, Y): raw A (2 blocks), masked G / M operands, B (2 blocks)
  i32x4 xa[2], xg[2], xm[2], xb[2], ya[2], yg[2], ym[2], yb[2];
  const i32x4 mask_g = {0x03030303, 0x03030303, 0x03030303, 0x03030303};
  const i32x4 mask_m = {0x01010101, 0x01010101, 0x01010101, 0x01010101};

#define I8P_DMA(j, SOFF)                                                                                          \
  do {                                                                                                            \
    __builtin_amdgcn_global_load_lds((gemma_gptr_t)src[j], (gemma_lptr_t)(i8lds + (SOFF) + dst[j]), 16, 0, 0);    \
    src[j] += I8_BK;                                                                                              \
  } while (0)
// fragment read q of K-step KS from stage offset SOFF: q 0,1 = A blocks, 2,3 = B blocks
#define I8P_READ(q, SOFF, KS, RA, RB)                                                                             \
  do {                                                                                                            \
    if ((q) < 2) RA[(q)&1] = *reinterpret_cast<const i32x4 *>(i8lds + (SOFF) + fa[KS] + ((q)&1) * 4096);         \
    else RB[(q)&1] = *reinterpret_cast<const i32x4 *>(i8lds + (SOFF) + fb[KS] + ((q)&1) * 4096);                  \
  } while (0)
#define I8P_MASK(i, RA, RG, RM)                                                                                   \
  do {                                                                                                            \
    RG[i] = RAW ? RA[i] : (RA[i] & mask_g);                                                                       \
    if (WITH_M) RM[i] = (RA[i] >> 4) & mask_m;                                                                    \
  } while (0)
// MFMA q of a K-step: block (i, j) = (q >> 2, (q >> 1) & 1), q & 1: 0 = G, 1 = M
#define I8P_MF(q, RG, RM, RB)                                                                                     \
  do {                                                                                                            \
    if (((q)&1) == 0)                                                                                             \
      accg[(q) >> 2][((q) >> 1) & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(RG[(q) >> 2], RB[((q) >> 1) & 1],    \
                                                                             accg[(q) >> 2][((q) >> 1) & 1], 0, 0, 0); \
    else if (WITH_M)                                                                                              \
      accm[(q) >> 2][((q) >> 1) & 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(RM[(q) >> 2], RB[((q) >> 1) & 1],    \
                                                                             accm[(q) >> 2][((q) >> 1) & 1], 0, 0, 0); \
  } while (0)
// one K-step on set (CG, CM, CB): MFMAs 0-7; reads of K-step NKS of stage NS into (NA, NB) behind MFMAs 0-3, their
// masks behind MFMAs 5-6; LDS-DMA pieces D0..D0+2 into stage DS behind MFMAs 4-6 when DMA is set
#define I8P_STEP(CG, CM, CB, NS, NKS, NA, NG, NM, NB, DMA, D0, DS)                                                \
  do {                                                                                                            \
    _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {                                                            \
      I8P_MF(q_, CG, CM, CB);                                                                                     \
      if (q_ < 4) I8P_READ(q_, NS, NKS, NA, NB);                                                                  \
      if ((DMA) && q_ >= 4 && q_ < 7) I8P_DMA((D0) + q_ - 4, DS);                                                 \
      if (q_ == 5) I8P_MASK(0, NA, NG, NM);                                                                       \
      if (q_ == 6) I8P_MASK(1, NA, NG, NM);                                                                       \
      GEMMA_SB();                                                                                                 \
    }                                                                                                             \
  } while (0)
// one K-tile from stage SC; MORE: tile t+1 exists in stage SN; LOAD2: tile t+2 exists and goes to stage SD
#define I8P_KTILE(SC, SN, SD, MORE, LOAD2)                                                                        \
  do {                                                                                                            \
    I8P_STEP(xg, xm, xb, SC, 1, ya, yg, ym, yb, LOAD2, 0, SD);                                                    \
    I8P_STEP(yg, ym, yb, SC, 2, xa, xg, xm, xb, LOAD2, 3, SD);                                                    \
    I8P_STEP(xg, xm, xb, SC, 3, ya, yg, ym, yb, false, 0, SD);                                                    \
    I8P_MF(0, yg, ym, yb); GEMMA_SB();                                                                            \
    I8P_MF(1, yg, ym, yb); GEMMA_SB();                                                                            \
    /* tile t+1 must have landed; still in flight: the 6 pieces of tile t+2 issued during this tile */            \
    if (LOAD2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                   \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                         \
    __builtin_amdgcn_s_barrier();                                                                                 \
    GEMMA_SB();                                                                                                   \
    _Pragma("unroll") for (int q_ = 2; q_ < 8; ++q_) {                                                            \
      I8P_MF(q_, yg, ym, yb);                                                                                     \
      if (MORE) {                                                                                                 \
        if (q_ == 2) I8P_READ(0, SN, 0, xa, xb);                                                                  \
        if (q_ == 3) I8P_READ(1, SN, 0, xa, xb);                                                                  \
        if (q_ == 4) I8P_READ(2, SN, 0, xa, xb);                                                                  \
        if (q_ == 5) I8P_READ(3, SN, 0, xa, xb);                                                                  \
        if (q_ == 6) I8P_MASK(0, xa, xg, xm);                                                                     \
        if (q_ == 7) I8P_MASK(1, xa, xg, xm);                                                                     \
      }                                                                                                           \
      GEMMA_SB();                                                                                                 \
    }                                                                                                             \
  } while (0)

  const int nk = g.nk;
  for (int dd = 0; dd < nd; ++dd) {
    if (dd > 0) { // second digit of a fused pair: acc = 256 * C_hi, then accumulate C_lo on top
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) { accg[i][j][r] <<= 8; if (WITH_M) accm[i][j][r] <<= 8; }
    }
    I8P_INIT_SRC(d_first - dd);
    // prologue: tiles 0 and 1 in flight, tile 0 landed (every LDS read of the previous digit completed before its
    // last barrier, so the stages are free)
#pragma unroll
    for (int j = 0; j < 6; ++j) I8P_DMA(j, 0);
    if (nk > 1) {
#pragma unroll
      for (int j = 0; j < 6; ++j) I8P_DMA(j, I8P_STAGE);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    GEMMA_SB();
#pragma unroll
    for (int q = 0; q < 4; ++q) I8P_READ(q, 0, 0, xa, xb);
    I8P_MASK(0, xa, xg, xm);
    I8P_MASK(1, xa, xg, xm);

    int sc = 0, sn = I8P_STAGE, sd = 2 * I8P_STAGE; // stage byte offsets: current, next, DMA target
    int kt = 0;
    for (; kt + 2 < nk; ++kt) {
      I8P_KTILE(sc, sn, sd, true, true);
      const int tmp = sc; sc = sn; sn = sd; sd = tmp;
    }
    if (nk >= 2) {
      I8P_KTILE(sc, sn, sd, true, false);
      const int tmp = sc; sc = sn; sn = sd; sd = tmp;
    }
    I8P_KTILE(sc, sn, sd, false, false);
  }
#undef I8P_INIT_SRC
#undef I8P_DMA
#undef I8P_READ
#undef I8P_MASK
#undef I8P_MF
#undef I8P_STEP
#undef I8P_KTILE

  int *Cg = g.C + (long)plane * g.strideC;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long col = (long)tn * I8_BN + wn * 64 + j * 32 + r32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long row = (long)tm * I8P_BM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        Cg[row * g.ldc + col] = accg[i][j][r];
        if (WITH_M) Cg[(g.m_row0 + row) * g.ldc + col] = accm[i][j][r];
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

// colmax_bits[j] -> the column's fixed-point scale: V = llrint(U * q_j) is cut into `digits` balanced base-256 digits and an integer
// sum over a column is worth qinv_j in U's units.
//   exact_max = 0 (rounds 1-5, GEMMA_HIP_I8_SCALE=pow2): q_j = 2^(8 D - 2 - e_j) with max|u| < 2^e_j -- |V| <= 2^(8 D - 2), half of what
//     the digits can hold, and up to another factor of two lost between the maximum and the next power of two;
//   exact_max = 1 (round 6, default): q_j = 0.99 * 2^(8 D - 1) / max|u| -- the largest entry of the column sits at 0.99 of the largest
//     magnitude D balanced digits represent (127/255 * (256^D - 1) = 0.996 * 2^(8 D - 1)): U is rounded at 1.01 * 2^(-8 D) of each
//     column's maximum instead of 2^(-8 D + 1) .. 2^(-8 D + 2) of it -- one to two bits for nothing but a multiplication by a
//     non-power-of-two when the integer sums are turned back into doubles (one more rounding at 2^-53, relative).
__global__ void u_scale_kernel(const unsigned long long *__restrict__ colmax_bits, long n, int digits, int exact_max,
                               double *__restrict__ q, double *__restrict__ qinv) {
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const double m = __longlong_as_double((long long)colmax_bits[j]);
  const bool finite_pos = m > 0.0 && m <= DBL_MAX;
  if (exact_max && finite_pos) {
    const double L = 0.99 * ldexp(1.0, 8 * digits - 1);
    q[j] = L / m;
    qinv[j] = m / L;
  } else {
    int e = 0;
    if (finite_pos) (void)frexp(m, &e);
    q[j] = ldexp(1.0, i8_scale_bits(digits) - e);
    qinv[j] = ldexp(1.0, e - i8_scale_bits(digits));
  }
}

// 32 x 32 tile of U (rows k, cols j) -> digit tiles [j][k]
__global__ __launch_bounds__(256) void u_digits_kernel(const double *__restrict__ U, long n, long ld,
                                                       const double *__restrict__ q, int8_t *__restrict__ Bt, long ldk,
                                                       long strideB, int digits) {
  __shared__ long long tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
  const long k0 = (long)blockIdx.y * 32, j0 = (long)blockIdx.x * 32;
  for (int r = ty; r < 32; r += 8) {
    const long k = k0 + r, j = j0 + tx;
    long long v = 0;
    if (k < n && j < n) v = llrint(U[k * ld + j] * q[j]); // q a power of two: exactly the ldexp of rounds 1-5
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const long j = j0 + r, k = k0 + tx; // write row j, column k (k contiguous across tx)
    long long v = tile[tx][r];
    for (int d = 0; d < digits; ++d) {
      const int dig = (int)(int8_t)(unsigned char)(v & 0xff); // low byte as a signed digit
      v = (v - dig) >> 8;                                      // exact: v - dig is a multiple of 256
      Bt[(long)d * strideB + j * ldk + k] = (int8_t)dig;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// PLINK rows -> packed left factor rows (byte = g | m << 4) and mean_s
struct IngestI8Args {
  const unsigned char *src;
  long ld, l;
  const int *idx_map;
  int n;
  int8_t *A;  // l x ldk, byte = g | (m << 4)
  long ldk;
  double *mean;
};
__global__ __launch_bounds__(256) void ingest_i8_kernel(IngestI8Args g) {
  const int lane = threadIdx.x & 63;
  const long s = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= g.l) return;
  const unsigned char *bs = g.src + s * g.ld;
  int8_t *gr = g.A + s * g.ldk;
  if (!g.idx_map && (reinterpret_cast<uintptr_t>(bs) & 3) == 0) {
    // every individual analysed (no indicator mapping): a lane turns one 32-bit word = 16 calls into one 16-byte store, the sums
    // as integers (the byte-per-lane loop below cost 0.6 ms per 20 000 x 20 000 block, this one 0.1).  Code c -> byte
    // (0x00011002 >> 8 c) & 0xFF: 0 -> 2, 1 -> 16 (missing), 2 -> 1, 3 -> 0.
    const long nbytes = ((long)g.n + 3) / 4;
    int itot = 0, imiss = 0;
    for (long k = lane; k < g.ldk / 16; k += 64) {
      const long i0 = 16 * k;
      unsigned out[4] = {0u, 0u, 0u, 0u};
      if (i0 < g.n) {
        const int nvalid = (g.n - i0 < 16) ? (int)(g.n - i0) : 16;
        unsigned w;
        if (i0 / 4 + 4 <= nbytes) {
          w = *reinterpret_cast<const unsigned *>(bs + i0 / 4);
        } else {
          w = 0u;
          for (int b = 0; b < 4; ++b)
            if (i0 / 4 + b < nbytes) w |= (unsigned)bs[i0 / 4 + b] << (8 * b);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const unsigned c = (w >> (2 * q)) & 3u;
          const unsigned byte = (q < nvalid) ? ((0x00011002u >> (8 * c)) & 0xFFu) : 0u;
          out[q >> 2] |= byte << (8 * (q & 3));
          itot += (int)(byte & 3u);
          imiss += (int)(byte >> 4);
        }
      }
      *reinterpret_cast<uint4 *>(gr + i0) = make_uint4(out[0], out[1], out[2], out[3]);
    }
    const double tot = wsum((double)itot), cnt = wsum((double)(0 - imiss)) + (double)g.n;
    if (lane == 0) g.mean[s] = tot / cnt;
    return;
  }
  double tot = 0.0, cnt = 0.0;
  for (int i = lane; i < g.n; i += 64) {
    const int p = g.idx_map ? g.idx_map[i] : i;
    bool miss;
    const double v = plink_value((bs[p >> 2] >> (2 * (p & 3))) & 3u, miss);
    gr[i] = miss ? (int8_t)16 : (int8_t)(int)v;
    if (!miss) { tot += v; cnt += 1.0; }
  }
  for (long i = g.n + lane; i < g.ldk; i += 64) gr[i] = 0; // K padding
  tot = wsum(tot);
  cnt = wsum(cnt);
  if (lane == 0) g.mean[s] = tot / cnt; // x_total / (ni_test - n_miss), as ingest_lmm_kernel
}

// fp64 SNP-major rows (BIMBAM / the reference's Xlarge after transposition) -> packed left factor, when the row is a
// hard-call row: every value is 0, 1 or 2 except one repeated "other" value.  nan_missing = 1: the other value must be
// NaN (missing; mean_s = sum / count of the calls, as ingest_lmm_kernel); nan_missing = 0: the input is already
// mean-imputed (src/lmm.cpp:1590-1618) and the other value must be ONE finite number v (mean_s = v, bit for bit).
// Any row that is not of that form clears *all_hard (the batch then takes the fp64 GEMM).
struct PackF64Args {
  const double *src; // l x ld
  long ld, l;
  int n;
  int nan_missing;
  int8_t *A;
  long ldk;
  double *mean;
  int *all_hard; // [0] every row is a hard-call row; [1] every row is a fixed-point dosage row with 3 decimals (values k / 1000
                 // in [0, 2], plus the missing marker: NaN, or with nan_missing = 0 one repeated other value); [2] ... with 2
                 // decimals (k / 100); [3] set when any row has a missing entry
};
// v is EXACTLY the double a decimal "d.ddd" parses to: the correctly rounded k / 1000 for an integer 0 <= k <= 2000 (IEEE
// division is correctly rounded, as is atof); *q = k.  A value that merely lies close to the grid is not a dosage and keeps
// the batch on the fp64 GEMM.
__device__ __forceinline__ bool dosage_on_grid(double v, int *q) {
  const double r = rint(v * 1000.0);
  *q = (int)r;
  return r >= 0.0 && r <= 2000.0 && v == r / 1000.0;
}
__global__ __launch_bounds__(256) void pack_f64_kernel(PackF64Args g) {
  const int lane = threadIdx.x & 63;
  const long s = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= g.l) return;
  const double *xs = g.src + s * g.ld;
  int8_t *gr = g.A + s * g.ldk;
  double tot = 0.0, cnt = 0.0, vfirst = 0.0, dfirst = 0.0;
  int n_nan = 0, have = 0, bad = 0, dhave = 0, dbad = 0, not100 = 0;
  for (int i = lane; i < g.n; i += 64) {
    const double v = xs[i];
    if (!isnan(v)) { // dosage verdict: on the 1/1000 grid, or the row's one repeated off-grid value (an imputed mean)
      int q;
      if (dosage_on_grid(v, &q)) {
        not100 |= (q % 10) != 0;
      } else if (!dhave) {
        dfirst = v;
        dhave = 1;
      } else if (v != dfirst) {
        dbad = 1;
      }
    }
    int8_t b;
    if (v == 0.0 || v == 1.0 || v == 2.0) {
      b = (int8_t)(int)v;
      tot += v;
      cnt += 1.0;
    } else {
      b = 16;
      if (isnan(v)) {
        ++n_nan;
      } else if (!have) {
        vfirst = v;
        have = 1;
      } else if (v != vfirst) {
        bad = 1;
      }
    }
    gr[i] = b;
  }
  for (long i = g.n + lane; i < g.ldk; i += 64) gr[i] = 0;
  tot = wsum(tot);
  cnt = wsum(cnt);
  // one finite "other" value for the whole row: take the lowest lane that saw one
  const unsigned long long hv = __ballot(have != 0);
  double v0 = 0.0;
  if (hv) {
    const int src_lane = __ffsll((long long)hv) - 1;
    v0 = __shfl(vfirst, src_lane, 64);
    if (have && vfirst != v0) bad = 1;
  }
  const bool any_nan = __ballot(n_nan != 0) != 0;
  bool row_bad = __ballot(bad != 0) != 0;
  if (g.nan_missing) {
    if (hv) row_bad = true; // a finite non-call value: dosage data
  } else {
    if (any_nan) row_bad = true;
  }
  // dosage verdict of the row
  const unsigned long long dv = __ballot(dhave != 0);
  if (dv) {
    const double d0 = __shfl(dfirst, __ffsll((long long)dv) - 1, 64);
    if (dhave && dfirst != d0) dbad = 1;
  }
  bool drow_bad = __ballot(dbad != 0) != 0;
  if (g.nan_missing ? dv != 0 : any_nan) drow_bad = true; // NaN input: no finite off-grid value; imputed input: no NaN
  const bool drow_not100 = __ballot(not100 != 0) != 0;
  const bool drow_missing = g.nan_missing ? any_nan : dv != 0;
  if (lane == 0) {
    g.mean[s] = g.nan_missing ? tot / cnt : v0;
    if (row_bad) atomicAnd(g.all_hard, 0);
    if (drow_bad) atomicAnd(g.all_hard + 1, 0);
    if (drow_bad || drow_not100) atomicAnd(g.all_hard + 2, 0);
    if (drow_missing) atomicOr(g.all_hard + 3, 1);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Fixed-point dosages (BIMBAM mean genotypes: "0.98, 0.04, 1.00", doc/manual.tex:398-404) on the int8 pipe.  With scale S = 100
// or 1000 a dosage is x = 1 + q' / S, q' = round(S x) - S in [-S, S]: ONE signed byte for S = 100, two balanced base-256 bytes
// for S = 1000 (q' = 256 a1 + a0).  With m the 0/1 mask of missing entries (q' = 0 there) and mean_s the SNP's mean,
//     (U^T x_s)[j] = ( sum_k q'_sk U[k][j] ) / S  +  sum_k U[k][j]  +  (mean_s - 1) sum_k m_sk U[k][j]:
// integer left factors again, one dense int8 product per byte plane and digit of U, the column sums of U once per setup
// (from the same digits).  The products run on i8gemm_packed_kernel_t<false, true>, one int32 plane per digit (|sum| <=
// 128 * 128 * n does not leave room to fuse two).
struct PackDosageArgs {
  const double *src; // l x ld
  long ld, l;
  int n;
  int nan_missing, two; // two: S = 1000 (planes a0, a1), else S = 100 (plane a0)
  int8_t *A0, *A1, *Am; // l x ldk each; Am may be null when the batch has no missing entry
  long ldk;
  double *mean;         // nan_missing: mean of the present entries; else the row's off-grid value (or 1.0 when it has none)
};
__global__ __launch_bounds__(256) void pack_dosage_kernel(PackDosageArgs g) {
  const int lane = threadIdx.x & 63;
  const long s = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= g.l) return;
  const double *xs = g.src + s * g.ld;
  int8_t *r0 = g.A0 + s * g.ldk, *r1 = g.two ? g.A1 + s * g.ldk : nullptr, *rm = g.Am ? g.Am + s * g.ldk : nullptr;
  double tot = 0.0, cnt = 0.0, other = 1.0;
  int have = 0;
  for (int i = lane; i < g.n; i += 64) {
    const double v = xs[i];
    int q = 0;
    const bool present = !isnan(v) && dosage_on_grid(v, &q);
    int qc = 0, m = 0;
    if (present) {
      qc = g.two ? q - 1000 : q / 10 - 100;
      tot += v;
      cnt += 1.0;
    } else {
      m = 1;
      if (!isnan(v)) { other = v; have = 1; }
    }
    const int a1 = (qc + 128) >> 8; // floor: balanced digits, a0 in [-128, 127]
    r0[i] = (int8_t)(qc - 256 * a1);
    if (r1) r1[i] = (int8_t)a1;
    if (rm) rm[i] = (int8_t)m;
  }
  for (long i = g.n + lane; i < g.ldk; i += 64) { // K padding
    r0[i] = 0;
    if (r1) r1[i] = 0;
    if (rm) rm[i] = 0;
  }
  tot = wsum(tot);
  cnt = wsum(cnt);
  const unsigned long long hv = __ballot(have != 0);
  if (hv) other = __shfl(other, __ffsll((long long)hv) - 1, 64);
  if (lane == 0) g.mean[s] = g.nan_missing ? tot / cnt : other;
}

// colsum[j] = qinv_j * sum_d 256^d sum_k D_d[j][k]: the column sums of U as the digit planes hold it; one
// wavefront per column
__global__ __launch_bounds__(256) void u_digit_colsum_kernel(const int8_t *__restrict__ Bt, long ldk, long strideB,
                                                             const double *__restrict__ qinv, long n, int digits,
                                                             double *__restrict__ colsum) {
  const int lane = threadIdx.x & 63;
  const long j = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= n) return;
  double t = 0.0;
  for (int d = digits - 1; d >= 0; --d) {
    const int8_t *row = Bt + (long)d * strideB + j * ldk;
    int acc = 0;
    for (long k = lane; k < ldk; k += 64) acc += row[k];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    t = t * 256.0 + (double)acc;
  }
  if (lane == 0) colsum[j] = t * qinv[j];
}

// UtX[s][j] = qinv_j * ( T0 / S + (mean_s - 1) TM ) + colsum[j],  T0 = sum_d 256^d (C0_d + 256 C1_d),
// TM = sum_d 256^d CM_d; planes: digit d of byte plane a at C + (a * digits + d) * strideC (a = 0: a0, 1: a1 if two, last: mask
// if have_m)
__global__ __launch_bounds__(256) void i8_combine_dosage_kernel(const int *__restrict__ C, long ldc, long strideC,
                                                                const double *__restrict__ mean, const double *__restrict__ qinv,
                                                                const double *__restrict__ colsum, long l, long n,
                                                                double *__restrict__ UtX, long ldx, int digits, int two,
                                                                int have_m, double inv_scale_is_S) {
  const long j = (long)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const double qi = qinv[j];
  const double cs = colsum[j];
  const int *Cm = C + (long)((two ? 2 : 1) * digits) * strideC;
  for (long s = blockIdx.y; s < l; s += gridDim.y) {
    double t0 = 0.0, t1 = 0.0, tm = 0.0;
    for (int d = digits - 1; d >= 0; --d) {
      t0 = t0 * 256.0 + (double)C[(long)d * strideC + s * ldc + j];
      if (two) t1 = t1 * 256.0 + (double)C[(long)(digits + d) * strideC + s * ldc + j];
      if (have_m) tm = tm * 256.0 + (double)Cm[(long)d * strideC + s * ldc + j];
    }
    double v = (t0 + 256.0 * t1) / inv_scale_is_S;
    if (have_m) v = fma(mean[s] - 1.0, tm, v);
    UtX[s * ldx + j] = v * qi + cs;
  }
}

// UtX[s][j] = qinv_j * sum_d 256^d (CG_d[s][j] + mean_s * CM_d[s][j]);  planes: one per digit (fuse = 0) or
// two digits per plane with 256 * C_{d+1} + C_d (fuse = 1; with an odd digit count plane 0 holds digit 0 alone)
// sur_cnt / sur_list (may be null): the calls the 2:4 sparse mask operand dropped, per row (i8gemm_sparse.hip.h:
// i8_surplus_list_kernel); rows with 1 .. 16 of them get mean_s * sum_e U[i_e][j] added here, in list order
// Four columns per thread (one 16-byte load per plane and operand, two 16-byte stores): a block moves 32 KB instead of 8 -- the
// pass is a plain HBM stream (9.6 GB of planes in, 3.2 GB out at n = B = 20 000) and was short of bytes in flight with one column
// per thread (2.8 ms = 4.6 TB/s).  Needs ldc % 4 == 0 and 16-byte aligned planes / UtX rows (ldx even): the library's buffers.
__global__ __launch_bounds__(256) void i8_combine_kernel(const int *__restrict__ C, long ldc, long strideC, long m_row0,
                                                         const double *__restrict__ mean, const double *__restrict__ qinv,
                                                         long l, long n, double *__restrict__ UtX, long ldx,
                                                         double m_scale, int fuse, int digits,
                                                         const int *__restrict__ sur_cnt = nullptr,
                                                         const int *__restrict__ sur_list = nullptr,
                                                         const double *__restrict__ U = nullptr, long ldu = 0,
                                                         int m_skip0 = 0 /* 1: plane 0 carries no mask product (7g6m) */,
                                                         const int *__restrict__ anymiss = nullptr /* Sparse2Args::anymiss: 0 = no
                                                         plane of this block carries one (their M rows were not written) */) {
  const long j = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (j >= n) return;
  const bool no_mask = anymiss != nullptr && *anymiss == 0; // uniform; a mask product of zeros adds +0.0: the same doubles
  const int nplanes = fuse ? (digits + 1) / 2 : digits;
  const int odd = digits & 1;
  const int nv = (n - j < 4) ? (int)(n - j) : 4; // columns of this thread inside the row (the planes are padded past n)
  double qi4[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) qi4[c] = qinv[c < nv ? j + c : j];
  for (long s = blockIdx.y; s < l; s += gridDim.y) { // gridDim.y is capped at 65535 rows per sweep
    double tg[4] = {0.0, 0.0, 0.0, 0.0}, tmk[4] = {0.0, 0.0, 0.0, 0.0};
    for (int q = nplanes - 1; q >= 0; --q) {
      // fused: plane q sits two digits above plane q - 1, except that an odd count leaves plane 0 one digit wide
      const double w = fuse ? ((q == 0 && odd) ? 256.0 : 65536.0) : 256.0;
      const int4 cg = *reinterpret_cast<const int4 *>(C + (long)q * strideC + s * ldc + j);
      const int4 cm = (no_mask || (m_skip0 && q == 0)) ? make_int4(0, 0, 0, 0)
                                          : *reinterpret_cast<const int4 *>(C + (long)q * strideC + (m_row0 + s) * ldc + j);
      const int g4[4] = {cg.x, cg.y, cg.z, cg.w}, m4[4] = {cm.x, cm.y, cm.z, cm.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        tg[c] = tg[c] * w + (double)g4[c];
        tmk[c] = tmk[c] * w + (double)m4[c];
      }
    }
    const double ms = mean[s] * m_scale; // m_scale: exact power of two
    double v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = fma(ms, tmk[c], tg[c]) * qi4[c];
    if (sur_cnt) {
      const int cnt = sur_cnt[s]; // uniform over the block: no divergence
      if (cnt > 0) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int e = 0; e < cnt; ++e) {
          const double *ur = U + (long)sur_list[s * SUR_MAX + e] * ldu + j;
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (c < nv) acc[c] += ur[c];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] += mean[s] * acc[c];
      }
    }
    double *o = UtX + s * ldx + j;
    if (nv == 4) {
      typedef double cmb_v2 __attribute__((ext_vector_type(2)));
      reinterpret_cast<cmb_v2 *>(o)[0] = cmb_v2{v[0], v[1]};
      reinterpret_cast<cmb_v2 *>(o)[1] = cmb_v2{v[2], v[3]};
    } else {
      for (int c = 0; c < nv; ++c) o[c] = v[c];
    }
  }
}

} // namespace gemma_hip
