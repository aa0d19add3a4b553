// i8gemm_packed_kernel_t with the missing-mask product on the 2:4 structured-sparse MFMA (v_smfmac_i32_32x32x64_i8).
//
// The mask operand is 99 % zeros: a group of four consecutive individuals holds more than two missing calls with probability
// 4e-6 at 1 % missingness.  Per pair of K-steps (64 bytes of K) the mask product then takes ONE sparse instruction per 32 x 32
// block instead of two dense ones -- 12 instead of 16 matrix instructions per pair, and the sparse one costs no more than a
// dense one (profiles/r02_smfmac_i8_rate.txt).  Operand layout as decoded by scripts/smfmac_probe.hip and checked end to end
// by scripts/i8_sparse_proto.hip:
//   * lane (row r, half h) of the sparse instruction of the pair (ks, ks+1) covers the 32 individuals of K-step ks + h in their
//     natural order; its operand is described by one 8-byte word -- eight index nibbles (positions of the first and second
//     missing call of each group of four) and sixteen "kept" bits (at least one / at least two missing calls per group); the
//     sixteen kept bytes are spread from the bits with one 24-bit multiply per four bytes;
//   * the B operand is the pair of dense B fragments the lane holds for the two K-steps;
//   * a group with three or four missing calls keeps its first two: the surplus calls are the caller's to add (fp64 fix-up
//     per flagged row), see sparse_meta_kernel.
// The words of a K-tile (two pairs, 32 bytes per row: 4 KiB per stage) travel like the operands: LDS-DMA two K-tiles ahead (one more
// 1 KiB piece per wavefront), one ds_read_b128 per row block and tile.  A first version fetched them into registers with plain
// global loads and rotated three register sets: the copy at the end of a tile waited for a load issued half a tile earlier.
#pragma once
#include "i8gemm.hip.h"

namespace gemma_hip {

typedef int i32x8 __attribute__((ext_vector_type(8)));
constexpr int SP_STAGE = I8P_STAGE + 4096; // operands + the mask words of 128 rows (32 bytes each)

struct SparseMeta {
  const uint4 *m4;   // [row][tile][h]: {idx pair 0, bits pair 0, idx pair 1, bits pair 1} of K-steps 2 p + h
  int *row_surplus;  // per row: number of calls the sparse operand drops (0 for almost every row)
  long ntiles;
};

// one thread per (row, tile, h)
__global__ __launch_bounds__(256) void sparse_meta_kernel(const int8_t *__restrict__ A, long lpad, long ldk, uint4 *__restrict__ m4,
                                                          int *__restrict__ row_surplus) {
  const long nk = ldk / I8_BK;
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= lpad * nk * 2) return;
  const long row = id / (nk * 2), rem = id % (nk * 2), tile = rem >> 1;
  const int h = (int)(rem & 1);
  unsigned w[4];
  int surplus = 0;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int8_t *src = A + row * ldk + tile * I8_BK + 32 * (2 * p + h);
    unsigned idx = 0, bits = 0;
    for (int gq = 0; gq < 8; ++gq) {
      const unsigned word = *reinterpret_cast<const unsigned *>(src + 4 * gq);
      const unsigned m = (word >> 4) & 0x01010101u;
      const unsigned pat = (m | (m >> 7) | (m >> 14) | (m >> 21)) & 0xFu; // bit q = call q of the group is missing
      const int cnt = __popc(pat);
      const int p0 = cnt >= 1 ? __ffs(pat) - 1 : 0;
      const unsigned rest = pat & (pat - 1);
      const int p1 = cnt >= 2 ? __ffs(rest) - 1 : (p0 == 3 ? 2 : 3);
      idx |= (unsigned)(p0 | (p1 << 2)) << (4 * gq);
      bits |= (unsigned)((cnt >= 1) | ((cnt >= 2) << 1)) << (2 * gq);
      surplus += cnt > 2 ? cnt - 2 : 0;
    }
    w[2 * p] = idx;
    w[2 * p + 1] = bits;
  }
  m4[id] = make_uint4(w[0], w[1], w[2], w[3]);
  if (surplus) atomicAdd(row_surplus + row, surplus); // integer count: order-independent
}

static inline int sparse_meta_build(const int8_t *A, long lpad, long ldk, SparseMeta *sm) {
  const long nk = ldk / I8_BK, total = lpad * nk * 2;
  uint4 *m4 = nullptr;
  int *rs = nullptr;
  if (hipMalloc(reinterpret_cast<void **>(&m4), (size_t)total * sizeof(uint4)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&rs), (size_t)lpad * sizeof(int)) != hipSuccess)
    return 1;
  (void)hipMemset(rs, 0, (size_t)lpad * sizeof(int));
  hipLaunchKernelGGL(sparse_meta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, A, lpad, ldk, m4, rs);
  sm->m4 = m4;
  sm->row_surplus = rs;
  sm->ntiles = nk;
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

// Rows whose sparse mask operand dropped calls (row_surplus > 0: a group of four with three or four missing calls keeps its
// first two): UtX[s][k] += mean_s * sum over the dropped individuals i of U[i][k], in fp64, after the digits were combined.
// Deterministic: the dropped individuals of 256 groups at a time are listed in group order by one thread, every thread k adds
// them in that order.  ~2 % of the rows at 1 % missingness, one or two individuals each.
__global__ __launch_bounds__(256) void i8_surplus_fix_kernel(const int8_t *__restrict__ A, long ldk,
                                                             const int *__restrict__ row_surplus,
                                                             const double *__restrict__ mean, const double *__restrict__ U,
                                                             long ldu, long n, long l, double *__restrict__ UtX, long ldx,
                                                             int skip_upto) {
  const long s = blockIdx.x;
  if (s >= l || row_surplus[s] <= skip_upto) return; // rows with up to skip_upto dropped calls were completed by the combine
  __shared__ int found[256][2], list[512], nlist;
  const int t = threadIdx.x;
  const double mu = mean[s];
  const long ngroups = ldk / 4;
  for (long g0 = 0; g0 < ngroups; g0 += 256) {
    const long gq = g0 + t;
    int f0 = -1, f1 = -1;
    if (gq < ngroups) {
      const unsigned word = *reinterpret_cast<const unsigned *>(A + s * ldk + 4 * gq);
      const unsigned m = (word >> 4) & 0x01010101u;
      unsigned pat = (m | (m >> 7) | (m >> 14) | (m >> 21)) & 0xFu;
      if (__popc(pat) > 2) {
        pat &= pat - 1; // drop the first two (kept by the sparse operand)
        pat &= pat - 1;
        f0 = (int)(4 * gq) + __ffs(pat) - 1;
        pat &= pat - 1;
        if (pat) f1 = (int)(4 * gq) + __ffs(pat) - 1;
      }
    }
    if (!__syncthreads_or(f0 >= 0)) continue;
    found[t][0] = f0;
    found[t][1] = f1;
    __syncthreads();
    if (t == 0) {
      int c = 0;
      for (int q = 0; q < 256; ++q)
        for (int e = 0; e < 2; ++e)
          if (found[q][e] >= 0) list[c++] = found[q][e];
      nlist = c;
    }
    __syncthreads();
    const int c = nlist;
    for (long k = t; k < n; k += 256) {
      double acc = 0.0;
      for (int e = 0; e < c; ++e) acc += U[(long)list[e] * ldu + k];
      UtX[s * ldx + k] += mu * acc;
    }
    __syncthreads();
  }
}

// The dropped calls of a row as a short list, so that the digit combine can add them in the same pass (a pass of its own over
// the fp64 rows costs 4.7 ms per 20 000-SNP block at 5 % missingness, where every row has two or three of them): one wavefront
// per row with row_surplus > 0 scans the row's groups of four in order and writes the first SUR_MAX dropped individuals to
// sur_list[row][..] (group order: deterministic), the count to sur_cnt[row]; a row with more (a SNP missing for most
// individuals) keeps cnt = -1 and is left to i8_surplus_fix_kernel.
static_assert(SUR_MAX >= 2, "SUR_MAX (i8gemm.hip.h) is the list stride of this kernel, of i8_combine_kernel and of the host's buffers");
__global__ __launch_bounds__(256) void i8_surplus_list_kernel(const int8_t *__restrict__ A, long ldk, const int *__restrict__ row_surplus,
                                                             long l, int *__restrict__ sur_cnt, int *__restrict__ sur_list) {
  const int lane = threadIdx.x & 63;
  const long s = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= l) return;
  const int want = row_surplus[s];
  if (want == 0 || want > SUR_MAX) {
    if (lane == 0) sur_cnt[s] = want == 0 ? 0 : -1;
    return;
  }
  int have = 0;
  const long ngroups = ldk / 4;
  for (long g0 = 0; g0 < ngroups && have < want; g0 += 64) {
    const long gq = g0 + lane;
    int f0 = -1, f1 = -1;
    if (gq < ngroups) {
      const unsigned word = *reinterpret_cast<const unsigned *>(A + s * ldk + 4 * gq);
      const unsigned m = (word >> 4) & 0x01010101u;
      unsigned pat = (m | (m >> 7) | (m >> 14) | (m >> 21)) & 0xFu;
      if (__popc(pat) > 2) {
        pat &= pat - 1;
        pat &= pat - 1;
        f0 = (int)(4 * gq) + __ffs(pat) - 1;
        pat &= pat - 1;
        if (pat) f1 = (int)(4 * gq) + __ffs(pat) - 1;
      }
    }
    const int mine = (f0 >= 0) + (f1 >= 0);
    int before = mine; // inclusive prefix sum over the lanes (group order)
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(before, o, 64);
      if (lane >= o) before += v;
    }
    const int base = have + before - mine;
    if (f0 >= 0) sur_list[s * SUR_MAX + base] = f0;
    if (f1 >= 0) sur_list[s * SUR_MAX + base + 1] = f1;
    have += __shfl(before, 63, 64);
  }
  if (lane == 0) sur_cnt[s] = have;
}

__device__ __forceinline__ i32x4 sp_expand(unsigned bits) {
  i32x4 v;
#pragma unroll
  for (int d = 0; d < 4; ++d) v[d] = (int)((((bits >> (4 * d)) & 0xFu) * 0x00204081u) & 0x01010101u);
  return v;
}

__global__ __launch_bounds__(512, 2) void i8gemm_sparse_kernel(I8PackArgs g, SparseMeta sm) {
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
  }
  const int plane = blockIdx.y;
  const int odd = g.digits & 1;
  const int d_first = g.fuse ? (odd ? (plane == 0 ? 0 : 2 * plane) : 2 * plane + 1) : plane;
  const int nd = (g.fuse && !(odd && plane == 0)) ? 2 : 1;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3; // rows wm*64, cols wn*64
  const int r32 = lane & 31, h = lane >> 5;

  const int8_t *src[6];
  int dst[6];
#define SP_INIT_SRC(DIGIT)                                                                                        \
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
  int fa[4], fb[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int sw = ((2 * ks + h) ^ ((r32 >> 1) & 7)) << 4;
    fa[ks] = (wm * 64 + r32) * 128 + sw;
    fb[ks] = 16384 + (wn * 64 + r32) * 128 + sw;
  }
  // mask words: piece (wave & 3) of a stage = rows 32 (wave & 3) .. + 31, lane l -> row l / 2, half l & 1 (wavefronts 4-7 repeat
  // the pieces of 0-3: every wavefront issues the same number of vector-memory operations, the counted waits stay uniform)
  const uint4 *msrc;
  const int mdst = I8P_STAGE + (wave & 3) * 1024;
  const int fm0 = I8P_STAGE + (wm * 64 + r32) * 32 + h * 16; // this lane's words of row block 0; block 1: + 32 * 32

  i32x16 accg[2][2], accm[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accg[i][j][r] = 0; accm[i][j][r] = 0; }

  // fragment sets X, Y: raw A, masked G operand, B; sparse operand of the current pair; mask words of tiles t, t+1, t+2
  i32x4 xa[2], xg[2], xb[2], ya[2], yg[2], yb[2], ms[2];
  unsigned mi[2];
  uint4 mc[2];
  const i32x4 mask_g = {0x03030303, 0x03030303, 0x03030303, 0x03030303};

#define SP_DMA(j, SOFF)                                                                                           \
  do {                                                                                                            \
    __builtin_amdgcn_global_load_lds((gemma_gptr_t)src[j], (gemma_lptr_t)(i8lds + (SOFF) + dst[j]), 16, 0, 0);    \
    src[j] += I8_BK;                                                                                              \
  } while (0)
#define SP_DMA_META(SOFF)                                                                                         \
  do {                                                                                                            \
    __builtin_amdgcn_global_load_lds((gemma_gptr_t)msrc, (gemma_lptr_t)(i8lds + (SOFF) + mdst), 16, 0, 0);        \
    msrc += 2;                                                                                                    \
  } while (0)
#define SP_READ_META(SOFF)                                                                                        \
  do {                                                                                                            \
    mc[0] = *reinterpret_cast<const uint4 *>(i8lds + (SOFF) + fm0);                                               \
    mc[1] = *reinterpret_cast<const uint4 *>(i8lds + (SOFF) + fm0 + 1024);                                        \
  } while (0)
#define SP_READ(q, SOFF, KS, RA, RB)                                                                              \
  do {                                                                                                            \
    if ((q) < 2) RA[(q)&1] = *reinterpret_cast<const i32x4 *>(i8lds + (SOFF) + fa[KS] + ((q)&1) * 4096);         \
    else RB[(q)&1] = *reinterpret_cast<const i32x4 *>(i8lds + (SOFF) + fb[KS] + ((q)&1) * 4096);                  \
  } while (0)
#define SP_MASK(i, RA, RG) RG[i] = RA[i] & mask_g
// dense genotype MFMA of block b = (i, j) = (b >> 1, b & 1)
#define SP_G(b, RG, RB)                                                                                           \
  accg[(b) >> 1][(b)&1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(RG[(b) >> 1], RB[(b)&1], accg[(b) >> 1][(b)&1], 0, 0, 0)
// sparse mask MFMA of block b over the pair of K-steps whose B fragments are PB (first step) and CB (second step)
#define SP_S(b, PB, CB)                                                                                           \
  accm[(b) >> 1][(b)&1] = __builtin_amdgcn_smfmac_i32_32x32x64_i8(                                                \
      ms[(b) >> 1], __builtin_shufflevector(PB[(b)&1], CB[(b)&1], 0, 1, 2, 3, 4, 5, 6, 7), accm[(b) >> 1][(b)&1],  \
      (int)mi[(b) >> 1], 0, 0)
// sparse operands of pair P (0 / 1) of the current tile from its mask words
#define SP_OPER(P)                                                                                                \
  do {                                                                                                            \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                            \
      mi[i_] = (P) ? mc[i_].z : mc[i_].x;                                                                         \
      ms[i_] = sp_expand((P) ? mc[i_].w : mc[i_].y);                                                              \
    }                                                                                                             \
  } while (0)
// first K-step of a pair on set (CG, CB): 4 genotype MFMAs; behind them the reads of the next K-step (B fragments first:
// the sparse MFMAs that open the next step need them), LDS-DMA pieces D0..D0+2 when DMA, the G masks of the next step
#define SP_STEP_EVEN(CG, CB, NS, NKS, NA, NG, NB, DMA, D0, DS, META)                                              \
  do {                                                                                                            \
    SP_G(0, CG, CB); SP_READ(2, NS, NKS, NA, NB); GEMMA_SB();                                                     \
    SP_READ(3, NS, NKS, NA, NB); if (DMA) SP_DMA((D0) + 0, DS); GEMMA_SB();                                       \
    SP_G(1, CG, CB); SP_READ(0, NS, NKS, NA, NB); GEMMA_SB();                                                     \
    SP_READ(1, NS, NKS, NA, NB); if (DMA) SP_DMA((D0) + 1, DS); GEMMA_SB();                                       \
    SP_G(2, CG, CB); if (DMA) SP_DMA((D0) + 2, DS);                                                               \
    if (META) SP_DMA_META(DS);                                                                                    \
    GEMMA_SB();                                                                                                   \
    SP_G(3, CG, CB); SP_MASK(0, NA, NG); SP_MASK(1, NA, NG); GEMMA_SB();                                          \
  } while (0)
// second K-step of a pair on set (CG, CB), PB = the first step's B fragments: the 4 sparse MFMAs first (they read PB, which the
// reads of the next step overwrite -- a matrix instruction has read its operands once it is issued), then 4 genotype MFMAs with
// the next step's reads behind them
#define SP_STEP_ODD(CG, CB, PB, NS, NKS, NA, NG, NB, DMA, D0, DS)                                                 \
  do {                                                                                                            \
    SP_S(0, PB, CB); SP_READ(0, NS, NKS, NA, NB); GEMMA_SB(); /* A fragments of the next step: not a sparse operand */ \
    SP_S(1, PB, CB); SP_READ(1, NS, NKS, NA, NB); GEMMA_SB();                                                     \
    SP_S(2, PB, CB); if (DMA) SP_DMA((D0) + 0, DS); GEMMA_SB();                                                   \
    SP_S(3, PB, CB); if (DMA) SP_DMA((D0) + 1, DS); GEMMA_SB();                                                   \
    SP_G(0, CG, CB); SP_READ(2, NS, NKS, NA, NB); GEMMA_SB();                                                     \
    SP_G(1, CG, CB); SP_READ(3, NS, NKS, NA, NB); GEMMA_SB();                                                     \
    SP_G(2, CG, CB); if (DMA) SP_DMA((D0) + 2, DS); SP_MASK(0, NA, NG); GEMMA_SB();                               \
    SP_G(3, CG, CB); SP_MASK(1, NA, NG); GEMMA_SB();                                                              \
  } while (0)
// one K-tile from stage SC (steps 0..3 = pairs 0, 1); MORE: tile t+1 exists in stage SN; LOAD2: tile t+2 exists -> stage SD
#define SP_KTILE(SC, SN, SD, MORE, LOAD2)                                                                         \
  do {                                                                                                            \
    SP_OPER(0);                                                                                                   \
    SP_STEP_EVEN(xg, xb, SC, 1, ya, yg, yb, LOAD2, 0, SD, false);                                                 \
    SP_STEP_ODD(yg, yb, xb, SC, 2, xa, xg, xb, LOAD2, 3, SD);                                                     \
    SP_OPER(1);                                                                                                   \
    SP_STEP_EVEN(xg, xb, SC, 3, ya, yg, yb, false, 0, SD, LOAD2);                                                 \
    SP_S(0, xb, yb); GEMMA_SB();                                                                                  \
    SP_S(1, xb, yb); GEMMA_SB();                                                                                  \
    SP_S(2, xb, yb); GEMMA_SB();                                                                                  \
    SP_S(3, xb, yb); GEMMA_SB();                                                                                  \
    SP_G(0, yg, yb); GEMMA_SB();                                                                                  \
    /* tile t+1 (operands and mask words) must have landed; in flight: the 7 pieces of tile t+2 */                 \
    if (LOAD2) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");                                                   \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                         \
    __builtin_amdgcn_s_barrier();                                                                                 \
    GEMMA_SB();                                                                                                   \
    SP_G(1, yg, yb); if (MORE) { SP_READ(0, SN, 0, xa, xb); SP_READ(1, SN, 0, xa, xb); } GEMMA_SB();              \
    SP_G(2, yg, yb); if (MORE) { SP_READ(2, SN, 0, xa, xb); SP_READ(3, SN, 0, xa, xb); } GEMMA_SB();              \
    SP_G(3, yg, yb); GEMMA_SB();                                                                                  \
    if (MORE) {                                                                                                   \
      SP_READ_META(SN);                                                                                           \
      SP_MASK(0, xa, xg); SP_MASK(1, xa, xg);                                                                     \
    }                                                                                                             \
  } while (0)

  const int nk = g.nk;
  for (int dd = 0; dd < nd; ++dd) {
    if (dd > 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) { accg[i][j][r] <<= 8; accm[i][j][r] <<= 8; }
    }
    SP_INIT_SRC(d_first - dd);
    msrc = sm.m4 + (((long)tm * I8P_BM + 32 * (wave & 3) + (lane >> 1)) * sm.ntiles) * 2 + (lane & 1);
#pragma unroll
    for (int j = 0; j < 6; ++j) SP_DMA(j, 0);
    SP_DMA_META(0);
    if (nk > 1) {
#pragma unroll
      for (int j = 0; j < 6; ++j) SP_DMA(j, SP_STAGE);
      SP_DMA_META(SP_STAGE);
      asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    GEMMA_SB();
#pragma unroll
    for (int q = 0; q < 4; ++q) SP_READ(q, 0, 0, xa, xb);
    SP_READ_META(0);
    SP_MASK(0, xa, xg);
    SP_MASK(1, xa, xg);

    int sc = 0, sn = SP_STAGE, sd = 2 * SP_STAGE;
    int kt = 0;
    for (; kt + 2 < nk; ++kt) {
      SP_KTILE(sc, sn, sd, true, true);
      const int tmp = sc; sc = sn; sn = sd; sd = tmp;
    }
    if (nk >= 2) {
      SP_KTILE(sc, sn, sd, true, false);
      const int tmp = sc; sc = sn; sn = sd; sd = tmp;
    }
    SP_KTILE(sc, sn, sd, false, false);
  }
#undef SP_INIT_SRC
#undef SP_DMA
#undef SP_READ
#undef SP_DMA_META
#undef SP_READ_META
#undef SP_MASK
#undef SP_G
#undef SP_S
#undef SP_OPER
#undef SP_STEP_EVEN
#undef SP_STEP_ODD
#undef SP_KTILE

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
        Cg[(g.m_row0 + row) * g.ldc + col] = accm[i][j][r];
      }
    }
}

} // namespace gemma_hip
