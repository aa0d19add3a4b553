// The dense byte-plane product (i8gemm_packed_kernel_t<false, RAW> of i8gemm.hip.h: one int8 left factor, no mask product) on the
// 16-ROW matrix instruction v_mfma_i32_16x16x64_i8 (round 5).  Two stages run on it: the fixed-point dosage planes
// (RAW = true: signed bytes q' as they are, one int32 plane per digit of U) and G^T G of the integer kinship (RAW = false: packed
// bytes g | m << 4 masked to the genotype, digits = 1).
//
// Why: the chip is at its power limit under these products and the instruction decides how much clock the VALUES cost
// (profiles/r04_mfma_power_probe.txt: 3.70 POP/s sustained on full-range bytes for v_mfma_i32_32x32x32_i8, 4.72 for the 16 x 16 x 64
// form of the same arithmetic -- its K = 64 sum is formed before the 32-bit accumulator is touched).  The records kernel took that
// step at the end of round 4 (49.8 against 54.6 ms); this is the same step for the dense kernel.
//
// Same tiles (128 rows x 256 columns x 128 K-bytes), same LDS images ([row][128 bytes of K], 16-byte chunk index XOR (row >> 1) & 7),
// same three 48 KiB stages, same LDS-DMA two K-tiles ahead with a counted vmcnt and a raw s_barrier as the kernel it replaces; 8
// wavefronts 2 x 4, wave tile 64 x 64 = 4 x 4 blocks of 16 x 16.  Lane (r16 = lane % 16, q = lane / 16) of a fragment holds the 16 K
// bytes 64 P + 16 q .. of row / column r16 of its block for the pair P of K-steps: chunk 4 P + q of the row -- ONE ds_read_b128 per
// (block, pair), 16 per K-tile and wavefront for 32 matrix instructions (the 32-row form: 16 for 16 of twice the length).  The
// accumulator of a block: lane (c16, q) holds rows 4 q + r, column c16.
#pragma once
#include "i8gemm.hip.h"

namespace gemma_hip {

template <bool RAW>
__global__ __launch_bounds__(512, 2) void i8gemm_dense16_kernel_t(I8PackArgs g) {
  extern __shared__ __attribute__((aligned(1024))) int8_t i8lds[];
  int tm, tn;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q_ = nwg >> 3, r = nwg & 7, x = b & 7, o = b >> 3;
    const int L = (x < r ? x * (q_ + 1) : r * (q_ + 1) + (x - r) * q_) + o;
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
  const int plane = blockIdx.y;
  const int odd = g.digits & 1;
  const int d_first = g.fuse ? (odd ? (plane == 0 ? 0 : 2 * plane) : 2 * plane + 1) : plane;
  const int nd = (g.fuse && !(odd && plane == 0)) ? 2 : 1;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3; // rows wm * 64, columns wn * 64
  const int r16 = lane & 15, q = lane >> 4;

  // LDS-DMA: a stage is 48 pieces of 1 KiB (0-15: A rows 8 p .., 16-47: B rows 8 (p - 16) ..); wave w moves pieces 6 w .. 6 w + 5
  const int8_t *src[6];
  int dst[6];
#define D16_INIT_SRC(DIGIT)                                                                                       \
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
  // fragment byte offsets inside a stage for pair P (block i / j: + 2048 i / j; the swizzle does not change with the block)
  int fa[2], fb[2];
#pragma unroll
  for (int P = 0; P < 2; ++P) {
    const int sw = ((4 * P + q) ^ ((r16 >> 1) & 7)) << 4;
    fa[P] = (wm * 64 + r16) * 128 + sw;
    fb[P] = 16384 + (wn * 64 + r16) * 128 + sw;
  }

  i32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (i32x4){0, 0, 0, 0};

  i32x4 xa[4], xb[4], ya[4], yb[4]; // two fragment sets: pair 0 / pair 1 of a K-tile
  const i32x4 mask_g = {0x03030303, 0x03030303, 0x03030303, 0x03030303};

#define D16_DMA(j, SOFF)                                                                                          \
  do {                                                                                                            \
    __builtin_amdgcn_global_load_lds((gemma_gptr_t)src[j], (gemma_lptr_t)(i8lds + (SOFF) + dst[j]), 16, 0, 0);    \
    src[j] += I8_BK;                                                                                              \
  } while (0)
// read e of pair P from stage SOFF: e 0..3 = A blocks, 4..7 = B blocks
#define D16_READ(e, SOFF, P, RA, RB)                                                                              \
  do {                                                                                                            \
    if ((e) < 4) RA[(e)&3] = *reinterpret_cast<const i32x4 *>(i8lds + (SOFF) + fa[P] + ((e)&3) * 2048);           \
    else RB[(e)&3] = *reinterpret_cast<const i32x4 *>(i8lds + (SOFF) + fb[P] + ((e)&3) * 2048);                   \
  } while (0)
#define D16_MASK(i, RA)                                                                                           \
  do {                                                                                                            \
    if (!RAW) RA[i] = RA[i] & mask_g;                                                                             \
  } while (0)
// matrix instruction m of a pair: block (i, j) = (m >> 2, m & 3) -- sixteen different accumulators in a row
#define D16_MF(m, RA, RB)                                                                                         \
  acc[(m) >> 2][(m)&3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(RA[(m) >> 2], RB[(m)&3], acc[(m) >> 2][(m)&3], 0, 0, 0)
// One K-tile from stage SC; at entry (xa, xb) hold pair 0.  Pair 0: the reads of pair 1 behind its first eight instructions, the six
// LDS-DMA pieces of tile t + 2 (stage SD) behind the next six.  Pair 1: four instructions, then the rendezvous for tile t + 1 (its six
// pieces were issued a tile ago: with this tile's six in flight the counted wait is vmcnt(6)), then pair 0 of tile t + 1 behind the rest.
#define D16_KTILE(SC, SN, SD, MORE, LOAD2)                                                                        \
  do {                                                                                                            \
    _Pragma("unroll") for (int m_ = 0; m_ < 16; ++m_) {                                                           \
      D16_MF(m_, xa, xb);                                                                                         \
      if (m_ < 8) D16_READ(m_, SC, 1, ya, yb);                                                                    \
      if ((LOAD2) && m_ >= 8 && m_ < 14) D16_DMA(m_ - 8, SD);                                                     \
      GEMMA_SB();                                                                                                 \
    }                                                                                                             \
    D16_MASK(0, ya); D16_MASK(1, ya); D16_MASK(2, ya); D16_MASK(3, ya);                                           \
    GEMMA_SB();                                                                                                   \
    _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_) {                                                            \
      D16_MF(m_, ya, yb);                                                                                         \
      GEMMA_SB();                                                                                                 \
    }                                                                                                             \
    if (LOAD2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                   \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                         \
    __builtin_amdgcn_s_barrier();                                                                                 \
    GEMMA_SB();                                                                                                   \
    _Pragma("unroll") for (int m_ = 4; m_ < 16; ++m_) {                                                           \
      D16_MF(m_, ya, yb);                                                                                         \
      if ((MORE) && m_ < 12) D16_READ(m_ - 4, SN, 0, xa, xb);                                                     \
      GEMMA_SB();                                                                                                 \
    }                                                                                                             \
    if (MORE) { D16_MASK(0, xa); D16_MASK(1, xa); D16_MASK(2, xa); D16_MASK(3, xa); }                             \
    GEMMA_SB();                                                                                                   \
  } while (0)

  const int nk = g.nk;
  for (int dd = 0; dd < nd; ++dd) {
    if (dd > 0) { // second digit of a fused pair: acc = 256 * C_hi, then C_lo on top
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] <<= 8;
    }
    D16_INIT_SRC(d_first - dd);
#pragma unroll
    for (int j = 0; j < 6; ++j) D16_DMA(j, 0);
    if (nk > 1) {
#pragma unroll
      for (int j = 0; j < 6; ++j) D16_DMA(j, I8P_STAGE);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    GEMMA_SB();
#pragma unroll
    for (int e = 0; e < 8; ++e) D16_READ(e, 0, 0, xa, xb);
    D16_MASK(0, xa); D16_MASK(1, xa); D16_MASK(2, xa); D16_MASK(3, xa);

    int sc = 0, sn = I8P_STAGE, sd = 2 * I8P_STAGE;
    int kt = 0;
    for (; kt + 2 < nk; ++kt) {
      D16_KTILE(sc, sn, sd, true, true);
      const int tmp = sc; sc = sn; sn = sd; sd = tmp;
    }
    if (nk >= 2) {
      D16_KTILE(sc, sn, sd, true, false);
      const int tmp = sc; sc = sn; sn = sd; sd = tmp;
    }
    D16_KTILE(sc, sn, sd, false, false);
  }
#undef D16_INIT_SRC
#undef D16_DMA
#undef D16_READ
#undef D16_MASK
#undef D16_MF
#undef D16_KTILE

  int *Cg = g.C + (long)plane * g.strideC;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long col = (long)tn * I8_BN + wn * 64 + 16 * j + r16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = (long)tm * I8P_BM + wm * 64 + 16 * i + 4 * q + r;
        Cg[row * g.ldc + col] = acc[i][j][r];
      }
    }
}

} // namespace gemma_hip
