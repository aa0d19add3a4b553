// EXPERIMENT (end of round 4; not part of the library): the records kernel of i8gemm_sparse2.hip.h with its GENOTYPE product on
// v_mfma_i32_16x16x64_i8 instead of v_mfma_i32_32x32x32_i8.  Same inputs (records, digit planes), same outputs (G rows and M rows
// of the int32 planes), same LDS stages and LDS-DMA pipeline; wavefronts 8 x 1 (32 rows x 128 columns each).
//
// Why: with operands in registers and nothing else running, the 32x32x32 form sustains 3.70 POP/s on full-range digit values and
// the 16x16x64 form 4.72 (profiles/r04_mfma_power_probe.txt): the kernel is power-limited, and two thirds of its matrix time is this
// product.
//
// Operand layouts (profiles/r04_mfma16_layout_probe.txt): lane l = (r16 = l % 16, q = l / 16) of the 16x16x64 form holds row /
// column r16 and the 16 K bytes of k-block q; D: lane (c16, q) holds column c16, rows 4 q + r.  The k-blocks of one instruction are
// the four 16-byte chunks of a PAIR of K-steps (64 K bytes), taken in the order sigma = (0, 2, 1, 3): lane q uses chunk
// 2 (q & 1) + (q >> 1), i.e. K-step 2 P + (q & 1), half q >> 1 -- word q & 1 of the record (row, pair P, half q >> 1) on the left,
// chunk 4 P + sigma(q) of the digit row on the right.  With that order the two digit fragments of a 32-column block (X: columns
// 32 j + c16, Y: columns 32 j + 16 + c16) turn into the sparse instruction's 32-byte operand of lane (c32, h) -- chunk (step 2 P,
// half h) then chunk (step 2 P + 1, half h) of column c32 -- by ONE v_permlane16_swap_b32 per dword (swap the odd 16-lane rows of
// X with the even rows of Y): no second set of LDS reads for the mask product.
#pragma once
#include "i8gemm_sparse2.hip.h"

namespace gemma_hip {

__global__ __launch_bounds__(512, 2) void i8gemm_sparse2_g16_kernel(Sparse2Args g) {
  extern __shared__ __attribute__((aligned(1024))) int8_t i8lds[];
  int tm, tn;
  if (g.tile_map) {
    const int2 t2 = g.tile_map[blockIdx.x];
    tm = __builtin_amdgcn_readfirstlane(t2.x);
    tn = __builtin_amdgcn_readfirstlane(t2.y);
  } else {
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
  const int r32 = lane & 31, h = lane >> 5;  // the sparse instruction's lanes
  const int r16 = lane & 15, q = lane >> 4;  // the 16x16x64 instruction's lanes
  const int sig = 2 * (q & 1) + (q >> 1);    // chunk of the pair this lane's k-block is

  const uint4 *asrc[2];
  const int8_t *bsrc[2];
  int adst[2], bdst[2];
#define G16_INIT_SRC(DIGIT)                                                                                       \
  do {                                                                                                            \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                               \
      const int qp = 2 * wave + j;                                                                                \
      const int row = 16 * qp + (lane >> 2);                                                                      \
      asrc[j] = g.AM + ((long)tm * g.nk * S2_BM + row) * 4 + ((lane & 3) ^ ((row >> 2) & 3));                     \
      adst[j] = qp * 1024;                                                                                        \
      const int col = 8 * qp + (lane >> 3);                                                                       \
      bsrc[j] = g.Bt + (long)(DIGIT) * g.strideB + ((long)tn * S2_BN + col) * g.ldk + 16 * ((lane & 7) ^ ((col >> 1) & 7)); \
      bdst[j] = S2_AMB + qp * 1024;                                                                               \
    }                                                                                                             \
  } while (0)
  // fragment byte offsets inside a stage, per pair P of K-steps
  int amo[2], aro[2], fbx[2];
  {
    const int row = wave * 32 + r32;
    const int rowd = wave * 32 + r16; // second group of 16 rows: + 1024 bytes, same swizzle
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      amo[p] = row * 64 + (((2 * p + h) ^ ((row >> 2) & 3)) << 4);
      aro[p] = rowd * 64 + (((2 * p + (q >> 1)) ^ ((rowd >> 2) & 3)) << 4);
      fbx[p] = S2_AMB + r16 * 128 + (((4 * p + sig) ^ ((r16 >> 1) & 7)) << 4); // Y: + 2048, block j: + 4096 j (same swizzle)
    }
  }

  i32x4 acc16[2][8];
  i32x16 accm[4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc16[i][j] = (i32x4){0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) accm[j][r] = 0;

  // register budget: 128 accumulators + the operands of ONE pair of K-steps; the next pair's fragments are requested into the same
  // registers as soon as the last instruction that reads them has issued (an LDS read returns long after that instruction has
  // taken its operands)
  i32x4 am[2];        // records in the sparse instruction's lane layout [pair parity]
  i32x4 ar[2];        // records in the 16x16x64 lane layout [group of 16 rows] (next pair's, until unpacked)
  i32x4 ga[2];        // genotype operands [group]
  i32x8 bxy[5];       // digit fragments [32-column block]: X (columns 32 j + c16) in elements 0..3, Y (+ 16) in 4..7 -- eight
                      // consecutive registers, which after the lane swap ARE the sparse instruction's operand; block 3 alternates
                      // between slots 3 and 4 with the pair's parity (its last reader is the last instruction of a pair: the next
                      // pair's fragment is requested a pair ahead)
  i32x4 ms[2];        // expanded kept bits [pair parity]

#define G16_DMA_A(j, SOFF)                                                                                        \
  do {                                                                                                            \
    __builtin_amdgcn_global_load_lds((gemma_gptr_t)asrc[j], (gemma_lptr_t)(i8lds + (SOFF) + adst[j]), 16, 0, 0);  \
    asrc[j] += S2_BM * 4;                                                                                         \
  } while (0)
#define G16_DMA_B(j, SOFF)                                                                                        \
  do {                                                                                                            \
    __builtin_amdgcn_global_load_lds((gemma_gptr_t)bsrc[j], (gemma_lptr_t)(i8lds + (SOFF) + bdst[j]), 16, 0, 0);  \
    bsrc[j] += I8_BK;                                                                                             \
  } while (0)
#define G16_RAM(SOFF, P) am[(P)&1] = *reinterpret_cast<const i32x4 *>(i8lds + (SOFF) + amo[(P)&1])
// (whole 16-byte records although only their two genotype words are used: in front of 8-byte LDS reads the compiler puts an
// s_waitcnt vmcnt(0) -- it cannot tell them from the LDS-DMA's targets -- and the prefetch would drain every K-tile)
#define G16_RAR(SOFF, P, i) ar[i] = *reinterpret_cast<const i32x4 *>(i8lds + (SOFF) + aro[(P)&1] + (i) * 1024)
// block j of pair P into register slot SL (SL = j for j < 3, 3 + (P & 1) for block 3)
#define G16_RB(SOFF, P, j, SL)                                                                                    \
  do {                                                                                                            \
    const i32x4 x_ = *reinterpret_cast<const i32x4 *>(i8lds + (SOFF) + fbx[(P)&1] + (j) * 4096);                  \
    const i32x4 y_ = *reinterpret_cast<const i32x4 *>(i8lds + (SOFF) + fbx[(P)&1] + (j) * 4096 + 2048);           \
    bxy[SL] = __builtin_shufflevector(__builtin_shufflevector(x_, x_, 0, 1, 2, 3, 0, 1, 2, 3), bxy[SL], 0, 1, 2, 3, 12, 13, 14, 15); \
    bxy[SL] = __builtin_shufflevector(bxy[SL], __builtin_shufflevector(y_, y_, 0, 1, 2, 3, 0, 1, 2, 3), 0, 1, 2, 3, 8, 9, 10, 11);   \
  } while (0)
// (the empty asm keeps all four registers of the record alive up to here: otherwise the unused halves of ar[0] and ar[1] share
// registers and the second read has to wait for the first)
#define G16_UNP(i)                                                                                                \
  do {                                                                                                            \
    ga[i] = s2_unpack_g((q & 1) ? ar[i][1] : ar[i][0]);                                                           \
    asm volatile("" ::"v"(ar[i]));                                                                                \
  } while (0)
#define G16_EXP(P) ms[(P)&1] = s2_expand(am[(P)&1][3])
// dense 16x16x64: group i of rows, sub-block 2 j (X) or 2 j + 1 (Y)
#define G16_DX(i, j, SL)                                                                                          \
  asm volatile("s_nop 1\n\tv_mfma_i32_16x16x64_i8 %0, %1, %2, %0"                                                \
               : "+v"(acc16[i][2 * (j)]) : "v"(ga[i]), "v"(__builtin_shufflevector(bxy[SL], bxy[SL], 0, 1, 2, 3)))
#define G16_DY(i, j, SL)                                                                                          \
  asm volatile("s_nop 1\n\tv_mfma_i32_16x16x64_i8 %0, %1, %2, %0"                                                \
               : "+v"(acc16[i][2 * (j) + 1]) : "v"(ga[i]), "v"(__builtin_shufflevector(bxy[SL], bxy[SL], 4, 5, 6, 7)))
// X, Y of block j -> the two halves of the sparse instruction's digit operand (in place)
#define G16_SWAP(SL)                                                                                              \
  do {                                                                                                            \
    _Pragma("unroll") for (int w_ = 0; w_ < 4; ++w_)                                                              \
      asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(bxy[SL][w_]), "+v"(bxy[SL][4 + w_]));                    \
  } while (0)
#define G16_S(P, j, SL)                                                                                           \
  do {                                                                                                            \
    asm volatile("s_nop 1\n\tv_smfmac_i32_32x32x64_i8 %0, %1, %2, %3"                                            \
                 : "+v"(accm[j]) : "v"(ms[(P)&1]), "v"(bxy[SL]), "v"(am[(P)&1][2]));                               \
  } while (0)
// All the work of pair P (parity known at compile time) from registers.  HEAD: the next pair's record reads and its block-3 fragment
// (other slot); RB0, RB12: its other digit-fragment reads, issued behind the sparse instruction that was the last reader of those
// registers; MID0 / MID1: LDS-DMA issue; TAIL: the next pair's genotype operands and kept bits (behind this pair's last dense
// instruction).  No read is younger than one sparse instruction and four lane swaps when the next pair starts.
#define G16_PAIR(P, HEAD, RB0, RB12, MID0, MID1, TAIL)                                                            \
  do {                                                                                                            \
    G16_DX(0, 0, 0); MID0; GEMMA_SB();                                                                            \
    G16_DY(0, 0, 0); MID1; GEMMA_SB();                                                                            \
    G16_DX(1, 0, 0); GEMMA_SB();                                                                                  \
    G16_DY(1, 0, 0); GEMMA_SB();                                                                                  \
    G16_DX(0, 1, 1); GEMMA_SB();                                                                                  \
    G16_DY(0, 1, 1); GEMMA_SB();                                                                                  \
    G16_DX(1, 1, 1); GEMMA_SB();                                                                                  \
    G16_DY(1, 1, 1); G16_SWAP(0); GEMMA_SB();                                                                     \
    G16_DX(0, 2, 2); HEAD; GEMMA_SB();                                                                            \
    G16_DY(0, 2, 2); GEMMA_SB();                                                                                  \
    G16_S(P, 0, 0); RB0; GEMMA_SB();                                                                              \
    G16_DX(1, 2, 2); GEMMA_SB();                                                                                  \
    G16_DY(1, 2, 2); G16_SWAP(1); GEMMA_SB();                                                                     \
    G16_DX(0, 3, 3 + ((P)&1)); GEMMA_SB();                                                                        \
    G16_DY(0, 3, 3 + ((P)&1)); GEMMA_SB();                                                                        \
    G16_S(P, 1, 1); GEMMA_SB();                                                                                   \
    G16_DX(1, 3, 3 + ((P)&1)); GEMMA_SB();                                                                        \
    G16_DY(1, 3, 3 + ((P)&1)); G16_SWAP(2); GEMMA_SB();                                                           \
    TAIL; GEMMA_SB();                                                                                             \
    G16_S(P, 2, 2); RB12; GEMMA_SB();                                                                             \
    G16_SWAP(3 + ((P)&1)); GEMMA_SB();                                                                            \
    G16_S(P, 3, 3 + ((P)&1)); GEMMA_SB();                                                                         \
    asm volatile("" ::"v"(am[(P)&1])); /* all four registers of the record alive up to here (see G16_UNP) */     \
  } while (0)
// one K-tile from stage SC (MORE: tile t+1 in stage SN; LOAD3: tile t+3 goes to stage SD; VMW as in i8gemm_sparse2.hip.h).  Every
// read of stage SC is issued before the rendezvous in the middle; the reads of stage SN come behind it.
#define G16_KTILE(SC, SN, SD, MORE, LOAD3, VMW)                                                                   \
  do {                                                                                                            \
    G16_PAIR(0, { G16_RAM(SC, 1); G16_RAR(SC, 1, 0); G16_RAR(SC, 1, 1); G16_RB(SC, 1, 3, 4); }, G16_RB(SC, 1, 0, 0), \
             { G16_RB(SC, 1, 1, 1); G16_RB(SC, 1, 2, 2); },                                                       \
             { if (LOAD3) { G16_DMA_A(0, SD); G16_DMA_A(1, SD); } }, { if (LOAD3) { G16_DMA_B(0, SD); G16_DMA_B(1, SD); } }, \
             { G16_UNP(0); G16_UNP(1); G16_EXP(1); });                                                            \
    asm volatile("s_waitcnt vmcnt(" #VMW ")" ::: "memory");                                                       \
    __builtin_amdgcn_s_barrier();                                                                                 \
    GEMMA_SB();                                                                                                   \
    G16_PAIR(1, { if (MORE) { G16_RAM(SN, 0); G16_RAR(SN, 0, 0); G16_RAR(SN, 0, 1); G16_RB(SN, 0, 3, 3); } },      \
             { if (MORE) G16_RB(SN, 0, 0, 0); }, { if (MORE) { G16_RB(SN, 0, 1, 1); G16_RB(SN, 0, 2, 2); } },      \
             {}, {}, { if (MORE) { G16_UNP(0); G16_UNP(1); G16_EXP(0); } });                                       \
  } while (0)

  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  const int nk = g.nk;
  for (int dd = 0; dd < nd; ++dd) {
    if (dd > 0) { // second digit of a fused pair: acc = 256 * C_hi, then accumulate C_lo on top
      asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc16[i][j][r] <<= 8;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accm[j][r] <<= 8;
    }
    G16_INIT_SRC(d_first - dd);
#pragma unroll
    for (int j = 0; j < 2; ++j) { G16_DMA_A(j, 0); G16_DMA_B(j, 0); }
    if (nk > 1) {
#pragma unroll
      for (int j = 0; j < 2; ++j) { G16_DMA_A(j, S2_STAGE); G16_DMA_B(j, S2_STAGE); }
    }
    if (nk > 2) {
#pragma unroll
      for (int j = 0; j < 2; ++j) { G16_DMA_A(j, 2 * S2_STAGE); G16_DMA_B(j, 2 * S2_STAGE); }
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if (nk > 1) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    GEMMA_SB();
    G16_RAM(0, 0);
    G16_RAR(0, 0, 0);
    G16_RAR(0, 0, 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) G16_RB(0, 0, j, j);
    G16_UNP(0);
    G16_UNP(1);
    G16_EXP(0);
    GEMMA_SB();

    int sc = 0, sn = S2_STAGE, s2 = 2 * S2_STAGE, sd = 3 * S2_STAGE;
    int kt = 0;
    for (; kt + 3 < nk; ++kt) {
      G16_KTILE(sc, sn, sd, true, true, 8);
      const int tmp = sc; sc = sn; sn = s2; s2 = sd; sd = tmp;
    }
    if (nk >= 3) {
      G16_KTILE(sc, sn, sd, true, false, 4);
      const int tmp = sc; sc = sn; sn = s2; s2 = sd; sd = tmp;
    }
    if (nk >= 2) {
      G16_KTILE(sc, sn, sd, true, false, 0);
      const int tmp = sc; sc = sn; sn = s2; s2 = sd; sd = tmp;
    }
    G16_KTILE(sc, sn, sd, false, false, 0);
  }

  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
  int *Cg = g.C + (long)plane * g.strideC;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int sb = 0; sb < 8; ++sb) {
      const long col = (long)tn * S2_BN + 16 * sb + r16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = (long)tm * S2_BM + wave * 32 + 16 * i + 4 * q + r;
        Cg[row * g.ldc + col] = acc16[i][sb][r];
      }
    }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long col = (long)tn * S2_BN + j * 32 + r32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long row = (long)tm * S2_BM + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      Cg[(g.m_row0 + row) * g.ldc + col] = accm[j][r];
    }
  }
#undef G16_INIT_SRC
#undef G16_DMA_A
#undef G16_DMA_B
#undef G16_RAM
#undef G16_RAR
#undef G16_RB
#undef G16_UNP
#undef G16_EXP
#undef G16_DX
#undef G16_DY
#undef G16_SWAP
#undef G16_S
#undef G16_PAIR
#undef G16_KTILE
}

} // namespace gemma_hip
