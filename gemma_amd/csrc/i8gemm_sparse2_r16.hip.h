// The records kernel (i8gemm_sparse2.hip.h: same records, same digit planes, same int32 planes out, same LDS stages and LDS-DMA
// pipeline, wavefronts 8 x 1) on the 16-ROW matrix instructions: genotype product on v_mfma_i32_16x16x64_i8, mask product on
// v_smfmac_i32_16x16x128_i8.  The shipped form since the end of round 4 (GEMMA_HIP_I8_ROWS=32 selects the 32-row kernel).
//
// Why: the kernel is power-limited, and how much clock the chip gives up depends on the instruction and on the VALUES it multiplies
// (DESIGN.md 3.1c).  With operands in registers and nothing else running v_mfma_i32_32x32x32_i8 sustains 3.70 POP/s on full-range
// digit values and the 16x16x64 form 4.72 (profiles/r04_mfma_power_probe.txt); this kernel needs 40.8 ms of schedule (all-zero
// digits; the 32-row kernel 38.7) and 49.8 ms with real digits where the 32-row kernel needs 54.6 on the same box
// (profiles/r04_i8_g16s_prototype.txt).  Every entry of every plane equals the 32-row kernel's on ten shapes -- 2 454 061 056 entries
// at n = 20 000, odd digit counts, unfused planes, K loops of 1 - 3 tiles, ragged tiles (profiles/r04_i8_r16_full_compare.txt,
// scripts/i8_kernel_bench.hip variant 7 with FULLCMP=1).
//
// Layouts (profiles/r04_mfma16_layout_probe.txt, r04_smfmac16_layout_probe.txt).  Lane l = (r16 = l % 16, q = l / 16).
//   dense 16x16x64, pair P of K-steps: lane q multiplies the 16 K bytes 64 P + 16 q .. of row / column r16 -- K-step 2 P + (q >> 1),
//     half q & 1: left operand = word q >> 1 of the record chunk 2 P + (q & 1) of the row, right operand F[P] = chunk 4 P + q of the
//     digit row;
//   sparse 16x16x128, the whole K-tile: lane quarter q of the left operand covers the 32 logical K bytes of K-step q (kept slots
//     0..7 its first 16, 8..15 its second 16) = index word and kept bits of the record chunk q of the row -- which the records
//     already hold; its right operand in lane (c16, qb) is (F[0], F[1]) of the SAME lane: chunk beta of quarter qb multiplies
//     quarter qa = (qb >> 1) + 2 beta, slots 8 (qb & 1) ..: K bytes 64 beta + 16 qb .. = F[beta] of lane qb.  So one 8-register
//     tuple per 16-column sub-block serves two dense and one sparse instruction per group of 16 rows: no lane exchange, no second
//     set of LDS reads.
// Per K-tile and wavefront: 4 record reads + 16 digit reads (ds_read_b128), 32 dense + 16 sparse instructions, 4 LDS-DMA pieces, one
// counted s_waitcnt vmcnt(8) + one s_barrier; tests/test_isa_schedule.py holds the built loop to that.
#pragma once
#include "i8gemm_sparse2.hip.h"

#ifndef S2_R16_PREP_SPREAD
#define S2_R16_PREP_SPREAD 0
#endif
// timing experiments only (scripts/i8_kernel_bench.hip, round 6 power table): the sparse (mask) / the dense (genotype) matrix
// instructions compiled out -- results wrong, everything else of the loop (LDS reads, LDS-DMA, operand preparation) unchanged
// LDS stages of the K pipeline (S2_STAGE = 32 KiB each): LDS-DMA runs S2_R16_NST - 1 K-tiles ahead.  4 = rounds 3-5; 5 fills the
// CU's 160 KiB exactly: 128 instead of 96 KiB of operands in flight per CU.  Round 6: the loop's data side is bound by
// (bytes in flight) / (latency of an L2 miss served across the fabric) -- profiles/r06_power_table.txt: with the dense matrix
// instructions compiled out the same loop still takes 29 ms for its 382 GB of LDS-DMA = 53 GB/s per CU = 96 KiB per ~1.8 us.
#ifndef S2_R16_NST
#define S2_R16_NST 4
#endif
#ifndef S2_R16_ABL_NO_S
#define S2_R16_ABL_NO_S 0
#endif
#ifndef S2_R16_ABL_NO_D
#define S2_R16_ABL_NO_D 0
#endif

namespace gemma_hip {

constexpr int S2_R16_LDS = S2_R16_NST * S2_STAGE; // dynamic LDS of a launch

// WITH_S = false: the genotype product alone (round 6, the "7g6m" form: the lowest of seven digits is multiplied with the genotypes
// only -- its mask term is below the rounding of the other six): no sparse instruction, no mask accumulators, no M rows
// written.  The same loop otherwise; one source so that the two cannot drift apart.
template <bool WITH_S>
__device__ __forceinline__ void s2_r16_body(const Sparse2Args &g) {
  extern __shared__ __attribute__((aligned(1024))) int8_t i8lds[];
  constexpr bool NO_S = !WITH_S || S2_R16_ABL_NO_S;
  if (g.run_if) { // Sparse2Args::anymiss: the form that is not this block's returns before it touches anything
    const int any = __builtin_amdgcn_readfirstlane(*g.anymiss);
    if ((g.run_if == 1) != (any != 0)) return;
  }
  int tm, tn;
  if (g.tile_map) {
    const int2 t2 = g.tile_map[blockIdx.x];
    tm = __builtin_amdgcn_readfirstlane(t2.x);
    tn = __builtin_amdgcn_readfirstlane(t2.y);
  } else {
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
  }
  const int plane = blockIdx.y + g.plane0;
  const int odd = g.digits & 1;
  const int d_first = g.fuse ? (odd ? (plane == 0 ? 0 : 2 * plane) : 2 * plane + 1) : plane;
  const int nd = (g.fuse && !(odd && plane == 0)) ? 2 : 1;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int r16 = lane & 15, q = lane >> 4;

  const uint4 *asrc[2];
  const int8_t *bsrc[2];
  int adst[2], bdst[2];
#define GS_INIT_SRC(DIGIT)                                                                                        \
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
  // byte offsets inside a stage: record chunk (q & 1) + 2 P of row wave * 32 + r16 (second row group: + 1024), digit chunk 4 P + q of
  // column r16 (sub-block sb: + 2048 sb; the swizzles do not change with either)
  int ro[2], fo[2];
  {
    const int row = wave * 32 + r16;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      ro[p] = row * 64 + (((2 * p + (q & 1)) ^ ((row >> 2) & 3)) << 4);
      fo[p] = S2_AMB + r16 * 128 + (((4 * p + q) ^ ((r16 >> 1) & 7)) << 4);
    }
  }

  i32x4 accg[2][8], accm[2][8];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { accg[i][j] = (i32x4){0, 0, 0, 0}; accm[i][j] = (i32x4){0, 0, 0, 0}; }

  i32x4 rec[2][2];  // records of the NEXT K-tile [row group][pair]: chunk 2 P + (q & 1)
  i32x4 ga[2][2];   // genotype operands [row group][pair]
  i32x4 ms[2];      // kept values of K-step q [row group]
  int ix[2];        // index words of K-step q [row group]
  i32x8 T[4];       // digit fragments (F[0], F[1]) of four sub-blocks in flight (ring: slot = sub-block % 4, the same in every K-tile)

#define GS_DMA_A(j, SOFF)                                                                                         \
  do {                                                                                                            \
    __builtin_amdgcn_global_load_lds((gemma_gptr_t)asrc[j], (gemma_lptr_t)(i8lds + (SOFF) + adst[j]), 16, 0, 0);  \
    asrc[j] += S2_BM * 4;                                                                                         \
  } while (0)
#define GS_DMA_B(j, SOFF)                                                                                         \
  do {                                                                                                            \
    __builtin_amdgcn_global_load_lds((gemma_gptr_t)bsrc[j], (gemma_lptr_t)(i8lds + (SOFF) + bdst[j]), 16, 0, 0);  \
    bsrc[j] += I8_BK;                                                                                             \
  } while (0)
#define GS_RREC(SOFF, i, P) rec[i][P] = *reinterpret_cast<const i32x4 *>(i8lds + (SOFF) + ro[P] + (i) * 1024)
// sub-block SB of the stage into ring slot SL
#define GS_RT(SOFF, SB, SL)                                                                                       \
  do {                                                                                                            \
    const i32x4 x_ = *reinterpret_cast<const i32x4 *>(i8lds + (SOFF) + fo[0] + (SB) * 2048);                      \
    const i32x4 y_ = *reinterpret_cast<const i32x4 *>(i8lds + (SOFF) + fo[1] + (SB) * 2048);                      \
    T[SL] = __builtin_shufflevector(x_, y_, 0, 1, 2, 3, 4, 5, 6, 7);                                              \
  } while (0)
// operands of a K-tile from its records: genotype word q >> 1 of (pair P, half q & 1); index word and kept bits of K-step q = the
// record of pair q >> 1 (its half is q & 1 by construction)
#define GS_PREP(i)                                                                                                \
  do {                                                                                                            \
    ga[i][0] = s2_unpack_g((q >> 1) ? rec[i][0][1] : rec[i][0][0]);                                               \
    ga[i][1] = s2_unpack_g((q >> 1) ? rec[i][1][1] : rec[i][1][0]);                                               \
    ix[i] = (q >> 1) ? rec[i][1][2] : rec[i][0][2];                                                               \
    ms[i] = s2_expand((q >> 1) ? rec[i][1][3] : rec[i][0][3]);                                                    \
    asm volatile("" ::"v"(rec[i][0]), "v"(rec[i][1]));                                                            \
  } while (0)
// the same in pieces (S2_R16_PREP_SPREAD): each operand register is rebuilt right behind the LAST matrix instruction of the K-tile
// that reads it (step 7: D(0,P0) D(1,P0) D(0,P1) D(1,P1) S(0) S(1)), so the 66 VALU instructions of the preparation sit between
// the step's matrix instructions instead of behind them
#define GS_PREP_G(i, P) ga[i][P] = s2_unpack_g((q >> 1) ? rec[i][P][1] : rec[i][P][0])
#define GS_PREP_M(i)                                                                                              \
  do {                                                                                                            \
    ix[i] = (q >> 1) ? rec[i][1][2] : rec[i][0][2];                                                               \
    ms[i] = s2_expand((q >> 1) ? rec[i][1][3] : rec[i][0][3]);                                                    \
    asm volatile("" ::"v"(rec[i][0]), "v"(rec[i][1]));                                                            \
  } while (0)
#define GS_D(i, SB, P, SL)                                                                                        \
  do {                                                                                                            \
    if (!S2_R16_ABL_NO_D)                                                                                         \
      asm volatile("s_nop 1\n\tv_mfma_i32_16x16x64_i8 %0, %1, %2, %0"                                             \
                   : "+v"(accg[i][SB])                                                                            \
                   : "v"(ga[i][P]), "v"(__builtin_shufflevector(T[SL], T[SL], 4 * (P), 4 * (P) + 1, 4 * (P) + 2, 4 * (P) + 3))); \
  } while (0)
#define GS_S(i, SB, SL)                                                                                           \
  do {                                                                                                            \
    if (!NO_S)                                                                                                    \
      asm volatile("s_nop 1\n\tv_smfmac_i32_16x16x128_i8 %0, %1, %2, %3" : "+v"(accm[i][SB]) : "v"(ms[i]), "v"(T[SL]), "v"(ix[i])); \
  } while (0)
// the six matrix instructions of sub-block SB (ring slot SL); X: statements issued behind the first two
#define GS_STEP(SB, SL, X)                                                                                        \
  do {                                                                                                            \
    GS_D(0, SB, 0, SL); GEMMA_SB();                                                                               \
    GS_D(1, SB, 0, SL); X; GEMMA_SB();                                                                            \
    GS_D(0, SB, 1, SL); GEMMA_SB();                                                                               \
    GS_D(1, SB, 1, SL); GEMMA_SB();                                                                               \
    GS_S(0, SB, SL); GEMMA_SB();                                                                                  \
    GS_S(1, SB, SL); GEMMA_SB();                                                                                  \
  } while (0)
// One K-tile from stage SC; at entry sub-blocks 0, 1, 2 are in ring slots 0, 1, 2 and the operands of this tile are prepared.
// Sub-block s + 3 is requested behind the first instructions of step s (its slot was freed by step s - 1).  The reads of stage SC
// end in step 4; the rendezvous for tile t + 1 follows step 4; steps 5, 6, 7 request its records and its first three sub-blocks.
#define GS_KTILE(SC, SN, SD, MORE, LOAD3, VMW)                                                                    \
  do {                                                                                                            \
    GS_STEP(0, 0, { GS_RT(SC, 3, 3); if (LOAD3) GS_DMA_A(0, SD); });                                               \
    GS_STEP(1, 1, { GS_RT(SC, 4, 0); if (LOAD3) GS_DMA_A(1, SD); });                                               \
    GS_STEP(2, 2, { GS_RT(SC, 5, 1); if (LOAD3) GS_DMA_B(0, SD); });                                               \
    GS_STEP(3, 3, { GS_RT(SC, 6, 2); if (LOAD3) GS_DMA_B(1, SD); });                                               \
    GS_STEP(4, 0, { GS_RT(SC, 7, 3); });                                                                           \
    asm volatile("s_waitcnt vmcnt(" #VMW ")" ::: "memory");                                                       \
    __builtin_amdgcn_s_barrier();                                                                                 \
    GEMMA_SB();                                                                                                   \
    GS_STEP(5, 1, { if (MORE) { GS_RREC(SN, 0, 0); GS_RREC(SN, 0, 1); GS_RREC(SN, 1, 0); GS_RREC(SN, 1, 1);       \
                                GS_RT(SN, 0, 0); } });                                                             \
    GS_STEP(6, 2, { if (MORE) { GS_RT(SN, 1, 1); } });                                                             \
    if (S2_R16_PREP_SPREAD && (MORE)) {                                                                           \
      GS_D(0, 7, 0, 3); GEMMA_SB();                                                                               \
      GS_D(1, 7, 0, 3); { GS_RT(SN, 2, 2); GS_PREP_G(0, 0); } GEMMA_SB();                                          \
      GS_D(0, 7, 1, 3); { GS_PREP_G(1, 0); } GEMMA_SB();                                                           \
      GS_D(1, 7, 1, 3); { GS_PREP_G(0, 1); } GEMMA_SB();                                                           \
      GS_S(0, 7, 3); { GS_PREP_G(1, 1); } GEMMA_SB();                                                              \
      GS_S(1, 7, 3); { GS_PREP_M(0); } GEMMA_SB();                                                                 \
      { GS_PREP_M(1); } GEMMA_SB();                                                                                \
    } else {                                                                                                      \
      GS_STEP(7, 3, { if (MORE) { GS_RT(SN, 2, 2); } });                                                           \
      if (MORE) { GS_PREP(0); GS_PREP(1); }                                                                       \
      GEMMA_SB();                                                                                                 \
    }                                                                                                             \
  } while (0)

  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  const int nk = g.nk;
  for (int dd = 0; dd < nd; ++dd) {
    if (dd > 0) {
      asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            accg[i][j][r] <<= 8;
            if (WITH_S) accm[i][j][r] <<= 8;
          }
    }
    GS_INIT_SRC(d_first - dd);
    // prologue: tiles 0 .. NST - 2 in flight, tile 0 landed
#pragma unroll
    for (int j = 0; j < 2; ++j) { GS_DMA_A(j, 0); GS_DMA_B(j, 0); }
    if (nk > 1) {
#pragma unroll
      for (int j = 0; j < 2; ++j) { GS_DMA_A(j, S2_STAGE); GS_DMA_B(j, S2_STAGE); }
    }
    if (nk > 2) {
#pragma unroll
      for (int j = 0; j < 2; ++j) { GS_DMA_A(j, 2 * S2_STAGE); GS_DMA_B(j, 2 * S2_STAGE); }
    }
    if (S2_R16_NST == 5 && nk > 3) {
#pragma unroll
      for (int j = 0; j < 2; ++j) { GS_DMA_A(j, 3 * S2_STAGE); GS_DMA_B(j, 3 * S2_STAGE); }
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    } else if (nk > 2) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if (nk > 1) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    GEMMA_SB();
    GS_RT(0, 0, 0);
    GS_RT(0, 1, 1);
    GS_RT(0, 2, 2);
    GS_RREC(0, 0, 0); GS_RREC(0, 0, 1); GS_RREC(0, 1, 0); GS_RREC(0, 1, 1);
    GS_PREP(0);
    GS_PREP(1);
    GEMMA_SB();

    // stage byte offsets: tile t, t + 1, ..., the LDS-DMA target (the stage tile t - 1 was read from)
    int sc = 0, sn = S2_STAGE, s2 = 2 * S2_STAGE, s3 = 3 * S2_STAGE, sd = (S2_R16_NST - 1) * S2_STAGE;
#define GS_ROTATE()                                                                                               \
  do {                                                                                                            \
    const int tmp = sc;                                                                                           \
    sc = sn; sn = s2;                                                                                             \
    if (S2_R16_NST == 5) { s2 = s3; s3 = sd; } else { s2 = sd; }                                                  \
    sd = tmp;                                                                                                     \
  } while (0)
    int kt = 0;
    if (S2_R16_NST == 5) {
      for (; kt + 4 < nk; ++kt) {
        GS_KTILE(sc, sn, sd, true, true, 12);
        GS_ROTATE();
      }
      if (nk >= 4) {
        GS_KTILE(sc, sn, sd, true, false, 8);
        GS_ROTATE();
      }
    } else {
      for (; kt + 3 < nk; ++kt) {
        GS_KTILE(sc, sn, sd, true, true, 8);
        GS_ROTATE();
      }
    }
    if (nk >= 3) {
      GS_KTILE(sc, sn, sd, true, false, 4);
      GS_ROTATE();
    }
    if (nk >= 2) {
      GS_KTILE(sc, sn, sd, true, false, 0);
      GS_ROTATE();
    }
    GS_KTILE(sc, sn, sd, false, false, 0);
#undef GS_ROTATE
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
        Cg[row * g.ldc + col] = accg[i][sb][r];
        if (WITH_S) Cg[(g.m_row0 + row) * g.ldc + col] = accm[i][sb][r];
      }
    }
#undef GS_INIT_SRC
#undef GS_DMA_A
#undef GS_DMA_B
#undef GS_RREC
#undef GS_RT
#undef GS_PREP
#undef GS_PREP_G
#undef GS_PREP_M
#undef GS_D
#undef GS_S
#undef GS_STEP
#undef GS_KTILE
}

__global__ __launch_bounds__(512, 2) void i8gemm_sparse2_r16_kernel(Sparse2Args g) { s2_r16_body<true>(g); }
// the genotype product alone (plane 0 of the 7g6m form)
__global__ __launch_bounds__(512, 2) void i8gemm_sparse2_r16_g_kernel(Sparse2Args g) { s2_r16_body<false>(g); }

} // namespace gemma_hip
