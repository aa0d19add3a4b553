// PROTOTYPE, NOT SHIPPED (round 5; scripts/i8_kernel_bench.hip variants 13 / 14; profiles/r05_i8_dense16w_prototype.txt).  Measured:
// SLOWER than the 64 x 64-per-wavefront kernel it was meant to replace -- 49.2 against 44.1 ms for six dosage planes at n = B = 20 000,
// 47.5 against 40.8 ms with all-zero digits: one wavefront per SIMD does not keep the matrix pipe fed (a K-tile of 64 bytes is 1 024
// cycles, the LDS-DMA three tiles ahead is 1.3 us of lead) -- and its planes DIFFER from the reference kernel's on three small shapes
// whose padded K is a power of two (n = 100, 200, 500; equal on six others incl. every plane entry at n = 20 000): not investigated,
// the form was dropped on the timing.  Its PREMISE below ("bound by the LDS pipe") was also wrong: counters taken afterwards show the
// LDS pipe active 0.31 of the cycles in the shipped kernel, which runs at 1.66 GHz with the matrix pipe busy 0.63 on these operands
// (profiles/r05_i8_dense_pmc.txt).  Kept for the next attempt at the dense kernel.
//
// The dense byte-plane product on v_mfma_i32_16x16x64_i8 with a 128 x 128 tile PER WAVEFRONT (second step after
// i8gemm_dense16.hip.h): 256 x 256 x 64 tiles, FOUR wavefronts (2 x 2, one per SIMD, up to 512 registers each), four 32 KiB LDS
// stages, LDS-DMA three K-tiles ahead with one counted s_waitcnt vmcnt and one s_barrier per K-tile.
//
// Why: the 16 x 16 x 64 instruction eats 2 KiB of operands per 16 cycles.  With a 64 x 64 wave tile (4 x 4 blocks) every fragment is
// used four times and the 8 wavefronts of a 128 x 256 x 128 tile read 128 KiB from LDS per K-tile on top of 48 KiB of DMA writes:
// 1 408 cycles of the CU's 128 B / cycle LDS pipe for 1 024 cycles of matrix work per SIMD -- the loop is bound by the LDS pipe, and
// both dense kernels need twice the pipe time with an all-zero operand (profiles/r05_i8_dense16.txt).  A 128 x 128 wave tile (8 x 8
// blocks) uses every fragment eight times: 4 x 16 KiB of reads + 32 KiB of DMA writes per K-tile of 64 bytes = 768 LDS cycles for
// 1 024 matrix cycles per SIMD.  The price is one wavefront per SIMD: nothing but this wavefront's own instruction order hides a
// latency, so the fragments of K-tile t + 1 are read behind the matrix instructions of K-tile t (two register sets), the rendezvous
// for tile t + 1 sits 48 instructions ahead of its first use, and the DMA runs three tiles ahead.
//
// Both operand tiles are "rows of 64 bytes" (the record geometry of i8gemm_sparse2.hip.h: piece = 16 rows, lane l -> row l / 4,
// 16-byte chunk l % 4, chunk index XOR-ed with (row >> 2) & 3 on the source address and on the fragment reads).  A fragment of block
// i for the one pair of K-steps of a tile: lane (r16, q) holds the 16 K bytes 16 q .. of row 16 i + r16: ONE ds_read_b128.
// RAW = true: the left bytes are signed values as they are (dosage planes); false: packed bytes g | m << 4 masked to the genotype.
#pragma once
#include "i8gemm.hip.h"

namespace gemma_hip {

constexpr int DW_BM = 256, DW_BN = 256, DW_BK = 64;
constexpr int DW_AB = 16384;    // bytes of the left tile in a stage; the digit tile follows
constexpr int DW_STAGE = 32768;
constexpr int DW_NST = 4;

struct DenseWArgs {
  const int8_t *A;  // rows x ldk bytes (row-major), rows a multiple of 256
  const int8_t *Bt; // plane d: columns x ldk (K contiguous), columns a multiple of 256
  int *C;           // plane d: rows x ldc
  long ldk, ldc, strideB, strideC;
  int tiles_m, tiles_n, nk, gm; // nk = ldk / 64: even (ldk is a multiple of 128)
  const int *tile_map = nullptr; // (tile_m, tile_n) per linear tile index when only some tiles are wanted
};

template <bool RAW>
__global__ __launch_bounds__(256, 1) void i8gemm_dense16w_kernel_t(DenseWArgs g) {
  extern __shared__ __attribute__((aligned(1024))) int8_t i8lds[];
  int tm, tn;
  { // every XCD its own contiguous range of tiles, GM tile rows inside, columns outside (the packed kernels' order)
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
  const int digit = blockIdx.y;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1; // rows wm * 128, columns wn * 128
  const int r16 = lane & 15, q = lane >> 4;

  // LDS-DMA: a stage is 16 + 16 pieces of 1 KiB (16 rows of 64 bytes each); wavefront w moves A pieces 4 w .. 4 w + 3 and the same B pieces
  const int8_t *asrc[4], *bsrc[4];
  int adst[4], bdst[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int qp = 4 * wave + j;
    const int row = 16 * qp + (lane >> 2);
    const int sw = 16 * ((lane & 3) ^ ((row >> 2) & 3));
    asrc[j] = g.A + ((long)tm * DW_BM + row) * g.ldk + sw;
    bsrc[j] = g.Bt + (long)digit * g.strideB + ((long)tn * DW_BN + row) * g.ldk + sw;
    adst[j] = qp * 1024;
    bdst[j] = DW_AB + qp * 1024;
  }
  // fragment byte offsets inside a stage (block i / j: + 16 rows = + 1024 bytes; the swizzle does not change with the block)
  const int sw_f = (q ^ ((r16 >> 2) & 3)) << 4;
  const int fa = (wm * 128 + r16) * 64 + sw_f;
  const int fb = DW_AB + (wn * 128 + r16) * 64 + sw_f;

  i32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (i32x4){0, 0, 0, 0};
  i32x4 xa[8], xb[8], ya[8], yb[8]; // two fragment sets: K-tile t / t + 1
  const i32x4 mask_g = {0x03030303, 0x03030303, 0x03030303, 0x03030303};

#define DW_DMA(e, SOFF)                                                                                           \
  do {                                                                                                            \
    if ((e) < 4) {                                                                                                \
      __builtin_amdgcn_global_load_lds((gemma_gptr_t)asrc[(e)&3], (gemma_lptr_t)(i8lds + (SOFF) + adst[(e)&3]), 16, 0, 0); \
      asrc[(e)&3] += DW_BK;                                                                                       \
    } else {                                                                                                      \
      __builtin_amdgcn_global_load_lds((gemma_gptr_t)bsrc[(e)&3], (gemma_lptr_t)(i8lds + (SOFF) + bdst[(e)&3]), 16, 0, 0); \
      bsrc[(e)&3] += DW_BK;                                                                                       \
    }                                                                                                             \
  } while (0)
// read e of a tile from stage SOFF: e 0..7 = A blocks, 8..15 = B blocks
#define DW_READ(e, SOFF, RA, RB)                                                                                  \
  do {                                                                                                            \
    if ((e) < 8) {                                                                                                \
      RA[(e)&7] = *reinterpret_cast<const i32x4 *>(i8lds + (SOFF) + fa + ((e)&7) * 1024);                         \
      if (!RAW) RA[(e)&7] = RA[(e)&7] & mask_g;                                                                   \
    } else {                                                                                                      \
      RB[(e)&7] = *reinterpret_cast<const i32x4 *>(i8lds + (SOFF) + fb + ((e)&7) * 1024);                         \
    }                                                                                                             \
  } while (0)
// matrix instruction m of a tile: block (i, j) = (m >> 3, m & 7): 64 different accumulators in a row
// (inline asm with the accumulator pinned to the ACCUMULATION registers, destination = source: as a builtin the register allocator
// gave 256 accumulator registers new homes from instruction to instruction and moved them back with 280 v_accvgpr copies per two
// tiles -- and spilled; the s_nop stands for the hazard the compiler cannot see behind a VALU-written operand)
#define DW_MF(m, RA, RB)                                                                                          \
  asm volatile("s_nop 1\n\tv_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(acc[(m) >> 3][(m)&7]) : "v"(RA[(m) >> 3]), "v"(RB[(m)&7]))
// One K-tile on the fragments (CA, CB).  LOAD3: tile t + 3 exists and goes to stage SD = the stage of tile t - 1 (every wavefront
// passed the previous tile's barrier after its last read of it): its eight pieces behind the first eight instructions.  MORE: tile
// t + 1 exists in stage SN: the counted wait (VMW pieces of later tiles may still be in flight) and the rendezvous behind instruction
// 15, its sixteen fragment reads behind instructions 16 .. 31 -- 32 instructions ahead of their first use.
#define DW_KTILE(CA, CB, NA, NB, SN, SD, MORE, LOAD3, VMW)                                                        \
  do {                                                                                                            \
    _Pragma("unroll") for (int m_ = 0; m_ < 16; ++m_) {                                                           \
      DW_MF(m_, CA, CB);                                                                                          \
      if ((LOAD3) && m_ < 8) DW_DMA(m_, SD);                                                                      \
      GEMMA_SB();                                                                                                 \
    }                                                                                                             \
    if (MORE) {                                                                                                   \
      asm volatile("s_waitcnt vmcnt(" #VMW ")" ::: "memory");                                                     \
      __builtin_amdgcn_s_barrier();                                                                               \
      GEMMA_SB();                                                                                                 \
    }                                                                                                             \
    _Pragma("unroll") for (int m_ = 16; m_ < 64; ++m_) {                                                          \
      DW_MF(m_, CA, CB);                                                                                          \
      if ((MORE) && m_ < 32) DW_READ(m_ - 16, SN, NA, NB);                                                        \
      GEMMA_SB();                                                                                                 \
    }                                                                                                             \
  } while (0)
#define DW_ROT()                                                                                                  \
  do {                                                                                                            \
    const int tmp_ = sc; sc = sn; sn = s2; s2 = sd; sd = tmp_;                                                    \
  } while (0)

  const int nk = g.nk;
  // prologue: tiles 0, 1, 2 in flight, tile 0 landed
#pragma unroll
  for (int e = 0; e < 8; ++e) DW_DMA(e, 0);
  if (nk > 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) DW_DMA(e, DW_STAGE);
  }
  if (nk > 2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) DW_DMA(e, 2 * DW_STAGE);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  } else if (nk > 1) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  GEMMA_SB();
#pragma unroll
  for (int e = 0; e < 16; ++e) DW_READ(e, 0, xa, xb);
  GEMMA_SB();

  int sc = 0, sn = DW_STAGE, s2 = 2 * DW_STAGE, sd = 3 * DW_STAGE; // stage byte offsets: tiles t, t + 1, t + 2, DMA target
  (void)sc; (void)s2;
  int kt = 0;
  for (; kt + 4 < nk; kt += 2) { // two tiles per trip: the fragment sets swap roles
    DW_KTILE(xa, xb, ya, yb, sn, sd, true, true, 16);
    DW_ROT();
    DW_KTILE(ya, yb, xa, xb, sn, sd, true, true, 16);
    DW_ROT();
  }
  if (nk - kt >= 4) { // the last four tiles
    DW_KTILE(xa, xb, ya, yb, sn, sd, true, true, 16);
    DW_ROT();
    DW_KTILE(ya, yb, xa, xb, sn, sd, true, false, 8);
    DW_ROT();
    DW_KTILE(xa, xb, ya, yb, sn, sd, true, false, 0);
    DW_ROT();
    DW_KTILE(ya, yb, xa, xb, sn, sd, false, false, 0);
  } else if (nk - kt == 2) { // nk == 2
    DW_KTILE(xa, xb, ya, yb, sn, sd, true, false, 0);
    DW_ROT();
    DW_KTILE(ya, yb, xa, xb, sn, sd, false, false, 0);
  } else { // nk == 1 (the host only launches even nk -- ldk is a multiple of 128 -- or 1)
    DW_KTILE(xa, xb, ya, yb, sn, sd, false, false, 0);
  }
#undef DW_DMA
#undef DW_READ
#undef DW_MF
#undef DW_KTILE
#undef DW_ROT

  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); // the matrix instructions are invisible to the hazard recogniser
  int *Cg = g.C + (long)digit * g.strideC;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long col = (long)tn * DW_BN + wn * 128 + 16 * j + r16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = (long)tm * DW_BM + wm * 128 + 16 * i + 4 * q + r;
        Cg[row * g.ldc + col] = acc[i][j][r];
      }
    }
}

} // namespace gemma_hip
