// EXPERIMENT (round 4, not part of the library; measured and dropped): dense int8-digit product for byte planes (fixed-point dosages, i8gemm.hip.h: pack_dosage_kernel), second form: the pipeline of
// i8gemm_sparse2.hip.h (LDS-DMA three K-tiles ahead into four 32 KiB stages, counted s_waitcnt vmcnt, one s_barrier per K-tile,
// matrix instructions as asm volatile in source order) on a tile that suits a product WITHOUT the mask half.
//
// Why another tile: i8gemm_packed_kernel_t<false, true> (128 x 256 x 128 K bytes, three 48 KiB stages) moves 48 KiB into LDS for
// 16 matrix instructions per wavefront -- at 64 B/clk into the CU that is 768 clocks of operand movement under 1024 clocks of
// matrix work per SIMD, and the kernel sits at 0.40 of the dense int8 peak.  Here the tile is 256 rows x 256 columns x 64 K
// bytes: 32 KiB per 16 matrix instructions per wavefront (512 clocks under 1024), 4 LDS-DMA pieces and 12 ds_read_b128 per
// wavefront and K-tile, 128 accumulator registers (wavefronts 4 x 2, each 64 rows x 128 columns = 2 x 4 blocks of 32 x 32).
// Both operand tiles are "rows of 64 bytes": the record geometry of i8gemm_sparse2.hip.h (piece q = 16 rows, lane l -> row l / 4,
// 16-byte chunk l % 4, chunk index XOR-ed with (row >> 2) & 3 on the source address and on the fragment reads).
#pragma once
#include "i8gemm.hip.h"

namespace gemma_hip {

constexpr int D2_BM = 256, D2_BN = 256, D2_BK = 64;
constexpr int D2_AB = 16384;    // bytes of the left tile in a stage; the digit tile follows
constexpr int D2_STAGE = 32768;
constexpr int D2_NST = 4;

struct Dense2Args {
  const int8_t *A;  // rows x ldk signed bytes (row-major), rows a multiple of 256
  const int8_t *Bt; // digit d: columns x ldk (the digits of U, K contiguous), columns a multiple of 256
  int *C;           // plane d: rows x ldc
  long ldk, ldc, strideB, strideC;
  int tiles_m, tiles_n, nk, gm; // nk = ldk / 64
};

__global__ __launch_bounds__(512, 2) void i8gemm_dense2_kernel(Dense2Args g) {
  extern __shared__ __attribute__((aligned(1024))) int8_t i8lds[];
  int tm, tn;
  { // every XCD its own contiguous range of tiles, GM tile rows inside, columns outside (the packed kernels' order)
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
  const int digit = blockIdx.y;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1; // rows wm * 64, columns wn * 128
  const int r32 = lane & 31, h = lane >> 5;

  const int8_t *asrc[2], *bsrc[2];
  int adst[2], bdst[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int qp = 2 * wave + j;
    const int row = 16 * qp + (lane >> 2);
    const int sw = 16 * ((lane & 3) ^ ((row >> 2) & 3));
    asrc[j] = g.A + ((long)tm * D2_BM + row) * g.ldk + sw;
    bsrc[j] = g.Bt + (long)digit * g.strideB + ((long)tn * D2_BN + row) * g.ldk + sw;
    adst[j] = qp * 1024;
    bdst[j] = D2_AB + qp * 1024;
  }
  // fragment byte offsets inside a stage: K-step ks of 32 bytes = chunks 2 ks + h; block i / j: + 32 rows = + 2048 bytes
  int fa[2], fb[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int sw = ((2 * ks + h) ^ ((r32 >> 2) & 3)) << 4;
    fa[ks] = (wm * 64 + r32) * 64 + sw;
    fb[ks] = D2_AB + (wn * 128 + r32) * 64 + sw;
  }

  i32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
  i32x4 ra[2][2], rb[2][4]; // fragments [K-step parity][block]

#define D2_DMA_A(j, SOFF)                                                                                         \
  do {                                                                                                            \
    __builtin_amdgcn_global_load_lds((gemma_gptr_t)asrc[j], (gemma_lptr_t)(i8lds + (SOFF) + adst[j]), 16, 0, 0);  \
    asrc[j] += D2_BK;                                                                                             \
  } while (0)
#define D2_DMA_B(j, SOFF)                                                                                         \
  do {                                                                                                            \
    __builtin_amdgcn_global_load_lds((gemma_gptr_t)bsrc[j], (gemma_lptr_t)(i8lds + (SOFF) + bdst[j]), 16, 0, 0);  \
    bsrc[j] += D2_BK;                                                                                             \
  } while (0)
#define D2_RA(SOFF, KS, i) ra[(KS)&1][i] = *reinterpret_cast<const i32x4 *>(i8lds + (SOFF) + fa[(KS)&1] + (i) * 2048)
#define D2_RB(SOFF, KS, j) rb[(KS)&1][j] = *reinterpret_cast<const i32x4 *>(i8lds + (SOFF) + fb[(KS)&1] + (j) * 2048)
#define D2_M(KS, i, j)                                                                                            \
  asm volatile("s_nop 1\n\tv_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(ra[(KS)&1][i]), "v"(rb[(KS)&1][j]))
// One K-tile from stage SC.  MORE: tile t+1 exists (stage SN); LOAD3: tile t+3 exists and goes to stage SD (the stage of tile
// t-1: every wavefront passed the barrier of the previous K-tile after its last read of it); VMW: LDS-DMA pieces that may still
// be in flight when tile t+1 must have landed.  At entry the fragments of K-step 0 are in ra[0], rb[0].  Reads are issued behind
// the first instructions of a group and consumed by the next group (no read younger than two matrix instructions at a wait).
#define D2_KTILE(SC, SN, SD, MORE, LOAD3, VMW)                                                                    \
  do {                                                                                                            \
    D2_M(0, 0, 0); D2_RA(SC, 1, 0); D2_RB(SC, 1, 0); GEMMA_SB();                                                  \
    D2_M(0, 0, 1); D2_RA(SC, 1, 1); D2_RB(SC, 1, 1); GEMMA_SB();                                                  \
    D2_M(0, 0, 2); D2_RB(SC, 1, 2); D2_RB(SC, 1, 3); GEMMA_SB();                                                  \
    D2_M(0, 0, 3); if (LOAD3) D2_DMA_A(0, SD); GEMMA_SB();                                                        \
    D2_M(0, 1, 0); if (LOAD3) D2_DMA_A(1, SD); GEMMA_SB();                                                        \
    D2_M(0, 1, 1); if (LOAD3) D2_DMA_B(0, SD); GEMMA_SB();                                                        \
    D2_M(0, 1, 2); if (LOAD3) D2_DMA_B(1, SD); GEMMA_SB();                                                        \
    D2_M(0, 1, 3); GEMMA_SB();                                                                                    \
    D2_M(1, 0, 0); GEMMA_SB();                                                                                    \
    D2_M(1, 0, 1); GEMMA_SB();                                                                                    \
    asm volatile("s_waitcnt vmcnt(" #VMW ")" ::: "memory");                                                       \
    __builtin_amdgcn_s_barrier();                                                                                 \
    GEMMA_SB();                                                                                                   \
    D2_M(1, 0, 2); if (MORE) { D2_RA(SN, 0, 0); D2_RB(SN, 0, 0); } GEMMA_SB();                                    \
    D2_M(1, 0, 3); if (MORE) { D2_RA(SN, 0, 1); D2_RB(SN, 0, 1); } GEMMA_SB();                                    \
    D2_M(1, 1, 0); if (MORE) { D2_RB(SN, 0, 2); D2_RB(SN, 0, 3); } GEMMA_SB();                                    \
    D2_M(1, 1, 1); GEMMA_SB();                                                                                    \
    D2_M(1, 1, 2); GEMMA_SB();                                                                                    \
    D2_M(1, 1, 3); GEMMA_SB();                                                                                    \
  } while (0)

  if (wave >= 4) __builtin_amdgcn_s_setprio(1); // as in i8gemm_sparse2_kernel_t: the second-dispatched half loses arbitration on age
  const int nk = g.nk;
  // prologue: tiles 0, 1, 2 in flight, tile 0 landed
#pragma unroll
  for (int j = 0; j < 2; ++j) { D2_DMA_A(j, 0); D2_DMA_B(j, 0); }
  if (nk > 1) {
#pragma unroll
    for (int j = 0; j < 2; ++j) { D2_DMA_A(j, D2_STAGE); D2_DMA_B(j, D2_STAGE); }
  }
  if (nk > 2) {
#pragma unroll
    for (int j = 0; j < 2; ++j) { D2_DMA_A(j, 2 * D2_STAGE); D2_DMA_B(j, 2 * D2_STAGE); }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  } else if (nk > 1) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  GEMMA_SB();
#pragma unroll
  for (int i = 0; i < 2; ++i) D2_RA(0, 0, i);
#pragma unroll
  for (int j = 0; j < 4; ++j) D2_RB(0, 0, j);
  GEMMA_SB();

  int sc = 0, sn = D2_STAGE, s2 = 2 * D2_STAGE, sd = 3 * D2_STAGE; // stage byte offsets: tiles t, t+1, t+2, DMA target
  int kt = 0;
  for (; kt + 3 < nk; ++kt) {
    D2_KTILE(sc, sn, sd, true, true, 8);
    const int tmp = sc; sc = sn; sn = s2; s2 = sd; sd = tmp;
  }
  if (nk >= 3) {
    D2_KTILE(sc, sn, sd, true, false, 4);
    const int tmp = sc; sc = sn; sn = s2; s2 = sd; sd = tmp;
  }
  if (nk >= 2) {
    D2_KTILE(sc, sn, sd, true, false, 0);
    const int tmp = sc; sc = sn; sn = s2; s2 = sd; sd = tmp;
  }
  D2_KTILE(sc, sn, sd, false, false, 0);
#undef D2_DMA_A
#undef D2_DMA_B
#undef D2_RA
#undef D2_RB
#undef D2_M
#undef D2_KTILE

  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); // the matrix instructions are invisible to the hazard recogniser
  int *Cg = g.C + (long)digit * g.strideC;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long col = (long)tn * D2_BN + wn * 128 + j * 32 + r32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long row = (long)tm * D2_BM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        Cg[row * g.ldc + col] = acc[i][j][r];
      }
    }
}

} // namespace gemma_hip
