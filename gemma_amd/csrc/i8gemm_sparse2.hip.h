// Exact int8-digit U^T x of hard-call genotypes, third form: 256 SNP rows x 128 columns per workgroup, the left factor
// streamed as 16-byte RECORDS -- 2-bit genotypes and the words of the 2:4-sparse missing-mask operand side by side --
// instead of one byte per call plus separate mask words.
//
// Why (profiles/r03_i8_sparse_ablation.txt): with the in-loop LDS-DMA of i8gemm_sparse_kernel compiled out the same matrix
// instructions take 49-53 ms instead of 65-68 -- a quarter of the kernel is the cost of moving operand bytes into LDS
// (7 one-KiB pieces per wavefront and 128 bytes of K) -- and its per-MFMA sched_barriers did not pin the order of the matrix
// instructions (they carry no chain: SelectionDAG bunched them, dependent pairs back to back), and a compiler-inserted
// s_waitcnt vmcnt(0) in front of the mask-word reads drained the two-tiles-ahead prefetch every K-tile.  Here:
//   * a call needs 2 bits, not 8: per (row, 32 individuals) the left factor is 8 bytes of genotypes + the 8-byte mask word;
//     the byte-wise B operand (digits of U) cannot shrink, so the tile is turned to 256 rows x 128 columns, which balances
//     16 B/row against 32 B/column per K-step: 32 KiB per 128 bytes of K instead of 52 KiB, 4 LDS-DMA pieces per wavefront
//     instead of 7 (and 12 ds_read_b128 instead of 18);
//   * one ds_read_b128 per 32-row block and PAIR of K-steps returns {genotype word of step 2p, of step 2p+1, index word, kept
//     bits} for lane (row r, half h) -- exactly what its two dense genotype MFMAs and its sparse mask MFMA of that pair consume;
//     the genotype word holds individual 16 h + 4 i + j at bits 8 j + 2 i, so operand dword i is (w >> 2 i) & 0x03030303;
//     the kept bits hold element 4 i + j at bit 8 j + i, so operand dword i is (bits >> i) & 0x01010101;
//   * four LDS stages of 32 KiB, LDS-DMA three K-tiles ahead, counted s_waitcnt vmcnt(8);
//   * the matrix instructions are asm volatile: their order against the sched_barriers, the LDS reads and the LDS-DMA is the
//     source order.  The compiler then no longer sees an MFMA, so what its hazard recogniser would have inserted is written
//     out: two wait states between a VALU write of an operand and the MFMA (s_nop 1 in front of each), and the full
//     MFMA-to-VALU distance before the accumulators are shifted / stored (s_nop 15; s_nop 7).
// Layout of the sparse operand, surplus calls (groups of four with more than two missing calls): as i8gemm_sparse.hip.h.
#pragma once
#include <algorithm>
#include <vector>
#include "i8gemm_sparse.hip.h"

namespace gemma_hip {

constexpr int S2_BM = 256, S2_BN = 128;
constexpr int S2_AMB = 16384;   // records of a K-tile: 256 rows x 4 chunks x 16 B
constexpr int S2_STAGE = 32768; // + digit tile 128 columns x 128 B
constexpr int S2_NST = 4;
#ifndef S2_ADVANCE
#define S2_ADVANCE 1 // 0: timing experiment only: every K-tile re-reads the first one (cache-hot operands), results wrong
#endif
#ifndef S2_LOOPDMA
#define S2_LOOPDMA 1 // 0: timing experiment only (scripts/i8_kernel_bench.hip): the in-loop LDS-DMA compiled out, results wrong
#endif

struct Sparse2Args {
  const uint4 *AM;  // [tile_m][ktile][row % 256][chunk 2 p + h]: {g2(step 2p, h), g2(step 2p+1, h), idx(step 2p+h), kept bits}
  const int8_t *Bt; // digit d: N x ldk
  int *C;           // plane q: (2 lpad) x ldc; rows [0, lpad) = G products, [lpad, 2 lpad) = M products
  long ldk, ldc, strideB, strideC, m_row0;
  int tiles_m, tiles_n, nk, gm, fuse, digits;
  int plane0 = 0; // first plane of this launch (i8gemm_sparse2_r16.hip.h: the 7g6m form launches plane 0 and planes 1.. separately)
  const int2 *tile_map = nullptr; // (tile_m, tile_n) of workgroup blockIdx.x: the cross-XCD raster of s2_build_raster; nullptr:
                                  // the per-XCD ranges of round 3 (every XCD sweeps its own tile rows)
  // Blocks without a missing call (round 6): sparse2_meta_kernel leaves *anymiss = 0 for them and the mask product is identically
  // zero.  The launch site then queues BOTH forms of the 16-row kernel and each one returns at once unless the flag is its own
  // (run_if 1: only when *anymiss != 0, 2: only when *anymiss == 0, 0: always) -- the choice is made on the device, so the
  // asynchronous pipeline needs no read-back, and neither K loop changes.
  const int *anymiss = nullptr;
  int run_if = 0;
};
// Tried in round 4 and dropped: one digit fewer for the MASK product (the mask product sums only the row's missing calls, so
// five digits keep its worst-case error at the level of the six-digit genotype product: -1/18 of the matrix instructions).  The
// lowest digit's pass then runs without the sparse instructions; as a second copy of the K loop it made the register allocator
// spill -- and a scratch access is a vector-memory operation the counted s_waitcnt vmcnt of the LDS-DMA pipeline does not know
// about: wrong results --, behind scalar branches in one loop the compiler treats the flag as divergent and puts an
// s_waitcnt lgkmcnt(0) in front of every sparse instruction.  Measured gain of the (broken) two-loop form: 1.8-3.5 %, the pass
// without the mask product being short of matrix work to hide its operand movement (profiles/r04_i8_raster_mask.txt).

// packed bytes g | m << 4 (lpad x ldk, lpad a multiple of 256) -> records; one thread per (row, K-tile, chunk)
__global__ __launch_bounds__(256) void sparse2_meta_kernel(const int8_t *__restrict__ A, long lpad, long ldk,
                                                           uint4 *__restrict__ AM, int *__restrict__ row_surplus,
                                                           int *anymiss = nullptr /* set to 1 when the block holds a missing call */) {
  const long nk = ldk / I8_BK;
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= lpad * nk * 4) return;
  const int c = (int)(id & 3);
  const long rr = (id >> 2) & 255, tk = id >> 10; // tk = tile_m * nk + ktile
  const long tmi = tk / nk, kt = tk - tmi * nk;
  const long row = tmi * S2_BM + rr;
  const int p = c >> 1, h = c & 1;
  const int8_t *base = A + row * ldk + kt * I8_BK;
  unsigned w[4];
#pragma unroll
  for (int e = 0; e < 2; ++e) { // genotype words of steps 2 p + e, half h
    const int8_t *src = base + 32 * (2 * p + e) + 16 * h;
    unsigned g2 = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) g2 |= (*reinterpret_cast<const unsigned *>(src + 4 * i) & 0x03030303u) << (2 * i);
    w[e] = g2;
  }
  { // mask word of the 32 individuals of K-step 2 p + h (natural order, groups of four)
    const int8_t *src = base + 32 * (2 * p + h);
    unsigned idx = 0, bits = 0;
    int surplus = 0;
    for (int gq = 0; gq < 8; ++gq) {
      const unsigned word = *reinterpret_cast<const unsigned *>(src + 4 * gq);
      const unsigned m = (word >> 4) & 0x01010101u;
      const unsigned pat = (m | (m >> 7) | (m >> 14) | (m >> 21)) & 0xFu;
      const int cnt = __popc(pat);
      const int p0 = cnt >= 1 ? __ffs(pat) - 1 : 0;
      const unsigned rest = pat & (pat - 1);
      const int p1 = cnt >= 2 ? __ffs(rest) - 1 : (p0 == 3 ? 2 : 3);
      idx |= (unsigned)(p0 | (p1 << 2)) << (4 * gq);
      // kept element e = 2 gq + slot -> bit 8 (e % 4) + e / 4
      const int e0 = 2 * gq, e1 = 2 * gq + 1;
      bits |= (unsigned)(cnt >= 1) << (8 * (e0 & 3) + (e0 >> 2));
      bits |= (unsigned)(cnt >= 2) << (8 * (e1 & 3) + (e1 >> 2));
      surplus += cnt > 2 ? cnt - 2 : 0;
    }
    w[2] = idx;
    w[3] = bits;
    if (surplus) atomicAdd(row_surplus + row, surplus);
    // every writer stores the same value (a wavefront's lanes coalesce into one write), and only while the flag still reads 0
    if (anymiss && bits != 0 && *reinterpret_cast<volatile int *>(anymiss) == 0) *reinterpret_cast<volatile int *>(anymiss) = 1;
  }
  AM[id] = make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ i32x4 s2_unpack_g(int w) {
  i32x4 v;
  const unsigned u = (unsigned)w;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = (int)((u >> (2 * i)) & 0x03030303u);
  return v;
}
__device__ __forceinline__ i32x4 s2_expand(int bits) {
  i32x4 v;
  const unsigned u = (unsigned)bits;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = (int)((u >> i) & 0x01010101u);
  return v;
}

// Cross-XCD raster (round 4).  Workgroup b runs on XCD b % 8 and an XCD takes its workgroups in order, 32 at a time (one per
// CU: 128 KiB of LDS each).  Round 3 gave every XCD its own contiguous range of tiles -- eight disjoint sweeps: each record
// panel (2.56 MB per tile row) comes from HBM once per 4-column step of its XCD (39 times per plane) and each digit panel once per
// 8-row group (10 times), 71 GB per launch behind the L2s.  Here the eight XCDs work on ONE super-patch of rb x 8 tile rows by
// (8 / rb) x 4 tile columns at a time (XCD x: row block x % rb, column block x / rb; its 32 tiles rows inside, columns outside as
// before, so the per-XCD L2 patch is unchanged) and the super-patches sweep the columns of a band of rb x 8 tile rows before the
// next band starts: the band's record panels (rb x 20 MB at n = 20 000) stay in the 256 MiB Infinity Cache for the whole
// sweep and a digit panel is fetched from HBM once per band and shared by the rb XCDs that need it at the same time.  HBM reads
// per launch: records once per plane (1.2 GB), digits once per band (24 / rb GB) instead of 71 GB; what the L2s request is
// unchanged.  Ragged bands / column steps give the XCDs sequences of different lengths: tiles are moved from the tails of the
// long ones to the short ones until every XCD has exactly the number of workgroups the hardware will hand it.
static inline void s2_build_raster(int tiles_m, int tiles_n, int rb, std::vector<int2> &map, int PR = 8) {
  const int NX = 8, PC = 32 / PR; // PR x PC = the 32 tiles an XCD works on at a time (8 x 4 unless an experiment says otherwise)
  if (rb != 1 && rb != 2 && rb != 4 && rb != 8) rb = 2;
  const int cbs = NX / rb; // column blocks per super-patch
  std::vector<std::vector<int2>> seq(NX);
  for (int band0 = 0, band = 0; band0 < tiles_m; band0 += rb * PR, ++band)
    for (int col0 = 0; col0 < tiles_n; col0 += cbs * PC)
      for (int xx = 0; xx < NX; ++xx) {
        const int x = (xx + band) % NX; // the XCDs that get the short blocks of a ragged last column step change from band to band
        const int r0 = band0 + (xx % rb) * PR, c0 = col0 + (xx / rb) * PC;
        const int nr = std::min(PR, std::min(tiles_m, band0 + rb * PR) - r0), nc = std::min(PC, std::min(tiles_n, col0 + cbs * PC) - c0);
        if (nr <= 0 || nc <= 0) continue;
        for (int c = 0; c < nc; ++c)
          for (int r = 0; r < nr; ++r) seq[x].push_back(make_int2(r0 + r, c0 + c));
      }
  const int total = tiles_m * tiles_n;
  std::vector<int> want(NX);
  for (int x = 0; x < NX; ++x) want[x] = total / NX + (x < total % NX ? 1 : 0);
  std::vector<int2> spare;
  for (int x = 0; x < NX; ++x)
    while ((int)seq[x].size() > want[x]) {
      spare.push_back(seq[x].back());
      seq[x].pop_back();
    }
  for (int x = 0; x < NX; ++x)
    while ((int)seq[x].size() < want[x]) {
      seq[x].push_back(spare.back());
      spare.pop_back();
    }
  map.assign((size_t)total, make_int2(0, 0));
  for (int x = 0; x < NX; ++x)
    for (int o = 0; o < want[x]; ++o) map[(size_t)o * NX + x] = seq[x][o];
}

// NI: 32-row blocks per wavefront.  NI = 2: wavefronts 4 (rows) x 2 (columns), 2 x 2 blocks each (the first form of the kernel).
// NI = 1: wavefronts 8 x 1, one row block x four column blocks each: every record is unpacked by ONE wavefront instead of two
// (42 instead of 84 VALU per K-tile and wavefront) at the price of 18 instead of 12 ds_read_b128 (the LDS pipe has room).
template <int NI>
__global__ __launch_bounds__(512, 2) void i8gemm_sparse2_kernel_t(Sparse2Args g) {
  constexpr int NJ = 4 / NI;
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
  const int wm = NI == 2 ? wave >> 1 : wave, wn = NI == 2 ? wave & 1 : 0; // rows wm * 32 NI, columns wn * 32 NJ
  const int r32 = lane & 31, h = lane >> 5;

  // LDS-DMA: a stage is 16 record pieces (piece q: rows 16 q .. 16 q + 15, lane l -> row l / 4, chunk l % 4) and 16 digit pieces
  // (piece q: columns 8 q .. 8 q + 7, lane l -> column l / 8, chunk l % 8); wavefront w moves pieces 2 w, 2 w + 1 of each.  The
  // 16-byte chunk index is XOR-ed with (row >> 2) & 3 / (column >> 1) & 7 on the SOURCE address (LDS-DMA writes lane-linearly)
  // and on the fragment reads: conflict-free ds_read_b128.
  const uint4 *asrc[2];
  const int8_t *bsrc[2];
  int adst[2], bdst[2];
#define S2_INIT_SRC(DIGIT)                                                                                        \
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
  // fragment byte offsets inside a stage
  int amo[2], fb[4];
  {
    const int row = wm * 32 * NI + r32; // block 1: + 32 rows = + 2048 bytes, same swizzle
#pragma unroll
    for (int p = 0; p < 2; ++p) amo[p] = row * 64 + (((2 * p + h) ^ ((row >> 2) & 3)) << 4);
    const int col = wn * 32 * NJ + r32; // block j: + 32 j columns = + 4096 j bytes (same swizzle)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb[ks] = S2_AMB + col * 128 + (((2 * ks + h) ^ ((col >> 1) & 7)) << 4);
  }

  i32x16 accg[NI][NJ], accm[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accg[i][j][r] = 0; accm[i][j][r] = 0; }

  i32x4 am[2][NI]; // records [pair parity][row block]
  i32x4 ga[2][NI]; // genotype operands [K-step parity][row block]
  i32x4 bb[2][NJ]; // digit fragments [K-step parity][column block]
  i32x4 ms[NI];    // sparse operand values of the current pair [row block]

#define S2_DMA_A(j, SOFF)                                                                                         \
  do {                                                                                                            \
    __builtin_amdgcn_global_load_lds((gemma_gptr_t)asrc[j], (gemma_lptr_t)(i8lds + (SOFF) + adst[j]), 16, 0, 0);  \
    if (S2_ADVANCE) asrc[j] += S2_BM * 4;                                                                         \
  } while (0)
#define S2_DMA_B(j, SOFF)                                                                                         \
  do {                                                                                                            \
    __builtin_amdgcn_global_load_lds((gemma_gptr_t)bsrc[j], (gemma_lptr_t)(i8lds + (SOFF) + bdst[j]), 16, 0, 0);  \
    if (S2_ADVANCE) bsrc[j] += I8_BK;                                                                             \
  } while (0)
#define S2_RAM(SOFF, P, i) am[(P)&1][i] = *reinterpret_cast<const i32x4 *>(i8lds + (SOFF) + amo[(P)&1] + (i) * 2048)
#define S2_RB(SOFF, KS, j) bb[(KS)&1][j] = *reinterpret_cast<const i32x4 *>(i8lds + (SOFF) + fb[KS] + (j) * 4096)
#define S2_UNP(KS, i) ga[(KS)&1][i] = s2_unpack_g(am[((KS) >> 1) & 1][i][(KS)&1])
#define S2_EXP(P, i) ms[i] = s2_expand(am[(P)&1][i][3])
// dense genotype MFMA of K-step KS, block (i, j); sparse mask MFMA of pair P, block (i, j)
#define S2_G(KS, i, j)                                                                                            \
  asm volatile("s_nop 1\n\tv_mfma_i32_32x32x32_i8 %0, %1, %2, %0"                                                 \
               : "+v"(accg[i][j]) : "v"(ga[(KS)&1][i]), "v"(bb[(KS)&1][j]))
#define S2_S(P, i, j)                                                                                             \
  do {                                                                                                            \
    const i32x8 bp_ = __builtin_shufflevector(bb[0][j], bb[1][j], 0, 1, 2, 3, 4, 5, 6, 7);                         \
    asm volatile("s_nop 1\n\tv_smfmac_i32_32x32x64_i8 %0, %1, %2, %3"                                             \
                 : "+v"(accm[i][j]) : "v"(ms[i]), "v"(bp_), "v"(am[(P)&1][i][2]));                                 \
  } while (0)
// One K-tile from stage SC.  MORE: tile t+1 exists (stage SN); LOAD3: tile t+3 exists and goes to stage SD; VMW: LDS-DMA pieces
// that may still be in flight when tile t+1 must have landed (4 per tile behind it).  At entry: records of pair 0 in am[0],
// genotype operand of step 0 in ga[0], digit fragments of step 0 in bb[0].  24 matrix instructions in six groups of four; the
// LDS reads of a group are issued behind its first two instructions and consumed by the NEXT group, whose first instruction
// carries the only s_waitcnt lgkmcnt of the group -- the compiler drains the counter to zero at every wait while an LDS-DMA is
// in flight (a FLAT instruction that touches both memories), so no read may be younger than two matrix instructions at a wait;
// VALU that needs a landed record comes BEFORE the reads of its slot for the same reason.
#define S2_KTILE(SC, SN, SD, MORE, LOAD3, VMW)                                                                    \
  do {                                                                                                            \
    /* group 0, step 0: genotype product; reads: digit fragments of step 1; VALU: sparse operand of pair 0 */     \
    S2_G(0, 0, 0); S2_RB(SC, 1, 0); GEMMA_SB();                                                                   \
    S2_G(0, 0, 1); S2_RB(SC, 1, 1); S2_EXP(0, 0); GEMMA_SB();                                                     \
    S2_G(0, 1, 0); if ((LOAD3) && S2_LOOPDMA) S2_DMA_A(0, SD); GEMMA_SB();                                                        \
    S2_G(0, 1, 1); S2_EXP(0, 1); GEMMA_SB();                                                                      \
    /* group 1, step 1: mask product of pair 0; reads: records of pair 1; VALU: genotype operand of step 1 */     \
    S2_S(0, 0, 0); S2_RAM(SC, 1, 0); GEMMA_SB();                                                                  \
    S2_S(0, 0, 1); S2_RAM(SC, 1, 1); S2_UNP(1, 0); GEMMA_SB();                                                    \
    S2_S(0, 1, 0); if ((LOAD3) && S2_LOOPDMA) S2_DMA_A(1, SD); GEMMA_SB();                                                        \
    S2_S(0, 1, 1); S2_UNP(1, 1); GEMMA_SB();                                                                      \
    /* group 2, step 1: genotype product; VALU: genotype operand of step 2; reads: digit fragments of step 2 */    \
    S2_G(1, 0, 0); S2_UNP(2, 0); GEMMA_SB(); S2_RB(SC, 2, 0); GEMMA_SB();                                         \
    S2_G(1, 0, 1); S2_RB(SC, 2, 1); GEMMA_SB();                                                                   \
    S2_G(1, 1, 0); S2_UNP(2, 1); if ((LOAD3) && S2_LOOPDMA) S2_DMA_B(0, SD); GEMMA_SB();                                          \
    S2_G(1, 1, 1); if ((LOAD3) && S2_LOOPDMA) S2_DMA_B(1, SD); GEMMA_SB();                                                        \
    /* group 3, step 2: genotype product; reads: digit fragments of step 3 (the last reads of this stage), then the */ \
    /* rendezvous for tile t+1; VALU: sparse operand of pair 1 */                                                 \
    S2_G(2, 0, 0); S2_RB(SC, 3, 0); GEMMA_SB();                                                                   \
    S2_G(2, 0, 1); S2_RB(SC, 3, 1); GEMMA_SB();                                                                   \
    asm volatile("s_waitcnt vmcnt(" #VMW ")" ::: "memory");                                                       \
    __builtin_amdgcn_s_barrier();                                                                                 \
    GEMMA_SB();                                                                                                   \
    S2_G(2, 1, 0); S2_EXP(1, 0); GEMMA_SB();                                                                      \
    S2_G(2, 1, 1); S2_EXP(1, 1); GEMMA_SB();                                                                      \
    /* group 4, step 3: mask product of pair 1; reads: records of pair 0 of tile t+1; VALU: genotype operand of step 3 */ \
    S2_S(1, 0, 0); if (MORE) S2_RAM(SN, 0, 0); GEMMA_SB();                                                        \
    S2_S(1, 0, 1); if (MORE) S2_RAM(SN, 0, 1); GEMMA_SB();                                                        \
    S2_S(1, 1, 0); S2_UNP(3, 0); GEMMA_SB();                                                                      \
    S2_S(1, 1, 1); S2_UNP(3, 1); GEMMA_SB();                                                                      \
    /* group 5, step 3: genotype product; VALU: genotype operand of step 0 of tile t+1; reads: its digit fragments */ \
    S2_G(3, 0, 0); if (MORE) S2_UNP(0, 0); GEMMA_SB();                                                            \
    if (MORE) { S2_RB(SN, 0, 0); S2_RB(SN, 0, 1); } GEMMA_SB();                                                   \
    S2_G(3, 0, 1); if (MORE) S2_UNP(0, 1); GEMMA_SB();                                                            \
    S2_G(3, 1, 0); GEMMA_SB();                                                                                    \
    S2_G(3, 1, 1); GEMMA_SB();                                                                                    \
  } while (0)

// The same K-tile for NI = 1 (one row block x four column blocks): groups G(0) | S(P0) | G(1) | G(2) | S(P1) | G(3) over the
// column blocks j = 0..3; the eight digit-fragment reads of a pair of steps go two per slot behind the first two instructions
// of a group (no read younger than two matrix instructions at the next group's wait).
#define S2_KTILE_81(SC, SN, SD, MORE, LOAD3, VMW)                                                                 \
  do {                                                                                                            \
    S2_G(0, 0, 0); S2_RB(SC, 1, 0); S2_RB(SC, 1, 1); GEMMA_SB();                                                  \
    S2_G(0, 0, 1); S2_RB(SC, 1, 2); S2_RB(SC, 1, 3); GEMMA_SB();                                                  \
    S2_G(0, 0, 2); S2_EXP(0, 0); if ((LOAD3) && S2_LOOPDMA) S2_DMA_A(0, SD); GEMMA_SB();                          \
    S2_G(0, 0, 3); GEMMA_SB();                                                                                    \
    S2_S(0, 0, 0); S2_RAM(SC, 1, 0); GEMMA_SB();                                                                  \
    S2_S(0, 0, 1); S2_UNP(1, 0); GEMMA_SB();                                                                      \
    S2_S(0, 0, 2); if ((LOAD3) && S2_LOOPDMA) S2_DMA_A(1, SD); GEMMA_SB();                                        \
    S2_S(0, 0, 3); GEMMA_SB();                                                                                    \
    S2_G(1, 0, 0); S2_UNP(2, 0); GEMMA_SB(); S2_RB(SC, 2, 0); S2_RB(SC, 2, 1); GEMMA_SB();                        \
    S2_G(1, 0, 1); S2_RB(SC, 2, 2); S2_RB(SC, 2, 3); GEMMA_SB();                                                  \
    S2_G(1, 0, 2); if ((LOAD3) && S2_LOOPDMA) S2_DMA_B(0, SD); GEMMA_SB();                                        \
    S2_G(1, 0, 3); if ((LOAD3) && S2_LOOPDMA) S2_DMA_B(1, SD); GEMMA_SB();                                        \
    S2_G(2, 0, 0); S2_RB(SC, 3, 0); S2_RB(SC, 3, 1); GEMMA_SB();                                                  \
    S2_G(2, 0, 1); S2_RB(SC, 3, 2); S2_RB(SC, 3, 3); GEMMA_SB();                                                  \
    asm volatile("s_waitcnt vmcnt(" #VMW ")" ::: "memory");                                                       \
    __builtin_amdgcn_s_barrier();                                                                                 \
    GEMMA_SB();                                                                                                   \
    S2_G(2, 0, 2); S2_EXP(1, 0); GEMMA_SB();                                                                      \
    S2_G(2, 0, 3); GEMMA_SB();                                                                                    \
    S2_S(1, 0, 0); if (MORE) S2_RAM(SN, 0, 0); GEMMA_SB();                                                        \
    S2_S(1, 0, 1); S2_UNP(3, 0); GEMMA_SB();                                                                      \
    S2_S(1, 0, 2); GEMMA_SB();                                                                                    \
    S2_S(1, 0, 3); GEMMA_SB();                                                                                    \
    S2_G(3, 0, 0); if (MORE) S2_UNP(0, 0); GEMMA_SB();                                                            \
    if (MORE) { S2_RB(SN, 0, 0); S2_RB(SN, 0, 1); } GEMMA_SB();                                                   \
    S2_G(3, 0, 1); if (MORE) { S2_RB(SN, 0, 2); S2_RB(SN, 0, 3); } GEMMA_SB();                                    \
    S2_G(3, 0, 2); GEMMA_SB();                                                                                    \
    S2_G(3, 0, 3); GEMMA_SB();                                                                                    \
  } while (0)
#define S2_TILE(SC, SN, SD, MORE, LOAD3, VMW)                                                                     \
  do {                                                                                                            \
    if constexpr (NI == 2) S2_KTILE(SC, SN, SD, MORE, LOAD3, VMW);                                                \
    else S2_KTILE_81(SC, SN, SD, MORE, LOAD3, VMW);                                                               \
  } while (0)
  // the second-dispatched half of the workgroup loses every issue arbitration on age; one static priority step evens it out
  // (MI355X_MICROARCH.md, two waves per SIMD, item 4): 57.2 -> 56.2-56.9 ms (profiles/r03_i8_sparse_ablation.txt)
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  const int nk = g.nk;
  for (int dd = 0; dd < nd; ++dd) {
    if (dd > 0) { // second digit of a fused pair: acc = 256 * C_hi, then accumulate C_lo on top
      asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) { accg[i][j][r] <<= 8; accm[i][j][r] <<= 8; }
    }
    S2_INIT_SRC(d_first - dd);
    // prologue: tiles 0, 1, 2 in flight, tile 0 landed
#pragma unroll
    for (int j = 0; j < 2; ++j) { S2_DMA_A(j, 0); S2_DMA_B(j, 0); }
    if (nk > 1) {
#pragma unroll
      for (int j = 0; j < 2; ++j) { S2_DMA_A(j, S2_STAGE); S2_DMA_B(j, S2_STAGE); }
    }
    if (nk > 2) {
#pragma unroll
      for (int j = 0; j < 2; ++j) { S2_DMA_A(j, 2 * S2_STAGE); S2_DMA_B(j, 2 * S2_STAGE); }
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if (nk > 1) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    GEMMA_SB();
#pragma unroll
    for (int i = 0; i < NI; ++i) S2_RAM(0, 0, i);
#pragma unroll
    for (int j = 0; j < NJ; ++j) S2_RB(0, 0, j);
#pragma unroll
    for (int i = 0; i < NI; ++i) S2_UNP(0, i);
    GEMMA_SB();

    int sc = 0, sn = S2_STAGE, s2 = 2 * S2_STAGE, sd = 3 * S2_STAGE; // stage byte offsets: tiles t, t+1, t+2, DMA target
    int kt = 0;
    for (; kt + 3 < nk; ++kt) {
      S2_TILE(sc, sn, sd, true, true, 8);
      const int tmp = sc; sc = sn; sn = s2; s2 = sd; sd = tmp;
    }
    if (nk >= 3) {
      S2_TILE(sc, sn, sd, true, false, 4);
      const int tmp = sc; sc = sn; sn = s2; s2 = sd; sd = tmp;
    }
    if (nk >= 2) {
      S2_TILE(sc, sn, sd, true, false, 0);
      const int tmp = sc; sc = sn; sn = s2; s2 = sd; sd = tmp;
    }
    S2_TILE(sc, sn, sd, false, false, 0);
  }
#undef S2_INIT_SRC
#undef S2_DMA_A
#undef S2_DMA_B
#undef S2_RAM
#undef S2_RB
#undef S2_UNP
#undef S2_EXP
#undef S2_G
#undef S2_S
#undef S2_KTILE
#undef S2_KTILE_81
#undef S2_TILE

  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
  int *Cg = g.C + (long)plane * g.strideC;
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const long col = (long)tn * S2_BN + wn * 32 * NJ + j * 32 + r32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long row = (long)tm * S2_BM + wm * 32 * NI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        Cg[row * g.ldc + col] = accg[i][j][r];
        Cg[(g.m_row0 + row) * g.ldc + col] = accm[i][j][r];
      }
    }
}

// the shipped form: wavefronts 8 x 1 (54.4-55.0 ms against 56.5 for 4 x 2 in the same session, profiles/r03_i8_sparse_ablation.txt)
#ifndef S2_DEFAULT_RASTER
#define S2_DEFAULT_RASTER 1 // row blocks of the cross-XCD super-patch (0: the per-XCD ranges of round 3)
#endif
#ifndef S2_DEFAULT_NI
#define S2_DEFAULT_NI 1
#endif
static constexpr auto i8gemm_sparse2_kernel = i8gemm_sparse2_kernel_t<S2_DEFAULT_NI>;

} // namespace gemma_hip
