// PROTOTYPE, not compiled into the library (round 6, lease 22; profiles/r06_kin_corr3_planes.txt): the genotype term of the kinship
// correction with a SNP's genotypes as two bit planes over the individuals that become EXECUTION MASKS of two v_add_f64 -- 8 VALU cycles
// per 64 (missing call, individual) pairs where kin_i8_corr2_kernel's extract + convert + FMA take ~15.  Exact (the 37 kinship tests pass
// with it) and TWICE AS SLOW: 16.2 ms per 20 000-SNP block against 7.97.  The masks must be wave-uniform, i.e. arrive through the scalar
// cache: one 64-byte line per (entry, 4 words), ~1 us from the L2 / Infinity Cache (the planes are 100 MB), and a wavefront's ~100 SGPRs
// hold at most one entry's 16 words -- a CU has < 13 KB of scalar registers to keep in flight where the stream needs ~20 KB per CU to
// cover that latency at 5 TB/s.  The SIMDs sit at 27 % VALU issue.  The vector-memory form (corr2) keeps 8 loads per lane in flight.
// Paste between kin_i8_corr2_kernel and kin_i8_fold_kernel of gemma_amd/csrc/kin_i8.hip.h to rebuild it; launch as in scripts/exp/r6_22.sh's commit.
// ---- the genotype term of the correction on BIT PLANES (round 6) ---------------------------------------------------------
// kin_i8_corr2_kernel pays a bit-field extract, a conversion and an FMA per (missing call of j, individual i) pair: ~15 cycles of a
// SIMD per 64 pairs.  With a SNP's genotypes as two bit planes over the individuals -- p1 = (g >= 1), p2 = (g == 2), one 64-bit word
// per 64 individuals -- mu g is "mu under p1, mu again under p2": the planes are wave-uniform, so they arrive through the SCALAR
// cache and become the execution mask of two v_add_f64 (8 cycles per 64 pairs, no extract, no conversion).  The lanes ARE the
// individuals: wave w of the block (j, seg) owns the 16 words KI8_SEG seg + 1024 w + 64 q + lane, q = 0 .. 15.
// A (l x ldk bytes, g | m << 4) -> planes (l x 64 nseg uint4): word 64 seg + w of row s = {p1 lo, p1 hi, p2 lo, p2 hi}
__global__ __launch_bounds__(256) void kin_i8_planes_kernel(const int8_t *__restrict__ A, long l, long ldk, int nseg,
                                                            uint4 *__restrict__ planes) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, seg = blockIdx.y;
  const long s = blockIdx.x;
  const int8_t *row = A + s * ldk;
  uint4 mine = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const long i = (long)seg * KI8_SEG + 1024 * wave + 64 * q + lane;
    const unsigned g2 = i < ldk ? ((unsigned)(unsigned char)row[i] & 3u) : 0u;
    const unsigned long long p1 = __ballot(g2 >= 1u), p2 = __ballot(g2 == 2u);
    if (lane == q) mine = make_uint4((unsigned)p1, (unsigned)(p1 >> 32), (unsigned)p2, (unsigned)(p2 >> 32));
  }
  if (lane < 16) planes[s * (64L * nseg) + 64 * seg + 16 * wave + lane] = mine;
}

struct KinCorr3Args {
  const uint4 *planes; // l x (64 nseg)
  long ldp;            // uint4 per row = 64 nseg
  const double *mean;
  long n;
  const int *offJ, *listJ, *offS, *listS, *sub;
  int nseg;
  const double *cj;
  double *S;
  const int *ok;
};
// acc += mu under the lanes of p1, then under the lanes of p2 (exec restored afterwards; the block has no divergence here)
#define KI8_MASKED_ADD4(A0, A1, A2, A3, W0, W1, W2, W3, MU)                                                                   \
  asm("s_mov_b64 %[sv], exec\n\t"                                                                                     \
               "s_mov_b64 exec, %[a1]\n\tv_add_f64 %[x0], %[x0], %[mu]\n\ts_mov_b64 exec, %[a2]\n\tv_add_f64 %[x0], %[x0], %[mu]\n\t" \
               "s_mov_b64 exec, %[b1]\n\tv_add_f64 %[x1], %[x1], %[mu]\n\ts_mov_b64 exec, %[b2]\n\tv_add_f64 %[x1], %[x1], %[mu]\n\t" \
               "s_mov_b64 exec, %[c1]\n\tv_add_f64 %[x2], %[x2], %[mu]\n\ts_mov_b64 exec, %[c2]\n\tv_add_f64 %[x2], %[x2], %[mu]\n\t" \
               "s_mov_b64 exec, %[d1]\n\tv_add_f64 %[x3], %[x3], %[mu]\n\ts_mov_b64 exec, %[d2]\n\tv_add_f64 %[x3], %[x3], %[mu]\n\t" \
               "s_mov_b64 exec, %[sv]"                                                                                         \
               : [x0] "+v"(A0), [x1] "+v"(A1), [x2] "+v"(A2), [x3] "+v"(A3), [sv] "=&s"(sv_)                                    \
               : [a1] "s"(ki8_lo64(W0)), [a2] "s"(ki8_hi64(W0)), [b1] "s"(ki8_lo64(W1)), [b2] "s"(ki8_hi64(W1)),                \
                 [c1] "s"(ki8_lo64(W2)), [c2] "s"(ki8_hi64(W2)), [d1] "s"(ki8_lo64(W3)), [d2] "s"(ki8_hi64(W3)), [mu] "s"(MU))
__device__ __forceinline__ unsigned long long ki8_lo64(const uint4 w) { return ((unsigned long long)w.y << 32) | w.x; }
__device__ __forceinline__ unsigned long long ki8_hi64(const uint4 w) { return ((unsigned long long)w.w << 32) | w.z; }
// grid (n individuals j, nseg ranges of KI8_SEG individuals), 256 threads
__global__ __launch_bounds__(256) void kin_i8_corr3_kernel(KinCorr3Args g) {
  __shared__ unsigned long long trow[KI8_SEG];
  if (!g.ok[0]) return;
  const long j = blockIdx.x;
  const int seg = blockIdx.y, t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const long i0 = (long)seg * KI8_SEG;
#pragma unroll
  for (int q = 0; q < KI8_SEG / 256; ++q) trow[t + 256 * q] = 0ull;
  const int lo = g.offJ[j], cnt = g.offJ[j + 1] - lo;
  const int *lst = g.listJ + lo;
  double acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0;
  const uint4 *pw = g.planes + 64 * seg + 16 * wave; // this wave's 16 words of a row
  unsigned long long sv_;
  // the list, the means and the planes are wave-uniform: scalar loads.  Two halves of eight words per entry, the next half on its way
  // (and the next entry's SNP index and mean) while the lanes add the current one
  if (cnt > 0) {
    int s0 = lst[0];
    long long mub = __double_as_longlong(g.mean[s0]);
    const uint4 *pr = pw + (long)s0 * g.ldp;
    uint4 wa[8], wb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) wa[q] = pr[q];
    for (int e = 0; e < cnt; ++e) {
      const int s1 = lst[e + 1 < cnt ? e + 1 : e];
      const long long mub1 = __double_as_longlong(g.mean[s1]);
#pragma unroll
      for (int q = 0; q < 8; ++q) wb[q] = pr[8 + q];
      KI8_MASKED_ADD4(acc[0], acc[1], acc[2], acc[3], wa[0], wa[1], wa[2], wa[3], mub);
      KI8_MASKED_ADD4(acc[4], acc[5], acc[6], acc[7], wa[4], wa[5], wa[6], wa[7], mub);
      pr = pw + (long)s1 * g.ldp;
#pragma unroll
      for (int q = 0; q < 8; ++q) wa[q] = pr[q];
      KI8_MASKED_ADD4(acc[8], acc[9], acc[10], acc[11], wb[0], wb[1], wb[2], wb[3], mub);
      KI8_MASKED_ADD4(acc[12], acc[13], acc[14], acc[15], wb[4], wb[5], wb[6], wb[7], mub);
      mub = mub1;
    }
  }
  __syncthreads(); // trow is cleared
  { // both missing: as in kin_i8_corr2_kernel
    for (int e0 = 64 * wave; e0 < cnt; e0 += 256) {
      const int e = e0 + lane;
      int b0 = 0, b1 = 0, off = 0;
      unsigned long long h = 0ull;
      if (e < cnt) {
        const int s = lst[e];
        const int *sb = g.sub + (long)s * (g.nseg + 1) + seg;
        b0 = sb[0];
        b1 = sb[1];
        off = g.offS[s];
        const double mu = g.mean[s];
        h = (unsigned long long)__double2ll_rn(0.5 * mu * mu * KI8_FIX);
      }
      const int cntw = cnt - e0 < 64 ? cnt - e0 : 64;
      for (int u = 0; u < cntw; u += 4) {
        int ii[4], ub1[4], uk[4], uoff[4];
        unsigned long long hh[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int uu = (u + q) & 63;
          uoff[q] = __shfl(off, uu, 64);
          ub1[q] = __shfl(b1, uu, 64);
          uk[q] = __shfl(b0, uu, 64) + lane;
          hh[q] = (unsigned long long)__shfl((long long)h, uu, 64);
          ii[q] = uk[q] < ub1[q] ? g.listS[uoff[q] + uk[q]] : 0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (uk[q] < ub1[q]) atomicAdd(&trow[ii[q] - (int)i0], hh[q]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          for (int k = uk[q] + 64; k < ub1[q]; k += 64) atomicAdd(&trow[g.listS[uoff[q] + k] - (int)i0], hh[q]);
      }
    }
  }
  __syncthreads();
  double *Sj = g.S + j * g.n;
  const double cj = g.cj[j];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int o = 1024 * wave + 64 * q + lane;
    const long i = i0 + o;
    if (i < g.n) Sj[i] += acc[q] + (double)trow[o] * (1.0 / KI8_FIX) - cj;
  }
}
#undef KI8_MASKED_ADD4

