// Two-stage reduction for the symmetric eigensolver (included by eigh.hip.h): dense -> band -> tridiagonal.
//
// The one-stage tridiagonalisation of eigh.hip.h streams the trailing matrix from HBM once per column (the SYMV half of
// its flops: (8/3) n^3 bytes, 2.4 s of 4.4 s at n = 20 000, 45+ s at n = 50 000).  Here the dense matrix is touched by
// GEMMs only:
//
//  stage 1 (sy2sb)  panels of E2_B = 128 columns: Householder QR of the block below the band (the panel is read as the
//                   contiguous ROWS j0 .. j0+127 of the symmetric matrix; one launch per column, every workgroup owns
//                   256 columns of the panel, the column norms and the v.y products of the next reflector ride on the
//                   update of the previous one), compact-WY factor T, and the two-sided update
//                   A22 -= V W^T + W V^T with W = Y - V (T^T (V^T Y)) / 2, Y = A22 V T  -- all on the fp64 MFMA GEMM.
//  stage 2 (sb2st)  bulge chasing on the band (n x 256 doubles: L2 / MALL resident).  Task (j, k) of sweep j works on
//                   rows j+1+128k ..: it applies the previous task's reflector from the right to the 128 x 128 block
//                   left of its diagonal block, annihilates that block's first column, and updates its diagonal block
//                   from both sides.  Tasks with 2 j + k = t are independent (scripts/two_stage_model.py checks the
//                   schedule and the group order below in numpy).  ONE persistent launch: workgroup w owns the chase
//                   positions 2w, 2w+1, walks the sweeps and waits for its neighbours on progress counters (bounded
//                   waits; fallback: one launch per time step).  Inside a task both blocks stay in registers from the
//                   global load to the store; only the two column sums go through LDS.
//  back-transform   Z^T <- Z^T Q2^T Q1^T.  Q2 (the n^2 / 256 reflectors of stage 2): 32 consecutive sweeps at the same
//                   k form one block reflector I - V T V^T whose V is a 160 x 32 parallelogram; groups are applied with k
//                   ascending outside and the sweep blocks descending inside, so the 160-column window of Z^T slides by
//                   32 columns per group.  One wavefront owns 16 rows of Z^T and keeps its window in REGISTERS in the
//                   accumulator layout of the transposed products (X^T tiles), which is also the MFMA operand layout:
//                   W^T = V^T X^T, W2^T = T W^T, X^T -= V W2^T chain without touching LDS; only V and T (52 KB per group,
//                   packed once by q2_pack_kernel) go through LDS.  Q1 (stage 1) reuses the three-GEMM panel update of
//                   eigh.hip.h, two panels (256 reflectors) per block reflector.
//
// LAPACK equivalents: dsytrd_sy2sb / dsytrd_sb2st / dormtr-like back-transformations; semantics of the whole solver as
// before (GEMMA src/lapack.cpp:149-291).
#pragma once

namespace gemma_hip {

typedef double e2_v4 __attribute__((ext_vector_type(4)));
typedef double e2_v2 __attribute__((ext_vector_type(2)));
constexpr int E2_B = EIG_NB;             // half-bandwidth after stage 1 (= the back-transform block of eigh.hip.h)
constexpr int E2_LDB = 2 * E2_B;         // doubles per column of the band storage: Bd[j * E2_LDB + (i - j)] = B(i, j), i >= j
constexpr int E2_NB = 32;                // sweeps per block reflector of the stage-2 back-transformation
constexpr int E2_WIN = E2_B + E2_NB;     // window of Z^T columns one group touches (b + nb - 1, rounded up to 16)
constexpr int E2_VLD = 34;               // leading dimension of the packed group (bank-conflict padding)
constexpr int E2_PACK = (E2_WIN + E2_NB) * E2_VLD; // doubles per packed group: V (160 x 34) then T (32 x 34)
constexpr int SB_COLS = 256;             // panel columns per workgroup of sb_panel_kernel
constexpr int GR_CH = 256, GR_KS = 16;   // Gram kernel: columns per workgroup (n / 256 workgroups per call), columns per LDS step

__device__ __forceinline__ double e2_bsum256(double v, double *red /* 8 doubles */) {
  v = eig_wsum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// LAPACK dlarfg from alpha = x[0] and |x[1:]|^2
__device__ __forceinline__ void e2_larfg(double alpha, double xnorm2, double &tau, double &beta, double &scale) {
  if (xnorm2 == 0.0) {
    tau = 0.0;
    beta = alpha;
    scale = 0.0;
    return;
  }
  double b = sqrt(alpha * alpha + xnorm2);
  if (alpha > 0.0) b = -b;
  tau = (b - alpha) / b;
  scale = 1.0 / (alpha - b);
  beta = b;
}

// Bounded by WALL CLOCK (the constant 100 MHz counter), not by a poll count: several processes can time-slice one device (ranks
// that share it in the tests, bench.py's children), and a predecessor that is merely descheduled must not turn into a failure.
// BC_WAIT_TICKS: the chase, whose co-residency is assumed, not guaranteed -- a real dead-lock is possible there and the per-step
// fall-back takes over; Q2_WAIT_TICKS: the stage-2 back-transformation, where the awaited task was claimed earlier and is running
// by construction, so the bound only has to outlast any descheduling.
constexpr long long BC_WAIT_TICKS = 400000000LL;   // 4 s
constexpr long long Q2_WAIT_TICKS = 6000000000LL;  // 60 s
__device__ __forceinline__ bool bc_wait(int *p, int target, int *err, long long max_ticks = BC_WAIT_TICKS) {
  long long t0 = 0;
  for (long it = 0;; ++it) {
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
    if ((it & 63) == 63) {
      if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
      const long long now = (long long)wall_clock64();
      if (t0 == 0) t0 = now;
      else if (now - t0 > max_ticks) break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return false;
}

// ---------------------------------------------------------------- stage 1: panel QR
// Launch c of panel j0 (c = 0 .. kk): applies reflector c-1 to the rows c .. 127 of the panel (columns >= its head) and
// stores it (row j0+c-1 of VT, head element 1), then forms the partial sums that define reflector c: for every row q >= c
// sum over the columns right of head_c of x_c[col] y_q[col] (q = c: |x_c[1:]|^2).  The head column's entries are kept in
// `heads` by the workgroup that owns it, so that nothing is read in place while another workgroup rewrites it.
struct SbPanelArgs {
  double *A;
  long n, j0;
  int c, kk, nwg;
  double *VT, *part, *heads, *tau, *betas;
};
__global__ __launch_bounds__(256) void sb_panel_kernel(SbPanelArgs g) {
  __shared__ double sf[E2_B], sred[E2_B];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const long n = g.n, j0 = g.j0, r0 = j0 + E2_B;
  const int c = g.c, prev = c - 1;
  const long head_prev = r0 + prev, head_c = r0 + c;
  const bool more = c < g.kk;
  const long base = r0 + (long)blockIdx.x * SB_COLS + lane;
  // every global load this launch depends on is issued here, in one round trip: the previous launch's partial sums and
  // head entries, row c-1 (the pending reflector), row c (the next one)
  // loads are unconditional (column clamped into the row, row clamped into the panel): a predicated load becomes its own
  // branch region and a select on its value a wait right behind it; what a clamped lane reads is masked where it is used
  bool valid[4];
  long ccol[4];
  double xprev[4], yc[4];
  const int prow = c > 0 ? prev : 0, crow = more ? c : (E2_B - 1);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long col = base + 64 * i;
    valid[i] = col < n;
    ccol[i] = valid[i] ? col : n - 1;
    xprev[i] = g.A[(j0 + prow) * n + ccol[i]];
    yc[i] = g.A[(j0 + crow) * n + ccol[i]];
  }
  const int qfirst = c + ((wave - c) & 3);
  double ynext[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int q = qfirst + 4 * u;
    const double *row = g.A + (j0 + (q < E2_B ? q : E2_B - 1)) * n;
#pragma unroll
    for (int i = 0; i < 4; ++i) ynext[u][i] = row[ccol[i]];
  }
  double scale = 0.0;
  if (c > 0) {
    const double alpha = g.heads[(prev & 1) * E2_B + prev];
    const double myhead = (t < E2_B && t > prev) ? g.heads[(prev & 1) * E2_B + t] : 0.0;
    if (t < E2_B && t >= prev) {
      const double *pp = g.part + (size_t)(prev & 1) * g.nwg * E2_B + t;
      double s = 0.0;
#pragma unroll 16
      for (int w = 0; w < g.nwg; ++w) s += pp[(size_t)w * E2_B];
      sred[t] = s;
    }
    __syncthreads();
    double tau, beta;
    e2_larfg(alpha, sred[prev], tau, beta, scale); // every thread: no second barrier for three scalars
    if (t == 0 && blockIdx.x == 0) {
      g.tau[j0 + prev] = tau;
      g.betas[j0 + prev] = beta;
    }
    if (t < E2_B && t > prev) sf[t] = tau * (myhead + scale * sred[t]);
    __syncthreads();
  }
  double v[4], xc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long col = base + 64 * i;
    v[i] = 0.0;
    xc[i] = 0.0;
    if (c > 0 && valid[i] && col >= head_prev) {
      v[i] = (col == head_prev) ? 1.0 : scale * xprev[i];
      if (wave == 0) g.VT[(j0 + prev) * n + col] = v[i];
    }
  }
  if (more) { // row c after the pending update: x of the next reflector (every wavefront needs it for its own columns)
    const double f = (c > 0) ? sf[c] : 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) xc[i] = (valid[i] && base + 64 * i > head_c) ? yc[i] - f * v[i] : 0.0;
  }
  // this wavefront's rows q = q0, q0 + 4, ...: four rows per step, all loads of a step in flight together and the next
  // step's loads issued before this step's arithmetic (the first step's were issued before the barriers above)
  for (int q0 = qfirst; q0 < E2_B; q0 += 16) {
    double y[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) y[u][i] = ynext[u][i];
    if (q0 + 16 < E2_B) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = q0 + 16 + 4 * u;
        const double *row = g.A + (j0 + (q < E2_B ? q : E2_B - 1)) * n;
#pragma unroll
        for (int i = 0; i < 4; ++i) ynext[u][i] = row[ccol[i]];
      }
    }
    double pr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = q0 + 4 * u;
      pr[u] = 0.0;
      if (q < E2_B) {
        double *row = g.A + (j0 + q) * n;
        const double f = (c > 0) ? sf[q] : 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          y[u][i] -= f * v[i];
          if (c > 0 && valid[i] && base + 64 * i >= head_prev) row[base + 64 * i] = y[u][i];
          pr[u] += xc[i] * y[u][i];
          if (more && valid[i] && base + 64 * i == head_c) g.heads[(c & 1) * E2_B + q] = y[u][i];
        }
      }
    }
    if (more) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int u = 0; u < 4; ++u) pr[u] += __shfl_xor(pr[u], off, 64);
      if (lane == 0)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (q0 + 4 * u < E2_B) sred[q0 + 4 * u] = pr[u];
    }
  }
  if (more) {
    __syncthreads();
    if (t < E2_B && t >= c) g.part[(size_t)(c & 1) * g.nwg * E2_B + (size_t)blockIdx.x * E2_B + t] = sred[t];
  }
}

// ---------------------------------------------------------------- stage 1: panel QR, ONE launch per panel (round 4)
// The launches above are a chain of 129 dependent kernels per panel (n in total: 0.29 s of the 0.80 s of stage 1 at n = 20 000,
// 14.5 us each: launch + the panel re-read from L2 / MALL + its write-back at every kernel boundary).  Here the panel stays in
// REGISTERS for all 129 steps: a workgroup owns 128 NCH columns, thread (q, h) row q and the columns 32 (4 k + h) .. + 31 of
// chunk k -- 32 NCH doubles.  A step applies reflector c - 1 to the thread's row (rank-1 update with v broadcast from LDS),
// publishes row c as x_c, and forms the thread's share of x_c . y_q: 32 NCH serial fused multiply-adds, no shuffles; the four
// column parts meet in LDS.  Between steps only the (128 - c) partial sums per workgroup and the head column's entries go
// through memory, in self-validating slots (see SbPersistArgs; wall-clock-bounded polls, error flag -> the host repeats the
// panel with the per-column launches: A is written only at the very end).  The
// arithmetic is the per-column kernel's except for the grouping of the partial sums (128 NCH columns per workgroup, summed in
// workgroup order).  Needs every workgroup resident: nwg <= number of CUs.  Only NCH = 1 (128 columns per workgroup) is used:
// with 256 columns per workgroup the kernel spills and loses to the launches (measured at n = 50 000), so panels wider than
// 128 x CUs columns keep the per-column launches until the trailing matrix has shrunk.
// data exchanged between workgroups of one launch (partial sums, head entries): relaxed atomics at agent scope -- on gfx950 a
// store with sc1 (written through to the memory side) and a load with sc1 (served from there, not from the XCD's L2)
__device__ __forceinline__ double sp_ld(const double *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sp_st(double *p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
struct SbPersistArgs {
  double *A;
  long n, j0;
  int kk, nwg;
  double *VT, *tau, *betas;
  double *part; // [step c][workgroup][row q]: partial sums of step c, filled with SP_EMPTY before the launch
  double *heads; // [step c][row q]: the head column's entries of step c, likewise
  int *err;
};
// What one workgroup hands to the others is SELF-VALIDATING: the buffers hold one slot per (step, workgroup, row), the host fills
// them with a NaN pattern no computation produces (all ones) before the launch, and a consumer polls the slots it needs until
// none of them is empty.  No counter, no grid barrier, no wait for a store's acknowledgement: the latency of a step is one store
// on its way out plus one load round trip (the barrier version: store + acknowledgement, arrival, poll, loads -- 10.4 us per
// column, this one: see DESIGN 3.5).  The sums are formed in workgroup order once every slot is there: deterministic.
constexpr long long SP_EMPTY = -1LL;
__device__ __forceinline__ bool sp_have(double x) { return __double_as_longlong(x) != SP_EMPTY; }
constexpr int SP_NH = 4;                 // column parts per row: thread (q, h), 128 SP_NH threads per workgroup
constexpr int SP_CW = 128 / SP_NH;       // columns of one part of one 128-column chunk
constexpr int SP_THREADS = E2_B * SP_NH;
template <int NCH>
__global__ __launch_bounds__(SP_THREADS) void sb_panel_persist_kernel(SbPersistArgs g) {
  constexpr int W = 128 * NCH; // columns of this workgroup
  constexpr int NY = SP_CW * NCH;
  __shared__ __attribute__((aligned(16))) double xs[W], vs[W];
  __shared__ double hsum[SP_NH][E2_B], sred[E2_B];
  __shared__ int s_bad;
  const int t = threadIdx.x, q = t & (E2_B - 1), h = t >> 7;
  const long n = g.n, j0 = g.j0, r0 = j0 + E2_B;
  const long cb = r0 + (long)blockIdx.x * W; // first column of the workgroup (even; n is even: 16-byte aligned rows)
  const bool full = cb + W <= n;             // only the last workgroup has columns past n (they hold zeros throughout)
  if (t == 0) s_bad = 0;
  double y[NY];
  {
    const double *row = g.A + (j0 + q) * n + cb;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int l0 = SP_CW * (SP_NH * k + h);
      if (full) {
#pragma unroll
        for (int j = 0; j < SP_CW; j += 2) {
          const e2_v2 v2 = *reinterpret_cast<const e2_v2 *>(row + l0 + j);
          y[SP_CW * k + j] = v2[0];
          y[SP_CW * k + j + 1] = v2[1];
        }
      } else {
#pragma unroll
        for (int j = 0; j < SP_CW; ++j) y[SP_CW * k + j] = (cb + l0 + j < n) ? row[l0 + j] : 0.0;
      }
    }
  }
  const int nwg = g.nwg;
  const int wq = (nwg + SP_NH - 1) / SP_NH, wlo = min(h * wq, nwg), whi = min(wlo + wq, nwg);
  const int hl0 = (blockIdx.x == 0) ? 0 : -(1 << 20); // local index of head column r0 + c is c + hl0: negative outside workgroup 0
  __syncthreads();
  for (int c = 0; c <= g.kk; ++c) {
    const int prev = c - 1;
    const bool more = c < g.kk;
    if (c > 0) {
      // the sums of step c - 1 over all workgroups, in workgroup order (SP_NH ranges, then their sum in order): poll until every
      // slot this thread needs is there (wall-clock bound, error flag: the host then repeats the panel with the per-column launches)
      const double *pp = g.part + (size_t)prev * nwg * E2_B + q;
      const double *hp = g.heads + (size_t)prev * E2_B;
      double sacc = 0.0, alpha = 0.0, myhead = 0.0;
      long long t0 = 0;
      for (long it = 0;; ++it) {
        bool all = true;
        alpha = sp_ld(hp + prev);
        all = all && sp_have(alpha);
        if (q > prev) {
          myhead = sp_ld(hp + q);
          all = all && sp_have(myhead);
        }
        if (q >= prev) {
          double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
          int w = wlo;
          for (; w + 3 < whi; w += 4) {
            const double x0 = sp_ld(pp + (size_t)w * E2_B), x1 = sp_ld(pp + (size_t)(w + 1) * E2_B);
            const double x2 = sp_ld(pp + (size_t)(w + 2) * E2_B), x3 = sp_ld(pp + (size_t)(w + 3) * E2_B);
            all = all && sp_have(x0) && sp_have(x1) && sp_have(x2) && sp_have(x3);
            a0 += x0; a1 += x1; a2 += x2; a3 += x3;
          }
          for (; w < whi; ++w) {
            const double x0 = sp_ld(pp + (size_t)w * E2_B);
            all = all && sp_have(x0);
            a0 += x0;
          }
          sacc = (a0 + a1) + (a2 + a3);
        }
        if (all) break;
        if ((it & 15) == 15) {
          const long long now = (long long)wall_clock64();
          if (t0 == 0) t0 = now;
          if (now - t0 > BC_WAIT_TICKS || __hip_atomic_load(g.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            __hip_atomic_store(g.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_bad = 1;
            break;
          }
        }
        __builtin_amdgcn_s_sleep(1);
      }
      hsum[h][q] = sacc;
      __syncthreads();
      if (s_bad) return; // A untouched
      if (h == 0) {
        double sm = 0.0;
#pragma unroll
        for (int p = 0; p < SP_NH; ++p) sm += hsum[p][q];
        sred[q] = sm;
      }
      __syncthreads();
      double tau, beta, scale;
      e2_larfg(alpha, sred[prev], tau, beta, scale);
      if (t == 0 && blockIdx.x == 0) {
        g.tau[j0 + prev] = tau;
        g.betas[j0 + prev] = beta;
      }
      const double f = q > prev ? tau * (myhead + scale * sred[q]) : 0.0;
      // v of the workgroup's columns from x_{c-1} (xs: published in the previous step, zero at and left of the head and past
      // n), by the threads of ONE row: into LDS for the update and into row j0 + prev of VT
      if (q == (c & (E2_B - 1))) {
        const int hl = prev + hl0; // local index of the head column, negative outside workgroup 0
        double *vrow = g.VT + (j0 + prev) * n + cb;
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
          for (int j = 0; j < SP_CW; ++j) {
            const int lc = SP_CW * (SP_NH * k + h) + j;
            const double v = lc == hl ? 1.0 : scale * xs[lc];
            vs[lc] = v;
            if (lc >= hl && (full || cb + lc < n)) vrow[lc] = v;
          }
      }
      __syncthreads();
      if (q > prev) {
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
          for (int j = 0; j < SP_CW; j += 2) {
            const e2_v2 v2 = *reinterpret_cast<const e2_v2 *>(vs + SP_CW * (SP_NH * k + h) + j);
            y[SP_CW * k + j] -= f * v2[0];
            y[SP_CW * k + j + 1] -= f * v2[1];
          }
      }
    }
    if (!more) break;
    if (q == c) {
      const int hl = c + hl0;
#pragma unroll
      for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int j = 0; j < SP_CW; ++j) {
          const int lc = SP_CW * (SP_NH * k + h) + j;
          xs[lc] = (lc > hl) ? y[SP_CW * k + j] : 0.0; // columns past n hold zeros already
        }
    }
    if (blockIdx.x == 0 && q >= c && h == (c / SP_CW)) {
      // the head column r0 + c is local column c of workgroup 0 (chunk 0): the entries the next reflector needs from every row
      double hv = 0.0;
#pragma unroll
      for (int j = 0; j < SP_CW; ++j) hv = (j == (c & (SP_CW - 1))) ? y[j] : hv;
      sp_st(g.heads + (size_t)c * E2_B + q, hv);
    }
    __syncthreads();
    double pr = 0.0;
    if (q >= c) {
      double p0 = 0.0, p1 = 0.0;
#pragma unroll
      for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int j = 0; j < SP_CW; j += 2) {
          const e2_v2 x2 = *reinterpret_cast<const e2_v2 *>(xs + SP_CW * (SP_NH * k + h) + j);
          p0 += x2[0] * y[SP_CW * k + j];
          p1 += x2[1] * y[SP_CW * k + j + 1];
        }
      pr = p0 + p1;
    }
    hsum[h][q] = pr;
    __syncthreads();
    if (h == 0 && q >= c) {
      double sm = 0.0;
#pragma unroll
      for (int p = 0; p < SP_NH; ++p) sm += hsum[p][q];
      sp_st(g.part + ((size_t)c * nwg + blockIdx.x) * E2_B + q, sm);
    }
  }
  {
    double *row = g.A + (j0 + q) * n + cb;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int l0 = SP_CW * (SP_NH * k + h);
      if (full) {
#pragma unroll
        for (int j = 0; j < SP_CW; j += 2) {
          e2_v2 v2 = {y[SP_CW * k + j], y[SP_CW * k + j + 1]};
          *reinterpret_cast<e2_v2 *>(row + l0 + j) = v2;
        }
      } else {
#pragma unroll
        for (int j = 0; j < SP_CW; ++j)
          if (cb + l0 + j < n) row[l0 + j] = y[SP_CW * k + j];
      }
    }
  }
}

// P[blockIdx.x] (128 x 128) = X[:, chunk] Y[:, chunk]^T over the chunk's GR_CH columns; X, Y: 128 rows, K columns
__global__ __launch_bounds__(256) void sb_gram_kernel(const double *__restrict__ X, long ldx,
                                                      const double *__restrict__ Y, long ldy, long K,
                                                      double *__restrict__ P) {
  __shared__ double Xs[GR_KS][E2_B + 4], Ys[GR_KS][E2_B + 4];
  const int t = threadIdx.x, ty = t >> 4, tx = t & 15;
  const int lrow = t >> 1, lk = (t & 1) * 8;
  double acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.0;
  const long k0 = (long)blockIdx.x * GR_CH;
  for (int kc = 0; kc < GR_CH; kc += GR_KS) {
    const long kb = k0 + kc + lk;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool ok = kb + e < K;
      Xs[lk + e][lrow] = ok ? X[lrow * ldx + kb + e] : 0.0;
      Ys[lk + e][lrow] = ok ? Y[lrow * ldy + kb + e] : 0.0;
    }
    __syncthreads();
#pragma unroll 4
    for (int kk = 0; kk < GR_KS; ++kk) {
      double a[8], b[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = Xs[kk][ty * 8 + i];
#pragma unroll
      for (int j = 0; j < 8; ++j) b[j] = Ys[kk][tx * 8 + j];
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] += a[i] * b[j];
    }
    __syncthreads();
  }
  double *out = P + (size_t)blockIdx.x * E2_B * E2_B;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) out[(ty * 8 + i) * E2_B + tx * 8 + j] = acc[i][j];
}
__global__ void sb_gram_reduce_kernel(const double *__restrict__ P, int nwg, double *__restrict__ S) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  double s = 0.0;
  for (int w = 0; w < nwg; ++w) s += P[(size_t)w * E2_B * E2_B + idx];
  S[idx] = s;
}

// forward compact-WY factor of 128 reflectors from S = V V^T (strict upper triangle read) and tau, in LDS (the global-
// memory recurrence of bt_tfactor_kernel takes 1.2 ms per panel; this one ~40 us).  T row-major upper triangular, ld 128.
__global__ __launch_bounds__(256) void sb_tfactor_kernel(const double *__restrict__ S, const double *__restrict__ tau,
                                                         double *__restrict__ T) {
  extern __shared__ double e2sm[];
  double *Tt = e2sm, *scol = e2sm + E2_B * E2_B; // Tt[c][r] = T[r][c]
  const int t = threadIdx.x;
  for (int idx = t; idx < E2_B * E2_B; idx += 256) Tt[idx] = 0.0;
  __syncthreads();
  for (int i = 0; i < E2_B; ++i) {
    const double ti = tau[i];
    if (t < i) scol[t] = S[t * E2_B + i];
    __syncthreads();
    if (t < i) {
      double acc = 0.0;
      for (int c = t; c < i; ++c) acc += Tt[c * E2_B + t] * scol[c];
      Tt[i * E2_B + t] = -ti * acc;
    }
    if (t == i) Tt[i * E2_B + i] = ti;
    __syncthreads();
  }
  for (int idx = t; idx < E2_B * E2_B; idx += 256) {
    const int r = idx >> 7, c = idx & (E2_B - 1);
    T[idx] = Tt[c * E2_B + r];
  }
}

// The same factor by halving: T of [V1 V2] is [[T1, -T1 (V1^T V2) T2], [0, T2]].  Four 32 x 32 diagonal blocks by the serial
// recurrence (one wavefront each, side by side: 32 steps instead of 128), then two levels of two small products.  ~25 us
// instead of 370 us per panel.
constexpr int TF_LD = 65;
constexpr int TF_LDS_DOUBLES = 4 * 64 * TF_LD;
// Round 4: every block of S a phase multiplies with is staged in LDS first (Sb) -- the recurrences and the small products read
// S from global memory inside their inner loops before: 134 us per panel, not the 25 the arithmetic needs.
__global__ __launch_bounds__(256) void sb_tfactor_blocked_kernel(const double *__restrict__ S,
                                                                 const double *__restrict__ tau, double *__restrict__ T) {
  extern __shared__ double e2sm[];
  double *A0 = e2sm, *A1 = e2sm + 64 * TF_LD, *X = e2sm + 2 * 64 * TF_LD; // A0, A1: the 64 x 64 diagonal blocks of T
  double *Sb = e2sm + 3 * 64 * TF_LD;
  __shared__ double stau[E2_B];
  const int t = threadIdx.x;
  for (int idx = t; idx < 3 * 64 * TF_LD; idx += 256) e2sm[idx] = 0.0;
  if (t < E2_B) stau[t] = tau[t];
  // the four 32 x 32 diagonal blocks of S: Sb[(b * 32 + r) * 33 + c] = S[32 b + r][32 b + c]
  for (int idx = t; idx < 4 * 32 * 32; idx += 256) {
    const int b = idx >> 10, r = (idx >> 5) & 31, c = idx & 31;
    Sb[(b * 32 + r) * 33 + c] = S[(32 * b + r) * E2_B + 32 * b + c];
  }
  __syncthreads();
  {
    const int b = t >> 6, l = t & 63; // block b: rows / columns 32 b .. 32 b + 31 of T
    double *Ab = (b >> 1) ? A1 : A0;
    const int o = 32 * (b & 1), g0 = 32 * b;
    const double *Sd = Sb + b * 32 * 33;
    for (int i = 0; i < 32; ++i) {
      const double ti = stau[g0 + i];
      if (l < i) {
        double acc = 0.0;
        for (int c = l; c < i; ++c) acc += Ab[(o + l) * TF_LD + o + c] * Sd[c * 33 + i];
        Ab[(o + l) * TF_LD + o + i] = -ti * acc;
      }
      if (l == i) Ab[(o + i) * TF_LD + o + i] = ti;
      __syncthreads();
    }
  }
  { // level 1: inside each 64-block, T01 = -T00 (S01 T11) with 32 x 32 blocks; Sb[(p * 32 + r) * 33 + k] = S[64 p + r][64 p + 32 + k]
    for (int idx = t; idx < 2 * 32 * 32; idx += 256) {
      const int p = idx >> 10, r = (idx >> 5) & 31, k = idx & 31;
      Sb[(p * 32 + r) * 33 + k] = S[(64 * p + r) * E2_B + 64 * p + 32 + k];
    }
    __syncthreads();
    const int p = t >> 7;
    double *Ap = p ? A1 : A0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int idx = (t & 127) * 8 + e, r = idx >> 5, c = idx & 31;
      double acc = 0.0;
      for (int k = 0; k <= c; ++k) acc += Sb[(p * 32 + r) * 33 + k] * Ap[(32 + k) * TF_LD + 32 + c];
      X[(p * 32 + r) * TF_LD + c] = acc;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int idx = (t & 127) * 8 + e, r = idx >> 5, c = idx & 31;
      double acc = 0.0;
      for (int k = r; k < 32; ++k) acc += Ap[r * TF_LD + k] * X[(p * 32 + k) * TF_LD + c];
      Ap[r * TF_LD + 32 + c] = -acc;
    }
    __syncthreads();
  }
  { // level 2: T[0:64, 64:128] = -T00 (S[0:64, 64:128] T11); Sb[r * 65 + k] = S[r][64 + k]
    for (int idx = t; idx < 64 * 64; idx += 256) {
      const int r = idx >> 6, k = idx & 63;
      Sb[r * TF_LD + k] = S[r * E2_B + 64 + k];
    }
    __syncthreads();
    const int r = t >> 2, c0 = (t & 3) * 16;
    for (int c = c0; c < c0 + 16; ++c) {
      double acc = 0.0;
      for (int k = 0; k <= c; ++k) acc += Sb[r * TF_LD + k] * A1[k * TF_LD + c];
      X[r * TF_LD + c] = acc;
    }
    __syncthreads();
    for (int c = c0; c < c0 + 16; ++c) {
      double acc = 0.0;
      for (int k = r; k < 64; ++k) acc += A0[r * TF_LD + k] * X[k * TF_LD + c];
      T[r * E2_B + 64 + c] = -acc;
    }
  }
  for (int idx = t; idx < 64 * 64; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    T[r * E2_B + c] = A0[r * TF_LD + c];
    T[(64 + r) * E2_B + 64 + c] = A1[r * TF_LD + c];
    T[(64 + r) * E2_B + c] = 0.0;
  }
}

// lower triangle <- transpose of the strict upper triangle (32 x 32 tiles; the upper triangle is only read)
__global__ void e2_mirror_upper_kernel(double *__restrict__ A, long m, long ld) {
  __shared__ double tile[32][33];
  const int bx = blockIdx.x, by = blockIdx.y;
  if (bx < by) return;
  const int tx = threadIdx.x, ty = threadIdx.y; // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const long i = (long)by * 32 + r, j = (long)bx * 32 + tx;
    tile[r][tx] = (i < m && j < m) ? A[i * ld + j] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const long j = (long)bx * 32 + r, i = (long)by * 32 + tx;
    if (i < m && j < m && j > i) A[j * ld + i] = tile[tx][r];
  }
}

// band storage from the reduced matrix: Bd[j][t] = A[j][j + t], t <= 128 (row j of the upper triangle is contiguous);
// the entry at distance 128 of a row that carries a reflector is that reflector's beta
__global__ __launch_bounds__(256) void sb_extract_band_kernel(const double *__restrict__ A, long n,
                                                              const double *__restrict__ betas, double *__restrict__ Bd) {
  const long j = blockIdx.x; // n .. n + 127: slack columns, all zero
  const int t = threadIdx.x;
  double v = 0.0;
  if (j < n && t <= E2_B && j + t < n) {
    v = A[j * n + j + t];
    if (t == E2_B) {
      const double b = betas[j];
      if (b == b) v = b; // not NaN: row j was eliminated by a stage-1 reflector
    }
  }
  Bd[j * E2_LDB + t] = v;
}
__global__ void sb_band_de_kernel(const double *__restrict__ Bd, long n, double *__restrict__ d, double *__restrict__ e) {
  const long j = (long)blockIdx.x * 256 + threadIdx.x;
  if (j < n) {
    d[j] = Bd[j * E2_LDB];
    if (j + 1 < n) e[j] = Bd[j * E2_LDB + 1];
  }
}

// ---------------------------------------------------------------- stage 2: one time step of the bulge chase
struct BcArgs {
  double *Bd;
  long n, t, jlo;
  double *V2, *tau2; // V2[(k n + j) 128 + i], tau2[k n + j]
};
constexpr int BC_LD = E2_B + 1; // padded column stride of the 128 x 128 block in LDS: row sums and column sums both conflict-free
constexpr int BC_NH = 2;        // column parts per row: 128 BC_NH threads per task.  4 (512 threads) was measured in round 3: 0.90 s
                                // against 0.86 s at n = 20000 -- the task is bound by the latency of its global loads and of the store
                                // drain before the flag (6.5 + 6.4 us of 27 with stamps), not by its per-thread loops
constexpr int BC_THREADS = E2_B * BC_NH;
constexpr int BC_LDS_DOUBLES = E2_B * BC_LD + (4 + 2 * BC_NH) * E2_B + 32;
// task (j, k): the caller guarantees that it exists (j <= n - 3, j + 1 + 128 k < n) and that (j, k-1) and (j-1, k+1) are done
#define BC_STAMP(i)                                                  \
  do {                                                               \
    if (DBG && threadIdx.x == 0) dbg[i] = (long long)wall_clock64(); \
  } while (0)
// value of lane `l` (compile-time constant after unrolling) as a wave-uniform scalar: VALU readlane, no LDS traffic
__device__ __forceinline__ double bc_bcast(double x, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
  return __hiloint2double(hi, lo);
}
// block sum over the BC_THREADS threads of a chase workgroup; red: 2 BC_NH doubles
__device__ __forceinline__ double bc_bsum(double v, double *red) {
  v = eig_wsum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < 2 * BC_NH; ++w) s += red[w];
  return s;
}
// task (j, k): the caller guarantees that it exists (j <= n - 3, j + 1 + 128 k < n) and that (j, k-1) and (j-1, k+1) are done.
// Thread (a, h) owns row a of the 128 x 128 blocks and the CW = 128 / BC_NH columns cb = CW h ..; its CW entries of E and of the
// diagonal block's lower triangle stay in REGISTERS from the global load to the global store.  Row sums (E v_p, D v) run on
// those registers with the vector's entries broadcast by readlane; only the column sums (v^T E, the transposed half of D v) need
// the block in LDS: one store pass and one read pass per block.  The chase is a chain of 2 n dependent tasks, so the task's
// latency is the stage's time (BC_NH = 4, four column quarters on 512 threads, halves every per-thread loop and was NOT faster:
// see BC_NH).
template <bool SC1> __device__ __forceinline__ double bc_ld(const double *p) {
  if constexpr (SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool SC1> __device__ __forceinline__ void bc_st(double *p, double v) {
  if constexpr (SC1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
// SC1 (round 4): everything a task exchanges with its neighbours -- the two band blocks, v and tau -- moves with agent-scope
// relaxed atomics (sp_ld / sp_st: loads that bypass the XCD's L2, stores written through), so that the persistent kernel needs
// no acquire / release FENCE around a task: at agent scope those are an invalidate and a write-back of the XCD's whole L2, and
// the stamps of round 3 showed 6.4 us of a 19 us task in front of the release alone.
// EARLY (bc_persist1_kernel): three hand-overs instead of one at the end of the task --
//   vflag  : v and tau are out (6 us into the task): the next position of this sweep may start;
//   eflag  : the left block is stored (its row 0 is the last row of the diagonal block of task (j + 1, k - 1), its (0, 0) entry that
//            task's last left-block entry): the previous position may start its next sweep;
//   cbox_out: the updated (0, 0) entry of the diagonal block -- the only entry of this task's D phase the task (j + 1, k - 1) needs
//            (the corner of ITS diagonal block) -- travels in a self-validating slot (all-ones NaN = empty) that the reader
//            empties again; cbox_in is the slot of task (j - 1, k + 1) this task reads its own corner from (nullptr: no such task).
template <bool DBG, bool SC1, bool EARLY = false>
__device__ __forceinline__ void bc_task(double *__restrict__ B, long n, long j, long k, double *__restrict__ V2g,
                                        double *__restrict__ tau2g, double *e2sm, long long *dbg = nullptr, int *vflag = nullptr,
                                        int *eflag = nullptr, double *cbox_in = nullptr, double *cbox_out = nullptr,
                                        int *err = nullptr, int *vwait = nullptr) {
  constexpr int CW = E2_B / BC_NH;
  BC_STAMP(0);
  double *E = e2sm; // E[c][a] (column stride 129): the block whose column sums are being formed
  double *vp = e2sm + E2_B * BC_LD, *v = vp + E2_B, *zc = v + E2_B, *wv = zc + E2_B;
  double *ybuf = wv + E2_B, *zbuf = ybuf + BC_NH * E2_B, *red = zbuf + BC_NH * E2_B;
  const int t = threadIdx.x, a = t & (E2_B - 1), h = t >> 7, lane = t & 63, lc = lane & (CW - 1);
  const long r = j + 1 + k * E2_B;
  const int L = (int)((n - r < E2_B) ? (n - r) : E2_B);
  const int cb = h * CW;
  double er[CW], dr[CW];
  double xa, ya = 0.0, taup = 0.0, vph = 0.0;
  // vwait (EARLY with the left-block hand-over): neither block depends on the previous position of THIS sweep -- only v_p and
  // tau_p do -- so both blocks are requested first and the wait for that position's reflector comes behind them: the blocks'
  // memory latency leaves the chain of dependent tasks
  const bool late_v = EARLY && vwait != nullptr;
  if (k > 0) {
    // v_p and tau_p first: the counter retires in order, so whoever waits for E has them too and nothing later in the E phase
    // has to wait behind the diagonal block's loads
    if (!late_v) {
      vph = bc_ld<SC1>(V2g + ((size_t)(k - 1) * n + j) * E2_B + cb + lc); // v_p of this thread group's columns, entry c in lane c (c < CW)
      taup = bc_ld<SC1>(tau2g + (k - 1) * n + j);
    }
    const double *src = B + (r - E2_B + cb) * E2_LDB + E2_B + a - cb;
#pragma unroll
    for (int c = 0; c < CW; ++c) er[c] = bc_ld<SC1>(src + c * (E2_LDB - 1)); // rows past n: slots of the band storage that stay zero
  } else {
    // first task of a sweep: the "block" is column j alone (there is no previous reflector: v_p = 0, tau_p = 0); the code below
    // is the same, which keeps every global load of the task in front of the first barrier
#pragma unroll
    for (int c = 0; c < CW; ++c) er[c] = 0.0;
    if (h == 0) er[0] = bc_ld<SC1>(B + j * E2_LDB + 1 + a); // rows past n: zero slots
  }
  if (!late_v) {
#pragma unroll
    for (int c = 0; c < CW; ++c) E[(cb + c) * BC_LD + a] = er[c];
  }
  {
    // the diagonal block's loads go out once E has arrived and stay in flight behind the whole E phase; unconditional (a select
    // on the loaded value would be placed right behind the load and wait for it): columns past n lie in the zeroed slack of the
    // band storage, rows past n in slots that stay zero; above the diagonal (a < column) the address falls into the previous
    // column and the value is masked where it is used
    const double *srd = B + (r + cb) * E2_LDB + a - cb;
#pragma unroll
    for (int c = 0; c < CW; ++c) dr[c] = bc_ld<SC1>(srd + c * (E2_LDB - 1));
  }
  if (late_v) {
#pragma unroll
    for (int c = 0; c < CW; ++c) E[(cb + c) * BC_LD + a] = er[c];
    if (t == 0) red[2 * BC_NH + 2] = bc_wait(vwait, (int)(j + 1), err) ? 1.0 : 0.0;
    __syncthreads();
    if (red[2 * BC_NH + 2] == 0.0) return; // the error flag is up: every workgroup leaves at its next wait
    vph = bc_ld<SC1>(V2g + ((size_t)(k - 1) * n + j) * E2_B + cb + lc);
    taup = bc_ld<SC1>(tau2g + (k - 1) * n + j);
  }
  {
    double ys = 0.0;
#pragma unroll
    for (int c = 0; c < CW; ++c) ys += er[c] * bc_bcast(vph, c);
    ybuf[h * E2_B + a] = ys;
    if ((t & 64) == 0 && lane < CW) vp[cb + lane] = vph; // the first wavefront of every quarter publishes its part of v_p
    __syncthreads();
    BC_STAMP(1);
    double ysum = 0.0;
#pragma unroll
    for (int q = 0; q < BC_NH; ++q) ysum += ybuf[q * E2_B + a];
    ya = taup * ysum;
    xa = E[a] - ya * vp[0]; // first column of E (I - tau_p v_p v_p^T)
  }
  BC_STAMP(2);
  if (t == 0) red[2 * BC_NH] = xa;
  const double xnorm2 = bc_bsum((h == 0 && a >= 1 && a < L) ? xa * xa : 0.0, red);
  double tau, beta, scale;
  e2_larfg(red[2 * BC_NH], xnorm2, tau, beta, scale);
  const double va = (a == 0) ? 1.0 : ((a < L) ? scale * xa : 0.0);
  if (h == 0) {
    v[a] = va;
    bc_st<SC1>(V2g + ((size_t)k * n + j) * E2_B + a, va);
  }
  if (t == 0) bc_st<SC1>(tau2g + k * n + j, tau);
  // s = v . y (the right-hand reflector's share of z = v^T E (I - tau_p v_p v_p^T))
  if (EARLY) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // v and tau have left the chip's caches (and the diagonal block has landed)
  const double sdot = bc_bsum((h == 0) ? va * ya : 0.0, red); // includes the barrier that publishes v
  // EARLY: the next position of this sweep needs nothing else from this task -- its blocks are disjoint from this task's -- so it
  // may start now, 6 us into an 18 us task (bc_persist1_kernel)
  if (EARLY && t == 0) __hip_atomic_store(vflag, (int)(j + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  BC_STAMP(3);
  if (k > 0) {
    // z0_c = sum_q v_q E[c][q]: thread = column a, rows cb ..
    const double vrow = v[cb + lc];
    double zs = 0.0;
#pragma unroll
    for (int q = 0; q < CW; ++q) zs += E[a * BC_LD + cb + q] * bc_bcast(vrow, q);
    zbuf[h * E2_B + a] = zs;
    __syncthreads();
    if (h == 0) {
      double zsum = 0.0;
#pragma unroll
      for (int q = 0; q < BC_NH; ++q) zsum += zbuf[q * E2_B + a];
      zc[a] = zsum - sdot * vp[a];
    }
    __syncthreads();
    const double zch = zc[cb + lc];
    const double tva = tau * va;
    double *dst = B + (r - E2_B + cb) * E2_LDB + E2_B + a - cb;
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      double e = er[c] - ya * bc_bcast(vph, c) - tva * bc_bcast(zch, c);
      if (cb + c == 0) e = (a == 0) ? beta : 0.0;
      if (a < L) bc_st<SC1>(dst + c * (E2_LDB - 1), e);
    }
  } else if (h == 0 && a < L) {
    bc_st<SC1>(B + j * E2_LDB + 1 + a, (a == 0) ? beta : 0.0);
  }
  BC_STAMP(4);
  // EARLY: the corner of this diagonal block, D(127, 127), is the (0, 0) entry of the diagonal block of task (j - 1, k + 1), which may
  // still be running: thread (127, last column part) requests it from that task's slot now and uses it after the second product
  const bool corner_late = EARLY && cbox_in != nullptr;
  const bool corner_mine = corner_late && t == BC_THREADS - 1;
  double cslot = 0.0;
  if (corner_mine) cslot = bc_ld<true>(cbox_in);
  if (tau == 0.0) { // H = I (uniform over the block): nothing to do to D, but the hand-overs must still happen
    if (EARLY) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) {
        __hip_atomic_store(eflag, (int)(j + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cbox_out) bc_st<true>(cbox_out, dr[0]); // D(0, 0) unchanged
      }
      if (corner_mine) { // the slot is emptied by its reader, and its value has no other way into the band
        for (long it = 0; __double_as_longlong(cslot) == -1LL && it < (1L << 22); ++it) cslot = bc_ld<true>(cbox_in);
        if (__double_as_longlong(cslot) == -1LL) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bc_st<true>(cbox_in, __longlong_as_double(-1LL));
        bc_st<SC1>(B + (r + E2_B - 1) * E2_LDB, cslot);
      }
    }
    return;
  }
  // diagonal block (lower triangle dr, diagonal included): p = tau D v = tau (L v + strict(L)^T v),
  // w = p - (tau/2)(v.p) v, D -= v w^T + w v^T
  __syncthreads(); // the column sums above are done with E
  const double vh = v[cb + lc];
  double p1 = 0.0;
#pragma unroll
  for (int c = 0; c < CW; ++c) {
    E[(cb + c) * BC_LD + a] = (a > cb + c) ? dr[c] : 0.0; // strictly lower part for the transposed product
    const bool late = corner_mine && c == CW - 1;           // the corner's product is added below, once the slot has been read
    p1 += ((a >= cb + c && !late) ? dr[c] : 0.0) * bc_bcast(vh, c);
  }
  ybuf[h * E2_B + a] = p1;
  if (EARLY) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the left block's stores (issued a product ago) have left the chip
  __syncthreads();
  if (EARLY && t == 0) __hip_atomic_store(eflag, (int)(j + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  BC_STAMP(5);
  {
    double p2 = 0.0;
#pragma unroll
    for (int q = 0; q < CW; ++q) p2 += E[a * BC_LD + cb + q] * bc_bcast(vh, q); // column a, rows cb + q; vh = v of those rows
    zbuf[h * E2_B + a] = p2;
  }
  if (corner_mine) {
    long long t0 = 0;
    for (long it = 0; __double_as_longlong(cslot) == -1LL; ++it) { // normally there long ago: the slot was requested two products back
      cslot = bc_ld<true>(cbox_in);
      if ((it & 63) == 63) {
        const long long now = (long long)wall_clock64();
        if (t0 == 0) t0 = now;
        if (now - t0 > BC_WAIT_TICKS || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
          __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          cslot = 0.0;
        }
      }
    }
    bc_st<true>(cbox_in, __longlong_as_double(-1LL)); // empty again for the sweep after next
    dr[CW - 1] = cslot;
    // the corner's term is the LAST of this thread's first product: added here it is the sum the loop above forms when the corner
    // comes with the block (the hand-over variants return the same bits: test_bulge_chase_hand_over_variants_are_bit_identical)
    p1 += cslot * v[E2_B - 1]; // same source form as the loop's update: the same contraction into a fused multiply-add
    ybuf[h * E2_B + a] = p1;
  }
  __syncthreads();
  double psum = 0.0;
#pragma unroll
  for (int q = 0; q < BC_NH; ++q) psum += ybuf[q * E2_B + a] + zbuf[q * E2_B + a];
  const double pa = tau * psum;
  const double gamma = bc_bsum((h == 0) ? va * pa : 0.0, red);
  const double wa = pa - 0.5 * tau * gamma * va;
  if (h == 0) wv[a] = wa;
  __syncthreads();
  BC_STAMP(6);
  {
    const double wh = wv[cb + lc];
    double *dst = B + (r + cb) * E2_LDB + a - cb;
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      const double dnew = dr[c] - va * bc_bcast(wh, c) - wa * bc_bcast(vh, c);
      // D(0, 0) is the corner of task (j + 1, k - 1)'s diagonal block: when that task exists the value goes into its slot ONLY (the
      // reader stores its own update of it to the band; a store from here could land after that one)
      const bool handed = EARLY && c == 0 && t == 0 && cbox_out != nullptr;
      if (handed) bc_st<true>(cbox_out, dnew);
      else if (a >= cb + c && a < L) bc_st<SC1>(dst + c * (E2_LDB - 1), dnew);
    }
  }
  BC_STAMP(7);
}

__global__ __launch_bounds__(BC_THREADS) void bc_step_kernel(BcArgs g) {
  extern __shared__ double e2sm[];
  const long n = g.n, j = g.jlo + blockIdx.x, k = g.t - 2 * j;
  if (k < 0 || j > n - 3 || j + 1 + k * E2_B >= n) return;
  bc_task<false, false>(g.Bd, n, j, k, g.V2, g.tau2, e2sm);
}

// The same chase as ONE launch: workgroup w owns the chase positions k = 2 w and 2 w + 1 and walks the sweeps j = 0, 1, ...;
// (j, k) waits for (j, k-1) and (j-1, k+1) through per-position progress counters (agent-scope release / acquire).  A
// sweep can start only two steps behind its predecessor, so one workgroup alternating between two adjacent positions is
// never the bottleneck, and ceil(positions / 2) <= number of CUs keeps every workgroup resident (the host checks it; the
// per-step launches above remain as the fallback).  Every wait is bounded: a time-out raises *err and all workgroups leave.
struct BcPersistArgs {
  double *Bd;
  long n;
  double *V2, *tau2;
  int *prog, *err;
  long long *dbg; // GEMMA_HIP_EIGH_BC_DBG=1: wall-clock stamps (100 MHz) of workgroup 1's first tasks, 16 per task
};
template <bool DBG, bool SC1> // DBG: wall-clock stamps of workgroup 1's tasks (compiled out otherwise: a conditional store in the task
                              // makes the compiler wait for every load in flight at the join); SC1: see bc_task
__global__ __launch_bounds__(BC_THREADS) void bc_persist_kernel(BcPersistArgs g) {
  extern __shared__ double e2sm[];
  __shared__ int s_ok;
  const long n = g.n;
  const long k0 = 2 * (long)blockIdx.x;
  for (long j = 0; j <= n - 3; ++j) {
    if (j + 1 + k0 * E2_B >= n) break;
    for (int sidx = 0; sidx < 2; ++sidx) {
      const long k = k0 + sidx;
      if (j + 1 + k * E2_B >= n) break;
      if (threadIdx.x == 0) {
        bool ok = true;
        if (sidx == 0 && k > 0) ok = bc_wait(g.prog + (k - 1), (int)(j + 1), g.err);
        if (ok && sidx == 1 && j >= 1 && j + (k + 1) * E2_B < n) ok = bc_wait(g.prog + (k + 1), (int)j, g.err);
        s_ok = ok ? 1 : 0;
      }
      __syncthreads();
      if (!s_ok) return;
      if (!SC1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      long long *dbg = (DBG && g.dbg && blockIdx.x == 1 && sidx == 0 && j < 512) ? g.dbg + 16 * j : nullptr;
      if (DBG && dbg) bc_task<true, SC1>(g.Bd, n, j, k, g.V2, g.tau2, e2sm, dbg);
      else bc_task<false, SC1>(g.Bd, n, j, k, g.V2, g.tau2, e2sm);
      if (SC1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this thread's write-through stores have left the chip's caches
      else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(g.prog + k, (int)(j + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (DBG && dbg && threadIdx.x == 0) dbg[8] = (long long)wall_clock64();
    }
  }
}

// One chase position per workgroup, the reflector handed on EARLY (round 4).  With the hand-over at the end of a task the chain
// (j, k) -> (j, k + 1) -> (j + 1, k) costs two task times T per sweep whatever the number of workgroups.  But task (j, k + 1)
// needs only v and tau of (j, k) -- the blocks the two tasks touch are disjoint -- and those exist a = 6 us into the T = 18 us
// task: with start(j, k) >= start(j, k - 1) + a and start(j, k) >= end(j - 1, k + 1) the sweep period drops from 2 T to a + T.
// vprog[k] = sweeps whose reflector position k has published, prog[k] = sweeps it has finished.  Needs positions <= CUs
// (n <= 32 768 + 128 on 256 CUs); beyond that bc_persist_kernel (two positions per workgroup) stays.
// Round 4, second step: of the previous sweep's neighbour only the LEFT block has to be stored (eprog, published a third into its
// diagonal-block phase); the one entry of its diagonal block this task needs comes through a slot (cbox, see bc_task), so
// start(j, k) >= left_block_done(j - 1, k + 1) and the period is a + (that offset) instead of a + T.  GEMMA_HIP_EIGH_BC_PIPE=1
// keeps the whole-task hand-over (handshake = 0).
struct BcPersist1Args {
  double *Bd;
  long n;
  double *V2, *tau2;
  int *prog, *vprog, *eprog, *err;
  double *cbox; // [position][sweep parity]
  int handshake;
  long long *dbg;
};
template <bool DBG>
__global__ __launch_bounds__(BC_THREADS) void bc_persist1_kernel(BcPersist1Args g) {
  extern __shared__ double e2sm[];
  __shared__ int s_ok;
  const long n = g.n, k = blockIdx.x;
  for (long j = 0; j <= n - 3; ++j) {
    if (j + 1 + k * E2_B >= n) break;
    const bool has_right = j >= 1 && j + (k + 1) * E2_B < n; // task (j - 1, k + 1) exists
    if (threadIdx.x == 0) {
      bool ok = true;
      if (k > 0 && !g.handshake) ok = bc_wait(g.vprog + (k - 1), (int)(j + 1), g.err); // handshake: inside the task, behind its loads
      if (ok && has_right) ok = bc_wait((g.handshake ? g.eprog : g.prog) + (k + 1), (int)j, g.err);
      s_ok = ok ? 1 : 0;
    }
    __syncthreads();
    if (!s_ok) return;
    long long *dbg = (DBG && g.dbg && blockIdx.x == 2 && j < 512) ? g.dbg + 16 * j : nullptr;
    double *cin = (g.handshake && has_right) ? g.cbox + 2 * (k + 1) + ((j - 1) & 1) : nullptr;
    double *cout = (g.handshake && k >= 1 && j + 1 <= n - 3) ? g.cbox + 2 * k + (j & 1) : nullptr;
    int *vwait = (g.handshake && k > 0) ? g.vprog + (k - 1) : nullptr;
    if (DBG && dbg) bc_task<true, true, true>(g.Bd, n, j, k, g.V2, g.tau2, e2sm, dbg, g.vprog + k, g.eprog + k, cin, cout, g.err, vwait);
    else bc_task<false, true, true>(g.Bd, n, j, k, g.V2, g.tau2, e2sm, nullptr, g.vprog + k, g.eprog + k, cin, cout, g.err, vwait);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this thread's write-through stores have left the chip's caches
    __syncthreads();
    if (threadIdx.x == 0) {
      // a task that returned early (tau = 0: H = I) has not published through the EARLY path's flag store?  it has: the flag
      // store sits before that return
      __hip_atomic_store(g.prog + k, (int)(j + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (DBG && dbg && threadIdx.x == 0) dbg[8] = (long long)wall_clock64();
  }
}

// ---------------------------------------------------------------- stage-2 back-transformation
// group (Jb, k): sweeps J0 = 32 Jb .. J0 + 31 at chase step k; first window column c0 = J0 + 1 + 128 k.
// pack: V as 160 x 34 (V[wc][jj] = v_{J0+jj,k}[wc - jj]) then T (32 x 34, forward compact WY, row-major upper).
struct Q2PackArgs {
  const double *V2, *tau2;
  long n;
  const long *goff; // first group index of sweep block Jb
  double *pack;
};
__global__ __launch_bounds__(256) void q2_pack_kernel(Q2PackArgs g) {
  __shared__ double Vd[E2_WIN][E2_NB + 1];
  __shared__ double Ss[E2_NB][E2_NB + 1], Tt[E2_NB][E2_NB + 1], stau[E2_NB];
  const int t = threadIdx.x;
  const long n = g.n, Jb = blockIdx.x, k = blockIdx.y, J0 = Jb * E2_NB;
  if (J0 > n - 3 || J0 + 1 + k * E2_B >= n) return;
  for (int idx = t; idx < E2_WIN * (E2_NB + 1); idx += 256) (&Vd[0][0])[idx] = 0.0;
  __syncthreads();
  for (int idx = t; idx < E2_NB * E2_B; idx += 256) {
    const int jj = idx >> 7, i = idx & (E2_B - 1);
    const long j = J0 + jj;
    const bool ex = (j <= n - 3) && (j + 1 + k * E2_B < n);
    if (ex) Vd[jj + i][jj] = g.V2[((size_t)k * n + j) * E2_B + i];
    if (i == 0) stau[jj] = ex ? g.tau2[k * n + j] : 0.0;
  }
  __syncthreads();
  for (int idx = t; idx < E2_NB * E2_NB; idx += 256) {
    const int p = idx >> 5, q = idx & (E2_NB - 1);
    double s = 0.0;
    if (p < q)
      for (int wc = q; wc < p + E2_B && wc < E2_WIN; ++wc) s += Vd[wc][p] * Vd[wc][q];
    Ss[p][q] = s;
    Tt[p][q] = 0.0;
  }
  __syncthreads();
  for (int i = 0; i < E2_NB; ++i) {
    if (t < i) {
      double acc = 0.0;
      for (int c = t; c < i; ++c) acc += Tt[t][c] * Ss[c][i];
      Tt[t][i] = -stau[i] * acc;
    }
    if (t == i) Tt[i][i] = stau[i];
    __syncthreads();
  }
  double *dst = g.pack + (size_t)(g.goff[Jb] + k) * E2_PACK;
  for (int idx = t; idx < E2_WIN * E2_VLD; idx += 256) {
    const int wc = idx / E2_VLD, jj = idx % E2_VLD;
    dst[idx] = (jj < E2_NB) ? Vd[wc][jj] : 0.0;
  }
  for (int idx = t; idx < E2_NB * E2_VLD; idx += 256) {
    const int p = idx / E2_VLD, q = idx % E2_VLD;
    // stored NEGATED (round 6): the apply kernel then forms -T W^T in its second product and adds V (-W2^T) in its third -- the same bits as
    // subtracting V W2^T, without a sign flip of every V operand in front of the third product's matrix instructions
    dst[E2_WIN * E2_VLD + idx] = (q < E2_NB) ? -Tt[p][q] : 0.0;
  }
}


struct Q2ApplyArgs {
  double *ZT;
  long nrows; // rows of Z^T (eigenvectors) this launch transforms: n, or one rank's slice (eigenvectors are independent)
  long n;
  const double *pack;
  const long *goff;
  int nJ, kmaxall;
  // dynamic schedule (sync != nullptr): tasks (segment of chase steps k, block of 64 rows), segment-major; sync[0] = next task,
  // sync[1] = error flag, sync[2 + rb] = segments done for row block rb; kseg[0 .. nseg] = first chase step of every segment
  int *sync;
  const int *kseg;
  int nseg, nrb;
};
// 4 wavefronts x 16 rows of Z^T per workgroup.  Lane l = (li = l & 15, lk = l >> 4) holds
// x[4 ct + r] = Z^T[row0 + li][c0 + 16 ct + lk + 4 r]: tile ct of X^T in the MFMA accumulator layout, and at the same
// time the operand fragment of k-step 4 ct + r.
constexpr int Q2_MAXJ = 2048; // sweep blocks the kernel can index (n <= 65 536)
constexpr int Q2_MAXSEG = 64;  // segments of chase steps of the dynamic schedule
// Balance (round 3): a row block is one wavefront per SIMD for the whole launch, and n / 64 row blocks on 2 x 256 workgroup
// slots leave the CUs that got two of them with twice the work (n = 20 000: 313 blocks, 0.57 s = the time of the 57 CUs with
// two; one block alone on a CU runs 1.5 x faster than each of two, so the rest idle for a third of the launch).  The window of
// Z^T is loaded at the first group of a chase step k and stored at its last one, so the steps of a row block can be handed from
// one workgroup to another between any two k: the launch is 2 x CUs persistent workgroups that draw tasks (segment of k, row
// block) from a counter in segment-major order; a task waits for the previous segment of its row block (claimed earlier, hence
// running: no deadlock; bounded wait and an error flag as in the chase) with agent-scope release / acquire around the hand-over.
__global__ __launch_bounds__(256, 2) void q2_apply_kernel(Q2ApplyArgs g) {
  __shared__ double Ls[E2_PACK];
  __shared__ int sgoff[Q2_MAXJ]; // first group of every sweep block: looked up through LDS so that the lookup never waits on
                                 // the vector-memory counter behind the window loads in flight
  __shared__ int s_task;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lk = lane >> 4;
  const long n = g.n;
  for (int i = t; i < g.nJ; i += 256) sgoff[i] = (int)g.goff[i];
  __syncthreads();
  constexpr int NT = E2_WIN / 16; // 10 window tiles
  constexpr int PK_N = (E2_PACK / 2 + 255) / 256; // 16-byte pieces of a pack per thread
  double x[4 * NT];
  e2_v2 pk[PK_N];
  const bool dyn = g.sync != nullptr;
 for (;;) {
  int rb = blockIdx.x, kfirst = 0, kend = g.kmaxall, segi = 0;
  if (dyn) {
    if (t == 0) {
      int task = atomicAdd(g.sync, 1);
      if (task < g.nseg * g.nrb) {
        const int sg = task / g.nrb;
        if (sg > 0 && !bc_wait(g.sync + 2 + (task - sg * g.nrb), sg, g.sync + 1, Q2_WAIT_TICKS)) task = 0x7fffffff;
      }
      s_task = task;
    }
    __syncthreads();
    const int task = s_task;
    __syncthreads();
    if (task >= g.nseg * g.nrb) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    segi = task / g.nrb;
    rb = task - segi * g.nrb;
    kfirst = g.kseg[segi];
    kend = g.kseg[segi + 1];
  }
  const long row = (long)rb * 64 + wave * 16 + li;
  const bool rok = row < g.nrows;
  double *zrow = g.ZT + (rok ? row : 0) * n;
  {
    const long lim0 = n - 2 - (long)kfirst * E2_B; // first group of this task: step kfirst, its last sweep block
    const long Jb0 = (lim0 / E2_NB < g.nJ - 1) ? lim0 / E2_NB : g.nJ - 1;
    const e2_v2 *src = reinterpret_cast<const e2_v2 *>(g.pack + (size_t)(sgoff[Jb0 < 0 ? 0 : Jb0] + kfirst) * E2_PACK);
#pragma unroll
    for (int q = 0; q < PK_N; ++q)
      if (t + 256 * q < E2_PACK / 2) pk[q] = src[t + 256 * q];
  }
  for (int k = kfirst; k < kend; ++k) {
    const long lim = n - 2 - (long)k * E2_B; // sweep blocks with J0 <= lim have a task at step k
    if (lim < 0) break;
    long Jbmax = lim / E2_NB;
    if (Jbmax > g.nJ - 1) Jbmax = g.nJ - 1;
    for (long Jb = Jbmax; Jb >= 0; --Jb) {
      const long c0 = Jb * E2_NB + 1 + (long)k * E2_B;
      // loads are unconditional (column clamped into the row; rows past n read row 0): a select on the loaded value would make
      // the compiler wait for it on the spot.  What such a lane holds is never stored, and the rows of V that multiply window
      // columns past n are zero.
      const bool inside = c0 + E2_WIN <= n; // the whole window lies inside the row: one base address, immediate offsets
      const double *zw = zrow + c0 + lk;
      if (Jb == Jbmax) {
        if (inside) {
#pragma unroll
          for (int q = 0; q < 4 * NT; ++q) x[q] = zw[4 * q];
        } else {
#pragma unroll
          for (int q = 0; q < 4 * NT; ++q) x[q] = zrow[c0 + lk + 4 * q < n ? c0 + lk + 4 * q : n - 1];
        }
      } else {
        // the window moved down by 32 columns: tiles 8, 9 leave (their columns are c0 + 160 .. c0 + 191 of the new c0)
#pragma unroll
        for (int ct = NT - 2; ct < NT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const long col = c0 + E2_NB + 16 * ct + lk + 4 * r;
            if (rok && col < n) zrow[col] = x[4 * ct + r];
          }
#pragma unroll
        for (int q = 4 * NT - 1; q >= 8; --q) x[q] = x[q - 8];
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = zw[4 * q]; // columns c0 .. c0 + 31 < n always (the group has a task)
      }
      __syncthreads(); // everyone is done with the previous group's V and T
#pragma unroll
      for (int q = 0; q < PK_N; ++q)
        if (t + 256 * q < E2_PACK / 2) reinterpret_cast<e2_v2 *>(Ls)[t + 256 * q] = pk[q];
      __syncthreads();
      const double *Vd = Ls, *Tm = Ls + E2_WIN * E2_VLD;
      // W^T (32 x 16) = V^T X^T
      e2_v4 wt0 = {0.0, 0.0, 0.0, 0.0}, wt1 = {0.0, 0.0, 0.0, 0.0};
      // k-steps 8 .. 39 first: the window tiles 0 and 1 (k-steps 0 .. 7) were requested from global memory at the top of this
      // group and get the MFMAs of the thirty-two older k-steps to arrive
#pragma unroll
      for (int kq = 0; kq < E2_WIN / 4; ++kq) {
        const int ks = (kq + 8) % (E2_WIN / 4);
        // V[wc][jj] = 0 outside 0 <= wc - jj < 128: sweeps 0..15 end at window column 142, sweeps 16..31 start at 16
        if (4 * ks < E2_B + 16) {
          const double a0 = Vd[(4 * ks + lk) * E2_VLD + li];
          wt0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, x[ks], wt0, 0, 0, 0);
        }
        if (4 * ks + 3 >= 16) {
          const double a1 = Vd[(4 * ks + lk) * E2_VLD + 16 + li];
          wt1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, x[ks], wt1, 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // W2^T = T W^T
      e2_v4 w20 = {0.0, 0.0, 0.0, 0.0}, w21 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < E2_NB / 4; ++ks) {
        const double b = (ks < 4) ? wt0[ks & 3] : wt1[ks & 3];
        const double a0 = Tm[li * E2_VLD + 4 * ks + lk], a1 = Tm[(16 + li) * E2_VLD + 4 * ks + lk];
        w20 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, w20, 0, 0, 0);
        w21 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, w21, 0, 0, 0);
      }
      {
        // the next group's pack (the next sweep block of this k, or the first one of k + 1) goes into registers now (after the first
        // two products: their accumulators' registers are free) and is in flight behind the 72 MFMAs of the third
        long nJb = Jb - 1, nk = k;
        if (nJb < 0) {
          nk = k + 1;
          const long nlim = n - 2 - nk * E2_B;
          nJb = (nk < kend && nlim >= 0) ? (nlim / E2_NB < g.nJ - 1 ? nlim / E2_NB : g.nJ - 1) : -1;
        }
        if (nJb >= 0) {
          const e2_v2 *src = reinterpret_cast<const e2_v2 *>(g.pack + (size_t)(sgoff[nJb] + nk) * E2_PACK);
#pragma unroll
          for (int q = 0; q < PK_N; ++q)
            if (t + 256 * q < E2_PACK / 2) pk[q] = src[t + 256 * q];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // X^T tile ct += V[16 ct .., :] (-W2^T)   (the pack holds -T: w20 / w21 are -W2^T).  Round 6: TWO window tiles at a time, their
      // matrix instructions alternating -- a tile's eight instructions are one dependent chain on its accumulator, and back to back each
      // waited for the one before it; the other tile's fill those slots (every chain keeps its own order: the same bits)
#pragma unroll
      for (int ct = 0; ct < NT; ct += 2) {
        e2_v4 acc0 = {x[4 * ct], x[4 * ct + 1], x[4 * ct + 2], x[4 * ct + 3]};
        e2_v4 acc1 = {x[4 * ct + 4], x[4 * ct + 5], x[4 * ct + 6], x[4 * ct + 7]};
#pragma unroll
        for (int ks = 0; ks < E2_NB / 4; ++ks) {
          const double b = (ks < 4) ? w20[ks & 3] : w21[ks & 3];
          if (!(ct == 0 && ks >= 4)) { // zero corners of the parallelogram: tile 0 has no sweeps 16..31, tile NT - 1 no sweeps 0..15
            const double a0 = Vd[(16 * ct + li) * E2_VLD + 4 * ks + lk];
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, acc0, 0, 0, 0);
          }
          if (!(ct + 1 == NT - 1 && ks < 4)) {
            const double a1 = Vd[(16 * (ct + 1) + li) * E2_VLD + 4 * ks + lk];
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, acc1, 0, 0, 0);
          }
        }
        x[4 * ct] = acc0[0];
        x[4 * ct + 1] = acc0[1];
        x[4 * ct + 2] = acc0[2];
        x[4 * ct + 3] = acc0[3];
        x[4 * ct + 4] = acc1[0];
        x[4 * ct + 5] = acc1[1];
        x[4 * ct + 6] = acc1[2];
        x[4 * ct + 7] = acc1[3];
        __builtin_amdgcn_sched_barrier(0); // keep the operand reads of later tiles from being hoisted (register pressure)
      }
      if (Jb == 0) {
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const long col = c0 + 16 * ct + lk + 4 * r;
            if (rok && col < n) zrow[col] = x[4 * ct + r];
          }
      }
    }
  }
  if (!dyn) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (t == 0) __hip_atomic_store(g.sync + 2 + rb, segi + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
 }
}

// ---------------------------------------------------------------- host orchestration
struct Eig2Ws {
  double *Bd = nullptr, *Bd0 = nullptr, *part = nullptr, *heads = nullptr, *betas = nullptr, *gramP = nullptr, *YT = nullptr;
  double *V2 = nullptr, *tau2 = nullptr, *pack = nullptr;
  long *goff = nullptr;
  double *P256 = nullptr, *Tpair = nullptr; // Q1 applied two panels at a time: three n x 256 buffers, one 256 x 256 factor
  int *prog = nullptr; // progress counters of the persistent bulge chase (+ the error flag)
  int *pbar = nullptr; // error flag of the persistent panel kernel ([1])
  double *ZS = nullptr; // K slices of V^T A22: E2_MAXSLICE x 128 x n
  double *ppart = nullptr, *pheads = nullptr; // its self-validating slots: [step][workgroup][row], [step][row]
  int *q2sync = nullptr; // q2_apply_kernel's task counter, error flag, per-row-block progress; then the segment table
  double *Qg = nullptr;  // eig2_apply_q1 with groups of panels: Gram of the group's reflectors, its compact-WY factor, scratch, K slices
  long q1_kpmax = 0;     // widest block reflector the buffers above (and P256) were sized for
  long ngroups = 0, kmaxall = 0, nJ = 0;
};

// C0 + C1 = alpha op(A) op(B) with the K range cut in two: skinny products (M or N = 128, K = thousands) have only
// n / 128 output tiles -- fewer than the chip has CUs -- so the two halves run side by side on the caller's stream and on the
// GEMM side stream; the consumer adds the halves in its own product (beta = 1), no reduction kernel.  Without a side
// stream C0 gets everything and C1 is cleared.
static inline int eig2_dgemm_split2(char ta, char tb, long M, long N, long K, double alpha, const double *A, long lda,
                                    const double *B, long ldb, double *C0, double *C1, long ldc, hipStream_t s,
                                    std::string &msg) {
  const long Kh = (K / 2) / GEMM_BK * GEMM_BK;
  if (!g_gemm_aux.stream || Kh < 4 * GEMM_BK) {
    EIG_HIP(launch_dgemm(ta, tb, M, N, K, alpha, A, lda, B, ldb, 0.0, C0, ldc, false, false, s));
    EIG_HIP(hipMemset2DAsync(C1, ldc * 8, 0, N * 8, M, s));
    return 0;
  }
  const bool tA = (ta == 'T'), tB = (tb == 'T');
  const double *A1 = tA ? A + Kh * lda : A + Kh, *B1 = tB ? B + Kh : B + Kh * ldb;
  hipStream_t es = g_gemm_aux.stream;
  EIG_HIP(hipEventRecord(g_gemm_aux.ready, s));
  EIG_HIP(hipStreamWaitEvent(es, g_gemm_aux.ready, 0));
  EIG_HIP(launch_dgemm(ta, tb, M, N, K - Kh, alpha, A1, lda, B1, ldb, 0.0, C1, ldc, false, false, es));
  EIG_HIP(hipEventRecord(g_gemm_aux.done, es));
  EIG_HIP(launch_dgemm(ta, tb, M, N, Kh, alpha, A, lda, B, ldb, 0.0, C0, ldc, false, false, s));
  EIG_HIP(hipStreamWaitEvent(s, g_gemm_aux.done, 0));
  return 0;
}

// K SLICES IN ONE LAUNCH (round 4): at n = 20 000 a half of the skinny product V^T A22 is 60-150 workgroups with a K loop of
// thousands of steps each -- the two halves together left half of the chip idle (1.74 ms for 0.76 ms of matrix work at
// m = 14 752).  The GEMM takes the slice from blockIdx.y (dgemm_mfma.hip.h: launch_dgemm_ksliced), so that ~2 workgroups per
// CU exist whatever m is; the slices land in w2.ZS and e2_sumk_kernel adds them in slice order (fixed: deterministic).  A first
// version put four K ranges on four streams: two of them shared a hardware queue (the runtime hands out four per process) and
// ran one after the other.
// D[r][c] = sum_k P[k * stride + r * ld + c] (rows x cols), k ascending
__global__ void e2_sumk_kernel(double *__restrict__ D, const double *__restrict__ P, int ns, long stride, long cols, long ld) {
  const long c = ((long)blockIdx.x * 256 + threadIdx.x) * 2, r = blockIdx.y;
  if (c >= cols) return;
  const long o = r * ld + c;
  if (c + 1 < cols) {
    e2_v2 acc = *reinterpret_cast<const e2_v2 *>(P + o);
    for (int k = 1; k < ns; ++k) {
      const e2_v2 x = *reinterpret_cast<const e2_v2 *>(P + (long)k * stride + o);
      acc[0] += x[0];
      acc[1] += x[1];
    }
    *reinterpret_cast<e2_v2 *>(D + o) = acc;
  } else {
    double acc = P[o];
    for (int k = 1; k < ns; ++k) acc += P[(long)k * stride + o];
    D[o] = acc;
  }
}
constexpr int E2_MAXSLICE = 8;

static inline int eig2_gram(const double *X, const double *Y, long ld, long K, double *P, double *S, hipStream_t s,
                            std::string &msg) {
  const int nwg = (int)((K + GR_CH - 1) / GR_CH);
  hipLaunchKernelGGL(sb_gram_kernel, dim3(nwg), dim3(256), 0, s, X, ld, Y, ld, K, P);
  hipLaunchKernelGGL(sb_gram_reduce_kernel, dim3(E2_B * E2_B / 256), dim3(256), 0, s, P, nwg, S);
  EIG_HIP(hipGetLastError());
  return 0;
}

// A (n x n, both triangles) -> band in w2.Bd; reflectors in ws.VT (row j0 + c: reflector c of the panel at j0, head at
// column j0 + 128 + c), their compact-WY factors in ws.Tall.  A is destroyed.
// the look-ahead's second stream and its two events (eig2_sy2sb); destroyed by eigh_tu_shutdown
struct Eig2LookAhead {
  hipStream_t stream = nullptr;
  hipEvent_t ready = nullptr, done = nullptr;
};
static Eig2LookAhead g_eig2_la;
static inline void eig2_lookahead_destroy() {
  if (g_eig2_la.stream) (void)hipStreamDestroy(g_eig2_la.stream);
  if (g_eig2_la.ready) (void)hipEventDestroy(g_eig2_la.ready);
  if (g_eig2_la.done) (void)hipEventDestroy(g_eig2_la.done);
  g_eig2_la = Eig2LookAhead();
}
static inline int eig2_sy2sb(double *A, long n, EigWs &ws, Eig2Ws &w2, hipStream_t s, std::string &msg) {
  const long nb2 = (long)E2_B * E2_B;
  EIG_HIP(hipMemsetAsync(ws.VT, 0, (size_t)n * n * 8, s));
  EIG_HIP(hipMemsetAsync(ws.tau, 0, (size_t)n * 8, s));
  EIG_HIP(hipMemsetAsync(w2.betas, 0xFF, (size_t)n * 8, s));
  EIG_HIP(hipMemsetAsync(ws.Tall, 0, (size_t)((n + E2_B - 1) / E2_B) * nb2 * 8, s));
  static bool attr = false;
  if (!attr) {
    EIG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sb_tfactor_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (E2_B * E2_B + E2_B) * 8));
    EIG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sb_tfactor_blocked_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, TF_LDS_DOUBLES * 8));
    attr = true;
  }
  const char *etf = getenv("GEMMA_HIP_EIGH_TFACTOR"); // "serial": the 128-step recurrence
  const bool tf_blocked = !(etf && etf[0] == 's');
  // GEMMA_HIP_EIGH_PANEL=launch: one launch per panel column (rounds 2-3); default: one persistent launch per panel
  // (sb_panel_persist_kernel) wherever all of its workgroups fit on the chip at once
  const char *epn = getenv("GEMMA_HIP_EIGH_PANEL");
  bool persist = !(epn && epn[0] == 'l');
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu <= 0) ncu = 1;
  }
  if (persist) EIG_HIP(hipMemsetAsync(w2.pbar, 0, 4 * sizeof(int), s));
  const char *emi = getenv("GEMMA_HIP_EIGH_MIRROR");
  const bool fused_mirror = !(emi && emi[0] == 'p');
  // the Householder QR of one panel (columns j0 .. j0 + 127 below the band) on stream ps; nonzero: a HIP error
  auto factor_panel = [&](long j0, hipStream_t ps) -> int {
    const long r0 = j0 + E2_B, m = n - r0;
    const int kk = (int)std::min<long>(E2_B, m - 1);
    bool done = false;
    if (persist) {
      // 128 columns per workgroup, every workgroup resident: panels up to 128 x CUs columns.  (256 columns per workgroup -- the
      // NCH = 2 instantiation -- spills, and measured at n = 50 000 it is slower than the per-column launches it would replace:
      // dense -> band 4.82 s against 4.72 s; wider panels therefore keep the launches.)
      const int nwgp = (int)((m + 127) / 128);
      if (nwgp <= ncu) {
        EIG_HIP(hipMemsetAsync(w2.ppart, 0xFF, (size_t)kk * nwgp * E2_B * 8, ps));
        EIG_HIP(hipMemsetAsync(w2.pheads, 0xFF, (size_t)kk * E2_B * 8, ps));
        SbPersistArgs pp{A, n, j0, kk, nwgp, ws.VT, ws.tau, w2.betas, w2.ppart, w2.pheads, w2.pbar + 1};
        hipLaunchKernelGGL(sb_panel_persist_kernel<1>, dim3(nwgp), dim3(SP_THREADS), 0, ps, pp);
        EIG_HIP(hipGetLastError());
        int err = 0;
        EIG_HIP(hipMemcpyAsync(&err, w2.pbar + 1, sizeof(int), hipMemcpyDeviceToHost, ps));
        EIG_HIP(hipStreamSynchronize(ps));
        if (!err) {
          done = true;
        } else {
          persist = false; // a workgroup waited too long (the chip was shared): A is untouched, per-column launches from here on
        }
      }
    }
    if (!done) {
      const int nwg = (int)((m + SB_COLS - 1) / SB_COLS);
      SbPanelArgs pa{A, n, j0, 0, kk, nwg, ws.VT, w2.part, w2.heads, ws.tau, w2.betas};
      for (int c = 0; c <= kk; ++c) {
        pa.c = c;
        hipLaunchKernelGGL(sb_panel_kernel, dim3(nwg), dim3(256), 0, ps, pa);
      }
      EIG_HIP(hipGetLastError());
    }
    return 0;
  };
  // LOOK-AHEAD (round 6).  Where the panel is wider than 128 x CUs columns its QR is a chain of 129 launches (2.5 ms) during which most
  // of the chip idles, and the trailing update that precedes it is the stage's largest product.  There the update is issued in two
  // pieces -- the first tile row (with its mirror: the next panel's columns), then the rest -- and the NEXT panel is factored on a
  // second, high-priority stream beside the second piece.  Same tiles, same K order: the reduced matrix is bit-identical either way.
  // Panels that fit the one-launch kernel are NOT looked ahead: its workgroups poll each other, and on a chip they share with the
  // update the chain slows down by more than the product hides (measured, profiles/r06_eigh_lookahead.txt: looking ahead at every
  // panel n = 20 000 loses 0.03 s and n = 33 000 0.06 s; n = 50 000 gains 0.13 s, and 0.19 s with the launches only).
  // GEMMA_HIP_EIGH_LOOKAHEAD=0: never; =1: at every panel (the bit-identity tests).
  hipStream_t &la_stream = g_eig2_la.stream;
  hipEvent_t &la_ready = g_eig2_la.ready, &la_done = g_eig2_la.done;
  const char *ela = getenv("GEMMA_HIP_EIGH_LOOKAHEAD");
  const int la_mode = (ela && ela[0] == '0') ? 0 : (ela && ela[0] == '1') ? 1 : 2; // 2: panels on the launch path only
  bool lookahead = la_mode > 0 && fused_mirror;
  auto la_streams = [&]() -> bool { // created at the first use
    if (la_stream) return true;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&la_stream, hipStreamNonBlocking, hi) != hipSuccess ||
        hipEventCreateWithFlags(&la_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&la_done, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      eig2_lookahead_destroy();
      lookahead = false;
      return false;
    }
    return true;
  };
  bool have_panel = false; // the panel of this j0 was factored beside the previous update
  for (long j0 = 0;; j0 += E2_B) {
    const long r0 = j0 + E2_B, m = n - r0;
    if (m < 2) break;
    if (!have_panel) {
      const int rcp = factor_panel(j0, s);
      if (rcp) return rcp;
    }
    have_panel = false;
    const long p = j0 / E2_B;
    const double *Vr = ws.VT + j0 * n + r0;
    double *A22 = A + r0 * n + r0;
    double *T = ws.Tall + p * nb2;
    int rc = eig2_gram(Vr, Vr, n, m, w2.gramP, ws.S, s, msg);
    if (rc) return rc;
    if (tf_blocked)
      hipLaunchKernelGGL(sb_tfactor_blocked_kernel, dim3(1), dim3(256), TF_LDS_DOUBLES * 8, s, ws.S, ws.tau + j0, T);
    else
      hipLaunchKernelGGL(sb_tfactor_kernel, dim3(1), dim3(256), (E2_B * E2_B + E2_B) * 8, s, ws.S, ws.tau + j0, T);
    EIG_HIP(hipGetLastError());
    // Z1 = V^T A22 (128 x m), Y^T = T^T Z1
    double *SA = w2.YT, *SB = w2.YT + (size_t)2 * E2_B * n; // stacked operands [V; W] and [W; V], 256 x m each (ld n)
    double *Wr = SA + (size_t)E2_B * n;
    {
      // Z1 = V^T A22 in K slices (about two workgroups per CU), summed into WT; GEMMA_HIP_EIGH_KSLICES=1: the two-way cut on two
      // streams of rounds 2-3
      const char *eks = getenv("GEMMA_HIP_EIGH_KSLICES");
      const long tiles = (m + GEMM_BN - 1) / GEMM_BN;
      int want = eks ? atoi(eks) : (int)std::min<long>(E2_MAXSLICE, std::max<long>(1, (2L * ncu + tiles - 1) / tiles));
      if (want > E2_MAXSLICE) want = E2_MAXSLICE;
      if (want >= 2 && w2.ZS) {
        int ns = 1;
        EIG_HIP(launch_dgemm_ksliced('N', 'N', E2_B, m, m, 1.0, Vr, n, A22, n, w2.ZS, n, (long)E2_B * n, want, &ns, s));
        if (ns > 1) {
          hipLaunchKernelGGL(e2_sumk_kernel, dim3((unsigned)((m / 2 + 255) / 256 + 1), (unsigned)E2_B), dim3(256), 0, s, ws.WT, w2.ZS, ns,
                             (long)E2_B * n, m, n);
          EIG_HIP(hipGetLastError());
          EIG_HIP(launch_dgemm('T', 'N', E2_B, m, E2_B, 1.0, T, E2_B, ws.WT, n, 0.0, Wr, n, false, false, s));
        } else {
          EIG_HIP(launch_dgemm('T', 'N', E2_B, m, E2_B, 1.0, T, E2_B, w2.ZS, n, 0.0, Wr, n, false, false, s));
        }
      } else {
        rc = eig2_dgemm_split2('N', 'N', E2_B, m, m, 1.0, Vr, n, A22, n, ws.WT, SB, n, s, msg);
        if (rc) return rc;
        EIG_HIP(launch_dgemm('T', 'N', E2_B, m, E2_B, 1.0, T, E2_B, ws.WT, n, 0.0, Wr, n, false, false, s));
        EIG_HIP(launch_dgemm('T', 'N', E2_B, m, E2_B, 1.0, T, E2_B, SB, n, 1.0, Wr, n, false, false, s));
      }
    }
    // W^T = Y^T - (Mid^T T / 2) V^T,  Mid = V^T Y
    rc = eig2_gram(Vr, Wr, n, m, w2.gramP, ws.S, s, msg);
    if (rc) return rc;
    EIG_HIP(launch_dgemm('T', 'N', E2_B, E2_B, E2_B, 0.5, ws.S, E2_B, T, E2_B, 0.0, ws.T, E2_B, false, false, s));
    EIG_HIP(launch_dgemm('N', 'N', E2_B, m, E2_B, -1.0, ws.T, E2_B, Vr, n, 1.0, Wr, n, false, false, s));
    // A22 -= V W^T + W V^T as ONE rank-256 product on the upper-triangle tiles ([V; W]^T [W; V]), then the mirror: the
    // trailing matrix is read and written once instead of twice, and half of the tiles are never formed
    EIG_HIP(hipMemcpy2DAsync(SA, n * 8, Vr, n * 8, m * 8, E2_B, hipMemcpyDeviceToDevice, s));
    EIG_HIP(hipMemcpy2DAsync(SB, n * 8, Wr, n * 8, m * 8, E2_B, hipMemcpyDeviceToDevice, s));
    EIG_HIP(hipMemcpy2DAsync(SB + (size_t)E2_B * n, n * 8, Vr, n * 8, m * 8, E2_B, hipMemcpyDeviceToDevice, s));
    // round 5: the product's epilogue writes the transposed tiles itself (GemmArgs::mirror) -- the mirror pass was 0.51 s of the
    // 4.56 s of this stage at n = 50 000 (390 launches, a read and a write of half the trailing matrix each);
    // GEMMA_HIP_EIGH_MIRROR=pass restores it
    const long m2 = m - E2_B; // rows below the next panel's band block
    if (lookahead && m2 >= 2 && (la_mode == 1 || !persist || (m2 + 127) / 128 > ncu) && la_streams()) {
      EIG_HIP(launch_dgemm('T', 'N', E2_B, m, 2 * E2_B, -1.0, SA, n, SB, n, 1.0, A22, n, false, false, s, true));
      EIG_HIP(hipEventRecord(la_ready, s));
      EIG_HIP(hipStreamWaitEvent(la_stream, la_ready, 0));
      EIG_HIP(launch_dgemm('T', 'N', m2, m2, 2 * E2_B, -1.0, SA + E2_B, n, SB + E2_B, n, 1.0, A22 + (size_t)E2_B * n + E2_B, n, true, false, s,
                           true));
      const int rcp = factor_panel(r0, la_stream);
      if (rcp) return rcp;
      EIG_HIP(hipEventRecord(la_done, la_stream));
      EIG_HIP(hipStreamWaitEvent(s, la_done, 0));
      have_panel = true;
      continue;
    }
    EIG_HIP(launch_dgemm('T', 'N', m, m, 2 * E2_B, -1.0, SA, n, SB, n, 1.0, A22, n, true, false, s, fused_mirror));
    if (!fused_mirror) {
      const unsigned nb32 = (unsigned)((m + 31) / 32);
      hipLaunchKernelGGL(e2_mirror_upper_kernel, dim3(nb32, nb32), dim3(32, 8), 0, s, A22, m, n);
    }
  }
  // 128 columns of zeroed slack behind the band: the chase reads whole 128 x 128 blocks without bounds checks
  hipLaunchKernelGGL(sb_extract_band_kernel, dim3((unsigned)(n + E2_B)), dim3(256), 0, s, A, n, w2.betas, w2.Bd);
  EIG_HIP(hipGetLastError());
  return 0;
}

// band -> tridiagonal (ws.d, ws.e); reflectors in w2.V2 / w2.tau2
static inline int eig2_sb2st(long n, EigWs &ws, Eig2Ws &w2, hipStream_t s, std::string &msg) {
  static bool attr = false;
  if (!attr) {
    EIG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(bc_step_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, BC_LDS_DOUBLES * 8));
    attr = true;
  }
  EIG_HIP(hipMemsetAsync(w2.tau2, 0, (size_t)w2.kmaxall * n * 8, s));
  const long b = E2_B, tmax = 2 * (n - 3);
  {
    // one persistent launch when every workgroup can be resident (one per CU: 140 KB of LDS each); GEMMA_HIP_EIGH_BC=steps
    // forces the per-step launches
    static int ncu = 0;
    if (!ncu) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
      if (ncu <= 0) ncu = 1;
      EIG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(bc_persist_kernel<false, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, BC_LDS_DOUBLES * 8));
      EIG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(bc_persist_kernel<true, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, BC_LDS_DOUBLES * 8));
      EIG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(bc_persist_kernel<false, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, BC_LDS_DOUBLES * 8));
      EIG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(bc_persist_kernel<true, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, BC_LDS_DOUBLES * 8));
    }
    const char *eb = getenv("GEMMA_HIP_EIGH_BC");
    const long nwg = (w2.kmaxall + 1) / 2;
    if (!(eb && eb[0] == 's') && nwg <= ncu) {
      EIG_HIP(hipMemsetAsync(w2.prog, 0, (size_t)(2 * w2.kmaxall + 4) * sizeof(int), s));
      EIG_HIP(hipMemcpyAsync(w2.Bd0, w2.Bd, (size_t)(n + E2_B) * E2_LDB * 8, hipMemcpyDeviceToDevice, s));
      BcPersistArgs pa{w2.Bd, n, w2.V2, w2.tau2, w2.prog, w2.prog + w2.kmaxall + 1, nullptr};
      const char *ed = getenv("GEMMA_HIP_EIGH_BC_DBG");
      long long *dbg_d = nullptr;
      if (ed && ed[0] == '1' && hipMalloc(reinterpret_cast<void **>(&dbg_d), 512 * 16 * 8) == hipSuccess) {
        (void)hipMemsetAsync(dbg_d, 0, 512 * 16 * 8, s);
        pa.dbg = dbg_d;
      }
      const char *esc = getenv("GEMMA_HIP_EIGH_BC_SC1"); // 0: acquire / release fences around every task (rounds 2-3)
      const bool sc1 = !(esc && esc[0] == '0');
      // GEMMA_HIP_EIGH_BC_PIPE=0: two positions per workgroup, hand-over at the end of a task (rounds 2-3)
      const char *epi = getenv("GEMMA_HIP_EIGH_BC_PIPE");
      const bool pipe1 = sc1 && !(epi && epi[0] == '0') && w2.kmaxall <= ncu;
      if (pipe1) {
        static bool attr1 = false;
        if (!attr1) {
          EIG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(bc_persist1_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, BC_LDS_DOUBLES * 8));
          EIG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(bc_persist1_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, BC_LDS_DOUBLES * 8));
          attr1 = true;
        }
        // ints: prog [0, kmaxall], err, vprog from kmaxall + 2, eprog from 2 kmaxall + 4; then the corner slots (doubles, all-ones = empty)
        const size_t cbox_off = ((size_t)3 * w2.kmaxall + 8 + 1) & ~(size_t)1;
        double *cbox = reinterpret_cast<double *>(w2.prog + cbox_off);
        EIG_HIP(hipMemsetAsync(w2.prog + w2.kmaxall + 2, 0, (cbox_off - w2.kmaxall - 2) * sizeof(int), s));
        EIG_HIP(hipMemsetAsync(cbox, 0xFF, (size_t)2 * (w2.kmaxall + 2) * sizeof(double), s));
        const bool handshake = !(epi && epi[0] == '1');
        BcPersist1Args p1{w2.Bd,   n,    w2.V2, w2.tau2, w2.prog, w2.prog + w2.kmaxall + 2, w2.prog + 2 * w2.kmaxall + 4, w2.prog + w2.kmaxall + 1,
                          cbox, handshake ? 1 : 0, dbg_d};
        if (dbg_d) hipLaunchKernelGGL(bc_persist1_kernel<true>, dim3((unsigned)w2.kmaxall), dim3(BC_THREADS), BC_LDS_DOUBLES * 8, s, p1);
        else hipLaunchKernelGGL(bc_persist1_kernel<false>, dim3((unsigned)w2.kmaxall), dim3(BC_THREADS), BC_LDS_DOUBLES * 8, s, p1);
      } else if (dbg_d) {
        if (sc1) hipLaunchKernelGGL((bc_persist_kernel<true, true>), dim3((unsigned)nwg), dim3(BC_THREADS), BC_LDS_DOUBLES * 8, s, pa);
        else hipLaunchKernelGGL((bc_persist_kernel<true, false>), dim3((unsigned)nwg), dim3(BC_THREADS), BC_LDS_DOUBLES * 8, s, pa);
      } else {
        // GEMMA_HIP_EIGH_BC_COOP=1: a COOPERATIVE launch -- the runtime starts the grid only with every workgroup resident, which
        // is what the progress counters assume (a kernel of another stream holding a few CUs would otherwise leave some
        // workgroups waiting for neighbours that have not started: bounded waits, then the slow per-step fall-back below).
        // NOT the default (round 3, measured): the chase itself runs as fast either way (0.86 s at n = 20 000), but a process
        // that has made one cooperative launch keeps a cooperative queue, and from then on ANOTHER process on the same device
        // runs at half speed while the first merely exists (bench.py's end-to-end child: 7.8 s instead of 4.3 s, its
        // eigensolver 5.5 s instead of 2.7 s; back to 4.35 s with the plain launch in the parent).  Ranks that share a device
        // -- the tests, the shm transport -- must not pay that.
        const char *ec = getenv("GEMMA_HIP_EIGH_BC_COOP");
        bool launched = false;
        if (ec && ec[0] == '1') {
          void *kargs[] = {reinterpret_cast<void *>(&pa)};
          launched = hipLaunchCooperativeKernel(sc1 ? reinterpret_cast<const void *>(bc_persist_kernel<false, true>)
                                                    : reinterpret_cast<const void *>(bc_persist_kernel<false, false>), dim3((unsigned)nwg),
                                                dim3(BC_THREADS), kargs, (unsigned)(BC_LDS_DOUBLES * 8), s) == hipSuccess;
          if (!launched) (void)hipGetLastError();
        }
        if (!launched) {
          if (sc1) hipLaunchKernelGGL((bc_persist_kernel<false, true>), dim3((unsigned)nwg), dim3(BC_THREADS), BC_LDS_DOUBLES * 8, s, pa);
          else hipLaunchKernelGGL((bc_persist_kernel<false, false>), dim3((unsigned)nwg), dim3(BC_THREADS), BC_LDS_DOUBLES * 8, s, pa);
        }
      }
      EIG_HIP(hipGetLastError());
      if (dbg_d) {
        std::vector<long long> hs(512 * 16);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(hs.data(), dbg_d, hs.size() * 8, hipMemcpyDeviceToHost);
        (void)hipFree(dbg_d);
        double acc[9] = {0};
        int cnt = 0;
        for (int jj = 128; jj < 512; ++jj) {
          const long long *q = hs.data() + 16 * jj;
          if (!q[0] || !q[8]) continue;
          for (int i = 1; i <= 8; ++i) acc[i] += (double)(q[i] - q[i - 1]) * 0.01;
          acc[0] += (jj + 1 < 512 && hs[16 * (jj + 1)]) ? (double)(hs[16 * (jj + 1)] - q[0]) * 0.01 : 0.0;
          ++cnt;
        }
        if (cnt)
          fprintf(stderr, "bulge chase, position 2 (us per task): load+LDS %.2f | y,update %.2f | larfg %.2f | z,store %.2f | "
                          "D to LDS %.2f | p,gamma %.2f | D update,store %.2f | fence+flag %.2f | sweep period %.2f\n",
                  acc[1] / cnt, acc[2] / cnt, acc[3] / cnt, acc[4] / cnt, acc[5] / cnt, acc[6] / cnt, acc[7] / cnt, acc[8] / cnt,
                  acc[0] / cnt);
      }
      int err = 0;
      EIG_HIP(hipMemcpyAsync(&err, w2.prog + w2.kmaxall + 1, sizeof(int), hipMemcpyDeviceToHost, s));
      EIG_HIP(hipStreamSynchronize(s));
      if (!err) {
        hipLaunchKernelGGL(sb_band_de_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w2.Bd, n, ws.d, ws.e);
        EIG_HIP(hipGetLastError());
        return 0;
      }
      // a workgroup waited too long for its neighbour (possible only when something else kept part of the chip busy, so that
      // not every workgroup was resident): the band is restored from the copy taken above and the chase repeated as one
      // launch per time step, which needs no co-residency
      EIG_HIP(hipMemcpyAsync(w2.Bd, w2.Bd0, (size_t)(n + E2_B) * E2_LDB * 8, hipMemcpyDeviceToDevice, s));
      EIG_HIP(hipMemsetAsync(w2.tau2, 0, (size_t)w2.kmaxall * n * 8, s));
    }
  }
  BcArgs a{w2.Bd, n, 0, 0, w2.V2, w2.tau2};
  for (long t = 0; t <= tmax; ++t) {
    const long num = t * b - n + 1;
    const long jlo = num >= 0 ? num / (2 * b - 1) + 1 : 0;
    const long jhi = std::min<long>(n - 3, t / 2);
    if (jlo > jhi) continue;
    a.t = t;
    a.jlo = jlo;
    hipLaunchKernelGGL(bc_step_kernel, dim3((unsigned)(jhi - jlo + 1)), dim3(BC_THREADS), BC_LDS_DOUBLES * 8, s, a);
    if ((t & 1023) == 0) EIG_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(sb_band_de_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w2.Bd, n, ws.d, ws.e);
  EIG_HIP(hipGetLastError());
  return 0;
}

// Z^T <- Z^T Q2^T (the reflectors of the bulge chase)
static inline int eig2_apply_q2(double *ZT, long n, long nrows, Eig2Ws &w2, hipStream_t s, std::string &msg) {
  Q2PackArgs pa{w2.V2, w2.tau2, n, w2.goff, w2.pack};
  hipLaunchKernelGGL(q2_pack_kernel, dim3((unsigned)w2.nJ, (unsigned)w2.kmaxall), dim3(256), 0, s, pa);
  EIG_HIP(hipGetLastError());
  Q2ApplyArgs aa{ZT, nrows, n, w2.pack, w2.goff, (int)w2.nJ, (int)w2.kmaxall, nullptr, nullptr, 0, 0};
  const int nrb = (int)((nrows + 63) / 64);
  int cus = 0;
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      cus = 0;
    (void)hipGetLastError();
  }
  // GEMMA_HIP_EIGH_Q2_DYNAMIC=0: one workgroup per row block for the whole launch (round 2); =1: the dynamic schedule even
  // where it cannot help (tests: small n); GEMMA_HIP_EIGH_Q2_WORKERS / _SEGMENTS override the persistent grid and the cut
  const char *ed = getenv("GEMMA_HIP_EIGH_Q2_DYNAMIC");
  // one workgroup alone on a CU does 1.07 units of work per unit of time, two do 1.41 together (measured: n = 16 384 / 32 768);
  // with C row blocks on 2 x CUs slots only C are ever runnable and they sit where the hardware put them, so below
  // C = 1.29 CUs (where 2 p (1 - p) 1.07 + p^2 1.41 = 1.07, p = C / (2 CUs)) one workgroup per CU is the better grid
  // (n = 20 000: 0.478 s against 0.512 s; one workgroup per row block: 0.570 s)
  int workers = (100L * nrb >= 129L * cus) ? 2 * cus : cus;
  if (const char *ew = getenv("GEMMA_HIP_EIGH_Q2_WORKERS")) workers = std::max(1, atoi(ew));
  const bool force = ed && ed[0] == '1';
  if (!(ed && ed[0] == '0') && cus > 0 && (force || (nrb > cus && nrb % workers != 0))) {
    // segments of chase steps with about the same number of groups each, ~24 tasks per persistent workgroup
    std::vector<long> groups((size_t)w2.kmaxall, 0);
    long total = 0;
    for (long k = 0; k < w2.kmaxall; ++k) {
      const long lim = n - 2 - k * E2_B;
      if (lim < 0) break;
      groups[k] = std::min<long>(lim / E2_NB, w2.nJ - 1) + 1;
      total += groups[k];
    }
    int nseg = (int)std::min<long>(std::min<long>(Q2_MAXSEG, w2.kmaxall), (24L * workers + nrb - 1) / nrb);
    if (const char *es = getenv("GEMMA_HIP_EIGH_Q2_SEGMENTS")) nseg = std::max(1, std::min<int>(atoi(es), std::min<long>(Q2_MAXSEG, w2.kmaxall)));
    std::vector<int> hs((size_t)nrb + 2, 0), kseg;
    kseg.push_back(0);
    long acc = 0;
    for (long k = 0; k < w2.kmaxall && (int)kseg.size() < nseg; ++k) {
      acc += groups[k];
      if (acc * nseg >= total * (long)kseg.size() && k + 1 < w2.kmaxall) kseg.push_back((int)(k + 1));
    }
    nseg = (int)kseg.size();
    kseg.push_back((int)w2.kmaxall);
    EIG_HIP(hipMemsetAsync(w2.q2sync, 0, ((size_t)nrb + 2) * sizeof(int), s));
    EIG_HIP(hipMemcpyAsync(w2.q2sync + nrb + 2, kseg.data(), kseg.size() * sizeof(int), hipMemcpyHostToDevice, s));
    EIG_HIP(hipStreamSynchronize(s)); // kseg is a local
    aa.sync = w2.q2sync;
    aa.kseg = w2.q2sync + nrb + 2;
    aa.nseg = nseg;
    aa.nrb = nrb;
    hipLaunchKernelGGL(q2_apply_kernel, dim3((unsigned)std::min(workers, nseg * nrb)), dim3(256), 0, s, aa);
    EIG_HIP(hipGetLastError());
    int flag = 0;
    EIG_HIP(hipMemcpyAsync(&flag, w2.q2sync + 1, sizeof(int), hipMemcpyDeviceToHost, s));
    EIG_HIP(hipStreamSynchronize(s));
    if (flag) {
      msg = "stage-2 back-transformation: a task waited too long for its predecessor";
      return 4;
    }
    return 0;
  }
  hipLaunchKernelGGL(q2_apply_kernel, dim3((unsigned)nrb), dim3(256), 0, s, aa);
  EIG_HIP(hipGetLastError());
  return 0;
}

// Panels per block reflector of the stage-1 back-transformation (round 6).  The rank-k update Z^T -= (Z^T Y^T T^T) Y reads and writes
// the touched part of Z^T once per block reflector: with two panels (k = 256, round 4) it moves 16 n Kc bytes per 1024 n Kc flop --
// 64 flop per byte, 1.1 TB/s at the GEMM's 73 TFLOP/s, and the update ran at 56; with four panels (k = 512) the traffic halves again.
// GEMMA_HIP_EIGH_Q1_GROUP = 1 | 2 | 4 | 8 (default 4); 2 takes the pair code of round 4.
constexpr int Q1_SLICES = 16;
static inline int eig2_q1_group() {
  const char *e = getenv("GEMMA_HIP_EIGH_Q1_GROUP");
  const int g = e ? atoi(e) : 4;
  return (g == 1 || g == 2 || g == 4 || g == 8) ? g : 4;
}
static inline long eig2_q1_kp(long n) {
  long kp = (long)eig2_q1_group() * E2_B;
  while (kp > 2 * E2_B && kp > n / 2) kp /= 2; // no wider than makes sense for a small matrix
  return kp;
}

// D (rows x cols, ld ldd) = 0 except: copies of `nblk` diagonal blocks T_i (E2_B x E2_B, ld E2_B) from Tsrc + i * E2_B^2
__global__ void q1_tdiag_kernel(double *__restrict__ D, long kp, const double *__restrict__ Tsrc, int nblk) {
  const long i = (long)blockIdx.y, j = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= kp || j >= kp) return;
  const long bi = i / E2_B, bj = j / E2_B;
  D[i * kp + j] = (bi == bj && bi < nblk) ? Tsrc[bi * (long)E2_B * E2_B + (i % E2_B) * E2_B + (j % E2_B)] : 0.0;
}

// The compact-WY factor of g consecutive panels pa .. pa + g - 1 (g a power of two) in TG (kp x kp, kp = g * 128):
//   (I - Va Ta Va^T)(I - Vb Tb Vb^T) = I - [Va Vb] [[Ta, -Ta (Va^T Vb) Tb], [0, Tb]] [Va Vb]^T, applied level by level;
// every Va^T Vb is a block of ONE Gram matrix Y Y^T of the group's reflectors (the later panels' rows are zero in front of their own
// first column, so the common column range changes nothing).
static inline int eig2_q1_group_T(const double *Y, long n, long Kc, int g, const double *Tpanels, Eig2Ws &w2, hipStream_t s,
                                  std::string &msg) {
  const long kp = (long)g * E2_B;
  double *Sg = w2.Qg, *TG = w2.Qg + kp * kp, *W = w2.Qg + 2 * kp * kp, *SL = w2.Qg + 3 * kp * kp;
  int ns = 1;
  EIG_HIP(launch_dgemm_ksliced('N', 'T', kp, kp, Kc, 1.0, Y, n, Y, n, SL, kp, kp * kp, Q1_SLICES, &ns, s));
  hipLaunchKernelGGL(e2_sumk_kernel, dim3((unsigned)((kp / 2 + 255) / 256), (unsigned)kp), dim3(256), 0, s, Sg, SL, ns, kp * kp, kp, kp);
  hipLaunchKernelGGL(q1_tdiag_kernel, dim3((unsigned)((kp + 255) / 256), (unsigned)kp), dim3(256), 0, s, TG, kp, Tpanels, g);
  EIG_HIP(hipGetLastError());
  for (int h = 1; h < g; h *= 2) {
    const long m = (long)h * E2_B;
    for (int i = 0; i + 2 * h <= g; i += 2 * h) {
      const long a0 = (long)i * E2_B, b0 = a0 + m;
      // W = Ta * S_ab ; T_ab = -W * Tb
      EIG_HIP(launch_dgemm('N', 'N', m, m, m, 1.0, TG + a0 * kp + a0, kp, Sg + a0 * kp + b0, kp, 0.0, W, m, false, false, s));
      EIG_HIP(launch_dgemm('N', 'N', m, m, m, -1.0, W, m, TG + b0 * kp + b0, kp, 0.0, TG + a0 * kp + b0, kp, false, false, s));
    }
  }
  return 0;
}

static inline int eig2_apply_q1_grouped(double *ZT, long n, long nrows, EigWs &ws, Eig2Ws &w2, int G, hipStream_t s, std::string &msg) {
  const long nb2 = (long)E2_B * E2_B;
  long npan = 0;
  for (long j0 = 0; n - (j0 + E2_B) >= 2; j0 += E2_B) ++npan;
  long pnl = npan - 1;
  while (pnl >= 0) {
    int g = G;
    while (g > 1 && (g > pnl + 1 || (long)g * E2_B > w2.q1_kpmax)) g /= 2; // groups counted from the last panel; what is left over in smaller ones
    const long pa = pnl - g + 1;
    const long j0 = pa * E2_B, c0 = j0 + E2_B, Kc = n - c0;
    const long kp = (long)g * E2_B;
    const double *Y = ws.VT + j0 * n + c0;
    const double *T = ws.Tall + pa * nb2;
    long ldt = E2_B;
    if (g > 1) {
      int rc = eig2_q1_group_T(Y, n, Kc, g, ws.Tall + pa * nb2, w2, s, msg);
      if (rc) return rc;
      T = w2.Qg + kp * kp;
      ldt = kp;
    }
    double *P = w2.P256, *P2 = w2.P256 + (size_t)n * kp, *Pb = w2.P256 + (size_t)2 * n * kp;
    int rc = eig2_dgemm_split2('N', 'T', nrows, kp, Kc, 1.0, ZT + c0, n, Y, n, P, Pb, kp, s, msg);
    if (rc) return rc;
    EIG_HIP(launch_dgemm('N', 'T', nrows, kp, kp, 1.0, P, kp, T, ldt, 0.0, P2, kp, false, false, s));
    EIG_HIP(launch_dgemm('N', 'T', nrows, kp, kp, 1.0, Pb, kp, T, ldt, 1.0, P2, kp, false, false, s));
    EIG_HIP(launch_dgemm('N', 'N', nrows, Kc, kp, -1.0, P2, kp, Y, n, 1.0, ZT + c0, n, false, false, s));
    pnl = pa - 1;
  }
  return 0;
}

// Z^T <- Z^T Q1^T (stage-1 panels, compact WY: three GEMMs per panel as in eig_backtransform)
static inline int eig2_apply_q1(double *ZT, long n, long nrows, EigWs &ws, Eig2Ws &w2, hipStream_t s, std::string &msg) {
  const long nb2 = (long)E2_B * E2_B;
  long npan = 0;
  for (long j0 = 0; n - (j0 + E2_B) >= 2; j0 += E2_B) ++npan;
  // Two consecutive panels are applied as ONE block reflector of 256 vectors,
  //   (I - V1 T1 V1^T)(I - V2 T2 V2^T) = I - [V1 V2] [[T1, -T1 (V1^T V2) T2], [0, T2]] [V1 V2]^T:
  // the rank-k update of Z^T has K = 256 (its reads and writes of Z^T halve) and the skinny product Z^T Y^T has two tile
  // columns.  GEMMA_HIP_EIGH_Q1_PAIR=0: panel by panel.
  {
    const int G = eig2_q1_group();
    if (G >= 4 && w2.Qg && w2.q1_kpmax >= 4 * E2_B) return eig2_apply_q1_grouped(ZT, n, nrows, ws, w2, G, s, msg);
  }
  const char *ep = getenv("GEMMA_HIP_EIGH_Q1_PAIR");
  const bool pair = !(ep && ep[0] == '0') && w2.P256 != nullptr && eig2_q1_group() != 1;
  long pnl = npan - 1;
  while (pnl >= 0) {
    const bool two = pair && pnl >= 1 && ((npan - 1 - pnl) % 2 == 0); // pairs (pnl-1, pnl) counted from the last panel
    const long pa = two ? pnl - 1 : pnl;
    const long j0 = pa * E2_B, c0 = j0 + E2_B, Kc = n - c0;
    const long kp = two ? 2 * E2_B : E2_B;
    const double *Y = ws.VT + j0 * n + c0;
    const double *T = ws.Tall + pa * nb2;
    if (two) {
      const long cB = c0 + E2_B; // first column where the second panel's reflectors are non-zero
      const double *T1 = ws.Tall + pa * nb2, *T2 = ws.Tall + (pa + 1) * nb2;
      double *TP = w2.Tpair;
      int rc = eig2_gram(ws.VT + j0 * n + cB, ws.VT + (j0 + E2_B) * n + cB, n, n - cB, w2.gramP, ws.S, s, msg); // V1^T V2
      if (rc) return rc;
      EIG_HIP(launch_dgemm('N', 'N', E2_B, E2_B, E2_B, 1.0, T1, E2_B, ws.S, E2_B, 0.0, ws.T, E2_B, false, false, s));
      EIG_HIP(hipMemsetAsync(TP, 0, (size_t)4 * nb2 * 8, s));
      EIG_HIP(launch_dgemm('N', 'N', E2_B, E2_B, E2_B, -1.0, ws.T, E2_B, T2, E2_B, 0.0, TP + E2_B, 2 * E2_B, false, false, s));
      EIG_HIP(hipMemcpy2DAsync(TP, 2 * E2_B * 8, T1, E2_B * 8, E2_B * 8, E2_B, hipMemcpyDeviceToDevice, s));
      EIG_HIP(hipMemcpy2DAsync(TP + (size_t)E2_B * 2 * E2_B + E2_B, 2 * E2_B * 8, T2, E2_B * 8, E2_B * 8, E2_B,
                               hipMemcpyDeviceToDevice, s));
      T = TP;
    }
    double *P = two ? w2.P256 : ws.P, *P2 = two ? w2.P256 + (size_t)n * kp : ws.P2;
    double *Pb = two ? w2.P256 + (size_t)2 * n * kp : w2.YT;
    int rc = eig2_dgemm_split2('N', 'T', nrows, kp, Kc, 1.0, ZT + c0, n, Y, n, P, Pb, kp, s, msg);
    if (rc) return rc;
    EIG_HIP(launch_dgemm('N', 'T', nrows, kp, kp, 1.0, P, kp, T, kp, 0.0, P2, kp, false, false, s));
    EIG_HIP(launch_dgemm('N', 'T', nrows, kp, kp, 1.0, Pb, kp, T, kp, 1.0, P2, kp, false, false, s));
    EIG_HIP(launch_dgemm('N', 'N', nrows, Kc, kp, -1.0, P2, kp, Y, n, 1.0, ZT + c0, n, false, false, s));
    pnl = pa - 1;
  }
  return 0;
}

// (Round 3 tried to allocate the buffers stage 1 does not touch -- Delta, Wk, V2, pack: 76 GB at n = 50 000 -- on a helper thread
// beside stage 1: hipMalloc costs 0.5-2.9 s for the same sizes from box to box, but the runtime serialises it with the kernel
// launches of the other thread; stage 1 took exactly as much longer as the helper spent in hipMalloc.)
static inline bool eig2_alloc(long n, EigWs &ws, Eig2Ws &w2) {
  const long nsweep = n - 2;
  w2.kmaxall = (n - 1 + E2_B - 1) / E2_B;
  w2.nJ = (nsweep + E2_NB - 1) / E2_NB;
  std::vector<long> goff((size_t)w2.nJ + 1, 0);
  for (long Jb = 0; Jb < w2.nJ; ++Jb) {
    const long J0 = Jb * E2_NB;
    goff[Jb + 1] = goff[Jb] + (n - 1 - J0 + E2_B - 1) / E2_B; // tasks of the block's first sweep
  }
  w2.ngroups = goff[w2.nJ];
  const size_t nwg_panel = (size_t)(n + SB_COLS - 1) / SB_COLS, nwg_gram = (size_t)(n + GR_CH - 1) / GR_CH;
  bool ok = ws.get(w2.Bd, (size_t)(n + E2_B) * E2_LDB) && ws.get(w2.Bd0, (size_t)(n + E2_B) * E2_LDB) && ws.get(w2.part, 2 * std::max(nwg_panel, (size_t)(n + E2_B - 1) / E2_B) * E2_B) && ws.get(w2.pbar, 4) && ws.get(w2.ppart, (size_t)E2_B * std::min<size_t>((size_t)(n + E2_B - 1) / E2_B, 1024) * E2_B) && ws.get(w2.pheads, (size_t)E2_B * E2_B) && ws.get(w2.ZS, (size_t)E2_MAXSLICE * E2_B * n) && ws.get(w2.heads, 2 * E2_B) &&
            ws.get(w2.betas, n) && ws.get(w2.gramP, nwg_gram * E2_B * E2_B) && ws.get(w2.YT, (size_t)4 * E2_B * n) &&
            ws.get(w2.V2, (size_t)w2.kmaxall * n * E2_B) && ws.get(w2.tau2, (size_t)w2.kmaxall * n) &&
            ws.get(w2.goff, (size_t)w2.nJ + 1) &&
            ws.get(w2.prog, 8 * (size_t)w2.kmaxall + 32) && ws.get(w2.P256, (size_t)3 * n * std::max<long>(2 * E2_B, eig2_q1_kp(n))) &&
            ws.get(w2.Qg, (size_t)eig2_q1_kp(n) * eig2_q1_kp(n) * (2 + 1 + Q1_SLICES)) &&
            ws.get(w2.Tpair, (size_t)4 * E2_B * E2_B) && ws.get(w2.q2sync, (size_t)(n + 63) / 64 + 2 + Q2_MAXSEG + 2);
  // the packed groups of the stage-2 back-transformation (0.8 n^2 doubles) are written after the divide & conquer and dead before
  // the final transpose: they live in the divide & conquer's Delta buffer (n^2), which is free in between (round 4: 16 GB less
  // to allocate at n = 50 000); a caller without that buffer (the stage diagnostics) or a pack that would not fit gets its own
  if (ok) {
    if (ws.Delta && (size_t)w2.ngroups * E2_PACK <= (size_t)n * n) w2.pack = ws.Delta;
    else ok = ws.get(w2.pack, (size_t)w2.ngroups * E2_PACK);
  }
  if (!ok) return false;
  w2.q1_kpmax = eig2_q1_kp(n);
  return hipMemcpy(w2.goff, goff.data(), goff.size() * sizeof(long), hipMemcpyHostToDevice) == hipSuccess;
}

} // namespace gemma_hip
