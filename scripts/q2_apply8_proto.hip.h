// PROTOTYPE, measured and dropped in round 6 (not compiled into the library; profiles/r06_eigh_q2_w8.txt).  It was built into eigh2.hip.h
// behind GEMMA_HIP_EIGH_Q2_WAVES=8 (host side: rows per block 128, one workgroup per CU, dynamic LDS Q2W8_LDS_BYTES): bit-identical to
// q2_apply_kernel on every shape tried and through the 78 eigensolver / two-rank tests, 201 registers and no scratch where the shipped kernel
// has 256 + 64 spilled -- and NOT faster: Q2 0.607 against 0.458 s at n = 20 000 (157 row blocks of 128 leave 99 CUs idle), 5.95 against
// 5.79 s at n = 50 000.  The shipped kernel's waits are not the register-starved operand reads its ISA suggested.
// The same transformation with EIGHT wavefronts per workgroup (round 6).  Counters of the four-wavefront kernel at n = 20 000
// (profiles/r06_eigh_q2_counters.txt): matrix pipe busy 0.57 at the full 2.38 GHz, 39 % of the wave cycles parked in a wait -- its ISA has
// every LDS operand read waited for on the spot: the kernel sits at its 256 registers (64 spilled) and the compiler has none left to read an
// operand ahead.  52 of those registers only carry the NEXT group's pack from global memory to LDS.  Here the pack goes to LDS by LDS-DMA
// (global_load_lds) into the other half of a double buffer -- no registers, ONE barrier per group instead of two -- which takes the whole
// CU's LDS (2 x 51 KiB + the group table), hence one workgroup of eight wavefronts (128 rows of Z^T) per CU instead of two of four: the
// same two wavefronts per SIMD, every pack fetched from L2 once per 128 rows instead of once per 64.  The arithmetic of a group is the
// four-wavefront kernel's, instruction for instruction (same bits).
constexpr int Q2W8_ROWS = 128;
constexpr int Q2W8_LDS_BYTES = 2 * E2_PACK * 8 + Q2_MAXJ * 4 + 16;
__global__ __launch_bounds__(512, 2) void q2_apply8_kernel(Q2ApplyArgs g) {
  extern __shared__ __attribute__((aligned(1024))) double q2lds[];
  int *sgoff = reinterpret_cast<int *>(q2lds + 2 * E2_PACK);
  int *s_task = sgoff + Q2_MAXJ;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), li = lane & 15, lk = lane >> 4;
  const long n = g.n;
  for (int i = t; i < g.nJ; i += 512) sgoff[i] = (int)g.goff[i];
  __syncthreads();
  constexpr int NT = E2_WIN / 16;
  constexpr int NPIECE = E2_PACK * 8 / 1024; // 51 one-KiB pieces of a pack
  static_assert(E2_PACK * 8 % 1024 == 0, "a pack is a whole number of LDS-DMA pieces");
  double x[4 * NT];
  const bool dyn = g.sync != nullptr;
  // pack `grp` -> half `half` of the double buffer: wavefront w moves pieces w, w + 8, ...
  auto pack_dma = [&](long grp, int half) {
    const char *src = reinterpret_cast<const char *>(g.pack + (size_t)grp * E2_PACK) + 16 * lane;
    char *dst = reinterpret_cast<char *>(q2lds + (size_t)half * E2_PACK);
#pragma unroll
    for (int q = 0; q < (NPIECE + 7) / 8; ++q) {
      const int pc = wave + 8 * q;
      if (pc < NPIECE) __builtin_amdgcn_global_load_lds((gemma_gptr_t)(src + 1024 * pc), (gemma_lptr_t)(dst + 1024 * pc), 16, 0, 0);
    }
  };
  for (;;) {
    int rb = blockIdx.x, kfirst = 0, kend = g.kmaxall, segi = 0;
    if (dyn) {
      if (t == 0) {
        int task = atomicAdd(g.sync, 1);
        if (task < g.nseg * g.nrb) {
          const int sg = task / g.nrb;
          if (sg > 0 && !bc_wait(g.sync + 2 + (task - sg * g.nrb), sg, g.sync + 1, Q2_WAIT_TICKS)) task = 0x7fffffff;
        }
        *s_task = task;
      }
      __syncthreads();
      const int task = *s_task;
      __syncthreads();
      if (task >= g.nseg * g.nrb) return;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      segi = task / g.nrb;
      rb = task - segi * g.nrb;
      kfirst = g.kseg[segi];
      kend = g.kseg[segi + 1];
    }
    const long row = (long)rb * Q2W8_ROWS + wave * 16 + li;
    const bool rok = row < g.nrows;
    double *zrow = g.ZT + (rok ? row : 0) * n;
    int half = 0;
    {
      const long lim0 = n - 2 - (long)kfirst * E2_B; // first group of this task: step kfirst, its last sweep block
      const long Jb0 = (lim0 / E2_NB < g.nJ - 1) ? lim0 / E2_NB : g.nJ - 1;
      pack_dma((long)sgoff[Jb0 < 0 ? 0 : Jb0] + kfirst, 0);
    }
    for (int k = kfirst; k < kend; ++k) {
      const long lim = n - 2 - (long)k * E2_B;
      if (lim < 0) break;
      long Jbmax = lim / E2_NB;
      if (Jbmax > g.nJ - 1) Jbmax = g.nJ - 1;
      for (long Jb = Jbmax; Jb >= 0; --Jb) {
        const long c0 = Jb * E2_NB + 1 + (long)k * E2_B;
        // this group's pack has landed (issued a whole group ago), and everyone is done with the other half: one rendezvous
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        {
          long nJb = Jb - 1, nk = k;
          if (nJb < 0) {
            nk = k + 1;
            const long nlim = n - 2 - nk * E2_B;
            nJb = (nk < kend && nlim >= 0) ? (nlim / E2_NB < g.nJ - 1 ? nlim / E2_NB : g.nJ - 1) : -1;
          }
          if (nJb >= 0) pack_dma((long)sgoff[nJb] + nk, half ^ 1);
        }
        const bool inside = c0 + E2_WIN <= n;
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
          for (int q = 0; q < 8; ++q) x[q] = zw[4 * q];
        }
        const double *Vd = q2lds + (size_t)half * E2_PACK, *Tm = Vd + E2_WIN * E2_VLD;
        e2_v4 wt0 = {0.0, 0.0, 0.0, 0.0}, wt1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kq = 0; kq < E2_WIN / 4; ++kq) {
          const int ks = (kq + 8) % (E2_WIN / 4);
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
        e2_v4 w20 = {0.0, 0.0, 0.0, 0.0}, w21 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < E2_NB / 4; ++ks) {
          const double b = (ks < 4) ? wt0[ks & 3] : wt1[ks & 3];
          const double a0 = Tm[li * E2_VLD + 4 * ks + lk], a1 = Tm[(16 + li) * E2_VLD + 4 * ks + lk];
          w20 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, w20, 0, 0, 0);
          w21 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, w21, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ct = 0; ct < NT; ct += 2) {
          e2_v4 acc0 = {x[4 * ct], x[4 * ct + 1], x[4 * ct + 2], x[4 * ct + 3]};
          e2_v4 acc1 = {x[4 * ct + 4], x[4 * ct + 5], x[4 * ct + 6], x[4 * ct + 7]};
#pragma unroll
          for (int ks = 0; ks < E2_NB / 4; ++ks) {
            const double b = (ks < 4) ? w20[ks & 3] : w21[ks & 3];
            if (!(ct == 0 && ks >= 4)) {
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
          __builtin_amdgcn_sched_barrier(0);
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
        half ^= 1;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (a pack requested for a step this task does not run cannot exist: nJb = -1 at the end)
    if (!dyn) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (t == 0) __hip_atomic_store(g.sync + 2 + rb, segi + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

