// Part of gemma_hip.hip (ONE translation unit: the parts share the context g_ctx and the helpers of its anonymous namespace, and are
// included there in this order; round 6: the 3 500-line file cut along its stages for reading -- no behaviour change).
// This part: CenterMatrix, EigenDecomp_Zeroed (the solver itself is eigh_tu.hip), CalcUtX, first-pass QC, LOCO kinship.

// ------------------------------------------------------------------------------ centring / eigen
extern "C" int gemma_hip_center_d(double *G, size_t n, void *stream) {
  NEED_INIT();
  if (!G || n == 0) return fail(GEMMA_HIP_EINVAL, "center: empty matrix");
  hipStream_t s = S(stream);
  if (g_ctx.scratch.reserve((n + 1) * 8)) return fail(GEMMA_HIP_ENOMEM, "center: scratch");
  double *Gw = g_ctx.scratch.as<double>();
  double *d = Gw + n;
  hipLaunchKernelGGL(rowsum_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, G, (long)n, (long)n, Gw);
  hipLaunchKernelGGL(total_kernel, dim3(1), dim3(1024), 0, s, Gw, (long)n, d);
  hipLaunchKernelGGL(center_update_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, s,
                     G, (long)n, (long)n, Gw, d);
  HIPCHK(hipGetLastError());
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_center(double *G, size_t n) {
  NEED_INIT();
  if (!G || n == 0) return fail(GEMMA_HIP_EINVAL, "center: empty matrix");
  DevBuf d;
  if (d.reserve(n * n * 8)) return fail(GEMMA_HIP_ENOMEM, "center: %zu bytes", n * n * 8);
  hipError_t e = hipMemcpy(d.p, G, n * n * 8, hipMemcpyHostToDevice);
  int rc = GEMMA_HIP_OK;
  if (e == hipSuccess) rc = gemma_hip_center_d(d.as<double>(), n, nullptr);
  if (e == hipSuccess && rc == GEMMA_HIP_OK) e = hipMemcpy(G, d.p, n * n * 8, hipMemcpyDeviceToHost);
  d.release();
  if (e != hipSuccess) return fail(GEMMA_HIP_ERUNTIME, "center: %s", hipGetErrorString(e));
  return rc;
}

// The communicator's two collectives as the eigensolver's unit sees them (eigh_tu.h: EighShard)
static int shard_bcast(void *, void *buf_d, size_t bytes, int root, hipStream_t s) {
  std::string err;
  return g_ctx.comm.bcast(buf_d, bytes, root, s, err) ? 1 : 0;
}
static int shard_allreduce(void *, double *buf_d, size_t count, hipStream_t s) {
  std::string err;
  return g_ctx.comm.allreduce_sum(buf_d, count, s, err) ? 1 : 0;
}
// the collective form of the eigensolver is in force for this call: fills sh
static bool eigh_shard_in_force(bool sharded, EighShard &sh) {
  const char *es = getenv("GEMMA_HIP_EIGH_SHARD"); // 0: every rank decomposes on its own (replicas), nothing is exchanged
  const bool use = sharded && g_ctx.comm.active && g_ctx.comm.world > 1 && !(es && es[0] == '0');
  if (use) {
    sh.rank = g_ctx.comm.rank;
    sh.world = g_ctx.comm.world;
    sh.bcast = shard_bcast;
    sh.allreduce_sum = shard_allreduce;
  }
  return use;
}
// a rank whose OWN setup fails before the collective solver (its copy of the matrix, its slot of the kept (U, eval))
// tells the others through the solver's first agreement instead of leaving them in it (eigh.hip.h: eigh_collective_abort)
static void eigh_abort_if_sharded(bool sharded, size_t n, hipStream_t s) {
  EighShard sh;
  if (eigh_shard_in_force(sharded, sh)) eigh_abort_x((long)n, s, &sh);
}
static int eigh_d_impl(double *G, size_t n, double *U, double *eval, double *trace_G, void *stream, bool sharded) {
  NEED_INIT();
  if (!G || !U || !eval || n == 0) return fail(GEMMA_HIP_EINVAL, "eigh: null/empty argument");
  hipStream_t s = S(stream);
  ProfScope ps(GEMMA_STAGE_EIGH, s);
  std::string msg;
  EighShard sh;
  const bool use = eigh_shard_in_force(sharded, sh);
  if (use) {
    // tests (tests/test_gpu_two_rank.py): GEMMA_HIP_EIGH_FAIL_RANK=<r> makes rank r fail as if its own allocations had, before the solver
    const char *efr = getenv("GEMMA_HIP_EIGH_FAIL_RANK");
    if (efr && *efr && atoi(efr) == g_ctx.comm.rank) {
      eigh_abort_x((long)n, s, &sh);
      return fail(GEMMA_HIP_ENOMEM, "eigh: allocation failure injected on rank %d (GEMMA_HIP_EIGH_FAIL_RANK)", g_ctx.comm.rank);
    }
  }
  int rc = eigh_device_x(G, (long)n, U, eval, s, msg, use ? &sh : nullptr);
  if (rc != GEMMA_HIP_OK) return fail(rc, "eigh: %s", msg.c_str());
  // EigenDecomp_Zeroed: eval < 1e-10 -> 0, trace = mean(eval)
  if (g_ctx.scratch.reserve(8)) return fail(GEMMA_HIP_ENOMEM, "eigh: scratch");
  hipLaunchKernelGGL(zero_small_eval_kernel, dim3(1), dim3(1024), 0, s, eval, (long)n,
                     g_ctx.scratch.as<double>());
  HIPCHK(hipGetLastError());
  double tr = 0.0;
  HIPCHK(hipMemcpyAsync(&tr, g_ctx.scratch.p, 8, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  if (trace_G) *trace_G = tr;
  return GEMMA_HIP_OK;
}
extern "C" int gemma_hip_eigh_d(double *G, size_t n, double *U, double *eval, double *trace_G, void *stream) {
  return eigh_d_impl(G, n, U, eval, trace_G, stream, false);
}
// COLLECTIVE over the library's communicator (gemma_hip_comm_init): every rank passes the same G and receives the same
// (U, eval); the back-transformations are shared out (csrc/eigh.hip.h "Several ranks").  One rank: gemma_hip_eigh_d.
extern "C" int gemma_hip_eigh_sharded_d(double *G, size_t n, double *U, double *eval, double *trace_G, void *stream) {
  return eigh_d_impl(G, n, U, eval, trace_G, stream, true);
}

// The eigensolver's workspace (~5 n^2 doubles) ahead of the solve, kept between solves (csrc/eigh.hip.h, EigPool).
extern "C" int gemma_hip_eigh_reserve(size_t n) {
  NEED_INIT();
  // may run on a thread of its own (include/gemma_hip.h): HIP's current device is per thread, and a fresh thread starts on device 0
  if (g_ctx.device >= 0 && hipSetDevice(g_ctx.device) != hipSuccess) {
    (void)hipGetLastError();
    return fail(GEMMA_HIP_ERUNTIME, "eigh_reserve: cannot select device %d", g_ctx.device);
  }
  std::string msg;
  const int rc = eigh_reserve_x((long)n, msg);
  if (rc) return fail(rc, "%s", msg.c_str());
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_eigh_release(size_t *bytes_freed) {
  const size_t b = eigh_release_x();
  if (bytes_freed) *bytes_freed = b;
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_dbg_eigh_last(double *t8) {
  if (!t8) return fail(GEMMA_HIP_EINVAL, "dbg_eigh_last: null argument");
  eigh_last_stages(t8);
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_eigh(double *G, size_t n, double *U, double *eval, double *trace_G) {
  NEED_INIT();
  if (!G || !U || !eval || n == 0) return fail(GEMMA_HIP_EINVAL, "eigh: null/empty argument");
  DevBuf dG, dU, dE;
  if (dG.reserve(n * n * 8) || dU.reserve(n * n * 8) || dE.reserve(n * 8)) {
    dG.release(); dU.release(); dE.release();
    return fail(GEMMA_HIP_ENOMEM, "eigh: cannot allocate 2 x %zu bytes", n * n * 8);
  }
  int rc = GEMMA_HIP_OK;
  hipError_t e = hipMemcpy(dG.p, G, n * n * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) rc = gemma_hip_eigh_d(dG.as<double>(), n, dU.as<double>(), dE.as<double>(), trace_G, nullptr);
  if (e == hipSuccess && rc == GEMMA_HIP_OK) e = hipMemcpy(U, dU.p, n * n * 8, hipMemcpyDeviceToHost);
  if (e == hipSuccess && rc == GEMMA_HIP_OK) e = hipMemcpy(eval, dE.p, n * 8, hipMemcpyDeviceToHost);
  dG.release(); dU.release(); dE.release();
  if (e != hipSuccess) return fail(GEMMA_HIP_ERUNTIME, "eigh: %s", hipGetErrorString(e));
  return rc;
}

// ---- diagnostics for the eigensolver stages (used by tests/test_gpu_eigh.py); bodies in eigh_tu.hip ----
extern "C" int gemma_hip_dbg_tridiag(const double *G, size_t n, double *d, double *e, double *tau, double *VT) {
  NEED_INIT();
  std::string msg;
  const int rc = dbg_tridiag_x(G, n, d, e, tau, VT, msg);
  if (rc) return fail(rc, "dbg_tridiag: %s", msg.c_str());
  return GEMMA_HIP_OK;
}
extern "C" int gemma_hip_dbg_eigh2(const double *G, size_t n, double *band, double *d, double *e) {
  NEED_INIT();
  std::string msg;
  const int rc = dbg_eigh2_x(G, n, band, d, e, msg);
  if (rc) return fail(rc, "dbg_eigh2: %s", msg.c_str());
  return GEMMA_HIP_OK;
}
extern "C" int gemma_hip_dbg_stedc(const double *d, const double *e, size_t n, double *w, double *ZT) {
  NEED_INIT();
  std::string msg;
  const int rc = dbg_stedc_x(d, e, n, w, ZT, msg);
  if (rc) return fail(rc, "dbg_stedc: %s", msg.c_str());
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_calc_utx(const double *U, const double *X, size_t n, size_t m, double *UtX) {
  // UtX (n x m) = U^T X : fast_dgemm("T","N",1.0,U,X,0.0,UtX), src/mathfunc.cpp:505
  return gemma_hip_dgemm('T', 'N', n, m, n, 1.0, U, n, X, m, 0.0, UtX, m);
}

// ------------------------------------------------------------------------------ first-pass QC
extern "C" int gemma_hip_snp_qc(int kind, const void *geno, size_t l, size_t ld, const int *indicator_idv,
                                size_t ni_total, const double *W, size_t n, size_t n_cvt, const gemma_qc_cfg *cfg,
                                int *indicator_snp, double *maf, size_t *n_miss) {
  NEED_INIT();
  if (!geno || !W || !cfg || !indicator_snp || n == 0 || n_cvt == 0 || ni_total < n)
    return fail(GEMMA_HIP_EINVAL, "snp_qc: bad arguments");
  if (kind != GEMMA_GENO_F64_SNP_MAJOR && kind != GEMMA_GENO_PLINK_2BIT)
    return fail(GEMMA_HIP_EINVAL, "snp_qc: geno_kind %d not supported here", kind);
  const size_t need = (kind == GEMMA_GENO_PLINK_2BIT) ? (ni_total + 3) / 4 : ni_total;
  if (ld < need) return fail(GEMMA_HIP_EINVAL, "snp_qc: ld=%zu < %zu", ld, need);
  if (l == 0) return GEMMA_HIP_OK;
  std::vector<int> map;
  if (indicator_idv) {
    for (size_t i = 0; i < ni_total; ++i)
      if (indicator_idv[i] != 0) map.push_back((int)i);
    if (map.size() != n) return fail(GEMMA_HIP_EINVAL, "snp_qc: %zu analysed individuals, n = %zu", map.size(), n);
  } else if (ni_total != n) {
    return fail(GEMMA_HIP_EINVAL, "snp_qc: no indicator but ni_total != n");
  }
  // W^T W and its inverse (host, c x c), W^T (device, covariate-major)
  const int c = (int)n_cvt;
  std::vector<double> WtW((size_t)c * c, 0.0), Wt((size_t)c * n);
  for (size_t i = 0; i < n; ++i)
    for (int a = 0; a < c; ++a) {
      Wt[(size_t)a * n + i] = W[i * c + a];
      for (int b = 0; b < c; ++b) WtW[(size_t)a * c + b] += W[i * c + a] * W[i * c + b];
    }
  if (!invert_small(WtW, c)) return fail(GEMMA_HIP_EINVAL, "snp_qc: W^T W is singular");
  const size_t esz = (kind == GEMMA_GENO_PLINK_2BIT) ? 1 : 8;
  const size_t ncol = QC_NSTAT + n_cvt;
  DevBuf &dG = g_ctx.qc_G, &dM = g_ctx.qc_M, &dW = g_ctx.qc_W, &dO = g_ctx.qc_O; // kept for the next block of the pass
  auto cleanup = [&]() { dG.release(); dM.release(); dW.release(); dO.release(); };
  if (dG.reserve(l * ld * esz) || dM.reserve(n * sizeof(int)) || dW.reserve(Wt.size() * 8) || dO.reserve(l * ncol * 8)) {
    cleanup();
    return fail(GEMMA_HIP_ENOMEM, "snp_qc: allocation");
  }
  hipError_t e = hipMemcpy2D(dG.p, ld * esz, geno, ld * esz, need * esz, l, hipMemcpyHostToDevice);
  if (e == hipSuccess && indicator_idv) e = hipMemcpy(dM.p, map.data(), n * sizeof(int), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dW.p, Wt.data(), Wt.size() * 8, hipMemcpyHostToDevice);
  std::vector<double> stats(l * ncol);
  if (e == hipSuccess) {
    QcArgs a;
    a.src = dG.p; a.ld = (long)ld; a.l = (long)l; a.idx_map = indicator_idv ? dM.as<int>() : nullptr;
    a.n = (int)n; a.c = c; a.Wt = dW.as<double>(); a.out = dO.as<double>();
    const unsigned grid = (unsigned)((l + 3) / 4);
    ProfScope ps(GEMMA_STAGE_INGEST, 0);
    if (kind == GEMMA_GENO_PLINK_2BIT)
      hipLaunchKernelGGL(snp_qc_kernel<true>, dim3(grid), dim3(256), 0, 0, a);
    else
      hipLaunchKernelGGL(snp_qc_kernel<false>, dim3(grid), dim3(256), 0, 0, a);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(stats.data(), dO.p, stats.size() * 8, hipMemcpyDeviceToHost);
  if (e != hipSuccess) {
    cleanup();
    return fail(GEMMA_HIP_ERUNTIME, "snp_qc: %s", hipGetErrorString(e));
  }
  QcCfgHost q = {cfg->maf_level, cfg->miss_level, cfg->hwe_level, cfg->r2_level};
  snp_qc_finish(stats.data(), l, (int)n, c, WtW.data(), kind == GEMMA_GENO_PLINK_2BIT, q, indicator_snp, maf, n_miss);
  return GEMMA_HIP_OK;
}

// K_loco = (ns_all * K_all - ns_chr * K_chr) / (ns_all - ns_chr)  (LOCO: the kinship of all SNPs not on a
// chromosome from the all-SNP kinship and the chromosome's own; in place on K_chr_d)
extern "C" int gemma_hip_kin_loco_d(const double *K_all_d, size_t ns_all, double *K_chr_d, size_t ns_chr, size_t n,
                                    void *stream) {
  NEED_INIT();
  if (!K_all_d || !K_chr_d || n == 0 || ns_all <= ns_chr) return fail(GEMMA_HIP_EINVAL, "kin_loco: bad arguments");
  const long total = (long)n * (long)n;
  hipLaunchKernelGGL(loco_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, S(stream), K_all_d, (double)ns_all,
                     K_chr_d, (double)ns_chr, total);
  HIPCHK(hipGetLastError());
  return GEMMA_HIP_OK;
}

// more than GEN_CMAX covariates: the wide kernels (one wavefront per workgroup, six tables of gen_ni_for(c) doubles in
// dynamic LDS)
static size_t wide_lds_bytes(size_t c) { return (size_t)6 * gen_ni_for((int)c) * 8; }
template <class K>
static int wide_attr(K kernel) {
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)wide_lds_bytes(GEN_CMAX_WIDE)));
  return GEMMA_HIP_OK;
}
