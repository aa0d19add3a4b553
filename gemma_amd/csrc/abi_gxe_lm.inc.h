// Part of gemma_hip.hip (ONE translation unit: the parts share the context g_ctx and the helpers of its anonymous namespace, and are
// included there in this order; round 6: the 3 500-line file cut along its stages for reading -- no behaviour change).
// This part: GXE, host-pointer batch, dbg_utx, the linear model (-lm), the null model, lmm_finish.

// ---- GXE variants: LMM::AnalyzeBimbamGXE / AnalyzePlinkGXE, src/lmm.cpp:2283-2608
// env over the analysed individuals (after lmm_setup): U^T env becomes the (c+1)-th shared covariate row (:2307-2309)
extern "C" int gemma_hip_lmm_set_env(const double *env) {
  NEED_INIT();
  if (!g_ctx.lmm_active) return fail(GEMMA_HIP_ESTATE, "lmm_set_env before lmm_setup");
  if (!env) return fail(GEMMA_HIP_EINVAL, "lmm_set_env: null pointer");
  const size_t n = g_ctx.cfg.n, c = g_ctx.cfg.n_cvt;
  if (c + 2 > (size_t)GEN_CMAX_WIDE)
    return fail(GEMMA_HIP_EINVAL, "lmm_set_env: n_cvt + 2 = %zu covariates not supported (<= %d)", c + 2, GEN_CMAX_WIDE);
  if (n <= c + 3) return fail(GEMMA_HIP_EINVAL, "lmm_set_env: n <= n_cvt + 3");
  if (g_ctx.gxe_env.reserve(n * 8) || g_ctx.gxe_UtWt.reserve((c + 1) * n * 8))
    return fail(GEMMA_HIP_ENOMEM, "lmm_set_env: buffers");
  HIPCHK(hipMemcpy(g_ctx.gxe_env.p, env, n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_ctx.gxe_UtWt.p, g_ctx.UtWt.p, c * n * 8, hipMemcpyDeviceToDevice));
  // U^T env (gsl_blas_dgemv(CblasTrans, U, env), :2308): (n x 1) = U^T (n x n) * env (n x 1)
  HIPCHK(launch_dgemm('T', 'N', (long)n, 1, (long)n, 1.0, g_ctx.U, (long)n, g_ctx.gxe_env.as<double>(), 1, 0.0,
                      g_ctx.gxe_UtWt.as<double>() + c * n, 1, false, false, 0));
  HIPCHK(hipDeviceSynchronize());
  const double df = (double)n - (double)(c + 2) - 1.0;
  g_ctx.gxe_lnbeta = lgamma(df / 2.0) + lgamma(0.5) - lgamma(df / 2.0 + 0.5);
  g_ctx.gxe_ready = true;
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_lmm_gxe_batch_d(int kind, const void *geno, size_t l, size_t ld, gemma_sumstat *out_d,
                                         void *stream) {
  NEED_INIT();
  if (!g_ctx.lmm_active || !g_ctx.gxe_ready) return fail(GEMMA_HIP_ESTATE, "lmm_gxe_batch before lmm_setup + lmm_set_env");
  if (l == 0) return GEMMA_HIP_OK;
  if (kind == GEMMA_GENO_F64_IDV_MAJOR) return fail(GEMMA_HIP_EINVAL, "lmm_gxe_batch: SNP-major input only");
  int rc = check_batch_args("lmm_gxe_batch", kind, geno, l, ld, out_d);
  if (rc) return rc;
  hipStream_t s = S(stream);
  const size_t n = g_ctx.cfg.n, c = g_ctx.cfg.n_cvt;
  const size_t ldx = (n + 1) & ~(size_t)1;
  if (int rcf = xp_flush(s)) return rcf; // blocks of the two-block pipeline still in flight share these buffers
  if (g_ctx.X.reserve(l * ldx * 8) || g_ctx.UtX.reserve(l * ldx * 8) || g_ctx.gxe_Z.reserve(l * ldx * 8) ||
      g_ctx.gxe_UtZ.reserve(l * ldx * 8) || g_ctx.gxe_flip.reserve(l * sizeof(int)))
    return fail(GEMMA_HIP_ENOMEM, "lmm_gxe_batch: cannot allocate 4 x %zu bytes", l * ldx * 8);
  double *X = g_ctx.X.as<double>(), *UtX = g_ctx.UtX.as<double>();
  double *Z = g_ctx.gxe_Z.as<double>(), *UtZ = g_ctx.gxe_UtZ.as<double>();
  {
    ProfScope ps(GEMMA_STAGE_INGEST, s);
    IngestGxeArgs a;
    a.src = geno; a.ld = (long)ld; a.l = (long)l;
    a.idx_map = g_ctx.have_map ? g_ctx.idx_map.as<int>() : nullptr;
    a.n = (int)n; a.env = g_ctx.gxe_env.as<double>(); a.X = X; a.Z = Z; a.ldo = (long)ldx;
    a.flip = g_ctx.gxe_flip.as<int>();
    const unsigned grid = (unsigned)((l + 3) / 4);
    if (kind == GEMMA_GENO_PLINK_2BIT)
      hipLaunchKernelGGL(ingest_gxe_kernel<true>, dim3(grid), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL(ingest_gxe_kernel<false>, dim3(grid), dim3(256), 0, s, a);
    HIPCHK(hipGetLastError());
  }
  {
    ProfScope ps(GEMMA_STAGE_UTX_GEMM, s); // U^T x_s (:2364) and U^T (x_s . env) (:2366); z is real-valued: fp64 GEMMs
    const double *Ug;
    long ldu;
    int rcu = gemm_U(&Ug, &ldu, s);
    if (rcu) return rcu;
    HIPCHK(launch_dgemm('N', 'N', (long)l, (long)n, (long)n, 1.0, X, (long)ldx, Ug, ldu, 0.0, UtX, (long)ldx,
                        false, false, s));
    HIPCHK(launch_dgemm('N', 'N', (long)l, (long)n, (long)n, 1.0, Z, (long)ldx, Ug, ldu, 0.0, UtZ, (long)ldx,
                        false, false, s));
  }
  AssocArgs a = g_ctx.assoc_proto;
  a.UtX = UtX; a.UtZ = UtZ; a.flip = g_ctx.gxe_flip.as<int>();
  a.ld = (long)ldx; a.l = (long)l;
  a.eval = g_ctx.eval; a.Uty = g_ctx.Uty; a.UtWt = g_ctx.gxe_UtWt.as<double>();
  a.out = reinterpret_cast<SumStat *>(out_d);
  a.lnbeta_half_df = g_ctx.gxe_lnbeta; // df = n - (c + 2) - 1
  a.grid_T = nullptr;
  a.have_grid = 0;
  a.have_logdet_ends = g_ctx.assoc_proto.have_logdet_ends;
  const unsigned grid = (unsigned)((l + 3) / 4);
  {
    ProfScope ps(GEMMA_STAGE_ASSOC, s);
    switch (c + 2) {
    case 3: hipLaunchKernelGGL(lmm_gxe_kernel<3>, dim3(grid), dim3(256), 0, s, a); break;
    case 4: hipLaunchKernelGGL(lmm_gxe_kernel<4>, dim3(grid), dim3(256), 0, s, a); break;
    default:
      if (c + 2 > (size_t)GEN_CMAX) {
        int rcw = wide_attr(lmm_gxe_wide_kernel);
        if (rcw) return rcw;
        hipLaunchKernelGGL(lmm_gxe_wide_kernel, dim3((unsigned)l), dim3(64), wide_lds_bytes(c + 2), s, a, (int)(c + 2));
      } else {
        hipLaunchKernelGGL(lmm_gxe_generic_kernel, dim3(grid), dim3(256), 0, s, a, (int)(c + 2));
      }
      break;
    }
    HIPCHK(hipGetLastError());
  }
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_lmm_gxe_batch(int kind, const void *geno, size_t l, size_t ld, gemma_sumstat *out) {
  NEED_INIT();
  if (!g_ctx.lmm_active || !g_ctx.gxe_ready) return fail(GEMMA_HIP_ESTATE, "lmm_gxe_batch before lmm_setup + lmm_set_env");
  if (l == 0) return GEMMA_HIP_OK;
  if (kind == GEMMA_GENO_F64_IDV_MAJOR) return fail(GEMMA_HIP_EINVAL, "lmm_gxe_batch: SNP-major input only");
  int rc = check_batch_args("lmm_gxe_batch", kind, geno, l, ld, out);
  if (rc) return rc;
  const size_t esz = (kind == GEMMA_GENO_PLINK_2BIT) ? 1 : 8;
  if (g_ctx.stage_in.reserve(l * ld * esz) || g_ctx.stage_out.reserve(l * sizeof(gemma_sumstat)))
    return fail(GEMMA_HIP_ENOMEM, "lmm_gxe_batch: staging %zu bytes", l * ld * esz);
  HIPCHK(hipMemcpy(g_ctx.stage_in.p, geno, l * ld * esz, hipMemcpyHostToDevice));
  rc = gemma_hip_lmm_gxe_batch_d(kind, g_ctx.stage_in.p, l, ld, g_ctx.stage_out.as<gemma_sumstat>(), nullptr);
  if (rc) return rc;
  HIPCHK(hipMemcpy(out, g_ctx.stage_out.p, l * sizeof(gemma_sumstat), hipMemcpyDeviceToHost));
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_dbg_utx(int kind, const void *geno, size_t l, size_t ld, int path, double *UtX_host) {
  NEED_INIT();
  if (!g_ctx.lmm_active) return fail(GEMMA_HIP_ESTATE, "dbg_utx before lmm_setup");
  if (l == 0) return GEMMA_HIP_OK;
  int rc = check_batch_args("dbg_utx", kind, geno, l, ld, UtX_host);
  if (rc) return rc;
  const size_t n = g_ctx.cfg.n;
  const size_t rows = (kind == GEMMA_GENO_F64_IDV_MAJOR) ? n : l;
  const size_t esz = (kind == GEMMA_GENO_PLINK_2BIT) ? 1 : 8;
  if (g_ctx.stage_in.reserve(rows * ld * esz)) return fail(GEMMA_HIP_ENOMEM, "dbg_utx: staging");
  HIPCHK(hipMemcpy(g_ctx.stage_in.p, geno, rows * ld * esz, hipMemcpyHostToDevice));
  double *UtX;
  size_t ldx;
  rc = compute_utx(kind, g_ctx.stage_in.p, l, ld, path ? 1 : 0, &UtX, &ldx, 0);
  if (rc) return rc;
  HIPCHK(hipMemcpy2D(UtX_host, n * 8, UtX, ldx * 8, n * 8, l, hipMemcpyDeviceToHost));
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_lmm_batch(int kind, const void *geno, size_t l, size_t ld, gemma_sumstat *out) {
  NEED_INIT();
  if (!g_ctx.lmm_active) return fail(GEMMA_HIP_ESTATE, "lmm_batch before lmm_setup");
  if (l == 0) return GEMMA_HIP_OK;
  const size_t n = g_ctx.cfg.n;
  const size_t per_row = (kind == GEMMA_GENO_PLINK_2BIT && g_ctx.have_map) ? g_ctx.ni_total : n;
  const size_t need = min_ld_for(kind, per_row, l);
  if (need == (size_t)-1) return fail(GEMMA_HIP_EINVAL, "lmm_batch: unknown geno_kind %d", kind);
  if (!geno || !out || ld < need) return fail(GEMMA_HIP_EINVAL, "lmm_batch: ld=%zu < %zu", ld, need);
  const size_t rows = (kind == GEMMA_GENO_F64_IDV_MAJOR) ? n : l;
  const size_t esz = (kind == GEMMA_GENO_PLINK_2BIT) ? 1 : 8;
  if (g_ctx.stage_in.reserve(rows * ld * esz) || g_ctx.stage_out.reserve(l * sizeof(gemma_sumstat)))
    return fail(GEMMA_HIP_ENOMEM, "lmm_batch: staging %zu bytes", rows * ld * esz);
  HIPCHK(hipMemcpy2D(g_ctx.stage_in.p, ld * esz, geno, ld * esz, need * esz, rows, hipMemcpyHostToDevice));
  int rc = gemma_hip_lmm_batch_d(kind, g_ctx.stage_in.p, l, ld, g_ctx.stage_out.as<gemma_sumstat>(), nullptr);
  if (rc) return rc;
  HIPCHK(hipMemcpy(out, g_ctx.stage_out.p, l * sizeof(gemma_sumstat), hipMemcpyDeviceToHost));
  return GEMMA_HIP_OK;
}

// ------------------------------------------------------------------------------ linear model (-lm)
extern "C" int gemma_hip_lm_setup(int a_mode, size_t n, size_t n_cvt, const double *W, const double *y) {
  NEED_INIT();
  g_ctx.knobs.load();
  if (g_ctx.lmm_active) return fail(GEMMA_HIP_ESTATE, "lm_setup while an LMM run is active");
  if (a_mode < 51 || a_mode > 54) return fail(GEMMA_HIP_EINVAL, "lm_setup: a_mode %d (51..54)", a_mode);
  if (!W || !y || n == 0 || n_cvt == 0 || n_cvt > (size_t)LM_CMAX || n <= n_cvt + 1 || n > 0x7fffffffUL)
    return fail(GEMMA_HIP_EINVAL, "lm_setup: bad arguments (n_cvt 1..%d)", LM_CMAX);
  const int c = (int)n_cvt;
  std::vector<double> WtW((size_t)c * c, 0.0), Wt((size_t)c * n), Wty(c, 0.0);
  double yy = 0.0;
  for (size_t i = 0; i < n; ++i) {
    yy += y[i] * y[i];
    for (int a = 0; a < c; ++a) {
      Wt[(size_t)a * n + i] = W[i * c + a];
      Wty[a] += W[i * c + a] * y[i];
      for (int b = 0; b < c; ++b) WtW[(size_t)a * c + b] += W[i * c + a] * W[i * c + b];
    }
  }
  if (!invert_small(WtW, c)) return fail(GEMMA_HIP_EINVAL, "lm_setup: W^T W is singular");
  double d = 0.0; // CalcvPv(WtWi, Wty, y, yPwy), src/lm.cpp:247-263
  for (int a = 0; a < c; ++a) {
    double t = 0.0;
    for (int b = 0; b < c; ++b) t += WtW[(size_t)a * c + b] * Wty[b];
    d += t * Wty[a];
  }
  if (g_ctx.lm_Wt.reserve(Wt.size() * 8) || g_ctx.lm_y.reserve(n * 8) || g_ctx.lm_small.reserve(((size_t)c * c + c) * 8))
    return fail(GEMMA_HIP_ENOMEM, "lm_setup: allocation");
  HIPCHK(hipMemcpy(g_ctx.lm_Wt.p, Wt.data(), Wt.size() * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_ctx.lm_y.p, y, n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_ctx.lm_small.p, WtW.data(), (size_t)c * c * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_ctx.lm_small.as<double>() + (size_t)c * c, Wty.data(), c * 8, hipMemcpyHostToDevice));
  LmArgs &a = g_ctx.lm_proto;
  memset(&a, 0, sizeof a);
  a.Wt = g_ctx.lm_Wt.as<double>();
  a.y = g_ctx.lm_y.as<double>();
  a.WtWi = g_ctx.lm_small.as<double>();
  a.Wty = g_ctx.lm_small.as<double>() + (size_t)c * c;
  a.yPwy = yy - d;
  a.n = (int)n;
  a.c = c;
  a.test_mode = a_mode - 50;
  const double df = (double)n - (double)c - 1.0;
  a.lnbeta_half_df = lgamma(df / 2.0) + lgamma(0.5) - lgamma(df / 2.0 + 0.5);
  g_ctx.cfg.n = n; // shared with the ingest / indicator code
  g_ctx.cfg.n_cvt = n_cvt;
  g_ctx.have_map = false;
  g_ctx.ni_total = 0;
  g_ctx.lm_active = true;
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_lm_batch_d(int kind, const void *geno, size_t l, size_t ld, gemma_sumstat *out_d, void *stream) {
  NEED_INIT();
  if (!g_ctx.lm_active) return fail(GEMMA_HIP_ESTATE, "lm_batch before lm_setup");
  if (l == 0) return GEMMA_HIP_OK;
  const size_t n = g_ctx.cfg.n;
  const size_t per_row = (kind == GEMMA_GENO_PLINK_2BIT && g_ctx.have_map) ? g_ctx.ni_total : n;
  const size_t need = min_ld_for(kind, per_row, l);
  if (need == (size_t)-1) return fail(GEMMA_HIP_EINVAL, "lm_batch: unknown geno_kind %d", kind);
  if (!geno || !out_d || ld < need) return fail(GEMMA_HIP_EINVAL, "lm_batch: ld=%zu < %zu", ld, need);
  hipStream_t s = S(stream);
  const size_t ldx = (n + 1) & ~(size_t)1;
  if (g_ctx.X.reserve(l * ldx * 8)) return fail(GEMMA_HIP_ENOMEM, "lm_batch: cannot allocate %zu bytes", l * ldx * 8);
  double *X = g_ctx.X.as<double>();
  {
    ProfScope ps(GEMMA_STAGE_INGEST, s);
    if (kind == GEMMA_GENO_F64_IDV_MAJOR) {
      dim3 grid((unsigned)((l + 31) / 32), (unsigned)((n + 31) / 32));
      hipLaunchKernelGGL(transpose_kernel, grid, dim3(32, 8), 0, s, reinterpret_cast<const double *>(geno), (long)n,
                         (long)l, (long)ld, X, (long)ldx);
    } else {
      IngestArgs a;
      a.src = geno; a.ld = (long)ld; a.l = (long)l;
      a.idx_map = g_ctx.have_map ? g_ctx.idx_map.as<int>() : nullptr;
      a.n = (int)n; a.dst = X; a.ldo = (long)ldx; a.k_mode = 0;
      const unsigned grid = (unsigned)((l + 3) / 4);
      if (kind == GEMMA_GENO_PLINK_2BIT)
        hipLaunchKernelGGL(ingest_lmm_kernel<true>, dim3(grid), dim3(256), 0, s, a);
      else
        hipLaunchKernelGGL(ingest_lmm_kernel<false>, dim3(grid), dim3(256), 0, s, a);
    }
    HIPCHK(hipGetLastError());
  }
  LmArgs a = g_ctx.lm_proto;
  a.X = X; a.ld = (long)ldx; a.l = (long)l;
  a.out = reinterpret_cast<SumStat *>(out_d);
  {
    ProfScope ps(GEMMA_STAGE_ASSOC, s);
    hipLaunchKernelGGL(lm_assoc_kernel, dim3((unsigned)((l + 3) / 4)), dim3(256), 0, s, a);
    HIPCHK(hipGetLastError());
  }
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_lm_batch(int kind, const void *geno, size_t l, size_t ld, gemma_sumstat *out) {
  NEED_INIT();
  if (!g_ctx.lm_active) return fail(GEMMA_HIP_ESTATE, "lm_batch before lm_setup");
  if (l == 0) return GEMMA_HIP_OK;
  const size_t n = g_ctx.cfg.n;
  const size_t per_row = (kind == GEMMA_GENO_PLINK_2BIT && g_ctx.have_map) ? g_ctx.ni_total : n;
  const size_t need = min_ld_for(kind, per_row, l);
  if (need == (size_t)-1) return fail(GEMMA_HIP_EINVAL, "lm_batch: unknown geno_kind %d", kind);
  if (!geno || !out || ld < need) return fail(GEMMA_HIP_EINVAL, "lm_batch: ld=%zu < %zu", ld, need);
  const size_t rows = (kind == GEMMA_GENO_F64_IDV_MAJOR) ? n : l;
  const size_t esz = (kind == GEMMA_GENO_PLINK_2BIT) ? 1 : 8;
  if (g_ctx.stage_in.reserve(rows * ld * esz) || g_ctx.stage_out.reserve(l * sizeof(gemma_sumstat)))
    return fail(GEMMA_HIP_ENOMEM, "lm_batch: staging %zu bytes", rows * ld * esz);
  HIPCHK(hipMemcpy2D(g_ctx.stage_in.p, ld * esz, geno, ld * esz, need * esz, rows, hipMemcpyHostToDevice));
  int rc = gemma_hip_lm_batch_d(kind, g_ctx.stage_in.p, l, ld, g_ctx.stage_out.as<gemma_sumstat>(), nullptr);
  if (rc) return rc;
  HIPCHK(hipMemcpy(out, g_ctx.stage_out.p, l * sizeof(gemma_sumstat), hipMemcpyDeviceToHost));
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_lm_finish(void) {
  NEED_INIT();
  if (!g_ctx.lm_active) return fail(GEMMA_HIP_ESTATE, "lm_finish before lm_setup");
  HIPCHK(hipDeviceSynchronize());
  g_ctx.lm_Wt.release(); g_ctx.lm_y.release(); g_ctx.lm_small.release();
  g_ctx.X.release(); g_ctx.stage_in.release(); g_ctx.stage_out.release(); g_ctx.idx_map.release();
  g_ctx.lm_active = false;
  return GEMMA_HIP_OK;
}

// Null model on device.  out[8] = { l_mle_null, logl_mle_H0, l_remle_null, logl_remle_H0,
// pve, pve_se, vg_remle, ve_remle } -- the quantities src/gemma.cpp:2711-2750 derives before
// the per-SNP loop (CalcLambda 'L'/'R' with calc_null, CalcPve src/lmm.cpp:2183-2205, and the
// vg/ve part of CalcLmmVgVeBeta :2253-2259).
extern "C" int gemma_hip_lmm_null(size_t n, size_t n_cvt, const double *eval, const double *UtW,
                                  const double *Uty, double l_min, double l_max, size_t n_region,
                                  double trace_G, double *out8) {
  NEED_INIT();
  if (!eval || !UtW || !Uty || !out8 || n == 0 || n_cvt == 0 || n_cvt > (size_t)GEN_CMAX_WIDE + 1)
    return fail(GEMMA_HIP_EINVAL, "lmm_null: bad arguments (n_cvt 1..%d)", GEN_CMAX_WIDE + 1);
  if (!(l_max > l_min) || n_region == 0 || n_region > (size_t)ASSOC_MAX_REGION || n <= n_cvt)
    return fail(GEMMA_HIP_EINVAL, "lmm_null: l_min/l_max/n_region/n");
  DevBuf dE, dW, dWt, dY, dO;
  auto cleanup = [&]() { dE.release(); dW.release(); dWt.release(); dY.release(); dO.release(); };
  if (dE.reserve(n * 8) || dW.reserve(n * n_cvt * 8) || dWt.reserve(n * n_cvt * 8) || dY.reserve(n * 8) ||
      dO.reserve(sizeof(NullOut))) {
    cleanup();
    return fail(GEMMA_HIP_ENOMEM, "lmm_null: allocation");
  }
  hipError_t e = hipMemcpy(dE.p, eval, n * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dW.p, UtW, n * n_cvt * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dY.p, Uty, n * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    dim3 grid((unsigned)((n_cvt + 31) / 32), (unsigned)((n + 31) / 32));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(32, 8), 0, 0, dW.as<double>(), (long)n, (long)n_cvt,
                       (long)n_cvt, dWt.as<double>(), (long)n);
    AssocArgs a;
    memset(&a, 0, sizeof a);
    a.n = (int)n; a.n_region = (int)n_region; a.l_min = l_min; a.l_max = l_max;
    a.eval = dE.as<double>(); a.Uty = dY.as<double>(); a.UtWt = dWt.as<double>();
    const double lambda_interval = log(l_max / l_min) / (double)n_region;
    for (size_t i = 0; i <= n_region; ++i) a.lam_grid[i] = l_min * exp(lambda_interval * (double)i);
    NullOut *o = dO.as<NullOut>();
    switch (n_cvt) {
    case 1: hipLaunchKernelGGL(lmm_null_kernel<0>, dim3(1), dim3(64), 0, 0, a, o); break;
    case 2: hipLaunchKernelGGL(lmm_null_kernel<1>, dim3(1), dim3(64), 0, 0, a, o); break;
    case 3: hipLaunchKernelGGL(lmm_null_kernel<2>, dim3(1), dim3(64), 0, 0, a, o); break;
    case 4: hipLaunchKernelGGL(lmm_null_kernel<3>, dim3(1), dim3(64), 0, 0, a, o); break;
    case 5: hipLaunchKernelGGL(lmm_null_kernel<4>, dim3(1), dim3(64), 0, 0, a, o); break;
    default:
      if (n_cvt - 1 > (size_t)GEN_CMAX) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(lmm_null_wide_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)wide_lds_bytes(GEN_CMAX_WIDE));
        hipLaunchKernelGGL(lmm_null_wide_kernel, dim3(1), dim3(64), wide_lds_bytes(n_cvt - 1), 0, a, (int)n_cvt - 1, o);
      } else {
        hipLaunchKernelGGL(lmm_null_generic_kernel, dim3(1), dim3(64), 0, 0, a, (int)n_cvt - 1, o);
      }
      break;
    }
    e = hipGetLastError();
  }
  NullOut h;
  if (e == hipSuccess) e = hipMemcpy(&h, dO.p, sizeof h, hipMemcpyDeviceToHost);
  cleanup();
  if (e != hipSuccess) return fail(GEMMA_HIP_ERUNTIME, "lmm_null: %s", hipGetErrorString(e));
  out8[0] = h.l_mle; out8[1] = h.logl_mle; out8[2] = h.l_remle; out8[3] = h.logl_remle;
  // CalcPve, src/lmm.cpp:2197-2200 (safe_sqrt semantics of src/mathfunc.cpp:122-131)
  double arg = -1.0 / h.dev2_remle, d1 = arg;
  if (arg < 0.001) d1 = fabs(arg);
  const double se = (d1 < 0.0) ? NAN : sqrt(d1);
  out8[4] = trace_G * h.l_remle / (trace_G * h.l_remle + 1.0);
  out8[5] = trace_G / ((trace_G * h.l_remle + 1.0) * (trace_G * h.l_remle + 1.0)) * se;
  out8[7] = h.Pyy_remle / (double)(n - n_cvt); // ve, src/lmm.cpp:2258
  out8[6] = out8[7] * h.l_remle;               // vg
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_lmm_finish(double *time_UtX_min, double *time_opt_min) {
  NEED_INIT();
  if (!g_ctx.lmm_active) return fail(GEMMA_HIP_ESTATE, "lmm_finish before lmm_setup");
  HIPCHK(hipDeviceSynchronize());
  pipe_release(); // blocks still in flight are dropped with the state
  xp_release();
  prof_collect(GEMMA_STAGE_UTX_GEMM);
  prof_collect(GEMMA_STAGE_UTX_POST);
  prof_collect(GEMMA_STAGE_ASSOC);
  if (time_UtX_min)
    *time_UtX_min = (g_ctx.prof[GEMMA_STAGE_UTX_GEMM].acc_ms + g_ctx.prof[GEMMA_STAGE_UTX_POST].acc_ms) / 60000.0;
  if (time_opt_min) *time_opt_min = g_ctx.prof[GEMMA_STAGE_ASSOC].acc_ms / 60000.0;
  g_ctx.own_U.release(); g_ctx.own_eval.release(); g_ctx.own_Uty.release(); g_ctx.own_UtW.release();
  g_ctx.UtWt.release(); g_ctx.idx_map.release(); g_ctx.X.release(); g_ctx.UtX.release();
  g_ctx.stage_in.release(); g_ctx.stage_out.release();
  g_ctx.grid_R.release(); g_ctx.grid_F.release(); g_ctx.grid_T.release();
  g_ctx.table_P.release();
  g_ctx.cheb_R.release(); g_ctx.cheb_F.release(); g_ctx.cheb_T.release(); g_ctx.cheb_slots.release();
  g_ctx.cheb_list.release(); g_ctx.cheb_count.release(); g_ctx.cheb_D.release(); g_ctx.cheb_Ck.release();
  g_ctx.cheb_Gk.release(); g_ctx.cheb_Lk.release(); g_ctx.cheb_iv.release(); g_ctx.cheb_dends.release(); g_ctx.cheb_res.release();
  g_ctx.i8_Bt.release(); g_ctx.i8_q.release(); g_ctx.i8_qinv.release(); g_ctx.i8_cmax.release(); g_ctx.i8_A.release(); g_ctx.i8_C.release();
  raster_release();
  g_ctx.i8_mean.release(); g_ctx.i8_meta.release(); g_ctx.i8_rowsur.release(); g_ctx.i8_colsum.release(); g_ctx.i8_surlist.release();
  g_ctx.i8_ready = false;
  g_ctx.i8_colsum_ready = false;
  g_ctx.gxe_env.release(); g_ctx.gxe_UtWt.release(); g_ctx.gxe_Z.release(); g_ctx.gxe_UtZ.release();
  g_ctx.mv_Yt.release(); g_ctx.mv_out.release(); g_ctx.mv_scratch.release();
  g_ctx.mv_ready = g_ctx.mv_gxe = false;
  g_ctx.gxe_flip.release();
  g_ctx.gxe_ready = false;
  g_ctx.U = g_ctx.eval = g_ctx.Uty = nullptr;
  g_ctx.U_even_of = nullptr;
  g_ctx.U_even.release();
  g_ctx.lmm_active = false;
  return GEMMA_HIP_OK;
}
