// Part of gemma_hip.hip (ONE translation unit: the parts share the context g_ctx and the helpers of its anonymous namespace, and are
// included there in this order; round 6: the 3 500-line file cut along its stages for reading -- no behaviour change).
// This part: multivariate LMM ABI (kernels in mvlmm_kernels*.hip).

// ---- multivariate LMM: MVLMM::AnalyzeBimbam / AnalyzePlink, src/mvlmm.cpp:2972-3899 (kernels in mvlmm_kernels.hip)
extern "C" int gemma_hip_mvlmm_launch_(const MvArgs *g, int d, int c, hipStream_t s);
extern "C" int gemma_hip_mvlmm_null_launch_(const MvNullArgs *a, int d, int c, hipStream_t s);
// mvlmm_kernels_rt.hip: the run-time (d, c) kernel
extern "C" size_t gemma_hip_mvlmm_rt_scratch_(int d, int c);
extern "C" int gemma_hip_mvlmm_launch_rt_(const MvArgs *g, unsigned grid, hipStream_t s);
extern "C" int gemma_hip_mvlmm_null_launch_rt_(const MvNullArgs *a, hipStream_t s);

// c = covariates of the model the caller names; extra = the rows of X on top of them (1: the SNP; 3: env, SNP, interaction)
static int mv_check_dims(const char *who, size_t d, size_t c, size_t extra = 1) {
  if (d < 1 || d > (size_t)MV_DMAX) return fail(GEMMA_HIP_EINVAL, "%s: %zu phenotypes not supported (1..%d)", who, d, MV_DMAX);
  // fixed kernels: d <= 5 with up to 3 covariates, d <= 3 with up to 6 (mvlmm_kernels*.hip); everything else up to MV_DMAX phenotypes
  // and MV_CMAX rows of X runs on the run-time kernel (mvlmm_kernels_rt.hip)
  const size_t cmax = (size_t)MV_CMAX - extra;
  if (c < 1 || c > cmax)
    return fail(GEMMA_HIP_EINVAL, "%s: %zu covariates not supported (1..%zu)", who, c, cmax);
  return GEMMA_HIP_OK;
}
// GEMMA_HIP_MVLMM_RT=1: the run-time kernel also where a fixed one exists (tests)
static bool mv_force_rt() { return g_ctx.knobs.mvlmm_rt != 0; }

static void mv_default_opt(gemma_mvlmm_opt &o, const gemma_mvlmm_opt *opt) {
  if (opt) {
    o = *opt;
    return;
  }
  o.em_iter = 10000;
  o.nr_iter = 100;
  o.em_prec = 1e-4;
  o.nr_prec = 1e-4;
  o.p_nr = 1e-3;
  o.crt = 0;
  o.gxe = 0;
}

// rows x cols (row-major, host) -> cols x rows on the device
static int mv_upload_transposed(const double *src, size_t rows, size_t cols, DevBuf &dst) {
  std::vector<double> t(rows * cols);
  for (size_t i = 0; i < rows; ++i)
    for (size_t j = 0; j < cols; ++j) t[j * rows + i] = src[i * cols + j];
  if (dst.reserve(rows * cols * 8)) return fail(GEMMA_HIP_ENOMEM, "mvlmm: %zu bytes", rows * cols * 8);
  HIPCHK(hipMemcpy(dst.p, t.data(), rows * cols * 8, hipMemcpyHostToDevice));
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_mvlmm_null(size_t n, size_t n_cvt, size_t d, const double *eval, const double *UtW,
                                    const double *UtY, double l_min, double l_max, size_t n_region,
                                    const gemma_mvlmm_opt *opt, gemma_mvlmm_null *out) {
  NEED_INIT();
  g_ctx.knobs.load();
  if (!eval || !UtW || !UtY || !out) return fail(GEMMA_HIP_EINVAL, "mvlmm_null: null pointer");
  int rc = mv_check_dims("mvlmm_null", d, n_cvt);
  if (rc) return rc;
  if (n <= n_cvt + 1) return fail(GEMMA_HIP_EINVAL, "mvlmm_null: n <= n_cvt + 1");
  gemma_mvlmm_opt o;
  mv_default_opt(o, opt);
  const size_t c = n_cvt;
  DevBuf d_eval, d_Wt, d_Yt, d_Ypair, d_out;
  struct Rel {
    DevBuf *b[5];
    ~Rel() { for (DevBuf *x : b) x->release(); }
  } rel{{&d_eval, &d_Wt, &d_Yt, &d_Ypair, &d_out}};
  constexpr size_t RES_MAX = 2 * (2 * MV_DMAX * MV_DMAX + MV_BMAX + 1);
  DevBuf d_scr;
  struct Rel2 {
    DevBuf *b;
    ~Rel2() { b->release(); }
  } rel2{&d_scr};
  if (d_eval.reserve(n * 8) || d_out.reserve(RES_MAX * 8)) return fail(GEMMA_HIP_ENOMEM, "mvlmm_null: buffers");
  HIPCHK(hipMemcpy(d_eval.p, eval, n * 8, hipMemcpyHostToDevice));
  if ((rc = mv_upload_transposed(UtW, n, c, d_Wt)) || (rc = mv_upload_transposed(UtY, n, d, d_Yt))) return rc;
  // MphInitial :2780-2797: the diagonals from one univariate REML fit per trait
  std::vector<double> Vg0(d * d, 0.0), Ve0(d * d, 0.0), ycol(n);
  for (size_t i = 0; i < d; ++i) {
    for (size_t k = 0; k < n; ++k) ycol[k] = UtY[k * d + i];
    double o8[8];
    rc = gemma_hip_lmm_null(n, c, eval, UtW, ycol.data(), l_min, l_max, n_region, 1.0, o8);
    if (rc) return rc;
    Vg0[i * d + i] = o8[6];
    Ve0[i * d + i] = o8[7];
  }
  auto run_fit = [&](size_t dd, const double *Yt_dev, const double *vg0, const double *ve0, double *host_out) -> int {
    MvNullArgs a;
    memset(&a, 0, sizeof a);
    a.g.n = (int)n;
    a.g.eval = d_eval.as<double>();
    a.g.Wt = d_Wt.as<double>();
    a.g.Yt = Yt_dev;
    a.g.nr_iter = (int)o.nr_iter;
    a.g.nr_prec = o.nr_prec;
    a.em_iter = (int)o.em_iter;
    a.em_prec = o.em_prec;
    for (size_t i = 0; i < dd * dd; ++i) {
      a.Vg0[i] = vg0[i];
      a.Ve0[i] = ve0[i];
    }
    a.out = d_out.as<double>();
    int lrc = mv_force_rt() ? -1 : gemma_hip_mvlmm_null_launch_(&a, (int)dd, (int)c, 0);
    if (lrc < 0) { // no fixed kernel for this shape
      a.g.d = (int)dd;
      a.g.c = (int)c;
      if (d_scr.reserve(gemma_hip_mvlmm_rt_scratch_((int)dd, (int)c) * 8)) return fail(GEMMA_HIP_ENOMEM, "mvlmm_null: scratch");
      a.g.scratch = d_scr.as<double>();
      lrc = gemma_hip_mvlmm_null_launch_rt_(&a, 0);
    }
    if (lrc) return fail(GEMMA_HIP_ERUNTIME, "mvlmm_null launch: %s", hipGetErrorString((hipError_t)lrc));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(host_out, d_out.p, 2 * (2 * dd * dd + dd * c + 1) * 8, hipMemcpyDeviceToHost));
    return GEMMA_HIP_OK;
  };
  std::vector<double> res(RES_MAX);
  if (d > 4) { // :2805-2884: off-diagonals from two-trait REML fits
    if (d_Ypair.reserve(2 * n * 8)) return fail(GEMMA_HIP_ENOMEM, "mvlmm_null: pair buffer");
    for (size_t i = 0; i < d; ++i)
      for (size_t j = i + 1; j < d; ++j) {
        HIPCHK(hipMemcpy(d_Ypair.p, d_Yt.as<double>() + i * n, n * 8, hipMemcpyDeviceToDevice));
        HIPCHK(hipMemcpy(d_Ypair.as<double>() + n, d_Yt.as<double>() + j * n, n * 8, hipMemcpyDeviceToDevice));
        const double vg2[4] = {Vg0[i * d + i], 0, 0, Vg0[j * d + j]}, ve2[4] = {Ve0[i * d + i], 0, 0, Ve0[j * d + j]};
        if ((rc = run_fit(2, d_Ypair.as<double>(), vg2, ve2, res.data()))) return rc;
        Vg0[i * d + j] = Vg0[j * d + i] = res[1];     // Vg_sub(0, 1) of the REMLE block
        Ve0[i * d + j] = Ve0[j * d + i] = res[4 + 1]; // Ve_sub(0, 1)
      }
  }
  if ((rc = run_fit(d, d_Yt.as<double>(), Vg0.data(), Ve0.data(), res.data()))) return rc;
  memset(out, 0, sizeof *out);
  const size_t blk = 2 * d * d + d * c + 1;
  for (size_t i = 0; i < d * d; ++i) {
    out->Vg_remle[i] = res[i];
    out->Ve_remle[i] = res[d * d + i];
    out->Vg_mle[i] = res[blk + i];
    out->Ve_mle[i] = res[blk + d * d + i];
  }
  for (size_t i = 0; i < d * c; ++i) {
    out->B_remle[i] = res[2 * d * d + i];
    out->B_mle[i] = res[blk + 2 * d * d + i];
  }
  out->logl_remle_H0 = res[2 * d * d + d * c];
  out->logl_mle_H0 = res[blk + 2 * d * d + d * c];
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_mvlmm_set(size_t d, const double *UtY, const gemma_mvlmm_null *nf, const gemma_mvlmm_opt *opt) {
  NEED_INIT();
  g_ctx.knobs.load();
  if (!g_ctx.lmm_active) return fail(GEMMA_HIP_ESTATE, "mvlmm_set before lmm_setup");
  if (!UtY || !nf) return fail(GEMMA_HIP_EINVAL, "mvlmm_set: null pointer");
  gemma_mvlmm_opt o;
  mv_default_opt(o, opt);
  const bool gxe = o.gxe == 1;
  if (gxe && !g_ctx.gxe_ready) return fail(GEMMA_HIP_ESTATE, "mvlmm_set with gxe before lmm_set_env");
  // gxe: the null fit is the one of (W, env) -- c covariates here; the per-SNP models add the SNP and its interaction row
  const size_t n = g_ctx.cfg.n, c = g_ctx.cfg.n_cvt + (gxe ? 1 : 0);
  int rc = mv_check_dims("mvlmm_set", d, c, gxe ? 2 : 1);
  if (rc) return rc;
  if (g_ctx.cfg.a_mode < 1 || g_ctx.cfg.a_mode > 4) return fail(GEMMA_HIP_EINVAL, "mvlmm_set: a_mode %d (1..4)", g_ctx.cfg.a_mode);
  if ((rc = mv_upload_transposed(UtY, n, d, g_ctx.mv_Yt))) return rc;
  MvArgs &a = g_ctx.mv_proto;
  memset(&a, 0, sizeof a);
  a.n = (int)n;
  for (size_t i = 0; i < d * d; ++i) {
    a.Vg_null[i] = nf->Vg_mle[i];
    a.Ve_null[i] = nf->Ve_mle[i];
  }
  for (size_t i = 0; i < d * c; ++i) a.B_null[i] = nf->B_mle[i];
  a.logl_H0 = nf->logl_mle_H0;
  a.a_mode = g_ctx.cfg.a_mode;
  a.em_iter = (int)(o.em_iter / 10);   // src/mvlmm.cpp:3310,3336
  a.em_prec = o.em_prec * 10;
  a.nr_iter = (int)(o.nr_iter / 10);   // :3321,3344
  a.nr_prec = o.nr_prec * 10;
  a.p_nr = o.p_nr;
  a.crt = o.crt == 1 ? 1 : 0;       // :3302,3329,3349 test crt == 1
  a.stride = (int)(d + 3 * (d * (d + 1) / 2) + 3);
  g_ctx.mv_d = d;
  g_ctx.mv_gxe = gxe;
  g_ctx.mv_ready = true;
  return GEMMA_HIP_OK;
}

// launches the per-SNP kernel: the fixed instance of (d, rows) if there is one, else the run-time kernel
static int mv_launch(MvArgs &a, size_t d, size_t rows, hipStream_t s) {
  int lrc = (a.UtX2 || mv_force_rt()) ? -1 : gemma_hip_mvlmm_launch_(&a, (int)d, (int)rows, s);
  if (lrc < 0) {
    a.d = (int)d;
    a.c = (int)rows;
    const unsigned grid = (unsigned)std::min<size_t>((size_t)a.l, 1024);
    const size_t per = gemma_hip_mvlmm_rt_scratch_(a.d, a.c);
    if (g_ctx.mv_scratch.reserve((size_t)grid * per * 8)) return fail(GEMMA_HIP_ENOMEM, "mvlmm_batch: %zu bytes of scratch", (size_t)grid * per * 8);
    a.scratch = g_ctx.mv_scratch.as<double>();
    lrc = gemma_hip_mvlmm_launch_rt_(&a, grid, s);
  }
  if (lrc) return fail(GEMMA_HIP_ERUNTIME, "mvlmm_batch launch: %s", hipGetErrorString((hipError_t)lrc));
  return GEMMA_HIP_OK;
}

// MVLMM::AnalyzeBimbamGXE / AnalyzePlinkGXE (src/mvlmm.cpp:3970-4414 / :4416-4870): x, x o env and the allele flip as in the
// univariate GXE path (ingest_gxe_kernel), both rotated by fp64 GEMMs
static int mvlmm_gxe_batch_d(int kind, const void *geno, size_t l, size_t ld, double *out_d, hipStream_t s) {
  if (kind == GEMMA_GENO_F64_IDV_MAJOR) return fail(GEMMA_HIP_EINVAL, "mvlmm_batch (gxe): SNP-major input only");
  const size_t n = g_ctx.cfg.n, c = g_ctx.cfg.n_cvt;
  const size_t ldx = (n + 1) & ~(size_t)1;
  if (int rcf = xp_flush(s)) return rcf; // blocks of the two-block pipeline still in flight share these buffers
  if (g_ctx.X.reserve(l * ldx * 8) || g_ctx.UtX.reserve(l * ldx * 8) || g_ctx.gxe_Z.reserve(l * ldx * 8) ||
      g_ctx.gxe_UtZ.reserve(l * ldx * 8) || g_ctx.gxe_flip.reserve(l * sizeof(int)))
    return fail(GEMMA_HIP_ENOMEM, "mvlmm_batch (gxe): cannot allocate 4 x %zu bytes", l * ldx * 8);
  double *X = g_ctx.X.as<double>(), *UtX = g_ctx.UtX.as<double>();
  double *Z = g_ctx.gxe_Z.as<double>(), *UtZ = g_ctx.gxe_UtZ.as<double>();
  {
    ProfScope ps(GEMMA_STAGE_INGEST, s);
    IngestGxeArgs ia;
    ia.src = geno; ia.ld = (long)ld; ia.l = (long)l;
    ia.idx_map = g_ctx.have_map ? g_ctx.idx_map.as<int>() : nullptr;
    ia.n = (int)n; ia.env = g_ctx.gxe_env.as<double>(); ia.X = X; ia.Z = Z; ia.ldo = (long)ldx;
    ia.flip = g_ctx.gxe_flip.as<int>();
    const unsigned grid = (unsigned)((l + 3) / 4);
    if (kind == GEMMA_GENO_PLINK_2BIT)
      hipLaunchKernelGGL(ingest_gxe_kernel<true>, dim3(grid), dim3(256), 0, s, ia);
    else
      hipLaunchKernelGGL(ingest_gxe_kernel<false>, dim3(grid), dim3(256), 0, s, ia);
    HIPCHK(hipGetLastError());
  }
  {
    ProfScope ps(GEMMA_STAGE_UTX_GEMM, s);
    const double *Ug;
    long ldu;
    int rcu = gemm_U(&Ug, &ldu, s);
    if (rcu) return rcu;
    HIPCHK(launch_dgemm('N', 'N', (long)l, (long)n, (long)n, 1.0, X, (long)ldx, Ug, ldu, 0.0, UtX, (long)ldx, false, false, s));
    HIPCHK(launch_dgemm('N', 'N', (long)l, (long)n, (long)n, 1.0, Z, (long)ldx, Ug, ldu, 0.0, UtZ, (long)ldx, false, false, s));
  }
  MvArgs a = g_ctx.mv_proto;
  a.UtX = UtX;
  a.UtX2 = UtZ;
  a.flip = g_ctx.gxe_flip.as<int>();
  a.ld = (long)ldx;
  a.l = (long)l;
  a.eval = g_ctx.eval;
  a.Wt = g_ctx.gxe_UtWt.as<double>(); // W then U^T env
  a.Yt = g_ctx.mv_Yt.as<double>();
  a.out = out_d;
  ProfScope ps(GEMMA_STAGE_ASSOC, s);
  return mv_launch(a, g_ctx.mv_d, c + 3, s);
}

extern "C" int gemma_hip_mvlmm_batch_d(int kind, const void *geno, size_t l, size_t ld, double *out_d, void *stream) {
  NEED_INIT();
  if (!g_ctx.lmm_active || !g_ctx.mv_ready) return fail(GEMMA_HIP_ESTATE, "mvlmm_batch before lmm_setup + mvlmm_set");
  if (l == 0) return GEMMA_HIP_OK;
  int rc = check_batch_args("mvlmm_batch", kind, geno, l, ld, out_d);
  if (rc) return rc;
  hipStream_t s = S(stream);
  if (g_ctx.mv_gxe) return mvlmm_gxe_batch_d(kind, geno, l, ld, out_d, s);
  double *UtX;
  size_t ldx;
  rc = compute_utx(kind, geno, l, ld, -1, &UtX, &ldx, s);
  if (rc) return rc;
  MvArgs a = g_ctx.mv_proto;
  a.UtX = UtX;
  a.ld = (long)ldx;
  a.l = (long)l;
  a.eval = g_ctx.eval;
  a.Wt = g_ctx.UtWt.as<double>();
  a.Yt = g_ctx.mv_Yt.as<double>();
  a.out = out_d;
  ProfScope ps(GEMMA_STAGE_ASSOC, s);
  return mv_launch(a, g_ctx.mv_d, g_ctx.cfg.n_cvt + 1, s);
}

extern "C" int gemma_hip_mvlmm_batch(int kind, const void *geno, size_t l, size_t ld, double *out) {
  NEED_INIT();
  if (!g_ctx.lmm_active || !g_ctx.mv_ready) return fail(GEMMA_HIP_ESTATE, "mvlmm_batch before lmm_setup + mvlmm_set");
  if (l == 0) return GEMMA_HIP_OK;
  int rc = check_batch_args("mvlmm_batch", kind, geno, l, ld, out);
  if (rc) return rc;
  const size_t n = g_ctx.cfg.n;
  const size_t per_row = (kind == GEMMA_GENO_PLINK_2BIT && g_ctx.have_map) ? g_ctx.ni_total : n;
  const size_t need = min_ld_for(kind, per_row, l);
  const size_t rows = (kind == GEMMA_GENO_F64_IDV_MAJOR) ? n : l;
  const size_t esz = (kind == GEMMA_GENO_PLINK_2BIT) ? 1 : 8;
  const size_t bytes_out = l * (size_t)g_ctx.mv_proto.stride * 8;
  if (g_ctx.stage_in.reserve(rows * ld * esz) || g_ctx.mv_out.reserve(bytes_out))
    return fail(GEMMA_HIP_ENOMEM, "mvlmm_batch: staging %zu bytes", rows * ld * esz + bytes_out);
  HIPCHK(hipMemcpy2D(g_ctx.stage_in.p, ld * esz, geno, ld * esz, need * esz, rows, hipMemcpyHostToDevice));
  rc = gemma_hip_mvlmm_batch_d(kind, g_ctx.stage_in.p, l, ld, g_ctx.mv_out.as<double>(), nullptr);
  if (rc) return rc;
  HIPCHK(hipMemcpy(out, g_ctx.mv_out.p, bytes_out, hipMemcpyDeviceToHost));
  return GEMMA_HIP_OK;
}
