// Part of gemma_hip.hip (ONE translation unit: the parts share the context g_ctx and the helpers of its anonymous namespace, and are
// included there in this order; round 6: the 3 500-line file cut along its stages for reading -- no behaviour change).
// This part: the per-SNP stage of LMM::Analyze: setup (lmm_setup*), fixed-lambda tables, Chebyshev series, launch_assoc.

// ------------------------------------------------------------------------------ LMM
static int lmm_common_setup(const gemma_lmm_cfg *cfg) {
  if (!cfg) return fail(GEMMA_HIP_EINVAL, "lmm_setup: null cfg");
  if (cfg->n == 0 || cfg->n_cvt == 0) return fail(GEMMA_HIP_EINVAL, "lmm_setup: n=%zu n_cvt=%zu", cfg->n, cfg->n_cvt);
  if (cfg->n_cvt > (size_t)GEN_CMAX_WIDE)
    return fail(GEMMA_HIP_EINVAL, "lmm_setup: n_cvt=%zu not supported by this build (1..%d)", cfg->n_cvt, GEN_CMAX_WIDE);
  if (!(cfg->a_mode == 1 || cfg->a_mode == 2 || cfg->a_mode == 3 || cfg->a_mode == 4 || cfg->a_mode == 9))
    return fail(GEMMA_HIP_EINVAL, "lmm_setup: a_mode %d", cfg->a_mode);
  if (!(cfg->l_max > cfg->l_min) || cfg->n_region == 0 || cfg->n_region > (size_t)ASSOC_MAX_REGION)
    return fail(GEMMA_HIP_EINVAL, "lmm_setup: l_min/l_max/n_region");
  if (cfg->n <= cfg->n_cvt + 1) return fail(GEMMA_HIP_EINVAL, "lmm_setup: n <= n_cvt + 1");
  if (cfg->n > 0x7fffffffUL) return fail(GEMMA_HIP_EINVAL, "lmm_setup: n too large");
  g_ctx.cfg = *cfg;
  {
    // the eigensolver's workspace pool (gemma_hip_eigh_reserve / GEMMA_HIP_EIGH_CACHE): a pool that holds more than a quarter of
    // the device would stand in the way of this setup's own buffers (n = 50 000: 100+ GB idle beside 70 GB of digit planes)
    size_t mf = 0, mt = 0;
    if (eigh_pool_idle_bytes_x() > 0 && hipMemGetInfo(&mf, &mt) == hipSuccess && eigh_pool_idle_bytes_x() > mt / 4) (void)eigh_release_x();
  }
  g_ctx.knobs.load(); // the environment switches of the batch path: once per setup
  AssocArgs &a = g_ctx.assoc_proto;
  memset(&a, 0, sizeof a);
  a.n = (int)cfg->n;
  a.a_mode = cfg->a_mode;
  a.n_region = (int)cfg->n_region;
  a.plink_nan_rule = cfg->plink_nan_rule;
  a.l_min = cfg->l_min;
  a.l_max = cfg->l_max;
  a.l_mle_null = cfg->l_mle_null;
  a.logl_mle_H0 = cfg->logl_mle_H0;
  const double df = (double)cfg->n - (double)cfg->n_cvt - 1.0;
  a.lnbeta_half_df = lgamma(df / 2.0) + lgamma(0.5) - lgamma(df / 2.0 + 0.5);
  // lambda grid exactly as src/lmm.cpp:1964-1969
  const double lambda_interval = log(cfg->l_max / cfg->l_min) / (double)cfg->n_region;
  for (size_t i = 0; i <= cfg->n_region; ++i) a.lam_grid[i] = cfg->l_min * exp(lambda_interval * (double)i);
  if (g_ctx.carry.reserve(4 * 8)) return fail(GEMMA_HIP_ENOMEM, "lmm_setup: carry");
  HIPCHK(hipMemset(g_ctx.carry.p, 0, 4 * 8));
  g_ctx.carry_flip = 0;
  g_ctx.have_map = false;
  g_ctx.ni_total = 0;
  g_ctx.i8_ready = false; // digits belong to the previous U
  g_ctx.i8_colsum_ready = false;
  g_ctx.gxe_ready = false;
  g_ctx.mv_ready = false;
  return GEMMA_HIP_OK;
}

// Fixed-lambda table, SNP-independent part (lmm_grid.hip.h): weight matrix in MFMA operand order and the sums over
// the covariate / phenotype pairs.  Built for the register kernels (c <= 4) and the default n_region = 10 (23
// weights); anything else keeps streaming every evaluation.  GEMMA_HIP_ASSOC_GRID=0 switches the table off.
static bool grid_blocks(size_t c, int nq, int *nbx, int *nba) {
  *nbx = (nq + 15) / 16;
  *nba = ((int)(c + 1) * nq + 15) / 16;
  return c >= 1 && c <= 4 && nq == 23;
}
// Chebyshev-in-log(lambda) series of the bracket intervals (lmm_search.hip.h), SNP-independent part: per interval
// [lam_grid[j], lam_grid[j + 1]] with lam_grid[j] >= CHEB_MIN_LAMBDA the weight matrix of the table product and the
// series of the covariate / phenotype pairs and of g = sum (1 - H).  Needs the fixed-lambda table (the scan reads it) and
// intervals no longer than the decade the accuracy figures were established on; GEMMA_HIP_ASSOC_CHEB=0 switches it off
// (every Brent / Newton evaluation then streams the row, as in round 1).
static int make_cheb(hipStream_t s) {
  AssocArgs &a = g_ctx.assoc_proto;
  a.have_cheb = 0;
  a.cheb_T = nullptr; a.cheb_F = nullptr; a.cheb_slots = nullptr; a.cheb_res = nullptr;
  const char *e = getenv("GEMMA_HIP_ASSOC_CHEB");
  if (e && e[0] == '0') return GEMMA_HIP_OK;
  const size_t n = g_ctx.cfg.n, c = g_ctx.cfg.n_cvt;
  const int nreg = (int)g_ctx.cfg.n_region;
  const double width = log(g_ctx.cfg.l_max / g_ctx.cfg.l_min) / (double)nreg;
  if (c < 1 || c > 4 || width > 2.31 || nreg > 62) return GEMMA_HIP_OK;
  // Intervals that start below lambda = 1e-3 are tabulated in Q form (series of sum a b delta H, the constant sum a b from
  // the fixed-lambda table) -- low-heritability traits stay on the table path; GEMMA_HIP_CHEB_LOWLAMBDA=0 leaves them to the
  // streaming evaluations as in round 2.
  int j0 = 0;
  const char *elow = getenv("GEMMA_HIP_CHEB_LOWLAMBDA");
  if (elow && elow[0] == '0')
    while (j0 < nreg && a.lam_grid[j0] < CHEB_MIN_LAMBDA * (1.0 - 1e-9)) ++j0;
  const int nint = nreg - j0;
  g_ctx.cheb_qmask = 0;
  if (nint <= 0) return GEMMA_HIP_OK;
  GridGeom gg;
  gg.nq = CHEB_N;
  gg.nbx = (CHEB_N + 15) / 16;
  gg.nba = ((int)(c + 1) * CHEB_N + 15) / 16;
  gg.nc = (int)((n + 15) / 16);
  if (!(gg.nbx == 2 && (gg.nba == 3 || gg.nba == 5 || gg.nba == 6 || gg.nba == 8))) return GEMMA_HIP_OK;
  const size_t nb = (size_t)(gg.nbx + gg.nba);
  const size_t r_elems = (size_t)gg.nc * nb * 256;
  const size_t npairs = (c + 1) * (c + 2) / 2;
  const size_t fld = (npairs + 3) * CHEB_N; // pairs, g, log|H|, sum (1 - H)^2
  if (g_ctx.cheb_R.reserve((size_t)nint * r_elems * 8) || g_ctx.cheb_F.reserve((size_t)nint * fld * 8) ||
      g_ctx.cheb_D.reserve(CHEB_N * CHEB_N * 8) || g_ctx.cheb_Ck.reserve(n * CHEB_N * 8) ||
      g_ctx.cheb_Gk.reserve(2 * n * CHEB_N * 8) || g_ctx.cheb_Lk.reserve(n * CHEB_N * 8) ||
      g_ctx.cheb_iv.reserve(2 * ASSOC_MAX_REGION * 8))
    return fail(GEMMA_HIP_ENOMEM, "lmm_setup: Chebyshev tables (%zu bytes)", (size_t)nint * r_elems * 8);
  // fit matrix: coefficients = D * node values (cheb_fit of lmm_search.hip.h)
  std::vector<double> D((size_t)CHEB_N * CHEB_N);
  for (int k = 0; k < CHEB_N; ++k)
    for (int m = 0; m < CHEB_N; ++m)
      D[(size_t)k * CHEB_N + m] = cos(M_PI * k * (m + 0.5) / CHEB_N) * (k == 0 ? 1.0 : 2.0) / CHEB_N;
  HIPCHK(hipMemcpyAsync(g_ctx.cheb_D.p, D.data(), D.size() * 8, hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s)); // D is a local
  AssocArgs k = a;
  k.eval = g_ctx.eval;
  k.Uty = g_ctx.Uty;
  k.UtWt = g_ctx.UtWt.as<double>();
  for (int q = 0; q < nint; ++q) {
    const ChebInterval iv = cheb_interval(a.lam_grid[j0 + q], a.lam_grid[j0 + q + 1], CHEB_MARGIN);
    g_ctx.cheb_mid[q] = iv.mid;
    g_ctx.cheb_inv_half[q] = 1.0 / iv.half;
    ChebNodes nd;
    for (int m = 0; m < CHEB_N; ++m) nd.lam[m] = exp(cheb_node(iv, m));
    // Q form by the interval's LOWER end: an interval of a non-default grid that straddles 1e-3 (e.g. [10^-3.5, 10^-2.5]) in
    // plain S form would carry 1e-13 / lambda of relative error in dS/dt at its low end; S0 - lambda Q stays well conditioned up
    // to the interval's upper end (<= a decade above, lambda <= 1e-2).  The default grid's nodes fall on 1e-3 either way.
    const int qform = a.lam_grid[j0 + q] < CHEB_MIN_LAMBDA * (1.0 - 1e-9) ? 1 : 0;
    if (qform) g_ctx.cheb_qmask |= 1ull << q;
    double *G2k = g_ctx.cheb_Gk.as<double>() + n * CHEB_N;
    hipLaunchKernelGGL(cheb_coeff_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, g_ctx.eval, (int)n, nd,
                       g_ctx.cheb_D.as<double>(), qform, g_ctx.cheb_Ck.as<double>(), g_ctx.cheb_Gk.as<double>(),
                       g_ctx.cheb_Lk.as<double>(), G2k);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(cheb_weights_kernel, dim3((unsigned)((r_elems + 255) / 256)), dim3(256), 0, s, k, gg, (int)c,
                       g_ctx.cheb_Ck.as<double>(), g_ctx.cheb_R.as<double>() + (size_t)q * r_elems);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(cheb_fixed_kernel, dim3((unsigned)(npairs + 3)), dim3(256), 0, s, k, (int)c,
                       g_ctx.cheb_Ck.as<double>(), g_ctx.cheb_Gk.as<double>(), g_ctx.cheb_Lk.as<double>(), G2k,
                       g_ctx.cheb_F.as<double>() + (size_t)q * fld);
    HIPCHK(hipGetLastError());
  }
  {
    std::vector<double> ivs(2 * (size_t)nint);
    for (int q = 0; q < nint; ++q) { ivs[2 * q] = g_ctx.cheb_mid[q]; ivs[2 * q + 1] = g_ctx.cheb_inv_half[q]; }
    HIPCHK(hipMemcpyAsync(g_ctx.cheb_iv.p, ivs.data(), ivs.size() * 8, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  g_ctx.cheb_geom = gg;
  a.cheb_iv = g_ctx.cheb_iv.as<double>();
  a.cheb_logdet_off = (int)((npairs + 1) * CHEB_N);
  a.cheb_F = g_ctx.cheb_F.as<double>();
  a.cheb_ld = (int)(nb * 16);
  a.cheb_fld = (int)fld;
  a.cheb_xa0 = gg.nbx * 16;
  a.cheb_j0 = j0;
  a.cheb_nint = nint;
  a.cheb_qmask = g_ctx.cheb_qmask;
  {
    // GEMMA_HIP_ASSOC_FINAL_SERIES=0: the final likelihood at lambda-hat streams the SNP's row as in round 2
    const char *ef = getenv("GEMMA_HIP_ASSOC_FINAL_SERIES");
    a.cheb_final = (ef && ef[0] == '0') ? 0 : 1;
  }
  a.have_cheb = 1;
  return GEMMA_HIP_OK;
}

static int make_grid(hipStream_t s) {
  AssocArgs &a = g_ctx.assoc_proto;
  a.have_grid = 0;
  a.grid_T = nullptr;
  a.grid_F = nullptr;
  const char *e = getenv("GEMMA_HIP_ASSOC_GRID");
  if (e && e[0] == '0') return GEMMA_HIP_OK;
  const size_t n = g_ctx.cfg.n, c = g_ctx.cfg.n_cvt;
  GridGeom gg;
  gg.nq = 1 + 2 * ((int)g_ctx.cfg.n_region + 1);
  if (!grid_blocks(c, gg.nq, &gg.nbx, &gg.nba)) return GEMMA_HIP_OK;
  gg.nc = (int)((n + 15) / 16);
  const size_t nb = (size_t)(gg.nbx + gg.nba);
  const size_t r_elems = (size_t)gg.nc * nb * 256;
  if (g_ctx.grid_R.reserve(r_elems * 8) || g_ctx.grid_F.reserve((size_t)gg.nq * GRID_FIX_LD * 8))
    return fail(GEMMA_HIP_ENOMEM, "lmm_setup: fixed-lambda table");
  AssocArgs k = a;
  k.eval = g_ctx.eval;
  k.Uty = g_ctx.Uty;
  k.UtWt = g_ctx.UtWt.as<double>();
  hipLaunchKernelGGL(grid_weights_kernel, dim3((unsigned)((r_elems + 255) / 256)), dim3(256), 0, s, k, gg, (int)c,
                     g_ctx.grid_R.as<double>());
  HIPCHK(hipGetLastError());
  hipLaunchKernelGGL(grid_fixed_kernel, dim3((unsigned)gg.nq), dim3(256), 0, s, k, (int)c, g_ctx.grid_F.as<double>());
  HIPCHK(hipGetLastError());
  g_ctx.grid_geom = gg;
  a.grid_F = g_ctx.grid_F.as<double>();
  a.grid_ld = (int)(nb * 16);
  a.grid_nq = gg.nq;
  a.grid_xa0 = gg.nbx * 16;
  a.have_grid = 1;
  return make_cheb(s);
}

// table_v2_kernel + table_reduce_kernel (lmm_grid.hip.h): T = [X.X | X] * R with RG * 16 rows per wave and the K range cut
// into slices; tg == nullptr: the dense fixed-lambda table of all l rows, else the per-interval gather tables
static bool table_v2_enabled() { return g_ctx.knobs.table_v2 != 0; }
template <int NBX, int NBA, int RG>
static int launch_table_v2_t(const GridGeom &gg, const double *UtX, size_t l, size_t ld, const double *R, double *T,
                             const TableGather *tg, int nint, hipStream_t s) {
  constexpr int NB16 = (NBX + NBA) * 16;
  const size_t rows_per_block = (size_t)RG * 16 * 4;
  const size_t bx = (l + rows_per_block - 1) / rows_per_block;
  // K slices: a function of n ALONE (a SNP's sums must not depend on the batch it arrives in: sharded == unsharded, and
  // tests/test_gpu_parity.py::test_lmm_reference_xlarge_layout_and_batching compares bits across batch sizes); 16 slices of
  // >= 32 chunks give ~5000 waves for a 20 000-row batch at n = 20 000
  const int ksplit = std::max(1, std::min(16, gg.nc / 32));
  const size_t planes = tg ? (size_t)nint : 1;
  if (g_ctx.table_P.reserve(planes * (size_t)ksplit * l * NB16 * 8))
    return fail(GEMMA_HIP_ENOMEM, "lmm_assoc: table partial sums (%zu bytes)", planes * (size_t)ksplit * l * NB16 * 8);
  TableV2 a;
  a.UtX = UtX; a.ld = (long)ld; a.l = (long)l; a.n = (int)g_ctx.cfg.n; a.nc = gg.nc; a.ksplit = ksplit; a.Rp = R;
  a.P = g_ctx.table_P.as<double>(); a.cap = (long)l;
  if (tg) a.tg = *tg; else a.tg = TableGather();
  const long total = (long)l * NB16;
  const bool pf = g_ctx.knobs.table_pf != 0;
  if (tg) {
    if (pf) hipLaunchKernelGGL((table_v2_kernel<NBX, NBA, RG, true, true>), dim3((unsigned)bx, (unsigned)ksplit, (unsigned)nint), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((table_v2_kernel<NBX, NBA, RG, true, false>), dim3((unsigned)bx, (unsigned)ksplit, (unsigned)nint), dim3(256), 0, s, a);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(table_reduce_kernel<true>, dim3((unsigned)((total + 255) / 256), (unsigned)nint), dim3(256), 0, s,
                       a.P, ksplit, a.cap, NB16, (long)l, tg->count, T);
  } else {
    if (pf) hipLaunchKernelGGL((table_v2_kernel<NBX, NBA, RG, false, true>), dim3((unsigned)bx, (unsigned)ksplit), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((table_v2_kernel<NBX, NBA, RG, false, false>), dim3((unsigned)bx, (unsigned)ksplit), dim3(256), 0, s, a);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(table_reduce_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a.P, ksplit,
                       a.cap, NB16, (long)l, (const int *)nullptr, T);
  }
  HIPCHK(hipGetLastError());
  return GEMMA_HIP_OK;
}
static int launch_table_v2(const GridGeom &gg, const double *UtX, size_t l, size_t ld, const double *R, double *T,
                           const TableGather *tg, int nint, hipStream_t s) {
  switch (gg.nba) {
  case 3: return launch_table_v2_t<2, 3, 4>(gg, UtX, l, ld, R, T, tg, nint, s);
  case 5: return launch_table_v2_t<2, 5, 2>(gg, UtX, l, ld, R, T, tg, nint, s);
  case 6: return launch_table_v2_t<2, 6, 2>(gg, UtX, l, ld, R, T, tg, nint, s);
  case 8: return launch_table_v2_t<2, 8, 2>(gg, UtX, l, ld, R, T, tg, nint, s);
  default: return fail(GEMMA_HIP_ERUNTIME, "lmm_assoc: no table kernel for %d column blocks", gg.nba);
  }
}

// the per-batch part: T = [X.X | X] * R for the l SNP rows of UtX
static int launch_grid_table(const double *UtX, size_t l, size_t ld, hipStream_t s) {
  const GridGeom &gg = g_ctx.grid_geom;
  const size_t nb = (size_t)(gg.nbx + gg.nba);
  if (g_ctx.grid_T.reserve(l * nb * 16 * 8)) return fail(GEMMA_HIP_ENOMEM, "lmm_assoc: fixed-lambda table");
  const unsigned grid = (unsigned)((l + 15) / 16);
  const double *R = g_ctx.grid_R.as<double>();
  double *T = g_ctx.grid_T.as<double>();
  const int n = (int)g_ctx.cfg.n;
  if (table_v2_enabled()) return launch_table_v2(gg, UtX, l, ld, R, T, nullptr, 0, s);
  switch (gg.nba) {
  case 3: hipLaunchKernelGGL((grid_table_kernel<2, 3, false>), dim3(grid), dim3(256), 0, s, UtX, (long)ld, (long)l, n, gg.nc, R, T, TableGather()); break;
  case 5: hipLaunchKernelGGL((grid_table_kernel<2, 5, false>), dim3(grid), dim3(256), 0, s, UtX, (long)ld, (long)l, n, gg.nc, R, T, TableGather()); break;
  case 6: hipLaunchKernelGGL((grid_table_kernel<2, 6, false>), dim3(grid), dim3(256), 0, s, UtX, (long)ld, (long)l, n, gg.nc, R, T, TableGather()); break;
  case 8: hipLaunchKernelGGL((grid_table_kernel<2, 8, false>), dim3(grid), dim3(256), 0, s, UtX, (long)ld, (long)l, n, gg.nc, R, T, TableGather()); break;
  default: return fail(GEMMA_HIP_ERUNTIME, "lmm_assoc: no fixed-lambda table kernel for %d column blocks", gg.nba);
  }
  HIPCHK(hipGetLastError());
  return GEMMA_HIP_OK;
}

// the per-batch part of the bracket-interval series: which (SNP, interval) pairs exist (scan over the fixed-lambda table),
// then the table product for exactly those rows.  `a` must already carry grid_T; fills a.cheb_T / a.cheb_slots.
static int launch_cheb_tables(AssocArgs &a, const double *UtX, size_t l, size_t ld, hipStream_t s) {
  const GridGeom &gg = g_ctx.cheb_geom;
  const size_t nb = (size_t)(gg.nbx + gg.nba), nint = (size_t)a.cheb_nint;
  if (g_ctx.cheb_T.reserve(nint * l * nb * 16 * 8) || g_ctx.cheb_slots.reserve(l * nint * sizeof(int)) ||
      g_ctx.cheb_list.reserve(nint * l * sizeof(int)) || g_ctx.cheb_count.reserve(ASSOC_MAX_REGION * sizeof(int)) ||
      g_ctx.cheb_dends.reserve(2 * nint * l * sizeof(double2)) || g_ctx.cheb_res.reserve(2 * nint * l * sizeof(ChebResult)))
    return fail(GEMMA_HIP_ENOMEM, "lmm_assoc: Chebyshev tables of the batch (%zu bytes)", nint * l * nb * 16 * 8);
  HIPCHK(hipMemsetAsync(g_ctx.cheb_count.p, 0, ASSOC_MAX_REGION * sizeof(int), s));
  a.cheb_T = g_ctx.cheb_T.as<double>();
  a.cheb_cap = (long)l;
  ChebScanArgs sc;
  sc.count = g_ctx.cheb_count.as<int>();
  sc.list = g_ctx.cheb_list.as<int>();
  sc.slots = g_ctx.cheb_slots.as<int>();
  sc.dends = g_ctx.cheb_dends.as<double2>();
  sc.cap = (long)l;
  const unsigned sgrid = (unsigned)((l + 3) / 4);
  const size_t c = g_ctx.cfg.n_cvt;
  switch (c) {
  case 1: hipLaunchKernelGGL(cheb_scan_kernel<1>, dim3(sgrid), dim3(256), 0, s, a, sc); break;
  case 2: hipLaunchKernelGGL(cheb_scan_kernel<2>, dim3(sgrid), dim3(256), 0, s, a, sc); break;
  case 3: hipLaunchKernelGGL(cheb_scan_kernel<3>, dim3(sgrid), dim3(256), 0, s, a, sc); break;
  default: hipLaunchKernelGGL(cheb_scan_kernel<4>, dim3(sgrid), dim3(256), 0, s, a, sc); break;
  }
  HIPCHK(hipGetLastError());
  TableGather tg;
  tg.list = sc.list;
  tg.count = sc.count;
  tg.cap = (long)l;
  tg.rp_stride = (long)gg.nc * (long)nb * 256;
  const dim3 grid((unsigned)((l + 15) / 16), (unsigned)nint);
  const double *R = g_ctx.cheb_R.as<double>();
  double *T = g_ctx.cheb_T.as<double>();
  const int n = (int)g_ctx.cfg.n;
  if (table_v2_enabled()) {
    int rc = launch_table_v2(gg, UtX, l, ld, R, T, &tg, (int)nint, s);
    if (rc) return rc;
  } else
  switch (gg.nba) {
  case 3: hipLaunchKernelGGL((grid_table_kernel<2, 3, true>), grid, dim3(256), 0, s, UtX, (long)ld, (long)l, n, gg.nc, R, T, tg); break;
  case 5: hipLaunchKernelGGL((grid_table_kernel<2, 5, true>), grid, dim3(256), 0, s, UtX, (long)ld, (long)l, n, gg.nc, R, T, tg); break;
  case 6: hipLaunchKernelGGL((grid_table_kernel<2, 6, true>), grid, dim3(256), 0, s, UtX, (long)ld, (long)l, n, gg.nc, R, T, tg); break;
  case 8: hipLaunchKernelGGL((grid_table_kernel<2, 8, true>), grid, dim3(256), 0, s, UtX, (long)ld, (long)l, n, gg.nc, R, T, tg); break;
  default: return fail(GEMMA_HIP_ERUNTIME, "lmm_assoc: no table kernel for %d column blocks", gg.nba);
  }
  HIPCHK(hipGetLastError());
  ChebSearchArgs sa;
  sa.count = sc.count;
  sa.list = sc.list;
  sa.qmask = g_ctx.cheb_qmask;
  sa.dends = sc.dends;
  sa.res = g_ctx.cheb_res.as<ChebResult>();
  memcpy(sa.mid, g_ctx.cheb_mid, sizeof sa.mid);
  memcpy(sa.inv_half, g_ctx.cheb_inv_half, sizeof sa.inv_half);
  const dim3 qgrid((unsigned)((l + 63) / 64), (unsigned)nint, 2);
  switch (c) {
  case 1: hipLaunchKernelGGL(cheb_search_kernel<1>, qgrid, dim3(64), 0, s, a, sa); break;
  case 2: hipLaunchKernelGGL(cheb_search_kernel<2>, qgrid, dim3(64), 0, s, a, sa); break;
  case 3: hipLaunchKernelGGL(cheb_search_kernel<3>, qgrid, dim3(64), 0, s, a, sa); break;
  default: hipLaunchKernelGGL(cheb_search_kernel<4>, qgrid, dim3(64), 0, s, a, sa); break;
  }
  HIPCHK(hipGetLastError());
  a.cheb_slots = sc.slots;
  a.cheb_res = sa.res;
  return GEMMA_HIP_OK;
}

// UtW (n x c row-major) -> UtWt (c x n)
static int make_utwt(const double *UtW_d, hipStream_t s) {
  const size_t n = g_ctx.cfg.n, c = g_ctx.cfg.n_cvt;
  if (g_ctx.UtWt.reserve(c * n * 8)) return fail(GEMMA_HIP_ENOMEM, "lmm_setup: UtWt");
  dim3 grid((unsigned)((c + 31) / 32), (unsigned)((n + 31) / 32));
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(32, 8), 0, s, UtW_d, (long)n, (long)c, (long)c,
                     g_ctx.UtWt.as<double>(), (long)n);
  HIPCHK(hipGetLastError());
  // SNP-independent log|H| at l_min and l_max
  if (g_ctx.scratch.reserve(16)) return fail(GEMMA_HIP_ENOMEM, "lmm_setup: scratch");
  hipLaunchKernelGGL(logdet_ends_kernel, dim3(1), dim3(64), 0, s, g_ctx.eval, (int)n, g_ctx.cfg.l_min, g_ctx.cfg.l_max,
                     g_ctx.scratch.as<double>());
  HIPCHK(hipGetLastError());
  double ends[2];
  HIPCHK(hipMemcpyAsync(ends, g_ctx.scratch.p, 16, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  g_ctx.assoc_proto.logdet_lmin = ends[0];
  g_ctx.assoc_proto.logdet_lmax = ends[1];
  g_ctx.assoc_proto.have_logdet_ends = 1;
  return make_grid(s);
}

extern "C" int gemma_hip_lmm_setup_d(const gemma_lmm_cfg *cfg, const double *U_d, const double *eval_d,
                                     const double *UtW_d, const double *Uty_d, void *stream) {
  NEED_INIT();
  if (!U_d || !eval_d || !UtW_d || !Uty_d) return fail(GEMMA_HIP_EINVAL, "lmm_setup: null pointer");
  int rc = lmm_common_setup(cfg);
  if (rc) return rc;
  g_ctx.U = U_d;
  g_ctx.U_even_of = nullptr;
  g_ctx.eval = eval_d;
  g_ctx.Uty = Uty_d;
  rc = make_utwt(UtW_d, S(stream));
  if (rc) return rc;
  g_ctx.lmm_active = true;
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_lmm_setup(const gemma_lmm_cfg *cfg, const double *U, const double *eval,
                                   const double *UtW, const double *Uty) {
  NEED_INIT();
  if (!U || !eval || !UtW || !Uty) return fail(GEMMA_HIP_EINVAL, "lmm_setup: null pointer");
  int rc = lmm_common_setup(cfg);
  if (rc) return rc;
  const size_t n = cfg->n, c = cfg->n_cvt;
  if (g_ctx.own_U.reserve(n * n * 8) || g_ctx.own_eval.reserve(n * 8) || g_ctx.own_Uty.reserve(n * 8) ||
      g_ctx.own_UtW.reserve(n * c * 8))
    return fail(GEMMA_HIP_ENOMEM, "lmm_setup: cannot allocate U (%zu bytes)", n * n * 8);
  HIPCHK(hipMemcpy(g_ctx.own_U.p, U, n * n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_ctx.own_eval.p, eval, n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_ctx.own_Uty.p, Uty, n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_ctx.own_UtW.p, UtW, n * c * 8, hipMemcpyHostToDevice));
  g_ctx.U = g_ctx.own_U.as<double>();
  g_ctx.U_even_of = nullptr;
  g_ctx.eval = g_ctx.own_eval.as<double>();
  g_ctx.Uty = g_ctx.own_Uty.as<double>();
  rc = make_utwt(g_ctx.own_UtW.as<double>(), 0);
  if (rc) return rc;
  HIPCHK(hipDeviceSynchronize());
  g_ctx.lmm_active = true;
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_lmm_set_indicator(const int *indicator_idv, size_t ni_total) {
  NEED_INIT();
  if (!g_ctx.lmm_active && !g_ctx.lm_active) return fail(GEMMA_HIP_ESTATE, "lmm_set_indicator before lmm_setup / lm_setup");
  if (!indicator_idv || ni_total == 0) {
    g_ctx.have_map = false;
    g_ctx.ni_total = 0;
    return GEMMA_HIP_OK;
  }
  std::vector<int> map;
  map.reserve(g_ctx.cfg.n);
  for (size_t i = 0; i < ni_total; ++i)
    if (indicator_idv[i] != 0) map.push_back((int)i);
  if (map.size() != g_ctx.cfg.n)
    return fail(GEMMA_HIP_EINVAL, "lmm_set_indicator: %zu analysed individuals, cfg.n = %zu", map.size(),
                g_ctx.cfg.n);
  if (map.size() == ni_total) { // everybody is analysed: the identity needs no mapping (and PLINK rows take the word-wise ingest)
    g_ctx.have_map = false;
    g_ctx.ni_total = 0;
    return GEMMA_HIP_OK;
  }
  if (g_ctx.idx_map.reserve(map.size() * sizeof(int))) return fail(GEMMA_HIP_ENOMEM, "idx_map");
  HIPCHK(hipMemcpy(g_ctx.idx_map.p, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice));
  g_ctx.have_map = true;
  g_ctx.ni_total = ni_total;
  return GEMMA_HIP_OK;
}

static int launch_assoc(const double *UtX, size_t l, size_t ld, gemma_sumstat *out_d, hipStream_t s) {
  AssocArgs a = g_ctx.assoc_proto;
  a.UtX = UtX;
  a.ld = (long)ld;
  a.l = (long)l;
  a.eval = g_ctx.eval;
  a.Uty = g_ctx.Uty;
  a.UtWt = g_ctx.UtWt.as<double>();
  a.out = reinterpret_cast<SumStat *>(out_d);
  const unsigned grid = (unsigned)((l + 3) / 4);
  {
    ProfScope ps(GEMMA_STAGE_ASSOC, s);
    // GEMMA_HIP_FORCE_GENERIC=1 routes every covariate count through the multi-pass kernel (tests)
    const size_t sel = g_ctx.knobs.force_generic ? 99 : g_ctx.cfg.n_cvt;
    a.grid_T = nullptr;
    if (a.have_grid && sel <= 4 && (ld & 1) == 0 && (reinterpret_cast<uintptr_t>(UtX) & 15) == 0 &&
        a.a_mode != 3) { // mode 3 (score only) never searches lambda
      // The tables are an accelerator, not a requirement: when their buffers do not fit (the K-slice partial sums are
      // planes x slices x l x 80 doubles -- 1.6 GB at l = n = 20000, c = 1) the batch falls back to the streaming evaluations
      // (the round-1 path, same statistics) instead of failing.
      int rc = launch_grid_table(UtX, l, ld, s);
      if (rc && rc != GEMMA_HIP_ENOMEM) return rc;
      a.grid_T = rc ? nullptr : g_ctx.grid_T.as<double>();
      a.cheb_T = nullptr;
      a.cheb_slots = nullptr;
      a.cheb_res = nullptr;
      if (!rc && a.have_cheb) {
        rc = launch_cheb_tables(a, UtX, l, ld, s);
        if (rc && rc != GEMMA_HIP_ENOMEM) return rc;
        if (rc) { a.cheb_T = nullptr; a.cheb_slots = nullptr; a.cheb_res = nullptr; }
      }
    }
    switch (sel) {
    case 1: {
      // streaming-loop unroll / wavefronts per SIMD of the c = 1 kernel; measured at n = 20000 (ms per 20000 SNPs):
      // 2/3: 14.3, 4/3: 14.1, 8/3: 13.9, 2/4: 12.9, 4/4: 12.8 (default), 4/2: 14.2
      const int var = g_ctx.knobs.assoc_variant;
      if (var == 43) hipLaunchKernelGGL((lmm_assoc1_variant_kernel<4, 3>), dim3(grid), dim3(256), 0, s, a);
      else if (var == 83) hipLaunchKernelGGL((lmm_assoc1_variant_kernel<8, 3>), dim3(grid), dim3(256), 0, s, a);
      else if (var == 24) hipLaunchKernelGGL((lmm_assoc1_variant_kernel<2, 4>), dim3(grid), dim3(256), 0, s, a);
      else if (var == 23) hipLaunchKernelGGL(lmm_assoc_kernel<1>, dim3(grid), dim3(256), 0, s, a);
      else if (var == 42) hipLaunchKernelGGL((lmm_assoc1_variant_kernel<4, 2>), dim3(grid), dim3(256), 0, s, a);
      else hipLaunchKernelGGL((lmm_assoc1_variant_kernel<4, 4>), dim3(grid), dim3(256), 0, s, a);
      break;
    }
    case 2: hipLaunchKernelGGL(lmm_assoc_kernel<2>, dim3(grid), dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL(lmm_assoc_kernel<3>, dim3(grid), dim3(256), 0, s, a); break;
    case 4: hipLaunchKernelGGL(lmm_assoc_kernel<4>, dim3(grid), dim3(256), 0, s, a); break;
    default: // more covariates: register-tiled multi-pass path
      if (g_ctx.cfg.n_cvt > (size_t)GEN_CMAX) {
        int rcw = wide_attr(lmm_assoc_wide_kernel);
        if (rcw) return rcw;
        hipLaunchKernelGGL(lmm_assoc_wide_kernel, dim3((unsigned)l), dim3(64), wide_lds_bytes(g_ctx.cfg.n_cvt), s, a,
                           (int)g_ctx.cfg.n_cvt);
      } else {
        hipLaunchKernelGGL(lmm_assoc_generic_kernel, dim3(grid), dim3(256), 0, s, a, (int)g_ctx.cfg.n_cvt);
      }
      break;
    }
    HIPCHK(hipGetLastError());
    if (g_ctx.cfg.plink_nan_rule && g_ctx.cfg.a_mode == 1) {
      double *cin = g_ctx.carry.as<double>() + 2 * g_ctx.carry_flip;
      double *cout = g_ctx.carry.as<double>() + 2 * (1 - g_ctx.carry_flip);
      hipLaunchKernelGGL(plink_carry_kernel, dim3((unsigned)((l + 255) / 256)), dim3(256), 0, s,
                         reinterpret_cast<SumStatRaw *>(out_d), (long)l, cin, cout);
      HIPCHK(hipGetLastError());
      g_ctx.carry_flip = 1 - g_ctx.carry_flip;
    }
  }
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_lmm_assoc_d(const double *UtX_d, size_t l, size_t ld_utx, gemma_sumstat *out_d,
                                     void *stream) {
  NEED_INIT();
  if (!g_ctx.lmm_active) return fail(GEMMA_HIP_ESTATE, "lmm_assoc before lmm_setup");
  if (l == 0) return GEMMA_HIP_OK;
  if (!UtX_d || !out_d || ld_utx < g_ctx.cfg.n) return fail(GEMMA_HIP_EINVAL, "lmm_assoc: bad UtX/ld");
  {
    int rcf = xp_flush_fwd(S(stream));
    if (rcf) return rcf;
  }
  return launch_assoc(UtX_d, l, ld_utx, out_d, S(stream));
}

// GEMMA_HIP_UTX_I8: 1 (default) = hard-call batches (PLINK 2-bit; fp64 input whose rows hold only 0/1/2 and one
// missing / imputed value) go through the exact int8-digit product (i8gemm.hip.h), 0 = always the fp64 MFMA GEMM.
// Real-valued dosages always take the fp64 GEMM.
static int utx_i8_mode() { return g_ctx.knobs.utx_i8; }

// which matrix kernel the product of the batch launched (gemma_hip_dbg_last_utx_kernel): bench.py labels its roofline from this,
// not from the environment
static void note_utx_kernel(int variant, int digits, int fuse, int raster) {
  static const char *const names[GEMMA_UTX_KERNEL_COUNT] = {
      "dgemm_mfma_glds_kernel", "i8gemm_packed_kernel_t<true>", "i8gemm_sparse_kernel", "i8gemm_sparse2_kernel",
      "i8gemm_sparse2_r16_kernel", "i8gemm_packed_kernel_t<false, true>", "i8gemm_dense16_kernel_t<true>"};
  gemma_utx_kernel_info &k = g_ctx.last_utx_kernel;
  k.variant = variant;
  k.rows = (variant == GEMMA_UTX_KERNEL_RECORDS_R16 || variant == GEMMA_UTX_KERNEL_DOSAGE_I8_R16) ? 16
                                                                                                   : (variant == GEMMA_UTX_KERNEL_DGEMM_F64 ? 0 : 32);
  k.digits = digits; k.fuse = fuse; k.raster = raster;
  k.launches += 1;
  snprintf(k.name, sizeof k.name, "%s", names[variant]);
}

static size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }

// digits of U in the exact int8 product (i8gemm.hip.h): 7, or 6 from n = 16384 up where the 2^-48 rounding of U stays at
// the level of an fp64 GEMM's own rounding; GEMMA_HIP_I8_DIGITS=6|7 forces either
static int i8_digits_for(size_t n) {
  if (g_ctx.knobs.i8_digits) return g_ctx.knobs.i8_digits;
  return n >= 16384 ? 6 : 7;
}

// one-time: per-column exponents of U and its 7 balanced base-256 digit matrices, transposed (K contiguous)
static int i8_prepare_u(hipStream_t s) {
  if (g_ctx.i8_ready) return GEMMA_HIP_OK;
  const size_t n = g_ctx.cfg.n;
  const size_t ldk = round_up(n, I8_BK), npad = round_up(n, I8_BN);
  g_ctx.i8_digits = i8_digits_for(n);
  if (g_ctx.i8_Bt.reserve((size_t)I8_DIGITS * npad * ldk) || g_ctx.i8_q.reserve(n * 8) || g_ctx.i8_qinv.reserve(n * 8) ||
      g_ctx.i8_cmax.reserve(n * 8))
    return fail(GEMMA_HIP_ENOMEM, "lmm_batch: int8 digits of U (%zu bytes)", (size_t)I8_DIGITS * npad * ldk);
  HIPCHK(hipMemsetAsync(g_ctx.i8_Bt.p, 0, (size_t)I8_DIGITS * npad * ldk, s));
  HIPCHK(hipMemsetAsync(g_ctx.i8_cmax.p, 0, n * 8, s));
  hipLaunchKernelGGL(u_colmax_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)((n + 1023) / 1024)), dim3(256), 0, s,
                     g_ctx.U, (long)n, (long)n, g_ctx.i8_cmax.as<unsigned long long>());
  HIPCHK(hipGetLastError());
  hipLaunchKernelGGL(u_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                     g_ctx.i8_cmax.as<unsigned long long>(), (long)n, g_ctx.i8_digits, g_ctx.knobs.i8_scale_max,
                     g_ctx.i8_q.as<double>(), g_ctx.i8_qinv.as<double>());
  HIPCHK(hipGetLastError());
  hipLaunchKernelGGL(u_digits_kernel, dim3((unsigned)((n + 31) / 32), (unsigned)((n + 31) / 32)), dim3(256), 0, s,
                     g_ctx.U, (long)n, (long)n, g_ctx.i8_q.as<double>(), g_ctx.i8_Bt.as<int8_t>(), (long)ldk,
                     (long)(npad * ldk), g_ctx.i8_digits);
  HIPCHK(hipGetLastError());
  g_ctx.i8_ldk = ldk;
  g_ctx.i8_npad = npad;
  g_ctx.i8_ready = true;
  return GEMMA_HIP_OK;
}
