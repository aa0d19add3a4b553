// Part of gemma_hip.hip (ONE translation unit: the parts share the context g_ctx and the helpers of its anonymous namespace, and are
// included there in this order; round 6: the 3 500-line file cut along its stages for reading -- no behaviour change).
// This part: kinship: PARAM::CalcKin -> BimbamKin / PlinkKin (gemma_hip_kin_begin / kin_add / kin_end), the exact-integer path of hard calls.

// ------------------------------------------------------------------------------ kinship
extern "C" int gemma_hip_kin_begin(size_t n_total, int k_mode) {
  NEED_INIT();
  if (n_total == 0) return fail(GEMMA_HIP_EINVAL, "kin_begin: n_total == 0");
  if (k_mode != 1 && k_mode != 2) return fail(GEMMA_HIP_EINVAL, "kin_begin: k_mode %d", k_mode);
  g_ctx.qc_G.release(); g_ctx.qc_M.release(); g_ctx.qc_W.release(); g_ctx.qc_O.release(); // the first pass is over
  g_ctx.kin_ingested_valid = false;
  if (g_ctx.kin_K.reserve(n_total * n_total * 8))
    return fail(GEMMA_HIP_ENOMEM, "kin_begin: cannot allocate K (%zu bytes)", n_total * n_total * 8);
  HIPCHK(hipMemsetAsync(g_ctx.kin_K.p, 0, n_total * n_total * 8, 0));
  g_ctx.kin_active = true;
  g_ctx.kin_n = n_total;
  g_ctx.kin_mode = k_mode;
  g_ctx.kin_ns = 0;
  // -gk 1 on PLINK 2-bit blocks: G^T G as an exact int8 product + a sparse pass over the missing calls (kin_i8.hip.h);
  // GEMMA_HIP_KIN_I8=0 keeps every block on the fp64 SYRK
  g_ctx.knobs.load();
  g_ctx.kin_i8 = (k_mode == 1) && g_ctx.knobs.kin_i8;
  g_ctx.kin_i8_used = false;
  return GEMMA_HIP_OK;
}

static void kin_i8_release() {
  g_ctx.kin_GtG.release(); g_ctx.kin_S.release(); g_ctx.kin_a.release(); g_ctx.kin_At.release(); g_ctx.kin_Gt.release();
  g_ctx.kin_A2.release(); g_ctx.kin_cnt.release(); g_ctx.kin_off.release(); g_ctx.kin_listS.release();
  g_ctx.kin_listJ.release(); g_ctx.kin_sub.release(); g_ctx.kin_cj.release(); g_ctx.kin_flag.release();
  g_ctx.kin_tmap.release();
  g_ctx.kin_tmap_tm = g_ctx.kin_tmap_tn = g_ctx.kin_tmap_count = 0;
  g_ctx.kin_i8_used = false;
}

// (tile_m, tile_n) of the 128 x 256 tiles of G^T G that hold an entry with column >= row, in the order the kernel's raster
// would visit them (groups of eight tile rows, columns outside, rows inside: one L2 patch per XCD)
static int kin_i8_tile_map(int tiles_m, int tiles_n) {
  if (g_ctx.kin_tmap_tm == tiles_m && g_ctx.kin_tmap_tn == tiles_n && g_ctx.kin_tmap.p) return GEMMA_HIP_OK;
  std::vector<int> map;
  const int GM = 8;
  for (int first = 0; first < tiles_m; first += GM) {
    const int gsz = std::min(GM, tiles_m - first);
    for (int tn = first >> 1; tn < tiles_n; ++tn)
      for (int tm = first; tm < first + gsz; ++tm)
        if (tn >= (tm >> 1)) { // columns 256 tn .. + 255 reach row 128 tm
          map.push_back(tm);
          map.push_back(tn);
        }
  }
  if (g_ctx.kin_tmap.reserve(map.size() * sizeof(int))) return fail(GEMMA_HIP_ENOMEM, "kin_add: tile map");
  HIPCHK(hipMemcpy(g_ctx.kin_tmap.p, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice));
  g_ctx.kin_tmap_tm = tiles_m;
  g_ctx.kin_tmap_tn = tiles_n;
  g_ctx.kin_tmap_count = (int)(map.size() / 2);
  return GEMMA_HIP_OK;
}

// one PLINK block through the integer path: packed rows, transposed operands, G^T G (int32, exact), accumulators
static int kin_add_i8(const void *geno, size_t l, size_t ld, hipStream_t s) {
  const size_t n = g_ctx.kin_n;
  const size_t ldk = (n + I8_BK - 1) / I8_BK * I8_BK;          // bytes per SNP-major row (K of the LMM product; here the i axis)
  const size_t ldl = (l + I8_BK - 1) / I8_BK * I8_BK;          // bytes per individual-major row (K of THIS product: SNPs)
  const size_t rows_a = (n + I8P_BM - 1) / I8P_BM * I8P_BM;    // A operand rows (128-row tiles)
  const size_t rows_b = (n + I8_BN - 1) / I8_BN * I8_BN;       // B operand rows (256-column tiles)
  const size_t rows_t = std::max(rows_a, rows_b);
  if (!g_ctx.kin_i8_used) {
    if (g_ctx.kin_GtG.reserve(n * n * 8) || g_ctx.kin_S.reserve(n * n * 8) || g_ctx.kin_a.reserve((n + 1) * 8))
      return fail(GEMMA_HIP_ENOMEM, "kin_add: integer-path accumulators (%zu bytes)", 2 * n * n * 8);
    HIPCHK(hipMemsetAsync(g_ctx.kin_GtG.p, 0, n * n * 8, s));
    HIPCHK(hipMemsetAsync(g_ctx.kin_S.p, 0, n * n * 8, s));
    HIPCHK(hipMemsetAsync(g_ctx.kin_a.p, 0, (n + 1) * 8, s));
    g_ctx.kin_i8_used = true;
  }
  if (g_ctx.i8_A.reserve(l * ldk) || g_ctx.i8_mean.reserve(l * 8) || g_ctx.kin_At.reserve(rows_t * ldl) ||
      g_ctx.kin_Gt.reserve(rows_t * ldl) || g_ctx.i8_C.reserve(rows_a * rows_b * 4))
    return fail(GEMMA_HIP_ENOMEM, "kin_add: integer-path buffers");
  {
    ProfScope ps(GEMMA_STAGE_INGEST, s);
    IngestI8Args a;
    a.src = reinterpret_cast<const unsigned char *>(geno); a.ld = (long)ld; a.l = (long)l; a.idx_map = nullptr;
    a.n = (int)n; a.A = g_ctx.i8_A.as<int8_t>(); a.ldk = (long)ldk; a.mean = g_ctx.i8_mean.as<double>();
    hipLaunchKernelGGL(ingest_i8_kernel, dim3((unsigned)((l + 3) / 4)), dim3(256), 0, s, a);
    HIPCHK(hipGetLastError());
    if (g_ctx.kin_host_call) { // ingest_i8_kernel was the only reader of gemma_hip_kin_add's staging buffer
      HIPCHK(hipEventRecord(g_ctx.kin_ingested, s));
      g_ctx.kin_ingested_valid = true;
    }
    hipLaunchKernelGGL(kin_i8_transpose_kernel, dim3((unsigned)((ldl + 63) / 64), (unsigned)((rows_t + 63) / 64)), dim3(256), 0,
                       s, g_ctx.i8_A.as<int8_t>(), (long)l, (long)ldk, (long)n, g_ctx.kin_At.as<int8_t>(),
                       g_ctx.kin_Gt.as<int8_t>(), (long)ldl, (long)rows_t);
    HIPCHK(hipGetLastError());
  }
  {
    ProfScope ps(GEMMA_STAGE_KIN_GEMM, s);
    static bool attr_set = false;
    if (!attr_set) {
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(i8gemm_packed_kernel_t<false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 3 * I8P_STAGE));
      attr_set = true;
    }
    I8PackArgs g;
    g.A = g_ctx.kin_At.as<int8_t>();  // rows = individuals, K = SNPs; the kernel masks g = a & 3 (WITH_M = false: no mask product)
    g.Bt = g_ctx.kin_Gt.as<int8_t>(); // the same block as plain genotypes: C = G^T G
    g.C = g_ctx.i8_C.as<int>();
    g.ldk = (long)ldl; g.ldc = (long)rows_b;
    g.strideB = 0; g.strideC = 0;
    g.m_row0 = (long)rows_a;
    g.tiles_m = (int)(rows_a / I8P_BM); g.tiles_n = (int)(rows_b / I8_BN);
    g.nk = (int)(ldl / I8_BK);
    g.gm = 0; g.fuse = 0; g.digits = 1;
    unsigned ntiles = (unsigned)(g.tiles_m * g.tiles_n);
    {
      // the product is symmetric and kin_i8_fold_kernel reads its upper triangle only: the tiles below it are not formed
      // (GEMMA_HIP_KIN_UPPER=0: all of them, as in round 2)
      if (g_ctx.knobs.kin_upper) {
        if (int rc = kin_i8_tile_map(g.tiles_m, g.tiles_n)) return rc;
        g.tile_map = g_ctx.kin_tmap.as<int>();
        ntiles = (unsigned)g_ctx.kin_tmap_count;
      }
    }
    hipLaunchKernelGGL(i8gemm_packed_kernel_t<false>, dim3(ntiles, 1), dim3(512), 3 * I8P_STAGE, s, g);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(kin_i8_accum_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)std::min<size_t>(n, 32768)), dim3(256), 0, s,
                       g_ctx.i8_C.as<int>(), (long)rows_b, (long)n, g_ctx.kin_GtG.as<double>());
    HIPCHK(hipGetLastError());
    KinCorrArgs c;
    c.A = g_ctx.i8_A.as<int8_t>(); c.At = g_ctx.kin_At.as<int8_t>(); c.mean = g_ctx.i8_mean.as<double>();
    c.l = (long)l; c.ldk = (long)ldk; c.ldl = (long)ldl; c.n = (long)n;
    c.S = g_ctx.kin_S.as<double>(); c.a = g_ctx.kin_a.as<double>(); c.smu2 = g_ctx.kin_a.as<double>() + n;
    c.lists_ok = nullptr;
    // the correction on lists of the missing calls (kin_i8.hip.h, round 3); GEMMA_HIP_KIN_LISTS=0 keeps the round-2 kernel,
    // GEMMA_HIP_KIN_LIST_CAP=<entries> overrides the list capacity (tests: forces the on-device fall-back)
    bool lists = g_ctx.knobs.kin_lists && l < ((size_t)1 << 18);
    const unsigned nseg = (unsigned)((n + KI8_SEG - 1) / KI8_SEG);
    size_t cap = std::max<size_t>(l * n / 16, (size_t)1 << 20);
    if (g_ctx.knobs.kin_list_cap) cap = std::max<size_t>((size_t)g_ctx.knobs.kin_list_cap, 1);
    cap = std::min<size_t>(cap, (size_t)1 << 30);
    const size_t ld2 = (size_t)256 * nseg; // dwords per row of the 2-bit copy (kin_i8_pack2_kernel)
    // the list buffers are an optimisation: when they do not fit (GEMMA_HIP_KIN_LISTS_OOM=1 simulates it) the round-2 kernel,
    // which needs none of them, takes the whole correction -- as launch_assoc degrades when its tables do not fit
    if (lists && (g_ctx.knobs.kin_lists_oom ||
                  g_ctx.kin_A2.reserve(l * ld2 * 4) || g_ctx.kin_cnt.reserve((l + n) * 4) || g_ctx.kin_off.reserve((l + n + 2) * 4) ||
                  g_ctx.kin_listS.reserve(cap * 4) || g_ctx.kin_listJ.reserve(cap * 4) ||
                  g_ctx.kin_sub.reserve(l * (size_t)(nseg + 1) * 4) || g_ctx.kin_cj.reserve(n * 8) || g_ctx.kin_flag.reserve(16))) {
      (void)hipGetLastError();
      lists = false;
    }
    if (lists) {
      int *cntS = g_ctx.kin_cnt.as<int>(), *cntJ = cntS + l, *offS = g_ctx.kin_off.as<int>(), *offJ = offS + l + 1;
      int *ok = g_ctx.kin_flag.as<int>();
      hipLaunchKernelGGL(kin_i8_pack2_kernel, dim3((unsigned)l, nseg), dim3(256), 0, s, g_ctx.i8_A.as<int8_t>(), (long)l,
                         (long)ldk, (int)nseg, g_ctx.kin_A2.as<unsigned>());
      hipLaunchKernelGGL(kin_i8_count_kernel, dim3((unsigned)((l + 3) / 4)), dim3(256), 0, s, g_ctx.i8_A.as<int8_t>(), (long)l,
                         (long)ldk, (long)ldk, cntS);
      hipLaunchKernelGGL(kin_i8_count_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, g_ctx.kin_At.as<int8_t>(), (long)n,
                         (long)ldl, (long)ldl, cntJ);
      KinScanArgs sc;
      sc.cntS = cntS; sc.cntJ = cntJ; sc.offS = offS; sc.offJ = offJ; sc.l = (long)l; sc.n = (long)n; sc.cap = (long)cap;
      sc.ok = ok; sc.mean = g_ctx.i8_mean.as<double>(); sc.smu2 = g_ctx.kin_a.as<double>() + n;
      hipLaunchKernelGGL(kin_i8_scan_kernel, dim3(1), dim3(1024), 0, s, sc);
      hipLaunchKernelGGL(kin_i8_fill_kernel<false>, dim3((unsigned)((l + 3) / 4)), dim3(256), 0, s, g_ctx.i8_A.as<int8_t>(),
                         (long)l, (long)ldk, (long)ldk, offS, g_ctx.kin_listS.as<int>(), ok, (const double *)nullptr, (long)l,
                         (double *)nullptr, (double *)nullptr);
      hipLaunchKernelGGL(kin_i8_fill_kernel<true>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, g_ctx.kin_At.as<int8_t>(),
                         (long)n, (long)ldl, (long)ldl, offJ, g_ctx.kin_listJ.as<int>(), ok, g_ctx.i8_mean.as<double>(), (long)l,
                         g_ctx.kin_a.as<double>(), g_ctx.kin_cj.as<double>());
      hipLaunchKernelGGL(kin_i8_sub_kernel, dim3((unsigned)((l * (nseg + 1) + 255) / 256)), dim3(256), 0, s, offS,
                         g_ctx.kin_listS.as<int>(), (long)l, (int)nseg, ok, g_ctx.kin_sub.as<int>());
      KinCorr2Args c2;
      c2.A2 = g_ctx.kin_A2.as<unsigned>(); c2.ld2 = (long)ld2; c2.mean = g_ctx.i8_mean.as<double>(); c2.n = (long)n;
      c2.offJ = offJ; c2.listJ = g_ctx.kin_listJ.as<int>(); c2.offS = offS; c2.listS = g_ctx.kin_listS.as<int>();
      c2.sub = g_ctx.kin_sub.as<int>(); c2.nseg = (int)nseg; c2.cj = g_ctx.kin_cj.as<double>();
      c2.S = g_ctx.kin_S.as<double>(); c2.ok = ok;
      c2.dbg_skip_pairs = c2.dbg_skip_main = 0;
#ifdef GEMMA_HIP_KIN_TIMING_SWITCHES // timing experiments only (results wrong): never in the shipped library
      {
        const char *ed = getenv("GEMMA_HIP_KIN_DBG");
        c2.dbg_skip_pairs = (ed && ed[0] == '1') ? 1 : 0;
        c2.dbg_skip_main = (ed && ed[0] == '2') ? 1 : 0;
      }
#endif
      hipLaunchKernelGGL(kin_i8_corr2_kernel, dim3((unsigned)n, nseg), dim3(256), 0, s, c2);
      HIPCHK(hipGetLastError());
      c.lists_ok = ok;
    }
    hipLaunchKernelGGL(kin_i8_corr_kernel, dim3((unsigned)n, nseg), dim3(256), 0, s, c);
    HIPCHK(hipGetLastError());
  }
  g_ctx.kin_ns += l;
  return GEMMA_HIP_OK;
}

// fold the integer-path accumulators into the (unscaled, upper-triangle) sums of kin_K; call before the scale / mirror
static int kin_fold_i8(hipStream_t s) {
  if (!g_ctx.kin_i8_used) return GEMMA_HIP_OK;
  const size_t n = g_ctx.kin_n;
  const unsigned nb = (unsigned)((n + 31) / 32);
  hipLaunchKernelGGL(kin_i8_fold_kernel, dim3(nb, nb), dim3(32, 8), 0, s, g_ctx.kin_K.as<double>(), (long)n,
                     g_ctx.kin_GtG.as<double>(), g_ctx.kin_S.as<double>(), g_ctx.kin_a.as<double>(),
                     g_ctx.kin_a.as<double>() + n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(s));
  kin_i8_release();
  return GEMMA_HIP_OK;
}

static size_t min_ld_for(int kind, size_t n_items_per_row, size_t l) {
  switch (kind) {
  case GEMMA_GENO_F64_SNP_MAJOR: return n_items_per_row;
  case GEMMA_GENO_PLINK_2BIT: return (n_items_per_row + 3) / 4;
  case GEMMA_GENO_F64_IDV_MAJOR: return l;
  default: return (size_t)-1;
  }
}

extern "C" int gemma_hip_kin_add_d(int kind, const void *geno, size_t l, size_t ld, void *stream) {
  NEED_INIT();
  if (!g_ctx.kin_active) return fail(GEMMA_HIP_ESTATE, "kin_add before kin_begin");
  if (l == 0) return GEMMA_HIP_OK;
  const size_t n = g_ctx.kin_n;
  const size_t need = min_ld_for(kind, n, l);
  if (need == (size_t)-1) return fail(GEMMA_HIP_EINVAL, "kin_add: unknown geno_kind %d", kind);
  if (!geno || ld < need) return fail(GEMMA_HIP_EINVAL, "kin_add: ld=%zu < %zu", ld, need);
  hipStream_t s = S(stream);
  if (g_ctx.kin_i8 && kind == GEMMA_GENO_PLINK_2BIT) return kin_add_i8(geno, l, ld, s);
  const size_t ldx = (n + 1) & ~(size_t)1;
  if (g_ctx.kin_X.reserve(l * ldx * 8))
    return fail(GEMMA_HIP_ENOMEM, "kin_add: cannot allocate %zu bytes", l * ldx * 8);
  double *X = g_ctx.kin_X.as<double>();
  {
    ProfScope ps(GEMMA_STAGE_INGEST, s);
    if (kind == GEMMA_GENO_F64_IDV_MAJOR) {
      dim3 grid((unsigned)((l + 31) / 32), (unsigned)((n + 31) / 32));
      hipLaunchKernelGGL(transpose_kernel, grid, dim3(32, 8), 0, s,
                         reinterpret_cast<const double *>(geno), (long)n, (long)l, (long)ld, X, (long)ldx);
    } else {
      IngestArgs a;
      a.src = geno; a.ld = (long)ld; a.l = (long)l; a.idx_map = nullptr; a.n = (int)n;
      a.dst = X; a.ldo = (long)ldx; a.k_mode = g_ctx.kin_mode;
      const unsigned grid = (unsigned)((l + 3) / 4);
      if (kind == GEMMA_GENO_PLINK_2BIT)
        hipLaunchKernelGGL(ingest_kin_kernel<true>, dim3(grid), dim3(256), 0, s, a);
      else
        hipLaunchKernelGGL(ingest_kin_kernel<false>, dim3(grid), dim3(256), 0, s, a);
    }
    HIPCHK(hipGetLastError());
  }
  if (g_ctx.kin_host_call) { // the staging buffer of gemma_hip_kin_add is free again from here on
    HIPCHK(hipEventRecord(g_ctx.kin_ingested, s));
    g_ctx.kin_ingested_valid = true;
  }
  {
    // K(upper tiles) += X^T X : A = X as [k = snp][m = individual]  -> ('T','N')
    ProfScope ps(GEMMA_STAGE_KIN_GEMM, s);
    HIPCHK(launch_dgemm('T', 'N', (long)n, (long)n, (long)l, 1.0, X, (long)ldx, X, (long)ldx, 1.0,
                        g_ctx.kin_K.as<double>(), (long)n, true, false, s));
  }
  g_ctx.kin_ns += l;
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_kin_add(int kind, const void *geno, size_t l, size_t ld) {
  NEED_INIT();
  if (!g_ctx.kin_active) return fail(GEMMA_HIP_ESTATE, "kin_add before kin_begin");
  if (l == 0) return GEMMA_HIP_OK;
  const size_t n = g_ctx.kin_n;
  const size_t need = min_ld_for(kind, n, l);
  if (need == (size_t)-1) return fail(GEMMA_HIP_EINVAL, "kin_add: unknown geno_kind %d", kind);
  if (!geno || ld < need) return fail(GEMMA_HIP_EINVAL, "kin_add: ld=%zu < %zu", ld, need);
  const size_t rows = (kind == GEMMA_GENO_F64_IDV_MAJOR) ? n : l;
  const size_t esz = (kind == GEMMA_GENO_PLINK_2BIT) ? 1 : 8;
  const size_t bytes = rows * ld * esz;
  if (bytes > g_ctx.kin_stage.cap) g_ctx.kin_ingested_valid = false; // (the hipFree inside reserve waits for the device)
  if (g_ctx.kin_stage.reserve(bytes)) return fail(GEMMA_HIP_ENOMEM, "kin_add: staging %zu bytes", bytes);
  // last row may be shorter than ld in the caller's buffer
  const size_t width = need * esz;
  if (!g_ctx.kin_copy) { // created once; without it the upload stays on the null stream (behind the previous block's kernels)
    if (hipStreamCreateWithFlags(&g_ctx.kin_copy, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&g_ctx.kin_copied, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g_ctx.kin_ingested, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      if (g_ctx.kin_copy) (void)hipStreamDestroy(g_ctx.kin_copy);
      g_ctx.kin_copy = nullptr;
    }
  }
  if (!g_ctx.kin_copy) {
    HIPCHK(hipMemcpy2D(g_ctx.kin_stage.p, ld * esz, geno, ld * esz, width, rows, hipMemcpyHostToDevice));
    return gemma_hip_kin_add_d(kind, g_ctx.kin_stage.p, l, ld, nullptr);
  }
  // the upload beside the previous block's kernels: behind that block's ingest (the staging buffer's only reader), in front of
  // this block's kernels on the null stream; the caller's buffer is consumed when the call returns
  if (g_ctx.kin_ingested_valid) HIPCHK(hipStreamWaitEvent(g_ctx.kin_copy, g_ctx.kin_ingested, 0));
  HIPCHK(hipMemcpy2DAsync(g_ctx.kin_stage.p, ld * esz, geno, ld * esz, width, rows, hipMemcpyHostToDevice, g_ctx.kin_copy));
  HIPCHK(hipEventRecord(g_ctx.kin_copied, g_ctx.kin_copy));
  HIPCHK(hipStreamWaitEvent(nullptr, g_ctx.kin_copied, 0));
  g_ctx.kin_host_call = true;
  const int rc = gemma_hip_kin_add_d(kind, g_ctx.kin_stage.p, l, ld, nullptr);
  g_ctx.kin_host_call = false;
  HIPCHK(hipEventSynchronize(g_ctx.kin_copied));
  return rc;
}

extern "C" int gemma_hip_kin_end_d(double *K_d, size_t *ns_used, void *stream) {
  NEED_INIT();
  if (!g_ctx.kin_active) return fail(GEMMA_HIP_ESTATE, "kin_end before kin_begin");
  const size_t n = g_ctx.kin_n;
  hipStream_t s = S(stream);
  {
    int rc = kin_fold_i8(s);
    if (rc) return rc;
  }
  if (ns_used) *ns_used = g_ctx.kin_ns;
  const double scale = g_ctx.kin_ns ? 1.0 / (double)g_ctx.kin_ns : 1.0;
  const unsigned nb = (unsigned)((n + 31) / 32);
  hipLaunchKernelGGL(symm_fill_scale_kernel, dim3(nb, nb), dim3(32, 8), 0, s, g_ctx.kin_K.as<double>(),
                     (long)n, (long)n, scale);
  HIPCHK(hipGetLastError());
  if (K_d) HIPCHK(hipMemcpyAsync(K_d, g_ctx.kin_K.p, n * n * 8, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipStreamSynchronize(s));
  g_ctx.kin_active = false;
  g_ctx.kin_X.release();
  g_ctx.kin_stage.release();
  g_ctx.kin_K.release();
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_kin_end(double *K, size_t *ns_used) {
  NEED_INIT();
  if (!g_ctx.kin_active) return fail(GEMMA_HIP_ESTATE, "kin_end before kin_begin");
  const size_t n = g_ctx.kin_n;
  {
    int rc = kin_fold_i8(nullptr);
    if (rc) return rc;
  }
  if (ns_used) *ns_used = g_ctx.kin_ns;
  const double scale = g_ctx.kin_ns ? 1.0 / (double)g_ctx.kin_ns : 1.0;
  const unsigned nb = (unsigned)((n + 31) / 32);
  hipLaunchKernelGGL(symm_fill_scale_kernel, dim3(nb, nb), dim3(32, 8), 0, 0, g_ctx.kin_K.as<double>(),
                     (long)n, (long)n, scale);
  HIPCHK(hipGetLastError());
  if (K) HIPCHK(hipMemcpy(K, g_ctx.kin_K.p, n * n * 8, hipMemcpyDeviceToHost));
  g_ctx.kin_active = false;
  g_ctx.kin_X.release();
  g_ctx.kin_stage.release();
  g_ctx.kin_K.release();
  return GEMMA_HIP_OK;
}
