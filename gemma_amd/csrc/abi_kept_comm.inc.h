// Part of gemma_hip.hip (ONE translation unit: the parts share the context g_ctx and the helpers of its anonymous namespace, and are
// included there in this order; round 6: the 3 500-line file cut along its stages for reading -- no behaviour change).
// This part: device-resident chain (kin_end_keep -> eigh_kept_K -> lmm_setup_kept), pinned host-block pipeline, the communicator, diagnostics.

// ------------------------------------------------------------------------------ device-resident chain
namespace {
__global__ void subselect_kernel(const double *__restrict__ K, long ni_total, const int *__restrict__ map, long n,
                                 double *__restrict__ G) {
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c < n) G[r * n + c] = K[(long)map[r] * ni_total + map[c]];
}
double *kept_U() { return g_ctx.kept_UE.as<double>(); }
double *kept_eval() { return g_ctx.kept_UE.as<double>() + g_ctx.kept_n * g_ctx.kept_n; }
} // namespace

extern "C" int gemma_hip_kin_end_keep(size_t *ns_used, int allreduce) {
  NEED_INIT();
  if (!g_ctx.kin_active) return fail(GEMMA_HIP_ESTATE, "kin_end before kin_begin");
  const size_t n = g_ctx.kin_n;
  size_t ns = g_ctx.kin_ns;
  {
    int rc = kin_fold_i8(nullptr); // the all-reduce below works on the folded, unscaled upper-triangle sums
    if (rc) return rc;
  }
  if (allreduce && g_ctx.comm.active && g_ctx.comm.world > 1) {
    // SNP-sharded kinship: every rank holds sum_s x_s x_s^T over ITS SNPs (unscaled, upper-triangle tiles); one all-reduce
    // of the n^2 sums and one of the SNP counts, then the common 1/ns scale and the mirror
    std::string err;
    if (g_ctx.comm.allreduce_sum(g_ctx.kin_K.as<double>(), n * n, nullptr, err)) return fail(GEMMA_HIP_ERUNTIME, "%s", err.c_str());
    if (g_ctx.scratch.reserve(16)) return fail(GEMMA_HIP_ENOMEM, "kin_end_keep: scratch");
    double cnt = (double)ns;
    HIPCHK(hipMemcpy(g_ctx.scratch.p, &cnt, 8, hipMemcpyHostToDevice));
    if (g_ctx.comm.allreduce_sum(g_ctx.scratch.as<double>(), 1, nullptr, err)) return fail(GEMMA_HIP_ERUNTIME, "%s", err.c_str());
    HIPCHK(hipMemcpy(&cnt, g_ctx.scratch.p, 8, hipMemcpyDeviceToHost));
    ns = (size_t)(cnt + 0.5);
  }
  if (ns_used) *ns_used = ns;
  const double scale = ns ? 1.0 / (double)ns : 1.0;
  const unsigned nb = (unsigned)((n + 31) / 32);
  hipLaunchKernelGGL(symm_fill_scale_kernel, dim3(nb, nb), dim3(32, 8), 0, 0, g_ctx.kin_K.as<double>(), (long)n, (long)n,
                     scale);
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  g_ctx.kept_K.release();
  g_ctx.kept_K = g_ctx.kin_K; // ownership moves: K stays where the SYRK left it
  g_ctx.kin_K = DevBuf();
  g_ctx.kept_K_n = n;
  g_ctx.kin_active = false;
  g_ctx.kin_X.release();
  g_ctx.kin_stage.release();
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_kept_K_get(double *K) {
  NEED_INIT();
  if (!g_ctx.kept_K_n) return fail(GEMMA_HIP_ESTATE, "kept_K_get: no kept K");
  if (!K) return fail(GEMMA_HIP_EINVAL, "kept_K_get: null pointer");
  HIPCHK(hipMemcpy(K, g_ctx.kept_K.p, g_ctx.kept_K_n * g_ctx.kept_K_n * 8, hipMemcpyDeviceToHost));
  return GEMMA_HIP_OK;
}

static int kept_alloc_ue(size_t n) {
  if (g_ctx.kept_UE.reserve((n * n + n) * 8)) return fail(GEMMA_HIP_ENOMEM, "kept U: %zu bytes", (n * n + n) * 8);
  g_ctx.kept_n = n;
  return GEMMA_HIP_OK;
}

static int kept_eigh_of(double *G_d, size_t n, double *eval, double *trace_G, bool sharded = false) {
  int rc = kept_alloc_ue(n);
  if (rc) {
    eigh_abort_if_sharded(sharded, n, nullptr);
    return rc;
  }
  double tr = 0.0;
  rc = eigh_d_impl(G_d, n, kept_U(), kept_eval(), &tr, nullptr, sharded);
  if (rc) {
    g_ctx.kept_n = 0;
    return rc;
  }
  g_ctx.kept_trace = tr;
  if (trace_G) *trace_G = tr;
  if (eval) HIPCHK(hipMemcpy(eval, kept_eval(), n * 8, hipMemcpyDeviceToHost));
  return GEMMA_HIP_OK;
}

static int eigh_kept_K_impl(const int *indicator_idv, size_t ni_total, double *eval, double *trace_G, bool sharded);
extern "C" int gemma_hip_eigh_kept_K(const int *indicator_idv, size_t ni_total, double *eval, double *trace_G) {
  return eigh_kept_K_impl(indicator_idv, ni_total, eval, trace_G, false);
}
// COLLECTIVE: every rank holds the same kept K (kin_end_keep with the all-reduce) and ends with the same kept (U, eval) --
// no gemma_hip_kept_bcast afterwards
extern "C" int gemma_hip_eigh_kept_K_sharded(const int *indicator_idv, size_t ni_total, double *eval, double *trace_G) {
  return eigh_kept_K_impl(indicator_idv, ni_total, eval, trace_G, true);
}
static int eigh_kept_K_impl(const int *indicator_idv, size_t ni_total, double *eval, double *trace_G, bool sharded) {
  NEED_INIT();
  if (!g_ctx.kept_K_n) return fail(GEMMA_HIP_ESTATE, "eigh_kept_K: no kept K (kin_end_keep first)");
  if (ni_total != g_ctx.kept_K_n) return fail(GEMMA_HIP_EINVAL, "eigh_kept_K: ni_total=%zu, kept K is %zu", ni_total, g_ctx.kept_K_n);
  std::vector<int> map;
  for (size_t i = 0; i < ni_total; ++i)
    if (!indicator_idv || indicator_idv[i] != 0) map.push_back((int)i);
  const size_t n = map.size();
  if (n == 0) return fail(GEMMA_HIP_EINVAL, "eigh_kept_K: no analysed individual");
  DevBuf G, dmap;
  if (G.reserve(n * n * 8) || dmap.reserve(n * sizeof(int))) {
    G.release(); dmap.release();
    eigh_abort_if_sharded(sharded, n, nullptr);
    return fail(GEMMA_HIP_ENOMEM, "eigh_kept_K: %zu bytes", n * n * 8);
  }
  int rc = GEMMA_HIP_OK;
  hipError_t e = hipMemcpy(dmap.p, map.data(), n * sizeof(int), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    // the rows / columns ReadFile_kin keeps (src/gemma_io.cpp:1205-1243), then CenterMatrix, then EigenDecomp_Zeroed
    hipLaunchKernelGGL(subselect_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, 0,
                       g_ctx.kept_K.as<double>(), (long)ni_total, dmap.as<int>(), (long)n, G.as<double>());
    e = hipGetLastError();
  }
  if (e == hipSuccess) rc = gemma_hip_center_d(G.as<double>(), n, nullptr);
  if (e == hipSuccess && rc == GEMMA_HIP_OK) rc = kept_eigh_of(G.as<double>(), n, eval, trace_G, sharded);
  else eigh_abort_if_sharded(sharded, n, nullptr); // sub-selection or centring failed on this rank alone
  G.release(); dmap.release();
  if (e != hipSuccess) return fail(GEMMA_HIP_ERUNTIME, "eigh_kept_K: %s", hipGetErrorString(e));
  return rc;
}

extern "C" int gemma_hip_eigh_keep(const double *G, size_t n, double *eval, double *trace_G) {
  NEED_INIT();
  if (!G || n == 0) return fail(GEMMA_HIP_EINVAL, "eigh_keep: null/empty argument");
  DevBuf dG;
  if (dG.reserve(n * n * 8)) return fail(GEMMA_HIP_ENOMEM, "eigh_keep: %zu bytes", n * n * 8);
  hipError_t e = hipMemcpy(dG.p, G, n * n * 8, hipMemcpyHostToDevice);
  int rc = GEMMA_HIP_OK;
  if (e == hipSuccess) rc = kept_eigh_of(dG.as<double>(), n, eval, trace_G);
  dG.release();
  if (e != hipSuccess) return fail(GEMMA_HIP_ERUNTIME, "eigh_keep: %s", hipGetErrorString(e));
  return rc;
}

extern "C" int gemma_hip_kept_n(size_t *n) {
  if (n) *n = g_ctx.kept_n;
  return GEMMA_HIP_OK;
}

// ONE ncclBroadcast of (U, eval) -- they share a buffer -- after a 16-byte header {n, trace_G} that tells the other
// ranks what to allocate
extern "C" int gemma_hip_kept_bcast(int root, double *trace_G) {
  NEED_INIT();
  Comm &cm = g_ctx.comm;
  if (!cm.active || cm.world == 1) {
    if (trace_G && g_ctx.kept_n) *trace_G = g_ctx.kept_trace;
    return GEMMA_HIP_OK;
  }
  if (root < 0 || root >= cm.world) return fail(GEMMA_HIP_EINVAL, "kept_bcast: root %d of %d", root, cm.world);
  if (cm.rank == root && !g_ctx.kept_n) return fail(GEMMA_HIP_ESTATE, "kept_bcast: the root holds no kept U");
  if (g_ctx.scratch.reserve(16)) return fail(GEMMA_HIP_ENOMEM, "kept_bcast: scratch");
  double hdr[2] = {(double)g_ctx.kept_n, g_ctx.kept_trace};
  std::string err;
  if (cm.rank == root) HIPCHK(hipMemcpy(g_ctx.scratch.p, hdr, 16, hipMemcpyHostToDevice));
  if (cm.bcast(g_ctx.scratch.p, 16, root, nullptr, err)) return fail(GEMMA_HIP_ERUNTIME, "%s", err.c_str());
  HIPCHK(hipMemcpy(hdr, g_ctx.scratch.p, 16, hipMemcpyDeviceToHost)); // synchronises with the broadcast on the null stream
  const size_t n = (size_t)(hdr[0] + 0.5);
  if (cm.rank != root) {
    int rc = kept_alloc_ue(n);
    if (rc) return rc;
    g_ctx.kept_trace = hdr[1];
  }
  if (cm.bcast(g_ctx.kept_UE.p, (n * n + n) * 8, root, nullptr, err)) return fail(GEMMA_HIP_ERUNTIME, "%s", err.c_str());
  HIPCHK(hipDeviceSynchronize());
  if (trace_G) *trace_G = g_ctx.kept_trace;
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_kept_U_get(double *U, double *eval) {
  NEED_INIT();
  if (!g_ctx.kept_n) return fail(GEMMA_HIP_ESTATE, "kept_U_get: no kept U");
  const size_t n = g_ctx.kept_n;
  if (U) HIPCHK(hipMemcpy(U, kept_U(), n * n * 8, hipMemcpyDeviceToHost));
  if (eval) HIPCHK(hipMemcpy(eval, kept_eval(), n * 8, hipMemcpyDeviceToHost));
  return GEMMA_HIP_OK;
}

// CalcUtX (src/mathfunc.cpp:504-506) on the kept U: UtX (n x m) = U^T X, X and UtX on the host
extern "C" int gemma_hip_calc_utx_kept(const double *X, size_t n, size_t m, double *UtX) {
  NEED_INIT();
  if (!g_ctx.kept_n) return fail(GEMMA_HIP_ESTATE, "calc_utx_kept: no kept U");
  if (n != g_ctx.kept_n || !X || !UtX || m == 0) return fail(GEMMA_HIP_EINVAL, "calc_utx_kept: n=%zu, kept U is %zu", n, g_ctx.kept_n);
  DevBuf dX, dO;
  if (dX.reserve(n * m * 8) || dO.reserve(n * m * 8)) {
    dX.release(); dO.release();
    return fail(GEMMA_HIP_ENOMEM, "calc_utx_kept: %zu bytes", 2 * n * m * 8);
  }
  hipError_t e = hipMemcpy(dX.p, X, n * m * 8, hipMemcpyHostToDevice);
  int rc = GEMMA_HIP_OK;
  if (e == hipSuccess)
    rc = gemma_hip_dgemm_d('T', 'N', n, m, n, 1.0, kept_U(), n, dX.as<double>(), m, 0.0, dO.as<double>(), m, nullptr);
  if (e == hipSuccess && rc == GEMMA_HIP_OK) e = hipMemcpy(UtX, dO.p, n * m * 8, hipMemcpyDeviceToHost);
  dX.release(); dO.release();
  if (e != hipSuccess) return fail(GEMMA_HIP_ERUNTIME, "calc_utx_kept: %s", hipGetErrorString(e));
  return rc;
}

extern "C" int gemma_hip_lmm_setup_kept(const gemma_lmm_cfg *cfg, const double *UtW, const double *Uty) {
  NEED_INIT();
  if (!g_ctx.kept_n) return fail(GEMMA_HIP_ESTATE, "lmm_setup_kept: no kept U");
  if (!cfg || !UtW || !Uty) return fail(GEMMA_HIP_EINVAL, "lmm_setup_kept: null pointer");
  if (cfg->n != g_ctx.kept_n) return fail(GEMMA_HIP_EINVAL, "lmm_setup_kept: cfg.n=%zu, kept U is %zu", cfg->n, g_ctx.kept_n);
  int rc = lmm_common_setup(cfg);
  if (rc) return rc;
  const size_t n = cfg->n, c = cfg->n_cvt;
  if (g_ctx.own_Uty.reserve(n * 8) || g_ctx.own_UtW.reserve(n * c * 8)) return fail(GEMMA_HIP_ENOMEM, "lmm_setup_kept");
  HIPCHK(hipMemcpy(g_ctx.own_Uty.p, Uty, n * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(g_ctx.own_UtW.p, UtW, n * c * 8, hipMemcpyHostToDevice));
  g_ctx.U = kept_U();
  g_ctx.U_even_of = nullptr;
  g_ctx.eval = kept_eval();
  g_ctx.Uty = g_ctx.own_Uty.as<double>();
  rc = make_utwt(g_ctx.own_UtW.as<double>(), 0);
  if (rc) return rc;
  HIPCHK(hipDeviceSynchronize());
  g_ctx.lmm_active = true;
  return GEMMA_HIP_OK;
}

// ---- pipelined host blocks --------------------------------------------------------------------------------------
static void pipe_release() {
  for (auto &p : g_ctx.pipe) {
    if (p.pin_in) (void)hipHostFree(p.pin_in);
    if (p.pin_out) (void)hipHostFree(p.pin_out);
    p.pin_in = p.pin_out = nullptr;
    p.pin_in_cap = p.pin_out_cap = 0;
    p.dev_in.release(); p.dev_out.release();
    if (p.h2d) (void)hipEventDestroy(p.h2d);
    if (p.done) (void)hipEventDestroy(p.done);
    p.h2d = p.done = nullptr;
    p.busy = false;
  }
  if (g_ctx.pipe_copy) (void)hipStreamDestroy(g_ctx.pipe_copy);
  if (g_ctx.pipe_comp) (void)hipStreamDestroy(g_ctx.pipe_comp);
  g_ctx.pipe_copy = g_ctx.pipe_comp = nullptr;
  g_ctx.pipe_head = g_ctx.pipe_count = 0;
}

extern "C" int gemma_hip_lmm_batch_submit(int kind, const void *geno, size_t l, size_t ld) {
  NEED_INIT();
  if (!g_ctx.lmm_active) return fail(GEMMA_HIP_ESTATE, "lmm_batch_submit before lmm_setup");
  if (g_ctx.pipe_count >= 2) return fail(GEMMA_HIP_ESTATE, "lmm_batch_submit: two blocks already in flight (collect first)");
  if (l == 0) return fail(GEMMA_HIP_EINVAL, "lmm_batch_submit: empty block");
  int dummy = 0;
  int rc = check_batch_args("lmm_batch_submit", kind, geno, l, ld, &dummy);
  if (rc) return rc;
  if (!g_ctx.pipe_copy) {
    HIPCHK(hipStreamCreateWithFlags(&g_ctx.pipe_copy, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&g_ctx.pipe_comp, hipStreamNonBlocking));
  }
  const int slot = (g_ctx.pipe_head + g_ctx.pipe_count) & 1;
  Ctx::PipeSlot &p = g_ctx.pipe[slot];
  const size_t esz = (kind == GEMMA_GENO_PLINK_2BIT) ? 1 : 8;
  const size_t rows = (kind == GEMMA_GENO_F64_IDV_MAJOR) ? g_ctx.cfg.n : l;
  const size_t n = g_ctx.cfg.n;
  const size_t per_row = (kind == GEMMA_GENO_PLINK_2BIT && g_ctx.have_map) ? g_ctx.ni_total : n;
  const size_t need = min_ld_for(kind, per_row, l);
  const size_t bytes = ((rows - 1) * ld + need) * esz; // the last row may be shorter than ld in the caller's buffer
  if (p.pin_in_cap < rows * ld * esz) {
    if (p.pin_in) (void)hipHostFree(p.pin_in);
    p.pin_in = nullptr;
    p.pin_in_cap = 0;
    HIPCHK(hipHostMalloc(&p.pin_in, rows * ld * esz, hipHostMallocDefault));
    p.pin_in_cap = rows * ld * esz;
  }
  if (p.pin_out_cap < l * sizeof(gemma_sumstat)) {
    if (p.pin_out) (void)hipHostFree(p.pin_out);
    p.pin_out = nullptr;
    p.pin_out_cap = 0;
    HIPCHK(hipHostMalloc(&p.pin_out, l * sizeof(gemma_sumstat), hipHostMallocDefault));
    p.pin_out_cap = l * sizeof(gemma_sumstat);
  }
  if (p.dev_in.reserve(rows * ld * esz) || p.dev_out.reserve(l * sizeof(gemma_sumstat)))
    return fail(GEMMA_HIP_ENOMEM, "lmm_batch_submit: staging %zu bytes", rows * ld * esz);
  if (!p.h2d) {
    HIPCHK(hipEventCreateWithFlags(&p.h2d, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&p.done, hipEventDisableTiming));
  }
  memcpy(p.pin_in, geno, bytes);
  HIPCHK(hipMemcpyAsync(p.dev_in.p, p.pin_in, bytes, hipMemcpyHostToDevice, g_ctx.pipe_copy));
  HIPCHK(hipEventRecord(p.h2d, g_ctx.pipe_copy));
  HIPCHK(hipStreamWaitEvent(g_ctx.pipe_comp, p.h2d, 0));
  rc = gemma_hip_lmm_batch_d(kind, p.dev_in.p, l, ld, p.dev_out.as<gemma_sumstat>(), g_ctx.pipe_comp);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(p.pin_out, p.dev_out.p, l * sizeof(gemma_sumstat), hipMemcpyDeviceToHost, g_ctx.pipe_comp));
  HIPCHK(hipEventRecord(p.done, g_ctx.pipe_comp));
  p.l = l;
  p.busy = true;
  g_ctx.pipe_count += 1;
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_lmm_batch_collect(gemma_sumstat *out, size_t *l) {
  NEED_INIT();
  if (g_ctx.pipe_count == 0) return fail(GEMMA_HIP_ESTATE, "lmm_batch_collect: nothing in flight");
  if (!out) return fail(GEMMA_HIP_EINVAL, "lmm_batch_collect: null pointer");
  Ctx::PipeSlot &p = g_ctx.pipe[g_ctx.pipe_head];
  HIPCHK(hipEventSynchronize(p.done));
  memcpy(out, p.pin_out, p.l * sizeof(gemma_sumstat));
  if (l) *l = p.l;
  p.busy = false;
  g_ctx.pipe_head ^= 1;
  g_ctx.pipe_count -= 1;
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_kept_release(void) {
  if (g_ctx.lmm_active && g_ctx.U == kept_U() && g_ctx.kept_n)
    return fail(GEMMA_HIP_ESTATE, "kept_release: the LMM state borrows the kept U (lmm_finish first)");
  g_ctx.kept_K.release();
  g_ctx.kept_UE.release();
  g_ctx.kept_K_n = g_ctx.kept_n = 0;
  return GEMMA_HIP_OK;
}

// ------------------------------------------------------------------------------ multi-GPU: RCCL (csrc/comm.hip.h)
extern "C" int gemma_hip_comm_unique_id(void *id) {
  if (!id) return fail(GEMMA_HIP_EINVAL, "comm_unique_id: null pointer");
  std::string err;
  if (g_ctx.comm.unique_id(id, err)) return fail(GEMMA_HIP_ERUNTIME, "%s", err.c_str());
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_comm_init(const void *id, int rank, int world) {
  NEED_INIT();
  if (world < 1 || rank < 0 || rank >= world) return fail(GEMMA_HIP_EINVAL, "comm_init: rank %d of %d", rank, world);
  std::string err;
  if (g_ctx.comm.init(id, rank, world, err)) return fail(GEMMA_HIP_ERUNTIME, "%s", err.c_str());
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_comm_info(int *rank, int *world, int *transport) {
  const Comm &cm = g_ctx.comm;
  if (rank) *rank = cm.active ? cm.rank : 0;
  if (world) *world = cm.active ? cm.world : 1;
  if (transport) *transport = (!cm.active || cm.world == 1) ? 0 : (cm.shm ? 2 : 1);
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_comm_bcast_d(void *buf_d, size_t bytes, int root, void *stream) {
  NEED_INIT();
  if (!buf_d && bytes) return fail(GEMMA_HIP_EINVAL, "comm_bcast: null pointer");
  std::string err;
  if (g_ctx.comm.bcast(buf_d, bytes, root, S(stream), err)) return fail(GEMMA_HIP_ERUNTIME, "%s", err.c_str());
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_comm_allreduce_sum_d(double *buf_d, size_t count, void *stream) {
  NEED_INIT();
  if (!buf_d && count) return fail(GEMMA_HIP_EINVAL, "comm_allreduce: null pointer");
  std::string err;
  if (g_ctx.comm.allreduce_sum(buf_d, count, S(stream), err)) return fail(GEMMA_HIP_ERUNTIME, "%s", err.c_str());
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_comm_finalize(void) {
  g_ctx.comm.finalize();
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_comm_selftest(void *stream) {
  NEED_INIT();
  std::string err;
  if (g_ctx.comm.selftest(S(stream), err)) return fail(GEMMA_HIP_ERUNTIME, "%s", err.c_str());
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_comm_stats(gemma_comm_stats *out) {
  if (!out) return fail(GEMMA_HIP_EINVAL, "comm_stats: null");
  const CommStats &st = g_ctx.comm.stats;
  out->allreduce_calls = st.allreduce_calls;
  out->allreduce_pieces = st.allreduce_pieces;
  out->bcast_calls = st.bcast_calls;
  out->bcast_pieces = st.bcast_pieces;
  out->allreduce_bytes = st.allreduce_bytes;
  out->bcast_bytes = st.bcast_bytes;
  out->allreduce_s = st.allreduce_s;
  out->bcast_s = st.bcast_s;
  return GEMMA_HIP_OK;
}


extern "C" int gemma_hip_dbg_last_utx_kernel(gemma_utx_kernel_info *info) {
  if (!info) return fail(GEMMA_HIP_EINVAL, "dbg_last_utx_kernel: null");
  *info = g_ctx.last_utx_kernel;
  return GEMMA_HIP_OK;
}

// the any-missing flag sparse2_meta_kernel left for the last records product of a plain (not pipelined) batch: 1 / 0, -1 when the
// complete-block form is off or no such product ran.  Synchronises the device.
extern "C" int gemma_hip_dbg_last_block_missing(int *any) {
  NEED_INIT();
  if (!any) return fail(GEMMA_HIP_EINVAL, "dbg_last_block_missing: null");
  *any = -1;
  if (g_ctx.i8_flag_at < 0 || !g_ctx.i8_rowsur.p) return GEMMA_HIP_OK;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(any, g_ctx.i8_rowsur.as<int>() + g_ctx.i8_flag_at, sizeof(int), hipMemcpyDeviceToHost));
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_reload_env(void) {
  const int digits0 = i8_digits_for(g_ctx.cfg.n), scale0 = g_ctx.knobs.i8_scale_max;
  g_ctx.knobs.load();
  // switches the digit planes of U were cut under: a change of the digit count or of the column scaling makes the next
  // batch cut them again instead of multiplying planes of the old form
  if (g_ctx.i8_ready && (i8_digits_for(g_ctx.cfg.n) != digits0 || g_ctx.knobs.i8_scale_max != scale0))
    g_ctx.i8_ready = g_ctx.i8_colsum_ready = false;
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_dbg_last_utx_path(int *path) {
  if (path) *path = g_ctx.last_utx_path;
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_dbg_i8_digits(size_t n, int *digits) {
  if (digits) *digits = i8_digits_for(n);
  return GEMMA_HIP_OK;
}
