// Part of gemma_hip.hip (ONE translation unit: the parts share the context g_ctx and the helpers of its anonymous namespace, and are
// included there in this order; round 6: the 3 500-line file cut along its stages for reading -- no behaviour change).
// This part: batch entry points: the two-block pipeline, gemma_hip_lmm_batch_d, -gene.

// ---- stream / buffer plumbing of gemma_hip_lmm_batch_pipe_d (the two-block pipeline, described where that entry point is defined)
static void xp_release() {
  Ctx::XPipe &x = g_ctx.xp;
  if (x.P) (void)hipStreamDestroy(x.P);
  if (x.Q) (void)hipStreamDestroy(x.Q);
  if (x.in_ready) (void)hipEventDestroy(x.in_ready);
  if (x.ingest_done) (void)hipEventDestroy(x.ingest_done);
  x.ingest_done = nullptr;
  x.ingest_valid = false;
  for (int i = 0; i < 2; ++i) {
    if (x.prod_done[i]) (void)hipEventDestroy(x.prod_done[i]);
    if (x.post_done[i]) (void)hipEventDestroy(x.post_done[i]);
    x.prod_done[i] = x.post_done[i] = nullptr;
    x.post_valid[i] = false;
  }
  x.P = x.Q = nullptr;
  x.in_ready = nullptr;
  x.count = 0;
  x.pending = false;
  x.cus = -1;
  x.shadow_A.release(); x.shadow_C.release(); x.shadow_mean.release(); x.shadow_rowsur.release();
}
static int xp_init() {
  Ctx::XPipe &x = g_ctx.xp;
  const int ncu = g_ctx.prop.multiProcessorCount;
  int cus = g_ctx.knobs.pipe_cus;
  if (cus < 0 || cus * 2 > ncu || ncu % 32 != 0 || cus % 8 != 0 || (cus && (ncu / 8) % (cus / 8) != 0)) cus = 0;
  if (x.P && x.cus == cus) return GEMMA_HIP_OK;
  HIPCHK(hipDeviceSynchronize());
  xp_release();
  if (cus > 0) {
    // Mask bit c = CU c / 8 of XCD c % 8 (scripts/xcc_mask_probe.hip, profiles/r05_pipeline_partition.txt), and a mask that leaves an
    // XCD WITHOUT CUs is not applied at all (the stream then runs on every CU) -- so the post partition takes the same cus / 8 CUs
    // out of EVERY XCD, evenly spaced over its 32 (an uneven cut lets the dispatcher's round over the XCDs wait for the short one:
    // 16 CUs taken from one XCD cost the product 75 %).
    const int words = ncu / 32, per_xcd = cus / 8, cu_per_xcd = ncu / 8, stepj = cu_per_xcd / per_xcd;
    std::vector<unsigned> mp((size_t)words, 0xFFFFFFFFu), mq((size_t)words, 0u);
    for (int j = 0; j < cu_per_xcd; j += stepj)
      for (int xcd = 0; xcd < 8; ++xcd) {
        const int c = 8 * j + xcd;
        mp[c >> 5] &= ~(1u << (c & 31));
        mq[c >> 5] |= 1u << (c & 31);
      }
    HIPCHK(hipExtStreamCreateWithCUMask(&x.P, (uint32_t)words, mp.data()));
    HIPCHK(hipExtStreamCreateWithCUMask(&x.Q, (uint32_t)words, mq.data()));
  } else {
    // blocking streams like the masked ones: ordered behind the legacy default stream without an event (see lmm_batch_pipe_d)
    HIPCHK(hipStreamCreateWithFlags(&x.P, hipStreamDefault));
    HIPCHK(hipStreamCreateWithFlags(&x.Q, hipStreamDefault));
  }
  HIPCHK(hipEventCreateWithFlags(&x.in_ready, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&x.ingest_done, hipEventDisableTiming));
  for (int i = 0; i < 2; ++i) {
    HIPCHK(hipEventCreateWithFlags(&x.prod_done[i], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&x.post_done[i], hipEventDisableTiming));
  }
  x.cus = cus;
  return GEMMA_HIP_OK;
}
// everything the pipeline still has in flight is ordered in front of whatever the caller puts on s next
static int xp_flush(hipStream_t s) {
  Ctx::XPipe &x = g_ctx.xp;
  if (!x.pending) return GEMMA_HIP_OK;
  for (int i = 0; i < 2; ++i)
    if (x.post_valid[i]) HIPCHK(hipStreamWaitEvent(s, x.post_done[i], 0));
  x.pending = false;
  x.ingest_valid = false; // every block's post stage (behind its ingest on P, through prod_done) is now in front of s
  return GEMMA_HIP_OK;
}
// A pipelined call that fails after it has switched buffer sets leaves the slot parity and the set parity out of step:
// wait for everything in flight and start the pipeline over (slot 0, nothing to wait for); the error is the caller's to report.
static int xp_abort(int rc) {
  Ctx::XPipe &x = g_ctx.xp;
  (void)hipDeviceSynchronize(); // both buffer sets are idle from here on: which of them is "live" no longer matters
  x.count = 0;
  x.post_valid[0] = x.post_valid[1] = false;
  x.pending = false;
  x.ingest_valid = false;
  return rc;
}
static void xp_swap_sets() {
  Ctx::XPipe &x = g_ctx.xp;
  std::swap(g_ctx.i8_A, x.shadow_A); std::swap(g_ctx.i8_C, x.shadow_C);
  std::swap(g_ctx.i8_mean, x.shadow_mean); std::swap(g_ctx.i8_rowsur, x.shadow_rowsur);
}


// UtX (l x ldx, SNP-major) = mean-imputed X (l x n) * U (n x n): row s is (U^T x_s)^T, i.e. the column the
// reference's fast_dgemm("T","N",U,Xlarge) (src/lmm.cpp:1521) produces for SNP s.  path < 0: by GEMMA_HIP_UTX_I8.
static int compute_utx(int kind, const void *geno, size_t l, size_t ld, int path, double **UtX_out, size_t *ldx_out,
                       hipStream_t s) {
  {
    int rcf = xp_flush(s); // blocks of gemma_hip_lmm_batch_pipe_d still in flight share this call's buffers
    if (rcf) return rcf;
  }
  const size_t n = g_ctx.cfg.n;
  const size_t ldx = (n + 1) & ~(size_t)1;
  const bool want_i8 = (path < 0 ? utx_i8_mode() == 1 : path == 1);
  const bool plink_i8 = want_i8 && kind == GEMMA_GENO_PLINK_2BIT;
  if (g_ctx.UtX.reserve(l * ldx * 8) || (!plink_i8 && g_ctx.X.reserve(l * ldx * 8)))
    return fail(GEMMA_HIP_ENOMEM, "lmm_batch: cannot allocate 2 x %zu bytes", l * ldx * 8);
  double *UtX = g_ctx.UtX.as<double>();
  *UtX_out = UtX;
  *ldx_out = ldx;
  g_ctx.last_utx_path = plink_i8 ? 1 : 0;
  if (plink_i8) return utx_plink_i8(geno, l, ld, UtX, ldx, s);
  double *X = g_ctx.X.as<double>();
  bool done = false;
  if (kind == GEMMA_GENO_F64_IDV_MAJOR) {
    { // the reference's Xlarge (individuals x SNPs, already mean-imputed) -> SNP-major
      ProfScope ps(GEMMA_STAGE_INGEST, s);
      dim3 grid((unsigned)((l + 31) / 32), (unsigned)((n + 31) / 32));
      hipLaunchKernelGGL(transpose_kernel, grid, dim3(32, 8), 0, s, reinterpret_cast<const double *>(geno),
                         (long)n, (long)l, (long)ld, X, (long)ldx);
      HIPCHK(hipGetLastError());
    }
    if (want_i8) { // hard calls with one imputed value per SNP take the exact int8-digit product as well
      int rc = utx_f64_try_i8(X, l, ldx, false, UtX, ldx, s, &done);
      if (rc) return rc;
    }
  } else if (kind == GEMMA_GENO_F64_SNP_MAJOR && want_i8) {
    int rc = utx_f64_try_i8(reinterpret_cast<const double *>(geno), l, ld, true, UtX, ldx, s, &done);
    if (rc) return rc;
  }
  if (done) return GEMMA_HIP_OK;
  if (kind != GEMMA_GENO_F64_IDV_MAJOR) {
    ProfScope ps(GEMMA_STAGE_INGEST, s);
    IngestArgs a;
    a.src = geno; a.ld = (long)ld; a.l = (long)l;
    a.idx_map = g_ctx.have_map ? g_ctx.idx_map.as<int>() : nullptr;
    a.n = (int)n; a.dst = X; a.ldo = (long)ldx; a.k_mode = 0;
    const unsigned grid = (unsigned)((l + 3) / 4);
    if (kind == GEMMA_GENO_PLINK_2BIT)
      hipLaunchKernelGGL(ingest_lmm_kernel<true>, dim3(grid), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL(ingest_lmm_kernel<false>, dim3(grid), dim3(256), 0, s, a);
    HIPCHK(hipGetLastError());
  }
  {
    ProfScope ps(GEMMA_STAGE_UTX_GEMM, s);
    const double *Ug;
    long ldu;
    int rcu = gemm_U(&Ug, &ldu, s);
    if (rcu) return rcu;
    note_utx_kernel(GEMMA_UTX_KERNEL_DGEMM_F64, 0, 0, 0);
    HIPCHK(launch_dgemm('N', 'N', (long)l, (long)n, (long)n, 1.0, X, (long)ldx, Ug, ldu, 0.0, UtX,
                        (long)ldx, false, false, s));
  }
  return GEMMA_HIP_OK;
}

static int check_batch_args(const char *who, int kind, const void *geno, size_t l, size_t ld, const void *out) {
  const size_t n = g_ctx.cfg.n;
  const size_t per_row = (kind == GEMMA_GENO_PLINK_2BIT && g_ctx.have_map) ? g_ctx.ni_total : n;
  const size_t need = min_ld_for(kind, per_row, l);
  if (need == (size_t)-1) return fail(GEMMA_HIP_EINVAL, "%s: unknown geno_kind %d", who, kind);
  if (!geno || !out || ld < need) return fail(GEMMA_HIP_EINVAL, "%s: ld=%zu < %zu", who, ld, need);
  return GEMMA_HIP_OK;
}

// PLINK blocks on the records kernel, in row chunks on two streams.  The int8 product is bound by the matrix pipe (and by
// power), the digit combine and the per-SNP stage by HBM and latency: 6 of a step's 64 ms at n = B = 20 000 that leave the
// matrix pipe idle.  The block is cut into `chunks` pieces of whole 256-row tiles; the caller's stream runs ingest + records for
// the block and then the products of the chunks back to back, the side stream runs combine + association of chunk c as soon as
// its product is done -- beside the product of chunk c + 1 (a product workgroup leaves 32 KiB of LDS and 24 wavefront slots per
// CU free).  Every buffer is partitioned by SNP rows (planes, UtX, records, lists, the output), the per-SNP stage's scratch is
// reused chunk after chunk in side-stream order, and the caller's stream waits for the side stream before the call returns
// control of it: the call has the semantics it had.
// MEASURED (round 3, n = B = 20 000, profiles/r03_overlap_two_streams.txt): it does not pay.  One stream 62.7 ms per step
// (product 55.7, combine 2.8, per-SNP stage 3.2); four chunks on two streams 64.2 ms -- the product takes 61.1 ms with the side
// stream's kernels among its workgroups (every CU slot and every watt they take is the product's), the per-SNP stage 10.5 ms;
// two chunks 63.3, eight 64.0.  The chip is at its power limit under the product alone, so concurrency is a zero-sum game
// here.  The path stays behind GEMMA_HIP_OVERLAP=1 (GEMMA_HIP_OVERLAP_CHUNKS, default 4), off by default, with its test.
static int overlap_chunks(size_t l) {
  if (!g_ctx.knobs.overlap) return 1;
  if (utx_i8_mode() != 1 || i8_sparse_mode() != 2) return 1;
  int q = g_ctx.knobs.overlap_chunks;
  q = std::max(1, std::min(q, 16));
  while (q > 1 && l < (size_t)q * 2 * S2_BM) --q; // at least two tile rows per chunk
  return q;
}
static int overlap_init() {
  if (g_ctx.ov_stream) return GEMMA_HIP_OK;
  HIPCHK(hipStreamCreateWithFlags(&g_ctx.ov_stream, hipStreamNonBlocking));
  for (auto &e : g_ctx.ov_ready) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&g_ctx.ov_done, hipEventDisableTiming));
  return GEMMA_HIP_OK;
}
static int lmm_batch_plink_chunked(const void *geno, size_t l, size_t ld, gemma_sumstat *out_d, int chunks, hipStream_t s) {
  const size_t n = g_ctx.cfg.n;
  const size_t ldx = (n + 1) & ~(size_t)1;
  int rc = overlap_init();
  if (rc) return rc;
  if (g_ctx.UtX.reserve(l * ldx * 8)) return fail(GEMMA_HIP_ENOMEM, "lmm_batch: cannot allocate %zu bytes", l * ldx * 8);
  double *UtX = g_ctx.UtX.as<double>();
  g_ctx.last_utx_path = 1;
  I8Dims d;
  if ((rc = i8_begin(l, &d, s))) return rc;
  {
    ProfScope ps(GEMMA_STAGE_INGEST, s);
    IngestI8Args a;
    a.src = reinterpret_cast<const unsigned char *>(geno); a.ld = (long)ld; a.l = (long)l;
    a.idx_map = g_ctx.have_map ? g_ctx.idx_map.as<int>() : nullptr;
    a.n = (int)d.n; a.A = g_ctx.i8_A.as<int8_t>(); a.ldk = (long)d.ldk;
    a.mean = g_ctx.i8_mean.as<double>();
    hipLaunchKernelGGL(ingest_i8_kernel, dim3((unsigned)((l + 3) / 4)), dim3(256), 0, s, a);
    HIPCHK(hipGetLastError());
  }
  if ((rc = i8_meta_build(d, s))) return rc;
  const size_t per = round_up((l + chunks - 1) / chunks, (size_t)S2_BM);
  hipStream_t side = g_ctx.ov_stream;
  int c = 0;
  for (size_t row0 = 0; row0 < l; row0 += per, ++c) {
    const size_t rows = std::min(per, l - row0), rows_pad = std::min(per, d.lpad - row0);
    if ((rc = i8_gemm_rows(d, row0, rows_pad, s))) break;
    if (hipEventRecord(g_ctx.ov_ready[c], s) != hipSuccess || hipStreamWaitEvent(side, g_ctx.ov_ready[c], 0) != hipSuccess) {
      rc = fail(GEMMA_HIP_ERUNTIME, "lmm_batch: %s", hipGetErrorString(hipGetLastError())); // and join below, as on every path
      break;
    }
    if ((rc = i8_post_rows(l, d, row0, rows, UtX, ldx, side))) break;
    if ((rc = launch_assoc(UtX + row0 * ldx, rows, ldx, out_d + row0, side))) break;
  }
  // whatever happened, the caller's stream is ordered behind the side stream again before this call hands it back
  (void)hipEventRecord(g_ctx.ov_done, side);
  (void)hipStreamWaitEvent(s, g_ctx.ov_done, 0);
  return rc;
}

// ---- two blocks in flight on a CU partition (round 5) -----------------------------------------------------------------------
// A step of the PLINK path is the int8 product (50 ms at n = B = 20 000: matrix pipe, power) followed by the digit combine and the
// per-SNP stage (5.6 ms: HBM and latency, the matrix pipe idle).  Side by side on ALL CUs they only take each other's slots and watts
// (round 3: 62.7 -> 64.2 ms, overlap_chunks above).  This entry point puts them on a PARTITION of the CUs
// (hipExtStreamCreateWithCUMask): block i + 1's ingest, records and product on stream P (all but GEMMA_HIP_PIPE_CUS CUs, the same
// number taken out of every XCD) while block i's combine and per-SNP stage run on stream Q (those CUs):
//   caller's stream s --in_ready--> P: [wait post_done(i - 1: same buffer set)] ingest, records, product --prod_done(i)--> Q: combine,
//   per-SNP stage --post_done(i)--> (flush: s waits for the last one)
// Every result is the one gemma_hip_lmm_batch_d gives, bit for bit (same kernels, same launch shapes; the PLINK carry chain runs in
// block order on Q): tests/test_gpu_parity.py::test_lmm_pipe_blocks_equal_plain_batches.
// MEASURED (round 5, n = B = 20 000, profiles/r05_pipeline_partition.txt): IT DOES NOT PAY ON THIS PART, so bench.py times the
// one-stream step (--pipeline 0) and this stays an option.  The records kernel on 224 CUs (4 out of every XCD) takes 55.2 ms
// against 50.4 on 256 (the clock gained from the smaller power draw gives back a third of the 8 / 7), the 32 CUs need 13.6 ms for the
// traffic of the stages behind it: 55.8 ms per step against 56.4.  Without a partition (two plain streams, or a mask that the
// runtime does not apply) the product takes 55.7 ms with the other stages' kernels among its workgroups: 56.7-56.9 ms per step
// against 56.7-57.0 one block at a time, in five configurations on two boxes.  The product is limited by power and the stages behind
// it by HBM; whatever runs beside the product takes its watts.
static int xp_flush_fwd(hipStream_t s) { return xp_flush(s); }

extern "C" int gemma_hip_lmm_pipe_flush(void *stream) {
  NEED_INIT();
  return xp_flush(S(stream));
}

extern "C" int gemma_hip_lmm_batch_pipe_d(int kind, const void *geno, size_t l, size_t ld, gemma_sumstat *out_d, void *stream) {
  NEED_INIT();
  if (!g_ctx.lmm_active) return fail(GEMMA_HIP_ESTATE, "lmm_batch_pipe before lmm_setup");
  if (l == 0) return GEMMA_HIP_OK;
  int rc = check_batch_args("lmm_batch_pipe", kind, geno, l, ld, out_d);
  if (rc) return rc;
  hipStream_t s = S(stream);
  if (kind != GEMMA_GENO_PLINK_2BIT || utx_i8_mode() != 1 || i8_sparse_mode() != 2) {
    // nothing to pipeline on this path: the plain batch, behind whatever is still in flight
    if ((rc = xp_flush(s))) return rc;
    return gemma_hip_lmm_batch_d(kind, geno, l, ld, out_d, stream);
  }
  if ((rc = xp_init())) return rc;
  Ctx::XPipe &x = g_ctx.xp;
  const int slot = (int)(x.count & 1);
  const size_t n = g_ctx.cfg.n;
  const size_t ldx = (n + 1) & ~(size_t)1;
  // allocations first (a growing buffer is freed and re-allocated: hipFree waits for the device, which is what an in-flight reader
  // of the old buffer needs)
  if (g_ctx.UtX.reserve(l * ldx * 8)) return fail(GEMMA_HIP_ENOMEM, "lmm_batch_pipe: cannot allocate %zu bytes", l * ldx * 8);
  // The header's contract: the previous block's genotype buffer may be overwritten by work queued on `stream` AFTER this call.  Its
  // ingest runs on P, possibly still behind the product before it -- so the caller's stream is put behind that ingest here (ADVICE r5:
  // without this a double-buffering caller on a non-default stream could overwrite block i before ingest(i) had read it; on the legacy
  // default stream P is a blocking stream and the order held by itself -- no operation is issued on stream 0 here either, see below).
  if (s != nullptr && x.ingest_valid) HIPCHK(hipStreamWaitEvent(s, x.ingest_done, 0));
  xp_swap_sets(); // this block's A / C / mean / rowsur: the set block i - 2 used (its post stage is waited for below)
  // The block handed in is ready when the work already queued on s is done.  For the legacy default stream (s == 0: torch's current
  // stream unless the caller made another) nothing is recorded: streams with a CU mask are BLOCKING streams (the creating call takes
  // no flags), so P is ordered behind everything issued to stream 0 before this call anyway -- and any operation ON stream 0,
  // an event record included, is a barrier across P and Q that would serialise the two partitions again (measured: that one
  // record per call took the whole overlap away, 57.3 against 56.8 ms per step).
  if (s != nullptr) {
    HIPCHK(hipEventRecord(x.in_ready, s));
    HIPCHK(hipStreamWaitEvent(x.P, x.in_ready, 0));
  }
  if (x.post_valid[slot]) HIPCHK(hipStreamWaitEvent(x.P, x.post_done[slot], 0));
  g_ctx.last_utx_path = 1;
  I8Dims d;
  if ((rc = i8_begin(l, &d, x.P))) return xp_abort(rc);
  {
    ProfScope ps(GEMMA_STAGE_INGEST, x.P);
    IngestI8Args a;
    a.src = reinterpret_cast<const unsigned char *>(geno); a.ld = (long)ld; a.l = (long)l;
    a.idx_map = g_ctx.have_map ? g_ctx.idx_map.as<int>() : nullptr;
    a.n = (int)d.n; a.A = g_ctx.i8_A.as<int8_t>(); a.ldk = (long)d.ldk;
    a.mean = g_ctx.i8_mean.as<double>();
    hipLaunchKernelGGL(ingest_i8_kernel, dim3((unsigned)((l + 3) / 4)), dim3(256), 0, x.P, a);
    if (hipGetLastError() != hipSuccess) return xp_abort(fail(GEMMA_HIP_ERUNTIME, "lmm_batch_pipe: ingest launch"));
  }
  if (hipEventRecord(x.ingest_done, x.P) != hipSuccess) return xp_abort(fail(GEMMA_HIP_ERUNTIME, "lmm_batch_pipe: event"));
  x.ingest_valid = true;
  if ((rc = i8_meta_build(d, x.P))) return xp_abort(rc);
  if ((rc = i8_gemm_rows(d, 0, d.lpad, x.P))) return xp_abort(rc);
  HIPCHK(hipEventRecord(x.prod_done[slot], x.P));
  HIPCHK(hipStreamWaitEvent(x.Q, x.prod_done[slot], 0));
  double *UtX = g_ctx.UtX.as<double>();
  rc = i8_post_rows(l, d, 0, l, UtX, ldx, x.Q);
  if (!rc) rc = launch_assoc(UtX, l, ldx, out_d, x.Q);
  // whatever happened, what was queued on Q is waited for by the next user of this buffer set and by the flush
  (void)hipEventRecord(x.post_done[slot], x.Q);
  x.post_valid[slot] = true;
  x.pending = true;
  x.count += 1;
  return rc;
}

extern "C" int gemma_hip_lmm_batch_d(int kind, const void *geno, size_t l, size_t ld, gemma_sumstat *out_d,
                                     void *stream) {
  NEED_INIT();
  if (!g_ctx.lmm_active) return fail(GEMMA_HIP_ESTATE, "lmm_batch before lmm_setup");
  if (l == 0) return GEMMA_HIP_OK;
  int rc = check_batch_args("lmm_batch", kind, geno, l, ld, out_d);
  if (rc) return rc;
  hipStream_t s = S(stream);
  if ((rc = xp_flush(s))) return rc; // blocks of gemma_hip_lmm_batch_pipe_d still in flight share this call's buffers
  if (kind == GEMMA_GENO_PLINK_2BIT) {
    const int chunks = overlap_chunks(l);
    if (chunks > 1) return lmm_batch_plink_chunked(geno, l, ld, out_d, chunks, s);
  }
  double *UtX;
  size_t ldx;
  rc = compute_utx(kind, geno, l, ld, -1, &UtX, &ldx, s);
  if (rc) return rc;
  return launch_assoc(UtX, l, ldx, out_d, s);
}

// LMM::AnalyzeGene (src/lmm.cpp:1365-1471): rows are phenotypes (gene expression over the analysed individuals), the
// tested variable is the fixed vector handed to lmm_setup in the Uty slot (U^T x).  Y_d: l x ld fp64, device.
extern "C" int gemma_hip_lmm_gene_batch_d(const double *Y_d, size_t l, size_t ld, gemma_sumstat *out_d, void *stream) {
  NEED_INIT();
  if (!g_ctx.lmm_active) return fail(GEMMA_HIP_ESTATE, "lmm_gene_batch before lmm_setup");
  if (l == 0) return GEMMA_HIP_OK;
  const size_t n = g_ctx.cfg.n, c = g_ctx.cfg.n_cvt;
  if (!Y_d || !out_d || ld < n) return fail(GEMMA_HIP_EINVAL, "lmm_gene_batch: ld=%zu < n=%zu", ld, n);
  hipStream_t s = S(stream);
  const size_t ldx = (n + 1) & ~(size_t)1;
  if (int rcf = xp_flush(S(stream))) return rcf; // blocks of the two-block pipeline still in flight share these buffers
  if (g_ctx.UtX.reserve(l * ldx * 8)) return fail(GEMMA_HIP_ENOMEM, "lmm_gene_batch: %zu bytes", l * ldx * 8);
  double *UtY = g_ctx.UtX.as<double>();
  {
    ProfScope ps(GEMMA_STAGE_UTX_GEMM, s); // U^T y_g for every row (:1415)
    const double *Ug;
    long ldu;
    int rcu = gemm_U(&Ug, &ldu, s);
    if (rcu) return rcu;
    HIPCHK(launch_dgemm('N', 'N', (long)l, (long)n, (long)n, 1.0, Y_d, (long)ld, Ug, ldu, 0.0, UtY, (long)ldx,
                        false, false, s));
  }
  AssocArgs a = g_ctx.assoc_proto;
  a.UtX = UtY; a.ld = (long)ldx; a.l = (long)l;
  a.eval = g_ctx.eval; a.Uty = g_ctx.Uty; a.UtWt = g_ctx.UtWt.as<double>();
  a.out = reinterpret_cast<SumStat *>(out_d);
  a.grid_T = nullptr;
  a.have_grid = 0;
  const unsigned grid = (unsigned)((l + 3) / 4);
  {
    ProfScope ps(GEMMA_STAGE_ASSOC, s);
    switch (c) {
    case 1: hipLaunchKernelGGL(lmm_gene_kernel<1>, dim3(grid), dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL(lmm_gene_kernel<2>, dim3(grid), dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL(lmm_gene_kernel<3>, dim3(grid), dim3(256), 0, s, a); break;
    case 4: hipLaunchKernelGGL(lmm_gene_kernel<4>, dim3(grid), dim3(256), 0, s, a); break;
    default:
      if (c > (size_t)GEN_CMAX) {
        int rcw = wide_attr(lmm_gene_wide_kernel);
        if (rcw) return rcw;
        hipLaunchKernelGGL(lmm_gene_wide_kernel, dim3((unsigned)l), dim3(64), wide_lds_bytes(c), s, a, (int)c);
      } else {
        hipLaunchKernelGGL(lmm_gene_generic_kernel, dim3(grid), dim3(256), 0, s, a, (int)c);
      }
      break;
    }
    HIPCHK(hipGetLastError());
  }
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_lmm_gene_batch(const double *Y, size_t l, size_t ld, gemma_sumstat *out) {
  NEED_INIT();
  if (!g_ctx.lmm_active) return fail(GEMMA_HIP_ESTATE, "lmm_gene_batch before lmm_setup");
  if (l == 0) return GEMMA_HIP_OK;
  const size_t n = g_ctx.cfg.n;
  if (!Y || !out || ld < n) return fail(GEMMA_HIP_EINVAL, "lmm_gene_batch: ld=%zu < n=%zu", ld, n);
  if (g_ctx.stage_in.reserve(l * ld * 8) || g_ctx.stage_out.reserve(l * sizeof(gemma_sumstat)))
    return fail(GEMMA_HIP_ENOMEM, "lmm_gene_batch: staging %zu bytes", l * ld * 8);
  HIPCHK(hipMemcpy(g_ctx.stage_in.p, Y, l * ld * 8, hipMemcpyHostToDevice));
  int rc = gemma_hip_lmm_gene_batch_d(g_ctx.stage_in.as<double>(), l, ld, g_ctx.stage_out.as<gemma_sumstat>(), nullptr);
  if (rc) return rc;
  HIPCHK(hipMemcpy(out, g_ctx.stage_out.p, l * sizeof(gemma_sumstat), hipMemcpyDeviceToHost));
  return GEMMA_HIP_OK;
}
