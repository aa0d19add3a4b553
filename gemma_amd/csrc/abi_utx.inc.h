// Part of gemma_hip.hip (ONE translation unit: the parts share the context g_ctx and the helpers of its anonymous namespace, and are
// included there in this order; round 6: the 3 500-line file cut along its stages for reading -- no behaviour change).
// This part: U^T x launch sites: the int8-digit products (records kernel, dosage planes), digit combine, compute_utx.

// ---- exact int8-digit U^T x (i8gemm.hip.h): buffers, the product on an already packed left factor, ingest variants
struct I8Dims {
  size_t n, ldk, npad, lpad, mrows;
  int fuse, digits, nplanes;
  int mdrop; // 1: the 7g6m form -- plane 0 (digit 0 alone) carries the genotype product only
  int complete; // 1: sparse2_meta_kernel flags the block's missing calls (the int after the row counters) and a block without one
                // takes the genotype product alone (Sparse2Args::anymiss): same planes, same U^T x, bit for bit
};
static const int *i8_anymiss(const I8Dims &d) { return d.complete ? g_ctx.i8_rowsur.as<int>() + d.lpad : nullptr; }
// GEMMA_HIP_I8_SPARSE: 0 = the mask product on dense MFMAs (i8gemm_packed_kernel_t), 1 = on the 2:4 sparse MFMA with byte-wise
// genotypes and separate mask words (i8gemm_sparse.hip.h), 2 (default) = sparse MFMA, left factor as 16-byte records of 2-bit
// genotypes + mask words, 256 x 128 tiles (i8gemm_sparse2.hip.h)
static int i8_sparse_mode() { return g_ctx.knobs.i8_sparse; }
static int i8_begin(size_t l, I8Dims *d, hipStream_t s) {
  int rc = i8_prepare_u(s);
  if (rc) return rc;
  d->n = g_ctx.cfg.n; d->ldk = g_ctx.i8_ldk; d->npad = g_ctx.i8_npad;
  d->lpad = round_up(l, i8_sparse_mode() == 2 ? (size_t)S2_BM : (size_t)I8P_BM); d->mrows = 2 * d->lpad;
  // two digits per int32 output plane while 256 * C_hi + C_lo cannot overflow: n * 2 * 128 * 257 < 2^31
  d->fuse = (g_ctx.knobs.i8_fuse && (double)d->n * 2.0 * 128.0 * 257.0 < 2147483648.0) ? 1 : 0;
  d->digits = g_ctx.i8_digits;
  d->nplanes = d->fuse ? (d->digits + 1) / 2 : d->digits;
  // the 7g6m form needs plane 0 to be digit 0 alone (odd count, fused planes) and the 16-row records kernel
  d->mdrop = (g_ctx.knobs.i8_mdrop && d->fuse && d->digits == 7 && i8_sparse_mode() == 2 && g_ctx.knobs.i8_rows == 16) ? 1 : 0;
  d->complete = (g_ctx.knobs.i8_complete && i8_sparse_mode() == 2 && g_ctx.knobs.i8_rows == 16) ? 1 : 0;
  const size_t c_elems = (size_t)d->nplanes * d->mrows * d->npad;
  if (g_ctx.i8_A.reserve(d->lpad * d->ldk) || g_ctx.i8_C.reserve(c_elems * 4) || g_ctx.i8_mean.reserve(l * 8))
    return fail(GEMMA_HIP_ENOMEM, "lmm_batch: int8 product buffers (%zu bytes)", d->lpad * d->ldk + c_elems * 4);
  if (d->lpad != l) HIPCHK(hipMemsetAsync(g_ctx.i8_A.p, 0, d->lpad * d->ldk, s)); // padding rows
  return GEMMA_HIP_OK;
}

// The int8 product in three pieces, each over the SNP rows [row0, row0 + rows) of the packed block (rows, row0 multiples of the
// kernel's tile height except for the last piece of a block): mask words / records, the matrix product, the digit combine.
static int i8_meta_build(const I8Dims &d, hipStream_t s) {
  const int mode = i8_sparse_mode();
  if (mode == 0) return GEMMA_HIP_OK;
  ProfScope ps(GEMMA_STAGE_INGEST, s);
  const size_t nk = d.ldk / I8_BK, total = d.lpad * nk * (mode == 2 ? 4 : 2);
  if (g_ctx.i8_meta.reserve(total * sizeof(uint4)) || g_ctx.i8_rowsur.reserve((d.lpad + 1) * sizeof(int)))
    return fail(GEMMA_HIP_ENOMEM, "lmm_batch: mask words of the sparse product");
  HIPCHK(hipMemsetAsync(g_ctx.i8_rowsur.p, 0, (d.lpad + 1) * sizeof(int), s)); // dropped calls per row, then the block's any-missing flag
  if (mode == 2)
    hipLaunchKernelGGL(sparse2_meta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g_ctx.i8_A.as<int8_t>(),
                       (long)d.lpad, (long)d.ldk, g_ctx.i8_meta.as<uint4>(), g_ctx.i8_rowsur.as<int>(),
                       const_cast<int *>(i8_anymiss(d)));
  else
    hipLaunchKernelGGL(sparse_meta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g_ctx.i8_A.as<int8_t>(),
                       (long)d.lpad, (long)d.ldk, g_ctx.i8_meta.as<uint4>(), g_ctx.i8_rowsur.as<int>());
  HIPCHK(hipGetLastError());
  g_ctx.i8_flag_at = (mode == 2 && d.complete) ? (long)d.lpad : -1;
  return GEMMA_HIP_OK;
}

// The raster of a launch shape: found in the cache or built into the least recently used slot.  A slot is only recycled when more
// than six shapes are alive (a block in row chunks has two); then the device is synchronised first, since a kernel on ANY stream may
// still read the map that goes.
static int raster_for(int tiles_m, int tiles_n, int rb, hipStream_t s, const int2 **map_d) {
  Ctx::RasterSlot *lru = &g_ctx.i8_raster[0];
  for (auto &r : g_ctx.i8_raster) {
    if (r.tm == tiles_m && r.tn == tiles_n && r.rb == rb && r.dev.p) {
      r.used = ++g_ctx.i8_raster_clock;
      *map_d = r.dev.as<int2>();
      return GEMMA_HIP_OK;
    }
    if (r.used < lru->used) lru = &r;
  }
  if (lru->dev.p) HIPCHK(hipDeviceSynchronize()); // recycling a map some launch may still read
  s2_build_raster(tiles_m, tiles_n, rb, lru->host);
  lru->tm = lru->tn = lru->rb = 0;
  if (lru->dev.reserve(lru->host.size() * sizeof(int2)))
    return fail(GEMMA_HIP_ENOMEM, "lmm_batch: tile raster (%zu bytes)", lru->host.size() * sizeof(int2));
  // once per launch shape, and synchronous: a later cache hit hands the same map to a launch on ANY stream, and nothing
  // would order that launch behind an upload still queued on this one
  HIPCHK(hipMemcpyAsync(lru->dev.p, lru->host.data(), lru->host.size() * sizeof(int2), hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));
  lru->tm = tiles_m; lru->tn = tiles_n; lru->rb = rb;
  lru->used = ++g_ctx.i8_raster_clock;
  *map_d = lru->dev.as<int2>();
  return GEMMA_HIP_OK;
}

// rows_pad: padded rows of this piece (a multiple of the tile height; row0 too).  Pieces other than the whole block are taken
// by the records kernel only (mode 2).
static int i8_gemm_rows(const I8Dims &d, size_t row0, size_t rows_pad, hipStream_t s) {
  // GEMMA_HIP_I8_SPARSE=0: the mask product on dense MFMAs (i8gemm_packed_kernel_t); default: on the 2:4 sparse MFMA
  // (i8gemm_sparse2.hip.h) -- rows that lose calls to the 2-of-4 limit are completed in fp64 after the digits are combined
  const int mode = i8_sparse_mode();
  const bool sparse = mode != 0;
  ProfScope ps(GEMMA_STAGE_UTX_GEMM, s);
  static bool attr_set = false;
  if (!attr_set) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(i8gemm_packed_kernel_t<true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 3 * I8P_STAGE));
    attr_set = true;
  }
  I8PackArgs g;
  g.A = g_ctx.i8_A.as<int8_t>();
  g.Bt = g_ctx.i8_Bt.as<int8_t>();
  g.C = g_ctx.i8_C.as<int>();
  g.ldk = (long)d.ldk; g.ldc = (long)d.npad;
  g.strideB = (long)(d.npad * d.ldk); g.strideC = (long)(d.mrows * d.npad);
  g.m_row0 = (long)d.lpad;
  g.tiles_m = (int)(d.lpad / I8P_BM); g.tiles_n = (int)(d.npad / I8_BN);
  g.nk = (int)(d.ldk / I8_BK);
  g.gm = g_ctx.knobs.i8_gm;
  g.fuse = d.fuse;
  g.digits = d.digits;
  const dim3 grid((unsigned)(g.tiles_m * g.tiles_n), (unsigned)d.nplanes);
  if (mode == 2) {
    static bool attr3 = false;
    if (!attr3) {
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(i8gemm_sparse2_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, S2_NST * S2_STAGE));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(i8gemm_sparse2_r16_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, S2_R16_LDS));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(i8gemm_sparse2_r16_g_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, S2_R16_LDS));
      attr3 = true;
    }
    Sparse2Args g2;
    // records: [tile_m][ktile][row % 256][chunk]; planes: G rows at row, M rows at lpad + row
    g2.AM = g_ctx.i8_meta.as<uint4>() + (row0 / S2_BM) * (size_t)g.nk * S2_BM * 4;
    g2.Bt = g.Bt; g2.C = g.C + row0 * (size_t)g.ldc; g2.ldk = g.ldk; g2.ldc = g.ldc; g2.strideB = g.strideB; g2.strideC = g.strideC;
    g2.m_row0 = g.m_row0;
    g2.tiles_m = (int)(rows_pad / S2_BM); g2.tiles_n = (int)(d.npad / S2_BN);
    g2.nk = g.nk; g2.gm = g.gm; g2.fuse = g.fuse; g2.digits = g.digits;
    int raster_rb = 0;
    {
      // GEMMA_HIP_I8_RASTER: 0 = every XCD sweeps its own tile rows (round 3); 1 / 2 / 4 / 8 = row blocks of the super-patch the
      // eight XCDs share (s2_build_raster)
      const int rb = g_ctx.knobs.i8_raster;
      if (rb > 0) {
        const int2 *map_d = nullptr;
        int rc_map = raster_for(g2.tiles_m, g2.tiles_n, rb, s, &map_d);
        if (rc_map) return rc_map;
        g2.tile_map = map_d;
        raster_rb = rb;
      }
    }
    // GEMMA_HIP_I8_ROWS=32: the kernel of rounds 3-4 on the 32-row matrix instructions; default: the same product on the 16-row
    // forms (i8gemm_sparse2_r16.hip.h: same records, same planes, every entry equal; 9 % faster under the power limit)
    note_utx_kernel(g_ctx.knobs.i8_rows == 32 ? GEMMA_UTX_KERNEL_RECORDS_R32 : GEMMA_UTX_KERNEL_RECORDS_R16, d.digits, d.fuse,
                    raster_rb);
    if (g_ctx.knobs.i8_rows == 32) {
      hipLaunchKernelGGL(i8gemm_sparse2_kernel, dim3((unsigned)(g2.tiles_m * g2.tiles_n), (unsigned)d.nplanes), dim3(512),
                         S2_NST * S2_STAGE, s, g2);
    } else {
      // 7g6m: planes 1..3 (digit pairs {2,1} {4,3} {6,5}) with both products, then plane 0 (digit 0) with the genotype product alone
      const unsigned wgs = (unsigned)(g2.tiles_m * g2.tiles_n), first = d.mdrop ? 1u : 0u;
      g2.anymiss = i8_anymiss(d);
      g2.plane0 = (int)first;
      g2.run_if = g2.anymiss ? 1 : 0; // both products: always, or (complete-block form) only when the block has a missing call
      hipLaunchKernelGGL(i8gemm_sparse2_r16_kernel, dim3(wgs, (unsigned)d.nplanes - first), dim3(512), S2_R16_LDS, s, g2);
      if (g2.anymiss) { // the same planes from the genotype product alone when it has none (the other launch returned at once)
        g2.run_if = 2;
        hipLaunchKernelGGL(i8gemm_sparse2_r16_g_kernel, dim3(wgs, (unsigned)d.nplanes - first), dim3(512), S2_R16_LDS, s, g2);
      }
      if (d.mdrop) {
        g2.plane0 = 0; g2.run_if = 0;
        hipLaunchKernelGGL(i8gemm_sparse2_r16_g_kernel, dim3(wgs, 1u), dim3(512), S2_R16_LDS, s, g2);
      }
    }
  } else if (sparse) {
    static bool attr2 = false;
    if (!attr2) {
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(i8gemm_sparse_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 3 * SP_STAGE));
      attr2 = true;
    }
    SparseMeta sm;
    sm.m4 = g_ctx.i8_meta.as<uint4>();
    sm.row_surplus = g_ctx.i8_rowsur.as<int>();
    sm.ntiles = (long)g.nk;
    note_utx_kernel(GEMMA_UTX_KERNEL_SPARSE_BYTES, d.digits, d.fuse, 0);
    hipLaunchKernelGGL(i8gemm_sparse_kernel, grid, dim3(512), 3 * SP_STAGE, s, g, sm);
  } else {
    note_utx_kernel(GEMMA_UTX_KERNEL_DENSE_I8, d.digits, d.fuse, 0);
    hipLaunchKernelGGL(i8gemm_packed_kernel_t<true>, grid, dim3(512), 3 * I8P_STAGE, s, g);
  }
  HIPCHK(hipGetLastError());
  return GEMMA_HIP_OK;
}

// digits -> fp64 for the rows [row0, row0 + rows) of a block of l SNPs
static int i8_post_rows(size_t l, const I8Dims &d, size_t row0, size_t rows, double *UtX, size_t ldx, hipStream_t s) {
  const bool sparse = i8_sparse_mode() != 0;
  ProfScope ps(GEMMA_STAGE_UTX_POST, s);
  // the calls the sparse mask operand dropped (groups of four with 3-4 missing calls): rows with up to SUR_MAX of them are
  // completed inside the digit combine from a short per-row list, the rare rows with more by the fp64 fix-up pass
  int *sur_cnt = nullptr, *sur_list = nullptr;
  const int8_t *Arow = g_ctx.i8_A.as<int8_t>() + row0 * d.ldk;
  if (sparse) {
    if (g_ctx.i8_surlist.reserve(l * (SUR_MAX + 1) * sizeof(int)))
      return fail(GEMMA_HIP_ENOMEM, "lmm_batch: dropped-call lists");
    sur_cnt = g_ctx.i8_surlist.as<int>() + row0;
    sur_list = g_ctx.i8_surlist.as<int>() + l + row0 * SUR_MAX;
    hipLaunchKernelGGL(i8_surplus_list_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, Arow, (long)d.ldk,
                       g_ctx.i8_rowsur.as<int>() + row0, (long)rows, sur_cnt, sur_list);
    HIPCHK(hipGetLastError());
  }
  hipLaunchKernelGGL(i8_combine_kernel, dim3((unsigned)((d.n + 1023) / 1024), (unsigned)std::min<size_t>(rows, 65535)),
                     dim3(256), 0, s,
                     g_ctx.i8_C.as<int>() + row0 * d.npad, (long)d.npad, (long)(d.mrows * d.npad), (long)d.lpad,
                     g_ctx.i8_mean.as<double>() + row0, g_ctx.i8_qinv.as<double>(), (long)rows, (long)d.n, UtX + row0 * ldx, (long)ldx,
                     1.0, d.fuse, d.digits, sur_cnt, sur_list, g_ctx.U, (long)d.n, d.mdrop, i8_anymiss(d));
  HIPCHK(hipGetLastError());
  if (sparse) {
    hipLaunchKernelGGL(i8_surplus_fix_kernel, dim3((unsigned)rows), dim3(256), 0, s, Arow, (long)d.ldk,
                       g_ctx.i8_rowsur.as<int>() + row0, g_ctx.i8_mean.as<double>() + row0, g_ctx.U, (long)d.n, (long)d.n,
                       (long)rows, UtX + row0 * ldx, (long)ldx, SUR_MAX);
    HIPCHK(hipGetLastError());
  }
  return GEMMA_HIP_OK;
}

// UtX (l x ldx) from the packed left factor in g_ctx.i8_A and the per-SNP means in g_ctx.i8_mean
static int i8_product(size_t l, const I8Dims &d, double *UtX, size_t ldx, hipStream_t s) {
  int rc = i8_meta_build(d, s);
  if (!rc) rc = i8_gemm_rows(d, 0, d.lpad, s);
  if (!rc) rc = i8_post_rows(l, d, 0, l, UtX, ldx, s);
  return rc;
}

// PLINK 2-bit batch
static int utx_plink_i8(const void *geno, size_t l, size_t ld, double *UtX, size_t ldx, hipStream_t s) {
  I8Dims d;
  int rc = i8_begin(l, &d, s);
  if (rc) return rc;
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
  return i8_product(l, d, UtX, ldx, s);
}

// Fixed-point dosage rows (i8gemm.hip.h: pack_dosage_kernel): byte planes a0 [, a1] [, mask] x the digits of U on the dense int8
// kernel, one int32 plane per (byte plane, digit); GEMMA_HIP_UTX_DOSAGE_I8=0 keeps such batches on the fp64 GEMM.
static bool dosage_i8_enabled() { return g_ctx.knobs.dosage_i8 != 0; }
static int utx_dosage_i8(const double *src, size_t l, size_t ld, bool nan_missing, bool two, bool have_m, const I8Dims &d,
                         double *UtX, size_t ldx, hipStream_t s) {
  const int np = (two ? 2 : 1) + (have_m ? 1 : 0);
  const size_t plane_a = d.lpad * d.ldk, plane_c = d.lpad * d.npad;
  if (g_ctx.i8_A.reserve((size_t)np * plane_a) || g_ctx.i8_C.reserve((size_t)np * d.digits * plane_c * 4) ||
      g_ctx.i8_colsum.reserve(d.n * 8))
    return fail(GEMMA_HIP_ENOMEM, "lmm_batch: dosage planes (%zu bytes)", (size_t)np * (plane_a + d.digits * plane_c * 4));
  int8_t *A0 = g_ctx.i8_A.as<int8_t>();
  {
    ProfScope ps(GEMMA_STAGE_INGEST, s);
    if (d.lpad != l) HIPCHK(hipMemsetAsync(A0, 0, (size_t)np * plane_a, s)); // padding rows of every plane
    PackDosageArgs a;
    a.src = src; a.ld = (long)ld; a.l = (long)l; a.n = (int)d.n; a.nan_missing = nan_missing ? 1 : 0; a.two = two ? 1 : 0;
    a.A0 = A0; a.A1 = two ? A0 + plane_a : nullptr; a.Am = have_m ? A0 + (size_t)(np - 1) * plane_a : nullptr;
    a.ldk = (long)d.ldk; a.mean = g_ctx.i8_mean.as<double>();
    hipLaunchKernelGGL(pack_dosage_kernel, dim3((unsigned)((l + 3) / 4)), dim3(256), 0, s, a);
    HIPCHK(hipGetLastError());
    if (!g_ctx.i8_colsum_ready) {
      hipLaunchKernelGGL(u_digit_colsum_kernel, dim3((unsigned)((d.n + 3) / 4)), dim3(256), 0, s, g_ctx.i8_Bt.as<int8_t>(),
                         (long)d.ldk, (long)(d.npad * d.ldk), g_ctx.i8_qinv.as<double>(), (long)d.n, d.digits,
                         g_ctx.i8_colsum.as<double>());
      HIPCHK(hipGetLastError());
      g_ctx.i8_colsum_ready = true;
    }
  }
  {
    ProfScope ps(GEMMA_STAGE_UTX_GEMM, s);
    static bool attr_set = false;
    if (!attr_set) {
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(i8gemm_packed_kernel_t<false, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 3 * I8P_STAGE));
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(i8gemm_dense16_kernel_t<true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 3 * I8P_STAGE));
      attr_set = true;
    }
    for (int a = 0; a < np; ++a) {
      I8PackArgs g;
      g.A = A0 + (size_t)a * plane_a;
      g.Bt = g_ctx.i8_Bt.as<int8_t>();
      g.C = g_ctx.i8_C.as<int>() + (size_t)a * d.digits * plane_c;
      g.ldk = (long)d.ldk; g.ldc = (long)d.npad;
      g.strideB = (long)(d.npad * d.ldk); g.strideC = (long)plane_c;
      g.m_row0 = 0;
      g.tiles_m = (int)(d.lpad / I8P_BM); g.tiles_n = (int)(d.npad / I8_BN);
      g.nk = (int)(d.ldk / I8_BK);
      g.gm = g_ctx.knobs.i8_gm;
      g.fuse = 0;
      g.digits = d.digits;
      // round 5: the byte planes on v_mfma_i32_16x16x64_i8 (i8gemm_dense16.hip.h: same tiles, same LDS images, every plane entry equal;
      // 44.5 against 46.8 ms for six planes at n = B = 20 000 under the power limit); GEMMA_HIP_DOSAGE_ROWS=32: the 32-row kernel
      if (g_ctx.knobs.dosage_rows == 32) {
        note_utx_kernel(GEMMA_UTX_KERNEL_DOSAGE_I8, d.digits, 0, 0);
        hipLaunchKernelGGL((i8gemm_packed_kernel_t<false, true>), dim3((unsigned)(g.tiles_m * g.tiles_n), (unsigned)d.digits),
                           dim3(512), 3 * I8P_STAGE, s, g);
      } else {
        note_utx_kernel(GEMMA_UTX_KERNEL_DOSAGE_I8_R16, d.digits, 0, 0);
        hipLaunchKernelGGL((i8gemm_dense16_kernel_t<true>), dim3((unsigned)(g.tiles_m * g.tiles_n), (unsigned)d.digits),
                           dim3(512), 3 * I8P_STAGE, s, g);
      }
      HIPCHK(hipGetLastError());
    }
  }
  {
    ProfScope ps(GEMMA_STAGE_UTX_POST, s);
    hipLaunchKernelGGL(i8_combine_dosage_kernel, dim3((unsigned)((d.n + 255) / 256), (unsigned)std::min<size_t>(l, 65535)),
                       dim3(256), 0, s, g_ctx.i8_C.as<int>(), (long)d.npad, (long)plane_c, g_ctx.i8_mean.as<double>(),
                       g_ctx.i8_qinv.as<double>(), g_ctx.i8_colsum.as<double>(), (long)l, (long)d.n, UtX, (long)ldx, d.digits,
                       two ? 1 : 0, have_m ? 1 : 0, two ? 1000.0 : 100.0);
    HIPCHK(hipGetLastError());
  }
  return GEMMA_HIP_OK;
}

// fp64 SNP-major rows (src: l x ld): if every row is a hard-call row (i8gemm.hip.h, pack_f64_kernel) the batch goes
// through the int8-digit product and *done = true; otherwise nothing is computed and the caller takes the fp64 GEMM.
// One stream synchronisation per batch (the verdict is read back).
static int utx_f64_try_i8(const double *src, size_t l, size_t ld, bool nan_missing, double *UtX, size_t ldx,
                          hipStream_t s, bool *done) {
  *done = false;
  I8Dims d;
  int rc = i8_begin(l, &d, s);
  if (rc) return rc;
  if (g_ctx.scratch.reserve(16)) return fail(GEMMA_HIP_ENOMEM, "lmm_batch: scratch");
  const int init[4] = {1, 1, 1, 0}; // hard calls, dosages k/1000, dosages k/100, any missing entry
  HIPCHK(hipMemcpyAsync(g_ctx.scratch.p, init, sizeof init, hipMemcpyHostToDevice, s));
  {
    ProfScope ps(GEMMA_STAGE_INGEST, s);
    PackF64Args a;
    a.src = src; a.ld = (long)ld; a.l = (long)l; a.n = (int)d.n; a.nan_missing = nan_missing ? 1 : 0;
    a.A = g_ctx.i8_A.as<int8_t>(); a.ldk = (long)d.ldk; a.mean = g_ctx.i8_mean.as<double>();
    a.all_hard = g_ctx.scratch.as<int>();
    hipLaunchKernelGGL(pack_f64_kernel, dim3((unsigned)((l + 3) / 4)), dim3(256), 0, s, a);
    HIPCHK(hipGetLastError());
  }
  int verdict[4] = {0, 0, 0, 0};
  HIPCHK(hipMemcpyAsync(verdict, g_ctx.scratch.p, sizeof verdict, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  if (verdict[0]) {
    *done = true;
    g_ctx.last_utx_path = 1;
    return i8_product(l, d, UtX, ldx, s);
  }
  if (verdict[1] && dosage_i8_enabled()) { // fixed-point dosages: k/100 on one byte plane, k/1000 on two
    *done = true;
    g_ctx.last_utx_path = verdict[2] ? 2 : 3;
    return utx_dosage_i8(src, l, ld, nan_missing, !verdict[2], verdict[3] != 0, d, UtX, ldx, s);
  }
  return GEMMA_HIP_OK;
}

// U as the right-hand operand of the fp64 GEMM.  With an odd n the caller's U (leading dimension n) would send every tile down
// the bounds-checked kernel (the LDS-DMA path wants even leading dimensions): a copy with leading dimension n + 1 is made
// once per lmm_setup and used instead.
static int gemm_U(const double **U, long *ld, hipStream_t s) {
  const size_t n = g_ctx.cfg.n;
  *U = g_ctx.U;
  *ld = (long)n;
  if ((n & 1) == 0) return GEMMA_HIP_OK;
  if (g_ctx.U_even_of != g_ctx.U) {
    if (g_ctx.U_even.reserve(n * (n + 1) * 8)) return fail(GEMMA_HIP_ENOMEM, "lmm_batch: even-ld copy of U");
    HIPCHK(hipMemcpy2DAsync(g_ctx.U_even.p, (n + 1) * 8, g_ctx.U, n * 8, n * 8, n, hipMemcpyDeviceToDevice, s));
    g_ctx.U_even_of = g_ctx.U;
  }
  *U = g_ctx.U_even.as<double>();
  *ld = (long)n + 1;
  return GEMMA_HIP_OK;
}
