// C ABI of the MI355X kinship + univariate-LMM path (see include/gemma_hip.h).
// Host-side glue only: device memory, streams, launches, staging copies.  All arithmetic of the
// hot path lives in the kernels of dgemm_mfma.hip.h / lmm_assoc.hip.h / ingest.hip.h / eigh.hip.h.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gemma_hip.h"
#include "dgemm_mfma.hip.h"
#include "eigh_tu.h"
#include "ingest.hip.h"
#include "lm_assoc.hip.h"
#include "lmm_assoc.hip.h"
#include "lmm_grid.hip.h"
#include "i8gemm.hip.h"
#include "i8gemm_sparse.hip.h"
#include "i8gemm_sparse2.hip.h"
#include "i8gemm_sparse2_r16.hip.h"
#include "i8gemm_dense16.hip.h"
#include "qc.hip.h"
#include "mvlmm.hip.h"
#include "comm.hip.h"
#include "kin_i8.hip.h"

using namespace gemma_hip;

namespace {

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return GEMMA_HIP_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    if (hipMalloc(&p, bytes) != hipSuccess) {
      (void)hipGetLastError();
      return GEMMA_HIP_ENOMEM;
    }
    cap = bytes;
    return GEMMA_HIP_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T> T *as() { return reinterpret_cast<T *>(p); }
};

struct StageProf {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  double acc_ms = 0.0;
  long launches = 0;
};

// Environment switches of the batch path.  Read ONCE per setup -- gemma_hip_init, every lmm_setup* / lm_setup / mvlmm_set,
// kin_begin and gemma_hip_reload_env -- never on a launch path (a getenv per launch walks the whole environment block under
// libc's lock; round 4 had three of them on every records-kernel launch).  Tests and bench legs that flip a switch between
// two batches of one setup call gemma_hip_reload_env().
struct Knobs {
  int utx_i8 = 1;          // GEMMA_HIP_UTX_I8: 1 = hard-call batches through the exact int8-digit product, 0 = always the fp64 GEMM
  int i8_digits = 0;       // GEMMA_HIP_I8_DIGITS: 6 | 7 forces the digit count (0: by n)
  int i8_sparse = 2;       // GEMMA_HIP_I8_SPARSE: 0 dense mask product, 1 sparse MFMA on byte genotypes, 2 records kernel
  int i8_fuse = 1;         // GEMMA_HIP_I8_FUSE: 0 = one int32 plane per digit
  int i8_gm = 0;           // GEMMA_HIP_I8_GM: tile rows per L2 patch of the non-rastered launch
  int i8_raster = S2_DEFAULT_RASTER; // GEMMA_HIP_I8_RASTER
  int i8_rows = 16;        // GEMMA_HIP_I8_ROWS: 32 = the records kernel on the 32-row matrix instructions
  int i8_scale_max = 1;    // GEMMA_HIP_I8_SCALE: "pow2" = columns of U scaled by a power of two (rounds 1-5); default: by their exact maximum
  int i8_mdrop = 0;        // GEMMA_HIP_I8_FORM=7g6m: seven digits for the genotype product, the mask product on the upper six
  int i8_complete = 1;     // GEMMA_HIP_I8_COMPLETE: 0 = blocks without a missing call take the mask product like any other block
  int dosage_i8 = 1;       // GEMMA_HIP_UTX_DOSAGE_I8
  int dosage_rows = 16;    // GEMMA_HIP_DOSAGE_ROWS: 32 = the dosage planes on the 32-row dense kernel (rounds 3-4)
  int overlap = 0;         // GEMMA_HIP_OVERLAP
  int overlap_chunks = 4;  // GEMMA_HIP_OVERLAP_CHUNKS
  int table_v2 = 1;        // GEMMA_HIP_TABLE_V2
  int table_pf = 0;        // GEMMA_HIP_TABLE_PF
  int force_generic = 0;   // GEMMA_HIP_FORCE_GENERIC
  int assoc_variant = 44;  // GEMMA_HIP_ASSOC_VARIANT
  int mvlmm_rt = 0;        // GEMMA_HIP_MVLMM_RT
  int pipe_cus = 64;       // GEMMA_HIP_PIPE_CUS: CUs of the post partition of gemma_hip_lmm_batch_pipe_d (0: two plain streams, no masks)
  int kin_i8 = 1, kin_upper = 1, kin_lists = 1, kin_lists_oom = 0; // GEMMA_HIP_KIN_I8 / _UPPER / _LISTS / _LISTS_OOM
  long long kin_list_cap = 0;                                       // GEMMA_HIP_KIN_LIST_CAP (0: by block size)
  static int geti(const char *name, int dflt) {
    const char *e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
  }
  void load() {
    utx_i8 = geti("GEMMA_HIP_UTX_I8", 1);
    const int dg = geti("GEMMA_HIP_I8_DIGITS", 0);
    i8_digits = (dg == 6 || dg == 7) ? dg : 0;
    const char *es = getenv("GEMMA_HIP_I8_SPARSE");
    i8_sparse = (es && es[0] >= '0' && es[0] <= '2') ? es[0] - '0' : 2;
    const char *ef = getenv("GEMMA_HIP_I8_FUSE");
    i8_fuse = (ef && ef[0] == '0') ? 0 : 1;
    i8_gm = geti("GEMMA_HIP_I8_GM", 0);
    i8_raster = geti("GEMMA_HIP_I8_RASTER", S2_DEFAULT_RASTER);
    i8_rows = geti("GEMMA_HIP_I8_ROWS", 16) == 32 ? 32 : 16;
    const char *esc = getenv("GEMMA_HIP_I8_SCALE");
    i8_scale_max = (esc && strcmp(esc, "pow2") == 0) ? 0 : 1;
    const char *efm = getenv("GEMMA_HIP_I8_FORM");
    i8_mdrop = (efm && strcmp(efm, "7g6m") == 0) ? 1 : 0;
    if (i8_mdrop) i8_digits = 7;
    const char *ecp = getenv("GEMMA_HIP_I8_COMPLETE");
    i8_complete = (ecp && ecp[0] == '0') ? 0 : 1;
    const char *ed = getenv("GEMMA_HIP_UTX_DOSAGE_I8");
    dosage_i8 = (ed && ed[0] == '0') ? 0 : 1;
    dosage_rows = geti("GEMMA_HIP_DOSAGE_ROWS", 16) == 32 ? 32 : 16;
    const char *eo = getenv("GEMMA_HIP_OVERLAP");
    overlap = (eo && eo[0] == '1') ? 1 : 0;
    overlap_chunks = geti("GEMMA_HIP_OVERLAP_CHUNKS", 4);
    const char *et = getenv("GEMMA_HIP_TABLE_V2");
    table_v2 = (et && et[0] == '0') ? 0 : 1;
    const char *ep = getenv("GEMMA_HIP_TABLE_PF");
    table_pf = (ep && ep[0] == '1') ? 1 : 0;
    const char *eg = getenv("GEMMA_HIP_FORCE_GENERIC");
    force_generic = (eg && eg[0] == '1') ? 1 : 0;
    assoc_variant = geti("GEMMA_HIP_ASSOC_VARIANT", 44);
    const char *er = getenv("GEMMA_HIP_MVLMM_RT");
    mvlmm_rt = (er && er[0] == '1') ? 1 : 0;
    pipe_cus = geti("GEMMA_HIP_PIPE_CUS", 64);
    const char *k1 = getenv("GEMMA_HIP_KIN_I8"), *k2 = getenv("GEMMA_HIP_KIN_UPPER"), *k3 = getenv("GEMMA_HIP_KIN_LISTS");
    const char *k4 = getenv("GEMMA_HIP_KIN_LISTS_OOM"), *k5 = getenv("GEMMA_HIP_KIN_LIST_CAP");
    kin_i8 = (k1 && k1[0] == '0') ? 0 : 1;
    kin_upper = (k2 && k2[0] == '0') ? 0 : 1;
    kin_lists = (k3 && k3[0] == '0') ? 0 : 1;
    kin_lists_oom = (k4 && k4[0] == '1') ? 1 : 0;
    kin_list_cap = (k5 && *k5) ? atoll(k5) : 0;
  }
};

struct Ctx {
  bool inited = false;
  Knobs knobs;
  int device = -1;
  int verbose = 0;
  std::string last_error;
  hipDeviceProp_t prop;
  bool profiling = false;
  StageProf prof[GEMMA_STAGE_COUNT];

  // kinship state
  bool kin_active = false;
  size_t kin_n = 0;
  int kin_mode = 1;
  size_t kin_ns = 0;
  DevBuf kin_K, kin_X, kin_stage;
  // gemma_hip_kin_add (host blocks): the upload of block k + 1 runs on its own stream beside the kernels of block k (round 6); it
  // waits for the ingest of block k -- the only reader of the staging buffer -- and the null stream waits for the upload
  hipStream_t kin_copy = nullptr;
  hipEvent_t kin_copied = nullptr, kin_ingested = nullptr;
  bool kin_ingested_valid = false, kin_host_call = false;
  DevBuf qc_G, qc_M, qc_W, qc_O; // gemma_hip_snp_qc: block, index map, W^T, statistics -- kept between the calls of a first pass (round 6:
                                 // four hipMalloc + hipFree per 100 MB block were a fifth of the pass), released by kin_begin / shutdown
  // exact-integer path of the centred kinship of hard calls (kin_i8.hip.h)
  bool kin_i8 = false, kin_i8_used = false;
  DevBuf i8_meta, i8_rowsur; // sparse mask operand: the words of the packed block, dropped calls per row
  DevBuf U_even;            // odd n: U copied to an even leading dimension for the fp64 GEMM's aligned path
  const double *U_even_of = nullptr; // the U that copy was made from
  DevBuf kin_GtG, kin_S, kin_a, kin_At, kin_Gt;
  DevBuf kin_A2, kin_cnt, kin_off, kin_listS, kin_listJ, kin_sub, kin_cj, kin_flag; // lists of the missing calls of a block
  DevBuf kin_tmap;                    // tiles of G^T G that meet the upper triangle
  // lmm_batch_d on PLINK blocks in row chunks: the digit combine and the per-SNP stage of chunk c on a side stream beside the
  // int8 product of chunk c + 1 (overlap_*)
  hipStream_t ov_stream = nullptr;
  hipEvent_t ov_ready[16] = {}, ov_done = nullptr;
  // gemma_hip_lmm_batch_pipe_d: the product of block i + 1 on one CU partition beside the combine + per-SNP stage of block i on the
  // other (xp_*).  What both stages of a block touch exists twice (A / C / mean / rowsur alternate between the live members of
  // this struct and `shadow`); what only one stream touches exists once.
  struct XPipe {
    hipStream_t P = nullptr, Q = nullptr;
    hipEvent_t in_ready = nullptr, prod_done[2] = {}, post_done[2] = {};
    hipEvent_t ingest_done = nullptr; // recorded on P behind the ingest of the last block handed in (the caller's stream waits for it)
    bool ingest_valid = false;
    bool post_valid[2] = {false, false};
    unsigned long long count = 0;
    bool pending = false;
    int cus = -1; // partition the streams were made for
    DevBuf shadow_A, shadow_C, shadow_mean, shadow_rowsur;
  } xp;
  int kin_tmap_tm = 0, kin_tmap_tn = 0, kin_tmap_count = 0;

  // lmm state
  bool lmm_active = false;
  gemma_lmm_cfg cfg;
  const double *U = nullptr, *eval = nullptr, *Uty = nullptr; // device
  DevBuf own_U, own_eval, own_Uty, own_UtW, UtWt, idx_map;
  size_t ni_total = 0; // PLINK rows cover this many individuals (0 = n)
  bool have_map = false;
  DevBuf X, UtX, stage_in, stage_out, carry;
  DevBuf grid_R, grid_F, grid_T; // fixed-lambda table (lmm_grid.hip.h)
  DevBuf cheb_R, cheb_F, cheb_T, cheb_slots, cheb_list, cheb_count, cheb_D, cheb_Ck, cheb_Gk, cheb_Lk, cheb_iv, cheb_dends, cheb_res; // bracket-interval series
  double cheb_mid[ASSOC_MAX_REGION], cheb_inv_half[ASSOC_MAX_REGION];
  GridGeom cheb_geom;
  DevBuf table_P; // K-slice partial sums of the table products (table_v2_kernel)
  DevBuf gxe_env, gxe_UtWt, gxe_Z, gxe_UtZ, gxe_flip; // GXE variants
  bool gxe_ready = false;
  double gxe_lnbeta = 0.0;
  DevBuf mv_Yt, mv_out; // multivariate LMM: U^T Y transposed (d x n)
  DevBuf mv_scratch;    // Newton-Raphson tables of the run-time kernel, one slab per workgroup
  bool mv_ready = false, mv_gxe = false;
  size_t mv_d = 0;
  MvArgs mv_proto;
  DevBuf i8_Bt, i8_q, i8_qinv, i8_cmax, i8_A, i8_C, i8_mean; // exact int8-digit U^T x (i8gemm.hip.h)
  unsigned long long cheb_qmask = 0; // bit k: tabulated interval k is in Q form (ends at or below lambda = 1e-3)
  // (tile_m, tile_n) per workgroup of the records kernel: the cross-XCD raster (i8gemm_sparse2.hip.h).  One map per launch shape,
  // each in its OWN buffer, built once (a block cut into row chunks has two shapes -- full chunks and the last one -- and
  // a single slot was rebuilt, with a stream synchronisation and a blocking copy, twice per batch; a map in use by a kernel on
  // another stream could be overwritten).  The host copy stays alive for the asynchronous upload.
  struct RasterSlot {
    DevBuf dev;
    std::vector<int2> host;
    int tm = 0, tn = 0, rb = 0, xcds = 8;
    unsigned long long used = 0;
  } i8_raster[6];
  unsigned long long i8_raster_clock = 0;
  DevBuf i8_surlist;           // per row: count + up to SUR_MAX individuals the sparse mask operand dropped
  DevBuf i8_colsum;            // column sums of U from its digit planes (fixed-point dosage path)
  bool i8_colsum_ready = false;
  int last_utx_path = 0;       // what the last U^T x took: 0 fp64 GEMM, 1 int8 hard calls, 2 int8 dosages k/100, 3 int8 dosages k/1000
  gemma_utx_kernel_info last_utx_kernel = {}; // the matrix kernel that product launched (gemma_hip_dbg_last_utx_kernel)
  long i8_flag_at = -1;                       // int index of the any-missing flag of the last records product in i8_rowsur (-1: none)
  bool i8_ready = false;
  size_t i8_ldk = 0, i8_npad = 0;
  int i8_digits = I8_DIGITS;
  GridGeom grid_geom;
  int carry_flip = 0;
  AssocArgs assoc_proto;

  // linear model (-lm) state
  bool lm_active = false;
  LmArgs lm_proto;
  DevBuf lm_Wt, lm_y, lm_small;

  // misc scratch
  DevBuf scratch;

  // pipelined host-block path (lmm_batch_submit / _collect): two pinned staging slots, a copy and a compute stream
  struct PipeSlot {
    void *pin_in = nullptr, *pin_out = nullptr;
    size_t pin_in_cap = 0, pin_out_cap = 0;
    DevBuf dev_in, dev_out;
    hipEvent_t h2d = nullptr, done = nullptr;
    size_t l = 0;
    bool busy = false;
  } pipe[2];
  hipStream_t pipe_copy = nullptr, pipe_comp = nullptr;
  int pipe_head = 0, pipe_count = 0; // oldest busy slot, number in flight

  // device-resident chain (kin_end_keep -> eigh_kept_K -> lmm_setup_kept) and the communicator
  DevBuf kept_K, kept_UE; // K: ni_total^2; UE: U (n^2) followed by eval (n) -- one buffer, one broadcast
  size_t kept_K_n = 0, kept_n = 0;
  double kept_trace = 0.0;
  Comm comm;
} g_ctx;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_ctx.last_error = buf;
  if (g_ctx.verbose) fprintf(stderr, "gemma_hip: %s\n", buf);
  return code;
}

#define HIPCHK(expr)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(GEMMA_HIP_ERUNTIME, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                                      \
  } while (0)

#define NEED_INIT()                                                                              \
  do {                                                                                           \
    if (!g_ctx.inited) {                                                                         \
      int rc_ = gemma_hip_init(-1, 0);                                                           \
      if (rc_ != GEMMA_HIP_OK) return rc_;                                                       \
    }                                                                                            \
  } while (0)

struct ProfScope {
  int stage;
  hipStream_t s;
  hipEvent_t a = nullptr, b = nullptr;
  ProfScope(int st, hipStream_t stream) : stage(st), s(stream) {
    if (g_ctx.profiling) {
      if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess)
        (void)hipEventRecord(a, s);
      else
        a = b = nullptr;
    }
  }
  ~ProfScope() {
    if (a && b) {
      (void)hipEventRecord(b, s);
      g_ctx.prof[stage].ev.emplace_back(a, b);
    }
  }
};

int prof_collect(int stage) {
  StageProf &p = g_ctx.prof[stage];
  for (auto &pr : p.ev) {
    (void)hipEventSynchronize(pr.second);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
      p.acc_ms += ms;
      p.launches += 1;
    }
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  p.ev.clear();
  return GEMMA_HIP_OK;
}

inline hipStream_t S(void *s) { return reinterpret_cast<hipStream_t>(s); }

} // namespace

static void pipe_release(); // pipelined host-block path, defined with lmm_batch_submit
static void xp_release();   // two-block product / post pipeline, defined with lmm_batch_pipe_d
static int xp_flush_fwd(hipStream_t s);
static void raster_release() {
  for (auto &r : g_ctx.i8_raster) {
    r.dev.release();
    r.host.clear();
    r.tm = r.tn = r.rb = 0;
    r.used = 0;
  }
}
static void kin_i8_release(); // integer kinship path, defined with kin_begin

// ------------------------------------------------------------------------------ lifetime
extern "C" int gemma_hip_abi_version(void) { return GEMMA_HIP_ABI_VERSION; }

extern "C" const char *gemma_hip_strerror(int code) {
  switch (code) {
  case GEMMA_HIP_OK: return "ok";
  case GEMMA_HIP_EINVAL: return "invalid argument (range error)";
  case GEMMA_HIP_ENODEV: return "no usable gfx950 device";
  case GEMMA_HIP_ENOMEM: return "device memory allocation failed";
  case GEMMA_HIP_ERUNTIME: return "HIP runtime error";
  case GEMMA_HIP_ESTATE: return "call sequence violated";
  case GEMMA_HIP_ENOCONV: return "eigensolver did not converge";
  default: return "unknown error";
  }
}

extern "C" const char *gemma_hip_last_error(void) { return g_ctx.last_error.c_str(); }

extern "C" int gemma_hip_init(int device, int verbose) {
  g_ctx.verbose = verbose;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return fail(GEMMA_HIP_ENODEV, "no HIP device visible (this library has no CPU fallback)");
  }
  if (device >= 0) {
    if (device >= ndev) return fail(GEMMA_HIP_ENODEV, "device %d out of range (%d visible)", device, ndev);
    HIPCHK(hipSetDevice(device));
  }
  int cur = 0;
  HIPCHK(hipGetDevice(&cur));
  HIPCHK(hipGetDeviceProperties(&g_ctx.prop, cur));
  if (strncmp(g_ctx.prop.gcnArchName, "gfx950", 6) != 0)
    return fail(GEMMA_HIP_ENODEV, "device %d is %s; this build targets gfx950 only", cur,
                g_ctx.prop.gcnArchName);
  g_ctx.device = cur;
  g_ctx.inited = true;
  g_ctx.knobs.load();
  gemm_aux_init();
  if (verbose)
    fprintf(stderr, "gemma_hip: device %d %s (%s), %d CUs, %.1f GB\n", cur, g_ctx.prop.name,
            g_ctx.prop.gcnArchName, g_ctx.prop.multiProcessorCount,
            (double)g_ctx.prop.totalGlobalMem / 1e9);
  return GEMMA_HIP_OK;
}

extern "C" void gemma_hip_shutdown(void) {
  if (!g_ctx.inited) return;
  (void)hipDeviceSynchronize();
  for (int s = 0; s < GEMMA_STAGE_COUNT; ++s) prof_collect(s);
  g_ctx.kin_K.release(); g_ctx.kin_X.release(); g_ctx.kin_stage.release();
  g_ctx.qc_G.release(); g_ctx.qc_M.release(); g_ctx.qc_W.release(); g_ctx.qc_O.release();
  kin_i8_release();
  g_ctx.own_U.release(); g_ctx.own_eval.release(); g_ctx.own_Uty.release(); g_ctx.own_UtW.release();
  g_ctx.UtWt.release(); g_ctx.idx_map.release(); g_ctx.X.release(); g_ctx.UtX.release();
  g_ctx.stage_in.release(); g_ctx.stage_out.release(); g_ctx.carry.release(); g_ctx.scratch.release();
  g_ctx.i8_Bt.release(); g_ctx.i8_q.release(); g_ctx.i8_qinv.release(); g_ctx.i8_cmax.release(); g_ctx.i8_A.release(); g_ctx.i8_C.release();
  raster_release();
  g_ctx.mv_Yt.release(); g_ctx.mv_out.release(); g_ctx.mv_scratch.release(); // shutdown without lmm_finish leaked these
  g_ctx.mv_ready = g_ctx.mv_gxe = false;
  g_ctx.i8_mean.release(); g_ctx.i8_meta.release(); g_ctx.i8_rowsur.release(); g_ctx.i8_colsum.release(); g_ctx.i8_surlist.release();
  g_ctx.i8_ready = g_ctx.i8_colsum_ready = false;
  g_ctx.table_P.release(); g_ctx.U_even.release();
  g_ctx.U_even_of = nullptr;
  pipe_release(); // pinned slots, copy stream and events of the pipelined host-block path
  xp_release();
  if (g_ctx.ov_stream) {
    (void)hipStreamDestroy(g_ctx.ov_stream);
    for (auto &e : g_ctx.ov_ready)
      if (e) (void)hipEventDestroy(e);
    if (g_ctx.ov_done) (void)hipEventDestroy(g_ctx.ov_done);
    g_ctx.ov_stream = nullptr;
    for (auto &e : g_ctx.ov_ready) e = nullptr;
    g_ctx.ov_done = nullptr;
  }
  if (g_ctx.kin_copy) {
    (void)hipStreamDestroy(g_ctx.kin_copy);
    if (g_ctx.kin_copied) (void)hipEventDestroy(g_ctx.kin_copied);
    if (g_ctx.kin_ingested) (void)hipEventDestroy(g_ctx.kin_ingested);
    g_ctx.kin_copy = nullptr;
    g_ctx.kin_copied = g_ctx.kin_ingested = nullptr;
    g_ctx.kin_ingested_valid = false;
  }
  g_ctx.kin_active = g_ctx.lmm_active = false;
  g_ctx.kept_K.release(); g_ctx.kept_UE.release();
  g_ctx.kept_K_n = g_ctx.kept_n = 0;
  g_ctx.comm.finalize();
  gemm_aux_destroy();
  eigh_tu_shutdown();
  g_ctx.inited = false;
}

extern "C" int gemma_hip_device_info(char *name, size_t len, int *n_cu, size_t *hbm_bytes) {
  NEED_INIT();
  if (name && len) {
    snprintf(name, len, "%s (%s)", g_ctx.prop.name, g_ctx.prop.gcnArchName);
  }
  if (n_cu) *n_cu = g_ctx.prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = g_ctx.prop.totalGlobalMem;
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_profile_enable(int on) {
  g_ctx.profiling = (on != 0);
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_profile_read(int stage, double *total_ms, long *launches, int reset) {
  NEED_INIT();
  if (stage < 0 || stage >= GEMMA_STAGE_COUNT) return fail(GEMMA_HIP_EINVAL, "bad stage %d", stage);
  prof_collect(stage);
  if (total_ms) *total_ms = g_ctx.prof[stage].acc_ms;
  if (launches) *launches = g_ctx.prof[stage].launches;
  if (reset) {
    g_ctx.prof[stage].acc_ms = 0.0;
    g_ctx.prof[stage].launches = 0;
  }
  return GEMMA_HIP_OK;
}

// ------------------------------------------------------------------------------ GEMM
static int check_gemm(char ta, char tb, size_t M, size_t N, size_t K, size_t lda, size_t ldb,
                      size_t ldc) {
  const bool tA = (ta == 'T' || ta == 't'), tB = (tb == 'T' || tb == 't');
  if (!tA && ta != 'N' && ta != 'n') return fail(GEMMA_HIP_EINVAL, "dgemm: bad TransA '%c'", ta);
  if (!tB && tb != 'N' && tb != 'n') return fail(GEMMA_HIP_EINVAL, "dgemm: bad TransB '%c'", tb);
  const size_t a_cols = tA ? M : K, b_cols = tB ? K : N;
  if (lda < a_cols || ldb < b_cols || ldc < N)
    return fail(GEMMA_HIP_EINVAL, "Range error in dgemm (lda=%zu ldb=%zu ldc=%zu for M=%zu N=%zu K=%zu)",
                lda, ldb, ldc, M, N, K);
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_dgemm_d(char ta, char tb, size_t M, size_t N, size_t K, double alpha,
                                 const double *A, size_t lda, const double *B, size_t ldb,
                                 double beta, double *C, size_t ldc, void *stream) {
  NEED_INIT();
  int rc = check_gemm(ta, tb, M, N, K, lda, ldb, ldc);
  if (rc) return rc;
  if (M == 0 || N == 0) return GEMMA_HIP_OK;
  ProfScope ps(GEMMA_STAGE_UTX_GEMM, S(stream));
  HIPCHK(launch_dgemm(ta, tb, (long)M, (long)N, (long)K, alpha, A, (long)lda, B, (long)ldb, beta, C,
                      (long)ldc, false, false, S(stream)));
  return GEMMA_HIP_OK;
}

extern "C" int gemma_hip_dgemm(char ta, char tb, size_t M, size_t N, size_t K, double alpha,
                               const double *A, size_t lda, const double *B, size_t ldb,
                               double beta, double *C, size_t ldc) {
  NEED_INIT();
  int rc = check_gemm(ta, tb, M, N, K, lda, ldb, ldc);
  if (rc) return rc;
  if (M == 0 || N == 0) return GEMMA_HIP_OK;
  const bool tA = (ta == 'T' || ta == 't'), tB = (tb == 'T' || tb == 't');
  const size_t a_rows = tA ? K : M, b_rows = tB ? N : K;
  const size_t a_bytes = a_rows * lda * 8, b_bytes = b_rows * ldb * 8, c_bytes = M * ldc * 8;
  DevBuf dA, dB, dC;
  if (dA.reserve(a_bytes ? a_bytes : 8) || dB.reserve(b_bytes ? b_bytes : 8) || dC.reserve(c_bytes)) {
    dA.release(); dB.release(); dC.release();
    return fail(GEMMA_HIP_ENOMEM, "dgemm: cannot allocate %zu bytes", a_bytes + b_bytes + c_bytes);
  }
  // the last row of a strided host view may be shorter than ld: copy row-wise via 2D copies
  auto h2d = [&](void *d, const double *h, size_t rows, size_t cols, size_t ld) -> hipError_t {
    if (rows == 0 || cols == 0) return hipSuccess;
    return hipMemcpy2D(d, ld * 8, h, ld * 8, cols * 8, rows, hipMemcpyHostToDevice);
  };
  hipError_t e = h2d(dA.p, A, a_rows, tA ? M : K, lda);
  if (e == hipSuccess) e = h2d(dB.p, B, b_rows, tB ? K : N, ldb);
  if (e == hipSuccess && beta != 0.0) e = h2d(dC.p, C, M, N, ldc);
  if (e == hipSuccess)
    e = launch_dgemm(ta, tb, (long)M, (long)N, (long)K, alpha, dA.as<double>(), (long)lda,
                     dB.as<double>(), (long)ldb, beta, dC.as<double>(), (long)ldc, false, false, 0);
  if (e == hipSuccess) e = hipMemcpy2D(C, ldc * 8, dC.p, ldc * 8, N * 8, M, hipMemcpyDeviceToHost);
  dA.release(); dB.release(); dC.release();
  if (e != hipSuccess) return fail(GEMMA_HIP_ERUNTIME, "dgemm: %s", hipGetErrorString(e));
  return GEMMA_HIP_OK;
}

// The stages of the C ABI, one file each (textual parts of THIS translation unit -- they share g_ctx; VERDICT r5 item 9: cut along the
// stages for reading.  A split into separately compiled units needs the argument structs that Ctx holds (AssocArgs, MvArgs, LmArgs,
// GridGeom, Comm) cut loose from the kernels defined beside them first: every unit would otherwise instantiate every kernel.)
#include "abi_kinship.inc.h"
#include "abi_eigen_qc.inc.h"
#include "abi_lmm_stage.inc.h"
#include "abi_utx.inc.h"
#include "abi_lmm_batch.inc.h"
#include "abi_mvlmm.inc.h"
#include "abi_gxe_lm.inc.h"
#include "abi_kept_comm.inc.h"
