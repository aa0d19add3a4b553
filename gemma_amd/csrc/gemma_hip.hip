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

// ------------------------------------------------------------------------------ kinship
extern "C" int gemma_hip_kin_begin(size_t n_total, int k_mode) {
  NEED_INIT();
  if (n_total == 0) return fail(GEMMA_HIP_EINVAL, "kin_begin: n_total == 0");
  if (k_mode != 1 && k_mode != 2) return fail(GEMMA_HIP_EINVAL, "kin_begin: k_mode %d", k_mode);
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
  if (g_ctx.kin_stage.reserve(bytes)) return fail(GEMMA_HIP_ENOMEM, "kin_add: staging %zu bytes", bytes);
  // last row may be shorter than ld in the caller's buffer
  const size_t width = need * esz;
  HIPCHK(hipMemcpy2D(g_ctx.kin_stage.p, ld * esz, geno, ld * esz, width, rows, hipMemcpyHostToDevice));
  return gemma_hip_kin_add_d(kind, g_ctx.kin_stage.p, l, ld, nullptr);
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
  DevBuf dG, dM, dW, dO;
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
  cleanup();
  if (e != hipSuccess) return fail(GEMMA_HIP_ERUNTIME, "snp_qc: %s", hipGetErrorString(e));
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

// digits of U in the exact int8 product (i8gemm.hip.h): 7, or 6 from n = 16384 up where the 2^-47 rounding of U stays at
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

// ---- exact int8-digit U^T x (i8gemm.hip.h): buffers, the product on an already packed left factor, ingest variants
struct I8Dims {
  size_t n, ldk, npad, lpad, mrows;
  int fuse, digits, nplanes;
  int mdrop; // 1: the 7g6m form -- plane 0 (digit 0 alone) carries the genotype product only
};
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
  if (g_ctx.i8_meta.reserve(total * sizeof(uint4)) || g_ctx.i8_rowsur.reserve(d.lpad * sizeof(int)))
    return fail(GEMMA_HIP_ENOMEM, "lmm_batch: mask words of the sparse product");
  HIPCHK(hipMemsetAsync(g_ctx.i8_rowsur.p, 0, d.lpad * sizeof(int), s));
  if (mode == 2)
    hipLaunchKernelGGL(sparse2_meta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g_ctx.i8_A.as<int8_t>(),
                       (long)d.lpad, (long)d.ldk, g_ctx.i8_meta.as<uint4>(), g_ctx.i8_rowsur.as<int>());
  else
    hipLaunchKernelGGL(sparse_meta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g_ctx.i8_A.as<int8_t>(),
                       (long)d.lpad, (long)d.ldk, g_ctx.i8_meta.as<uint4>(), g_ctx.i8_rowsur.as<int>());
  HIPCHK(hipGetLastError());
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
    } else if (d.mdrop) {
      // 7g6m: planes 1..3 (digit pairs {2,1} {4,3} {6,5}) with both products, then plane 0 (digit 0) with the genotype product alone
      g2.plane0 = 1;
      hipLaunchKernelGGL(i8gemm_sparse2_r16_kernel, dim3((unsigned)(g2.tiles_m * g2.tiles_n), (unsigned)(d.nplanes - 1)), dim3(512),
                         S2_R16_LDS, s, g2);
      g2.plane0 = 0;
      hipLaunchKernelGGL(i8gemm_sparse2_r16_g_kernel, dim3((unsigned)(g2.tiles_m * g2.tiles_n), 1u), dim3(512), S2_R16_LDS, s, g2);
    } else {
      hipLaunchKernelGGL(i8gemm_sparse2_r16_kernel, dim3((unsigned)(g2.tiles_m * g2.tiles_n), (unsigned)d.nplanes), dim3(512),
                         S2_R16_LDS, s, g2);
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
                     1.0, d.fuse, d.digits, sur_cnt, sur_list, g_ctx.U, (long)d.n, d.mdrop);
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
