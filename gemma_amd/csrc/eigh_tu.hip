// The symmetric eigensolver as its own translation unit (see eigh_tu.h): one-stage and two-stage reductions, divide and
// conquer, back-transformations (eigh.hip.h, eigh2.hip.h), and the host-pointer stage diagnostics the GPU tests call.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gemma_hip.h"
#include "eigh.hip.h"
#include "eigh_tu.h"

namespace gemma_hip {

namespace {
struct TuBuf { // device scratch of a diagnostic call
  void *p = nullptr;
  bool reserve(size_t bytes) {
    if (hipMalloc(&p, std::max<size_t>(bytes, 8)) != hipSuccess) {
      (void)hipGetLastError();
      p = nullptr;
      return false;
    }
    return true;
  }
  ~TuBuf() {
    if (p) (void)hipFree(p);
  }
};
} // namespace

int eigh_device_x(double *G, long n, double *U, double *eval, hipStream_t s, std::string &msg, const EighShard *sh) {
  gemm_aux_init();
  return eigh_device(G, n, U, eval, s, msg, sh);
}

void eigh_abort_x(long n, hipStream_t s, const EighShard *sh) { eigh_collective_abort(n, s, sh); }

void eigh_last_stages(double *t8) {
  for (int i = 0; i < 8; ++i) t8[i] = g_eig_last[i];
}

void eigh_tu_shutdown() {
  gemm_aux_destroy();
  eig2_lookahead_destroy();
  eig_pool().keep = false;
  (void)eig_pool().drop_idle();
}

int eigh_reserve_x(long n, std::string &msg) {
  if (n < 1) {
    msg = "eigh_reserve: n < 1";
    return GEMMA_HIP_EINVAL;
  }
  const long ne = eig_effective_n(n); // an odd order runs embedded in n + 1
  static long reserved_for = 0;
  if (reserved_for != ne) (void)eig_pool().drop_idle(); // another order: its blocks would only sit beside the new ones
  reserved_for = ne;
  eig_pool().keep = true;
  EigWs ws;
  ws.n = ne;
  Eig2Ws w2;
  const bool ok = eig_alloc_all(ne, ws, w2);
  ws.release(); // into the pool (keep is on); a partial set stays too: the solve reports the shortage itself
  if (!ok) {
    msg = "eigh_reserve: cannot allocate the eigensolver workspace (about 5 n^2 doubles)";
    return GEMMA_HIP_ENOMEM;
  }
  return 0;
}

size_t eigh_release_x() {
  eig_pool().keep = false;
  return eig_pool().drop_idle();
}

size_t eigh_pool_idle_bytes_x() { return eig_pool().idle_bytes(); }

// Householder tridiagonalisation only: G (host, n x n) -> d[n], e[n-1], tau[n], VT (n x n, row j = u_j)
int dbg_tridiag_x(const double *G, size_t n, double *d, double *e, double *tau, double *VT, std::string &msg) {
  gemm_aux_init();
  EigWs ws;
  TuBuf dG;
  const size_t nn = n * n;
  if (!dG.reserve(nn * 8)) return GEMMA_HIP_ENOMEM;
  bool ok = ws.get(ws.VT, nn) && ws.get(ws.WT, (size_t)EIG_NB * n) && ws.get(ws.xcol, n + 2) && ws.get(ws.p, n) &&
            ws.get(ws.ab, 2 * EIG_NB) && ws.get(ws.ssbuf, n / TD_ROWS + 2) && ws.get(ws.dotbuf, n / TD_ROWS + 2) &&
            ws.get(ws.wtmp, n) && ws.get(ws.d, n) && ws.get(ws.e, n) && ws.get(ws.tau, n);
  {
    const char *ev = getenv("GEMMA_HIP_EIGH_SYMV");
    if (ok && (n & 1) == 0 && !(ev && ev[0] == '0')) {
      const size_t nseg = (n + TS_SEG_MIN - 1) / TS_SEG_MIN, nstrip = (n + TS_STRIP - 1) / TS_STRIP;
      ok = ws.get(ws.rowP, nseg * n) && ws.get(ws.colP, nstrip * n);
    }
  }
  int rc = ok ? 0 : GEMMA_HIP_ENOMEM;
  if (!rc && hipMemcpy(dG.p, G, nn * 8, hipMemcpyHostToDevice) != hipSuccess) rc = GEMMA_HIP_ERUNTIME;
  if (!rc) rc = eig_tridiagonalize(static_cast<double *>(dG.p), (long)n, ws, 0, msg);
  if (!rc && hipDeviceSynchronize() != hipSuccess) rc = GEMMA_HIP_ERUNTIME;
  if (!rc) {
    (void)hipMemcpy(d, ws.d, n * 8, hipMemcpyDeviceToHost);
    if (n > 1) (void)hipMemcpy(e, ws.e, (n - 1) * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(tau, ws.tau, n * 8, hipMemcpyDeviceToHost);
    if (VT) (void)hipMemcpy(VT, ws.VT, nn * 8, hipMemcpyDeviceToHost);
  }
  ws.release();
  return rc;
}

// two-stage reduction only (eigh2.hip.h): G (host, n x n, n even, n >= 384) -> band after stage 1 (n x 129: row j holds
// B(j .. j+128, j)) and the tridiagonal d[n], e[n-1] after the bulge chase
int dbg_eigh2_x(const double *G, size_t n, double *band, double *d, double *e, std::string &msg) {
  gemm_aux_init();
  if (n < 3 * (size_t)E2_B || (n & 1)) {
    msg = "n must be even and >= " + std::to_string(3 * E2_B);
    return GEMMA_HIP_EINVAL;
  }
  EigWs ws;
  Eig2Ws w2;
  TuBuf dG;
  const size_t nn = n * n;
  if (!dG.reserve(nn * 8)) return GEMMA_HIP_ENOMEM;
  bool ok = ws.get(ws.VT, nn) && ws.get(ws.WT, (size_t)EIG_NB * n) && ws.get(ws.d, n) && ws.get(ws.e, n) &&
            ws.get(ws.tau, n) && ws.get(ws.S, (size_t)EIG_NB * EIG_NB) && ws.get(ws.T, (size_t)EIG_NB * EIG_NB) &&
            ws.get(ws.Tall, ((n + EIG_NB - 1) / EIG_NB) * EIG_NB * EIG_NB) && eig2_alloc((long)n, ws, w2);
  int rc = ok ? 0 : GEMMA_HIP_ENOMEM;
  if (!rc && hipMemcpy(dG.p, G, nn * 8, hipMemcpyHostToDevice) != hipSuccess) rc = GEMMA_HIP_ERUNTIME;
  if (!rc) rc = eig2_sy2sb(static_cast<double *>(dG.p), (long)n, ws, w2, 0, msg);
  if (!rc && hipDeviceSynchronize() != hipSuccess) rc = GEMMA_HIP_ERUNTIME;
  if (!rc && band &&
      hipMemcpy2D(band, (E2_B + 1) * 8, w2.Bd, E2_LDB * 8, (E2_B + 1) * 8, n, hipMemcpyDeviceToHost) != hipSuccess)
    rc = GEMMA_HIP_ERUNTIME;
  if (!rc) rc = eig2_sb2st((long)n, ws, w2, 0, msg);
  if (!rc && hipDeviceSynchronize() != hipSuccess) rc = GEMMA_HIP_ERUNTIME;
  if (!rc) {
    (void)hipMemcpy(d, ws.d, n * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(e, ws.e, (n - 1) * 8, hipMemcpyDeviceToHost);
  }
  ws.release();
  if (rc && msg.empty()) msg = hipGetErrorString(hipGetLastError());
  return rc;
}

// divide-and-conquer on a symmetric tridiagonal (host d[n], e[n-1]) -> w[n] ascending, ZT (n x n, row k = eigenvector k)
int dbg_stedc_x(const double *d, const double *e, size_t n, double *w, double *ZT, std::string &msg) {
  gemm_aux_init();
  EigWs ws;
  const size_t nn = n * n;
  double *QA = nullptr, *QB = nullptr;
  bool ok = ws.get(QA, nn) && ws.get(QB, nn) && ws.get(ws.d, n) && ws.get(ws.e, n) && ws.get(ws.Delta, nn) &&
            ws.get(ws.Wk, nn) && ws.get(ws.zbuf, n) && ws.get(ws.dl, n) && ws.get(ws.w, n) && ws.get(ws.lam, n) &&
            ws.get(ws.zhat, n) && ws.get(ws.dphys, n) && ws.get(ws.ibuf, 2 * n + 64) && ws.get(ws.ibuf2, 4 * n + 64) && ws.get(ws.info, 1) &&
            ws.get(ws.rot, n);
  int rc = ok ? 0 : GEMMA_HIP_ENOMEM;
  std::vector<double> hd(d, d + n), he(e, e + (n > 1 ? n - 1 : 0)), dphys;
  if (he.empty()) he.push_back(0.0);
  double *Z = nullptr;
  if (!rc && n == 1) {
    w[0] = d[0];
    ZT[0] = 1.0;
    ws.release();
    return 0;
  }
  if (!rc) rc = eig_stedc((long)n, hd, he, QA, QB, ws, 0, &Z, dphys, msg);
  if (!rc) {
    std::vector<int> perm(n);
    for (size_t i = 0; i < n; ++i) perm[i] = (int)i;
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int c) { return dphys[a] < dphys[c]; });
    std::vector<double> tmp(nn);
    if (hipMemcpy(tmp.data(), Z, nn * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = GEMMA_HIP_ERUNTIME;
    for (size_t t = 0; t < n && !rc; ++t) {
      w[t] = dphys[perm[t]];
      memcpy(ZT + t * n, tmp.data() + (size_t)perm[t] * n, n * 8);
    }
  }
  ws.release();
  return rc;
}

} // namespace gemma_hip
