"""Host-side mirror of the GEMMA interfaces on the kinship + univariate-LMM path, over the C ABI.

Names, argument meaning and error behaviour follow the reference (file:line relative to the
GEMMA tree) so that the parity tests read like the reference's own tests:

    fast_dgemm            src/fastblas.h:34-36
    CenterMatrix          src/mathfunc.cpp:147-177
    EigenDecomp_Zeroed    src/lapack.cpp:260-291
    CalcUtX               src/mathfunc.cpp:504-506
    CalcKin / BimbamKin / PlinkKin   src/param.cpp:1300-1321, src/gemma_io.cpp:1418-1738
    CalcLambdaNull / CalcPve         src/lmm.cpp:2143-2205
    class LMM  (CopyFromParam fields, Analyze*, WriteFiles)   src/lmm.h:49-125

Arrays may be numpy arrays (host entry points; the library stages through HBM) or torch CUDA
tensors (device entry points on torch's current stream: torch is only the allocator/stream here).
All compute happens in gemma_amd/libgemma_hip.so; nothing in this module has a CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _lib as L

SUMSTAT_DTYPE = np.dtype([(k, "f8") for k in
                          ("beta", "se", "lambda_remle", "lambda_mle", "p_wald", "p_lrt", "p_score",
                           "logl_H1")])
LMM_BATCH_SIZE = 20000  # src/lmm.h:33
K_BATCH_SIZE = 20000  # src/param.h:32


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _np64(a, name):
    if not isinstance(a, np.ndarray) or a.dtype != np.float64:
        raise TypeError("%s must be a float64 numpy array" % name)
    if a.ndim == 2 and a.strides[1] != 8:
        raise ValueError("%s must be row-major with unit column stride" % name)
    return a


def _ld(a):
    return a.shape[1] if a.ndim == 2 and a.shape[0] <= 1 else (a.strides[0] // 8 if a.ndim == 2 else a.shape[0])


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _tld(t):
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError("device matrices must be 2-D row-major")
    return t.stride(0) if t.shape[0] > 1 else t.shape[1]


def init(device=-1, verbose=0):
    L.check(L.lib().gemma_hip_init(device, verbose), "gemma_hip_init")


def device_info():
    name = C.create_string_buffer(256)
    cu = C.c_int()
    mem = C.c_size_t()
    L.check(L.lib().gemma_hip_device_info(name, 256, C.byref(cu), C.byref(mem)), "device_info")
    return name.value.decode(), cu.value, mem.value


def profile_enable(on=True):
    L.check(L.lib().gemma_hip_profile_enable(1 if on else 0), "profile_enable")


def profile_read(stage, reset=False):
    ms = C.c_double()
    n = C.c_long()
    L.check(L.lib().gemma_hip_profile_read(stage, C.byref(ms), C.byref(n), 1 if reset else 0), "profile_read")
    return ms.value, n.value


UTX_PATHS = {0: "fp64 MFMA GEMM", 1: "int8 digits, hard calls", 2: "int8 digits, dosages k/100", 3: "int8 digits, dosages k/1000"}


def last_utx_path():
    """What the last U^T x (LMM.batch, LMM.dbg_utx) ran on: a key of UTX_PATHS."""
    p = C.c_int()
    L.check(L.lib().gemma_hip_dbg_last_utx_path(C.byref(p)), "dbg_last_utx_path")
    return p.value


def last_utx_kernel():
    """The matrix kernel the last U^T x launched (gemma_hip_dbg_last_utx_kernel): dict(variant, rows, digits, fuse, raster,
    launches, name) -- name is the kernel symbol as rocprofv3 prints it."""
    k = L.UtxKernelInfo()
    L.check(L.lib().gemma_hip_dbg_last_utx_kernel(C.byref(k)), "dbg_last_utx_kernel")
    return {"variant": k.variant, "rows": k.rows, "digits": k.digits, "fuse": k.fuse, "raster": k.raster,
            "launches": k.launches, "name": k.name.decode()}


def last_block_missing():
    """gemma_hip_dbg_last_block_missing: 1 / 0 = the last records product's block held / did not hold a missing call (a block without
    one takes the genotype product alone); -1 = the complete-block form is off or no such product ran."""
    a = C.c_int()
    L.check(L.lib().gemma_hip_dbg_last_block_missing(C.byref(a)), "dbg_last_block_missing")
    return a.value


def reload_env():
    """Have the library re-read its GEMMA_HIP_* switches (it reads them once per setup, never per launch)."""
    L.check(L.lib().gemma_hip_reload_env(), "reload_env")


# ----------------------------------------------------------------------------- B2
def fast_dgemm(TransA, TransB, alpha, A, B, beta, Cm):
    """C = alpha*op(A)*op(B) + beta*C (row-major).  Shape mismatch -> GemmaHipError(EINVAL), the
    reference's fail_msg("Range error in dgemm") (src/fastblas.cpp:207)."""
    ta, tb = TransA.upper()[0], TransB.upper()[0]
    M, N = Cm.shape
    if ta not in "NT" or tb not in "NT":
        raise L.GemmaHipError(L.EINVAL, "fast_dgemm", "bad transpose flag")
    Ka = A.shape[0] if ta == "T" else A.shape[1]
    Ma = A.shape[1] if ta == "T" else A.shape[0]
    Kb = B.shape[1] if tb == "T" else B.shape[0]
    Nb = B.shape[0] if tb == "T" else B.shape[1]
    if Ma != M or Nb != N or Ka != Kb:
        raise L.GemmaHipError(L.EINVAL, "fast_dgemm", "Range error in dgemm")
    if _is_torch(Cm):
        rc = L.lib().gemma_hip_dgemm_d(ta.encode(), tb.encode(), M, N, Ka, alpha, C.c_void_p(A.data_ptr()),
                                       _tld(A), C.c_void_p(B.data_ptr()), _tld(B), beta,
                                       C.c_void_p(Cm.data_ptr()), _tld(Cm), _stream())
    else:
        _np64(A, "A"); _np64(B, "B"); _np64(Cm, "C")
        rc = L.lib().gemma_hip_dgemm(ta.encode(), tb.encode(), M, N, Ka, alpha, _ptr(A), _ld(A), _ptr(B),
                                     _ld(B), beta, _ptr(Cm), _ld(Cm))
    L.check(rc, "fast_dgemm")
    return Cm


fast_eigen_dgemm = fast_dgemm  # src/fastblas.h:37-39 (historical alias)


# ----------------------------------------------------------------------------- B3
def CenterMatrix(G):
    """In place, src/mathfunc.cpp:147-177."""
    n = G.shape[0]
    if _is_torch(G):
        L.check(L.lib().gemma_hip_center_d(C.c_void_p(G.data_ptr()), n, _stream()), "CenterMatrix")
    else:
        _np64(G, "G")
        if not G.flags.c_contiguous:
            raise ValueError("G must be contiguous")
        L.check(L.lib().gemma_hip_center(_ptr(G), n), "CenterMatrix")
    return G


def EigenDecomp_Zeroed(G, U, eval_):
    """G is destroyed; U gets eigenvectors in columns, eval_ ascending with values < 1e-10 zeroed.
    Returns trace_G = mean(eval) (src/lapack.cpp:260-291)."""
    n = G.shape[0]
    tr = C.c_double()
    if _is_torch(G):
        rc = L.lib().gemma_hip_eigh_d(C.c_void_p(G.data_ptr()), n, C.c_void_p(U.data_ptr()),
                                      C.c_void_p(eval_.data_ptr()), C.byref(tr), _stream())
    else:
        _np64(G, "G"); _np64(U, "U"); _np64(eval_, "eval")
        rc = L.lib().gemma_hip_eigh(_ptr(G), n, _ptr(U), _ptr(eval_), C.byref(tr))
    L.check(rc, "EigenDecomp_Zeroed")
    return tr.value


def eigh_reserve(n):
    """gemma_hip_eigh_reserve: allocate the eigensolver's workspace of order n ahead of the solve and keep the buffers of every later
    solve in the library's pool (csrc/eigh.hip.h, EigPool) until eigh_release()."""
    L.check(L.lib().gemma_hip_eigh_reserve(int(n)), "eigh_reserve")


def eigh_release():
    """gemma_hip_eigh_release: hand the pool's idle buffers back; returns the bytes freed."""
    b = C.c_size_t()
    L.check(L.lib().gemma_hip_eigh_release(C.byref(b)), "eigh_release")
    return b.value


def EigenDecomp_Zeroed_sharded(G, U, eval_):
    """EigenDecomp_Zeroed as a COLLECTIVE over the library's communicator (gemma_amd.dist.native_comm_init): every rank passes
    the same device matrix G (destroyed) and receives the same (U, eval_); the two back-transformations are shared out by
    eigenvector.  torch tensors on the device only."""
    n = G.shape[0]
    tr = C.c_double()
    rc = L.lib().gemma_hip_eigh_sharded_d(C.c_void_p(G.data_ptr()), n, C.c_void_p(U.data_ptr()),
                                          C.c_void_p(eval_.data_ptr()), C.byref(tr), _stream())
    L.check(rc, "EigenDecomp_Zeroed_sharded")
    return tr.value


def CalcUtX(U, X):
    """UtX = U^T X (src/mathfunc.cpp:504-506).  X: n x m (or n,) numpy."""
    X2 = np.ascontiguousarray(X.reshape(X.shape[0], -1), dtype=np.float64)
    n, m = X2.shape
    out = np.zeros((n, m))
    L.check(L.lib().gemma_hip_calc_utx(_ptr(_np64(U, "U")), _ptr(X2), n, m, _ptr(out)), "CalcUtX")
    return out.reshape(X.shape)


# ----------------------------------------------------------------------------- B1
def kin_begin(n_total, k_mode=1):
    L.check(L.lib().gemma_hip_kin_begin(n_total, k_mode), "kin_begin")


def kin_add(geno, geno_kind, l=None, ld=None):
    if _is_torch(geno):
        if l is None:  # individual-major: SNP j is column j (the reference's Xlarge)
            l = geno.shape[1] if geno_kind == L.GENO_F64_IDV_MAJOR else geno.shape[0]
        ld = _tld(geno) if ld is None else ld
        rc = L.lib().gemma_hip_kin_add_d(geno_kind, C.c_void_p(geno.data_ptr()), l, ld, _stream())
    else:
        if geno_kind == L.GENO_F64_IDV_MAJOR:
            l = geno.shape[1] if l is None else l
        else:
            l = geno.shape[0] if l is None else l
        ld = (geno.strides[0] // geno.itemsize) if ld is None else ld
        rc = L.lib().gemma_hip_kin_add(geno_kind, _ptr(geno), l, ld)
    L.check(rc, "kin_add")


def kin_end(K=None):
    ns = C.c_size_t()
    if K is not None and _is_torch(K):
        L.check(L.lib().gemma_hip_kin_end_d(C.c_void_p(K.data_ptr()), C.byref(ns), _stream()), "kin_end")
    else:
        L.check(L.lib().gemma_hip_kin_end(_ptr(K) if K is not None else None, C.byref(ns)), "kin_end")
    return ns.value


# ---- the device-resident chain (include/gemma_hip.h "kept"): K, U, eval never leave the device; with a communicator of several
# ranks kin_end_keep(allreduce=True) is the ONE ncclAllReduce of the SNP-sharded kinship and EigenDecomp_kept_K(sharded=True) the
# collective decomposition -- the flow of tests/cpp/gemma_file_driver.cpp -gpus N and of bench.py --gpus N
def kin_end_keep(allreduce=False):
    """Ends kin_begin / kin_add and KEEPS K on the device; allreduce: sum the ranks' partial kinships first (every rank then holds
    the kinship of all SNPs).  Returns the SNP count (of all ranks with allreduce)."""
    ns = C.c_size_t()
    L.check(L.lib().gemma_hip_kin_end_keep(C.byref(ns), 1 if allreduce else 0), "kin_end_keep")
    return ns.value


def EigenDecomp_kept_K(ni_total, indicator_idv=None, sharded=False):
    """Sub-select (ReadFile_kin's indicator), CenterMatrix, EigenDecomp_Zeroed of the kept K, all on the device; U and eval stay
    there (lmm setup_kept / CalcUtX_kept use them).  sharded: the collective form (every rank holds the same kept K).  Returns
    (eval numpy, trace_G)."""
    ind = None if indicator_idv is None else np.ascontiguousarray(indicator_idv, dtype=np.int32)
    n = int(ni_total if ind is None else int((ind != 0).sum()))
    ev = np.zeros(n)
    tr = C.c_double()
    fn = L.lib().gemma_hip_eigh_kept_K_sharded if sharded else L.lib().gemma_hip_eigh_kept_K
    L.check(fn(_ptr(ind) if ind is not None else None, ni_total, _ptr(ev), C.byref(tr)), "EigenDecomp_kept_K")
    return ev, tr.value


def CalcUtX_kept(X):
    """U^T X on the kept U (host X: n x m or n,; returns numpy of the same shape)."""
    X2 = np.ascontiguousarray(np.asarray(X, dtype=np.float64).reshape(len(X), -1))
    out = np.zeros_like(X2)
    L.check(L.lib().gemma_hip_calc_utx_kept(_ptr(X2), X2.shape[0], X2.shape[1], _ptr(out)), "CalcUtX_kept")
    return out.reshape(np.asarray(X).shape)


def kept_release():
    L.check(L.lib().gemma_hip_kept_release(), "kept_release")


def comm_info():
    """(rank, world, transport) of the library's communicator: transport 0 none, 1 RCCL, 2 the shm test transport."""
    r, w, t = C.c_int(), C.c_int(), C.c_int()
    L.check(L.lib().gemma_hip_comm_info(C.byref(r), C.byref(w), C.byref(t)), "comm_info")
    return r.value, w.value, t.value


def comm_stats():
    """gemma_hip_comm_stats as a dict: calls, 1-GiB pieces and bytes per collective since comm_init (seconds only with
    GEMMA_HIP_COMM_TIMING=1 in the environment)."""
    st = L.CommStats()
    L.check(L.lib().gemma_hip_comm_stats(C.byref(st)), "comm_stats")
    return {k: getattr(st, k) for k, _ in L.CommStats._fields_}


def CalcKin(geno, geno_kind, n_total, k_mode=1, batch=K_BATCH_SIZE):
    """PARAM::CalcKin -> BimbamKin / PlinkKin: streams `geno` (SNP-major rows) in blocks of
    K_BATCH_SIZE SNPs and returns the n_total x n_total kinship matrix (numpy)."""
    kin_begin(n_total, k_mode)
    p = geno.shape[0]
    for s0 in range(0, p, batch):
        kin_add(geno[s0:s0 + batch], geno_kind)
    K = np.zeros((n_total, n_total))
    kin_end(K)
    return K


def SnpQC(geno, geno_kind, indicator_idv, W, maf_level=0.01, miss_level=0.05, hwe_level=0.0, r2_level=0.9999):
    """First-pass SNP filters of ReadFile_geno / ReadFile_bed (src/gemma_io.cpp:639-873 / :876-1064) on device.
    geno: SNP-major rows over ALL individuals (fp64 with NaN, or .bed bytes); W: covariates of the analysed
    individuals.  Returns (indicator_snp, maf, n_miss)."""
    ind = None if indicator_idv is None else np.ascontiguousarray(indicator_idv, dtype=np.int32)
    W = np.ascontiguousarray(W, dtype=np.float64).reshape(len(W), -1)
    n, c = W.shape
    l = geno.shape[0]
    ni_total = n if ind is None else ind.size
    cfg = L.QcCfg(maf_level, miss_level, hwe_level, r2_level)
    out_i = np.zeros(l, dtype=np.int32)
    out_maf = np.zeros(l)
    out_nm = np.zeros(l, dtype=np.uint64)
    L.check(L.lib().gemma_hip_snp_qc(geno_kind, _ptr(geno), l, geno.strides[0] // geno.itemsize,
                                     _ptr(ind) if ind is not None else None, ni_total, _ptr(W), n, c, C.byref(cfg),
                                     _ptr(out_i), _ptr(out_maf), _ptr(out_nm)), "SnpQC")
    return out_i, out_maf, out_nm.astype(np.int64)


def CalcKinLOCO(geno, geno_kind, n_total, chr_of_snp, k_mode=1, batch=K_BATCH_SIZE):
    """-loco for every chromosome at once (torch device tensors inside): the all-SNP kinship plus one
    per-chromosome kinship, combined on device as (ns K - ns_c K_c)/(ns - ns_c).  Returns {chr: K_loco numpy}."""
    import torch
    chr_of_snp = np.asarray(chr_of_snp)
    dev = torch.device("cuda", torch.cuda.current_device())
    Kall = torch.empty((n_total, n_total), dtype=torch.float64, device=dev)
    kin_begin(n_total, k_mode)
    for s0 in range(0, geno.shape[0], batch):
        kin_add(geno[s0:s0 + batch], geno_kind)
    ns_all = kin_end(Kall)
    out = {}
    for ch in sorted(set(chr_of_snp.tolist())):
        sel = np.flatnonzero(chr_of_snp == ch)
        Kc = torch.empty_like(Kall)
        kin_begin(n_total, k_mode)
        g = np.ascontiguousarray(geno[sel])
        for s0 in range(0, g.shape[0], batch):
            kin_add(g[s0:s0 + batch], geno_kind)
        ns_c = kin_end(Kc)
        L.check(L.lib().gemma_hip_kin_loco_d(C.c_void_p(Kall.data_ptr()), ns_all, C.c_void_p(Kc.data_ptr()), ns_c,
                                             n_total, _stream()), "CalcKinLOCO")
        torch.cuda.synchronize()
        out[ch] = Kc.cpu().numpy()
    return out


def WriteMatrix10(M):
    """The text hand-off `-gk` -> `-lmm`: PARAM::WriteMatrix at precision(10) (src/param.cpp:1886-1911) read back
    by ReadFile_kin (src/gemma_io.cpp:1186-1243).  Returns the matrix as the second GEMMA run would see it."""
    flat = np.array([float("%.10g" % v) for v in np.asarray(M, dtype=np.float64).ravel()])
    return flat.reshape(np.asarray(M).shape)


# ----------------------------------------------------------------------------- null model
def CalcLambdaNull(eval_, UtW, Uty, l_min=1e-5, l_max=1e5, n_region=10, trace_G=1.0):
    """Returns dict(l_mle_null, logl_mle_H0, l_remle_null, logl_remle_H0, pve, pve_se, vg, ve):
    src/gemma.cpp:2711-2750."""
    UtW = np.ascontiguousarray(UtW, dtype=np.float64).reshape(len(eval_), -1)
    out = np.zeros(8)
    n, c = UtW.shape
    # converted copies stay bound to names until the call returns (_ptr keeps only the address)
    ev_c = _np64(np.ascontiguousarray(eval_, dtype=np.float64), "eval")
    Uty_c = np.ascontiguousarray(Uty, dtype=np.float64)
    L.check(L.lib().gemma_hip_lmm_null(n, c, _ptr(ev_c), _ptr(UtW), _ptr(Uty_c), l_min, l_max,
                                       n_region, trace_G, _ptr(out)), "CalcLambdaNull")
    keys = ("l_mle_null", "logl_mle_H0", "l_remle_null", "logl_remle_H0", "pve", "pve_se", "vg_remle",
            "ve_remle")
    return dict(zip(keys, out.tolist()))


# ----------------------------------------------------------------------------- B4
class LMM:
    """Mirror of class LMM (src/lmm.h:49-125): the fields CopyFromParam fills (src/lmm.cpp:56-90)
    and the Analyze* drivers.  sumStat is a numpy record array with SUMSTAT's fields."""

    def __init__(self, a_mode=1, l_min=1e-5, l_max=1e5, n_region=10, l_mle_null=0.0, logl_mle_H0=0.0):
        self.a_mode = a_mode
        self.l_min, self.l_max, self.n_region = l_min, l_max, n_region
        self.l_mle_null, self.logl_mle_H0 = l_mle_null, logl_mle_H0
        self.ni_test = 0
        self.n_cvt = 0
        self.time_UtX = 0.0
        self.time_opt = 0.0
        self.sumStat = np.zeros(0, dtype=SUMSTAT_DTYPE)
        self._active = False

    # -- state handling ---------------------------------------------------------------
    def _cfg(self, n, c, plink):
        cfg = L.LmmCfg()
        cfg.a_mode = self.a_mode
        cfg.n, cfg.n_cvt = n, c
        cfg.l_min, cfg.l_max, cfg.n_region = self.l_min, self.l_max, self.n_region
        cfg.l_mle_null, cfg.logl_mle_H0 = self.l_mle_null, self.logl_mle_H0
        cfg.plink_nan_rule = 1 if plink else 0
        return cfg

    def setup(self, U, eval_, UtW, Uty, plink=False):
        n = U.shape[0]
        if _is_torch(U):
            c = UtW.shape[1] if UtW.dim() == 2 else 1
            cfg = self._cfg(n, c, plink)
            self._keep = (U, eval_, UtW, Uty)  # borrowed by the library until finish()
            rc = L.lib().gemma_hip_lmm_setup_d(C.byref(cfg), C.c_void_p(U.data_ptr()),
                                               C.c_void_p(eval_.data_ptr()), C.c_void_p(UtW.data_ptr()),
                                               C.c_void_p(Uty.data_ptr()), _stream())
        else:
            UtW = np.ascontiguousarray(UtW, dtype=np.float64).reshape(n, -1)
            c = UtW.shape[1]
            cfg = self._cfg(n, c, plink)
            # converted copies stay bound to names until the call returns (_ptr keeps only the address)
            U_c = _np64(np.ascontiguousarray(U), "U")
            ev_c = np.ascontiguousarray(eval_, dtype=np.float64)
            Uty_c = np.ascontiguousarray(Uty, dtype=np.float64)
            rc = L.lib().gemma_hip_lmm_setup(C.byref(cfg), _ptr(U_c), _ptr(ev_c), _ptr(UtW), _ptr(Uty_c))
        L.check(rc, "LMM.setup")
        self.ni_test, self.n_cvt = n, c
        self._active = True

    def setup_kept(self, UtW, Uty, plink=False):
        """lmm_setup on the kept (U, eval) of EigenDecomp_kept_K (host UtW n x c, Uty n)."""
        n = len(Uty)
        UtW = np.ascontiguousarray(UtW, dtype=np.float64).reshape(n, -1)
        Uty_c = np.ascontiguousarray(Uty, dtype=np.float64)
        cfg = self._cfg(n, UtW.shape[1], plink)
        L.check(L.lib().gemma_hip_lmm_setup_kept(C.byref(cfg), _ptr(UtW), _ptr(Uty_c)), "LMM.setup_kept")
        self.ni_test, self.n_cvt = n, UtW.shape[1]
        self._active = True

    def set_indicator(self, indicator_idv):
        ind = np.ascontiguousarray(indicator_idv, dtype=np.int32)
        L.check(L.lib().gemma_hip_lmm_set_indicator(_ptr(ind), ind.size), "LMM.set_indicator")

    def batch(self, geno, geno_kind, out=None, l=None, ld=None):
        """One block of SNPs -> SUMSTAT records (numpy in/out, or torch device tensors in/out)."""
        if _is_torch(geno):
            import torch
            l = geno.shape[0] if l is None else l
            ld = _tld(geno) if ld is None else ld
            if out is None:
                out = torch.empty((l, 8), dtype=torch.float64, device=geno.device)
            rc = L.lib().gemma_hip_lmm_batch_d(geno_kind, C.c_void_p(geno.data_ptr()), l, ld,
                                               C.c_void_p(out.data_ptr()), _stream())
            L.check(rc, "LMM.batch")
            return out
        if geno_kind == L.GENO_F64_IDV_MAJOR:
            l = geno.shape[1] if l is None else l
        else:
            l = geno.shape[0] if l is None else l
        ld = (geno.strides[0] // geno.itemsize) if ld is None else ld
        if out is None:
            out = np.zeros(l, dtype=SUMSTAT_DTYPE)
        L.check(L.lib().gemma_hip_lmm_batch(geno_kind, _ptr(geno), l, ld, _ptr(out)), "LMM.batch")
        return out

    def batch_pipe(self, geno, geno_kind, out):
        """LMM.batch with two blocks in flight (gemma_hip_lmm_batch_pipe_d; torch device tensors): the product of this block beside
        the combine and per-SNP stage of the previous one, on a CU partition.  `out` holds this block's records only after
        pipe_flush() (or any other batch call)."""
        l = geno.shape[0]
        rc = L.lib().gemma_hip_lmm_batch_pipe_d(geno_kind, C.c_void_p(geno.data_ptr()), l, _tld(geno),
                                                C.c_void_p(out.data_ptr()), _stream())
        L.check(rc, "LMM.batch_pipe")
        return out

    def pipe_flush(self):
        L.check(L.lib().gemma_hip_lmm_pipe_flush(_stream()), "LMM.pipe_flush")

    def assoc(self, UtX, out=None):
        """Per-SNP stage only, on a device-resident SNP-major UtX (torch)."""
        import torch
        l = UtX.shape[0]
        if out is None:
            out = torch.empty((l, 8), dtype=torch.float64, device=UtX.device)
        L.check(L.lib().gemma_hip_lmm_assoc_d(C.c_void_p(UtX.data_ptr()), l, _tld(UtX),
                                              C.c_void_p(out.data_ptr()), _stream()), "LMM.assoc")
        return out

    def dbg_utx(self, geno, geno_kind, path):
        """The U^T x stage alone (after setup): (l x n) array, row s = (U^T x_s)^T.  path 0: fp64 MFMA GEMM,
        path 1: exact int8-digit product where the input allows it (PLINK 2-bit, fp64 hard calls, fixed-point dosages);
        last_utx_path() says what ran."""
        geno = np.ascontiguousarray(geno)
        l, ld = geno.shape[0], geno.shape[1]
        if geno_kind == L.GENO_F64_IDV_MAJOR:
            l = geno.shape[1]
        out = np.empty((l, self.ni_test), dtype=np.float64)
        L.check(L.lib().gemma_hip_dbg_utx(geno_kind, _ptr(geno), l, ld, path, _ptr(out)), "LMM.dbg_utx")
        return out

    def finish(self):
        a, b = C.c_double(), C.c_double()
        L.check(L.lib().gemma_hip_lmm_finish(C.byref(a), C.byref(b)), "LMM.finish")
        self.time_UtX, self.time_opt = a.value, b.value
        self._active = False
        self._keep = None

    # -- drivers ----------------------------------------------------------------------
    def Analyze(self, U, eval_, UtW, Uty, geno, geno_kind=L.GENO_F64_SNP_MAJOR, plink=False,
                indicator_idv=None, batch=LMM_BATCH_SIZE):
        """LMM::Analyze (src/lmm.cpp:1474-1658): stream SNP-major `geno` in blocks of LMM_BATCH_SIZE;
        results are appended in SNP order to self.sumStat."""
        self.setup(U, eval_, UtW, Uty, plink=plink)
        try:
            if indicator_idv is not None:
                self.set_indicator(indicator_idv)
            outs = []
            for s0 in range(0, geno.shape[0], batch):
                outs.append(self.batch(geno[s0:s0 + batch], geno_kind))
            self.sumStat = np.concatenate(outs) if outs else np.zeros(0, dtype=SUMSTAT_DTYPE)
        finally:
            self.finish()
        return self.sumStat

    def AnalyzeBimbam(self, U, eval_, UtW, Uty, X_snpmajor_nan):
        """src/lmm.cpp:1660-1706 with the file reader factored out: X is SNP-major over the analysed
        individuals, NaN = "NA"."""
        return self.Analyze(U, eval_, UtW, Uty, np.ascontiguousarray(X_snpmajor_nan, dtype=np.float64),
                            L.GENO_F64_SNP_MAJOR, plink=False)

    def AnalyzeGene(self, U, eval_, UtW, Utx, Y, batch=LMM_BATCH_SIZE):
        """LMM::AnalyzeGene (src/lmm.cpp:1365-1471): Y (genes x n) holds one phenotype per row, Utx = U^T x is the
        fixed tested variable; every row gets its own null fit (l_H0, logl_H0)."""
        Y = np.ascontiguousarray(Y, dtype=np.float64)
        self.setup(U, eval_, UtW, Utx)
        try:
            outs = []
            for s0 in range(0, Y.shape[0], batch):
                blk = Y[s0:s0 + batch]
                out = np.zeros(blk.shape[0], dtype=SUMSTAT_DTYPE)
                L.check(L.lib().gemma_hip_lmm_gene_batch(_ptr(blk), blk.shape[0], blk.shape[1], _ptr(out)),
                        "LMM.AnalyzeGene")
                outs.append(out)
        finally:
            self.finish()
        self.sumStat = np.concatenate(outs) if outs else np.zeros(0, dtype=SUMSTAT_DTYPE)
        return self.sumStat

    def AnalyzeGXE(self, U, eval_, UtW, Uty, env, geno, geno_kind=L.GENO_F64_SNP_MAJOR, indicator_idv=None,
                   batch=LMM_BATCH_SIZE):
        """LMM::AnalyzeBimbamGXE / AnalyzePlinkGXE (src/lmm.cpp:2283-2608): covariates [W, env, x_s], tested variable
        x_s . env; env is over the analysed individuals."""
        plink = geno_kind == L.GENO_PLINK_2BIT
        self.setup(U, eval_, UtW, Uty, plink=plink)
        try:
            if indicator_idv is not None:
                self.set_indicator(indicator_idv)
            env = np.ascontiguousarray(env, dtype=np.float64)
            L.check(L.lib().gemma_hip_lmm_set_env(_ptr(env)), "LMM.set_env")
            geno = np.ascontiguousarray(geno)
            outs = []
            for s0 in range(0, geno.shape[0], batch):
                blk = geno[s0:s0 + batch]
                out = np.zeros(blk.shape[0], dtype=SUMSTAT_DTYPE)
                L.check(L.lib().gemma_hip_lmm_gxe_batch(geno_kind, _ptr(blk), blk.shape[0], blk.shape[1], _ptr(out)),
                        "LMM.AnalyzeGXE")
                outs.append(out)
        finally:
            self.finish()
        self.sumStat = np.concatenate(outs) if outs else np.zeros(0, dtype=SUMSTAT_DTYPE)
        return self.sumStat

    def AnalyzePlink(self, U, eval_, UtW, Uty, bed_rows, indicator_idv):
        """src/lmm.cpp:1710-1903: bed_rows = the .bed payload of the analysed SNPs (uint8, one row of
        ceil(ni_total/4) bytes per SNP); non-analysed individuals are dropped on device."""
        return self.Analyze(U, eval_, UtW, Uty, np.ascontiguousarray(bed_rows, dtype=np.uint8),
                            L.GENO_PLINK_2BIT, plink=True, indicator_idv=indicator_idv)

    def WriteFiles(self, path, snp_info):
        """LMM::WriteFiles (src/lmm.cpp:101-225): `.assoc.txt`.  snp_info: iterable of dicts with
        chr, rs, ps, n_miss, allele1, allele0, af for every analysed SNP, in order."""
        m = self.a_mode
        head = "chr\trs\tps\tn_miss\tallele1\tallele0\taf\t"
        cols = {1: ["beta", "se", "logl_H1", "l_remle", "p_wald"],
                2: ["logl_H1", "l_mle", "p_lrt"],
                3: ["beta", "se", "p_score"],
                4: ["beta", "se", "logl_H1", "l_remle", "l_mle", "p_wald", "p_lrt", "p_score"],
                9: ["beta", "se", "l_mle", "p_lrt"]}[m]
        field = {"l_remle": "lambda_remle", "l_mle": "lambda_mle"}
        with open(path, "w") as f:
            f.write(head + "\t".join(cols) + "\n")
            for info, st in zip(snp_info, self.sumStat):
                f.write("%s\t%s\t%s\t%d\t%s\t%s\t%.3f" % (info["chr"], info["rs"], info["ps"], info["n_miss"],
                                                          info["allele1"], info["allele0"], info["af"]))
                for cname in cols:
                    f.write("\t%.6e" % st[field.get(cname, cname)])
                f.write("\n")


class LM:
    """Mirror of class LM (src/lm.h) for `-lm 1..4` (a_mode 51..54): ordinary per-SNP regression, no kinship."""

    def __init__(self, a_mode=51):
        self.a_mode = a_mode
        self.sumStat = np.zeros(0, dtype=SUMSTAT_DTYPE)

    def Analyze(self, W, y, geno, geno_kind=L.GENO_F64_SNP_MAJOR, indicator_idv=None, batch=LMM_BATCH_SIZE):
        """LM::AnalyzeBimbam / AnalyzePlink (src/lm.cpp:382-640) over SNP-major `geno`."""
        y = np.ascontiguousarray(y, dtype=np.float64)
        W = np.ascontiguousarray(W, dtype=np.float64).reshape(len(y), -1)
        L.check(L.lib().gemma_hip_lm_setup(self.a_mode, W.shape[0], W.shape[1], _ptr(W), _ptr(y)), "LM.setup")
        geno = np.ascontiguousarray(geno)  # SNP-major rows (a Fortran-ordered array would hand over a stride of 1)
        try:
            if indicator_idv is not None:
                ind = np.ascontiguousarray(indicator_idv, dtype=np.int32)
                L.check(L.lib().gemma_hip_lmm_set_indicator(_ptr(ind), ind.size), "LM.set_indicator")
            outs = []
            for s0 in range(0, geno.shape[0], batch):
                blk = geno[s0:s0 + batch]
                out = np.zeros(blk.shape[0], dtype=SUMSTAT_DTYPE)
                L.check(L.lib().gemma_hip_lm_batch(geno_kind, _ptr(blk), blk.shape[0], blk.strides[0] // blk.itemsize,
                                                   _ptr(out)), "LM.batch")
                outs.append(out)
            self.sumStat = np.concatenate(outs) if outs else np.zeros(0, dtype=SUMSTAT_DTYPE)
        finally:
            L.check(L.lib().gemma_hip_lm_finish(), "LM.finish")
        return self.sumStat


class MVLMM:
    """Mirror of class MVLMM (src/mvlmm.h:32-104): the fields CopyFromParam fills (src/mvlmm.cpp:51-90) and
    AnalyzeBimbam / AnalyzePlink (:2972-3899); crt = 1 is the reference's -crt (CalcCRT / PCRT).  sumStat is a dict of arrays with MPHSUMSTAT's fields
    (src/param.h:68-77): beta (l x d), Vbeta / Vg / Ve (l x d(d+1)/2, upper triangles row by row), p_wald, p_lrt, p_score."""

    def __init__(self, a_mode=1, l_min=1e-5, l_max=1e5, n_region=10, em_iter=10000, nr_iter=100, em_prec=1e-4,
                 nr_prec=1e-4, p_nr=1e-3, crt=0):
        self.a_mode = a_mode
        self.crt = crt
        self.l_min, self.l_max, self.n_region = l_min, l_max, n_region
        self.em_iter, self.nr_iter, self.em_prec, self.nr_prec, self.p_nr = em_iter, nr_iter, em_prec, nr_prec, p_nr
        self.sumStat = {}
        self.null = None

    def _opt(self, gxe=0):
        return L.MvOpt(self.em_iter, self.nr_iter, self.em_prec, self.nr_prec, self.p_nr, self.crt, gxe)

    def fit_null(self, eval_, UtW, UtY):
        """The null block (src/mvlmm.cpp:3056-3208) -> dict with Vg/Ve/B/logl for 'remle' and 'mle'; also fills the
        members WriteFiles' callers read (Vg_remle_null, ..., logl_mle_H0)."""
        UtY = np.ascontiguousarray(UtY, dtype=np.float64)
        n, d = UtY.shape
        UtW = np.ascontiguousarray(UtW, dtype=np.float64).reshape(n, -1)
        c = UtW.shape[1]
        ev = np.ascontiguousarray(eval_, dtype=np.float64)
        nf, opt = L.MvNull(), self._opt()
        L.check(L.lib().gemma_hip_mvlmm_null(n, c, d, _ptr(ev), _ptr(UtW), _ptr(UtY), self.l_min, self.l_max,
                                             self.n_region, C.byref(opt), C.byref(nf)), "MVLMM.fit_null")
        self._null_struct = nf
        g = lambda a, r, k: np.array(a[:r * k]).reshape(r, k)
        self.null = {"Vg_remle": g(nf.Vg_remle, d, d), "Ve_remle": g(nf.Ve_remle, d, d), "B_remle": g(nf.B_remle, d, c),
                     "logl_remle": nf.logl_remle_H0, "Vg_mle": g(nf.Vg_mle, d, d), "Ve_mle": g(nf.Ve_mle, d, d),
                     "B_mle": g(nf.B_mle, d, c), "logl_mle": nf.logl_mle_H0}
        self.logl_remle_H0, self.logl_mle_H0 = nf.logl_remle_H0, nf.logl_mle_H0
        return self.null

    def Analyze(self, U, eval_, UtW, UtY, geno, geno_kind, indicator_idv=None, batch=LMM_BATCH_SIZE, env=None):
        """env (over the analysed individuals): the interaction test of AnalyzeBimbamGXE / AnalyzePlinkGXE (src/mvlmm.cpp:3970-4870):
        the null model is fitted on (W, env) (:4046-4070), per SNP x o env is tested with (W, env, x) as covariates."""
        UtY = np.ascontiguousarray(UtY, dtype=np.float64)
        n, d = UtY.shape
        UtW = np.ascontiguousarray(UtW, dtype=np.float64).reshape(n, -1)
        if env is not None:
            env = np.ascontiguousarray(env, dtype=np.float64)
            Ute = CalcUtX(np.ascontiguousarray(U, dtype=np.float64), env)  # gsl_blas_dgemv(CblasTrans, U, env), :4047 -- on the library's GEMM
            self.fit_null(eval_, np.column_stack([UtW, Ute]), UtY)
        else:
            self.fit_null(eval_, UtW, UtY)
        lmm = LMM(a_mode=self.a_mode, l_min=self.l_min, l_max=self.l_max, n_region=self.n_region)
        lmm.setup(U, eval_, UtW, np.ascontiguousarray(UtY[:, 0]), plink=(geno_kind == L.GENO_PLINK_2BIT))
        v = d * (d + 1) // 2
        stride = d + 3 * v + 3
        outs = []
        try:
            if indicator_idv is not None:
                lmm.set_indicator(indicator_idv)
            if env is not None:
                L.check(L.lib().gemma_hip_lmm_set_env(_ptr(env)), "MVLMM.set_env")
            opt = self._opt(gxe=1 if env is not None else 0)
            L.check(L.lib().gemma_hip_mvlmm_set(d, _ptr(UtY), C.byref(self._null_struct), C.byref(opt)), "MVLMM.set")
            geno = np.ascontiguousarray(geno)
            for s0 in range(0, geno.shape[0], batch):
                blk = geno[s0:s0 + batch]
                out = np.zeros((blk.shape[0], stride))
                L.check(L.lib().gemma_hip_mvlmm_batch(geno_kind, _ptr(blk), blk.shape[0], blk.shape[1], _ptr(out)),
                        "MVLMM.Analyze")
                outs.append(out)
        finally:
            lmm.finish()
        o = np.concatenate(outs) if outs else np.zeros((0, stride))
        self.sumStat = {"beta": o[:, :d], "Vbeta": o[:, d:d + v], "Vg": o[:, d + v:d + 2 * v],
                        "Ve": o[:, d + 2 * v:d + 3 * v], "p_wald": o[:, d + 3 * v], "p_lrt": o[:, d + 3 * v + 1],
                        "p_score": o[:, d + 3 * v + 2]}
        return self.sumStat

    def AnalyzeBimbam(self, U, eval_, UtW, UtY, X_snpmajor, batch=LMM_BATCH_SIZE):
        """src/mvlmm.cpp:2972-3416: X_snpmajor holds one row per analysed SNP over the analysed individuals, NaN = NA."""
        return self.Analyze(U, eval_, UtW, UtY, np.ascontiguousarray(X_snpmajor, dtype=np.float64), L.GENO_F64_SNP_MAJOR,
                            batch=batch)

    def AnalyzePlink(self, U, eval_, UtW, UtY, bed_rows, indicator_idv, batch=LMM_BATCH_SIZE):
        """src/mvlmm.cpp:3418-3899"""
        return self.Analyze(U, eval_, UtW, UtY, np.ascontiguousarray(bed_rows, dtype=np.uint8), L.GENO_PLINK_2BIT,
                            indicator_idv=indicator_idv, batch=batch)

    def AnalyzeBimbamGXE(self, U, eval_, UtW, UtY, env, X_snpmajor, batch=LMM_BATCH_SIZE):
        """src/mvlmm.cpp:3970-4414"""
        return self.Analyze(U, eval_, UtW, UtY, np.ascontiguousarray(X_snpmajor, dtype=np.float64), L.GENO_F64_SNP_MAJOR,
                            batch=batch, env=env)

    def AnalyzePlinkGXE(self, U, eval_, UtW, UtY, env, bed_rows, indicator_idv, batch=LMM_BATCH_SIZE):
        """src/mvlmm.cpp:4416-4870"""
        return self.Analyze(U, eval_, UtW, UtY, np.ascontiguousarray(bed_rows, dtype=np.uint8), L.GENO_PLINK_2BIT,
                            indicator_idv=indicator_idv, batch=batch, env=env)

    def WriteFiles(self, path, snp_info):
        """MVLMM::WriteFiles (src/mvlmm.cpp:117-210)"""
        d = self.sumStat["beta"].shape[1]
        head = ["chr", "rs", "ps", "n_miss", "allele1", "allele0", "af"] + ["beta_%d" % (i + 1) for i in range(d)]
        head += ["Vbeta_%d_%d" % (i + 1, j + 1) for i in range(d) for j in range(i, d)]
        pcols = {1: ["p_wald"], 2: ["p_lrt"], 3: ["p_score"], 4: ["p_wald", "p_lrt", "p_score"]}[self.a_mode]
        with open(path, "w") as f:
            f.write("\t".join(head + pcols) + "\n")
            for t, si in enumerate(snp_info):
                row = [str(si["chr"]), str(si["rs"]), str(si["ps"]), str(si["n_miss"]), str(si["allele1"]),
                       str(si["allele0"]), "%.3f" % si["af"]]
                row += ["%.6e" % x for x in self.sumStat["beta"][t]] + ["%.6e" % x for x in self.sumStat["Vbeta"][t]]
                row += ["%.6e" % self.sumStat[k][t] for k in pcols]
                f.write("\t".join(row) + "\n")
