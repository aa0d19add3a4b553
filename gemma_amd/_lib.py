"""ctypes binding of include/gemma_hip.h.  There is no CPU fallback: if the shared library is
missing this module raises, and every entry point returns GEMMA_HIP_ENODEV without a gfx950 GPU."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libgemma_hip.so")

# every symbol include/gemma_hip.h declares
SYMBOLS = [
    "gemma_hip_init", "gemma_hip_shutdown", "gemma_hip_abi_version", "gemma_hip_strerror",
    "gemma_hip_last_error", "gemma_hip_device_info", "gemma_hip_dgemm", "gemma_hip_dgemm_d",
    "gemma_hip_kin_begin", "gemma_hip_kin_add", "gemma_hip_kin_add_d", "gemma_hip_kin_end",
    "gemma_hip_kin_end_d", "gemma_hip_kin_loco_d", "gemma_hip_snp_qc", "gemma_hip_center", "gemma_hip_center_d", "gemma_hip_eigh",
    "gemma_hip_eigh_d", "gemma_hip_eigh_reserve", "gemma_hip_eigh_release", "gemma_hip_dbg_eigh_last", "gemma_hip_eigh_sharded_d", "gemma_hip_eigh_kept_K_sharded", "gemma_hip_calc_utx", "gemma_hip_lmm_setup", "gemma_hip_lmm_setup_d",
    "gemma_hip_lmm_null", "gemma_hip_lmm_set_indicator", "gemma_hip_lmm_batch", "gemma_hip_lmm_batch_d",
    "gemma_hip_lmm_assoc_d", "gemma_hip_lmm_finish", "gemma_hip_lm_setup", "gemma_hip_lm_batch", "gemma_hip_lm_batch_d",
    "gemma_hip_lm_finish", "gemma_hip_profile_enable",
    "gemma_hip_profile_read", "gemma_hip_dbg_tridiag", "gemma_hip_dbg_stedc", "gemma_hip_dbg_eigh2", "gemma_hip_dbg_utx", "gemma_hip_lmm_gene_batch", "gemma_hip_lmm_gene_batch_d", "gemma_hip_lmm_set_env",
    "gemma_hip_lmm_gxe_batch", "gemma_hip_lmm_gxe_batch_d",
    "gemma_hip_mvlmm_null", "gemma_hip_mvlmm_set", "gemma_hip_mvlmm_batch", "gemma_hip_mvlmm_batch_d",
    "gemma_hip_kin_end_keep", "gemma_hip_kept_K_get", "gemma_hip_eigh_kept_K", "gemma_hip_eigh_keep", "gemma_hip_kept_n",
    "gemma_hip_kept_bcast", "gemma_hip_kept_U_get", "gemma_hip_calc_utx_kept", "gemma_hip_lmm_setup_kept", "gemma_hip_kept_release",
    "gemma_hip_comm_unique_id", "gemma_hip_comm_init", "gemma_hip_comm_info", "gemma_hip_comm_bcast_d",
    "gemma_hip_comm_allreduce_sum_d", "gemma_hip_comm_finalize", "gemma_hip_comm_selftest", "gemma_hip_comm_stats", "gemma_hip_dbg_i8_digits", "gemma_hip_dbg_last_utx_path", "gemma_hip_lmm_batch_submit", "gemma_hip_lmm_batch_collect",
    "gemma_hip_dbg_last_utx_kernel", "gemma_hip_dbg_last_block_missing", "gemma_hip_reload_env", "gemma_hip_lmm_batch_pipe_d", "gemma_hip_lmm_pipe_flush",
]
COMM_ID_BYTES = 128

OK, EINVAL, ENODEV, ENOMEM, ERUNTIME, ESTATE, ENOCONV = range(7)
GENO_F64_SNP_MAJOR, GENO_PLINK_2BIT, GENO_F64_IDV_MAJOR = 0, 1, 2
STAGE_INGEST, STAGE_UTX_GEMM, STAGE_ASSOC, STAGE_KIN_GEMM, STAGE_EIGH, STAGE_UTX_POST = range(6)


class SumStat(C.Structure):
    _fields_ = [(k, C.c_double) for k in
                ("beta", "se", "lambda_remle", "lambda_mle", "p_wald", "p_lrt", "p_score", "logl_H1")]


class LmmCfg(C.Structure):
    _fields_ = [("a_mode", C.c_int), ("n", C.c_size_t), ("n_cvt", C.c_size_t), ("l_min", C.c_double),
                ("l_max", C.c_double), ("n_region", C.c_size_t), ("l_mle_null", C.c_double),
                ("logl_mle_H0", C.c_double), ("plink_nan_rule", C.c_int)]


class UtxKernelInfo(C.Structure):
    """gemma_utx_kernel_info: the matrix kernel the last U^T x launched, as the library's launch site recorded it"""
    _fields_ = [("variant", C.c_int), ("rows", C.c_int), ("digits", C.c_int), ("fuse", C.c_int), ("raster", C.c_int),
                ("launches", C.c_long), ("name", C.c_char * 64)]


UTX_KERNEL_DGEMM_F64, UTX_KERNEL_DENSE_I8, UTX_KERNEL_SPARSE_BYTES, UTX_KERNEL_RECORDS_R32, UTX_KERNEL_RECORDS_R16, \
    UTX_KERNEL_DOSAGE_I8, UTX_KERNEL_DOSAGE_I8_R16 = range(7)


class CommStats(C.Structure):
    """gemma_comm_stats: calls / pieces / bytes (and, with GEMMA_HIP_COMM_TIMING=1, seconds) of the library's collectives"""
    _fields_ = [("allreduce_calls", C.c_long), ("allreduce_pieces", C.c_long), ("bcast_calls", C.c_long), ("bcast_pieces", C.c_long),
                ("allreduce_bytes", C.c_double), ("bcast_bytes", C.c_double), ("allreduce_s", C.c_double), ("bcast_s", C.c_double)]


class QcCfg(C.Structure):
    _fields_ = [("maf_level", C.c_double), ("miss_level", C.c_double), ("hwe_level", C.c_double),
                ("r2_level", C.c_double)]


class MvNull(C.Structure):
    """gemma_mvlmm_null: V_g, V_e (d x d, leading dimension d), B (d x n_cvt), logl for the REMLE and the MLE null fit"""
    _fields_ = [("Vg_remle", C.c_double * 64), ("Ve_remle", C.c_double * 64), ("B_remle", C.c_double * 96),  # GEMMA_MV_DMAX 8, _CMAX 12
                ("logl_remle_H0", C.c_double), ("Vg_mle", C.c_double * 64), ("Ve_mle", C.c_double * 64),
                ("B_mle", C.c_double * 96), ("logl_mle_H0", C.c_double)]


class MvOpt(C.Structure):
    _fields_ = [("em_iter", C.c_size_t), ("nr_iter", C.c_size_t), ("em_prec", C.c_double), ("nr_prec", C.c_double),
                ("p_nr", C.c_double), ("crt", C.c_size_t), ("gxe", C.c_size_t)]


class GemmaHipError(RuntimeError):
    def __init__(self, code, where, detail):
        self.code = code
        super().__init__("%s: error %d (%s)" % (where, code, detail))


_lib = None


def lib():
    """Load gemma_amd/libgemma_hip.so (built by gemma_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("gemma_amd/libgemma_hip.so is missing -- run `python -m gemma_amd.build` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    # One HIP runtime per process: when torch is present (it only supplies device memory, streams and
    # torch.distributed here) it must be imported BEFORE this library so that both resolve to the
    # libamdhip64 torch ships; loading /opt/rocm's runtime first leaves torch without a device.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, dp, sz, ci, cd = C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_double
    L.gemma_hip_init.argtypes = [ci, ci]
    L.gemma_hip_shutdown.restype = None
    L.gemma_hip_strerror.restype = C.c_char_p
    L.gemma_hip_strerror.argtypes = [ci]
    L.gemma_hip_last_error.restype = C.c_char_p
    L.gemma_hip_device_info.argtypes = [C.c_char_p, sz, C.POINTER(ci), C.POINTER(sz)]
    gemm = [C.c_char, C.c_char, sz, sz, sz, cd, dp, sz, dp, sz, cd, dp, sz]
    L.gemma_hip_dgemm.argtypes = gemm
    L.gemma_hip_dgemm_d.argtypes = gemm + [vp]
    L.gemma_hip_kin_begin.argtypes = [sz, ci]
    L.gemma_hip_kin_add.argtypes = [ci, vp, sz, sz]
    L.gemma_hip_kin_add_d.argtypes = [ci, vp, sz, sz, vp]
    L.gemma_hip_kin_end.argtypes = [dp, C.POINTER(sz)]
    L.gemma_hip_kin_end_d.argtypes = [dp, C.POINTER(sz), vp]
    L.gemma_hip_kin_loco_d.argtypes = [dp, sz, dp, sz, sz, vp]
    L.gemma_hip_snp_qc.argtypes = [ci, vp, sz, sz, vp, sz, dp, sz, sz, C.POINTER(QcCfg), vp, dp, vp]
    L.gemma_hip_center.argtypes = [dp, sz]
    L.gemma_hip_center_d.argtypes = [dp, sz, vp]
    L.gemma_hip_eigh.argtypes = [dp, sz, dp, dp, C.POINTER(cd)]
    L.gemma_hip_eigh_d.argtypes = [dp, sz, dp, dp, C.POINTER(cd), vp]
    L.gemma_hip_eigh_sharded_d.argtypes = [dp, sz, dp, dp, C.POINTER(cd), vp]
    L.gemma_hip_dbg_eigh_last.argtypes = [dp]
    L.gemma_hip_eigh_reserve.argtypes = [sz]
    L.gemma_hip_eigh_release.argtypes = [C.POINTER(sz)]
    L.gemma_hip_calc_utx.argtypes = [dp, dp, sz, sz, dp]
    L.gemma_hip_lmm_setup.argtypes = [C.POINTER(LmmCfg), dp, dp, dp, dp]
    L.gemma_hip_lmm_setup_d.argtypes = [C.POINTER(LmmCfg), dp, dp, dp, dp, vp]
    L.gemma_hip_lmm_null.argtypes = [sz, sz, dp, dp, dp, cd, cd, sz, cd, dp]
    L.gemma_hip_lmm_set_indicator.argtypes = [vp, sz]
    L.gemma_hip_lmm_batch.argtypes = [ci, vp, sz, sz, vp]
    L.gemma_hip_lmm_batch_d.argtypes = [ci, vp, sz, sz, vp, vp]
    L.gemma_hip_lmm_assoc_d.argtypes = [dp, sz, sz, vp, vp]
    L.gemma_hip_lmm_gene_batch.argtypes = [dp, sz, sz, vp]
    L.gemma_hip_lmm_set_env.argtypes = [dp]
    L.gemma_hip_lmm_gxe_batch.argtypes = [ci, vp, sz, sz, vp]
    L.gemma_hip_lmm_gxe_batch_d.argtypes = [ci, vp, sz, sz, vp, vp]
    L.gemma_hip_lmm_gene_batch_d.argtypes = [dp, sz, sz, vp, vp]
    L.gemma_hip_mvlmm_null.argtypes = [sz, sz, sz, dp, dp, dp, cd, cd, sz, C.POINTER(MvOpt), C.POINTER(MvNull)]
    L.gemma_hip_mvlmm_set.argtypes = [sz, dp, C.POINTER(MvNull), C.POINTER(MvOpt)]
    L.gemma_hip_mvlmm_batch.argtypes = [ci, vp, sz, sz, dp]
    L.gemma_hip_mvlmm_batch_d.argtypes = [ci, vp, sz, sz, dp, vp]
    L.gemma_hip_lmm_finish.argtypes = [C.POINTER(cd), C.POINTER(cd)]
    L.gemma_hip_lm_setup.argtypes = [ci, sz, sz, dp, dp]
    L.gemma_hip_lm_batch.argtypes = [ci, vp, sz, sz, vp]
    L.gemma_hip_lm_batch_d.argtypes = [ci, vp, sz, sz, vp, vp]
    L.gemma_hip_profile_enable.argtypes = [ci]
    L.gemma_hip_profile_read.argtypes = [ci, C.POINTER(cd), C.POINTER(C.c_long), ci]
    L.gemma_hip_dbg_tridiag.argtypes = [dp, sz, dp, dp, dp, dp]
    L.gemma_hip_dbg_stedc.argtypes = [dp, dp, sz, dp, dp]
    L.gemma_hip_dbg_eigh2.argtypes = [dp, sz, dp, dp, dp]
    L.gemma_hip_dbg_utx.argtypes = [C.c_int, vp, sz, sz, C.c_int, dp]
    L.gemma_hip_kin_end_keep.argtypes = [C.POINTER(sz), ci]
    L.gemma_hip_kept_K_get.argtypes = [dp]
    L.gemma_hip_eigh_kept_K.argtypes = [vp, sz, dp, C.POINTER(cd)]
    L.gemma_hip_eigh_kept_K_sharded.argtypes = [vp, sz, dp, C.POINTER(cd)]
    L.gemma_hip_eigh_keep.argtypes = [dp, sz, dp, C.POINTER(cd)]
    L.gemma_hip_kept_n.argtypes = [C.POINTER(sz)]
    L.gemma_hip_kept_bcast.argtypes = [ci, C.POINTER(cd)]
    L.gemma_hip_kept_U_get.argtypes = [dp, dp]
    L.gemma_hip_calc_utx_kept.argtypes = [dp, sz, sz, dp]
    L.gemma_hip_lmm_setup_kept.argtypes = [C.POINTER(LmmCfg), dp, dp]
    L.gemma_hip_dbg_i8_digits.argtypes = [sz, C.POINTER(ci)]
    L.gemma_hip_dbg_last_utx_path.argtypes = [C.POINTER(ci)]
    L.gemma_hip_dbg_last_utx_kernel.argtypes = [C.POINTER(UtxKernelInfo)]
    L.gemma_hip_dbg_last_block_missing.argtypes = [C.POINTER(ci)]
    L.gemma_hip_reload_env.argtypes = []
    L.gemma_hip_lmm_batch_pipe_d.argtypes = [ci, vp, sz, sz, vp, vp]
    L.gemma_hip_lmm_pipe_flush.argtypes = [vp]
    L.gemma_hip_lmm_batch_submit.argtypes = [ci, vp, sz, sz]
    L.gemma_hip_lmm_batch_collect.argtypes = [vp, C.POINTER(sz)]
    L.gemma_hip_comm_unique_id.argtypes = [vp]
    L.gemma_hip_comm_init.argtypes = [vp, ci, ci]
    L.gemma_hip_comm_info.argtypes = [C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)]
    L.gemma_hip_comm_bcast_d.argtypes = [vp, sz, ci, vp]
    L.gemma_hip_comm_allreduce_sum_d.argtypes = [dp, sz, vp]
    L.gemma_hip_comm_selftest.argtypes = [vp]
    L.gemma_hip_comm_stats.argtypes = [C.POINTER(CommStats)]
    for s in SYMBOLS:
        getattr(L, s)  # AttributeError if the library does not export what the header declares
    _lib = L
    return L


def check(rc, where):
    if rc != OK:
        L = lib()
        detail = L.gemma_hip_last_error().decode() or L.gemma_hip_strerror(rc).decode()
        raise GemmaHipError(rc, where, detail)
