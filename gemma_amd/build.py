"""Build recipe for the gfx950 shared library (explicit hipcc, in-tree output)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgemma_hip.so")
# one object per source, compiled concurrently; the headers each one depends on (staleness by mtime)
UNITS = {
    "gemma_hip.hip": ["dgemm_mfma.hip.h", "lmm_grid.hip.h", "i8gemm.hip.h", "i8gemm_sparse.hip.h", "i8gemm_sparse2.hip.h", "i8gemm_sparse2_r16.hip.h", "i8gemm_dense16.hip.h", "lmm_assoc.hip.h",
                      "lmm_search.hip.h", "comm.hip.h", "comm_shm.hpp", "kin_i8.hip.h", "ingest.hip.h", "qc.hip.h", "lm_assoc.hip.h", "mvlmm.hip.h",
                      "mvlmm_kernels.hip.h", "eigh_tu.h",
                      # the stages of the C ABI: textual parts of gemma_hip.hip (one translation unit around g_ctx)
                      "abi_kinship.inc.h", "abi_eigen_qc.inc.h", "abi_lmm_stage.inc.h", "abi_utx.inc.h", "abi_lmm_batch.inc.h",
                      "abi_mvlmm.inc.h", "abi_gxe_lm.inc.h", "abi_kept_comm.inc.h"],
    "eigh_tu.hip": ["dgemm_mfma.hip.h", "eigh.hip.h", "eigh2.hip.h", "eigh_tu.h"],  # the eigensolver: its own object file
    "mvlmm_kernels.hip": ["mvlmm.hip.h", "mvlmm_kernels.hip.h"],
    "mvlmm_kernels_wide.hip": ["mvlmm.hip.h", "mvlmm_kernels.hip.h"],
    "mvlmm_kernels_rt.hip": ["mvlmm.hip.h", "mvlmm_kernels.hip.h"],  # the run-time (d, c) instance
    "mvlmm_kernels_d6.hip": ["mvlmm.hip.h", "mvlmm_kernels.hip.h"],  # six phenotypes, fixed form (two wavefronts per workgroup)
    "mvlmm_kernels_d7.hip": ["mvlmm.hip.h", "mvlmm_kernels.hip.h"],  # seven phenotypes (one wavefront per workgroup)
    "mvlmm_kernels_d8.hip": ["mvlmm.hip.h", "mvlmm_kernels.hip.h"],  # eight phenotypes, one covariate
}
PUBLIC_HEADER_USERS = ("gemma_hip.hip", "eigh_tu.hip")
SOURCES = list(UNITS)
HEADERS = sorted({h for hs in UNITS.values() for h in hs})


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the gfx950 library cannot be built")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    deps.append(os.path.join(HERE, "..", "include", "gemma_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> gemma_amd/libgemma_hip.so (cross-compiles without a GPU)."""
    if not force and not stale():
        return LIB
    cc = hipcc()
    objdir = os.path.join(HERE, "_obj")
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        hdrs = UNITS[src]
        deps = [os.path.join(CSRC, f) for f in [src] + hdrs]
        if src in PUBLIC_HEADER_USERS:  # the multivariate kernels' units do not include the public header: a comment edit there
            deps.append(os.path.join(HERE, "..", "include", "gemma_hip.h"))  # must not cost them a five-minute rebuild
        if not force and os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps):
            continue
        cmd = [cc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [cc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
