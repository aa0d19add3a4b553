// Entry points of the eigensolver's translation unit (eigh_tu.hip = eigh.hip.h + eigh2.hip.h + its own instance of the
// fp64 MFMA GEMM).  The solver is a separate object file so that a change to one of its kernels does not rebuild the
// association path and vice versa; nothing here is exported from the shared library (C++ linkage, hidden by the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <string>

namespace gemma_hip {

// G (n x n symmetric, device, destroyed) -> U (row-major, eigenvector k in column k), eval ascending.  0 on success,
// otherwise a GEMMA_HIP_E* code with the reason in msg.  LAPACK equivalent: dsyevr_ (GEMMA src/lapack.cpp:149-236).
int eigh_device_x(double *G, long n, double *U, double *eval, hipStream_t s, std::string &msg);

// stage diagnostics behind gemma_hip_dbg_tridiag / _dbg_eigh2 / _dbg_stedc (host pointers; tests/test_gpu_eigh.py)
int dbg_tridiag_x(const double *G, size_t n, double *d, double *e, double *tau, double *VT, std::string &msg);
int dbg_eigh2_x(const double *G, size_t n, double *band, double *d, double *e, std::string &msg);
int dbg_stedc_x(const double *d, const double *e, size_t n, double *w, double *ZT, std::string &msg);

void eigh_tu_shutdown(); // side stream / events of this unit's GEMM launcher

} // namespace gemma_hip
