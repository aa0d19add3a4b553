// Entry points of the eigensolver's translation unit (eigh_tu.hip = eigh.hip.h + eigh2.hip.h + its own instance of the
// fp64 MFMA GEMM).  The solver is a separate object file so that a change to one of its kernels does not rebuild the
// association path and vice versa; nothing here is exported from the shared library (C++ linkage, hidden by the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <string>

namespace gemma_hip {

// Several ranks, one decomposition (eigh.hip.h, "Several ranks"): the two collectives of the library's communicator, handed
// in by the unit that owns it.  nullptr / world == 1: the whole job on this rank.
struct EighShard {
  int rank = 0, world = 1;
  void *ctx = nullptr;
  int (*bcast)(void *ctx, void *buf_d, size_t bytes, int root, hipStream_t s) = nullptr; // in place, device buffer
  int (*allreduce_sum)(void *ctx, double *buf_d, size_t count, hipStream_t s) = nullptr;
};

// G (n x n symmetric, device, destroyed) -> U (row-major, eigenvector k in column k), eval ascending.  0 on success,
// otherwise a GEMMA_HIP_E* code with the reason in msg.  LAPACK equivalent: dsyevr_ (GEMMA src/lapack.cpp:149-236).
// With sh (world > 1) the call is a COLLECTIVE: every rank passes the same matrix and receives the same (U, eval).
int eigh_device_x(double *G, long n, double *U, double *eval, hipStream_t s, std::string &msg, const EighShard *sh = nullptr);

// A rank that fails BEFORE eigh_device_x in a collective solve of order n (its own allocations) calls this instead: it takes part in
// the solver's first agreement with "failed", so that the other ranks return an error instead of waiting (no-op where a solve of
// that order exchanges nothing).
void eigh_abort_x(long n, hipStream_t s, const EighShard *sh);

// stage diagnostics behind gemma_hip_dbg_tridiag / _dbg_eigh2 / _dbg_stedc (host pointers; tests/test_gpu_eigh.py)
int dbg_tridiag_x(const double *G, size_t n, double *d, double *e, double *tau, double *VT, std::string &msg);
int dbg_eigh2_x(const double *G, size_t n, double *band, double *d, double *e, std::string &msg);
int dbg_stedc_x(const double *d, const double *e, size_t n, double *w, double *ZT, std::string &msg);

// stage seconds of the last solve that ran with GEMMA_HIP_EIGH_TIMING=1: {reduction, bulge chase, divide & conquer, Q2, Q1 /
// one-stage back-transformation, sort + transpose, n, stages}
void eigh_last_stages(double *t8);

// The solver's workspace pool (eigh.hip.h, EigPool): reserve = allocate every buffer a solve of order n takes and leave them idle in
// the pool (which from then on keeps the buffers of every solve); release = hand all idle buffers back; bytes = what sits idle.
int eigh_reserve_x(long n, std::string &msg);
size_t eigh_release_x();
size_t eigh_pool_idle_bytes_x();

void eigh_tu_shutdown(); // side stream / events of this unit's GEMM launcher; the workspace pool

} // namespace gemma_hip
