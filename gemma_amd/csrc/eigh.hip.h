// Symmetric eigensolver (placeholder until the tridiagonalisation + divide-and-conquer kernels land).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
namespace gemma_hip {
static inline int eigh_device(double *, long, double *, double *, hipStream_t, std::string &msg) {
  msg = "eigensolver not built into this library yet";
  return 4; // GEMMA_HIP_ERUNTIME
}
} // namespace gemma_hip
