// Multi-GPU transport of the library: one process per GPU, RCCL over xGMI.
//
// The path needs two collectives and no more: ONE broadcast of (U, eval) from the rank that ran the eigensolver, and --
// when the kinship is SNP-sharded -- ONE all-reduce of the n^2 partial sums.  They are issued directly on RCCL
// (ncclBroadcast / ncclAllReduce, fp64, on the caller's HIP stream); no torch, no MPI.  librccl is bound lazily with
// dlopen the first time a communicator of more than one rank is created, so single-GPU runs never load it (and a
// process that already holds torch's copy of librccl.so.1 shares that one: same SONAME).
//
// Bootstrap: rank 0 asks for an id (gemma_hip_comm_unique_id -> ncclGetUniqueId, 128 bytes), the host program ships it
// to the other ranks however it likes (the C++ driver: a pipe from before the fork; bench.py: torch.distributed's store),
// every rank calls gemma_hip_comm_init(id, rank, world) with its own device current (ncclCommInitRank).
//
// Test transport: RCCL refuses two ranks on one device, and the 1-GPU test box has one.  With GEMMA_HIP_COMM=shm in
// the environment the same entry points run over a POSIX shared-memory segment named by the id (device -> host ->
// segment -> host -> device, process-shared barrier): it exists so that `-gpus 2 -samegpu` and the 2-rank tests
// exercise the REAL protocol and the REAL kernels on one device; it is never chosen implicitly.
#pragma once
#include <dlfcn.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "comm_shm.hpp"

namespace gemma_hip {

struct RcclApi {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool load(std::string &err) {
    if (handle) return true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
      handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
      if (handle) break;
    }
    if (!handle) {
      err = std::string("cannot load librccl: ") + dlerror();
      return false;
    }
#define GH_SYM(field, name)                                                      \
  field = reinterpret_cast<decltype(field)>(dlsym(handle, name));                \
  if (!field) {                                                                  \
    err = std::string("librccl lacks ") + name;                                  \
    return false;                                                                \
  }
    GH_SYM(GetUniqueId, "ncclGetUniqueId")
    GH_SYM(CommInitRank, "ncclCommInitRank")
    GH_SYM(CommDestroy, "ncclCommDestroy")
    GH_SYM(Broadcast, "ncclBroadcast")
    GH_SYM(AllReduce, "ncclAllReduce")
    GH_SYM(GetErrorString, "ncclGetErrorString")
#undef GH_SYM
    return true;
  }
};

// One collective never carries more than this many bytes: the n^2 kinship sums of config 4 (n = 50 000) are 20 GB, and a single
// ncclAllReduce / ncclBroadcast of that size has never run on this library's communicator (round 6).  The pieces
// are issued back to back on the caller's stream (stream order = piece order on every rank).
constexpr size_t COMM_PIECE_BYTES = size_t(1) << 30;

// what the communicator has carried since it was created (gemma_hip_comm_stats); the seconds are host wall time around a
// synchronised call and are only taken when GEMMA_HIP_COMM_TIMING=1 (a synchronisation per collective otherwise costs overlap)
struct CommStats {
  long allreduce_calls = 0, allreduce_pieces = 0, bcast_calls = 0, bcast_pieces = 0;
  double allreduce_bytes = 0, bcast_bytes = 0, allreduce_s = 0, bcast_s = 0;
};

struct Comm {
  int rank = 0, world = 1;
  bool active = false;
  bool shm = false;
  RcclApi api;
  ncclComm_t nccl = nullptr;
  ShmTransport tr;        // test transport
  void *pinned = nullptr; // COMM_SHM_CHUNK staging for it
  CommStats stats;

  // failure injection for the tests of the callers' fall-backs (tests/test_gpu_bench_launch.py, tests/test_dist_gloo.py):
  // GEMMA_HIP_COMM_FAIL=init | selftest | allreduce | bcast makes that entry point return an error on EVERY rank; allreduce_large /
  // bcast_large only from 1 MiB up (a collective that fails AFTER a passed self-test)
  static bool fail_at(const char *what) {
    const char *e = getenv("GEMMA_HIP_COMM_FAIL");
    return e && strcmp(e, what) == 0;
  }
  static bool timing() {
    const char *e = getenv("GEMMA_HIP_COMM_TIMING");
    return e && atoi(e) != 0;
  }

  static bool want_shm() {
    const char *e = getenv("GEMMA_HIP_COMM");
    return e && strcmp(e, "shm") == 0;
  }

  int unique_id(void *id, std::string &err) {
    memset(id, 0, NCCL_UNIQUE_ID_BYTES);
    if (want_shm()) {
      ShmTransport::make_id(id);
      return 0;
    }
    if (!api.load(err)) return 1;
    ncclUniqueId u;
    const ncclResult_t r = api.GetUniqueId(&u);
    if (r != ncclSuccess) {
      err = std::string("ncclGetUniqueId: ") + api.GetErrorString(r);
      return 1;
    }
    memcpy(id, &u, NCCL_UNIQUE_ID_BYTES);
    return 0;
  }

  int init(const void *id, int rank_, int world_, std::string &err) {
    if (active) finalize();
    rank = rank_;
    world = world_;
    shm = false;
    if (world <= 1) {
      rank = 0;
      world = 1;
      active = true;
      return 0;
    }
    if (!id) {
      err = "comm_init: world > 1 needs the id of rank 0";
      return 1;
    }
    if (fail_at("init")) {
      err = "comm_init: failure injected (GEMMA_HIP_COMM_FAIL=init)";
      return 1;
    }
    stats = CommStats();
    if (want_shm()) {
      if (!tr.open(id, rank, world, err)) return 1;
      if (hipHostMalloc(&pinned, COMM_SHM_CHUNK, hipHostMallocDefault) != hipSuccess) {
        err = "comm_init(shm): pinned staging";
        return 1;
      }
      shm = true;
      active = true;
      return 0;
    }
    if (!api.load(err)) return 1;
    ncclUniqueId u;
    memcpy(&u, id, NCCL_UNIQUE_ID_BYTES);
    const ncclResult_t r = api.CommInitRank(&nccl, world, u, rank);
    if (r != ncclSuccess) {
      err = std::string("ncclCommInitRank: ") + api.GetErrorString(r);
      nccl = nullptr;
      return 1;
    }
    active = true;
    return 0;
  }

  // in place on buf_d (device), bytes from `root` to everyone
  int bcast(void *buf_d, size_t bytes, int root, hipStream_t s, std::string &err) {
    if (!active || world == 1 || bytes == 0) return 0;
    if (fail_at("bcast") || (bytes >= (size_t(1) << 20) && fail_at("bcast_large"))) {
      err = "comm bcast: failure injected (GEMMA_HIP_COMM_FAIL)";
      return 1;
    }
    const bool tm = timing();
    if (tm && hipStreamSynchronize(s) != hipSuccess) { err = "comm bcast: stream"; return 1; }
    const auto t0 = std::chrono::steady_clock::now();
    stats.bcast_calls += 1;
    stats.bcast_bytes += (double)bytes;
    if (!shm) {
      for (size_t off = 0; off < bytes; off += COMM_PIECE_BYTES) {
        const size_t len = bytes - off < COMM_PIECE_BYTES ? bytes - off : COMM_PIECE_BYTES;
        char *d = static_cast<char *>(buf_d) + off;
        const ncclResult_t r = api.Broadcast(d, d, len, ncclUint8, root, nccl, s);
        if (r != ncclSuccess) {
          err = std::string("ncclBroadcast: ") + api.GetErrorString(r);
          return 1;
        }
        stats.bcast_pieces += 1;
      }
      if (tm) {
        if (hipStreamSynchronize(s) != hipSuccess) { err = "comm bcast: stream"; return 1; }
        stats.bcast_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      }
      return 0;
    }
    if (hipStreamSynchronize(s) != hipSuccess) { err = "comm bcast: stream"; return 1; }
    for (size_t off = 0; off < bytes; off += COMM_SHM_CHUNK) {
      const size_t len = bytes - off < COMM_SHM_CHUNK ? bytes - off : COMM_SHM_CHUNK;
      char *d = static_cast<char *>(buf_d) + off;
      if (rank == root && hipMemcpy(pinned, d, len, hipMemcpyDeviceToHost) != hipSuccess) { err = "comm bcast: D2H"; return 1; }
      tr.bcast_chunk(pinned, len, root);
      if (rank != root && hipMemcpy(d, pinned, len, hipMemcpyHostToDevice) != hipSuccess) { err = "comm bcast: H2D"; return 1; }
      stats.bcast_pieces += 1;
    }
    if (tm) stats.bcast_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return 0;
  }

  // in place sum over ranks of `count` doubles on the device (the shm transport adds in rank order on every rank)
  int allreduce_sum(double *buf_d, size_t count, hipStream_t s, std::string &err) {
    if (!active || world == 1 || count == 0) return 0;
    if (fail_at("allreduce") || (count >= (size_t(1) << 17) && fail_at("allreduce_large"))) {
      err = "comm allreduce: failure injected (GEMMA_HIP_COMM_FAIL)";
      return 1;
    }
    const bool tm = timing();
    if (tm && hipStreamSynchronize(s) != hipSuccess) { err = "comm allreduce: stream"; return 1; }
    const auto t0 = std::chrono::steady_clock::now();
    stats.allreduce_calls += 1;
    stats.allreduce_bytes += 8.0 * (double)count;
    if (!shm) {
      const size_t piece = COMM_PIECE_BYTES / sizeof(double);
      for (size_t off = 0; off < count; off += piece) {
        const size_t len = count - off < piece ? count - off : piece;
        const ncclResult_t r = api.AllReduce(buf_d + off, buf_d + off, len, ncclFloat64, ncclSum, nccl, s);
        if (r != ncclSuccess) {
          err = std::string("ncclAllReduce: ") + api.GetErrorString(r);
          return 1;
        }
        stats.allreduce_pieces += 1;
      }
      if (tm) {
        if (hipStreamSynchronize(s) != hipSuccess) { err = "comm allreduce: stream"; return 1; }
        stats.allreduce_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      }
      return 0;
    }
    if (hipStreamSynchronize(s) != hipSuccess) { err = "comm allreduce: stream"; return 1; }
    const size_t per = COMM_SHM_CHUNK / sizeof(double);
    for (size_t off = 0; off < count; off += per) {
      const size_t len = count - off < per ? count - off : per;
      if (hipMemcpy(pinned, buf_d + off, len * 8, hipMemcpyDeviceToHost) != hipSuccess) { err = "comm allreduce: D2H"; return 1; }
      tr.allreduce_chunk(static_cast<double *>(pinned), len);
      if (hipMemcpy(buf_d + off, pinned, len * 8, hipMemcpyHostToDevice) != hipSuccess) { err = "comm allreduce: H2D"; return 1; }
      stats.allreduce_pieces += 1;
    }
    if (tm) stats.allreduce_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return 0;
  }

  // The first contact of a new communicator: ONE KiB through both collectives before anything n^2 is trusted
  // to them -- an all-reduce of (rank + 1) in 128 doubles, a broadcast of a pattern from rank 0 and one from the last rank, every
  // value checked on every rank.  Synchronous; the caller runs it under its own wall-clock deadline (a bootstrap that hangs in
  // here costs that deadline, not the run).
  int selftest(hipStream_t s, std::string &err) {
    if (!active || world == 1) return 0;
    if (fail_at("selftest")) {
      err = "comm selftest: failure injected (GEMMA_HIP_COMM_FAIL=selftest)";
      return 1;
    }
    const size_t cnt = 128;
    double *d = nullptr;
    if (hipMalloc(&d, cnt * sizeof(double)) != hipSuccess) { err = "comm selftest: device buffer"; return 1; }
    std::vector<double> h(cnt);
    int rc = 0;
    do {
      for (size_t i = 0; i < cnt; ++i) h[i] = (double)(rank + 1) * (double)(i + 1);
      if (hipMemcpyAsync(d, h.data(), cnt * 8, hipMemcpyHostToDevice, s) != hipSuccess) { err = "comm selftest: H2D"; rc = 1; break; }
      if (allreduce_sum(d, cnt, s, err)) { rc = 1; break; }
      if (hipMemcpyAsync(h.data(), d, cnt * 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        err = "comm selftest: D2H after the all-reduce"; rc = 1; break;
      }
      const double want = 0.5 * world * (world + 1);
      for (size_t i = 0; i < cnt; ++i)
        if (h[i] != want * (double)(i + 1)) { err = "comm selftest: the all-reduce returned a wrong sum"; rc = 1; break; }
      if (rc) break;
      const int roots[2] = {0, world - 1};
      for (int k = 0; k < 2 && !rc; ++k) {
        const int root = roots[k];
        for (size_t i = 0; i < cnt; ++i) h[i] = rank == root ? 1000.0 * (root + 1) + (double)i : -1.0;
        if (hipMemcpyAsync(d, h.data(), cnt * 8, hipMemcpyHostToDevice, s) != hipSuccess) { err = "comm selftest: H2D"; rc = 1; break; }
        if (bcast(d, cnt * 8, root, s, err)) { rc = 1; break; }
        if (hipMemcpyAsync(h.data(), d, cnt * 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
          err = "comm selftest: D2H after the broadcast"; rc = 1; break;
        }
        for (size_t i = 0; i < cnt; ++i)
          if (h[i] != 1000.0 * (root + 1) + (double)i) { err = "comm selftest: the broadcast delivered wrong bytes"; rc = 1; break; }
      }
    } while (0);
    (void)hipFree(d);
    return rc;
  }

  void finalize() {
    if (!active) return;
    if (shm) {
      if (pinned) (void)hipHostFree(pinned);
      pinned = nullptr;
      tr.close_segment();
    } else if (nccl) {
      api.CommDestroy(nccl);
      nccl = nullptr;
    }
    active = false;
    rank = 0;
    world = 1;
    shm = false;
  }
};

} // namespace gemma_hip
