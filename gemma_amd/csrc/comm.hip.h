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

#include <cstdlib>
#include <cstring>
#include <string>

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

struct Comm {
  int rank = 0, world = 1;
  bool active = false;
  bool shm = false;
  RcclApi api;
  ncclComm_t nccl = nullptr;
  ShmTransport tr;        // test transport
  void *pinned = nullptr; // COMM_SHM_CHUNK staging for it

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
    if (!shm) {
      const ncclResult_t r = api.Broadcast(buf_d, buf_d, bytes, ncclUint8, root, nccl, s);
      if (r != ncclSuccess) {
        err = std::string("ncclBroadcast: ") + api.GetErrorString(r);
        return 1;
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
    }
    return 0;
  }

  // in place sum over ranks of `count` doubles on the device (the shm transport adds in rank order on every rank)
  int allreduce_sum(double *buf_d, size_t count, hipStream_t s, std::string &err) {
    if (!active || world == 1 || count == 0) return 0;
    if (!shm) {
      const ncclResult_t r = api.AllReduce(buf_d, buf_d, count, ncclFloat64, ncclSum, nccl, s);
      if (r != ncclSuccess) {
        err = std::string("ncclAllReduce: ") + api.GetErrorString(r);
        return 1;
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
    }
    return 0;
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
