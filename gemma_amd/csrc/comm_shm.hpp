// Host shared-memory transport for the two collectives of the path (broadcast, all-reduce of doubles) between the
// processes of one node: a POSIX segment named by the communicator id, one payload slot per rank, a process-shared
// barrier.  TEST transport only (GEMMA_HIP_COMM=shm): RCCL refuses two ranks on one device, so this is what lets
// `-gpus 2 -samegpu` and the 2-rank tests run the real protocol on a 1-GPU box; tests/cpp/abi_double.cpp (the
// CPU test double of the C ABI) uses it too, on host arrays.  Pure POSIX: no HIP here.
#pragma once
#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace gemma_hip {

constexpr size_t COMM_SHM_CHUNK = size_t(64) << 20; // payload bytes per rank slot
constexpr size_t COMM_ID_BYTES = 128;               // == NCCL_UNIQUE_ID_BYTES == GEMMA_HIP_COMM_ID_BYTES

class ShmTransport {
public:
  static void make_id(void *id) {
    // a name no earlier run can have left behind: pid + 64 bits from the kernel's generator (a stale segment of a crashed run
    // with a recycled pid would carry ready = 1 and an initialised barrier)
    memset(id, 0, COMM_ID_BYTES);
    unsigned long long r = 0;
    int fd = ::open("/dev/urandom", O_RDONLY);
    if (fd >= 0) {
      if (read(fd, &r, sizeof r) != (ssize_t)sizeof r) r = 0;
      ::close(fd);
    }
    if (!r) {
      struct timespec ts;
      clock_gettime(CLOCK_MONOTONIC, &ts);
      r = (unsigned long long)ts.tv_nsec * 2654435761ull ^ (unsigned long long)ts.tv_sec;
    }
    snprintf(static_cast<char *>(id), COMM_ID_BYTES, "/gemma_hip_comm_%d_%016llx", (int)getpid(), r);
  }
  bool open(const void *id, int rank, int world, std::string &err) {
    rank_ = rank;
    world_ = world;
    name_.assign(static_cast<const char *>(id), strnlen(static_cast<const char *>(id), COMM_ID_BYTES));
    bytes_ = sizeof(Header) + (size_t)world * COMM_SHM_CHUNK;
    int fd = -1;
    if (rank == 0) {
      shm_unlink(name_.c_str()); // never adopt a segment of that name: create it afresh, exclusively
      fd = shm_open(name_.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0 || ftruncate(fd, (off_t)bytes_) != 0) {
        err = "comm(shm): cannot create " + name_;
        return false;
      }
    } else {
      for (int t = 0; t < 120000 && fd < 0; ++t) { // rank 0 may be slower: wait up to two minutes
        fd = shm_open(name_.c_str(), O_RDWR, 0600);
        struct stat st;
        if (fd >= 0 && (fstat(fd, &st) != 0 || (size_t)st.st_size < bytes_)) {
          close(fd);
          fd = -1;
        }
        if (fd < 0) usleep(1000);
      }
      if (fd < 0) {
        err = "comm(shm): segment " + name_ + " never appeared";
        return false;
      }
    }
    void *m = mmap(nullptr, bytes_, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) {
      err = "comm(shm): mmap failed";
      return false;
    }
    hdr_ = static_cast<Header *>(m);
    slots_ = static_cast<char *>(m) + sizeof(Header);
    if (rank == 0) {
      hdr_->ready = 0; // (a fresh segment is zero-filled; stated for the reader)
      __sync_synchronize();
      pthread_barrierattr_t at;
      pthread_barrierattr_init(&at);
      pthread_barrierattr_setpshared(&at, PTHREAD_PROCESS_SHARED);
      pthread_barrier_init(&hdr_->barrier, &at, (unsigned)world);
      pthread_barrierattr_destroy(&at);
      hdr_->world = world;
      __sync_synchronize();
      hdr_->ready = 1;
    } else {
      for (int t = 0; t < 120000 && !hdr_->ready; ++t) usleep(1000);
      if (!hdr_->ready || hdr_->world != world) {
        err = "comm(shm): rank 0 never initialised the segment";
        return false;
      }
    }
    barrier();
    if (rank == 0) shm_unlink(name_.c_str()); // every rank has it mapped now
    return true;
  }
  void close_segment() {
    if (hdr_) munmap(hdr_, bytes_);
    hdr_ = nullptr;
    slots_ = nullptr;
  }
  bool is_open() const { return hdr_ != nullptr; }
  void barrier() {
    if (hdr_) pthread_barrier_wait(&hdr_->barrier);
  }
  // one chunk (len <= COMM_SHM_CHUNK bytes) of a broadcast: the root's bytes end up in every rank's `chunk`
  void bcast_chunk(void *chunk, size_t len, int root) {
    if (rank_ == root) memcpy(slots_, chunk, len);
    barrier();
    if (rank_ != root) memcpy(chunk, slots_, len);
    barrier();
  }
  // one chunk (len <= COMM_SHM_CHUNK / 8 doubles) of a sum over ranks, added in rank order on every rank
  void allreduce_chunk(double *chunk, size_t len) {
    memcpy(slots_ + (size_t)rank_ * COMM_SHM_CHUNK, chunk, len * 8);
    barrier();
    for (size_t i = 0; i < len; ++i) chunk[i] = 0.0;
    for (int r = 0; r < world_; ++r) {
      const double *src = reinterpret_cast<const double *>(slots_ + (size_t)r * COMM_SHM_CHUNK);
      for (size_t i = 0; i < len; ++i) chunk[i] += src[i];
    }
    barrier();
  }
  // whole host arrays (the test double)
  void bcast_host(void *buf, size_t bytes, int root) {
    for (size_t off = 0; off < bytes; off += COMM_SHM_CHUNK)
      bcast_chunk(static_cast<char *>(buf) + off, bytes - off < COMM_SHM_CHUNK ? bytes - off : COMM_SHM_CHUNK, root);
  }
  void allreduce_host(double *buf, size_t count) {
    const size_t per = COMM_SHM_CHUNK / 8;
    for (size_t off = 0; off < count; off += per) allreduce_chunk(buf + off, count - off < per ? count - off : per);
  }

private:
  struct Header {
    pthread_barrier_t barrier;
    int world;
    int ready;
  };
  int rank_ = 0, world_ = 1;
  std::string name_;
  size_t bytes_ = 0;
  Header *hdr_ = nullptr;
  char *slots_ = nullptr;
};

} // namespace gemma_hip
