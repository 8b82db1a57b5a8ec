// shm_comm.hpp -- a communicator whose ranks are PROCESSES of one host, set up through a POSIX shared-memory segment
// (msw_comm_create_shm).  Test infrastructure for the multi-process paths on a box with ONE GPU: RCCL refuses two ranks
// on one device, so neither the process-per-rank sharded solve nor the hipIpc set-up of the peer-write all-reduce
// (peer_comm.hpp) could run there.  Host-staged like LocalComm: a sense-reversing barrier and one staging row per rank
// in the segment, sums in rank order (the same bits on every rank).  Every wait is bounded (60 s).
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <thread>

#include "comm.hpp"

namespace msw {

struct ShmComm final : msw_comm {
  struct Hdr {
    std::atomic<uint32_t> magic, arrived, generation, failed;
    std::atomic<int64_t> created_s;  // CLOCK_REALTIME seconds at which rank 0 initialised the segment
  };
  static int64_t now_s() { return (int64_t)std::chrono::duration_cast<std::chrono::seconds>(std::chrono::system_clock::now().time_since_epoch()).count(); }
  static constexpr int64_t kFreshSeconds = 120;  // ranks of one launch start within seconds of each other
  static constexpr size_t kRowBytes = 1u << 20;  // staging row per rank; longer messages go in pieces
  static constexpr uint32_t kMagic = 0x6d737763u;
  int r, n;
  std::string name;
  size_t bytes = 0;
  Hdr *hdr = nullptr;
  unsigned char *rows = nullptr;

  ShmComm(const std::string &name_, int rank_, int nranks) : r(rank_), n(nranks), name(name_) {
    if (name.empty() || name[0] != '/') throw HipError("msw_comm_create_shm: the name must start with '/'");
    bytes = 4096 + (size_t)n * kRowBytes;
    const auto t0 = std::chrono::steady_clock::now();
    auto late = [&] { return std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60); };
    int fd = -1;
    if (r == 0) {
      (void)shm_unlink(name.c_str());
      fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) throw HipError("msw_comm_create_shm: cannot create " + name);
    }
    // rank 0 creates the segment; the others wait for it, for its size and for its header -- and refuse a segment of
    // the same name that an earlier run left behind (it crashed between creating and unlinking the name): such a
    // segment is initialised, possibly marked failed, and every rank that joined it would wait out its 60 s.  A
    // segment is taken only if rank 0 stamped it within the last kFreshSeconds; an older one is dropped and the
    // name polled again until rank 0 of THIS launch has replaced it (it unlinks the name first).
    void *p = MAP_FAILED;
    for (;;) {
      if (r != 0) {
        fd = shm_open(name.c_str(), O_RDWR, 0600);
        struct stat st;
        if (!(fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= bytes)) {
          if (fd >= 0) close(fd);
          if (late()) throw HipError("msw_comm_create_shm: rank 0 did not create " + name);
          std::this_thread::sleep_for(std::chrono::milliseconds(2));
          continue;
        }
      }
      p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      close(fd);
      if (p == MAP_FAILED) throw HipError("msw_comm_create_shm: mmap failed");
      hdr = static_cast<Hdr *>(p);
      if (r == 0) {
        hdr->arrived = 0, hdr->generation = 0, hdr->failed = 0;
        hdr->created_s.store(now_s());
        hdr->magic.store(kMagic, std::memory_order_release);
        break;
      }
      bool fresh = false;
      while (!late()) {
        if (hdr->magic.load(std::memory_order_acquire) == kMagic) {
          fresh = now_s() - hdr->created_s.load() <= kFreshSeconds;
          break;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
      }
      if (fresh) break;
      munmap(p, bytes);
      hdr = nullptr;
      if (late()) throw HipError("msw_comm_create_shm: no freshly initialised segment " + name + " appeared (a stale one of an earlier run was ignored)");
      std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
    rows = reinterpret_cast<unsigned char *>(hdr) + 4096;
    barrier();  // every rank has mapped the segment: rank 0 may unlink the name
    if (r == 0) (void)shm_unlink(name.c_str());
  }
  ~ShmComm() override {
    if (hdr) munmap(hdr, bytes);
  }
  int rank() const override { return r; }
  int size() const override { return n; }
  void abort() override {
    if (hdr) hdr->failed.store(1);
  }
  void barrier() {
    const auto t0 = std::chrono::steady_clock::now();
    if (hdr->failed.load()) throw HipError("ShmComm: another rank of the group failed");
    const uint32_t g = hdr->generation.load(std::memory_order_acquire);
    if (hdr->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)n) {
      hdr->arrived.store(0, std::memory_order_relaxed);
      hdr->generation.fetch_add(1, std::memory_order_acq_rel);
      return;
    }
    for (uint32_t spins = 0; hdr->generation.load(std::memory_order_acquire) == g; ++spins) {
      if (hdr->failed.load()) throw HipError("ShmComm: another rank of the group failed");
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) {
        hdr->failed.store(1);
        throw HipError("ShmComm: a rank did not reach the barrier within 60 s");
      }
      if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
  }
  // rows of up to kRowBytes at a time: recv[k * cnt + i] = rank k's send[i]
  void allgather_host(const double *send, size_t cnt, double *recv) override {
    const size_t per = kRowBytes / sizeof(double);
    for (size_t o = 0; o == 0 || o < cnt; o += per) {
      const size_t m = std::min(per, cnt - o);
      std::memcpy(rows + (size_t)r * kRowBytes, send + o, m * sizeof(double));
      barrier();
      for (int k = 0; k < n; ++k) std::memcpy(recv + (size_t)k * cnt + o, rows + (size_t)k * kRowBytes, m * sizeof(double));
      barrier();  // every rank has copied: the rows may be overwritten
    }
  }
  template <class T>
  void allreduce_t(T *dev, size_t cnt, hipStream_t stream) {
    static_assert(sizeof(T) == sizeof(double), "staged as 8-byte words");
    std::vector<double> mine(cnt), all(cnt * (size_t)n);
    MSW_HIP(hipMemcpyAsync(mine.data(), dev, cnt * sizeof(T), hipMemcpyDeviceToHost, stream));
    MSW_HIP(hipStreamSynchronize(stream));
    allgather_host(mine.data(), cnt, all.data());
    std::vector<T> sum(cnt, T(0));
    for (int k = 0; k < n; ++k) {  // T = double: summed in rank order; T = uint64_t: exact
      const T *src = reinterpret_cast<const T *>(all.data() + (size_t)k * cnt);
      for (size_t i = 0; i < cnt; ++i) sum[i] += src[i];
    }
    MSW_HIP(hipMemcpyAsync(dev, sum.data(), cnt * sizeof(T), hipMemcpyHostToDevice, stream));
    MSW_HIP(hipStreamSynchronize(stream));
  }
  void allreduce(double *dev, size_t cnt, hipStream_t stream) override { allreduce_t(dev, cnt, stream); }
  void allreduce_u64(uint64_t *dev, size_t cnt, hipStream_t stream) override { allreduce_t(dev, cnt, stream); }
};

}  // namespace msw
