// peer_comm.hpp -- the one-shot peer-write all-reduce of SURVEY.md 5.8, behind MSWEEP_ALLREDUCE=peer (RCCL stays the
// default).  The EC-sharded solve exchanges one scalar and one (3G + 4)-word vector per iteration (comm.hpp): tens of
// kilobytes, where a ring's n - 1 hops of latency are the whole cost.  Here every rank owns an INBOX in its HBM with one
// slot per source rank; one kernel per collective
//   1. writes the rank's message into its slot of every peer's inbox (xGMI stores; 2 KB chunks, one workgroup each),
//   2. raises a per-chunk flag there carrying the message's sequence number,
//   3. waits for the same chunk from every source, and sums the n contributions IN RANK ORDER: integer words exactly,
//      doubles in the same order on every rank, so the result is the same bits everywhere and the replicated O(G)
//      kernels keep their lock-step.
// No grid-wide step: chunk c of every rank only talks to chunk c of the others.  Two slot sets alternate with the parity
// of the sequence number: a source can be one message ahead of a receiver, never two (it needs the receiver's flag of the
// message in between).  Every wait is bounded (MSWEEP_PEER_TIMEOUT_MS, default 2000): a rank that never arrives turns
// into an error on the host, not a hung grid.  The bound covers device-side skew only: the ranks meet on the host
// (rendezvous(): one token through the set-up communicator) at the start of every sharded solve or continuation, so a
// caller that comes back seconds after its peers costs them nothing; and once a wait has timed out every later
// collective of the rank returns at once (the status word), so a dead peer costs ONE timeout, not one per queued kernel.
//   * one process per GPU : inboxes are exchanged as hipIpc handles through the RCCL communicator that also keeps the
//                           host all-gather and the abort path;
//   * thread-ranks        : plain device pointers (peer access enabled between devices of the process).
#pragma once
#include <chrono>

#include "comm.hpp"

namespace msw {

constexpr int kPeerMaxRanks = 16;
constexpr int kPeerChunk = 256;  // 8-byte words per workgroup: 2 KB, one word per thread

struct PeerBoxes {
  unsigned long long *box[kPeerMaxRanks];
};

// inbox words: data[2][n][cap] then flags[2][n][cap / kPeerChunk]
__global__ __launch_bounds__(kPeerChunk) void k_peer_allreduce(PeerBoxes pb, int rank, int n, size_t cap,
                                                                unsigned long long seq, unsigned long long *a, size_t na,
                                                                double *b, size_t nb, unsigned long long timeout_ticks,
                                                                unsigned *status) {
  __shared__ int s_bad;
  const int t = threadIdx.x;
  const size_t c = blockIdx.x, w = c * kPeerChunk + t, chunks = cap / kPeerChunk;
  const size_t par = (size_t)(seq & 1);
  const bool live = w < na + nb;
  // a wait of this rank has already timed out (check() clears the word on the host): the solve is lost, its queued
  // collectives drain without waiting again
  if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return;
  unsigned long long v = 0;
  if (live) v = w < na ? a[w] : (unsigned long long)__double_as_longlong(b[w - na]);
  if (t == 0) s_bad = 0;
  for (int d = 0; d < n; ++d)
    if (d != rank && live)
      __hip_atomic_store(pb.box[d] + (par * n + rank) * cap + w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: the payload has left for the peers' memory
  __syncthreads();
  if (t < n && t != rank)
    __hip_atomic_store(pb.box[t] + 2 * (size_t)n * cap + (par * n + rank) * chunks + c, seq, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  if (t < n && t != rank) {
    const unsigned long long *flag = pb.box[rank] + 2 * (size_t)n * cap + (par * n + t) * chunks + c;
    const unsigned long long t0 = wall_clock64();
    unsigned spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
      __builtin_amdgcn_s_sleep(4);
      if ((++spins & 63u) == 0 && wall_clock64() - t0 > timeout_ticks) {  // every wave reaches an exit
        s_bad = 1;
        break;
      }
    }
  }
  __syncthreads();
  if (s_bad) {
    if (t == 0) __hip_atomic_fetch_or(status, 1u << (rank & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  if (!live) return;
  const unsigned long long *mine = pb.box[rank] + par * n * cap + w;
  if (w < na) {
    unsigned long long s = 0;
    for (int r = 0; r < n; ++r)
      s += r == rank ? v : __hip_atomic_load(mine + (size_t)r * cap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    a[w] = s;
  } else {
    double s = 0.0;
    for (int r = 0; r < n; ++r)
      s += __longlong_as_double((long long)(r == rank ? v : __hip_atomic_load(mine + (size_t)r * cap, __ATOMIC_RELAXED,
                                                                                 __HIP_MEMORY_SCOPE_SYSTEM)));
    b[w - na] = s;
  }
}

// Wraps the communicator that set the ranks up (RcclComm or LocalComm): its host all-gather exchanges the inboxes and
// serves allgather_host; the all-reduces are the kernel above.
struct PeerComm final : msw_comm {
  std::unique_ptr<msw_comm> base;
  bool ipc;  // one process per rank: inboxes travel as hipIpc handles
  size_t cap = 0;
  unsigned long long seq = 0;
  unsigned long long *mine = nullptr;
  PeerBoxes boxes{};
  std::vector<void *> opened;
  std::vector<unsigned long long *> retired;  // outgrown inboxes: freed with the communicator (hipFree waits for the
                                              // whole device, and a thread-rank's peers may be waiting for this rank)
  unsigned *status = nullptr, *status_dev = nullptr;  // pinned: the kernel's timeout word, read by check()
  unsigned long long timeout_ticks;
  size_t launches = 0;
  int skip_rank = -1;  // MSWEEP_PEER_TEST_SKIP_RANK (tests only)

  PeerComm(std::unique_ptr<msw_comm> b, bool ipc_) : base(std::move(b)), ipc(ipc_) {
    if (base->size() > kPeerMaxRanks) throw HipError("MSWEEP_ALLREDUCE=peer: at most 16 ranks");
    const char *e = std::getenv("MSWEEP_PEER_TIMEOUT_MS");
    const double ms = e && std::atof(e) > 0 ? std::atof(e) : 30000.0;  // (ShmComm's host waits: 60 s; RCCL: none)
    timeout_ticks = (unsigned long long)(ms * 1e5);  // wall_clock64: 100 MHz
    if (const char *s = std::getenv("MSWEEP_PEER_TEST_SKIP_RANK")) skip_rank = std::atoi(s);
  }
  ~PeerComm() override { release(); }
  int rank() const override { return base->rank(); }
  int size() const override { return base->size(); }
  void abort() override { base->abort(); }
  void allgather_host(const double *send, size_t cnt, double *recv) override { base->allgather_host(send, cnt, recv); }
  void rendezvous() override {
    if (dynamic_cast<LocalComm *>(base.get())) return;  // thread-ranks launch every collective together already (launch())
    double token = 0.0;
    std::vector<double> all((size_t)size());
    base->allgather_host(&token, 1, all.data());
  }

  void release() {
    for (void *p : opened) (void)hipIpcCloseMemHandle(p);
    opened.clear();
    if (mine) (void)hipFree(mine);
    for (auto *p : retired) (void)hipFree(p);
    retired.clear();
    mine = nullptr;
    if (status) (void)hipHostFree(status);
    status = nullptr;
    cap = 0;
  }

  // Collective (every rank reaches it at the same call with the same `words`): a bigger inbox on every rank.  The
  // stream is drained first: nothing of this rank is in flight, and -- a message completes only with every peer's flag
  // -- nothing of a peer still targets the old inbox.
  void ensure(size_t words, hipStream_t stream) {
    if (words <= cap) return;
    MSW_HIP(hipStreamSynchronize(stream));
    check();
    const int n = size(), r = rank();
    const size_t want = std::max<size_t>(4096, (words + kPeerChunk - 1) / kPeerChunk * kPeerChunk);
    const size_t total = 2 * (size_t)n * want + 2 * (size_t)n * (want / kPeerChunk);
    unsigned long long *fresh = nullptr;
    // fine-grained: stores arriving over xGMI are seen by a kernel already running here
    MSW_HIP(hipExtMallocWithFlags(reinterpret_cast<void **>(&fresh), total * 8, hipDeviceMallocFinegrained));
    MSW_HIP(hipMemsetAsync(fresh, 0, total * 8, stream));
    MSW_HIP(hipStreamSynchronize(stream));  // (no device-wide wait: thread-ranks share the device)
    if (!status) {
      MSW_HIP(hipHostMalloc(reinterpret_cast<void **>(&status), sizeof(unsigned), hipHostMallocMapped));
      *status = 0;
      MSW_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&status_dev), status, 0));
    }
    int dev = 0;
    MSW_HIP(hipGetDevice(&dev));
    // one record per rank: {handle or pointer (64 B), device, process} as 10 doubles' worth of bytes
    constexpr size_t kRec = 10;
    double rec[kRec] = {};
    if (ipc) {
      hipIpcMemHandle_t hd;
      static_assert(sizeof(hd) <= 64, "hipIpcMemHandle_t is 64 bytes");
      if (hipIpcGetMemHandle(&hd, fresh) != hipSuccess) {
        // (no ordinary allocation instead: coarse-grained memory is coherent for its OWNER only at kernel boundaries --
        // a peer's xGMI store does not invalidate the owner's L2, so the waiting kernel could miss the flag, or see
        // the flag with a stale payload)
        (void)hipGetLastError();
        (void)hipFree(fresh);
        throw HipError("MSWEEP_ALLREDUCE=peer: this runtime cannot export fine-grained device memory over hipIpc; "
                       "use MSWEEP_ALLREDUCE=rccl (the default)");
      }
      std::memcpy(rec, &hd, sizeof hd);
    } else {
      std::memcpy(rec, &fresh, sizeof fresh);
    }
    const long long devll = dev;
    std::memcpy(rec + 8, &devll, 8);
    std::vector<double> all(kRec * (size_t)n);
    base->allgather_host(rec, kRec, all.data());
    std::vector<void *> fresh_opened;
    PeerBoxes nb{};
    for (int k = 0; k < n; ++k) {
      if (k == r) {
        nb.box[k] = fresh;
        continue;
      }
      const double *pr = all.data() + kRec * (size_t)k;
      if (ipc) {
        hipIpcMemHandle_t hd;
        std::memcpy(&hd, pr, sizeof hd);
        void *p = nullptr;
        MSW_HIP(hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess));
        fresh_opened.push_back(p);
        nb.box[k] = static_cast<unsigned long long *>(p);
      } else {
        unsigned long long *p = nullptr;
        long long pdev = 0;
        std::memcpy(&p, pr, sizeof p);
        std::memcpy(&pdev, pr + 8, 8);
        if ((int)pdev != dev) {
          const hipError_t rc = hipDeviceEnablePeerAccess((int)pdev, 0);
          if (rc != hipSuccess && rc != hipErrorPeerAccessAlreadyEnabled) MSW_HIP(rc);
          (void)hipGetLastError();
        }
        nb.box[k] = p;
      }
    }
    // every rank has mapped the new inboxes before anyone writes into one, and has stopped using the old ones
    double token = 0.0;
    std::vector<double> tokens((size_t)n);
    base->allgather_host(&token, 1, tokens.data());
    for (void *p : opened) (void)hipIpcCloseMemHandle(p);
    opened = std::move(fresh_opened);
    if (mine) retired.push_back(mine);
    mine = fresh;
    boxes = nb;
    cap = want;
  }

  void launch(unsigned long long *a, size_t na, double *b, size_t nb, hipStream_t stream) {
    const size_t words = na + nb;
    if (words == 0) return;
    ensure(words, stream);
    // Thread-ranks share one process, often one device: a rank whose kernel waits here while a peer has yet to launch
    // its own must not meet that peer in a device-wide wait (hipFree, a growing buffer).  Launch together.
    if (LocalComm *lc = dynamic_cast<LocalComm *>(base.get())) lc->grp->barrier();
    ++seq;
    if (skip_rank == rank()) return;  // fault injection of the tests: this rank's message never leaves
    ++launches;
    hipLaunchKernelGGL(k_peer_allreduce, dim3((unsigned)((words + kPeerChunk - 1) / kPeerChunk)), dim3(kPeerChunk), 0,
                       stream, boxes, rank(), size(), cap, seq, a, na, b, nb, timeout_ticks, status_dev);
    MSW_HIP(hipGetLastError());
  }
  void allreduce(double *dev, size_t cnt, hipStream_t stream) override { launch(nullptr, 0, dev, cnt, stream); }
  // the integer-only all-reduces belong to the sharded likelihood BUILD (host_build.inc: status words, group hit
  // counts; twice per build, between allocations that synchronise the device): they stay on the set-up communicator
  void allreduce_u64(uint64_t *dev, size_t cnt, hipStream_t stream) override { base->allreduce_u64(dev, cnt, stream); }
  void allreduce_mixed(uint64_t *a, size_t na, double *b, size_t nb, hipStream_t stream) override {
    launch(reinterpret_cast<unsigned long long *>(a), na, b, nb, stream);
  }
  // after a stream synchronisation: did a wait of this rank time out?
  void check() override {
    if (status && *status) {
      *status = 0;
      throw HipError("peer all-reduce: a rank did not deliver its message within MSWEEP_PEER_TIMEOUT_MS "
                     "(a peer failed, or the ranks' kernels cannot run side by side on this device)");
    }
  }
};

inline bool peer_allreduce_requested() {
  const char *e = std::getenv("MSWEEP_ALLREDUCE");
  if (!e || !*e || std::string(e) == "rccl") return false;
  if (std::string(e) == "peer") return true;
  throw HipError(std::string("MSWEEP_ALLREDUCE=") + e + ": expected `rccl` or `peer`");
}

}  // namespace msw
