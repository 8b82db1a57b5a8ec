// comm.hpp -- communicators for the EC-sharded single solve: every rank
// holds a contiguous block of ECs; per iteration the ranks exchange ONE scalar (|g|^2 after
// pass A) and ONE (G + 4)-vector (column sums + ELBO terms after pass B).  All O(G) kernels then
// run redundantly on identical data, so every rank takes the same decisions and the loops stay
// in lock-step without any further traffic.
//   * RcclComm : ncclAllReduce on the solve stream (RCCL over xGMI), one process per GPU.
//   * LocalComm: ranks are host threads of one process (tests on a single GPU, or several GPUs
//                driven from one process); host-staged, summed in rank order (bitwise identical
//                on every rank).
//   * PeerComm (peer_comm.hpp, MSWEEP_ALLREDUCE=peer): wraps either; one kernel per all-reduce that writes into the
//                peers' inboxes.
#pragma once
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>

#include "common.hpp"

struct msw_comm {
  virtual ~msw_comm() = default;
  virtual int rank() const = 0;
  virtual int size() const = 0;
  // in-place sum over ranks of n doubles in device memory, ordered on `stream`
  virtual void allreduce(double *dev, size_t n, hipStream_t stream) = 0;
  // the same for unsigned 64-bit integers (group hit counts of the sharded likelihood build)
  virtual void allreduce_u64(uint64_t *dev, size_t n, hipStream_t stream) = 0;
  // both at once (one fused launch under RCCL): na integers and nb doubles, in place
  virtual void allreduce_mixed(uint64_t *a, size_t na, double *b, size_t nb, hipStream_t stream) {
    allreduce_u64(a, na, stream);
    allreduce(b, nb, stream);
  }
  // recv[r * n .. (r + 1) * n) = rank r's send[0 .. n); HOST buffers (the bootstrap abundances live on
  // the host: include/Sample.hpp:157), blocking
  virtual void allgather_host(const double *send, size_t n, double *recv) = 0;
  // a rank that fails outside a collective calls this so that its peers do not wait for ever
  virtual void abort() {}
  // called after the solve stream has been synchronised: a collective that failed ON THE DEVICE throws here
  virtual void check() {}
  // the ranks meet on the HOST before a stretch of device-side collectives (the start of a sharded solve or of its
  // continuation): transports whose waits are bounded on the device (peer_comm.hpp) must not charge the host skew
  // between the ranks' calls -- seconds, when a caller does other work between two solves -- to that bound
  virtual void rendezvous() {}
};

namespace msw {

struct RcclComm final : msw_comm {
  ncclComm_t comm = nullptr;
  int r = 0, n = 1;
  RcclComm(const ncclUniqueId &id, int rank_, int nranks) : r(rank_), n(nranks) {
    const ncclResult_t rc = ncclCommInitRank(&comm, nranks, id, rank_);
    if (rc != ncclSuccess) throw HipError(std::string("ncclCommInitRank: ") + ncclGetErrorString(rc));
  }
  ~RcclComm() override {
    if (ag_stream) (void)hipStreamDestroy(ag_stream);
    if (comm) (void)ncclCommDestroy(comm);
  }
  int rank() const override { return r; }
  int size() const override { return n; }
  // a rank that gives up between collectives: ncclCommAbort fails the operations its peers have in flight (or
  // enqueue later) instead of leaving them blocked in ncclAllReduce / ncclAllGather for ever
  void abort() override {
    if (comm) (void)ncclCommAbort(comm);
    comm = nullptr;  // the destructor skips ncclCommDestroy; further collectives on this rank fail below
  }
  void need_comm() const {
    if (!comm) throw HipError("RCCL communicator was aborted after a failure on this rank");
  }
  void allreduce(double *dev, size_t cnt, hipStream_t stream) override {
    need_comm();
    const ncclResult_t rc = ncclAllReduce(dev, dev, cnt, ncclDouble, ncclSum, comm, stream);
    if (rc != ncclSuccess) throw HipError(std::string("ncclAllReduce: ") + ncclGetErrorString(rc));
  }
  void allreduce_u64(uint64_t *dev, size_t cnt, hipStream_t stream) override {
    need_comm();
    const ncclResult_t rc = ncclAllReduce(dev, dev, cnt, ncclUint64, ncclSum, comm, stream);
    if (rc != ncclSuccess) throw HipError(std::string("ncclAllReduce: ") + ncclGetErrorString(rc));
  }
  void allreduce_mixed(uint64_t *a, size_t na, double *b, size_t nb, hipStream_t stream) override {
    need_comm();
    ncclResult_t rc = ncclGroupStart();
    if (rc == ncclSuccess) rc = ncclAllReduce(a, a, na, ncclUint64, ncclSum, comm, stream);
    if (rc == ncclSuccess) rc = ncclAllReduce(b, b, nb, ncclDouble, ncclSum, comm, stream);
    const ncclResult_t rc2 = ncclGroupEnd();
    if (rc == ncclSuccess) rc = rc2;
    if (rc != ncclSuccess) throw HipError(std::string("ncclAllReduce (grouped): ") + ncclGetErrorString(rc));
  }
  // staging buffers and stream are kept: a second call costs two small copies and the collective
  DevBuf<double> ag_send, ag_recv;
  hipStream_t ag_stream = nullptr;
  void allgather_host(const double *send, size_t cnt, double *recv) override {
    need_comm();
    if (!ag_stream) MSW_HIP(hipStreamCreateWithFlags(&ag_stream, hipStreamNonBlocking));
    ag_send.alloc(cnt);
    ag_recv.alloc(cnt * (size_t)n);
    MSW_HIP(hipMemcpyAsync(ag_send.p, send, cnt * sizeof(double), hipMemcpyHostToDevice, ag_stream));
    const ncclResult_t rc = ncclAllGather(ag_send.p, ag_recv.p, cnt, ncclDouble, comm, ag_stream);
    if (rc != ncclSuccess) throw HipError(std::string("ncclAllGather: ") + ncclGetErrorString(rc));
    MSW_HIP(hipMemcpyAsync(recv, ag_recv.p, cnt * (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ag_stream));
    MSW_HIP(hipStreamSynchronize(ag_stream));
  }
  int count() const {  // what RCCL itself says the communicator spans
    int c = 0;
    return comm && ncclCommCount(comm, &c) == ncclSuccess ? c : -1;
  }
};

struct LocalGroup {
  int n;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t generation = 0;
  bool failed = false;  // a rank gave up: every waiter (now and later) fails instead of blocking
  std::vector<std::vector<double>> stage;
  std::vector<const double *> gather_src;
  explicit LocalGroup(int n_) : n(n_), stage(n_), gather_src(n_, nullptr) {}
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (failed) throw HipError("LocalComm: another rank of the group failed");
    const uint64_t g = generation;
    if (++arrived == n) {
      arrived = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != g || failed; });
      if (generation == g) throw HipError("LocalComm: another rank of the group failed");
    }
  }
  void fail() {
    std::lock_guard<std::mutex> lk(mu);
    failed = true;
    cv.notify_all();
  }
};

struct LocalComm final : msw_comm {
  std::shared_ptr<LocalGroup> grp;
  int r;
  LocalComm(std::shared_ptr<LocalGroup> g, int rank_) : grp(std::move(g)), r(rank_) {}
  int rank() const override { return r; }
  int size() const override { return grp->n; }
  void abort() override { grp->fail(); }
  // T = double: summed in rank order; T = uint64_t: exact
  template <class T>
  void allreduce_t(T *dev, size_t cnt, hipStream_t stream) {
    static_assert(sizeof(T) == sizeof(double), "staged as 8-byte words");
    std::vector<double> &mine = grp->stage[r];
    mine.resize(cnt);
    MSW_HIP(hipMemcpyAsync(mine.data(), dev, cnt * sizeof(T), hipMemcpyDeviceToHost, stream));
    MSW_HIP(hipStreamSynchronize(stream));
    grp->barrier();
    std::vector<T> sum(cnt, T(0));
    bool same = true;
    for (int k = 0; k < grp->n; ++k) {
      same = same && grp->stage[k].size() == cnt;
      if (!same) break;
      const T *src = reinterpret_cast<const T *>(grp->stage[k].data());
      for (size_t i = 0; i < cnt; ++i) sum[i] += src[i];
    }
    grp->barrier();  // everyone has read the staging buffers (and reached the same verdict on the sizes)
    if (!same) throw HipError("LocalComm: ranks disagree on the message size");
    MSW_HIP(hipMemcpyAsync(dev, sum.data(), cnt * sizeof(T), hipMemcpyHostToDevice, stream));
    MSW_HIP(hipStreamSynchronize(stream));
  }
  void allreduce(double *dev, size_t cnt, hipStream_t stream) override { allreduce_t(dev, cnt, stream); }
  void allreduce_u64(uint64_t *dev, size_t cnt, hipStream_t stream) override { allreduce_t(dev, cnt, stream); }
  void allgather_host(const double *send, size_t cnt, double *recv) override {
    grp->gather_src[r] = send;
    grp->barrier();
    for (int k = 0; k < grp->n; ++k) std::memcpy(recv + (size_t)k * cnt, grp->gather_src[k], cnt * sizeof(double));
    grp->barrier();  // every rank has copied: the send buffers may go
  }
};

}  // namespace msw
