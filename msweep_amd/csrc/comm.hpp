// comm.hpp -- communicators for the EC-sharded single solve: every rank
// holds a contiguous block of ECs; per iteration the ranks exchange ONE scalar (|g|^2 after
// pass A) and ONE (G + 4)-vector (column sums + ELBO terms after pass B).  All O(G) kernels then
// run redundantly on identical data, so every rank takes the same decisions and the loops stay
// in lock-step without any further traffic.
//   * RcclComm : ncclAllReduce on the solve stream (RCCL over xGMI), one process per GPU.
//   * LocalComm: ranks are host threads of one process (tests on a single GPU, or several GPUs
//                driven from one process); host-staged, summed in rank order (bitwise identical
//                on every rank).
#pragma once
#include <rccl/rccl.h>

#include <condition_variable>
#include <memory>
#include <mutex>

#include "common.hpp"

struct msw_comm {
  virtual ~msw_comm() = default;
  virtual int rank() const = 0;
  virtual int size() const = 0;
  // in-place sum over ranks of n doubles in device memory, ordered on `stream`
  virtual void allreduce(double *dev, size_t n, hipStream_t stream) = 0;
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
    if (comm) (void)ncclCommDestroy(comm);
  }
  int rank() const override { return r; }
  int size() const override { return n; }
  void allreduce(double *dev, size_t cnt, hipStream_t stream) override {
    const ncclResult_t rc = ncclAllReduce(dev, dev, cnt, ncclDouble, ncclSum, comm, stream);
    if (rc != ncclSuccess) throw HipError(std::string("ncclAllReduce: ") + ncclGetErrorString(rc));
  }
};

struct LocalGroup {
  int n;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t generation = 0;
  std::vector<std::vector<double>> stage;
  explicit LocalGroup(int n_) : n(n_), stage(n_) {}
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const uint64_t g = generation;
    if (++arrived == n) {
      arrived = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != g; });
    }
  }
};

struct LocalComm final : msw_comm {
  std::shared_ptr<LocalGroup> grp;
  int r;
  LocalComm(std::shared_ptr<LocalGroup> g, int rank_) : grp(std::move(g)), r(rank_) {}
  int rank() const override { return r; }
  int size() const override { return grp->n; }
  void allreduce(double *dev, size_t cnt, hipStream_t stream) override {
    std::vector<double> &mine = grp->stage[r];
    mine.resize(cnt);
    MSW_HIP(hipMemcpyAsync(mine.data(), dev, cnt * sizeof(double), hipMemcpyDeviceToHost, stream));
    MSW_HIP(hipStreamSynchronize(stream));
    grp->barrier();
    std::vector<double> sum(cnt, 0.0);
    for (int k = 0; k < grp->n; ++k) {
      if (grp->stage[k].size() != cnt) throw HipError("LocalComm: ranks disagree on the message size");
      for (size_t i = 0; i < cnt; ++i) sum[i] += grp->stage[k][i];
    }
    grp->barrier();  // everyone has read the staging buffers
    MSW_HIP(hipMemcpyAsync(dev, sum.data(), cnt * sizeof(double), hipMemcpyHostToDevice, stream));
    MSW_HIP(hipStreamSynchronize(stream));
  }
};

}  // namespace msw
