// rcgpar_hip.hpp -- C++ host shim over the C ABI (include/msweep_core.h) with the signatures
// mSWEEP's rcg_optl() wrapper calls (reference: src/mSWEEP.cpp:176-205, 419-423, 512-516):
//
//   rcgpar::rcg_optl_torch(logl, log_times_observed, alpha0, tol, max_iters, log)          :194
//   rcgpar::rcg_optl_omp  (logl, log_times_observed, alpha0, tol, max_iters, log)          :198
//   rcgpar::em_torch      (logl, log_times_observed, alpha0, tol, max_iters, log, precision) :202
//   rcgpar::mixture_components[_torch](probs, log_times_observed)                          :420,422,513,515
//
// `logl` is any matrix type with get_rows() / get_cols() / operator()(row, col) -- the part of
// seamat::Matrix<double> the reference uses (include/Likelihood.hpp:176,182,258; Sample.hpp:84-85).
// The optimisers return msw::Gamma, which converts to whatever dense matrix type the caller names
// (constructible as (rows, cols, fill), writable through operator()(row, col)): the reference's
//     const seamat::DenseMatrix<double> &ec_probs = rcgpar::rcg_optl_omp(ll_mat, ...);
// compiles as written, with no template argument and no seamat header here (tests/cpp/
// reference_calls_test.cpp holds the five call expressions against stand-in seamat types).
// rcg_optl_omp -- the reference's DEFAULT --algorithm rcgcpu -- runs the same algorithm as
// rcg_optl_torch and is served by the same HIP kernels: this core has no CPU path.
// A non-zero C-ABI status becomes std::runtime_error, which mSWEEP's try/catch blocks
// (src/mSWEEP.cpp:400-406, 506-511) already handle.
//
// The preferred integration keeps the likelihood on the device: msw::DeviceLikelihood owns a
// core handle, is filled once per grouping (set_csr / build / set_dense) and serves the
// 1 + --iters estimation calls without re-uploading.
//
// The UNMODIFIED call sites get the same effect through a one-entry cache: mSWEEP passes the same `ll_mat` object to
// rcg_optl() once for the estimate and once per bootstrap replicate (src/mSWEEP.cpp:402,507 -- `log_likelihoods->
// log_mat()`, const for the whole grouping), so the drop-ins below keep the DeviceLikelihood of the last matrix they saw,
// keyed by (address, rows, columns, a hash of the matrix, device): calls 2 .. 1 + --iters cost their solve, not another
// host copy + upload + device compression of the G x E matrix.  The hash covers EVERY cell up to 2^27 cells (1 GB of
// fp64; round 5: a single cell edited in place is caught -- tests/cpp/reference_calls_test.cpp -- at a few tens of
// milliseconds on the caller's threads); larger matrices, where walking the caller's operator() over every cell would
// cost as much as the solve it saves, fall back to 65 536 sampled cells + the corners, which catches a rewrite with
// the probability of the sample only: callers that DO rewrite such a matrix at the same address between calls (the
// reference does not) call rcgpar::forget_likelihood() or set MSWEEP_SHIM_CACHE=0.  MSWEEP_SHIM_FULL_HASH_CELLS moves
// the limit (0: always sampled).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <ostream>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/msweep_core.h"

namespace msw {

class DenseMatrix {  // rows = groups, row-major: the subset of seamat::DenseMatrix<double> mSWEEP uses
 public:
  DenseMatrix() = default;
  DenseMatrix(size_t rows, size_t cols, double fill = 0.0) : r_(rows), c_(cols), v_(rows * cols, fill) {}
  size_t get_rows() const { return r_; }
  size_t get_cols() const { return c_; }
  double &operator()(size_t r, size_t c) { return v_[r * c_ + c]; }
  const double &operator()(size_t r, size_t c) const { return v_[r * c_ + c]; }
  double *data() { return v_.data(); }
  const double *data() const { return v_.data(); }

 private:
  size_t r_ = 0, c_ = 0;
  std::vector<double> v_;
};

inline void check(msw_handle h, int rc) {
  if (rc != 0) throw std::runtime_error(msw_last_error(h));
}

// The G x E log-responsibility matrix an optimiser returns (rows = groups), convertible to the caller's
// dense matrix type.
class Gamma {
 public:
  Gamma(size_t rows, size_t cols) : r_(rows), c_(cols), v_(rows * cols) {}
  size_t get_rows() const { return r_; }
  size_t get_cols() const { return c_; }
  const double &operator()(size_t r, size_t c) const { return v_[r * c_ + c]; }
  double *data() { return v_.data(); }
  template <class DenseT, class = decltype(DenseT(size_t(1), size_t(1), 0.0)),
            class = decltype(std::declval<DenseT &>()(size_t(0), size_t(0)) = 0.0)>
  operator DenseT() const {
    DenseT out(r_, c_, 0.0);
    for (size_t g = 0; g < r_; ++g)
      for (size_t j = 0; j < c_; ++j) out(g, j) = v_[g * c_ + j];
    return out;
  }

 private:
  size_t r_, c_;
  std::vector<double> v_;
};

// One grouping's likelihood, resident on one GPU.
class DeviceLikelihood {
 public:
  explicit DeviceLikelihood(int device = 0) {
    if (msw_core_create(device, &h_) != 0) throw std::runtime_error(msw_last_error(nullptr));
  }
  ~DeviceLikelihood() { msw_core_destroy(h_); }
  DeviceLikelihood(const DeviceLikelihood &) = delete;
  DeviceLikelihood &operator=(const DeviceLikelihood &) = delete;
  msw_handle handle() const { return h_; }
  // LDS-bank ordering of the cells at upload (msw_core_set_pack_schedule): pays from about the 1 000th iteration on the
  // same likelihood -- set it for bootstrap runs (--iters >= 5), leave it off for one solve.  Before set_* / build.
  void set_pack_schedule(bool on) { check(h_, msw_core_set_pack_schedule(h_, on ? 1 : 0)); }

  template <class MatrixT>
  void set_dense(const MatrixT &logl) {  // rows = groups (seamat layout)
    const size_t G = logl.get_rows(), E = logl.get_cols();
    std::vector<double> buf(G * E);
    for (size_t g = 0; g < G; ++g)
      for (size_t j = 0; j < E; ++j) buf[g * E + j] = logl(g, j);
    check(h_, msw_core_set_dense_logl(h_, buf.data(), G, E, E));
  }
  void set_csr(const std::vector<uint64_t> &rowptr, const std::vector<uint32_t> &grp,
               const std::vector<uint32_t> &cnt, const std::vector<double> &lut, size_t lut_ld, double logzi,
               size_t n_groups) {
    check(h_, msw_core_set_csr(h_, rowptr.data(), grp.data(), cnt.data(), lut.data(), lut_ld, logzi, n_groups,
                               rowptr.size() - 1));
    mask_.assign(n_groups, true);
    logc_.clear();
    built_ = false;
  }
  // ConstructAdaptiveLikelihood (include/Likelihood.hpp:333-380) on the device: the pseudoalignment's
  // equivalence classes (targets per EC, reads per EC), the group indicators and sizes, -q / -e,
  // --min-hits and the zero inflation.  Afterwards log_counts() / groups_considered() hold what the
  // reference's accessors of the same names return (:325-331) and solve() may pass no log counts at all.
  void build(const std::vector<uint64_t> &ec_tptr, const std::vector<uint32_t> &ec_targets,
             const std::vector<uint32_t> &target_group, const std::vector<uint64_t> &group_sizes,
             const std::vector<uint64_t> &ec_counts, double q, double e, size_t min_hits, double zero_inflation) {
    const size_t E = ec_counts.size(), G = group_sizes.size();
    if (ec_tptr.size() != E + 1) throw std::runtime_error("DeviceLikelihood::build: ec_tptr must have n_ecs + 1 entries");
    std::vector<uint8_t> mask(G, 0);
    logc_.assign(E, 0.0);
    size_t kept = 0;
    check(h_, msw_core_build_likelihood(h_, ec_tptr.data(), ec_targets.data(), E, target_group.data(),
                                        target_group.size(), group_sizes.data(), G, ec_counts.data(), q, e,
                                        zero_inflation, min_hits, &kept, mask.data(), logc_.data()));
    mask_.assign(mask.begin(), mask.end());
    built_ = true;
  }
  // The same from the Themisto plaintext files themselves, ON THE DEVICE: the reader as kernels
  // (msw_alignment_read_device: text -> equivalence classes in device memory) and the build from its resident arrays
  // (msw_core_build_likelihood_aln) -- Alignment::read + collapse + ConstructAdaptiveLikelihood
  // (src/mSWEEP.cpp:324-346) without the pseudoalignment crossing PCIe twice.  Afterwards ec_counts() holds the reads
  // per class (totals, bootstrap weights) and n_reads() / n_aligned() what Alignment::n_reads() and the sum of the
  // counts are.  Text the kernels do not judge goes through the host parser: the reference's messages.
  void build_from_files(const std::vector<std::string> &paths, bool union_mode, const std::vector<uint32_t> &target_group,
                        const std::vector<uint64_t> &group_sizes, double q, double e, size_t min_hits, double zero_inflation) {
    std::vector<const char *> c;
    for (const auto &p : paths) c.push_back(p.c_str());
    msw_alignment_t aln = nullptr;
    if (msw_alignment_read_device(h_, c.data(), c.size(), target_group.size(), union_mode ? MSW_MERGE_UNION : MSW_MERGE_INTERSECTION,
                                  &aln))
      throw std::runtime_error(msw_alignment_last_error());
    struct Guard {
      msw_alignment_t a;
      ~Guard() { msw_alignment_destroy(a); }
    } guard{aln};
    size_t E = 0, hits = 0;
    msw_alignment_shape(aln, &E, &n_reads_, &hits, &n_aligned_);
    if (E == 0) throw std::runtime_error("no read aligned against the reference");
    ec_counts_.assign(E, 0);
    if (msw_alignment_export(aln, nullptr, nullptr, ec_counts_.data(), nullptr, nullptr))
      throw std::runtime_error("msw_alignment_export failed");
    const size_t G = group_sizes.size();
    std::vector<uint8_t> mask(G, 0);
    logc_.assign(E, 0.0);
    size_t kept = 0;
    check(h_, msw_core_build_likelihood_aln(h_, aln, target_group.data(), target_group.size(), group_sizes.data(), G, q, e,
                                            zero_inflation, min_hits, &kept, mask.data(), logc_.data()));
    mask_.assign(mask.begin(), mask.end());
    built_ = true;
  }
  const std::vector<uint64_t> &ec_counts() const { return ec_counts_; }   // reads per class (build_from_files)
  size_t n_reads() const { return n_reads_; }
  size_t n_aligned() const { return n_aligned_; }
  const std::vector<double> &log_counts() const { return logc_; }          // Likelihood::log_counts()
  const std::vector<bool> &groups_considered() const { return mask_; }     // Likelihood::groups_considered()
  bool built_on_device() const { return built_; }
  size_t n_groups() const {  // groups of the resident likelihood (after --min-hits masking)
    size_t g = 0;
    check(h_, msw_core_shape(h_, &g, nullptr, nullptr));
    return g;
  }
  size_t n_ecs() const {
    size_t e = 0;
    check(h_, msw_core_shape(h_, nullptr, &e, nullptr));
    return e;
  }

 private:
  msw_handle h_ = nullptr;
  std::vector<double> logc_;
  std::vector<uint64_t> ec_counts_;
  size_t n_reads_ = 0, n_aligned_ = 0;
  std::vector<bool> mask_;
  bool built_ = false;
};

struct Estimate {
  std::vector<double> theta;
  size_t iters = 0;
  double bound = 0.0;
};

inline Estimate solve(DeviceLikelihood &lik, const std::vector<double> &log_times_observed,
                      const std::vector<double> &alpha0, double tol, size_t max_iters, int algo, int prec,
                      std::ostream *log) {
  Estimate r;
  r.theta.resize(alpha0.size());
  // no log counts given and the likelihood was built on the device: they are still there (no upload)
  const double *logc = log_times_observed.empty() && lik.built_on_device() ? nullptr : log_times_observed.data();
  check(lik.handle(), msw_core_solve(lik.handle(), logc, alpha0.data(), tol, max_iters, algo, prec, r.theta.data(),
                                     &r.iters, &r.bound));
  if (log && log->good()) {  // rcgpar logs every 5th iteration
    const size_t n = r.iters < 4096 ? r.iters : 4096;
    std::vector<double> b(n), g(n);
    size_t got = 0;
    check(lik.handle(), msw_core_trace(lik.handle(), n, b.data(), g.data(), nullptr, nullptr, nullptr, &got));
    for (size_t k = 0; k < got; k += 5) *log << "  iter: " << k << ", bound: " << b[k] << ", |g|: " << g[k] << '\n';
  }
  return r;
}

inline Gamma gamma_of(DeviceLikelihood &lik, size_t n_groups, size_t n_ecs) {
  Gamma out(n_groups, n_ecs);
  check(lik.handle(), msw_core_gamma(lik.handle(), out.data(), n_ecs));
  return out;
}

// ---- the likelihood of the unmodified call sites, kept across calls (file header) -------------------------------
namespace detail {
struct ShimCache {
  const void *addr = nullptr;
  size_t rows = 0, cols = 0;
  uint64_t hash = 0;
  int device = -1;
  DeviceLikelihood *lik = nullptr;  // owned; deliberately NOT destroyed at process exit (the HIP runtime may be gone
                                    // by then: the driver reclaims device memory anyway) -- forget_likelihood() frees it
  size_t hits = 0, uploads = 0;
};
inline ShimCache &shim_cache() {
  static thread_local ShimCache c;  // the reference calls from its single main thread (src/mSWEEP.cpp:402,507)
  return c;
}
// FNV-1a over the bit patterns of kSample cells at fixed pseudo-random positions plus the four corners
template <class MatrixT>
uint64_t sample_hash(const MatrixT &logl) {
  const size_t G = logl.get_rows(), E = logl.get_cols();
  uint64_t h = 1469598103934665603ull;
  auto eat = [&](size_t g, size_t j) {
    const double v = logl(g, j);
    uint64_t b;
    static_assert(sizeof b == sizeof v, "double is 64 bits");
    __builtin_memcpy(&b, &v, sizeof b);
    for (int k = 0; k < 8; ++k) h = (h ^ ((b >> (8 * k)) & 0xffu)) * 1099511628211ull;
  };
  if (G == 0 || E == 0) return h;
  constexpr size_t kSample = 65536;
  const unsigned __int128 n = (unsigned __int128)G * E;
  uint64_t x = 0x9e3779b97f4a7c15ull;
  for (size_t i = 0; i < kSample; ++i) {
    x = x * 6364136223846793005ull + 1442695040888963407ull;
    const size_t pos = (size_t)(((unsigned __int128)x * n) >> 64);
    eat(pos / E, pos % E);
  }
  eat(0, 0), eat(0, E - 1), eat(G - 1, 0), eat(G - 1, E - 1);
  return h;
}
// every cell: the row-major index space cut into one range per thread (up to 8), a 64-bit multiply-xor chain over the
// cells' bit patterns per range, the ranges' digests chained in order -- one multiplication per cell, memory-bound
template <class MatrixT>
uint64_t full_hash(const MatrixT &logl) {
  const size_t G = logl.get_rows(), E = logl.get_cols(), n = G * E;
  const size_t T = std::max<size_t>(1, std::min<size_t>({(size_t)8, (size_t)std::thread::hardware_concurrency(), n >> 20}));
  std::vector<uint64_t> part(T, 0);
  auto work = [&](size_t t) {
    uint64_t h = 0x9e3779b97f4a7c15ull ^ t;
    const size_t i0 = n * t / T, i1 = n * (t + 1) / T;
    size_t g = i0 / E, j = i0 % E;
    for (size_t i = i0; i < i1; ++i) {
      const double v = logl(g, j);
      uint64_t b;
      __builtin_memcpy(&b, &v, sizeof b);
      h = (h ^ b) * 0xff51afd7ed558ccdull;
      h ^= h >> 29;
      if (++j == E) j = 0, ++g;
    }
    part[t] = h;
  };
  std::vector<std::thread> pool;
  for (size_t t = 1; t < T; ++t) pool.emplace_back(work, t);
  work(0);
  for (auto &th : pool) th.join();
  uint64_t h = 1469598103934665603ull ^ (uint64_t)n;
  for (uint64_t v : part) h = (h ^ v) * 1099511628211ull;
  return h | 1ull;  // (never the sampled hash of the same matrix by accident of a zero)
}
inline size_t full_hash_limit() {
  if (const char *e = std::getenv("MSWEEP_SHIM_FULL_HASH_CELLS")) return (size_t)std::strtoull(e, nullptr, 10);
  return (size_t)1 << 27;
}
template <class MatrixT>
DeviceLikelihood &resident(const MatrixT &logl, int device) {
  ShimCache &c = shim_cache();
  const char *sw = std::getenv("MSWEEP_SHIM_CACHE");
  const bool use = !(sw && sw[0] == '0');
  const bool full = logl.get_rows() * logl.get_cols() <= full_hash_limit();
  const uint64_t h = use ? (full ? full_hash(logl) : sample_hash(logl)) : 0;
  if (use && c.lik && c.addr == static_cast<const void *>(&logl) && c.rows == logl.get_rows() &&
      c.cols == logl.get_cols() && c.hash == h && c.device == device) {
    ++c.hits;
    return *c.lik;
  }
  delete c.lik;
  c.lik = nullptr;
  c.lik = new DeviceLikelihood(device);
  c.lik->set_pack_schedule(false);  // the shortest way to the FIRST estimate; a bootstrap pays 5 % per iteration for it
  c.lik->set_dense(logl);
  c.addr = &logl, c.rows = logl.get_rows(), c.cols = logl.get_cols(), c.hash = h, c.device = device;
  ++c.uploads;
  return *c.lik;
}
}  // namespace detail

}  // namespace msw

namespace rcgpar {

// Drop-in for the call at src/mSWEEP.cpp:194 (the dense `ll_mat` is uploaded by the first call that sees it and stays
// resident for the calls that follow: msw::detail::resident).
template <class MatrixT>
msw::Gamma rcg_optl_torch(const MatrixT &logl, const std::vector<double> &log_times_observed,
                          const std::vector<double> &alpha0, const double &tol, size_t max_iters, std::ostream &log,
                          int device = 0) {
  msw::DeviceLikelihood &lik = msw::detail::resident(logl, device);
  msw::solve(lik, log_times_observed, alpha0, tol, max_iters, MSW_ALGO_RCG, MSW_PREC_DOUBLE, &log);
  return msw::gamma_of(lik, logl.get_rows(), logl.get_cols());
}

// Frees the likelihood the drop-ins keep resident between calls (and makes the next call upload again).
inline void forget_likelihood() {
  msw::detail::ShimCache &c = msw::detail::shim_cache();
  delete c.lik;
  c.lik = nullptr;
  c.addr = nullptr;
}
// (uploads, cache hits) of the drop-ins on this thread so far: diagnostics / tests
inline std::pair<size_t, size_t> likelihood_cache_stats() {
  const msw::detail::ShimCache &c = msw::detail::shim_cache();
  return {c.uploads, c.hits};
}

// Drop-in for the call at src/mSWEEP.cpp:198 (--algorithm rcgcpu, the reference's default): the same
// RCG algorithm, run by the HIP kernels.
template <class MatrixT>
msw::Gamma rcg_optl_omp(const MatrixT &logl, const std::vector<double> &log_times_observed,
                        const std::vector<double> &alpha0, const double &tol, size_t max_iters, std::ostream &log,
                        int device = 0) {
  return rcg_optl_torch(logl, log_times_observed, alpha0, tol, max_iters, log, device);
}

// Drop-in for the call at src/mSWEEP.cpp:202.
template <class MatrixT>
msw::Gamma em_torch(const MatrixT &logl, const std::vector<double> &log_times_observed,
                    const std::vector<double> &alpha0, const double &tol, size_t max_iters, std::ostream &log,
                    std::string precision, int device = 0) {
  if (precision != "double" && precision != "float") throw std::runtime_error("em_torch: unknown precision " + precision);
  msw::DeviceLikelihood &lik = msw::detail::resident(logl, device);
  msw::solve(lik, log_times_observed, alpha0, tol, max_iters, MSW_ALGO_EM,
             precision == "float" ? MSW_PREC_FLOAT : MSW_PREC_DOUBLE, &log);
  return msw::gamma_of(lik, logl.get_rows(), logl.get_cols());
}

// rcgpar::mixture_components[_torch] (src/mSWEEP.cpp:420,422,513,515): theta_g = sum_j exp(gamma_gj + logc_j) /
// sum_j c_j.  Host loop over the returned matrix, as the reference does; with msw::solve() the same vector
// is already available as Estimate::theta without materialising gamma.
template <class MatrixT>
std::vector<double> mixture_components_torch(const MatrixT &probs, const std::vector<double> &log_times_observed) {
  const size_t G = probs.get_rows(), E = probs.get_cols();
  double total = 0.0;
  for (size_t j = 0; j < E; ++j) total += std::exp(log_times_observed[j]);
  std::vector<double> theta(G, 0.0);
  for (size_t g = 0; g < G; ++g) {
    double acc = 0.0;
    for (size_t j = 0; j < E; ++j) acc += std::exp(probs(g, j) + log_times_observed[j]);
    theta[g] = acc / total;
  }
  return theta;
}
template <class MatrixT>
std::vector<double> mixture_components(const MatrixT &probs, const std::vector<double> &log_times_observed) {
  return mixture_components_torch(probs, log_times_observed);
}

}  // namespace rcgpar
