// The five rcgpar call expressions of the reference, as written there (src/mSWEEP.cpp:194,198,202,420,422),
// compiled against msweep_amd/cpp/rcgpar_hip.hpp and stand-in seamat types that expose what mSWEEP uses of
// seamat::Matrix<double> / seamat::DenseMatrix<double> (virtual operator()(row, col), get_rows(), get_cols(),
// the (rows, cols, fill) constructor; include/Likelihood.hpp:98,176,182,252,258, include/Sample.hpp:84-85).
// Test scaffolding for the shim's signatures only -- nothing of the reference is built from these.
// Reads a dense problem from stdin (G E, L rows = groups, logc, alpha0), or -- argv[2] = a path -- from a raw binary
// file (uint64 G, E; then L, logc, alpha0 as doubles); argv[1] = algorithm; argv[3] = B: the bootstrap loop of
// src/mSWEEP.cpp:496-518 -- B more calls of the SAME expressions on the SAME `ll_mat` with other log counts (rotated
// here; the real resampling is not this test's subject), timed per call, then one more call after the matrix has been
// rewritten IN PLACE (the shim's resident copy must not be served for it).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "../../msweep_amd/cpp/rcgpar_hip.hpp"

namespace seamat {
template <typename T>
class Matrix {
 public:
  virtual ~Matrix() = default;
  virtual T &operator()(size_t row, size_t col) = 0;
  virtual const T &operator()(size_t row, size_t col) const = 0;
  size_t get_rows() const { return rows; }
  size_t get_cols() const { return cols; }

 protected:
  size_t rows = 0, cols = 0;
};
template <typename T>
class DenseMatrix : public Matrix<T> {
 public:
  DenseMatrix() = default;
  DenseMatrix(size_t r, size_t c, T fill) : v(r * c, fill) {
    this->rows = r;
    this->cols = c;
  }
  T &operator()(size_t row, size_t col) override { return v[row * this->cols + col]; }
  const T &operator()(size_t row, size_t col) const override { return v[row * this->cols + col]; }

 private:
  std::vector<T> v;
};
}  // namespace seamat

// what the reference's rcg_optl() reads from cxxargs::Arguments (src/mSWEEP.cpp:192-203)
struct Args {
  std::string algorithm, emprecision = "double";
  double tol = 1e-6;
  size_t max_iters = 5000;
  bool verbose = false;
};
struct Log {
  std::ostream &stream() { return std::cerr; }
};

// src/mSWEEP.cpp:176-205 with `args.value<T>("x")` spelled as members; the rcgpar calls are verbatim
seamat::DenseMatrix<double> rcg_optl(const Args &args, const seamat::Matrix<double> &ll_mat,
                                     const std::vector<double> &log_ec_counts, const std::vector<double> &prior_counts,
                                     Log &log) {
  std::ofstream of;
  if (args.algorithm == "rcggpu") {
    const seamat::DenseMatrix<double> &ec_probs = rcgpar::rcg_optl_torch(ll_mat, log_ec_counts, prior_counts, args.tol, args.max_iters, (args.verbose ? log.stream() : of));
    return ec_probs;
  } else if (args.algorithm == "rcgcpu") {
    const seamat::DenseMatrix<double> &ec_probs = rcgpar::rcg_optl_omp(ll_mat, log_ec_counts, prior_counts, args.tol, args.max_iters, (args.verbose ? log.stream() : of));
    return ec_probs;
  } else {
    const seamat::DenseMatrix<double> &ec_probs = rcgpar::em_torch(ll_mat, log_ec_counts, prior_counts, args.tol, args.max_iters, (args.verbose ? log.stream() : of), args.emprecision);
    return ec_probs;
  }
}

int main(int argc, char **argv) {
  size_t G = 0, E = 0;
  seamat::DenseMatrix<double> L;
  std::vector<double> logc, alpha;
  if (argc > 2 && argv[2][0]) {
    std::ifstream f(argv[2], std::ios::binary);
    uint64_t dims[2];
    if (!f.read(reinterpret_cast<char *>(dims), sizeof dims)) return 2;
    G = dims[0], E = dims[1];
    L = seamat::DenseMatrix<double>(G, E, 0.0);
    std::vector<double> row(E);
    for (size_t g = 0; g < G; ++g) {
      f.read(reinterpret_cast<char *>(row.data()), E * sizeof(double));
      for (size_t j = 0; j < E; ++j) L(g, j) = row[j];
    }
    logc.resize(E), alpha.resize(G);
    f.read(reinterpret_cast<char *>(logc.data()), E * sizeof(double));
    if (!f.read(reinterpret_cast<char *>(alpha.data()), G * sizeof(double))) return 2;
  } else {
    if (!(std::cin >> G >> E)) return 2;
    L = seamat::DenseMatrix<double>(G, E, 0.0);
    for (size_t g = 0; g < G; ++g)
      for (size_t j = 0; j < E; ++j) std::cin >> L(g, j);
    logc.resize(E), alpha.resize(G);
    for (auto &x : logc) std::cin >> x;
    for (auto &x : alpha) std::cin >> x;
  }
  const size_t B = argc > 3 ? (size_t)std::atoll(argv[3]) : 0;
  Args args;
  args.algorithm = argc > 1 ? argv[1] : "rcgcpu";
  Log log;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
  };
  auto estimate = [&](const std::vector<double> &lc, double *wall_ms) {
    const seamat::Matrix<double> &ll_mat = L;
    const auto t0 = now();
    seamat::DenseMatrix<double> probs = rcg_optl(args, ll_mat, lc, alpha, log);
    if (wall_ms) *wall_ms = ms(t0, now());
    const seamat::Matrix<double> &get_probs = probs;
    if (args.algorithm == "rcgcpu") return rcgpar::mixture_components(get_probs, lc);  // src/mSWEEP.cpp:419-423
    return rcgpar::mixture_components_torch(get_probs, lc);
  };
  auto print = [](const char *tag, const std::vector<double> &theta) {
    std::printf("%s", tag);
    for (double t : theta) std::printf(" %.17g", t);
    std::printf("\n");
  };
  try {
    double first_ms = 0.0;
    print("theta", estimate(logc, &first_ms));
    if (B) {
      // src/mSWEEP.cpp:496-518: the same `log_likelihoods->log_mat()`, other log counts, B times
      std::vector<double> lc(E), later(B), solve_ms(B);
      std::vector<double> th;
      for (size_t b = 0; b < B; ++b) {
        for (size_t j = 0; j < E; ++j) lc[j] = logc[(j + b + 1) % E];
        th = estimate(lc, &later[b]);
        msw_timing t;
        msw_core_last_timing(msw::detail::shim_cache().lik->handle(), &t);
        solve_ms[b] = t.solve_ms;
        if (b == 0) print("theta_b1", th);
      }
      const auto st = rcgpar::likelihood_cache_stats();
      double lsum = 0.0, ssum = 0.0;
      for (size_t b = 0; b < B; ++b) lsum += later[b], ssum += solve_ms[b];
      std::printf("uploads %zu\nhits %zu\nfirst_ms %.3f\nlater_ms %.3f\nlater_solve_ms %.3f\n", st.first, st.second, first_ms,
                  lsum / B, ssum / B);
      // the matrix rewritten in place, same address, same shape: must be noticed (sampled hash) and uploaded again
      for (size_t g = 0; g < G; ++g)
        for (size_t j = 0; j < E; ++j)
          if (L(g, j) > -4.0) L(g, j) -= 0.25;
      print("theta_changed", estimate(logc, nullptr));
      std::printf("uploads_after_change %zu\n", rcgpar::likelihood_cache_stats().first);
      // ONE cell edited in place (a position no sample of 65 536 cells is likely to hold): the full hash sees it
      L(G / 2, (E * 2) / 3 + 1) -= 1e-3;
      estimate(logc, nullptr);
      std::printf("uploads_after_one_cell %zu\n", rcgpar::likelihood_cache_stats().first);
      estimate(logc, nullptr);  // and nothing changed since: served from the resident likelihood again
      std::printf("uploads_after_no_change %zu\n", rcgpar::likelihood_cache_stats().first);
      rcgpar::forget_likelihood();
    }
  } catch (std::exception &e) {
    std::printf("exception %s\n", e.what());
    return 1;
  }
  return 0;
}
