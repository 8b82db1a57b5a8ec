// The five rcgpar call expressions of the reference, as written there (src/mSWEEP.cpp:194,198,202,420,422),
// compiled against msweep_amd/cpp/rcgpar_hip.hpp and stand-in seamat types that expose what mSWEEP uses of
// seamat::Matrix<double> / seamat::DenseMatrix<double> (virtual operator()(row, col), get_rows(), get_cols(),
// the (rows, cols, fill) constructor; include/Likelihood.hpp:98,176,182,252,258, include/Sample.hpp:84-85).
// Test scaffolding for the shim's signatures only -- nothing of the reference is built from these.
// Reads a dense problem from stdin (G E, L rows = groups, logc, alpha0); argv[1] = algorithm.
#include <cstdio>
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
  size_t G, E;
  if (!(std::cin >> G >> E)) return 2;
  seamat::DenseMatrix<double> L(G, E, 0.0);
  for (size_t g = 0; g < G; ++g)
    for (size_t j = 0; j < E; ++j) std::cin >> L(g, j);
  std::vector<double> logc(E), alpha(G);
  for (auto &x : logc) std::cin >> x;
  for (auto &x : alpha) std::cin >> x;
  Args args;
  args.algorithm = argc > 1 ? argv[1] : "rcgcpu";
  Log log;
  try {
    const seamat::Matrix<double> &ll_mat = L;
    seamat::DenseMatrix<double> probs = rcg_optl(args, ll_mat, logc, alpha, log);
    const seamat::Matrix<double> &get_probs = probs;
    std::vector<double> theta;
    if (args.algorithm == "rcgcpu") {  // src/mSWEEP.cpp:419-423
      theta = rcgpar::mixture_components(get_probs, logc);
    } else {
      theta = rcgpar::mixture_components_torch(get_probs, logc);
    }
    std::printf("theta");
    for (double t : theta) std::printf(" %.17g", t);
    std::printf("\n");
  } catch (std::exception &e) {
    std::printf("exception %s\n", e.what());
    return 1;
  }
  return 0;
}
