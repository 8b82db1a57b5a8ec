// Exercises the rcgpar-shaped C++ shim the way src/mSWEEP.cpp:176-205,419-423 calls rcgpar.
// Reads a dense problem from stdin (G E, then L row-major rows = groups, logc, alpha0) and prints
// iters, theta (from mixture_components_torch on the returned gamma) and column-sum checks.
#include <cmath>
#include <cstdio>
#include <iostream>
#include <algorithm>
#include <sstream>

#include "../../msweep_amd/cpp/rcgpar_hip.hpp"

int main(int argc, char **argv) {
  size_t G, E;
  if (!(std::cin >> G >> E)) return 2;
  msw::DenseMatrix L(G, E);
  for (size_t g = 0; g < G; ++g)
    for (size_t j = 0; j < E; ++j) std::cin >> L(g, j);
  std::vector<double> logc(E), alpha(G);
  for (auto &x : logc) std::cin >> x;
  for (auto &x : alpha) std::cin >> x;
  const bool em = argc > 1 && std::string(argv[1]) == "em";
  std::ostringstream log;
  try {
    msw::DenseMatrix gamma = em ? rcgpar::em_torch(L, logc, alpha, 1e-6, 5000, log, "double")
                                : rcgpar::rcg_optl_torch(L, logc, alpha, 1e-6, 5000, log);
    std::vector<double> theta = rcgpar::mixture_components_torch(gamma, logc);
    double worst = 0.0;
    for (size_t j = 0; j < E; ++j) {
      double s = 0.0;
      for (size_t g = 0; g < G; ++g) s += std::exp(gamma(g, j));
      worst = std::max(worst, std::fabs(s - 1.0));
    }
    const std::string logged = log.str();
    std::printf("colsum_err %.3e\nlog_lines %zu\ntheta", worst, (size_t)std::count(logged.begin(), logged.end(), '\n'));
    for (double t : theta) std::printf(" %.17g", t);
    std::printf("\n");
    // error path: a bad call surfaces as std::runtime_error
    try {
      std::vector<double> short_alpha(G + 3, 1.0);
      msw::DeviceLikelihood lik(0);
      msw::solve(lik, logc, short_alpha, 1e-6, 10, MSW_ALGO_RCG, MSW_PREC_DOUBLE, nullptr);
      std::printf("error_path missing\n");
    } catch (const std::runtime_error &ex) {
      std::printf("error_path ok: %s\n", ex.what());
    }
  } catch (const std::exception &ex) {
    std::printf("exception %s\n", ex.what());
    return 1;
  }
  return 0;
}
