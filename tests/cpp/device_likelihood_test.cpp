// msw::DeviceLikelihood::build() / log_counts() / groups_considered() + msw::solve() with the log counts left
// on the device: the C++ face of ConstructAdaptiveLikelihood + rcg_optl (include/Likelihood.hpp:321-380,
// src/mSWEEP.cpp:346,402).  stdin: E H T G min_hits; ec_tptr[E+1]; ec_targets[H]; target_group[T];
// group_sizes[G]; ec_counts[E].
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>

#include "../../msweep_amd/cpp/rcgpar_hip.hpp"

// `files <union:0|1> <min_hits> path...` with stdin T G; target_group[T]; group_sizes[G]: the same through
// DeviceLikelihood::build_from_files (the reader on the device + the build from its resident classes)
static int from_files(int argc, char **argv) {
  const bool union_mode = std::atoi(argv[2]) != 0;
  const size_t min_hits = (size_t)std::atoll(argv[3]);
  std::vector<std::string> paths(argv + 4, argv + argc);
  size_t T, G;
  if (!(std::cin >> T >> G)) return 2;
  std::vector<uint32_t> tg(T);
  std::vector<uint64_t> sizes(G);
  for (auto &x : tg) std::cin >> x;
  for (auto &x : sizes) std::cin >> x;
  try {
    msw::DeviceLikelihood lik(0);
    lik.build_from_files(paths, union_mode, tg, sizes, 0.65, 0.01, min_hits, 0.01);
    std::printf("n_groups %zu\nn_ecs %zu\nn_reads %zu\nn_aligned %zu\nmask", lik.n_groups(), lik.n_ecs(), lik.n_reads(), lik.n_aligned());
    for (bool b : lik.groups_considered()) std::printf(" %d", b ? 1 : 0);
    std::printf("\ncounts");
    for (uint64_t c : lik.ec_counts()) std::printf(" %llu", (unsigned long long)c);
    std::printf("\nlogc");
    for (double x : lik.log_counts()) std::printf(" %.17g", x);
    std::vector<double> alpha(lik.n_groups(), 1.0);
    msw::Estimate est = msw::solve(lik, {}, alpha, 1e-6, 5000, MSW_ALGO_RCG, MSW_PREC_DOUBLE, nullptr);
    std::printf("\niters %zu\ntheta", est.iters);
    for (double t : est.theta) std::printf(" %.17g", t);
    std::printf("\n");
  } catch (const std::exception &ex) {
    std::printf("exception %s\n", ex.what());
    return 1;
  }
  return 0;
}

int main(int argc, char **argv) {
  if (argc >= 5 && std::string(argv[1]) == "files") return from_files(argc, argv);
  size_t E, H, T, G, min_hits;
  if (!(std::cin >> E >> H >> T >> G >> min_hits)) return 2;
  std::vector<uint64_t> tptr(E + 1), sizes(G), counts(E);
  std::vector<uint32_t> targets(H), tg(T);
  for (auto &x : tptr) std::cin >> x;
  for (auto &x : targets) std::cin >> x;
  for (auto &x : tg) std::cin >> x;
  for (auto &x : sizes) std::cin >> x;
  for (auto &x : counts) std::cin >> x;
  try {
    msw::DeviceLikelihood lik(0);
    lik.build(tptr, targets, tg, sizes, counts, 0.65, 0.01, min_hits, 0.01);
    std::printf("n_groups %zu\nn_ecs %zu\nmask", lik.n_groups(), lik.n_ecs());
    for (bool b : lik.groups_considered()) std::printf(" %d", b ? 1 : 0);
    std::printf("\nlogc");
    for (double x : lik.log_counts()) std::printf(" %.17g", x);
    std::vector<double> alpha(lik.n_groups(), 1.0);
    msw::Estimate est = msw::solve(lik, {}, alpha, 1e-6, 5000, MSW_ALGO_RCG, MSW_PREC_DOUBLE, nullptr);
    std::printf("\niters %zu\ntheta", est.iters);
    for (double t : est.theta) std::printf(" %.17g", t);
    std::printf("\n");
  } catch (const std::exception &ex) {
    std::printf("exception %s\n", ex.what());
    return 1;
  }
  return 0;
}
