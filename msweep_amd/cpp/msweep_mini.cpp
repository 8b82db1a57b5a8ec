// msweep_mini.cpp -- the estimation path of mSWEEP's main() (src/mSWEEP.cpp:258-551) as a native
// host program over the C ABI: group indicators (-i, include/Reference.hpp / Grouping.hpp), Themisto
// plaintext pseudoalignments (native reader: msw_alignment_read), likelihood built and kept on the
// GPU (msw_core_build_likelihood), RCG / EM abundances (--algorithm rcggpu|emgpu), bootstrap
// (--iters / --seed / --bootstrap-count, src/mSWEEP.cpp:496-518) and abundances.txt in the format of
// PlainSample / BootstrapSample::write_abundances[2] (src/PlainSample.cpp:32-71,
// src/BootstrapSample.cpp:75-130).  Same flags and messages as the reference for what it covers;
// the Python mirror `python -m msweep_amd` carries the remaining outputs (probs, likelihood files,
// RATE) and is held byte-for-byte against this program in tests/test_gpu_cli_toy.py.
//
//   g++ -std=c++17 -O2 -o msweep_mini msweep_mini.cpp -L.. -lmsweep_core -Wl,-rpath,..
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/msweep_core.h"

namespace {

const char *kVersion = "msweep-amd-0.1.0";

struct Args {
  std::vector<std::string> themisto;
  std::string mode = "intersection", indicators, prefix, algorithm = "rcgcpu", emprecision = "double", alphas;
  size_t iters = 0, seed = 26012023, bootstrap_count = 0, min_hits = 0, max_iters = 5000;
  double q = 0.65, e = 0.01, zero_inflation = 0.01, tol = 1e-6;
  int gpu = 0;
  bool verbose = false;
};

std::vector<std::string> split(const std::string &s, char d) {
  std::vector<std::string> out;
  std::stringstream ss(s);
  std::string p;
  while (std::getline(ss, p, d)) out.push_back(p);
  return out;
}

Args parse(int argc, char **argv) {
  Args a;
  std::string t1, t2;
  for (int i = 1; i < argc; ++i) {
    const std::string k = argv[i];
    auto val = [&]() -> std::string {
      if (i + 1 >= argc) throw std::runtime_error("missing value for " + k);
      return argv[++i];
    };
    if (k == "--themisto") a.themisto = split(val(), ',');
    else if (k == "--themisto-1") t1 = val();
    else if (k == "--themisto-2") t2 = val();
    else if (k == "--themisto-mode") a.mode = val();
    else if (k == "-i") a.indicators = val();
    else if (k == "-o") a.prefix = val();
    else if (k == "--iters") a.iters = std::stoul(val());
    else if (k == "--seed") a.seed = std::stoul(val());
    else if (k == "--bootstrap-count") a.bootstrap_count = std::stoul(val());
    else if (k == "--min-hits") a.min_hits = std::stoul(val());
    else if (k == "--max-iters") a.max_iters = std::stoul(val());
    else if (k == "-q") a.q = std::stod(val());
    else if (k == "-e") a.e = std::stod(val());
    else if (k == "--zero-inflation") a.zero_inflation = std::stod(val());
    else if (k == "--tol") a.tol = std::stod(val());
    else if (k == "--alphas") a.alphas = val();
    else if (k == "--algorithm") a.algorithm = val();
    else if (k == "--emprecision") a.emprecision = val();
    else if (k == "--gpu-index") a.gpu = std::stoi(val());
    else if (k == "-t") (void)val();  // host threads: nothing to set here
    else if (k == "--verbose") a.verbose = true;
    else throw std::runtime_error("unknown argument " + k);
  }
  if (a.themisto.empty()) {
    if (!t1.empty()) a.themisto.push_back(t1);
    if (!t2.empty()) a.themisto.push_back(t2);
  }
  if (a.indicators.empty()) throw std::runtime_error("-i <group indicators> is required");
  return a;
}

// ConstructAdaptiveReference / AdaptiveGrouping::add_sequence (src/Reference.cpp:31-56,
// include/Grouping.hpp:75-80): one line per reference sequence, first tab-separated column = group
// name, group ids in order of first appearance
struct Grouping {
  std::vector<std::string> names;
  std::vector<uint64_t> sizes;
  std::vector<uint32_t> indicators;
};
Grouping read_grouping(const std::string &path) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error("cannot open " + path);
  Grouping g;
  std::unordered_map<std::string, uint32_t> ids;
  std::string line;
  while (std::getline(in, line)) {
    const std::string name = line.substr(0, line.find('\t'));
    auto it = ids.find(name);
    if (it == ids.end()) {
      it = ids.emplace(name, (uint32_t)g.names.size()).first;
      g.names.push_back(name);
      g.sizes.push_back(0);
    }
    ++g.sizes[it->second];
    g.indicators.push_back(it->second);
  }
  if (g.indicators.empty()) throw std::runtime_error("The grouping contains 0 reference sequences");
  return g;
}

void check(msw_handle h, int rc) {
  if (rc != 0) throw std::runtime_error(msw_last_error(h));
}

}  // namespace

int main(int argc, char **argv) {
  Args a;
  try {
    a = parse(argc, argv);
  } catch (const std::exception &ex) {
    std::cerr << "Parsing arguments failed:\n  " << ex.what() << "\nexiting\n";
    return 1;
  }
  Grouping grouping;
  std::vector<uint64_t> ec_tptr, ec_counts;
  std::vector<uint32_t> ec_targets;
  size_t n_ecs = 0, n_reads = 0, n_hits = 0, n_aligned = 0;
  try {
    grouping = read_grouping(a.indicators);
    if (a.themisto.empty()) throw std::runtime_error("no pseudoalignment files given");
    if (a.mode != "intersection" && a.mode != "union")
      throw std::runtime_error("Unrecognized option `" + a.mode + "` for --themisto-mode");
    std::vector<const char *> paths;
    for (auto &p : a.themisto) paths.push_back(p.c_str());
    msw_alignment_t aln = nullptr;
    if (msw_alignment_read(paths.data(), paths.size(), grouping.indicators.size(),
                           a.mode == "union" ? MSW_MERGE_UNION : MSW_MERGE_INTERSECTION, &aln))
      throw std::runtime_error(msw_alignment_last_error());
    msw_alignment_shape(aln, &n_ecs, &n_reads, &n_hits, &n_aligned);
    ec_tptr.resize(n_ecs + 1);
    ec_counts.resize(n_ecs);
    ec_targets.resize(n_hits);
    msw_alignment_export(aln, ec_tptr.data(), ec_targets.data(), ec_counts.data(), nullptr, nullptr);
    msw_alignment_destroy(aln);
  } catch (const std::exception &ex) {
    std::cerr << "Reading the pseudoalignments failed:\n  " << ex.what() << "\nexiting\n";
    return 1;
  }
  if (a.algorithm == "rcgcpu" && a.verbose)  // the reference's default: the same RCG algorithm on the host; no CPU path here
    std::cerr << "note: --algorithm rcgcpu is served by the GPU RCG kernels (same algorithm as rcggpu)\n";
  const int algo = (a.algorithm == "rcggpu" || a.algorithm == "rcgcpu") ? MSW_ALGO_RCG : MSW_ALGO_EM;  // else em (src/mSWEEP.cpp:200)
  const int prec = a.emprecision == "float" ? MSW_PREC_FLOAT : MSW_PREC_DOUBLE;
  if (algo == MSW_ALGO_EM && prec == MSW_PREC_FLOAT)
    std::cerr << "note: --emprecision float is computed in double here (no G x E matrix exists whose footprint "
                 "float would halve); results are those of --emprecision double\n";
  const size_t G = grouping.names.size();
  msw_handle h = nullptr;
  size_t n_kept = 0;
  std::vector<uint8_t> mask(G, 1);
  try {
    if (msw_core_create(a.gpu, &h) != 0) throw std::runtime_error(msw_last_error(nullptr));
    if (n_ecs == 0) throw std::runtime_error("no read aligned against the reference");
    // ordering the cells for the LDS banks pays from about the 1 000th iteration on: bootstrap runs
    check(h, msw_core_set_pack_schedule(h, a.iters >= 5 ? 1 : 0));
    check(h, msw_core_build_likelihood(h, ec_tptr.data(), ec_targets.data(), n_ecs, grouping.indicators.data(),
                                       grouping.indicators.size(), grouping.sizes.data(), G, ec_counts.data(), a.q,
                                       a.e, a.zero_inflation, a.min_hits, &n_kept, mask.data(), nullptr));
  } catch (const std::exception &ex) {
    std::cerr << "Building the log-likelihood array failed:\n  " << ex.what() << "\nexiting\n";
    msw_core_destroy(h);
    return 1;
  }
  std::vector<double> prior(n_kept, 1.0);
  if (!a.alphas.empty()) {
    const auto parts = split(a.alphas, ',');
    if (parts.size() != n_kept) {
      std::cerr << "Error: --alphas must have the same number of values as there are groups.";
      msw_core_destroy(h);
      return 1;
    }
    for (size_t i = 0; i < n_kept; ++i) prior[i] = std::stod(parts[i]);
  }
  uint64_t total = 0;
  for (uint64_t c : ec_counts) total += c;
  std::vector<std::vector<double>> results;  // [0] = estimate without resampling (include/Sample.hpp:157)
  try {
    std::vector<double> theta(n_kept);
    size_t it = 0;
    double bound = 0.0;
    // logc = NULL: the log counts stay where the build left them, on the device
    check(h, msw_core_solve(h, nullptr, prior.data(), a.tol, a.max_iters, algo, prec, theta.data(), &it, &bound));
    if (a.verbose) {
      const size_t n = std::min<size_t>(it, 4096);
      std::vector<double> b(n), nn(n);
      size_t got = 0;
      check(h, msw_core_trace(h, n, b.data(), nn.data(), nullptr, nullptr, nullptr, &got));
      char buf[128];
      for (size_t k = 0; k < got; k += 5) {
        snprintf(buf, sizeof buf, "  iter: %zu, bound: %g, |g|: %g\n", k, b[k], nn[k]);
        std::cerr << buf;
      }
    }
    results.push_back(theta);
    if (a.iters > 0) {
      int32_t seed;
      if (a.seed == 26012023) {  // the reference's "random seed" sentinel (src/BootstrapSample.cpp:48-50)
        seed = (int32_t)(std::random_device{}() & 0x7fffffffu);
      } else {
        seed = (int32_t)(uint32_t)a.seed;  // size_t -> int32 narrowing (include/Sample.hpp:169)
      }
      // ConstructSample quirk (src/Sample.cpp:38-39): --bootstrap-count without --bin-reads passes the
      // number of ITERATIONS as the count
      const size_t draws = a.bootstrap_count > 0 ? a.iters : (size_t)total;
      std::vector<uint32_t> w(ec_counts.begin(), ec_counts.end());
      std::vector<double> thetas(a.iters * n_kept);
      check(h, msw_core_bootstrap(h, w.data(), seed, draws, 0, a.iters, prior.data(), a.tol, a.max_iters, algo, prec,
                                  thetas.data(), nullptr));
      for (size_t b = 0; b < a.iters; ++b)
        results.emplace_back(thetas.begin() + b * n_kept, thetas.begin() + (b + 1) * n_kept);
    }
  } catch (const std::exception &ex) {
    std::cerr << "Estimating relative abundances failed:\n  " << ex.what() << "\nexiting\n";
    msw_core_destroy(h);
    return 1;
  }
  msw_core_destroy(h);

  // ---- abundances (default ostream formatting = 6 significant digits, as the reference) ----------
  std::ofstream file;
  if (!a.prefix.empty()) file.open(a.prefix + "_abundances.txt");
  std::ostream &of = a.prefix.empty() ? std::cout : file;
  of << "#mSWEEP_version:\t" << kVersion << '\n' << "#num_reads:\t" << n_reads << '\n' << "#num_aligned:\t" << total << '\n';
  if (a.iters > 0) of << "#bootstrap_iters:\t" << a.iters << '\n' << "#c_id\tmean_theta\tbootstrap_mean_thetas\n";
  else of << "#c_id\tmean_theta\n";
  size_t row = 0;
  for (size_t g = 0; g < G; ++g) {  // estimated groups (all of them unless --min-hits pruned some)
    if (!mask[g]) continue;
    of << grouping.names[g];
    for (auto &r : results) of << '\t' << r[row];
    of << '\n';
    ++row;
  }
  if (a.min_hits > 0)
    for (size_t g = 0; g < G; ++g) {
      if (mask[g]) continue;
      of << grouping.names[g];
      for (size_t k = 0; k < results.size(); ++k) of << "\t0";
      of << '\n';
    }
  of.flush();
  return 0;
}
