// msweep_mini.cpp -- the estimation path of mSWEEP's main() (src/mSWEEP.cpp:258-551) as a native
// host program over the C ABI: group indicators (-i, include/Reference.hpp / Grouping.hpp), Themisto
// plaintext pseudoalignments (the reader on the device: msw_alignment_read_device), likelihood built and kept on the
// GPU (msw_core_build_likelihood), RCG / EM abundances (--algorithm rcggpu|emgpu), bootstrap
// (--iters / --seed / --bootstrap-count, src/mSWEEP.cpp:496-518) and abundances.txt in the format of
// PlainSample / BootstrapSample::write_abundances[2] (src/PlainSample.cpp:32-71,
// src/BootstrapSample.cpp:75-130), plus the consumers around the path: --write-probs / --print-probs (Sample::write_probs,
// src/Sample.cpp:63-85,154-186: streamed from the device in blocks of ECs), --write-likelihood / --read-likelihood /
// --no-fit-model (include/Likelihood.hpp:224-273, src/mSWEEP.cpp:357-386) and --run-rate (src/Sample.cpp:99-152,
// src/mSWEEP.cpp:524-548).  Same flags and messages as the reference for what it covers; held byte-for-byte against
// the Python mirror `python -m msweep_amd` in tests/test_gpu_cli_toy.py.
//
//   g++ -std=c++17 -O2 -o msweep_mini msweep_mini.cpp -L.. -lmsweep_core -Wl,-rpath,..
#include <algorithm>
#include <cmath>
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
  bool write_probs = false, print_probs = false, write_likelihood = false, no_fit_model = false, run_rate = false;
  std::string read_likelihood;
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
    else if (k == "--write-probs") a.write_probs = true;
    else if (k == "--print-probs") a.print_probs = true;
    else if (k == "--write-likelihood") a.write_likelihood = true;
    else if (k == "--read-likelihood") a.read_likelihood = val();
    else if (k == "--no-fit-model") a.no_fit_model = true;
    else if (k == "--run-rate") a.run_rate = true;
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

// a number as the reference's `*of << x` prints it (default ostream formatting: 6 significant digits)
std::string g6(double x) {
  char buf[64];
  snprintf(buf, sizeof buf, "%g", x);
  return buf;
}

// LL_WOR21::from_file (include/Likelihood.hpp:224-253): one line per equivalence class, `count \t L(0,j) ... L(G-1,j)`
void read_likelihood_file(const std::string &path, size_t G, std::vector<uint64_t> &counts, std::vector<double> &L) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error("Could not read from the likelihoods file.");
  std::vector<std::vector<double>> cols;
  std::string line;
  while (std::getline(in, line)) {
    const auto parts = split(line, '\t');
    if (parts.size() != G + 1) throw std::runtime_error("Could not read from the likelihoods file.");
    counts.push_back(std::stoull(parts[0]));
    std::vector<double> c(G);
    for (size_t g = 0; g < G; ++g) c[g] = std::stod(parts[g + 1]);
    cols.push_back(std::move(c));
  }
  const size_t E = cols.size();
  L.assign(G * E, 0.0);
  for (size_t j = 0; j < E; ++j)
    for (size_t g = 0; g < G; ++g) L[g * E + j] = cols[j][g];
}

// Sample::write_probs[2] (src/Sample.cpp:63-85,154-186): header `ec_id` + group names, one line per equivalence class
// of exp(gamma); the G x E matrix is streamed from the device 8192 classes at a time (msw_core_gamma_block)
void write_probs(std::ostream &of, msw_handle h, const std::vector<std::string> &names, const std::vector<std::string> &zero_names,
                 size_t n_groups, size_t n_ecs) {
  of << "ec_id";
  for (auto &n : names) of << '\t' << n;
  for (auto &n : zero_names) of << '\t' << n;
  of << '\n';
  const size_t block = 8192;
  std::vector<double> buf(n_groups * block);
  for (size_t e0 = 0; e0 < n_ecs; e0 += block) {
    const size_t w = std::min(block, n_ecs - e0);
    check(h, msw_core_gamma_block(h, e0, e0 + w, buf.data(), w));
    for (size_t jj = 0; jj < w; ++jj) {
      of << e0 + jj;
      for (size_t g = 0; g < n_groups; ++g) of << '\t' << g6(std::exp(buf[g * w + jj]));
      for (size_t z = 0; z < zero_names.size(); ++z) of << "\t0";
      of << '\n';
    }
  }
  of << '\n';
  of.flush();
}

// digamma as the reference evaluates it (src/Sample.cpp:87-97)
double digamma_ref(double x) {
  double result = 0, xx, xx2, xx4;
  for (; x < 7; ++x) result -= 1 / x;
  x -= 1.0 / 2.0;
  xx = 1.0 / x;
  xx2 = xx * xx;
  xx4 = xx2 * xx2;
  result += std::log(x) + (1. / 24.) * xx2 - (7.0 / 960.0) * xx4 + (31.0 / 8064.0) * xx4 * xx2 - (127.0 / 30720.0) * xx4 * xx4;
  return result;
}
// Sample::dirichlet_kld + get_rates (src/Sample.cpp:99-152) on alphas_i = theta_i * sum c -- the column sums the solve
// already reduced on the device
void dirichlet_kld_rate(const std::vector<double> &alphas, std::vector<double> &kld, std::vector<double> &rate) {
  double a0 = 0.0;
  for (double a : alphas) a0 += a;
  std::vector<double> lk(alphas.size());
  double mx = 0.0;
  for (size_t i = 0; i < alphas.size(); ++i) {
    const double aj = alphas[i];
    const double v = std::lgamma(a0) - std::lgamma(a0 - aj) - std::lgamma(aj) + aj * (digamma_ref(aj) - digamma_ref(a0));
    lk[i] = std::log(std::max(v, 1e-16));
    mx = std::max(mx, lk[i]);
  }
  double s = 0.0;
  for (double v : lk) s += std::exp(v - mx);
  const double lsum = std::log(s) + mx;
  kld.resize(lk.size());
  rate.resize(lk.size());
  for (size_t i = 0; i < lk.size(); ++i) {
    kld[i] = std::exp(lk[i]);
    rate[i] = std::exp(lk[i] - lsum);
  }
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
  std::vector<uint64_t> ec_counts;
  // The reader runs ON THE DEVICE (msw_alignment_read_device: text -> equivalence classes in HBM) and the likelihood
  // build consumes its arrays there (msw_core_build_likelihood_aln); only the classes' read counts come to the host.
  msw_handle h = nullptr;
  if (msw_core_create(a.gpu, &h) != 0) {
    std::cerr << "Initialising the GPU failed:\n  " << msw_last_error(nullptr) << "\nexiting\n";
    return 1;
  }
  msw_alignment_t aln_keep = nullptr;
  size_t n_ecs = 0, n_reads = 0, n_hits = 0, n_aligned = 0;
  try {
    grouping = read_grouping(a.indicators);
    if (a.read_likelihood.empty()) {  // (--read-likelihood needs no pseudoalignments: src/mSWEEP.cpp:296-370)
      if (a.themisto.empty()) throw std::runtime_error("no pseudoalignment files given");
      if (a.mode != "intersection" && a.mode != "union")
        throw std::runtime_error("Unrecognized option `" + a.mode + "` for --themisto-mode");
      std::vector<const char *> paths;
      for (auto &p : a.themisto) paths.push_back(p.c_str());
      msw_alignment_t aln = nullptr;
      if (msw_alignment_read_device(h, paths.data(), paths.size(), grouping.indicators.size(),
                                    a.mode == "union" ? MSW_MERGE_UNION : MSW_MERGE_INTERSECTION, &aln))
        throw std::runtime_error(msw_alignment_last_error());
      msw_alignment_shape(aln, &n_ecs, &n_reads, &n_hits, &n_aligned);
      ec_counts.resize(n_ecs);
      msw_alignment_export(aln, nullptr, nullptr, ec_counts.data(), nullptr, nullptr);
      aln_keep = aln;
    }
  } catch (const std::exception &ex) {
    std::cerr << "Reading the pseudoalignments failed:\n  " << ex.what() << "\nexiting\n";
    msw_core_destroy(h);
    return 1;
  }
  if (a.algorithm == "rcgcpu" && a.verbose)  // the reference's default: the same RCG algorithm on the host; no CPU path here
    std::cerr << "note: --algorithm rcgcpu is served by the GPU RCG kernels (same algorithm as rcggpu)\n";
  const int algo = (a.algorithm == "rcggpu" || a.algorithm == "rcgcpu") ? MSW_ALGO_RCG : MSW_ALGO_EM;  // else em (src/mSWEEP.cpp:200)
  const int prec = a.emprecision == "float" ? MSW_PREC_FLOAT : MSW_PREC_DOUBLE;
  // (--emprecision float: fp32 kernels where the layout allows, msweep_amd/csrc/em_f32_kernels.hpp)
  const size_t G = grouping.names.size();
  size_t n_kept = 0;
  std::vector<uint8_t> mask(G, 1);
  std::vector<double> logc_file;  // --read-likelihood: the log counts of the file (the build leaves its own on the device)
  try {
    // ordering the cells for the LDS banks pays from about the 1 000th iteration on: bootstrap runs
    check(h, msw_core_set_pack_schedule(h, a.iters >= 5 ? 1 : 0));
    if (!a.read_likelihood.empty()) {
      // --read-likelihood (include/Likelihood.hpp:224-253): the dense matrix of a file through the dense boundary
      std::vector<double> L;
      read_likelihood_file(a.read_likelihood, G, ec_counts, L);
      n_ecs = ec_counts.size();
      if (n_ecs == 0) throw std::runtime_error("Could not read from the likelihoods file.");
      check(h, msw_core_set_dense_logl(h, L.data(), G, n_ecs, n_ecs));
      n_kept = G;
      for (uint64_t c : ec_counts) {
        logc_file.push_back(std::log((double)c));
        n_reads += c;   // (no alignment: the reads are those the file counts, as the Python mirror reports them)
      }
    } else {
      if (n_ecs == 0) throw std::runtime_error("no read aligned against the reference");
      check(h, msw_core_build_likelihood_aln(h, aln_keep, grouping.indicators.data(), grouping.indicators.size(),
                                             grouping.sizes.data(), G, a.q, a.e, a.zero_inflation, a.min_hits, &n_kept,
                                             mask.data(), nullptr));
      // (the likelihood is resident and the pseudoalignment could go -- but device memory given back is scrubbed before
      // it is handed out again, and the solver state is allocated next: it goes at the end)
    }
    if (a.write_likelihood) {
      // --write-likelihood (include/Likelihood.hpp:255-273; the file: src/OutfileDesignator.cpp:67-74)
      std::vector<double> L(n_kept * n_ecs);
      check(h, msw_core_get_dense_logl(h, L.data(), n_ecs));
      std::ofstream lf(a.prefix.empty() ? std::string("likelihoods.tsv") : a.prefix + "_likelihoods.tsv");
      for (size_t j = 0; j < n_ecs; ++j) {
        lf << ec_counts[j];
        for (size_t g = 0; g < n_kept; ++g) lf << '\t' << g6(L[g * n_ecs + j]);
        lf << '\n';
      }
    }
  } catch (const std::exception &ex) {
    std::cerr << "Building the log-likelihood array failed:\n  " << ex.what() << "\nexiting\n";
    msw_core_destroy(h);
    return 1;
  }
  if (a.no_fit_model) {  // src/mSWEEP.cpp:385-386
    msw_core_destroy(h);
    return 0;
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
  std::vector<std::string> est_names, zero_names;
  for (size_t g = 0; g < G; ++g) (mask[g] ? est_names : zero_names).push_back(grouping.names[g]);
  std::vector<std::vector<double>> results;  // [0] = estimate without resampling (include/Sample.hpp:157)
  try {
    std::vector<double> theta(n_kept);
    size_t it = 0;
    double bound = 0.0;
    // logc = NULL: the log counts stay where the build left them, on the device
    check(h, msw_core_solve(h, logc_file.empty() ? nullptr : logc_file.data(), prior.data(), a.tol, a.max_iters, algo, prec,
                            theta.data(), &it, &bound));
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
    // --write-probs / --print-probs: the probabilities of the un-resampled estimate (the replicates ran on solver
    // states of their own: the handle still holds it -- the reference writes them before its replicate loop,
    // src/mSWEEP.cpp:471-493)
    if (a.write_probs || a.print_probs) {
      const std::vector<std::string> none;
      const std::vector<std::string> &zn = a.min_hits > 0 ? zero_names : none;
      if (a.write_probs && !a.prefix.empty()) {
        std::ofstream pf(a.prefix + "_probs.tsv");
        write_probs(pf, h, est_names, zn, n_kept, n_ecs);
      }
      if (a.print_probs || (a.write_probs && a.prefix.empty())) write_probs(std::cout, h, est_names, zn, n_kept, n_ecs);
    }
  } catch (const std::exception &ex) {
    std::cerr << "Estimating relative abundances failed:\n  " << ex.what() << "\nexiting\n";
    msw_core_destroy(h);
    return 1;
  }
  msw_alignment_destroy(aln_keep);
  msw_core_destroy(h);

  // ---- abundances (default ostream formatting = 6 significant digits, as the reference) ----------
  std::ofstream file;
  if (!a.prefix.empty()) file.open(a.prefix + "_abundances.txt");
  std::ostream &of = a.prefix.empty() ? std::cout : file;
  of << "#mSWEEP_version:\t" << kVersion << '\n' << "#num_reads:\t" << n_reads << '\n' << "#num_aligned:\t" << total << '\n';
  if (a.run_rate) {
    // experimental RATE / KLD (src/Sample.cpp:99-152; the table: src/mSWEEP.cpp:529-545)
    std::vector<double> alphas(n_kept), kld, rate;
    for (size_t i = 0; i < n_kept; ++i) alphas[i] = results[0][i] * (double)total;
    dirichlet_kld_rate(alphas, kld, rate);
    of << "#c_id\tmean_theta\tRATE\tKLD\n";
    for (size_t i = 0; i < n_kept; ++i)
      of << est_names[i] << '\t' << g6(results[0][i]) << '\t' << g6(rate[i]) << '\t' << g6(kld[i]) << '\n';
    for (auto &n : zero_names) of << n << "\t0\t0\t0\n";
    of.flush();
    return 0;
  }
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
