// sample_oracle.cpp -- CPU ORACLE (test infrastructure, see msweep_oracle.h header note).
//
// Restates the experimental --run-rate statistics of the reference: Sample::dirichlet_kld
// (src/Sample.cpp:99-131) and Sample::get_rates (src/Sample.cpp:133-152), which consume the
// log-responsibility matrix the optimiser returned.  Source in the reference tree: pinned by
// reading, no fixtures exist there.
#include <algorithm>
#include <cmath>
#include <vector>

#include "msweep_oracle.h"

extern "C" {

// gamma: G x E row-major (rows = groups), logc[E].  alphas_out / kld_out / rate_out: [G], any may be NULL.
void orc_dirichlet_kld_rate(const double *gamma, size_t G, size_t E, const double *logc, double *alphas_out,
                            double *kld_out, double *rate_out) {
  // :103-112  alphas[i] = sum over ECs of num_hits additions of exp(gamma(i, j))
  std::vector<double> alphas(G, 0.0);
  for (size_t i = 0; i < G; ++i) {
    for (size_t j = 0; j < E; ++j) {
      const size_t num_hits = (size_t)std::round(std::exp(logc[j]));
      for (size_t k = 0; k < num_hits; ++k) alphas[i] += std::exp(gamma[i * E + j]);
    }
  }
  // :114-118
  double alpha0 = 0.0;
  for (size_t i = 0; i < G; ++i) alpha0 += alphas[i];
  // :120-128 (log_theta and alpha_k of the source are unused there)
  std::vector<double> log_KLDs(G);
  for (size_t i = 0; i < G; ++i) {
    const double alpha_j = alphas[i];
    const double KLD = std::max(std::lgamma(alpha0) - std::lgamma(alpha0 - alpha_j) - std::lgamma(alpha_j) +
                                    alpha_j * (orc_digamma(alpha_j) - orc_digamma(alpha0)),
                                1e-16);
    log_KLDs[i] = std::log(KLD);
  }
  // :133-152  (max_elem starts at 0.0 in the source: the shift is max(0, max log KLD))
  double max_elem = 0.0;
  for (size_t i = 0; i < G; ++i) max_elem = (max_elem > log_KLDs[i] ? max_elem : log_KLDs[i]);
  double tmp_sum = 0.0;
  for (size_t i = 0; i < G; ++i) tmp_sum += std::exp(log_KLDs[i] - max_elem);
  const double log_KLDs_sum = std::log(tmp_sum) + max_elem;
  for (size_t i = 0; i < G; ++i) {
    if (alphas_out) alphas_out[i] = alphas[i];
    if (kld_out) kld_out[i] = std::exp(log_KLDs[i]);  // the table prints exp(log_KLD) (src/mSWEEP.cpp:529-545)
    if (rate_out) rate_out[i] = std::exp(log_KLDs[i] - log_KLDs_sum);
  }
}

}  // extern "C"
