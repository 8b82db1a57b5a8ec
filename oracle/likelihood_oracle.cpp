// likelihood_oracle.cpp -- CPU ORACLE (test infrastructure, see msweep_oracle.h).
// Restates include/Likelihood.hpp of the reference: the scaled log beta-binomial pmf
// (:47-60), the per-group parameters (:198-207), the lookup table (:92-107), the
// EC x group hit counts (:122-139), the --min-hits mask and compaction (:141-171), the
// dense materialisation (:176-185) and the log EC counts (:188-195).
#include "msweep_oracle.h"

#include <cmath>
#include <vector>

extern "C" {

double orc_lbeta(double x, double y) {  // Likelihood.hpp:47-50
  return std::lgamma(x) + std::lgamma(y) - std::lgamma(x + y);
}

static double log_bin_coeff(uint64_t n, uint64_t k) {  // Likelihood.hpp:52-55
  return std::lgamma((double)(n + 1)) - std::lgamma((double)(k + 1)) -
         std::lgamma((double)(n - k + 1));
}

double orc_ldbb_scaled(uint64_t k, uint64_t n, double alpha, double beta) {  // :57-60
  return log_bin_coeff(n, k) + orc_lbeta((double)k + alpha, (double)(n - k) + beta) -
         orc_lbeta((double)n + alpha, beta);
}

void orc_bb_params(const uint64_t *sizes, size_t G, double q, double eps, double *alpha_out,
                   double *beta_out) {  // :198-207 with bb_constants = {q, e}
  for (size_t i = 0; i < G; ++i) {
    const double n = (double)sizes[i];
    const double e = n * q;
    const double phi = 1.0 / (n - e + eps);
    const double beta = phi * (n - e);
    const double alpha = (e * beta) / (n - e);
    alpha_out[i] = alpha;
    beta_out[i] = beta;
  }
}

void orc_precalc_lls(const uint64_t *sizes, size_t G, double q, double eps, double zi,
                     double *lut, size_t ld) {  // :92-107
  uint64_t max_size = 0;
  for (size_t i = 0; i < G; ++i) max_size = sizes[i] > max_size ? sizes[i] : max_size;
  std::vector<double> al(G), be(G);
  orc_bb_params(sizes, G, q, eps, al.data(), be.data());
  const double lzi = std::log(zi), l1m = std::log1p(-zi);
  for (size_t i = 0; i < G; ++i) {
    for (size_t j = 0; j < ld; ++j) lut[i * ld + j] = lzi;
    // the reference also fills k > n_g (never indexed; lgamma of a negative integer).
    // The oracle keeps those at log(zi) so fixtures stay finite.
    for (uint64_t j = 1; j <= max_size && j <= sizes[i] && j < ld; ++j)
      lut[i * ld + j] = orc_ldbb_scaled(j, sizes[i], al[i], be[i]) + l1m;
  }
}

void orc_group_counts(const uint64_t *tptr, const uint32_t *targets, size_t E,
                      const uint32_t *target_group, size_t G, uint32_t *counts) {  // :122-139
  for (size_t i = 0; i < G * E; ++i) counts[i] = 0;
  for (size_t i = 0; i < E; ++i)
    for (uint64_t k = tptr[i]; k < tptr[i + 1]; ++k)
      counts[(size_t)target_group[targets[k]] * E + i] += 1;
}

size_t orc_fill_ll_mat(const uint32_t *counts, const uint64_t *ec_counts, size_t E,
                       const uint64_t *sizes, size_t G, double q, double eps, double zi,
                       size_t min_hits, double *L_out, uint8_t *mask_out) {  // :141-186
  const bool mask_groups = min_hits > 0;
  std::vector<uint64_t> masked_sizes;
  std::vector<size_t> pos(G, 0);
  size_t n_masked = 0;
  if (mask_groups) {
    for (size_t g = 0; g < G; ++g) {
      uint64_t hits = 0;
      for (size_t i = 0; i < E; ++i) hits += (counts[g * E + i] > 0) * ec_counts[i];
      mask_out[g] = hits >= min_hits;
      if (mask_out[g]) {
        pos[g] = n_masked++;
        masked_sizes.push_back(sizes[g]);
      }
    }
  } else {
    for (size_t g = 0; g < G; ++g) { mask_out[g] = 1; pos[g] = g; masked_sizes.push_back(sizes[g]); }
    n_masked = G;
  }
  if (n_masked == 0) return 0;
  uint64_t max_size = 0;
  for (uint64_t s : masked_sizes) max_size = s > max_size ? s : max_size;
  const size_t ld = max_size + 1;
  std::vector<double> lut(n_masked * ld);
  orc_precalc_lls(masked_sizes.data(), n_masked, q, eps, zi, lut.data(), ld);
  for (size_t g = 0; g < G; ++g) {
    if (!mask_out[g]) continue;
    for (size_t j = 0; j < E; ++j) L_out[pos[g] * E + j] = lut[pos[g] * ld + counts[g * E + j]];
  }
  return n_masked;
}

void orc_fill_ec_counts(const uint64_t *ec_counts, size_t E, double *logc) {  // :188-195
  for (size_t i = 0; i < E; ++i) logc[i] = std::log((double)ec_counts[i]);
}

}  // extern "C"
