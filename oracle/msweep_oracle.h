/*
 * msweep_oracle.h -- CPU ORACLE for the mSWEEP abundance-estimation hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check in
 * __graft_entry__.py and the `cpu_baseline` leg of bench.py may load it.  The
 * shipped library (msweep_amd/csrc -> libmsweep_core.so) never links, loads or calls
 * anything in this directory.
 *
 * PARITY STATUS: "parity unpinned" for the optimiser loop.  The reference tree
 * (/root/reference) holds no tests, golden vectors or fixtures and
 * the RCG / EM loops live in the un-vendored dependency rcgpar v1.2.1
 * (CMakeLists.txt:274-311), which is absent from this container.  The optimiser part
 * of this oracle therefore restates rcgpar's *published* algorithm (Hensman et al.
 * 2012 / BitSeq VariationalBayes; rcgpar src/rcg.cpp as published on GitHub at tag
 * v1.2.1) and is anchored on the reference's own call sites
 * (src/mSWEEP.cpp:176-205, 391-398, 419-423, 496-518).  The parts whose source IS in the
 * reference tree are pinned:
 *   - likelihood LUT   (include/Likelihood.hpp:47-60,92-107,198-207) against
 *     mpmath/scipy golden vectors (tests/golden/lut_*.json),
 *   - digamma          (src/Sample.cpp:87-97) against scipy,
 *   - bootstrap stream (src/BootstrapSample.cpp:33-73) against libstdc++'s own
 *     std::mt19937_64 + std::discrete_distribution<uint32_t> (the exact types the
 *     reference instantiates), golden vectors in tests/golden/bootstrap_*.json.
 */
#ifndef MSWEEP_ORACLE_H
#define MSWEEP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- knobs that are "from memory" of rcgpar (SURVEY.md 3.2 dagger) ------------------ */
typedef struct orc_rcg_opts {
  double init_bound;      /* rcgpar: long double bound = -100000.0 before the loop      */
  int    weight_newnorm;  /* 0 = rcgpar (newnorm not weighted by EC counts)             */
  int    max_trace;       /* number of iterations to record in trace arrays             */
  int    check_every;     /* 1 = rcgpar as restated (stop rule tested after every iteration);
                           * n > 1: tested only after iterations n, 2n, ... (every published
                           * iteration count is a multiple of 5, docs/gpubenchmarks.md:15-25 -- the verbose log
                           * prints every 5th iteration, and "checked on a 5-grid" is the other reading)  */
  int    extended;        /* 0 = fp64 arithmetic as in rcgpar and in the HIP kernels (default); 1 = everything between
                           * the fp64 state and the fp64 results in x87 extended precision (64-bit significand):
                           * orc_rcg_optl_csr's exponentials, row sums, U - sum_nz differences, column sums;
                           * orc_rcg_optl_dense's four G x E matrices.  The judge of tools/fuzz_parity.py on inputs
                           * that amplify rounding (a Fletcher-Reeves factor ~ 100): the structured fp64 oracle drifted
                           * 1e-6 there while the HIP path, whose column sums are exact integers, stayed at 1e-8. */
} orc_rcg_opts;

/* EM variants (rcgpar::em_torch is absent: [UPSTREAM-UNVERIFIED]; the alternatives are explicit) */
typedef struct orc_em_opts {
  int prior_mode;   /* 0 = MAP: Dirichlet(alpha0) as pseudo-counts alpha0 - 1 (default); 1 = ML: alpha0 ignored */
  int stop_rule;    /* 0 = gain of the weighted log-likelihood < tol (default); 1 = max_g |theta_new - theta_old| < tol */
  int check_every;  /* as orc_rcg_opts.check_every */
} orc_em_opts;

/* per-iteration trace (arrays of length opts.max_trace, may be NULL) */
typedef struct orc_rcg_trace {
  double *bound;    /* bound after the iteration (after a reset re-evaluation)          */
  double *newnorm;  /* |g| of the iteration                                             */
  double *beta;     /* Fletcher-Reeves beta                                             */
  int    *didreset; /* 1 if the iteration took the steepest-descent retry               */
  double *theta;    /* [max_trace x G] mixture weights after the iteration              */
} orc_rcg_trace;

void orc_default_opts(orc_rcg_opts *o);

/* ---- scalar helpers ----------------------------------------------------------------- */
double orc_digamma(double x);                                   /* Sample.cpp:87-97 */
double orc_lbeta(double x, double y);                           /* Likelihood.hpp:47-50 */
double orc_ldbb_scaled(uint64_t k, uint64_t n, double alpha, double beta); /* :57-60 */

/* ---- likelihood (include/Likelihood.hpp) -------------------------------------------- */
/* :198-207; bb_constants[0]=q (-q flag), [1]=e (-e flag) via :212-214, mSWEEP.cpp:346 */
void orc_bb_params(const uint64_t *group_sizes, size_t n_groups, double q, double e,
                   double *alpha_out, double *beta_out);
/* :92-107; lut is n_groups x ld row-major, ld >= max_size+1; column 0 = log(zi) */
void orc_precalc_lls(const uint64_t *group_sizes, size_t n_groups, double q, double e,
                     double zero_inflation, double *lut, size_t ld);
/* :122-139 from an EC->target hit list (CSR over targets): counts[g*E + i] dense u32 */
void orc_group_counts(const uint64_t *ec_tptr, const uint32_t *ec_targets, size_t n_ecs,
                      const uint32_t *target_group, size_t n_groups, uint32_t *counts);
/* :141-186; returns n_masked_groups; L_out is n_masked x E row-major (rows = groups),
 * mask_out[G] (1 = considered), masked group order = original order among kept groups. */
size_t orc_fill_ll_mat(const uint32_t *counts, const uint64_t *ec_counts, size_t n_ecs,
                       const uint64_t *group_sizes, size_t n_groups, double q, double e,
                       double zero_inflation, size_t min_hits, double *L_out,
                       uint8_t *mask_out);
/* :188-195 */
void orc_fill_ec_counts(const uint64_t *ec_counts, size_t n_ecs, double *logc_out);

/* ---- optimiser (rcgpar v1.2.1 restated) ------------------------------- */
/* Dense-state RCG exactly as the reference structures it: logl is G x E row-major
 * (rows = groups), gamma/step/oldstep are allocated G x E.  gamma_out (G x E) receives
 * the log-responsibilities.  Returns the number of iterations executed. */
size_t orc_rcg_optl_dense(const double *logl, size_t G, size_t E, const double *logc,
                          const double *alpha0, double tol, size_t max_iters,
                          const orc_rcg_opts *opts, double *gamma_out, double *bound_out,
                          orc_rcg_trace *trace);

/* Structured restatement (gamma_gj = a*L_gj + u_g - lse_j) on a
 * CSR-of-ECs likelihood: for EC j, cells k in [rowptr[j], rowptr[j+1]) hold group id
 * grp[k] and LUT slot lutidx[k]; L = lut[lutidx[k]]; every other cell is logzi.
 * theta_out[G], gamma_out optional (G x E row-major) or NULL. */
size_t orc_rcg_optl_csr(const uint64_t *rowptr, const uint32_t *grp, const uint32_t *lutidx,
                        const double *lut, size_t n_lut, double logzi, size_t G, size_t E,
                        const double *logc, const double *alpha0, double tol,
                        size_t max_iters, const orc_rcg_opts *opts, double *theta_out,
                        double *gamma_out, double *bound_out, orc_rcg_trace *trace);

/* Structured restatement on a dense L (rows = groups, G x E row-major). */
size_t orc_rcg_optl_dense_structured(const double *logl, size_t G, size_t E,
                                     const double *logc, const double *alpha0, double tol,
                                     size_t max_iters, const orc_rcg_opts *opts,
                                     double *theta_out, double *gamma_out,
                                     double *bound_out, orc_rcg_trace *trace);

/* rcgpar::mixture_components (call sites mSWEEP.cpp:420,513) */
void orc_mixture_components(const double *gamma, size_t G, size_t E, const double *logc,
                            double *theta_out);

/* rcgpar::em_torch restated (call site mSWEEP.cpp:202); dense L, fp64.  [UPSTREAM-UNVERIFIED] */
size_t orc_em_dense(const double *logl, size_t G, size_t E, const double *logc,
                    const double *alpha0, double tol, size_t max_iters, double *gamma_out,
                    double *theta_out, double *bound_out);
/* the same with the variant switches (opts == NULL: the defaults = orc_em_dense) */
size_t orc_em_dense_opts(const double *logl, size_t G, size_t E, const double *logc,
                         const double *alpha0, double tol, size_t max_iters, const orc_em_opts *opts,
                         double *gamma_out, double *theta_out, double *bound_out);
/* em_torch with precision "float" (src/mSWEEP.cpp:129,202) restated: fp32 arithmetic, the log-likelihood rounded to
 * float once per iteration for the stop rule (rcg_oracle.cpp).  theta_trace: n_trace x G or NULL. */
size_t orc_em_dense_f32(const double *logl, size_t G, size_t E, const double *logc, const double *alpha0, double tol,
                        size_t max_iters, const orc_em_opts *opts, double *theta_out, double *bound_out,
                        double *theta_trace, size_t n_trace);

/* ---- bootstrap (src/BootstrapSample.cpp:33-73) --------------------------------------- */
/* libstdc++ types, exactly as the reference instantiates them.  One call = `n_reps`
 * consecutive resample_counts() calls on one generator seeded with `seed`.
 * counts_out is n_reps x n_ecs (uint32). */
void orc_bootstrap_counts_stdlib(const uint32_t *weights, size_t n_ecs, int32_t seed,
                                 size_t bootstrap_count, size_t n_reps,
                                 uint32_t *counts_out);
/* the libstdc++ replicate loop entered in the middle of the stream, from a state libstdc++ printed itself
 * (operator<< of std::mt19937_64: 312 words + position; tests/golden/mt_deep_state.json) */
void orc_bootstrap_counts_stdlib_from_state(const uint32_t *weights, size_t n_ecs, const uint64_t *state312,
                                            uint64_t pos, size_t bootstrap_count, size_t n_reps, uint32_t *out);
/* From-scratch restatement of the same stream (MT19937-64 + generate_canonical +
 * lower_bound on the normalised partial sums); this is what the GPU path mirrors. */
void orc_bootstrap_counts_restated(const uint32_t *weights, size_t n_ecs, int32_t seed,
                                   size_t bootstrap_count, size_t n_reps,
                                   uint32_t *counts_out);
/* cumulative table the restatement searches (length n_ecs) */
void orc_discrete_cp(const uint32_t *weights, size_t n_ecs, double *cp_out);
/* raw MT19937-64 words from a seed (for testing the device generator) */
void orc_mt19937_64_words(uint64_t seed, size_t skip, size_t n, uint64_t *out);

/* ---- --run-rate statistics (src/Sample.cpp:99-152) -------------------------------------- */
/* Sample::dirichlet_kld + Sample::get_rates on the G x E log-responsibility matrix: alphas_i = sum_j
 * round(c_j) additions of exp(gamma_ij), KLD_i clamped at 1e-16, RATE_i = KLD_i / sum KLD through the
 * source's log-sum-exp (shift max(0, max log KLD)).  Outputs [G], any may be NULL. */
void orc_dirichlet_kld_rate(const double *gamma, size_t G, size_t E, const double *logc,
                            double *alphas_out, double *kld_out, double *rate_out);

int orc_num_threads(void);
void orc_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
