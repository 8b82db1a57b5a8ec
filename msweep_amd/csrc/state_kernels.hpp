// state_kernels.hpp -- the O(G) kernels between the sweeps: Fletcher-Reeves step, column-sum
// reduction + N_g / lgamma / digamma (spread over many workgroups: the transcendental work of
// 5k groups on ONE CU cost more than a sweep), ELBO + reset / convergence decision, and the
// gradient preparation for the next pass A.
//
// Per iteration:  k_passA -> k_step -> k_passB -> k_redfin -> k_fin(0)
//                 [-> k_passB(cond) -> k_redfin(cond) -> k_fin(1)]   (only after a rejected step)
#pragma once
#include "device_util.hpp"

namespace msw {

struct TraceDev {
  double *bound, *newnorm, *beta, *theta;
  int32_t *didreset;
};

// From (a, u): M = max u, e_g = exp(u_g - M), U = sum e, p0 = exp(a*logzi) and the table
// X[i] = exp(a*T_i).  One 1024-thread workgroup.  Index G of e is the sentinel slot (zero).
__device__ inline void prepB_block(Scalars *sc, double a, int G, int n_lut, const double *u,
                                   const double *lut, double *e, double *X, double *sh) {
  const int tid = threadIdx.x, nt = blockDim.x;
  double m = -INFINITY;
  for (int g = tid; g < G; g += nt) m = fmax(m, u[g]);
  const double M = block_max(m, sh);
  double su = 0.0;
  for (int g = tid; g < G; g += nt) {
    const double eg = exp(u[g] - M);
    e[g] = eg;
    su += eg;
  }
  const double U = block_sum(su, sh);
  const double p0 = exp(a * sc->logzi);
  for (int i = tid; i < n_lut; i += nt) X[i] = exp(a * lut[i]);
  if (tid == 0) {
    sc->M = M;
    sc->U = U;
    sc->p0 = p0;
  }
}

__global__ __launch_bounds__(1024) void k_prepB(Scalars *sc, int G, int n_lut, const double *u,
                                               const double *lut, double *e, double *X) {
  __shared__ double sh[32];
  if (sc->done) return;
  if (sc->flavor != 0) return;  // dense flavour needs no tables
  const double a = sc->a;
  __syncthreads();
  prepB_block(sc, a, G, n_lut, u, lut, e, X, sh);
}

// Gradient preparation for pass A from w_g = digamma(N_g) - 1 - u_g (k_redfin) and e_g:
// centring constant kappa, {e_g, w_g - kappa} pairs, V1c = sum e*s0, V2c = sum e*s0^2 with
// s0_g = (1-a)*logzi + w_g - kappa.  No transcendentals.
__device__ inline void prepA_block(Scalars *sc, double a, int G, const double *w, const double *e,
                                   double2 *ew, double *sh) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const double oma = 1.0 - a, logzi = sc->logzi;
  double su = 0.0, sv = 0.0;
  for (int g = tid; g < G; g += nt) {
    const double eg = e[g];
    su += eg;
    sv += eg * (oma * logzi + w[g]);
  }
  const double U = block_sum(su, sh);
  const double V1 = block_sum(sv, sh);
  const double kappa = V1 / U;  // a per-EC shift leaves the variance unchanged
  double s1 = 0.0, s2 = 0.0;
  for (int g = tid; g < G; g += nt) {
    const double wcg = w[g] - kappa;
    const double eg = e[g];
    ew[g] = make_double2(eg, wcg);
    const double s0 = oma * logzi + wcg;
    s1 += eg * s0;
    s2 += eg * s0 * s0;
  }
  const double V1c = block_sum(s1, sh);
  const double V2c = block_sum(s2, sh);
  if (tid == 0) {
    sc->V1c = V1c;
    sc->V2c = V2c;
  }
}

// Fletcher-Reeves step (rcgpar rcg_optl_mat: beta_FR, oldstep scaling, gamma += step) on
// the (a, u) state, followed by the pass-B preparation.
__global__ __launch_bounds__(1024) void k_step(Scalars *sc, int G, int n_lut, int n_partA,
                                              const double *partA, const double *w, double *u,
                                              double *os_u, double *step_u, const double *lut,
                                              double *e, double *X) {
  __shared__ double sh[32];
  if (sc->done) return;
  const int tid = threadIdx.x, nt = blockDim.x;
  const double a = sc->a, oldnorm = sc->oldnorm, bound = sc->bound;
  double os_a = sc->os_a;
  const int didreset = sc->didreset;
  double pn = 0.0;
  for (int i = tid; i < n_partA; i += nt) pn += partA[i];
  const double newnorm = block_sum(pn, sh);
  const double beta = newnorm / oldnorm;
  double step_a = 1.0 - a;
  if (didreset) {
    os_a *= 0.0;
  } else if (beta > 0) {
    os_a *= beta;
    step_a += os_a;
  }
  for (int g = tid; g < G; g += nt) {
    double osu = os_u[g], su = w[g];
    if (didreset) {
      osu *= 0.0;
    } else if (beta > 0) {
      osu *= beta;
      su += osu;
    }
    os_u[g] = osu;
    step_u[g] = su;
    u[g] += su;
  }
  const double a_new = a + step_a;
  __syncthreads();
  if (tid == 0) {
    sc->a = a_new;
    sc->os_a = os_a;
    sc->step_a = step_a;
    sc->oldnorm = newnorm;
    sc->newnorm = newnorm;
    sc->beta = beta;
    sc->didreset = 0;
    sc->oldbound = bound;
  }
  if (sc->flavor == 0) prepB_block(sc, a_new, G, n_lut, u, lut, e, X, sh);
}

// Column sums across workgroups (fixed order) fused with the per-group math that follows them:
// Nc_g, N_g, lgamma(N_g), (M - u_g) * Nc_g and w_g = digamma(N_g) - 1 - u_g.  One workgroup per
// 64 groups so that the lgamma / digamma evaluations spread over ~G/64 CUs.
//   nblk > 0: sum partAcc[b*G + g] over b;  nblk == 0: Acc already holds the totals.
__global__ __launch_bounds__(256) void k_redfin(const Scalars *sc, int cond_reset, int G, int nblk,
                                               int npartS, const double *partAcc, const double *Acc,
                                               const double *partS, const double *e, const double *u,
                                               const double *alpha0, double *Nc, double *N, double *w,
                                               double *partR) {
  __shared__ double sh[32];
  __shared__ double accs[4][64];
  if (sc->done) return;
  if (cond_reset && !sc->reset_pending) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int g = blockIdx.x * 64 + lane;
  // W = sum_j r_j : every workgroup forms it in the same fixed order
  double pw = 0.0;
  for (int b = tid; b < npartS; b += 256) pw += partS[4 * b + 2];
  const double W = block_sum(pw, sh);
  double s = 0.0;
  if (g < G) {
    if (nblk > 0) {
      for (int b = wv; b < nblk; b += 4) s += partAcc[(size_t)b * G + g];
    } else if (wv == 0) {
      s = Acc[g];
    }
  }
  accs[wv][lane] = s;
  __syncthreads();
  double lgv = 0.0, muv = 0.0;
  if (wv == 0 && g < G) {
    const double A = ((accs[0][lane] + accs[1][lane]) + accs[2][lane]) + accs[3][lane];
    double nc;
    const double ug = u[g];
    if (sc->flavor == 0) {
      nc = e[g] * (sc->p0 * W + A);
      muv = (sc->M - ug) * nc;
    } else {
      nc = A;
    }
    const double n = alpha0[g] + nc;
    Nc[g] = nc;
    N[g] = n;
    lgv = lgamma(n);
    w[g] = digamma_ref(n) - 1.0 - ug;
  }
  if (wv == 0) {
    lgv = wave_sum(lgv);
    muv = wave_sum(muv);
    if (lane == 0) {
      partR[2 * blockIdx.x] = lgv;
      partR[2 * blockIdx.x + 1] = muv;
    }
  }
}

// ELBO (rcgpar ELBO_rcg_mat + bound_const), the bound < oldbound steepest-descent retry
// (revert_step), the convergence test and the gradient preparation for the next pass A.
//   mode 2: initial update_N_k only;  mode 0: first evaluation of an iteration;
//   mode 1: re-evaluation after a reset (runs only when reset_pending).
__global__ __launch_bounds__(1024) void k_fin(Scalars *sc, int mode, int G, int n_lut, int npartS,
                                             int npartR, const double *partS, const double *partR,
                                             const double *Nc, const double *w, double *u,
                                             double *os_u, const double *step_u, const double *lut,
                                             double *e, double *X, double2 *ew, TraceDev tr) {
  __shared__ double sh[32];
  if (sc->done) return;
  if (mode == 1 && !sc->reset_pending) return;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int flavor = sc->flavor;
  const double a = sc->a, oldbound = sc->oldbound;
  const double beta = sc->beta, tol = sc->tol, csum = sc->csum;
  double p1 = 0.0, p2 = 0.0, p3 = 0.0, p4 = 0.0;
  for (int b = tid; b < npartS; b += nt) {
    p1 += partS[4 * b];
    p2 += partS[4 * b + 1];
  }
  for (int b = tid; b < npartR; b += nt) {
    p3 += partR[2 * b];
    p4 += partR[2 * b + 1];
  }
  const double s_clogZ = block_sum(p1, sh);
  const double s_rH = block_sum(p2, sh);
  const double lg = block_sum(p3, sh);
  const double mu = block_sum(p4, sh);
  if (mode == 2) {
    if (flavor == 0) prepA_block(sc, a, G, w, e, ew, sh);
    return;
  }
  const double coef = (flavor == 0) ? (1.0 - a) : 1.0;
  const double bound = sc->bound_const + s_clogZ + coef * s_rH + mu + lg;
  int didreset = sc->didreset;
  __syncthreads();
  if (mode == 0 && bound < oldbound) {
    // bad step: revert to steepest descent (gamma += oldm; gamma -= oldstep) and re-evaluate
    double a2 = a;
    if (beta > 0) {
      a2 = a - sc->os_a;
      for (int g = tid; g < G; g += nt) u[g] -= os_u[g];
    }
    __syncthreads();
    if (tid == 0) {
      sc->a = a2;
      sc->didreset = 1;
      sc->reset_pending = 1;
      sc->bound = bound;
    }
    if (flavor == 0) prepB_block(sc, a2, G, n_lut, u, lut, e, X, sh);
    return;
  }
  if (mode == 0) {
    // oldstep = step
    for (int g = tid; g < G; g += nt) os_u[g] = step_u[g];
  }
  const int it = sc->iter;
  if (it < sc->trace_theta && tr.theta) {
    for (int g = tid; g < G; g += nt) tr.theta[(size_t)it * G + g] = Nc[g] / csum;
  }
  int done = 0;
  if (!sc->fixed_iters && (bound - oldbound < tol) && !didreset) done = 1;
  if (it + 1 >= sc->max_iters) done = 1;
  __syncthreads();
  if (tid == 0) {
    if (mode == 0) sc->os_a = sc->step_a;
    sc->bound = bound;
    sc->reset_pending = 0;
    if (it < kMaxTrace) {
      tr.bound[it] = bound;
      tr.newnorm[it] = sc->newnorm;
      tr.beta[it] = beta;
      tr.didreset[it] = didreset;
    }
    sc->iter = it + 1;
    sc->done = done;
  }
  if (!done && flavor == 0) prepA_block(sc, a, G, w, e, ew, sh);
}

// ---------------------------------------------------------------------------------------
// Solve set-up: c_j = exp(logc_j) (or the bootstrap counts) gathered into the permuted EC
// order, sum of counts, bound constant (rcgpar calc_bound_const), initial gamma = log(1/G).
// perm == nullptr: identity (dense flavour).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cvec_from_logc(const double *logc, const uint32_t *perm,
                                                       uint32_t E, double *cvec, double *part) {
  __shared__ double sh[32];
  double s = 0.0;
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < E; j += gridDim.x * blockDim.x) {
    const double c = exp(logc[perm ? perm[j] : j]);
    cvec[j] = c;
    s += c;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_cvec_from_counts(const uint32_t *cnt, const uint32_t *perm,
                                                         uint32_t E, double *cvec, double *part) {
  __shared__ double sh[32];
  double s = 0.0;
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < E; j += gridDim.x * blockDim.x) {
    const double c = (double)cnt[perm ? perm[j] : j];
    cvec[j] = c;
    s += c;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(1024) void k_init_state(Scalars *sc, int G, int npart, const double *part,
                                                    const double *alpha0, double *u, double *os_u,
                                                    double *step_u, double tol, int max_iters,
                                                    int fixed_iters, int trace_theta, int flavor,
                                                    double logzi, double init_bound) {
  __shared__ double sh[32];
  const int tid = threadIdx.x, nt = blockDim.x;
  double s = 0.0;
  for (int i = tid; i < npart; i += nt) s += part[i];
  const double csum = block_sum(s, sh);
  double sa = 0.0, sl = 0.0;
  for (int g = tid; g < G; g += nt) {
    sa += alpha0[g];
    sl += lgamma(alpha0[g]);
    u[g] = 0.0;
    os_u[g] = 0.0;
    step_u[g] = 0.0;
  }
  sa = block_sum(sa, sh);
  sl = block_sum(sl, sh);
  if (tid == 0) {
    Scalars z = {};
    z.a = 0.0;
    z.oldnorm = 1.0;
    z.bound = init_bound;
    z.oldbound = init_bound;
    z.bound_const = lgamma(sa) - lgamma(sa + csum) - sl;
    z.tol = tol;
    z.csum = csum;
    z.logzi = logzi;
    z.max_iters = max_iters;
    z.fixed_iters = fixed_iters;
    z.trace_theta = trace_theta;
    z.flavor = flavor;
    *sc = z;
  }
}

}  // namespace msw
