// em_kernels.hpp -- EM variant behind `--algorithm emgpu` (reference call site
// src/mSWEEP.cpp:200-203, rcgpar::em_torch [UPSTREAM-UNVERIFIED, restated in oracle/rcg_oracle.cpp]).
// An EM iteration is ONE pass-B sweep with (a, u) = (1, log theta): the same kernel that serves
// RCG computes the E-step responsibilities' column sums and the weighted log-likelihood; this
// file only holds the O(G) M-step.
#pragma once
#include "device_util.hpp"
#include "state_kernels.hpp"

namespace msw {

__global__ __launch_bounds__(1024) void k_em_init(Scalars *sc, int G, int n_lut, double *u,
                                                 const double *lut, double *e, TabDev X) {
  __shared__ double sh[32];
  const int tid = threadIdx.x, nt = blockDim.x;
  const double l = log(1.0 / (double)G);
  for (int g = tid; g < G; g += nt) u[g] = l;
  __syncthreads();
  if (tid == 0) {
    sc->a = 1.0;
    sc->bound = -INFINITY;
  }
  __syncthreads();
  if (sc->flavor == 0) prepB_block(sc, 1.0, G, n_lut, u, lut, e, X, sh);
}

// M-step: theta_g = max(0, (Nc_g + alpha_g - 1) / (sum c + sum(alpha - 1))), u = log theta;
// stop when the weighted log-likelihood gain drops below tol (after the first iteration).
__global__ __launch_bounds__(1024) void k_em_fin(Scalars *sc, int G, int n_lut, int npartS,
                                                const double *partS, const double *Nc,
                                                const double *alpha0, double *u, double *theta,
                                                const double *lut, double *e, TabDev X, TraceDev tr) {
  __shared__ double sh[32];
  if (sc->done) return;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int flavor = sc->flavor;
  const double csum = sc->csum, oldll = sc->bound, tol = sc->tol, M = sc->M;
  double p1 = 0.0, p2 = 0.0;
  for (int b = tid; b < npartS; b += nt) {
    p1 += partS[4 * b];
    p2 += partS[4 * b + 1];
  }
  const double s_clogZ = block_sum(p1, sh);
  const double s_rH = block_sum(p2, sh);
  double sa = 0.0, su = 0.0;
  for (int g = tid; g < G; g += nt) {
    sa += alpha0[g] - 1.0;
    const double nc = Nc[g];
    if (flavor != 0 && nc != 0.0) su += u[g] * nc;
  }
  sa = block_sum(sa, sh);
  su = block_sum(su, sh);
  const double ll = (flavor == 0) ? s_clogZ + (M + sc->tref) * csum : s_clogZ + s_rH + su;  // a = 1: Z carries exp(-tref)
  const double denom = csum + sa;
  for (int g = tid; g < G; g += nt) {
    double t = (Nc[g] + alpha0[g] - 1.0) / denom;
    t = t > 0.0 ? t : 0.0;
    theta[g] = t;
    u[g] = log(t);
  }
  const int it = sc->iter;
  int done = 0;
  if (!sc->fixed_iters && it > 0 && (ll - oldll < tol)) done = 1;
  if (it + 1 >= sc->max_iters) done = 1;
  if (it < sc->trace_theta && tr.theta)
    for (int g = tid; g < G; g += nt) tr.theta[(size_t)it * G + g] = theta[g];
  __syncthreads();
  if (tid == 0) {
    sc->oldbound = oldll;
    sc->bound = ll;
    if (it < kMaxTrace) {
      tr.bound[it] = ll;
      tr.newnorm[it] = 0.0;
      tr.beta[it] = 0.0;
      tr.didreset[it] = 0;
    }
    sc->iter = it + 1;
    sc->done = done;
  }
  if (!done && flavor == 0) prepB_block(sc, 1.0, G, n_lut, u, lut, e, X, sh);
}

}  // namespace msw
