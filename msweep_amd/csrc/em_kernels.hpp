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
// One workgroup between two sweeps, organised like k_step: every load issued up front, the scalar state read
// once, one pair of barriers for the four sums, u from the logarithm to e_g = exp(u - M) in registers; a = 1
// throughout, so the per-slot tables k_em_init built stay (only M, U and e change).
__global__ __launch_bounds__(1024) void k_em_fin(Scalars *sc, int G, int n_lut, int npartS,
                                                const double *partS, const double *Nc,
                                                const double *alpha0, double *u, double *theta,
                                                const double *lut, double *e, TabDev X, TraceDev tr, float *e32) {
  // e32 != nullptr: --emprecision float served by the fp32 sweep (em_f32_kernels.hpp): the log-likelihood is rounded
  // to float before the stop rule sees it, theta, log theta and e_g are floats (kept as doubles of float value where
  // fp64 kernels read them: k_redfin's e_g, the gamma kernels' u), e32 receives e_g for the next sweep
  const bool f32 = e32 != nullptr;
  __shared__ double sh[16 * 4];
  const int tid = threadIdx.x, nt = blockDim.x;
  double q[4] = {0.0, 0.0, 0.0, 0.0};  // sum c log Z, sum r H, sum (alpha - 1), sum u Nc
  for (int b = tid; b < npartS; b += nt) {
    q[0] += partS[4 * b];
    q[1] += partS[4 * b + 1];
  }
  const bool inreg = G <= kStepRegs * nt;
  double ncv[kStepRegs], alv[kStepRegs], uv[kStepRegs];
  if (inreg) {
#pragma unroll
    for (int k = 0; k < kStepRegs; ++k) {
      const int g = tid + k * nt;
      if (g < G) {
        ncv[k] = Nc[g];
        alv[k] = alpha0[g];
        uv[k] = u[g];
      }
    }
  }
  const Scalars s0 = *sc;
  if (s0.done) return;
  const int flavor = s0.flavor;
  const double csum = s0.csum, oldll = s0.bound, tol = s0.tol;
  // variants (msw_core_set_option): ML instead of MAP -- the prior's pseudo-counts dropped; stop on the largest
  // move of a weight instead of the log-likelihood gain; the rule tested on a grid of iterations
  const bool ml = s0.em_prior == 1, stop_theta = s0.em_stop == 1;
  if (inreg) {
#pragma unroll
    for (int k = 0; k < kStepRegs; ++k) {
      if (tid + k * nt < G) {
        if (!ml) q[2] += alv[k] - 1.0;
        if (flavor != 0 && ncv[k] != 0.0) q[3] += uv[k] * ncv[k];
      }
    }
  } else {
    for (int g = tid; g < G; g += nt) {
      if (!ml) q[2] += alpha0[g] - 1.0;
      const double nc = Nc[g];
      if (flavor != 0 && nc != 0.0) q[3] += u[g] * nc;
    }
  }
  block_sum_n<4>(q, sh);
  const double s_clogZ = q[0], s_rH = q[1], sa = q[2], su = q[3];
  const double ll_raw = (flavor == 0) ? s_clogZ + (s0.M + s0.tref) * csum : s_clogZ + s_rH + su;  // a = 1: Z carries exp(-tref)
  const double ll = f32 ? (double)(float)ll_raw : ll_raw;
  const double denom = csum + sa;
  const int it = s0.iter;
  int done = 0;
  const bool grid = s0.check_every <= 1 || (it + 1) % s0.check_every == 0;
  if (!s0.fixed_iters && it > 0 && !stop_theta && (ll - oldll < tol) && grid) done = 1;
  if (it + 1 >= s0.max_iters) done = 1;
  const bool trace = it < s0.trace_theta && tr.theta;
  double m = -INFINITY, dmax = 0.0;
  const double t_first = 1.0 / (double)G;  // the weights before the first M-step
  if (inreg) {
#pragma unroll
    for (int k = 0; k < kStepRegs; ++k) {
      const int g = tid + k * nt;
      if (g < G) {
        double t = (ml ? ncv[k] : ncv[k] + alv[k] - 1.0) / denom;
        t = t > 0.0 ? t : 0.0;
        if (f32) t = (double)(float)t;
        if (stop_theta) dmax = fmax(dmax, fabs(t - (it > 0 ? theta[g] : t_first)));
        theta[g] = t;
        if (trace) tr.theta[(size_t)it * G + g] = t;
        uv[k] = f32 ? (double)logf((float)t) : log(t);
        u[g] = uv[k];
        m = fmax(m, uv[k]);
      }
    }
  } else {
    for (int g = tid; g < G; g += nt) {
      double t = (ml ? Nc[g] : Nc[g] + alpha0[g] - 1.0) / denom;
      t = t > 0.0 ? t : 0.0;
      if (stop_theta) dmax = fmax(dmax, fabs(t - (it > 0 ? theta[g] : t_first)));
      theta[g] = t;
      if (trace) tr.theta[(size_t)it * G + g] = t;
      u[g] = log(t);
    }
  }
  if (stop_theta) {  // wave-uniform: the largest move of a weight decides
    dmax = block_max(dmax, sh);
    if (!s0.fixed_iters && it > 0 && dmax < tol && grid) done = 1;
  }
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
  if (done || flavor != 0) return;
  if (!inreg) {  // many groups: the looping form (it rebuilds the tables too, to the same values)
    __syncthreads();
    prepB_block(sc, 1.0, G, n_lut, u, lut, e, X, sh);
    return;
  }
  const double M = block_max(m, sh);
  double se = 0.0;
#pragma unroll
  for (int k = 0; k < kStepRegs; ++k) {
    const int g = tid + k * nt;
    if (g < G) {
      double eg = flush_denormal(exp(uv[k] - M));
      if (f32) {
        const float ef = (float)eg;
        e32[g] = ef;
        eg = (double)ef;
      }
      e[g] = eg;
      se += eg;
    }
  }
  const double U = block_sum(se, sh);
  if (tid == 0) {
    sc->M = M;
    sc->U = U;
  }
}

}  // namespace msw
