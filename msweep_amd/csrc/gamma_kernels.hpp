// gamma_kernels.hpp -- materialisation of gamma / the dense likelihood (utility, not timed).
#pragma once
#include "device_util.hpp"
#include "sell.hpp"

namespace msw {

// ---------------------------------------------------------------------------------------
// gamma materialisation (K6): gamma(g, j) = a*L(g, j) + u_g - lse_j, rows = groups, columns in
// the ORIGINAL EC order, any block of ECs.  With (a, u, lse) = (1, 0, none) the same kernels expand the
// resident likelihood.  Utility kernels, not on the timed path.
// ---------------------------------------------------------------------------------------
// iperm[original EC index] = permuted position (built once per likelihood, on first use)
__global__ __launch_bounds__(256) void k_invert_perm(const uint32_t *perm, uint32_t E, uint32_t *iperm) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < E) iperm[perm[p]] = p;
}

// gamma (or, with (a, u, lse) = (1, 0, none), the likelihood itself) of the ECs [e0, e1) in the ORIGINAL EC
// order: out[g * ld + (j - e0)], all groups.  A thread per EC: its normaliser lse_j, the background value
// of every group, then its listed cells.  What msw_core_gamma_block serves --write-probs / the binning input
// from, block by block (src/Sample.cpp:63-85, src/mSWEEP.cpp:437-469), without a G x E buffer anywhere.
template <int ENC>
__global__ __launch_bounds__(256) void k_gamma_block(SellDev S, const uint32_t *iperm, uint32_t e0, uint32_t e1,
                                                    double a, double logzi, double tref, const double *u,
                                                    const double *lut, int normalise, double *out, size_t ld) {
  __shared__ double sh[32];
  const int tid = threadIdx.x;
  double M = 0.0, U = 0.0;
  if (normalise) {
    double m = -INFINITY;
    for (uint32_t g = tid; g < S.n_groups; g += blockDim.x) m = fmax(m, u[g]);
    M = block_max(m, sh);
    double su = 0.0;
    for (uint32_t g = tid; g < S.n_groups; g += blockDim.x) su += exp(u[g] - M);
    U = block_sum(su, sh);
  }
  // every exp(a T) relative to tref (the table value with the largest a T, log zi included): <= 1
  const double p0 = exp(a * (logzi - tref));
  for (uint32_t j = e0 + blockIdx.x * blockDim.x + tid; j < e1; j += gridDim.x * blockDim.x) {
    const uint32_t p = iperm[j];
    double lse = 0.0;
    if (normalise) {
      double zs = 0.0;
      for_each_cell<ENC>(S, p, [&](uint32_t g, double T) { zs += exp(u[g] - M) * (exp(a * (T - tref)) - p0); });
      double Z = p0 * U + zs;
      if (!(Z >= p0 * U * kGuardRatio)) {  // guarded EC (sell.hpp): every group visited instead
        Z = 0.0;
        for (uint32_t g = 0; g < S.n_groups; ++g) {
          double xg = p0;
          for_each_cell<ENC>(S, p, [&](uint32_t gg, double T) { if (gg == g) xg = exp(a * (T - tref)); });
          Z += exp(u[g] - M) * xg;
        }
      }
      lse = M + log(Z) + a * tref;
    }
    double *col = out + (j - e0);
    for (uint32_t g = 0; g < S.n_groups; ++g) col[(size_t)g * ld] = a * logzi + u[g] - lse;
    for_each_cell<ENC>(S, p, [&](uint32_t g, double T) { col[(size_t)g * ld] = a * T + u[g] - lse; });
  }
}

// dense flavour: gamma from Lt (EC-major) -> rows = groups slab [g_begin, g_end)
// (ECs [e0, e0 + E) of the matrix, written to columns 0 .. E - 1 of out)
__global__ __launch_bounds__(256) void k_gamma_dense(const double *Lt, int G, uint32_t E, double a,
                                                    const double *u, int sub_lse, double *out,
                                                    size_t ld, int g_begin, int g_end, uint32_t e0) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= E) return;
  const double *row = Lt + ((size_t)e0 + j) * G;
  double lse = 0.0;
  if (sub_lse) {
    double m = -INFINITY;
    for (int g = 0; g < G; ++g) m = fmax(m, a * row[g] + u[g]);
    double Z = 0.0;
    for (int g = 0; g < G; ++g) Z += exp(a * row[g] + u[g] - m);
    lse = m + log(Z);
  }
  for (int g = g_begin; g < g_end; ++g) out[(size_t)(g - g_begin) * ld + j] = a * row[g] + u[g] - lse;
}

}  // namespace msw
