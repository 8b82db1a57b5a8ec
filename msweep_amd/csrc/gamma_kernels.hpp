// gamma_kernels.hpp -- materialisation of gamma / the dense likelihood (utility, not timed).
#pragma once
#include "device_util.hpp"
#include "sell.hpp"

namespace msw {

// ---------------------------------------------------------------------------------------
// gamma materialisation (K6): gamma(g, j) = a*L(g, j) + u_g - lse_j, rows = groups, columns in
// the ORIGINAL EC order.  With (a, u, lse) = (1, 0, none) the same kernels expand the resident
// likelihood.  Utility kernels, not on the timed path.
// ---------------------------------------------------------------------------------------
template <bool WIDE>
__global__ __launch_bounds__(256) void k_lse_sell(SellDev S, double a, double logzi, const double *u,
                                                 const double *lut, double *lse /*original order*/) {
  __shared__ double sh[32];
  const int tid = threadIdx.x;
  double m = -INFINITY;
  for (uint32_t g = tid; g < S.n_groups; g += blockDim.x) m = fmax(m, u[g]);
  const double M = block_max(m, sh);
  double su = 0.0;
  for (uint32_t g = tid; g < S.n_groups; g += blockDim.x) su += exp(u[g] - M);
  const double U = block_sum(su, sh);
  const double p0 = exp(a * logzi);
  for (uint32_t p = blockIdx.x * blockDim.x + tid; p < S.n_ecs; p += gridDim.x * blockDim.x) {
    double zs = 0.0;
    for_each_cell<WIDE>(S, p, [&](uint32_t g, uint32_t i) { zs += exp(u[g] - M) * (exp(a * lut[i]) - p0); });
    double Z = p0 * U + zs;
    if (!(Z >= p0 * U * kGuardRatio)) {
      // guarded EC (sell.hpp): background and listed cells cancel -- every group visited instead
      // (a thread per EC scanning its own cells for each group: utility kernel, rare path)
      Z = 0.0;
      for (uint32_t g = 0; g < S.n_groups; ++g) {
        double xg = p0;
        for_each_cell<WIDE>(S, p, [&](uint32_t gg, uint32_t i) { if (gg == g) xg = exp(a * lut[i]); });
        Z += exp(u[g] - M) * xg;
      }
    }
    lse[S.perm[p]] = M + log(Z);
  }
}

__global__ __launch_bounds__(256) void k_gamma_fill(double *out, size_t ld, int g_begin, int g_end,
                                                   uint32_t E, double a, double logzi,
                                                   const double *u, const double *lse) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= E) return;
  const double l = lse ? lse[j] : 0.0;
  for (int g = g_begin; g < g_end; ++g)
    out[(size_t)(g - g_begin) * ld + j] = a * logzi + u[g] - l;
}

template <bool WIDE>
__global__ __launch_bounds__(256) void k_gamma_scatter(SellDev S, double *out, size_t ld, int g_begin,
                                                      int g_end, double a, const double *u,
                                                      const double *lut, const double *lse) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= S.n_ecs) return;
  const uint32_t j = S.perm[p];
  const double l = lse ? lse[j] : 0.0;
  for_each_cell<WIDE>(S, p, [&](uint32_t gu, uint32_t i) {
    const int g = (int)gu;
    if (g >= g_begin && g < g_end) out[(size_t)(g - g_begin) * ld + j] = a * lut[i] + u[g] - l;
  });
}

// dense flavour: gamma from Lt (EC-major) -> rows = groups slab [g_begin, g_end)
__global__ __launch_bounds__(256) void k_gamma_dense(const double *Lt, int G, uint32_t E, double a,
                                                    const double *u, int sub_lse, double *out,
                                                    size_t ld, int g_begin, int g_end) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= E) return;
  const double *row = Lt + (size_t)j * G;
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
