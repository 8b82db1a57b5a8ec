// dense_kernels.hpp -- sweeps for a dense likelihood (--read-likelihood path) and the transpose.
#pragma once
#include <type_traits>

#include "device_util.hpp"

namespace msw {

// ---------------------------------------------------------------------------------------
// Dense-L kernels.  L is kept EC-major on the device (Lt[j*G + g]) so that a wavefront
// streams one EC's G values with coalesced loads; lane l owns groups l, l+64, ... and
// keeps their u_g / w_g / column-sum accumulators in registers (no atomics).
// ---------------------------------------------------------------------------------------
// Round 5 (the review's weak 8: 0.54 / 0.41 of the HBM peak at 1 M x 500, one fp64 exp per cell): (i) the NEXT EC's row
// is in flight while the current one is worked on -- a wavefront used to issue its row's loads, wait out the HBM
// latency, then spend as long again in reductions and exponentials with nothing in flight; (ii) the per-EC maximum --
// any common offset near the maximum serves the softmax -- is reduced as a FLOAT (one register per DPP step instead
// of two, and a third of the instructions of the fp64 tree); the value every lane ends with is the same float, so
// the arithmetic stays identical across the lanes of an EC.
__device__ __forceinline__ float wave_max_f(float v) {
  const float ninf = -INFINITY;
  auto mv = [](float x, float fill, auto ctrl, auto mask) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(x), decltype(ctrl)::value,
                                                      decltype(mask)::value, 0xf, false));
  };
  v = fmaxf(v, mv(v, ninf, std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{}));
  v = fmaxf(v, mv(v, ninf, std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{}));
  v = fmaxf(v, mv(v, ninf, std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{}));
  v = fmaxf(v, mv(v, ninf, std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{}));
  v = fmaxf(v, mv(v, ninf, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{}));
  v = fmaxf(v, mv(v, ninf, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{}));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// the row of EC j into x (lanes past G: 0)
template <int NREG>
__device__ __forceinline__ void dense_load_row(const double *Lt, uint32_t j, int G, int lane, double (&x)[NREG]) {
  const double *row = Lt + (size_t)j * G;
#pragma unroll
  for (int i = 0; i < NREG; ++i) {
    const int g = lane + 64 * i;
    x[i] = g < G ? row[g] : 0.0;
  }
}

template <int NREG>
__global__ __launch_bounds__(256) void k_dense_passA(const Scalars *sc, const double *Lt, int G,
                                                    uint32_t E, const double *u, const double *w,
                                                    double *partA) {
  __shared__ double sh[32];
  if (sc->done) return;  // (pass A runs before the verdict on the state it sweeps: k_finstep)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t gw = blockIdx.x * 4 + wv, nw = gridDim.x * 4;
  const double a = sc->a, oma = 1.0 - a;
  double uu[NREG], ww[NREG];
#pragma unroll
  for (int i = 0; i < NREG; ++i) {
    const int g = lane + 64 * i;
    uu[i] = g < G ? u[g] : -INFINITY;  // (lanes past G: exp(-inf) = 0 in every sum)
    ww[i] = g < G ? w[g] : 0.0;
  }
  double nn = 0.0;
  double xn[NREG];
  if (gw < E) dense_load_row<NREG>(Lt, gw, G, lane, xn);
  for (uint32_t j = gw; j < E; j += nw) {
    double x[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) x[i] = xn[i];
    if (j + nw < E) dense_load_row<NREG>(Lt, j + nw, G, lane, xn);  // in flight under this EC's arithmetic
    double mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NREG; ++i) mx = fmax(mx, fma(a, x[i], uu[i]));
    const double m = (double)wave_max_f((float)mx);
    double Z = 0.0, S1 = 0.0, pe[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      pe[i] = exp(fma(a, x[i], uu[i]) - m);
      Z += pe[i];
      S1 = fma(pe[i], fma(oma, x[i], ww[i]), S1);
    }
    Z = wave_sum(Z);
    S1 = wave_sum(S1);
    const double iZ = 1.0 / Z, sbar = S1 * iZ;
    // (the variance about the mean, not from two moments: near convergence it is ten orders below the mean's square)
    double v = 0.0;
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      const double d = fma(oma, x[i], ww[i]) - sbar;
      v = fma(pe[i] * d, d, v);
    }
    v = wave_sum(v);
    nn += v * iZ;
  }
  // every lane of a wave holds the same nn; take lane 0 of each wave
  double t = (lane == 0) ? nn : 0.0;
  t = block_sum(t, sh);
  if (threadIdx.x == 0) partA[blockIdx.x] = t;
}

template <int NREG>
__global__ __launch_bounds__(256) void k_dense_passB(const Scalars *sc, const double *Lt, int G, uint32_t E,
                                                    const double *cvec, const double *u,
                                                    double *partAcc, double *partS) {
  extern __shared__ __align__(16) unsigned char smem[];
  double *sh = reinterpret_cast<double *>(smem);
  double *accl = sh + 32;  // [4][G]
  if (sc->done) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t gw = blockIdx.x * 4 + wv, nw = gridDim.x * 4;
  const double a = sc->a;
  double uu[NREG], acc[NREG];
#pragma unroll
  for (int i = 0; i < NREG; ++i) {
    const int g = lane + 64 * i;
    uu[i] = g < G ? u[g] : -INFINITY;
    acc[i] = 0.0;
  }
  double s_clogZ = 0.0, s_rH = 0.0;
  double xn[NREG], cn = 0.0;
  if (gw < E) {
    dense_load_row<NREG>(Lt, gw, G, lane, xn);
    cn = cvec[gw];
  }
  for (uint32_t j = gw; j < E; j += nw) {
    double x[NREG];
#pragma unroll
    for (int i = 0; i < NREG; ++i) x[i] = xn[i];
    const double c = cn;
    if (j + nw < E) {  // the next EC's row in flight under this one's arithmetic
      dense_load_row<NREG>(Lt, j + nw, G, lane, xn);
      cn = cvec[j + nw];
    }
    double mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NREG; ++i) mx = fmax(mx, fma(a, x[i], uu[i]));
    const double m = (double)wave_max_f((float)mx);
    double Z = 0.0, hs = 0.0;
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      const double y = fma(a, x[i], uu[i]) - m;
      const double pe = exp(y);
      hs += pe != 0.0 ? pe * (x[i] - y) : 0.0;  // (lanes past G, underflowed cells: y = -inf)
      x[i] = pe;
      Z += pe;
    }
    Z = wave_sum(Z);
    hs = wave_sum(hs);
    if (c != 0.0) {
      const double rj = c / Z;
      s_clogZ += c * log(Z);
      s_rH += rj * hs;
#pragma unroll
      for (int i = 0; i < NREG; ++i) acc[i] = fma(rj, x[i], acc[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < NREG; ++i) {
    const int g = lane + 64 * i;
    if (g < G) accl[wv * G + g] = acc[i];
  }
  double t1 = (lane == 0) ? s_clogZ : 0.0, t2 = (lane == 0) ? s_rH : 0.0;
  t1 = block_sum(t1, sh);
  t2 = block_sum(t2, sh);
  if (threadIdx.x == 0) {
    partS[4 * blockIdx.x + 0] = t1;
    partS[4 * blockIdx.x + 1] = t2;
    partS[4 * blockIdx.x + 2] = 0.0;
    partS[4 * blockIdx.x + 3] = 0.0;
  }
  __syncthreads();
  double *dst = partAcc + (size_t)blockIdx.x * G;
  for (int g = threadIdx.x; g < G; g += blockDim.x)
    dst[g] = ((accl[g] + accl[G + g]) + accl[2 * G + g]) + accl[3 * G + g];
}

// [G][E] (ld) -> [E][G] transpose through LDS, 64 x 64 tiles, 256 threads.
__global__ __launch_bounds__(256) void k_transpose(const double *src, size_t ld, int G, uint32_t E,
                                                  double *dst) {
  __shared__ double tile[64][65];
  const uint32_t j0 = blockIdx.x * 64;
  const int g0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int g = g0 + r;
    const uint32_t j = j0 + tx;
    tile[r][tx] = (g < G && j < E) ? src[(size_t)g * ld + j] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const uint32_t j = j0 + r;
    const int g = g0 + tx;
    if (g < G && j < E) dst[(size_t)j * G + g] = tile[tx][r];
  }
}


}  // namespace msw

// ---------------------------------------------------------------------------------------
// Dense sweeps for 1024 < G <= 8192 groups: an EC's G values no longer fit a wavefront's
// registers, so each EC is swept twice (the second read is served by L2): an online-softmax
// sweep for the row statistics, then a sweep that adds the normalised responsibilities into the
// lane-owned column-sum registers.  u / w are read through L2 (shared by all waves).
// ---------------------------------------------------------------------------------------
namespace msw {

template <int NREG>
__global__ __launch_bounds__(256) void k_dense_big_passA(const Scalars *sc, const double *Lt, int G,
                                                        uint32_t E, const double *u, const double *w,
                                                        double *partA) {
  __shared__ double sh[32];
  if (sc->done) return;  // (pass A runs before the verdict on the state it sweeps: k_finstep)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t gw = blockIdx.x * 4 + wv, nw = gridDim.x * 4;
  const double a = sc->a, oma = 1.0 - a;
  double nn = 0.0;
  for (uint32_t j = gw; j < E; j += nw) {
    const double *row = Lt + (size_t)j * G;
    // sweep 1: running max m, S0 = sum exp(y - m), S1 = sum exp(y - m) * s
    double m = -INFINITY, S0 = 0.0, S1 = 0.0;
#pragma unroll 4
    for (int i = 0; i < NREG; ++i) {
      const int g = lane + 64 * i;
      if (g < G) {
        const double x = row[g];
        const double y = a * x + u[g], s = oma * x + w[g];
        if (y > m) {
          const double f = exp(m - y);
          S0 = S0 * f + 1.0;
          S1 = S1 * f + s;
          m = y;
        } else {
          const double p = exp(y - m);
          S0 += p;
          S1 += p * s;
        }
      }
    }
    const double mw = wave_max(m);
    const double f = (m == -INFINITY) ? 0.0 : exp(m - mw);
    const double Z = wave_sum(S0 * f), sbar = wave_sum(S1 * f) / Z;
    // sweep 2: variance about the mean
    double v = 0.0;
#pragma unroll 4
    for (int i = 0; i < NREG; ++i) {
      const int g = lane + 64 * i;
      if (g < G) {
        const double x = row[g];
        const double d = oma * x + w[g] - sbar;
        v += exp(a * x + u[g] - mw) * d * d;
      }
    }
    nn += wave_sum(v) / Z;
  }
  double t = (lane == 0) ? nn : 0.0;
  t = block_sum(t, sh);
  if (threadIdx.x == 0) partA[blockIdx.x] = t;
}

template <int NREG>
__global__ __launch_bounds__(256) void k_dense_big_passB(const Scalars *sc, const double *Lt, int G, uint32_t E,
                                                        const double *cvec, const double *u,
                                                        double *partAcc, double *partS) {
  __shared__ double sh[32];
  if (sc->done) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t gw = blockIdx.x * 4 + wv, nw = gridDim.x * 4;
  const double a = sc->a;
  double acc[NREG];
#pragma unroll
  for (int i = 0; i < NREG; ++i) acc[i] = 0.0;
  double s_clogZ = 0.0, s_rH = 0.0;
  for (uint32_t j = gw; j < E; j += nw) {
    const double c = cvec[j];
    if (c == 0.0) continue;  // wave-uniform: a zero-count EC contributes nothing to pass B
    const double *row = Lt + (size_t)j * G;
    double m = -INFINITY, S0 = 0.0, S1 = 0.0;  // S1 = sum exp(y - m) * (x - y)
#pragma unroll 4
    for (int i = 0; i < NREG; ++i) {
      const int g = lane + 64 * i;
      if (g < G) {
        const double x = row[g];
        const double y = a * x + u[g];
        if (y > m) {
          const double f = exp(m - y);
          S0 = S0 * f + 1.0;
          S1 = S1 * f + (x - y);
          m = y;
        } else {
          const double p = exp(y - m);
          S0 += p;
          S1 += p * (x - y);
        }
      }
    }
    const double mw = wave_max(m);
    const double f = (m == -INFINITY) ? 0.0 : exp(m - mw);
    const double Z = wave_sum(S0 * f);
    const double hs = wave_sum(S1 * f) + mw * Z;  // sum p * (x - (y - mw))
    const double rj = c / Z;
    s_clogZ += c * log(Z);
    s_rH += rj * hs;
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      const int g = lane + 64 * i;
      if (g < G) acc[i] += rj * exp(a * row[g] + u[g] - mw);
    }
  }
  // per-wave partial rows: partAcc has 4 rows per workgroup (k_redfin sums rows in fixed order)
  double *dst = partAcc + ((size_t)blockIdx.x * 4 + wv) * G;
#pragma unroll
  for (int i = 0; i < NREG; ++i) {
    const int g = lane + 64 * i;
    if (g < G) dst[g] = acc[i];
  }
  double t1 = (lane == 0) ? s_clogZ : 0.0, t2 = (lane == 0) ? s_rH : 0.0;
  t1 = block_sum(t1, sh);
  t2 = block_sum(t2, sh);
  if (threadIdx.x == 0) {
    for (int q = 0; q < 4; ++q) {  // one partS slot per partial row keeps npartS == number of rows
      double *o = partS + 4 * ((size_t)blockIdx.x * 4 + q);
      o[0] = q == 0 ? t1 : 0.0;
      o[1] = q == 0 ? t2 : 0.0;
      o[2] = 0.0;
      o[3] = 0.0;
    }
  }
}

}  // namespace msw
