// device_util.hpp -- wave / block reductions and the reference digamma series (gfx950, wave64).
#pragma once
#include "common.hpp"

namespace msw {

// ---------------------------------------------------------------------------------------
// wave / block reductions (wave64; xor butterflies: every lane ends with the same value,
// the order of additions is fixed -> bitwise reproducible)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, kWave));
  return v;
}
// sh: >= 16 doubles of LDS scratch.  Result valid in every thread.
__device__ __forceinline__ double block_sum(double v, double *sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double r = 0.0;
  for (int i = 0; i < nw; ++i) r += sh[i];
  return r;
}
// N sums at once with one pair of barriers.  sh: >= 16 * N doubles.  Per value the order of
// additions is that of block_sum, so both give bitwise the same result.
template <int N>
__device__ __forceinline__ void block_sum_n(double (&v)[N], double *sh) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = wave_sum(v[i]);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) sh[w * N + i] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double r = 0.0;
    for (int j = 0; j < nw; ++j) r += sh[j * N + i];
    v[i] = r;
  }
}
__device__ __forceinline__ double block_max(double v, double *sh) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double r = sh[0];
  for (int i = 1; i < nw; ++i) r = fmax(r, sh[i]);
  return r;
}

// digamma: the 7-shift asymptotic series of the reference (src/Sample.cpp:87-97; rcgpar
// carries the same function for the RCG gradient).
__device__ __forceinline__ double digamma_ref(double x) {
  double result = 0.0;
  for (; x < 7.0; x += 1.0) result -= 1.0 / x;
  x -= 0.5;
  const double xx = 1.0 / x, xx2 = xx * xx, xx4 = xx2 * xx2;
  result += log(x) + (1. / 24.) * xx2 - (7.0 / 960.0) * xx4 + (31.0 / 8064.0) * xx4 * xx2 -
            (127.0 / 30720.0) * xx4 * xx4;
  return result;
}

}  // namespace msw
