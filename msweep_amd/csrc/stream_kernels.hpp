// stream_kernels.hpp -- the practical HBM ceiling of the box the benchmark runs on ("use a
// measured stream rate as the practical ceiling and state both"): a read-only sweep shaped like the sweeps'
// record stream (16-byte loads, one pass, nothing written) and the classic triad (two reads, one write).
// Measurement only: no product path calls these.
#pragma once
#include "common.hpp"

namespace msw {

// first launch of the process: loads the library's code object (~15 ms for its few hundred sweep instantiations) at
// msw_core_create instead of inside the first msw_core_set_csr / msw_core_build_likelihood
__global__ void k_warm(int *p) {
  if (p) *p = 0;
}


__global__ __launch_bounds__(1024) void k_stream_read(const double2 *a, size_t n, double *sink) {
  double s0 = 0.0, s1 = 0.0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 7 * stride < n; i += 8 * stride) {  // eight independent 16-byte loads in flight per lane
    double2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = a[i + u * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s0 += v[u].x;
      s1 += v[u].y;
    }
  }
  for (; i < n; i += stride) {
    const double2 v = a[i];
    s0 += v.x;
    s1 += v.y;
  }
  if (s0 + s1 == 12345.678) *sink = s0;  // never true for the zero-filled buffer: keeps the loads alive
}

__global__ __launch_bounds__(1024) void k_stream_triad(double2 *a, const double2 *b, const double2 *c, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + stride < n; i += 2 * stride) {
    const double2 x0 = b[i], y0 = c[i], x1 = b[i + stride], y1 = c[i + stride];
    a[i] = make_double2(fma(3.0, y0.x, x0.x), fma(3.0, y0.y, x0.y));
    a[i + stride] = make_double2(fma(3.0, y1.x, x1.x), fma(3.0, y1.y, x1.y));
  }
  for (; i < n; i += stride) {
    const double2 x = b[i], y = c[i];
    a[i] = make_double2(fma(3.0, y.x, x.x), fma(3.0, y.y, x.y));
  }
}

}  // namespace msw
