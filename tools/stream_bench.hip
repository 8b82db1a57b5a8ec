// micro-benchmark: record stream of the sweeps as 4-byte records (dword per lane, 256-B rows) vs
// 3-byte records in aligned 12-byte groups of four (dwordx3 per lane, 768-B quad rows).
// Persistent 256 x 1024 threads, one slice per wave at a time with one slice prefetched,
// alternating direction between passes (Infinity Cache reuse) -- the structure of k_passA / k_passB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct U3 { uint32_t a, b, c; };
// rows_per_slice rows of 256 B per slice
__global__ __launch_bounds__(1024) void k4(const uint32_t *rec, uint32_t nslices, int rows, int reverse, uint32_t *out) {
  const uint32_t w = blockIdx.x * 16 + (threadIdx.x >> 6), nw = gridDim.x * 16, lane = threadIdx.x & 63;
  const uint32_t n_mine = w < nslices ? (nslices - w + nw - 1) / nw : 0;
  uint32_t acc = 0;
  uint32_t cur[16], nxt[16];
  auto sl = [&](uint32_t i) { return w + (reverse ? n_mine - 1 - i : i) * nw; };
  auto fetch = [&](uint32_t i, uint32_t (&b)[16]) {
    const uint32_t *p = rec + (size_t)sl(i) * rows * 64 + lane;
#pragma unroll
    for (int k = 0; k < 16; ++k) if (k < rows) b[k] = p[k * 64];
  };
  if (n_mine) fetch(0, cur);
  for (uint32_t i = 0; i < n_mine; ++i) {
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (i + 1 < n_mine) fetch(i + 1, nxt);
#pragma unroll
    for (int k = 0; k < 16; ++k) if (k < rows) acc += cur[k] * (k + 1);
#pragma unroll
    for (int k = 0; k < 16; ++k) cur[k] = nxt[k];
  }
  if (acc == 0x12345678u) out[0] = acc;
}
// quads rows of 768 B per slice
__global__ __launch_bounds__(1024) void k3(const U3 *rec, uint32_t nslices, int quads, int reverse, uint32_t *out) {
  const uint32_t w = blockIdx.x * 16 + (threadIdx.x >> 6), nw = gridDim.x * 16, lane = threadIdx.x & 63;
  const uint32_t n_mine = w < nslices ? (nslices - w + nw - 1) / nw : 0;
  uint32_t acc = 0;
  U3 cur[4], nxt[4];
  auto sl = [&](uint32_t i) { return w + (reverse ? n_mine - 1 - i : i) * nw; };
  auto fetch = [&](uint32_t i, U3 (&b)[4]) {
    const U3 *p = rec + (size_t)sl(i) * quads * 64 + lane;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (k < quads) b[k] = p[k * 64];
  };
  if (n_mine) fetch(0, cur);
  for (uint32_t i = 0; i < n_mine; ++i) {
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (i + 1 < n_mine) fetch(i + 1, nxt);
#pragma unroll
    for (int k = 0; k < 4; ++k) if (k < quads) acc += cur[k].a + cur[k].b * 3 + cur[k].c * 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) cur[k] = nxt[k];
  }
  if (acc == 0x12345678u) out[0] = acc;
}
int main() {
  const uint32_t nslices = 146000;
  const int rows = 10;   // 4-byte records: 10 rows x 256 B  = 2560 B per slice -> 374 MB
  const int quads = 3;   // 3-byte records: 3 quad rows x 768 B = 2304 B per slice (12 cells) -> 336 MB
  uint32_t *d4; U3 *d3; uint32_t *out;
  const size_t b4 = (size_t)nslices * rows * 256, b3 = (size_t)nslices * quads * 768;
  CK(hipMalloc(&d4, b4)); CK(hipMalloc(&d3, b3)); CK(hipMalloc(&out, 64));
  CK(hipMemset(d4, 1, b4)); CK(hipMemset(d3, 1, b3));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int variant = 0; variant < 4; ++variant) {
    const int alt = variant & 1, fmt = variant >> 1;
    for (int it = 0; it < 4; ++it) { if (fmt) hipLaunchKernelGGL(k3, dim3(256), dim3(1024), 0, 0, d3, nslices, quads, alt ? (it & 1) : 0, out); else hipLaunchKernelGGL(k4, dim3(256), dim3(1024), 0, 0, d4, nslices, rows, alt ? (it & 1) : 0, out); }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int N = 40;
    for (int it = 0; it < N; ++it) { if (fmt) hipLaunchKernelGGL(k3, dim3(256), dim3(1024), 0, 0, d3, nslices, quads, alt ? (it & 1) : 0, out); else hipLaunchKernelGGL(k4, dim3(256), dim3(1024), 0, 0, d4, nslices, rows, alt ? (it & 1) : 0, out); }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = fmt ? b3 : b4;
    printf("%s records, %s direction: %.1f us per pass, %.0f MB, %.2f TB/s\n", fmt ? "3-byte quads (dwordx3)" : "4-byte (dword)    ", alt ? "alternating" : "same       ", ms / N * 1e3, bytes / 1e6, bytes / (ms / N * 1e-3) / 1e12);
  }
  return 0;
}
