// Developer tool: cost of a grid barrier + a small data exchange on MI355X as a function of the number of
// workgroups (cooperative launch, agent-scope release/acquire on one counter).  The measurement behind the decision
// NOT to run the iteration as one persistent kernel: hipcc --offload-arch=gfx950 -O3 tools/barrier_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
// sense-free counting barrier: generation counter; every WG adds 1, waits until count reaches gen*nwg
__device__ __forceinline__ void grid_barrier(unsigned *cnt, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}
__global__ __launch_bounds__(1024) void k(unsigned *cnt, int iters, double *data, int n, double *out) {
  double acc = 0;
  for (int it = 0; it < iters; ++it) {
    // a little work: each WG writes its slot, after the barrier reads everyone's
    if (threadIdx.x == 0) data[blockIdx.x] = it + blockIdx.x;
    grid_barrier(cnt, (unsigned)(it + 1) * gridDim.x);
    for (int i = threadIdx.x; i < (int)gridDim.x; i += blockDim.x) acc += __hip_atomic_load(&data[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (acc == 1.2345) out[0] = acc;
}
int main() {
  unsigned *cnt; double *data, *out;
  hipMalloc(&cnt, 4); hipMalloc(&data, 8 * 4096); hipMalloc(&out, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nwg : {8, 16, 32, 64, 128, 256}) for (int thr : {256, 1024}) {
    const int iters = 2000;
    float best = 1e9;
    for (int r = 0; r < 4; ++r) {
      hipMemset(cnt, 0, 4);
      hipEventRecord(e0);
      void *args[] = {&cnt, (void *)&iters, &data, (void *)&nwg, &out};
      hipLaunchCooperativeKernel((void *)k, dim3(nwg), dim3(thr), args, 0, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("nwg %4d thr %4d: %.2f us per barrier+exchange\n", nwg, thr, best * 1e3 / iters); fflush(stdout);
  }
  return 0;
}
