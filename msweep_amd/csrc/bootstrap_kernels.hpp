// bootstrap_kernels.hpp -- device bootstrap resampling, bit-exact with the reference's
// std::mt19937_64 + std::discrete_distribution<uint32_t> (src/BootstrapSample.cpp:33-73).
//
// The reference draws ALL replicates from ONE sequential Mersenne-Twister stream, so replicate
// b starts at draw b * bootstrap_count.  k_mt64 advances that stream on the device (one
// wavefront; the 312-word recurrence has 156-wide parallelism per phase) and emits tempered
// words; k_resample maps every word to an EC exactly as libstdc++ does
// (generate_canonical<double,53> = double(x) * 2^-64 clamped below 1, then lower_bound on the
// normalised partial sums) and counts with integer atomics (order independent).
#pragma once
#include "common.hpp"

namespace msw {

constexpr int kMtN = 312, kMtM = 156;

struct MtState {
  uint64_t mt[kMtN];
  uint64_t produced;  // words of the stream consumed so far
  uint32_t idx;       // next word inside mt[] (kMtN = refill needed)
  uint32_t pad;
};

__global__ __launch_bounds__(64) void k_mt64_seed(MtState *st, uint64_t seed) {
  if (threadIdx.x == 0) {
    uint64_t x = seed;
    st->mt[0] = x;
    for (int i = 1; i < kMtN; ++i) {
      x = 6364136223846793005ULL * (x ^ (x >> 62)) + (uint64_t)i;
      st->mt[i] = x;
    }
    st->idx = kMtN;
    st->produced = 0;
  }
}

__device__ __forceinline__ uint64_t mt_temper(uint64_t x) {
  x ^= (x >> 29) & 0x5555555555555555ULL;
  x ^= (x << 17) & 0x71D67FFFEDA60000ULL;
  x ^= (x << 37) & 0xFFF7EEE000000000ULL;
  x ^= (x >> 43);
  return x;
}
__device__ __forceinline__ uint64_t mt_twist(uint64_t hi, uint64_t lo, uint64_t far) {
  const uint64_t x = (hi & 0xFFFFFFFF80000000ULL) | (lo & 0x7FFFFFFFULL);
  return far ^ (x >> 1) ^ ((x & 1ULL) ? 0xB5026F5AA96619E9ULL : 0ULL);
}

// Skips `skip` words, then writes `n` tempered words to out (out may be null when n == 0).
// One wavefront; the state lives in LDS while the kernel runs.
__global__ __launch_bounds__(64) void k_mt64(MtState *st, uint64_t skip, uint64_t n, uint64_t *out) {
  __shared__ uint64_t mt[kMtN];
  const int lane = threadIdx.x;
  for (int i = lane; i < kMtN; i += 64) mt[i] = st->mt[i];
  uint32_t idx = st->idx;
  const uint64_t produced0 = st->produced;
  __syncthreads();
  uint64_t todo_skip = skip, todo = n, written = 0;
  while (todo_skip + todo > 0) {
    if (idx >= (uint32_t)kMtN) {
      // refill: phase 1 (i < 156: old inputs), phase 2 (156 <= i < 311: new mt[i-156]), phase 3 (i = 311)
      uint64_t nv[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int i = lane + 64 * c;
        if (i < kMtM) nv[c] = mt_twist(mt[i], mt[i + 1], mt[i + kMtM]);
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int i = lane + 64 * c;
        if (i < kMtM) mt[i] = nv[c];
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int i = kMtM + lane + 64 * c;
        if (i < kMtN - 1) nv[c] = mt_twist(mt[i], mt[i + 1], mt[i - kMtM]);
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int i = kMtM + lane + 64 * c;
        if (i < kMtN - 1) mt[i] = nv[c];
      }
      __syncthreads();
      if (lane == 0) mt[kMtN - 1] = mt_twist(mt[kMtN - 1], mt[0], mt[kMtM - 1]);
      __syncthreads();
      idx = 0;
    }
    const uint64_t avail = (uint64_t)(kMtN - idx);
    if (todo_skip > 0) {
      const uint64_t s = todo_skip < avail ? todo_skip : avail;
      idx += (uint32_t)s;
      todo_skip -= s;
      continue;
    }
    const uint64_t take = todo < avail ? todo : avail;
    for (uint64_t i = lane; i < take; i += 64) out[written + i] = mt_temper(mt[idx + i]);
    idx += (uint32_t)take;
    written += take;
    todo -= take;
  }
  __syncthreads();
  for (int i = lane; i < kMtN; i += 64) st->mt[i] = mt[i];
  if (lane == 0) {
    st->idx = idx;
    st->produced = produced0 + skip + n;
  }
}

// counts[lower_bound(cp, p)] += 1 for every word (std::discrete_distribution::operator()).
__global__ __launch_bounds__(256) void k_resample(const uint64_t *words, uint64_t n, const double *cp,
                                                 uint32_t n_ecs, uint32_t *counts) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (uint64_t)gridDim.x * blockDim.x) {
    double p = __ull2double_rn(words[i]) * 0x1p-64;
    if (p >= 1.0) p = 0x1.fffffffffffffp-1;  // nextafter(1.0, 0.0)
    uint32_t lo = 0, hi = n_ecs;             // first position with cp[pos] >= p
    while (lo < hi) {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      if (cp[mid] < p) lo = mid + 1;
      else hi = mid;
    }
    atomicAdd(&counts[lo], 1u);
  }
}

}  // namespace msw
