// bootstrap_kernels.hpp -- device bootstrap resampling, bit-exact with the reference's
// std::mt19937_64 + std::discrete_distribution<uint32_t> (src/BootstrapSample.cpp:33-73).
//
// The reference draws ALL replicates from ONE sequential Mersenne-Twister stream, so replicate
// b starts at draw b * bootstrap_count.  k_mt64 advances that stream on the device (one
// workgroup, 156 words per step) and emits tempered words on a second stream underneath the
// previous replicate's solve; k_resample maps every word to an EC exactly as libstdc++ does
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
    st->idx = 0;
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
// Stream form of the recurrence: with x_0..x_311 the seeded state and x_k the k-th state word ever
// produced, x_{k+312} = x_{k+156} ^ twist(x_k, x_{k+1}); every new word depends only on words at
// least 156 positions back, so 156 words are produced per step with ONE barrier (the textbook
// in-place refill needs three dependent phases per 312 words).  Output word j = temper(x_{312+j}).
// st->mt holds the last 312 state words; the ring lives in LDS while the kernel runs.
constexpr int kMtRing = 1024, kMtStep = 156;
__global__ __launch_bounds__(256) void k_mt64(MtState *st, uint64_t skip, uint64_t n, uint64_t *out) {
  __shared__ uint64_t ring[kMtRing];
  const int t = threadIdx.x;
  for (int i = t; i < kMtN; i += 256) ring[i] = st->mt[i];
  uint32_t p = kMtN;  // ring position (mod kMtRing) of the next word
  __syncthreads();
  uint64_t todo_skip = skip, todo = n, written = 0;
  while (todo_skip + todo > 0) {
    const bool skipping = todo_skip > 0;
    const uint64_t rem = skipping ? todo_skip : todo;
    const uint32_t cnt = rem < (uint64_t)kMtStep ? (uint32_t)rem : (uint32_t)kMtStep;
    if ((uint32_t)t < cnt) {
      const uint64_t a = ring[(p - 312 + t) & (kMtRing - 1)];
      const uint64_t b = ring[(p - 311 + t) & (kMtRing - 1)];
      const uint64_t c = ring[(p - 156 + t) & (kMtRing - 1)];
      const uint64_t v = mt_twist(a, b, c);
      ring[(p + t) & (kMtRing - 1)] = v;  // never a slot read in this step (window 312 < ring - step)
      if (!skipping) out[written + t] = mt_temper(v);
    }
    __syncthreads();
    p += cnt;
    if (skipping) {
      todo_skip -= cnt;
    } else {
      todo -= cnt;
      written += cnt;
    }
  }
  for (int i = t; i < kMtN; i += 256) st->mt[i] = ring[(p - 312 + i) & (kMtRing - 1)];
  if (t == 0) st->produced += skip + n;
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
