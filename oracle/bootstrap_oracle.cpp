// bootstrap_oracle.cpp -- CPU ORACLE (test infrastructure, see msweep_oracle.h).
// Restates src/BootstrapSample.cpp:33-73 of the reference.
//   orc_bootstrap_counts_stdlib   uses the very libstdc++ types the reference instantiates
//                                 (std::mt19937_64, std::discrete_distribution<uint32_t>);
//   orc_bootstrap_counts_restated is a from-scratch restatement of the same stream
//                                 (MT19937-64 recurrence, generate_canonical<double,53>,
//                                 lower_bound on the normalised partial sums) -- the form
//                                 the HIP kernels mirror.  tests/ pin one against the other.
#include "msweep_oracle.h"

#include <algorithm>
#include <cmath>
#include <random>
#include <sstream>
#include <vector>

extern "C" {

void orc_bootstrap_counts_stdlib(const uint32_t *weights, size_t n_ecs, int32_t seed,
                                 size_t bootstrap_count, size_t n_reps, uint32_t *out) {
  // BootstrapSample::construct :46-58 (explicit seed branch) + init_bootstrap :33-44
  std::mt19937_64 gen(seed);
  std::discrete_distribution<uint32_t> ec_distribution(weights, weights + n_ecs);
  for (size_t r = 0; r < n_reps; ++r) {
    // resample_counts :60-73
    std::vector<uint32_t> tmp_counts(n_ecs);
    for (size_t i = 0; i < bootstrap_count; ++i) {
      size_t ec_id = ec_distribution(gen);
      tmp_counts[ec_id] += 1;
    }
    std::copy(tmp_counts.begin(), tmp_counts.end(), out + r * n_ecs);
  }
}

// The same replicate loop entered in the MIDDLE of the stream: the generator takes the state libstdc++ itself
// printed (operator<< of std::mt19937_64: 312 words, then the position of the next word) after stepping there --
// tests/golden/mt_deep_state.json holds it for the start of replicate 999 of BASELINE config 4, which is 10^10 - 10^7
// words from the seed (src/mSWEEP.cpp:496-518 gets there by drawing the 999 replicates before it).
void orc_bootstrap_counts_stdlib_from_state(const uint32_t *weights, size_t n_ecs, const uint64_t *state312,
                                            uint64_t pos, size_t bootstrap_count, size_t n_reps, uint32_t *out) {
  std::ostringstream os;
  for (int i = 0; i < 312; ++i) os << state312[i] << ' ';
  os << pos;
  std::istringstream is(os.str());
  std::mt19937_64 gen;
  is >> gen;
  std::discrete_distribution<uint32_t> ec_distribution(weights, weights + n_ecs);
  for (size_t r = 0; r < n_reps; ++r) {
    std::vector<uint32_t> tmp_counts(n_ecs);
    for (size_t i = 0; i < bootstrap_count; ++i) tmp_counts[ec_distribution(gen)] += 1;
    std::copy(tmp_counts.begin(), tmp_counts.end(), out + r * n_ecs);
  }
}

}  // extern "C"

namespace {
// MT19937-64 (Matsumoto & Nishimura), parameters as in std::mt19937_64.
struct MT64 {
  static constexpr int NN = 312, MM = 156;
  static constexpr uint64_t MATRIX_A = 0xB5026F5AA96619E9ULL, UM = 0xFFFFFFFF80000000ULL,
                            LM = 0x7FFFFFFFULL;
  uint64_t mt[NN];
  int mti;
  explicit MT64(uint64_t seed) {
    mt[0] = seed;
    for (mti = 1; mti < NN; ++mti)
      mt[mti] = 6364136223846793005ULL * (mt[mti - 1] ^ (mt[mti - 1] >> 62)) + (uint64_t)mti;
  }
  void refill() {
    for (int i = 0; i < NN; ++i) {
      uint64_t x = (mt[i] & UM) | (mt[(i + 1) % NN] & LM);
      mt[i] = mt[(i + MM) % NN] ^ (x >> 1) ^ ((x & 1ULL) ? MATRIX_A : 0ULL);
    }
    mti = 0;
  }
  uint64_t next() {
    if (mti >= NN) refill();
    uint64_t x = mt[mti++];
    x ^= (x >> 29) & 0x5555555555555555ULL;
    x ^= (x << 17) & 0x71D67FFFEDA60000ULL;
    x ^= (x << 37) & 0xFFF7EEE000000000ULL;
    x ^= (x >> 43);
    return x;
  }
};
}  // namespace

extern "C" {

void orc_mt19937_64_words(uint64_t seed, size_t skip, size_t n, uint64_t *out) {
  MT64 g(seed);
  for (size_t i = 0; i < skip; ++i) g.next();
  for (size_t i = 0; i < n; ++i) out[i] = g.next();
}

void orc_discrete_cp(const uint32_t *weights, size_t n, double *cp) {
  // libstdc++ discrete_distribution::param_type::_M_initialize
  // (/usr/include/c++/11/bits/random.tcc): probabilities = double(w) / sum (sequential
  // accumulate), partial_sum, last entry forced to 1.0.
  if (n == 0) return;
  double sum = 0.0;
  for (size_t i = 0; i < n; ++i) sum += (double)weights[i];
  double run = 0.0;
  for (size_t i = 0; i < n; ++i) {
    run += (double)weights[i] / sum;
    cp[i] = run;
  }
  cp[n - 1] = 1.0;
}

void orc_bootstrap_counts_restated(const uint32_t *weights, size_t n_ecs, int32_t seed,
                                   size_t bootstrap_count, size_t n_reps, uint32_t *out) {
  // std::mt19937_64(int32 seed): the seed is converted to result_type (uint64), i.e. a
  // negative int32 sign-extends.
  MT64 gen((uint64_t)(int64_t)seed);
  std::vector<double> cp(n_ecs);
  orc_discrete_cp(weights, n_ecs, cp.data());
  for (size_t r = 0; r < n_reps; ++r) {
    uint32_t *c = out + r * n_ecs;
    std::fill(c, c + n_ecs, 0u);
    for (size_t i = 0; i < bootstrap_count; ++i) {
      size_t pos = 0;
      if (n_ecs > 1) {  // libstdc++ returns 0 without drawing when there is one weight
        // generate_canonical<double,53>: one 64-bit draw, double(x) / 2^64, clamp below 1
        double p = (double)gen.next() * 0x1p-64;
        if (p >= 1.0) p = std::nextafter(1.0, 0.0);
        pos = std::lower_bound(cp.begin(), cp.end(), p) - cp.begin();
      }
      c[pos] += 1;
    }
  }
}

}  // extern "C"
