// Generates mt_deep_state.json: the state of libstdc++'s std::mt19937_64 -- the generator the reference draws
// ALL bootstrap replicates from (src/BootstrapSample.cpp:46-73) -- at the start of replicate 999 of BASELINE
// config 4 (`--iters 1000 --seed 42` on 10^7 aligned reads: 999 * 10^7 = 9 990 000 000 words into the stream),
// reached the only way the reference can reach it: by stepping (discard).  ~15 s of one core.
// Build + run:  g++ -O2 -std=c++17 tests/golden/gen_mt_deep_state.cpp -o /tmp/gen_mt_deep && /tmp/gen_mt_deep > tests/golden/mt_deep_state.json
#include <cstdint>
#include <cstdio>
#include <random>
#include <sstream>
#include <string>
#include <vector>

static void emit(bool first, int32_t seed, unsigned long long skip) {
  std::mt19937_64 gen(seed);
  gen.discard(skip);
  // operator<< writes the 312 state words and then the position of the next word inside them
  std::ostringstream os;
  os << gen;
  std::istringstream is(os.str());
  std::vector<unsigned long long> v;
  unsigned long long x;
  while (is >> x) v.push_back(x);
  printf("%s {\"seed\": %d, \"skip\": %llu, \"state\": [", first ? "" : ",\n", seed, skip);
  for (size_t i = 0; i < 312; ++i) printf("%s%llu", i ? "," : "", v[i]);
  printf("], \"pos\": %llu, \"next_words\": [", v[312]);
  for (int i = 0; i < 16; ++i) printf("%s%llu", i ? "," : "", (unsigned long long)gen());
  printf("]}");
}

int main() {
  printf("{\"generator\": \"tests/golden/gen_mt_deep_state.cpp (libstdc++ std::mt19937_64::discard)\",\n \"cases\": [\n");
  emit(true, 42, 9990000000ull);     // replicate 999 of cfg4 (10^7 draws per replicate)
  emit(false, 42, 1500000000ull);    // replicate 150
  emit(false, -7, 4294967296ull + 12345ull);  // beyond 2^32 words, negative seed (sign-extends)
  printf("\n ]}\n");
  return 0;
}
