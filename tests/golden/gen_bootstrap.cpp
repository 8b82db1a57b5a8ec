// Generates bootstrap_golden.json with the very libstdc++ types the reference instantiates in
// src/BootstrapSample.cpp:33-73 (std::mt19937_64 seeded with an int32, std::discrete_distribution
// <uint32_t> over the EC counts, bootstrap_count sequential draws per replicate).
// Build + run:  g++ -O2 -std=c++17 tests/golden/gen_bootstrap.cpp -o /tmp/gen_bootstrap && /tmp/gen_bootstrap > tests/golden/bootstrap_golden.json
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

static void emit_case(bool first, int32_t seed, const std::vector<uint32_t> &w, size_t draws, size_t reps) {
  std::mt19937_64 gen(seed);
  std::discrete_distribution<uint32_t> dist(w.begin(), w.end());
  printf("%s {\"seed\": %d, \"draws\": %zu, \"weights\": [", first ? "" : ",\n", seed, draws);
  for (size_t i = 0; i < w.size(); ++i) printf("%s%u", i ? "," : "", w[i]);
  printf("], \"counts\": [");
  for (size_t r = 0; r < reps; ++r) {
    std::vector<uint32_t> c(w.size());
    for (size_t i = 0; i < draws; ++i) c[dist(gen)] += 1;
    printf("%s[", r ? "," : "");
    for (size_t i = 0; i < c.size(); ++i) printf("%s%u", i ? "," : "", c[i]);
    printf("]");
  }
  printf("]}");
}

int main() {
  printf("{\"generator\": \"tests/golden/gen_bootstrap.cpp (libstdc++ std::mt19937_64 + std::discrete_distribution<uint32_t>)\",\n \"mt_first_words_seed42\": [");
  std::mt19937_64 g(42);
  for (int i = 0; i < 8; ++i) printf("%s%llu", i ? "," : "", (unsigned long long)g());
  printf("],\n \"mt_word_10000_seed5489\": %llu,\n \"cases\": [\n", [] { std::mt19937_64 m; m.discard(9999); return (unsigned long long)m(); }());
  std::vector<uint32_t> w1 = {5, 1, 3, 7, 2, 9, 4, 4, 1, 30};
  emit_case(true, 42, w1, 66, 3);
  emit_case(false, -7, w1, 1000, 2);
  std::vector<uint32_t> w2(257);
  for (size_t i = 0; i < w2.size(); ++i) w2[i] = (uint32_t)((i * 2654435761u) % 97 + 1);
  emit_case(false, 20230126, w2, 5000, 2);
  std::vector<uint32_t> w3 = {0, 3, 0, 0, 5, 0};   // zero-weight ECs are never drawn
  emit_case(false, 1, w3, 50, 2);
  std::vector<uint32_t> w4 = {11};                 // a single EC: libstdc++ draws nothing
  emit_case(false, 3, w4, 20, 2);
  printf("\n ]}\n");
  return 0;
}
