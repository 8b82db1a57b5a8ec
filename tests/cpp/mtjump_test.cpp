// Checks the GF(2) jump-ahead of the MT19937-64 stream (msweep_amd/csrc/host_mtjump.inc) against
// libstdc++'s std::mt19937_64::discard, the generator the reference draws from
// (src/BootstrapSample.cpp:60-73).  Prints "ok" lines; exit code 1 on any mismatch.
#include <array>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>

#include "host_mtjump.inc"

static uint64_t temper(uint64_t x) {
  x ^= (x >> 29) & 0x5555555555555555ULL;
  x ^= (x << 17) & 0x71D67FFFEDA60000ULL;
  x ^= (x << 37) & 0xFFF7EEE000000000ULL;
  x ^= (x >> 43);
  return x;
}
static std::array<uint64_t, 312> seeded(uint64_t x) {
  std::array<uint64_t, 312> w;
  w[0] = x;
  for (int i = 1; i < 312; ++i) {
    x = 6364136223846793005ULL * (x ^ (x >> 62)) + (uint64_t)i;
    w[i] = x;
  }
  return w;
}

// deep positions no test can step to in seconds: libstdc++'s own state there (tests/golden/mt_deep_state.json, written by
// discard() in gen_mt_deep_state.cpp), handed over by tests/test_mtjump.py as lines of `seed skip w0 .. w311 pos`
static int deep_cases(const char *path) {
  int bad = 0;
  std::ifstream in(path);
  std::string line;
  while (std::getline(in, line)) {
    if (line.empty()) continue;
    std::istringstream is(line);
    long long seed;
    unsigned long long skip;
    is >> seed >> skip;
    std::mt19937_64 ref;
    is >> ref;  // 312 words + position, as operator<< printed them
    if (!is) {
      std::printf("deep: unreadable state line\n");
      return 1;
    }
    auto w = seeded((uint64_t)(int64_t)(int32_t)seed);
    mtjump::jump(w, skip);
    mtjump::Window win;
    win.w = w;
    bool ok = true;
    for (int i = 0; i < 2000; ++i) ok &= temper(win.step()) == ref();
    std::printf("deep jump seed %lld skip %llu %s\n", seed, skip, ok ? "ok" : "MISMATCH");
    bad += !ok;
  }
  return bad;
}

int main(int argc, char **argv) {
  int bad = 0;
  if (argc > 1) return deep_cases(argv[1]) ? 1 : 0;
  const uint64_t seed = (uint64_t)(int64_t)(int32_t)-7;  // how std::mt19937_64(int32) widens
  for (uint64_t J : {0ull, 1ull, 155ull, 156ull, 311ull, 312ull, 313ull, 1000003ull, (1ull << 27) + 12345ull}) {
    auto w = seeded(seed);
    mtjump::jump(w, J);
    mtjump::Window win;
    win.w = w;
    std::mt19937_64 ref(seed);
    ref.discard(J);
    bool ok = true;
    for (int i = 0; i < 2000; ++i) ok &= temper(win.step()) == ref();
    std::printf("jump %llu %s\n", (unsigned long long)J, ok ? "ok" : "MISMATCH");
    bad += !ok;
  }
  // distances no one can step through: jumps compose
  auto a = seeded(5), b = a;
  mtjump::jump(a, 40000000000ull);
  mtjump::jump(a, 0xfffffff000000000ull - 40000000000ull);
  mtjump::jump(b, 0xfffffff000000000ull);
  mtjump::Window wa, wb;
  wa.w = a;
  wb.w = b;
  bool ok = true;
  for (int i = 0; i < 2000; ++i) ok &= wa.step() == wb.step();
  std::printf("compose %s\n", ok ? "ok" : "MISMATCH");
  bad += !ok;
  return bad ? 1 : 0;
}
