// Checks the GF(2) jump-ahead of the MT19937-64 stream (msweep_amd/csrc/host_mtjump.inc) against
// libstdc++'s std::mt19937_64::discard, the generator the reference draws from
// (src/BootstrapSample.cpp:60-73).  Prints "ok" lines; exit code 1 on any mismatch.
#include <array>
#include <cstdint>
#include <cstdio>
#include <random>
#include <stdexcept>

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

int main() {
  int bad = 0;
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
