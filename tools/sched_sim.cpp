// offline: does choosing WHICH ECs share a lane block (flat residue histograms) cut the modelled LDS cycles?
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include <algorithm>
static const uint8_t kRGroup[64] = {0,0,0,0,1,1,1,1,1,1,1,1,0,0,0,0,1,1,1,1,0,0,0,0,0,0,0,0,1,1,1,1,
                                    2,2,2,2,3,3,3,3,3,3,3,3,2,2,2,2,3,3,3,3,2,2,2,2,2,2,2,2,3,3,3,3};
// 8-lane blocks = intersections of the contiguous 16-lane groups and the b128 groups
static int block_of(int l) { return (l >> 4) * 2 + (kRGroup[l] & 1); }  // 0..7
struct Cost { double at = 0, ew = 0, e = 0; long steps = 0; };
static void eval(const std::vector<std::vector<uint32_t>> &sched, Cost &c) {
  for (auto &row : sched) {
    auto cyc = [&](int ngroups, auto group_of, uint32_t nb, bool combine) {
      int tot = 0;
      for (int gr = 0; gr < ngroups; ++gr) {
        int mx = 0;
        for (uint32_t b = 0; b < nb; ++b) {
          uint32_t seen[64]; int ns = 0;
          for (int l = 0; l < 64; ++l) if (group_of(l) == gr && row[l] % nb == b) {
            bool dup = false; if (combine) for (int q = 0; q < ns; ++q) dup |= seen[q] == row[l];
            if (!dup) seen[ns++] = row[l];
          }
          mx = std::max(mx, ns);
        }
        tot += mx;
      }
      return tot;
    };
    c.at += cyc(4, [](int l) { return l >> 4; }, 16, false);
    c.ew += cyc(4, [](int l) { return (int)kRGroup[l]; }, 16, true);
    c.e += cyc(2, [](int l) { return l >> 5; }, 32, true);
    c.steps++;
  }
}
int main(int argc, char **argv) {
  int mode = argc > 1 ? atoi(argv[1]) : 0;  // 0: ECs in arrival order; 1: flat blocks
  int K = argc > 2 ? atoi(argv[2]) : 32;
  const bool tie = argc > 3 && atoi(argv[3]);
  const uint32_t G = 5000; std::mt19937_64 rng(1);
  std::vector<double> th(G); std::gamma_distribution<double> gd(0.05, 1.0);
  for (auto &t : th) t = gd(rng);
  std::discrete_distribution<uint32_t> src(th.begin(), th.end());
  Cost cost;
  for (int len0 = 1; len0 <= 16; ++len0) {
    const int len = len0 + (len0 & 1);
    const int NEC = 64 * 200;
    std::vector<std::vector<uint32_t>> ecs(NEC);
    for (auto &v : ecs) { v.push_back(src(rng)); while ((int)v.size() < len0) { uint32_t g = rng() % G; if (std::find(v.begin(), v.end(), g) == v.end()) v.push_back(g); } }
    std::vector<int> order(NEC);
    for (int i = 0; i < NEC; ++i) order[i] = i;
    if (mode >= 1) {
      // greedy: K open 8-EC blocks; each EC goes where it adds the least overflow above ceil(len0/2)
      struct Blk { int n = 0; int cnt[16] = {0}; std::vector<int> ids; };
      std::vector<Blk> open(K); std::vector<int> out;
      const int cap = (len0 + 1) / 2;
      for (int i = 0; i < NEC; ++i) {
        int best = -1, bc = 1 << 30;
        for (int b = 0; b < K; ++b) {
          int c = 0;
          int tmp[16]; memcpy(tmp, open[b].cnt, sizeof tmp);
          for (uint32_t g : ecs[i]) { if (++tmp[g & 15] > cap) c += 4; }
          c = c * 16 + (8 - open[b].n);  // prefer fuller blocks on ties
          if (c < bc) { bc = c; best = b; }
        }
        Blk &B = open[best];
        for (uint32_t g : ecs[i]) ++B.cnt[g & 15];
        B.ids.push_back(i);
        if (++B.n == 8) { for (int id : B.ids) out.push_back(id); B = Blk(); }
      }
      for (auto &B : open) for (int id : B.ids) out.push_back(id);
      order = out;
    }
    for (int s = 0; s < NEC / 64; ++s) {
      // lanes: block b (0..7) takes 8 consecutive ECs of `order`
      std::vector<std::vector<uint32_t>> cells(64);
      int fill[8] = {0};
      int next = s * 64;
      for (int l = 0; l < 64; ++l) {}
      // assign: for block b its 8 lanes in increasing lane order
      std::vector<int> lanes_of[8];
      for (int l = 0; l < 64; ++l) lanes_of[block_of(l)].push_back(l);
      for (int b = 0; b < 8; ++b) for (int i = 0; i < 8; ++i) cells[lanes_of[b][i]] = ecs[order[next++]];
      (void)fill;
      std::vector<std::vector<uint32_t>> sched(len, std::vector<uint32_t>(64));
      std::vector<std::vector<char>> taken(64);
      for (int l = 0; l < 64; ++l) taken[l].assign(cells[l].size(), 0);
      for (int k = 0; k < len; ++k) {
        uint32_t rg[4][16], hg[2][32]; uint16_t at[4] = {0,0,0,0};
        memset(rg, 0xff, sizeof rg); memset(hg, 0xff, sizeof hg);
        int remC[4][16] = {{0}}, remR[4][16] = {{0}};
        for (int l = 0; l < 64; ++l) for (size_t c = 0; c < cells[l].size(); ++c) if (!taken[l][c]) { ++remC[l >> 4][cells[l][c] & 15]; ++remR[kRGroup[l]][cells[l][c] & 15]; }
        for (int li = 0; li < 64; ++li) {
          const int l = (li + k * 7) & 63; const int R = kRGroup[l], C = l >> 4, H = l >> 5;
          int best = -1, bs = -1;
          for (size_t c = 0; c < cells[l].size(); ++c) if (!taken[l][c]) {
            const uint32_t g = cells[l][c]; int sc = 0;
            if (!(at[C] >> (g & 15) & 1)) sc += 8;
            if (rg[R][g & 15] == UINT32_MAX || rg[R][g & 15] == g) sc += 6;
            if (hg[H][g & 31] == UINT32_MAX || hg[H][g & 31] == g) sc += 2;
            sc *= 1024;
            if (tie) sc += remC[C][g & 15] + remR[R][g & 15];
            if (sc > bs) { bs = sc; best = (int)c; }
          }
          if (best < 0) { sched[k][l] = G + l; continue; }
          const uint32_t g = cells[l][best]; taken[l][best] = 1;
          at[C] |= 1u << (g & 15);
          if (rg[R][g & 15] == UINT32_MAX) rg[R][g & 15] = g;
          if (hg[H][g & 31] == UINT32_MAX) hg[H][g & 31] = g;
          sched[k][l] = g;
        }
      }
      eval(sched, cost);
    }
  }
  printf("mode %d K %d: atomics %.2f  ew %.2f  e %.2f (ideal 4 / 4 / 2)\n", mode, K, cost.at / cost.steps, cost.ew / cost.steps, cost.e / cost.steps);
}
