// offline: does choosing WHICH ECs share a lane block (flat residue histograms) cut the modelled LDS cycles?
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include <algorithm>
#include <cstdlib>
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
    if (getenv("SIM_RENUMBER")) {  // groups renumbered by descending frequency: the hottest 16 sit in 16 different banks
      static std::vector<uint32_t> rank;
      if (rank.empty()) {
        std::vector<uint32_t> ord(G); for (uint32_t g = 0; g < G; ++g) ord[g] = g;
        std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return th[a] > th[b]; });
        rank.resize(G); for (uint32_t r = 0; r < G; ++r) rank[ord[r]] = r;
      }
      for (auto &v : ecs) for (auto &g : v) g = rank[g];
    }
    std::vector<int> order(NEC);
    for (int i = 0; i < NEC; ++i) order[i] = i;
    if (mode == 1 || mode == 7) {
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
      if (mode == 6 || mode == 7) {
        // proper edge colouring (alternating paths) of lanes x residues per contiguous 16-lane group
        std::vector<int> step_of[64];
        for (int l = 0; l < 64; ++l) step_of[l].assign(cells[l].size(), -1);
        for (int C = 0; C < 4; ++C) {
          const int NC = 64;
          struct Edge { int u, v, c; };
          std::vector<Edge> ed;
          for (int l = C * 16; l < C * 16 + 16; ++l) for (size_t c = 0; c < cells[l].size(); ++c) ed.push_back({l - C * 16, (int)(cells[l][c] & 15), -1});
          std::vector<std::vector<int>> atU(16, std::vector<int>(NC, -1)), atV(16, std::vector<int>(NC, -1));
          for (int e = 0; e < (int)ed.size(); ++e) {
            int u = ed[e].u, v = ed[e].v;
            int a = 0; while (atU[u][a] >= 0) ++a;
            int b = 0; while (atV[v][b] >= 0) ++b;
            if (a != b) {
              std::vector<int> path; int side = 1, node = v, col = a;
              while (true) {
                int e2 = side ? atV[node][col] : atU[node][col];
                if (e2 < 0) break;
                path.push_back(e2);
                node = side ? ed[e2].u : ed[e2].v; side ^= 1; col = (col == a) ? b : a;
              }
              for (int e2 : path) { atU[ed[e2].u][ed[e2].c] = -1; atV[ed[e2].v][ed[e2].c] = -1; }
              for (int e2 : path) { ed[e2].c = (ed[e2].c == a) ? b : a; }
              for (int e2 : path) { atU[ed[e2].u][ed[e2].c] = e2; atV[ed[e2].v][ed[e2].c] = e2; }
            }
            ed[e].c = a; atU[u][a] = e; atV[v][a] = e;
          }
          int idx = 0;
          for (int l = C * 16; l < C * 16 + 16; ++l) {
            std::vector<char> used(len, 0);
            size_t base = idx;
            for (size_t c = 0; c < cells[l].size(); ++c) { int col = ed[base + c].c; if (col < len) used[col] = 1; }
            for (size_t c = 0; c < cells[l].size(); ++c) {
              int col = ed[base + c].c;
              if (col >= len) { col = 0; while (used[col]) ++col; used[col] = 1; }
              step_of[l][c] = col;
            }
            idx += (int)cells[l].size();
          }
        }
        for (int k = 0; k < len; ++k) for (int l = 0; l < 64; ++l) sched[k][l] = G + l;
        for (int l = 0; l < 64; ++l) for (size_t c = 0; c < cells[l].size(); ++c) sched[step_of[l][c]][l] = cells[l][c];
        eval(sched, cost);
        continue;
      }
      if (mode == 5) {
        // residue-major: per step and 16-lane contiguous group, residues by decreasing remaining degree
        // each take the lane with the most unplaced cells that still has a cell of that residue
        for (int k = 0; k < len; ++k) {
          int pickc[64]; for (int l = 0; l < 64; ++l) pickc[l] = -1;
          for (int C = 0; C < 4; ++C) {
            int deg[16] = {0};
            for (int l = C * 16; l < C * 16 + 16; ++l) for (size_t c = 0; c < cells[l].size(); ++c) if (!taken[l][c]) ++deg[cells[l][c] & 15];
            int ord[16]; for (int r = 0; r < 16; ++r) ord[r] = r;
            std::stable_sort(ord, ord + 16, [&](int a, int b) { return deg[a] > deg[b]; });
            for (int oi = 0; oi < 16; ++oi) {
              const int r = ord[oi]; if (!deg[r]) break;
              int bl = -1, bc = -1, brem = -1;
              for (int l = C * 16; l < C * 16 + 16; ++l) if (pickc[l] < 0) {
                int rem = 0; int cc = -1;
                for (size_t c = 0; c < cells[l].size(); ++c) if (!taken[l][c]) { ++rem; if ((int)(cells[l][c] & 15) == r && cc < 0) cc = (int)c; }
                if (cc >= 0 && rem > brem) { brem = rem; bl = l; bc = cc; }
              }
              if (bl >= 0) pickc[bl] = bc;
            }
            // lanes without a pick that cannot wait (remaining cells == remaining steps) take anything
            for (int l = C * 16; l < C * 16 + 16; ++l) if (pickc[l] < 0) {
              int rem = 0, first = -1;
              for (size_t c = 0; c < cells[l].size(); ++c) if (!taken[l][c]) { ++rem; if (first < 0) first = (int)c; }
              if (rem >= len - k && first >= 0) pickc[l] = first;
            }
          }
          for (int l = 0; l < 64; ++l) {
            if (pickc[l] >= 0) { taken[l][pickc[l]] = 1; sched[k][l] = cells[l][pickc[l]]; }
            else sched[k][l] = G + l;
          }
        }
        eval(sched, cost);
        continue;
      }
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
