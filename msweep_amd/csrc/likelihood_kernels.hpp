// likelihood_kernels.hpp -- device build of the CSR-of-ECs likelihood from the pseudoalignment:
//   K0  k_lut_build        beta-binomial lookup table (include/Likelihood.hpp:47-60,92-107,198-207)
//   K1  k_ec_groups_*      per-EC (group, #sequences hit) lists (include/Likelihood.hpp:122-139;
//                          "currently the slowest part in the input reading", :121)
//   K2  k_group_hits,      --min-hits mask and compaction (include/Likelihood.hpp:141-171)
//       k_compact_*
// The reference probes an E x T bit matrix cell by cell (O(E*T)); here each EC's hit list is
// reduced by one wavefront with register shuffles (O(hits)), or by a workgroup with an LDS
// histogram over the groups when an EC hits more than 64 targets.
#pragma once
#include "device_util.hpp"

namespace msw {

// ---- exclusive scan of uint32 (n up to 2^32): 2048 elements per workgroup --------------------
constexpr int kScanChunk = 2048;

__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t *sh /*>=33*/, uint32_t *total) {
  // 256 threads; returns the exclusive prefix of v within the block
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  __syncthreads();
  if (lane == 63) sh[wv] = x;
  __syncthreads();
  uint32_t off = 0, tot = 0;
  for (int i = 0; i < 4; ++i) {
    if (i < wv) off += sh[i];
    tot += sh[i];
  }
  if (total) *total = tot;
  return off + x - v;
}

__global__ __launch_bounds__(256) void k_scan_local(const uint32_t *in, uint64_t n, uint32_t *out,
                                                   uint32_t *block_tot) {
  __shared__ uint32_t sh[40];
  const uint64_t base = (uint64_t)blockIdx.x * kScanChunk;
  uint32_t v[8], s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint64_t j = base + (uint64_t)threadIdx.x * 8 + i;
    v[i] = j < n ? in[j] : 0u;
    s += v[i];
  }
  uint32_t tot;
  uint32_t off = block_excl_scan_u32(s, sh, &tot);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint64_t j = base + (uint64_t)threadIdx.x * 8 + i;
    if (j < n) out[j] = off;
    off += v[i];
  }
  if (threadIdx.x == 0) block_tot[blockIdx.x] = tot;
}

// serial-over-chunks scan of the block totals by ONE workgroup (<= a few thousand entries);
// also writes the grand total to out_total (as uint64 to detect overflow on the host)
__global__ __launch_bounds__(256) void k_scan_totals(uint32_t *block_tot, uint32_t nblocks, uint64_t *out_total) {
  __shared__ uint32_t sh[40];
  uint64_t run = 0;
  for (uint32_t base = 0; base < nblocks; base += 256) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < nblocks ? block_tot[i] : 0u;
    uint32_t tot;
    const uint32_t off = block_excl_scan_u32(v, sh, &tot);
    if (i < nblocks) block_tot[i] = (uint32_t)(run + off);
    run += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) *out_total = run;
}

__global__ __launch_bounds__(256) void k_scan_add(uint32_t *out, uint64_t n, const uint32_t *block_off) {
  const uint64_t base = (uint64_t)blockIdx.x * kScanChunk;
  const uint32_t o = block_off[blockIdx.x];
  for (int i = threadIdx.x; i < kScanChunk; i += 256) {
    const uint64_t j = base + i;
    if (j < n) out[j] += o;
  }
}

// ---- K0: lookup table -------------------------------------------------------------------------
// lut[c*ld + k] for size class c (group size n_c): log(zi) for k = 0 or k > n_c, else
// ldbb_scaled(k, n_c, alpha_c, beta_c) + log1p(-zi).
__global__ __launch_bounds__(256) void k_lut_build(const uint32_t *class_size, uint32_t n_class, uint32_t ld,
                                                  double q, double eps, double zi, double *lut) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (uint64_t)n_class * ld) return;
  const uint32_t c = (uint32_t)(i / ld), k = (uint32_t)(i % ld);
  const double n = (double)class_size[c];
  double v = log(zi);
  if (k >= 1 && (double)k <= n) {
    // update_bb_parameters (:198-207), bb_constants = {q, e}
    const double e = n * q;
    const double phi = 1.0 / (n - e + eps);
    const double beta = phi * (n - e);
    const double alpha = (e * beta) / (n - e);
    const double kk = (double)k;
    const double lbc = lgamma(n + 1.0) - lgamma(kk + 1.0) - lgamma(n - kk + 1.0);
    const double lb1 = lgamma(kk + alpha) + lgamma(n - kk + beta) - lgamma(kk + alpha + (n - kk + beta));
    const double lb2 = lgamma(n + alpha) + lgamma(beta) - lgamma(n + alpha + beta);
    v = lbc + lb1 - lb2 + log1p(-zi);
  }
  lut[i] = v;
}

// ---- K1: per-EC group counts ------------------------------------------------------------------
// Three paths by the number of hits of an EC:
//   <= 16 : one THREAD per EC -- the groups of its hits sorted by a 16-key network in registers, runs of equal
//           groups counted (most ECs of an alignment: 12 hits on average at cfg5; until round 3 these took a
//           wavefront each with 12 of 64 lanes busy: 30 ms of the build);
//   17..64: one wavefront per EC, lane l holds the group of hit l;
//   > 64  : one workgroup per EC, histogram over the groups.
// The first pass of the thread kernel sorts the ECs into the three paths (id lists appended per wavefront) and checks
// what the host used to walk the arrays for: ec_tptr monotone, target ids in range.
//   EMIT = false: nd[i] = number of distinct groups;
//   EMIT = true : (group, count) pairs written at rowptr[i] in ascending group order.
constexpr uint32_t kEcThreadMax = 16, kEcWaveMax = 64;
enum : uint32_t { kCtrBad = 0, kCtrMid = 1, kCtrLong = 2, kBuildCtrs = 4 };
enum : uint32_t { kBadTptr = 1, kBadTarget = 2 };

// bitonic sorting network, 16 keys in registers, ascending
__device__ __forceinline__ void sort16(uint32_t (&v)[16]) {
#pragma unroll
  for (int k = 2; k <= 16; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int l = i ^ j;
        if (l > i) {
          const uint32_t lo = min(v[i], v[l]), hi = max(v[i], v[l]);
          const bool up = (i & k) == 0;
          v[i] = up ? lo : hi;
          v[l] = up ? hi : lo;
        }
      }
    }
  }
}

// append `id` to a list for every lane with `want` set, at *at (wave-uniform: the wavefront's own range of the list,
// reserved with ONE atomic per wavefront and kernel -- an atomic per round, 146 k of them on one word at cfg3, queued in
// L2 for as long as the rest of the kernel took)
__device__ __forceinline__ void wave_append_at(bool want, uint32_t id, uint32_t *list, uint32_t &at) {
  const unsigned long long m = __ballot(want);
  if (m == 0) return;
  const int lane = threadIdx.x & 63;
  if (want) list[at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = id;
  at += (uint32_t)__popcll(m);
}
// the wavefront's range of a list: `mine` = entries this LANE will append over the whole kernel
__device__ __forceinline__ uint32_t wave_reserve(uint32_t mine, uint32_t *counter) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
  uint32_t base = 0;
  if ((threadIdx.x & 63) == 0 && mine) base = atomicAdd(counter, mine);
  return (uint32_t)__shfl((int)base, 0, 64);
}

template <bool EMIT>
__global__ __launch_bounds__(256) void k_ec_groups_thread(const uint64_t *tptr, const uint32_t *targets,
                                                         const uint32_t *tgroup, uint32_t E, uint32_t T, uint64_t ntot,
                                                         uint32_t *nd, const uint32_t *rowptr, uint32_t *out_grp,
                                                         uint32_t *out_cnt, uint32_t *mid_ids, uint32_t *long_ids,
                                                         uint32_t *ctr) {
  const uint32_t stride = gridDim.x * blockDim.x;
  [[maybe_unused]] uint32_t at_mid = 0, at_long = 0;
  if (!EMIT) {  // the lengths of the wavefront's ECs once more, up front: where its entries of the two lists go
    uint32_t n_mid = 0, n_long = 0;
    for (uint32_t i0 = blockIdx.x * blockDim.x; i0 < E; i0 += stride) {
      const uint32_t i = i0 + threadIdx.x;
      if (i < E) {
        const uint64_t b = tptr[i], e = tptr[i + 1];
        const uint64_t n64 = (e < b || e > ntot) ? 0 : e - b;
        n_mid += n64 > kEcThreadMax && n64 <= kEcWaveMax;
        n_long += n64 > kEcWaveMax;
      }
    }
    at_mid = wave_reserve(n_mid, &ctr[kCtrMid]);
    at_long = wave_reserve(n_long, &ctr[kCtrLong]);
  }
  // (every lane of a wavefront runs the same number of rounds: the list appends are wave operations)
  for (uint32_t i0 = blockIdx.x * blockDim.x; i0 < E; i0 += stride) {
    const uint32_t i = i0 + threadIdx.x;
    const bool in = i < E;
    uint64_t b = 0, e = 0;
    if (in) b = tptr[i], e = tptr[i + 1];
    const bool broken = e < b || e > ntot;  // (no EC of a broken ec_tptr is read: every access stays inside ec_targets)
    const uint64_t n64 = broken ? 0 : e - b;
    if (!EMIT) {
      if (broken) atomicOr(&ctr[kCtrBad], kBadTptr);
      wave_append_at(in && n64 > kEcThreadMax && n64 <= kEcWaveMax, i, mid_ids, at_mid);
      wave_append_at(in && n64 > kEcWaveMax, i, long_ids, at_long);
      if (in && n64 == 0) nd[i] = 0;
    }
    if (!in || n64 == 0 || n64 > kEcThreadMax) continue;
    const uint32_t n = (uint32_t)n64;
    uint32_t v[16];
#pragma unroll
    for (uint32_t k = 0; k < 16; ++k) {
      v[k] = 0xffffffffu;
      if (k < n) {
        uint32_t t = targets[b + k];
        if (t >= T) {
          if (!EMIT) atomicOr(&ctr[kCtrBad], kBadTarget);
          t = 0;
        }
        v[k] = tgroup[t];
      }
    }
    sort16(v);
    if (!EMIT) {
      uint32_t d = 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) d += (uint32_t)k < n && (k == 0 || v[k] != v[k - 1]);
      nd[i] = d;
    } else {
      uint32_t o = rowptr[i], run = 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if ((uint32_t)k < n) {
          ++run;
          if ((uint32_t)k + 1 == n || v[(k + 1) & 15] != v[k]) {  // the last hit of its group
            out_grp[o] = v[k];
            out_cnt[o] = run;
            ++o;
            run = 0;
          }
        }
      }
    }
  }
}

// ECs of 17..64 hits (the list the thread kernel made): one wavefront per EC, lane l holds the group of hit l.
template <bool EMIT>
__global__ __launch_bounds__(256) void k_ec_groups_wave(const uint32_t *ids, uint32_t n_ids, const uint64_t *tptr,
                                                       const uint32_t *targets, const uint32_t *tgroup, uint32_t T,
                                                       uint32_t *nd, const uint32_t *rowptr, uint32_t *out_grp,
                                                       uint32_t *out_cnt, uint32_t *ctr) {
  const int lane = threadIdx.x & 63;
  const uint32_t gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  for (uint32_t r = gw; r < n_ids; r += nw) {
    const uint32_t i = ids[r];
    const uint64_t b = tptr[i];
    const uint32_t n = (uint32_t)(tptr[i + 1] - b);  // 17..64
    uint32_t g = 0xffffffffu;
    if ((uint32_t)lane < n) {
      uint32_t t = targets[b + lane];
      if (t >= T) {
        if (!EMIT) atomicOr(&ctr[kCtrBad], kBadTarget);
        t = 0;
      }
      g = tgroup[t];
    }
    uint32_t cnt = 0;
    bool first = (uint32_t)lane < n;
    for (uint32_t k = 0; k < n; ++k) {
      const uint32_t gk = __shfl(g, (int)k, 64);
      cnt += (gk == g);
      if (k < (uint32_t)lane && gk == g) first = false;
    }
    const unsigned long long mask = __ballot(first);
    if (!EMIT) {
      if (lane == 0) nd[i] = (uint32_t)__popcll(mask);
    } else {
      uint32_t rank = 0;
      for (uint32_t k = 0; k < n; ++k) {
        if ((mask >> k) & 1ull) {
          const uint32_t gk = __shfl(g, (int)k, 64);
          rank += (gk < g);
        }
      }
      if (first) {
        const uint32_t o = rowptr[i] + rank;
        out_grp[o] = g;
        out_cnt[o] = cnt;
      }
    }
  }
}

// Long ECs (> 64 hits): one workgroup per EC with a histogram over the groups (LDS when it fits,
// else a per-workgroup slab of `scratch`).
template <bool EMIT>
__global__ __launch_bounds__(256) void k_ec_groups_long(const uint32_t *long_ids, uint32_t n_long,
                                                       const uint64_t *tptr, const uint32_t *targets,
                                                       const uint32_t *tgroup, uint32_t G, uint32_t T,
                                                       uint32_t *scratch, uint32_t *nd, const uint32_t *rowptr,
                                                       uint32_t *out_grp, uint32_t *out_cnt, uint32_t *ctr) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ uint32_t sh[40];
  uint32_t *H = scratch ? scratch + (size_t)blockIdx.x * G : reinterpret_cast<uint32_t *>(smem);
  for (uint32_t r = blockIdx.x; r < n_long; r += gridDim.x) {
    const uint32_t i = long_ids[r];
    const uint64_t b = tptr[i], e = tptr[i + 1];
    for (uint32_t g = threadIdx.x; g < G; g += 256) H[g] = 0;
    __syncthreads();
    for (uint64_t k = b + threadIdx.x; k < e; k += 256) {
      uint32_t t = targets[k];
      if (t >= T) {
        if (!EMIT) atomicOr(&ctr[kCtrBad], kBadTarget);
        t = 0;
      }
      atomicAdd(&H[tgroup[t]], 1u);
    }
    __syncthreads();
    uint32_t run = 0;
    for (uint32_t base = 0; base < G; base += 256) {
      const uint32_t g = base + threadIdx.x;
      const uint32_t c = g < G ? H[g] : 0u;
      uint32_t tot;
      const uint32_t off = block_excl_scan_u32(c != 0u ? 1u : 0u, sh, &tot);
      if (EMIT && c != 0u) {
        const uint32_t o = rowptr[i] + run + off;
        out_grp[o] = g;
        out_cnt[o] = c;
      }
      run += tot;
      __syncthreads();
    }
    if (!EMIT && threadIdx.x == 0) nd[i] = run;
    __syncthreads();
  }
}

// ---- K2: --min-hits ---------------------------------------------------------------------------
// hits_g = sum over ECs hitting g of reads_in_ec (include/Likelihood.hpp:149-154); integer atomics -- into a
// histogram in LDS over the groups [g0, g0 + ng) (one workgroup per CU; 64-bit LDS atomics), flushed once per
// workgroup.  (Until round 3: one global atomic per cell on G addresses, 18 ms at cfg5.)
constexpr uint32_t kHitsGroups = 16384;  // groups per pass: 128 KB of LDS
__global__ __launch_bounds__(1024) void k_group_hits(const uint32_t *rowptr, const uint32_t *grp, uint32_t E,
                                                    const uint64_t *ec_counts, uint32_t g0, uint32_t ng,
                                                    unsigned long long *hits) {
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned long long *H = reinterpret_cast<unsigned long long *>(smem);
  for (uint32_t g = threadIdx.x; g < ng; g += blockDim.x) H[g] = 0ull;
  __syncthreads();
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < E; i += gridDim.x * blockDim.x) {
    const unsigned long long c = ec_counts[i];
    for (uint32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
      const uint32_t g = grp[k] - g0;
      if (g < ng) atomicAdd(&H[g], c);
    }
  }
  __syncthreads();
  for (uint32_t g = threadIdx.x; g < ng; g += blockDim.x)
    if (H[g]) atomicAdd(&hits[g0 + g], H[g]);
}

// cells of kept groups per EC (count), then compaction with group ids remapped to their position
// among the kept groups (original order, include/Likelihood.hpp:156-163)
__global__ __launch_bounds__(256) void k_compact_count(const uint32_t *rowptr, const uint32_t *grp, uint32_t E,
                                                      const int32_t *pos, uint32_t *nd) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < E; i += gridDim.x * blockDim.x) {
    uint32_t n = 0;
    for (uint32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) n += pos[grp[k]] >= 0;
    nd[i] = n;
  }
}
__global__ __launch_bounds__(256) void k_compact_emit(const uint32_t *rowptr, const uint32_t *grp,
                                                     const uint32_t *cnt, uint32_t E, const int32_t *pos,
                                                     const uint32_t *rowptr2, uint32_t *grp2, uint32_t *cnt2) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < E; i += gridDim.x * blockDim.x) {
    uint32_t o = rowptr2[i];
    for (uint32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
      const int32_t p = pos[grp[k]];
      if (p >= 0) {
        grp2[o] = (uint32_t)p;
        cnt2[o] = cnt[k];
        ++o;
      }
    }
  }
}

// LUT slot of every cell: idx = lut_off[group] + count; flags counts above the group size
__global__ __launch_bounds__(256) void k_cell_lutidx(const uint32_t *grp, const uint32_t *cnt, uint64_t nnz,
                                                    const uint32_t *lut_off, const uint32_t *gsize,
                                                    uint32_t *idx, int *bad) {
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t g = grp[k], c = cnt[k];
    if (c > gsize[g]) *bad = 1;
    idx[k] = lut_off[g] + (c > gsize[g] ? 0u : c);
  }
}

// msw_core_set_csr: the caller's 64-bit row pointers as 32-bit ones, checked on the way (bad |= 4: a row ends before
// it starts, or beyond the nnz the last pointer promises)
__global__ __launch_bounds__(256) void k_rowptr_narrow(const uint64_t *rp, uint64_t n, uint64_t nnz, uint32_t *out, int *bad) {
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t v = rp[j];
    if (v > nnz || (j + 1 < n && rp[j + 1] < v)) atomicOr(bad, 4);
    out[j] = (uint32_t)(v > nnz ? nnz : v);
  }
}

// msw_core_set_csr: LUT slot of every cell from the caller's (group, count) pairs, idx = lut_off[group] +
// count, with the range checks of the upload (bad: 1 = group id out of range, 2 = count beyond the table)
__global__ __launch_bounds__(256) void k_csr_lutidx(const uint32_t *grp, const uint32_t *cnt, uint64_t nnz,
                                                   uint32_t n_groups, uint32_t lut_ld, const uint32_t *lut_off,
                                                   uint32_t *idx, int *bad) {
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t g = grp[k], c = cnt[k];
    if (g >= n_groups) {
      atomicOr(bad, 1);
      idx[k] = 0;
    } else if (c >= lut_ld) {
      atomicOr(bad, 2);
      idx[k] = 0;
    } else {
      idx[k] = lut_off[g] + c;
    }
  }
}

// fill_ec_counts (include/Likelihood.hpp:188-195)
__global__ __launch_bounds__(256) void k_log_counts(const uint64_t *ec_counts, uint32_t E, double *logc) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < E; i += gridDim.x * blockDim.x)
    logc[i] = log((double)ec_counts[i]);
}

}  // namespace msw
