// reader_kernels.hpp -- device side of the Themisto plaintext reader (round 5): the text of a
// pseudoalignment file is parsed, collapsed into equivalence classes and handed to the likelihood build without
// leaving HBM.  Same outcome as the host reader (host_alignment.inc), array for array; reference:
// include/mSWEEP_alignment.hpp:54-94 (line parser), :97-135 (paired-end merge), :137-215 (collapse).
//
// Byte work: every kernel here is a streaming pass (text: 1 byte per character, twice; tokens: 4 bytes each) or a
// gather over rows of ~16 integers; nothing is shaped for the matrix cores.
//
// Text -> tokens.  A TOKEN is a maximal run of digits; a line is `read_id target target ...`.  The text is cut into
// tiles of kTileBytes; pass 1 counts tokens and newlines per tile and checks the bytes (anything the host parser
// would not take silently raises a flag, and the caller hands the file to the host parser, whose word is final);
// the tile counts are scanned; pass 2 converts every token and writes
//   tokens[k]          value of the k-th token of the file
//   line_first[l]      index of the first token (the read id) of line l; line_first[n_lines] = n_tokens.
#pragma once
#include "common.hpp"

#include <cstdlib>

#include <map>
#include <mutex>
#include <unordered_map>

namespace msw {

constexpr int kSegBytes = 16;                             // bytes of text per thread
constexpr int kTileThreads = 256;
constexpr int kTileBytes = kSegBytes * kTileThreads;      // 4 KB per workgroup
// what the text checks report (ReaderCtr::flags)
constexpr uint32_t kTxtBadChar = 1u;      // a byte that is no digit, blank, line feed or carriage return
constexpr uint32_t kTxtBadCr = 2u;        // a carriage return that does not end its line
constexpr uint32_t kTxtBadLineStart = 4u; // a line that does not start with a digit (empty lines too)
constexpr uint32_t kTxtOverflow = 8u;     // a token of more than 10 digits or beyond 2^32 - 1
constexpr uint32_t kTxtBadTarget = 16u;   // a target id >= n_targets

struct ReaderCtr {
  uint32_t flags;
  uint32_t max_id;      // largest read id
  uint32_t unsorted;    // flag: some rows' targets do not ascend strictly
  uint32_t shrunk;      // flag: some rows lost duplicates
  uint32_t long_rows;   // flag: rows out of order with more than 64 targets (sorted as keys, host_reader.inc)
};

// the 16 bytes of a thread's segment as bit masks: digits, line feeds; `bad` collects the byte checks
struct SegMasks {
  uint32_t dig, nl, cr, blank;
};
__device__ __forceinline__ SegMasks seg_masks(const unsigned char (&c)[kSegBytes], uint32_t valid) {
  SegMasks m = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < kSegBytes; ++i) {
    const uint32_t b = c[i];
    m.dig |= (uint32_t)(b - '0' <= 9u) << i;
    m.nl |= (uint32_t)(b == '\n') << i;
    m.cr |= (uint32_t)(b == '\r') << i;
    m.blank |= (uint32_t)(b == ' ') << i;
  }
  m.dig &= valid, m.nl &= valid, m.cr &= valid, m.blank &= valid;
  return m;
}
__device__ __forceinline__ void load_seg(const unsigned char *txt, uint64_t at, unsigned char (&c)[kSegBytes]) {
  const uint4 v = *reinterpret_cast<const uint4 *>(txt + at);  // (the buffer is padded to a multiple of the tile)
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < kSegBytes; ++i) c[i] = (unsigned char)(w[i >> 2] >> (8 * (i & 3)));
}
__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ uint32_t wave_umax(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor(v, o));
  return v;
}
// inclusive scan of v over the 256 threads of a workgroup (sh: 4 words); returns the inclusive value, *total = the sum
__device__ __forceinline__ uint32_t tile_scan_incl(uint32_t v, uint32_t *sh, uint32_t *total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t u = __shfl_up(v, o);
    if (lane >= o) v += u;
  }
  if (lane == 63) sh[w] = v;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < kTileThreads / 64; ++k) {
    const uint32_t s = sh[k];
    if (k < w) base += s;
    tot += s;
  }
  *total = tot;
  return v + base;
}

// Pass 1 over the text: tokens and line feeds per tile, byte checks.
// cnt_tok[tile] = tokens that START in the tile, cnt_nl[tile] = line feeds.
__global__ __launch_bounds__(kTileThreads) void k_text_count(const unsigned char *txt, uint64_t n, uint32_t *cnt_tok,
                                                              uint32_t *cnt_nl, ReaderCtr *ctr) {
  __shared__ uint32_t sh[8];
  const uint64_t at = ((uint64_t)blockIdx.x * kTileThreads + threadIdx.x) * kSegBytes;
  uint32_t ntok = 0, nnl = 0, bad = 0;
  if (at < n) {
    unsigned char c[kSegBytes];
    load_seg(txt, at, c);
    const uint32_t valid = n - at >= (uint64_t)kSegBytes ? 0xffffu : (1u << (uint32_t)(n - at)) - 1u;
    const SegMasks m = seg_masks(c, valid);
    const unsigned char prev = at ? txt[at - 1] : (unsigned char)'\n';
    const bool have_next = at + kSegBytes < n;
    const unsigned char next = have_next ? txt[at + kSegBytes] : (unsigned char)'\n';
    const uint32_t pdig = (m.dig << 1 | (uint32_t)(prev - '0' <= 9u)) & 0xffffu;   // digit in front of byte i
    const uint32_t pnl = (m.nl << 1 | (uint32_t)(prev == '\n')) & 0xffffu;         // line feed in front of byte i
    const uint32_t starts = m.dig & ~pdig;
    ntok = __popc(starts);
    nnl = __popc(m.nl);
    if ((m.dig | m.nl | m.cr | m.blank) != valid) bad |= kTxtBadChar;
    // a carriage return ends its line: the next byte is a line feed (or the text ends there)
    const uint32_t nl_after = (m.nl >> 1) | ((have_next ? (uint32_t)(next == '\n') : 1u) << (kSegBytes - 1));
    // (inside a last, partial segment the byte after the last valid one is the end of the text)
    const uint32_t end_after = valid != 0xffffu ? (valid + 1u) >> 1 : 0u;
    if (m.cr & ~(nl_after | end_after)) bad |= kTxtBadCr;
    // every line starts with a digit: the byte behind a line feed (and the first byte of the text)
    if (pnl & valid & ~m.dig) bad |= kTxtBadLineStart;
  }
  // block sums
  uint32_t tot = 0;
  const uint32_t packed = ntok | nnl << 16;  // (<= 8 tokens and 16 line feeds per thread, 4096 per tile)
  (void)tile_scan_incl(packed, sh, &tot);
  bad = wave_or(bad);
  if (threadIdx.x == 0) {
    cnt_tok[blockIdx.x] = tot & 0xffffu;
    cnt_nl[blockIdx.x] = tot >> 16;
  }
  if (bad && (threadIdx.x & 63) == 0) atomicOr(&ctr->flags, bad);
}

// Pass 2: the tokens' values and the lines' first tokens.  base_tok[tile], base_nl[tile] = tokens / line feeds in
// front of the tile (exclusive scans of pass 1's counts).
__global__ __launch_bounds__(kTileThreads) void k_text_parse(const unsigned char *txt, uint64_t n, const uint64_t *base_tok,
                                                              const uint64_t *base_nl, uint32_t n_targets, uint32_t *tokens,
                                                              uint64_t *line_first, uint32_t *tile_max, ReaderCtr *ctr) {
  __shared__ uint32_t sh[8];
  const uint64_t at = ((uint64_t)blockIdx.x * kTileThreads + threadIdx.x) * kSegBytes;
  uint32_t starts = 0, nls = 0, pnl = 0;
  uint64_t W[4] = {0, 0, 0, 0};  // the segment and the 16 bytes behind it (a token that starts at byte 15 ends by byte 25)
  if (at < n) {
    unsigned char a[kSegBytes];
    load_seg(txt, at, a);
    // (the buffer is padded with line feeds beyond the text: the bytes behind the last segment end a token like
    // the end of the text does)
    const uint4 v0 = *reinterpret_cast<const uint4 *>(txt + at), v1 = *reinterpret_cast<const uint4 *>(txt + at + kSegBytes);
    W[0] = (uint64_t)v0.y << 32 | v0.x, W[1] = (uint64_t)v0.w << 32 | v0.z;
    W[2] = (uint64_t)v1.y << 32 | v1.x, W[3] = (uint64_t)v1.w << 32 | v1.z;
    const uint32_t valid = n - at >= (uint64_t)kSegBytes ? 0xffffu : (1u << (uint32_t)(n - at)) - 1u;
    const SegMasks m = seg_masks(a, valid);
    const unsigned char prev = at ? txt[at - 1] : (unsigned char)'\n';
    const uint32_t pdig = (m.dig << 1 | (uint32_t)(prev - '0' <= 9u)) & 0xffffu;
    pnl = (m.nl << 1 | (uint32_t)(prev == '\n')) & 0xffffu;
    starts = m.dig & ~pdig;
    nls = m.nl;
  }
  uint32_t tot = 0;
  const uint32_t mine = (uint32_t)__popc(starts) | (uint32_t)__popc(nls) << 16;
  const uint32_t incl = tile_scan_incl(mine, sh, &tot);
  const uint32_t excl = incl - mine;
  uint64_t tk = base_tok[blockIdx.x] + (excl & 0xffffu);
  const uint64_t l0 = base_nl[blockIdx.x] + (excl >> 16);
  uint32_t bad = 0, max_id = 0;
  // tokens: at most 10 digits, the byte behind them in hand (a token that starts at byte 15 ends by byte 25)
  uint32_t s = starts;
  const uint64_t tk0 = tk;
  while (s) {
    const int i = __ffs(s) - 1;
    s &= s - 1;
    // the 16 bytes from byte i on, by shifts (no indexed register array)
    const bool up = i >= 8;
    const uint32_t shb = (uint32_t)(i & 7) * 8u;
    const uint64_t A = up ? W[1] : W[0], B = up ? W[2] : W[1], C = up ? W[3] : W[2];
    const uint64_t lo = shb ? A >> shb | B << (64u - shb) : A, hi = shb ? B >> shb | C << (64u - shb) : B;
    uint64_t v = 0;
    int k = 0;
#pragma unroll
    for (int d = 0; d < 11; ++d) {
      const uint32_t ch = (uint32_t)((d < 8 ? lo >> (8 * d) : hi >> (8 * (d - 8))) & 0xffu);
      if (k == d && ch - '0' <= 9u) {
        v = v * 10 + (ch - '0');
        ++k;
      }
    }
    if (k > 10 || v > 0xffffffffull) bad |= kTxtOverflow;
    if ((pnl >> i) & 1u) max_id = max(max_id, (uint32_t)v);   // the line's first token: its read id
    else if (v >= n_targets) bad |= kTxtBadTarget;
    tokens[tk++] = (uint32_t)v;
  }
  // line l + 1 starts behind the l-th line feed: its first token is the next one to start
  uint32_t q = nls, ln = 0;
  while (q) {
    const int i = __ffs(q) - 1;
    q &= q - 1;
    line_first[l0 + ln + 1] = tk0 + (uint32_t)__popc(starts & ((1u << i) - 1u));
    ++ln;
  }
  // the largest read id: per tile, then k_max_u32 (an atomic per wavefront on ONE word was 10 of this kernel's 11.6 ms
  // at 1 GB of text: a million same-address atomics take their turns in L2)
  bad = wave_or(bad);
  max_id = wave_umax(max_id);
  if ((threadIdx.x & 63) == 0) {
    if (bad) atomicOr(&ctr->flags, bad);
    sh[4 + (threadIdx.x >> 6)] = max_id;
  }
  __syncthreads();
  if (threadIdx.x == 0) tile_max[blockIdx.x] = max(max(sh[4], sh[5]), max(sh[6], sh[7]));
}
// *out = max(*out, in[0 .. n)): one workgroup
__global__ __launch_bounds__(1024) void k_max_u32(const uint32_t *in, uint64_t n, uint32_t *out) {
  __shared__ uint32_t sh[16];
  uint32_t m = 0;
  for (uint64_t i = threadIdx.x; i < n; i += 1024) m = max(m, in[i]);
  m = wave_umax(m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 16; ++k) m = max(m, sh[k]);
    *out = max(*out, m);
  }
}

// Rows of at most 64 targets that do not ascend: sorted by the wavefront that finds them (bitonic network over the
// lanes, 21 exchanges), duplicates dropped, written back in place; cnt[r] = the new length, flen[r] = 0 (done).  Rows
// of more than 64 targets keep flen[r] = their length for the key sort of the host side.
__global__ __launch_bounds__(256) void k_rows_sort_short(const uint64_t *ptr, uint32_t *raw, uint64_t n_ids, uint32_t *flen,
                                                          uint32_t *cnt, ReaderCtr *ctr) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  uint32_t lost = 0, left = 0;  // (lost: lane 0's count; left: this lane's rows)
  for (uint64_t r0 = wave * 64; r0 < n_ids; r0 += n_waves * 64) {
    const uint64_t mine = r0 + lane;
    const uint32_t fl = mine < n_ids ? flen[mine] : 0u;
    unsigned long long todo = __ballot(fl != 0u && fl <= 64u);
    left += fl > 64u;
    while (todo) {
      const int k = __ffsll(todo) - 1;
      todo &= todo - 1;
      const uint64_t r = r0 + k;
      const uint32_t len = __shfl(fl, k);
      const uint64_t b = ptr[r];
      uint32_t v = lane < len ? raw[b + lane] : 0xffffffffu;  // (target ids are < 2^32 - 1: the padding sorts last)
#pragma unroll
      for (uint32_t size = 2; size <= 64; size <<= 1) {
#pragma unroll
        for (uint32_t j = size >> 1; j > 0; j >>= 1) {
          const uint32_t o = __shfl_xor(v, j);
          const bool up = (lane & size) == 0, low = (lane & j) == 0;
          v = (low == up) ? min(v, o) : max(v, o);
        }
      }
      const uint32_t left_v = __shfl_up(v, 1);
      const bool keep = lane < len && (lane == 0 || v != left_v);
      const unsigned long long km = __ballot(keep);
      if (keep) raw[b + __popcll(km & ((1ull << lane) - 1ull))] = v;
      const uint32_t kept = (uint32_t)__popcll(km);
      if (lane == 0) {
        cnt[r] = kept;
        flen[r] = 0u;
        lost += kept != len;
      }
    }
  }
  if (lane == 0 && lost) ctr->shrunk = 1u;   // (flags: plain stores)
  if (__ballot(left != 0) && lane == 0) ctr->long_rows = 1u;
}

// The lanes of a wavefront copy 64 rows together: lane l names row l (len elements from src[so ..] to dst[d0 ..]); element
// e of the 64 rows laid end to end is copied by lane e mod 64, so consecutive lanes touch consecutive addresses of a row
// (a thread per row walks 64 rows 64 bytes apart with every load instruction).
__device__ __forceinline__ void wave_copy_rows(const uint32_t *src, uint64_t so, uint32_t *dst, uint64_t d0, uint32_t len) {
  const uint32_t lane = threadIdx.x & 63;
  uint32_t end = len;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t u = __shfl_up(end, o);
    if (lane >= (uint32_t)o) end += u;
  }
  const uint32_t total = __shfl(end, 63);
  for (uint32_t e0 = 0; e0 < total; e0 += 64) {
    const uint32_t e = e0 + lane;
    uint32_t r = 0;  // the row that holds element e: the first lane whose rows end beyond e
#pragma unroll
    for (uint32_t step = 32; step > 0; step >>= 1) {
      const uint32_t pe = __shfl(end, r + step - 1);
      if (pe <= e) r += step;
    }
    r = min(r, 63u);
    const uint32_t r_end = __shfl(end, r), r_len = __shfl(len, r);
    const uint64_t r_so = (uint64_t)__shfl((uint32_t)(so >> 32), r) << 32 | (uint32_t)__shfl((uint32_t)so, r);
    const uint64_t r_d0 = (uint64_t)__shfl((uint32_t)(d0 >> 32), r) << 32 | (uint32_t)__shfl((uint32_t)d0, r);
    if (e < total) {
      const uint32_t i = e - (r_end - r_len);
      dst[r_d0 + i] = src[r_so + i];
    }
  }
}

// ---- rows by read id ------------------------------------------------------------------------------------------------
// A read is the SET of targets of all lines that carry its id (include/mSWEEP_alignment.hpp:62-90: every line sets bits
// of its read's row); ids at or beyond n_ids (= min(largest id + 1, lines)) never take part (:148).
__global__ void k_row_count(const uint32_t *tokens, const uint64_t *line_first, uint64_t n_lines, uint64_t n_ids,
                            uint32_t *cnt) {
  for (uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; l < n_lines; l += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t f = line_first[l];
    const uint32_t len = (uint32_t)(line_first[l + 1] - f - 1);
    const uint32_t rid = tokens[f];
    if (rid < n_ids && len) atomicAdd(&cnt[rid], len);
  }
}
__global__ __launch_bounds__(256) void k_row_fill(const uint32_t *tokens, const uint64_t *line_first, uint64_t n_lines,
                                                   uint64_t n_ids, const uint64_t *ptr, uint32_t *cur, uint32_t *raw) {
  const uint64_t t0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t l0 = t0 & ~63ull; l0 < n_lines; l0 += step) {  // (a wavefront's 64 lines together: wave_copy_rows)
    const uint64_t l = l0 + (threadIdx.x & 63);
    uint64_t so = 0, d0 = 0;
    uint32_t len = 0;
    if (l < n_lines) {
      const uint64_t f = line_first[l];
      const uint32_t n = (uint32_t)(line_first[l + 1] - f - 1);
      const uint32_t rid = tokens[f];
      if (rid < n_ids && n) {
        len = n;
        so = f + 1;
        d0 = ptr[rid] + atomicAdd(&cur[rid], n);
      }
    }
    wave_copy_rows(tokens, so, raw, d0, len);
  }
}
// rows whose targets do not ascend strictly (Themisto promises no order): flen[r] = the row's length, 0 for rows in order
__global__ void k_row_check(const uint64_t *ptr, const uint32_t *raw, uint64_t n_ids, uint32_t *flen, ReaderCtr *ctr) {
  uint32_t mine = 0;
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_ids; r += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t b = ptr[r], e = ptr[r + 1];
    bool asc = true;
    for (uint64_t k = b; k + 1 < e; ++k) asc = asc && raw[k] < raw[k + 1];
    flen[r] = asc ? 0u : (uint32_t)(e - b);
    mine += !asc;
  }
  // (a flag, not a count: a plain store -- one atomic per wavefront on one word is 0.3 ms of queueing in L2 at 10 M rows)
  if (__ballot(mine != 0) && (threadIdx.x & 63) == 0) ctr->unsorted = 1u;
}
// the rows out of order as 64-bit keys (row, target): sorted as a whole, every row comes back ascending
__global__ void k_row_keys(const uint64_t *ptr, const uint32_t *raw, uint64_t n_ids, const uint32_t *flen,
                           const uint64_t *foff, uint64_t *keys) {
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_ids; r += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t len = flen[r];
    if (!len) continue;
    const uint32_t *src = raw + ptr[r];
    uint64_t *dst = keys + foff[r];
    for (uint32_t j = 0; j < len; ++j) dst[j] = r << 32 | src[j];
  }
}
// keep[k] = the sorted key differs from its predecessor (duplicates of a target inside a row go)
__global__ void k_keys_keep(const uint64_t *keys, uint64_t m, uint32_t *keep) {
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < m; k += (uint64_t)gridDim.x * blockDim.x)
    keep[k] = k == 0 || keys[k] != keys[k - 1];
}
// the rows written back in order, without duplicates; cnt[r] = the row's new length
__global__ void k_rows_sorted_back(const uint64_t *keys, const uint32_t *keep, const uint64_t *kidx, const uint64_t *ptr,
                                   const uint32_t *flen, const uint64_t *foff, uint64_t n_ids, uint64_t m, uint32_t *raw,
                                   uint32_t *cnt, ReaderCtr *ctr) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t k = i; k < m; k += step) {
    if (!keep[k]) continue;
    const uint64_t r = keys[k] >> 32;
    raw[ptr[r] + (kidx[k] - kidx[foff[r]])] = (uint32_t)keys[k];
  }
  uint32_t lost = 0;
  for (uint64_t r = i; r < n_ids; r += step) {
    const uint32_t len = flen[r];
    if (!len) continue;
    const uint32_t kept = (uint32_t)(kidx[foff[r] + len] - kidx[foff[r]]);
    cnt[r] = kept;
    lost += kept != len;
  }
  if (__ballot(lost != 0) && (threadIdx.x & 63) == 0) ctr->shrunk = 1u;
}
__global__ __launch_bounds__(256) void k_rows_compact(const uint64_t *ptr, const uint32_t *raw, const uint32_t *cnt,
                                                      const uint64_t *nptr, uint64_t n_ids, uint32_t *out) {
  const uint64_t t0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t r0 = t0 & ~63ull; r0 < n_ids; r0 += step) {
    const uint64_t r = r0 + (threadIdx.x & 63);
    const bool in = r < n_ids;
    wave_copy_rows(raw, in ? ptr[r] : 0, out, in ? nptr[r] : 0, in ? cnt[r] : 0u);
  }
}

// ---- paired-end merge (include/mSWEEP_alignment.hpp:123-133): intersection / union of the two strands' rows ----------
// WRITE = false: len[i] = length of read i's merged row; true: the row written at optr[i].
// One merge loop per read -- but a lane walking its own two rows in memory touches, with its 63 neighbours, 64 cache lines
// per load.  A wavefront therefore stages the rows of its 64 reads in LDS (wave_copy_rows: whole lines), every lane merges
// its read there, and the merged rows -- one contiguous range of the output -- go back in whole lines.  64 reads whose rows
// exceed the staging area (kMergeCap targets per strand) merge in memory.
constexpr int kMergeCap = 1280;
constexpr int kMergeThreads = 128;
template <bool WRITE, class GetA, class GetB, class Put>
__device__ __forceinline__ uint32_t merge_one(uint32_t la, uint32_t lb, bool intersect, GetA geta, GetB getb, Put put) {
  uint32_t a = 0, b = 0, k = 0;
  if (la && lb) {
    uint32_t x = geta(0), y = getb(0);
    for (;;) {
      if (x == y) {
        if (WRITE) put(k, x);
        ++k, ++a, ++b;
        if (a == la || b == lb) break;
        x = geta(a), y = getb(b);
      } else if (x < y) {
        if (!intersect) {
          if (WRITE) put(k, x);
          ++k;
        }
        if (++a == la) break;
        x = geta(a);
      } else {
        if (!intersect) {
          if (WRITE) put(k, y);
          ++k;
        }
        if (++b == lb) break;
        y = getb(b);
      }
    }
  }
  if (!intersect) {
    for (; a < la; ++a, ++k)
      if (WRITE) put(k, geta(a));
    for (; b < lb; ++b, ++k)
      if (WRITE) put(k, getb(b));
  }
  return k;
}
template <bool WRITE>
__global__ __launch_bounds__(kMergeThreads) void k_merge_rows(const uint64_t *aptr, const uint32_t *atgt, uint64_t na,
                                                              const uint64_t *bptr, const uint32_t *btgt, uint64_t nb,
                                                              uint64_t n, int intersect, uint32_t *len, const uint64_t *optr,
                                                              uint32_t *out) {
  __shared__ uint32_t sh[kMergeThreads / 64][(WRITE ? 4 : 2) * kMergeCap];
  const uint32_t lane = threadIdx.x & 63;
  uint32_t *sa = sh[threadIdx.x >> 6], *sb = sa + kMergeCap, *so = sa + 2 * kMergeCap;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (uint64_t i0 = wave * 64; i0 < n; i0 += n_waves * 64) {
    const uint64_t i = i0 + lane;
    uint64_t a0 = 0, b0 = 0;
    uint32_t la = 0, lb = 0;
    if (i < n && i < na) a0 = aptr[i], la = (uint32_t)(aptr[i + 1] - a0);
    if (i < n && i < nb) b0 = bptr[i], lb = (uint32_t)(bptr[i + 1] - b0);
    uint32_t ea = la, eb = lb;  // inclusive prefix sums over the lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t ua = __shfl_up(ea, o), ub = __shfl_up(eb, o);
      if (lane >= (uint32_t)o) ea += ua, eb += ub;
    }
    const uint32_t ta = __shfl(ea, 63), tb = __shfl(eb, 63);
    uint32_t k;
    if (ta <= (uint32_t)kMergeCap && tb <= (uint32_t)kMergeCap) {  // (wave-uniform)
      const uint32_t oa = ea - la, ob = eb - lb;
      wave_copy_rows(atgt, a0, sa, oa, la);
      wave_copy_rows(btgt, b0, sb, ob, lb);
      // (the rows a lane merges were written by other lanes of its wavefront)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const uint64_t o_first = WRITE ? optr[i0] : 0;
      const uint32_t oo = WRITE && i < n ? (uint32_t)(optr[i] - o_first) : 0u;
      k = merge_one<WRITE>(la, lb, intersect != 0, [&](uint32_t j) { return sa[oa + j]; }, [&](uint32_t j) { return sb[ob + j]; },
                           [&](uint32_t j, uint32_t v) { so[oo + j] = v; });
      if (WRITE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint64_t i1 = i0 + 64 < n ? i0 + 64 : n;
        const uint32_t total = (uint32_t)(optr[i1] - o_first);
        for (uint32_t e = lane; e < total; e += 64) out[o_first + e] = so[e];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (the staging area is written again in the next round)
      __builtin_amdgcn_wave_barrier();
    } else {
      uint32_t *dst = WRITE && i < n ? out + optr[i] : nullptr;
      k = merge_one<WRITE>(la, lb, intersect != 0, [&](uint32_t j) { return atgt[a0 + j]; }, [&](uint32_t j) { return btgt[b0 + j]; },
                           [&](uint32_t j, uint32_t v) { dst[j] = v; });
    }
    if (!WRITE && i < n) len[i] = k;
  }
}

// ---- collapse (include/mSWEEP_alignment.hpp:137-215) --------------------------------------------------------------------
__global__ void k_flag_aligned(const uint64_t *ptr, uint64_t n, uint32_t *flag) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    flag[i] = ptr[i + 1] != ptr[i];
}
// the aligned reads keyed by the reference's hash of their ascending target ids (:152-156)
__global__ void k_hash_reads(const uint64_t *ptr, const uint32_t *tgt, uint64_t n, const uint64_t *idx, uint64_t *keys,
                             uint32_t *ids) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t b = ptr[i], e = ptr[i + 1];
    if (b == e) continue;
    uint64_t h = 0;
    for (uint64_t k = b; k < e; ++k) h ^= (uint64_t)tgt[k] + 0x517cc1b727220a95ull + (h << 6) + (h >> 2);
    keys[idx[i]] = h;
    ids[idx[i]] = (uint32_t)i;
  }
}
__global__ void k_ec_heads(const uint64_t *keys, uint64_t K, uint32_t *head) {
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < K; k += (uint64_t)gridDim.x * blockDim.x)
    head[k] = k == 0 || keys[k] != keys[k - 1];
}
// per class: where its reads start, the length of its representative's row (the first read of the class, :199-203)
__global__ void k_ec_meta(const uint32_t *head, const uint64_t *eidx, const uint32_t *ids, const uint64_t *ptr, uint64_t K,
                          uint64_t *rptr, uint32_t *tlen) {
  for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < K; k += (uint64_t)gridDim.x * blockDim.x) {
    if (!head[k]) continue;
    const uint64_t e = eidx[k];
    const uint32_t rep = ids[k];
    rptr[e] = k;
    tlen[e] = (uint32_t)(ptr[rep + 1] - ptr[rep]);
  }
}
__global__ __launch_bounds__(256) void k_ec_rows(const uint64_t *rptr, const uint32_t *ids, const uint64_t *ptr,
                                                 const uint32_t *tgt, const uint64_t *tptr, uint64_t E, uint64_t *counts,
                                                 uint32_t *out) {
  const uint64_t t0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t e0 = t0 & ~63ull; e0 < E; e0 += step) {
    const uint64_t e = e0 + (threadIdx.x & 63);
    uint64_t so = 0, d0 = 0;
    uint32_t len = 0;
    if (e < E) {
      counts[e] = rptr[e + 1] - rptr[e];
      so = ptr[ids[rptr[e]]];
      d0 = tptr[e];
      len = (uint32_t)(tptr[e + 1] - d0);
    }
    wave_copy_rows(tgt, so, out, d0, len);
  }
}

// ---- host: pinned staging of the text on its way to the device (host_reader.inc) ------------------------------------------
// Every reader thread streams its own interleaved share of the file -- blocks k, k + T, k + 2 T, ... -- through two
// pinned sub-buffers of its own: pread, hipMemcpyAsync, the next block into the other sub-buffer while this one
// travels.  No meeting point between the threads (a gang that filled ONE staging chunk together and was started per
// chunk paid ~0.3 ms of thread start-up per 32 MB).  Owned by the handle: pinning costs ~0.4 ms per MB, once.
struct TextStager {
  size_t block = 0, threads = 0;   // bytes per sub-buffer (MSWEEP_READER_BLOCK_MB, developer switch), threads it is laid out for
  void *pinned = nullptr;          // 2 * threads * block bytes
  std::vector<hipEvent_t> ev;      // [2 * threads]: the copy that last read a sub-buffer
  hipStream_t copy = nullptr;      // the text travels on a stream of its own: the next strand's under this strand's kernels
  void ready(size_t T) {
    if (!block) {
      const char *e = getenv("MSWEEP_READER_BLOCK_MB");
      const long mb = e ? atol(e) : 0;
      block = (size_t)(mb >= 1 && mb <= 256 ? mb : 4) << 20;  // (1-2 MB: the fixed cost per copy shows; 4-8 MB alike)
    }
    if (T > threads) {
      release();
      MSW_HIP(hipHostMalloc(&pinned, 2 * T * block, hipHostMallocDefault));
      ev.assign(2 * T, nullptr);
      for (auto &e : ev) MSW_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      threads = T;
    }
    if (!copy) MSW_HIP(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
  }
  char *sub(size_t thread, int slot) const { return static_cast<char *>(pinned) + (2 * thread + slot) * block; }
  void release() {
    if (pinned) (void)hipHostFree(pinned);
    for (auto e : ev)
      if (e) (void)hipEventDestroy(e);
    pinned = nullptr;
    ev.clear();
    threads = 0;
  }
  ~TextStager() {
    release();
    if (copy) (void)hipStreamDestroy(copy);
  }
};

// Device memory of the reader, kept on the handle between calls.  Two facts of the runtime shape it: hipFree waits for
// EVERY stream of the device (also the copy stream that carries the next strand's text), and memory given back is
// scrubbed by the driver before it is handed out again -- ~10 GB of temporaries at cfg3 stalled the allocations of
// the likelihood build that follows by 0.2 s (tools/reader_probe.py: 0.230 s, 0.021 s after a pause).  So nothing is
// freed while the handle lives: blocks handed back during a call become reusable from the NEXT call on (no ordering
// question between the streams of one call), a second read of the same size allocates nothing, and msw_core_trim /
// msw_core_destroy give the memory back.
struct ReaderPool {
  std::mutex mu;
  std::multimap<size_t, void *> idle;        // reusable blocks by size
  std::unordered_map<void *, size_t> live;   // blocks handed out (and those handed back during this call)
  std::vector<void *> returned;
  size_t bytes = 0;
  void *take(size_t need) {
    {
      std::lock_guard<std::mutex> g(mu);
      auto it = idle.lower_bound(need);
      if (it != idle.end() && it->first <= need + need / 2 + (1u << 20)) {
        void *p = it->second;
        live[p] = it->first;
        idle.erase(it);
        return p;
      }
    }
    void *p = nullptr;
    MSW_HIP(hipMalloc(&p, need));
    std::lock_guard<std::mutex> g(mu);
    live[p] = need;
    bytes += need;
    return p;
  }
  void give(void *p) {
    std::lock_guard<std::mutex> g(mu);
    returned.push_back(p);
  }
  void recycle() {  // between calls: what was handed back may be handed out again
    std::lock_guard<std::mutex> g(mu);
    for (void *p : returned) {
      auto it = live.find(p);
      if (it == live.end()) continue;
      idle.emplace(it->second, p);
      live.erase(it);
    }
    returned.clear();
  }
  size_t idle_bytes() {
    std::lock_guard<std::mutex> g(mu);
    size_t b = 0;
    for (auto &kv : idle) b += kv.first;
    return b;
  }
  void trim() {  // every idle block back to the device
    recycle();
    std::lock_guard<std::mutex> g(mu);
    for (auto &kv : idle) {
      (void)hipFree(kv.second);
      bytes -= kv.first;
    }
    idle.clear();
  }
  ~ReaderPool() {
    trim();
    for (auto &kv : live) (void)hipFree(kv.first);
  }
};

}  // namespace msw
