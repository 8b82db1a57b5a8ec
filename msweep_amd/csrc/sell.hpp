// sell.hpp -- device-resident CSR-of-ECs likelihood in SELL-64 form.
#pragma once
#include "common.hpp"

namespace msw {

// ---------------------------------------------------------------------------------------
// Device-resident CSR-of-ECs likelihood in SELL-64 form.
//   ECs are permuted: first the "long" ECs (more than kLongRow cells, kept as plain CSR and
//   swept by a whole workgroup), then all others sorted by descending cell count and cut into
//   slices of 64 consecutive ECs.  A slice stores its records column-major
//   (rec[(off + k) * 64 + lane] = k-th cell of the slice's lane-th EC), padded to the slice's
//   longest EC with sentinel records (group id n_groups + lane, whose e_g is 0).  A wavefront
//   sweeps one slice: lane l streams EC l's cells with perfectly coalesced loads.
//
//   A record carries the two table positions of its cell as ready-made BYTE OFFSETS, so the
//   sweeps spend one shift / one mask per lookup instead of unpack + scale + base:
//     hi = bhi + 8 * group   (byte offset of e_g in pass B's LDS image; pass A doubles it:
//                             2 * hi = 2 * bhi + 16 * group addresses its 16-byte {e, w} entries)
//     lo = 16 * entry        (byte offset of the cell's 16-byte entry in the SLOT AREA: the LUT
//                             slots some cell refers to, compacted, followed by 8 bank-private
//                             replicas of the hottest ones -- see upload_sell)
//   narrow record (4 B, ENC 0): hi << shift | lo, with lo < 2^(shift-1) so that rec >> (shift-1) == 2*hi;
//   wide record (8 B, ENC 1): {hi, lo};
//   index record (4 B, ENC 2): group << shift | entry -- for slot areas that do not fit LDS beside the group
//     vectors (thousands of used (group size, hit count) pairs: real groupings have sizes up to hundreds) and
//     whose byte offsets would need 8-byte records: two VALU operations per gather instead of one, half the
//     record stream.  Always goes with the HYBRID slot area: the entries are ordered by use, the first
//     n_tab_lds of them are kept in LDS as well, and every slice of up to 16 rows is cut into a HOT segment
//     (rows [0, nhot): every lane's cell refers to an LDS-resident entry -- the straight-line LDS code) and a
//     COLD one (rows [nhot, len), at most kColdRows: gathered from the table in memory, issued before the hot
//     segment is worked through and consumed after it).  The split is wave-uniform (one byte per slice,
//     slice_hot): no per-gather branch.  Slices whose ECs hold more cold cells than kColdRows take every table
//     entry from memory (nhot = 0), like the streaming slices of more than 16 rows.
//     The rows of a HOT segment carry their entry pre-multiplied by 16 (round 4): group << shiftH | 16 * entry, shiftH =
//     bits of the LDS image of the table -- one VALU operation for the slot gather, as with byte-offset records (hot
//     entries are below n_tab_lds, so this form always fits 32 bits); cold rows, whole-memory slices, streaming slices and
//     long ECs keep group << shift | entry, whose entry index may need all the bits the group leaves.
//   value record (12 B, ENC 3): {hi, T} -- the cell's log-likelihood itself instead of a table position, for
//     matrices whose listed values are (nearly) all different (a dense `logl` with continuous values handed to
//     msw_core_set_dense_logl: one table slot per cell would cost 16 bytes gathered + 40 bytes rebuilt per cell
//     and iteration); the sweeps form exp(a (T - tref)) per cell.  Stored row by row: 64 hi words, then the
//     row's 64 doubles (kValRowWords dwords per row of 64 cells).
// ---------------------------------------------------------------------------------------
constexpr uint32_t kSentinels = 64;  // one sentinel group per lane: padding never shares an address

// ---------------------------------------------------------------------------------------
// Slice classes (round 3): an EC of up to 16 cells takes ONE lane of its slice; an EC of 17..1024 cells takes
// m = 2, 4, ... 64 lanes (the smallest m with cells <= 16 m; cell k of the EC in sub-lane k mod m, row k / m), so
// that every slice has at most 16 rows and stays on the sweeps' register path: no second walk over the records
// and no second gathers in pass B (the streaming path such ECs took before ran pass B at half the per-cell
// rate), the m partial row sums meet in log2 m steps (DPP inside a row of 16 lanes), and a slice holds 64 / m
// ECs -- the chain of a wavefront that has few slices is m times shorter.  The ECs are sorted by descending
// length, so the classes are contiguous: class c = 0..6 <-> m = 64 >> c.  s0 / p0: first slice / first EC position
// (within the sliced part of the permuted order) of every class.
// ---------------------------------------------------------------------------------------
// rows a slice lane holds at most = records per EC lane the sweeps keep in registers (sweep_kernels.hpp kRegCells).
// 16 by default; 8 (a developer build, tools/ab_build.py) halves the sweeps' record buffers and kept values -- ECs of
// 9..16 cells then take two lanes -- which is what 16 wavefronts per workgroup of pass B need to stay in registers.
#ifndef MSW_REG_CELLS
#define MSW_REG_CELLS 16
#endif
constexpr int kRowsPerLane = MSW_REG_CELLS;
static_assert(kRowsPerLane == 16 || kRowsPerLane == 12 || kRowsPerLane == 8, "slices of at most 16, 12 or 8 rows");
constexpr int kSliceClasses = 7;
constexpr uint32_t kMaxLgm = kSliceClasses - 1;  // class c: 2^(kMaxLgm - c) lanes per EC
struct SliceClasses {
  uint32_t s0[kSliceClasses + 1], p0[kSliceClasses + 1];
};
// class of an EC of `len` cells (multilane = false: every EC one lane -- the developer switch MSWEEP_MULTILANE=0):
// the smallest m = 2^lgm with len <= 16 m
__host__ __device__ inline int slice_class_of(uint32_t len, bool multilane) {
  if (!multilane) return kSliceClasses - 1;
  uint32_t lgm = 0;
  while (lgm < kMaxLgm && len > ((uint32_t)kRowsPerLane << lgm)) ++lgm;
  return (int)(kMaxLgm - lgm);
}
struct SliceGeo {
  uint32_t lgm, ec0, nec;  // log2 lanes per EC; first EC position of the slice; ECs in it
};
__host__ __device__ inline SliceGeo slice_geo(const SliceClasses &C, uint32_t s) {
  uint32_t c = 0;
#pragma unroll
  for (int i = 1; i < kSliceClasses; ++i) c += s >= C.s0[i];
  const uint32_t lgm = kMaxLgm - c, per = 64u >> lgm;
  const uint32_t ec0 = C.p0[c] + (s - C.s0[c]) * per;
  const uint32_t left = C.p0[c + 1] - ec0;
  return SliceGeo{lgm, ec0, left < per ? left : per};
}
// slice and lane group of the EC at position q of the sliced part
__host__ __device__ inline void slice_of_position(const SliceClasses &C, uint32_t q, uint32_t &s, uint32_t &lgm,
                                                  uint32_t &first_lane) {
  uint32_t c = 0;
#pragma unroll
  for (int i = 1; i < kSliceClasses; ++i) c += q >= C.p0[i];
  lgm = kMaxLgm - c;
  const uint32_t per = 64u >> lgm, d = q - C.p0[c];
  s = C.s0[c] + d / per;
  first_lane = (d % per) << lgm;
}

struct SellDev {
  const uint32_t *rec;        // SELL records
  const uint32_t *slice_off;  // [nslices + 1], in units of 64 records
  const uint32_t *long_ptr;   // [n_long + 1] offsets into rec_long
  const uint32_t *rec_long;   // records of the long ECs (CSR)
  const uint32_t *perm;       // [E] permuted position -> original EC index
  const double *cvec;         // [E] EC multiplicities, permuted order
  const uint8_t *c8;          // [E] the same as a byte; kC8Escape = not a small integer, read cvec
  const uint8_t *c8s;         // [64 * nslices] the byte of the EC every slice lane works for (0: none): pass B's stream
  const uint32_t *area_slot;  // [n_area] LUT slot held by each 16-byte entry of the slot area
  const uint8_t *slice_hot;   // [nslices] index records: rows of the slice's hot segment (0: all from memory)
  uint32_t nslices, n_long, n_ecs, n_groups, n_lut, n_area;
  uint32_t n_tab_lds;         // slot-area entries held in LDS (all of them, the hot head of a hybrid area, or 0)
  uint32_t shift, mask, bhi;  // record encoding (narrow: shift / lo mask; all: bhi = LDS byte offset of e_g[0])
  uint32_t bhiA;              // LDS byte offset of pass A's {e, w}[0] (2 * bhi but for index records)
  uint32_t shiftH, maskH;     // index records, rows of a hot segment: group << shiftH | 16 * entry
  const double *lut_area;     // table value of every slot-area entry (utility kernels: the value of a cell)
  SliceClasses cls;           // slice classes: lanes per EC (above)
};

constexpr uint32_t kC8Escape = 255;
// weights of the greedy LDS-bank scheduler (both packers): a cell scores a weight for every access it
// can make without a bank conflict in the current step -- column-sum atomic, {e,w} b128, slot b128, e_g b64
#ifndef MSW_W_AT
#define MSW_W_AT 8
#define MSW_W_EW 6
#define MSW_W_XT 4
#define MSW_W_E 2
#endif
constexpr int kLongRow = 64 * kRowsPerLane;  // 1024: ECs with more cells than this take the wavefront-per-EC path (64 lanes x 16 rows)
constexpr int kLongRowOneLane = 256;  // ... when every EC takes one lane (MSWEEP_MULTILANE=0: slices of up to 256 rows)
#ifndef MSW_COLD_ROWS
#define MSW_COLD_ROWS 2
#endif
constexpr int kColdRows = MSW_COLD_ROWS;  // index records: rows of a slice's cold segment, 2 or 4 (beyond: the whole slice from memory)
static_assert(kColdRows == 2 || kColdRows == 4, "cold segments are cut in pairs of rows");
constexpr uint32_t kGeoHotShift = 27;  // slice geometry in LDS: rows of the hot segment above the slice offset
// ... and above the slice's rows (<= kLongRow): log2 lanes per EC, ECs in the slice (slice classes)
constexpr uint32_t kGeoLgmShift = 11, kGeoNecShift = 14;

// what a sweep needs to decode a record (SGPRs)
struct RecDec {
  uint32_t shift, mask, bhi, bhiA;
  uint32_t shiftH, maskH;  // index records: the hot rows' form
};
enum { kEncNarrow = 0, kEncWide = 1, kEncIndex = 2, kEncValue = 3 };
// A slice has as many rows as its longest EC has cells (until round 3 that was rounded up to an even number: 5 % of
// cfg3's rows, 10 % of cfg5's); the sweeps, which work through a slice two rows at a time, make the missing last row
// of an odd slice in registers (sweep_kernels.hpp load_slice).  MSW_ODD_SLICES=0: the padded layout, for A/B timing.
#ifndef MSW_ODD_SLICES
#define MSW_ODD_SLICES 1
#endif
__host__ __device__ constexpr bool odd_slices(int) { return MSW_ODD_SLICES != 0; }
constexpr uint32_t kValRowWords = 192;  // value records: dwords per row of 64 cells (64 hi words + 64 doubles)
struct ValRec {
  uint32_t hi;  // byte offset of e_g (8 * group: no table in front of the group vectors)
  double t;     // the cell's log-likelihood
};
template <int ENC>
struct Rec;
//   e_off : byte offset of e_g in pass B's LDS image      ew_off: of {e, w}_g in pass A's
//   t_off : byte offset of the cell's entry in the slot area (LDS image and the tables in memory alike)
template <>
struct Rec<kEncNarrow> {
  using T = uint32_t;
  static __device__ __forceinline__ T load(const uint32_t *p, size_t i) { return p[i]; }
  static __device__ __forceinline__ uint32_t e_off(T r, const RecDec &d) { return r >> d.shift; }
  static __device__ __forceinline__ uint32_t ew_off(T r, const RecDec &d) { return r >> (d.shift - 1); }
  static __host__ __device__ __forceinline__ uint32_t t_off(T r, const RecDec &d) { return r & d.mask; }
  static __host__ __device__ __forceinline__ uint32_t grp(T r, const RecDec &d) { return ((r >> d.shift) - d.bhi) >> 3; }
  static __host__ __device__ __forceinline__ T make(uint32_t g, uint32_t entry, const RecDec &d) {
    return ((d.bhi + 8u * g) << d.shift) | (16u * entry);
  }
};
template <>
struct Rec<kEncWide> {
  using T = uint2;
  static __device__ __forceinline__ T load(const uint32_t *p, size_t i) {
    return reinterpret_cast<const uint2 *>(p)[i];
  }
  static __device__ __forceinline__ uint32_t e_off(T r, const RecDec &) { return r.x; }
  static __device__ __forceinline__ uint32_t ew_off(T r, const RecDec &) { return r.x << 1; }
  static __host__ __device__ __forceinline__ uint32_t t_off(T r, const RecDec &) { return r.y; }
  static __host__ __device__ __forceinline__ uint32_t grp(T r, const RecDec &d) { return (r.x - d.bhi) >> 3; }
  static __host__ __device__ __forceinline__ T make(uint32_t g, uint32_t entry, const RecDec &d) {
    return make_uint2(d.bhi + 8u * g, 16u * entry);
  }
};
template <>
struct Rec<kEncIndex> {
  using T = uint32_t;
  static __device__ __forceinline__ T load(const uint32_t *p, size_t i) { return p[i]; }
  static __device__ __forceinline__ uint32_t e_off(T r, const RecDec &d) { return ((r >> d.shift) << 3) + d.bhi; }
  static __device__ __forceinline__ uint32_t ew_off(T r, const RecDec &d) { return ((r >> d.shift) << 4) + d.bhiA; }
  static __host__ __device__ __forceinline__ uint32_t t_off(T r, const RecDec &d) { return (r & d.mask) << 4; }
  static __host__ __device__ __forceinline__ uint32_t grp(T r, const RecDec &d) { return r >> d.shift; }
  static __host__ __device__ __forceinline__ T make(uint32_t g, uint32_t entry, const RecDec &d) { return (g << d.shift) | entry; }
  // the rows of a hot segment (file header): entry pre-multiplied by 16
  static __device__ __forceinline__ uint32_t e_off_h(T r, const RecDec &d) { return ((r >> d.shiftH) << 3) + d.bhi; }
  static __device__ __forceinline__ uint32_t ew_off_h(T r, const RecDec &d) { return ((r >> d.shiftH) << 4) + d.bhiA; }
  static __host__ __device__ __forceinline__ uint32_t t_off_h(T r, const RecDec &d) { return r & d.maskH; }
  static __host__ __device__ __forceinline__ uint32_t grp_h(T r, const RecDec &d) { return r >> d.shiftH; }
  static __host__ __device__ __forceinline__ T make_h(uint32_t g, uint32_t entry, const RecDec &d) { return (g << d.shiftH) | (entry << 4); }
};
template <>
struct Rec<kEncValue> {
  using T = ValRec;
  static __device__ __forceinline__ T load(const uint32_t *p, size_t i) {
    const size_t w = (i >> 6) * kValRowWords;
    return ValRec{p[w + (i & 63)], reinterpret_cast<const double *>(p + w + 64)[i & 63]};
  }
  static __device__ __forceinline__ void store(uint32_t *p, size_t i, T r) {
    const size_t w = (i >> 6) * kValRowWords;
    p[w + (i & 63)] = r.hi;
    reinterpret_cast<double *>(p + w + 64)[i & 63] = r.t;
  }
  static __device__ __forceinline__ uint32_t e_off(T r, const RecDec &) { return r.hi; }
  static __device__ __forceinline__ uint32_t ew_off(T r, const RecDec &) { return r.hi << 1; }
  static __host__ __device__ __forceinline__ uint32_t grp(T r, const RecDec &) { return r.hi >> 3; }
};
// the three byte offsets of a record; HOT = a row of an index-record slice's hot segment (every other encoding and
// row: one form)
template <int ENC, bool HOT>
__device__ __forceinline__ uint32_t rec_e_off(typename Rec<ENC>::T r, const RecDec &d) {
  if constexpr (ENC == kEncIndex && HOT) return Rec<ENC>::e_off_h(r, d);
  else return Rec<ENC>::e_off(r, d);
}
template <int ENC, bool HOT>
__device__ __forceinline__ uint32_t rec_ew_off(typename Rec<ENC>::T r, const RecDec &d) {
  if constexpr (ENC == kEncIndex && HOT) return Rec<ENC>::ew_off_h(r, d);
  else return Rec<ENC>::ew_off(r, d);
}
template <int ENC, bool HOT>
__device__ __forceinline__ uint32_t rec_t_off(typename Rec<ENC>::T r, const RecDec &d) {
  if constexpr (ENC == kEncIndex && HOT) return Rec<ENC>::t_off_h(r, d);
  else if constexpr (ENC == kEncValue) return 0u;
  else return Rec<ENC>::t_off(r, d);
}
// dwords of a record array of n cells
__host__ __device__ inline size_t rec_words(int enc, size_t n) {
  return enc == kEncValue ? ((n + 63) / 64) * kValRowWords : (enc == kEncWide ? 2 * n : n);
}
__host__ __device__ inline RecDec rec_dec(const SellDev &S) { return RecDec{S.shift, S.mask, S.bhi, S.bhiA, S.shiftH, S.maskH}; }
// is row k of slice s a row of a hot segment (index records: the other record form)?
template <int ENC>
__device__ __forceinline__ bool rec_row_hot(const SellDev &S, uint32_t s, uint32_t k) {
  if constexpr (ENC == kEncIndex) return k < (uint32_t)S.slice_hot[s];
  else return false;
}
// group id / slot-area entry / log-likelihood of a record (utility kernels; the sweeps never form them)
template <int ENC>
__device__ __forceinline__ uint32_t rec_grp(const SellDev &S, typename Rec<ENC>::T r, bool hot = false) {
  if constexpr (ENC == kEncIndex) return hot ? Rec<ENC>::grp_h(r, rec_dec(S)) : Rec<ENC>::grp(r, rec_dec(S));
  else return Rec<ENC>::grp(r, rec_dec(S));
}
template <int ENC>
__device__ __forceinline__ uint32_t rec_entry(const SellDev &S, typename Rec<ENC>::T r, bool hot = false) {
  if constexpr (ENC == kEncIndex) return (hot ? Rec<ENC>::t_off_h(r, rec_dec(S)) : Rec<ENC>::t_off(r, rec_dec(S))) >> 4;
  else return Rec<ENC>::t_off(r, rec_dec(S)) >> 4;
}
template <int ENC>
__device__ __forceinline__ double rec_value(const SellDev &S, typename Rec<ENC>::T r, bool hot = false) {
  if constexpr (ENC == kEncValue) return r.t;
  else return S.lut_area[rec_entry<ENC>(S, r, hot)];
}
template <int ENC>
__device__ __forceinline__ uint32_t rec_idx(const SellDev &S, typename Rec<ENC>::T r, bool hot = false) {
  return S.area_slot[rec_entry<ENC>(S, r, hot)];
}

// Visit the cells of the EC at permuted position p (utility kernels only): f(group id, log-likelihood).
template <int ENC, class F>
__device__ __forceinline__ void for_each_cell(const SellDev &S, uint32_t p, F f) {
  using R = Rec<ENC>;
  if (p < S.n_long) {
    for (uint32_t k = S.long_ptr[p]; k < S.long_ptr[p + 1]; ++k) {
      const typename R::T r = R::load(S.rec_long, k);
      f(rec_grp<ENC>(S, r), rec_value<ENC>(S, r));
    }
  } else {
    uint32_t s, lgm, l0;
    slice_of_position(S.cls, p - S.n_long, s, lgm, l0);
    const uint32_t o0 = S.slice_off[s], len = S.slice_off[s + 1] - o0;
    for (uint32_t k = 0; k < len; ++k) {
      const bool hot = rec_row_hot<ENC>(S, s, k);
      for (uint32_t t = 0; t < (1u << lgm); ++t) {
        const typename R::T r = R::load(S.rec, ((size_t)o0 + k) * 64 + l0 + t);
        const uint32_t g = rec_grp<ENC>(S, r, hot);
        if (g < S.n_groups) f(g, rec_value<ENC>(S, r, hot));
      }
    }
  }
}

// The cells of the EC at permuted position p, one per lane and step, for a whole wavefront:
// f(group id, log-likelihood).  Sentinel padding is skipped.  (Rare-path code: the guarded ECs.)
template <int ENC, class F>
__device__ __forceinline__ void wave_cells(const SellDev &S, uint32_t p, uint32_t lane, F f) {
  using R = Rec<ENC>;
  if (p < S.n_long) {
    const uint32_t k0 = S.long_ptr[p], k1 = S.long_ptr[p + 1];
    for (uint32_t k = k0 + lane; k < k1; k += 64) {
      const typename R::T r = R::load(S.rec_long, k);
      f(rec_grp<ENC>(S, r), rec_value<ENC>(S, r));
    }
  } else {
    uint32_t s, lgm, l0;
    slice_of_position(S.cls, p - S.n_long, s, lgm, l0);
    const uint32_t o0 = S.slice_off[s], len = S.slice_off[s + 1] - o0;
    // a lane per (row, sub-lane) of the EC
    for (uint32_t i = lane; i < (len << lgm); i += 64) {
      const typename R::T r = R::load(S.rec, ((size_t)o0 + (i >> lgm)) * 64 + l0 + (i & ((1u << lgm) - 1u)));
      const bool hot = rec_row_hot<ENC>(S, s, i >> lgm);
      const uint32_t g = rec_grp<ENC>(S, r, hot);
      if (g < S.n_groups) f(g, rec_value<ENC>(S, r, hot));
    }
  }
}

// ---------------------------------------------------------------------------------------
// Guarded ECs (SURVEY.md 7.3 ii).  The sweeps form Z_j = p0 * U + sum_listed e_g (x_gj - p0): when
// the listed groups hold nearly all of U and their cells sit far below the background (x << p0:
// an isolate's read that hits the dominant lineage with one k-mer, priors << 1) the two parts cancel
// and Z_j, r_j = c_j / Z_j and the listed groups' column sums lose their digits -- or their sign.
// An EC whose Z_j comes out below 2^-8 of the background sum (the unguarded relative error is thus
// at most 2^-45; with the reference's WOR21 tables, whose values stay within e^-1 of log(zi) up to
// groups of 400 sequences, no EC comes near it) is set aside by the sweep (its position appended to the workgroup's list) and
// evaluated WITHOUT the background trick at the end of the workgroup, a wavefront per EC:
// Z_j = sum_listed e_g x_gj + p0 * (sum over the groups NOT listed, one by one), and every group's
// share of c_j added directly.  O(G) per guarded EC; none exist in ordinary inputs.
// ---------------------------------------------------------------------------------------
constexpr double kGuardRatio = 0x1p-8;
struct GuardDev {
  uint32_t *list;          // [workgroups * cap] positions of the ECs set aside, per workgroup
  uint32_t *bits;          // [workgroups * 16 * words] one bitmap of listed groups per wavefront
  unsigned long long *tail;  // [2 * G] the guarded ECs' shares per group, two fixed-point limbs (pass B)
  const double *lut_area;  // table value of every slot-area entry
  int *err;                // set when an EC has no probability under any group
  unsigned long long *visits;  // ECs pass B has taken through the guard path so far (reporting: msw_core_guarded_visits)
  uint32_t cap, words;
};

// LDS image of the sweeps (byte offsets; bhi = sell_bhi(n_tab_lds)):
//   [0, 16 * n_tab_lds)                 slot area: 16-byte per-slot entries of the pass -- all of them (tlds), the
//                                       hot head of a hybrid area (index records), or none
//   pass A: [bhiA, bhiA + 16 * Gp)      {e_g, wc_g}   (glds; bhiA = 2 * bhi, index records: bhi)
//   pass B: [bhi, bhi + 8 * Gp) e_g,  column sums pass_acc_off() bytes behind      (glds);
//           with too many groups for both (GMODE 3) the column sums alone, e_g gathered from memory
//   then 32 doubles of reduction scratch and 8 KB of slice geometry.   Gp = G + kSentinels
__host__ __device__ inline uint32_t sell_bhi(uint32_t n_tab_lds) {
  return n_tab_lds ? ((16u * n_tab_lds + 255u) & ~255u) : 0u;
}
constexpr uint32_t kGeoStride = 66 * 8;  // bytes of slice geometry per wavefront
constexpr uint32_t kAccFixed = 65528;  // largest 8-byte-aligned ds immediate offset
// byte distance from e_g to the column sum of the same group in pass B's LDS image
__host__ __device__ inline uint32_t pass_acc_off(int gmode, uint32_t G) {
  return gmode >= 3 ? 0u : (gmode == 2 ? kAccFixed : 8u * (G + kSentinels));  // 3, 4: the sums take e_g's place
}
// pass B mode 4 (any number of groups): the sweep is run once per range of kRangeGroups groups,
// each run accumulating only its range's column sums in LDS
constexpr uint32_t kRangeGroups = 16384;
struct RangeB {
  uint32_t g0, n;  // groups [g0, g0 + n) of this run
  int first;       // the run that also delivers the per-EC ELBO terms
};
// index = index records (pass A's group entries right behind the table instead of at twice pass B's offset)
__host__ __device__ inline size_t pass_scratch_off(int gmode, uint32_t n_tab_lds, uint32_t G, bool passA, bool index) {
  const size_t bhi = sell_bhi(n_tab_lds), Gp = (size_t)G + kSentinels;
  if (gmode == 0 || (passA && gmode >= 3)) return bhi;  // pass A of modes 3, 4 gathers {e, w} from memory
  if (passA) return (index ? bhi : 2 * bhi) + 16 * Gp;
  if (gmode == 4) return bhi + 8 * (size_t)kRangeGroups;
  return bhi + pass_acc_off(gmode, G) + 8 * Gp;
}
constexpr size_t kPassTailBytes = 32 * sizeof(double) + 16 * kGeoStride;  // scratch + per-wave slice geometry
__host__ __device__ inline size_t pass_lds_bytes(int gmode, uint32_t n_tab_lds, uint32_t G, bool passA, bool index) {
  // + per-wave slice geometry (sweep_kernels.hpp SliceStream): 16 waves x (64 + 2) pairs
  return pass_scratch_off(gmode, n_tab_lds, G, passA, index) + kPassTailBytes;
}

}  // namespace msw
