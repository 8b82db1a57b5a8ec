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
//   longest EC with a sentinel record (group id == n_groups, whose e_g is 0).  A wavefront
//   sweeps one slice: lane l streams EC l's cells with perfectly coalesced loads.
//   A record is (lutidx << 16 | grp) when both fit 16 bits, else {grp, lutidx}.
// ---------------------------------------------------------------------------------------
struct SellDev {
  const uint32_t *rec;        // SELL records
  const uint32_t *slice_off;  // [nslices + 1], in units of 64 records
  const uint32_t *long_ptr;   // [n_long + 1] offsets into rec_long
  const uint32_t *rec_long;   // records of the long ECs (CSR)
  const uint32_t *perm;       // [E] permuted position -> original EC index
  const double *cvec;         // [E] EC multiplicities, permuted order
  uint32_t nslices, n_long, n_ecs, n_groups, n_lut;
};

constexpr int kLongRow = 256;  // ECs with more cells than this take the workgroup path

template <bool WIDE>
struct Rec;
template <>
struct Rec<false> {
  using T = uint32_t;
  static __device__ __forceinline__ T load(const uint32_t *p, size_t i) { return p[i]; }
  static __device__ __forceinline__ uint32_t grp(T r) { return r & 0xffffu; }
  static __device__ __forceinline__ uint32_t idx(T r) { return r >> 16; }
};
template <>
struct Rec<true> {
  using T = uint2;
  static __device__ __forceinline__ T load(const uint32_t *p, size_t i) {
    return reinterpret_cast<const uint2 *>(p)[i];
  }
  static __device__ __forceinline__ uint32_t grp(T r) { return r.x; }
  static __device__ __forceinline__ uint32_t idx(T r) { return r.y; }
};

// Visit the cells of the EC at permuted position p (utility kernels only).
template <bool WIDE, class F>
__device__ __forceinline__ void for_each_cell(const SellDev &S, uint32_t p, F f) {
  using R = Rec<WIDE>;
  if (p < S.n_long) {
    for (uint32_t k = S.long_ptr[p]; k < S.long_ptr[p + 1]; ++k) f(R::load(S.rec_long, k));
  } else {
    const uint32_t q = p - S.n_long, s = q >> 6, lane = q & 63;
    const uint32_t o0 = S.slice_off[s], len = S.slice_off[s + 1] - o0;
    for (uint32_t k = 0; k < len; ++k) {
      const typename R::T r = R::load(S.rec, ((size_t)o0 + k) * 64 + lane);
      if (R::grp(r) != S.n_groups) f(r);
    }
  }
}

// LDS bytes of the sweeps.  Pass A keeps {e_g, wc_g} pairs, pass B keeps e_g and the column-sum
// accumulators; both keep the per-pass table X[i] = exp(a*T_i) and the static table T[i] as two
// separate 8-byte arrays (consecutive slots in consecutive bank pairs: the 32-byte AoS entries
// of the first version mapped the hot slots onto 8 bank octets and cost 5x in LDS conflicts).
__host__ __device__ inline size_t pass_lds_bytes(bool glds, bool tlds, uint32_t G, uint32_t n_lut,
                                                 bool passA) {
  (void)passA;
  size_t b = 32 * sizeof(double);  // reduction scratch
  if (glds) b += 2 * ((size_t)G + 1) * sizeof(double);
  if (tlds) b += 2 * (size_t)n_lut * sizeof(double);
  return b;
}

}  // namespace msw
