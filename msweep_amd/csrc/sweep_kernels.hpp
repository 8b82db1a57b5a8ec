// sweep_kernels.hpp -- the two HBM-streaming sweeps of one RCG iteration over the SELL-64
// likelihood: pass A (natural-gradient norm) and pass B (softmax / column sums / ELBO).
//
// Both kernels: one persistent workgroup per CU (pass A 16 wavefronts, pass B 12; value records 8); its
// wavefronts take slices round-robin.  A slice of up to kRegCells cells per EC is held in registers: while
// slice s is being processed the records of the wave's next slice are already in flight (the record stream
// is the only HBM traffic; everything else is gathered from LDS).  Slice bounds are wave-uniform
// and kept in SGPRs (readfirstlane) so the cell loops are scalar branches, not exec-mask loops.
//
// Pass A is bound by its record stream, pass B by VALU issue + LDS gathers in balance, so the
// per-cell instruction count is what is optimised here: records carry ready-made LDS byte offsets
// (sell.hpp: one shift + one mask per cell), every per-slot quantity that does not depend on the
// group is tabulated once per pass by prepB_block (state_kernels.hpp), and padding records point
// at per-lane sentinel groups so that no per-cell test is needed.
//
// Template parameter ENC = the record encoding (sell.hpp): byte-offset records (the case above), 8-byte records,
// index records with the HYBRID slot area (a slice is cut into a hot segment served from LDS and a short cold one
// whose table entries come from memory), value records (the cell's log-likelihood inline, exp per cell).
#pragma once
#include <type_traits>

#include "device_util.hpp"
#include "sell.hpp"

namespace msw {

constexpr int kRegCells = kRowsPerLane;  // 16: cells per EC a wave keeps in registers (longer slices stream; sell.hpp)
#ifndef MSW_LONG_STEP
#define MSW_LONG_STEP 8
#endif
constexpr int kLongStep = MSW_LONG_STEP;  // records per lane and step on the wavefront-per-EC path (multiple of 4)
// tuning knobs (defaults measured on MI355X; tools/ab_build.py builds variants)
#ifndef MSW_REVERSE_B
#define MSW_REVERSE_B true
#endif
#ifndef MSW_B_KEEPN
#define MSW_B_KEEPN MSW_REG_CELLS
#endif
#ifndef MSW_PASSA_BATCH
#define MSW_PASSA_BATCH 4
#endif
// the first slice of every wavefront requested before the LDS fill (SliceStream::prime); MSW_NO_PRIME: A/B builds
#ifdef MSW_NO_PRIME
constexpr bool kPrimeFirstSlice = false;
#else
constexpr bool kPrimeFirstSlice = true;
#endif
#ifndef MSW_PASSB_BATCH
#define MSW_PASSB_BATCH 4
#endif
#ifndef MSW_LONG_KEEP
#define MSW_LONG_KEEP 1
#endif
// Column sums in 64-bit FIXED POINT (default): a cell adds rint(2^K * f_g * r_j * (x - p0)) with an INTEGER
// LDS atomic (f_g: e_g, or a power-of-two multiple of it for the groups far below the largest --
// device_util.hpp fx_factor); k_redfin turns the totals back into reads.  Integer addition is associative:
// the sums no longer depend on the order in which the wavefronts of a workgroup reach the atomics, so two
// runs of a solve are bit-identical (fp64 atomics: the stop test sits in their rounding noise and a
// 10 M-read run stopped at 209 or 210 iterations), and the totals are the same whatever the number of
// workgroups or ranks the ECs are spread over.  2^K = Scalars::fx_scale, sum c * 2^K < 2^61.
// MSW_FX=0 builds the fp64-atomic sweeps (A/B timing only); kFx lives in common.hpp.
// double -> integer by the magic-number trick: for |q| < 2^51 the low 52 bits of (q + 1.5 * 2^52) hold
// rint(q) in two's complement; subtracting the magic's bit pattern leaves it as a 64-bit integer.  The
// magic's low dword is zero: the subtraction is ONE 32-bit operation on the high dword.
constexpr double kFxMagic = 6755399441055744.0;             // 1.5 * 2^52 = 0x4338000000000000
constexpr unsigned long long kFxMagicBits = 0x4338000000000000ull;
__device__ __forceinline__ unsigned long long fx_bits(double scaled_r, double pk) {
  const double v = fma(scaled_r, pk, kFxMagic);
  const uint32_t hi = (uint32_t)__double2hiint(v) - (uint32_t)(kFxMagicBits >> 32);
  return ((unsigned long long)hi << 32) | (uint32_t)__double2loint(v);
}

// c / z and 1 / z for the EC epilogues of the sweeps: z is a softmax denominator that passed the guard test
// (sell.hpp) -- a normal number between 2^-8 of the background sum and twice the number of groups -- so the
// operand scaling and the special-case fix-up of the IEEE division (4 of its 12 instructions, 7 of 12 for a
// reciprocal) are not needed.  Reciprocal estimate and two Newton steps (within an ulp of 1 / z), times c: within
// two ulps of the quotient -- the residual correction that brought it to one (two more operations per EC: round 4
// counted the EC epilogue, 23 of pass B's vector operations per cell at 5 cells per EC) changes nothing a test can
// see: MSW_DIV_CORRECT=1 brings it back.  (The guarded ECs' own evaluation, whose z may be anything above zero,
// divides in full.)
#ifndef MSW_FAST_DIV
#define MSW_FAST_DIV 1
#endif
#ifndef MSW_DIV_CORRECT
#define MSW_DIV_CORRECT 0
#endif
__device__ __forceinline__ double ec_rcp(double z) {
#if MSW_FAST_DIV
  double r = __builtin_amdgcn_rcp(z);
  r = fma(fma(-z, r, 1.0), r, r);
  r = fma(fma(-z, r, 1.0), r, r);
  return r;
#else
  return 1.0 / z;
#endif
}
__device__ __forceinline__ double ec_div(double c, double z) {
#if MSW_FAST_DIV
  const double r = ec_rcp(z);
  const double q = c * r;
#if MSW_DIV_CORRECT
  return fma(fma(-z, q, c), r, q);
#else
  return q;
#endif
#else
  return c / z;
#endif
}

__device__ __forceinline__ uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// a wave-uniform double that was loaded through a vector load: move it to SGPRs
__device__ __forceinline__ double uniform_d(double v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double((int)hi, (int)lo);
}

// One slice held in registers (<= kRegCells cells per EC): records, geometry and -- pass B -- the
// EC's multiplicity.
template <int ENC, int RC = kRegCells>
struct SliceBuf {
  typename Rec<ENC>::T r[RC];
  // index records (sell.hpp): the rows of the slice's cold segment; r then holds the hot rows only
  typename Rec<ENC>::T rc[ENC == kEncIndex ? kColdRows : 1];
  uint32_t sl, o, len, nhot;
  uint32_t lgm, nec;  // slice class (sell.hpp): log2 lanes per EC, ECs in the slice
  uint32_t c8;  // byte image of the EC's multiplicity (sell.hpp)
};

// Issue the loads of one slice (<= kRegCells cells per EC) into registers.  The sweeps work through a slice two
// rows at a time; a slice may have an odd number of rows (sell.hpp odd_slices: until round 3
// every slice was padded in memory -- 5 % of cfg3's record stream, 10 % of cfg5's): its missing last row is the
// lane's null record, made here instead of being read.
template <int ENC, int RC>
__device__ __forceinline__ void load_slice(const uint32_t *rec, size_t base, uint32_t len,
                                           typename Rec<ENC>::T (&r)[RC], typename Rec<ENC>::T nullr) {
#pragma unroll
  for (int k = 0; k < RC; k += 2) {
    if ((uint32_t)k < len) {
      r[k] = Rec<ENC>::load(rec, base + (size_t)k * 64);
      if constexpr (odd_slices(ENC)) {
        r[k + 1] = nullr;
        if ((uint32_t)k + 1 < len) r[k + 1] = Rec<ENC>::load(rec, base + (size_t)(k + 1) * 64);
      } else {
        r[k + 1] = Rec<ENC>::load(rec, base + (size_t)(k + 1) * 64);
      }
    }
  }
}
// index records: rows [0, nhot) into r, the cold rows [nhot, len) -- at most kColdRows, or the packer has set
// nhot = 0 and the whole slice is taken from memory -- into rc
__device__ __forceinline__ void load_slice_split(const uint32_t *rec, size_t base, uint32_t len, uint32_t nhot,
                                                 uint32_t (&r)[kRegCells], uint32_t (&rc)[kColdRows], uint32_t nullr,
                                                 uint32_t nullr_hot) {
  const uint32_t ncold = len - nhot;
  if (ncold <= (uint32_t)kColdRows) {
    load_slice<kEncIndex>(rec, base, nhot, r, nullr_hot);  // (the rows of a hot segment: the hot record form, sell.hpp)
#pragma unroll
    for (int j = 0; j < kColdRows; ++j)
      if ((uint32_t)j < ncold) rc[j] = rec[base + (size_t)(nhot + j) * 64];
  } else {
    load_slice<kEncIndex>(rec, base, len, r, nullr);
  }
}

// a record that contributes nothing: sentinel group g (e = 0), slot entry 0 (value records: the background
// value, which lies inside the range the pass scales its exponentials by)
template <int ENC>
__device__ __forceinline__ typename Rec<ENC>::T null_record(uint32_t g, const RecDec &d, double tnull) {
  if constexpr (ENC == kEncValue) return ValRec{8u * g, tnull};
  else return Rec<ENC>::make(g, 0u, d);
}
// threads per workgroup of the sweeps: value records take three registers per cell (96 for the two slice
// buffers) -- 8 wavefronts, 256 registers per lane, in both sweeps
#ifndef MSW_VAL_THREADS
#define MSW_VAL_THREADS 512
#endif
#ifndef MSW_HYB_THREADS_A
#define MSW_HYB_THREADS_A MSW_PASS_THREADS_A
#endif
// (index records: the cold segment's records and gathered entries push pass A past the 128 registers of 16
// wavefronts -- 12, like pass B)
template <int ENC>
constexpr int pass_threads_A() {
  return ENC == kEncValue ? MSW_VAL_THREADS : (ENC == kEncIndex ? MSW_HYB_THREADS_A : kPassThreads);
}
template <int ENC, int RC = kRegCells>
constexpr int pass_threads_B() { return ENC == kEncValue ? MSW_VAL_THREADS : (RC < kRegCells ? kPassThreadsB8 : kPassThreadsB); }

// s_waitcnt vmcnt(0), leaving the other counters alone (gfx9 layout: vmcnt = imm[3:0] | imm[15:14] << 4)
__device__ __forceinline__ void wait_vm0() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// The record stream of one wavefront: its slices s_first, s_first + nw, ... go through two
// register buffers in ping-pong -- while one slice is processed the records of the next are in
// flight; the explicit vmcnt(0) sits BEFORE the next buffer's loads are issued, so it only waits
// for the buffer about to be consumed.  The slice geometry (slice_off pairs) of 64 slices at a time
// is fetched with one gather per lane and parked in LDS: no dependent global load sits between two
// slices (a register copy handed out with v_readlane would make the compiler's waitcnt analysis
// drain ALL vector loads at every use).  A third buffer (two slices in flight) was built and
// measured: no faster -- the stream already runs at the achievable HBM rate -- and it needs a fixed
// number of loads per slice plus dummy fetches to keep the compiler's path-insensitive vmcnt
// accounting exact.
// MSW_EXPERIMENT_NOSTREAM (tools/ab_build.py; never the shipped build): every slice takes its rows from the first
// 256 KB of the record stream -- cache-resident -- so that a sweep's time WITHOUT its HBM stream can be measured
// (replicates sharing a pass).  Plain offset records only: the result is that of another problem.
#ifdef MSW_EXPERIMENT_NOSTREAM
#define MSW_REC_ROW(o) ((o) & 1023u)
#else
#define MSW_REC_ROW(o) (o)
#endif
template <int ENC, bool REVERSE, int RC = kRegCells>
struct SliceStream {
  const SellDev &S;
  uint32_t s_first, nw, n_mine, lane;
  uint32_t geo;  // LDS byte offset of this wave's 64 (+2 dummy) {slice_off[s], slice_off[s+1]} pairs
  typename Rec<ENC>::T nullr = {};  // the lane's null record (the missing last row of an odd slice: load_slice)
  typename Rec<ENC>::T nullr_hot = {};  // ... in the form of a hot segment's rows (index records)
  uint2 pend = make_uint2(0, 0);
  __device__ __forceinline__ SliceStream(const SellDev &S_, uint32_t first, uint32_t nw_, uint32_t lane_,
                                         uint32_t geo_)
      : S(S_), s_first(first), nw(nw_), lane(lane_), geo(geo_) {
    n_mine = first < S.nslices ? (S.nslices - first + nw_ - 1) / nw_ : 0;
    gather_offs(0);
  }
  // i-th slice this wave visits.  Pass B walks the list backwards: the sweeps alternate, so each
  // starts on the part of the record stream the other has just left in the memory-side cache
  // (the stream is larger than the 256 MB Infinity Cache: walking it the same way every time
  // would evict every line before its reuse).
  __device__ __forceinline__ uint32_t slice_at(uint32_t i) const {
    return s_first + (REVERSE ? (i < n_mine ? n_mine - 1 - i : i) : i) * nw;
  }
  // geometry of the wave's slices [64 * chunk, 64 * chunk + 64): one gather per lane, parked in LDS
  __device__ __forceinline__ void gather_offs(uint32_t chunk) {
    const uint32_t i = chunk * 64 + lane;
    pend = make_uint2(0, 0);  // past the wave's last slice: an empty slice
    if (i < n_mine) {
      const uint32_t sl = slice_at(i);
      // (the slice's rows, and above them its class: lanes per EC, ECs -- a few vector operations per 64 slices
      // here instead of scalar ones per slice in fetch())
      const uint32_t o = S.slice_off[sl];
      const SliceGeo sg = slice_geo(S.cls, sl);
      pend = make_uint2(o, (S.slice_off[sl + 1] - o) | sg.lgm << kGeoLgmShift | sg.nec << kGeoNecShift);
      // index records: the rows of the slice's hot segment ride above its offset (< 2^27: checked at upload)
      if constexpr (ENC == kEncIndex) pend.x |= (uint32_t)S.slice_hot[sl] << kGeoHotShift;
    }
  }
  __device__ __forceinline__ void commit_offs() {
    typedef uint32_t v2u_t __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) v2u_t lds_u2_t;
    v2u_t v = {pend.x, pend.y}, z = {0u, 0u};
    *(lds_u2_t *)(size_t)(geo + lane * 8) = v;
    if (lane < 2) *(lds_u2_t *)(size_t)(geo + (64 + lane) * 8) = z;  // read by the look-ahead fetch
  }
  // j = position inside the current 64-slice chunk; j >= n_chunk fetches an empty slice
  template <class Issue>
  __device__ __forceinline__ void fetch(uint32_t base, uint32_t j, SliceBuf<ENC, RC> &b, Issue &issue) {
    typedef uint32_t v2u_t __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) const v2u_t lds_cu2_t;
    const v2u_t oe = *(lds_cu2_t *)(size_t)(geo + j * 8);
    b.sl = slice_at(base + j);
    const uint32_t oy = uniform(oe.y);
    b.len = oy & ((1u << kGeoLgmShift) - 1u);
    b.lgm = (oy >> kGeoLgmShift) & 7u;
    b.nec = oy >> kGeoNecShift;
    if constexpr (ENC == kEncIndex) {
      const uint32_t ox = uniform(oe.x);
      b.o = ox & ((1u << kGeoHotShift) - 1u);
      b.nhot = ox >> kGeoHotShift;
      if (b.len <= (uint32_t)RC) load_slice_split(S.rec, (size_t)b.o * 64 + lane, b.len, b.nhot, b.r, b.rc, nullr, nullr_hot);
    } else {
      b.o = uniform(oe.x);
      if (b.len <= (uint32_t)RC) load_slice<ENC>(S.rec, (size_t)MSW_REC_ROW(b.o) * 64 + lane, b.len, b.r, nullr);
    }
    issue(b);
  }
  // The wavefront's FIRST slice requested at the head of the kernel, before the workgroup fills its LDS image and
  // meets at the barrier behind it (round 5): the geometry a wavefront parks in LDS is its own (no barrier between
  // its write and its read), so the first records' trip to HBM runs under the fill instead of after it -- every
  // sweep used to open with that round trip exposed.  `first` is handed to run().
  template <class Issue>
  __device__ __forceinline__ void prime(SliceBuf<ENC, RC> &first, Issue &issue) {
    commit_offs();
    fetch(0, 0, first, issue);
  }
  // issue(buf): further loads of a slice; process(buf): its arithmetic; chunk_end(): every 64 slices
  // (the geometry is renewed there, with the stream drained); primed: the buffer prime() filled, or null
  template <class Issue, class Process, class ChunkEnd>
  __device__ __forceinline__ void run(Issue issue, Process process, ChunkEnd chunk_end,
                                      const SliceBuf<ENC, RC> *primed = nullptr) {
    SliceBuf<ENC, RC> A = {}, B = {};
    if (primed) A = *primed;  // (copied once, here: the caller's buffer is dead from now on)
    for (uint32_t base = 0; base == 0 || base < n_mine; base += 64) {
      if (base) gather_offs(base >> 6);
      if (base || !primed) commit_offs();
      const uint32_t n_chunk = n_mine > base ? (n_mine - base < 64u ? n_mine - base : 64u) : 0u;
      uint32_t j = 0;
      if (base || !primed) fetch(base, 0, A, issue);
      for (;;) {
        if (j >= n_chunk) break;
        wait_vm0();  // A has landed
        fetch(base, j + 1, B, issue);
        process(A);
        if (++j >= n_chunk) break;
        wait_vm0();
        fetch(base, j + 1, A, issue);
        process(B);
        ++j;
      }
      wait_vm0();
      chunk_end();
    }
  }
};

// 16-byte / 8-byte table entries at a byte offset, from the LDS image or from global memory.
// The LDS image starts at LDS address 0 (the sweeps have no static __shared__; checked on the
// host before the first launch), so a record field IS the ds address: building the pointer from
// the integer keeps the compiler from adding the dynamic-LDS base symbol to every gather.
typedef double v2d_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const v2d_t lds_cd2_t;
typedef __attribute__((address_space(3))) const double lds_cd_t;
typedef __attribute__((address_space(3))) double lds_d_t;
template <bool INLDS>
__device__ __forceinline__ double2 tab16(const unsigned char *glob, uint32_t off) {
  if constexpr (INLDS) {
    const v2d_t v = *(lds_cd2_t *)(size_t)off;
    return make_double2(v.x, v.y);
  }
  else return *reinterpret_cast<const double2 *>(glob + off);
}
template <bool INLDS>
__device__ __forceinline__ double tab8(const unsigned char *glob, uint32_t off) {
  if constexpr (INLDS) return *(lds_cd_t *)(size_t)off;
  else return *reinterpret_cast<const double *>(glob + off);
}

// ---------------------------------------------------------------------------------------
// Pass A: newnorm = sum_j Var_{q_j}(step_.j), q_j = softmax_g(a*L + u).  The variance is
// invariant to a per-EC shift of the step values, so they are taken relative to the background
// cell of the same EC:  s_gj = D_i + wc_g on a listed cell (D_i = (1-a)*(T_i - logzi), slot
// table {x_i, D_i}), s_gj = wc_g elsewhere (group table {e_g, wc_g}).  10 fp64 operations per cell.
// ---------------------------------------------------------------------------------------
struct AccA {
  double zs, t1, t2;
};
__device__ __forceinline__ void cellA(AccA &c, const double p0, const double e, const double w,
                                      const double x, const double D) {
  const double xm = x - p0;
  const double xD = x * D;
  const double wx = w * xm;
  c.zs = fma(e, xm, c.zs);
  c.t1 = fma(e, xD + wx, c.t1);
  c.t2 = fma(e, fma(xD, D, w * fma(2.0, xD, wx)), c.t2);
}

// ENC = kEncIndex (index records): the hybrid slot area -- TLDS is then false by convention, the LDS image
// holds the hot head of the area (S.n_tab_lds entries) and the whole area lives in memory (sell.hpp).
template <int ENC, bool GLDS, bool TLDS, bool ML>
__global__ __launch_bounds__(pass_threads_A<ENC>()) void k_passA(const Scalars *sc, SellDev S,
                                                       const double2 *ew_g, const double2 *tabA_g,
                                                       double *partA, const double *partR, int npartR, GuardDev GD) {
  extern __shared__ __align__(16) unsigned char smem[];
  using R = Rec<ENC>;
  using RT = typename R::T;
  constexpr bool HYB = ENC == kEncIndex;
  constexpr bool VAL = ENC == kEncValue;  // value records: no table, exp per cell
  constexpr int NT = pass_threads_A<ENC>();
  static_assert(!((HYB || VAL) && TLDS), "index records go with the hybrid slot area, value records have no table");
  constexpr bool TL = TLDS || HYB;  // the LDS image starts with (a part of) the slot table
  const RecDec D = rec_dec(S);
  const uint32_t n_tab = S.n_tab_lds;
  // (the test sits behind the LDS fill so that the fill's loads do not wait for this one: one memory round trip
  // less at the head of every sweep)
  const int skip = sc->done;
  const int tid = threadIdx.x, lane = tid & 63;
  const uint32_t G = S.n_groups, Gp = G + kSentinels;
  const uint32_t bhiA = S.bhiA;
  const uint32_t scratch_off = (uint32_t)pass_scratch_off(GLDS ? 1 : 0, n_tab, G, true, HYB);
  double *sh = reinterpret_cast<double *>(smem + scratch_off);
  // slice geometry of this wave: the gather is in flight while the LDS image is filled
  SliceStream<ENC, false> stream(S, uniform(blockIdx.x * (NT / 64) + (tid >> 6)),
                              gridDim.x * (NT / 64), (uint32_t)lane,
                              scratch_off + 256u + uniform(tid >> 6) * kGeoStride);
  // The two background moments S1 = sum_g e_g s0_g, S2 = sum_g e_g s0_g^2 of the step values (s0_g = w_g - kappa):
  // k_redfin left one partial pair per workgroup (partR[5 b + 3], [5 b + 4]); the first wavefront of every workgroup adds
  // them up, in the same fixed order everywhere -- loads issued here, under the LDS fill, the DPP sums in front of the
  // barrier that follows it, the two totals handed to the other wavefronts through LDS.  (All sixteen wavefronts
  // summing for themselves cost pass A 2.5 us: 33 MB of L2 reads at the head of the sweep.  Until round 4 k_fin
  // summed them into the scalar state and pass A had to wait for its verdict: k_finstep.)
  // (eight pairs per lane in flight at a time -- 512 workgroups of k_redfin, 8192 groups -- not one round trip each)
  // (lanes past the end of a long EC, and the missing last row of an odd slice, take a record of the lane's own
  // sentinel group)
  [[maybe_unused]] const double vlogzi0 = uniform_d(sc->logzi);
  const RT null_rec = null_record<ENC>(S.n_groups + (uint32_t)lane, D, vlogzi0);
  stream.nullr = null_rec;
  if constexpr (HYB) stream.nullr_hot = R::make_h(S.n_groups + (uint32_t)lane, 0u, D);
  else stream.nullr_hot = null_rec;
  auto issue = [&](SliceBuf<ENC> &) {};
  // (offset records only: with the 8-byte, index and value records the primed buffer costs the kernels 100-300 bytes
  // of scratch per lane -- cfg2's value-record sweeps went from 30 to 44 us -- and they open as before)
  constexpr bool PRIME = kPrimeFirstSlice && ENC == kEncNarrow;
  SliceBuf<ENC> first = {};
  if constexpr (PRIME) stream.prime(first, issue);  // the first slice's records are in flight under the LDS fill
  double m1 = 0.0, m2 = 0.0;
  for (int b0 = 0; tid < 64 && b0 < npartR; b0 += 512) {
    double t1[8], t2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int b = b0 + 64 * k + lane;
      t1[k] = b < npartR ? partR[kRedfinParts * b + 3] : 0.0;
      t2[k] = b < npartR ? partR[kRedfinParts * b + 4] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) m1 += t1[k], m2 += t2[k];
  }
  if (TL) {
    double2 *t = reinterpret_cast<double2 *>(smem);
    for (uint32_t i = tid; i < n_tab; i += NT) t[i] = tabA_g[i];
  }
  if (GLDS) {
    double2 *t = reinterpret_cast<double2 *>(smem + bhiA);
    for (uint32_t g = tid; g < Gp; g += NT) t[g] = ew_g[g];
  }
  // global fallbacks see the same byte offsets as the LDS image
  const unsigned char *ew_b = reinterpret_cast<const unsigned char *>(ew_g) - bhiA;
  const unsigned char *xt_b = reinterpret_cast<const unsigned char *>(tabA_g);
  auto EW_ = [&](RT r) -> double2 { return tab16<GLDS>(ew_b, R::ew_off(r, D)); };
  auto EWh_ = [&](RT r) -> double2 { return tab16<GLDS>(ew_b, rec_ew_off<ENC, true>(r, D)); };  // a hot segment's row
  // the cell's {x, D} = {exp(a (T - tref)), (1 - a)(T - log zi)}: from the slot table, or -- value records -- formed here
  [[maybe_unused]] const double va = uniform_d(sc->a), vtref = uniform_d(sc->tref), vlogzi = uniform_d(sc->logzi);
  [[maybe_unused]] const double voma = 1.0 - va;
  auto XTg_ = [&](RT r) -> double2 {   // any entry (hybrid: from memory)
    if constexpr (VAL) return make_double2(exp_le0(va * (r.t - vtref)), voma * (r.t - vlogzi));
    else return tab16<TLDS>(xt_b, R::t_off(r, D));
  };
  auto XT_ = [&](RT r) -> double2 {    // hybrid: hot entries only, from a hot segment's row (16 * entry ready-made)
    if constexpr (VAL) return XTg_(r);
    else return tab16<TL>(xt_b, rec_t_off<ENC, true>(r, D));
  };
  const double p0 = uniform_d(sc->p0), U = uniform_d(sc->U);
  const double zbase = p0 * U;
  const double gthr = fmax(zbase * kGuardRatio, 2.2250738585072014e-308);  // (Z = 0 is set aside too: reported, not divided by)  // ECs whose Z falls below it are set aside (sell.hpp, guarded ECs)
  const uint32_t gcnt_off = scratch_off + 128u;
  typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
  auto defer = [&](uint32_t p) {
    const uint32_t i = __hip_atomic_fetch_add((lds_u32_t *)(size_t)gcnt_off, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (i < GD.cap) GD.list[(size_t)blockIdx.x * GD.cap + i] = p;
    else *GD.err = 2;  // cannot happen while guard_cap mirrors the stride distribution (alloc_solve_state)
  };
  double nn = 0.0;
  if (skip) return;
  if (tid == 0) *(lds_u32_t *)(size_t)gcnt_off = 0u;
  if (tid < 64) {  // (wave-uniform)
    m1 = wave_sum(m1);
    m2 = wave_sum(m2);
    if (tid == 0) sh[18] = m1, sh[19] = m2;  // (reduction scratch: doubles 0..15 the block sums, byte 128 the guard counter)
  }
  if (tid == 0 && blockIdx.x == 0) MSW_STAMP(sc->iter, 0, 0);
  __syncthreads();
  if (tid == 0 && blockIdx.x == 0) MSW_STAMP(sc->iter, 0, 1);
  const double b1 = p0 * uniform_d(sh[18]), b2 = p0 * uniform_d(sh[19]);

  auto process = [&](SliceBuf<ENC> &sb) {
    const uint32_t len = sb.len;
    AccA c = {0.0, 0.0, 0.0};
    // straight-line code per cell count (wave-uniform): all LDS gathers of a batch can be
    // in flight together instead of one scalar-branched pair at a time
    // (ANY: the cells may refer to any entry of a hybrid slot area -- gathered from memory)
    auto fixed = [&](RT(&b)[kRegCells], auto LEN, auto ANY) {
      constexpr int L = decltype(LEN)::value;
      constexpr int B = MSW_PASSA_BATCH;  // cells gathered together (2 x ds_read_b128 each)
#pragma unroll
      for (int k0 = 0; k0 < L; k0 += B) {
        double2 ewv[B], xtv[B];
#pragma unroll
        for (int k = 0; k < B; ++k) {
          if (k0 + k < L) {
            if constexpr (decltype(ANY)::value) ewv[k] = EW_(b[k0 + k]), xtv[k] = XTg_(b[k0 + k]);
            else ewv[k] = EWh_(b[k0 + k]), xtv[k] = XT_(b[k0 + k]);
          }
        }
#pragma unroll
        for (int k = 0; k < B; ++k)
          if (k0 + k < L) cellA(c, p0, ewv[k].x, ewv[k].y, xtv[k].x, xtv[k].y);
      }
    };
    auto cells = [&](RT(&b)[kRegCells], uint32_t n) {
      n += n & 1;  // an odd slice's missing last row is a null record in the registers (load_slice)
      switch (n) {
        case 0: break;
        case 2: fixed(b, std::integral_constant<int, 2>{}, std::false_type{}); break;
        case 4: fixed(b, std::integral_constant<int, 4>{}, std::false_type{}); break;
        case 6: fixed(b, std::integral_constant<int, 6>{}, std::false_type{}); break;
#if MSW_REG_CELLS >= 12
        case 8: fixed(b, std::integral_constant<int, 8>{}, std::false_type{}); break;
        case 10: fixed(b, std::integral_constant<int, 10>{}, std::false_type{}); break;
#endif
#if MSW_REG_CELLS == 16
        case 12: fixed(b, std::integral_constant<int, 12>{}, std::false_type{}); break;
        case 14: fixed(b, std::integral_constant<int, 14>{}, std::false_type{}); break;
#endif
        default: fixed(b, std::integral_constant<int, kRegCells>{}, std::false_type{}); break;
      }
    };
    // a slice from memory, pair by pair (rare shapes only: no second copy of the per-count code)
    auto pairs_any = [&](RT(&b)[kRegCells], uint32_t n) {
#pragma unroll
      for (int k = 0; k < kRegCells; k += 2) {
        if ((uint32_t)k < n) {  // (n odd: the pair's second record is the lane's null record, load_slice)
          const double2 a0 = EW_(b[k]), a1 = EW_(b[k + 1]);
          const double2 x0 = XTg_(b[k]), x1 = XTg_(b[k + 1]);
          cellA(c, p0, a0.x, a0.y, x0.x, x0.y);
          cellA(c, p0, a1.x, a1.y, x1.x, x1.y);
        }
      }
    };
    if (len <= (uint32_t)kRegCells) {
      if constexpr (HYB) {
        const uint32_t nhot = sb.nhot, ncold = len - nhot;
        if (ncold <= (uint32_t)kColdRows) {
          // the cold rows' table entries come from memory: issued first, consumed after the hot rows
          double2 cew[kColdRows], cxt[kColdRows];
#pragma unroll
          for (int j = 0; j < kColdRows; ++j) {
            if ((uint32_t)j < ncold) {
              cxt[j] = XTg_(sb.rc[j]);
              cew[j] = EW_(sb.rc[j]);
            }
          }
          cells(sb.r, nhot);
#pragma unroll
          for (int j = 0; j < kColdRows; ++j)
            if ((uint32_t)j < ncold) cellA(c, p0, cew[j].x, cew[j].y, cxt[j].x, cxt[j].y);
        } else {
          pairs_any(sb.r, len);
        }
      } else {
        cells(sb.r, len);
      }
    } else {
      // more cells per EC than the registers hold (the stream has not fetched this slice): chunks of
      // kRegCells records through the same registers and the same straight-line code; the other
      // wavefronts of the workgroup cover each chunk's load latency
      // (registers of their own: a load into the stream's buffers inside process() would make the
      // compiler's path-insensitive waitcnt bookkeeping drain the prefetch on the short path too)
      const size_t base = (size_t)sb.o * 64 + lane;
      RT t[kRegCells];
      uint32_t k0 = 0;
      for (; k0 + (uint32_t)kRegCells <= len; k0 += kRegCells) {
        load_slice<ENC>(S.rec, base + (size_t)k0 * 64, kRegCells, t, null_rec);
        fixed(t, std::integral_constant<int, kRegCells>{}, std::true_type{});
      }
      if (k0 < len) {  // the last chunk: pair by pair (a second copy of the per-count code costs the
                       // short path 5 % through its sheer size)
        const uint32_t n = len - k0;
        load_slice<ENC>(S.rec, base + (size_t)k0 * 64, n, t, null_rec);
        pairs_any(t, n);
      }
    }
    // an EC over several lanes (sell.hpp slice classes): its partial sums meet; its first lane speaks for it
    // (ML = false: every slice one lane per EC)
    const uint32_t lgm = ML ? sb.lgm : 0u;
    bool mine = ((uint32_t)lane >> lgm) < sb.nec;
    if (ML && lgm != 0u) {
      c.zs = group_sum(c.zs, lgm);
      c.t1 = group_sum(c.t1, lgm);
      c.t2 = group_sum(c.t2, lgm);
      mine = mine && ((uint32_t)lane & ((1u << lgm) - 1u)) == 0u;
    }
    if (mine) {
      const double Zt = zbase + c.zs;
      if (Zt >= gthr) {
        const double iZ = ec_rcp(Zt);
        const double S1 = (b1 + c.t1) * iZ, S2 = (b2 + c.t2) * iZ;
        nn += S2 - S1 * S1;
      } else {
        defer(S.n_long + (ML ? slice_geo(S.cls, sb.sl).ec0 + ((uint32_t)lane >> lgm) : sb.sl * 64 + lane));
      }
    }
  };
  if constexpr (PRIME) stream.run(issue, process, [] {}, &first);
  else stream.run(issue, process, [] {});
  // long ECs (plain CSR): one wavefront per EC, a cell per lane and step -- the groups of one EC are
  // distinct, so the gathers of a step never meet on an address, and the three sums are wave
  // reductions (no barrier)
  // (lanes past the end take a record of their own sentinel group)
  // The wavefront's ECs are one sequence of steps of LS * 64 cells; the records of the NEXT step (of
  // this EC or of the next one) are always in flight while a step is processed -- a step costs a
  // full memory round trip otherwise, and a wavefront walks some sixty of them one after the other.
  {
    constexpr int LS = ENC == kEncWide || VAL ? 4 : kLongStep;  // 8- and 12-byte records: more registers
    auto load_long = [&](uint32_t kb, uint32_t k1, RT(&dst)[LS]) {
#pragma unroll
      for (int u = 0; u < LS; ++u) {
        dst[u] = null_rec;  // wave-uniform test first: whole 64-cell groups past the end cost nothing
        if (kb + 64u * u < k1 && kb + 64u * u + lane < k1) dst[u] = R::load(S.rec_long, kb + 64u * u + lane);
      }
    };
    uint32_t r = stream.s_first, kb = 0, k1 = 0;
    bool have = r < S.n_long;
    if (have) kb = S.long_ptr[r], k1 = S.long_ptr[r + 1];
    RT cur[LS], nxt[LS];
    load_long(kb, k1, cur);
    AccA c = {0.0, 0.0, 0.0};
    while (have) {
      // where the next step is
      uint32_t nr = r, nkb = kb + LS * 64, nk1 = k1;
      bool nhave = true;
      if (nkb >= k1) {
        nr = r + stream.nw;
        nhave = nr < S.n_long;
        nkb = nk1 = 0;
        if (nhave) nkb = S.long_ptr[nr], nk1 = S.long_ptr[nr + 1];
      }
      load_long(nkb, nk1, nxt);
#pragma unroll
      for (int q = 0; q < LS; q += 4) {
        if (kb + 64u * q < k1) {  // wave-uniform: quads past the end are skipped
          double2 a0[4], x0[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) a0[u] = EW_(cur[q + u]), x0[u] = XTg_(cur[q + u]);
#pragma unroll
          for (int u = 0; u < 4; ++u) cellA(c, p0, a0[u].x, a0[u].y, x0[u].x, x0[u].y);
        }
      }
      if (nr != r) {  // that was the EC's last step
        const double zs = wave_sum(c.zs), t1 = wave_sum(c.t1), t2 = wave_sum(c.t2);
        if (lane == 0) {
          const double Zt = zbase + zs;
          if (Zt >= gthr) {
            const double iZ = ec_rcp(Zt);
            const double S1 = (b1 + t1) * iZ, S2 = (b2 + t2) * iZ;
            nn += S2 - S1 * S1;
          } else {
            defer(r);
          }
        }
        c = {0.0, 0.0, 0.0};
      }
#pragma unroll
      for (int u = 0; u < LS; ++u) cur[u] = nxt[u];
      r = nr, kb = nkb, k1 = nk1, have = nhave;
    }
  }
  // guarded ECs (sell.hpp): a wavefront each, every group visited -- no background term to cancel against
  __syncthreads();
  const uint32_t n_guard = min(*(lds_u32_t *)(size_t)gcnt_off, GD.cap);
  if (n_guard) {
    const uint32_t wv = uniform(tid >> 6), nwv = NT / 64;
    uint32_t *bits = GD.bits + (size_t)(blockIdx.x * 16 + wv) * GD.words;
    const double a = uniform_d(sc->a), oma = 1.0 - a, logzi = uniform_d(sc->logzi), tref = uniform_d(sc->tref);
    for (uint32_t i = wv; i < n_guard; i += nwv) {
      const uint32_t p = GD.list[(size_t)blockIdx.x * GD.cap + i];
      double z = 0.0, t1 = 0.0, t2 = 0.0;
      wave_cells<ENC>(S, p, (uint32_t)lane, [&](uint32_t g, double T) {
        const double x = exp(a * (T - tref)), sv = oma * (T - logzi) + ew_g[g].y;
        const double q = ew_g[g].x * x;
        atomicOr(&bits[g >> 5], 1u << (g & 31));
        z += q;
        t1 = fma(q, sv, t1);
        t2 = fma(q * sv, sv, t2);
      });
      __builtin_amdgcn_s_waitcnt(0);  // the bitmap updates have reached memory before it is read back
      for (uint32_t w0 = lane; w0 < GD.words; w0 += 64) {
        const uint32_t listed = atomicOr(&bits[w0], 0u);
        for (uint32_t b = 0; b < 32; ++b) {
          const uint32_t g = w0 * 32 + b;
          if (g < G && !((listed >> b) & 1u)) {
            const double2 ev = ew_g[g];
            const double q = ev.x * p0;
            z += q;
            t1 = fma(q, ev.y, t1);
            t2 = fma(q * ev.y, ev.y, t2);
          }
        }
        atomicAnd(&bits[w0], 0u);
      }
      z = wave_sum(z), t1 = wave_sum(t1), t2 = wave_sum(t2);
      if (lane == 0) {
        if (z > 0.0) {
          const double S1 = t1 / z, S2 = t2 / z;
          nn += S2 - S1 * S1;
        } else {
          *GD.err = 1;  // no probability under any group: the solve reports it
        }
      }
      __builtin_amdgcn_s_waitcnt(0);
    }
  }
  nn = block_sum_fixed1<NT / 64>(nn, sh);
  if (tid == 0) MSW_STAMP_MAX(sc->iter, 0, 7);
  if (tid == 0) partA[blockIdx.x] = nn;
}

// ---------------------------------------------------------------------------------------
// Pass B: per EC Z_j (softmax denominator), r_j = c_j / Z_j, the ELBO data terms and the column
// sums A_g = sum_j r_j (x_gj - p0) accumulated in an LDS-private table (rcgpar logsumexp +
// update_N_k + ELBO_rcg_mat in one sweep).  Slot table {x_i - p0, x_i*T_i - p0*logzi}: two FMAs
// per cell in the row sums; the x - p0 of an EC stay in registers between the row sum and the
// scatter.
// ---------------------------------------------------------------------------------------
// GMODE: 3 = column sums in LDS (where e_g would be), e_g gathered from memory: groups up to ~17 k
// (global fp64 atomics, mode 0, cost 20 x more than everything else in the sweep);
// 4 = as 3 for any number of groups: one run of the sweep per range of kRangeGroups groups (RangeB);
// 0 = e_g / column sums in global memory; 1 = in LDS, column sums right behind e_g;
// 2 = in LDS, column sums at the fixed distance kAccFixed (an instruction immediate: one VALU
// operation less per scattered cell; needs 8 * Gp <= kAccFixed).
// RC = records per slice lane kept in registers: 16 (kRegCells: every slice of the layout), or 8 -- the SHORT-SLICE
// instantiation (round 5) for likelihoods whose slices have (nearly all) at most 8 rows: record buffers 32 -> 16
// registers, kept values 32 -> 16, 117 instead of 144 registers per lane, so SIXTEEN wavefronts per workgroup fit
// (four per SIMD instead of three) without the spills that sank the 14 / 16-wavefront builds of the 16-row kernel
// (56 bytes of scratch per lane, reloaded in every slice).  Pass B is bound by LDS time and VALU issue in balance,
// overlapped by the wavefronts of a SIMD: a fourth one is worth 16 % of the sweep at cfg5 (243 -> 204 us in one job).
// Longer slices take the streaming branch in chunks of 8 rows (host: launch_passB_t picks RC by the rows they hold).
template <int ENC, int GMODE, bool TLDS, bool ML, int RC = kRegCells>
__global__ __launch_bounds__((pass_threads_B<ENC, RC>())) void k_passB(const Scalars *sc, SellDev S, const double *e_g,
                                                       const double2 *tabB_g, double *partAcc,
                                                       double *partS, double *accGlobal, RangeB rg, GuardDev GD) {
  extern __shared__ __align__(16) unsigned char smem[];
  using R = Rec<ENC>;
  using RT = typename R::T;
  constexpr bool HYB = ENC == kEncIndex;  // index records + hybrid slot area (see k_passA)
  constexpr bool VAL = ENC == kEncValue;  // value records: no table, exp per cell
  constexpr int NT = pass_threads_B<ENC, RC>();
  static_assert(!((HYB || VAL) && TLDS), "index records go with the hybrid slot area, value records have no table");
  constexpr bool TL = TLDS || HYB;
  const RecDec D = rec_dec(S);
  const uint32_t n_tab = S.n_tab_lds;
  const int skip = sc->done;  // tested behind the LDS fill (see pass A)
  const int tid = threadIdx.x, lane = tid & 63;
  const uint32_t G = S.n_groups, Gp = G + kSentinels;
  const uint32_t bhi = S.bhi;
  constexpr bool GLDS = GMODE == 1 || GMODE == 2;  // e_g in LDS
  constexpr bool ALDS = GMODE > 0;                  // column sums in LDS
  const uint32_t acc_off = pass_acc_off(GMODE, G);
  const uint32_t scratch_off = (uint32_t)pass_scratch_off(GMODE, n_tab, G, false, HYB);
  double *sh = reinterpret_cast<double *>(smem + scratch_off);
  SliceStream<ENC, MSW_REVERSE_B, RC> stream(S, uniform(blockIdx.x * (NT / 64) + (tid >> 6)),
                              gridDim.x * (NT / 64), (uint32_t)lane,
                              scratch_off + 256u + uniform(tid >> 6) * kGeoStride);
  // the lane's null record and the multiplicity bytes of a slice (issue): needed by the first fetch, which is
  // requested here, before the LDS fill (SliceStream::prime)
  [[maybe_unused]] const double vlogzi0 = uniform_d(sc->logzi);
  const uint32_t n_lanes = S.nslices * 64u;
  const RT null_rec = null_record<ENC>(G + (uint32_t)lane, D, vlogzi0);
  stream.nullr = null_rec;
  if constexpr (HYB) stream.nullr_hot = R::make_h(G + (uint32_t)lane, 0u, D);
  else stream.nullr_hot = null_rec;
  auto issue = [&](SliceBuf<ENC, RC> &sb) {
    // (a byte per slice lane: the lanes of an EC that takes several hold the same one; lanes without an EC 0)
    const uint32_t q = sb.sl * 64 + lane;
    const uint32_t cj = S.c8s[q < n_lanes ? q : 0u];
    sb.c8 = q < n_lanes ? cj : 0u;
  };
  // (see pass A; modes 0 and 3 -- e_g gathered from memory -- sit at the register limit: 16 bytes of scratch with it)
  constexpr bool PRIME = kPrimeFirstSlice && ENC == kEncNarrow && GMODE != 0 && GMODE != 3;
  SliceBuf<ENC, RC> first = {};
  if constexpr (PRIME) stream.prime(first, issue);
  if (TL) {
    double2 *t = reinterpret_cast<double2 *>(smem);
    for (uint32_t i = tid; i < n_tab; i += NT) t[i] = tabB_g[i];
  }
  if (ALDS) {
    double *el = reinterpret_cast<double *>(smem + bhi), *al = reinterpret_cast<double *>(smem + bhi + acc_off);
    const uint32_t n_acc = GMODE == 4 ? rg.n : Gp;
    for (uint32_t g = tid; g < n_acc; g += NT) {
      if (GLDS) el[g] = e_g[g];
      al[g] = 0.0;
    }
  }
  const unsigned char *e_b = reinterpret_cast<const unsigned char *>(e_g) - bhi;
  unsigned char *acc_b = reinterpret_cast<unsigned char *>(accGlobal) - bhi;
  const unsigned char *xt_b = reinterpret_cast<const unsigned char *>(tabB_g);
  auto E_ = [&](RT r) -> double { return tab8<GLDS>(e_b, R::e_off(r, D)); };
  auto Eh_ = [&](RT r) -> double { return tab8<GLDS>(e_b, rec_e_off<ENC, true>(r, D)); };  // a hot segment's row
  // the cell's {x - p0, x T - p0 log zi}, x = exp(a (T - tref)): from the slot table, or -- value records -- formed here
  [[maybe_unused]] const double va = uniform_d(sc->a), vtref = uniform_d(sc->tref), vlogzi = uniform_d(sc->logzi);
  [[maybe_unused]] const double vp0 = uniform_d(sc->p0), vp0l = vp0 * vlogzi;
  auto XTg_ = [&](RT r) -> double2 {   // any entry (hybrid: from memory)
    if constexpr (VAL) {
      const double x = exp_le0(va * (r.t - vtref));
      return make_double2(x - vp0, fma(x, r.t, -vp0l));
    } else {
      return tab16<TLDS>(xt_b, R::t_off(r, D));
    }
  };
  auto XT_ = [&](RT r) -> double2 {    // hybrid: hot entries only, from a hot segment's row (16 * entry ready-made)
    if constexpr (VAL) return XTg_(r);
    else return tab16<TL>(xt_b, rec_t_off<ENC, true>(r, D));
  };
  auto XMg_ = [&](RT r) -> double {
    if constexpr (VAL) return exp_le0(va * (r.t - vtref)) - vp0;
    else return tab8<TLDS>(xt_b, R::t_off(r, D));
  };
  auto XMgh_ = [&](RT r) -> double {   // the same for a hot segment's row
    if constexpr (VAL) return exp_le0(va * (r.t - vtref)) - vp0;
    else return tab8<TLDS>(xt_b, rec_t_off<ENC, true>(r, D));
  };
  typedef __attribute__((address_space(3))) unsigned long long lds_u64_t;
  // one column-sum update: v = the value to add (fp64 build), or its fixed-point image (kFx)
  auto addACCx = [&](RT r, auto v, auto HOT) {
    using V = decltype(v);
    using LT = typename std::conditional<std::is_same<V, double>::value, lds_d_t, lds_u64_t>::type;
    const uint32_t off = rec_e_off<ENC, decltype(HOT)::value>(r, D);
    if constexpr (GMODE == 2)
      __hip_atomic_fetch_add((LT *)(size_t)(off + kAccFixed), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if constexpr (GMODE == 1)
      __hip_atomic_fetch_add((LT *)(size_t)(off + acc_off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if constexpr (GMODE == 3)
      __hip_atomic_fetch_add((LT *)(size_t)off, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if constexpr (GMODE == 4) {
      const uint32_t d = off - bhi - 8u * rg.g0;  // unsigned: groups below the range wrap around
      if (d < 8u * rg.n)
        __hip_atomic_fetch_add((LT *)(size_t)(bhi + d), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    else
      atomicAdd(reinterpret_cast<V *>(acc_b + off), v);
  };
  // kFx: a cell adds rint(rs * pk), rs = 2^K * r_j, pk = f_g * (x - p0).  |rs * pk| < 2^51 is what the
  // magic-number conversion needs; |pk| <= max(e_g |x - p0|, 2^(1-s) max(x, p0)) <= max(Z + zbase, 2^(1-s) xb)
  // (Scalars::xb bounds every table value of the pass), so ONE test per EC covers all its cells.  ECs
  // that fail it -- a large multiplicity over a small Z -- split each addend into two parts of 32 and 51
  // bits (two atomics; sums are modulo 2^64, so the parts need not be added together).
  auto addACC = [&](RT r, auto v) { addACCx(r, v, std::false_type{}); };
  auto addFXx = [&](RT r, double rs, double pk, auto HOT) { addACCx(r, fx_bits(rs, pk), HOT); };
  auto addFXwidex = [&](RT r, double rs, double pk, auto HOT) {
    const double q = rs * pk;
    const double vh = fma(q, 0x1p-32, kFxMagic);
    const double qh = vh - kFxMagic;                       // rint(q / 2^32), exact
    const double ql = fma(-qh, 0x1p32, q);                 // |ql| <= 2^31, exact
    addACCx(r, (unsigned long long)(uint32_t)__double2loint(vh) << 32, HOT);
    addACCx(r, fx_bits(1.0, ql), HOT);
  };
  auto addFX = [&](RT r, double rs, double pk) { addFXx(r, rs, pk, std::false_type{}); };
  auto addFXwide = [&](RT r, double rs, double pk) { addFXwidex(r, rs, pk, std::false_type{}); };
  const double p0 = uniform_d(sc->p0), U = uniform_d(sc->U);
  const double zbase = p0 * U, hbase = p0 * uniform_d(sc->logzi) * U;
  const int fxe = (int)uniform((uint32_t)fx_expbits(sc->fx_shift));
  const double fxs = uniform_d(sc->fx_scale), fxb = ldexp(uniform_d(sc->xb), 1 - (int)uniform((uint32_t)sc->fx_shift));
  const double fxt1 = 0x1p51 / fxs, fxt2 = 0x1p51 / (fxs * fxb);  // the narrow-add test of an EC (below)
  const double gthr = fmax(zbase * kGuardRatio, 2.2250738585072014e-308);  // (Z = 0 is set aside too: reported, not divided by)  // ECs whose Z falls below it are set aside (sell.hpp, guarded ECs)
  const uint32_t gcnt_off = scratch_off + 128u;
  typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
  auto defer = [&](uint32_t p) {
    const uint32_t i = __hip_atomic_fetch_add((lds_u32_t *)(size_t)gcnt_off, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (i < GD.cap) GD.list[(size_t)blockIdx.x * GD.cap + i] = p;
    else *GD.err = 2;  // cannot happen while guard_cap mirrors the stride distribution (alloc_solve_state)
  };
  double s_clogZ = 0.0, s_rH = 0.0, s_W = 0.0;
  double lp_mant = 1.0;  // deferred logarithms: product of mantissas in [2^-960, 1] ...
  int lp_exp = 0;        // ... and sum of exponents, per lane
  auto flush_logs = [&] {
    s_clogZ += log(lp_mant) + (double)lp_exp * 0.693147180559945309417232121458;
    lp_mant = 1.0;
    lp_exp = 0;
  };
  if (skip) return;
  if (tid == 0) *(lds_u32_t *)(size_t)gcnt_off = 0u;
  if (tid == 0 && blockIdx.x == 0) MSW_STAMP(sc->iter, 2, 0);
  __syncthreads();
  if (tid == 0 && blockIdx.x == 0) MSW_STAMP(sc->iter, 2, 1);

  auto process = [&](SliceBuf<ENC, RC> &sb) {
    const uint32_t o = sb.o, len = sb.len;
    double c = (double)sb.c8;
    // not a small integer: rare.  A wave-uniform branch with the wait for its load INSIDE: loads return in
    // order, so a load issued here sits behind the next slice's prefetch -- left pending, the compiler's
    // wait for it at the first use of c (the EC epilogue of EVERY slice, escape or not) is vmcnt(0) and
    // drains the prefetch half-way through the slice
    if (__builtin_amdgcn_ballot_w64(sb.c8 == kC8Escape)) {
      if (sb.c8 == kC8Escape)
        c = S.cvec[S.n_long + (ML ? slice_geo(S.cls, sb.sl).ec0 + ((uint32_t)lane >> sb.lgm) : sb.sl * 64 + lane)];
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    }
    double zs = 0.0, hs = 0.0;
    // row sums: straight-line code per cell count (wave-uniform).  x - p0 of the first KEEPN
    // cells stays in registers for the scatter, any others are gathered a second time (with 16
    // wavefronts per workgroup all 16 would push the kernel into scratch, and a scratch reload
    // drains the record prefetch: hence 12 wavefronts, common.hpp).
    constexpr int KEEPN = ENC == kEncWide ? 4 : (MSW_B_KEEPN < RC ? MSW_B_KEEPN : RC);  // 8-byte records take twice the registers
    // (value records keep all 16: the alternative is a second exp per cell)
    double xv[KEEPN > 0 ? KEEPN : 1];
    // (ANY: the cells may refer to any entry of a hybrid slot area -- gathered from memory)
    auto fixed = [&](RT(&b)[RC], auto LEN, auto KEEP, auto ANY) {
      constexpr int L = decltype(LEN)::value;
      constexpr bool KP = decltype(KEEP)::value;
      constexpr int B = MSW_PASSB_BATCH;  // cells whose gathers are issued together
#pragma unroll
      for (int k0 = 0; k0 < L; k0 += B) {
        double ev[B];
        double2 xt[B];
#pragma unroll
        for (int k = 0; k < B; ++k) {
          if (k0 + k < L) {
            if constexpr (decltype(ANY)::value) ev[k] = E_(b[k0 + k]), xt[k] = XTg_(b[k0 + k]);
            else ev[k] = Eh_(b[k0 + k]), xt[k] = XT_(b[k0 + k]);
          }
        }
#pragma unroll
        for (int k = 0; k < B; ++k) {
          if (k0 + k < L) {
            zs = fma(ev[k], xt[k].x, zs);
            hs = fma(ev[k], xt[k].y, hs);
            // the scatter adds r_j * (x - p0) -- times f_g in the fixed-point build
            if constexpr (KP && KEEPN > 0)
              if (k0 + k < KEEPN) xv[k0 + k] = kFx ? fx_factor(ev[k], fxe) * xt[k].x : xt[k].x;
          }
        }
      }
    };
    auto cells = [&](RT(&b)[RC], uint32_t n, auto KEEP) {
      n += n & 1;  // an odd slice's missing last row is a null record in the registers (load_slice)
      // (RC = 8: the cases above 8 fold into the last one)
      auto at_most = [&](auto N) {
        constexpr int L = decltype(N)::value < RC ? decltype(N)::value : RC;
        fixed(b, std::integral_constant<int, L>{}, KEEP, std::false_type{});
      };
      switch (n) {
        case 0: break;
        case 2: at_most(std::integral_constant<int, 2>{}); break;
        case 4: at_most(std::integral_constant<int, 4>{}); break;
        case 6: at_most(std::integral_constant<int, 6>{}); break;
        case 8: at_most(std::integral_constant<int, 8>{}); break;
        case 10: at_most(std::integral_constant<int, 10>{}); break;
        case 12: at_most(std::integral_constant<int, 12>{}); break;
        case 14: at_most(std::integral_constant<int, 14>{}); break;
        default: at_most(std::integral_constant<int, 16>{}); break;
      }
    };
    // row sums of a slice from memory, pair by pair (rare shapes only: no second copy of the per-count code);
    // keeps nothing: the scatter gathers again
    auto pairs_any = [&](RT(&b)[RC], uint32_t n) {
#pragma unroll
      for (int k = 0; k < RC; k += 2) {
        if ((uint32_t)k < n) {  // (n odd: the pair's second record is the lane's null record, load_slice)
          const double e0 = E_(b[k]), e1 = E_(b[k + 1]);
          const double2 t0 = XTg_(b[k]), t1 = XTg_(b[k + 1]);
          zs = fma(e0, t0.x, zs);
          hs = fma(e0, t0.y, hs);
          zs = fma(e1, t1.x, zs);
          hs = fma(e1, t1.y, hs);
        }
      }
    };
    // scatter of up to RC cells held in b; padding records point at the lane's own sentinel
    // group: no test, no shared address
    // (kFx: rj is 2^K * r_j, the kept values are f_g * (x - p0); WIDE_ADD: the two-part adds)
    auto scatter = [&](RT(&b)[RC], uint32_t n, double rj, auto KEPT, auto WIDE_ADD) {
      constexpr bool KP = decltype(KEPT)::value;
      constexpr bool WA = decltype(WIDE_ADD)::value;
#pragma unroll
      for (int k = 0; k < RC; k += 2) {
        if ((uint32_t)k < n) {  // (n odd: the pair's second record is the lane's null record, load_slice)
          double x0, x1;
          // (KEPT rows: the slice's register rows with their kept values -- for index records a hot segment's rows)
          auto e_of = [&](RT r) { if constexpr (KP) return Eh_(r); else return E_(r); };
          auto xm_of = [&](RT r) { if constexpr (KP) return XMgh_(r); else return XMg_(r); };
          if constexpr (kFx) {
            x0 = KP && k < KEEPN ? xv[k < KEEPN ? k : 0] : fx_factor(e_of(b[k]), fxe) * xm_of(b[k]);
            x1 = KP && k + 1 < KEEPN ? xv[k + 1 < KEEPN ? k + 1 : 0] : fx_factor(e_of(b[k + 1]), fxe) * xm_of(b[k + 1]);
            if constexpr (WA) {
              addFXwidex(b[k], rj, x0, KEPT);
              addFXwidex(b[k + 1], rj, x1, KEPT);
            } else {
              addFXx(b[k], rj, x0, KEPT);
              addFXx(b[k + 1], rj, x1, KEPT);
            }
          } else {
            x0 = KP && k < KEEPN ? xv[k < KEEPN ? k : 0] : xm_of(b[k]);
            x1 = KP && k + 1 < KEEPN ? xv[k + 1 < KEEPN ? k + 1 : 0] : xm_of(b[k + 1]);
            addACCx(b[k], rj * x0, KEPT);
            addACCx(b[k + 1], rj * x1, KEPT);
          }
        }
      }
    };
    if (len <= (uint32_t)RC) {
      // index records: the slice is cut into a hot segment (rows in sb.r, LDS table) and a cold one of at most
      // kColdRows rows (sb.rc, table entries from memory: issued first, consumed after the hot rows) -- or,
      // with more cold cells than that, taken from memory as a whole (all rows in sb.r)
      [[maybe_unused]] double xc[kColdRows];
      uint32_t nsc = len, ncold = 0;  // rows the scatter finds in sb.r / in sb.rc
      bool kept = true;
      if constexpr (HYB) {
        const uint32_t nhot = sb.nhot;
        if (len - nhot <= (uint32_t)kColdRows) {
          ncold = len - nhot;
          nsc = nhot;
          double ce[kColdRows];
          double2 cx[kColdRows];
#pragma unroll
          for (int j = 0; j < kColdRows; ++j) {
            if ((uint32_t)j < ncold) {
              cx[j] = XTg_(sb.rc[j]);
              ce[j] = E_(sb.rc[j]);
            }
          }
          cells(sb.r, nhot, std::true_type{});
#pragma unroll
          for (int j = 0; j < kColdRows; ++j) {
            if ((uint32_t)j < ncold) {
              zs = fma(ce[j], cx[j].x, zs);
              hs = fma(ce[j], cx[j].y, hs);
              xc[j] = kFx ? fx_factor(ce[j], fxe) * cx[j].x : cx[j].x;
            }
          }
        } else {
          pairs_any(sb.r, len);
          kept = false;
        }
      } else {
        cells(sb.r, len, std::true_type{});
      }
      // scatter of the slice: the rows held in sb.r, then the cold rows
      auto scatter_all = [&](double rs, auto WIDE_ADD) {
        if (kept) scatter(sb.r, nsc, rs, std::true_type{}, WIDE_ADD);
        else scatter(sb.r, nsc, rs, std::false_type{}, WIDE_ADD);
        if constexpr (HYB) {
#pragma unroll
          for (int j = 0; j < kColdRows; ++j) {
            if ((uint32_t)j < ncold) {
              if constexpr (kFx) {
                if constexpr (decltype(WIDE_ADD)::value) addFXwide(sb.rc[j], rs, xc[j]);
                else addFX(sb.rc[j], rs, xc[j]);
              } else {
                addACC(sb.rc[j], rs * xc[j]);
              }
            }
          }
        }
      };
      // an EC over several lanes (sell.hpp slice classes): its partial row sums meet in every one of them -- the
      // same bits, so all take the same branches below -- and its first lane alone adds to the scalar sums
      // (ML = false: every slice one lane per EC)
      const uint32_t lgm = ML ? sb.lgm : 0u;
      bool spoke = true;
      if (ML && lgm != 0u) {
        zs = group_sum(zs, lgm);
        hs = group_sum(hs, lgm);
        spoke = ((uint32_t)lane & ((1u << lgm) - 1u)) == 0u;
      }
      if (c != 0.0 && !(zbase + zs >= gthr)) {
        if (spoke) defer(S.n_long + (ML ? slice_geo(S.cls, sb.sl).ec0 + ((uint32_t)lane >> lgm) : sb.sl * 64 + lane));
      } else if (c != 0.0) {
        const double Z = zbase + zs;
        const double rj = ec_div(c, Z);
        double rja = rj;  // what the scalar sums see of this lane
        if (ML && lgm != 0u) rja = spoke ? rj : 0.0;
        s_rH = fma(rja, hs, s_rH);  // (sum r_j H_j = hbase sum r_j + sum r_j hs_j: the first part at the end, from s_W)
        s_W += rja;
        if constexpr (kFx) {
          const double rs = rj * fxs;
          // rs (Z + zbase) = 2^K (c + r_j zbase) and rs fxb against 2^51: two comparisons with per-pass constants
          if (fma(rj, zbase, c) < fxt1 && rj < fxt2) scatter_all(rs, std::false_type{});
          else scatter_all(rs, std::true_type{});
        } else {
          scatter_all(rj, std::false_type{});
        }
        // sum c log Z after the scatter (its registers are free by now).  For the multiplicities
        // 1..15 the logarithm is deferred: the mantissas are multiplied up per lane and one log per 64
        // slices is taken of the product (flush_logs): ~8 operations per EC instead of ~40, and if anything
        // a smaller rounding error than the sum of logs.  1..3 -- nearly all ECs of an alignment -- take two
        // conditional multiplications; a bootstrap replicate's counts (Poisson-like: one EC in fifty at 4 or
        // more, so most wavefronts hold one) take m^c by binary powering, still a third of a logarithm, which
        // the whole wavefront would otherwise pay for its one lane (pass B 99 -> 88 us on resampled counts).
        // Both give the same bits for 1..3.
        __builtin_amdgcn_sched_barrier(0);
        if (sb.c8 <= 15u) {
          // (Z passed the guard test: finite, positive, normal -- the instructions themselves, without frexp()'s
          // handling of zeros, infinities and NaNs)
          const int ez = __builtin_amdgcn_frexp_exp(Z);
          const double m = __builtin_amdgcn_frexp_mant(Z);
          double r;
          if (__builtin_amdgcn_ballot_w64(sb.c8 > 3u) == 0) {  // wave-uniform
            r = m;
            r *= sb.c8 >= 2u ? m : 1.0;
            r *= sb.c8 >= 3u ? m : 1.0;
          } else {
            const double m2 = m * m, m4 = m2 * m2, m8 = m4 * m4;
            r = (sb.c8 & 1u) ? m : 1.0;
            r *= (sb.c8 & 2u) ? m2 : 1.0;
            r *= (sb.c8 & 4u) ? m4 : 1.0;
            r *= (sb.c8 & 8u) ? m8 : 1.0;
          }
          int ezc = __mul24(ez, (int)sb.c8);  // |ez| < 2^11, c8 < 2^4
          if (ML && lgm != 0u && !spoke) r = 1.0, ezc = 0;
          lp_mant *= r;  // >= 2^-15 per slice: 2^-960 between two flushes
          lp_exp += ezc;
        } else if (spoke) {
          s_clogZ += c * log(Z);
        }
      }
    } else {
      // more cells per EC than the registers hold (the stream has not fetched this slice): chunks of
      // RC records through the same registers -- once for the row sums, once more (from L2)
      // for the scatter; the other wavefronts of the workgroup cover each chunk's load latency
      // (registers of their own: see pass A)
      // (keeping the first chunk's f_g (x - p0) for the scatter, like a short slice's, was tried in round 3: the
      // values live across the chunk loops push every instantiation into 300 bytes of scratch)
      const size_t base = (size_t)o * 64 + lane;
      RT t[RC];
      uint32_t k0 = 0;
      for (; k0 + (uint32_t)RC <= len; k0 += RC) {
        load_slice<ENC>(S.rec, base + (size_t)k0 * 64, RC, t, null_rec);
        fixed(t, std::integral_constant<int, RC>{}, std::false_type{}, std::true_type{});
      }
      if (k0 < len) {
        const uint32_t n = len - k0;
        load_slice<ENC>(S.rec, base + (size_t)k0 * 64, n, t, null_rec);
        pairs_any(t, n);
      }
      if (c != 0.0 && !(zbase + zs >= gthr)) {  // (slices of more than 16 rows hold one lane per EC: lgm = 0)
        defer(S.n_long + (ML ? slice_geo(S.cls, sb.sl).ec0 : sb.sl * 64) + lane);
      } else if (c != 0.0) {
        const double Z = zbase + zs;
        const double rj = ec_div(c, Z);
        s_clogZ += c * log(Z);
        s_rH = fma(rj, hs, s_rH);
        s_W += rj;
        const double rs = kFx ? rj * fxs : rj;
        const bool narrow = !kFx || (fma(rj, zbase, c) < fxt1 && rj < fxt2);
        for (k0 = 0; k0 < len; k0 += RC) {
          const uint32_t n = len - k0 < (uint32_t)RC ? len - k0 : (uint32_t)RC;
          load_slice<ENC>(S.rec, base + (size_t)k0 * 64, n, t, null_rec);
          if (narrow) scatter(t, n, rs, std::false_type{}, std::false_type{});
          else scatter(t, n, rs, std::false_type{}, std::true_type{});
        }
      }
    }
  };
  if constexpr (PRIME) stream.run(issue, process, flush_logs, &first);
  else stream.run(issue, process, flush_logs);
  // long ECs (plain CSR): one wavefront per EC, a cell per lane and step (see pass A)
  // The current EC's first kLongStep * 64 records stay in registers for the scatter (one reload less: 10 % on
  // ECs of 300..1000 cells), and the next EC's first ones are fetched before the current one is
  // processed.  (The same prefetch changes nothing in pass A, which keeps the plain loop: the
  // wavefront-per-EC path is not bound by its load latency -- DESIGN.md 7.)
  {
    uint32_t r = stream.s_first, c0 = 0, c1 = 0;
    if (r < S.n_long) c0 = S.long_ptr[r], c1 = S.long_ptr[r + 1];
    // kLongStep records per lane starting at cell kb of an EC that ends at k1 (wave-uniform test
    // first: whole 64-cell groups past the end cost nothing)
    auto load_long = [&](uint32_t kb, uint32_t k1, RT(&dst)[kLongStep]) {
#pragma unroll
      for (int u = 0; u < kLongStep; ++u) {
        dst[u] = null_rec;  // (a separate branch for rows that lie wholly inside the EC -- no per-lane test -- was
                            // measured in round 3: 4-6 % SLOWER on ECs of 300..1000 cells)
        if (kb + 64u * u < k1 && kb + 64u * u + lane < k1) dst[u] = R::load(S.rec_long, kb + 64u * u + lane);
      }
    };
    RT pre[kLongStep];
    load_long(c0, c1, pre);
    uint32_t n_deferred = 0;  // ECs whose logarithm sits in lp_mant / lp_exp (wave-uniform)
    while (r < S.n_long) {
      const uint32_t rn = r + stream.nw;
      uint32_t n0 = 0, n1 = 0;
      if (rn < S.n_long) n0 = S.long_ptr[rn], n1 = S.long_ptr[rn + 1];
      const double c = S.cvec[r];
      RT first[kLongStep], rc[kLongStep];
#pragma unroll
      for (int u = 0; u < kLongStep; ++u) first[u] = rc[u] = pre[u];
      load_long(n0, n1, pre);
      double zs = 0.0, hs = 0.0;
      double pkf[kLongStep];  // f_g (x - p0) of the first step's cells: no second gather (or exponential) for them
      for (uint32_t kb = c0;;) {
#pragma unroll
        for (int q = 0; q < kLongStep; q += 4) {
          if (kb + 64u * q < c1) {  // wave-uniform: quads past the end are skipped
            double eg[4];
            double2 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) eg[u] = E_(rc[q + u]), t[u] = XTg_(rc[q + u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              zs = fma(eg[u], t[u].x, zs);
              hs = fma(eg[u], t[u].y, hs);
            }
#if MSW_LONG_KEEP
            if (kb == c0) {
#pragma unroll
              for (int u = 0; u < 4; ++u) pkf[q + u] = kFx ? fx_factor(eg[u], fxe) * t[u].x : t[u].x;
            }
#endif
          }
        }
        kb += kLongStep * 64;
        if (kb >= c1) break;
        load_long(kb, c1, rc);
      }
      zs = wave_sum(zs);
      hs = wave_sum(hs);
      if (c != 0.0 && !(zbase + zs >= gthr)) {  // wave-uniform
        if (lane == 0) defer(r);
      } else if (c != 0.0) {
        const double Z = zbase + zs;
        const double rj = ec_div(c, Z);
        // c log Z deferred as in the slices (multiplicities 1..15: the mantissas multiplied up in lane 0, one
        // logarithm per 32 ECs): the full logarithm is 70 dependent instructions in the middle of a
        // wavefront's one-EC-at-a-time chain
        const uint32_t c8l = uniform((uint32_t)S.c8[r]);
        if (c8l <= 15u) {
          if (lane == 0) {
            int ez;
            const double m = frexp(Z, &ez);
            const double m2 = m * m, m4 = m2 * m2, m8 = m4 * m4;
            double pw = (c8l & 1u) ? m : 1.0;
            pw *= (c8l & 2u) ? m2 : 1.0;
            pw *= (c8l & 4u) ? m4 : 1.0;
            pw *= (c8l & 8u) ? m8 : 1.0;
            lp_mant *= pw;
            lp_exp += ez * (int)c8l;
          }
          if (++n_deferred == 32u) {
            flush_logs();
            n_deferred = 0;
          }
        } else if (lane == 0) {
          s_clogZ += c * log(Z);
        }
        if (lane == 0) {
          s_rH = fma(rj, hs, s_rH);
          s_W += rj;
        }
#pragma unroll
        for (int u = 0; u < kLongStep; ++u) rc[u] = first[u];
        const double rs = kFx ? rj * fxs : rj;
        const bool narrow = !kFx || (fma(rj, zbase, c) < fxt1 && rj < fxt2);  // wave-uniform
        for (uint32_t kb = c0;;) {
#pragma unroll
          for (int q = 0; q < kLongStep; q += 4) {
            if (kb + 64u * q < c1) {
              double xm[4];
              if (MSW_LONG_KEEP && kb == c0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) xm[u] = pkf[q + u];
              } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) xm[u] = kFx ? fx_factor(E_(rc[q + u]), fxe) * XMg_(rc[q + u]) : XMg_(rc[q + u]);
              }
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                if constexpr (kFx) {
                  if (narrow) addFX(rc[q + u], rs, xm[u]);
                  else addFXwide(rc[q + u], rs, xm[u]);
                } else {
                  addACC(rc[q + u], rj * xm[u]);
                }
              }
            }
          }
          kb += kLongStep * 64;
          if (kb >= c1) break;
          load_long(kb, c1, rc);
        }
      }
      r = rn, c0 = n0, c1 = n1;
    }
    flush_logs();
  }
  // guarded ECs (sell.hpp): a wavefront each, every group visited.  Each group receives its share of the
  // EC's c_j directly -- listed: c e_g x / Z, not listed: c e_g p0 / Z -- and the EC stays out of
  // W = sum r_j (the background share of the ordinary ECs).
  __syncthreads();
  const uint32_t n_guard = (GMODE != 4 || rg.first) ? min(*(lds_u32_t *)(size_t)gcnt_off, GD.cap) : 0u;  // once, not per range run
  if (n_guard) {
    if (tid == 0) atomicAdd(GD.visits, (unsigned long long)n_guard);
    const uint32_t wv = uniform(tid >> 6), nwv = NT / 64;
    uint32_t *bits = GD.bits + (size_t)(blockIdx.x * 16 + wv) * GD.words;
    const double a = uniform_d(sc->a), logzi = uniform_d(sc->logzi), tref = uniform_d(sc->tref);
    // A group's share of a guarded EC, in reads, goes to two global 64-bit fixed-point accumulators
    // of the group (units 2^-t and 2^-(t+36) reads, 2^t = Scalars::fx_tscale): such a share can exceed any
    // bound the group's own exponent allows for (c e_g p0 / Z with Z far below the background sum),
    // while the shares of one EC always add up to c_j.  Global integer atomics: rare path.
    const double tls = uniform_d(sc->fx_tscale);
    auto add_share = [&](uint32_t g, double share) {
      const double vh = fma(share, tls, kFxMagic);
      const double qh = vh - kFxMagic;                     // rint(share * 2^t), exact
      const double ql = fma(share, tls, -qh) * 0x1p36;     // remainder, |.| <= 2^35
      atomicAdd(&GD.tail[2 * (size_t)g], fx_bits(1.0, qh));
      atomicAdd(&GD.tail[2 * (size_t)g + 1], fx_bits(1.0, ql));
    };
    for (uint32_t i = wv; i < n_guard; i += nwv) {
      const uint32_t p = GD.list[(size_t)blockIdx.x * GD.cap + i];
      const double c = S.cvec[p];
      double z = 0.0, h = 0.0;
      wave_cells<ENC>(S, p, (uint32_t)lane, [&](uint32_t g, double T) {
        const double q = e_g[g] * exp(a * (T - tref));
        atomicOr(&bits[g >> 5], 1u << (g & 31));
        z += q;
        h = fma(q, T, h);
      });
      __builtin_amdgcn_s_waitcnt(0);  // the bitmap updates have reached memory before it is read back
      double r0 = 0.0;
      for (uint32_t w0 = lane; w0 < GD.words; w0 += 64) {
        const uint32_t listed = atomicOr(&bits[w0], 0u);
        for (uint32_t b = 0; b < 32; ++b) {
          const uint32_t g = w0 * 32 + b;
          if (g < G && !((listed >> b) & 1u)) r0 += e_g[g];
        }
      }
      z = wave_sum(z), h = wave_sum(h), r0 = wave_sum(r0);
      const double Z = fma(p0, r0, z), H = fma(p0 * logzi, r0, h);
      if (Z > 0.0) {
        const double rj = c / Z;
        if (lane == 0) {
          s_clogZ += c * log(Z);
          s_rH += rj * H;
        }
        wave_cells<ENC>(S, p, (uint32_t)lane, [&](uint32_t g, double T) {
          add_share(g, e_g[g] * rj * exp(a * (T - tref)));
        });
        for (uint32_t w0 = lane; w0 < GD.words; w0 += 64) {
          const uint32_t listed = atomicAnd(&bits[w0], 0u);  // read and clear
          for (uint32_t b = 0; b < 32; ++b) {
            const uint32_t g = w0 * 32 + b;
            if (g < G && !((listed >> b) & 1u)) add_share(g, e_g[g] * (rj * p0));
          }
        }
      } else {
        if (lane == 0) *GD.err = 1;  // no probability under any group: the solve reports it
        for (uint32_t w0 = lane; w0 < GD.words; w0 += 64) atomicAnd(&bits[w0], 0u);
      }
      __builtin_amdgcn_s_waitcnt(0);
    }
  }
  s_rH = fma(hbase, s_W, s_rH);  // the background part of sum r_j H_j (the ordinary ECs' r_j: exactly those in s_W)
  {  // the three ELBO sums of the workgroup with one pair of barriers (the slice geometry behind the 32 doubles of
     // reduction scratch is free by now: 48 doubles needed)
    double t3[3] = {s_clogZ, s_rH, s_W};
    block_sum_n<3>(t3, sh + 32);
    if (tid == 0 && (GMODE != 4 || rg.first)) {
      partS[4 * blockIdx.x + 0] = t3[0];
      partS[4 * blockIdx.x + 1] = t3[1];
      partS[4 * blockIdx.x + 2] = t3[2];
      partS[4 * blockIdx.x + 3] = 0.0;
    }
  }
  if (tid == 0) MSW_STAMP_MAX(sc->iter, 2, 6);
  if (ALDS) {
    __syncthreads();
    const double *al = reinterpret_cast<const double *>(smem + bhi + acc_off);
    double *dst = partAcc + (size_t)blockIdx.x * G;
    if (GMODE == 4) {
      for (uint32_t g = tid; g < rg.n; g += NT) dst[rg.g0 + g] = al[g];
    } else {
      for (uint32_t g = tid; g < G; g += NT) dst[g] = al[g];
    }
  }
  if (tid == 0) MSW_STAMP_MAX(sc->iter, 2, 7);
}

}  // namespace msw
