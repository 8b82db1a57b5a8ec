// msweep_core.hip -- host side of libmsweep_core.so: the C ABI declared in
// include/msweep_core.h over the gfx950 kernels in kernels.hpp.
//
// No CPU fallback exists in this library: every numeric step of the hot path (likelihood
// expansion, RCG / EM sweeps, column reductions, ELBO, bootstrap resampling) is a HIP
// kernel; the host only validates shapes, lays out buffers and enqueues launches.
#include "../../include/msweep_core.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <chrono>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "kernels.hpp"
#include "pack_kernels.hpp"
#include "compress_kernels.hpp"
#include "comm.hpp"
#include "shm_comm.hpp"
#include "peer_comm.hpp"
#include "likelihood_kernels.hpp"
#include "bootstrap_kernels.hpp"
#include "em_kernels.hpp"
#include "em_f32_kernels.hpp"
#include "stream_kernels.hpp"
#include "reader_kernels.hpp"

using namespace msw;

namespace {
thread_local std::string g_create_error;
constexpr size_t kLdsMax = 160 * 1024;
constexpr int kIterBatch = 16;
}  // namespace

struct msw_core {
  int device = 0;
  int n_cu = 256;
  hipStream_t stream = nullptr;
  std::string err;
  TextStager text_stage;  // pinned staging of msw_alignment_read_device (host_reader.inc)
  ReaderPool reader_pool;  // ... and its device memory, kept between calls (reader_kernels.hpp)

  // ---- resident likelihood -----------------------------------------------------------
  int flavor = -1;  // -1 none, 0 CSR-of-ECs, 1 dense
  uint32_t G = 0, E = 0, n_lut = 0, nslices = 0, n_long = 0;
  uint64_t nnz = 0, nslots = 0;
  int enc = kEncNarrow;  // record encoding (sell.hpp): narrow byte offsets / wide / index records (hybrid area)
  bool glds = true, tlds = true;
  int gmodeB = 1;  // k_passB GMODE (sweep_kernels.hpp)
  uint32_t enc_shift = 0, enc_mask = 0, enc_bhi = 0, enc_bhiA = 0;  // record encoding (sell.hpp)
  uint32_t enc_shiftH = 0, enc_maskH = 0;               // index records: the rows of a hot segment
  uint32_t n_area = 0;                                  // 16-byte entries of the slot area
  uint32_t n_tab_lds = 0;                               // ... of which the LDS images hold (all, the hot head, none)
  DevBuf<uint8_t> slice_hot;                            // index records: rows of every slice's hot segment
  SliceClasses cls = {};                                // slice classes: lanes per EC (sell.hpp)
  bool no_hybrid = false;                               // re-planning without the hybrid area (host_pack.inc: slice geometry beyond 2^27 rows)
  bool pack_schedule = true;                            // LDS-bank scheduling of the cells at upload (msw_core_set_pack_schedule)
  bool packed_scheduled = false;                        // ... as the resident likelihood was packed
  bool wide() const { return enc == kEncWide; }
  bool hybrid() const { return enc == kEncIndex; }
  RecDec dec() const { return RecDec{enc_shift, enc_mask, enc_bhi, enc_bhiA, enc_shiftH, enc_maskH}; }
  uint32_t long_row = kLongRow;                         // ECs with more cells go one per wavefront (reset_likelihood)
  uint64_t rows_over8 = 0;                              // rows of the slices of more than 8 rows (finish_sell)
  bool passB_rc8 = false;                               // pass B runs its short-slice instantiation (sweep_kernels.hpp, RC = 8)
  DevBuf<uint32_t> area_slot;
  DevBuf<double> lut_area;  // lut[area_slot[i]]: what the per-slot tables are built from, in their order
  DevBuf<int> tab_built;    // k_tables bookkeeping
  int n_tab_inline() const { return flavor == 0 && n_area <= (uint32_t)kTabInline ? (int)n_area : 0; }
  double logzi = 0.0;
  DevBuf<uint32_t> rec, slice_off, long_ptr, rec_long, perm;
  DevBuf<uint32_t> iperm;  // original EC index -> permuted position (gamma blocks; built on first use)
  DevBuf<double> lut, Lt;
  int nblk = 0;      // persistent workgroups of the CSR sweeps
  int nblk_dense = 0;
  int nreg = 0;
  // rows of per-workgroup (or per-wave) column-sum partials pass B leaves for k_redfin
  int npart_rows() const { return flavor == 0 ? nblk : (nreg >= 32 ? 4 * nblk_dense : nblk_dense); }

  // ---- solve state ---------------------------------------------------------------------
  DevBuf<double> cvec, logc_d, alpha0, u, os_u, step_u, w, e, N, Nc, Acc;
  DevBuf<uint8_t> c8, c8s;  // byte image of cvec by EC position / by slice lane (sell.hpp)
  DevBuf<double> logc_res;  // log counts left on the device by msw_core_build_likelihood
  bool have_logc_res = false;
  DevBuf<double2> ew, tabA, tabB;  // group table of pass A; per-slot tables of both sweeps (TabDev)
  TabDev tabs() const { return TabDev{tabA.p, tabB.p}; }
  DevBuf<double> partA, partS, partAcc, partC, partR, totS;
  // EC-sharded solve: this handle holds one rank's block of ECs (comm.hpp)
  size_t lds_attr[3][80] = {};  // dynamic-LDS limit already granted per sweep instantiation ([2]: pass B's short-slice ones)
  msw_comm *comm = nullptr;
  bool in_collective = false;  // a solve / sharded build is under way: a failure now strands the peers (guarded())
  DevBuf<double> commA, commB;  // 1 and G + 4 doubles
  // guarded ECs (sell.hpp): per-workgroup lists, per-wavefront bitmaps, error flag
  DevBuf<uint32_t> guard_list, guard_bits;
  DevBuf<unsigned long long> guard_tail;  // [2 G] the guarded ECs' shares per group, two fixed-point limbs
  DevBuf<int> guard_err;
  DevBuf<unsigned long long> guard_visits;
  DevBuf<double> trange;  // {max, min} of the table values (bounds x_i = exp(a T_i) per pass: Scalars::xb)
  uint32_t guard_cap = 0, guard_words = 0;
  GuardDev guard_view() const {
    return GuardDev{guard_list.p, guard_bits.p, guard_tail.p, lut_area.p, guard_err.p, guard_visits.p, guard_cap, guard_words};
  }
  DevBuf<Scalars> sc;
  Scalars *sc_host = nullptr;  // pinned
  DevBuf<double> tr_bound, tr_newnorm, tr_beta, tr_theta;
  DevBuf<int32_t> tr_reset;
  size_t trace_theta = 0;
  bool have_solution = false;
  bool prepared = false;
  int last_algo = MSW_ALGO_RCG;
  SolveOpts opts;  // msw_core_set_option
  // EM state
  DevBuf<double> logth;
  DevBuf<float> e32, tab32;  // --emprecision float: e_g and the slot table {x_i - p0} as floats (em_f32_kernels.hpp)
  bool em_f32 = false;       // the EM run under way is served by the fp32 kernels (launch_passB)

  // ---- bootstrap -------------------------------------------------------------------------
  DevBuf<double> cp;
  std::vector<uint32_t> cp_counts;  // the EC counts `cp` was made from (host_bootstrap.inc: kept across calls)
  bool cp_hit = false;              // ... and whether the last call found them again
  DevBuf<uint64_t> mtwords;
  DevBuf<uint32_t> bcounts, bcounts2;
  DevBuf<MtState> mt;
  hipStream_t stream2 = nullptr;  // resampling of the next replicate, under the current solve
  hipEvent_t ev_counts[2] = {nullptr, nullptr};
  uint32_t one_count = 0;
  bool mt_valid = false;
  int32_t mt_seed = 0;
  uint64_t mt_pos = 0;
  // further solver states on the same resident likelihood (borrowed buffers, streams of their own):
  // the bootstrap driver runs several replicates at a time (host_bootstrap.inc)
  std::vector<std::unique_ptr<msw_core>> clones;

  // ---- measurement ---------------------------------------------------------------------------
  bool profiling = false, fixed_iters = false;
  msw_timing timing = {};
  msw_bootstrap_timing btiming = {};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> evA, evB, evC;  // pass A, pass B, collectives (sharded solve)
  size_t evA_used = 0, evB_used = 0, evC_used = 0;

  ~msw_core() {
    if (sc_host) (void)hipHostFree(sc_host);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    for (auto &p : evA) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto &p : evB) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto &p : evC) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto &e : ev_counts)
      if (e) (void)hipEventDestroy(e);
    if (stream2) (void)hipStreamDestroy(stream2);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

namespace {

struct Fail : std::runtime_error {
  using std::runtime_error::runtime_error;
};
// a solve that failed NUMERICALLY (likelihood underflow, non-finite bound: where the reference returns NaN weights).
// The bootstrap driver turns exactly these into a row of NaN; every other failure fails the call.
struct NumericFail : Fail {
  using Fail::Fail;
};

// marks the stretch of a call in which this rank's peers wait for it in collectives
struct CollectiveScope {
  msw_core *h;
  explicit CollectiveScope(msw_core *h_) : h(h_) { h->in_collective = true; }
  void leave() { h->in_collective = false; }  // on success only: guarded() reads the flag after a throw
};

template <class F>
int guarded(msw_handle h, F &&f) {
  if (!h) return 1;
  try {
    MSW_HIP(hipSetDevice(h->device));
    f();
    return 0;
  } catch (const std::exception &ex) {
    h->err = ex.what();
    (void)hipGetLastError();
    // peers of a sharded solve must not wait for this rank for ever -- but only a failure BETWEEN collectives
    // (a solve or a sharded build was under way) strands them: argument and state errors touch no collective
    if (h->comm && h->in_collective) h->comm->abort();
    h->in_collective = false;
    return 1;
  }
}

std::pair<hipEvent_t, hipEvent_t> &next_pair(std::vector<std::pair<hipEvent_t, hipEvent_t>> &v,
                                             size_t &used) {
  if (used == v.size()) {
    hipEvent_t a, b;
    MSW_HIP(hipEventCreate(&a));
    MSW_HIP(hipEventCreate(&b));
    v.emplace_back(a, b);
  }
  return v[used++];
}

SellDev sell_view(msw_core *h) {
  SellDev S;
  S.rec = h->rec.p;
  S.slice_off = h->slice_off.p;
  S.long_ptr = h->long_ptr.p;
  S.rec_long = h->rec_long.p;
  S.perm = h->perm.p;
  S.cvec = h->cvec.p;
  S.c8 = h->c8.p;
  S.c8s = h->c8s.p;
  S.nslices = h->nslices;
  S.n_long = h->n_long;
  S.n_ecs = h->E;
  S.n_groups = h->G;
  S.n_lut = h->n_lut;
  S.n_area = h->n_area;
  S.area_slot = h->area_slot.p;
  S.shift = h->enc_shift;
  S.mask = h->enc_mask;
  S.bhi = h->enc_bhi;
  S.bhiA = h->enc_bhiA;
  S.shiftH = h->enc_shiftH;
  S.maskH = h->enc_maskH;
  S.n_tab_lds = h->n_tab_lds;
  S.slice_hot = h->slice_hot.p;
  S.lut_area = h->lut_area.p;
  S.cls = h->cls;
  return S;
}

// pass B's mode for a given placement of the group vectors (glds) and n_tab slot entries in LDS; -1 = no fit
int passB_mode(const msw_core *h, bool glds, uint32_t n_tab, bool index) {
  const uint32_t G = h->G;
  if (glds) {
    if (pass_lds_bytes(1, n_tab, G, false, index) > kLdsMax) return -1;
    // column sums at the fixed immediate distance when e_g fits below it and the image still fits
    if (8ull * (G + kSentinels) <= kAccFixed && pass_lds_bytes(2, n_tab, G, false, index) <= kLdsMax) return 2;
    return 1;
  }
  if (pass_lds_bytes(0, n_tab, G, false, index) > kLdsMax) return -1;
  if (getenv("MSWEEP_GLOBAL_ATOMICS")) return 0;  // developer switch: mode 0 (column sums in HBM)
  // too many groups for {e, w} / e + sums in LDS: the column sums alone may still fit (mode 3) ...
  if (pass_lds_bytes(3, n_tab, G, false, index) <= kLdsMax) return 3;
  // ... and beyond that one range of groups at a time does (mode 4), whatever the group count
  if (pass_lds_bytes(4, n_tab, G, false, index) <= kLdsMax) return 4;
  return 0;
}

// MSWEEP_MULTILANE=0 (developer switch): every EC of up to 256 cells one lane (slices of up to 256 rows on the
// streaming path of the sweeps), the layout of rounds 1-2
bool multilane() {
  const char *e = getenv("MSWEEP_MULTILANE");
  return !(e && atoi(e) == 0);
}
// slice / position boundaries of the classes from the ECs per class (n[c]: class c = 64 >> c lanes per EC)
SliceClasses make_slice_classes(const uint32_t *n) {
  SliceClasses C = {};
  for (int c = 0; c < kSliceClasses; ++c) {
    const uint32_t per = 64u >> (kMaxLgm - c);
    C.p0[c + 1] = C.p0[c] + n[c];
    C.s0[c + 1] = C.s0[c] + (n[c] + per - 1) / per;
  }
  return C;
}

void choose_lds_mode(msw_core *h) {
  const bool opts[4][2] = {{true, true}, {true, false}, {false, true}, {false, false}};
  const char *force = getenv("MSWEEP_FORCE_LDS");  // developer switch: "gt", e.g. "10" = groups in LDS, slots not
  for (auto &o : opts) {
    if (force && strlen(force) == 2 && (o[0] != (force[0] == '1') || o[1] != (force[1] == '1'))) continue;
    const uint32_t n_tab = o[1] ? h->n_area : 0u;
    const int gmB = passB_mode(h, o[0], n_tab, false);
    // (too many groups for the group vectors AND a slot table that leaves no room for the column sums: the table
    // goes to memory -- or into the hybrid area -- rather than the column sums into HBM atomics, 20 x the cost)
    if (!force && !o[0] && o[1] && gmB == 0 && !getenv("MSWEEP_GLOBAL_ATOMICS")) continue;
    if (gmB >= 0 && pass_lds_bytes(o[0] ? 1 : 0, n_tab, h->G, true, false) <= kLdsMax) {
      h->glds = o[0];
      h->tlds = o[1];
      h->gmodeB = gmB;
      h->n_tab_lds = n_tab;
      return;
    }
  }
  throw Fail("internal: no LDS configuration fits");
}

// LDS mode, then the record encoding that goes with it (sell.hpp): the narrowest split of a
// 32-bit record that holds both byte offsets, else 8-byte records.  Runs before the SELL packing.
void choose_layout(msw_core *h) {
  choose_lds_mode(h);
  h->enc_bhi = sell_bhi(h->n_tab_lds);
  h->enc_bhiA = 2 * h->enc_bhi;
  const uint64_t lo_end = 16ull * std::max<uint32_t>(h->n_area, 1);              // lo < lo_end
  const uint64_t hi_end = (uint64_t)h->enc_bhi + 8ull * ((uint64_t)h->G + kSentinels);  // hi < hi_end
  h->enc = kEncWide;
  h->enc_shift = 0;
  h->enc_mask = 0xffffffffu;
  const char *force = getenv("MSWEEP_RECORD_BYTES");  // developer switch: 8 = skip the 4-byte formats
  for (uint32_t s = 5; s <= 31 && !(force && atoi(force) == 8); ++s) {
    if (lo_end <= (1ull << (s - 1)) && hi_end <= (1ull << (32 - s))) {
      h->enc = kEncNarrow;
      h->enc_shift = s;
      h->enc_mask = (1u << (s - 1)) - 1u;
      break;
    }
  }
  if (h->wide() && (hi_end > (1ull << 31) || lo_end > (1ull << 32)))
    throw Fail("likelihood too large: group / lookup-table offsets exceed the 8-byte record fields");
}

// The hybrid slot area (sell.hpp, index records) for likelihoods whose slot tables do not fit LDS beside the
// group vectors: 4-byte records of (group, entry) INDICES when both fit 32 bits, the group vectors where
// choose_lds_mode put them, and as many of the most-used entries in LDS as both sweeps' images leave room for.
// Returns false (layout untouched) when it does not apply.
bool choose_hybrid_layout(msw_core *h) {
  if (h->tlds || h->no_hybrid) return false;
  if (const char *e = getenv("MSWEEP_HYBRID"))  // developer switch: 0 = the all-memory tables (and wide records)
    if (atoi(e) == 0) return false;
  const char *force = getenv("MSWEEP_RECORD_BYTES");
  if (force && atoi(force) == 8) return false;
  uint32_t eb = 1, gb = 1;
  while ((1ull << eb) < std::max<uint32_t>(h->n_area, 2)) ++eb;
  while ((1ull << gb) < (uint64_t)h->G + kSentinels) ++gb;
  if (eb + gb > 32) return false;
  // room for the table: the largest multiple of 16 entries (256 bytes) that both images hold
  uint32_t n_hot = 0;
  {
    const int gmA = h->glds ? 1 : 0;
    uint32_t lo = 0, hi = std::min<uint32_t>(h->n_area, (uint32_t)(kLdsMax / 16)) / 16;  // in units of 16 entries
    while (lo < hi) {
      const uint32_t mid = (lo + hi + 1) / 2, n = mid * 16;
      const bool ok = pass_lds_bytes(gmA, n, h->G, true, true) <= kLdsMax && passB_mode(h, h->glds, n, true) >= 0 &&
                      (h->glds || passB_mode(h, false, n, true) == passB_mode(h, false, 0, true));
      if (ok) lo = mid;
      else hi = mid - 1;
    }
    n_hot = lo * 16;
  }
  if (const char *e = getenv("MSWEEP_HYBRID_HOT")) n_hot = std::min<uint32_t>(n_hot, (uint32_t)atoi(e) & ~15u);  // developer switch
  h->enc = kEncIndex;
  h->enc_shift = eb;
  h->enc_mask = (1u << eb) - 1u;
  h->n_tab_lds = std::min(n_hot, h->n_area);
  // the rows of a hot segment carry 16 * entry (entry < n_tab_lds): as many bits as the table's LDS image takes
  uint32_t hb = 4;
  while ((1ull << hb) < 16ull * std::max<uint32_t>(h->n_tab_lds, 1)) ++hb;
  if (hb + gb > 32) return false;  // (cannot happen while the table's image and the group vectors share 160 KB of LDS)
  h->enc_shiftH = hb;
  h->enc_maskH = (1u << hb) - 1u;
  h->enc_bhi = h->enc_bhiA = sell_bhi(h->n_tab_lds);
  h->gmodeB = passB_mode(h, h->glds, h->n_tab_lds, true);
  return true;
}

void alloc_solve_state(msw_core *h) {
  const uint32_t G = h->G, E = h->E;
  for (DevBuf<double> *b : {&h->alpha0, &h->u, &h->os_u, &h->step_u, &h->w, &h->e, &h->N, &h->Nc,
                            &h->Acc, &h->logth})
    b->alloc((size_t)G + kSentinels);
  h->ew.alloc((size_t)G + kSentinels);
  // entries G.. of e / ew are the sentinel groups of SELL padding records: zero, never rewritten
  h->e.zero(h->stream);
  h->ew.zero(h->stream);
  h->cvec.alloc(E);
  h->c8.alloc((size_t)E + 64);
  h->c8s.alloc((size_t)h->nslices * 64 + 64);  // lanes without an EC stay 0: no EC
  h->c8s.zero(h->stream);
  h->logc_d.alloc(E);
  h->tabA.alloc((size_t)std::max<uint32_t>(h->n_area, 1));
  h->tabB.alloc((size_t)std::max<uint32_t>(h->n_area, 1));
  h->tab_built.alloc(2);
  if (h->flavor != 0) h->lut_area.alloc(1);
  const int nb = std::max(h->nblk, std::max(h->nblk_dense, h->npart_rows()));
  h->partA.alloc(std::max(nb, 1024));
  h->partS.alloc(4 * (size_t)std::max(nb, 1024));
  h->partR.alloc(kRedfinParts * ((size_t)G / kRedfinGroups + 2));
  h->totS.alloc(4);
  h->commA.alloc(1);
  h->commB.alloc(3 * (size_t)G + 4);  // column sums, two limbs of the guarded ECs' shares, ELBO terms
  h->partAcc.alloc((size_t)std::max(nb, 1) * G);
  h->partC.alloc(1024);
  if (h->flavor == 0) {
    const uint32_t nb0 = (uint32_t)std::max(h->nblk, 1);
    // every EC a workgroup can see: its wavefronts take slices (and long ECs) round-robin, SliceStream's stride
    static_assert(kPassThreads / 64 <= 16 && kPassThreadsB / 64 <= 16, "guard_cap / guard_bits assume <= 16 wavefronts per workgroup");
    h->guard_cap = 64u * ((h->nslices + nb0 - 1) / nb0 + 16u) + (h->n_long + nb0 - 1) / nb0 + 16u;
    h->guard_words = (G + 31u) / 32u;
    h->guard_list.alloc((size_t)nb0 * h->guard_cap);
    h->guard_bits.alloc((size_t)nb0 * 16 * h->guard_words);
    h->guard_bits.zero(h->stream);
    h->guard_tail.alloc(2 * (size_t)G);
    h->guard_tail.zero(h->stream);
  }
  if (!h->trange.p) {  // dense flavour: no tables
    h->trange.alloc(2);
    h->trange.zero(h->stream);
  }
  h->guard_err.alloc(1);
  h->guard_err.zero(h->stream);
  h->guard_visits.alloc(1);
  h->guard_visits.zero(h->stream);
  h->sc.alloc(1);
  h->tr_bound.alloc(kMaxTrace);
  h->tr_newnorm.alloc(kMaxTrace);
  h->tr_beta.alloc(kMaxTrace);
  h->tr_reset.alloc(kMaxTrace);
  if (!h->sc_host) MSW_HIP(hipHostMalloc((void **)&h->sc_host, sizeof(Scalars)));
  if (!h->ev0) {
    MSW_HIP(hipEventCreate(&h->ev0));
    MSW_HIP(hipEventCreate(&h->ev1));
  }
}

// ---- launch helpers for the templated sweeps ----------------------------------------------
// The sweeps address their LDS image by absolute ds addresses taken from the records: the image
// must start at LDS address 0, i.e. the kernels must not own static __shared__ memory.
template <class K>
void prepare_sweep(K k, size_t lds, size_t &lds_set) {
  if (lds > lds_set) {  // raise the dynamic-LDS limit of this instantiation only when it grows
    hipFuncAttributes fa;
    MSW_HIP(hipFuncGetAttributes(&fa, (const void *)k));
    if (fa.sharedSizeBytes != 0) throw Fail("internal: sweep kernel owns static LDS");
    MSW_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_set = lds;
  }
}
template <int ENC, bool GL, bool TL>
void launch_passA_t(msw_core *h) {
  const size_t lds = pass_lds_bytes(GL ? 1 : 0, h->n_tab_lds, h->G, true, ENC == kEncIndex);
  // ML: some slices hold ECs over several lanes (sell.hpp slice classes) -- an instantiation of its own: the few
  // scalar operations and branches the classes cost per slice are 2-3 % of a sweep over short slices (cfg3, cfg5)
  const bool ml = h->cls.s0[kSliceClasses - 1] > 0;
  auto k = ml ? k_passA<ENC, GL, TL, true> : k_passA<ENC, GL, TL, false>;
  prepare_sweep(k, lds, h->lds_attr[0][(ml ? 40 : 0) + ENC * 4 + (GL ? 2 : 0) + (TL ? 1 : 0)]);
  hipLaunchKernelGGL(k, dim3(h->nblk), dim3(pass_threads_A<ENC>()), lds, h->stream, h->sc.p, sell_view(h),
                     h->ew.p, h->tabA.p, h->partA.p, h->partR.p, (int)((h->G + kRedfinGroups - 1) / kRedfinGroups),
                     h->guard_view());
}
template <int ENC, int GM, bool TL>
void launch_passB_t(msw_core *h) {
  const size_t lds = pass_lds_bytes(GM, h->n_tab_lds, h->G, false, ENC == kEncIndex);
  const bool ml = h->cls.s0[kSliceClasses - 1] > 0;
  if constexpr (ENC == kEncNarrow) {
    if (h->passB_rc8 && !ml) {  // (nearly) every slice at most 8 rows: 16 wavefronts per workgroup (finish_sell)
      auto k8 = k_passB<ENC, GM, TL, false, 8>;
      prepare_sweep(k8, lds, h->lds_attr[2][2 * GM + (TL ? 1 : 0)]);
      const auto rg_of = [&](uint32_t g0) {
        return GM == 4 ? RangeB{g0, std::min<uint32_t>(kRangeGroups, h->G - g0), g0 == 0 ? 1 : 0} : RangeB{0, 0, 1};
      };
      for (uint32_t g0 = 0; g0 < (GM == 4 ? h->G : 1u); g0 += kRangeGroups)
        hipLaunchKernelGGL(k8, dim3(h->nblk), dim3(pass_threads_B<ENC, 8>()), lds, h->stream, h->sc.p, sell_view(h), h->e.p,
                           h->tabB.p, h->partAcc.p, h->partS.p, h->Acc.p, rg_of(g0), h->guard_view());
      return;
    }
  }
  auto k = ml ? k_passB<ENC, GM, TL, true> : k_passB<ENC, GM, TL, false>;
  prepare_sweep(k, lds, h->lds_attr[1][(ml ? 40 : 0) + ENC * 10 + 2 * GM + (TL ? 1 : 0)]);
  if (GM == 4) {  // one run per range of groups; the first also delivers the ELBO terms
    for (uint32_t g0 = 0; g0 < h->G; g0 += kRangeGroups)
      hipLaunchKernelGGL(k, dim3(h->nblk), dim3(pass_threads_B<ENC>()), lds, h->stream, h->sc.p, sell_view(h), h->e.p,
                         h->tabB.p, h->partAcc.p, h->partS.p, h->Acc.p,
                         RangeB{g0, std::min<uint32_t>(kRangeGroups, h->G - g0), g0 == 0 ? 1 : 0}, h->guard_view());
    return;
  }
  hipLaunchKernelGGL(k, dim3(h->nblk), dim3(pass_threads_B<ENC>()), lds, h->stream, h->sc.p, sell_view(h), h->e.p,
                     h->tabB.p, h->partAcc.p, h->partS.p, h->Acc.p, RangeB{0, 0, 1}, h->guard_view());
}

#define MSW_DISPATCH3(fn, ...)                                                       \
  do {                                                                               \
    const int key = h->enc * 4 + (h->glds ? 2 : 0) + (h->tlds ? 1 : 0);              \
    switch (key) {                                                                   \
      case 0: fn<kEncNarrow, false, false>(__VA_ARGS__); break;                      \
      case 1: fn<kEncNarrow, false, true>(__VA_ARGS__); break;                       \
      case 2: fn<kEncNarrow, true, false>(__VA_ARGS__); break;                       \
      case 3: fn<kEncNarrow, true, true>(__VA_ARGS__); break;                        \
      case 4: fn<kEncWide, false, false>(__VA_ARGS__); break;                        \
      case 5: fn<kEncWide, false, true>(__VA_ARGS__); break;                         \
      case 6: fn<kEncWide, true, false>(__VA_ARGS__); break;                         \
      case 7: fn<kEncWide, true, true>(__VA_ARGS__); break;                          \
      case 8: fn<kEncIndex, false, false>(__VA_ARGS__); break;                       \
      case 10: fn<kEncIndex, true, false>(__VA_ARGS__); break;                       \
      case 12: fn<kEncValue, false, false>(__VA_ARGS__); break;                      \
      case 14: fn<kEncValue, true, false>(__VA_ARGS__); break;                       \
      default: throw Fail("internal: no sweep for this layout");                     \
    }                                                                                \
  } while (0)
#define MSW_DISPATCH_B(fn, ...)                                                      \
  do {                                                                               \
    const int key = h->enc * 10 + 2 * h->gmodeB + (h->tlds ? 1 : 0);                 \
    switch (key) {                                                                   \
      case 0: fn<kEncNarrow, 0, false>(__VA_ARGS__); break;                          \
      case 1: fn<kEncNarrow, 0, true>(__VA_ARGS__); break;                           \
      case 2: fn<kEncNarrow, 1, false>(__VA_ARGS__); break;                          \
      case 3: fn<kEncNarrow, 1, true>(__VA_ARGS__); break;                           \
      case 4: fn<kEncNarrow, 2, false>(__VA_ARGS__); break;                          \
      case 5: fn<kEncNarrow, 2, true>(__VA_ARGS__); break;                           \
      case 6: fn<kEncNarrow, 3, false>(__VA_ARGS__); break;                          \
      case 7: fn<kEncNarrow, 3, true>(__VA_ARGS__); break;                           \
      case 8: fn<kEncNarrow, 4, false>(__VA_ARGS__); break;                          \
      case 9: fn<kEncNarrow, 4, true>(__VA_ARGS__); break;                           \
      case 10: fn<kEncWide, 0, false>(__VA_ARGS__); break;                           \
      case 11: fn<kEncWide, 0, true>(__VA_ARGS__); break;                            \
      case 12: fn<kEncWide, 1, false>(__VA_ARGS__); break;                           \
      case 13: fn<kEncWide, 1, true>(__VA_ARGS__); break;                            \
      case 14: fn<kEncWide, 2, false>(__VA_ARGS__); break;                           \
      case 15: fn<kEncWide, 2, true>(__VA_ARGS__); break;                            \
      case 16: fn<kEncWide, 3, false>(__VA_ARGS__); break;                           \
      case 17: fn<kEncWide, 3, true>(__VA_ARGS__); break;                            \
      case 18: fn<kEncWide, 4, false>(__VA_ARGS__); break;                           \
      case 19: fn<kEncWide, 4, true>(__VA_ARGS__); break;                            \
      case 20: fn<kEncIndex, 0, false>(__VA_ARGS__); break;                          \
      case 22: fn<kEncIndex, 1, false>(__VA_ARGS__); break;                          \
      case 24: fn<kEncIndex, 2, false>(__VA_ARGS__); break;                          \
      case 26: fn<kEncIndex, 3, false>(__VA_ARGS__); break;                          \
      case 28: fn<kEncIndex, 4, false>(__VA_ARGS__); break;                          \
      case 30: fn<kEncValue, 0, false>(__VA_ARGS__); break;                          \
      case 32: fn<kEncValue, 1, false>(__VA_ARGS__); break;                          \
      case 34: fn<kEncValue, 2, false>(__VA_ARGS__); break;                          \
      case 36: fn<kEncValue, 3, false>(__VA_ARGS__); break;                          \
      case 38: fn<kEncValue, 4, false>(__VA_ARGS__); break;                          \
      default: throw Fail("internal: no sweep for this layout");                     \
    }                                                                                \
  } while (0)

template <int NREG>
void launch_dense_A(msw_core *h) {
  hipLaunchKernelGGL(k_dense_passA<NREG>, dim3(h->nblk_dense), dim3(256), 0, h->stream, h->sc.p,
                     h->Lt.p, (int)h->G, h->E, h->u.p, h->w.p, h->partA.p);
}
template <int NREG>
void launch_dense_B(msw_core *h) {
  const size_t lds = (32 + 4 * (size_t)h->G) * sizeof(double);
  hipLaunchKernelGGL(k_dense_passB<NREG>, dim3(h->nblk_dense), dim3(256), lds, h->stream, h->sc.p,
                     h->Lt.p, (int)h->G, h->E, h->cvec.p, h->u.p, h->partAcc.p, h->partS.p);
}
template <int NREG>
void launch_dense_big_A(msw_core *h) {
  hipLaunchKernelGGL(k_dense_big_passA<NREG>, dim3(h->nblk_dense), dim3(256), 0, h->stream, h->sc.p,
                     h->Lt.p, (int)h->G, h->E, h->u.p, h->w.p, h->partA.p);
}
template <int NREG>
void launch_dense_big_B(msw_core *h) {
  hipLaunchKernelGGL(k_dense_big_passB<NREG>, dim3(h->nblk_dense), dim3(256), 0, h->stream, h->sc.p,
                     h->Lt.p, (int)h->G, h->E, h->cvec.p, h->u.p, h->partAcc.p, h->partS.p);
}
#define MSW_DISPATCH_NREG(fn, fnbig, ...)                \
  do {                                                   \
    switch (h->nreg) {                                   \
      case 1: fn<1>(__VA_ARGS__); break;                 \
      case 2: fn<2>(__VA_ARGS__); break;                 \
      case 4: fn<4>(__VA_ARGS__); break;                 \
      case 8: fn<8>(__VA_ARGS__); break;                 \
      case 16: fn<16>(__VA_ARGS__); break;               \
      case 32: fnbig<32>(__VA_ARGS__); break;            \
      case 64: fnbig<64>(__VA_ARGS__); break;            \
      default: fnbig<128>(__VA_ARGS__); break;           \
    }                                                    \
  } while (0)

void launch_passA(msw_core *h) {
  std::pair<hipEvent_t, hipEvent_t> *ev = nullptr;
  if (h->profiling) {
    ev = &next_pair(h->evA, h->evA_used);
    MSW_HIP(hipEventRecord(ev->first, h->stream));
  }
  if (h->flavor == 0) MSW_DISPATCH3(launch_passA_t, h);
  else MSW_DISPATCH_NREG(launch_dense_A, launch_dense_big_A, h);
  MSW_HIP(hipGetLastError());
  if (ev) MSW_HIP(hipEventRecord(ev->second, h->stream));
  h->timing.passA_launches++;
}

// --emprecision float: the fp32 sweep of em_f32_kernels.hpp (run_em decides; one persistent workgroup per CU)
bool em_f32_layout_ok(const msw_core *h) {
  return kFx && h->flavor == 0 && h->enc == kEncNarrow && h->glds && h->tlds && !h->comm && h->n_tab_lds == h->n_area &&
         h->G <= (uint32_t)(kStepRegs * 1024) && em_f32_lds_bytes(h->n_tab_lds, h->G) <= kLdsMax &&
         !getenv("MSWEEP_EM_FLOAT_AS_DOUBLE");  // (developer switch: the fp64 kernels under MSW_PREC_FLOAT, as until round 4)
}
void launch_em_passB_f32(msw_core *h) {
  const size_t lds = em_f32_lds_bytes(h->n_tab_lds, h->G);
  const bool ml = h->cls.s0[kSliceClasses - 1] > 0;
  auto k = ml ? k_em_passB_f32<true> : k_em_passB_f32<false>;
  prepare_sweep(k, lds, h->lds_attr[2][30 + (ml ? 1 : 0)]);
  hipLaunchKernelGGL(k, dim3(h->nblk), dim3(1024), lds, h->stream, h->sc.p, sell_view(h), h->e.p, h->e32.p, h->tab32.p,
                     h->partAcc.p, h->partS.p, h->guard_view());
}

void launch_passB(msw_core *h) {
  std::pair<hipEvent_t, hipEvent_t> *ev = nullptr;
  if (h->profiling) {
    ev = &next_pair(h->evB, h->evB_used);
    MSW_HIP(hipEventRecord(ev->first, h->stream));
  }
  if (h->flavor == 0 && h->em_f32) {
    launch_em_passB_f32(h);
  } else if (h->flavor == 0) {
    if (h->gmodeB == 0) MSW_HIP(hipMemsetAsync(h->Acc.p, 0, ((size_t)h->G + kSentinels) * sizeof(double), h->stream));
    MSW_DISPATCH_B(launch_passB_t, h);
  } else {
    MSW_DISPATCH_NREG(launch_dense_B, launch_dense_big_B, h);
  }
  MSW_HIP(hipGetLastError());
  if (ev) MSW_HIP(hipEventRecord(ev->second, h->stream));
  h->timing.passB_launches++;
  // column sums across workgroups + N_g / lgamma / digamma, spread over G/16 workgroups
  const bool partials = (h->flavor == 1) || h->gmodeB > 0 || h->em_f32;
  const int nb = h->npart_rows();
  // the CSR sweeps leave fixed-point integer rows (kFx); 2: the fp32 EM sweep's, without the per-group factor
  const int fxrows = h->flavor == 0 ? (h->em_f32 ? 2 : 1) : 0;
  if (h->comm) {
    // EC-sharded: local column sums + ELBO terms -> one all-reduce -> k_redfin on the totals.  The
    // fixed-point column sums are all-reduced as INTEGERS: exact, so the totals -- and with them every
    // N_g -- are the same bits whatever the number of ranks the ECs are spread over.
    const size_t G3 = 3 * (size_t)h->G;
    std::pair<hipEvent_t, hipEvent_t> *evc = nullptr;
    if (h->profiling) {
      evc = &next_pair(h->evC, h->evC_used);
      MSW_HIP(hipEventRecord(evc->first, h->stream));
    }
    hipLaunchKernelGGL(k_colsum, dim3((h->G + 63) / 64), dim3(1024), 0, h->stream, h->sc.p, (int)h->G,
                       partials ? nb : 0, fxrows, nb, h->partAcc.p, h->Acc.p, h->partS.p,
                       fxrows ? h->guard_tail.p : nullptr, h->commB.p);
    if (kFx && fxrows) {
      h->comm->allreduce_mixed(reinterpret_cast<uint64_t *>(h->commB.p), G3, h->commB.p + G3, 4, h->stream);
    } else {  // fp64 column sums (dense flavour, MSW_FX=0 builds): doubles, then the integer limbs
      h->comm->allreduce(h->commB.p, (size_t)h->G, h->stream);
      h->comm->allreduce_mixed(reinterpret_cast<uint64_t *>(h->commB.p) + h->G, 2 * (size_t)h->G, h->commB.p + G3, 4,
                               h->stream);
    }
    if (evc) MSW_HIP(hipEventRecord(evc->second, h->stream));
    hipLaunchKernelGGL(k_redfin, dim3((h->G + kRedfinGroups - 1) / kRedfinGroups), dim3(kRedfinThreads), 0, h->stream,
                       h->sc.p, (int)h->G, 0, fxrows, reinterpret_cast<unsigned long long *>(h->commB.p) + h->G, 0, 1,
                       h->partAcc.p, h->commB.p, h->commB.p + G3, h->e.p, h->u.p,
                       h->alpha0.p, h->Nc.p, h->N.p, h->w.p, h->ew.p, h->partR.p, h->totS.p);
    return;
  }
  hipLaunchKernelGGL(k_redfin, dim3((h->G + kRedfinGroups - 1) / kRedfinGroups), dim3(kRedfinThreads), 0, h->stream,
                     h->sc.p, (int)h->G, partials ? nb : 0, fxrows, fxrows ? h->guard_tail.p : nullptr, 1, nb,
                     h->partAcc.p, h->Acc.p, h->partS.p, h->e.p,
                     h->u.p, h->alpha0.p, h->Nc.p, h->N.p, h->w.p, h->ew.p, h->partR.p, h->totS.p);
}

// partS as seen by k_fin / k_em_fin: the all-reduced totals when sharded
const double *fin_partS(msw_core *h) { return h->comm ? h->commB.p + 3 * (size_t)h->G : h->partS.p; }
int fin_npartS(msw_core *h) { return h->comm ? 1 : h->npart_rows(); }

// slot areas beyond kTabInline entries: the tables are rebuilt by their own kernel after every
// kernel that may have moved a (it returns at once when they are current)
void launch_tables(msw_core *h) {
  if (h->flavor != 0 || h->n_area <= (uint32_t)kTabInline) return;
  const unsigned nb = std::min<unsigned>((h->n_area + 255) / 256, (unsigned)h->n_cu * 8);
  hipLaunchKernelGGL(k_tables, dim3(nb), dim3(256), 0, h->stream, h->sc.p, (int)h->n_area, h->lut_area.p, h->tabs(),
                     h->tab_built.p);
}

// the verdict on the pending evaluation + the next step (state_kernels.hpp k_finstep); mode 1: the verdict alone
void launch_finstep(msw_core *h, int mode) {
  TraceDev tr{h->tr_bound.p, h->tr_newnorm.p, h->tr_beta.p, h->tr_theta.p, h->tr_reset.p};
  const double *pA = h->partA.p;
  int npA = h->flavor == 0 ? h->nblk : h->nblk_dense;
  if (h->comm) {  // |g|^2 summed over the EC shards (run_rcg)
    pA = h->commA.p;
    npA = 1;
  }
  hipLaunchKernelGGL(k_finstep, dim3(1), dim3(1024), 0, h->stream, h->sc.p, mode, (int)h->G, h->n_tab_inline(), npA, pA,
                     (int)((h->G + kRedfinGroups - 1) / kRedfinGroups), h->totS.p, h->partR.p, h->Nc.p, h->w.p, h->u.p,
                     h->os_u.p, h->step_u.p, h->lut_area.p, h->e.p, h->tabs(), tr);
  launch_tables(h);
}

void poll(msw_core *h) {
  MSW_HIP(hipMemcpyAsync(h->sc_host, h->sc.p, sizeof(Scalars), hipMemcpyDeviceToHost, h->stream));
  MSW_HIP(hipStreamSynchronize(h->stream));
  if (h->comm) h->comm->check();
}

// Inputs of one solve: c_j (from log counts or from bootstrap counts already on the device) and
// the prior.  Leaves per-block partial sums of c in partC for k_init_state.
constexpr int kCvecBlocks = 512;
void prepare_inputs(msw_core *h, const double *logc_host, const uint32_t *counts_dev,
                    const double *alpha0_host) {
  if (h->flavor < 0) throw Fail("no likelihood resident: call msw_core_set_csr / set_dense_logl first");
  const uint32_t E = h->E, G = h->G;
  if (counts_dev) {
    hipLaunchKernelGGL(k_cvec_from_counts, dim3(kCvecBlocks), dim3(256), 0, h->stream, counts_dev,
                       h->flavor == 0 ? h->perm.p : nullptr, E, h->cvec.p, h->c8.p, h->partC.p, h->cls, h->n_long,
                       h->flavor == 0 ? h->c8s.p : nullptr);
  } else {
    const double *src = h->logc_d.p;
    if (logc_host) {
      MSW_HIP(hipMemcpyAsync(h->logc_d.p, logc_host, E * sizeof(double), hipMemcpyHostToDevice, h->stream));
    } else {  // the log counts msw_core_build_likelihood left on the device: no 8 * E byte upload
      if (!h->have_logc_res) throw Fail("null logc: only a likelihood built by msw_core_build_likelihood keeps its log counts");
      src = h->logc_res.p;
    }
    hipLaunchKernelGGL(k_cvec_from_logc, dim3(kCvecBlocks), dim3(256), 0, h->stream, src,
                       h->flavor == 0 ? h->perm.p : nullptr, E, h->cvec.p, h->c8.p, h->partC.p, h->cls, h->n_long,
                       h->flavor == 0 ? h->c8s.p : nullptr);
  }
  if (alpha0_host)
    MSW_HIP(hipMemcpyAsync(h->alpha0.p, alpha0_host, G * sizeof(double), hipMemcpyHostToDevice, h->stream));
  MSW_HIP(hipGetLastError());
  // logc_host / alpha0_host may be pageable caller memory: finish the copies before returning
  MSW_HIP(hipStreamSynchronize(h->stream));
  h->prepared = true;
}

// argument / state errors of a solve: thrown BEFORE anything that a peer of a sharded solve could wait for
void validate_solve(const msw_core *, size_t max_iters, int algo, int prec) {
  if (algo != MSW_ALGO_RCG && algo != MSW_ALGO_EM) throw Fail("unknown algorithm id");
  if (prec != MSW_PREC_DOUBLE && prec != MSW_PREC_FLOAT) throw Fail("unknown precision id");
  if (max_iters == 0 || max_iters > (size_t)std::numeric_limits<int32_t>::max())
    throw Fail("max_iters out of range");
}

void begin_solve(msw_core *h, double tol, size_t max_iters) {
  const uint32_t G = h->G;
  if (h->trace_theta) h->tr_theta.alloc(h->trace_theta * G);
  const double *cpart = h->partC.p;
  int ncpart = kCvecBlocks;
  if (h->comm) {  // global sum of the EC counts (bound constant, theta normalisation)
    h->comm->rendezvous();  // the ranks' calls may be seconds apart on the host: meet before the device-side waits
    hipLaunchKernelGGL(k_sum_scalar, dim3(1), dim3(1024), 0, h->stream, h->sc.p, 0, kCvecBlocks, h->partC.p,
                       h->commA.p);
    h->comm->allreduce(h->commA.p, 1, h->stream);
    cpart = h->commA.p;
    ncpart = 1;
  }
  hipLaunchKernelGGL(k_init_state, dim3(1), dim3(1024), 0, h->stream, h->sc.p, (int)G, ncpart,
                     cpart, h->alpha0.p, h->u.p, h->os_u.p, h->step_u.p, tol, (int)max_iters,
                     h->fixed_iters ? 1 : 0, (int)h->trace_theta, h->flavor, h->logzi, h->opts, h->tab_built.p,
                     h->trange.p);
  MSW_HIP(hipGetLastError());
}

// iters_start > 0: the solve on the handle is continued (msw_core_continue) -- no initial evaluation
void run_rcg(msw_core *h, size_t max_iters, size_t iters_start = 0) {
  const int G = (int)h->G, n_lut = h->n_tab_inline();
  if (iters_start > 0 && h->comm) h->comm->rendezvous();  // msw_core_continue: as begin_solve
  if (iters_start == 0) {
    // initial update_N_k on gamma = log(1/G): the first slot's k_finstep finds it as Scalars::have_eval = 2
    hipLaunchKernelGGL(k_prepB, dim3(1), dim3(1024), 0, h->stream, h->sc.p, G, n_lut, h->u.p, h->lut_area.p,
                       h->e.p, h->tabs());
    launch_tables(h);
    launch_passB(h);
    h->timing.passB_launches--;  // the initial evaluation is not an iteration
    if (h->profiling && h->evB_used) h->evB_used--;
  }
  const int nbA = h->flavor == 0 ? h->nblk : h->nblk_dense;
  // Slots (state_kernels.hpp k_finstep): k_passA -> k_finstep -> k_passB (+ k_redfin); one per iteration plus one per
  // rejected step.  The verdict on a slot's evaluation is taken by the NEXT slot's k_finstep, so the iteration count
  // the host polls lags the evaluations by one: enqueue as many slots as iterations are still missing, poll, repeat;
  // slots enqueued past the end return at once (Scalars::done), and a verdict-only launch closes the run where the
  // count, not the tolerance, ends it.
  size_t iters_done = iters_start;
  for (;;) {
    // fixed-iteration runs know how many slots are missing; otherwise poll every kIterBatch
    const size_t missing = std::max<size_t>(1, max_iters - std::min(max_iters, iters_done));
    const size_t batch = h->fixed_iters ? std::min<size_t>(256, missing) : std::min<size_t>(kIterBatch, missing);
    for (size_t b = 0; b < batch; ++b) {
      launch_passA(h);
      if (h->comm) {  // |g|^2 summed over the EC shards
        std::pair<hipEvent_t, hipEvent_t> *evc = nullptr;
        if (h->profiling) {
          evc = &next_pair(h->evC, h->evC_used);
          MSW_HIP(hipEventRecord(evc->first, h->stream));
        }
        hipLaunchKernelGGL(k_sum_scalar, dim3(1), dim3(1024), 0, h->stream, h->sc.p, 1, nbA, h->partA.p,
                           h->commA.p);
        h->comm->allreduce(h->commA.p, 1, h->stream);
        if (evc) MSW_HIP(hipEventRecord(evc->second, h->stream));
      }
      launch_finstep(h, 0);
      launch_passB(h);
    }
    MSW_HIP(hipGetLastError());
    poll(h);
    if (h->sc_host->done) break;
    if (h->sc_host->have_eval && (size_t)h->sc_host->iter + 1 >= max_iters) {  // the last evaluation's verdict ends the run
      launch_finstep(h, 1);
      MSW_HIP(hipGetLastError());
      poll(h);
      if (h->sc_host->done) break;
    }
    iters_done = (size_t)h->sc_host->iter;
  }
}

void finish_solve(msw_core *h, double *theta_out, size_t *iters_out, double *bound_out) {
  poll(h);
  const uint32_t G = h->G;
  int gerr = 0;
  MSW_HIP(hipMemcpy(&gerr, h->guard_err.p, sizeof gerr, hipMemcpyDeviceToHost));
  if (gerr) {
    MSW_HIP(hipMemset(h->guard_err.p, 0, sizeof gerr));
    if (gerr == 2) throw Fail("internal: a workgroup's list of guarded equivalence classes overflowed");
    throw NumericFail("likelihood underflow: an equivalence class has zero probability under every group "
                      "(exp(a * log-likelihood) and the group weights underflow fp64 together)");
  }
  // (an EM run that was asked for no iteration leaves its initial bound, -inf: nothing is wrong)
  if (h->sc_host->iter > 0 && !std::isfinite(h->sc_host->bound))
    throw NumericFail("the evidence lower bound is not finite: the likelihood or the prior counts are out of range");
  if (theta_out) {
    if (h->last_algo == MSW_ALGO_EM) {
      MSW_HIP(hipMemcpy(theta_out, h->logth.p, G * sizeof(double), hipMemcpyDeviceToHost));  // theta of the last M-step
    } else {
      std::vector<double> nc(G);
      MSW_HIP(hipMemcpy(nc.data(), h->Nc.p, G * sizeof(double), hipMemcpyDeviceToHost));
      const double csum = h->sc_host->csum;
      for (uint32_t g = 0; g < G; ++g) theta_out[g] = nc[g] / csum;
    }
  }
  if (iters_out) *iters_out = (size_t)h->sc_host->iter;
  if (bound_out) *bound_out = h->sc_host->bound;
  h->have_solution = true;
}

void collect_timing(msw_core *h) {
  float ms = 0.f;
  MSW_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  h->timing.solve_ms = ms;
  h->timing.passA_ms = h->timing.passB_ms = 0.0;
  if (h->profiling) {
    // launches enqueued after `done` was set return immediately; they are still counted
    for (size_t i = 0; i < h->evA_used; ++i) {
      MSW_HIP(hipEventElapsedTime(&ms, h->evA[i].first, h->evA[i].second));
      h->timing.passA_ms += ms;
    }
    for (size_t i = 0; i < h->evB_used; ++i) {
      MSW_HIP(hipEventElapsedTime(&ms, h->evB[i].first, h->evB[i].second));
      h->timing.passB_ms += ms;
    }
    h->timing.collective_ms = 0.0;
    for (size_t i = 0; i < h->evC_used; ++i) {
      MSW_HIP(hipEventElapsedTime(&ms, h->evC[i].first, h->evC[i].second));
      h->timing.collective_ms += ms;
    }
    h->timing.collectives = h->evC_used;
  }
  h->timing.iters = (uint64_t)h->sc_host->iter;
  const uint64_t recsz = h->enc == kEncValue ? 12 : (h->wide() ? 8 : 4);
  if (h->flavor == 0) {
    // algorithmic bytes: every real cell record once + the per-EC count vector in
    // pass B; SELL padding, slice offsets and the L2-served second read of pass B are not counted
    // (slot tables that do not fit LDS are read from memory: every used 16-byte entry at least once per sweep)
    // (a hybrid area: the entries beyond its LDS-resident head)
    const uint64_t tab = h->tlds ? 0ull : 16ull * (h->n_area - h->n_tab_lds);
    h->timing.bytes_passA = h->nnz * recsz + tab;
    h->timing.bytes_passB = h->nnz * recsz + 1ull * h->E + tab;  // + one byte per EC (its multiplicity)
  } else {
    h->timing.bytes_passA = 8ull * h->E * h->G;
    h->timing.bytes_passB = 8ull * h->E * h->G + 8ull * h->E;
  }
}

void run_em(msw_core *h, size_t max_iters, int prec);

// n more iterations of the fixed-iteration RCG solve that last ran on the handle: the state carries on
// where it stood (no re-initialisation, no initial evaluation) -- what a benchmark's "W warm-up steps,
// then exactly K timed steps" means for an iterative solver.
void continue_impl(msw_core *h, size_t n_iters, double *theta_out, size_t *iters_out, double *bound_out) {
  if (!h->have_solution || !h->fixed_iters || h->last_algo != MSW_ALGO_RCG)
    throw Fail("msw_core_continue: needs a fixed-iteration RCG solve on the handle (msw_core_set_fixed_iters, msw_core_run)");
  const size_t start = (size_t)h->sc_host->iter;
  if (n_iters == 0 || start + n_iters > (size_t)std::numeric_limits<int32_t>::max()) throw Fail("msw_core_continue: iteration count out of range");
  h->timing = {};
  h->evA_used = h->evB_used = h->evC_used = 0;
  CollectiveScope cs(h);
  hipLaunchKernelGGL(k_extend, dim3(1), dim3(1), 0, h->stream, h->sc.p, (int)n_iters);
  MSW_HIP(hipEventRecord(h->ev0, h->stream));
  run_rcg(h, start + n_iters, start);
  MSW_HIP(hipEventRecord(h->ev1, h->stream));
  finish_solve(h, theta_out, iters_out, bound_out);
  collect_timing(h);
  h->timing.iters = (uint64_t)h->sc_host->iter - start;
  cs.leave();
}

void run_impl(msw_core *h, double tol, size_t max_iters, int algo, int prec, double *theta_out,
              size_t *iters_out, double *bound_out) {
  validate_solve(h, max_iters, algo, prec);
  if (!h->prepared) throw Fail("msw_core_run: inputs not prepared (call msw_core_prepare)");
  h->timing = {};
  h->evA_used = h->evB_used = h->evC_used = 0;
  CollectiveScope cs(h);  // from here on a failure strands the peers of a sharded solve (guarded())
  begin_solve(h, tol, max_iters);
  MSW_HIP(hipEventRecord(h->ev0, h->stream));
  if (algo == MSW_ALGO_RCG) run_rcg(h, max_iters);
  else run_em(h, max_iters, prec);
  MSW_HIP(hipEventRecord(h->ev1, h->stream));
  h->last_algo = algo;
  finish_solve(h, theta_out, iters_out, bound_out);
  collect_timing(h);
  cs.leave();
}

}  // namespace

namespace {
// MSWEEP_BUILD_TIMING=1 (developer switch): wall time of every stage of the build / the packer to stderr (each
// mark drains the stream first, so the stages are what they say; off: no synchronisation, no output)
struct StageTimer {
  hipStream_t st;
  bool on;
  std::chrono::steady_clock::time_point t0;
  explicit StageTimer(hipStream_t s) : st(s), on(getenv("MSWEEP_BUILD_TIMING") != nullptr), t0(std::chrono::steady_clock::now()) {}
  void mark(const char *what) {
    if (!on) return;
    (void)hipStreamSynchronize(st);
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[msweep build] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};
}  // namespace

#include "host_likelihood.inc"
#include "host_pack.inc"
#include "host_em.inc"
#include "host_mtjump.inc"
#include "host_bootstrap.inc"
#include "host_build.inc"
#include "host_compress.inc"
#include "host_alignment.inc"
#include "host_reader.inc"

// =========================================================================================
// C ABI
// =========================================================================================
extern "C" {

#ifndef MSW_SRC_HASH
#define MSW_SRC_HASH "unhashed"
#endif
// "... src <hash>": sha256 prefix of the sources this library was compiled from (__graft_entry__.source_hash)
const char *msw_core_version(void) { return "msweep_core 0.2 gfx950 (HIP, wave64) src " MSW_SRC_HASH; }

const char *msw_last_error(msw_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int msw_core_create(int device, msw_handle *out) {
  if (!out) return 1;
  *out = nullptr;
  try {
    int n = 0;
    MSW_HIP(hipGetDeviceCount(&n));
    if (n <= 0) throw Fail("no HIP device visible: libmsweep_core has no CPU fallback");
    if (device < 0 || device >= n) throw Fail("device index out of range");
    MSW_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    MSW_HIP(hipGetDeviceProperties(&prop, device));
    std::unique_ptr<msw_core> h(new msw_core);
    h->device = device;
    h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    MSW_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    hipLaunchKernelGGL(k_warm, dim3(1), dim3(1), 0, h->stream, (int *)nullptr);  // code object load: here, once per process
    MSW_HIP(hipGetLastError());
    MSW_HIP(hipStreamSynchronize(h->stream));
    *out = h.release();
    return 0;
  } catch (const std::exception &ex) {
    g_create_error = ex.what();
    (void)hipGetLastError();
    return 1;
  }
}

void msw_core_destroy(msw_handle h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  delete h;
}

int msw_core_shape(msw_handle h, size_t *n_groups, size_t *n_ecs, size_t *nnz) {
  return guarded(h, [&] {
    if (h->flavor < 0) throw Fail("no likelihood resident");
    if (n_groups) *n_groups = h->G;
    if (n_ecs) *n_ecs = h->E;
    if (nnz) *nnz = h->flavor == 0 ? h->nnz : (size_t)h->G * h->E;
  });
}

int msw_core_set_dense_logl(msw_handle h, const double *L, size_t n_groups, size_t n_ecs, size_t ld) {
  return guarded(h, [&] { set_dense_impl(h, L, n_groups, n_ecs, ld); });
}

int msw_core_set_csr(msw_handle h, const uint64_t *rowptr, const uint32_t *grp, const uint32_t *cnt,
                     const double *lut, size_t lut_ld, double logzi, size_t n_groups, size_t n_ecs) {
  return guarded(h, [&] { set_csr_impl(h, rowptr, grp, cnt, lut, lut_ld, logzi, n_groups, n_ecs); });
}

int msw_core_build_likelihood(msw_handle h, const uint64_t *ec_tptr, const uint32_t *ec_targets,
                              size_t n_ecs, const uint32_t *target_group, size_t n_targets,
                              const uint64_t *group_sizes, size_t n_groups, const uint64_t *ec_counts,
                              double q, double e, double zero_inflation, size_t min_hits,
                              size_t *n_groups_out, uint8_t *mask_out, double *logc_out) {
  return guarded(h, [&] {
    build_likelihood_impl(h, ec_tptr, ec_targets, n_ecs, target_group, n_targets, group_sizes,
                          n_groups, ec_counts, q, e, zero_inflation, min_hits, n_groups_out, mask_out,
                          logc_out, false, 0);
  });
}

int msw_core_build_likelihood_aln(msw_handle h, msw_alignment_t a, const uint32_t *target_group, size_t n_targets,
                                  const uint64_t *group_sizes, size_t n_groups, double q, double e,
                                  double zero_inflation, size_t min_hits, size_t *n_groups_out, uint8_t *mask_out,
                                  double *logc_out) {
  return guarded(h, [&] {
    if (!a) throw Fail("msw_core_build_likelihood_aln: null alignment");
    if (a->on_device && a->device == h->device) {
      // the reader's arrays are consumed where they lie (same device; the reader ran on this handle's stream or
      // has synchronised its own)
      build_likelihood_impl(h, a->d_tptr.p, a->d_targets.p, a->E, target_group, n_targets, group_sizes, n_groups,
                            a->d_counts.p, q, e, zero_inflation, min_hits, n_groups_out, mask_out, logc_out, true, a->H);
    } else {
      a->to_host(msw_alignment::kAlnTptr | msw_alignment::kAlnTargets | msw_alignment::kAlnCounts);
      build_likelihood_impl(h, a->ec_tptr.data(), a->ec_targets.data(), a->ec_counts.size(), target_group, n_targets,
                            group_sizes, n_groups, a->ec_counts.data(), q, e, zero_inflation, min_hits, n_groups_out,
                            mask_out, logc_out, false, 0);
    }
  });
}

int msw_core_trim(msw_handle h) {
  return guarded(h, [&] {
    MSW_HIP(hipStreamSynchronize(h->stream));
    h->reader_pool.trim();
  });
}

int msw_alignment_read_device(msw_handle h, const char *const *paths, size_t n_paths, size_t n_targets, int merge_mode,
                              msw_alignment_t *out) {
  if (out) *out = nullptr;
  const int rc = guarded(h, [&] {
    check_reader_args(paths, out, n_paths, n_targets, merge_mode);
    std::unique_ptr<msw_alignment> a(new msw_alignment);
    a->device = h->device;
    MSW_HIP(hipStreamSynchronize(h->stream));
    h->reader_pool.recycle();  // (blocks the previous read handed back: nothing of it is in flight any more)
    ReaderCtx cx(h->stream, h->n_cu, &h->text_stage, &h->reader_pool, h->device);
    bool fits = true;
    {  // ~5 bytes of device memory per byte of text (measured: 10 GB at cfg3's 2.09 GB, 45 GB at 9.5 GB): a text that
       // would not fit beside what the device already holds goes to the host reader
      uint64_t text = 0;
      for (size_t i = 0; i < n_paths; ++i) {
        struct stat sb;
        if (paths[i] && stat(paths[i], &sb) == 0) text += (uint64_t)sb.st_size * (path_is_gzip(paths[i]) ? 8u : 1u);
      }
      size_t free_b = 0, total_b = 0;
      MSW_HIP(hipMemGetInfo(&free_b, &total_b));
      // (what earlier reads left idle on the handle counts as room; when it would be needed as FRESH memory -- a text of
      // another size class -- it goes back to the device first)
      const uint64_t need = 6 * text + (1ull << 30);
      if (need > (uint64_t)free_b && h->reader_pool.idle_bytes()) {
        h->reader_pool.trim();
        MSW_HIP(hipMemGetInfo(&free_b, &total_b));
      }
      fits = need <= (uint64_t)free_b + h->reader_pool.idle_bytes();
      if (getenv("MSWEEP_READER_FORCE_HOST")) fits = false;  // developer switch (tests): the host reader behind this entry
    }
    try {
      if (!fits) throw ReaderFallback{};
      try {
        read_alignment_device(paths, n_paths, n_targets, merge_mode, cx, *a);
      } catch (const HipError &ex) {
        if (!strstr(ex.what(), "out of memory")) throw;
        (void)hipGetLastError();
        throw ReaderFallback{};
      }
    } catch (const ReaderFallback &) {
      // text the kernels do not judge: the host reader's outcome -- a result or the reference's message -- stands
      msw_alignment_t host = nullptr;
      if (msw_alignment_read(paths, n_paths, n_targets, merge_mode, &host) != 0) throw Fail(msw_alignment_last_error());
      *out = host;
      return;
    }
    *out = a.release();
  });
  if (rc != 0 && h) g_aln_error = msw_last_error(h);
  return rc;
}

int msw_core_layout_info(msw_handle h, msw_layout_info *out) {
  return guarded(h, [&] {
    if (!out) throw Fail("null out");
    if (h->flavor != 0) throw Fail("msw_core_layout_info: no CSR-of-ECs likelihood resident");
    msw_layout_info li = {};
    li.record_bytes = h->enc == kEncValue ? 12 : (h->wide() ? 8 : 4);
    li.index_records = h->hybrid() ? 1 : 0;
    li.groups_in_lds = h->glds ? 1 : 0;
    li.table_in_lds = h->tlds ? 1 : 0;
    li.passB_mode = h->gmodeB;
    li.slot_entries = h->n_area;
    li.slot_entries_in_lds = h->n_tab_lds;
    li.n_slices = h->nslices;
    li.n_long_ecs = h->n_long;
    li.bank_scheduled = h->packed_scheduled ? 1 : 0;
    li.passB_reg_cells = h->passB_rc8 ? 8 : kRegCells;
    li.rows_over_8 = h->rows_over8;
    std::vector<uint32_t> off((size_t)h->nslices + 1);
    MSW_HIP(hipMemcpy(off.data(), h->slice_off.p, off.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    li.rows = off[h->nslices];
    for (int c = 0; c < kSliceClasses; ++c) li.slices_by_lanes[c] = h->cls.s0[c + 1] - h->cls.s0[c];
    for (uint32_t s2 = 0; s2 < h->nslices; ++s2) li.max_rows = std::max(li.max_rows, off[s2 + 1] - off[s2]);
    if (h->hybrid()) {
      std::vector<uint8_t> hot(std::max<uint32_t>(h->nslices, 1));
      MSW_HIP(hipMemcpy(hot.data(), h->slice_hot.p, hot.size(), hipMemcpyDeviceToHost));
      for (uint32_t s2 = 0; s2 < h->nslices; ++s2) li.rows_from_memory += (off[s2 + 1] - off[s2]) - hot[s2];
    } else if (!h->tlds && h->enc != kEncValue) {
      li.rows_from_memory = li.rows;
    }
    *out = li;
  });
}

int msw_core_layout_hash(msw_handle h, uint64_t *hash_out) {
  return guarded(h, [&] {
    if (!hash_out) throw Fail("null out");
    *hash_out = layout_hash(h);
  });
}

int msw_core_get_dense_logl(msw_handle h, double *L_out, size_t ld) {
  return guarded(h, [&] { materialise_impl(h, L_out, ld, /*gamma=*/false, 0, h->E); });
}

int msw_core_gamma(msw_handle h, double *gamma_out, size_t ld) {
  return guarded(h, [&] {
    if (!h->have_solution) throw Fail("msw_core_gamma: no solve has run on this handle");
    materialise_impl(h, gamma_out, ld, /*gamma=*/true, 0, h->E);
  });
}

int msw_core_gamma_block(msw_handle h, size_t ec_begin, size_t ec_end, double *gamma_out, size_t ld) {
  return guarded(h, [&] {
    if (!h->have_solution) throw Fail("msw_core_gamma_block: no solve has run on this handle");
    materialise_impl(h, gamma_out, ld, /*gamma=*/true, ec_begin, ec_end);
  });
}

int msw_core_solve(msw_handle h, const double *logc, const double *alpha0, double tol, size_t max_iters,
                   int algo, int prec, double *theta_out, size_t *iters_out, double *bound_out) {
  return guarded(h, [&] {
    if (!alpha0) throw Fail("msw_core_solve: null alpha0");
    prepare_inputs(h, logc, nullptr, alpha0);
    run_impl(h, tol, max_iters, algo, prec, theta_out, iters_out, bound_out);
  });
}

int msw_core_prepare(msw_handle h, const double *logc, const double *alpha0) {
  return guarded(h, [&] {
    if (!alpha0) throw Fail("msw_core_prepare: null alpha0");
    prepare_inputs(h, logc, nullptr, alpha0);
  });
}

int msw_core_run(msw_handle h, double tol, size_t max_iters, int algo, int prec, double *theta_out,
                 size_t *iters_out, double *bound_out) {
  return guarded(h, [&] { run_impl(h, tol, max_iters, algo, prec, theta_out, iters_out, bound_out); });
}

int msw_core_continue(msw_handle h, size_t n_iters, double *theta_out, size_t *iters_out, double *bound_out) {
  return guarded(h, [&] { continue_impl(h, n_iters, theta_out, iters_out, bound_out); });
}

int msw_core_set_trace_theta(msw_handle h, size_t n_iters) {
  return guarded(h, [&] {
    if (n_iters > (size_t)kMaxTrace) throw Fail("trace_theta: at most 4096 iterations");
    h->trace_theta = n_iters;
  });
}

int msw_core_trace(msw_handle h, size_t n, double *bound, double *newnorm, double *beta,
                   int32_t *didreset, double *theta_trace, size_t *n_out) {
  return guarded(h, [&] {
    if (!h->have_solution) throw Fail("msw_core_trace: no solve has run on this handle");
    size_t have = std::min<size_t>((size_t)h->sc_host->iter, kMaxTrace);
    n = std::min(n, have);
    if (bound) MSW_HIP(hipMemcpy(bound, h->tr_bound.p, n * sizeof(double), hipMemcpyDeviceToHost));
    if (newnorm) MSW_HIP(hipMemcpy(newnorm, h->tr_newnorm.p, n * sizeof(double), hipMemcpyDeviceToHost));
    if (beta) MSW_HIP(hipMemcpy(beta, h->tr_beta.p, n * sizeof(double), hipMemcpyDeviceToHost));
    if (didreset) MSW_HIP(hipMemcpy(didreset, h->tr_reset.p, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (theta_trace) {
      const size_t nt = std::min(n, h->trace_theta);
      if (nt) MSW_HIP(hipMemcpy(theta_trace, h->tr_theta.p, nt * h->G * sizeof(double), hipMemcpyDeviceToHost));
    }
    if (n_out) *n_out = n;
  });
}

int msw_core_bootstrap(msw_handle h, const uint32_t *ec_counts, int32_t seed, size_t bootstrap_count,
                       size_t rep_begin, size_t rep_end, const double *alpha0, double tol,
                       size_t max_iters, int algo, int prec, double *theta_out, size_t *iters_out) {
  return guarded(h, [&] {
    bootstrap_impl(h, ec_counts, seed, bootstrap_count, rep_begin, rep_end, alpha0, tol, max_iters,
                   algo, prec, theta_out, iters_out);
  });
}

int msw_core_resample_counts(msw_handle h, const uint32_t *ec_counts, size_t n_ecs, int32_t seed,
                             size_t bootstrap_count, size_t rep_begin, size_t rep_end,
                             uint32_t *counts_out) {
  return guarded(h, [&] {
    resample_impl(h, ec_counts, n_ecs, seed, bootstrap_count, rep_begin, rep_end, counts_out);
  });
}

int msw_core_bootstrap_dist(msw_handle h, msw_comm_t comm, const uint32_t *ec_counts, int32_t seed,
                            size_t bootstrap_count, size_t n_replicates, const double *alpha0, double tol,
                            size_t max_iters, int algo, int prec, double *theta_out, size_t *iters_out) {
  // (a rank whose block fails still joins the all-gather and reports through its status word: every rank
  // returns non-zero, none is left waiting, and the communicator stays usable -- host_bootstrap.inc)
  return guarded(h, [&] {
    bootstrap_dist_impl(h, comm, ec_counts, seed, bootstrap_count, n_replicates, alpha0, tol, max_iters, algo,
                        prec, theta_out, iters_out);
  });
}

const char *msw_comm_last_error(void) { return g_create_error.c_str(); }

int msw_comm_size(msw_comm_t c, int *nranks, int *rank) {
  if (!c) {
    g_create_error = "msw_comm_size: null communicator";
    return 1;
  }
  if (nranks) *nranks = c->size();
  if (rank) *rank = c->rank();
  return 0;
}

int msw_comm_rccl_count(msw_comm_t c, int *count) {
  if (!c || !count) {
    g_create_error = "msw_comm_rccl_count: null argument";
    return 1;
  }
  if (const PeerComm *pc = dynamic_cast<const PeerComm *>(c)) c = pc->base.get();
  const RcclComm *rc = dynamic_cast<const RcclComm *>(c);
  *count = rc ? rc->count() : 0;
  return 0;
}

int msw_comm_allgather(msw_comm_t c, const double *send, size_t n, double *recv) {
  if (!c || (n && (!send || !recv))) {
    g_create_error = "msw_comm_allgather: null argument";
    return 1;
  }
  try {
    c->allgather_host(send, n, recv);
    return 0;
  } catch (const std::exception &ex) {
    g_create_error = ex.what();
    (void)hipGetLastError();
    c->abort();
    return 1;
  }
}

int msw_comm_allreduce(msw_comm_t c, uint64_t *ints, size_t ni, double *reals, size_t nr, int repeats, double *ms_per_call) {
  if (!c || (ni && !ints) || (nr && !reals) || repeats < 1) {
    g_create_error = "msw_comm_allreduce: bad arguments";
    return 1;
  }
  hipStream_t st = nullptr;
  try {
    MSW_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    DevBuf<double> buf, work;
    buf.alloc(ni + nr + 1);
    work.alloc(ni + nr + 1);
    if (ni) MSW_HIP(hipMemcpyAsync(buf.p, ints, ni * 8, hipMemcpyHostToDevice, st));
    if (nr) MSW_HIP(hipMemcpyAsync(buf.p + ni, reals, nr * 8, hipMemcpyHostToDevice, st));
    // repeats > 1: the same message again and again (timing; the sums of the last one are returned)
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < repeats; ++k) {
      MSW_HIP(hipMemcpyAsync(work.p, buf.p, (ni + nr) * 8, hipMemcpyDeviceToDevice, st));
      c->allreduce_mixed(reinterpret_cast<uint64_t *>(work.p), ni, work.p + ni, nr, st);
    }
    MSW_HIP(hipStreamSynchronize(st));
    c->check();
    if (ms_per_call)
      *ms_per_call = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / repeats;
    if (ni) MSW_HIP(hipMemcpyAsync(ints, work.p, ni * 8, hipMemcpyDeviceToHost, st));
    if (nr) MSW_HIP(hipMemcpyAsync(reals, work.p + ni, nr * 8, hipMemcpyDeviceToHost, st));
    MSW_HIP(hipStreamSynchronize(st));
    (void)hipStreamDestroy(st);
    return 0;
  } catch (const std::exception &ex) {
    g_create_error = ex.what();
    (void)hipGetLastError();
    if (st) (void)hipStreamDestroy(st);
    c->abort();
    return 1;
  }
}

int msw_comm_unique_id(unsigned char id_out[128]) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  const ncclResult_t rc = ncclGetUniqueId(&id);
  if (rc != ncclSuccess) {
    g_create_error = std::string("ncclGetUniqueId: ") + ncclGetErrorString(rc);
    return 1;
  }
  std::memcpy(id_out, &id, 128);
  return 0;
}

int msw_comm_create_rccl(const unsigned char id[128], int rank, int nranks, int device, msw_comm_t *out) {
  if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks) {
    g_create_error = "msw_comm_create_rccl: bad arguments";
    return 1;
  }
  try {
    MSW_HIP(hipSetDevice(device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, 128);
    std::unique_ptr<msw_comm> c(new RcclComm(uid, rank, nranks));
    if (peer_allreduce_requested() && nranks > 1) c.reset(new PeerComm(std::move(c), /*ipc=*/true));
    *out = c.release();
    return 0;
  } catch (const std::exception &ex) {
    g_create_error = ex.what();
    return 1;
  }
}

int msw_comm_create_local(int nranks, msw_comm_t *out) {
  if (!out || nranks < 1) {
    g_create_error = "msw_comm_create_local: bad arguments";
    return 1;
  }
  try {
    const bool peer = peer_allreduce_requested() && nranks > 1;
    auto grp = std::make_shared<LocalGroup>(nranks);
    for (int r = 0; r < nranks; ++r) out[r] = nullptr;
    for (int r = 0; r < nranks; ++r) {
      std::unique_ptr<msw_comm> c(new LocalComm(grp, r));
      if (peer) c.reset(new PeerComm(std::move(c), /*ipc=*/false));
      out[r] = c.release();
    }
    return 0;
  } catch (const std::exception &ex) {
    for (int r = 0; r < nranks; ++r) delete out[r];
    g_create_error = ex.what();
    return 1;
  }
}

int msw_comm_create_shm(const char *name, int rank, int nranks, int device, msw_comm_t *out) {
  if (!out || !name || nranks < 1 || rank < 0 || rank >= nranks) {
    g_create_error = "msw_comm_create_shm: bad arguments";
    return 1;
  }
  try {
    MSW_HIP(hipSetDevice(device));
    std::unique_ptr<msw_comm> c(new ShmComm(name, rank, nranks));
    if (peer_allreduce_requested() && nranks > 1) c.reset(new PeerComm(std::move(c), /*ipc=*/true));
    *out = c.release();
    return 0;
  } catch (const std::exception &ex) {
    g_create_error = ex.what();
    return 1;
  }
}

void msw_comm_destroy(msw_comm_t c) { delete c; }

int msw_core_set_comm(msw_handle h, msw_comm_t comm) {
  return guarded(h, [&] { h->comm = comm; });
}

int msw_core_set_profiling(msw_handle h, int enabled) {
  return guarded(h, [&] { h->profiling = enabled != 0; });
}
int msw_core_set_fixed_iters(msw_handle h, int enabled) {
  return guarded(h, [&] { h->fixed_iters = enabled != 0; });
}
int msw_core_set_pack_schedule(msw_handle h, int enabled) {
  return guarded(h, [&] { h->pack_schedule = enabled != 0; });
}
int msw_core_set_option(msw_handle h, int option, double value) {
  return guarded(h, [&] {
    switch (option) {
      case MSW_OPT_CHECK_EVERY:
        if (!(value >= 1.0 && value <= 65536.0) || value != std::floor(value)) throw Fail("MSW_OPT_CHECK_EVERY: an integer in [1, 65536]");
        h->opts.check_every = (int32_t)value;
        break;
      case MSW_OPT_INIT_BOUND:
        if (std::isnan(value) || value == INFINITY) throw Fail("MSW_OPT_INIT_BOUND: a number below +inf");
        h->opts.init_bound = value;
        break;
      case MSW_OPT_EM_PRIOR:
        if (value != 0.0 && value != 1.0) throw Fail("MSW_OPT_EM_PRIOR: 0 (MAP) or 1 (ML)");
        h->opts.em_prior = (int32_t)value;
        break;
      case MSW_OPT_EM_STOP:
        if (value != 0.0 && value != 1.0) throw Fail("MSW_OPT_EM_STOP: 0 (log-likelihood gain) or 1 (largest move of a weight)");
        h->opts.em_stop = (int32_t)value;
        break;
      default: throw Fail("msw_core_set_option: unknown option id");
    }
  });
}
int msw_core_get_option(msw_handle h, int option, double *value) {
  return guarded(h, [&] {
    if (!value) throw Fail("null out");
    switch (option) {
      case MSW_OPT_CHECK_EVERY: *value = h->opts.check_every; break;
      case MSW_OPT_INIT_BOUND: *value = h->opts.init_bound; break;
      case MSW_OPT_EM_PRIOR: *value = h->opts.em_prior; break;
      case MSW_OPT_EM_STOP: *value = h->opts.em_stop; break;
      default: throw Fail("msw_core_get_option: unknown option id");
    }
  });
}
int msw_core_last_timing(msw_handle h, msw_timing *out) {
  return guarded(h, [&] {
    if (!out) throw Fail("null out");
    *out = h->timing;
  });
}

#ifdef MSW_STAMPS
// diagnostic build only (tools/chain_timeline.py): the phase stamps of the last <= 64 iterations; clear = 1 zeroes them
int msw_debug_stamps(msw_handle h, uint64_t *out, int clear) {
  return guarded(h, [&] {
    MSW_HIP(hipStreamSynchronize(h->stream));
    if (out) MSW_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 64 * 40));
    if (clear) {
      std::vector<unsigned long long> z(64 * 40, 0);
      MSW_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z.data(), z.size() * sizeof(unsigned long long)));
    }
  });
}
#endif

int msw_core_guarded_visits(msw_handle h, uint64_t *out) {
  return guarded(h, [&] {
    if (!out) throw Fail("null out");
    if (h->flavor != 0) throw Fail("msw_core_guarded_visits: no CSR-of-ECs likelihood resident");
    unsigned long long v = 0;
    MSW_HIP(hipStreamSynchronize(h->stream));
    MSW_HIP(hipMemcpy(&v, h->guard_visits.p, sizeof v, hipMemcpyDeviceToHost));
    *out = v;
  });
}

int msw_core_last_bootstrap_timing(msw_handle h, msw_bootstrap_timing *out) {
  return guarded(h, [&] {
    if (!out) throw Fail("null out");
    *out = h->btiming;
  });
}

int msw_core_hbm_stream_rates(msw_handle h, size_t n_bytes, int reps, double *read_gbs, double *triad_gbs) {
  return guarded(h, [&] {
    if (!read_gbs || !triad_gbs) throw Fail("null out");
    if (n_bytes < (1u << 20) || reps < 1) throw Fail("msw_core_hbm_stream_rates: at least 1 MiB and one repetition");
    const size_t n = n_bytes / sizeof(double2);
    DevBuf<double2> a, b, c;
    DevBuf<double> sink;
    a.alloc(n);
    b.alloc(n);
    c.alloc(n);
    sink.alloc(1);
    MSW_HIP(hipMemsetAsync(a.p, 0, n * sizeof(double2), h->stream));
    MSW_HIP(hipMemsetAsync(b.p, 0, n * sizeof(double2), h->stream));
    MSW_HIP(hipMemsetAsync(c.p, 0, n * sizeof(double2), h->stream));
    hipEvent_t e0, e1;
    MSW_HIP(hipEventCreate(&e0));
    MSW_HIP(hipEventCreate(&e1));
    // the rate depends on the launch shape by +-10 %: the best of a few shapes is the ceiling
    const int shapes[7][2] = {{256, 512}, {256, 1024}, {512, 256}, {512, 1024}, {1024, 512}, {2048, 1024}, {8192, 1024}};
    double best_r = 0.0, best_t = 0.0;
    for (int r = 0; r < reps + 2; ++r) {  // the first two rounds warm the clocks and the TLB
      for (const auto &sh : shapes) {
        float ms = 0.f;
        MSW_HIP(hipEventRecord(e0, h->stream));
        k_stream_read<<<sh[0], sh[1], 0, h->stream>>>(a.p, n, sink.p);
        MSW_HIP(hipEventRecord(e1, h->stream));
        MSW_HIP(hipEventSynchronize(e1));
        MSW_HIP(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) best_r = std::max(best_r, (double)(n * sizeof(double2)) / (ms * 1e-3) / 1e9);
        MSW_HIP(hipEventRecord(e0, h->stream));
        k_stream_triad<<<sh[0], sh[1], 0, h->stream>>>(a.p, b.p, c.p, n);
        MSW_HIP(hipEventRecord(e1, h->stream));
        MSW_HIP(hipEventSynchronize(e1));
        MSW_HIP(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) best_t = std::max(best_t, (double)(3 * n * sizeof(double2)) / (ms * 1e-3) / 1e9);
      }
    }
    MSW_HIP(hipGetLastError());
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *read_gbs = best_r;
    *triad_gbs = best_t;
  });
}

}  // extern "C"
