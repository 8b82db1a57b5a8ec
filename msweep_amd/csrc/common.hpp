// common.hpp -- shared host/device declarations of libmsweep_core (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

namespace msw {

constexpr int kWave = 64;          // CDNA wavefront
// One persistent workgroup per CU for the CSR sweeps.  Pass A: 16 wavefronts.  Pass B: 12, i.e. 168
// instead of 128 registers per lane -- enough to keep an EC's 16 x - p0 between the row sum and the
// scatter (no second gather); measured 3 % faster than 16 wavefronts with the hybrid.
#ifndef MSW_PASS_THREADS_A
#define MSW_PASS_THREADS_A 1024
#endif
#ifndef MSW_PASS_THREADS_B
#define MSW_PASS_THREADS_B 768
#endif
// (pass B's short-slice instantiation -- at most 8 rows per slice lane, 117 registers -- runs 16, sweep_kernels.hpp)
#ifndef MSW_PASS_THREADS_B8
#define MSW_PASS_THREADS_B8 1024
#endif
constexpr int kPassThreads = MSW_PASS_THREADS_A;
constexpr int kPassThreadsB = MSW_PASS_THREADS_B;
constexpr int kPassThreadsB8 = MSW_PASS_THREADS_B8;
constexpr int kMaxTrace = 4096;
constexpr int kRedfinParts = 5;    // doubles per workgroup of k_redfin's partial sums (state_kernels.hpp)
constexpr int kRedfinGroups = 16;  // groups per workgroup of k_redfin
// (512 threads: at k_redfin's 68 registers three such workgroups share a CU, so cfg3's 313 workgroups run in ONE round
// on 256 CUs; with 1024 threads a workgroup is alone on its CU and 57 CUs ran two, one after the other: -3 us per
// iteration at 5 000 groups in same-box A/Bs, nothing at 2 000)
#ifndef MSW_REDFIN_THREADS
#define MSW_REDFIN_THREADS 512
#endif
constexpr int kRedfinThreads = MSW_REDFIN_THREADS;           // 512 | 1024: threads per workgroup of k_redfin
constexpr int kRedfinSlots = kRedfinThreads / kRedfinGroups;  // row slots that meet in LDS
// column sums of the CSR sweeps as 64-bit fixed point + integer atomics (sweep_kernels.hpp); 0 = fp64
// atomics, an A/B timing build only
#ifndef MSW_FX
#define MSW_FX 1
#endif
constexpr bool kFx = MSW_FX != 0;

struct HipError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

#define MSW_HIP(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      char _b[512];                                                                        \
      snprintf(_b, sizeof _b, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),       \
               __FILE__, __LINE__);                                                        \
      throw ::msw::HipError(_b);                                                           \
    }                                                                                      \
  } while (0)

// Simple owning device buffer.
template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  bool borrowed = false;  // a view of another handle's buffer: never freed here
  void release() {
    if (p && !borrowed) (void)hipFree(p);
    p = nullptr;
    n = 0;
    borrowed = false;
  }
  void borrow(const DevBuf &o) {
    release();
    p = o.p;
    n = o.n;
    borrowed = o.p != nullptr;
  }
  void alloc(size_t count) {
    if (count <= n && p) return;
    release();
    if (count == 0) count = 1;
    MSW_HIP(hipMalloc((void **)&p, count * sizeof(T)));
    n = count;
  }
  void upload(const T *src, size_t count, hipStream_t s) {
    alloc(count);
    if (count) MSW_HIP(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s));
  }
  void zero(hipStream_t s) {
    if (p) MSW_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s));
  }
};

// Solver options of a handle (include/msweep_core.h MSW_OPT_*): the knobs of rcgpar's loop that are restated from
// memory; the defaults are the restatement itself.
struct SolveOpts {
  int32_t check_every = 1;     // stop rule tested after every iteration (n: after iterations n, 2n, ... only)
  double init_bound = -100000.0;  // rcgpar: `long double bound = -100000.0`
  int32_t em_prior = 0;        // 0 MAP (alpha0 - 1 pseudo-counts), 1 ML
  int32_t em_stop = 0;         // 0 log-likelihood gain < tol, 1 largest move of a weight < tol
};

// MSW_STAMPS (diagnostic build only, tools/chain_timeline.py): s_memrealtime stamps (100 MHz) of the phases of the five
// kernels of an iteration -- g_stamps[(iteration % 64) * 40 + kernel * 8 + phase]; no stamp exists in the product build.
#ifdef MSW_STAMPS
__device__ unsigned long long g_stamps[64 * 40];
#define MSW_STAMP(it, K, p) (g_stamps[((it) & 63) * 40 + (K) * 8 + (p)] = __builtin_amdgcn_s_memrealtime())
#define MSW_STAMP_MAX(it, K, p) atomicMax(&g_stamps[((it) & 63) * 40 + (K) * 8 + (p)], (unsigned long long)__builtin_amdgcn_s_memrealtime())
#else
#define MSW_STAMP(it, K, p) ((void)0)
#define MSW_STAMP_MAX(it, K, p) ((void)0)
#endif

// Per-pass slot tables (n_lut 16-byte entries each), rebuilt by prepB_block whenever `a` moves:
//   A[i] = {x_i, D_i}            x_i = exp(a*T_i), D_i = (1-a)*(T_i - logzi)      (pass A)
//   B[i] = {x_i - p0, x_i*T_i - p0*logzi}              p0 = exp(a*logzi)           (pass B)
struct TabDev {
  double2 *A, *B;
};

// Scalar state of one solve; lives in device memory, mirrored to pinned host memory when
// the host polls.  Kept POD.
struct Scalars {
  // gamma = a*L + u - lse ; oldstep = (os_a, os_u) ; step = (step_a, step_u)
  double a, os_a, step_a;
  double oldnorm, newnorm, beta;
  double bound, oldbound, bound_const;
  double tol, csum;
  // per-pass shared quantities
  double M, U, p0, kappa;
  double logzi;
  // fixed-point column sums (sweep_kernels.hpp kFx): 2^K and 2^-K; 2^t / 2^-t of the guarded ECs' shares
  // (sell.hpp); xb >= every table value x_i = exp(a T_i) and p0 of the current pass; extreme table values
  double fx_scale, fx_inv, fx_tscale, fx_tinv, xb, tmax, tmin;
  // every exp(a T) of a pass is formed as exp(a (T - tref)), tref = the table value with the largest a T
  // (incl. log zi): x_i <= 1, p0 <= 1 whatever a does; the ELBO gets a * tref * sum c back (k_finstep)
  double tref;
  int32_t didreset, reset_pending, done, iter;
  int32_t have_eval, eval_pad;  // what the previous slot left for k_finstep's verdict: 0 nothing, 1 an evaluation, 2 the initial one
  int32_t max_iters, fixed_iters, trace_theta, flavor;  // flavor: 0 csr, 1 dense
  // solver options (msw_core_set_option): the stop rule is tested after iterations n, 2n, ... only; EM variants
  int32_t check_every, em_prior, em_stop, opt_pad;
  int32_t tab_ver, fx_shift;  // fx_shift: device_util.hpp fx_factor, set with p0 / xb  // bumped whenever (a) changes the per-slot tables: k_tables (large slot areas) follows
};

}  // namespace msw
