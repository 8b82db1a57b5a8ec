// state_kernels.hpp -- the O(G) kernels between the sweeps: Fletcher-Reeves step, column-sum
// reduction + N_g / lgamma / digamma (spread over many workgroups: the transcendental work of
// 5k groups on ONE CU cost more than a sweep), ELBO + reset / convergence decision, and the
// gradient preparation for the next pass A.
//
// One "slot" = k_passA -> k_finstep -> k_passB -> k_redfin (k_finstep below says what a slot does).
#pragma once
#include "device_util.hpp"

namespace msw {

struct TraceDev {
  double *bound, *newnorm, *beta, *theta;
  int32_t *didreset;
};

// From (a, u): M = max u, e_g = exp(u_g - M), U = sum e, p0 = exp(a*logzi) and the per-slot
// tables of both sweeps (common.hpp TabDev).  One 1024-thread workgroup.  Entries G.. of e are
// the sentinel groups of SELL padding (zero, never rewritten).
__device__ inline void prepB_block(Scalars *sc, double a, int G, int n_lut, const double *u,
                                   const double *lut, double *e, TabDev X, double *sh) {
  const int tid = threadIdx.x, nt = blockDim.x;
  double m = -INFINITY;
  for (int g = tid; g < G; g += nt) m = fmax(m, u[g]);
  const double M = block_max(m, sh);
  double su = 0.0;
  for (int g = tid; g < G; g += nt) {
    const double eg = flush_denormal(exp(u[g] - M));
    e[g] = eg;
    su += eg;
  }
  const double U = block_sum(su, sh);
  const double logzi = sc->logzi, oma = 1.0 - a;
  const double tref = tref_of(a, sc->tmax, sc->tmin);  // tmax / tmin include log zi
  const double p0 = exp(a * (logzi - tref));
  const double xtop = 1.0;  // exp(a (tref - tref)): the largest table value of the pass
  for (int i = tid; i < n_lut; i += nt) {
    const double T = lut[i], x = exp(a * (T - tref));
    X.A[i] = make_double2(x, oma * (T - logzi));
    X.B[i] = make_double2(x - p0, x * T - p0 * logzi);
  }
  if (tid == 0) {
    sc->M = M;
    sc->U = U;
    sc->p0 = p0;
    sc->tref = tref;
    sc->xb = fmax(xtop, p0);
    sc->fx_shift = fx_shift_of(fmax(xtop, p0), p0);
    sc->tab_ver = sc->tab_ver + 1;
  }
}

// The per-slot tables for slot areas too large to be rebuilt by the single workgroup of k_finstep /
// k_prepB (those are then called with n_lut = 0 and this kernel follows them): any number
// of workgroups, and nothing to do when the tables already belong to the current a (built[0] =
// version they were built for, built[1] = workgroups that have finished).
constexpr int kTabInline = 16384;
__global__ __launch_bounds__(256) void k_tables(const Scalars *sc, int n_tab, const double *lut, TabDev X,
                                               int *built) {
  const int ver = sc->tab_ver;
  if (built[0] != ver) {
    const double a = sc->a, logzi = sc->logzi, oma = 1.0 - a;
    const double tref = tref_of(a, sc->tmax, sc->tmin);
    const double p0 = exp(a * (logzi - tref));
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_tab; i += gridDim.x * blockDim.x) {
      const double T = lut[i], x = exp(a * (T - tref));
      X.A[i] = make_double2(x, oma * (T - logzi));
      X.B[i] = make_double2(x - p0, x * T - p0 * logzi);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(&built[1], 1) == (int)gridDim.x - 1) {  // the last workgroup: all have read built[0]
    built[1] = 0;
    built[0] = ver;
  }
}

// {max, min} of n values (at upload: the table values of the slot area, or -- value records -- every listed cell's
// value, up to 5e8 of them: any number of workgroups, each leaving its pair in out[2 b], out[2 b + 1]; a second
// launch with stride2 = 1 over the pairs -- one workgroup -- leaves the result in out[0], out[1])
__global__ __launch_bounds__(1024) void k_minmax(const double *v, uint64_t n, int pairs, double *out) {
  __shared__ double sh[32];
  double mx = -INFINITY, mn = INFINITY;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  if (pairs) {  // v holds n {max, min} pairs
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
      mx = fmax(mx, v[2 * i]);
      mn = fmin(mn, v[2 * i + 1]);
    }
  } else {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
      mx = fmax(mx, v[i]);
      mn = fmin(mn, v[i]);
    }
  }
  mx = block_max(mx, sh);
  mn = -block_max(-mn, sh);
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = mx;
    out[2 * blockIdx.x + 1] = mn;
  }
}

__global__ __launch_bounds__(1024) void k_prepB(Scalars *sc, int G, int n_lut, const double *u,
                                               const double *lut, double *e, TabDev X) {
  __shared__ double sh[32];
  if (sc->done) return;
  if (sc->flavor != 0) return;  // dense flavour needs no tables
  const double a = sc->a;
  __syncthreads();
  prepB_block(sc, a, G, n_lut, u, lut, e, X, sh);
}

// ONE kernel between pass A and pass B (round 4; k_fin + k_step until then -- two kernels, two launch boundaries
// and two chains of memory round trips per iteration, tools/chain_timeline.py): the VERDICT on the evaluation the
// previous slot left behind (rcgpar ELBO_rcg_mat + bound_const, the `bound < oldbound` steepest-descent retry --
// revert_step --, the convergence test), then the Fletcher-Reeves STEP (rcgpar rcg_optl_mat: beta_FR, oldstep scaling,
// gamma += step) on the (a, u) state and the pass-B preparation.  Pass A therefore runs BEFORE the verdict on the
// state it sweeps: its |g|^2 is wasted only after a rejected step, and it takes the two background moments it needs
// (S1 = sum e s0, S2 = sum e s0^2) from k_redfin's partial sums itself (sweep_kernels.hpp).
//
// One "slot" = k_passA -> k_finstep -> k_passB -> k_redfin.  Scalars::have_eval says what the previous slot left:
//   2 the initial evaluation (update_N_k on gamma = log(1/G)): bookkeeping only, then the first step;
//   1 an evaluation: accepted -> iteration count, trace, stop rule, then the next step (none when done);
//                    rejected (bound < oldbound) -> revert to steepest descent, NO step: the pass B of this slot
//                    re-evaluates, and the next slot's verdict accepts it unconditionally (reset_pending);
//   0 nothing pending (a closing verdict has been taken: mode 1 below): the step alone.
// mode 1 = verdict only: closes a run of slots (fixed-iteration runs, max_iters) -- the last evaluation's verdict
// without a further step.  A solve needs iterations + rejected steps slots, as before.
// This single-workgroup kernel sits between the two sweeps of every iteration, so it is organised around memory
// round trips, not arithmetic: every load is issued up front, BEFORE the state decides what runs; u stays in
// registers from the step to the exp; one pair of barriers serves the six sums.  Groups beyond kStepRegs * 1024 take
// the (slower) looping paths.
constexpr int kStepRegs = 6;  // groups per thread the register paths hold (6144 groups; more: the looping paths)
__global__ __launch_bounds__(1024) void k_finstep(Scalars *sc, int mode, int G, int n_lut, int n_partA,
                                                 const double *partA, int npartR, const double *totS,
                                                 const double *partR, const double *Nc, const double *w, double *u,
                                                 double *os_u, double *step_u, const double *lut, double *e, TabDev X,
                                                 TraceDev tr) {
  __shared__ double sh[16 * (kRedfinParts + 1)];
  const int tid = threadIdx.x, nt = blockDim.x;
  const bool inreg = G <= kStepRegs * nt;
  // ---- every load of the kernel: one memory round trip on the critical path between the sweeps
  double q[kRedfinParts + 1] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // k_redfin's five sums, |g|^2
  for (int b = tid; b < npartR; b += nt)
    for (int i = 0; i < kRedfinParts; ++i) q[i] += partR[kRedfinParts * b + i];
  for (int i = tid; i < n_partA; i += nt) q[kRedfinParts] += partA[i];
  const double s_clogZ = totS[0], s_rH = totS[1];
  // (w, u and the last step -- the oldstep of the common case, an accepted evaluation; the scaled oldstep os_u is
  // read where it is needed: after a rejected step and for the first step of a run)
  double wv[kStepRegs], sv[kStepRegs], uv[kStepRegs];
  if (inreg) {
#pragma unroll
    for (int k = 0; k < kStepRegs; ++k) {
      const int g = tid + k * nt;
      if (g < G) {
        wv[k] = w[g];
        sv[k] = step_u[g];
        uv[k] = u[g];
      }
    }
  }
  const double lt = tid < n_lut ? lut[tid] : 0.0;  // first table entry of this thread
  if (tid == 0) MSW_STAMP(sc->iter, 1, 0);
  const Scalars s0 = *sc;  // one read of the whole state: no dependent scalar round trips later
  if (s0.done) return;
  block_sum_fixed<kRedfinParts + 1, 16>(q, sh);  // one pair of barriers for the six sums
  if (tid == 0) MSW_STAMP(s0.iter, 1, 2);
  const double lg = q[0], mu = q[1], S0 = q[2], S1 = q[3], pnsum = q[5];
  const int flavor = s0.flavor;
  const double logzi = s0.logzi;
  double a = s0.a, os_a = s0.os_a, kappa = s0.kappa, bound = s0.bound;
  int didreset = s0.didreset, it = s0.iter, done = 0;
  bool old_is_step = false;  // the accepted step becomes oldstep (rcgpar: oldstep = step) -- folded into the step below

  // ---- the verdict on the pending evaluation ------------------------------------------------------------
  if (s0.have_eval == 2) {
    if (flavor == 0) kappa += S1 / S0;  // the initial update_N_k: no bound, no iteration
  } else if (s0.have_eval == 1) {
    const int reeval = s0.reset_pending;
    const double coef = (flavor == 0) ? (1.0 - a) : 1.0;
    // (the sweeps' Z carries exp(-a tref): sum c log Z gets a * tref * sum c back; 0 for the dense flavour)
    const double nb = s0.bound_const + s_clogZ + coef * s_rH + mu + lg + (flavor == 0 ? a * s0.tref * s0.csum : 0.0);
    if (!reeval && nb < s0.oldbound) {
      // bad step: revert to steepest descent (gamma += oldm; gamma -= oldstep); this slot's pass B re-evaluates
      double a2 = a;
      if (s0.beta > 0) {
        a2 = a - os_a;
        if (inreg) {
#pragma unroll
          for (int k = 0; k < kStepRegs; ++k)
            if (tid + k * nt < G) u[tid + k * nt] = uv[k] - os_u[tid + k * nt];
        } else {
          for (int g = tid; g < G; g += nt) u[g] -= os_u[g];
        }
      }
      __syncthreads();
      if (tid == 0) {
        sc->a = a2;
        sc->didreset = 1;
        sc->reset_pending = 1;
        sc->bound = nb;
        // the re-evaluation: by this slot's pass B -- or, when this is a closing verdict (mode 1: no sweep follows
        // this launch), by the next slot's, which finds reset_pending without an evaluation and takes no step
        sc->have_eval = mode == 1 ? 0 : 1;
      }
      if (flavor == 0) prepB_block(sc, a2, G, n_lut, u, lut, e, X, sh);
      return;
    }
    // accepted (or the re-evaluation after a rejected step, which is taken as it is)
    if (it < s0.trace_theta && tr.theta)
      for (int g = tid; g < G; g += nt) tr.theta[(size_t)it * G + g] = Nc[g] / s0.csum;
    // (check_every > 1: the rule is tested after iterations n, 2n, ... only -- msw_core_set_option)
    if (!s0.fixed_iters && (nb - s0.oldbound < s0.tol) && !didreset && (s0.check_every <= 1 || (it + 1) % s0.check_every == 0))
      done = 1;
    if (it + 1 >= s0.max_iters) done = 1;
    if (tid == 0 && it < kMaxTrace) {
      tr.bound[it] = nb;
      tr.newnorm[it] = s0.newnorm;
      tr.beta[it] = s0.beta;
      tr.didreset[it] = didreset;
    }
    old_is_step = !reeval;
    if (old_is_step) os_a = s0.step_a;
    if (flavor == 0) kappa += S1 / S0;
    bound = nb;
    it += 1;
  } else if (s0.reset_pending) {
    // a closing verdict (mode 1) rejected the last evaluation: this slot's pass B re-evaluates first
    if (tid == 0) sc->have_eval = 1;
    return;
  }
  if (done || mode == 1) {  // the verdict alone: the state stays the evaluated one
    if (old_is_step) {      // ... with its step as oldstep, should the solve be continued (msw_core_continue)
      if (inreg) {
#pragma unroll
        for (int k = 0; k < kStepRegs; ++k)
          if (tid + k * nt < G) os_u[tid + k * nt] = sv[k];
      } else {
        for (int g = tid; g < G; g += nt) os_u[g] = step_u[g];
      }
    }
    if (tid == 0) {
      sc->os_a = os_a;
      sc->bound = bound;
      sc->kappa = kappa;
      sc->reset_pending = 0;
      sc->iter = it;
      sc->done = done;
      sc->have_eval = 0;
      MSW_STAMP(it, 4, 7);
    }
    return;
  }

  // ---- the step ---------------------------------------------------------------------------------------------
  // (one group: q_j is the constant 1 and the variance exactly 0 -- the sweep's background form would
  // leave rounding noise, and a ratio of two noises as beta)
  const double newnorm = G == 1 ? 0.0 : pnsum;
  // (an exactly stationary start -- identical groups under a symmetric prior -- makes the ratio x/0: the
  // reference carries the inf / NaN into its state and returns NaN weights; here such a step has no momentum)
  const double ratio = newnorm / s0.oldnorm;
  const double beta = ratio < INFINITY ? ratio : 0.0;
  double step_a = 1.0 - a;
  if (didreset) {
    os_a *= 0.0;
  } else if (beta > 0) {
    os_a *= beta;
    step_a += os_a;
  }
  const double a_new = a + step_a;
  if (!inreg) {
    const double *osrc = old_is_step ? step_u : os_u;
    for (int g = tid; g < G; g += nt) {
      double osu = osrc[g], su = w[g];
      if (didreset) {
        osu *= 0.0;
      } else if (beta > 0) {
        osu *= beta;
        su += osu;
      }
      os_u[g] = osu;
      step_u[g] = su;
      u[g] += su;
    }
    __syncthreads();
    if (tid == 0) {
      sc->a = a_new;
      sc->os_a = os_a;
      sc->step_a = step_a;
      sc->oldnorm = newnorm;
      sc->newnorm = newnorm;
      sc->beta = beta;
      sc->didreset = 0;
      sc->reset_pending = 0;
      sc->bound = bound;
      sc->oldbound = bound;
      sc->kappa = kappa;
      sc->iter = it;
      sc->have_eval = 1;
    }
    if (flavor == 0) prepB_block(sc, a_new, G, n_lut, u, lut, e, X, sh);
    return;
  }
  double m = -INFINITY;
#pragma unroll
  for (int k = 0; k < kStepRegs; ++k) {
    const int g = tid + k * nt;
    if (g < G) {
      double osu = old_is_step ? sv[k] : os_u[g], su = wv[k];
      if (didreset) {
        osu *= 0.0;
      } else if (beta > 0) {
        osu *= beta;
        su += osu;
      }
      uv[k] += su;
      os_u[g] = osu;
      step_u[g] = su;
      u[g] = uv[k];
      m = fmax(m, uv[k]);
    }
  }
  double p0 = 0.0, tref = 0.0;
  if (flavor == 0) {  // per-slot tables of both sweeps (prepB_block's arithmetic)
    const double oma = 1.0 - a_new;
    tref = tref_of(a_new, s0.tmax, s0.tmin);
    p0 = exp(a_new * (logzi - tref));
    for (int i = tid; i < n_lut; i += nt) {
      const double T = i == tid ? lt : lut[i], x = exp(a_new * (T - tref));
      X.A[i] = make_double2(x, oma * (T - logzi));
      X.B[i] = make_double2(x - p0, x * T - p0 * logzi);
    }
  }
  double M = 0.0, U = 0.0;
  if (flavor == 0) {
    M = block_max_fixed<16>(m, sh);
    double su = 0.0;
#pragma unroll
    for (int k = 0; k < kStepRegs; ++k) {
      const int g = tid + k * nt;
      if (g < G) {
        const double eg = flush_denormal(exp(uv[k] - M));
        e[g] = eg;
        su += eg;
      }
    }
    U = block_sum_fixed1<16>(su, sh);
  }
  if (tid == 0) MSW_STAMP(s0.iter, 1, 6);
  if (tid == 0) {
    sc->a = a_new;
    sc->tab_ver = s0.tab_ver + 1;
    sc->os_a = os_a;
    sc->step_a = step_a;
    sc->oldnorm = newnorm;
    sc->newnorm = newnorm;
    sc->beta = beta;
    sc->didreset = 0;
    sc->reset_pending = 0;
    sc->bound = bound;
    sc->oldbound = bound;
    sc->kappa = kappa;
    sc->iter = it;
    sc->have_eval = 1;
    if (flavor == 0) {
      sc->M = M;
      sc->U = U;
      sc->p0 = p0;
      sc->tref = tref;
      sc->xb = 1.0;
      sc->fx_shift = fx_shift_of(1.0, p0);
    }
  }
}

// Column sums across workgroups (fixed order) fused with the per-group math that follows them:
// Nc_g, N_g, lgamma(N_g), (M - u_g) * Nc_g, w_g = digamma(N_g) - 1 - u_g and the pass-A gradient
// preparation {e_g, w_g - kappa} with its sums S0 = sum e, S1 = sum e*s0, S2 = sum e*s0^2
// (s0_g = w_g - kappa: the step value of a background cell relative to which pass A measures the
// listed cells; kappa = lagged centring constant, advanced by k_finstep to the e-weighted mean just measured:
// kappa += S1 / S0).  One workgroup of kRedfinThreads = 512 threads per
// kRedfinGroups = 16 groups (313 workgroups at 5k groups: the 10 MB of partial rows pass B left
// behind are read by the whole chip, not by 79 CUs): thread t sums rows t/16, t/16 + 32, ... of
// group t%16 (a row's 16 groups are one 128-byte line), 32 row slots meet in LDS in fixed order.
//   nblk > 0: sum partAcc[b*G + g] over b;  nblk == 0: Acc already holds the totals.
// Block 0 also leaves the totals of the per-workgroup ELBO terms for k_finstep in totS[0..2].
__global__ __launch_bounds__(kRedfinThreads) void k_redfin(const Scalars *sc, int G, int nblk, int fxrows,
                                                unsigned long long *tail, int zero_tail,
                                                int npartS, const double *partAcc, const double *Acc,
                                                const double *partS, const double *e, const double *u,
                                                const double *alpha0, double *Nc, double *N, double *w,
                                                double2 *ew, double *partR, double *totS) {
  __shared__ double sh[48];
  __shared__ double accs[kRedfinSlots][kRedfinGroups];
  const int tid = threadIdx.x, gl = tid & (kRedfinGroups - 1), rs = tid >> 4;
  const int g = blockIdx.x * kRedfinGroups + gl;
  if (tid == 0 && blockIdx.x == 0) MSW_STAMP(sc->iter, 3, 0);
  const Scalars s0 = *sc;  // in flight together with the partial rows below
  // CSR flavour: the rows are 64-bit fixed-point integers (sweep_kernels.hpp kFx) -- summed as integers,
  // exactly, whatever the number of rows; dense flavour: fp64 rows in fixed order
  const bool fx = kFx && fxrows;
  double s = 0.0;
  long long si = 0;
  if (g < G) {
    if (nblk > 0) {
      if (fx) {
        const long long *pi = reinterpret_cast<const long long *>(partAcc);
        for (int b = rs; b < nblk; b += kRedfinSlots) si += pi[(size_t)b * G + g];
      } else {
        for (int b = rs; b < nblk; b += kRedfinSlots) s += partAcc[(size_t)b * G + g];
      }
    } else if (rs == 0) {
      if (fx) si = reinterpret_cast<const long long *>(Acc)[g];
      else s = Acc[g];
    }
  }
  if (fx) s = __longlong_as_double(si);  // carried through the LDS exchange as bits
  // per-group operands of the math below: loaded now, needed after the reductions
  // (the math is shared by the first two wavefronts, 16 lanes each: wavefront 0 takes lgamma and the bound's terms,
  // wavefront 1 digamma and the gradient's -- two latency chains side by side instead of one after the other)
  const bool mathlane = (tid & 63) < kRedfinGroups && tid < 128 && g < G;
  const int role = tid >> 6;
  double ug0 = 0.0, eg0 = 0.0, al0 = 0.0;
  long long th = 0, tl = 0;  // the guarded ECs' shares of the group (sell.hpp), two fixed-point limbs
  if (mathlane) {
    ug0 = u[g];
    eg0 = e[g];
    al0 = alpha0[g];
    if (tail) {
      th = (long long)tail[2 * (size_t)g];
      tl = (long long)tail[2 * (size_t)g + 1];
    }
  }
  // W = sum_j r_j (and, for k_finstep, the other two ELBO sums): every workgroup forms them in the
  // same fixed order
  double t[3] = {0.0, 0.0, 0.0};
  for (int b = tid; b < npartS; b += kRedfinThreads) {
    t[0] += partS[4 * b];
    t[1] += partS[4 * b + 1];
    t[2] += partS[4 * b + 2];
  }
  if (s0.done) return;
  if (tail && zero_tail && mathlane && role == 0 && (th | tl)) {  // consumed: ready for the next pass B
    tail[2 * (size_t)g] = 0;
    tail[2 * (size_t)g + 1] = 0;
  }
  accs[rs][gl] = s;
  if (tid == 0 && blockIdx.x == 0) MSW_STAMP(s0.iter, 3, 1);
  block_sum_fixed<3, kRedfinThreads / 64>(t, sh);  // its barriers also publish accs
  if (tid == 0 && blockIdx.x == 0) MSW_STAMP(s0.iter, 3, 2);
  const double W = t[2];
  if (blockIdx.x == 0 && tid == 0) {
    totS[0] = t[0];
    totS[1] = t[1];
    totS[2] = t[2];
  }
  if (tid >= 128) return;
  double lgv = 0.0, muv = 0.0, s0v = 0.0, s1v = 0.0, s2v = 0.0;
  if (mathlane) {
    double A = 0.0;
    long long ai = 0;
    if (fx) {
#pragma unroll
      for (int i = 0; i < kRedfinSlots; ++i) ai += __double_as_longlong(accs[i][gl]);
    } else {
#pragma unroll
      for (int i = 0; i < kRedfinSlots; ++i) A += accs[i][gl];
    }
    double nc;
    const double ug = ug0;
    const int flavor = s0.flavor;
    if (flavor == 0 && fx) {
      // N_g = e_g p0 W + e_g sum_j r_j (x_gj - p0).  The second part arrives as an integer in units of
      // 2^-K e_g / f_g reads -- MODULO 2^64: near the guard threshold it is up to 2^8 times -sum c (the
      // background share it cancels against) and does not fit 64 bits.  The first part is therefore added in
      // the same units and the same modular arithmetic: the sum, N_g itself, is below 2^62 units and comes
      // out right whatever its two parts wrapped to.  (2^K f_g p0 W < 2^83: split at 2^32 like the cells'
      // two-part adds; its fp64 rounding, 2^-53 of the background share, is what the guard bounds.)
      // (fxrows == 2: the fp32 EM sweep's rows, em_f32_kernels.hpp -- no per-group factor, units of 2^-K reads)
      const double fg = fxrows == 2 ? (eg0 > 0.0 ? eg0 : 1.0) : fx_factor(eg0, fx_expbits(s0.fx_shift));
      const double t1 = fg * (s0.p0 * W) * s0.fx_scale;
      const double th1 = floor(t1 * 0x1p-32), tl1 = fma(-th1, 0x1p32, t1);
      const unsigned long long b =
          ((unsigned long long)(uint32_t)__double2loint(th1 + 6755399441055744.0) << 32) + (unsigned long long)__double2ll_rn(tl1);
      const long long tot = (long long)((unsigned long long)ai + b);
      nc = fmax((double)tot * s0.fx_inv * (eg0 / fg), 0.0);
      nc += (double)th * s0.fx_tinv + (double)tl * (s0.fx_tinv * 0x1p-36);
      muv = (s0.M - ug) * nc;
    } else if (flavor == 0) {
      nc = eg0 * (s0.p0 * W + A);
      nc += (double)th * s0.fx_tinv + (double)tl * (s0.fx_tinv * 0x1p-36);
      muv = (s0.M - ug) * nc;
    } else {
      nc = A;
    }
    const double n = al0 + nc;
    if (role == 0) {
      Nc[g] = nc;
      N[g] = n;
      lgv = lgamma(n);
    } else {
      muv = 0.0;  // (the bound's terms belong to wavefront 0)
      const double wg = digamma_ref(n) - 1.0 - ug;
      w[g] = wg;
      if (flavor == 0) {
        const double eg = eg0, wc = wg - s0.kappa;
        const double sb = wc;
        ew[g] = make_double2(eg, wc);
        s0v = eg;
        s1v = eg * sb;
        s2v = eg * sb * sb;
      }
    }
  }
  double *o = partR + kRedfinParts * blockIdx.x;
  if (role == 0) {  // wave-uniform
    lgv = wave_sum(lgv);
    muv = wave_sum(muv);
    if (tid == 0) {
      o[0] = lgv;
      o[1] = muv;
    }
  } else {
    s0v = wave_sum(s0v);
    s1v = wave_sum(s1v);
    s2v = wave_sum(s2v);
    if (tid == 64) {
      MSW_STAMP_MAX(s0.iter, 3, 7);
      o[2] = s0v;
      o[3] = s1v;
      o[4] = s2v;
    }
  }
}

// ---------------------------------------------------------------------------------------
// EC-sharded solve (comm.hpp): local sums packed for the two all-reduces of an iteration.
// ---------------------------------------------------------------------------------------
// out[0] = sum of part[0..n) (fixed order).  `gate` = 1: skipped like pass A.
__global__ __launch_bounds__(1024) void k_sum_scalar(const Scalars *sc, int gate, int n, const double *part,
                                                    double *out) {
  __shared__ double sh[32];
  if (gate && sc->done) return;
  double p = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) p += part[i];
  p = block_sum(p, sh);
  if (threadIdx.x == 0) out[0] = p;
}

// out[g] = column sum of this rank's ECs (g < G); out[G .. G+3] = {sum c log Z, sum r H, sum r, 0}
// in the layout of one partS entry, so that k_redfin / k_finstep consume `out` as totals.
// out = [G column sums][2 G limbs of the guarded ECs' shares (tail; zeroed once read)][4 ELBO terms]
__global__ __launch_bounds__(1024) void k_colsum(const Scalars *sc, int G, int nrows, int fxrows, int npartS,
                                                const double *partAcc, const double *Acc,
                                                const double *partS, unsigned long long *tail, double *out) {
  __shared__ double sh[32];
  __shared__ double accs[16][64];
  if (sc->done) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int g = blockIdx.x * 64 + lane;
  if (blockIdx.x == 0) {
    double p1 = 0.0, p2 = 0.0, p3 = 0.0;
    for (int b = tid; b < npartS; b += 1024) {
      p1 += partS[4 * b];
      p2 += partS[4 * b + 1];
      p3 += partS[4 * b + 2];
    }
    p1 = block_sum(p1, sh);
    p2 = block_sum(p2, sh);
    p3 = block_sum(p3, sh);
    if (tid == 0) {
      out[3 * (size_t)G] = p1;
      out[3 * (size_t)G + 1] = p2;
      out[3 * (size_t)G + 2] = p3;
      out[3 * (size_t)G + 3] = 0.0;
    }
  }
  if (wv == 1 && g < G) {
    unsigned long long *o = reinterpret_cast<unsigned long long *>(out) + G;
    const unsigned long long a0 = tail ? tail[2 * (size_t)g] : 0ull, a1 = tail ? tail[2 * (size_t)g + 1] : 0ull;
    o[2 * (size_t)g] = a0;
    o[2 * (size_t)g + 1] = a1;
    if (a0 | a1) {
      tail[2 * (size_t)g] = 0;
      tail[2 * (size_t)g + 1] = 0;
    }
  }
  const bool fx = kFx && fxrows;  // fixed-point rows: integer sums, handed on as bits (all-reduced as integers)
  double s = 0.0;
  long long si = 0;
  if (g < G) {
    if (nrows > 0) {
      if (fx) {
        const long long *pi = reinterpret_cast<const long long *>(partAcc);
        for (int b = wv; b < nrows; b += 16) si += pi[(size_t)b * G + g];
      } else {
        for (int b = wv; b < nrows; b += 16) s += partAcc[(size_t)b * G + g];
      }
    } else if (wv == 0) {
      if (fx) si = reinterpret_cast<const long long *>(Acc)[g];
      else s = Acc[g];
    }
  }
  accs[wv][lane] = fx ? __longlong_as_double(si) : s;
  __syncthreads();
  if (wv == 0 && g < G) {
    if (fx) {
      long long ai = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) ai += __double_as_longlong(accs[i][lane]);
      out[g] = __longlong_as_double(ai);
    } else {
      double A = 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) A += accs[i][lane];
      out[g] = A;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Solve set-up: c_j = exp(logc_j) (or the bootstrap counts) gathered into the permuted EC
// order, sum of counts, bound constant (rcgpar calc_bound_const), initial gamma = log(1/G).
// perm == nullptr: identity (dense flavour).
// ---------------------------------------------------------------------------------------
// Byte image of a multiplicity for pass B's stream (8 bytes per EC would be a sixth of that
// sweep's HBM traffic): small integer counts -- the normal case -- as themselves, anything else
// as kC8Escape.  c is snapped to the integer it is within rounding of (exp(log(n)) need not be n).
__device__ __forceinline__ uint8_t c8_of(double &c) {
  const double r = rint(c);
  if (r >= 0.0 && r < 255.0 && fabs(c - r) <= 1e-12 * r) {
    c = r;
    return (uint8_t)r;
  }
  return (uint8_t)255;
}

// c8s: the byte again for every slice lane that works for the EC (sell.hpp slice classes: pass B reads it with the
// slice, one byte per lane, whatever the class); nullptr: no slices (dense flavour)
__device__ __forceinline__ void c8_to_lanes(uint8_t *c8s, const SliceClasses &cls, uint32_t n_long, uint32_t j, uint8_t b) {
  if (!c8s || j < n_long) return;
  uint32_t s, lgm, l0;
  slice_of_position(cls, j - n_long, s, lgm, l0);
  for (uint32_t i = 0; i < (1u << lgm); ++i) c8s[(size_t)s * 64 + l0 + i] = b;
}

__global__ __launch_bounds__(256) void k_cvec_from_logc(const double *logc, const uint32_t *perm,
                                                       uint32_t E, double *cvec, uint8_t *c8, double *part,
                                                       SliceClasses cls, uint32_t n_long, uint8_t *c8s) {
  __shared__ double sh[32];
  double s = 0.0;
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < E; j += gridDim.x * blockDim.x) {
    double c = exp(logc[perm ? perm[j] : j]);
    c8[j] = c8_of(c);
    c8_to_lanes(c8s, cls, n_long, j, c8[j]);
    cvec[j] = c;
    s += c;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_cvec_from_counts(const uint32_t *cnt, const uint32_t *perm,
                                                         uint32_t E, double *cvec, uint8_t *c8, double *part,
                                                         SliceClasses cls, uint32_t n_long, uint8_t *c8s) {
  __shared__ double sh[32];
  double s = 0.0;
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < E; j += gridDim.x * blockDim.x) {
    double c = (double)cnt[perm ? perm[j] : j];
    c8[j] = c8_of(c);
    c8_to_lanes(c8s, cls, n_long, j, c8[j]);
    cvec[j] = c;
    s += c;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// fixed-iteration runs: n more iterations of the solve that has just finished its quota
__global__ void k_extend(Scalars *sc, int n) {
  sc->max_iters += n;
  sc->done = 0;
}

__global__ __launch_bounds__(1024) void k_init_state(Scalars *sc, int G, int npart, const double *part,
                                                    const double *alpha0, double *u, double *os_u,
                                                    double *step_u, double tol, int max_iters,
                                                    int fixed_iters, int trace_theta, int flavor,
                                                    double logzi, SolveOpts so, int *tab_built,
                                                    const double *trange) {
  const double init_bound = so.init_bound;
  __shared__ double sh[32];
  const int tid = threadIdx.x, nt = blockDim.x;
  double s = 0.0;
  for (int i = tid; i < npart; i += nt) s += part[i];
  const double csum = block_sum(s, sh);
  double sa = 0.0, sl = 0.0;
  for (int g = tid; g < G; g += nt) {
    sa += alpha0[g];
    sl += lgamma(alpha0[g]);
    u[g] = 0.0;
    os_u[g] = 0.0;
    step_u[g] = 0.0;
  }
  sa = block_sum(sa, sh);
  sl = block_sum(sl, sh);
  if (tid == 0) {
    Scalars z = {};
    z.a = 0.0;
    z.oldnorm = 1.0;
    z.bound = init_bound;
    z.oldbound = init_bound;
    z.bound_const = lgamma(sa) - lgamma(sa + csum) - sl;
    z.tol = tol;
    z.csum = csum;
    z.logzi = logzi;
    z.max_iters = max_iters;
    z.fixed_iters = fixed_iters;
    z.trace_theta = trace_theta;
    z.flavor = flavor;
    int ex = 0;
    frexp(csum > 1.0 ? csum : 1.0, &ex);  // csum < 2^ex
    // csum * 2^K < 2^61 (device_util.hpp fx_factor).  No other cap: with few reads most ECs then take the
    // two-part adds (their addends exceed the 2^51 of the one-fma conversion), which only costs time where
    // there is none to lose, and a toy problem keeps 18 digits below its single read
    const int k = 61 - ex < 120 ? 61 - ex : 120;
    // guarded ECs' shares, in reads: a share is at most its EC's count <= csum < 2^ex, and share * 2^t has
    // to stay below the 2^51 of the one-fma conversion (the second limb carries the next 36 bits)
    const int t = 50 - ex;
    z.fx_scale = ldexp(1.0, k);
    z.fx_inv = ldexp(1.0, -k);
    z.fx_tscale = ldexp(1.0, t);
    z.fx_tinv = ldexp(1.0, -t);
    z.tmax = fmax(trange[0], logzi);
    z.tmin = fmin(trange[1], logzi);
    z.xb = 1.0;
    z.tref = 0.0;
    z.fx_shift = 9;
    z.have_eval = 2;  // the initial evaluation follows (k_finstep)
    z.check_every = so.check_every < 1 ? 1 : so.check_every;
    z.em_prior = so.em_prior;
    z.em_stop = so.em_stop;
    *sc = z;
    tab_built[0] = -1;  // no tables yet
    tab_built[1] = 0;
  }
}

}  // namespace msw
