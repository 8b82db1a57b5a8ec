// kernels.hpp -- hand-written gfx950 kernels of the RCG abundance-estimation loop.
//
// Formulation: the log-responsibilities keep the closed form
//     gamma(g, j) = a * L(g, j) + u_g - lse_j ,   lse_j = logsumexp_g(a*L(g,j) + u_g)
// through every operation of rcgpar's RCG loop (reference call site
// src/mSWEEP.cpp:194-198), so the G x E matrices gamma / step / oldstep of the reference
// never exist.  One iteration = one "pass A" sweep (|g|^2 of the natural gradient) and
// one "pass B" sweep (per-EC softmax, column sums N_g, ELBO terms) over the read-only
// likelihood, plus O(G) kernels in between.
#pragma once
#include "common.hpp"

namespace msw {

// ---------------------------------------------------------------------------------------
// wave / block reductions (wave64; xor butterflies: every lane ends with the same value,
// the order of additions is fixed -> bitwise reproducible)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, kWave));
  return v;
}
// sh: >= 16 doubles of LDS scratch.  Result valid in every thread.
__device__ __forceinline__ double block_sum(double v, double *sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double r = 0.0;
  for (int i = 0; i < nw; ++i) r += sh[i];
  return r;
}
__device__ __forceinline__ double block_max(double v, double *sh) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double r = sh[0];
  for (int i = 1; i < nw; ++i) r = fmax(r, sh[i]);
  return r;
}

// digamma: the 7-shift asymptotic series of the reference (src/Sample.cpp:87-97; rcgpar
// carries the same function for the RCG gradient).
__device__ __forceinline__ double digamma_ref(double x) {
  double result = 0.0;
  for (; x < 7.0; x += 1.0) result -= 1.0 / x;
  x -= 0.5;
  const double xx = 1.0 / x, xx2 = xx * xx, xx4 = xx2 * xx2;
  result += log(x) + (1. / 24.) * xx2 - (7.0 / 960.0) * xx4 + (31.0 / 8064.0) * xx4 * xx2 -
            (127.0 / 30720.0) * xx4 * xx4;
  return result;
}

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

// LDS bytes of the sweeps.  Pass A keeps {e_g, wc_g} pairs and 4 doubles per LUT slot, pass B
// keeps e_g, the column-sum accumulators and 2 doubles per LUT slot.
__host__ __device__ inline size_t pass_lds_bytes(bool glds, bool tlds, uint32_t G, uint32_t n_lut,
                                                 bool passA) {
  size_t b = 32 * sizeof(double);  // reduction scratch
  if (glds) b += 2 * ((size_t)G + 1) * sizeof(double);
  if (tlds) b += (size_t)(passA ? 4 : 2) * n_lut * sizeof(double);
  return b;
}

// ---------------------------------------------------------------------------------------
// O(G) "prep" helpers, all executed by ONE 1024-thread workgroup.
// Index G of e / ew is the sentinel slot: zeroed once at set-up and never written here.
// ---------------------------------------------------------------------------------------
// From (a, u): M = max u, e_g = exp(u_g - M), U = sum e, p0 = exp(a*logzi) and the pass-B
// table {x - p0, x*T - p0*logzi}, x = exp(a*T).
__device__ inline void prepB_block(Scalars *sc, double a, int G, int n_lut, const double *u,
                                   const double *lut, double *e, double *tabB, double *sh) {
  const int tid = threadIdx.x, nt = blockDim.x;
  double m = -INFINITY;
  for (int g = tid; g < G; g += nt) m = fmax(m, u[g]);
  const double M = block_max(m, sh);
  double su = 0.0;
  for (int g = tid; g < G; g += nt) {
    const double eg = exp(u[g] - M);
    e[g] = eg;
    su += eg;
  }
  const double U = block_sum(su, sh);
  const double logzi = sc->logzi;
  const double p0 = exp(a * logzi);
  for (int i = tid; i < n_lut; i += nt) {
    const double T = lut[i];
    const double x = exp(a * T);
    tabB[2 * i] = x - p0;
    tabB[2 * i + 1] = x * T - p0 * logzi;
  }
  if (tid == 0) {
    sc->M = M;
    sc->U = U;
    sc->p0 = p0;
  }
}

__global__ __launch_bounds__(1024) void k_prepB(Scalars *sc, int G, int n_lut, const double *u,
                                               const double *lut, double *e, double *tabB) {
  __shared__ double sh[32];
  if (sc->done) return;
  if (sc->flavor != 0) return;  // dense flavour needs no tables
  const double a = sc->a;
  __syncthreads();
  prepB_block(sc, a, G, n_lut, u, lut, e, tabB, sh);
}

// Gradient preparation: w_g = digamma(N_g) - 1 - u_g (the group part of rcgpar's
// mixt_negnatgrad step), {e_g, centred w_g} pairs and the pass-A table.
__global__ __launch_bounds__(1024) void k_prepA(Scalars *sc, int G, int n_lut, const double *N,
                                               const double *u, const double *lut, double *w,
                                               double *e, double2 *ew, double *tabA) {
  __shared__ double sh[32];
  if (sc->done) return;
  const int tid = threadIdx.x, nt = blockDim.x;
  const double a = sc->a, oma = 1.0 - a, logzi = sc->logzi;
  const int flavor = sc->flavor;
  if (flavor != 0) {  // dense: only w is needed
    for (int g = tid; g < G; g += nt) w[g] = digamma_ref(N[g]) - 1.0 - u[g];
    return;
  }
  double m = -INFINITY;
  for (int g = tid; g < G; g += nt) m = fmax(m, u[g]);
  const double M = block_max(m, sh);
  double su = 0.0, sv = 0.0;
  for (int g = tid; g < G; g += nt) {
    const double wg = digamma_ref(N[g]) - 1.0 - u[g];
    const double eg = exp(u[g] - M);
    w[g] = wg;
    e[g] = eg;
    su += eg;
    sv += eg * (oma * logzi + wg);
  }
  const double U = block_sum(su, sh);
  const double V1 = block_sum(sv, sh);
  const double kappa = V1 / U;  // centring constant: a per-EC shift leaves the variance unchanged
  double s1 = 0.0, s2 = 0.0;
  for (int g = tid; g < G; g += nt) {
    const double wcg = w[g] - kappa;
    const double eg = e[g];
    ew[g] = make_double2(eg, wcg);
    const double s0 = oma * logzi + wcg;
    s1 += eg * s0;
    s2 += eg * s0 * s0;
  }
  const double V1c = block_sum(s1, sh);
  const double V2c = block_sum(s2, sh);
  const double p0 = exp(a * logzi);
  for (int i = tid; i < n_lut; i += nt) {
    const double T = lut[i];
    const double x = exp(a * T);
    tabA[4 * i] = x - p0;
    tabA[4 * i + 1] = oma * (x * T - p0 * logzi);
    tabA[4 * i + 2] = oma * oma * (x * T * T - p0 * logzi * logzi);
    tabA[4 * i + 3] = 0.0;
  }
  if (tid == 0) {
    sc->M = M;
    sc->U = U;
    sc->p0 = p0;
    sc->V1c = V1c;
    sc->V2c = V2c;
  }
}

// Fletcher-Reeves step (rcgpar rcg_optl_mat: beta_FR, oldstep scaling, gamma += step) on
// the (a, u) state, followed by the pass-B preparation.
__global__ __launch_bounds__(1024) void k_step(Scalars *sc, int G, int n_lut, int n_partA,
                                              const double *partA, const double *w, double *u,
                                              double *os_u, double *step_u, const double *lut,
                                              double *e, double *tabB) {
  __shared__ double sh[32];
  if (sc->done) return;
  const int tid = threadIdx.x, nt = blockDim.x;
  const double a = sc->a, oldnorm = sc->oldnorm, bound = sc->bound;
  double os_a = sc->os_a;
  const int didreset = sc->didreset;
  double pn = 0.0;
  for (int i = tid; i < n_partA; i += nt) pn += partA[i];
  const double newnorm = block_sum(pn, sh);
  const double beta = newnorm / oldnorm;
  double step_a = 1.0 - a;
  if (didreset) {
    os_a *= 0.0;
  } else if (beta > 0) {
    os_a *= beta;
    step_a += os_a;
  }
  for (int g = tid; g < G; g += nt) {
    double osu = os_u[g], su = w[g];
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
  const double a_new = a + step_a;
  __syncthreads();
  if (tid == 0) {
    sc->a = a_new;
    sc->os_a = os_a;
    sc->step_a = step_a;
    sc->oldnorm = newnorm;
    sc->newnorm = newnorm;
    sc->beta = beta;
    sc->didreset = 0;
    sc->oldbound = bound;
  }
  if (sc->flavor == 0) prepB_block(sc, a_new, G, n_lut, u, lut, e, tabB, sh);
}

// ---------------------------------------------------------------------------------------
// Pass A (SELL): newnorm = sum_j Var_{q_j}(step_.j), q_j = softmax_g(a*L + u),
// step_gj = (1-a)*L_gj + w_g  (+ an irrelevant per-EC constant).
// One persistent 1024-thread workgroup per CU; its 16 wavefronts take slices round-robin.
// ---------------------------------------------------------------------------------------
struct AccA {
  double zs, t1, t2;
};
__device__ __forceinline__ void cellA(AccA &c, const double2 ew, const double *t) {
  const double xm = t[0], A1 = t[1], A2 = t[2];
  const double wx = ew.y * xm;
  c.zs += ew.x * xm;
  c.t1 += ew.x * (A1 + wx);
  c.t2 += ew.x * (A2 + ew.y * (2.0 * A1 + wx));
}

template <bool WIDE, bool GLDS, bool TLDS>
__global__ __launch_bounds__(kPassThreads) void k_passA(const Scalars *sc, SellDev S,
                                                       const double2 *ew_g, const double *tabA_g,
                                                       double *partA) {
  extern __shared__ __align__(16) unsigned char smem[];
  using R = Rec<WIDE>;
  if (sc->done) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const uint32_t G = S.n_groups, n_lut = S.n_lut;
  double *sh = reinterpret_cast<double *>(smem);
  double *p = sh + 32;
  const double2 *ew = ew_g;
  const double *tab = tabA_g;
  if (GLDS) {
    double2 *l = reinterpret_cast<double2 *>(p);
    p += 2 * ((size_t)G + 1);
    for (uint32_t g = tid; g <= G; g += kPassThreads) l[g] = ew_g[g];
    ew = l;
  }
  if (TLDS) {
    double *tl = p;
    for (uint32_t i = tid; i < 4 * n_lut; i += kPassThreads) tl[i] = tabA_g[i];
    tab = tl;
  }
  const double p0 = sc->p0, U = sc->U;
  const double zbase = p0 * U, b1 = p0 * sc->V1c, b2 = p0 * sc->V2c;
  double nn = 0.0;
  __syncthreads();

  const uint32_t n_sell = S.n_ecs - S.n_long;
  const uint32_t gw = blockIdx.x * (kPassThreads / 64) + (tid >> 6), nw = gridDim.x * (kPassThreads / 64);
  for (uint32_t s = gw; s < S.nslices; s += nw) {
    const uint32_t o0 = S.slice_off[s], len = S.slice_off[s + 1] - o0;
    const size_t base = (size_t)o0 * 64 + lane;
    AccA c = {0.0, 0.0, 0.0};
    uint32_t k = 0;
    for (; k + 4 <= len; k += 4) {
      const typename R::T r0 = R::load(S.rec, base + (size_t)k * 64);
      const typename R::T r1 = R::load(S.rec, base + (size_t)(k + 1) * 64);
      const typename R::T r2 = R::load(S.rec, base + (size_t)(k + 2) * 64);
      const typename R::T r3 = R::load(S.rec, base + (size_t)(k + 3) * 64);
      const double2 e0 = ew[R::grp(r0)], e1 = ew[R::grp(r1)], e2 = ew[R::grp(r2)], e3 = ew[R::grp(r3)];
      cellA(c, e0, tab + 4 * R::idx(r0));
      cellA(c, e1, tab + 4 * R::idx(r1));
      cellA(c, e2, tab + 4 * R::idx(r2));
      cellA(c, e3, tab + 4 * R::idx(r3));
    }
    for (; k < len; ++k) {
      const typename R::T r = R::load(S.rec, base + (size_t)k * 64);
      cellA(c, ew[R::grp(r)], tab + 4 * R::idx(r));
    }
    if (s * 64 + lane < n_sell) {
      const double iZ = 1.0 / (zbase + c.zs);
      const double S1 = (b1 + c.t1) * iZ, S2 = (b2 + c.t2) * iZ;
      nn += S2 - S1 * S1;
    }
  }
  // long ECs: the whole workgroup strides over one EC's cells
  for (uint32_t r = blockIdx.x; r < S.n_long; r += gridDim.x) {
    AccA c = {0.0, 0.0, 0.0};
    for (uint32_t k = S.long_ptr[r] + tid; k < S.long_ptr[r + 1]; k += kPassThreads) {
      const typename R::T rc = R::load(S.rec_long, k);
      cellA(c, ew[R::grp(rc)], tab + 4 * R::idx(rc));
    }
    const double zs = block_sum(c.zs, sh), t1 = block_sum(c.t1, sh), t2 = block_sum(c.t2, sh);
    if (tid == 0) {
      const double iZ = 1.0 / (zbase + zs);
      const double S1 = (b1 + t1) * iZ, S2 = (b2 + t2) * iZ;
      nn += S2 - S1 * S1;
    }
  }
  nn = block_sum(nn, sh);
  if (tid == 0) partA[blockIdx.x] = nn;
}

// ---------------------------------------------------------------------------------------
// Pass B (SELL): per EC Z_j (softmax denominator), r_j = c_j / Z_j, the ELBO data terms
// and the column sums A_g = sum_j r_j (x_gj - p0) accumulated in an LDS-private table
// (rcgpar logsumexp + update_N_k + ELBO_rcg_mat in one sweep).
// ---------------------------------------------------------------------------------------
template <bool WIDE, bool GLDS, bool TLDS>
__global__ __launch_bounds__(kPassThreads) void k_passB(const Scalars *sc, int cond_reset,
                                                       SellDev S, const double *e_g,
                                                       const double *tabB_g, double *partAcc,
                                                       double *partS, double *accGlobal) {
  extern __shared__ __align__(16) unsigned char smem[];
  using R = Rec<WIDE>;
  if (sc->done) return;
  if (cond_reset && !sc->reset_pending) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const uint32_t G = S.n_groups, n_lut = S.n_lut;
  double *sh = reinterpret_cast<double *>(smem);
  double *p = sh + 32;
  const double *e_l = e_g, *tab = tabB_g;
  double *acc = accGlobal;
  if (GLDS) {
    double *el = p;
    acc = p + (G + 1);
    p += 2 * ((size_t)G + 1);
    for (uint32_t g = tid; g <= G; g += kPassThreads) {
      el[g] = e_g[g];
      acc[g] = 0.0;
    }
    e_l = el;
  }
  if (TLDS) {
    double *tl = p;
    for (uint32_t i = tid; i < 2 * n_lut; i += kPassThreads) tl[i] = tabB_g[i];
    tab = tl;
  }
  const double p0 = sc->p0, U = sc->U, logzi = sc->logzi;
  const double zbase = p0 * U, hbase = p0 * logzi * U;
  double s_clogZ = 0.0, s_rH = 0.0, s_W = 0.0;
  __syncthreads();

  const uint32_t n_sell = S.n_ecs - S.n_long;
  const uint32_t gw = blockIdx.x * (kPassThreads / 64) + (tid >> 6), nw = gridDim.x * (kPassThreads / 64);
  for (uint32_t s = gw; s < S.nslices; s += nw) {
    const uint32_t o0 = S.slice_off[s], len = S.slice_off[s + 1] - o0;
    const size_t base = (size_t)o0 * 64 + lane;
    double zs = 0.0, hs = 0.0;
    uint32_t k = 0;
    for (; k + 4 <= len; k += 4) {
      const typename R::T r0 = R::load(S.rec, base + (size_t)k * 64);
      const typename R::T r1 = R::load(S.rec, base + (size_t)(k + 1) * 64);
      const typename R::T r2 = R::load(S.rec, base + (size_t)(k + 2) * 64);
      const typename R::T r3 = R::load(S.rec, base + (size_t)(k + 3) * 64);
      const double e0 = e_l[R::grp(r0)], e1 = e_l[R::grp(r1)], e2 = e_l[R::grp(r2)], e3 = e_l[R::grp(r3)];
      const double2 t0 = *reinterpret_cast<const double2 *>(tab + 2 * R::idx(r0));
      const double2 t1 = *reinterpret_cast<const double2 *>(tab + 2 * R::idx(r1));
      const double2 t2 = *reinterpret_cast<const double2 *>(tab + 2 * R::idx(r2));
      const double2 t3 = *reinterpret_cast<const double2 *>(tab + 2 * R::idx(r3));
      zs += e0 * t0.x; hs += e0 * t0.y;
      zs += e1 * t1.x; hs += e1 * t1.y;
      zs += e2 * t2.x; hs += e2 * t2.y;
      zs += e3 * t3.x; hs += e3 * t3.y;
    }
    for (; k < len; ++k) {
      const typename R::T r = R::load(S.rec, base + (size_t)k * 64);
      const double eg = e_l[R::grp(r)];
      const double2 t = *reinterpret_cast<const double2 *>(tab + 2 * R::idx(r));
      zs += eg * t.x;
      hs += eg * t.y;
    }
    const uint32_t q = s * 64 + lane;
    if (q < n_sell) {
      const double c = S.cvec[S.n_long + q];
      if (c != 0.0) {
        const double Z = zbase + zs, H = hbase + hs;
        const double rj = c / Z;
        s_clogZ += c * log(Z);
        s_rH += rj * H;
        s_W += rj;
        for (uint32_t kk = 0; kk < len; ++kk) {
          const typename R::T r = R::load(S.rec, base + (size_t)kk * 64);
          atomicAdd(&acc[R::grp(r)], rj * tab[2 * R::idx(r)]);
        }
      }
    }
  }
  for (uint32_t r = blockIdx.x; r < S.n_long; r += gridDim.x) {
    double zs = 0.0, hs = 0.0;
    for (uint32_t k = S.long_ptr[r] + tid; k < S.long_ptr[r + 1]; k += kPassThreads) {
      const typename R::T rc = R::load(S.rec_long, k);
      const double eg = e_l[R::grp(rc)];
      zs += eg * tab[2 * R::idx(rc)];
      hs += eg * tab[2 * R::idx(rc) + 1];
    }
    zs = block_sum(zs, sh);
    hs = block_sum(hs, sh);
    const double c = S.cvec[r];
    if (c != 0.0) {
      const double Z = zbase + zs, H = hbase + hs;
      const double rj = c / Z;
      if (tid == 0) {
        s_clogZ += c * log(Z);
        s_rH += rj * H;
        s_W += rj;
      }
      for (uint32_t k = S.long_ptr[r] + tid; k < S.long_ptr[r + 1]; k += kPassThreads) {
        const typename R::T rc = R::load(S.rec_long, k);
        atomicAdd(&acc[R::grp(rc)], rj * tab[2 * R::idx(rc)]);
      }
    }
  }
  s_clogZ = block_sum(s_clogZ, sh);
  s_rH = block_sum(s_rH, sh);
  s_W = block_sum(s_W, sh);
  if (tid == 0) {
    partS[4 * blockIdx.x + 0] = s_clogZ;
    partS[4 * blockIdx.x + 1] = s_rH;
    partS[4 * blockIdx.x + 2] = s_W;
    partS[4 * blockIdx.x + 3] = 0.0;
  }
  if (GLDS) {
    __syncthreads();
    double *dst = partAcc + (size_t)blockIdx.x * G;
    for (uint32_t g = tid; g < G; g += kPassThreads) dst[g] = acc[g];
  }
}

// Column sums across workgroups, fixed order.
__global__ __launch_bounds__(256) void k_redB(const Scalars *sc, int cond_reset, int G, int nblk,
                                             const double *partAcc, double *Acc) {
  if (sc->done) return;
  if (cond_reset && !sc->reset_pending) return;
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += partAcc[(size_t)b * G + g];
  Acc[g] = s;
}

// ---------------------------------------------------------------------------------------
// End of pass B: N_g, ELBO (rcgpar ELBO_rcg_mat + bound_const), the bound < oldbound
// steepest-descent retry (revert_step) and the convergence test, all on the device.
//   mode 2: initial update_N_k only;  mode 0: first evaluation of an iteration;
//   mode 1: re-evaluation after a reset (runs only when reset_pending).
// ---------------------------------------------------------------------------------------
struct TraceDev {
  double *bound, *newnorm, *beta, *theta;
  int32_t *didreset;
};

__global__ __launch_bounds__(1024) void k_finB(Scalars *sc, int mode, int G, int n_lut, int nblk,
                                              const double *partS, const double *Acc,
                                              const double *alpha0, double *u,
                                              double *os_u, const double *step_u,
                                              const double *lut, double *e, double *tabB,
                                              double *Nc, double *N, TraceDev tr) {
  __shared__ double sh[32];
  if (sc->done) return;
  if (mode == 1 && !sc->reset_pending) return;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int flavor = sc->flavor;
  const double a = sc->a, M = sc->M, p0 = sc->p0, oldbound = sc->oldbound;
  const double beta = sc->beta, tol = sc->tol, csum = sc->csum;
  double p1 = 0.0, p2 = 0.0, p3 = 0.0;
  for (int b = tid; b < nblk; b += nt) {
    p1 += partS[4 * b];
    p2 += partS[4 * b + 1];
    p3 += partS[4 * b + 2];
  }
  const double s_clogZ = block_sum(p1, sh);
  const double s_rH = block_sum(p2, sh);
  const double W = block_sum(p3, sh);
  double lg = 0.0, mu = 0.0;
  for (int g = tid; g < G; g += nt) {
    double nc;
    if (flavor == 0) {
      nc = e[g] * (p0 * W + Acc[g]);
      mu += (M - u[g]) * nc;
    } else {
      nc = Acc[g];
    }
    const double n = alpha0[g] + nc;
    Nc[g] = nc;
    N[g] = n;
    lg += lgamma(n);
  }
  lg = block_sum(lg, sh);
  mu = block_sum(mu, sh);
  if (mode == 2) return;
  const double coef = (flavor == 0) ? (1.0 - a) : 1.0;
  const double bound = sc->bound_const + s_clogZ + coef * s_rH + mu + lg;
  int didreset = sc->didreset;
  __syncthreads();
  if (mode == 0 && bound < oldbound) {
    // bad step: revert to steepest descent (gamma += oldm; gamma -= oldstep) and re-evaluate
    double a2 = a;
    if (beta > 0) {
      a2 = a - sc->os_a;
      for (int g = tid; g < G; g += nt) u[g] -= os_u[g];
    }
    __syncthreads();
    if (tid == 0) {
      sc->a = a2;
      sc->didreset = 1;
      sc->reset_pending = 1;
      sc->bound = bound;
    }
    if (flavor == 0) prepB_block(sc, a2, G, n_lut, u, lut, e, tabB, sh);
    return;
  }
  if (mode == 0) {
    // oldstep = step
    for (int g = tid; g < G; g += nt) os_u[g] = step_u[g];
  }
  const int it = sc->iter;
  if (it < sc->trace_theta && tr.theta) {
    for (int g = tid; g < G; g += nt) tr.theta[(size_t)it * G + g] = Nc[g] / csum;
  }
  __syncthreads();
  if (tid == 0) {
    if (mode == 0) sc->os_a = sc->step_a;
    sc->bound = bound;
    sc->reset_pending = 0;
    if (it < kMaxTrace) {
      tr.bound[it] = bound;
      tr.newnorm[it] = sc->newnorm;
      tr.beta[it] = beta;
      tr.didreset[it] = didreset;
    }
    int done = 0;
    if (!sc->fixed_iters && (bound - oldbound < tol) && !didreset) done = 1;
    const int nit = it + 1;
    if (nit >= sc->max_iters) done = 1;
    sc->iter = nit;
    sc->done = done;
  }
}

// ---------------------------------------------------------------------------------------
// Solve set-up: c_j = exp(logc_j) (or the bootstrap counts) gathered into the permuted EC
// order, sum of counts, bound constant (rcgpar calc_bound_const), initial gamma = log(1/G).
// perm == nullptr: identity (dense flavour).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cvec_from_logc(const double *logc, const uint32_t *perm,
                                                       uint32_t E, double *cvec, double *part) {
  __shared__ double sh[32];
  double s = 0.0;
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < E; j += gridDim.x * blockDim.x) {
    const double c = exp(logc[perm ? perm[j] : j]);
    cvec[j] = c;
    s += c;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_cvec_from_counts(const uint32_t *cnt, const uint32_t *perm,
                                                         uint32_t E, double *cvec, double *part) {
  __shared__ double sh[32];
  double s = 0.0;
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < E; j += gridDim.x * blockDim.x) {
    const double c = (double)cnt[perm ? perm[j] : j];
    cvec[j] = c;
    s += c;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(1024) void k_init_state(Scalars *sc, int G, int npart, const double *part,
                                                    const double *alpha0, double *u, double *os_u,
                                                    double *step_u, double tol, int max_iters,
                                                    int fixed_iters, int trace_theta, int flavor,
                                                    double logzi, double init_bound) {
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
    *sc = z;
  }
}

// ---------------------------------------------------------------------------------------
// Dense-L kernels.  L is kept EC-major on the device (Lt[j*G + g]) so that a wavefront
// streams one EC's G values with coalesced loads; lane l owns groups l, l+64, ... and
// keeps their u_g / w_g / column-sum accumulators in registers (no atomics).
// ---------------------------------------------------------------------------------------
template <int NREG>
__global__ __launch_bounds__(256) void k_dense_passA(const Scalars *sc, const double *Lt, int G,
                                                    uint32_t E, const double *u, const double *w,
                                                    double *partA) {
  __shared__ double sh[32];
  if (sc->done) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t gw = blockIdx.x * 4 + wv, nw = gridDim.x * 4;
  const double a = sc->a, oma = 1.0 - a;
  double uu[NREG], ww[NREG];
#pragma unroll
  for (int i = 0; i < NREG; ++i) {
    const int g = lane + 64 * i;
    uu[i] = g < G ? u[g] : 0.0;
    ww[i] = g < G ? w[g] : 0.0;
  }
  double nn = 0.0;
  for (uint32_t j = gw; j < E; j += nw) {
    const double *row = Lt + (size_t)j * G;
    double pz[NREG], s[NREG];
    double m = -INFINITY;
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      const int g = lane + 64 * i;
      const double x = g < G ? row[g] : 0.0;
      pz[i] = g < G ? a * x + uu[i] : -INFINITY;
      s[i] = oma * x + ww[i];
      m = fmax(m, pz[i]);
    }
    m = wave_max(m);
    double Z = 0.0, S1 = 0.0;
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      const int g = lane + 64 * i;
      const double pe = g < G ? exp(pz[i] - m) : 0.0;
      pz[i] = pe;
      Z += pe;
      S1 += pe * s[i];
    }
    Z = wave_sum(Z);
    S1 = wave_sum(S1);
    const double iZ = 1.0 / Z, sbar = S1 * iZ;
    double v = 0.0;
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      const double d = s[i] - sbar;
      v += pz[i] * d * d;
    }
    v = wave_sum(v);
    nn += v * iZ;
  }
  // every lane of a wave holds the same nn; take lane 0 of each wave
  double t = (lane == 0) ? nn : 0.0;
  t = block_sum(t, sh);
  if (threadIdx.x == 0) partA[blockIdx.x] = t;
}

template <int NREG>
__global__ __launch_bounds__(256) void k_dense_passB(const Scalars *sc, int cond_reset,
                                                    const double *Lt, int G, uint32_t E,
                                                    const double *cvec, const double *u,
                                                    double *partAcc, double *partS) {
  extern __shared__ __align__(16) unsigned char smem[];
  double *sh = reinterpret_cast<double *>(smem);
  double *accl = sh + 32;  // [4][G]
  if (sc->done) return;
  if (cond_reset && !sc->reset_pending) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t gw = blockIdx.x * 4 + wv, nw = gridDim.x * 4;
  const double a = sc->a;
  double uu[NREG], acc[NREG];
#pragma unroll
  for (int i = 0; i < NREG; ++i) {
    const int g = lane + 64 * i;
    uu[i] = g < G ? u[g] : 0.0;
    acc[i] = 0.0;
  }
  double s_clogZ = 0.0, s_rH = 0.0;
  for (uint32_t j = gw; j < E; j += nw) {
    const double *row = Lt + (size_t)j * G;
    double x[NREG], pz[NREG];
    double m = -INFINITY;
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      const int g = lane + 64 * i;
      x[i] = g < G ? row[g] : 0.0;
      pz[i] = g < G ? a * x[i] + uu[i] : -INFINITY;
      m = fmax(m, pz[i]);
    }
    m = wave_max(m);
    double Z = 0.0, hs = 0.0;
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      const int g = lane + 64 * i;
      const double y = pz[i] - m;
      const double pe = g < G ? exp(y) : 0.0;
      hs += g < G ? pe * (x[i] - y) : 0.0;
      pz[i] = pe;
      Z += pe;
    }
    Z = wave_sum(Z);
    hs = wave_sum(hs);
    const double c = cvec[j];
    if (c != 0.0) {
      const double rj = c / Z;
      s_clogZ += c * log(Z);
      s_rH += rj * hs;
#pragma unroll
      for (int i = 0; i < NREG; ++i) acc[i] += rj * pz[i];
    }
  }
#pragma unroll
  for (int i = 0; i < NREG; ++i) {
    const int g = lane + 64 * i;
    if (g < G) accl[wv * G + g] = acc[i];
  }
  double t1 = (lane == 0) ? s_clogZ : 0.0, t2 = (lane == 0) ? s_rH : 0.0;
  t1 = block_sum(t1, sh);
  t2 = block_sum(t2, sh);
  if (threadIdx.x == 0) {
    partS[4 * blockIdx.x + 0] = t1;
    partS[4 * blockIdx.x + 1] = t2;
    partS[4 * blockIdx.x + 2] = 0.0;
    partS[4 * blockIdx.x + 3] = 0.0;
  }
  __syncthreads();
  double *dst = partAcc + (size_t)blockIdx.x * G;
  for (int g = threadIdx.x; g < G; g += blockDim.x)
    dst[g] = ((accl[g] + accl[G + g]) + accl[2 * G + g]) + accl[3 * G + g];
}

// [G][E] (ld) -> [E][G] transpose through LDS, 64 x 64 tiles, 256 threads.
__global__ __launch_bounds__(256) void k_transpose(const double *src, size_t ld, int G, uint32_t E,
                                                  double *dst) {
  __shared__ double tile[64][65];
  const uint32_t j0 = blockIdx.x * 64;
  const int g0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int g = g0 + r;
    const uint32_t j = j0 + tx;
    tile[r][tx] = (g < G && j < E) ? src[(size_t)g * ld + j] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const uint32_t j = j0 + r;
    const int g = g0 + tx;
    if (g < G && j < E) dst[(size_t)j * G + g] = tile[tx][r];
  }
}

// ---------------------------------------------------------------------------------------
// gamma materialisation (K6): gamma(g, j) = a*L(g, j) + u_g - lse_j, rows = groups, columns in
// the ORIGINAL EC order.  With (a, u, lse) = (1, 0, none) the same kernels expand the resident
// likelihood.  Utility kernels, not on the timed path.
// ---------------------------------------------------------------------------------------
template <bool WIDE>
__global__ __launch_bounds__(256) void k_lse_sell(SellDev S, double a, double logzi, const double *u,
                                                 const double *lut, double *lse /*original order*/) {
  using R = Rec<WIDE>;
  __shared__ double sh[32];
  const int tid = threadIdx.x;
  double m = -INFINITY;
  for (uint32_t g = tid; g < S.n_groups; g += blockDim.x) m = fmax(m, u[g]);
  const double M = block_max(m, sh);
  double su = 0.0;
  for (uint32_t g = tid; g < S.n_groups; g += blockDim.x) su += exp(u[g] - M);
  const double U = block_sum(su, sh);
  const double p0 = exp(a * logzi);
  for (uint32_t p = blockIdx.x * blockDim.x + tid; p < S.n_ecs; p += gridDim.x * blockDim.x) {
    double zs = 0.0;
    for_each_cell<WIDE>(S, p, [&](typename R::T r) {
      zs += exp(u[R::grp(r)] - M) * (exp(a * lut[R::idx(r)]) - p0);
    });
    lse[S.perm[p]] = M + log(p0 * U + zs);
  }
}

__global__ __launch_bounds__(256) void k_gamma_fill(double *out, size_t ld, int g_begin, int g_end,
                                                   uint32_t E, double a, double logzi,
                                                   const double *u, const double *lse) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= E) return;
  const double l = lse ? lse[j] : 0.0;
  for (int g = g_begin; g < g_end; ++g)
    out[(size_t)(g - g_begin) * ld + j] = a * logzi + u[g] - l;
}

template <bool WIDE>
__global__ __launch_bounds__(256) void k_gamma_scatter(SellDev S, double *out, size_t ld, int g_begin,
                                                      int g_end, double a, const double *u,
                                                      const double *lut, const double *lse) {
  using R = Rec<WIDE>;
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= S.n_ecs) return;
  const uint32_t j = S.perm[p];
  const double l = lse ? lse[j] : 0.0;
  for_each_cell<WIDE>(S, p, [&](typename R::T r) {
    const int g = (int)R::grp(r);
    if (g >= g_begin && g < g_end)
      out[(size_t)(g - g_begin) * ld + j] = a * lut[R::idx(r)] + u[g] - l;
  });
}

// dense flavour: gamma from Lt (EC-major) -> rows = groups slab [g_begin, g_end)
__global__ __launch_bounds__(256) void k_gamma_dense(const double *Lt, int G, uint32_t E, double a,
                                                    const double *u, int sub_lse, double *out,
                                                    size_t ld, int g_begin, int g_end) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= E) return;
  const double *row = Lt + (size_t)j * G;
  double lse = 0.0;
  if (sub_lse) {
    double m = -INFINITY;
    for (int g = 0; g < G; ++g) m = fmax(m, a * row[g] + u[g]);
    double Z = 0.0;
    for (int g = 0; g < G; ++g) Z += exp(a * row[g] + u[g] - m);
    lse = m + log(Z);
  }
  for (int g = g_begin; g < g_end; ++g) out[(size_t)(g - g_begin) * ld + j] = a * row[g] + u[g] - lse;
}

}  // namespace msw
