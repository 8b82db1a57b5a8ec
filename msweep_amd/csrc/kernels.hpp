// kernels.hpp -- hand-written gfx950 kernels of the RCG / EM abundance-estimation loop.
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
// device-resident CSR-of-ECs likelihood
// ---------------------------------------------------------------------------------------
struct CsrDev {
  const uint32_t *rowptr;    // [E+1]
  const uint32_t *rec;       // narrow: [nnz] (lutidx << 16 | grp); wide: [2*nnz] {grp, lutidx}
  const uint32_t *tile_row;  // [ntiles+1] first row of each tile
  const double *cvec;        // [E] EC multiplicities as fp64
  uint32_t ntiles;
  uint32_t n_ecs;
  uint32_t n_groups;
  uint32_t n_lut;
};

constexpr int kTileRows = kPassThreads;  // one row per thread
constexpr int kTileCap = 8192;           // staged nz records per tile (32 KiB narrow)

template <bool WIDE>
struct Rec;
template <>
struct Rec<false> {
  using T = uint32_t;
  static __device__ __forceinline__ T load(const uint32_t *p, uint32_t i) { return p[i]; }
  static __device__ __forceinline__ uint32_t grp(T r) { return r & 0xffffu; }
  static __device__ __forceinline__ uint32_t idx(T r) { return r >> 16; }
};
template <>
struct Rec<true> {
  using T = uint2;
  static __device__ __forceinline__ T load(const uint32_t *p, uint32_t i) {
    return reinterpret_cast<const uint2 *>(p)[i];
  }
  static __device__ __forceinline__ uint32_t grp(T r) { return r.x; }
  static __device__ __forceinline__ uint32_t idx(T r) { return r.y; }
};

// LDS carve-up shared by pass A and pass B.  nvec = number of G-length fp64 vectors kept in
// LDS (2 in both passes), ntab = doubles per LUT slot (3 in pass A, 2 in pass B).
__host__ __device__ inline size_t pass_lds_bytes(bool wide, bool glds, bool tlds, uint32_t G,
                                                 uint32_t n_lut, int ntab) {
  size_t b = 32 * sizeof(double);                      // reduction scratch
  if (glds) b += 2 * (size_t)G * sizeof(double);
  if (tlds) b += (size_t)ntab * n_lut * sizeof(double);
  b += (kTileRows + 1 + 3) / 4 * 4 * sizeof(uint32_t);  // row pointers of the tile
  b += (size_t)kTileCap * (wide ? 8 : 4);              // staged records
  return b;
}

// ---------------------------------------------------------------------------------------
// O(G) "prep" helpers, all executed by ONE 1024-thread workgroup.
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
// mixt_negnatgrad step), e_g, the centred w and the pass-A table.
__global__ __launch_bounds__(1024) void k_prepA(Scalars *sc, int G, int n_lut, const double *N,
                                               const double *u, const double *lut, double *w,
                                               double *e, double *wc, double *tabA) {
  __shared__ double sh[32];
  if (sc->done) return;
  const int tid = threadIdx.x, nt = blockDim.x;
  const double a = sc->a, oma = 1.0 - a, logzi = sc->logzi;
  const int flavor = sc->flavor;
  if (flavor != 0) {  // dense: only w is needed
    for (int g = tid; g < G; g += nt) {
      const double wg = digamma_ref(N[g]) - 1.0 - u[g];
      w[g] = wg;
      wc[g] = wg;
    }
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
    wc[g] = wcg;
    const double s0 = oma * logzi + wcg;
    const double eg = e[g];
    s1 += eg * s0;
    s2 += eg * s0 * s0;
  }
  const double V1c = block_sum(s1, sh);
  const double V2c = block_sum(s2, sh);
  const double p0 = exp(a * logzi);
  for (int i = tid; i < n_lut; i += nt) {
    const double T = lut[i];
    const double x = exp(a * T);
    tabA[3 * i] = x - p0;
    tabA[3 * i + 1] = oma * (x * T - p0 * logzi);
    tabA[3 * i + 2] = oma * oma * (x * T * T - p0 * logzi * logzi);
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
// Pass A (CSR): newnorm = sum_j Var_{q_j}(step_.j), q_j = softmax_g(a*L + u),
// step_gj = (1-a)*L_gj + w_g  (+ an irrelevant per-EC constant).
// One persistent 1024-thread workgroup per CU; a tile = up to 1024 consecutive ECs whose
// records are staged through LDS with coalesced loads; one thread per EC.
// ---------------------------------------------------------------------------------------
template <bool WIDE, bool GLDS, bool TLDS>
__global__ __launch_bounds__(kPassThreads) void k_passA(const Scalars *sc, CsrDev S,
                                                       const double *e_g, const double *wc_g,
                                                       const double *tabA_g, double *partA) {
  extern __shared__ __align__(16) unsigned char smem[];
  using R = Rec<WIDE>;
  if (sc->done) return;
  const int tid = threadIdx.x;
  const uint32_t G = S.n_groups, n_lut = S.n_lut;
  double *sh = reinterpret_cast<double *>(smem);
  double *p = sh + 32;
  const double *e_l = e_g, *wc_l = wc_g, *tab = tabA_g;
  if (GLDS) {
    double *el = p, *wl = p + G;
    p += 2 * (size_t)G;
    for (uint32_t g = tid; g < G; g += kPassThreads) {
      el[g] = e_g[g];
      wl[g] = wc_g[g];
    }
    e_l = el;
    wc_l = wl;
  }
  if (TLDS) {
    double *tl = p;
    p += 3 * (size_t)n_lut;
    for (uint32_t i = tid; i < 3 * n_lut; i += kPassThreads) tl[i] = tabA_g[i];
    tab = tl;
  }
  uint32_t *rp = reinterpret_cast<uint32_t *>(p);
  typename R::T *nzl = reinterpret_cast<typename R::T *>(rp + (kTileRows + 1 + 3) / 4 * 4);
  const double p0 = sc->p0, U = sc->U;
  const double zbase = p0 * U, b1 = p0 * sc->V1c, b2 = p0 * sc->V2c;
  double nn = 0.0;
  __syncthreads();

  for (uint32_t t = blockIdx.x; t < S.ntiles; t += gridDim.x) {
    const uint32_t rs = S.tile_row[t], nr = S.tile_row[t + 1] - rs;
    if ((uint32_t)tid < nr) rp[tid] = S.rowptr[rs + tid];
    if (tid == 0) rp[nr] = S.rowptr[rs + nr];
    __syncthreads();
    const uint32_t n0 = rp[0], nn_t = rp[nr] - n0;
    if (nn_t <= (uint32_t)kTileCap) {
      for (uint32_t i = tid; i < nn_t; i += kPassThreads) nzl[i] = R::load(S.rec, n0 + i);
      __syncthreads();
      if ((uint32_t)tid < nr) {
        const uint32_t kb = rp[tid] - n0, ke = rp[tid + 1] - n0;
        double zs = 0.0, t1 = 0.0, t2 = 0.0;
        for (uint32_t k = kb; k < ke; ++k) {
          const typename R::T r = nzl[k];
          const uint32_t g = R::grp(r), i = R::idx(r);
          const double eg = e_l[g], wg = wc_l[g];
          const double xm = tab[3 * i], A1 = tab[3 * i + 1], A2 = tab[3 * i + 2];
          const double wx = wg * xm;
          zs += eg * xm;
          t1 += eg * (A1 + wx);
          t2 += eg * (A2 + wg * (2.0 * A1 + wx));
        }
        const double iZ = 1.0 / (zbase + zs);
        const double S1 = (b1 + t1) * iZ, S2 = (b2 + t2) * iZ;
        nn += S2 - S1 * S1;
      }
    } else {
      // one long EC: the whole workgroup strides over its records straight from HBM
      double zs = 0.0, t1 = 0.0, t2 = 0.0;
      for (uint32_t k = tid; k < nn_t; k += kPassThreads) {
        const typename R::T r = R::load(S.rec, n0 + k);
        const uint32_t g = R::grp(r), i = R::idx(r);
        const double eg = e_l[g], wg = wc_l[g];
        const double xm = tab[3 * i], A1 = tab[3 * i + 1], A2 = tab[3 * i + 2];
        const double wx = wg * xm;
        zs += eg * xm;
        t1 += eg * (A1 + wx);
        t2 += eg * (A2 + wg * (2.0 * A1 + wx));
      }
      zs = block_sum(zs, sh);
      t1 = block_sum(t1, sh);
      t2 = block_sum(t2, sh);
      if (tid == 0) {
        const double iZ = 1.0 / (zbase + zs);
        const double S1 = (b1 + t1) * iZ, S2 = (b2 + t2) * iZ;
        nn += S2 - S1 * S1;
      }
    }
    __syncthreads();
  }
  nn = block_sum(nn, sh);
  if (tid == 0) partA[blockIdx.x] = nn;
}

// ---------------------------------------------------------------------------------------
// Pass B (CSR): per EC Z_j (softmax denominator), r_j = c_j / Z_j, the ELBO data terms
// and the column sums A_g = sum_j r_j (x_gj - p0) accumulated in an LDS-private table
// (rcgpar logsumexp + update_N_k + ELBO_rcg_mat in one sweep).
// ---------------------------------------------------------------------------------------
template <bool WIDE, bool GLDS, bool TLDS>
__global__ __launch_bounds__(kPassThreads) void k_passB(const Scalars *sc, int cond_reset,
                                                       CsrDev S, const double *e_g,
                                                       const double *tabB_g, double *partAcc,
                                                       double *partS, double *accGlobal) {
  extern __shared__ __align__(16) unsigned char smem[];
  using R = Rec<WIDE>;
  if (sc->done) return;
  if (cond_reset && !sc->reset_pending) return;
  const int tid = threadIdx.x;
  const uint32_t G = S.n_groups, n_lut = S.n_lut;
  double *sh = reinterpret_cast<double *>(smem);
  double *p = sh + 32;
  const double *e_l = e_g, *tab = tabB_g;
  double *acc = accGlobal;
  if (GLDS) {
    double *el = p;
    acc = p + G;
    p += 2 * (size_t)G;
    for (uint32_t g = tid; g < G; g += kPassThreads) {
      el[g] = e_g[g];
      acc[g] = 0.0;
    }
    e_l = el;
  }
  if (TLDS) {
    double *tl = p;
    p += 2 * (size_t)n_lut;
    for (uint32_t i = tid; i < 2 * n_lut; i += kPassThreads) tl[i] = tabB_g[i];
    tab = tl;
  }
  uint32_t *rp = reinterpret_cast<uint32_t *>(p);
  typename R::T *nzl = reinterpret_cast<typename R::T *>(rp + (kTileRows + 1 + 3) / 4 * 4);
  const double p0 = sc->p0, U = sc->U, logzi = sc->logzi;
  const double zbase = p0 * U, hbase = p0 * logzi * U;
  double s_clogZ = 0.0, s_rH = 0.0, s_W = 0.0;
  __syncthreads();

  for (uint32_t t = blockIdx.x; t < S.ntiles; t += gridDim.x) {
    const uint32_t rs = S.tile_row[t], nr = S.tile_row[t + 1] - rs;
    if ((uint32_t)tid < nr) rp[tid] = S.rowptr[rs + tid];
    if (tid == 0) rp[nr] = S.rowptr[rs + nr];
    __syncthreads();
    const uint32_t n0 = rp[0], nn_t = rp[nr] - n0;
    if (nn_t <= (uint32_t)kTileCap) {
      for (uint32_t i = tid; i < nn_t; i += kPassThreads) nzl[i] = R::load(S.rec, n0 + i);
      __syncthreads();
      if ((uint32_t)tid < nr) {
        const uint32_t kb = rp[tid] - n0, ke = rp[tid + 1] - n0;
        double zs = 0.0, hs = 0.0;
        for (uint32_t k = kb; k < ke; ++k) {
          const typename R::T r = nzl[k];
          const double eg = e_l[R::grp(r)];
          const uint32_t i = R::idx(r);
          zs += eg * tab[2 * i];
          hs += eg * tab[2 * i + 1];
        }
        const double Z = zbase + zs, H = hbase + hs;
        const double c = S.cvec[rs + tid];
        if (c != 0.0) {
          const double rj = c / Z;
          s_clogZ += c * log(Z);
          s_rH += rj * H;
          s_W += rj;
          for (uint32_t k = kb; k < ke; ++k) {
            const typename R::T r = nzl[k];
            atomicAdd(&acc[R::grp(r)], rj * tab[2 * R::idx(r)]);
          }
        }
      }
    } else {
      double zs = 0.0, hs = 0.0;
      for (uint32_t k = tid; k < nn_t; k += kPassThreads) {
        const typename R::T r = R::load(S.rec, n0 + k);
        const double eg = e_l[R::grp(r)];
        const uint32_t i = R::idx(r);
        zs += eg * tab[2 * i];
        hs += eg * tab[2 * i + 1];
      }
      zs = block_sum(zs, sh);
      hs = block_sum(hs, sh);
      const double Z = zbase + zs, H = hbase + hs;
      const double c = S.cvec[rs];
      if (c != 0.0) {
        const double rj = c / Z;
        if (tid == 0) {
          s_clogZ += c * log(Z);
          s_rH += rj * H;
          s_W += rj;
        }
        for (uint32_t k = tid; k < nn_t; k += kPassThreads) {
          const typename R::T r = R::load(S.rec, n0 + k);
          atomicAdd(&acc[R::grp(r)], rj * tab[2 * R::idx(r)]);
        }
      }
    }
    __syncthreads();
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
// Solve set-up: c_j = exp(logc_j) (or the bootstrap counts), sum of counts, bound constant
// (rcgpar calc_bound_const), initial state gamma = log(1/G).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cvec_from_logc(const double *logc, uint32_t E, double *cvec,
                                                       double *part) {
  __shared__ double sh[32];
  double s = 0.0;
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < E; j += gridDim.x * blockDim.x) {
    const double c = exp(logc[j]);
    cvec[j] = c;
    s += c;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_cvec_from_counts(const uint32_t *cnt, uint32_t E, double *cvec,
                                                         double *part) {
  __shared__ double sh[32];
  double s = 0.0;
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < E; j += gridDim.x * blockDim.x) {
    const double c = (double)cnt[j];
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
// gamma materialisation (K6): gamma(g, j) = a*L(g, j) + u_g - lse_j, rows = groups.
// With (a, u, sub_lse) = (1, 0, false) the same kernels expand the resident likelihood.
// ---------------------------------------------------------------------------------------
template <bool WIDE>
__global__ __launch_bounds__(256) void k_lse_csr(CsrDev S, double a, double logzi, const double *u,
                                                const double *lut, double *lse) {
  // straightforward one-thread-per-EC evaluation (not on the timed path)
  using R = Rec<WIDE>;
  __shared__ double sh[32];
  const int tid = threadIdx.x;
  // M, U recomputed per block (G is small)
  double m = -INFINITY;
  for (uint32_t g = tid; g < S.n_groups; g += blockDim.x) m = fmax(m, u[g]);
  const double M = block_max(m, sh);
  double su = 0.0;
  for (uint32_t g = tid; g < S.n_groups; g += blockDim.x) su += exp(u[g] - M);
  const double U = block_sum(su, sh);
  const double p0 = exp(a * logzi);
  for (uint32_t j = blockIdx.x * blockDim.x + tid; j < S.n_ecs; j += gridDim.x * blockDim.x) {
    double zs = 0.0;
    for (uint32_t k = S.rowptr[j]; k < S.rowptr[j + 1]; ++k) {
      const typename R::T r = R::load(S.rec, k);
      zs += exp(u[R::grp(r)] - M) * (exp(a * lut[R::idx(r)]) - p0);
    }
    lse[j] = M + log(p0 * U + zs);
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
__global__ __launch_bounds__(256) void k_gamma_scatter(CsrDev S, double *out, size_t ld, int g_begin,
                                                      int g_end, double a, const double *u,
                                                      const double *lut, const double *lse) {
  using R = Rec<WIDE>;
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= S.n_ecs) return;
  const double l = lse ? lse[j] : 0.0;
  for (uint32_t k = S.rowptr[j]; k < S.rowptr[j + 1]; ++k) {
    const typename R::T r = R::load(S.rec, k);
    const int g = (int)R::grp(r);
    if (g >= g_begin && g < g_end)
      out[(size_t)(g - g_begin) * ld + j] = a * lut[R::idx(r)] + u[g] - l;
  }
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
