// sweep_kernels.hpp -- the two HBM-streaming sweeps of one RCG iteration over the SELL-64
// likelihood: pass A (natural-gradient norm) and pass B (softmax / column sums / ELBO).
//
// Both kernels: one persistent 1024-thread workgroup per CU; its 16 wavefronts take slices
// round-robin.  A slice of up to kRegCells cells per EC is held in registers: while slice s is
// being processed the records of the wave's next slice are already in flight (the record stream
// is the only HBM traffic; everything else is gathered from LDS).  Slice bounds are wave-uniform
// and kept in SGPRs (readfirstlane) so the cell loops are scalar branches, not exec-mask loops.
#pragma once
#include <type_traits>

#include "device_util.hpp"
#include "sell.hpp"

namespace msw {

constexpr int kRegCells = 16;  // cells per EC a wave keeps in registers (longer slices stream)

__device__ __forceinline__ uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// Issue the loads of one slice (<= kRegCells cells per EC, even count) into registers.
template <bool WIDE>
__device__ __forceinline__ void load_slice(const uint32_t *rec, size_t base, uint32_t len,
                                           typename Rec<WIDE>::T (&r)[kRegCells]) {
#pragma unroll
  for (int k = 0; k < kRegCells; k += 2) {
    if ((uint32_t)k < len) {
      r[k] = Rec<WIDE>::load(rec, base + (size_t)k * 64);
      r[k + 1] = Rec<WIDE>::load(rec, base + (size_t)(k + 1) * 64);
    }
  }
}

// ---------------------------------------------------------------------------------------
// Pass A: newnorm = sum_j Var_{q_j}(step_.j), q_j = softmax_g(a*L + u),
// step_gj = (1-a)*L_gj + w_g  (+ an irrelevant per-EC constant).
// ---------------------------------------------------------------------------------------
struct AccA {
  double zs, t1, t2;
};
struct CstA {
  double p0, oma, oma2, p0l, p0l2;  // p0, (1-a), (1-a)^2, p0*logzi, p0*logzi^2
};
__device__ __forceinline__ void cellA(AccA &c, const CstA &k, const double e, const double w,
                                      const double x, const double T) {
  const double xm = x - k.p0;
  const double xT = x * T;
  const double A1 = k.oma * (xT - k.p0l);
  const double A2 = k.oma2 * (xT * T - k.p0l2);
  const double wx = w * xm;
  c.zs += e * xm;
  c.t1 += e * (A1 + wx);
  c.t2 += e * (A2 + w * (2.0 * A1 + wx));
}

template <bool WIDE, bool GLDS, bool TLDS>
__global__ __launch_bounds__(kPassThreads) void k_passA(const Scalars *sc, SellDev S,
                                                       const double2 *ew_g, const double *X_g,
                                                       const double *T_g, double *partA) {
  extern __shared__ __align__(16) unsigned char smem[];
  using R = Rec<WIDE>;
  using RT = typename R::T;
  if (sc->done || sc->reset_pending) return;  // a pending re-evaluation skips pass A
  const int tid = threadIdx.x, lane = tid & 63;
  const uint32_t G = S.n_groups, n_lut = S.n_lut;
  double *sh = reinterpret_cast<double *>(smem);
  double *p = sh + 32;
  // {e_g, w_g} and {X_i, T_i} as 16-byte entries: one ds_read_b128 per lookup.  A wave can have
  // at most 16 LDS operations in flight (lgkmcnt is 4 bits), so wide reads double the cells whose
  // gathers overlap with arithmetic.
  double2 *ew_l = reinterpret_cast<double2 *>(p);
  if (GLDS) {
    p += 2 * ((size_t)G + 1);
    for (uint32_t g = tid; g <= G; g += kPassThreads) ew_l[g] = ew_g[g];
  }
  double2 *xt_l = reinterpret_cast<double2 *>(p);
  if (TLDS) {
    for (uint32_t i = tid; i < n_lut; i += kPassThreads) xt_l[i] = make_double2(X_g[i], T_g[i]);
  }
  auto EW_ = [&](uint32_t g) -> double2 { return GLDS ? ew_l[g] : ew_g[g]; };
  auto XT_ = [&](uint32_t i) -> double2 { return TLDS ? xt_l[i] : make_double2(X_g[i], T_g[i]); };
  const double p0 = sc->p0, U = sc->U, logzi = sc->logzi, oma = 1.0 - sc->a;
  const CstA cst = {p0, oma, oma * oma, p0 * logzi, p0 * logzi * logzi};
  const double zbase = p0 * U, b1 = p0 * sc->V1c, b2 = p0 * sc->V2c;
  double nn = 0.0;
  __syncthreads();

  const uint32_t n_sell = S.n_ecs - S.n_long;
  const uint32_t nw = gridDim.x * (kPassThreads / 64);
  // Two register buffers in ping-pong: while one slice is processed the records of the wave's
  // next slice are in flight.  The explicit vmcnt(0) sits BEFORE the next buffer's loads are
  // issued, so it only waits for the buffer about to be consumed.
  RT buf0[kRegCells] = {}, buf1[kRegCells] = {};
  uint32_t o0 = 0, len0 = 0, o1 = 0, len1 = 0;
  auto issue = [&](uint32_t sl, RT(&b)[kRegCells], uint32_t &o, uint32_t &len) {
    o = uniform(S.slice_off[sl]);
    len = uniform(S.slice_off[sl + 1]) - o;
    if (len <= (uint32_t)kRegCells) load_slice<WIDE>(S.rec, (size_t)o * 64 + lane, len, b);
  };
  auto process = [&](uint32_t sl, RT(&b)[kRegCells], uint32_t o, uint32_t len) {
    AccA c = {0.0, 0.0, 0.0};
    if (len <= (uint32_t)kRegCells) {
      // straight-line code per slice length (len is wave-uniform and even): all LDS gathers of
      // the slice can be in flight together instead of one scalar-branched pair at a time
      auto fixed = [&](auto LEN) {
        constexpr int L = decltype(LEN)::value;
        constexpr int B = 8;  // cells gathered together: 16 x ds_read_b128 in flight
#pragma unroll
        for (int k0 = 0; k0 < L; k0 += B) {
          double2 ewv[B], xtv[B];
#pragma unroll
          for (int k = 0; k < B; ++k) {
            if (k0 + k < L) {
              ewv[k] = EW_(R::grp(b[k0 + k]));
              xtv[k] = XT_(R::idx(b[k0 + k]));
            }
          }
#pragma unroll
          for (int k = 0; k < B; ++k)
            if (k0 + k < L) cellA(c, cst, ewv[k].x, ewv[k].y, xtv[k].x, xtv[k].y);
        }
      };
      switch (len) {
        case 0: break;
        case 2: fixed(std::integral_constant<int, 2>{}); break;
        case 4: fixed(std::integral_constant<int, 4>{}); break;
        case 6: fixed(std::integral_constant<int, 6>{}); break;
        case 8: fixed(std::integral_constant<int, 8>{}); break;
        case 10: fixed(std::integral_constant<int, 10>{}); break;
        case 12: fixed(std::integral_constant<int, 12>{}); break;
        case 14: fixed(std::integral_constant<int, 14>{}); break;
        default: fixed(std::integral_constant<int, 16>{}); break;
      }
    } else {
      const size_t base = (size_t)o * 64 + lane;
      for (uint32_t k = 0; k < len; k += 2) {
        const RT r0 = R::load(S.rec, base + (size_t)k * 64);
        const RT r1 = R::load(S.rec, base + (size_t)(k + 1) * 64);
        const double2 a0 = EW_(R::grp(r0)), a1 = EW_(R::grp(r1));
        const double2 x0 = XT_(R::idx(r0)), x1 = XT_(R::idx(r1));
        cellA(c, cst, a0.x, a0.y, x0.x, x0.y);
        cellA(c, cst, a1.x, a1.y, x1.x, x1.y);
      }
    }
    if (sl * 64 + lane < n_sell) {
      const double iZ = 1.0 / (zbase + c.zs);
      const double S1 = (b1 + c.t1) * iZ, S2 = (b2 + c.t2) * iZ;
      nn += S2 - S1 * S1;
    }
  };
  uint32_t s0 = uniform(blockIdx.x * (kPassThreads / 64) + (tid >> 6)), s1;
  if (s0 < S.nslices) issue(s0, buf0, o0, len0);
  while (s0 < S.nslices) {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): buf0 has landed
    s1 = s0 + nw;
    if (s1 < S.nslices) issue(s1, buf1, o1, len1);
    process(s0, buf0, o0, len0);
    if (s1 >= S.nslices) break;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // buf1 has landed
    s0 = s1 + nw;
    if (s0 < S.nslices) issue(s0, buf0, o0, len0);
    process(s1, buf1, o1, len1);
  }
  // long ECs: the whole workgroup strides over one EC's cells
  for (uint32_t r = blockIdx.x; r < S.n_long; r += gridDim.x) {
    AccA c = {0.0, 0.0, 0.0};
    for (uint32_t k = S.long_ptr[r] + tid; k < S.long_ptr[r + 1]; k += kPassThreads) {
      const RT rc = R::load(S.rec_long, k);
      const double2 a0 = EW_(R::grp(rc)), x0 = XT_(R::idx(rc));
      cellA(c, cst, a0.x, a0.y, x0.x, x0.y);
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
// Pass B: per EC Z_j (softmax denominator), r_j = c_j / Z_j, the ELBO data terms and the column
// sums A_g = sum_j r_j (x_gj - p0) accumulated in an LDS-private table (rcgpar logsumexp +
// update_N_k + ELBO_rcg_mat in one sweep).  The (group, x - p0) pairs of an EC stay in registers
// between the row sum and the scatter.
// ---------------------------------------------------------------------------------------
template <bool WIDE, bool GLDS, bool TLDS>
__global__ __launch_bounds__(kPassThreads) void k_passB(const Scalars *sc, SellDev S, const double *e_g,
                                                       const double *X_g, const double *T_g,
                                                       double *partAcc, double *partS,
                                                       double *accGlobal) {
  extern __shared__ __align__(16) unsigned char smem[];
  using R = Rec<WIDE>;
  using RT = typename R::T;
  if (sc->done) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const uint32_t G = S.n_groups, n_lut = S.n_lut;
  double *sh = reinterpret_cast<double *>(smem);
  double *p = sh + 32;
  const double *e_l = e_g;
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
  double2 *xt_l = reinterpret_cast<double2 *>(p);  // {X_i, T_i}: one ds_read_b128 per lookup
  if (TLDS) {
    for (uint32_t i = tid; i < n_lut; i += kPassThreads) xt_l[i] = make_double2(X_g[i], T_g[i]);
  }
  auto XT_ = [&](uint32_t i) -> double2 { return TLDS ? xt_l[i] : make_double2(X_g[i], T_g[i]); };
  const double p0 = sc->p0, U = sc->U, logzi = sc->logzi;
  const double p0l = p0 * logzi;
  const double zbase = p0 * U, hbase = p0l * U;
  double s_clogZ = 0.0, s_rH = 0.0, s_W = 0.0;
  __syncthreads();

  const uint32_t n_sell = S.n_ecs - S.n_long;
  const uint32_t nw = gridDim.x * (kPassThreads / 64);
  RT buf0[kRegCells] = {}, buf1[kRegCells] = {};
  uint32_t o0 = 0, len0 = 0, o1 = 0, len1 = 0;
  double c0 = 0.0, c1 = 0.0;
  auto issue = [&](uint32_t sl, RT(&b)[kRegCells], uint32_t &o, uint32_t &len, double &c) {
    o = uniform(S.slice_off[sl]);
    len = uniform(S.slice_off[sl + 1]) - o;
    if (len <= (uint32_t)kRegCells) load_slice<WIDE>(S.rec, (size_t)o * 64 + lane, len, b);
    c = (sl * 64 + lane < n_sell) ? S.cvec[S.n_long + sl * 64 + lane] : 0.0;
  };
  auto process = [&](RT(&b)[kRegCells], uint32_t o, uint32_t len, double c) {
    double zs = 0.0, hs = 0.0;
    if (len <= (uint32_t)kRegCells) {
      // row sums: straight-line code per slice length (wave-uniform, even); x - p0 of every cell
      // stays in registers for the scatter
      double xv[kRegCells];
      auto fixed = [&](auto LEN) {
        constexpr int L = decltype(LEN)::value;
        constexpr int B = 4;
#pragma unroll
        for (int k0 = 0; k0 < L; k0 += B) {
          double ev[B];
          double2 xt[B];
#pragma unroll
          for (int k = 0; k < B; ++k) {
            if (k0 + k < L) {
              ev[k] = e_l[R::grp(b[k0 + k])];
              xt[k] = XT_(R::idx(b[k0 + k]));
            }
          }
#pragma unroll
          for (int k = 0; k < B; ++k) {
            if (k0 + k < L) {
              const double m = xt[k].x - p0;
              zs += ev[k] * m;
              hs += ev[k] * (xt[k].x * xt[k].y - p0l);
              xv[k0 + k] = m;
            }
          }
        }
      };
      switch (len) {
        case 0: break;
        case 2: fixed(std::integral_constant<int, 2>{}); break;
        case 4: fixed(std::integral_constant<int, 4>{}); break;
        case 6: fixed(std::integral_constant<int, 6>{}); break;
        case 8: fixed(std::integral_constant<int, 8>{}); break;
        case 10: fixed(std::integral_constant<int, 10>{}); break;
        case 12: fixed(std::integral_constant<int, 12>{}); break;
        case 14: fixed(std::integral_constant<int, 14>{}); break;
        default: fixed(std::integral_constant<int, 16>{}); break;
      }
      if (c != 0.0) {
        const double Z = zbase + zs, H = hbase + hs;
        const double rj = c / Z;
        s_clogZ += c * log(Z);
        s_rH += rj * H;
        s_W += rj;
#pragma unroll
        for (int k = 0; k < kRegCells; k += 2) {
          if ((uint32_t)k < len) {
            // padding records (group id == G) all target one address: skip them instead of
            // serialising up to 64 same-address LDS atomics per step
            const uint32_t g0 = R::grp(b[k]), g1 = R::grp(b[k + 1]);
            if (g0 != G) atomicAdd(&acc[g0], rj * xv[k]);
            if (g1 != G) atomicAdd(&acc[g1], rj * xv[k + 1]);
          }
        }
      }
    } else {
      const size_t base = (size_t)o * 64 + lane;
      for (uint32_t k = 0; k < len; k += 2) {
        const RT r0 = R::load(S.rec, base + (size_t)k * 64);
        const RT r1 = R::load(S.rec, base + (size_t)(k + 1) * 64);
        const double e0 = e_l[R::grp(r0)], e1 = e_l[R::grp(r1)];
        const double2 t0 = XT_(R::idx(r0)), t1 = XT_(R::idx(r1));
        zs += e0 * (t0.x - p0);
        hs += e0 * (t0.x * t0.y - p0l);
        zs += e1 * (t1.x - p0);
        hs += e1 * (t1.x * t1.y - p0l);
      }
      if (c != 0.0) {
        const double Z = zbase + zs, H = hbase + hs;
        const double rj = c / Z;
        s_clogZ += c * log(Z);
        s_rH += rj * H;
        s_W += rj;
        for (uint32_t k = 0; k < len; ++k) {
          const RT r = R::load(S.rec, base + (size_t)k * 64);
          atomicAdd(&acc[R::grp(r)], rj * (XT_(R::idx(r)).x - p0));
        }
      }
    }
  };
  uint32_t s0 = uniform(blockIdx.x * (kPassThreads / 64) + (tid >> 6)), s1;
  if (s0 < S.nslices) issue(s0, buf0, o0, len0, c0);
  while (s0 < S.nslices) {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): buf0 / c0 have landed
    s1 = s0 + nw;
    if (s1 < S.nslices) issue(s1, buf1, o1, len1, c1);
    process(buf0, o0, len0, c0);
    if (s1 >= S.nslices) break;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    s0 = s1 + nw;
    if (s0 < S.nslices) issue(s0, buf0, o0, len0, c0);
    process(buf1, o1, len1, c1);
  }
  for (uint32_t r = blockIdx.x; r < S.n_long; r += gridDim.x) {
    double zs = 0.0, hs = 0.0;
    for (uint32_t k = S.long_ptr[r] + tid; k < S.long_ptr[r + 1]; k += kPassThreads) {
      const RT rc = R::load(S.rec_long, k);
      const double eg = e_l[R::grp(rc)];
      const double2 t = XT_(R::idx(rc));
      zs += eg * (t.x - p0);
      hs += eg * (t.x * t.y - p0l);
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
        const RT rc = R::load(S.rec_long, k);
        atomicAdd(&acc[R::grp(rc)], rj * (XT_(R::idx(rc)).x - p0));
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

}  // namespace msw
