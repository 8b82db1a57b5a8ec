// device_util.hpp -- wave / block reductions and the reference digamma series (gfx950, wave64).
#pragma once
#include "common.hpp"

namespace msw {

// ---------------------------------------------------------------------------------------
// wave / block reductions (wave64).  The cross-lane steps are DPP moves (row_shr 1/2/4/8 inside a
// row of 16 lanes, row_bcast:15 / row_bcast:31 across rows -- gfx9-family controls) feeding an
// ordinary fp64 operation: no LDS round trip per step as with ds_bpermute-based shuffles.  Lane 63
// ends with the result of a fixed tree and is broadcast, so every lane returns the same value and
// the order of the operations is fixed -> bitwise reproducible.
// ---------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double v, double fill) {
  // lanes whose source is outside the row, or whose row is masked, receive `fill`
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(fill), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bcast_lane63(double v) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_move<0x111, 0xf>(v, 0.0);  // row_shr:1
  v += dpp_move<0x112, 0xf>(v, 0.0);  // row_shr:2
  v += dpp_move<0x114, 0xf>(v, 0.0);  // row_shr:4
  v += dpp_move<0x118, 0xf>(v, 0.0);  // row_shr:8  -> lane 15 of each row: the row's sum
  v += dpp_move<0x142, 0xa>(v, 0.0);  // row_bcast:15 into rows 1, 3
  v += dpp_move<0x143, 0xc>(v, 0.0);  // row_bcast:31 into rows 2, 3 -> lane 63: the total
  return bcast_lane63(v);
}
__device__ __forceinline__ double wave_max(double v) {
  const double ninf = -INFINITY;
  v = fmax(v, dpp_move<0x111, 0xf>(v, ninf));
  v = fmax(v, dpp_move<0x112, 0xf>(v, ninf));
  v = fmax(v, dpp_move<0x114, 0xf>(v, ninf));
  v = fmax(v, dpp_move<0x118, 0xf>(v, ninf));
  v = fmax(v, dpp_move<0x142, 0xa>(v, ninf));
  v = fmax(v, dpp_move<0x143, 0xc>(v, ninf));
  return bcast_lane63(v);
}
// Sum over the 2^lgm lanes of an aligned lane group, lgm <= 6 and wave-uniform (the lanes of one EC in a slice of
// more than one lane per EC, sell.hpp): every lane of the group ends with the same bits (each step adds two partial
// sums in either order: fp addition is commutative).  quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror,
// row_mirror: all inside a row of 16 lanes; across rows (32 and 64 lanes per EC) through the LDS crossbar.
__device__ __forceinline__ double group_sum(double v, uint32_t lgm) {
  if (lgm >= 1) v += dpp_move<0xB1, 0xf>(v, 0.0);
  if (lgm >= 2) v += dpp_move<0x4E, 0xf>(v, 0.0);
  if (lgm >= 3) v += dpp_move<0x141, 0xf>(v, 0.0);
  if (lgm >= 4) v += dpp_move<0x140, 0xf>(v, 0.0);
  if (lgm >= 5) v += __shfl_xor(v, 16, 64);
  if (lgm >= 6) v += __shfl_xor(v, 32, 64);
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
// N sums at once with one pair of barriers.  sh: >= 16 * N doubles.  Per value the order of
// additions is that of block_sum, so both give bitwise the same result.
template <int N>
__device__ __forceinline__ void block_sum_n(double (&v)[N], double *sh) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = wave_sum(v[i]);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) sh[w * N + i] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double r = 0.0;
    for (int j = 0; j < nw; ++j) r += sh[j * N + i];
    v[i] = r;
  }
}
// The same reductions for a workgroup of NW wavefronts known at compile time (the kernels between the sweeps are
// chains of such reductions): the second stage is straight-line code -- the NW partial results of a sum are read from
// LDS together and added in wavefront order (the same order, the same bits as block_sum_n) instead of a loop of NW
// dependent LDS round trips.  sh: >= NW * N doubles.
template <int N, int NW>
__device__ __forceinline__ void block_sum_fixed(double (&v)[N], double *sh) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = wave_sum(v[i]);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) sh[w * N + i] = v[i];
  }
  __syncthreads();
  constexpr int C = NW < 8 ? NW : 8;  // partial results in flight at a time (bounds the registers)
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double r = 0.0;
#pragma unroll
    for (int j0 = 0; j0 < NW; j0 += C) {
      double t[C];
#pragma unroll
      for (int j = 0; j < C; ++j) t[j] = j0 + j < NW ? sh[(j0 + j) * N + i] : 0.0;
#pragma unroll
      for (int j = 0; j < C; ++j)
        if (j0 + j < NW) r += t[j];
    }
    v[i] = r;
  }
}
template <int NW>
__device__ __forceinline__ double block_sum_fixed1(double v, double *sh) {
  double a[1] = {v};
  block_sum_fixed<1, NW>(a, sh);
  return a[0];
}
template <int NW>
__device__ __forceinline__ double block_max_fixed(double v, double *sh) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double t[NW];
#pragma unroll
  for (int k = 0; k < NW; ++k) t[k] = sh[k];
  double r = t[0];
#pragma unroll
  for (int k = 1; k < NW; ++k) r = fmax(r, t[k]);
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

// exp(y) for the value records' per-cell exponentials, y = a (T - tref) <= 0 (sweep_kernels.hpp).  MSW_EXP_LE0=1 (an A/B
// build macro, tools/ab_build.py) replaces ocml's exp -- ~35 vector instructions with its handling of the whole range
// -- by k = rint(y / ln 2), r = y - k ln 2 in two fused steps (|r| <= 0.347), the Taylor polynomial of degree 13
// (r^14 / 14! < 5e-18) and v_ldexp: 20 instructions, within 1 ulp of the correctly rounded value over [-745, 1] (60 M
// random arguments against long double on the host).  Arguments below -746, and -inf, give exactly 0 through the clamp.
#ifndef MSW_EXP_LE0
#define MSW_EXP_LE0 0
#endif
__device__ __forceinline__ double exp_le0(double y) {
#if MSW_EXP_LE0
  const double yc = fmax(y, -746.0);
  const double k = rint(yc * 1.4426950408889634);
  double r = fma(k, -0x1.62e42fefa39efp-1, yc);
  r = fma(k, -0x1.abc9e3b39803fp-56, r);
  double p = 1.6059043836821613e-10;
  p = fma(p, r, 2.08767569878681e-09);
  p = fma(p, r, 2.505210838544172e-08);
  p = fma(p, r, 2.755731922398589e-07);
  p = fma(p, r, 2.7557319223985893e-06);
  p = fma(p, r, 2.48015873015873e-05);
  p = fma(p, r, 0.0001984126984126984);
  p = fma(p, r, 0.001388888888888889);
  p = fma(p, r, 0.008333333333333333);
  p = fma(p, r, 0.041666666666666664);
  p = fma(p, r, 0.16666666666666666);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)k);
#else
  return exp(y);
#endif
}

// Fixed-point column sums (sweep_kernels.hpp): a cell of group g adds rint(2^K * r_j * (x - p0) * f_g),
//   f_g = e_g                    for e_g >= 2^-s   (units of 2^-K reads)
//   f_g = mantissa(e_g) * 2^-s   below             (units of 2^-K * e_g / f_g reads: finer by a power of two)
// i.e. f_g = max(e_g, mantissa(e_g) * 2^-s), and k_redfin scales the group's total by 2^-K * e_g / f_g.
// Range: the accumulators hold only the listed cells' part of N_g = e_g p0 W + e_g sum_j r_j (x_gj - p0),
// which near the guard threshold is up to 2^8 times -sum c and does NOT fit: they are sums MODULO 2^64, and
// k_redfin adds the background part in the same units and the same modular arithmetic -- what has to fit is
// N_g itself.  N_g <= sum c for the first kind of group; for the second, N_g / e_g = sum_j r_j (x_gj or p0)
// with r_j <= 2^8 c_j / (p0 U) (the guard keeps Z_j above 2^-8 of the background sum p0 U) and x <= xb,
// hence N_g / e_g <= 2^8 (xb / p0) sum c, and with s = 9 + ceil(log2(xb / p0)) (Scalars::fx_shift, per
// pass) f_g N_g / e_g < 2 sum c: both stay below 2 * sum c * 2^K < 2^62.  A group that dies out keeps
// its relative precision (with priors below one digamma(N_g) has slope 1 / N_g^2 and a fixed grid in
// reads is felt), the large groups keep the absolute one.  e_g = 0: f_g = 2^-s, e_g / f_g = 0 -- whatever
// the cells add is discarded (denormal weights are flushed to 0 where e_g is formed).
// fx_expbits(s): the exponent field of 2^-s, the form fx_factor takes s in.
__device__ __forceinline__ int fx_expbits(int s) { return (1023 - s) << 20; }
__device__ __forceinline__ double fx_factor(double e, int expbits) {
  // max(e, mantissa(e) * 2^-s): the two differ in the exponent field only and e >= 0, so the larger double is
  // the one with the larger high word -- an integer max (fmax costs two canonicalising v_max_f64 more per cell)
#ifdef MSW_FXF_FMAX  // A/B build macro (tools/ab_build.py): the floating-point max
  return fmax(e, __hiloint2double((__double2hiint(e) & 0x000FFFFF) | expbits, __double2loint(e)));
#else
  const int hi = __double2hiint(e);
  int ms;  // (hi & 0xFFFFF) | expbits as one bit-field insert (the compiler emits v_and + v_or)
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(ms) : "v"(0x000FFFFF), "v"(hi), "v"(expbits));
  return __hiloint2double(hi > ms ? hi : ms, __double2loint(e));
#endif
}
// reference value of a pass's tables: exp(a (T - tref)) <= 1 for every table value T and for log zi
__device__ __forceinline__ double tref_of(double a, double tmax, double tmin) { return a >= 0.0 ? tmax : tmin; }
// s for a pass: 9 + ceil(log2(xb / p0)), at least 9
__device__ __forceinline__ int fx_shift_of(double xb, double p0) {
  int ex = 0;
  frexp(xb / p0, &ex);  // xb / p0 < 2^ex
  ex = ex < 0 ? 0 : ex;
  return ex > 900 ? 909 : 9 + ex;
}
__device__ __forceinline__ double flush_denormal(double e) { return e < 0x1p-1000 ? 0.0 : e; }

// digamma: the 7-shift asymptotic series of the reference (src/Sample.cpp:87-97; rcgpar
// carries the same function for the RCG gradient).
__device__ __forceinline__ double digamma_ref(double x) {
  double result = 0.0;
  // the shift loop `for (; x < 7; ++x) result -= 1 / x` with its (up to seven) divisions independent of each other:
  // the same operands, the same order of the subtractions, the same bits -- one division's latency instead of seven
  // (most groups of a sample sit at N_g = alpha + nearly nothing: the whole loop)
  if (x >= 0.0 && x < 7.0) {
    double xs[8], q[7];
    xs[0] = x;
#pragma unroll
    for (int k = 1; k < 8; ++k) xs[k] = xs[k - 1] + 1.0;  // the loop's own additions
#pragma unroll
    for (int k = 0; k < 7; ++k) q[k] = 1.0 / xs[k];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      if (xs[k] < 7.0) {  // (a prefix: xs ascends)
        result -= q[k];
        x = xs[k + 1];
      }
    }
  } else {
    for (; x < 7.0; x += 1.0) result -= 1.0 / x;  // (x < 0: never met, N_g >= alpha_g > 0; NaN: no iteration)
  }
  x -= 0.5;
  const double xx = 1.0 / x, xx2 = xx * xx, xx4 = xx2 * xx2;
  result += log(x) + (1. / 24.) * xx2 - (7.0 / 960.0) * xx4 + (31.0 / 8064.0) * xx4 * xx2 -
            (127.0 / 30720.0) * xx4 * xx4;
  return result;
}

}  // namespace msw
