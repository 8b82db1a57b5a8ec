// em_f32_kernels.hpp -- `--algorithm emgpu --emprecision float` (src/mSWEEP.cpp:129,200-203: rcgpar::em_torch with
// precision "float" [UPSTREAM-UNVERIFIED]; the reference's fastest GPU mode, docs/gpubenchmarks.md:22,25) as REAL
// fp32 arithmetic.  Until round 4 MSW_PREC_FLOAT ran the fp64 kernels.
//
// What "float" means here (the reading restated in oracle/rcg_oracle.cpp orc_em_dense_f32):
//   * the likelihood values, the weights theta, the responsibilities' numerators e_g exp(T - tref), the row sums
//     Z_j and the quotients r_j = c_j / Z_j are fp32;
//   * c_j log Z_j is formed in fp32 per EC; the sum over the ECs -- like every sum over millions of terms in this
//     library -- is accumulated in fp64, and the log-likelihood is ROUNDED TO FLOAT once per iteration: the stop rule
//     compares two floats (a float near 1e8 moves in steps of 8, which is why the reference's float mode stops after a
//     few hundred iterations where double runs into --max-iters: docs/gpubenchmarks.md:20-22);
//   * the column sums stay 64-bit fixed-point integers (exact, order-independent): a cell adds rint(2^K r_j e_g x)
//     with the product formed in fp32.  No per-group power-of-two factor (device_util.hpp fx_factor): 2^-K reads of
//     absolute resolution is far below fp32's relative 6e-8 of any weight a float run can resolve.
// An EM iteration with a = 1: x_i = exp(T_i - tref) never changes, so the slot table {x_i - p0} is built ONCE per solve
// and is 4 bytes per entry; e_g is 4 bytes per group: the sweep's LDS image holds the records' own byte offsets
// shifted down -- entry offset >> 2, group offset >> 1 -- so the 4-byte records of the fp64 sweeps serve unchanged.
// Per cell: 2 ds_read_b32 + 1 ds_add_u64 against ds_read_b64 + ds_read_b128 + ds_add_u64, ~7 instead of ~11 vector
// operations, no second table column (EM's objective has no entropy term).
//
// Served layouts: 4-byte offset records with the whole slot table and the group vectors in LDS (every likelihood built
// from a pseudoalignment up to ~11 000 groups), any EC length (slice classes, wavefront-per-EC), one GPU.  Everything
// else -- wide / index / value records, dense matrices without background structure, EC-sharded solves -- runs the
// fp64 kernels under MSW_PREC_FLOAT as before; msw_timing::em_float_kernels says which it was.
#pragma once
#include "sweep_kernels.hpp"

namespace msw {

// ---- fp32 lane reductions (DPP, as device_util.hpp's fp64 ones) ---------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move_f(float v, float fill) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_f(float v) {
  v += dpp_move_f<0x111, 0xf>(v, 0.0f);
  v += dpp_move_f<0x112, 0xf>(v, 0.0f);
  v += dpp_move_f<0x114, 0xf>(v, 0.0f);
  v += dpp_move_f<0x118, 0xf>(v, 0.0f);
  v += dpp_move_f<0x142, 0xa>(v, 0.0f);
  v += dpp_move_f<0x143, 0xc>(v, 0.0f);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float group_sum_f(float v, uint32_t lgm) {
  if (lgm >= 1) v += dpp_move_f<0xB1, 0xf>(v, 0.0f);
  if (lgm >= 2) v += dpp_move_f<0x4E, 0xf>(v, 0.0f);
  if (lgm >= 3) v += dpp_move_f<0x141, 0xf>(v, 0.0f);
  if (lgm >= 4) v += dpp_move_f<0x140, 0xf>(v, 0.0f);
  if (lgm >= 5) v += __shfl_xor(v, 16, 64);
  if (lgm >= 6) v += __shfl_xor(v, 32, 64);
  return v;
}

// LDS image of the fp32 sweep (byte offsets; bhi = sell_bhi(n_tab) as in the fp64 image, Gp = G + kSentinels):
//   [0, 4 n_tab)                       x_i - p0 as float, entry i at (16 i) >> 2
//   [bhi / 2, bhi / 2 + 4 Gp)          e_g as float, group g at (bhi + 8 g) >> 1
//   [bhi + C, bhi + C + 8 Gp)          column sums (64-bit fixed point), group g at (bhi + 8 g) + C
//   then 48 doubles of reduction scratch and the slice geometry of 16 wavefronts (as the fp64 image's tail)
__host__ __device__ inline uint32_t em_f32_acc_off(uint32_t G) { return ((4u * (G + kSentinels)) + 7u) & ~7u; }
__host__ __device__ inline size_t em_f32_scratch_off(uint32_t n_tab, uint32_t G) {
  return (size_t)sell_bhi(n_tab) + em_f32_acc_off(G) + 8 * ((size_t)G + kSentinels);
}
__host__ __device__ inline size_t em_f32_lds_bytes(uint32_t n_tab, uint32_t G) {
  return em_f32_scratch_off(n_tab, G) + 64 * sizeof(double) + 16 * kGeoStride;
}

// Once per float solve, behind k_em_init (which left M, U, p0, tref and the fp64 e_g of the uniform start):
// the fp32 slot table from the FLOAT-rounded table values, e_g rounded to float (and written back as such: every
// consumer of e_g -- k_redfin's background share, the guarded ECs -- sees the values the sweep uses), U over them.
__global__ __launch_bounds__(1024) void k_em_f32_prep(Scalars *sc, int G, int n_tab, const double *lut_area, double *e,
                                                     float *e32, float *tab32) {
  __shared__ double sh[32];
  const int tid = threadIdx.x, nt = blockDim.x;
  const double tref = sc->tref, p0 = (double)(float)sc->p0;
  for (int i = tid; i < n_tab; i += nt)
    tab32[i] = (float)(exp((double)(float)lut_area[i] - tref) - p0);
  double su = 0.0;
  for (int g = tid; g < G + (int)kSentinels; g += nt) {
    const float ef = g < G ? (float)e[g] : 0.0f;
    e32[g] = ef;
    if (g < G) {
      e[g] = (double)ef;
      su += (double)ef;
    }
  }
  const double U = block_sum(su, sh);
  if (tid == 0) {
    sc->U = U;
    sc->p0 = p0;
  }
}

typedef __attribute__((address_space(3))) const float lds_cf_t;
typedef __attribute__((address_space(3))) unsigned long long lds_u64f_t;

template <bool ML>
__global__ __launch_bounds__(1024) void k_em_passB_f32(const Scalars *sc, SellDev S, const double *e_g, const float *e32_g,
                                                      const float *tab32_g, double *partAcc, double *partS, GuardDev GD) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int ENC = kEncNarrow, NT = 1024;
  using R = Rec<ENC>;
  const RecDec D = rec_dec(S);
  const uint32_t n_tab = S.n_tab_lds;
  const int skip = sc->done;
  const int tid = threadIdx.x, lane = tid & 63;
  const uint32_t G = S.n_groups, Gp = G + kSentinels, bhi = S.bhi;
  const uint32_t accC = em_f32_acc_off(G);
  const uint32_t scratch_off = (uint32_t)em_f32_scratch_off(n_tab, G);
  double *sh = reinterpret_cast<double *>(smem + scratch_off);
  SliceStream<ENC, MSW_REVERSE_B> stream(S, uniform(blockIdx.x * (NT / 64) + (tid >> 6)), gridDim.x * (NT / 64),
                                         (uint32_t)lane, scratch_off + 64 * 8 + uniform(tid >> 6) * kGeoStride);
  // (the first slice's records in flight under the LDS fill: SliceStream::prime)
  const uint32_t n_lanes = S.nslices * 64u;
  const uint32_t null_rec = R::make(G + (uint32_t)lane, 0u, D);  // the lane's own sentinel group: e = 0
  stream.nullr = null_rec;
  stream.nullr_hot = null_rec;
  auto issue = [&](SliceBuf<ENC> &sb) {
    const uint32_t q = sb.sl * 64 + lane;
    const uint32_t cj = S.c8s[q < n_lanes ? q : 0u];
    sb.c8 = q < n_lanes ? cj : 0u;
  };
  SliceBuf<ENC> first = {};
  stream.prime(first, issue);
  {
    float *t = reinterpret_cast<float *>(smem);
    for (uint32_t i = tid; i < n_tab; i += NT) t[i] = tab32_g[i];
    float *el = reinterpret_cast<float *>(smem + bhi / 2);
    unsigned long long *al = reinterpret_cast<unsigned long long *>(smem + bhi + accC);
    for (uint32_t g = tid; g < Gp; g += NT) {
      el[g] = e32_g[g];
      al[g] = 0ull;
    }
  }
  auto E_ = [&](uint32_t r) -> float { return *(lds_cf_t *)(size_t)((r >> D.shift) >> 1); };
  auto X_ = [&](uint32_t r) -> float { return *(lds_cf_t *)(size_t)((r & D.mask) >> 2); };
  // one column-sum update: rint(2^K q) as a 64-bit integer (sweep_kernels.hpp fx_bits; q = r_j e_g (x - p0) in fp32)
  const double fxs = uniform_d(sc->fx_scale);
  // (|q| <= 2^8 c_j: e_g x <= Z_j, and the guard keeps Z_j above 2^-8 of the background sum.  The one-fma conversion
  // needs |2^K q| < 2^51: an EC that holds more than ~2^-18 of all reads -- toy inputs -- splits its addends into two
  // parts of 32 and 51 bits like the fp64 sweep's wide adds; the sums are modulo 2^64, the parts need not meet)
  const float narrow_c = (float)(0x1p43 / fxs);
  auto add = [&](uint32_t r, float q, bool narrow) {
    lds_u64f_t *dst = (lds_u64f_t *)(size_t)((r >> D.shift) + accC);
    if (narrow) {
      __hip_atomic_fetch_add(dst, fx_bits((double)q, fxs), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      const double qq = (double)q * fxs;                     // exact: fxs is a power of two
      const double vh = fma(qq, 0x1p-32, kFxMagic);
      const double qh = vh - kFxMagic;                       // rint(qq / 2^32)
      const double ql = fma(-qh, 0x1p32, qq);                // |ql| <= 2^31, exact
      __hip_atomic_fetch_add(dst, (unsigned long long)(uint32_t)__double2loint(vh) << 32, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(dst, fx_bits(1.0, ql), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  };
  const float p0 = (float)uniform_d(sc->p0), zbase = p0 * (float)uniform_d(sc->U);
  const float gthr = fmaxf(zbase * (float)kGuardRatio, 1.17549435e-38f);
  const uint32_t gcnt_off = scratch_off + 128u;
  typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
  auto defer = [&](uint32_t p) {
    const uint32_t i = __hip_atomic_fetch_add((lds_u32_t *)(size_t)gcnt_off, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (i < GD.cap) GD.list[(size_t)blockIdx.x * GD.cap + i] = p;
    else *GD.err = 2;
  };
  double s_clogZ = 0.0, s_W = 0.0;  // sums over millions of ECs: fp64 accumulators of fp32 terms (file header)
  if (skip) return;
  if (tid == 0) *(lds_u32_t *)(size_t)gcnt_off = 0u;
  __syncthreads();

  // Z, r, the EC's log-likelihood term and W for one EC whose row sum is zs; returns r (0: nothing to scatter)
  auto epilogue = [&](float zs, float c, bool spoke, uint32_t pos) -> float {
    const float Z = zbase + zs;
    if (c == 0.0f) return 0.0f;
    if (!(Z >= gthr)) {
      if (spoke) defer(pos);
      return 0.0f;
    }
    const float rj = c / Z;
    if (spoke) {
      s_W += (double)rj;
      s_clogZ += (double)(c * logf(Z));
    }
    return rj;
  };
  auto process = [&](SliceBuf<ENC> &sb) {
    const uint32_t len = sb.len;
    const uint32_t lgm = ML ? sb.lgm : 0u;
    const uint32_t pos = S.n_long + (ML ? slice_geo(S.cls, sb.sl).ec0 + ((uint32_t)lane >> lgm) : sb.sl * 64 + lane);
    float c = (float)sb.c8;
    if (__builtin_amdgcn_ballot_w64(sb.c8 == kC8Escape)) {  // (wave-uniform; the wait inside: see k_passB)
      if (sb.c8 == kC8Escape) c = (float)S.cvec[pos];
      __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    const bool spoke = !ML || lgm == 0u || ((uint32_t)lane & ((1u << lgm) - 1u)) == 0u;
    if (len <= (uint32_t)kRegCells) {
      float zs = 0.0f, pk[kRegCells];
#pragma unroll
      for (int k0 = 0; k0 < kRegCells; k0 += 4) {
        if ((uint32_t)k0 < len) {  // (rows past an odd slice's end hold the lane's null record: load_slice)
          float ev[4], xv[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint32_t r = (uint32_t)(k0 + k) < len + (len & 1u) ? sb.r[k0 + k] : null_rec;
            ev[k] = E_(r), xv[k] = X_(r);
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            pk[k0 + k] = ev[k] * xv[k];
            zs += pk[k0 + k];
          }
        }
      }
      if (ML && lgm != 0u) zs = group_sum_f(zs, lgm);
      const float rj = epilogue(zs, c, spoke, pos);
      if (rj != 0.0f) {
        const bool narrow = c < narrow_c;
#pragma unroll
        for (int k = 0; k < kRegCells; k += 2) {
          if ((uint32_t)k < len) {
            add(sb.r[k], rj * pk[k], narrow);
            add(sb.r[k + 1], rj * pk[k + 1], narrow);  // (an odd slice's missing row: the null record, pk = 0)
          }
        }
      }
    } else {
      // more rows than the registers hold (MSWEEP_MULTILANE=0 layouts only: one lane per EC): row by row, twice
      const size_t base = (size_t)sb.o * 64 + lane;
      float zs = 0.0f;
      for (uint32_t k = 0; k < len; ++k) {
        const uint32_t r = S.rec[base + (size_t)k * 64];
        zs += E_(r) * X_(r);
      }
      const float rj = epilogue(zs, c, true, pos);
      if (rj != 0.0f)
        for (uint32_t k = 0; k < len; ++k) {
          const uint32_t r = S.rec[base + (size_t)k * 64];
          add(r, rj * (E_(r) * X_(r)), c < narrow_c);
        }
    }
  };
  stream.run(issue, process, [] {}, &first);
  // long ECs (plain CSR): one wavefront per EC, a cell per lane and step
  for (uint32_t r = stream.s_first; r < S.n_long; r += stream.nw) {
    const uint32_t k0 = S.long_ptr[r], k1 = S.long_ptr[r + 1];
    float zs = 0.0f;
    for (uint32_t k = k0 + lane; k < k1; k += 64) {
      const uint32_t rc = S.rec_long[k];
      zs += E_(rc) * X_(rc);
    }
    zs = wave_sum_f(zs);
    const float c = (float)S.cvec[r];
    const float rj = epilogue(zs, c, lane == 0, r);
    if (rj != 0.0f)
      for (uint32_t k = k0 + lane; k < k1; k += 64) {
        const uint32_t rc = S.rec_long[k];
        add(rc, rj * (E_(rc) * X_(rc)), c < narrow_c);
      }
  }
  // guarded ECs (sell.hpp): as in k_passB -- a wavefront each, every group visited, fp64 (rare path); each group
  // receives its share of c_j through the two global fixed-point limbs k_redfin adds to N_g
  __syncthreads();
  const uint32_t n_guard = min(*(lds_u32_t *)(size_t)gcnt_off, GD.cap);
  if (n_guard) {
    if (tid == 0) atomicAdd(GD.visits, (unsigned long long)n_guard);
    const uint32_t wv = uniform(tid >> 6), nwv = NT / 64;
    uint32_t *bits = GD.bits + (size_t)(blockIdx.x * 16 + wv) * GD.words;
    const double logzi = uniform_d(sc->logzi), tref = uniform_d(sc->tref), p0d = uniform_d(sc->p0);
    const double tls = uniform_d(sc->fx_tscale);
    auto add_share = [&](uint32_t g, double share) {
      const double vh = fma(share, tls, kFxMagic);
      const double qh = vh - kFxMagic;
      const double ql = fma(share, tls, -qh) * 0x1p36;
      atomicAdd(&GD.tail[2 * (size_t)g], fx_bits(1.0, qh));
      atomicAdd(&GD.tail[2 * (size_t)g + 1], fx_bits(1.0, ql));
    };
    (void)logzi;
    for (uint32_t i = wv; i < n_guard; i += nwv) {
      const uint32_t p = GD.list[(size_t)blockIdx.x * GD.cap + i];
      const double c = S.cvec[p];
      double z = 0.0;
      wave_cells<ENC>(S, p, (uint32_t)lane, [&](uint32_t g, double T) {
        atomicOr(&bits[g >> 5], 1u << (g & 31));
        z += e_g[g] * exp((double)(float)T - tref);
      });
      __builtin_amdgcn_s_waitcnt(0);
      double r0 = 0.0;
      for (uint32_t w0 = lane; w0 < GD.words; w0 += 64) {
        const uint32_t listed = atomicOr(&bits[w0], 0u);
        for (uint32_t b = 0; b < 32; ++b) {
          const uint32_t g = w0 * 32 + b;
          if (g < G && !((listed >> b) & 1u)) r0 += e_g[g];
        }
      }
      z = wave_sum(z), r0 = wave_sum(r0);
      const double Z = fma(p0d, r0, z);
      if (Z > 0.0) {
        const double rj = c / Z;
        if (lane == 0) s_clogZ += c * log(Z);
        wave_cells<ENC>(S, p, (uint32_t)lane, [&](uint32_t g, double T) {
          add_share(g, e_g[g] * rj * exp((double)(float)T - tref));
        });
        for (uint32_t w0 = lane; w0 < GD.words; w0 += 64) {
          const uint32_t listed = atomicAnd(&bits[w0], 0u);
          for (uint32_t b = 0; b < 32; ++b) {
            const uint32_t g = w0 * 32 + b;
            if (g < G && !((listed >> b) & 1u)) add_share(g, e_g[g] * (rj * p0d));
          }
        }
      } else {
        if (lane == 0) *GD.err = 1;
        for (uint32_t w0 = lane; w0 < GD.words; w0 += 64) atomicAnd(&bits[w0], 0u);
      }
      __builtin_amdgcn_s_waitcnt(0);
    }
  }
  {
    double t3[3] = {s_clogZ, 0.0, s_W};
    block_sum_n<3>(t3, sh + 8);
    if (tid == 0) {
      partS[4 * blockIdx.x + 0] = t3[0];
      partS[4 * blockIdx.x + 1] = 0.0;
      partS[4 * blockIdx.x + 2] = t3[2];
      partS[4 * blockIdx.x + 3] = 0.0;
    }
  }
  __syncthreads();
  const double *al = reinterpret_cast<const double *>(smem + bhi + accC);
  double *dst = partAcc + (size_t)blockIdx.x * G;
  for (uint32_t g = tid; g < G; g += NT) dst[g] = al[g];
}

}  // namespace msw
