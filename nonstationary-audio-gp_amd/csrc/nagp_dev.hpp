// nagp_dev.hpp -- device-side building blocks shared by all kernels (gfx950, wave64).
//   * wave / lane-group reductions on DPP (no LDS traffic)
//   * 4x4 block-tile helpers (the state is a sequence of <=4-wide diagonal blocks; every S x S
//     matrix is held as M x M tiles of 4x4 doubles, zero padded, "tile-major": tile (I,J) at
//     index I*M+J, 16 contiguous doubles, row-major inside the tile)
//   * mom_eval: the reference's `mom` callback (likModulatorPower.m:25-100,
//     likModulatorNMFPower.m:28-87, experiments/likModulatorPreCalcwn.m:28-86) as a
//     workgroup-cooperative cubature.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nagp {

constexpr int MAXM = 64;        // sites (= diagonal blocks) per step
constexpr double kSqrt2Pi = 2.5066282746310002;

// ---------------------------------------------------------------------------------------------
// DPP helpers.  CTRL: quad_perm [1,0,3,2]=0xB1, [2,3,0,1]=0x4E, row_half_mirror=0x141, row_mirror=0x140
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// Sum over aligned groups of G adjacent lanes (G = 1,2,4,8,16); every lane of the group gets the sum.
__device__ __forceinline__ double group_sum(double v, int G) {
  if (G >= 2) v += dpp_mov<0xB1>(v);
  if (G >= 4) v += dpp_mov<0x4E>(v);
  if (G >= 8) v += dpp_mov<0x141>(v);
  if (G >= 16) v += dpp_mov<0x140>(v);
  return v;
}

__device__ __forceinline__ double readlane_d(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// Full-wave sum (all 64 lanes must be active); result is wave-uniform.  Fixed summation order.
__device__ __forceinline__ double wave_sum(double v) {
  v = group_sum(v, 16);
  return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}
__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, dpp_mov<0xB1>(v));
  v = fmax(v, dpp_mov<0x4E>(v));
  v = fmax(v, dpp_mov<0x141>(v));
  v = fmax(v, dpp_mov<0x140>(v));
  return fmax(fmax(readlane_d(v, 0), readlane_d(v, 16)), fmax(readlane_d(v, 32), readlane_d(v, 48)));
}

// Workgroup barrier for LDS hand-offs only: waits for this wave's LDS traffic, NOT for outstanding
// global loads/stores (a __syncthreads() would drain vmcnt and expose HBM store latency every step).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// MATLAB max(x,0): NaN -> 0 (SURVEY C-3)
__device__ __forceinline__ double max0(double x) { return (x > 0.0) ? x : 0.0; }

// ---------------------------------------------------------------------------------------------
// 4x4 tile helpers (row-major t[4*i+j])
__device__ __forceinline__ void tile_zero(double* t) {
#pragma unroll
  for (int i = 0; i < 16; ++i) t[i] = 0.0;
}
__device__ __forceinline__ void tile_load(double* t, const double* __restrict__ p) {
  const double2* q = reinterpret_cast<const double2*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) { double2 v = q[i]; t[2 * i] = v.x; t[2 * i + 1] = v.y; }
}
__device__ __forceinline__ void tile_store(double* p, const double* t) {
  double2* q = reinterpret_cast<double2*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) q[i] = make_double2(t[2 * i], t[2 * i + 1]);
}
// c += a * b
__device__ __forceinline__ void tile_mma(double* c, const double* a, const double* b) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const double ail = a[4 * i + l];
#pragma unroll
      for (int j = 0; j < 4; ++j) c[4 * i + j] = fma(ail, b[4 * l + j], c[4 * i + j]);
    }
}
// c += a * b'
__device__ __forceinline__ void tile_mma_nt(double* c, const double* a, const double* b) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double s = c[4 * i + j];
#pragma unroll
      for (int l = 0; l < 4; ++l) s = fma(a[4 * i + l], b[4 * j + l], s);
      c[4 * i + j] = s;
    }
}
// c -= a * b
__device__ __forceinline__ void tile_mms(double* c, const double* a, const double* b) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const double ail = -a[4 * i + l];
#pragma unroll
      for (int j = 0; j < 4; ++j) c[4 * i + j] = fma(ail, b[4 * l + j], c[4 * i + j]);
    }
}
// c -= a * b'
__device__ __forceinline__ void tile_mms_nt(double* c, const double* a, const double* b) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double s = c[4 * i + j];
#pragma unroll
      for (int l = 0; l < 4; ++l) s = fma(-a[4 * i + l], b[4 * j + l], s);
      c[4 * i + j] = s;
    }
}
// t <- aI * t * aJ'   (block-diagonal congruence of one tile)
__device__ __forceinline__ void tile_congruence(double* t, const double* aI, const double* aJ) {
  double x[16];
  tile_zero(x);
  tile_mma(x, aI, t);
  tile_zero(t);
  tile_mma_nt(t, x, aJ);
}

// ---------------------------------------------------------------------------------------------
// mom: tilted-distribution moments by cubature.
//
// The sigma points of the reference's rules (fully symmetric sets, tensor Gauss-Hermite grids) take
// only a handful of distinct coordinate values (<= 5 for ut3/5/7/9), so link(mu_g + sqrt(s2_g)*xi) is
// evaluated once per (dimension, distinct value) -- cdim*nd transcendental evaluations instead of
// n_pts*cdim -- and every sigma point gathers its values through a byte code table.
struct MomCfg {
  int lik_kind;      // nagp_lik
  int link_kind;     // nagp_link
  double link_shift;
  int n_pts;
  int cdim;          // cubature dimension: N (NMF) or D (POWER)
  int D;             // sub-bands
  int DG;            // lanes per sigma point in phase 1 (power of two, <= 16, <= D)
  int nd;            // distinct unit coordinates (<= 64)
  const double* wn;  // [n_pts]
  const double* xd;  // [nd] distinct unit coordinate values
  const unsigned char* code;  // [n_pts][cdim] index into xd
  double jitter;
  int cache_tabs;    // keep wn / code in LDS (n_pts small enough)
  int store_a;       // NMF: keep a[p][d] = (link(xn) W')_d between the phases instead of link(xn)[p][:]
  unsigned long long* stamps;  // developer diagnostics: per-phase cycle sums of thread 0 (null in production)
};

__host__ __device__ inline int mom_chunk(const MomCfg& c) { return c.n_pts < 1024 ? c.n_pts : 1024; }
__host__ __device__ inline size_t mom_ws_core(const MomCfg& c) {
  const int CH = mom_chunk(c);
  const size_t row = (c.lik_kind != 0 && c.store_a) ? (size_t)c.D : (size_t)c.cdim;
  // rows[CH][row] + c0,c1,c2[CH] + sg[cdim] + lkv[cdim*nd] + sums1,sums2[nout] + xd[nd] + pad
  return (size_t)CH * (row + 3) + c.cdim + 3 * (size_t)c.cdim * c.nd + 2 * (size_t)(c.D + c.cdim + 1) + c.nd + 2;
}
__host__ __device__ inline size_t mom_lds_doubles(const MomCfg& c) {
  return mom_ws_core(c) + (c.cache_tabs ? (size_t)c.n_pts + ((size_t)c.n_pts * c.cdim + 7) / 8 + 1 : 0);
}
// called once per kernel after carving `ws`
__device__ inline void mom_cache_tables(const MomCfg& c, double* ws) {
  const int CH = mom_chunk(c);
  const size_t row = (c.lik_kind != 0 && c.store_a) ? (size_t)c.D : (size_t)c.cdim;
  double* xdl = ws + (size_t)CH * (row + 3) + c.cdim + 3 * (size_t)c.cdim * c.nd + 2 * (size_t)(c.D + c.cdim + 1);
  for (int i = threadIdx.x; i < c.nd; i += blockDim.x) xdl[i] = c.xd[i];
  if (!c.cache_tabs) return;
  double* tw = ws + mom_ws_core(c);
  unsigned char* tc = reinterpret_cast<unsigned char*>(tw + c.n_pts);
  for (int i = threadIdx.x; i < c.n_pts; i += blockDim.x) tw[i] = c.wn[i];
  for (int i = threadIdx.x; i < c.n_pts * c.cdim; i += blockDim.x) tc[i] = c.code[i];
}

__device__ __forceinline__ double link_eval(int kind, double shift, double g) {
  // literal reference formulas: log(1+exp(g-shift)) (demo_toy_modulators.m:16) / exp(g)
  return kind == 0 ? log(1.0 + exp(g - shift)) : exp(g);
}

// Workgroup-cooperative.  ALL threads of the block must call it (contains barriers).
//   mu, s2 : LDS, M = D + cdim entries (sub-bands first), must be visible (caller synchronised)
//   Wl     : LDS D x N row-major NMF weights (ignored for POWER)
//   out    : dl[M], d2l[M] (LDS) and *lZ (LDS scalar) valid after the function returns
//            (the function ends with a barrier).
// One copy of the code serves the three likelihoods: the variant flags are wave-uniform scalars
// (readfirstlane) so the compiler branches on them instead of evaluating e.g. the f64 sqrt of the
// sqrt-amplitude variant unconditionally; the cubature dimension (<= 8 for the NMF likelihoods) is a
// predicate on fully unrolled j loops, so link(xn) stays in statically indexed registers.
template <int CD>   // CD = cubature dimension for the NMF likelihoods (1..8); 0 = POWER (cdim = D)
__device__ __forceinline__ void mom_eval_impl(const MomCfg& c, const double* Wl, double sn2, double alpha,
                                              double y, const double* mu, const double* s2, double* ws, double* lZ, double* dl,
                                              double* d2l, unsigned long long* acc_st, double pEP) {
  const int tid = threadIdx.x, NT = blockDim.x;
  const int D = c.D, cd = (CD > 0) ? CD : c.cdim, CH = mom_chunk(c), nd = c.nd;
  const int nout = D + cd + 1;
  constexpr bool nmf = (CD > 0);
  const bool sq = __builtin_amdgcn_readfirstlane(c.lik_kind == 2 ? 1 : 0) != 0;
  const bool sta = __builtin_amdgcn_readfirstlane((c.lik_kind != 0 && c.store_a) ? 1 : 0) != 0;
  const bool TL = __builtin_amdgcn_readfirstlane(c.cache_tabs ? 1 : 0) != 0;
  const int rw = (!nmf || sta) ? D : cd;
  double* rows = ws;                    // [rw][CH]  a[d][p] (store_a / POWER) or link(xn)[j][p]: point index fastest
  double* c0 = rows + (size_t)CH * rw;  // [CH]
  double* c1 = c0 + CH;
  double* c2 = c1 + CH;
  double* sg = c2 + CH;                 // [cd] sqrt(s2_g)
  double* lkv = sg + cd;                // [cd][nd] link at the distinct coordinates
  double* xgv = lkv + (size_t)cd * nd;    // [cd][nd] (xn - mu_g)/s2_g at the distinct coordinates
  double* xg2v = xgv + (size_t)cd * nd;   // [cd][nd] xg^2 - 1/s2_g
  double* sums1 = xg2v + (size_t)cd * nd; // [nout]
  double* sums2 = sums1 + nout;         // [nout]
  const double* xdl = sums2 + nout;     // [nd]
  const double* lds_wn = ws + mom_ws_core(c);
  const unsigned char* lds_code = reinterpret_cast<const unsigned char*>(lds_wn + c.n_pts);
  const double* mu_z = mu;
  const double* mu_g = mu + D;
  const double* s2_z = s2;
  const double* s2_g = s2 + D;

  unsigned long long t_a = 0, t_b = 0;
#define NAGP_STAMP(slot) do { if (c.stamps && tid == 0) { t_b = __builtin_readcyclecounter(); acc_st[slot] += t_b - t_a; t_a = t_b; } } while (0)
  if (c.stamps && tid == 0) t_a = __builtin_readcyclecounter();
  // ---- phase 0/1a: link (and the modulator integrand factors) at the distinct coordinates of every dimension
  for (int t = tid; t < cd * nd; t += NT) {
    const int j = t / nd, ci = t - j * nd;
    const double sgj = sqrt(s2_g[j]);
    if (ci == 0) sg[j] = sgj;
    const double xn = mu_g[j] + sgj * xdl[ci];
    lkv[t] = link_eval(c.link_kind, c.link_shift, xn);
    const double xg = (xn - mu_g[j]) / s2_g[j];
    xgv[t] = xg;
    xg2v[t] = xg * xg - 1.0 / s2_g[j];
  }
  if (tid < nout) { sums1[tid] = 0.0; sums2[tid] = 0.0; }
  lds_barrier();
  NAGP_STAMP(0);

  const int DG = c.DG;
  const double sn2a = sn2 / alpha;
  for (int base = 0; base < c.n_pts; base += CH) {
    const int npc = (c.n_pts - base < CH) ? (c.n_pts - base) : CH;
    // ---- phase 1b: one sigma point per group of DG lanes
    for (int item = tid; item < npc * DG; item += NT) {
      const int pl = item / DG, sub = item - pl * DG;
      const int p = base + pl;
      double* row = rows + pl;
      double s2a[2] = {0, 0}, sma[2] = {0, 0};
      if (nmf) {
        constexpr int CDX = (CD > 0) ? CD : 1;
        double lkj[CDX];
        {
          const unsigned char* cp = (TL ? lds_code : c.code) + (size_t)p * cd;
#pragma unroll
          for (int j = 0; j < CDX; ++j) lkj[j] = lkv[j * nd + cp[j]];
        }
        // two d's per trip: the LDS reads of both are issued before the dependent FMAs
        int d = sub;
        for (; d + DG < D; d += 2 * DG) {
          double w[2][CDX], a[2];
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < CDX; ++j) w[u][j] = Wl[(d + u * DG) * CDX + j];
          const double sz0 = s2_z[d], sz1 = s2_z[d + DG], mz0 = mu_z[d], mz1 = mu_z[d + DG];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < CDX; ++j) acc = fma(w[u][j], lkj[j], acc);
            a[u] = acc;
          }
          if (sq) { a[0] = sqrt(a[0]); a[1] = sqrt(a[1]); }
          if (sta) { row[(size_t)d * CH] = a[0]; row[(size_t)(d + DG) * CH] = a[1]; }
          s2a[0] = fma(a[0] * a[0], sz0, s2a[0]); s2a[1] = fma(a[1] * a[1], sz1, s2a[1]);
          sma[0] = fma(a[0], mz0, sma[0]); sma[1] = fma(a[1], mz1, sma[1]);
        }
        for (; d < D; d += DG) {
          double acc = 0.0;
#pragma unroll
          for (int j = 0; j < CDX; ++j) acc = fma(Wl[d * CDX + j], lkj[j], acc);
          if (sq) acc = sqrt(acc);
          if (sta) row[(size_t)d * CH] = acc;
          s2a[0] = fma(acc * acc, s2_z[d], s2a[0]);
          sma[0] = fma(acc, mu_z[d], sma[0]);
        }
        if (!sta && sub == 0) {
#pragma unroll
          for (int j = 0; j < CDX; ++j) row[(size_t)j * CH] = lkj[j];
        }
      } else {
        const unsigned char* cp = (TL ? lds_code : c.code) + (size_t)p * cd;
        for (int d = sub; d < D; d += DG) {
          const double a = lkv[d * nd + cp[d]];
          row[(size_t)d * CH] = a;
          s2a[0] = fma(a * a, s2_z[d], s2a[0]);
          sma[0] = fma(a, mu_z[d], sma[0]);
        }
      }
      double sa2 = s2a[0] + s2a[1], sam = sma[0] + sma[1];
      sa2 = group_sum(sa2, DG);
      sam = group_sum(sam, DG);
      if (sub == 0) {
        const double sig2 = sn2a + sa2;
        const double sd = sqrt(sig2);
        const double r = (y - sam) / sd;
        const double pdf = exp(-0.5 * r * r) / (kSqrt2Pi * sd);
        const double w0 = (TL ? lds_wn[p] : c.wn[p]) * pdf;
        const double q = (y - sam) / sig2;
        c0[pl] = w0;
        c1[pl] = w0 * q;
        c2[pl] = w0 * (q * q - 1.0 / sig2);
      }
    }
    lds_barrier();
    NAGP_STAMP(1);
    // ---- phase 2: one output per 16-lane group (4 outputs per wave at a time); the group's lanes
    // stride over the points of the chunk; DPP sum inside the group (fixed order).  The modulator
    // outputs (heavier gathers) are handed out first.
    {
      const int grp = tid >> 4, gl = tid & 15, ngrp = NT >> 4;
      for (int oo = grp; oo < nout; oo += ngrp) {
        const int o = (oo < cd + 1) ? (D + oo) : (oo - cd - 1);   // modulators, Z, then the sub-bands
        double a1 = 0.0, a2 = 0.0, b1 = 0.0, b2 = 0.0;
        if (o < D) {
          if (nmf && !sta) {
            constexpr int CDX = (CD > 0) ? CD : 1;
            double wr[CDX];
#pragma unroll
            for (int j = 0; j < CDX; ++j) wr[j] = Wl[o * CDX + j];
            for (int pl = gl; pl < npc; pl += 16) {
              double a = 0.0;
#pragma unroll
              for (int j = 0; j < CDX; ++j) a = fma(wr[j], rows[(size_t)j * CH + pl], a);
              if (sq) a = sqrt(a);
              a1 = fma(a, c1[pl], a1);
              a2 = fma(a * a, c2[pl], a2);
            }
          } else {
            const double* ro = rows + (size_t)o * CH;
            int pl = gl;
            for (; pl + 16 < npc; pl += 32) {
              const double x0 = ro[pl], x1 = ro[pl + 16];
              a1 = fma(x0, c1[pl], a1); b1 = fma(x1, c1[pl + 16], b1);
              a2 = fma(x0 * x0, c2[pl], a2); b2 = fma(x1 * x1, c2[pl + 16], b2);
            }
            if (pl < npc) { const double x0 = ro[pl]; a1 = fma(x0, c1[pl], a1); a2 = fma(x0 * x0, c2[pl], a2); }
          }
        } else if (o < D + cd) {
          const int j = o - D;
          const unsigned char* cb = (TL ? lds_code : c.code) + (size_t)base * cd + j;
          int pl = gl;
          for (; pl + 16 < npc; pl += 32) {
            const int i0 = j * nd + cb[(size_t)pl * cd], i1 = j * nd + cb[(size_t)(pl + 16) * cd];
            a1 = fma(xgv[i0], c0[pl], a1); b1 = fma(xgv[i1], c0[pl + 16], b1);
            a2 = fma(xg2v[i0], c0[pl], a2); b2 = fma(xg2v[i1], c0[pl + 16], b2);
          }
          if (pl < npc) { const int i0 = j * nd + cb[(size_t)pl * cd]; a1 = fma(xgv[i0], c0[pl], a1); a2 = fma(xg2v[i0], c0[pl], a2); }
        } else {
          for (int pl = gl; pl < npc; pl += 16) a1 += c0[pl];
        }
        a1 += b1; a2 += b2;
        a1 = group_sum(a1, 16);
        a2 = group_sum(a2, 16);
        if (gl == 0) { sums1[o] += a1; sums2[o] += a2; }
      }
    }
    lds_barrier();
    NAGP_STAMP(2);
  }
  // ---- phase 3
  {
    const double Zs = sums1[D + cd];
    const double Z = pEP * ((Zs > c.jitter) ? Zs : c.jitter);  // max(NaN,jitter)=jitter
    const double Zinv = 1.0 / Z;
    if (tid < D + cd) {
      const double d1 = Zinv * pEP * sums1[tid];
      dl[tid] = d1;
      d2l[tid] = -d1 * d1 + Zinv * pEP * sums2[tid];
    }
    if (tid == 0) *lZ = log(Z);
  }
  lds_barrier();
  NAGP_STAMP(3);
#undef NAGP_STAMP
}

// Wl: NMF weights in LDS (D x N row-major)
// power-EP normaliser of likModulatorPreCalcwn.m:48 (1 for the other likelihoods); constant per kernel
__device__ inline double mom_pEP(const MomCfg& c, double sn2, double alpha) {
  return (c.lik_kind == 2) ? pow(2.0 * 3.14159265358979323846 * sn2, 0.5 * (1.0 - alpha)) / sqrt(alpha) : 1.0;
}
__device__ __forceinline__ void mom_eval(const MomCfg& c, const double* Wl, double pEP, double sn2, double alpha,
                                         double y, const double* mu, const double* s2, double* ws, double* lZ, double* dl,
                                         double* d2l, unsigned long long* acc_st = nullptr) {
  unsigned long long dummy_st[4];
  if (!acc_st) acc_st = dummy_st;
  const int sel = __builtin_amdgcn_readfirstlane(c.lik_kind == 0 ? 0 : c.cdim);
#ifdef NAGP_ONLY_CD
  (void)sel;
  mom_eval_impl<NAGP_ONLY_CD>(c, Wl, sn2, alpha, y, mu, s2, ws, lZ, dl, d2l, acc_st, pEP);
  return;
#endif
  switch (sel) {
    case 0: mom_eval_impl<0>(c, Wl, sn2, alpha, y, mu, s2, ws, lZ, dl, d2l, acc_st, pEP); break;
    case 1: mom_eval_impl<1>(c, Wl, sn2, alpha, y, mu, s2, ws, lZ, dl, d2l, acc_st, pEP); break;
    case 2: mom_eval_impl<2>(c, Wl, sn2, alpha, y, mu, s2, ws, lZ, dl, d2l, acc_st, pEP); break;
    case 3: mom_eval_impl<3>(c, Wl, sn2, alpha, y, mu, s2, ws, lZ, dl, d2l, acc_st, pEP); break;
    case 4: mom_eval_impl<4>(c, Wl, sn2, alpha, y, mu, s2, ws, lZ, dl, d2l, acc_st, pEP); break;
    case 5: mom_eval_impl<5>(c, Wl, sn2, alpha, y, mu, s2, ws, lZ, dl, d2l, acc_st, pEP); break;
    case 6: mom_eval_impl<6>(c, Wl, sn2, alpha, y, mu, s2, ws, lZ, dl, d2l, acc_st, pEP); break;
    case 7: mom_eval_impl<7>(c, Wl, sn2, alpha, y, mu, s2, ws, lZ, dl, d2l, acc_st, pEP); break;
    default: mom_eval_impl<8>(c, Wl, sn2, alpha, y, mu, s2, ws, lZ, dl, d2l, acc_st, pEP); break;
  }
}

}  // namespace nagp
