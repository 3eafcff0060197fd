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
struct MomCfg {
  int lik_kind;      // nagp_lik
  int link_kind;     // nagp_link
  double link_shift;
  int n_pts;
  int cdim;          // cubature dimension: N (NMF) or D (POWER)
  int D;             // sub-bands
  int DG;            // lanes per sigma point in phase 1 (power of two, <= 16, <= D)
  const double* wn;  // [n_pts]
  const double* xi;  // [n_pts][cdim] unit sigma points, point-major
  double jitter;
  int cache_tabs;    // copy wn / xi into LDS at kernel start (when the workgroup's LDS budget allows)
};

// LDS workspace (doubles) needed by mom_eval for a chunk of CH points
__host__ __device__ inline int mom_chunk(const MomCfg& c) { return c.n_pts < 1024 ? c.n_pts : 1024; }
__host__ __device__ inline size_t mom_lds_doubles(const MomCfg& c) {
  const int CH = mom_chunk(c);
  // lk[CH][cdim] + c0,c1,c2[CH] + sg[cdim] + sums1,sums2 [D+cdim+1]
  return (size_t)CH * (c.cdim + 3) + c.cdim + 2 * (size_t)(c.D + c.cdim + 1) + 2 +
         (c.cache_tabs ? (size_t)c.n_pts * (c.cdim + 1) : 0);
}
// kernels call this once after carving `ws`: returns where the cached tables live (end of the workspace)
__device__ inline void mom_cache_tables(const MomCfg& c, double* ws) {
  if (!c.cache_tabs) return;
  const int CH = mom_chunk(c);
  double* tw = ws + (size_t)CH * (c.cdim + 3) + c.cdim + 2 * (size_t)(c.D + c.cdim + 1) + 2;
  double* tx = tw + c.n_pts;
  for (int i = threadIdx.x; i < c.n_pts; i += blockDim.x) tw[i] = c.wn[i];
  for (int i = threadIdx.x; i < c.n_pts * c.cdim; i += blockDim.x) tx[i] = c.xi[i];
}

__device__ __forceinline__ double link_eval(int kind, double shift, double g) {
  // literal reference formulas: log(1+exp(g-shift)) (demo_toy_modulators.m:16) / exp(g)
  return kind == 0 ? log(1.0 + exp(g - shift)) : exp(g);
}


// phase-1 body for the NMF likelihoods with the cubature dimension as a compile-time constant
// (keeps link(xn) in registers; host enforces cdim <= 8 for these likelihoods)
template <int CD>
__device__ __forceinline__ void mom_p1_nmf(const MomCfg& c, const double* Wl, const double* mu_g, const double* sg,
                                           const double* xip, const double* mu_z, const double* s2_z, double* lkrow,
                                           int sub, int DG, int D, bool sq, double& sa2, double& sam) {
  double lkj[CD];
#pragma unroll
  for (int j = 0; j < CD; ++j) lkj[j] = link_eval(c.link_kind, c.link_shift, mu_g[j] + sg[j] * xip[j]);
  for (int d = sub; d < D; d += DG) {
    double a = 0.0;
#pragma unroll
    for (int j = 0; j < CD; ++j) a = fma(Wl[d * CD + j], lkj[j], a);
    if (sq) a = sqrt(a);
    sa2 = fma(a * a, s2_z[d], sa2);
    sam = fma(a, mu_z[d], sam);
  }
  if (sub == 0) {
#pragma unroll
    for (int j = 0; j < CD; ++j) lkrow[j] = lkj[j];
  }
}

// Workgroup-cooperative.  ALL threads of the block must call it (contains __syncthreads).
//   mu, s2 : LDS, M = D + cdim entries (sub-bands first), must be visible (caller synchronised)
//   Wl     : LDS D x N row-major NMF weights (ignored for POWER)
//   out    : dl[M], d2l[M] (LDS) and *lZ (LDS scalar) valid after the function returns
//            (the function ends with a __syncthreads()).
template <bool TL>
__device__ __forceinline__ void mom_eval_t(const MomCfg& c, const double* Wl, double sn2, double alpha, double y,
                                           const double* mu, const double* s2, double* ws, double* lZ, double* dl,
                                           double* d2l) {
  const int tid = threadIdx.x, NT = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nwaves = NT >> 6;
  const int D = c.D, cd = c.cdim, CH = mom_chunk(c);
  const int nout = D + cd + 1;
  const bool nmf = (c.lik_kind != 0);
  const bool sq = (c.lik_kind == 2);
  double* lk = ws;                    // [CH][cd]
  double* c0 = lk + (size_t)CH * cd;  // [CH]
  double* c1 = c0 + CH;
  double* c2 = c1 + CH;
  double* sg = c2 + CH;               // [cd] sqrt(s2_g)
  double* sums1 = sg + cd;            // [nout]
  double* sums2 = sums1 + nout;       // [nout]
  const double* t_wn = TL ? (sums2 + nout + 2) : c.wn;       // cached tables sit right behind the workspace
  const double* t_xi = TL ? (sums2 + nout + 2 + c.n_pts) : c.xi;
  const double* mu_z = mu;
  const double* mu_g = mu + D;
  const double* s2_z = s2;
  const double* s2_g = s2 + D;

  if (tid < cd) sg[tid] = sqrt(s2_g[tid]);
  if (tid < nout) { sums1[tid] = 0.0; sums2[tid] = 0.0; }
  lds_barrier();

  const int DG = c.DG;
  const double sn2a = sn2 / alpha;
  for (int base = 0; base < c.n_pts; base += CH) {
    const int npc = (c.n_pts - base < CH) ? (c.n_pts - base) : CH;
    // ---- phase 1: one sigma point per group of DG lanes
    for (int item = tid; item < npc * DG; item += NT) {
      const int pl = item / DG, sub = item - pl * DG;
      const int p = base + pl;
      const double* xip = t_xi + (size_t)p * cd;
      double sa2 = 0.0, sam = 0.0;
      if (nmf) {
        switch (cd) {
          case 1: mom_p1_nmf<1>(c, Wl, mu_g, sg, xip, mu_z, s2_z, lk + (size_t)pl * cd, sub, DG, D, sq, sa2, sam); break;
          case 2: mom_p1_nmf<2>(c, Wl, mu_g, sg, xip, mu_z, s2_z, lk + (size_t)pl * cd, sub, DG, D, sq, sa2, sam); break;
          case 3: mom_p1_nmf<3>(c, Wl, mu_g, sg, xip, mu_z, s2_z, lk + (size_t)pl * cd, sub, DG, D, sq, sa2, sam); break;
          case 4: mom_p1_nmf<4>(c, Wl, mu_g, sg, xip, mu_z, s2_z, lk + (size_t)pl * cd, sub, DG, D, sq, sa2, sam); break;
          case 5: mom_p1_nmf<5>(c, Wl, mu_g, sg, xip, mu_z, s2_z, lk + (size_t)pl * cd, sub, DG, D, sq, sa2, sam); break;
          case 6: mom_p1_nmf<6>(c, Wl, mu_g, sg, xip, mu_z, s2_z, lk + (size_t)pl * cd, sub, DG, D, sq, sa2, sam); break;
          case 7: mom_p1_nmf<7>(c, Wl, mu_g, sg, xip, mu_z, s2_z, lk + (size_t)pl * cd, sub, DG, D, sq, sa2, sam); break;
          default: mom_p1_nmf<8>(c, Wl, mu_g, sg, xip, mu_z, s2_z, lk + (size_t)pl * cd, sub, DG, D, sq, sa2, sam); break;
        }
      } else {
        for (int d = sub; d < D; d += DG) {
          const double xn = mu_g[d] + sg[d] * xip[d];
          const double a = link_eval(c.link_kind, c.link_shift, xn);
          lk[(size_t)pl * cd + d] = a;
          sa2 = fma(a * a, s2_z[d], sa2);
          sam = fma(a, mu_z[d], sam);
        }
      }
      sa2 = group_sum(sa2, DG);
      sam = group_sum(sam, DG);
      if (sub == 0) {
        const double sig2 = sn2a + sa2;
        const double sd = sqrt(sig2);
        const double r = (y - sam) / sd;
        const double pdf = exp(-0.5 * r * r) / (kSqrt2Pi * sd);
        const double w0 = t_wn[p] * pdf;
        const double q = (y - sam) / sig2;
        c0[pl] = w0;
        c1[pl] = w0 * q;
        c2[pl] = w0 * (q * q - 1.0 / sig2);
      }
    }
    lds_barrier();
    // ---- phase 2: one output per wave pass, lanes stride over the points of the chunk
    for (int o = wave; o < nout; o += nwaves) {
      double a1 = 0.0, a2 = 0.0;
      if (o < D) {
        for (int pl = lane; pl < npc; pl += 64) {
          double a;
          if (nmf) {
            a = 0.0;
            for (int j = 0; j < cd; ++j) a = fma(Wl[o * cd + j], lk[(size_t)pl * cd + j], a);
            if (sq) a = sqrt(a);
          } else {
            a = lk[(size_t)pl * cd + o];
          }
          a1 = fma(a, c1[pl], a1);
          a2 = fma(a * a, c2[pl], a2);
        }
      } else if (o < D + cd) {
        const int j = o - D;
        const double mg = mu_g[j], s2g = s2_g[j], sgj = sg[j];
        for (int pl = lane; pl < npc; pl += 64) {
          const double xn = mg + sgj * t_xi[(size_t)(base + pl) * cd + j];
          const double xg = (xn - mg) / s2g;
          a1 = fma(xg, c0[pl], a1);
          a2 = fma(xg * xg - 1.0 / s2g, c0[pl], a2);
        }
      } else {
        for (int pl = lane; pl < npc; pl += 64) a1 += c0[pl];
      }
      a1 = wave_sum(a1);
      a2 = wave_sum(a2);
      if (lane == 0) { sums1[o] += a1; sums2[o] += a2; }
    }
    lds_barrier();
  }
  // ---- phase 3
  {
    double pEP = 1.0;
    if (sq) pEP = pow(2.0 * 3.14159265358979323846 * sn2, 0.5 * (1.0 - alpha)) / sqrt(alpha);
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
}

__device__ __forceinline__ void mom_eval(const MomCfg& c, const double* Wl, double sn2, double alpha, double y,
                                         const double* mu, const double* s2, double* ws, double* lZ, double* dl,
                                         double* d2l) {
  if (c.cache_tabs) mom_eval_t<true>(c, Wl, sn2, alpha, y, mu, s2, ws, lZ, dl, d2l);
  else mom_eval_t<false>(c, Wl, sn2, alpha, y, mu, s2, ws, lZ, dl, d2l);
}

}  // namespace nagp
