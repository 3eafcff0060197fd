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

typedef double v4d __attribute__((ext_vector_type(4)));   // accumulator of v_mfma_f64_16x16x4_f64

constexpr int MAXM = 64;        // sites (= diagonal blocks) per step
constexpr double kSqrt2Pi = 2.5066282746310002;
constexpr double kInvSqrt2Pi = 0.3989422804014327;

// 1/sqrt(s) to about one ulp: hardware estimate and two Newton steps (an f64 sqrt followed by a division is a chain of
// ~30 dependent instructions, this one of 8).  s = +inf -> 0 like 1/sqrt(inf); s <= 0 or NaN -> NaN (0 -> NaN: the
// formulas below divide 0 by 0 there as well).
__device__ __forceinline__ double rsqrt_nr(double s) {
  double r = __builtin_amdgcn_rsq(s);
  double e = fma(-s * r, r, 1.0); r = fma(0.5 * r, e, r);
  e = fma(-s * r, r, 1.0); r = fma(0.5 * r, e, r);
  return (s == __builtin_inf()) ? 0.0 : r;
}
// Gaussian weight of one sigma point: pdf = N(y; sam, sig2), q = (y - sam)/sig2, inv = 1/sig2 -- normpdf(y,mu,sigma),
// (y-mu)./sigma2 and 1./sigma2 of likModulatorNMFPower.m:52-70 through ONE reciprocal square root instead of a square root
// and three divisions (same values to ~2 ulp; the per-point phase of mom was half divisions)
__device__ __forceinline__ void gauss_terms(double y, double sam, double sig2, double& pdf, double& q, double& inv) {
  const double rs = rsqrt_nr(sig2);
  inv = rs * rs;
  const double dy = y - sam;
  q = dy * inv;
  pdf = exp(-0.5 * dy * q) * (rs * kInvSqrt2Pi);
}

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
// Block-structured Wnmf (source-separation mixtures, experiments/gf_ep_mods_nmf_mixture.m:101: Wnmf = blkdiag(W_j)):
// sub-band d of source j only sees the components of source j, so a_d takes one value per DISTINCT projection of the
// sigma points onto those components (77 for ut9 in 3 of 9 dimensions, against 3 973 points).  Host-built tables:
constexpr int MOM_MAXSRC = 4;
struct MomSrc {
  int n_src;                   // < 2: unstructured
  int d0[MOM_MAXSRC + 1];      // sub-band range of every source
  int n0[MOM_MAXSRC + 1];      // component range of every source
  int nb[MOM_MAXSRC];          // distinct projected points ("tuples") per source
  int nbmax, dlmax, n_items;   // max tuples / sub-bands per source ; work items of the binning pass
  // one byte blob (copied into LDS by mom_cache_tables; every table starts on a 16-byte boundary):
  //   off[0] tup    u8  [n_src][n_pts]       tuple of every sigma point
  //   off[1] tcode  u8  [n_src][nbmax][cdim] coordinate codes of the tuple's components
  //   off[2] perm   u16 [n_src][n_pts]       sigma points ordered by tuple
  //   off[3] items  i32 [n_items][4]         source, tuple, start in perm, length ; longest first
  //   off[4] bin_first i32 [n_src*nbmax + 1] CSR of the items of every bin ...
  //   off[5] bitem  i32 [n_items]            ... their positions in `items`, fixed order
  const unsigned char* blob;
  int blob_bytes;
  int off[6];
};
typedef const unsigned short __attribute__((address_space(3))) * lds_u16p;
typedef const int __attribute__((address_space(3))) * lds_i32p;

// sparse-point form of likModulatorNMFPower (nagp_momsp.hpp); tables built by the host
constexpr int MSP_NZ = 4;     // non-centre coordinates per sigma point
struct MomSp {
  int enabled;
  int c0;             // code of the coordinate value 0
  int nzmax;          // largest number of non-centre coordinates of a sigma point
  const int* pdesc;   // [n_pts][MSP_NZ]: j*nd + c of the non-centre coordinates, -1 = none
};

struct MomCfg {
  int lik_kind;      // nagp_lik
  int link_kind;     // nagp_link
  double link_shift;
  int n_pts;
  int cdim;          // cubature dimension: N (NMF) or D (POWER)
  int D;             // sub-bands
  int DG;            // lanes per sigma point (power of two, 4..16 for the NMF likelihoods; <= 16, <= D for POWER)
  int nd;            // distinct unit coordinates (<= 64)
  const double* wn;  // [n_pts]
  const double* xd;  // [nd] distinct unit coordinate values
  const unsigned char* code;  // [n_pts][cdim] index into xd
  double jitter;
  int cache_tabs;    // keep wn / code in LDS (n_pts small enough)
  int store_a;       // POWER_NMF_SQRT: keep a[d][p] = sqrt(W_d . link(xn_p)) in LDS between the phases (one sqrt per (p, d))
  unsigned long long* stamps;  // developer diagnostics: per-phase cycle sums of thread 0 (null in production)
  MomSrc src;        // n_src >= 2: mom_src instead of the per-point evaluation (needs cache_tabs)
  int chunk_cap;     // sigma points per pass of the per-point evaluation (0: 1024); smaller for LDS-tight launches (the per-point arrays are 3+ doubles per point)
  MomSp sp;          // enabled: the kernels that have the staged form (nagp_momsp.hpp) use it instead of mom_eval
  int sq_form;       // likModulatorPreCalcwn in the staged form of nagp_momsq.hpp (sp.c0 = code of the centre coordinate)
};

typedef const unsigned char __attribute__((address_space(3))) * lds_u8p;   // explicit LDS pointers: a select between
typedef const double __attribute__((address_space(3))) * lds_f64p;          // an LDS and a global table must not decay to FLAT loads
constexpr int MOM_NDM = 4;    // NMF likelihoods: sub-bands per lane and pass (their W rows live in registers)
constexpr int MOM_MAXW = 8;   // waves per workgroup the cross-wave reduction buffers are sized for (launch bounds <= 512)

__host__ __device__ inline int mom_chunk(const MomCfg& c) { const int cap = c.chunk_cap > 0 ? c.chunk_cap : 1024; return c.n_pts < cap ? c.n_pts : cap; }
// LDS workspace layout (offsets in doubles)
constexpr int MOM_REP = 4;    // POWER_NMF: replicas (16-lane groups) per phase-2 task
struct MomLay { size_t rows, c0, c1, c2, sg, lkv, xgv, xg2v, sums1, sums2, part, acc, qv, xd, core, tw, tc, total, atab, s2b, smb, cb, ipart; };
__host__ __device__ inline int mom_nslots(int cd) { return cd * cd + 3 * cd + 1; }   // u, R, g1, g2, Z
__host__ __device__ inline MomLay mom_layout(const MomCfg& c) {
  MomLay l;
  const bool st = c.src.n_src >= 2;   // structured: no per-point arrays, no partial-sum buffers
  const size_t CH = st ? 0 : (size_t)mom_chunk(c), nout = (size_t)c.D + c.cdim + 1, tab = (size_t)c.cdim * c.nd;
  const size_t ns = (size_t)mom_nslots(c.cdim);
  l.rows = 0;   // POWER: link(xn)[d][p] ; POWER_NMF: link(xn)[j][p] ; POWER_NMF_SQRT: a[d][p] (store_a) or none
  l.c0 = (c.lik_kind == 0) ? CH * c.D : (c.lik_kind == 1) ? CH * c.cdim : (c.store_a ? CH * c.D : 0);
  l.c1 = l.c0 + CH; l.c2 = l.c1 + CH;
  l.sg = l.c2 + CH;
  l.lkv = l.sg + c.cdim; l.xgv = l.lkv + tab; l.xg2v = l.xgv + tab;
  l.sums1 = l.xg2v + tab; l.sums2 = l.sums1 + nout;
  l.part = l.sums2 + nout;   // POWER_NMF_SQRT: [MOM_MAXW][2*nout] ; POWER_NMF: [MOM_MAXW][16x16] (N <= 7) or [MOM_REP][ns]
  l.acc = l.part + (st ? 0 : (c.lik_kind == 0) ? 0 : (c.lik_kind == 1) ? (size_t)(c.cdim <= 7 ? MOM_MAXW * 256 : MOM_REP * ns) : (size_t)MOM_MAXW * 2 * nout);
  l.qv = l.acc + ((c.lik_kind == 1 && !st) ? ns : 0);                    // POWER_NMF: Q = W' diag(s2_z) W [cd][cd], v = W' mu_z [cd]
  l.xd = l.qv + ((c.lik_kind == 1 && !st) ? (size_t)c.cdim * (c.cdim + 1) : 0);
  l.xd = (l.xd + 1) & ~(size_t)1;
  l.core = (l.xd + c.nd + 1) & ~(size_t)1;
  l.tw = l.core;
  l.tc = l.tw + c.n_pts;
  l.total = c.cache_tabs ? l.tc + ((size_t)c.n_pts * c.cdim + 7) / 8 + 1 : l.core;
  l.atab = l.s2b = l.smb = l.cb = l.ipart = l.total;
  if (st) {   // wn at tw, the table blob at tc (always resident), then the per-call tuple tables
    const size_t nbin = (size_t)c.src.n_src * c.src.nbmax;
    l.atab = l.tc + ((size_t)c.src.blob_bytes + 7) / 8 + 1;         // a[j][b][d_local]
    l.s2b = l.atab + nbin * c.src.dlmax;                            // sum_d a^2 s2_d per (source, tuple)
    l.smb = l.s2b + nbin;                                           // sum_d a mu_d
    l.cb = l.smb + nbin;                                            // binned weights c0, c1, c2
    l.ipart = l.cb + 3 * nbin;                                      // per-item partial sums
    l.total = l.ipart + 3 * (size_t)c.src.n_items;
  }
  return l;
}
__host__ __device__ inline size_t mom_lds_doubles(const MomCfg& c) { return mom_layout(c).total; }
// called once per kernel after carving `ws`
__device__ inline void mom_cache_tables(const MomCfg& c, double* ws) {
  const MomLay l = mom_layout(c);
  for (int i = threadIdx.x; i < c.nd; i += blockDim.x) ws[l.xd + i] = c.xd[i];
  if (c.src.n_src >= 2) {
    double* tw = ws + l.tw;
    unsigned int* tt = reinterpret_cast<unsigned int*>(ws + l.tc);
    const unsigned int* sb = reinterpret_cast<const unsigned int*>(c.src.blob);
    for (int i = threadIdx.x; i < c.n_pts; i += blockDim.x) tw[i] = c.wn[i];
    for (int i = threadIdx.x; i < c.src.blob_bytes / 4; i += blockDim.x) tt[i] = sb[i];
    return;
  }
  if (!c.cache_tabs) return;
  double* tw = ws + l.tw;
  unsigned char* tc = reinterpret_cast<unsigned char*>(ws + l.tc);
  for (int i = threadIdx.x; i < c.n_pts; i += blockDim.x) tw[i] = c.wn[i];
  for (int i = threadIdx.x; i < c.n_pts * c.cdim; i += blockDim.x) tc[i] = c.code[i];
}

__device__ __forceinline__ double link_eval(int kind, double shift, double g) {
  // literal reference formulas: log(1+exp(g-shift)) (demo_toy_modulators.m:16) / exp(g)
  return kind == 0 ? log(1.0 + exp(g - shift)) : exp(g);
}

#define NAGP_STAMP(slot) do { if (c.stamps && tid == 0) { t_b = __builtin_readcyclecounter(); acc_st[slot] += t_b - t_a; t_a = t_b; } } while (0)

// phase 1a (all likelihoods): link and the modulator integrand factors at the distinct coordinates of
// every cubature dimension; zero the output sums.  Ends with a barrier.
__device__ __forceinline__ void mom_phase1a(const MomCfg& c, const MomLay& l, int cd, int nout, const double* mu_g,
                                            const double* s2_g, double* ws) {
  const int tid = threadIdx.x, NT = blockDim.x, nd = c.nd;
  for (int t = tid; t < cd * nd; t += NT) {
    const int j = t / nd, ci = t - j * nd;
    const double sgj = sqrt(s2_g[j]);
    if (ci == 0) ws[l.sg + j] = sgj;
    const double xn = mu_g[j] + sgj * ws[l.xd + ci];
    ws[l.lkv + t] = link_eval(c.link_kind, c.link_shift, xn);
    const double xg = (xn - mu_g[j]) / s2_g[j];
    ws[l.xgv + t] = xg;
    ws[l.xg2v + t] = xg * xg - 1.0 / s2_g[j];
  }
  if (tid < nout) { ws[l.sums1 + tid] = 0.0; ws[l.sums2 + tid] = 0.0; }
  lds_barrier();
}
// phase 3 (all likelihoods): Z, d lZ, d2 lZ from the weighted sums.  Ends with a barrier.
template <bool LOGZ>
__device__ __forceinline__ void mom_phase3(const MomCfg& c, const MomLay& l, int nsite, double pEP, const double* ws,
                                           double* lZ, double* dl, double* d2l) {
  const int tid = threadIdx.x;
  const double Zs = ws[l.sums1 + nsite];
  const double Z = pEP * ((Zs > c.jitter) ? Zs : c.jitter);  // max(NaN,jitter)=jitter
  const double Zinv = 1.0 / Z;
  if (tid < nsite) {
    const double d1 = Zinv * pEP * ws[l.sums1 + tid];
    dl[tid] = d1;
    d2l[tid] = -d1 * d1 + Zinv * pEP * ws[l.sums2 + tid];
  }
  if (tid == 0) *lZ = LOGZ ? log(Z) : Z;
  lds_barrier();
}

// Workgroup-cooperative.  ALL threads of the block must call it (contains barriers).
//   mu, s2 : LDS, M = D + cdim entries (sub-bands first), must be visible (caller synchronised)
//   Wl     : LDS D x N row-major NMF weights (ignored for POWER)
//   out    : dl[M], d2l[M] (LDS) and *lZ (LDS scalar) valid after the function returns
//            (the function ends with a barrier).
//
// NMF likelihoods (likModulatorNMFPower.m:28-87, likModulatorPreCalcwn.m:28-86), cubature dimension CD = N:
//   1b  lane (p, sub) of a DG-lane group: a_d = link(xn_p) . W_d for its <= MOM_NDM sub-bands (W rows in
//       registers, loaded once per call), partial sum_d a_d^2 s2_d and sum_d a_d mu_d, DPP sum over the group
//   1c  one lane per sigma point: sigma^2, N(y; mu, sigma^2), the three weights c0, c1, c2
//   2   lanes regrouped so that the 64/DG lanes sharing a sub-band set are adjacent: every lane runs over its
//       points (a_d recomputed from the registers -- cheaper than an LDS round trip), DPP sum over the
//       adjacent lanes, one LDS partial per wave, fixed-order sum over the waves.
//       Lane set j <= CD of the same pass accumulates the modulator outputs / Z.
template <int CD, bool LOGZ>
__device__ __forceinline__ void mom_nmf(const MomCfg& c, const double* Wl, double sn2, double alpha, double y,
                                        const double* mu, const double* s2, double* ws, double* lZ, double* dl, double* d2l,
                                        unsigned long long* acc_st, double pEP) {
  const int tid = threadIdx.x, NT = blockDim.x, lane = tid & 63, wave = tid >> 6, NW = NT >> 6;
  const int D = c.D, CH = mom_chunk(c), nd = c.nd;
  constexpr int cd = CD;
  const int nout = D + cd + 1;
  const bool sq = __builtin_amdgcn_readfirstlane(c.lik_kind == 2 ? 1 : 0) != 0;
  const bool TL = __builtin_amdgcn_readfirstlane(c.cache_tabs ? 1 : 0) != 0;
  const bool sta = __builtin_amdgcn_readfirstlane(c.store_a ? 1 : 0) != 0;
  const int DG = __builtin_amdgcn_readfirstlane(c.DG);
  const int lgDG = 31 - __builtin_clz(DG);
  const int G = 64 >> lgDG;       // adjacent lanes per sub-band set in phase 2 (<= 16)
  const MomLay l = mom_layout(c);
  double* rows = ws + l.rows;           // [D][CH] a[d][p] (store_a)
  double* c0 = ws + l.c0;
  double* c1 = ws + l.c1;
  double* c2 = ws + l.c2;
  const double* lkv = ws + l.lkv;
  const double* xgv = ws + l.xgv;
  const double* xg2v = ws + l.xg2v;
  double* sums1 = ws + l.sums1;
  double* sums2 = ws + l.sums2;
  double* part = ws + l.part;
  const lds_f64p lds_wn = (lds_f64p)(ws + l.tw);
  const lds_u8p lds_code = (lds_u8p)(ws + l.tc);
  const unsigned char* g_code = c.code;
  const double* mu_z = mu;
  const double* s2_z = s2;

  unsigned long long t_a = 0, t_b = 0;
  if (c.stamps && tid == 0) t_a = __builtin_readcyclecounter();
  mom_phase1a(c, l, cd, nout, mu + D, s2 + D, ws);
  NAGP_STAMP(0);

  const double sn2a = sn2 / alpha;
  const int sub = tid & (DG - 1);
  const int sub2 = lane >> (6 - lgDG), slot = lane & (G - 1);
  for (int base = 0; base < c.n_pts; base += CH) {
    const int npc = (c.n_pts - base < CH) ? (c.n_pts - base) : CH;
    // ---- phase 1b
    for (int dc = 0; dc < D; dc += MOM_NDM * DG) {
      double w[MOM_NDM][CD], sz[MOM_NDM], mz[MOM_NDM];
#pragma unroll
      for (int u = 0; u < MOM_NDM; ++u) {
        const int d = dc + sub + u * DG;
        const bool ok = d < D;
        const int dd = ok ? d : 0;
#pragma unroll
        for (int j = 0; j < CD; ++j) { const double v = Wl[dd * CD + j]; w[u][j] = ok ? v : 0.0; }
        const double a_ = s2_z[dd], b_ = mu_z[dd];
        sz[u] = ok ? a_ : 0.0; mz[u] = ok ? b_ : 0.0;
      }
      for (int item = tid; item < (npc << lgDG); item += NT) {
        const int pl = item >> lgDG;
        int cj[CD];
        if (TL) {
#pragma unroll
          for (int j = 0; j < CD; ++j) cj[j] = lds_code[(base + pl) * CD + j];
        } else {
#pragma unroll
          for (int j = 0; j < CD; ++j) cj[j] = g_code[(size_t)(base + pl) * CD + j];
        }
        double lkj[CD];
#pragma unroll
        for (int j = 0; j < CD; ++j) lkj[j] = lkv[j * nd + cj[j]];
        double s2a = 0.0, sma = 0.0;
#pragma unroll
        for (int u = 0; u < MOM_NDM; ++u) {
          double a = 0.0;
#pragma unroll
          for (int j = 0; j < CD; ++j) a = fma(w[u][j], lkj[j], a);
          if (sq) { a = sqrt(a); asm volatile("" : "+v"(a)); }   // the asm keeps the f64 sqrt from being if-converted
          if (sta) { const int d = dc + sub + u * DG; if (d < D) rows[(size_t)d * CH + pl] = a; }
          s2a = fma(a * a, sz[u], s2a);
          sma = fma(a, mz[u], sma);
        }
        s2a = group_sum(s2a, DG);
        sma = group_sum(sma, DG);
        if (sub == 0) {
          if (dc == 0) { c1[pl] = s2a; c2[pl] = sma; }
          else { c1[pl] += s2a; c2[pl] += sma; }
        }
      }
    }
    lds_barrier();
    // ---- phase 1c
    for (int pl = tid; pl < npc; pl += NT) {
      const double sig2 = sn2a + c1[pl], sam = c2[pl];
      double pdf, q, inv;
      gauss_terms(y, sam, sig2, pdf, q, inv);
      double wq;
      if (TL) wq = lds_wn[base + pl]; else wq = c.wn[base + pl];
      const double w0 = wq * pdf;
      c0[pl] = w0;
      c1[pl] = w0 * q;
      c2[pl] = w0 * (q * q - inv);
    }
    lds_barrier();
    NAGP_STAMP(1);
    // ---- phase 2
    for (int dc = 0; dc < D; dc += MOM_NDM * DG) {
      double w[MOM_NDM][CD];
#pragma unroll
      for (int u = 0; u < MOM_NDM; ++u) {
        const int d = dc + sub2 + u * DG;
        const bool ok = d < D;
        const int dd = ok ? d : 0;
#pragma unroll
        for (int j = 0; j < CD; ++j) { const double v = Wl[dd * CD + j]; w[u][j] = ok ? v : 0.0; }
      }
      double a1[MOM_NDM], a2[MOM_NDM], g1 = 0.0, g2 = 0.0;
#pragma unroll
      for (int u = 0; u < MOM_NDM; ++u) { a1[u] = 0.0; a2[u] = 0.0; }
      const bool modl = (dc == 0) && (sub2 <= cd);   // this lane set also owns modulator output sub2 (or Z)
      const int jg = (sub2 < cd) ? sub2 : 0;
      int dsel[MOM_NDM];      // rows of the stored a (clamped; the W = 0 padding of phase 1b left a = 0 unwritten)
      double dok[MOM_NDM];
#pragma unroll
      for (int u = 0; u < MOM_NDM; ++u) { const int d = dc + sub2 + u * DG; dsel[u] = (d < D) ? d : 0; dok[u] = (d < D) ? 1.0 : 0.0; }
      for (int pl = wave * G + slot; pl < npc; pl += NW * G) {
        const double c0p = c0[pl], c1p = c1[pl], c2p = c2[pl];
        if (sta) {
#pragma unroll
          for (int u = 0; u < MOM_NDM; ++u) {
            const double a = rows[(size_t)dsel[u] * CH + pl] * dok[u];
            a1[u] = fma(a, c1p, a1[u]);
            a2[u] = fma(a * a, c2p, a2[u]);
          }
        } else {
          int cj[CD];
          if (TL) {
#pragma unroll
            for (int j = 0; j < CD; ++j) cj[j] = lds_code[(base + pl) * CD + j];
          } else {
#pragma unroll
            for (int j = 0; j < CD; ++j) cj[j] = g_code[(size_t)(base + pl) * CD + j];
          }
          double lkj[CD];
#pragma unroll
          for (int j = 0; j < CD; ++j) lkj[j] = lkv[j * nd + cj[j]];
#pragma unroll
          for (int u = 0; u < MOM_NDM; ++u) {
            double a = 0.0;
#pragma unroll
            for (int j = 0; j < CD; ++j) a = fma(w[u][j], lkj[j], a);
            if (sq) { a = sqrt(a); asm volatile("" : "+v"(a)); }   // the asm keeps the f64 sqrt from being if-converted
            a1[u] = fma(a, c1p, a1[u]);
            a2[u] = fma(a * a, c2p, a2[u]);
          }
        }
        if (modl) {
          if (sub2 < cd) {
            int cg;
            if (TL) cg = lds_code[(base + pl) * CD + jg]; else cg = g_code[(size_t)(base + pl) * CD + jg];
            const int ix = jg * nd + cg;
            g1 = fma(xgv[ix], c0p, g1);
            g2 = fma(xg2v[ix], c0p, g2);
          } else {
            g1 += c0p;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < MOM_NDM; ++u) { a1[u] = group_sum(a1[u], G); a2[u] = group_sum(a2[u], G); }
      g1 = group_sum(g1, G);
      g2 = group_sum(g2, G);
      if (slot == 0) {
        double* pw = part + (size_t)wave * 2 * nout;
#pragma unroll
        for (int u = 0; u < MOM_NDM; ++u) {
          const int d = dc + sub2 + u * DG;
          if (d < D) { pw[d] = a1[u]; pw[nout + d] = a2[u]; }
        }
        if (modl) { pw[D + sub2] = g1; pw[nout + D + sub2] = g2; }
      }
      lds_barrier();
      for (int o = tid; o < nout; o += NT) {
        const bool mine = (o < D) ? (o >= dc && o < dc + MOM_NDM * DG) : (dc == 0);
        if (mine) {
          double s1 = 0.0, s2_ = 0.0;
          for (int wv = 0; wv < NW; ++wv) { s1 += part[(size_t)wv * 2 * nout + o]; s2_ += part[(size_t)wv * 2 * nout + nout + o]; }
          sums1[o] += s1; sums2[o] += s2_;
        }
      }
      lds_barrier();
    }
    NAGP_STAMP(2);
  }
  mom_phase3<LOGZ>(c, l, D + cd, pEP, ws, lZ, dl, d2l);
  NAGP_STAMP(3);
}

// POWER_NMF (likModulatorNMFPower.m:28-87) without the sqrt amplitude: a_d = W_d . lk is linear in
// lk = link(xn), so the sub-band sums over d collapse to N x N forms that are built once per call:
//   sum_d a_d^2 s2_d = lk' Q lk,  Q = W' diag(s2_z) W ;   sum_d a_d mu_d = lk . v,  v = W' mu_z
//   sum_p c1_p a_d   = W_d . u,   u = sum_p c1_p lk_p ;   sum_p c2_p a_d^2 = W_d' R W_d,  R = sum_p c2_p lk_p lk_p'
// (same arithmetic up to summation order; n_pts*(N^2+..) instead of n_pts*D*N multiply-adds).
//   1a  link tables (wave 0) | Q, v by 8-lane groups of the other waves
//   1b  one lane per sigma point: gather lk, quadratic form, Gaussian weight, c0 c1 c2
//   2   16-lane groups: task j < N accumulates u_j and R_{j,j'>=j}; task N+j the modulator sums (task N also Z);
//       up to MOM_REP groups share a task, fixed-order sums of the partials
//   3   thread d: W_d . u, W_d' R W_d ; thread D+j: modulator outputs ; Z
template <int CD, bool LOGZ>
__device__ __forceinline__ void mom_quad(const MomCfg& c, const double* Wl, double sn2, double alpha, double y,
                                         const double* mu, const double* s2, double* ws, double* lZ, double* dl, double* d2l,
                                         unsigned long long* acc_st, double pEP) {
  const int tid = threadIdx.x, NT = blockDim.x;
  const int D = c.D, CH = mom_chunk(c), nd = c.nd;
  constexpr int cd = CD;
  constexpr int NS = CD * CD + 3 * CD + 1;
  constexpr int oR = CD, oG1 = CD + CD * CD, oG2 = oG1 + CD, oZ = oG2 + CD;
  const int nout = D + cd + 1;
  const bool TL = __builtin_amdgcn_readfirstlane(c.cache_tabs ? 1 : 0) != 0;
  const MomLay l = mom_layout(c);
  double* lkp = ws + l.rows;            // [cd][CH]
  double* c0 = ws + l.c0;
  double* c1 = ws + l.c1;
  double* c2 = ws + l.c2;
  const double* lkv = ws + l.lkv;
  const double* xgv = ws + l.xgv;
  const double* xg2v = ws + l.xg2v;
  double* part = ws + l.part;
  double* acc = ws + l.acc;
  double* Qm = ws + l.qv;               // [cd][cd]
  double* vv = Qm + cd * cd;            // [cd]
  const lds_f64p lds_wn = (lds_f64p)(ws + l.tw);
  const lds_u8p lds_code = (lds_u8p)(ws + l.tc);
  const unsigned char* g_code = c.code;
  const double* mu_z = mu;
  const double* s2_z = s2;

  unsigned long long t_a = 0, t_b = 0;
  if (c.stamps && tid == 0) t_a = __builtin_readcyclecounter();
  // ---- phase 1a
  {
    const int first = (NT > 64) ? 64 : 0;          // keep wave 0 for the link tables when there are other waves
    if (tid >= first) {
      const int g8 = (tid - first) >> 3, sub = tid & 7, ng8 = (NT - first) >> 3;
      for (int o = g8; o < cd * cd + cd; o += ng8) {
        double sacc = 0.0;
        if (o < cd * cd) {
          const int j = o / cd, j2 = o - j * cd;
          for (int d = sub; d < D; d += 8) sacc = fma(Wl[d * CD + j] * Wl[d * CD + j2], s2_z[d], sacc);
        } else {
          const int j = o - cd * cd;
          for (int d = sub; d < D; d += 8) sacc = fma(Wl[d * CD + j], mu_z[d], sacc);
        }
        sacc = group_sum(sacc, 8);
        if (sub == 0) Qm[o] = sacc;
      }
    }
    for (int o = tid; o < NS; o += NT) acc[o] = 0.0;
  }
  mom_phase1a(c, l, cd, nout, mu + D, s2 + D, ws);
  NAGP_STAMP(0);

  const double sn2a = sn2 / alpha;
  for (int base = 0; base < c.n_pts; base += CH) {
    const int npc = (c.n_pts - base < CH) ? (c.n_pts - base) : CH;
    // ---- phase 1b
    for (int pl = tid; pl < npc; pl += NT) {
      int cj[CD];
      if (TL) {
#pragma unroll
        for (int j = 0; j < CD; ++j) cj[j] = lds_code[(base + pl) * CD + j];
      } else {
#pragma unroll
        for (int j = 0; j < CD; ++j) cj[j] = g_code[(size_t)(base + pl) * CD + j];
      }
      double lk[CD];     // all gathers first, then the stores (LDS loads cannot be hoisted over LDS stores by the compiler)
#pragma unroll
      for (int j = 0; j < CD; ++j) lk[j] = lkv[j * nd + cj[j]];
#pragma unroll
      for (int j = 0; j < CD; ++j) lkp[(size_t)j * CH + pl] = lk[j];
      double sa2 = 0.0, sam = 0.0;
#pragma unroll
      for (int j = 0; j < CD; ++j) {
        double t = 0.0;
#pragma unroll
        for (int j2 = 0; j2 < CD; ++j2) t = fma(Qm[j * CD + j2], lk[j2], t);
        sa2 = fma(t, lk[j], sa2);
        sam = fma(vv[j], lk[j], sam);
      }
      const double sig2 = sn2a + sa2;
      double pdf, q, inv;
      gauss_terms(y, sam, sig2, pdf, q, inv);
      double wq;
      if (TL) wq = lds_wn[base + pl]; else wq = c.wn[base + pl];
      const double w0 = wq * pdf;
      c0[pl] = w0;
      c1[pl] = w0 * q;
      c2[pl] = w0 * (q * q - inv);
    }
    lds_barrier();
    NAGP_STAMP(1);
    // ---- phase 2
    if constexpr (CD <= 7) {
      // All weighted sums over the sigma points are one 16x16 block of  A' B  with per-point rows
      //   A_p = [c2 lk_0..lk_{N-1} | c1 | c0 xg2_0..xg2_{N-1} | c0]      B_p = [lk_0..lk_{N-1} | xg_0..xg_{N-1} | 1]
      // (R = rows/cols < N, u = row N, g2 = rows N+1.. x col 2N, g1 = row 2N+1 x cols N.., Z = row 2N+1 x col 2N):
      // v_mfma_f64_16x16x4 consumes four points per instruction; every wave takes every NW-th step.
      const int lane = tid & 63, NW = NT >> 6;
      const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the step loop below must be wave-uniform
      const int i = lane & 15, kq = lane >> 4;
      // branch-free operand assembly: A = (sA_lk*lk + sA_tb*xg2 + sA_1) * w,  B = sB_lk*lk + sB_tb*xg + sB_1
      const double* wA = (i < CD) ? c2 : ((i == CD) ? c1 : c0);
      const int jA = (i < CD) ? i : ((i > CD && i <= 2 * CD) ? i - CD - 1 : 0);
      const int jB = (i < CD) ? i : ((i < 2 * CD) ? i - CD : 0);
      const double sA_lk = (i < CD) ? 1.0 : 0.0, sA_tb = (i > CD && i <= 2 * CD) ? 1.0 : 0.0;
      const double sA_1 = (i == CD || i == 2 * CD + 1) ? 1.0 : 0.0;
      const double sB_lk = (i < CD) ? 1.0 : 0.0, sB_tb = (i >= CD && i < 2 * CD) ? 1.0 : 0.0, sB_1 = (i == 2 * CD) ? 1.0 : 0.0;
      const double* lA = lkp + (size_t)jA * CH;
      const double* lB = lkp + (size_t)jB * CH;
      const double* tA = xg2v + jA * nd;
      const double* tB = xgv + jB * nd;
      v4d accv = {0.0, 0.0, 0.0, 0.0};
      const int nstep = (npc + 3) >> 2;
      // batches of SB steps: the code bytes of a batch, then all dependent operands, then SB MFMAs (two LDS
      // round trips per batch instead of four per step)
      constexpr int SB = 4;
      for (int st0 = wave * SB; st0 < nstep; st0 += NW * SB) {
        int pl[SB], ca[SB], cb[SB];
        double okf[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
          const int q = 4 * (st0 + u) + kq;
          okf[u] = (q < npc) ? 1.0 : 0.0;
          pl[u] = (q < npc) ? q : npc - 1;
        }
        if (TL) {
#pragma unroll
          for (int u = 0; u < SB; ++u) { ca[u] = lds_code[(base + pl[u]) * CD + jA]; cb[u] = lds_code[(base + pl[u]) * CD + jB]; }
        } else {
#pragma unroll
          for (int u = 0; u < SB; ++u) { ca[u] = g_code[(size_t)(base + pl[u]) * CD + jA]; cb[u] = g_code[(size_t)(base + pl[u]) * CD + jB]; }
        }
        double w[SB], la[SB], lb[SB], ta[SB], tb_[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
          w[u] = wA[pl[u]]; la[u] = lA[pl[u]]; lb[u] = lB[pl[u]]; ta[u] = tA[ca[u]]; tb_[u] = tB[cb[u]];
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
          const double fa = fma(sA_lk, la[u], fma(sA_tb, ta[u], sA_1)) * (w[u] * okf[u]);
          const double fb = fma(sB_lk, lb[u], fma(sB_tb, tb_[u], sB_1));
          accv = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb, accv, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) part[wave * 256 + (kq + 4 * r) * 16 + i] = accv[r];
      lds_barrier();
      for (int o = tid; o < NS; o += NT) {
        int row, col;
        if (o < oR) { row = CD; col = o; }
        else if (o < oG1) { row = (o - oR) / CD; col = (o - oR) % CD; if (col < row) continue; }
        else if (o < oG2) { row = 2 * CD + 1; col = CD + (o - oG1); }
        else if (o < oZ) { row = CD + 1 + (o - oG2); col = 2 * CD; }
        else { row = 2 * CD + 1; col = 2 * CD; }
        double sacc = acc[o];
        for (int wv = 0; wv < NW; ++wv) sacc += part[wv * 256 + row * 16 + col];
        acc[o] = sacc;
      }
      lds_barrier();
    } else
    {
      const int grp = tid >> 4, gl = tid & 15, ngrp = NT >> 4;
      constexpr int ntask = 2 * CD;
      int rep = ngrp / ntask;
      rep = rep < 1 ? 1 : (rep > MOM_REP ? MOM_REP : rep);
      const int stride = 16 * rep, nit = (npc + stride - 1) / stride;
      for (int tk = grp; tk < ntask * rep; tk += ngrp) {
        const int task = tk % ntask, r = tk / ntask;
        double* pw = part + (size_t)r * NS;
        if (task < cd) {
          const int j = task;
          double au = 0.0, aR[CD];
#pragma unroll
          for (int jj = 0; jj < CD; ++jj) aR[jj] = 0.0;
          const double* lj = lkp + (size_t)j * CH;
#pragma unroll 4
          for (int it = 0; it < nit; ++it) {     // uniform trip count: out-of-range points get zero weight
            int pl = gl + 16 * r + it * stride;
            const bool ok = pl < npc;
            pl = ok ? pl : npc - 1;
            const double x = lj[pl];
            const double k1 = ok ? c1[pl] : 0.0, k2 = ok ? c2[pl] : 0.0;
            au = fma(k1, x, au);
            const double t = k2 * x;
#pragma unroll
            for (int jj = 0; jj < CD; ++jj)
              if (j + jj < cd) aR[jj] = fma(t, lj[(size_t)jj * CH + pl], aR[jj]);
          }
          au = group_sum(au, 16);
#pragma unroll
          for (int jj = 0; jj < CD; ++jj) aR[jj] = group_sum(aR[jj], 16);
          if (gl == 0) {
            pw[j] = au;
#pragma unroll
            for (int jj = 0; jj < CD; ++jj)
              if (j + jj < cd) pw[oR + j * CD + j + jj] = aR[jj];
          }
        } else {
          const int j = task - cd;
          double a1 = 0.0, a2 = 0.0, az = 0.0;
#pragma unroll 4
          for (int it = 0; it < nit; ++it) {
            int pl = gl + 16 * r + it * stride;
            const bool ok = pl < npc;
            pl = ok ? pl : npc - 1;
            int cg;
            if (TL) cg = lds_code[(base + pl) * CD + j]; else cg = g_code[(size_t)(base + pl) * CD + j];
            const double w0 = ok ? c0[pl] : 0.0;
            a1 = fma(xgv[j * nd + cg], w0, a1);
            a2 = fma(xg2v[j * nd + cg], w0, a2);
            az += w0;
          }
          a1 = group_sum(a1, 16);
          a2 = group_sum(a2, 16);
          az = group_sum(az, 16);
          if (gl == 0) { pw[oG1 + j] = a1; pw[oG2 + j] = a2; if (j == 0) pw[oZ] = az; }
        }
      }
      lds_barrier();
      for (int o = tid; o < NS; o += NT) {
        const int jr = (o >= oR && o < oG1) ? (o - oR) : -1;
        if (jr >= 0 && (jr % CD) < (jr / CD)) continue;     // lower triangle of R is not formed
        double sacc = acc[o];
        for (int r = 0; r < rep; ++r) sacc += part[(size_t)r * NS + o];
        acc[o] = sacc;
      }
      lds_barrier();
    }
    NAGP_STAMP(2);
  }
  // ---- phase 3
  {
    const double Zs = acc[oZ];
    const double Z = pEP * ((Zs > c.jitter) ? Zs : c.jitter);  // max(NaN,jitter)=jitter
    const double Zinv = 1.0 / Z;
    if (tid < D + cd) {
      double s1 = 0.0, s2_ = 0.0;
      if (tid < D) {
        double wd[CD];
#pragma unroll
        for (int j = 0; j < CD; ++j) wd[j] = Wl[tid * CD + j];
#pragma unroll
        for (int j = 0; j < CD; ++j) {
          s1 = fma(wd[j], acc[j], s1);
          double t = 0.5 * acc[oR + j * CD + j] * wd[j];
#pragma unroll
          for (int j2 = j + 1; j2 < CD; ++j2) t = fma(acc[oR + j * CD + j2], wd[j2], t);
          s2_ = fma(2.0 * wd[j], t, s2_);
        }
      } else {
        s1 = acc[oG1 + tid - D];
        s2_ = acc[oG2 + tid - D];
      }
      const double d1 = Zinv * pEP * s1;
      dl[tid] = d1;
      d2l[tid] = -d1 * d1 + Zinv * pEP * s2_;
    }
    if (tid == 0) *lZ = LOGZ ? log(Z) : Z;
    lds_barrier();
  }
  NAGP_STAMP(3);
}

// Block-structured NMF likelihoods (see MomSrc).  With a_d a function of the tuple b = tup_j(p) of its source only,
//   sum_d a_d^2 s2_d = sum_j S2_j[tup_j(p)],   sum_d a_d mu_d = sum_j SM_j[tup_j(p)],
//   sum_p c1_p a_d = sum_b a_d[b] C1_j[b],     sum_p c2_p a_d^2 = sum_b a_d[b]^2 C2_j[b],   C*_j[b] = sum_{p in b} c*_p
// and the modulator sums run over the tuples of their source as well.
//   1t  16-lane group per (source, tuple): a for the source's sub-bands (lanes), S2, SM
//   2   16-lane group per work item (a slice of one bin's sigma points, host-balanced): Gaussian weight of every
//       point from three table look-ups, c0 c1 c2 summed over the slice
//   2b  thread per bin: fixed-order sum of its items
//   3   thread per output: sums over the tuples of its source
// a[j] for a per-lane j without a vector load from the kernel-argument segment
__device__ __forceinline__ int pick_src(const int* a, int j) {
  const int a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
  return j == 0 ? a0 : (j == 1 ? a1 : (j == 2 ? a2 : a3));
}
__device__ __forceinline__ int pick_src1(const int* a, int j) {   // a[j + 1], a has MOM_MAXSRC + 1 entries
  const int a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4];
  return j == 0 ? a1 : (j == 1 ? a2 : (j == 2 ? a3 : a4));
}
template <int CD, bool LOGZ>
__device__ __forceinline__ void mom_src(const MomCfg& c, const double* Wl, double sn2, double alpha, double y,
                                        const double* mu, const double* s2, double* ws, double* lZ, double* dl, double* d2l,
                                        unsigned long long* acc_st, double pEP) {
  const int tid = threadIdx.x, NT = blockDim.x;
  const int D = c.D, nd = c.nd, npt = c.n_pts;
  constexpr int cd = CD;
  const int nout = D + cd + 1;
  const MomSrc& sc = c.src;
  const int nsrc = sc.n_src, nbmax = sc.nbmax, dlmax = sc.dlmax;
  const bool sq = __builtin_amdgcn_readfirstlane(c.lik_kind == 2 ? 1 : 0) != 0;
  const MomLay l = mom_layout(c);
  const double* lkv = ws + l.lkv;
  const double* xgv = ws + l.xgv;
  const double* xg2v = ws + l.xg2v;
  double* sums1 = ws + l.sums1;
  double* sums2 = ws + l.sums2;
  const lds_f64p lds_wn = (lds_f64p)(ws + l.tw);
  const lds_u8p blob = (lds_u8p)(ws + l.tc);
  const lds_u8p lds_tup = blob + sc.off[0];
  const lds_u8p lds_tcode = blob + sc.off[1];
  const lds_u16p lds_perm = (lds_u16p)(blob + sc.off[2]);
  const lds_i32p lds_items = (lds_i32p)(blob + sc.off[3]);
  const lds_i32p lds_bfirst = (lds_i32p)(blob + sc.off[4]);
  const lds_i32p lds_bitem = (lds_i32p)(blob + sc.off[5]);
  double* atab = ws + l.atab;
  double* s2b = ws + l.s2b;
  double* smb = ws + l.smb;
  double* cb = ws + l.cb;
  double* ipart = ws + l.ipart;
  const double* mu_z = mu;
  const double* s2_z = s2;

  unsigned long long t_a = 0, t_b = 0;
  if (c.stamps && tid == 0) t_a = __builtin_readcyclecounter();
  mom_phase1a(c, l, cd, nout, mu + D, s2 + D, ws);
  NAGP_STAMP(0);

  const int grp = tid >> 4, gl = tid & 15, ngrp = NT >> 4;
  // ---- 1t: one lane per (source, tuple); the sub-bands of the source in an unrolled loop (independent chains)
  for (int tk = tid; tk < nsrc * nbmax; tk += NT) {
    const int j = tk / nbmax, bq = tk - j * nbmax;
    if (bq >= pick_src(sc.nb, j)) continue;
    const int nj0 = pick_src(sc.n0, j), nc = pick_src1(sc.n0, j) - nj0, dj0 = pick_src(sc.d0, j), Dj = pick_src1(sc.d0, j) - dj0;
    const lds_u8p tcp = lds_tcode + tk * cd;
    double lk[CD];
#pragma unroll
    for (int k = 0; k < CD; ++k) {
      const int kk = (k < nc) ? k : 0;
      const double v = lkv[(nj0 + kk) * nd + tcp[kk]];
      lk[k] = (k < nc) ? v : 0.0;                      // components of other sources: zero weight (their W entries are 0 too)
    }
    double s2acc = 0.0, smacc = 0.0;
    double* ar = atab + (size_t)tk * dlmax;
#pragma unroll 4
    for (int dl_ = 0; dl_ < Dj; ++dl_) {
      const int d = dj0 + dl_;
      const double* wr = Wl + d * CD + nj0;
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < CD; ++k) { const int kk = (k < nc) ? k : 0; a = fma(wr[kk], lk[k], a); }
      if (sq) a = sqrt(a);
      ar[dl_] = a;
      s2acc = fma(a * a, s2_z[d], s2acc);
      smacc = fma(a, mu_z[d], smacc);
    }
    s2b[tk] = s2acc; smb[tk] = smacc;
  }
  lds_barrier();
  NAGP_STAMP(1);
  // ---- 2
  const double sn2a = sn2 / alpha;
  for (int e = grp; e < sc.n_items; e += ngrp) {
    const int j = lds_items[4 * e], start = lds_items[4 * e + 2], len = lds_items[4 * e + 3];
    const lds_u16p pm = lds_perm + j * npt + start;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int i0 = gl; i0 < len; i0 += 32) {            // two points per lane and trip: independent dependency chains
      double sig2[2], sam[2], okf[2];
      int pp[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = i0 + 16 * u;
        okf[u] = (i < len) ? 1.0 : 0.0;
        pp[u] = pm[(i < len) ? i : 0];
        sig2[u] = sn2a; sam[u] = 0.0;
      }
#pragma unroll
      for (int jj = 0; jj < MOM_MAXSRC; ++jj)
        if (jj < nsrc) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int t = lds_tup[jj * npt + pp[u]];
            sig2[u] += s2b[jj * nbmax + t];
            sam[u] += smb[jj * nbmax + t];
          }
        }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        double pdf, q, inv;
        gauss_terms(y, sam[u], sig2[u], pdf, q, inv);
        const double w0 = lds_wn[pp[u]] * pdf * okf[u];
        a0 += w0;
        a1 = fma(w0, q, a1);
        a2 = fma(w0, q * q - inv, a2);
      }
    }
    a0 = group_sum(a0, 16); a1 = group_sum(a1, 16); a2 = group_sum(a2, 16);
    if (gl == 0) { ipart[3 * e] = a0; ipart[3 * e + 1] = a1; ipart[3 * e + 2] = a2; }
  }
  lds_barrier();
  // ---- 2b
  for (int t = tid; t < nsrc * nbmax; t += NT) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int q = lds_bfirst[t]; q < lds_bfirst[t + 1]; ++q) {
      const int e = lds_bitem[q];
      a0 += ipart[3 * e]; a1 += ipart[3 * e + 1]; a2 += ipart[3 * e + 2];
    }
    cb[3 * t] = a0; cb[3 * t + 1] = a1; cb[3 * t + 2] = a2;
  }
  lds_barrier();
  NAGP_STAMP(2);
  // ---- 3: one 16-lane group per output, lanes stride over the tuples of the output's source
  for (int o = grp; o <= D + cd; o += ngrp) {
    double s1 = 0.0, s2_ = 0.0;
    if (o < D) {
      int j = 0;
      while (j + 1 < nsrc && o >= pick_src1(sc.d0, j)) ++j;
      const int dl_ = o - pick_src(sc.d0, j), nbj = pick_src(sc.nb, j);
      const double* at = atab + (size_t)j * nbmax * dlmax + dl_;
      const double* cp = cb + (size_t)3 * j * nbmax;
      for (int b = gl; b < nbj; b += 16) {
        const double a = at[(size_t)b * dlmax];
        s1 = fma(a, cp[3 * b + 1], s1);
        s2_ = fma(a * a, cp[3 * b + 2], s2_);
      }
    } else if (o < D + cd) {
      const int n = o - D;
      int j = 0;
      while (j + 1 < nsrc && n >= pick_src1(sc.n0, j)) ++j;
      const int k = n - pick_src(sc.n0, j), nbj = pick_src(sc.nb, j);
      const lds_u8p tcp = lds_tcode + j * nbmax * cd + k;
      const double* cp = cb + (size_t)3 * j * nbmax;
      for (int b = gl; b < nbj; b += 16) {
        const int ix = n * nd + tcp[b * cd];
        s1 = fma(xgv[ix], cp[3 * b], s1);
        s2_ = fma(xg2v[ix], cp[3 * b], s2_);
      }
    } else {
      for (int b = gl; b < sc.nb[0]; b += 16) s1 += cb[3 * b];   // Z: every sigma point lies in exactly one bin of source 0
    }
    s1 = group_sum(s1, 16); s2_ = group_sum(s2_, 16);
    if (gl == 0) { sums1[o] = s1; sums2[o] = s2_; }
  }
  lds_barrier();
  mom_phase3<LOGZ>(c, l, D + cd, pEP, ws, lZ, dl, d2l);
  NAGP_STAMP(3);
}

// POWER likelihood (likModulatorPower.m:25-100): cubature dimension = D, a_d = link(xn_d).
template <bool LOGZ>
__device__ __forceinline__ void mom_power(const MomCfg& c, double sn2, double alpha, double y, const double* mu,
                                          const double* s2, double* ws, double* lZ, double* dl, double* d2l,
                                          unsigned long long* acc_st, double pEP) {
  const int tid = threadIdx.x, NT = blockDim.x;
  const int D = c.D, cd = c.cdim, CH = mom_chunk(c), nd = c.nd;
  const int nout = D + cd + 1;
  const bool TL = __builtin_amdgcn_readfirstlane(c.cache_tabs ? 1 : 0) != 0;
  const MomLay l = mom_layout(c);
  double* rows = ws + l.rows;           // [D][CH] link(xn)[d][p]: point index fastest
  double* c0 = ws + l.c0;
  double* c1 = ws + l.c1;
  double* c2 = ws + l.c2;
  const double* lkv = ws + l.lkv;
  const double* xgv = ws + l.xgv;
  const double* xg2v = ws + l.xg2v;
  double* sums1 = ws + l.sums1;
  double* sums2 = ws + l.sums2;
  const double* lds_wn = ws + l.tw;
  const unsigned char* codes = TL ? reinterpret_cast<const unsigned char*>(ws + l.tc) : c.code;
  const double* mu_z = mu;
  const double* s2_z = s2;

  unsigned long long t_a = 0, t_b = 0;
  if (c.stamps && tid == 0) t_a = __builtin_readcyclecounter();
  mom_phase1a(c, l, cd, nout, mu + D, s2 + D, ws);
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
      double sa2 = 0.0, sam = 0.0;
      const unsigned char* cp = codes + (size_t)p * cd;
      for (int d = sub; d < D; d += DG) {
        const double a = lkv[d * nd + cp[d]];
        row[(size_t)d * CH] = a;
        sa2 = fma(a * a, s2_z[d], sa2);
        sam = fma(a, mu_z[d], sam);
      }
      sa2 = group_sum(sa2, DG);
      sam = group_sum(sam, DG);
      if (sub == 0) {
        const double sig2 = sn2a + sa2;
        double pdf, q, inv;
        gauss_terms(y, sam, sig2, pdf, q, inv);
        const double w0 = (TL ? lds_wn[p] : c.wn[p]) * pdf;
        c0[pl] = w0;
        c1[pl] = w0 * q;
        c2[pl] = w0 * (q * q - inv);
      }
    }
    lds_barrier();
    NAGP_STAMP(1);
    // ---- phase 2: one output per 16-lane group; the group's lanes stride over the points of the chunk;
    // DPP sum inside the group (fixed order).  The modulator outputs (heavier gathers) are handed out first.
    {
      const int grp = tid >> 4, gl = tid & 15, ngrp = NT >> 4;
      for (int oo = grp; oo < nout; oo += ngrp) {
        const int o = (oo < cd + 1) ? (D + oo) : (oo - cd - 1);   // modulators, Z, then the sub-bands
        double a1 = 0.0, a2 = 0.0, b1 = 0.0, b2 = 0.0;
        if (o < D) {
          const double* ro = rows + (size_t)o * CH;
          int pl = gl;
          for (; pl + 16 < npc; pl += 32) {
            const double x0 = ro[pl], x1 = ro[pl + 16];
            a1 = fma(x0, c1[pl], a1); b1 = fma(x1, c1[pl + 16], b1);
            a2 = fma(x0 * x0, c2[pl], a2); b2 = fma(x1 * x1, c2[pl + 16], b2);
          }
          if (pl < npc) { const double x0 = ro[pl]; a1 = fma(x0, c1[pl], a1); a2 = fma(x0 * x0, c2[pl], a2); }
        } else if (o < D + cd) {
          const int j = o - D;
          const unsigned char* cb = codes + (size_t)base * cd + j;
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
  mom_phase3<LOGZ>(c, l, D + cd, pEP, ws, lZ, dl, d2l);
  NAGP_STAMP(3);
}
#undef NAGP_STAMP

// Wl: NMF weights in LDS (D x N row-major)
// power-EP normaliser of likModulatorPreCalcwn.m:48 (1 for the other likelihoods); constant per kernel
__device__ inline double mom_pEP(const MomCfg& c, double sn2, double alpha) {
  return (c.lik_kind == 2) ? pow(2.0 * 3.14159265358979323846 * sn2, 0.5 * (1.0 - alpha)) / sqrt(alpha) : 1.0;
}
// MV: the cubature dimension the calling kernel was instantiated for (0 = POWER, 1..9 = N of the NMF likelihoods)
// LOGZ = false: *lZ receives Z itself (the sequential filters take the logarithm off the critical path)
// SRC = false: the block-structured path is compiled out (kernels whose register budget belongs to covariance tiles)
template <int MV, bool LOGZ = true, bool SRC = true>
__device__ __forceinline__ void mom_eval(const MomCfg& c, const double* Wl, double pEP, double sn2, double alpha,
                                         double y, const double* mu, const double* s2, double* ws, double* lZ, double* dl,
                                         double* d2l, unsigned long long* acc_st = nullptr) {
  unsigned long long dummy_st[8];
  if (!acc_st) acc_st = dummy_st;
  if constexpr (MV == 0) {
    mom_power<LOGZ>(c, sn2, alpha, y, mu, s2, ws, lZ, dl, d2l, acc_st, pEP);
  } else {
#if defined(NAGP_EXPERIMENT_ONLY_QUAD)
    mom_quad<MV, LOGZ>(c, Wl, sn2, alpha, y, mu, s2, ws, lZ, dl, d2l, acc_st, pEP);
#elif defined(NAGP_EXPERIMENT_ONLY_SQRT)
    mom_nmf<MV, LOGZ>(c, Wl, sn2, alpha, y, mu, s2, ws, lZ, dl, d2l, acc_st, pEP);
#else
    if (SRC && __builtin_amdgcn_readfirstlane(c.src.n_src) >= 2) mom_src<MV, LOGZ>(c, Wl, sn2, alpha, y, mu, s2, ws, lZ, dl, d2l, acc_st, pEP);
    else if (__builtin_amdgcn_readfirstlane(c.lik_kind) == 1) mom_quad<MV, LOGZ>(c, Wl, sn2, alpha, y, mu, s2, ws, lZ, dl, d2l, acc_st, pEP);
    else mom_nmf<MV, LOGZ>(c, Wl, sn2, alpha, y, mu, s2, ws, lZ, dl, d2l, acc_st, pEP);
#endif
  }
}
constexpr int MOM_MAXCD = 9;      // IHGP filter, site refresh, mom on its own
constexpr int MOM_MAXCD_GF = 9;   // kernels with register-resident covariance tiles
__host__ __device__ inline int mom_variant(const MomCfg& c) { return c.lik_kind == 0 ? 0 : c.cdim; }

}  // namespace nagp
