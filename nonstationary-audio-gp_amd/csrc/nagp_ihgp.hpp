// nagp_ihgp.hpp -- HIP kernels of the infinite-horizon (steady-state) path.
//   ihgp_filter_kernel  forward filter with per-channel DARE table look-ups, mean-only recursion
//                       (matlab/ihgp_ep_modulator_nmf.m:233-310); one workgroup per problem; block n's
//                       state lives in the registers of thread n, the workgroup cooperates on `mom`.
//   ihgp_scan_kernel    backward mean recursion with looked-up steady-state smoother gains (:373-394)
// The EP refresh (:397-436) reuses ep_site_kernel (parallel over steps).
#pragma once
#include "nagp_kernels.hpp"
#include "nagp_momsq.hpp"

namespace nagp {

// Device look-up tables of ONE problem (all doubles), derived on the host from PPlist / PGlist:
//   hph[M][NG]      h_n^2 * PP(1,1)             (diag(H*PP*H'))
//   wcol[M][NG][BS] h_n * PP(:,1)               (W(ii,n) = PP(ii,ii) * H(n,ii)')          BS = Shape::BS: 4, or 8 when a block has 5 .. 8 states
//   hph0[M], wcol0[M][BS] the same from Pinf (k = 1 of the reference uses PP = Pinf)
//   gtab[M][NG][BS*BS] smoother gain block G (row-major, zero padded)
//   vtab[M][NG]     h_n^2 * PS2(1,1)            (diag(H*P*H') of the looked-up smoother covariance)
struct IhgpTabs {
  int NG;
  const double* r;     // [NG] shared grid
  double lr0;          // log10(r[0])
  double inv_dlr;      // (NG-1)/(log10(r[NG-1])-log10(r[0]))
  const double* base;  // [B][ihgp_tab_size]
};
__host__ __device__ inline size_t itab_hph(const Shape&, int) { return 0; }
__host__ __device__ inline size_t itab_wcol(const Shape& s, int NG) { return (size_t)s.M * NG; }
__host__ __device__ inline size_t itab_hph0(const Shape& s, int NG) { return (size_t)s.M * NG * (1 + s.BS); }
__host__ __device__ inline size_t itab_wcol0(const Shape& s, int NG) { return (size_t)s.M * NG * (1 + s.BS) + s.M; }
__host__ __device__ inline size_t itab_g(const Shape& s, int NG) { return (size_t)s.M * NG * (1 + s.BS) + (1 + s.BS) * (size_t)s.M; }
__host__ __device__ inline size_t itab_v(const Shape& s, int NG) { return itab_g(s, NG) + (size_t)s.M * NG * s.BS * s.BS; }
__host__ __device__ inline size_t itab_size(const Shape& s, int NG) { return itab_v(s, NG) + (size_t)s.M * NG; }

// [~,ind] = min(abs(r-R)): first minimiser; NaN / +-Inf distances everywhere -> index 0 (SURVEY C-4)
__device__ __forceinline__ int nearest_idx(const IhgpTabs& tb, double R) {
  if (!(R == R) || isinf(R)) return 0;
  int est = 0;
  if (R > 0.0) {
    const double f = (log10(R) - tb.lr0) * tb.inv_dlr;
    est = (f <= 0.0) ? 0 : ((f >= (double)(tb.NG - 1)) ? tb.NG - 1 : (int)(f + 0.5));
  }
  if (R > tb.r[tb.NG - 1]) {
    // beyond the grid |r_i - R| is non-increasing in i and may ROUND to the same value for several -- for
    // R > ~1e20 all -- grid points: MATLAB's min returns the first of them (e.g. index 0 for R = 1/ttau = 1e100)
    const double dmin = fabs(tb.r[tb.NG - 1] - R);
    if (fabs(tb.r[0] - R) == dmin) return 0;
    int i = tb.NG - 1;
    while (i > 0 && fabs(tb.r[i - 1] - R) == dmin) --i;
    return i;
  }
  const int lo = (est - 2 < 0) ? 0 : est - 2;
  const int hi = (est + 2 > tb.NG - 1) ? tb.NG - 1 : est + 2;
  int best = lo;
  double bd = fabs(tb.r[lo] - R);
  for (int i = lo + 1; i <= hi; ++i) {
    const double d = fabs(tb.r[i] - R);
    if (d < bd) { bd = d; best = i; }
  }
  return best;
}

struct IhgpPar {
  int itt;
  double ep_damp;
  int mom_all;       // sweep 1: mom at every step; later only at k == T-1
  int64_t k_start;   // first step to process (sweeps >= 2 run only k = T-1 here; the rest is ihgp_aff_*)
  double R_init;     // exp(lik) (or 0 for the constraints variant): initial content of R(:,k)
  int hph_lds;       // filter: keep the H PP H' look-up table [M][NG] in LDS
  int kb;            // steps per I/O block of the filter (LDS ring), <= IH_KB
  int dbg_wave;      // developer diagnostics (NAGP_STAMPS): which worker's time line goes to stamps[8..15] (NAGP_STAMP_WORKER, default 0)
  double w_old, w_new, mom_alpha;   // as FilterPar: (1-d, d, 1) ihgp_ep_modulator_nmf.m:210-211 ; (1-d, d/alpha, alpha) experiments/ihgp_ep_mods_nmf_mixture.m:291-297
};

constexpr int IH_KB = 16;   // steps per I/O block of the filter (LDS ring); fewer when the LDS is needed elsewhere

__host__ __device__ inline size_t ihgp_ring_doubles(const Shape& s, int kb) { return (size_t)kb * (4 * s.M + s.S + 3); }
__host__ __device__ inline size_t ihgp_filter_lds_doubles(const Shape& s, const MomCfg& mc, int NG, int hph_lds, int kb = IH_KB) {
  return LDS_INT_DOUBLES + (size_t)s.D * s.N + 6 * (size_t)s.M + 8 + NG + (hph_lds ? (size_t)s.M * NG : 0) +
         ihgp_ring_doubles(s, kb) + mom_lds_doubles(mc);
}

// log10 to ~0.003 absolute (exponent + quadratic in the mantissa): only seeds the +-2 window below
__device__ __forceinline__ double coarse_log10(double R) {
  const int hi = __double2hiint(R);
  const int e = ((hi >> 20) & 0x7ff) - 1023;
  const double t = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, __double2loint(R)) - 1.0;
  return ((double)e + t + 0.3466 * t * (1.0 - t)) * 0.30102999566398120;
}

// first minimiser of |r_i - R| with the grid in LDS (same semantics as nearest_idx)
__device__ __forceinline__ int nearest_idx_lds(const double* r, int NG, double lr0, double inv_dlr, double R) {
  if (!(R == R) || isinf(R)) return 0;
  int est = 0;
  if (R > 0.0) {
    const double f = (coarse_log10(R) - lr0) * inv_dlr;
    est = (f <= 0.0) ? 0 : ((f >= (double)(NG - 1)) ? NG - 1 : (int)(f + 0.5));
  }
  if (R > r[NG - 1]) {   // rounding ties beyond the grid: first minimiser (see nearest_idx)
    const double dmin = fabs(r[NG - 1] - R);
    if (fabs(r[0] - R) == dmin) return 0;
    int i = NG - 1;
    while (i > 0 && fabs(r[i - 1] - R) == dmin) --i;
    return i;
  }
  const int lo = (est - 2 < 0) ? 0 : est - 2;
  const int hi = (est + 2 > NG - 1) ? NG - 1 : est + 2;
  int best = lo;
  double bd = fabs(r[lo] - R);
  for (int i = lo + 1; i <= hi; ++i) {
    const double d = fabs(r[i] - R);
    if (d < bd) { bd = d; best = i; }
  }
  return best;
}

// All per-step global traffic goes through an LDS ring of IH_KB steps that is filled / flushed with
// coalesced transfers once per block, so the sequential loop body contains no global-memory waits
// except the (L2-resident) table gather.
// SRC: the block-structured mom path (source-separation mixtures) and a run-time ring depth; compiled out otherwise
// BS: doubles per block row in the packed model and the tables (4; 8 for plans with a block of 5 .. 8 states)
template <int MV, bool SRC, int BS = 4>
__global__ void __launch_bounds__(MV >= 9 ? 512 : 256) ihgp_filter_kernel(Shape sh, Bufs b, MomCfg mc, IhgpTabs tb, IhgpPar ip) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x, NT = blockDim.x;
  const int S = sh.S, M = sh.M, NG = tb.NG;
  const int64_t T = sh.T;
  const int pb = blockIdx.x;
  const double* mdl = b.model + (size_t)pb * mdl_size(sh);
  const double* tab = tb.base + (size_t)pb * itab_size(sh, NG);

  int* ioff = reinterpret_cast<int*>(lds);
  int* ibsz = ioff + (MAXM + 1);
  double* sW = lds + LDS_INT_DOUBLES;
  double* fmu = sW + (size_t)sh.D * sh.N;
  double* HPH = fmu + M;
  double* dl = HPH + M;
  double* d2l = dl + M;
  double* misc = d2l + M;
  double* rg = misc + 8 + 2 * M;          // [NG] look-up grid
  double* thph = rg + NG;                  // [M][NG] H PP H' table (ip.hph_lds)
  const int KB = SRC ? ip.kb : IH_KB;
  double* ry = thph + (ip.hph_lds ? (size_t)M * NG : 0);   // ring: y[KB]
  double* rlZ = ry + KB;                //       lZ[KB]
  double* rZ = rlZ + KB;                //       Z of the steps that called mom (< 0: none); log taken at the flush
  double* rtt = rZ + KB;                //       ttau[KB][M]
  double* rtn = rtt + (size_t)KB * M;   //       tnu
  double* rR = rtn + (size_t)KB * M;    //       R
  double* rfm = rR + (size_t)KB * M;    //       H*m (filtered)
  double* rMF = rfm + (size_t)KB * M;   //       m (filtered) [KB][S]
  double* ws = rMF + (size_t)KB * S;
  for (int i = tid; i <= M; i += NT) ioff[i] = sh.off[i];
  for (int i = tid; i < M; i += NT) ibsz[i] = sh.bsz[i];
  for (int i = tid; i < sh.D * sh.N; i += NT) sW[i] = mdl[mdl_W(sh) + i];
  for (int i = tid; i < NG; i += NT) rg[i] = tb.r[i];
  if (ip.hph_lds)
    for (int i = tid; i < M * NG; i += NT) thph[i] = tab[itab_hph(sh, NG) + i];
  const double sn2 = mdl[mdl_sn2(sh)];
  mom_cache_tables(mc, ws);
  const double pEP1 = mom_pEP(mc, sn2, ip.mom_alpha);
  __syncthreads();

  // thread n < M owns block n
  const int n = tid;
  const bool act = n < M;
  double A4[BS * BS], mreg[BS];
#pragma unroll
  for (int i = 0; i < BS; ++i) mreg[i] = 0.0;
  double hn = 0.0;
  int o = 0, bs = 0;
  if (act) {
    if constexpr (BS == 4) tile_load(A4, mdl + mdl_A(sh) + (size_t)n * 16);
    else {
#pragma unroll
      for (int e = 0; e < BS * BS; ++e) A4[e] = mdl[mdl_A(sh) + (size_t)n * BS * BS + e];
    }
    hn = mdl[mdl_h(sh) + n];
    o = ioff[n]; bs = ibsz[n];
    if (ip.k_start > 0) {      // continue from the filtered mean of the previous step
      const double* mp = b.MF + ((size_t)pb * T + (ip.k_start - 1)) * S;
#pragma unroll
      for (int i = 0; i < BS; ++i)
        if (i < bs) mreg[i] = mp[o + i];
    } else if (ip.itt > 1) {   // m is NOT reset between sweeps (SURVEY C-22): smoothed mean at k=0
      const double* ms0 = b.MS + (size_t)pb * T * S;
#pragma unroll
      for (int i = 0; i < BS; ++i)
        if (i < bs) mreg[i] = ms0[o + i];
    }
  }
  const double* yv = b.y + (size_t)pb * T;
  double* g_tt = b.ttau + (size_t)pb * T * M;
  double* g_tn = b.tnu + (size_t)pb * T * M;
  double* g_R = b.R + (size_t)pb * T * M;
  double* g_lZ = b.lZ + (size_t)pb * T;
  double* g_MF = b.MF + (size_t)pb * T * S;
  double* g_fm = b.fm + (size_t)pb * T * M;
  double Rprev = (act && ip.k_start > 0) ? b.R[((size_t)pb * T + (ip.k_start - 1)) * M + n] : 0.0;
  unsigned long long n_clamped = 0;
  unsigned long long st_a = 0, st_b = 0, st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (mc.stamps && tid == 0) st_a = __builtin_readcyclecounter();

  for (int64_t k0 = ip.k_start; k0 < T; k0 += KB) {
    const int nb = (T - k0 < KB) ? (int)(T - k0) : KB;
    // ---- fill the ring for steps k0 .. k0+nb-1
    for (int i = tid; i < nb; i += NT) { ry[i] = yv[k0 + i]; rlZ[i] = g_lZ[k0 + i]; rZ[i] = -1.0; }
    for (int i = tid; i < nb * M; i += NT) {
      rtt[i] = g_tt[(size_t)k0 * M + i]; rtn[i] = g_tn[(size_t)k0 * M + i]; rR[i] = g_R[(size_t)k0 * M + i];
    }
    __syncthreads();
    for (int kk = 0; kk < nb; ++kk) {
      const int64_t k = k0 + kk;
      const double yk = ry[kk];
      double hph = 0.0, wc[BS], Am[BS], fmun = 0.0;
#pragma unroll
      for (int i = 0; i < BS; ++i) { wc[i] = 0.0; Am[i] = 0.0; }
      if (act) {
        if (k > 0) {
          const int idx = nearest_idx_lds(rg, NG, tb.lr0, tb.inv_dlr, Rprev);
          hph = ip.hph_lds ? thph[n * NG + idx] : tab[itab_hph(sh, NG) + (size_t)n * NG + idx];
          const double* w = tab + itab_wcol(sh, NG) + ((size_t)n * NG + idx) * BS;
#pragma unroll
          for (int i = 0; i < BS; ++i) wc[i] = w[i];
        } else {
          hph = tab[itab_hph0(sh, NG) + n];
          const double* w = tab + itab_wcol0(sh, NG) + (size_t)n * BS;
#pragma unroll
          for (int i = 0; i < BS; ++i) wc[i] = w[i];
        }
#pragma unroll
        for (int i = 0; i < BS; ++i) {
          double a = 0.0;
#pragma unroll
          for (int l = 0; l < BS; ++l) a = fma(A4[BS * i + l], mreg[l], a);
          Am[i] = a;
        }
        fmun = hn * Am[0];
        fmu[n] = fmun; HPH[n] = hph;
      }
      const bool do_mom = ip.mom_all || (k == T - 1);
      double tnew = 0.0, nnew = 0.0, Rn = 0.0;
      if (do_mom) {
        lds_barrier();
        if (mc.stamps && tid == 0) { st_b = __builtin_readcyclecounter(); st[4] += st_b - st_a; }
        mom_eval<MV, false, SRC>(mc, sW, pEP1, sn2, ip.mom_alpha, yk, fmu, HPH, ws, &misc[0], dl, d2l, st);
        if (act) {
          const double d1 = dl[n], d2 = d2l[n];
          const double t_old = rtt[kk * M + n], n_old = rtn[kk * M + n];
          tnew = ip.w_old * t_old + ip.w_new * (-d2 / (1.0 + d2 * hph));
          nnew = ip.w_old * n_old + ip.w_new * ((d1 - fmun * d2) / (1.0 + d2 * hph));
          Rn = 1.0 / tnew;                      // before the clamp (:269)
        }
        if (tid == 0) rZ[kk] = misc[0];
        if (mc.stamps && tid == 0) st_a = __builtin_readcyclecounter();
      } else if (act) {
        tnew = rtt[kk * M + n]; nnew = rtn[kk * M + n];
        Rn = rR[kk * M + n];
      }
      if (act) {
        if (!(tnew > 0.0)) ++n_clamped;
        tnew = max0(tnew);                       // :274 (NaN -> 0, C-3)
        const double ys = nnew / tnew;
        if (tnew == 0.0) {
          Rn = INFINITY;
#pragma unroll
          for (int i = 0; i < BS; ++i) mreg[i] = Am[i];
        } else {
          const double den = hph + Rn;
#pragma unroll
          for (int i = 0; i < BS; ++i) mreg[i] = Am[i] + (wc[i] / den) * (ys - fmun);
        }
        rtt[kk * M + n] = tnew; rtn[kk * M + n] = nnew; rR[kk * M + n] = Rn;
#pragma unroll
        for (int i = 0; i < BS; ++i)
          if (i < bs) rMF[(size_t)kk * S + o + i] = mreg[i];
        rfm[kk * M + n] = hn * mreg[0];
        Rprev = Rn;
      }
      if (do_mom) lds_barrier();   // fmu/HPH/dl reuse
      if (mc.stamps && tid == 0) { st_b = __builtin_readcyclecounter(); st[5] += st_b - st_a; st_a = st_b; }
    }
    // ---- flush the ring
    __syncthreads();
    for (int i = tid; i < nb; i += NT) g_lZ[k0 + i] = (rZ[i] < 0.0) ? rlZ[i] : log(rZ[i]);
    for (int i = tid; i < nb * M; i += NT) {
      g_tt[(size_t)k0 * M + i] = rtt[i]; g_tn[(size_t)k0 * M + i] = rtn[i]; g_R[(size_t)k0 * M + i] = rR[i];
      g_fm[(size_t)k0 * M + i] = rfm[i];
    }
    for (int i = tid; i < nb * S; i += NT) g_MF[(size_t)k0 * S + i] = rMF[i];
    __syncthreads();
  }
  if (act && n_clamped) atomicAdd(&b.counters[(size_t)pb * 4 + 1], n_clamped);
  if (mc.stamps && tid == 0)
    for (int i = 0; i < 8; ++i) mc.stamps[i] += st[i];
}

// ---------------------------------------------------------------------------------------------
// ADF sweep of the infinite-horizon filter for likModulatorNMFPower on fully symmetric sigma-point sets
// (ihgp_ep_modulator_nmf.m:233-310 with the mom of likModulatorNMFPower.m:28-87): the same step as ihgp_filter_kernel,
// organised around the latency of ONE sequential chain.  Wave 0 owns the sub-band sites (lane d = block d), wave 1 the modulator
// sites (lane j = block D+j); each runs, without any workgroup barrier, everything from the reduced cubature sums of step k to the
// inputs of the cubature of step k+1:
//   sums -> d lZ, d2 lZ -> site update, clamp, R -> gain, mean update, ring -> table look-up for k+1 -> A m, fmu, H PP H'
// The modulator sites have the short tail (their sums come ready-made out of the cubature), so wave 1 goes straight on to the
// link tables of step k+1 -- the exp / log chain of the step -- while wave 0 still forms W_d'RW_d for the sub-bands.
// The four waves then share the remaining cubature stages of nagp_momsp.hpp (five barriers per step).  Every processed step calls
// mom (sweep 1: all steps; later sweeps: launched for k = T-1 only).
// Ring of the filtered means: [KB][M][4], block padded (two 16-byte stores per lane and step); the flush strips the padding.
__host__ __device__ inline size_t ihgp_adf_ring_doubles(const Shape& s, int kb) { return (size_t)kb * (8 * s.M + 3); }
__host__ __device__ inline size_t ihgp_adf_lds_doubles(const Shape& s, int CD, int NG, int hph_lds, int kb) {
  return (size_t)s.D * s.N + 2 * 68 + NG + (hph_lds ? (size_t)s.M * NG : 0) + ihgp_adf_ring_doubles(s, kb) + (s.S + 1) / 2 + 2 +
         msp_lds_doubles(CD, s.D);
}
__host__ __device__ inline size_t ihgp_adf8_lds_doubles(const Shape& s, int CD, int NG, int hph_lds, int kb) {
  return ihgp_adf_lds_doubles(s, CD, NG, hph_lds, kb) - msp_lds_doubles(CD, s.D) + msp_lds_doubles(CD, s.D, 1);
}

// [~,ind] = min(abs(r-R)) as nearest_idx_lds, with a three-point window around the seed: the seed's error (0.003 in log10,
// a tenth of a grid step, plus 0.02 steps between the linear and the logarithmic mid-point) keeps the minimiser inside it.
// Grids that are not the reference's logspace(-2,4,200) (ihgp_ep_modulator_nmf.m:131) take the five-point form.
__device__ __forceinline__ int nearest_idx3(msp_rp r, int NG, double lr0, double inv_dlr, double rmax, double R) {
  if (!(R == R) || isinf(R)) return 0;
  if (R > rmax) {   // rounding ties beyond the grid: first minimiser (see nearest_idx)
    const double dmin = fabs(r[NG - 1] - R);
    if (fabs(r[0] - R) == dmin) return 0;
    int i = NG - 1;
    while (i > 0 && fabs(r[i - 1] - R) == dmin) --i;
    return i;
  }
  int est = 0;
  if (R > 0.0) {
    const double f = (coarse_log10(R) - lr0) * inv_dlr;
    est = (f <= 0.0) ? 0 : ((f >= (double)(NG - 1)) ? NG - 1 : (int)(f + 0.5));
  }
  const int mid = (est < 1) ? 1 : ((est > NG - 2) ? NG - 2 : est);
  const double d0 = fabs(r[mid - 1] - R), d1 = fabs(r[mid] - R), d2 = fabs(r[mid + 1] - R);
  int best = mid - 1; double bd = d0;
  if (d1 < bd) { bd = d1; best = mid; }
  if (d2 < bd) best = mid + 1;
  return best;
}

template <int CD>
__global__ void __launch_bounds__(MSP_NT) ihgp_adf_kernel(Shape sh, Bufs b, MomCfg mc, MomSp sp, IhgpTabs tb, IhgpPar ip) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NT = MSP_NT;
  const int S = sh.S, M = sh.M, D = sh.D, NG = tb.NG;
  const int64_t T = sh.T;
  const int pb = blockIdx.x;
  const double* mdl = b.model + (size_t)pb * mdl_size(sh);
  const double* tab = tb.base + (size_t)pb * itab_size(sh, NG);

  const int KB = ip.kb;
  double* rMF = lds;                                 // ring: m (filtered) [KB][M][4]   (16-byte aligned)
  double* rtt = rMF + (size_t)KB * M * 4;            //       ttau[KB][M]
  double* rtn = rtt + (size_t)KB * M;                //       tnu
  double* rR = rtn + (size_t)KB * M;                 //       R
  double* rfm = rR + (size_t)KB * M;                 //       H*m (filtered)
  double* ry = rfm + (size_t)KB * M;                 //       y[KB]
  double* rlZ = ry + KB;                             //       lZ[KB]
  double* rZ = rlZ + KB;                             //       Z of the step; log taken at the flush
  double* sW = rZ + KB;                              // [D][CD]
  double* fmu = sW + (size_t)D * CD;                 // [68]: M sites, zero padded (stage A of the cubature reads 4*K entries)
  double* HPH = fmu + 68;
  double* rg = HPH + 68;                          // [NG] look-up grid
  double* thph = rg + NG;                            // [M][NG] H PP H' table (ip.hph_lds)
  int* smap = reinterpret_cast<int*>(thph + (ip.hph_lds ? (size_t)M * NG : 0));   // [S] state -> padded ring slot 4*block + row
  double* ws = reinterpret_cast<double*>(smap) + (S + 1) / 2;
  for (int i = tid; i < D * CD; i += NT) sW[i] = mdl[mdl_W(sh) + i];
  for (int i = tid; i < NG; i += NT) rg[i] = tb.r[i];
  if (ip.hph_lds)
    for (int i = tid; i < M * NG; i += NT) thph[i] = tab[itab_hph(sh, NG) + i];
  for (int i = tid; i < 68; i += NT) { fmu[i] = 0.0; HPH[i] = 0.0; }
  for (int i = tid; i < M; i += NT)
    for (int r = 0; r < sh.bsz[i]; ++r) smap[sh.off[i] + r] = 4 * i + r;
  const double sn2 = mdl[mdl_sn2(sh)];
  const double sn2a = sn2 / ip.mom_alpha;
  const double pEP1 = mom_pEP(mc, sn2, ip.mom_alpha);
  const double rmax = tb.r[NG - 1];
  __syncthreads();
  MspCtx<CD> x;
  msp_setup<CD>(x, mc, sp, sW, fmu, HPH, ws, 1);

  // wave 0, lane d < D owns sub-band block d; wave 1, lane j < N owns modulator block D + j
  const int lane = tid & 63;
  const bool sub = (wave == 0) && lane < D;
  const bool act = sub || ((wave == 1) && lane < sh.N);
  const int n = (wave == 0) ? lane : D + lane;
  const int nn = act ? n : 0;
  double A4[16], mreg[4] = {0, 0, 0, 0}, wrow[CD];
  double hn = 0.0;
#pragma unroll
  for (int j = 0; j < CD; ++j) wrow[j] = 0.0;
  tile_zero(A4);
  if (act) {
    tile_load(A4, mdl + mdl_A(sh) + (size_t)n * 16);
    hn = mdl[mdl_h(sh) + n];
    const int o = sh.off[n], bs = sh.bsz[n];
    if (sub) {
#pragma unroll
      for (int j = 0; j < CD; ++j) wrow[j] = sW[n * CD + j];
    }
    if (ip.k_start > 0) {      // continue from the filtered mean of the previous step
      const double* mp = b.MF + ((size_t)pb * T + (ip.k_start - 1)) * S;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < bs) mreg[i] = mp[o + i];
    } else if (ip.itt > 1) {   // m is NOT reset between sweeps (SURVEY C-22): smoothed mean at k=0
      const double* ms0 = b.MS + (size_t)pb * T * S;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < bs) mreg[i] = ms0[o + i];
    }
  }
  // per-lane ring / table addresses (vector registers; the step index adds an immediate-free scalar offset)
  const msp_wp p_tt = (msp_wp)(rtt + nn), p_tn = (msp_wp)(rtn + nn), p_R = (msp_wp)(rR + nn), p_fm = (msp_wp)(rfm + nn);
  const msp_wp p_MF = (msp_wp)(rMF + 4 * nn);
  const msp_wp p_fmu = (msp_wp)(fmu + nn), p_HPH = (msp_wp)(HPH + nn);
  const msp_rp p_hph = (msp_rp)(thph + (size_t)nn * NG);
  const msp_rp p_rg = (msp_rp)rg + opaque_zero();
  const double* g_wcol = tab + itab_wcol(sh, NG) + (size_t)nn * NG * 4;
  const double* g_hph = tab + itab_hph(sh, NG) + (size_t)nn * NG;
  const double* yv = b.y + (size_t)pb * T;
  double* g_tt = b.ttau + (size_t)pb * T * M;
  double* g_tn = b.tnu + (size_t)pb * T * M;
  double* g_R = b.R + (size_t)pb * T * M;
  double* g_lZ = b.lZ + (size_t)pb * T;
  double* g_MF = b.MF + (size_t)pb * T * S;
  double* g_fm = b.fm + (size_t)pb * T * M;
  double Rprev = (act && ip.k_start > 0) ? b.R[((size_t)pb * T + (ip.k_start - 1)) * M + n] : 0.0;
  unsigned int n_clamped = 0;
  unsigned long long st_a = 0, st_b = 0, st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool stamp = mc.stamps && tid == 0;      // wave 0's time line
#define IH_STAMP(slot) do { if (stamp) { st_b = __builtin_readcyclecounter(); st[slot] += st_b - st_a; st_a = st_b; } } while (0)

  // head of step k: table look-up, A m, the cubature's inputs (wave 0, no barrier)
  double hph = 0.0, wc[4] = {0, 0, 0, 0}, Am[4] = {0, 0, 0, 0}, fmun = 0.0;
  auto head = [&](int64_t k) {
    if (act) {
      if (k > 0) {
        const int idx = nearest_idx3(p_rg, NG, tb.lr0, tb.inv_dlr, rmax, Rprev);
        if (stamp) { asm volatile("" :: "v"(idx)); IH_STAMP(6); }
        hph = ip.hph_lds ? p_hph[idx] : g_hph[idx];
        const double2* w = reinterpret_cast<const double2*>(g_wcol + (size_t)idx * 4);
        const double2 w0 = w[0], w1 = w[1];
        wc[0] = w0.x; wc[1] = w0.y; wc[2] = w1.x; wc[3] = w1.y;
      } else {
        hph = tab[itab_hph0(sh, NG) + n];
        const double* w = tab + itab_wcol0(sh, NG) + (size_t)n * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) wc[i] = w[i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double a0 = A4[4 * i] * mreg[0], a1 = A4[4 * i + 1] * mreg[1];
        a0 = fma(A4[4 * i + 2], mreg[2], a0); a1 = fma(A4[4 * i + 3], mreg[3], a1);
        Am[i] = a0 + a1;
      }
      fmun = hn * Am[0];
      *p_fmu = fmun; *p_HPH = hph;
      if (stamp) { IH_STAMP(7); }
    }
  };
  if (wave <= 1) head(ip.k_start);
  if (wave == 1) { msp_wave_fence(); msp_link<CD>(x, mc); }      // link tables of the first step
  if (stamp) st_a = __builtin_readcyclecounter();

  for (int64_t k0 = ip.k_start; k0 < T; k0 += KB) {
    const int nb = (T - k0 < KB) ? (int)(T - k0) : KB;
    // ---- fill the ring for steps k0 .. k0+nb-1
    for (int i = tid; i < nb; i += NT) { ry[i] = yv[k0 + i]; rlZ[i] = g_lZ[k0 + i]; rZ[i] = -1.0; }
    for (int i = tid; i < nb * M; i += NT) { rtt[i] = g_tt[(size_t)k0 * M + i]; rtn[i] = g_tn[(size_t)k0 * M + i]; }
    __syncthreads();
    for (int kk = 0; kk < nb; ++kk) {
      const int64_t k = k0 + kk;
      lds_barrier();                 // B1: fmu, HPH of step k; its link tables (written by wave 1 on its way here)
      IH_STAMP(3);
      msp_qv<CD>(x, mc);             // waves 0, 2, 3
      lds_barrier();                 // B2
      IH_STAMP(0);
      msp_stageB<CD>(x, mc, ws);
      lds_barrier();                 // B3
      msp_stage1b<CD>(x, mc, sp, sn2a, ry[kk], ws);
      lds_barrier();                 // B4
      IH_STAMP(1);
      msp_stage2<CD>(x, mc, ws);
      lds_barrier();                 // B5
      IH_STAMP(2);
      if (wave <= 1) {
        msp_reduce<CD>(x);
        msp_wave_fence();
        if (act) {
          const int ko = kk * M;
          double Z, d1, d2;
          msp_outputs<CD>(x.accp, sub, n - D, wrow, pEP1, mc.jitter, Z, d1, d2);
          if (stamp) { asm volatile("" :: "v"(d2)); IH_STAMP(4); }
          const double t_old = p_tt[ko], n_old = p_tn[ko];
          // site update (:265-266): -d2/(1+d2 HPH), (d1 - fmu d2)/(1+d2 HPH) through one reciprocal
          const double r1 = rcp_nr(fma(d2, hph, 1.0));
          double tnew = fma(ip.w_new, -d2 * r1, ip.w_old * t_old);
          const double nnew = fma(ip.w_new, fma(-fmun, d2, d1) * r1, ip.w_old * n_old);
          if (!(tnew > 0.0)) ++n_clamped;
          tnew = max0(tnew);                                               // :274 (NaN -> 0, C-3)
          double Rn = 1.0 / tnew;                                          // R = 1/ttau: the look-up key and an output, exact division
          // (for tnew > 0 the reference's R(:,k) = 1./ttau before the clamp is this value; otherwise :287 overwrites it with Inf)
          // R = inf: ttau = 0, or ttau of underflow size (1/ttau overflows while ys = tnu/ttau stays finite): the reference's
          // (ys - fmu)/(HPH + R) is 0 there; the reciprocal form below would multiply inf by 0
          double g = 0.0;
          if (Rn < INFINITY) g = fma(nnew, Rn, -fmun) * rcp_nr(hph + Rn);  // (ys - fmu)/(HPH + R), ys = tnu/ttau (:277, :289-292)
          typedef double d2v __attribute__((ext_vector_type(2)));
          d2v m01, m23;
          mreg[0] = fma(wc[0], g, Am[0]); mreg[1] = fma(wc[1], g, Am[1]); mreg[2] = fma(wc[2], g, Am[2]); mreg[3] = fma(wc[3], g, Am[3]);
          m01.x = mreg[0]; m01.y = mreg[1]; m23.x = mreg[2]; m23.y = mreg[3];
          p_tt[ko] = tnew; p_tn[ko] = nnew; p_R[ko] = Rn;
          typedef d2v __attribute__((address_space(3))) * lds_d2p;
          lds_d2p mf = (lds_d2p)(p_MF + 4 * ko);
          mf[0] = m01; mf[1] = m23;
          p_fm[ko] = hn * mreg[0];
          Rprev = Rn;
          if (n == 0) rZ[kk] = Z;
          if (stamp) { asm volatile("" :: "v"(Rprev)); IH_STAMP(5); }
        }
        if (k + 1 < T) {
          head(k + 1);
          if (wave == 1) { msp_wave_fence(); msp_link<CD>(x, mc); }     // fmu / HPH of the modulators just written by this wave
        }
      }
    }
    // ---- flush the ring
    __syncthreads();
    for (int i = tid; i < nb; i += NT) g_lZ[k0 + i] = (rZ[i] < 0.0) ? rlZ[i] : log(rZ[i]);
    for (int i = tid; i < nb * M; i += NT) {
      g_tt[(size_t)k0 * M + i] = rtt[i]; g_tn[(size_t)k0 * M + i] = rtn[i]; g_R[(size_t)k0 * M + i] = rR[i];
      g_fm[(size_t)k0 * M + i] = rfm[i];
    }
    for (int i = tid; i < nb * S; i += NT) { const int q = i / S, e = i - q * S; g_MF[(size_t)k0 * S + i] = rMF[(size_t)q * M * 4 + smap[e]]; }
    __syncthreads();
  }
#undef IH_STAMP
  if (act && n_clamped) atomicAdd(&b.counters[(size_t)pb * 4 + 1], (unsigned long long)n_clamped);
  if (stamp)
    for (int i = 0; i < 8; ++i) mc.stamps[i] += st[i];
}

// The same sweep with role-specialised waves (nagp_momsp.hpp, role layout): 512 threads, waves 0 / 1 as above, waves 2..7 run the
// parallel stages of the cubature in their own loop.  A wave holds the registers of its role only, which is what lets two
// waves share a SIMD (the sigma points take one round, the MFMA steps of two waves alternate on the matrix core).
template <int CD, bool PACK>
__global__ void __launch_bounds__(MSR_NT) ihgp_adf8_kernel(Shape sh, Bufs b, MomCfg mc, MomSp sp, IhgpTabs tb, IhgpPar ip) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NT = MSR_NT;
  const int S = sh.S, M = sh.M, D = sh.D, NG = tb.NG;
  const int64_t T = sh.T;
  const int pb = blockIdx.x;
  const double* mdl = b.model + (size_t)pb * mdl_size(sh);
  const double* tab = tb.base + (size_t)pb * itab_size(sh, NG);

  const int KB = ip.kb;
  double* rMF = lds;                                 // ring: m (filtered) [KB][M][4]   (16-byte aligned)
  double* rtt = rMF + (size_t)KB * M * 4;            //       ttau[KB][M]
  double* rtn = rtt + (size_t)KB * M;                //       tnu
  double* rR = rtn + (size_t)KB * M;                 //       R
  double* rfm = rR + (size_t)KB * M;                 //       H*m (filtered)
  double* ry = rfm + (size_t)KB * M;                 //       y[KB]
  double* rlZ = ry + KB;                             //       lZ[KB]
  double* rZ = rlZ + KB;                             //       Z of the step; log taken at the flush
  double* sW = rZ + KB;                              // [D][CD]
  double* fmu = sW + (size_t)D * CD;                 // [68]: M sites, zero padded (stage A of the cubature reads 4*K entries)
  double* HPH = fmu + 68;
  double* rg = HPH + 68;                          // [NG] look-up grid
  double* thph = rg + NG;                            // [M][NG] H PP H' table (ip.hph_lds)
  int* smap = reinterpret_cast<int*>(thph + (ip.hph_lds ? (size_t)M * NG : 0));   // [S] state -> padded ring slot 4*block + row
  double* ws = reinterpret_cast<double*>(smap) + (S + 1) / 2;
  for (int i = tid; i < D * CD; i += NT) sW[i] = mdl[mdl_W(sh) + i];
  for (int i = tid; i < NG; i += NT) rg[i] = tb.r[i];
  if (ip.hph_lds)
    for (int i = tid; i < M * NG; i += NT) thph[i] = tab[itab_hph(sh, NG) + i];
  for (int i = tid; i < 68; i += NT) { fmu[i] = 0.0; HPH[i] = 0.0; }
  for (int i = tid; i < M; i += NT)
    for (int r = 0; r < sh.bsz[i]; ++r) smap[sh.off[i] + r] = 4 * i + r;
  const double sn2 = mdl[mdl_sn2(sh)];
  const double sn2a = sn2 / ip.mom_alpha;
  const double pEP1 = mom_pEP(mc, sn2, ip.mom_alpha);
  const double rmax = tb.r[NG - 1];
  __syncthreads();
  // global rows of this problem (both roles fill and flush the ring)
  const double* yv = b.y + (size_t)pb * T;
  double* g_tt = b.ttau + (size_t)pb * T * M;
  double* g_tn = b.tnu + (size_t)pb * T * M;
  double* g_R = b.R + (size_t)pb * T * M;
  double* g_lZ = b.lZ + (size_t)pb * T;
  double* g_MF = b.MF + (size_t)pb * T * S;
  double* g_fm = b.fm + (size_t)pb * T * M;
  msr_init(CD, D, ws);
  __syncthreads();
  // developer A/B, ip.dbg_wave & 64 (NAGP_STAMP_WORKER): tables / q0 / s0 of a step on every worker wave for itself (msr_fold) instead
  // of on wave 1 / workers 3, 4 between two barriers.  Measured slower (profiles/r03_stamps_cycles.txt): a lone wave issues a dependent
  // VALU instruction every 8 - 10 cycles, so the ~80 instructions of the stage on EVERY worker cost more than barrier B3 saves.
  const bool fold = (CD * CD <= 48) && (ip.dbg_wave & 64) != 0;
  if (wave >= MSR_W0) {
    // ================= worker role: the parallel stages of the cubature; the same barriers as the serial role below
    const MspLay lay = msp_layout(CD, D, 1);
    MsrW<CD, PACK> xw;
    msr_setup_W<CD, PACK>(xw, mc, sp, sW, fmu, HPH, ws);
    if (ip.dbg_wave & 8) xw.m_on = (wave - MSR_W0 < MSR_NWK) ? 1 : 0;   // developer A/B (NAGP_STAMP_WORKER=8): every worker runs the accumulation stage
    // developer diagnostics (NAGP_STAMPS): time lines of worker 0 (an MFMA worker) in stamps[8..15] and of the last worker (marginal
    // sums in the packed form) in stamps[16..23]
    const int wk_slot = (wave == MSR_W0 + (ip.dbg_wave & 7)) ? 8 : ((wave == MSR_W0 + MSR_NWK - 1) ? 16 : -1);
    const bool wk_stamp = mc.stamps && wk_slot >= 0 && (tid & 63) == 0;
    unsigned long long wk_a = 0, wk_b = 0, wk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (wk_stamp) wk_a = __builtin_readcyclecounter();
#define WK_STAMP(slot) do { if (wk_stamp) { wk_b = __builtin_readcyclecounter(); wk[slot] += wk_b - wk_a; wk_a = wk_b; } } while (0)
  for (int64_t k0 = ip.k_start; k0 < T; k0 += KB) {
    const int nb = (T - k0 < KB) ? (int)(T - k0) : KB;
    // ---- fill the ring for steps k0 .. k0+nb-1
    for (int i = tid; i < nb; i += NT) { ry[i] = yv[k0 + i]; rlZ[i] = g_lZ[k0 + i]; rZ[i] = -1.0; }
    for (int i = tid; i < nb * M; i += NT) { rtt[i] = g_tt[(size_t)k0 * M + i]; rtn[i] = g_tn[(size_t)k0 * M + i]; }
    __syncthreads();
    for (int kk = 0; kk < nb; ++kk) {
      lds_barrier();                 // B1
      WK_STAMP(0);                   // (wait at B1: the serial waves' tail and head)
      msp_qv<CD>(xw, mc);            // workers 0..2
      WK_STAMP(1);
      lds_barrier();                 // B2
      if (fold) {                    // tables, q0, s0 on every worker wave for itself: no barrier B3
        double q0, s0;
        if constexpr (CD * CD <= 48) msr_fold<CD>(xw, mc, q0, s0); else { q0 = 0.0; s0 = 0.0; }
        WK_STAMP(2);
        msp_stage1b_qs<CD>(xw, mc, sp, sn2a, ry[kk], q0, s0);
      } else {
        if (wave == MSR_W0 + 3 || wave == MSR_W0 + 4) msr_q0_or_s0(xw, wave == MSR_W0 + 3, ws + lay.q0, ws + lay.s0);
        lds_barrier();                 // B3
        WK_STAMP(2);                   // (B2 .. B3: tables on wave 1, q0 / s0 on worker 3)
        msp_stage1b<CD>(xw, mc, sp, sn2a, ry[kk], ws);
      }
      WK_STAMP(3);
      lds_barrier();                 // B4
      WK_STAMP(4);
      if constexpr (PACK) { if (wave >= MSR_W0 + MSR_NWK - 2) msr_marginals<CD>(xw); }
      WK_STAMP(5);
      msp_stage2<CD>(xw, mc, ws);
      WK_STAMP(6);
      lds_barrier();                 // B5
      WK_STAMP(7);
    }
    // ---- flush the ring
    __syncthreads();
    for (int i = tid; i < nb; i += NT) g_lZ[k0 + i] = (rZ[i] < 0.0) ? rlZ[i] : log(rZ[i]);
    for (int i = tid; i < nb * M; i += NT) {
      g_tt[(size_t)k0 * M + i] = rtt[i]; g_tn[(size_t)k0 * M + i] = rtn[i]; g_R[(size_t)k0 * M + i] = rR[i];
      g_fm[(size_t)k0 * M + i] = rfm[i];
    }
    for (int i = tid; i < nb * S; i += NT) { const int q = i / S, e = i - q * S; g_MF[(size_t)k0 * S + i] = rMF[(size_t)q * M * 4 + smap[e]]; }
    __syncthreads();
  }
    if (wk_stamp)
      for (int i = 0; i < 8; ++i) mc.stamps[wk_slot + i] += wk[i];
#undef WK_STAMP
    return;
  }
  // ================= serial role (waves 0 and 1)
  MsrS<CD, PACK> x;
  msr_setup_S<CD, PACK>(x, mc, sp, fmu, HPH, ws);

  // wave 0, lane d < D owns sub-band block d; wave 1, lane j < N owns modulator block D + j
  const int lane = tid & 63;
  const bool sub = (wave == 0) && lane < D;
  const bool act = sub || ((wave == 1) && lane < sh.N);
  const int n = (wave == 0) ? lane : D + lane;
  const int nn = act ? n : 0;
  double A4[16], mreg[4] = {0, 0, 0, 0}, wrow[CD];
  double hn = 0.0;
#pragma unroll
  for (int j = 0; j < CD; ++j) wrow[j] = 0.0;
  tile_zero(A4);
  if (act) {
    tile_load(A4, mdl + mdl_A(sh) + (size_t)n * 16);
    hn = mdl[mdl_h(sh) + n];
    const int o = sh.off[n], bs = sh.bsz[n];
    if (sub) {
#pragma unroll
      for (int j = 0; j < CD; ++j) wrow[j] = sW[n * CD + j];
    }
    if (ip.k_start > 0) {      // continue from the filtered mean of the previous step
      const double* mp = b.MF + ((size_t)pb * T + (ip.k_start - 1)) * S;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < bs) mreg[i] = mp[o + i];
    } else if (ip.itt > 1) {   // m is NOT reset between sweeps (SURVEY C-22): smoothed mean at k=0
      const double* ms0 = b.MS + (size_t)pb * T * S;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < bs) mreg[i] = ms0[o + i];
    }
  }
  // per-lane ring / table addresses (vector registers; the step index adds an immediate-free scalar offset)
  const msp_wp p_tt = (msp_wp)(rtt + nn), p_tn = (msp_wp)(rtn + nn), p_R = (msp_wp)(rR + nn), p_fm = (msp_wp)(rfm + nn);
  const msp_wp p_MF = (msp_wp)(rMF + 4 * nn);
  const msp_wp p_fmu = (msp_wp)(fmu + nn), p_HPH = (msp_wp)(HPH + nn);
  const msp_rp p_hph = (msp_rp)(thph + (size_t)nn * NG);
  const msp_rp p_rg = (msp_rp)rg + opaque_zero();
  const double* g_wcol = tab + itab_wcol(sh, NG) + (size_t)nn * NG * 4;
  const double* g_hph = tab + itab_hph(sh, NG) + (size_t)nn * NG;
  double Rprev = (act && ip.k_start > 0) ? b.R[((size_t)pb * T + (ip.k_start - 1)) * M + n] : 0.0;
  unsigned int n_clamped = 0;
  unsigned long long st_a = 0, st_b = 0, st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool stamp = mc.stamps && tid == ((ip.dbg_wave & 32) ? 64 : 0);      // time line of wave 0 (NAGP_STAMP_WORKER & 32: of wave 1)
#define IH_STAMP(slot) do { if (stamp) { st_b = __builtin_readcyclecounter(); st[slot] += st_b - st_a; st_a = st_b; } } while (0)

  // head of step k: table look-up, A m, the cubature's inputs (wave 0, no barrier)
  double hph = 0.0, wc[4] = {0, 0, 0, 0}, Am[4] = {0, 0, 0, 0}, fmun = 0.0;
  auto head = [&](int64_t k) {
    if (act) {
      if (k > 0) {
        const int idx = nearest_idx3(p_rg, NG, tb.lr0, tb.inv_dlr, rmax, Rprev);
        if (stamp) { asm volatile("" :: "v"(idx)); IH_STAMP(6); }
        hph = ip.hph_lds ? p_hph[idx] : g_hph[idx];
        const double2* w = reinterpret_cast<const double2*>(g_wcol + (size_t)idx * 4);
        const double2 w0 = w[0], w1 = w[1];
        wc[0] = w0.x; wc[1] = w0.y; wc[2] = w1.x; wc[3] = w1.y;
      } else {
        hph = tab[itab_hph0(sh, NG) + n];
        const double* w = tab + itab_wcol0(sh, NG) + (size_t)n * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) wc[i] = w[i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double a0 = A4[4 * i] * mreg[0], a1 = A4[4 * i + 1] * mreg[1];
        a0 = fma(A4[4 * i + 2], mreg[2], a0); a1 = fma(A4[4 * i + 3], mreg[3], a1);
        Am[i] = a0 + a1;
      }
      fmun = hn * Am[0];
      *p_fmu = fmun; *p_HPH = hph;
      if (stamp) { IH_STAMP(7); }
    }
  };
  if (wave <= 1) head(ip.k_start);
  // link tables of a step (the exp / log chain of the modulators' sigma-point coordinates, ~1 000 cycles on wave 1): evaluated BEHIND
  // barrier B1, beside the worker waves' Q / 2Q / v stage -- nothing reads them before wave 1's own msp_tables behind B2.  (Evaluated
  // ahead of B1 they were the tail of the serial chain: wave 0 waited ~800 cycles at B1 for them.)  ip.dbg_wave & 16: the old placement.
  const bool link_early = (ip.dbg_wave & 16) != 0;
  if (wave == 1 && link_early) { msp_wave_fence(); msp_link<CD>(x, mc); }      // link tables of the first step
  if (stamp) st_a = __builtin_readcyclecounter();

  for (int64_t k0 = ip.k_start; k0 < T; k0 += KB) {
    const int nb = (T - k0 < KB) ? (int)(T - k0) : KB;
    // ---- fill the ring for steps k0 .. k0+nb-1
    for (int i = tid; i < nb; i += NT) { ry[i] = yv[k0 + i]; rlZ[i] = g_lZ[k0 + i]; rZ[i] = -1.0; }
    for (int i = tid; i < nb * M; i += NT) { rtt[i] = g_tt[(size_t)k0 * M + i]; rtn[i] = g_tn[(size_t)k0 * M + i]; }
    __syncthreads();
    for (int kk = 0; kk < nb; ++kk) {
      const int64_t k = k0 + kk;
      lds_barrier();                 // B1: fmu, HPH of step k; its link tables (written by wave 1 on its way here)
      IH_STAMP(3);
      // (Q / 2Q / v: worker waves)
      if (wave == 1 && !link_early) msp_link<CD>(x, mc);
      lds_barrier();                 // B2
      IH_STAMP(0);
      if (!fold) {
        if (wave == 1) msp_tables<CD>(x, mc);
        lds_barrier();                 // B3
      }
      // (tables, q0, s0, Gaussian weights: worker waves)
      lds_barrier();                 // B4
      IH_STAMP(1);
      // (MFMA sums: worker waves)
      lds_barrier();                 // B5
      IH_STAMP(2);
      {
        msp_reduce<CD>(x);
        msp_wave_fence();
        if (act) {
          const int ko = kk * M;
          double Z, d1, d2;
          msp_outputs<CD>(x.accp, sub, n - D, wrow, pEP1, mc.jitter, Z, d1, d2);
          if (stamp) { asm volatile("" :: "v"(d2)); IH_STAMP(4); }
          const double t_old = p_tt[ko], n_old = p_tn[ko];
          // site update (:265-266): -d2/(1+d2 HPH), (d1 - fmu d2)/(1+d2 HPH) through one reciprocal
          const double r1 = rcp_nr(fma(d2, hph, 1.0));
          double tnew = fma(ip.w_new, -d2 * r1, ip.w_old * t_old);
          const double nnew = fma(ip.w_new, fma(-fmun, d2, d1) * r1, ip.w_old * n_old);
          if (!(tnew > 0.0)) ++n_clamped;
          tnew = max0(tnew);                                               // :274 (NaN -> 0, C-3)
          double Rn = 1.0 / tnew;                                          // R = 1/ttau: the look-up key and an output, exact division
          // (for tnew > 0 the reference's R(:,k) = 1./ttau before the clamp is this value; otherwise :287 overwrites it with Inf)
          // R = inf: ttau = 0, or ttau of underflow size (1/ttau overflows while ys = tnu/ttau stays finite): the reference's
          // (ys - fmu)/(HPH + R) is 0 there; the reciprocal form below would multiply inf by 0
          double g = 0.0;
          if (Rn < INFINITY) g = fma(nnew, Rn, -fmun) * rcp_nr(hph + Rn);  // (ys - fmu)/(HPH + R), ys = tnu/ttau (:277, :289-292)
          typedef double d2v __attribute__((ext_vector_type(2)));
          d2v m01, m23;
          mreg[0] = fma(wc[0], g, Am[0]); mreg[1] = fma(wc[1], g, Am[1]); mreg[2] = fma(wc[2], g, Am[2]); mreg[3] = fma(wc[3], g, Am[3]);
          m01.x = mreg[0]; m01.y = mreg[1]; m23.x = mreg[2]; m23.y = mreg[3];
          p_tt[ko] = tnew; p_tn[ko] = nnew; p_R[ko] = Rn;
          typedef d2v __attribute__((address_space(3))) * lds_d2p;
          lds_d2p mf = (lds_d2p)(p_MF + 4 * ko);
          mf[0] = m01; mf[1] = m23;
          p_fm[ko] = hn * mreg[0];
          Rprev = Rn;
          if (n == 0) rZ[kk] = Z;
          if (stamp) { asm volatile("" :: "v"(Rprev)); IH_STAMP(5); }
        }
        if (k + 1 < T) {
          head(k + 1);
          if (wave == 1 && link_early) { msp_wave_fence(); msp_link<CD>(x, mc); }     // fmu / HPH of the modulators just written by this wave
        }
      }
    }
    // ---- flush the ring
    __syncthreads();
    for (int i = tid; i < nb; i += NT) g_lZ[k0 + i] = (rZ[i] < 0.0) ? rlZ[i] : log(rZ[i]);
    for (int i = tid; i < nb * M; i += NT) {
      g_tt[(size_t)k0 * M + i] = rtt[i]; g_tn[(size_t)k0 * M + i] = rtn[i]; g_R[(size_t)k0 * M + i] = rR[i];
      g_fm[(size_t)k0 * M + i] = rfm[i];
    }
    for (int i = tid; i < nb * S; i += NT) { const int q = i / S, e = i - q * S; g_MF[(size_t)k0 * S + i] = rMF[(size_t)q * M * 4 + smap[e]]; }
    __syncthreads();
  }
#undef IH_STAMP
  if (act && n_clamped) atomicAdd(&b.counters[(size_t)pb * 4 + 1], (unsigned long long)n_clamped);
  if (stamp)
    for (int i = 0; i < 8; ++i) mc.stamps[i] += st[i];
}

// The role-specialised sweep for likModulatorPreCalcwn (experiments/likModulatorPreCalcwn.m:28-86; nagp_momsq.hpp): 512 threads, waves
// 0 / 1 carry the sites exactly as in ihgp_adf8_kernel, waves 2 .. MSQ_NWK + 1 the staged cubature of the square-root amplitudes.  sp.c0 = code of
// the centre coordinate.  D <= 32 sub-bands, <= 6 components, <= 336 sigma points (the host checks).
struct MsqS { int lw; double xdc; msp_rp a_mu, a_s2; msp_wp a_out; msp_rp accp, partp; };
__host__ __device__ inline size_t ihgp_adf8sq_lds_doubles(const Shape& s, int CD, int NG, int hph_lds, int kb) {
  return ihgp_adf_lds_doubles(s, CD, NG, hph_lds, kb) - msp_lds_doubles(CD, s.D) + msq_lds_doubles<MsqRole>(CD) + 512;
}
template <int CD>
__global__ void __launch_bounds__(MSQ_NT) ihgp_adf8sq_kernel(Shape sh, Bufs b, MomCfg mc, MomSp sp, IhgpTabs tb, IhgpPar ip) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NT = MSQ_NT;
  const int S = sh.S, M = sh.M, D = sh.D, NG = tb.NG;
  const int64_t T = sh.T;
  const int pb = blockIdx.x;
  const double* mdl = b.model + (size_t)pb * mdl_size(sh);
  const double* tab = tb.base + (size_t)pb * itab_size(sh, NG);

  const int KB = ip.kb;
  double* rMF = lds;                                 // ring: m (filtered) [KB][M][4]   (16-byte aligned)
  double* rtt = rMF + (size_t)KB * M * 4;            //       ttau[KB][M]
  double* rtn = rtt + (size_t)KB * M;                //       tnu
  double* rR = rtn + (size_t)KB * M;                 //       R
  double* rfm = rR + (size_t)KB * M;                 //       H*m (filtered)
  double* ry = rfm + (size_t)KB * M;                 //       y[KB]
  double* rlZ = ry + KB;                             //       lZ[KB]
  double* rZ = rlZ + KB;                             //       Z of the step; log taken at the flush
  double* sW = rZ + KB;                              // [D][CD]
  double* fmu = sW + (size_t)D * CD;                 // [68]: M sites, zero padded (stage A of the cubature reads 4*K entries)
  double* HPH = fmu + 68;
  double* rg = HPH + 68;                          // [NG] look-up grid
  double* thph = rg + NG;                            // [M][NG] H PP H' table (ip.hph_lds)
  int* smap = reinterpret_cast<int*>(thph + (ip.hph_lds ? (size_t)M * NG : 0));   // [S] state -> padded ring slot 4*block + row
  double* wt = reinterpret_cast<double*>(smap) + (S + 1) / 2;      // [8][64] W transposed for t = W' s2 (stage A)
  double* ws = wt + 512;
  for (int i = tid; i < D * CD; i += NT) sW[i] = mdl[mdl_W(sh) + i];
  for (int i = tid; i < NG; i += NT) rg[i] = tb.r[i];
  if (ip.hph_lds)
    for (int i = tid; i < M * NG; i += NT) thph[i] = tab[itab_hph(sh, NG) + i];
  for (int i = tid; i < 68; i += NT) { fmu[i] = 0.0; HPH[i] = 0.0; }
  for (int i = tid; i < M; i += NT)
    for (int r = 0; r < sh.bsz[i]; ++r) smap[sh.off[i] + r] = 4 * i + r;
  const double sn2 = mdl[mdl_sn2(sh)];
  const double sn2a = sn2 / ip.mom_alpha;
  const double pEP1 = mom_pEP(mc, sn2, ip.mom_alpha);
  const double rmax = tb.r[NG - 1];
  __syncthreads();
  // global rows of this problem (both roles fill and flush the ring)
  const double* yv = b.y + (size_t)pb * T;
  double* g_tt = b.ttau + (size_t)pb * T * M;
  double* g_tn = b.tnu + (size_t)pb * T * M;
  double* g_R = b.R + (size_t)pb * T * M;
  double* g_lZ = b.lZ + (size_t)pb * T;
  double* g_MF = b.MF + (size_t)pb * T * S;
  double* g_fm = b.fm + (size_t)pb * T * M;
  msq_init<MsqRole>(CD, ws, NT);
  __syncthreads();
  const MsqLay lay = msq_layout<MsqRole>(CD);
  if (wave >= MSR_W0) {
    // ================= worker role: the parallel stages of the cubature; the same barriers as the serial role below
    MsqW<CD, MsqRole> xw;
    msq_setup_W<CD, MsqRole>(xw, mc, sp.c0, sW, fmu, HPH, ws, wave - MSR_W0, tid - 64 * MSR_W0, wt);
    double amp[2 * MsqRole::NST];
#pragma unroll
    for (int i = 0; i < 2 * MsqRole::NST; ++i) amp[i] = 0.0;
    // developer diagnostics (NAGP_STAMPS): time lines of worker 0 in stamps[8..15] and of the last worker (marginal sums) in stamps[16..23]
    const int wk_slot = (wave == MSR_W0 + (ip.dbg_wave & 7)) ? 8 : ((wave == MSR_W0 + MSQ_NWK - 1) ? 16 : -1);
    const bool wk_stamp = mc.stamps && wk_slot >= 0 && (tid & 63) == 0;
    unsigned long long wk_a = 0, wk_b = 0, wk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (wk_stamp) wk_a = __builtin_readcyclecounter();
#define WK_STAMP(slot) do { if (wk_stamp) { wk_b = __builtin_readcyclecounter(); wk[slot] += wk_b - wk_a; wk_a = wk_b; } } while (0)
  for (int64_t k0 = ip.k_start; k0 < T; k0 += KB) {
    const int nb = (T - k0 < KB) ? (int)(T - k0) : KB;
    // ---- fill the ring for steps k0 .. k0+nb-1
    for (int i = tid; i < nb; i += NT) { ry[i] = yv[k0 + i]; rlZ[i] = g_lZ[k0 + i]; rZ[i] = -1.0; }
    for (int i = tid; i < nb * M; i += NT) { rtt[i] = g_tt[(size_t)k0 * M + i]; rtn[i] = g_tn[(size_t)k0 * M + i]; }
    __syncthreads();
    for (int kk = 0; kk < nb; ++kk) {
      lds_barrier();                 // B1
      WK_STAMP(0);                   // (wait at B1: the serial waves' tail and head)
      msq_stageA<CD, MsqRole>(xw);            // worker 0: t = W' s2_z
      WK_STAMP(1);
      lds_barrier();                 // B2: link tables (wave 1)
      msq_stageS<CD, MsqRole>(xw, amp);       // gather, square roots, mu_p
      WK_STAMP(2);
      lds_barrier();                 // B3
      msq_stage1b<CD, MsqRole>(xw, sn2a, ry[kk]);
      WK_STAMP(3);
      lds_barrier();                 // B4
      WK_STAMP(4);
      WK_STAMP(5);
      msq_stageS2<CD, MsqRole>(xw, amp);
      WK_STAMP(6);
      lds_barrier();                 // B5
      WK_STAMP(7);
    }
    // ---- flush the ring
    __syncthreads();
    for (int i = tid; i < nb; i += NT) g_lZ[k0 + i] = (rZ[i] < 0.0) ? rlZ[i] : log(rZ[i]);
    for (int i = tid; i < nb * M; i += NT) {
      g_tt[(size_t)k0 * M + i] = rtt[i]; g_tn[(size_t)k0 * M + i] = rtn[i]; g_R[(size_t)k0 * M + i] = rR[i];
      g_fm[(size_t)k0 * M + i] = rfm[i];
    }
    for (int i = tid; i < nb * S; i += NT) { const int q = i / S, e = i - q * S; g_MF[(size_t)k0 * S + i] = rMF[(size_t)q * M * 4 + smap[e]]; }
    __syncthreads();
  }
    if (wk_stamp)
      for (int i = 0; i < 8; ++i) mc.stamps[wk_slot + i] += wk[i];
#undef WK_STAMP
    return;
  }
  // ================= serial role (waves 0 and 1)
  MsqS x;
  {
    const int nd = mc.nd, TN = CD * nd, tl = tid - 64;
    const int t = (tl >= 0 && tl < TN) ? tl : 0;
    const int j = t / nd, cc = t - j * nd;
    x.lw = 1; x.xdc = mc.xd[cc];
    x.a_mu = (msp_rp)(fmu + D + j); x.a_s2 = (msp_rp)(HPH + D + j);
    x.a_out = (msp_wp)(ws + t);
    x.accp = (msp_rp)(ws + lay.acc + ((wave == 1) ? 32 : 0)) + opaque_zero();
    const int dl = tid & 63;
    x.partp = (msp_rp)(ws + lay.part + (dl & 15) + 16 * ((dl >> 4) & 1));
  }
  MsqM xm;      // marginal sums of c0 -> g1, g2, Z: the serial waves, idle between B4 and B5, take half of the dimensions each
  msq_setup_M<CD, MsqRole>(xm, mc, sp.c0, ws, wave);

  // wave 0, lane d < D owns sub-band block d; wave 1, lane j < N owns modulator block D + j
  const int lane = tid & 63;
  const bool sub = (wave == 0) && lane < D;
  const bool act = sub || ((wave == 1) && lane < sh.N);
  const int n = (wave == 0) ? lane : D + lane;
  const int nn = act ? n : 0;
  double A4[16], mreg[4] = {0, 0, 0, 0};
  double hn = 0.0;
  tile_zero(A4);
  if (act) {
    tile_load(A4, mdl + mdl_A(sh) + (size_t)n * 16);
    hn = mdl[mdl_h(sh) + n];
    const int o = sh.off[n], bs = sh.bsz[n];
    if (ip.k_start > 0) {      // continue from the filtered mean of the previous step
      const double* mp = b.MF + ((size_t)pb * T + (ip.k_start - 1)) * S;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < bs) mreg[i] = mp[o + i];
    } else if (ip.itt > 1) {   // m is NOT reset between sweeps (SURVEY C-22): smoothed mean at k=0
      const double* ms0 = b.MS + (size_t)pb * T * S;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < bs) mreg[i] = ms0[o + i];
    }
  }
  // per-lane ring / table addresses (vector registers; the step index adds an immediate-free scalar offset)
  const msp_wp p_tt = (msp_wp)(rtt + nn), p_tn = (msp_wp)(rtn + nn), p_R = (msp_wp)(rR + nn), p_fm = (msp_wp)(rfm + nn);
  const msp_wp p_MF = (msp_wp)(rMF + 4 * nn);
  const msp_wp p_fmu = (msp_wp)(fmu + nn), p_HPH = (msp_wp)(HPH + nn);
  const msp_rp p_hph = (msp_rp)(thph + (size_t)nn * NG);
  const msp_rp p_rg = (msp_rp)rg + opaque_zero();
  const double* g_wcol = tab + itab_wcol(sh, NG) + (size_t)nn * NG * 4;
  const double* g_hph = tab + itab_hph(sh, NG) + (size_t)nn * NG;
  double Rprev = (act && ip.k_start > 0) ? b.R[((size_t)pb * T + (ip.k_start - 1)) * M + n] : 0.0;
  unsigned int n_clamped = 0;
  unsigned long long st_a = 0, st_b = 0, st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool stamp = mc.stamps && tid == ((ip.dbg_wave & 32) ? 64 : 0);      // time line of wave 0 (NAGP_STAMP_WORKER & 32: of wave 1)
#define IH_STAMP(slot) do { if (stamp) { st_b = __builtin_readcyclecounter(); st[slot] += st_b - st_a; st_a = st_b; } } while (0)

  // head of step k: table look-up, A m, the cubature's inputs (wave 0, no barrier)
  double hph = 0.0, wc[4] = {0, 0, 0, 0}, Am[4] = {0, 0, 0, 0}, fmun = 0.0;
  auto head = [&](int64_t k) {
    if (act) {
      if (k > 0) {
        const int idx = nearest_idx3(p_rg, NG, tb.lr0, tb.inv_dlr, rmax, Rprev);
        if (stamp) { asm volatile("" :: "v"(idx)); IH_STAMP(6); }
        hph = ip.hph_lds ? p_hph[idx] : g_hph[idx];
        const double2* w = reinterpret_cast<const double2*>(g_wcol + (size_t)idx * 4);
        const double2 w0 = w[0], w1 = w[1];
        wc[0] = w0.x; wc[1] = w0.y; wc[2] = w1.x; wc[3] = w1.y;
      } else {
        hph = tab[itab_hph0(sh, NG) + n];
        const double* w = tab + itab_wcol0(sh, NG) + (size_t)n * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) wc[i] = w[i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double a0 = A4[4 * i] * mreg[0], a1 = A4[4 * i + 1] * mreg[1];
        a0 = fma(A4[4 * i + 2], mreg[2], a0); a1 = fma(A4[4 * i + 3], mreg[3], a1);
        Am[i] = a0 + a1;
      }
      fmun = hn * Am[0];
      *p_fmu = fmun; *p_HPH = hph;
      if (stamp) { IH_STAMP(7); }
    }
  };
  if (wave <= 1) head(ip.k_start);
  // link tables of a step (the exp / log chain of the modulators' sigma-point coordinates, ~1 000 cycles on wave 1): evaluated BEHIND
  // barrier B1, beside the worker waves' Q / 2Q / v stage -- nothing reads them before wave 1's own msp_tables behind B2.  (Evaluated
  // ahead of B1 they were the tail of the serial chain: wave 0 waited ~800 cycles at B1 for them.)  ip.dbg_wave & 16: the old placement.
  const bool link_early = (ip.dbg_wave & 16) != 0;
  if (wave == 1 && link_early) { msp_wave_fence(); msp_link<CD>(x, mc); }      // link tables of the first step
  if (stamp) st_a = __builtin_readcyclecounter();

  for (int64_t k0 = ip.k_start; k0 < T; k0 += KB) {
    const int nb = (T - k0 < KB) ? (int)(T - k0) : KB;
    // ---- fill the ring for steps k0 .. k0+nb-1
    for (int i = tid; i < nb; i += NT) { ry[i] = yv[k0 + i]; rlZ[i] = g_lZ[k0 + i]; rZ[i] = -1.0; }
    for (int i = tid; i < nb * M; i += NT) { rtt[i] = g_tt[(size_t)k0 * M + i]; rtn[i] = g_tn[(size_t)k0 * M + i]; }
    __syncthreads();
    for (int kk = 0; kk < nb; ++kk) {
      const int64_t k = k0 + kk;
      lds_barrier();                 // B1: fmu, HPH of step k; its link tables (written by wave 1 on its way here)
      IH_STAMP(3);
      // (t = W' s2_z: worker 0)
      if (wave == 1 && !link_early) msp_link<CD>(x, mc);
      lds_barrier();                 // B2
      IH_STAMP(0);
      // (gather, square roots, mu_p: worker waves)
      lds_barrier();                 // B3
      // (Gaussian weights: worker waves)
      lds_barrier();                 // B4
      IH_STAMP(1);
      msq_marginals<CD, MsqRole>(xm);         // (sum c1 a, sum c2 a^2: worker waves)
      lds_barrier();                 // B5
      IH_STAMP(2);
      {
        if (act) {
          const int ko = kk * M;
          double Z, d1, d2;
          msq_outputs<CD, MsqRole>(x.accp, x.partp, sub, n - D, pEP1, mc.jitter, Z, d1, d2);
          if (stamp) { asm volatile("" :: "v"(d2)); IH_STAMP(4); }
          const double t_old = p_tt[ko], n_old = p_tn[ko];
          // site update (:265-266): -d2/(1+d2 HPH), (d1 - fmu d2)/(1+d2 HPH) through one reciprocal
          const double r1 = rcp_nr(fma(d2, hph, 1.0));
          double tnew = fma(ip.w_new, -d2 * r1, ip.w_old * t_old);
          const double nnew = fma(ip.w_new, fma(-fmun, d2, d1) * r1, ip.w_old * n_old);
          if (!(tnew > 0.0)) ++n_clamped;
          tnew = max0(tnew);                                               // :274 (NaN -> 0, C-3)
          double Rn = 1.0 / tnew;                                          // R = 1/ttau: the look-up key and an output, exact division
          // (for tnew > 0 the reference's R(:,k) = 1./ttau before the clamp is this value; otherwise :287 overwrites it with Inf)
          // R = inf: ttau = 0, or ttau of underflow size (1/ttau overflows while ys = tnu/ttau stays finite): the reference's
          // (ys - fmu)/(HPH + R) is 0 there; the reciprocal form below would multiply inf by 0
          double g = 0.0;
          if (Rn < INFINITY) g = fma(nnew, Rn, -fmun) * rcp_nr(hph + Rn);  // (ys - fmu)/(HPH + R), ys = tnu/ttau (:277, :289-292)
          typedef double d2v __attribute__((ext_vector_type(2)));
          d2v m01, m23;
          mreg[0] = fma(wc[0], g, Am[0]); mreg[1] = fma(wc[1], g, Am[1]); mreg[2] = fma(wc[2], g, Am[2]); mreg[3] = fma(wc[3], g, Am[3]);
          m01.x = mreg[0]; m01.y = mreg[1]; m23.x = mreg[2]; m23.y = mreg[3];
          p_tt[ko] = tnew; p_tn[ko] = nnew; p_R[ko] = Rn;
          typedef d2v __attribute__((address_space(3))) * lds_d2p;
          lds_d2p mf = (lds_d2p)(p_MF + 4 * ko);
          mf[0] = m01; mf[1] = m23;
          p_fm[ko] = hn * mreg[0];
          Rprev = Rn;
          if (n == 0) rZ[kk] = Z;
          if (stamp) { asm volatile("" :: "v"(Rprev)); IH_STAMP(5); }
        }
        if (k + 1 < T) {
          head(k + 1);
          if (wave == 1 && link_early) { msp_wave_fence(); msp_link<CD>(x, mc); }     // fmu / HPH of the modulators just written by this wave
        }
      }
    }
    // ---- flush the ring
    __syncthreads();
    for (int i = tid; i < nb; i += NT) g_lZ[k0 + i] = (rZ[i] < 0.0) ? rlZ[i] : log(rZ[i]);
    for (int i = tid; i < nb * M; i += NT) {
      g_tt[(size_t)k0 * M + i] = rtt[i]; g_tn[(size_t)k0 * M + i] = rtn[i]; g_R[(size_t)k0 * M + i] = rR[i];
      g_fm[(size_t)k0 * M + i] = rfm[i];
    }
    for (int i = tid; i < nb * S; i += NT) { const int q = i / S, e = i - q * S; g_MF[(size_t)k0 * S + i] = rMF[(size_t)q * M * 4 + smap[e]]; }
    __syncthreads();
  }
#undef IH_STAMP
  if (act && n_clamped) atomicAdd(&b.counters[(size_t)pb * 4 + 1], (unsigned long long)n_clamped);
  if (stamp)
    for (int i = 0; i < 8; ++i) mc.stamps[i] += st[i];
}

// Backward mean recursion: m <- MF_k + G (m - A MF_k) with (P,G) looked up from R(:,k)
// (Inf -> last grid row, ihgp_ep_modulator_nmf.m:379-380).  One wave per problem, thread n = block n.
template <int BS>
static __global__ void __launch_bounds__(64) ihgp_scan_kernel(Shape sh, Bufs b, IhgpTabs tb, double* vprev /* [B][M] */) {
  const int n = threadIdx.x, pb = blockIdx.x;
  const int S = sh.S, M = sh.M, NG = tb.NG;
  const int64_t T = sh.T;
  const double* mdl = b.model + (size_t)pb * mdl_size(sh);
  const double* tab = tb.base + (size_t)pb * itab_size(sh, NG);
  double mxM = 0.0, mxP = 0.0;
  if (n < M) {
    double A4[BS * BS], mreg[BS];
#pragma unroll
    for (int i = 0; i < BS; ++i) mreg[i] = 0.0;
#pragma unroll
    for (int e = 0; e < BS * BS; ++e) A4[e] = mdl[mdl_A(sh) + (size_t)n * BS * BS + e];
    const double hn = mdl[mdl_h(sh) + n];
    const int o = sh.off[n], bs = sh.bsz[n];
    const double* g_MF = b.MF + (size_t)pb * T * S;
    double* g_MS = b.MS + (size_t)pb * T * S;
    const double* g_R = b.R + (size_t)pb * T * M;
    double* g_sm = b.sm + (size_t)pb * T * M;
    double* g_sv = b.sv + (size_t)pb * T * M;
    #pragma unroll
    for (int i = 0; i < BS; ++i) if (i < bs) { mreg[i] = g_MF[(size_t)(T - 1) * S + o + i]; g_MS[(size_t)(T - 1) * S + o + i] = mreg[i]; }
    double vlast = 0.0;   // P = zeros(size(A)) before the loop (:364)
    for (int64_t k = T - 2; k >= 0; --k) {
      const double Rk = g_R[(size_t)k * M + n];
      int idx = nearest_idx(tb, Rk);
      if (isinf(Rk)) idx = NG - 1;
      double G4[BS * BS], mf[BS], d[BS];
#pragma unroll
      for (int i = 0; i < BS; ++i) mf[i] = 0.0;
#pragma unroll
      for (int e = 0; e < BS * BS; ++e) G4[e] = tab[itab_g(sh, NG) + ((size_t)n * NG + idx) * BS * BS + e];
      vlast = tab[itab_v(sh, NG) + (size_t)n * NG + idx];
      #pragma unroll
      for (int i = 0; i < BS; ++i) if (i < bs) mf[i] = g_MF[(size_t)k * S + o + i];
#pragma unroll
      for (int i = 0; i < BS; ++i) {
        double a = mreg[i];
#pragma unroll
        for (int l = 0; l < BS; ++l) a = fma(-A4[BS * i + l], mf[l], a);
        d[i] = a;
      }
#pragma unroll
      for (int i = 0; i < BS; ++i) {
        double a = mf[i];
#pragma unroll
        for (int l = 0; l < BS; ++l) a = fma(G4[BS * i + l], d[l], a);
        mreg[i] = a;
      }
      #pragma unroll
      for (int i = 0; i < BS; ++i) if (i < bs) g_MS[(size_t)k * S + o + i] = mreg[i];
      const double mnew = hn * mreg[0];
      mxM = fmax(mxM, fabs(g_sm[(size_t)k * M + n] - mnew));
      g_sm[(size_t)k * M + n] = mnew;
      g_sv[(size_t)k * M + n] = vlast;
    }
    // maxDiffP = max|H*PSP*H' - H*P*H'| with P = the last looked-up blocks (k = 0)
    mxP = fabs(vprev[(size_t)pb * M + n] - vlast);
    vprev[(size_t)pb * M + n] = vlast;
  }
  mxM = wave_max(mxM);
  mxP = wave_max(mxP);
  if (n == 0) { b.red[(size_t)pb * 8 + 1] = mxM; b.red[(size_t)pb * 8 + 2] = mxP; }
}


// ---------------------------------------------------------------------------------------------
// Parallel-in-time form of the two mean recursions that involve no `mom` call:
//   forward  (filter of sweeps >= 2, ihgp_ep_modulator_nmf.m:280-304 with fixed sites):
//            m_k = (A - K h A(1,:)) m_{k-1} + K ys          (or A m_{k-1} for clamped sites)
//   backward (smoother, :391):  m_k = MF_k + G_k (m_{k+1} - A MF_k)
// Both are affine maps x -> F x + g per diagonal block with F, g computable from stored arrays, so the
// sequence is cut into spans: compose (one thread per span and block), boundary (sequential over the
// spans, one thread per block), apply (replay inside the span, write outputs).
struct AffPar {
  int mode;        // 0 forward filter over k = 0 .. kend-1 ; 1 backward smoother over k = kend-1 .. 0
  int64_t kend;    // number of steps covered
  int L;           // span length
  int ns;          // spans
  double* spanbuf; // [B][ns][M][BS*BS + BS]  Phi + c   (BS = Shape::BS: 4, or 8 for blocks of 5 .. 8 states)
  double* bnd;     // [B][ns][M][BS]   value entering the span
  double* vprev;   // [B][M] (smoother: previous sweep's marginal variance at k = 0)
};

// coefficients of step k for block n: x_new = F x + g
template <int MODE, int BS = 4>
__device__ __forceinline__ void ihgp_coeffs(const Shape& sh, const Bufs& b, const IhgpTabs& tb, const double* tab,
                                            const double* A4, double hn, int n, int pb, int64_t k, int bs, int o,
                                            double* F, double* g, double& aux) {
  const int M = sh.M, NG = tb.NG;
  const int64_t T = sh.T;
  const size_t ix = ((size_t)pb * T + k) * M + n;
  if (MODE == 0) {
    const double tt = max0(b.ttau[ix]);
    if (tt == 0.0) {
#pragma unroll
      for (int e = 0; e < BS * BS; ++e) F[e] = A4[e];
#pragma unroll
      for (int i = 0; i < BS; ++i) g[i] = 0.0;
      aux = INFINITY;                       // R(n,k) = inf
    } else {
      const double Rk = b.R[ix];
      double hph, wc[BS];
      if (k > 0) {
        const double tprev = max0(b.ttau[ix - M]);
        const double Rprev = (tprev == 0.0) ? INFINITY : b.R[ix - M];
        const int idx = nearest_idx(tb, Rprev);
        hph = tab[itab_hph(sh, NG) + (size_t)n * NG + idx];
        const double* w = tab + itab_wcol(sh, NG) + ((size_t)n * NG + idx) * BS;
#pragma unroll
        for (int i = 0; i < BS; ++i) wc[i] = w[i];
      } else {
        hph = tab[itab_hph0(sh, NG) + n];
        const double* w = tab + itab_wcol0(sh, NG) + (size_t)n * BS;
#pragma unroll
        for (int i = 0; i < BS; ++i) wc[i] = w[i];
      }
      const double den = hph + Rk;
      const double ys = b.tnu[ix] / tt;
#pragma unroll
      for (int i = 0; i < BS; ++i) {
        const double Ki = wc[i] / den;
        g[i] = Ki * ys;
#pragma unroll
        for (int j = 0; j < BS; ++j) F[BS * i + j] = A4[BS * i + j] - Ki * hn * A4[j];
      }
      aux = Rk;
    }
  } else {
    const double Rk = b.R[ix];
    int idx = nearest_idx(tb, Rk);
    if (isinf(Rk)) idx = NG - 1;
#pragma unroll
    for (int e = 0; e < BS * BS; ++e) F[e] = tab[itab_g(sh, NG) + ((size_t)n * NG + idx) * BS * BS + e];
    aux = tab[itab_v(sh, NG) + (size_t)n * NG + idx];
    double mf[BS], amf[BS];
#pragma unroll
    for (int i = 0; i < BS; ++i) mf[i] = 0.0;
    const double* mfp = b.MF + ((size_t)pb * T + k) * sh.S + o;
#pragma unroll
    for (int i = 0; i < BS; ++i)
      if (i < bs) mf[i] = mfp[i];
#pragma unroll
    for (int i = 0; i < BS; ++i) {
      double a = 0.0;
#pragma unroll
      for (int l = 0; l < BS; ++l) a = fma(A4[BS * i + l], mf[l], a);
      amf[i] = a;
    }
#pragma unroll
    for (int i = 0; i < BS; ++i) {
      double a = mf[i];
#pragma unroll
      for (int l = 0; l < BS; ++l) a = fma(-F[BS * i + l], amf[l], a);
      g[i] = a;
    }
  }
}

// span j covers recursion steps [j*L, min((j+1)L, kend)) counted in recursion order; k = that index
// (forward) or kend-1-index (backward).
template <int MODE, int BS = 4>
__global__ void __launch_bounds__(256) ihgp_aff_compose_kernel(Shape sh, Bufs b, IhgpTabs tb, AffPar ap) {
  const int pb = blockIdx.y, M = sh.M;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= ap.ns * M) return;
  const int j = gid / M, n = gid - j * M;
  const double* mdl = b.model + (size_t)pb * mdl_size(sh);
  const double* tab = tb.base + (size_t)pb * itab_size(sh, tb.NG);
  double A4[BS * BS];
#pragma unroll
  for (int e = 0; e < BS * BS; ++e) A4[e] = mdl[mdl_A(sh) + (size_t)n * BS * BS + e];
  const double hn = mdl[mdl_h(sh) + n];
  const int bs = sh.bsz[n], o = sh.off[n];
  double Phi[BS * BS], c[BS];
#pragma unroll
  for (int e = 0; e < BS * BS; ++e) Phi[e] = ((e % BS) == (e / BS)) ? 1.0 : 0.0;
#pragma unroll
  for (int i = 0; i < BS; ++i) c[i] = 0.0;
  const int64_t s0 = (int64_t)j * ap.L, s1 = (s0 + ap.L < ap.kend) ? s0 + ap.L : ap.kend;
  for (int64_t s = s0; s < s1; ++s) {
    const int64_t k = (MODE == 0) ? s : (ap.kend - 1 - s);
    double F[BS * BS], g[BS], aux;
    ihgp_coeffs<MODE, BS>(sh, b, tb, tab, A4, hn, n, pb, k, bs, o, F, g, aux);
    double P2[BS * BS], c2[BS];
#pragma unroll
    for (int i = 0; i < BS; ++i) {
      double a = g[i];
#pragma unroll
      for (int l = 0; l < BS; ++l) a = fma(F[BS * i + l], c[l], a);
      c2[i] = a;
#pragma unroll
      for (int jj = 0; jj < BS; ++jj) {
        double q = 0.0;
#pragma unroll
        for (int l = 0; l < BS; ++l) q = fma(F[BS * i + l], Phi[BS * l + jj], q);
        P2[BS * i + jj] = q;
      }
    }
#pragma unroll
    for (int e = 0; e < BS * BS; ++e) Phi[e] = P2[e];
#pragma unroll
    for (int i = 0; i < BS; ++i) c[i] = c2[i];
  }
  double* out = ap.spanbuf + (((size_t)pb * ap.ns + j) * M + n) * (BS * BS + BS);
#pragma unroll
  for (int e = 0; e < BS * BS; ++e) out[e] = Phi[e];
#pragma unroll
  for (int i = 0; i < BS; ++i) out[BS * BS + i] = c[i];
}

template <int MODE, int BS = 4>
__global__ void __launch_bounds__(64) ihgp_aff_boundary_kernel(Shape sh, Bufs b, AffPar ap, int itt) {
  const int n = threadIdx.x, pb = blockIdx.x, M = sh.M, S = sh.S;
  const int64_t T = sh.T;
  if (n >= M) return;
  const int bs = sh.bsz[n], o = sh.off[n];
  double x[BS];
#pragma unroll
  for (int i = 0; i < BS; ++i) x[i] = 0.0;
  if (MODE == 0) {            // filter of sweep itt >= 2 starts from the smoothed mean at k = 0 (SURVEY C-22)
    if (itt > 1)
#pragma unroll
      for (int i = 0; i < BS; ++i) if (i < bs) x[i] = b.MS[(size_t)pb * T * S + o + i];
  } else {                    // smoother starts from the last filtered mean
#pragma unroll
    for (int i = 0; i < BS; ++i) if (i < bs) x[i] = b.MF[((size_t)pb * T + (T - 1)) * S + o + i];
  }
  for (int j = 0; j < ap.ns; ++j) {
    double* bj = ap.bnd + (((size_t)pb * ap.ns + j) * M + n) * BS;
    const double* sp = ap.spanbuf + (((size_t)pb * ap.ns + j) * M + n) * (BS * BS + BS);
    double y[BS];
#pragma unroll
    for (int i = 0; i < BS; ++i) {
      bj[i] = x[i];
      double a = sp[BS * BS + i];
#pragma unroll
      for (int l = 0; l < BS; ++l) a = fma(sp[BS * i + l], x[l], a);
      y[i] = a;
    }
#pragma unroll
    for (int i = 0; i < BS; ++i) x[i] = y[i];
  }
}

template <int MODE, int BS = 4>
__global__ void __launch_bounds__(256) ihgp_aff_apply_kernel(Shape sh, Bufs b, IhgpTabs tb, AffPar ap) {
  const int pb = blockIdx.y, M = sh.M, S = sh.S;
  const int64_t T = sh.T;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  double mxM = 0.0, mxP = 0.0;
  if (gid < ap.ns * M) {
    const int j = gid / M, n = gid - j * M;
    const double* mdl = b.model + (size_t)pb * mdl_size(sh);
    const double* tab = tb.base + (size_t)pb * itab_size(sh, tb.NG);
    double A4[BS * BS];
#pragma unroll
    for (int e = 0; e < BS * BS; ++e) A4[e] = mdl[mdl_A(sh) + (size_t)n * BS * BS + e];
    const double hn = mdl[mdl_h(sh) + n];
    const int bs = sh.bsz[n], o = sh.off[n];
    double x[BS];
    const double* bj = ap.bnd + (((size_t)pb * ap.ns + j) * M + n) * BS;
#pragma unroll
    for (int i = 0; i < BS; ++i) x[i] = bj[i];
    const int64_t s0 = (int64_t)j * ap.L, s1 = (s0 + ap.L < ap.kend) ? s0 + ap.L : ap.kend;
    for (int64_t s = s0; s < s1; ++s) {
      const int64_t k = (MODE == 0) ? s : (ap.kend - 1 - s);
      double F[BS * BS], g[BS], aux;
      ihgp_coeffs<MODE, BS>(sh, b, tb, tab, A4, hn, n, pb, k, bs, o, F, g, aux);
      double y[BS];
#pragma unroll
      for (int i = 0; i < BS; ++i) {
        double a = g[i];
#pragma unroll
        for (int l = 0; l < BS; ++l) a = fma(F[BS * i + l], x[l], a);
        y[i] = a;
      }
#pragma unroll
      for (int i = 0; i < BS; ++i) x[i] = y[i];
      const size_t ix = ((size_t)pb * T + k) * M + n;
      if (MODE == 0) {
        double* mfp = b.MF + ((size_t)pb * T + k) * S + o;
#pragma unroll
        for (int i = 0; i < BS; ++i)
          if (i < bs) mfp[i] = x[i];
        b.fm[ix] = hn * x[0];
        b.ttau[ix] = max0(b.ttau[ix]);     // the clamp of :274 is stored
        b.R[ix] = aux;                      // unchanged, or Inf for clamped sites (:287)
      } else {
        double* msp = b.MS + ((size_t)pb * T + k) * S + o;
#pragma unroll
        for (int i = 0; i < BS; ++i)
          if (i < bs) msp[i] = x[i];
        const double mnew = hn * x[0];
        mxM = fmax(mxM, fabs(b.sm[ix] - mnew));
        b.sm[ix] = mnew;
        b.sv[ix] = aux;
        if (k == 0) {
          mxP = fabs(ap.vprev[(size_t)pb * M + n] - aux);
          ap.vprev[(size_t)pb * M + n] = aux;
        }
      }
    }
  }
  if (MODE == 1) {
    mxM = wave_max(mxM);
    mxP = wave_max(mxP);
    if ((threadIdx.x & 63) == 0) {
      atomicMax(reinterpret_cast<unsigned long long*>(&b.red[(size_t)pb * 8 + 1]), (unsigned long long)__double_as_longlong(mxM));
      atomicMax(reinterpret_cast<unsigned long long*>(&b.red[(size_t)pb * 8 + 2]), (unsigned long long)__double_as_longlong(mxP));
    }
  }
}

}  // namespace nagp
