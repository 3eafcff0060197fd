// nagp_ihgp.hpp -- HIP kernels of the infinite-horizon (steady-state) path.
//   ihgp_filter_kernel  forward filter with per-channel DARE table look-ups, mean-only recursion
//                       (matlab/ihgp_ep_modulator_nmf.m:233-310); one workgroup per problem; block n's
//                       state lives in the registers of thread n, the workgroup cooperates on `mom`.
//   ihgp_scan_kernel    backward mean recursion with looked-up steady-state smoother gains (:373-394)
// The EP refresh (:397-436) reuses ep_site_kernel (parallel over steps).
#pragma once
#include "nagp_kernels.hpp"

namespace nagp {

// Device look-up tables of ONE problem (all doubles), derived on the host from PPlist / PGlist:
//   hph[M][NG]      h_n^2 * PP(1,1)             (diag(H*PP*H'))
//   wcol[M][NG][4]  h_n * PP(:,1)               (W(ii,n) = PP(ii,ii) * H(n,ii)')
//   hph0[M], wcol0[M][4]  the same from Pinf (k = 1 of the reference uses PP = Pinf)
//   gtab[M][NG][16] smoother gain block G (row-major, zero padded)
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
__host__ __device__ inline size_t itab_hph0(const Shape& s, int NG) { return (size_t)s.M * NG * 5; }
__host__ __device__ inline size_t itab_wcol0(const Shape& s, int NG) { return (size_t)s.M * NG * 5 + s.M; }
__host__ __device__ inline size_t itab_g(const Shape& s, int NG) { return (size_t)s.M * NG * 5 + 5 * (size_t)s.M; }
__host__ __device__ inline size_t itab_v(const Shape& s, int NG) { return itab_g(s, NG) + (size_t)s.M * NG * 16; }
__host__ __device__ inline size_t itab_size(const Shape& s, int NG) { return itab_v(s, NG) + (size_t)s.M * NG; }

// [~,ind] = min(abs(r-R)): first minimiser; NaN / +-Inf distances everywhere -> index 0 (SURVEY C-4)
__device__ __forceinline__ int nearest_idx(const IhgpTabs& tb, double R) {
  if (!(R == R) || isinf(R)) return 0;
  int est = 0;
  if (R > 0.0) {
    const double f = (log10(R) - tb.lr0) * tb.inv_dlr;
    est = (f <= 0.0) ? 0 : ((f >= (double)(tb.NG - 1)) ? tb.NG - 1 : (int)(f + 0.5));
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
  double R_init;     // exp(lik) (or 0 for the constraints variant): initial content of R(:,k)
};

__host__ __device__ inline size_t ihgp_filter_lds_doubles(const Shape& s, const MomCfg& mc) {
  return LDS_INT_DOUBLES + (size_t)s.D * s.N + 6 * (size_t)s.M + 8 + mom_lds_doubles(mc);
}

__global__ void __launch_bounds__(512) ihgp_filter_kernel(Shape sh, Bufs b, MomCfg mc, IhgpTabs tb, IhgpPar ip) {
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
  double* ws = misc + 8 + 2 * M;
  for (int i = tid; i <= M; i += NT) ioff[i] = sh.off[i];
  for (int i = tid; i < M; i += NT) ibsz[i] = sh.bsz[i];
  for (int i = tid; i < sh.D * sh.N; i += NT) sW[i] = mdl[mdl_W(sh) + i];
  const double sn2 = mdl[mdl_sn2(sh)];
  mom_cache_tables(mc, ws);
  lds_barrier();

  // thread n < M owns block n
  const int n = tid;
  const bool act = n < M;
  double A4[16], mreg[4] = {0, 0, 0, 0};
  double hn = 0.0;
  int o = 0, bs = 0;
  if (act) {
    tile_load(A4, mdl + mdl_A(sh) + (size_t)n * 16);
    hn = mdl[mdl_h(sh) + n];
    o = ioff[n]; bs = ibsz[n];
    if (ip.itt > 1) {   // m is NOT reset between sweeps (SURVEY C-22): smoothed mean at k=0
      const double* ms0 = b.MS + (size_t)pb * T * S;
      for (int i = 0; i < bs; ++i) mreg[i] = ms0[o + i];
    }
  }
  const double* yv = b.y + (size_t)pb * T;
  double* g_tt = b.ttau + (size_t)pb * T * M;
  double* g_tn = b.tnu + (size_t)pb * T * M;
  double* g_R = b.R + (size_t)pb * T * M;
  double* g_lZ = b.lZ + (size_t)pb * T;
  double* g_MF = b.MF + (size_t)pb * T * S;
  double* g_fm = b.fm + (size_t)pb * T * M;
  double Rprev = 0.0;
  unsigned long long n_clamped = 0;

  double y_nx = yv[0], tt_nx = 0.0, tn_nx = 0.0, R_nx = 0.0;
  if (act) { tt_nx = g_tt[n]; tn_nx = g_tn[n]; R_nx = g_R[n]; }
  for (int64_t k = 0; k < T; ++k) {
    const double yk = y_nx, tt_k = tt_nx, tn_k = tn_nx, R_k = R_nx;
    if (k + 1 < T) {
      y_nx = yv[k + 1];
      if (act) { tt_nx = g_tt[(size_t)(k + 1) * M + n]; tn_nx = g_tn[(size_t)(k + 1) * M + n]; R_nx = g_R[(size_t)(k + 1) * M + n]; }
    }
    double hph = 0.0, wc[4] = {0, 0, 0, 0}, Am[4] = {0, 0, 0, 0}, fmun = 0.0;
    if (act) {
      if (k > 0) {
        const int idx = nearest_idx(tb, Rprev);
        hph = tab[itab_hph(sh, NG) + (size_t)n * NG + idx];
        const double* w = tab + itab_wcol(sh, NG) + ((size_t)n * NG + idx) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) wc[i] = w[i];
      } else {
        hph = tab[itab_hph0(sh, NG) + n];
        const double* w = tab + itab_wcol0(sh, NG) + (size_t)n * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) wc[i] = w[i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double a = 0.0;
#pragma unroll
        for (int l = 0; l < 4; ++l) a = fma(A4[4 * i + l], mreg[l], a);
        Am[i] = a;
      }
      fmun = hn * Am[0];
      fmu[n] = fmun; HPH[n] = hph;
    }
    const bool do_mom = ip.mom_all || (k == T - 1);
    double tnew = 0.0, nnew = 0.0, Rn = 0.0;
    if (do_mom) {
      lds_barrier();
      mom_eval(mc, sW, sn2, 1.0, yk, fmu, HPH, ws, &misc[0], dl, d2l);
      if (act) {
        const double d1 = dl[n], d2 = d2l[n];
        const double t_old = tt_k, n_old = tn_k;
        tnew = (1.0 - ip.ep_damp) * t_old + ip.ep_damp * (-d2 / (1.0 + d2 * hph));
        nnew = (1.0 - ip.ep_damp) * n_old + ip.ep_damp * ((d1 - fmun * d2) / (1.0 + d2 * hph));
        Rn = 1.0 / tnew;                      // before the clamp (:269)
      }
      if (tid == 0) g_lZ[k] = misc[0];
    } else if (act) {
      tnew = tt_k; nnew = tn_k;
      Rn = R_k;
    }
    if (act) {
      if (!(tnew > 0.0)) ++n_clamped;
      tnew = max0(tnew);                       // :274 (NaN -> 0, C-3)
      const double ys = nnew / tnew;
      if (tnew == 0.0) {
        Rn = INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) mreg[i] = Am[i];
      } else {
        const double den = hph + Rn;
#pragma unroll
        for (int i = 0; i < 4; ++i) mreg[i] = Am[i] + (wc[i] / den) * (ys - fmun);
      }
      g_tt[(size_t)k * M + n] = tnew; g_tn[(size_t)k * M + n] = nnew; g_R[(size_t)k * M + n] = Rn;
      for (int i = 0; i < bs; ++i) g_MF[(size_t)k * S + o + i] = mreg[i];
      g_fm[(size_t)k * M + n] = hn * mreg[0];
      Rprev = Rn;
    }
    if (do_mom) lds_barrier();   // fmu/HPH/dl reuse
  }
  if (act && n_clamped) atomicAdd(&b.counters[(size_t)pb * 4 + 1], n_clamped);
}

// Backward mean recursion: m <- MF_k + G (m - A MF_k) with (P,G) looked up from R(:,k)
// (Inf -> last grid row, ihgp_ep_modulator_nmf.m:379-380).  One wave per problem, thread n = block n.
__global__ void __launch_bounds__(64) ihgp_scan_kernel(Shape sh, Bufs b, IhgpTabs tb, double* vprev /* [B][M] */) {
  const int n = threadIdx.x, pb = blockIdx.x;
  const int S = sh.S, M = sh.M, NG = tb.NG;
  const int64_t T = sh.T;
  const double* mdl = b.model + (size_t)pb * mdl_size(sh);
  const double* tab = tb.base + (size_t)pb * itab_size(sh, NG);
  double mxM = 0.0, mxP = 0.0;
  if (n < M) {
    double A4[16], mreg[4] = {0, 0, 0, 0};
    tile_load(A4, mdl + mdl_A(sh) + (size_t)n * 16);
    const double hn = mdl[mdl_h(sh) + n];
    const int o = sh.off[n], bs = sh.bsz[n];
    const double* g_MF = b.MF + (size_t)pb * T * S;
    double* g_MS = b.MS + (size_t)pb * T * S;
    const double* g_R = b.R + (size_t)pb * T * M;
    double* g_sm = b.sm + (size_t)pb * T * M;
    double* g_sv = b.sv + (size_t)pb * T * M;
    for (int i = 0; i < bs; ++i) { mreg[i] = g_MF[(size_t)(T - 1) * S + o + i]; g_MS[(size_t)(T - 1) * S + o + i] = mreg[i]; }
    double vlast = 0.0;   // P = zeros(size(A)) before the loop (:364)
    for (int64_t k = T - 2; k >= 0; --k) {
      const double Rk = g_R[(size_t)k * M + n];
      int idx = nearest_idx(tb, Rk);
      if (isinf(Rk)) idx = NG - 1;
      double G4[16], mf[4] = {0, 0, 0, 0}, d[4];
      tile_load(G4, tab + itab_g(sh, NG) + ((size_t)n * NG + idx) * 16);
      vlast = tab[itab_v(sh, NG) + (size_t)n * NG + idx];
      for (int i = 0; i < bs; ++i) mf[i] = g_MF[(size_t)k * S + o + i];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double a = mreg[i];
#pragma unroll
        for (int l = 0; l < 4; ++l) a = fma(-A4[4 * i + l], mf[l], a);
        d[i] = a;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double a = mf[i];
#pragma unroll
        for (int l = 0; l < 4; ++l) a = fma(G4[4 * i + l], d[l], a);
        mreg[i] = a;
      }
      for (int i = 0; i < bs; ++i) g_MS[(size_t)k * S + o + i] = mreg[i];
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

}  // namespace nagp
