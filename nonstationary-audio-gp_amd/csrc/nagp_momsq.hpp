// nagp_momsq.hpp -- likModulatorPreCalcwn (matlab/experiments/likModulatorPreCalcwn.m:28-86: the likelihood of every paper
// experiment, train_model.m:55, noise_reduction_speech.m:41, source_sep_piano.m:94) in a staged form for the sequential ADF
// step and the site refresh, beside nagp_momsp.hpp (likModulatorNMFPower).
//
// The amplitudes are a_d(p) = sqrt(W_d . lk_p), lk_p = link(xn_p) (:44).  What collapses and what does not:
//   sum_d a_d^2 s2_d = t . lk_p,  t = W' s2_z                         linear in lk: N multiply-adds per sigma point   (:46)
//   sum_d a_d mu_d                                                    one square root per (sigma point, sub-band)     (:47)
//   d lZ / d mu_z(d)   = (1/Z) sum_p c1_p a_d(p)                      (:59-62)  -- a_d(p) again: kept in registers
//   d2 lZ / d mu_z(d)2 = -(.)^2 + (1/Z) sum_p c2_p a_d(p)^2           (:75-77)
//   modulators: sum_p c0_p xg_j(p), sum_p c0_p xg2_j(p)               (:70-80)  from the marginal sums of c0 (as the packed form of
//                                                                     nagp_momsp.hpp: static member lists)
// with c0 = w_p N(y; mu_p, sig2_p), c1 = c0 (y-mu_p)/sig2_p, c2 = c0 (((y-mu_p)/sig2_p)^2 - 1/sig2_p).
// W_d . lk_p is summed over ALL N components with non-negative terms (no "centre + deviation" form: the square root of a
// difference of nearly equal sums would lose digits).
//
// Lane layout of the square-root stage: a wave step takes four sigma points x 16 sub-bands (lane = 16 * point + sub-band), twice
// when D > 16 (sub-bands d and d + 16 on the same lane); the sum over the sub-bands is a 16-lane DPP sum; the amplitudes stay in
// the lane's registers until the weights c1, c2 exist, then sum_p c1 a and sum_p c2 a^2 accumulate per lane (fixed sub-band) and
// are added over the four point rows of the wave once per time step.
//
// Stages (W worker waves; a workgroup barrier between them):
//   A    link wave: lk, xg, xg2 tables | one worker: t = W' s2_z | every worker: mu_z of its sub-bands
//   S    every worker: lkp[p][j] = lk[j][code_p(j)] of its own points (LDS gather), then the square-root steps -> mu_p
//   1b   one lane per sigma point: sig2_p = sn2/alpha + t . lkp[p], Gaussian weight -> c0, c1, c2
//   S2   every worker: sum c1 a, sum c2 a^2 of its points; two workers: marginal sums of c0 -> g1, g2, Z
//   out  sub-band lane d: fixed-order sum of the workers' partials
#pragma once
#include "nagp_momsp.hpp"

namespace nagp {

constexpr int MSQ_NWK = 6;        // worker waves (role layout of ihgp_adf8sq_kernel)
constexpr int MSQ_NST = 14;       // four-point steps per worker wave: 6 * 14 * 4 = 336 sigma points
constexpr int MSQ_NP = 4 * MSQ_NWK * MSQ_NST;
constexpr int MSQ_NGA = 6;        // gather entries per lane: 4 * MSQ_NST * CD / 64 <= 6 for CD <= 6
constexpr int MSQ_MAXCD = 6;
constexpr int MSQ_MAXD = 32;      // sub-bands: two per lane of a 16-lane row

// square root of x >= 0 (NaN stays NaN): hardware reciprocal square root, one coupled Goldschmidt step and two corrections
// (the sequence the compiler emits for sqrt(), without its rescaling of arguments below 2^-767)
__device__ __forceinline__ double sqrt_pos(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  double d = fma(-g, g, x);
  g = fma(d, h, g);
  d = fma(-g, g, x);
  g = fma(d, h, g);
  return (x == 0.0 || x == __builtin_inf()) ? x : g;
}

// LDS workspace (offsets in doubles)
struct MsqLay { int lk, xg, xg2, t, lkp, sam, c0, c1, c2, marg, acc, part, scr, total; };
__host__ __device__ inline int msq_cdp(int CD) { return (CD + 1) & ~1; }      // row stride of lkp (16-byte rows)
__host__ __device__ inline MsqLay msq_layout(int CD) {
  MsqLay l;
  l.lk = 0; l.xg = MSP_TS; l.xg2 = 2 * MSP_TS;
  l.t = 3 * MSP_TS;                       // [8]
  l.lkp = l.t + 8;                        // [MSQ_NP][cdp]
  l.sam = l.lkp + MSQ_NP * msq_cdp(CD);   // [MSQ_NP]
  l.c0 = l.sam + MSQ_NP;                  // c0 | c1 | c2, stride MSQ_NP + 64 (the sums over c0 read 384 entries)
  l.c1 = l.c0 + MSQ_NP + 64; l.c2 = l.c1 + MSQ_NP + 64;
  l.marg = l.c2 + MSQ_NP + 64;            // [MSR_NMARG]
  l.acc = l.marg + MSR_NMARG;             // [2][32]: g1[8] | g2[8] | Z, one copy per serial wave
  l.part = l.acc + 64;                    // [MSQ_NWK][2][32]
  l.scr = l.part + MSQ_NWK * 64;          // [64] scratch: the stores of lanes that own nothing (stage S: one word per step)
  l.total = l.scr + 64;
  return l;
}
__host__ __device__ inline size_t msq_lds_doubles(int CD) { return (size_t)msq_layout(CD).total; }

// Register-resident state of a worker lane; everything here is computed once per kernel.
template <int CD>
struct MsqW {
  int wr;                                    // worker rank 0 .. MSQ_NWK-1
  int nst, st0;                              // this wave's steps: points 4*(st0 + s) + row
  int two;                                   // D > 16: the lane owns sub-bands d and d + 16 (wave-uniform)
  double wlo[CD], whi[CD];                   // W rows of the lane's sub-bands (zero rows beyond D)
  msp_rp mu_lo, mu_hi;                       // fmu of the two sub-bands (zero padding beyond D)
  msp_rp ga_src[MSQ_NGA]; msp_wp ga_dst[MSQ_NGA];      // gather lk[j][code] -> lkp[p][j] of this wave's points
  msp_rp s_lkp;                              // lkp row of the lane's point of step 0 (+ 4 * cdp per step)
  msp_wp s_sam;                              // sam of that point (lane & 15 == 0 writes)
  msp_rp s_c1;                               // c1 of that point (c2 at + MSQ_NP + 64)
  msp_wp s_part;                             // this wave's partial sums, + sub-band
  // stage 1b: lane = sigma point
  msp_rp p_lkp, p_sam; msp_wp p_c; double p_wn; bool p_ok; int p_any;
  // t = W' s2_z on worker 0: four lanes per component
  int t_on; msp_rp t_w, t_src; msp_wp t_out;
  // marginal sums (the last two workers), as MsrW
  msp_rp g_mem[MSR_NMEM]; msp_wp g_out;
  msp_rp h_marg, h_xg, h_xg2, h_c0p; msp_wp h_acc0, h_acc1, h_z0, h_z1; int h_nd, h_c0, h_nj, h_z;
};

// `wr`: rank of this wave among the W workers; `tl`: index of the thread among the worker threads (0 .. 64 W - 1)
template <int CD>
__device__ __forceinline__ void msq_setup_W(MsqW<CD>& x, const MomCfg& c, int c0code, const double* Wl /* LDS D x CD */, const double* fmu,
                                             const double* HPH, double* ws, int wr, int tl, double* wt /* LDS [D][CD] copy for stage A */) {
  const int lane = threadIdx.x & 63;
  const int nd = c.nd, D = c.D, npt = c.n_pts;
  const MsqLay l = msq_layout(CD);
  const int cdp = msq_cdp(CD);
  x.wr = wr;
  // ---- steps: contiguous ranges, the first (nstep mod W) waves one longer
  {
    const int nstep = (npt + 3) >> 2;
    const int base = nstep / MSQ_NWK, rem = nstep - base * MSQ_NWK;
    x.st0 = wr * base + (wr < rem ? wr : rem);
    x.nst = base + (wr < rem ? 1 : 0);
  }
  x.two = (D > 16) ? 1 : 0;
  const int d = lane & 15, row = lane >> 4;
#pragma unroll
  for (int j = 0; j < CD; ++j) {
    x.wlo[j] = (d < D) ? Wl[d * CD + j] : 0.0;
    x.whi[j] = (d + 16 < D) ? Wl[(d + 16) * CD + j] : 0.0;
  }
  x.mu_lo = (msp_rp)(fmu + d);               // fmu is zero padded to 68 entries; D <= 32 keeps d + 16 < 48 -- sites of the modulators
  x.mu_hi = (msp_rp)(fmu + d + 16);          // sit behind the sub-bands there and meet zero W rows
  // ---- gather: entry e = lane + 64 u of this wave's 4 * nst * CD (point, component) pairs
  {
    const int n_ent = 4 * x.nst * CD;
#pragma unroll
    for (int u = 0; u < MSQ_NGA; ++u) {
      const int e = lane + 64 * u;
      int p = 4 * x.st0 + e / CD; const int j = e % CD;
      const bool ok = e < n_ent && p < npt;
      if (!ok) p = 0;
      // beyond the points: lk of the centre (finite; the weights of those points are zero, their amplitudes never reach a sum)
      const int code = ok ? c.code[(size_t)p * CD + j] : c0code;
      x.ga_src[u] = (msp_rp)(ws + l.lk + j * nd + code);
      x.ga_dst[u] = (e < n_ent) ? (msp_wp)(ws + l.lkp + (4 * x.st0 + e / CD) * cdp + j) : (msp_wp)(ws + l.acc + 63);      // scratch word
    }
  }
  {
    const int p0 = 4 * x.st0 + row;
    x.s_lkp = (msp_rp)(ws + l.lkp + p0 * cdp);
    x.s_sam = (d == 0) ? (msp_wp)(ws + l.sam + p0) : (msp_wp)(ws + l.scr);        // (+ 4 per step: 4 * MSQ_NST <= 64)
    x.s_c1 = (msp_rp)(ws + l.c1 + p0);
    x.s_part = (msp_wp)(ws + l.part + wr * 64 + d);
  }
  // ---- stage 1b: worker thread tl = sigma point
  {
    int p = tl;
    const bool ok = p < npt;
    x.p_ok = ok; x.p_any = (__builtin_amdgcn_ballot_w64(ok) != 0) ? 1 : 0;
    if (!ok) p = 0;
    x.p_lkp = (msp_rp)(ws + l.lkp + p * cdp); x.p_sam = (msp_rp)(ws + l.sam + p);
    x.p_c = (msp_wp)(ws + l.c0 + p);
    x.p_wn = ok ? c.wn[p] : 0.0;
  }
  // ---- t_j = sum_d W_dj s2_d: worker 0, four lanes per component, sub-bands sub, sub + 4, ...
  {
    const int j = lane >> 2, sub = lane & 3;
    x.t_on = (wr == 0 && j < CD) ? 1 : 0;
    const int jj = (j < CD) ? j : 0;
    if (wr == 0)          // W transposed for stride-free reads: wt[q][lane] = W(sub + 4 q, j), zero beyond D
      for (int q = 0; q < 8; ++q) wt[q * 64 + lane] = (j < CD && sub + 4 * q < D) ? Wl[(sub + 4 * q) * CD + jj] : 0.0;
    x.t_w = (msp_rp)(wt + lane);
    x.t_src = (msp_rp)(HPH + sub);
    x.t_out = (x.t_on && sub == 0) ? (msp_wp)(ws + l.t + jj) : (msp_wp)(ws + l.acc + 61);
  }
  // ---- marginal sums of c0 on the last two workers (the packed form of nagp_momsp.hpp: lane = 4 * (local marginal) + quarter)
  {
    const msp_rp zero = (msp_rp)(ws + l.acc + 60);      // a word that stays zero
    const int jsplit = (CD + 1) / 2;
    const bool mw = wr >= MSQ_NWK - 2;
    const int jlo = (wr == MSQ_NWK - 2) ? 0 : jsplit, jhi = (wr == MSQ_NWK - 2) ? jsplit : CD;
    const int ml = lane >> 2, quarter = lane & 3;
    const int jj = jlo + ml / (nd - 1), cc = ml % (nd - 1);
    const bool valid = mw && jj < jhi;
    const int j = valid ? jj : 0;
    const int code = (cc < c0code) ? cc : cc + 1;
    x.g_out = (valid && quarter == 0) ? (msp_wp)(ws + l.marg + j * (nd - 1) + cc) : (msp_wp)(ws + l.acc + 59);
    const int jm = (jlo + lane < jhi) ? jlo + lane : 0;
    x.h_nd = nd; x.h_c0 = c0code; x.h_nj = mw ? jhi - jlo : 0;
    x.h_z = (wr == MSQ_NWK - 2) ? 1 : 0;
    x.h_marg = (msp_rp)(ws + l.marg + jm * (nd - 1));
    x.h_xg = (msp_rp)(ws + l.xg + jm * nd);
    x.h_xg2 = (msp_rp)(ws + l.xg2 + jm * nd);
    x.h_acc0 = (msp_wp)(ws + l.acc + jm); x.h_acc1 = (msp_wp)(ws + l.acc + 32 + jm);
    x.h_z0 = (msp_wp)(ws + l.acc + 16); x.h_z1 = (msp_wp)(ws + l.acc + 48);
    x.h_c0p = (msp_rp)(ws + l.c0 + lane);
    int pos = 0, cnt = 0;
#pragma unroll
    for (int k = 0; k < MSR_NMEM; ++k) {
      int found = -1;
      while (valid && pos < npt && found < 0) {
        if (c.code[(size_t)pos * CD + j] == code) {
          if ((cnt & 3) == quarter) found = pos;
          ++cnt;
        }
        ++pos;
      }
      x.g_mem[k] = (found >= 0) ? (msp_rp)(ws + l.c0 + found) : zero;
    }
  }
}

// zero entries of the tables and weights, the scratch words (every thread of the workgroup; NT threads)
__device__ __forceinline__ void msq_init(int CD, double* ws, int NT) {
  const MsqLay l = msq_layout(CD);
  for (int i = threadIdx.x; i < l.total; i += NT) ws[i] = 0.0;
}

// stage A, worker part: t = W' s2_z (after the barrier that publishes HPH); no barrier
template <int CD>
__device__ __forceinline__ void msq_stageA(const MsqW<CD>& x) {
  if (__builtin_amdgcn_readfirstlane(x.wr) != 0) return;
  double w[8], v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { w[q] = x.t_w[64 * q]; v[q] = x.t_src[4 * q]; }
  double a0 = 0.0, a1 = 0.0;
#pragma unroll
  for (int q = 0; q < 8; q += 2) { a0 = fma(w[q], v[q], a0); a1 = fma(w[q + 1], v[q + 1], a1); }
  double a = a0 + a1;
  a += dpp_mov<0xB1>(a);
  a += dpp_mov<0x4E>(a);
  *x.t_out = a;            // lanes other than the first of a component: scratch word
}

// stage S: gather, square roots, mu_p.  After the barrier behind the link tables; ends without a barrier.
// amp[2 * s], amp[2 * s + 1]: the amplitudes of the lane's two sub-bands at the point of step s (kept for stage S2)
template <int CD>
__device__ __forceinline__ void msq_stageS(const MsqW<CD>& x, double* amp /* [2 * MSQ_NST] */) {
  constexpr int cdp = (CD + 1) & ~1;
  {
    double g[MSQ_NGA];
#pragma unroll
    for (int u = 0; u < MSQ_NGA; ++u) g[u] = *x.ga_src[u];
#pragma unroll
    for (int u = 0; u < MSQ_NGA; ++u) *x.ga_dst[u] = g[u];
  }
  const double mlo = *x.mu_lo, mhi = *x.mu_hi;
  msp_wave_fence();                      // the gathered rows are this wave's own writes
  const int nst = __builtin_amdgcn_readfirstlane(x.nst);
  const bool two = __builtin_amdgcn_readfirstlane(x.two) != 0;
#pragma unroll
  for (int s = 0; s < MSQ_NST; ++s) {
    if (s < nst) {
      double lk[cdp];
#pragma unroll
      for (int j = 0; j < cdp; ++j) lk[j] = x.s_lkp[4 * cdp * s + j];
      double al = x.wlo[0] * lk[0], ah = x.whi[0] * lk[0];
#pragma unroll
      for (int j = 1; j < CD; ++j) { al = fma(x.wlo[j], lk[j], al); if (two) ah = fma(x.whi[j], lk[j], ah); }
      al = sqrt_pos(al);
      double sm = al * mlo;
      if (two) { ah = sqrt_pos(ah); sm = fma(ah, mhi, sm); } else ah = 0.0;
      amp[2 * s] = al; amp[2 * s + 1] = ah;
      sm = group_sum(sm, 16);
      x.s_sam[4 * s] = sm;               // lanes d != 0: scratch word
    }
  }
}

// stage 1b: one lane per sigma point.  After a barrier behind stage S; ends without a barrier.
template <int CD>
__device__ __forceinline__ void msq_stage1b(const MsqW<CD>& x, double sn2a, double y, const double* ws) {
  if (__builtin_amdgcn_readfirstlane(x.p_any) == 0) return;
  const MsqLay l = msq_layout(CD);
  double lk[CD], t[CD];
#pragma unroll
  for (int j = 0; j < CD; ++j) { lk[j] = x.p_lkp[j]; t[j] = ws[l.t + j]; }
  const double sam = *x.p_sam;
  double s0 = sn2a, s1 = 0.0;
#pragma unroll
  for (int j = 0; j < CD; ++j) { if (j & 1) s1 = fma(t[j], lk[j], s1); else s0 = fma(t[j], lk[j], s0); }
  double pdf, q, inv;
  gauss_terms(y, sam, s0 + s1, pdf, q, inv);
  const double w0 = x.p_wn * pdf;
  if (x.p_ok) {
    x.p_c[0] = w0;
    x.p_c[MSQ_NP + 64] = w0 * q;
    x.p_c[2 * (MSQ_NP + 64)] = w0 * (q * q - inv);
  }
}

// sum over the four 16-lane rows of the wave; every lane gets the sum of the lanes with its (lane & 15)
__device__ __forceinline__ double rows_sum(double v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// stage S2: sum_p c1 a_d(p), sum_p c2 a_d(p)^2 over this wave's points -> part[wave][0 / 1][sub-band].  After a barrier behind stage 1b.
template <int CD>
__device__ __forceinline__ void msq_stageS2(const MsqW<CD>& x, const double* amp) {
  const int nst = __builtin_amdgcn_readfirstlane(x.nst);
  double s1l = 0.0, s1h = 0.0, s2l = 0.0, s2h = 0.0;
#pragma unroll
  for (int s = 0; s < MSQ_NST; ++s) {
    if (s < nst) {
      const double c1 = x.s_c1[4 * s], c2 = x.s_c1[4 * s + MSQ_NP + 64];
      const double al = amp[2 * s], ah = amp[2 * s + 1];
      s1l = fma(c1, al, s1l); s1h = fma(c1, ah, s1h);
      s2l = fma(c2 * al, al, s2l); s2h = fma(c2 * ah, ah, s2h);
    }
  }
  s1l = rows_sum(s1l); s1h = rows_sum(s1h); s2l = rows_sum(s2l); s2h = rows_sum(s2h);
  if (((int)threadIdx.x & 63) < 16) {
    x.s_part[0] = s1l; x.s_part[16] = s1h; x.s_part[32] = s2l; x.s_part[48] = s2h;
  }
}

// marginal sums of c0, g1_j and g2_j of this wave's dimensions, Z (the last two workers; after the barrier behind stage 1b)
template <int CD>
__device__ __forceinline__ void msq_marginals(const MsqW<CD>& x) {
  const int lane = threadIdx.x & 63;
  const int nd = __builtin_amdgcn_readfirstlane(x.h_nd), c0 = __builtin_amdgcn_readfirstlane(x.h_c0);
  double mem[MSR_NMEM];
#pragma unroll
  for (int k = 0; k < MSR_NMEM; ++k) mem[k] = *x.g_mem[k];
  const double z0 = x.h_c0p[0] + x.h_c0p[64], z1 = x.h_c0p[128] + x.h_c0p[192], z2 = x.h_c0p[256] + x.h_c0p[320];
  const double bc = x.h_xg2[c0];
  double s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int k = 0; k < MSR_NMEM; k += 2) { s0 += mem[k]; s1 += mem[k + 1]; }
  double s_ = s0 + s1;
  s_ += dpp_mov<0xB1>(s_);
  s_ += dpp_mov<0x4E>(s_);
  *x.g_out = s_;
  const double zraw = wave_sum((z0 + z1) + z2);      // the weights beyond n_pts are zero
  msp_wave_fence();
  if (lane < __builtin_amdgcn_readfirstlane(x.h_nj)) {
    double g1 = 0.0, g2 = 0.0, ms = 0.0;
    for (int cc = 0; cc < nd - 1; ++cc) {
      const int code = (cc < c0) ? cc : cc + 1;
      const double mm = x.h_marg[cc];
      g1 = fma(mm, x.h_xg[code], g1); g2 = fma(mm, x.h_xg2[code], g2); ms += mm;
    }
    g2 = fma(zraw - ms, bc, g2);      // xg of the centre coordinate is zero
    x.h_acc0[0] = g1; x.h_acc0[8] = g2; x.h_acc1[0] = g1; x.h_acc1[8] = g2;
  }
  if (lane == 0 && __builtin_amdgcn_readfirstlane(x.h_z)) { *x.h_z0 = zraw; *x.h_z1 = zraw; }
}

// outputs of one site (after the barrier behind stage S2).  Sub-band d: fixed-order sums of the worker partials; modulator j: g1, g2.
// acc: this serial wave's copy of [g1: 8][g2: 8][Z]
template <int CD>
__device__ __forceinline__ void msq_outputs(msp_rp acc, msp_rp part /* + (d & 15) + 16 * (d >> 4) */, bool sub, int jmod, double pEP, double jitter,
                                            double& Z, double& d1, double& d2) {
  const double Zs = acc[16];
  Z = pEP * ((Zs > jitter) ? Zs : jitter);          // max(NaN, jitter) = jitter
  const double Zinv = pEP * rcp_nr(Z);
  double s1, s2;
  if (sub) {
    double a[MSQ_NWK], b[MSQ_NWK];
#pragma unroll
    for (int w = 0; w < MSQ_NWK; ++w) { a[w] = part[64 * w]; b[w] = part[64 * w + 32]; }
    s1 = a[0]; s2 = b[0];
#pragma unroll
    for (int w = 1; w < MSQ_NWK; ++w) { s1 += a[w]; s2 += b[w]; }
  } else {
    s1 = acc[jmod]; s2 = acc[8 + jmod];
  }
  d1 = Zinv * s1;
  d2 = fma(-d1, d1, Zinv * s2);
}

}  // namespace nagp
