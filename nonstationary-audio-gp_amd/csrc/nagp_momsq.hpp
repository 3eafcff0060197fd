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
// when D > 16 (sub-bands d and d + 16 on the same lane).  Four steps form a TILE: the 16 x 16 block of arguments W_d . lk_p is one
// product on v_mfma_f64_16x16x4 (A = the points' link values, B = W, eight components in two k-steps) whose accumulator layout
// (lane = column + 16 kq, register r <-> row 4 r + kq) hands register r the argument of step r at this lane's (point row, sub-band);
// eight independent square-root chains per tile; the sums over the sub-bands of the four steps are ONE transposing 16-lane DPP
// reduction (27 instructions instead of 4 x 12).  The amplitudes stay in the lane's registers until the weights c1, c2 exist, then
// sum_p c1 a and sum_p c2 a^2 accumulate per lane (fixed sub-band) and are added over the four point rows of the wave once per
// time step.  Every wave owns 64 ROWS of the per-point arrays (its points first, zero padding behind them): a step beyond the wave's
// range multiplies zero rows -- no branches, no selects.
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

// Worker waves of the role layout (ihgp_adf8sq_kernel).  A lone wave issues an FP64 instruction every ~8 cycles whatever its
// neighbours do and a SIMD takes one every 4, so the stage is priced in instructions per wave (profiles/r04_sqrt_stages.txt).
// Layout configurations: NTW waves that own tiles, NTL tiles of four steps each.
//   MsqRole: the role layout of ihgp_adf8sq_kernel -- six worker waves x four tiles beside two serial waves (512 threads)
//   MsqFlat: the 256-thread kernels (ADF launches of gf_filter_kernel, ep_site_sq_kernel) -- four waves x five tiles, every wave in every stage
template <int NTW_, int NTL_>
struct MsqC {
  static constexpr int NTW = NTW_, NTL = NTL_;
  static constexpr int NST = 4 * NTL_;            // four-point steps per tile wave
  static constexpr int RW = 4 * NST;              // rows (sigma points + padding) per tile wave
  static constexpr int NP = NTW_ * RW;            // rows in all
  static constexpr int NGA = (RW * 6 + 63) / 64;  // gather entries per lane (CD <= 6)
  static constexpr int NPS = (NTW_ == 4) ? 2 : 1; // sigma points per lane of stage 1b (flat: points 256 .. 319 on the last wave)
};
typedef MsqC<6, 4> MsqRole;
typedef MsqC<4, 5> MsqFlat;
constexpr int MSQ_NWK = MsqRole::NTW;
constexpr int MSQ_NT = 64 * (MSQ_NWK + 2);                // threads of the role kernel: two serial waves + the workers
constexpr int MSQ_MAXCD = 6;
constexpr int MSQ_MAXD = 32;      // sub-bands: two per lane of a 16-lane row
constexpr int MSQ_MAXPTS = 320;     // 80 steps = 20 tiles dealt 3 : 3 : 4 : 4 : 3 : 3

// square root of x >= 0 (NaN stays NaN, 0 -> 0): hardware reciprocal square root (2^-23), one Goldschmidt step (2^-45) and one
// correction -- the compiler's sqrt() sequence without the update of the slope, the second correction and the rescaling of arguments
// below 2^-767.
// (x = +inf gives NaN: an overflowed link value, which nothing downstream survives either.)
__device__ __forceinline__ double sqrt_pos(double x) {
  const double y = fmin(__builtin_amdgcn_rsq(x), 1e150);      // x = 0: rsq = inf -> a finite factor, every product below is 0
  double g = x * y;
  const double h = 0.5 * y;                                    // 1 / (2 sqrt x) to 2^-23: the slope of both corrections
  g = fma(fma(-h, g, 0.5), g, g);                              // Goldschmidt step: 2^-45
  return fma(fma(-g, g, x), h, g);                             // correction: error 2^-45 * 2^-23 beyond the rounding
}

// LDS workspace (offsets in doubles).  lkp, sam, c0, c1, c2 are indexed by ROW: tile wave w owns rows RW w .. RW w + RW - 1.
struct MsqLay { int lk, xg, xg2, t, lkp, sam, c0, c1, c2, marg, acc, part, scr, total; };
template <class C>
__host__ __device__ inline MsqLay msq_layout(int CD) {
  MsqLay l;
  (void)CD;
  l.lk = 0; l.xg = MSP_TS; l.xg2 = 2 * MSP_TS;
  l.t = 3 * MSP_TS;                       // [8]
  l.lkp = l.t + 8;                        // [NP][8]: link values of the row's point, components 6, 7 zero
  l.sam = l.lkp + C::NP * 8;              // [NP]
  l.c0 = l.sam + C::NP;                   // c0 | c1 | c2
  l.c1 = l.c0 + C::NP; l.c2 = l.c1 + C::NP;
  l.marg = l.c2 + C::NP;                  // [MSR_NMARG]
  l.acc = l.marg + MSR_NMARG;             // [2][32]: g1[8] | g2[8] | Z, two copies; words 56 .. 63: scratch / zero
  l.part = l.acc + 64;                    // [NTW][2][32]
  l.scr = l.part + C::NTW * 64;           // [64 + 16 NTL] scratch: one word per lane (+ 16 per tile: the stores of stage S of the lanes that own no step)
  l.total = l.scr + 64 + 16 * C::NTL;
  return l;
}
template <class C>
__host__ __device__ inline size_t msq_lds_doubles(int CD) { return (size_t)msq_layout<C>(CD).total; }

// Steps of tile wave w: whole tiles of four steps, contiguous ranges.  Role layout: waves w and w + 4 of the workgroup share a SIMD --
// workers 0, 4 and 1, 5 are two busy waves on theirs (the SIMD's FP64 issue -- one instruction per 4 cycles, 64 per MFMA -- is the limit),
// workers 2, 3 sit beside the serial waves, which idle through the stage (a lone wave issues every 8 cycles).  Measured with equal
// shares: 6 000 cycles on workers 0 .. 3, 8 000 - 9 100 on workers 4, 5 (profiles/r04_sqrt_stages.txt); hence tiles dealt 3 : 3 : 4 : 4 : 3 : 3.
// Flat layout: equal shares.
template <class C>
__host__ __device__ inline void msq_steps(int npt, int w, int& st0, int& nst) {
  const int nstep = (npt + 3) >> 2, ntile = (nstep + 3) >> 2;
  const int wt6[6] = {3, 3, 4, 4, 3, 3};
  int tl[C::NTW];
  for (int v = 0; v < C::NTW; ++v) tl[v] = 0;
  for (int t = 0; t < ntile; ++t) {          // the next tile goes to the wave with the smallest (tiles + 1) / weight
    int best = -1;
    for (int v = 0; v < C::NTW; ++v) {
      if (tl[v] >= C::NTL) continue;
      const int wv = (C::NTW == 6) ? wt6[v % 6] : 1, wb = (best < 0) ? 1 : ((C::NTW == 6) ? wt6[best % 6] : 1);
      if (best < 0 || (tl[v] + 1) * wb < (tl[best] + 1) * wv) best = v;
    }
    if (best < 0) break;                       // (more tiles than NTW * NTL: the host refuses such rules)
    ++tl[best];
  }
  int t0 = 0;
  for (int v = 0; v < w; ++v) t0 += tl[v];
  st0 = 4 * t0;
  nst = 4 * tl[w];
  if (st0 + nst > nstep) nst = (nstep > st0) ? nstep - st0 : 0;
}
template <class C>
__host__ __device__ inline int msq_row(int npt, int p) {
  const int step = p >> 2;
  for (int w = 0; w < C::NTW; ++w) {
    int st0, nst;
    msq_steps<C>(npt, w, st0, nst);
    if (step < st0 + nst) return C::RW * w + 4 * (step - st0) + (p & 3);
  }
  return 0;
}

// Register-resident state of a lane of a tile wave; everything here is computed once per kernel.
template <int CD, class C = MsqRole>
struct MsqW {
  typedef C Cfg;
  int wr;                                    // tile-wave rank 0 .. NTW-1
  int ntl;                                   // tiles this wave has steps in (wave-uniform)
  int two;                                   // D > 16: the lane owns sub-bands d and d + 16 (wave-uniform)
  double b0lo, b1lo, b0hi, b1hi;             // B operands: W(d, kq), W(d, 4 + kq) of the lane's two sub-bands (zero beyond D / CD)
  msp_rp mu_lo, mu_hi;                       // fmu of the two sub-bands (zero padding beyond D)
  msp_rp ga_src[C::NGA]; msp_wp ga_dst[C::NGA];      // gather lk[j][code] -> lkp[row][j] of this wave's points
  msp_rp a_lkp;                              // A operand of tile 0, k-step 0: lkp[RW w + (lane & 15)][lane >> 4] (+ 4: k-step 1, + 128: next tile)
  msp_wp s_sam;                              // sam of the step this lane writes for tile 0 (+ 16 per tile); lanes with (lane & 15) >= 4: scratch
  msp_rp s_c1;                               // c1 of the row of step 0 at this lane's point row (+ 4 per step; c2 at + NP)
  msp_wp s_part;                             // this wave's partial sums, + sub-band
  // stage 1b: lane = sigma point (NPS slots)
  msp_rp p_lkp[C::NPS], p_sam[C::NPS], p_t; msp_wp p_c[C::NPS]; double p_wn[C::NPS]; bool p_ok[C::NPS]; int p_any[C::NPS];
  // t = W' s2_z on one wave: four lanes per component
  int t_on; msp_rp t_w, t_src; msp_wp t_out;
};
// marginal sums of c0 (two waves, each half of the dimensions; the packed form of nagp_momsp.hpp: lane = 4 * (local marginal) + quarter)
struct MsqM {
  msp_rp g_mem[MSR_NMEM]; msp_wp g_out;
  msp_rp h_marg, h_xg, h_xg2, h_c0p; msp_wp h_acc0, h_acc1, h_z0, h_z1; int h_nd, h_c0, h_nj, h_z;
};

// `wr`: rank of this wave among the tile waves; `tl`: index of the thread among the threads that take sigma points in stage 1b
// (0 .. 64 NTW - 1; flat layout: slot 1 = points 256 .. on the last wave); `twave`: the wave that forms t = W' s2_z
template <int CD, class C>
__device__ __forceinline__ void msq_setup_W(MsqW<CD, C>& x, const MomCfg& c, int c0code, const double* Wl /* LDS D x CD */, const double* fmu,
                                             const double* HPH, double* ws, int wr, int tl, double* wt /* LDS [8][64]: W transposed for stage A */, int twave = 0) {
  const int lane = threadIdx.x & 63;
  const int nd = c.nd, D = c.D, npt = c.n_pts;
  const MsqLay l = msq_layout<C>(CD);
  x.wr = wr;
  int st0, nst;
  msq_steps<C>(npt, wr, st0, nst);
  x.ntl = (nst + 3) >> 2;
  x.two = (D > 16) ? 1 : 0;
  const int d = lane & 15, kq = lane >> 4;
  x.b0lo = (d < D && kq < CD) ? Wl[d * CD + kq] : 0.0;
  x.b1lo = (d < D && 4 + kq < CD) ? Wl[d * CD + 4 + kq] : 0.0;
  x.b0hi = (d + 16 < D && kq < CD) ? Wl[(d + 16) * CD + kq] : 0.0;
  x.b1hi = (d + 16 < D && 4 + kq < CD) ? Wl[(d + 16) * CD + 4 + kq] : 0.0;
  x.mu_lo = (msp_rp)(fmu + d);               // fmu is zero padded to 68 entries; D <= 32 keeps d + 16 < 48 -- sites of the modulators
  x.mu_hi = (msp_rp)(fmu + d + 16);          // sit behind the sub-bands there and meet zero W rows
  // ---- gather: entry e = lane + 64 u of this wave's 4 * nst * CD (point, component) pairs
  {
    const int n_ent = 4 * nst * CD;
#pragma unroll
    for (int u = 0; u < C::NGA; ++u) {
      const int e = lane + 64 * u;
      int p = 4 * st0 + e / CD; const int j = e % CD;
      const bool ok = e < n_ent && p < npt;
      if (!ok) p = 0;
      // beyond the points (the unused rows of the last step): lk of the centre -- finite; the weights of those rows stay zero
      const int code = ok ? c.code[(size_t)p * CD + j] : c0code;
      x.ga_src[u] = (msp_rp)(ws + l.lk + j * nd + code);
      x.ga_dst[u] = (e < n_ent) ? (msp_wp)(ws + l.lkp + (C::RW * wr + e / CD) * 8 + j) : (msp_wp)(ws + l.acc + 63);      // scratch word
    }
  }
  x.a_lkp = (msp_rp)(ws + l.lkp + (C::RW * wr + d) * 8 + kq);
  x.s_sam = (d < 4) ? (msp_wp)(ws + l.sam + C::RW * wr + 4 * d + kq) : (msp_wp)(ws + l.scr + lane);
  x.s_c1 = (msp_rp)(ws + l.c1 + C::RW * wr + kq);
  x.s_part = (msp_wp)(ws + l.part + wr * 64 + d);
  // ---- stage 1b: thread tl = sigma point (slot u: + 64 NTW u, on the last wave only)
  x.p_t = (msp_rp)(ws + l.t);
#pragma unroll
  for (int u = 0; u < C::NPS; ++u) {
    int p = (u == 0) ? tl : (64 * C::NTW * u + (tl - 64 * (C::NTW - 1)));
    const bool ok = (u == 0) ? (p >= 0 && p < npt) : (tl >= 64 * (C::NTW - 1) && p < npt);
    x.p_ok[u] = ok; x.p_any[u] = (__builtin_amdgcn_ballot_w64(ok) != 0) ? 1 : 0;
    if (!ok) p = 0;
    const int q = msq_row<C>(npt, p);
    x.p_lkp[u] = (msp_rp)(ws + l.lkp + q * 8); x.p_sam[u] = (msp_rp)(ws + l.sam + q);
    x.p_c[u] = (msp_wp)(ws + l.c0 + q);
    x.p_wn[u] = ok ? c.wn[p] : 0.0;
  }
  // ---- t_j = sum_d W_dj s2_d: one wave, four lanes per component, sub-bands sub, sub + 4, ...
  {
    const int j = lane >> 2, sub = lane & 3;
    x.t_on = (wr == twave && j < CD) ? 1 : 0;
    const int jj = (j < CD) ? j : 0;
    if (wr == twave)      // W transposed for stride-free reads: wt[q][lane] = W(sub + 4 q, j), zero beyond D
      for (int q = 0; q < 8; ++q) wt[q * 64 + lane] = (j < CD && sub + 4 * q < D) ? Wl[(sub + 4 * q) * CD + jj] : 0.0;
    x.t_w = (msp_rp)(wt + lane);
    x.t_src = (msp_rp)(HPH + sub);
    x.t_out = (x.t_on && sub == 0) ? (msp_wp)(ws + l.t + jj) : (msp_wp)(ws + l.acc + 61);
  }
}

// which = 0: the dimensions below (CD + 1) / 2 and the sum of all c0 (Z), which = 1: the other dimensions
template <int CD, class C>
__device__ __forceinline__ void msq_setup_M(MsqM& x, const MomCfg& c, int c0code, double* ws, int which) {
  const int lane = threadIdx.x & 63;
  const int nd = c.nd, npt = c.n_pts;
  const MsqLay l = msq_layout<C>(CD);
  const msp_rp zero = (msp_rp)(ws + l.acc + 60);      // a word that stays zero
  const int jsplit = (CD + 1) / 2;
  const int jlo = (which == 0) ? 0 : jsplit, jhi = (which == 0) ? jsplit : CD;
  const int ml = lane >> 2, quarter = lane & 3;
  const int jj = jlo + ml / (nd - 1), cc = ml % (nd - 1);
  const bool valid = (which == 0 || which == 1) && jj < jhi;
  const int j = valid ? jj : 0;
  const int code = (cc < c0code) ? cc : cc + 1;
  x.g_out = (valid && quarter == 0) ? (msp_wp)(ws + l.marg + j * (nd - 1) + cc) : (msp_wp)(ws + l.acc + 59 - (which & 1));
  const int jm = (jlo + lane < jhi) ? jlo + lane : 0;
  x.h_nd = nd; x.h_c0 = c0code; x.h_nj = jhi - jlo;
  x.h_z = (which == 0) ? 1 : 0;
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
    x.g_mem[k] = (found >= 0) ? (msp_rp)(ws + l.c0 + msq_row<C>(npt, found)) : zero;
  }
}

// zero entries of the tables and weights, the scratch words (every thread of the workgroup; NT threads)
template <class C>
__device__ __forceinline__ void msq_init(int CD, double* ws, int NT) {
  const MsqLay l = msq_layout<C>(CD);
  for (int i = threadIdx.x; i < l.total; i += NT) ws[i] = 0.0;
}

// stage A, tile-wave part: t = W' s2_z on the wave chosen at set-up (after the barrier that publishes HPH); no barrier
template <int CD, class C>
__device__ __forceinline__ void msq_stageA(const MsqW<CD, C>& x) {
  if (__builtin_amdgcn_ballot_w64(x.t_on != 0) == 0) return;
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

// sums over the 16 lanes of a row of FOUR values at once: lane c of the row ends with the sum of v[c & 3]
// (two transposing levels -- keep one value of a pair, hand the other to the neighbour -- then two plain levels over the quads)
__device__ __forceinline__ double row_sum4(const double v[4]) {
  const int c = threadIdx.x;
  const bool b0 = (c & 1) != 0, b1 = (c & 2) != 0;
  const double w01 = (b0 ? v[1] : v[0]) + dpp_mov<0xB1>(b0 ? v[0] : v[1]);
  const double w23 = (b0 ? v[3] : v[2]) + dpp_mov<0xB1>(b0 ? v[2] : v[3]);
  double z = (b1 ? w23 : w01) + dpp_mov<0x4E>(b1 ? w01 : w23);
  z += dpp_mov<0x124>(z);      // row_ror:4
  z += dpp_mov<0x128>(z);      // row_ror:8
  return z;
}

// stage S: gather, arguments (MFMA), square roots, mu_p.  After the barrier behind the link tables; ends without a barrier.
// amp[8 * T + 2 * r], amp[.. + 1]: the amplitudes of the lane's two sub-bands at the point of step 4 T + r (kept for stage S2)
template <int CD, class C, bool TWO, int NTLX>      // NTLX tiles, straight-line (a branch per tile costs a register copy per amplitude)
__device__ __forceinline__ void msq_stageS_tiles(const MsqW<CD, C>& x, double* amp, double mlo, double mhi) {
#pragma unroll
  for (int T = 0; T < NTLX; ++T) {
    const double a0 = x.a_lkp[128 * T], a1 = x.a_lkp[128 * T + 4];      // (components beyond CD: zero columns of lkp, zero B operands)
    const v4d z = {0.0, 0.0, 0.0, 0.0};
    v4d lo = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, x.b0lo, z, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, x.b1lo, lo, 0, 0, 0);
    v4d hi = z;
    if (TWO) {
      hi = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, x.b0hi, z, 0, 0, 0);
      hi = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, x.b1hi, hi, 0, 0, 0);
    }
    double sm[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double al = sqrt_pos(lo[r]);
      sm[r] = al * mlo;
      amp[8 * T + 2 * r] = al;
      if (TWO) { const double ah = sqrt_pos(hi[r]); sm[r] = fma(ah, mhi, sm[r]); amp[8 * T + 2 * r + 1] = ah; }
    }
    x.s_sam[16 * T] = row_sum4(sm);      // lane c < 4 of a row: step 4 T + c; the other lanes: scratch words
  }
}
template <int CD, class C, bool TWO>
__device__ __forceinline__ void msq_stageS_two(const MsqW<CD, C>& x, double* amp, double mlo, double mhi) {
  switch (__builtin_amdgcn_readfirstlane(x.ntl)) {
    case 0: break;
    case 1: msq_stageS_tiles<CD, C, TWO, 1>(x, amp, mlo, mhi); break;
    case 2: msq_stageS_tiles<CD, C, TWO, 2>(x, amp, mlo, mhi); break;
    case 3: msq_stageS_tiles<CD, C, TWO, 3>(x, amp, mlo, mhi); break;
    case 4: msq_stageS_tiles<CD, C, TWO, 4>(x, amp, mlo, mhi); break;
    default: if constexpr (C::NTL >= 5) msq_stageS_tiles<CD, C, TWO, 5>(x, amp, mlo, mhi); break;
  }
}
template <int CD, class C>
__device__ __forceinline__ void msq_stageS(const MsqW<CD, C>& x, double* amp /* [2 * C::NST], zero beyond the wave's tiles */) {
  {
    double g[C::NGA];
#pragma unroll
    for (int u = 0; u < C::NGA; ++u) g[u] = *x.ga_src[u];
#pragma unroll
    for (int u = 0; u < C::NGA; ++u) *x.ga_dst[u] = g[u];
  }
  const double mlo = *x.mu_lo, mhi = *x.mu_hi;
  msp_wave_fence();                      // the gathered rows are this wave's own writes
  if (__builtin_amdgcn_readfirstlane(x.two)) msq_stageS_two<CD, C, true>(x, amp, mlo, mhi);
  else msq_stageS_two<CD, C, false>(x, amp, mlo, mhi);
}

// stage 1b: one lane per sigma point.  After a barrier behind stage S; ends without a barrier.
template <int CD, class C>
__device__ __forceinline__ void msq_stage1b(const MsqW<CD, C>& x, double sn2a, double y) {
#pragma unroll
  for (int u = 0; u < C::NPS; ++u) {
    if (__builtin_amdgcn_readfirstlane(x.p_any[u]) == 0) continue;
    double lk[CD], t[CD];
#pragma unroll
    for (int j = 0; j < CD; ++j) { lk[j] = x.p_lkp[u][j]; t[j] = x.p_t[j]; }
    const double sam = *x.p_sam[u];
    double s0 = sn2a, s1 = 0.0;
#pragma unroll
    for (int j = 0; j < CD; ++j) { if (j & 1) s1 = fma(t[j], lk[j], s1); else s0 = fma(t[j], lk[j], s0); }
    double pdf, q, inv;
    gauss_terms(y, sam, s0 + s1, pdf, q, inv);
    const double w0 = x.p_wn[u] * pdf;
    if (x.p_ok[u]) {
      x.p_c[u][0] = w0;
      x.p_c[u][C::NP] = w0 * q;
      x.p_c[u][2 * C::NP] = w0 * (q * q - inv);
    }
  }
}

// sum over the four 16-lane rows of the wave; every lane gets the sum of the lanes with its (lane & 15)
__device__ __forceinline__ double rows_sum(double v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// stage S2: sum_p c1 a_d(p), sum_p c2 a_d(p)^2 over this wave's points -> part[wave][0 / 1][sub-band].  After a barrier behind stage 1b.
// All tiles without a branch (every weight of a tile first: one LDS round trip): the amplitudes beyond the wave's tiles are zero, the
// weights of padding rows as well.
template <int CD, class C>
__device__ __forceinline__ void msq_stageS2(const MsqW<CD, C>& x, const double* amp) {
  double s1l[2] = {0.0, 0.0}, s1h[2] = {0.0, 0.0}, s2l[2] = {0.0, 0.0}, s2h[2] = {0.0, 0.0};
#pragma unroll
  for (int T = 0; T < C::NTL; ++T) {      // (tiles beyond the wave's: zero amplitudes, weights of padding rows)
    double c1[4], c2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { c1[r] = x.s_c1[16 * T + 4 * r]; c2[r] = x.s_c1[16 * T + 4 * r + C::NP]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(c1[r]), "+v"(c2[r]));      // (all reads of the tile before its arithmetic)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double al = amp[8 * T + 2 * r], ah = amp[8 * T + 2 * r + 1];
      s1l[r & 1] = fma(c1[r], al, s1l[r & 1]); s1h[r & 1] = fma(c1[r], ah, s1h[r & 1]);
      s2l[r & 1] = fma(c2[r] * al, al, s2l[r & 1]); s2h[r & 1] = fma(c2[r] * ah, ah, s2h[r & 1]);
    }
  }
  const double a = rows_sum(s1l[0] + s1l[1]), b = rows_sum(s1h[0] + s1h[1]), c = rows_sum(s2l[0] + s2l[1]), d = rows_sum(s2h[0] + s2h[1]);
  if (((int)threadIdx.x & 63) < 16) {
    x.s_part[0] = a; x.s_part[16] = b; x.s_part[32] = c; x.s_part[48] = d;
  }
}

// marginal sums of c0, g1_j and g2_j of this wave's dimensions, Z (two waves; after the barrier behind stage 1b)
template <int CD, class C>
__device__ __forceinline__ void msq_marginals(const MsqM& x) {
  const int lane = threadIdx.x & 63;
  const int nd = __builtin_amdgcn_readfirstlane(x.h_nd), c0 = __builtin_amdgcn_readfirstlane(x.h_c0);
  double mem[MSR_NMEM];
#pragma unroll
  for (int k = 0; k < MSR_NMEM; ++k) mem[k] = *x.g_mem[k];
  double zs[C::NP / 64];
#pragma unroll
  for (int u = 0; u < C::NP / 64; ++u) zs[u] = x.h_c0p[64 * u];
  double zl = zs[0];
#pragma unroll
  for (int u = 1; u < C::NP / 64; ++u) zl += zs[u];
  const double bc = x.h_xg2[c0];
  double s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int k = 0; k < MSR_NMEM; k += 2) { s0 += mem[k]; s1 += mem[k + 1]; }
  double s_ = s0 + s1;
  s_ += dpp_mov<0xB1>(s_);
  s_ += dpp_mov<0x4E>(s_);
  *x.g_out = s_;
  const double zraw = wave_sum(zl);      // the weights of padding rows are zero
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

// outputs of one site (after the barrier behind stage S2).  Sub-band d: fixed-order sums of the tile waves' partials; modulator j: g1, g2.
// acc: a copy of [g1: 8][g2: 8][Z]
template <int CD, class C>
__device__ __forceinline__ void msq_outputs(msp_rp acc, msp_rp part /* + (d & 15) + 16 * (d >> 4) */, bool sub, int jmod, double pEP, double jitter,
                                            double& Z, double& d1, double& d2) {
  const double Zs = acc[16];
  Z = pEP * ((Zs > jitter) ? Zs : jitter);          // max(NaN, jitter) = jitter
  const double Zinv = pEP * rcp_nr(Z);
  double s1, s2;
  if (sub) {
    double a[C::NTW], b[C::NTW];
#pragma unroll
    for (int w = 0; w < C::NTW; ++w) { a[w] = part[64 * w]; b[w] = part[64 * w + 32]; }
    s1 = a[0]; s2 = b[0];
#pragma unroll
    for (int w = 1; w < C::NTW; ++w) { s1 += a[w]; s2 += b[w]; }
  } else {
    s1 = acc[jmod]; s2 = acc[8 + jmod];
  }
  d1 = Zinv * s1;
  d2 = fma(-d1, d1, Zinv * s2);
}

// The cubature of one step in the FLAT layout (256 threads, four waves in every stage): fmu / HPH of all sites are visible (a barrier
// behind their writes) -> Z, d lZ, d2 lZ of site tid < M on wave 0.  Five workgroup barriers; wave 0 evaluates the link tables, wave 1
// forms t = W' s2_z, waves 2, 3 the marginal sums behind their stage S2.
struct MsqLink { int lw; double xdc; msp_rp a_mu, a_s2; msp_wp a_out; };
template <int CD>
__device__ __forceinline__ void msqf_setup_link(MsqLink& x, const MomCfg& c, const double* fmu, const double* HPH, double* ws) {
  const int nd = c.nd, TN = CD * nd, tl = threadIdx.x;
  const int t = (tl < TN) ? tl : 0;
  const int j = t / nd, cc = t - j * nd;
  x.lw = 0; x.xdc = c.xd[cc];
  x.a_mu = (msp_rp)(fmu + c.D + j); x.a_s2 = (msp_rp)(HPH + c.D + j);
  x.a_out = (msp_wp)(ws + t);
}
template <int CD>
__device__ __forceinline__ void msqf_eval(const MsqW<CD, MsqFlat>& x, const MsqLink& xl, const MsqM& xm, const MomCfg& c, double* amp, double sn2a, double y,
                                          msp_rp accp, msp_rp partp, bool sub, int jmod, double pEP, bool site, double& Z, double& d1, double& d2) {
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  if (wave == 0) msp_link<CD>(xl, c);
  msq_stageA<CD, MsqFlat>(x);          // (wave 1)
  lds_barrier();
  msq_stageS<CD, MsqFlat>(x, amp);
  lds_barrier();
  msq_stage1b<CD, MsqFlat>(x, sn2a, y);
  lds_barrier();
  msq_stageS2<CD, MsqFlat>(x, amp);
  if (wave >= 2) msq_marginals<CD, MsqFlat>(xm);
  lds_barrier();
  if (site) msq_outputs<CD, MsqFlat>(accp, partp, sub, jmod, pEP, c.jitter, Z, d1, d2);
}

}  // namespace nagp
