// nagp_momsp.hpp -- likModulatorNMFPower (matlab/likModulatorNMFPower.m:28-87) for fully symmetric sigma-point sets,
// laid out for the sequential ADF step: one wave per SIMD, every LDS address static and register resident.
//
// The sigma points of ut3/5/7/9_ws (matlab/symmetric-cubature-rules) are the centre plus points with at most four
// non-centre coordinates.  With lk_p = l0 + dev_p (l0 = link at the centre of every modulator, dev_p non-zero in the
// non-centre dimensions only) the per-point quantities of likModulatorNMFPower.m:44-47 become
//   sum_d a_d mu_d   = s0 + sum_r ve[j_r][c_r]                      s0 = v.l0,     ve[j][c] = v_j e_j(c)
//   sum_d a_d^2 s2_d = q0 + sum_r t1[j_r][c_r] + sum_{r<s} 2 Q(j_r,j_s) e_r e_s    q0 = l0'Q l0,  t1 = e (2 (Q l0)_j + Q_jj e)
// with Q = W' diag(s2_z) W, v = W' mu_z (the N x N forms of mom_quad) and e_j(c) = link(xn_{j,c}) - l0_j: a dozen table
// reads per sigma point instead of an N x N quadratic form.  The weighted sums over the points (:59-80) are one 16x16
// block of A'B on v_mfma_f64_16x16x4 exactly as in mom_quad, but every lane's operand addresses are computed ONCE per
// kernel (they depend on the static cubature codes only), so a step of the accumulation is three LDS reads, one multiply
// and the MFMA.  Same arithmetic as the reference up to summation order.
//
// Stages (256 threads = 4 waves; a workgroup barrier between stages):
//   A   wave 0, lane (j,c): link, xg, xg2 tables            | waves 1-3: Q, 2Q, v by 4-lane groups (static W products in LDS)
//   B   wave 0, lane (j,c): e, t1, ve tables                | wave 1: q0, s0
//   1b  one lane per sigma point (<= MSP_NPS per lane): Gaussian weight -> c0, c1, c2
//   2   every wave: its share of the MFMA steps -> 16x16 partial in LDS
//   3   (caller's choice of lanes, after a barrier) fixed-order sum of the partials, outputs d lZ, d2 lZ, Z
#pragma once
#include "nagp_dev.hpp"

namespace nagp {

constexpr int MSP_NT = 256;   // threads per workgroup the stages are written for
constexpr int MSP_NW = 4;
constexpr int MSP_NPS = 2;    // sigma points per lane in stage 1b
constexpr int MSP_NST = 20;   // MFMA steps per wave in stage 2 (4 points each)
constexpr int MSP_DT = 16;    // sub-bands per lane of a 4-lane group in stage A (D <= 64)
constexpr int MSP_MAXCD = 7;  // 2*CD + 2 <= 16 rows of the MFMA block
constexpr int MSP_TS = 64;    // stride of the (dimension, coordinate) tables: CD*nd <= 63, entry 63 of e / t1 / ve stays zero
constexpr int MSP_CS = 4 * MSP_NW * MSP_NST + 1;   // stride of c0 / c1 / c2: every point an MFMA step can address, zero beyond n_pts

typedef const double __attribute__((address_space(3))) * msp_rp;   // LDS read pointer (32-bit, register resident)
typedef double __attribute__((address_space(3))) * msp_wp;

// terms per lane of the 4-lane groups that form Q and v (stage A): the next of 4, 8, 16 that covers D sub-bands
__host__ __device__ inline int msp_qterms(int D) { return D <= 16 ? 4 : (D <= 32 ? 8 : 16); }

// LDS workspace (offsets in doubles).  Tables lk | xg | xg2 | e | t1 | ve, MSP_TS entries each.
struct MspLay { int lk, xg, xg2, e, t1, ve, one, zero, Q, Q2, v, q0, s0, c0, c1, c2, part, acc, marg, wwt, total; };
// LAY 0: the 256-thread layout (four waves share every stage); LAY 1: the role layout of 512 threads (nagp_ihgp.hpp:
// two serial waves, six worker waves: six partial blocks, 96 addressable MFMA steps)
constexpr int MSR_NWK = 6;     // worker waves of the role layout
constexpr int MSR_NST = 20;    // MFMA steps of a worker wave that has its SIMD's matrix core to itself (the others take half)
constexpr int MSR_NSTP = 16;   // ... in the packed form (eight points per step, four MFMA workers)
constexpr int MSR_NMEM = 16;   // members of a marginal sum per lane (four lanes per marginal)
constexpr int MSR_NMARG = 32;  // marginal sums (non-centre (dimension, coordinate) pairs)
constexpr int MSR_CS = 8 * 48 + 8 * MSR_NSTP + 1;      // every point a (zero-padded) MFMA step can address
__host__ __device__ inline MspLay msp_layout(int CD, int D, int LAY = 0) {
  MspLay l;
  const int cs = LAY ? MSR_CS : MSP_CS, nparts = LAY ? MSR_NWK : MSP_NW;
  l.lk = 0; l.xg = MSP_TS; l.xg2 = 2 * MSP_TS; l.e = 3 * MSP_TS; l.t1 = 4 * MSP_TS; l.ve = 5 * MSP_TS;
  l.zero = l.e + MSP_TS - 1;                     // e[63] (t1[63], ve[63] are zero as well)
  l.one = 6 * MSP_TS;
  l.Q = l.one + 1; l.Q2 = l.Q + CD * CD; l.v = l.Q2 + CD * CD; l.q0 = l.v + CD; l.s0 = l.q0 + 1;
  int o = (l.s0 + 2) & ~1;
  l.c0 = o; l.c1 = o + cs; l.c2 = o + 2 * cs;
  o = (o + 3 * cs + 1) & ~1;
  l.part = o; o += nparts * 256;
  l.acc = o; o += 128;
  l.marg = o; o += LAY ? MSR_NMARG : 0;
  l.wwt = o; o += msp_qterms(D) * 192;            // static W products of stage A, [q][lane of waves 1..3], zero for sub-bands >= D
  l.total = o;
  return l;
}
__host__ __device__ inline size_t msp_lds_doubles(int CD, int D, int LAY = 0) { return (size_t)msp_layout(CD, D, LAY).total; }
__host__ __device__ inline int msp_nacc(int CD) { return CD + CD * (CD + 1) / 2 + 2 * CD + 1; }   // u, R (upper), g1, g2, Z

// 1/x by the hardware estimate and two Newton steps (~1 ulp)
// (x = 0, inf or of underflow size: the refinement is NaN -- callers on such values branch to true divisions)
__device__ __forceinline__ double rcp_nr(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0); r = fma(r, e, r);
  e = fma(-x, r, 1.0); r = fma(r, e, r);
  return r;
}
// a zero the compiler cannot see through: added to a wave-uniform LDS address it keeps the address in ONE vector register
// (uniform addresses are otherwise materialised one scalar register per constant offset, and spilled)
__device__ __forceinline__ int opaque_zero() { int z = 0; asm volatile("" : "+v"(z)); return z; }

// Register-resident state of one thread.  Everything here is computed once per kernel.
template <int CD>
struct MspCtx {
  static constexpr int NPS = MSP_NPS, NST = MSP_NST, WSTR = 4 * MSP_NW, NPART = MSP_NW, CS = MSP_CS;
  static constexpr bool PACKED = false;
  int lw, qw;          // wave that evaluates the link tables / wave that forms q0, s0 (wave-uniform)
  // stage A / B, wave lw: lane t = j*nd + c
  double xdc; msp_rp a_mu, a_s2, a_l0[CD], a_l0own, a_qrow, a_qjj, a_v; msp_wp a_out;   // a_out[k*MSP_TS]: lk, xg, xg2, e, t1, ve
  // stage B, wave 1: lane L < CD*CD: Q(j,j') l0_j l0_j' ; next CD lanes: v_j l0_j
  int b_kind; msp_rp b_p0, b_p1, b_p2;
  // stage A, waves 1-3: 4-lane group -> one entry of Q (and 2Q) or v
  int q_kind;          // 0: none, 1: Q(j,j'), 2: v(j)
  msp_rp q_ww, q_src;  // products at q_ww[192*q] (zero for sub-bands >= D), operands at q_src[4*q]
  msp_wp q_out0, q_out1, q_out2, q_out3;
  // stage 1b
  msp_rp p_e[MSP_NPS][MSP_NZ], p_q[MSP_NPS][6];   // e at p_e[0], t1 at [MSP_TS], ve at [2*MSP_TS]
  msp_wp p_c[MSP_NPS];
  double p_wn[MSP_NPS];
  bool p_ok[MSP_NPS];
  int p_any[MSP_NPS];  // wave-uniform: some lane of this wave owns a point in the slot
  // stage 2
  msp_rp m_a[MSP_NST], m_b[MSP_NST], m_w0;   // weight of step s at m_w0[WSTR*s]
  msp_wp m_part;       // this wave's 16x16 partial block, + lane
  int nst, m_on;       // steps of this wave; the wave takes part in stage 2 (wave-uniform)
  // partial-sum reduction: lane o < nacc of the reducing wave
  msp_rp r_src; msp_wp r_dst;
  msp_rp accp;         // reduced sums (one vector register, immediate offsets)
};

template <int CD>
__device__ __forceinline__ void msp_setup(MspCtx<CD>& x, const MomCfg& c, const MomSp& sp, const double* Wl /* LDS D x CD */,
                                           const double* fmu, const double* HPH, double* ws, int LW = 0) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nd = c.nd, D = c.D, TN = CD * nd, npt = c.n_pts;
  const MspLay l = msp_layout(CD, D);
  const int oz = opaque_zero();
  // roles: the link tables on wave LW (0: the wave that owns the sites; 1: a kernel that gives the modulator sites to wave 1
  // lets their link evaluation run beside the sub-band sites' serial work), Q / v on the three other waves, q0 / s0 on wave qw
  x.lw = LW; x.qw = (LW == 1) ? 2 : 1;
  const int tl = tid - 64 * LW;                                  // lane index on the link wave
  const int wq = (wave < LW) ? wave : wave - 1;                  // rank among the Q / v waves
  const int Lq = (wave != LW && wave < MSP_NW) ? wq * 64 + lane : -1;   // 0 .. 191
  // constants, zero entries of the tables, zero weights beyond the points
  for (int i = tid; i < 6 * MSP_TS; i += MSP_NT) ws[i] = 0.0;
  if (tid == 0) ws[l.one] = 1.0;
  for (int i = tid; i < 3 * MSP_CS; i += MSP_NT) ws[l.c0 + i] = 0.0;
  // ---- stage A / B of wave 0
  {
    const int t = (tl >= 0 && tl < TN) ? tl : 0;
    const int j = t / nd, cc = t - j * nd;
    x.xdc = c.xd[cc];
    x.a_mu = (msp_rp)(fmu + D + j); x.a_s2 = (msp_rp)(HPH + D + j);
#pragma unroll
    for (int j2 = 0; j2 < CD; ++j2) x.a_l0[j2] = (msp_rp)(ws + l.lk + j2 * nd + sp.c0) + oz;
    x.a_l0own = (msp_rp)(ws + l.lk + j * nd + sp.c0);
    x.a_qrow = (msp_rp)(ws + l.Q + j * CD);
    x.a_qjj = (msp_rp)(ws + l.Q + j * CD + j);
    x.a_v = (msp_rp)(ws + l.v + j);
    x.a_out = (msp_wp)(ws + t);
  }
  // ---- stage B: q0, s0 on wave qw
  {
    const int L = tid - 64 * x.qw;
    x.b_kind = 0; x.b_p0 = x.b_p1 = x.b_p2 = (msp_rp)(ws + l.zero);
    if (L >= 0 && L < CD * CD) {
      const int j = L / CD, j2 = L - j * CD;
      x.b_kind = 1; x.b_p0 = (msp_rp)(ws + l.Q + L); x.b_p1 = (msp_rp)(ws + l.lk + j * nd + sp.c0); x.b_p2 = (msp_rp)(ws + l.lk + j2 * nd + sp.c0);
    } else if (L >= CD * CD && L < CD * CD + CD) {
      const int j = L - CD * CD;
      x.b_kind = 2; x.b_p0 = (msp_rp)(ws + l.v + j); x.b_p1 = (msp_rp)(ws + l.lk + j * nd + sp.c0); x.b_p2 = (msp_rp)(ws + l.one);
    }
  }
  // ---- stage A of waves 1..3
  {
    x.q_kind = 0;
    x.q_out0 = x.q_out1 = x.q_out2 = x.q_out3 = (msp_wp)(ws + l.acc + 127);   // scratch slot
    x.q_ww = x.q_src = (msp_rp)(ws + l.zero);
    const int L = Lq;
    if (L >= 0 && L < 192) {      // the three waves beside the link wave (a larger workgroup's further waves take no part in the cubature)
      const int g = L >> 2, sub = L & 3;
      const int nq = CD * (CD + 1) / 2;
      int j = 0, j2 = 0;
      if (g < nq) {          // upper-triangular pair (j, j2), j <= j2
        int r = g; j = 0;
        while (r >= CD - j) { r -= CD - j; ++j; }
        j2 = j + r;
        x.q_kind = 1;
        x.q_out0 = (msp_wp)(ws + l.Q + j * CD + j2); x.q_out1 = (msp_wp)(ws + l.Q + j2 * CD + j);
        x.q_out2 = (msp_wp)(ws + l.Q2 + j * CD + j2); x.q_out3 = (msp_wp)(ws + l.Q2 + j2 * CD + j);
      } else if (g < nq + CD) {
        j = g - nq; x.q_kind = 2;
        x.q_out0 = (msp_wp)(ws + l.v + j);
      }
      // zero weights for sub-bands >= D: the operand read there (a modulator's entry, or the zero padding of fmu / HPH) is multiplied by 0
      for (int q = 0; q < msp_qterms(D); ++q) {
        const int d = sub + 4 * q;
        double v = 0.0;
        if (x.q_kind && d < D) v = (x.q_kind == 1) ? Wl[d * CD + j] * Wl[d * CD + j2] : Wl[d * CD + j];
        ws[l.wwt + q * 192 + L] = v;
      }
      x.q_ww = (msp_rp)(ws + l.wwt + L);
      x.q_src = (msp_rp)(((x.q_kind == 1) ? HPH : fmu) + sub);
    }
  }
  // ---- stage 1b: slot 0 = point tid; slot 1 = points 256.. on the LAST wave (wave 0 carries the serial stages)
  {
    const msp_rp zero = (msp_rp)(ws + l.zero);
#pragma unroll
    for (int u = 0; u < MSP_NPS; ++u) {
      int p = (u == 0) ? tid : (MSP_NT * u + (tid - (MSP_NT - 64)));
      const bool ok = (tid < MSP_NT) && ((u == 0) ? (p < npt) : (tid >= MSP_NT - 64 && p < npt));
      x.p_ok[u] = ok;
      x.p_any[u] = (__builtin_amdgcn_ballot_w64(ok) != 0) ? 1 : 0;
      if (!ok) p = 0;
      int tj[MSP_NZ];
#pragma unroll
      for (int r = 0; r < MSP_NZ; ++r) {
        tj[r] = ok ? sp.pdesc[(size_t)p * MSP_NZ + r] : -1;
        if (tj[r] >= 0) tj[r] = (tj[r] / nd) * 64 + (tj[r] % nd);      // (j, c) packed as j*64 + c
      }
#pragma unroll
      for (int r = 0; r < MSP_NZ; ++r) x.p_e[u][r] = (tj[r] >= 0) ? (msp_rp)(ws + l.e + (tj[r] >> 6) * nd + (tj[r] & 63)) : zero;
      int pi = 0;
#pragma unroll
      for (int r = 0; r < MSP_NZ; ++r)
#pragma unroll
        for (int s = r + 1; s < MSP_NZ; ++s) {
          x.p_q[u][pi] = (tj[r] >= 0 && tj[s] >= 0) ? (msp_rp)(ws + l.Q2 + (tj[r] >> 6) * CD + (tj[s] >> 6)) : zero;
          ++pi;
        }
      x.p_c[u] = (msp_wp)(ws + l.c0 + p);
      x.p_wn[u] = ok ? c.wn[p] : 0.0;
    }
  }
  // ---- stage 2: wave w takes the steps w', w'+4, ... ; lane (i = lane & 15, kq = lane >> 4) feeds row/col i with point 4*step+kq
  //   A_p = [c2 lk_0..lk_{N-1} | c1 | c0 xg2_0..xg2_{N-1} | c0]      B_p = [lk_0..lk_{N-1} | xg_0..xg_{N-1} | 1]
  {
    const int i = lane & 15, kq = lane >> 4;
    const int nstep = (npt + 3) >> 2;
    // wave 0 goes on to the serial part of the step: it takes the short share
    const int wv = (wave + MSP_NW - 1) % MSP_NW;
    x.nst = (nstep - wv + MSP_NW - 1) / MSP_NW;
    if (x.nst < 0 || wave >= MSP_NW) x.nst = 0;
    int wbase = l.c0;                                // rows without a weight multiply a zero operand
    if (i < CD) wbase = l.c2; else if (i == CD) wbase = l.c1;
    x.m_w0 = (msp_rp)(ws + wbase + 4 * wv + kq);     // point of step s: 4*(wv + 4s) + kq
    x.m_part = (msp_wp)(ws + l.part + ((wave < MSP_NW) ? wave : 0) * 256 + kq * 16 + i);
    x.m_on = (wave < MSP_NW) ? 1 : 0;
#pragma unroll
    for (int s = 0; s < MSP_NST; ++s) {
      const int p = 4 * (wv + MSP_NW * s) + kq;
      const bool ok = (s < x.nst) && (p < npt) && (wave < MSP_NW);
      int offA = l.zero, offB = l.zero;             // beyond the points: zero operands (their weights are zero too)
      if (ok) {
        const unsigned char* cp = c.code + (size_t)p * CD;
        if (i < CD) offA = l.lk + i * nd + cp[i];
        else if (i == CD) offA = l.one;
        else if (i <= 2 * CD) offA = l.xg2 + (i - CD - 1) * nd + cp[i - CD - 1];
        else if (i == 2 * CD + 1) offA = l.one;
        if (i < CD) offB = l.lk + i * nd + cp[i];
        else if (i < 2 * CD) offB = l.xg + (i - CD) * nd + cp[i - CD];
        else if (i == 2 * CD) offB = l.one;
      }
      x.m_a[s] = (msp_rp)(ws + offA); x.m_b[s] = (msp_rp)(ws + offB);
    }
  }
  // ---- reduction of the four 16x16 partials: lane o -> (row, col) of the block
  {
    const int o = lane, nq = CD * (CD + 1) / 2;
    int row = 0, col = 0;
    if (o < CD) { row = CD; col = o; }                                        // u_j
    else if (o < CD + nq) { int r = o - CD, j = 0; while (r >= CD - j) { r -= CD - j; ++j; } row = j; col = j + r; }   // R(j,j'), j <= j'
    else if (o < 2 * CD + nq) { row = 2 * CD + 1; col = CD + (o - CD - nq); }    // g1_j
    else if (o < 3 * CD + nq) { row = CD + 1 + (o - 2 * CD - nq); col = 2 * CD; } // g2_j
    else { row = 2 * CD + 1; col = 2 * CD; }                                      // Z
    x.r_src = (msp_rp)(ws + l.part + row * 16 + col);
    const int ab = (wave == 1) ? 64 : 0;                           // waves 0 and 1 may both reduce: own buffers
    x.r_dst = (msp_wp)(ws + l.acc + ab + o);
    x.accp = (msp_rp)(ws + l.acc + ab) + oz;
  }
}

template <int K>
__device__ __forceinline__ void msp_qsum(msp_rp ww, msp_rp src, double& a0, double& a1) {
  double w[K], v[K];
#pragma unroll
  for (int q = 0; q < K; ++q) { w[q] = ww[192 * q]; v[q] = src[4 * q]; }
#pragma unroll
  for (int q = 0; q < K; q += 2) { a0 = fma(w[q], v[q], a0); a1 = fma(w[q + 1], v[q + 1], a1); }
}

// stage A, link part (wave lw): needs fmu / HPH of the MODULATOR sites; no barrier
template <int CD, class X>
__device__ __forceinline__ void msp_link(const X& x, const MomCfg& c) {
  const int tl = (int)threadIdx.x - 64 * x.lw, TN = CD * c.nd;
  if (tl >= 0 && tl < TN) {
    const double mu = *x.a_mu, s2 = *x.a_s2;
    // sqrt(s2) and 1/s2 through one reciprocal square root -- inside the range where its Newton steps are exact to an ulp; a cavity
    // variance of 1e200 or 1e-200 (the site refresh of an ill-conditioned sweep produces them) takes the square root and the division
    double sg, inv;
    if (s2 > 1e-150 && s2 < 1e150) { const double rs = rsqrt_nr(s2); sg = s2 * rs; inv = rs * rs; }
    else { sg = sqrt(s2); inv = 1.0 / s2; }
    const double xn = mu + sg * x.xdc;                                   // likModulatorNMFPower.m:34
    const double lk = link_eval(c.link_kind, c.link_shift, xn);
    const double xg = (xn - mu) * inv;                                   // (xn - mu_g)./s2_g  (:72)
    x.a_out[0] = lk; x.a_out[MSP_TS] = xg; x.a_out[2 * MSP_TS] = xg * xg - inv;   // :79
  }
}
// stage A, Q / 2Q / v part (the three other waves): needs fmu / HPH of the SUB-BAND sites; no barrier
template <int CD, class X>
__device__ __forceinline__ void msp_qv(const X& x, const MomCfg& c) {
  if (x.q_kind) {
    const int K = __builtin_amdgcn_readfirstlane(msp_qterms(c.D));
    double a0 = 0.0, a1 = 0.0;
    if (K == 4) msp_qsum<4>(x.q_ww, x.q_src, a0, a1);
    else if (K == 8) msp_qsum<8>(x.q_ww, x.q_src, a0, a1);
    else msp_qsum<16>(x.q_ww, x.q_src, a0, a1);
    double a = a0 + a1;
    a += dpp_mov<0xB1>(a);
    a += dpp_mov<0x4E>(a);
    if ((threadIdx.x & 3) == 0) {
      if (x.q_kind == 1) { *x.q_out0 = a; *x.q_out1 = a; *x.q_out2 = 2.0 * a; *x.q_out3 = 2.0 * a; }
      else *x.q_out0 = a;
    }
  }
}
// stage A.  Needs fmu / HPH of all sites visible; ends WITHOUT a barrier (the caller places it).
template <int CD>
__device__ __forceinline__ void msp_stageA(const MspCtx<CD>& x, const MomCfg& c) {
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  if (wave >= MSP_NW) return;
  if (wave == __builtin_amdgcn_readfirstlane(x.lw)) msp_link<CD>(x, c); else msp_qv<CD>(x, c);
}

// stage B, table part (lanes t < CD*nd of the link wave): e, t1, ve from lk, Q, v
template <int CD, class X>
__device__ __forceinline__ void msp_tables(const X& x, const MomCfg& c) {
  const int TN = CD * c.nd;
  if (((int)threadIdx.x & 63) < TN) {
    // every operand first -- ONE LDS round trip (left to itself the compiler reads a pair, waits, multiplies, reads the next pair: a dozen
    // round trips of ~100 cycles on a wave that has its SIMD to itself) -- then arithmetic, then the three stores
    double lk = ((msp_rp)x.a_out)[0], l0o = *x.a_l0own, qjj = *x.a_qjj, vv = *x.a_v;
    double qr[CD], l0v[CD];
#pragma unroll
    for (int j2 = 0; j2 < CD; ++j2) { qr[j2] = x.a_qrow[j2]; l0v[j2] = *x.a_l0[j2]; }
#pragma unroll
    for (int j2 = 0; j2 < CD; ++j2) asm volatile("" : "+v"(qr[j2]), "+v"(l0v[j2]));
    asm volatile("" : "+v"(lk), "+v"(l0o), "+v"(qjj), "+v"(vv));
    double ql0 = 0.0, ql1 = 0.0;
#pragma unroll
    for (int j2 = 0; j2 < CD; ++j2) { if (j2 & 1) ql1 = fma(qr[j2], l0v[j2], ql1); else ql0 = fma(qr[j2], l0v[j2], ql0); }
    const double ql = ql0 + ql1;
    const double e = lk - l0o;
    x.a_out[3 * MSP_TS] = e;
    x.a_out[4 * MSP_TS] = e * fma(qjj, e, 2.0 * ql);
    x.a_out[5 * MSP_TS] = vv * e;
  }
}
// stage B, q0 = l0' Q l0 and s0 = v' l0 (one wave)
template <class X>
__device__ __forceinline__ void msp_q0s0(const X& x, double* q0, double* s0) {
  const double t = (*x.b_p0) * (*x.b_p1) * (*x.b_p2);      // unused lanes: zero * ...
  double tq = (x.b_kind == 1) ? t : 0.0, tsv = (x.b_kind == 2) ? t : 0.0;
  tq = wave_sum(tq);
  tsv = wave_sum(tsv);
  if (((int)threadIdx.x & 63) == 0) { *q0 = tq; *s0 = tsv; }
}
// the same, one sum per wave (role layout): the wave holding the Q terms writes q0, the wave holding the v terms writes s0
template <class X>
__device__ __forceinline__ void msr_q0_or_s0(const X& x, bool is_q0, double* q0, double* s0) {
  const double t = (*x.b_p0) * (*x.b_p1) * (*x.b_p2);      // unused lanes: zero * ...
  const double sum = wave_sum(t);
  if (((int)threadIdx.x & 63) == 0) { if (is_q0) *q0 = sum; else *s0 = sum; }
}
// stage B.  After a barrier behind stage A; ends without a barrier.
template <int CD>
__device__ __forceinline__ void msp_stageB(const MspCtx<CD>& x, const MomCfg& c, double* ws) {
  const MspLay l = msp_layout(CD, c.D);
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  if (wave == __builtin_amdgcn_readfirstlane(x.lw)) msp_tables<CD>(x, c);
  else if (wave == __builtin_amdgcn_readfirstlane(x.qw)) msp_q0s0(x, ws + l.q0, ws + l.s0);
}

__device__ __forceinline__ void msp_wave_fence();
// Folded stage B of the role layout (CD*CD <= 48): this wave's own copy of the tables -- every worker wave writes the same values to the
// same LDS words and reads them back behind its own writes (DS operations of a wave complete in issue order) -- and q0 = l0' Q l0,
// s0 = v' l0 from one 16-lane group sum: rows 0..2 of the wave hold the q0 terms, row 3 the s0 terms, at the lane positions (inside a row)
// and in the association order of the two-wave form (msr_q0_or_s0), so both sums keep their bits.
template <int CD, class X>
__device__ __forceinline__ void msr_fold(const X& x, const MomCfg& c, double& q0, double& s0) {
  static_assert(CD * CD <= 48, "q0 terms beyond three rows of the wave");
  const double t = (*x.f_p0) * (*x.f_p1) * (*x.f_p2);      // unused lanes: zero * ...
  msp_tables<CD>(x, c);
  const double v = group_sum(t, 16);
  q0 = (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + 0.0);
  s0 = (readlane_d(v, 48) + 0.0) + (0.0 + 0.0);
  msp_wave_fence();
}

// stage 1b.  After a barrier behind stage B; ends without a barrier.
template <int CD, class X>
__device__ __forceinline__ void msp_stage1b_qs(const X& x, const MomCfg& c, const MomSp& sp, double sn2a, double y, double q0, double s0);
template <int CD, class X>
__device__ __forceinline__ void msp_stage1b(const X& x, const MomCfg& c, const MomSp& sp, double sn2a, double y, const double* ws) {
  const MspLay l = msp_layout(CD, c.D);      // q0, s0 sit at the same offsets in both layouts
  msp_stage1b_qs<CD>(x, c, sp, sn2a, y, ws[l.q0], ws[l.s0]);
}
template <int CD, class X>
__device__ __forceinline__ void msp_stage1b_qs(const X& x, const MomCfg& c, const MomSp& sp, double sn2a, double y, double q0, double s0) {
  const bool four = __builtin_amdgcn_readfirstlane(sp.nzmax > 3 ? 1 : 0) != 0;
  const bool three = __builtin_amdgcn_readfirstlane(sp.nzmax > 2 ? 1 : 0) != 0;
#pragma unroll
  for (int u = 0; u < X::NPS; ++u) {
    if (__builtin_amdgcn_readfirstlane(x.p_any[u]) == 0) continue;   // wave-uniform skip
    // all table entries of the point first (one LDS round trip; the entries of components a point does not have are reads of the zero
    // word), then arithmetic in the order of the reference form
    double ev[MSP_NZ], tv[MSP_NZ], vv[MSP_NZ], qv[6];
#pragma unroll
    for (int r = 0; r < MSP_NZ; ++r) {
      const bool on = r < 2 || (r == 2 && three) || (r == 3 && four);
      ev[r] = tv[r] = vv[r] = 0.0;
      if (on) { ev[r] = x.p_e[u][r][0]; tv[r] = x.p_e[u][r][MSP_TS]; vv[r] = x.p_e[u][r][2 * MSP_TS]; }
    }
    qv[0] = *x.p_q[u][0];
    qv[1] = qv[2] = qv[3] = qv[4] = qv[5] = 0.0;
    if (three) { qv[1] = *x.p_q[u][1]; qv[3] = *x.p_q[u][3]; }
    if (four) { qv[2] = *x.p_q[u][2]; qv[4] = *x.p_q[u][4]; qv[5] = *x.p_q[u][5]; }
#pragma unroll
    for (int r = 0; r < MSP_NZ; ++r) asm volatile("" : "+v"(ev[r]), "+v"(tv[r]), "+v"(vv[r]));
#pragma unroll
    for (int r = 0; r < 6; ++r) asm volatile("" : "+v"(qv[r]));
    const double e0 = ev[0], e1 = ev[1];
    double sam = (s0 + vv[0]) + vv[1];
    double sa2 = (q0 + tv[0]) + tv[1];
    double cr = qv[0] * e0 * e1;
    if (three) {
      const double e2 = ev[2];
      sam += vv[2]; sa2 += tv[2];
      cr = fma(qv[1] * e0, e2, cr);
      cr = fma(qv[3] * e1, e2, cr);
      if (four) {
        const double e3 = ev[3];
        sam += vv[3]; sa2 += tv[3];
        cr = fma(qv[2] * e0, e3, cr);
        cr = fma(qv[4] * e1, e3, cr);
        cr = fma(qv[5] * e2, e3, cr);
      }
    }
    sa2 += cr;
    double pdf, q, inv;
    gauss_terms(y, sam, sn2a + sa2, pdf, q, inv);
    const double w0 = x.p_wn[u] * pdf;
    if (x.p_ok[u]) {
      x.p_c[u][0] = w0;
      x.p_c[u][X::CS] = w0 * q;
      x.p_c[u][2 * X::CS] = w0 * (q * q - inv);
    }
  }
}

// stage 2.  After a barrier behind stage 1b; leaves this wave's 16x16 partial in LDS, no barrier.
// Two accumulators (the dependent-accumulator latency of the f64 MFMA is longer than its issue time) and the operands of
// the next four steps in flight while the current four multiply.
template <int CD, class X>
__device__ __forceinline__ void msp_stage2(const X& x, const MomCfg& c, double* ws) {
  if (__builtin_amdgcn_readfirstlane(x.m_on) == 0) return;      // a wave outside the stage has no partial block
  const int nst = __builtin_amdgcn_readfirstlane(x.nst);
  constexpr int NST = X::NST, WS = X::WSTR;
  constexpr bool SAMEB = X::PACKED;      // packed form: the column operand is the unweighted row operand (no second read)
  v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
  double a[4], bb[4], w[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { a[u] = *x.m_a[u]; bb[u] = SAMEB ? a[u] : *x.m_b[u]; w[u] = x.m_w0[WS * u]; }
#pragma unroll
  for (int s0 = 0; s0 < NST; s0 += 4) {
    if (s0 < nst) {      // uniform; steps beyond nst inside the group of four carry zero operands
      double an[4], bn[4], wn_[4];
      if (s0 + 4 < NST) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { an[u] = *x.m_a[s0 + 4 + u]; bn[u] = SAMEB ? an[u] : *x.m_b[s0 + 4 + u]; wn_[u] = x.m_w0[WS * (s0 + 4 + u)]; }
      }
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0] * w[0], bb[0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1] * w[1], bb[1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2] * w[2], bb[2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3] * w[3], bb[3], acc1, 0, 0, 0);
      if (s0 + 4 < NST) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = an[u]; bb[u] = bn[u]; w[u] = wn_[u]; }
      }
    }
  }
  if constexpr (X::PACKED) {
    // the two diagonal 8x8 blocks (points 8s+kq / 8s+4+kq) added in registers: element (8+r, 8+c) sits in lane i = c + 8 of the
    // same 16-lane row, register 2 + r/4; the partial block that goes to LDS is 8 x 8
    const v4d a = acc0 + acc1;
    const double t0 = a[0] + dpp_mov<0x108>(a[2]), t1 = a[1] + dpp_mov<0x108>(a[3]);     // row_shl:8
    if (((int)threadIdx.x & 15) < 8) { x.m_part[0] = t0; x.m_part[64] = t1; }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) x.m_part[64 * r] = acc0[r] + acc1[r];      // element (kq + 4r, i) of the block
  }
}

// fixed-order sum of the partials: lanes o < msp_nacc(CD) of ONE wave; the same wave may read acc after msp_wave_fence()
template <int CD, class X>
__device__ __forceinline__ void msp_reduce(const X& x) {
  const int lane = threadIdx.x & 63;
  constexpr int nq = CD * (CD + 1) / 2;
  bool on = lane < msp_nacc(CD);
  if constexpr (X::PACKED) on = on && !(lane >= CD + nq && lane < 3 * CD + nq);     // g1, g2 come from the marginal sums
  if (on) {
    double a = x.r_src[0];
#pragma unroll
    for (int w = 1; w < X::NPART; ++w) a += x.r_src[256 * w];
    *x.r_dst = a;
  }
}
// orders the LDS traffic of one wave (DS operations of a wave complete in issue order; no workgroup barrier)
__device__ __forceinline__ void msp_wave_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// acc layout: [u: CD][R upper, row-major: CD(CD+1)/2][g1: CD][g2: CD][Z]
// sub-band site (sub): s1 = W_n . u, s2 = W_n' R W_n ;  modulator site j: s1 = g1_j, s2 = g2_j
// returns d lZ and d2 lZ (likModulatorNMFPower.m:59-80); Z = pEP*max(sum, jitter) (:55)
template <int CD>
__device__ __forceinline__ void msp_outputs(msp_rp acc, bool sub, int jmod, const double* wrow, double pEP, double jitter,
                                            double& Z, double& d1, double& d2) {
  constexpr int nq = CD * (CD + 1) / 2;
  const double Zs = acc[3 * CD + nq];
  Z = pEP * ((Zs > jitter) ? Zs : jitter);          // max(NaN, jitter) = jitter
  const double Zinv = pEP * rcp_nr(Z);              // Z >= pEP*jitter > 0, finite (or inf: Zinv = NaN like inf/inf)
  double s1, s2;
  if (sub) {
    // all reads first (one LDS round trip), then arithmetic only
    double av[CD + nq];
#pragma unroll
    for (int q = 0; q < CD + nq; ++q) av[q] = acc[q];
#pragma unroll
    for (int q = 0; q < CD + nq; ++q) asm volatile("" : "+v"(av[q]));
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int j = 0; j < CD; ++j) { if (j & 1) a1 = fma(wrow[j], av[j], a1); else a0 = fma(wrow[j], av[j], a0); }
    s1 = a0 + a1;
    // w'Rw = sum_j 2 w_j (R_jj w_j / 2 + sum_{j2 > j} R_jj2 w_j2): CD independent chains, then one
    double b0 = 0.0, b1 = 0.0;
    int q = 0;
#pragma unroll
    for (int j = 0; j < CD; ++j) {
      double t = 0.5 * av[CD + q] * wrow[j]; ++q;
#pragma unroll
      for (int j2 = j + 1; j2 < CD; ++j2) { t = fma(av[CD + q], wrow[j2], t); ++q; }
      if (j & 1) b1 = fma(2.0 * wrow[j], t, b1); else b0 = fma(2.0 * wrow[j], t, b0);
    }
    s2 = b0 + b1;
  } else {
    s1 = acc[CD + nq + jmod];
    s2 = acc[2 * CD + nq + jmod];
  }
  d1 = Zinv * s1;
  d2 = fma(-d1, d1, Zinv * s2);
}

// =====================================================================================================================
// Role layout (LAY = 1): 512 threads.  Waves 0 and 1 carry the serial stages of the caller (wave 1 also evaluates the link
// tables and e / t1 / ve); waves 2 .. 7 carry the parallel ones: Q / 2Q / v on waves 2..4, q0 / s0 on wave 5, one sigma
// point per lane of the six (<= 384 points), the MFMA steps round the six.  The two roles run in separate loops of the
// kernel, so a wave holds the registers of its own role only: two waves per SIMD within 256 registers each.
constexpr int MSR_NT = 512;
constexpr int MSR_W0 = 2;      // first worker wave

// PACK (<= 6 components): an MFMA step takes EIGHT points -- rows / columns 0..7 of the block belong to points 8s+kq, rows /
// columns 8..15 to points 8s+4+kq; A_p = [c2 lk_0.. | c1 | c0], B_p = [lk_0.. | 1], and the two diagonal 8x8 blocks are summed.
// Half the MFMA steps; the sums over c0 xg_j and c0 xg2_j, which no longer fit the block, come from the marginal sums
// C0(j,c) = sum of c0 over the points with coordinate c in dimension j (static member lists, two lanes of one worker wave per
// marginal, beside the MFMA steps):  g1_j = sum_c C0(j,c) xg(j,c),  g2_j = sum_c C0(j,c) xg2(j,c),  C0(j,centre) = sum c0 - rest.
template <int CD, bool PACK>
struct MsrS {
  static constexpr int NPART = MSR_NWK;
  static constexpr bool PACKED = PACK;
  int lw;
  double xdc; msp_rp a_mu, a_s2, a_l0[CD], a_l0own, a_qrow, a_qjj, a_v; msp_wp a_out;
  msp_rp r_src; msp_wp r_dst; msp_rp accp;
};
template <int CD, bool PACK>
struct MsrW {
  static constexpr int NPS = 1, NST = PACK ? MSR_NSTP : MSR_NST, WSTR = PACK ? 8 : 4, CS = MSR_CS;
  static constexpr bool PACKED = PACK;
  int q_kind; msp_rp q_ww, q_src; msp_wp q_out0, q_out1, q_out2, q_out3;
  int b_kind; msp_rp b_p0, b_p1, b_p2;
  // folded form (msr_fold): EVERY worker wave forms the tables e / t1 / ve (lanes < CD*nd) and q0, s0 for itself between B2 and the
  // weights -- no barrier B3, no wait for the serial wave
  msp_rp a_l0[CD], a_l0own, a_qrow, a_qjj, a_v; msp_wp a_out; msp_rp f_p0, f_p1, f_p2;
  msp_rp p_e[1][MSP_NZ], p_q[1][6]; msp_wp p_c[1]; double p_wn[1]; bool p_ok[1]; int p_any[1];
  msp_rp m_a[NST], m_b[NST], m_w0; msp_wp m_part; int nst, m_on;
  msp_rp g_mem[PACK ? MSR_NMEM : 1]; msp_wp g_out;              // last worker: members of this lane's half of a marginal
  msp_rp h_marg, h_xg, h_xg2, h_c0p; msp_wp h_acc; int h_nd, h_c0, h_nj;   // ... lane < h_nj: g1_j, g2_j of its dimension; every lane: its share of sum c0
};

// constants, zero entries of the tables, zero weights beyond the points (every thread of the workgroup)
__device__ __forceinline__ void msr_init(int CD, int D, double* ws) {
  const MspLay l = msp_layout(CD, D, 1);
  for (int i = threadIdx.x; i < 6 * MSP_TS; i += MSR_NT) ws[i] = 0.0;
  if (threadIdx.x == 0) ws[l.one] = 1.0;
  for (int i = threadIdx.x; i < 3 * MSR_CS; i += MSR_NT) ws[l.c0 + i] = 0.0;
  for (int i = threadIdx.x; i < MSR_NWK * 256; i += MSR_NT) ws[l.part + i] = 0.0;   // partial blocks of workers without MFMA steps stay zero
}

template <int CD, bool PACK>
__device__ __forceinline__ void msr_setup_S(MsrS<CD, PACK>& x, const MomCfg& c, const MomSp& sp, const double* fmu, const double* HPH, double* ws) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nd = c.nd, D = c.D, TN = CD * nd;
  const MspLay l = msp_layout(CD, D, 1);
  const int oz = opaque_zero();
  x.lw = 1;
  {
    const int tl = tid - 64;
    const int t = (tl >= 0 && tl < TN) ? tl : 0;
    const int j = t / nd, cc = t - j * nd;
    x.xdc = c.xd[cc];
    x.a_mu = (msp_rp)(fmu + D + j); x.a_s2 = (msp_rp)(HPH + D + j);
#pragma unroll
    for (int j2 = 0; j2 < CD; ++j2) x.a_l0[j2] = (msp_rp)(ws + l.lk + j2 * nd + sp.c0) + oz;
    x.a_l0own = (msp_rp)(ws + l.lk + j * nd + sp.c0);
    x.a_qrow = (msp_rp)(ws + l.Q + j * CD);
    x.a_qjj = (msp_rp)(ws + l.Q + j * CD + j);
    x.a_v = (msp_rp)(ws + l.v + j);
    x.a_out = (msp_wp)(ws + t);
  }
  {
    const int o = lane, nq = CD * (CD + 1) / 2;
    int row = 0, col = 0;
    if (o < CD) { row = CD; col = o; }
    else if (o < CD + nq) { int r = o - CD, j = 0; while (r >= CD - j) { r -= CD - j; ++j; } row = j; col = j + r; }
    else if (o < 2 * CD + nq) { row = 2 * CD + 1; col = CD + (o - CD - nq); }
    else if (o < 3 * CD + nq) { row = CD + 1 + (o - 2 * CD - nq); col = 2 * CD; }
    else { row = 2 * CD + 1; col = 2 * CD; }
    if constexpr (PACK) {       // block rows [c2 lk_j | c1 | c0], columns [lk_j | 1]: u and R sit where they sat; the g1 / g2 lanes
      if (o >= CD + nq && o < 3 * CD + nq) { row = 0; col = 7; }      // read a zero column, Z is (c0 row, column of ones)
      else if (o >= 3 * CD + nq) { row = CD + 1; col = CD; }
    }
    x.r_src = (msp_rp)(ws + l.part + row * 16 + col);
    const int ab = (wave == 1) ? 64 : 0;
    x.r_dst = (msp_wp)(ws + l.acc + ab + o);
    x.accp = (msp_rp)(ws + l.acc + ab) + oz;
  }
}

template <int CD, bool PACK>
__device__ __forceinline__ void msr_setup_W(MsrW<CD, PACK>& x, const MomCfg& c, const MomSp& sp, const double* Wl /* LDS D x CD */,
                                             const double* fmu, const double* HPH, double* ws) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave - MSR_W0;     // wr = 0 .. 5
  const int nd = c.nd, D = c.D, npt = c.n_pts;
  const MspLay l = msp_layout(CD, D, 1);
  // ---- q0 on worker 3, s0 on worker 4 (one wave sum each, side by side between barriers B2 and B3)
  {
    const int L = tid - 64 * (MSR_W0 + 3), L2 = tid - 64 * (MSR_W0 + 4);
    x.b_kind = 0; x.b_p0 = x.b_p1 = x.b_p2 = (msp_rp)(ws + l.zero);
    if (L >= 0 && L < CD * CD) {
      const int j = L / CD, j2 = L - j * CD;
      x.b_kind = 1; x.b_p0 = (msp_rp)(ws + l.Q + L); x.b_p1 = (msp_rp)(ws + l.lk + j * nd + sp.c0); x.b_p2 = (msp_rp)(ws + l.lk + j2 * nd + sp.c0);
    } else if (L2 >= 0 && L2 < CD) {
      const int j = L2;
      x.b_kind = 2; x.b_p0 = (msp_rp)(ws + l.v + j); x.b_p1 = (msp_rp)(ws + l.lk + j * nd + sp.c0); x.b_p2 = (msp_rp)(ws + l.one);
    }
  }
  // ---- folded form: table lane t = lane (as on the serial wave's lanes, msr_setup_S); q0 terms on lanes 0 .. CD*CD-1 and the s0 terms
  // on lanes 48 .. 48+CD-1 of ONE 16-lane group sum (CD*CD <= 48), or the layout of the two-wave form twice (msr_fold)
  {
    const int TN = CD * nd;
    const int t = (lane < TN) ? lane : 0;
    const int j = t / nd;
    const int oz = opaque_zero();
#pragma unroll
    for (int j2 = 0; j2 < CD; ++j2) x.a_l0[j2] = (msp_rp)(ws + l.lk + j2 * nd + sp.c0) + oz;
    x.a_l0own = (msp_rp)(ws + l.lk + j * nd + sp.c0);
    x.a_qrow = (msp_rp)(ws + l.Q + j * CD);
    x.a_qjj = (msp_rp)(ws + l.Q + j * CD + j);
    x.a_v = (msp_rp)(ws + l.v + j);
    x.a_out = (msp_wp)(ws + t);
    x.f_p0 = x.f_p1 = x.f_p2 = (msp_rp)(ws + l.zero);
    if (lane < CD * CD && CD * CD <= 48) {
      const int jq = lane / CD, j2 = lane - jq * CD;
      x.f_p0 = (msp_rp)(ws + l.Q + lane); x.f_p1 = (msp_rp)(ws + l.lk + jq * nd + sp.c0); x.f_p2 = (msp_rp)(ws + l.lk + j2 * nd + sp.c0);
    } else if (lane >= 48 && lane < 48 + CD && CD * CD <= 48) {
      const int js = lane - 48;
      x.f_p0 = (msp_rp)(ws + l.v + js); x.f_p1 = (msp_rp)(ws + l.lk + js * nd + sp.c0); x.f_p2 = (msp_rp)(ws + l.one);
    }
  }
  // ---- Q / 2Q / v on workers 0..2: 4-lane group -> one entry
  {
    x.q_kind = 0;
    x.q_out0 = x.q_out1 = x.q_out2 = x.q_out3 = (msp_wp)(ws + l.acc + 127);
    x.q_ww = x.q_src = (msp_rp)(ws + l.zero);
    const int L = (wr >= 0 && wr < 3) ? wr * 64 + lane : -1;
    if (L >= 0) {
      const int g = L >> 2, sub = L & 3;
      const int nq = CD * (CD + 1) / 2;
      int j = 0, j2 = 0;
      if (g < nq) {
        int r = g; j = 0;
        while (r >= CD - j) { r -= CD - j; ++j; }
        j2 = j + r;
        x.q_kind = 1;
        x.q_out0 = (msp_wp)(ws + l.Q + j * CD + j2); x.q_out1 = (msp_wp)(ws + l.Q + j2 * CD + j);
        x.q_out2 = (msp_wp)(ws + l.Q2 + j * CD + j2); x.q_out3 = (msp_wp)(ws + l.Q2 + j2 * CD + j);
      } else if (g < nq + CD) {
        j = g - nq; x.q_kind = 2;
        x.q_out0 = (msp_wp)(ws + l.v + j);
      }
      for (int q = 0; q < msp_qterms(D); ++q) {
        const int d = sub + 4 * q;
        double v = 0.0;
        if (x.q_kind && d < D) v = (x.q_kind == 1) ? Wl[d * CD + j] * Wl[d * CD + j2] : Wl[d * CD + j];
        ws[l.wwt + q * 192 + L] = v;
      }
      x.q_ww = (msp_rp)(ws + l.wwt + L);
      x.q_src = (msp_rp)(((x.q_kind == 1) ? HPH : fmu) + sub);
    }
  }
  // ---- stage 1b: worker lane (tid - 128) = point
  {
    const msp_rp zero = (msp_rp)(ws + l.zero);
    int p = tid - 64 * MSR_W0;
    const bool ok = p >= 0 && p < npt;
    x.p_ok[0] = ok;
    x.p_any[0] = (__builtin_amdgcn_ballot_w64(ok) != 0) ? 1 : 0;
    if (!ok) p = 0;
    int tj[MSP_NZ];
#pragma unroll
    for (int r = 0; r < MSP_NZ; ++r) {
      tj[r] = ok ? sp.pdesc[(size_t)p * MSP_NZ + r] : -1;
      if (tj[r] >= 0) tj[r] = (tj[r] / nd) * 64 + (tj[r] % nd);
    }
#pragma unroll
    for (int r = 0; r < MSP_NZ; ++r) x.p_e[0][r] = (tj[r] >= 0) ? (msp_rp)(ws + l.e + (tj[r] >> 6) * nd + (tj[r] & 63)) : zero;
    int pi = 0;
#pragma unroll
    for (int r = 0; r < MSP_NZ; ++r)
#pragma unroll
      for (int s_ = r + 1; s_ < MSP_NZ; ++s_) {
        x.p_q[0][pi] = (tj[r] >= 0 && tj[s_] >= 0) ? (msp_rp)(ws + l.Q2 + (tj[r] >> 6) * CD + (tj[s_] >> 6)) : zero;
        ++pi;
      }
    x.p_c[0] = (msp_wp)(ws + l.c0 + p);
    x.p_wn[0] = ok ? c.wn[p] : 0.0;
  }
  // ---- stage 2: contiguous step ranges.  A wave's FP64 MFMA holds its SIMD's issue, so two workers on one SIMD gain nothing
  // over one: waves w and w+4 share a SIMD, the serial waves 0 / 1 sit out this stage, hence workers 2 and 3 (waves 4, 5) have
  // a matrix core to themselves and take a double share; the steps are dealt in eight slots (the first nstep % 8 one longer).
  {
    constexpr int PPS = PACK ? 8 : 4;                 // points per step
    constexpr int NSTX = MsrW<CD, PACK>::NST;
    const int i = lane & 15, kq = lane >> 4;
    const int grp = PACK ? (i >> 3) : 0, f = PACK ? (i & 7) : i;
    const int nstep = (npt + PPS - 1) / PPS;
    const bool on = wr >= 0 && wr < MSR_NWK;
    // PACK: the last two workers form the marginal sums instead, on the SIMDs of workers 0 and 1 -- which therefore take one
    // slot each, workers 2 and 3 three
    const int base = nstep >> 3, rem = nstep & 7;
    auto slot_start = [&](int sl) { return sl * base + (sl < rem ? sl : rem); };     // first step of slot sl (sl = 8: nstep)
    int sl0, sl1;
    if (PACK) { sl0 = (wr <= 1) ? wr : ((wr == 2) ? 2 : 5); sl1 = (wr <= 1) ? wr + 1 : ((wr == 2) ? 5 : 8); }
    else { sl0 = (wr <= 2) ? ((wr == 2) ? 2 : wr) : ((wr == 3) ? 4 : wr + 2); sl1 = (wr == 2 || wr == 3) ? sl0 + 2 : sl0 + 1; }
    const bool mf = on && (!PACK || wr < 4);
    const int st0 = mf ? slot_start(sl0) : 0, st1 = mf ? slot_start(sl1) : 0;
    x.nst = st1 - st0;
    x.m_on = (on && x.nst > 0) ? 1 : 0;   // a worker without steps skips the stage: its partial block was zeroed by msr_init
    int wbase = l.c0;
    if (PACK) { if (f < CD) wbase = l.c2; else if (f == CD) wbase = l.c1; }
    else { if (i < CD) wbase = l.c2; else if (i == CD) wbase = l.c1; }
    x.m_w0 = (msp_rp)(ws + wbase + PPS * st0 + 4 * grp + kq);
    x.m_part = (msp_wp)(ws + l.part + (on ? wr : 0) * 256 + kq * 16 + i);
#pragma unroll
    for (int s_ = 0; s_ < NSTX; ++s_) {
      const int p = PPS * (st0 + s_) + 4 * grp + kq;
      const bool ok = on && (s_ < x.nst) && (p < npt);
      int offA = l.zero, offB = l.zero;
      if (ok) {
        const unsigned char* cp = c.code + (size_t)p * CD;
        if constexpr (PACK) {
          if (f < CD) { offA = l.lk + f * nd + cp[f]; offB = offA; }
          else if (f == CD || f == CD + 1) { offA = l.one; offB = l.one; }     // (column CD+1 duplicates the column of ones: unused sums)
        } else {
          if (i < CD) offA = l.lk + i * nd + cp[i];
          else if (i == CD) offA = l.one;
          else if (i <= 2 * CD) offA = l.xg2 + (i - CD - 1) * nd + cp[i - CD - 1];
          else if (i == 2 * CD + 1) offA = l.one;
          if (i < CD) offB = l.lk + i * nd + cp[i];
          else if (i < 2 * CD) offB = l.xg + (i - CD) * nd + cp[i - CD];
          else if (i == 2 * CD) offB = l.one;
        }
      }
      x.m_a[s_] = (msp_rp)(ws + offA); x.m_b[s_] = (msp_rp)(ws + offB);
    }
  }
  // ---- marginal sums (PACK): workers 4 and 5 take the dimensions below / from jsplit; lane = 4 * (local marginal) + quarter,
  // the members of a marginal go round its four lanes
  x.g_out = (msp_wp)(ws + l.acc + 127);
  if constexpr (PACK) {
    const msp_rp zero = (msp_rp)(ws + l.zero);
    const int jsplit = (CD + 1) / 2;
    const int jlo = (wr == MSR_NWK - 2) ? 0 : jsplit, jhi = (wr == MSR_NWK - 2) ? jsplit : CD;
    const int ml = lane >> 2, quarter = lane & 3;
    const int jj = jlo + ml / (nd - 1), cc = ml % (nd - 1);
    const bool valid = (wr >= MSR_NWK - 2) && jj < jhi;
    const int j = valid ? jj : 0;
    const int code = (cc < sp.c0) ? cc : cc + 1;
    if (valid && quarter == 0) x.g_out = (msp_wp)(ws + l.marg + j * (nd - 1) + cc);
    {   // g1_j, g2_j of modulator j = jlo + lane, written into the copy of the reduced sums that wave 1 (the modulator sites) reads
      const int nq = CD * (CD + 1) / 2;
      const int jm = (jlo + lane < jhi) ? jlo + lane : 0;
      x.h_nd = nd; x.h_c0 = sp.c0; x.h_nj = (wr >= MSR_NWK - 2) ? jhi - jlo : 0;
      x.h_marg = (msp_rp)(ws + l.marg + jm * (nd - 1));
      x.h_xg = (msp_rp)(ws + l.xg + jm * nd);
      x.h_xg2 = (msp_rp)(ws + l.xg2 + jm * nd);
      x.h_acc = (msp_wp)(ws + l.acc + 64 + CD + nq + jm);
      x.h_c0p = (msp_rp)(ws + l.c0 + lane);
    }
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

// marginal sums of c0, then g1_j, g2_j of this wave's dimensions (workers 4 and 5, beside the MFMA steps of the others;
// after the barrier behind stage 1b)
template <int CD, class X>
__device__ __forceinline__ void msr_marginals(const X& x) {
  const int lane = threadIdx.x & 63;
  const int nd = __builtin_amdgcn_readfirstlane(x.h_nd), c0 = __builtin_amdgcn_readfirstlane(x.h_c0);
  // everything that does not depend on the marginals first (one LDS round trip): members, this lane's share of sum c0, the
  // table values of the g1 / g2 lanes (the other lanes read dimension 0's)
  double mem[MSR_NMEM];
#pragma unroll
  for (int k = 0; k < MSR_NMEM; ++k) mem[k] = *x.g_mem[k];
  const double z0 = x.h_c0p[0] + x.h_c0p[64], z1 = x.h_c0p[128] + x.h_c0p[192], z2 = x.h_c0p[256] + x.h_c0p[320];
  const bool small = nd <= 5;
  double a[4], b[4];
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) {
    const int ci = (cc < nd - 1) ? cc : 0;
    const int code = (ci < c0) ? ci : ci + 1;
    a[cc] = x.h_xg[code]; b[cc] = x.h_xg2[code];
  }
  const double bc = x.h_xg2[c0];
  double s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int k = 0; k < MSR_NMEM; k += 2) { s0 += mem[k]; s1 += mem[k + 1]; }
  double s_ = s0 + s1;
  s_ += dpp_mov<0xB1>(s_);
  s_ += dpp_mov<0x4E>(s_);            // the four lanes of the marginal
  *x.g_out = s_;                      // lanes other than the first of a marginal: scratch slot
  const double zraw = wave_sum((z0 + z1) + z2);      // the weights beyond n_pts are zero; MSR_CS >= 384
  msp_wave_fence();
  if (lane < __builtin_amdgcn_readfirstlane(x.h_nj)) {
    double g1 = 0.0, g2 = 0.0, ms = 0.0;
    if (small) {
      double m[4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) m[cc] = x.h_marg[(cc < nd - 1) ? cc : 0];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const double mm = (cc < nd - 1) ? m[cc] : 0.0;
        g1 = fma(mm, a[cc], g1); g2 = fma(mm, b[cc], g2); ms += mm;
      }
    } else {
      for (int cc = 0; cc < nd - 1; ++cc) {
        const int code = (cc < c0) ? cc : cc + 1;
        const double mm = x.h_marg[cc];
        g1 = fma(mm, x.h_xg[code], g1); g2 = fma(mm, x.h_xg2[code], g2); ms += mm;
      }
    }
    g2 = fma(zraw - ms, bc, g2);      // xg of the centre coordinate is zero
    x.h_acc[0] = g1; x.h_acc[CD] = g2;
  }
}

}  // namespace nagp
