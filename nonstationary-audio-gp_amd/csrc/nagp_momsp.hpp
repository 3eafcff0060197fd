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
//   A   wave 0, lane (j,c): link, xg, xg2 tables            | waves 1-3: Q, 2Q, v by 4-lane groups (static W products)
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
constexpr int MSP_NZ = 4;     // non-centre coordinates per sigma point
constexpr int MSP_DT = 16;    // sub-bands per lane of a 4-lane group in stage A (D <= 64)
constexpr int MSP_MAXCD = 7;  // 2*CD + 2 <= 16 rows of the MFMA block

typedef const double __attribute__((address_space(3))) * msp_rp;   // LDS read pointer (32-bit, register resident)
typedef double __attribute__((address_space(3))) * msp_wp;

struct MomSp {
  int enabled;
  int c0;             // code of the coordinate value 0
  int nzmax;          // largest number of non-centre coordinates of a sigma point
  const int* pdesc;   // [n_pts][MSP_NZ]: j*nd + c of the non-centre coordinates, -1 = none
};

// LDS workspace (offsets in doubles)
struct MspLay { int lk, xg, xg2, e, t1, ve, one, zero, Q, Q2, v, q0, s0, c0, c1, c2, part, acc, total; };
__host__ __device__ inline MspLay msp_layout(int CD, int nd, int n_pts) {
  MspLay l;
  const int TN = CD * nd;
  l.lk = 0; l.xg = TN; l.xg2 = 2 * TN; l.e = 3 * TN; l.t1 = 4 * TN; l.ve = 5 * TN; l.one = 6 * TN; l.zero = 6 * TN + 1;
  l.Q = 6 * TN + 2; l.Q2 = l.Q + CD * CD; l.v = l.Q2 + CD * CD; l.q0 = l.v + CD; l.s0 = l.q0 + 1;
  int o = (l.s0 + 2) & ~1;
  const int cs = (n_pts + 4) | 1;          // + zero-weight dummy points for the padding of the last MFMA steps
  l.c0 = o; l.c1 = o + cs; l.c2 = o + 2 * cs;
  o = (o + 3 * cs + 1) & ~1;
  l.part = o; o += MSP_NW * 256;
  l.acc = o; o += 64;
  l.total = o;
  return l;
}
__host__ __device__ inline size_t msp_lds_doubles(int CD, int nd, int n_pts) { return (size_t)msp_layout(CD, nd, n_pts).total; }
__host__ __device__ inline int msp_nacc(int CD) { return CD + CD * (CD + 1) / 2 + 2 * CD + 1; }   // u, R (upper), g1, g2, Z

// 1/x by the hardware estimate and two Newton steps (~1 ulp); x = 0, inf, NaN are the caller's business
__device__ __forceinline__ double rcp_nr(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0); r = fma(r, e, r);
  e = fma(-x, r, 1.0); r = fma(r, e, r);
  return r;
}

// Register-resident state of one thread.  Everything here is computed once per kernel.
template <int CD>
struct MspCtx {
  // stage A / B, wave 0: lane t = j*nd + c
  int jA; double xdc; msp_rp a_mu, a_s2, a_l0[CD], a_l0own, a_qrow, a_qjj, a_v; msp_wp a_out;   // a_out + {lk,xg,xg2,e,t1,ve}*TN; a_l0[j'] = l0 of modulator j'
  // stage B, wave 1: lane L < CD*CD: Q(j,j') l0_j l0_j' ; next CD lanes: v_j l0_j
  int b_kind; msp_rp b_p0, b_p1, b_p2;
  // stage A, waves 1-3: 4-lane group -> one entry of Q (and 2Q) or v
  int q_kind;          // 0: none, 1: Q(j,j'), 2: v(j)
  double ww[MSP_DT];   // W_dj W_dj' (or W_dj) for d = sub, sub+4, ...
  msp_rp q_src, q_last; double ww_last;   // common operand pointer; the last term (d may pass D) has its own, aimed at a zero
  msp_wp q_out0, q_out1, q_out2, q_out3;
  // stage 1b
  msp_rp p_ve[MSP_NPS][MSP_NZ], p_t1[MSP_NPS][MSP_NZ], p_e[MSP_NPS][MSP_NZ], p_q[MSP_NPS][6];
  msp_wp p_c[MSP_NPS];
  double p_wn[MSP_NPS];
  bool p_ok[MSP_NPS];
  int p_any[MSP_NPS];  // wave-uniform: some lane of this wave owns a point in the slot
  // stage 2
  msp_rp m_a[MSP_NST], m_b[MSP_NST], m_w[MSP_NST];
  int nst;             // steps of this wave
  // partial-sum reduction: lane o < nacc of the reducing wave
  msp_rp r_src; msp_wp r_dst;
};

template <int CD>
__device__ __forceinline__ void msp_setup(MspCtx<CD>& x, const MomCfg& c, const MomSp& sp, const double* Wl /* LDS D x CD */,
                                           const double* fmu, const double* HPH, double* ws) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nd = c.nd, D = c.D, TN = CD * nd, npt = c.n_pts;
  const MspLay l = msp_layout(CD, nd, npt);
  // ---- stage A / B of wave 0
  {
    const int t = (tid < TN) ? tid : 0;
    const int j = t / nd, cc = t - j * nd;
    x.jA = j; x.xdc = c.xd[cc];
    x.a_mu = (msp_rp)(fmu + D + j); x.a_s2 = (msp_rp)(HPH + D + j);
#pragma unroll
    for (int j2 = 0; j2 < CD; ++j2) x.a_l0[j2] = (msp_rp)(ws + l.lk + j2 * nd + sp.c0);
    x.a_l0own = (msp_rp)(ws + l.lk + j * nd + sp.c0);
    x.a_qrow = (msp_rp)(ws + l.Q + j * CD);
    x.a_qjj = (msp_rp)(ws + l.Q + j * CD + j);
    x.a_v = (msp_rp)(ws + l.v + j);
    x.a_out = (msp_wp)(ws + t);
  }
  // ---- stage B of wave 1
  {
    const int L = tid - 64;
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
#pragma unroll
    for (int q = 0; q < MSP_DT; ++q) x.ww[q] = 0.0;
    x.q_out0 = x.q_out1 = x.q_out2 = x.q_out3 = (msp_wp)(ws + l.acc + 63);   // scratch slot
    const int L = tid - 64;
    if (L >= 0) {
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
      const int DTn = (D + 3) >> 2;
      if (x.q_kind) {
#pragma unroll
        for (int q = 0; q < MSP_DT; ++q) {
          const int d = sub + 4 * q;
          if (d < D) x.ww[q] = (x.q_kind == 1) ? Wl[d * CD + j] * Wl[d * CD + j2] : Wl[d * CD + j];
        }
      }
      // terms q < DTn-1 lie inside the sub-bands for every lane; the last one may not: its own pointer and weight
      const double* srcv = (x.q_kind == 1) ? HPH : fmu;
      x.q_src = (msp_rp)(srcv + sub);
      const int dl = sub + 4 * (DTn - 1);
      const bool inl = x.q_kind && dl < D;
      x.q_last = inl ? (msp_rp)(srcv + dl) : (msp_rp)(ws + l.zero);
      x.ww_last = inl ? ((x.q_kind == 1) ? Wl[dl * CD + j] * Wl[dl * CD + j2] : Wl[dl * CD + j]) : 0.0;
    } else {
      x.q_src = x.q_last = (msp_rp)(ws + l.zero); x.ww_last = 0.0;
    }
  }
  // ---- stage 1b: slot 0 = point tid; slot 1 = points 256.. on the LAST wave (wave 0 carries the serial stages)
  {
    const msp_rp zero = (msp_rp)(ws + l.zero);
#pragma unroll
    for (int u = 0; u < MSP_NPS; ++u) {
      int p = (u == 0) ? tid : (MSP_NT * u + (tid - (MSP_NT - 64)));
      const bool ok = (u == 0) ? (p < npt) : (tid >= MSP_NT - 64 && p < npt);
      x.p_ok[u] = ok;
      x.p_any[u] = (__builtin_amdgcn_ballot_w64(ok) != 0) ? 1 : 0;
      if (!ok) p = 0;
      int tj[MSP_NZ];
#pragma unroll
      for (int r = 0; r < MSP_NZ; ++r) tj[r] = ok ? sp.pdesc[(size_t)p * MSP_NZ + r] : -1;
#pragma unroll
      for (int r = 0; r < MSP_NZ; ++r) {
        x.p_ve[u][r] = (tj[r] >= 0) ? (msp_rp)(ws + l.ve + tj[r]) : zero;
        x.p_t1[u][r] = (tj[r] >= 0) ? (msp_rp)(ws + l.t1 + tj[r]) : zero;
        x.p_e[u][r] = (tj[r] >= 0) ? (msp_rp)(ws + l.e + tj[r]) : zero;
      }
      int pi = 0;
#pragma unroll
      for (int r = 0; r < MSP_NZ; ++r)
#pragma unroll
        for (int s = r + 1; s < MSP_NZ; ++s) {
          x.p_q[u][pi] = (tj[r] >= 0 && tj[s] >= 0) ? (msp_rp)(ws + l.Q2 + (tj[r] / nd) * CD + (tj[s] / nd)) : zero;
          ++pi;
        }
      x.p_c[u] = (msp_wp)(ws + l.c0 + p);
      x.p_wn[u] = ok ? c.wn[p] : 0.0;
    }
  }
  // ---- stage 2: wave w takes the steps w, w+4, ... ; lane (i = lane & 15, kq = lane >> 4) feeds row/col i with point 4*step+kq
  //   A_p = [c2 lk_0..lk_{N-1} | c1 | c0 xg2_0..xg2_{N-1} | c0]      B_p = [lk_0..lk_{N-1} | xg_0..xg_{N-1} | 1]
  {
    const int i = lane & 15, kq = lane >> 4;
    const int nstep = (npt + 3) >> 2;
    // wave 0 goes on to the serial part of the step: it takes the short share
    const int wv = (wave + MSP_NW - 1) % MSP_NW;
    x.nst = (nstep - wv + MSP_NW - 1) / MSP_NW;
    if (x.nst < 0) x.nst = 0;
    const int cs = (npt + 4) | 1;
#pragma unroll
    for (int s = 0; s < MSP_NST; ++s) {
      const int p = 4 * (wv + MSP_NW * s) + kq;
      const bool ok = (s < x.nst) && (p < npt);
      const int pp = ok ? p : 0;
      int offA = l.zero, offB = l.zero, offW = l.c0 + npt;     // zero operand, zero-weight dummy point
      if (ok) {
        const unsigned char* cp = c.code + (size_t)pp * CD;
        if (i < CD) { offA = l.lk + i * nd + cp[i]; offW = l.c2 + pp; }
        else if (i == CD) { offA = l.one; offW = l.c1 + pp; }
        else if (i <= 2 * CD) { offA = l.xg2 + (i - CD - 1) * nd + cp[i - CD - 1]; offW = l.c0 + pp; }
        else if (i == 2 * CD + 1) { offA = l.one; offW = l.c0 + pp; }
        if (i < CD) offB = l.lk + i * nd + cp[i];
        else if (i < 2 * CD) offB = l.xg + (i - CD) * nd + cp[i - CD];
        else if (i == 2 * CD) offB = l.one;
      }
      (void)cs;
      x.m_a[s] = (msp_rp)(ws + offA); x.m_b[s] = (msp_rp)(ws + offB); x.m_w[s] = (msp_rp)(ws + offW);
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
    x.r_dst = (msp_wp)(ws + l.acc + ((o < 64) ? o : 63));
  }
  // constants and the dummy points
  if (tid == 0) { ws[l.one] = 1.0; ws[l.zero] = 0.0; }
  if (tid < 4) { ws[l.c0 + npt + tid] = 0.0; ws[l.c1 + npt + tid] = 0.0; ws[l.c2 + npt + tid] = 0.0; }
}

// stage A.  Needs fmu / HPH of all sites visible; ends WITHOUT a barrier (the caller places it).
template <int CD>
__device__ __forceinline__ void msp_stageA(const MspCtx<CD>& x, const MomCfg& c) {
  const int tid = threadIdx.x, TN = CD * c.nd;
  if (tid < 64) {
    if (tid < TN) {
      const double mu = *x.a_mu, s2 = *x.a_s2;
      const double rs = rsqrt_nr(s2);
      const double sg = s2 * rs, inv = rs * rs;
      const double xn = mu + sg * x.xdc;                                   // likModulatorNMFPower.m:34
      const double lk = link_eval(c.link_kind, c.link_shift, xn);
      const double xg = (xn - mu) * inv;                                   // (xn - mu_g)./s2_g  (:72)
      x.a_out[0] = lk; x.a_out[TN] = xg; x.a_out[2 * TN] = xg * xg - inv;   // :79
    }
  } else if (x.q_kind) {
    const int DTn = (c.D + 3) >> 2;
    double a0 = x.ww_last * (*x.q_last), a1 = 0.0;
#pragma unroll
    for (int q = 0; q < MSP_DT - 1; ++q) {
      if (q < DTn - 1) {       // uniform
        if (q & 1) a1 = fma(x.ww[q], x.q_src[4 * q], a1); else a0 = fma(x.ww[q], x.q_src[4 * q], a0);
      }
    }
    double a = a0 + a1;
    a += dpp_mov<0xB1>(a);
    a += dpp_mov<0x4E>(a);
    if ((tid & 3) == 0) {
      if (x.q_kind == 1) { *x.q_out0 = a; *x.q_out1 = a; *x.q_out2 = 2.0 * a; *x.q_out3 = 2.0 * a; }
      else *x.q_out0 = a;
    }
  }
}

// stage B.  After a barrier behind stage A; ends without a barrier.
template <int CD>
__device__ __forceinline__ void msp_stageB(const MspCtx<CD>& x, const MomCfg& c, const MomSp& sp, double* ws) {
  const int tid = threadIdx.x, nd = c.nd, TN = CD * nd;
  const MspLay l = msp_layout(CD, nd, c.n_pts);
  if (tid < 64) {
    if (tid < TN) {
      const double lk = ((msp_rp)x.a_out)[0];
      double ql = 0.0;
#pragma unroll
      for (int j2 = 0; j2 < CD; ++j2) ql = fma(x.a_qrow[j2], *x.a_l0[j2], ql);
      const double l0 = *x.a_l0own;
      const double e = lk - l0;
      const double qjj = *x.a_qjj;
      x.a_out[3 * TN] = e;
      x.a_out[4 * TN] = e * fma(qjj, e, 2.0 * ql);
      x.a_out[5 * TN] = (*x.a_v) * e;
    }
  } else if (tid < 128) {
    const int L = tid - 64;
    const double t = (*x.b_p0) * (*x.b_p1) * (*x.b_p2);      // unused lanes: zero * ...
    double tq = (x.b_kind == 1) ? t : 0.0, tsv = (x.b_kind == 2) ? t : 0.0;
    tq = wave_sum(tq);
    tsv = wave_sum(tsv);
    if (L == 0) { ws[l.q0] = tq; ws[l.s0] = tsv; }
  }
}

// stage 1b.  After a barrier behind stage B; ends without a barrier.
template <int CD>
__device__ __forceinline__ void msp_stage1b(const MspCtx<CD>& x, const MomCfg& c, const MomSp& sp, double sn2a, double y, const double* ws) {
  const MspLay l = msp_layout(CD, c.nd, c.n_pts);
  const int cs = (c.n_pts + 4) | 1;
  const bool four = __builtin_amdgcn_readfirstlane(sp.nzmax > 3 ? 1 : 0) != 0;
  const bool three = __builtin_amdgcn_readfirstlane(sp.nzmax > 2 ? 1 : 0) != 0;
  const double q0 = ws[l.q0], s0 = ws[l.s0];
#pragma unroll
  for (int u = 0; u < MSP_NPS; ++u) {
    if (__builtin_amdgcn_readfirstlane(x.p_any[u]) == 0) continue;   // wave-uniform skip
    double sam = s0 + *x.p_ve[u][0], sa2 = q0 + *x.p_t1[u][0];
    const double e0 = *x.p_e[u][0], e1 = *x.p_e[u][1];
    sam += *x.p_ve[u][1]; sa2 += *x.p_t1[u][1];
    double cr = (*x.p_q[u][0]) * e0 * e1;
    if (three) {
      const double e2 = *x.p_e[u][2];
      sam += *x.p_ve[u][2]; sa2 += *x.p_t1[u][2];
      cr = fma((*x.p_q[u][1]) * e0, e2, cr);
      cr = fma((*x.p_q[u][3]) * e1, e2, cr);
      if (four) {
        const double e3 = *x.p_e[u][3];
        sam += *x.p_ve[u][3]; sa2 += *x.p_t1[u][3];
        cr = fma((*x.p_q[u][2]) * e0, e3, cr);
        cr = fma((*x.p_q[u][4]) * e1, e3, cr);
        cr = fma((*x.p_q[u][5]) * e2, e3, cr);
      }
    }
    sa2 += cr;
    double pdf, q, inv;
    gauss_terms(y, sam, sn2a + sa2, pdf, q, inv);
    const double w0 = x.p_wn[u] * pdf;
    if (x.p_ok[u]) {
      x.p_c[u][0] = w0;
      x.p_c[u][cs] = w0 * q;
      x.p_c[u][2 * cs] = w0 * (q * q - inv);
    }
  }
}

// stage 2.  After a barrier behind stage 1b; leaves this wave's 16x16 partial in LDS, no barrier.
template <int CD>
__device__ __forceinline__ void msp_stage2(const MspCtx<CD>& x, const MomCfg& c, double* ws) {
  const MspLay l = msp_layout(CD, c.nd, c.n_pts);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nst = __builtin_amdgcn_readfirstlane(x.nst);
  v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s0 = 0; s0 < MSP_NST; s0 += 4) {
    if (s0 < nst) {      // uniform; steps beyond nst inside the group of four carry zero operands
      double a[4], bb[4], w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { a[u] = *x.m_a[s0 + u]; bb[u] = *x.m_b[s0 + u]; w[u] = *x.m_w[s0 + u]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u] * w[u], bb[u], acc, 0, 0, 0);
    }
  }
  const int i = lane & 15, kq = lane >> 4;
  double* part = ws + l.part + wave * 256;
#pragma unroll
  for (int r = 0; r < 4; ++r) part[(kq + 4 * r) * 16 + i] = acc[r];
}

// fixed-order sum of the partials: lanes o < msp_nacc(CD) of ONE wave; the same wave may read acc after msp_wave_fence()
template <int CD>
__device__ __forceinline__ void msp_reduce(const MspCtx<CD>& x) {
  const int lane = threadIdx.x & 63;
  if (lane < msp_nacc(CD)) {
    const double a = ((x.r_src[0] + x.r_src[256]) + x.r_src[512]) + x.r_src[768];
    *x.r_dst = a;
  }
}
// orders the LDS traffic of one wave (DS operations of a wave complete in issue order; no workgroup barrier)
__device__ __forceinline__ void msp_wave_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// acc layout: [u: CD][R upper, row-major: CD(CD+1)/2][g1: CD][g2: CD][Z]
// site n < D: s1 = W_n . u, s2 = W_n' R W_n (w2[] = (2 - delta) W_nj W_nj' in the order of the upper triangle);
// site D + j: s1 = g1_j, s2 = g2_j
template <int CD>
__device__ __forceinline__ void msp_outputs(const double* acc, int n, int D, const double* wrow, const double* w2, double pEP, double jitter,
                                            double& Z, double& d1, double& d2) {
  constexpr int nq = CD * (CD + 1) / 2;
  const double Zs = acc[3 * CD + nq];
  Z = pEP * ((Zs > jitter) ? Zs : jitter);          // max(NaN, jitter) = jitter (likModulatorNMFPower.m:55)
  const double Zinv = pEP / Z;
  double s1, s2;
  if (n < D) {
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0;
#pragma unroll
    for (int j = 0; j < CD; ++j) { if (j & 1) a1 = fma(wrow[j], acc[j], a1); else a0 = fma(wrow[j], acc[j], a0); }
#pragma unroll
    for (int q = 0; q < nq; ++q) {
      const double r = acc[CD + q];
      if (q % 3 == 0) b0 = fma(w2[q], r, b0); else if (q % 3 == 1) b1 = fma(w2[q], r, b1); else b2 = fma(w2[q], r, b2);
    }
    s1 = a0 + a1; s2 = (b0 + b1) + b2;
  } else {
    s1 = acc[CD + nq + (n - D)];
    s2 = acc[2 * CD + nq + (n - D)];
  }
  d1 = Zinv * s1;
  d2 = fma(-d1, d1, Zinv * s2);
}

}  // namespace nagp
