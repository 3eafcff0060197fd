// nagp_gfadf8.hpp -- the ADF sweep of the full-covariance filter (gf_ep_modulator_nmf.m:126-176, every step calls mom) with
// role-specialised waves: 512 threads.
//   * ALL eight waves own covariance tiles (TPT lower-triangular 4x4 tiles per thread) and run the covariance phases of a step --
//     prediction A P A' + Q, the panel W = P H', the rank-M update, the PF stores -- two waves per SIMD (the 256-thread launch of
//     gf_filter_kernel<..., SP = 1> runs them on one wave per SIMD, at the issue rate of a lone wave: profiles/r03_filter_phase_costs.txt);
//   * between the panel and the update the cubature of likModulatorNMFPower runs in the role layout of nagp_momsp.hpp exactly as in
//     ihgp_adf8_kernel: waves 2..7 the parallel stages (Q / 2Q / v, Gaussian weights, MFMA sums, marginal sums), wave 1 the link
//     tables, waves 0 / 1 the sites (lane d of wave 0 = sub-band d, lane j of wave 1 = modulator j): moments, site update, gain
//     coefficients.  The two roles are two loops of the kernel, so a wave holds the cubature registers of its own role only.
// Same inputs, outputs, ring and progress protocol as gf_filter_kernel; serves the launches with mom at every step (sweep 1).
#pragma once
#include "nagp_kernels.hpp"
#include <type_traits>

namespace nagp {

__host__ __device__ inline size_t gf_adf8_lds_doubles(const Shape& s, int CD, int kb) {
  size_t n = LDS_INT_DOUBLES + 2 * (size_t)s.M * TS + s.M + (size_t)s.D * s.N + s.S + 4 * (size_t)s.M * s.M +
             6 * (size_t)s.M + 2 * 68 + 8 + 2 + filter_ring_doubles(s, kb) + 2 + msp_lds_doubles(CD, s.D, 1);
  return (n + 1) & ~(size_t)1;
}

// ST = false: the two serial waves own NO tiles (the tiles of <= 384 * TPT lower tiles sit on the six worker waves): the serial role --
// the tail of every step's dependence chain -- then holds its sites and nothing else
template <int TPT, int CD, bool PACK, bool ST>
__global__ void __launch_bounds__(MSR_NT) gf_adf8_kernel(Shape sh, Bufs b, MomCfg mc, FilterPar fp) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x;
  constexpr int NT = MSR_NT;
  constexpr int NTT = ST ? MSR_NT : MSR_NT - 64 * MSR_W0;     // threads that own tiles
  const int tix = ST ? tid : tid - 64 * MSR_W0;                 // tile-thread index (< 0: none)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = sh.S, M = sh.M, D = sh.D, KB = fp.kb;
  const int64_t T = sh.T;
  const int pb = blockIdx.x;
  const double* mdl = b.model + (size_t)pb * mdl_size(sh);

  int* ioff = reinterpret_cast<int*>(lds);          // [MAXM+1]
  int* ibsz = ioff + (MAXM + 1);                     // [MAXM]
  double* sA = lds + LDS_INT_DOUBLES;                // A blocks, stride TS
  double* sQ = sA + (size_t)M * TS;                  // Q blocks, stride TS
  double* shv = sQ + (size_t)M * TS;
  double* sW = shv + M;
  double* m = sW + (size_t)sh.D * sh.N;
  double* Wl = lds + ((((size_t)((m + S) - lds)) + 1) & ~(size_t)1);   // panel W = P H' (layout: gf_filter_kernel)
  double* fmu = Wl + (size_t)M * 4 * M;
  double* HPH = fmu + 68;
  double* tt = HPH + 68;
  double* tn = tt + M;
  double* cA = tn + M;
  double* cm = cA + M;
  double* misc = cm + 3 * M;
  double* ry = misc + 8;                   // ring: y[KB]
  double* rlZ = ry + KB;
  double* rZ = rlZ + KB;
  double* rtt = rZ + KB;
  double* rtn = rtt + (size_t)KB * M;
  double* rR = rtn + (size_t)KB * M;
  double* rfm = rR + (size_t)KB * M;
  double* rfv = rfm + (size_t)KB * M;
  double* rMF = rfv + (size_t)KB * M;
  double* ws = lds + ((((size_t)((rMF + (size_t)KB * S) - lds)) + 1) & ~(size_t)1);   // cubature workspace, role layout

  for (int i = tid; i <= M; i += NT) ioff[i] = sh.off[i];
  for (int i = tid; i < M; i += NT) ibsz[i] = sh.bsz[i];
  for (int i = tid; i < M * 16; i += NT) {
    sA[(i >> 4) * TS + (i & 15)] = mdl[mdl_A(sh) + i];
    sQ[(i >> 4) * TS + (i & 15)] = mdl[mdl_Q(sh) + i];
  }
  for (int i = tid; i < M; i += NT) shv[i] = mdl[mdl_h(sh) + i];
  for (int i = tid; i < sh.D * sh.N; i += NT) sW[i] = mdl[mdl_W(sh) + i];
  for (int i = tid; i < 68; i += NT) { fmu[i] = 0.0; HPH[i] = 0.0; }
  const double sn2 = mdl[mdl_sn2(sh)];
  const double sn2a = sn2 / fp.mom_alpha;
  const double pEP1 = mom_pEP(mc, sn2, fp.mom_alpha);
  msr_init(CD, D, ws);

  // lower-triangular tiles (I >= J), tile t = tid + q * NT
  struct { int I[TPT], J[TPT]; bool ok[TPT]; } own;
  const int nlow = M * (M + 1) / 2;
#pragma unroll
  for (int q = 0; q < TPT; ++q) {
    const int t = tix + q * NTT;
    own.ok[q] = tix >= 0 && t < nlow;
    const int tt_ = own.ok[q] ? t : 0;
    int I = (int)((sqrt(8.0 * tt_ + 1.0) - 1.0) * 0.5);
    while ((I + 1) * (I + 2) / 2 <= tt_) ++I;
    while (I * (I + 1) / 2 > tt_) --I;
    own.I[q] = I; own.J[q] = tt_ - I * (I + 1) / 2;
  }
  double P[TPT][16];
  const double* st = b.state + (size_t)pb * ((size_t)sh.ntiles * 16 + S);
#pragma unroll
  for (int q = 0; q < TPT; ++q) {
    tile_zero(P[q]);
    if (own.ok[q]) {
      if (fp.k_begin > 0)
        pf_tile_load(P[q], b.PF + ((size_t)pb * T + (fp.k_begin - 1)) * (size_t)(((nlow + 63) & ~63) * 16), tix + q * NTT);
      else if (fp.init_from_state && !fp.reset_P)
        tile_load(P[q], st + (size_t)(own.I[q] * M + own.J[q]) * 16);
      else if (own.I[q] == own.J[q])
        tile_load(P[q], mdl + mdl_P(sh) + (size_t)own.I[q] * 16);
    }
  }
  for (int i = tid; i < S; i += NT)
    m[i] = (fp.k_begin > 0) ? b.MF[((size_t)pb * T + (fp.k_begin - 1)) * S + i]
                            : (fp.init_from_state ? st[(size_t)sh.ntiles * 16 + i] : 0.0);
  __syncthreads();
  // state lanes: thread i < S carries state i through the vector work of a step
  const bool slane = tid < S;
  int myblk = 0, myrow = 0, my_o = 0, my_bs = 0;
  if (slane) {
    while (ioff[myblk + 1] <= tid) ++myblk;
    myrow = tid - ioff[myblk];
    my_o = ioff[myblk]; my_bs = ibsz[myblk];
  }

  const double* yv = b.y + (size_t)pb * T;
  double* g_tt = b.ttau + (size_t)pb * T * M;
  double* g_tn = b.tnu + (size_t)pb * T * M;
  double* g_R = b.R + (size_t)pb * T * M;
  double* g_lZ = b.lZ + (size_t)pb * T;
  double* g_MF = b.MF + (size_t)pb * T * S;
  double* g_fm = b.fm + (size_t)pb * T * M;
  double* g_fv = b.fv + (size_t)pb * T * M;
  const int pf_tiles = (nlow + 63) & ~63;
  double* g_PF = (b.PF && fp.store_PF) ? b.PF + (size_t)pb * T * pf_tiles * 16 : nullptr;
  unsigned long long n_nan = 0;

  // ---- the covariance phases of a step, shared by both roles (inlined into each loop)
  // S0: prediction (registers), publish W = P H', diag(H P H'), fmu = H m; returns the predicted mean of this state lane
  auto phase_predict = [&](bool pred, auto tiles) -> double {
    double rm = 0.0;
    if (slane) {
      if (pred) {
        const double* a = sA + (size_t)myblk * TS + 4 * myrow;
        const double* mb = m + my_o;
#pragma unroll
        for (int l = 0; l < 4; ++l)
          if (l < my_bs) rm = fma(a[l], mb[l], rm);
      } else {
        rm = m[tid];
      }
      if (myrow == 0) fmu[myblk] = shv[myblk] * rm;
    }
    if constexpr (decltype(tiles)::value)
#pragma unroll
    for (int q = 0; q < TPT; ++q) {
      if (own.ok[q]) {
        const int I = own.I[q], J = own.J[q];
        if (pred) {
          tile_congruence(P[q], sA + (size_t)I * TS, sA + (size_t)J * TS);
          if (I == J) {
            const double* Qb = sQ + (size_t)I * TS;
#pragma unroll
            for (int e = 0; e < 16; ++e) P[q][e] += Qb[e];
          }
        }
        const double hJ = shv[J], hI = shv[I];
#pragma unroll
        for (int i = 0; i < 4; ++i) Wl[(((size_t)J * 2 + (i >> 1)) * M + I) * 2 + (i & 1)] = hJ * P[q][4 * i];
        if (I == J) {
          HPH[I] = hI * hI * P[q][0];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) Wl[(((size_t)I * 2 + (j >> 1)) * M + J) * 2 + (j & 1)] = hI * P[q][j];
        }
      }
    }
    return rm;
  };
  // mean update m += W cm (state lanes), P -= sum_n cA[n] W[:,n] W[:,n]' (tiles)
  auto phase_update = [&](double rm, auto tiles) -> double {
    if (slane) {
      double a0 = rm, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      const double* wp = Wl + ((size_t)(myrow >> 1) * M + myblk) * 2 + (myrow & 1);
      int n = 0;
      for (; n + 8 <= M; n += 8) {
        double w8[8], c8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { w8[u] = wp[(size_t)(n + u) * 4 * M]; c8[u] = cm[n + u]; }
        a0 = fma(w8[0], c8[0], a0); a1 = fma(w8[1], c8[1], a1); a2 = fma(w8[2], c8[2], a2); a3 = fma(w8[3], c8[3], a3);
        a0 = fma(w8[4], c8[4], a0); a1 = fma(w8[5], c8[5], a1); a2 = fma(w8[6], c8[6], a2); a3 = fma(w8[7], c8[7], a3);
      }
      for (; n + 4 <= M; n += 4) {
        a0 = fma(wp[(size_t)(n + 0) * 4 * M], cm[n + 0], a0);
        a1 = fma(wp[(size_t)(n + 1) * 4 * M], cm[n + 1], a1);
        a2 = fma(wp[(size_t)(n + 2) * 4 * M], cm[n + 2], a2);
        a3 = fma(wp[(size_t)(n + 3) * 4 * M], cm[n + 3], a3);
      }
      if (n < M) {
        double wt[3], ct[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) { const int nn = (n + u < M) ? n + u : M - 1; wt[u] = wp[(size_t)nn * 4 * M]; ct[u] = cm[nn]; }
#pragma unroll
        for (int u = 0; u < 3; ++u) if (n + u < M) a0 = fma(wt[u], ct[u], a0);
      }
      rm = (a0 + a1) + (a2 + a3);
      m[tid] = rm;
    }
    if constexpr (decltype(tiles)::value)
#pragma unroll
    for (int q = 0; q < TPT; ++q) {
      if (own.ok[q]) {
        const double* wbase = Wl + (size_t)own.I[q] * 2;
        const double* rbase = Wl + (size_t)own.J[q] * 2;
        int n0 = 0;
        for (; n0 + 2 <= M; n0 += 2) {
          double w4[2][4], r4[2][4];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const double c = -cA[n0 + u];
            const double2* wq = reinterpret_cast<const double2*>(wbase + (size_t)(n0 + u) * 4 * M);
            const double2* rq = reinterpret_cast<const double2*>(rbase + (size_t)(n0 + u) * 4 * M);
            const double2 w01 = wq[0], w23 = wq[M], r01 = rq[0], r23 = rq[M];
            w4[u][0] = w01.x * c; w4[u][1] = w01.y * c; w4[u][2] = w23.x * c; w4[u][3] = w23.y * c;
            r4[u][0] = r01.x; r4[u][1] = r01.y; r4[u][2] = r23.x; r4[u][3] = r23.y;
          }
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j) P[q][4 * i + j] = fma(w4[u][i], r4[u][j], P[q][4 * i + j]);
        }
        if (n0 < M) {
          const double c = -cA[n0];
          double w1[4], r1[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) w1[i] = wbase[(size_t)n0 * 4 * M + (size_t)(i >> 1) * 2 * M + (i & 1)] * c;
#pragma unroll
          for (int j = 0; j < 4; ++j) r1[j] = rbase[(size_t)n0 * 4 * M + (size_t)(j >> 1) * 2 * M + (j & 1)];
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) P[q][4 * i + j] = fma(w1[i], r1[j], P[q][4 * i + j]);
        }
      }
    }
    return rm;
  };
  // per-step outputs -> ring ; covariance tiles -> HBM
  auto phase_outputs = [&](int kk, int64_t k, double rm, auto tiles) {
    if (slane) {
      rMF[(size_t)kk * S + tid] = rm;
      if (myrow == 0) rfm[kk * M + myblk] = shv[myblk] * rm;
    }
    if constexpr (decltype(tiles)::value) {
#pragma unroll
    for (int q = 0; q < TPT; ++q)
      if (own.ok[q] && own.I[q] == own.J[q])
        rfv[kk * M + own.I[q]] = shv[own.I[q]] * shv[own.I[q]] * P[q][0];
    }
    if (decltype(tiles)::value && g_PF) {
#pragma unroll
      for (int q = 0; q < TPT; ++q)
        if (own.ok[q]) pf_tile_store(g_PF + (size_t)k * pf_tiles * 16, tix + q * NTT, P[q]);
    }
  };
  auto ring_fill = [&](int64_t k0, int nb) {
    for (int i = tid; i < nb; i += NT) { ry[i] = yv[k0 + i]; rlZ[i] = g_lZ[k0 + i]; rZ[i] = -1.0; }
    for (int i = tid; i < nb * M; i += NT) {
      rtt[i] = g_tt[(size_t)k0 * M + i]; rtn[i] = g_tn[(size_t)k0 * M + i]; rR[i] = g_R[(size_t)k0 * M + i];
    }
    __syncthreads();
  };
  auto ring_flush = [&](int64_t k0, int nb) {
    for (int i = tid; i < nb; i += NT) g_lZ[k0 + i] = (rZ[i] < 0.0) ? rlZ[i] : log(rZ[i]);
    for (int i = tid; i < nb * M; i += NT) {
      g_tt[(size_t)k0 * M + i] = rtt[i]; g_tn[(size_t)k0 * M + i] = rtn[i];
      if (fp.write_R) g_R[(size_t)k0 * M + i] = rR[i];
      g_fm[(size_t)k0 * M + i] = rfm[i]; g_fv[(size_t)k0 * M + i] = rfv[i];
    }
    for (int i = tid; i < nb * S; i += NT) g_MF[(size_t)k0 * S + i] = rMF[i];
    const bool publish = fp.progress && ((k0 + nb) / fp.progress_every != k0 / fp.progress_every || k0 + nb == fp.k_end);
    if (publish) __threadfence_system();
    __syncthreads();
    if (publish && tid == 0)
      __hip_atomic_store(&fp.progress[pb], (unsigned long long)(k0 + nb), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  };

  if (wave >= MSR_W0) {
    // ================= worker role: covariance tiles + the parallel stages of the cubature
    const MspLay lay = msp_layout(CD, D, 1);
    MsrW<CD, PACK> xw;
    msr_setup_W<CD, PACK>(xw, mc, mc.sp, sW, fmu, HPH, ws);
    __syncthreads();                     // (pairs with the serial role's barrier behind its set-up)
    for (int64_t k0 = fp.k_begin; k0 < fp.k_end; k0 += KB) {
      const int nb = (fp.k_end - k0 < KB) ? (int)(fp.k_end - k0) : KB;
      ring_fill(k0, nb);
      for (int kk = 0; kk < nb; ++kk) {
        const int64_t k = k0 + kk;
        const double yk = ry[kk];
        const bool pred = (k > 0) || fp.predict_k1;
        const bool upd = !(yk != yk);
        double rm = phase_predict(pred, std::true_type{});
        lds_barrier();                   // B1: panel, fmu, HPH
        if (slane) m[tid] = rm;
        if (upd) {
          msp_qv<CD>(xw, mc);            // workers 0..2
          lds_barrier();                 // B2
          if (wave == MSR_W0 + 3 || wave == MSR_W0 + 4) msr_q0_or_s0(xw, wave == MSR_W0 + 3, ws + lay.q0, ws + lay.s0);
          lds_barrier();                 // B3
          msp_stage1b<CD>(xw, mc, mc.sp, sn2a, yk, ws);
          lds_barrier();                 // B4
          if constexpr (PACK) { if (wave >= MSR_W0 + MSR_NWK - 2) msr_marginals<CD>(xw); }
          msp_stage2<CD>(xw, mc, ws);
          lds_barrier();                 // B5
          // (moments, site update, gain coefficients: serial waves)
          if (fp.legacy_update) lds_barrier();
          lds_barrier();                 // B6: cA, cm
          rm = phase_update(rm, std::true_type{});
        }
        phase_outputs(kk, k, rm, std::true_type{});
        lds_barrier();                   // B7
      }
      ring_flush(k0, nb);
    }
    return;
  }

  // ================= serial role (waves 0 and 1): the sites (+ covariance tiles when ST)
  MsrS<CD, PACK> x;
  msr_setup_S<CD, PACK>(x, mc, mc.sp, fmu, HPH, ws);
  const int lane = tid & 63;
  const bool sub = (wave == 0) && lane < D;
  const bool act = sub || ((wave == 1) && lane < sh.N);
  const int n = (wave == 0) ? lane : D + lane;
  const int nn = act ? n : 0;
  double wrow[CD];
#pragma unroll
  for (int j = 0; j < CD; ++j) wrow[j] = sub ? sW[nn * CD + j] : 0.0;
  unsigned long long n_clamped = 0;
  __syncthreads();
  for (int64_t k0 = fp.k_begin; k0 < fp.k_end; k0 += KB) {
    const int nb = (fp.k_end - k0 < KB) ? (int)(fp.k_end - k0) : KB;
    ring_fill(k0, nb);
    for (int kk = 0; kk < nb; ++kk) {
      const int64_t k = k0 + kk;
      const double yk = ry[kk];
      const bool pred = (k > 0) || fp.predict_k1;
      const bool upd = !(yk != yk);
      double rm = phase_predict(pred, std::integral_constant<bool, ST>{});
      lds_barrier();                     // B1
      if (slane) m[tid] = rm;
      if (upd) {
        if (wave == 1) msp_link<CD>(x, mc);
        lds_barrier();                   // B2
        if (wave == 1) msp_tables<CD>(x, mc);
        lds_barrier();                   // B3
        lds_barrier();                   // B4
        lds_barrier();                   // B5
        msp_reduce<CD>(x);
        msp_wave_fence();
        if (act) {
          double Z, d1, d2;
          msp_outputs<CD>(x.accp, sub, n - D, wrow, pEP1, mc.jitter, Z, d1, d2);
          if (n == 0) rZ[kk] = Z;
          // site update (gf_ep_modulator_nmf.m:147-148), clamp (:150), R (:151)
          const double hp = HPH[n], f = fmu[n];
          const double t_old = rtt[kk * M + n], n_old = rtn[kk * M + n];
          double tnew = fp.w_old * t_old + fp.w_new * (-d2 / (1.0 + d2 * hp));
          const double nnew = fp.w_old * n_old + fp.w_new * ((d1 - f * d2) / (1.0 + d2 * hp));
          if (!(tnew > 0.0)) ++n_clamped;
          const double traw = tnew;
          tnew = max0(tnew);
          tt[n] = tnew; tn[n] = nnew;
          rtt[kk * M + n] = tnew; rtn[kk * M + n] = nnew;
          if (fp.write_R) rR[kk * M + n] = 1.0 / (fp.R_raw ? traw : tnew);
        }
        if (fp.legacy_update) lds_barrier();
        if (act) {
          const double t = tt[n], nu = tn[n], hp = HPH[n], f = fmu[n];
          bool formA = (t == 0.0);
          if (fp.legacy_update) {
            double mn = tt[0];
            for (int q = 1; q < M; ++q) mn = fmin(mn, tt[q]);   // MATLAB min ignores NaN like fmin
            formA = (mn == 0.0);
          }
          if (formA) {   // z = t*hp+1; K = W*(t/z); v = t*f - n; m -= W*(v/z); P -= K*W'
            const double z = t * hp + 1.0;
            cA[n] = t / z;
            cm[n] = -(t * f - nu) / z;
          } else {       // K = W/(hp+1/t); v = n/t - f; m += K*v; P -= K*H*P
            const double s = 1.0 / (hp + 1.0 / t);
            cA[n] = s;
            cm[n] = s * (nu / t - f);
          }
        }
        lds_barrier();                   // B6
        rm = phase_update(rm, std::integral_constant<bool, ST>{});
      } else if (tid == 0) {
        ++n_nan;
      }
      phase_outputs(kk, k, rm, std::integral_constant<bool, ST>{});
      lds_barrier();                     // B7
    }
    ring_flush(k0, nb);
  }
  if (act && n_clamped) atomicAdd(&b.counters[(size_t)pb * 4 + 1], n_clamped);
  if (tid == 0 && n_nan) atomicAdd(&b.counters[(size_t)pb * 4 + 2], n_nan);
}

}  // namespace nagp
