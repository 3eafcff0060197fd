// nagp_filter_mfma.hpp -- the fixed-site Kalman filter step on the FP64 matrix cores.
//
// Sweeps >= 2 of the Power-EP loops run the forward filter with FIXED sites for k < T-1 (gf_ep_modulator_nmf.m:126-184 with the
// `itt==1 || k==T` branch not taken): predict, then the two-branch diagonal-innovation update
//     sites with ttau == 0:  z = ttau*HPH + 1,  P -= W (ttau/z) W',   m -= W (ttau*fmu - tnu)/z
//     the others:            s = 1/(HPH + 1/ttau),  P -= W s W',      m += W s (tnu/ttau - fmu)            (:159-176)
// i.e. P -= W diag(c) W' with the M columns W = P H' -- a rank-M update, 2 M S^2 flop per step and 88 % of the step time of the
// 4x4-tile VALU kernel (gf_filter_kernel<.., MV = -1>: 741 tiles x 38 sites x 16 FMA on one wave per SIMD, LDS-bound).
// Here the covariance lives in the ACCUMULATOR layout of v_mfma_f64_16x16x4 -- lower 16x16 tiles of the padded dense matrix
// (dense index = 4*block + row), a few tiles per wave -- and the update is NTL(NTL+1)/2 x NTL matrix instructions:
//   * prediction P <- A P A' + Q with the block-diagonal A: the dense index is chosen so that the four rows of a block are the four
//     registers of a lane and its four columns four lanes of a row of 16 -- the row transform is register-local, the column
//     transform three DPP row rotations; no LDS matrix, no LDS cross-lane traffic;
//   * W = P H' (H = scaled selection: first column of every block): the owners of those elements scatter them into an LDS panel
//     laid out [row][kq][s] (site n = 4s + kq), so that a lane's operands of all NTL k-steps are contiguous;
//   * update: acc(TI,TJ) += (-c_n W[16TI+i][n]) * W[16TJ+col][n], both operands straight from the panel;
//   * the filtered covariance goes to HBM in the compact 4x4-tile layout every consumer expects (gain kernels, the ADF launch
//     that continues at k = T-1), 32-byte runs.
// Same inputs, outputs and ring / progress protocol as gf_filter_kernel; serves the plain predict-mode rule only (no nlml /
// mixture variants -- those keep the VALU kernel).  Results equal the VALU kernel's to rounding (different summation order).
#pragma once
#include "nagp_kernels.hpp"
#include "nagp_mfma.hpp"

namespace nagp {

__host__ __device__ inline int flm_nsp(int NTL) { return (NTL + 1) & ~1; }                 // k-steps per row and kq, even
__host__ __device__ inline int flm_rs(int NTL) { return 4 * flm_nsp(NTL) + 2; }           // panel row stride (doubles)
__host__ __device__ inline size_t flm_lds_doubles(const Shape& s, int NTL, int kb) {
  const size_t Sp = 16 * (size_t)NTL;
  return 2 * (size_t)(s.M + 1) * 16 + s.M + Sp + 2 + (size_t)Sp * flm_rs(NTL) + 4 * 4 * (size_t)NTL + 16 +
         (size_t)kb * (1 + 2 * s.M) + (size_t)kb * (s.S + 2 * s.M) + 2 * MAXM + 8;
}

// 64-bit value of the lane `rot` places away in the same row of 16 lanes (DPP row_ror: a VALU move, no LDS traffic)
template <int CTRL>
__device__ __forceinline__ double flm_dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

// INTERNAL dense index of state (block beta, row rho): d = 16*(beta>>2) + 4*rho + (beta&3).  With it the four rows of a block are
// the four REGISTERS of one lane of an accumulator tile (row 4t+kq: rho = t, beta = 4*TI + kq) and its four columns are the lanes
// i, i+4, i+8, i+12 of a row of 16 (col i: rho = i>>2, beta = 4*TJ + (i&3)): the row transform of the prediction is register-local,
// the column transform three DPP row rotations.
template <int NTL, int NW>
__global__ void __launch_bounds__(64 * NW) gf_filter_lin_mfma_kernel(Shape sh, Bufs b, FilterPar fp) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int Sp = 16 * NTL, NT = 64 * NW, NLT = NTL * (NTL + 1) / 2, TPW = (NLT + NW - 1) / NW;
  constexpr int NSP = (NTL + 1) & ~1, RS = 4 * NSP + 2, KP = 4 * NTL;
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, kq = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = sh.S, M = sh.M, KB = fp.kb;
  const int64_t T = sh.T;
  const int pb = blockIdx.x;
  const double* mdl = b.model + (size_t)pb * mdl_size(sh);

  double* sA = lds;                          // [M][16] (+ one zero tile for the padding blocks)
  double* sQ = sA + (size_t)(M + 1) * 16;    // [M][16] (+ zero tile)
  double* shv = sQ + (size_t)(M + 1) * 16;   // [M]
  double* m = shv + M;                       // [Sp] state, internal index
  double* Wp = m + Sp + 2;                   // [Sp][RS] panel: h_n P[:, c_n] at [row][n&3][n>>2]
  double* cA = Wp + (size_t)Sp * RS;         // [KP] (zero beyond M)
  double* cm = cA + KP;                      // [KP]
  double* HPH = cm + KP;                     // [KP]
  double* fmu = HPH + KP;                    // [KP]
  double* ry = fmu + KP + 16;                // ring in: y[KB], ttau[KB][M], tnu[KB][M]
  double* rtt = ry + KB;
  double* rtn = rtt + (size_t)KB * M;
  double* rMF = rtn + (size_t)KB * M;        // ring out: m [KB][S], H m [KB][M], diag(H P H') [KB][M]
  double* rfm = rMF + (size_t)KB * S;
  double* rfv = rfm + (size_t)KB * M;
  int* ioff = reinterpret_cast<int*>(rfv + (size_t)KB * M);   // [MAXM+1]
  int* ibsz = ioff + MAXM + 1;                                 // [MAXM]

  for (int q = tid; q <= M; q += NT) ioff[q] = sh.off[q];
  for (int q = tid; q < M; q += NT) { ibsz[q] = sh.bsz[q]; shv[q] = mdl[mdl_h(sh) + q]; }
  for (int q = tid; q < (M + 1) * 16; q += NT) { sA[q] = (q < M * 16) ? mdl[mdl_A(sh) + q] : 0.0; sQ[q] = (q < M * 16) ? mdl[mdl_Q(sh) + q] : 0.0; }
  for (int q = tid; q < Sp * RS; q += NT) Wp[q] = 0.0;
  for (int q = tid; q < 4 * KP; q += NT) cA[q] = 0.0;
  __syncthreads();

  // ---- this wave's tiles: lower tiles in row-major order, dealt round robin
  int TI[TPW], TJ[TPW]; bool tok[TPW];
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int lt = wv + q * NW;
    tok[q] = lt < NLT;
    const int l2 = tok[q] ? lt : 0;
    int I = 0;
    while ((I + 1) * (I + 2) / 2 <= l2) ++I;
    TI[q] = I; TJ[q] = l2 - I * (I + 1) / 2;
  }
  // which rho'' the rotations deliver (the direction of row_ror is read off the lane ids themselves)
  const int rc = i >> 2;                                       // rho of this lane's column
  const int rs4 = (__builtin_amdgcn_update_dpp(0, i, 0x124, 0xF, 0xF, false) >> 2) & 3;
  const int rs8 = (__builtin_amdgcn_update_dpp(0, i, 0x128, 0xF, 0xF, false) >> 2) & 3;
  const int rs12 = (__builtin_amdgcn_update_dpp(0, i, 0x12C, 0xF, 0xF, false) >> 2) & 3;

  // ---- initial state
  v4d P[TPW];
  const int64_t kb0 = fp.k_begin;
  const double* PF0 = (kb0 > 0) ? b.PF + ((size_t)pb * T + (kb0 - 1)) * pf_step_doubles(sh) : nullptr;
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int br = 4 * TI[q] + kq, bc = 4 * TJ[q] + (i & 3);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      double v = 0.0;
      if (tok[q] && br < M && bc < M) {
        if (PF0) v = pf_elem(PF0, br, bc, t, rc);
        else if (br == bc) v = mdl[mdl_P(sh) + (size_t)br * 16 + 4 * t + rc];
      }
      P[q][t] = v;
    }
  }
  // vector threads: internal index d = tid < Sp
  const int vb = 4 * (tid >> 4) + (tid & 3), vr = (tid >> 2) & 3;
  const bool vin = tid < Sp && vb < M;
  const bool vok = vin && vr < ibsz[vin ? vb : 0];
  if (tid < Sp) m[tid] = (vok && kb0 > 0) ? b.MF[((size_t)pb * T + (kb0 - 1)) * S + ioff[vb] + vr] : 0.0;
  __syncthreads();

  const double* yv = b.y + (size_t)pb * T;
  const double* g_tt = b.ttau + (size_t)pb * T * M;
  const double* g_tn = b.tnu + (size_t)pb * T * M;
  double* g_MF = b.MF + (size_t)pb * T * S;
  double* g_fm = b.fm + (size_t)pb * T * M;
  double* g_fv = b.fv + (size_t)pb * T * M;
  double* g_PF = (b.PF && fp.store_PF) ? b.PF + (size_t)pb * T * pf_step_doubles(sh) : nullptr;
  unsigned long long n_nan = 0;

  for (int64_t k0 = fp.k_begin; k0 < fp.k_end; k0 += KB) {
    const int nb = (fp.k_end - k0 < KB) ? (int)(fp.k_end - k0) : KB;
    for (int q = tid; q < nb; q += NT) ry[q] = yv[k0 + q];
    for (int q = tid; q < nb * M; q += NT) { rtt[q] = g_tt[(size_t)k0 * M + q]; rtn[q] = g_tn[(size_t)k0 * M + q]; }
    __syncthreads();
    for (int kk = 0; kk < nb; ++kk) {
      const int64_t k = k0 + kk;
      const double yk = ry[kk];
      const bool pred = (k > 0) || fp.predict_k1;
      const bool upd = !(yk != yk);
      // ---- prediction: mean (vector threads), covariance (registers + DPP), panel W = P H', H P H', H m
      double mpv = 0.0;
      if (tid < Sp) {
        if (pred && vin) {
          const double* a = sA + (size_t)vb * 16 + 4 * vr;
          const int d0 = tid & ~12;                              // state (vb, 0)
#pragma unroll
          for (int l = 0; l < 4; ++l) mpv = fma(a[l], m[d0 + 4 * l], mpv);
        } else {
          mpv = m[tid];
        }
        if (vr == 0 && vin) fmu[vb] = shv[vb] * mpv;
      }
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
        if (!tok[q]) continue;
        const int br = 4 * TI[q] + kq, bc = 4 * TJ[q] + (i & 3);
        const int brc = br < M ? br : M, bcc = bc < M ? bc : M;   // padding blocks read the zero tile
        if (pred) {
          const double* ar = sA + (size_t)brc * 16;
          double x[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const double2 a01 = *reinterpret_cast<const double2*>(ar + 4 * t), a23 = *reinterpret_cast<const double2*>(ar + 4 * t + 2);
            x[t] = fma(a01.x, P[q][0], fma(a01.y, P[q][1], fma(a23.x, P[q][2], a23.y * P[q][3])));
          }
          const double* ac = sA + (size_t)bcc * 16 + 4 * rc;
          const double c0 = ac[rc], c4 = ac[rs4], c8 = ac[rs8], c12 = ac[rs12];
          const bool dg = (br == bc) && br < M;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            double y2 = x[t] * c0;
            y2 = fma(flm_dpp<0x124>(x[t]), c4, y2);
            y2 = fma(flm_dpp<0x128>(x[t]), c8, y2);
            y2 = fma(flm_dpp<0x12C>(x[t]), c12, y2);
            if (dg) y2 += sQ[(size_t)br * 16 + 4 * t + rc];
            P[q][t] = y2;
          }
        }
        // panel / marginals from the lower triangle only (tile order; inside a diagonal tile the internal index)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int dr = 16 * TI[q] + 4 * t + kq, dc = 16 * TJ[q] + i;
          if (dr >= dc) {
            const double y2 = P[q][t];
            if (rc == 0 && bc < M) Wp[(size_t)dr * RS + (bc & 3) * NSP + (bc >> 2)] = shv[bc] * y2;
            if (t == 0 && br < M && dr > dc) Wp[(size_t)dc * RS + (br & 3) * NSP + (br >> 2)] = shv[br] * y2;
            if (dr == dc && t == 0 && br < M) HPH[br] = shv[br] * shv[br] * y2;
          }
        }
      }
      lds_barrier();   // B1: panel, HPH, fmu
      if (upd) {
        if (tid < M) {
          const double t = rtt[kk * M + tid], n_ = rtn[kk * M + tid], hp = HPH[tid], f = fmu[tid];
          if (t == 0.0) { const double z = t * hp + 1.0; cA[tid] = t / z; cm[tid] = -(t * f - n_) / z; }
          else { const double s = 1.0 / (hp + 1.0 / t); cA[tid] = s; cm[tid] = s * (n_ / t - f); }
        }
        lds_barrier(); // B2: coefficients
        if (tid < Sp) {
          double a0 = mpv, a1 = 0.0;
          const double* wr = Wp + (size_t)tid * RS;
          for (int n = 0; n + 1 < M; n += 2) {
            a0 = fma(wr[(n & 3) * NSP + (n >> 2)], cm[n], a0);
            a1 = fma(wr[((n + 1) & 3) * NSP + ((n + 1) >> 2)], cm[n + 1], a1);
          }
          if (M & 1) a0 = fma(wr[((M - 1) & 3) * NSP + ((M - 1) >> 2)], cm[M - 1], a0);
          mpv = a0 + a1;
        }
        // P -= W diag(cA) W'
        double cs[NTL];
#pragma unroll
        for (int s = 0; s < NTL; ++s) cs[s] = -cA[4 * s + kq];
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
          if (!tok[q]) continue;
          const double* pa = Wp + (size_t)(16 * TI[q] + i) * RS + kq * NSP;
          const double* pbb = Wp + (size_t)(16 * TJ[q] + i) * RS + kq * NSP;
          v4d acc = P[q];
#pragma unroll
          for (int s = 0; s + 1 < NTL; s += 2) {
            const double2 a2 = *reinterpret_cast<const double2*>(pa + s), b2 = *reinterpret_cast<const double2*>(pbb + s);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(cs[s] * a2.x, b2.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(cs[s + 1] * a2.y, b2.y, acc, 0, 0, 0);
          }
          if (NTL & 1) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(cs[NTL - 1] * pa[NTL - 1], pbb[NTL - 1], acc, 0, 0, 0);
          P[q] = acc;
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        ++n_nan;
      }
      // ---- per-step outputs -> ring ; covariance tiles -> HBM
      if (tid < Sp) {
        m[tid] = mpv;
        if (vok) rMF[(size_t)kk * S + ioff[vb] + vr] = mpv;
        if (vr == 0 && vin) rfm[kk * M + vb] = shv[vb] * mpv;
      }
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
        if (!tok[q]) continue;
        const int br = 4 * TI[q] + kq, bc = 4 * TJ[q] + (i & 3);
        if (br < M && bc < M && br >= bc) {
          if (g_PF) {
            double* dst = g_PF + (size_t)k * pf_step_doubles(sh);      // element x of tile (br, bc) at pf_off(tile, x)
            const int pft = br * (br + 1) / 2 + bc;
#pragma unroll
            for (int t = 0; t < 4; ++t) dst[pf_off(pft, rc + 4 * t)] = P[q][t];
          }
          if (br == bc && rc == 0) rfv[kk * M + br] = shv[br] * shv[br] * P[q][0];
        }
      }
      lds_barrier();   // B3: the panel may be overwritten, m is final
    }
    // ---- flush the ring
    for (int q = tid; q < nb * M; q += NT) { g_fm[(size_t)k0 * M + q] = rfm[q]; g_fv[(size_t)k0 * M + q] = rfv[q]; }
    for (int q = tid; q < nb * S; q += NT) g_MF[(size_t)k0 * S + q] = rMF[q];
    const bool publish = fp.progress && ((k0 + nb) / fp.progress_every != k0 / fp.progress_every || k0 + nb == fp.k_end);
    if (publish) __threadfence_system();
    __syncthreads();
    if (publish && tid == 0)
      __hip_atomic_store(&fp.progress[pb], (unsigned long long)(k0 + nb), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (tid == 0 && n_nan) atomicAdd(&b.counters[(size_t)pb * 4 + 2], n_nan);
}

}  // namespace nagp
