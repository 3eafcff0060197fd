// nagp_gain_mfma.hpp -- RTS gain on the FP64 matrix cores (dense Sp x Sp output, the input layout of the MFMA smoother passes).
//
//   B = PS_k A' ; PSkp = A B + Q ; L = chol(PSkp,'lower') (jitter retry) ; G = B / L' / L ; Delta_k = PS_{k+1} - PSkp ;
//   delta_k = MF_{k+1} - A MF_k                                                        (gf_ep_modulator_nmf.m:210-230)
//
// One workgroup of NTL = Sp/16 waves per (step, problem); the work is organised on 16x16 tiles of the padded dense matrices
// (dense index = 4*block + row) instead of the 4x4 tiles of rts_gain_kernel, whose ~3M barrier-separated phases per step
// (114 at 38 sites) left the CUs waiting:
//   * build: wave J forms tile column J of B' = A PS_k (accumulator layout, in registers for the rest of the kernel) and the
//     lower tiles (I >= J) of PSkp = B' A' + Q, which go to LDS; Delta goes straight to HBM;
//   * factorisation: right-looking on 16x16 tiles, NTL block columns.  Wave j factors the diagonal tile (4x4 sub-tiles, its
//     own LDS traffic only, no workgroup barrier) and inverts it; the panel L_Ij = A_Ij inv(L_jj)' and the trailing update
//     A_IK -= L_Ij L_Kj' are 16x16x16 products on v_mfma_f64_16x16x4 with both operands read from LDS in the orientation
//     they are stored in (rows of the left factor, rows of the transposed right factor);
//   * solves: G' = L'^-1 (L^-1 B').  The right-hand-side columns are independent, so wave J solves ITS tile column with no
//     synchronisation at all: forward  Y_I = inv(L_II) (B'_I - sum_{K<I} L_IK Y_K), backward W_I = inv(L_II)' (Y_I - sum_{K>I}
//     L_KI' W_K).  The accumulator layout of the MFMA (lane = col + 16*kq, register t <-> row 4t+kq) IS its B-operand layout,
//     so Y_K, W_K are multiplied from the registers they were accumulated in; only the L tiles travel (LDS -> A operand);
//   * G = W' is written to HBM as 32-byte runs.
// LDS: NTL(NTL+1)/2 + NTL tiles of 2 KiB (130 KiB at Sp = 160).  Columns of a tile are stored permuted (column 4s+kq at
// position 4kq+s) so that the four k-steps of a lane's A operand are 32 contiguous bytes.
#pragma once
#include "nagp_mfma.hpp"

namespace nagp {

__host__ __device__ inline size_t gainm_lds_doubles(int NTL) {
  return (size_t)(NTL * (NTL + 1) / 2 + NTL) * 256 + (size_t)MAXM * 16 + 64 + 16;
}
__device__ __forceinline__ int gm_p(int c) { return ((c & 3) << 2) + (c >> 2); }                  // stored position of column c
__device__ __forceinline__ int gm_at(int r, int c) { return r * 16 + gm_p(c); }
__device__ __forceinline__ int gm_tix(int K, int L) { return (K * (K + 1) / 2 + L) * 256; }      // K >= L

// acc += sgn * X * Yt'  with X, Yt 16x16 tiles in LDS (rows of X, rows of Yt = columns of Yt'): both "direct" reads
__device__ __forceinline__ v4d gm_mma_xyT(const double* X, const double* Yt, int i, int kq, v4d acc, bool neg) {
  const double2 a0 = *reinterpret_cast<const double2*>(X + i * 16 + 4 * kq), a1 = *reinterpret_cast<const double2*>(X + i * 16 + 4 * kq + 2);
  const double2 b0 = *reinterpret_cast<const double2*>(Yt + i * 16 + 4 * kq), b1 = *reinterpret_cast<const double2*>(Yt + i * 16 + 4 * kq + 2);
  const double sg = neg ? -1.0 : 1.0;
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sg * a0.x, b0.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sg * a0.y, b0.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sg * a1.x, b1.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sg * a1.y, b1.y, acc, 0, 0, 0);
  return acc;
}
// acc += sgn * X * Bacc  (X in LDS, rows direct; Bacc an accumulator = B operand)
__device__ __forceinline__ v4d gm_mma_xb(const double* X, v4d Bacc, int i, int kq, v4d acc, bool neg) {
  const double2 a0 = *reinterpret_cast<const double2*>(X + i * 16 + 4 * kq), a1 = *reinterpret_cast<const double2*>(X + i * 16 + 4 * kq + 2);
  const double sg = neg ? -1.0 : 1.0;
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sg * a0.x, Bacc[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sg * a0.y, Bacc[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sg * a1.x, Bacc[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sg * a1.y, Bacc[3], acc, 0, 0, 0);
  return acc;
}
// acc += sgn * X' * Bacc  (transposed reads of X)
__device__ __forceinline__ v4d gm_mma_xTb(const double* X, v4d Bacc, int i, int kq, v4d acc, bool neg) {
  const int pi = gm_p(i);
  const double sg = neg ? -1.0 : 1.0;
#pragma unroll
  for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sg * X[(4 * s + kq) * 16 + pi], Bacc[s], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ void gm_store_acc(double* Tl, v4d v, int i, int kq) {
  const int pi = gm_p(i);
#pragma unroll
  for (int t = 0; t < 4; ++t) Tl[(4 * t + kq) * 16 + pi] = v[t];
}
__device__ __forceinline__ v4d gm_load_acc(const double* Tl, int i, int kq) {
  const int pi = gm_p(i);
  v4d v;
#pragma unroll
  for (int t = 0; t < 4; ++t) v[t] = Tl[(4 * t + kq) * 16 + pi];
  return v;
}
__device__ __forceinline__ void gm_wave_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Cholesky of the 16x16 tile T (LDS, lower part meaningful) in place and its inverse into Ti, by ONE wave: lanes 0..15 own the
// 4x4 sub-tiles (a = lane>>2, b = lane&3); four block columns, the wave's own LDS traffic orders the phases.  Returns false
// (to every lane) when a pivot was not positive.
__device__ __attribute__((noinline)) bool gm_chol16(double* T, double* Ti, double* rdv, int lane) {
  const int a = (lane >> 2) & 3, b = lane & 3;
  const bool on = lane < 16 && a >= b;
  bool ok = true;
  auto ld = [&](const double* base, int ta, int tb, double* t) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) t[4 * r + c] = base[gm_at(4 * ta + r, 4 * tb + c)];
  };
  auto st = [&](double* base, int ta, int tb, const double* t) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) base[gm_at(4 * ta + r, 4 * tb + c)] = t[4 * r + c];
  };
  for (int bb = 0; bb < 4; ++bb) {
    if (on && a == bb && b == bb) {
      double t[16], rd[4];
      ld(T, bb, bb, t);
      if (!tile_chol(t, 4, rd)) ok = false;
      st(T, bb, bb, t);
#pragma unroll
      for (int q = 0; q < 4; ++q) rdv[4 * bb + q] = rd[q];      // 1 / L(q,q): the solves and the inverse multiply by it
    }
    gm_wave_fence();
    if (on && b == bb && a > bb) {          // X L_bb' = T_ab
      double t[16], l[16];
      ld(T, a, bb, t); ld(T, bb, bb, l);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double v = t[4 * r + c];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (q < c) v = fma(-t[4 * r + q], l[4 * c + q], v);
          t[4 * r + c] = v * rdv[4 * bb + c];
        }
      st(T, a, bb, t);
    }
    gm_wave_fence();
    if (on && b > bb) {                     // trailing update T_ab -= X_a X_b'
      double t[16], xa[16], xb[16];
      ld(T, a, b, t); ld(T, a, bb, xa); ld(T, b, bb, xb);
      tile_mms_nt(t, xa, xb);
      st(T, a, b, t);
    }
    gm_wave_fence();
  }
  // zero the strictly upper sub-tiles of L (the panel products read whole rows)
  if (lane < 16 && a < b) { double z[16]; tile_zero(z); st(T, a, b, z); }
  // inverse, diagonal sub-tiles first, then the sub-diagonals: X_ab = -X_aa sum_{c=b}^{a-1} L_ac X_cb
  for (int dl = 0; dl < 4; ++dl) {
    if (on && a - b == dl) {
      double x[16];
      tile_zero(x);
      if (dl == 0) {
        double l[16];
        ld(T, a, a, l);
#pragma unroll
        for (int c = 0; c < 4; ++c) {       // column c of inv(L_aa) by forward substitution
          x[4 * c + c] = rdv[4 * a + c];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (r > c) {
              double v = 0.0;
#pragma unroll
              for (int q = 0; q < 4; ++q)
                if (q >= c && q < r) v = fma(-l[4 * r + q], x[4 * q + c], v);
              x[4 * r + c] = v * rdv[4 * a + r];
            }
        }
      } else {
        double acc[16], xaa[16];
        tile_zero(acc);
        for (int c = b; c < a; ++c) {
          double l[16], xc[16];
          ld(T, a, c, l); ld(Ti, c, b, xc);
          tile_mma(acc, l, xc);
        }
        ld(Ti, a, a, xaa);
        tile_mma(x, xaa, acc);
#pragma unroll
        for (int e = 0; e < 16; ++e) x[e] = -x[e];
      }
      st(Ti, a, b, x);
    }
    if (dl == 0 && lane < 16 && a < b) { double z[16]; tile_zero(z); st(Ti, a, b, z); }
    gm_wave_fence();
  }
  const unsigned long long bad = __ballot(!ok);
  return bad == 0ull;
}

template <int NTL>
__global__ void __launch_bounds__(64 * NTL) rts_gain_mfma_kernel(Shape sh, Bufs b, GainPar gp) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int Sp = 16 * NTL, NT = 64 * NTL;
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, kq = lane >> 4;
  const int J = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = sh.S, M = sh.M;
  const int64_t T = sh.T;
  const int kk = blockIdx.x, pb = blockIdx.y;
  const int64_t k = gp.k0 + kk;
  const double* mdl = b.model + (size_t)pb * mdl_size(sh);
  double* Lt = lds;                                             // lower tiles of PSkp -> L
  double* Li = Lt + (size_t)(NTL * (NTL + 1) / 2) * 256;        // inverses of the diagonal tiles
  double* sA = Li + (size_t)NTL * 256;                          // [M][16]
  int* ibsz = reinterpret_cast<int*>(sA + (size_t)MAXM * 16);   // [MAXM]
  int* flag = ibsz + MAXM + 2;                                  // [2]
  double* rdv = sA + (size_t)MAXM * 16 + 48;                    // [16] reciprocal pivots of the diagonal tile being factored
  for (int q = tid; q < M * 16; q += NT) sA[q] = mdl[mdl_A(sh) + q];
  for (int q = tid; q < M; q += NT) ibsz[q] = sh.bsz[q];
  if (tid == 0) { flag[0] = 0; flag[1] = 0; }
  __syncthreads();

  const double* PFk = b.PF + ((size_t)pb * T + k) * pf_step_doubles(sh);
  const double* PFk1 = PFk + pf_step_doubles(sh);
  const size_t SS = (size_t)Sp * Sp;
  double* Gout = b.Gbuf + (((size_t)pb * gp.chunk + kk) * 2) * SS;
  double* Dout = Gout + SS;

  // ---- delta_k = MF_{k+1} - A MF_k
  if (tid < S) {
    int blk = 0;
    while (sh.off[blk + 1] <= tid) ++blk;
    const int row = tid - sh.off[blk];
    const double* mf = b.MF + ((size_t)pb * T + k) * S;
    double acc = mf[S + tid];
    for (int l = 0; l < ibsz[blk]; ++l) acc = fma(-sA[(size_t)blk * 16 + 4 * row + l], mf[sh.off[blk] + l], acc);
    b.dbuf[((size_t)pb * gp.chunk + kk) * S + tid] = acc;
  }

  // ---- element (16I + 4t + kq, 16J + i) of B' = A PS_k: row kq of block br = 4I+t, column ci of block bc.  The four lanes kq of a
  // column need the same four entries PS[(br, l)][(bc, ci)], l = 0..3: every lane loads the one with l = kq, the others arrive by
  // cross-lane reads -- a quarter of the loads
  const int bc = 4 * J + (i >> 2), ci = i & 3;                  // block / column-in-block of this lane's column
  const bool colok = bc < M && ci < ibsz[bc < M ? bc : 0];
  auto bprime = [&](int br) -> double {
    const bool rowok = br < M && kq < ibsz[br < M ? br : 0];
    double mine = 0.0;
    if (rowok && colok) mine = pf_elem(PFk, br, bc, kq, ci);
    double v = 0.0;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const double pl = __shfl(mine, i + 16 * l, 64);
      if (br < M) v = fma(sA[(size_t)br * 16 + 4 * kq + l], pl, v);
    }
    return (rowok && colok) ? v : 0.0;
  };

  bool failed = false;
  for (int attempt = 0; attempt < 2; ++attempt) {
    // ---- PSkp = B' A' + Q (+ jitter): the whole tile column J (Delta = PS_{k+1} - PSkp leaves as 128-byte row runs, first attempt
    // only); the lower tiles stay in LDS for the factorisation
#pragma unroll 1
    for (int I = 0; I < NTL; ++I) {
      if (attempt == 1 && I < J) continue;
      v4d ps;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int br = 4 * I + t;
        const double bp = (gp.dbg & 32) ? 0.0 : bprime(br);
        double v = 0.0;
        // sum_l B'[r][(bc,l)] A_bc[ci][l]: the four columns of block bc sit in the four lanes of this lane's quad
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          const double bl = __shfl(bp, (lane & ~3) | l, 64);
          if (bc < M) v = fma(bl, sA[(size_t)bc * 16 + 4 * ci + l], v);
        }
        const bool rowok = br < M && kq < ibsz[br < M ? br : 0];
        if (rowok && colok && br == bc) {
          v += mdl[mdl_Q(sh) + (size_t)br * 16 + 4 * kq + ci];
          if (attempt == 1 && kq == ci) v += 0.01 * 0.5;        // sqrt(1e-4)*diag(rand): deterministic 0.5 in place of rand
        }
        if (!(rowok && colok)) v = (16 * I + 4 * t + kq == 16 * J + i) ? 1.0 : 0.0;      // padding: identity
        ps[t] = v;
        if (attempt == 0 && !(gp.dbg & 8)) {
          double d = 0.0;
          if (rowok && colok) d = pf_elem(PFk1, br, bc, kq, ci) - v;
          Dout[(16 * I + 4 * t + kq) * Sp + 16 * J + i] = d;
        }
      }
      if (I >= J) gm_store_acc(Lt + gm_tix(I, J), ps, i, kq);
    }
    __syncthreads();
    // ---- right-looking Cholesky on 16x16 tiles
    for (int j = 0; j < NTL; ++j) {
      if (J == j) {
        if (!(gp.dbg & 1) && !gm_chol16(Lt + gm_tix(j, j), Li + (size_t)j * 256, rdv, lane)) { if (lane == 0) flag[attempt] = 1; }
      }
      __syncthreads();
      // panel: L_Ij = A_Ij inv(L_jj)'  (wave I, I > j)
      if (J > j) {
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        acc = gm_mma_xyT(Lt + gm_tix(J, j), Li + (size_t)j * 256, i, kq, acc, false);
        gm_store_acc(Lt + gm_tix(J, j), acc, i, kq);
      }
      __syncthreads();
      // trailing update: A_IK -= L_Ij L_Kj', j < K <= I, tiles dealt round robin over the waves
      if (!(gp.dbg & 4)) {
        int cnt = 0;
        for (int I = j + 1; I < NTL; ++I)
          for (int K = j + 1; K <= I; ++K, ++cnt) {
            if (cnt % NTL != J) continue;
            v4d acc = gm_load_acc(Lt + gm_tix(I, K), i, kq);
            acc = gm_mma_xyT(Lt + gm_tix(I, j), Lt + gm_tix(K, j), i, kq, acc, true);
            gm_store_acc(Lt + gm_tix(I, K), acc, i, kq);
          }
      }
      __syncthreads();
    }
    failed = (flag[attempt] != 0);
    if (!failed) break;
    __syncthreads();
  }
  if (tid == 0) {
    if (flag[0]) atomicAdd(&b.counters[(size_t)pb * 4 + 0], 1ull);
    if (flag[0] && flag[1]) atomicAdd(&b.counters[(size_t)pb * 4 + 3], 1ull);
  }

  // ---- B' again, now into the registers it stays in through both solves (PS_k comes from L2 this time)
  v4d R[NTL];
#pragma unroll
  for (int I = 0; I < NTL; ++I) {
#pragma unroll
    for (int t = 0; t < 4; ++t) R[I][t] = (gp.dbg & 64) ? 1.0 : bprime(4 * I + t);
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- forward: Y_I = inv(L_II) (B'_I - sum_{K<I} L_IK Y_K)   (tile column J, no synchronisation)
  if (!(gp.dbg & 2)) {
#pragma unroll
    for (int I = 0; I < NTL; ++I) {
      v4d acc = R[I];
#pragma unroll
      for (int K = 0; K < NTL; ++K)
        if (K < I) { acc = gm_mma_xb(Lt + gm_tix(I, K), R[K], i, kq, acc, true); if (K & 1) __builtin_amdgcn_sched_barrier(0); }
      v4d y = {0.0, 0.0, 0.0, 0.0};
      R[I] = gm_mma_xb(Li + (size_t)I * 256, acc, i, kq, y, false);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- backward: W_I = inv(L_II)' (Y_I - sum_{K>I} L_KI' W_K)
#pragma unroll
    for (int I = NTL - 1; I >= 0; --I) {
      v4d acc = R[I];
#pragma unroll
      for (int K = 0; K < NTL; ++K)
        if (K > I) { acc = gm_mma_xTb(Lt + gm_tix(K, I), R[K], i, kq, acc, true); if (K & 1) __builtin_amdgcn_sched_barrier(0); }
      v4d w = {0.0, 0.0, 0.0, 0.0};
      R[I] = gm_mma_xTb(Li + (size_t)I * 256, acc, i, kq, w, false);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- G = W': G[16J + i][16I + 4t + kq] = W_I[4t+kq][i] (32-byte runs); padding rows / columns cleaned
  const int gbase = (16 * J + i) * Sp + kq;
#pragma unroll
  for (int I = 0; I < NTL; ++I) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int br = 4 * I + t;
      const bool rowok = br < M && kq < ibsz[br < M ? br : 0];
      if (!(gp.dbg & 16)) Gout[gbase + 16 * I + 4 * t] = (rowok && colok) ? R[I][t] : 0.0;
    }
  }
}

}  // namespace nagp
