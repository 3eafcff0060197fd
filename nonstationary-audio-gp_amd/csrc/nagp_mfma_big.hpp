// nagp_mfma_big.hpp -- FP64 MFMA versions of the parallel-in-time RTS smoother passes for padded state dimensions
// 96 < Sp <= 160 (25 .. 40 sites: the 32-channel / 6-component configurations), where neither G_k nor the recursion
// state fits in LDS next to the other.  Same maps as nagp_mfma.hpp (gf_ep_modulator_nmf.m:229-234 as affine maps
// E <- G (E + Delta) G'), same dense row-major Sp x Sp buffers, different residence:
//
//   * one workgroup of NTL = Sp/16 waves per span; wave J owns tile COLUMN J;
//   * the symmetric state Y = E + Delta lives in LDS as its lower-triangular 16x16 tiles (XOR-swizzled rows);
//   * phase 1: Z = Y G'.  Column J of Z needs rows 16J..16J+15 of G only: the B operand comes straight from global
//     memory (4 contiguous doubles per lane and k-tile: the k index inside a k-tile is taken as 4*(lane>>4)+step, which
//     both operands are free to agree on); the A operand is a tile of Y or the transpose of its mirror tile;
//   * phase 2: E' = G Z.  The accumulator layout of v_mfma_f64_16x16x4 (lane = col + 16*kq, register r <-> row 4r+kq)
//     IS the B-operand layout of k-step r, so column J of Z is multiplied from the registers it was accumulated in;
//     G streams through LDS as 16-row panels (double buffered, columns permuted so that a lane's four k-steps are
//     contiguous).  By symmetry only NTL(NTL+1)/2 tiles are formed: wave J takes the tiles (I,J) with
//     (I-J) mod NTL <= (NTL-1)/2 (and the antipodal ones for J < NTL/2) and stores those above the diagonal transposed;
//   * the mean recursion e <- G (e + delta) rides on the panels: the wave after the active ones multiplies each panel
//     with the vector while the others run the matrix cores.
// The Phi chain of the compose pass (Phi <- G Phi, no symmetry) is its own kernel: column J of Phi in registers,
// the upper half of the new column parked in LDS until the old one is dead.
#pragma once
#include "nagp_mfma.hpp"

namespace nagp {

constexpr int BIG_PLD = 162;    // row stride of a G panel in LDS (doubles): 16-byte aligned rows, 4 banks apart

__host__ __device__ inline size_t big_lds_doubles(int NTL) {
  return (size_t)(NTL * (NTL + 1) / 2) * 256 + 2 * 16 * BIG_PLD + 2 * 16 * (size_t)NTL + MAXM + 8;
}
// element (r, c) of a 16x16 tile of Y: rows 16 doubles apart, 16-byte chunks XOR-swizzled with the row pair
__device__ __forceinline__ int big_phys(int r, int c) { return r * 16 + ((((c >> 1) ^ (r >> 1)) & 7) << 1) + (c & 1); }
__device__ __forceinline__ int big_tix(int K, int L) { return (K * (K + 1) / 2 + L) * 256; }   // K >= L

// Five tile columns (Sp = 80, 74 KB of LDS): capped at 128 registers (up to 34 spilled) so that TWO workgroups share a CU -- five waves put two on the first
// SIMD, and at 148 .. 170 registers (three waves per SIMD) the second workgroup is never placed there.  With the span count of a chunk chosen for 512 resident
// workgroups (nagp_api_sweep.hpp, SM_BIG) the span passes of cfg2_batch take 243 -> 180 ms; either change alone: nothing (profiles/r05_gain_mfma.txt).
#define BIG_REG_CAP(NTL) __attribute__((amdgpu_waves_per_eu((NTL) <= 5 ? 4 : 1, (NTL) <= 5 ? 4 : 8)))

template <int NTL>
struct BigCtx {
  int tid, wave, lane, i, kq;
  double* Y;       // lower tiles of the symmetric state
  double* P;       // two panels of 16 x BIG_PLD
  double* vp;      // e + delta, permuted like the panel columns
  double* en;      // G (e + delta)
  int offD0, offD1, offT[4];   // lane offsets inside a tile of Y: direct (two 16-byte reads) and transposed A operand
};

template <int NTL>
__device__ __forceinline__ void big_ctx_init(BigCtx<NTL>& c, double* lds) {
  c.tid = threadIdx.x; c.wave = __builtin_amdgcn_readfirstlane(c.tid >> 6); c.lane = c.tid & 63; c.i = c.lane & 15; c.kq = c.lane >> 4;
  c.Y = lds;
  c.P = c.Y + (size_t)(NTL * (NTL + 1) / 2) * 256;
  c.vp = c.P + 2 * 16 * BIG_PLD;
  c.en = c.vp + 16 * NTL;
  c.offD0 = c.i * 16 + ((((2 * c.kq) ^ (c.i >> 1)) & 7) << 1);
  c.offD1 = c.i * 16 + ((((2 * c.kq + 1) ^ (c.i >> 1)) & 7) << 1);
#pragma unroll
  for (int s = 0; s < 4; ++s) c.offT[s] = big_phys(4 * c.kq + s, c.i);
}

// does wave J form tile (I, J)?
template <int NTL>
__device__ __forceinline__ bool big_active(int I, int J) {
  const int d = (I - J + NTL) % NTL;
  if (d <= (NTL - 1) / 2) return true;
  return (NTL % 2 == 0) && d == NTL / 2 && J < NTL / 2;
}

// rows 16I .. 16I+15 of a dense matrix -> registers (4 consecutive columns per thread), then -> panel with the columns
// of every 16-block permuted (column 4s+kq at position 4kq+s)
template <int NTL>
__device__ __forceinline__ void big_panel_fetch(const double* __restrict__ G, int I, int tid, double (&st)[4]) {
  constexpr int Sp = 16 * NTL;
  const int row = tid / (4 * NTL), g = tid - row * (4 * NTL);
  const double2* src = reinterpret_cast<const double2*>(G + (size_t)(16 * I + row) * Sp + 4 * g);
  const double2 v0 = src[0], v1 = src[1];
  st[0] = v0.x; st[1] = v0.y; st[2] = v1.x; st[3] = v1.y;
}
template <int NTL>
__device__ __forceinline__ void big_panel_store(double* P, int tid, const double (&st)[4]) {
  const int row = tid / (4 * NTL), g = tid - row * (4 * NTL);
  double* d = P + row * BIG_PLD + (g >> 2) * 16 + (g & 3);
  d[0] = st[0]; d[4] = st[1]; d[8] = st[2]; d[12] = st[3];
}

// tile (I,J) of a dense matrix in the accumulator layout
template <int NTL>
__device__ __forceinline__ v4d big_tile_load(const double* __restrict__ A, int I, int J, int i, int kq) {
  constexpr int Sp = 16 * NTL;
  v4d v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = A[(size_t)(16 * I + 4 * r + kq) * Sp + 16 * J + i];
  return v;
}

// tile (I,J) of the symmetric Delta stored as its lower 16x16 tiles (GainPar::dpacked), in the accumulator layout
__device__ __forceinline__ v4d big_delta_load(const double* __restrict__ Dp, int I, int J, int i, int kq) {
  v4d v;
  if (I >= J) {
    const double* t = Dp + (size_t)(I * (I + 1) / 2 + J) * 256;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = t[(4 * r + kq) * 16 + i];
  } else {      // the mirror tile, transposed
    const double* t = Dp + (size_t)(J * (J + 1) / 2 + I) * 256;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = t[i * 16 + 4 * r + kq];
  }
  return v;
}

// value of tile (I,J) -> state Y (mirrored across the diagonal when I < J), optionally dense global (both triangles)
template <int NTL>
__device__ __forceinline__ void big_emit(const BigCtx<NTL>& c, int I, int J, v4d v, double* __restrict__ gout) {
  constexpr int Sp = 16 * NTL;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * r + c.kq, col = c.i;
    if (I >= J) c.Y[big_tix(I, J) + big_phys(row, col)] = v[r];
    else c.Y[big_tix(J, I) + big_phys(col, row)] = v[r];
    if (gout) {
      gout[(size_t)(16 * I + row) * Sp + 16 * J + col] = v[r];
      if (I != J) gout[(size_t)(16 * J + col) * Sp + 16 * I + row] = v[r];
    }
  }
}

// phase 1: Z(:, J) = Y G(J rows, :)'
template <int NTL>
__device__ __forceinline__ void big_phase1(const BigCtx<NTL>& c, const double* __restrict__ Gk, v4d (&Z)[NTL]) {
  constexpr int Sp = 16 * NTL;
  const double* grow = Gk + (size_t)(16 * c.wave + c.i) * Sp + 4 * c.kq;
#pragma unroll
  for (int K = 0; K < NTL; ++K) Z[K] = (v4d){0.0, 0.0, 0.0, 0.0};
  double2 g0 = reinterpret_cast<const double2*>(grow)[0], g1 = reinterpret_cast<const double2*>(grow)[1];
  // L is a run-time loop and the scheduler is fenced after every pair of tiles: with two or three waves per SIMD the
  // other waves cover the LDS latency, and hoisting the reads of many tiles costs more registers than the 168 a lane has
#pragma unroll 1
  for (int L = 0; L < NTL; ++L) {
    const double gj[4] = {g0.x, g0.y, g1.x, g1.y};
    if (L + 1 < NTL) { g0 = reinterpret_cast<const double2*>(grow + 16 * (L + 1))[0]; g1 = reinterpret_cast<const double2*>(grow + 16 * (L + 1))[1]; }
    const double* tdir = c.Y + L * 256;                         // + 256 * K(K+1)/2: tile (K, L), K >= L
    const double* ttr = c.Y + (L * (L + 1) / 2) * 256;          // + 256 * K: tile (L, K), K < L
#pragma unroll
    for (int K = 0; K < NTL; ++K) {
      double a[4];
      if (K >= L) {
        const double* t = tdir + (K * (K + 1) / 2) * 256;
        const double2 a0 = *reinterpret_cast<const double2*>(t + c.offD0), a1 = *reinterpret_cast<const double2*>(t + c.offD1);
        a[0] = a0.x; a[1] = a0.y; a[2] = a1.x; a[3] = a1.y;
      } else {
        const double* t = ttr + K * 256;
#pragma unroll
        for (int s = 0; s < 4; ++s) a[s] = t[c.offT[s]];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) Z[K] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], gj[s], Z[K], 0, 0, 0);
      if (K & 1) __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// tile (I, J) of G B, G's rows 16I.. in panel P, column J of B in the accumulator registers Bc
template <int NTL>
__device__ __forceinline__ v4d big_panel_tile(const BigCtx<NTL>& c, const double* P, const v4d (&Bc)[NTL]) {
  v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
  const double* pr = P + c.i * BIG_PLD + 4 * c.kq;
#pragma unroll
  for (int K = 0; K < NTL; ++K) {
    const double2 a0 = *reinterpret_cast<const double2*>(pr + 16 * K), a1 = *reinterpret_cast<const double2*>(pr + 16 * K + 2);
    if (K & 1) {
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, Bc[K][0], acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, Bc[K][1], acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, Bc[K][2], acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, Bc[K][3], acc1, 0, 0, 0);
    } else {
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, Bc[K][0], acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, Bc[K][1], acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, Bc[K][2], acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, Bc[K][3], acc0, 0, 0, 0);
    }
    if (K & 1) __builtin_amdgcn_sched_barrier(0);
  }
  return acc0 + acc1;
}

// rows of panel P times the permuted vector vp -> en[16I + i]  (one wave)
template <int NTL>
__device__ __forceinline__ void big_panel_matvec(const BigCtx<NTL>& c, const double* P, int I) {
  const double* pr = P + c.i * BIG_PLD + 4 * c.kq;
  const double* vv = c.vp + 4 * c.kq;
  double s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int K = 0; K < NTL; ++K) {
    const double2 a0 = *reinterpret_cast<const double2*>(pr + 16 * K), a1 = *reinterpret_cast<const double2*>(pr + 16 * K + 2);
    const double2 v0 = *reinterpret_cast<const double2*>(vv + 16 * K), v1 = *reinterpret_cast<const double2*>(vv + 16 * K + 2);
    s0 = fma(a0.x, v0.x, s0); s1 = fma(a0.y, v0.y, s1); s0 = fma(a1.x, v1.x, s0); s1 = fma(a1.y, v1.y, s1);
  }
  double s = s0 + s1;
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  if (c.kq == 0) c.en[16 * I + c.i] = s;
}

// position of dense index d in the permuted vector
__device__ __forceinline__ int big_vperm(int d) { const int cc = d & 15; return (d & ~15) + ((cc & 3) << 2) + (cc >> 2); }

// MODE 0: compose, C chain and c (pass 1) ; 1: boundary (pass 2) ; 2: apply (pass 3)
template <int NTL, int MODE>
__global__ void __launch_bounds__(64 * NTL) BIG_REG_CAP(NTL) rts_big_kernel(Shape sh, Bufs b, MfmaPar sp) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int Sp = 16 * NTL, NT = 64 * NTL;
  BigCtx<NTL> c;
  big_ctx_init<NTL>(c, lds);
  const int tid = c.tid, J = c.wave;
  int j = (MODE == 1) ? 0 : blockIdx.x; const int pb = (MODE == 1) ? blockIdx.x : blockIdx.y;
  if (MODE == 2 && sp.tab) j = chunk_select(sp, b, j);      // merged apply launch of several chunks
  const int S = sh.S, M = sh.M;
  const int64_t T = sh.T;
  const size_t SS = (size_t)Sp * Sp;
  double* hv = c.en + Sp;
  if (MODE == 2) {
    const double* mdl = b.model + (size_t)pb * mdl_size(sh);
    for (int i = tid; i < M; i += NT) hv[i] = mdl[mdl_h(sh) + i];
  }
  int sidx = -1, myblk = 0, myrow = 0;
  if (tid < Sp) {
    myblk = tid >> 2; myrow = tid & 3;
    if (myblk < M && myrow < sh.bsz[myblk]) sidx = sh.off[myblk] + myrow;
  }
  // ---- the chain: steps n = n_hi-1 .. n_lo, each with a matrix G_n, a term added AFTER the product (the next step's
  // Delta, or C of the span) and, before the first product, a term added to the initial state
  int n_lo, n_hi;
  if (MODE == 1) { n_lo = 0; n_hi = sp.ns; }
  else { n_lo = j * sp.L; n_hi = (n_lo + sp.L < sp.nk) ? n_lo + sp.L : sp.nk; }
  auto Gmat = [&](int n) -> const double* {
    return (MODE == 1) ? sp.spanbuf + (((size_t)pb * sp.ns_max + n) * 2) * SS : b.Gbuf + (b.gpstride ? (size_t)pb * b.gpstride + (size_t)n * gd_step_doubles(Sp, sp.dpacked) : ((size_t)pb * sp.chunk + n) * gd_step_doubles(Sp, sp.dpacked));
  };
  auto Dvec = [&](int n) -> double {     // vector term of index n (delta_n, or c of span n), dense row tid
    if (sidx < 0) return 0.0;
    return (MODE == 1) ? sp.spanvec[((size_t)pb * sp.ns_max + n) * S + sidx] : b.dbuf[((size_t)pb * sp.chunk + n) * S + sidx];
  };
  double* Est = sp.stateD + (size_t)pb * (SS + S);
  // ---- initial state
  {
    const double* E0 = nullptr;           // nullptr: zero
    const double* A0 = nullptr;
    double* g0 = nullptr;
    if (MODE == 0) { if (n_hi > n_lo) A0 = Gmat(n_hi - 1) + SS; }
    if (MODE == 1) { if (!sp.first) E0 = Est; if (n_hi > 0) g0 = sp.bnd + ((size_t)pb * sp.ns_max + (n_hi - 1)) * (SS + S); }
    if (MODE == 2) { E0 = sp.bnd + ((size_t)pb * sp.ns_max + j) * (SS + S); if (n_hi > n_lo) A0 = Gmat(n_hi - 1) + SS; }
    for (int I = 0; I < NTL; ++I)
      if (big_active<NTL>(I, J)) {
        v4d v = {0.0, 0.0, 0.0, 0.0};
        if (E0) v = big_tile_load<NTL>(E0, I, J, c.i, c.kq);
        if (A0) v += (MODE != 1 && sp.dpacked) ? big_delta_load(A0, I, J, c.i, c.kq) : big_tile_load<NTL>(A0, I, J, c.i, c.kq);
        big_emit<NTL>(c, I, J, v, g0);
      }
    if (tid < Sp) {
      double e0 = 0.0;
      if (sidx >= 0) {
        if (MODE == 1 && !sp.first) e0 = Est[SS + sidx];
        if (MODE == 2) e0 = sp.bnd[((size_t)pb * sp.ns_max + j) * (SS + S) + SS + sidx];
        if (MODE == 1 && n_hi > 0) sp.bnd[((size_t)pb * sp.ns_max + (n_hi - 1)) * (SS + S) + SS + sidx] = e0;
        if (MODE != 1 && n_hi > n_lo) e0 += Dvec(n_hi - 1);
      }
      c.vp[big_vperm(tid)] = e0;
    }
  }
  double mxM = 0.0, mxP = 0.0;
  __syncthreads();
  for (int n = n_hi - 1; n >= n_lo; --n) {
    const double* Gk = Gmat(n);
    const int64_t k = sp.k0 + n;                  // MODE 2: time step of the outputs
    const bool last = (n == n_lo);
    // term added after the product, and where the sum goes besides Y
    const double* Add = nullptr;
    double* gout = nullptr;
    if (MODE == 1) { Add = Gk + SS; gout = (n > 0) ? sp.bnd + ((size_t)pb * sp.ns_max + (n - 1)) * (SS + S) : Est; }
    else {
      if (!last) Add = Gmat(n - 1) + SS;
      if (MODE == 0 && last) gout = sp.spanbuf + (((size_t)pb * sp.ns_max + j) * 2 + 1) * SS;
    }
    const double dnext = (MODE == 1) ? Dvec(n) : (last ? 0.0 : Dvec(n - 1));
    // ---- phase 1
    v4d Z[NTL];
    big_phase1<NTL>(c, Gk, Z);
    double st[4];
    big_panel_fetch<NTL>(Gk, 0, tid, st);
    big_panel_store<NTL>(c.P, tid, st);
    // ---- phase 2 over the panels
#pragma unroll 1
    for (int I = 0; I < NTL; ++I) {
      __syncthreads();      // panel I in place; I = 0: every wave is through with Y
      const double* Pc = c.P + (I & 1) * 16 * BIG_PLD;
      if (I + 1 < NTL) big_panel_fetch<NTL>(Gk, I + 1, tid, st);
      if (big_active<NTL>(I, J)) {
        v4d addv = {0.0, 0.0, 0.0, 0.0};
        if (Add) addv = (MODE != 1 && sp.dpacked) ? big_delta_load(Add, I, J, c.i, c.kq) : big_tile_load<NTL>(Add, I, J, c.i, c.kq);
        const v4d e = big_panel_tile<NTL>(c, Pc, Z);
        if (MODE == 2 && I == J) {
          // smoothed marginal variances: E(4m,4m) of the diagonal tile sits in register m of lane 4m
          if ((c.lane & 3) == 0 && c.lane < 16) {
            const int rr = c.lane >> 2, nn = 4 * J + rr;
            if (nn < M) {
              const double ev_ = (rr == 0) ? e[0] : ((rr == 1) ? e[1] : ((rr == 2) ? e[2] : e[3]));
              const size_t ix = ((size_t)pb * T + k) * M + nn;
              const double vnew = b.fv[ix] + hv[nn] * hv[nn] * ev_;
              mxP = fmax(mxP, fabs(b.sv[ix] - vnew));
              b.sv[ix] = vnew;
            }
          }
        }
        big_emit<NTL>(c, I, J, e + addv, gout);
      } else if (J == (I + 1) % NTL) {
        big_panel_matvec<NTL>(c, Pc, I);
      }
      if (I + 1 < NTL) big_panel_store<NTL>(c.P + ((I + 1) & 1) * 16 * BIG_PLD, tid, st);
    }
    __syncthreads();        // Y and en complete
    if (tid < Sp) {
      const double e_new = c.en[tid];
      if (sidx >= 0) {
        if (MODE == 2) {
          const double ms = b.MF[((size_t)pb * T + k) * S + sidx] + e_new;
          b.MS[((size_t)pb * T + k) * S + sidx] = ms;
          if (k == 0) b.state[(size_t)pb * ((size_t)sh.ntiles * 16 + S) + (size_t)sh.ntiles * 16 + sidx] = ms;
          if (myrow == 0) {
            const size_t ix = ((size_t)pb * T + k) * M + myblk;
            const double mnew = hv[myblk] * ms;
            mxM = fmax(mxM, fabs(b.sm[ix] - mnew));
            b.sm[ix] = mnew;
          }
        }
        if (MODE == 0 && last) sp.spanvec[((size_t)pb * sp.ns_max + j) * S + sidx] = e_new;
        if (MODE == 1) {
          double* dst = (n > 0) ? sp.bnd + ((size_t)pb * sp.ns_max + (n - 1)) * (SS + S) : Est;
          dst[SS + sidx] = e_new + dnext;
        }
      }
      c.vp[big_vperm(tid)] = (sidx >= 0) ? e_new + dnext : 0.0;
    }
    if (MODE == 2 && k == 0) {
      // restart state of the next sweep: P^s_0 = P_0 + E_0 in tile-major layout (last step of the bottom span; Y = E_0)
      const double* PFk = b.PF + ((size_t)pb * T + k) * pf_step_doubles(sh);
      for (int q = tid; q < Sp * Sp; q += NT) {
        const int row = q / Sp, col = q - row * Sp;
        const int Ib = row >> 2, Jb = col >> 2;
        if (Ib < M && Jb < M) {
          const int TI = row >> 4, TJ = col >> 4;
          const double ev_ = (TI >= TJ) ? c.Y[big_tix(TI, TJ) + big_phys(row & 15, col & 15)] : c.Y[big_tix(TJ, TI) + big_phys(col & 15, row & 15)];
          b.state[(size_t)pb * ((size_t)sh.ntiles * 16 + S) + ((size_t)Ib * M + Jb) * 16 + 4 * (row & 3) + (col & 3)] = pf_elem(PFk, Ib, Jb, row & 3, col & 3) + ev_;
        }
      }
    }
    __syncthreads();        // vp in place; the panels are free
  }
  if (MODE == 2) {
    mxM = wave_max(mxM);
    mxP = wave_max(mxP);
    if ((tid & 63) == 0) {
      atomicMax(reinterpret_cast<unsigned long long*>(&b.red[(size_t)pb * 8 + 1]), (unsigned long long)__double_as_longlong(mxM));
      atomicMax(reinterpret_cast<unsigned long long*>(&b.red[(size_t)pb * 8 + 2]), (unsigned long long)__double_as_longlong(mxP));
    }
  }
}

// ---- pass 1, Phi chain: Phi <- G Phi over the span, column J of Phi in the registers of wave J
template <int NTL>
__global__ void __launch_bounds__(64 * NTL) BIG_REG_CAP(NTL) rts_big_phi_kernel(Shape sh, Bufs b, MfmaPar sp) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int Sp = 16 * NTL, HALF = (NTL + 1) / 2, NPK = NTL - HALF;
  BigCtx<NTL> c;
  big_ctx_init<NTL>(c, lds);
  const int tid = c.tid, J = c.wave;
  const int j = blockIdx.x, pb = blockIdx.y;
  const size_t SS = (size_t)Sp * Sp;
  double* park = c.Y + ((size_t)J * NPK) * 256 + c.lane;      // [tile][r][lane]: the Y region is free in this kernel
  v4d Ph[NTL];
#pragma unroll
  for (int K = 0; K < NTL; ++K)
#pragma unroll
    for (int r = 0; r < 4; ++r) Ph[K][r] = (K == J && 4 * r + c.kq == c.i) ? 1.0 : 0.0;
  const int n_lo = j * sp.L, n_hi = (n_lo + sp.L < sp.nk) ? n_lo + sp.L : sp.nk;
  for (int n = n_hi - 1; n >= n_lo; --n) {
    const double* Gk = b.Gbuf + (b.gpstride ? (size_t)pb * b.gpstride + (size_t)n * gd_step_doubles(Sp, sp.dpacked) : ((size_t)pb * sp.chunk + n) * gd_step_doubles(Sp, sp.dpacked));
    double st[4];
    big_panel_fetch<NTL>(Gk, 0, tid, st);
    big_panel_store<NTL>(c.P, tid, st);
    v4d Nw[HALF];
#pragma unroll
    for (int I = 0; I < NTL; ++I) {
      __syncthreads();
      const double* Pc = c.P + (I & 1) * 16 * BIG_PLD;
      if (I + 1 < NTL) big_panel_fetch<NTL>(Gk, I + 1, tid, st);
      const v4d e = big_panel_tile<NTL>(c, Pc, Ph);
      if (I < HALF) Nw[I] = e;
      else {
#pragma unroll
        for (int r = 0; r < 4; ++r) park[((I - HALF) * 4 + r) * 64] = e[r];
      }
      if (I + 1 < NTL) big_panel_store<NTL>(c.P + ((I + 1) & 1) * 16 * BIG_PLD, tid, st);
    }
#pragma unroll
    for (int I = 0; I < HALF; ++I) Ph[I] = Nw[I];
#pragma unroll
    for (int I = HALF; I < NTL; ++I)
#pragma unroll
      for (int r = 0; r < 4; ++r) Ph[I][r] = park[((I - HALF) * 4 + r) * 64];   // this lane's own stores
    __syncthreads();        // the panels are free
  }
  double* Pout = sp.spanbuf + (((size_t)pb * sp.ns_max + j) * 2) * SS;
#pragma unroll
  for (int K = 0; K < NTL; ++K)
#pragma unroll
    for (int r = 0; r < 4; ++r) Pout[(size_t)(16 * K + 4 * r + c.kq) * Sp + 16 * J + c.i] = Ph[K][r];
}

}  // namespace nagp
