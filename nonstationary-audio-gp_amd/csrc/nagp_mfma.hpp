// nagp_mfma.hpp -- FP64 MFMA (v_mfma_f64_16x16x4_f64) versions of the parallel-in-time RTS smoother passes
// for padded state dimensions Sp = 4M rounded up to 16 with Sp <= 96 (M <= 24: the 4-, 16- and 19-site
// configurations).  Matrices are dense row-major Sp x Sp in global memory (element (4I+i, 4J+j) = entry (i,j)
// of block tile (I,J), zero padded), so the gain kernel's output feeds the matrix cores directly.
//
// One workgroup of 256 threads (4 waves, one per SIMD) per span.  The recursion state E (and Phi, C in the
// compose pass) lives in MFMA accumulators; G_k and the right-hand operand sit in LDS (row stride Sp+1:
// conflict-free A-operand reads).  The C/D layout of the f64 MFMA (col = lane&15, row = (lane>>4) + 4*reg)
// is also the row-major address of the element, so accumulators are written to / read from LDS and global
// memory with the same index arithmetic.
#pragma once
#include "nagp_kernels.hpp"

namespace nagp {


struct MfmaPar {
  int64_t k0;
  int nk, chunk, L, ns, ns_max;
  int Sp;           // padded dimension (multiple of 16)
  int first, write_PSs;
  double* spanbuf;  // [B][ns_max][2][Sp*Sp]  Phi, C   (dense)
  double* spanvec;  // [B][ns_max][S]
  double* bnd;      // [B][ns_max][Sp*Sp + S]  E_top (dense), e_top
  double* stateD;   // [B][Sp*Sp + S]          dense carry between chunks
  const ChunkTab* tab;  // merged apply launch (see SpanPar)
  int ntab;
  double* xbuf;         // (unused by the MFMA passes; keeps chunk_select one template)
  int dpacked;          // column-owner passes: Delta of the chunk's slot as packed lower 16x16 tiles (GainPar::dpacked)
};

__host__ __device__ inline size_t mfma_lds_doubles(int Sp) { return 2 * (size_t)Sp * (Sp + 1) + 3 * (size_t)Sp + MAXM + 8; }

// D (16x16 tile t of an NTL x NTL tiling) += A[rows of ti, :] * B[:, cols of tj], operands in LDS:
//   As: row-major [row][k] stride LD ; Bs: row-major [k][col] stride LD  (BT = false)
//   or Bs holds the TRANSPOSED factor row-major [col][k] (BT = true: D += A * Bt')
template <bool BT>
__device__ __forceinline__ v4d mfma_tile(const double* As, const double* Bs, int LD, int Sp, int ti, int tj, v4d acc) {
  const int lane = threadIdx.x & 63;
  const int lr = lane & 15, lk = lane >> 4;
  const double* ap = As + (size_t)(16 * ti + lr) * LD + lk;
  const double* bp = BT ? (Bs + (size_t)(16 * tj + lr) * LD + lk) : (Bs + (size_t)lk * LD + 16 * tj + lr);
  const int bstep = BT ? 4 : 4 * LD;
#pragma unroll 4
  for (int ks = 0; ks < Sp / 4; ++ks) {
    const double a = ap[4 * ks];
    const double b = bp[(size_t)ks * bstep];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  return acc;
}

// All TW tiles of this wave at once: per k-step the A / B fragments of every owned tile are fetched first, then
// TW independent MFMAs are issued (a single accumulator chain runs at 84 instead of 64 cycles per MFMA, and
// the LDS latency of the next fragments hides under the matrix-core work).
template <int NTL, int TW, bool BT>
__device__ __forceinline__ void mfma_gemm_all(const double* As, const double* Bs, int LD, int Sp, int wave, v4d (&acc)[TW]) {
  const int lane = threadIdx.x & 63;
  const int lr = lane & 15, lk = lane >> 4;
  const double* ap[TW]; const double* bp[TW];
#pragma unroll
  for (int q = 0; q < TW; ++q) {
    int t = wave + 4 * q;
    if (t >= NTL * NTL) t = wave;                 // harmless duplicate; result discarded by the caller
    const int ti = t / NTL, tj = t - ti * NTL;
    ap[q] = As + (size_t)(16 * ti + lr) * LD + lk;
    bp[q] = BT ? (Bs + (size_t)(16 * tj + lr) * LD + lk) : (Bs + (size_t)lk * LD + 16 * tj + lr);
  }
  const int bstep = BT ? 4 : 4 * LD;
#pragma unroll 2
  for (int ks = 0; ks < Sp / 4; ++ks) {
    double a[TW], bb[TW];
#pragma unroll
    for (int q = 0; q < TW; ++q) { a[q] = ap[q][4 * ks]; bb[q] = bp[q][(size_t)ks * bstep]; }
#pragma unroll
    for (int q = 0; q < TW; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], bb[q], acc[q], 0, 0, 0);
  }
}

// y = G[row,:] . v  with four independent partial sums (LDS latency pipelined)
__device__ __forceinline__ double row_dot(const double* gr, const double* v, int Sp) {
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll 2
  for (int cc = 0; cc < Sp; cc += 4) {
    a0 = fma(gr[cc], v[cc], a0); a1 = fma(gr[cc + 1], v[cc + 1], a1);
    a2 = fma(gr[cc + 2], v[cc + 2], a2); a3 = fma(gr[cc + 3], v[cc + 3], a3);
  }
  return (a0 + a1) + (a2 + a3);
}

// accumulator element (tile t, register r) of lane -> dense (row, col)
__device__ __forceinline__ void acc_rc(int NTL, int t, int r, int& row, int& col) {
  const int lane = threadIdx.x & 63;
  const int ti = t / NTL, tj = t - ti * NTL;
  row = 16 * ti + (lane >> 4) + 4 * r;
  col = 16 * tj + (lane & 15);
}

template <int NTL>
struct MfmaCtx {
  static constexpr int NT2 = NTL * NTL;
  static constexpr int TW = (NT2 + 3) / 4;   // tiles per wave
  int wave, Sp, LD;
  __device__ int tile(int q) const { return wave + 4 * q; }
  __device__ bool ok(int q) const { return tile(q) < NT2; }
};

// ---- pass 3 (and the body shared with pass 2): E <- G (E + Delta) G' inside one span
template <int NTL>
__global__ void __launch_bounds__(256) rts_apply_mfma_kernel(Shape sh, Bufs b, MfmaPar sp) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  using C = MfmaCtx<NTL>;
  const int tid = threadIdx.x, pb = blockIdx.y;
  int j = blockIdx.x;
  if (sp.tab) j = chunk_select(sp, b, j);
  constexpr int Sp = 16 * NTL, LD = Sp + 1;     // == sp.Sp: the host instantiates NTL = Sp / 16; compile-time strides
  const int S = sh.S, M = sh.M;
  const int64_t T = sh.T;
  C c; c.wave = tid >> 6; c.Sp = Sp; c.LD = LD;
  double* Gs = lds;
  double* Bs = Gs + (size_t)Sp * LD;
  double* ev = Bs + (size_t)Sp * LD;      // e + delta (dense index 4I+i)
  double* en = ev + Sp;
  double* hv = en + Sp;                   // h_val[M]
  int* roff = reinterpret_cast<int*>(hv + MAXM);   // not used (kept for alignment)
  (void)roff;
  const double* mdl = b.model + (size_t)pb * mdl_size(sh);
  for (int i = tid; i < M; i += 256) hv[i] = mdl[mdl_h(sh) + i];
  const size_t SS = (size_t)Sp * Sp;
  double* Bj = sp.bnd + ((size_t)pb * sp.ns_max + j) * (SS + S);
  // state vector e in dense indexing: dense row d = 4*blk + row-in-block  <->  state index
  // thread tid < Sp handles dense row tid: its state index (or -1 for padding rows)
  int sidx = -1, myblk = 0, myrow = 0;
  if (tid < Sp) {
    myblk = tid >> 2; myrow = tid & 3;
    if (myblk < M && myrow < sh.bsz[myblk]) sidx = sh.off[myblk] + myrow;
  }
  double e_cur = (sidx >= 0) ? Bj[SS + sidx] : 0.0;
  v4d E[C::TW];
#pragma unroll
  for (int q = 0; q < C::TW; ++q) {
    E[q] = (v4d){0, 0, 0, 0};
    if (c.ok(q)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { int row, col; acc_rc(NTL, c.tile(q), r, row, col); E[q][r] = Bj[(size_t)row * Sp + col]; }
    }
  }
  double mxM = 0.0, mxP = 0.0;
  const int a0 = j * sp.L, e0 = (a0 + sp.L < sp.nk) ? a0 + sp.L : sp.nk;
  // register prefetch (one step ahead) of G_k, Delta_k, delta_k: the global latency hides under the MFMAs
  constexpr int NG = (NTL * NTL * 256 + 255) / 256;   // = NTL*NTL doubles of G per thread
  double gpre[NG]; v4d dpre[C::TW]; double dvpre = 0.0;
  auto goff = [&](int u) { const int i = tid + 256 * u; const int r = i / Sp; return r * LD + (i - r * Sp); };   // LDS offset of dense element i
  auto prefetch = [&](int kk) {
    const double* Gk = b.Gbuf + (((size_t)pb * sp.chunk + kk) * 2) * SS;
    const double* Dk = Gk + SS;
#pragma unroll
    for (int u = 0; u < NG; ++u) gpre[u] = Gk[tid + 256 * u];
#pragma unroll
    for (int q = 0; q < C::TW; ++q)
      if (c.ok(q)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { int row, col; acc_rc(NTL, c.tile(q), r, row, col); dpre[q][r] = Dk[(size_t)row * Sp + col]; }
      }
    if (sidx >= 0) dvpre = b.dbuf[((size_t)pb * sp.chunk + kk) * S + sidx];
  };
  if (e0 > a0) prefetch(e0 - 1);
  for (int kk = e0 - 1; kk >= a0; --kk) {
    const int64_t k = sp.k0 + kk;
    // G -> LDS, Y = E + Delta -> LDS, ev = e + delta
#pragma unroll
    for (int u = 0; u < NG; ++u) Gs[goff(u)] = gpre[u];
#pragma unroll
    for (int q = 0; q < C::TW; ++q)
      if (c.ok(q)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { int row, col; acc_rc(NTL, c.tile(q), r, row, col); Bs[(size_t)row * LD + col] = E[q][r] + dpre[q][r]; }
      }
    if (tid < Sp) ev[tid] = (sidx >= 0) ? e_cur + dvpre : 0.0;
    __syncthreads();
    if (kk > a0) prefetch(kk - 1);
    // X = G Y ; e' = G (e + delta)
    v4d X[C::TW];
#pragma unroll
    for (int q = 0; q < C::TW; ++q) X[q] = (v4d){0, 0, 0, 0};
    mfma_gemm_all<NTL, C::TW, false>(Gs, Bs, LD, Sp, c.wave, X);
    if (tid < Sp) {
      e_cur = row_dot(Gs + (size_t)tid * LD, ev, Sp);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < C::TW; ++q)
      if (c.ok(q)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { int row, col; acc_rc(NTL, c.tile(q), r, row, col); Bs[(size_t)row * LD + col] = X[q][r]; }
      }
    __syncthreads();
    // E' = X G'
#pragma unroll
    for (int q = 0; q < C::TW; ++q) E[q] = (v4d){0, 0, 0, 0};
    mfma_gemm_all<NTL, C::TW, true>(Bs, Gs, LD, Sp, c.wave, E);
    // ---- outputs.  Smoothed marginal variances: E(4n,4n) sits in the diagonal tile n/4, accumulator register n%4 of lane
    // 4*(n%4) (C-layout row = (lane>>4) + 4r, col = lane&15) -- picked directly instead of scanning every element.
    {
      const int lane = tid & 63;
      if ((lane & 3) == 0 && lane < 16) {
#pragma unroll
        for (int q = 0; q < C::TW; ++q) {
          const int t = c.tile(q);
          if (c.ok(q) && t % (NTL + 1) == 0) {
            const int rr = lane >> 2, n = 4 * (t / (NTL + 1)) + rr;
            if (n < M) {
              const double ev_ = (rr == 0) ? E[q][0] : ((rr == 1) ? E[q][1] : ((rr == 2) ? E[q][2] : E[q][3]));
              const size_t ix = ((size_t)pb * T + k) * M + n;
              const double vnew = b.fv[ix] + hv[n] * hv[n] * ev_;
              mxP = fmax(mxP, fabs(b.sv[ix] - vnew));
              b.sv[ix] = vnew;
            }
          }
        }
      }
    }
    if (sp.write_PSs || k == 0) {     // rare (block-uniform): smoothed covariances requested, or the restart state of the next EKF sweep
      // E goes through LDS with compile-time accumulator indices: a run-time index into E[] would keep the whole
      // accumulator array in scratch memory for every step of the loop
      __syncthreads();                 // every wave is done with Bs (A operand of the second product)
#pragma unroll
      for (int q = 0; q < C::TW; ++q)
        if (c.ok(q)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { int row, col; acc_rc(NTL, c.tile(q), r, row, col); Bs[(size_t)row * LD + col] = E[q][r]; }
        }
      __syncthreads();
      const double* PFk = b.PF + ((size_t)pb * T + k) * pf_step_doubles(sh);
      for (int i = tid; i < Sp * Sp; i += 256) {
        const int row = i / Sp, col = i - row * Sp;
        const int I = row >> 2, J = col >> 2;
        if (I < M && J < M) {
          const size_t tix = (((size_t)pb * T + k) * sh.ntiles + (size_t)I * M + J) * 16 + 4 * (row & 3) + (col & 3);
          const double ps = pf_elem(PFk, I, J, row & 3, col & 3) + Bs[(size_t)row * LD + col];
          if (sp.write_PSs) b.PSs[tix] = ps;
          if (k == 0) b.state[(size_t)pb * ((size_t)sh.ntiles * 16 + S) + ((size_t)I * M + J) * 16 + 4 * (row & 3) + (col & 3)] = ps;
        }
      }
    }
    if (sidx >= 0) {
      const double ms = b.MF[((size_t)pb * T + k) * S + sidx] + e_cur;
      b.MS[((size_t)pb * T + k) * S + sidx] = ms;
      if (k == 0) b.state[(size_t)pb * ((size_t)sh.ntiles * 16 + S) + (size_t)sh.ntiles * 16 + sidx] = ms;
      if (myrow == 0) {
        const size_t ix = ((size_t)pb * T + k) * M + myblk;
        const double mnew = hv[myblk] * ms;
        mxM = fmax(mxM, fabs(b.sm[ix] - mnew));
        b.sm[ix] = mnew;
      }
    }
    __syncthreads();
  }
  mxM = wave_max(mxM);
  mxP = wave_max(mxP);
  if ((tid & 63) == 0) {
    atomicMax(reinterpret_cast<unsigned long long*>(&b.red[(size_t)pb * 8 + 1]), (unsigned long long)__double_as_longlong(mxM));
    atomicMax(reinterpret_cast<unsigned long long*>(&b.red[(size_t)pb * 8 + 2]), (unsigned long long)__double_as_longlong(mxP));
  }
}

// ---- pass 1: Phi <- G Phi ; C <- G (C + Delta) G' ; c <- G (c + delta)
template <int NTL>
__global__ void __launch_bounds__(256) rts_compose_mfma_kernel(Shape sh, Bufs b, MfmaPar sp) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  using C = MfmaCtx<NTL>;
  const int tid = threadIdx.x, j = blockIdx.x, pb = blockIdx.y;
  constexpr int Sp = 16 * NTL, LD = Sp + 1;     // == sp.Sp: the host instantiates NTL = Sp / 16; compile-time strides
  const int S = sh.S, M = sh.M;
  C c; c.wave = tid >> 6; c.Sp = Sp; c.LD = LD;
  double* Gs = lds;
  double* Bs = Gs + (size_t)Sp * LD;
  double* ev = Bs + (size_t)Sp * LD;
  const size_t SS = (size_t)Sp * Sp;
  int sidx = -1;
  if (tid < Sp) {
    const int blk = tid >> 2, row = tid & 3;
    if (blk < M && row < sh.bsz[blk]) sidx = sh.off[blk] + row;
  }
  double c_cur = 0.0;
  v4d Ph[C::TW], Cm[C::TW];
#pragma unroll
  for (int q = 0; q < C::TW; ++q) {
    Ph[q] = (v4d){0, 0, 0, 0}; Cm[q] = (v4d){0, 0, 0, 0};
    if (c.ok(q)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { int row, col; acc_rc(NTL, c.tile(q), r, row, col); Ph[q][r] = (row == col) ? 1.0 : 0.0; }
    }
  }
  const int a0 = j * sp.L, e0 = (a0 + sp.L < sp.nk) ? a0 + sp.L : sp.nk;
  constexpr int NG = NTL * NTL;
  double gpre[NG]; v4d dpre[C::TW]; double dvpre = 0.0;
  auto goff = [&](int u) { const int i = tid + 256 * u; const int r = i / Sp; return r * LD + (i - r * Sp); };   // LDS offset of dense element i
  auto prefetch = [&](int kk) {
    const double* Gk = b.Gbuf + (((size_t)pb * sp.chunk + kk) * 2) * SS;
    const double* Dk = Gk + SS;
#pragma unroll
    for (int u = 0; u < NG; ++u) gpre[u] = Gk[tid + 256 * u];
#pragma unroll
    for (int q = 0; q < C::TW; ++q)
      if (c.ok(q)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { int row, col; acc_rc(NTL, c.tile(q), r, row, col); dpre[q][r] = Dk[(size_t)row * Sp + col]; }
      }
    if (sidx >= 0) dvpre = b.dbuf[((size_t)pb * sp.chunk + kk) * S + sidx];
  };
  if (e0 > a0) prefetch(e0 - 1);
  for (int kk = e0 - 1; kk >= a0; --kk) {
#pragma unroll
    for (int u = 0; u < NG; ++u) Gs[goff(u)] = gpre[u];
#pragma unroll
    for (int q = 0; q < C::TW; ++q)
      if (c.ok(q)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { int row, col; acc_rc(NTL, c.tile(q), r, row, col); Bs[(size_t)row * LD + col] = Ph[q][r]; }
      }
    if (tid < Sp) ev[tid] = (sidx >= 0) ? c_cur + dvpre : 0.0;
    v4d dcur[C::TW];
#pragma unroll
    for (int q = 0; q < C::TW; ++q) dcur[q] = dpre[q];
    __syncthreads();
    if (kk > a0) prefetch(kk - 1);
#pragma unroll
    for (int q = 0; q < C::TW; ++q) Ph[q] = (v4d){0, 0, 0, 0};
    mfma_gemm_all<NTL, C::TW, false>(Gs, Bs, LD, Sp, c.wave, Ph);
    if (tid < Sp) {
      c_cur = row_dot(Gs + (size_t)tid * LD, ev, Sp);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < C::TW; ++q)
      if (c.ok(q)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { int row, col; acc_rc(NTL, c.tile(q), r, row, col); Bs[(size_t)row * LD + col] = Cm[q][r] + dcur[q][r]; }
      }
    __syncthreads();
    v4d X[C::TW];
#pragma unroll
    for (int q = 0; q < C::TW; ++q) X[q] = (v4d){0, 0, 0, 0};
    mfma_gemm_all<NTL, C::TW, false>(Gs, Bs, LD, Sp, c.wave, X);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < C::TW; ++q)
      if (c.ok(q)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { int row, col; acc_rc(NTL, c.tile(q), r, row, col); Bs[(size_t)row * LD + col] = X[q][r]; }
      }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < C::TW; ++q) Cm[q] = (v4d){0, 0, 0, 0};
    mfma_gemm_all<NTL, C::TW, true>(Bs, Gs, LD, Sp, c.wave, Cm);
    __syncthreads();
  }
  double* Pout = sp.spanbuf + (((size_t)pb * sp.ns_max + j) * 2) * SS;
  double* Cout = Pout + SS;
#pragma unroll
  for (int q = 0; q < C::TW; ++q)
    if (c.ok(q)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row, col; acc_rc(NTL, c.tile(q), r, row, col);
        Pout[(size_t)row * Sp + col] = Ph[q][r];
        Cout[(size_t)row * Sp + col] = Cm[q][r];
      }
    }
  if (sidx >= 0) sp.spanvec[((size_t)pb * sp.ns_max + j) * S + sidx] = c_cur;
}

// ---- pass 2: E_bot = Phi E_top Phi' + C, sequential over the spans of the chunk
template <int NTL>
__global__ void __launch_bounds__(256) rts_boundary_mfma_kernel(Shape sh, Bufs b, MfmaPar sp) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  using C = MfmaCtx<NTL>;
  const int tid = threadIdx.x, pb = blockIdx.x;
  constexpr int Sp = 16 * NTL, LD = Sp + 1;     // == sp.Sp: the host instantiates NTL = Sp / 16; compile-time strides
  const int S = sh.S, M = sh.M;
  C c; c.wave = tid >> 6; c.Sp = Sp; c.LD = LD;
  double* Gs = lds;
  double* Bs = Gs + (size_t)Sp * LD;
  double* ev = Bs + (size_t)Sp * LD;
  const size_t SS = (size_t)Sp * Sp;
  double* Est = sp.stateD + (size_t)pb * (SS + S);
  int sidx = -1;
  if (tid < Sp) {
    const int blk = tid >> 2, row = tid & 3;
    if (blk < M && row < sh.bsz[blk]) sidx = sh.off[blk] + row;
  }
  double e_cur = (sidx >= 0 && !sp.first) ? Est[SS + sidx] : 0.0;
  v4d E[C::TW];
#pragma unroll
  for (int q = 0; q < C::TW; ++q) {
    E[q] = (v4d){0, 0, 0, 0};
    if (c.ok(q) && !sp.first) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { int row, col; acc_rc(NTL, c.tile(q), r, row, col); E[q][r] = Est[(size_t)row * Sp + col]; }
    }
  }
  constexpr int NG = NTL * NTL;
  double gpre[NG]; v4d cpre[C::TW]; double cvpre = 0.0;
  auto goff = [&](int u) { const int i = tid + 256 * u; const int r = i / Sp; return r * LD + (i - r * Sp); };   // LDS offset of dense element i
  auto prefetch = [&](int jj) {
    const double* Phi = sp.spanbuf + (((size_t)pb * sp.ns_max + jj) * 2) * SS;
    const double* Cm = Phi + SS;
#pragma unroll
    for (int u = 0; u < NG; ++u) gpre[u] = Phi[tid + 256 * u];
#pragma unroll
    for (int q = 0; q < C::TW; ++q)
      if (c.ok(q)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { int row, col; acc_rc(NTL, c.tile(q), r, row, col); cpre[q][r] = Cm[(size_t)row * Sp + col]; }
      }
    if (sidx >= 0) cvpre = sp.spanvec[((size_t)pb * sp.ns_max + jj) * S + sidx];
  };
  if (sp.ns > 0) prefetch(sp.ns - 1);
  for (int j = sp.ns - 1; j >= 0; --j) {
    double* Bj = sp.bnd + ((size_t)pb * sp.ns_max + j) * (SS + S);
#pragma unroll
    for (int u = 0; u < NG; ++u) Gs[goff(u)] = gpre[u];
    v4d ccur[C::TW];
#pragma unroll
    for (int q = 0; q < C::TW; ++q) ccur[q] = cpre[q];
    const double cvcur = cvpre;
#pragma unroll
    for (int q = 0; q < C::TW; ++q)
      if (c.ok(q)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int row, col; acc_rc(NTL, c.tile(q), r, row, col);
          Bj[(size_t)row * Sp + col] = E[q][r];             // E_top of span j
          Bs[(size_t)row * LD + col] = E[q][r];
        }
      }
    if (sidx >= 0) Bj[SS + sidx] = e_cur;
    if (tid < Sp) ev[tid] = (sidx >= 0) ? e_cur : 0.0;
    __syncthreads();
    if (j > 0) prefetch(j - 1);
    v4d X[C::TW];
#pragma unroll
    for (int q = 0; q < C::TW; ++q) X[q] = (v4d){0, 0, 0, 0};
    mfma_gemm_all<NTL, C::TW, false>(Gs, Bs, LD, Sp, c.wave, X);
    if (tid < Sp) {
      e_cur = row_dot(Gs + (size_t)tid * LD, ev, Sp) + cvcur;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < C::TW; ++q)
      if (c.ok(q)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { int row, col; acc_rc(NTL, c.tile(q), r, row, col); Bs[(size_t)row * LD + col] = X[q][r]; }
      }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < C::TW; ++q) E[q] = ccur[q];
    mfma_gemm_all<NTL, C::TW, true>(Bs, Gs, LD, Sp, c.wave, E);
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < C::TW; ++q)
    if (c.ok(q)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { int row, col; acc_rc(NTL, c.tile(q), r, row, col); Est[(size_t)row * Sp + col] = E[q][r]; }
    }
  if (sidx >= 0) Est[SS + sidx] = e_cur;
}

}  // namespace nagp
