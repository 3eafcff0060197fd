// nagp_gain_mfma.hpp -- RTS gain on the FP64 matrix cores (dense Sp x Sp output, the input layout of the MFMA smoother passes).
//
//   B = PS_k A' ; PSkp = A B + Q ; L = chol(PSkp,'lower') (jitter retry) ; G = B / L' / L ; Delta_k = PS_{k+1} - PSkp ;
//   delta_k = MF_{k+1} - A MF_k                                                        (gf_ep_modulator_nmf.m:210-230)
//
// One workgroup of NTL + 1 waves (NTL = Sp/16) per (step, problem), on 16x16 tiles of the padded dense matrices (dense index =
// 4*block + row).  A gain step is 2.33 S^3 multiply-adds behind ONE dependence chain -- diagonal tile J factored -> sub-diagonal
// tile (J+1,J) -> diagonal tile J+1 updated -> factored ... -- and what a step costs is how much of everything else runs beside
// that chain.  So the chain has a wave of its own and nobody ever waits for anything but it:
//   * one wave, the CHAIN wave: per block column J it finishes the two tiles the chain runs through -- (J+1,J) and (J+1,J+1):
//     the contribution of column J-1 (whose tiles the other waves finished one interval earlier), the panel product with
//     inv(L_JJ)', the square of the new sub-diagonal tile -- with its own LDS traffic only, then factors AND inverts tile J+1 in
//     registers (nagp_chol16.hpp: a row per lane, DPP broadcasts, no LDS), and publishes inv(L_{J+1,J+1}).  ONE workgroup barrier per
//     block column;
//   * the other NTL waves, the COLUMN waves, in the same interval: the trailing update with column J-1 of every other live tile (fused
//     with the panel product for the tiles of column J), dealt round robin, and -- wave c owns tile column c of B' = A PS_k in
//     accumulator registers from the prologue to the G store -- row J of the forward solve Y = L^-1 B', which needs row J of L only
//     and therefore runs INSIDE the factorisation, lagging one interval behind it;
//   * after the last column: the backward solve W = L'^-1 Y, wave c on its own tile column, no synchronisation at all.
//     The accumulator layout of the MFMA (lane = col + 16*kq, register t <-> row 4t+kq) IS its B-operand layout, so Y_K, W_K are
//     multiplied from the registers they were accumulated in; only the L tiles travel (LDS -> A operand), and the triangular
//     solves with the diagonal tiles are products with their inverses;
//   * prologue: PS_k is copied into LDS once (coalesced 16-byte pieces, the layout of PF), every column wave forms its tile column
//     of B' from there (4x4 block products on the VALU), PSkp = B' A' + Q comes out of the B' registers by DPP quad broadcasts
//     (the four columns of a state block sit in the four lanes of a quad) and goes to LDS over the dead staging copy; Delta_k is
//     one pass over those tiles against PS_{k+1}, whose loads are issued before PSkp is formed;
//   * G = W' leaves as 32-byte runs, Delta as 512-byte runs (packed lower 16x16 tiles) or 128-byte row pieces (dense).
// LDS: max(NTL(NTL+1)/2 tiles of 2 KiB, PS_k) + NTL inverse tiles (130 KiB at Sp = 160: one workgroup per CU; 40 KiB at Sp = 80).
// Columns of a tile are stored permuted (column 4s+kq at position 4kq+s) so that the four k-steps of a lane's A operand are 32
// contiguous bytes.
#pragma once
#include "nagp_mfma.hpp"
#include "nagp_chol16.hpp"

namespace nagp {

__host__ __device__ inline size_t gainm_union_doubles(int NTL, const Shape& sh) {
  const size_t lt = (size_t)(NTL * (NTL + 1) / 2) * 256, st = pf_step_doubles(sh);
  return lt > st ? lt : st;
}
__host__ __device__ inline size_t gainm_lds_doubles(int NTL, const Shape& sh) {
  return gainm_union_doubles(NTL, sh) + (size_t)NTL * 256 + (size_t)MAXM * 16 + MAXM + 8;
}
__device__ __forceinline__ int gm_p(int c) { return ((c & 3) << 2) + (c >> 2); }                  // stored position of column c
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
// store the accumulator TRANSPOSED: element (4t+kq, i) of the accumulator becomes element (i, 4t+kq) of the stored tile
__device__ __forceinline__ void gm_store_accT(double* Tl, v4d v, int i, int kq) {
#pragma unroll
  for (int t = 0; t < 4; ++t) Tl[i * 16 + gm_p(4 * t + kq)] = v[t];
}
// the transposed tile in accumulator layout: element (4t+kq, i) of Tl'
__device__ __forceinline__ v4d gm_load_accT(const double* Tl, int i, int kq) {
  v4d v;
#pragma unroll
  for (int t = 0; t < 4; ++t) v[t] = Tl[i * 16 + gm_p(4 * t + kq)];
  return v;
}
__device__ __forceinline__ void gm_wave_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// value of lane q of this lane's quad (DPP quad_perm broadcast).  The builtin must be the operand of a plain `return` of the right type:
// passed straight into an overloaded function (fma) clang types the call as int and converts the bit pattern numerically.
template <int CTRL>
__device__ __forceinline__ double gm_quad(double v) { return __builtin_amdgcn_update_dpp(0.0, v, CTRL, 0xF, 0xF, true); }
// the value is computed HERE: without it the IR-level sinking moves the multiply-adds of B' down to their first use two phases later, every LDS
// operand stays live until then and is spilled the moment it is read (477 spilled registers in the B' phase alone)
__device__ __forceinline__ void gm_pin(double& v) { asm volatile("" : "+v"(v)); }
// an opaque copy of a lane index: address arithmetic that depends on it is not hoisted out of the enclosing loop (the compiler otherwise
// computes the ~200 per-lane LDS addresses of a whole gain step in front of the retry loop and spills them)
__device__ __forceinline__ int gm_opaque(int v) { asm volatile("" : "+v"(v)); return v; }

// the chain wave: tile T (LDS, lower part meaningful) -> inv(chol(T)) into Ti (LDS, full tile, zeros above the diagonal)
// (measured as a non-inlined function: 6 900 instead of 6 400 cycles per tile -- inlined it stays)
__device__ __forceinline__ bool gm_chol_inv_tile(const double* T, double* Ti, int lane) {
  const int row = lane & 15, g = lane >> 4;
  double a[16], x[4];
  const double2* src = reinterpret_cast<const double2*>(T + row * 16);
#pragma unroll
  for (int m = 0; m < 8; ++m) {      // positions 2m, 2m+1 hold columns 4(q&3) + (q>>2)
    const double2 v = src[m];
    a[4 * ((2 * m) & 3) + ((2 * m) >> 2)] = v.x;
    a[4 * ((2 * m + 1) & 3) + ((2 * m + 1) >> 2)] = v.y;
  }
  const bool ok = chol16_inv_rows(a, x, row, g);
  // lane row g holds columns g, 4+g, 8+g, 12+g of its row of the inverse: positions 4g .. 4g+3 of the stored row -- 32 contiguous bytes
  double2* dst = reinterpret_cast<double2*>(Ti + row * 16 + 4 * g);
  dst[0] = make_double2(x[0], x[1]);
  dst[1] = make_double2(x[2], x[3]);
  return __ballot(!ok) == 0ull;
}

// Which tile column a column wave owns.  In the solve form every column is the same work and the order is the identity.  In the
// explicit-inverse form column c is a triangle of (NTL - c)(NTL - c + 1)/2 tile products, in the forward rows and again in the backward
// solve, and a wave's SIMD is fixed by its index (waves go to the four SIMDs round robin; wave 3 is the chain): the largest column goes to
// the chain's SIMD-mate (wave 7: the chain needs its SIMD least while the others solve backward), the rest largest first to the SIMD
// with the least work so far -- {55}, {45,10,1}, {36,15,3}, {28,21,6} at ten columns instead of {55,28,6}, {45,21,3}, {36,15,1}, {10}.
struct GmColMap { int col[12]; };
template <int NTL, bool INV>
constexpr GmColMap gm_col_map() {
  GmColMap m{};
  constexpr int CW = (NTL >= 4) ? 3 : NTL, NWV = NTL + 1;
  for (int w = 0; w < 12; ++w) m.col[w] = -1;
  if (!INV) { for (int w = 0; w < NWV; ++w) m.col[w] = (w == CW) ? -1 : (w > CW ? w - 1 : w); return m; }
  int load[4] = {0, 0, 0, 0};
  int next = 0;
  if (NWV > 7) { m.col[7] = 0; load[3] += NTL * (NTL + 1) / 2; next = 1; }
  for (int c = next; c < NTL; ++c) {
    int best = -1;
    for (int w = 0; w < NWV; ++w) {
      if (w == CW || m.col[w] >= 0) continue;
      if (best < 0 || load[w & 3] < load[best & 3]) best = w;
    }
    m.col[best] = c; load[best & 3] += (NTL - c) * (NTL - c + 1) / 2;
  }
  return m;
}

// LDS offset (doubles) of the first element of tile t in the layout of PF (pf_off(t, 0))
__device__ __forceinline__ int gm_tpart(int t) { return ((t >> 6) << 10) + ((t & 63) << 1); }

// INV: the gain through the explicit inverse.  With X = PSkp^-1 and PS_k A' = A^-1 (PSkp - Q):  G = PS_k A' X = A^-1 - (A^-1 Q) X.  A and Q are
// block diagonal, so once X is known G is a 4x4-block operation on it -- and X costs HALF the solves: the tile columns of L^-1 (forward
// substitution on the identity: column c starts at tile row c) and of X = L'^-1 L^-1 (backward substitution, needed for tile rows >= c only,
// X being symmetric) are triangles, 220 + 220 tile products at ten tile columns instead of 550 + 550; B' = A PS_k is not formed at all
// (PSkp alone, for the 55 lower tiles).  The host enables it when every block of A is comfortably invertible (GainPar::ainv holds
// A^-1 and A^-1 Q per block; |A_b^-1| <= 8: the error of G is that of X times |A^-1|) -- otherwise, and for any caller that passes no
// inverses, the solve form below runs.
// Up to six tile columns the kernel is capped at 128 registers (four waves per SIMD; 30 / 53 registers spilled at NTL = 5 / 6): its LDS lets two workgroups
// share a CU there (four at NTL = 3), but a workgroup of NTL + 1 waves puts two of them on the first SIMDs, and a second workgroup is only placed
// beside it if EVERY SIMD can take its share -- at 164 registers (three waves per SIMD) the occupancy query says two, the hardware runs one
// (profiles/r05_gain_mfma.txt: 31.5 -> 27.4 us per step and CU at Sp = 80, 13.4 -> 8.3 at Sp = 48; cfg2_batch 444 -> 404 ms).
template <int NTL, bool INV>
__global__ void __launch_bounds__(64 * (NTL + 1)) __attribute__((amdgpu_waves_per_eu(NTL <= 6 ? 4 : 1, NTL <= 6 ? 4 : 8))) rts_gain_mfma_kernel(Shape sh, Bufs b, GainPar gp) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int Sp = 16 * NTL, NT = 64 * (NTL + 1), NLOW = NTL * (NTL + 1) / 2;
  constexpr int ND = 6;                                           // Delta tiles per column wave and round
  constexpr int NLD = ((2 * NTL * (4 * NTL + 1) + 63) / 64 * 64 * 8 + NT - 1) / NT;      // 16-byte pieces of PS_k per thread (M <= 4 NTL)
  const int tid = threadIdx.x, lane = tid & 63, i0 = lane & 15, kq0 = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wave 3 is the chain wave: waves are dealt to the four SIMDs round robin, so with NTL + 1 <= 11 waves SIMD 3 holds the chain and ONE
  // column wave instead of two (an FP64 MFMA holds its SIMD's issue for its duration: every MFMA of a SIMD-mate delays the chain)
  constexpr int CW = (NTL >= 4) ? 3 : NTL;
  const bool chain = (w == CW);
  constexpr GmColMap CMAP = gm_col_map<NTL, INV>();
  int c = 0;                                                      // tile column of a column wave (gm_col_map)
#pragma unroll
  for (int q = 0; q < NTL + 1; ++q) if (q == w && CMAP.col[q] >= 0) c = CMAP.col[q];
  const int S = sh.S, M = sh.M;
  const int64_t T = sh.T;
  // Workgroups go to the eight XCDs round robin in dispatch order, and each XCD has an L2 of its own.  Step k reads PS_k AND PS_{k+1}
  // (Delta_k), so consecutive steps on consecutive XCDs fetch every filtered covariance from HBM twice; with the steps of an XCD contiguous
  // (launch: gridDim.x = nk rounded up to a multiple of eight) the second reader finds it in its L2.
  const int pb = blockIdx.y;
  const int kk = (int)(blockIdx.x & 7u) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
  if (kk >= gp.nk) return;
  const int64_t k = gp.k0 + kk;
  const double* mdl = b.model + (size_t)pb * mdl_size(sh);
  double* Lt = lds;                                               // lower tiles of PSkp -> L; before that: PS_k in the layout of PF
  double* Li = Lt + gainm_union_doubles(NTL, sh);                 // inverses of the diagonal factors
  double* sA = Li + (size_t)NTL * 256;                            // [M][16]
  int* ibsz = reinterpret_cast<int*>(sA + (size_t)MAXM * 16);     // [MAXM]
  int* ioff = ibsz + MAXM;                                        // [MAXM + 1]  (the kernel argument itself is only indexed statically:
                                                                  // one dynamic index and the whole struct is copied to scratch)
  int* flag = ioff + MAXM + 1;                                    // [2]
  // developer diagnostics (NAGP_STAMPS): cycles per phase of lane 0 of column wave 0 ([0..15]) and of the chain wave ([16..31]) of every 64th
  // workgroup -- slot 0 staging | 1 barrier waits of the prologue | 2 B' (chain: delta_k) | 3 PSkp | 4 Delta (chain: tile 0) | 5 trailing tasks
  // (chain: its four products) | 6 chain: factor + invert | 7 forward row | 8 interval barrier wait | 9 flag, retry | 10 backward | 11 G store
  // | 12 workgroups sampled
  const bool stamp = gp.stamps && lane == 0 && (w == 0 || chain) && (kk & 63) == 0;
  unsigned long long st_a = stamp ? __builtin_readcyclecounter() : 0ull;
#define GM_STAMP(slot) do { if (stamp) { const unsigned long long st_b = __builtin_readcyclecounter(); atomicAdd(&gp.stamps[(chain ? 16 : 0) + (slot)], st_b - st_a); st_a = st_b; } } while (0)

  const double* PFk = b.PF + ((size_t)pb * T + k) * pf_step_doubles(sh);
  const double* PFk1 = PFk + pf_step_doubles(sh);
  const size_t SS = (size_t)Sp * Sp;
  const size_t gstep = gd_step_doubles(Sp, gp.dpacked);
  double* Gout = b.Gbuf + (b.gpstride ? (size_t)pb * b.gpstride + (size_t)kk * gstep : ((size_t)pb * gp.chunk + kk) * gstep);
  double* Dout = Gout + SS;
  const int NTOT = gp.dpacked ? NLOW : NTL * NTL;                 // tiles of Delta that are stored
  auto dtile = [&](int q, int& I, int& K) {
    if (gp.dpacked) { I = 0; while ((I + 1) * (I + 2) / 2 <= q) ++I; K = q - I * (I + 1) / 2; }
    else { I = q / NTL; K = q - I * NTL; }
  };
  // validity of this lane's rows and columns: bit br of rowbits <=> row kq of block br exists; bit K of colbits <=> column i of tile
  // column K exists (block 4K + (i >> 2), column i & 3).  Block sizes straight from the kernel arguments (static indices)
  unsigned long long rowbits = 0;
  unsigned colbits = 0;
#pragma unroll
  for (int br = 0; br < 4 * NTL; ++br)
    if (br < M && kq0 < sh.bsz[br]) rowbits |= 1ull << br;
#pragma unroll
  for (int K = 0; K < NTL; ++K) {
    const int q = i0 >> 2;
    const int bs = (q == 0) ? sh.bsz[4 * K] : (q == 1) ? sh.bsz[4 * K + 1] : (q == 2) ? sh.bsz[4 * K + 2] : sh.bsz[4 * K + 3];
    if (4 * K + q < M && (i0 & 3) < bs) colbits |= 1u << K;
  }
  if (tid < MAXM) {
    int bs = 0, of = 0;
#pragma unroll
    for (int q = 0; q < 4 * NTL; ++q) if (q == tid) { bs = sh.bsz[q]; of = sh.off[q]; }
    ibsz[tid] = bs; ioff[tid] = (tid >= M) ? S : of;
  }
  if (tid == MAXM) ioff[MAXM] = S;
  if (tid == 0) { flag[0] = 0; flag[1] = 0; }

  const int bc0 = 4 * c + (i0 >> 2);                              // block of this lane's column (column waves)
  const bool bcin = !chain && bc0 < M;
  const bool colok = !chain && ((colbits >> c) & 1u);
  double qd = 0.0;                                                // entry (kq, ci) of this column's diagonal block of Q (loaded beside the staged copy)
  const int diag_t = (((i0 - kq0) & 3) == 0 && i0 >= kq0) ? ((i0 - kq0) >> 2) : -1;             // register t with 4t + kq == i (the tile's diagonal)
  // staged copy of PS_k (the layout of PF: pf_off): every 16-byte piece of the step in flight at once
  // (the A blocks of the model and this lane's entry of Q ride along on the first attempt: every global load of the prologue is in
  // flight at once, and nothing but kernel arguments is needed to issue them)
  auto stage = [&](bool first) {
    const double2* src = reinterpret_cast<const double2*>(PFk);
    double2* dst = reinterpret_cast<double2*>(Lt);
    const int n2 = (int)(pf_step_doubles(sh) / 2);
    const int tq = gm_opaque(tid);      // (addresses formed here: hoisted out of the retry loop they are spilled, and every reload waits for the load before it)
    static_assert(NLD <= 10, "the pin below takes ten pieces");
    double2 v[10];
#pragma unroll
    for (int u = 0; u < 10; ++u) { const int q = tq + u * NT; v[u] = (u < NLD) ? src[q < n2 ? q : n2 - 1] : make_double2(0.0, 0.0); }      // unconditional loads
    double2 va = make_double2(0.0, 0.0);
    double vq = 0.0;
    if (first) {
      va = reinterpret_cast<const double2*>(mdl + mdl_A(sh))[tq < M * 8 ? tq : 0];
      vq = mdl[mdl_Q(sh) + (size_t)(bcin ? bc0 : 0) * 16 + 4 * kq0 + (i0 & 3)];
    }
    // ONE statement that needs all of them: the compiler otherwise sinks every load into the conditional store below -- load, wait, LDS
    // write, next load: ten HBM round trips in a row (38 k cycles of a 260 k-cycle step)
    asm volatile("" : "+v"(v[0].x), "+v"(v[0].y), "+v"(v[1].x), "+v"(v[1].y), "+v"(v[2].x), "+v"(v[2].y), "+v"(v[3].x), "+v"(v[3].y), "+v"(v[4].x), "+v"(v[4].y),
                      "+v"(v[5].x), "+v"(v[5].y), "+v"(v[6].x), "+v"(v[6].y), "+v"(v[7].x), "+v"(v[7].y), "+v"(v[8].x), "+v"(v[8].y), "+v"(v[9].x), "+v"(v[9].y),
                      "+v"(va.x), "+v"(va.y), "+v"(vq));
#pragma unroll
    for (int u = 0; u < NLD; ++u) { const int q = tq + u * NT; if (q < n2) dst[q] = v[u]; }
    if (first) {
      if (tq < M * 8) reinterpret_cast<double2*>(sA)[tq] = va;
      qd = colok ? vq : 0.0;
    }
  };

  // The two roles are two separate code paths from here to the end of the kernel (each with its own copy of the retry loop and the
  // same sequence of barriers): in one control flow graph the tile column of the column waves stays live through the chain wave's
  // blocks and the other way round, and the 168 registers of a three-waves-per-SIMD launch do not hold both.
  if (chain) {
    __builtin_amdgcn_s_setprio(3);      // the chain's instructions go in front of its SIMD-mates' (an FP64 MFMA holds the SIMD's issue for its duration)
    for (int attempt = 0; attempt < 2; ++attempt) {
      const int i = gm_opaque(i0), kq = gm_opaque(kq0);
      stage(attempt == 0);
      GM_STAMP(0);
      lds_barrier();                                                                     // b1
      GM_STAMP(1);
      // ---- delta_k = MF_{k+1} - A MF_k  (the chain wave has nothing else to do yet)
      if (attempt == 0)
        for (int s = lane; s < S; s += 64) {
          int blk = 0;
          while (ioff[blk + 1] <= s) ++blk;
          const int row = s - ioff[blk];
          const double* mf = b.MF + ((size_t)pb * T + k) * S;
          double acc = mf[S + s];
          for (int l = 0; l < ibsz[blk]; ++l) acc = fma(-sA[(size_t)blk * 16 + 4 * row + l], mf[ioff[blk] + l], acc);
          b.dbuf[((size_t)pb * gp.chunk + kk) * S + s] = acc;
        }
      GM_STAMP(2);
      lds_barrier();                                                                     // b2
      lds_barrier();                                                                     // b3: PSkp complete
      GM_STAMP(1);
      if (!gm_chol_inv_tile(Lt + gm_tix(0, 0), Li, lane)) { if (lane == 0) flag[attempt] = 1; }
      GM_STAMP(4);
      lds_barrier();                                                                     // b4: inv(L_00) published
      GM_STAMP(1);
#pragma unroll 1
      for (int J = 0; J < NTL; ++J) {
        if (J + 1 < NTL) {
          double* s10 = Lt + gm_tix(J + 1, J);
          double* s11 = Lt + gm_tix(J + 1, J + 1);
          v4d a1 = gm_load_acc(s10, i, kq), a2 = gm_load_acc(s11, i, kq);
          if (J >= 1) {      // column J-1 (finished by the other waves in the interval before)
            a1 = gm_mma_xyT(Lt + gm_tix(J + 1, J - 1), Lt + gm_tix(J, J - 1), i, kq, a1, true);
            a2 = gm_mma_xyT(Lt + gm_tix(J + 1, J - 1), Lt + gm_tix(J + 1, J - 1), i, kq, a2, true);
          }
          gm_store_acc(s10, a1, i, kq);
          gm_wave_fence();
          v4d l = {0.0, 0.0, 0.0, 0.0};
          l = gm_mma_xyT(s10, Li + (size_t)J * 256, i, kq, l, false);       // L_{J+1,J} = A_{J+1,J} inv(L_JJ)'
          gm_store_acc(s10, l, i, kq);
          gm_wave_fence();
          a2 = gm_mma_xyT(s10, s10, i, kq, a2, true);
          gm_store_acc(s11, a2, i, kq);
          gm_wave_fence();
          GM_STAMP(5);
          if (!gm_chol_inv_tile(s11, Li + (size_t)(J + 1) * 256, lane)) { if (lane == 0) flag[attempt] = 1; }
          GM_STAMP(6);
        }
        lds_barrier();
        GM_STAMP(8);
      }
      if (flag[attempt] == 0) break;
      lds_barrier();
    }
    GM_STAMP(9);
    if (stamp) atomicAdd(&gp.stamps[16 + 12], 1ull);
    return;
  }

  // ================================================= column waves =================================================
  v4d Y[NTL];
  for (int attempt = 0; attempt < 2; ++attempt) {
    const int i = gm_opaque(i0), kq = gm_opaque(kq0);
    const int ci = i & 3;
    const int bcc = bcin ? 4 * c + (i >> 2) : 0;                  // (clamped: lanes of padding columns read block 0 and are masked)
    stage(attempt == 0);
    GM_STAMP(0);
    lds_barrier();                                                                       // b1
    GM_STAMP(1);
    // INV: only the tiles of PSkp this wave forms are computed (tile row I of B' = A PS_k, then PSkp(I, c) = B' A_c' by quad broadcasts),
    // kept in registers until the staged copy is dead
    const auto mine = [&](int I) -> bool { return (I < c) ? (((c - I) & 1) == 0) : (I == c || ((I - c) & 1) != 0); };
    {
      // ---- B' = A PS_k, tile column c: element (16I + 4t + kq, 16c + i) = row kq of block br = 4I+t times column ci of PS(br, bc).
      // Half a tile row at a time, its PS entries and rows of A in flight together.  Tiles below the diagonal tile read the stored
      // lower tile (br, bc) -- element 4l + ci, four 8-byte reads a quarter-tile apart; tiles above it the mirror tile (bc, br) --
      // elements 4ci + l, two 16-byte reads; the diagonal tile selects per lane.
      const int lane_lo = (ci >> 1) * 128 + (ci & 1);            // offset of element ci inside a stored tile (pf_off)
      const int lane_up = ci * 256;                               // offset of element 4 ci
      const int Tbc = bcc * (bcc + 1) / 2;
      const double* arow = sA + 4 * kq;
#pragma unroll
      for (int I = 0; I < NTL; ++I) {
        if (INV && !mine(I)) continue;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          double p[2][4];
          double2 a01[2], a23[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int br = 4 * I + 2 * h + u, brc = br < M ? br : 0;      // (uniform)
            const double2* ap = reinterpret_cast<const double2*>(arow + brc * 16);
            a01[u] = ap[0]; a23[u] = ap[1];
          }
          if (I > c) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int br = 4 * I + 2 * h + u, brc = br < M ? br : 0;
              const double* q = Lt + gm_tpart(brc * (brc + 1) / 2 + bcc) + lane_lo;
#pragma unroll
              for (int l = 0; l < 4; ++l) p[u][l] = q[l * 256];
            }
          } else if (I < c) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int br = 4 * I + 2 * h + u, brc = br < M ? br : 0;
              const double2* q = reinterpret_cast<const double2*>(Lt + gm_tpart(Tbc + brc) + lane_up);
              const double2 d0 = q[0], d1 = q[64];
              p[u][0] = d0.x; p[u][1] = d0.y; p[u][2] = d1.x; p[u][3] = d1.y;
            }
          } else {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int br = 4 * I + 2 * h + u, brc = br < M ? br : 0;
              const bool lo = brc >= bcc;
              const double* q = Lt + (lo ? gm_tpart(brc * (brc + 1) / 2 + bcc) + lane_lo : gm_tpart(Tbc + brc) + lane_up);
              p[u][0] = q[0]; p[u][1] = q[lo ? 256 : 1]; p[u][2] = q[lo ? 512 : 128]; p[u][3] = q[lo ? 768 : 129];
            }
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int t = 2 * h + u;
            double v = a01[u].x * p[u][0];
            v = fma(a01[u].y, p[u][1], v); v = fma(a23[u].x, p[u][2], v); v = fma(a23[u].y, p[u][3], v);
            v = (((rowbits >> (4 * I + t)) & 1ull) && colok) ? v : 0.0;
            gm_pin(v);
            Y[I][t] = v;
          }
          __builtin_amdgcn_sched_barrier(0);      // half a tile row at a time: the reads of all ten rows hoisted together spill
        }
      }
    }
    GM_STAMP(2);
    lds_barrier();                                                                       // b2: the staged PS_k is dead
    GM_STAMP(1);
    double dpf[ND][4];
    {
      // ---- PSkp = B' A' + Q (+ jitter), lower tiles (I >= c) -> LDS.  sum_l B'[r][(bc,l)] A_bc[ci][l]: the four columns of block bc
      // sit in the four lanes of this lane's quad
      double abc[4];
      {
        const double2* ap = reinterpret_cast<const double2*>(sA + bcc * 16 + 4 * ci);
        const double2 a0 = ap[0], a1 = ap[1];
        abc[0] = a0.x; abc[1] = a0.y; abc[2] = a1.x; abc[3] = a1.y;
      }
      // tile (I, J), I > J, comes out of the registers of wave J (rows of tile row I of its column) or, transposed, of wave I (rows of
      // tile row J of ITS column, PSkp is symmetric): dealt by the parity of I - J, five or six tiles per wave instead of ten on wave 0
      // and one on the last
#pragma unroll
      for (int I = 0; I < NTL; ++I) {
        const bool mirror = I < c;                                // this wave forms tile (I, c) of the upper triangle and stores it as (c, I)
        if (!mine(I)) continue;
        v4d ps;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const double bp = Y[I][t];
          const double b0 = gm_quad<0x00>(bp), b1 = gm_quad<0x55>(bp), b2 = gm_quad<0xAA>(bp), b3 = gm_quad<0xFF>(bp);
          double v = abc[0] * b0;
          v = fma(abc[1], b1, v); v = fma(abc[2], b2, v); v = fma(abc[3], b3, v);
          ps[t] = (((rowbits >> (4 * I + t)) & 1ull) && colok) ? v : 0.0;
        }
        if (I == c) {      // the diagonal tile: Q (+ jitter) on the diagonal blocks, identity in the padding
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const bool ok = ((rowbits >> (4 * I + t)) & 1ull) && colok;
            if (ok && t == (i >> 2)) ps[t] += (attempt == 1 && kq == ci) ? qd + 0.01 * 0.5 : qd;      // sqrt(1e-4)*diag(rand): deterministic 0.5 in place of rand
            if (!ok && t == diag_t) ps[t] = 1.0;
          }
        }
        if (mirror) gm_store_accT(Lt + gm_tix(c, I), ps, i, kq);
        else gm_store_acc(Lt + gm_tix(I, c), ps, i, kq);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- PS_{k+1} for Delta: the loads go out now (indices clamped into the stored tiles: the padding is masked below) and land
      // while the chain wave factors tile 0
      if (attempt == 0) {
#pragma unroll
        for (int n = 0; n < ND; ++n) {
          const int q = c + NTL * n;
          int I = 0, K = 0;
          if (q < NTOT) dtile(q, I, K);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int br = 4 * I + t, bcK = 4 * K + (i >> 2);
            dpf[n][t] = pf_elem(PFk1, br < M ? br : M - 1, bcK < M ? bcK : M - 1, kq, i & 3);
          }
        }
      }
    }
    GM_STAMP(3);
    lds_barrier();                                                                       // b3: PSkp complete
    GM_STAMP(1);
    if (attempt == 0) {
      // ---- Delta_k = PS_{k+1} - PSkp, tiles dealt round robin over the column waves
      for (int n0 = 0; n0 * NTL + c < NTOT; n0 += ND) {
#pragma unroll
        for (int n = 0; n < ND; ++n) {
          const int q = c + NTL * (n0 + n);
          if (q >= NTOT) continue;
          int I, K;
          dtile(q, I, K);
          const v4d pk = (I >= K) ? gm_load_acc(Lt + gm_tix(I, K), i, kq) : gm_load_accT(Lt + gm_tix(K, I), i, kq);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const bool ok = ((rowbits >> (4 * I + t)) & 1ull) && ((colbits >> K) & 1u);
            const int br = 4 * I + t, bcK = 4 * K + (i >> 2);
            const double pf1 = (n0 == 0) ? dpf[n][t] : pf_elem(PFk1, br < M ? br : M - 1, bcK < M ? bcK : M - 1, kq, i & 3);
            const double d = ok ? pf1 - pk[t] : 0.0;
            if (gp.dpacked) Dout[(size_t)(I * (I + 1) / 2 + K) * 256 + (4 * t + kq) * 16 + i] = d;
            else Dout[(size_t)(16 * I + 4 * t + kq) * Sp + 16 * K + i] = d;
          }
        }
      }
    }
    GM_STAMP(4);
    lds_barrier();                                                                       // b4: inv(L_00) published, Delta has read PSkp
    GM_STAMP(1);
    // ---- NTL intervals, one barrier each
#pragma unroll
    for (int J = 0; J < NTL; ++J) {
      // trailing update with column J-1 (+ the panel product for the tiles of column J); the chain wave owns (J+1,J), (J+1,J+1)
      // (dealt round robin over the column waves EXCEPT the one on the chain's SIMD -- wave 7 when there is one: every MFMA
      // it issues holds that SIMD and delays the chain)
      constexpr int NDEAL = (NTL >= 7) ? NTL - 1 : NTL;
      constexpr int MATE = (NTL >= 7) ? CMAP.col[7] : -1;          // the column of the chain's SIMD-mate
      const int cdeal = (NTL >= 7) ? (c == MATE ? -1 : (c > MATE ? c - 1 : c)) : c;
      int cnt = 0;
      for (int K = J; K < (J == 0 ? 1 : NTL); ++K)
        for (int I = K; I < NTL; ++I) {
          if (I == K && (K == J || K == J + 1)) continue;
          if (K == J && I == J + 1) continue;
          if ((cnt++) % NDEAL != cdeal) continue;
          double* slot = Lt + gm_tix(I, K);
          v4d acc = gm_load_acc(slot, i, kq);
          if (J >= 1) acc = gm_mma_xyT(Lt + gm_tix(I, J - 1), Lt + gm_tix(K, J - 1), i, kq, acc, true);
          if (K == J) {
            gm_store_acc(slot, acc, i, kq);
            gm_wave_fence();
            v4d z = {0.0, 0.0, 0.0, 0.0};
            acc = gm_mma_xyT(slot, Li + (size_t)J * 256, i, kq, z, false);
          }
          gm_store_acc(slot, acc, i, kq);
        }
      GM_STAMP(5);
      // forward solve, row J: Y_J = inv(L_JJ) (B'_J - sum_{K<J} L_JK Y_K)
      if constexpr (!INV) {
        v4d acc = Y[J];
#pragma unroll
        for (int K = 0; K < NTL; ++K)
          if (K < J) acc = gm_mma_xb(Lt + gm_tix(J, K), Y[K], i, kq, acc, true);
        v4d y = {0.0, 0.0, 0.0, 0.0};
        Y[J] = gm_mma_xb(Li + (size_t)J * 256, acc, i, kq, y, false);
      } else {
        // tile column c of L^-1 (right-hand side: the identity): zero above tile row c, inv(L_cc) in it, a forward row below
        if (J == c) Y[J] = gm_load_acc(Li + (size_t)J * 256, i, kq);
        else if (J > c) {
          // (the products of a row are independent of each other: three accumulators in turn -- the first tile columns hold most of the
          // triangle and have their SIMD almost to themselves, where one accumulator chain runs at the latency of a dependent MFMA)
          v4d acc = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int K = 0; K < NTL; ++K)
            if (K < J && K >= c) {
              if (K % 3 == 0) acc = gm_mma_xb(Lt + gm_tix(J, K), Y[K], i, kq, acc, true);
              else if (K % 3 == 1) acc1 = gm_mma_xb(Lt + gm_tix(J, K), Y[K], i, kq, acc1, true);
              else acc2 = gm_mma_xb(Lt + gm_tix(J, K), Y[K], i, kq, acc2, true);
            }
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] += acc1[t] + acc2[t];
          v4d y = {0.0, 0.0, 0.0, 0.0};
          Y[J] = gm_mma_xb(Li + (size_t)J * 256, acc, i, kq, y, false);
        }
      }
      GM_STAMP(7);
      lds_barrier();
      GM_STAMP(8);
    }
    if (flag[attempt] == 0) break;
    lds_barrier();
  }
  if (w == 0 && lane == 0) {
    if (flag[0]) atomicAdd(&b.counters[(size_t)pb * 4 + 0], 1ull);
    if (flag[0] && flag[1]) atomicAdd(&b.counters[(size_t)pb * 4 + 3], 1ull);
  }
  GM_STAMP(9);
  const int i = gm_opaque(i0), kq = gm_opaque(kq0);

  if constexpr (!INV) {
  // ---- backward: W_I = inv(L_II)' (Y_I - sum_{K>I} L_KI' W_K)
#pragma unroll
  for (int I = NTL - 1; I >= 0; --I) {
    v4d acc = Y[I];
#pragma unroll
    for (int K = 0; K < NTL; ++K)
      if (K > I) { acc = gm_mma_xTb(Lt + gm_tix(K, I), Y[K], i, kq, acc, true); if (K & 1) __builtin_amdgcn_sched_barrier(0); }
    v4d wv = {0.0, 0.0, 0.0, 0.0};
    Y[I] = gm_mma_xTb(Li + (size_t)I * 256, acc, i, kq, wv, false);
    __builtin_amdgcn_sched_barrier(0);
  }
  GM_STAMP(10);
  // ---- G = W': G[16c + i][16I + 4t + kq] = W_I[4t+kq][i] (32-byte runs); padding rows / columns cleaned
  const int gbase = (16 * c + i) * Sp + kq;
#pragma unroll
  for (int I = 0; I < NTL; ++I) {
#pragma unroll
    for (int t = 0; t < 4; ++t) Gout[gbase + 16 * I + 4 * t] = (((rowbits >> (4 * I + t)) & 1ull) && colok) ? Y[I][t] : 0.0;
  }
  } else {
  // ---- the block operands of G = A^-1 - (A^-1 Q) X: loaded now, they land under the backward solve
  const double* ai = gp.ainv + (size_t)pb * M * 32;
  const int bci = bcin ? 4 * c + (i >> 2) : 0;
  double wd[NTL], wq[4];
  // (after a jitter retry the matrix that was inverted is PSkp + J, J = 0.005 I, while PS_k A' is still A^-1 (PSkp - Q): the reference's
  // G = PS_k A' / chol(PSkp + J) is then A^-1 - A^-1 (Q + J) X, i.e. the block factor gains 0.005 A^-1)
  const double jit = (flag[0] != 0) ? 0.01 * 0.5 : 0.0;
#pragma unroll
  for (int I = 0; I < NTL; ++I) {                                                                                                          // (A^-1 (Q + J))_br [i & 3][kq]
    const int br = 4 * I + (i >> 2);
    const double* e = ai + (size_t)(br < M ? br : 0) * 32 + 4 * (i & 3) + kq;
    wd[I] = fma(jit, e[0], e[16]);
  }
#pragma unroll
  for (int l = 0; l < 4; ++l) { const double* e = ai + (size_t)bci * 32 + 4 * (i & 3) + l; wq[l] = fma(jit, e[0], e[16]); }                // (A^-1 (Q + J))_bc [ci][l]
  const double aid = ai[(size_t)bci * 32 + 4 * kq + (i & 3)];                                                                              // (A^-1)_bc [kq][ci]
  // ---- backward, tile rows >= c: X_I = inv(L_II)' (Y_I - sum_{K>I} L_KI' X_K)
#pragma unroll
  for (int I = NTL - 1; I >= 0; --I) {
    if (I < c) continue;
    v4d acc = Y[I], acc1 = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int K = 0; K < NTL; ++K)
      if (K > I) {
        if (K % 3 == 0) acc = gm_mma_xTb(Lt + gm_tix(K, I), Y[K], i, kq, acc, true);
        else if (K % 3 == 1) acc1 = gm_mma_xTb(Lt + gm_tix(K, I), Y[K], i, kq, acc1, true);
        else acc2 = gm_mma_xTb(Lt + gm_tix(K, I), Y[K], i, kq, acc2, true);
      }
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] += acc1[t] + acc2[t];
    v4d wv = {0.0, 0.0, 0.0, 0.0};
    Y[I] = gm_mma_xTb(Li + (size_t)I * 256, acc, i, kq, wv, false);
    __builtin_amdgcn_sched_barrier(0);
  }
  GM_STAMP(10);
  // ---- G from the lower tiles X(I, c), I >= c, of this wave.  Tile (I, c) of G = [I == c] A^-1 - blockdiag(A^-1 Q)_I X(I, c): one product
  // on the matrix cores with the block-diagonal factor as the A operand (k-step s carries block s: one value per lane), stored as
  // 128-byte rows; tile (c, I), I > c, = -(A^-1 Q)_c X(c, I) = -(A^-1 Q)_c X(I, c)': the four columns of a block sit in a quad, stored
  // as 32-byte runs.
#pragma unroll
  for (int I = 0; I < NTL; ++I) {
    if (I < c) continue;
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    if (I == c) {
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = (t == (i >> 2)) ? aid : 0.0;
    }
#pragma unroll
    for (int sx = 0; sx < 4; ++sx) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(((i >> 2) == sx) ? -wd[I] : 0.0, Y[I][sx], acc, 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bool ok = ((rowbits >> (4 * I + t)) & 1ull) && colok;
      Gout[(size_t)(16 * I + 4 * t + kq) * Sp + 16 * c + i] = ok ? acc[t] : 0.0;
      if (I > c) {
        const double xv = Y[I][t];
        const double b0 = gm_quad<0x00>(xv), b1 = gm_quad<0x55>(xv), b2 = gm_quad<0xAA>(xv), b3 = gm_quad<0xFF>(xv);
        double g = wq[0] * b0;
        g = fma(wq[1], b1, g); g = fma(wq[2], b2, g); g = fma(wq[3], b3, g);
        Gout[(size_t)(16 * c + i) * Sp + 16 * I + 4 * t + kq] = ok ? -g : 0.0;
      }
    }
  }
  }
  GM_STAMP(11);
  if (stamp) atomicAdd(&gp.stamps[12], 1ull);
#undef GM_STAMP
}

}  // namespace nagp
