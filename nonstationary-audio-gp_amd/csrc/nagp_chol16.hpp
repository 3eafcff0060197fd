// nagp_chol16.hpp -- Cholesky factor of one 16x16 tile AND the inverse of that factor, by ONE wave, entirely in registers.
//
// The diagonal tiles are the dependence chain of the blocked Cholesky of PSkp = A PS_k A' + Q (gf_ep_modulator_nmf.m:213-221,
// `chol(PSkp,'lower')`): tile J+1 cannot start before tile J has been factored, inverted, multiplied into the sub-diagonal tile
// and that one squared into tile J+1.  Its length, not the matrix-core work, is what a gain step costs, so the tile is factored
// with no LDS traffic and no cross-wave synchronisation at all:
//   * lane l holds ROW (l & 15) of the tile in sixteen registers (the four 16-lane rows of the wave hold identical copies: DPP
//     broadcasts act inside a row of 16 lanes, and redundancy costs nothing on a lone wave);
//   * column j:  p = a_jj of lane j (v_mov_b64_dpp row_newbcast:j),  r = 1/sqrt(p) (v_rsq_f64 + two Newton steps),  l_ij = a_ij r,
//     trailing update  a_ik -= l_ij l_kj  with l_kj = lane k's l_.j through one DPP broadcast per (j, k);
//   * the inverse X = L^-1 rides along: unscaled rows U_i = e_i - sum_{k<i} L_ik X_k, X_k = r_k U_k; at column j every lane i > j
//     takes  U_i -= (l_ij r_j) U_j  with U_j broadcast from lane j, one DPP broadcast and one multiply-add per entry -- and the four lane
//     rows, which all know L, each keep a QUARTER of the columns of X (column 4r + g in register r of lane row g); the rows are
//     scaled by their own r at the end.
// 120 + 40 broadcast + multiply-add pairs and 16 reciprocal square roots per tile (120 + 136 with the inverse kept redundantly: 2.26 us per tile on a lone wave by the event clock
// (tools/ubench/chol16.hip; ~840 instructions).  The panel products and both triangular solves of the gain
// kernel multiply by inv(L_JJ) on the matrix cores instead of substituting through L_JJ, so the factor itself is never stored.
#pragma once
#include "nagp_dev.hpp"

namespace nagp {

// value of lane K of this lane's row of 16 lanes
template <int K>
__device__ __forceinline__ double c16_nb(double v) { return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + K, 0xF, 0xF, true); }
__device__ __forceinline__ double c16_nbk(double v, int k) {
  switch (k) {
    case 0: return c16_nb<0>(v); case 1: return c16_nb<1>(v); case 2: return c16_nb<2>(v); case 3: return c16_nb<3>(v);
    case 4: return c16_nb<4>(v); case 5: return c16_nb<5>(v); case 6: return c16_nb<6>(v); case 7: return c16_nb<7>(v);
    case 8: return c16_nb<8>(v); case 9: return c16_nb<9>(v); case 10: return c16_nb<10>(v); case 11: return c16_nb<11>(v);
    case 12: return c16_nb<12>(v); case 13: return c16_nb<13>(v); case 14: return c16_nb<14>(v); default: return c16_nb<15>(v);
  }
}
// d -= (lane K's a) * b  (v_mov_b64_dpp + v_fma_f64; the fused v_fmac_f64_dpp through inline asm, -DNAGP_C16_ASM, measured SLOWER -- 2.63 against
// 2.26 us per tile, tools/ubench/chol16.hip: the wait states of a DPP operand have to be inside the string for every statement)
#ifdef NAGP_C16_ASM
template <int K>
__device__ __forceinline__ void c16_fmsub(double& d, double a, double b) {
  asm("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(a), "v"(b), "n"(K));
}
#else
template <int K>
__device__ __forceinline__ void c16_fmsub(double& d, double a, double b) { d = fma(-c16_nb<K>(a), b, d); }
#endif
__device__ __forceinline__ void c16_fmsubk(double& d, double a, double b, int k) {
  switch (k) {
    case 0: c16_fmsub<0>(d, a, b); break; case 1: c16_fmsub<1>(d, a, b); break; case 2: c16_fmsub<2>(d, a, b); break;
    case 3: c16_fmsub<3>(d, a, b); break; case 4: c16_fmsub<4>(d, a, b); break; case 5: c16_fmsub<5>(d, a, b); break;
    case 6: c16_fmsub<6>(d, a, b); break; case 7: c16_fmsub<7>(d, a, b); break; case 8: c16_fmsub<8>(d, a, b); break;
    case 9: c16_fmsub<9>(d, a, b); break; case 10: c16_fmsub<10>(d, a, b); break; case 11: c16_fmsub<11>(d, a, b); break;
    case 12: c16_fmsub<12>(d, a, b); break; case 13: c16_fmsub<13>(d, a, b); break; case 14: c16_fmsub<14>(d, a, b); break;
    default: c16_fmsub<15>(d, a, b); break;
  }
}

// a[c] = entry (row, c) of the symmetric positive definite tile (only c <= row is read), identical in the four 16-lane rows of the wave.
// On return x[r] = entry (row, 4r + g) of inv(L), L = chol(tile, 'lower'), g = lane >> 4: the INVERSE is split over the four lane rows by
// columns -- a DPP broadcast stays inside a lane row, and lane j of row g holds exactly the columns of U_j that row needs -- so the 136
// broadcast + multiply-add pairs of the redundant form are 40.  Entries above the diagonal are exact zeros.  Returns false (to every
// lane) when a pivot was not positive (NaN included): the caller reports it (jitter retry, gf_ep_modulator_nmf.m:214-221).
__device__ __forceinline__ bool chol16_inv_rows(double (&a)[16], double (&x)[4], int row, int g) {
#pragma unroll
  for (int r = 0; r < 4; ++r) x[r] = (4 * r + g == row) ? 1.0 : 0.0;
  bool ok = true;
  double rown = 1.0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const double p = c16_nbk(a[j], j);
    ok = ok && (p > 0.0);
    const double r = rsqrt_nr(p);
    const double lj = (row >= j) ? a[j] * r : 0.0;
    a[j] = lj;
    if (row == j) rown = r;
#pragma unroll
    for (int k = j + 1; k < 16; ++k) c16_fmsubk(a[k], lj, lj, k);            // a_ik -= l_kj l_ij
    const double am = (row > j) ? lj * r : 0.0;
    // U_i -= (l_ij r_j) U_j over the columns c = 4r + g <= j of this lane row
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      if (4 * rr + 3 <= j) c16_fmsubk(x[rr], x[rr], am, j);                    // (every lane row)
      else if (4 * rr <= j) c16_fmsubk(x[rr], x[rr], (4 * rr + g <= j) ? am : 0.0, j);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) x[r] *= rown;
  return ok;
}

}  // namespace nagp
