// Standalone check and timing of rts_gain_mfma_kernel (nagp_gain_mfma.hpp) against a host computation in double: random SPD
// filtered covariances in the compact layout of PF, block-diagonal A, Q; one launch over nk steps.  Developer tool.
// build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -I nonstationary-audio-gp_amd/csrc -I include -o tools/ubench/gain_check tools/ubench/gain_check.hip
// run:   tools/ubench/gain_check [M=38] [nk=2048] [dpacked=1] [inv=0: 1 = the explicit-inverse form] [two=0: 2-state blocks]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "nagp_gain_mfma.hpp"
using namespace nagp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <int N, bool INV> static void launch(dim3 gr, size_t lds, const Shape& sh, const Bufs& b, const GainPar& gp) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(rts_gain_mfma_kernel<N, INV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  static bool said = false;
  if (!said) { int nb = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rts_gain_mfma_kernel<N, INV>, 64 * (N + 1), lds); hipFuncAttributes fa; hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(rts_gain_mfma_kernel<N, INV>)); printf("occupancy: %d workgroup(s) per CU; %d registers, %zu B scratch per lane\n", nb, fa.numRegs, (size_t)fa.localSizeBytes); said = true; }
  hipLaunchKernelGGL((rts_gain_mfma_kernel<N, INV>), gr, dim3(64 * (N + 1)), lds, 0, sh, b, gp);
}
template <bool INV> static void launch_n(int ntl, dim3 gr, size_t lds, const Shape& sh, const Bufs& b, const GainPar& gp) {
  switch (ntl) { case 1: launch<1, INV>(gr, lds, sh, b, gp); break; case 2: launch<2, INV>(gr, lds, sh, b, gp); break; case 3: launch<3, INV>(gr, lds, sh, b, gp); break;
    case 4: launch<4, INV>(gr, lds, sh, b, gp); break; case 5: launch<5, INV>(gr, lds, sh, b, gp); break; case 6: launch<6, INV>(gr, lds, sh, b, gp); break;
    case 7: launch<7, INV>(gr, lds, sh, b, gp); break; case 8: launch<8, INV>(gr, lds, sh, b, gp); break; case 9: launch<9, INV>(gr, lds, sh, b, gp); break;
    default: launch<10, INV>(gr, lds, sh, b, gp); break; }
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 38, nk = argc > 2 ? atoi(argv[2]) : 2048, dpacked = argc > 3 ? atoi(argv[3]) : 1;
  const int two = argc > 5 ? atoi(argv[5]) : 0;
  const int inv = argc > 4 ? atoi(argv[4]) : 0;      // 1: the explicit-inverse form (rts_gain_mfma_kernel<.., true>)
  const int dbg = argc > 6 ? atoi(argv[6]) : 0;
  Shape sh{};
  sh.M = M; sh.D = M - 2; sh.N = 2; sh.ntiles = M * M; sh.BS = 4; sh.Ms = M;
  int S = 0;
  for (int m = 0; m < M; ++m) { sh.bsz[m] = (two && m < M - 2) ? 2 : ((m % 5 == 4) ? 3 : 4); sh.off[m] = S; S += sh.bsz[m]; }
  sh.off[M] = S; sh.S = S;
  const int T = nk + 1; sh.T = T;
  const int Sp = ((4 * M + 15) / 16) * 16, ntl = Sp / 16;
  const size_t pfs = pf_step_doubles(sh), gstep = gd_step_doubles(Sp, dpacked), SS = (size_t)Sp * Sp;
  printf("M %d S %d Sp %d ntl %d nk %d dpacked %d form %s  lds %zu bytes\n", M, S, Sp, ntl, nk, dpacked, inv ? "explicit inverse" : "solves", gainm_lds_doubles(ntl, sh) * 8);
  srand(7);
  auto rnd = [] { return rand() / (double)RAND_MAX - 0.5; };
  // model: A, Q blocks (zero outside bs x bs)
  std::vector<double> mdl(mdl_size(sh), 0.0);
  for (int m = 0; m < M; ++m) {
    const int bs = sh.bsz[m];
    double R[16];
    for (int i = 0; i < 16; ++i) R[i] = rnd();
    for (int i = 0; i < bs; ++i)
      for (int j = 0; j < bs; ++j) {
        mdl[mdl_A(sh) + m * 16 + 4 * i + j] = (i == j ? 0.9 : 0.0) + 0.2 * rnd();
        double q = (i == j) ? 0.05 : 0.0;
        for (int l = 0; l < 4; ++l) q += 0.05 * R[4 * i + l] * R[4 * j + l];
        mdl[mdl_Q(sh) + m * 16 + 4 * i + j] = q;
      }
  }
  // nd distinct dense SPD matrices (state order), reused round robin over the steps
  const int nd = 5;
  std::vector<std::vector<double>> Pd(nd, std::vector<double>((size_t)S * S));
  for (int d = 0; d < nd; ++d) {
    std::vector<double> F((size_t)S * (S + 4));
    for (auto& v : F) v = rnd();
    for (int i = 0; i < S; ++i)
      for (int j = 0; j <= i; ++j) {
        double s = (i == j) ? 0.02 : 0.0;
        for (int l = 0; l < S + 4; ++l) s += F[(size_t)i * (S + 4) + l] * F[(size_t)j * (S + 4) + l] / (S + 4);
        Pd[d][(size_t)i * S + j] = Pd[d][(size_t)j * S + i] = s;
      }
  }
  std::vector<double> PF((size_t)T * pfs, 0.0), MF((size_t)T * S);
  for (auto& v : MF) v = rnd();
  for (int k = 0; k < T; ++k) {
    const std::vector<double>& P = Pd[k % nd];
    for (int I = 0; I < M; ++I)
      for (int J = 0; J <= I; ++J)
        for (int i = 0; i < sh.bsz[I]; ++i)
          for (int j = 0; j < sh.bsz[J]; ++j) PF[(size_t)k * pfs + pf_off(I * (I + 1) / 2 + J, 4 * i + j)] = P[(size_t)(sh.off[I] + i) * S + sh.off[J] + j];
  }
  Bufs b{};
  double *d_mdl, *d_PF, *d_MF, *d_G, *d_d; unsigned long long* d_cnt;
  CK(hipMalloc(&d_mdl, mdl.size() * 8)); CK(hipMalloc(&d_PF, PF.size() * 8)); CK(hipMalloc(&d_MF, MF.size() * 8));
  CK(hipMalloc(&d_G, (size_t)nk * gstep * 8)); CK(hipMalloc(&d_d, (size_t)nk * S * 8)); CK(hipMalloc(&d_cnt, 32));
  CK(hipMemcpy(d_mdl, mdl.data(), mdl.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_PF, PF.data(), PF.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_MF, MF.data(), MF.size() * 8, hipMemcpyHostToDevice)); CK(hipMemset(d_cnt, 0, 32)); CK(hipMemset(d_G, 0xff, (size_t)nk * gstep * 8));
  // per block A^-1 and A^-1 Q (zero padded), as nagp_api.hip: gain_inverse_blocks
  std::vector<double> ainv((size_t)M * 32, 0.0);
  for (int m = 0; m < M; ++m) {
    const int bs = sh.bsz[m];
    double a[16], x[16];
    for (int i = 0; i < 16; ++i) { a[i] = 0; x[i] = 0; }
    for (int i = 0; i < bs; ++i) for (int j = 0; j < bs; ++j) a[4 * i + j] = mdl[mdl_A(sh) + m * 16 + 4 * i + j];
    for (int i = 0; i < bs; ++i) x[4 * i + i] = 1.0;
    for (int col = 0; col < bs; ++col) {      // Gauss-Jordan with partial pivoting
      int piv = col; for (int r = col + 1; r < bs; ++r) if (fabs(a[4 * r + col]) > fabs(a[4 * piv + col])) piv = r;
      for (int j = 0; j < 4; ++j) { std::swap(a[4 * col + j], a[4 * piv + j]); std::swap(x[4 * col + j], x[4 * piv + j]); }
      const double d = 1.0 / a[4 * col + col];
      for (int j = 0; j < 4; ++j) { a[4 * col + j] *= d; x[4 * col + j] *= d; }
      for (int r = 0; r < bs; ++r) if (r != col) { const double f = a[4 * r + col]; for (int j = 0; j < 4; ++j) { a[4 * r + j] -= f * a[4 * col + j]; x[4 * r + j] -= f * x[4 * col + j]; } }
    }
    for (int i = 0; i < bs; ++i) for (int j = 0; j < bs; ++j) {
      ainv[(size_t)m * 32 + 4 * i + j] = x[4 * i + j];
      double w = 0; for (int l = 0; l < bs; ++l) w += x[4 * i + l] * mdl[mdl_Q(sh) + m * 16 + 4 * l + j];
      ainv[(size_t)m * 32 + 16 + 4 * i + j] = w;
    }
  }
  double* d_ainv; CK(hipMalloc(&d_ainv, ainv.size() * 8)); CK(hipMemcpy(d_ainv, ainv.data(), ainv.size() * 8, hipMemcpyHostToDevice));
  b.model = d_mdl; b.PF = d_PF; b.MF = d_MF; b.Gbuf = d_G; b.dbuf = d_d; b.counters = d_cnt; b.gpstride = 0;
  GainPar gp{}; gp.k0 = 0; gp.nk = nk; gp.chunk = nk; gp.dense_sp = Sp; gp.dpacked = dpacked; gp.dbg = dbg; gp.ainv = d_ainv;
  unsigned long long* d_st; CK(hipMalloc(&d_st, 32 * 8)); CK(hipMemset(d_st, 0, 32 * 8)); gp.stamps = d_st;
  const size_t lds = gainm_lds_doubles(ntl, sh) * 8;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    if (inv) launch_n<true>(ntl, dim3((nk + 7) / 8 * 8, 1), lds, sh, b, gp); else launch_n<false>(ntl, dim3((nk + 7) / 8 * 8, 1), lds, sh, b, gp);
    hipEventRecord(e1); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.33 * (double)S * S * S * 2 * nk;
    printf("rep %d: %.3f ms for %d steps = %.2f us per step-CU-slot (x256 CUs), %.2f TFLOP/s (2.33 S^3 fma) = %.1f %% of 78.6\n", rep, ms, nk, ms * 1e3 / nk * 256, flop / ms * 1e-9,
           flop / ms * 1e-9 / 78.6 * 100);
  }
  unsigned long long cnt[4]; CK(hipMemcpy(cnt, d_cnt, 32, hipMemcpyDeviceToHost));
  {
    unsigned long long st[32]; CK(hipMemcpy(st, d_st, sizeof(st), hipMemcpyDeviceToHost));
    const char* nm[12] = {"staging", "prologue barriers", "B' | delta_k", "PSkp", "Delta | tile 0", "trailing | 4 products", "factor+invert", "forward row", "interval barrier", "retry check", "backward", "G store"};
    for (int r = 0; r < 2; ++r) {
      printf("%s wave, cycles per workgroup (%llu sampled):", r ? "chain " : "column", st[16 * r + 12]);
      unsigned long long tot = 0;
      for (int q = 0; q < 12; ++q) { printf(" %s %llu |", nm[q], st[16 * r + 12] ? st[16 * r + q] / st[16 * r + 12] : 0ull); tot += st[16 * r + q]; }
      printf(" total %llu\n", st[16 * r + 12] ? tot / st[16 * r + 12] : 0ull);
    }
  }
  printf("counters: retries %llu, failures %llu (3 launches)\n", cnt[0], cnt[3]);
  // ---- host reference for the first nd steps
  std::vector<double> G((size_t)nd * gstep), dl((size_t)nd * S);
  CK(hipMemcpy(G.data(), d_G, G.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(dl.data(), d_d, dl.size() * 8, hipMemcpyDeviceToHost));
  auto dense_ix = [&](int s) { int m = 0; while (sh.off[m + 1] <= s) ++m; return 4 * m + (s - sh.off[m]); };
  double wG = 0, wD = 0, wd = 0, sG = 0, sD = 0, wpad = 0;
  for (int k = 0; k < nd && k < nk; ++k) {
    const std::vector<double>& P = Pd[k % nd]; const std::vector<double>& P1 = Pd[(k + 1) % nd];
    std::vector<double> A((size_t)S * S, 0.0), Q((size_t)S * S, 0.0), Bm((size_t)S * S), PK((size_t)S * S), L((size_t)S * S, 0.0), X((size_t)S * S), Gr((size_t)S * S);
    for (int m = 0; m < M; ++m)
      for (int i = 0; i < sh.bsz[m]; ++i)
        for (int j = 0; j < sh.bsz[m]; ++j) {
          A[(size_t)(sh.off[m] + i) * S + sh.off[m] + j] = mdl[mdl_A(sh) + m * 16 + 4 * i + j];
          Q[(size_t)(sh.off[m] + i) * S + sh.off[m] + j] = mdl[mdl_Q(sh) + m * 16 + 4 * i + j];
        }
    for (int i = 0; i < S; ++i) for (int j = 0; j < S; ++j) { double s = 0; for (int l = 0; l < S; ++l) s += P[(size_t)i * S + l] * A[(size_t)j * S + l]; Bm[(size_t)i * S + j] = s; }
    for (int i = 0; i < S; ++i) for (int j = 0; j < S; ++j) { double s = Q[(size_t)i * S + j]; for (int l = 0; l < S; ++l) s += A[(size_t)i * S + l] * Bm[(size_t)l * S + j]; PK[(size_t)i * S + j] = s; }
    for (int j = 0; j < S; ++j) {
      double s = PK[(size_t)j * S + j]; for (int l = 0; l < j; ++l) s -= L[(size_t)j * S + l] * L[(size_t)j * S + l];
      L[(size_t)j * S + j] = sqrt(s);
      for (int i = j + 1; i < S; ++i) { double v = PK[(size_t)i * S + j]; for (int l = 0; l < j; ++l) v -= L[(size_t)i * S + l] * L[(size_t)j * S + l]; L[(size_t)i * S + j] = v / L[(size_t)j * S + j]; }
    }
    for (int r = 0; r < S; ++r) {      // X L' = B ; G L = X
      for (int cidx = 0; cidx < S; ++cidx) { double v = Bm[(size_t)r * S + cidx]; for (int l = 0; l < cidx; ++l) v -= X[(size_t)r * S + l] * L[(size_t)cidx * S + l]; X[(size_t)r * S + cidx] = v / L[(size_t)cidx * S + cidx]; }
      for (int cidx = S - 1; cidx >= 0; --cidx) { double v = X[(size_t)r * S + cidx]; for (int l = cidx + 1; l < S; ++l) v -= Gr[(size_t)r * S + l] * L[(size_t)l * S + cidx]; Gr[(size_t)r * S + cidx] = v / L[(size_t)cidx * S + cidx]; }
    }
    const double* Gk = G.data() + (size_t)k * gstep; const double* Dk = Gk + SS;
    if (dbg == 1 || dbg == 2) {
      double w = 0; int wi = -1, wj = -1; int nbadp = 0;
      for (int i = 0; i < S; ++i)
        for (int j = 0; j < S; ++j) {
          const int di = dense_ix(i), dj = dense_ix(j);
          double ref, got;
          if (dbg == 1) { ref = Bm[(size_t)j * S + i]; got = Gk[(size_t)di * Sp + dj]; }     // B' = A PS = (PS A')'
          else { if ((di >> 4) < (dj >> 4)) continue; ref = PK[(size_t)i * S + j]; got = Dk[(size_t)((di >> 4) * ((di >> 4) + 1) / 2 + (dj >> 4)) * 256 + (di & 15) * 16 + (dj & 15)]; }
          if (!(fabs(got - ref) <= w)) { w = fabs(got - ref); wi = di; wj = dj; }
          if (k == 0 && !(fabs(got - ref) <= 1e-12) && nbadp++ < 40) printf("   bad (%d,%d): got %.6e ref %.6e\n", di, dj, got, ref);
        }
      printf("dbg %d step %d: worst |got - ref| %.3e at dense (%d, %d)\n", dbg, k, w, wi, wj);
      continue;
    }
    std::vector<char> seen(SS, 0);
    for (int i = 0; i < S; ++i)
      for (int j = 0; j < S; ++j) {
        const int di = dense_ix(i), dj = dense_ix(j);
        const double g = Gk[(size_t)di * Sp + dj];
        seen[(size_t)di * Sp + dj] = 1;
        wG = fmax(wG, fabs(g - Gr[(size_t)i * S + j])); sG = fmax(sG, fabs(Gr[(size_t)i * S + j]));
        const double dref = P1[(size_t)i * S + j] - PK[(size_t)i * S + j];
        double dv;
        if (dpacked) { const int TI = di >> 4, TJ = dj >> 4; if (TI < TJ) continue; dv = Dk[(size_t)(TI * (TI + 1) / 2 + TJ) * 256 + (di & 15) * 16 + (dj & 15)]; }
        else dv = Dk[(size_t)di * Sp + dj];
        wD = fmax(wD, fabs(dv - dref)); sD = fmax(sD, fabs(dref));
      }
    for (size_t e = 0; e < SS; ++e) if (!seen[e]) wpad = fmax(wpad, fabs(Gk[e]));
    for (int s = 0; s < S; ++s) {
      double v = MF[(size_t)(k + 1) * S + s];
      for (int l = 0; l < S; ++l) v -= A[(size_t)s * S + l] * MF[(size_t)k * S + l];
      wd = fmax(wd, fabs(v - dl[(size_t)k * S + s]));
    }
  }
  printf("max |G - host| %.3e (scale %.3e), max |Delta - host| %.3e (scale %.3e), max |delta - host| %.3e, max |G padding| %.3e\n", wG, sG, wD, sD, wd, wpad);
  return (wG < 1e-9 * fmax(sG, 1.0) && wD < 1e-12 * fmax(sD, 1.0) && wpad == 0.0) ? 0 : 2;
}
