// nagp_grad.hip -- EKF energy WITH its gradient recursion:
// matlab/gf_giekf_modulator_nmf_constraints.m:332-480 with GradObj = 'on' (the same statements in gf_giekf_modulator_nmf.m:296-439).
//
// One workgroup per (parameter j, problem): it runs the plain EKF pass (prediction at every step incl. the first, one update per
// step, no isnan guard -- :385-472) and, beside it, the sensitivity recursion of ITS parameter:
//   prediction  dm_j <- dA_j m + A dm_j ;  dP_j <- dA_j P A' + A dP_j A' + (dA_j P A')' + dQ_j          (:387-401)
//   update      dS = dmdJH P JH' + JH dP_j JH' + JH P dmdJH' + dR_j ;  g_j += ... ;  dK, dm_j, dP_j        (:437-466)
// with dA_j = the lower-left block of expm([F 0; dF_j F]) and dQ_j = dPinf_j - dA_j Pinf A' - A dPinf_j A' - (dA_j Pinf A')'
// formed once on the host (:355-366, :392-394 -- time-invariant).  A, dA_j, Q, dQ_j, Pinf are block diagonal with the blocks of
// the model (the SDE is block diagonal and so is every derivative of it), so P and dP_j live as 4x4 register tiles, one or
// several per thread, and the prediction is tile-local; JH, dmdJH are supported on the first state of every block (H is a scaled
// selection), so the update needs two LDS panels -- those columns of P and of dP_j -- and a handful of vectors.
// Which terms a parameter takes is data (three flags per parameter), so that the host can ask for the reference's statements as
// written (kernel parameters: dmdJH = dm' d2h; the last D*N slices: dmdJH = dh(.; W_) with the kernel parameter's dm, dP --
// :438-444) or for the consistent gradient of the energy (see nagp/api.py: giekf_nlml_grad).
#include "nagp_dev.hpp"
#include "../../include/nagp.h"

#include <cmath>
#include <cstdio>
#include <vector>

extern "C" __attribute__((visibility("hidden"))) void nagp_internal_set_error(const char* msg);   // nagp_api.hip: the text nagp_last_error() returns on this thread

namespace nagp {

struct GradPar {
  int S, M, D, N, n_param;
  long long T;
  const int* off;          // [M+1]
  const double* mdl;       // [B][ A tiles M*16 | Q tiles M*16 | Pinf tiles M*16 | h M | W D*N (row-major d, j) | sn2 ]
  const double* par;       // [B][n_param][ dA tiles M*16 | dQ tiles M*16 | dPinf tiles M*16 ]
  const double* dR;        // [n_param]
  const int* hess;         // [n_param] dmdJH takes dm' * d2h
  const int* widx;         // [n_param] >= 0: + dh(.; W_) with W_ = unit matrix at (widx % D, widx / D)
  const int* wdir;         // [n_param] dmu takes h(.; W_) as well
  const double* y;         // [B][T]
  double* edata;           // [B]
  double* gdata;           // [B][n_param]
  int* status;             // [B] 1: innovation variance not positive even with the jitter (the reference returns NaN, :423-426)
};
__host__ __device__ inline size_t gmdl_size(int M, int D, int N) { return (size_t)M * 49 + (size_t)D * N + 1; }

template <int TPT>
__global__ void __launch_bounds__(256) ekf_grad_kernel(GradPar gp) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x, NT = 256;
  const int j = blockIdx.x, pb = blockIdx.y;
  const int M = gp.M, D = gp.D, N = gp.N, SP = 4 * M;           // padded state: 4 rows per block
  const double* mdl = gp.mdl + (size_t)pb * gmdl_size(M, D, N);
  const double* par = gp.par + ((size_t)pb * gp.n_param + j) * (size_t)M * 48;
  double* sA = lds;                      // [M][16]
  double* sdA = sA + M * 16;
  double* sQ = sdA + M * 16;
  double* sdQ = sQ + M * 16;
  double* hv = sdQ + M * 16;             // [M]
  double* sW = hv + M;                   // [D*N]
  double* m = sW + D * N;                // [SP] (padding rows stay 0)
  double* dm = m + SP;
  double* PJ = dm + SP;                  // P JH'
  double* Pd = PJ + SP;                  // P dmdJH'
  double* dPJ = Pd + SP;                 // dP JH'
  double* Kv = dPJ + SP;
  double* dKv = Kv + SP;
  double* JHc = dKv + SP;                // [M] JH at the first state of block n
  double* dJc = JHc + M;                 // [M] dmdJH there
  double* sc = dJc + M;                  // [8]: mu, dmu, Sx, dS
  double* colP = sc + 8;                 // [M][SP]  column (first state of block n) of P
  double* colD = colP + (size_t)M * SP;  // [M][SP]  ... of dP_j
  int* bsz = reinterpret_cast<int*>(colD + (size_t)M * SP);   // [M]

  for (int i = tid; i < M * 16; i += NT) { sA[i] = mdl[i]; sQ[i] = mdl[M * 16 + i]; sdA[i] = par[i]; sdQ[i] = par[M * 16 + i]; }
  for (int i = tid; i < M; i += NT) { hv[i] = mdl[M * 48 + i]; bsz[i] = gp.off[i + 1] - gp.off[i]; }
  for (int i = tid; i < D * N; i += NT) sW[i] = mdl[M * 49 + i];
  for (int i = tid; i < SP; i += NT) { m[i] = 0.0; dm[i] = 0.0; }
  const double R = mdl[M * 49 + D * N];
  const double dRj = gp.dR[j];
  const int hess = gp.hess[j], widx = gp.widx[j], wdir = gp.wdir[j];
  const int wd = widx >= 0 ? widx % D : 0, wj = widx >= 0 ? widx / D : 0;      // column-major (d, j) of W_

  // tiles (I, J) of P and dP_j, all M x M of them (the recursion keeps both symmetric; no use is made of it)
  int tI[TPT], tJ[TPT]; bool ok[TPT];
  double P[TPT][16], dP[TPT][16];
#pragma unroll
  for (int q = 0; q < TPT; ++q) {
    const int t = tid + q * NT;
    ok[q] = t < M * M;
    tI[q] = ok[q] ? t / M : 0; tJ[q] = ok[q] ? t - tI[q] * M : 0;
    tile_zero(P[q]); tile_zero(dP[q]);
    if (ok[q] && tI[q] == tJ[q]) { tile_load(P[q], mdl + M * 32 + (size_t)tI[q] * 16); tile_load(dP[q], par + M * 32 + (size_t)tI[q] * 16); }
  }
  __syncthreads();
  const double* yv = gp.y + (size_t)pb * gp.T;
  const int blk = tid >> 2, row = tid & 3;
  double e_acc = 0.0, g_acc = 0.0; bool bad = false;

  for (long long k = 0; k < gp.T; ++k) {
    // ---- prediction: dm <- dA m + A dm ; m <- A m  (old m on the right-hand sides), tiles
    double mn = 0.0, dmn = 0.0;
    if (tid < SP) {
      const double* a = sA + blk * 16 + 4 * row; const double* da = sdA + blk * 16 + 4 * row;
#pragma unroll
      for (int l = 0; l < 4; ++l) { mn = fma(a[l], m[4 * blk + l], mn); dmn = fma(da[l], m[4 * blk + l], dmn); dmn = fma(a[l], dm[4 * blk + l], dmn); }
    }
#pragma unroll
    for (int q = 0; q < TPT; ++q)
      if (ok[q]) {
        const double* aI = sA + tI[q] * 16; const double* aJ = sA + tJ[q] * 16;
        const double* dI = sdA + tI[q] * 16; const double* dJ = sdA + tJ[q] * 16;
        double x[16], z[16], w[16];
        tile_zero(x); tile_mma(x, aI, P[q]);            // A_I P
        tile_zero(z); tile_mma(z, dI, P[q]);            // dA_I P
        tile_zero(w); tile_mma(w, aI, dP[q]);           // A_I dP
#pragma unroll
        for (int e = 0; e < 16; ++e) w[e] += z[e];      // (dA_I P + A_I dP) A_J'
        tile_zero(dP[q]); tile_mma_nt(dP[q], w, aJ);
        tile_mma_nt(dP[q], x, dJ);                      // + A_I P dA_J'
        tile_zero(P[q]); tile_mma_nt(P[q], x, aJ);      // A_I P A_J'
        if (tI[q] == tJ[q]) {
#pragma unroll
          for (int e = 0; e < 16; ++e) { P[q][e] += sQ[tI[q] * 16 + e]; dP[q][e] += sdQ[tI[q] * 16 + e]; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { colP[(size_t)tJ[q] * SP + 4 * tI[q] + i] = P[q][4 * i]; colD[(size_t)tJ[q] * SP + 4 * tI[q] + i] = dP[q][4 * i]; }
      }
    __syncthreads();
    if (tid < SP) { m[tid] = mn; dm[tid] = dmn; }
    __syncthreads();
    // ---- measurement model at the predicted mean: h, dh, (dm' d2h), dh(.; W_)  (funh / funhd / funhd2, :492-514)
    if (tid < M) {
      const int n = tid;
      double part, dj = 0.0;
      if (n < D) {
        part = 0.0;
        for (int c = 0; c < N; ++c) {
          const double g = hv[D + c] * m[4 * (D + c)], eg = exp(g);
          part = fma(sW[n * N + c], log(1.0 + eg), part);
          if (hess) dj = fma(hv[D + c] * dm[4 * (D + c)], sW[n * N + c] * (eg / (eg + 1.0)), dj);       // sum_c dg_c W(n,c) dlink(g_c)
        }
        if (widx >= 0 && n == wd) { const double g = hv[D + wj] * m[4 * (D + wj)]; dj += log(1.0 + exp(g)); }
      } else {
        const int c = n - D;
        const double g = hv[n] * m[4 * n], eg = exp(g), dl = eg / (eg + 1.0);
        double zw = 0.0, dzw = 0.0;
        for (int d = 0; d < D; ++d) { zw = fma(hv[d] * m[4 * d], sW[d * N + c], zw); dzw = fma(hv[d] * dm[4 * d], sW[d * N + c], dzw); }
        part = zw * dl;
        if (hess) dj = dzw * dl + hv[n] * dm[4 * n] * zw * (dl * (1.0 - dl));
        if (widx >= 0 && c == wj) dj += hv[wd] * m[4 * wd] * dl;
      }
      JHc[n] = part * hv[n];
      dJc[n] = dj * hv[n];
    }
    __syncthreads();
    // ---- P JH', P dmdJH', dP JH' (states), then the scalars (every thread, same order)
    if (tid < SP) {
      double a = 0.0, b = 0.0, c = 0.0;
      for (int n = 0; n < M; ++n) {
        const double cp = colP[(size_t)n * SP + tid];
        a = fma(JHc[n], cp, a); b = fma(dJc[n], cp, b); c = fma(JHc[n], colD[(size_t)n * SP + tid], c);
      }
      PJ[tid] = a; Pd[tid] = b; dPJ[tid] = c;
    }
    __syncthreads();
    double mu = 0.0, dmu = 0.0, Sx = R, dS = dRj;
    for (int n = 0; n < D; ++n) mu = fma(hv[n] * m[4 * n], JHc[n] / hv[n], mu);         // sum_d z_d partials_d
    for (int n = 0; n < M; ++n) {
      dmu = fma(JHc[n], dm[4 * n], dmu);
      Sx = fma(JHc[n], PJ[4 * n], Sx);
      dS = fma(2.0 * dJc[n], PJ[4 * n], dS);
      dS = fma(JHc[n], dPJ[4 * n], dS);
    }
    if (wdir && widx >= 0) { const double g = hv[D + wj] * m[4 * (D + wj)]; dmu += hv[wd] * m[4 * wd] * log(1.0 + exp(g)); }
    if (!(Sx > 0.0)) { Sx += 0.5e-4; if (!(Sx > 0.0)) bad = true; }              // jitter 1e-4 * rand, rand -> 0.5 (C-7)
    const double v = yv[k] - mu, vtiS = v / Sx;
    g_acc += 0.5 * dS / Sx - 0.5 * dmu * vtiS - 0.5 * vtiS * dS * vtiS - 0.5 * vtiS * dmu;
    e_acc += 0.5 * log(2.0 * M_PI) + log(sqrt(Sx)) + 0.5 * vtiS * v;
    __syncthreads();                     // PJ, dm read above by everybody before they change
    if (tid < SP) {
      const double Kk = PJ[tid] / Sx;
      const double dK = dPJ[tid] / Sx + Pd[tid] / Sx - PJ[tid] / Sx * dS / Sx;       // dP HtiS + P dmdJH'/S - P HtiS dS/S
      Kv[tid] = Kk; dKv[tid] = dK;
      dm[tid] = dm[tid] + dK * v - Kk * dmu;
      m[tid] = m[tid] + Kk * v;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < TPT; ++q)
      if (ok[q]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const double ki = Kv[4 * tI[q] + i], kj = Kv[4 * tJ[q] + c], di = dKv[4 * tI[q] + i], dj = dKv[4 * tJ[q] + c];
            dP[q][4 * i + c] = dP[q][4 * i + c] - (di * kj) * Sx - (ki * kj) * dS - (ki * dj) * Sx;
            P[q][4 * i + c] -= (ki * kj) * Sx;
          }
      }
    // (the next step's prediction reads m, dm written above: the barrier in front of its LDS writes orders them)
    __syncthreads();
  }
  if (tid == 0) {
    gp.gdata[(size_t)pb * gp.n_param + j] = bad ? NAN : g_acc;
    if (j == 0) gp.edata[pb] = bad ? NAN : e_acc;
    if (bad) gp.status[pb] = 1;
  }
}

__host__ inline size_t grad_lds_bytes(int M, int D, int N) {
  const size_t SP = 4 * (size_t)M;
  return ((size_t)M * 16 * 4 + M + (size_t)D * N + 7 * SP + 2 * M + 8 + 2 * (size_t)M * SP) * sizeof(double) + (size_t)M * sizeof(int) + 16;
}

}  // namespace nagp

using namespace nagp;

#define GFAIL(code, ...) do { char _b[512]; snprintf(_b, sizeof _b, __VA_ARGS__); nagp_internal_set_error(_b); return (code); } while (0)
#define GHIP(x) do { hipError_t _e = (x); if (_e != hipSuccess) { for (void* v : allocs) (void)hipFree(v); (void)hipGetLastError(); GFAIL(_e == hipErrorOutOfMemory ? NAGP_ENOMEM : NAGP_EHIP, "nagp_giekf_nlml_grad: %s -> %s", #x, hipGetErrorString(_e)); } } while (0)

extern "C" int nagp_giekf_nlml_grad(int32_t B, const nagp_model* models, const double* const* ys, int64_t T, int32_t n_param,
                                    const double* const* dA, const double* const* dQ, const double* const* dPinf, const double* dR,
                                    const int32_t* hess, const int32_t* w_index, const int32_t* w_direct, double* edata, double* gdata,
                                    int32_t device) {
  std::vector<void*> allocs;
  if (B < 1 || !models || !ys || T < 1 || n_param < 1 || !dA || !dQ || !dPinf || !dR || !hess || !w_index || !w_direct || !edata || !gdata)
    GFAIL(NAGP_EINVAL, "nagp_giekf_nlml_grad: null/empty argument");
  const nagp_model& m0 = models[0];
  if (!m0.block_offsets || m0.M < 1 || m0.M != m0.D + m0.N || m0.D < 1 || m0.N < 1) GFAIL(NAGP_EINVAL, "nagp_giekf_nlml_grad: the EKF needs M = D + N");
  const int M = m0.M, D = m0.D, N = m0.N, S = m0.S;
  if (M * M > 1024) GFAIL(NAGP_EUNSUPPORTED, "nagp_giekf_nlml_grad: M = %d sites (more than 1024 covariance tiles)", M);
  for (int n = 0; n < M; ++n) {
    const int bs = m0.block_offsets[n + 1] - m0.block_offsets[n];
    if (bs < 1 || bs > 4) GFAIL(NAGP_EUNSUPPORTED, "nagp_giekf_nlml_grad: block %d has size %d (supported: 1..4)", n, bs);
  }
  if (m0.block_offsets[0] != 0 || m0.block_offsets[M] != S) GFAIL(NAGP_EINVAL, "nagp_giekf_nlml_grad: block_offsets do not span 0..S");
  for (int j = 0; j < n_param; ++j)
    if (w_index[j] >= D * N) GFAIL(NAGP_EINVAL, "nagp_giekf_nlml_grad: w_index[%d] outside Wnmf", j);
  for (int q = 0; q < B; ++q) {
    const nagp_model& mq = models[q];
    if (!mq.A || !mq.Q || !mq.Pinf || !mq.h_val || !mq.Wnmf || !mq.block_offsets || !ys[q] || !dA[q] || !dQ[q] || !dPinf[q])
      GFAIL(NAGP_EINVAL, "nagp_giekf_nlml_grad: problem %d: NULL pointer", q);
    bool same = mq.S == S && mq.M == M && mq.D == D && mq.N == N;
    for (int n = 0; same && n <= M; ++n) same = mq.block_offsets[n] == m0.block_offsets[n];
    if (!same) GFAIL(NAGP_EINVAL, "nagp_giekf_nlml_grad: problem %d has a different shape", q);
  }
  // pack: diagonal blocks as 4x4 row-major tiles, zero padded
  const size_t msz = gmdl_size(M, D, N), psz = (size_t)M * 48;
  std::vector<double> hm((size_t)B * msz, 0.0), hp((size_t)B * n_param * psz, 0.0);
  auto tiles = [&](double* dst, const double* dense) {      // dense S x S column-major -> [M][16]
    for (int n = 0; n < M; ++n) {
      const int o = m0.block_offsets[n], bs = m0.block_offsets[n + 1] - o;
      for (int i = 0; i < bs; ++i)
        for (int c = 0; c < bs; ++c) dst[(size_t)n * 16 + 4 * i + c] = dense[(size_t)(o + i) + (size_t)S * (o + c)];
    }
  };
  for (int q = 0; q < B; ++q) {
    double* d = hm.data() + (size_t)q * msz;
    tiles(d, models[q].A); tiles(d + (size_t)M * 16, models[q].Q); tiles(d + (size_t)M * 32, models[q].Pinf);
    for (int n = 0; n < M; ++n) d[(size_t)M * 48 + n] = models[q].h_val[n];
    for (int dd = 0; dd < D; ++dd)
      for (int c = 0; c < N; ++c) d[(size_t)M * 49 + (size_t)dd * N + c] = models[q].Wnmf[dd + (size_t)D * c];
    d[(size_t)M * 49 + (size_t)D * N] = std::exp(models[q].lik_param);
    for (int j = 0; j < n_param; ++j) {
      double* pj = hp.data() + ((size_t)q * n_param + j) * psz;
      const size_t SS = (size_t)S * S;
      tiles(pj, dA[q] + (size_t)j * SS); tiles(pj + (size_t)M * 16, dQ[q] + (size_t)j * SS); tiles(pj + (size_t)M * 32, dPinf[q] + (size_t)j * SS);
    }
  }
  // the device is looked at only after every pure-host step (validation AND packing: those run under ASan on GPU-less machines)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) GFAIL(NAGP_ENODEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) GFAIL(NAGP_EINVAL, "device ordinal %d out of range", device);
  GHIP(hipSetDevice(device));
  auto dmal = [&](void** p, size_t bytes) -> hipError_t { hipError_t e = hipMalloc(p, bytes ? bytes : 8); if (e == hipSuccess) allocs.push_back(*p); return e; };
  double *d_m = nullptr, *d_p = nullptr, *d_dR = nullptr, *d_y = nullptr, *d_e = nullptr, *d_g = nullptr; int *d_i = nullptr, *d_st = nullptr, *d_off = nullptr;
  GHIP(dmal((void**)&d_m, hm.size() * 8)); GHIP(dmal((void**)&d_p, hp.size() * 8)); GHIP(dmal((void**)&d_dR, (size_t)n_param * 8));
  GHIP(dmal((void**)&d_y, (size_t)B * T * 8)); GHIP(dmal((void**)&d_e, (size_t)B * 8)); GHIP(dmal((void**)&d_g, (size_t)B * n_param * 8));
  GHIP(dmal((void**)&d_i, (size_t)3 * n_param * 4)); GHIP(dmal((void**)&d_st, (size_t)B * 4)); GHIP(dmal((void**)&d_off, (size_t)(M + 1) * 4));
  GHIP(hipMemcpy(d_m, hm.data(), hm.size() * 8, hipMemcpyHostToDevice));
  GHIP(hipMemcpy(d_p, hp.data(), hp.size() * 8, hipMemcpyHostToDevice));
  GHIP(hipMemcpy(d_dR, dR, (size_t)n_param * 8, hipMemcpyHostToDevice));
  for (int q = 0; q < B; ++q) GHIP(hipMemcpy(d_y + (size_t)q * T, ys[q], (size_t)T * 8, hipMemcpyHostToDevice));
  GHIP(hipMemcpy(d_i, hess, (size_t)n_param * 4, hipMemcpyHostToDevice));
  GHIP(hipMemcpy(d_i + n_param, w_index, (size_t)n_param * 4, hipMemcpyHostToDevice));
  GHIP(hipMemcpy(d_i + 2 * n_param, w_direct, (size_t)n_param * 4, hipMemcpyHostToDevice));
  GHIP(hipMemcpy(d_off, m0.block_offsets, (size_t)(M + 1) * 4, hipMemcpyHostToDevice));
  GHIP(hipMemset(d_st, 0, (size_t)B * 4));
  GradPar gp{S, M, D, N, n_param, (long long)T, d_off, d_m, d_p, d_dR, d_i, d_i + n_param, d_i + 2 * n_param, d_y, d_e, d_g, d_st};
  const size_t lds = grad_lds_bytes(M, D, N);
  if (lds > 160 * 1024) { for (void* v : allocs) (void)hipFree(v); GFAIL(NAGP_EUNSUPPORTED, "nagp_giekf_nlml_grad: %zu B of LDS (> 160 KiB)", lds); }
  const int tpt = (M * M + 255) / 256;
  dim3 grid(n_param, B), bl(256);
#define LG(TP) do { if (lds > 48 * 1024) GHIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ekf_grad_kernel<TP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
                    hipLaunchKernelGGL((ekf_grad_kernel<TP>), grid, bl, lds, 0, gp); } while (0)
  if (tpt <= 1) LG(1); else if (tpt == 2) LG(2); else LG(4);
#undef LG
  GHIP(hipGetLastError());
  GHIP(hipDeviceSynchronize());
  GHIP(hipMemcpy(edata, d_e, (size_t)B * 8, hipMemcpyDeviceToHost));
  GHIP(hipMemcpy(gdata, d_g, (size_t)B * n_param * 8, hipMemcpyDeviceToHost));
  for (void* v : allocs) (void)hipFree(v);
  return NAGP_OK;
}
