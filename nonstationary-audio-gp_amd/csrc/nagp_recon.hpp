// nagp_recon.hpp -- what the drivers do with the outputs of the hot path (matlab/demo_toy_modulators_nmf.m:119-158):
// the reconstructed signal  sig = sum_d (W link(g))_d z_d  and the modulator amplitudes link(g_n) under the INDEPENDENT posterior
// marginals z_d ~ N(Eft_d, Varft_d), g_n ~ N(Eft_{D+n}, Varft_{D+n}) of one time step:
//   Eft_mod = mean link(g_n), Varft_mod = var link(g_n), Esig = mean sig, Vsig = var sig.
// Embarrassingly parallel over t.  Two forms:
//   moments   the population values of those means / variances: one-dimensional Gauss-Hermite quadrature of link and link^2 per
//             modulator (exp link: closed form), then
//               Esig = sum_d a_d m_d,  a = W mu_lk
//               Vsig = sum_d a_d^2 v_d + sum_n var_lk,n [ (sum_d W_dn m_d)^2 + sum_d W_dn^2 v_d ]
//   sampling  the reference's own estimator (s = 250 draws per marginal, :123, sample variance with s-1), draws from a
//             counter-based generator (Philox4x32-10, key = seed, counter = (t, sample block, site)) + Box-Muller, so that a host
//             restatement reproduces them.  One wave per time step, four samples per lane and trip.
#pragma once
#include "nagp_dev.hpp"

namespace nagp {

struct ReconPar {
  int D, N, M;
  int64_t T;
  int link_kind; double link_shift;
  const double* W;       // [D][N] row-major
  const double* Eft;     // [T][M]
  const double* Varft;   // [T][M]
  int n_gh; const double* gh_x; const double* gh_w;   // moments form (standard-normal weight)
  int n_samp; unsigned long long seed;                 // sampling form
  double* Esig; double* Vsig;                          // [T]
  double* Emod; double* Vmod;                          // [T][N]
};

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned* out) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// four standard normals from one counter (Box-Muller on 32-bit uniforms, (x + 0.5) / 2^32)
__device__ __forceinline__ void normal4(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, double* z) {
  unsigned u[4];
  philox4x32_10(c0, c1, c2, c3, k0, k1, u);
  const double s = 2.3283064365386963e-10;   // 2^-32
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const double u1 = ((double)u[2 * h] + 0.5) * s, u2 = ((double)u[2 * h + 1] + 0.5) * s;
    const double r = sqrt(-2.0 * log(u1)), a = 6.283185307179586 * u2;
    z[2 * h] = r * cos(a); z[2 * h + 1] = r * sin(a);
  }
}

// moments form: one thread per time step
__global__ void __launch_bounds__(256) recon_moments_kernel(ReconPar rp) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* sW = lds;                       // [D][N]
  double* gx = sW + rp.D * rp.N;          // [n_gh]
  double* gw = gx + rp.n_gh;
  for (int i = threadIdx.x; i < rp.D * rp.N; i += blockDim.x) sW[i] = rp.W[i];
  for (int i = threadIdx.x; i < rp.n_gh; i += blockDim.x) { gx[i] = rp.gh_x[i]; gw[i] = rp.gh_w[i]; }
  __syncthreads();
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rp.T) return;
  const int D = rp.D, N = rp.N, M = rp.M;
  const double* m = rp.Eft + (size_t)t * M;
  const double* v = rp.Varft + (size_t)t * M;
  double vs = 0.0, es = 0.0;
  double mu[MOM_MAXCD], va[MOM_MAXCD];
  for (int n = 0; n < N; ++n) {
    const double mg = m[D + n], vg = v[D + n];
    double e1, e2;
    if (rp.link_kind == 1) {              // exp link: E exp(g) = exp(m + v/2), E exp(2g) = exp(2m + 2v)
      e1 = exp(mg + 0.5 * vg); e2 = exp(2.0 * mg + 2.0 * vg);
    } else {
      const double sg = sqrt(vg);
      e1 = 0.0; e2 = 0.0;
      for (int q = 0; q < rp.n_gh; ++q) {
        const double l = link_eval(0, rp.link_shift, mg + sg * gx[q]);
        e1 = fma(gw[q], l, e1); e2 = fma(gw[q] * l, l, e2);
      }
    }
    mu[n] = e1; va[n] = e2 - e1 * e1;
    rp.Emod[(size_t)t * N + n] = e1; rp.Vmod[(size_t)t * N + n] = va[n];
  }
  for (int d = 0; d < D; ++d) {
    double a = 0.0;
    for (int n = 0; n < N; ++n) a = fma(sW[d * N + n], mu[n], a);
    es = fma(a, m[d], es);
    vs = fma(a * a, v[d], vs);
  }
  for (int n = 0; n < N; ++n) {
    double wm = 0.0, wv = 0.0;
    for (int d = 0; d < D; ++d) { const double w = sW[d * N + n]; wm = fma(w, m[d], wm); wv = fma(w * w, v[d], wv); }
    vs = fma(va[n], fma(wm, wm, wv), vs);
  }
  rp.Esig[t] = es; rp.Vsig[t] = vs;
}

// sampling form: one wave per time step
__global__ void __launch_bounds__(64) recon_sample_kernel(ReconPar rp) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* sW = lds;
  for (int i = threadIdx.x; i < rp.D * rp.N; i += 64) sW[i] = rp.W[i];
  __syncthreads();
  const int lane = threadIdx.x, D = rp.D, N = rp.N, M = rp.M, S = rp.n_samp;
  const unsigned k0 = (unsigned)rp.seed, k1 = (unsigned)(rp.seed >> 32);
  for (int64_t t = blockIdx.x; t < rp.T; t += gridDim.x) {
    const double* m = rp.Eft + (size_t)t * M;
    const double* v = rp.Varft + (size_t)t * M;
    // sums of (x - c) and (x - c)^2 with the shift c = value at the marginal mean (keeps the variance formula well conditioned)
    double cl[MOM_MAXCD], s1[MOM_MAXCD], s2[MOM_MAXCD];
    for (int n = 0; n < N; ++n) { cl[n] = link_eval(rp.link_kind, rp.link_shift, m[D + n]); s1[n] = 0.0; s2[n] = 0.0; }
    double csig = 0.0;
    for (int d = 0; d < D; ++d) { double a = 0.0; for (int n = 0; n < N; ++n) a = fma(sW[d * N + n], cl[n], a); csig = fma(a, m[d], csig); }
    double g1 = 0.0, g2 = 0.0;
    for (int q0 = 0; q0 * 4 < S; q0 += 64) {
      const int q = q0 + lane;                       // sample block: samples 4q .. 4q+3
      double lk[MOM_MAXCD][4], sig[4] = {0, 0, 0, 0};
      for (int n = 0; n < N; ++n) {
        double z[4];
        normal4((unsigned)t, (unsigned)((unsigned long long)t >> 32), (unsigned)q, (unsigned)(D + n), k0, k1, z);
        const double sg = sqrt(v[D + n]);
#pragma unroll
        for (int e = 0; e < 4; ++e) lk[n][e] = link_eval(rp.link_kind, rp.link_shift, fma(sg, z[e], m[D + n]));
      }
      for (int d = 0; d < D; ++d) {
        double z[4];
        normal4((unsigned)t, (unsigned)((unsigned long long)t >> 32), (unsigned)q, (unsigned)d, k0, k1, z);
        const double sd = sqrt(v[d]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          double a = 0.0;
          for (int n = 0; n < N; ++n) a = fma(sW[d * N + n], lk[n][e], a);
          sig[e] = fma(a, fma(sd, z[e], m[d]), sig[e]);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (4 * q + e < S) {
          for (int n = 0; n < N; ++n) { const double x = lk[n][e] - cl[n]; s1[n] += x; s2[n] = fma(x, x, s2[n]); }
          const double x = sig[e] - csig; g1 += x; g2 = fma(x, x, g2);
        }
      }
    }
    g1 = wave_sum(g1); g2 = wave_sum(g2);
    for (int n = 0; n < N; ++n) { s1[n] = wave_sum(s1[n]); s2[n] = wave_sum(s2[n]); }
    if (lane == 0) {
      const double inv = 1.0 / S, inv1 = 1.0 / (S - 1);
      rp.Esig[t] = csig + g1 * inv; rp.Vsig[t] = (g2 - g1 * g1 * inv) * inv1;
      for (int n = 0; n < N; ++n) { rp.Emod[(size_t)t * N + n] = cl[n] + s1[n] * inv; rp.Vmod[(size_t)t * N + n] = (s2[n] - s1[n] * s1[n] * inv) * inv1; }
    }
  }
}

}  // namespace nagp
