// ORACLE / CPU BASELINE (test and measurement infrastructure only -- never linked, loaded or called by the product path).
//
// Compiled CPU restatement of the reference's hot loops, "dense as written": every S x S product of the .m text is a dense
// product in the .m's association order (no block structure exploited, (K*H)*P evaluated as two dense products, H*A*m as
// (H*A)*m), plain loops, no BLAS / LAPACK, own code.  It is the second, independent restatement of the same algorithm (the
// first is the NumPy oracle of this directory): tests/test_cpu_restatement.py holds the two against each other on the committed
// golden vectors, and bench.py times this one as `cpu_baseline` (kind "port": MATLAB / Octave do not exist here).
// PARITY UNPINNED like the rest of oracle/: the reference ships no fixture for this path.
//
// Follows (file:line under /root/reference/matlab):
//   gf_ep_modulator_nmf.m:113-283 (predict-mode EP sweeps; gf_ep_modulator.m:119-291 with predict_at_k1 and W = I)
//   ihgp_ep_modulator_nmf.m:223-454, :484-524 (look-up tables come in as arrays: the DARE set-up is not the timed loop)
//   gf_giekf_modulator_nmf.m:126-221 with iekf_update1.m:110-117 and the Jacobian of gf_giekf_modulator_nmf_constraints.m:492-502
//   likModulatorPower.m:25-100, likModulatorNMFPower.m:28-87, experiments/likModulatorPreCalcwn.m:28-86
// Arrays are row-major (NumPy C order); M x T outputs are [M][T].
//
// `structured` = 0: dense as written (the cost the MATLAB text has).  `structured` = 1: the same recursion with the structure a
// careful CPU implementation would use -- block-diagonal A (products over the diagonal blocks only), H as a scaled selection
// (W = P*H' is a column gather, H*P*H' an entry gather), the symmetric form P - K*W' of the gain update, no dense S x S work in the
// infinite-horizon loop at all -- the STRONGER baseline; results equal the dense form to rounding (tests/test_cpu_restatement.py).
//
//   g++ -O3 -march=native -fopenmp -shared -fPIC nagp_cpu.cpp -o libnagp_cpu.so      (oracle/cpu/__init__.py does it)
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#if defined(_OPENMP)
#include <omp.h>
#endif

namespace {

typedef std::vector<double> vec;
const double kNaN = std::numeric_limits<double>::quiet_NaN();
const double kInf = std::numeric_limits<double>::infinity();

// C (n x m) = A (n x k) * B (k x m)
void gemm_nn(double* __restrict C, const double* __restrict A, const double* __restrict B, int n, int k, int m) {
  for (int i = 0; i < n; ++i) {
    double* c = C + (size_t)i * m;
    for (int j = 0; j < m; ++j) c[j] = 0.0;
    for (int l = 0; l < k; ++l) {
      const double a = A[(size_t)i * k + l];
      const double* b = B + (size_t)l * m;
      for (int j = 0; j < m; ++j) c[j] += a * b[j];
    }
  }
}
// C (n x m) = A (n x k) * B' (B is m x k)
void gemm_nt(double* __restrict C, const double* __restrict A, const double* __restrict B, int n, int k, int m) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) {
      const double* a = A + (size_t)i * k;
      const double* b = B + (size_t)j * k;
      double s = 0.0;
#pragma omp simd reduction(+ : s)
      for (int l = 0; l < k; ++l) s += a[l] * b[l];
      C[(size_t)i * m + j] = s;
    }
}
void gemv(double* __restrict y, const double* __restrict A, const double* __restrict x, int n, int k) {
  for (int i = 0; i < n; ++i) {
    double s = 0.0;
#pragma omp simd reduction(+ : s)
    for (int l = 0; l < k; ++l) s += A[(size_t)i * k + l] * x[l];
    y[i] = s;
  }
}

// block-diagonal A (blocks ilist[n]..ilist[n+1]):  C = A * B  and  C = B * A'
void bd_left(double* __restrict C, const double* __restrict A, const double* __restrict B, const int32_t* il, int M, int S) {
  for (int n = 0; n < M; ++n) {
    const int o = il[n], b = il[n + 1] - o;
    for (int i = 0; i < b; ++i) {
      double* c = C + (size_t)(o + i) * S;
      for (int j = 0; j < S; ++j) c[j] = 0.0;
      for (int l = 0; l < b; ++l) {
        const double a = A[(size_t)(o + i) * S + o + l];
        const double* br = B + (size_t)(o + l) * S;
        for (int j = 0; j < S; ++j) c[j] += a * br[j];
      }
    }
  }
}
void bd_right_t(double* __restrict C, const double* __restrict B, const double* __restrict A, const int32_t* il, int M, int S) {
  for (int i = 0; i < S; ++i) {
    const double* br = B + (size_t)i * S; double* c = C + (size_t)i * S;
    for (int n = 0; n < M; ++n) {
      const int o = il[n], b = il[n + 1] - o;
      for (int j = 0; j < b; ++j) {
        double s_ = 0.0;
        for (int l = 0; l < b; ++l) s_ += br[o + l] * A[(size_t)(o + j) * S + o + l];
        c[o + j] = s_;
      }
    }
  }
}
void bd_gemv(double* __restrict y, const double* __restrict A, const double* __restrict x, const int32_t* il, int M, int S) {
  for (int n = 0; n < M; ++n) {
    const int o = il[n], b = il[n + 1] - o;
    for (int i = 0; i < b; ++i) { double s_ = 0.0; for (int l = 0; l < b; ++l) s_ += A[(size_t)(o + i) * S + o + l] * x[o + l]; y[o + i] = s_; }
  }
}

// lower Cholesky of the matrix whose LOWER triangle is X (chol(.,'lower')); false when a pivot is not positive
bool chol_lower(double* L, const double* X, int n) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) L[(size_t)i * n + j] = (j <= i) ? X[(size_t)i * n + j] : 0.0;
  for (int j = 0; j < n; ++j) {
    double d = L[(size_t)j * n + j];
#pragma omp simd reduction(- : d)
    for (int l = 0; l < j; ++l) d -= L[(size_t)j * n + l] * L[(size_t)j * n + l];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    L[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = L[(size_t)i * n + j];
#pragma omp simd reduction(- : s)
      for (int l = 0; l < j; ++l) s -= L[(size_t)i * n + l] * L[(size_t)j * n + l];
      L[(size_t)i * n + j] = s / d;
    }
  }
  return true;
}

struct Cub {
  int lik_kind, link_kind; double link_shift;
  int npts, dim; const double* wn; const double* xn;   // xn: [dim][npts]
};

inline double linkf(const Cub& c, double g) { return c.link_kind == 0 ? std::log(1.0 + std::exp(g - c.link_shift)) : std::exp(g); }

// [lZ, dlZ, d2lZ] = mom(hyp, mu, s2, Wnmf, ep_frac, yall, k)   (likModulator*Power.m)
double mom(const Cub& c, double sn2, double y, const double* mu, const double* s2, const double* W, int D, int N, double ep_fraction,
           double* dlZ, double* d2lZ, vec& ws) {
  const int n = c.npts;
  const bool power = c.lik_kind == 0, sq = c.lik_kind == 2;
  const double jitter = power ? 1e-8 : 1e-10;
  const double pEP = sq ? std::pow(2.0 * M_PI * sn2, 0.5 * (1.0 - ep_fraction)) / std::sqrt(ep_fraction) : 1.0;
  const double* mu_z = mu; const double* mu_g = mu + D; const double* s2_z = s2; const double* s2_g = s2 + D;
  ws.assign((size_t)n * (N + D) + 4 * (size_t)n, 0.0);
  double* xn = ws.data();                 // [n][N]
  double* a = xn + (size_t)n * N;         // [n][D]  link(xn) * W'
  double* wp = a + (size_t)n * D;         // wn .* normpdf
  double* c1 = wp + n;                    // (y - mu) / sn2_link
  double* c2 = c1 + n;                    // c1^2 - 1 / sn2_link
  vec lk(N);
  double ssum = 0.0;
  for (int p = 0; p < n; ++p) {
    for (int j = 0; j < N; ++j) {
      xn[(size_t)p * N + j] = mu_g[j] + std::sqrt(s2_g[j]) * c.xn[(size_t)j * n + p];
      lk[j] = linkf(c, xn[(size_t)p * N + j]);
    }
    double v = 0.0, lm = 0.0;
    for (int d = 0; d < D; ++d) {
      double ad;
      if (power) ad = lk[d];
      else { ad = 0.0; for (int j = 0; j < N; ++j) ad += lk[j] * W[(size_t)d * N + j]; }
      if (sq) ad = std::sqrt(ad);
      a[(size_t)p * D + d] = ad;
      v += ad * ad * s2_z[d];
      lm += ad * mu_z[d];
    }
    v += sn2 / ep_fraction;
    const double sd = std::sqrt(v);
    const double z = (y - lm) / sd;
    const double pdf = std::exp(-0.5 * z * z) / (std::sqrt(2.0 * M_PI) * sd);
    wp[p] = c.wn[p] * pdf;
    c1[p] = (y - lm) / v;
    c2[p] = c1[p] * c1[p] - 1.0 / v;
    ssum += wp[p];
  }
  const double Z = pEP * ((std::isnan(ssum) || ssum < jitter) ? jitter : ssum);   // MATLAB max(NaN, jitter) = jitter
  const double Zi = 1.0 / Z;
  for (int d = 0; d < D; ++d) {
    double s1 = 0.0, s2a = 0.0;
    for (int p = 0; p < n; ++p) { const double ad = a[(size_t)p * D + d]; s1 += wp[p] * c1[p] * ad; s2a += wp[p] * c2[p] * ad * ad; }
    dlZ[d] = Zi * pEP * s1;
    d2lZ[d] = -dlZ[d] * dlZ[d] + Zi * pEP * s2a;
  }
  for (int j = 0; j < N; ++j) {
    double s1 = 0.0, s2a = 0.0;
    for (int p = 0; p < n; ++p) { const double xg = (xn[(size_t)p * N + j] - mu_g[j]) / s2_g[j]; s1 += wp[p] * xg; s2a += wp[p] * (xg * xg - 1.0 / s2_g[j]); }
    dlZ[D + j] = Zi * pEP * s1;
    d2lZ[D + j] = -dlZ[D + j] * dlZ[D + j] + Zi * pEP * s2a;
  }
  return std::log(Z);
}

inline double max0(double v) { return v > 0.0 ? v : 0.0; }   // MATLAB max(v,0): NaN -> 0

struct Model { int S, M, D, N; const double *A, *Q, *H, *Pinf, *W; double lik_param; const int32_t* il; int structured; };

// One RTS step (gf_ep_modulator_nmf.m:210-230), dense as written.  Returns 0, 1 (retry taken) or -1 (both attempts failed).
int rts_step(const Model& md, const double* PSk, const double* MSk, double* m, double* P, vec& w) {
  const int S = md.S; const size_t SS = (size_t)S * S;
  w.resize(8 * SS + 2 * (size_t)S);
  double *t1 = w.data(), *PSkp = t1 + SS, *L = PSkp + SS, *B = L + SS, *X = B + SS, *G = X + SS, *t2 = G + SS, *Lt = t2 + SS, *v = Lt + SS, *v2 = v + S;
  if (md.structured) { bd_left(t1, md.A, PSk, md.il, md.M, S); bd_right_t(PSkp, t1, md.A, md.il, md.M, S); }
  else { gemm_nn(t1, md.A, PSk, S, S, S); gemm_nt(PSkp, t1, md.A, S, S, S); }
  for (size_t i = 0; i < SS; ++i) PSkp[i] += md.Q[i];
  int ret = 0;
  if (!chol_lower(L, PSkp, S)) {
    ret = 1;
    for (size_t i = 0; i < SS; ++i) t1[i] = PSkp[i];
    for (int i = 0; i < S; ++i) t1[(size_t)i * S + i] += std::sqrt(1e-4) * 0.5;   // sqrt(1e-4)*diag(rand): rand -> 0.5 (C-7)
    if (!chol_lower(L, t1, S)) return -1;
  }
  if (md.structured) bd_right_t(B, PSk, md.A, md.il, md.M, S);
  else gemm_nt(B, PSk, md.A, S, S, S);                             // PS_k * A'
  // X = B / L'  (X L' = B): row by row forward substitution; G = X / L (G L = X): backward
  for (int i = 0; i < S; ++i) for (int j = 0; j < S; ++j) Lt[(size_t)j * S + i] = L[(size_t)i * S + j];
  for (int i = 0; i < S; ++i) {
    double* x = X + (size_t)i * S; const double* b = B + (size_t)i * S;
    for (int j = 0; j < S; ++j) {
      double s = b[j];
#pragma omp simd reduction(- : s)
      for (int l = 0; l < j; ++l) s -= x[l] * L[(size_t)j * S + l];
      x[j] = s / L[(size_t)j * S + j];
    }
    double* g = G + (size_t)i * S;
    for (int j = S - 1; j >= 0; --j) {
      double s = x[j];
      const double* lt = Lt + (size_t)j * S;             // row j of L' = column j of L, contiguous
#pragma omp simd reduction(- : s)
      for (int l = j + 1; l < S; ++l) s -= g[l] * lt[l];
      g[j] = s / L[(size_t)j * S + j];
    }
  }
  if (md.structured) bd_gemv(v, md.A, MSk, md.il, md.M, S); else gemv(v, md.A, MSk, S, S);
  for (int i = 0; i < S; ++i) v[i] = m[i] - v[i];
  gemv(v2, G, v, S, S);
  for (int i = 0; i < S; ++i) m[i] = MSk[i] + v2[i];
  for (size_t i = 0; i < SS; ++i) t1[i] = P[i] - PSkp[i];
  gemm_nn(t2, G, t1, S, S, S);
  gemm_nt(t1, t2, G, S, S, S);
  for (size_t i = 0; i < SS; ++i) P[i] = PSk[i] + t1[i];
  return ret;
}

// H*P*H' (M x M) dense as written
void hph_full(double* out, const Model& md, const double* P, vec& w) {
  if (md.structured) {       // H = scaled selection: an entry gather
    for (int a = 0; a < md.M; ++a)
      for (int b = 0; b < md.M; ++b)
        out[(size_t)a * md.M + b] = md.H[(size_t)a * md.S + md.il[a]] * P[(size_t)md.il[a] * md.S + md.il[b]] * md.H[(size_t)b * md.S + md.il[b]];
    return;
  }
  w.resize((size_t)md.M * md.S);
  gemm_nn(w.data(), md.H, P, md.M, md.S, md.S);
  gemm_nt(out, w.data(), md.H, md.M, md.S, md.M);
}

}  // namespace

extern "C" {

int nagp_cpu_threads(void) {
#if defined(_OPENMP)
  return omp_get_max_threads();
#else
  return 1;
#endif
}

// the mom callback on its own (checked against oracle/lik.py)
double nagp_cpu_mom(int lik_kind, int link_kind, double link_shift, int npts, int dim, const double* wn, const double* xn, double lik_param,
                    double y, const double* mu, const double* s2, const double* W, int D, int N, double ep_fraction, double* dlZ, double* d2lZ) {
  Cub c{lik_kind, link_kind, link_shift, npts, dim, wn, xn};
  vec ws;
  return mom(c, std::exp(lik_param), y, mu, s2, W, D, lik_kind == 0 ? D : N, ep_fraction, dlZ, d2lZ, ws);
}

// gf_ep_modulator{,_nmf,_nmf_constraints} predict mode on an assembled model.  Outputs [M][T] / [T] / [I]; counters[2] = chol retries,
// not-PD steps.  Returns 0, or -6 when a Cholesky failed twice (where MATLAB's chol throws).
int nagp_cpu_gf_predict(int S, int M, int D, int N, const double* A, const double* Q, const double* H, const double* Pinf, const double* W,
                        double lik_param, int lik_kind, int link_kind, double link_shift, int npts, int dim, const double* wn, const double* xn,
                        const double* y, int64_t T, double ep_fraction, const double* ep_damping, int I, int predict_at_k1,
                        const int32_t* ilist, int structured,
                        double* Eft, double* Varft, double* nlZ, double* ttau, double* tnu, double* lZ, double* mdM, double* mdP, int64_t* counters) {
  Model md{S, M, D, N, A, Q, H, Pinf, W, lik_param, ilist, structured};
  Cub cub{lik_kind, link_kind, link_shift, npts, dim, wn, xn};
  const int Nc = (lik_kind == 0) ? D : N;
  const size_t SS = (size_t)S * S;
  const double sn2 = std::exp(lik_param);
  vec MS((size_t)T * S, 0.0), PS((size_t)T * SS, 0.0), m(S), P(SS), t1(SS), t2(SS), Wm((size_t)S * M), K((size_t)S * M), KH(SS);
  vec fmu(M), HPH(M), dl(M), d2(M), hpsp((size_t)T * M * M, 0.0), hmsp((size_t)T * M, 0.0), hp((size_t)M * M), hm(M), ws, wr, wh;
  for (int64_t i = 0; i < (int64_t)M * T; ++i) { ttau[i] = 0.0; tnu[i] = 0.0; }
  for (int64_t k = 0; k < T; ++k) lZ[k] = 0.0;
  counters[0] = counters[1] = 0;
  double ep_damp = ep_damping[0];
  for (int itt = 1; itt <= I; ++itt) {
    std::fill(m.begin(), m.end(), 0.0);
    std::memcpy(P.data(), Pinf, SS * sizeof(double));
    double maxDiffP = 0.0, maxDiffM = 0.0;
    for (int64_t k = 0; k < T; ++k) {
      if (k > 0 || predict_at_k1) {
        if (structured) {
          bd_gemv(t1.data(), A, m.data(), ilist, M, S); std::memcpy(m.data(), t1.data(), S * sizeof(double));
          bd_left(t1.data(), A, P.data(), ilist, M, S);
          bd_right_t(t2.data(), t1.data(), A, ilist, M, S);
        } else {
          gemv(t1.data(), A, m.data(), S, S); std::memcpy(m.data(), t1.data(), S * sizeof(double));
          gemm_nn(t1.data(), A, P.data(), S, S, S);
          gemm_nt(t2.data(), t1.data(), A, S, S, S);
        }
        for (size_t i = 0; i < SS; ++i) P[i] = t2[i] + Q[i];
      }
      if (!std::isnan(y[k])) {
        if (structured) {      // H = scaled selection
          for (int n = 0; n < M; ++n) {
            const int c = ilist[n]; const double h = H[(size_t)n * S + c];
            fmu[n] = h * m[c];
            for (int l = 0; l < S; ++l) Wm[(size_t)l * M + n] = P[(size_t)l * S + c] * h;
            HPH[n] = h * Wm[(size_t)c * M + n];
          }
        } else {
          gemv(fmu.data(), H, m.data(), M, S);
          gemm_nt(Wm.data(), P.data(), H, S, S, M);                          // W = P*H'  (S x M)
          for (int n = 0; n < M; ++n) { double s = 0.0; for (int l = 0; l < S; ++l) s += H[(size_t)n * S + l] * Wm[(size_t)l * M + n]; HPH[n] = s; }
        }
        double* tt = ttau; double* tn = tnu;
        if (itt == 1 || k == T - 1) {
          lZ[k] = mom(cub, sn2, y[k], fmu.data(), HPH.data(), W, D, Nc, 1.0, dl.data(), d2.data(), ws);
          for (int n = 0; n < M; ++n) {
            const double den = 1.0 + d2[n] * HPH[n];
            tt[(size_t)n * T + k] = max0((1 - ep_damp) * tt[(size_t)n * T + k] + ep_damp * (-d2[n] / den));
            tn[(size_t)n * T + k] = (1 - ep_damp) * tn[(size_t)n * T + k] + ep_damp * ((dl[n] - fmu[n] * d2[n]) / den);
          }
        }
        // sites with ttau == 0: K = W*(ttau/z), m -= W*(v/z), P -= K*W';  the others: K = W./(HPH+1/ttau)', m += K*v, P -= (K*H)*P
        bool any0 = false, anyn = false;
        for (int n = 0; n < M; ++n) { if (tt[(size_t)n * T + k] == 0.0) any0 = true; else anyn = true; }
        if (any0) {
          for (int n = 0; n < M; ++n) {
            if (tt[(size_t)n * T + k] != 0.0) continue;
            const double z = tt[(size_t)n * T + k] * HPH[n] + 1.0, v = tt[(size_t)n * T + k] * fmu[n] - tn[(size_t)n * T + k];
            for (int l = 0; l < S; ++l) m[l] -= Wm[(size_t)l * M + n] * (v / z);
            const double kz = tt[(size_t)n * T + k] / z;      // = 0: the covariance term vanishes, evaluated as written
            for (int i = 0; i < S; ++i)
              for (int j = 0; j < S; ++j) P[(size_t)i * S + j] -= Wm[(size_t)i * M + n] * kz * Wm[(size_t)j * M + n];
          }
        }
        if (anyn) {
          std::fill(K.begin(), K.end(), 0.0);
          for (int n = 0; n < M; ++n) {
            if (tt[(size_t)n * T + k] == 0.0) continue;
            const double den = HPH[n] + 1.0 / tt[(size_t)n * T + k], v = tn[(size_t)n * T + k] / tt[(size_t)n * T + k] - fmu[n];
            for (int l = 0; l < S; ++l) { K[(size_t)l * M + n] = Wm[(size_t)l * M + n] / den; m[l] += K[(size_t)l * M + n] * v; }
          }
          if (structured) {      // H*P = W' for the symmetric P the recursion keeps: P -= K*W'
            gemm_nt(t1.data(), K.data(), Wm.data(), S, M, S);
          } else {               // (K*H)*P with the rows of H that belong to the selected sites (the columns of K of the others are zero)
            gemm_nn(KH.data(), K.data(), H, S, M, S);
            gemm_nn(t1.data(), KH.data(), P.data(), S, S, S);
          }
          for (size_t i = 0; i < SS; ++i) P[i] -= t1[i];
        }
      }
      std::memcpy(&MS[(size_t)k * S], m.data(), S * sizeof(double));
      std::memcpy(&PS[(size_t)k * SS], P.data(), SS * sizeof(double));
    }
    if (itt == 1) { double s = 0.0; for (int64_t k = 0; k < T; ++k) s += lZ[k]; nlZ[0] = -s; }
    if (itt < I) ep_damp = ep_damping[itt];
    for (int64_t k = T - 2; k >= 0; --k) {
      const int r = rts_step(md, &PS[(size_t)k * SS], &MS[(size_t)k * S], m.data(), P.data(), wr);
      if (r > 0) ++counters[0];
      if (r < 0) { ++counters[0]; ++counters[1]; return -6; }
      std::memcpy(&MS[(size_t)k * S], m.data(), S * sizeof(double));
      std::memcpy(&PS[(size_t)k * SS], P.data(), SS * sizeof(double));
      gemv(hm.data(), H, m.data(), M, S);
      hph_full(hp.data(), md, P.data(), wh);
      if (itt < I && !std::isnan(y[k])) {
        vec mc(M), vc(M);
        for (int n = 0; n < M; ++n) {
          const double vm = hp[(size_t)n * M + n];
          vc[n] = 1.0 / (1.0 / vm - ep_fraction * ttau[(size_t)n * T + k]);
          mc[n] = vc[n] * (hm[n] / vm - ep_fraction * tnu[(size_t)n * T + k]);
        }
        lZ[k] = mom(cub, sn2, y[k], mc.data(), vc.data(), W, D, Nc, ep_fraction, dl.data(), d2.data(), ws);
        for (int n = 0; n < M; ++n) {
          if (!(vc[n] > 0.0)) continue;
          const double den = 1.0 + d2[n] * vc[n];
          ttau[(size_t)n * T + k] = (1 - ep_damp * ep_fraction) * ttau[(size_t)n * T + k] + ep_damp * (-d2[n] / den);
          tnu[(size_t)n * T + k] = (1 - ep_damp * ep_fraction) * tnu[(size_t)n * T + k] + ep_damp * ((dl[n] - mc[n] * d2[n]) / den);
        }
        for (int n = 0; n < M; ++n) ttau[(size_t)n * T + k] = max0(ttau[(size_t)n * T + k]);
      }
      // diagnostics against the previous sweep's smoothed values (H*MSP, H*PSP*H' kept from the end of that sweep)
      for (int n = 0; n < M; ++n) maxDiffM = std::fmax(maxDiffM, std::fabs(hmsp[(size_t)k * M + n] - hm[n]));
      for (int i = 0; i < M * M; ++i) maxDiffP = std::fmax(maxDiffP, std::fabs(hpsp[(size_t)k * M * M + i] - hp[i]));
      std::memcpy(&hmsp[(size_t)k * M], hm.data(), M * sizeof(double));
      std::memcpy(&hpsp[(size_t)k * M * M], hp.data(), (size_t)M * M * sizeof(double));
    }
    {   // the last step is never smoothed: its "previous sweep" value is the filtered one of this sweep
      gemv(hm.data(), H, &MS[(size_t)(T - 1) * S], M, S);
      hph_full(hp.data(), md, &PS[(size_t)(T - 1) * SS], wh);
      std::memcpy(&hmsp[(size_t)(T - 1) * M], hm.data(), M * sizeof(double));
      std::memcpy(&hpsp[(size_t)(T - 1) * M * M], hp.data(), (size_t)M * M * sizeof(double));
    }
    if (itt < I) { double s = 0.0; for (int64_t k = 0; k < T; ++k) s += lZ[k]; nlZ[itt] = -s; }
    mdM[itt - 1] = maxDiffM; mdP[itt - 1] = maxDiffP;
  }
  for (int64_t k = 0; k < T; ++k) {
    gemv(hm.data(), H, &MS[(size_t)k * S], M, S);
    hph_full(hp.data(), md, &PS[(size_t)k * SS], wh);
    for (int n = 0; n < M; ++n) { Eft[(size_t)n * T + k] = hm[n]; Varft[(size_t)n * T + k] = hp[(size_t)n * M + n]; }
  }
  return 0;
}

// ihgp_ep_modulator_nmf{,_constraints} predict mode.  Tables: r[NG]; channel n: PP rows of b*b (column-major b x b) at pp_off[n],
// PG rows of 2*b*b = [PS2(:)' G(:)'] at pg_off[n]; ilist[M+1] block starts.
int nagp_cpu_ihgp_predict(int S, int M, int D, int N, const double* A, const double* Q, const double* H, const double* Pinf, const double* W,
                          double lik_param, int lik_kind, int link_kind, double link_shift, int npts, int dim, const double* wn, const double* xn,
                          const int32_t* ilist, int NG, const double* r, const double* PP, const int64_t* pp_off, const double* PG, const int64_t* pg_off,
                          const double* y, int64_t T, double ep_fraction, const double* ep_damping, int I, int constraints_variant, int structured,
                          double* Eft, double* Varft, double* nlZ, double* ttau, double* tnu, double* Rout, double* mdM, double* mdP) {
  (void)Q;
  Cub cub{lik_kind, link_kind, link_shift, npts, dim, wn, xn};
  const int Nc = (lik_kind == 0) ? D : N;
  const size_t SS = (size_t)S * S;
  const double sn2 = std::exp(lik_param);
  auto nearest = [&](double Rv) {          // [~,ind] = min(abs(r-Rv)): first minimiser, all-NaN -> 1
    int best = 0; double bd = kNaN;
    for (int g = 0; g < NG; ++g) { const double d = std::fabs(r[g] - Rv); if (std::isnan(d)) continue; if (std::isnan(bd) || d < bd) { bd = d; best = g; } }
    return best;
  };
  vec MS((size_t)T * S, 0.0), m(S, 0.0), P(Pinf, Pinf + SS), PPm(SS), G(SS), HA((size_t)M * S), Wm((size_t)S * M), fmu(M), HPH(M), dl(M), d2(M), ys(M),
      hmsp((size_t)T * M, 0.0), hm(M), t1(S), t2(S), PSP(SS), hp((size_t)M * M), hp2((size_t)M * M), ws, wh;
  double* R = Rout;
  for (int64_t i = 0; i < (int64_t)M * T; ++i) { ttau[i] = 0.0; tnu[i] = 0.0; R[i] = constraints_variant ? 0.0 : sn2; }
  Model md{S, M, D, N, A, Q, H, Pinf, W, lik_param, ilist, structured};
  double ep_damp = ep_damping[0];
  for (int itt = 1; itt <= I; ++itt) {
    double lZs = 0.0, maxDiffM = 0.0;
    PSP = P;
    for (int64_t k = 0; k < T; ++k) {
      if (k > 0) {
        if (!structured) std::fill(PPm.begin(), PPm.end(), 0.0);      // PP = zeros(S) as written; the structured form reads the diagonal blocks only
        for (int n = 0; n < M; ++n) {
          const int o = ilist[n], b = ilist[n + 1] - o, ind = nearest(R[(size_t)n * T + k - 1]);
          const double* row = PP + pp_off[n] + (size_t)ind * b * b;
          for (int i = 0; i < b; ++i) for (int j = 0; j < b; ++j) PPm[(size_t)(o + i) * S + o + j] = row[i + b * j];
        }
      } else {
        std::memcpy(PPm.data(), Pinf, SS * sizeof(double));
      }
      if (structured) {        // per block: fmu_n = h_n (A_nn m_n)(1), W(ii,n) = PP_nn(:,1) h_n
        for (int n = 0; n < M; ++n) {
          const int o = ilist[n], b = ilist[n + 1] - o; const double h = H[(size_t)n * S + o];
          double s = 0.0; for (int j = 0; j < b; ++j) s += A[(size_t)o * S + o + j] * m[o + j];
          fmu[n] = h * s;
          for (int i = 0; i < b; ++i) Wm[(size_t)(o + i) * M + n] = PPm[(size_t)(o + i) * S + o] * h;
          HPH[n] = h * Wm[(size_t)o * M + n];
        }
      } else {
        gemm_nn(HA.data(), H, A, M, S, S);                       // fmu = (H*A)*m
        gemv(fmu.data(), HA.data(), m.data(), M, S);
        gemm_nt(Wm.data(), PPm.data(), H, S, S, M);              // W = PP*H'
        for (int n = 0; n < M; ++n) { double s = 0.0; for (int l = 0; l < S; ++l) s += H[(size_t)n * S + l] * Wm[(size_t)l * M + n]; HPH[n] = s; }
      }
      if (itt == 1 || k == T - 1) {
        lZs += mom(cub, sn2, y[k], fmu.data(), HPH.data(), W, D, Nc, 1.0, dl.data(), d2.data(), ws);
        for (int n = 0; n < M; ++n) {
          const double den = 1.0 + d2[n] * HPH[n];
          ttau[(size_t)n * T + k] = (1 - ep_damp) * ttau[(size_t)n * T + k] + ep_damp * (-d2[n] / den);
          tnu[(size_t)n * T + k] = (1 - ep_damp) * tnu[(size_t)n * T + k] + ep_damp * ((dl[n] - fmu[n] * d2[n]) / den);
          R[(size_t)n * T + k] = 1.0 / ttau[(size_t)n * T + k];                  // before the clamp (:269)
        }
      }
      for (int n = 0; n < M; ++n) { ttau[(size_t)n * T + k] = max0(ttau[(size_t)n * T + k]); ys[n] = tnu[(size_t)n * T + k] / ttau[(size_t)n * T + k]; }
      for (int n = 0; n < M; ++n) {
        const int o = ilist[n], b = ilist[n + 1] - o;
        double mi[8], Kb[8];
        if (ttau[(size_t)n * T + k] == 0.0) {
          R[(size_t)n * T + k] = kInf;
          for (int i = 0; i < b; ++i) { double s = 0.0; for (int j = 0; j < b; ++j) s += A[(size_t)(o + i) * S + o + j] * m[o + j]; mi[i] = s; }
          for (int i = 0; i < b; ++i) { m[o + i] = mi[i]; for (int j = 0; j < b; ++j) P[(size_t)(o + i) * S + o + j] = PPm[(size_t)(o + i) * S + o + j]; }
        } else {
          const double Rn = R[(size_t)n * T + k];
          for (int i = 0; i < b; ++i) Kb[i] = Wm[(size_t)(o + i) * M + n] / (HPH[n] + Rn);
          // AKHA = A_ii - (K*H_n,ii)*A_ii ; m_ii = AKHA*m_ii + K*ys
          double AK[64];
          for (int i = 0; i < b; ++i)
            for (int j = 0; j < b; ++j) {
              double s = 0.0;
              for (int l = 0; l < b; ++l) s += (Kb[i] * H[(size_t)n * S + o + l]) * A[(size_t)(o + l) * S + o + j];
              AK[i * b + j] = A[(size_t)(o + i) * S + o + j] - s;
            }
          for (int i = 0; i < b; ++i) { double s = 0.0; for (int j = 0; j < b; ++j) s += AK[i * b + j] * m[o + j]; mi[i] = s + Kb[i] * ys[n]; }
          for (int i = 0; i < b; ++i) { m[o + i] = mi[i]; for (int j = 0; j < b; ++j) P[(size_t)(o + i) * S + o + j] = PPm[(size_t)(o + i) * S + o + j] - Kb[i] * Kb[j] * Rn; }
        }
      }
      std::memcpy(&MS[(size_t)k * S], m.data(), S * sizeof(double));
    }
    if (itt == 1) nlZ[0] = -lZs;
    std::fill(P.begin(), P.end(), 0.0); std::fill(G.begin(), G.end(), 0.0);
    if (itt < I) ep_damp = ep_damping[itt];
    for (int64_t k = T - 2; k >= 0; --k) {
      for (int n = 0; n < M; ++n) {
        const int o = ilist[n], b = ilist[n + 1] - o;
        int ind = nearest(R[(size_t)n * T + k]);
        if (std::isinf(R[(size_t)n * T + k])) ind = NG - 1;
        const double* row = PG + pg_off[n] + (size_t)ind * 2 * b * b;
        for (int i = 0; i < b; ++i) for (int j = 0; j < b; ++j) { P[(size_t)(o + i) * S + o + j] = row[i + b * j]; G[(size_t)(o + i) * S + o + j] = row[b * b + i + b * j]; }
      }
      const double* MSk = &MS[(size_t)k * S];
      if (structured) {
        bd_gemv(t1.data(), A, MSk, ilist, M, S);
        for (int i = 0; i < S; ++i) t1[i] = m[i] - t1[i];
        bd_gemv(t2.data(), G.data(), t1.data(), ilist, M, S);
      } else {
        gemv(t1.data(), A, MSk, S, S);
        for (int i = 0; i < S; ++i) t1[i] = m[i] - t1[i];
        gemv(t2.data(), G.data(), t1.data(), S, S);
      }
      for (int i = 0; i < S; ++i) m[i] = MSk[i] + t2[i];
      std::memcpy(&MS[(size_t)k * S], m.data(), S * sizeof(double));
      gemv(hm.data(), H, m.data(), M, S);
      if (itt < I && !std::isnan(y[k])) {
        hph_full(hp.data(), md, P.data(), wh);
        vec mc(M), vc(M);
        for (int n = 0; n < M; ++n) {
          const double vm = hp[(size_t)n * M + n];
          vc[n] = 1.0 / (1.0 / vm - ep_fraction * ttau[(size_t)n * T + k]);
          mc[n] = vc[n] * (hm[n] / vm - ep_fraction * tnu[(size_t)n * T + k]);
        }
        const double lzk = mom(cub, sn2, y[k], mc.data(), vc.data(), W, D, Nc, ep_fraction, dl.data(), d2.data(), ws);
        if (itt > 1) lZs += lzk;
        for (int n = 0; n < M; ++n) {
          if (!(vc[n] > 0.0)) continue;
          const double den = 1.0 + d2[n] * vc[n];
          ttau[(size_t)n * T + k] = (1 - ep_damp * ep_fraction) * ttau[(size_t)n * T + k] + ep_damp * (-d2[n] / den);
          tnu[(size_t)n * T + k] = (1 - ep_damp * ep_fraction) * tnu[(size_t)n * T + k] + ep_damp * ((dl[n] - mc[n] * d2[n]) / den);
          R[(size_t)n * T + k] = 1.0 / ttau[(size_t)n * T + k];                    // no clamp here (:427-434)
        }
      }
      for (int n = 0; n < M; ++n) maxDiffM = std::fmax(maxDiffM, std::fabs(hmsp[(size_t)k * M + n] - hm[n]));
      std::memcpy(&hmsp[(size_t)k * M], hm.data(), M * sizeof(double));
    }
    gemv(hm.data(), H, &MS[(size_t)(T - 1) * S], M, S);
    std::memcpy(&hmsp[(size_t)(T - 1) * M], hm.data(), M * sizeof(double));
    hph_full(hp.data(), md, PSP.data(), wh); hph_full(hp2.data(), md, P.data(), wh);
    double maxDiffP = 0.0;
    for (int i = 0; i < M * M; ++i) maxDiffP = std::fmax(maxDiffP, std::fabs(hp[i] - hp2[i]));
    if (itt < I) nlZ[itt] = -lZs;
    mdM[itt - 1] = maxDiffM; mdP[itt - 1] = maxDiffP;
  }
  hph_full(hp.data(), md, P.data(), wh);
  for (int64_t k = 0; k < T; ++k) {
    gemv(hm.data(), H, &MS[(size_t)k * S], M, S);
    for (int n = 0; n < M; ++n) { Eft[(size_t)n * T + k] = hm[n]; Varft[(size_t)n * T + k] = constraints_variant ? hp[(size_t)n * M + n] : std::fabs(hp[(size_t)n * M + n]); }
  }
  return 0;
}

// gf_giekf_modulator_nmf{,_constraints} predict mode (Jacobian of the constraints file for both, as the oracle and the library do)
int nagp_cpu_giekf_predict(int S, int M, int D, int N, const double* A, const double* Q, const double* H, const double* Pinf, const double* W,
                           double lik_param, const double* y, int64_t T, int g_iter, int l_iter, int constraints_variant,
                           const int32_t* ilist, int structured,
                           double* Eft, double* Varft, double* mdP, int64_t* counters) {
  Model md{S, M, D, N, A, Q, H, Pinf, W, lik_param, ilist, structured};
  const size_t SS = (size_t)S * S;
  const double sigma2 = std::exp(lik_param);
  vec MS((size_t)T * S, 0.0), PS((size_t)T * SS, 0.0), m(S, 0.0), P(SS), t1(SS), t2(SS), hx(M), part(M), J(S), PJ(S), K(S), hpsp((size_t)T * M * M, 0.0), hp((size_t)M * M), hm(M), wr, wh;
  counters[0] = counters[1] = 0;
  for (int itt = 1; itt <= g_iter; ++itt) {
    if (itt == 1) { std::fill(m.begin(), m.end(), 0.0); std::memcpy(P.data(), Pinf, SS * sizeof(double)); }
    if (constraints_variant) std::memcpy(P.data(), Pinf, SS * sizeof(double));
    double maxDiffP = 0.0;
    for (int64_t k = 0; k < T; ++k) {
      if (k > 0) {
        if (structured) {
          bd_gemv(t1.data(), A, m.data(), ilist, M, S); std::memcpy(m.data(), t1.data(), S * sizeof(double));
          bd_left(t1.data(), A, P.data(), ilist, M, S);
          bd_right_t(t2.data(), t1.data(), A, ilist, M, S);
        } else {
          gemv(t1.data(), A, m.data(), S, S); std::memcpy(m.data(), t1.data(), S * sizeof(double));
          gemm_nn(t1.data(), A, P.data(), S, S, S);
          gemm_nt(t2.data(), t1.data(), A, S, S, S);
        }
        for (size_t i = 0; i < SS; ++i) P[i] = t2[i] + Q[i];
      }
      if (!std::isnan(y[k])) {
        double Sx = 0.0;
        for (int it = 0; it < l_iter; ++it) {
          gemv(hx.data(), H, m.data(), M, S);
          // h = (H_z x)' W softplus(H_g x);  partials = [W*link(g); ((H_z x)'*W)'.*dlink(g)];  J = partials'*H
          double MU = 0.0;
          for (int d = 0; d < D; ++d) { double s = 0.0; for (int j = 0; j < N; ++j) s += W[(size_t)d * N + j] * std::log(1.0 + std::exp(hx[D + j])); part[d] = s; MU += hx[d] * s; }
          for (int j = 0; j < N; ++j) { double s = 0.0; for (int d = 0; d < D; ++d) s += hx[d] * W[(size_t)d * N + j]; const double e = std::exp(hx[D + j]); part[D + j] = s * (e / (e + 1.0)); }
          for (int l = 0; l < S; ++l) { double s = 0.0; for (int n = 0; n < M; ++n) s += part[n] * H[(size_t)n * S + l]; J[l] = s; }
          gemv(PJ.data(), P.data(), J.data(), S, S);
          Sx = sigma2; for (int l = 0; l < S; ++l) Sx += J[l] * PJ[l];
          for (int l = 0; l < S; ++l) { K[l] = PJ[l] / Sx; m[l] += K[l] * (y[k] - MU); }
        }
        for (int i = 0; i < S; ++i) for (int j = 0; j < S; ++j) P[(size_t)i * S + j] -= K[i] * K[j] * Sx;
      }
      std::memcpy(&MS[(size_t)k * S], m.data(), S * sizeof(double));
      std::memcpy(&PS[(size_t)k * SS], P.data(), SS * sizeof(double));
    }
    for (int64_t k = T - 2; k >= 0; --k) {
      const int r = rts_step(md, &PS[(size_t)k * SS], &MS[(size_t)k * S], m.data(), P.data(), wr);
      if (r > 0) ++counters[0];
      if (r < 0) { ++counters[0]; ++counters[1]; return -6; }
      std::memcpy(&MS[(size_t)k * S], m.data(), S * sizeof(double));
      std::memcpy(&PS[(size_t)k * SS], P.data(), SS * sizeof(double));
      hph_full(hp.data(), md, P.data(), wh);
      for (int i = 0; i < M * M; ++i) maxDiffP = std::fmax(maxDiffP, std::fabs(hpsp[(size_t)k * M * M + i] - hp[i]));
      std::memcpy(&hpsp[(size_t)k * M * M], hp.data(), (size_t)M * M * sizeof(double));
    }
    hph_full(hp.data(), md, &PS[(size_t)(T - 1) * SS], wh);
    std::memcpy(&hpsp[(size_t)(T - 1) * M * M], hp.data(), (size_t)M * M * sizeof(double));
    mdP[itt - 1] = maxDiffP;
  }
  for (int64_t k = 0; k < T; ++k) {
    gemv(hm.data(), H, &MS[(size_t)k * S], M, S);
    hph_full(hp.data(), md, &PS[(size_t)k * SS], wh);
    for (int n = 0; n < M; ++n) { Eft[(size_t)n * T + k] = hm[n]; Varft[(size_t)n * T + k] = hp[(size_t)n * M + n]; }
  }
  return 0;
}

// `nseg` independent segments of one function family over the host cores (OpenMP, one segment per thread at a time): the path's own
// parallel axis.  kind 0 = gf, 1 = ihgp, 2 = giekf; every segment has the same model (the timing does not depend on the
// numbers) and its own observations ys[s*T ..]; outputs are discarded except nlZ[s*I ..] (giekf: maxDiffP).  Returns the worst status.
int nagp_cpu_segments(int kind, int nseg, int S, int M, int D, int N, const double* A, const double* Q, const double* H, const double* Pinf, const double* W,
                      double lik_param, int lik_kind, int link_kind, double link_shift, int npts, int dim, const double* wn, const double* xn,
                      const int32_t* ilist, int NG, const double* r, const double* PP, const int64_t* pp_off, const double* PG, const int64_t* pg_off,
                      const double* ys, int64_t T, double ep_fraction, const double* ep_damping, int I, int l_iter, int structured, int nthreads, double* nlZ) {
  int worst = 0;
  if (nthreads < 1) nthreads = nagp_cpu_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
  for (int s = 0; s < nseg; ++s) {
    vec Eft((size_t)M * T), Varft((size_t)M * T), tt((size_t)M * T), tn((size_t)M * T), lz(T), Rb((size_t)M * T), a(I), b(I);
    int64_t cnt[2];
    int st;
    if (kind == 0)
      st = nagp_cpu_gf_predict(S, M, D, N, A, Q, H, Pinf, W, lik_param, lik_kind, link_kind, link_shift, npts, dim, wn, xn, ys + (size_t)s * T, T, ep_fraction,
                               ep_damping, I, 0, ilist, structured, Eft.data(), Varft.data(), nlZ + (size_t)s * I, tt.data(), tn.data(), lz.data(), a.data(), b.data(), cnt);
    else if (kind == 1)
      st = nagp_cpu_ihgp_predict(S, M, D, N, A, Q, H, Pinf, W, lik_param, lik_kind, link_kind, link_shift, npts, dim, wn, xn, ilist, NG, r, PP, pp_off, PG, pg_off,
                                 ys + (size_t)s * T, T, ep_fraction, ep_damping, I, 0, structured, Eft.data(), Varft.data(), nlZ + (size_t)s * I, tt.data(), tn.data(), Rb.data(),
                                 a.data(), b.data());
    else
      st = nagp_cpu_giekf_predict(S, M, D, N, A, Q, H, Pinf, W, lik_param, ys + (size_t)s * T, T, I, l_iter, 0, ilist, structured, Eft.data(), Varft.data(), nlZ + (size_t)s * I, cnt);
#pragma omp critical
    if (st < worst) worst = st;
  }
  return worst;
}

}  // extern "C"
