// micro-benchmark + check of nagp_chol16.hpp: one wave factors and inverts 16x16 SPD tiles held in registers (row per lane).
// build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -I nonstationary-audio-gp_amd/csrc -o tools/ubench/chol16 tools/ubench/chol16.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "nagp_chol16.hpp"
using namespace nagp;
__global__ void __launch_bounds__(64) k(const double* tiles, double* inv, long long* t, int ntile, int reps) {
  const int lane = threadIdx.x & 63, row = lane & 15, g = lane >> 4;
  long long cyc = 0;
  for (int rep = 0; rep < reps; ++rep)
    for (int q = 0; q < ntile; ++q) {
      double a[16], x[4];
#pragma unroll
      for (int c = 0; c < 16; ++c) a[c] = tiles[(size_t)q * 256 + row * 16 + c];
      __builtin_amdgcn_s_waitcnt(0);
      const long long c0 = clock64();
      const bool ok = chol16_inv_rows(a, x, row, g);
      asm volatile("" :: "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"((int)ok));
      const long long c1 = clock64();
      cyc += c1 - c0;
#pragma unroll
      for (int r = 0; r < 4; ++r) inv[(size_t)q * 256 + row * 16 + 4 * r + g] = ok ? x[r] : -1.0;      // lane row g: columns 4r + g
    }
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = cyc;
}
int main() {
  const int nt = 64, reps = 20;
  std::vector<double> h((size_t)nt * 256), Lh((size_t)nt * 256), Xh((size_t)nt * 256);
  srand(3);
  for (int q = 0; q < nt; ++q) {
    double Bm[256];
    for (int i = 0; i < 256; ++i) Bm[i] = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double s = (i == j) ? 1e-3 * (q + 1) : 0.0;
        for (int l = 0; l < 16; ++l) s += Bm[i * 16 + l] * Bm[j * 16 + l];
        h[(size_t)q * 256 + i * 16 + j] = s;
      }
    // host: L then X = inv(L)
    double L[256] = {0}, X[256] = {0};
    for (int j = 0; j < 16; ++j) {
      double s = h[(size_t)q * 256 + j * 16 + j];
      for (int l = 0; l < j; ++l) s -= L[j * 16 + l] * L[j * 16 + l];
      L[j * 16 + j] = sqrt(s);
      for (int i = j + 1; i < 16; ++i) {
        double v = h[(size_t)q * 256 + i * 16 + j];
        for (int l = 0; l < j; ++l) v -= L[i * 16 + l] * L[j * 16 + l];
        L[i * 16 + j] = v / L[j * 16 + j];
      }
    }
    for (int c = 0; c < 16; ++c)
      for (int i = 0; i < 16; ++i) {
        double v = (i == c) ? 1.0 : 0.0;
        for (int l = 0; l < i; ++l) v -= L[i * 16 + l] * X[l * 16 + c];
        X[i * 16 + c] = v / L[i * 16 + i];
      }
    for (int i = 0; i < 256; ++i) Xh[(size_t)q * 256 + i] = X[i];
  }
  double *d, *dx; long long* t;
  hipMalloc(&d, h.size() * 8); hipMalloc(&dx, h.size() * 8); hipMallocManaged(&t, 16);
  hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {1, 256}) {
    k<<<grid, 64>>>(d, dx, t, nt, reps); hipDeviceSynchronize();
    hipEventRecord(e0); k<<<grid, 64>>>(d, dx, t, nt, reps); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<double> o(h.size());
    hipMemcpy(o.data(), dx, h.size() * 8, hipMemcpyDeviceToHost);
    double worst = 0, scale = 0;
    for (size_t i = 0; i < o.size(); ++i) { worst = fmax(worst, fabs(o[i] - Xh[i])); scale = fmax(scale, fabs(Xh[i])); }
    // (the in-kernel cycle counter brackets are not ordering points for the arithmetic between them: the wall time of the launch is the figure)
    printf("grid %3d: %.2f us per tile by the event clock, loads and stores included (one wave, %d tiles x %d) = ~%.0f cycles at 2.4 GHz | in-kernel bracket %.0f | max |inv - host| = %.3e (scale %.3e)\n",
           grid, ms * 1e3 / (nt * reps), nt, reps, ms * 1e3 / (nt * reps) * 2400.0, (double)t[0] / (nt * reps), worst, scale);
  }
  return 0;
}
