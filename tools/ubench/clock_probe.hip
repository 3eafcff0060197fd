// micro-benchmark: what one busy CU gets.  Times N dependent-free f64 MFMAs and f64 FMAs on a single workgroup (and on a full
// grid) with the event clock, s_memtime (clock64) and the constant 100 MHz counter (wall_clock64).  Developer tool.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(double* out, long long* t, int iters, int mode) {
  v4d a0 = {0,0,0,0}, a1 = a0;
  double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-6;
  double f[8]; for (int j = 0; j < 8; ++j) f[j] = threadIdx.x + j;
  __syncthreads();
  const long long c0 = clock64(), w0 = wall_clock64();
  if (mode == 0) {
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
    }
  } else if (mode == 1) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = fma(f[j], 1.0000001, 1e-9);
    }
  } else if (mode == 2) {     // one f64 multiply feeding each MFMA (the multiply of step s+1 is independent of the MFMA of step s)
    for (int i = 0; i < iters; ++i) {
      f[0] = f[0] * 1.0000001; a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(f[0], y, a0, 0, 0, 0);
      f[1] = f[1] * 1.0000001; a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(f[1], y, a1, 0, 0, 0);
    }
  } else if (mode == 3) {     // an f64 multiply that does NOT feed the MFMAs, between them
    for (int i = 0; i < iters; ++i) {
      f[0] = f[0] * 1.0000001; a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
      f[1] = f[1] * 1.0000001; a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
    }
  } else {                    // a 32-bit multiply feeding nothing, between the MFMAs
    float g0 = x, g1 = y;
    for (int i = 0; i < iters; ++i) {
      g0 = g0 * 1.0000001f; a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
      g1 = g1 * 1.0000001f; a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
    }
    f[2] = g0 + g1;
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  double s = a0[0] + a1[1]; for (int j = 0; j < 8; ++j) s += f[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
int main() {
  double* d; hipMalloc(&d, 256 * 1024 * 8);
  long long* t; hipMallocManaged(&t, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 5; ++mode)
    for (int grid : {1, 256})
      for (int rep = 0; rep < 1; ++rep) {
        const int iters = 200000;
        hipEventRecord(e0); k<<<grid, 256>>>(d, t, iters, mode); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double nop = (double)iters * (mode == 1 ? 8 : 2);
        printf("%s grid %3d: %.2f ns per op per wave (events) | s_memtime ticks/op %.1f -> %.2f GHz | 100MHz ticks -> %.2f ms (events %.2f ms)\n",
               mode == 0 ? "mfma_f64_16x16x4" : mode == 1 ? "v_fma_f64       " : mode == 2 ? "mul->mfma       " : mode == 3 ? "mul64 | mfma    " : "mul32 | mfma    ", grid, ms * 1e6 / nop, t[0] / nop, t[0] / (ms * 1e6), t[1] / 1e5, ms);
      }
  return 0;
}
