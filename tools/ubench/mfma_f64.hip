// micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 and v_fma_f64 on gfx950 (developer tool)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k_mfma(double* out, int iters, int nacc) {
  v4d a0 = {0,0,0,0}, a1 = a0, a2 = a0, a3 = a0;
  double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-6;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
    if (nacc > 1) a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
    if (nacc > 2) { a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0); }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
__global__ void k_fma(double* out, int iters) {
  double a[8]; for (int j = 0; j < 8; ++j) a[j] = threadIdx.x + j;
  double x = 1.0000001, y = 1e-9;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = fma(a[j], x, y);
  }
  double s = 0; for (int j = 0; j < 8; ++j) s += a[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double* d; hipMalloc(&d, 256 * 1024 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves = 1; waves <= 2; ++waves)
    for (int nacc : {1, 2, 4}) {
      const int iters = 20000;
      k_mfma<<<256, 256 * waves>>>(d, 10, nacc); hipDeviceSynchronize();
      hipEventRecord(e0); k_mfma<<<256, 256 * waves>>>(d, iters, nacc); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double n_mfma = (double)iters * nacc;   // per wave
      printf("mfma_f64_16x16x4: %d wave(s)/SIMD, %d independent acc: %.1f ns per MFMA per wave  -> %.2f TFLOP/s chip\n", waves, nacc,
             ms * 1e6 / n_mfma, 2048.0 * n_mfma * 256 * 4 * waves / (ms * 1e-3) / 1e12);
    }
  for (int waves = 1; waves <= 2; ++waves) {
    const int iters = 20000;
    k_fma<<<256, 256 * waves>>>(d, 10); hipDeviceSynchronize();
    hipEventRecord(e0); k_fma<<<256, 256 * waves>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("v_fma_f64 (8 independent chains): %d wave(s)/SIMD: %.2f ns per wave-FMA -> %.2f TFLOP/s chip\n", waves, ms * 1e6 / (iters * 8.0),
           2.0 * 64 * iters * 8.0 * 256 * 4 * waves / (ms * 1e-3) / 1e12);
  }
  return 0;
}
