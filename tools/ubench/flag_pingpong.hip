// micro-benchmark: round trip of a flag (and of a flag + payload) between two CO-RESIDENT workgroups through global memory on gfx950
// -- the price of splitting one sequential filter step over two workgroups (VERDICT r02 item 6 ii).  Developer tool.
//   hipcc --offload-arch=gfx950 -O3 -o flag_pingpong flag_pingpong.hip && ./flag_pingpong
// Workgroup A: [write payload] -> release-store flag = i ; spin (acquire) until ack == i.   Workgroup B: spin until flag == i ->
// [read payload, write reply] -> release-store ack = i.   One round trip = two one-way hand-offs.  Every spin loop is bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define SPIN_MAX 4000000

template <int SCOPE>   // __HIP_MEMORY_SCOPE_AGENT or __HIP_MEMORY_SCOPE_WORKGROUP (the latter is NOT sufficient across CUs: shown for the gap only)
__global__ void pingpong(unsigned* flags, double* payload, long long* out, int* xcc, int iters, int blk_a, int blk_b, int n_payload) {
  const int b = blockIdx.x;
  if (threadIdx.x == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[b] = (int)(id & 0xf);
  }
  if (b != blk_a && b != blk_b) return;
  unsigned* flag = flags; unsigned* ack = flags + 64;      // separate cache lines
  double* pa = payload; double* pb = payload + 1024;
  const int t = threadIdx.x;
  double acc = 0.0;
  __syncthreads();
  const long long w0 = wall_clock64();
  int bad = 0;
  for (int i = 1; i <= iters; ++i) {
    if (b == blk_a) {
      if (t < n_payload) pa[t] = (double)i + t;
      if (n_payload) { __threadfence(); __syncthreads(); }
      if (t == 0) {
        __hip_atomic_store(flag, (unsigned)i, __ATOMIC_RELEASE, SCOPE);
        int s = 0;
        while (__hip_atomic_load(ack, __ATOMIC_ACQUIRE, SCOPE) != (unsigned)i && ++s < SPIN_MAX) {}
        if (s >= SPIN_MAX) bad = 1;
      }
      __syncthreads();
      if (t < n_payload) acc += __hip_atomic_load(&pb[t], __ATOMIC_RELAXED, SCOPE);
    } else {
      if (t == 0) {
        int s = 0;
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, SCOPE) != (unsigned)i && ++s < SPIN_MAX) {}
        if (s >= SPIN_MAX) bad = 1;
      }
      __syncthreads();
      if (t < n_payload) { const double v = __hip_atomic_load(&pa[t], __ATOMIC_RELAXED, SCOPE); pb[t] = v * 2.0; acc += v; }
      if (n_payload) { __threadfence(); __syncthreads(); }
      if (t == 0) __hip_atomic_store(ack, (unsigned)i, __ATOMIC_RELEASE, SCOPE);
    }
    if (__syncthreads_or(bad)) break;
  }
  const long long w1 = wall_clock64();
  if (t == 0) { out[2 * (b == blk_a ? 0 : 1)] = w1 - w0; out[2 * (b == blk_a ? 0 : 1) + 1] = bad; }
  if (acc == 12345.678) out[7] = 1;
}

int main() {
  unsigned* flags; double* payload; long long* out; int* xcc;
  hipMalloc(&flags, 1024); hipMalloc(&payload, 2048 * 8); hipMalloc(&out, 64); hipMalloc(&xcc, 64 * 4);
  const int iters = 20000, nblk = 32;
  std::vector<int> hx(nblk);
  // where do the blocks of a small grid land?
  hipMemset(flags, 0, 1024); hipMemset(out, 0, 64);
  pingpong<__HIP_MEMORY_SCOPE_AGENT><<<nblk, 64>>>(flags, payload, out, xcc, 10, 0, 1, 0);
  hipDeviceSynchronize();
  hipMemcpy(hx.data(), xcc, nblk * 4, hipMemcpyDeviceToHost);
  printf("XCC id of blocks 0..%d of a %d-block grid:", nblk - 1, nblk);
  for (int i = 0; i < nblk; ++i) printf(" %d", hx[i]);
  printf("\n");
  int same = -1, other = -1;
  for (int i = 1; i < nblk; ++i) { if (same < 0 && hx[i] == hx[0]) same = i; if (other < 0 && hx[i] != hx[0]) other = i; }
  printf("partner of block 0 on the same XCD: block %d; on another XCD: block %d   (wall_clock64 = 100 MHz)\n", same, other);
  for (int pass = 0; pass < 2; ++pass) {
    const int partner = pass == 0 ? same : other;
    if (partner < 0) continue;
    for (int npay : {0, 38, 64, 256}) {
      for (int threads : {64, 256}) {
        if (npay > threads) continue;
        hipMemset(flags, 0, 1024); hipMemset(out, 0, 64);
        pingpong<__HIP_MEMORY_SCOPE_AGENT><<<nblk, threads>>>(flags, payload, out, xcc, iters, 0, partner, npay);
        hipDeviceSynchronize();
        long long ho[8]; hipMemcpy(ho, out, 64, hipMemcpyDeviceToHost);
        printf("%-12s agent scope, %3d-thread workgroups, payload %3d doubles each way: %.3f us per round trip (%.3f us one way)%s\n",
               pass == 0 ? "same XCD:" : "other XCD:", threads, npay, ho[0] * 10.0 / iters / 1e3, ho[0] * 10.0 / iters / 2e3, (ho[1] || ho[3]) ? "  [spin bound hit]" : "");
      }
    }
  }
  return 0;
}
