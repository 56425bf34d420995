// Microbenchmark: sustained v_fmac_f32 issue rate with an SGPR operand, as a function of waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int CH>
__global__ void __launch_bounds__(64) k(float* out, const float* s, int iters) {
  float a[CH];
#pragma unroll
  for (int j = 0; j < CH; j++) a[j] = threadIdx.x * 0.001f + j;
  const float* sp = (const float*)__builtin_assume_aligned(s, 64);
  for (int it = 0; it < iters; it++) {
    float s0 = __builtin_nontemporal_load(sp + (it & 15));
    s0 = __builtin_amdgcn_readfirstlane(s0);
#pragma unroll
    for (int r = 0; r < 16; r++) {
#pragma unroll
      for (int j = 0; j < CH; j++) a[j] = __builtin_fmaf(s0, a[j], 0.5f);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float t = 0;
#pragma unroll
  for (int j = 0; j < CH; j++) t += a[j];
  out[blockIdx.x * 64 + threadIdx.x] = t;
}
int main() {
  float *out, *s;
  hipMalloc(&out, 1 << 24); hipMalloc(&s, 4096); hipMemset(s, 0, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  for (int wps = 1; wps <= 8; wps++) {
    int blocks = 256 * 4 * wps;  // waves
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL((k<9>), dim3(blocks), dim3(64), 0, 0, out, s, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr = (double)blocks * iters * 16 * 9;
    printf("waves/SIMD %d: %.3f ms, %.2f GHz-cycles/instr/SIMD at 2.4GHz, %.1f TFLOP/s\n", wps, ms,
           ms * 1e-3 * 2.4e9 / (instr / 1024), instr * 128 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
