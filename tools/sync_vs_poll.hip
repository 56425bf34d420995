// Microbenchmark: how long after a kernel launch does the host learn that the kernel is done —
// (a) hipStreamSynchronize, (b) polling a word the kernel's last instruction writes to pinned host memory.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k(volatile unsigned* host_word, unsigned seq, int spin) {
  unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)spin) {}
  if (threadIdx.x == 0) {
    __threadfence_system();
    __hip_atomic_store((unsigned*)host_word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
int main() {
  unsigned* w; hipHostMalloc((void**)&w, 64, hipHostMallocDefault); *w = 0;
  unsigned* wd; (void)hipHostGetDevicePointer((void**)&wd, w, 0);
  hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  for (int spin : {0, 1000}) {   // kernel body: ~0 and ~10 us (100 MHz clock)
    double a = 0, b = 0; const int reps = 2000;
    unsigned seq = 0;
    for (int mode = 0; mode < 2; mode++) {
      for (int r = 0; r < reps + 100; r++) {
        seq++;
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, st, wd, seq, spin);
        if (mode == 0) (void)hipStreamSynchronize(st);
        else while (__atomic_load_n(w, __ATOMIC_ACQUIRE) != seq) __builtin_ia32_pause();
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (r >= 100) (mode == 0 ? a : b) += us;
      }
      (void)hipStreamSynchronize(st);
    }
    printf("kernel body ~%d us: launch + hipStreamSynchronize %.2f us, launch + poll of a host word %.2f us\n", spin / 100, a / reps, b / reps);
  }
  return 0;
}
