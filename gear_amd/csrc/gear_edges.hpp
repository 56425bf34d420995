// gear_edges.hpp — edge batches from their END POINTS (gfx950).
//
// An RRT "edge" is the straight sequence of samples that rrt 0.4.0's stepping produces between two joint
// configurations, and the reference evaluates its closure on each of them in turn
// (src/planner/joint_path_planner.rs:168-188).  The samples are a function of the two end points
// and the step length alone, so a large batch of edges need not cross PCIe sample by sample: the caller sends
// 2 x D values per edge, one lane per edge replays the recurrence on the device and writes the samples into the
// scene's workspace, and the ordinary batch path evaluates them.
//
// The recurrence is ITERATED, not closed-form: every sample is computed from the previous one,
//     d = sqrt(sum_i (cur_i - target_i)^2)          (accumulated in index order)
//     next_i = cur_i + (target_i - cur_i) * step / d
// so that the coordinates are bit-identical to what the host stepping (gear_host.hpp connect_line /
// smooth_path, i.e. rrt's Tree::extend and smooth_path) hands to the predicate.  Everything below is therefore
// compiled without contraction, and with the IEEE-correct division and square root hipcc uses by default.
//
// Two stepping rules (`mode`):
//   GH_EDGE_CONNECT  rrt::Tree::connect: repeat extend() from the last new vertex until a sample lies within
//                    `step` of the target (ExtendStatus::Reached); a step that would overshoot lands ON the target.
//   GH_EDGE_WALK     rrt::smooth_path's walker: while the walker is at least `step` away from the target, advance
//                    it by `step`; the target itself is never evaluated.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace gear_hip {

constexpr uint32_t kEdgeConnect = 0u, kEdgeWalk = 1u;
constexpr uint32_t kEdgeTruncated = 0x80000000u;  // n_samples[e] bit 31: the edge did not end within max_len samples
constexpr int kEdgeMaxDof = 64;                   // a lane keeps its walker in registers / scratch: D <= 64

#pragma clang fp contract(off)
template <class T> __device__ __forceinline__ T edge_sqrt(T x);
// the IEEE square root (llvm.sqrt without approximation flags; hipcc rounds f32 division and square root correctly by
// default).  NOT __fsqrt_rn: in this toolchain's headers that is the NATIVE square root unless
// OCML_BASIC_ROUNDED_OPERATIONS is defined, and whether it comes out refined depended on the instantiation.
template <> __device__ __forceinline__ float edge_sqrt<float>(float x) { return __builtin_sqrtf(x); }
template <> __device__ __forceinline__ double edge_sqrt<double>(double x) { return __builtin_sqrt(x); }

// One edge's recurrence.  emit(j, cur) is called with sample j's coordinates (cur[0..D)); returns the number of
// samples, with kEdgeTruncated OR-ed in when max_len ended the edge.
// DD > 0: the number of columns is DD, known to the compiler — walker and target live in registers and the loops
// unroll (same operations in the same order: same bits).  DD == 0: any D up to kEdgeMaxDof, the walker in scratch
// memory (a serial chain of ~120 steps per edge through scratch took 4 us a step: 1.07 ms of a 3.5 ms call for the
// 6-DOF scene's 65536 edges before the columns became a template argument).
template <class T, int DD, class Emit>
__device__ __forceinline__ uint32_t edge_walk(const T* __restrict__ from, const T* __restrict__ to_mem, int D_rt, T step,
                                              uint32_t max_len, uint32_t mode, Emit&& emit) {
  constexpr int N = DD > 0 ? DD : kEdgeMaxDof;
  const int D = DD > 0 ? DD : D_rt;
  T cur[N];
  T to_reg[DD > 0 ? DD : 1];
  for (int i = 0; i < D; i++) cur[i] = from[i];
  if (DD > 0)
    for (int i = 0; i < D; i++) to_reg[i] = to_mem[i];
  const T* to = DD > 0 ? to_reg : to_mem;
  uint32_t n = 0;
  for (;;) {
    T s = T(0);
    for (int i = 0; i < D; i++) { const T x = cur[i] - to[i]; s += x * x; }
    const T d = edge_sqrt<T>(s);
    const bool near = d < step;
    if (mode == kEdgeWalk && near) return n;        // the walker has arrived: nothing more to evaluate
    if (n >= max_len) return n | kEdgeTruncated;
    if (near) {
      for (int i = 0; i < D; i++) cur[i] = to[i];   // extend(): q_new = q_target
    } else {
      for (int i = 0; i < D; i++) cur[i] = cur[i] + (to[i] - cur[i]) * step / d;
    }
    emit(n, cur);
    n++;
    if (mode == kEdgeConnect) {
      if (near) return n;                           // distance(q_new, q_target) = 0 < step: Reached
      T s2 = T(0);
      for (int i = 0; i < D; i++) { const T x = cur[i] - to[i]; s2 += x * x; }
      if (edge_sqrt<T>(s2) < step) return n;        // Reached
    }
  }
}
#pragma clang fp contract(fast)

// pass 1: samples per edge
template <class T, int DD>
__global__ void __launch_bounds__(256)
k_edge_count(const T* __restrict__ from, const T* __restrict__ to, size_t n_edges, int D, T step, uint32_t max_len,
             uint32_t mode, uint32_t* __restrict__ cnt) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n_edges) return;
  cnt[e] = edge_walk<T, DD>(from + e * (size_t)D, to + e * (size_t)D, D, step, max_len, mode, [](uint32_t, const T*) {});
}

// exclusive prefix sum of cnt (bit 31 masked off) into off[0..n_edges], three small launches:
// per block of 1024 edges / the block totals (one workgroup, any count) / add
__global__ void __launch_bounds__(1024)
k_edge_scan1(const uint32_t* __restrict__ cnt, size_t n_edges, uint32_t* __restrict__ off, uint32_t* __restrict__ bsum) {
  __shared__ uint32_t sh[1024];
  const size_t e = (size_t)blockIdx.x * 1024 + threadIdx.x;
  const uint32_t v = e < n_edges ? (cnt[e] & ~kEdgeTruncated) : 0u;
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const uint32_t t = threadIdx.x >= (unsigned)d ? sh[threadIdx.x - d] : 0u;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  if (e < n_edges) off[e] = sh[threadIdx.x] - v;
  if (threadIdx.x == 1023) bsum[blockIdx.x] = sh[1023];
}
__global__ void __launch_bounds__(1024)
k_edge_scan2(uint32_t* __restrict__ bsum, size_t n_blocks) {
  __shared__ uint32_t sh[1024];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (size_t base = 0; base < n_blocks; base += 1024) {
    const size_t b = base + threadIdx.x;
    const uint32_t v = b < n_blocks ? bsum[b] : 0u;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const uint32_t t = threadIdx.x >= (unsigned)d ? sh[threadIdx.x - d] : 0u;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (b < n_blocks) bsum[b] = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) bsum[n_blocks] = carry;  // the total
}
__global__ void __launch_bounds__(1024)
k_edge_scan3(uint32_t* __restrict__ off, size_t n_edges, const uint32_t* __restrict__ bsum, size_t n_blocks) {
  const size_t e = (size_t)blockIdx.x * 1024 + threadIdx.x;
  if (e < n_edges) off[e] += bsum[blockIdx.x];
  if (e == 0) off[n_edges] = bsum[n_blocks];
}

// pass 2: the samples themselves, row-major at off[e]
template <class T, int DD>
__global__ void __launch_bounds__(256)
k_edge_fill(const T* __restrict__ from, const T* __restrict__ to, size_t n_edges, int D, T step, uint32_t max_len,
            uint32_t mode, const uint32_t* __restrict__ off, T* __restrict__ q) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n_edges) return;
  const int Dn = DD > 0 ? DD : D;  // (a bound the compiler knows keeps the walker in registers)
  T* out = q + (size_t)off[e] * (size_t)Dn;
  edge_walk<T, DD>(from + e * (size_t)Dn, to + e * (size_t)Dn, D, step, max_len, mode, [&](uint32_t j, const T* cur) {
    T* row = out + (size_t)j * (size_t)Dn;
    for (int i = 0; i < Dn; i++) row[i] = cur[i];
  });
}

// f(std::integral_constant<int, D>) for D = 1..8, f(integral_constant<int, 0>) (any D, at run time) beyond
template <class F> inline void by_columns(int D, F&& f) {
  switch (D) {
    case 1: f(std::integral_constant<int, 1>{}); break;
    case 2: f(std::integral_constant<int, 2>{}); break;
    case 3: f(std::integral_constant<int, 3>{}); break;
    case 4: f(std::integral_constant<int, 4>{}); break;
    case 5: f(std::integral_constant<int, 5>{}); break;
    case 6: f(std::integral_constant<int, 6>{}); break;
    case 7: f(std::integral_constant<int, 7>{}); break;
    case 8: f(std::integral_constant<int, 8>{}); break;
    default: f(std::integral_constant<int, 0>{}); break;
  }
}

}  // namespace gear_hip
