// gear_hip.hip — the C ABI of include/gear_hip.h over the gfx950 kernels.
// Built with: hipcc --offload-arch=gfx950 -O3 -shared -fPIC (see gear_amd/build.py).
// There is no CPU fallback in this library: without a HIP device every compute
// entry point fails with GH_ERR_NO_DEVICE / GH_ERR_HIP.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/gear_hip.h"
#include "gear_kernels.hpp"
#include "gear_small.hpp"
#include "gear_edges.hpp"
#include "gear_shard.hpp"
#include "gear_scene.hpp"

using namespace gear_hip;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

#define GH_HIP(expr)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail(e_ == hipErrorNoDevice ? GH_ERR_NO_DEVICE : GH_ERR_HIP,                    \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                        \
  } while (0)

// LDS budget per wavefront-workgroup for the per-lane pose columns; beyond it the
// columns move to a global workspace (chunked launches).
constexpr size_t kLdsBudget = 36 * 1024;
constexpr size_t kChunk = size_t(1) << 22;  // configurations per launch when a global store is in use

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  int ensure(size_t need) {
    if (need <= bytes) return GH_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    GH_HIP(hipMalloc(&p, need));
    bytes = need;
    return GH_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
};

}  // namespace

// Host memory that the device reads and writes in place (hipHostMalloc: pinned, fine-grained coherent, mapped).  The small
// synchronous calls of the host-buffer entry points (an RRT line, a start/goal check) stage their
// arguments here and launch on it directly: no copy commands, one stream synchronisation.
struct PinBuf {
  void* p = nullptr;   // host address
  void* d = nullptr;   // the same memory as the device addresses it
  size_t bytes = 0;
  int ensure(size_t need) {
    if (need <= bytes) return GH_OK;
    release();
    need = std::max<size_t>(need, 4096);
    // coherent + mapped, explicitly: the host-poll completion protocol (wait_notified) needs the kernel's system-scope
    // release store to be visible to the host in order, which a non-coherent pinned block does not promise
    GH_HIP(hipHostMalloc(&p, need, hipHostMallocCoherent | hipHostMallocMapped));
    if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess) { (void)hipHostFree(p); p = d = nullptr; return fail(GH_ERR_HIP, "hipHostGetDevicePointer failed"); }
    bytes = need;
    return GH_OK;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = d = nullptr;
    bytes = 0;
  }
};
constexpr size_t kZeroCopyMax = 4096;  // configurations (and edges) up to which a host-buffer call runs in place
// Host-buffer batches above that: up to kStagedMax configurations go through the scene's page-locked staging (a
// copy out of pageable memory costs ~0.5 ms before the first byte moves; a memcpy of a few MB is cheaper); larger
// ones are copied straight from the caller's memory in pieces of kHostPiece, piece i + 1 on its way while piece i
// is evaluated (tools/host_buffers.py).
constexpr size_t kStagedMax = size_t(1) << 18;
constexpr size_t kHostPiece = size_t(1) << 20;

constexpr int kMaxSets = 4;

// ---- gh_check_batch_noalloc: workspace that nobody allocates --------------------------------------
// One array per GPU, part of the code object: it exists from the moment the library's code is loaded on a
// device until the process ends.  `scratch` is the host-side stand-in of the symbol, NOT a device address:
// host code only ever uses the per-device address that hipGetSymbolAddress returned (scratch_on_device).
constexpr size_t kScratchBytes = size_t(64) << 20;
constexpr size_t kNoAllocMaxConfigs = size_t(1) << 16;
constexpr int kMaxDevices = 64;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

// Tuning and test hooks.  Read from the environment ONCE, when a scene is created; no entry point looks
// at the environment afterwards.
struct Knobs {
  size_t pipe_chunk = size_t(4) << 20;  // GEAR_HIP_PIPE_CHUNK: configurations per pipeline pass (at most; a call's chunks are equal)
  int pipe_streams = kMaxSets;          // GEAR_HIP_PIPE_STREAMS: 1 keeps every chunk on the caller's stream
  size_t exact_cap = 0;                 // GEAR_HIP_EXACT_CAP: exact entries per sweep wavefront, fixed (tests force re-checks with it)
  int fail_part = -1;                   // GEAR_HIP_TEST_FAIL_PART: this share of a device group's call reports a failure (tests)
  size_t exact_cap0 = 0;                // GEAR_HIP_EXACT_CAP0: ... to start with; grows as usual when regions run full (tests)
  bool zero_copy = true;                // GEAR_HIP_ZERO_COPY=0: host-buffer calls always copy (measuring aid)
  bool sweep_clds = true;               // GEAR_HIP_SWEEP_CLDS=0: cull table never in LDS
  int sweep_waves = 0;                  // GEAR_HIP_SWEEP_WAVES: wavefronts per sweep workgroup (0: chosen by LDS use)
  int sweep_q = -1;                     // GEAR_HIP_SWEEP_Q: where the sweep keeps its batch row (0 memory, 2 LDS column; -1: chosen)
  bool n2_mlds = true;                  // GEAR_HIP_N2_MLDS=0: mesh table never in LDS
  bool n2_hist = false;                 // GEAR_HIP_N2_HIST: histogram of GJK iterations (profiling)
  int n2_tile = 0;                      // GEAR_HIP_N2_TILE: sweep wavefronts per exact-stage workgroup (8, 16, 32; 0: by chunk size)
  bool small_fused = false;             // GEAR_HIP_SMALL_FUSED: small batches through the fused kernel (measuring aid)
  size_t small_max = 16384;             // GEAR_HIP_SMALL_MAX: batches up to this size take k_small (0: never); f64 scenes: 3/8 of it
                                        // (measured crossovers with the pipeline: 18-60 k configurations in f32 on four scenes, ~7 k in f64)
  unsigned small_qcap = 1024;           // GEAR_HIP_SMALL_QCAP: item queue entries of a k_small workgroup (tests shrink it)
  static Knobs from_env() {
    Knobs k;
    auto num = [](const char* name, long long dflt) { const char* e = getenv(name); return e ? strtoll(e, nullptr, 0) : dflt; };
    long long x = num("GEAR_HIP_PIPE_CHUNK", 0);
    if (x >= 4096 && x <= (1ll << 24)) k.pipe_chunk = ((size_t)x + kWave - 1) / kWave * kWave;
    x = num("GEAR_HIP_PIPE_STREAMS", kMaxSets);
    k.pipe_streams = (int)(x < 1 ? 1 : x > kMaxSets ? kMaxSets : x);
    k.exact_cap = (size_t)std::max<long long>(0, num("GEAR_HIP_EXACT_CAP", 0));
    k.exact_cap0 = (size_t)std::max<long long>(0, num("GEAR_HIP_EXACT_CAP0", 0));
    k.fail_part = (int)num("GEAR_HIP_TEST_FAIL_PART", -1);
    k.zero_copy = num("GEAR_HIP_ZERO_COPY", 1) != 0;
    k.sweep_clds = num("GEAR_HIP_SWEEP_CLDS", 1) != 0;
    k.sweep_waves = (int)num("GEAR_HIP_SWEEP_WAVES", 0);
    k.sweep_q = (int)num("GEAR_HIP_SWEEP_Q", -1);
    k.n2_mlds = num("GEAR_HIP_N2_MLDS", 1) != 0;
    k.n2_hist = getenv("GEAR_HIP_N2_HIST") != nullptr;
    k.n2_tile = (int)num("GEAR_HIP_N2_TILE", 0);
    k.small_fused = getenv("GEAR_HIP_SMALL_FUSED") != nullptr;
    k.small_max = (size_t)std::max<long long>(0, num("GEAR_HIP_SMALL_MAX", 16384));
    x = num("GEAR_HIP_SMALL_QCAP", 1024);
    k.small_qcap = (unsigned)(x < 1 ? 1 : x > 8192 ? 8192 : x);
    return k;
  }
};

namespace {
std::mutex g_scratch_mu;
void* g_scratch_ptr[kMaxDevices] = {};  // device k's copy of `scratch`, as device k addresses it
// Every scene on a device shares that device's array, and with stream == NULL each scene launches on its own
// stream: two scenes called back to back from one host thread would overlap on the device.  The last call's end is
// therefore recorded in an event per device (created with the address, i.e. when a scene is created: the entry point
// itself still allocates nothing); the next call's stream waits for it before its first launch.  Calls are thereby
// ordered in the order they were ENQUEUED; the enqueue runs under g_scratch_mu.
hipEvent_t g_scratch_ev[kMaxDevices] = {};
bool g_scratch_used[kMaxDevices] = {};

// Address of the CURRENT device's copy of the array (looked up once per device; the caller has made
// `device` current).
int scratch_on_device(int device, void** out) {
  if (device < 0 || device >= kMaxDevices) return fail(GH_ERR_INVALID, "device index beyond the no-allocation workspace table");
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (!g_scratch_ptr[device]) {
    void* p = nullptr;
    GH_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return fail(GH_ERR_HIP, "hipGetSymbolAddress returned no address for the workspace array");
    GH_HIP(hipEventCreateWithFlags(&g_scratch_ev[device], hipEventDisableTiming));
    g_scratch_ptr[device] = p;
  }
  *out = g_scratch_ptr[device];
  return GH_OK;
}

// A caller-less bump allocator over a fixed region (the no-allocation entry point's workspace)
struct Carve {
  char* base = nullptr;
  size_t bytes = 0, used = 0;
  void* take(size_t need) {  // 256-byte aligned; nullptr when the region is exhausted (used keeps counting)
    const size_t at = (used + 255) / 256 * 256;
    used = at + need;
    return used <= bytes ? base + at : nullptr;
  }
};
}  // namespace

// ---- the pipeline's workspace: ONE pool per device, shared by every scene on it ---------------------
// A planner holds a scene per (joint set, obstacle set); a workspace set is 0.69 GB for a 1M-configuration
// chunk of the metric scene (f32: 660 B per configuration), and up to four sets are in flight in one call.  Scenes therefore own only their tables and staging buffers; the sets live here, grow to the
// largest call seen on the device, and are freed when the device's last scene is destroyed.  Calls on
// different scenes and streams may overlap in time, so a set carries the event of its last use: whoever
// takes it next makes its stream wait for that event first (the enqueue itself runs under the pool's lock).
struct PipeBufs { DevBuf hit, xcnt, xkey, xpose_a, xpose_b, fq, pc; hipEvent_t last_use = nullptr; bool used = false; };
struct PipePool {
  std::mutex mu;
  PipeBufs sets[kMaxSets];
  int scenes = 0;
};
namespace {
PipePool g_pool[kMaxDevices];
}

struct gh_scene {
  int precision = 32, device = 0;
  int D = 0, N = 0, L = 0, O = 0, P = 0, n_slots = 0;
  int O_user = 0;  // obstacle shapes as the caller counts them (O counts a general mesh's triangles one by one)
  int max_shape_node = -1;  // last node that owns a robot shape
  double margin = 0;
  SceneDev<float> s32{};
  SceneDev<double> s64{};
  std::vector<void*> tables;  // device allocations owned by the scene
  hipStream_t stream = nullptr;
  std::mutex mu;
  DevBuf q, flags, out, edges, store, first;
  DevBuf ends, ecnt, ebsum;  // gh_check_edges_from_endpoints: end points, samples per edge, block totals of the scan
  PinBuf hq, hflags, hedges, hfirst;  // small host-buffer calls run on these in place
  // the pipeline's workspace sets are the device's (PipePool); consecutive chunks of one call take different
  // sets so that they can be in flight together (set 0 on the caller's stream, set k on aux[k])
  bool pooled = false;  // this scene is counted in its device's PipePool
  hipStream_t aux[kMaxSets] = {};  // aux[0] unused: set 0 runs on the caller's stream
  hipEvent_t ev_fork = nullptr, ev_join[kMaxSets] = {};
  hipEvent_t ev_h2d[2] = {}, ev_d2h[2] = {};  // large host-buffer calls: a piece's input is in / its flags are out (made on first use)
  int last_sets = 0;  // workspace sets the last pipeline call used
  unsigned exact_cap = 128;  // exact entries per sweep wavefront before the fused path takes the wavefront over
  unsigned exact_cap0 = 128; // ... as the scene was created with (the no-allocation entry point never grows it)
  PinBuf hover;              // a host word a sweep wavefront sets when its exact region ran full: the next call doubles exact_cap
  bool has_mesh = false;
  Knobs knobs;
  CountersDev* counters = nullptr;
  unsigned* small_done = nullptr;  // k_small's finished-workgroup ticket (behind `counters`; zero between calls)
  PinBuf hseq;                     // the word a small host-buffer call's kernel stores its sequence number in when it is done
  unsigned seq = 0;
  std::vector<gh_scene*> parts;  // gh_scene_create_multi: one scene per listed device; this object only routes
  size_t elem() const { return precision == 32 ? 4 : 8; }
};

namespace {

template <class U>
int upload(gh_scene* sc, const std::vector<U>& v, const U** out) {
  void* d = nullptr;
  size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(U);
  GH_HIP(hipMalloc(&d, bytes));
  sc->tables.push_back(d);
  if (!v.empty()) GH_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(U), hipMemcpyHostToDevice));
  *out = (const U*)d;
  return GH_OK;
}

template <class T>
int build_and_upload(gh_scene* sc, const gh_robot_desc& r, const gh_obstacle_desc& o, double margin, SceneDev<T>& S) {
  HostTables<T> H;
  BuildError err{GH_OK, ""};
  if (!build_tables<T>(r, o, margin, H, err)) return fail(err.code, err.msg);
  S.n_nodes = (int)H.nodes.size();
  S.n_cols = r.n_cols;
  S.n_shapes = (int)H.rshapes.size();
  S.n_obs = (int)H.oshapes.size();
  S.n_pairs = (int)H.pairs.size();
  S.n_limits = (int)H.limits.size();
  S.n_slots = H.n_slots;
  S.margin = H.margin;
  int rc;
  if ((rc = upload(sc, H.nodes, &S.nodes))) return rc;
  if ((rc = upload(sc, H.limits, &S.limits))) return rc;
  if ((rc = upload(sc, H.rshapes, &S.rshapes))) return rc;
  if ((rc = upload(sc, H.oshapes, &S.oshapes))) return rc;
  if ((rc = upload(sc, H.ocull, &S.ocull))) return rc;
  if ((rc = upload(sc, H.ocullq, &S.ocullq))) return rc;
  S.n_obs_pad = (int)(H.ocullq.size() / 5);
  if ((rc = upload(sc, H.pairs, &S.pairs))) return rc;
  if ((rc = upload(sc, H.spairs, &S.spairs))) return rc;
  if ((rc = upload(sc, H.spair_r, &S.spair_r))) return rc;
  S.n_spairs = (int)H.spair_r.size();
  S.n_first = H.n_first;
  if ((rc = upload(sc, H.sp_first_slot, &S.sp_first_slot))) return rc;
  if ((rc = upload(sc, H.sp_late_off, &S.sp_late_off))) return rc;
  if ((rc = upload(sc, H.sp_late, &S.sp_late))) return rc;
  if ((rc = upload(sc, H.sp_late_rec, &S.sp_late_rec))) return rc;
  if ((rc = upload(sc, H.hot, &S.hot))) return rc;
  if ((rc = upload(sc, H.verts, &S.verts))) return rc;
  if ((rc = upload(sc, H.planes, &S.planes))) return rc;
  {
    S.n_verts = (int)(H.verts.size() / 3);
    S.n_planes = (int)(H.planes.size() / 4);
    std::vector<T> m4((size_t)(S.n_verts + S.n_planes) * 4, T(0));
    for (int v = 0; v < S.n_verts; v++)
      for (int c = 0; c < 3; c++) m4[(size_t)4 * v + c] = H.verts[(size_t)3 * v + c];
    std::copy(H.planes.begin(), H.planes.end(), m4.begin() + (size_t)4 * S.n_verts);
    if ((rc = upload(sc, m4, &S.mesh4))) return rc;
  }
  if ((rc = upload(sc, H.obs_owner, &S.obs_owner))) return rc;
  S.n_obs_user = H.n_obs_user;
  sc->O_user = H.n_obs_user;
  if ((rc = upload(sc, H.grids, &S.grids))) return rc;
  if ((rc = upload(sc, H.shape_grid, &S.shape_grid))) return rc;
  if ((rc = upload(sc, H.cell_blk, &S.cell_blk))) return rc;
  S.n_grids = (int)H.grids.size();
  sc->D = S.n_cols; sc->N = S.n_nodes; sc->L = S.n_shapes; sc->O = S.n_obs; sc->P = S.n_pairs;
  sc->n_slots = S.n_slots;
  sc->max_shape_node = H.rshapes.empty() ? -1 : H.rshapes.back().node;
  S.all_blk = H.all_blk;
  // Exact entries per sweep wavefront (64 configurations).  Box / ball scenes hand the exact stage the rare
  // box pairs whose best axis shows a gap inside the margin; with meshes (URDF cylinders included) every
  // candidate whose bounding boxes are closer than the margin goes there.  A wavefront that needs more is
  // re-checked by the fused kernel (GEAR_HIP_EXACT_CAP overrides: tests force that).
  sc->has_mesh = !H.verts.empty();
  {
    // 256 with meshes / 128 without: twice what the busiest measured scene fills (profiles/r03/exact_cap.txt: 42-120
    // entries per wavefront on average over five scenes, no full region in 16384 wavefronts at these capacities; the
    // round-2 sizes, 512 / 128, reserved 12x the mean).  An input distribution that needs more is noticed — a wavefront
    // whose region runs full says so in a host word — and the capacity doubles for the scene's later calls.
    size_t cap = sc->has_mesh ? 256 : 128;
    if (sc->knobs.exact_cap0) cap = sc->knobs.exact_cap0;
    if (sc->knobs.exact_cap) cap = sc->knobs.exact_cap;
    sc->exact_cap = sc->exact_cap0 = (unsigned)std::min<size_t>(std::max<size_t>(cap, 1), 4096);
  }
  return GH_OK;
}

// How many pose columns a launch needs and where they live.
struct StorePlan {
  size_t lds_bytes = 0;  // dynamic LDS per workgroup (one wavefront)
  bool global = false;
  size_t slots = 0;
};

StorePlan plan_store(const gh_scene* sc, bool shape_poses) {
  StorePlan p;
  p.slots = (size_t)sc->n_slots + (shape_poses ? (size_t)sc->L : 0);
  size_t bytes = p.slots * 7 * kWave * sc->elem();
  if (bytes > kLdsBudget) p.global = true;
  else p.lds_bytes = bytes;
  return p;
}

// Runs `launch(offset, count, gstore, gstride)` over [0, n), in chunks when the
// pose columns live in a global workspace.
template <class F>
int for_chunks(gh_scene* sc, size_t n, const StorePlan& plan, F&& launch, Carve* fixed = nullptr) {
  if (n == 0) return GH_OK;
  if (!plan.global) {
    launch((size_t)0, n, nullptr, (size_t)0);
    GH_HIP(hipGetLastError());
    return GH_OK;
  }
  size_t chunk = std::min(n, kChunk);
  size_t stride = (chunk + kWave - 1) / kWave * kWave;
  void* store;
  if (fixed) {
    store = fixed->take(plan.slots * 7 * stride * sc->elem());
    if (!store) return fail(GH_ERR_TOO_LARGE, "this scene's pose columns for " + std::to_string(n) + " configurations need " +
                            std::to_string(fixed->used) + " bytes; the no-allocation workspace holds " + std::to_string(fixed->bytes));
  } else {
    int rc = sc->store.ensure(plan.slots * 7 * stride * sc->elem());
    if (rc) return rc;
    store = sc->store.p;
  }
  for (size_t off = 0; off < n; off += chunk) {
    launch(off, std::min(chunk, n - off), store, stride);
    GH_HIP(hipGetLastError());
  }
  return GH_OK;
}

unsigned grid_for(size_t n) { return (unsigned)((n + kWave - 1) / kWave); }

// shapes per sweep unit: the instantiation that wastes the fewest shape slots (ties: the larger K)
int pick_chunk(int n_shapes) {
  const int ks[3] = {9, 7, 5};
  int best = 9, waste = 1 << 30;
  for (int k : ks) {
    int w = (n_shapes + k - 1) / k * k - n_shapes;
    if (w < waste) { waste = w; best = k; }
  }
  return best;
}

template <class T, int K>
void launch_check_k(gh_scene* sc, const SceneDev<T>& S, const T* q, size_t cnt, uint32_t what, uint8_t* flags,
                    hipStream_t st, bool counted, const StorePlan& plan, void* gs, size_t gstride, const unsigned* hit) {
  const size_t lds = plan.lds_bytes + queue_bytes(K);
  // (hit: the per-block re-check marks of the pipeline; a workgroup then looks at 64 of them)
  const dim3 grid(hit ? (grid_for(cnt) + kWave - 1) / kWave : grid_for(cnt)), block(kWave);
  if (plan.global) {
    if (counted) hipLaunchKernelGGL((k_check<T, true, true, K>), grid, block, lds, st, S, q, cnt, what, flags, sc->counters, (T*)gs, gstride, hit);
    else hipLaunchKernelGGL((k_check<T, false, true, K>), grid, block, lds, st, S, q, cnt, what, flags, sc->counters, (T*)gs, gstride, hit);
  } else {
    if (counted) hipLaunchKernelGGL((k_check<T, true, false, K>), grid, block, lds, st, S, q, cnt, what, flags, sc->counters, (T*)gs, gstride, hit);
    else hipLaunchKernelGGL((k_check<T, false, false, K>), grid, block, lds, st, S, q, cnt, what, flags, sc->counters, (T*)gs, gstride, hit);
  }
}

template <class T>
void launch_check(gh_scene* sc, const SceneDev<T>& S, const T* q, size_t cnt, uint32_t what, uint8_t* flags,
                  hipStream_t st, bool counted, const StorePlan& plan, void* gs, size_t gstride,
                  const unsigned* hit = nullptr) {
  switch (pick_chunk(sc->L)) {
    case 5: launch_check_k<T, 5>(sc, S, q, cnt, what, flags, st, counted, plan, gs, gstride, hit); break;
    case 7: launch_check_k<T, 7>(sc, S, q, cnt, what, flags, st, counted, plan, gs, gstride, hit); break;
    default: launch_check_k<T, 9>(sc, S, q, cnt, what, flags, st, counted, plan, gs, gstride, hit); break;
  }
}

// ---- the pipeline (gear_pipeline.hpp) ------------------------------------------------------------
// Every scene whose sweep fits the LDS takes the pipeline at every batch size: an RRT line of a few hundred
// samples costs three small launches against ~100 us for the fused kernel, whose one wavefront sweeps all
// L x O pairs and runs an exact stage of mixed pair classes.
constexpr size_t kMultiStreamMin = size_t(6) << 20;  // batches from here on rotate their chunks over several streams
constexpr uint32_t kForceFused = 0x1000u, kForcePipe = 0x2000u;  // `what` bits for tests and profiling
constexpr uint32_t kNoGrid = 0x4000u;                             // pipeline without the broadphase grid
constexpr size_t kLdsCu = 160 * 1024;

// LDS layout of one k_sweep_n1 launch (see the kernel): cull table + robot hot table per workgroup, then
// per wavefront the FK branch slots, the poses of the "earlier" self-pair shapes, the ring and the hit words
struct SweepPlan {
  bool ok = false, clds = false;
  int qmode = Q_MEM;
  unsigned table = 0, per_wave = 0, first = 0;
  int waves = 4;
  size_t lds() const { return table + rhot + (size_t)per_wave * waves; }
  unsigned rhot = 0;
};

template <class T>
SweepPlan plan_sweep(const gh_scene* sc, const SceneDev<T>& S, bool do_env, bool do_self) {
  SweepPlan best;
  const size_t slot_bytes = (size_t)S.n_slots * 7 * kWave * sizeof(T);
  const size_t first_bytes = do_self ? (size_t)S.n_first * 3 * kWave * sizeof(T) : 0;  // centres; the rotations wait in memory
  const size_t q_bytes = (size_t)S.n_cols * kWave * sizeof(T);
  const size_t base_wave = slot_bytes + first_bytes + (size_t)kRing * 8 * sizeof(T) + 32 + (size_t)kHitList * 4;
  const size_t rhot = ((size_t)S.n_shapes * 4 * sizeof(T) + 15) / 16 * 16;
  const size_t table = (size_t)(S.n_obs + 1) * 4 * sizeof(float);  // in LDS the cull table is f32 in both precisions (k_sweep_n1)
  const size_t reg_waves = sizeof(T) == 4 ? 16 : 8;  // wavefronts per CU the registers allow (128 / 215 VGPRs)
  // The cull table in LDS takes the cell blocks' record fetches (and the re-reads of the per-hit passes) off
  // the texture-address path, which is what this kernel otherwise queues on: it wins even where it costs a
  // resident wavefront or two, so it is taken whenever the table is at most 16 KB and the workgroup still
  // fits; the workgroup size is then the one that keeps most wavefronts resident.
  const bool want_clds = do_env && sc->knobs.sweep_clds && table <= 16 * 1024;
  size_t best_occ = 0;
  for (int clds = want_clds ? 1 : 0; clds >= 0 && !best.ok; clds--) {
    // the batch row moves into an LDS column when that does not cost a resident wavefront (first hit wins ties)
    for (int qmode : {Q_LDS, Q_MEM}) {
      if (sc->knobs.sweep_q >= 0 && qmode != sc->knobs.sweep_q) continue;
      const size_t per_wave = (base_wave + (qmode == Q_LDS ? q_bytes : 0) + 15) / 16 * 16;
      for (int wv : {4, 2, 1}) {
        if (sc->knobs.sweep_waves && wv != sc->knobs.sweep_waves) continue;
        const size_t wg = (clds ? table : 0) + rhot + per_wave * wv;
        if (wg > kLdsCu) continue;
        const size_t occ = std::min(reg_waves, kLdsCu / wg * wv);
        if (occ > best_occ) {  // ties: the larger workgroup (one table copy serves more wavefronts)
          best_occ = occ;
          best.ok = true;
          best.clds = clds != 0;
          best.table = clds ? (unsigned)table : 0u;
          best.rhot = (unsigned)rhot;
          best.per_wave = (unsigned)per_wave;
          best.first = (unsigned)first_bytes;
          best.waves = wv;
          best.qmode = qmode;
        }
      }
    }
  }
  return best;
}

template <class T>
int pipeline_dev(gh_scene* sc, const SceneDev<T>& S, const T* q, size_t n, uint32_t what, uint8_t* flags,
                 hipStream_t st, Carve* fixed = nullptr) {
  const bool do_self = (what & 2u) && S.n_spairs > 0;
  const bool do_env = (what & 1u) && S.n_obs > 0;
  const SweepPlan sp = plan_sweep<T>(sc, S, do_env, do_self);
  if (!sp.ok) return fail(GH_ERR_TOO_LARGE, "internal: pipeline chosen for a scene whose sweep does not fit the LDS");
  // a wavefront of an earlier call on this scene found its exact region full (it was re-checked by the fused path: slow,
  // not wrong): later calls get twice the room.  Tests that force a capacity (GEAR_HIP_EXACT_CAP) keep it.
  if (!fixed && sc->hover.p && !sc->knobs.exact_cap) {
    volatile unsigned* note = (volatile unsigned*)sc->hover.p;
    if (*note != 0 && sc->exact_cap < 2048) {
      *note = 0;
      sc->exact_cap *= 2;
    }
  }
  const unsigned capw = fixed ? sc->exact_cap0 : sc->exact_cap;
  // keep a chunk's exact arrays under 3 GiB (they are reserved for the worst case and touched only where entries
  // are written: capacity, not traffic; a 2M chunk of a scene with meshes and self pairs is 1.2 GiB in f32,
  // 2.3 GiB in f64)
  const size_t entry_bytes = 8 + 8 * sizeof(T) * (do_self ? 2 : 1);
  const size_t max_waves = std::max<size_t>((size_t(3) << 30) / (entry_bytes * capw), 64);
  const size_t n_pad = (n + kWave - 1) / kWave * kWave;
  const size_t chunk_max = std::min({n_pad, sc->knobs.pipe_chunk, max_waves * kWave, (size_t)1 << 27});
  // the fused kernel is the last stage (re-checks of full exact regions): it needs its own pose columns
  StorePlan plan = plan_store(sc, true);
  // Consecutive chunks are independent.  With several workspace sets they rotate over the caller's
  // stream and the aux streams (fork / join through events), so that one chunk's exact stage, which
  // cannot fill the chip, overlaps the next chunk's sweep.
  // Forking onto the scene's streams and joining back costs ~0.1 ms per call whatever the size, and one large chunk
  // runs its kernels' tails once: up to 4M configurations a single chunk on the caller's stream beats four quarters
  // on four streams by 4-8 %, from 6M on the streams win by 0-6 % (profiles/r02/chunking.txt), so only batches of
  // at least kMultiStreamMin configurations rotate over several streams.
  // Chunks of one call are equal (a short last chunk would leave its stream idle while the others finish), and a
  // call that rotates over k streams is cut into a multiple of k chunks: 4 x 1M beats 2 x 2M, 4 x 2M beats 8 x 1M
  // (MI355X, four scenes: profiles/r02/chunking.txt).
  int n_sets = (!plan.global && n >= kMultiStreamMin) ? sc->knobs.pipe_streams : 1;
  if (fixed) n_sets = 1;
  size_t n_chunks = (n_pad + chunk_max - 1) / chunk_max;
  if (n_sets > 1) n_chunks = (n_chunks + (size_t)n_sets - 1) / (size_t)n_sets * (size_t)n_sets;
  const size_t chunk = std::min(chunk_max, ((n_pad + n_chunks - 1) / n_chunks + kWave - 1) / kWave * kWave);
  n_sets = (int)std::min<size_t>(n_sets, (n + chunk - 1) / chunk);
  const bool two = n_sets > 1;
  const size_t waves = chunk / kWave;
  PipeWs<T> Ws[kMaxSets];
  int rc;
  void* gs = nullptr;
  size_t gstride = 0;
  if (fixed) {  // the same arrays, cut from the fixed region: nothing is allocated, nothing of the scene's is touched
    PipeWs<T>& W = Ws[0];
    W = PipeWs<T>{};
    W.capw = capw;
    W.note = nullptr;
    W.hit = (unsigned*)fixed->take(chunk * 4);
    W.xcnt = (unsigned*)fixed->take(waves * 4);
    W.xkey = (uint2*)fixed->take(waves * capw * 8);
    W.xpose_a = (T*)fixed->take(waves * capw * 8 * sizeof(T));
    W.xpose_b = do_self ? (T*)fixed->take(waves * capw * 8 * sizeof(T)) : nullptr;
    W.fq = do_self ? (T*)fixed->take(std::max<size_t>(waves * (size_t)S.n_first * kWave * 4 * sizeof(T), 16)) : nullptr;
    W.pc = (PipeCounters*)fixed->take(sizeof(PipeCounters));
    if (plan.global) {
      gs = fixed->take(plan.slots * 7 * chunk * sizeof(T));
      gstride = chunk;
    }
    if (fixed->used > fixed->bytes) {
      const size_t per_wave = fixed->used / waves + 1;
      return fail(GH_ERR_TOO_LARGE, "the workspace of this scene for " + std::to_string(n) + " configurations is " +
                  std::to_string(fixed->used) + " bytes; the no-allocation workspace holds " + std::to_string(fixed->bytes) +
                  " (about " + std::to_string(fixed->bytes / per_wave * kWave) + " configurations of this scene per call)");
    }
  }
  PipePool& pool = g_pool[sc->device];
  std::unique_lock<std::mutex> pool_lock(pool.mu, std::defer_lock);
  if (!fixed) pool_lock.lock();  // held until every launch of this call is enqueued
  for (int k = 0; k < n_sets && !fixed; k++) {
    PipeWs<T>& W = Ws[k];
    PipeBufs& b = pool.sets[k];
    if (!b.last_use) GH_HIP(hipEventCreateWithFlags(&b.last_use, hipEventDisableTiming));
    // whatever used this set last (another scene, another stream) finishes before this call touches it
    if (b.used) GH_HIP(hipStreamWaitEvent(k ? sc->aux[k] : st, b.last_use, 0));
    W = PipeWs<T>{};
    W.capw = capw;
    W.note = (unsigned*)sc->hover.d;
    if ((rc = b.hit.ensure(chunk * 4)) || (rc = b.xcnt.ensure(waves * 4)) || (rc = b.xkey.ensure(waves * capw * 8)) ||
        (rc = b.xpose_a.ensure(waves * capw * 8 * sizeof(T))) ||
        (do_self && (rc = b.xpose_b.ensure(waves * capw * 8 * sizeof(T)))) ||
        (do_self && (rc = b.fq.ensure(std::max<size_t>(waves * (size_t)S.n_first * kWave * 4 * sizeof(T), 16)))) ||
        (rc = b.pc.ensure(sizeof(PipeCounters))))
      return rc;
    W.hit = (unsigned*)b.hit.p;
    W.xcnt = (unsigned*)b.xcnt.p;
    W.xkey = (uint2*)b.xkey.p;
    W.xpose_a = (T*)b.xpose_a.p;
    W.xpose_b = do_self ? (T*)b.xpose_b.p : nullptr;
    W.fq = do_self ? (T*)b.fq.p : nullptr;
    W.pc = (PipeCounters*)b.pc.p;
  }
  if (!fixed) sc->last_sets = n_sets;
  if (plan.global && !fixed) {
    if ((rc = sc->store.ensure(plan.slots * 7 * chunk * sizeof(T)))) return rc;
    gs = sc->store.p;
    gstride = chunk;
  }
  if (two) {
    GH_HIP(hipEventRecord(sc->ev_fork, st));
    for (int k = 1; k < n_sets; k++) GH_HIP(hipStreamWaitEvent(sc->aux[k], sc->ev_fork, 0));
  }
  const int all_blk = (what & kNoGrid) ? S.all_blk : -1;
  size_t ci = 0;
  for (size_t off = 0; off < n; off += chunk, ci++) {
    const size_t cnt = std::min(chunk, n - off);
    const T* qc = q + off * sc->D;
    const int set = (int)(ci % (size_t)n_sets);
    PipeWs<T>& W = Ws[set];
    W.flags = flags + off;
    W.n_cfg = (unsigned)cnt;
    hipStream_t cs = set ? sc->aux[set] : st;
    if ((what & 0x8000u) || sc->knobs.n2_hist) GH_HIP(hipMemsetAsync(W.pc, 0, sizeof(PipeCounters), cs));
    const unsigned n_sw = grid_for(cnt);  // sweep wavefronts
    {
      const dim3 grid((n_sw + sp.waves - 1) / sp.waves), block(kWave * sp.waves);
      auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, grid, block, sp.lds(), cs, S, qc, cnt, what, W, sp.table, sp.per_wave, sp.first, all_blk);
      };
      switch ((sp.clds ? 4 : 0) + sp.qmode) {
        case 0: go(k_sweep_n1<T, false, Q_MEM>); break;
        case 2: go(k_sweep_n1<T, false, Q_LDS>); break;
        case 4: go(k_sweep_n1<T, true, Q_MEM>); break;
        default: go(k_sweep_n1<T, true, Q_LDS>); break;
      }
    }
    {
      // the mesh pools (4-wide records) join the workgroup in LDS when they are at most 48 KB
      const size_t mesh_bytes = (size_t)(S.n_verts + S.n_planes) * 4 * sizeof(T);
      // ~1000 workgroups per launch: tiles of 32 sweep wavefronts from 1.5M-configuration chunks on (fewer, fuller
      // workgroups; +1.5..3.5 % at 2M), tiles of 16 below (32 there: -5..-35 %, the launch no longer fills the chip)
      const int tile = sc->knobs.n2_tile == 8 || sc->knobs.n2_tile == 16 || sc->knobs.n2_tile == 32 ? sc->knobs.n2_tile
                       : (chunk >= (size_t(3) << 19) ? 32 : 16);
      const dim3 grid((n_sw + tile - 1) / tile), block(kN2Threads);
      const bool mlds = sc->knobs.n2_mlds && mesh_bytes > 0 && mesh_bytes <= 48 * 1024;
      auto go = [&](auto kern, size_t lds) { hipLaunchKernelGGL(kern, grid, block, lds, cs, S, what, W, n_sw); };
      if (sc->knobs.n2_hist) {
        go(k_narrow2<T, false, true, 16>, 0);
        GH_HIP(hipStreamSynchronize(cs));
        PipeCounters h;
        GH_HIP(hipMemcpy(&h, W.pc, sizeof(h), hipMemcpyDeviceToHost));
        for (int r = 0; r < 2; r++) {
          fprintf(stderr, r ? "k_narrow2 largest GJK iteration count per wavefront:" : "k_narrow2 GJK iterations per item:");
          for (int k = 0; k < 48; k++) fprintf(stderr, " %u", h.hist[r][k]);
          fprintf(stderr, "\n");
        }
      } else if (mlds) {
        if (tile == 8) go(k_narrow2<T, true, false, 8>, mesh_bytes);
        else if (tile == 32) go(k_narrow2<T, true, false, 32>, mesh_bytes);
        else go(k_narrow2<T, true, false, 16>, mesh_bytes);
      } else {
        if (tile == 8) go(k_narrow2<T, false, false, 8>, 0);
        else if (tile == 32) go(k_narrow2<T, false, false, 32>, 0);
        else go(k_narrow2<T, false, false, 16>, 0);
      }
    }
    launch_check<T>(sc, S, qc, cnt, what, flags + off, cs, false, plan, gs, gstride, W.xcnt);
    GH_HIP(hipGetLastError());
  }
  for (int k = 0; k < n_sets && !fixed; k++) {
    GH_HIP(hipEventRecord(pool.sets[k].last_use, k ? sc->aux[k] : st));
    pool.sets[k].used = true;
  }
  if (two) {
    for (int k = 1; k < n_sets; k++) {
      GH_HIP(hipEventRecord(sc->ev_join[k], sc->aux[k]));
      GH_HIP(hipStreamWaitEvent(st, sc->ev_join[k], 0));
    }
  }
  return GH_OK;
}

// ---- small batches (gear_small.hpp): one launch, a configuration's work spread over many lanes ------
constexpr uint32_t kForceSmall = 0x20u;
constexpr size_t kSmallLdsMax = 64 * 1024;

struct SmallEdges {  // gh_check_edges through k_small: the first-bad reduction rides in the same launch
  const uint32_t* edge_off = nullptr;
  size_t n_edges = 0;
  uint8_t bad_mask = 0;
  int32_t* first_bad = nullptr;
  uint8_t* flags_near = nullptr;  // device copy of the flags when the caller's are host memory
  unsigned* notify = nullptr;     // host word (as the device addresses it) that receives `seq` when the call's results are in host memory
  unsigned seq = 0;
};
constexpr size_t kSmallEdgesMax = 256;  // edges up to which the reduction rides in the k_small launch

template <class T>
SmallPlan small_plan_of(const gh_scene* sc, const SceneDev<T>& S, size_t n) {
  return plan_small<T>(S.n_nodes, S.n_shapes, S.n_spairs, S.n_limits, S.n_slots, S.n_obs, S.n_cols, sc->knobs.small_qcap, n);
}

template <class T>
int small_dev(gh_scene* sc, const SceneDev<T>& S, const T* q, size_t n, uint32_t what, uint8_t* flags, hipStream_t st,
              const SmallEdges* ed) {
  const SmallPlan sp = small_plan_of<T>(sc, S, n);
  const int all_blk = (what & kNoGrid) ? S.all_blk : -1;
  const size_t groups = (n + (size_t)sp.C - 1) / (size_t)sp.C;
  hipLaunchKernelGGL((k_small<T>), dim3((unsigned)groups), dim3(kSmThreads), sp.lds, st, S, q, n, what, flags, sp.C, sp.qcap, all_blk, sp.cull_lds ? 1 : 0,
                     ed ? ed->edge_off : nullptr, ed ? (unsigned)ed->n_edges : 0u, ed ? (unsigned)ed->bad_mask : 0u,
                     ed ? ed->first_bad : nullptr, ed ? ed->flags_near : nullptr, sc->small_done, ed ? ed->notify : nullptr,
                     ed ? ed->seq : 0u);
  GH_HIP(hipGetLastError());
  return GH_OK;
}

// does this call take k_small?
bool small_takes(const gh_scene* sc, size_t n, uint32_t what, bool counted) {
  if (counted || !(what & 3u) || sc->L <= 0 || n == 0 || n >= (size_t(1) << 30) || sc->N > 4000) return false;  // (the walk packs slot indices in 12 bits)
  if (what & (kForceFused | kForcePipe | 0x8000u)) return false;
  if (!(what & kForceSmall) && n > (sc->precision == 32 ? sc->knobs.small_max : sc->knobs.small_max * 3 / 8)) return false;
  const size_t lds = sc->precision == 32 ? small_plan_of<float>(sc, sc->s32, n).lds : small_plan_of<double>(sc, sc->s64, n).lds;
  return lds <= kSmallLdsMax;
}

// Waits until the kernel of a small host-buffer call has stored `seq` in the scene's host word: a short poll,
// then the stream's own completion (which also reports a kernel that died).  The stream is left to drain by
// itself: the next launch on it is ordered behind the kernel's last instructions anyway.
// the scene's notify word: allocated on first use and ZEROED (pinned blocks are recycled: a word left behind by a
// destroyed scene could otherwise equal this scene's first sequence number)
int ensure_notify_word(gh_scene* sc) {
  if (sc->hseq.p) return GH_OK;
  int rc = sc->hseq.ensure(64);
  if (rc) return rc;
  std::memset(sc->hseq.p, 0, 64);
  sc->seq = 0;
  return GH_OK;
}

int wait_notified(gh_scene* sc, unsigned seq) {
  const volatile unsigned* w = (const volatile unsigned*)sc->hseq.p;
  for (int spin = 0; spin < 20000; spin++) {
    if (__atomic_load_n(w, __ATOMIC_ACQUIRE) == seq) return GH_OK;
    __builtin_ia32_pause();
  }
  GH_HIP(hipStreamSynchronize(sc->stream));
  if (__atomic_load_n(w, __ATOMIC_ACQUIRE) != seq) return fail(GH_ERR_HIP, "internal: a small-batch kernel finished without reporting");
  return GH_OK;
}

int check_dev(gh_scene* sc, const void* q, size_t n, uint32_t what, uint8_t* flags, hipStream_t st, bool counted,
              Carve* fixed = nullptr) {
#ifndef GEAR_HIP_ABLATE
  what &= kWhatPublic;  // undeclared bits never reach a kernel (the ablation build of tools/ keeps them)
#endif
  if (!(sc->margin > 0)) what &= ~3u;  // distance >= 0 is never < a non-positive margin: only limits can fail
  if (small_takes(sc, n, what, counted)) {
    return sc->precision == 32 ? small_dev<float>(sc, sc->s32, (const float*)q, n, what, flags, st, nullptr)
                               : small_dev<double>(sc, sc->s64, (const double*)q, n, what, flags, st, nullptr);
  }
  bool pipe_ok = !counted && (what & 3u) && sc->L > 0 && !(what & kForceFused);
  if (pipe_ok && sc->knobs.small_fused && n < (size_t(1) << 14) && !(what & kForcePipe)) pipe_ok = false;
  if (pipe_ok) {
    const bool ds = (what & 2u) && sc->P > 0, de = (what & 1u) && sc->O > 0;
    pipe_ok = sc->precision == 32 ? plan_sweep<float>(sc, sc->s32, de, ds).ok : plan_sweep<double>(sc, sc->s64, de, ds).ok;
  }
  if (pipe_ok) {
    if (n == 0) return GH_OK;
    return sc->precision == 32 ? pipeline_dev<float>(sc, sc->s32, (const float*)q, n, what, flags, st, fixed)
                               : pipeline_dev<double>(sc, sc->s64, (const double*)q, n, what, flags, st, fixed);
  }
  StorePlan plan = plan_store(sc, true);  // the batched kernel keeps every shape pose in the lane's column
  return for_chunks(sc, n, plan, [&](size_t off, size_t cnt, void* gs, size_t gstride) {
    if (sc->precision == 32)
      launch_check<float>(sc, sc->s32, (const float*)q + off * sc->D, cnt, what, flags + off, st, counted, plan, gs, gstride);
    else
      launch_check<double>(sc, sc->s64, (const double*)q + off * sc->D, cnt, what, flags + off, st, counted, plan, gs, gstride);
  }, fixed);
}

int fk_dev(gh_scene* sc, const void* q, size_t n, void* poses, hipStream_t st) {
  StorePlan plan = plan_store(sc, false);
  return for_chunks(sc, n, plan, [&](size_t off, size_t cnt, void* gs, size_t gstride) {
    if (sc->precision == 32)
      hipLaunchKernelGGL((k_fk<float>), dim3(grid_for(cnt)), dim3(kWave), plan.lds_bytes, st, sc->s32,
                         (const float*)q + off * sc->D, cnt, (float*)poses + off * sc->N * 7, (float*)gs, gstride);
    else
      hipLaunchKernelGGL((k_fk<double>), dim3(grid_for(cnt)), dim3(kWave), plan.lds_bytes, st, sc->s64,
                         (const double*)q + off * sc->D, cnt, (double*)poses + off * sc->N * 7, (double*)gs, gstride);
  });
}

struct Guard {
  gh_scene* sc;
  int prev = -1;
  bool ok = false;
  explicit Guard(gh_scene* s) : sc(s) {
    sc->mu.lock();
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = hipSetDevice(sc->device) == hipSuccess;
  }
  ~Guard() {
    if (prev >= 0) (void)hipSetDevice(prev);
    sc->mu.unlock();
  }
};

// ---- device groups (gh_scene_create_multi) --------------------------------------------------------
// The batch shards naturally: configurations are independent and an edge needs only its own samples.  A
// group gives every listed device a contiguous share (whole edges), runs the shares concurrently — one
// host thread per device, each through the ordinary single-device entry point with its own stream,
// staging buffers and workspace — and the results land in the caller's buffers.  No collective: the
// scene was replicated at creation.
template <class F>
int run_parts(gh_scene* sc, size_t n_parts, F&& body) {  // body(k) -> status; errors keep their text
  std::vector<int> rc(n_parts, GH_OK);
  std::vector<std::string> msg(n_parts);
  std::vector<std::thread> pool;
  auto guarded = [&](size_t k) {  // nothing may throw across the extern "C" boundary (nor out of a thread)
    try {
      if (sc->knobs.fail_part == (int)k) throw std::runtime_error("share " + std::to_string(k) + " was told to fail (GEAR_HIP_TEST_FAIL_PART)");
      rc[k] = body(k);
      if (rc[k]) msg[k] = g_last_error;
    } catch (const std::exception& e) {
      rc[k] = GH_ERR_HIP;
      msg[k] = std::string("exception: ") + e.what();
    }
  };
  for (size_t k = 1; k < n_parts; k++) {
    try {
      pool.emplace_back(guarded, k);
    } catch (const std::system_error&) {  // no thread to be had: this share runs on the calling thread
      guarded(k);
    }
  }
  guarded(0);
  for (auto& t : pool) t.join();
  for (size_t k = 0; k < n_parts; k++)
    if (rc[k]) return fail(rc[k], "device " + std::to_string(sc->parts[k]->device) + ": " + msg[k]);
  return GH_OK;
}

}  // namespace

// ---- copies out of PAGEABLE host memory ------------------------------------------------------------------
// hipMemcpyAsync from pageable memory is staged by the runtime on the calling thread: 24-27 GB/s on the GPU box, half of what
// the bus carries from page-locked memory (profiles/r02/host_buffers.txt).  A large batch from pageable memory is therefore
// copied into the scene's page-locked staging by FOUR host threads (the caller and three helpers that the library keeps
// parked between calls) one piece ahead of the bus, and crosses the bus from there.
namespace {
class CopyPool {
 public:
  static CopyPool& get() {
    static CopyPool* p = new CopyPool;  // never destroyed: its threads only ever touch host memory, and are parked at exit
    return *p;
  }
  // dst[0, bytes) = src[0, bytes); returns when done.  One copy at a time (large host-buffer calls queue here).
  void copy(void* dst, const void* src, size_t bytes) {
    if (bytes < (size_t(1) << 20) || !ok_) { std::memcpy(dst, src, bytes); return; }
    std::lock_guard<std::mutex> one(call_mu_);
    const size_t slice = (bytes / (kHelpers + 1) + 4095) & ~size_t(4095);
    {
      std::lock_guard<std::mutex> lk(mu_);
      dst_ = (char*)dst; src_ = (const char*)src; bytes_ = bytes; slice_ = slice;
      pending_ = kHelpers;
      generation_++;
    }
    cv_.notify_all();
    std::memcpy(dst, src, std::min(slice, bytes));  // the caller's own slice
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [&] { return pending_ == 0; });
  }

 private:
  static constexpr int kHelpers = 3;
  CopyPool() {
    try {
      for (int k = 0; k < kHelpers; k++) std::thread([this, k] { run(k + 1); }).detach();
      ok_ = true;
    } catch (const std::system_error&) {
      ok_ = false;  // no threads to be had: plain memcpy
    }
  }
  void run(int k) {
    unsigned seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> lk(mu_);
      cv_.wait(lk, [&] { return generation_ != seen; });
      seen = generation_;
      char* d = dst_; const char* s = src_;
      const size_t lo = std::min(bytes_, slice_ * (size_t)k), hi = std::min(bytes_, slice_ * (size_t)(k + 1));
      lk.unlock();
      if (hi > lo) std::memcpy(d + lo, s + lo, hi - lo);
      lk.lock();
      if (--pending_ == 0) done_.notify_all();
    }
  }
  std::mutex call_mu_, mu_;
  std::condition_variable cv_, done_;
  char* dst_ = nullptr;
  const char* src_ = nullptr;
  size_t bytes_ = 0, slice_ = 0;
  int pending_ = 0;
  unsigned generation_ = 0;
  bool ok_ = false;
};

// is this host pointer page-locked memory the device can read directly (hipHostMalloc / hipHostRegister)?
bool is_page_locked(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // an ordinary (pageable) pointer is reported as an error: clear it
    return false;
  }
  return a.type == hipMemoryTypeHost;
}
}  // namespace

// A large host-buffer batch, the scene's Guard held: device staging sc->q / sc->flags and page-locked sc->hflags are
// ensured by the caller.  Inputs travel in pieces of kHostPiece on a stream of their own (a copy out of pageable memory
// returns when the runtime has staged it: the device evaluates piece i meanwhile); the flags come back into
// page-locked memory, so that no copy into the caller's pageable array makes the stream wait, and are handed over one
// piece behind.  `tail` enqueues on sc->stream whatever has to follow the last piece (gh_check_edges: the first-bad
// reduction over sc->flags); the stream is drained before the function returns.
template <class Tail>
static int host_pieces(gh_scene* sc, const void* q, size_t n, uint32_t what, uint8_t* flags, Tail&& tail) {
  int rc;
  for (int k = 0; k < 2; k++) {
    if (!sc->ev_h2d[k]) GH_HIP(hipEventCreateWithFlags(&sc->ev_h2d[k], hipEventDisableTiming));
    if (!sc->ev_d2h[k]) GH_HIP(hipEventCreateWithFlags(&sc->ev_d2h[k], hipEventDisableTiming));
  }
  const size_t row = (size_t)sc->D * sc->elem();
  // pieces of 1M configurations; a batch of less than 4M is cut into four (at least 256k each) so that it overlaps too
  const size_t piece = n >= 4 * kHostPiece ? kHostPiece : std::max<size_t>(kStagedMax, ((n + 3) / 4 + 65535) / 65536 * 65536);
  hipStream_t in = sc->aux[kMaxSets - 1];
  size_t prev_off = 0, prev_cnt = 0;
  int i = 0;
  // pageable source: pieces go through two page-locked staging slots, filled by the copy pool while the previous piece is
  // on the bus (slot i & 1 is free again once the copy recorded in ev_h2d[i & 1] two pieces ago has finished)
  const bool staged = !is_page_locked(q);
  if (staged && (rc = sc->hq.ensure(2 * piece * row))) return rc;
  for (size_t off = 0; off < n; off += piece, i++) {
    const size_t cnt = std::min(piece, n - off);
    const char* src = (const char*)q + off * row;
    if (staged) {
      char* slot = (char*)sc->hq.p + (size_t)(i & 1) * piece * row;
      if (i >= 2) GH_HIP(hipEventSynchronize(sc->ev_h2d[i & 1]));
      CopyPool::get().copy(slot, src, cnt * row);
      src = slot;
    }
    GH_HIP(hipMemcpyAsync((char*)sc->q.p + off * row, src, cnt * row, hipMemcpyHostToDevice, in));
    GH_HIP(hipEventRecord(sc->ev_h2d[i & 1], in));
    GH_HIP(hipStreamWaitEvent(sc->stream, sc->ev_h2d[i & 1], 0));
    if ((rc = check_dev(sc, (char*)sc->q.p + off * row, cnt, what, (uint8_t*)sc->flags.p + off, sc->stream, false))) {
      (void)hipStreamSynchronize(sc->stream);
      return rc;
    }
    GH_HIP(hipMemcpyAsync((char*)sc->hflags.p + off, (char*)sc->flags.p + off, cnt, hipMemcpyDeviceToHost, sc->stream));
    if (prev_cnt) {  // (its event is recorded again only two pieces on)
      GH_HIP(hipEventSynchronize(sc->ev_d2h[(i - 1) & 1]));
      std::memcpy(flags + prev_off, (char*)sc->hflags.p + prev_off, prev_cnt);
    }
    GH_HIP(hipEventRecord(sc->ev_d2h[i & 1], sc->stream));
    prev_off = off;
    prev_cnt = cnt;
  }
  if ((rc = tail())) {
    (void)hipStreamSynchronize(sc->stream);
    return rc;
  }
  GH_HIP(hipStreamSynchronize(sc->stream));
  std::memcpy(flags + prev_off, (char*)sc->hflags.p + prev_off, prev_cnt);
  return GH_OK;
}

extern "C" {

const char* gh_last_error(void) { return g_last_error.c_str(); }
const char* gh_version(void) { return "gear_hip 0.1 (gfx950)"; }

int gh_scene_create(const gh_robot_desc* robot, const gh_obstacle_desc* obstacles, double margin, int precision,
                    int device, gh_scene** out) {
  if (!robot || !obstacles || !out) return fail(GH_ERR_INVALID, "null argument");
  if (precision != 32 && precision != 64) return fail(GH_ERR_INVALID, "precision must be 32 or 64");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) return fail(GH_ERR_NO_DEVICE, "no HIP device: this library has no CPU path");
  if (device < 0 || device >= ndev) return fail(GH_ERR_INVALID, "device index out of range");
  gh_scene* sc = new (std::nothrow) gh_scene;
  if (!sc) return fail(GH_ERR_INVALID, "out of host memory");
  sc->precision = precision;
  sc->device = device;
  sc->margin = margin;
  sc->knobs = Knobs::from_env();
  int rc;
  {
    Guard g(sc);
    if (!g.ok) rc = fail(GH_ERR_HIP, "hipSetDevice failed");
    else {
      rc = precision == 32 ? build_and_upload<float>(sc, *robot, *obstacles, margin, sc->s32)
                           : build_and_upload<double>(sc, *robot, *obstacles, margin, sc->s64);
      if (rc == GH_OK && hipStreamCreateWithFlags(&sc->stream, hipStreamNonBlocking) != hipSuccess)
        rc = fail(GH_ERR_HIP, "hipStreamCreate failed");
      void* ws = nullptr;  // this device's copy of the no-allocation workspace: its address is looked up here, once
      if (rc == GH_OK) rc = scratch_on_device(device, &ws);
      if (rc == GH_OK && hipEventCreateWithFlags(&sc->ev_fork, hipEventDisableTiming) != hipSuccess)
        rc = fail(GH_ERR_HIP, "hipEventCreate failed");
      for (int k = 1; k < kMaxSets && rc == GH_OK; k++)
        if (hipStreamCreateWithFlags(&sc->aux[k], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&sc->ev_join[k], hipEventDisableTiming) != hipSuccess)
          rc = fail(GH_ERR_HIP, "hipStreamCreate failed");
      if (rc == GH_OK) {
        if (hipMalloc((void**)&sc->counters, sizeof(CountersDev) + 64) != hipSuccess ||
            hipMemset(sc->counters, 0, sizeof(CountersDev) + 64) != hipSuccess)
          rc = fail(GH_ERR_HIP, "counter allocation failed");
        else sc->small_done = reinterpret_cast<unsigned*>(sc->counters + 1);
      }
      if (rc == GH_OK && (rc = sc->hover.ensure(64)) == GH_OK) std::memset(sc->hover.p, 0, 64);
    }
  }
  if (rc != GH_OK) {
    std::string keep = g_last_error;
    gh_scene_destroy(sc);
    g_last_error = keep;
    return rc;
  }
  {
    std::lock_guard<std::mutex> lk(g_pool[device].mu);
    g_pool[device].scenes++;
    sc->pooled = true;
  }
  *out = sc;
  return GH_OK;
}

int gh_scene_create_multi(const gh_robot_desc* robot, const gh_obstacle_desc* obstacles, double margin, int precision,
                          const int* devices, int n_devices, gh_scene** out) {
  if (!out) return fail(GH_ERR_INVALID, "null argument");
  *out = nullptr;
  if (!devices || n_devices < 1 || n_devices > 64) return fail(GH_ERR_INVALID, "device list must hold 1..64 entries");
  gh_scene* grp = new (std::nothrow) gh_scene;
  if (!grp) return fail(GH_ERR_INVALID, "out of host memory");
  for (int k = 0; k < n_devices; k++) {
    gh_scene* part = nullptr;
    int rc = gh_scene_create(robot, obstacles, margin, precision, devices[k], &part);
    if (rc != GH_OK) {
      std::string keep = g_last_error;
      for (gh_scene* p : grp->parts) gh_scene_destroy(p);
      delete grp;
      g_last_error = keep;
      return rc;
    }
    grp->parts.push_back(part);
  }
  const gh_scene* p0 = grp->parts[0];
  grp->precision = p0->precision; grp->device = p0->device; grp->margin = p0->margin;
  grp->D = p0->D; grp->N = p0->N; grp->L = p0->L; grp->O = p0->O; grp->O_user = p0->O_user; grp->P = p0->P;
  grp->knobs = p0->knobs;
  *out = grp;
  return GH_OK;
}

int gh_scene_device_count(const gh_scene* sc) { return sc ? (sc->parts.empty() ? 1 : (int)sc->parts.size()) : GH_ERR_INVALID; }

void gh_scene_destroy(gh_scene* sc) {
  if (!sc) return;
  if (!sc->parts.empty()) {
    for (gh_scene* p : sc->parts) gh_scene_destroy(p);
    delete sc;
    return;
  }
  {
    Guard g(sc);
    if (sc->stream) { (void)hipStreamSynchronize(sc->stream); (void)hipStreamDestroy(sc->stream); }
    for (void* p : sc->tables) (void)hipFree(p);
    if (sc->counters) (void)hipFree(sc->counters);
    sc->q.release(); sc->flags.release(); sc->out.release(); sc->edges.release(); sc->store.release(); sc->first.release();
    sc->ends.release(); sc->ecnt.release(); sc->ebsum.release();
    sc->hq.release(); sc->hflags.release(); sc->hedges.release(); sc->hfirst.release(); sc->hseq.release(); sc->hover.release();
    for (int k = 1; k < kMaxSets; k++) {
      if (sc->aux[k]) { (void)hipStreamSynchronize(sc->aux[k]); (void)hipStreamDestroy(sc->aux[k]); }
      if (sc->ev_join[k]) (void)hipEventDestroy(sc->ev_join[k]);
    }
    if (sc->ev_fork) (void)hipEventDestroy(sc->ev_fork);
    for (int k = 0; k < 2; k++) {
      if (sc->ev_h2d[k]) (void)hipEventDestroy(sc->ev_h2d[k]);
      if (sc->ev_d2h[k]) (void)hipEventDestroy(sc->ev_d2h[k]);
    }
    if (sc->pooled) {  // the device's last scene takes the workspace pool with it
      PipePool& pool = g_pool[sc->device];
      std::lock_guard<std::mutex> lk(pool.mu);
      if (--pool.scenes == 0) {
        (void)hipDeviceSynchronize();
        for (auto& b : pool.sets) {
          b.hit.release(); b.xcnt.release(); b.xkey.release(); b.xpose_a.release(); b.xpose_b.release(); b.fq.release(); b.pc.release();
          if (b.last_use) (void)hipEventDestroy(b.last_use);
          b.last_use = nullptr;
          b.used = false;
        }
      }
    }
  }
  delete sc;
}

int gh_scene_dof(const gh_scene* sc) { return sc ? sc->D : GH_ERR_INVALID; }
int gh_scene_nodes(const gh_scene* sc) { return sc ? sc->N : GH_ERR_INVALID; }
int gh_scene_precision(const gh_scene* sc) { return sc ? sc->precision : GH_ERR_INVALID; }
int gh_scene_device(const gh_scene* sc) { return sc ? sc->device : GH_ERR_INVALID; }
int gh_link_mask_words(const gh_scene* sc) { return sc ? (sc->N + 63) / 64 : GH_ERR_INVALID; }

int gh_check_batch_dev(gh_scene* sc, const void* q_dev, size_t n, uint32_t what, uint8_t* flags_dev, void* stream) {
  if (sc && !sc->parts.empty()) return fail(GH_ERR_INVALID, "device-pointer entry points take a single-device scene: use the parts of a group one by one, or the host-buffer entry points");
  if (!sc || (n && (!q_dev || !flags_dev))) return fail(GH_ERR_INVALID, "null argument");
  if (n && sc->D == 0) return fail(GH_ERR_DOF_MISMATCH, "scene has no batch columns");
  Guard g(sc);
  if (!g.ok) return fail(GH_ERR_HIP, "hipSetDevice failed");
  return check_dev(sc, q_dev, n, what, flags_dev, stream ? (hipStream_t)stream : sc->stream, false);
}

int gh_check_batch_noalloc(gh_scene* sc, const void* q_dev, size_t n, uint32_t what, uint8_t* flags_dev, void* stream) {
  if (sc && !sc->parts.empty()) return fail(GH_ERR_INVALID, "device-pointer entry points take a single-device scene");
  if (!sc || (n && (!q_dev || !flags_dev))) return fail(GH_ERR_INVALID, "null argument");
  if (n && sc->D == 0) return fail(GH_ERR_DOF_MISMATCH, "scene has no batch columns");
  if (n > kNoAllocMaxConfigs)
    return fail(GH_ERR_TOO_LARGE, "gh_check_batch_noalloc takes at most " + std::to_string(kNoAllocMaxConfigs) +
                " configurations per call (" + std::to_string(n) + " given): split the batch, or use gh_check_batch_dev");
  Guard g(sc);
  if (!g.ok) return fail(GH_ERR_HIP, "hipSetDevice failed");
  // held from the wait on the previous call's event to the record of this call's: calls on one device are ordered
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  Carve region;
  region.base = (char*)g_scratch_ptr[sc->device];  // set when the scene was created on this device
  if (!region.base) return fail(GH_ERR_HIP, "no workspace address for this device");
  region.bytes = kScratchBytes;
  hipStream_t st = stream ? (hipStream_t)stream : sc->stream;
  if (g_scratch_used[sc->device]) GH_HIP(hipStreamWaitEvent(st, g_scratch_ev[sc->device], 0));
  const int rc = check_dev(sc, q_dev, n, what, flags_dev, st, false, &region);
  // (recorded after a failed enqueue too: whatever part of the call was launched still uses the array)
  GH_HIP(hipEventRecord(g_scratch_ev[sc->device], st));
  g_scratch_used[sc->device] = true;
  return rc;
}

int gh_check_batch(gh_scene* sc, const void* q, size_t n, uint32_t what, uint8_t* flags) {
  if (sc && !sc->parts.empty()) {
    if (n && (!q || !flags)) return fail(GH_ERR_INVALID, "null argument");
    // a batch that one device finishes in a single ~20 us launch is not worth a host thread per device
    if (n <= kZeroCopyMax) return gh_check_batch(sc->parts[0], q, n, what, flags);
    const size_t G = std::min<size_t>(sc->parts.size(), std::max<size_t>(n, 1));
    const size_t row = (size_t)sc->D * sc->elem();
    return run_parts(sc, G, [&](size_t k) {
      size_t lo, hi;
      shard_range(n, k, G, &lo, &hi);
      return gh_check_batch(sc->parts[k], (const char*)q + lo * row, hi - lo, what, flags + lo);
    });
  }
  if (!sc || (n && (!q || !flags))) return fail(GH_ERR_INVALID, "null argument");
  if (n == 0) return GH_OK;
  Guard g(sc);
  if (!g.ok) return fail(GH_ERR_HIP, "hipSetDevice failed");
  int rc;
  size_t qb = n * sc->D * sc->elem();
  if (n <= kZeroCopyMax && sc->knobs.zero_copy) {
    if ((rc = sc->hq.ensure(qb)) || (rc = sc->hflags.ensure(n)) || (rc = ensure_notify_word(sc))) return rc;
    std::memcpy(sc->hq.p, q, qb);
    uint32_t w = what;
#ifndef GEAR_HIP_ABLATE
    w &= kWhatPublic;
#endif
    if (sc->margin > 0 && small_takes(sc, n, w, false)) {  // one launch; its last workgroup tells the host when the flags are in
      SmallEdges ex;
      ex.notify = (unsigned*)sc->hseq.d;
      ex.seq = ++sc->seq;
      rc = sc->precision == 32 ? small_dev<float>(sc, sc->s32, (const float*)sc->hq.d, n, w, (uint8_t*)sc->hflags.d, sc->stream, &ex)
                               : small_dev<double>(sc, sc->s64, (const double*)sc->hq.d, n, w, (uint8_t*)sc->hflags.d, sc->stream, &ex);
      if (rc || (rc = wait_notified(sc, ex.seq))) return rc;
    } else {
      if ((rc = check_dev(sc, sc->hq.d, n, what, (uint8_t*)sc->hflags.d, sc->stream, false))) return rc;
      GH_HIP(hipStreamSynchronize(sc->stream));
    }
    std::memcpy(flags, sc->hflags.p, n);
    return GH_OK;
  }
  if ((rc = sc->q.ensure(qb)) || (rc = sc->flags.ensure(n)) || (rc = sc->hflags.ensure(n))) return rc;
  if (n <= kStagedMax) {
    if ((rc = sc->hq.ensure(qb))) return rc;
    std::memcpy(sc->hq.p, q, qb);
    GH_HIP(hipMemcpyAsync(sc->q.p, sc->hq.p, qb, hipMemcpyHostToDevice, sc->stream));
    if ((rc = check_dev(sc, sc->q.p, n, what, (uint8_t*)sc->flags.p, sc->stream, false))) return rc;
    GH_HIP(hipMemcpyAsync(sc->hflags.p, sc->flags.p, n, hipMemcpyDeviceToHost, sc->stream));
    GH_HIP(hipStreamSynchronize(sc->stream));
    std::memcpy(flags, sc->hflags.p, n);
    return GH_OK;
  }
  return host_pieces(sc, q, n, what, flags, [] { return (int)GH_OK; });
}

int gh_check_batch_counted(gh_scene* sc, const void* q_dev, size_t n, uint32_t what, uint8_t* flags_dev,
                           gh_counters* out) {
  if (sc && !sc->parts.empty()) return fail(GH_ERR_INVALID, "device-pointer entry points take a single-device scene");
  if (!sc || !out || (n && (!q_dev || !flags_dev))) return fail(GH_ERR_INVALID, "null argument");
  Guard g(sc);
  if (!g.ok) return fail(GH_ERR_HIP, "hipSetDevice failed");
  GH_HIP(hipMemsetAsync(sc->counters, 0, sizeof(CountersDev), sc->stream));
  int rc = check_dev(sc, q_dev, n, what, flags_dev, sc->stream, true);
  if (rc) return rc;
  CountersDev h;
  GH_HIP(hipMemcpyAsync(&h, sc->counters, sizeof(h), hipMemcpyDeviceToHost, sc->stream));
  GH_HIP(hipStreamSynchronize(sc->stream));
  std::memset(out, 0, sizeof(*out));
  out->configs = h.configs; out->pairs = h.pairs; out->cull_pass = h.cull_pass;
  out->sat_tests = h.sat_tests; out->exact_tests = h.exact_tests; out->exact_iters = h.exact_iters;
  return GH_OK;
}

int gh_pipeline_last_stats(gh_scene* sc, uint64_t out[4]) {
  if (sc && !sc->parts.empty()) sc = sc->parts[0];
  if (!sc || !out) return fail(GH_ERR_INVALID, "null argument");
  Guard g(sc);
  if (!g.ok) return fail(GH_ERR_HIP, "hipSetDevice failed");
  out[0] = out[1] = out[2] = out[3] = 0;
  PipePool& pool = g_pool[sc->device];
  std::lock_guard<std::mutex> lk(pool.mu);
  if (!pool.sets[0].pc.p) return GH_OK;
  GH_HIP(hipDeviceSynchronize());
  for (int k = 0; k < sc->last_sets; k++) {  // the last chunk of each workspace set the last call used
    PipeBufs& b = pool.sets[k];
    if (!b.pc.p) continue;
    PipeCounters h;
    GH_HIP(hipMemcpy(&h, b.pc.p, sizeof(h), hipMemcpyDeviceToHost));
    out[0] += h.candidates;
    out[1] += h.exact;
    out[2] += h.redo_waves;
  }
  out[3] = sc->exact_cap;
  return GH_OK;
}

// flags of every sample + first flagged sample per edge; the caller holds the scene's Guard
static int check_edges_locked(gh_scene* sc, const void* q_dev, const uint32_t* edge_off_dev, size_t n_edges,
                              size_t n_samples, uint32_t what, uint8_t bad_mask, uint8_t* flags_dev,
                              int32_t* first_bad_dev, hipStream_t st, bool exclusive = false, uint8_t* flags_near = nullptr,
                              unsigned* notify = nullptr, unsigned seq = 0, bool* notified = nullptr) {
#ifndef GEAR_HIP_ABLATE
  what &= kWhatPublic;
#endif
  // exclusive: the caller keeps the scene's lock until the stream has drained (the host-buffer entry point),
  // so the one ticket counter of the scene serves this call alone: one launch does everything
  if (exclusive && n_edges > 0 && n_edges <= kSmallEdgesMax && sc->margin > 0 && small_takes(sc, n_samples, what, false)) {
    SmallEdges ed;
    ed.edge_off = edge_off_dev; ed.n_edges = n_edges; ed.bad_mask = bad_mask; ed.first_bad = first_bad_dev; ed.flags_near = flags_near;
    ed.notify = notify; ed.seq = seq;
    if (notified) *notified = notify != nullptr;
    return sc->precision == 32 ? small_dev<float>(sc, sc->s32, (const float*)q_dev, n_samples, what, flags_dev, st, &ed)
                               : small_dev<double>(sc, sc->s64, (const double*)q_dev, n_samples, what, flags_dev, st, &ed);
  }
  int rc = check_dev(sc, q_dev, n_samples, what, flags_dev, st, false);
  if (rc) return rc;
  if (n_edges) {
    hipLaunchKernelGGL(k_edge_first, dim3((unsigned)((n_edges + 255) / 256)), dim3(256), 0, st, flags_dev, edge_off_dev,
                       n_edges, bad_mask, first_bad_dev);
    GH_HIP(hipGetLastError());
  }
  return GH_OK;
}

int gh_check_edges_dev(gh_scene* sc, const void* q_dev, const uint32_t* edge_off_dev, size_t n_edges, size_t n_samples,
                       uint32_t what, uint8_t bad_mask, uint8_t* flags_dev, int32_t* first_bad_dev, void* stream) {
  if (sc && !sc->parts.empty()) return fail(GH_ERR_INVALID, "device-pointer entry points take a single-device scene");
  if (!sc || (n_edges && (!edge_off_dev || !first_bad_dev)) || (n_samples && (!q_dev || !flags_dev)))
    return fail(GH_ERR_INVALID, "null argument");
  Guard g(sc);
  if (!g.ok) return fail(GH_ERR_HIP, "hipSetDevice failed");
  return check_edges_locked(sc, q_dev, edge_off_dev, n_edges, n_samples, what, bad_mask, flags_dev, first_bad_dev,
                            stream ? (hipStream_t)stream : sc->stream);
}

int gh_check_edges(gh_scene* sc, const void* q, const uint32_t* edge_off, size_t n_edges, uint32_t what,
                   uint8_t bad_mask, uint8_t* flags, int32_t* first_bad) {
  if (sc && !sc->parts.empty()) {
    if (!edge_off || (n_edges && !first_bad)) return fail(GH_ERR_INVALID, "null argument");
    if (n_edges == 0) return GH_OK;
    if (edge_off[0] != 0) return fail(GH_ERR_INVALID, "edge_off[0] must be 0");
    for (size_t e = 0; e < n_edges; e++)
      if (edge_off[e + 1] < edge_off[e]) return fail(GH_ERR_INVALID, "edge_off must be non-decreasing");
    const size_t n = edge_off[n_edges];
    if (n && (!q || !flags)) return fail(GH_ERR_INVALID, "null argument");
    if (n <= kZeroCopyMax && n_edges <= kZeroCopyMax)  // an RRT line or a few: one device, one launch, no threads
      return gh_check_edges(sc->parts[0], q, edge_off, n_edges, what, bad_mask, flags, first_bad);
    // contiguous shares of whole edges, balanced by samples: share k starts at the first edge whose
    // first sample is at or beyond n * k / G
    const size_t G = std::min<size_t>(sc->parts.size(), n_edges);
    const std::vector<size_t> cut = shard_edges(edge_off, n_edges, G);
    const size_t row = (size_t)sc->D * sc->elem();
    return run_parts(sc, G, [&](size_t k) -> int {
      const size_t e0 = cut[k], e1 = cut[k + 1];
      if (e0 == e1) return GH_OK;
      std::vector<uint32_t> off(e1 - e0 + 1);
      for (size_t e = e0; e <= e1; e++) off[e - e0] = edge_off[e] - edge_off[e0];
      return gh_check_edges(sc->parts[k], (const char*)q + (size_t)edge_off[e0] * row, off.data(), e1 - e0, what, bad_mask,
                            flags + edge_off[e0], first_bad + e0);
    });
  }
  if (!sc || !edge_off || (n_edges && !first_bad)) return fail(GH_ERR_INVALID, "null argument");
  if (n_edges == 0) return GH_OK;
  for (size_t e = 0; e < n_edges; e++)
    if (edge_off[e + 1] < edge_off[e]) return fail(GH_ERR_INVALID, "edge_off must be non-decreasing");
  if (edge_off[0] != 0) return fail(GH_ERR_INVALID, "edge_off[0] must be 0");
  size_t n = edge_off[n_edges];
  if (n && (!q || !flags)) return fail(GH_ERR_INVALID, "null argument");
  int rc;
  // ONE guard from staging to read-back on either path: the staging buffers, the workspace and the
  // stream belong to the scene, and another host thread may be calling into the same scene
  Guard g(sc);
  if (!g.ok) return fail(GH_ERR_HIP, "hipSetDevice failed");
  const size_t qb = n * sc->D * sc->elem();
  if (n <= kZeroCopyMax && n_edges <= kZeroCopyMax && sc->knobs.zero_copy) {
    if ((rc = sc->hq.ensure(std::max<size_t>(qb, 1))) || (rc = sc->hflags.ensure(std::max<size_t>(n, 1))) ||
        (rc = sc->hedges.ensure((n_edges + 1) * 4)) || (rc = sc->hfirst.ensure(n_edges * 4)))
      return rc;
    if (qb) std::memcpy(sc->hq.p, q, qb);
    std::memcpy(sc->hedges.p, edge_off, (n_edges + 1) * 4);
    if ((rc = sc->flags.ensure(kZeroCopyMax)) || (rc = ensure_notify_word(sc))) return rc;
    bool notified = false;
    const unsigned seq = ++sc->seq;
    if ((rc = check_edges_locked(sc, sc->hq.d, (const uint32_t*)sc->hedges.d, n_edges, n, what, bad_mask,
                                 (uint8_t*)sc->hflags.d, (int32_t*)sc->hfirst.d, sc->stream, true, (uint8_t*)sc->flags.p,
                                 (unsigned*)sc->hseq.d, seq, &notified)))
      return rc;
    if (notified) { if ((rc = wait_notified(sc, seq))) return rc; }
    else GH_HIP(hipStreamSynchronize(sc->stream));
    if (n) std::memcpy(flags, sc->hflags.p, n);
    std::memcpy(first_bad, sc->hfirst.p, n_edges * 4);
    return GH_OK;
  }
  if ((rc = sc->q.ensure(std::max<size_t>(qb, 1))) || (rc = sc->flags.ensure(std::max<size_t>(n, 1))) ||
      (rc = sc->edges.ensure((n_edges + 1) * 4)) || (rc = sc->first.ensure(n_edges * 4)))
    return rc;
  // mid-size calls go through the scene's page-locked staging: four copies out of / into pageable memory cost more
  // than the memcpys (kStagedMax)
  const bool staged = n <= kStagedMax && n_edges <= kStagedMax;
  const void* q_src = q;
  const uint32_t* off_src = edge_off;
  uint8_t* flags_dst = flags;
  int32_t* first_dst = first_bad;
  if (staged) {
    if ((rc = sc->hq.ensure(std::max<size_t>(qb, 1))) || (rc = sc->hflags.ensure(std::max<size_t>(n, 1))) ||
        (rc = sc->hedges.ensure((n_edges + 1) * 4)) || (rc = sc->hfirst.ensure(n_edges * 4)))
      return rc;
    if (qb) std::memcpy(sc->hq.p, q, qb);
    std::memcpy(sc->hedges.p, edge_off, (n_edges + 1) * 4);
    q_src = sc->hq.p;
    off_src = (const uint32_t*)sc->hedges.p;
    flags_dst = (uint8_t*)sc->hflags.p;
    first_dst = (int32_t*)sc->hfirst.p;
  }
  if (!staged && n > kHostPiece && n_edges > kSmallEdgesMax) {
    // a large batch of edges: the samples move and are evaluated piece by piece, the first-bad reduction follows
    if ((rc = sc->hflags.ensure(n))) return rc;
    GH_HIP(hipMemcpyAsync(sc->edges.p, edge_off, (n_edges + 1) * 4, hipMemcpyHostToDevice, sc->stream));
    uint32_t w = what;
#ifndef GEAR_HIP_ABLATE
    w &= kWhatPublic;
#endif
    return host_pieces(sc, q, n, w, flags, [&]() -> int {
      hipLaunchKernelGGL(k_edge_first, dim3((unsigned)((n_edges + 255) / 256)), dim3(256), 0, sc->stream, (uint8_t*)sc->flags.p,
                         (const uint32_t*)sc->edges.p, n_edges, bad_mask, (int32_t*)sc->first.p);
      GH_HIP(hipGetLastError());
      GH_HIP(hipMemcpyAsync(first_bad, sc->first.p, n_edges * 4, hipMemcpyDeviceToHost, sc->stream));
      return GH_OK;
    });
  }
  if (qb) GH_HIP(hipMemcpyAsync(sc->q.p, q_src, qb, hipMemcpyHostToDevice, sc->stream));
  GH_HIP(hipMemcpyAsync(sc->edges.p, off_src, (n_edges + 1) * 4, hipMemcpyHostToDevice, sc->stream));
  if ((rc = check_edges_locked(sc, sc->q.p, (const uint32_t*)sc->edges.p, n_edges, n, what, bad_mask,
                               (uint8_t*)sc->flags.p, (int32_t*)sc->first.p, sc->stream, true)))
    return rc;
  if (n) GH_HIP(hipMemcpyAsync(flags_dst, sc->flags.p, n, hipMemcpyDeviceToHost, sc->stream));
  GH_HIP(hipMemcpyAsync(first_dst, sc->first.p, n_edges * 4, hipMemcpyDeviceToHost, sc->stream));
  GH_HIP(hipStreamSynchronize(sc->stream));
  if (staged) {
    if (n) std::memcpy(flags, sc->hflags.p, n);
    std::memcpy(first_bad, sc->hfirst.p, n_edges * 4);
  }
  return GH_OK;
}

// ---- edges from their end points (gear_edges.hpp) ---------------------------------------------------
// Enqueues on `st`: samples per edge -> cnt_dev, their exclusive prefix sum -> off_dev[0..n_edges] (off_dev[n_edges] =
// total).  The caller holds the scene's Guard.
static int edges_count_scan(gh_scene* sc, const void* from_dev, const void* to_dev, size_t n_edges, double step,
                            uint32_t max_len, uint32_t mode, uint32_t* cnt_dev, uint32_t* off_dev, hipStream_t st) {
  const size_t n_blocks = (n_edges + 1023) / 1024;
  int rc = sc->ebsum.ensure((n_blocks + 1) * 4);
  if (rc) return rc;
  const dim3 g256((unsigned)((n_edges + 255) / 256));
  // (the number of columns as a template argument up to 8: the walker stays in registers)
  auto go = [&](auto dd) {
    constexpr int DD = decltype(dd)::value;
    if (sc->precision == 32)
      hipLaunchKernelGGL((k_edge_count<float, DD>), g256, dim3(256), 0, st, (const float*)from_dev, (const float*)to_dev, n_edges, sc->D,
                         (float)step, max_len, mode, cnt_dev);
    else
      hipLaunchKernelGGL((k_edge_count<double, DD>), g256, dim3(256), 0, st, (const double*)from_dev, (const double*)to_dev, n_edges, sc->D,
                         step, max_len, mode, cnt_dev);
  };
  by_columns(sc->D, go);
  hipLaunchKernelGGL(k_edge_scan1, dim3((unsigned)n_blocks), dim3(1024), 0, st, cnt_dev, n_edges, off_dev, (uint32_t*)sc->ebsum.p);
  hipLaunchKernelGGL(k_edge_scan2, dim3(1), dim3(1024), 0, st, (uint32_t*)sc->ebsum.p, n_blocks);
  hipLaunchKernelGGL(k_edge_scan3, dim3((unsigned)n_blocks), dim3(1024), 0, st, off_dev, n_edges, (const uint32_t*)sc->ebsum.p, n_blocks);
  GH_HIP(hipGetLastError());
  return GH_OK;
}

static void edges_fill(gh_scene* sc, const void* from_dev, const void* to_dev, size_t n_edges, double step, uint32_t max_len,
                       uint32_t mode, const uint32_t* off_dev, void* q_dev, hipStream_t st) {
  const dim3 g256((unsigned)((n_edges + 255) / 256));
  auto go = [&](auto dd) {
    constexpr int DD = decltype(dd)::value;
    if (sc->precision == 32)
      hipLaunchKernelGGL((k_edge_fill<float, DD>), g256, dim3(256), 0, st, (const float*)from_dev, (const float*)to_dev, n_edges, sc->D,
                         (float)step, max_len, mode, off_dev, (float*)q_dev);
    else
      hipLaunchKernelGGL((k_edge_fill<double, DD>), g256, dim3(256), 0, st, (const double*)from_dev, (const double*)to_dev, n_edges, sc->D,
                         step, max_len, mode, off_dev, (double*)q_dev);
  };
  by_columns(sc->D, go);
}

static int edges_args_ok(const gh_scene* sc, double step, uint32_t max_len, uint32_t mode) {
  if (!(step > 0)) return fail(GH_ERR_INVALID, "step must be positive");
  if (max_len == 0 || max_len > (1u << 20)) return fail(GH_ERR_INVALID, "max_len must be in 1..1048576");
  if (mode != GH_EDGE_CONNECT && mode != GH_EDGE_WALK) return fail(GH_ERR_INVALID, "mode must be GH_EDGE_CONNECT or GH_EDGE_WALK");
  if (sc->D > kEdgeMaxDof) return fail(GH_ERR_TOO_LARGE, "edges from end points support at most 64 batch columns");
  if (sc->D == 0) return fail(GH_ERR_DOF_MISMATCH, "scene has no batch columns");
  return GH_OK;
}

int gh_check_edges_from_endpoints_dev(gh_scene* sc, const void* from_dev, const void* to_dev, size_t n_edges, double step,
                                      uint32_t max_len, uint32_t mode, uint32_t what, uint8_t bad_mask, void* q_dev,
                                      size_t cap_samples, uint32_t* edge_off_dev, uint32_t* n_samples_dev, uint8_t* flags_dev,
                                      int32_t* first_bad_dev, size_t* total_samples, void* stream) {
  if (sc && !sc->parts.empty()) return fail(GH_ERR_INVALID, "device-pointer entry points take a single-device scene");
  if (!sc || !total_samples || (n_edges && (!from_dev || !to_dev || !edge_off_dev || !n_samples_dev || !first_bad_dev)))
    return fail(GH_ERR_INVALID, "null argument");
  *total_samples = 0;
  if (n_edges == 0) return GH_OK;
  if (n_edges >= (size_t(1) << 31)) return fail(GH_ERR_TOO_LARGE, "too many edges");
  int rc = edges_args_ok(sc, step, max_len, mode);
  if (rc) return rc;
  Guard g(sc);
  if (!g.ok) return fail(GH_ERR_HIP, "hipSetDevice failed");
  hipStream_t st = stream ? (hipStream_t)stream : sc->stream;
  if ((rc = edges_count_scan(sc, from_dev, to_dev, n_edges, step, max_len, mode, n_samples_dev, edge_off_dev, st))) return rc;
  // the one synchronisation of this call: the number of samples decides buffer use and which kernels run
  uint32_t total = 0;
  GH_HIP(hipMemcpyAsync(&total, edge_off_dev + n_edges, 4, hipMemcpyDeviceToHost, st));
  GH_HIP(hipStreamSynchronize(st));
  *total_samples = total;
  if (total > cap_samples)
    return fail(GH_ERR_TOO_LARGE, "the edges have " + std::to_string(total) + " samples; q_dev / flags_dev hold " + std::to_string(cap_samples));
  if (total && (!q_dev || !flags_dev)) return fail(GH_ERR_INVALID, "null argument");
  edges_fill(sc, from_dev, to_dev, n_edges, step, max_len, mode, edge_off_dev, q_dev, st);
  GH_HIP(hipGetLastError());
  return check_edges_locked(sc, q_dev, edge_off_dev, n_edges, total, what, bad_mask, flags_dev, first_bad_dev, st);
}

int gh_check_edges_from_endpoints(gh_scene* sc, const void* from, const void* to, size_t n_edges, double step, uint32_t max_len,
                                  uint32_t mode, uint32_t what, uint8_t bad_mask, int32_t* first_bad, uint32_t* n_samples) {
  if (sc && !sc->parts.empty()) {
    if (n_edges && (!from || !to || !first_bad || !n_samples)) return fail(GH_ERR_INVALID, "null argument");
    if (n_edges <= kZeroCopyMax) return gh_check_edges_from_endpoints(sc->parts[0], from, to, n_edges, step, max_len, mode, what, bad_mask, first_bad, n_samples);
    const size_t G = std::min<size_t>(sc->parts.size(), n_edges);  // equal shares of edges (their lengths are not known yet)
    const size_t row = (size_t)sc->D * sc->elem();
    return run_parts(sc, G, [&](size_t k) {
      size_t lo, hi;
      shard_range(n_edges, k, G, &lo, &hi);
      return gh_check_edges_from_endpoints(sc->parts[k], (const char*)from + lo * row, (const char*)to + lo * row, hi - lo, step,
                                           max_len, mode, what, bad_mask, first_bad + lo, n_samples + lo);
    });
  }
  if (!sc || (n_edges && (!from || !to || !first_bad || !n_samples))) return fail(GH_ERR_INVALID, "null argument");
  if (n_edges == 0) return GH_OK;
  if (n_edges >= (size_t(1) << 31)) return fail(GH_ERR_TOO_LARGE, "too many edges");
  int rc = edges_args_ok(sc, step, max_len, mode);
  if (rc) return rc;
  Guard g(sc);
  if (!g.ok) return fail(GH_ERR_HIP, "hipSetDevice failed");
  const size_t eb = n_edges * (size_t)sc->D * sc->elem();
  if ((rc = sc->ends.ensure(2 * eb)) || (rc = sc->ecnt.ensure(n_edges * 4)) || (rc = sc->edges.ensure((n_edges + 1) * 4)) ||
      (rc = sc->first.ensure(n_edges * 4)))
    return rc;
  char* from_dev = (char*)sc->ends.p;
  char* to_dev = from_dev + eb;
  GH_HIP(hipMemcpyAsync(from_dev, from, eb, hipMemcpyHostToDevice, sc->stream));
  GH_HIP(hipMemcpyAsync(to_dev, to, eb, hipMemcpyHostToDevice, sc->stream));
  if ((rc = edges_count_scan(sc, from_dev, to_dev, n_edges, step, max_len, mode, (uint32_t*)sc->ecnt.p, (uint32_t*)sc->edges.p, sc->stream)))
    return rc;
  uint32_t total = 0;
  GH_HIP(hipMemcpyAsync(&total, (uint32_t*)sc->edges.p + n_edges, 4, hipMemcpyDeviceToHost, sc->stream));
  GH_HIP(hipMemcpyAsync(n_samples, sc->ecnt.p, n_edges * 4, hipMemcpyDeviceToHost, sc->stream));
  GH_HIP(hipStreamSynchronize(sc->stream));
  if ((rc = sc->q.ensure(std::max<size_t>((size_t)total * sc->D * sc->elem(), 1))) || (rc = sc->flags.ensure(std::max<size_t>(total, 1))))
    return rc;
  edges_fill(sc, from_dev, to_dev, n_edges, step, max_len, mode, (const uint32_t*)sc->edges.p, sc->q.p, sc->stream);
  GH_HIP(hipGetLastError());
  if ((rc = check_edges_locked(sc, sc->q.p, (const uint32_t*)sc->edges.p, n_edges, total, what, bad_mask, (uint8_t*)sc->flags.p,
                               (int32_t*)sc->first.p, sc->stream, true)))
    return rc;
  GH_HIP(hipMemcpyAsync(first_bad, sc->first.p, n_edges * 4, hipMemcpyDeviceToHost, sc->stream));
  GH_HIP(hipStreamSynchronize(sc->stream));
  return GH_OK;
}

int gh_fk_batch_dev(gh_scene* sc, const void* q_dev, size_t n, void* poses_dev, void* stream) {
  if (sc && !sc->parts.empty()) return fail(GH_ERR_INVALID, "device-pointer entry points take a single-device scene");
  if (!sc || (n && (!q_dev || !poses_dev))) return fail(GH_ERR_INVALID, "null argument");
  Guard g(sc);
  if (!g.ok) return fail(GH_ERR_HIP, "hipSetDevice failed");
  return fk_dev(sc, q_dev, n, poses_dev, stream ? (hipStream_t)stream : sc->stream);
}

int gh_fk_batch(gh_scene* sc, const void* q, size_t n, void* poses) {
  if (sc && !sc->parts.empty()) {
    if (n && (!q || !poses)) return fail(GH_ERR_INVALID, "null argument");
    if (n <= kZeroCopyMax) return gh_fk_batch(sc->parts[0], q, n, poses);
    const size_t G = std::min<size_t>(sc->parts.size(), std::max<size_t>(n, 1));
    const size_t row = (size_t)sc->D * sc->elem(), prow = (size_t)sc->N * 7 * sc->elem();
    return run_parts(sc, G, [&](size_t k) {
      size_t lo, hi;
      shard_range(n, k, G, &lo, &hi);
      return gh_fk_batch(sc->parts[k], (const char*)q + lo * row, hi - lo, (char*)poses + lo * prow);
    });
  }
  if (!sc || (n && (!q || !poses))) return fail(GH_ERR_INVALID, "null argument");
  if (n == 0) return GH_OK;
  Guard g(sc);
  if (!g.ok) return fail(GH_ERR_HIP, "hipSetDevice failed");
  int rc;
  size_t qb = n * sc->D * sc->elem(), pb = n * sc->N * 7 * sc->elem();
  if ((rc = sc->q.ensure(std::max<size_t>(qb, 1))) || (rc = sc->out.ensure(pb))) return rc;
  if (qb) GH_HIP(hipMemcpyAsync(sc->q.p, q, qb, hipMemcpyHostToDevice, sc->stream));
  if ((rc = fk_dev(sc, sc->q.p, n, sc->out.p, sc->stream))) return rc;
  GH_HIP(hipMemcpyAsync(poses, sc->out.p, pb, hipMemcpyDeviceToHost, sc->stream));
  GH_HIP(hipStreamSynchronize(sc->stream));
  return GH_OK;
}

int gh_link_hits(gh_scene* sc, const void* q, size_t n, uint32_t what, uint64_t* link_mask) {
  if (sc && !sc->parts.empty()) sc = sc->parts[0];  // a diagnostic: one device serves it
  if (!sc || (n && (!q || !link_mask))) return fail(GH_ERR_INVALID, "null argument");
  if (n == 0 || sc->O == 0) return GH_OK;
  if (sc->N > 128) return fail(GH_ERR_TOO_LARGE, "gh_link_hits supports at most 128 nodes");
  Guard g(sc);
  if (!g.ok) return fail(GH_ERR_HIP, "hipSetDevice failed");
  int W = (sc->N + 63) / 64, rc;
  size_t qb = n * sc->D * sc->elem(), mb = n * sc->O_user * W * 8;
  if ((rc = sc->q.ensure(std::max<size_t>(qb, 1))) || (rc = sc->out.ensure(mb))) return rc;
  if (qb) GH_HIP(hipMemcpyAsync(sc->q.p, q, qb, hipMemcpyHostToDevice, sc->stream));
  if (!(sc->margin > 0)) {
    GH_HIP(hipMemsetAsync(sc->out.p, 0, mb, sc->stream));
  } else {
    StorePlan plan = plan_store(sc, true);
    rc = for_chunks(sc, n, plan, [&](size_t off, size_t cnt, void* gs, size_t gstride) {
      unsigned long long* m = (unsigned long long*)sc->out.p + off * sc->O_user * W;
      if (sc->precision == 32)
        hipLaunchKernelGGL((k_link_hits<float>), dim3(grid_for(cnt)), dim3(kWave), plan.lds_bytes, sc->stream, sc->s32,
                           (const float*)sc->q.p + off * sc->D, cnt, what & kWhatPublic, m, W, (float*)gs, gstride);
      else
        hipLaunchKernelGGL((k_link_hits<double>), dim3(grid_for(cnt)), dim3(kWave), plan.lds_bytes, sc->stream, sc->s64,
                           (const double*)sc->q.p + off * sc->D, cnt, what & kWhatPublic, m, W, (double*)gs, gstride);
    });
    if (rc) return rc;
  }
  GH_HIP(hipMemcpyAsync(link_mask, sc->out.p, mb, hipMemcpyDeviceToHost, sc->stream));
  GH_HIP(hipStreamSynchronize(sc->stream));
  return GH_OK;
}

int gh_self_hits(gh_scene* sc, const void* q, size_t n, uint32_t what, uint64_t* pair_mask) {
  if (sc && !sc->parts.empty()) sc = sc->parts[0];
  if (!sc || (n && (!q || !pair_mask))) return fail(GH_ERR_INVALID, "null argument");
  if (n == 0 || sc->P == 0) return GH_OK;
  Guard g(sc);
  if (!g.ok) return fail(GH_ERR_HIP, "hipSetDevice failed");
  int W = (sc->P + 63) / 64, rc;
  size_t qb = n * sc->D * sc->elem(), mb = n * W * 8;
  if ((rc = sc->q.ensure(std::max<size_t>(qb, 1))) || (rc = sc->out.ensure(mb))) return rc;
  if (qb) GH_HIP(hipMemcpyAsync(sc->q.p, q, qb, hipMemcpyHostToDevice, sc->stream));
  if (!(sc->margin > 0)) {
    GH_HIP(hipMemsetAsync(sc->out.p, 0, mb, sc->stream));
  } else {
    StorePlan plan = plan_store(sc, true);
    rc = for_chunks(sc, n, plan, [&](size_t off, size_t cnt, void* gs, size_t gstride) {
      unsigned long long* m = (unsigned long long*)sc->out.p + off * W;
      if (sc->precision == 32)
        hipLaunchKernelGGL((k_self_hits<float>), dim3(grid_for(cnt)), dim3(kWave), plan.lds_bytes, sc->stream, sc->s32,
                           (const float*)sc->q.p + off * sc->D, cnt, what & kWhatPublic, m, W, (float*)gs, gstride);
      else
        hipLaunchKernelGGL((k_self_hits<double>), dim3(grid_for(cnt)), dim3(kWave), plan.lds_bytes, sc->stream, sc->s64,
                           (const double*)sc->q.p + off * sc->D, cnt, what & kWhatPublic, m, W, (double*)gs, gstride);
    });
    if (rc) return rc;
  }
  GH_HIP(hipMemcpyAsync(pair_mask, sc->out.p, mb, hipMemcpyDeviceToHost, sc->stream));
  GH_HIP(hipStreamSynchronize(sc->stream));
  return GH_OK;
}

}  // extern "C"
