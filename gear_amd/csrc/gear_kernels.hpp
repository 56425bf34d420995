// gear_kernels.hpp — gfx950 kernels of the batched collision-validation path.
//
// One lane = one joint configuration.  A workgroup is ONE wavefront (64 lanes):
// lanes never exchange data, so there is no barrier anywhere, and the per-lane
// pose columns in LDS are private to their lane.
//
//   k_check      is_feasible / is_feasible_with_self over a batch
//                (src/planner/joint_path_planner.rs:71-98,113-128)
//   k_fk         k::Chain::update_transforms over a batch
//   k_link_hits  colliding_link_names            (joint_path_planner.rs:100-110)
//   k_self_hits  colliding_link_names_with_self  (joint_path_planner.rs:130-134)
//   k_edge_first first infeasible sample per edge (rrt stepping)
//
// Scene tables are read through the constant address space: every index into
// them is wave-uniform, so the reads become scalar loads (s_load_*) and the
// obstacle data reaches the VALU as SGPR operands instead of through LDS.
#pragma once
#include "gear_device.hpp"

namespace gear_hip {

#define GEAR_CAS __attribute__((address_space(4)))
template <class U> GD const GEAR_CAS U* to_cas(const U* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  return (const GEAR_CAS U*)p;
#pragma clang diagnostic pop
}

constexpr int kWave = 64;

// `what` bits a caller may pass (include/gear_hip.h): GH_CHECK_ENV | GH_CHECK_SELF | GH_CHECK_NO_LIMITS and
// the GH_PATH_* / GH_COUNT_CANDIDATES evaluation hints.  Every extern "C" entry point masks `what` to
// these, so a stray high bit can never change a result.
constexpr uint32_t kWhatPublic = 0x3u | 0x10u | 0x20u | 0xf000u;
constexpr uint32_t kWhatNoLimits = 0x10u;
// Profiling ablations (skip a stage, force wave-uniform records, ...) give WRONG flags on purpose.  They
// exist only in the -DGEAR_HIP_ABLATE build that tools/ablate_*.py load (gear_amd/build.py build_ablate);
// in the product build the tests are compile-time false and the code behind them is dead.
#ifdef GEAR_HIP_ABLATE
#define GEAR_DBG(what, bits) (((what) & (bits)) != 0)
#else
#define GEAR_DBG(what, bits) false
#endif
constexpr unsigned kHitRedo = 0x80u;  // pipeline hit word, bit 7: re-check this configuration with the fused kernel
constexpr unsigned kXRedo = 0xffffffffu;  // PipeWs::xcnt of a sweep wavefront whose exact region ran full: same request, per wavefront

template <class T> struct Views {
  const GEAR_CAS NodeDev<T>* nodes;
  const GEAR_CAS LimitDev<T>* limits;
  const GEAR_CAS ShapeDev<T>* rshapes;
  const GEAR_CAS ShapeDev<T>* oshapes;
  const GEAR_CAS T* ocull;
  const GEAR_CAS T* ocullq;
  const GEAR_CAS PairDev* pairs;
  const GEAR_CAS int32_t* spairs;
  const GEAR_CAS T* spair_r;
  const GEAR_CAS T* verts;
  const GEAR_CAS T* planes;
  __device__ explicit Views(const SceneDev<T>& S)
      : nodes(to_cas(S.nodes)), limits(to_cas(S.limits)), rshapes(to_cas(S.rshapes)), oshapes(to_cas(S.oshapes)),
        ocull(to_cas(S.ocull)), ocullq(to_cas(S.ocullq)), pairs(to_cas(S.pairs)), spairs(to_cas(S.spairs)),
        spair_r(to_cas(S.spair_r)), verts(to_cas(S.verts)), planes(to_cas(S.planes)) {}
};

template <class T, class SP> GD ShapeRef<T> shape_ref(SP sh) {
  ShapeRef<T> r;
  r.kind = sh->kind;
  r.half = {sh->half[0], sh->half[1], sh->half[2]};
  r.br = sh->br;
  r.vb = sh->v_begin; r.vc = sh->v_count;
  r.fb = sh->f_begin; r.fc = sh->f_count;
  r.pn = sh->prism_n;
  return r;
}
template <class T, class SP> GD Pose<T> shape_pose(SP sh) {
  return {{sh->lt[0], sh->lt[1], sh->lt[2]}, {sh->lq[0], sh->lq[1], sh->lq[2], sh->lq[3]}};
}

// Per-lane pose columns: element (slot, component) of this lane lives at
// base[(slot*7 + component) * stride + idx]  — LDS (stride 64, idx lane) or, for
// scenes too large for LDS, a global workspace (stride = padded n, idx = config).
template <class T> struct PoseStore {
  T* base;
  size_t stride, idx;
  GD void put(int slot, const Pose<T>& p) const {
    T* b = base + (size_t)slot * 7 * stride + idx;
    b[0] = p.t.x; b[stride] = p.t.y; b[2 * stride] = p.t.z;
    b[3 * stride] = p.r.i; b[4 * stride] = p.r.j; b[5 * stride] = p.r.k; b[6 * stride] = p.r.w;
  }
  GD Pose<T> get(int slot) const {
    const T* b = base + (size_t)slot * 7 * stride + idx;
    return {{b[0], b[stride], b[2 * stride]}, {b[3 * stride], b[4 * stride], b[5 * stride], b[6 * stride]}};
  }
};

#pragma clang fp contract(off)
// Mimic::mimic_position: offset + multiplier * parent_position, un-contracted
template <class T> GD T joint_value(T qv, int direct, T mult, T off) { return direct ? qv : off + mult * qv; }
#pragma clang fp contract(fast)

// k::Chain::set_joint_positions: true when every Range::is_valid holds
// `qrow` below is anything that yields this lane's value of batch column c as qrow[c]: a pointer into the
// batch, or one of the sweep's row holders (registers / an LDS column; gear_pipeline.hpp)
template <class T, class Q> __device__ bool limits_ok(const SceneDev<T>& S, const Views<T>& V, const Q& qrow) {
  bool ok = true;
  for (int l = 0; l < S.n_limits; l++) {
    const GEAR_CAS LimitDev<T>* L = V.limits + l;
    T v = joint_value(qrow[L->col], L->direct, L->mult, L->off);
    ok = ok && (v <= L->hi && v >= L->lo);
  }
  return ok;
}

// joint position of node k for this lane's configuration
template <class T, class Q> __device__ __forceinline__ T node_position(const GEAR_CAS NodeDev<T>* nd, const Q& qrow) {
  T pos = nd->const_pos;
  int col = nd->col;
  if (col >= 0) pos = joint_value(qrow[col], nd->direct, nd->mult, nd->off);
  return pos;
}

// A node's record as scalars.  Read field by field where each is needed, the record costs the walk four
// dependent trips to the constant cache per node (parent slot -> column -> mimic terms -> origin); read as a
// whole it is one.  (Requesting the NEXT node's record ahead of this node's arithmetic was measured too: the
// 21 scalar registers it keeps live spill, -2 %.)
template <class T> struct NodeRec {
  V3<T> ot, axis;
  Q4<T> oq;
  T mult, off, const_pos;
  int joint_type, col, direct, parent_slot, save_slot;
};
template <class T> __device__ __forceinline__ NodeRec<T> node_rec(const GEAR_CAS NodeDev<T>* nd) {
  NodeRec<T> r;
  r.ot = {nd->ot[0], nd->ot[1], nd->ot[2]};
  r.oq = {nd->oq[0], nd->oq[1], nd->oq[2], nd->oq[3]};
  r.axis = {nd->axis[0], nd->axis[1], nd->axis[2]};
  r.mult = nd->mult; r.off = nd->off; r.const_pos = nd->const_pos;
  r.joint_type = nd->joint_type; r.col = nd->col; r.direct = nd->direct;
  r.parent_slot = nd->parent_slot; r.save_slot = nd->save_slot;
  return r;
}

// world pose of a node given the walk state; updates `cur`
template <class T, class Q>
__device__ __forceinline__ void node_world_rec(const NodeRec<T>& r, const Q& qrow, const PoseStore<T>& st, Pose<T>& cur) {
  Pose<T> parent = cur;
  if (r.parent_slot == PARENT_IDENTITY) parent = {{T(0), T(0), T(0)}, {T(0), T(0), T(0), T(1)}};
  else if (r.parent_slot >= 0) parent = st.get(r.parent_slot);
  T pos = r.const_pos;
  if (r.col >= 0) pos = joint_value(qrow[r.col], r.direct, r.mult, r.off);
#ifdef GEAR_HIP_ABLATE_SINCOS  // attribution build (tools/attrib_sweep.sh): library sincosf instead of the f64-rounded one
  T s = T(0), c = T(1);
  if (r.joint_type == JT_ROT) { s = sin(pos / T(2)); c = cos(pos / T(2)); }
  cur = fk_step_sc(parent, r.ot, r.oq, r.axis, r.joint_type, pos, s, c);
#else
  cur = fk_step(parent, r.ot, r.oq, r.axis, r.joint_type, pos);
#endif
  if (r.save_slot >= 0) st.put(r.save_slot, cur);
}

// world pose of node k given the walk state; updates `cur`.  SC: sin/cos of the half angle were
// computed ahead of the walk and sit at sc[(2k) * 64 + lane], sc[(2k + 1) * 64 + lane].
template <class T, bool SC = false, class Q = const T*>
__device__ void node_world(const Views<T>& V, int k, const Q& qrow, const PoseStore<T>& st, Pose<T>& cur,
                           const T* sc = nullptr) {
  const GEAR_CAS NodeDev<T>* nd = V.nodes + k;
  if (!SC) {
    node_world_rec(node_rec<T>(nd), qrow, st, cur);
    return;
  }
  Pose<T> parent = cur;
  int ps = nd->parent_slot;
  if (ps == PARENT_IDENTITY) parent = {{T(0), T(0), T(0)}, {T(0), T(0), T(0), T(1)}};
  else if (ps >= 0) parent = st.get(ps);
  const T pos = node_position(nd, qrow);
  const V3<T> ot{nd->ot[0], nd->ot[1], nd->ot[2]}, axis{nd->axis[0], nd->axis[1], nd->axis[2]};
  const Q4<T> oq{nd->oq[0], nd->oq[1], nd->oq[2], nd->oq[3]};
  T s = T(0), c = T(1);
  if (nd->joint_type == JT_ROT) { s = sc[(2 * k) * kWave]; c = sc[(2 * k + 1) * kWave]; }
  cur = fk_step_sc(parent, ot, oq, axis, nd->joint_type, pos, s, c);
  int ss = nd->save_slot;
  if (ss >= 0) st.put(ss, cur);
}

// world centre of a shape's bounding sphere
template <class T, class SP> GD V3<T> bound_centre(SP sh, const Pose<T>& w) {
  if (sh->kind != SK_MESH) return w.t;
  return w.t + qrot(w.r, V3<T>{sh->bc[0], sh->bc[1], sh->bc[2]});
}

template <class T> __device__ PoseStore<T> make_store(T* gstore, size_t gstride, size_t i) {
  extern __shared__ unsigned char gear_lds_raw[];
  if (gstore) return {gstore, gstride, i};
  return {reinterpret_cast<T*>(gear_lds_raw), (size_t)kWave, (size_t)threadIdx.x};
}

// ---------------------------------------------------------------------------
// k_check — the hot kernel
//
// Per wavefront (64 configurations):
//   1. limits + FK walk; every robot shape's world pose goes to the lane's pose
//      column (LDS, or a global workspace for very large robots).
//   2. cull sweep: 32 (shape, obstacle) pairs at a time, five VALU ops per pair
//      with the obstacle record in SGPRs; the sign bit of d^2 - R^2 is shifted
//      into a per-lane survivor mask (v_alignbit), so there is no branch and no
//      compare in the sweep.  Self pairs are swept the same way.
//   3. survivors are rare (a few per configuration) and land on different lanes
//      at different times, so running their narrow phase in place would leave
//      63 of 64 lanes idle.  Instead every survivor is pushed to a wave-shared
//      queue in LDS (ballot + mbcnt compaction) and the queue is drained 64
//      items at a time, one item per lane: closed forms / SAT first; what those
//      cannot decide is re-queued and drained by the GJK margin query.  Hits
//      are OR-ed into a per-wave bit mask in LDS indexed by the owning lane.
// ---------------------------------------------------------------------------
constexpr int kQCap = 448;  // stage-1 queue entries per wave
constexpr int kGCap = 192;  // stage-2 (exact) queue entries per wave

GD unsigned sign_bits(float t) { return __float_as_uint(t); }
GD unsigned sign_bits(double t) { return (unsigned)__double2hiint(t); }

template <class T> struct Eps;
template <> struct Eps<float> { static constexpr float v = 1.2e-7f; };
template <> struct Eps<double> { static constexpr double v = 2.3e-16; };

template <class T> struct WaveStore {
  T* base;
  size_t stride, col0;  // column of lane l = col0 + l
  GD void put(int slot, int lane, const Pose<T>& p) const {
    T* b = base + (size_t)slot * 7 * stride + col0 + lane;
    b[0] = p.t.x; b[stride] = p.t.y; b[2 * stride] = p.t.z;
    b[3 * stride] = p.r.i; b[4 * stride] = p.r.j; b[5 * stride] = p.r.k; b[6 * stride] = p.r.w;
  }
  GD Pose<T> get(int slot, int lane) const {
    const T* b = base + (size_t)slot * 7 * stride + col0 + lane;
    return {{b[0], b[stride], b[2 * stride]}, {b[3 * stride], b[4 * stride], b[5 * stride], b[6 * stride]}};
  }
  GD V3<T> centre(int slot, int lane) const {
    const T* b = base + (size_t)slot * 7 * stride + col0 + lane;
    return {b[0], b[stride], b[2 * stride]};
  }
};

template <bool GSTORE> __device__ __forceinline__ void wave_sync() {
  // lanes of one wave exchange data through LDS (and, with a global pose store, through memory)
  if (GSTORE) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Sweep of K robot shapes against 32 obstacles (4 blocks of 8).  Obstacle blocks are the outer loop and
// the K shapes the inner one: a block's 40 scalars feed 48*K VALU ops, and exactly ONE block load is
// in flight while the previous block is being used.  (Scalar loads return out of order, so the only
// wait there is is "all outstanding loads": a second load in flight would be waited for as soon as
// the first one's data is needed.)
// Half a cull block: 4 obstacles' records, 20 scalars [5][4] (wave-uniform -> SGPRs).  Two of them
// are 40 SGPRs, which leaves the rest of the scalar file to the kernel (two 8-obstacle blocks do not:
// the compiler then parks prefetched records in VGPR lanes, one VALU op per scalar each way).
template <class T> struct CullQuad {
  T v[20];
  template <class P> GD void load(P p) {
#pragma unroll
    for (int i = 0; i < 20; i++) v[i] = p[i];
  }
};

// Four obstacles against K shapes.  For a fixed obstacle the K shapes are K independent chains
// (add, 4 fma, alignbit), issued stage by stage so that dependent instructions sit K slots apart.
template <class T, int K>
__device__ __forceinline__ void sweep_quad(const CullQuad<T>& b, const V3<T> (&c)[K], const T (&rho)[K],
                                           const T (&a0)[K], unsigned (&m)[K]) {
#pragma unroll
  for (int k = 0; k < 4; k++) {
    T t[K];
#pragma unroll
    for (int j = 0; j < K; j++) t[j] = a0[j] + b.v[16 + k];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < K; j++) t[j] = t_fma(b.v[k], c[j].x, t[j]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < K; j++) t[j] = t_fma(b.v[4 + k], c[j].y, t[j]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < K; j++) t[j] = t_fma(b.v[8 + k], c[j].z, t[j]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < K; j++) t[j] = t_fma(b.v[12 + k], rho[j], t[j]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < K; j++) m[j] = __builtin_amdgcn_alignbit(m[j], sign_bits(t[j]), 31);  // m = (m << 1) | (t < 0)
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Sweep of K robot shapes against 32 obstacles (8 quads).  Obstacle records are the outer loop and
// the K shapes the inner one: a quad's 20 scalars feed 24*K VALU ops, and exactly ONE quad load is
// in flight while the previous quad is being used.  (Scalar loads return out of order, so the only
// wait there is is "all outstanding loads": a second load in flight would be waited for as soon as
// the first one's data is needed.)
template <class T, int K>
__device__ __forceinline__ void sweep_unit(const GEAR_CAS T* rec, const V3<T> (&c)[K], const T (&rho)[K],
                                           const T (&a0)[K], unsigned (&m)[K]) {
  CullQuad<T> A, B;
#pragma unroll
  for (int j = 0; j < K; j++) m[j] = 0u;
  A.load(rec);
#pragma unroll
  for (int h = 0; h < 8; h += 2) {
    __builtin_amdgcn_sched_barrier(0);
    B.load(rec + 20 * (h + 1));
    sweep_quad<T, K>(A, c, rho, a0, m);
    __builtin_amdgcn_sched_barrier(0);
    if (h + 2 < 8) A.load(rec + 20 * (h + 2));
    sweep_quad<T, K>(B, c, rho, a0, m);
  }
}

constexpr size_t queue_bytes(int K) { return (size_t)(kQCap + kGCap + 4 + K * kWave) * 4; }

template <class T, bool COUNT, bool GSTORE, int K>
__global__ void __launch_bounds__(kWave)
k_check(SceneDev<T> S, const T* __restrict__ q, size_t n, uint32_t what, uint8_t* __restrict__ flags,
        CountersDev* __restrict__ ctr, T* gstore, size_t gstride, const unsigned* __restrict__ redo) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gear_lds[];
  const Views<T> V(S);
  const int lane = threadIdx.x;
  // redo == nullptr: the workgroup evaluates block blockIdx.x of 64 configurations.
  // redo != nullptr: last stage of the pipeline (gear_pipeline.hpp), whose exact stage has written every flag byte:
  // redo[b] == kXRedo marks a block whose sweep wavefront ran out of room and asks for this kernel's answer
  // instead.  The workgroup looks at the marks of blocks 64 * blockIdx.x ... + 63 (one load) and evaluates the
  // marked ones one after the other (normally none: the launch is a sixty-fourth of the blocks and returns).
  const size_t n_blocks = (n + kWave - 1) / kWave;
  const size_t base = redo ? (size_t)blockIdx.x * kWave : (size_t)blockIdx.x;
  unsigned long long todo = 1;
  if (redo) todo = __ballot(base + lane < n_blocks && redo[base + lane < n_blocks ? base + lane : 0] == kXRedo);
  while (todo != 0) {
  const size_t wb = base + (redo ? (size_t)__builtin_ctzll(todo) : (size_t)0);
  todo &= todo - 1;
  wave_sync<GSTORE>();  // (a second block reuses the first one's LDS)
  size_t i = wb * kWave + lane;
  const bool live = i < n;
  if (!live) i = n - 1;  // keep the lane on valid data; it pushes and stores nothing
  const T* qrow = q + i * (size_t)S.n_cols;
  const int pose_slots = S.n_slots + S.n_shapes;
  WaveStore<T> st;
  unsigned* q1;
  if (GSTORE) {
    st = {gstore, gstride, wb * kWave};
    q1 = reinterpret_cast<unsigned*>(gear_lds);
  } else {
    st = {reinterpret_cast<T*>(gear_lds), (size_t)kWave, (size_t)0};
    q1 = reinterpret_cast<unsigned*>(gear_lds + (size_t)pose_slots * 7 * kWave * sizeof(T));
  }
  unsigned* q2 = q1 + kQCap;
  unsigned* hits = q2 + kGCap;  // [0..1] env bits of lanes 0-31 / 32-63, [2..3] self bits
  unsigned* mst = hits + 4;     // [K][64]: survivor masks of the current sweep unit (each lane reads its own)
  if (lane < 4) hits[lane] = 0;

  const bool do_env = (what & 1u) && S.n_obs > 0;
  const bool do_self = (what & 2u) && S.n_spairs > 0;
  const bool dbg_no_exact = GEAR_DBG(what, 0x100u), dbg_no_drain = GEAR_DBG(what, 0x200u);  // profiling ablations
  const T margin = S.margin;
  LaneCounters lc{0, 0, 0, 0};
  unsigned long long npairs = 0;

  const bool ok = (what & kWhatNoLimits) != 0 || limits_ok(S, V, qrow);  // GH_CHECK_NO_LIMITS: geometry only

  // ---- 1. FK walk -----------------------------------------------------------
  {
    PoseStore<T> ps{st.base, st.stride, st.col0 + lane};
    Pose<T> cur{{T(0), T(0), T(0)}, {T(0), T(0), T(0), T(1)}};
    for (int k = 0; k < S.n_nodes; k++) {
      node_world(V, k, qrow, ps, cur);
      for (int s = V.nodes[k].shape_begin; s < V.nodes[k].shape_end; s++)
        ps.put(S.n_slots + s, pose_mul(cur, shape_pose<T>(V.rshapes + s)));
    }
  }
  wave_sync<GSTORE>();

  // ---- 2 + 3. cull sweep feeding the survivor queues -----------------------
  // A unit is K shapes x 32 obstacles (env) or 32 self shape pairs.
  const int n_og = S.n_obs_pad / 32;
  const int env_units = do_env ? (S.n_shapes + K - 1) / K * n_og : 0;
  const int self_groups = do_self ? (S.n_spairs + 31) / 32 : 0;
  const int units = env_units + self_groups;
  const bool push_ok = live && ok;  // the reference evaluates no collision after a limit error
  bool env_done = false, self_done = false;
  int qn = 0, gn = 0;  // wave-uniform queue fills
  int s0 = 0, o0 = 0;
  V3<T> c[K];
  T a0[K], rho[K];
#pragma unroll
  for (int j = 0; j < K; j++) { c[j] = {T(0), T(0), T(0)}; a0[j] = T(0); rho[j] = T(0); }

  for (int u = 0; u < units; u++) {
    const bool is_env = u < env_units;
    int nsub = 1;
    unsigned unit_base;
    if (is_env) {
      if (o0 == 0) {
#pragma unroll
        for (int j = 0; j < K; j++) {
          const int s = min(s0 + j, S.n_shapes - 1);  // a chunk's tail repeats the last shape; its masks are not read
          c[j] = st.centre(S.n_slots + s, lane);
          T r = V.rshapes[s].br;
          asm volatile("" : "+v"(r));  // keep the radius in a VGPR: the obstacle record supplies the SGPR operand
          rho[j] = r;
          const T cc = dot(c[j], c[j]), r2 = r * r;
          a0[j] = (cc - r2) - T(16) * Eps<T>::v * (cc + r2);
        }
      }
      unsigned m[K];
      if (GEAR_DBG(what, 0x800u)) {  // profiling ablation: no sweep
#pragma unroll
        for (int j = 0; j < K; j++) m[j] = 0u;
      } else {
        sweep_unit<T, K>(V.ocullq + (size_t)5 * o0, c, rho, a0, m);
      }
#pragma unroll
      for (int j = 0; j < K; j++) mst[j * kWave + lane] = m[j];
      nsub = min(K, S.n_shapes - s0);
      if (COUNT) npairs += (unsigned)(nsub * min(32, S.n_obs - o0 > 0 ? S.n_obs - o0 : 0));
      unit_base = ((unsigned)o0 << 16) | ((unsigned)s0 << 6) | (unsigned)lane;
      o0 += 32;
      if (o0 >= S.n_obs_pad) { o0 = 0; s0 += K; }
    } else {
      const int p0 = (u - env_units) * 32;
      const int cnt = min(32, S.n_spairs - p0);
      unsigned m = 0;
      for (int k = 0; k < 32; k++) {
        unsigned bit = 0;
        if (k < cnt) {
          const int s1 = V.spairs[4 * (p0 + k)], s2 = V.spairs[4 * (p0 + k) + 1];
          V3<T> d = st.centre(S.n_slots + s1, lane) - st.centre(S.n_slots + s2, lane);
          T rr = V.spair_r[p0 + k];
          bit = dot(d, d) < rr * rr ? 1u : 0u;
        }
        m = (m << 1) | bit;
      }
      mst[lane] = m;
      if (COUNT) npairs += (unsigned)cnt;
      unit_base = 0x80000000u | ((unsigned)p0 << 6) | (unsigned)lane;
    }

    for (int j = 0; j < nsub; j++) {
      unsigned m = mst[j * kWave + lane];
      const unsigned item_base = unit_base + ((unsigned)j << 6);
      if (is_env ? env_done : self_done) m = 0;
      if (!push_ok || GEAR_DBG(what, 0x400u)) m = 0;  // 0x400: profiling ablation, no push
      if (COUNT) lc.cull_pass += __popc(m);

      const bool last_group = u == units - 1 && j == nsub - 1;
      for (;;) {
        // push one survivor per lane per round while the queue has room for a full round
        while (qn + kWave <= kQCap) {
          const bool has = m != 0;
          const unsigned long long b = __ballot(has);
          if (b == 0) break;
          if (has) {
            const int z = __clz(m);  // bit 31 - z  <->  k = z within the group
            m &= ~(0x80000000u >> z);
            const unsigned it = is_env ? item_base + ((unsigned)z << 16) : item_base + ((unsigned)z << 6);
            q1[qn + __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u))] = it;
          }
          qn += __popcll(b);
        }
        const bool more = __ballot(m != 0) != 0;
        if (!more && !(last_group && (qn > 0 || gn > 0))) break;
        const bool final = last_group && !more;  // last drain of the wave: empty both queues

        // ---- drain: the single narrow-phase site -----------------------------
        wave_sync<GSTORE>();
        for (int base = 0; base < qn || (final && base == 0); base += kWave) {
          const int jq = base + lane;
          bool valid = jq < qn;
          const unsigned it = valid ? q1[jq] : 0u;
          const int ln = it & 63;
          const bool sf = (it >> 31) != 0;
          int sa, sbx;
          if (sf) { const int pi = (it >> 6) & 0x1ffffff; sa = V.spairs[4 * pi]; sbx = V.spairs[4 * pi + 1]; }
          else { sa = (it >> 6) & 1023; sbx = (it >> 16) & 32767; }
          int r = PR_FREE;
          if (valid) {
            const unsigned hw = hits[(sf ? 2 : 0) + (ln >> 5)];
            valid = ((hw >> (ln & 31)) & 1u) == 0;  // owner already known to collide
          }
          if (dbg_no_drain) valid = false;
          if (valid) {
            const ShapeRef<T> A = shape_ref<T>(V.rshapes + sa);
            const Pose<T> wa = st.get(S.n_slots + sa, ln);
            ShapeRef<T> B;
            Pose<T> wb;
            if (sf) { B = shape_ref<T>(V.rshapes + sbx); wb = st.get(S.n_slots + sbx, ln); }
            else { B = shape_ref<T>(V.oshapes + sbx); wb = shape_pose<T>(V.oshapes + sbx); }
            r = pair_stage1<T, COUNT>(A, wa, B, wb, margin, lc);
            if (r == PR_HIT) atomicOr(&hits[(sf ? 2 : 0) + (ln >> 5)], 1u << (ln & 31));
          }
          const bool ex = r == PR_EXACT;
          const unsigned long long eb = __ballot(ex);
          if (ex) q2[gn + __builtin_amdgcn_mbcnt_hi((unsigned)(eb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)eb, 0u))] = it;
          gn += __popcll(eb);
          const bool final_drain = final && base + kWave >= qn;
          while (gn >= kWave || (final_drain && gn > 0)) {
            wave_sync<GSTORE>();
            const int take = gn >= kWave ? kWave : gn;
            const int g0 = gn - take;
            bool v2 = lane < take;
            const unsigned it2 = v2 ? q2[g0 + lane] : 0u;
            const int ln2 = it2 & 63;
            const bool sf2 = (it2 >> 31) != 0;
            int sa2, sb2;
            if (sf2) { const int pi = (it2 >> 6) & 0x1ffffff; sa2 = V.spairs[4 * pi]; sb2 = V.spairs[4 * pi + 1]; }
            else { sa2 = (it2 >> 6) & 1023; sb2 = (it2 >> 16) & 32767; }
            if (v2) {
              const unsigned hw = hits[(sf2 ? 2 : 0) + (ln2 >> 5)];
              v2 = ((hw >> (ln2 & 31)) & 1u) == 0;
            }
            if (dbg_no_exact) v2 = false;
            if (v2) {
              const ShapeRef<T> A = shape_ref<T>(V.rshapes + sa2);
              const Pose<T> wa = st.get(S.n_slots + sa2, ln2);
              ShapeRef<T> B;
              Pose<T> wb;
              if (sf2) { B = shape_ref<T>(V.rshapes + sb2); wb = st.get(S.n_slots + sb2, ln2); }
              else { B = shape_ref<T>(V.oshapes + sb2); wb = shape_pose<T>(V.oshapes + sb2); }
              if (pair_stage2<T, COUNT>(A, wa, B, wb, V.verts, V.planes, margin, lc))
                atomicOr(&hits[(sf2 ? 2 : 0) + (ln2 >> 5)], 1u << (ln2 & 31));
            }
            gn = g0;
            wave_sync<GSTORE>();
          }
        }
        qn = 0;
        wave_sync<GSTORE>();
        env_done = ((hits[lane >> 5] >> (lane & 31)) & 1u) != 0;
        self_done = ((hits[2 + (lane >> 5)] >> (lane & 31)) & 1u) != 0;
        if (is_env ? env_done : self_done) m = 0;
        if (!more) break;
      }
    }
  }

  wave_sync<GSTORE>();
  const bool env_hit = ((hits[lane >> 5] >> (lane & 31)) & 1u) != 0;
  const bool self_hit = ((hits[2 + (lane >> 5)] >> (lane & 31)) & 1u) != 0;
  if (live) flags[i] = ok ? (uint8_t)((env_hit ? 1 : 0) | (self_hit ? 2 : 0)) : (uint8_t)4;
  if (COUNT) {
    // counters live on whichever lane ran the test; only their sums are meaningful
    atomicAdd(&ctr->configs, live ? 1ull : 0ull);
    atomicAdd(&ctr->pairs, live ? npairs : 0ull);
    atomicAdd(&ctr->cull_pass, (unsigned long long)lc.cull_pass);
    atomicAdd(&ctr->sat_tests, (unsigned long long)lc.sat_tests);
    atomicAdd(&ctr->exact_tests, (unsigned long long)lc.exact_tests);
    atomicAdd(&ctr->exact_iters, (unsigned long long)lc.exact_iters);
  }
  }  // while (todo)
}

// ---------------------------------------------------------------------------
// k_fk: poses[n][N][7]
// ---------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(kWave)
k_fk(SceneDev<T> S, const T* __restrict__ q, size_t n, T* __restrict__ poses, T* gstore, size_t gstride) {
  const Views<T> V(S);
  size_t i = (size_t)blockIdx.x * kWave + threadIdx.x;
  const bool live = i < n;
  if (!live) i = n - 1;
  const T* qrow = q + i * (size_t)S.n_cols;
  const PoseStore<T> st = make_store<T>(gstore, gstride, i);
  Pose<T> cur{{T(0), T(0), T(0)}, {T(0), T(0), T(0), T(1)}};
  for (int k = 0; k < S.n_nodes; k++) {
    node_world(V, k, qrow, st, cur);
    if (live) {
      T* o = poses + (i * (size_t)S.n_nodes + k) * 7;
      o[0] = cur.t.x; o[1] = cur.t.y; o[2] = cur.t.z;
      o[3] = cur.r.i; o[4] = cur.r.j; o[5] = cur.r.k; o[6] = cur.r.w;
    }
  }
}

// shared by the two diagnostic kernels: FK walk that leaves every robot shape's
// world pose in the store at slot n_slots + s
template <class T>
__device__ void walk_store_shapes(const SceneDev<T>& S, const Views<T>& V, const T* qrow, const PoseStore<T>& st) {
  Pose<T> cur{{T(0), T(0), T(0)}, {T(0), T(0), T(0), T(1)}};
  for (int k = 0; k < S.n_nodes; k++) {
    node_world(V, k, qrow, st, cur);
    for (int s = V.nodes[k].shape_begin; s < V.nodes[k].shape_end; s++)
      st.put(S.n_slots + s, pose_mul(cur, shape_pose<T>(V.rshapes + s)));
  }
}

// ---------------------------------------------------------------------------
// k_link_hits: masks[n][O][W]; bit k set when node k is yielded by check_env
// ---------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(kWave)
k_link_hits(SceneDev<T> S, const T* __restrict__ q, size_t n, uint32_t what, unsigned long long* __restrict__ masks,
            int W, T* gstore, size_t gstride) {
  const Views<T> V(S);
  size_t i = (size_t)blockIdx.x * kWave + threadIdx.x;
  const bool live = i < n;
  if (!live) i = n - 1;
  const T* qrow = q + i * (size_t)S.n_cols;
  const PoseStore<T> st = make_store<T>(gstore, gstride, i);
  const bool ok = (what & kWhatNoLimits) != 0 || limits_ok(S, V, qrow);  // GH_CHECK_NO_LIMITS: geometry only
  walk_store_shapes(S, V, qrow, st);
  LaneCounters lc{0, 0, 0, 0};
  const GEAR_CAS int32_t* owner = to_cas(S.obs_owner);
  unsigned long long m0 = 0, m1 = 0;
  for (int o = 0; o < S.n_obs; o++) {
    const GEAR_CAS ShapeDev<T>* ob = V.oshapes + o;
    const ShapeRef<T> B = shape_ref<T>(ob);
    const Pose<T> wb = shape_pose<T>(ob);
    // the triangles of a general mesh are consecutive device obstacles with one owner: their bits are OR-ed
    // (the reference tests the link against the whole mesh: a minimum over its triangles)
    if (o == 0 || owner[o] != owner[o - 1]) m0 = m1 = 0;
    for (int s = 0; s < S.n_shapes; s++) {
      const GEAR_CAS ShapeDev<T>* sh = V.rshapes + s;
      const int node = sh->node;
      const bool seen = node < 64 ? ((m0 >> node) & 1ull) : ((m1 >> (node - 64)) & 1ull);
      const Pose<T> ws = st.get(S.n_slots + s);
      const V3<T> c = bound_centre(sh, ws);
      T dx = c.x - V.ocull[4 * o], dy = c.y - V.ocull[4 * o + 1], dz = c.z - V.ocull[4 * o + 2];
      T rr = sh->br + V.ocull[4 * o + 3];
      if (!seen && dx * dx + dy * dy + dz * dz < rr * rr) {
        if (pair_within<T, false>(shape_ref<T>(sh), ws, B, wb, V.verts, V.planes, S.margin, lc)) {
          if (node < 64) m0 |= 1ull << node;
          else m1 |= 1ull << (node - 64);
        }
      }
    }
    if (live && (o == S.n_obs - 1 || owner[o + 1] != owner[o])) {
      unsigned long long* out = masks + (i * (size_t)S.n_obs_user + (size_t)owner[o]) * W;
      out[0] = ok ? m0 : 0ull;
      if (W > 1) out[1] = ok ? m1 : 0ull;
    }
  }
}

// ---------------------------------------------------------------------------
// k_self_hits: masks[n][ceil(P/64)]; bit p set when self pair p is yielded
// ---------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(kWave)
k_self_hits(SceneDev<T> S, const T* __restrict__ q, size_t n, uint32_t what, unsigned long long* __restrict__ masks,
            int W, T* gstore, size_t gstride) {
  const Views<T> V(S);
  size_t i = (size_t)blockIdx.x * kWave + threadIdx.x;
  const bool live = i < n;
  if (!live) i = n - 1;
  const T* qrow = q + i * (size_t)S.n_cols;
  const PoseStore<T> st = make_store<T>(gstore, gstride, i);
  const bool ok = (what & kWhatNoLimits) != 0 || limits_ok(S, V, qrow);  // GH_CHECK_NO_LIMITS: geometry only
  walk_store_shapes(S, V, qrow, st);
  LaneCounters lc{0, 0, 0, 0};
  unsigned long long m = 0;
  for (int p = 0; p < S.n_pairs; p++) {
    const PairDev pr{V.pairs[p].s1_begin, V.pairs[p].s1_end, V.pairs[p].s2_begin, V.pairs[p].s2_end};
    bool hit = false;
    for (int s1 = pr.s1_begin; s1 < pr.s1_end; s1++) {
      const GEAR_CAS ShapeDev<T>* h1 = V.rshapes + s1;
      const Pose<T> w1 = st.get(S.n_slots + s1);
      const V3<T> c1 = bound_centre(h1, w1);
      for (int s2 = pr.s2_begin; s2 < pr.s2_end; s2++) {
        const GEAR_CAS ShapeDev<T>* h2 = V.rshapes + s2;
        const Pose<T> w2 = st.get(S.n_slots + s2);
        const V3<T> d = c1 - bound_centre(h2, w2);
        T rr = (h1->br + h2->br + S.margin) * (T(1) + T(1e-4));
        if (!hit && dot(d, d) < rr * rr)
          hit = pair_within<T, false>(shape_ref<T>(h1), w1, shape_ref<T>(h2), w2, V.verts, V.planes, S.margin, lc);
      }
    }
    if (hit) m |= 1ull << (p & 63);
    if ((p & 63) == 63 || p == S.n_pairs - 1) {
      if (live) masks[i * (size_t)W + (p >> 6)] = ok ? m : 0ull;
      m = 0;
    }
  }
}

// ---------------------------------------------------------------------------
// k_edge_first: first sample of each edge with (flags & bad_mask) != 0
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_edge_first(const uint8_t* __restrict__ flags, const uint32_t* __restrict__ edge_off, size_t n_edges,
             uint8_t bad_mask, int32_t* __restrict__ first_bad) {
  size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n_edges) return;
  uint32_t lo = edge_off[e], hi = edge_off[e + 1];
  int32_t r = -1;
  for (uint32_t j = lo; j < hi; j++) {
    if (flags[j] & bad_mask) { r = (int32_t)(j - lo); break; }
  }
  first_bad[e] = r;
}

}  // namespace gear_hip
