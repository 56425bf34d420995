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

template <class T> struct Views {
  const GEAR_CAS NodeDev<T>* nodes;
  const GEAR_CAS LimitDev<T>* limits;
  const GEAR_CAS ShapeDev<T>* rshapes;
  const GEAR_CAS ShapeDev<T>* oshapes;
  const GEAR_CAS T* ocull;
  const GEAR_CAS PairDev* pairs;
  const GEAR_CAS T* verts;
  const GEAR_CAS T* planes;
  __device__ explicit Views(const SceneDev<T>& S)
      : nodes(to_cas(S.nodes)), limits(to_cas(S.limits)), rshapes(to_cas(S.rshapes)), oshapes(to_cas(S.oshapes)),
        ocull(to_cas(S.ocull)), pairs(to_cas(S.pairs)), verts(to_cas(S.verts)), planes(to_cas(S.planes)) {}
};

template <class T, class SP> GD ShapeRef<T> shape_ref(SP sh) {
  ShapeRef<T> r;
  r.kind = sh->kind;
  r.half = {sh->half[0], sh->half[1], sh->half[2]};
  r.br = sh->br;
  r.vb = sh->v_begin; r.vc = sh->v_count;
  r.fb = sh->f_begin; r.fc = sh->f_count;
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
template <class T> __device__ bool limits_ok(const SceneDev<T>& S, const Views<T>& V, const T* qrow) {
  bool ok = true;
  for (int l = 0; l < S.n_limits; l++) {
    const GEAR_CAS LimitDev<T>* L = V.limits + l;
    T v = joint_value(qrow[L->col], L->direct, L->mult, L->off);
    ok = ok && (v <= L->hi && v >= L->lo);
  }
  return ok;
}

// world pose of node k given the walk state; updates `cur`
template <class T>
__device__ void node_world(const Views<T>& V, int k, const T* qrow, const PoseStore<T>& st, Pose<T>& cur) {
  const GEAR_CAS NodeDev<T>* nd = V.nodes + k;
  Pose<T> parent = cur;
  int ps = nd->parent_slot;
  if (ps == PARENT_IDENTITY) parent = {{T(0), T(0), T(0)}, {T(0), T(0), T(0), T(1)}};
  else if (ps >= 0) parent = st.get(ps);
  T pos = nd->const_pos;
  int col = nd->col;
  if (col >= 0) pos = joint_value(qrow[col], nd->direct, nd->mult, nd->off);
  cur = fk_step(parent, V3<T>{nd->ot[0], nd->ot[1], nd->ot[2]}, Q4<T>{nd->oq[0], nd->oq[1], nd->oq[2], nd->oq[3]},
                V3<T>{nd->axis[0], nd->axis[1], nd->axis[2]}, nd->joint_type, pos);
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
// k_check
// ---------------------------------------------------------------------------
template <class T, bool COUNT>
__global__ void __launch_bounds__(kWave)
k_check(SceneDev<T> S, const T* __restrict__ q, size_t n, uint32_t what, uint8_t* __restrict__ flags,
        CountersDev* __restrict__ ctr, T* gstore, size_t gstride) {
  const Views<T> V(S);
  size_t i = (size_t)blockIdx.x * kWave + threadIdx.x;
  const bool live = i < n;
  if (!live) i = n - 1;  // keep the lane on valid data; it stores nothing
  const T* qrow = q + i * (size_t)S.n_cols;
  const PoseStore<T> st = make_store<T>(gstore, gstride, i);
  const bool do_env = (what & 1u) && S.n_obs > 0;
  const bool do_self = (what & 2u) && S.n_pairs > 0;
  const T margin = S.margin;
  LaneCounters lc{0, 0, 0, 0};
  unsigned long long npairs = 0;

  const bool ok = limits_ok(S, V, qrow);
  bool env_hit = false, self_hit = false;

  Pose<T> cur{{T(0), T(0), T(0)}, {T(0), T(0), T(0), T(1)}};
  for (int k = 0; k < S.n_nodes; k++) {
    node_world(V, k, qrow, st, cur);
    const int sb = V.nodes[k].shape_begin, se = V.nodes[k].shape_end;
    for (int s = sb; s < se; s++) {
      const GEAR_CAS ShapeDev<T>* sh = V.rshapes + s;
      const Pose<T> ws = pose_mul(cur, shape_pose<T>(sh));
      if (do_self) st.put(S.n_slots + s, ws);
      if (!do_env) continue;
      const V3<T> c = bound_centre(sh, ws);
      const T br = sh->br;
      const ShapeRef<T> A = shape_ref<T>(sh);
      for (int o = 0; o < S.n_obs; o++) {
        // bounding-sphere cull; ocull.w already holds (obstacle radius + margin) * (1 + slack)
        T dx = c.x - V.ocull[4 * o], dy = c.y - V.ocull[4 * o + 1], dz = c.z - V.ocull[4 * o + 2];
        T rr = br + V.ocull[4 * o + 3];
        bool near = dx * dx + dy * dy + dz * dz < rr * rr;
        if (COUNT) npairs++;
        if (near && !env_hit) {
          if (COUNT) lc.cull_pass++;
          const GEAR_CAS ShapeDev<T>* ob = V.oshapes + o;
          if (pair_within<T, COUNT>(A, ws, shape_ref<T>(ob), shape_pose<T>(ob), V.verts, V.planes, margin, lc))
            env_hit = true;
        }
      }
    }
  }

  if (do_self) {
    for (int p = 0; p < S.n_pairs; p++) {
      const PairDev pr{V.pairs[p].s1_begin, V.pairs[p].s1_end, V.pairs[p].s2_begin, V.pairs[p].s2_end};
      for (int s1 = pr.s1_begin; s1 < pr.s1_end; s1++) {
        const GEAR_CAS ShapeDev<T>* h1 = V.rshapes + s1;
        const Pose<T> w1 = st.get(S.n_slots + s1);
        const V3<T> c1 = bound_centre(h1, w1);
        const ShapeRef<T> A = shape_ref<T>(h1);
        for (int s2 = pr.s2_begin; s2 < pr.s2_end; s2++) {
          const GEAR_CAS ShapeDev<T>* h2 = V.rshapes + s2;
          const Pose<T> w2 = st.get(S.n_slots + s2);
          const V3<T> d = c1 - bound_centre(h2, w2);
          T rr = (h1->br + h2->br + margin) * (T(1) + T(1e-4));
          if (COUNT) npairs++;
          if (dot(d, d) < rr * rr && !self_hit) {
            if (COUNT) lc.cull_pass++;
            if (pair_within<T, COUNT>(A, w1, shape_ref<T>(h2), w2, V.verts, V.planes, margin, lc)) self_hit = true;
          }
        }
      }
    }
  }

  if (live) {
    uint8_t f = ok ? (uint8_t)((env_hit ? 1 : 0) | (self_hit ? 2 : 0)) : (uint8_t)4;
    flags[i] = f;
  }
  if (COUNT) {
    if (!live) { lc = {0, 0, 0, 0}; npairs = 0; }
    atomicAdd(&ctr->configs, live ? 1ull : 0ull);
    atomicAdd(&ctr->pairs, npairs);
    atomicAdd(&ctr->cull_pass, (unsigned long long)lc.cull_pass);
    atomicAdd(&ctr->sat_tests, (unsigned long long)lc.sat_tests);
    atomicAdd(&ctr->exact_tests, (unsigned long long)lc.exact_tests);
    atomicAdd(&ctr->exact_iters, (unsigned long long)lc.exact_iters);
  }
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
k_link_hits(SceneDev<T> S, const T* __restrict__ q, size_t n, unsigned long long* __restrict__ masks, int W,
            T* gstore, size_t gstride) {
  const Views<T> V(S);
  size_t i = (size_t)blockIdx.x * kWave + threadIdx.x;
  const bool live = i < n;
  if (!live) i = n - 1;
  const T* qrow = q + i * (size_t)S.n_cols;
  const PoseStore<T> st = make_store<T>(gstore, gstride, i);
  const bool ok = limits_ok(S, V, qrow);
  walk_store_shapes(S, V, qrow, st);
  LaneCounters lc{0, 0, 0, 0};
  for (int o = 0; o < S.n_obs; o++) {
    const GEAR_CAS ShapeDev<T>* ob = V.oshapes + o;
    const ShapeRef<T> B = shape_ref<T>(ob);
    const Pose<T> wb = shape_pose<T>(ob);
    unsigned long long m0 = 0, m1 = 0;
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
    if (live) {
      unsigned long long* out = masks + (i * (size_t)S.n_obs + o) * W;
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
k_self_hits(SceneDev<T> S, const T* __restrict__ q, size_t n, unsigned long long* __restrict__ masks, int W,
            T* gstore, size_t gstride) {
  const Views<T> V(S);
  size_t i = (size_t)blockIdx.x * kWave + threadIdx.x;
  const bool live = i < n;
  if (!live) i = n - 1;
  const T* qrow = q + i * (size_t)S.n_cols;
  const PoseStore<T> st = make_store<T>(gstore, gstride, i);
  const bool ok = limits_ok(S, V, qrow);
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
