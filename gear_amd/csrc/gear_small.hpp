// gear_small.hpp — k_small: the check for SMALL batches (an RRT connect line, a smoothing shortcut, a
// start/goal test: joint_path_planner.rs:168-188 calls is_feasible once per sample of such a line).
//
// The large-batch kernels give a configuration one lane, so a call of a few hundred samples is the serial
// life of a handful of wavefronts on an otherwise empty chip (~55 us through three launches).  Here the
// work of ONE configuration is spread over many lanes and everything happens in one launch:
//
//   P0  one lane per (configuration, node):   the node's transform relative to its parent (sin/cos of the
//                                             joint's half angle, origin * joint)           -> LDS
//       one lane per (configuration, limit):  Range::is_valid
//   P1  one lane per configuration:           the FK walk, world = parent * local          -> LDS (node poses)
//   P2  eight lanes per (configuration, shape): world pose of the shape, grid cell, ONE candidate obstacle
//                                             each (a slot of the cell's block), bounding-sphere cull,
//                                             oriented-box cull; survivors are pushed to the workgroup's
//                                             item queue in LDS
//       one lane per (configuration, self shape pair): centre cull, survivors pushed likewise
//   P3a one lane per queued item:             closed forms / SAT (pair_stage1); a hit is OR-ed into the
//                                             configuration's flags, an undecided item queued again
//   P3b one lane per undecided item:          GJK margin query + hollow-mesh rule (pair_stage2), unless the
//                                             configuration is known to collide by now
//   P4  one lane per configuration:           the flag byte
//   P5  (gh_check_edges) the last workgroup:  first flagged sample of every edge
//
// A workgroup takes as many configurations as let each phase finish in one pass of its 256 lanes
// (plan_small), so a call is five short chains of dependent memory trips, not five per shape.  When the
// queue runs full the workgroup walks the broadphase a second time and decides every survivor in place: the
// result never depends on the queue's capacity.
//
// The arithmetic is the large-batch kernels': the same t_sincos / fk_step_sc / pose_mul sequence per node
// and shape (bit-identical poses) and the same pair_stage1 / pair_stage2 per pair; the culls in front of
// them are conservative filters, so the flags equal k_check's bit for bit (tests/test_gpu_small.py).
#pragma once
#include "gear_pipeline.hpp"

namespace gear_hip {

constexpr int kSmThreads = 256;
constexpr int kSmMaxC = 64;  // configurations per workgroup (the item codes keep 6 bits for it)

// Poses in k_small's LDS: one 8-element record per (slot or node, configuration), read and written with two
// 16-byte accesses (four in f64) at one computed address.  (The per-lane columns of the large-batch kernels
// serve 64 lanes reading the same component; here a handful of lanes read whole poses.)
template <class T> struct PoseRecs {
  T* base;
  int C, c;
  GD void put(int slot, const Pose<T>& p) const {
    Quad<T>* r = reinterpret_cast<Quad<T>*>(base + ((size_t)slot * C + c) * 8);
    r[0] = Quad<T>{p.t.x, p.t.y, p.t.z, p.r.i};
    r[1] = Quad<T>{p.r.j, p.r.k, p.r.w, T(0)};
  }
  GD Pose<T> get(int slot) const {
    const Quad<T>* r = reinterpret_cast<const Quad<T>*>(base + ((size_t)slot * C + c) * 8);
    const Quad<T> a = r[0], b = r[1];
    return {{a.x, a.y, a.z}, {a.w, b.x, b.y, b.z}};
  }
};

struct SmallPlan {
  int C = 0;            // configurations per workgroup
  unsigned qcap = 0;    // item queue entries
  size_t lds = 0;       // dynamic LDS bytes
  bool cull_lds = false;  // the obstacles' cull records (16 B each in f32) sit in LDS
};

// C, the configurations per workgroup.  P2 has eight lanes per (configuration, shape) — one per slot of a
// cell block — and a lane per (configuration, self shape pair); P0 a lane per (configuration, node) and per
// (configuration, limit).  A small batch leaves the chip empty, so it is cut into MANY small workgroups, each
// of which finishes every phase in one pass of its 256 lanes (a second pass is a second chain of dependent
// memory trips); a batch of thousands gets larger workgroups, about two per compute unit.
constexpr int kSmSlotLanes = 8;
constexpr size_t kSmGroupsTarget = 512;  // two workgroups per compute unit are resident at once (register use)
constexpr int kSmCullLdsMax = 1024;  // obstacles (+ the dummy) up to which the cull table is copied into LDS

template <class T> inline SmallPlan plan_small(int n_nodes, int n_shapes, int n_spairs, int n_limits, int n_slots, int n_obs,
                                               int n_cols, unsigned qcap, size_t n) {
  SmallPlan p;
  const int fine = std::max({kSmSlotLanes * n_shapes + n_spairs, n_nodes + n_limits, 1});
  const int coarse = std::max({n_shapes + n_spairs, n_nodes + n_limits, 1});
  const int c_lo = std::max(1, kSmThreads / fine), c_hi = std::min(kSmMaxC, std::max(1, kSmThreads / coarse));
  const size_t want = (n + kSmGroupsTarget - 1) / kSmGroupsTarget;
  p.C = (int)std::min<size_t>(std::max<size_t>(want, (size_t)c_lo), (size_t)std::max(c_lo, c_hi));
  p.qcap = qcap;
  p.cull_lds = n_obs + 1 <= kSmCullLdsMax;
  const size_t poses = (size_t)(n_slots + n_nodes) * 8 * p.C * sizeof(T);
  p.lds = (poses + 15) / 16 * 16 + ((size_t)(2 * kSmMaxC + 4 + 2 * qcap + n_nodes) * 4 + 15) / 16 * 16 +
          (p.cull_lds ? (size_t)(n_obs + 1) * 4 * sizeof(T) : 0) + (size_t)p.C * (size_t)n_cols * sizeof(T);
  return p;
}

// first_bad != nullptr: the call is gh_check_edges' — the workgroup that finishes last (a ticket from
// *done, which it leaves at zero again) finds every edge's first flagged sample, so the whole call is this
// one launch.  One such call at a time per counter (the host-buffer entry point holds the scene's lock
// until the stream has drained).  flags_near: a second copy of the flag bytes in device memory for that
// reduction to read when `flags` itself is host memory (a byte-wise scan over the fabric would cost more than
// the launch it saves); null: read `flags`.
template <class T>
__global__ void __launch_bounds__(kSmThreads)
k_small(SceneDev<T> S, const T* __restrict__ q, size_t n, uint32_t what, uint8_t* flags, int C, unsigned qcap, int all_blk,
        int cull_lds,
        const uint32_t* __restrict__ edge_off, unsigned n_edges, unsigned bad_mask, int32_t* __restrict__ first_bad,
        uint8_t* flags_near, unsigned* done, unsigned* notify, unsigned notify_seq) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gear_lds[];
  const Views<T> V(S);
  const int tid = threadIdx.x;
  const size_t c0 = (size_t)blockIdx.x * (size_t)C;
  const int Cl = (int)(n - c0 < (size_t)C ? n - c0 : (size_t)C);  // configurations of this workgroup (>= 1)
  const int N = S.n_nodes, L = S.n_shapes;

  // LDS: [branch slots][node poses] as records (PoseRecs), flag words, limit words, queue fills, queues, node meta
  T* const slot_base = reinterpret_cast<T*>(gear_lds);
  T* const node_base = slot_base + (size_t)S.n_slots * 8 * C;
  const size_t t_bytes = ((size_t)(S.n_slots + N) * 8 * C * sizeof(T) + 15) / 16 * 16;
  unsigned* const cflag = reinterpret_cast<unsigned*>(gear_lds + t_bytes);  // [64] bit0 env, bit1 self
  unsigned* const cbad = cflag + kSmMaxC;                                     // [64] != 0: outside a limit
  unsigned* const qfill = cbad + kSmMaxC;                                     // [4] queue fill, last-workgroup flag, exact-queue fill
  unsigned* const queue = qfill + 4;                                          // [qcap] survivors of the culls
  unsigned* const xqueue = queue + qcap;                                      // [qcap] items stage 1 could not decide
  unsigned* const nmeta = xqueue + qcap;                                      // [N] what the walk needs of a node, packed
  Quad<T>* const lcull = reinterpret_cast<Quad<T>*>(gear_lds + t_bytes + ((size_t)(2 * kSmMaxC + 4 + 2 * qcap + N) * 4 + 15) / 16 * 16);  // [n_obs + 1]
  T* const qs = reinterpret_cast<T*>(lcull + (cull_lds ? S.n_obs + 1 : 0));  // [C][n_cols] this workgroup's rows of the batch

  const bool do_env = (what & 1u) && S.n_obs > 0;
  const bool do_self = (what & 2u) && S.n_spairs > 0;
  const bool do_limits = (what & kWhatNoLimits) == 0;  // GH_CHECK_NO_LIMITS: geometry only
  const T margin = S.margin;
  LaneCounters lc{0, 0, 0, 0};

  // What a P2 lane needs of its shape (or of its self pair's two shapes) depends on the lane alone, not on the
  // batch: it is requested here, ahead of P0 and P1, and is in registers by the time the node poses are.
  struct EnvUnit { Pose<T> lp; T br, half[3]; int kind, node; T org[3], inv_h; int nx, ny, nz, off_base; };
  struct SelfUnit { Pose<T> lp1, lp2; int node1, node2; T rr; };
  auto load_env = [&](int s) {
    const ShapeDev<T>* sh = S.rshapes + s;
    EnvUnit e;
    e.lp = shape_pose<T>(sh);
    e.br = sh->br;
    e.half[0] = sh->half[0]; e.half[1] = sh->half[1]; e.half[2] = sh->half[2];
    e.kind = sh->kind;
    e.node = sh->node;
    e.org[0] = e.org[1] = e.org[2] = e.inv_h = T(0);
    e.nx = e.ny = e.nz = e.off_base = 0;
    if (all_blk < 0) {
      const GridDev<T>* g = S.grids + S.shape_grid[s];
      e.org[0] = g->org[0]; e.org[1] = g->org[1]; e.org[2] = g->org[2];
      e.inv_h = g->inv_h;
      e.nx = g->nx; e.ny = g->ny; e.nz = g->nz; e.off_base = g->off_base;
    }
    return e;
  };
  auto load_self = [&](int p) {
    const ShapeDev<T>* h1 = S.rshapes + S.spairs[4 * p];
    const ShapeDev<T>* h2 = S.rshapes + S.spairs[4 * p + 1];
    SelfUnit u;
    u.lp1 = shape_pose<T>(h1);
    u.lp2 = shape_pose<T>(h2);
    u.node1 = h1->node;
    u.node2 = h2->node;
    u.rr = S.spair_r[p];
    return u;
  };
  const int n_env = do_env && !GEAR_DBG(what, 0x400u) ? Cl * L * kSmSlotLanes : 0;  // 0x400: profiling ablation, no broadphase at all
  const int n_self = do_self && !GEAR_DBG(what, 0x400u) ? Cl * S.n_spairs : 0;
  EnvUnit eu0{};
  SelfUnit su0{};
  if (tid < n_env) eu0 = load_env((tid / kSmSlotLanes) / Cl);
  else if (tid < n_env + n_self) su0 = load_self((tid - n_env) / Cl);
  // the cull table, on its way to LDS: requested now, parked behind P0's arithmetic
  Quad<T> ctab[kSmCullLdsMax / kSmThreads];
  if (cull_lds && do_env) {
#pragma unroll
    for (int k = 0; k < kSmCullLdsMax / kSmThreads; k++) {
      const int o = k * kSmThreads + tid;
      ctab[k] = reinterpret_cast<const Quad<T>*>(S.ocull)[o <= S.n_obs ? o : S.n_obs];
    }
  }

#ifdef GEAR_HIP_ABLATE  // phase stamps of one workgroup, printed at its end (what & 0x10000000; tools/ablate_small.py)
  unsigned long long stamp[8];
  int n_stamp = 0;
#define GEAR_SMALL_STAMP() do { if (n_stamp < 8) stamp[n_stamp++] = wall_clock64(); } while (0)
#else
#define GEAR_SMALL_STAMP() do { } while (0)
#endif
  GEAR_SMALL_STAMP();
  // P0's own inputs: the node (or limit) record of this lane's first unit is requested now, and the workgroup's
  // rows of the batch go to LDS with one coalesced read — the joint value then costs P0 no second memory trip
  // behind the record that names its column
  struct NodeUnit { V3<T> ot, axis; Q4<T> oq; T mult, off, const_pos; int jt, col, direct; unsigned meta; };
  struct LimUnit { T mult, off, lo, hi; int col, direct; };
  auto load_node = [&](int k) {
    const NodeDev<T>* nd = S.nodes + k;
    NodeUnit u;
    u.ot = {nd->ot[0], nd->ot[1], nd->ot[2]};
    u.axis = {nd->axis[0], nd->axis[1], nd->axis[2]};
    u.oq = {nd->oq[0], nd->oq[1], nd->oq[2], nd->oq[3]};
    u.mult = nd->mult; u.off = nd->off; u.const_pos = nd->const_pos;
    u.jt = nd->joint_type; u.col = nd->col; u.direct = nd->direct;
    // parent slot + 2 | (save slot + 1) << 12 | owns shapes << 31: read back by the walk from LDS, whose loads
    // complete in order (a scalar load of the node record would make every wait a wait for everything)
    u.meta = (unsigned)(nd->parent_slot + 2) | ((unsigned)(nd->save_slot + 1) << 12) | (nd->shape_end > nd->shape_begin ? 0x80000000u : 0u);
    return u;
  };
  auto load_lim = [&](int l) {
    const LimitDev<T>* lim = S.limits + l;
    return LimUnit{lim->mult, lim->off, lim->lo, lim->hi, lim->col, lim->direct};
  };
  const int n_loc = (do_env || do_self) ? Cl * N : 0;
  const int n_lim = do_limits ? Cl * S.n_limits : 0;
  NodeUnit nu0{};
  LimUnit lu0{};
  if (tid < n_loc) nu0 = load_node(tid / Cl);
  else if (tid < n_loc + n_lim) lu0 = load_lim((tid - n_loc) / Cl);
  for (int u = tid; u < Cl * S.n_cols; u += kSmThreads) qs[u] = q[c0 * (size_t)S.n_cols + (size_t)u];
  if (tid < kSmMaxC) { cflag[tid] = 0; cbad[tid] = 0; }
  if (tid == 0) { qfill[0] = 0; qfill[2] = 0; }
  __syncthreads();

  // ---- P0: every node's transform relative to its parent; every limit ------------------------------------
  for (int u = tid; u < n_loc + n_lim; u += kSmThreads) {
    if (u < n_loc) {
      const int c = u % Cl, k = u / Cl;
      const NodeUnit nd = u == tid ? nu0 : load_node(k);
      T pos = nd.const_pos;  // node_position, with a lane-varying node
      if (nd.col >= 0) pos = joint_value(qs[c * S.n_cols + nd.col], nd.direct, nd.mult, nd.off);
      T s = T(0), cs = T(1);
      if (nd.jt == JT_ROT) t_sincos(pos / T(2), &s, &cs);
      PoseRecs<T>{node_base, C, c}.put(k, fk_local_sc(nd.ot, nd.oq, nd.axis, nd.jt, pos, s, cs));
      if (c == 0) nmeta[k] = nd.meta;
    } else {
      const int v = u - n_loc;
      const int c = v % Cl, l = v / Cl;
      const LimUnit lim = u == tid ? lu0 : load_lim(l);
      const T val = joint_value(qs[c * S.n_cols + lim.col], lim.direct, lim.mult, lim.off);
      if (!(val <= lim.hi && val >= lim.lo)) cbad[c] = 1u;  // Range::is_valid, as limits_ok spells it
    }
  }
  if (cull_lds && do_env) {
#pragma unroll
    for (int k = 0; k < kSmCullLdsMax / kSmThreads; k++) {
      const int o = k * kSmThreads + tid;
      if (o <= S.n_obs) lcull[o] = ctab[k];
    }
  }
  __syncthreads();
  GEAR_SMALL_STAMP();

  // ---- P1: the FK walk, one lane per configuration: world = parent * local, root first ------------------
  if (tid < Cl && (do_env || do_self) && !cbad[tid] && !GEAR_DBG(what, 0x800u)) {  // the reference evaluates no collision after a limit error
    const int c = tid;
    const PoseRecs<T> slots{slot_base, C, c};
    const PoseRecs<T> nodes{node_base, C, c};
    Pose<T> cur{{T(0), T(0), T(0)}, {T(0), T(0), T(0), T(1)}};
    // the next node's record and local transform are requested before this node's product is formed: the walk is
    // one dependent chain, and their latency would otherwise sit inside it
    unsigned meta = nmeta[0];
    Pose<T> loc = nodes.get(0);
    for (int k = 0; k < N; k++) {
      const int kn = k + 1 < N ? k + 1 : k;
      const unsigned meta_n = nmeta[kn];
      const Pose<T> loc_n = nodes.get(kn);
      const int ps = (int)(meta & 0xfffu) - 2, ss = (int)((meta >> 12) & 0xfffu) - 1;
      Pose<T> parent = cur;
      if (ps == PARENT_IDENTITY) parent = {{T(0), T(0), T(0)}, {T(0), T(0), T(0), T(1)}};
      else if (ps >= 0) parent = slots.get(ps);
      cur = pose_mul(parent, loc);
      if (ss >= 0) slots.put(ss, cur);
      if (meta >> 31) nodes.put(k, cur);
      meta = meta_n; loc = loc_n;
    }
  }
  __syncthreads();
  GEAR_SMALL_STAMP();

  // One item: both shapes, both poses (node pose from LDS times the shape's local pose: the very product the
  // walk of the large-batch kernels forms).  Codes as in the pipeline, with the configuration's index inside
  // the workgroup in the low 6 bits.  Needs nothing but the node poses, so it can run anywhere after P1.
  auto fetch = [&](unsigned code, ShapeRef<T>& A, Pose<T>& wa, ShapeRef<T>& B, Pose<T>& wb) {
    const PoseRecs<T> nodes{node_base, C, (int)(code & 63u)};
    if ((code >> 31) != 0) {
      const int pi = (int)((code >> 6) & 0x1ffffffu);
      const ShapeDev<T>* h1 = S.rshapes + S.spairs[4 * pi];
      const ShapeDev<T>* h2 = S.rshapes + S.spairs[4 * pi + 1];
      A = shape_ref<T>(h1);
      B = shape_ref<T>(h2);
      wa = pose_mul(nodes.get(h1->node), shape_pose<T>(h1));
      wb = pose_mul(nodes.get(h2->node), shape_pose<T>(h2));
    } else {
      const ShapeDev<T>* h1 = S.rshapes + (int)((code >> 6) & 1023u);
      const ShapeDev<T>* h2 = S.oshapes + (int)((code >> 16) & 32767u);
      A = shape_ref<T>(h1);
      B = shape_ref<T>(h2);
      wa = pose_mul(nodes.get(h1->node), shape_pose<T>(h1));
      wb = shape_pose<T>(h2);
    }
  };
  // survivors of the culls wait in the queue for P3, where they spread over all lanes.  When the queue is full
  // the workgroup notes it and, after P3, walks the broadphase once more with every survivor decided on the
  // spot (see the end of the kernel): rare, slow, and out of the way of the instruction stream that matters.
  auto push = [&](unsigned code) {
    if (GEAR_DBG(what, 0x200u)) return;  // profiling ablation (tools/ablate_small.py): culls run, nothing is decided
    const unsigned at = atomicAdd(&qfill[0], 1u);
    if (at < qcap) queue[at] = code;
  };
  auto decide_now = [&](unsigned code) {
    const int c = (int)(code & 63u);
    const unsigned bit = (code >> 31) ? 2u : 1u;
    if (cflag[c] & bit) return;  // already known to collide: the flags are an OR
    ShapeRef<T> A, B;
    Pose<T> wa, wb;
    fetch(code, A, wa, B, wb);
    if (pair_within<T, false>(A, wa, B, wb, S.verts, S.planes, margin, lc)) atomicOr(&cflag[c], bit);
  };

  // ---- P2: broadphase: a lane per (configuration, shape), then a lane per (configuration, self pair) ------
  auto broadphase = [&](auto&& sink) {
    const uint4* __restrict__ blocks = reinterpret_cast<const uint4*>(S.cell_blk);
    const Quad<T>* __restrict__ ocull4 = reinterpret_cast<const Quad<T>*>(S.ocull);
    const unsigned sentinel = (unsigned)S.n_obs;
    for (int u = tid; u < n_env + n_self; u += kSmThreads) {
      if (u < n_env) {
        // eight lanes per (configuration, shape): lane j looks at slot j of the cell's block(s)
        const int j = u % kSmSlotLanes, v = u / kSmSlotLanes;
        const int c = v % Cl, s = v / Cl;
        if (cbad[c]) continue;
        const EnvUnit e = u == tid ? eu0 : load_env(s);
        const Pose<T> ws = pose_mul(PoseRecs<T>{node_base, C, c}.get(e.node), e.lp);
        unsigned blk = 0;  // block 0 is the empty list
        if (all_blk >= 0) {
          blk = (unsigned)all_blk;  // GH_PATH_NO_GRID: the chain that lists every obstacle
        } else {
          const T fx = (ws.t.x - e.org[0]) * e.inv_h, fy = (ws.t.y - e.org[1]) * e.inv_h, fz = (ws.t.z - e.org[2]) * e.inv_h;
          const bool inside = fx >= T(0) && fy >= T(0) && fz >= T(0) && fx < T(e.nx) && fy < T(e.ny) && fz < T(e.nz);
          if (inside) blk = (unsigned)e.off_base + ((unsigned)(int)fz * (unsigned)e.ny + (unsigned)(int)fy) * (unsigned)e.nx + (unsigned)(int)fx;
        }
        const T rho = e.br;
        const unsigned code_base = ((unsigned)s << 6) | (unsigned)c;
        // the shape's oriented box, as in k_sweep_n1
        const bool is_ball = e.kind == SK_BALL;
        const T pad = T(1e-5) * (T(1) + rho);
        const T hbx = e.half[0] + pad, hby = (is_ball ? e.half[0] : e.half[1]) + pad, hbz = (is_ball ? e.half[0] : e.half[2]) + pad;
        while (blk != 0) {
          const uint4 w = blocks[blk];
          const bool cont = (w.w >> 31) != 0;
          const unsigned wz = j < 4 ? (j < 2 ? w.x : w.y) : (j < 6 ? w.z : w.w);
          unsigned id = (wz >> ((j & 1) << 4)) & 0xffffu;
          if (cont && j >= 6) id = sentinel;  // a chained block: slots 6 and 7 hold the link
          blk = cont ? (w.w & 0x7fffffffu) : 0u;
          if (id >= sentinel) continue;
          const Quad<T> o = cull_lds ? lcull[id] : ocull4[id];
          const T dx = o.x - ws.t.x, dy = o.y - ws.t.y, dz = o.z - ws.t.z;
          const T rr = rho + o.w;
          if (!(dx * dx + dy * dy + dz * dz < rr * rr)) continue;
          // R^T d = d + w t' + u x t' with u = -(i, j, k), t' = 2 u x d
          const T ui = -ws.r.i, uj = -ws.r.j, uk = -ws.r.k;
          const T tx = T(2) * (uj * dz - uk * dy), ty = T(2) * (uk * dx - ui * dz), tz = T(2) * (ui * dy - uj * dx);
          const T px = dx + ws.r.w * tx + (uj * tz - uk * ty), py = dy + ws.r.w * ty + (uk * tx - ui * tz), pz = dz + ws.r.w * tz + (ui * ty - uj * tx);
          const T ex = t_max(t_abs(px) - hbx, T(0)), ey = t_max(t_abs(py) - hby, T(0)), ez = t_max(t_abs(pz) - hbz, T(0));
          if (ex * ex + ey * ey + ez * ez < o.w * o.w) sink((id << 16) | code_base);
        }
      } else {
        const int v = u - n_env;
        const int c = v % Cl, p = v / Cl;
        if (cbad[c]) continue;
        const SelfUnit su = u == tid ? su0 : load_self(p);
        const PoseRecs<T> nodes{node_base, C, c};
        const V3<T> c1 = pose_mul(nodes.get(su.node1), su.lp1).t;
        const V3<T> c2 = pose_mul(nodes.get(su.node2), su.lp2).t;
        const V3<T> d = c1 - c2;
        if (dot(d, d) < su.rr * su.rr) sink(kSelfItem | ((unsigned)p << 6) | (unsigned)c);
      }
    }
  };
  broadphase(push);
  __syncthreads();
  GEAR_SMALL_STAMP();

  // ---- P3a: stage 1 of the narrow phase (closed forms, SAT), one item per lane ------------------------------
  // Stage 2 waits behind a barrier: most configurations that collide at all are found out here, by a cheap
  // item, and the expensive items of such a configuration are then never run (the flags are an OR).
  {
    const unsigned pushed = qfill[0];
    const unsigned cnt = pushed < qcap ? pushed : qcap;
    if (!GEAR_DBG(what, 0x100u))  // profiling ablation: queued items are dropped
      for (unsigned k = (unsigned)tid; k < cnt; k += kSmThreads) {
        const unsigned code = queue[k];
        const int c = (int)(code & 63u);
        const unsigned bit = (code >> 31) ? 2u : 1u;
        if (cflag[c] & bit) continue;
        ShapeRef<T> A, B;
        Pose<T> wa, wb;
        fetch(code, A, wa, B, wb);
        const int r = pair_stage1<T, false>(A, wa, B, wb, margin, lc);
        if (r == PR_HIT) atomicOr(&cflag[c], bit);
        else if (r == PR_EXACT) xqueue[atomicAdd(&qfill[2], 1u)] = code;  // at most cnt <= qcap entries
      }
  }
  __syncthreads();
  GEAR_SMALL_STAMP();

  // ---- P3b: stage 2 (GJK margin query, hollow-mesh rule) for what is still undecided ---------------------------
  {
    const unsigned cnt = qfill[2];
    for (unsigned k = (unsigned)tid; k < cnt; k += kSmThreads) {
      const unsigned code = xqueue[k];
      const int c = (int)(code & 63u);
      const unsigned bit = (code >> 31) ? 2u : 1u;
      if (cflag[c] & bit) continue;
      ShapeRef<T> A, B;
      Pose<T> wa, wb;
      fetch(code, A, wa, B, wb);
      if (pair_stage2<T, false>(A, wa, B, wb, S.verts, S.planes, margin, lc)) atomicOr(&cflag[c], bit);
    }
  }
  __syncthreads();
  GEAR_SMALL_STAMP();
  if (qfill[0] > qcap) {  // the queue ran full (uniform): every survivor once more, decided by the lane that finds it
    broadphase(decide_now);
    __syncthreads();
  }
#ifdef GEAR_HIP_ABLATE
  if ((what & 0x10000000u) && tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2))
    printf("k_small wg %u: P0 %llu P1 %llu P2 %llu P3a %llu P3b %llu (x10 ns)  items %u exact %u\n", blockIdx.x, stamp[1] - stamp[0],
           stamp[2] - stamp[1], stamp[3] - stamp[2], stamp[4] - stamp[3], stamp[5] - stamp[4], qfill[0], qfill[2]);
#endif

  // ---- P4: flag bytes -------------------------------------------------------------------------------------
  if (tid < Cl) {
    const unsigned out = cbad[tid] ? 4u : (cflag[tid] & 3u);
    flags[c0 + (size_t)tid] = (uint8_t)out;
    if (flags_near) flags_near[c0 + (size_t)tid] = (uint8_t)out;
  }
  if (first_bad == nullptr && notify == nullptr) return;

  // ---- P5: the last workgroup to get here (a ticket from *done) finishes the call ---------------------------
  // gh_check_edges: it reduces the flags of the whole call to every edge's first flagged sample.
  // notify (host-buffer calls, whose flags and first-bad indices go straight to pinned host memory): when
  // everything is written and visible to the host it stores the call's sequence number in a host word that
  // the calling thread polls — that costs 5 us less per call than waiting for the stream's completion signal
  // (tools/sync_vs_poll.hip).  Ordering: every workgroup fences its flag bytes at SYSTEM scope before taking
  // its ticket; the last one fences again after the ticket (and after the reduction's stores) and only then
  // releases the sequence number.
  if (notify) __threadfence_system();
  else __threadfence();  // this workgroup's flag bytes are visible device-wide before its ticket is
  __syncthreads();
  if (tid == 0) {
    const unsigned ticket = atomicAdd(done, 1u);
    qfill[1] = ticket == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (qfill[1] == 0) return;
  if (notify) __threadfence_system();
  else __threadfence();  // ... and the other workgroups' are visible here
  if (first_bad != nullptr) {
    const volatile uint8_t* vf = flags_near ? flags_near : flags;
    if ((size_t)n_edges * 16 <= n) {
      // long edges (connect lines): a wavefront per edge, 64 samples per step, stop at the first step with a hit
      const unsigned lane = (unsigned)tid & 63u;
      for (unsigned e = (unsigned)tid >> 6; e < n_edges; e += kSmThreads / 64) {
        const unsigned lo = edge_off[e], hi = edge_off[e + 1];
        int32_t fb = -1;
        for (unsigned k0 = lo; k0 < hi; k0 += 64) {
          const unsigned k = k0 + lane;
          const unsigned long long hb = __ballot(k < hi && (vf[k] & bad_mask) != 0);
          if (hb != 0) { fb = (int32_t)(k0 - lo) + __builtin_ctzll(hb); break; }
        }
        if (lane == 0) first_bad[e] = fb;
      }
    } else {
      for (unsigned e = (unsigned)tid; e < n_edges; e += kSmThreads) {
        const unsigned lo = edge_off[e], hi = edge_off[e + 1];
        int32_t fb = -1;
        for (unsigned k = lo; k < hi; k++)
          if (vf[k] & bad_mask) { fb = (int32_t)(k - lo); break; }
        first_bad[e] = fb;
      }
    }
  }
  if (tid == 0) *done = 0u;  // ready for the next call
  if (notify) {
    __threadfence_system();  // every lane's first-bad stores have reached the host's memory
    __syncthreads();
    if (tid == 0) __hip_atomic_store(notify, notify_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

#undef GEAR_SMALL_STAMP

}  // namespace gear_hip
