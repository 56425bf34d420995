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
//   P2  one lane per (configuration, shape):  world pose of the shape, grid cell, candidate
//                                             obstacles, bounding-sphere cull, oriented-box cull; survivors
//                                             are pushed to the workgroup's item queue in LDS
//       one lane per (configuration, self shape pair): centre cull, survivors pushed likewise
//   P3  one lane per queued item:             closed forms / SAT, then GJK margin query + hollow-mesh rule
//                                             (pair_within); a hit is OR-ed into the configuration's flags
//   P4  one lane per configuration:           the flag byte
//   P5  (gh_check_edges) the last workgroup:  first flagged sample of every edge
//
// A workgroup takes as many configurations as let each phase finish in one pass of its 256 lanes
// (plan_small), so a call is five short chains of dependent memory trips, not five per shape.  An item that
// finds the queue full is decided in place by the lane that found it: the result never depends on the
// queue's capacity.
//
// The arithmetic is the large-batch kernels': the same t_sincos / fk_step_sc / pose_mul sequence per node
// and shape (bit-identical poses) and the same pair_stage1 / pair_stage2 per pair; the culls in front of
// them are conservative filters, so the flags equal k_check's bit for bit (tests/test_gpu_small.py).
#pragma once
#include "gear_pipeline.hpp"

namespace gear_hip {

constexpr int kSmThreads = 256;
constexpr int kSmMaxC = 64;  // configurations per workgroup (the item codes keep 6 bits for it)

struct SmallPlan {
  int C = 0;            // configurations per workgroup
  unsigned qcap = 0;    // item queue entries
  size_t lds = 0;       // dynamic LDS bytes
};

// C: as many configurations as let every phase finish in one pass of the workgroup's 256 lanes — P2 has a
// lane per (configuration, shape) AND per (configuration, self shape pair), P0 a lane per (configuration,
// node) and per (configuration, limit).  Small batches leave the chip empty, so more, smaller workgroups cost
// nothing, while a second pass of a phase is a second chain of dependent memory trips.
template <class T> inline SmallPlan plan_small(int n_nodes, int n_shapes, int n_spairs, int n_limits, int n_slots, unsigned qcap) {
  SmallPlan p;
  const int per_config = std::max({n_shapes + n_spairs, n_nodes + n_limits, 1});
  const int c = kSmThreads / per_config;
  p.C = c < 1 ? 1 : (c > kSmMaxC ? kSmMaxC : c);
  p.qcap = qcap;
  const size_t poses = (size_t)(n_slots + n_nodes) * 7 * p.C * sizeof(T);
  p.lds = (poses + 15) / 16 * 16 + (size_t)(2 * kSmMaxC + 4 + qcap) * 4;
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
        const uint32_t* __restrict__ edge_off, unsigned n_edges, unsigned bad_mask, int32_t* __restrict__ first_bad,
        uint8_t* flags_near, unsigned* done) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gear_lds[];
  const Views<T> V(S);
  const int tid = threadIdx.x;
  const size_t c0 = (size_t)blockIdx.x * (size_t)C;
  const int Cl = (int)(n - c0 < (size_t)C ? n - c0 : (size_t)C);  // configurations of this workgroup (>= 1)
  const int N = S.n_nodes, L = S.n_shapes;

  // LDS: [branch slots][node poses] as columns of C, flag words, limit words, queue fill, queue
  T* const slot_base = reinterpret_cast<T*>(gear_lds);
  T* const node_base = slot_base + (size_t)S.n_slots * 7 * C;
  const size_t t_bytes = ((size_t)(S.n_slots + N) * 7 * C * sizeof(T) + 15) / 16 * 16;
  unsigned* const cflag = reinterpret_cast<unsigned*>(gear_lds + t_bytes);  // [64] bit0 env, bit1 self
  unsigned* const cbad = cflag + kSmMaxC;                                     // [64] != 0: outside a limit
  unsigned* const qfill = cbad + kSmMaxC;                                     // [4] queue fill, last-workgroup flag
  unsigned* const queue = qfill + 4;                                          // [qcap]

  const bool do_env = (what & 1u) && S.n_obs > 0;
  const bool do_self = (what & 2u) && S.n_spairs > 0;
  const bool do_limits = (what & kWhatNoLimits) == 0;  // GH_CHECK_NO_LIMITS: geometry only
  const T margin = S.margin;
  LaneCounters lc{0, 0, 0, 0};

  if (tid < kSmMaxC) { cflag[tid] = 0; cbad[tid] = 0; }
  if (tid == 0) qfill[0] = 0;
  __syncthreads();

  // ---- P0: every node's transform relative to its parent; every limit ------------------------------------
  {
    const int n_loc = (do_env || do_self) ? Cl * N : 0;
    const int n_lim = do_limits ? Cl * S.n_limits : 0;
    for (int u = tid; u < n_loc + n_lim; u += kSmThreads) {
      if (u < n_loc) {
        const int c = u % Cl, k = u / Cl;
        const NodeDev<T>* nd = S.nodes + k;
        const T* qrow = q + (c0 + (size_t)c) * (size_t)S.n_cols;
        T pos = nd->const_pos;  // node_position, with a lane-varying node
        const int col = nd->col;
        if (col >= 0) pos = joint_value(qrow[col], nd->direct, nd->mult, nd->off);
        const V3<T> ot{nd->ot[0], nd->ot[1], nd->ot[2]}, axis{nd->axis[0], nd->axis[1], nd->axis[2]};
        const Q4<T> oq{nd->oq[0], nd->oq[1], nd->oq[2], nd->oq[3]};
        T s = T(0), cs = T(1);
        if (nd->joint_type == JT_ROT) t_sincos(pos / T(2), &s, &cs);
        PoseStore<T>{node_base, (size_t)C, (size_t)c}.put(k, fk_local_sc(ot, oq, axis, nd->joint_type, pos, s, cs));
      } else {
        const int v = u - n_loc;
        const int c = v % Cl, l = v / Cl;
        const LimitDev<T>* lim = S.limits + l;
        const T* qrow = q + (c0 + (size_t)c) * (size_t)S.n_cols;
        const T val = joint_value(qrow[lim->col], lim->direct, lim->mult, lim->off);
        if (!(val <= lim->hi && val >= lim->lo)) cbad[c] = 1u;  // Range::is_valid, as limits_ok spells it
      }
    }
  }
  __syncthreads();

  // ---- P1: the FK walk, one lane per configuration: world = parent * local, root first ------------------
  if (tid < Cl && (do_env || do_self) && !cbad[tid]) {  // the reference evaluates no collision after a limit error
    const int c = tid;
    const PoseStore<T> slots{slot_base, (size_t)C, (size_t)c};
    const PoseStore<T> nodes{node_base, (size_t)C, (size_t)c};
    Pose<T> cur{{T(0), T(0), T(0)}, {T(0), T(0), T(0), T(1)}};
    for (int k = 0; k < N; k++) {
      const GEAR_CAS NodeDev<T>* nd = V.nodes + k;
      Pose<T> parent = cur;
      const int ps = nd->parent_slot;
      if (ps == PARENT_IDENTITY) parent = {{T(0), T(0), T(0)}, {T(0), T(0), T(0), T(1)}};
      else if (ps >= 0) parent = slots.get(ps);
      cur = pose_mul(parent, nodes.get(k));
      const int ss = nd->save_slot;
      if (ss >= 0) slots.put(ss, cur);
      if (nd->shape_end > nd->shape_begin) nodes.put(k, cur);
    }
  }
  __syncthreads();

  // One item: both shapes, both poses (node pose from LDS times the shape's local pose: the very product the
  // walk of the large-batch kernels forms), both stages.  Codes as in the pipeline, with the configuration's
  // index inside the workgroup in the low 6 bits.  Needs nothing but the node poses, so it can run anywhere
  // after P1.
  auto decide = [&](unsigned code) {
    const int c = (int)(code & 63u);
    const bool sf = (code >> 31) != 0;
    const unsigned bit = sf ? 2u : 1u;
    if (cflag[c] & bit) return;  // already known to collide: the flags are an OR
    const PoseStore<T> nodes{node_base, (size_t)C, (size_t)c};
    ShapeRef<T> A, B;
    Pose<T> wa, wb;
    if (sf) {
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
    if (pair_within<T, false>(A, wa, B, wb, S.verts, S.planes, margin, lc)) atomicOr(&cflag[c], bit);
  };
  // survivors of the culls wait in the queue for P3, where they spread over all lanes; one that finds the queue
  // full is decided on the spot by the lane that found it
  auto push = [&](unsigned code) {
    const unsigned at = atomicAdd(&qfill[0], 1u);
    if (at < qcap) queue[at] = code;
    else decide(code);
  };

  // ---- P2: broadphase: a lane per (configuration, shape), then a lane per (configuration, self pair) ------
  {
    const uint4* __restrict__ blocks = reinterpret_cast<const uint4*>(S.cell_blk);
    const Cull4<T>* __restrict__ ocull = reinterpret_cast<const Cull4<T>*>(S.ocull);
    const unsigned sentinel = (unsigned)S.n_obs;
    const int n_env = do_env ? Cl * L : 0;
    const int n_self = do_self ? Cl * S.n_spairs : 0;
    for (int u = tid; u < n_env + n_self; u += kSmThreads) {
      if (u < n_env) {
        const int c = u % Cl, s = u / Cl;
        if (cbad[c]) continue;
        const ShapeDev<T>* sh = S.rshapes + s;
        const Pose<T> ws = pose_mul(PoseStore<T>{node_base, (size_t)C, (size_t)c}.get(sh->node), shape_pose<T>(sh));
        unsigned blk = 0;  // block 0 is the empty list
        if (all_blk >= 0) {
          blk = (unsigned)all_blk;  // GH_PATH_NO_GRID: the chain that lists every obstacle
        } else {
          const GridDev<T>* g = S.grids + S.shape_grid[s];
          const T fx = (ws.t.x - g->org[0]) * g->inv_h, fy = (ws.t.y - g->org[1]) * g->inv_h, fz = (ws.t.z - g->org[2]) * g->inv_h;
          const bool inside = fx >= T(0) && fy >= T(0) && fz >= T(0) && fx < T(g->nx) && fy < T(g->ny) && fz < T(g->nz);
          if (inside) blk = (unsigned)g->off_base + ((unsigned)(int)fz * (unsigned)g->ny + (unsigned)(int)fy) * (unsigned)g->nx + (unsigned)(int)fx;
        }
        const T rho = sh->br;
        const unsigned code_base = ((unsigned)s << 6) | (unsigned)c;
        // the shape's oriented box, as in k_sweep_n1
        const bool is_ball = sh->kind == SK_BALL;
        const T pad = T(1e-5) * (T(1) + rho);
        const T hbx = sh->half[0] + pad, hby = (is_ball ? sh->half[0] : sh->half[1]) + pad, hbz = (is_ball ? sh->half[0] : sh->half[2]) + pad;
        while (blk != 0) {
          const uint4 w = blocks[blk];
          const bool cont = (w.w >> 31) != 0;
          unsigned id[8] = {w.x & 0xffffu, w.x >> 16, w.y & 0xffffu, w.y >> 16, w.z & 0xffffu, w.z >> 16, w.w & 0xffffu, w.w >> 16};
          if (cont) id[6] = id[7] = sentinel;
          Cull4<T> oc[8];
#pragma unroll
          for (int j = 0; j < 8; j++) oc[j] = ocull[id[j]];  // eight independent loads; a sentinel reads the far dummy
#pragma unroll
          for (int j = 0; j < 8; j++) {
            const T dx = oc[j].x - ws.t.x, dy = oc[j].y - ws.t.y, dz = oc[j].z - ws.t.z;
            const T rr = rho + oc[j].r;
            if (!(dx * dx + dy * dy + dz * dz < rr * rr)) continue;
            // R^T d = d + w t' + u x t' with u = -(i, j, k), t' = 2 u x d
            const T ui = -ws.r.i, uj = -ws.r.j, uk = -ws.r.k;
            const T tx = T(2) * (uj * dz - uk * dy), ty = T(2) * (uk * dx - ui * dz), tz = T(2) * (ui * dy - uj * dx);
            const T px = dx + ws.r.w * tx + (uj * tz - uk * ty), py = dy + ws.r.w * ty + (uk * tx - ui * tz), pz = dz + ws.r.w * tz + (ui * ty - uj * tx);
            const T ex = t_max(t_abs(px) - hbx, T(0)), ey = t_max(t_abs(py) - hby, T(0)), ez = t_max(t_abs(pz) - hbz, T(0));
            if (ex * ex + ey * ey + ez * ez < oc[j].r * oc[j].r) push((id[j] << 16) | code_base);
          }
          blk = cont ? (w.w & 0x7fffffffu) : 0u;
        }
      } else {
        const int v = u - n_env;
        const int c = v % Cl, p = v / Cl;
        if (cbad[c]) continue;
        const PoseStore<T> nodes{node_base, (size_t)C, (size_t)c};
        const ShapeDev<T>* h1 = S.rshapes + S.spairs[4 * p];
        const ShapeDev<T>* h2 = S.rshapes + S.spairs[4 * p + 1];
        const V3<T> c1 = pose_mul(nodes.get(h1->node), shape_pose<T>(h1)).t;
        const V3<T> c2 = pose_mul(nodes.get(h2->node), shape_pose<T>(h2)).t;
        const V3<T> d = c1 - c2;
        const T rr = S.spair_r[p];
        if (dot(d, d) < rr * rr) push(kSelfItem | ((unsigned)p << 6) | (unsigned)c);
      }
    }
  }
  __syncthreads();

  // ---- P3: the narrow phase, one item per lane ------------------------------------------------------------
  {
    const unsigned pushed = qfill[0];
    const unsigned cnt = pushed < qcap ? pushed : qcap;
    for (unsigned k = (unsigned)tid; k < cnt; k += kSmThreads) decide(queue[k]);
  }
  __syncthreads();

  // ---- P4: flag bytes -------------------------------------------------------------------------------------
  if (tid < Cl) {
    const unsigned out = cbad[tid] ? 4u : (cflag[tid] & 3u);
    flags[c0 + (size_t)tid] = (uint8_t)out;
    if (flags_near) flags_near[c0 + (size_t)tid] = (uint8_t)out;
  }
  if (first_bad == nullptr) return;

  // ---- P5 (gh_check_edges): the last workgroup to get here reduces the flags of the whole call ------------
  __threadfence();  // this workgroup's flag bytes are visible device-wide before its ticket is
  __syncthreads();
  if (tid == 0) {
    const unsigned ticket = atomicAdd(done, 1u);
    qfill[1] = ticket == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (qfill[1] == 0) return;
  __threadfence();  // ... and the other workgroups' are visible here
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
  if (tid == 0) *done = 0u;  // ready for the next call
}

}  // namespace gear_hip
