// gear_pipeline.hpp — the large-batch form of the check: three lean gfx950 kernels in
// front of the fused k_check, so that every stage runs at the occupancy and register
// budget it can use (the fused kernel carries the narrow phase's ~200 VGPRs through
// its sweep and is limited to two waves per SIMD by its LDS pose columns).
//
//   k_sweep    one lane per configuration: limits, FK walk with the world poses of up
//              to K robot shapes held in REGISTERS; whenever K shapes are ready they
//              are swept against all obstacle records (obstacle block outer, shapes
//              inner, records as SGPR operands).  Survivors are appended, pose
//              included, to the wave's own region of a global queue (no atomics: one
//              region per wave).  With self pairs: every shape pose also goes to a
//              global pose store, shape centres to LDS, and the self shape pairs are
//              swept from there.
//   k_narrow1  workgroup b = the items of sweep wave b, one item per lane: closed
//              forms / SAT / box prefilter; hits OR-ed in LDS; undecided items are
//              appended to one dense exact queue (one atomic per workgroup).
//   k_narrow2  one lane per exact item: GJK margin query + hollow-mesh rule.
//   k_check<REDO>  hit word -> flag byte; a wave whose queue region overflowed (or an
//              exact queue that ran full) is re-checked by the fused path, so the
//              result never depends on a queue capacity.
//
// Reference semantics are unchanged: per configuration this evaluates is_feasible /
// is_feasible_with_self (src/planner/joint_path_planner.rs:71-84,113-121); only the
// order in which pairs are looked at differs, and the flags are an OR over pairs.
#pragma once
#include "gear_kernels.hpp"

namespace gear_hip {

constexpr int kExLds = 512;           // per-workgroup staging of exact entries in k_narrow1
constexpr int kItemBits = 13;          // an exact entry packs wave << kItemBits | item index
constexpr int kMaxItemCap = 1 << kItemBits;
constexpr int kMaxPipeWaves = 1 << (32 - kItemBits);
constexpr int kClassBits = 4;          // pair classes of the exact stage (exact_class)
constexpr int kClasses = 1 << kClassBits;
constexpr int kSubs = 16;              // queue segments per class (k_narrow1 workgroup b appends to sub-segment b % kSubs)

// Pair class of an undecided item: which support functions and which parts of the hollow-mesh rule its
// exact query runs.  The exact queue keeps one set of segments per class, so the lanes of a k_narrow2
// wavefront take the same branches (a wavefront of mixed items executes the union of all their paths,
// and with one or two waves per SIMD nothing hides that).
GD unsigned exact_class(int kind_a, int kind_b, bool self) {
  // kinds: box 0, ball 1, mesh 2.  box-box 0, box-mesh 1, ball-mesh 2, mesh-box 3, mesh-ball 4, mesh-mesh 5
  // (ball-ball and box-ball are decided in stage 1: 6); self pairs + 8
  const unsigned long long lut = 0x543266160ull;
  return (unsigned)((lut >> (4 * (kind_a * 3 + kind_b))) & 15u) + (self ? 8u : 0u);
}

// self item whose two poses are in its own and the next record: kSelfInline | s1 << 16 | s2 << 6 | lane (shapes < 1024)
constexpr unsigned kSelfInline = 0xc0000000u;
constexpr unsigned kItemSkip = 0xffffffffu;    // second record of such an item: not an item

struct WaveHdr {
  unsigned count;     // items pushed by the wave (may exceed the capacity: then `overflow` is set)
  unsigned limit_lo;  // lanes whose configuration violates a joint limit
  unsigned limit_hi;
  unsigned overflow;
};

constexpr int kSegs = kClasses * kSubs;  // segments of the exact queue: segment = class * kSubs + sub

struct PipeCounters {
  // The exact queue is kSegs independent segments, each with its own fill counter in its own 128-byte
  // line: k_narrow1's workgroups append with one atomic per class they hold, all in one trip, and a
  // single counter for all of them serialised the kernel (0.1 ms per 1M configurations, measured).
  unsigned seg[kSegs][32];  // [s][0] = fill of segment s
  unsigned exact;     // unused
  unsigned overflow;  // unused
  unsigned long long candidates;  // broadphase candidates tested (only with GH_COUNT_CANDIDATES in `what`)
  unsigned hist[2][48];  // GEAR_HIP_N2_HIST=1 (profiling): GJK iterations per exact item / the largest count in a wavefront
};

// layout of one launch chunk's workspace
template <class T> struct PipeWs {
  WaveHdr* hdr;          // [waves]
  T* items;              // [waves][cap] records of 8 T: {code, world pose (t, q) of an env item's robot shape}
  unsigned* hit;         // [configs]
  unsigned* exact;       // [kSegs][cap2]  wave << kItemBits | item
  PipeCounters* pc;
  T* pstore;             // [L][7][stride]  every shape pose (self pairs only)
  unsigned long long* stamps;  // [waves][8] cycle stamps of k_narrow1's phases (GEAR_HIP_STAMPS=1, profiling), else null
  size_t stride;
  unsigned cap, cap2;    // items per queue region; entries per exact-queue segment
};

// One survivor = one 8-element record (32 B in f32, 64 B in f64), written and read with 16-byte
// accesses: consecutive queue slots are consecutive records, so both the sweep's pushes and the narrow
// phase's fetches touch whole 32-byte sectors.  Element 0 carries the item code (bit pattern); a self
// item has only its code (its two poses come from the pose store).
template <class T> struct ItemIO;
template <> struct ItemIO<float> {
  struct Rec { uint4 a, b; };  // a record in registers
  GD static Rec load(const float* rec) { const uint4* r = reinterpret_cast<const uint4*>(rec); return {r[0], r[1]}; }
  GD static unsigned code(const Rec& r) { return r.a.x; }
  GD static Pose<float> pose(const Rec& r) {
    return {{__uint_as_float(r.a.y), __uint_as_float(r.a.z), __uint_as_float(r.a.w)},
            {__uint_as_float(r.b.x), __uint_as_float(r.b.y), __uint_as_float(r.b.z), __uint_as_float(r.b.w)}};
  }
  GD static void store(float* rec, unsigned code, const Pose<float>& p) {
    uint4* r = reinterpret_cast<uint4*>(rec);
    r[0] = make_uint4(code, __float_as_uint(p.t.x), __float_as_uint(p.t.y), __float_as_uint(p.t.z));
    r[1] = make_uint4(__float_as_uint(p.r.i), __float_as_uint(p.r.j), __float_as_uint(p.r.k), __float_as_uint(p.r.w));
  }
  GD static void store_code(float* rec, unsigned code) { *reinterpret_cast<unsigned*>(rec) = code; }
  GD static unsigned code(const float* rec) { return *reinterpret_cast<const unsigned*>(rec); }
  GD static Pose<float> pose(const float* rec) {
    const uint4* r = reinterpret_cast<const uint4*>(rec);
    const uint4 a = r[0], b = r[1];
    return {{__uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w)},
            {__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w)}};
  }
};
template <> struct ItemIO<double> {
  struct Rec { double2 a, b, c, d; };
  GD static Rec load(const double* rec) { const double2* r = reinterpret_cast<const double2*>(rec); return {r[0], r[1], r[2], r[3]}; }
  GD static unsigned code(const Rec& r) { return (unsigned)__double_as_longlong(r.a.x); }
  GD static Pose<double> pose(const Rec& r) { return {{r.a.y, r.b.x, r.b.y}, {r.c.x, r.c.y, r.d.x, r.d.y}}; }
  GD static void store(double* rec, unsigned code, const Pose<double>& p) {
    double2* r = reinterpret_cast<double2*>(rec);
    r[0] = make_double2(__longlong_as_double((long long)code), p.t.x);
    r[1] = make_double2(p.t.y, p.t.z);
    r[2] = make_double2(p.r.i, p.r.j);
    r[3] = make_double2(p.r.k, p.r.w);
  }
  GD static void store_code(double* rec, unsigned code) { *reinterpret_cast<unsigned*>(rec) = code; }
  GD static unsigned code(const double* rec) { return *reinterpret_cast<const unsigned*>(rec); }
  GD static Pose<double> pose(const double* rec) {
    const double2* r = reinterpret_cast<const double2*>(rec);
    const double2 a = r[0], b = r[1], c = r[2], d = r[3];
    return {{a.y, b.x, b.y}, {c.x, c.y, d.x, d.y}};
  }
};

// The first workgroup of a chunk's sweep clears the exact queue's fill counters (k_narrow1, the next
// kernel on the stream, is their first user): no separate memset launch per chunk.
__device__ __forceinline__ void clear_segment_counters(PipeCounters* pc) {
  if (blockIdx.x == 0)
    for (int k = threadIdx.x; k < kSegs; k += blockDim.x) pc->seg[k][0] = 0;
}

__device__ __forceinline__ unsigned lane_rank(unsigned long long b) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
}

// exclusive prefix sum of a small per-lane count over the wave, and the wave total
__device__ __forceinline__ void wave_offsets(unsigned nl, unsigned& off, unsigned& tot) {
  const unsigned long long bh = __ballot(nl >= 8u);
  if (bh == 0) {
    const unsigned long long b0 = __ballot((nl & 1u) != 0), b1 = __ballot((nl & 2u) != 0), b2 = __ballot((nl & 4u) != 0);
    off = lane_rank(b0) + 2u * lane_rank(b1) + 4u * lane_rank(b2);
    tot = (unsigned)__popcll(b0) + 2u * (unsigned)__popcll(b1) + 4u * (unsigned)__popcll(b2);
  } else {
    unsigned inc = nl;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned o = __shfl_up(inc, d);
      if (lane >= d) inc += o;
    }
    off = inc - nl;
    tot = __shfl(inc, 63);
  }
}

// ---------------------------------------------------------------------------
// k_sweep
// ---------------------------------------------------------------------------
template <class T, int K>
__global__ void __launch_bounds__(kWave)
k_sweep(SceneDev<T> S, const T* __restrict__ q, size_t n, uint32_t what, PipeWs<T> W, int lds_centres, int max_node) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gear_lds[];
  const Views<T> V(S);
  clear_segment_counters(W.pc);
  const int lane = threadIdx.x;
  const size_t wave = blockIdx.x;
  const size_t col = wave * kWave + lane;
  const bool live = col < n;
  const size_t i = live ? col : n - 1;
  const T* qrow = q + i * (size_t)S.n_cols;
  const PoseStore<T> slots{reinterpret_cast<T*>(gear_lds), (size_t)kWave, (size_t)lane};
  // half-angle sin/cos of every node, computed ahead of the walk: the f64-rounded sincos of the f32
  // path needs far more registers than the walk, and must not be live together with the K poses
  T* sc = reinterpret_cast<T*>(gear_lds + (size_t)S.n_slots * 7 * kWave * sizeof(T)) + lane;  // [N][2][64]
  T* cen = sc - lane + (size_t)(max_node + 1) * 2 * kWave;  // [L][3][64] when lds_centres

  const bool ok = (what & kWhatNoLimits) != 0 || limits_ok(S, V, qrow);  // GH_CHECK_NO_LIMITS: geometry only
  const bool do_env = (what & 1u) && S.n_obs > 0;
  const bool do_self = (what & 2u) && S.n_spairs > 0;
  const bool push_ok = live && ok;  // the reference evaluates no collision after a limit error
  const bool dbg_no_sweep = GEAR_DBG(what, 0x800u), dbg_no_push = GEAR_DBG(what, 0x400u);  // profiling ablations

  T* items = W.items + wave * 8 * (size_t)W.cap;
  unsigned count = 0;  // wave-uniform
  bool overflow = false;

  const T never = T(1e30);  // a0 of a lane / register slot that must not produce survivors
  Pose<T> P[K];
  T a0[K], rho[K];
#pragma unroll
  for (int j = 0; j < K; j++) {
    P[j] = {{T(0), T(0), T(0)}, {T(0), T(0), T(0), T(1)}};
    a0[j] = never;
    rho[j] = T(0);
  }
  int fill = 0, node = -1;
  Pose<T> cur{{T(0), T(0), T(0)}, {T(0), T(0), T(0), T(1)}};
  const int n_walk = (do_env || do_self) ? S.n_shapes : 0;
  const bool lazy_store = do_self && lds_centres && S.n_shapes <= K;  // P[j] is shape j's pose after the walk
  if (n_walk > 0) {
    for (int k = 0; k <= max_node; k++) {
      const GEAR_CAS NodeDev<T>* nd = V.nodes + k;
      if (nd->joint_type != JT_ROT) continue;
      T sn, cs;
      t_sincos(node_position(nd, qrow) / T(2), &sn, &cs);
      sc[(2 * k) * kWave] = sn;
      sc[(2 * k + 1) * kWave] = cs;
    }
  }

  for (int s = 0; s < n_walk; s++) {
    const GEAR_CAS ShapeDev<T>* sh = V.rshapes + s;
    const int target = sh->node;
    while (node < target) {
      node++;
      node_world<T, true>(V, node, qrow, slots, cur, sc);
    }
    const Pose<T> ws = pose_mul(cur, shape_pose<T>(sh));
    if (do_self) {
      // a robot whose shapes all fit the K register slots stores only the poses its self survivors
      // need (after the self sweep, below); a larger one stores every pose here
      if (live && !lazy_store) {
        T* b = W.pstore + (size_t)s * 7 * W.stride + col;
        b[0] = ws.t.x; b[W.stride] = ws.t.y; b[2 * W.stride] = ws.t.z;
        b[3 * W.stride] = ws.r.i; b[4 * W.stride] = ws.r.j; b[5 * W.stride] = ws.r.k; b[6 * W.stride] = ws.r.w;
      }
      if (lds_centres) {
        T* cc = cen + (size_t)s * 3 * kWave + lane;
        cc[0] = ws.t.x; cc[kWave] = ws.t.y; cc[2 * kWave] = ws.t.z;
      }
    }
    if (!do_env && !lazy_store) continue;
    {
      T r = sh->br;
      asm volatile("" : "+v"(r));  // the radius in a VGPR: the obstacle record supplies the SGPR operand
      const T cc = dot(ws.t, ws.t), r2 = r * r;
      const T av = push_ok ? (cc - r2) - T(16) * Eps<T>::v * (cc + r2) : never;
#pragma unroll
      for (int j = 0; j < K; j++)
        if (fill == j) { P[j] = ws; a0[j] = av; rho[j] = r; }
      fill++;
    }
    if (fill < K && s != n_walk - 1) continue;
    if (!do_env) { fill = lazy_store ? fill : 0; continue; }

    // ---- sweep the `fill` shapes s0 .. s against every obstacle ----------------
    const int s0 = s + 1 - fill;
    V3<T> c[K];
#pragma unroll
    for (int j = 0; j < K; j++) c[j] = P[j].t;
    for (int o0 = 0; o0 < S.n_obs_pad; o0 += 32) {
      unsigned m[K];
      if (dbg_no_sweep) {
#pragma unroll
        for (int j = 0; j < K; j++) m[j] = 0u;
      } else {
        sweep_unit<T, K>(V.ocullq + (size_t)5 * o0, c, rho, a0, m);
      }
      unsigned nl = 0;
#pragma unroll
      for (int j = 0; j < K; j++) nl += (unsigned)__popc(m[j]);
      if (dbg_no_push) nl = 0;
      if (__ballot(nl != 0) == 0) continue;
      unsigned off, tot;
      wave_offsets(nl, off, tot);
      if (count + tot > W.cap) {
        overflow = true;
      } else if (nl != 0) {
        unsigned pos = count + off;
#pragma unroll
        for (int j = 0; j < K; j++) {
          unsigned mm = m[j];
          while (mm != 0) {
            const int z = __clz(mm);  // bit 31 - z  <->  obstacle o0 + z
            mm &= ~(0x80000000u >> z);
            ItemIO<T>::store(items + (size_t)pos * 8, ((unsigned)(o0 + z) << 16) | ((unsigned)(s0 + j) << 6) | (unsigned)lane, P[j]);
            pos++;
          }
        }
      }
      count += tot;
    }
    fill = 0;
#pragma unroll
    for (int j = 0; j < K; j++) a0[j] = never;
  }

  // ---- self shape pairs ---------------------------------------------------------
  if (do_self) {
    unsigned need = 0;  // lazy_store: bit j set when shape j's pose must reach the pose store
    if (!lds_centres) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // own stores -> own loads of the pose store
    for (int p0 = 0; p0 < S.n_spairs; p0 += 32) {
      const int cnt = min(32, S.n_spairs - p0);
      unsigned m = 0;
      for (int k = 0; k < cnt; k++) {
        const int s1 = V.spairs[4 * (p0 + k)], s2 = V.spairs[4 * (p0 + k) + 1];
        V3<T> c1, c2;
        if (lds_centres) {
          const T* a = cen + (size_t)s1 * 3 * kWave + lane;
          const T* b = cen + (size_t)s2 * 3 * kWave + lane;
          c1 = {a[0], a[kWave], a[2 * kWave]};
          c2 = {b[0], b[kWave], b[2 * kWave]};
        } else {
          const T* a = W.pstore + (size_t)s1 * 7 * W.stride + (live ? col : 0);
          const T* b = W.pstore + (size_t)s2 * 7 * W.stride + (live ? col : 0);
          c1 = {a[0], a[W.stride], a[2 * W.stride]};
          c2 = {b[0], b[W.stride], b[2 * W.stride]};
        }
        const V3<T> d = c1 - c2;
        const T rr = V.spair_r[p0 + k];
        m |= (push_ok && dot(d, d) < rr * rr) ? (0x80000000u >> k) : 0u;
      }
      if (dbg_no_push) m = 0;
      const unsigned nl = (unsigned)__popc(m);
      if (__ballot(nl != 0) == 0) continue;
      if (lazy_store) {
        for (int k = 0; k < cnt; k++) {  // shapes whose pose a survivor of this lane needs
          const unsigned both = (1u << V.spairs[4 * (p0 + k)]) | (1u << V.spairs[4 * (p0 + k) + 1]);
          need |= (m & (0x80000000u >> k)) ? both : 0u;
        }
      }
      unsigned off, tot;
      wave_offsets(nl, off, tot);
      if (count + tot > W.cap) {
        overflow = true;
      } else {
        unsigned pos = count + off;
        while (m != 0) {
          const int z = __clz(m);
          m &= ~(0x80000000u >> z);
          ItemIO<T>::store_code(items + (size_t)(pos++) * 8, 0x80000000u | ((unsigned)(p0 + z) << 6) | (unsigned)lane);
        }
      }
      count += tot;
    }
    if (lazy_store) {
#pragma unroll
      for (int j = 0; j < K; j++) {
        if ((need >> j) & 1u) {
          T* b = W.pstore + (size_t)j * 7 * W.stride + col;
          b[0] = P[j].t.x; b[W.stride] = P[j].t.y; b[2 * W.stride] = P[j].t.z;
          b[3 * W.stride] = P[j].r.i; b[4 * W.stride] = P[j].r.j; b[5 * W.stride] = P[j].r.k; b[6 * W.stride] = P[j].r.w;
        }
      }
    }
  }

  const unsigned long long bad = __ballot(live && !ok);
  if (lane == 0) W.hdr[wave] = {count, (unsigned)bad, (unsigned)(bad >> 32), overflow ? 1u : 0u};
}

// ---------------------------------------------------------------------------
// k_sweep_grid: the sweep with a broadphase.  Instead of testing a robot shape against every
// obstacle, the lane looks up the grid cell its shape centre falls in (one grid per class of shape
// radius, built on the host: gear_scene.hpp build_grids) and tests only the obstacles listed there.
// Shapes are handled one at a time as the FK walk produces them, so no pose has to stay in
// registers.  A cell's list is a chain of 16-byte blocks of eight obstacle indices: ONE load brings a
// lane's whole list (almost always), the eight cull records are then fetched together, and the
// survivors of the block take consecutive queue slots from ballot prefix sums (no atomics).  Per
// shape that is two dependent memory round trips, not two per candidate.  With self pairs every shape
// pose goes to the pose store.  Same queue format as k_sweep; the flags are identical because every
// pair whose bounding spheres overlap is listed (conservative lists), and pairs that do not cannot
// collide.
// ---------------------------------------------------------------------------
template <class T> struct alignas(16) Cull4 { T x, y, z, r; };

constexpr int kSgWaves = 4;  // wavefronts per k_sweep_grid workgroup (they share the LDS copy of the cull table)

// CLDS: the obstacles' cull records (16 bytes each) sit in LDS, one copy per workgroup: the eight
// record fetches of a cell block then go through the LDS pipe instead of the texture-address path,
// which is what this kernel queues on.
template <class T, bool CLDS>
__global__ void __launch_bounds__(kWave * kSgWaves)
k_sweep_grid(SceneDev<T> S, const T* __restrict__ q, size_t n, uint32_t what, PipeWs<T> W, unsigned per_wave_lds) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gear_lds[];
  const Views<T> V(S);
  clear_segment_counters(W.pc);
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const size_t n_waves = (n + kWave - 1) / kWave;
  const size_t wave_raw = (size_t)blockIdx.x * kSgWaves + wv;
  const bool wave_live = wave_raw < n_waves;      // the last workgroup may hold idle wavefronts
  const size_t wave = wave_live ? wave_raw : n_waves - 1;
  const size_t col = wave * kWave + lane;
  const bool live = wave_live && col < n;
  const size_t i = col < n ? col : n - 1;
  const T* qrow = q + i * (size_t)S.n_cols;
  // LDS: [cull table (CLDS)] [per wavefront: branch-point pose slots, poses of "earlier" self-pair shapes]
  const size_t table_bytes = CLDS ? ((size_t)(S.n_obs + 1) * 4 * sizeof(T) + 15) / 16 * 16 : 0;
  unsigned char* my_lds = gear_lds + table_bytes + (size_t)wv * per_wave_lds;
  const PoseStore<T> slots{reinterpret_cast<T*>(my_lds), (size_t)kWave, (size_t)lane};
  // poses of the shapes that are the earlier one of a self pair: [n_first][7][64]
  const PoseStore<T> first{reinterpret_cast<T*>(my_lds + (size_t)S.n_slots * 7 * kWave * sizeof(T)), (size_t)kWave, (size_t)lane};
  const GEAR_CAS int32_t* sp_first_slot = to_cas(S.sp_first_slot);
  const GEAR_CAS int32_t* sp_late_off = to_cas(S.sp_late_off);
  const GEAR_CAS int32_t* sp_late = to_cas(S.sp_late);
  if (CLDS) {
    T* tab = reinterpret_cast<T*>(gear_lds);
    for (int k = threadIdx.x; k < (S.n_obs + 1) * 4; k += kWave * kSgWaves) tab[k] = S.ocull[k];
    __syncthreads();
  }

  const bool ok = (what & kWhatNoLimits) != 0 || limits_ok(S, V, qrow);  // GH_CHECK_NO_LIMITS: geometry only
  const bool do_env = (what & 1u) && S.n_obs > 0;
  const bool do_self = (what & 2u) && S.n_spairs > 0;
  const bool push_ok = live && ok;  // the reference evaluates no collision after a limit error
  const bool dbg_no_sweep = GEAR_DBG(what, 0x800u);

  T* items = W.items + wave * 8 * (size_t)W.cap;
  const GEAR_CAS GridDev<T>* grids = to_cas(S.grids);
  const GEAR_CAS int32_t* shape_grid = to_cas(S.shape_grid);
  const uint4* __restrict__ blocks = reinterpret_cast<const uint4*>(S.cell_blk);
  const Cull4<T>* __restrict__ ocull = CLDS ? reinterpret_cast<const Cull4<T>*>(gear_lds) : reinterpret_cast<const Cull4<T>*>(S.ocull);
  const unsigned sentinel = (unsigned)S.n_obs;

  const int n_walk = (do_env || do_self) ? S.n_shapes : 0;
  unsigned count = 0;  // wave-uniform
  bool overflow = false;
  unsigned long long ncand = 0;

  int node = -1;
  Pose<T> cur{{T(0), T(0), T(0)}, {T(0), T(0), T(0), T(1)}};
  for (int s = 0; s < n_walk; s++) {
    const GEAR_CAS ShapeDev<T>* sh = V.rshapes + s;
    const int target = sh->node;
    while (node < target) {
      node++;
      node_world(V, node, qrow, slots, cur);
    }
    const Pose<T> ws = pose_mul(cur, shape_pose<T>(sh));
    if (do_self) {
      // self shape pairs whose later shape is s: its pose is live, the other one's waits in LDS.  A
      // survivor takes two queue records, {code, pose of the pair's first shape} and {skip, pose of its
      // second shape}: nothing else has to be kept for the narrow phase.
      for (int e = sp_late_off[s]; e < sp_late_off[s + 1]; e++) {
        const int p = sp_late[e];
        const int s1 = V.spairs[4 * p], s2 = V.spairs[4 * p + 1];
        const int other = s1 == s ? s2 : s1;
        Pose<T> po = ws;
        if (other != s) po = first.get(sp_first_slot[other]);
        const V3<T> d = ws.t - po.t;  // (c1 - c2) or (c2 - c1): the squares are the same
        const T rr = V.spair_r[p];
        const bool hit = push_ok && dot(d, d) < rr * rr && !GEAR_DBG(what, 0x400u);
        const unsigned long long hb = __ballot(hit);
        if (hb == 0) continue;
        const unsigned tot = 2u * (unsigned)__popcll(hb);
        if (count + tot > W.cap) {
          overflow = true;
        } else if (hit) {
          const unsigned pos = count + 2u * lane_rank(hb);
          const bool s_first = s1 == s;  // s is the pair's first shape
          ItemIO<T>::store(items + (size_t)pos * 8, kSelfInline | ((unsigned)s1 << 16) | ((unsigned)s2 << 6) | (unsigned)lane, s_first ? ws : po);
          ItemIO<T>::store(items + (size_t)(pos + 1) * 8, kItemSkip, s_first ? po : ws);
        }
        count += tot;
      }
      const int fs = sp_first_slot[s];
      if (fs >= 0) first.put(fs, ws);
    }
    if (!do_env || dbg_no_sweep) continue;
    // ---- cell lookup ---------------------------------------------------------------------------
    const GEAR_CAS GridDev<T>* g = grids + shape_grid[s];
    const T fx = (ws.t.x - g->org[0]) * g->inv_h, fy = (ws.t.y - g->org[1]) * g->inv_h, fz = (ws.t.z - g->org[2]) * g->inv_h;
    const bool inside = push_ok && fx >= T(0) && fy >= T(0) && fz >= T(0) && fx < T(g->nx) && fy < T(g->ny) && fz < T(g->nz);
    unsigned blk = 0;  // block 0 is the empty list
    if (inside) blk = (unsigned)g->off_base + ((unsigned)(int)fz * (unsigned)g->ny + (unsigned)(int)fy) * (unsigned)g->nx + (unsigned)(int)fx;
    const T rho = sh->br;
    const unsigned code_base = ((unsigned)s << 6) | (unsigned)lane;
    // half extents of the shape's oriented box, padded by the rounding of the transformed offset
    const bool is_ball = sh->kind == SK_BALL;
    const T pad = T(1e-5) * (T(1) + rho);
    const T hbx = sh->half[0] + pad, hby = (is_ball ? sh->half[0] : sh->half[1]) + pad, hbz = (is_ball ? sh->half[0] : sh->half[2]) + pad;
    bool more;
    do {
      const uint4 w = blocks[blk];
      const bool cont = (w.w >> 31) != 0;
      unsigned id[8] = {w.x & 0xffffu, w.x >> 16, w.y & 0xffffu, w.y >> 16, w.z & 0xffffu, w.z >> 16, w.w & 0xffffu, w.w >> 16};
      if (cont) id[6] = id[7] = sentinel;
      Cull4<T> oc[8];
#pragma unroll
      for (int j = 0; j < 8; j++) oc[j] = ocull[id[j]];  // eight independent loads; a sentinel reads the far dummy
      unsigned h = 0;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const T dx = ws.t.x - oc[j].x, dy = ws.t.y - oc[j].y, dz = ws.t.z - oc[j].z;
        const T rr = rho + oc[j].r;
        h |= (dx * dx + dy * dy + dz * dz < rr * rr) ? (1u << j) : 0u;
      }
      if (what & 0x8000u) {  // diagnostic count of the candidates
#pragma unroll
        for (int j = 0; j < 8; j++) ncand += id[j] != sentinel ? 1u : 0u;
      }
      // second cull: the obstacle's bounding sphere against the shape's ORIENTED box (a box's own extents,
      // a mesh's local bounding box, the cube around a ball).  Conservative like the first (the record's
      // radius carries margin and slack); it halves what is queued for the narrow phase.
      if (!GEAR_DBG(what, 0x200000u) && __ballot(h != 0) != 0) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
          if (__ballot((h >> j) & 1u) == 0) continue;  // uniform
          const T dx = oc[j].x - ws.t.x, dy = oc[j].y - ws.t.y, dz = oc[j].z - ws.t.z;
          const V3<T> pl = qrot(qconj(ws.r), V3<T>{dx, dy, dz});  // the offset in the shape's frame
          const T ex = t_max(t_abs(pl.x) - hbx, T(0)), ey = t_max(t_abs(pl.y) - hby, T(0)), ez = t_max(t_abs(pl.z) - hbz, T(0));
          if (ex * ex + ey * ey + ez * ez >= oc[j].r * oc[j].r) h &= ~(1u << j);
        }
      }
      if (GEAR_DBG(what, 0x400u)) h = 0;  // profiling ablation: no compaction, no stores
      const unsigned nl = (unsigned)__popc(h);
      if (__ballot(nl != 0) != 0) {
        unsigned off, tot;
        wave_offsets(nl, off, tot);
        if (count + tot > W.cap) {
          overflow = true;
        } else {
          // one pass per hit of the busiest lane (two or three), not one per slot of the block (eight):
          // fewer store instructions, each with more active lanes
          unsigned pos = count + off;
          unsigned hh = h;
          while (__ballot(hh != 0) != 0) {
            if (hh != 0) {
              const unsigned z = (unsigned)__builtin_ctz(hh);
              hh &= hh - 1u;
              // slot z of the block: 16 bits of one of its four words (a hit is never on a sentinel slot,
              // so the raw words serve even in a chained block)
              const unsigned wz = z < 4u ? (z < 2u ? w.x : w.y) : (z < 6u ? w.z : w.w);
              const unsigned idz = (wz >> ((z & 1u) << 4)) & 0xffffu;
              ItemIO<T>::store(items + (size_t)pos * 8, (idz << 16) | code_base, ws);
              pos++;
            }
          }
        }
        count += tot;
      }
      blk = cont ? (w.w & 0x7fffffffu) : 0u;
      more = __ballot(cont) != 0;
    } while (more);
  }
  if (what & 0x8000u) atomicAdd(&W.pc->candidates, ncand);

  const unsigned long long bad = __ballot(live && !ok);
  if (lane == 0 && wave_live) W.hdr[wave] = {count, (unsigned)bad, (unsigned)(bad >> 32), overflow ? 1u : 0u};
}

// fetch the two shapes and poses of an item of wave `wave`
template <class T>
__device__ __forceinline__ void item_fetch(const SceneDev<T>& S, const Views<T>& V, const PipeWs<T>& W, size_t wave,
                                           unsigned j, unsigned code, ShapeRef<T>& A, Pose<T>& wa, ShapeRef<T>& B,
                                           Pose<T>& wb, unsigned dbg = 0) {
  if ((code >> 30) == 3u) {
    const int s1 = (code >> 16) & 1023, s2 = (code >> 6) & 1023;
    const T* rec = W.items + (wave * (size_t)W.cap + j) * 8;
    A = shape_ref<T>(V.rshapes + s1);
    wa = ItemIO<T>::pose(rec);
    B = shape_ref<T>(V.rshapes + s2);
    wb = ItemIO<T>::pose(rec + 8);
  } else if (code >> 31) {
    const int pi = (code >> 6) & 0xffffff;
    const int s1 = V.spairs[4 * pi], s2 = V.spairs[4 * pi + 1];
    const size_t col = wave * kWave + (code & 63u);
    const T* a = W.pstore + (size_t)s1 * 7 * W.stride + col;
    const T* b = W.pstore + (size_t)s2 * 7 * W.stride + col;
    A = shape_ref<T>(V.rshapes + s1);
    wa = {{a[0], a[W.stride], a[2 * W.stride]}, {a[3 * W.stride], a[4 * W.stride], a[5 * W.stride], a[6 * W.stride]}};
    B = shape_ref<T>(V.rshapes + s2);
    wb = {{b[0], b[W.stride], b[2 * W.stride]}, {b[3 * W.stride], b[4 * W.stride], b[5 * W.stride], b[6 * W.stride]}};
  } else {
    int s = (code >> 6) & 1023;
    int o = (code >> 16) & 32767;
    if (GEAR_DBG(dbg, 0x10000u)) o &= 3;  // profiling ablation: four distinct obstacle records (results are wrong)
    if (GEAR_DBG(dbg, 0x20000u)) o = __builtin_amdgcn_readfirstlane(o);  // ablation: one obstacle record per wave, scalar loads
    if (GEAR_DBG(dbg, 0x40000u)) s = __builtin_amdgcn_readfirstlane(s);  // ablation: one robot record per wave
    A = shape_ref<T>(V.rshapes + s);
    wa = ItemIO<T>::pose(W.items + (wave * (size_t)W.cap + j) * 8);
    B = shape_ref<T>(V.oshapes + o);
    wb = shape_pose<T>(V.oshapes + o);
  }
}

// ---------------------------------------------------------------------------
// k_narrow1: workgroup b (4 wavefronts) handles the items of sweep wave b, one item per lane.
//
// The stage is a chain of memory round trips, each 1-2 us under load (cycle stamps, GEAR_HIP_STAMPS=1),
// so everything that depends on nothing is requested at once when the workgroup starts: the wave
// header, the item records of the first TWO rounds (code and pose together; indices clamped to the
// region instead of waiting for the count), and the compact per-shape table of what stage 1 needs
// (SceneDev::hot): the robot's part goes to LDS, the obstacles' part is gathered per item from its
// 48-byte records (OLDS stages it in LDS as well; measured slightly slower).  The poses of a self item are its own record
// and the next one: the neighbour lane holds it (lane 63 re-reads it).
// ---------------------------------------------------------------------------
constexpr int kN1Threads = 256;

template <class T> GD ShapeRef<T> hot_ref(T hx, T hy, T hz, T kind) {
  ShapeRef<T> r;
  r.kind = (int)kind;
  r.half = {hx, hy, hz};
  r.br = T(0);
  r.vb = r.vc = r.fb = r.fc = r.pn = 0;
  return r;
}

template <class T, bool OLDS>
__global__ void __launch_bounds__(kN1Threads)
k_narrow1(SceneDev<T> S, size_t n, uint32_t what, PipeWs<T> W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gear_lds[];
  __shared__ unsigned hits[4];      // [0..1] env bits of lanes 0-31 / 32-63, [2..3] self bits
  __shared__ unsigned exq[kExLds];  // staged exact entries: item | owner lane << kItemBits | self << (kItemBits + 6) | class << 28
  __shared__ unsigned en;           // fill of exq
  __shared__ unsigned cbin[kClasses];   // kept entries per class
  __shared__ unsigned cbase[kClasses];  // where the class's entries go in its queue segment
  __shared__ unsigned cfit[kClasses];   // how many of them the segment still takes
  using Rec = typename ItemIO<T>::Rec;
  const Views<T> V(S);
  const int tid = threadIdx.x;
  const size_t wave = blockIdx.x;
  auto stamp = [&](int k) {
    if (W.stamps && tid == 0) W.stamps[wave * 8 + k] = (unsigned long long)clock64();
  };
  stamp(0);
  T* hot = reinterpret_cast<T*>(gear_lds);  // [L][4], then (OLDS) [O][12]
  const T* items = W.items + wave * 8 * (size_t)W.cap;
  // ---- one trip: header, two rounds of records, the table ------------------------------------------
  const WaveHdr h = W.hdr[wave];
  Rec rec[2];
  rec[0] = ItemIO<T>::load(items + (size_t)min((unsigned)tid, W.cap - 1u) * 8);
  rec[1] = ItemIO<T>::load(items + (size_t)min((unsigned)tid + kN1Threads, W.cap - 1u) * 8);
  {
    const int n_hot = S.n_shapes * 4 + (OLDS ? S.n_obs * 12 : 0);
    for (int i = tid; i < n_hot; i += kN1Threads) hot[i] = S.hot[i];
  }
  if (h.overflow) {  // the region ran full: the fused path re-checks these configurations
    const size_t cfg = wave * kWave + tid;
    if (tid < kWave && cfg < n) W.hit[cfg] = kHitRedo;
    return;
  }
  if (tid < 4) hits[tid] = 0;
  if (tid == 4) en = 0;
  if (tid >= 64 && tid < 64 + kClasses) cbin[tid - 64] = 0;
  __syncthreads();
  stamp(1);  // header, records and table in
  const T* ohot_g = S.hot + (size_t)S.n_shapes * 4;  // obstacle part in memory (!OLDS)
  const T* ohot_l = hot + (size_t)S.n_shapes * 4;
  const T margin = S.margin;
  LaneCounters lc{0, 0, 0, 0};
  __shared__ unsigned redo_l;  // a segment of the exact queue ran full
  if (tid == 5) redo_l = 0;

  for (unsigned base = 0; base < h.count; base += 2 * kN1Threads) {
    if (base != 0) {  // more than two rounds: rare
      rec[0] = ItemIO<T>::load(items + (size_t)min(base + (unsigned)tid, W.cap - 1u) * 8);
      rec[1] = ItemIO<T>::load(items + (size_t)min(base + (unsigned)tid + kN1Threads, W.cap - 1u) * 8);
    }
#pragma unroll
    for (int half = 0; half < 2; half++) {
      const unsigned rbase = base + (unsigned)half * kN1Threads;
      if (rbase >= h.count) break;  // uniform
      const unsigned j = rbase + tid;
      const Rec& rc = rec[half];
      const unsigned code = ItemIO<T>::code(rc);
      bool valid = j < h.count && code != kItemSkip;
      const int ln = code & 63;
      const bool sf = (code >> 31) != 0;
      if (valid) {
        const unsigned hw = hits[(sf ? 2 : 0) + (ln >> 5)];
        valid = ((hw >> (ln & 31)) & 1u) == 0;  // owner already known to collide
      }
      if (GEAR_DBG(what, 0x200u)) valid = false;  // profiling ablation
      // the next record's pose, for self items: the neighbour lane holds it
      const Pose<T> mine = ItemIO<T>::pose(rc);
      Pose<T> next;
      next.t.x = __shfl_down(mine.t.x, 1); next.t.y = __shfl_down(mine.t.y, 1); next.t.z = __shfl_down(mine.t.z, 1);
      next.r.i = __shfl_down(mine.r.i, 1); next.r.j = __shfl_down(mine.r.j, 1); next.r.k = __shfl_down(mine.r.k, 1);
      next.r.w = __shfl_down(mine.r.w, 1);
      int r = PR_FREE;
      unsigned cls = 0;
      if (valid) {
        ShapeRef<T> A, B;
        Pose<T> wa, wb;
        if ((code >> 30) == 3u) {  // self pair, both poses inline
          const int s1 = (code >> 16) & 1023, s2 = (code >> 6) & 1023;
          A = hot_ref(hot[4 * s1], hot[4 * s1 + 1], hot[4 * s1 + 2], hot[4 * s1 + 3]);
          B = hot_ref(hot[4 * s2], hot[4 * s2 + 1], hot[4 * s2 + 2], hot[4 * s2 + 3]);
          wa = mine;
          wb = (tid & 63) == 63 ? ItemIO<T>::pose(items + (size_t)(j + 1) * 8) : next;
        } else if (sf) {  // self pair of the brute-force sweep: poses in the pose store
          item_fetch(S, V, W, wave, j, code, A, wa, B, wb, what);
        } else {
          const int s = (code >> 6) & 1023, o = (code >> 16) & 32767;
          A = hot_ref(hot[4 * s], hot[4 * s + 1], hot[4 * s + 2], hot[4 * s + 3]);
          wa = mine;
          const T* ob = (OLDS ? ohot_l : ohot_g) + (size_t)o * 12;
          B = hot_ref(ob[8], ob[9], ob[10], ob[3]);
          wb = {{ob[0], ob[1], ob[2]}, {ob[4], ob[5], ob[6], ob[7]}};
        }
        r = pair_stage1<T, false>(A, wa, B, wb, margin, lc);
        cls = exact_class(A.kind, B.kind, sf);
        if (r == PR_HIT) atomicOr(&hits[(sf ? 2 : 0) + (ln >> 5)], 1u << (ln & 31));
      }
      if (rbase == 0) stamp(2);  // first round computed
      // stage the undecided items; when the staging area cannot take another full round, move it to
      // the global exact queue with one atomic for the whole workgroup
      const bool ex = r == PR_EXACT;
      const unsigned long long eb = __ballot(ex);
      unsigned slot = 0;
      if (eb != 0) {
        if ((tid & 63) == (int)__builtin_ctzll(eb)) slot = atomicAdd(&en, (unsigned)__popcll(eb));
        slot = __shfl(slot, (int)__builtin_ctzll(eb)) + lane_rank(eb);
        if (ex) exq[slot] = j | ((unsigned)ln << kItemBits) | (sf ? 1u << (kItemBits + 6) : 0u) | (cls << (32 - kClassBits));  // en <= kExLds - kN1Threads held before this round
      }
      __syncthreads();
      if (rbase == 0) stamp(3);  // barrier passed
      const bool last = rbase + kN1Threads >= h.count;
      if (en + kN1Threads > kExLds || last) {
        const unsigned cnt = en;
        if (cnt != 0) {
          // Most undecided items belong to a configuration that some other item has meanwhile shown to
          // collide (the metric scene: seven of eight): the flags are an OR, so those need no exact
          // query.  Only the rest goes to the exact queue, which keeps k_narrow2's wavefronts full.
          // A kept entry takes its rank inside its class; then one atomic per class present, issued
          // together by kClasses lanes (one trip), reserves the segments' slots.
          unsigned e2[kExLds / kN1Threads], rk[kExLds / kN1Threads];
#pragma unroll
          for (int part = 0; part < kExLds / kN1Threads; part++) {
            const unsigned k = (unsigned)(part * kN1Threads + tid);
            const unsigned ent = exq[min(k, (unsigned)kExLds - 1u)];
            const unsigned eln = (ent >> kItemBits) & 63u, esf = (ent >> (kItemBits + 6)) & 1u;
            const bool keep = k < cnt && (((hits[2u * esf + (eln >> 5)] >> (eln & 31u)) & 1u) == 0 || GEAR_DBG(what, 0x1000000u));
            e2[part] = ent;
            rk[part] = ~0u;
            if (keep) rk[part] = atomicAdd(&cbin[ent >> (32 - kClassBits)], 1u);
          }
          __syncthreads();
          const unsigned sub = (unsigned)(wave % kSubs);
          if (tid < kClasses) {
            const unsigned c = cbin[tid];
            // a segment that cannot take all c entries takes the ones that fit: every slot below
            // min(fill, cap2) is then a written one, which is what k_narrow2 walks
            unsigned b = 0, fit = 0;
            if (c != 0) {
              b = atomicAdd(&W.pc->seg[(unsigned)tid * kSubs + sub][0], c);
              fit = b < W.cap2 ? min(c, W.cap2 - b) : 0u;
              if (fit < c) redo_l = 1u;  // the fused path re-checks this wave's configurations
            }
            cbase[tid] = b;
            cfit[tid] = fit;
          }
          __syncthreads();
#pragma unroll
          for (int part = 0; part < kExLds / kN1Threads; part++) {
            const unsigned cl = e2[part] >> (32 - kClassBits);
            if (rk[part] < cfit[cl])  // not a kept entry: rank ~0u
              W.exact[(cl * kSubs + sub) * (size_t)W.cap2 + cbase[cl] + rk[part]] = ((unsigned)wave << kItemBits) | (e2[part] & (unsigned)(kMaxItemCap - 1));
          }
          if (!last) {  // the staging areas are used again
            __syncthreads();
            if (tid == 0) en = 0;
            if (tid < kClasses) cbin[tid] = 0;
            __syncthreads();
          }
        }
      }
    }
  }
  __syncthreads();
  const bool redo = redo_l != 0;  // uniform over the workgroup
  stamp(6);  // all rounds and flushes done
  if (tid < kWave) {
    const size_t cfg = wave * kWave + tid;
    const bool env_hit = ((hits[tid >> 5] >> (tid & 31)) & 1u) != 0;
    const bool self_hit = ((hits[2 + (tid >> 5)] >> (tid & 31)) & 1u) != 0;
    const bool lim = (((tid < 32 ? h.limit_lo : h.limit_hi) >> (tid & 31)) & 1u) != 0;
    unsigned out = lim ? 4u : ((env_hit ? 1u : 0u) | (self_hit ? 2u : 0u));
    if (redo) out |= kHitRedo;
    if (cfg < n) W.hit[cfg] = out;
  }
  stamp(7);
}

// ---------------------------------------------------------------------------
// k_narrow2: the exact stage over the segments of the exact queue (grid row = segment = one pair class,
// grid-stride inside): the items of a wavefront run the same support functions and the same parts of the
// hollow-mesh rule.  One wavefront per workgroup, so that the dispatcher spreads a short queue over
// all SIMDs of the chip.
// MLDS: the scene's mesh vertices and face planes sit in LDS as 16-byte records (SceneDev::mesh4), one
// access per vertex or plane.
// ---------------------------------------------------------------------------
constexpr int kN2Threads = 64;

template <class T, bool MLDS, bool HIST = false>
__global__ void __launch_bounds__(kN2Threads)
k_narrow2(SceneDev<T> S, uint32_t what, PipeWs<T> W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gear_lds[];
  const Views<T> V(S);
  const int tid = threadIdx.x;
  const unsigned seg = blockIdx.y;
  const unsigned total = min(W.pc->seg[seg][0], W.cap2);
  if (blockIdx.x * kN2Threads >= total) return;  // nothing for this workgroup (uniform)
  const Vec4Tab<T> lverts{reinterpret_cast<const Quad<T>*>(gear_lds)};
  const Vec4Tab<T> lplanes{reinterpret_cast<const Quad<T>*>(gear_lds) + S.n_verts};
  if (MLDS) {
    const Quad<T>* src = reinterpret_cast<const Quad<T>*>(S.mesh4);
    Quad<T>* dst = reinterpret_cast<Quad<T>*>(gear_lds);
    for (int k = tid; k < S.n_verts + S.n_planes; k += kN2Threads) dst[k] = src[k];
    __syncthreads();
  }
  const T margin = S.margin;
  const unsigned* queue = W.exact + seg * (size_t)W.cap2;
  LaneCounters lc{0, 0, 0, 0};
  for (unsigned k = blockIdx.x * kN2Threads + (unsigned)tid; k < total; k += gridDim.x * kN2Threads) {
    unsigned e = queue[k];
    if (GEAR_DBG(what, 0x400000u)) e = __builtin_amdgcn_readfirstlane(e);  // profiling ablation: one item per wave, no divergence
    const size_t wave = e >> kItemBits;
    const unsigned j = e & (unsigned)(kMaxItemCap - 1);
    const unsigned code = ItemIO<T>::code(W.items + (wave * (size_t)W.cap + j) * 8);
    const size_t cfg = wave * kWave + (code & 63u);
    const unsigned bit = (code >> 31) ? 2u : 1u;
    if (W.hit[cfg] & (bit | kHitRedo)) continue;
    if (GEAR_DBG(what, 0x100u)) continue;  // profiling ablation
    ShapeRef<T> A, B;
    Pose<T> wa, wb;
    item_fetch(S, V, W, wave, j, code, A, wa, B, wb);
    bool hit;
    if (MLDS) hit = pair_stage2<T, HIST>(A, wa, B, wb, lverts, lplanes, margin, lc);
    else hit = pair_stage2<T, HIST>(A, wa, B, wb, V.verts, V.planes, margin, lc);
    if (hit) atomicOr(&W.hit[cfg], bit);
    if (HIST) {
      atomicAdd(&W.pc->hist[0][min(lc.exact_iters, 47u)], 1u);
      unsigned mx = lc.exact_iters;
      for (int d = 32; d > 0; d >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, d));
      if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(__ballot(1))) atomicAdd(&W.pc->hist[1][min(mx, 47u)], 1u);
      lc.exact_iters = 0;
    }
  }
}

}  // namespace gear_hip
