// gear_pipeline.hpp — the batched form of the check on gfx950: two lean kernels in front of the fused
// k_check, so that every stage runs at the occupancy and register budget it can use (the fused kernel
// carries the exact stage's ~200 VGPRs through its sweep and keeps every shape pose in LDS).
//
//   k_sweep_n1  one lane per configuration: limits, FK walk, and per robot shape the broadphase (grid
//               cell -> candidate obstacles -> bounding-sphere cull -> oriented-box cull).  Survivors go,
//               pose included, into a ring in the wavefront's own LDS; whenever the ring holds 64 items the
//               wavefront turns round and decides them, one item per lane (closed forms / SAT: stage 1 of
//               the narrow phase).  Nothing of that leaves the chip: what is written to memory is one hit
//               word per configuration and the few items stage 1 could not decide (a mesh is involved,
//               or two boxes are closer than the margin on their best axis) with their poses.
//   k_narrow2   the exact stage over those items: drops the ones whose configuration is already known
//               to collide, bins the rest by pair class in LDS so that the lanes of a wavefront take the
//               same branches, then GJK margin query + hollow-mesh rule, one item per lane — a configuration's
//               first item first, the later ones only where that did not hit — and writes the flag bytes.
//   k_check<redo>  a sweep wavefront whose exact region ran full is re-checked by the fused path (a workgroup
//               looks at 64 marks and normally finds none), so the result never depends on a capacity.
//
// Reference semantics are unchanged: per configuration this evaluates is_feasible /
// is_feasible_with_self (src/planner/joint_path_planner.rs:71-84,113-121); only the order in which pairs
// are looked at differs, and the flags are an OR over pairs.
#pragma once
#include <type_traits>

#include "gear_kernels.hpp"

namespace gear_hip {

constexpr int kClassBits = 4;  // pair classes of the exact stage (exact_class)
constexpr int kClasses = 1 << kClassBits;
// exact entry key, word x: class << 28 | kKeyLater | configuration within the chunk (chunks hold at most 2^27)
constexpr unsigned kKeyLater = 1u << 27, kKeyCfg = kKeyLater - 1u;

// Pair class of an undecided item: which support functions and which parts of the hollow-mesh rule its
// exact query runs.  A wavefront of mixed items executes the union of all their paths, and with one or
// two waves per SIMD nothing hides that, so the exact stage groups its items by class.
GD unsigned exact_class(int kind_a, int kind_b, bool self) {
  // kinds: box 0, ball 1, mesh 2.  box-box 0, box-mesh 1, ball-mesh 2, mesh-box 3, mesh-ball 4, mesh-mesh 5
  // (ball-ball and box-ball are decided in stage 1: 6); self pairs + 8
  const unsigned long long lut = 0x543266160ull;
  return (unsigned)((lut >> (4 * (kind_a * 3 + kind_b))) & 15u) + (self ? 8u : 0u);
}

// Item codes.  env: obstacle << 16 | robot shape << 6 | lane (obstacle < 32768, shape < 1024).
// self: kSelfItem | self shape pair << 6 | lane.
constexpr unsigned kSelfItem = 0x80000000u;
constexpr int kRing = 128;  // ring entries per wavefront (a push pass adds at most 64, a drain takes 64)
constexpr int kHitList = 256;  // hit-list entries per wavefront (64 lanes x the four slots of half a cell block)

struct PipeCounters {
  unsigned long long candidates;  // broadphase candidates tested (only with GH_COUNT_CANDIDATES in `what`)
  unsigned long long exact;       // items handed to the exact stage (same)
  unsigned redo_waves;            // wavefronts whose exact region ran full (same)
  unsigned pad;
  unsigned hist[2][48];  // GEAR_HIP_N2_HIST=1 (profiling): GJK iterations per exact item / the largest count in a wavefront
};

// One launch chunk's workspace.  Sweep wavefront w owns entries [w * capw, (w + 1) * capw) of the exact
// arrays: no atomics, no shared fill counter, and the order of the entries does not depend on timing.
template <class T> struct PipeWs {
  unsigned* hit;    // [configs] bit0 env, bit1 self, bit2 limit, kHitRedo
  unsigned* xcnt;   // [waves] exact entries of sweep wave w (0 when the wave asks for the fused re-check)
  uint2* xkey;      // [waves][capw] {pair class << 28 | configuration index inside the chunk, item code}
  T* xpose_a;       // [waves][capw][8] world pose of the item's first shape (element 0 unused)
  T* xpose_b;       // [waves][capw][8] world pose of a self item's second shape (null: the call has no self pairs)
  T* fq;            // [waves][n_first][64][4] rotation of every shape that is the earlier one of a self pair (same)
  PipeCounters* pc;
  unsigned capw;
  unsigned* note;
  uint8_t* flags;   // the chunk's flag bytes (written by the exact stage, k_narrow2) and how many
  unsigned n_cfg;   // host word (as the device addresses it; may be null): set when a wavefront's exact region ran full
};

// One ring / pose record = 8 T (32 B in f32, 64 B in f64), written and read with 16-byte accesses.
// Element 0 carries the item code (bit pattern).
template <class T> struct ItemIO;
template <> struct ItemIO<float> {
  struct Rec { uint4 a, b; };
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
  GD static Pose<float> pose(const float* rec) { return pose(load(rec)); }
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
  GD static Pose<double> pose(const double* rec) { return pose(load(rec)); }
};

__device__ __forceinline__ unsigned lane_rank(unsigned long long b) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
}

// lanes of ONE wavefront exchange data through its LDS region: order the accesses, nothing else
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <class T> GD ShapeRef<T> hot_ref(T hx, T hy, T hz, T kind) {
  ShapeRef<T> r;
  r.kind = (int)kind;
  r.half = {hx, hy, hz};
  r.br = T(0);
  r.vb = r.vc = r.fb = r.fc = r.pn = 0;
  return r;
}

template <class T> struct alignas(16) Cull4 { T x, y, z, r; };

// Where the sweep keeps its lane's row of the batch.  Read straight from memory, every joint value costs the
// walk a scalar load (which column?) and then a dependent vector load.  Q_LDS fetches the whole row with
// independent loads when the wavefront starts and parks it in an LDS column: +5 % on env-only calls; it is
// used when the column does not cost a resident wavefront (a register vector indexed by the uniform column
// was measured too: the extra spills cost more than the loads).
enum : int { Q_MEM = 0, Q_LDS = 2 };
template <class T> struct QLds {
  const T* col;  // this lane's column: element c at col[c * 64]
  __device__ __forceinline__ T operator[](int c) const { return col[c * kWave]; }
};

// ---------------------------------------------------------------------------
// k_sweep_n1
//
// LDS of a workgroup: [cull table (CLDS)] [robot part of SceneDev::hot] then per wavefront
// [branch-point pose slots][centres of the shapes that are the earlier one of a self pair][ring][hit words][hit list].
// (Those shapes' rotations wait in memory, PipeWs::fq: only the ~2 self pairs per configuration that survive
// the centre cull need them, and 4 of 7 pose components out of LDS is what lets a fourth wavefront per SIMD
// be resident.)
// The cull table (16 bytes per obstacle) joins the workgroup in LDS when it is small enough: a cell block's
// record fetches then go through the LDS pipe instead of the texture-address path.
//
// Broadphase per robot shape: the grid cell of the shape centre (one grid per class of shape radius, built
// on the host: gear_scene.hpp build_grids) heads a chain of 16-byte blocks of eight 16-bit obstacle
// indices: ONE load brings the lane's whole candidate list (almost always), the eight cull records are
// fetched together and tested against the shape's bounding sphere.  Each lane then walks ITS hits (two or
// three, not the block's eight slots): re-reads the obstacle's record, tests it against the shape's
// oriented box (a box's own extents, a mesh's local bounding box, the cube around a ball; conservative
// like the first cull), and what survives takes a ring slot from a ballot prefix sum.
// Conservative lists + conservative culls: every pair that can be within the margin is decided by the
// narrow phase, so the flags equal the brute-force sweep's.
// ---------------------------------------------------------------------------
constexpr int kSwThreads = 256;
// Wavefronts per SIMD the sweep is compiled for: four in f32 (128 VGPRs; a dozen values spill to scratch on
// rare paths, which costs less than the fourth wavefront hides), two in f64.
template <class T> constexpr int sweep_occupancy() { return sizeof(T) == 4 ? 4 : 2; }

template <class T, bool CLDS, int QMODE>
__global__ void __launch_bounds__(kSwThreads, sweep_occupancy<T>())
k_sweep_n1(SceneDev<T> S, const T* __restrict__ q, size_t n, uint32_t what, PipeWs<T> W, unsigned table_bytes,
           unsigned per_wave_lds, unsigned first_bytes, int all_blk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gear_lds[];
  using Rec = typename ItemIO<T>::Rec;
  // The two culls are conservative filters in front of an exact decision: their table is inflated by 1e-4 of the
  // radii, the cell lists by 1e-5 m and more, the oriented box by 1e-5.  With the table in LDS they are evaluated in
  // f32 in BOTH precisions (the f64 kernel's vector unit runs f64 at half the rate, its table took twice the LDS):
  // a double rounded to float moves by 6e-8 of its size, far inside those paddings, so the survivors still include
  // every pair within the margin and the flags stay the brute-force sweep's.
  using C = typename std::conditional<CLDS, float, T>::type;
  constexpr bool kSameC = std::is_same<C, T>::value;
  const Views<T> V(S);
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6, n_wv = blockDim.x >> 6;
  const size_t n_waves = (n + kWave - 1) / kWave;
  const size_t wave_raw = (size_t)blockIdx.x * n_wv + wv;
  const bool wave_live = wave_raw < n_waves;  // the last workgroup may hold idle wavefronts
  const size_t wave = wave_live ? wave_raw : n_waves - 1;
  const size_t col = wave * kWave + lane;
  const bool live = wave_live && col < n;
  const size_t i = col < n ? col : n - 1;
  const T* qmem = q + i * (size_t)S.n_cols;

  T* rhot = reinterpret_cast<T*>(gear_lds + table_bytes);  // [L][4] half extents, kind
  const unsigned rhot_bytes = ((unsigned)S.n_shapes * 4u * (unsigned)sizeof(T) + 15u) & ~15u;
  unsigned char* my_lds = gear_lds + table_bytes + rhot_bytes + (size_t)wv * per_wave_lds;
  const size_t slot_bytes = (size_t)S.n_slots * 7 * kWave * sizeof(T);
  const PoseStore<T> slots{reinterpret_cast<T*>(my_lds), (size_t)kWave, (size_t)lane};
  T* const first_base = reinterpret_cast<T*>(my_lds + slot_bytes);  // [n_first][3][64] centres
  T* const fq_wave = W.fq + wave * (size_t)S.n_first * kWave * 4;    // [n_first][64][4] rotations (memory)
  T* const ring = reinterpret_cast<T*>(my_lds + slot_bytes + first_bytes);  // [kRing][8]
  unsigned* const hits = reinterpret_cast<unsigned*>(ring + (size_t)kRing * 8);  // [0..1] env bits of lanes 0-31 / 32-63, [2..3] self, [4..7] "has an exact entry", likewise
  unsigned* const hlist = hits + 8;  // [kHitList] the hits of half a cell block: owner lane << 16 | obstacle
  // this lane's row of the batch
  QLds<T> qlds{reinterpret_cast<const T*>(hlist + kHitList) + lane};  // Q_LDS: [n_cols][64] behind the hit list
  if (QMODE == Q_LDS) {
    T* w = reinterpret_cast<T*>(hlist + kHitList) + lane;
    for (int c = 0; c < S.n_cols; c++) w[c * kWave] = qmem[c];
  }
  auto qrow = [&]() -> decltype(auto) {
    if constexpr (QMODE == Q_LDS) return (const QLds<T>&)qlds;
    else return (const T* const&)qmem;
  };
  {
    if (CLDS) {
      C* tab = reinterpret_cast<C*>(gear_lds);
      for (int k = threadIdx.x; k < (S.n_obs + 1) * 4; k += blockDim.x) tab[k] = (C)S.ocull[k];
    }
    for (int k = threadIdx.x; k < S.n_shapes * 4; k += blockDim.x) rhot[k] = S.hot[k];
    if (lane < 8) hits[lane] = 0;
    __syncthreads();
  }
  const T* __restrict__ ohot = S.hot + (size_t)S.n_shapes * 4;  // obstacle part: 12 T per obstacle, gathered per item
  const int4* __restrict__ spairs4 = reinterpret_cast<const int4*>(S.spairs);
  const GEAR_CAS int32_t* sp_first_slot = to_cas(S.sp_first_slot);
  const GEAR_CAS int32_t* sp_late_off = to_cas(S.sp_late_off);
  const GEAR_CAS SpLate<T>* sp_late_rec = to_cas(S.sp_late_rec);
  const GEAR_CAS GridDev<T>* grids = to_cas(S.grids);
  const GEAR_CAS int32_t* shape_grid = to_cas(S.shape_grid);
  const uint4* __restrict__ blocks = reinterpret_cast<const uint4*>(S.cell_blk);
  const Cull4<C>* __restrict__ ocull = CLDS ? reinterpret_cast<const Cull4<C>*>(gear_lds) : reinterpret_cast<const Cull4<C>*>(S.ocull);
  const unsigned sentinel = (unsigned)S.n_obs;

  const bool ok = (what & kWhatNoLimits) != 0 || limits_ok(S, V, qrow());  // GH_CHECK_NO_LIMITS: geometry only
  const bool do_env = (what & 1u) && S.n_obs > 0;
  const bool do_self = (what & 2u) && S.n_spairs > 0;
  const bool push_ok = live && ok;  // the reference evaluates no collision after a limit error
  const bool dbg_no_sweep = GEAR_DBG(what, 0x800u);
  const T margin = S.margin;

  unsigned head = 0, count = 0;  // ring state, wave-uniform
  unsigned xcount = 0;           // exact entries written by this wavefront, wave-uniform
  bool xover = false;
  bool env_done = false, self_done = false;  // this lane's configuration is already known to collide
  unsigned long long ncand = 0;  // wave-uniform
  LaneCounters lc{0, 0, 0, 0};

  // ---- the narrow-phase site: decide up to 64 ring items, one per lane ----------------------------------
  auto drain = [&]() {
    wave_lds_sync();
    const unsigned take = count < (unsigned)kWave ? count : (unsigned)kWave;
    const bool have = (unsigned)lane < take;
    const Rec rc = ItemIO<T>::load(ring + (size_t)((head + (unsigned)lane) & (unsigned)(kRing - 1)) * 8);
    head = (head + take) & (unsigned)(kRing - 1);
    count -= take;
    const unsigned code = ItemIO<T>::code(rc);
    const Pose<T> mine = ItemIO<T>::pose(rc);
    const int ln = code & 63;
    const bool sf = (code >> 31) != 0;
    bool valid = have;
    if (valid) valid = ((hits[(sf ? 2 : 0) + (ln >> 5)] >> (ln & 31)) & 1u) == 0;  // owner already known to collide
    // a self item reads a rotation that another lane of this wavefront stored to PipeWs::fq earlier on
    if (__ballot(valid && sf) != 0) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    if (GEAR_DBG(what, 0x200u)) valid = false;  // profiling ablation
    int r = PR_FREE;
    unsigned cls = 0;
    Pose<T> wa = mine, wb = mine;
    if (valid) {
      ShapeRef<T> A, B;
      if (sf) {
        // the record holds the pose of the pair's LATER shape (the walk had it live); the earlier one's waits in
        // the owner lane's LDS column
        const int4 sp = spairs4[(code >> 6) & 0x1ffffffu];
        const int s1 = sp.x, s2 = sp.y;
        A = hot_ref(rhot[4 * s1], rhot[4 * s1 + 1], rhot[4 * s1 + 2], rhot[4 * s1 + 3]);
        B = hot_ref(rhot[4 * s2], rhot[4 * s2 + 1], rhot[4 * s2 + 2], rhot[4 * s2 + 3]);
        if (sp.w >= 0) {
          const T* b = first_base + (size_t)sp.w * 3 * kWave + ln;
          const Quad<T> oq = reinterpret_cast<const Quad<T>*>(fq_wave)[(size_t)sp.w * kWave + ln];
          const Pose<T> other{{b[0], b[kWave], b[2 * kWave]}, {oq.x, oq.y, oq.z, oq.w}};
          if (s1 > s2) wb = other; else wa = other;
        }
      } else {
        const int s = (code >> 6) & 1023, o = (code >> 16) & 32767;
        A = hot_ref(rhot[4 * s], rhot[4 * s + 1], rhot[4 * s + 2], rhot[4 * s + 3]);
        const T* ob = ohot + (size_t)o * 12;
        B = hot_ref(ob[8], ob[9], ob[10], ob[3]);
        wb = {{ob[0], ob[1], ob[2]}, {ob[4], ob[5], ob[6], ob[7]}};
      }
      r = pair_stage1<T, false>(A, wa, B, wb, margin, lc);
      cls = exact_class(A.kind, B.kind, sf);
      if (r == PR_HIT) atomicOr(&hits[(sf ? 2 : 0) + (ln >> 5)], 1u << (ln & 31));
    }
    // undecided items leave the chip, poses included, for the exact stage
    const bool ex = r == PR_EXACT;
    const unsigned long long eb = __ballot(ex);
    if (eb != 0) {
      const unsigned tot = (unsigned)__popcll(eb);
      if (xcount + tot > W.capw) {
        xover = true;  // the fused path re-checks this wavefront's configurations
      } else if (ex) {
        // A configuration's FIRST entry of a kind is marked as such: the exact stage decides those first and looks
        // at the others only where the first one did not hit (four of five entries hit on cylinder-rich scenes).
        const unsigned obit = 1u << (ln & 31);
        const unsigned later = (atomicOr(&hits[(sf ? 6 : 4) + (ln >> 5)], obit) & obit) ? kKeyLater : 0u;
        const size_t e = wave * (size_t)W.capw + xcount + lane_rank(eb);
        W.xkey[e] = make_uint2((cls << 28) | later | (unsigned)(wave * kWave + (size_t)ln), code);
        ItemIO<T>::store(W.xpose_a + e * 8, code, wa);
        if (sf) ItemIO<T>::store(W.xpose_b + e * 8, code, wb);
      }
      xcount += tot;
    }
    wave_lds_sync();
    env_done = ((hits[lane >> 5] >> (lane & 31)) & 1u) != 0;
    self_done = ((hits[2 + (lane >> 5)] >> (lane & 31)) & 1u) != 0;
  };

  const int n_walk = (do_env || do_self) ? S.n_shapes : 0;
  int node = -1;
  Pose<T> cur{{T(0), T(0), T(0)}, {T(0), T(0), T(0), T(1)}};
  for (int s = 0; s < n_walk; s++) {
    const GEAR_CAS ShapeDev<T>* sh = V.rshapes + s;
    const int target = sh->node;
    while (node < target) {
      node++;
      node_world<T, false>(V, node, qrow(), slots, cur);
    }
    const Pose<T> ws = pose_mul(cur, shape_pose<T>(sh));
    if (do_self) {
      // self shape pairs whose later shape is s: its pose is live, the other one's centre is read from LDS
      for (int e = sp_late_off[s]; e < sp_late_off[s + 1]; e++) {
        const GEAR_CAS SpLate<T>* lr = sp_late_rec + e;  // one 16-byte scalar load: pair, LDS slot of its earlier shape, radius sum
        const int p = lr->p;
        const int fs = lr->fs;  // -1: the pair is (s, s)
        V3<T> oc = ws.t;
        if (fs >= 0) {
          const T* b = first_base + (size_t)fs * 3 * kWave + lane;
          oc = {b[0], b[kWave], b[2 * kWave]};
        }
        const V3<T> d = ws.t - oc;  // (c1 - c2) or (c2 - c1): the squares are the same
        const T rr = lr->rr;
        const bool hit = push_ok && !self_done && dot(d, d) < rr * rr && !GEAR_DBG(what, 0x400u);
        const unsigned long long hb = __ballot(hit);
        if (hb == 0) continue;
        if (count >= (unsigned)kWave) drain();
        if (hit) ItemIO<T>::store(ring + (size_t)((head + count + lane_rank(hb)) & (unsigned)(kRing - 1)) * 8,
                                  kSelfItem | ((unsigned)p << 6) | (unsigned)lane, ws);
        count += (unsigned)__popcll(hb);
      }
      const int fs = sp_first_slot[s];
      if (fs >= 0) {
        T* b = first_base + (size_t)fs * 3 * kWave + lane;
        b[0] = ws.t.x; b[kWave] = ws.t.y; b[2 * kWave] = ws.t.z;
        if (wave_live) reinterpret_cast<Quad<T>*>(fq_wave)[(size_t)fs * kWave + lane] = Quad<T>{ws.r.i, ws.r.j, ws.r.k, ws.r.w};
      }
    }
    if (!do_env || dbg_no_sweep) continue;
    // ---- cell lookup -------------------------------------------------------------------------------
    unsigned blk = 0;  // block 0 is the empty list
    if (all_blk >= 0) {
      blk = (push_ok && !env_done) ? (unsigned)all_blk : 0u;  // GH_PATH_NO_GRID: the chain that lists every obstacle
    } else {
      const GEAR_CAS GridDev<T>* g = grids + shape_grid[s];
      const C fx = ((C)ws.t.x - (C)g->org[0]) * (C)g->inv_h, fy = ((C)ws.t.y - (C)g->org[1]) * (C)g->inv_h,
              fz = ((C)ws.t.z - (C)g->org[2]) * (C)g->inv_h;
      const bool inside = push_ok && !env_done && fx >= C(0) && fy >= C(0) && fz >= C(0) && fx < C(g->nx) && fy < C(g->ny) && fz < C(g->nz);
      if (inside) blk = (unsigned)g->off_base + ((unsigned)(int)fz * (unsigned)g->ny + (unsigned)(int)fy) * (unsigned)g->nx + (unsigned)(int)fx;
    }
    if (__ballot(blk != 0) == 0) continue;  // no lane of the wavefront has a list to look at
    const C rho = (C)sh->br;
    // the shape's oriented box: half extents padded by the rounding of the transformed offset (a ball: the
    // cube around it).  The offset is rotated with the pose quaternion itself: a rotation matrix would be
    // nine more registers live across the drain below.
    const bool is_ball = sh->kind == SK_BALL;
    const C pad = C(1e-5) * (C(1) + rho);
    const C hbx = (C)sh->half[0] + pad, hby = (C)(is_ball ? sh->half[0] : sh->half[1]) + pad, hbz = (C)(is_ball ? sh->half[0] : sh->half[2]) + pad;
    const Pose<C> wc{{(C)ws.t.x, (C)ws.t.y, (C)ws.t.z}, {(C)ws.r.i, (C)ws.r.j, (C)ws.r.k, (C)ws.r.w}};  // the shape's pose as the culls see it
    bool more;
    do {
      const uint4 w = blocks[blk];
      const bool cont = (w.w >> 31) != 0;
      unsigned id[8] = {w.x & 0xffffu, w.x >> 16, w.y & 0xffffu, w.y >> 16, w.z & 0xffffu, w.z >> 16, w.w & 0xffffu, w.w >> 16};
      if (cont) id[6] = id[7] = sentinel;
      Cull4<C> oc[8];
#pragma unroll
      for (int j = 0; j < 8; j++) oc[j] = ocull[id[j]];  // eight independent loads; a sentinel reads the far dummy
      unsigned h = 0;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const C dx = wc.t.x - oc[j].x, dy = wc.t.y - oc[j].y, dz = wc.t.z - oc[j].z;
        const C rr = rho + oc[j].r;
        h |= (dx * dx + dy * dy + dz * dz < rr * rr) ? (1u << j) : 0u;
      }
      if (what & 0x8000u) {  // diagnostic count of the candidates (wave-uniform: no register per lane)
#pragma unroll
        for (int j = 0; j < 8; j++) ncand += (unsigned)__popcll(__ballot(id[j] != sentinel));
      }
      if (env_done || GEAR_DBG(what, 0x400u)) h = 0;  // the configuration already collides: its flags are an OR
      // The hits of the wavefront's 64 lanes (one or two per lane, unevenly spread) are made dense before the
      // second cull: half a block at a time, every hit is listed as (owner lane, obstacle) in LDS, and the
      // wavefront then takes the list 64 entries per pass, one hit per lane — the owner's shape pose comes over
      // by lane shuffles.  (A loop of one pass per hit of the BUSIEST lane ran ~5 passes per shape at ~20 %
      // lane use; this runs ~2.)
#pragma unroll
      for (int half = 0; half < 2; half++) {
        const unsigned hh = (h >> (4 * half)) & 15u;
        if (__ballot(hh != 0) == 0) continue;
        unsigned tot = 0;  // wave-uniform
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const bool on = ((hh >> j) & 1u) != 0;
          const unsigned long long b = __ballot(on);
          // (the index is taken from the block's words again: eight unpacked indices would be eight registers live
          // across the narrow-phase site below)
          if (on) hlist[tot + lane_rank(b)] = ((unsigned)lane << 16) | (((half ? (j < 2 ? w.z : w.w) : (j < 2 ? w.x : w.y)) >> ((j & 1) << 4)) & 0xffffu);
          tot += (unsigned)__popcll(b);
        }
        wave_lds_sync();
        for (unsigned p0 = 0; p0 < tot; p0 += (unsigned)kWave) {
          const bool have = p0 + (unsigned)lane < tot;
          unsigned e = hlist[have ? p0 + (unsigned)lane : 0u];
          int src = (int)(e >> 16);
          unsigned idz = have ? (e & 0xffffu) : sentinel;
          // the owner's pose: as the cull sees it (f32 in both precisions when the table is in LDS) and, for the
          // record, exact
          Pose<T> sp;
          Pose<C> sc;
          if constexpr (kSameC) {
            sp = {{__shfl(ws.t.x, src), __shfl(ws.t.y, src), __shfl(ws.t.z, src)},
                  {__shfl(ws.r.i, src), __shfl(ws.r.j, src), __shfl(ws.r.k, src), __shfl(ws.r.w, src)}};
            sc = sp;
          } else {
            sc = {{__shfl(wc.t.x, src), __shfl(wc.t.y, src), __shfl(wc.t.z, src)},
                  {__shfl(wc.r.i, src), __shfl(wc.r.j, src), __shfl(wc.r.k, src), __shfl(wc.r.w, src)}};
          }
          bool pass = have;
          if (!GEAR_DBG(what, 0x200000u)) {
            const Cull4<C> c = ocull[idz];
            const C dx = c.x - sc.t.x, dy = c.y - sc.t.y, dz = c.z - sc.t.z;
            // R^T d = d + w t' + u x t' with u = -(i, j, k), t' = 2 u x d
            const C ui = -sc.r.i, uj = -sc.r.j, uk = -sc.r.k;
            const C tx = C(2) * (uj * dz - uk * dy), ty = C(2) * (uk * dx - ui * dz), tz = C(2) * (ui * dy - uj * dx);
            const C px = dx + sc.r.w * tx + (uj * tz - uk * ty), py = dy + sc.r.w * ty + (uk * tx - ui * tz), pz = dz + sc.r.w * tz + (ui * ty - uj * tx);
            const C ex = t_max(t_abs(px) - hbx, C(0)), ey = t_max(t_abs(py) - hby, C(0)), ez = t_max(t_abs(pz) - hbz, C(0));
            pass = have && (ex * ex + ey * ey + ez * ez < c.r * c.r);
          }
          if (__ballot(pass) == 0) continue;
          if (count >= (unsigned)kWave) {
            drain();
            // nothing of this pass but `pass` itself is kept across the narrow-phase site (it needs the registers):
            // entry and pose are fetched again
            e = hlist[have ? p0 + (unsigned)lane : 0u];
            src = (int)(e >> 16);
            idz = have ? (e & 0xffffu) : sentinel;
            if constexpr (kSameC)
              sp = {{__shfl(ws.t.x, src), __shfl(ws.t.y, src), __shfl(ws.t.z, src)},
                    {__shfl(ws.r.i, src), __shfl(ws.r.j, src), __shfl(ws.r.k, src), __shfl(ws.r.w, src)}};
            pass = pass && ((hits[src >> 5] >> (src & 31)) & 1u) == 0;  // the owner was decided meanwhile
          }
          if constexpr (!kSameC)
            sp = {{__shfl(ws.t.x, src), __shfl(ws.t.y, src), __shfl(ws.t.z, src)},
                  {__shfl(ws.r.i, src), __shfl(ws.r.j, src), __shfl(ws.r.k, src), __shfl(ws.r.w, src)}};
          const unsigned long long pb = __ballot(pass);
          if (pass) ItemIO<T>::store(ring + (size_t)((head + count + lane_rank(pb)) & (unsigned)(kRing - 1)) * 8,
                                     (idz << 16) | ((unsigned)s << 6) | (unsigned)src, sp);
          count += (unsigned)__popcll(pb);
        }
      }
      blk = (cont && !env_done) ? (w.w & 0x7fffffffu) : 0u;
      more = __ballot(blk != 0) != 0;
    } while (more);
  }
  while (count != 0) drain();

  if (what & 0x8000u) {
    if (lane == 0 && wave_live) {
      atomicAdd(&W.pc->candidates, ncand);
      atomicAdd(&W.pc->exact, (unsigned long long)xcount);
      if (xover) atomicAdd(&W.pc->redo_waves, 1u);
    }
  }
  wave_lds_sync();
  {
    const bool env_hit = ((hits[lane >> 5] >> (lane & 31)) & 1u) != 0;
    const bool self_hit = ((hits[2 + (lane >> 5)] >> (lane & 31)) & 1u) != 0;
    unsigned out = !ok ? 4u : ((env_hit ? 1u : 0u) | (self_hit ? 2u : 0u));
    if (xover) out |= kHitRedo;
    if (xover && lane == 0 && wave_live && W.note) __hip_atomic_store(W.note, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (live) W.hit[wave * kWave + lane] = out;
    if (lane == 0 && wave_live) W.xcnt[wave] = xover ? kXRedo : xcount;
  }
}

// ---------------------------------------------------------------------------
// k_narrow2: the exact stage.  Workgroup b takes the exact entries of kN2Tile consecutive sweep
// wavefronts.  In windows of at most kN2Window entries it (1) drops the entries whose configuration is
// already known to collide (the flags are an OR; on the metric scene that is most of them), (2) bins the
// rest by pair class with a counting sort in LDS, and (3) hands groups of 64 consecutive sorted entries to
// its wavefronts: the lanes of a group run the same support functions and the same parts of the
// hollow-mesh rule, except in the one group that straddles a class boundary.
// MLDS: the scene's mesh vertices and face planes sit in LDS as 16-byte records (SceneDev::mesh4), one
// access per vertex or plane.
// ---------------------------------------------------------------------------
constexpr int kN2Threads = 256;
#ifndef GEAR_N2_APART_MIN
#define GEAR_N2_APART_MIN (kN2Threads / 2)  // later entries per window from which they wait for the first ones (64 ... 512 measure alike)
#endif
constexpr int kN2Window = 4096;  // entries binned at a time

// kN2Tile: sweep wavefronts per workgroup (a power of two)
template <class T, bool MLDS, bool HIST = false, int kN2Tile = 32>
__global__ void __launch_bounds__(kN2Threads, sweep_occupancy<T>())  // f32: four wavefronts per SIMD (the stage lives on latency hiding)
k_narrow2(SceneDev<T> S, uint32_t what, PipeWs<T> W, unsigned n_sweep_waves) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gear_lds[];
  __shared__ unsigned sorted[kN2Window];
  __shared__ unsigned wpre[kN2Tile + 1];  // prefix of the tile's entry counts
  __shared__ unsigned cbin[kClasses], cfill[kClasses], coff[kClasses + 1];
  __shared__ unsigned cbin2[kClasses];  // the later entries' counts
  __shared__ unsigned nlater[3];        // later entries of the window that wait: listed, still needed; do they wait at all
  const Views<T> V(S);
  const int tid = threadIdx.x;
  const unsigned w0 = blockIdx.x * (unsigned)kN2Tile;
  const unsigned nw = min((unsigned)kN2Tile, n_sweep_waves - w0);
  if (tid == 0) {
    unsigned acc = 0;
    for (unsigned k = 0; k < nw; k++) {
      const unsigned x = W.xcnt[w0 + k];
      wpre[k] = acc;
      acc += x == kXRedo ? 0u : min(x, W.capw);  // a full region's wavefront is evaluated again by k_check
    }
    for (unsigned k = nw; k <= (unsigned)kN2Tile; k++) wpre[k] = acc;
  }
  __syncthreads();
  const unsigned total = wpre[kN2Tile];
  if (total != 0) {  // uniform
  const Vec4Tab<T> lverts{reinterpret_cast<const Quad<T>*>(gear_lds)};
  const Vec4Tab<T> lplanes{reinterpret_cast<const Quad<T>*>(gear_lds) + S.n_verts};
  if (MLDS) {
    const Quad<T>* src = reinterpret_cast<const Quad<T>*>(S.mesh4);
    Quad<T>* dst = reinterpret_cast<Quad<T>*>(gear_lds);
    for (int k = tid; k < S.n_verts + S.n_planes; k += kN2Threads) dst[k] = src[k];
  }
  const T margin = S.margin;
  LaneCounters lc{0, 0, 0, 0};
  const size_t ebase = (size_t)w0 * W.capw;
  // flattened entry k of the tile -> offset into the exact arrays, relative to the tile's first entry
  auto entry_of = [&](unsigned k) -> unsigned {
    unsigned w = 0;
#pragma unroll
    for (int step = kN2Tile / 2; step > 0; step >>= 1)
      if (wpre[w + step] <= k) w += step;
    return w * W.capw + (k - wpre[w]);
  };
  for (unsigned k0 = 0; k0 < total; k0 += (unsigned)kN2Window) {
    const unsigned kend = min(total, k0 + (unsigned)kN2Window);
    if (tid < kClasses) { cbin[tid] = 0; cbin2[tid] = 0; cfill[tid] = 0; }
    if (tid < 3) nlater[tid] = 0;
    __syncthreads();  // also: the mesh table is in
    // (1) + (2a): which entries still matter, how many per class — the configurations' first entries and the later
    // ones counted apart
    for (unsigned k = k0 + (unsigned)tid; k < kend; k += kN2Threads) {
      const uint2 key = W.xkey[ebase + entry_of(k)];
      const unsigned bit = (key.y >> 31) ? 2u : 1u;
      if ((W.hit[key.x & kKeyCfg] & (bit | kHitRedo)) == 0 && !GEAR_DBG(what, 0x100u))
        atomicAdd((key.x & kKeyLater) ? &cbin2[key.x >> 28] : &cbin[key.x >> 28], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      // Later entries wait for the first ones' answers when there are enough of them to pay for a second round of
      // queries (a round is one query per thread); otherwise everything is decided in one go, as before round 3.
      // (Metric scene: a few later entries per window, one go.  6-DOF arm in a field of cylinders: 1.9 entries per
      // configuration, five of six hit: the exact stage 287 -> 140 us per 1M, the call 2.05 -> 2.87 G configs/s.)
      unsigned later = 0;
      for (int c = 0; c < kClasses; c++) later += cbin2[c];
      const bool apart = later >= (unsigned)GEAR_N2_APART_MIN;
      unsigned acc = 0;
      for (int c = 0; c < kClasses; c++) { coff[c] = acc; acc += cbin[c] + (apart ? 0u : cbin2[c]); }
      coff[kClasses] = acc;
      nlater[2] = apart ? 1u : 0u;
    }
    __syncthreads();
    const bool apart = nlater[2] != 0;  // uniform
    // (2b): place them: by class at the near end of sorted[]; later entries that wait, from the far end (every entry
    // of the window goes to one list or the other)
    for (unsigned k = k0 + (unsigned)tid; k < kend; k += kN2Threads) {
      const unsigned eo = entry_of(k);
      const uint2 key = W.xkey[ebase + eo];
      const unsigned bit = (key.y >> 31) ? 2u : 1u;
      if ((W.hit[key.x & kKeyCfg] & (bit | kHitRedo)) == 0 && !GEAR_DBG(what, 0x100u)) {
        if (apart && (key.x & kKeyLater)) {
          sorted[kN2Window - 1 - atomicAdd(&nlater[0], 1u)] = eo;
        } else {
          const unsigned c = key.x >> 28;
          sorted[coff[c] + atomicAdd(&cfill[c], 1u)] = eo;
        }
      }
    }
    __syncthreads();
    // (3): groups of 64 listed entries, round robin over the wavefronts — the first entries, then what is left of the
    // later ones
    for (int phase = 0; phase < 2; phase++) {
      unsigned kept = coff[kClasses];
      bool back = false;  // the list runs down from the far end of sorted[]
      if (phase == 1) {
        const unsigned n_later = nlater[0];
        if (n_later == 0) break;  // uniform
        // the first entries' hits (device-scope atomics of this workgroup's wavefronts) decide which later entries
        // are still needed: each wavefront waits for its own atomics, barrier, device-scope atomic loads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        if (n_later <= (unsigned)kN2Window / 2) {  // made dense at the near end (the two regions cannot meet)
          for (unsigned j = (unsigned)tid; j < n_later; j += kN2Threads) {
            const unsigned eo = sorted[kN2Window - 1 - j];
            const uint2 key = W.xkey[ebase + eo];
            const unsigned bit = (key.y >> 31) ? 2u : 1u;
            const unsigned h = __hip_atomic_load(W.hit + (key.x & kKeyCfg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((h & bit) == 0) sorted[atomicAdd(&nlater[1], 1u)] = eo;
          }
          __syncthreads();
          kept = nlater[1];
        } else {  // (a window of mostly later entries: taken where they lie, each lane looking at its hit word)
          kept = n_later;
          back = true;
        }
      }
    for (unsigned g = (unsigned)(tid >> 6) * kWave; g < kept; g += kN2Threads) {
      const unsigned j = g + (unsigned)(tid & 63);
      if (j >= kept) continue;
      size_t e = ebase + sorted[back ? kN2Window - 1 - j : j];
      if (GEAR_DBG(what, 0x400000u)) e = ebase + sorted[g];  // profiling ablation: one item per wave, no divergence
      const uint2 key = W.xkey[e];
      const unsigned code = key.y;
      const size_t cfg = key.x & kKeyCfg;
      const bool sf = (code >> 31) != 0;
      if (back && (__hip_atomic_load(W.hit + cfg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (sf ? 2u : 1u))) continue;
      ShapeRef<T> A, B;
      Pose<T> wb;
      const Pose<T> wa = ItemIO<T>::pose(W.xpose_a + e * 8);
      if (sf) {
        const int pi = (code >> 6) & 0x1ffffff;
        const int s1 = V.spairs[4 * pi], s2 = V.spairs[4 * pi + 1];
        A = shape_ref<T>(V.rshapes + s1);
        B = shape_ref<T>(V.rshapes + s2);
        wb = ItemIO<T>::pose(W.xpose_b + e * 8);
      } else {
        const int s = (code >> 6) & 1023, o = (code >> 16) & 32767;
        A = shape_ref<T>(V.rshapes + s);
        B = shape_ref<T>(V.oshapes + o);
        wb = shape_pose<T>(V.oshapes + o);
      }
      bool hit;
      if (MLDS) hit = pair_stage2<T, HIST>(A, wa, B, wb, lverts, lplanes, margin, lc);
      else hit = pair_stage2<T, HIST>(A, wa, B, wb, V.verts, V.planes, margin, lc);
      if (hit) atomicOr(&W.hit[cfg], sf ? 2u : 1u);
      if (HIST) {
        atomicAdd(&W.pc->hist[0][min(lc.exact_iters, 47u)], 1u);
        unsigned mx = lc.exact_iters;
        for (int d = 32; d > 0; d >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, d));
        if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(__ballot(1))) atomicAdd(&W.pc->hist[1][min(mx, 47u)], 1u);
        lc.exact_iters = 0;
      }
    }
    }  // phase
    __syncthreads();  // sorted[] is reused by the next window
  }
  }
  // Every hit of the tile's configurations is in (their entries all belong to this workgroup): hit word -> flag
  // byte, eight configurations per thread: one pass for a tile of 32 wavefronts.  (Until round 3 a launch of its
  // own did this: 21 us per 4M configurations, all of it workgroup dispatch.)  The words were updated by atomics of
  // this workgroup's wavefronts (performed in the L2): each wavefront waits for its own before the barrier, and the
  // loads behind it are device-scope atomic loads, which do not look at this compute unit's cached copies.  (A
  // device-scope fence instead would write the L2 back per workgroup: measured, 3x the kernel.)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  const size_t c0 = (size_t)w0 * kWave;
  const size_t c1 = min((size_t)(w0 + nw) * kWave, (size_t)W.n_cfg);
  const bool wide = (reinterpret_cast<uintptr_t>(W.flags) & 7u) == 0;  // uniform; chunk offsets are multiples of 64
  auto to_flag = [](unsigned h) -> unsigned { return (h & 4u) ? 4u : (h & 3u); };
  for (size_t c = c0 + (size_t)tid * 8; c < c1; c += (size_t)kN2Threads * 8) {
    if (c + 8 <= c1) {
      unsigned h[8];
#pragma unroll
      for (int k = 0; k < 8; k++) h[k] = __hip_atomic_load(W.hit + c + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned lo = to_flag(h[0]) | (to_flag(h[1]) << 8) | (to_flag(h[2]) << 16) | (to_flag(h[3]) << 24);
      const unsigned hi = to_flag(h[4]) | (to_flag(h[5]) << 8) | (to_flag(h[6]) << 16) | (to_flag(h[7]) << 24);
      if (wide) {
        *reinterpret_cast<uint2*>(W.flags + c) = make_uint2(lo, hi);
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) { W.flags[c + k] = (uint8_t)(lo >> (8 * k)); W.flags[c + 4 + k] = (uint8_t)(hi >> (8 * k)); }
      }
    } else {
      for (size_t k = c; k < c1; k++) W.flags[k] = (uint8_t)to_flag(__hip_atomic_load(W.hit + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
  }
}

}  // namespace gear_hip
