// oracle_capi.cpp — C entry points of the CPU ORACLE (test infrastructure only;
// see gear_oracle.hpp).  Loaded with ctypes by tests/, smoke() and bench.py's
// cpu_baseline leg; never by the product under gear_amd/.
#include <atomic>
#include <chrono>
#include <thread>

#include "gear_oracle.hpp"

using namespace gear_oracle;

struct OrcScene {
  int precision;
  Scene<float>* s32 = nullptr;
  Scene<double>* s64 = nullptr;
};

template <class F>
static void parallel_for(size_t n, int threads, F&& body) {
  if (threads <= 1 || n < 2) {
    body(0, n, 0);
    return;
  }
  std::vector<std::thread> pool;
  size_t chunk = (n + threads - 1) / threads;
  for (int t = 0; t < threads; t++) {
    size_t lo = std::min(n, chunk * t), hi = std::min(n, chunk * (t + 1));
    if (lo >= hi) break;
    pool.emplace_back([=, &body] { body(lo, hi, t); });
  }
  for (auto& th : pool) th.join();
}

template <class T>
static void do_check(const Scene<T>& s, const T* q, size_t n, uint32_t what, int faithful,
                     int redundant_fk, int threads, uint8_t* flags, double* band, uint64_t* stats) {
  int D = s.dof();
  std::vector<Stats> st(std::max(threads, 1));
  parallel_for(n, threads, [&](size_t lo, size_t hi, int t) {
    for (size_t i = lo; i < hi; i++) {
      T b[2];
      flags[i] = s.check_config(q + i * D, what, faithful != 0, redundant_fk != 0,
                                band ? b : nullptr, stats ? &st[t] : nullptr);
      if (band) { band[2 * i] = (double)b[0]; band[2 * i + 1] = (double)b[1]; }
    }
  });
  if (stats) {
    stats[0] = stats[1] = stats[2] = 0;
    for (auto& x : st) {
      stats[0] += x.pair_tests;
      stats[1] += x.gjk_calls;
      stats[2] += x.gjk_iters;
    }
  }
}

template <class T>
static void do_fk(const Scene<T>& s, const T* q, size_t n, T* poses) {
  int D = s.dof();
  size_t N = s.nodes.size();
  std::vector<T> pos;
  std::vector<Iso<T>> world;
  for (size_t i = 0; i < n; i++) {
    s.initial_positions(pos);
    s.set_joint_positions_unchecked(q + i * D, pos);
    s.update_transforms(pos, world);
    for (size_t k = 0; k < N; k++) {
      T* o = poses + (i * N + k) * 7;
      o[0] = world[k].t.x; o[1] = world[k].t.y; o[2] = world[k].t.z;
      o[3] = world[k].r.i; o[4] = world[k].r.j; o[5] = world[k].r.k; o[6] = world[k].r.w;
    }
  }
}

// colliding_link_names (joint_path_planner.rs:100-110): per obstacle, the
// DFS-ordered nodes check_env yields, as bit masks [n][O][W].
template <class T>
static void do_link_hits(const Scene<T>& s, const T* q, size_t n, uint32_t what, uint64_t* masks) {
  int D = s.dof();
  size_t O = s.obstacles.size();
  size_t W = (s.nodes.size() + 63) / 64;
  std::vector<T> pos;
  std::vector<Iso<T>> world;
  std::memset(masks, 0, n * O * W * sizeof(uint64_t));
  for (size_t i = 0; i < n; i++) {
    s.initial_positions(pos);
    if (what & GH_CHECK_NO_LIMITS) s.set_joint_positions_unchecked(q + i * D, pos);
    else if (!s.set_joint_positions(q + i * D, pos)) continue;
    for (size_t o = 0; o < O; o++) {
      s.update_transforms(pos, world);
      uint64_t* m = masks + (i * O + o) * W;
      s.check_env(world, s.obstacles[o], [&](int ni) { m[ni >> 6] |= 1ull << (ni & 63); return true; },
                  nullptr, nullptr);
    }
  }
}

// colliding_link_names_with_self (joint_path_planner.rs:130-134)
template <class T>
static void do_self_hits(const Scene<T>& s, const T* q, size_t n, uint32_t what, uint64_t* masks) {
  int D = s.dof();
  size_t W = (s.pairs.size() + 63) / 64;
  std::vector<T> pos;
  std::vector<Iso<T>> world;
  std::memset(masks, 0, n * W * sizeof(uint64_t));
  for (size_t i = 0; i < n; i++) {
    s.initial_positions(pos);
    if (what & GH_CHECK_NO_LIMITS) s.set_joint_positions_unchecked(q + i * D, pos);
    else if (!s.set_joint_positions(q + i * D, pos)) continue;
    s.update_transforms(pos, world);
    uint64_t* m = masks + i * W;
    s.check_self(world, [&](int p) { m[p >> 6] |= 1ull << (p & 63); return true; }, nullptr, nullptr);
  }
}

extern "C" {

void* orc_scene_create(const gh_robot_desc* r, const gh_obstacle_desc* o, double margin, int precision) {
  if (!r || !o || (precision != 32 && precision != 64)) return nullptr;
  OrcScene* s = new OrcScene;
  s->precision = precision;
  if (precision == 32) s->s32 = new Scene<float>(*r, *o, margin);
  else s->s64 = new Scene<double>(*r, *o, margin);
  return s;
}

void orc_scene_destroy(void* h) {
  OrcScene* s = (OrcScene*)h;
  if (!s) return;
  delete s->s32;
  delete s->s64;
  delete s;
}

int orc_check(void* h, const void* q, size_t n, uint32_t what, int faithful, int redundant_fk,
              int threads, uint8_t* flags, double* band, uint64_t* stats) {
  OrcScene* s = (OrcScene*)h;
  if (!s) return -1;
  if (s->precision == 32) do_check(*s->s32, (const float*)q, n, what, faithful, redundant_fk, threads, flags, band, stats);
  else do_check(*s->s64, (const double*)q, n, what, faithful, redundant_fk, threads, flags, band, stats);
  return 0;
}

// Diagnostic for tests/parity.py: the close or uncertified pairs of ONE configuration (Scene::explain_config).
// out: records of 9 doubles {is_self, id_a, id_b, kind_a, kind_b, d, lower, gjk iterations, GjkExit}; returns
// the number of records found (at most `cap` are written).
int orc_explain(void* h, const void* q_row, uint32_t what, double window, double* out, int cap) {
  OrcScene* s = (OrcScene*)h;
  if (!s) return -1;
  int n = 0;
  auto emit = [&](int self, int ida, int idb, int ka, int kb, double d, double lo, int iters, int exit) {
    if (n < cap) {
      double* r = out + 9 * (size_t)n;
      r[0] = self; r[1] = ida; r[2] = idb; r[3] = ka; r[4] = kb; r[5] = d; r[6] = lo; r[7] = iters; r[8] = exit;
    }
    n++;
  };
  if (s->precision == 32) s->s32->explain_config((const float*)q_row, what, (float)window, emit);
  else s->s64->explain_config((const double*)q_row, what, window, emit);
  return n;
}

// first infeasible sample per edge, sequentially, stopping at the first failure
// exactly like rrt's stepping with the is_feasible closure
// (joint_path_planner.rs:168-188).  flags of samples after the first failure of
// an edge are left 0xFF (= never evaluated by the reference).
int orc_check_edges(void* h, const void* q, const uint32_t* edge_off, size_t n_edges, uint32_t what,
                    uint8_t bad_mask, uint8_t* flags, int32_t* first_bad) {
  OrcScene* s = (OrcScene*)h;
  if (!s) return -1;
  for (size_t e = 0; e < n_edges; e++) {
    first_bad[e] = -1;
    for (uint32_t i = edge_off[e]; i < edge_off[e + 1]; i++) flags[i] = 0xFF;
    for (uint32_t i = edge_off[e]; i < edge_off[e + 1]; i++) {
      uint8_t f;
      if (s->precision == 32)
        f = s->s32->check_config((const float*)q + (size_t)i * s->s32->dof(), what, true, true, nullptr, nullptr);
      else
        f = s->s64->check_config((const double*)q + (size_t)i * s->s64->dof(), what, true, true, nullptr, nullptr);
      flags[i] = f;
      if (f & bad_mask) {
        first_bad[e] = (int32_t)(i - edge_off[e]);
        break;
      }
    }
  }
  return 0;
}

int orc_fk(void* h, const void* q, size_t n, void* poses) {
  OrcScene* s = (OrcScene*)h;
  if (!s) return -1;
  if (s->precision == 32) do_fk(*s->s32, (const float*)q, n, (float*)poses);
  else do_fk(*s->s64, (const double*)q, n, (double*)poses);
  return 0;
}

int orc_link_hits(void* h, const void* q, size_t n, uint32_t what, uint64_t* masks) {
  OrcScene* s = (OrcScene*)h;
  if (!s) return -1;
  if (s->precision == 32) do_link_hits(*s->s32, (const float*)q, n, what, masks);
  else do_link_hits(*s->s64, (const double*)q, n, what, masks);
  return 0;
}

int orc_self_hits(void* h, const void* q, size_t n, uint32_t what, uint64_t* masks) {
  OrcScene* s = (OrcScene*)h;
  if (!s) return -1;
  if (s->precision == 32) do_self_hits(*s->s32, (const float*)q, n, what, masks);
  else do_self_hits(*s->s64, (const double*)q, n, what, masks);
  return 0;
}

// ncollide3d::query::distance between two described shapes (world poses in
// the descriptors' t/q).  For analytic known-answer tests.
double orc_distance(const gh_shape_desc* a, const gh_mesh_desc* meshes_a, const gh_shape_desc* b,
                    const gh_mesh_desc* meshes_b, int precision) {
  if (precision == 32) {
    Shape<float> sa = make_shape<float>(*a, meshes_a), sb = make_shape<float>(*b, meshes_b);
    return (double)distance(make_iso<float>(a->t, a->q), sa, make_iso<float>(b->t, b->q), sb, (Stats*)nullptr);
  }
  Shape<double> sa = make_shape<double>(*a, meshes_a), sb = make_shape<double>(*b, meshes_b);
  return distance(make_iso<double>(a->t, a->q), sa, make_iso<double>(b->t, b->q), sb, (Stats*)nullptr);
}

// same, also returning the run's certified lower bound (true distance in [*lower, result])
double orc_distance_bounds(const gh_shape_desc* a, const gh_mesh_desc* meshes_a, const gh_shape_desc* b,
                           const gh_mesh_desc* meshes_b, int precision, double* lower) {
  if (precision == 32) {
    Shape<float> sa = make_shape<float>(*a, meshes_a), sb = make_shape<float>(*b, meshes_b);
    float lo = 0;
    float d = distance(make_iso<float>(a->t, a->q), sa, make_iso<float>(b->t, b->q), sb, (Stats*)nullptr, &lo);
    *lower = lo;
    return d;
  }
  Shape<double> sa = make_shape<double>(*a, meshes_a), sb = make_shape<double>(*b, meshes_b);
  return distance(make_iso<double>(a->t, a->q), sa, make_iso<double>(b->t, b->q), sb, (Stats*)nullptr, lower);
}

// how the calling thread's last GJK run ended (gear_oracle::GjkExit) and after how many iterations
void orc_last_gjk(int* iters, int* exit_reason) {
  *iters = last_gjk().iters;
  *exit_reason = last_gjk().exit;
}

// closest point to the origin on a simplex of n (1..4) points, through the
// same Voronoi-region code GJK uses (unit tests vs brute force).
int orc_project_simplex(const double* pts, int n, double* out) {
  V3<double> p[4];
  for (int i = 0; i < n; i++) p[i] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
  Loc<double> loc;
  V3<double> r;
  V3<double> o{0, 0, 0};
  if (n == 1) r = p[0];
  else if (n == 2) r = project_segment(p[0], p[1], o, loc);
  else if (n == 3) r = project_triangle(p[0], p[1], p[2], o, loc);
  else r = project_tetra(p[0], p[1], p[2], p[3], o, loc);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
  return n == 1 ? 0 : (int)loc.kind;
}

// vertices of the cylinder TriMesh as generated for GH_SHAPE_CYLINDER
int orc_cylinder_vertices(double radius, double length, int precision, double* out /*[60][3]*/) {
  if (precision == 32) {
    Shape<float> s = make_cylinder<float>(radius, length);
    for (size_t i = 0; i < s.verts.size(); i++) { out[3 * i] = s.verts[i].x; out[3 * i + 1] = s.verts[i].y; out[3 * i + 2] = s.verts[i].z; }
    return (int)s.verts.size();
  }
  Shape<double> s = make_cylinder<double>(radius, length);
  for (size_t i = 0; i < s.verts.size(); i++) { out[3 * i] = s.verts[i].x; out[3 * i + 1] = s.verts[i].y; out[3 * i + 2] = s.verts[i].z; }
  return (int)s.verts.size();
}

// UnitQuaternion::from_euler_angles(r, p, y) in f64 (k::urdf::quaternion_from)
void orc_quat_from_rpy(double r, double p, double y, double* out) {
  Quat<double> q = q_euler<double>(r, p, y);
  out[0] = q.i; out[1] = q.j; out[2] = q.k; out[3] = q.w;
}

int orc_hardware_threads(void) { return (int)std::thread::hardware_concurrency(); }

}  // extern "C"
