// gear_device.hpp — device-side geometry for gfx950: pose algebra, forward
// kinematics step, and the narrow-phase "is distance < margin" predicates.
//
// What it answers is exactly the reference's test
//     query::distance(link_pose, link_shape, target_pose, target_shape) < prediction
// (src/collision/collision_checker.rs:83-92, 177-185) but NOT by running
// ncollide's GJK to a distance value: the kernel only needs the boolean, so it
// uses closed forms (ball pairs), a 15-axis separating-axis test with margin
// (box pairs) and, for what is left, a GJK *margin query* that stops as soon as
// a lower bound reaches the margin or an upper bound drops below it.  The
// reference's distance carries GJK's relative tolerance sqrt(10*eps); the two
// can therefore disagree only when |d - margin| is inside that tolerance band
// (DESIGN.md "Parity band").
#pragma once
#include <hip/hip_runtime.h>

#include "gear_types.hpp"

namespace gear_hip {

// __host__ too: tests/host_harness.cpp compiles these same functions for the CPU so
// that the geometry can be unit-tested without a GPU; the product never calls them on the host.
#define GD __host__ __device__ __forceinline__

template <class T> struct V3 { T x, y, z; };
template <class T> GD V3<T> operator+(V3<T> a, V3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class T> GD V3<T> operator-(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class T> GD V3<T> operator-(V3<T> a) { return {-a.x, -a.y, -a.z}; }
template <class T> GD V3<T> operator*(V3<T> a, T s) { return {a.x * s, a.y * s, a.z * s}; }
template <class T> GD T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T> GD V3<T> cross(V3<T> a, V3<T> b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

template <class T> struct Q4 { T i, j, k, w; };
template <class T> struct Pose { V3<T> t; Q4<T> r; };

template <class T> GD T t_abs(T x) { return x < T(0) ? -x : x; }
GD float t_copysign(float m, float s) { return copysignf(m, s); }
GD double t_copysign(double m, double s) { return copysign(m, s); }
GD float t_sqrt(float x) { return sqrtf(x); }
GD double t_sqrt(double x) { return sqrt(x); }
// f32 joints take sin/cos through f64 and round once: the f32 reference (glibc/LLVM sinf, ~0.5 ulp)
// and ocml's sincosf (~1-2 ulp) otherwise drift apart by a few 1e-7 over a 7-joint chain, which is
// the whole 1e-6 pose budget.  7 f64 sincos per configuration are noise next to the pair tests.
// The f64 evaluation only has to be good to ~1e-12 (it is rounded to f32 at once), so it is a short
// one: two-term Cody-Waite reduction by pi/2 and the fdlibm kernel polynomials on [-pi/4, pi/4]
// (|error| < 1e-15 there; the reduction adds < |x| * 2e-16).  Huge or non-finite arguments, which no
// joint limit produces, take the library routine.
GD void t_sincos(float x, float* s, float* c) {
  const double xd = (double)x;
  if (!(fabs(xd) < 1.0e5)) {
    double sd, cd;
    sincos(xd, &sd, &cd);
    *s = (float)sd;
    *c = (float)cd;
    return;
  }
  const double kf = rint(xd * 0.63661977236758134308);  // x * 2/pi
  double r = fma(-kf, 1.57079632679489655800e+00, xd);
  r = fma(-kf, 6.12323399573676603587e-17, r);
  const double z = r * r;
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(z, ps, 2.75573137070700676789e-06);
  ps = fma(z, ps, -1.98412698298579493134e-04);
  ps = fma(z, ps, 8.33333333332248946124e-03);
  ps = fma(z, ps, -1.66666666666666324348e-01);
  double sn = fma(z * r, ps, r);
  sn = xd == 0.0 ? xd : sn;  // sin(-0) = -0, as the library returns it
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(z, pc, -2.75573143513906633035e-07);
  pc = fma(z, pc, 2.48015872894767294178e-05);
  pc = fma(z, pc, -1.38888888888741095749e-03);
  pc = fma(z, pc, 4.16666666666666019037e-02);
  const double cs = fma(z * z, pc, fma(z, -0.5, 1.0));
  const int k = (int)kf;
  const double ss = (k & 1) ? cs : sn, cc = (k & 1) ? sn : cs;
  *s = (float)((k & 2) ? -ss : ss);
  *c = (float)(((k + 1) & 2) ? -cc : cc);
}
GD void t_sincos(double x, double* s, double* c) { sincos(x, s, c); }
GD float t_atan2(float y, float x) { return atan2f(y, x); }
GD double t_atan2(double y, double x) { return atan2(y, x); }
GD float t_rint(float x) { return rintf(x); }
GD double t_rint(double x) { return rint(x); }
template <class T> GD T t_max(T a, T b) { return a > b ? a : b; }
template <class T> GD T t_min(T a, T b) { return a < b ? a : b; }

// ---- pose algebra: nalgebra's Isometry3 * Isometry3, every product-sum written as an explicit fma chain.
// Explicit, not left to the compiler's contraction: the chain is then the same instruction sequence in every
// kernel that walks a robot (k_small, the sweep, the fused kernel, k_fk), which is what makes their poses and
// flags bit-identical, and it is 37 vector instructions per composition instead of the 61 of the separate
// multiplies and adds the reference's f32 arithmetic performs.  The rounding differs from the reference's by
// at most half an ulp per fused operation (the fused form is the more accurate one); DESIGN.md "Pose algebra"
// has the measured pose deviation against the oracle, which keeps the reference's unfused sequence.
#pragma clang fp contract(off)
GD float t_fma(float a, float b, float c) { return fmaf(a, b, c); }
GD double t_fma(double a, double b, double c) { return fma(a, b, c); }
template <class T> GD Q4<T> qmul(Q4<T> a, Q4<T> b) {
  return {t_fma(a.w, b.i, t_fma(a.i, b.w, t_fma(a.j, b.k, -(a.k * b.j)))),
          t_fma(a.w, b.j, t_fma(-a.i, b.k, t_fma(a.j, b.w, a.k * b.i))),
          t_fma(a.w, b.k, t_fma(a.i, b.j, t_fma(-a.j, b.i, a.k * b.w))),
          t_fma(a.w, b.w, t_fma(-a.i, b.i, t_fma(-a.j, b.j, -(a.k * b.k))))};
}
template <class T> GD Q4<T> qconj(Q4<T> a) { return {-a.i, -a.j, -a.k, a.w}; }
// v + w t + u x t with u = (i, j, k), t = 2 u x v
template <class T> GD V3<T> qrot(Q4<T> q, V3<T> v) {
  const T cx = t_fma(q.j, v.z, -(q.k * v.y)), cy = t_fma(q.k, v.x, -(q.i * v.z)), cz = t_fma(q.i, v.y, -(q.j * v.x));
  const T tx = cx + cx, ty = cy + cy, tz = cz + cz;
  return {t_fma(q.j, tz, t_fma(-q.k, ty, t_fma(q.w, tx, v.x))),
          t_fma(q.k, tx, t_fma(-q.i, tz, t_fma(q.w, ty, v.y))),
          t_fma(q.i, ty, t_fma(-q.j, tx, t_fma(q.w, tz, v.z)))};
}
template <class T> GD Pose<T> pose_mul(const Pose<T>& a, const Pose<T>& b) {
  V3<T> shift = qrot(a.r, b.t);
  return {a.t + shift, qmul(a.r, b.r)};
}

// One step of k::Chain::update_transforms: world = parent * (origin * joint(q)).
// (s, c) = sin, cos of position / 2 for a revolute joint (unused otherwise).
// origin * joint(q): the node's transform relative to its parent
template <class T>
GD Pose<T> fk_local_sc(V3<T> ot, Q4<T> oq, V3<T> axis, int joint_type, T position, T s, T c) {
  Pose<T> local{ot, oq};
  if (joint_type == JT_ROT) {
    Q4<T> jq{axis.x * s, axis.y * s, axis.z * s, c};
    local.r = qmul(oq, jq);
  } else if (joint_type == JT_LIN) {
    local.t = ot + qrot(oq, axis * position);
  }
  return local;
}
template <class T>
GD Pose<T> fk_step_sc(const Pose<T>& parent, V3<T> ot, Q4<T> oq, V3<T> axis, int joint_type, T position, T s, T c) {
  return pose_mul(parent, fk_local_sc(ot, oq, axis, joint_type, position, s, c));
}
template <class T>
GD Pose<T> fk_step(const Pose<T>& parent, V3<T> ot, Q4<T> oq, V3<T> axis, int joint_type, T position) {
  T s = T(0), c = T(1);
  if (joint_type == JT_ROT) t_sincos(position / T(2), &s, &c);
  return fk_step_sc(parent, ot, oq, axis, joint_type, position, s, c);
}
#pragma clang fp contract(fast)

// ---- relative frame: x_A = R x_B + t ----------------------------------------
template <class T> struct Rel { T r[9]; V3<T> t; };

template <class T> GD Rel<T> make_rel(const Pose<T>& a, const Pose<T>& b) {
  Q4<T> ac = qconj(a.r);
  Q4<T> q = qmul(ac, b.r);
  Rel<T> rel;
  rel.t = qrot(ac, b.t - a.t);
  T ii = q.i * q.i, jj = q.j * q.j, kk = q.k * q.k;
  T ij = q.i * q.j, ik = q.i * q.k, jk = q.j * q.k;
  T iw = q.i * q.w, jw = q.j * q.w, kw = q.k * q.w;
  rel.r[0] = T(1) - T(2) * (jj + kk); rel.r[1] = T(2) * (ij - kw);        rel.r[2] = T(2) * (ik + jw);
  rel.r[3] = T(2) * (ij + kw);        rel.r[4] = T(1) - T(2) * (ii + kk); rel.r[5] = T(2) * (jk - iw);
  rel.r[6] = T(2) * (ik - jw);        rel.r[7] = T(2) * (jk + iw);        rel.r[8] = T(1) - T(2) * (ii + jj);
  return rel;
}
template <class T> GD V3<T> rel_mul(const Rel<T>& m, V3<T> v) {  // R v
  return {m.r[0] * v.x + m.r[1] * v.y + m.r[2] * v.z, m.r[3] * v.x + m.r[4] * v.y + m.r[5] * v.z,
          m.r[6] * v.x + m.r[7] * v.y + m.r[8] * v.z};
}
template <class T> GD V3<T> rel_tmul(const Rel<T>& m, V3<T> v) {  // R^T v
  return {m.r[0] * v.x + m.r[3] * v.y + m.r[6] * v.z, m.r[1] * v.x + m.r[4] * v.y + m.r[7] * v.z,
          m.r[2] * v.x + m.r[5] * v.y + m.r[8] * v.z};
}

// ---- shapes as seen by the narrow phase -----------------------------------
template <class T> struct ShapeRef {
  int kind;
  V3<T> half;  // box half extents; ball: half.x = radius
  T br;        // bounding radius
  int vb, vc;  // mesh vertex range
  int fb, fc;  // mesh plane range
  int pn;      // > 0: regular pn-gon prism along z (vertices vb..vb+pn-1 bottom ring, vb+pn.. top ring)
};

template <class T> GD V3<T> sel(bool k, V3<T> x, V3<T> y) { return {k ? x.x : y.x, k ? x.y : y.y, k ? x.z : y.z}; }

// ---- mesh tables ------------------------------------------------------------
// Vertices and face planes are read either from the scene's flat pools (verts: 3 T per vertex, planes:
// 4 T per face; any pointer type) or from a table of 4-wide, 16-byte aligned records (Vec4Tab: the
// exact stage's LDS copy of SceneDev::mesh4), where one access brings a whole vertex or plane.
template <class T> struct alignas(16) Quad { T x, y, z, w; };
template <class T> struct Vec4Tab { const Quad<T>* p; };

template <class T, class VP> GD V3<T> load_vert(VP verts, int i) { return {verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]}; }
template <class T> GD V3<T> load_vert(Vec4Tab<T> t, int i) { const Quad<T> v = t.p[i]; return {v.x, v.y, v.z}; }
template <class T, class PP> GD Quad<T> load_plane(PP planes, int i) {
  return {planes[4 * i], planes[4 * i + 1], planes[4 * i + 2], planes[4 * i + 3]};
}
template <class T> GD Quad<T> load_plane(Vec4Tab<T> t, int i) { return t.p[i]; }

// support point of the CORE shape (ball -> its centre) in shape-local direction d
template <class T, class VP> GD V3<T> support_core(const ShapeRef<T>& s, VP verts, V3<T> d) {
  if (s.kind == SK_BOX) return {t_copysign(s.half.x, d.x), t_copysign(s.half.y, d.y), t_copysign(s.half.z, d.z)};
  if (s.kind == SK_BALL) return {T(0), T(0), T(0)};
  if (s.pn > 0) {
    // regular prism: the ring vertex nearest the direction's azimuth (its two neighbours are
    // compared too, which covers the rounding of the angle and of the table's accumulated angles);
    // the ring is chosen by the sign of d.z
    const int n = s.pn;
    int k = (int)t_rint(t_atan2(d.y, d.x) * (T(n) * T(0.15915494309189533577)));
    k = k < 0 ? k + n : k;
    k = k >= n ? k - n : k;
    const int km = k == 0 ? n - 1 : k - 1, kp = k == n - 1 ? 0 : k + 1;
    const int ring = s.vb + (d.z < T(0) ? 0 : n);
    const V3<T> p0 = load_vert<T>(verts, ring + km), p1 = load_vert<T>(verts, ring + k), p2 = load_vert<T>(verts, ring + kp);
    const T d0 = dot(p0, d), d1 = dot(p1, d), d2 = dot(p2, d);
    V3<T> best = sel(d0 > d1, p0, p1);
    return sel(d2 > t_max(d0, d1), p2, best);
  }
  V3<T> best = load_vert<T>(verts, s.vb);
  T bd = dot(best, d);
  for (int j = 1; j < s.vc; j++) {
    const V3<T> p = load_vert<T>(verts, s.vb + j);
    T pd = dot(p, d);
    if (pd > bd) { bd = pd; best = p; }
  }
  return best;
}

// ---- closest point of a simplex to the origin ------------------------------
// Keeps only the vertices that support the closest feature.  Returns false when
// the origin is inside a full tetrahedron (shapes intersect).
// The four points live in named registers (runtime-indexed arrays would be
// demoted to scratch memory); p0..p(n-1) are valid.
template <class T> struct Simplex { V3<T> a, b, c, d; int n; };

template <class T> GD void simplex_push(Simplex<T>& s, V3<T> w) {
  // value selects, not "if (n == 1) s.b = w; else ...": the compiler turns that chain into a store
  // through a selected ADDRESS, which pins b, c, d in scratch memory (two memory round trips in every
  // GJK iteration)
  const int n = s.n;
  s.b = sel(n == 1, w, s.b);
  s.c = sel(n == 2, w, s.c);
  s.d = sel(n >= 3, w, s.d);
  s.n = n + 1;
}
// rebuild from the kept subset (3-bit mask) of (a, b, c)
template <class T> GD void simplex_set(Simplex<T>& s, V3<T> a, V3<T> b, V3<T> c, int keep) {
  bool ka = keep & 1, kb = keep & 2, kc = keep & 4;
  s.a = sel(ka, a, sel(kb, b, c));
  s.b = sel(ka, sel(kb, b, c), c);
  s.c = c;
  s.n = (int)ka + (int)kb + (int)kc;
}

template <class T> GD void closest_segment(V3<T> a, V3<T> b, V3<T>& v, int& keep) {
  V3<T> ab = b - a;
  T t = -dot(a, ab), den = dot(ab, ab);
  if (t <= T(0)) { v = a; keep = 1; }
  else if (t >= den) { v = b; keep = 2; }
  else { v = a + ab * (t / den); keep = 3; }
}

template <class T> GD void closest_triangle(V3<T> a, V3<T> b, V3<T> c, V3<T>& v, int& keep) {
  // Ericson, "Real-Time Collision Detection" 5.1.5, query point = origin
  V3<T> ab = b - a, ac = c - a;
  T d1 = -dot(ab, a), d2 = -dot(ac, a);
  if (d1 <= T(0) && d2 <= T(0)) { v = a; keep = 1; return; }
  T d3 = -dot(ab, b), d4 = -dot(ac, b);
  if (d3 >= T(0) && d4 <= d3) { v = b; keep = 2; return; }
  T vc = d1 * d4 - d3 * d2;
  if (vc <= T(0) && d1 >= T(0) && d3 <= T(0)) { v = a + ab * (d1 / (d1 - d3)); keep = 3; return; }
  T d5 = -dot(ab, c), d6 = -dot(ac, c);
  if (d6 >= T(0) && d5 <= d6) { v = c; keep = 4; return; }
  T vb = d5 * d2 - d1 * d6;
  if (vb <= T(0) && d2 >= T(0) && d6 <= T(0)) { v = a + ac * (d2 / (d2 - d6)); keep = 5; return; }
  T va = d3 * d6 - d5 * d4;
  if (va <= T(0) && (d4 - d3) >= T(0) && (d5 - d6) >= T(0)) {
    v = b + (c - b) * ((d4 - d3) / ((d4 - d3) + (d5 - d6)));
    keep = 6;
    return;
  }
  // interior: project the origin on the plane (more robust than barycentric sums)
  V3<T> n = cross(ab, ac);
  T nn = dot(n, n);
  if (nn > T(0)) v = n * (dot(a, n) / nn);
  else { T den = T(1) / (va + vb + vc); v = a + ab * (vb * den) + ac * (vc * den); }
  keep = 7;
}

// one face (a,b,c) of a tetrahedron whose fourth vertex is d: if the origin is on
// the outer side, offer the face's closest point as a candidate
template <class T>
GD void tetra_face(V3<T> a, V3<T> b, V3<T> c, V3<T> d, bool& any, T& best, V3<T>& bv, V3<T>& ta, V3<T>& tb,
                   V3<T>& tc, int& bkeep) {
  V3<T> n = cross(b - a, c - a);
  T sd = dot(n, d - a), so = -dot(n, a);
  bool outside = (sd == T(0)) || (so * sd < T(0));
  if (outside) {
    V3<T> cv;
    int keep;
    closest_triangle(a, b, c, cv, keep);
    T dd = dot(cv, cv);
    if (!any || dd < best) { best = dd; bv = cv; ta = a; tb = b; tc = c; bkeep = keep; any = true; }
  }
}

template <class T> GD bool simplex_closest(Simplex<T>& s, V3<T>& v) {
  if (s.n == 1) { v = s.a; return true; }
  if (s.n == 2) {
    int keep;
    closest_segment(s.a, s.b, v, keep);
    simplex_set(s, s.a, s.b, s.b, keep);
    return true;
  }
  if (s.n == 3) {
    int keep;
    closest_triangle(s.a, s.b, s.c, v, keep);
    simplex_set(s, s.a, s.b, s.c, keep);
    return true;
  }
  // tetrahedron: the origin is inside iff it is on the inner side of all four faces
  bool any = false;
  T best = T(0);
  V3<T> bv{0, 0, 0}, ta{}, tb{}, tc{};
  int bkeep = 0;
  tetra_face(s.a, s.b, s.c, s.d, any, best, bv, ta, tb, tc, bkeep);
  tetra_face(s.a, s.c, s.d, s.b, any, best, bv, ta, tb, tc, bkeep);
  tetra_face(s.a, s.d, s.b, s.c, any, best, bv, ta, tb, tc, bkeep);
  tetra_face(s.b, s.d, s.c, s.a, any, best, bv, ta, tb, tc, bkeep);
  if (!any) return false;  // origin enclosed
  v = bv;
  simplex_set(s, ta, tb, tc, bkeep);
  return true;
}

// GEAR_GJK_MAXIT: test builds of the host harness force the iteration cap with a tiny value
// (tests/test_narrowphase_host.py); the product never defines it.
#ifndef GEAR_GJK_MAXIT
#define GEAR_GJK_MAXIT_F32 40
#define GEAR_GJK_MAXIT_F64 64
#else
#define GEAR_GJK_MAXIT_F32 GEAR_GJK_MAXIT
#define GEAR_GJK_MAXIT_F64 GEAR_GJK_MAXIT
#endif
template <class T> struct Tol;
template <> struct Tol<float> { static constexpr float rel = 2e-6f; static constexpr int maxit = GEAR_GJK_MAXIT_F32; };
template <> struct Tol<double> { static constexpr double rel = 1e-13; static constexpr int maxit = GEAR_GJK_MAXIT_F64; };

// GJK margin query between two convex cores in A's frame: true iff
// dist(coreA, coreB) < m (m > 0).  CSO = A - B.
template <class T, class VP>
GD bool gjk_within(const ShapeRef<T>& A, const ShapeRef<T>& B, const Rel<T>& rel, VP verts, T m, int& iters) {
  const T m2 = m * m;
  auto cso = [&](V3<T> d) -> V3<T> {  // support of A - B' in direction d (A frame)
    V3<T> pa = support_core(A, verts, d);
    V3<T> pb = rel_mul(rel, support_core(B, verts, rel_tmul(rel, -d))) + rel.t;
    return pa - pb;
  };
  V3<T> d0 = -rel.t;  // from B's centre towards A's
  if (dot(d0, d0) == T(0)) d0 = {T(1), T(0), T(0)};
  Simplex<T> s;
  s.a = cso(-d0);  // the CSO point nearest the origin along the centre line
  s.b = s.c = s.d = s.a;
  s.n = 1;
  V3<T> v = s.a;
  int it = 0;
  for (; it < Tol<T>::maxit; it++) {
    T vv = dot(v, v);
    if (vv < m2) { iters = it; return true; }  // upper bound below the margin
    V3<T> w = cso(-v);
    T vw = dot(v, w);
    // every CSO point x has v.x >= v.w: the distance is at least v.w / |v|
    if (vw > T(0) && vw * vw >= m2 * vv) { iters = it; return false; }
    if (vv - vw <= Tol<T>::rel * vv) { iters = it; return false; }  // converged, |v| >= m
    simplex_push(s, w);
    if (!simplex_closest(s, v)) { iters = it; return true; }  // origin enclosed: intersecting
    // No progress although neither bound decided: rounding stalled the descent with the margin still
    // inside [v.w / |v|, |v|].  Undecided resolves to "within", the safe side for a collision checker;
    // a stall happens a few ulps from convergence, i.e. deep inside the parity band.
    if (dot(v, v) >= vv) { iters = it; return true; }
  }
  // Iteration cap: ncollide's distance query turns GJK non-convergence into distance 0, which the
  // reference's `dist < prediction` reads as colliding (oracle/gear_oracle.hpp gjk_distance).
  iters = it;
  return true;
}

// ---- 15-axis separating-axis test with margin for two boxes ---------------
// 0: separated by >= m on some axis (free); 1: overlapping (distance 0);
// 2: 0 <= separation < m on the best axis — needs the exact query.
template <class T> GD int sat_box_box(V3<T> a, V3<T> b, const Rel<T>& rel, T m) {
  const T eps = sizeof(T) == 4 ? T(1e-6) : T(1e-12);
  T ar[9];
#pragma unroll
  for (int i = 0; i < 9; i++) ar[i] = t_abs(rel.r[i]) + eps;
  const T* R = rel.r;
  T tx = rel.t.x, ty = rel.t.y, tz = rel.t.z;
  bool apart = false;   // some axis shows separation >= m
  bool gap = false;     // some axis shows separation >= 0
  T s;
  // A's face normals
  s = t_abs(tx) - (a.x + b.x * ar[0] + b.y * ar[1] + b.z * ar[2]); apart |= s >= m; gap |= s >= T(0);
  s = t_abs(ty) - (a.y + b.x * ar[3] + b.y * ar[4] + b.z * ar[5]); apart |= s >= m; gap |= s >= T(0);
  s = t_abs(tz) - (a.z + b.x * ar[6] + b.y * ar[7] + b.z * ar[8]); apart |= s >= m; gap |= s >= T(0);
  // B's face normals
  s = t_abs(tx * R[0] + ty * R[3] + tz * R[6]) - (b.x + a.x * ar[0] + a.y * ar[3] + a.z * ar[6]); apart |= s >= m; gap |= s >= T(0);
  s = t_abs(tx * R[1] + ty * R[4] + tz * R[7]) - (b.y + a.x * ar[1] + a.y * ar[4] + a.z * ar[7]); apart |= s >= m; gap |= s >= T(0);
  s = t_abs(tx * R[2] + ty * R[5] + tz * R[8]) - (b.z + a.x * ar[2] + a.y * ar[5] + a.z * ar[8]); apart |= s >= m; gap |= s >= T(0);
  // edge x edge axes A_i x B_j (length sqrt(1 - R_ij^2) <= 1): separated by >= m
  // iff s >= m * len, tested as s >= 0 && s^2 >= m^2 (1 - R_ij^2)
  const T m2 = m * m;
#define GEAR_EDGE_AXIS(EXPR, RA, RB, RIJ)                                  \
  s = t_abs(EXPR) - ((RA) + (RB));                                         \
  gap |= s >= T(0);                                                        \
  apart |= (s >= T(0)) && (s * s >= m2 * (T(1) - (RIJ) * (RIJ)));
  GEAR_EDGE_AXIS(tz * R[3] - ty * R[6], a.y * ar[6] + a.z * ar[3], b.y * ar[2] + b.z * ar[1], R[0])
  GEAR_EDGE_AXIS(tz * R[4] - ty * R[7], a.y * ar[7] + a.z * ar[4], b.x * ar[2] + b.z * ar[0], R[1])
  GEAR_EDGE_AXIS(tz * R[5] - ty * R[8], a.y * ar[8] + a.z * ar[5], b.x * ar[1] + b.y * ar[0], R[2])
  GEAR_EDGE_AXIS(tx * R[6] - tz * R[0], a.x * ar[6] + a.z * ar[0], b.y * ar[5] + b.z * ar[4], R[3])
  GEAR_EDGE_AXIS(tx * R[7] - tz * R[1], a.x * ar[7] + a.z * ar[1], b.x * ar[5] + b.z * ar[3], R[4])
  GEAR_EDGE_AXIS(tx * R[8] - tz * R[2], a.x * ar[8] + a.z * ar[2], b.x * ar[4] + b.y * ar[3], R[5])
  GEAR_EDGE_AXIS(ty * R[0] - tx * R[3], a.x * ar[3] + a.y * ar[0], b.y * ar[8] + b.z * ar[7], R[6])
  GEAR_EDGE_AXIS(ty * R[1] - tx * R[4], a.x * ar[4] + a.y * ar[1], b.x * ar[8] + b.z * ar[6], R[7])
  GEAR_EDGE_AXIS(ty * R[2] - tx * R[5], a.x * ar[5] + a.y * ar[2], b.x * ar[7] + b.y * ar[6], R[8])
#undef GEAR_EDGE_AXIS
  if (apart) return 0;
  return gap ? 2 : 1;
}

// X strictly inside the closed convex mesh P with clearance >= m on every face.
// A TriMesh is a surface (src/collision/urdf.rs:26-50 builds one for cylinders):
// a shape that sits inside it without touching a triangle is at a positive
// distance from it, so the reference reports it free (DESIGN.md "Hollow meshes").
// p_is_a: P is shape A (frame of `rel`), X is B; else P is B and X is A.
template <class T, class VP, class PP>
GD bool inside_with_clearance(const ShapeRef<T>& P, const ShapeRef<T>& X, const Rel<T>& rel, bool p_is_a,
                              VP verts, PP planes, T m) {
  if (P.fc == 0) return false;           // a lone triangle of a general mesh encloses nothing
  if (X.br > T(2) * P.br) return false;  // enclosing-ball argument: X inside P needs br(X) <= 2 br(P)
  T xr = X.kind == SK_BALL ? X.half.x : T(0);
  // cheap necessary condition first, for the kinds whose frame origin is a point of their core (box,
  // ball, prism; a general mesh's bounding-box centre need not be): that point must clear every face
  // by m + xr.  Most intersecting pairs fail it, and it needs no support evaluation.
  if (X.kind != SK_MESH || X.pn > 0) {
    const V3<T> cx = p_is_a ? rel.t : -rel_tmul(rel, rel.t);  // X's origin in P's frame
    // four faces at a time: their plane reads are in flight together (a face index past the end
    // repeats the last face)
    for (int f = 0; f < P.fc; f += 4) {
      bool clear = true;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const Quad<T> pl = load_plane<T>(planes, P.fb + (f + j < P.fc ? f + j : P.fc - 1));
        clear = clear && (pl.w - dot(V3<T>{pl.x, pl.y, pl.z}, cx) - xr >= m);
      }
      if (!clear) return false;
    }
    if (X.kind == SK_BALL) return true;  // a ball's core is its centre: the loop above was the whole test
  }
  for (int f = 0; f < P.fc; f++) {
    const Quad<T> pl = load_plane<T>(planes, P.fb + f);
    const V3<T> n{pl.x, pl.y, pl.z};
    T o = pl.w;
    T h;
    if (p_is_a) {
      V3<T> px = rel_mul(rel, support_core(X, verts, rel_tmul(rel, n))) + rel.t;
      h = dot(n, px);
    } else {
      V3<T> da = rel_mul(rel, n);
      V3<T> pa = support_core(X, verts, da);
      h = dot(da, pa - rel.t);
    }
    if (!(o - h - xr >= m)) return false;
  }
  return true;
}

struct LaneCounters { unsigned cull_pass, sat_tests, exact_tests, exact_iters; };

enum : int { PR_FREE = 0, PR_HIT = 1, PR_EXACT = 2 };

// Stage 1 of "distance(poseA, A, poseB, B) < margin" for a pair that survived the
// cull: closed forms (ball pairs) and the separating-axis test (box pairs).
// PR_EXACT = undecided here (a mesh is involved, or the boxes' best axis shows a
// gap inside [0, margin)): stage 2 decides.
template <class T, bool COUNT>
GD int pair_stage1(const ShapeRef<T>& A, const Pose<T>& wa, const ShapeRef<T>& B, const Pose<T>& wb, T margin,
                   LaneCounters& lc) {
  const bool mesh = A.kind == SK_MESH || B.kind == SK_MESH;
  if (mesh && !(T(0) < margin)) return PR_FREE;
  if (A.kind == SK_BALL && B.kind == SK_BALL) {
    // distance_ball_ball, verbatim semantics
    V3<T> d = wa.t - wb.t;
    T d2 = dot(d, d), sum = A.half.x + B.half.x;
    T dist = d2 <= sum * sum ? T(0) : t_sqrt(d2) - sum;
    return dist < margin ? PR_HIT : PR_FREE;
  }
  Rel<T> rel = make_rel(wa, wb);
  if (!mesh && (A.kind == SK_BALL || B.kind == SK_BALL)) {
    // ball vs box: point-to-box distance in the box's frame
    V3<T> p, h;
    T r;
    if (A.kind == SK_BOX) { p = rel.t; h = A.half; r = B.half.x; }
    else { p = -rel_tmul(rel, rel.t); h = B.half; r = A.half.x; }
    V3<T> e{t_max(t_abs(p.x) - h.x, T(0)), t_max(t_abs(p.y) - h.y, T(0)), t_max(t_abs(p.z) - h.z, T(0))};
    T dist = t_max(t_sqrt(dot(e, e)) - r, T(0));
    return dist < margin ? PR_HIT : PR_FREE;
  }
  // One separating-axis evaluation serves both remaining cases (a single site: lanes of a wavefront
  // hold different pair kinds).  box vs box: decisive unless the best axis shows a gap in [0, margin).
  // A mesh carries the half extents of its local bounding box in `half` (a ball: the cube around it):
  // boxes that are >= margin apart prove the shapes are; anything closer goes to the exact stage.
  V3<T> ha = A.kind == SK_BALL ? V3<T>{A.half.x, A.half.x, A.half.x} : A.half;
  V3<T> hb = B.kind == SK_BALL ? V3<T>{B.half.x, B.half.x, B.half.x} : B.half;
  if (COUNT) lc.sat_tests++;
  const int r = sat_box_box(ha, hb, rel, margin);
  if (mesh) return r == 0 ? PR_FREE : PR_EXACT;
  if (r == 2) return PR_EXACT;
  return (r == 1 && T(0) < margin) ? PR_HIT : PR_FREE;
}

// Stage 2: GJK margin query on the convex cores (ball radii folded into the
// margin) plus the hollow-mesh rule.
template <class T, bool COUNT, class VP, class PP>
GD bool pair_stage2(const ShapeRef<T>& A, const Pose<T>& wa, const ShapeRef<T>& B, const Pose<T>& wb, VP verts,
                    PP planes, T margin, LaneCounters& lc) {
  if (!(T(0) < margin)) return false;
  Rel<T> rel = make_rel(wa, wb);
  T m = margin + (A.kind == SK_BALL ? A.half.x : T(0)) + (B.kind == SK_BALL ? B.half.x : T(0));
  if (COUNT) lc.exact_tests++;
  int it;
  bool hit = gjk_within(A, B, rel, verts, m, it);
  if (COUNT) lc.exact_iters += it + 1;
  if (!hit) return false;
  if (A.kind == SK_MESH && inside_with_clearance(A, B, rel, true, verts, planes, margin)) return false;
  if (B.kind == SK_MESH && inside_with_clearance(B, A, rel, false, verts, planes, margin)) return false;
  return true;
}

// both stages back to back (diagnostic kernels, host harness)
template <class T, bool COUNT, class VP, class PP>
GD bool pair_within(const ShapeRef<T>& A, const Pose<T>& wa, const ShapeRef<T>& B, const Pose<T>& wb, VP verts,
                    PP planes, T margin, LaneCounters& lc) {
  int r = pair_stage1<T, COUNT>(A, wa, B, wb, margin, lc);
  if (r != PR_EXACT) return r == PR_HIT;
  return pair_stage2<T, COUNT>(A, wa, B, wb, verts, planes, margin, lc);
}

}  // namespace gear_hip
