// gear_scene.hpp — host-side flattening of gh_robot_desc / gh_obstacle_desc into
// the device tables of gear_types.hpp.  Pure host C++ (no HIP calls): the C ABI
// layer uploads the resulting vectors.
//
// Mirrors, for the hot path only:
//   k::Chain::from(&urdf_robot) wiring + mimic      (k 0.20, restated)
//   CollisionChecker::from_urdf_robot_with_base_dir  src/collision/collision_checker.rs:234-262
//   urdf_geometry_to_shape_handle                    src/collision/urdf.rs:10-73
//   Compound<f64>::from_urdf_robot                   src/collision/collision_checker.rs:349-369
#pragma once
#include <cstdlib>
#include <algorithm>
#include <array>
#include <cmath>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/gear_hip.h"
#include "gear_types.hpp"

namespace gear_hip {

template <class T>
struct HostTables {
  std::vector<NodeDev<T>> nodes;
  std::vector<LimitDev<T>> limits;
  std::vector<ShapeDev<T>> rshapes, oshapes;
  std::vector<int32_t> obs_owner;  // [oshapes] the caller's obstacle index of every device obstacle shape (a mesh may be many)
  int n_obs_user = 0;
  std::vector<T> ocull, ocullq;
  std::vector<PairDev> pairs;
  std::vector<int32_t> spairs;  // flattened self shape pairs: [n][4] = s1, s2, node-pair index, LDS slot of the earlier shape
  std::vector<T> spair_r;       // [n] bounding radii sum + margin, inflated
  std::vector<T> verts, planes;
  std::vector<GridDev<T>> grids;
  std::vector<int32_t> shape_grid;
  std::vector<uint32_t> cell_blk;  // [n_blocks][4], see SceneDev::cell_blk
  std::vector<int32_t> sp_first_slot, sp_late_off, sp_late;  // see SceneDev
  std::vector<SpLate<T>> sp_late_rec;
  std::vector<T> hot;                                        // see SceneDev::hot
  int n_first = 0;
  int n_slots = 0;
  int all_blk = -1;  // head block of the chain listing every obstacle, -1: no obstacles
  T margin = 0;
};

struct BuildError {
  int code;
  std::string msg;
};

namespace detail {

// urdf.rs:26-50: Cylinder(len/2, r).to_trimesh(30), xzy swizzle -> 30-gon prism
// along Z; angles accumulate by repeated addition in T, as ncollide's
// procedural push_circle does.
template <class T>
inline void make_prism(double radius, double length, std::vector<double>& verts, std::vector<int>& tris) {
  const int n = 30;
  T r = T(radius), hh = T(length * 0.5);
  T dia = r * T(2), hgt = hh * T(2);
  T dth = T(6.283185307179586476925286766559) * (T(1) / T(n));
  for (int cap = 0; cap < 2; cap++) {
    T zc = cap == 0 ? T(-0.5) : T(0.5);
    T th = 0;
    for (int i = 0; i < n; i++) {
      T x = std::cos(th) * T(0.5), y = std::sin(th) * T(0.5);
      verts.push_back((double)(x * dia));
      verts.push_back((double)(y * dia));
      verts.push_back((double)(zc * hgt));
      th = th + dth;
    }
  }
  for (int i = 0; i < n; i++) {
    int j = (i + 1) % n;
    tris.insert(tris.end(), {i, j, n + i, j, n + j, n + i});
  }
  for (int i = 1; i + 1 < n; i++) {
    tris.insert(tris.end(), {0, i + 1, i, n, n + i, n + i + 1});
  }
}

// Face planes (outward unit normal, offset) of a closed convex triangle mesh;
// coplanar triangles collapse into one plane.  Returns false (with a reason)
// when the mesh is not closed or not convex.
inline bool convex_planes(const std::vector<double>& v, const std::vector<int>& tris, std::vector<double>& planes,
                          std::string& why) {
  size_t nv = v.size() / 3, nt = tris.size() / 3;
  if (nv < 4 || nt < 4) { why = "mesh needs at least 4 vertices and 4 triangles"; return false; }
  double c[3] = {0, 0, 0}, lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (size_t i = 0; i < nv; i++)
    for (int k = 0; k < 3; k++) {
      c[k] += v[3 * i + k] / nv;
      lo[k] = std::min(lo[k], v[3 * i + k]);
      hi[k] = std::max(hi[k], v[3 * i + k]);
    }
  double scale = std::max({hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]});
  if (!(scale > 0)) { why = "degenerate mesh"; return false; }
  // closed: every undirected edge is shared by exactly two triangles
  std::map<std::pair<int, int>, int> edges;
  for (size_t t = 0; t < nt; t++)
    for (int e = 0; e < 3; e++) {
      int a = tris[3 * t + e], b = tris[3 * t + (e + 1) % 3];
      if (a < 0 || b < 0 || a >= (int)nv || b >= (int)nv || a == b) { why = "bad triangle index"; return false; }
      edges[{std::min(a, b), std::max(a, b)}]++;
    }
  for (auto& kv : edges)
    if (kv.second != 2) { why = "mesh is not closed (an edge is not shared by exactly two triangles)"; return false; }
  const double tol = 1e-6 * scale;
  planes.clear();
  for (size_t t = 0; t < nt; t++) {
    const double* a = &v[3 * tris[3 * t]];
    const double* b = &v[3 * tris[3 * t + 1]];
    const double* d = &v[3 * tris[3 * t + 2]];
    double e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {d[0] - a[0], d[1] - a[1], d[2] - a[2]};
    double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    double nn = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    if (nn < 1e-12 * scale * scale) continue;  // sliver: carries no plane
    for (int k = 0; k < 3; k++) n[k] /= nn;
    double o = n[0] * a[0] + n[1] * a[1] + n[2] * a[2];
    if (n[0] * c[0] + n[1] * c[1] + n[2] * c[2] > o) { for (int k = 0; k < 3; k++) n[k] = -n[k]; o = -o; }
    for (size_t i = 0; i < nv; i++)
      if (n[0] * v[3 * i] + n[1] * v[3 * i + 1] + n[2] * v[3 * i + 2] > o + tol) { why = "mesh is not convex"; return false; }
    bool dup = false;
    for (size_t p = 0; p < planes.size() / 4 && !dup; p++)
      dup = std::fabs(planes[4 * p] - n[0]) < 1e-5 && std::fabs(planes[4 * p + 1] - n[1]) < 1e-5 &&
            std::fabs(planes[4 * p + 2] - n[2]) < 1e-5 && std::fabs(planes[4 * p + 3] - o) < tol;
    if (!dup) planes.insert(planes.end(), {n[0], n[1], n[2], o});
  }
  if (planes.size() < 16) { why = "mesh has fewer than 4 distinct faces"; return false; }
  return true;
}

// One descriptor -> one device shape, or — a mesh that is not a closed convex polytope — one device shape PER
// TRIANGLE.  The reference turns any mesh file into an ncollide TriMesh, a triangle soup whose distance to
// another shape is the minimum over its triangles (src/collision/urdf.rs:54-71, mesh.rs:39-69), and the path
// only asks "is that minimum below the margin", which is an OR over the triangles.  So a general mesh joins the
// scene as its triangles: each is a convex piece with three vertices and no face planes (nothing can be
// "inside" it), each has its own bounding sphere, and the broadphase grid does for them what a BVH would.
// A closed convex mesh stays ONE shape (hull + the hollow-surface rule): the same answers from one GJK query
// instead of one per triangle.  `piece_of` is the caller's index of the descriptor, repeated per piece.
template <class T>
inline bool flatten_shape(const gh_shape_desc& d, const gh_mesh_desc* meshes, int n_meshes, int node,
                          HostTables<T>& H, std::vector<ShapeDev<T>>& out, BuildError& err) {
  ShapeDev<T> s{};
  for (int k = 0; k < 3; k++) s.lt[k] = T(d.t[k]);
  for (int k = 0; k < 4; k++) s.lq[k] = T(d.q[k]);
  s.node = node;
  s.v_begin = s.v_count = s.f_begin = s.f_count = 0;
  s.prism_n = 0;
  auto bad = [&](const char* m) { err = {GH_ERR_INVALID, m}; return false; };
  // a convex piece given by its vertices (and, for a closed polytope, its face planes): re-centred on the centre
  // of its bounding box, the shift folded into the shape's pose (t += R(q) c), so that for every shape kind the
  // bounding sphere sits at the pose translation
  auto emit_polytope = [&](const std::vector<double>& v, const std::vector<double>& planes, int prism_n) {
    ShapeDev<T> m = s;
    m.kind = SK_MESH;
    m.prism_n = prism_n;
    m.v_begin = (int)(H.verts.size() / 3);
    m.v_count = (int)(v.size() / 3);
    m.f_begin = (int)(H.planes.size() / 4);
    m.f_count = (int)(planes.size() / 4);
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (size_t i = 0; i < v.size() / 3; i++)
      for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], v[3 * i + k]); hi[k] = std::max(hi[k], v[3 * i + k]); }
    double c[3] = {(lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, (lo[2] + hi[2]) / 2}, r2 = 0;
    for (size_t i = 0; i < v.size() / 3; i++) {
      double dd = 0;
      for (int k = 0; k < 3; k++) dd += (v[3 * i + k] - c[k]) * (v[3 * i + k] - c[k]);
      r2 = std::max(r2, dd);
    }
    {
      double q[4] = {d.q[0], d.q[1], d.q[2], d.q[3]};
      double tx = 2 * (q[1] * c[2] - q[2] * c[1]), ty = 2 * (q[2] * c[0] - q[0] * c[2]), tz = 2 * (q[0] * c[1] - q[1] * c[0]);
      double rc[3] = {c[0] + q[3] * tx + (q[1] * tz - q[2] * ty), c[1] + q[3] * ty + (q[2] * tx - q[0] * tz),
                      c[2] + q[3] * tz + (q[0] * ty - q[1] * tx)};
      for (int k = 0; k < 3; k++) m.lt[k] = T(d.t[k] + rc[k]);
    }
    for (int k = 0; k < 3; k++) m.bc[k] = T(0);
    // local bounding box (centred after the shift), padded a little: stage 1 uses it as a prefilter
    for (int k = 0; k < 3; k++) m.half[k] = T((hi[k] - lo[k]) * 0.5 * (1.0 + 1e-6) + 1e-9);
    m.br = T(std::sqrt(r2) * (1.0 + 1e-6) + 1e-12);
    for (size_t i = 0; i < v.size(); i++) H.verts.push_back(T(v[i] - c[i % 3]));
    for (size_t pl = 0; pl < planes.size() / 4; pl++) {
      double off = planes[4 * pl + 3] - (planes[4 * pl] * c[0] + planes[4 * pl + 1] * c[1] + planes[4 * pl + 2] * c[2]);
      H.planes.push_back(T(planes[4 * pl]));
      H.planes.push_back(T(planes[4 * pl + 1]));
      H.planes.push_back(T(planes[4 * pl + 2]));
      H.planes.push_back(T(off));
    }
    out.push_back(m);
  };
  switch (d.kind) {
    case GH_SHAPE_BOX: {
      s.kind = SK_BOX;
      double r2 = 0;
      for (int k = 0; k < 3; k++) {
        if (!(d.params[k] > 0)) return bad("box half extents must be positive");
        s.half[k] = T(d.params[k]);
        r2 += (double)s.half[k] * (double)s.half[k];
      }
      s.br = T(std::sqrt(r2) * (1.0 + 1e-6));
      out.push_back(s);
      break;
    }
    case GH_SHAPE_BALL:
      if (!(d.params[0] > 0)) return bad("ball radius must be positive");
      s.kind = SK_BALL;
      s.half[0] = T(d.params[0]);
      s.br = s.half[0];
      out.push_back(s);
      break;
    case GH_SHAPE_CYLINDER:
    case GH_SHAPE_MESH: {
      std::vector<double> v;
      std::vector<int> tris;
      if (d.kind == GH_SHAPE_CYLINDER) {
        if (!(d.params[0] > 0) || !(d.params[1] > 0)) return bad("cylinder radius and length must be positive");
        make_prism<T>(d.params[0], d.params[1], v, tris);
      } else {
        if (d.mesh < 0 || d.mesh >= n_meshes || !meshes) return bad("mesh index out of range");
        const gh_mesh_desc& m = meshes[d.mesh];
        if (m.n_vertices <= 0 || m.n_triangles <= 0 || !m.vertices || !m.triangles) return bad("empty mesh");
        for (int i = 0; i < 3 * m.n_vertices; i++) v.push_back((double)T(m.vertices[i]));
        tris.assign(m.triangles, m.triangles + 3 * m.n_triangles);
        for (int t : tris)
          if (t < 0 || t >= m.n_vertices) return bad("mesh triangle index out of range");
      }
      std::vector<double> planes;
      std::string why;
      if (convex_planes(v, tris, planes, why) && v.size() / 3 <= 64) {
        emit_polytope(v, planes, d.kind == GH_SHAPE_CYLINDER ? 30 : 0);
        break;
      }
      if (d.kind == GH_SHAPE_CYLINDER) { err = {GH_ERR_INVALID, "internal: the cylinder's prism is not convex: " + why}; return false; }
      // a triangle soup: one piece per triangle
      for (size_t t = 0; t < tris.size() / 3; t++) {
        std::vector<double> tv;
        for (int e = 0; e < 3; e++)
          for (int k = 0; k < 3; k++) tv.push_back(v[3 * (size_t)tris[3 * t + e] + k]);
        emit_polytope(tv, {}, 0);
      }
      break;
    }
    default:
      return bad("unknown shape kind");
  }
  return true;
}

// Broadphase grids, one per class of robot-shape bounding radius (at most 4 classes, split at the
// radius quantiles; a class uses its largest radius).  Conservative by construction: obstacle o is
// listed in every cell whose box is within R_o + margin + rho_class + slack of its centre, all in f64;
// the slack covers the f32 rounding of the cell lookup on the device.
template <class T>
inline void build_grids(HostTables<T>& H, double margin) {
  const size_t O = H.oshapes.size(), L = H.rshapes.size();
  H.shape_grid.assign(L, 0);
  H.all_blk = -1;
  if (O == 0 || L == 0) return;
  {
    const uint32_t sentinel = (uint32_t)O, empty = sentinel | (sentinel << 16);
    H.cell_blk.assign(4, empty);  // block 0: the empty list
    // the chain that lists EVERY obstacle (GH_PATH_NO_GRID: the brute-force cross-check of the grids)
    H.all_blk = 1;
    size_t pos = 0;
    for (;;) {
      const size_t cur = H.cell_blk.size() / 4;
      H.cell_blk.resize((cur + 1) * 4, empty);
      const size_t left = O - pos, take = left <= 8 ? left : 6;
      for (size_t j = 0; j < take; j++) {
        uint32_t& w = H.cell_blk[cur * 4 + j / 2];
        w = (j & 1) ? ((w & 0xffffu) | ((uint32_t)(pos + j) << 16)) : ((w & 0xffff0000u) | (uint32_t)(pos + j));
      }
      pos += take;
      if (pos == O) break;
      H.cell_blk[cur * 4 + 3] = 0x80000000u | (uint32_t)(cur + 1);
    }
  }
  std::vector<double> radii;
  for (auto& s : H.rshapes) radii.push_back((double)s.br);
  std::vector<double> sorted = radii;
  std::sort(sorted.begin(), sorted.end());
  sorted.erase(std::unique(sorted.begin(), sorted.end()), sorted.end());
  const char* e_cls = getenv("GEAR_HIP_GRID_CLASSES");  // tuning aids
  const char* e_h = getenv("GEAR_HIP_GRID_H");
  const size_t max_cls = e_cls ? (size_t)std::max(1, atoi(e_cls)) : 4;
  // cell edge = half of max(extent / 48, half the median reach, 0.02): cells a quarter of the median reach list ~2.5
  // candidates per shape instead of 3.7 and make the sweep 2-8 % faster on five scenes (finer: nothing more; the
  // cell blocks then leave the L2)
  const double h_scale = e_h ? std::max(0.05, atof(e_h)) : 0.5;
  const int n_cls = (int)std::min<size_t>(max_cls, sorted.size());
  std::vector<double> cls_r(n_cls);
  for (int c = 0; c < n_cls; c++) cls_r[c] = sorted[(size_t)((c + 1) * sorted.size() / n_cls) - 1];
  for (size_t s = 0; s < L; s++) {
    int c = 0;
    while (radii[s] > cls_r[c]) c++;
    H.shape_grid[s] = c;
  }
  for (int c = 0; c < n_cls; c++) {
    std::vector<double> reach(O);
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (size_t o = 0; o < O; o++) {
      const double R = (double)H.ocull[4 * o + 3];  // bounding radius + margin, already inflated
      reach[o] = (R + cls_r[c]) * (1.0 + 1e-5);
      for (int k = 0; k < 3; k++) {
        lo[k] = std::min(lo[k], (double)H.ocull[4 * o + k] - reach[o]);
        hi[k] = std::max(hi[k], (double)H.ocull[4 * o + k] + reach[o]);
      }
    }
    (void)margin;
    const double ext = std::max({hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]});
    const double slack = 1e-5 + 2e-6 * (ext + std::fabs(lo[0]) + std::fabs(lo[1]) + std::fabs(lo[2]));
    std::vector<double> sr(reach);
    std::sort(sr.begin(), sr.end());
    const double h = std::max({ext / 48.0, 0.5 * sr[O / 2], 0.02}) * h_scale;
    GridDev<T> g;
    int n[3];
    for (int k = 0; k < 3; k++) {
      lo[k] -= slack;
      n[k] = std::max(1, (int)std::ceil((hi[k] + slack - lo[k]) / h));
      g.org[k] = T(lo[k]);
    }
    g.inv_h = T(1.0 / h);
    g.nx = n[0]; g.ny = n[1]; g.nz = n[2];
    g.off_base = (int32_t)(H.cell_blk.size() / 4);
    const size_t ncell = (size_t)n[0] * n[1] * n[2];
    // the device computes floor((p - org) * inv_h) with ITS org and inv_h: use the same rounded values
    const double orgd[3] = {(double)g.org[0], (double)g.org[1], (double)g.org[2]}, hd = 1.0 / (double)g.inv_h;
    std::vector<std::vector<uint16_t>> cells(ncell);
    for (size_t o = 0; o < O; o++) {
      const double cx[3] = {(double)H.ocull[4 * o], (double)H.ocull[4 * o + 1], (double)H.ocull[4 * o + 2]};
      const double r = reach[o] + slack;
      int a[3], b[3];
      for (int k = 0; k < 3; k++) {
        a[k] = std::max(0, (int)std::floor((cx[k] - r - orgd[k]) / hd) - 1);
        b[k] = std::min(n[k] - 1, (int)std::floor((cx[k] + r - orgd[k]) / hd) + 1);
      }
      for (int iz = a[2]; iz <= b[2]; iz++)
        for (int iy = a[1]; iy <= b[1]; iy++)
          for (int ix = a[0]; ix <= b[0]; ix++) {
            const int ic[3] = {ix, iy, iz};
            double d2 = 0;
            for (int k = 0; k < 3; k++) {
              const double bl = orgd[k] + ic[k] * hd, bh = bl + hd;
              const double d = cx[k] < bl ? bl - cx[k] : cx[k] > bh ? cx[k] - bh : 0.0;
              d2 += d * d;
            }
            if (d2 <= r * r) cells[((size_t)iz * n[1] + iy) * n[0] + ix].push_back((uint16_t)o);
          }
    }
    // head blocks first (one per cell, contiguous), overflow blocks of this grid after them
    const uint32_t sentinel = (uint32_t)O, empty = sentinel | (sentinel << 16);
    const size_t base = H.cell_blk.size() / 4;
    H.cell_blk.resize((base + ncell) * 4, empty);
    for (size_t i = 0; i < ncell; i++) {
      const std::vector<uint16_t>& list = cells[i];
      size_t cur = base + i, pos = 0;
      for (;;) {
        const size_t left = list.size() - pos;
        const size_t take = left <= 8 ? left : 6;
        for (size_t j = 0; j < take; j++) {
          uint32_t& w = H.cell_blk[cur * 4 + j / 2];
          w = (j & 1) ? ((w & 0xffffu) | ((uint32_t)list[pos + j] << 16)) : ((w & 0xffff0000u) | list[pos + j]);
        }
        pos += take;
        if (pos == list.size()) break;
        const size_t nb = H.cell_blk.size() / 4;
        H.cell_blk.resize((nb + 1) * 4, empty);
        H.cell_blk[cur * 4 + 3] = 0x80000000u | (uint32_t)nb;
        cur = nb;
      }
    }
    H.grids.push_back(g);
  }
}

}  // namespace detail

template <class T>
inline bool build_tables(const gh_robot_desc& r, const gh_obstacle_desc& o, double margin, HostTables<T>& H,
                         BuildError& err) {
  auto bad = [&](const char* m) { err = {GH_ERR_INVALID, m}; return false; };
  if (r.n_nodes < 1 || !r.nodes) return bad("robot needs at least the root node");
  if (r.n_cols < 0 || (r.n_cols > 0 && !r.col_node)) return bad("col_node missing");
  if (r.n_shapes < 0 || (r.n_shapes > 0 && !r.shapes)) return bad("robot shapes missing");
  if (o.n_shapes < 0 || (o.n_shapes > 0 && !o.shapes)) return bad("obstacle shapes missing");
  if (r.n_pairs < 0 || (r.n_pairs > 0 && !r.pairs)) return bad("pairs missing");
  if (r.n_shapes > 1023) { err = {GH_ERR_TOO_LARGE, "at most 1023 robot collision shapes"}; return false; }
  if (o.n_shapes > 32767) { err = {GH_ERR_TOO_LARGE, "at most 32767 obstacle shapes"}; return false; }
  const int N = r.n_nodes;
  H.margin = T(margin);
  H.nodes.assign(N, NodeDev<T>{});
  std::vector<bool> needs_save(N, false);
  for (int k = 0; k < N; k++) {
    const gh_node_desc& d = r.nodes[k];
    if (k == 0 ? d.parent != -1 : (d.parent < 0 || d.parent >= k)) return bad("nodes must be in depth-first order");
    if (d.joint_type < 0 || d.joint_type > 2) return bad("unknown joint type");
    if (d.mimic_parent >= N || d.mimic_parent == k) return bad("bad mimic parent");
    NodeDev<T>& n = H.nodes[k];
    // k::urdf::axis_from: convert to T, normalise in T
    T ax[3] = {T(d.axis[0]), T(d.axis[1]), T(d.axis[2])};
    T an = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    for (int j = 0; j < 3; j++) n.axis[j] = an > T(0) ? ax[j] / an : ax[j];
    if (d.joint_type != GH_JOINT_FIXED && !(an > T(0))) return bad("movable joint with a zero axis");
    for (int j = 0; j < 3; j++) n.ot[j] = T(d.origin_t[j]);
    for (int j = 0; j < 4; j++) n.oq[j] = T(d.origin_q[j]);
    n.joint_type = d.joint_type;
    n.col = -1;
    n.direct = 1;
    n.mult = T(1);
    n.off = T(0);
    n.const_pos = T(d.position);
    n.parent_slot = k == 0 ? PARENT_IDENTITY : PARENT_PREV;
    n.save_slot = -1;
    if (k > 0 && d.parent != k - 1) needs_save[d.parent] = true;
  }
  H.n_slots = 0;
  for (int k = 0; k < N; k++)
    if (needs_save[k]) H.nodes[k].save_slot = H.n_slots++;
  for (int k = 1; k < N; k++)
    if (r.nodes[k].parent != k - 1) H.nodes[k].parent_slot = H.nodes[r.nodes[k].parent].save_slot;

  // k::Chain::set_joint_positions, column by column
  std::vector<std::vector<int>> mimic_children(N);
  for (int k = 0; k < N; k++)
    if (r.nodes[k].mimic_parent >= 0) mimic_children[r.nodes[k].mimic_parent].push_back(k);
  for (int c = 0; c < r.n_cols; c++) {
    int k = r.col_node[c];
    if (k < 0 || k >= N) return bad("col_node out of range");
    const gh_node_desc& d = r.nodes[k];
    if (d.joint_type == GH_JOINT_FIXED) return bad("a batch column drives a fixed joint (k::Error::SetToFixedError)");
    if (d.mimic_parent >= 0) continue;  // a mimic joint ignores its own input
    H.nodes[k].col = c;
    H.nodes[k].direct = 1;
    if (d.has_limit) H.limits.push_back({T(1), T(0), T(d.limit_lo), T(d.limit_hi), c, 1});
    for (int ch : mimic_children[k]) {
      const gh_node_desc& cd = r.nodes[ch];
      if (cd.joint_type == GH_JOINT_FIXED) continue;
      H.nodes[ch].col = c;
      H.nodes[ch].direct = 0;
      H.nodes[ch].mult = T(cd.mimic_mult);
      H.nodes[ch].off = T(cd.mimic_offset);
      if (cd.has_limit) H.limits.push_back({T(cd.mimic_mult), T(cd.mimic_offset), T(cd.limit_lo), T(cd.limit_hi), c, 0});
    }
  }

  // robot shapes grouped by node, keeping the caller's order inside a node
  std::vector<std::vector<int>> by_node(N);
  for (int s = 0; s < r.n_shapes; s++) {
    int k = r.shapes[s].node;
    if (k < 0 || k >= N) return bad("shape node out of range");
    by_node[k].push_back(s);
  }
  T br_max = 0;
  for (int k = 0; k < N; k++) {
    H.nodes[k].shape_begin = (int)H.rshapes.size();
    for (int s : by_node[k]) {
      std::vector<ShapeDev<T>> pieces;
      if (!detail::flatten_shape<T>(r.shapes[s], r.meshes, r.n_meshes, k, H, pieces, err)) return false;
      for (const ShapeDev<T>& sd : pieces) {
        br_max = std::max(br_max, sd.br);
        H.rshapes.push_back(sd);
      }
    }
    H.nodes[k].shape_end = (int)H.rshapes.size();
  }
  if (H.rshapes.size() > 1023) {
    err = {GH_ERR_TOO_LARGE, "at most 1023 robot collision shapes, counting every triangle of a link mesh that is not a closed convex polytope (" +
                                 std::to_string(H.rshapes.size()) + " here)"};
    return false;
  }
  std::vector<std::array<double, 5>> cull_rec;
  // obstacles: world poses; cull record = world bounding sphere, radius pre-inflated by the margin
  std::vector<ShapeDev<T>> opieces;
  for (int s = 0; s < o.n_shapes; s++) {
    const size_t before = opieces.size();
    if (!detail::flatten_shape<T>(o.shapes[s], o.meshes, o.n_meshes, -1, H, opieces, err)) return false;
    for (size_t k = before; k < opieces.size(); k++) H.obs_owner.push_back(s);
  }
  H.n_obs_user = o.n_shapes;
  if (opieces.size() > 32767) {
    err = {GH_ERR_TOO_LARGE, "at most 32767 obstacle shapes, counting every triangle of a mesh that is not a closed convex polytope (" +
                                 std::to_string(opieces.size()) + " here)"};
    return false;
  }
  for (const ShapeDev<T>& sd : opieces) {
    H.oshapes.push_back(sd);
    // centre = t + R(q) * bc, in double
    double q[4] = {(double)sd.lq[0], (double)sd.lq[1], (double)sd.lq[2], (double)sd.lq[3]};
    double b[3] = {(double)sd.bc[0], (double)sd.bc[1], (double)sd.bc[2]};
    double tx = 2 * (q[1] * b[2] - q[2] * b[1]), ty = 2 * (q[2] * b[0] - q[0] * b[2]), tz = 2 * (q[0] * b[1] - q[1] * b[0]);
    double cx = b[0] + q[3] * tx + (q[1] * tz - q[2] * ty);
    double cy = b[1] + q[3] * ty + (q[2] * tx - q[0] * tz);
    double cz = b[2] + q[3] * tz + (q[0] * ty - q[1] * tx);
    H.ocull.push_back(T((double)sd.lt[0] + cx));
    H.ocull.push_back(T((double)sd.lt[1] + cy));
    H.ocull.push_back(T((double)sd.lt[2] + cz));
    double rr = (double)sd.br + std::max(margin, 0.0);
    H.ocull.push_back(T(rr + 1e-4 * (rr + (double)br_max) + 1e-9));
    // batched-kernel record; tau absorbs the rounding of the expanded form so that the cull stays
    // conservative (a pair within reach is never dropped)
    const double eps = sizeof(T) == 4 ? 1.2e-7 : 2.3e-16;
    double ox = (double)sd.lt[0] + cx, oy = (double)sd.lt[1] + cy, oz = (double)sd.lt[2] + cz;
    double R = rr * (1.0 + 1e-5) + 1e-9, c2 = ox * ox + oy * oy + oz * oz;
    cull_rec.push_back({-2 * ox, -2 * oy, -2 * oz, -2 * R, c2 - R * R - 16 * eps * (c2 + R * R)});
  }
  // far dummies pad the sweep to whole groups of 32: k is huge and positive, the test value never goes negative
  while (cull_rec.size() % 32 != 0) cull_rec.push_back({0, 0, 0, 0, 1e30});
  // quads of 4 obstacles, component-major inside a quad: [quad][5][4]
  H.ocullq.assign(cull_rec.size() * 5, T(0));
  for (size_t i = 0; i < cull_rec.size(); i++)
    for (int comp = 0; comp < 5; comp++) H.ocullq[(i / 4) * 20 + comp * 4 + (i % 4)] = T(cull_rec[i][comp]);
  detail::build_grids<T>(H, std::max(margin, 0.0));
  // the cell lists' sentinel index n_obs: a record no robot shape can reach
  H.ocull.insert(H.ocull.end(), {T(1e18), T(0), T(0), T(0)});
  // self pairs -> shape ranges; a pair whose joint has no model never yields
  // (collision_checker.rs:150-159)
  for (int p = 0; p < r.n_pairs; p++) {
    int a = r.pairs[2 * p], b = r.pairs[2 * p + 1];
    if (a < 0 || a >= N || b < 0 || b >= N) return bad("self pair node out of range");
    PairDev pd{H.nodes[a].shape_begin, H.nodes[a].shape_end, H.nodes[b].shape_begin, H.nodes[b].shape_end};
    H.pairs.push_back(pd);  // an empty range simply never hits; pair indices stay the caller's
    for (int s1 = pd.s1_begin; s1 < pd.s1_end; s1++)
      for (int s2 = pd.s2_begin; s2 < pd.s2_end; s2++) {
        H.spairs.insert(H.spairs.end(), {s1, s2, p, 0});
        double rr = (double)H.rshapes[s1].br + (double)H.rshapes[s2].br + std::max(margin, 0.0);
        H.spair_r.push_back(T(rr * (1.0 + 1e-4) + 1e-9));
      }
  }
  for (const ShapeDev<T>& r : H.rshapes) H.hot.insert(H.hot.end(), {r.half[0], r.half[1], r.half[2], T(r.kind)});
  for (const ShapeDev<T>& o : H.oshapes)
    H.hot.insert(H.hot.end(), {o.lt[0], o.lt[1], o.lt[2], T(o.kind), o.lq[0], o.lq[1], o.lq[2], o.lq[3],
                               o.half[0], o.half[1], o.half[2], T(0)});
  // self shape pairs by their later shape (walk order = shape index order)
  {
    const size_t L = H.rshapes.size(), NP = H.spair_r.size();
    H.sp_first_slot.assign(L, -1);
    std::vector<std::vector<int32_t>> by_late(L);
    for (size_t p = 0; p < NP; p++) {
      const int s1 = H.spairs[4 * p], s2 = H.spairs[4 * p + 1];
      by_late[(size_t)std::max(s1, s2)].push_back((int32_t)p);
      if (s1 != s2) H.sp_first_slot[(size_t)std::min(s1, s2)] = 0;
    }
    H.n_first = 0;
    for (size_t s = 0; s < L; s++)
      if (H.sp_first_slot[s] == 0) H.sp_first_slot[s] = H.n_first++;
    // a self shape pair's fourth word: LDS slot of its earlier shape (-1: both are the same shape)
    for (size_t p = 0; p < NP; p++) {
      const int s1 = H.spairs[4 * p], s2 = H.spairs[4 * p + 1];
      H.spairs[4 * p + 3] = s1 == s2 ? -1 : H.sp_first_slot[(size_t)std::min(s1, s2)];
    }
    H.sp_late_off.assign(1, 0);
    for (size_t s = 0; s < L; s++) {
      H.sp_late.insert(H.sp_late.end(), by_late[s].begin(), by_late[s].end());
      H.sp_late_off.push_back((int32_t)H.sp_late.size());
    }
    for (int32_t p : H.sp_late) {
      SpLate<T> r{};
      r.p = p;
      r.fs = H.spairs[4 * (size_t)p + 3];
      r.rr = H.spair_r[(size_t)p];
      H.sp_late_rec.push_back(r);
    }
  }
  return true;
}

}  // namespace gear_hip
