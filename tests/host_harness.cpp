// host_harness.cpp — TEST HARNESS.  Compiles the product's narrow-phase
// (gear_amd/csrc/gear_device.hpp, __host__ __device__) for the CPU so that the
// geometry can be checked against the oracle in the GPU-less container.
// Built host-only by tests/harness.py; never part of libgear_hip.so.
#include <hip/hip_runtime.h>

#include "../gear_amd/csrc/gear_device.hpp"
#include "../gear_amd/csrc/gear_scene.hpp"
#include "../gear_amd/csrc/gear_shard.hpp"

using namespace gear_hip;

template <class T>
static int run_pair(const gh_shape_desc* a, const gh_mesh_desc* ma, int nma, const gh_shape_desc* b,
                    const gh_mesh_desc* mb, int nmb, double margin) {
  HostTables<T> H;
  BuildError err{0, ""};
  std::vector<ShapeDev<T>> pa, pb;  // a mesh that is not a closed convex polytope comes back as one piece per triangle
  if (!detail::flatten_shape<T>(*a, ma, nma, -1, H, pa, err)) return err.code;
  if (!detail::flatten_shape<T>(*b, mb, nmb, -1, H, pb, err)) return err.code;
  auto ref = [](const ShapeDev<T>& s) {
    ShapeRef<T> r;
    r.kind = s.kind; r.half = {s.half[0], s.half[1], s.half[2]}; r.br = s.br;
    r.vb = s.v_begin; r.vc = s.v_count; r.fb = s.f_begin; r.fc = s.f_count;
    r.pn = s.prism_n;
    return r;
  };
  if (!(margin > 0)) return 0;
  LaneCounters lc{0, 0, 0, 0};
  const T* verts = H.verts.data();
  const T* planes = H.planes.data();
  for (const ShapeDev<T>& sa : pa)
    for (const ShapeDev<T>& sb : pb) {  // the flags are an OR over the pieces' pairs
      Pose<T> wa{{sa.lt[0], sa.lt[1], sa.lt[2]}, {sa.lq[0], sa.lq[1], sa.lq[2], sa.lq[3]}};
      Pose<T> wb{{sb.lt[0], sb.lt[1], sb.lt[2]}, {sb.lq[0], sb.lq[1], sb.lq[2], sb.lq[3]}};
      // bounding-sphere cull exactly as the kernels apply it
      V3<T> ca = wa.t, cb = wb.t;
      if (sa.kind == SK_MESH) ca = wa.t + qrot(wa.r, V3<T>{sa.bc[0], sa.bc[1], sa.bc[2]});
      if (sb.kind == SK_MESH) cb = wb.t + qrot(wb.r, V3<T>{sb.bc[0], sb.bc[1], sb.bc[2]});
      V3<T> d = ca - cb;
      T rr = (sa.br + sb.br + T(margin)) * (T(1) + T(1e-4));
      if (!(dot(d, d) < rr * rr)) continue;
      if (pair_within<T, false>(ref(sa), wa, ref(sb), wb, verts, planes, T(margin), lc)) return 1;
    }
  return 0;
}

extern "C" int hh_pair_within(const gh_shape_desc* a, const gh_mesh_desc* ma, int nma, const gh_shape_desc* b,
                              const gh_mesh_desc* mb, int nmb, double margin, int precision) {
  return precision == 32 ? run_pair<float>(a, ma, nma, b, mb, nmb, margin)
                         : run_pair<double>(a, ma, nma, b, mb, nmb, margin);
}

// the device group's sharding arithmetic (gear_shard.hpp): share k of G of n configurations
extern "C" void hh_shard_range(size_t n, size_t k, size_t G, size_t* lo, size_t* hi) { shard_range(n, k, G, lo, hi); }
// cut points of G shares of whole edges; cut: [G + 1]
extern "C" void hh_shard_edges(const uint32_t* edge_off, size_t n_edges, size_t G, size_t* cut) {
  const std::vector<size_t> c = shard_edges(edge_off, n_edges, G);
  for (size_t k = 0; k <= G; k++) cut[k] = c[k];
}
