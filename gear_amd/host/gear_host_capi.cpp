// gear_host_capi.cpp — plain-C entry points over gear_host.hpp so that the C++ host layer can be
// driven from the Python test-suite (ctypes) and from C callers.  Declarations and per-function
// reference citations: include/gear_host.h.  Built into gear_amd/libgear_host.so, which links
// libgear_hip.so; nothing here computes a collision or a pose.
#include "../../include/gear_host.h"

#include <cstdlib>
#include <cstring>
#include <sstream>

#include "gear_host.hpp"

using namespace gear;

namespace {

thread_local std::string g_err;

struct PlannerBox {
  int precision;
  std::unique_ptr<JointPathPlanner<float>> p32;
  std::unique_ptr<JointPathPlanner<double>> p64;
  std::string text;  // last string result
};

template <class F> int guarded(F&& f) {
  try {
    return f();
  } catch (const Error& e) {
    g_err = e.what();
    switch (e.kind) {
      case Error::Collision: return e.part == CollisionPart::Start ? GP_ERR_COLLISION_START : GP_ERR_COLLISION_END;
      case Error::SelfCollision: return e.part == CollisionPart::Start ? GP_ERR_SELF_COLLISION_START : GP_ERR_SELF_COLLISION_END;
      case Error::PathPlanFail: return GP_ERR_PATH_PLAN_FAIL;
      case Error::DofMismatch: return GP_ERR_DOF_MISMATCH;
      case Error::NotFound: return GP_ERR_NOT_FOUND;
      case Error::ParseError: return GP_ERR_PARSE;
      case Error::Io: return GP_ERR_IO;
      case Error::Gpu: return GP_ERR_GPU;
      default: return GP_ERR_OTHER;
    }
  } catch (const std::exception& e) {
    g_err = e.what();
    return GP_ERR_OTHER;
  }
}

std::vector<int> using_vec(const int32_t* u, int n) { return std::vector<int>(u, u + n); }

template <class N> std::vector<N> to_vec(const double* q, int n) {
  std::vector<N> v(n);
  for (int i = 0; i < n; i++) v[i] = (N)q[i];
  return v;
}

std::string join_names(const std::vector<std::string>& v) {
  std::string s;
  for (size_t i = 0; i < v.size(); i++) s += (i ? "\n" : "") + v[i];
  return s;
}

template <class N>
int plan_t(JointPathPlanner<N>& p, PlannerBox* box, const int32_t* u, int n_using, const double* start, const double* goal,
           const Compound* objects, int with_self, double* path, int cap_rows, int* rows, uint64_t* stats) {
  p.stats = rrt::Stats{};
  std::vector<std::vector<N>> out;
  int rc = GP_OK;
  try {
    out = with_self ? p.plan_avoid_self_collision(using_vec(u, n_using), to_vec<N>(start, n_using), to_vec<N>(goal, n_using))
                    : p.plan(using_vec(u, n_using), to_vec<N>(start, n_using), to_vec<N>(goal, n_using), *objects);
  } catch (const Error& e) {
    if (e.kind == Error::Collision) box->text = join_names(e.collision_link_names);
    else if (e.kind == Error::SelfCollision) {
      std::vector<std::string> v;
      for (auto& pr : e.collision_link_pairs) v.push_back(pr.first + ":" + pr.second);
      box->text = join_names(v);
    }
    rc = guarded([&]() -> int { throw e; });
  }
  if (stats) {
    stats[0] = p.stats.gpu_calls; stats[1] = p.stats.samples; stats[2] = p.stats.consumed;
    stats[3] = p.stats.iterations; stats[4] = p.stats.shortcuts; stats[5] = p.stats.eval_ns;
    stats[6] = p.stats.draw_ns; stats[7] = p.stats.apply_ns;
  }
  if (rc != GP_OK) return rc;
  *rows = (int)out.size();
  if ((int)out.size() > cap_rows) { g_err = "path buffer too small"; return GP_ERR_OTHER; }
  for (size_t r = 0; r < out.size(); r++)
    for (int c = 0; c < n_using; c++) path[r * n_using + c] = (double)out[r][c];
  return GP_OK;
}

}  // namespace

extern "C" {

const char* gp_last_error(void) { return g_err.c_str(); }

int gp_planner_create(const char* urdf_path, double margin, int precision, int device, double step_length, int max_try,
                      int num_smoothing, const char* const* self_pairs, int n_self_pairs, uint64_t seed, gp_planner** out) {
  if (!urdf_path || !out || (precision != 32 && precision != 64)) { g_err = "invalid argument"; return GP_ERR_OTHER; }
  *out = nullptr;
  return guarded([&]() -> int {
    std::vector<std::string> strs;
    for (int i = 0; i < n_self_pairs; i++) strs.emplace_back(self_pairs[i]);
    auto pairs = parse_colon_separated_pairs(strs);
    auto box = std::make_unique<PlannerBox>();
    box->precision = precision;
    if (precision == 32) {
      auto b = JointPathPlannerBuilder<float>::from_urdf_file(urdf_path);
      box->p32 = b.collision_check_margin((float)margin).step_length((float)step_length).max_try((size_t)max_try).num_smoothing((size_t)num_smoothing).self_collision_pairs(pairs).finalize();
      box->p32->device = device;
      box->p32->rng = Rng(seed);
    } else {
      auto b = JointPathPlannerBuilder<double>::from_urdf_file(urdf_path);
      box->p64 = b.collision_check_margin(margin).step_length(step_length).max_try((size_t)max_try).num_smoothing((size_t)num_smoothing).self_collision_pairs(pairs).finalize();
      box->p64->device = device;
      box->p64->rng = Rng(seed);
    }
    *out = reinterpret_cast<gp_planner*>(box.release());
    return GP_OK;
  });
}

void gp_planner_destroy(gp_planner* p) { delete reinterpret_cast<PlannerBox*>(p); }

#define GP_BOX(p) PlannerBox* box = reinterpret_cast<PlannerBox*>(p)
#define GP_DISPATCH(expr32, expr64) (box->precision == 32 ? (expr32) : (expr64))

int gp_planner_set_speculation(gp_planner* p, int speculate, int smoothing_lookahead, uint64_t seed) {
  GP_BOX(p);
  if (box->p32) { box->p32->speculate = speculate != 0; box->p32->smoothing_lookahead = (size_t)smoothing_lookahead; box->p32->rng = Rng(seed); }
  if (box->p64) { box->p64->speculate = speculate != 0; box->p64->smoothing_lookahead = (size_t)smoothing_lookahead; box->p64->rng = Rng(seed); }
  return GP_OK;
}

int gp_compound_from_urdf_file(const char* path, gp_compound** out) {
  return guarded([&]() -> int {
    *out = reinterpret_cast<gp_compound*>(new Compound(Compound::from_urdf_file(path)));
    return GP_OK;
  });
}

int gp_compound_from_shapes(const gh_shape_desc* shapes, int n, gp_compound** out) {
  return guarded([&]() -> int {
    std::vector<ShapeSpec> v;
    for (int i = 0; i < n; i++) {
      if (shapes[i].kind == GH_SHAPE_MESH) throw Error(Error::Other, "gp_compound_from_shapes takes primitives only");
      ShapeSpec s;
      s.kind = shapes[i].kind;
      for (int k = 0; k < 3; k++) { s.params[k] = shapes[i].params[k]; s.t[k] = shapes[i].t[k]; }
      for (int k = 0; k < 4; k++) s.q[k] = shapes[i].q[k];
      v.push_back(s);
    }
    auto c = std::make_unique<Compound>(std::move(v));
    *out = reinterpret_cast<gp_compound*>(c.release());
    return GP_OK;
  });
}

void gp_compound_destroy(gp_compound* c) { delete reinterpret_cast<Compound*>(c); }
int gp_compound_size(const gp_compound* c) { return (int)reinterpret_cast<const Compound*>(c)->shapes().size(); }

int gp_using_joints(gp_planner* p, const char* end_joint, int32_t* out, int cap) {
  GP_BOX(p);
  return guarded([&]() -> int {
    const Chain& ch = box->precision == 32 ? box->p32->collision_check_robot : box->p64->collision_check_robot;
    std::vector<int> u;
    if (!end_joint || !*end_joint) u = ch.movable_nodes();
    else {
      int e = ch.find(end_joint);
      if (e < 0) throw Error(Error::NotFound, std::string("Node name ") + end_joint + " not found");
      u = ch.serial_chain_from_end(e);
    }
    if ((int)u.size() > cap) throw Error(Error::Other, "using_joints buffer too small");
    for (size_t i = 0; i < u.size(); i++) out[i] = u[i];
    return (int)u.size();
  });
}

const char* gp_describe(gp_planner* p) {
  GP_BOX(p);
  const Chain& ch = box->precision == 32 ? box->p32->collision_check_robot : box->p64->collision_check_robot;
  const CollisionChecker& cc = box->precision == 32 ? box->p32->collision_checker : box->p64->collision_checker;
  std::ostringstream o;
  o.precision(17);
  o << "{\"nodes\":[";
  for (size_t i = 0; i < ch.nodes.size(); i++) {
    const Node& n = ch.nodes[i];
    o << (i ? "," : "") << "{\"name\":\"" << n.name << "\",\"parent\":" << n.parent << ",\"joint_type\":" << n.joint_type
      << ",\"has_limit\":" << (n.has_limit ? 1 : 0) << ",\"lo\":" << n.limits.min << ",\"hi\":" << n.limits.max
      << ",\"mimic_parent\":" << n.mimic_parent << ",\"mimic_mult\":" << n.mimic_multiplier << ",\"mimic_offset\":" << n.mimic_offset
      << ",\"axis\":[" << n.axis[0] << "," << n.axis[1] << "," << n.axis[2] << "],\"origin_t\":[" << n.origin_t[0] << "," << n.origin_t[1]
      << "," << n.origin_t[2] << "],\"origin_q\":[" << n.origin_q[0] << "," << n.origin_q[1] << "," << n.origin_q[2] << "," << n.origin_q[3]
      << "],\"shapes\":[";
    auto it = cc.name_collision_model_map.find(n.name);
    if (it != cc.name_collision_model_map.end())
      for (size_t s = 0; s < it->second.size(); s++) {
        const ShapeSpec& sp = it->second[s];
        o << (s ? "," : "") << "{\"kind\":" << sp.kind << ",\"params\":[" << sp.params[0] << "," << sp.params[1] << "," << sp.params[2]
          << "],\"t\":[" << sp.t[0] << "," << sp.t[1] << "," << sp.t[2] << "],\"q\":[" << sp.q[0] << "," << sp.q[1] << "," << sp.q[2] << ","
          << sp.q[3] << "]";
        if (sp.kind == GH_SHAPE_MESH) {  // the flattened mesh: what load_obj_mesh made of the file
          o << ",\"vertices\":[";
          for (size_t v = 0; v < sp.vertices.size(); v++) o << (v ? "," : "") << sp.vertices[v];
          o << "],\"triangles\":[";
          for (size_t v = 0; v < sp.triangles.size(); v++) o << (v ? "," : "") << sp.triangles[v];
          o << "]";
        }
        o << "}";
      }
    o << "]}";
  }
  o << "],\"prediction\":" << cc.prediction << "}";
  box->text = o.str();
  return box->text.c_str();
}

int gp_is_feasible(gp_planner* p, const int32_t* using_joints, int n_using, const double* q, const gp_compound* objects) {
  GP_BOX(p);
  const Compound& c = *reinterpret_cast<const Compound*>(objects);
  return guarded([&]() -> int {
    return GP_DISPATCH(box->p32->is_feasible(using_vec(using_joints, n_using), to_vec<float>(q, n_using), c),
                       box->p64->is_feasible(using_vec(using_joints, n_using), to_vec<double>(q, n_using), c)) ? 1 : 0;
  });
}

int gp_is_feasible_with_self(gp_planner* p, const int32_t* using_joints, int n_using, const double* q) {
  GP_BOX(p);
  return guarded([&]() -> int {
    return GP_DISPATCH(box->p32->is_feasible_with_self(using_vec(using_joints, n_using), to_vec<float>(q, n_using)),
                       box->p64->is_feasible_with_self(using_vec(using_joints, n_using), to_vec<double>(q, n_using))) ? 1 : 0;
  });
}

int gp_has_any_colliding(gp_planner* p, const gp_compound* objects) {
  GP_BOX(p);
  const Compound& c = *reinterpret_cast<const Compound*>(objects);
  return guarded([&]() -> int { return GP_DISPATCH(box->p32->has_any_colliding(c), box->p64->has_any_colliding(c)) ? 1 : 0; });
}

int gp_has_any_colliding_with_self(gp_planner* p) {
  GP_BOX(p);
  return guarded([&]() -> int { return GP_DISPATCH(box->p32->has_any_colliding_with_self(), box->p64->has_any_colliding_with_self()) ? 1 : 0; });
}

int gp_colliding_link_names(gp_planner* p, const gp_compound* objects, const char** names) {
  GP_BOX(p);
  const Compound& c = *reinterpret_cast<const Compound*>(objects);
  return guarded([&]() -> int {
    auto v = GP_DISPATCH(box->p32->colliding_link_names(c), box->p64->colliding_link_names(c));
    box->text = join_names(v);
    *names = box->text.c_str();
    return (int)v.size();
  });
}

int gp_colliding_link_names_with_self(gp_planner* p, const char** names) {
  GP_BOX(p);
  return guarded([&]() -> int {
    auto v = GP_DISPATCH(box->p32->colliding_link_names_with_self(), box->p64->colliding_link_names_with_self());
    std::vector<std::string> s;
    for (auto& pr : v) s.push_back(pr.first + ":" + pr.second);
    box->text = join_names(s);
    *names = box->text.c_str();
    return (int)v.size();
  });
}

int gp_is_feasible_batch(gp_planner* p, const int32_t* using_joints, int n_using, const double* q, size_t n, const gp_compound* objects,
                         int with_self, uint8_t* ok) {
  GP_BOX(p);
  const Compound* c = reinterpret_cast<const Compound*>(objects);
  return guarded([&]() -> int {
    std::vector<uint8_t> r = GP_DISPATCH(box->p32->is_feasible_batch(using_vec(using_joints, n_using), to_vec<float>(q, (int)(n * n_using)), c, with_self != 0),
                                         box->p64->is_feasible_batch(using_vec(using_joints, n_using), to_vec<double>(q, (int)(n * n_using)), c, with_self != 0));
    std::memcpy(ok, r.data(), n);
    return GP_OK;
  });
}

int gp_plan(gp_planner* p, const int32_t* using_joints, int n_using, const double* start, const double* goal, const gp_compound* objects,
            double* path, int cap_rows, int* rows, uint64_t* stats) {
  GP_BOX(p);
  if (!objects) { g_err = "objects missing"; return GP_ERR_OTHER; }
  const Compound* c = reinterpret_cast<const Compound*>(objects);
  return guarded([&]() -> int {
    return box->precision == 32 ? plan_t<float>(*box->p32, box, using_joints, n_using, start, goal, c, 0, path, cap_rows, rows, stats)
                                : plan_t<double>(*box->p64, box, using_joints, n_using, start, goal, c, 0, path, cap_rows, rows, stats);
  });
}

int gp_plan_avoid_self_collision(gp_planner* p, const int32_t* using_joints, int n_using, const double* start, const double* goal,
                                 double* path, int cap_rows, int* rows, uint64_t* stats) {
  GP_BOX(p);
  return guarded([&]() -> int {
    return box->precision == 32 ? plan_t<float>(*box->p32, box, using_joints, n_using, start, goal, nullptr, 1, path, cap_rows, rows, stats)
                                : plan_t<double>(*box->p64, box, using_joints, n_using, start, goal, nullptr, 1, path, cap_rows, rows, stats);
  });
}

const char* gp_last_collision_names(gp_planner* p) { return reinterpret_cast<PlannerBox*>(p)->text.c_str(); }

int gp_feasible_ik_seeds(gp_planner* p, const int32_t* using_joints, int n_using, const double* initial, size_t k,
                         const gp_compound* objects, double* seeds, uint8_t* ok) {
  GP_BOX(p);
  const Compound& c = *reinterpret_cast<const Compound*>(objects);
  return guarded([&]() -> int {
    if (box->precision == 32) {
      auto r = box->p32->feasible_ik_seeds(using_vec(using_joints, n_using), to_vec<float>(initial, n_using), k, c);
      for (size_t i = 0; i < r.first.size(); i++) seeds[i] = (double)r.first[i];
      std::memcpy(ok, r.second.data(), k);
    } else {
      auto r = box->p64->feasible_ik_seeds(using_vec(using_joints, n_using), to_vec<double>(initial, n_using), k, c);
      for (size_t i = 0; i < r.first.size(); i++) seeds[i] = r.first[i];
      std::memcpy(ok, r.second.data(), k);
    }
    return GP_OK;
  });
}

int gp_joint_positions(gp_planner* p, const int32_t* using_joints, int n_using, double* q) {
  GP_BOX(p);
  const Chain& ch = box->precision == 32 ? box->p32->collision_check_robot : box->p64->collision_check_robot;
  auto v = ch.joint_positions(using_vec(using_joints, n_using));
  for (int i = 0; i < n_using; i++) q[i] = v[i];
  return GP_OK;
}

const char* gp_joint_names(gp_planner* p) {
  GP_BOX(p);
  box->text = join_names(box->precision == 32 ? box->p32->joint_names() : box->p64->joint_names());
  return box->text.c_str();
}

// ---- rrt 0.4.0's stepping, as the host layer restates it (test support: the device replay of
// gh_check_edges_from_endpoints must give these samples bit for bit) ----
int gp_edge_line(const double* from, const double* to, int dim, double step, int precision, int mode, double* out, int cap_rows) {
  return guarded([&]() -> int {
    int rows;
    if (precision == 32) {
      std::vector<float> q;
      auto a = to_vec<float>(from, dim), b = to_vec<float>(to, dim);
      if (mode == GH_EDGE_WALK) rrt::walk_line<float>(a.data(), b.data(), (size_t)dim, (float)step, q);
      else rrt::connect_line<float>(a.data(), b.data(), (size_t)dim, (float)step, q);
      rows = (int)(q.size() / (size_t)dim);
      for (int i = 0; i < std::min(rows, cap_rows) * dim; i++) out[i] = (double)q[(size_t)i];
    } else {
      std::vector<double> q;
      if (mode == GH_EDGE_WALK) rrt::walk_line<double>(from, to, (size_t)dim, step, q);
      else rrt::connect_line<double>(from, to, (size_t)dim, step, q);
      rows = (int)(q.size() / (size_t)dim);
      for (int i = 0; i < std::min(rows, cap_rows) * dim; i++) out[i] = q[(size_t)i];
    }
    return rows;
  });
}

int gp_first_infeasible_from_endpoints(gp_planner* p, const int32_t* using_joints, int n_using, const double* from, const double* to,
                                       size_t n_edges, int mode, const gp_compound* objects, int with_self, int32_t* first_bad,
                                       uint32_t* n_samples) {
  GP_BOX(p);
  const Compound* c = reinterpret_cast<const Compound*>(objects);
  return guarded([&]() -> int {
    const int n = (int)(n_edges * (size_t)n_using);
    auto r = GP_DISPATCH(box->p32->first_infeasible_from_endpoints(using_vec(using_joints, n_using), to_vec<float>(from, n), to_vec<float>(to, n),
                                                                   (uint32_t)mode, c, with_self != 0),
                         box->p64->first_infeasible_from_endpoints(using_vec(using_joints, n_using), to_vec<double>(from, n), to_vec<double>(to, n),
                                                                   (uint32_t)mode, c, with_self != 0));
    std::memcpy(first_bad, r.first.data(), n_edges * sizeof(int32_t));
    std::memcpy(n_samples, r.second.data(), n_edges * sizeof(uint32_t));
    return GP_OK;
  });
}

// ---- src/funcs.rs helpers, callable without a planner (host arithmetic only: no GPU involved) ----
namespace {
Limits limits_from_c(const double* lo, const double* hi, const uint8_t* has, int n) {
  Limits l((size_t)n);
  for (int i = 0; i < n; i++) l[(size_t)i] = {has[i] != 0, Range{lo[i], hi[i]}};
  return l;
}
}  // namespace

int gp_generate_clamped_joint_positions_from_limits(const double* angles, int n_angles, const double* lo, const double* hi,
                                                    const uint8_t* has_limit, int n_limits, int precision, double* out) {
  return guarded([&]() -> int {
    const Limits l = limits_from_c(lo, hi, has_limit, n_limits);
    if (precision == 32) {
      auto r = generate_clamped_joint_positions_from_limits<float>(to_vec<float>(angles, n_angles), l);
      for (size_t i = 0; i < r.size(); i++) out[i] = (double)r[i];
    } else {
      auto r = generate_clamped_joint_positions_from_limits<double>(to_vec<double>(angles, n_angles), l);
      for (size_t i = 0; i < r.size(); i++) out[i] = r[i];
    }
    return GP_OK;
  });
}

int gp_generate_random_joint_positions_from_limits(const double* lo, const double* hi, const uint8_t* has_limit, int n_limits,
                                                   int precision, uint64_t seed, size_t k, double* out) {
  return guarded([&]() -> int {
    const Limits l = limits_from_c(lo, hi, has_limit, n_limits);
    Rng rng(seed);
    for (size_t r = 0; r < k; r++) {
      if (precision == 32) {
        auto v = generate_random_joint_positions_from_limits<float>(l, rng);
        for (size_t i = 0; i < v.size(); i++) out[r * (size_t)n_limits + i] = (double)v[i];
      } else {
        auto v = generate_random_joint_positions_from_limits<double>(l, rng);
        for (size_t i = 0; i < v.size(); i++) out[r * (size_t)n_limits + i] = v[i];
      }
    }
    return GP_OK;
  });
}

int gp_modify_to_nearest_angle(const double* vec1, double* vec2, int n, const double* lo, const double* hi,
                               const uint8_t* has_limit, int precision) {
  return guarded([&]() -> int {
    const Limits l = limits_from_c(lo, hi, has_limit, n);
    if (precision == 32) {
      auto b = to_vec<float>(vec2, n);
      modify_to_nearest_angle<float>(to_vec<float>(vec1, n), b, l);
      for (int i = 0; i < n; i++) vec2[i] = (double)b[(size_t)i];
    } else {
      auto b = to_vec<double>(vec2, n);
      modify_to_nearest_angle<double>(to_vec<double>(vec1, n), b, l);
      for (int i = 0; i < n; i++) vec2[i] = b[(size_t)i];
    }
    return GP_OK;
  });
}

}  // extern "C"
