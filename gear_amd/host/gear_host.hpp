// gear_host.hpp — the host side of the hot path in C++17, above the C ABI of
// include/gear_hip.h.  The reference is a Rust crate (no cargo/rustc in this image), so the
// caller of the batched engine is written in the other compiled language at hand; names,
// argument meaning and error behaviour follow the reference:
//
//   gear::JointPathPlanner / JointPathPlannerBuilder   src/planner/joint_path_planner.rs:26-365
//   gear::CollisionChecker, Compound (FromUrdf)        src/collision/collision_checker.rs:202-262,349-369
//   urdf_geometry_to_shape_handle                      src/collision/urdf.rs:10-73
//   funcs                                              src/funcs.rs:28-119
//   seed generation of RandomInitializeIKSolver        src/ik.rs:67-79
//   rrt::dual_rrt_connect / smooth_path                crate rrt 0.4.0 (un-vendored; restated)
//   k::Chain wiring, set_joint_positions               crate k 0.20   (un-vendored; restated)
//
// What changes against the reference is only HOW the feasibility closure is evaluated: the
// samples of one RRT `connect` line, or of a batch of smoothing shortcuts, are handed to the
// GPU together (gh_check_edges) and the first infeasible sample of each line comes back, which
// is where the reference's sample-by-sample loop would have stopped.  With `speculate = false`
// every sample is evaluated on its own; both modes return the same path for the same seed
// (tests/test_gpu_host_cpp.py).
//
// No GPU arithmetic happens here and there is no CPU fallback for it: every predicate goes
// through libgear_hip.so.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/gear_hip.h"

namespace gear {

// ---------------------------------------------------------------------------
// errors (src/errors.rs:20-69)
// ---------------------------------------------------------------------------
enum class CollisionPart { Start, End };

struct Error : std::runtime_error {
  enum Kind { Other, Collision, SelfCollision, PathPlanFail, DofMismatch, NotFound, ParseError, Io, Gpu };
  Kind kind;
  CollisionPart part = CollisionPart::Start;
  std::vector<std::string> collision_link_names;                         // Error::Collision
  std::vector<std::pair<std::string, std::string>> collision_link_pairs;  // Error::SelfCollision
  Error(Kind k, const std::string& m) : std::runtime_error(m), kind(k) {}
};

inline Error gpu_error(int rc) { return Error(Error::Gpu, std::string("gear_hip error ") + std::to_string(rc) + ": " + gh_last_error()); }

// ---------------------------------------------------------------------------
// a small XML reader: elements, attributes, comments, declarations; text is ignored
// ---------------------------------------------------------------------------
struct XmlNode {
  std::string tag;
  std::vector<std::pair<std::string, std::string>> attrs;
  std::vector<XmlNode> children;
  const std::string* attr(const std::string& k) const {
    for (auto& a : attrs)
      if (a.first == k) return &a.second;
    return nullptr;
  }
  const XmlNode* child(const std::string& t) const {
    for (auto& c : children)
      if (c.tag == t) return &c;
    return nullptr;
  }
};

namespace detail {
struct XmlReader {
  const std::string& s;
  size_t p = 0;
  explicit XmlReader(const std::string& text) : s(text) {}
  [[noreturn]] void fail(const char* what) const { throw Error(Error::ParseError, std::string("XML: ") + what + " at offset " + std::to_string(p)); }
  void skip_ws() { while (p < s.size() && std::isspace((unsigned char)s[p])) p++; }
  bool starts(const char* lit) const { return s.compare(p, std::strlen(lit), lit) == 0; }
  void skip_misc() {  // whitespace, text, comments, declarations, up to the next element tag or '</'
    for (;;) {
      while (p < s.size() && s[p] != '<') p++;
      if (p >= s.size()) return;
      if (starts("<!--")) { size_t e = s.find("-->", p); if (e == std::string::npos) fail("unterminated comment"); p = e + 3; }
      else if (starts("<?")) { size_t e = s.find("?>", p); if (e == std::string::npos) fail("unterminated declaration"); p = e + 2; }
      else if (starts("<!")) { size_t e = s.find('>', p); if (e == std::string::npos) fail("unterminated <!"); p = e + 1; }
      else return;
    }
  }
  std::string name() {
    size_t b = p;
    while (p < s.size() && (std::isalnum((unsigned char)s[p]) || s[p] == '_' || s[p] == '-' || s[p] == ':' || s[p] == '.')) p++;
    if (p == b) fail("name expected");
    return s.substr(b, p - b);
  }
  XmlNode element() {
    if (s[p] != '<') fail("'<' expected");
    p++;
    XmlNode n;
    n.tag = name();
    for (;;) {
      skip_ws();
      if (p >= s.size()) fail("unterminated tag");
      if (s[p] == '/') { if (!starts("/>")) fail("'/>' expected"); p += 2; return n; }
      if (s[p] == '>') { p++; break; }
      std::string k = name();
      skip_ws();
      if (p >= s.size() || s[p] != '=') fail("'=' expected");
      p++;
      skip_ws();
      if (p >= s.size() || (s[p] != '"' && s[p] != '\'')) fail("quote expected");
      char qc = s[p++];
      size_t e = s.find(qc, p);
      if (e == std::string::npos) fail("unterminated attribute");
      n.attrs.emplace_back(k, s.substr(p, e - p));
      p = e + 1;
    }
    for (;;) {
      skip_misc();
      if (p >= s.size()) fail("unterminated element");
      if (starts("</")) {
        p += 2;
        if (name() != n.tag) fail("mismatched closing tag");
        skip_ws();
        if (p >= s.size() || s[p] != '>') fail("'>' expected");
        p++;
        return n;
      }
      n.children.push_back(element());
    }
  }
};
}  // namespace detail

inline XmlNode parse_xml(const std::string& text) {
  detail::XmlReader r(text);
  r.skip_misc();
  if (r.p >= text.size()) throw Error(Error::ParseError, "XML: no root element");
  return r.element();
}

// ---------------------------------------------------------------------------
// URDF subset, as urdf-rs 0.4 reads it (links' <collision> geometry, joints)
// ---------------------------------------------------------------------------
namespace urdf {

struct Geometry {
  enum Kind { Box, Cylinder, Sphere, Mesh } kind = Box;
  double size[3] = {0, 0, 0};
  double radius = 0, length = 0;
  std::string filename;
  double scale[3] = {1, 1, 1};
};
struct Collision { double xyz[3] = {0, 0, 0}, rpy[3] = {0, 0, 0}; Geometry geometry; };
struct Link { std::string name; std::vector<Collision> collisions; };
struct Joint {
  std::string name, type, parent, child, mimic_joint;
  double xyz[3] = {0, 0, 0}, rpy[3] = {0, 0, 0}, axis[3] = {1, 0, 0};
  double lower = 0, upper = 0, mimic_multiplier = 1, mimic_offset = 0;
};
struct Robot { std::string name, base_dir; std::vector<Link> links; std::vector<Joint> joints; };

namespace detail {
inline void floats(const std::string* text, int n, double* out) {
  if (!text) return;  // keep the defaults
  std::istringstream is(*text);
  std::vector<double> v;
  double x;
  while (is >> x) v.push_back(x);
  if ((int)v.size() != n) throw Error(Error::ParseError, "URDF: expected " + std::to_string(n) + " numbers in '" + *text + "'");
  for (int i = 0; i < n; i++) out[i] = v[i];
}
inline double number(const std::string* text, double dflt) { return text ? std::stod(*text) : dflt; }
inline void origin(const XmlNode* e, double* xyz, double* rpy) {
  if (!e) return;
  floats(e->attr("xyz"), 3, xyz);
  floats(e->attr("rpy"), 3, rpy);
}
inline Geometry geometry(const XmlNode& e) {
  Geometry g;
  for (auto& c : e.children) {
    if (c.tag == "box") { g.kind = Geometry::Box; floats(c.attr("size"), 3, g.size); return g; }
    if (c.tag == "cylinder") { g.kind = Geometry::Cylinder; g.radius = number(c.attr("radius"), 0); g.length = number(c.attr("length"), 0); return g; }
    if (c.tag == "sphere") { g.kind = Geometry::Sphere; g.radius = number(c.attr("radius"), 0); return g; }
    if (c.tag == "mesh") { g.kind = Geometry::Mesh; if (auto f = c.attr("filename")) g.filename = *f; floats(c.attr("scale"), 3, g.scale); return g; }
  }
  throw Error(Error::ParseError, "URDF: unsupported <geometry>");
}
}  // namespace detail

inline Robot parse(const std::string& text, const std::string& base_dir = "") {
  XmlNode root = parse_xml(text);
  if (root.tag != "robot") throw Error(Error::ParseError, "URDF: not a <robot>");
  Robot r;
  r.base_dir = base_dir;
  if (auto n = root.attr("name")) r.name = *n;
  for (auto& e : root.children) {
    if (e.tag == "link") {
      Link l;
      if (auto n = e.attr("name")) l.name = *n;
      for (auto& ce : e.children) {
        if (ce.tag != "collision") continue;
        Collision c;
        detail::origin(ce.child("origin"), c.xyz, c.rpy);
        const XmlNode* ge = ce.child("geometry");
        if (!ge) throw Error(Error::ParseError, "URDF: <collision> without <geometry>");
        c.geometry = detail::geometry(*ge);
        l.collisions.push_back(c);
      }
      r.links.push_back(l);
    } else if (e.tag == "joint") {
      Joint j;
      if (auto n = e.attr("name")) j.name = *n;
      if (auto t = e.attr("type")) j.type = *t;
      detail::origin(e.child("origin"), j.xyz, j.rpy);
      const XmlNode *pe = e.child("parent"), *ce = e.child("child");
      if (!pe || !ce || !pe->attr("link") || !ce->attr("link")) throw Error(Error::ParseError, "URDF: joint '" + j.name + "' needs parent and child links");
      j.parent = *pe->attr("link");
      j.child = *ce->attr("link");
      if (auto ae = e.child("axis")) detail::floats(ae->attr("xyz"), 3, j.axis);
      if (auto le = e.child("limit")) { j.lower = detail::number(le->attr("lower"), 0); j.upper = detail::number(le->attr("upper"), 0); }
      if (auto me = e.child("mimic")) {
        if (auto n = me->attr("joint")) j.mimic_joint = *n;
        j.mimic_multiplier = detail::number(me->attr("multiplier"), 1);
        j.mimic_offset = detail::number(me->attr("offset"), 0);
      }
      r.joints.push_back(j);
    }
  }
  return r;
}

inline Robot read_file(const std::string& path) {
  std::ifstream f(path);
  if (!f) throw Error(Error::Io, "cannot open " + path);
  std::stringstream ss;
  ss << f.rdbuf();
  size_t slash = path.find_last_of('/');
  return parse(ss.str(), slash == std::string::npos ? "." : path.substr(0, slash));
}

}  // namespace urdf

// nalgebra UnitQuaternion::from_euler_angles(roll, pitch, yaw) -> (i, j, k, w), in f64
inline void quat_from_rpy(const double rpy[3], double q[4]) {
  double sr = std::sin(rpy[0] * 0.5), cr = std::cos(rpy[0] * 0.5);
  double sp = std::sin(rpy[1] * 0.5), cp = std::cos(rpy[1] * 0.5);
  double sy = std::sin(rpy[2] * 0.5), cy = std::cos(rpy[2] * 0.5);
  q[0] = sr * cp * cy - cr * sp * sy;
  q[1] = cr * sp * cy + sr * cp * sy;
  q[2] = cr * cp * sy - sr * sp * cy;
  q[3] = cr * cp * cy + sr * sp * sy;
}

// ---------------------------------------------------------------------------
// k::Chain: one node per joint under a synthetic fixed "root", depth-first with a stack
// (children are visited in REVERSE insertion order — the order the reference's known-answer
// test pins, joint_path_planner.rs:383-418)
// ---------------------------------------------------------------------------
struct Range { double min, max; };

struct Node {
  std::string name;
  int joint_type = GH_JOINT_FIXED;
  double axis[3] = {1, 0, 0}, origin_t[3] = {0, 0, 0}, origin_q[4] = {0, 0, 0, 1};
  bool has_limit = false;
  Range limits{0, 0};
  int parent = -1;
  std::vector<int> children;
  int mimic_parent = -1;
  double mimic_multiplier = 1, mimic_offset = 0;
  double position = 0;
  bool is_movable() const { return joint_type != GH_JOINT_FIXED; }
};

struct Chain {
  std::vector<Node> nodes;

  static Chain from_urdf(const urdf::Robot& robot) {
    std::vector<Node> tmp(1);
    tmp[0].name = "root";
    std::map<std::string, int> child_link_to_node, joint_name_to_node;
    std::map<std::string, std::vector<int>> parent_link_to_nodes;
    for (auto& j : robot.joints) {
      Node n;
      n.name = j.name;
      n.joint_type = (j.type == "revolute" || j.type == "continuous") ? GH_JOINT_ROTATIONAL : j.type == "prismatic" ? GH_JOINT_LINEAR : GH_JOINT_FIXED;
      for (int k = 0; k < 3; k++) { n.axis[k] = j.axis[k]; n.origin_t[k] = j.xyz[k]; }
      quat_from_rpy(j.rpy, n.origin_q);
      n.has_limit = (j.upper - j.lower) != 0.0;
      n.limits = {j.lower, j.upper};
      int idx = (int)tmp.size();
      tmp.push_back(n);
      child_link_to_node[j.child] = idx;
      parent_link_to_nodes[j.parent].push_back(idx);
      joint_name_to_node[j.name] = idx;
    }
    for (auto& l : robot.links) {
      auto it = child_link_to_node.find(l.name);
      if (it == child_link_to_node.end()) continue;
      auto kids = parent_link_to_nodes.find(l.name);
      if (kids == parent_link_to_nodes.end()) continue;
      for (int cn : kids->second) { tmp[cn].parent = it->second; tmp[it->second].children.push_back(cn); }
    }
    for (size_t i = 1; i < tmp.size(); i++)
      if (tmp[i].parent < 0) { tmp[i].parent = 0; tmp[0].children.push_back((int)i); }
    for (auto& j : robot.joints) {
      if (j.mimic_joint.empty()) continue;
      auto p = joint_name_to_node.find(j.mimic_joint);
      if (p == joint_name_to_node.end()) throw Error(Error::NotFound, "Node name " + j.mimic_joint + " not found");
      Node& c = tmp[joint_name_to_node[j.name]];
      c.mimic_parent = p->second;
      c.mimic_multiplier = j.mimic_multiplier;
      c.mimic_offset = j.mimic_offset;
    }
    std::vector<int> order, stack{0};
    while (!stack.empty()) {
      int n = stack.back();
      stack.pop_back();
      order.push_back(n);
      for (int c : tmp[n].children) stack.push_back(c);
    }
    std::vector<int> remap(tmp.size(), -1);
    for (size_t i = 0; i < order.size(); i++) remap[order[i]] = (int)i;
    Chain ch;
    for (int old : order) {
      Node n = tmp[old];
      n.parent = n.parent >= 0 ? remap[n.parent] : -1;
      for (int& c : n.children) c = remap[c];
      n.mimic_parent = n.mimic_parent >= 0 ? remap[n.mimic_parent] : -1;
      ch.nodes.push_back(n);
    }
    return ch;
  }

  int find(const std::string& joint_name) const {  // k::Chain::find: by JOINT name; -1 when absent
    for (size_t i = 0; i < nodes.size(); i++)
      if (nodes[i].name == joint_name) return (int)i;
    return -1;
  }
  std::vector<int> movable_nodes() const {  // iter_joints()
    std::vector<int> v;
    for (size_t i = 0; i < nodes.size(); i++)
      if (nodes[i].is_movable()) v.push_back((int)i);
    return v;
  }
  std::vector<int> serial_chain_from_end(int end) const {  // k::SerialChain::from_end(node).iter_joints()
    std::vector<int> path;
    for (int n = end; n >= 0; n = nodes[n].parent) path.push_back(n);
    std::reverse(path.begin(), path.end());
    std::vector<int> v;
    for (int i : path)
      if (nodes[i].is_movable()) v.push_back(i);
    return v;
  }
  std::vector<double> joint_positions(const std::vector<int>& using_joints) const {
    std::vector<double> q;
    for (int i : using_joints) q.push_back(nodes[i].position);
    return q;
  }
  // k::Chain::set_joint_positions: Err on a length mismatch (thrown) or on the first value outside
  // its Range (returns false), leaving the joints set so far where they are.
  bool set_joint_positions(const std::vector<int>& using_joints, const double* q, size_t n) {
    if (n != using_joints.size()) throw Error(Error::DofMismatch, "DoF mismatch " + std::to_string(n) + " != " + std::to_string(using_joints.size()));
    for (size_t c = 0; c < n; c++) {
      Node& nd = nodes[using_joints[c]];
      if (nd.mimic_parent >= 0) continue;  // a mimic joint ignores its own input
      double v = q[c];
      if (nd.has_limit && !(v >= nd.limits.min && v <= nd.limits.max)) return false;
      nd.position = v;
      for (auto& ch : nodes) {
        if (ch.mimic_parent != using_joints[c] || !ch.is_movable()) continue;
        double cv = ch.mimic_offset + ch.mimic_multiplier * v;
        if (ch.has_limit && !(cv >= ch.limits.min && cv <= ch.limits.max)) return false;
        ch.position = cv;
      }
    }
    return true;
  }
  std::vector<std::string> joint_names() const {
    std::vector<std::string> v;
    for (auto& n : nodes)
      if (n.is_movable()) v.push_back(n.name);
    return v;
  }
  // j.limits of iter_joints() as funcs.rs uses them (None for a joint without a Range)
  std::vector<std::pair<bool, Range>> limits(const std::vector<int>& using_joints) const {
    std::vector<std::pair<bool, Range>> v;
    for (int i : using_joints) v.emplace_back(nodes[i].has_limit, nodes[i].limits);
    return v;
  }
};

using Limits = std::vector<std::pair<bool, Range>>;

// ---------------------------------------------------------------------------
// shapes, CollisionChecker, Compound
// ---------------------------------------------------------------------------
struct ShapeSpec {  // one (ShapeHandle, Isometry3)
  int kind = GH_SHAPE_BOX;
  double params[3] = {0, 0, 0}, t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1};
  std::vector<double> vertices;    // mesh: [n][3]
  std::vector<int32_t> triangles;  // mesh: [m][3]
};

// Wavefront-OBJ triangles, per-axis scale, sub-meshes merged: stands in for the assimp import of
// src/collision/mesh.rs:39-69 (triangles only: faces of any other size are dropped, :55-65; assimp holds
// f32 coordinates and the scale is applied in f64, `v.x as f64 * scale[0]`, :49-53; the indices of every
// sub-mesh continue after the previous one's vertices, :66)
inline bool load_obj_mesh(const std::string& path, const double scale[3], ShapeSpec& out) {
  std::ifstream f(path);
  if (!f) return false;
  std::string line;
  while (std::getline(f, line)) {
    std::istringstream is(line);
    std::string tok;
    if (!(is >> tok)) continue;
    if (tok == "v") {
      double x, y, z;
      if (is >> x >> y >> z) { out.vertices.push_back((double)(float)x * scale[0]); out.vertices.push_back((double)(float)y * scale[1]); out.vertices.push_back((double)(float)z * scale[2]); }
    } else if (tok == "f") {
      std::vector<int> idx;
      while (is >> tok) {
        int i = std::stoi(tok.substr(0, tok.find('/')));
        idx.push_back(i > 0 ? i - 1 : (int)(out.vertices.size() / 3) + i);
      }
      if (idx.size() == 3) out.triangles.insert(out.triangles.end(), idx.begin(), idx.end());
    }
  }
  return !out.vertices.empty() && !out.triangles.empty();
}

// urdf_geometry_to_shape_handle (src/collision/urdf.rs:10-73) without the pose; false = None
inline bool shape_from_urdf_geometry(const urdf::Geometry& g, const std::string& base_dir, ShapeSpec& s) {
  switch (g.kind) {
    case urdf::Geometry::Box:
      s.kind = GH_SHAPE_BOX;
      for (int k = 0; k < 3; k++) s.params[k] = g.size[k] * 0.5;
      return true;
    case urdf::Geometry::Cylinder:
      s.kind = GH_SHAPE_CYLINDER;
      s.params[0] = g.radius;
      s.params[1] = g.length;
      return true;
    case urdf::Geometry::Sphere:
      s.kind = GH_SHAPE_BALL;
      s.params[0] = g.radius;
      return true;
    case urdf::Geometry::Mesh: {
      std::string path = g.filename;
      if (path.rfind("package://", 0) != 0 && !base_dir.empty() && !path.empty() && path[0] != '/') path = base_dir + "/" + path;
      s.kind = GH_SHAPE_MESH;
      return load_obj_mesh(path, g.scale, s);  // a missing file is logged and skipped in the reference (urdf.rs:60-63)
    }
  }
  return false;
}

inline std::vector<ShapeSpec> collision_specs(const urdf::Link& link, const std::string& base_dir) {
  std::vector<ShapeSpec> out;
  for (auto& c : link.collisions) {
    ShapeSpec s;
    if (!shape_from_urdf_geometry(c.geometry, base_dir, s)) continue;
    for (int k = 0; k < 3; k++) s.t[k] = c.xyz[k];
    quat_from_rpy(c.rpy, s.q);
    out.push_back(std::move(s));
  }
  return out;
}

// gear::CollisionChecker's data (collision_checker.rs:202-262): joint name -> shapes, and the margin
struct CollisionChecker {
  std::map<std::string, std::vector<ShapeSpec>> name_collision_model_map;
  double prediction = 0.0;

  static CollisionChecker from_urdf_robot_with_base_dir(const urdf::Robot& robot, const std::string& base_dir, double prediction) {
    CollisionChecker c;
    c.prediction = prediction;
    std::map<std::string, std::string> link_joint;  // k::urdf::link_to_joint_map
    for (auto& j : robot.joints) link_joint[j.child] = j.name;
    for (auto& l : robot.links) {
      auto specs = collision_specs(l, base_dir);
      if (specs.empty()) continue;
      auto it = link_joint.find(l.name);
      c.name_collision_model_map[it == link_joint.end() ? "root" : it->second] = std::move(specs);
    }
    return c;
  }
  static CollisionChecker from_urdf_robot(const urdf::Robot& robot, double prediction) {
    return from_urdf_robot_with_base_dir(robot, robot.base_dir, prediction);
  }
};

// ncollide3d Compound<f64> as gear builds it (collision_checker.rs:349-369): a flat list of posed shapes
// Like ncollide's, a Compound is immutable once built (Compound::new(shapes) / shapes()): the planner
// caches one GPU scene per compound, keyed on `id()`.  An id is drawn from a process-wide counter when the
// shape list is set and travels with copies of that list, so equal ids always mean equal contents, and a
// compound that is destroyed and rebuilt (even at the same address, with the same shape count) gets a new one.
class Compound {
 public:
  Compound() : id_(next_id()) {}
  explicit Compound(std::vector<ShapeSpec> shapes) : shapes_(std::move(shapes)), id_(next_id()) {}
  const std::vector<ShapeSpec>& shapes() const { return shapes_; }
  uint64_t id() const { return id_; }
  static Compound from_urdf_robot(const urdf::Robot& robot) {
    std::vector<ShapeSpec> v;
    for (auto& l : robot.links)
      for (auto& s : collision_specs(l, "")) v.push_back(s);
    return Compound(std::move(v));
  }
  static Compound from_urdf_file(const std::string& path) { return from_urdf_robot(urdf::read_file(path)); }

 private:
  static uint64_t next_id() {
    static std::atomic<uint64_t> counter{1};
    return counter.fetch_add(1, std::memory_order_relaxed);
  }
  std::vector<ShapeSpec> shapes_;
  uint64_t id_;
};

// collision_checker.rs:314-329
inline std::vector<std::pair<std::string, std::string>> parse_colon_separated_pairs(const std::vector<std::string>& pair_strs) {
  std::vector<std::pair<std::string, std::string>> pairs;
  for (auto& s : pair_strs) {
    size_t c = s.find(':');
    if (c == std::string::npos) throw Error(Error::ParseError, "failed to parse " + s);
    size_t c2 = s.find(':', c + 1);
    pairs.emplace_back(s.substr(0, c), s.substr(c + 1, c2 == std::string::npos ? std::string::npos : c2 - c - 1));
  }
  return pairs;
}

// ---------------------------------------------------------------------------
// funcs (src/funcs.rs) and the random source
// ---------------------------------------------------------------------------
// The reference draws from rand 0.3's thread_rng (unseedable); this is a seedable stand-in
// (xoshiro256++ seeded through splitmix64), so plans can be reproduced.
struct Rng {
  uint64_t s[4];
  explicit Rng(uint64_t seed = 0) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull;
    for (auto& w : s) {
      uint64_t x = (z += 0x9E3779B97F4A7C15ull);
      x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
      x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
      w = x ^ (x >> 31);
    }
  }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next() {
    uint64_t r = rotl(s[0] + s[3], 23) + s[0], t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return r;
  }
  double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }  // [0, 1)
  size_t range(size_t lo, size_t hi) { return lo + (size_t)(uniform() * (double)(hi - lo)); }  // [lo, hi)
};

template <class N>
std::vector<N> generate_clamped_joint_positions_from_limits(const std::vector<N>& angles, const Limits& limits) {
  if (angles.size() != limits.size()) throw Error(Error::DofMismatch, "DoF mismatch " + std::to_string(angles.size()) + " != " + std::to_string(limits.size()));
  std::vector<N> out(angles);
  for (size_t i = 0; i < out.size(); i++)
    if (limits[i].first) {
      if (out[i] > (N)limits[i].second.max) out[i] = (N)limits[i].second.max;
      else if (out[i] < (N)limits[i].second.min) out[i] = (N)limits[i].second.min;
    }
  return out;
}

// funcs.rs:76-87: (max - min) * u + min, or (u - 0.5) * 2 pi without a Range
template <class N> std::vector<N> generate_random_joint_positions_from_limits(const Limits& limits, Rng& rng) {
  std::vector<N> v;
  v.reserve(limits.size());
  for (auto& l : limits) {
    if (l.first) v.push_back(((N)l.second.max - (N)l.second.min) * (N)rng.uniform() + (N)l.second.min);
    else v.push_back((N)(rng.uniform() - 0.5) * (N)(2.0 * 3.14159265358979323846));
  }
  return v;
}

// funcs.rs:98-119
template <class N> void modify_to_nearest_angle(const std::vector<N>& vec1, std::vector<N>& vec2, const Limits& limits) {
  for (size_t i = 0; i < vec1.size(); i++) {
    if (limits[i].first) continue;
    const N pi2 = (N)3.14159265358979323846 * (N)2.0;
    N dist1 = std::fabs(vec1[i] - vec2[i]), dist2 = std::fabs(vec1[i] - (vec2[i] - pi2));
    if (dist1 > dist2) vec2[i] -= pi2;
    else if (dist1 > std::fabs(vec1[i] - (vec2[i] + pi2))) vec2[i] += pi2;
  }
}

// ---------------------------------------------------------------------------
// Scene: a gh_scene with the descriptors it was built from
// ---------------------------------------------------------------------------
class Scene {
 public:
  Scene(const Chain& chain, const CollisionChecker& checker, const std::vector<int>& using_joints,
        const std::vector<std::pair<std::string, std::string>>& self_pairs, const Compound* objects, int precision, int device)
      : precision_(precision), dof_((int)using_joints.size()), n_nodes_((int)chain.nodes.size()) {
    std::vector<gh_node_desc> nodes(chain.nodes.size());
    for (size_t i = 0; i < nodes.size(); i++) {
      const Node& n = chain.nodes[i];
      gh_node_desc& d = nodes[i];
      std::memset(&d, 0, sizeof(d));
      d.parent = n.parent;
      d.joint_type = n.joint_type;
      for (int k = 0; k < 3; k++) { d.axis[k] = n.axis[k]; d.origin_t[k] = n.origin_t[k]; }
      for (int k = 0; k < 4; k++) d.origin_q[k] = n.origin_q[k];
      d.has_limit = n.has_limit ? 1 : 0;
      d.limit_lo = n.has_limit ? n.limits.min : 0.0;
      d.limit_hi = n.has_limit ? n.limits.max : 0.0;
      d.mimic_parent = n.mimic_parent;
      d.mimic_mult = n.mimic_multiplier;
      d.mimic_offset = n.mimic_offset;
      d.position = n.position;
    }
    std::vector<gh_shape_desc> rshapes, oshapes;
    std::vector<gh_mesh_desc> rmeshes, omeshes;
    for (size_t i = 0; i < chain.nodes.size(); i++) {  // EnvCollisionNames walks the nodes in chain order
      auto it = checker.name_collision_model_map.find(chain.nodes[i].name);
      if (it == checker.name_collision_model_map.end()) continue;
      for (auto& s : it->second) rshapes.push_back(shape_desc(s, (int)i, rmeshes));
    }
    std::vector<int32_t> pairs;
    for (auto& p : self_pairs) {  // pairs without a model or a node never yield (collision_checker.rs:150-169)
      if (!checker.name_collision_model_map.count(p.first) || !checker.name_collision_model_map.count(p.second)) continue;
      int a = chain.find(p.first), b = chain.find(p.second);
      if (a < 0 || b < 0) continue;
      pairs.push_back(a);
      pairs.push_back(b);
      pair_names_.push_back(p);
    }
    if (objects)
      for (auto& s : objects->shapes()) oshapes.push_back(shape_desc(s, -1, omeshes));
    std::vector<int32_t> cols(using_joints.begin(), using_joints.end());
    gh_robot_desc rd;
    std::memset(&rd, 0, sizeof(rd));
    rd.n_nodes = (int)nodes.size();
    rd.n_cols = (int)cols.size();
    rd.n_shapes = (int)rshapes.size();
    rd.n_pairs = (int)pairs.size() / 2;
    rd.n_meshes = (int)rmeshes.size();
    rd.nodes = nodes.data();
    rd.col_node = cols.data();
    rd.shapes = rshapes.data();
    rd.pairs = pairs.data();
    rd.meshes = rmeshes.data();
    gh_obstacle_desc od;
    std::memset(&od, 0, sizeof(od));
    od.n_shapes = (int)oshapes.size();
    od.n_meshes = (int)omeshes.size();
    od.shapes = oshapes.data();
    od.meshes = omeshes.data();
    n_obstacles_ = od.n_shapes;
    int rc = gh_scene_create(&rd, &od, checker.prediction, precision, device, &h_);
    if (rc != GH_OK) throw gpu_error(rc);
  }
  ~Scene() { if (h_) gh_scene_destroy(h_); }
  Scene(const Scene&) = delete;
  Scene& operator=(const Scene&) = delete;

  gh_scene* handle() const { return h_; }
  int dof() const { return dof_; }
  int n_nodes() const { return n_nodes_; }
  int n_obstacles() const { return n_obstacles_; }
  const std::vector<std::pair<std::string, std::string>>& pair_names() const { return pair_names_; }

  // q: [n][dof] in the scene's scalar type
  void check_batch(const void* q, size_t n, uint32_t what, uint8_t* flags) const {
    int rc = gh_check_batch(h_, q, n, what, flags);
    if (rc != GH_OK) throw gpu_error(rc);
  }
  void check_edges_from_endpoints(const void* from, const void* to, size_t n_edges, double step, uint32_t max_len, uint32_t mode,
                                  uint32_t what, uint8_t bad_mask, int32_t* first_bad, uint32_t* n_samples) const {
    int rc = gh_check_edges_from_endpoints(h_, from, to, n_edges, step, max_len, mode, what, bad_mask, first_bad, n_samples);
    if (rc != GH_OK) throw gpu_error(rc);
  }
  void check_edges(const void* q, const uint32_t* edge_off, size_t n_edges, uint32_t what, uint8_t bad_mask, uint8_t* flags, int32_t* first_bad) const {
    int rc = gh_check_edges(h_, q, edge_off, n_edges, what, bad_mask, flags, first_bad);
    if (rc != GH_OK) throw gpu_error(rc);
  }
  std::vector<uint64_t> link_hits(const void* q, size_t n, uint32_t what = 0) const {
    std::vector<uint64_t> m(n * (size_t)n_obstacles_ * (size_t)gh_link_mask_words(h_));
    int rc = gh_link_hits(h_, q, n, what, m.data());
    if (rc != GH_OK) throw gpu_error(rc);
    return m;
  }
  std::vector<uint64_t> self_hits(const void* q, size_t n, uint32_t what = 0) const {
    std::vector<uint64_t> m(n * std::max<size_t>((pair_names_.size() + 63) / 64, 1));
    int rc = gh_self_hits(h_, q, n, what, m.data());
    if (rc != GH_OK) throw gpu_error(rc);
    return m;
  }

 private:
  static gh_shape_desc shape_desc(const ShapeSpec& s, int node, std::vector<gh_mesh_desc>& meshes) {
    gh_shape_desc d;
    std::memset(&d, 0, sizeof(d));
    d.kind = s.kind;
    d.node = node;
    for (int k = 0; k < 3; k++) { d.params[k] = s.params[k]; d.t[k] = s.t[k]; }
    for (int k = 0; k < 4; k++) d.q[k] = s.q[k];
    d.mesh = -1;
    if (s.kind == GH_SHAPE_MESH) {
      gh_mesh_desc m;
      m.n_vertices = (int)(s.vertices.size() / 3);
      m.n_triangles = (int)(s.triangles.size() / 3);
      m.vertices = s.vertices.data();
      m.triangles = s.triangles.data();
      d.mesh = (int)meshes.size();
      meshes.push_back(m);
    }
    return d;
  }
  gh_scene* h_ = nullptr;
  int precision_, dof_, n_nodes_, n_obstacles_ = 0;
  std::vector<std::pair<std::string, std::string>> pair_names_;
};

// ---------------------------------------------------------------------------
// rrt 0.4.0 restated over a batched line evaluator
// ---------------------------------------------------------------------------
namespace rrt {

struct Stats {
  uint64_t gpu_calls = 0;     // evaluator invocations
  uint64_t samples = 0;       // samples handed to the evaluator
  uint64_t consumed = 0;      // samples the sequential algorithm would have evaluated
  uint64_t iterations = 0;    // dual_rrt_connect loop iterations
  uint64_t shortcuts = 0;     // smoothing shortcuts applied
  uint64_t eval_ns = 0;       // wall time spent inside the evaluator (staging, launch, synchronisation)
  uint64_t draw_ns = 0;       // wall time spent drawing iterations ahead: random samples, nearest-vertex lookups, stepping
  uint64_t apply_ns = 0;      // wall time spent consuming the evaluated iterations (validation against the grown trees, tree growth)
};
inline uint64_t now_ns() {
  return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Evaluates lines of samples: `q` holds all samples row-major, line e = rows [off[e], off[e+1]).
// Returns per line the index of its first infeasible sample, or -1.
template <class N> using LineEval = std::function<std::vector<int32_t>(const std::vector<N>& q, const std::vector<uint32_t>& off)>;

template <class N> N distance(const N* a, const N* b, size_t d) {  // squared_euclidean(..).sqrt()
  N s = 0;
  for (size_t i = 0; i < d; i++) s += (a[i] - b[i]) * (a[i] - b[i]);
  return std::sqrt(s);
}

template <class N> struct Tree {
  size_t dim;
  const char* name;
  std::vector<N> data;      // vertices, row-major
  std::vector<int> parent;  // -1 for the root
  Tree(const char* nm, size_t d) : dim(d), name(nm) {}
  size_t size() const { return parent.size(); }
  const N* at(size_t i) const { return data.data() + i * dim; }
  int add_vertex(const N* q, int par) {
    data.insert(data.end(), q, q + dim);
    parent.push_back(par);
    return (int)parent.size() - 1;
  }
  // the crate asks a kd-tree; a linear scan returns the same vertex except on exact distance ties
  int get_nearest_index(const N* q) const {
    int best = 0;
    N bd = -1;
    for (size_t i = 0; i < size(); i++) {
      N s = 0;
      const N* v = at(i);
      for (size_t k = 0; k < dim; k++) s += (v[k] - q[k]) * (v[k] - q[k]);
      if (bd < 0 || s < bd) { bd = s; best = (int)i; }
    }
    return best;
  }
  std::vector<std::vector<N>> get_until_root(int index) const {
    std::vector<std::vector<N>> out;
    for (int i = index; i >= 0; i = parent[i]) out.emplace_back(at(i), at(i) + dim);
    return out;
  }
};

// one `extend` step from `from` towards `target` (Tree::extend's q_new)
template <class N> void step_towards(const N* from, const N* target, size_t dim, N extend_length, N* out) {
  N d = distance(from, target, dim);
  if (d < extend_length) { std::copy(target, target + dim, out); return; }
  for (size_t i = 0; i < dim; i++) out[i] = from[i] + (target[i] - from[i]) * extend_length / d;
}

// the samples `connect` would evaluate from vertex `from` towards `target`, assuming all are free: extend() after
// extend(), each from the vertex the last one added, until one lands within extend_length of the target.
// (extend() measures the distance to the target twice per step — before stepping and, for ExtendStatus::Reached,
// after it; the second measurement of one step IS the first of the next, same operands, same operations, so it
// is computed once here: the samples are bit for bit those of the step-by-step form.)
template <class N> void connect_line(const N* from, const N* target, size_t dim, N extend_length, std::vector<N>& out) {
  size_t at = out.size();
  out.resize(at + dim);
  N d = distance(from, target, dim);
  {
    N* nxt = out.data() + at;
    if (d < extend_length) std::copy(target, target + dim, nxt);
    else for (size_t i = 0; i < dim; i++) nxt[i] = from[i] + (target[i] - from[i]) * extend_length / d;
  }
  for (;;) {
    d = distance(out.data() + at, target, dim);
    if (d < extend_length) return;  // ExtendStatus::Reached
    out.resize(at + 2 * dim);
    const N* cur = out.data() + at;
    N* nxt = out.data() + at + dim;
    for (size_t i = 0; i < dim; i++) nxt[i] = cur[i] + (target[i] - cur[i]) * extend_length / d;  // (d >= extend_length here)
    at += dim;
  }
}

// the samples smooth_path's walker evaluates from `from` towards `target`, assuming all are free: it advances by
// extend_length while it is at least extend_length away; the target itself is not evaluated
template <class N> void walk_line(const N* from, const N* target, size_t dim, N extend_length, std::vector<N>& out) {
  N d = distance(from, target, dim);
  size_t at = out.size();
  std::vector<N> first(from, from + dim);  // `from` may point into `out`
  bool have = false;
  while (!(d < extend_length)) {
    out.resize(at + dim);
    const N* c = have ? out.data() + at - dim : first.data();
    N* nxt = out.data() + at;
    for (size_t i = 0; i < dim; i++) nxt[i] = c[i] + (target[i] - c[i]) * extend_length / d;
    d = distance(nxt, target, dim);
    at += dim;
    have = true;
  }
}

// rrt::dual_rrt_connect.  One iteration = one extend sample on tree_a and, when that is free, tree_b's
// connect line towards it.  tree_b's line does not depend on whether the extend sample is free, and an
// iteration whose extend sample is NOT free (ExtendStatus::Trapped) changes nothing but the roles of
// the trees — so with `lookahead` > 0 that many iterations are drawn ahead and their samples and lines
// evaluated in one call; the first iteration whose extend succeeds is applied, the random source is
// rewound to its state after that iteration, and the rest is discarded.  lookahead == 0 evaluates
// sample by sample, as the crate does; both give the same trees for the same seed.
template <class N>
std::vector<std::vector<N>> dual_rrt_connect(const std::vector<N>& start, const std::vector<N>& goal, const LineEval<N>& eval,
                                             const std::function<std::vector<N>(Rng&)>& random_sample, N extend_length,
                                             size_t num_max_try, size_t lookahead, Rng& rng, Stats& st) {
  const size_t dim = start.size();
  if (goal.size() != dim) throw Error(Error::DofMismatch, "start and goal differ in length");
  Tree<N> ta("start", dim), tb("goal", dim);
  ta.add_vertex(start.data(), -1);
  tb.add_vertex(goal.data(), -1);
  Tree<N>*a = &ta, *b = &tb;
  auto eval_counted = [&](const std::vector<N>& q, const std::vector<uint32_t>& off) {
    st.gpu_calls++;
    st.samples += off.back();
    return eval(q, off);
  };
  struct Iter { std::vector<N> q_rand, q_new, line; int near_a, near_b; uint32_t m; Rng after; };
  // is vertex `stored` still what get_nearest_index(q) returns now that the tree has grown past `old_size`?
  // (the scan keeps the first minimum, so a later vertex replaces it only when strictly nearer)
  auto still_nearest = [dim](const Tree<N>& t, const N* q, int stored, size_t old_size) {
    auto d2 = [&](const N* v) { N s = 0; for (size_t k = 0; k < dim; k++) s += (v[k] - q[k]) * (v[k] - q[k]); return s; };
    const N bd = d2(t.at((size_t)stored));
    for (size_t i = old_size; i < t.size(); i++)
      if (d2(t.at(i)) < bd) return false;
    return true;
  };
  const size_t ahead = std::max<size_t>(lookahead, 1);
  size_t it = 0;
  while (it < num_max_try) {
    const uint64_t t_draw = now_ns();
    std::vector<Iter> iters;
    iters.reserve(ahead);
    Rng r = rng;
    Tree<N>*ka = a, *kb = b;
    const size_t size_a0 = a->size(), size_b0 = b->size();  // the trees the speculation looked at
    for (size_t k = 0; k < ahead && it + k < num_max_try; k++) {
      Iter x;
      x.q_rand = random_sample(r);
      const std::vector<N>& q_rand = x.q_rand;
      x.near_a = ka->get_nearest_index(q_rand.data());
      x.q_new.resize(dim);
      step_towards(ka->at(x.near_a), q_rand.data(), dim, extend_length, x.q_new.data());
      x.near_b = kb->get_nearest_index(x.q_new.data());
      x.line.reserve(dim * 64);
      connect_line(kb->at(x.near_b), x.q_new.data(), dim, extend_length, x.line);
      x.m = (uint32_t)(x.line.size() / dim);
      x.after = r;
      iters.push_back(std::move(x));
      std::swap(ka, kb);
    }
    st.draw_ns += now_ns() - t_draw;
    std::vector<int32_t> bad;  // per iteration: [extend sample, line]
    if (lookahead > 0) {
      std::vector<N> q;
      std::vector<uint32_t> off{0u};
      size_t rows = 0;
      for (auto& x : iters) rows += 1 + x.m;
      q.reserve(rows * dim);
      off.reserve(2 * iters.size() + 1);
      for (auto& x : iters) {
        q.insert(q.end(), x.q_new.begin(), x.q_new.end());
        off.push_back(off.back() + 1);
        q.insert(q.end(), x.line.begin(), x.line.end());
        off.push_back(off.back() + x.m);
      }
      bad = eval_counted(q, off);
    } else {
      Iter& x = iters[0];
      bad = {eval_counted(x.q_new, {0u, 1u})[0], -1};
      if (bad[0] < 0)
        for (uint32_t k = 0; k < x.m; k++) {
          std::vector<N> one(x.line.begin() + k * dim, x.line.begin() + (k + 1) * dim);
          if (eval_counted(one, {0u, 1u})[0] >= 0) { bad[1] = (int32_t)k; break; }
        }
    }
    // Iterations are consumed in order.  One that was applied has grown both trees; the next one's evaluation
    // is still the sequential algorithm's as long as its two nearest-vertex lookups come out the same on the
    // grown trees (then its extend sample and its connect line are the same too).  The first iteration for
    // which they do not ends the batch: the random source goes back to its state before that iteration.
    const uint64_t t_apply = now_ns();
    size_t last = 0;
    bool grown = false;
    Tree<N>* const a0 = a;  // the tree whose size was size_a0
    for (size_t k = 0; k < iters.size(); k++) {
      Iter& x = iters[k];
      if (grown) {
        const size_t old_a = a == a0 ? size_a0 : size_b0, old_b = b == a0 ? size_a0 : size_b0;
        if (!still_nearest(*a, x.q_rand.data(), x.near_a, old_a) || !still_nearest(*b, x.q_new.data(), x.near_b, old_b)) break;
      }
      last = k;
      it++;
      st.iterations++;
      st.consumed++;
      const bool new_free = bad[2 * k] < 0;
      if (!new_free) { std::swap(a, b); continue; }  // Trapped
      const int32_t line_bad = bad[2 * k + 1];
      st.consumed += line_bad < 0 ? x.m : (uint32_t)line_bad + 1;
      const int new_index = a->add_vertex(x.q_new.data(), x.near_a);
      const uint32_t ok = line_bad < 0 ? x.m : (uint32_t)line_bad;
      int prev = x.near_b;
      for (uint32_t j = 0; j < ok; j++) prev = b->add_vertex(x.line.data() + j * dim, prev);
      if (line_bad < 0) {  // ExtendStatus::Reached
        auto a_all = a->get_until_root(new_index);
        auto b_all = b->get_until_root(prev);
        std::reverse(a_all.begin(), a_all.end());
        a_all.insert(a_all.end(), b_all.begin(), b_all.end());
        if (std::strcmp(b->name, "start") == 0) std::reverse(a_all.begin(), a_all.end());
        rng = x.after;
        st.apply_ns += now_ns() - t_apply;
        return a_all;
      }
      std::swap(a, b);
      grown = true;
    }
    rng = iters[last].after;
    st.apply_ns += now_ns() - t_apply;
  }
  throw Error(Error::PathPlanFail, "failed");
}

// rrt::smooth_path.  A trial draws (ind1, ind2) and walks from path[ind1] towards path[ind2]; when
// every step is free the points in between are removed.  A failed trial leaves the path untouched,
// so up to `speculate` trials are drawn ahead and their lines evaluated in one call; after the first
// successful one the random source is rewound to its state after that trial, which makes the result
// identical to evaluating trial by trial (speculate <= 1).
template <class N>
void smooth_path(std::vector<std::vector<N>>& path, const LineEval<N>& eval, N extend_length, size_t num_max_try, Rng& rng,
                 size_t speculate, Stats& st) {
  if (path.size() < 3) return;
  const size_t dim = path[0].size();
  const size_t ahead = std::max<size_t>(speculate, 1);
  size_t trial = 0;
  while (trial < num_max_try) {
    struct Trial { size_t ind1, ind2; uint32_t begin, end; Rng after; };
    std::vector<Trial> trials;
    std::vector<N> q;
    std::vector<uint32_t> off{0u};
    Rng r = rng;
    for (size_t k = 0; k < ahead && trial + k < num_max_try; k++) {
      const size_t ind1 = r.range(0, path.size() - 2);
      const size_t ind2 = r.range(ind1 + 2, path.size());
      // the samples the trial's loop evaluates while they stay free; it ends when the walker is
      // within extend_length of path[ind2]
      const uint32_t begin = (uint32_t)(q.size() / dim);
      walk_line(path[ind1].data(), path[ind2].data(), dim, extend_length, q);
      trials.push_back({ind1, ind2, begin, (uint32_t)(q.size() / dim), r});
      off.push_back((uint32_t)(q.size() / dim));
    }
    std::vector<int32_t> bad(trials.size(), -1);
    if (off.back() > 0) {
      if (speculate > 1) {
        st.gpu_calls++;
        st.samples += off.back();
        bad = eval(q, off);
      } else {  // sample by sample, stopping at the first failure
        for (uint32_t k = 0; k < off.back(); k++) {
          std::vector<N> one(q.begin() + k * dim, q.begin() + (k + 1) * dim);
          st.gpu_calls++;
          st.samples++;
          if (eval(one, {0u, 1u})[0] >= 0) { bad[0] = (int32_t)k; break; }
        }
      }
    }
    bool changed = false;
    for (size_t k = 0; k < trials.size(); k++) {
      trial++;
      const Trial& t = trials[k];
      st.consumed += bad[k] < 0 ? t.end - t.begin : (uint32_t)bad[k] + 1;
      if (bad[k] >= 0) continue;
      path.erase(path.begin() + t.ind1 + 1, path.begin() + t.ind2);
      st.shortcuts++;
      rng = t.after;
      changed = true;
      if (path.size() == 2) return;
      break;
    }
    if (!changed) rng = trials.back().after;
  }
}

}  // namespace rrt

// ---------------------------------------------------------------------------
// JointPathPlanner (src/planner/joint_path_planner.rs:26-256)
// ---------------------------------------------------------------------------
template <class N> class JointPathPlanner {
 public:
  Chain collision_check_robot;  // shared mutable joint state, like the reference's Rc<RefCell> nodes
  CollisionChecker collision_checker;
  N step_length;
  size_t max_try, num_smoothing;
  std::shared_ptr<urdf::Robot> urdf_robot;
  std::vector<std::pair<std::string, std::string>> self_collision_pairs;
  // not in the reference: where and how the batched engine runs
  int device = 0;
  bool speculate = true;            // hand whole lines, and several iterations / trials, to the GPU per call
  size_t rrt_lookahead = 12;        // RRT iterations drawn ahead per GPU call when speculating (plan in the 96-obstacle field: 4: 29 ms, 8: 19 ms, 12: 17.3 ms, 16: 17.3 ms, 32: 21 ms; tools/plan_time.py)
  size_t smoothing_lookahead = 16;  // smoothing trials evaluated per GPU call when speculating
  Rng rng{0};
  rrt::Stats stats;

  JointPathPlanner(Chain chain, CollisionChecker checker, N step, size_t max_try_, size_t num_smoothing_)
      : collision_check_robot(std::move(chain)), collision_checker(std::move(checker)), step_length(step), max_try(max_try_), num_smoothing(num_smoothing_) {
    if (const char* e = std::getenv("GEAR_HOST_RRT_LOOKAHEAD")) rrt_lookahead = (size_t)std::max(1, std::atoi(e));  // tuning aid
  }

  std::vector<int> all_joints() const { return collision_check_robot.movable_nodes(); }

  // Check if the joint_positions are OK (:71-84)
  bool is_feasible(const std::vector<int>& using_joints, const std::vector<N>& joint_positions, const Compound& objects) {
    if (!set_positions(using_joints, joint_positions)) return false;
    return !has_any_colliding(objects);
  }
  // :86-98
  bool has_any_colliding(const Compound& objects) {
    auto u = all_joints();
    std::vector<N> q = current(u);
    uint8_t f = 0;
    // no limit gate: the reference's has_any_colliding evaluates distances on whatever state the chain holds
    scene(u, &objects).check_batch(q.data(), 1, GH_CHECK_ENV | GH_CHECK_NO_LIMITS, &f);
    return (f & GH_FLAG_ENV) != 0;
  }
  // :100-110 — obstacle-major, per obstacle in chain order, duplicates kept
  std::vector<std::string> colliding_link_names(const Compound& objects) {
    auto u = all_joints();
    std::vector<N> q = current(u);
    Scene& sc = scene(u, &objects);
    auto m = sc.link_hits(q.data(), 1, GH_CHECK_NO_LIMITS);
    const int W = gh_link_mask_words(sc.handle());
    std::vector<std::string> names;
    for (int o = 0; o < sc.n_obstacles(); o++)
      for (int k = 0; k < sc.n_nodes(); k++)
        if ((m[(size_t)o * W + (k >> 6)] >> (k & 63)) & 1) names.push_back(collision_check_robot.nodes[k].name);
    return names;
  }
  // :113-121
  bool is_feasible_with_self(const std::vector<int>& using_joints, const std::vector<N>& joint_positions) {
    if (!set_positions(using_joints, joint_positions)) return false;
    return !has_any_colliding_with_self();
  }
  // :123-128
  bool has_any_colliding_with_self() {
    auto u = all_joints();
    std::vector<N> q = current(u);
    uint8_t f = 0;
    scene(u, nullptr).check_batch(q.data(), 1, GH_CHECK_SELF | GH_CHECK_NO_LIMITS, &f);
    return (f & GH_FLAG_SELF) != 0;
  }
  // :130-134
  std::vector<std::pair<std::string, std::string>> colliding_link_names_with_self() {
    auto u = all_joints();
    std::vector<N> q = current(u);
    Scene& sc = scene(u, nullptr);
    auto m = sc.self_hits(q.data(), 1, GH_CHECK_NO_LIMITS);
    std::vector<std::pair<std::string, std::string>> out;
    for (size_t p = 0; p < sc.pair_names().size(); p++)
      if ((m[p >> 6] >> (p & 63)) & 1) out.push_back(sc.pair_names()[p]);
    return out;
  }

  // batched forms: what the GPU is for.  The chain state is not touched.
  std::vector<uint8_t> is_feasible_batch(const std::vector<int>& using_joints, const std::vector<N>& q, const Compound* objects, bool with_self) {
    const size_t n = using_joints.empty() ? 0 : q.size() / using_joints.size();
    std::vector<uint8_t> f(n), ok(n);
    scene(using_joints, with_self ? nullptr : objects).check_batch(q.data(), n, with_self ? GH_CHECK_SELF : GH_CHECK_ENV, f.data());
    const uint8_t bad = (with_self ? GH_FLAG_SELF : GH_FLAG_ENV) | GH_FLAG_LIMIT;
    for (size_t i = 0; i < n; i++) ok[i] = (f[i] & bad) == 0;
    return ok;
  }
  std::vector<int32_t> first_infeasible(const std::vector<int>& using_joints, const std::vector<N>& q, const std::vector<uint32_t>& off, const Compound* objects, bool with_self) {
    std::vector<uint8_t> f(off.back());
    std::vector<int32_t> bad(off.size() - 1, -1);
    scene(using_joints, with_self ? nullptr : objects)
        .check_edges(q.data(), off.data(), off.size() - 1, with_self ? GH_CHECK_SELF : GH_CHECK_ENV,
                     (uint8_t)((with_self ? GH_FLAG_SELF : GH_FLAG_ENV) | GH_FLAG_LIMIT), f.data(), bad.data());
    return bad;
  }

  // A large batch of straight edges given by their end points (from, to: [n_edges][dof]); the samples rrt's stepping
  // would evaluate are generated on the device (gh_check_edges_from_endpoints).  mode GH_EDGE_CONNECT: Tree::connect
  // lines; GH_EDGE_WALK: smooth_path shortcuts.  Returns per edge {first infeasible sample or -1, sample count}.
  // For bulk validation (thousands of edges): plan() itself keeps host-side stepping, which is faster for a few lines.
  std::pair<std::vector<int32_t>, std::vector<uint32_t>> first_infeasible_from_endpoints(
      const std::vector<int>& using_joints, const std::vector<N>& from, const std::vector<N>& to, uint32_t mode, const Compound* objects,
      bool with_self, uint32_t max_len = 1u << 16) {
    const size_t n = using_joints.empty() ? 0 : from.size() / using_joints.size();
    if (to.size() != from.size()) throw Error(Error::DofMismatch, "from and to differ in length");
    std::vector<int32_t> bad(n, -1);
    std::vector<uint32_t> cnt(n, 0);
    scene(using_joints, with_self ? nullptr : objects)
        .check_edges_from_endpoints(from.data(), to.data(), n, (double)step_length, max_len, mode, with_self ? GH_CHECK_SELF : GH_CHECK_ENV,
                                    (uint8_t)((with_self ? GH_FLAG_SELF : GH_FLAG_ENV) | GH_FLAG_LIMIT), bad.data(), cnt.data());
    return {bad, cnt};
  }

  // Plan the sequence of joint angles of `using_joints` (:144-190)
  std::vector<std::vector<N>> plan(const std::vector<int>& using_joints, const std::vector<N>& start_angles,
                                   const std::vector<N>& goal_angles, const Compound& objects) {
    return plan_impl(using_joints, start_angles, goal_angles, &objects, false);
  }
  // Plan ... to avoid self collision (:198-243)
  std::vector<std::vector<N>> plan_avoid_self_collision(const std::vector<int>& using_joints, const std::vector<N>& start_angles,
                                                        const std::vector<N>& goal_angles) {
    return plan_impl(using_joints, start_angles, goal_angles, nullptr, true);
  }

  // The seeds RandomInitializeIKSolver would draw (src/ik.rs:77-78) — K of them, filtered on the GPU
  // by is_feasible before the host's Jacobian solver is tried on the survivors.  Returns the seeds
  // row-major and one feasibility byte each.  Not a reference function: the reference retries IK from
  // every seed and finds collisions only later, in plan().
  std::pair<std::vector<N>, std::vector<uint8_t>> feasible_ik_seeds(const std::vector<int>& using_joints, const std::vector<N>& initial_angles,
                                                                    size_t k, const Compound& objects) {
    const Limits limits = collision_check_robot.limits(using_joints);
    std::vector<N> seeds;
    for (size_t i = 0; i < k; i++) {
      std::vector<N> a = generate_random_joint_positions_from_limits<N>(limits, rng);
      modify_to_nearest_angle(initial_angles, a, limits);
      seeds.insert(seeds.end(), a.begin(), a.end());
    }
    return {seeds, is_feasible_batch(using_joints, seeds, &objects, false)};
  }

  std::vector<std::string> joint_names() const { return collision_check_robot.joint_names(); }

 private:
  // is_feasible folds every Err of set_joint_positions into `false` (:77-83), a length mismatch included
  bool set_positions(const std::vector<int>& using_joints, const std::vector<N>& q) {
    if (q.size() != using_joints.size()) return false;
    std::vector<double> d(q.begin(), q.end());
    return collision_check_robot.set_joint_positions(using_joints, d.data(), d.size());
  }
  std::vector<N> current(const std::vector<int>& u) const {
    std::vector<N> q;
    for (int i : u) q.push_back((N)collision_check_robot.nodes[i].position);
    return q;
  }
  // one gh_scene per (using_joints, obstacle compound, state of the joints the batch does not drive)
  Scene& scene(const std::vector<int>& using_joints, const Compound* objects) {
    std::vector<bool> driven(collision_check_robot.nodes.size(), false);
    for (int i : using_joints) {
      driven[i] = true;
      for (size_t k = 0; k < collision_check_robot.nodes.size(); k++)
        if (collision_check_robot.nodes[k].mimic_parent == i) driven[k] = true;
    }
    std::ostringstream key;
    key.precision(17);
    for (int i : using_joints) key << i << ',';
    key << '|' << (objects ? objects->id() : 0) << '|' << collision_checker.prediction << '|';  // Compound::id: contents, not address
    for (size_t k = 0; k < driven.size(); k++)
      if (!driven[k]) key << collision_check_robot.nodes[k].position << ',';
    for (auto& p : self_collision_pairs) key << p.first << ':' << p.second << ',';
    auto it = scenes_.find(key.str());
    if (it != scenes_.end()) return *it->second;
    if (scenes_.size() > 16) scenes_.clear();
    auto sc = std::make_unique<Scene>(collision_check_robot, collision_checker, using_joints, self_collision_pairs, objects,
                                      (int)sizeof(N) * 8, device);
    return *(scenes_[key.str()] = std::move(sc));
  }

  std::vector<std::vector<N>> plan_impl(const std::vector<int>& using_joints, const std::vector<N>& start_angles,
                                        const std::vector<N>& goal_angles, const Compound* objects, bool with_self) {
    const Limits limits = collision_check_robot.limits(using_joints);
    const std::vector<double> current_angles = collision_check_robot.joint_positions(using_joints);
    auto feasible = [&](const std::vector<N>& q) { return with_self ? is_feasible_with_self(using_joints, q) : is_feasible(using_joints, q, *objects); };
    auto restore = [&]() { collision_check_robot.set_joint_positions(using_joints, current_angles.data(), current_angles.size()); };
    auto collision = [&](CollisionPart part) {
      // the reference restores the joints (:156) BEFORE it collects the names for the error (:159), so
      // collision_link_names describes the restored state, not the rejected start or goal
      restore();
      Error e(with_self ? Error::SelfCollision : Error::Collision, with_self ? "Self Collision error" : "Collision error");
      e.part = part;
      if (with_self) e.collision_link_pairs = colliding_link_names_with_self();
      else e.collision_link_names = colliding_link_names(*objects);
      return e;
    };
    if (!feasible(start_angles)) throw collision(CollisionPart::Start);
    if (!feasible(goal_angles)) throw collision(CollisionPart::End);
    rrt::LineEval<N> eval = [&](const std::vector<N>& q, const std::vector<uint32_t>& off) {
      const auto t0 = std::chrono::steady_clock::now();
      auto bad = first_infeasible(using_joints, q, off, objects, with_self);
      stats.eval_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
      return bad;
    };
    std::vector<std::vector<N>> path;
    try {
      path = rrt::dual_rrt_connect<N>(start_angles, goal_angles, eval,
                                      [&](Rng& r) { return generate_random_joint_positions_from_limits<N>(limits, r); }, step_length,
                                      max_try, speculate ? std::max<size_t>(rrt_lookahead, 1) : 0, rng, stats);
    } catch (const Error&) {
      restore();
      throw;
    }
    rrt::smooth_path<N>(path, eval, step_length, num_smoothing, rng, speculate ? smoothing_lookahead : 1, stats);
    // the reference's closure leaves the chain at the last feasible sample it set; batched evaluation
    // does not move the chain, so it is left at the goal, the last configuration is_feasible accepted
    return path;
  }

  std::map<std::string, std::unique_ptr<Scene>> scenes_;
};

// Builder pattern to create `JointPathPlanner` (:258-365)
template <class N> class JointPathPlannerBuilder {
 public:
  JointPathPlannerBuilder(urdf::Robot robot, CollisionChecker checker)
      : robot_(std::make_shared<urdf::Robot>(std::move(robot))), checker_(std::move(checker)) {}
  static JointPathPlannerBuilder from_urdf_file(const std::string& path) {
    urdf::Robot r = urdf::read_file(path);
    CollisionChecker c = CollisionChecker::from_urdf_robot_with_base_dir(r, r.base_dir, 0.0);  // default margin 0.0 (:341)
    return JointPathPlannerBuilder(std::move(r), std::move(c));
  }
  static JointPathPlannerBuilder from_urdf_robot(urdf::Robot r) {
    CollisionChecker c = CollisionChecker::from_urdf_robot(r, 0.0);
    return JointPathPlannerBuilder(std::move(r), std::move(c));
  }
  JointPathPlannerBuilder& collision_check_margin(N length) { margin_ = length; has_margin_ = true; return *this; }
  JointPathPlannerBuilder& step_length(N v) { step_length_ = v; return *this; }
  JointPathPlannerBuilder& max_try(size_t v) { max_try_ = v; return *this; }
  JointPathPlannerBuilder& num_smoothing(size_t v) { num_smoothing_ = v; return *this; }
  JointPathPlannerBuilder& self_collision_pairs(std::vector<std::pair<std::string, std::string>> p) { pairs_ = std::move(p); return *this; }
  std::unique_ptr<JointPathPlanner<N>> finalize() {
    if (has_margin_) checker_.prediction = (double)margin_;
    auto p = std::make_unique<JointPathPlanner<N>>(Chain::from_urdf(*robot_), checker_, step_length_, max_try_, num_smoothing_);
    p->urdf_robot = robot_;
    p->self_collision_pairs = pairs_;
    return p;
  }

 private:
  std::shared_ptr<urdf::Robot> robot_;
  CollisionChecker checker_;
  N step_length_ = (N)0.1, margin_ = 0;
  bool has_margin_ = false;
  size_t max_try_ = 5000, num_smoothing_ = 100;
  std::vector<std::pair<std::string, std::string>> pairs_;
};

}  // namespace gear
