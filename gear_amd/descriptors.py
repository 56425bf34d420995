"""ctypes mirror of include/gear_hip.h and the flattening of a robot + obstacle
compound into those descriptors.

Flattening follows the reference's constructors:
  * CollisionChecker::from_urdf_robot_with_base_dir  src/collision/collision_checker.rs:234-262
  * urdf_geometry_to_shape_handle                    src/collision/urdf.rs:10-73
  * impl FromUrdf for Compound<f64>                  src/collision/collision_checker.rs:349-369
"""
from __future__ import annotations

import ctypes as C
import itertools
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

from . import urdf as U

GH_SHAPE_BOX, GH_SHAPE_BALL, GH_SHAPE_CYLINDER, GH_SHAPE_MESH = 0, 1, 2, 3
GH_FLAG_ENV, GH_FLAG_SELF, GH_FLAG_LIMIT = 1, 2, 4
GH_CHECK_ENV, GH_CHECK_SELF = 1, 2
GH_CHECK_NO_LIMITS = 0x10
GH_EDGE_CONNECT, GH_EDGE_WALK, GH_EDGE_TRUNCATED = 0, 1, 0x80000000   # gh_check_edges_from_endpoints: stepping rule, n_samples bit 31
GH_PATH_SMALL = 0x20
GH_PATH_FUSED, GH_PATH_PIPELINE, GH_PATH_NO_GRID, GH_COUNT_CANDIDATES = 0x1000, 0x2000, 0x4000, 0x8000   # evaluation-path selection bits of `what` (same results)


class gh_node_desc(C.Structure):
    _fields_ = [
        ("parent", C.c_int32), ("joint_type", C.c_int32),
        ("axis", C.c_double * 3), ("origin_t", C.c_double * 3), ("origin_q", C.c_double * 4),
        ("has_limit", C.c_int32), ("mimic_parent", C.c_int32),
        ("limit_lo", C.c_double), ("limit_hi", C.c_double),
        ("mimic_mult", C.c_double), ("mimic_offset", C.c_double),
        ("position", C.c_double),
    ]


class gh_shape_desc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("node", C.c_int32),
        ("params", C.c_double * 3), ("t", C.c_double * 3), ("q", C.c_double * 4),
        ("mesh", C.c_int32), ("reserved", C.c_int32),
    ]


class gh_mesh_desc(C.Structure):
    _fields_ = [
        ("n_vertices", C.c_int32), ("n_triangles", C.c_int32),
        ("vertices", C.POINTER(C.c_double)), ("triangles", C.POINTER(C.c_int32)),
    ]


class gh_robot_desc(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int32), ("n_cols", C.c_int32), ("n_shapes", C.c_int32),
        ("n_pairs", C.c_int32), ("n_meshes", C.c_int32), ("reserved", C.c_int32),
        ("nodes", C.POINTER(gh_node_desc)), ("col_node", C.POINTER(C.c_int32)),
        ("shapes", C.POINTER(gh_shape_desc)), ("pairs", C.POINTER(C.c_int32)),
        ("meshes", C.POINTER(gh_mesh_desc)),
    ]


class gh_obstacle_desc(C.Structure):
    _fields_ = [
        ("n_shapes", C.c_int32), ("n_meshes", C.c_int32),
        ("shapes", C.POINTER(gh_shape_desc)), ("meshes", C.POINTER(gh_mesh_desc)),
    ]


class gh_counters(C.Structure):
    _fields_ = [
        ("configs", C.c_uint64), ("pairs", C.c_uint64), ("cull_pass", C.c_uint64),
        ("sat_tests", C.c_uint64), ("exact_tests", C.c_uint64), ("exact_iters", C.c_uint64),
        ("reserved", C.c_uint64 * 2),
    ]


# ---------------------------------------------------------------------------
# host-side shape model
# ---------------------------------------------------------------------------
@dataclass
class ShapeSpec:
    """One (ShapeHandle, Isometry3) entry."""
    kind: int
    params: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    t: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    q: Tuple[float, float, float, float] = (0.0, 0.0, 0.0, 1.0)
    vertices: Optional[Sequence[Sequence[float]]] = None  # mesh only
    triangles: Optional[Sequence[Sequence[int]]] = None

    @staticmethod
    def box(half, t=(0, 0, 0), q=(0, 0, 0, 1)):
        return ShapeSpec(GH_SHAPE_BOX, tuple(float(x) for x in half), tuple(t), tuple(q))

    @staticmethod
    def ball(r, t=(0, 0, 0), q=(0, 0, 0, 1)):
        return ShapeSpec(GH_SHAPE_BALL, (float(r), 0.0, 0.0), tuple(t), tuple(q))

    @staticmethod
    def cylinder(r, length, t=(0, 0, 0), q=(0, 0, 0, 1)):
        return ShapeSpec(GH_SHAPE_CYLINDER, (float(r), float(length), 0.0), tuple(t), tuple(q))

    @staticmethod
    def mesh(vertices, triangles, t=(0, 0, 0), q=(0, 0, 0, 1)):
        return ShapeSpec(GH_SHAPE_MESH, (0.0, 0.0, 0.0), tuple(t), tuple(q), vertices, triangles)


def shape_from_urdf_geometry(g: U.Geometry, base_dir: Optional[str]) -> Optional[ShapeSpec]:
    """urdf_geometry_to_shape_handle (src/collision/urdf.rs:10-73), without the pose."""
    if g.kind == "box":
        return ShapeSpec(GH_SHAPE_BOX, (g.size[0] * 0.5, g.size[1] * 0.5, g.size[2] * 0.5))
    if g.kind == "cylinder":
        return ShapeSpec(GH_SHAPE_CYLINDER, (g.radius, g.length, 0.0))
    if g.kind == "sphere":
        return ShapeSpec(GH_SHAPE_BALL, (g.radius, 0.0, 0.0))
    if g.kind == "mesh":
        # expand_package_path: "package://" stays unresolved here; relative paths use base_dir.
        path = g.filename
        if not path.startswith("package://") and base_dir and not os.path.isabs(path):
            path = os.path.join(base_dir, path)
        if not os.path.exists(path):
            return None  # urdf.rs:60-63: error!(..) and None
        verts, tris = U.load_obj_mesh(path, g.scale)
        if not verts or not tris:
            return None  # urdf.rs:66-69: load_mesh failed, error!(..) and None
        return ShapeSpec(GH_SHAPE_MESH, (0.0, 0.0, 0.0), vertices=verts, triangles=tris)
    return None


def _collision_specs(link: U.Link, base_dir: Optional[str]) -> List[ShapeSpec]:
    out = []
    for col in link.collisions:
        s = shape_from_urdf_geometry(col.geometry, base_dir)
        if s is not None:
            s.t = tuple(col.xyz)
            s.q = U.quat_from_rpy(*col.rpy)
            out.append(s)
    return out


@dataclass
class CollisionModel:
    """gear::CollisionChecker's data: name_collision_model_map + prediction."""
    name_collision_model_map: Dict[str, List[ShapeSpec]]
    prediction: float = 0.0
    self_collision_pairs: List[Tuple[str, str]] = field(default_factory=list)

    @classmethod
    def from_urdf_robot(cls, robot: U.UrdfRobot, prediction: float) -> "CollisionModel":
        m: Dict[str, List[ShapeSpec]] = {}
        l2j = U.link_to_joint_map(robot)
        for link in robot.links:
            specs = _collision_specs(link, robot.base_dir)
            if specs and link.name in l2j:
                m[l2j[link.name]] = specs
        return cls(m, prediction)


_compound_ids = itertools.count(1)


@dataclass
class Compound:
    """ncollide3d::shape::Compound<f64> as gear builds it: a flat list of posed shapes.
    Treat it as immutable once built, as ncollide's is: `uid` (drawn at construction, never reused
    by another object, unlike id()) is what the planner keys its GPU scene cache on."""
    shapes: List[ShapeSpec]
    uid: int = field(default_factory=lambda: next(_compound_ids), compare=False)

    @classmethod
    def from_urdf_robot(cls, robot: U.UrdfRobot) -> "Compound":
        out: List[ShapeSpec] = []
        for link in robot.links:
            out.extend(_collision_specs(link, None))
        return cls(out)

    @classmethod
    def from_urdf_file(cls, path: str) -> "Compound":
        return cls.from_urdf_robot(U.read_urdf(path))


def parse_colon_separated_pairs(pair_strs: Sequence[str]) -> List[Tuple[str, str]]:
    """src/collision/collision_checker.rs:314-329."""
    pairs = []
    for s in pair_strs:
        sp = s.split(":")
        if len(sp) < 2:
            raise ValueError(f"failed to parse {s}")  # Error::ParseError
        pairs.append((sp[0], sp[1]))
    return pairs


# ---------------------------------------------------------------------------
# flattening into the C descriptors
# ---------------------------------------------------------------------------
class _Keep:
    """Holds a descriptor plus every ctypes buffer it points into."""

    def __init__(self, desc, refs):
        self.desc = desc
        self._refs = refs

    def byref(self):
        return C.byref(self.desc)


def _fill_shapes(specs: Sequence[Tuple[int, ShapeSpec]], refs: list):
    arr = (gh_shape_desc * max(len(specs), 1))()
    meshes = []
    for i, (node, s) in enumerate(specs):
        d = arr[i]
        d.kind, d.node = s.kind, node
        d.params[:] = list(s.params)
        d.t[:] = list(s.t)
        d.q[:] = list(s.q)
        d.mesh = -1
        if s.kind == GH_SHAPE_MESH:
            nv, nt = len(s.vertices), len(s.triangles)
            vb = (C.c_double * (3 * nv))(*[float(x) for v in s.vertices for x in v])
            tb = (C.c_int32 * (3 * nt))(*[int(x) for t in s.triangles for x in t])
            refs += [vb, tb]
            md = gh_mesh_desc(nv, nt, C.cast(vb, C.POINTER(C.c_double)), C.cast(tb, C.POINTER(C.c_int32)))
            d.mesh = len(meshes)
            meshes.append(md)
    marr = (gh_mesh_desc * max(len(meshes), 1))(*meshes)
    refs += [arr, marr]
    return arr, marr, len(meshes)


def flatten_robot(chain: U.Chain, model: CollisionModel, using_joints: Sequence[int],
                  self_pairs: Optional[Sequence[Tuple[str, str]]] = None) -> _Keep:
    """Robot side of gh_scene_create.

    using_joints: node indices of the movable joints the batch columns drive, in
    iter_joints() order (whole chain: chain.movable_nodes(); arm:
    chain.serial_chain_from_end(end)).
    self_pairs: joint-name pairs; pairs whose model or node is missing are
    dropped, as SelfCollisionPairs::next skips them
    (collision_checker.rs:150-169).
    """
    refs: list = []
    nodes = (gh_node_desc * len(chain.nodes))()
    for i, n in enumerate(chain.nodes):
        d = nodes[i]
        d.parent, d.joint_type = n.parent, n.joint_type
        d.axis[:] = list(n.axis)
        d.origin_t[:] = list(n.origin_t)
        d.origin_q[:] = list(n.origin_q)
        d.has_limit = 1 if n.limits is not None else 0
        d.limit_lo, d.limit_hi = n.limits if n.limits is not None else (0.0, 0.0)
        d.mimic_parent = n.mimic_parent
        d.mimic_mult, d.mimic_offset = n.mimic_multiplier, n.mimic_offset
        d.position = n.position
    cols = (C.c_int32 * max(len(using_joints), 1))(*using_joints)
    specs = []
    for i, n in enumerate(chain.nodes):  # EnvCollisionNames walks nodes in chain order
        for s in model.name_collision_model_map.get(n.name, []):
            specs.append((i, s))
    shapes, meshes, n_meshes = _fill_shapes(specs, refs)
    pair_idx: List[int] = []
    for (a, b) in (self_pairs if self_pairs is not None else model.self_collision_pairs):
        if a not in model.name_collision_model_map or b not in model.name_collision_model_map:
            continue
        na, nb = chain.find(a), chain.find(b)
        if na is None or nb is None:
            continue
        pair_idx += [na, nb]
    pairs = (C.c_int32 * max(len(pair_idx), 1))(*pair_idx)
    refs += [nodes, cols, pairs]
    desc = gh_robot_desc(len(chain.nodes), len(using_joints), len(specs), len(pair_idx) // 2, n_meshes, 0,
                         C.cast(nodes, C.POINTER(gh_node_desc)), C.cast(cols, C.POINTER(C.c_int32)),
                         C.cast(shapes, C.POINTER(gh_shape_desc)), C.cast(pairs, C.POINTER(C.c_int32)),
                         C.cast(meshes, C.POINTER(gh_mesh_desc)))
    keep = _Keep(desc, refs)
    keep.pair_names = [(chain.nodes[pair_idx[2 * k]].name, chain.nodes[pair_idx[2 * k + 1]].name)
                       for k in range(len(pair_idx) // 2)]
    return keep


def flatten_obstacles(objects: Compound) -> _Keep:
    refs: list = []
    shapes, meshes, n_meshes = _fill_shapes([(-1, s) for s in objects.shapes], refs)
    desc = gh_obstacle_desc(len(objects.shapes), n_meshes, C.cast(shapes, C.POINTER(gh_shape_desc)),
                            C.cast(meshes, C.POINTER(gh_mesh_desc)))
    return _Keep(desc, refs)
