"""URDF subset reader and k::Chain construction (host side, no GPU work).

Mirrors what gear gets from `urdf_rs::read_file` + `k::Chain::from(&robot)`:
  * joints / links / <collision> geometry as read by urdf-rs 0.4
    (reference call sites: src/collision/collision_checker.rs:234-262,349-369,
    src/planner/joint_path_planner.rs:336-348);
  * the chain's node order: a synthetic fixed joint "root", then depth-first with
    a stack, so children are visited in REVERSE insertion order — the order the
    reference's only known-answer test pins (joint_path_planner.rs:383-418:
    l_gripper_linear2 before l_gripper_linear1, l_tool_fixed last).

The `k` / `urdf-rs` crates are not vendored in the reference; their behaviour is
restated from their published sources ("Oracle").
"""
from __future__ import annotations

import math
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

ROOT_JOINT_NAME = "root"

FIXED, ROTATIONAL, LINEAR = 0, 1, 2


def quat_from_rpy(r: float, p: float, y: float) -> Tuple[float, float, float, float]:
    """nalgebra UnitQuaternion::from_euler_angles(roll, pitch, yaw) -> (i, j, k, w), in f64
    (k::urdf::quaternion_from computes in f64, then converts)."""
    sr, cr = math.sin(r * 0.5), math.cos(r * 0.5)
    sp, cp = math.sin(p * 0.5), math.cos(p * 0.5)
    sy, cy = math.sin(y * 0.5), math.cos(y * 0.5)
    return (
        sr * cp * cy - cr * sp * sy,
        cr * sp * cy + sr * cp * sy,
        cr * cp * sy - sr * sp * cy,
        cr * cp * cy + sr * sp * sy,
    )


@dataclass
class Geometry:
    kind: str  # "box" | "cylinder" | "sphere" | "mesh"
    size: Tuple[float, float, float] = (0.0, 0.0, 0.0)  # box
    radius: float = 0.0
    length: float = 0.0
    filename: str = ""
    scale: Tuple[float, float, float] = (1.0, 1.0, 1.0)


@dataclass
class Collision:
    xyz: Tuple[float, float, float]
    rpy: Tuple[float, float, float]
    geometry: Geometry


@dataclass
class Link:
    name: str
    collisions: List[Collision] = field(default_factory=list)


@dataclass
class Joint:
    name: str
    type: str
    xyz: Tuple[float, float, float]
    rpy: Tuple[float, float, float]
    parent: str
    child: str
    axis: Tuple[float, float, float]
    lower: float
    upper: float
    mimic_joint: str = ""
    mimic_multiplier: float = 1.0
    mimic_offset: float = 0.0


@dataclass
class UrdfRobot:
    name: str
    links: List[Link]
    joints: List[Joint]
    base_dir: Optional[str] = None


def _floats(text: Optional[str], n: int, default: Sequence[float]) -> Tuple[float, ...]:
    if text is None:
        return tuple(default)
    vals = [float(v) for v in text.split()]
    if len(vals) != n:
        raise ValueError(f"expected {n} numbers, got {text!r}")
    return tuple(vals)


def _origin(elem: Optional[ET.Element]):
    if elem is None:
        return (0.0, 0.0, 0.0), (0.0, 0.0, 0.0)
    return _floats(elem.get("xyz"), 3, (0, 0, 0)), _floats(elem.get("rpy"), 3, (0, 0, 0))


def _geometry(elem: ET.Element) -> Geometry:
    for child in elem:
        if child.tag == "box":
            return Geometry("box", size=_floats(child.get("size"), 3, (0, 0, 0)))
        if child.tag == "cylinder":
            return Geometry("cylinder", radius=float(child.get("radius")), length=float(child.get("length")))
        if child.tag == "sphere":
            return Geometry("sphere", radius=float(child.get("radius")))
        if child.tag == "mesh":
            return Geometry("mesh", filename=child.get("filename", ""),
                            scale=_floats(child.get("scale"), 3, (1, 1, 1)))
    raise ValueError("unsupported <geometry>")


def parse_urdf(text: str, base_dir: Optional[str] = None) -> UrdfRobot:
    root = ET.fromstring(text)
    if root.tag != "robot":
        raise ValueError("not a URDF <robot>")
    links, joints = [], []
    for le in root.findall("link"):
        link = Link(le.get("name"))
        for ce in le.findall("collision"):
            xyz, rpy = _origin(ce.find("origin"))
            link.collisions.append(Collision(xyz, rpy, _geometry(ce.find("geometry"))))
        links.append(link)
    for je in root.findall("joint"):
        xyz, rpy = _origin(je.find("origin"))
        axis_e, limit_e, mimic_e = je.find("axis"), je.find("limit"), je.find("mimic")
        axis = _floats(axis_e.get("xyz") if axis_e is not None else None, 3, (1, 0, 0))
        lower = float(limit_e.get("lower", 0.0)) if limit_e is not None else 0.0
        upper = float(limit_e.get("upper", 0.0)) if limit_e is not None else 0.0
        j = Joint(je.get("name"), je.get("type"), xyz, rpy, je.find("parent").get("link"),
                  je.find("child").get("link"), axis, lower, upper)
        if mimic_e is not None:
            j.mimic_joint = mimic_e.get("joint", "")
            j.mimic_multiplier = float(mimic_e.get("multiplier", 1.0))
            j.mimic_offset = float(mimic_e.get("offset", 0.0))
        joints.append(j)
    return UrdfRobot(root.get("name", ""), links, joints, base_dir)


def read_urdf(path: str) -> UrdfRobot:
    with open(path, "r") as fh:
        return parse_urdf(fh.read(), os.path.dirname(os.path.abspath(path)))


# ---------------------------------------------------------------------------
# k::Chain
# ---------------------------------------------------------------------------
@dataclass
class Node:
    """One k::Node: a joint plus its place in the tree."""
    name: str
    joint_type: int
    axis: Tuple[float, float, float]
    origin_t: Tuple[float, float, float]
    origin_q: Tuple[float, float, float, float]
    limits: Optional[Tuple[float, float]]
    parent: int = -1
    children: List[int] = field(default_factory=list)
    mimic_parent: int = -1
    mimic_multiplier: float = 1.0
    mimic_offset: float = 0.0
    position: float = 0.0

    @property
    def is_movable(self) -> bool:
        return self.joint_type != FIXED


class Chain:
    """k::Chain: nodes in depth-first order from the synthetic root joint."""

    def __init__(self, nodes: List[Node]):
        self.nodes = nodes

    @classmethod
    def from_urdf(cls, robot: UrdfRobot) -> "Chain":
        # k::urdf: one node per joint, parent/child wiring through link names,
        # parentless joints hang under the synthetic "root" node.
        tmp: List[Node] = [Node(ROOT_JOINT_NAME, FIXED, (1.0, 0.0, 0.0), (0.0, 0.0, 0.0),
                                (0.0, 0.0, 0.0, 1.0), None)]
        child_link_to_node: Dict[str, int] = {}
        parent_link_to_nodes: Dict[str, List[int]] = {}
        joint_name_to_node: Dict[str, int] = {}
        for j in robot.joints:
            if j.type in ("revolute", "continuous"):
                jt = ROTATIONAL
            elif j.type == "prismatic":
                jt = LINEAR
            else:
                jt = FIXED
            limits = None if (j.upper - j.lower) == 0.0 else (j.lower, j.upper)
            idx = len(tmp)
            tmp.append(Node(j.name, jt, tuple(j.axis), tuple(j.xyz), quat_from_rpy(*j.rpy), limits))
            child_link_to_node[j.child] = idx
            parent_link_to_nodes.setdefault(j.parent, []).append(idx)
            joint_name_to_node[j.name] = idx
        for link in robot.links:
            pn = child_link_to_node.get(link.name)
            if pn is not None:
                for cn in parent_link_to_nodes.get(link.name, []):
                    tmp[cn].parent = pn
                    tmp[pn].children.append(cn)
        for idx in range(1, len(tmp)):
            if tmp[idx].parent < 0:
                tmp[idx].parent = 0
                tmp[0].children.append(idx)
        for j in robot.joints:
            if j.mimic_joint:
                c = joint_name_to_node[j.name]
                tmp[c].mimic_parent = joint_name_to_node[j.mimic_joint]
                tmp[c].mimic_multiplier = j.mimic_multiplier
                tmp[c].mimic_offset = j.mimic_offset
        # Descendants iterator: pop from a stack, push children in insertion order.
        order: List[int] = []
        stack = [0]
        while stack:
            n = stack.pop()
            order.append(n)
            stack.extend(tmp[n].children)
        remap = {old: new for new, old in enumerate(order)}
        nodes = []
        for old in order:
            n = tmp[old]
            nodes.append(Node(n.name, n.joint_type, n.axis, n.origin_t, n.origin_q, n.limits,
                              remap[n.parent] if n.parent >= 0 else -1,
                              [remap[c] for c in n.children],
                              remap[n.mimic_parent] if n.mimic_parent >= 0 else -1,
                              n.mimic_multiplier, n.mimic_offset, 0.0))
        return cls(nodes)

    def find(self, joint_name: str) -> Optional[int]:
        """k::Chain::find: lookup by JOINT name."""
        for i, n in enumerate(self.nodes):
            if n.name == joint_name:
                return i
        return None

    def movable_nodes(self) -> List[int]:
        """iter_joints(): movable joints in chain order."""
        return [i for i, n in enumerate(self.nodes) if n.is_movable]

    def dof(self) -> int:
        return len(self.movable_nodes())

    def serial_chain_from_end(self, end: int) -> List[int]:
        """k::SerialChain::from_end(node).iter_joints(): movable joints root -> end."""
        path = []
        n = end
        while n >= 0:
            path.append(n)
            n = self.nodes[n].parent
        path.reverse()
        return [i for i in path if self.nodes[i].is_movable]

    def joint_names(self) -> List[str]:
        return [self.nodes[i].name for i in self.movable_nodes()]


def link_to_joint_map(robot: UrdfRobot) -> Dict[str, str]:
    """k::urdf::link_to_joint_map: link name -> name of the joint whose child it is;
    links that are nobody's child map to "root"."""
    m = {j.child: j.name for j in robot.joints}
    for link in robot.links:
        m.setdefault(link.name, ROOT_JOINT_NAME)
    return m


def load_obj_mesh(path: str, scale=(1.0, 1.0, 1.0)):
    """Minimal Wavefront-OBJ reader standing in for src/collision/mesh.rs:39-69 (assimp): triangles only
    (faces of another size are dropped, :55-65), all sub-meshes merged (:66), coordinates held as f32 by
    assimp and scaled per axis in f64 (`v.x as f64 * scale[0]`, :49-53)."""
    import numpy as np
    f32 = lambda t: float(np.float32(float(t)))
    verts, tris = [], []
    with open(path, "r") as fh:
        for line in fh:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                verts.append((f32(p[1]) * scale[0], f32(p[2]) * scale[1], f32(p[3]) * scale[2]))
            elif p[0] == "f":
                idx = [int(tok.split("/")[0]) for tok in p[1:]]
                if len(idx) == 3:  # mesh.rs keeps triangles only
                    tris.append(tuple(i - 1 if i > 0 else len(verts) + i for i in idx))
    return verts, tris


# ---------------------------------------------------------------------------
# JSON form of a parsed robot (tests/golden fixtures are stored this way)
# ---------------------------------------------------------------------------
def robot_to_dict(robot: UrdfRobot) -> dict:
    import dataclasses
    return {"name": robot.name,
            "links": [dataclasses.asdict(l) for l in robot.links],
            "joints": [dataclasses.asdict(j) for j in robot.joints]}


def robot_from_dict(d: dict, base_dir: Optional[str] = None) -> UrdfRobot:
    links = []
    for l in d["links"]:
        cols = []
        for c in l["collisions"]:
            g = c["geometry"]
            cols.append(Collision(tuple(c["xyz"]), tuple(c["rpy"]),
                                  Geometry(g["kind"], tuple(g["size"]), g["radius"], g["length"],
                                           g["filename"], tuple(g["scale"]))))
        links.append(Link(l["name"], cols))
    joints = []
    for j in d["joints"]:
        joints.append(Joint(j["name"], j["type"], tuple(j["xyz"]), tuple(j["rpy"]), j["parent"], j["child"],
                            tuple(j["axis"]), j["lower"], j["upper"], j["mimic_joint"],
                            j["mimic_multiplier"], j["mimic_offset"]))
    return UrdfRobot(d["name"], links, joints, base_dir)


def robot_to_xml(robot: UrdfRobot) -> str:
    """Serialise a parsed robot back to URDF text (17 significant digits), e.g. to hand a JSON
    fixture to the C++ host layer, which reads URDF files."""
    def f(v):
        return " ".join(repr(float(x)) for x in v)
    out = [f'<?xml version="1.0"?>', f'<robot name="{robot.name}">']
    for l in robot.links:
        out.append(f'  <link name="{l.name}">')
        for c in l.collisions:
            g = c.geometry
            if g.kind == "box":
                ge = f'<box size="{f(g.size)}"/>'
            elif g.kind == "cylinder":
                ge = f'<cylinder radius="{float(g.radius)!r}" length="{float(g.length)!r}"/>'
            elif g.kind == "sphere":
                ge = f'<sphere radius="{float(g.radius)!r}"/>'
            else:
                ge = f'<mesh filename="{g.filename}" scale="{f(g.scale)}"/>'
            out.append(f'    <collision><origin xyz="{f(c.xyz)}" rpy="{f(c.rpy)}"/><geometry>{ge}</geometry></collision>')
        out.append('  </link>')
    for j in robot.joints:
        out.append(f'  <joint name="{j.name}" type="{j.type}">')
        out.append(f'    <origin xyz="{f(j.xyz)}" rpy="{f(j.rpy)}"/><parent link="{j.parent}"/><child link="{j.child}"/>')
        out.append(f'    <axis xyz="{f(j.axis)}"/><limit lower="{float(j.lower)!r}" upper="{float(j.upper)!r}"/>')
        if j.mimic_joint:
            out.append(f'    <mimic joint="{j.mimic_joint}" multiplier="{float(j.mimic_multiplier)!r}" '
                       f'offset="{float(j.mimic_offset)!r}"/>')
        out.append('  </joint>')
    out.append('</robot>')
    return "\n".join(out) + "\n"
