"""Seeded synthetic workloads for the hot path.

UR10 / Sawyer / PR2 descriptions are not in the reference (its README finds them
with `rospack`), so the robots here are builder-authored stand-ins with the same
DOF and link-shape mix; only the sample.urdf scene is reference-pinned.
Obstacle fields use boxes and balls — the two shapes gear's
`Compound<f64>::from_urdf_robot` can hold (a TriMesh sub-shape makes
ncollide's Compound::new panic); cylinders can be mixed in on request and are
then "reference-undefined" (DESIGN.md).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import descriptors as D
from . import urdf as U


def _link(name, cols):
    return U.Link(name, [U.Collision(tuple(xyz), tuple(rpy), g) for (xyz, rpy, g) in cols])


def _box(sx, sy, sz):
    return U.Geometry("box", size=(sx, sy, sz))


def _cyl(r, l):
    return U.Geometry("cylinder", radius=r, length=l)


def _ball(r):
    return U.Geometry("sphere", radius=r)


def _rev(name, parent, child, xyz, axis, lo, hi, rpy=(0.0, 0.0, 0.0)):
    return U.Joint(name, "revolute", tuple(xyz), tuple(rpy), parent, child, tuple(axis), lo, hi)


def arm7_robot() -> U.UrdfRobot:
    """A Sawyer-like 7-DOF serial arm: 9 nodes (root, 7 revolute, 1 fixed tool),
    9 collision shapes (5 boxes, 3 cylinders, 1 ball)."""
    hp = math.pi / 2
    links = [
        _link("base", [((0, 0, 0.08), (0, 0, 0), _box(0.2, 0.2, 0.16))]),
        _link("l0", [((0, 0, 0.12), (0, 0, 0), _cyl(0.07, 0.24))]),
        _link("l1", [((0.2, 0, 0), (0, 0, 0), _box(0.4, 0.1, 0.1))]),
        _link("l2", [((0.08, 0, 0), (0, hp, 0), _cyl(0.055, 0.16))]),
        _link("l3", [((0.18, 0, 0), (0, 0, 0), _box(0.36, 0.09, 0.09))]),
        _link("l4", [((0.07, 0, 0), (0, hp, 0), _cyl(0.05, 0.14))]),
        _link("l5", [((0.1, 0, 0), (0, 0, 0), _box(0.2, 0.08, 0.08))]),
        _link("l6", [((0.05, 0, 0), (0, 0, 0), _ball(0.06)), ((0.12, 0, 0), (0, 0, 0), _box(0.08, 0.12, 0.04))]),
        _link("tool", []),
    ]
    joints = [
        _rev("j0", "base", "l0", (0, 0, 0.16), (0, 0, 1), -3.05, 3.05),
        _rev("j1", "l0", "l1", (0.08, 0, 0.24), (0, 1, 0), -3.8, 2.2),
        _rev("j2", "l1", "l2", (0.4, 0, 0), (1, 0, 0), -3.04, 3.04),
        _rev("j3", "l2", "l3", (0.16, 0, 0), (0, 1, 0), -3.04, 3.04),
        _rev("j4", "l3", "l4", (0.36, 0, 0), (1, 0, 0), -2.97, 2.97),
        _rev("j5", "l4", "l5", (0.14, 0, 0), (0, 1, 0), -2.97, 2.97),
        _rev("j6", "l5", "l6", (0.2, 0, 0), (1, 0, 0), -4.7, 4.7),
        U.Joint("tool_fixed", "fixed", (0.16, 0, 0), (0, 0, 0), "l6", "tool", (1, 0, 0), 0.0, 0.0),
    ]
    return U.UrdfRobot("arm7", links, joints)


ARM7_SELF_PAIRS = [
    ("root", "j3"), ("root", "j4"), ("root", "j5"), ("root", "j6"),
    ("j0", "j3"), ("j0", "j4"), ("j0", "j5"), ("j0", "j6"),
    ("j1", "j4"), ("j1", "j5"), ("j1", "j6"), ("j2", "j5"), ("j2", "j6"), ("j3", "j6"),
]


def arm6_robot() -> U.UrdfRobot:
    """A UR10-like 6-DOF arm: boxes and cylinders, limits +-2pi."""
    hp = math.pi / 2
    tp = 2 * math.pi
    links = [
        _link("base", [((0, 0, 0.05), (0, 0, 0), _cyl(0.09, 0.1))]),
        _link("shoulder", [((0, 0, 0.06), (0, 0, 0), _cyl(0.08, 0.18))]),
        _link("upper", [((0.3, 0, 0), (0, 0, 0), _box(0.6, 0.12, 0.12))]),
        _link("fore", [((0.28, 0, 0), (0, hp, 0), _cyl(0.05, 0.56))]),
        _link("w1", [((0, 0, 0), (0, 0, 0), _box(0.1, 0.1, 0.12))]),
        _link("w2", [((0, 0, 0), (0, 0, 0), _box(0.1, 0.12, 0.1))]),
        _link("w3", [((0.04, 0, 0), (0, hp, 0), _cyl(0.045, 0.08))]),
    ]
    joints = [
        _rev("shoulder_pan", "base", "shoulder", (0, 0, 0.1), (0, 0, 1), -tp, tp),
        _rev("shoulder_lift", "shoulder", "upper", (0, 0.14, 0.08), (0, 1, 0), -tp, tp),
        _rev("elbow", "upper", "fore", (0.61, -0.12, 0), (0, 1, 0), -tp, tp),
        _rev("wrist_1", "fore", "w1", (0.57, 0, 0), (0, 1, 0), -tp, tp),
        _rev("wrist_2", "w1", "w2", (0, 0.11, 0), (0, 0, 1), -tp, tp),
        _rev("wrist_3", "w2", "w3", (0, 0, 0.11), (1, 0, 0), -tp, tp),
    ]
    return U.UrdfRobot("arm6", links, joints)


ARM6_SELF_PAIRS = [("root", "wrist_1"), ("root", "wrist_2"), ("root", "wrist_3"), ("shoulder_pan", "wrist_1"),
                   ("shoulder_pan", "wrist_2"), ("shoulder_pan", "wrist_3"), ("shoulder_lift", "wrist_2"),
                   ("shoulder_lift", "wrist_3")]


def _rand_quat(rng) -> Tuple[float, float, float, float]:
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    return tuple(float(x) for x in q)


def obstacle_field(n: int, seed: int = 0, kinds: Sequence[str] = ("box", "ball"),
                   weights: Optional[Sequence[float]] = None, cube: float = 2.4,
                   base=(0.0, 0.0, 0.2), keep_out: float = 0.35) -> D.Compound:
    """n obstacles with centres uniform in a cube of side `cube` (z from -0.2), none within
    keep_out (+ its own bounding radius) of the robot base, random rotations."""
    rng = np.random.Generator(np.random.Philox(seed))
    weights = weights if weights is not None else [1.0 / len(kinds)] * len(kinds)
    shapes: List[D.ShapeSpec] = []
    base = np.asarray(base, float)
    while len(shapes) < n:
        kind = kinds[int(rng.choice(len(kinds), p=weights))]
        c = np.array([rng.uniform(-cube / 2, cube / 2), rng.uniform(-cube / 2, cube / 2),
                      rng.uniform(-0.2, cube - 0.2)])
        q = _rand_quat(rng)
        if kind == "box":
            h = rng.uniform(0.02, 0.15, 3)
            br, spec = float(np.linalg.norm(h)), D.ShapeSpec.box(h, tuple(c), q)
        elif kind == "ball":
            r = float(rng.uniform(0.03, 0.12))
            br, spec = r, D.ShapeSpec.ball(r, tuple(c), q)
        else:
            r, l = float(rng.uniform(0.02, 0.1)), float(rng.uniform(0.05, 0.4))
            br, spec = math.hypot(r, l / 2), D.ShapeSpec.cylinder(r, l, tuple(c), q)
        if np.linalg.norm(c - base) < keep_out + br:
            continue
        shapes.append(spec)
    return D.Compound(shapes)


def joint_limits(chain: U.Chain, using: Sequence[int]):
    lo = np.array([chain.nodes[i].limits[0] if chain.nodes[i].limits else -math.pi for i in using])
    hi = np.array([chain.nodes[i].limits[1] if chain.nodes[i].limits else math.pi for i in using])
    return lo, hi


def random_configs(chain: U.Chain, using: Sequence[int], n: int, seed: int = 0, dtype=np.float32,
                   overshoot: float = 0.0) -> np.ndarray:
    """generate_random_joint_positions_from_limits (src/funcs.rs:76-87): (max - min) * u + min per joint,
    u in [0,1); no limit -> (u - 0.5) * 2 pi.  overshoot > 0 widens the range so that some
    configurations violate their limits (for exercising GH_FLAG_LIMIT)."""
    rng = np.random.Generator(np.random.Philox(seed))
    lo, hi = joint_limits(chain, using)
    span = hi - lo
    lo2, span2 = lo - overshoot * span, span * (1 + 2 * overshoot)
    u = rng.random((n, len(using)))
    return (span2 * u + lo2).astype(dtype)


def straight_edges_lockstep(chain: U.Chain, using: Sequence[int], n_edges: int, seed: int = 0, step: float = 0.1,
                            min_len: int = 8, max_len: int = 64, dtype=np.float32):
    """straight_edges for large edge counts: the same recurrence, advanced for all edges in lockstep
    (one numpy step per sample index instead of one Python iteration per sample)."""
    rng = np.random.Generator(np.random.Philox(seed))
    lo, hi = joint_limits(chain, using)
    d = len(using)
    a = ((hi - lo) * rng.random((n_edges, d)) + lo).astype(dtype)
    b = ((hi - lo) * rng.random((n_edges, d)) + lo).astype(dtype)
    want = rng.integers(min_len, max_len + 1, size=n_edges)
    stp = dtype(step)
    base = a.copy()
    alive = np.ones(n_edges, dtype=bool)
    count = np.zeros(n_edges, dtype=np.int64)
    rows = []          # (edge ids, samples) per step
    for _ in range(max_len):
        ids = np.nonzero(alive)[0]
        if len(ids) == 0:
            break
        diff = b[ids] - base[ids]
        dist = np.sqrt(np.sum(diff * diff, axis=1, dtype=dtype)).astype(dtype)
        near = dist < stp
        nxt = np.where(near[:, None], b[ids], base[ids] + diff * (stp / np.where(near, dtype(1), dist))[:, None]).astype(dtype)
        base[ids] = nxt
        rows.append((ids, nxt))
        count[ids] += 1
        alive[ids] = ~near & (count[ids] < want[ids])
    off = np.concatenate([[0], np.cumsum(count)]).astype(np.uint32)
    q = np.empty((int(off[-1]), d), dtype=dtype)
    for k, (ids, smp) in enumerate(rows):
        q[off[ids] + k] = smp
    return q, off


def straight_edges(chain: U.Chain, using: Sequence[int], n_edges: int, seed: int = 0, step: float = 0.1,
                   min_len: int = 8, max_len: int = 64, dtype=np.float32):
    """Edge samples as rrt's stepping produces them: from a random start, repeatedly
    base += (target - base) * step / dist until the target is reached, computed with the iterated
    recurrence in the batch dtype (not the closed form), truncated to [min_len, max_len] samples."""
    rng = np.random.Generator(np.random.Philox(seed))
    lo, hi = joint_limits(chain, using)
    qs, off = [], [0]
    stp = dtype(step)
    for _ in range(n_edges):
        a = ((hi - lo) * rng.random(len(using)) + lo).astype(dtype)
        b = ((hi - lo) * rng.random(len(using)) + lo).astype(dtype)
        want = int(rng.integers(min_len, max_len + 1))
        base = a.copy()
        samples = []
        while len(samples) < want:
            diff = b - base
            dist = dtype(np.sqrt(np.sum(diff * diff, dtype=dtype)))
            if dist < stp:
                samples.append(b.copy())
                break
            base = base + diff * (stp / dist)
            samples.append(base.copy())
        qs.extend(samples)
        off.append(len(qs))
    return np.asarray(qs, dtype=dtype), np.asarray(off, np.uint32)


def edge_samples_lockstep(a: np.ndarray, b: np.ndarray, step: float, mode: int = D.GH_EDGE_CONNECT, max_len: int = 1 << 20):
    """The samples rrt 0.4.0's stepping evaluates on the edges a[e] -> b[e], all edges advanced in
    lockstep with numpy in the arrays' dtype: the iterated recurrence  d = sqrt(sum_i (cur_i - b_i)^2) (summed in index
    order), next = cur + (b - cur) * step / d.  mode GH_EDGE_CONNECT: Tree::connect (a step that would overshoot lands on
    b; the line ends with the first sample within `step` of b); GH_EDGE_WALK: smooth_path's walker (advance while at least
    `step` away; b itself is not a sample).  Returns (q [n][D], off [E + 1]); an edge is cut after max_len samples."""
    dtype = a.dtype.type
    n_edges, dim = a.shape
    stp = dtype(step)

    def dist(x, y):
        s = np.zeros(len(x), dtype)
        for c in range(dim):                      # in index order, one rounding per term: what the scalar loops do
            t = x[:, c] - y[:, c]
            s = s + t * t
        return np.sqrt(s)

    cur = a.copy()
    alive = np.ones(n_edges, bool)
    count = np.zeros(n_edges, np.int64)
    rows = []
    while alive.any():
        ids = np.nonzero(alive)[0]
        d = dist(cur[ids], b[ids])
        near = d < stp
        if mode == D.GH_EDGE_WALK:
            alive[ids[near]] = False
            ids, d, near = ids[~near], d[~near], near[~near]
            if len(ids) == 0:
                break
        with np.errstate(divide="ignore", invalid="ignore"):
            nxt = cur[ids] + (b[ids] - cur[ids]) * stp / d[:, None]
        nxt = np.where(near[:, None], b[ids], nxt).astype(dtype)
        rows.append((ids, count[ids].copy(), nxt))
        cur[ids] = nxt
        count[ids] += 1
        if mode == D.GH_EDGE_CONNECT:
            done = near | (dist(nxt, b[ids]) < stp)
            alive[ids[done]] = False
        alive[ids[count[ids] >= max_len]] = False
    off = np.concatenate([[0], np.cumsum(count)]).astype(np.uint32)
    q = np.empty((int(off[-1]), dim), dtype)
    for ids, k, smp in rows:
        q[off[ids] + k] = smp
    return q, off


def convex_mesh(rng, scale, n_min=8, n_max=24):
    """A random closed convex triangle mesh (<= 64 vertices): the hull of points on a squashed sphere."""
    from scipy.spatial import ConvexHull
    pts = rng.normal(size=(int(rng.integers(n_min, n_max + 1)), 3))
    pts *= np.asarray(scale) / np.linalg.norm(pts, axis=1, keepdims=True)
    hull = ConvexHull(pts)
    used = np.unique(hull.simplices)
    remap = {int(o): i for i, o in enumerate(used)}
    return pts[used].tolist(), [[remap[int(i)] for i in s] for s in hull.simplices]


def dual_arm_robot() -> U.UrdfRobot:
    """A PR2-like upper body: base, a prismatic torso lift, and two 7-DOF arms branching off the torso
    (17 nodes with the root, 15 movable joints).  Link geometry here is boxes; `pr2_workload` replaces
    most of it by convex meshes."""
    links = [_link("base", [((0, 0, 0.15), (0, 0, 0), _box(0.6, 0.6, 0.3))]),
             _link("torso", [((-0.05, 0, 0.35), (0, 0, 0), _box(0.3, 0.4, 0.7))])]
    joints = [U.Joint("torso_lift", "prismatic", (0, 0, 0.3), (0, 0, 0), "base", "torso", (0, 0, 1), 0.0, 0.3)]
    for side, sy in (("l", 1.0), ("r", -1.0)):
        names = [f"{side}_{n}" for n in ("shoulder_pan", "shoulder_lift", "upper_arm", "elbow", "forearm", "wrist_flex",
                                         "gripper")]
        sizes = [(0.16, 0.16, 0.2), (0.14, 0.14, 0.14), (0.4, 0.12, 0.12), (0.12, 0.12, 0.12), (0.32, 0.1, 0.1),
                 (0.1, 0.1, 0.1), (0.14, 0.12, 0.06)]
        offs = [(0, 0, 0), (0, 0, 0), (0.2, 0, 0), (0, 0, 0), (0.16, 0, 0), (0, 0, 0), (0.08, 0, 0)]
        for n, sz, of in zip(names, sizes, offs):
            links.append(_link(n + "_link", [(of, (0, 0, 0), _box(*sz))]))
        org = [(0.0, 0.19 * sy, 0.55), (0.1, 0, 0), (0, 0, 0), (0.4, 0, 0), (0, 0, 0), (0.32, 0, 0), (0, 0, 0)]
        axes = [(0, 0, 1), (0, 1, 0), (1, 0, 0), (0, 1, 0), (1, 0, 0), (0, 1, 0), (1, 0, 0)]
        lim = [(-2.1 if sy < 0 else -0.7, 0.7 if sy < 0 else 2.1), (-0.5, 1.3), (-3.0, 3.0), (-2.3, 0.0), (-3.1, 3.1),
               (-2.0, 0.0), (-3.1, 3.1)]
        parent = "torso"
        for n, o, a, (lo, hi) in zip(names, org, axes, lim):
            joints.append(_rev(n, parent, n + "_link", o, a, lo, hi))
            parent = n + "_link"
    return U.UrdfRobot("dual_arm", links, joints)


def dual_arm_self_pairs() -> List[Tuple[str, str]]:
    """Every left-arm joint against every right-arm joint from the upper arm down, both forearms and
    grippers against the torso and the base."""
    tail = ("upper_arm", "elbow", "forearm", "wrist_flex", "gripper")
    pairs = [(f"l_{a}", f"r_{b}") for a in tail for b in tail]
    for side in ("l", "r"):
        for n in ("forearm", "wrist_flex", "gripper"):
            pairs += [(f"{side}_{n}", "torso_lift"), (f"{side}_{n}", "root")]
    return pairs


class Workload:
    """A flattened robot + obstacle scene plus what is needed to draw inputs for it."""

    def __init__(self, name: str, robot: U.UrdfRobot, objects: D.Compound, margin: float,
                 self_pairs: Sequence[Tuple[str, str]] = (), end_joint: Optional[str] = None, model_edit=None):
        self.name = name
        self.robot = robot
        self.chain = U.Chain.from_urdf(robot)
        self.model = D.CollisionModel.from_urdf_robot(robot, margin)
        if model_edit is not None:
            model_edit(self.model)
        self.model.self_collision_pairs = list(self_pairs)
        self.objects = objects
        self.margin = margin
        if end_joint is None:
            self.using = self.chain.movable_nodes()
        else:
            self.using = self.chain.serial_chain_from_end(self.chain.find(end_joint))
        self.robot_keep = D.flatten_robot(self.chain, self.model, self.using)
        self.obstacle_keep = D.flatten_obstacles(objects)

    @property
    def dof(self) -> int:
        return len(self.using)

    def configs(self, n, seed=0, dtype=np.float32, overshoot=0.0):
        return random_configs(self.chain, self.using, n, seed, dtype, overshoot)

    def edges(self, n_edges, seed=0, dtype=np.float32, **kw):
        return straight_edges(self.chain, self.using, n_edges, seed, dtype=dtype, **kw)

    def edges_lockstep(self, n_edges, seed=0, dtype=np.float32, **kw):
        return straight_edges_lockstep(self.chain, self.using, n_edges, seed, dtype=dtype, **kw)


def metric_workload(n_obstacles: int = 256, seed: int = 0, margin: float = 0.01) -> Workload:
    """BASELINE.json's metric scene: 7-DOF arm vs 256 obstacles, env + self + limit flags."""
    return Workload("arm7_vs_%d_box_ball" % n_obstacles, arm7_robot(),
                    obstacle_field(n_obstacles, seed, ("box", "ball"), (0.6, 0.4)), margin, ARM7_SELF_PAIRS)


def metric_workload_sparse(n_obstacles: int = 256, seed: int = 0, margin: float = 0.01) -> Workload:
    """The metric scene's robot in a free-space-dominant field: the same obstacles spread over a 4.8 m cube
    (13 % of random configurations collide with the environment instead of 59 %)."""
    return Workload("arm7_vs_%d_box_ball_sparse" % n_obstacles, arm7_robot(),
                    obstacle_field(n_obstacles, seed, ("box", "ball"), (0.6, 0.4), cube=4.8), margin, ARM7_SELF_PAIRS)


def ur10_workload(n_obstacles: int = 256, seed: int = 1, margin: float = 0.01) -> Workload:
    """configs[2]: 6-DOF arm vs a box/cylinder field (cylinder obstacles: reference-undefined)."""
    return Workload("arm6_vs_%d_box_cyl" % n_obstacles, arm6_robot(),
                    obstacle_field(n_obstacles, seed, ("box", "cyl"), (0.5, 0.5)), margin, ARM6_SELF_PAIRS)


def sawyer_workload(n_obstacles: int = 4096, seed: int = 2, margin: float = 0.01) -> Workload:
    """configs[3]: 7-DOF arm vs 4096 mixed primitives (a denser, larger field)."""
    return Workload("arm7_vs_%d_mixed" % n_obstacles, arm7_robot(),
                    obstacle_field(n_obstacles, seed, ("box", "ball", "cyl"), (0.4, 0.3, 0.3), cube=4.0), margin,
                    ARM7_SELF_PAIRS)


def pr2_workload(n_obstacles: int = 128, seed: int = 4, margin: float = 0.01, end_joint: Optional[str] = None) -> Workload:
    """configs[4]: a dual-arm robot whose arm links are convex meshes (<= 64 vertices; gear's
    urdf.rs:54-71 would load them with assimp), with the full arm-vs-arm self pair list.  17 robot
    shapes in a branching chain: exercises the multi-chunk sweep, the FK branch slots and the general
    mesh support function."""
    rng = np.random.Generator(np.random.Philox(seed + 1000))

    def to_meshes(model: D.CollisionModel):
        for name, specs in model.name_collision_model_map.items():
            if name in ("root", "torso_lift") or name.endswith("elbow"):
                continue                       # boxes stay boxes
            for i, sp in enumerate(specs):
                v, t = convex_mesh(rng, np.asarray(sp.params) * rng.uniform(0.9, 1.1, 3))
                specs[i] = D.ShapeSpec.mesh(v, t, sp.t, sp.q)

    field = obstacle_field(n_obstacles, seed, ("box", "ball"), (0.6, 0.4), cube=2.8, base=(0.0, 0.0, 0.5), keep_out=0.55)
    return Workload("dual_arm_mesh_vs_%d" % n_obstacles, dual_arm_robot(), field, margin, dual_arm_self_pairs(),
                    end_joint, model_edit=to_meshes)


def sample_workload(sample_robot: U.UrdfRobot, obstacles_robot: U.UrdfRobot, margin: float = 0.01,
                    arm_only: bool = False, self_pairs: Sequence[Tuple[str, str]] = ()) -> Workload:
    """configs[0]/[1]: the reference's sample.urdf vs obstacles.urdf (examples/minimum.rs:26-37)."""
    return Workload("sample_urdf", sample_robot, D.Compound.from_urdf_robot(obstacles_robot), margin, self_pairs,
                    "l_tool_fixed" if arm_only else None)
