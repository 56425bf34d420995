"""Host-side mirror of gear's planner predicates over the HIP engine.

Same names, argument meaning and error behaviour as the reference for the hot path only
(src/planner/joint_path_planner.rs:26-134, 258-365): the RRT / IK drivers are out of scope and keep
calling these predicates.  Every predicate has a batched twin, which is what the GPU is for.

A `k::Chain` is shared mutable state in the reference (`Rc<RefCell>` nodes): `is_feasible` leaves the
joints it managed to set where they are, and `has_any_colliding` / `colliding_link_names` look at
that state.  `JointPathPlanner.positions` plays that role here.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import descriptors as D
from . import engine
from . import urdf as U


class Error(Exception):
    """gear::Error (src/errors.rs:26-66), the variants the hot path can raise."""


class DofMismatch(Error):
    def __init__(self, a, b):
        super().__init__(f"DoF mismatch {a} != {b}")


class NotFound(Error):
    def __init__(self, name):
        super().__init__(f"Node name {name} not found")


class Collision(Error):
    def __init__(self, part, names):
        super().__init__(f"Collision error: {names} is colliding ({part})")
        self.part, self.collision_link_names = part, names


class SelfCollision(Error):
    def __init__(self, part, names):
        super().__init__(f"Self Collision error: {names} is colliding ({part})")
        self.part, self.collision_link_names = part, names


class JointPathPlanner:
    """Collision Avoidance Path Planner — feasibility predicates on the GPU."""

    def __init__(self, chain: U.Chain, collision_checker: D.CollisionModel, step_length=0.1, max_try=5000,
                 num_smoothing=100, precision: int = 64, device: int = 0):
        self.collision_check_robot = chain
        self.collision_checker = collision_checker
        self.step_length, self.max_try, self.num_smoothing = step_length, max_try, num_smoothing
        self.self_collision_pairs: List[Tuple[str, str]] = []
        self.urdf_robot: Optional[U.UrdfRobot] = None
        self.precision, self.device = precision, device
        self.positions = [n.position for n in chain.nodes]
        self._scenes: Dict[tuple, engine.Scene] = {}

    # -- scene cache: one gh_scene per (using_joints, obstacle compound, frozen joints' state) ---
    def _scene(self, using: Sequence[int], objects: Optional[D.Compound]) -> engine.Scene:
        using = tuple(using)
        driven = set(using)
        for i in using:
            driven.update(k for k, n in enumerate(self.collision_check_robot.nodes) if n.mimic_parent == i)
        frozen = tuple((k, p) for k, p in enumerate(self.positions) if k not in driven)
        key = (using, objects.uid if objects is not None else 0, frozen, self.collision_checker.prediction, tuple(self.self_collision_pairs))
        sc = self._scenes.get(key)
        if sc is None:
            for k, p in enumerate(self.positions):
                self.collision_check_robot.nodes[k].position = p
            rk = D.flatten_robot(self.collision_check_robot, self.collision_checker, using,
                                 self.self_collision_pairs)
            ok = D.flatten_obstacles(objects if objects is not None else D.Compound([]))
            sc = engine.Scene(rk, ok, self.collision_checker.prediction, self.precision, self.device)
            sc.pair_names = rk.pair_names
            if len(self._scenes) > 16:
                self._scenes.pop(next(iter(self._scenes))).close()
            self._scenes[key] = sc
        return sc

    def _all(self) -> List[int]:
        return self.collision_check_robot.movable_nodes()

    def _set_joint_positions(self, using: Sequence[int], q: Sequence[float]) -> bool:
        """k::Chain::set_joint_positions incl. the partially-set state it leaves behind on Err."""
        nodes = self.collision_check_robot.nodes
        if len(q) != len(using):
            return False          # Err(DofMismatch) folds to `false` like any other Err (:79-82)
        for i, v in zip(using, q):
            n = nodes[i]
            if n.mimic_parent >= 0:
                continue
            if n.limits is not None and not (n.limits[0] <= v <= n.limits[1]):
                return False
            self.positions[i] = float(v)
            for k, c in enumerate(nodes):
                if c.mimic_parent == i:
                    cv = c.mimic_offset + c.mimic_multiplier * float(v)
                    if c.limits is not None and not (c.limits[0] <= cv <= c.limits[1]):
                        return False
                    self.positions[k] = cv
        return True

    def _current(self, using):
        return np.array([[self.positions[i] for i in using]], np.float64)

    # -- the six predicates (joint_path_planner.rs:71-134) ---------------------------------------
    def is_feasible(self, using_joints: Sequence[int], joint_positions: Sequence[float], objects: D.Compound) -> bool:
        if not self._set_joint_positions(using_joints, joint_positions):
            return False
        return not self.has_any_colliding(objects)

    def has_any_colliding(self, objects: D.Compound) -> bool:
        # no limit gate: the reference evaluates distances on whatever state the chain holds (:86-98)
        f = self._scene(self._all(), objects).check_batch(self._current(self._all()),
                                                          D.GH_CHECK_ENV | D.GH_CHECK_NO_LIMITS)
        return bool(f[0] & D.GH_FLAG_ENV)

    def colliding_link_names(self, objects: D.Compound) -> List[str]:
        sc = self._scene(self._all(), objects)
        m = sc.link_hits(self._current(self._all()), D.GH_CHECK_NO_LIMITS)[0]
        names = []
        for o in range(m.shape[0]):                       # obstacle-major, duplicates kept
            for k, n in enumerate(self.collision_check_robot.nodes):
                if (int(m[o, k >> 6]) >> (k & 63)) & 1:
                    names.append(n.name)
        return names

    def is_feasible_with_self(self, using_joints: Sequence[int], joint_positions: Sequence[float]) -> bool:
        if not self._set_joint_positions(using_joints, joint_positions):
            return False
        return not self.has_any_colliding_with_self()

    def has_any_colliding_with_self(self) -> bool:
        f = self._scene(self._all(), None).check_batch(self._current(self._all()),
                                                       D.GH_CHECK_SELF | D.GH_CHECK_NO_LIMITS)
        return bool(f[0] & D.GH_FLAG_SELF)

    def colliding_link_names_with_self(self) -> List[Tuple[str, str]]:
        sc = self._scene(self._all(), None)
        m = sc.self_hits(self._current(self._all()), D.GH_CHECK_NO_LIMITS)[0]
        return [pn for p, pn in enumerate(sc.pair_names) if (int(m[p >> 6]) >> (p & 63)) & 1]

    # -- batched forms: what the GPU is for -------------------------------------------------------
    def is_feasible_batch(self, using_joints, q, objects, with_self=False) -> np.ndarray:
        """is_feasible (or is_feasible_with_self) for every row of q; chain state is not touched."""
        sc = self._scene(using_joints, None if with_self else objects)
        f = sc.check_batch(q, D.GH_CHECK_SELF if with_self else D.GH_CHECK_ENV)
        bad = (D.GH_FLAG_SELF if with_self else D.GH_FLAG_ENV) | D.GH_FLAG_LIMIT
        return (f & bad) == 0

    def first_infeasible(self, using_joints, q, edge_off, objects, with_self=False):
        """Per edge, the index of the first sample rrt's stepping would reject, or -1."""
        sc = self._scene(using_joints, None if with_self else objects)
        return sc.check_edges(q, edge_off, D.GH_CHECK_SELF if with_self else D.GH_CHECK_ENV,
                              (D.GH_FLAG_SELF if with_self else D.GH_FLAG_ENV) | D.GH_FLAG_LIMIT)[1]

    def check_start_goal(self, using_joints, start, goal, objects):
        """The validation at the top of plan() (joint_path_planner.rs:151-167)."""
        current = [self.positions[i] for i in using_joints]
        for part, q in (("Start", start), ("End", goal)):
            if not self.is_feasible(using_joints, q, objects):
                # the reference restores the joints (:156) before it collects the names (:159)
                self._set_joint_positions(using_joints, current)
                raise Collision(part, self.colliding_link_names(objects))

    def joint_names(self) -> List[str]:
        return self.collision_check_robot.joint_names()


class JointPathPlannerBuilder:
    """Builder pattern to create `JointPathPlanner` (joint_path_planner.rs:258-365)."""

    def __init__(self, urdf_robot: U.UrdfRobot, collision_checker: D.CollisionModel):
        self.urdf_robot = urdf_robot
        self.collision_checker = collision_checker
        self._step_length, self._max_try, self._num_smoothing = 0.1, 5000, 100
        self._margin: Optional[float] = None
        self._pairs: List[Tuple[str, str]] = []
        self._precision, self._device = 64, 0

    @classmethod
    def from_urdf_file(cls, path: str) -> "JointPathPlannerBuilder":
        robot = U.read_urdf(path)
        return cls(robot, D.CollisionModel.from_urdf_robot(robot, 0.0))   # default margin 0.0 (:341)

    @classmethod
    def from_urdf_robot(cls, robot: U.UrdfRobot) -> "JointPathPlannerBuilder":
        return cls(robot, D.CollisionModel.from_urdf_robot(robot, 0.0))

    def collision_check_margin(self, length: float):
        self._margin = length
        return self

    def step_length(self, v: float):
        self._step_length = v
        return self

    def max_try(self, v: int):
        self._max_try = v
        return self

    def num_smoothing(self, v: int):
        self._num_smoothing = v
        return self

    def self_collision_pairs(self, pairs):
        self._pairs = list(pairs)
        return self

    def precision(self, bits: int, device: int = 0):
        """Not in the reference: N = f32|f64 is a type parameter there."""
        self._precision, self._device = bits, device
        return self

    def finalize(self) -> JointPathPlanner:
        if self._margin is not None:
            self.collision_checker.prediction = self._margin
        p = JointPathPlanner(U.Chain.from_urdf(self.urdf_robot), self.collision_checker, self._step_length,
                             self._max_try, self._num_smoothing, self._precision, self._device)
        p.urdf_robot = self.urdf_robot
        p.self_collision_pairs = self._pairs
        return p
