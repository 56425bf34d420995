"""ctypes binding of libgear_host.so — the C++ host layer (gear_amd/host/gear_host.hpp) behind the C
entry points of include/gear_host.h.  Test and tooling support: the C++ layer is the host side of
the product; this module only lets Python drive it."""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import descriptors as D
from .build import HOST_LIB_PATH

_LIB = None

SYMBOLS = [
    "gp_last_error", "gp_planner_create", "gp_planner_destroy", "gp_planner_set_speculation",
    "gp_compound_from_urdf_file", "gp_compound_from_shapes", "gp_compound_destroy", "gp_compound_size",
    "gp_using_joints", "gp_describe", "gp_is_feasible", "gp_is_feasible_with_self", "gp_has_any_colliding",
    "gp_has_any_colliding_with_self", "gp_colliding_link_names", "gp_colliding_link_names_with_self",
    "gp_is_feasible_batch", "gp_plan", "gp_plan_avoid_self_collision", "gp_last_collision_names",
    "gp_feasible_ik_seeds", "gp_joint_positions", "gp_joint_names",
    "gp_generate_clamped_joint_positions_from_limits", "gp_generate_random_joint_positions_from_limits",
    "gp_modify_to_nearest_angle", "gp_edge_line", "gp_first_infeasible_from_endpoints",
]

GP_ERR = {-10: "CollisionStart", -11: "CollisionEnd", -12: "SelfCollisionStart", -13: "SelfCollisionEnd",
          -14: "PathPlanFail", -15: "DofMismatch", -16: "NotFound", -17: "ParseError", -18: "Io", -19: "Gpu", -1: "Other"}


class GearHostError(RuntimeError):
    def __init__(self, code: int, msg: str, names: Optional[List[str]] = None):
        super().__init__(f"gear_host {GP_ERR.get(code, code)}: {msg}")
        self.code, self.kind, self.names = code, GP_ERR.get(code, str(code)), names or []


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError(f"{HOST_LIB_PATH} is missing: run `python -m gear_amd.build`")
        L = C.CDLL(HOST_LIB_PATH)
        vp, cp, i32p, dp, u8p = C.c_void_p, C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_uint8)
        L.gp_last_error.restype = cp
        L.gp_planner_create.argtypes = [cp, C.c_double, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.POINTER(cp),
                                        C.c_int, C.c_uint64, C.POINTER(vp)]
        L.gp_planner_destroy.argtypes = [vp]
        L.gp_planner_destroy.restype = None
        L.gp_planner_set_speculation.argtypes = [vp, C.c_int, C.c_int, C.c_uint64]
        L.gp_compound_from_urdf_file.argtypes = [cp, C.POINTER(vp)]
        L.gp_compound_from_shapes.argtypes = [vp, C.c_int, C.POINTER(vp)]
        L.gp_compound_destroy.argtypes = [vp]
        L.gp_compound_destroy.restype = None
        L.gp_compound_size.argtypes = [vp]
        L.gp_using_joints.argtypes = [vp, cp, i32p, C.c_int]
        L.gp_describe.argtypes = [vp]
        L.gp_describe.restype = cp
        L.gp_is_feasible.argtypes = [vp, i32p, C.c_int, dp, vp]
        L.gp_is_feasible_with_self.argtypes = [vp, i32p, C.c_int, dp]
        L.gp_has_any_colliding.argtypes = [vp, vp]
        L.gp_has_any_colliding_with_self.argtypes = [vp]
        L.gp_colliding_link_names.argtypes = [vp, vp, C.POINTER(cp)]
        L.gp_colliding_link_names_with_self.argtypes = [vp, C.POINTER(cp)]
        L.gp_is_feasible_batch.argtypes = [vp, i32p, C.c_int, dp, C.c_size_t, vp, C.c_int, u8p]
        L.gp_plan.argtypes = [vp, i32p, C.c_int, dp, dp, vp, dp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
        L.gp_plan_avoid_self_collision.argtypes = [vp, i32p, C.c_int, dp, dp, dp, C.c_int, C.POINTER(C.c_int),
                                                   C.POINTER(C.c_uint64)]
        L.gp_last_collision_names.argtypes = [vp]
        L.gp_last_collision_names.restype = cp
        L.gp_feasible_ik_seeds.argtypes = [vp, i32p, C.c_int, dp, C.c_size_t, vp, dp, u8p]
        L.gp_joint_positions.argtypes = [vp, i32p, C.c_int, dp]
        L.gp_joint_names.argtypes = [vp]
        L.gp_joint_names.restype = cp
        L.gp_generate_clamped_joint_positions_from_limits.argtypes = [dp, C.c_int, dp, dp, u8p, C.c_int, C.c_int, dp]
        L.gp_generate_random_joint_positions_from_limits.argtypes = [dp, dp, u8p, C.c_int, C.c_int, C.c_uint64, C.c_size_t, dp]
        L.gp_modify_to_nearest_angle.argtypes = [dp, dp, C.c_int, dp, dp, u8p, C.c_int]
        L.gp_edge_line.argtypes = [dp, dp, C.c_int, C.c_double, C.c_int, C.c_int, dp, C.c_int]
        L.gp_first_infeasible_from_endpoints.argtypes = [vp, i32p, C.c_int, dp, dp, C.c_size_t, C.c_int, vp, C.c_int,
                                                         i32p, C.POINTER(C.c_uint32)]
        _LIB = L
    return _LIB


def _i32(v):
    a = np.ascontiguousarray(v, np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


def _f64(v):
    a = np.ascontiguousarray(v, np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def _limits(limits):
    """limits: sequence of None | (lo, hi), the reference's Vec<Option<Range>>."""
    lo = np.array([0.0 if l is None else l[0] for l in limits], np.float64)
    hi = np.array([0.0 if l is None else l[1] for l in limits], np.float64)
    has = np.array([l is not None for l in limits], np.uint8)
    return lo, hi, has


def _raise(rc):
    if rc < 0:
        raise GearHostError(rc, lib().gp_last_error().decode())


def generate_clamped_joint_positions_from_limits(angles, limits, precision: int = 64) -> np.ndarray:
    """gear::generate_clamped_joint_positions_from_limits of the C++ host layer (funcs.rs:25-54)."""
    a, ap = _f64(angles)
    lo, hi, has = _limits(limits)
    out = np.zeros(len(limits), np.float64)
    _raise(lib().gp_generate_clamped_joint_positions_from_limits(
        ap, len(a), lo.ctypes.data_as(C.POINTER(C.c_double)), hi.ctypes.data_as(C.POINTER(C.c_double)),
        has.ctypes.data_as(C.POINTER(C.c_uint8)), len(limits), precision, out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


def generate_random_joint_positions_from_limits(limits, k: int, seed: int = 0, precision: int = 64) -> np.ndarray:
    """k draws of gear::generate_random_joint_positions_from_limits (funcs.rs:76-87), [k][len(limits)]."""
    lo, hi, has = _limits(limits)
    out = np.zeros((k, len(limits)), np.float64)
    _raise(lib().gp_generate_random_joint_positions_from_limits(
        lo.ctypes.data_as(C.POINTER(C.c_double)), hi.ctypes.data_as(C.POINTER(C.c_double)),
        has.ctypes.data_as(C.POINTER(C.c_uint8)), len(limits), precision, seed, k, out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


def modify_to_nearest_angle(vec1, vec2, limits, precision: int = 64) -> np.ndarray:
    """gear::modify_to_nearest_angle (funcs.rs:98-119); returns the modified copy of vec2."""
    a, ap = _f64(vec1)
    b = np.array(vec2, np.float64)
    lo, hi, has = _limits(limits)
    _raise(lib().gp_modify_to_nearest_angle(ap, b.ctypes.data_as(C.POINTER(C.c_double)), len(b),
                                            lo.ctypes.data_as(C.POINTER(C.c_double)), hi.ctypes.data_as(C.POINTER(C.c_double)),
                                            has.ctypes.data_as(C.POINTER(C.c_uint8)), precision))
    return b


def edge_line(start, goal, step: float, precision: int, mode: int = D.GH_EDGE_CONNECT, cap: int = 4096) -> np.ndarray:
    """The samples rrt's stepping evaluates between two configurations, by the C++ host layer."""
    a, ap = _f64(start)
    b, bp = _f64(goal)
    out = np.zeros((cap, len(a)), np.float64)
    n = lib().gp_edge_line(ap, bp, len(a), step, precision, mode, out.ctypes.data_as(C.POINTER(C.c_double)), cap)
    _raise(n)
    assert n <= cap
    return out[:n].copy()


class HostCompound:
    def __init__(self, handle):
        self.h = handle

    @classmethod
    def from_urdf_file(cls, path: str) -> "HostCompound":
        h = C.c_void_p()
        rc = lib().gp_compound_from_urdf_file(path.encode(), C.byref(h))
        if rc != 0:
            raise GearHostError(rc, lib().gp_last_error().decode())
        return cls(h)

    @classmethod
    def from_specs(cls, specs: Sequence[D.ShapeSpec]) -> "HostCompound":
        refs: list = []
        arr, _, _ = D._fill_shapes([(-1, s) for s in specs], refs)
        h = C.c_void_p()
        rc = lib().gp_compound_from_shapes(C.cast(arr, C.c_void_p), len(specs), C.byref(h))
        if rc != 0:
            raise GearHostError(rc, lib().gp_last_error().decode())
        return cls(h)

    def __len__(self):
        return lib().gp_compound_size(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().gp_compound_destroy(self.h)
            self.h = None


class HostPlanner:
    """gear::JointPathPlanner<N> of the C++ host layer."""

    def __init__(self, urdf_path: str, margin: float, precision: int = 64, device: int = 0, step_length: float = 0.1,
                 max_try: int = 5000, num_smoothing: int = 100, self_pairs: Sequence[str] = (), seed: int = 0):
        arr = (C.c_char_p * max(len(self_pairs), 1))(*[s.encode() for s in self_pairs])
        self.h = C.c_void_p()
        rc = lib().gp_planner_create(urdf_path.encode(), margin, precision, device, step_length, max_try, num_smoothing,
                                     arr, len(self_pairs), seed, C.byref(self.h))
        if rc != 0:
            raise GearHostError(rc, lib().gp_last_error().decode())

    def __del__(self):
        if getattr(self, "h", None):
            lib().gp_planner_destroy(self.h)
            self.h = None

    def _check(self, rc):
        if rc < 0:
            names = [s for s in lib().gp_last_collision_names(self.h).decode().split("\n") if s] if -13 <= rc <= -10 else []
            raise GearHostError(rc, lib().gp_last_error().decode(), names)
        return rc

    def set_speculation(self, speculate: bool, lookahead: int = 16, seed: int = 0):
        lib().gp_planner_set_speculation(self.h, int(speculate), lookahead, seed)

    def using_joints(self, end_joint: Optional[str] = None) -> List[int]:
        buf = (C.c_int32 * 256)()
        n = self._check(lib().gp_using_joints(self.h, end_joint.encode() if end_joint else None, buf, 256))
        return [buf[i] for i in range(n)]

    def describe(self) -> dict:
        return json.loads(lib().gp_describe(self.h).decode())

    def is_feasible(self, using, q, objects: HostCompound) -> bool:
        u, up = _i32(using)
        a, ap = _f64(q)
        return bool(self._check(lib().gp_is_feasible(self.h, up, len(u), ap, objects.h)))

    def is_feasible_with_self(self, using, q) -> bool:
        u, up = _i32(using)
        a, ap = _f64(q)
        return bool(self._check(lib().gp_is_feasible_with_self(self.h, up, len(u), ap)))

    def has_any_colliding(self, objects: HostCompound) -> bool:
        return bool(self._check(lib().gp_has_any_colliding(self.h, objects.h)))

    def has_any_colliding_with_self(self) -> bool:
        return bool(self._check(lib().gp_has_any_colliding_with_self(self.h)))

    def colliding_link_names(self, objects: HostCompound) -> List[str]:
        s = C.c_char_p()
        self._check(lib().gp_colliding_link_names(self.h, objects.h, C.byref(s)))
        return [x for x in s.value.decode().split("\n") if x]

    def colliding_link_names_with_self(self) -> List[Tuple[str, str]]:
        s = C.c_char_p()
        self._check(lib().gp_colliding_link_names_with_self(self.h, C.byref(s)))
        return [tuple(x.split(":")) for x in s.value.decode().split("\n") if x]

    def is_feasible_batch(self, using, q, objects: Optional[HostCompound], with_self=False) -> np.ndarray:
        u, up = _i32(using)
        a, ap = _f64(q)
        ok = np.zeros(a.shape[0], np.uint8)
        self._check(lib().gp_is_feasible_batch(self.h, up, len(u), ap, a.shape[0], objects.h if objects else None,
                                               int(with_self), ok.ctypes.data_as(C.POINTER(C.c_uint8))))
        return ok.astype(bool)

    def first_infeasible_from_endpoints(self, using, start, goal, objects: Optional[HostCompound], mode=D.GH_EDGE_CONNECT, with_self=False):
        u, up = _i32(using)
        a, ap = _f64(start)
        b, bp = _f64(goal)
        first = np.zeros(a.shape[0], np.int32)
        cnt = np.zeros(a.shape[0], np.uint32)
        self._check(lib().gp_first_infeasible_from_endpoints(self.h, up, len(u), ap, bp, a.shape[0], mode, objects.h if objects else None,
                                                             int(with_self), first.ctypes.data_as(C.POINTER(C.c_int32)),
                                                             cnt.ctypes.data_as(C.POINTER(C.c_uint32))))
        return first, cnt

    def _plan(self, fn, using, start, goal, objects):
        u, up = _i32(using)
        s, sp = _f64(start)
        g, gp = _f64(goal)
        cap = 20000
        path = np.zeros((cap, len(u)), np.float64)
        rows = C.c_int(0)
        stats = (C.c_uint64 * 8)()
        pp = path.ctypes.data_as(C.POINTER(C.c_double))
        if objects is None:
            rc = fn(self.h, up, len(u), sp, gp, pp, cap, C.byref(rows), stats)
        else:
            rc = fn(self.h, up, len(u), sp, gp, objects.h, pp, cap, C.byref(rows), stats)
        self.last_stats = dict(zip(("gpu_calls", "samples", "consumed", "iterations", "shortcuts", "eval_ns", "draw_ns", "apply_ns"), [int(x) for x in stats]))
        self._check(rc)
        return path[:rows.value].copy()

    def plan(self, using, start, goal, objects: HostCompound) -> np.ndarray:
        return self._plan(lib().gp_plan, using, start, goal, objects)

    def plan_avoid_self_collision(self, using, start, goal) -> np.ndarray:
        return self._plan(lib().gp_plan_avoid_self_collision, using, start, goal, None)

    def feasible_ik_seeds(self, using, initial, k: int, objects: HostCompound):
        u, up = _i32(using)
        a, ap = _f64(initial)
        seeds = np.zeros((k, len(u)), np.float64)
        ok = np.zeros(k, np.uint8)
        self._check(lib().gp_feasible_ik_seeds(self.h, up, len(u), ap, k, objects.h,
                                               seeds.ctypes.data_as(C.POINTER(C.c_double)),
                                               ok.ctypes.data_as(C.POINTER(C.c_uint8))))
        return seeds, ok.astype(bool)

    def joint_positions(self, using) -> np.ndarray:
        u, up = _i32(using)
        q = np.zeros(len(u), np.float64)
        lib().gp_joint_positions(self.h, up, len(u), q.ctypes.data_as(C.POINTER(C.c_double)))
        return q

    def joint_names(self) -> List[str]:
        return lib().gp_joint_names(self.h).decode().split("\n")
