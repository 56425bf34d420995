"""ctypes binding of the CPU ORACLE (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
and from nowhere else: the product (gear_amd/) never touches this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from gear_amd import descriptors as D

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    path = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("oracle_capi.cpp", "gear_oracle.hpp")]
    srcs.append(os.path.join(_HERE, "..", "include", "gear_hip.h"))
    stale = (not os.path.exists(path)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(path) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return path


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_check.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_explain.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_double, C.c_void_p, C.c_int]
        L.orc_check_edges.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint8,
                                      C.c_void_p, C.c_void_p]
        L.orc_fk.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_link_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]
        L.orc_self_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]
        L.orc_distance.restype = C.c_double
        L.orc_distance.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_distance_bounds.restype = C.c_double
        L.orc_distance_bounds.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_last_gjk.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_last_gjk.restype = None
        L.orc_project_simplex.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_cylinder_vertices.argtypes = [C.c_double, C.c_double, C.c_int, C.c_void_p]
        L.orc_quat_from_rpy.argtypes = [C.c_double, C.c_double, C.c_double, C.c_void_p]
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleScene:
    """CPU restatement of gear's CollisionChecker + k::Chain for one scene."""

    def __init__(self, robot_keep, obstacle_keep, margin: float, precision: int):
        self._keep = (robot_keep, obstacle_keep)
        self.precision = precision
        self.dtype = np.float32 if precision == 32 else np.float64
        self.dof = robot_keep.desc.n_cols
        self.n_nodes = robot_keep.desc.n_nodes
        self.n_obstacles = obstacle_keep.desc.n_shapes
        self.n_pairs = robot_keep.desc.n_pairs
        self.margin = float(np.float32(margin)) if precision == 32 else float(margin)   # CollisionChecker::prediction as T
        self.h = lib().orc_scene_create(C.addressof(robot_keep.desc), C.addressof(obstacle_keep.desc),
                                        margin, precision)
        if not self.h:
            raise RuntimeError("orc_scene_create failed")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_scene_destroy(self.h)
            self.h = None

    def _q(self, q):
        q = np.ascontiguousarray(q, dtype=self.dtype)
        assert q.ndim == 2 and q.shape[1] == self.dof, (q.shape, self.dof)
        return q

    def check(self, q, what=D.GH_CHECK_ENV | D.GH_CHECK_SELF, faithful=False, redundant_fk=False,
              threads=1, want_band=True, want_stats=False):
        """flags and, when not faithful, band[n][2]: band[:,0] = min |d - margin| over all pairs,
        band[:,1] = min (certified lower bound - margin) over pairs the oracle reports free."""
        q = self._q(q)
        n = q.shape[0]
        flags = np.zeros(n, np.uint8)
        band = np.zeros((n, 2), np.float64) if (want_band and not faithful) else None
        stats = np.zeros(3, np.uint64) if want_stats else None
        lib().orc_check(self.h, _ptr(q), n, what, int(faithful), int(redundant_fk), threads, _ptr(flags),
                        _ptr(band), _ptr(stats))
        out = [flags]
        if want_band and not faithful:
            out.append(band)
        if want_stats:
            out.append(stats)
        return out[0] if len(out) == 1 else tuple(out)

    KINDS = ("box", "ball", "triangle", "mesh")
    GJK_EXITS = ("none", "zero_projection", "no_progress", "converged", "point_rejected", "tetra_closest",
                 "tetra_intersection", "iteration_cap")

    def explain(self, q_row, what=D.GH_CHECK_ENV | D.GH_CHECK_SELF, window=0.0, cap=None):
        """Every pair of ONE configuration that is within `window` of the margin or reported free without
        its lower bound certifying it: list of dicts (self, id_a, id_b, kinds, d, lower, iters, exit).
        window=inf lists every pair."""
        q = self._q(np.asarray(q_row).reshape(1, -1))
        if cap is None:
            cap = 256 if window < 1.0 else 1 << 17
        out = np.zeros((cap, 9), np.float64)
        n = lib().orc_explain(self.h, _ptr(q), what, float(min(window, 1e30)), _ptr(out), cap)
        recs = []
        for r in out[:min(n, cap)]:
            recs.append({"self": bool(r[0]), "id_a": int(r[1]), "id_b": int(r[2]),
                         "kinds": self.KINDS[int(r[3])] + "-" + self.KINDS[int(r[4])], "d": float(r[5]),
                         "lower": float(r[6]), "iters": int(r[7]), "exit": self.GJK_EXITS[int(r[8])]})
        return recs

    def check_edges(self, q, edge_off, what=D.GH_CHECK_ENV, bad_mask=D.GH_FLAG_ENV | D.GH_FLAG_LIMIT):
        q = self._q(q)
        edge_off = np.ascontiguousarray(edge_off, np.uint32)
        E = len(edge_off) - 1
        flags = np.zeros(q.shape[0], np.uint8)
        first_bad = np.zeros(E, np.int32)
        lib().orc_check_edges(self.h, _ptr(q), _ptr(edge_off), E, what, bad_mask, _ptr(flags), _ptr(first_bad))
        return flags, first_bad

    def fk(self, q):
        q = self._q(q)
        poses = np.zeros((q.shape[0], self.n_nodes, 7), self.dtype)
        lib().orc_fk(self.h, _ptr(q), q.shape[0], _ptr(poses))
        return poses

    def link_hits(self, q, what=0):
        q = self._q(q)
        W = (self.n_nodes + 63) // 64
        m = np.zeros((q.shape[0], self.n_obstacles, W), np.uint64)
        lib().orc_link_hits(self.h, _ptr(q), q.shape[0], what, _ptr(m))
        return m

    def self_hits(self, q, what=0):
        q = self._q(q)
        W = max((self.n_pairs + 63) // 64, 1)
        m = np.zeros((q.shape[0], W), np.uint64)
        lib().orc_self_hits(self.h, _ptr(q), q.shape[0], what, _ptr(m))
        return m


def distance(spec_a: D.ShapeSpec, spec_b: D.ShapeSpec, precision: int = 64) -> float:
    """ncollide3d::query::distance restated, for two posed shapes."""
    refs = []
    sa, ma, _ = D._fill_shapes([(-1, spec_a)], refs)
    sb, mb, _ = D._fill_shapes([(-1, spec_b)], refs)
    return lib().orc_distance(C.addressof(sa), C.addressof(ma), C.addressof(sb), C.addressof(mb), precision)


def distance_bounds(spec_a: D.ShapeSpec, spec_b: D.ShapeSpec, precision: int = 64):
    """(distance as the restated reference returns it, lower bound its GJK run certified)."""
    refs = []
    sa, ma, _ = D._fill_shapes([(-1, spec_a)], refs)
    sb, mb, _ = D._fill_shapes([(-1, spec_b)], refs)
    lo = C.c_double(0.0)
    d = lib().orc_distance_bounds(C.addressof(sa), C.addressof(ma), C.addressof(sb), C.addressof(mb), precision,
                                  C.byref(lo))
    return d, lo.value


def last_gjk():
    """(iterations, exit reason) of the last GJK run made on this thread (OracleScene.GJK_EXITS names)."""
    it, ex = C.c_int(0), C.c_int(0)
    lib().orc_last_gjk(C.byref(it), C.byref(ex))
    return it.value, OracleScene.GJK_EXITS[ex.value]


def project_simplex(points) -> np.ndarray:
    p = np.ascontiguousarray(points, np.float64)
    out = np.zeros(3, np.float64)
    lib().orc_project_simplex(_ptr(p), p.shape[0], _ptr(out))
    return out


def cylinder_vertices(radius: float, length: float, precision: int = 64) -> np.ndarray:
    out = np.zeros((60, 3), np.float64)
    n = lib().orc_cylinder_vertices(radius, length, precision, _ptr(out))
    return out[:n]
