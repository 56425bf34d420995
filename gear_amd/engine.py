"""ctypes binding of libgear_hip.so — the C ABI declared in include/gear_hip.h.

The HIP library is the only compute path: if it is missing, or no MI355X/HIP
device is present, construction raises; nothing here falls back to the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import descriptors as D
from .build import LIB_PATH

_LIB = None

SYMBOLS = [
    "gh_scene_create", "gh_scene_create_multi", "gh_scene_device_count", "gh_scene_destroy", "gh_scene_dof", "gh_scene_nodes", "gh_scene_precision",
    "gh_scene_device", "gh_check_batch", "gh_check_batch_dev", "gh_check_edges", "gh_check_edges_dev",
    "gh_fk_batch", "gh_fk_batch_dev", "gh_link_hits", "gh_link_mask_words", "gh_self_hits",
    "gh_check_batch_counted", "gh_check_batch_noalloc", "gh_pipeline_last_stats", "gh_last_error", "gh_version",
    "gh_check_edges_from_endpoints", "gh_check_edges_from_endpoints_dev",
]


class GearHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"gear_hip error {code}: {msg}")
        self.code = code


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("GEAR_HIP_LIB", LIB_PATH)   # another build of the same library (A/B measurements)
        if not os.path.exists(path):
            raise GearHipError(-3, f"{path} is missing: run `python -m gear_amd.build` (no CPU fallback exists)")
        L = C.CDLL(path)
        vp, sz, u32, u8 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint8
        L.gh_scene_create.argtypes = [vp, vp, C.c_double, C.c_int, C.c_int, C.POINTER(vp)]
        L.gh_scene_create_multi.argtypes = [vp, vp, C.c_double, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
        L.gh_scene_device_count.argtypes = [vp]
        L.gh_scene_destroy.argtypes = [vp]
        L.gh_scene_destroy.restype = None
        for name in ("gh_scene_dof", "gh_scene_nodes", "gh_scene_precision", "gh_scene_device", "gh_link_mask_words"):
            getattr(L, name).argtypes = [vp]
        L.gh_check_batch.argtypes = [vp, vp, sz, u32, vp]
        L.gh_check_batch_dev.argtypes = [vp, vp, sz, u32, vp, vp]
        L.gh_check_batch_noalloc.argtypes = [vp, vp, sz, u32, vp, vp]
        L.gh_check_edges.argtypes = [vp, vp, vp, sz, u32, u8, vp, vp]
        L.gh_check_edges_dev.argtypes = [vp, vp, vp, sz, sz, u32, u8, vp, vp, vp]
        L.gh_check_edges_from_endpoints.argtypes = [vp, vp, vp, sz, C.c_double, u32, u32, u32, u8, vp, vp]
        L.gh_check_edges_from_endpoints_dev.argtypes = [vp, vp, vp, sz, C.c_double, u32, u32, u32, u8, vp, sz, vp, vp, vp, vp,
                                                        C.POINTER(sz), vp]
        L.gh_fk_batch.argtypes = [vp, vp, sz, vp]
        L.gh_fk_batch_dev.argtypes = [vp, vp, sz, vp, vp]
        L.gh_link_hits.argtypes = [vp, vp, sz, u32, vp]
        L.gh_self_hits.argtypes = [vp, vp, sz, u32, vp]
        L.gh_check_batch_counted.argtypes = [vp, vp, sz, u32, vp, vp]
        L.gh_pipeline_last_stats.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.gh_last_error.restype = C.c_char_p
        L.gh_version.restype = C.c_char_p
        _LIB = L
    return _LIB


def _check(rc: int):
    if rc != 0:
        raise GearHipError(rc, lib().gh_last_error().decode())


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Scene:
    """An immutable robot + obstacle scene resident in HBM (gh_scene)."""

    def __init__(self, robot_keep, obstacle_keep, margin: float, precision: int = 32, device: int = 0, devices=None):
        """devices: a list of device indices -> gh_scene_create_multi (the batch is sharded over them)."""
        self._keep = (robot_keep, obstacle_keep)
        self.h = C.c_void_p()
        if devices is not None:
            arr = (C.c_int * len(devices))(*devices)
            _check(lib().gh_scene_create_multi(C.addressof(robot_keep.desc), C.addressof(obstacle_keep.desc), float(margin),
                                               int(precision), arr, len(devices), C.byref(self.h)))
            device = devices[0]
        else:
            _check(lib().gh_scene_create(C.addressof(robot_keep.desc), C.addressof(obstacle_keep.desc), float(margin),
                                         int(precision), int(device), C.byref(self.h)))
        self.precision = precision
        self.dtype = np.float32 if precision == 32 else np.float64
        self.device = device
        self.margin = margin
        self.dof = lib().gh_scene_dof(self.h)
        self.n_nodes = lib().gh_scene_nodes(self.h)
        self.n_obstacles = obstacle_keep.desc.n_shapes
        self.n_pairs = robot_keep.desc.n_pairs

    def close(self):
        if getattr(self, "h", None):
            lib().gh_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-buffer entry points (numpy) ----
    def _q(self, q):
        q = np.ascontiguousarray(q, dtype=self.dtype)
        if q.ndim != 2 or q.shape[1] != self.dof:
            raise GearHipError(-2, f"DoF mismatch {q.shape[-1] if q.ndim else 0} != {self.dof}")
        return q

    def check_batch(self, q, what=D.GH_CHECK_ENV | D.GH_CHECK_SELF) -> np.ndarray:
        q = self._q(q)
        flags = np.zeros(q.shape[0], np.uint8)
        _check(lib().gh_check_batch(self.h, _ptr(q), q.shape[0], what, _ptr(flags)))
        return flags

    def check_edges(self, q, edge_off, what=D.GH_CHECK_ENV, bad_mask=D.GH_FLAG_ENV | D.GH_FLAG_LIMIT):
        q = self._q(q)
        edge_off = np.ascontiguousarray(edge_off, np.uint32)
        E = len(edge_off) - 1
        if E >= 0 and len(edge_off) and int(edge_off[-1]) != q.shape[0]:
            raise GearHipError(-1, "edge_off[-1] must equal the number of samples")
        flags = np.zeros(q.shape[0], np.uint8)
        first_bad = np.full(max(E, 0), -1, np.int32)
        _check(lib().gh_check_edges(self.h, _ptr(q), _ptr(edge_off), max(E, 0), what, bad_mask, _ptr(flags),
                                    _ptr(first_bad)))
        return flags, first_bad

    def check_edges_from_endpoints(self, start, goal, step=0.1, max_len=4096, mode=D.GH_EDGE_CONNECT, what=D.GH_CHECK_ENV,
                                   bad_mask=D.GH_FLAG_ENV | D.GH_FLAG_LIMIT):
        """(first_bad[E], n_samples[E]) of the edges start[e] -> goal[e], their samples generated on the device."""
        a, b = self._q(start), self._q(goal)
        assert a.shape == b.shape
        E = a.shape[0]
        first_bad = np.full(E, -1, np.int32)
        n_samples = np.zeros(E, np.uint32)
        _check(lib().gh_check_edges_from_endpoints(self.h, _ptr(a), _ptr(b), E, float(step), int(max_len), int(mode), what, bad_mask,
                                                   _ptr(first_bad), _ptr(n_samples)))
        return first_bad, n_samples

    def fk_batch(self, q) -> np.ndarray:
        q = self._q(q)
        poses = np.zeros((q.shape[0], self.n_nodes, 7), self.dtype)
        _check(lib().gh_fk_batch(self.h, _ptr(q), q.shape[0], _ptr(poses)))
        return poses

    def link_hits(self, q, what=0) -> np.ndarray:
        q = self._q(q)
        W = lib().gh_link_mask_words(self.h)
        m = np.zeros((q.shape[0], self.n_obstacles, W), np.uint64)
        _check(lib().gh_link_hits(self.h, _ptr(q), q.shape[0], what, _ptr(m)))
        return m

    def self_hits(self, q, what=0) -> np.ndarray:
        q = self._q(q)
        W = max((self.n_pairs + 63) // 64, 1)
        m = np.zeros((q.shape[0], W), np.uint64)
        _check(lib().gh_self_hits(self.h, _ptr(q), q.shape[0], what, _ptr(m)))
        return m

    # ---- device-buffer entry points (torch tensors on this scene's device) ----
    def check_batch_dev(self, q_t, flags_t, what=D.GH_CHECK_ENV | D.GH_CHECK_SELF, stream: Optional[int] = None):
        n = q_t.shape[0]
        assert q_t.is_contiguous() and flags_t.is_contiguous() and q_t.shape[1] == self.dof and flags_t.numel() >= n
        _check(lib().gh_check_batch_dev(self.h, q_t.data_ptr(), n, what, flags_t.data_ptr(), stream))

    def check_edges_dev(self, q_t, edge_off_t, flags_t, first_bad_t, what=D.GH_CHECK_ENV,
                        bad_mask=D.GH_FLAG_ENV | D.GH_FLAG_LIMIT, stream: Optional[int] = None):
        n, E = q_t.shape[0], edge_off_t.numel() - 1
        _check(lib().gh_check_edges_dev(self.h, q_t.data_ptr(), edge_off_t.data_ptr(), E, n, what, bad_mask,
                                        flags_t.data_ptr(), first_bad_t.data_ptr(), stream))

    def check_edges_from_endpoints_dev(self, from_t, to_t, q_t, edge_off_t, n_samples_t, flags_t, first_bad_t, step=0.1, max_len=4096,
                                       mode=D.GH_EDGE_CONNECT, what=D.GH_CHECK_ENV, bad_mask=D.GH_FLAG_ENV | D.GH_FLAG_LIMIT,
                                       stream: Optional[int] = None) -> int:
        """Device arrays in, device arrays out; returns the number of samples written to q_t / flags_t."""
        E = from_t.shape[0]
        total = C.c_size_t(0)
        _check(lib().gh_check_edges_from_endpoints_dev(self.h, from_t.data_ptr(), to_t.data_ptr(), E, float(step), int(max_len), int(mode),
                                                       what, bad_mask, q_t.data_ptr(), q_t.shape[0], edge_off_t.data_ptr(),
                                                       n_samples_t.data_ptr(), flags_t.data_ptr(), first_bad_t.data_ptr(),
                                                       C.byref(total), stream))
        return int(total.value)

    def fk_batch_dev(self, q_t, poses_t, stream: Optional[int] = None):
        _check(lib().gh_fk_batch_dev(self.h, q_t.data_ptr(), q_t.shape[0], poses_t.data_ptr(), stream))

    def pipeline_last_stats(self) -> dict:
        out = (C.c_uint64 * 4)()
        _check(lib().gh_pipeline_last_stats(self.h, out))
        return {"candidates": int(out[0]), "exact_items": int(out[1]), "redo_waves": int(out[2]), "exact_cap": int(out[3])}

    def check_batch_counted(self, q_t, flags_t, what=D.GH_CHECK_ENV | D.GH_CHECK_SELF) -> dict:
        c = D.gh_counters()
        _check(lib().gh_check_batch_counted(self.h, q_t.data_ptr(), q_t.shape[0], what, flags_t.data_ptr(),
                                            C.byref(c)))
        return {k: int(getattr(c, k)) for k in ("configs", "pairs", "cull_pass", "sat_tests", "exact_tests",
                                                "exact_iters")}
