"""Builds and loads tests/libhost_harness.so (the product's narrow phase compiled for the CPU;
test harness only)."""
import ctypes as C
import os
import subprocess

from gear_amd import descriptors as D

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _load(so_name, extra_flags):
    so = os.path.join(_HERE, so_name)
    src = os.path.join(_HERE, "host_harness.cpp")
    deps = [src] + [os.path.join(_HERE, "..", "gear_amd", "csrc", f)
                    for f in ("gear_device.hpp", "gear_scene.hpp", "gear_types.hpp", "gear_shard.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["hipcc", "-x", "hip", "--cuda-host-only", "-O2", "-std=c++17", "-shared", "-fPIC"]
                              + extra_flags + ["-o", so, src])
    L = C.CDLL(so)
    L.hh_pair_within.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                 C.c_double, C.c_int]
    L.hh_shard_range.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.hh_shard_range.restype = None
    L.hh_shard_edges.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
    L.hh_shard_edges.restype = None
    return L


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _load("libhost_harness.so", [])
    return _LIB


_LIB_CAPPED = None


def lib_capped():
    """The same narrow phase with the GJK iteration cap forced down to ONE iteration
    (-DGEAR_GJK_MAXIT=1): whatever one iteration cannot decide leaves through the cap exit."""
    global _LIB_CAPPED
    if _LIB_CAPPED is None:
        _LIB_CAPPED = _load("libhost_harness_maxit1.so", ["-DGEAR_GJK_MAXIT=1"])
    return _LIB_CAPPED


def pair_within(a: D.ShapeSpec, b: D.ShapeSpec, margin: float, precision: int, capped: bool = False) -> int:
    refs = []
    sa, ma, na = D._fill_shapes([(-1, a)], refs)
    sb, mb, nb = D._fill_shapes([(-1, b)], refs)
    return (lib_capped() if capped else lib()).hh_pair_within(C.addressof(sa), C.addressof(ma), na, C.addressof(sb), C.addressof(mb), nb,
                                margin, precision)


def shard_range(n: int, k: int, G: int):
    lo, hi = C.c_size_t(), C.c_size_t()
    lib().hh_shard_range(n, k, G, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def shard_edges(edge_off, G: int):
    import numpy as np
    off = np.ascontiguousarray(edge_off, np.uint32)
    cut = np.zeros(G + 1, np.uint64)
    lib().hh_shard_edges(off.ctypes.data_as(C.c_void_p), len(off) - 1, G, cut.ctypes.data_as(C.c_void_p))
    return cut.astype(np.int64)
