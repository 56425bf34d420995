"""gh_check_batch_noalloc (include/gear_hip.h): the same bytes as gh_check_batch_dev, through the C ABI, on
two calls in a row — the second with a smaller input, so that what the first call left in the shared
workspace array is still there — and the size limits are refused by name.  Device memory comes from the
HIP runtime the library itself is linked to (no torch in this process)."""
import ctypes as C

import numpy as np
import pytest

from gear_amd import descriptors as D
from gear_amd import engine
from gear_amd import workloads as W

pytestmark = pytest.mark.gpu

BOTH = D.GH_CHECK_ENV | D.GH_CHECK_SELF
GH_ERR_TOO_LARGE = -4


class DeviceMem:
    """hipMalloc'ed bytes, through the runtime libgear_hip.so is linked to."""

    def __init__(self, nbytes):
        L = engine.lib()
        self.hip = L
        for f in ("hipMalloc", "hipFree", "hipMemcpy", "hipDeviceSynchronize", "hipMemset"):
            getattr(L, f).restype = C.c_int
        L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        L.hipFree.argtypes = [C.c_void_p]
        L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.p = C.c_void_p()
        assert L.hipMalloc(C.byref(self.p), max(nbytes, 256)) == 0
        self.nbytes = nbytes

    def put(self, a):
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        assert self.hip.hipMemcpy(self.p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0

    def fill(self, byte, n):
        assert self.hip.hipMemset(self.p, byte, n) == 0

    def get(self, n, dtype=np.uint8):
        out = np.empty(n, dtype)
        assert self.hip.hipDeviceSynchronize() == 0
        assert self.hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), self.p, out.nbytes, 2) == 0
        return out

    def free(self):
        if self.p:
            self.hip.hipFree(self.p)
            self.p = None


def scene_of(name):
    from tests.conftest import load_golden
    from gear_amd import urdf
    if name.startswith("metric"):
        return W.metric_workload()
    if name.startswith("pr2"):
        return W.pr2_workload()
    return W.sample_workload(urdf.robot_from_dict(load_golden("sample_robot.json")),
                             urdf.robot_from_dict(load_golden("obstacles.json")), 0.01,
                             self_pairs=[("root", "l_wrist_yaw"), ("root", "l_wrist_pitch"), ("root", "l_gripper_linear1")])


@pytest.mark.parametrize("name,prec", [("metric", 32), ("metric", 64), ("sample_self", 32), ("pr2_mesh", 32)])
def test_noalloc_gives_the_bytes_of_check_batch_dev(name, prec):
    L = engine.lib()
    w, what = scene_of(name), BOTH
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    n1, n2 = (20000, 1500) if "mesh" in name else (40000, 2500) if prec == 64 else (65536, 2500)
    q1 = w.configs(n1, seed=71, dtype=hip.dtype, overshoot=0.01)
    q2 = w.configs(n2, seed=72, dtype=hip.dtype, overshoot=0.01)
    dq, df, dg = DeviceMem(q1.nbytes), DeviceMem(n1), DeviceMem(n1)
    try:
        for q in (q1, q2):                                  # the second call is the smaller one
            n = len(q)
            dq.put(q)
            df.fill(0xEE, n1)
            dg.fill(0xDD, n1)
            assert L.gh_check_batch_dev(hip.h, dq.p, n, what, df.p, None) == 0, L.gh_last_error()
            want = df.get(n1)
            assert L.gh_check_batch_noalloc(hip.h, dq.p, n, what, dg.p, None) == 0, L.gh_last_error()
            got = dg.get(n1)
            assert (got[:n] == want[:n]).all()
            assert (got[n:] == 0xDD).all() and (want[n:] == 0xEE).all()      # nothing beyond n is written
            assert (got[:n] == hip.check_batch(q, what)).all()
            assert (got[:n] & 3).any() and (got[:n] == 0).any()
        # limits only, no collision bits: the fused kernel without any workspace
        assert L.gh_check_batch_noalloc(hip.h, dq.p, n2, 0, dg.p, None) == 0
        assert (dg.get(n2) == hip.check_batch(q2, 0)).all()
        assert L.gh_check_batch_noalloc(hip.h, dq.p, 0, what, dg.p, None) == 0
    finally:
        for b in (dq, df, dg):
            b.free()


def test_noalloc_refuses_what_does_not_fit():
    L = engine.lib()
    w = W.metric_workload()
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32)
    d = DeviceMem(1 << 20)
    try:
        rc = L.gh_check_batch_noalloc(hip.h, d.p, (1 << 16) + 1, BOTH, d.p, None)      # refused before any use
        assert rc == GH_ERR_TOO_LARGE and b"at most 65536" in L.gh_last_error()
        assert L.gh_check_batch_noalloc(hip.h, None, 5, BOTH, d.p, None) == -1
    finally:
        d.free()
    # a mesh scene's exact regions are four times larger: 65536 configurations of it do not fit the array
    p = W.pr2_workload()
    hp = engine.Scene(p.robot_keep, p.obstacle_keep, p.margin, 64)
    n = 1 << 16
    q = p.configs(n, seed=73, dtype=np.float64)
    dq, df = DeviceMem(q.nbytes), DeviceMem(n)
    try:
        dq.put(q)
        rc = L.gh_check_batch_noalloc(hp.h, dq.p, n, BOTH, df.p, None)
        if rc != 0:
            assert rc == GH_ERR_TOO_LARGE and b"no-allocation workspace holds 67108864" in L.gh_last_error()
            assert L.gh_check_batch_dev(hp.h, dq.p, n, BOTH, df.p, None) == 0          # the allocating call takes it
            df.get(n)
    finally:
        dq.free()
        df.free()
