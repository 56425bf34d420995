"""gh_check_batch_noalloc (include/gear_hip.h): the same bytes as gh_check_batch_dev, through the C ABI, on
two calls in a row — the second with a smaller input that still takes the workspace-using pipeline, so that
what the first call left in the shared workspace array is still there — on every evaluation path (default,
pipeline, fused), on a scene whose pose columns live in the workspace too, with no device memory allocated
by the call (hipMemGetInfo), with two scenes called back to back on their own streams, and with the size
limits refused by name.  Device memory comes from the HIP runtime the library itself is linked to (no torch
in this process)."""
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
        for f in ("hipMalloc", "hipFree", "hipMemcpy", "hipDeviceSynchronize", "hipMemset", "hipMemGetInfo"):
            getattr(L, f).restype = C.c_int
        L.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
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


def free_bytes():
    L = engine.lib()
    assert L.hipDeviceSynchronize() == 0
    free, total = C.c_size_t(), C.c_size_t()
    assert L.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


# (n1, n2): the second call is the smaller one and still larger than the small-batch kernel's limit (16384
# configurations in f32, 6144 in f64), so that it runs the pipeline on the workspace the first call left behind.
# pr2_mesh in f64: 20 pose columns of 64 doubles exceed the 36 KB LDS budget of the fused stage, so the pose
# columns are carved from the workspace array as well.
CASES = [("metric", 32, 65536, 20000), ("metric", 64, 40000, 9000), ("sample_self", 32, 65536, 17000),
         ("pr2_mesh", 32, 20000, 16500), ("pr2_mesh", 64, 12000, 7000)]


@pytest.mark.parametrize("name,prec,n1,n2", CASES)
def test_noalloc_gives_the_bytes_of_check_batch_dev(name, prec, n1, n2):
    L = engine.lib()
    w, what = scene_of(name), BOTH
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    q1 = w.configs(n1, seed=71, dtype=hip.dtype, overshoot=0.01)
    q2 = w.configs(n2, seed=72, dtype=hip.dtype, overshoot=0.01)
    dq, df, dg = DeviceMem(q1.nbytes), DeviceMem(n1), DeviceMem(n1)
    try:
        for q in (q1, q2):                                  # the second call is the smaller one
            n = len(q)
            dq.put(q)
            ref = hip.check_batch(q, what)
            assert (ref & 3).any() and (ref == 0).any()
            for path in (0, D.GH_PATH_PIPELINE, D.GH_PATH_FUSED):
                df.fill(0xEE, n1)
                dg.fill(0xDD, n1)
                assert L.gh_check_batch_dev(hip.h, dq.p, n, what | path, df.p, None) == 0, L.gh_last_error()
                want = df.get(n1)
                assert L.gh_check_batch_noalloc(hip.h, dq.p, n, what | path, dg.p, None) == 0, L.gh_last_error()
                got = dg.get(n1)
                assert (got[:n] == want[:n]).all(), (name, prec, n, path)
                assert (got[n:] == 0xDD).all() and (want[n:] == 0xEE).all()      # nothing beyond n is written
                assert (got[:n] == ref).all(), (name, prec, n, path)
        # a small batch (the single-launch kernel: no workspace at all) and limits only (the fused kernel)
        assert L.gh_check_batch_noalloc(hip.h, dq.p, 1500, what, dg.p, None) == 0
        assert (dg.get(1500) == hip.check_batch(q2[:1500], what)).all()
        assert L.gh_check_batch_noalloc(hip.h, dq.p, n2, 0, dg.p, None) == 0
        assert (dg.get(n2) == hip.check_batch(q2, 0)).all()
        assert L.gh_check_batch_noalloc(hip.h, dq.p, 0, what, dg.p, None) == 0
    finally:
        for b in (dq, df, dg):
            b.free()


@pytest.mark.parametrize("name,prec,n1,n2", [("metric", 32, 65536, 20000), ("pr2_mesh", 64, 12000, 7000)])
def test_noalloc_first_on_a_fresh_scene_allocates_nothing(name, prec, n1, n2):
    """The no-allocation call comes FIRST on a scene that no allocating entry point has touched (when this file
    runs alone, the device's workspace pool does not exist at all), and the device's free memory is the same
    before and after it.  The reference bytes come from a second scene afterwards."""
    import gc
    gc.collect()                                            # scenes of earlier tests release the device's pool
    L = engine.lib()
    w = scene_of(name)
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    dtype = np.float32 if prec == 32 else np.float64
    q1 = w.configs(n1, seed=75, dtype=dtype, overshoot=0.01)
    q2 = w.configs(n2, seed=76, dtype=dtype, overshoot=0.01)
    dq, dg = DeviceMem(q1.nbytes), DeviceMem(n1)
    try:
        # the first launch of a kernel may make the runtime set up its per-queue scratch (register spills): one
        # warm-up call of the smallest pipeline size, then the measured calls, the first of them larger
        dq.put(q2)
        assert L.gh_check_batch_noalloc(hip.h, dq.p, 6200 if prec == 64 else 16500, BOTH | D.GH_PATH_PIPELINE, dg.p, None) == 0, L.gh_last_error()
        got = []
        before = free_bytes()
        for q in (q1, q2):
            dq.put(q)
            dg.fill(0xDD, n1)
            assert L.gh_check_batch_noalloc(hip.h, dq.p, len(q), BOTH | D.GH_PATH_PIPELINE, dg.p, None) == 0, L.gh_last_error()
            got.append(dg.get(len(q)))
        after = free_bytes()
        assert after == before, f"device free memory changed by {before - after} bytes across gh_check_batch_noalloc"
        other = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
        for q, g in zip((q1, q2), got):
            assert (g == other.check_batch(q, BOTH)).all()
    finally:
        dq.free()
        dg.free()


def test_noalloc_two_scenes_back_to_back_on_their_own_streams():
    """stream = NULL launches on each scene's own stream; every scene of a GPU shares that GPU's array.  The
    library orders the calls (an event per device): two scenes called back to back, with nothing in between,
    return what gh_check_batch_dev returns."""
    L = engine.lib()
    wa, wb = W.metric_workload(), W.metric_workload(256, seed=9)
    a = engine.Scene(wa.robot_keep, wa.obstacle_keep, wa.margin, 32)
    b = engine.Scene(wb.robot_keep, wb.obstacle_keep, wb.margin, 32)
    n = 65536
    q = wa.configs(n, seed=77, dtype=np.float32, overshoot=0.01)
    dq, fa, fb = DeviceMem(q.nbytes), DeviceMem(n), DeviceMem(n)
    try:
        dq.put(q)
        want_a, want_b = a.check_batch(q, BOTH), b.check_batch(q, BOTH)
        assert (want_a != want_b).any()
        for rep in range(6):
            fa.fill(0xDD, n)
            fb.fill(0xDD, n)
            for k in range(3):                               # A, B, A, B, ... with no synchronisation in between
                assert L.gh_check_batch_noalloc(a.h, dq.p, n, BOTH | D.GH_PATH_PIPELINE, fa.p, None) == 0
                assert L.gh_check_batch_noalloc(b.h, dq.p, n, BOTH | D.GH_PATH_PIPELINE, fb.p, None) == 0
            assert (fa.get(n) == want_a).all() and (fb.get(n) == want_b).all(), rep
    finally:
        for m in (dq, fa, fb):
            m.free()


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
