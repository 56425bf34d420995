"""The two evaluation paths of gh_check_batch — the fused kernel (small batches) and the
sweep / narrow-phase pipeline (large batches) — must return identical flag bytes, and the
pipeline must match the CPU oracle outside the parity band just like the fused path."""
import numpy as np
import pytest

from gear_amd import descriptors as D
from gear_amd import engine
from gear_amd import workloads as W
from oracle import binding as orc
from tests.parity import check_flags

pytestmark = pytest.mark.gpu

BOTH = D.GH_CHECK_ENV | D.GH_CHECK_SELF


def _workloads(sample_robot, obstacles_robot):
    pairs = [("root", "l_wrist_yaw"), ("root", "l_wrist_pitch"), ("l_shoulder_yaw", "l_wrist_pitch"),
             ("root", "l_gripper_linear1"), ("l_shoulder_pitch", "l_gripper_linear2")]
    return [W.sample_workload(sample_robot, obstacles_robot, 0.01, self_pairs=pairs), W.metric_workload(),
            W.ur10_workload(96), W.sawyer_workload(1024)]


@pytest.mark.parametrize("prec", [32, 64])
def test_pipeline_equals_fused(sample_robot, obstacles_robot, prec):
    for w in _workloads(sample_robot, obstacles_robot):
        hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
        n = 6000 if w.obstacle_keep.desc.n_shapes > 512 else 40000
        q = w.configs(n, seed=21, dtype=hip.dtype, overshoot=0.02)
        for what in (BOTH, D.GH_CHECK_ENV, D.GH_CHECK_SELF):
            a = hip.check_batch(q, what | D.GH_PATH_FUSED)
            b = hip.check_batch(q, what | D.GH_PATH_PIPELINE)                       # broadphase grid
            c = hip.check_batch(q, what | D.GH_PATH_PIPELINE | D.GH_PATH_NO_GRID)  # every obstacle swept
            assert (a == b).all(), (w.name, prec, what, int((a != b).sum()))
            assert (a == c).all(), (w.name, prec, what, int((a != c).sum()))
        assert ((a | b) != 0).any()
        # ragged sizes around the wavefront width, through the pipeline
        for m in (1, 63, 64, 65, 129):
            assert (hip.check_batch(q[:m], BOTH | D.GH_PATH_PIPELINE) ==
                    hip.check_batch(q[:m], BOTH | D.GH_PATH_FUSED)).all()


@pytest.mark.parametrize("prec", [32, 64])
def test_pipeline_matches_oracle(prec):
    w = W.metric_workload()
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    ora = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    q = w.configs(3000, seed=22, dtype=hip.dtype, overshoot=0.01)
    ora64 = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, 64) if prec == 32 else None
    check_flags(f"pipeline_matches_oracle[f{prec}]", hip, ora, q, BOTH | D.GH_PATH_PIPELINE, prec, ora64)


def test_pipeline_queue_overflow_is_rechecked():
    """A scene so dense that a wave's survivor region (1024 items) overflows: those waves are
    re-checked by the fused kernel, so the flags still equal the fused path's."""
    w = W.Workload("dense", W.arm7_robot(), W.obstacle_field(2048, 3, ("box", "ball"), (0.5, 0.5), cube=1.6),
                   0.01, W.ARM7_SELF_PAIRS)
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32)
    q = w.configs(4096, seed=23)
    a = hip.check_batch(q, BOTH | D.GH_PATH_FUSED)
    b = hip.check_batch(q, BOTH | D.GH_PATH_PIPELINE)
    c = hip.check_batch(q, BOTH | D.GH_PATH_PIPELINE | D.GH_PATH_NO_GRID)
    assert (a == b).all() and (a == c).all()


@pytest.mark.parametrize("prec", [32, 64])
def test_small_batches_default_path_equals_fused(prec):
    """Scenes with a broadphase grid send every batch size through the pipeline (an RRT line is a few
    hundred samples): tiny and ragged batch sizes must give the fused kernel's flags, in every mode."""
    for w in (W.metric_workload(), W.pr2_workload(64)):
        hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
        q = w.configs(5000, seed=77, dtype=hip.dtype, overshoot=0.02)
        for n in (1, 2, 63, 64, 65, 300, 5000):
            for what in (BOTH, D.GH_CHECK_ENV, D.GH_CHECK_SELF):
                got = hip.check_batch(q[:n], what)
                assert (got == hip.check_batch(q[:n], what | D.GH_PATH_FUSED)).all(), (w.name, n, what)
        ora = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, prec)
        ora64 = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, 64) if prec == 32 else None
        check_flags(f"small_batches_default_path[{w.name},f{prec}]", hip, ora, q[:300], BOTH, prec, ora64)


@pytest.mark.parametrize("prec", [32, 64])
def test_in_place_host_calls_equal_copied_ones(prec):
    """Host-buffer calls of <= 4096 configurations run in place on the scene's pinned staging buffers;
    larger ones copy.  Both sides of the limit must give the same flags and first-bad indices."""
    w = W.metric_workload()
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    q = w.configs(9000, seed=91, dtype=hip.dtype, overshoot=0.02)
    ref = hip.check_batch(q, BOTH)                      # copied (9000 > 4096)
    for n in (1, 100, 4095, 4096, 4097):
        assert (hip.check_batch(q[:n], BOTH) == ref[:n]).all(), n
    bad = D.GH_FLAG_ENV | D.GH_FLAG_LIMIT
    for n, step in ((4096, 64), (4000, 37), (8192, 64)):
        off = np.arange(0, n + 1, step, dtype=np.uint32)
        if off[-1] != n:
            off = np.append(off, np.uint32(n))
        flags, first = hip.check_edges(q[:n], off, D.GH_CHECK_ENV, bad)
        want = hip.check_batch(q[:n], D.GH_CHECK_ENV | D.GH_PATH_FUSED)
        assert (flags == want).all(), n
        for e in range(len(off) - 1):
            seg = np.nonzero(want[off[e]:off[e + 1]] & bad)[0]
            assert first[e] == (seg[0] if len(seg) else -1), (n, e)


def test_flag_bytes_land_at_any_alignment_and_nowhere_else():
    """The exact stage writes the flag bytes (eight per store when the caller's array allows it): device arrays that
    start at every byte offset, ragged counts, and the bytes either side of the array stay as they were."""
    import ctypes as C
    from tests.test_zz_no_alloc import DeviceMem
    w = W.metric_workload()
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32)
    n = 70000 + 37
    q = w.configs(n, seed=29, dtype=hip.dtype, overshoot=0.01)
    want = hip.check_batch(q, BOTH | D.GH_PATH_FUSED)
    dq, buf = DeviceMem(q.nbytes), DeviceMem(n + 64)
    dq.put(q)
    try:
        for shift in range(9):
            for m in (n, 65536 + 5, 129):
                buf.fill(0xAB, n + 64)
                out = C.c_void_p(buf.p.value + 8 + shift)
                rc = engine.lib().gh_check_batch_dev(hip.h, dq.p, m, BOTH | D.GH_PATH_PIPELINE, out, None)
                assert rc == 0, engine.lib().gh_last_error()
                got = buf.get(n + 64)
                assert (got[8 + shift: 8 + shift + m] == want[:m]).all(), (shift, m)
                assert (got[:8 + shift] == 0xAB).all() and (got[8 + shift + m:] == 0xAB).all(), (shift, m)
    finally:
        dq.free()
        buf.free()
