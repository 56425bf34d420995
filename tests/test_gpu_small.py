"""k_small (gear_amd/csrc/gear_small.hpp), the single-launch path of small batches — the RRT lines of
joint_path_planner.rs:168-188: flag bytes identical to the fused kernel's and to the pipeline's at every
size and in every mode, independent of its item queue's capacity, and equal to the oracle's outside the
parity band."""
import os

import numpy as np
import pytest

from gear_amd import descriptors as D
from gear_amd import engine
from gear_amd import workloads as W
from oracle import binding as orc
from tests.parity import check_flags

pytestmark = pytest.mark.gpu

BOTH = D.GH_CHECK_ENV | D.GH_CHECK_SELF
SMALL, FUSED, PIPE = D.GH_PATH_SMALL, D.GH_PATH_FUSED, D.GH_PATH_PIPELINE


def _workloads(sample_robot, obstacles_robot):
    pairs = [("root", "l_wrist_yaw"), ("root", "l_wrist_pitch"), ("l_shoulder_yaw", "l_wrist_pitch"),
             ("root", "l_gripper_linear1"), ("l_shoulder_pitch", "l_gripper_linear2")]
    return [W.sample_workload(sample_robot, obstacles_robot, 0.01, self_pairs=pairs), W.metric_workload(),
            W.ur10_workload(96), W.pr2_workload(64), W.sawyer_workload(1024)]


@pytest.mark.parametrize("prec", [32, 64])
def test_small_equals_fused_and_pipeline(sample_robot, obstacles_robot, prec):
    for w in _workloads(sample_robot, obstacles_robot):
        hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
        n = 3000 if w.obstacle_keep.desc.n_shapes > 512 else 12000
        q = w.configs(n, seed=81, dtype=hip.dtype, overshoot=0.02)
        for what in (BOTH, D.GH_CHECK_ENV, D.GH_CHECK_SELF, BOTH | D.GH_CHECK_NO_LIMITS):
            a = hip.check_batch(q, what | FUSED)
            b = hip.check_batch(q, what | SMALL)                       # forced beyond the default size limit
            c = hip.check_batch(q, what | SMALL | D.GH_PATH_NO_GRID)  # every obstacle listed
            assert (a == b).all(), (w.name, prec, what, int((a != b).sum()))
            assert (a == c).all(), (w.name, prec, what, int((a != c).sum()))
            assert (hip.check_batch(q, what | PIPE) == b).all()
        assert (a != 0).any() and (a == 0).any()
        # ragged sizes around the workgroup's share of configurations, on the default path (k_small)
        for m in (1, 2, 3, 14, 15, 27, 28, 29, 63, 64, 65, 257, 1000):
            assert (hip.check_batch(q[:m], BOTH) == hip.check_batch(q[:m], BOTH | FUSED)).all(), (w.name, m)


@pytest.mark.parametrize("qcap", ["1", "7", "64"])
def test_small_queue_capacity_never_changes_a_result(qcap):
    """An item that finds the workgroup's queue full is decided by the lane that found it."""
    os.environ["GEAR_HIP_SMALL_QCAP"] = qcap                   # read once, at scene creation
    try:
        scenes = [(w, engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32))
                  for w in (W.metric_workload(), W.pr2_workload(64))]
    finally:
        del os.environ["GEAR_HIP_SMALL_QCAP"]
    for w, hip in scenes:
        q = w.configs(4000, seed=82, overshoot=0.02)
        for what in (BOTH, D.GH_CHECK_ENV, D.GH_CHECK_SELF):
            assert (hip.check_batch(q, what | SMALL) == hip.check_batch(q, what | FUSED)).all(), (w.name, what)


@pytest.mark.parametrize("prec", [32, 64])
def test_small_matches_oracle(prec):
    for w in (W.metric_workload(), W.pr2_workload(64)):
        hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
        ora = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, prec)
        ora64 = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, 64) if prec == 32 else None
        q = w.configs(2500, seed=83, dtype=hip.dtype, overshoot=0.01)
        check_flags(f"small_matches_oracle[{w.name},f{prec}]", hip, ora, q, BOTH | SMALL, prec, ora64)


@pytest.mark.parametrize("prec", [32, 64])
def test_small_edges_first_bad(prec):
    """gh_check_edges on RRT-sized calls: k_small with the first-bad reduction in the same launch (a wavefront
    per edge for long edges, a lane per edge for short ones), the separate reduction beyond 256 edges, and the
    copy path beyond 4096 samples — against the fused kernel followed by the separate reduction."""
    w = W.metric_workload()
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    bad = D.GH_FLAG_ENV | D.GH_FLAG_LIMIT
    seen = set()
    for n_edges, lo, hi, seed in ((1, 1, 1, 1), (1, 200, 200, 2), (7, 20, 90, 3), (60, 1, 30, 4), (200, 1, 6, 5),
                                  (300, 1, 12, 6), (250, 17, 40, 7), (3, 0, 2, 8)):
        qe, off = w.edges(n_edges, seed=seed, min_len=max(lo, 1), max_len=max(hi, 1))
        qe = qe.astype(hip.dtype)
        for what in (D.GH_CHECK_ENV, BOTH):
            f0, b0 = hip.check_edges(qe, off, what | FUSED, bad)
            f1, b1 = hip.check_edges(qe, off, what, bad)
            assert (f0 == f1).all() and (b0 == b1).all(), (n_edges, lo, hi)
            f2, b2 = hip.check_edges(qe, off, what, bad)                    # the ticket counter is back at zero
            assert (f0 == f2).all() and (b0 == b2).all()
        seen.add((len(qe) > 4096, n_edges > 256, len(qe) >= 16 * n_edges))
        assert (b0 >= 0).any() or n_edges < 5
    assert len(seen) >= 4, seen
    # empty edges between full ones
    qe, off = w.edges(6, seed=9, min_len=5, max_len=9)
    off2 = np.array([0, off[1], off[1], off[2], off[2], off[2], off[3], off[4], off[5], off[6], off[6]], np.uint32)
    f0, b0 = hip.check_edges(qe.astype(hip.dtype), off2, D.GH_CHECK_ENV | FUSED, bad)
    f1, b1 = hip.check_edges(qe.astype(hip.dtype), off2, D.GH_CHECK_ENV, bad)
    assert (f0 == f1).all() and (b0 == b1).all() and (b1[[1, 3, 4, 9]] == -1).all()


@pytest.mark.parametrize("prec", [32, 64])
def test_small_host_calls_report_complete_results(prec):
    """Host-buffer calls of up to 4096 samples do not wait for the stream: the kernel's last workgroup stores the
    call's sequence number in a host word once every flag byte and first-bad index is in host memory, and the
    caller polls that word.  Thousands of calls of changing size and content, each against bytes computed
    beforehand through the fused path (which synchronises the stream): a result read too early would show the
    previous call's bytes."""
    w = W.metric_workload()
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    rng = np.random.default_rng(91)
    bad = D.GH_FLAG_ENV | D.GH_FLAG_LIMIT
    pool = []
    for k in range(24):
        n = int(rng.choice([1, 2, 5, 33, 64, 200, 777, 1500, 4096]))
        q = w.configs(n, seed=900 + k, dtype=hip.dtype, overshoot=0.02)
        qe, off = w.edges(int(rng.integers(1, 40)), seed=950 + k, min_len=1, max_len=40)
        qe = qe.astype(hip.dtype)
        pool.append((q, hip.check_batch(q, BOTH | FUSED), qe, off, hip.check_edges(qe, off, D.GH_CHECK_ENV | FUSED, bad)))
    for it in range(2500):
        q, want, qe, off, (wf, wb) = pool[int(rng.integers(len(pool)))]
        if it % 2:
            assert (hip.check_batch(q, BOTH) == want).all(), it
        else:
            f, b = hip.check_edges(qe, off, D.GH_CHECK_ENV, bad)
            assert (f == wf).all() and (b == wb).all(), it
