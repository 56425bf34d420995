"""gh_scene_create_multi: a group of devices shares one batch (contiguous shards, whole edges, results in
the caller's buffers).  The GPU box has one device, so the group lists it several times: every entry is a
share with its own stream, staging buffers and workspace, which is what runs on distinct devices too."""
import numpy as np
import pytest

from gear_amd import descriptors as D
from gear_amd import engine
from gear_amd import workloads as W

pytestmark = pytest.mark.gpu

BOTH = D.GH_CHECK_ENV | D.GH_CHECK_SELF


@pytest.mark.parametrize("prec", [32, 64])
def test_group_matches_single_device(prec):
    w = W.metric_workload()
    one = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    bad = D.GH_FLAG_ENV | D.GH_FLAG_LIMIT
    for devs in ([0, 0], [0, 0, 0]):
        grp = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec, devices=devs)
        assert engine.lib().gh_scene_device_count(grp.h) == len(devs) and engine.lib().gh_scene_device_count(one.h) == 1
        for n in (0, 1, 2, 4097, 100001):
            q = w.configs(n, seed=51, dtype=one.dtype, overshoot=0.01)
            assert (grp.check_batch(q, BOTH) == one.check_batch(q, BOTH)).all(), (devs, n)
        q = w.configs(3000, seed=52, dtype=one.dtype)
        assert (grp.fk_batch(q) == one.fk_batch(q)).all()
        # edges: ragged, some empty, fewer edges than shares, many edges
        for n_edges, seed in ((1, 1), (2, 2), (400, 3)):
            qe, off = w.edges(n_edges, seed=seed, dtype=one.dtype, min_len=3, max_len=50)
            off = np.concatenate([off[:1], off[:1], off[1:], off[-1:]]).astype(np.uint32)   # an empty edge at either end
            f1, b1 = one.check_edges(qe, off, D.GH_CHECK_ENV, bad)
            f2, b2 = grp.check_edges(qe, off, D.GH_CHECK_ENV, bad)
            assert (f1 == f2).all() and (b1 == b2).all(), (devs, n_edges)
        # the diagnostics are served by the first device
        assert (grp.link_hits(q[:50]) == one.link_hits(q[:50])).all()
        assert (grp.self_hits(q[:50]) == one.self_hits(q[:50])).all()


def test_group_rejects_device_pointer_calls_and_bad_lists():
    w = W.metric_workload(64)
    grp = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32, devices=[0, 0])
    # a group has no single device: the device-pointer entry points refuse it before they look at the pointers
    q = np.zeros((64, grp.dof), np.float32)
    f = np.zeros(64, np.uint8)
    rc = engine.lib().gh_check_batch_dev(grp.h, q.ctypes.data, 64, BOTH, f.ctypes.data, None)
    assert rc == -1 and b"single-device" in engine.lib().gh_last_error()
    with pytest.raises(engine.GearHipError):
        engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32, devices=[0, 99])
    with pytest.raises(engine.GearHipError):
        engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32, devices=[])


def test_a_failing_share_is_reported_and_nothing_hangs():
    """One share of a group call fails (a test hook makes share 1 throw): the call returns that share's error with its
    device named, the other shares have finished (no thread is left behind, the call returns), and the group, a second
    group and a plain scene all work afterwards — the per-device pools and locks were released."""
    import os
    w = W.metric_workload(64)
    q = w.configs(300000, seed=12, overshoot=0.01)          # large enough to be cut into shares
    one = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32)
    want = one.check_batch(q, BOTH)
    os.environ["GEAR_HIP_TEST_FAIL_PART"] = "1"
    try:
        bad = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32, devices=[0, 0, 0])
    finally:
        del os.environ["GEAR_HIP_TEST_FAIL_PART"]
    for _ in range(3):
        with pytest.raises(engine.GearHipError) as e:
            bad.check_batch(q, BOTH)
        assert "device 0" in str(e.value) and "told to fail" in str(e.value)
    off = np.arange(0, 300001, 50, dtype=np.uint32)
    with pytest.raises(engine.GearHipError):
        bad.check_edges(q, off)
    # a call small enough for one device's share never reaches share 1
    assert (bad.check_batch(q[:1000], BOTH) == want[:1000]).all()
    good = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32, devices=[0, 0, 0])
    assert (good.check_batch(q, BOTH) == want).all()
    assert (one.check_batch(q, BOTH) == want).all()
