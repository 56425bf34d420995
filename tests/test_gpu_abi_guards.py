"""Guards at the C ABI (round-2 fixes): undeclared `what` bits cannot change a result, a scene shared
by host threads gives the single-threaded bytes, GH_CHECK_NO_LIMITS evaluates geometry only, and the
host layer's scene cache follows a compound's contents, not its address."""
import copy
import threading

import numpy as np
import pytest

from gear_amd import descriptors as D
from gear_amd import engine, urdf
from gear_amd import host_binding as H
from gear_amd import workloads as W
from oracle import binding as orc
from tests.conftest import load_golden
from tests.parity import compare_flags, record

pytestmark = pytest.mark.gpu

BOTH = D.GH_CHECK_ENV | D.GH_CHECK_SELF
# every bit the round-1 library used for profiling ablations (skip the exact stage, drop stage-1 items,
# no push, no sweep, wave-uniform records, one item per wave, ...), plus some never used at all
STRAY = 0x100 | 0x200 | 0x400 | 0x800 | 0x10000 | 0x20000 | 0x40000 | 0x200000 | 0x400000 | 0x1000000 | 0x80000000


@pytest.fixture(scope="module")
def metric():
    w = W.metric_workload()
    return w, engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32)


@pytest.mark.parametrize("path", [0, D.GH_PATH_FUSED, D.GH_PATH_PIPELINE, D.GH_PATH_PIPELINE | D.GH_PATH_NO_GRID])
def test_undeclared_what_bits_change_nothing(metric, path):
    w, hip = metric
    q = w.configs(20000, seed=41, overshoot=0.01)
    want = hip.check_batch(q, BOTH | path)
    assert (want & 3).any() and (want == 0).any()
    for stray in (0x10700, STRAY):
        got = hip.check_batch(q, BOTH | path | stray)
        assert (got == want).all(), hex(stray)
    # the edge entry points and the small in-place path mask the same way
    qe, off = w.edges(40, seed=42, min_len=4, max_len=30)
    f0, b0 = hip.check_edges(qe, off, D.GH_CHECK_ENV | path, D.GH_FLAG_ENV | D.GH_FLAG_LIMIT)
    f1, b1 = hip.check_edges(qe, off, D.GH_CHECK_ENV | path | STRAY, D.GH_FLAG_ENV | D.GH_FLAG_LIMIT)
    assert (f0 == f1).all() and (b0 == b1).all()


def test_two_threads_share_one_scene(metric):
    """include/gear_hip.h: a scene may be shared by host threads.  Thread A keeps calling gh_check_edges on
    the copy path (> 4096 samples: staging buffers, launch, read-back), thread B calls gh_check_batch
    with other sizes (which reallocates and overwrites the same scene-owned buffers) — every result must
    equal the single-threaded one."""
    w, hip = metric
    qe, off = w.edges(400, seed=43, min_len=10, max_len=40)
    assert len(qe) > 4096
    qa = w.configs(30000, seed=44, overshoot=0.01)
    qb = w.configs(777, seed=45, overshoot=0.01)
    bad = D.GH_FLAG_ENV | D.GH_FLAG_LIMIT
    want_e = hip.check_edges(qe, off, D.GH_CHECK_ENV, bad)
    want_a, want_b = hip.check_batch(qa), hip.check_batch(qb)
    errors = []

    def edges_loop():
        for _ in range(12):
            f, b = hip.check_edges(qe, off, D.GH_CHECK_ENV, bad)
            if not ((f == want_e[0]).all() and (b == want_e[1]).all()):
                errors.append("edges")

    def batch_loop():
        for k in range(12):
            q, want = (qa, want_a) if k % 2 else (qb, want_b)
            if not (hip.check_batch(q) == want).all():
                errors.append("batch")

    ts = [threading.Thread(target=edges_loop), threading.Thread(target=batch_loop), threading.Thread(target=edges_loop)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("prec", [32, 64])
def test_no_limits_bit_matches_oracle(sample_robot, obstacles_robot, prec):
    """GH_CHECK_NO_LIMITS = has_any_colliding / has_any_colliding_with_self on a state set without the
    limit check (joint_path_planner.rs:86-98,123-128): no GH_FLAG_LIMIT, collisions evaluated everywhere."""
    w = W.sample_workload(sample_robot, obstacles_robot, 0.01,
                          self_pairs=[("root", "l_wrist_yaw"), ("root", "l_wrist_pitch"), ("root", "l_gripper_linear1")])
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    ora = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    q = w.configs(6000, seed=46, dtype=hip.dtype, overshoot=0.3)
    gated = hip.check_batch(q, BOTH)
    assert ((gated & D.GH_FLAG_LIMIT) != 0).mean() > 0.2
    for path in (D.GH_PATH_FUSED, D.GH_PATH_PIPELINE):
        got = hip.check_batch(q, BOTH | D.GH_CHECK_NO_LIMITS | path)
        want, band = ora.check(q, BOTH | D.GH_CHECK_NO_LIMITS, threads=16)
        rep = compare_flags(got, want, band, prec)
        record(f"no_limits_bit[f{prec},path={path:#x}]", rep)
        assert rep["outside"] == 0, rep
        assert (got & D.GH_FLAG_LIMIT).max() == 0
        inside = (gated & D.GH_FLAG_LIMIT) == 0
        assert (got[inside] == gated[inside]).all()          # inside the limits the bit changes nothing
        assert (got[~inside] != 0).any()                     # outside them collisions are now reported
    # the diagnostics take the same bit
    ql = q[:300]
    _, bl = ora.check(ql, BOTH | D.GH_CHECK_NO_LIMITS, threads=8)
    safe = bl[:, 0] > (2e-5 if prec == 32 else 1e-9)
    assert (hip.link_hits(ql, D.GH_CHECK_NO_LIMITS)[safe] == ora.link_hits(ql, D.GH_CHECK_NO_LIMITS)[safe]).all()
    assert (hip.self_hits(ql, D.GH_CHECK_NO_LIMITS)[safe] == ora.self_hits(ql, D.GH_CHECK_NO_LIMITS)[safe]).all()
    assert (hip.link_hits(ql)[(gated[:300] & 4) != 0] == 0).all()      # without it: all-zero masks on a violation


def test_has_any_colliding_ignores_limits(tmp_path, obstacles_robot):
    """A joint whose range excludes its never-set position 0: the reference's has_any_colliding and
    colliding_link_names never look at limits (joint_path_planner.rs:86-110), so the wall in front of the
    robot is reported; is_feasible at that position is false through the limit (:77-82)."""
    d = copy.deepcopy(load_golden("sample_robot.json"))
    for j in d["joints"]:
        if j["name"] == "l_elbow_pitch":
            j["lower"], j["upper"] = 0.5, 1.0            # excludes the default position 0
    robot = urdf.robot_from_dict(d)
    p = tmp_path / "limited.urdf"
    p.write_text(urdf.robot_to_xml(robot))
    pl = H.HostPlanner(str(p), 0.01, 32, self_pairs=["root:l_wrist_yaw"])
    g = load_golden("collision_check.json")
    wall = H.HostCompound.from_specs([D.ShapeSpec.box(g["target_half_extents"], t=g["cases"][0]["target_xyz"])])
    assert pl.has_any_colliding(wall)
    assert pl.colliding_link_names(wall) == g["cases"][0]["names"]      # the q = 0 case of the golden lists
    u = pl.using_joints()
    assert pl.is_feasible(u, np.zeros(len(u)), wall) is False
    far = H.HostCompound.from_specs([D.ShapeSpec.box((0.1, 0.1, 0.1), t=(5.0, 5.0, 5.0))])
    assert not pl.has_any_colliding(far) and pl.colliding_link_names(far) == []
    pl.has_any_colliding_with_self()                                     # runs on the out-of-range state too


def test_scene_cache_follows_compound_contents(tmp_path, sample_robot):
    """a compound destroyed and rebuilt — same shape count, likely the same heap
    address — must not be served the previous compound's GPU scene."""
    p = tmp_path / "sample.urdf"
    p.write_text(urdf.robot_to_xml(sample_robot))
    pl = H.HostPlanner(str(p), 0.01, 32)
    u = pl.using_joints()
    g = load_golden("collision_check.json")
    q = np.array(g["cases"][0]["q"], np.float64)
    for k in range(6):
        near = H.HostCompound.from_specs([D.ShapeSpec.box(g["target_half_extents"], t=g["cases"][0]["target_xyz"])])
        assert pl.is_feasible(u, q, near) is False, k
        assert pl.colliding_link_names(near) == g["cases"][0]["names"]
        del near                                                           # gp_compound_destroy
        far = H.HostCompound.from_specs([D.ShapeSpec.box(g["target_half_extents"], t=(9.0, 9.0, 9.0))])
        assert pl.is_feasible(u, q, far) is True, k
        assert pl.colliding_link_names(far) == []
        del far


def test_scenes_share_the_device_workspace():
    """The pipeline's workspace belongs to the device, not to a scene: calls on different scenes — other
    robots, other precisions, other sizes, from different host threads — take the same sets one after the
    other and each still gets its own bytes."""
    cases = []
    for w, prec, n in ((W.metric_workload(), 32, 90000), (W.pr2_workload(64), 64, 30000), (W.ur10_workload(96), 32, 50000)):
        hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
        q = w.configs(n, seed=47, dtype=hip.dtype, overshoot=0.01)
        cases.append((hip, q, hip.check_batch(q, BOTH | D.GH_PATH_FUSED)))
    errors = []

    def loop(k):
        hip, q, want = cases[k]
        for r in range(6):
            m = len(q) - 997 * r                                  # sizes change, so the sets are regrown and reused
            if not (hip.check_batch(q[:m], BOTH | D.GH_PATH_PIPELINE) == want[:m]).all():
                errors.append((k, r))

    ts = [threading.Thread(target=loop, args=(k,)) for k in range(len(cases))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    # destroying every scene of the device frees the pool; the next scene starts a new one
    for hip, _, _ in cases:
        hip.close()
    w = W.metric_workload()
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32)
    q = w.configs(20000, seed=48, overshoot=0.01)
    assert (hip.check_batch(q, BOTH | D.GH_PATH_PIPELINE) == hip.check_batch(q, BOTH | D.GH_PATH_FUSED)).all()
