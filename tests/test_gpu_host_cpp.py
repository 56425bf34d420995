"""The C++ host layer on the GPU: gear's planner API (JointPathPlanner predicates, plan,
plan_avoid_self_collision) over the batched engine, checked against the reference's golden lists,
the Python mirror, and the CPU oracle; speculative line evaluation must not change a plan."""
import numpy as np
import pytest

from gear_amd import descriptors as D
from gear_amd import host_binding as H
from gear_amd import planner as P
from gear_amd import urdf
from gear_amd import workloads as W
from oracle import binding as orc
from tests.parity import certain

pytestmark = pytest.mark.gpu

PAIRS = ["root:l_wrist_yaw", "root:l_wrist_pitch", "l_shoulder_yaw:l_wrist_pitch", "root:l_gripper_linear1",
         "l_shoulder_pitch:l_gripper_linear2"]


@pytest.fixture(scope="module")
def paths(tmp_path_factory, sample_robot, obstacles_robot):
    d = tmp_path_factory.mktemp("urdf")
    (d / "sample.urdf").write_text(urdf.robot_to_xml(sample_robot))
    (d / "obstacles.urdf").write_text(urdf.robot_to_xml(obstacles_robot))
    return str(d / "sample.urdf"), str(d / "obstacles.urdf")


def test_golden_collision_check_lists(paths, collision_check_golden):
    """src/planner/joint_path_planner.rs:373-419 through the C++ JointPathPlanner<f32>."""
    g = collision_check_golden
    pl = H.HostPlanner(paths[0], g["margin"], 32)
    u = pl.using_joints()
    for case in g["cases"]:
        target = H.HostCompound.from_specs([D.ShapeSpec.box(g["target_half_extents"], t=case["target_xyz"])])
        assert pl.is_feasible(u, case["q"], target) is False
        assert pl.colliding_link_names(target) == case["names"]
        assert pl.has_any_colliding(target)


@pytest.mark.parametrize("prec", [32, 64])
def test_predicates_match_python_mirror(paths, sample_robot, obstacles_robot, prec):
    pl = H.HostPlanner(paths[0], 0.01, prec, self_pairs=PAIRS)
    obs = H.HostCompound.from_urdf_file(paths[1])
    assert len(obs) == 2
    py = P.JointPathPlannerBuilder.from_urdf_robot(sample_robot).collision_check_margin(0.01) \
        .self_collision_pairs(D.parse_colon_separated_pairs(PAIRS)).precision(prec).finalize()
    objects = D.Compound.from_urdf_robot(obstacles_robot)
    w = W.sample_workload(sample_robot, obstacles_robot, 0.01)
    for end in (None, "l_tool_fixed"):
        u = pl.using_joints(end)
        chain = urdf.Chain.from_urdf(sample_robot)
        q = W.random_configs(chain, u, 3000, seed=31, dtype=np.float64, overshoot=0.02)
        if prec == 32:
            q = q.astype(np.float32).astype(np.float64)
        for with_self in (False, True):
            got = pl.is_feasible_batch(u, q, obs, with_self)
            want = py.is_feasible_batch(u, q.astype(py._scene(u, objects).dtype), objects, with_self)
            assert (got == want).all()
            assert got.any() and not got.all()
    # the single-configuration predicates and the state they leave behind
    u = pl.using_joints()
    q = w.configs(40, seed=32, dtype=np.float64)
    for row in q:
        assert pl.is_feasible(u, row, obs) == py.is_feasible(u, row, objects)
        assert pl.colliding_link_names(obs) == py.colliding_link_names(objects)
        assert pl.is_feasible_with_self(u, row) == py.is_feasible_with_self(u, row)
        assert pl.colliding_link_names_with_self() == py.colliding_link_names_with_self()
    assert py.is_feasible(u, q[0][:3], objects) is False      # Err(DofMismatch) folds to false (:79-82)


def _walk(a, b, step, dtype):
    """the samples rrt's stepping evaluates between two way points (iterated recurrence)"""
    a, b = a.astype(dtype), b.astype(dtype)
    out, base, step = [], a.copy(), dtype(step)
    while True:
        d = dtype(np.sqrt(np.sum((base - b) ** 2, dtype=dtype)))
        if d < step:
            return out
        base = base + (b - base) * step / d
        out.append(base.copy())


def _find_goal(pl, u, obs, seed, lo, hi):
    rng = np.random.default_rng(seed)
    for _ in range(200):
        g = lo + (hi - lo) * rng.random(len(u))
        if pl.is_feasible_batch(u, g[None, :], obs)[0]:
            return g
    raise AssertionError("no feasible goal found")


@pytest.mark.parametrize("prec", [32, 64])
def test_plan_is_valid_and_speculation_does_not_change_it(paths, sample_robot, obstacles_robot, prec):
    dtype = np.float32 if prec == 32 else np.float64
    pl = H.HostPlanner(paths[0], 0.01, prec, step_length=0.1, max_try=5000, num_smoothing=100, seed=7)
    obs = H.HostCompound.from_urdf_file(paths[1])
    u = pl.using_joints("l_tool_fixed")            # examples/minimum.rs:35: the arm to l_tool_fixed
    assert len(u) == 6
    chain = urdf.Chain.from_urdf(sample_robot)
    lo, hi = W.joint_limits(chain, u)
    start = np.zeros(6)
    goal = _find_goal(pl, u, obs, 5, lo, hi)
    results = {}
    for spec in (True, False):
        pl.set_speculation(spec, 16, seed=1234)
        path = pl.plan(u, start, goal, obs)
        results[spec] = (path, dict(pl.last_stats))
    p_spec, s_spec = results[True]
    p_seq, s_seq = results[False]
    print(prec, "way points", len(p_spec), "speculative", s_spec, "sequential", s_seq)
    assert p_spec.shape == p_seq.shape and (p_spec == p_seq).all()            # same plan either way
    assert s_spec["consumed"] == s_seq["samples"] == s_seq["consumed"]        # same samples consumed
    assert s_spec["gpu_calls"] < s_seq["gpu_calls"]
    np.testing.assert_array_equal(p_spec[0], start.astype(dtype).astype(np.float64))
    np.testing.assert_array_equal(p_spec[-1], goal.astype(dtype).astype(np.float64))
    # every way point, and every sample rrt's stepping evaluates between way points that are more than
    # a step apart, is feasible — by the engine and by the CPU oracle outside the parity band
    samples = [p_spec[0]]
    for a, b in zip(p_spec[:-1], p_spec[1:]):
        samples += [s.astype(np.float64) for s in _walk(a, b, 0.1, dtype)] + [b]
    samples = np.array(samples)
    assert pl.is_feasible_batch(u, samples, obs).all()
    w = W.sample_workload(sample_robot, obstacles_robot, 0.01, arm_only=True)
    ora = orc.OracleScene(w.robot_keep, w.obstacle_keep, 0.01, prec)
    flags, band = ora.check(samples.astype(dtype), D.GH_CHECK_ENV, threads=8)
    safe = certain(band, prec)
    assert safe.mean() > 0.8
    assert ((flags[safe] & (D.GH_FLAG_ENV | D.GH_FLAG_LIMIT)) == 0).all()
    # the joints are left at the goal
    np.testing.assert_array_equal(pl.joint_positions(u), goal.astype(dtype).astype(np.float64))


def test_plan_errors(paths):
    pl = H.HostPlanner(paths[0], 0.01, 64, max_try=3, num_smoothing=0, seed=1)
    obs = H.HostCompound.from_urdf_file(paths[1])
    u = pl.using_joints()
    free = np.zeros(len(u))
    assert pl.is_feasible(u, free, obs)
    # a box around the whole robot: start collides -> Error::Collision{Start}; the names are those of the
    # restored state (joint_path_planner.rs:156-160 restores the joints before collecting them)
    wall = H.HostCompound.from_specs([D.ShapeSpec.box((0.5, 1.0, 0.5), t=(0.9, 0.0, 0.0))])
    before = pl.joint_positions(u)
    with pytest.raises(H.GearHostError) as e:
        pl.plan(u, free, free, wall)
    assert e.value.kind == "CollisionStart"
    np.testing.assert_array_equal(pl.joint_positions(u), before)
    # goal out of limits -> is_feasible(goal) is false -> Error::Collision{End}
    bad_goal = free.copy()
    bad_goal[0] = 100.0
    with pytest.raises(H.GearHostError) as e:
        pl.plan(u, free, bad_goal, obs)
    assert e.value.kind == "CollisionEnd"
    # three RRT iterations cannot connect distant configurations -> Error::PathPlanFail
    far = free.copy()
    far[0], far[1] = 2.5, -1.2
    if pl.is_feasible_batch(u, far[None, :], obs)[0]:
        with pytest.raises(H.GearHostError) as e:
            pl.plan(u, free, far, obs)
        assert e.value.kind == "PathPlanFail"


def test_plan_avoid_self_collision_and_ik_seeds(paths, sample_robot):
    pl = H.HostPlanner(paths[0], 0.01, 64, self_pairs=PAIRS, seed=3)
    obs = H.HostCompound.from_urdf_file(paths[1])
    u = pl.using_joints("l_tool_fixed")
    chain = urdf.Chain.from_urdf(sample_robot)
    lo, hi = W.joint_limits(chain, u)
    rng = np.random.default_rng(9)
    cand = lo + (hi - lo) * rng.random((400, len(u)))
    ok = pl.is_feasible_batch(u, cand, None, with_self=True)
    assert ok.any() and not ok.all()
    a, b = cand[ok][0], cand[ok][-1]
    path = pl.plan_avoid_self_collision(u, a, b)
    assert len(path) >= 2 and (path[0] == a).all() and (path[-1] == b).all()
    samples = [path[0]]
    for p, q in zip(path[:-1], path[1:]):
        samples += _walk(p, q, 0.1, np.float64) + [q]
    assert pl.is_feasible_batch(u, np.array(samples), None, with_self=True).all()
    bad = cand[~ok][0]
    with pytest.raises(H.GearHostError) as e:
        pl.plan_avoid_self_collision(u, bad, b)
    assert e.value.kind == "SelfCollisionStart"
    # IK seed batch (src/ik.rs:77-78): seeds inside the limits, feasibility identical to the predicate
    seeds, okk = pl.feasible_ik_seeds(u, np.zeros(len(u)), 2048, obs)
    assert (seeds >= lo).all() and (seeds <= hi).all()
    assert (okk == pl.is_feasible_batch(u, seeds, obs)).all()
    assert okk.any() and not okk.all()


def test_plan_in_a_cluttered_field(tmp_path):
    """7-DOF arm among 96 box/ball obstacles: an RRT that needs many iterations.  The speculative and the
    sample-by-sample evaluation must agree on every tree decision, hence on the path."""
    robot = W.arm7_robot()
    p = tmp_path / "arm7.urdf"
    p.write_text(urdf.robot_to_xml(robot))
    field = W.obstacle_field(96, 5, ("box", "ball"), (0.6, 0.4))
    obs = H.HostCompound.from_specs(field.shapes)
    pl = H.HostPlanner(str(p), 0.01, 32, step_length=0.1, max_try=5000, num_smoothing=100, seed=11)
    u = pl.using_joints()
    chain = urdf.Chain.from_urdf(robot)
    lo, hi = W.joint_limits(chain, u)
    rng = np.random.default_rng(3)
    cand = (lo + (hi - lo) * rng.random((600, len(u)))).astype(np.float32).astype(np.float64)
    ok = pl.is_feasible_batch(u, cand, obs)
    free = cand[ok]
    assert len(free) >= 2
    start, goal = free[0], free[1]
    out = {}
    for spec in (True, False):
        pl.set_speculation(spec, 16, seed=99)
        out[spec] = (pl.plan(u, start, goal, obs), dict(pl.last_stats))
    (p1, s1), (p0, s0) = out[True], out[False]
    print("cluttered: way points", len(p1), "speculative", s1, "sequential", s0)
    assert p1.shape == p0.shape and (p1 == p0).all()
    assert s1["consumed"] == s0["consumed"] and s1["iterations"] == s0["iterations"]
    assert s1["gpu_calls"] * 3 < s0["gpu_calls"]
    samples = [p1[0]]
    for a, b in zip(p1[:-1], p1[1:]):
        samples += [s.astype(np.float64) for s in _walk(a, b, 0.1, np.float32)] + [b]
    assert pl.is_feasible_batch(u, np.array(samples), obs).all()
    w = W.Workload("arm7_field", robot, field, 0.01)
    ora = orc.OracleScene(w.robot_keep, w.obstacle_keep, 0.01, 32)
    flags, band = ora.check(np.array(samples, np.float32), D.GH_CHECK_ENV, threads=8)
    safe = certain(band, 32)
    assert safe.mean() > 0.8
    assert ((flags[safe] & (D.GH_FLAG_ENV | D.GH_FLAG_LIMIT)) == 0).all()


def test_ik_seeds_on_a_robot_with_continuous_joints(tmp_path):
    """RandomInitializeIKSolver's seed draw (src/ik.rs:77-79) on joints WITHOUT limits: the [-pi, pi) draw of
    funcs.rs:84 and the wrap of modify_to_nearest_angle (funcs.rs:104-116) run in the C++ host layer, and the
    batch's feasibility bits equal the oracle's on the same seeds."""
    import dataclasses
    robot = W.arm7_robot()
    for k in (0, 4, 6):      # j0, j4, j6 become `continuous`: no Range (k::urdf), so no limit gate either
        robot.joints[k] = dataclasses.replace(robot.joints[k], type="continuous", lower=0.0, upper=0.0)
    p = tmp_path / "arm7c.urdf"
    p.write_text(urdf.robot_to_xml(robot))
    field = W.obstacle_field(64, 21, ("box", "ball"), (0.6, 0.4))
    obs = H.HostCompound.from_specs(field.shapes)
    for prec, dtype in ((32, np.float32), (64, np.float64)):
        pl = H.HostPlanner(str(p), 0.01, prec, seed=5)
        u = pl.using_joints()
        d = pl.describe()
        free_cols = [c for c, i in enumerate(u) if not d["nodes"][i]["has_limit"]]
        assert free_cols == [0, 4, 6]
        initial = np.array([3.0, 0.1, 0.2, -0.3, -3.0, 0.4, 2.5])
        seeds, ok = pl.feasible_ik_seeds(u, initial, 4096, obs)
        ini = initial.astype(dtype).astype(np.float64)
        tol = 1e-5 if prec == 32 else 1e-12
        for c in free_cols:      # nearest representative of the drawn angle: within pi of the initial value
            assert (np.abs(seeds[:, c] - ini[c]) <= np.pi + tol).all()
            assert (np.abs(seeds[:, c]) > np.pi).any()           # the +-2 pi branches ran
            assert np.ptp(seeds[:, c]) > 6.0                     # the whole circle was drawn from
        chain = urdf.Chain.from_urdf(robot)
        lo, hi = W.joint_limits(chain, u)
        for c in range(len(u)):
            if c not in free_cols:
                assert (seeds[:, c] >= lo[c]).all() and (seeds[:, c] <= hi[c]).all()
        assert (ok == pl.is_feasible_batch(u, seeds, obs)).all()
        assert ok.any() and not ok.all()
        w = W.Workload("arm7_continuous", robot, field, 0.01)
        ora = orc.OracleScene(w.robot_keep, w.obstacle_keep, 0.01, prec)
        flags, band = ora.check(seeds.astype(dtype), D.GH_CHECK_ENV, threads=8)
        safe = certain(band, prec)
        assert safe.mean() > 0.95
        assert (flags & D.GH_FLAG_LIMIT).sum() == 0              # unlimited joints beyond +-pi raise no limit flag
        assert (ok[safe] == ((flags[safe] & (D.GH_FLAG_ENV | D.GH_FLAG_LIMIT)) == 0)).all()
