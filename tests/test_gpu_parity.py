"""GPU parity tests proper: the HIP path, called through the C ABI (libgear_hip.so),
against the CPU oracle on identical seeded inputs and against the committed golden
fixtures.  Flags are bit-exact outside the parity band; poses within 1e-6 / 1e-12."""
import numpy as np
import pytest

from gear_amd import descriptors as D
from gear_amd import engine, urdf
from gear_amd import workloads as W
from oracle import binding as orc
from tests import oracle_fixtures as F
from tests.parity import POSE_TOL, certain, check_flags, check_flags_fixture

pytestmark = pytest.mark.gpu

BOTH = D.GH_CHECK_ENV | D.GH_CHECK_SELF


def _pair(w, prec):
    return (engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec),
            orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, prec))


@pytest.fixture(scope="module")
def sample_w(sample_robot, obstacles_robot):
    return W.sample_workload(sample_robot, obstacles_robot, 0.01,
                             self_pairs=[("root", "l_wrist_yaw"), ("root", "l_wrist_pitch"),
                                         ("l_shoulder_yaw", "l_wrist_pitch"), ("root", "l_gripper_linear1"),
                                         ("l_shoulder_pitch", "l_gripper_linear2"), ("nonexistent", "root")])


def _names(chain, mask_row):
    return [chain.nodes[k].name for k in range(len(chain.nodes)) if (int(mask_row[k >> 6]) >> (k & 63)) & 1]


def test_golden_collision_check_lists(sample_robot, collision_check_golden):
    """src/planner/joint_path_planner.rs:373-419 through gh_link_hits."""
    g = collision_check_golden
    chain = urdf.Chain.from_urdf(sample_robot)
    model = D.CollisionModel.from_urdf_robot(sample_robot, g["margin"])
    for case in g["cases"]:
        target = D.Compound([D.ShapeSpec.box(g["target_half_extents"], t=case["target_xyz"])])
        sc = engine.Scene(D.flatten_robot(chain, model, chain.movable_nodes()), D.flatten_obstacles(target),
                          g["margin"], 32)
        m = sc.link_hits(np.array([case["q"]], np.float32))
        assert _names(chain, m[0, 0]) == case["names"]
        flags = sc.check_batch(np.array([case["q"]], np.float32), D.GH_CHECK_ENV)
        assert flags[0] == D.GH_FLAG_ENV


@pytest.mark.parametrize("prec", [32, 64])
def test_fk_poses_match_oracle(sample_w, prec):
    for w in (sample_w, W.metric_workload(8)):
        hip, ora = _pair(w, prec)
        q = w.configs(20000, seed=3, dtype=hip.dtype, overshoot=0.05)
        got, want = hip.fk_batch(q), ora.fk(q)
        err = np.abs(got.astype(np.float64) - want.astype(np.float64)).max()
        print(w.name, prec, "max pose error", err)
        assert err <= POSE_TOL[prec]


def _check_flags(name, w, prec, n, seed, overshoot=0.0, what=BOTH):
    hip, ora = _pair(w, prec)
    ora64 = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, 64) if prec == 32 else None
    q = w.configs(n, seed=seed, dtype=hip.dtype, overshoot=overshoot)
    return check_flags(f"{name}[f{prec}]", hip, ora, q, what, prec, ora64)


@pytest.mark.parametrize("prec", [32, 64])
def test_flags_sample_urdf(sample_w, prec):
    rep, want = _check_flags("flags_sample_urdf", sample_w, prec, 20000, seed=0, overshoot=0.02)
    assert ((want & 1) > 0).mean() > 0.02 and ((want & 4) > 0).mean() > 0.02 and ((want & 2) > 0).mean() > 0.001


@pytest.mark.parametrize("prec", [32, 64])
def test_flags_sample_urdf_arm_only(sample_robot, obstacles_robot, prec):
    # using_joints = the 6-joint arm to l_tool_fixed (examples/minimum.rs:35); grippers keep their state
    w = W.sample_workload(sample_robot, obstacles_robot, 0.01, arm_only=True)
    assert w.dof == 6
    _check_flags("flags_sample_urdf_arm_only", w, prec, 5000, seed=1, overshoot=0.02, what=D.GH_CHECK_ENV)


def _fixture_flags(name, fixture, prec, what_extra=0):
    fx = F.load(f"{fixture}_f{prec}")
    w = fx["w"]
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    return check_flags_fixture(f"{name}[f{prec}]", hip, fx, fx["spec"]["what"] | what_extra)


@pytest.mark.parametrize("prec", [32, 64])
def test_flags_metric_scene(prec):
    # 16384 configurations of the metric scene against the committed oracle answers, on the default path (the
    # single-launch kernel at this size) and through the pipeline
    rep, got = _fixture_flags("flags_metric_scene", "metric", prec)
    rep2, got2 = _fixture_flags("flags_metric_scene_pipeline", "metric", prec, D.GH_PATH_PIPELINE)
    assert (got == got2).all()


@pytest.mark.parametrize("prec", [32, 64])
def test_flags_cylinder_obstacles(prec):
    # cylinder obstacles are reference-undefined (Compound::new rejects TriMesh); parity is against the
    # oracle's surface-distance definition
    _fixture_flags("flags_cylinder_obstacles", "cylinder_obstacles", prec)


@pytest.mark.parametrize("prec", [32, 64])
def test_flags_4096_obstacles(prec):
    _fixture_flags("flags_4096_obstacles", "config3_4096", prec)     # default path; pipeline and fused: test_gpu_configs.py


@pytest.mark.parametrize("prec", [32, 64])
def test_link_and_self_hits(sample_w, prec):
    hip, ora = _pair(sample_w, prec)
    q = sample_w.configs(300, seed=6, dtype=hip.dtype, overshoot=0.02)
    flags, band = ora.check(q, BOTH, threads=8)
    safe = certain(band, prec)   # outside the stated band, every free pair certified free by the oracle itself
    assert safe.mean() > 0.9
    gl, wl = hip.link_hits(q), ora.link_hits(q)
    gs, ws = hip.self_hits(q), ora.self_hits(q)
    assert (gl[safe] == wl[safe]).all()
    assert (gs[safe] == ws[safe]).all()
    assert wl.any() and ws.any()


@pytest.mark.parametrize("prec", [32, 64])
def test_edges_first_bad(sample_w, prec):
    hip, ora = _pair(sample_w, prec)
    q, off = sample_w.edges(300, seed=7, dtype=hip.dtype, min_len=4, max_len=40)
    gf, gb = hip.check_edges(q, off, D.GH_CHECK_ENV, D.GH_FLAG_ENV | D.GH_FLAG_LIMIT)
    wf, wb = ora.check_edges(q, off, D.GH_CHECK_ENV, D.GH_FLAG_ENV | D.GH_FLAG_LIMIT)
    allf, band = ora.check(q, D.GH_CHECK_ENV, threads=8)
    # an edge is comparable when no sample of it sits in the band
    safe_s = certain(band, prec)
    safe_e = np.array([safe_s[off[e]:off[e + 1]].all() for e in range(len(off) - 1)])
    assert safe_e.mean() > 0.8
    assert (gb[safe_e] == wb[safe_e]).all()
    assert (wb >= 0).sum() > 10 and (wb < 0).sum() > 10
    # the reference never evaluates samples after the first failure (0xFF in the oracle's output)
    seen = wf != 0xFF
    assert ((gf == wf) | ~seen | ~safe_s).all()


def test_edge_cases(sample_w):
    hip, _ = _pair(sample_w, 32)
    assert hip.check_batch(np.zeros((0, hip.dof), np.float32)).shape == (0,)
    f, b = hip.check_edges(np.zeros((0, hip.dof), np.float32), np.zeros(1, np.uint32))
    assert f.shape == (0,) and b.shape == (0,)
    # ragged edges including an empty one
    q = sample_w.configs(5, seed=1)
    f, b = hip.check_edges(q, np.array([0, 0, 2, 5], np.uint32))
    assert b[0] == -1 and f.shape == (5,)
    # NaN / inf inputs fold to a limit violation, like OutOfLimitError -> false
    bad = np.zeros((3, hip.dof), np.float32)
    bad[0, 0], bad[1, 2], bad[2, 7] = np.nan, np.inf, 1e30
    assert (hip.check_batch(bad) == D.GH_FLAG_LIMIT).all()
    with pytest.raises(engine.GearHipError):
        hip.check_batch(np.zeros((4, hip.dof + 1), np.float32))
    # margin 0.0 (the builder's default, joint_path_planner.rs:341): distance is never < 0, nothing collides
    w0 = W.Workload("m0", sample_w.robot, sample_w.objects, 0.0)
    hip0 = engine.Scene(w0.robot_keep, w0.obstacle_keep, 0.0, 32)
    assert (hip0.check_batch(sample_w.configs(2000, seed=2)) == 0).all()


def test_full_size_properties():
    """configs[1] size (1M configs): properties that need no oracle."""
    w = W.metric_workload()
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32)
    n = 1 << 20
    q = w.configs(n, seed=11, overshoot=0.01)
    f1 = hip.check_batch(q)
    assert (f1 == hip.check_batch(q)).all()                          # deterministic
    perm = np.random.default_rng(0).permutation(n)
    assert (hip.check_batch(q[perm]) == f1[perm]).all()             # lanes are independent
    parts = np.concatenate([hip.check_batch(q[:333]), hip.check_batch(q[333:70001]), hip.check_batch(q[70001:])])
    assert (parts == f1).all()                                       # batch-split invariant
    env_only = hip.check_batch(q, D.GH_CHECK_ENV)
    self_only = hip.check_batch(q, D.GH_CHECK_SELF)
    assert ((env_only | self_only) == f1).all()                      # bits are independent predicates
    big = engine.Scene(w.robot_keep, w.obstacle_keep, 0.03, 32).check_batch(q)
    assert ((f1 & ~big) == 0).all() and (big != f1).any()            # a larger margin only adds collisions
    assert ((f1 & 4) == 0).sum() > 0.5 * n
