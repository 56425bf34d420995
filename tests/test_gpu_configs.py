"""BASELINE.json configs[2..4] as parity cases (the robots are builder-authored stand-ins: UR10 /
Sawyer / PR2 descriptions are not in the reference; see gear_amd/workloads.py):
  [2] 6-DOF arm, RRT edge validation vs a 256 box/cylinder field
  [3] 7-DOF arm vs 4096 mixed primitives, large batch
  [4] dual-arm robot with convex-mesh links: env + full self pair list, IK random-seed batch."""
import numpy as np
import pytest

from gear_amd import descriptors as D
from gear_amd import engine
from gear_amd import workloads as W
from oracle import binding as orc
from tests import oracle_fixtures as F
from tests.parity import POSE_TOL, certain, check_flags_fixture

pytestmark = pytest.mark.gpu

BOTH = D.GH_CHECK_ENV | D.GH_CHECK_SELF


def _first_is_function_of_flags(flags, off, first, bad_mask):
    for e in range(len(off) - 1):
        nz = np.nonzero(flags[off[e]:off[e + 1]] & bad_mask)[0]
        assert first[e] == (nz[0] if len(nz) else -1)


@pytest.mark.parametrize("prec", [32, 64])
def test_config2_edge_validation_256_box_cyl(prec):
    w = W.ur10_workload(256)
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    bad_mask = D.GH_FLAG_ENV | D.GH_FLAG_LIMIT
    # 640 straight segments of 16..64 samples at step 0.1, through the pipeline path
    q, off = w.edges(640, seed=41, dtype=hip.dtype, min_len=16, max_len=64)
    assert len(q) >= 16384
    flags, first = hip.check_edges(q, off, D.GH_CHECK_ENV, bad_mask)
    _first_is_function_of_flags(flags, off, first, bad_mask)   # first_bad is a pure function of the flags
    # the field is dense (two thirds of random configurations collide): few whole edges stay free
    assert (first >= 0).sum() > 50 and (first < 0).sum() >= 20 and (first > 0).sum() > 50
    # the oracle's answers for the first 256 of these edges (about 10 k samples) are committed: tests/oracle_fixtures.py
    fx = F.load(f"config2_edges_f{prec}")
    k, qs, offs = len(fx["off"]) - 1, fx["q"], fx["off"]
    assert k == 256 and (qs == q[:off[k]]).all() and (offs == off[:k + 1]).all()     # the same edges, drawn again
    f2, b2 = hip.check_edges(qs, offs, D.GH_CHECK_ENV, bad_mask)                       # (the small-batch kernel's path)
    assert (f2 == flags[:off[k]]).all() and (b2 == first[:k]).all()
    safe_s = certain(fx["band"], prec)
    safe_e = np.array([safe_s[offs[e]:offs[e + 1]].all() for e in range(k)])
    assert safe_e.sum() >= k // 2
    assert (first[:k][safe_e] == fx["first_bad"][safe_e]).all()                        # rrt's stop-at-first-failure answer
    seen = fx["flags_faithful"] != 0xFF                                               # samples the reference evaluates at all
    assert ((f2 == fx["flags_faithful"]) | ~seen | ~safe_s).all()
    check_flags_fixture(f"config2_edge_validation[f{prec}]", hip, fx, got=f2)


@pytest.mark.parametrize("prec", [32, 64])
def test_config3_4096_obstacles_large_batch(prec):
    w = W.sawyer_workload(4096)
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    q = w.configs(1 << 15, seed=42, dtype=hip.dtype)
    a = hip.check_batch(q, BOTH | D.GH_PATH_PIPELINE)
    b = hip.check_batch(q[:8192], BOTH | D.GH_PATH_FUSED)
    assert (a[:8192] == b).all()
    assert 0.05 < ((a & 1) != 0).mean() < 0.999
    fx = F.load(f"config3_4096_f{prec}")                     # 8192 configurations against the committed oracle answers
    for path in (D.GH_PATH_PIPELINE, D.GH_PATH_FUSED, 0):
        check_flags_fixture(f"config3_4096_obstacles[f{prec},path={path:#x}]", hip, fx, BOTH | path)


@pytest.mark.parametrize("prec", [32, 64])
def test_config4_dual_arm_mesh_links(prec):
    w = W.pr2_workload(128)
    d = w.robot_keep.desc
    assert (d.n_nodes, d.n_cols, d.n_shapes, d.n_meshes) == (16, 15, 16, 12) and d.n_pairs == 37
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    ora = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    q = w.configs(2500, seed=43, dtype=hip.dtype, overshoot=0.01)
    err = np.abs(hip.fk_batch(q).astype(np.float64) - ora.fk(q).astype(np.float64)).max()
    assert err <= POSE_TOL[prec], err
    fx = F.load(f"config4_dual_arm_f{prec}")                 # 8192 configurations, oracle answers committed
    for path in (D.GH_PATH_FUSED, D.GH_PATH_PIPELINE, 0):
        check_flags_fixture(f"config4_dual_arm_mesh_links[f{prec},path={path:#x}]", hip, fx, BOTH | path)
    want = fx["flags"]
    assert ((want & 1) != 0).mean() > 0.05 and ((want & 2) != 0).mean() > 0.05 and ((want & 4) != 0).mean() > 0.01
    # diagnostics on the branching chain: compared where the oracle's own evidence is unambiguous
    _, band = ora.check(q[:300], BOTH, threads=16)
    safe = certain(band, prec)
    assert safe.sum() > 200
    assert (hip.link_hits(q[:300])[safe] == ora.link_hits(q[:300])[safe]).all()
    assert (hip.self_hits(q[:300])[safe] == ora.self_hits(q[:300])[safe]).all()


def test_config4_ik_seed_batch():
    """RandomInitializeIKSolver's seeds (src/ik.rs:77-78: random from limits, then modify_to_nearest_angle)
    for the left arm, as one feasibility batch; the other arm keeps its current joint state."""
    w = W.pr2_workload(128, end_joint="l_gripper")
    assert w.dof == 8                      # torso lift + 7 arm joints
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 64)
    # every joint of this robot has limits: modify_to_nearest_angle is the identity (the wrap branch runs in
    # tests/test_gpu_host_cpp.py::test_ik_seeds_on_a_robot_with_continuous_joints)
    seeds = w.configs(20000, seed=44, dtype=np.float64)
    got = hip.check_batch(seeds, BOTH)
    assert (got == hip.check_batch(seeds, BOTH | D.GH_PATH_FUSED)).all()
    fx = F.load("config4_ik_seeds_f64")
    assert (fx["q"] == seeds[:8192]).all()
    check_flags_fixture("config4_ik_seed_batch[f64]", hip, fx, got=got[:8192])
    feasible = (got & (D.GH_FLAG_ENV | D.GH_FLAG_SELF | D.GH_FLAG_LIMIT)) == 0
    assert 0.05 < feasible.mean() < 0.99
