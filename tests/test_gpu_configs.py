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
from tests.parity import POSE_TOL, adjudicate, certain, check_flags, compare_flags, record

pytestmark = pytest.mark.gpu

BOTH = D.GH_CHECK_ENV | D.GH_CHECK_SELF


@pytest.mark.parametrize("prec", [32, 64])
def test_config2_edge_validation_256_box_cyl(prec):
    w = W.ur10_workload(256)
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    ora = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    ora64 = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, 64)
    # 640 straight segments of 16..64 samples at step 0.1, through the pipeline path
    q, off = w.edges(640, seed=41, dtype=hip.dtype, min_len=16, max_len=64)
    assert len(q) >= 16384
    bad_mask = D.GH_FLAG_ENV | D.GH_FLAG_LIMIT
    flags, first = hip.check_edges(q, off, D.GH_CHECK_ENV, bad_mask)
    # first_bad is a pure function of the flags
    for e in range(len(off) - 1):
        seg = flags[off[e]:off[e + 1]] & bad_mask
        nz = np.nonzero(seg)[0]
        assert first[e] == (nz[0] if len(nz) else -1)
    # the field is dense (two thirds of random configurations collide): few whole edges stay free
    assert (first >= 0).sum() > 50 and (first < 0).sum() >= 20 and (first > 0).sum() > 50
    # oracle on a bounded subset of the edges
    k = 12
    sub_q, sub_off = q[:off[k]], off[:k + 1]
    wf, wb = ora.check_edges(sub_q, sub_off, D.GH_CHECK_ENV, bad_mask)
    allf, band = ora.check(sub_q, D.GH_CHECK_ENV, threads=16)
    safe_s = certain(band, prec)
    safe_e = np.array([safe_s[sub_off[e]:sub_off[e + 1]].all() for e in range(k)])
    assert safe_e.sum() >= k // 2
    assert (first[:k][safe_e] == wb[safe_e]).all()
    rep = compare_flags(flags[:off[k]], allf, band, prec)
    if prec == 32:
        adjudicate(rep, flags[:off[k]], allf, sub_q, D.GH_CHECK_ENV, ora64)
    record(f"config2_edge_validation[f{prec}]", rep, edges=k)
    assert rep["outside"] == 0, rep


@pytest.mark.parametrize("prec", [32, 64])
def test_config3_4096_obstacles_large_batch(prec):
    w = W.sawyer_workload(4096)
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    q = w.configs(1 << 15, seed=42, dtype=hip.dtype)
    a = hip.check_batch(q, BOTH | D.GH_PATH_PIPELINE)
    b = hip.check_batch(q[:8192], BOTH | D.GH_PATH_FUSED)
    assert (a[:8192] == b).all()
    assert 0.05 < ((a & 1) != 0).mean() < 0.999
    ora = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    ora64 = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, 64) if prec == 32 else None
    check_flags(f"config3_4096_obstacles[f{prec}]", hip, ora, q[:320], BOTH | D.GH_PATH_PIPELINE, prec, ora64)


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
    ora64 = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, 64) if prec == 32 else None
    for path in (D.GH_PATH_FUSED, D.GH_PATH_PIPELINE):
        rep, want = check_flags(f"config4_dual_arm_mesh_links[f{prec},path={path:#x}]", hip, ora, q, BOTH | path, prec, ora64)
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
    ora = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, 64)
    seeds = w.configs(20000, seed=44, dtype=np.float64)     # every joint has limits: modify_to_nearest_angle is the identity
    got = hip.check_batch(seeds, BOTH)
    assert (got == hip.check_batch(seeds, BOTH | D.GH_PATH_FUSED)).all()
    check_flags("config4_ik_seed_batch[f64]", hip, ora, seeds[:1500], BOTH, 64)
    feasible = (got & (D.GH_FLAG_ENV | D.GH_FLAG_SELF | D.GH_FLAG_LIMIT)) == 0
    assert 0.05 < feasible.mean() < 0.99
