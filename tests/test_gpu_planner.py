"""The host mirror of JointPathPlanner's predicates, written like the reference's own tests
(src/planner/joint_path_planner.rs:367-427) and checked against the oracle."""
import numpy as np
import pytest

from gear_amd import descriptors as D
from gear_amd import planner as P
from gear_amd import urdf
from gear_amd import workloads as W
from oracle import binding as orc

pytestmark = pytest.mark.gpu


def _planner(sample_robot, prec=32, pairs=()):
    b = P.JointPathPlannerBuilder.from_urdf_robot(sample_robot).collision_check_margin(0.01).precision(prec)
    return b.self_collision_pairs(list(pairs)).finalize()


def test_collision_check_like_the_reference(sample_robot, collision_check_golden):
    g = collision_check_golden
    planner = _planner(sample_robot, 32)
    using = planner.collision_check_robot.movable_nodes()
    for case in g["cases"]:
        target = D.Compound([D.ShapeSpec.box(g["target_half_extents"], t=case["target_xyz"])])
        # robot.set_joint_positions(&angles).unwrap(); checker.check_env(..).collect()
        assert planner._set_joint_positions(using, case["q"])
        assert planner.colliding_link_names(target) == case["names"]
        assert planner.has_any_colliding(target)
        assert not planner.is_feasible(using, case["q"], target)


def test_from_urdf_defaults(sample_robot):
    p = P.JointPathPlannerBuilder.from_urdf_robot(sample_robot).finalize()
    assert (p.step_length, p.max_try, p.num_smoothing) == (0.1, 5000, 100)
    assert p.collision_checker.prediction == 0.0
    # default margin 0.0: nothing ever collides (strict `<` on a distance >= 0)
    target = D.Compound([D.ShapeSpec.box((0.5, 1.0, 0.5), t=(0.9, 0, 0))])
    assert not p.has_any_colliding(target)
    assert p.colliding_link_names(target) == []


def test_predicates_match_oracle_and_keep_state(sample_robot, obstacles_robot):
    pairs = [("root", "l_wrist_yaw"), ("root", "l_wrist_pitch"), ("l_shoulder_yaw", "l_wrist_pitch")]
    planner = _planner(sample_robot, 64, pairs)
    objects = D.Compound.from_urdf_robot(obstacles_robot)
    w = W.sample_workload(sample_robot, obstacles_robot, 0.01, self_pairs=pairs)
    ora = orc.OracleScene(w.robot_keep, w.obstacle_keep, 0.01, 64)
    using = planner.collision_check_robot.movable_nodes()
    q = w.configs(64, seed=9, dtype=np.float64, overshoot=0.03)
    flags, band = ora.check(q)
    for i in range(len(q)):
        if band[i, 0] <= 1e-9:     # the stated f64 band
            continue
        assert planner.is_feasible(using, q[i], objects) == ((flags[i] & (D.GH_FLAG_ENV | D.GH_FLAG_LIMIT)) == 0)
        assert planner.is_feasible_with_self(using, q[i]) == ((flags[i] & (D.GH_FLAG_SELF | D.GH_FLAG_LIMIT)) == 0)
    ok = planner.is_feasible_batch(using, q, objects)
    safe = band[:, 0] > 1e-9
    assert (ok[safe] == ((flags[safe] & (D.GH_FLAG_ENV | D.GH_FLAG_LIMIT)) == 0)).all()
    # arm-only using_joints (examples/minimum.rs:35): the gripper keeps the state set above
    arm = planner.collision_check_robot.serial_chain_from_end(planner.collision_check_robot.find("l_tool_fixed"))
    assert len(arm) == 6
    assert planner.is_feasible(arm, q[0], objects) is False   # Err(DofMismatch) folds to false (:79-82)
    assert isinstance(planner.is_feasible(arm, q[0][:6], objects), bool)
    # start/goal validation raises gear::Error::Collision with the colliding link names
    colliding = q[np.nonzero((flags & 1) != 0)[0][0]]
    free = q[np.nonzero(flags == 0)[0][0]]
    with pytest.raises(P.Collision) as e:
        planner.check_start_goal(using, free, colliding, objects)
    # the reference restores the joints before collecting the names (:156-159): they describe the
    # restored state, which is the free one here
    assert e.value.part == "End" and e.value.collision_link_names == planner.colliding_link_names(objects)
