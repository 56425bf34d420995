"""Generates tests/golden/*.json from the reference's own test data.

Run once in the build container (the reference does not travel to the GPU box):
    python tests/golden/make_fixtures.py

Inputs (data files the reference's tests hold, read as data, not copied):
    /root/reference/sample.urdf      robot of `collision_check`  (joint_path_planner.rs:375)
    /root/reference/obstacles.urdf   obstacle compound of examples/minimum.rs:37
Outputs:
    sample_robot.json, obstacles.json   the numeric content, in gear_amd.urdf's dict form
    collision_check.json                the three known-answer cases of
                                        src/planner/joint_path_planner.rs:373-419 (f32, margin 0.01)
    funcs.json                          the vectors of src/funcs.rs's doctest (:91-97) and of
                                        test_funcs (:189-212): inputs and expected outputs
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from gear_amd import urdf  # noqa: E402

REF = "/root/reference"


def main():
    for src, dst in (("sample.urdf", "sample_robot.json"), ("obstacles.urdf", "obstacles.json")):
        robot = urdf.read_urdf(os.path.join(REF, src))
        with open(os.path.join(HERE, dst), "w") as fh:
            json.dump(urdf.robot_to_dict(robot), fh, indent=1)
    # joint_path_planner.rs:373-419 — transcribed test vectors (inputs and expected outputs)
    golden = {
        "source": "src/planner/joint_path_planner.rs:373-419 (test collision_check)",
        "precision": 32,
        "margin": 0.01,
        "target_half_extents": [0.5, 1.0, 0.5],
        "cases": [
            {"q": [0.0] * 8, "target_xyz": [0.9, 0.0, 0.0],
             "names": ["l_elbow_pitch", "l_wrist_yaw", "l_wrist_pitch", "l_gripper_linear2", "l_gripper_linear1"]},
            {"q": [-1.3, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0], "target_xyz": [0.9, 0.0, 0.0],
             "names": ["l_wrist_yaw", "l_wrist_pitch", "l_gripper_linear2", "l_gripper_linear1"]},
            {"q": [-1.3, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0], "target_xyz": [0.7, 0.0, 0.0],
             "names": ["l_shoulder_roll", "l_elbow_pitch", "l_wrist_yaw", "l_wrist_pitch",
                       "l_gripper_linear2", "l_gripper_linear1"]},
        ],
    }
    with open(os.path.join(HERE, "collision_check.json"), "w") as fh:
        json.dump(golden, fh, indent=1)
    # src/funcs.rs — transcribed test vectors.  A limit is null (None) or [min, max] (k::joint::Range::new).
    funcs = {
        "source": "src/funcs.rs:91-97 (doctest of modify_to_nearest_angle), src/funcs.rs:189-212 (test_funcs)",
        "modify_to_nearest_angle": {
            "vec1": [1.0, 0.5], "vec2": [0.1, 10.0], "limits": [[0.0, 0.2], None],
            "expect_exact": {"0": 0.1}, "expect_close": {"1": 3.716814}, "tolerance": 0.000001},
        "generate_clamped_joint_positions_from_limits": {
            "limits": [None, [-1.0, 1.0], [0.0, 0.1]],
            "dof_mismatch_input": [0.1],
            "input": [100.0, -2.0, 0.5], "expect": [100.0, -1.0, 0.1], "tolerance": 0.00001},
        "generate_random_joint_positions_from_limits": {
            "limits": [None, [-1.0, 1.0], [0.0, 0.1]], "draws": 1000,
            "ranges_half_open": [["-PI", "PI"], [-1.0, 1.0], [0.0, 0.1]]},
    }
    with open(os.path.join(HERE, "funcs.json"), "w") as fh:
        json.dump(funcs, fh, indent=1)


if __name__ == "__main__":
    main()
