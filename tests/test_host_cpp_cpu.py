"""The C++ host layer (gear_amd/host/, libgear_host.so) without a GPU: the library loads, exports what
include/gear_host.h declares, and its URDF reader / k::Chain wiring / CollisionChecker map agree with
the Python mirror field by field.  (Planner construction touches no GPU; predicates do.)"""
import re
import os

import numpy as np
import pytest

from gear_amd import descriptors as D
from gear_amd import host_binding as H
from gear_amd import urdf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    text = open(os.path.join(ROOT, "include", "gear_host.h")).read()
    declared = set(re.findall(r"\b(gp_[a-z_]+)\s*\(", text))
    assert declared == set(H.SYMBOLS)
    L = H.lib()
    for s in declared:
        assert hasattr(L, s), s


@pytest.fixture()
def sample_urdf_path(tmp_path, sample_robot):
    p = tmp_path / "sample.urdf"
    p.write_text(urdf.robot_to_xml(sample_robot))
    return str(p)


def test_chain_and_collision_map_match_python(sample_urdf_path, sample_robot):
    pl = H.HostPlanner(sample_urdf_path, 0.01, 32, self_pairs=["root:l_wrist_yaw"])
    d = pl.describe()
    chain = urdf.Chain.from_urdf(sample_robot)
    model = D.CollisionModel.from_urdf_robot(sample_robot, 0.01)
    assert [n["name"] for n in d["nodes"]] == [n.name for n in chain.nodes]
    assert d["prediction"] == float(np.float32(0.01))      # N = f32: the margin is an f32, as in the reference
    for got, want in zip(d["nodes"], chain.nodes):
        assert got["parent"] == want.parent and got["joint_type"] == want.joint_type
        assert got["mimic_parent"] == want.mimic_parent
        assert got["has_limit"] == (1 if want.limits is not None else 0)
        if want.limits is not None:
            assert (got["lo"], got["hi"]) == tuple(want.limits)
        np.testing.assert_array_equal(got["origin_q"], want.origin_q)
        np.testing.assert_array_equal(got["origin_t"], want.origin_t)
        np.testing.assert_array_equal(got["axis"], want.axis)
        specs = model.name_collision_model_map.get(want.name, [])
        assert len(got["shapes"]) == len(specs)
        for gs, ws in zip(got["shapes"], specs):
            assert gs["kind"] == ws.kind
            np.testing.assert_array_equal(gs["params"], ws.params)
            np.testing.assert_array_equal(gs["t"], ws.t)
            np.testing.assert_array_equal(gs["q"], ws.q)
    assert pl.using_joints() == chain.movable_nodes()
    assert pl.using_joints("l_tool_fixed") == chain.serial_chain_from_end(chain.find("l_tool_fixed"))
    assert pl.joint_names() == chain.joint_names()
    with pytest.raises(H.GearHostError) as e:
        pl.using_joints("l_wrist2")            # a LINK name: k::Chain::find looks up joints
    assert e.value.kind == "NotFound"


def test_errors(tmp_path, sample_urdf_path):
    with pytest.raises(H.GearHostError) as e:
        H.HostPlanner(str(tmp_path / "missing.urdf"), 0.01)
    assert e.value.kind == "Io"
    bad = tmp_path / "bad.urdf"
    bad.write_text("<robot name='x'><link name='a'></robot>")
    with pytest.raises(H.GearHostError) as e:
        H.HostPlanner(str(bad), 0.01)
    assert e.value.kind == "ParseError"
    with pytest.raises(H.GearHostError) as e:
        H.HostPlanner(sample_urdf_path, 0.01, self_pairs=["abcd"])   # collision_checker.rs:334-342
    assert e.value.kind == "ParseError"


def test_xml_reader_handles_comments_and_quotes(tmp_path):
    p = tmp_path / "r.urdf"
    p.write_text("""<?xml version='1.0'?>
<!-- a comment with <tags> inside -->
<robot name="r">
  <link name='a'><visual><geometry><box size="1 1 1"/></geometry></visual>
    <collision><geometry><sphere radius="0.5"/></geometry></collision></link>
  <link name="b"/><link name="c"/><link name="d"/>
  <joint name="j1" type="continuous"><parent link="a"/><child link="b"/><axis xyz="0 0 2"/></joint>
  <joint name="j2" type="prismatic"><parent link="b"/><child link="c"/><limit lower="-1" upper="1" effort="3"/></joint>
  <joint name="j3" type="floating"><parent link="b"/><child link="d"/></joint>
</robot>""")
    d = H.HostPlanner(str(p), 0.0).describe()
    assert [n["name"] for n in d["nodes"]] == ["root", "j1", "j3", "j2"]     # children popped in reverse order
    assert d["nodes"][0]["shapes"][0]["kind"] == D.GH_SHAPE_BALL
    assert d["nodes"][1]["has_limit"] == 0 and d["nodes"][3]["has_limit"] == 1


# ---- src/funcs.rs helpers of the C++ host layer against the reference's own vectors (tests/golden/funcs.json) ----
def _limits_of(v):
    return [None if l is None else tuple(l) for l in v]


@pytest.fixture(scope="module")
def funcs_golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "funcs.json")) as fh:
        return json.load(fh)


def test_modify_to_nearest_angle_doctest(funcs_golden):
    g = funcs_golden["modify_to_nearest_angle"]              # funcs.rs:91-97
    a = H.modify_to_nearest_angle(g["vec1"], g["vec2"], _limits_of(g["limits"]), 64)
    for k, v in g["expect_exact"].items():
        assert a[int(k)] == v, "no change"
    for k, v in g["expect_close"].items():
        assert abs(a[int(k)] - v) < g["tolerance"]
    # the two other branches (funcs.rs:107-116): +2 pi, and already nearest
    lim = [None]
    assert abs(H.modify_to_nearest_angle([3.0], [-3.0], lim)[0] - (-3.0 + 2 * np.pi)) < 1e-12
    assert H.modify_to_nearest_angle([0.2], [0.3], lim)[0] == 0.3
    # f32 arithmetic (N = f32): same doctest inside the f32 rounding of the result
    a32 = H.modify_to_nearest_angle(g["vec1"], g["vec2"], _limits_of(g["limits"]), 32)
    assert a32[0] == float(np.float32(0.1)) and abs(a32[1] - 3.716814) < 1e-5


def test_generate_clamped_joint_positions_from_limits(funcs_golden):
    g = funcs_golden["generate_clamped_joint_positions_from_limits"]   # funcs.rs:203-211
    lim = _limits_of(g["limits"])
    with pytest.raises(H.GearHostError) as e:
        H.generate_clamped_joint_positions_from_limits(g["dof_mismatch_input"], lim)
    assert e.value.kind == "DofMismatch"
    got = H.generate_clamped_joint_positions_from_limits(g["input"], lim)
    assert np.all(np.abs(got - np.array(g["expect"])) < g["tolerance"])
    np.testing.assert_array_equal(got, g["expect"])          # clamping returns the limit itself


def test_generate_random_joint_positions_from_limits(funcs_golden):
    g = funcs_golden["generate_random_joint_positions_from_limits"]    # funcs.rs:196-202
    for precision in (64, 32):
        a = H.generate_random_joint_positions_from_limits(_limits_of(g["limits"]), g["draws"], seed=7, precision=precision)
        assert a.shape == (g["draws"], len(g["limits"]))
        assert np.all(a[:, 0] >= -np.pi) and np.all(a[:, 0] < np.pi)       # None: [-PI, PI)
        assert np.all(a[:, 1] >= -1.0) and np.all(a[:, 1] < 1.0)
        assert np.all(a[:, 2] >= 0.0) and np.all(a[:, 2] < 0.1 + (1e-8 if precision == 32 else 0.0))
        assert a[:, 0].min() < -2.5 and a[:, 0].max() > 2.5                # the whole circle is drawn from

