"""Host-side logic: URDF subset, chain wiring, descriptor flattening, input generators, sharding."""
import math

import numpy as np
import pytest

from gear_amd import descriptors as D
from gear_amd import sharding, urdf
from gear_amd import workloads as W


def test_parse_colon_separated_pairs():
    # mirrors src/collision/collision_checker.rs:334-342 and the doctest at :306-313
    pairs = D.parse_colon_separated_pairs(["j0:j1", "j2:j0"])
    assert pairs == [("j0", "j1"), ("j2", "j0")]
    pairs = D.parse_colon_separated_pairs(["ab:cd", "ab:ef"])
    assert len(pairs) == 2 and pairs[0] == ("ab", "cd") and pairs[1] == ("ab", "ef")
    with pytest.raises(ValueError):
        D.parse_colon_separated_pairs(["abcd"])


def test_collision_model_keys_and_shapes(sample_robot):
    # collision_checker.rs:234-262: keyed by the joint whose child is the link; root link -> "root";
    # l_tool has no <collision> and is omitted; the root box uses the COLLISION origin z = 0.2
    m = D.CollisionModel.from_urdf_robot(sample_robot, 0.01)
    assert set(m.name_collision_model_map) == {
        "root", "l_shoulder_yaw", "l_shoulder_pitch", "l_shoulder_roll", "l_elbow_pitch", "l_wrist_yaw",
        "l_wrist_pitch", "l_gripper_linear1", "l_gripper_linear2"}
    root = m.name_collision_model_map["root"][0]
    assert root.kind == D.GH_SHAPE_BOX and root.params == (0.1, 0.1, 0.3) and root.t == (0.0, 0.0, 0.2)
    cyl = m.name_collision_model_map["l_gripper_linear1"][0]
    assert cyl.kind == D.GH_SHAPE_CYLINDER and cyl.params[:2] == (0.02, 0.15)
    np.testing.assert_allclose(cyl.q, (math.sin(0.785), 0, 0, math.cos(0.785)), atol=1e-15)


def test_flatten_robot_descriptor(sample_robot):
    chain = urdf.Chain.from_urdf(sample_robot)
    model = D.CollisionModel.from_urdf_robot(sample_robot, 0.01)
    keep = D.flatten_robot(chain, model, chain.movable_nodes(),
                           [("root", "l_wrist_yaw"), ("nope", "root"), ("l_tool_fixed", "root")])
    d = keep.desc
    assert (d.n_nodes, d.n_cols, d.n_shapes) == (10, 8, 9)
    assert d.n_pairs == 1          # pairs with a missing model or node are skipped (collision_checker.rs:150-169)
    assert [d.col_node[i] for i in range(8)] == [1, 2, 3, 4, 5, 6, 7, 8]
    g2 = d.nodes[7]                # l_gripper_linear2 mimics l_gripper_linear1 with multiplier -1
    assert g2.mimic_parent == 8 and g2.mimic_mult == -1.0 and g2.joint_type == urdf.LINEAR
    assert [d.nodes[i].parent for i in range(10)] == [-1, 0, 1, 2, 3, 4, 5, 6, 6, 6]
    assert [d.shapes[i].node for i in range(9)] == [0, 1, 2, 3, 4, 5, 6, 7, 8]


def test_urdf_defaults_and_joint_types():
    robot = urdf.parse_urdf("""<robot name="r">
      <link name="a"/><link name="b"/><link name="c"/><link name="d"/>
      <joint name="j1" type="continuous"><parent link="a"/><child link="b"/><axis xyz="0 0 2"/></joint>
      <joint name="j2" type="prismatic"><parent link="b"/><child link="c"/>
         <limit lower="-1" upper="1"/></joint>
      <joint name="j3" type="floating"><parent link="b"/><child link="d"/></joint>
    </robot>""")
    chain = urdf.Chain.from_urdf(robot)
    assert [n.name for n in chain.nodes] == ["root", "j1", "j3", "j2"]    # children popped in reverse order
    j1, j3, j2 = chain.nodes[1], chain.nodes[2], chain.nodes[3]
    assert j1.joint_type == urdf.ROTATIONAL and j1.limits is None           # continuous: no limit
    assert j2.joint_type == urdf.LINEAR and j2.limits == (-1.0, 1.0) and j2.axis == (1.0, 0.0, 0.0)
    assert j3.joint_type == urdf.FIXED
    assert chain.dof() == 2


def test_random_configs_follow_funcs_rs(sample_robot):
    # src/funcs.rs:189-202 (test_funcs): within [min, max); no limit -> [-pi, pi)
    chain = urdf.Chain.from_urdf(sample_robot)
    q = W.random_configs(chain, chain.movable_nodes(), 1000, seed=0, dtype=np.float64)
    lo, hi = W.joint_limits(chain, chain.movable_nodes())
    assert q.shape == (1000, 8) and (q >= lo).all() and (q < hi).all()
    assert (W.random_configs(chain, chain.movable_nodes(), 1000, seed=0, dtype=np.float64) == q).all()


def test_edge_sampling_is_the_iterated_recurrence(sample_robot):
    chain = urdf.Chain.from_urdf(sample_robot)
    q, off = W.straight_edges(chain, chain.movable_nodes(), 20, seed=3, step=0.1, min_len=3, max_len=500)
    assert off[0] == 0 and off[-1] == len(q) and (np.diff(off.astype(np.int64)) >= 1).all()
    for e in range(20):
        seg = q[off[e]:off[e + 1]]
        steps = np.linalg.norm(np.diff(seg.astype(np.float64), axis=0), axis=1)
        assert (steps <= 0.1 + 1e-5).all()
        if len(steps) > 1:
            np.testing.assert_allclose(steps[:-1], 0.1, atol=1e-5)


def test_shard_bounds_cover_and_order():
    for n in (0, 1, 7, 64, 1000003):
        for world in (1, 2, 3, 8):
            cuts = [sharding.shard_bounds(n, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_shard_edges_keep_edges_whole():
    rng = np.random.default_rng(0)
    off = np.concatenate([[0], np.cumsum(rng.integers(1, 65, 500))])
    for world in (1, 2, 4, 8):
        prev_e, prev_s = 0, 0
        for r in range(world):
            e_lo, e_hi, s_lo, s_hi = sharding.shard_edges(off, world, r)
            assert e_lo == prev_e and s_lo == prev_s and s_lo == off[e_lo] and s_hi == off[e_hi]
            prev_e, prev_s = e_hi, s_hi
        assert prev_e == 500 and prev_s == off[-1]


def test_device_group_sharding_arithmetic():
    """gh_scene_create_multi's shares (gear_amd/csrc/gear_shard.hpp, compiled for the host): contiguous, covering the
    batch exactly once, whole edges only, balanced by samples — for every share count a node can have and for the
    awkward sizes (fewer configurations or edges than shares, empty edges, one giant edge)."""
    from tests import harness
    for n in (0, 1, 2, 7, 8, 9, 4097, 100001, (1 << 23) + 5):
        for G in (1, 2, 3, 4, 8, 64):
            g = min(G, max(n, 1))                                   # the library never makes more shares than configurations
            prev = 0
            sizes = []
            for k in range(g):
                lo, hi = harness.shard_range(n, k, g)
                assert lo == prev and hi >= lo
                prev = hi
                sizes.append(hi - lo)
            assert prev == n and max(sizes) - min(sizes) <= 1        # exact cover, balanced to one configuration
    rng = np.random.default_rng(4)
    cases = [rng.integers(0, 65, 1000), np.array([0, 0, 5, 0, 0]), np.array([100000, 1, 1, 1]), np.array([3]), np.zeros(6, int),
             rng.integers(8, 65, 65536)]
    for lens in cases:
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
        E, n = len(lens), int(off[-1])
        for G in (1, 2, 3, 8):
            g = min(G, E)
            cut = harness.shard_edges(off, g)
            assert cut[0] == 0 and cut[-1] == E and (np.diff(cut) >= 0).all()       # whole edges, in order, all of them
            samples = off[cut[1:]].astype(np.int64) - off[cut[:-1]].astype(np.int64)
            assert samples.sum() == n
            if E >= 8 * g and lens.max() <= 64:                      # many short edges: balanced to within one edge per share
                assert samples.max() - samples.min() <= 2 * 64
