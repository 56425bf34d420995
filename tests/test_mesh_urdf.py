"""`<mesh filename= scale=>` collision geometry, through both host layers.

What the reference does with it (src/collision/urdf.rs:54-71, src/collision/mesh.rs:39-69): the file
name goes through expand_package_path (relative names are taken from the URDF's directory); a missing
or unreadable file is logged and the collision element skipped; assimp's f32 vertices are scaled per
axis in f64; faces that are not triangles are dropped; every sub-mesh's indices continue after the
previous sub-mesh's vertices.  assimp itself is not in this image: the loader here reads Wavefront OBJ
(the triangle subset of what assimp accepts), and these tests pin the semantics above on files they
write.  CPU tests compare the Python reader and the C++ reader (libgear_host.so) field by field; the
GPU tests check flags against the oracle on a robot and an obstacle set that carry such meshes."""
import copy
import os

import numpy as np
import pytest

from gear_amd import descriptors as D
from gear_amd import host_binding as H
from gear_amd import urdf
from gear_amd import workloads as W
from tests.conftest import load_golden

SCALE = (0.5, 2.0, 1.5)


def write_obj(path, verts, tris, split=None, quad=True):
    """A closed convex mesh as OBJ text: two `o` objects (faces split between them, one vertex list as
    the format has it), `v/vt/vn` index tokens, one face written with negative (relative) indices, and a
    quad face that the reader must drop."""
    split = len(tris) // 2 if split is None else split
    lines = ["# written by tests/test_mesh_urdf.py", "o upper"]
    lines += ["v %.9f %.9f %.9f" % tuple(v) for v in verts]
    lines += ["vn 0 0 1", "vt 0 0"]
    for k, t in enumerate(tris):
        if k == split:
            lines.append("o lower")
        if k == 0:
            lines.append("f " + " ".join(str(i - len(verts)) for i in t))          # relative indices
        elif k % 2:
            lines.append("f " + " ".join(f"{i + 1}/1/1" for i in t))
        else:
            lines.append("f " + " ".join(f"{i + 1}//1" for i in t))
    if quad:
        lines.append("f 1 2 3 4")                                                     # not a triangle: dropped
        lines.append("f 1 2")                                                         # a line element: dropped
    with open(path, "w") as fh:
        fh.write("\n".join(lines) + "\n")


@pytest.fixture()
def hull():
    v, t = W.convex_mesh(np.random.default_rng(5), (0.1, 0.03, 0.05), 12, 12)
    return np.asarray(v), t


def mesh_robot(tmp_path, hull, missing=True):
    """The reference's sample robot with the forearm-side link geometry replaced by the mesh file (named
    relative to the URDF), and a second collision element on that link whose file does not exist."""
    os.makedirs(tmp_path / "meshes", exist_ok=True)
    write_obj(tmp_path / "meshes" / "part.obj", hull[0] / np.asarray(SCALE), hull[1])
    d = copy.deepcopy(load_golden("sample_robot.json"))
    edited = []
    for l in d["links"]:
        if l["name"] in ("l_elbow1", "l_wrist1", "l_shoulder2") and l["collisions"]:
            g = l["collisions"][0]["geometry"]
            g.update(kind="mesh", filename="meshes/part.obj", scale=list(SCALE))
            if missing:
                c2 = copy.deepcopy(l["collisions"][0])
                c2["geometry"]["filename"] = "meshes/not_there.obj"
                l["collisions"].insert(0, c2)
            edited.append(l["name"])
    assert edited, [l["name"] for l in d["links"]]
    p = tmp_path / "mesh_robot.urdf"
    p.write_text(urdf.robot_to_xml(urdf.robot_from_dict(d)))
    return str(p), edited


def test_obj_reader_semantics(tmp_path, hull):
    verts, tris = hull
    p = tmp_path / "m.obj"
    write_obj(p, verts, tris)
    v, t = urdf.load_obj_mesh(str(p), SCALE)
    want = np.asarray([[float(np.float32(float("%.9f" % c))) for c in row] for row in verts]) * np.asarray(SCALE)
    np.testing.assert_array_equal(np.asarray(v), want)           # f32 coordinate, then the f64 per-axis scale
    assert not np.array_equal(want, verts * np.asarray(SCALE))   # ... which differs from scaling the f64 text
    assert [list(x) for x in t] == [list(x) for x in tris]       # both objects, file order; quad and line dropped
    v1, _ = urdf.load_obj_mesh(str(p))
    np.testing.assert_array_equal(np.asarray(v1) * np.asarray(SCALE), want)


def test_urdf_mesh_python_and_cpp_agree(tmp_path, hull):
    path, edited = mesh_robot(tmp_path, hull)
    robot = urdf.read_urdf(path)
    assert robot.base_dir == str(tmp_path)
    model = D.CollisionModel.from_urdf_robot(robot, 0.01)
    chain = urdf.Chain.from_urdf(robot)
    d = H.HostPlanner(path, 0.01, 64).describe()
    l2j = urdf.link_to_joint_map(robot)
    n_mesh = 0
    for got, node in zip(d["nodes"], chain.nodes):
        specs = model.name_collision_model_map.get(node.name, [])
        assert len(got["shapes"]) == len(specs), node.name
        for gs, ws in zip(got["shapes"], specs):
            assert gs["kind"] == ws.kind
            np.testing.assert_array_equal(gs["t"], ws.t)
            np.testing.assert_array_equal(gs["q"], ws.q)
            if ws.kind == D.GH_SHAPE_MESH:
                n_mesh += 1
                np.testing.assert_array_equal(np.asarray(gs["vertices"]).reshape(-1, 3), np.asarray(ws.vertices))
                assert np.asarray(gs["triangles"]).reshape(-1, 3).tolist() == [list(x) for x in ws.triangles]
                np.testing.assert_allclose(np.asarray(ws.vertices), hull[0], rtol=0, atol=2e-8)
            else:
                np.testing.assert_array_equal(gs["params"], ws.params)
    assert n_mesh == len(edited)
    # the element whose file is missing is skipped, the link's other element stays (urdf.rs:60-63)
    for name in edited:
        kinds = [s.kind for s in model.name_collision_model_map[l2j[name]]]
        assert kinds.count(D.GH_SHAPE_MESH) == 1


def test_link_whose_only_mesh_is_missing_has_no_entry(tmp_path):
    p = tmp_path / "r.urdf"
    p.write_text("""<robot name="r">
  <link name="a"><collision><geometry><mesh filename="nowhere.obj"/></geometry></collision></link>
  <link name="b"><collision><origin xyz="0 0 1"/><geometry><mesh filename="empty.obj" scale="1 1 1"/></geometry></collision>
     <collision><geometry><sphere radius="0.1"/></geometry></collision></link>
  <joint name="j" type="revolute"><parent link="a"/><child link="b"/><axis xyz="0 0 1"/><limit lower="-1" upper="1"/></joint>
</robot>""")
    (tmp_path / "empty.obj").write_text("# no geometry\n")        # load_mesh fails: logged and skipped (urdf.rs:66-69)
    robot = urdf.read_urdf(str(p))
    model = D.CollisionModel.from_urdf_robot(robot, 0.0)
    assert "root" not in model.name_collision_model_map or model.name_collision_model_map["root"] == []
    assert [s.kind for s in model.name_collision_model_map["j"]] == [D.GH_SHAPE_BALL]
    d = H.HostPlanner(str(p), 0.0).describe()
    assert d["nodes"][0]["shapes"] == [] and [s["kind"] for s in d["nodes"][1]["shapes"]] == [D.GH_SHAPE_BALL]


ROCK_URDF = """<robot name="rock"><link name="rock"><collision>
  <origin xyz="0.45 0.25 0.55" rpy="0.3 0.2 0.1"/><geometry><mesh filename="%s" scale="1 1.5 1"/></geometry>
</collision></link></robot>"""


def test_obstacle_file_meshes_have_no_base_dir(tmp_path, hull, monkeypatch):
    """`Compound::from_urdf_robot` passes base_dir = None (collision_checker.rs:358): a relative mesh name
    in an obstacle file is looked up from the working directory, not next to the URDF."""
    write_obj(tmp_path / "rock.obj", hull[0] * 3.0, hull[1])
    (tmp_path / "rel.urdf").write_text(ROCK_URDF % "rock.obj")
    (tmp_path / "abs.urdf").write_text(ROCK_URDF % str(tmp_path / "rock.obj"))
    other = tmp_path / "elsewhere"
    other.mkdir()
    monkeypatch.chdir(other)
    assert D.Compound.from_urdf_file(str(tmp_path / "rel.urdf")).shapes == []
    assert len(H.HostCompound.from_urdf_file(str(tmp_path / "rel.urdf"))) == 0
    a = D.Compound.from_urdf_file(str(tmp_path / "abs.urdf"))
    assert [s.kind for s in a.shapes] == [D.GH_SHAPE_MESH] and len(H.HostCompound.from_urdf_file(str(tmp_path / "abs.urdf"))) == 1
    np.testing.assert_array_equal(np.asarray(a.shapes[0].vertices)[:, 1],
                                  np.asarray(urdf.load_obj_mesh(str(tmp_path / "rock.obj"))[0])[:, 1] * 1.5)
    monkeypatch.chdir(tmp_path)                                   # ... and is found when the caller stands next to it
    assert len(D.Compound.from_urdf_file("rel.urdf").shapes) == 1
    assert len(H.HostCompound.from_urdf_file("rel.urdf")) == 1


# ---------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("prec", [32, 64])
def test_gpu_mesh_urdf_flags_match_oracle(tmp_path, hull, obstacles_robot, prec):
    from gear_amd import engine
    from oracle import binding as orc
    from tests.parity import check_flags
    path, edited = mesh_robot(tmp_path, hull)
    robot = urdf.read_urdf(path)
    # obstacles: the reference's obstacles.urdf plus a mesh obstacle file read through the same code
    write_obj(tmp_path / "rock.obj", hull[0] * 3.0, hull[1])
    (tmp_path / "rock.urdf").write_text(ROCK_URDF % str(tmp_path / "rock.obj"))
    rock = D.Compound.from_urdf_robot(urdf.read_urdf(str(tmp_path / "rock.urdf")))
    base = D.Compound.from_urdf_robot(obstacles_robot)
    objects = D.Compound(list(base.shapes) + list(rock.shapes))
    assert [s.kind for s in rock.shapes] == [D.GH_SHAPE_MESH]
    w = W.Workload("mesh_urdf", robot, objects, 0.01, self_pairs=[("root", "l_wrist_yaw"), ("root", "l_wrist_pitch")])
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    ora = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    ora64 = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, 64) if prec == 32 else None
    q = w.configs(4000, seed=61, dtype=hip.dtype, overshoot=0.02)
    rep, want = check_flags(f"mesh_urdf[f{prec}]", hip, ora, q, D.GH_CHECK_ENV | D.GH_CHECK_SELF, prec, ora64)
    assert 0.02 < ((want & 1) > 0).mean() < 0.98

    # the C++ host layer reads the same files itself and must come to the same flags
    pl = H.HostPlanner(path, 0.01, prec, self_pairs=["root:l_wrist_yaw", "root:l_wrist_pitch"])
    comp = H.HostCompound.from_urdf_file(str(tmp_path / "rock.urdf"))
    assert len(comp) == 1
    rock_only = W.Workload("mesh_urdf_rock", robot, rock, 0.01)
    hip_r = engine.Scene(rock_only.robot_keep, rock_only.obstacle_keep, rock_only.margin, prec)
    u = pl.using_joints()
    ok = pl.is_feasible_batch(u, q.astype(np.float64), comp)
    flags = hip_r.check_batch(q, D.GH_CHECK_ENV)
    assert (ok == (flags == 0)).all()
    assert 0 < ok.sum() < len(ok)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [32, 64])
def test_gpu_general_meshes_match_oracle(tmp_path, hull, sample_robot, obstacles_robot, prec):
    """A mesh that is not a closed convex polytope — an open shell read from an OBJ file, a non-convex star, a
    wavy sheet — is a triangle soup, as the reference's TriMesh is (urdf.rs:54-71): flags, and the per-obstacle
    link masks of colliding_link_names, against the oracle, with such meshes as obstacles and as link geometry."""
    from gear_amd import engine
    from oracle import binding as orc
    from tests.parity import check_flags, certain
    from tests.test_narrowphase_host import rand_soup
    verts, tris = hull
    write_obj(tmp_path / "open.obj", verts * 2.5, tris[:-3], quad=False)             # a hull with three faces missing
    v, t = urdf.load_obj_mesh(str(tmp_path / "open.obj"))
    rng = np.random.default_rng(17)
    base = D.Compound.from_urdf_robot(obstacles_robot)
    extra = [D.ShapeSpec(D.GH_SHAPE_MESH, (0.0, 0.0, 0.0), t=(0.5, 0.25, 0.5), q=urdf.quat_from_rpy(0.2, 0.4, 0.1), vertices=v, triangles=t)]
    for kind, pos in (("star", (0.45, -0.3, 0.7)), ("sheet", (0.3, 0.3, 0.95)), ("star", (0.6, 0.0, 0.3))):
        sv, st = rand_soup(rng, 0.18, kind)
        extra.append(D.ShapeSpec(D.GH_SHAPE_MESH, (0.0, 0.0, 0.0), t=pos, q=urdf.quat_from_rpy(*rng.uniform(-1, 1, 3)), vertices=sv, triangles=st))
    objects = D.Compound(list(base.shapes[:1]) + extra + list(base.shapes[1:]))      # soups between ordinary obstacles

    def edit(model):                                                                  # ... and one as link geometry
        sv, st = rand_soup(rng, 0.07, "sheet")
        model.name_collision_model_map.setdefault("l_wrist_pitch", []).append(D.ShapeSpec(D.GH_SHAPE_MESH, (0.0, 0.0, 0.0), t=(0.0, 0.0, 0.05), vertices=sv, triangles=st))
    w = W.Workload("general_meshes", sample_robot, objects, 0.01, self_pairs=[("root", "l_wrist_pitch"), ("l_shoulder_yaw", "l_wrist_pitch")],
                   model_edit=edit)
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    ora = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    ora64 = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, 64) if prec == 32 else None
    q = w.configs(3000, seed=62, dtype=hip.dtype, overshoot=0.02)
    both = D.GH_CHECK_ENV | D.GH_CHECK_SELF
    rep, want = check_flags(f"general_meshes[f{prec}]", hip, ora, q, both, prec, ora64)
    assert 0.05 < ((want & 1) > 0).mean() < 0.95
    for path in (D.GH_PATH_FUSED, D.GH_PATH_PIPELINE, D.GH_PATH_PIPELINE | D.GH_PATH_NO_GRID, D.GH_PATH_SMALL):
        assert (hip.check_batch(q, both | path) == hip.check_batch(q, both)).all(), hex(path)
    # per (obstacle, node) bits, indexed by the CALLER's obstacles: a mesh's triangles are OR-ed
    ql = q[:400]
    _, band = ora.check(ql, both | D.GH_CHECK_NO_LIMITS, threads=16)
    sure = certain(band, prec)
    got, ref = hip.link_hits(ql, D.GH_CHECK_NO_LIMITS), ora.link_hits(ql, D.GH_CHECK_NO_LIMITS)
    assert got.shape == ref.shape == (400, len(objects.shapes), 1)
    assert (got[sure] == ref[sure]).all() and sure.mean() > 0.9
    assert got[:, 1:5].any()                                                          # the soups are hit
    assert (hip.self_hits(ql, D.GH_CHECK_NO_LIMITS)[sure] == ora.self_hits(ql, D.GH_CHECK_NO_LIMITS)[sure]).all()


@pytest.mark.gpu
def test_gpu_mesh_capacity_is_refused_by_name(sample_robot):
    """Every triangle of a general mesh is a device shape: the obstacle capacity (32767) counts them."""
    from gear_amd import engine
    n = 182                                                                           # 2 * 181 * 181 > 32767 triangles
    xs = np.linspace(-1, 1, n)
    verts = [[x, y, 0.1 * np.sin(3 * x) * np.cos(2 * y)] for x in xs for y in xs]
    tris = []
    for i in range(n - 1):
        for j in range(n - 1):
            a, b, c, d = i * n + j, i * n + j + 1, (i + 1) * n + j, (i + 1) * n + j + 1
            tris += [[a, b, c], [b, d, c]]
    objects = D.Compound([D.ShapeSpec(D.GH_SHAPE_MESH, (0.0, 0.0, 0.0), vertices=verts, triangles=tris)])
    w = W.Workload("huge_mesh", sample_robot, objects, 0.01)
    with pytest.raises(engine.GearHipError, match="32767 obstacle shapes"):
        engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32)
