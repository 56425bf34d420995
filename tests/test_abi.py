"""The C-ABI library loads, exports every entry point include/gear_hip.h declares, and fails
loudly (no CPU fallback) when no HIP device is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from gear_amd import descriptors as D
from gear_amd import engine, urdf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gear_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gh_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = engine.lib()
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"libgear_hip.so lacks {n}"
    assert set(engine.SYMBOLS) == set(names)
    assert b"gfx950" in lib.gh_version()


def test_struct_layouts_match_header():
    # sizes the C compiler gives the descriptors (x86-64 SysV): guards the ctypes mirror
    assert C.sizeof(D.gh_node_desc) == 8 + 8 * 10 + 8 + 8 * 5
    assert C.sizeof(D.gh_shape_desc) == 8 + 8 * 10 + 8
    assert C.sizeof(D.gh_mesh_desc) == 24
    assert C.sizeof(D.gh_robot_desc) == 24 + 5 * 8
    assert C.sizeof(D.gh_obstacle_desc) == 8 + 2 * 8
    assert C.sizeof(D.gh_counters) == 64


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="only meaningful without a GPU")
def test_no_device_is_a_loud_error(sample_robot, obstacles_robot):
    chain = urdf.Chain.from_urdf(sample_robot)
    model = D.CollisionModel.from_urdf_robot(sample_robot, 0.01)
    rk = D.flatten_robot(chain, model, chain.movable_nodes())
    ok = D.flatten_obstacles(D.Compound.from_urdf_robot(obstacles_robot))
    with pytest.raises(engine.GearHipError) as e:
        engine.Scene(rk, ok, 0.01, 32)
    assert e.value.code == -6 and "no CPU path" in str(e.value)


def test_bad_arguments_are_rejected_without_a_device():
    lib = engine.lib()
    out = C.c_void_p()
    assert lib.gh_scene_create(None, None, 0.01, 32, 0, C.byref(out)) == -1
    assert b"null" in lib.gh_last_error()
    assert lib.gh_check_batch(None, None, 4, 3, None) == -1
    assert lib.gh_scene_dof(None) == -1


def test_constants_match_header():
    """Every GH_* enumerator the Python binding names has the header's value."""
    import re
    from gear_amd import descriptors as D
    text = open(os.path.join(ROOT, "include", "gear_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    header = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"\b(GH_[A-Z0-9_]+)\s*=\s*(-?(?:0x[0-9a-fA-F]+|\d+))", text)}
    header.update({m.group(1): int(m.group(2), 0) for m in re.finditer(r"#define\s+(GH_[A-Z0-9_]+)\s+(0x[0-9a-fA-F]+|\d+)u?\b", text)})
    assert {"GH_CHECK_ENV", "GH_CHECK_SELF", "GH_CHECK_NO_LIMITS", "GH_PATH_SMALL", "GH_PATH_FUSED", "GH_PATH_PIPELINE",
            "GH_PATH_NO_GRID", "GH_COUNT_CANDIDATES", "GH_FLAG_ENV", "GH_FLAG_SELF", "GH_FLAG_LIMIT", "GH_SHAPE_MESH"} <= set(header)
    named = {k: v for k, v in vars(D).items() if k.startswith("GH_") and isinstance(v, int)}
    assert len(named) >= 12
    for k, v in named.items():
        assert header.get(k) == v, (k, v, header.get(k))
