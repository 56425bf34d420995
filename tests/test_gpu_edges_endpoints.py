"""gh_check_edges_from_endpoints: the device replays rrt's stepping from the two end points of every edge.
The samples it generates must be bit-identical to the host stepping's (gear_host.hpp connect_line / walk_line,
restating rrt 0.4.0's Tree::connect and smooth_path; joint_path_planner.rs:168-188 is where the reference hands
its closure to them), and first_bad must be what gh_check_edges returns for those samples."""
import numpy as np
import pytest
import torch

from gear_amd import descriptors as D
from gear_amd import engine
from gear_amd import host_binding as H
from gear_amd import workloads as W

pytestmark = pytest.mark.gpu

BAD = D.GH_FLAG_ENV | D.GH_FLAG_LIMIT


def _ends(w, n, dtype, seed):
    a, b = w.configs(n, seed=seed, dtype=dtype), w.configs(n, seed=seed + 1, dtype=dtype)
    b[:40] = a[:40] + dtype(0.01)     # shorter than one step
    b[40:50] = a[40:50]               # zero length
    return a, b


@pytest.mark.parametrize("prec", [32, 64])
@pytest.mark.parametrize("mode", [D.GH_EDGE_CONNECT, D.GH_EDGE_WALK])
def test_device_samples_are_the_host_steppings_bits(prec, mode):
    """configs[2] size: 65536 edges of the 6-DOF arm (about 4M samples).  Device arrays in and out."""
    w = W.ur10_workload(256)
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    E = 65536
    a, b = _ends(w, E, hip.dtype, seed=61)
    q_ref, off_ref = W.edge_samples_lockstep(a, b, 0.1, mode)            # numpy, the scalar loops' arithmetic
    n = int(off_ref[-1])
    assert n > 3_000_000
    # the numpy stepping is itself the C++ host layer's, bit for bit (2000 edges of it)
    for e in list(range(60)) + list(range(60, E, E // 1900)):
        ref = H.edge_line(a[e], b[e], 0.1, prec, mode)
        assert (ref == q_ref[off_ref[e]:off_ref[e + 1]].astype(np.float64)).all() and len(ref) == off_ref[e + 1] - off_ref[e]
    dev = torch.device("cuda", 0)
    tdt = torch.float32 if prec == 32 else torch.float64
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    cap = n + 1000
    q_t = torch.full((cap, hip.dof), float("nan"), dtype=tdt, device=dev)
    off_t = torch.zeros(E + 1, dtype=torch.int32, device=dev)
    cnt_t = torch.zeros(E, dtype=torch.int32, device=dev)
    flags_t = torch.full((cap,), 0xEE, dtype=torch.uint8, device=dev)
    first_t = torch.zeros(E, dtype=torch.int32, device=dev)
    total = hip.check_edges_from_endpoints_dev(ta, tb, q_t, off_t, cnt_t, flags_t, first_t, 0.1, 4096, mode, D.GH_CHECK_ENV, BAD)
    torch.cuda.synchronize()
    assert total == n
    assert (off_t.cpu().numpy().astype(np.uint32) == off_ref).all()
    assert (cnt_t.cpu().numpy() == (off_ref[1:] - off_ref[:-1]).astype(np.int32)).all()
    got_q = q_t[:n].cpu().numpy()
    assert got_q.tobytes() == q_ref.tobytes()                              # bit for bit, signed zeros included
    assert torch.isnan(q_t[n:]).all() and (flags_t[n:] == 0xEE).all()      # nothing beyond the samples is written
    # flags and first_bad: what gh_check_edges says about the same samples
    flags, first = hip.check_edges(q_ref, off_ref, D.GH_CHECK_ENV, BAD)
    assert (flags_t[:n].cpu().numpy() == flags).all()
    assert (first_t.cpu().numpy() == first).all()
    assert (first >= 0).sum() > 1000 and (first < 0).sum() > 50
    # the host-buffer form: end points over PCIe, results back
    fb, ns = hip.check_edges_from_endpoints(a, b, 0.1, 4096, mode, D.GH_CHECK_ENV, BAD)
    assert (fb == first).all() and (ns == off_ref[1:] - off_ref[:-1]).all()
    if mode == D.GH_EDGE_WALK:
        assert (ns[:50] == 0).all() and (fb[:50] == -1).all()             # shorter than a step: nothing to evaluate
    else:
        assert (ns[:50] == 1).all()                                        # one sample: the target itself


def test_truncation_small_batches_and_errors():
    w = W.metric_workload(64)
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32)
    a, b = _ends(w, 300, np.float32, seed=63)
    q_ref, off_ref = W.edge_samples_lockstep(a, b, 0.1, D.GH_EDGE_CONNECT)
    full = off_ref[1:] - off_ref[:-1]
    # max_len = 7: longer edges are cut after 7 samples and say so in bit 31
    fb, ns = hip.check_edges_from_endpoints(a, b, 0.1, 7, D.GH_EDGE_CONNECT)
    cut = full > 7
    assert cut.any() and (~cut).any()
    assert (ns[~cut] == full[~cut]).all() and (ns[cut] == (7 | D.GH_EDGE_TRUNCATED)).all()
    q7, off7 = W.edge_samples_lockstep(a, b, 0.1, D.GH_EDGE_CONNECT, max_len=7)
    assert (fb == hip.check_edges(q7, off7)[1]).all()
    # one edge, two edges (the small-batch kernel underneath), none
    for k in (1, 2):
        fb, ns = hip.check_edges_from_endpoints(a[60:60 + k], b[60:60 + k], 0.1, 4096, D.GH_EDGE_CONNECT)
        qk, offk = W.edge_samples_lockstep(a[60:60 + k], b[60:60 + k], 0.1, D.GH_EDGE_CONNECT)
        assert (ns == offk[1:] - offk[:-1]).all() and (fb == hip.check_edges(qk, offk)[1]).all()
    fb, ns = hip.check_edges_from_endpoints(a[:0], b[:0])
    assert fb.shape == (0,) and ns.shape == (0,)
    # NaN end point: every sample is NaN, the limit gate flags the first one; the edge never "arrives"
    an = a[:3].copy()
    an[1, 2] = np.nan
    fb, ns = hip.check_edges_from_endpoints(an, b[60:63], 0.1, 50, D.GH_EDGE_CONNECT)
    assert ns[1] == (50 | D.GH_EDGE_TRUNCATED) and fb[1] == 0
    for bad_kw in (dict(step=0.0), dict(max_len=0), dict(mode=7)):
        with pytest.raises(engine.GearHipError):
            hip.check_edges_from_endpoints(a[:4], b[:4], **bad_kw)
    # a device group shares the edges
    grp = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32, devices=[0, 0, 0])
    a2, b2 = _ends(w, 9000, np.float32, seed=65)
    f1, n1 = hip.check_edges_from_endpoints(a2, b2, 0.1, 4096, D.GH_EDGE_WALK)
    f2, n2 = grp.check_edges_from_endpoints(a2, b2, 0.1, 4096, D.GH_EDGE_WALK)
    assert (f1 == f2).all() and (n1 == n2).all()


def test_planner_bulk_validation_from_endpoints(tmp_path):
    """JointPathPlanner::first_infeasible_from_endpoints (C++ host layer) against its own line-by-line evaluation."""
    from gear_amd import urdf
    robot = W.arm7_robot()
    p = tmp_path / "arm7.urdf"
    p.write_text(urdf.robot_to_xml(robot))
    field = W.obstacle_field(96, 5, ("box", "ball"), (0.6, 0.4))
    obs = H.HostCompound.from_specs(field.shapes)
    pl = H.HostPlanner(str(p), 0.01, 32, step_length=0.1)
    u = pl.using_joints()
    w = W.Workload("arm7_field", robot, field, 0.01)
    a, b = w.configs(5000, seed=66), w.configs(5000, seed=67)
    first, cnt = pl.first_infeasible_from_endpoints(u, a, b, obs, D.GH_EDGE_CONNECT)
    q, off = W.edge_samples_lockstep(a, b, 0.1, D.GH_EDGE_CONNECT)
    assert (cnt == off[1:] - off[:-1]).all()
    ok = pl.is_feasible_batch(u, q.astype(np.float64), obs)
    o = off.astype(np.int64)
    want = np.array([next((j - o[e] for j in range(o[e], o[e + 1]) if not ok[j]), -1) for e in range(5000)])
    assert (first == want).all()


@pytest.mark.parametrize("prec", [32, 64])
def test_every_column_count_steps_alike(prec):
    """The walker is compiled per column count up to 8 (registers) and once for any count beyond (scratch memory):
    7 columns (the metric arm) and 15 (the dual-arm robot) must give the host stepping's bits like the 6 above."""
    for w in (W.metric_workload(64), W.pr2_workload(32)):
        hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
        E = 3000
        a, b = _ends(w, E, hip.dtype, seed=71)
        for mode in (D.GH_EDGE_CONNECT, D.GH_EDGE_WALK):
            q_ref, off_ref = W.edge_samples_lockstep(a, b, 0.1, mode)
            n = int(off_ref[-1])
            first, cnt = hip.check_edges_from_endpoints(a, b, 0.1, 4096, mode, D.GH_CHECK_ENV, BAD)
            assert (cnt == (off_ref[1:] - off_ref[:-1])).all(), (w.name, mode)
            flags, want = hip.check_edges(q_ref, off_ref, D.GH_CHECK_ENV, BAD)
            assert (first == want).all(), (w.name, mode)                  # the same samples: the same first failures
            dev = torch.device("cuda", 0)
            q_t = torch.zeros((n + 8, hip.dof), dtype=torch.float32 if prec == 32 else torch.float64, device=dev)
            off_t = torch.zeros(E + 1, dtype=torch.int32, device=dev)
            cnt_t = torch.zeros(E, dtype=torch.int32, device=dev)
            flags_t = torch.zeros(n + 8, dtype=torch.uint8, device=dev)
            first_t = torch.zeros(E, dtype=torch.int32, device=dev)
            hip.check_edges_from_endpoints_dev(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), q_t, off_t, cnt_t, flags_t,
                                               first_t, 0.1, 4096, mode, D.GH_CHECK_ENV, BAD)
            torch.cuda.synchronize()
            assert q_t[:n].cpu().numpy().tobytes() == q_ref.tobytes(), (w.name, mode)
