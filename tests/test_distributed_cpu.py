"""N>1 path on CPU: world_size-2 gloo.  Each rank checks its shard (the oracle stands in for the
GPU here — this test is about sharding and the result gather, not the kernels) and the gathered
flags / first-bad indices must equal the single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gear_amd import descriptors as D
from gear_amd import sharding, urdf
from gear_amd import workloads as W
from tests.conftest import load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _workload():
    return W.sample_workload(urdf.robot_from_dict(load_golden("sample_robot.json")),
                             urdf.robot_from_dict(load_golden("obstacles.json")), 0.01)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as orc
    w = _workload()
    ora = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, 32)
    # --- independent configurations ---
    q = w.configs(1001, seed=5, overshoot=0.02)
    sizes = [b - a for a, b in (sharding.shard_bounds(len(q), world, r) for r in range(world))]
    lo, hi = sharding.shard_bounds(len(q), world, rank)
    local = torch.from_numpy(ora.check(q[lo:hi], D.GH_CHECK_ENV, want_band=False))
    flags = sharding.all_gather_ragged(dist, local, sizes)
    counts = torch.tensor([int((local & 1).ne(0).sum()), int((local & 4).ne(0).sum())])
    dist.all_reduce(counts)
    # --- edges: whole edges per shard ---
    qe, off = w.edges(60, seed=6, min_len=3, max_len=30)
    e_lo, e_hi, s_lo, s_hi = sharding.shard_edges(off, world, rank)
    loc_off = (off[e_lo:e_hi + 1].astype(np.int64) - s_lo).astype(np.uint32)
    _, fb = ora.check_edges(qe[s_lo:s_hi], loc_off)
    esizes = []
    for r in range(world):
        a, b, _, _ = sharding.shard_edges(off, world, r)
        esizes.append(b - a)
    first_bad = sharding.all_gather_ragged(dist, torch.from_numpy(fb), esizes)
    if rank == 0:
        np.savez(os.path.join(out_dir, "gathered.npz"), flags=flags.numpy(), counts=counts.numpy(),
                 first_bad=first_bad.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "gathered.npz")
    from oracle import binding as orc
    w = _workload()
    ora = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, 32)
    q = w.configs(1001, seed=5, overshoot=0.02)
    want = ora.check(q, D.GH_CHECK_ENV, want_band=False)
    assert (got["flags"] == want).all()
    assert got["counts"][0] == ((want & 1) != 0).sum() and got["counts"][1] == ((want & 4) != 0).sum()
    qe, off = w.edges(60, seed=6, min_len=3, max_len=30)
    _, fb = ora.check_edges(qe, off)
    assert (got["first_bad"] == fb).all()
    assert (fb >= 0).any() and (fb < 0).any()
