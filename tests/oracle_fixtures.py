"""Committed oracle outputs for the large parity cases (tests/golden/oracle_*.npz).

The CPU oracle needs 10-150 ms per configuration on the big scenes; run live on the GPU box it bounded the
oracle-compared samples to a few hundred configurations and took three quarters of the GPU suite's time.  The
oracle is therefore run HERE, in the build container, by `python tests/golden/make_oracle_fixtures.py`, on the
seeded inputs described in SPECS; the GPU tests regenerate the same inputs from the same seeds (checked by a
digest), call the HIP path through the C ABI and compare with the stored answers.

These files are outputs of the builder's own oracle (oracle/gear_oracle.hpp), NOT of the reference: they pin the
GPU path to the oracle, and the oracle stays pinned to the reference only by tests/test_oracle_golden.py.

Per fixture: `flags` (all-pairs mode), `band` [n][2] (tests/parity.py), `digest` of the input bytes; f32 fixtures
also carry the f64 oracle's verdict on every configuration the f32 oracle leaves uncertified (`adj_idx`,
`adj_flags64`, `adj_band64`: the f64 oracle run on the same f32-valued rows), so that the adjudication of the
second band needs no oracle run either; edge fixtures add the sequential first-failure answers (`first_bad`,
`flags_faithful`, 0xFF = never evaluated by the reference)."""
import hashlib
import os

import numpy as np

from gear_amd import descriptors as D
from gear_amd import workloads as W

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BOTH = D.GH_CHECK_ENV | D.GH_CHECK_SELF
BAD = D.GH_FLAG_ENV | D.GH_FLAG_LIMIT

# name -> workload factory, precision, predicates, input recipe
SPECS = {}
for _p in (32, 64):
    SPECS[f"metric_f{_p}"] = dict(workload="metric", prec=_p, what=BOTH, kind="configs", n=16384, seed=2, overshoot=0.01)
    SPECS[f"config2_edges_f{_p}"] = dict(workload="ur10", prec=_p, what=D.GH_CHECK_ENV, kind="edges", n=256, seed=41,
                                         min_len=16, max_len=64)
    SPECS[f"cylinder_obstacles_f{_p}"] = dict(workload="ur10_64", prec=_p, what=BOTH, kind="configs", n=4096, seed=4, overshoot=0.0)
    SPECS[f"config3_4096_f{_p}"] = dict(workload="sawyer", prec=_p, what=BOTH, kind="configs", n=8192, seed=42, overshoot=0.0)
    SPECS[f"config4_dual_arm_f{_p}"] = dict(workload="pr2", prec=_p, what=BOTH, kind="configs", n=8192, seed=43, overshoot=0.01)
SPECS["config4_ik_seeds_f64"] = dict(workload="pr2_left_arm", prec=64, what=BOTH, kind="configs", n=8192, seed=44, overshoot=0.0)

_WORKLOADS = {
    "metric": lambda: W.metric_workload(),
    "ur10": lambda: W.ur10_workload(256),
    "ur10_64": lambda: W.ur10_workload(64),
    "sawyer": lambda: W.sawyer_workload(4096),
    "pr2": lambda: W.pr2_workload(128),
    "pr2_left_arm": lambda: W.pr2_workload(128, end_joint="l_gripper"),
}
_cache = {}


def workload(spec):
    k = spec["workload"]
    if k not in _cache:
        _cache[k] = _WORKLOADS[k]()
    return _cache[k]


def inputs(spec):
    """(q, edge_off or None) of a fixture, from its seeds."""
    w = workload(spec)
    dtype = np.float32 if spec["prec"] == 32 else np.float64
    if spec["kind"] == "edges":
        return w.edges(spec["n"], seed=spec["seed"], dtype=dtype, min_len=spec["min_len"], max_len=spec["max_len"])
    return w.configs(spec["n"], seed=spec["seed"], dtype=dtype, overshoot=spec["overshoot"]), None


def digest(q, off=None):
    h = hashlib.sha1(np.ascontiguousarray(q).tobytes())
    if off is not None:
        h.update(np.ascontiguousarray(off, np.uint32).tobytes())
    return h.hexdigest()


def path(name):
    return os.path.join(GOLDEN, f"oracle_{name}.npz")


def load(name):
    """The stored oracle answers of fixture `name` plus its regenerated inputs: dict(spec, w, q, off, flags, band, ...)."""
    spec = SPECS[name]
    if not os.path.exists(path(name)):   # a fixture still being generated must not take the rest of the suite down with it
        import pytest
        pytest.skip(f"tests/golden/oracle_{name}.npz is missing: python tests/golden/make_oracle_fixtures.py {name}")
    q, off = inputs(spec)
    z = np.load(path(name), allow_pickle=False)
    fx = {k: z[k] for k in z.files}
    assert str(fx["digest"]) == digest(q, off), f"{name}: the regenerated inputs differ from the ones the oracle was run on"
    fx.update(name=name, spec=spec, w=workload(spec), q=q, off=off)
    return fx


def generate(name, threads=8):
    """Runs the oracle for one fixture (build container only) and writes its file."""
    from oracle import binding as orc
    from tests.parity import BAND
    spec = SPECS[name]
    w = workload(spec)
    q, off = inputs(spec)
    prec, what = spec["prec"], spec["what"]
    ora = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    # in slices, each kept in a scratch file as it finishes: an interrupted run (an hour per precision for the
    # 4096-obstacle scene) resumes where it stopped
    n, step = q.shape[0], 512
    part = os.path.join(GOLDEN, f".partial_{name}.npz")
    flags, band, done = np.zeros(n, np.uint8), np.zeros((n, 2)), 0
    if os.path.exists(part):
        z = np.load(part, allow_pickle=False)
        if str(z["digest"]) == digest(q, off):
            flags, band, done = z["flags"].copy(), z["band"].copy(), int(z["done"])
    while done < n:
        hi = min(done + step, n)
        flags[done:hi], band[done:hi] = ora.check(np.ascontiguousarray(q[done:hi]), what, threads=threads)
        done = hi
        np.savez(part, flags=flags, band=band, done=np.array(done), digest=np.array(digest(q, off)))
        print(f"  {name}: {done}/{n}", flush=True)
    out = dict(flags=flags, band=band, digest=np.array(digest(q, off)))
    if prec == 32:   # the f64 oracle's verdict wherever the f32 oracle's own evidence is open (tests/parity.py)
        idx = np.nonzero(band[:, 1] - BAND[32] < 0)[0]
        o64 = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, 64)
        if len(idx):
            f64, b64 = o64.check(np.ascontiguousarray(q[idx]).astype(np.float64), what, threads=threads)
        else:
            f64, b64 = np.zeros(0, np.uint8), np.zeros((0, 2))
        out.update(adj_idx=idx.astype(np.int64), adj_flags64=f64, adj_band64=b64[:, 0] if len(idx) else np.zeros(0))
    if spec["kind"] == "edges":
        ff, fb = ora.check_edges(q, off, what, BAD)
        out.update(flags_faithful=ff, first_bad=fb)
    np.savez_compressed(path(name), **out)
    if os.path.exists(part):
        os.remove(part)
    return out
