"""BASELINE.json's full sizes through the C ABI, checked by size-independent properties (the oracle
needs ~10 ms per configuration, so it only spot-checks): the large-batch path — chunks rotating over
several streams and workspace sets — must equal the fused kernel, commute with a permutation of the
batch, not depend on chunk boundaries or on what the workspace held before, and first_bad must be the
first flagged sample of every edge."""
import os

import numpy as np
import pytest

from gear_amd import descriptors as D
from gear_amd import engine
from gear_amd import workloads as W
from oracle import binding as orc
from tests.parity import adjudicate, compare_flags, record

pytestmark = pytest.mark.gpu

BOTH = D.GH_CHECK_ENV | D.GH_CHECK_SELF


@pytest.mark.parametrize("prec", [32, 64])
def test_metric_scene_multi_chunk_batch(prec):
    """1M + 3 configurations of the metric scene: one call of the large-batch path with a ragged tail."""
    w = W.metric_workload()
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    n = (1 << 20) + 3
    q = w.configs(n, seed=31, dtype=hip.dtype, overshoot=0.005)
    got = hip.check_batch(q, BOTH)                           # default path: pipeline, several streams
    assert (got == hip.check_batch(q, BOTH | D.GH_PATH_FUSED)).all()
    # a permutation of the batch permutes the flags (other chunk boundaries, other wave mates)
    perm = np.random.default_rng(5).permutation(n)
    assert (hip.check_batch(np.ascontiguousarray(q[perm]), BOTH) == got[perm]).all()
    # a smaller call afterwards reuses the workspace sets: stale queue contents must not leak
    m = 70001
    assert (hip.check_batch(q[:m], BOTH) == got[:m]).all()
    assert (hip.check_batch(q[:m], D.GH_CHECK_SELF) == (got[:m] & ~np.uint8(D.GH_FLAG_ENV))).all()
    assert (hip.check_batch(q[:m], D.GH_CHECK_ENV) == (got[:m] & ~np.uint8(D.GH_FLAG_SELF))).all()
    rates = [float(((got & b) != 0).mean()) for b in (1, 2, 4)]
    assert rates[0] > 0.3 and rates[1] > 0.1 and rates[2] > 0.001, rates
    # the oracle on a bounded sample spread over all chunks
    ora = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    idx = np.linspace(0, n - 1, 1024).astype(np.int64)
    qs = np.ascontiguousarray(q[idx])
    want, band = ora.check(qs, BOTH, threads=16)
    rep = compare_flags(got[idx], want, band, prec)
    if prec == 32:
        adjudicate(rep, got[idx], want, qs, BOTH, orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, 64))
    record(f"metric_scene_multi_chunk_batch[f{prec}]", rep)
    assert rep["outside"] == 0, rep


def test_config1_sample_urdf_1m(sample_robot, obstacles_robot):
    """configs[1]: sample.urdf (8 columns, mimic gripper) vs obstacles.urdf, 1M random configurations."""
    pairs = [("root", "l_wrist_yaw"), ("root", "l_wrist_pitch"), ("l_shoulder_yaw", "l_wrist_pitch")]
    w = W.sample_workload(sample_robot, obstacles_robot, 0.01, self_pairs=pairs)
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32)
    n = 1 << 20
    q = w.configs(n, seed=0, overshoot=0.002)
    got = hip.check_batch(q, BOTH)
    assert (got == hip.check_batch(q, BOTH | D.GH_PATH_FUSED)).all()
    assert 0.05 < ((got & 1) != 0).mean() < 0.4            # SURVEY 8d expects about 14 %
    lim = (got & 4) != 0
    assert lim.any() and (got[lim] == 4).all()             # a limit violation reports nothing else
    ora = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, 32)
    want, band = ora.check(q[:20000], BOTH, threads=16)
    rep = compare_flags(got[:20000], want, band, 32)
    adjudicate(rep, got[:20000], want, q[:20000], BOTH, orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, 64))
    record("config1_sample_urdf_1m[f32]", rep)
    assert rep["outside"] == 0, rep


def test_config2_64k_edges():
    """configs[2]: 64k straight segments of 8..64 samples (about 2.3M samples) in one gh_check_edges call."""
    w = W.ur10_workload(256)
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32)
    q, off = w.edges_lockstep(65536, seed=9, min_len=8, max_len=64)
    assert len(q) > 2_000_000
    bad = D.GH_FLAG_ENV | D.GH_FLAG_LIMIT
    flags, first = hip.check_edges(q, off, D.GH_CHECK_ENV, bad)
    hit = (flags & bad) != 0
    # first_bad == first flagged sample of the edge, -1 if none (vectorised: running count of hits)
    csum = np.concatenate([[0], np.cumsum(hit)])
    any_hit = csum[off[1:]] > csum[off[:-1]]
    pos = np.searchsorted(csum, csum[off[:-1]] + 1, side="left") - 1   # index of the first hit at or after off[e]
    want_first = np.where(any_hit, pos - off[:-1], -1)
    assert (first == want_first).all()
    sub = slice(0, 300000)
    assert (flags[sub] == hip.check_batch(q[sub], D.GH_CHECK_ENV | D.GH_PATH_FUSED)).all()


def test_config3_full_shard_4096_obstacles():
    """configs[3]: 64M configurations against 4096 mixed obstacles over 8 GPUs, i.e. 8M per GPU — one GPU's
    shard in one call.  Size-independent properties: a permutation of the batch permutes the flags (other chunk
    boundaries, other wave mates), a slice evaluated on its own gives the slice of the flags, and the fused
    kernel's brute-force L x O sweep agrees on a sample spread over the whole shard."""
    w = W.sawyer_workload(4096)
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32)
    n = 1 << 23
    q = w.configs(n, seed=45, overshoot=0.002)
    got = hip.check_batch(q, BOTH)
    rates = [float(((got & b) != 0).mean()) for b in (1, 2, 4)]
    assert 0.5 < rates[0] < 0.95 and rates[1] > 0.05 and rates[2] > 0.0005, rates
    rng = np.random.default_rng(6)
    perm = rng.permutation(n)
    assert (hip.check_batch(np.ascontiguousarray(q[perm]), BOTH) == got[perm]).all()
    lo = 3 * (1 << 20) + 17
    assert (hip.check_batch(q[lo:lo + 2_500_001], BOTH) == got[lo:lo + 2_500_001]).all()
    idx = np.sort(rng.choice(n, 60000, replace=False))
    assert (hip.check_batch(np.ascontiguousarray(q[idx]), BOTH | D.GH_PATH_FUSED) == got[idx]).all()


def _scene_with_env(w, prec, **env):
    """Tuning hooks are read once, when a scene is created."""
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.mark.parametrize("prec", [32, 64])
def test_full_exact_regions_are_rechecked(prec):
    """GEAR_HIP_EXACT_CAP=2: a sweep wavefront may hand only two items to the exact stage, so nearly
    every wavefront overflows and its configurations are decided by the fused re-check inside the
    pipeline's last stage; with larger capacities only some wavefronts do (the last stage looks at 64 marks per
    workgroup and evaluates the marked blocks one after the other).  The flags must not depend on that capacity."""
    for w in (W.metric_workload(), W.ur10_workload(128)):
        n = 100000 if prec == 32 else 40037
        n_waves = (n + 63) // 64
        ref = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
        q = w.configs(n, seed=35, dtype=ref.dtype, overshoot=0.01)
        want = ref.check_batch(q, BOTH | D.GH_PATH_FUSED)
        assert (want == ref.check_batch(q, BOTH)).all()
        ref.check_batch(q, BOTH | D.GH_COUNT_CANDIDATES)
        assert ref.pipeline_last_stats()["redo_waves"] == 0           # the default capacity is not reached here
        scattered = False
        for cap in (2, 24, 48, 96):
            hip = _scene_with_env(w, prec, GEAR_HIP_EXACT_CAP=cap)
            got = hip.check_batch(q, BOTH | D.GH_COUNT_CANDIDATES)
            st = hip.pipeline_last_stats()
            assert st["exact_cap"] == cap, st
            if cap == 2:
                assert st["redo_waves"] > n_waves // 16, st
            scattered = scattered or 0 < st["redo_waves"] < n_waves // 2
            assert (got == want).all(), (w.name, cap, st)
        assert scattered, w.name


def test_chunks_over_several_streams():
    """From 6M configurations on a call is cut into equal chunks, a multiple of the stream count, that rotate
    over the caller's stream and the scene's own: four chunks of 1.5M by default; 21 ragged ones over three
    streams and nine over one with the knobs forced.  Every cut must give the fused kernel's bytes, and so must a
    second call that finds the workspace sets used."""
    w = W.metric_workload()
    n = 6 * (1 << 20) + 5
    q = w.configs(n, seed=37, overshoot=0.005)
    hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32)
    ref = hip.check_batch(q, BOTH | D.GH_PATH_FUSED)
    got = hip.check_batch(q, BOTH)
    assert (got == ref).all()
    assert (hip.check_batch(q, BOTH) == ref).all()
    for env in ({"GEAR_HIP_PIPE_STREAMS": 3, "GEAR_HIP_PIPE_CHUNK": 300000},
                {"GEAR_HIP_PIPE_STREAMS": 1, "GEAR_HIP_PIPE_CHUNK": 700000},
                {"GEAR_HIP_PIPE_STREAMS": 2, "GEAR_HIP_N2_TILE": 8}):
        other = _scene_with_env(w, 32, **env)
        assert (other.check_batch(q, BOTH) == ref).all(), env


def test_sweep_workgroup_shapes_agree():
    """The sweep runs with 4, 2 or 1 wavefronts per workgroup and with the cull table in LDS or not,
    whichever keeps most wavefronts resident; every shape must give the same flags."""
    w = W.metric_workload()
    q = w.configs(70000, seed=36, overshoot=0.01)
    ref = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32).check_batch(q, BOTH)
    for env in ({"GEAR_HIP_SWEEP_WAVES": 1}, {"GEAR_HIP_SWEEP_WAVES": 2}, {"GEAR_HIP_SWEEP_WAVES": 4, "GEAR_HIP_SWEEP_CLDS": 0},
                {"GEAR_HIP_PIPE_CHUNK": 8192}):
        hip = _scene_with_env(w, 32, **env)
        assert (hip.check_batch(q, BOTH) == ref).all(), env


def test_exact_capacity_grows_when_regions_run_full():
    """A sweep wavefront whose exact region runs full says so in a host word, and the scene's later calls get twice the
    room: from a deliberately small start (GEAR_HIP_EXACT_CAP0=16) the capacity doubles call by call until no region
    is full any more; the flags are the same at every capacity."""
    w = W.metric_workload()
    q = w.configs(200000, seed=38, overshoot=0.01)
    ref = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32)
    want = ref.check_batch(q, BOTH | D.GH_PATH_FUSED)
    assert (ref.check_batch(q, BOTH | D.GH_PATH_PIPELINE) == want).all()
    assert ref.pipeline_last_stats()["exact_cap"] == 256                 # the default with meshes (URDF cylinders)
    hip = _scene_with_env(w, 32, GEAR_HIP_EXACT_CAP0=16)
    caps, full = [], []
    for _ in range(7):
        got = hip.check_batch(q, BOTH | D.GH_PATH_PIPELINE | D.GH_COUNT_CANDIDATES)
        assert (got == want).all()
        st = hip.pipeline_last_stats()
        caps.append(st["exact_cap"])
        full.append(st["redo_waves"])
    print("capacity per call", caps, "full regions", full)
    assert caps[0] == 16 and caps[1] == 32 and caps[2] == 64 and caps[-1] in (64, 128, 256)
    assert full[0] > 1000 and full[-1] == 0
    assert caps[-1] == caps[-2]                                          # it stops growing once nothing is full


@pytest.mark.parametrize("prec", [32, 64])
def test_exact_stage_windows_and_waiting_entries(prec):
    """The 6-DOF arm among 256 boxes and cylinders hands 1.9 entries per configuration to the exact stage.  With tiles
    of 32 sweep wavefronts a workgroup then holds ~3900 entries — many tiles more than one window of 4096 — and most
    windows hold enough later entries for them to wait for the first ones' answers: every such schedule (tile 8, 16, 32)
    must give the fused kernel's bytes."""
    w = W.ur10_workload(256)
    n = 200000 + 11
    ref = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    q = w.configs(n, seed=77, dtype=ref.dtype, overshoot=0.005)
    want = ref.check_batch(q, BOTH | D.GH_PATH_FUSED)
    ref.check_batch(q, BOTH | D.GH_COUNT_CANDIDATES)
    st = ref.pipeline_last_stats()
    assert st["exact_items"] > 1.2 * n and st["redo_waves"] == 0, st         # several entries per configuration
    for tile in (8, 16, 32):
        hip = _scene_with_env(w, prec, GEAR_HIP_N2_TILE=tile)
        assert (hip.check_batch(q, BOTH | D.GH_PATH_PIPELINE) == want).all(), tile
        assert (hip.check_batch(q, D.GH_CHECK_ENV | D.GH_PATH_PIPELINE) == (want & 5)).all(), tile
