#!/usr/bin/env python3
"""bench.py — collision-checked configs/sec on BASELINE.json's metric scene
(7-DOF arm vs 256 obstacles; env + self + limit flags), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (gh_check_batch_dev) over one batch of
synthetic joint configurations already resident in HBM; with N > 1 every rank
checks its own shard of the batch (scene replicated per GPU, no data-path
collective) and the per-config flag bytes are gathered over RCCL each step.
Rank 0 prints ONE JSON line.  At N = 1 the same line also carries the f64 figure,
a free-space-dominant variant of the scene and the CPU baseline.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

from gear_amd import descriptors as D
from gear_amd import engine
from gear_amd import workloads as W

VALU_PEAK_TFLOPS = {32: 157.3, 64: 78.6}   # MI355X vector peaks (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0
# committed rocprofv3 PMC summaries (bench.py itself cannot run under the profiler): fabric traffic per configuration
# (tools/pmc_traffic.py) and the issue-side counters of the dominant kernel (tools/attrib_sweep.sh); newest round first
def _first_existing(*names):
    for n in names:
        if os.path.exists(n):
            return n
    return names[-1]


TRAFFIC_JSON = _first_existing(os.path.join(ROOT, "profiles", "r03", "pipeline_traffic.json"),
                               os.path.join(ROOT, "profiles", "r02", "pipeline_traffic.json"))
ISSUE_JSON = os.path.join(ROOT, "profiles", "r03", "attrib_k_sweep_n1.json")
WHAT = D.GH_CHECK_ENV | D.GH_CHECK_SELF


def measured_traffic(batch):
    """Fabric bytes per step from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in
    separate runs, gfx950 correction applied; tools/pmc_traffic.py), scaled to this batch.  None when
    the summary is absent: bench.py itself cannot run under the profiler."""
    try:
        with open(TRAFFIC_JSON) as f:
            t = json.load(f)
        return float(t["total_bytes_per_config"]) * batch
    except (OSError, KeyError, ValueError):
        return None


def measured_issue():
    """VALU issue occupancy of the dominant kernel from the committed SQ-counter passes (a constant read from a
    file, like `traffic`: not measured in this run)."""
    try:
        with open(ISSUE_JSON) as f:
            t = json.load(f)
        return {k: t["issue"][k] for k in ("valu_issue_frac", "valu_busy_cost_weighted_estimate", "valu_insts_per_wave", "wave_cycles", "source")
                if k in t["issue"]}
    except (OSError, KeyError, ValueError):
        return None


def algorithmic_flops_per_config(w, counters):
    """F = F_fk + pairs*F_cull + sat*F_sat + exact*F_gjk with the measured
    per-config pair / SAT / exact-query counts of an instrumented run."""
    nodes = w.chain.nodes
    movable = sum(1 for n in nodes if n.is_movable)
    fixed = len(nodes) - movable
    L = w.robot_keep.desc.n_shapes
    f_fk = 94 * movable + 61 * fixed + 61 * L
    n = max(counters["configs"], 1)
    pairs, sat, exact = counters["pairs"] / n, counters["sat_tests"] / n, counters["exact_tests"] / n
    return f_fk + 11.0 * pairs + 220.0 * sat + 600.0 * exact, {
        "pairs_per_config": pairs, "cull_survivors_per_config": counters["cull_pass"] / n,
        "sat_per_config": sat, "exact_per_config": exact,
        "exact_iters_per_query": counters["exact_iters"] / max(counters["exact_tests"], 1), "F_fk": f_fk}


def device_configs(w, n, seed, dtype, device):
    """generate_random_joint_positions_from_limits (src/funcs.rs:76-87) on the device: (max-min)*u + min."""
    lo, hi = W.joint_limits(w.chain, w.using)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    u = torch.rand((n, w.dof), generator=g, device=device, dtype=torch.float64)
    lo_t = torch.tensor(lo, device=device, dtype=torch.float64)
    hi_t = torch.tensor(hi, device=device, dtype=torch.float64)
    return ((hi_t - lo_t) * u + lo_t).to(dtype).contiguous()


def host_threads():
    """Threads for the CPU baseline: every CPU this process may run on (sched_getaffinity), capped by the
    cgroup's CPU quota where one is set — threads beyond the quota are time-sliced, not added."""
    affinity = len(os.sched_getaffinity(0))
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            a, b = f.read().split()
            if a != "max":
                quota = max(1, int(float(a) / float(b) + 0.5))
    except (OSError, ValueError):
        pass
    threads = min(affinity, quota) if quota else affinity
    return max(1, min(threads, 1000)), affinity, quota


def cpu_baseline(w, precision, q_gpu=None, flags_gpu=None, seconds=12.0):
    """The reference's CPU path as restated by oracle/ ("port"): is_feasible-style early exits and the
    per-obstacle FK the reference does, (i) single-threaded (the reference is !Send) and (ii) one
    independent chain per thread on all host cores available to this process.  This leg is the only
    place bench.py touches oracle/; while it is here it also checks the GPU's flags of configurations spread
    over the last timed step (q_gpu, flags_gpu: host copies) against it, in the three figures of tests/parity.py:
    in_band / oracle_uncertified (each adjudicated by the f64 oracle, evidence attached) / outside."""
    from oracle import binding as orc
    ora = orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, precision)
    dtype = np.float32 if precision == 32 else np.float64
    cores, affinity, quota = host_threads()
    parity = None
    if q_gpu is not None:
        from tests.parity import adjudicate, compare_flags
        want, band = ora.check(q_gpu, WHAT, threads=cores)
        rep = compare_flags(flags_gpu, want, band, precision)
        if precision == 32:
            adjudicate(rep, flags_gpu, want, q_gpu, WHAT, orc.OracleScene(w.robot_keep, w.obstacle_keep, w.margin, 64), ora32=ora)
        else:
            rep["adjudicated"], rep["unresolved"] = 0, rep["oracle_uncertified"]
            rep["outside"] += rep["oracle_uncertified"]      # the f64 oracle certifies its own answers: nothing is excused
        parity = {kk: rep[kk] for kk in ("total", "mismatched", "in_band", "oracle_uncertified", "adjudicated", "unresolved", "outside")}
        parity["max_margin_minus_d64"] = rep.get("max_margin_minus_d64")
        parity["cases"] = rep.get("cases", [])[:8]
        parity["sample"] = "configurations evenly spread over the batch of the last timed step"
    probe = w.configs(32, seed=999, dtype=dtype)
    t0 = time.perf_counter()
    ora.check(probe, faithful=True, redundant_fk=True, threads=1, want_band=False)
    rate1 = len(probe) / max(time.perf_counter() - t0, 1e-9)
    n1 = max(32, int(rate1 * seconds * 0.4))
    q1 = w.configs(n1, seed=1000, dtype=dtype)
    t0 = time.perf_counter()
    ora.check(q1, faithful=True, redundant_fk=True, threads=1, want_band=False)
    single = n1 / (time.perf_counter() - t0)
    nm = max(64 * cores, int(single * cores * seconds * 0.6))
    qm = w.configs(nm, seed=1001, dtype=dtype)
    t0 = time.perf_counter()
    ora.check(qm, faithful=True, redundant_fk=True, threads=cores, want_band=False)
    multi = nm / (time.perf_counter() - t0)
    return {"value": multi, "unit": "configs/s", "cores": cores, "kind": "port",
            "affinity_cpus": affinity, "cgroup_cpu_quota": quota,
            "single_thread_value": single, "gpu_parity_spot_check": parity,
            "sample": f"{nm} configs on {cores} threads (all CPUs of sched_getaffinity, capped by the cgroup CPU quota) and "
                      f"{n1} configs on 1 thread of the same scene and input distribution; C++ restatement of "
                      f"gear's path (oracle/), not the Rust crate"}


N_INPUTS = 3   # synthetic input batches used in rotation: no step re-reads the batch of the step before it


def timed_steps(scene, qs, flags, stream, steps, warmup):
    """warmup + steps calls of gh_check_batch_dev on `stream`, step k on input batch k % len(qs); returns
    (wall seconds, mean HIP-event ms per step)."""
    for k in range(warmup):
        scene.check_batch_dev(qs[k % len(qs)], flags, WHAT, stream)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        ev[k][0].record()
        scene.check_batch_dev(qs[k % len(qs)], flags, WHAT, stream)
        ev[k][1].record()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, float(np.mean([a.elapsed_time(b) for a, b in ev]))


def side_measurement(w, precision, B, device, stream, steps, warmup, seed):
    """One extra single-GPU figure in the same run (f64, or another scene): value, ms per step, formula frac."""
    scene = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, precision, device=device.index)
    dtype = torch.float32 if precision == 32 else torch.float64
    q = device_configs(w, B, seed, dtype, device)
    flags = torch.zeros(B, dtype=torch.uint8, device=device)
    ns = min(B, 1 << 16)
    counters = scene.check_batch_counted(q[:ns], flags[:ns], WHAT)
    fpc, mix = algorithmic_flops_per_config(w, counters)
    qs = [q] + [device_configs(w, B, seed + 7919 * j, dtype, device) for j in range(1, N_INPUTS)]
    wall, kms = timed_steps(scene, qs, flags, stream, steps, warmup)
    rates = [float((flags & b).ne(0).float().mean()) for b in (1, 2, 4)]
    return {"workload": w.name, "dtype": "f32" if precision == 32 else "f64", "value": B * steps / wall, "unit": "configs/s",
            "batch": B, "ms_per_step": wall / steps * 1e3, "kernel_ms": kms,
            "algorithmic_frac": fpc * B / (kms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS[precision],
            "algorithmic_flops_per_config": fpc, "sat_per_config": mix["sat_per_config"], "exact_per_config": mix["exact_per_config"],
            "flag_rates": {"env": rates[0], "self": rates[1], "limit": rates[2]}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1 << 23,
                    help="configurations per GPU per step (8M: the per-GPU share of BASELINE configs[3]'s 64M batch on 8 GPUs)")
    ap.add_argument("--precision", type=int, default=32, choices=(32, 64))
    ap.add_argument("--obstacles", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the f64 and free-space sub-records")
    ap.add_argument("--parity-sample", type=int, default=4096, help="configurations of the last timed step checked against the oracle")
    ap.add_argument("--force-collectives", action="store_true",
                    help="take the N > 1 code path (RCCL process group, asynchronous all-gather of the flag bytes, double flag "
                         "buffers, the step on the caller's stream, counter all-reduce) whatever the world size: lets a 1-GPU box "
                         "run every line the 8-GPU run runs")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    sharded = world > 1 or args.force_collectives     # the N > 1 code path
    if sharded:
        import torch.distributed as dist
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)               # plain `python bench.py --force-collectives`: a one-rank group
        dist.init_process_group("nccl", device_id=device)
        # the scene would rotate a call's chunks over the caller's stream and three of its own; the box gives a
        # process four hardware queues and RCCL's stream needs one for the flag gather.  Measured on the one-rank
        # group with the round-3 kernels (profiles/r03/rccl_path_n1.json): two streams 4.77 G configs/s, three 4.70,
        # four 4.63, one 4.56 — so the sharded run rotates its chunks over the caller's stream and one more.
        os.environ.setdefault("GEAR_HIP_PIPE_STREAMS", "2")
    if args.gpus != world and rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    w = W.metric_workload(args.obstacles)
    scene = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, args.precision, device=local_rank)
    dtype = torch.float32 if args.precision == 32 else torch.float64
    B = args.batch
    q = device_configs(w, B, seed=1234 + rank, dtype=dtype, device=device)   # this rank's shard
    qs = [q] + [device_configs(w, B, seed=1234 + rank + 7919 * j, dtype=dtype, device=device) for j in range(1, N_INPUTS)]
    flags = torch.zeros(B, dtype=torch.uint8, device=device)
    gathered = torch.zeros(B * world, dtype=torch.uint8, device=device) if sharded else None
    # a non-default torch stream: its handle is what the C ABI launches on, so torch.cuda.Event
    # (HIP events on that same stream) brackets the kernels themselves
    tstream = torch.cuda.Stream(device=device)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0

    # instrumented run (survivor fractions for the algorithmic-flop figure) on a 64k sample
    ns = min(B, 1 << 16)
    torch.cuda.synchronize()
    counters = scene.check_batch_counted(q[:ns], flags[:ns], WHAT)
    flops_per_config, mix = algorithmic_flops_per_config(w, counters)
    # what the pipeline EXECUTES with its broadphase grid: candidates listed in the looked-up cells
    # instead of every (shape, obstacle) pair.  256K configurations.
    nc = min(B, 1 << 18)
    scene.check_batch_dev(q[:nc], flags[:nc], WHAT | D.GH_COUNT_CANDIDATES, stream)
    torch.cuda.synchronize()
    st = scene.pipeline_last_stats()
    n_self_pairs = mix["pairs_per_config"] - w.robot_keep.desc.n_shapes * w.obstacle_keep.desc.n_shapes
    cand_per_config = st["candidates"] / nc + n_self_pairs
    executed_flops = mix["F_fk"] + 11.0 * cand_per_config + 220.0 * mix["sat_per_config"] + 600.0 * mix["exact_per_config"]

    # With N > 1 the gather of step k's flag bytes (RCCL, its own stream) overlaps the check of step k+1:
    # two flag buffers, the collective asynchronous, waited for one step later and before the clock stops.
    fbuf = [flags, torch.zeros_like(flags)] if sharded else [flags]
    pending = None

    def step(k):
        nonlocal pending
        f = fbuf[k % len(fbuf)]
        scene.check_batch_dev(qs[k % N_INPUTS], f, WHAT, stream)
        if sharded:
            if pending is not None:
                pending.wait()
            pending = dist.all_gather_into_tensor(gathered, f, async_op=True)

    for k in range(args.warmup):
        step(k)
    if pending is not None:
        pending.wait()
        pending = None
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    if sharded:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        f = fbuf[k % len(fbuf)]
        ev[k][0].record()
        scene.check_batch_dev(qs[k % N_INPUTS], f, WHAT, stream)
        ev[k][1].record()
        if sharded:
            if pending is not None:
                pending.wait()
            pending = dist.all_gather_into_tensor(gathered, f, async_op=True)
    if pending is not None:
        pending.wait()
        pending = None
    torch.cuda.synchronize()
    if sharded:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if sharded:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    flags = fbuf[(args.steps - 1) % len(fbuf)]   # the buffer the last step wrote
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    # collision statistics and the instrumented run's counters; when sharded, ONE all-reduce(sum) makes them whole-job
    # figures (collision, limit, cull and SAT counters)
    stats = torch.stack([(flags & 1).ne(0).sum(), (flags & 2).ne(0).sum(), (flags & 4).ne(0).sum()]).to(torch.int64)
    stats = torch.cat([stats, torch.tensor([counters["configs"], counters["cull_pass"], counters["sat_tests"], counters["exact_tests"]],
                                           dtype=torch.int64, device=device)])
    gather_ok = None
    if sharded:
        dist.all_reduce(stats)
        # the gathered buffer holds every rank's flags of the last step: this rank's slice must be its own bytes
        gather_ok = bool((gathered[rank * B:(rank + 1) * B] == flags).all().item())
    total_configs = B * world * args.steps
    value = total_configs / elapsed

    if rank == 0:
        peak = VALU_PEAK_TFLOPS[args.precision]
        algorithmic_tflops = flops_per_config * B / (kernel_ms * 1e-3) / 1e12     # SURVEY 8d's formula: the reference's pair set
        executed_tflops = executed_flops * B / (kernel_ms * 1e-3) / 1e12          # the arithmetic the kernels really run
        bytes_per_config = w.dof * (4 if args.precision == 32 else 8) + 1
        traffic = measured_traffic(B) if args.precision == 32 else None
        out = {
            "metric": "collision-checked configs/sec (whole node), 7-DOF arm vs 256 obstacles",
            "value": value, "unit": "configs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.precision == 32 else "f64", "data": "synthetic (%d input batches in rotation)" % N_INPUTS,
            "config": {"workload": w.name, "dof": w.dof, "nodes": len(w.chain.nodes),
                       "robot_shapes": int(w.robot_keep.desc.n_shapes), "obstacles": int(w.obstacle_keep.desc.n_shapes),
                       "self_pairs": int(w.robot_keep.desc.n_pairs), "margin": w.margin,
                       "batch_per_gpu": B, "checks": "env+self+limit", "sharding": f"dp{world}, scene replicated",
                       "collectives": ("all_gather(flags) per step, overlapped with the next step; one all_reduce(sum) of the "
                                       "collision / limit / cull / SAT counters; all_reduce(max) of the elapsed time") if sharded else "none",
                       "forced_collectives": bool(args.force_collectives and world == 1),
                       "gather_self_check": gather_ok},
            "roofline": {
                "bound": "valu",
                "kernel": "k_sweep_n1 (FK + broadphase + stage 1 of the narrow phase; dominant) + k_narrow2 (exact stage) + "
                          "k_check (re-check of sweep wavefronts whose exact region ran full: normally none; the exact stage writes the flag bytes), one launch of each per chunk of at most 4M configurations (four 2M "
                          "chunks per 8M step on four streams, two of 4M on one), chunks rotating over 4 streams from 6M configurations per call",
                # `achieved` / `frac`: arithmetic the kernels EXECUTE (FK + listed broadphase candidates + SAT + exact
                # queries, priced as SURVEY 8d prices them) against the vector peak: a utilisation figure, <= 1 by
                # construction.  The formula figure, which prices the reference's whole L x O pair set although the
                # broadphase grid never executes it, is kept beside it as `algorithmic_*` (a speed-up-over-brute-force
                # figure: it would pass 1.0 with a better grid).
                "achieved": executed_tflops, "peak": peak, "unit": "TFLOP/s",
                "frac": executed_tflops / peak,
                "algorithmic_achieved": algorithmic_tflops, "algorithmic_frac": algorithmic_tflops / peak,
                "valu_issue": measured_issue(),
                "traffic": traffic,
                "traffic_unit": "bytes per step (fabric reads + writes, rocprofv3 PMC, %s: a constant read from the committed "
                                "profile, not measured in this run)" % os.path.relpath(TRAFFIC_JSON, ROOT),
                "algorithmic_bytes": bytes_per_config * B,
                "kernel_ms": kernel_ms, "algorithmic_flops_per_config": flops_per_config, "mix": mix,
                "executed": {
                    "broadphase_candidates_per_config": cand_per_config, "flops_per_config": executed_flops,
                    "TFLOPs": executed_tflops, "frac": executed_tflops / peak},
                "hbm_algorithmic_GBps": bytes_per_config * B / (kernel_ms * 1e-3) / 1e9,
                "hbm_peak_GBps": HBM_PEAK_GBS,
                "hbm_traffic_GBps": None if traffic is None else traffic / (kernel_ms * 1e-3) / 1e9,
                "hbm_traffic_frac_of_peak": None if traffic is None else traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "sustained_fma_TFLOPs_measured": {"3_waves_per_simd": 111.9, "8_waves_per_simd": 129.9,
                                                  "source": "tools/valu_rate.hip, profiles/r01/valu_rate.txt (constants from that file)"},
                "note": "VALU-bound path (SURVEY 8d): the f32 vector peak equals the f32 MFMA peak; MFMA unused by "
                        "design.  kernel_ms = HIP events on the launch stream around one step (all kernels of the step)"},
            "counters_all_reduced": {"instrumented_configs": int(stats[3].item()), "cull_pass": int(stats[4].item()),
                                     "sat_tests": int(stats[5].item()), "exact_tests": int(stats[6].item())},
            "flag_rates": {"env": stats[0].item() / (B * world), "self": stats[1].item() / (B * world),
                           "limit": stats[2].item() / (B * world)},
        }
        if world == 1 and not args.no_extras:
            # the same scene in the other precision, and a free-space-dominant variant of it (the metric scene
            # is dense: 59 % of random configurations collide, which lets the pipeline drop most of its
            # undecided items): same code path, same batch, shorter runs
            other = 64 if args.precision == 32 else 32
            side_steps = max(2, args.steps // 2)
            out["f64" if other == 64 else "f32"] = side_measurement(w, other, B, device, stream, side_steps, 1, 1234)
            out["free_space"] = side_measurement(W.metric_workload_sparse(args.obstacles), args.precision, B, device, stream,
                                                 side_steps, 1, 1234)
        if world == 1 and not args.no_cpu_baseline:
            k = 0 if args.no_parity else min(args.parity_sample, B)
            q_last = qs[(args.steps - 1) % N_INPUTS]   # the input batch of the step that wrote `flags`
            pick = torch.linspace(0, B - 1, max(k, 1), device=device, dtype=torch.float64).round().to(torch.int64) if k else None   # spread over every chunk
            out["cpu_baseline"] = cpu_baseline(w, args.precision, q_last[pick].cpu().numpy() if k else None,
                                               flags[pick].cpu().numpy() if k else None)
            spot = out["cpu_baseline"]["gpu_parity_spot_check"]
            if spot is not None and spot["outside"] != 0:   # the line is still printed: the record says what was measured
                print(f"bench: WARNING: the timed step's flags differ from the oracle's outside the band: {spot}", file=sys.stderr)
        print(json.dumps(out))
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
