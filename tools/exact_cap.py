"""How many exact-stage entries a sweep wavefront needs: for each scene and each capacity, the share of sweep wavefronts
whose region ran full (their 64 configurations are re-checked by the fused path) and the throughput.  Usage (GPU box):
python tools/exact_cap.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gear_amd import descriptors as D, engine, workloads as W
import bench
dev = torch.device('cuda', 0)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
B = 1 << 20
for name, mk in (("metric", W.metric_workload), ("metric_sparse", W.metric_workload_sparse), ("ur10", W.ur10_workload), ("sawyer", W.sawyer_workload), ("pr2", W.pr2_workload)):
    w = mk()
    for prec in (32, 64):
        dt = torch.float32 if prec == 32 else torch.float64
        q = bench.device_configs(w, B, 1234, dt, dev)
        f = torch.zeros(B, dtype=torch.uint8, device=dev)
        for cap in (0, 16, 32, 64, 128, 256):
            if cap: os.environ["GEAR_HIP_EXACT_CAP"] = str(cap)
            else: os.environ.pop("GEAR_HIP_EXACT_CAP", None)
            sc = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
            sc.check_batch_dev(q, f, 3 | D.GH_PATH_PIPELINE | D.GH_COUNT_CANDIDATES, ts.cuda_stream)
            torch.cuda.synchronize()
            st = sc.pipeline_last_stats()
            for _ in range(2): sc.check_batch_dev(q, f, 3 | D.GH_PATH_PIPELINE, ts.cuda_stream)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5): sc.check_batch_dev(q, f, 3 | D.GH_PATH_PIPELINE, ts.cuda_stream)
            b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 5
            print(f"{name:14s} f{prec} cap {st['exact_cap']:4d}{' (default)' if not cap else '':10s} exact items/config {st['exact_items'] / B:6.3f}  "
                  f"full regions {st['redo_waves']:6d} of {B // 64} wavefronts ({100.0 * st['redo_waves'] / (B // 64):6.3f} %)  {B / ms / 1e6:6.3f} G configs/s", flush=True)
            del sc
