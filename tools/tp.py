"""Throughput of gh_check_batch_dev on the metric scene for a given `what` and batch (measuring helper).
Usage: python tools/tp.py [what=3] [batch=4194304] [precision=32] [workload=metric|ur10|sawyer|pr2]"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gear_amd import engine, workloads as W
import bench
what = int(sys.argv[1], 0) if len(sys.argv) > 1 else 3
B = int(sys.argv[2], 0) if len(sys.argv) > 2 else 1 << 22
prec = int(sys.argv[3]) if len(sys.argv) > 3 else 32
wl = sys.argv[4] if len(sys.argv) > 4 else "metric"
w = {"metric": W.metric_workload, "ur10": W.ur10_workload, "sawyer": W.sawyer_workload, "pr2": W.pr2_workload}[wl]()
sc = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
dev = torch.device('cuda', 0)
dt = torch.float32 if prec == 32 else torch.float64
q = bench.device_configs(w, B, 1234, dt, dev)
f = torch.zeros(B, dtype=torch.uint8, device=dev)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
for _ in range(3): sc.check_batch_dev(q, f, what, ts.cuda_stream)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 10
a.record()
for _ in range(reps): sc.check_batch_dev(q, f, what, ts.cuda_stream)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / reps
print(f"{wl} what={what:#x} B={B} f{prec}: {ms:.3f} ms  {B / ms / 1e6:.3f} G/s  rates env {float(((f & 1) != 0).float().mean()):.3f} self {float(((f & 2) != 0).float().mean()):.3f}", flush=True)
