"""GJK iteration histogram of the exact stage on the metric scene (profiling helper; run with GEAR_HIP_N2_HIST=1)."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gear_amd import engine, workloads as W
import bench
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 32
w = W.metric_workload()
sc = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
dev = torch.device('cuda', 0)
B = 1 << 18
q = bench.device_configs(w, B, 1234, torch.float32 if prec == 32 else torch.float64, dev)
f = torch.zeros(B, dtype=torch.uint8, device=dev)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
sc.check_batch_dev(q, f, 3 | 0x2000, ts.cuda_stream)
torch.cuda.synchronize()
