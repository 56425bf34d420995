"""Latency of small batches (RRT lines) through the fused kernel: old vs new library."""
import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from gear_amd import engine, workloads as W
import bench
w = W.metric_workload()
sc = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32)
dev = torch.device('cuda', 0)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
path = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0   # 0x1000 fused, 0x2000 pipeline, 0x6000 pipeline without the grid
for B in [int(x, 0) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (64, 256, 1024, 4096, 16000, 65536):
    q = bench.device_configs(w, B, 5, torch.float32, dev)
    f = torch.zeros(B, dtype=torch.uint8, device=dev)
    reps = 50 if B <= (1 << 18) else 10
    for _ in range(5): sc.check_batch_dev(q, f, 3 | path, ts.cuda_stream)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): sc.check_batch_dev(q, f, 3 | path, ts.cuda_stream)
    b.record(); torch.cuda.synchronize()
    print(f"B={B:6d} {a.elapsed_time(b)/reps*1000:8.1f} us per call", flush=True)
