"""Latency of small batches (RRT lines), device buffers, back-to-back calls on one stream.
usage: small_batch.py [path bits: 0 default, 0x20 k_small, 0x1000 fused, 0x2000 pipeline, 0xffff limits only] [sizes] [precision] [metric|ur10|sawyer|pr2]"""
import sys, os, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gear_amd import engine, workloads as W
import bench
wl = sys.argv[4] if len(sys.argv) > 4 else 'metric'
w = {'metric': W.metric_workload, 'ur10': W.ur10_workload, 'sawyer': W.sawyer_workload, 'pr2': W.pr2_workload}[wl]()
prec = int(sys.argv[3]) if len(sys.argv) > 3 else 32
sc = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
dt = torch.float32 if prec == 32 else torch.float64
dev = torch.device('cuda', 0)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
path = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
bits = 0 if path == 0xffff else 3 | path   # 0xffff: limits only (what = 0): the floor of one launch
for B in [int(x, 0) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 and sys.argv[2] else (1, 64, 256, 1024, 4096, 8192, 16000, 65536):
    q = bench.device_configs(w, B, 5, dt, dev)
    f = torch.zeros(B, dtype=torch.uint8, device=dev)
    reps = 50 if B <= (1 << 18) else 10
    for _ in range(5): sc.check_batch_dev(q, f, bits, ts.cuda_stream)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): sc.check_batch_dev(q, f, bits, ts.cuda_stream)
    b.record(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        sc.check_batch_dev(q, f, bits, ts.cuda_stream); ts.synchronize()
    t1 = time.perf_counter()
    print(f"{wl} path={path:#x} f{prec} B={B:6d} {a.elapsed_time(b)/reps*1000:8.1f} us per call back to back, {(t1-t0)/reps*1e6:8.1f} us launch+sync", flush=True)
