"""Pipeline vs fused: flags must be identical; per-path timings (profiling helper, not a test)."""
import sys, torch, numpy as np
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
from gear_amd import engine, workloads as W
import bench
FUSED, PIPE, NOGRID = 0x1000, 0x2000, 0x4000
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 32
w = W.metric_workload()
sc = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
dev = torch.device('cuda', 0)
B = 1 << 20
dt = torch.float32 if prec == 32 else torch.float64
q = bench.device_configs(w, B, 1234, dt, dev)
f1 = torch.zeros(B, dtype=torch.uint8, device=dev)
f2 = torch.full((B,), 99, dtype=torch.uint8, device=dev)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
for what in (3, 1, 2):
    sc.check_batch_dev(q, f1, what | FUSED, ts.cuda_stream)
    sc.check_batch_dev(q, f2, what | PIPE, ts.cuda_stream)
    torch.cuda.synchronize()
    f3 = torch.full((B,), 99, dtype=torch.uint8, device=dev)
    sc.check_batch_dev(q, f3, what | PIPE | NOGRID, ts.cuda_stream)
    torch.cuda.synchronize()
    print("what", what, "mismatches", int((f1 != f2).sum()), int((f1 != f3).sum()), "rates", [float(((f2 & b) != 0).float().mean()) for b in (1, 2, 4)])
def run(what, label, reps=5):
    for _ in range(2): sc.check_batch_dev(q, f2, what, ts.cuda_stream)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): sc.check_batch_dev(q, f2, what, ts.cuda_stream)
    b.record(); torch.cuda.synchronize()
    print(f"{label:44s} {a.elapsed_time(b)/reps:8.3f} ms")
run(3 | FUSED, "fused env+self")
run(3 | PIPE, "pipe env+self")
run(3 | PIPE | NOGRID, "pipe env+self, brute-force sweep")
run(1 | PIPE | NOGRID | 0x300, "pipe env brute, no narrow")
run(1 | PIPE, "pipe env")
run(2 | PIPE, "pipe self")
run(3 | PIPE | 0x100, "pipe env+self, no exact")
run(3 | PIPE | 0x300, "pipe env+self, no narrow")
run(1 | PIPE | 0x300, "pipe env, no narrow")
run(1 | PIPE | 0x700, "pipe env, sweep only (no push)")
run(1 | PIPE | 0xb00, "pipe env, push only (no sweep)")
run(1 | PIPE | 0xf00, "pipe env, neither")
sc.check_batch_dev(q, f2, 3 | PIPE | 0x8000, ts.cuda_stream)
st = sc.pipeline_last_stats()
print("grid candidates per config", st["candidates"] / B, "exact items per config", st["exact_items"] / B, st)
