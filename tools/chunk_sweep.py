import sys, torch
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
from gear_amd import engine, workloads as W
import bench
w = W.metric_workload()
sc = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32)
dev = torch.device('cuda', 0)
B = 1 << 22
q = bench.device_configs(w, B, 1234, torch.float32, dev)
f = torch.zeros(B, dtype=torch.uint8, device=dev)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
for _ in range(2): sc.check_batch_dev(q, f, 3, ts.cuda_stream)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): sc.check_batch_dev(q, f, 3, ts.cuda_stream)
b.record(); torch.cuda.synchronize()
import os
print(os.environ.get("GEAR_HIP_PIPE_CHUNK"), f"{a.elapsed_time(b)/5:.3f} ms per 4M -> {B/(a.elapsed_time(b)/5*1e-3)/1e9:.3f} G/s")
