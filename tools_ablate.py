import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from gear_amd import engine, workloads as W, descriptors as D
w = W.metric_workload()
sc = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32)
import bench
dev = torch.device('cuda', 0)
B = 1 << 20
q = bench.device_configs(w, B, 1234, torch.float32, dev)
flags = torch.zeros(B, dtype=torch.uint8, device=dev)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
def run(what, label, reps=5):
    for _ in range(2): sc.check_batch_dev(q, flags, what, ts.cuda_stream)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): sc.check_batch_dev(q, flags, what, ts.cuda_stream)
    b.record(); torch.cuda.synchronize()
    print(f"{label:40s} {a.elapsed_time(b)/reps:8.3f} ms")
run(3, "env+self full")
run(1, "env full")
run(2, "self full")
run(0, "fk+limits only")
run(3 | 0x100, "env+self, no exact stage")
run(3 | 0x200, "env+self, no drain at all")
run(1 | 0x200, "env, no drain (sweep+push)")
run(2 | 0x200, "self, no drain")
