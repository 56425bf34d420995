"""Per-stage timings of the grid pipeline on the metric scene (profiling helper, not a test)."""
import sys, torch
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
from gear_amd import engine, workloads as W
import bench
PIPE = 0x2000
w = W.metric_workload()
sc = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32)
dev = torch.device('cuda', 0)
B = 1 << 20
q = bench.device_configs(w, B, 1234, torch.float32, dev)
f2 = torch.zeros(B, dtype=torch.uint8, device=dev)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
def run(what, label, reps=5):
    for _ in range(2): sc.check_batch_dev(q, f2, what, ts.cuda_stream)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): sc.check_batch_dev(q, f2, what, ts.cuda_stream)
    b.record(); torch.cuda.synchronize()
    print(f"{label:52s} {a.elapsed_time(b)/reps:8.3f} ms", flush=True)
run(3 | PIPE, "env+self")
run(1 | PIPE, "env")
run(2 | PIPE, "self")
run(3 | PIPE | 0x100, "env+self, no exact stage")
run(3 | PIPE | 0x300, "env+self, no narrow phase (sweep + finalize)")
run(1 | PIPE | 0x300, "env, no narrow phase")
run(1 | PIPE | 0xb00, "env, no narrow, no cell lookup (FK + finalize)")
run(3 | PIPE | 0xb00, "env+self, no narrow, no cell lookup")
run(3 | PIPE | 0x10100, "env+self, no exact, 4 distinct obstacle records")
run(3 | PIPE | 0x20100, "env+self, no exact, wave-uniform obstacle record")
run(3 | PIPE | 0x60100, "env+self, no exact, wave-uniform obstacle + robot")
run(3 | PIPE | 0x300, "env+self, no narrow (again)")
run(3 | PIPE | 0x80300, "env+self, no narrow, no pose store")
run(3 | PIPE, "env+self (again)")
run(3 | PIPE | 0x400000, "env+self, exact stage without divergence (one item per wave)")
