import sys, torch
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
from gear_amd import engine, workloads as W
import bench
what = int(sys.argv[1], 0)
w = W.metric_workload()
sc = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32)
dev = torch.device('cuda', 0)
B = 1 << 19
q = bench.device_configs(w, B, 1234, torch.float32, dev)
flags = torch.zeros(B, dtype=torch.uint8, device=dev)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
for _ in range(3): sc.check_batch_dev(q, flags, what, ts.cuda_stream)
torch.cuda.synchronize()
