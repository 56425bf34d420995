"""Exact-stage items per configuration by obstacle kind on the metric robot (profiling helper)."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gear_amd import engine, workloads as W
import bench
dev = torch.device('cuda', 0)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
B = 1 << 18
for name, kinds, probs, pairs in (("box+ball 60/40, self", ("box", "ball"), (0.6, 0.4), W.ARM7_SELF_PAIRS),
                                  ("box+ball 60/40, no self", ("box", "ball"), (0.6, 0.4), []),
                                  ("box only, no self", ("box",), (1.0,), []),
                                  ("ball only, no self", ("ball",), (1.0,), [])):
    w = W.Workload(name, W.arm7_robot(), W.obstacle_field(256, 0, kinds, probs), 0.01, pairs)
    sc = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32)
    q = bench.device_configs(w, B, 7, torch.float32, dev)
    f = torch.zeros(B, dtype=torch.uint8, device=dev)
    sc.check_batch_dev(q, f, 3 | 0x2000, ts.cuda_stream)
    torch.cuda.synchronize()
    st = sc.pipeline_last_stats()
    print(f"{name:28s} exact items/config {st['exact_items'] / B:.4f}  env rate {float(((f & 1) != 0).float().mean()):.3f}", flush=True)
