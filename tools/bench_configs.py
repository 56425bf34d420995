"""Throughput of the BASELINE configs other than the metric scene (profiling helper)."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gear_amd import engine, workloads as W
import bench
dev = torch.device('cuda', 0)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
for name, w, prec, B in (("sample.urdf vs obstacles.urdf (configs[1], 1M)", None, 32, 1 << 20),
                         ("arm6 vs 256 box/cyl", W.ur10_workload(256), 32, 1 << 20),
                         ("arm7 vs 4096 mixed", W.sawyer_workload(4096), 32, 1 << 18),
                         ("dual-arm mesh vs 128", W.pr2_workload(128), 32, 1 << 19),
                         ("metric scene f64", W.metric_workload(), 64, 1 << 20)):
    if w is None:
        import json
        from gear_amd import urdf
        g = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')
        w = W.sample_workload(urdf.robot_from_dict(json.load(open(g + '/sample_robot.json'))),
                              urdf.robot_from_dict(json.load(open(g + '/obstacles.json'))), 0.01,
                              self_pairs=[("root", "l_wrist_yaw"), ("root", "l_wrist_pitch"), ("l_shoulder_yaw", "l_wrist_pitch")])
    sc = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
    dt = torch.float32 if prec == 32 else torch.float64
    q = bench.device_configs(w, B, 7, dt, dev)
    f = torch.zeros(B, dtype=torch.uint8, device=dev)
    for _ in range(2): sc.check_batch_dev(q, f, 3, ts.cuda_stream)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): sc.check_batch_dev(q, f, 3, ts.cuda_stream)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 3
    print(f"{name:48s} f{prec} B={B:8d} {ms:8.3f} ms  {B / ms / 1e6:8.3f} G configs/s   rates env/self/limit",
          [round(float(((f & k) != 0).float().mean()), 3) for k in (1, 2, 4)], flush=True)
