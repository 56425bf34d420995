"""Where a k_small call spends its time: the -DGEAR_HIP_ABLATE build (gear_amd/build.py build_ablate) with
stages switched off by `what` bits (WRONG flags on purpose; the product build ignores these bits).
0x100 queued items dropped (no narrow phase) | 0x200 nothing queued (culls still run) | 0x400 no broadphase |
0x800 no FK walk (P1).  usage: GEAR_HIP_LIB=<ablate build> python tools/ablate_small.py [sizes]"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gear_amd import engine, workloads as W
import bench
w = W.metric_workload()
sc = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32)
dev = torch.device('cuda', 0)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
sizes = [int(x, 0) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else (64, 1024)
for B in sizes:
    q = bench.device_configs(w, B, 5, torch.float32, dev)
    f = torch.zeros(B, dtype=torch.uint8, device=dev)
    for name, bits in (("full", 0), ("no narrow phase", 0x100), ("culls only", 0x200), ("no broadphase", 0x400),
                       ("no broadphase, no walk", 0xc00), ("env only", -2), ("self only", -1), ("limits only (fused)", None)):
        what = 0 if bits is None else (1 | 0x20) if bits == -2 else (2 | 0x20) if bits == -1 else (3 | 0x20 | bits)
        for _ in range(5): sc.check_batch_dev(q, f, what, ts.cuda_stream)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100): sc.check_batch_dev(q, f, what, ts.cuda_stream)
        b.record(); torch.cuda.synchronize()
        print(f"B={B:5d} {name:28s} {a.elapsed_time(b) * 10:7.1f} us per call", flush=True)
    sc.check_batch_dev(q, f, 3 | 0x20 | 0x10000000, ts.cuda_stream)   # phase stamps of two workgroups
    torch.cuda.synchronize()
