"""BASELINE configs[2] at full size through the host-buffer entry points: 65536 straight edges of the 6-DOF arm against
the 256 box/cylinder field (about 4M samples), (a) every sample over PCIe (gh_check_edges, pageable numpy arrays),
(b) the end points only, the samples generated on the device (gh_check_edges_from_endpoints).  Usage (GPU box):
python tools/edges_endpoints.py [precision]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from gear_amd import descriptors as D, engine, workloads as W

prec = int(sys.argv[1]) if len(sys.argv) > 1 else 32
w = W.ur10_workload(256)
hip = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
E = 65536
a, b = w.configs(E, seed=61, dtype=hip.dtype), w.configs(E, seed=62, dtype=hip.dtype)
q, off = W.edge_samples_lockstep(a, b, 0.1, D.GH_EDGE_CONNECT)
n = int(off[-1])
BAD = D.GH_FLAG_ENV | D.GH_FLAG_LIMIT


def timed(f, reps=6):
    f()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        t.append(time.perf_counter() - t0)
    return min(t), float(np.median(t))


r1 = hip.check_edges(q, off, D.GH_CHECK_ENV, BAD)[1]
r2 = hip.check_edges_from_endpoints(a, b, 0.1, 4096, D.GH_EDGE_CONNECT, D.GH_CHECK_ENV, BAD)[0]
assert (r1 == r2).all()
ta = timed(lambda: hip.check_edges(q, off, D.GH_CHECK_ENV, BAD))
tb = timed(lambda: hip.check_edges_from_endpoints(a, b, 0.1, 4096, D.GH_EDGE_CONNECT, D.GH_CHECK_ENV, BAD))
row = hip.dof * q.itemsize
print(f"f{prec}: {E} edges, {n} samples")
print(f"  gh_check_edges (samples over PCIe, {n * row / 1e6:.1f} MB in, {n / 1e6:.1f} MB flags out): best {ta[0] * 1e3:.2f} ms, median {ta[1] * 1e3:.2f} ms"
      f" = {n / ta[0] / 1e9:.3f} G samples/s")
print(f"  gh_check_edges_from_endpoints ({2 * E * row / 1e6:.1f} MB in, {8 * E / 1e6:.1f} MB out): best {tb[0] * 1e3:.2f} ms, median {tb[1] * 1e3:.2f} ms"
      f" = {n / tb[0] / 1e9:.3f} G samples/s   ({ta[0] / tb[0]:.2f}x)")
