"""PCIe-inclusive rate of the host-buffer entry point gh_check_batch on the metric scene: the caller's arrays
pageable (numpy) and page-locked (hipHostMalloc), against the device-resident rate bench.py reports.
usage: python tools/host_buffers.py [sizes]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gear_amd import descriptors as D, engine, workloads as W

hip = C.CDLL("libamdhip64.so")
w = W.metric_workload()
sc = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32)
sizes = [int(x, 0) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [1 << 16, 1 << 20, 1 << 22, 1 << 23]
for B in sizes:
    q0 = w.configs(B, seed=3)
    for pinned in (False, True):
        if pinned:   # the caller's arrays in page-locked memory (hipHostMalloc)
            pq, pf = C.c_void_p(), C.c_void_p()
            assert hip.hipHostMalloc(C.byref(pq), C.c_size_t(q0.nbytes), 0) == 0
            assert hip.hipHostMalloc(C.byref(pf), C.c_size_t(B), 0) == 0
            q = np.ctypeslib.as_array(C.cast(pq, C.POINTER(C.c_float)), shape=q0.shape)
            q[:] = q0
            flags = np.ctypeslib.as_array(C.cast(pf, C.POINTER(C.c_uint8)), shape=(B,))
        else:
            q, flags = q0, np.zeros(B, dtype=np.uint8)

        def call():
            rc = engine.lib().gh_check_batch(sc.h, q.ctypes.data_as(C.c_void_p), B, D.GH_CHECK_ENV | D.GH_CHECK_SELF,
                                             flags.ctypes.data_as(C.POINTER(C.c_uint8)))
            assert rc == 0, rc

        for _ in range(2):
            call()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        dt = (time.perf_counter() - t0) / reps
        print(f"host buffers {'page-locked' if pinned else 'pageable   '} B={B:8d}: {dt * 1e3:8.3f} ms  {B / dt / 1e9:6.3f} G configs/s"
              f"  ({B * 29 / dt / 1e9:5.1f} GB/s over the bus)", flush=True)
        if pinned:
            del q, flags
            hip.hipHostFree(pq)
            hip.hipHostFree(pf)

# a device group (gh_scene_create_multi) on this one-GPU box: the device listed twice and four times, i.e. the group's host
# threads, per-share streams, staging and workspace sets, all on one device (no second GPU to add its own PCIe link and CUs)
for devs in ([0, 0], [0, 0, 0, 0]):
    grp = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, 32, devices=devs)
    B = 1 << 23
    q0 = w.configs(B, seed=3)
    for pinned in (False, True):
        if pinned:
            pq, pf = C.c_void_p(), C.c_void_p()
            assert hip.hipHostMalloc(C.byref(pq), C.c_size_t(q0.nbytes), 0) == 0
            assert hip.hipHostMalloc(C.byref(pf), C.c_size_t(B), 0) == 0
            q = np.ctypeslib.as_array(C.cast(pq, C.POINTER(C.c_float)), shape=q0.shape)
            q[:] = q0
            flags = np.ctypeslib.as_array(C.cast(pf, C.POINTER(C.c_uint8)), shape=(B,))
        else:
            q, flags = q0, np.zeros(B, dtype=np.uint8)

        def gcall():
            rc = engine.lib().gh_check_batch(grp.h, q.ctypes.data_as(C.c_void_p), B, D.GH_CHECK_ENV | D.GH_CHECK_SELF,
                                             flags.ctypes.data_as(C.POINTER(C.c_uint8)))
            assert rc == 0, rc

        for _ in range(2):
            gcall()
        t0 = time.perf_counter()
        for _ in range(5):
            gcall()
        dt = (time.perf_counter() - t0) / 5
        print(f"device group {devs} host buffers {'page-locked' if pinned else 'pageable   '} B={B:8d}: {dt * 1e3:8.3f} ms  {B / dt / 1e9:6.3f} G configs/s"
              f"  ({B * 29 / dt / 1e9:5.1f} GB/s over the bus)", flush=True)
        if pinned:
            del q, flags
            hip.hipHostFree(pq)
            hip.hipHostFree(pf)
    del grp

# gh_check_edges, mid-size: segments of 16..64 samples
for n_edges in (256, 2048, 8192):
    qe, off = w.edges(n_edges, seed=41, dtype=np.float32, min_len=16, max_len=64)
    for _ in range(3):
        sc.check_edges(qe, off, D.GH_CHECK_ENV, D.GH_FLAG_ENV | D.GH_FLAG_LIMIT)
    t0 = time.perf_counter()
    for _ in range(10):
        sc.check_edges(qe, off, D.GH_CHECK_ENV, D.GH_FLAG_ENV | D.GH_FLAG_LIMIT)
    dt = (time.perf_counter() - t0) / 10
    print(f"gh_check_edges pageable {n_edges:6d} edges, {len(qe):7d} samples: {dt * 1e3:8.3f} ms", flush=True)

# configs[2]: 64k straight segments of 8..64 samples against the 256 box / cylinder field, one gh_check_edges call
w2 = W.ur10_workload(256)
sc2 = engine.Scene(w2.robot_keep, w2.obstacle_keep, w2.margin, 32)
qe, off = w2.edges_lockstep(65536, seed=9, min_len=8, max_len=64)
for _ in range(2):
    sc2.check_edges(qe, off, D.GH_CHECK_ENV, D.GH_FLAG_ENV | D.GH_FLAG_LIMIT)
t0 = time.perf_counter()
for _ in range(5):
    sc2.check_edges(qe, off, D.GH_CHECK_ENV, D.GH_FLAG_ENV | D.GH_FLAG_LIMIT)
dt = (time.perf_counter() - t0) / 5
print(f"gh_check_edges pageable  65536 edges, {len(qe):7d} samples (configs[2]): {dt * 1e3:8.3f} ms  {len(qe) / dt / 1e9:6.3f} G samples/s", flush=True)
