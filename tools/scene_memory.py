"""Device memory a scene's tables take (hipMemGetInfo before / after gh_scene_create) and the time its creation takes."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gear_amd import engine, workloads as W
L = engine.lib()
L.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
def free():
    L.hipDeviceSynchronize()
    f, t = C.c_size_t(), C.c_size_t()
    assert L.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value
warm = engine.Scene(*(lambda w: (w.robot_keep, w.obstacle_keep, w.margin))(W.metric_workload(8)), 32)
for name, mk in (("metric (256)", W.metric_workload), ("ur10 (256 box/cyl)", lambda: W.ur10_workload(256)), ("sawyer (4096)", lambda: W.sawyer_workload(4096)),
                 ("pr2 (128)", lambda: W.pr2_workload(128))):
    w = mk()
    for prec in (32, 64):
        a = free(); t0 = time.time()
        sc = engine.Scene(w.robot_keep, w.obstacle_keep, w.margin, prec)
        dt = time.time() - t0
        print(f"{name:22s} f{prec}: {(a - free()) / 2**20:7.1f} MiB of tables, streams and staging; created in {dt * 1e3:6.1f} ms")
        sc.close()
