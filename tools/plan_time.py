"""Wall time of JointPathPlanner::plan (C++ host layer over the C ABI) in the 96-obstacle field of
tests/test_gpu_host_cpp.py::test_plan_in_a_cluttered_field, speculative and sample-by-sample
(measuring helper; GEAR_HIP_SMALL_FUSED=1 sends the small calls through the fused kernel as before v8)."""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gear_amd import host_binding as H, urdf, workloads as W
robot = W.arm7_robot()
p = os.path.join(tempfile.mkdtemp(), "arm7.urdf")
open(p, "w").write(urdf.robot_to_xml(robot))
field = W.obstacle_field(96, 5, ("box", "ball"), (0.6, 0.4))
obs = H.HostCompound.from_specs(field.shapes)
pl = H.HostPlanner(p, 0.01, 32, step_length=0.1, max_try=5000, num_smoothing=100, seed=11)
u = pl.using_joints()
chain = urdf.Chain.from_urdf(robot)
lo, hi = W.joint_limits(chain, u)
rng = np.random.default_rng(3)
cand = (lo + (hi - lo) * rng.random((600, len(u)))).astype(np.float32).astype(np.float64)
free = cand[pl.is_feasible_batch(u, cand, obs)]
start, goal = free[0], free[1]
for spec in (True, False, True, False):
    pl.set_speculation(spec, 16, seed=99)
    t0 = time.perf_counter()
    path = pl.plan(u, start, goal, obs)
    dt = time.perf_counter() - t0
    st = dict(pl.last_stats)
    print(f"speculative={spec} plan {dt * 1e3:8.1f} ms  way points {len(path)}  gpu calls {st['gpu_calls']}  "
          f"{dt * 1e6 / max(st['gpu_calls'], 1):6.1f} us per call, of which {st['eval_ns'] / 1e3 / max(st['gpu_calls'], 1):6.1f} us inside the evaluator  {st}", flush=True)
