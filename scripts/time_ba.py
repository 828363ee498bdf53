"""C3 local-BA timing: device-resident LM vs the host Schur loop (both around HIP kernels)."""
import copy, importlib, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import ba_scenes
S = importlib.import_module("opencv-simpleslam_amd.ba_solver")
bau = importlib.import_module("opencv-simpleslam_amd.slam.core.ba_utils")
wmap, kfs, K = ba_scenes.scaled_scene()
prob, _, _ = bau.snapshot_problem(wmap, K, kfs, list(range(5, 15)), list(range(0, 5)), 5000)
print(f"C3 scene: {len(prob.obs_pose)} observations, {len(prob.X)} points, {int((~prob.pose_const).sum())} opt + {int(prob.pose_const.sum())} fixed poses")
for name, fn in (("device", S.solve_device), ("host", S.solve_host)):
    fn(copy.deepcopy(prob), 12, 2.0)                      # warm-up
    ts = []
    for _ in range(5 if name == "device" else 2):
        p = copy.deepcopy(prob)
        t0 = time.perf_counter(); s = fn(p, 12, 2.0); ts.append(time.perf_counter() - t0)
    print(f"{name:6s}: {np.median(ts)*1e3:9.2f} ms per solve  iters={s.iterations} steps={s.successful_steps} "
          f"cost {s.initial_cost:.1f} -> {s.final_cost:.1f} ({s.termination})")

# global-BA shape (ba_utils.py:170-218): every keyframe but the first free
wmap, kfs, K = ba_scenes.scaled_scene(n_kf=30, n_points=5000)
prob, _, _ = bau.snapshot_problem(wmap, K, kfs, list(range(30)), [0], 30000)
print(f"global shape: {len(prob.obs_pose)} observations, {len(prob.X)} points, {int((~prob.pose_const).sum())} opt + {int(prob.pose_const.sum())} fixed poses")
for name, fn in (("device", S.solve_device), ("host", S.solve_host)):
    fn(copy.deepcopy(prob), 12, 2.0)
    ts = []
    for _ in range(3 if name == "device" else 1):
        p = copy.deepcopy(prob)
        t0 = time.perf_counter(); s = fn(p, 12, 2.0); ts.append(time.perf_counter() - t0)
    print(f"{name:6s}: {np.median(ts)*1e3:9.2f} ms per solve  iters={s.iterations} steps={s.successful_steps} "
          f"cost {s.initial_cost:.1f} -> {s.final_cost:.1f} ({s.termination})")
