"""Generate tests/golden/ba_assembly.npz by running the REFERENCE's own
`slam.core.ba_utils.local_bundle_adjustment` against recording stubs of its
absent third-party dependencies (cv2, pyceres, pycolmap).

Run in the build container only (needs /root/reference on disk):
    python tests/golden/make_ba_assembly_golden.py

What this pins (everything in `_core_ba` except the third-party arithmetic):
window selection, parameter-block order, which blocks are constant / carry the
quaternion manifold, the Huber delta, which (point, keyframe, uv) residual
blocks are added and in which order, `max_iters` forwarding, the < 10
residuals early-out, and the write-back of optimised poses into
`kfs[k].pose` / `world_map.poses[k]` and of points in place.

The .npz holds scene inputs and the recorded trace (data only).
"""
import sys
import types
import numpy as np

REC = {}


def _install_stubs():
    cv2 = types.ModuleType("cv2")

    class KeyPoint:
        def __init__(self, x, y, size=1):
            self.pt = (float(x), float(y))
            self.size = size
    cv2.KeyPoint = KeyPoint
    sys.modules["cv2"] = cv2

    pyceres = types.ModuleType("pyceres")

    class Problem:
        def __init__(self):
            self.blocks = []          # list of arrays in add order
            self.sizes = []
            self.constant = []        # ids
            self.manifold = []        # ids
            self.residuals = []       # (uv, [param arrays])
            REC["problem"] = self

        def add_parameter_block(self, arr, size):
            if not any(arr is b for b in self.blocks):
                self.blocks.append(arr)
                self.sizes.append(size)

        def set_manifold(self, arr, m):
            self.manifold.append(id(arr))

        def set_parameter_block_constant(self, arr):
            self.constant.append(id(arr))

        def add_residual_block(self, cost, loss, params):
            self.residuals.append((cost.uv.copy(), loss.delta, list(params)))

    class HuberLoss:
        def __init__(self, delta):
            self.delta = float(delta)

    class EigenQuaternionManifold:
        pass

    class SolverOptions:
        max_num_iterations = 50
        minimizer_progress_to_stdout = True

    class SolverSummary:
        final_cost = 1.25
        iterations_used = 3

    def solve(opts, problem, summary):
        REC["max_iters"] = opts.max_num_iterations
        # deterministic fake "optimisation": nudge every non-constant block in place
        for i, b in enumerate(problem.blocks):
            if id(b) in problem.constant:
                continue
            b += 1e-3 * (1 + (i % 5)) * np.arange(1, b.size + 1)

    pyceres.Problem = Problem
    pyceres.HuberLoss = HuberLoss
    pyceres.EigenQuaternionManifold = EigenQuaternionManifold
    pyceres.SolverOptions = SolverOptions
    pyceres.SolverSummary = SolverSummary
    pyceres.solve = solve
    sys.modules["pyceres"] = pyceres

    pycolmap = types.ModuleType("pycolmap")
    cf = types.ModuleType("pycolmap.cost_functions")

    class _Cost:
        def __init__(self, model, uv):
            assert model == "PINHOLE"
            self.uv = np.asarray(uv, np.float64)
    cf.ReprojErrorCost = _Cost
    pycolmap.cost_functions = cf

    class CameraModelId:
        PINHOLE = "PINHOLE"
    pycolmap.CameraModelId = CameraModelId
    sys.modules["pycolmap"] = pycolmap
    sys.modules["pycolmap.cost_functions"] = cf
    return cv2


def _scene(cv2, n_kf=9, n_pts=40, seed=3):
    from slam.core.landmark_utils import Map
    rng = np.random.default_rng(seed)
    wmap = Map()
    kfs = []
    K = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1.0]])
    for k in range(n_kf):
        th = 0.03 * k
        T = np.eye(4)
        T[:3, :3] = [[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]]
        T[:3, 3] = [-0.4 * k, 0.01 * k, 0.02 * k]
        wmap.add_pose(T, is_keyframe=True)
        kfs.append(types.SimpleNamespace(pose=T.copy(), kps=[]))
    pts = np.column_stack([rng.uniform(-4, 4, n_pts), rng.uniform(-1, 1, n_pts),
                           rng.uniform(6, 20, n_pts)])
    ids = wmap.add_points(pts)
    obs_rows = []
    for pid in ids:
        first = int(rng.integers(0, n_kf - 1))
        n_obs = int(rng.integers(1, 6))
        for k in range(first, min(n_kf, first + n_obs)):
            Xc = kfs[k].pose[:3, :3] @ wmap.points[pid].position + kfs[k].pose[:3, 3]
            u = K[0, 0] * Xc[0] / Xc[2] + K[0, 2] + rng.normal(0, 1.0)
            v = K[1, 1] * Xc[1] / Xc[2] + K[1, 2] + rng.normal(0, 1.0)
            kfs[k].kps.append(cv2.KeyPoint(u, v, 1))
            wmap.points[pid].add_observation(k, len(kfs[k].kps) - 1,
                                             rng.standard_normal(8).astype(np.float32))
            obs_rows.append((pid, k, len(kfs[k].kps) - 1, u, v))
    return wmap, kfs, K, np.array(obs_rows, np.float64)


def main(out="tests/golden/ba_assembly.npz"):
    cv2 = _install_stubs()
    sys.path.insert(0, "/root/reference")
    from slam.core import ba_utils

    wmap, kfs, K, obs_rows = _scene(cv2)
    in_poses = np.array([k.pose.copy() for k in kfs])
    in_points = np.array([wmap.points[i].position.copy() for i in wmap.points])
    in_point_ids = np.array(list(wmap.points.keys()))
    center, window, max_points, max_iters = 6, 3, 20, 7

    ba_utils.local_bundle_adjustment(wmap, K, kfs, center, window_size=window,
                                     max_points=max_points, max_iters=max_iters)
    prob = REC["problem"]
    rec_max_iters = int(REC["max_iters"])

    # classify blocks
    kinds, owners = [], []
    pos_id = {id(mp.position): pid for pid, mp in wmap.points.items()}
    for b, sz in zip(prob.blocks, prob.sizes):
        if sz == 3 and id(b) in pos_id:
            kinds.append(2); owners.append(pos_id[id(b)])       # point
        elif sz == 4 and id(b) in prob.manifold:
            kinds.append(0); owners.append(-1)                   # quaternion
        elif sz == 4:
            kinds.append(3); owners.append(-1)                   # intrinsics
        else:
            kinds.append(1); owners.append(-1)                   # translation
    const = [int(id(b) in prob.constant) for b in prob.blocks]

    res_uv = np.array([r[0] for r in prob.residuals])
    res_delta = np.array([r[1] for r in prob.residuals])
    blk_index = {id(b): i for i, b in enumerate(prob.blocks)}
    res_blocks = np.array([[blk_index[id(p)] for p in r[2]] for r in prob.residuals])
    res_pid = np.array([pos_id[id(r[2][2])] for r in prob.residuals])

    # (kf index of each residual): recover through uv lookup in obs_rows
    res_kf = []
    for uv, pid in zip(res_uv, res_pid):
        m = (obs_rows[:, 0] == pid) & np.isclose(obs_rows[:, 3], uv[0]) & np.isclose(obs_rows[:, 4], uv[1])
        res_kf.append(int(obs_rows[m][0, 1]))
    res_kf = np.array(res_kf)

    out_kf_poses = np.array([k.pose.copy() for k in kfs])
    out_map_poses = np.array([p.copy() for p in wmap.poses])
    out_points = np.array([wmap.points[i].position.copy() for i in wmap.points])

    # second call: too few residuals -> early return, nothing solved
    REC.pop("max_iters", None)
    wm2, kf2, K2, _ = _scene(cv2, n_kf=3, n_pts=3, seed=5)
    ba_utils.local_bundle_adjustment(wm2, K2, kf2, 2, window_size=2, max_points=10, max_iters=5)
    few_n_res = len(REC["problem"].residuals)
    few_solved = int("max_iters" in REC)

    np.savez(out,
             K=K, in_poses=in_poses, in_points=in_points, in_point_ids=in_point_ids,
             obs_rows=obs_rows, center=center, window=window, max_points=max_points,
             max_iters=max_iters,
             block_kind=np.array(kinds), block_owner=np.array(owners),
             block_const=np.array(const), block_size=np.array(prob.sizes),
             res_uv=res_uv, res_delta=res_delta, res_blocks=res_blocks,
             res_pid=res_pid, res_kf=res_kf, rec_max_iters=rec_max_iters,
             out_kf_poses=out_kf_poses, out_map_poses=out_map_poses, out_points=out_points,
             few_n_res=few_n_res, few_solved=few_solved)
    print("wrote", out, "blocks", len(prob.blocks), "residuals", len(prob.residuals),
          "few:", few_n_res, few_solved)



if __name__ == "__main__":
    main()
