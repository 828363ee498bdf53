"""Levenberg-Marquardt driver for local BA around the HIP residual/Jacobian
kernel (`sslam_ba_residual_jacobian_*`).

Stands where `pyceres.solve(opts, problem, summary)` stands in the reference
(slam/core/ba_utils.py:288-293).  The per-observation arithmetic (residual,
three Jacobian blocks) runs on the GPU; this module does what Ceres' trust
region loop does around it - Huber re-weighting (HuberLoss(2.0), Ceres
corrector with rho'' <= 0 -> plain sqrt(rho') scaling), the quaternion
manifold (EigenQuaternionManifold plus-Jacobian), normal equations reduced by
the Schur complement onto the <= window_size poses, LM radius update with
Ceres' defaults (initial radius 1e4, min_relative_decrease 1e-3,
function/gradient/parameter tolerances 1e-6 / 1e-10 / 1e-8).

Two drivers share that algorithm:

* `solve_device` - `sslam_ba_solve_host` (csrc/ba_lm.hip): the whole loop on the GPU, control
  state in a device control block, one enqueue per solve.
  Takes up to MAX_DEVICE_POSES optimised poses: local BA (window_size 6 / 10, reduced system
  factored in LDS) and global BA of a map of that many keyframes (factored in device memory)
  as long as its dense Schur operands ([poses][points][18] twice) stay under
  MAX_DEVICE_SCHUR_BYTES.
* `solve_host` - the numpy loop below around the HIP residual/Jacobian kernel; used beyond
  those bounds and as the independent restatement the device path is tested against.  Its
  Schur complement is accumulated sparsely, per pair of observations of one landmark.

`solve` picks by problem size (override: SSLAM_BA_SOLVER=host|device).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import _native


@dataclass
class BAProblem:
    """SoA snapshot of one `_core_ba` problem."""
    q: np.ndarray            # [P,4] xyzw
    t: np.ndarray            # [P,3]
    pose_const: np.ndarray   # [P] bool
    X: np.ndarray            # [Q,3]
    intr: np.ndarray         # [4]
    obs_pose: np.ndarray     # [n] int32 -> row of q/t
    obs_point: np.ndarray    # [n] int32 -> row of X
    obs_uv: np.ndarray       # [n,2]


@dataclass
class BASummary:
    iterations: int = 0
    successful_steps: int = 0
    initial_cost: float = float("nan")
    final_cost: float = float("nan")
    termination: str = ""


def _plus_jacobian(q):
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return np.stack([np.stack([w, z, -y], -1), np.stack([-z, w, x], -1),
                     np.stack([y, -x, w], -1), np.stack([-x, -y, -z], -1)], -2)   # [P,4,3]


def _quat_plus(q, delta):
    """q_new = exp(delta) (x) q, batched; q [P,4] xyzw, delta [P,3]."""
    nd = np.linalg.norm(delta, axis=1)
    s = np.where(nd > 0, np.sin(nd) / np.where(nd > 0, nd, 1.0), 1.0)
    dx, dy, dz = s * delta[:, 0], s * delta[:, 1], s * delta[:, 2]
    dw = np.cos(nd)
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return np.stack([dw * x + dx * w + dy * z - dz * y,
                     dw * y - dx * z + dy * w + dz * x,
                     dw * z + dx * y - dy * x + dz * w,
                     dw * w - dx * x - dy * y - dz * z], axis=1)


class _Evaluator:
    def __init__(self, prob: BAProblem, ctx):
        self.p = prob
        self.ctx = ctx
        n = len(prob.obs_pose)
        self.r = np.empty((n, 2))
        self.Jq = np.empty((n, 2, 4))
        self.Jt = np.empty((n, 2, 3))
        self.JX = np.empty((n, 2, 3))

    def __call__(self, q, t, X, jac: bool):
        p = self.p
        L = _native.lib()
        q = np.ascontiguousarray(q)
        t = np.ascontiguousarray(t)
        X = np.ascontiguousarray(X)
        P = _native.ptr
        _native.check(L.sslam_ba_residual_jacobian_host(
            self.ctx.handle, len(p.obs_pose), P(p.obs_pose), P(p.obs_point), P(p.obs_uv),
            len(q), P(q), P(t), len(X), P(X), P(p.intr), P(self.r),
            P(self.Jq) if jac else None, P(self.Jt) if jac else None, P(self.JX) if jac else None),
            "sslam_ba_residual_jacobian_host")
        return self.r


def _huber(s, delta):
    b = delta * delta
    rt = np.sqrt(np.maximum(s, 1e-300))
    rho = np.where(s > b, 2.0 * delta * rt - b, s)
    w = np.where(s > b, delta / rt, 1.0)
    return rho, w


MAX_DEVICE_POSES = 256                 # csrc/ba_lm.hip MAX_PO_BIG
MAX_DEVICE_SCHUR_BYTES = 32 << 30      # dense [Po][Q][18] f64 x 2 (the library refuses more)
_PAIR_CHUNK = 1 << 18        # (a, b) observation pairs per accumulation chunk of the host Schur loop
_TERMINATION = {0: "max iterations", 1: "gradient tolerance", 2: "parameter tolerance",
                3: "function tolerance", 4: "trust region collapsed"}


def solve_device(prob: BAProblem, max_iters: int, huber_delta: float = 2.0, ctx=None,
                 points_const: bool = False) -> BASummary:
    """Device-resident LM (csrc/ba_lm.hip).  Optimises prob.q/t/X in place."""
    ctx = ctx or _native.default_context()
    P = _native.ptr
    q = np.ascontiguousarray(prob.q, np.float64).copy()
    t = np.ascontiguousarray(prob.t, np.float64).copy()
    X = np.ascontiguousarray(prob.X, np.float64).copy()
    const = np.ascontiguousarray(prob.pose_const, np.uint8)
    obs_pose = np.ascontiguousarray(prob.obs_pose, np.int32)
    obs_point = np.ascontiguousarray(prob.obs_point, np.int32)
    uv = np.ascontiguousarray(prob.obs_uv, np.float64)
    intr = np.ascontiguousarray(prob.intr, np.float64)
    out = np.zeros(8, np.float64)
    _native.check(_native.lib().sslam_ba_solve_host(
        ctx.handle, len(obs_pose), P(obs_pose), P(obs_point), P(uv), len(q), P(q), P(t), P(const),
        len(X), P(X), P(intr), int(max_iters), float(huber_delta), int(bool(points_const)), P(out)),
        "sslam_ba_solve_host")
    prob.q[:] = q
    prob.t[:] = t
    prob.X[:] = X
    return BASummary(iterations=int(out[0]), successful_steps=int(out[1]), initial_cost=float(out[2]),
                     final_cost=float(out[3]), termination=_TERMINATION.get(int(out[4]), "?"))


def solve(prob: BAProblem, max_iters: int, huber_delta: float = 2.0, ctx=None,
          points_const: bool = False) -> BASummary:
    """Optimise prob.q/t (non-constant rows) and prob.X in place.  With `points_const` the
    landmarks are held fixed (pose-only BA).  Device LM when the window fits, else host loop."""
    import os
    mode = os.environ.get("SSLAM_BA_SOLVER", "auto")
    n_opt = int(np.count_nonzero(~np.asarray(prob.pose_const, bool)))
    fits = n_opt <= MAX_DEVICE_POSES and n_opt * len(prob.X) * 288 <= MAX_DEVICE_SCHUR_BYTES
    if mode == "device" or (mode == "auto" and fits and int(max_iters) > 0):
        return solve_device(prob, max_iters, huber_delta, ctx, points_const)
    return solve_host(prob, max_iters, huber_delta, ctx, points_const)


def solve_host(prob: BAProblem, max_iters: int, huber_delta: float = 2.0, ctx=None,
               points_const: bool = False) -> BASummary:
    """numpy LM / Schur loop around the HIP residual + Jacobian kernel."""
    ctx = ctx or _native.default_context()
    ev = _Evaluator(prob, ctx)
    summ = BASummary()

    opt_rows = np.flatnonzero(~prob.pose_const)
    slot = -np.ones(len(prob.q), np.int64)
    slot[opt_rows] = np.arange(len(opt_rows))
    Po, Q = len(opt_rows), len(prob.X)
    obs_slot = slot[prob.obs_pose]                 # -1 for fixed poses
    has_pose = obs_slot >= 0
    oi = np.flatnonzero(has_pose)

    # observations of optimised poses sorted by landmark, and every ordered pair (a, b) of them that
    # shares a landmark: the Schur complement couples exactly those (sum_j k_j^2 pairs)
    so = oi[np.argsort(prob.obs_point[oi], kind="stable")]
    so_point = prob.obs_point[so].astype(np.int64)
    so_slot = obs_slot[so]
    if len(so):
        k_of_point = np.bincount(so_point, minlength=Q)
        start_of_point = np.concatenate([[0], np.cumsum(k_of_point)[:-1]])
        k_a = k_of_point[so_point]
        pair_a = np.repeat(np.arange(len(so)), k_a)
        first = np.concatenate([[0], np.cumsum(k_a)[:-1]])
        pair_b = np.repeat(start_of_point[so_point], k_a) + (np.arange(len(pair_a)) - np.repeat(first, k_a))
    else:
        pair_a = pair_b = np.zeros(0, np.int64)

    q, t, X = prob.q.copy(), prob.t.copy(), prob.X.copy()

    def cost_of(r):
        rho, _ = _huber(np.sum(r * r, axis=1), huber_delta)
        return 0.5 * float(np.sum(rho))

    r = ev(q, t, X, True).copy()
    cost = cost_of(r)
    summ.initial_cost = cost
    radius, decrease = 1e4, 2.0
    need_jac = False          # Jacobian buffers currently match (q,t,X)

    for it in range(int(max_iters)):
        summ.iterations = it + 1
        if need_jac:
            r = ev(q, t, X, True).copy()
            need_jac = False
        s = np.sum(r * r, axis=1)
        _, w = _huber(s, huber_delta)
        sw = np.sqrt(w)
        rw = r * sw[:, None]
        JXw = ev.JX * sw[:, None, None]
        if points_const:
            JXw = np.zeros_like(JXw)       # constant blocks get no Jacobian -> dX == 0
        # pose Jacobian in the tangent space: [Jq @ plus(q) | Jt]  -> [n,2,6]
        Jp = np.zeros((len(r), 2, 6))
        if Po:
            pj = _plus_jacobian(q)[prob.obs_pose[oi]]
            Jp[oi, :, :3] = ev.Jq[oi] @ pj
            Jp[oi, :, 3:] = ev.Jt[oi]
            Jp *= sw[:, None, None]

        # ---- normal equations (block sparse) --------------------------------
        V = np.zeros((Q, 3, 3))
        np.add.at(V, prob.obs_point, np.einsum("nia,nib->nab", JXw, JXw))
        gX = np.zeros((Q, 3))
        np.add.at(gX, prob.obs_point, np.einsum("nia,ni->na", JXw, rw))
        U = np.zeros((Po, 6, 6))
        gP = np.zeros((Po, 6))
        if Po:
            np.add.at(U, obs_slot[oi], np.einsum("nia,nib->nab", Jp[oi], Jp[oi]))
            np.add.at(gP, obs_slot[oi], np.einsum("nia,ni->na", Jp[oi], rw[oi]))
            # off-diagonal blocks W = Jp^T JX are kept PER OBSERVATION (sorted by landmark), never
            # as a dense [pose, point] array: memory and work follow the sparsity of the problem
            Wo = np.einsum("nia,nib->nab", Jp[so], JXw[so])       # [m,6,3]

        gmax = max(np.abs(gX).max(initial=0.0), np.abs(gP).max(initial=0.0))
        if gmax < 1e-10:
            summ.termination = "gradient tolerance"
            break

        # ---- LM step via Schur complement -----------------------------------
        dV = np.clip(np.einsum("qaa->qa", V), 1e-6, 1e32) / radius
        dU = np.clip(np.einsum("paa->pa", U), 1e-6, 1e32) / radius if Po else np.zeros((0, 6))
        Vd = V + np.einsum("qa,ab->qab", dV, np.eye(3))
        Vinv = np.linalg.inv(Vd)
        if Po:
            Yo = Wo @ Vinv[so_point]                              # [m,6,3]  Y = W V^-1
            # S = U - sum_j sum_{a,b in obs(j)} Y_a W_b^T, accumulated pair by pair in chunks
            S4 = np.zeros((Po, Po, 6, 6))
            for c0 in range(0, len(pair_a), _PAIR_CHUNK):
                pa, pb = pair_a[c0:c0 + _PAIR_CHUNK], pair_b[c0:c0 + _PAIR_CHUNK]
                np.subtract.at(S4, (so_slot[pa], so_slot[pb]), np.einsum("nab,ncb->nac", Yo[pa], Wo[pb]))
            Ud = U + np.einsum("pa,ab->pab", dU, np.eye(6))
            S4[np.arange(Po), np.arange(Po)] += Ud
            S = S4.transpose(0, 2, 1, 3).reshape(6 * Po, 6 * Po)
            YgX = np.zeros((Po, 6))
            np.add.at(YgX, so_slot, np.einsum("nab,nb->na", Yo, gX[so_point]))
            rhs = -(gP - YgX).reshape(-1)
            try:
                c = np.linalg.cholesky(S)
                dP = np.linalg.solve(c.T, np.linalg.solve(c, rhs)).reshape(Po, 6)
            except np.linalg.LinAlgError:
                dP = np.linalg.lstsq(S, rhs, rcond=None)[0].reshape(Po, 6)
            WtdP = np.zeros((Q, 3))
            np.add.at(WtdP, so_point, np.einsum("nab,na->nb", Wo, dP[so_slot]))
            dX = np.einsum("jab,jb->ja", Vinv, -gX - WtdP)
        else:
            dP = np.zeros((0, 6))
            dX = np.einsum("jab,jb->ja", Vinv, -gX)

        # model cost change = -(J d)^T (r + J d / 2) on the re-weighted system
        Jd = np.einsum("nia,na->ni", JXw, dX[prob.obs_point])
        if Po:
            Jd[oi] += np.einsum("nia,na->ni", Jp[oi], dP[obs_slot[oi]])
        model_change = -float(np.sum(Jd * (rw + 0.5 * Jd)))

        step_norm = np.sqrt(np.sum(dX * dX) + np.sum(dP * dP))
        x_norm = np.sqrt(np.sum(X * X) + np.sum(q[opt_rows] ** 2) + np.sum(t[opt_rows] ** 2))
        if step_norm <= 1e-8 * (x_norm + 1e-8):
            summ.termination = "parameter tolerance"
            break

        q_new, t_new = q.copy(), t.copy()
        if Po:
            q_new[opt_rows] = _quat_plus(q[opt_rows], dP[:, :3])
            t_new[opt_rows] = t[opt_rows] + dP[:, 3:]
        X_new = X + dX
        r_new = ev(q_new, t_new, X_new, False)
        new_cost = cost_of(r_new) if np.all(np.isfinite(r_new)) else np.inf
        rel = (cost - new_cost) / model_change if model_change > 0 else -1.0

        if rel > 1e-3 and np.isfinite(new_cost):
            change = cost - new_cost
            q, t, X = q_new, t_new, X_new
            cost = new_cost
            summ.successful_steps += 1
            radius = min(1e16, radius / max(1.0 / 3.0, 1.0 - (2.0 * rel - 1.0) ** 3))
            decrease = 2.0
            need_jac = True
            if abs(change) < 1e-6 * cost:
                summ.termination = "function tolerance"
                break
        else:
            radius /= decrease
            decrease *= 2.0
            if radius < 1e-32:
                summ.termination = "trust region collapsed"
                break
    else:
        summ.termination = "max iterations"

    prob.q[:] = q
    prob.t[:] = t
    prob.X[:] = X
    summ.final_cost = cost
    return summ


def solve_pose_only(prob: BAProblem, max_iters: int, huber_delta: float = 2.0, ctx=None) -> BASummary:
    """`pose_only_ba` of the reference (ba_utils.py:89-140): landmarks constant."""
    return solve(prob, max_iters, huber_delta, ctx, points_const=True)
