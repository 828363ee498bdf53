// ba_lm.hip - device-resident Levenberg-Marquardt for local BA (fp64).
//
// Stands where `pyceres.solve(opts, problem, summary)` stands in the reference
// (slam/core/ba_utils.py:288-293) for the sliding-window problem `_core_ba` builds
// (:220-286): <= 12 optimised poses (window_size 6 / 10), all earlier keyframes constant,
// <= max_points landmarks, Huber(2.0) on every reprojection block, Eigen-quaternion manifold -
// and, r03, for `global_bundle_adjustment`'s problem (:170-218) up to 256 optimised poses (the
// reduced system is then factored in device memory, lm_solve_big_kernel).
// It is the same algorithm as the host loop in ba_solver.py (Ceres' trust-region policy with
// its default constants, Schur complement onto the poses); here the whole loop - linearise,
// reduce, factor, step, evaluate, accept/reject, terminate - runs as a FIXED launch sequence
// whose control state lives in a device control block, so a solve is one enqueue and one
// read-back, with no host round trip per iteration.
//
// Every reduction is order-fixed (per-point loops over a CSR of observations, column sums over
// LDS slabs, two-stage partial sums), so a solve is bit-reproducible run to run.
//
// Roofline: HBM / latency.  Per iteration ~ n_obs * (21 + 20) f64 written + read for the
// linearisation (5 MB at 30 k observations) plus the dense Schur operands (Po * Q * 36 f64);
// the reduced system (<= 72 x 72) is factored by one workgroup in LDS.
#include "common.hpp"
#include "ba_math.hpp"

namespace {

constexpr int LM_T = 256;          // threads per block, element-wise kernels
constexpr int LM_RT = 128;         // threads per block, slab reductions
constexpr int MAX_PO = 12;         // optimised poses whose reduced system is factored in LDS (local BA: window 6 / 10)
constexpr int MAX_M = 6 * MAX_PO;
constexpr int MAX_PO_BIG = 256;    // optimised poses the device path takes at all (global BA): the reduced system is then
constexpr int MAX_M_BIG = 6 * MAX_PO_BIG;   // factored in place in device memory by one 1024-thread workgroup
constexpr int LM_BT = 1024;

struct LMCtrl {
    int cur;            // which (q, t, X) buffer holds the accepted iterate
    int done;           // 0 running, 1 gradient tol, 2 parameter tol, 3 function tol, 4 radius collapsed
    int need_lin;       // Jacobian buffers must be rebuilt (first iteration / after an accepted step)
    int iterations, successful, chol_fail, pad0, pad1;
    double radius, decrease, cost, initial_cost;
    double step2_pose, x2_pose;
};

struct LMArgs {
    int n_obs, P, Q, Po, points_const, nb_obs, nb_pt;
    double delta;
    const int32_t *obs_pose, *obs_point, *obs_slot;      // [n]
    const double* uv;                                    // [n][2]
    const int32_t *pt_ptr, *pt_obs;                      // CSR by point
    const int32_t *ps_ptr, *ps_obs;                      // CSR by optimised-pose slot
    const int32_t* slot_pose;                            // [Po] -> pose row
    const int32_t* pose_slot;                            // [P]  -> slot or -1
    const double* intr;
    double *q[2], *t[2], *X[2];
    double *rw, *JXw, *Jpw;                              // [n][2], [n][6], [n][12]
    double *V, *gX, *Vinv;                               // [Q][6], [Q][3], [Q][6]
    double *Wd, *Y;                                      // [Po][Q][18]
    double *U, *gP;                                      // [Po][36], [Po][6]
    double *S, *rhs, *dP;                                // [m][m], [m], [Po][6]
    double* Lbig;                                        // [m][m] factor of the reduced system when m > MAX_M
    double* dX;                                          // [Q][3]
    double *pc, *pm, *pstep, *px, *pgmax;                // block partials
    int* pbad;
    LMCtrl* ctrl;
};

__device__ __forceinline__ double block_sum(double v, double* sh) {
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

__device__ __forceinline__ double block_max(double v, double* sh) {
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

__device__ __forceinline__ void load_obs(const LMArgs& a, int i, int buf, sslam::BAObsIn& in) {
    const int pi = a.obs_pose[i], xi = a.obs_point[i];
    const double *q = a.q[buf], *t = a.t[buf], *X = a.X[buf];
    in.ax = q[4 * pi + 0]; in.ay = q[4 * pi + 1]; in.az = q[4 * pi + 2]; in.w = q[4 * pi + 3];
    in.tx = t[3 * pi + 0]; in.ty = t[3 * pi + 1]; in.tz = t[3 * pi + 2];
    in.Xx = X[3 * xi + 0]; in.Xy = X[3 * xi + 1]; in.Xz = X[3 * xi + 2];
    in.fx = a.intr[0]; in.fy = a.intr[1]; in.cx = a.intr[2]; in.cy = a.intr[3];
    in.u = a.uv[2 * i]; in.v = a.uv[2 * i + 1];
}

// Huber(delta) on s = |r|^2: rho and the IRLS weight rho' (Ceres corrector with rho'' <= 0)
__device__ __forceinline__ void huber(double s, double delta, double& rho, double& w) {
    const double b = delta * delta;
    const double rt = sqrt(fmax(s, 1e-300));
    rho = s > b ? 2.0 * delta * rt - b : s;
    w = s > b ? delta / rt : 1.0;
}

// ---- 1. linearise at the current iterate (thread / observation) -----------------------------
__global__ __launch_bounds__(LM_T) void lm_linearise_kernel(LMArgs a) {
    const LMCtrl* c = a.ctrl;
    if (c->done || !c->need_lin) return;
    const int i = blockIdx.x * LM_T + threadIdx.x;
    if (i >= a.n_obs) return;
    sslam::BAObsIn in;
    load_obs(a, i, c->cur, in);
    double r0, r1, jq[8], jt[6], jx[6];
    sslam::ba_reproj<true>(in, r0, r1, jq, jt, jx);
    double rho, w;
    huber(r0 * r0 + r1 * r1, a.delta, rho, w);
    const double sw = sqrt(w);
    a.rw[2 * i] = r0 * sw;
    a.rw[2 * i + 1] = r1 * sw;
#pragma unroll
    for (int k = 0; k < 6; ++k) a.JXw[6 * (size_t)i + k] = a.points_const ? 0.0 : jx[k] * sw;
    if (a.obs_slot[i] >= 0) {
        // tangent-space pose Jacobian [Jq . plus(q) | Jt]; plus(q) rows (EigenQuaternionManifold):
        //   [ w  z -y ; -z  w  x ;  y -x  w ; -x -y -z ]
        const double x = in.ax, y = in.ay, z = in.az, ww = in.w;
#pragma unroll
        for (int row = 0; row < 2; ++row) {
            const double j0 = jq[4 * row], j1 = jq[4 * row + 1], j2 = jq[4 * row + 2], j3 = jq[4 * row + 3];
            double* o = a.Jpw + 12 * (size_t)i + 6 * row;
            o[0] = (((j0 * ww) + (j1 * -z)) + (j2 * y)) + (j3 * -x);
            o[1] = (((j0 * z) + (j1 * ww)) + (j2 * -x)) + (j3 * -y);
            o[2] = (((j0 * -y) + (j1 * x)) + (j2 * ww)) + (j3 * -z);
            o[0] *= sw; o[1] *= sw; o[2] *= sw;
            o[3] = jt[3 * row] * sw; o[4] = jt[3 * row + 1] * sw; o[5] = jt[3 * row + 2] * sw;
        }
    }
}

// ---- 2. per landmark: V, gX, damped inverse, dense W / Y columns (thread / point) -----------
__global__ __launch_bounds__(LM_T) void lm_point_kernel(LMArgs a) {
    __shared__ double sh[LM_T];
    const LMCtrl* c = a.ctrl;
    if (c->done) return;
    const int j = blockIdx.x * LM_T + threadIdx.x;
    double gm = 0.0;
    if (j < a.Q) {
        double V[6], g[3];
        const int o0 = a.pt_ptr[j], o1 = a.pt_ptr[j + 1];
        if (c->need_lin) {
#pragma unroll
            for (int k = 0; k < 6; ++k) V[k] = 0.0;
            g[0] = g[1] = g[2] = 0.0;
            for (int e = o0; e < o1; ++e) {
                const int i = a.pt_obs[e];
                const double* J = a.JXw + 6 * (size_t)i;
                const double ra = a.rw[2 * i], rb = a.rw[2 * i + 1];
                V[0] += J[0] * J[0] + J[3] * J[3]; V[1] += J[0] * J[1] + J[3] * J[4]; V[2] += J[0] * J[2] + J[3] * J[5];
                V[3] += J[1] * J[1] + J[4] * J[4]; V[4] += J[1] * J[2] + J[4] * J[5]; V[5] += J[2] * J[2] + J[5] * J[5];
                g[0] += J[0] * ra + J[3] * rb; g[1] += J[1] * ra + J[4] * rb; g[2] += J[2] * ra + J[5] * rb;
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) a.V[6 * (size_t)j + k] = V[k];
            a.gX[3 * j] = g[0]; a.gX[3 * j + 1] = g[1]; a.gX[3 * j + 2] = g[2];
            // dense columns of W = Jp^T JX (zero where the pose does not see the point)
            for (int p = 0; p < a.Po; ++p) {
                double* w = a.Wd + ((size_t)p * a.Q + j) * 18;
#pragma unroll
                for (int k = 0; k < 18; ++k) w[k] = 0.0;
            }
            for (int e = o0; e < o1; ++e) {
                const int i = a.pt_obs[e], sl = a.obs_slot[i];
                if (sl < 0) continue;
                const double* Jp = a.Jpw + 12 * (size_t)i;
                const double* J = a.JXw + 6 * (size_t)i;
                double* w = a.Wd + ((size_t)sl * a.Q + j) * 18;
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) w[3 * r + cc] += Jp[r] * J[cc] + Jp[6 + r] * J[3 + cc];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) V[k] = a.V[6 * (size_t)j + k];
            g[0] = a.gX[3 * j]; g[1] = a.gX[3 * j + 1]; g[2] = a.gX[3 * j + 2];
        }
        gm = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
        // LM damping: D = diag(clip(diag V, 1e-6, 1e32)) / radius  (Ceres min/max_lm_diagonal)
        const double rad = c->radius;
        const double a00 = V[0] + fmin(fmax(V[0], 1e-6), 1e32) / rad;
        const double a11 = V[3] + fmin(fmax(V[3], 1e-6), 1e32) / rad;
        const double a22 = V[5] + fmin(fmax(V[5], 1e-6), 1e32) / rad;
        const double a01 = V[1], a02 = V[2], a12 = V[4];
        const double c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
        const double det = a00 * c00 + a01 * c01 + a02 * c02;
        const double id = 1.0 / det;
        double I[6];
        I[0] = c00 * id; I[1] = c01 * id; I[2] = c02 * id;
        I[3] = (a00 * a22 - a02 * a02) * id; I[4] = (a01 * a02 - a00 * a12) * id; I[5] = (a00 * a11 - a01 * a01) * id;
#pragma unroll
        for (int k = 0; k < 6; ++k) a.Vinv[6 * (size_t)j + k] = I[k];
        // Y = W Vinv (6x3 . 3x3 symmetric)
        for (int p = 0; p < a.Po; ++p) {
            const double* w = a.Wd + ((size_t)p * a.Q + j) * 18;
            double* y = a.Y + ((size_t)p * a.Q + j) * 18;
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const double w0 = w[3 * r], w1 = w[3 * r + 1], w2 = w[3 * r + 2];
                y[3 * r] = w0 * I[0] + w1 * I[1] + w2 * I[2];
                y[3 * r + 1] = w0 * I[1] + w1 * I[3] + w2 * I[4];
                y[3 * r + 2] = w0 * I[2] + w1 * I[4] + w2 * I[5];
            }
        }
    }
    const double bm = block_max(gm, sh);
    if (threadIdx.x == 0) a.pgmax[blockIdx.x] = bm;
}

// ---- 3. per optimised pose: U = sum Jp^T Jp, gP = sum Jp^T r (block / slot) ------------------
__global__ __launch_bounds__(LM_RT) void lm_pose_kernel(LMArgs a) {
    __shared__ double slab[LM_RT][43];
    const LMCtrl* c = a.ctrl;
    if (c->done || !c->need_lin) return;
    const int p = blockIdx.x;
    double acc[42];
#pragma unroll
    for (int k = 0; k < 42; ++k) acc[k] = 0.0;
    for (int e = a.ps_ptr[p] + threadIdx.x; e < a.ps_ptr[p + 1]; e += LM_RT) {
        const int i = a.ps_obs[e];
        const double* Jp = a.Jpw + 12 * (size_t)i;
        const double ra = a.rw[2 * i], rb = a.rw[2 * i + 1];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) acc[6 * r + cc] += Jp[r] * Jp[cc] + Jp[6 + r] * Jp[6 + cc];
            acc[36 + r] += Jp[r] * ra + Jp[6 + r] * rb;
        }
    }
#pragma unroll
    for (int k = 0; k < 42; ++k) slab[threadIdx.x][k] = acc[k];
    __syncthreads();
    if (threadIdx.x < 42) {
        double s = 0.0;
        for (int r = 0; r < LM_RT; ++r) s += slab[r][threadIdx.x];
        if (threadIdx.x < 36) a.U[36 * p + threadIdx.x] = s;
        else a.gP[6 * p + threadIdx.x - 36] = s;
    }
}

// ---- 4. reduced camera system: S = U + D - sum_j Y_a W_b^T, rhs = -(gP - sum_j Y_a gX) -------
__global__ __launch_bounds__(LM_RT) void lm_schur_kernel(LMArgs a) {
    __shared__ double slab[LM_RT][37];
    const LMCtrl* c = a.ctrl;
    if (c->done) return;
    const int pa = blockIdx.x, pb = blockIdx.y, m = 6 * a.Po;
    const bool is_rhs = pb == a.Po;
    double acc[36];
#pragma unroll
    for (int k = 0; k < 36; ++k) acc[k] = 0.0;
    for (int j = threadIdx.x; j < a.Q; j += LM_RT) {
        const double* y = a.Y + ((size_t)pa * a.Q + j) * 18;
        if (is_rhs) {
            const double g0 = a.gX[3 * j], g1 = a.gX[3 * j + 1], g2 = a.gX[3 * j + 2];
#pragma unroll
            for (int r = 0; r < 6; ++r) acc[r] += y[3 * r] * g0 + y[3 * r + 1] * g1 + y[3 * r + 2] * g2;
        } else {
            const double* w = a.Wd + ((size_t)pb * a.Q + j) * 18;
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int cc = 0; cc < 6; ++cc)
                    acc[6 * r + cc] += y[3 * r] * w[3 * cc] + y[3 * r + 1] * w[3 * cc + 1] + y[3 * r + 2] * w[3 * cc + 2];
        }
    }
#pragma unroll
    for (int k = 0; k < 36; ++k) slab[threadIdx.x][k] = acc[k];
    __syncthreads();
    if (threadIdx.x < (is_rhs ? 6 : 36)) {
        double s = 0.0;
        for (int r = 0; r < LM_RT; ++r) s += slab[r][threadIdx.x];
        if (is_rhs) {
            a.rhs[6 * pa + threadIdx.x] = -(a.gP[6 * pa + threadIdx.x] - s);
        } else {
            const int r = threadIdx.x / 6, cc = threadIdx.x % 6;
            double v = -s;
            if (pa == pb) {
                const double u = a.U[36 * pa + threadIdx.x];
                v += u;
                if (r == cc) v += fmin(fmax(u, 1e-6), 1e32) / c->radius;
            }
            a.S[(size_t)(6 * pa + r) * m + 6 * pb + cc] = v;
        }
    }
}

// ---- 5. gradient test + dense Cholesky solve of the reduced system (one block) ---------------
__global__ __launch_bounds__(LM_T) void lm_solve_kernel(LMArgs a) {
    __shared__ double L[MAX_M * (MAX_M + 1)];
    __shared__ double sh[LM_T];
    __shared__ int fail;
    LMCtrl* c = a.ctrl;
    if (c->done) return;
    const int t = threadIdx.x, m = 6 * a.Po, ld = MAX_M + 1;
    double gm = 0.0;
    for (int k = t; k < a.nb_pt; k += LM_T) gm = fmax(gm, a.pgmax[k]);
    for (int k = t; k < m; k += LM_T) gm = fmax(gm, fabs(a.gP[k]));
    gm = block_max(gm, sh);
    if (t == 0) { c->iterations += 1; c->chol_fail = 0; fail = 0; }
    __syncthreads();
    if (gm < 1e-10) {
        if (t == 0) c->done = 1;
        return;
    }
    if (m == 0) return;
    for (int e = t; e < m * m; e += LM_T) L[(e / m) * ld + e % m] = a.S[e];
    __syncthreads();
    for (int k = 0; k < m; ++k) {
        if (t == 0) {
            const double d = L[k * ld + k];
            if (!(d > 0.0) || !isfinite(d)) fail = 1;
            L[k * ld + k] = sqrt(d);
        }
        __syncthreads();
        if (fail) break;
        const double dk = L[k * ld + k];
        for (int i = k + 1 + t; i < m; i += LM_T) L[i * ld + k] /= dk;
        __syncthreads();
        const int rem = m - k - 1;
        for (int e = t; e < rem * rem; e += LM_T) {
            const int i = k + 1 + e / rem, jj = k + 1 + e % rem;
            if (jj <= i) L[i * ld + jj] -= L[i * ld + k] * L[jj * ld + k];
        }
        __syncthreads();
    }
    if (t == 0) {
        if (fail) {
            c->chol_fail = 1;
            for (int k = 0; k < m; ++k) a.dP[k] = 0.0;
        } else {
            double* y = sh;                         // m <= 72 < LM_T
            for (int i = 0; i < m; ++i) {
                double s = a.rhs[i];
                for (int k = 0; k < i; ++k) s -= L[i * ld + k] * y[k];
                y[i] = s / L[i * ld + i];
            }
            for (int i = m - 1; i >= 0; --i) {
                double s = y[i];
                for (int k = i + 1; k < m; ++k) s -= L[k * ld + i] * y[k];
                y[i] = s / L[i * ld + i];
            }
            for (int k = 0; k < m; ++k) a.dP[k] = y[k];
        }
    }
}

// ---- 5b. the same for more than MAX_PO optimised poses (global BA, ba_utils.py:170-218): the factor lives in device
// memory (L2-resident: 242 KB at 29 poses, 18.9 MB at 256), one workgroup of 1024 threads runs the right-looking
// Cholesky column by column and the two triangular solves column-oriented (every y[k] takes its subtractions in the
// same order as the serial loop of the LDS kernel would apply them in the forward pass)
__global__ __launch_bounds__(LM_BT) void lm_solve_big_kernel(LMArgs a) {
    __shared__ double sh[LM_BT];
    __shared__ double y[MAX_M_BIG];
    __shared__ int fail;
    LMCtrl* c = a.ctrl;
    if (c->done) return;
    const int t = threadIdx.x, m = 6 * a.Po;
    double gm = 0.0;
    for (int k = t; k < a.nb_pt; k += LM_BT) gm = fmax(gm, a.pgmax[k]);
    for (int k = t; k < m; k += LM_BT) gm = fmax(gm, fabs(a.gP[k]));
    gm = block_max(gm, sh);
    if (t == 0) { c->iterations += 1; c->chol_fail = 0; fail = 0; }
    __syncthreads();
    if (gm < 1e-10) {
        if (t == 0) c->done = 1;
        return;
    }
    double* L = a.Lbig;
    for (size_t e = t; e < (size_t)m * m; e += LM_BT) L[e] = a.S[e];
    __syncthreads();
    for (int k = 0; k < m; ++k) {
        if (t == 0) {
            const double d = L[(size_t)k * m + k];
            if (!(d > 0.0) || !isfinite(d)) fail = 1;
            L[(size_t)k * m + k] = sqrt(d);
        }
        __syncthreads();
        if (fail) break;
        const double dk = L[(size_t)k * m + k];
        for (int i = k + 1 + t; i < m; i += LM_BT) L[(size_t)i * m + k] /= dk;
        __syncthreads();
        // trailing update of the lower triangle: a wave per row i, lanes over the columns jj <= i
        const int lane = t & 63, wave = t >> 6;
        for (int i = k + 1 + wave; i < m; i += LM_BT / 64) {
            const double lik = L[(size_t)i * m + k];
            for (int jj = k + 1 + lane; jj <= i; jj += 64) L[(size_t)i * m + jj] -= lik * L[(size_t)jj * m + k];
        }
        __syncthreads();
    }
    if (fail) {
        if (t == 0) c->chol_fail = 1;
        for (int k = t; k < m; k += LM_BT) a.dP[k] = 0.0;
        return;
    }
    for (int i = t; i < m; i += LM_BT) y[i] = a.rhs[i];
    __syncthreads();
    for (int i = 0; i < m; ++i) {                     // L y = rhs
        if (t == 0) y[i] /= L[(size_t)i * m + i];
        __syncthreads();
        const double yi = y[i];
        for (int k = i + 1 + t; k < m; k += LM_BT) y[k] -= L[(size_t)k * m + i] * yi;
        __syncthreads();
    }
    for (int i = m - 1; i >= 0; --i) {                // L^T x = y
        if (t == 0) y[i] /= L[(size_t)i * m + i];
        __syncthreads();
        const double yi = y[i];
        for (int k = t; k < i; k += LM_BT) y[k] -= L[(size_t)i * m + k] * yi;
        __syncthreads();
    }
    for (int k = t; k < m; k += LM_BT) a.dP[k] = y[k];
}

// ---- 6. back-substitute landmarks, build the candidate (thread / point) ----------------------
__global__ __launch_bounds__(LM_T) void lm_update_points_kernel(LMArgs a) {
    __shared__ double sh[LM_T];
    const LMCtrl* c = a.ctrl;
    if (c->done) return;
    const int j = blockIdx.x * LM_T + threadIdx.x;
    double s2 = 0.0, x2 = 0.0;
    if (j < a.Q) {
        double b0 = -a.gX[3 * j], b1 = -a.gX[3 * j + 1], b2 = -a.gX[3 * j + 2];
        for (int p = 0; p < a.Po; ++p) {
            const double* w = a.Wd + ((size_t)p * a.Q + j) * 18;
            const double* d = a.dP + 6 * p;
#pragma unroll
            for (int r = 0; r < 6; ++r) { b0 -= w[3 * r] * d[r]; b1 -= w[3 * r + 1] * d[r]; b2 -= w[3 * r + 2] * d[r]; }
        }
        const double* I = a.Vinv + 6 * (size_t)j;
        const double d0 = I[0] * b0 + I[1] * b1 + I[2] * b2;
        const double d1 = I[1] * b0 + I[3] * b1 + I[4] * b2;
        const double d2 = I[2] * b0 + I[4] * b1 + I[5] * b2;
        a.dX[3 * j] = d0; a.dX[3 * j + 1] = d1; a.dX[3 * j + 2] = d2;
        const double* X = a.X[c->cur] + 3 * (size_t)j;
        double* Xn = a.X[c->cur ^ 1] + 3 * (size_t)j;
        Xn[0] = X[0] + d0; Xn[1] = X[1] + d1; Xn[2] = X[2] + d2;
        s2 = d0 * d0 + d1 * d1 + d2 * d2;
        x2 = X[0] * X[0] + X[1] * X[1] + X[2] * X[2];
    }
    const double bs = block_sum(s2, sh), bx = block_sum(x2, sh);
    if (threadIdx.x == 0) { a.pstep[blockIdx.x] = bs; a.px[blockIdx.x] = bx; }
}

// q_new = exp(delta) (x) q (EigenQuaternionManifold::Plus), t_new = t + dt; constant poses copied
__global__ void lm_update_poses_kernel(LMArgs a) {
    LMCtrl* c = a.ctrl;
    if (c->done) return;
    const int cur = c->cur;
    double s2 = 0.0, x2 = 0.0;
    if (threadIdx.x == 0) {
        for (int p = 0; p < a.P; ++p) {
            const double* q = a.q[cur] + 4 * p; const double* t = a.t[cur] + 3 * p;
            double* qn = a.q[cur ^ 1] + 4 * p; double* tn = a.t[cur ^ 1] + 3 * p;
            const int sl = a.pose_slot[p];
            if (sl < 0) {
                qn[0] = q[0]; qn[1] = q[1]; qn[2] = q[2]; qn[3] = q[3];
                tn[0] = t[0]; tn[1] = t[1]; tn[2] = t[2];
                continue;
            }
            const double* d = a.dP + 6 * sl;
            const double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            const double sc = nd > 0.0 ? sin(nd) / nd : 1.0;
            const double dx = sc * d[0], dy = sc * d[1], dz = sc * d[2], dw = cos(nd);
            const double x = q[0], y = q[1], z = q[2], w = q[3];
            qn[0] = dw * x + dx * w + dy * z - dz * y;
            qn[1] = dw * y - dx * z + dy * w + dz * x;
            qn[2] = dw * z + dx * y - dy * x + dz * w;
            qn[3] = dw * w - dx * x - dy * y - dz * z;
            tn[0] = t[0] + d[3]; tn[1] = t[1] + d[4]; tn[2] = t[2] + d[5];
            for (int k = 0; k < 6; ++k) s2 += d[k] * d[k];
            x2 += q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + t[0] * t[0] + t[1] * t[1] + t[2] * t[2];
        }
        c->step2_pose = s2;
        c->x2_pose = x2;
    }
}

// ---- 7. cost at the current / candidate iterate (+ model cost change) (thread / observation) --
template <bool CAND>
__global__ __launch_bounds__(LM_T) void lm_eval_kernel(LMArgs a) {
    __shared__ double sh[LM_T];
    const LMCtrl* c = a.ctrl;
    if (c->done) return;
    const int i = blockIdx.x * LM_T + threadIdx.x;
    double rho = 0.0, mc = 0.0, bad = 0.0;
    if (i < a.n_obs) {
        sslam::BAObsIn in;
        load_obs(a, i, CAND ? c->cur ^ 1 : c->cur, in);
        double r0, r1, w;
        sslam::ba_reproj<false>(in, r0, r1, nullptr, nullptr, nullptr);
        if (!(isfinite(r0) && isfinite(r1))) bad = 1.0;
        huber(r0 * r0 + r1 * r1, a.delta, rho, w);
        if constexpr (CAND) {
            // model cost change = -(J d)^T (r + J d / 2) on the re-weighted system
            const double* J = a.JXw + 6 * (size_t)i;
            const double* d = a.dX + 3 * (size_t)a.obs_point[i];
            double j0 = J[0] * d[0] + J[1] * d[1] + J[2] * d[2];
            double j1 = J[3] * d[0] + J[4] * d[1] + J[5] * d[2];
            const int sl = a.obs_slot[i];
            if (sl >= 0) {
                const double* Jp = a.Jpw + 12 * (size_t)i;
                const double* dp = a.dP + 6 * sl;
                double e0 = 0.0, e1 = 0.0;
#pragma unroll
                for (int k = 0; k < 6; ++k) { e0 += Jp[k] * dp[k]; e1 += Jp[6 + k] * dp[k]; }
                j0 += e0; j1 += e1;
            }
            mc = j0 * (a.rw[2 * i] + 0.5 * j0) + j1 * (a.rw[2 * i + 1] + 0.5 * j1);
        }
    }
    const double br = block_sum(rho, sh);
    const double bm = CAND ? block_sum(mc, sh) : 0.0;
    const double bb = block_max(bad, sh);
    if (threadIdx.x == 0) { a.pc[blockIdx.x] = br; a.pm[blockIdx.x] = bm; a.pbad[blockIdx.x] = bb > 0.0; }
}

__global__ __launch_bounds__(LM_T) void lm_init_kernel(LMArgs a) {
    __shared__ double sh[LM_T];
    LMCtrl* c = a.ctrl;
    double s = 0.0;
    for (int k = threadIdx.x; k < a.nb_obs; k += LM_T) s += a.pc[k];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) {
        c->cost = c->initial_cost = 0.5 * s;
        c->radius = 1e4; c->decrease = 2.0;            // Ceres initial_trust_region_radius
        c->need_lin = 1; c->done = 0; c->iterations = 0; c->successful = 0; c->chol_fail = 0;
    }
}

// ---- 8. accept / reject, trust-region radius, termination (one block) ------------------------
__global__ __launch_bounds__(LM_T) void lm_decide_kernel(LMArgs a) {
    __shared__ double sh[LM_T];
    LMCtrl* c = a.ctrl;
    if (c->done) return;
    double s = 0.0, mcs = 0.0, st = 0.0, xx = 0.0, bad = 0.0;
    for (int k = threadIdx.x; k < a.nb_obs; k += LM_T) { s += a.pc[k]; mcs += a.pm[k]; bad = fmax(bad, (double)a.pbad[k]); }
    for (int k = threadIdx.x; k < a.nb_pt; k += LM_T) { st += a.pstep[k]; xx += a.px[k]; }
    s = block_sum(s, sh); mcs = block_sum(mcs, sh); st = block_sum(st, sh); xx = block_sum(xx, sh);
    bad = block_max(bad, sh);
    if (threadIdx.x != 0) return;
    const double step_norm = sqrt(st + c->step2_pose), x_norm = sqrt(xx + c->x2_pose);
    if (!c->chol_fail && step_norm <= 1e-8 * (x_norm + 1e-8)) { c->done = 2; return; }   // parameter_tolerance
    const double new_cost = (bad > 0.0 || c->chol_fail) ? INFINITY : 0.5 * s;
    const double model_change = -mcs;
    const double rel = model_change > 0.0 ? (c->cost - new_cost) / model_change : -1.0;
    if (rel > 1e-3 && isfinite(new_cost)) {            // min_relative_decrease
        const double change = c->cost - new_cost;
        c->cur ^= 1;
        c->cost = new_cost;
        c->successful += 1;
        const double f = 1.0 - (2.0 * rel - 1.0) * (2.0 * rel - 1.0) * (2.0 * rel - 1.0);
        c->radius = fmin(1e16, c->radius / fmax(1.0 / 3.0, f));
        c->decrease = 2.0;
        c->need_lin = 1;
        if (fabs(change) < 1e-6 * new_cost) c->done = 3;                  // function_tolerance
    } else {
        c->radius /= c->decrease;
        c->decrease *= 2.0;
        c->need_lin = 0;
        if (c->radius < 1e-32) c->done = 4;
    }
}

}  // namespace

extern "C" int sslam_ba_solve_host(sslam_ctx* ctx, int n_obs, const int32_t* pose_idx, const int32_t* point_idx,
                                   const double* uv, int n_poses, double* q, double* t,
                                   const unsigned char* pose_const, int n_points, double* X, const double* intr,
                                   int max_iters, double huber_delta, int points_const, double* summary) {
    SSLAM_REQUIRE(ctx != nullptr, "sslam_ba_solve_host: ctx is NULL");
    SSLAM_REQUIRE(n_obs > 0 && n_poses > 0 && n_points > 0, "sslam_ba_solve_host: empty problem");
    SSLAM_REQUIRE(pose_idx && point_idx && uv && q && t && pose_const && X && intr && summary,
                  "sslam_ba_solve_host: NULL argument");
    SSLAM_REQUIRE(max_iters >= 0 && huber_delta > 0.0, "sslam_ba_solve_host: bad max_iters / huber_delta");
    std::vector<int32_t> pose_slot(n_poses, -1), slot_pose;
    for (int p = 0; p < n_poses; ++p)
        if (!pose_const[p]) { pose_slot[p] = (int)slot_pose.size(); slot_pose.push_back(p); }
    const int Po = (int)slot_pose.size();
    SSLAM_REQUIRE(Po <= MAX_PO_BIG, "sslam_ba_solve_host: %d optimised poses, the device solver takes <= %d "
                  "(use the host Schur loop beyond)", Po, MAX_PO_BIG);
    // the Schur operands are dense [Po][Q][18] (a local window sees most of its points from most of its poses);
    // a global problem too large for that form is the host loop's
    SSLAM_REQUIRE((double)Po * (double)n_points * 288.0 <= 32.0 * 1024 * 1024 * 1024,
                  "sslam_ba_solve_host: %d optimised poses x %d points need %.1f GB of dense Schur operands "
                  "(limit 32 GB; use the host Schur loop)", Po, n_points, (double)Po * n_points * 288.0 / 1073741824.0);
    // CSR of observations by point and by optimised-pose slot (stable: observation order kept)
    std::vector<int32_t> obs_slot(n_obs), pt_ptr(n_points + 1, 0), pt_obs(n_obs), ps_ptr(Po + 1, 0), ps_obs;
    for (int i = 0; i < n_obs; ++i) {
        SSLAM_REQUIRE(pose_idx[i] >= 0 && pose_idx[i] < n_poses, "sslam_ba: pose_idx[%d]=%d out of range", i, pose_idx[i]);
        SSLAM_REQUIRE(point_idx[i] >= 0 && point_idx[i] < n_points, "sslam_ba: point_idx[%d]=%d out of range", i, point_idx[i]);
        obs_slot[i] = pose_slot[pose_idx[i]];
        pt_ptr[point_idx[i] + 1]++;
        if (obs_slot[i] >= 0) ps_ptr[obs_slot[i] + 1]++;
    }
    for (int j = 0; j < n_points; ++j) pt_ptr[j + 1] += pt_ptr[j];
    for (int p = 0; p < Po; ++p) ps_ptr[p + 1] += ps_ptr[p];
    ps_obs.resize(ps_ptr[Po]);
    {
        std::vector<int32_t> fp(pt_ptr.begin(), pt_ptr.end() - 1), fs(ps_ptr.begin(), ps_ptr.end() - 1);
        for (int i = 0; i < n_obs; ++i) {
            pt_obs[fp[point_idx[i]]++] = i;
            if (obs_slot[i] >= 0) ps_obs[fs[obs_slot[i]]++] = i;
        }
    }

    SSLAM_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t N = (size_t)n_obs, P = (size_t)n_poses, Q = (size_t)n_points, m = 6 * (size_t)Po;
    const int nb_obs = sslam::cdiv(n_obs, LM_T), nb_pt = sslam::cdiv(n_points, LM_T);
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = sslam::align_up(off + bytes + 8, 256); return o; };
    const size_t o_pi = carve(N * 4), o_xi = carve(N * 4), o_sl = carve(N * 4), o_uv = carve(N * 16);
    const size_t o_pp = carve((Q + 1) * 4), o_po = carve(N * 4), o_sp = carve((Po + 1) * 4), o_so = carve(ps_obs.size() * 4);
    const size_t o_slp = carve((size_t)Po * 4), o_psl = carve(P * 4), o_in = carve(32);
    const size_t o_q0 = carve(P * 32), o_q1 = carve(P * 32), o_t0 = carve(P * 24), o_t1 = carve(P * 24);
    const size_t o_X0 = carve(Q * 24), o_X1 = carve(Q * 24);
    const size_t o_rw = carve(N * 16), o_JX = carve(N * 48), o_Jp = carve(N * 96);
    const size_t o_V = carve(Q * 48), o_gX = carve(Q * 24), o_Vi = carve(Q * 48);
    const size_t o_Wd = carve((size_t)Po * Q * 144), o_Y = carve((size_t)Po * Q * 144);
    const size_t o_U = carve((size_t)Po * 288), o_gP = carve((size_t)Po * 48);
    const size_t o_S = carve(m * m * 8), o_rhs = carve(m * 8), o_dP = carve(m * 8), o_dX = carve(Q * 24);
    const bool big = Po > MAX_PO;
    const size_t o_Lb = carve(big ? m * m * 8 : 0);
    const size_t o_pc = carve((size_t)nb_obs * 8), o_pm = carve((size_t)nb_obs * 8), o_pb = carve((size_t)nb_obs * 4);
    const size_t o_ps = carve((size_t)nb_pt * 8), o_px = carve((size_t)nb_pt * 8), o_pg = carve((size_t)nb_pt * 8);
    const size_t o_ctrl = carve(sizeof(LMCtrl));
    if (off > ctx->ba_scratch_bytes) {
        if (ctx->ba_scratch) SSLAM_HIP_CHECK(hipFree(ctx->ba_scratch));
        ctx->ba_scratch = nullptr;
        ctx->ba_scratch_bytes = 0;
        SSLAM_HIP_CHECK(hipMalloc(&ctx->ba_scratch, off));
        ctx->ba_scratch_bytes = off;
    }
    char* b = (char*)ctx->ba_scratch;
    hipStream_t s = ctx->stream;
    auto up = [&](size_t o, const void* src, size_t bytes) {
        return bytes ? hipMemcpyAsync(b + o, src, bytes, hipMemcpyHostToDevice, s) : hipSuccess;
    };
    SSLAM_HIP_CHECK(up(o_pi, pose_idx, N * 4)); SSLAM_HIP_CHECK(up(o_xi, point_idx, N * 4));
    SSLAM_HIP_CHECK(up(o_sl, obs_slot.data(), N * 4)); SSLAM_HIP_CHECK(up(o_uv, uv, N * 16));
    SSLAM_HIP_CHECK(up(o_pp, pt_ptr.data(), (Q + 1) * 4)); SSLAM_HIP_CHECK(up(o_po, pt_obs.data(), N * 4));
    SSLAM_HIP_CHECK(up(o_sp, ps_ptr.data(), (Po + 1) * 4)); SSLAM_HIP_CHECK(up(o_so, ps_obs.data(), ps_obs.size() * 4));
    SSLAM_HIP_CHECK(up(o_slp, slot_pose.data(), (size_t)Po * 4)); SSLAM_HIP_CHECK(up(o_psl, pose_slot.data(), P * 4));
    SSLAM_HIP_CHECK(up(o_in, intr, 32));
    SSLAM_HIP_CHECK(up(o_q0, q, P * 32)); SSLAM_HIP_CHECK(up(o_t0, t, P * 24)); SSLAM_HIP_CHECK(up(o_X0, X, Q * 24));
    SSLAM_HIP_CHECK(hipMemsetAsync(b + o_ctrl, 0, sizeof(LMCtrl), s));

    LMArgs a{};
    a.n_obs = n_obs; a.P = n_poses; a.Q = n_points; a.Po = Po; a.points_const = points_const ? 1 : 0;
    a.nb_obs = nb_obs; a.nb_pt = nb_pt; a.delta = huber_delta;
    a.obs_pose = (const int32_t*)(b + o_pi); a.obs_point = (const int32_t*)(b + o_xi); a.obs_slot = (const int32_t*)(b + o_sl);
    a.uv = (const double*)(b + o_uv);
    a.pt_ptr = (const int32_t*)(b + o_pp); a.pt_obs = (const int32_t*)(b + o_po);
    a.ps_ptr = (const int32_t*)(b + o_sp); a.ps_obs = (const int32_t*)(b + o_so);
    a.slot_pose = (const int32_t*)(b + o_slp); a.pose_slot = (const int32_t*)(b + o_psl);
    a.intr = (const double*)(b + o_in);
    a.q[0] = (double*)(b + o_q0); a.q[1] = (double*)(b + o_q1); a.t[0] = (double*)(b + o_t0); a.t[1] = (double*)(b + o_t1);
    a.X[0] = (double*)(b + o_X0); a.X[1] = (double*)(b + o_X1);
    a.rw = (double*)(b + o_rw); a.JXw = (double*)(b + o_JX); a.Jpw = (double*)(b + o_Jp);
    a.V = (double*)(b + o_V); a.gX = (double*)(b + o_gX); a.Vinv = (double*)(b + o_Vi);
    a.Wd = (double*)(b + o_Wd); a.Y = (double*)(b + o_Y); a.U = (double*)(b + o_U); a.gP = (double*)(b + o_gP);
    a.Lbig = (double*)(b + o_Lb);
    a.S = (double*)(b + o_S); a.rhs = (double*)(b + o_rhs); a.dP = (double*)(b + o_dP); a.dX = (double*)(b + o_dX);
    a.pc = (double*)(b + o_pc); a.pm = (double*)(b + o_pm); a.pbad = (int*)(b + o_pb);
    a.pstep = (double*)(b + o_ps); a.px = (double*)(b + o_px); a.pgmax = (double*)(b + o_pg);
    a.ctrl = (LMCtrl*)(b + o_ctrl);

    (void)hipGetLastError();     // (a stale error of another library on this thread is not ours)
    hipLaunchKernelGGL(lm_eval_kernel<false>, dim3(nb_obs), dim3(LM_T), 0, s, a);
    hipLaunchKernelGGL(lm_init_kernel, dim3(1), dim3(LM_T), 0, s, a);
    for (int it = 0; it < max_iters; ++it) {
        hipLaunchKernelGGL(lm_linearise_kernel, dim3(nb_obs), dim3(LM_T), 0, s, a);
        hipLaunchKernelGGL(lm_point_kernel, dim3(nb_pt), dim3(LM_T), 0, s, a);
        if (Po) {
            hipLaunchKernelGGL(lm_pose_kernel, dim3(Po), dim3(LM_RT), 0, s, a);
            hipLaunchKernelGGL(lm_schur_kernel, dim3(Po, Po + 1), dim3(LM_RT), 0, s, a);
        }
        if (big) hipLaunchKernelGGL(lm_solve_big_kernel, dim3(1), dim3(LM_BT), 0, s, a);
        else hipLaunchKernelGGL(lm_solve_kernel, dim3(1), dim3(LM_T), 0, s, a);
        hipLaunchKernelGGL(lm_update_points_kernel, dim3(nb_pt), dim3(LM_T), 0, s, a);
        hipLaunchKernelGGL(lm_update_poses_kernel, dim3(1), dim3(64), 0, s, a);
        hipLaunchKernelGGL(lm_eval_kernel<true>, dim3(nb_obs), dim3(LM_T), 0, s, a);
        hipLaunchKernelGGL(lm_decide_kernel, dim3(1), dim3(LM_T), 0, s, a);
    }
    SSLAM_HIP_CHECK(hipGetLastError());
    LMCtrl h{};
    SSLAM_HIP_CHECK(hipMemcpyAsync(&h, b + o_ctrl, sizeof(LMCtrl), hipMemcpyDeviceToHost, s));
    SSLAM_HIP_CHECK(hipStreamSynchronize(s));
    const size_t oq = h.cur ? o_q1 : o_q0, ot = h.cur ? o_t1 : o_t0, oX = h.cur ? o_X1 : o_X0;
    SSLAM_HIP_CHECK(hipMemcpyAsync(q, b + oq, P * 32, hipMemcpyDeviceToHost, s));
    SSLAM_HIP_CHECK(hipMemcpyAsync(t, b + ot, P * 24, hipMemcpyDeviceToHost, s));
    SSLAM_HIP_CHECK(hipMemcpyAsync(X, b + oX, Q * 24, hipMemcpyDeviceToHost, s));
    SSLAM_HIP_CHECK(hipStreamSynchronize(s));
    summary[0] = h.iterations; summary[1] = h.successful; summary[2] = h.initial_cost; summary[3] = h.cost;
    summary[4] = h.done;       // 0 = max iterations reached
    summary[5] = h.radius; summary[6] = Po; summary[7] = 0.0;
    return 0;
}
