// ba_math.hpp - the per-observation arithmetic of local BA (fp64), shared by the
// residual/Jacobian kernel (ba_kernels.hip) and the device-resident LM (ba_lm.hip).
//
// COLMAP 3.10 ReprojErrorCostFunction<PinholeCameraModel> as Ceres differentiates it
// (reference call site slam/core/ba_utils.py:56-68): Eigen's un-normalised quaternion
// sandwich, pinhole projection, derivative w.r.t. the four AMBIENT quaternion
// coordinates (x, y, z, w), translation and point.
#pragma once
#include <hip/hip_runtime.h>

namespace sslam {

struct BAObsIn {
    double ax, ay, az, w;      // quaternion xyzw
    double tx, ty, tz;
    double Xx, Xy, Xz;
    double fx, fy, cx, cy;
    double u, v;
};

// r[2]; when JAC also jq[2x4], jt[2x3], jx[2x3] (row-major)
template <bool JAC>
__device__ __forceinline__ void ba_reproj(const BAObsIn& in, double& r0, double& r1, double* __restrict__ jq,
                                          double* __restrict__ jt, double* __restrict__ jx) {
    const double ax = in.ax, ay = in.ay, az = in.az, w = in.w;
    const double tx = in.tx, ty = in.ty, tz = in.tz;
    const double Xx = in.Xx, Xy = in.Xy, Xz = in.Xz;
    const double fx = in.fx, fy = in.fy, cx = in.cx, cy = in.cy;
    const double2 m = make_double2(in.u, in.v);

    // Eigen Quaternion::_transformVector (no normalisation): uv = 2 a x X;
    // p = X + w uv + a x uv + t
    const double ux = 2.0 * (ay * Xz - az * Xy);
    const double uy = 2.0 * (az * Xx - ax * Xz);
    const double uz = 2.0 * (ax * Xy - ay * Xx);
    const double px = Xx + w * ux + (ay * uz - az * uy) + tx;
    const double py = Xy + w * uy + (az * ux - ax * uz) + ty;
    const double pz = Xz + w * uz + (ax * uy - ay * ux) + tz;
    const double d = 1.0 / pz;
    r0 = fx * px * d + cx - m.x;
    r1 = fy * py * d + cy - m.y;

    if constexpr (JAC) {
        // dr/dp (2x3, sparse)
        const double a00 = fx * d, a02 = -fx * px * d * d;
        const double a11 = fy * d, a12 = -fy * py * d * d;
        jt[0] = a00; jt[1] = 0.0; jt[2] = a02;
        jt[3] = 0.0; jt[4] = a11; jt[5] = a12;

        // dp/dX = I + 2w[a]x + 2(a a^T - |a|^2 I)
        const double a2 = ax * ax + ay * ay + az * az;
        double R[9];
        R[0] = 1.0 + 2.0 * (ax * ax - a2);
        R[1] = 2.0 * (-w * az + ax * ay);
        R[2] = 2.0 * (w * ay + ax * az);
        R[3] = 2.0 * (w * az + ay * ax);
        R[4] = 1.0 + 2.0 * (ay * ay - a2);
        R[5] = 2.0 * (-w * ax + ay * az);
        R[6] = 2.0 * (-w * ay + az * ax);
        R[7] = 2.0 * (w * ax + az * ay);
        R[8] = 1.0 + 2.0 * (az * az - a2);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            jx[c] = a00 * R[c] + a02 * R[6 + c];
            jx[3 + c] = a11 * R[3 + c] + a12 * R[6 + c];
        }

        // dp/da = -2w[X]x + 2((a.X) I + a X^T - 2 X a^T);  dp/dw = 2 a x X = (ux,uy,uz)
        const double aX = ax * Xx + ay * Xy + az * Xz;
        double D[12];  // 3x4 row-major, columns x,y,z,w
        D[0] = 2.0 * (aX + ax * Xx - 2.0 * Xx * ax);
        D[1] = -2.0 * w * (-Xz) + 2.0 * (ax * Xy - 2.0 * Xx * ay);
        D[2] = -2.0 * w * (Xy) + 2.0 * (ax * Xz - 2.0 * Xx * az);
        D[3] = ux;
        D[4] = -2.0 * w * (Xz) + 2.0 * (ay * Xx - 2.0 * Xy * ax);
        D[5] = 2.0 * (aX + ay * Xy - 2.0 * Xy * ay);
        D[6] = -2.0 * w * (-Xx) + 2.0 * (ay * Xz - 2.0 * Xy * az);
        D[7] = uy;
        D[8] = -2.0 * w * (-Xy) + 2.0 * (az * Xx - 2.0 * Xz * ax);
        D[9] = -2.0 * w * (Xx) + 2.0 * (az * Xy - 2.0 * Xz * ay);
        D[10] = 2.0 * (aX + az * Xz - 2.0 * Xz * az);
        D[11] = uz;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            jq[c] = a00 * D[c] + a02 * D[8 + c];
            jq[4 + c] = a11 * D[4 + c] + a12 * D[8 + c];
        }
    }
}

}  // namespace sslam
