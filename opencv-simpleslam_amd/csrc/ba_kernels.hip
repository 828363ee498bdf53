// ba_kernels.hip - batched reprojection residual + Jacobian for local BA (fp64).
//
// Replaces the inner loop of `pyceres.solve` over the residual blocks that the
// reference adds at slam/core/ba_utils.py:56-68 (`cost_functions.ReprojErrorCost(
// CameraModelId.PINHOLE, uv)`, parameter blocks [quat xyzw, trans, point, intr]).
//
// Roofline: HBM.  Algorithmic bytes per observation = 2 i32 + 2 f64 read
// (24 B; q/t/X gathers hit L2: <= 15 poses and each point is re-used by its
// 2-10 observations) + 22 f64 written (176 B); SURVEY.md section 8(d) prices it at
// 312 B/obs counting the 14 gathered f64 too.  ~150 FLOP/obs -> 0.5 FLOP/B.
//
// Layout: one thread per observation computes the 22 outputs in registers; the
// four output arrays are AoS ([n][2], [n][8], [n][6], [n][6]) as the host LM
// consumes them, so each is staged through a 16 KB LDS slab and written back as
// fully coalesced 16-byte stores (a lane-strided direct store would touch
// 64 B-strided 16 B pieces).
#include "common.hpp"
#include "ba_math.hpp"

namespace {

constexpr int BA_THREADS = 256;

template <int W>
__device__ __forceinline__ void store_slab(double* __restrict__ dst, const double* __restrict__ slab,
                                           long long base_elem, long long total_elems, int n_in_block) {
    // slab holds n_in_block * W doubles; write as double2 (16 B) where aligned.
    const int n = n_in_block * W;
    double* out = dst + base_elem;
    for (int e = threadIdx.x * 2; e < n; e += BA_THREADS * 2) {
        if (e + 1 < n) {
            *reinterpret_cast<double2*>(out + e) = make_double2(slab[e], slab[e + 1]);
        } else {
            out[e] = slab[e];
        }
    }
    (void)total_elems;
}

__global__ __launch_bounds__(BA_THREADS) void ba_residual_jacobian_kernel(
    int n_obs, const int32_t* __restrict__ pose_idx, const int32_t* __restrict__ point_idx,
    const double* __restrict__ uv, const double* __restrict__ q, const double* __restrict__ t,
    const double* __restrict__ X, const double* __restrict__ intr, double* __restrict__ r_out,
    double* __restrict__ Jq_out, double* __restrict__ Jt_out, double* __restrict__ JX_out) {
    __shared__ __attribute__((aligned(16))) double slab[BA_THREADS * 8];

    const long long block_first = (long long)blockIdx.x * BA_THREADS;
    const int n_in_block = (int)min((long long)BA_THREADS, (long long)n_obs - block_first);
    const int i = (int)(block_first + threadIdx.x);
    const bool active = threadIdx.x < n_in_block;

    double r0 = 0, r1 = 0;
    double jq[8], jt[6], jx[6];
#pragma unroll
    for (int k = 0; k < 8; ++k) jq[k] = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) { jt[k] = 0; jx[k] = 0; }

    if (active) {
        const int pi = pose_idx[i];
        const int xi = point_idx[i];
        sslam::BAObsIn in;
        in.ax = q[4 * pi + 0]; in.ay = q[4 * pi + 1]; in.az = q[4 * pi + 2]; in.w = q[4 * pi + 3];
        in.tx = t[3 * pi + 0]; in.ty = t[3 * pi + 1]; in.tz = t[3 * pi + 2];
        in.Xx = X[3 * xi + 0]; in.Xy = X[3 * xi + 1]; in.Xz = X[3 * xi + 2];
        in.fx = intr[0]; in.fy = intr[1]; in.cx = intr[2]; in.cy = intr[3];
        const double2 m = *reinterpret_cast<const double2*>(uv + 2ll * i);
        in.u = m.x; in.v = m.y;
        if (Jq_out != nullptr) sslam::ba_reproj<true>(in, r0, r1, jq, jt, jx);
        else sslam::ba_reproj<false>(in, r0, r1, jq, jt, jx);
    }

    // ---- residual
    slab[threadIdx.x * 2 + 0] = r0;
    slab[threadIdx.x * 2 + 1] = r1;
    __syncthreads();
    store_slab<2>(r_out, slab, block_first * 2, 0, n_in_block);
    if (Jq_out == nullptr) return;  // uniform: residual-only evaluation
    __syncthreads();
    // ---- Jq
#pragma unroll
    for (int k = 0; k < 8; ++k) slab[threadIdx.x * 8 + k] = jq[k];
    __syncthreads();
    store_slab<8>(Jq_out, slab, block_first * 8, 0, n_in_block);
    __syncthreads();
    // ---- Jt
#pragma unroll
    for (int k = 0; k < 6; ++k) slab[threadIdx.x * 6 + k] = jt[k];
    __syncthreads();
    store_slab<6>(Jt_out, slab, block_first * 6, 0, n_in_block);
    __syncthreads();
    // ---- JX
#pragma unroll
    for (int k = 0; k < 6; ++k) slab[threadIdx.x * 6 + k] = jx[k];
    __syncthreads();
    store_slab<6>(JX_out, slab, block_first * 6, 0, n_in_block);
}

int launch_ba(sslam_ctx* ctx, int n_obs, const int32_t* pose_idx, const int32_t* point_idx,
              const double* uv, const double* q, const double* t, const double* X,
              const double* intr, double* r, double* Jq, double* Jt, double* JX) {
    if (n_obs == 0) return 0;
    (void)hipGetLastError();     // (a stale error of another library on this thread is not ours)
    const int blocks = sslam::cdiv(n_obs, BA_THREADS);
    hipLaunchKernelGGL(ba_residual_jacobian_kernel, dim3(blocks), dim3(BA_THREADS), 0, ctx->stream,
                       n_obs, pose_idx, point_idx, uv, q, t, X, intr, r, Jq, Jt, JX);
    SSLAM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" {

int sslam_ba_residual_jacobian_dev(sslam_ctx* ctx, int n_obs, const int32_t* pose_idx,
                                   const int32_t* point_idx, const double* uv, int n_poses,
                                   const double* q, const double* t, int n_points,
                                   const double* X, const double* intr, double* r, double* Jq,
                                   double* Jt, double* JX) {
    SSLAM_REQUIRE(ctx != nullptr, "sslam_ba_residual_jacobian_dev: ctx is NULL");
    SSLAM_REQUIRE(n_obs >= 0 && n_poses >= 0 && n_points >= 0, "sslam_ba: negative size");
    SSLAM_REQUIRE(n_obs == 0 || (pose_idx && point_idx && uv && q && t && X && intr && r),
                  "sslam_ba_residual_jacobian_dev: NULL input");
    SSLAM_REQUIRE((Jq == nullptr) == (Jt == nullptr) && (Jq == nullptr) == (JX == nullptr),
                  "sslam_ba: Jq/Jt/JX must be all NULL or all non-NULL");
    return launch_ba(ctx, n_obs, pose_idx, point_idx, uv, q, t, X, intr, r, Jq, Jt, JX);
}

int sslam_ba_residual_jacobian_host(sslam_ctx* ctx, int n_obs, const int32_t* pose_idx,
                                    const int32_t* point_idx, const double* uv, int n_poses,
                                    const double* q, const double* t, int n_points,
                                    const double* X, const double* intr, double* r, double* Jq,
                                    double* Jt, double* JX) {
    SSLAM_REQUIRE(ctx != nullptr, "sslam_ba_residual_jacobian_host: ctx is NULL");
    SSLAM_REQUIRE(n_obs >= 0 && n_poses >= 0 && n_points >= 0, "sslam_ba: negative size");
    if (n_obs == 0) return 0;
    SSLAM_REQUIRE(pose_idx && point_idx && uv && q && t && X && intr && r,
                  "sslam_ba_residual_jacobian_host: NULL input");
    SSLAM_REQUIRE((Jq == nullptr) == (Jt == nullptr) && (Jq == nullptr) == (JX == nullptr),
                  "sslam_ba: Jq/Jt/JX must be all NULL or all non-NULL");
    // index validation on the host: an out-of-range gather would fault the GPU
    for (int i = 0; i < n_obs; ++i) {
        SSLAM_REQUIRE(pose_idx[i] >= 0 && pose_idx[i] < n_poses, "sslam_ba: pose_idx[%d]=%d out of range",
                      i, pose_idx[i]);
        SSLAM_REQUIRE(point_idx[i] >= 0 && point_idx[i] < n_points,
                      "sslam_ba: point_idx[%d]=%d out of range", i, point_idx[i]);
    }
    SSLAM_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t N = (size_t)n_obs;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = sslam::align_up(off + bytes, 256); return o; };
    const size_t o_pi = carve(N * 4), o_xi = carve(N * 4), o_uv = carve(N * 16);
    const size_t o_q = carve((size_t)n_poses * 32), o_t = carve((size_t)n_poses * 24);
    const size_t o_X = carve((size_t)n_points * 24), o_in = carve(32);
    const size_t o_r = carve(N * 16), o_Jq = carve(N * 64), o_Jt = carve(N * 48), o_JX = carve(N * 48);
    if (off > ctx->ba_scratch_bytes) {
        if (ctx->ba_scratch) SSLAM_HIP_CHECK(hipFree(ctx->ba_scratch));
        ctx->ba_scratch = nullptr;
        ctx->ba_scratch_bytes = 0;
        SSLAM_HIP_CHECK(hipMalloc(&ctx->ba_scratch, off));
        ctx->ba_scratch_bytes = off;
    }
    char* b = (char*)ctx->ba_scratch;
    hipStream_t s = ctx->stream;
    SSLAM_HIP_CHECK(hipMemcpyAsync(b + o_pi, pose_idx, N * 4, hipMemcpyHostToDevice, s));
    SSLAM_HIP_CHECK(hipMemcpyAsync(b + o_xi, point_idx, N * 4, hipMemcpyHostToDevice, s));
    SSLAM_HIP_CHECK(hipMemcpyAsync(b + o_uv, uv, N * 16, hipMemcpyHostToDevice, s));
    SSLAM_HIP_CHECK(hipMemcpyAsync(b + o_q, q, (size_t)n_poses * 32, hipMemcpyHostToDevice, s));
    SSLAM_HIP_CHECK(hipMemcpyAsync(b + o_t, t, (size_t)n_poses * 24, hipMemcpyHostToDevice, s));
    SSLAM_HIP_CHECK(hipMemcpyAsync(b + o_X, X, (size_t)n_points * 24, hipMemcpyHostToDevice, s));
    SSLAM_HIP_CHECK(hipMemcpyAsync(b + o_in, intr, 32, hipMemcpyHostToDevice, s));
    const bool jac = Jq != nullptr;
    int rc = launch_ba(ctx, n_obs, (const int32_t*)(b + o_pi), (const int32_t*)(b + o_xi),
                       (const double*)(b + o_uv), (const double*)(b + o_q), (const double*)(b + o_t),
                       (const double*)(b + o_X), (const double*)(b + o_in), (double*)(b + o_r),
                       jac ? (double*)(b + o_Jq) : nullptr, jac ? (double*)(b + o_Jt) : nullptr,
                       jac ? (double*)(b + o_JX) : nullptr);
    if (rc) return rc;
    SSLAM_HIP_CHECK(hipMemcpyAsync(r, b + o_r, N * 16, hipMemcpyDeviceToHost, s));
    if (jac) {
        SSLAM_HIP_CHECK(hipMemcpyAsync(Jq, b + o_Jq, N * 64, hipMemcpyDeviceToHost, s));
        SSLAM_HIP_CHECK(hipMemcpyAsync(Jt, b + o_Jt, N * 48, hipMemcpyDeviceToHost, s));
        SSLAM_HIP_CHECK(hipMemcpyAsync(JX, b + o_JX, N * 48, hipMemcpyDeviceToHost, s));
    }
    SSLAM_HIP_CHECK(hipStreamSynchronize(s));
    return 0;
}

}  // extern "C"
