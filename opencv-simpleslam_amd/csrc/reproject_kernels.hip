// reproject_kernels.hip - projection + radius search + descriptor matching of map points against
// the current frame's keypoints (2D-3D association for PnP tracking).
//
// Replaces the per-point Python loop of `reproject_and_match_2d3d`
// (slam/core/pnp_utils.py:224-304: `_project_points` :127-141, cKDTree ball query :238/:265,
// `_best_mp_distance_to_cur_desc` :107-120, greedy `used_kps` assignment :260-286) for float
// descriptors - SURVEY.md section 8(f) rank 3.
//
// The loop is sequential only through `used_kps`; everything it consumes is independent per map
// point: (1) project (float64, float32 pixels like the reference), (2) per candidate point the
// keypoints within `radius_px` (squared distance in float64, as cKDTree compares) and for each of
// them the minimum L2 distance to the point's last <= 6 observation descriptors (float32), one
// wave per point; (3) one lane then replays the greedy pass over the stored (keypoint, distance)
// lists in map order.  HBM-bound: Q x 6 x 512 B of observation descriptors read once.
#include "common.hpp"

namespace {

constexpr int RP_MAXC = 128;      // keypoints kept per map point (within radius)
constexpr int RP_DIM = 128;

struct RPArgs {
    int Q, N, img_w, img_h;
    double radius2, thr;
    const double* pts;            // [Q][3]
    const int32_t* obs_cnt;       // [Q] descriptors among the last six observations (0: point is skipped)
    const float* obs_desc;        // [Q][6][128], the valid ones first
    const double* K;              // [9]
    const double* Tcw;            // [16]
    const float* kp;              // [N][2]
    const float* des;             // [N][128]
    float* uv;                    // [Q][2]
    int32_t* cand_n;              // [Q]  (-1: not a candidate)
    int32_t* cand_kp;             // [Q][RP_MAXC]
    float* cand_d;                // [Q][RP_MAXC]
    int32_t* kp_of_point;         // [Q] out
    int32_t* info;                // [0] matches, [1] overflow flag, [2] candidates
};

// one wave per map point: project, collect the keypoints in range (ascending index), score them
__global__ __launch_bounds__(256) void rp_pairs_kernel(RPArgs a) {
    __shared__ int list[4][RP_MAXC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave;
    if (q >= a.Q) return;
    // `_project_points`: Xc = p R^T + t (float64), uv = (K (Xc / z))[:2] as float32, -1 where z <= 1e-8
    const double px = a.pts[3 * q], py = a.pts[3 * q + 1], pz = a.pts[3 * q + 2];
    const double* T = a.Tcw;
    const double xc = (px * T[0] + py * T[1] + pz * T[2]) + T[3];
    const double yc = (px * T[4] + py * T[5] + pz * T[6]) + T[7];
    const double zc = (px * T[8] + py * T[9] + pz * T[10]) + T[11];
    float u = -1.0f, v = -1.0f;
    if (zc > 1e-8) {
        const double xn = xc / zc, yn = yc / zc, zn = zc / zc;
        u = (float)((a.K[0] * xn + a.K[1] * yn) + a.K[2] * zn);
        v = (float)((a.K[3] * xn + a.K[4] * yn) + a.K[5] * zn);
    }
    const bool cand = zc > 0.0 && u >= 0.0f && u < (float)a.img_w && v >= 0.0f && v < (float)a.img_h && a.obs_cnt[q] > 0;
    if (lane == 0) { a.uv[2 * q] = u; a.uv[2 * q + 1] = v; a.kp_of_point[q] = -1; }
    if (!cand) { if (lane == 0) a.cand_n[q] = -1; return; }
    // radius search, ascending keypoint index (ballot compaction)
    int n = 0;
    for (int base = 0; base < a.N; base += 64) {
        const int i = base + lane;
        bool in = false;
        if (i < a.N) {
            const double dx = (double)a.kp[2 * i] - (double)u, dy = (double)a.kp[2 * i + 1] - (double)v;
            in = dx * dx + dy * dy <= a.radius2;
        }
        const unsigned long long m = __ballot(in);
        if (in) {
            const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < RP_MAXC) list[wave][pos] = i;
        }
        n += __popcll(m);
    }
    if (n > RP_MAXC) { if (lane == 0) atomicOr(&a.info[1], 1); n = RP_MAXC; }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // min over the stored observation descriptors of |obs - des|_2 (float32), lane = 2 dimensions
    const int cnt = a.obs_cnt[q];
    float2 o[6];
    for (int j = 0; j < 6; ++j)
        o[j] = j < cnt ? *reinterpret_cast<const float2*>(a.obs_desc + ((size_t)q * 6 + j) * RP_DIM + 2 * lane) : make_float2(0.0f, 0.0f);
    for (int c = 0; c < n; ++c) {
        const int i = list[wave][c];
        const float2 d = *reinterpret_cast<const float2*>(a.des + (size_t)i * RP_DIM + 2 * lane);
        float best = INFINITY;
        for (int j = 0; j < cnt; ++j) {
            const float ex = o[j].x - d.x, ey = o[j].y - d.y;
            float s = ex * ex + ey * ey;
            for (int sft = 32; sft > 0; sft >>= 1) s += __shfl_xor(s, sft);
            best = fminf(best, sqrtf(s));
        }
        if (lane == 0) { a.cand_kp[(size_t)q * RP_MAXC + c] = i; a.cand_d[(size_t)q * RP_MAXC + c] = best; }
    }
    if (lane == 0) a.cand_n[q] = n;
}

// the greedy pass, in map order (one lane; `used` bitmap in LDS)
__global__ __launch_bounds__(64) void rp_assign_kernel(RPArgs a) {
    extern __shared__ unsigned used[];
    for (int i = threadIdx.x; i < (a.N + 31) / 32; i += 64) used[i] = 0u;
    __syncthreads();
    if (threadIdx.x != 0) return;
    int matches = 0, cands = 0;
    for (int q = 0; q < a.Q; ++q) {
        const int n = a.cand_n[q];
        if (n < 0) continue;
        ++cands;
        int best_i = -1;
        float best_d = 1e9f;
        for (int c = 0; c < n; ++c) {
            const int i = a.cand_kp[(size_t)q * RP_MAXC + c];
            if (used[i >> 5] & (1u << (i & 31))) continue;
            const float d = a.cand_d[(size_t)q * RP_MAXC + c];
            if (d < best_d) { best_d = d; best_i = i; }
        }
        if (best_i < 0 || (double)best_d > a.thr) continue;
        used[best_i >> 5] |= 1u << (best_i & 31);
        a.kp_of_point[q] = best_i;
        ++matches;
    }
    a.info[0] = matches;
    a.info[2] = cands;
}

}  // namespace

namespace {

// scratch of one association call, carved from the context's scratch slab
struct RPScratch { char* base; size_t K, T, uv, cn, ck, cd, out, info, pts, cnt, od, kp, des; };

int rp_scratch(sslam_ctx* ctx, size_t Q, size_t N, bool host_inputs, RPScratch& sc) {
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = sslam::align_up(off + bytes + 8, 256); return o; };
    sc.K = carve(72); sc.T = carve(128); sc.uv = carve(Q * 8); sc.cn = carve(Q * 4);
    sc.ck = carve(Q * RP_MAXC * 4); sc.cd = carve(Q * RP_MAXC * 4); sc.out = carve(Q * 4); sc.info = carve(16);
    if (host_inputs) {
        sc.pts = carve(Q * 24); sc.cnt = carve(Q * 4); sc.od = carve(Q * 6 * RP_DIM * 4);
        sc.kp = carve(N * 8); sc.des = carve(N * RP_DIM * 4);
    }
    if (off > ctx->ba_scratch_bytes) {
        (void)hipStreamSynchronize(ctx->stream);          // earlier enqueued work may still use the old slab
        if (ctx->ba_scratch) SSLAM_HIP_CHECK(hipFree(ctx->ba_scratch));
        ctx->ba_scratch = nullptr;
        ctx->ba_scratch_bytes = 0;
        SSLAM_HIP_CHECK(hipMalloc(&ctx->ba_scratch, off));
        ctx->ba_scratch_bytes = off;
    }
    sc.base = (char*)ctx->ba_scratch;
    return 0;
}

// enqueue the two kernels on device-resident inputs; K9 / Tcw16 are host values
int rp_enqueue(sslam_ctx* ctx, const RPScratch& sc, int n_points, const double* pts_d, const int32_t* cnt_d,
               const float* desc_d, const double* K9, const double* Tcw16, int n_kp, const float* kp_d, const float* des_d,
               int img_w, int img_h, double radius_px, double max_dist, int32_t* out_d, float* uv_d, int32_t* info_d) {
    hipStream_t s = ctx->stream;
    char* b = sc.base;
    (void)hipGetLastError();     // (a stale error of another library on this thread is not ours)
    SSLAM_HIP_CHECK(hipMemcpyAsync(b + sc.K, K9, 72, hipMemcpyHostToDevice, s));
    SSLAM_HIP_CHECK(hipMemcpyAsync(b + sc.T, Tcw16, 128, hipMemcpyHostToDevice, s));
    SSLAM_HIP_CHECK(hipMemsetAsync(info_d, 0, 16, s));
    RPArgs a{};
    a.Q = n_points; a.N = n_kp; a.img_w = img_w; a.img_h = img_h; a.radius2 = radius_px * radius_px; a.thr = max_dist;
    a.pts = pts_d; a.obs_cnt = cnt_d; a.obs_desc = desc_d;
    a.K = (const double*)(b + sc.K); a.Tcw = (const double*)(b + sc.T); a.kp = kp_d; a.des = des_d;
    a.uv = uv_d ? uv_d : (float*)(b + sc.uv); a.cand_n = (int32_t*)(b + sc.cn);
    a.cand_kp = (int32_t*)(b + sc.ck); a.cand_d = (float*)(b + sc.cd); a.kp_of_point = out_d; a.info = info_d;
    hipLaunchKernelGGL(rp_pairs_kernel, dim3(sslam::cdiv(n_points, 4)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(rp_assign_kernel, dim3(1), dim3(64), (((size_t)n_kp + 31) / 32) * 4, s, a);
    SSLAM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

/* Device-resident variant: the map arrays (an incrementally maintained SoA map keeps them on the GPU)
 * and the current frame's keypoints / descriptors (straight from sslam_aliked_extract_dev) are device
 * pointers; K9 / Tcw16 are host values.  Enqueue only.  kp_of_point[n_points], uv_out[n_points*2]
 * (may be NULL) and info_out[4] = {matches, overflow flag, candidate points, 0} are device buffers. */
extern "C" int sslam_reproject_match_dev(sslam_ctx* ctx, int n_points, const double* pts3d, const int32_t* obs_cnt,
                                         const float* obs_desc, const double* K9, const double* Tcw16, int n_kp,
                                         const float* kp_xy, const float* des, int img_w, int img_h, double radius_px,
                                         double max_dist, int32_t* kp_of_point, float* uv_out, int32_t* info_out) {
    SSLAM_REQUIRE(ctx != nullptr, "sslam_reproject_match_dev: ctx is NULL");
    SSLAM_REQUIRE(n_points > 0 && n_kp > 0, "sslam_reproject_match_dev: empty input (the caller returns early)");
    SSLAM_REQUIRE(pts3d && obs_cnt && obs_desc && K9 && Tcw16 && kp_xy && des && kp_of_point && info_out,
                  "sslam_reproject_match_dev: NULL argument");
    SSLAM_REQUIRE(radius_px >= 0.0 && img_w > 0 && img_h > 0, "sslam_reproject_match_dev: bad radius / image size");
    SSLAM_HIP_CHECK(hipSetDevice(ctx->device));
    RPScratch sc{};
    if (int rc = rp_scratch(ctx, (size_t)n_points, (size_t)n_kp, false, sc)) return rc;
    return rp_enqueue(ctx, sc, n_points, pts3d, obs_cnt, obs_desc, K9, Tcw16, n_kp, kp_xy, des, img_w, img_h, radius_px,
                      max_dist, kp_of_point, uv_out, info_out);
}

extern "C" int sslam_reproject_match_host(sslam_ctx* ctx, int n_points, const double* pts3d, const int32_t* obs_cnt,
                                          const float* obs_desc, const double* K9, const double* Tcw16, int n_kp,
                                          const float* kp_xy, const float* des, int img_w, int img_h, double radius_px,
                                          double max_dist, int32_t* kp_of_point, float* uv_out, int32_t* info_out) {
    SSLAM_REQUIRE(ctx != nullptr, "sslam_reproject_match_host: ctx is NULL");
    SSLAM_REQUIRE(n_points > 0 && n_kp > 0, "sslam_reproject_match_host: empty input (the caller returns early)");
    SSLAM_REQUIRE(pts3d && obs_cnt && obs_desc && K9 && Tcw16 && kp_xy && des && kp_of_point,
                  "sslam_reproject_match_host: NULL argument");
    SSLAM_REQUIRE(radius_px >= 0.0 && img_w > 0 && img_h > 0, "sslam_reproject_match_host: bad radius / image size");
    for (int q = 0; q < n_points; ++q)
        SSLAM_REQUIRE(obs_cnt[q] >= 0 && obs_cnt[q] <= 6, "sslam_reproject_match_host: obs_cnt[%d]=%d not in [0,6]", q, obs_cnt[q]);
    SSLAM_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t Q = (size_t)n_points, N = (size_t)n_kp;
    RPScratch sc{};
    if (int rc = rp_scratch(ctx, Q, N, true, sc)) return rc;
    char* b = sc.base;
    hipStream_t s = ctx->stream;
    auto up = [&](size_t o, const void* src, size_t bytes) { return hipMemcpyAsync(b + o, src, bytes, hipMemcpyHostToDevice, s); };
    SSLAM_HIP_CHECK(up(sc.pts, pts3d, Q * 24)); SSLAM_HIP_CHECK(up(sc.cnt, obs_cnt, Q * 4));
    SSLAM_HIP_CHECK(up(sc.od, obs_desc, Q * 6 * RP_DIM * 4));
    SSLAM_HIP_CHECK(up(sc.kp, kp_xy, N * 8)); SSLAM_HIP_CHECK(up(sc.des, des, N * RP_DIM * 4));
    if (int rc = rp_enqueue(ctx, sc, n_points, (const double*)(b + sc.pts), (const int32_t*)(b + sc.cnt),
                            (const float*)(b + sc.od), K9, Tcw16, n_kp, (const float*)(b + sc.kp), (const float*)(b + sc.des),
                            img_w, img_h, radius_px, max_dist, (int32_t*)(b + sc.out), (float*)(b + sc.uv),
                            (int32_t*)(b + sc.info)))
        return rc;
    int32_t info[4] = {0, 0, 0, 0};
    SSLAM_HIP_CHECK(hipMemcpyAsync(kp_of_point, b + sc.out, Q * 4, hipMemcpyDeviceToHost, s));
    if (uv_out) SSLAM_HIP_CHECK(hipMemcpyAsync(uv_out, b + sc.uv, Q * 8, hipMemcpyDeviceToHost, s));
    SSLAM_HIP_CHECK(hipMemcpyAsync(info, b + sc.info, 16, hipMemcpyDeviceToHost, s));
    SSLAM_HIP_CHECK(hipStreamSynchronize(s));
    SSLAM_REQUIRE(info[1] == 0, "sslam_reproject_match_host: more than %d keypoints within %.1f px of one projection",
                  RP_MAXC, radius_px);
    if (info_out) { info_out[0] = info[0]; info_out[1] = info[2]; }
    return 0;
}
