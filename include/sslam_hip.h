/*
 * sslam_hip.h - C-ABI of libsslam_hip.so, the MI355X (gfx950) backend for the
 * ALIKED + LightGlue + local-BA hot path of KlrShaK/opencv-SimpleSLAM.
 *
 * The reference has no FFI of its own: its hot path is Python calling
 * third-party wheels.  The drop-in boundary is the set
 * of Python names `slam/monocular/main_revamped.py` imports; this C-ABI sits
 * directly behind those names and is what a ctypes binding of that path binds
 * (INTEGRATION.md shows the binding).  Each entry point cites the reference
 * call it replaces as  file:line  relative to the reference root.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / C++ types in any signature
 *   - every function returns 0 on success, non-zero on error;
 *     sslam_last_error() returns the message of the calling thread's last error
 *   - "_host" entry points take host pointers and block until results are in
 *     the caller's buffers; "_dev" entry points take device pointers, enqueue
 *     on the context's HIP stream and return without synchronising
 *   - outputs are caller-allocated
 *   - one context per process per GPU; a context is not thread-safe
 */
#ifndef SSLAM_HIP_H
#define SSLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sslam_ctx sslam_ctx;          /* device + stream + timers            */
typedef struct sslam_aliked sslam_aliked;    /* ALIKED-n16 extractor instance       */
typedef struct sslam_lightglue sslam_lightglue; /* LightGlue(features='aliked') inst.*/

/* ------------------------------------------------------------------ library */
int sslam_abi_version(void);
const char* sslam_last_error(void);
int sslam_device_count(int* n_out);

/* ------------------------------------------------------------------ context
 * Replaces the implicit `torch.cuda` device selection at
 * slam/core/features_utils.py:24 and :222.  `stream` may be NULL (the context
 * then owns a new non-blocking HIP stream) or an existing hipStream_t. */
int sslam_ctx_create(int device, void* stream, sslam_ctx** out);
int sslam_ctx_destroy(sslam_ctx* ctx);
int sslam_ctx_sync(sslam_ctx* ctx);
void* sslam_ctx_stream(sslam_ctx* ctx);
/* HIP-event timer on the context's stream (bench.py roofline measurement). */
int sslam_timer_start(sslam_ctx* ctx);
int sslam_timer_stop(sslam_ctx* ctx, float* elapsed_ms_out);
/* Device memory helpers so a ctypes host needs no other GPU runtime. */
int sslam_malloc(sslam_ctx* ctx, size_t bytes, void** dptr_out);
int sslam_free(sslam_ctx* ctx, void* dptr);
int sslam_memcpy_h2d(sslam_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int sslam_memcpy_d2h(sslam_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
/* Enqueue-only helpers on the context's stream, and events to order one context's stream behind
 * another's: what the frame pipeline (several extractor / matcher contexts on one GPU) needs, so
 * the single-GPU product path runs without any other GPU runtime. */
int sslam_memcpy_d2d_async(sslam_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes);
int sslam_memset_async(sslam_ctx* ctx, void* dst_dev, int value, size_t bytes);
/* Page-locked host memory and enqueue-only host <-> device copies on the context's stream (the host buffer
 * must stay valid, and for a real overlap be page-locked, until the stream has passed the copy): the
 * drop-in `feature_extractor` / `feature_matcher` (slam/core/features_utils.py:85-171) stage the image and
 * read the result records back through these instead of the torch `.cuda()` / `.cpu()` calls at :222, :97. */
int sslam_host_alloc(sslam_ctx* ctx, size_t bytes, void** hptr_out);
int sslam_host_free(sslam_ctx* ctx, void* hptr);
int sslam_memcpy_h2d_async(sslam_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int sslam_memcpy_d2h_async(sslam_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int sslam_event_create(sslam_ctx* ctx, void** event_out);
int sslam_event_destroy(void* event);
int sslam_event_record(sslam_ctx* ctx, void* event);      /* on ctx's stream */
int sslam_ctx_wait_event(sslam_ctx* ctx, void* event);    /* ctx's stream waits, host does not */
/* Timing events (the ordering events above are created without timestamps) and the time between two recorded
 * ones (waits for the second): per-round durations of a running pipeline without synchronising inside it. */
int sslam_timing_event_create(sslam_ctx* ctx, void** event_out);
int sslam_event_elapsed_ms(void* start_event, void* stop_event, float* ms_out);

/* ------------------------------------------------------------------ local BA
 * Batched reprojection residual + Jacobian.  Replaces the per-observation
 * `cost_functions.ReprojErrorCost(CameraModelId.PINHOLE, uv)` blocks that
 * slam/core/ba_utils.py:56-68 adds and Ceres evaluates inside
 * `pyceres.solve` (ba_utils.py:293).  All float64.
 *   pose_idx[n_obs], point_idx[n_obs] : int32 indices into q/t and X
 *   uv[n_obs*2]; q[n_poses*4] quaternion (x,y,z,w) cam-from-world;
 *   t[n_poses*3]; X[n_points*3]; intr[4] = fx,fy,cx,cy
 *   r[n_obs*2]; Jq[n_obs*8] (2x4 row-major, ambient xyzw); Jt[n_obs*6];
 *   JX[n_obs*6].  Any of Jq/Jt/JX may be NULL (residual only). */
int sslam_ba_residual_jacobian_host(sslam_ctx* ctx, int n_obs,
                                    const int32_t* pose_idx, const int32_t* point_idx,
                                    const double* uv, int n_poses, const double* q,
                                    const double* t, int n_points, const double* X,
                                    const double* intr, double* r, double* Jq,
                                    double* Jt, double* JX);
int sslam_ba_residual_jacobian_dev(sslam_ctx* ctx, int n_obs,
                                   const int32_t* pose_idx, const int32_t* point_idx,
                                   const double* uv, int n_poses, const double* q,
                                   const double* t, int n_points, const double* X,
                                   const double* intr, double* r, double* Jq,
                                   double* Jt, double* JX);

/* Whole local-BA solve on the device: replaces `pyceres.solve(opts, problem, summary)` at
 * slam/core/ba_utils.py:288-293 for the problem `_core_ba` assembles (:220-286): Huber(delta)
 * on every reprojection block (:236), EigenQuaternionManifold on every quaternion (:245-249),
 * constant blocks for the gauge keyframes (:250-257) and - with points_const - for the
 * landmarks (pose_only_ba, :89-140); Ceres' default trust-region constants, at most
 * `max_iters` iterations (:289-292).  Levenberg-Marquardt with the Schur complement onto the
 * optimised poses; linearisation, reduction, Cholesky, step evaluation and the accept / reject /
 * terminate logic all run on the GPU as one enqueue (no host round trip per iteration).
 *   q[n_poses*4], t[n_poses*3], X[n_points*3] : in = initial values, out = optimised
 *   pose_const[n_poses] : 1 = held fixed.  At most 256 non-constant poses (more is refused with an error;
 *                         up to 12 - local BA - the reduced pose system is factored in LDS, beyond that in
 *                         device memory by one workgroup; the dense Schur blocks must fit 32 GiB)
 *   summary[8] : iterations, successful steps, initial cost, final cost (0.5 sum rho),
 *                termination (0 max iterations, 1 gradient, 2 parameter, 3 function tolerance,
 *                4 trust region collapsed), final radius, #optimised poses, 0 */
int sslam_ba_solve_host(sslam_ctx* ctx, int n_obs, const int32_t* pose_idx,
                        const int32_t* point_idx, const double* uv, int n_poses, double* q,
                        double* t, const unsigned char* pose_const, int n_points, double* X,
                        const double* intr, int max_iters, double huber_delta, int points_const,
                        double* summary);

/* ------------------------------------------------------- F-matrix RANSAC
 * Replaces `cv2.findFundamentalMat(pts1, pts2, cv2.FM_RANSAC, thresh, 0.99)` inside
 * `filter_matches_ransac` (slam/core/features_utils.py:185-200): OpenCV 4.x's classic path
 * (7-point RANSAC for >= 15 matches, LMedS for 8..14, cv::RNG sample stream), scored on the GPU.
 *   pts1, pts2 : float32 [n][2] matched pixel coordinates (host), n >= 8
 *   thresh <= 0 -> 3, confidence outside (0,1) -> 0.99, max_iters <= 0 -> 1000 (cv2 defaults)
 *   mask_out[n] : 1 = inlier;  F_out[9] (may be NULL): row-major, F[8] = 1 or 0
 *   info_out[4] (may be NULL): inliers (-1: no model, cv2 would return mask None),
 *                              iterations the sequential loop runs, 1 if LMedS, winning sample */
int sslam_fmat_ransac_host(sslam_ctx* ctx, int n, const float* pts1, const float* pts2,
                           double thresh, double confidence, int max_iters,
                           unsigned char* mask_out, double* F_out, int* info_out);

/* Device-resident form of the same filter, for a tracking step that keeps the matcher's output on the
 * GPU: consumes what `sslam_lightglue_match_dev` wrote and leaves what `filter_matches_ransac` would
 * return (features_utils.py:185-200) on the device; enqueued on ctx's stream, no host round trip.
 *   n_max: bound on the match count;  n_dev: device int32 count (e.g. the matcher's info[0]), clamped to
 *   [0, n_max]; NULL = exactly n_max matches
 *   xy1_dev, xy2_dev: keypoints [*][2] float32;  ij_dev: int32 [n][2] (query, train) index pairs
 *   mask_out_dev[n_max] (may be NULL): 1 = kept;  ij_out_dev[n_max][2] (may be NULL): the kept pairs, in order
 *   F_out_dev[9] double (may be NULL);  info_out_dev[4] int32: pairs kept, iterations, 1 if LMedS,
 *   winning sample (-1: no model -> nothing kept, as the reference's `mask is None`; -2: fewer than 8
 *   matches, all kept unfiltered as the reference does)
 * The context's scratch buffer is (re)allocated when n_max grows: call once with the largest n_max
 * before capturing or pipelining. */
int sslam_fmat_ransac_dev(sslam_ctx* ctx, int n_max, const int32_t* n_dev, const float* xy1_dev,
                          const float* xy2_dev, const int32_t* ij_dev, double thresh, double confidence,
                          int max_iters, unsigned char* mask_out_dev, int32_t* ij_out_dev,
                          double* F_out_dev, int32_t* info_out_dev);

/* ------------------------------------------- 2D-3D association for tracking
 * Replaces the per-point loop of `reproject_and_match_2d3d` (slam/core/pnp_utils.py:224-304) for
 * float descriptors: projection (`_project_points` :127-141), radius search (cKDTree :238, :265),
 * min L2 over the point's last six observation descriptors (:107-120) and the greedy `used_kps`
 * pass in map order (:260-286).
 *   pts3d[n_points*3] float64 world points in `world_map.points` order
 *   obs_cnt[n_points]: 0..6 = descriptors among the point's last six observations; 0 also for a point
 *     whose LAST observation has none (the reference skips it, :270-272)
 *   obs_desc[n_points*6*128]: those descriptors, valid ones first
 *   K9 row-major 3x3, Tcw16 row-major 4x4 (camera-from-world), kp_xy[n_kp*2], des[n_kp*128] float32
 *   kp_of_point[n_points]: out, matched keypoint index or -1;  uv_out[n_points*2] (may be NULL)
 *   info_out[2] (may be NULL): matches, candidate points */
int sslam_reproject_match_host(sslam_ctx* ctx, int n_points, const double* pts3d,
                               const int32_t* obs_cnt, const float* obs_desc, const double* K9,
                               const double* Tcw16, int n_kp, const float* kp_xy, const float* des,
                               int img_w, int img_h, double radius_px, double max_dist,
                               int32_t* kp_of_point, float* uv_out, int32_t* info_out);

/* Device-resident variant (enqueue only): map arrays and keypoints / descriptors are device pointers
 * (an incrementally maintained SoA map - slam/core/landmark_utils.py of the overlay - keeps the
 * former on the GPU, sslam_aliked_extract_dev writes the latter); K9 / Tcw16 stay host values.
 * kp_of_point[n_points], uv_out (may be NULL), info_out[4] = {matches, candidate-list overflow flag,
 * candidate points, 0} are device buffers. */
int sslam_reproject_match_dev(sslam_ctx* ctx, int n_points, const double* pts3d,
                              const int32_t* obs_cnt, const float* obs_desc, const double* K9,
                              const double* Tcw16, int n_kp, const float* kp_xy, const float* des,
                              int img_w, int img_h, double radius_px, double max_dist,
                              int32_t* kp_of_point, float* uv_out, int32_t* info_out);

/* ------------------------------------------------------------------ ALIKED
 * Replaces `ALIKED(max_num_keypoints=...).eval().to(device)` at
 * slam/core/features_utils.py:25 and `_bgr_to_tensor` + `detector.extract` +
 * `rbd` + descriptor re-normalisation at features_utils.py:92-100, :219-222.
 * `weights` = blob from opencv-simpleslam_amd/weights.py::pack_aliked.
 * max_h/max_w bound the input image, max_kpts the keypoints per call. */
int sslam_aliked_create(sslam_ctx* ctx, const float* weights, size_t n_floats, int max_h, int max_w,
                        int max_kpts, sslam_aliked** out);
/* An instance whose _extract_batch_dev entry takes up to max_frames (<= 16) frames per call: one workspace block
 * per frame (about 0.75 GB each at the full 1056 x 1056 network capacity). */
int sslam_aliked_create_batched(sslam_ctx* ctx, const float* weights, size_t n_floats, int max_h, int max_w,
                                int max_kpts, int max_frames, sslam_aliked** out);
int sslam_aliked_destroy(sslam_aliked* al);
/* img: uint8 HWC, C = 3 (BGR as cv2.imread gives), 1 (gray) or 4 (BGRA).
 * xy_out[2*max_kpts] (x, y) in input-image pixels, desc_out[128*max_kpts]
 * unit-norm rows, score_out[max_kpts] (may be NULL), n_out = keypoints found
 * (<= max_kpts; threshold mode, ordered as upstream: raster order, or by
 * descending score when more than max_kpts pass the detection threshold; -1 on the _dev entries: range overflow, below). */
int sslam_aliked_extract_host(sslam_aliked* al, const uint8_t* img, int H, int W, int C, int max_kpts,
                              float* xy_out, float* desc_out, float* score_out, int32_t* n_out);
/* Device-pointer variant (all pointers device, n_out[1] device int32); enqueue only. */
int sslam_aliked_extract_dev(sslam_aliked* al, const uint8_t* img, int H, int W, int C, int max_kpts,
                             float* xy_out, float* desc_out, float* score_out, int32_t* n_out);
/* n_frames frames of ONE size through ONE launch sequence (the reference extracts frame by frame,
 * features_utils.py:92-100; a frame stream hands over several at once): imgs / xy_out / desc_out / score_out / n_out
 * are host arrays of n_frames device pointers (score_out, or any of its entries, may be NULL).  Every kernel carries
 * the frame in a grid dimension, so the ~50 launches of a frame become ~50 per BATCH and the small stages (1/8 and
 * 1/32 resolution maps, selection, descriptor head) fill the chip; a frame's arithmetic does not depend on the batch:
 * results are those of n_frames sslam_aliked_extract_dev calls, bit for bit.  Enqueue only. */
int sslam_aliked_extract_batch_dev(sslam_aliked* al, int n_frames, const uint8_t* const* imgs, int H, int W, int C,
                                   int max_kpts, float* const* xy_out, float* const* desc_out,
                                   float* const* score_out, int32_t* const* n_out);
/* RANGE.  The dense stages, the deformable layers and the descriptor head carry their operands as fp16 (hi, lo) plane pairs
 * (the split-precision matrix path): a FINITE activation with |value| >= 65520 does not fit.  Such a value raises the frame's
 * flag on the device; the frame's keypoint count n_out then reads -1 (sslam_lightglue_match_dev treats a negative count as an
 * empty frame), sslam_aliked_extract_host fails with a message, and the instance keeps a sticky word that this call returns
 * and clears (synchronises the stream) - for callers of the _dev / _batch_dev entries that do not read the counts.
 * sslam_aliked_create refuses weights (BN scales folded) that do not fit.  No trained checkpoint comes near the limit; the
 * guard exists so that a silent inf / NaN descriptor cannot happen. */
int sslam_aliked_range_overflow(sslam_aliked* al, int* flag_out);
/* Replay the launch sequence of sslam_aliked_extract_dev as a cached hipGraph (one graph per distinct
 * argument tuple; for callers that cycle through a fixed set of buffers).  Same results, ~10 us of
 * host time per call instead of ~45 launches. */
int sslam_aliked_use_graphs(sslam_aliked* al, int enable);
/* Test hook: copy an internal buffer to the host (see aliked_kernels.hip). */
int sslam_aliked_debug_read(sslam_aliked* al, int which, void* dst, size_t bytes);

/* ------------------------------------------------------------------ LightGlue
 * Replaces `LightGlue(features='aliked').eval().to(device)` at
 * slam/core/features_utils.py:26 and the forward + confidence filter at
 * features_utils.py:157-169.  `weights` is the fp32 blob produced by
 * opencv-simpleslam_amd/weights.py::pack_lightglue from an upstream state dict
 * (host pointer, copied).  max_kpts bounds M and N of every later call. */
int sslam_lightglue_create(sslam_ctx* ctx, const float* weights, size_t n_floats, int max_kpts,
                           sslam_lightglue** out);
/* Same, with workspace for up to max_pairs (1..16) pairs per sslam_lightglue_match_batch_dev call.
 * sslam_lightglue_create == max_pairs 1.  Workspace is ~60 MB per pair at max_kpts 2048. */
int sslam_lightglue_create_batched(sslam_ctx* ctx, const float* weights, size_t n_floats, int max_kpts,
                                   int max_pairs, sslam_lightglue** out);
int sslam_lightglue_destroy(sslam_lightglue* lg);
int sslam_lightglue_capacity(sslam_lightglue* lg, int* kc_out);
int sslam_lightglue_batch_capacity(sslam_lightglue* lg, int* pairs_out);
/* Upstream conf: depth_confidence 0.95, width_confidence 0.99, filter_threshold
 * 0.1; prune_min_kpts = pruning_keypoint_thresholds[device] (-1 on CPU: pruning
 * evaluated after every layer; pass a value >= max_kpts to disable). */
int sslam_lightglue_set_conf(sslam_lightglue* lg, float depth_confidence, float width_confidence,
                             float filter_threshold, int prune_min_kpts);
/* Arithmetic of the 9 transformer layers.  0: every contraction on the exact-fp32 matrix-core
 * instruction (v_mfma_f32_32x32x2_f32).  1: fp16 hi/lo split operands, three
 * v_mfma_f32_32x32x16_f16 per product, fp32 accumulation (~2^-22 relative error per product).
 * 2 (DEFAULT since r05): as 1, but attention carries the softmax weights P as ONE fp16 plane in P.V (two MFMAs per product
 * there, the row sum over the rounded weights): -12 % attention time.  On north_star's bar (match indices, floats within 1e-3)
 * modes 1 and 2 are indistinguishable over 131 199 oracle matches - the same two score-at-threshold events, score error
 * 4.4e-5 / 1.06e-4 (profiles/r05_flip_soak.md); token states 2.4e-5 from exact in mode 2, 4e-6 in mode 1.
 * final_proj and the similarity GEMM run on the same split-operand pipe in modes 1 and 2 (exact-fp32 instruction in
 * mode 0); the dual softmax, the arg-max and the score arithmetic are fp32 in every mode. */
int sslam_lightglue_set_precision(sslam_lightglue* lg, int mode);
/* xy0[M*2], desc0[M*128], xy1[N*2], desc1[N*128] float32.
 * ij_out[2*min(M,N)] int32 (queryIdx, trainIdx) pairs, ascending queryIdx;
 * score_out[min(M,N)]; only matches with score > filter_threshold AND
 * score > min_conf (features_utils.py:167-169) are emitted.
 * M == 0 or N == 0 -> k_out = 0 (features_utils.py:118-124). */
int sslam_lightglue_match_host(sslam_lightglue* lg, const float* xy0, const float* desc0, int M,
                               const float* xy1, const float* desc1, int N, float min_conf,
                               int32_t* ij_out, float* score_out, int32_t* k_out,
                               int32_t* stop_layer_out);
/* The same with the 'image_size' of each feature set (size0 / size1: host float[2] = (W, H), or NULL): LightGlue then
 * normalises the keypoints by the image size instead of their bounding box - what happens when the features
 * come straight from `extractor.extract`, i.e. the reference's legacy pair entry `_lightglue_detect_and_match`
 * (slam/core/features_utils.py:233-247; cvg/LightGlue `normalize_keypoints(kpts, size)`). */
int sslam_lightglue_match_host_sized(sslam_lightglue* lg, const float* xy0, const float* desc0, int M, const float* size0,
                                     const float* xy1, const float* desc1, int N, const float* size1, float min_conf,
                                     int32_t* ij_out, float* score_out, int32_t* k_out, int32_t* stop_layer_out);
/* Device-pointer variant; enqueue only.  M, N bound the rows of the input arrays; m_dev /
 * n_dev (device int32[1], may be NULL) carry the actual keypoint counts when they are only
 * known on the device (written by sslam_aliked_extract_dev), so an extract -> match chain
 * needs no host round trip.  info_out[4] (device) = {K, stop_layer, n0, n1 after pruning}; K = -1: range
 * overflow of the split-precision path (see sslam_lightglue_range_overflow). */
int sslam_lightglue_match_dev(sslam_lightglue* lg, const float* xy0, const float* desc0, int M,
                              const float* xy1, const float* desc1, int N, const int32_t* m_dev,
                              const int32_t* n_dev, float min_conf, int32_t* ij_out, float* score_out,
                              int32_t* info_out);
/* Batch variant: n_pairs independent pairs in ONE enqueue (every launch covers all pairs, so the chip
 * is filled without splitting the keys of a pair and the launch sequence is paid once per batch).
 * This is what a frame stream calls: the reference matches one (t-1, t) pair per frame
 * (slam/monocular/main_revamped.py:321-328); consecutive pairs are independent.
 * Host arrays of n_pairs entries: xy0[p] / desc0[p] / xy1[p] / desc1[p] device pointers, M[p] / N[p]
 * row bounds, m_dev[p] / n_dev[p] device counts (the arrays or single entries may be NULL).
 * Outputs (device): pair p writes ij_out + p*out_stride*2, score_out + p*out_stride, info_out + 4p;
 * out_stride >= min(M[p], N[p]).  Each pair's result is the one sslam_lightglue_match_dev gives. */
int sslam_lightglue_match_batch_dev(sslam_lightglue* lg, int n_pairs, const float* const* xy0,
                                    const float* const* desc0, const int32_t* const* m_dev, const int32_t* M,
                                    const float* const* xy1, const float* const* desc1,
                                    const int32_t* const* n_dev, const int32_t* N, float min_conf,
                                    int32_t* ij_out, float* score_out, int32_t* info_out, int out_stride);
/* Replay the launch sequence of sslam_lightglue_match_dev / _batch_dev as a cached hipGraph (one graph
 * per distinct argument tuple; ~190 launches become one hipGraphLaunch).  Same results. */
int sslam_lightglue_use_graphs(sslam_lightglue* lg, int enable);
/* The split-precision path carries fp32 values as fp16 plane pairs: a FINITE activation with
 * |value| >= 65520 does not fit (the exact-fp32 path, precision 0, has no such limit).  Such a value is flagged
 * on the device PER PAIR (and saturated or turned non-finite, never silently wrapped): that pair's match count
 * info_out[0] becomes -1, so the verdict travels with the result (sslam_fmat_ransac_dev clamps it to 0 matches);
 * sslam_lightglue_match_host fails with a message.  The instance also keeps a sticky word of its own (never
 * shared with other instances): callers of the _dev / _batch_dev entries that do not read info_out poll it
 * here (synchronises the stream, returns and clears it). */
int sslam_lightglue_range_overflow(sslam_lightglue* lg, int* flag_out);
/* Measurement hook (bench.py): bracket each attention launch - the dominant kernel - with HIP
 * events on the context stream; _read synchronises and returns their summed duration and count. */
int sslam_lightglue_profile(sslam_lightglue* lg, int enable);
int sslam_lightglue_profile_read(sslam_lightglue* lg, float* total_ms_out, int32_t* launches_out);
/* Test hooks: limit the executed layers; copy an internal buffer to the host.
 * debug_key_split: 0 = by batch size (no split for batched launches, 2 or 4 key ranges for one pair, merged by the fused
 * FFN's tiles; the hand-scheduled assembly attention kernel either way), -5 = the same with the merge as a launch of its
 * own, -4 = that policy on the r02 4-wave kernel, 1 / 2 / 4 = that
 * many key ranges (4-wave kernel), 101 / 102 / 104 = that many (assembly kernel), -1 = no split, r02 4-wave kernel,
 * -3 = no split, the assembly kernel at any batch size (for one split the two kernels give bit-identical results). */
int sslam_lightglue_debug_layers(sslam_lightglue* lg, int layers, int self_only);
int sslam_lightglue_debug_key_split(sslam_lightglue* lg, int ks);
/* Linear-kernel form: -1 by token count (default), 0 = 64-row ring kernels, 1 = batched form (128 x 128 projections +
 * the fused FFN kernel, its tile size by token count), 2 / 3 = batched form with 64- / 32-token FFN tiles forced
 * (a token's FFN arithmetic is the same in both: bit-identical), 5 = batched form with the token heads (early stop /
 * pruning inputs) as a launch of their own instead of in the cross block's fused FFN. */
int sslam_lightglue_debug_big_gemm(sslam_lightglue* lg, int mode);
/* Precision-study hook (profiles/r04_split_study.md; the product never sets it): drop cross terms of the three-term
 * split products and measure what that does to the matches.  mask: 0x01 / 0x02 K / Q as one fp16 plane in the logits,
 * 0x04 / 0x08 P / V as one plane in the context, 0x10 / 0x20 activation low plane dropped in the projections / the FFN,
 * 0x40 / 0x80 weight low plane dropped there.  0 = the product arithmetic. */
int sslam_lightglue_debug_split_form(sslam_lightglue* lg, int mask);
int sslam_lightglue_debug_read(sslam_lightglue* lg, int which, void* dst, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* SSLAM_HIP_H */
