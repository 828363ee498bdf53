// ransac_kernels.hip - fundamental-matrix RANSAC / LMedS outlier filter on the GPU.
//
// Replaces `cv2.findFundamentalMat(pts1, pts2, cv2.FM_RANSAC, thresh, 0.99)` as called by
// `filter_matches_ransac` (slam/core/features_utils.py:185-200; callers main_revamped.py:126,
// keyframe_utils.py:154, triangulation_utils.py:132) - SURVEY.md section 8(f) rank 2.
//
// The algorithm is OpenCV 4.x's classic (non-USAC) path, restated from its published source
// (modules/calib3d/src/fundam.cpp, ptsetreg.cpp; opencv_python==4.11.0.86 in requirements.txt):
//   * >= 15 points: RANSAC, 7-point minimal solver (1-3 models per sample), error = max of the two
//     squared point-to-epipolar-line distances (computed in double, rounded to float), inlier
//     iff err <= (float)(thresh^2), best = strictly larger inlier count, iteration budget
//     re-estimated after every improvement (RANSACUpdateNumIters), no refit on the inliers;
//   * 8..14 points: LMedS (least median of the same error; sigma = 2.5*1.4826*(1+5/(n-7))*
//     sqrt(median), inliers at sigma), budget from a fixed 45 % outlier ratio;
//   * samples: cv::RNG (multiply-with-carry, state 2^64-1), getSubset (re-draw duplicates,
//     reject subsets whose last point is collinear with / too close to an earlier pair).
// PARITY UNPINNED: cv2 is absent here; the null space comes from Gauss-Jordan elimination with
// complete pivoting instead of OpenCV's Jacobi SVD (same solutions up to rounding).
//
// The sequential loop is data-dependent only through (a) the RNG stream and (b) the running
// best / iteration budget.  (a) is replayed by one lane (a few thousand integer ops); the
// expensive part - solving every sample and scoring its models against every match - runs
// wide (workgroup per sample); (b) is then replayed exactly by one lane over the stored scores.
// The result is what the sequential loop would have produced.  Seven launches per call (r05):
// head [gather, control block, samples of chunk 0] - per chunk [solve + score] and [replay +
// samples of the next chunk] - tail [replay, winner, mask, compaction].
#include "common.hpp"

#include <algorithm>
#include <cfloat>

namespace {

constexpr int RS_MAX_ITERS = 1000;       // cv::findFundamentalMat default maxIters
constexpr int RS_MP = 7;                 // model points
constexpr int RS_T = 256;
constexpr int RS_SUBSET_ATTEMPTS = 10000;
constexpr int RS_LMEDS_MAX = 14;         // < 15 points -> LMedS

struct RSCtrl {
    int n_subsets;      // samples generated so far (getSubset may give up: `exhausted`)
    int lmeds;          // 1 = LMedS path
    int niters;         // iterations the sequential loop has run
    int best_h, best_k; // winning sample / model (-1: none)
    int best_count;     // inliers of the winner (RANSAC) / final count
    int budget;         // the loop's current iteration budget (`niters` in ptsetreg.cpp)
    int max_good;       // best inlier count so far (RANSAC)
    int exhausted;      // getSubset failed: the loop ended
    int pad;
    unsigned long long rng_state;
    double min_median;  // LMedS
    double thresh;      // threshold actually applied by the mask pass (pixels)
    double F[9];
};

struct RSArgs {
    int n; int max_iters;                     // n: match count, or its bound when n_dev is given
    const int32_t* n_dev;                     // device-resident count (the matcher's), clamped to [0, n]; may be NULL
    const float* xy1; const float* xy2;       // _dev entry: keypoint arrays the index pairs refer to
    const int32_t* ij; int32_t* ij_out; int32_t* info_out; double* F_out;
    int h0, h1;                               // sample range of this chunk
    double thresh, confidence;
    const float* p1; const float* p2;         // [n][2]
    int* subsets;                             // [max_iters][7]
    double* models;                           // [max_iters][3][9]
    int* nmodels;                             // [max_iters]
    int* counts;                              // [max_iters][3]  RANSAC inlier counts
    float* medians;                           // [max_iters][3]  LMedS medians
    unsigned char* mask;                      // [n]
    RSCtrl* ctrl;
};

__device__ __forceinline__ int rs_n(const RSArgs& a) { return a.n_dev ? min(max(a.n_dev[0], 0), a.n) : a.n; }

// ---- cv::RNG ------------------------------------------------------------------------------
struct CvRng {
    unsigned long long state;
    __device__ unsigned next() {
        state = (unsigned long long)(unsigned)state * 4164903690ULL + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    __device__ int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

// haveCollinearPoints(m, count): last point against every earlier pair
__device__ bool last_point_collinear(const float* p, const int* idx, int count) {
    const int i = count - 1;
    const float xi = p[2 * idx[i]], yi = p[2 * idx[i] + 1];
    for (int j = 0; j < i; ++j) {
        const double dx1 = p[2 * idx[j]] - xi, dy1 = p[2 * idx[j] + 1] - yi;
        for (int k = 0; k < j; ++k) {
            const double dx2 = p[2 * idx[k]] - xi, dy2 = p[2 * idx[k] + 1] - yi;
            if (fabs(dx2 * dy1 - dy2 * dx1) <= (double)FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2)))
                return true;
        }
    }
    return false;
}

// RANSACUpdateNumIters(p, ep, modelPoints, maxIters)
__device__ int update_num_iters(double p, double ep, int model_points, int max_iters) {
    p = fmax(p, 0.0); p = fmin(p, 1.0);
    ep = fmax(ep, 0.0); ep = fmin(ep, 1.0);
    double num = fmax(1.0 - p, DBL_MIN);
    double denom = 1.0 - pow(1.0 - ep, (double)model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)rint(num / denom);
}

// ---- 1. replay the sample stream for samples [h0, h1) (one lane) ------------------------------
// r05: the lane's collinearity tests read 2 x 7 points per attempt - dependent global loads, ~3 us per sample, 50 us for the
// 16-sample first chunk that is all a matcher's inlier ratios ever need.  The workgroup now copies both point sets into LDS
// first (when they fit: RS_LDS_POINTS) and the lane reads them from there; a chunk that has nothing to do (budget already
// below its first sample) leaves before the copy.
// A step of the whole workgroup (256 threads; rs_pts = dynamic LDS, [p1: n x 2 | p2: n x 2] when n <= RS_LDS_POINTS;
// pts_ready: the caller has already filled it).  Ends with every thread past the last use of rs_pts.
constexpr int RS_LDS_POINTS = 4096;      // 2 x 4096 x 8 bytes = 64 KB of dynamic LDS at most
__device__ void rs_subsets_step(const RSArgs& a, int h0, int h1, float* rs_pts, bool pts_ready, int* go) {
    RSCtrl* c = a.ctrl;
    const int n = rs_n(a);
    if (threadIdx.x == 0) *go = !c->exhausted && h0 < min(h1, c->budget) && h0 == c->niters;
    __syncthreads();
    if (!*go) return;
    const bool in_lds = n <= RS_LDS_POINTS;
    if (in_lds && !pts_ready) {
        for (int i = threadIdx.x; i < 2 * n; i += blockDim.x) { rs_pts[i] = a.p1[i]; rs_pts[2 * n + i] = a.p2[i]; }
        __syncthreads();
    }
    if (threadIdx.x >= 64) return;                 // one wave samples: lane 0 owns the RNG, lanes 0 .. 29 share the collinearity tests
    const int lane = threadIdx.x;
    const float* p1 = in_lds ? rs_pts : a.p1;
    const float* p2 = in_lds ? rs_pts + 2 * n : a.p2;
    // haveCollinearPoints tests the LAST point of the subset against every earlier pair (j, k < j): 15 pairs x 2 point sets =
    // 30 independent tests, one per lane (a single lane spent ~1 200 instructions per sample on them, most of the 46 us of the
    // first chunk); the verdict is their OR, as in last_point_collinear
    const int q = lane % 15, set = lane / 15;
    int tj = 1, tk = 0;
    for (int t = 0; t < q; ++t) { if (++tk == tj) { ++tj; tk = 0; } }        // q -> (j, k): (1,0) (2,0) (2,1) (3,0) ...
    CvRng rng{c->rng_state};
    const int end = min(h1, c->budget);
    int made = c->n_subsets;
    bool exhausted = false;
    for (int it = h0; it < end; ++it) {
        int idx[RS_MP];
        int attempts = 0;
        for (; attempts < RS_SUBSET_ATTEMPTS; ++attempts) {
            if (lane == 0) {
                for (int i = 0; i < RS_MP; ++i) {
                    int v;
                    bool dup;
                    do {
                        v = rng.uniform(0, n);
                        dup = false;
                        for (int j = 0; j < i; ++j) dup |= idx[j] == v;
                    } while (dup);
                    idx[i] = v;
                }
            }
#pragma unroll
            for (int i = 0; i < RS_MP; ++i) idx[i] = __shfl(idx[i], 0);
            bool bad = false;
            if (lane < 30) {
                const float* p = set ? p2 : p1;
                const float xi = p[2 * idx[RS_MP - 1]], yi = p[2 * idx[RS_MP - 1] + 1];
                const double dx1 = p[2 * idx[tj]] - xi, dy1 = p[2 * idx[tj] + 1] - yi;
                const double dx2 = p[2 * idx[tk]] - xi, dy2 = p[2 * idx[tk] + 1] - yi;
                bad = fabs(dx2 * dy1 - dy2 * dx1) <= (double)FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2));
            }
            if (!__any(bad)) break;
        }
        if (attempts == RS_SUBSET_ATTEMPTS) { exhausted = true; break; }   // getSubset failed: the loop ends here
        if (lane < RS_MP) a.subsets[it * RS_MP + lane] = idx[lane];
        ++made;
    }
    if (lane == 0) {
        if (exhausted) c->exhausted = 1;
        c->n_subsets = made;
        c->rng_state = rng.state;
    }
}

// ---- 2. 7-point solver (thread / sample) -------------------------------------------------------
__device__ __forceinline__ double det3(const double* m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// cv::solveCubic for c[0] x^3 + c[1] x^2 + c[2] x + c[3] = 0; returns the number of real roots
__device__ int solve_cubic(const double* c, double* r) {
    double a0 = c[0], a1 = c[1], a2 = c[2], a3 = c[3];
    int n = 0;
    double x0 = 0, x1 = 0, x2 = 0;
    if (a0 == 0) {
        if (a1 == 0) {
            if (a2 == 0) n = a3 == 0 ? -1 : 0;
            else { x0 = -a3 / a2; n = 1; }
        } else {
            double d = a2 * a2 - 4 * a1 * a3;
            if (d >= 0) {
                d = sqrt(d);
                const double q1 = (-a2 + d) * 0.5, q2 = (a2 + d) * -0.5;
                if (fabs(q1) > fabs(q2)) { x0 = q1 / a1; x1 = a3 / q1; }
                else { x0 = q2 / a1; x1 = a3 / q2; }
                n = d > 0 ? 2 : 1;
            }
        }
    } else {
        a0 = 1. / a0; a1 *= a0; a2 *= a0; a3 *= a0;
        const double Q = (a1 * a1 - 3 * a2) * (1. / 9);
        const double R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54);
        const double Qcubed = Q * Q * Q;
        double d = Qcubed - R * R;
        if (d > 0) {
            const double theta = acos(R / sqrt(Qcubed));
            const double sqrtQ = sqrt(Q);
            const double t0 = -2 * sqrtQ, t1 = theta * (1. / 3), t2 = a1 * (1. / 3);
            x0 = t0 * cos(t1) - t2;
            x1 = t0 * cos(t1 + (2. * M_PI / 3)) - t2;
            x2 = t0 * cos(t1 + (4. * M_PI / 3)) - t2;
            n = 3;
        } else if (d == 0) {
            if (R >= 0) { x0 = -2 * pow(R, 1. / 3) - a1 / 3; x1 = pow(R, 1. / 3) - a1 / 3; }
            else { x0 = 2 * pow(-R, 1. / 3) - a1 / 3; x1 = -pow(-R, 1. / 3) - a1 / 3; }
            x2 = 0;
            n = x0 == x1 ? 1 : 2;
            x1 = x0 == x1 ? 0 : x1;
        } else {
            d = sqrt(-d);
            double e = pow(d + fabs(R), 1. / 3);
            if (R > 0) e = -e;
            x0 = (e + Q / e) - a1 * (1. / 3);
            n = 1;
        }
    }
    r[0] = x0; r[1] = x1; r[2] = x2;
    return n;
}

// One lane solves sample h: the 7 x 9 system lives in LDS (`As`, 63 doubles) - the elimination below indexes it with the PIVOT's
// row and column, which are run-time values: as a private array it lived in scratch memory (512 bytes per thread, every access
// a ~1 us round trip: 33 us for the eight samples of the first chunk, r05 kernel trace).  Writes the sample's 1 - 3 models to
// `out` (27 doubles) and returns their number.
__device__ int rs_solve7(const RSArgs& a, int h, double* As, double* out) {
#define RS_A(i, j) As[(i) * 9 + (j)]
    // rows: (m2, 1)^T F (m1, 1) = 0
    for (int i = 0; i < RS_MP; ++i) {
        const int id = a.subsets[h * RS_MP + i];
        const double x0 = a.p1[2 * id], y0 = a.p1[2 * id + 1], x1 = a.p2[2 * id], y1 = a.p2[2 * id + 1];
        RS_A(i, 0) = x1 * x0; RS_A(i, 1) = x1 * y0; RS_A(i, 2) = x1; RS_A(i, 3) = y1 * x0; RS_A(i, 4) = y1 * y0; RS_A(i, 5) = y1;
        RS_A(i, 6) = x0; RS_A(i, 7) = y0; RS_A(i, 8) = 1;
    }
    // Gauss-Jordan with complete pivoting -> two free columns span the null space
    int colp[9];
    for (int j = 0; j < 9; ++j) colp[j] = j;
    bool ok = true;
    for (int k = 0; k < RS_MP; ++k) {
        int pr = k, pc = k;
        double best = -1.0;
        for (int i = k; i < RS_MP; ++i)
            for (int j = k; j < 9; ++j) {
                const double v = fabs(RS_A(i, j));
                if (v > best) { best = v; pr = i; pc = j; }
            }
        if (!(best > 0.0)) { ok = false; break; }
        for (int j = 0; j < 9; ++j) { const double tmp = RS_A(k, j); RS_A(k, j) = RS_A(pr, j); RS_A(pr, j) = tmp; }
        for (int i = 0; i < RS_MP; ++i) { const double tmp = RS_A(i, k); RS_A(i, k) = RS_A(i, pc); RS_A(i, pc) = tmp; }
        { const int tmp = colp[k]; colp[k] = colp[pc]; colp[pc] = tmp; }
        const double inv = 1.0 / RS_A(k, k);
        for (int j = k; j < 9; ++j) RS_A(k, j) *= inv;
        for (int i = 0; i < RS_MP; ++i) {
            if (i == k) continue;
            const double f = RS_A(i, k);
            if (f != 0.0)
                for (int j = k; j < 9; ++j) RS_A(i, j) -= f * RS_A(k, j);
        }
    }
    int nm = 0;
    if (ok) {
        double f1[9], f2[9];             // null vectors for free columns 7 and 8 (permuted order)
        for (int k = 0; k < RS_MP; ++k) { f1[colp[k]] = -RS_A(k, 7); f2[colp[k]] = -RS_A(k, 8); }
        f1[colp[7]] = 1.0; f1[colp[8]] = 0.0;
        f2[colp[7]] = 0.0; f2[colp[8]] = 1.0;
        // F(lambda) = lambda f1 + (1 - lambda) f2 = lambda (f1 - f2) + f2; det F = 0 is a cubic
        double g[9];
        for (int i = 0; i < 9; ++i) g[i] = f1[i] - f2[i];
        double c[4], m[9];
        c[0] = det3(g);
        c[3] = det3(f2);
        c[1] = 0.0; c[2] = 0.0;
        for (int row = 0; row < 3; ++row) {
            for (int i = 0; i < 9; ++i) m[i] = g[i];
            for (int j = 0; j < 3; ++j) m[3 * row + j] = f2[3 * row + j];
            c[1] += det3(m);                                  // two rows of g, one of f2 -> lambda^2
            for (int i = 0; i < 9; ++i) m[i] = f2[i];
            for (int j = 0; j < 3; ++j) m[3 * row + j] = g[3 * row + j];
            c[2] += det3(m);                                  // one row of g, two of f2 -> lambda
        }
        double r[3];
        const int n = solve_cubic(c, r);
        if (n >= 1 && n <= 3) {
            for (int k = 0; k < n; ++k) {
                double lambda = r[k], mu = 1.0;
                const double s = g[8] * r[k] + f2[8];
                double F8;
                if (fabs(s) > DBL_EPSILON) { mu = 1.0 / s; lambda *= mu; F8 = 1.0; } else F8 = 0.0;
                for (int i = 0; i < 8; ++i) out[9 * nm + i] = g[i] * lambda + f2[i] * mu;
                out[9 * nm + 8] = F8;
                ++nm;
            }
        }
    }
    return nm;
#undef RS_A
}

// symmetric epipolar error of FMEstimatorCallback::computeError (double arithmetic, float result)
__device__ __forceinline__ float fm_error(const double* F, float x1, float y1, float x2, float y2) {
    double a = F[0] * x1 + F[1] * y1 + F[2];
    double b = F[3] * x1 + F[4] * y1 + F[5];
    double c = F[6] * x1 + F[7] * y1 + F[8];
    const double s2 = 1. / (a * a + b * b);
    const double d2 = x2 * a + y2 * b + c;
    a = F[0] * x2 + F[3] * y2 + F[6];
    b = F[1] * x2 + F[4] * y2 + F[7];
    c = F[2] * x2 + F[5] * y2 + F[8];
    const double s1 = 1. / (a * a + b * b);
    const double d1 = x1 * a + y1 * b + c;
    return (float)fmax(d1 * d1 * s1, d2 * d2 * s2);
}

// ---- 2 + 3. solve sample h and score its models against every match (workgroup / sample) -----------------------
// r05: one launch instead of two (thread / sample, then workgroup / model): the solver is a serial fp64 chain of ~10 us
// whatever the grid, and the 1 - 3 models of a sample are scored by the workgroup that has them in LDS.
__device__ __forceinline__ int rs_block_sum(int v, int* sh) {      // sum over the workgroup (RS_T threads), valid in every thread
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = RS_T / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    const int tot = sh[0];
    __syncthreads();
    return tot;
}

__global__ __launch_bounds__(RS_T) void rs_models_score_kernel(RSArgs a) {
    __shared__ double As[RS_MP * 9];
    __shared__ double Fm[27];
    __shared__ int sh[RS_T];
    __shared__ int s_nm;
    const int h = a.h0 + blockIdx.x;
    const RSCtrl* c = a.ctrl;
    if (h >= a.h1 || h >= c->n_subsets || a.h0 != c->niters) return;     // (beyond the samples drawn / the loop ended before this chunk)
    if (threadIdx.x == 0) {
        const int nm = rs_solve7(a, h, As, Fm);
        for (int i = 0; i < 9 * nm; ++i) a.models[(size_t)h * 27 + i] = Fm[i];
        a.nmodels[h] = nm;
        s_nm = nm;
    }
    __syncthreads();
    const int nm = s_nm, n = rs_n(a);
    if (c->lmeds) {
        // n <= 14: one lane per model sorts the errors and takes the median
        if ((int)threadIdx.x < nm) {
            const double* F = Fm + 9 * threadIdx.x;
            float e[RS_LMEDS_MAX];
            for (int i = 0; i < n; ++i) e[i] = fm_error(F, a.p1[2 * i], a.p1[2 * i + 1], a.p2[2 * i], a.p2[2 * i + 1]);
            for (int i = 1; i < n; ++i) {
                const float v = e[i];
                int j = i - 1;
                // NaN-safe insertion: a NaN compares false and stays where it is
                while (j >= 0 && e[j] > v) { e[j + 1] = e[j]; --j; }
                e[j + 1] = v;
            }
            a.medians[h * 3 + threadIdx.x] = n % 2 != 0 ? e[n / 2] : (e[n / 2 - 1] + e[n / 2]) * 0.5f;
        }
        return;
    }
    const float t = (float)(a.thresh * a.thresh);
    for (int k = 0; k < nm; ++k) {
        const double* F = Fm + 9 * k;
        int good = 0;
        for (int i = threadIdx.x; i < n; i += RS_T)
            good += fm_error(F, a.p1[2 * i], a.p1[2 * i + 1], a.p2[2 * i], a.p2[2 * i + 1]) <= t;
        good = rs_block_sum(good, sh);
        if (threadIdx.x == 0) a.counts[h * 3 + k] = good;
    }
}

// ---- 4. replay the sequential best / budget logic over samples [h0, h1) (one lane) -----------
__device__ void rs_select_step(const RSArgs& a, int h0, int h1) {
    RSCtrl* c = a.ctrl;
    const int n = rs_n(a);
    int it = h0;
    if (it != c->niters) return;                       // the loop already ended before this chunk
    if (c->lmeds) {
        for (; it < h1 && it < c->n_subsets && it < c->budget; ++it)
            for (int k = 0; k < a.nmodels[it]; ++k) {
                const double med = a.medians[it * 3 + k];
                if (med < c->min_median) { c->min_median = med; c->best_h = it; c->best_k = k; }
            }
    } else {
        for (; it < h1 && it < c->n_subsets && it < c->budget; ++it)
            for (int k = 0; k < a.nmodels[it]; ++k) {
                const int good = a.counts[it * 3 + k];
                if (good > max(c->max_good, RS_MP - 1)) {
                    c->max_good = good;
                    c->best_h = it; c->best_k = k;
                    c->budget = update_num_iters(a.confidence, (double)(n - good) / n, RS_MP, c->budget);
                }
            }
    }
    c->niters = it;
}

// ---- first launch: (device entry) the matched pixel pairs from the matcher's index pairs, the control block, the samples
// of the first chunk.  The points go to LDS as they are gathered (and to `p1` / `p2` for the launches that follow).
__global__ __launch_bounds__(RS_T) void rs_head_kernel(RSArgs a, float* p1w, float* p2w) {
    extern __shared__ __attribute__((aligned(16))) float rs_pts[];
    __shared__ int go;
    RSCtrl* c = a.ctrl;
    const int n = rs_n(a);
    const bool gather = a.ij != nullptr, in_lds = n <= RS_LDS_POINTS;
    if (gather)
        for (int i = threadIdx.x; i < n; i += RS_T) {
            const int q = a.ij[2 * i], t = a.ij[2 * i + 1];
            const float x1 = a.xy1[2 * q], y1 = a.xy1[2 * q + 1], x2 = a.xy2[2 * t], y2 = a.xy2[2 * t + 1];
            p1w[2 * i] = x1; p1w[2 * i + 1] = y1; p2w[2 * i] = x2; p2w[2 * i + 1] = y2;
            if (in_lds) { rs_pts[2 * i] = x1; rs_pts[2 * i + 1] = y1; rs_pts[2 * n + 2 * i] = x2; rs_pts[2 * n + 2 * i + 1] = y2; }
        }
    if (threadIdx.x == 0) {
        c->lmeds = n <= RS_LMEDS_MAX;
        c->budget = max(a.max_iters, 1);
        if (c->lmeds) c->budget = max(update_num_iters(a.confidence, 0.45, RS_MP, a.max_iters), 1);   // LMeDS: fixed budget
        c->rng_state = 0xffffffffffffffffULL;
        c->n_subsets = 0; c->exhausted = 0;
        c->best_h = c->best_k = -1;
        c->best_count = 0; c->niters = 0; c->max_good = 0;
        c->min_median = DBL_MAX;
        if (n < 8) c->exhausted = 1;          // _dev entry with a device count: fewer than 8 matches pass through unfiltered
    }
    __syncthreads();                          // (the control block is read back by this workgroup only: same CU, no other cache)
    // n > RS_LDS_POINTS with a gather: the sampler would read p1 / p2 from memory this workgroup has just written - the host
    // entry point launches rs_gather_kernel first in that case and passes ij = NULL here
    rs_subsets_step(a, a.h0, a.h1, rs_pts, gather && in_lds, &go);
}

// the separate gather of the sizes rs_head_kernel does not take (more matches than RS_LDS_POINTS)
__global__ __launch_bounds__(RS_T) void rs_gather_kernel(RSArgs a, float* p1, float* p2) {
    const int n = rs_n(a);
    for (int i = blockIdx.x * RS_T + threadIdx.x; i < n; i += gridDim.x * RS_T) {
        const int q = a.ij[2 * i], t = a.ij[2 * i + 1];
        p1[2 * i] = a.xy1[2 * q]; p1[2 * i + 1] = a.xy1[2 * q + 1];
        p2[2 * i] = a.xy2[2 * t]; p2[2 * i + 1] = a.xy2[2 * t + 1];
    }
}

// ---- between chunks: the best / budget replay over the chunk just scored [h0, h1), then the samples of the next [h1, h2)
__global__ __launch_bounds__(RS_T) void rs_step_kernel(RSArgs a, int h2) {
    extern __shared__ __attribute__((aligned(16))) float rs_pts[];
    __shared__ int go;
    if (threadIdx.x == 0) rs_select_step(a, a.h0, a.h1);
    __syncthreads();
    rs_subsets_step(a, a.h1, h2, rs_pts, false, &go);
}

// ---- last launch: the replay over the last chunk, the winner's threshold and matrix, its inlier mask and (device entry) what
// filter_matches_ransac returns (features_utils.py:185-200): the pairs whose mask is set, in order; all of them below 8
// matches; none when OpenCV finds no model (mask None)
__global__ __launch_bounds__(1024) void rs_tail_kernel(RSArgs a) {
    __shared__ int wsum[16], base;
    __shared__ double Fw[9];
    __shared__ float s_t;
    RSCtrl* c = a.ctrl;
    const int n = rs_n(a), lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        rs_select_step(a, a.h0, a.h1);
        c->thresh = a.thresh;
        if (c->lmeds && c->best_h >= 0) {
            double sigma = 2.5 * 1.4826 * (1 + 5. / (n - RS_MP)) * sqrt(c->min_median);
            c->thresh = fmax(sigma, 0.001);
        }
        if (c->best_h >= 0)
            for (int i = 0; i < 9; ++i) { c->F[i] = a.models[(size_t)c->best_h * 27 + 9 * c->best_k + i]; Fw[i] = c->F[i]; }
        s_t = (float)(c->thresh * c->thresh);
        base = 0;
    }
    __syncthreads();
    const bool have = c->best_h >= 0;
    const float t = s_t;
    const bool pass = n < 8;                   // features_utils.py:189-190: fewer than 8 matches are returned as they are
    // inlier mask of the winner, then the order-preserving compaction of the index pairs, 1024 matches per turn
    int good = 0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const int in = pass || (have && fm_error(Fw, a.p1[2 * i], a.p1[2 * i + 1], a.p2[2 * i], a.p2[2 * i + 1]) <= t);
        a.mask[i] = (unsigned char)in;
        good += in;
    }
    {
        for (int o = 32; o > 0; o >>= 1) good += __shfl_xor(good, o);
        if (lane == 0) wsum[w] = good;
        __syncthreads();
        if (threadIdx.x == 0) { int tot = 0; for (int j = 0; j < 16; ++j) tot += wsum[j]; c->best_count = tot; }
        __syncthreads();
    }
    if (!a.ij_out && !a.info_out && !a.F_out) return;         // (host entry: mask and control block are read back as they are)
    const bool none = !pass && (!have || (c->lmeds && c->best_count < RS_MP));
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        const bool keep = i < n && !none && a.mask[i];          // (this thread's own store above)
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) wsum[w] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int j = 0; j < w; ++j) off += wsum[j];
        if (keep && a.ij_out) {
            const int o = off + __popcll(bal & ((1ull << lane) - 1));
            a.ij_out[2 * o] = a.ij[2 * i];
            a.ij_out[2 * o + 1] = a.ij[2 * i + 1];
        }
        __syncthreads();
        if (threadIdx.x == 0) { int tsum = 0; for (int j = 0; j < 16; ++j) tsum += wsum[j]; base += tsum; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (a.info_out) {
            a.info_out[0] = base;                                   // matches kept
            a.info_out[1] = c->niters;
            a.info_out[2] = c->lmeds;
            a.info_out[3] = pass ? -2 : (none ? -1 : c->best_h);   // -2: passed through, -1: no model
        }
        if (a.F_out) for (int i = 0; i < 9; ++i) a.F_out[i] = (!pass && !none) ? c->F[i] : 0.0;
    }
}

// the whole filter (shared by the two entries): 7 launches - head, then per chunk [solve + score], [replay + next samples] -
// against the 17 of the thread / sample + workgroup / model form (r05 kernel trace: 116 us per call on ~600 matches, 40 of them
// the eight early-exit launches of the two chunks a matcher's inlier ratios never reach).
void rs_enqueue(hipStream_t s, RSArgs a, int max_iters, float* p1w, float* p2w) {
    // the sample loop in chunks: a chunk whose first sample lies beyond the (shrinking) budget is two early-exit launches.
    // The sample stream is replayed by ONE lane (cv::RNG is sequential: ~1.4 us per 7-point sample), so the first chunk is
    // small - with the inlier ratios a matcher's output has, OpenCV's budget drops to 3 - 5 after the first all-inlier sample
    // and the 128-sample first chunk of r03 spent 180 us drawing samples the loop never looks at - and the rest are few and
    // large: the result does not depend on the chunking.
    const int bounds[] = {0, std::min(8, max_iters), std::min(128, max_iters), max_iters};
    const size_t pts_lds = (size_t)std::min(a.n, RS_LDS_POINTS) * 16;
    (void)hipGetLastError();     // (a stale error of another library on this thread is not ours)
    if (a.ij && a.n > RS_LDS_POINTS) {      // too many matches for the head kernel's LDS image: gather in a launch of its own
        hipLaunchKernelGGL(rs_gather_kernel, dim3(std::min(sslam::cdiv(a.n, RS_T), 64)), dim3(RS_T), 0, s, a, p1w, p2w);
        RSArgs h = a; h.ij = nullptr; h.h0 = bounds[0]; h.h1 = bounds[1];
        hipLaunchKernelGGL(rs_head_kernel, dim3(1), dim3(RS_T), pts_lds, s, h, p1w, p2w);
    } else {
        RSArgs h = a; h.h0 = bounds[0]; h.h1 = bounds[1];
        hipLaunchKernelGGL(rs_head_kernel, dim3(1), dim3(RS_T), pts_lds, s, h, p1w, p2w);
    }
    for (int ci = 0; ci < 3; ++ci) {
        a.h0 = bounds[ci]; a.h1 = bounds[ci + 1];
        if (a.h1 > a.h0) hipLaunchKernelGGL(rs_models_score_kernel, dim3(a.h1 - a.h0), dim3(RS_T), 0, s, a);
        if (ci < 2) hipLaunchKernelGGL(rs_step_kernel, dim3(1), dim3(RS_T), pts_lds, s, a, bounds[ci + 2]);
    }
    hipLaunchKernelGGL(rs_tail_kernel, dim3(1), dim3(1024), 0, s, a);       // (a.h0, a.h1: the last chunk)
}

struct RSScratch { size_t p1, p2, sub, mod, nm, cnt, med, mask, ctrl, total; };
RSScratch rs_layout(size_t N, size_t H) {
    RSScratch L{};
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = sslam::align_up(off + bytes + 8, 256); return o; };
    L.p1 = carve(N * 8); L.p2 = carve(N * 8); L.sub = carve(H * RS_MP * 4); L.mod = carve(H * 27 * 8);
    L.nm = carve(H * 4); L.cnt = carve(H * 12); L.med = carve(H * 12); L.mask = carve(N); L.ctrl = carve(sizeof(RSCtrl));
    L.total = off;
    return L;
}

int rs_reserve(sslam_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->ba_scratch_bytes) return 0;
    // (re)allocation synchronises the device: a pipeline sizes the scratch once, with its largest problem
    if (ctx->ba_scratch) SSLAM_HIP_CHECK(hipFree(ctx->ba_scratch));
    ctx->ba_scratch = nullptr;
    ctx->ba_scratch_bytes = 0;
    SSLAM_HIP_CHECK(hipMalloc(&ctx->ba_scratch, bytes));
    ctx->ba_scratch_bytes = bytes;
    return 0;
}

}  // namespace

extern "C" int sslam_fmat_ransac_dev(sslam_ctx* ctx, int n_max, const int32_t* n_dev, const float* xy1_dev,
                                     const float* xy2_dev, const int32_t* ij_dev, double thresh, double confidence,
                                     int max_iters, unsigned char* mask_out_dev, int32_t* ij_out_dev,
                                     double* F_out_dev, int32_t* info_out_dev) {
    SSLAM_REQUIRE(ctx != nullptr, "sslam_fmat_ransac_dev: ctx is NULL");
    SSLAM_REQUIRE(n_max >= 1, "sslam_fmat_ransac_dev: n_max %d < 1", n_max);
    SSLAM_REQUIRE(xy1_dev && xy2_dev && ij_dev, "sslam_fmat_ransac_dev: NULL argument");
    if (thresh <= 0) thresh = 3;
    if (confidence < DBL_EPSILON || confidence > 1 - DBL_EPSILON) confidence = 0.99;
    if (max_iters <= 0 || max_iters > RS_MAX_ITERS) max_iters = RS_MAX_ITERS;
    SSLAM_HIP_CHECK(hipSetDevice(ctx->device));
    const RSScratch L = rs_layout((size_t)n_max, (size_t)max_iters);
    if (int rc = rs_reserve(ctx, L.total)) return rc;
    char* b = (char*)ctx->ba_scratch;
    hipStream_t s = ctx->stream;
    RSArgs a{};
    a.n = n_max; a.n_dev = n_dev; a.max_iters = max_iters; a.thresh = thresh; a.confidence = confidence;
    a.xy1 = xy1_dev; a.xy2 = xy2_dev; a.ij = ij_dev; a.ij_out = ij_out_dev; a.info_out = info_out_dev; a.F_out = F_out_dev;
    a.p1 = (const float*)(b + L.p1); a.p2 = (const float*)(b + L.p2);
    a.subsets = (int*)(b + L.sub); a.models = (double*)(b + L.mod); a.nmodels = (int*)(b + L.nm);
    a.counts = (int*)(b + L.cnt); a.medians = (float*)(b + L.med);
    a.mask = mask_out_dev ? mask_out_dev : (unsigned char*)(b + L.mask);
    a.ctrl = (RSCtrl*)(b + L.ctrl);
    rs_enqueue(s, a, max_iters, (float*)(b + L.p1), (float*)(b + L.p2));
    SSLAM_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int sslam_fmat_ransac_host(sslam_ctx* ctx, int n, const float* pts1, const float* pts2, double thresh,
                                      double confidence, int max_iters, unsigned char* mask_out, double* F_out,
                                      int* info_out) {
    SSLAM_REQUIRE(ctx != nullptr, "sslam_fmat_ransac_host: ctx is NULL");
    SSLAM_REQUIRE(n >= 8, "sslam_fmat_ransac_host: %d matches, need >= 8 (the caller returns fewer unchanged)", n);
    SSLAM_REQUIRE(pts1 && pts2 && mask_out, "sslam_fmat_ransac_host: NULL argument");
    // cv::findFundamentalMat's own defaulting of bad parameters
    if (thresh <= 0) thresh = 3;
    if (confidence < DBL_EPSILON || confidence > 1 - DBL_EPSILON) confidence = 0.99;
    if (max_iters <= 0 || max_iters > RS_MAX_ITERS) max_iters = RS_MAX_ITERS;
    SSLAM_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t N = (size_t)n;
    const RSScratch L = rs_layout(N, (size_t)max_iters);
    if (int rc = rs_reserve(ctx, L.total)) return rc;
    char* b = (char*)ctx->ba_scratch;
    const size_t o_p1 = L.p1, o_p2 = L.p2, o_mask = L.mask, o_ctrl = L.ctrl;
    hipStream_t s = ctx->stream;
    SSLAM_HIP_CHECK(hipMemcpyAsync(b + o_p1, pts1, N * 8, hipMemcpyHostToDevice, s));
    SSLAM_HIP_CHECK(hipMemcpyAsync(b + o_p2, pts2, N * 8, hipMemcpyHostToDevice, s));
    RSArgs a{};
    a.n = n; a.max_iters = max_iters; a.thresh = thresh; a.confidence = confidence;
    a.p1 = (const float*)(b + o_p1); a.p2 = (const float*)(b + o_p2);
    a.subsets = (int*)(b + L.sub); a.models = (double*)(b + L.mod); a.nmodels = (int*)(b + L.nm);
    a.counts = (int*)(b + L.cnt); a.medians = (float*)(b + L.med); a.mask = (unsigned char*)(b + o_mask);
    a.ctrl = (RSCtrl*)(b + o_ctrl);
    rs_enqueue(s, a, max_iters, nullptr, nullptr);
    SSLAM_HIP_CHECK(hipGetLastError());
    RSCtrl h{};
    SSLAM_HIP_CHECK(hipMemcpyAsync(&h, b + o_ctrl, sizeof(RSCtrl), hipMemcpyDeviceToHost, s));
    SSLAM_HIP_CHECK(hipMemcpyAsync(mask_out, b + o_mask, N, hipMemcpyDeviceToHost, s));
    SSLAM_HIP_CHECK(hipStreamSynchronize(s));
    if (F_out) for (int i = 0; i < 9; ++i) F_out[i] = h.best_h >= 0 ? h.F[i] : 0.0;
    if (info_out) {
        info_out[0] = h.best_h >= 0 ? h.best_count : -1;      // -1: no model (cv2 returns mask None)
        info_out[1] = h.niters;
        info_out[2] = h.lmeds;
        info_out[3] = h.best_h;
    }
    return 0;
}
