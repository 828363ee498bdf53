// lightglue_kernels.hip - LightGlue(features='aliked') forward on gfx950.
//
// Replaces `matcher({...})` at slam/core/features_utils.py:157-162 (the
// cvg/LightGlue forward) plus the confidence filter at :164-169.
//
// Numeric type: fp32 RESULTS, two selectable arithmetic paths (sslam_lightglue_set_precision):
//   0  every contraction on the exact-fp32 matrix-core instruction v_mfma_f32_32x32x2_f32 (differs from a
//      torch-CPU fp32 run only by summation order; 157.3 TFLOP/s peak);
//   1  (default) split precision: an fp32 operand is two fp16 planes a = hi + lo 2^-11, a product is three
//      v_mfma_f32_32x32x16_f16 into fp32 accumulators (gemm_f16x3.hpp); the roofline is the dense f16 MFMA
//      peak (2.5 PFLOP/s), of which an algorithmic product can reach at most a third.  The final
//      assignment (final_proj, similarity, dual softmax, arg-max) stays on the exact-fp32 path in both modes.
//
// BATCH.  One enqueue processes up to NB pairs at once (`sslam_lightglue_match_batch_dev`): pair p
// owns images 2p (query side) and 2p+1; every token buffer is image-major over NI = 2 NB images and
// every launch covers all pairs, so a launch fills the 256 CUs without splitting the keys of one
// pair and the ~170 launches of a forward are paid once per batch, not once per pair.
//
// Data layout in HBM (per instance, sized for max_kpts = Kc rows per image, NI images):
//   x      [NI][Kc][256]   token states, image-major (image i at row i*Kc)
//   enc    [NI][Kc][32]    cos / sin of the Fourier positional projection
//   q,k,v  [NI][4][Kc][64] head-major, rotary already applied to q,k
//   msg    [NI][Kc][256]   attention context / message
//   hid    [NI][Kc][512]   FFN hidden (single-pair path only: the batched FFN keeps it on chip, ffn_fused.hpp)
//   sim    [NB][Kc][Kc]    final similarity
// Control flow that the reference decides on the host per layer (early stop,
// point pruning) lives in a device-side control block `LGCtrl` PER PAIR; every kernel
// reads its row counts from it, so a batch is a fixed launch sequence with no
// host round trip (graph-capturable).
#include <mutex>
#include <type_traits>
#include "common.hpp"
#include "gemm_f32.hpp"
#include "gemm_f16x3.hpp"
#include "gemm_f16x3_big.hpp"
#include "ffn_fused.hpp"

namespace {

using sslam::f32x16;
using sslam::mfma32;
using sslam::acc_row;
using sslam::GemmSmem;
using sslam::GemmA;
using sslam::gemm_mainloop;
using sslam::half8;
using sslam::half4;
using sslam::SplitPtr;
using sslam::GemmSmemH;
using sslam::GemmAH;
using sslam::gemm_mainloop_h;
using sslam::gemm_mainloop_ring;
using sslam::panel_index;
using sslam::mfma16;
using sslam::split_f32;
using sslam::SPLIT_INV;

constexpr int D = 256;       // descriptor_dim
constexpr int DH = 64;       // head dim
constexpr int NH = 4;        // heads
constexpr int DIN = 128;     // ALIKED descriptor dim
constexpr int NL = 9;        // layers
constexpr int ENC = 32;      // rotary frequencies per token

struct LGCtrl {
    int n[2];        // current (possibly pruned) token count per image
    int n_prev[2];   // count before the last pruning step
    int n_orig[2];   // M, N of the call
    int stop;        // 0 running, 1 stopped early, 2 empty set
    int stop_layer;  // index i of the layer whose log_assignment is used
    int unconf;      // #tokens with confidence < threshold (both images)
    int n_matches;   // K (-1: the split-precision range flag below was raised, the result is invalid)
    int range_overflow;  // a finite |value| >= 65520 reached an fp16 split while this pair was processed (gemm_f16x3.hpp)
    int pad[5];
};
constexpr int MAX_PAIRS = 16;    // batch capacity bound (kernel-argument tables are sized for it)

__device__ __forceinline__ LGCtrl& ctrl_of(LGCtrl* c, int img) { return c[img >> 1]; }
__device__ __forceinline__ const LGCtrl& ctrl_of(const LGCtrl* c, int img) { return c[img >> 1]; }
__device__ __forceinline__ int n_of(const LGCtrl* c, int img) { return c[img >> 1].n[img & 1]; }
// the range flag of the pair an image belongs to (the one word of the control block that kernels which
// otherwise only read it may write)
__device__ __forceinline__ int* range_flag_of(const LGCtrl* c, int img) {
    return &const_cast<LGCtrl*>(c)[img >> 1].range_overflow;
}

// Per-image input sources of one call (host-built table passed by value): where the keypoints /
// descriptors of image i live, the host-side bound on their count and, optionally, the device
// count written by the extractor.
struct StageSrc {
    const float* xy[2 * MAX_PAIRS];
    const float* desc[2 * MAX_PAIRS];
    const int32_t* cnt[2 * MAX_PAIRS];
    int bound[2 * MAX_PAIRS];
    float size_w[2 * MAX_PAIRS], size_h[2 * MAX_PAIRS];   // 'image_size' of the features (W, H); 0 = none: bounding box
};

// ------------------------------------------------------------------------ //
//  0. prepare: bbox-normalise keypoints, rotary tables, control block
//     (lightglue.py normalize_keypoints(size=None), LearnableFourierPositionalEncoding)
// ------------------------------------------------------------------------ //
__global__ __launch_bounds__(1024) void lg_prepare_kernel(
    StageSrc src, int Kc, float* __restrict__ in_xy /*[NI][Kc][2]*/, float* __restrict__ in_desc /*[NI][Kc][128]*/,
    float* __restrict__ bbox /*[NI][4]: shift x, shift y, scale, -*/, int* __restrict__ ind,
    int* __restrict__ prune, LGCtrl* __restrict__ ctrl) {
    const int img = blockIdx.x, side = img & 1;
    const float* xy = src.xy[img];
    // device-resident counts (written by the extractor) are clamped to the host-side bound
    int n = src.bound[img], n_other = src.bound[img ^ 1];
    if (src.cnt[img]) n = min(max(src.cnt[img][0], 0), n);
    if (src.cnt[img ^ 1]) n_other = min(max(src.cnt[img ^ 1][0], 0), n_other);
    __shared__ float red[4][32];
    {   // descriptors into the staging rows the input projection reads (16-byte pieces): 1 MB per image at 2048
        // keypoints - one workgroup moved it at a single CU's share of the memory system (28 of the kernel's 45 us), so the
        // copy is striped over gridDim.y workgroups; workgroup y = 0 does the rest of the preparation
        const float4* sd = reinterpret_cast<const float4*>(src.desc[img]);
        float4* dd = reinterpret_cast<float4*>(in_desc + (size_t)img * Kc * DIN);
        const int total = n * (DIN / 4), per = (total + (int)gridDim.y - 1) / (int)gridDim.y;
        const int lo = (int)blockIdx.y * per, hi = min(total, lo + per);
        for (int i = lo + (int)threadIdx.x; i < hi; i += blockDim.x) dd[i] = sd[i];
    }
    if (blockIdx.y != 0) return;
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    float* dxy = in_xy + (size_t)img * Kc * 2;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float x = xy[2 * i], y = xy[2 * i + 1];
        dxy[2 * i] = x; dxy[2 * i + 1] = y;
        mnx = fminf(mnx, x); mxx = fmaxf(mxx, x);
        mny = fminf(mny, y); mxy = fmaxf(mxy, y);
    }
    for (int o = 32; o > 0; o >>= 1) {
        mnx = fminf(mnx, __shfl_xor(mnx, o)); mxx = fmaxf(mxx, __shfl_xor(mxx, o));
        mny = fminf(mny, __shfl_xor(mny, o)); mxy = fmaxf(mxy, __shfl_xor(mxy, o));
    }
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) { red[0][w] = mnx; red[1][w] = mxx; red[2][w] = mny; red[3][w] = mxy; }
    __syncthreads();
    const int nw = blockDim.x >> 6;
    mnx = red[0][0]; mxx = red[1][0]; mny = red[2][0]; mxy = red[3][0];
    for (int i = 1; i < nw; ++i) {
        mnx = fminf(mnx, red[0][i]); mxx = fmaxf(mxx, red[1][i]);
        mny = fminf(mny, red[2][i]); mxy = fmaxf(mxy, red[3][i]);
    }
    // size = 1 + max - min (or the image's (W, H) when the features carry 'image_size': the legacy pair entry,
    // features_utils.py:233-247) ; shift = size / 2 ; scale = max(size) / 2
    const bool sized = src.size_w[img] > 0.0f && src.size_h[img] > 0.0f;
    const float sx = sized ? src.size_w[img] : 1.0f + mxx - mnx, sy = sized ? src.size_h[img] : 1.0f + mxy - mny;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        ind[img * Kc + i] = i;
        prune[img * Kc + i] = 1;
    }
    if (threadIdx.x == 0) {
        LGCtrl& c = ctrl_of(ctrl, img);
        bbox[img * 4 + 0] = sx / 2.0f; bbox[img * 4 + 1] = sy / 2.0f; bbox[img * 4 + 2] = fmaxf(sx, sy) / 2.0f;
        c.n[side] = n; c.n_prev[side] = n; c.n_orig[side] = n;
        if (side == 0) { c.stop = (n == 0 || n_other == 0) ? 2 : 0; c.stop_layer = NL - 1;
                         c.unconf = 0; c.n_matches = 0; c.range_overflow = 0; }
    }
}

// rotary tables: cos / sin of Wr . normalised keypoint (one thread per (token, frequency))
__global__ __launch_bounds__(256) void lg_posenc_kernel(const float* __restrict__ in_xy,
                                                        const float* __restrict__ bbox, const float* __restrict__ Wr,
                                                        float* __restrict__ enc_cos, float* __restrict__ enc_sin,
                                                        int Kc, int NI, const LGCtrl* __restrict__ ctrl) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int img = i / (Kc * ENC), rem = i % (Kc * ENC), tok = rem / ENC, f = rem % ENC;
    if (img >= NI || tok >= n_of(ctrl, img)) return;
    const float* xy = in_xy + (size_t)img * Kc * 2;
    const float kx = (xy[2 * tok] - bbox[img * 4 + 0]) / bbox[img * 4 + 2];
    const float ky = (xy[2 * tok + 1] - bbox[img * 4 + 1]) / bbox[img * 4 + 2];
    const float proj = kx * Wr[2 * f] + ky * Wr[2 * f + 1];
    enc_cos[((size_t)img * Kc + tok) * ENC + f] = cosf(proj);
    enc_sin[((size_t)img * Kc + tok) * ENC + f] = sinf(proj);
}

// Row-block -> (image, first row) for the two-image token buffers.
struct RowDom {
    int img, row0, n;
};
template <int BM>
__device__ __forceinline__ RowDom row_domain(const LGCtrl* ctrl, int Kc) {
    const int nb = (Kc + BM - 1) / BM;
    RowDom d;
    d.img = blockIdx.y / nb;
    d.row0 = (blockIdx.y % nb) * BM;
    d.n = n_of(ctrl, d.img);
    return d;
}

// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (id % 8 picks the XCD,
// each with a private 4 MiB L2 that does not survive a kernel boundary).  Tiles that share an
// operand panel (all column tiles of one row block; all query blocks of one K/V slab) are given
// ids that are congruent mod 8 and adjacent in dispatch order, so the panel misses L2 once per
// XCD instead of once per tile.  Placement only changes speed, never results.
__device__ __forceinline__ void xcd_tile(int n_inner, int& outer, int& inner) {
    // linear id -> (outer, inner) with all `inner` of one `outer` on the same XCD;
    // requires gridDim.y (outer count) % 8 == 0, otherwise identity
    const int b = blockIdx.y * gridDim.x + blockIdx.x;
    if ((gridDim.y & 7) != 0) { outer = blockIdx.y; inner = blockIdx.x; return; }
    const int xcd = b & 7, idx = b >> 3;
    outer = xcd + 8 * (idx / n_inner);
    inner = idx % n_inner;
}

enum { EPI_PLAIN = 0, EPI_RESID = 1, EPI_QKV = 2, EPI_CROSSQKV = 3 };

struct LinearArgs {
    const float* A0; int lda0; const float* A1; int lda1; int K0; int K;   // per-image strides = Kc*ld
    const float* W; const float* bias; int N;
    long w_layer_stride;  // floats between consecutive layers' W (used with by_stop_layer)
    long b_layer_stride;
    int by_stop_layer;    // select W/bias by ctrl->stop_layer (log_assignment[i])
    float out_scale;      // multiplies (acc + bias)
    float* out; int ldo;  // PLAIN / RESID destination (per-image stride Kc*ldo); RESID adds `out` itself
    float* q; float* k; float* v;           // QKV destinations [2][4][Kc][64]
    const float* enc_cos; const float* enc_sin;
    const LGCtrl* ctrl; int Kc; int NI;
    int ignore_stop;      // final_proj runs after the stop
};

template <int BM, int BN, int TM, int TN, int EPI>
__global__ __launch_bounds__(256) void lg_linear_kernel(LinearArgs p) {
    __shared__ GemmSmem<BM, BN> sm;
    const RowDom rd = row_domain<BM>(p.ctrl, p.Kc);
    const LGCtrl& pc = ctrl_of(p.ctrl, rd.img);
    if (pc.stop && !p.ignore_stop) return;
    if (pc.stop == 2) return;
    if (rd.row0 >= rd.n) return;
    const int col0 = blockIdx.x * BN;
    const size_t ibase = (size_t)rd.img * p.Kc;
    GemmA ga{p.A0 + ibase * p.lda0, p.lda0, p.A1 ? p.A1 + ibase * p.lda1 : nullptr, p.lda1, p.K0};
    const float* W = p.W;
    const float* bias = p.bias;
    if (p.by_stop_layer) {
        W += (size_t)pc.stop_layer * p.w_layer_stride;
        bias += (size_t)pc.stop_layer * p.b_layer_stride;
    }
    f32x16 acc[TM][TN];
    gemm_mainloop<BM, BN, TM, TN>(ga, W, p.K, p.K, rd.row0, p.Kc, col0, p.N, sm, acc);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = col0 + wn * 32 * TN + j * 32 + (lane & 31);
            const float b = bias ? bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rd.row0 + wm * 32 * TM + i * 32 + acc_row(r, lane);
                float val = (acc[i][j][r] + b) * p.out_scale;
                if constexpr (EPI == EPI_QKV) {
                    // col order is [s][head][d] (weights.py _qkv_row_perm); rotary on q,k:
                    // out[2i] = x[2i] cos_i - x[2i+1] sin_i ; out[2i+1] = x[2i+1] cos_i + x[2i] sin_i
                    const int s = col >> 8, hd = (col >> 6) & 3, d = col & 63;
                    const float partner = __shfl_xor(val, 1);
                    const int rr = min(row, p.Kc - 1);
                    if (s < 2) {
                        const float c = p.enc_cos[(ibase + rr) * ENC + (d >> 1)];
                        const float sn = p.enc_sin[(ibase + rr) * ENC + (d >> 1)];
                        val = (d & 1) ? (val * c + partner * sn) : (val * c - partner * sn);
                    }
                    float* dst = s == 0 ? p.q : (s == 1 ? p.k : p.v);
                    if (row < rd.n) dst[(((size_t)rd.img * NH + hd) * p.Kc + row) * DH + d] = val;
                } else if constexpr (EPI == EPI_CROSSQKV) {
                    const int s = col >> 8, hd = (col >> 6) & 3, d = col & 63;   // s: 0 = qk, 1 = v
                    float* dst = s == 0 ? p.q : p.v;
                    if (row < rd.n) dst[(((size_t)rd.img * NH + hd) * p.Kc + row) * DH + d] = val;
                } else {
                    if (row < rd.n) {
                        float* o = p.out + (ibase + row) * p.ldo + col;
                        if constexpr (EPI == EPI_RESID) val += *o;
                        *o = val;
                    }
                }
            }
        }
}

// ------------------------------------------------------------------------ //
//  2. LayerNorm(512) + exact GELU, in place on the FFN hidden (one wave / row)
// ------------------------------------------------------------------------ //
__global__ __launch_bounds__(256) void lg_ln_gelu_kernel(float* __restrict__ hid,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta,
                                                         const LGCtrl* __restrict__ ctrl, int Kc, int NI) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);     // global wave = row over all images
    const int img = gw / Kc, row = gw % Kc;
    if (img >= NI || ctrl_of(ctrl, img).stop || row >= n_of(ctrl, img)) return;
    float* p = hid + ((size_t)img * Kc + row) * 512;
    float4 a = *reinterpret_cast<float4*>(p + lane * 4);
    float4 b = *reinterpret_cast<float4*>(p + 256 + lane * 4);
    float s = (a.x + a.y) + (a.z + a.w) + (b.x + b.y) + (b.z + b.w);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / 512.0f;
    float v[8] = {a.x - mean, a.y - mean, a.z - mean, a.w - mean, b.x - mean, b.y - mean, b.z - mean, b.w - mean};
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) q += v[i] * v[i];
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / 512.0f + 1e-5f);
    float out[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = (i < 4 ? 0 : 256) + lane * 4 + (i & 3);
        const float y = v[i] * rstd * gamma[c] + beta[c];
        out[i] = 0.5f * y * (1.0f + erff(y * 0.70710678118654752440f));
    }
    *reinterpret_cast<float4*>(p + lane * 4) = make_float4(out[0], out[1], out[2], out[3]);
    *reinterpret_cast<float4*>(p + 256 + lane * 4) = make_float4(out[4], out[5], out[6], out[7]);
}

// ------------------------------------------------------------------------ //
//  3. Attention (flash-style, fp32 MFMA).  One wave = 32 query rows; block =
//     4 waves = 128 queries of one (image, head); grid.z splits the keys.
//     S^T = K.Q^T so a lane owns one query column: softmax runs over registers
//     (+ one cross-half exchange) and P feeds PV straight from the accumulator.
// ------------------------------------------------------------------------ //
constexpr int AQ = 128;          // queries per block
constexpr int AK = 64;           // keys per LDS tile
constexpr int AK_LD = DH + 4;

struct __attribute__((aligned(16))) AttnSmem {
    float k[2][AK * AK_LD];
    float v[2][AK * DH];
};

struct AttnArgs {
    const float* Q; const float* K; const float* V;   // [NI][4][Kc][64]
    int cross;                                        // keys/values come from the other image of the pair
    float* o_part; float* m_part; float* l_part;      // [KS][NIc][4][Kc][64] / [KS][NIc][4][Kc]
    int KS; int Kc; int NIc;                          // NIc: image capacity of the instance (partial strides)
    const LGCtrl* ctrl;
};

__global__ __launch_bounds__(256) void lg_attention_kernel(AttnArgs p) {
    __shared__ AttnSmem sm;
    const int img = blockIdx.y >> 2, head = blockIdx.y & 3;
    if (ctrl_of(p.ctrl, img).stop) return;
    const int kimg = p.cross ? (img ^ 1) : img;
    const int nq = n_of(p.ctrl, img), nk = n_of(p.ctrl, kimg);
    const int q0 = blockIdx.x * AQ;
    if (q0 >= nq) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5, lr = lane & 31;
    const int z = blockIdx.z;
    const int ntiles = (nk + AK - 1) / AK;
    const int t0 = (int)((long)z * ntiles / p.KS), t1 = (int)((long)(z + 1) * ntiles / p.KS);

    const float* Qb = p.Q + ((size_t)img * NH + head) * p.Kc * DH;
    const float* Kb = p.K + ((size_t)kimg * NH + head) * p.Kc * DH;
    const float* Vb = p.V + ((size_t)kimg * NH + head) * p.Kc * DH;

    // Q fragment: B operand of S^T = K.Q^T -> lane holds Q[i = lr][dims 8g+4h .. +3], pre-scaled
    // by 1/sqrt(64) * log2(e) so the softmax uses exp2.
    const float qscale = 0.125f * 1.4426950408889634f;
    const int qi = min(q0 + wave * 32 + lr, p.Kc - 1);
    float qreg[32];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const float4 f = *reinterpret_cast<const float4*>(Qb + (size_t)qi * DH + g * 8 + 4 * h);
        qreg[g * 4 + 0] = f.x * qscale; qreg[g * 4 + 1] = f.y * qscale;
        qreg[g * 4 + 2] = f.z * qscale; qreg[g * 4 + 3] = f.w * qscale;
    }

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.0f; o1[r] = 0.0f; }
    float m_run = -INFINITY, l_run = 0.0f;

    float4 rk0, rk1, rk2, rk3, rv0, rv1, rv2, rv3;      // named: see gemm_mainloop
    rk0 = rk1 = rk2 = rk3 = rv0 = rv1 = rv2 = rv3 = make_float4(0, 0, 0, 0);
    const int ar = t >> 4, ac4 = (t & 15) * 4;       // (row, dim offset) inside a 16-row slab
#define ATTN_GLOAD(tile_)                                                                   \
    {                                                                                       \
        const int rb_ = (tile_) * AK + ar;                                                  \
        const size_t o0_ = (size_t)min(rb_, p.Kc - 1) * DH + ac4;                           \
        const size_t o1_ = (size_t)min(rb_ + 16, p.Kc - 1) * DH + ac4;                      \
        const size_t o2_ = (size_t)min(rb_ + 32, p.Kc - 1) * DH + ac4;                      \
        const size_t o3_ = (size_t)min(rb_ + 48, p.Kc - 1) * DH + ac4;                      \
        LD4(rk0, Kb + o0_); LD4(rk1, Kb + o1_); LD4(rk2, Kb + o2_); LD4(rk3, Kb + o3_);     \
        LD4(rv0, Vb + o0_); LD4(rv1, Vb + o1_); LD4(rv2, Vb + o2_); LD4(rv3, Vb + o3_);     \
        /* keys >= nk get P = 0; keep their (stale) V rows out: 0 x non-finite is not 0 */  \
        if (rb_ >= nk) rv0 = make_float4(0, 0, 0, 0);                                       \
        if (rb_ + 16 >= nk) rv1 = make_float4(0, 0, 0, 0);                                  \
        if (rb_ + 32 >= nk) rv2 = make_float4(0, 0, 0, 0);                                  \
        if (rb_ + 48 >= nk) rv3 = make_float4(0, 0, 0, 0);                                  \
    }
#define ATTN_SSTORE(buf_)                                                                   \
    {                                                                                       \
        float* dk_ = &sm.k[buf_][ar * AK_LD + ac4];                                         \
        float* dv_ = &sm.v[buf_][ar * DH + ac4];                                            \
        ST4(dk_, rk0); ST4(dk_ + 16 * AK_LD, rk1); ST4(dk_ + 32 * AK_LD, rk2);              \
        ST4(dk_ + 48 * AK_LD, rk3);                                                         \
        ST4(dv_, rv0); ST4(dv_ + 16 * DH, rv1); ST4(dv_ + 32 * DH, rv2); ST4(dv_ + 48 * DH, rv3); \
    }

    if (t0 < t1) {
        ATTN_GLOAD(t0);
        ATTN_SSTORE(0);
    }
    __syncthreads();
    int cur = 0;
    for (int tile = t0; tile < t1; ++tile) {
        if (tile + 1 < t1) ATTN_GLOAD(tile + 1);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.0f;
            const float* sk = sm.k[cur] + (sub * 32 + lr) * AK_LD + 4 * h;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float4 kf = *reinterpret_cast<const float4*>(sk + g * 8);
                s = mfma32(kf.x, qreg[g * 4 + 0], s);
                s = mfma32(kf.y, qreg[g * 4 + 1], s);
                s = mfma32(kf.z, qreg[g * 4 + 2], s);
                s = mfma32(kf.w, qreg[g * 4 + 3], s);
            }
            // mask keys beyond nk, tile max
            const int kbase = tile * AK + sub * 32;
            float tmax = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (kbase + acc_row(r, lane) >= nk) s[r] = -INFINITY;
                tmax = fmaxf(tmax, s[r]);
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
            const float m_new = fmaxf(m_run, tmax);
            // (m_new is finite: every block's first sub-tile holds at least one valid key)
            const float alpha = exp2f(m_run - m_new);
            m_run = m_new;
            float psum = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = exp2f(s[r] - m_new);
                psum += s[r];
            }
            l_run = l_run * alpha + psum;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            // O^T[d][i] += V^T[d][j] P^T[j][i] ; acc reg r of S^T is key row acc_row(r) -> k index
            const float* sv = sm.v[cur] + (size_t)(sub * 32) * DH + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = (r & 3) + 8 * (r >> 2) + 4 * h;
                o0 = mfma32(sv[j * DH], s[r], o0);
                o1 = mfma32(sv[j * DH + 32], s[r], o1);
            }
        }
        if (tile + 1 < t1) ATTN_SSTORE(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
#undef ATTN_GLOAD
#undef ATTN_SSTORE

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const int qrow = q0 + wave * 32 + lr;
    if (qrow < nq) {
        const size_t pbase = (((size_t)z * p.NIc + img) * NH + head) * p.Kc + qrow;
        float* op = p.o_part + pbase * DH;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            *reinterpret_cast<float4*>(op + 8 * g4 + 4 * h) =
                make_float4(o0[4 * g4], o0[4 * g4 + 1], o0[4 * g4 + 2], o0[4 * g4 + 3]);
            *reinterpret_cast<float4*>(op + 32 + 8 * g4 + 4 * h) =
                make_float4(o1[4 * g4], o1[4 * g4 + 1], o1[4 * g4 + 2], o1[4 * g4 + 3]);
        }
        if (h == 0) { p.m_part[pbase] = m_run; p.l_part[pbase] = l_tot; }
    }
}

// merge key-split partials -> msg[img][row][head*64 + d]
__global__ __launch_bounds__(256) void lg_attn_merge_kernel(const float* __restrict__ o_part,
                                                            const float* __restrict__ m_part,
                                                            const float* __restrict__ l_part,
                                                            float* __restrict__ msg, int KS, int Kc, int NI,
                                                            int NIc, const LGCtrl* __restrict__ ctrl) {
    // one thread = one float4 of one (img, head, row)
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c4 = (int)(gid & 15);
    const long rid = gid >> 4;                      // (img*4+head)*Kc + row
    if (rid >= (long)NI * NH * Kc) return;
    const int row = (int)(rid % Kc), ih = (int)(rid / Kc), img = ih >> 2, head = ih & 3;
    if (ctrl_of(ctrl, img).stop || row >= n_of(ctrl, img)) return;
    const size_t zs = (size_t)NIc * NH * Kc;        // partial stride of one key slab
    float M = -INFINITY;
    for (int z = 0; z < KS; ++z) M = fmaxf(M, m_part[(size_t)z * zs + rid]);
    float4 acc = make_float4(0, 0, 0, 0);
    float L = 0.0f;
    for (int z = 0; z < KS; ++z) {
        const size_t pb = (size_t)z * zs + rid;
        const float mz = m_part[pb];
        const float wz = (mz == -INFINITY) ? 0.0f : exp2f(mz - M);
        const float4 o = *reinterpret_cast<const float4*>(o_part + pb * DH + c4 * 4);
        acc.x += o.x * wz; acc.y += o.y * wz; acc.z += o.z * wz; acc.w += o.w * wz;
        L += l_part[pb] * wz;
    }
    const float inv = 1.0f / L;
    *reinterpret_cast<float4*>(msg + ((size_t)img * Kc + row) * D + head * DH + c4 * 4) =
        make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}

// ------------------------------------------------------------------------ //
//  4. token confidence + matchability, early-stop decision, point pruning
// ------------------------------------------------------------------------ //
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float logsigmoidf_(float x) {   // min(x,0) - log1p(exp(-|x|))
    return fminf(x, 0.0f) - log1pf(expf(-fabsf(x)));
}

// conf = sigmoid(w_c.x + b_c), mat = w_m.x + b_m per token; a block adds its count of unconfident
// tokens with ONE atomic (same-address atomics retire one per ~11 ns).  The kernel's ~12 us are the
// launch, the dependent read of the control block written by the previous kernel and the atomic:
// a wave-per-16-tokens layout and this lane-per-token one measure the same.

__global__ __launch_bounds__(256) void lg_token_heads_kernel(
    const float* __restrict__ x, const float* __restrict__ wc, const float* __restrict__ bc,
    const float* __restrict__ wm, const float* __restrict__ bm, long m_layer_stride, int use_stop_layer,
    float conf_thr, float* __restrict__ conf, float* __restrict__ mat, LGCtrl* __restrict__ ctrl, int Kc,
    int count_unconf) {
    // one lane per token: the two 256-long dot products run down the lane's own row (weights are
    // wave-uniform scalar loads), so there is no cross-lane reduction at all; a wave-per-token
    // layout spent its time in 12 dependent shuffles per token
    __shared__ int s_unconf[4];
    const int img = blockIdx.y;                              // one image per grid row: a block never
    LGCtrl& pc = ctrl_of(ctrl, img);                         // mixes the counts of two pairs
    if (pc.stop == 2) return;
    if (pc.stop && !use_stop_layer) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (use_stop_layer) {
        wm += (size_t)pc.stop_layer * m_layer_stride;
        bm += (size_t)pc.stop_layer * m_layer_stride;   // both padded to the same stride
    }
    const int row = blockIdx.x * 256 + threadIdx.x;
    const bool live = row < Kc && row < pc.n[img & 1];
    const float* xr = x + ((size_t)img * Kc + (live ? row : 0)) * D;
    float sm0 = 0.0f, sm1 = 0.0f, sc0 = 0.0f, sc1 = 0.0f;
#pragma unroll 8
    for (int k = 0; k < D; k += 8) {
        const float4 a = *reinterpret_cast<const float4*>(xr + k), b = *reinterpret_cast<const float4*>(xr + k + 4);
        sm0 += a.x * wm[k] + a.y * wm[k + 1] + a.z * wm[k + 2] + a.w * wm[k + 3];
        sm1 += b.x * wm[k + 4] + b.y * wm[k + 5] + b.z * wm[k + 6] + b.w * wm[k + 7];
        if (wc) {
            sc0 += a.x * wc[k] + a.y * wc[k + 1] + a.z * wc[k + 2] + a.w * wc[k + 3];
            sc1 += b.x * wc[k + 4] + b.y * wc[k + 5] + b.z * wc[k + 6] + b.w * wc[k + 7];
        }
    }
    bool unconf = false;
    if (live) {
        const float zm = (sm0 + sm1) + bm[0];
        mat[img * Kc + row] = zm;
        // final call (assignment): log sigmoid of the matchability once per token, here - the two arg-max
        // passes evaluated it per (row, column) ELEMENT (log1p + exp on 4 M elements per pair: they ran at
        // 1.1 TB/s of `sim` instead of HBM speed)
        if (use_stop_layer) conf[img * Kc + row] = logsigmoidf_(zm);
        if (wc) {
            const float c = sigmoidf_((sc0 + sc1) + bc[0]);
            conf[img * Kc + row] = c;
            unconf = c < conf_thr;
        }
    }
    if (!count_unconf) return;
    const int wcount = __popcll(__ballot(unconf));
    if (lane == 0) s_unconf[wave] = wcount;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = s_unconf[0] + s_unconf[1] + s_unconf[2] + s_unconf[3];
        if (tot) atomicAdd(&pc.unconf, tot);
    }
}

// single block: early-stop test, then order-preserving compaction maps.  The two images of the pair are compacted side
// by side, 512 threads each (r03: one after the other before - two dependent load / scan / store chains per launch on the
// single-pair path's critical path); a thread parks the old indices of its chunk in LDS with the keep flag in bit 15
// (indices < 8192 = sslam_lightglue_create's bound on max_kpts).
constexpr int DECIDE_MAX_KC = 8192;
__global__ __launch_bounds__(1024) void lg_decide_kernel(
    int layer, float conf_thr, float depth_conf, float width_conf, int prune_min, int do_stop,
    const float* __restrict__ conf, const float* __restrict__ mat, int* __restrict__ ind,
    int* __restrict__ gmap, int* __restrict__ prune, LGCtrl* __restrict__ ctrl, int Kc) {
    __shared__ unsigned short s_ind[2][DECIDE_MAX_KC];
    __shared__ int wsum[2][8];
    __shared__ int s_stop;
    const int pair = blockIdx.x;
    ctrl += pair;                                    // this pair's control block and 2 Kc-row slices
    conf += (size_t)pair * 2 * Kc; mat += (size_t)pair * 2 * Kc; ind += (size_t)pair * 2 * Kc;
    gmap += (size_t)pair * 2 * Kc; prune += (size_t)pair * 2 * Kc;
    const int t = threadIdx.x;
    const int img = t >> 9, th = t & 511, lane = th & 63, wave = th >> 6;
    const int per = (Kc + 511) / 512;
    // r05: this one-workgroup kernel is a chain of memory latencies on the single-pair path's critical path (9 us x 8 layers):
    // every thread issues the loads of its first four elements and of the image's row count BEFORE the stop decision is
    // known - they do not depend on it, and a stopped pair simply drops them
    const int stopped = ctrl->stop;
    const int n = ctrl->n[img];
    float mv[4], cv[4]; int iv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = min(th * per + u, Kc - 1);
        mv[u] = mat[img * Kc + i]; cv[u] = do_stop ? conf[img * Kc + i] : 0.0f; iv[u] = ind[img * Kc + i];
    }
    if (stopped) return;
    if (t == 0) {
        int stop = 0;
        if (do_stop) {
            const float ratio = 1.0f - (float)ctrl->unconf / (float)(ctrl->n_orig[0] + ctrl->n_orig[1]);
            stop = ratio > depth_conf;
        }
        ctrl->unconf = 0;
        if (stop) { ctrl->stop = 1; ctrl->stop_layer = layer; }
        s_stop = stop;
        ctrl->n_prev[0] = ctrl->n[0];
        ctrl->n_prev[1] = ctrl->n[1];
    }
    __syncthreads();
    if (s_stop || width_conf <= 0.0f) return;
    const bool act = n > prune_min;                  // (uniform over the image's eight waves)
    // thread owns the contiguous chunk [th*per, th*per+per) of its image
    int cnt = 0;
    if (act) {
        // (r04: the chunk's loads first, four elements at a time - rolled, every element waited for its own three loads)
        for (int j0 = 0; j0 < per; j0 += 4) {
            if (j0 > 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = min(th * per + j0 + u, Kc - 1);
                    mv[u] = mat[img * Kc + i]; cv[u] = do_stop ? conf[img * Kc + i] : 0.0f; iv[u] = ind[img * Kc + i];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = th * per + j0 + u;
                if (j0 + u < per && i < n) {
                    const float sc = sigmoidf_(mv[u]);
                    int k = sc > (1.0f - width_conf);
                    if (do_stop) k |= cv[u] <= conf_thr;
                    s_ind[img][i] = (unsigned short)(iv[u] | (k << 15));
                    cnt += k;
                }
            }
        }
    }
    int incl = cnt;
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[img][wave] = incl;
    __syncthreads();       // (also: all reads of ind[] done before any write)
    if (!act) return;
    int base = 0, total = 0;
    for (int w = 0; w < 8; ++w) {
        if (w < wave) base += wsum[img][w];
        total += wsum[img][w];
    }
    int pos = base + incl - cnt;
    for (int j = 0; j < per; ++j) {
        const int i = th * per + j;
        if (i < n && (s_ind[img][i] & 0x8000)) {
            const int old = s_ind[img][i] & 0x7fff;
            ind[img * Kc + pos] = old;
            gmap[img * Kc + pos] = i;
            prune[img * Kc + old] += 1;
            ++pos;
        }
    }
    if (th == 0) {
        ctrl->n[img] = total;
        if (total == 0) { ctrl->stop = 2; ctrl->stop_layer = layer; }
    }
}

// gather surviving rows: x, enc -> tmp (back == 0), then tmp -> x, enc (back == 1).  On the split-precision path the
// copy back also refreshes the (hi, lo) planes of the rows it moves (xs_hi != nullptr: what a separate
// lg_split_rows_kernel(only_moved) launch did - the same rows, the same split, one launch less per layer)
__global__ __launch_bounds__(256) void lg_gather_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ ec,
                                                        const float* __restrict__ es,
                                                        const int* __restrict__ gmap,
                                                        float* __restrict__ tx, float* __restrict__ tc,
                                                        float* __restrict__ ts,
                                                        const LGCtrl* __restrict__ ctrl, int Kc, int NI, int back,
                                                        _Float16* __restrict__ xs_hi, _Float16* __restrict__ xs_lo,
                                                        int NIc) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int img = gw / Kc, row = gw % Kc;
    if (img >= NI) return;
    const LGCtrl& pc = ctrl_of(ctrl, img);
    if (pc.stop || pc.n[img & 1] == pc.n_prev[img & 1] || row >= pc.n[img & 1]) return;
    const size_t dst = (size_t)img * Kc + row;
    const size_t src = back ? dst : (size_t)img * Kc + gmap[img * Kc + row];
    // back == 1: tmp -> x (plain copy); back == 0: x[gmap] -> tmp
    const float* sx = back ? tx : x; float* dx = back ? const_cast<float*>(x) : tx;
    const float* sc = back ? tc : ec; float* dc = back ? const_cast<float*>(ec) : tc;
    const float* ss = back ? ts : es; float* ds = back ? const_cast<float*>(es) : ts;
    const float4 val = *reinterpret_cast<const float4*>(sx + src * D + lane * 4);
    *reinterpret_cast<float4*>(dx + dst * D + lane * 4) = val;
    if (back && xs_hi) {
        const float v4[4] = {val.x, val.y, val.z, val.w};
        half4 hh, ll;
#pragma unroll
        for (int e = 0; e < 4; ++e) { _Float16 a, b; split_f32(v4[e], a, b, range_flag_of(ctrl, img)); hh[e] = a; ll[e] = b; }
        const size_t o = panel_index(img * Kc + row, lane * 4, NIc * Kc);
        *reinterpret_cast<half4*>(xs_hi + o) = hh;
        *reinterpret_cast<half4*>(xs_lo + o) = ll;
    }
    if (lane < 32) dc[dst * ENC + lane] = sc[src * ENC + lane];
    else ds[dst * ENC + lane - 32] = ss[src * ENC + lane - 32];
}

// ------------------------------------------------------------------------ //
//  5. assignment: sim GEMM, dual log-softmax statistics, arg-max, mutual check
// ------------------------------------------------------------------------ //
struct SimArgs { const float* md; float* sim; int Kc; const LGCtrl* ctrl; };

template <int BM, int BN, int TM, int TN>
__global__ __launch_bounds__(256) void lg_sim_kernel(SimArgs p) {
    __shared__ GemmSmem<BM, BN> sm;
    // XCD-aware order (r04): workgroups go to the 8 XCDs round-robin by their linear id and each XCD has its own L2.  With the
    // pair in the grid's z every XCD fetched every pair's operands (PMC r03: 151 MB fetched for 33.5 MB of inputs); with the
    // pair count a multiple of 8 the pair comes from the low bits of the id instead, so one pair's 4 MB stay in ONE L2.
    int pair = blockIdx.z, bx = blockIdx.x, by = blockIdx.y;
    if ((gridDim.z & 7) == 0) {
        const unsigned T = gridDim.x * gridDim.y;
        const unsigned L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned grp = L / (8 * T), rem = L % (8 * T);
        pair = (int)(grp * 8 + (rem & 7));
        const unsigned tix = rem >> 3;
        bx = (int)(tix % gridDim.x); by = (int)(tix / gridDim.x);
    }
    const LGCtrl& pc = p.ctrl[pair];
    if (pc.stop == 2) return;
    const int n0 = pc.n[0], n1 = pc.n[1];
    const int row0 = by * BM, col0 = bx * BN;
    if (row0 >= n0 || col0 >= n1) return;
    const float* md0 = p.md + (size_t)pair * 2 * p.Kc * D;      // image 2 pair; image 2 pair + 1 follows
    float* simp = p.sim + (size_t)pair * p.Kc * p.Kc;
    GemmA ga{md0, D, nullptr, 0, D};
    f32x16 acc[TM][TN];
    gemm_mainloop<BM, BN, TM, TN>(ga, md0 + (size_t)p.Kc * D, D, D, row0, p.Kc, col0, p.Kc, sm, acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = col0 + wn * 32 * TN + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wm * 32 * TM + i * 32 + acc_row(r, lane);
                if (row < n0 && col < n1) simp[(size_t)row * p.Kc + col] = acc[i][j][r];
            }
        }
}

constexpr int STAT_CACHE = 32;    // values of `sim` a thread keeps between the max and the sum pass (rows / column slabs up to 2048)

// row statistics: one wave per row i: max_j, log(sum_j exp(sim - max))
__global__ __launch_bounds__(256) void lg_row_stats_kernel(const float* __restrict__ sim,
                                                           float* __restrict__ rmax, float* __restrict__ rlog,
                                                           int Kc, const LGCtrl* __restrict__ ctrl) {
    const int pair = blockIdx.y;
    ctrl += pair; sim += (size_t)pair * Kc * Kc; rmax += (size_t)pair * Kc; rlog += (size_t)pair * Kc;
    if (ctrl->stop == 2) return;
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n0 = ctrl->n[0], n1 = ctrl->n[1];
    if (row >= n0) return;
    const float* p = sim + (size_t)row * Kc;
    float m = -INFINITY, s = 0.0f;
    if (n1 <= 64 * STAT_CACHE) {
        // the row fits the wave's registers (32 values per lane): ONE read of `sim`, same per-lane order
        float v[STAT_CACHE];
#pragma unroll
        for (int k = 0; k < STAT_CACHE; ++k) {
            const int j = lane + 64 * k;
            v[k] = j < n1 ? p[j] : -INFINITY;
            m = fmaxf(m, v[k]);
        }
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
#pragma unroll
        for (int k = 0; k < STAT_CACHE; ++k)
            if (lane + 64 * k < n1) s += expf(v[k] - m);
    } else {
        for (int j = lane; j < n1; j += 64) m = fmaxf(m, p[j]);
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        for (int j = lane; j < n1; j += 64) s += expf(p[j] - m);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) { rmax[row] = m; rlog[row] = logf(s); }
}

// column statistics over sim[n0][n1]: grid (n1/64, CSLAB); a block owns 64 columns x one
// slab of rows (4 row-interleaved partials inside the block).  Partials are merged by
// lg_col_stats_merge_kernel (max / rescaled sum), so the row dimension is spread over the chip.
constexpr int CSLAB = 16;

__global__ __launch_bounds__(256) void lg_col_stats_kernel(const float* __restrict__ sim,
                                                           float* __restrict__ pmax, float* __restrict__ psum,
                                                           int Kc, const LGCtrl* __restrict__ ctrl) {
    __shared__ float sh[4][64];
    const int pair = blockIdx.z;
    ctrl += pair; sim += (size_t)pair * Kc * Kc;
    pmax += (size_t)pair * CSLAB * Kc; psum += (size_t)pair * CSLAB * Kc;
    if (ctrl->stop == 2) return;
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int n0 = ctrl->n[0], n1 = ctrl->n[1];
    if (blockIdx.x * 64 >= n1) return;
    const int rows_per = (n0 + CSLAB - 1) / CSLAB;
    const int r0 = blockIdx.y * rows_per, r1 = min(n0, r0 + rows_per);
    const bool ok = col < n1;
    const bool cached = rows_per <= 4 * STAT_CACHE;          // block-uniform: the slab's values stay in registers
    float v[STAT_CACHE];
    float m = -INFINITY;
    if (cached) {
#pragma unroll
        for (int k = 0; k < STAT_CACHE; ++k) {
            const int i = r0 + part + 4 * k;
            v[k] = (ok && i < r1) ? sim[(size_t)i * Kc + col] : -INFINITY;
            m = fmaxf(m, v[k]);
        }
    } else if (ok) {
        for (int i = r0 + part; i < r1; i += 4) m = fmaxf(m, sim[(size_t)i * Kc + col]);
    }
    sh[part][lane] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sh[0][lane], sh[1][lane]), fmaxf(sh[2][lane], sh[3][lane]));
    __syncthreads();
    float s = 0.0f;
    if (ok && m > -INFINITY) {
        if (cached) {
#pragma unroll
            for (int k = 0; k < STAT_CACHE; ++k)
                if (r0 + part + 4 * k < r1) s += expf(v[k] - m);
        } else {
            for (int i = r0 + part; i < r1; i += 4) s += expf(sim[(size_t)i * Kc + col] - m);
        }
    }
    sh[part][lane] = s;
    __syncthreads();
    if (part == 0 && ok) {
        pmax[(size_t)blockIdx.y * Kc + col] = m;
        psum[(size_t)blockIdx.y * Kc + col] = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
    }
}

__global__ __launch_bounds__(256) void lg_col_stats_merge_kernel(const float* __restrict__ pmax,
                                                                 const float* __restrict__ psum,
                                                                 float* __restrict__ cmax, float* __restrict__ clog,
                                                                 int Kc, const LGCtrl* __restrict__ ctrl) {
    const int pair = blockIdx.y;
    ctrl += pair; pmax += (size_t)pair * CSLAB * Kc; psum += (size_t)pair * CSLAB * Kc;
    cmax += (size_t)pair * Kc; clog += (size_t)pair * Kc;
    if (ctrl->stop == 2) return;
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ctrl->n[1]) return;
    float m = -INFINITY;
    for (int z = 0; z < CSLAB; ++z) m = fmaxf(m, pmax[(size_t)z * Kc + col]);
    float s = 0.0f;
    for (int z = 0; z < CSLAB; ++z) {
        const float pm = pmax[(size_t)z * Kc + col];
        if (pm > -INFINITY) s += psum[(size_t)z * Kc + col] * expf(pm - m);
    }
    cmax[col] = m;
    clog[col] = logf(s);
}

// scores[i][j] = ((sim - rmax_i) - rlog_i) + ((sim - cmax_j) - clog_j) + (ls0_i + ls1_j)
__device__ __forceinline__ float score_ij(float s, float rm, float rl, float cm, float cl, float a, float b) {
    return (((s - rm) - rl) + ((s - cm) - cl)) + (a + b);
}

// merge of the CSLAB slab partials of column `col` by the first CSLAB lanes of a wave (one load pair each, then a 16-lane
// butterfly: larger value wins, equal values -> the smaller row, as the sequential scan over the slabs); valid in lane 0
static_assert(CSLAB == 16, "the butterfly below spans 16 lanes");
__device__ __forceinline__ int col_argmax_merge_wave(const float* __restrict__ pval, const int* __restrict__ parg,
                                                     int Kc, int col, int lane) {
    float bv = -INFINITY; int bi = 0x7fffffff;
    if (lane < CSLAB) { bv = pval[(size_t)lane * Kc + col]; bi = parg[(size_t)lane * Kc + col]; }
    for (int o = 8; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    return bi;
}

// Runs AFTER lg_col_argmax_kernel (r05): the wave of row r also merges the slab partials of column r into arg1[r] - the single
// workgroup of lg_emit_kernel did that merge per candidate match, 2 x CSLAB scattered 4-byte loads each, ~65 k of them through
// one CU's L1: most of its 19 us.
__global__ __launch_bounds__(256) void lg_row_argmax_kernel(
    const float* __restrict__ sim, const float* __restrict__ rmax, const float* __restrict__ rlog,
    const float* __restrict__ cmax, const float* __restrict__ clog, const float* __restrict__ z,
    float* __restrict__ best0, int* __restrict__ arg0, const float* __restrict__ pval, const int* __restrict__ parg,
    int* __restrict__ arg1, int Kc, const LGCtrl* __restrict__ ctrl) {
    const int pair = blockIdx.y;
    ctrl += pair; sim += (size_t)pair * Kc * Kc; rmax += (size_t)pair * Kc; rlog += (size_t)pair * Kc;
    cmax += (size_t)pair * Kc; clog += (size_t)pair * Kc; z += (size_t)pair * 2 * Kc;
    best0 += (size_t)pair * Kc; arg0 += (size_t)pair * Kc;
    pval += (size_t)pair * CSLAB * Kc; parg += (size_t)pair * CSLAB * Kc; arg1 += (size_t)pair * Kc;
    if (ctrl->stop == 2) return;
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n0 = ctrl->n[0], n1 = ctrl->n[1];
    if (row < n1 && row < Kc) {                                      // (column `row` of the pair; uniform over the wave)
        const int ci = col_argmax_merge_wave(pval, parg, Kc, row, lane);
        if (lane == 0) arg1[row] = ci;
    }
    if (row >= n0) return;
    const float rm = rmax[row], rl = rlog[row], a = z[row];          // z: log sigmoid(matchability), per token
    float bv = -INFINITY; int bj = 0x7fffffff;
#pragma unroll 8                                 // 32 loads in flight (rolled, the wave sat in s_waitcnt 82 % of its cycles)
    for (int j = lane; j < n1; j += 64) {
        const float v = score_ij(sim[(size_t)row * Kc + j], rm, rl, cmax[j], clog[j], a, z[Kc + j]);
        if (v > bv) { bv = v; bj = j; }          // ascending j per lane: first maximum kept
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o); const int oj = __shfl_xor(bj, o);
        if (ov > bv || (ov == bv && oj < bj)) { bv = ov; bj = oj; }
    }
    if (lane == 0) { best0[row] = bv; arg0[row] = bj; }
}

__global__ __launch_bounds__(256) void lg_col_argmax_kernel(
    const float* __restrict__ sim, const float* __restrict__ rmax, const float* __restrict__ rlog,
    const float* __restrict__ cmax, const float* __restrict__ clog, const float* __restrict__ z,
    float* __restrict__ pval, int* __restrict__ parg, int Kc, const LGCtrl* __restrict__ ctrl) {
    __shared__ float shv[4][64];
    __shared__ int shi[4][64];
    const int pair = blockIdx.z;
    ctrl += pair; sim += (size_t)pair * Kc * Kc; rmax += (size_t)pair * Kc; rlog += (size_t)pair * Kc;
    cmax += (size_t)pair * Kc; clog += (size_t)pair * Kc; z += (size_t)pair * 2 * Kc;
    pval += (size_t)pair * CSLAB * Kc; parg += (size_t)pair * CSLAB * Kc;
    if (ctrl->stop == 2) return;
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int n0 = ctrl->n[0], n1 = ctrl->n[1];
    if (blockIdx.x * 64 >= n1) return;
    const int rows_per = (n0 + CSLAB - 1) / CSLAB;
    const int r0 = blockIdx.y * rows_per, r1 = min(n0, r0 + rows_per);
    float bv = -INFINITY; int bi = 0x7fffffff;
    if (col < n1) {
        const float cm = cmax[col], cl = clog[col], b = z[Kc + col];
#pragma unroll 8
        for (int i = r0 + part; i < r1; i += 4) {
            const float v = score_ij(sim[(size_t)i * Kc + col], rmax[i], rlog[i], cm, cl, z[i], b);
            if (v > bv) { bv = v; bi = i; }
        }
    }
    shv[part][lane] = bv; shi[part][lane] = bi;
    __syncthreads();
    if (part == 0 && col < n1) {
        for (int q = 1; q < 4; ++q) {
            const float ov = shv[q][lane]; const int oi = shi[q][lane];
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        pval[(size_t)blockIdx.y * Kc + col] = bv;
        parg[(size_t)blockIdx.y * Kc + col] = bi;
    }
}

// single block: mutual check, thresholds, ordered emission in original indices
__global__ __launch_bounds__(1024) void lg_emit_kernel(
    const float* __restrict__ best0, const int* __restrict__ arg0, const int* __restrict__ arg1, const int* __restrict__ ind, float filter_thr, float min_conf, int32_t* __restrict__ ij_out,
    float* __restrict__ score_out, int32_t* __restrict__ info_out, LGCtrl* __restrict__ ctrl, int Kc,
    long out_stride, int* __restrict__ range_sticky) {
    __shared__ int wsum[16];
    const int pair = blockIdx.x;
    ctrl += pair; best0 += (size_t)pair * Kc; arg0 += (size_t)pair * Kc;
    arg1 += (size_t)pair * Kc; ind += (size_t)pair * 2 * Kc;
    ij_out += (size_t)pair * out_stride * 2; score_out += (size_t)pair * out_stride; info_out += pair * 4;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (Kc + 1023) / 1024;               // <= 8 (max_kpts <= 8192)
    // r05: (a one-workgroup chain of memory latencies, 19 us at the end of every forward) the row results are loaded before
    // the row counts are known, and the per-thread lists are unrolled into registers (rolled, they lived in scratch memory)
    int a0[8]; float b0[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
        if (q < per) { const int i = min(t * per + q, Kc - 1); a0[q] = arg0[i]; b0[q] = best0[i]; }
    const int n0 = ctrl->stop == 2 ? 0 : ctrl->n[0];
    const int n1 = ctrl->n[1];
    int keep[8], jj[8], cnt = 0; float sc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        keep[q] = 0; jj[q] = 0; sc[q] = 0.0f;
        if (q < per) {
            const int i = t * per + q;
            int k = 0;
            if (i < n0) {
                const int j = a0[q];
                const float s = expf(b0[q]);
                // a row whose scores are all NaN (non-finite input) has no arg-max: no match, no lookup
                const bool valid = (unsigned)j < (unsigned)n1;
                k = valid && (arg1[j] == i) && (s > filter_thr) && (s > min_conf);
                jj[q] = valid ? j : 0; sc[q] = s;
            }
            keep[q] = k; cnt += k;
        }
    }
    int incl = cnt;
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < 16; ++w) { if (w < wave) base += wsum[w]; total += wsum[w]; }
    int pos = base + incl - cnt;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int i = t * per + q;
        if (q < per && i < n0 && keep[q]) {
            ij_out[2 * pos] = ind[i];
            ij_out[2 * pos + 1] = ind[Kc + jj[q]];
            score_out[pos] = sc[q];
            ++pos;
        }
    }
    if (t == 0) {
        // a value left the fp16 range of the split-precision path while this pair was processed: its
        // matches are not fp32-grade - the count says so (-1) wherever the result travels, and the
        // instance remembers it (sslam_lightglue_range_overflow)
        if (ctrl->range_overflow) { total = -1; *range_sticky = 1; }
        ctrl->n_matches = total;
        info_out[0] = total;
        info_out[1] = ctrl->stop_layer + 1;     // upstream "stop" = i + 1
        info_out[2] = ctrl->n[0];
        info_out[3] = ctrl->n[1];
    }
}


// ======================================================================== //
//  Split-precision path (gemm_f16x3.hpp): activations that feed a contraction
//  live in HBM as fp16 (hi, lo) plane pairs written by their producer.
// ======================================================================== //
struct SplitOut { _Float16* hi; _Float16* lo; };

// weight matrix W[N][K] (row-major fp32) -> split planes in k-panel layout [K/64][N][64]
__global__ void lg_split_weight_kernel(const float* __restrict__ src, _Float16* __restrict__ hi,
                                       _Float16* __restrict__ lo, int N, int K, int* __restrict__ range_flag) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * K) return;
    const int n = (int)(i / K), k = (int)(i % K);
    const size_t o = panel_index(n, k, N);
    split_f32(src[i], hi[o], lo[o], range_flag);
}

// token states x[2][Kc][256] -> split planes in k-panel layout over the 2*Kc rows
// only_moved: refresh after pruning - an image whose rows did not move (n == n_prev) still has
// the planes its producer's epilogue wrote
__global__ __launch_bounds__(256) void lg_split_rows_kernel(const float* __restrict__ src, _Float16* __restrict__ hi,
                                                            _Float16* __restrict__ lo, int ld, int Kc, int NI, int NIc,
                                                            const LGCtrl* __restrict__ ctrl, int only_moved) {
    // grid (SPLIT_BLOCKS_PER_IMAGE, NI): a block owns a slice of ONE image and leaves at once when that
    // image has nothing to refresh (the common case after a pruning step that removed nothing: a
    // per-element grid of 32 k blocks spent 20 us just being dispatched)
    const int img = blockIdx.y;
    const LGCtrl& pc = ctrl_of(ctrl, img);
    const int n = pc.n[img & 1];
    if (pc.stop || (only_moved && n == pc.n_prev[img & 1])) return;
    const int cpr = ld / 8;                                // 8-column units per row (ld % 8 == 0)
    const size_t total = (size_t)n * cpr;
    const float* s = src + (size_t)img * Kc * ld;
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(u / cpr), col = (int)(u % cpr) * 8;
        const float4 a = *reinterpret_cast<const float4*>(s + (size_t)row * ld + col);
        const float4 b = *reinterpret_cast<const float4*>(s + (size_t)row * ld + col + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        half8 hh, ll;
#pragma unroll
        for (int e = 0; e < 8; ++e) { _Float16 x, y; split_f32(v[e], x, y, range_flag_of(ctrl, img)); hh[e] = x; ll[e] = y; }
        const size_t o = panel_index(img * Kc + row, col, NIc * Kc);
        *reinterpret_cast<half8*>(hi + o) = hh;
        *reinterpret_cast<half8*>(lo + o) = ll;
    }
}
constexpr int SPLIT_BLOCKS_PER_IMAGE = 64;

enum { EPH_QKV = 0, EPH_CROSS = 1, EPH_SPLIT = 2, EPH_F32 = 3, EPH_RESID = 4 };

struct LinearArgsH {
    SplitPtr A0, A1; int lda; int K0; int K;
    SplitPtr W; const float* bias; int N;
    float* out; int ldo;               // fp32 destination (F32 / RESID)
    SplitOut outs;                     // split destination (SPLIT / RESID), ld = ldo
    SplitOut q, k, vt;                 // QKV / CROSS destinations
    float q_scale, k_scale;
    const float* enc_cos; const float* enc_sin;
    const LGCtrl* ctrl; int Kc; int NIc;   // NIc: image capacity of the instance (plane rows = NIc * Kc)
};

// LDS ring depth of the split-precision GEMM.  Depth beyond 2 bought nothing measurable (the k-loop is
// bound by the per-CU load rate, not by tiles in flight), while a small footprint lets blocks of
// concurrently running kernels (other pairs / the extractor on other streams) share a CU.
constexpr int RING_MAX = 2;
template <int BM, int BN>
constexpr int ring_depth() {
    constexpr int stage_bytes = sslam::ring_stage_halves<BM, BN>() * 2;
    constexpr int d = (160 * 1024) / stage_bytes;
    return d > RING_MAX ? RING_MAX : (d < 2 ? 2 : d);
}

// erf for the GELU of the split-precision path: Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 absolute
// (+ fp32 rounding), branch-free: one v_rcp, one v_exp, five fma.  The libm erff the exact-fp32 path
// keeps is two divergent polynomial branches (~27 exec-mask switches per 8 elements in the ISA) and
// made this kernel VALU-bound; GELU(y) = y/2 (1 + erf(y / sqrt 2)) carries the absolute error times
// |y| / 2, i.e. below the fp32 rounding of the O(1) values it is added to downstream.
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
    return copysignf(fmaf(-p * t, e, 1.0f), x);
}

// ---- shared epilogue of the split-precision linears (ring kernel: 4 consumer waves as 2 x 2; big-tile
// kernel: 8 waves as WM x WN).  `lds` is the (now free) operand ring, NT the threads taking part.
template <int BM, int BN, int TM, int TN, int WN, int NT, int EPI>
__device__ __forceinline__ void linear_h_epilogue(const LinearArgsH& p, const RowDom& rd, int col0, size_t ibase,
                                                  const f32x16 (&c1)[TM][TN], const f32x16 (&c2)[TM][TN],
                                                  _Float16* lds) {
    // ---- epilogue.  The accumulator tile has its columns on lanes and its rows in registers, so a
    // direct store would be 2-byte (split planes) or 4-byte pieces per lane and is store-issue bound
    // (measured 5-13 us of a 18-24 us kernel).  Stage the fp32 tile in LDS (the ring is free now),
    // then every thread owns 8 consecutive columns of one row (or 8 consecutive rows of one V
    // column) and writes 16-byte pieces.
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // fp32 row stride of the staged tile: BN + 4 keeps the float4 row reads aligned; a V column tile
    // (read by columns, 8 tokens of one column per unit) uses BN + 1 so the 16 token runs of a
    // column fall on different banks
    constexpr int VCOL_FIRST = (EPI == EPH_QKV) ? 2 * D : D;
    const bool v_tile = (EPI == EPH_QKV || EPI == EPH_CROSS) && col0 >= VCOL_FIRST;          // wholly V columns
    const bool has_v = (EPI == EPH_QKV || EPI == EPH_CROSS) && col0 + BN > VCOL_FIRST;     // (a 192-wide tile can straddle k | v)
    const int ELD = v_tile ? BN + 1 : BN + 4;
    float* epi = reinterpret_cast<float*>(lds);
    // Row pass: thread t owns unit u = t + it NT = (row u / CH, 8-column chunk u % CH), ITERS units in all.
    // What a unit needs from HBM (the rotary table of its row, the residual it adds to) is asked for
    // HERE, before the tile is staged through LDS: inside the unit loop each load would be an exposed
    // ~1.5 us round trip per iteration (QKV: 83 us per 8-pair launch against 48 for the main loop alone).
    constexpr int CH = BN / 8;                         // 8-column chunks per row
    static_assert((BM * CH) % NT == 0, "units divide over the threads");
    constexpr int ITERS = BM * CH / NT;
    const int grow0 = (int)ibase + rd.row0;            // global plane row of tile row 0
    float4 pre_a[(EPI == EPH_QKV || EPI == EPH_RESID) ? ITERS : 1], pre_b[(EPI == EPH_QKV || EPI == EPH_RESID) ? ITERS : 1];
#ifndef LG_EPI_PREFETCH
#define LG_EPI_PREFETCH 1      // A/B switch (scripts/ab_lib.sh): 0 = load inside the unit loop
#endif
    auto pre_load = [&](int it) {
        const int u = t + it * NT, rl = u / CH, cl = (u % CH) * 8;
        if constexpr (EPI == EPH_QKV) {
            const int d = (col0 + cl) & 63;
            pre_a[it] = *reinterpret_cast<const float4*>(p.enc_cos + (size_t)(grow0 + rl) * ENC + (d >> 1));
            pre_b[it] = *reinterpret_cast<const float4*>(p.enc_sin + (size_t)(grow0 + rl) * ENC + (d >> 1));
        } else if constexpr (EPI == EPH_RESID) {
            const size_t o = (size_t)(grow0 + rl) * p.ldo + col0 + cl;
            pre_a[it] = *reinterpret_cast<const float4*>(p.out + o);
            pre_b[it] = *reinterpret_cast<const float4*>(p.out + o + 4);
        }
    };
    if constexpr (LG_EPI_PREFETCH && (EPI == EPH_QKV || EPI == EPH_RESID)) {
        if (!v_tile) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) pre_load(it);
        }
    }
    __syncthreads();                                   // consumers are done with the last k-tile
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cl = wn * 32 * TN + j * 32 + (lane & 31);
            const float bv = p.bias[col0 + cl];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wm * 32 * TM + i * 32 + acc_row(r, lane);
                epi[rl * ELD + cl] = (c1[i][j][r] + c2[i][j][r] * SPLIT_INV) + bv;
            }
        }
    __syncthreads();

#ifndef LG_EPI_FAST_SPLIT
#define LG_EPI_FAST_SPLIT 1    // A/B switch (scripts/ab_lib.sh): 0 = the branchy scalar split_f32 per value
#endif
    float amax = 0.0f;
    auto split8 = [&](const float (&v)[8], uint4& hi, uint4& lo) {
#if LG_EPI_FAST_SPLIT
        sslam::split8_fast(v, hi, lo, amax);
#else
        half8 hh, ll;
#pragma unroll
        for (int e = 0; e < 8; ++e) { _Float16 a, b2; split_f32(v[e], a, b2, range_flag_of(p.ctrl, rd.img)); hh[e] = a; ll[e] = b2; }
        hi = *reinterpret_cast<uint4*>(&hh);
        lo = *reinterpret_cast<uint4*>(&ll);
#endif
    };
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        if (v_tile) break;
        const int u = t + it * NT;
        const int rl = u / CH, cl = (u % CH) * 8;
        const int row = rd.row0 + rl, col = col0 + cl;
        if (row >= rd.n) continue;
        if constexpr (!LG_EPI_PREFETCH && (EPI == EPH_QKV || EPI == EPH_RESID)) pre_load(it);
        float v[8];
        {
            const float4 a = *reinterpret_cast<const float4*>(&epi[rl * ELD + cl]);
            const float4 b4 = *reinterpret_cast<const float4*>(&epi[rl * ELD + cl + 4]);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b4.x; v[5] = b4.y; v[6] = b4.z; v[7] = b4.w;
        }
        if constexpr (EPI == EPH_QKV || EPI == EPH_CROSS) {
            const int s = col >> 8, hd = (col >> 6) & 3, d = col & 63;
            const bool is_v = (EPI == EPH_QKV) ? (s == 2) : (s == 1);
            if (is_v) continue;                        // V goes out transposed below
            const bool is_k = (EPI == EPH_QKV) && s == 1;
            if constexpr (EPI == EPH_QKV) {
                // rotary: out[2i] = x[2i] cos_i - x[2i+1] sin_i ; out[2i+1] = x[2i+1] cos_i + x[2i] sin_i
                const float4 c4 = pre_a[it], s4 = pre_b[it];
                const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x0 = v[2 * e], x1 = v[2 * e + 1];
                    v[2 * e] = x0 * cc[e] - x1 * ss[e];
                    v[2 * e + 1] = x1 * cc[e] + x0 * ss[e];
                }
            }
            const float scale = is_k ? p.k_scale : p.q_scale;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= scale;
            uint4 hi, lo;
            split8(v, hi, lo);
            const SplitOut dst = is_k ? p.k : p.q;
            const size_t o = (((size_t)rd.img * NH + hd) * p.Kc + row) * DH + d;
            *reinterpret_cast<uint4*>(dst.hi + o) = hi;
            *reinterpret_cast<uint4*>(dst.lo + o) = lo;
        } else {
            const size_t o = (size_t)(grow0 + rl) * p.ldo + col;
            if constexpr (EPI == EPH_RESID) {
                const float4 a = pre_a[it], b4 = pre_b[it];
                v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b4.x; v[5] += b4.y; v[6] += b4.z; v[7] += b4.w;
            }
            if constexpr (EPI == EPH_F32 || EPI == EPH_RESID) {
                *reinterpret_cast<float4*>(p.out + o) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(p.out + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
            if constexpr (EPI == EPH_SPLIT || EPI == EPH_RESID) {
                uint4 hi, lo;
                split8(v, hi, lo);
                const size_t po = panel_index(grow0 + rl, col, p.NIc * p.Kc);
                *reinterpret_cast<uint4*>(p.outs.hi + po) = hi;
                *reinterpret_cast<uint4*>(p.outs.lo + po) = lo;
            }
        }
    }
    if constexpr (EPI == EPH_QKV || EPI == EPH_CROSS) {
        // V columns, transposed and key-tile-major: vt[img][head][token / 64][d][pos(token % 64)].
        // Inside every group of 16 keys the two middle quads are swapped (pos = key with bits 2 and 3
        // exchanged): the P.V MFMA takes its k-slots in the order of the accumulator rows of S^T -
        // keys {4h..4h+3, 8+4h..8+4h+3} of the group for lane half h - and in this order those eight
        // keys are ONE 16-byte run, i.e. one ds_read_b128 per operand instead of two ds_read_b64 plus
        // register shuffling.  A unit = one column x the 8 tokens {b..b+3, b+8..b+11} of such a run.
        // Unit order: the 16 runs of a column on consecutive lanes - 8 runs are one whole 128-byte line
        // of V^T (64 key positions of one d), and the next column's lines follow contiguously, so a
        // wave stores two 512-byte segments per plane instead of 64 separate 16-byte pieces.
        constexpr int RG = BM / 8;
        if (!has_v) { sslam::split_range_check(amax, range_flag_of(p.ctrl, rd.img)); return; }
        for (int u = t; u < BN * RG; u += NT) {
            const int rg = u % RG, cl = u / RG;
            const int col = col0 + cl;
            if (col < VCOL_FIRST) continue;
            const int hd = (col >> 6) & 3, d = col & 63;
            const int rl0 = (rg >> 1) * 16 + (rg & 1) * 4;     // first token of the run inside the tile
            const int row = rd.row0 + rl0;
            float v[8];
            // keys >= n of the last 64-key tile are written as exact zeros: attention gives them
            // P = 0, and 0 x (stale, possibly non-finite) would not be 0
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int dr = (e & 3) + 8 * (e >> 2);         // token offset of run element e
                v[e] = row + dr < rd.n ? epi[(rl0 + dr) * ELD + cl] : 0.0f;
            }
            uint4 hi, lo;
            split8(v, hi, lo);
            const int pos = ((row & 63) & ~15) + (rg & 1) * 8; // run position inside the 64-key tile
            const size_t o = ((((size_t)rd.img * NH + hd) * (p.Kc / AK) + (row >> 6)) * DH + d) * AK + pos;
            *reinterpret_cast<uint4*>(p.vt.hi + o) = hi;
            *reinterpret_cast<uint4*>(p.vt.lo + o) = lo;
        }
    }
    sslam::split_range_check(amax, range_flag_of(p.ctrl, rd.img));
}

template <int BM, int BN, int TM, int TN, int EPI>
__global__ __launch_bounds__(512) void lg_linear_h_kernel(LinearArgsH p) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lg_ring[];
    int rb, cb;
    xcd_tile(gridDim.x, rb, cb);
    const int nb = (p.Kc + BM - 1) / BM;
    RowDom rd;
    rd.img = rb / nb; rd.row0 = (rb % nb) * BM; rd.n = n_of(p.ctrl, rd.img);
    if (ctrl_of(p.ctrl, rd.img).stop) return;
    if (rd.row0 >= rd.n) return;
    const int col0 = cb * BN;
    const size_t ibase = (size_t)rd.img * p.Kc;
    GemmAH ga{{p.A0.hi, p.A0.lo}, {p.A1.hi ? p.A1.hi : p.A0.hi, p.A1.lo ? p.A1.lo : p.A0.lo}, p.lda, p.K0};
    f32x16 c1[TM][TN], c2[TM][TN];
    // plane rows are global token rows (image-major, NIc*Kc per panel); clamp inside this image
    gemm_mainloop_ring<BM, BN, TM, TN, ring_depth<BM, BN>()>(ga, p.W, p.NIc * p.Kc, p.K, (int)ibase + rd.row0,
                                                             (int)ibase + p.Kc, col0, p.N, lg_ring, c1, c2);
    if (threadIdx.x >= 256) return;                    // producer waves (4-7) are done
#if defined(SSLAM_DBG_NOEPI)
    if (c1[0][0][0] != 123456.0f) return;
#endif
    linear_h_epilogue<BM, BN, TM, TN, 2, 256, EPI>(p, rd, col0, ibase, c1, c2, lg_ring);
}

// Big-tile form for batched token sets (gemm_f16x3_big.hpp): 128 x 128 tile, 4 waves (2 x 2), 2-stage ring,
// TWO workgroups per CU (one's epilogue runs under the other's main loop).  Used by the two projections of a
// block (QKV, shared-qk cross); the FFN is one kernel of its own (ffn_fused.hpp).
template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM * WN * 64, 2) void lg_linear_big_kernel(LinearArgsH p) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lg_ring[];
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    int rb, cb;
    xcd_tile(gridDim.x, rb, cb);
    const int nb = (p.Kc + BM - 1) / BM;
    RowDom rd;
    rd.img = rb / nb; rd.row0 = (rb % nb) * BM; rd.n = n_of(p.ctrl, rd.img);
    if (ctrl_of(p.ctrl, rd.img).stop) return;
    if (rd.row0 >= rd.n) return;
    const int col0 = cb * BN;
    const size_t ibase = (size_t)rd.img * p.Kc;
    GemmAH ga{{p.A0.hi, p.A0.lo}, {p.A1.hi ? p.A1.hi : p.A0.hi, p.A1.lo ? p.A1.lo : p.A0.lo}, p.lda, p.K0};
    f32x16 c1[TM][TN], c2[TM][TN];
    constexpr int NW = WM * WN;
    sslam::gemm_mainloop_big<BM, BN, WM, WN>(ga, p.W, p.NIc * p.Kc, p.K, (int)ibase + rd.row0, (int)ibase + p.Kc, col0,
                                             p.N, lg_ring, c1, c2);
#if defined(SSLAM_DBG_NOEPI)
    if (c1[0][0][0] != 123456.0f) return;
#endif
    linear_h_epilogue<BM, BN, TM, TN, WN, NW * 64, EPI>(p, rd, col0, ibase, c1, c2, lg_ring);
}

// ---- r04: the input projection, the final projection and the similarity GEMM on the split pipe (precision 1).  On the fp32 matrix
// instruction the similarity ran at 63 % of ITS peak (17.2 GFLOP per 8 pairs in 175 us = 98 TFLOP/s of 157) and the two projections
// took 45 us each, each followed by a pass that split its output for the next consumer (lg_split_rows 13 us, a split of md 14 us).
// Here: three f16 MFMAs per product into two fp32 accumulators as everywhere else (gemm_f16x3.hpp), on the big-tile main loop
// (gemm_f16x3_big.hpp: LDS-DMA ring, two workgroups per CU - a register-staged loop spends one global -> LDS round trip per 32-deep
// k-tile, and at K = 128 / 256 that round trip is the kernel: 75 / 32 us against 66 / 26), and the projections' epilogues write
// the fp32 result AND the planes the next consumer reads.  Every plane in k-panel layout (A: token-state / descriptor planes over
// NIc Kc rows; W: the weight's own panel block, chosen by the pair's stop layer, or - similarity - the other image's rows of the
// same planes).  Bit-identical to the register-staged form; match indices unchanged against the fp32 kernels on every parity test.
struct ProjBigArgs {
    const _Float16 *a_hi, *a_lo; int a_rows; int K;
    const _Float16 *w_hi, *w_lo; long w_layer_stride;
    const float* bias; long b_layer_stride; int by_stop_layer; int ignore_stop;
    float out_scale; float* out;
    _Float16 *o_hi, *o_lo; int o_rows;                          // output planes: k-panel layout over o_rows rows
    const LGCtrl* ctrl; int Kc;
};
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, 2) void lg_proj_big_kernel(ProjBigArgs p) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lg_ring[];
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    const int img = blockIdx.z;
    const LGCtrl& pc = ctrl_of(p.ctrl, img);
    if (pc.stop == 2 || (pc.stop && !p.ignore_stop)) return;
    const int n = pc.n[img & 1], row0 = blockIdx.y * BM, col0 = blockIdx.x * BN;
    if (row0 >= n) return;
    const size_t lw = p.by_stop_layer ? (size_t)pc.stop_layer * p.w_layer_stride : 0;
    const float* bias = p.bias + (p.by_stop_layer ? (size_t)pc.stop_layer * p.b_layer_stride : 0);
    const int g0 = img * p.Kc;
    const sslam::SplitPtr a{p.a_hi, p.a_lo};
    GemmAH ga{a, a, 0, p.K};
    f32x16 c1[TM][TN], c2[TM][TN];
    sslam::gemm_mainloop_big<BM, BN, WM, WN>(ga, sslam::SplitPtr{p.w_hi + lw, p.w_lo + lw}, p.a_rows, p.K, g0 + row0, g0 + p.Kc, col0, D,
                                             lg_ring, c1, c2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave / WN, wn = wave % WN;
    float amax = 0.0f;
    const bool odd = lane & 1;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = col0 + wn * 32 * TN + j * 32 + (lane & 31), cpair = col & ~1;
            const float b = bias[col];
            // (neighbouring lanes trade every other value: the even lane finishes columns (c, c + 1) of row r, the odd lane of row r + 1)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float va = (c1[i][j][r] + c2[i][j][r] * sslam::SPLIT_INV + b) * p.out_scale;
                const float vb = (c1[i][j][r + 1] + c2[i][j][r + 1] * sslam::SPLIT_INV + b) * p.out_scale;
                const float got = __shfl_xor(odd ? va : vb, 1);
                const int row = row0 + wm * 32 * TM + i * 32 + acc_row(odd ? r + 1 : r, lane);
                const bool live = row < n;                     // (rows past the count: kept out of the range check)
                const float x0 = live ? (odd ? got : va) : 0.0f, x1 = live ? (odd ? vb : got) : 0.0f;
                unsigned h2, l2;
                sslam::split2_fast(x0, x1, h2, l2, amax);
                if (live) {
                    const size_t o = panel_index(g0 + row, cpair, p.o_rows);
                    *reinterpret_cast<float2*>(p.out + (size_t)(g0 + row) * D + cpair) = make_float2(x0, x1);
                    *reinterpret_cast<unsigned*>(p.o_hi + o) = h2;
                    *reinterpret_cast<unsigned*>(p.o_lo + o) = l2;
                }
            }
        }
    sslam::split_range_check(amax, range_flag_of(p.ctrl, img));
}

struct SimBigArgs { const _Float16* md_hi; const _Float16* md_lo; float* sim; int Kc; int rows_total; const LGCtrl* ctrl; };
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, 2) void lg_sim_big_kernel(SimBigArgs p) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lg_ring[];
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    int pair = blockIdx.z, bx = blockIdx.x, by = blockIdx.y;         // XCD-aware order: see lg_sim_kernel
    if ((gridDim.z & 7) == 0) {
        const unsigned T = gridDim.x * gridDim.y;
        const unsigned L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned grp = L / (8 * T), rem = L % (8 * T);
        pair = (int)(grp * 8 + (rem & 7));
        const unsigned tix = rem >> 3;
        bx = (int)(tix % gridDim.x); by = (int)(tix / gridDim.x);
    }
    const LGCtrl& pc = p.ctrl[pair];
    if (pc.stop == 2) return;
    const int n0 = pc.n[0], n1 = pc.n[1];
    const int row0 = by * BM, col0 = bx * BN;
    if (row0 >= n0 || col0 >= n1) return;
    const int g0 = 2 * pair * p.Kc;                                  // plane row of image 0's token 0; image 1 follows
    float* simp = p.sim + (size_t)pair * p.Kc * p.Kc;
    const sslam::SplitPtr a{p.md_hi, p.md_lo};
    GemmAH ga{a, a, 0, D};
    // W = image 1's rows of the same planes: the pointer moved by its first row inside every panel, `col_cap` = the planes' row count
    const size_t w0 = (size_t)(g0 + p.Kc) * sslam::PANEL_K;
    f32x16 c1[TM][TN], c2[TM][TN];
    sslam::gemm_mainloop_big<BM, BN, WM, WN>(ga, sslam::SplitPtr{p.md_hi + w0, p.md_lo + w0}, p.rows_total, D, g0 + row0, g0 + p.Kc, col0,
                                             p.rows_total, lg_ring, c1, c2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave / WN, wn = wave % WN;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = col0 + wn * 32 * TN + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wm * 32 * TM + i * 32 + acc_row(r, lane);
                if (row < n0 && col < n1) simp[(size_t)row * p.Kc + col] = c1[i][j][r] + c2[i][j][r] * sslam::SPLIT_INV;
            }
        }
}

// ---- the whole FFN of a block as one kernel (ffn_fused.hpp): batched token sets.  One workgroup = 32 TT tokens:
// 64 when the token set gives every CU a tile, 32 below that (one pair: 64 -> 128 workgroups of half the work).
struct FfnKArgs { sslam::FfnFusedArgs f; const LGCtrl* ctrl; int Kc; };

template <int TT, bool FOLD = false>
__global__ __launch_bounds__(512, 2) void lg_ffn_fused_kernel(FfnKArgs p) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lg_ring[];
    constexpr int TOK = 32 * TT;
    const int nb = p.Kc / TOK;                                 // Kc is a multiple of 128
    const int img = blockIdx.x / nb, row0 = (blockIdx.x % nb) * TOK;
    const LGCtrl& pc = ctrl_of(p.ctrl, img);
    if (pc.stop) return;
    const int n = pc.n[img & 1];
    if (row0 >= n) return;
    const int ibase = img * p.Kc;
    sslam::FfnFusedArgs f = p.f;
    f.unconf = &const_cast<LGCtrl*>(p.ctrl)[img >> 1].unconf;          // (used with the token heads only)
    if constexpr (FOLD) { f.part_base = (long)img * NH * p.Kc + row0; f.part_kc = p.Kc; }
    sslam::ffn_fused_tile<TT, FOLD>(f, ibase + row0, ibase + p.Kc, min(TOK, n - row0), range_flag_of(p.ctrl, img), lg_ring);
}


// W1 [512][512] / W2 [256][512] (row-major fp32) -> split planes in the fragment order the fused FFN streams
__global__ void lg_pack_ffn_kernel(const float* __restrict__ w1, const float* __restrict__ w2, _Float16* __restrict__ w1f,
                                   _Float16* __restrict__ w2f, int* __restrict__ range_flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 512 * 512) {
        _Float16 a, b;
        split_f32(w1[i], a, b, range_flag);
        w1f[sslam::ffn_w1_frag_index(0, i >> 9, i & 511)] = a;
        w1f[sslam::ffn_w1_frag_index(1, i >> 9, i & 511)] = b;
    }
    if (i < 256 * 512) {
        _Float16 a, b;
        split_f32(w2[i], a, b, range_flag);
        w2f[sslam::ffn_w2_frag_index(0, i >> 9, i & 511)] = a;
        w2f[sslam::ffn_w2_frag_index(1, i >> 9, i & 511)] = b;
    }
}

// LayerNorm(512) + GELU: fp32 hidden in, split planes out (one wave / row)
__global__ __launch_bounds__(256) void lg_ln_gelu_h_kernel(const float* __restrict__ hid, SplitOut outs,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const LGCtrl* __restrict__ ctrl, int Kc, int NI, int NIc) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int img = gw / Kc, row = gw % Kc;
    if (img >= NI || ctrl_of(ctrl, img).stop || row >= n_of(ctrl, img)) return;
    const size_t base = ((size_t)img * Kc + row) * 512;
    const float4 a = *reinterpret_cast<const float4*>(hid + base + lane * 4);
    const float4 b = *reinterpret_cast<const float4*>(hid + base + 256 + lane * 4);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + lane * 4), gb = *reinterpret_cast<const float4*>(gamma + 256 + lane * 4);
    const float4 ba = *reinterpret_cast<const float4*>(beta + lane * 4), bb = *reinterpret_cast<const float4*>(beta + 256 + lane * 4);
    float s = (a.x + a.y) + (a.z + a.w) + (b.x + b.y) + (b.z + b.w);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / 512.0f;
    float v[8] = {a.x - mean, a.y - mean, a.z - mean, a.w - mean, b.x - mean, b.y - mean, b.z - mean, b.w - mean};
    const float gm[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
    const float bt[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) q += v[i] * v[i];
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / 512.0f + 1e-5f);
    half4 h0, l0, h1, l1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float y = v[i] * rstd * gm[i] + bt[i];
        const float g = 0.5f * y * (1.0f + erf_as(y * 0.70710678118654752440f));
        _Float16 hh, ll;
        split_f32(g, hh, ll, range_flag_of(ctrl, img));
        if (i < 4) { h0[i] = hh; l0[i] = ll; } else { h1[i - 4] = hh; l1[i - 4] = ll; }
    }
    const size_t p0 = panel_index(img * Kc + row, lane * 4, NIc * Kc), p1 = panel_index(img * Kc + row, 256 + lane * 4, NIc * Kc);
    *reinterpret_cast<half4*>(outs.hi + p0) = h0;
    *reinterpret_cast<half4*>(outs.lo + p0) = l0;
    *reinterpret_cast<half4*>(outs.hi + p1) = h1;
    *reinterpret_cast<half4*>(outs.lo + p1) = l1;
}

// ---- attention, split precision --------------------------------------------------------
// K / V^T tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no
// ds_write).  A wave-instruction writes 1 KiB = 8 rows of 128 B linearly, so the LDS image is
// un-padded; bank conflicts are avoided by XOR-swizzling the 16-byte chunk index with
// f(row) = (row >> 1) & 7, applied to the per-lane SOURCE address and to every read.
struct __attribute__((aligned(16))) AttnSmemH {
    _Float16 k_hi[2][AK * DH];
    _Float16 k_lo[2][AK * DH];
    _Float16 vt_hi[2][DH * AK];
    _Float16 vt_lo[2][DH * AK];
};

struct AttnArgsH {
    SplitPtr Q, K, VT;                                // Q,K [NI][4][Kc][64]; VT [NI][4][Kc/64][64][64]
    int cross;
    float* o_part; float* m_part; float* l_part;      // key-split partials (KS > 1)
    SplitOut msg;                                     // KS == 1: the normalised context goes straight to the split planes
    int KS; int Kc; int NIc; const LGCtrl* ctrl;
    int p_single;                                     // precision study only (sslam_lightglue_debug_split_form bit 0x04): P as ONE fp16 plane
};

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// ---- attention, split precision, software-pipelined ---------------------------------------
// Flash-style loop over 64-key tiles, re-timed so that
// one iteration (a 32-key sub-step j) holds three INDEPENDENT instruction streams the scheduler
// can interleave inside one basic block:
//     MFMA   O += V^T(j-1) P(j-1)          (P of the previous sub-step, 12 MFMA)
//     MFMA   S(j+1) = K(j+1) Q^T           (next sub-step's logits, 12 MFMA)
//     VALU   softmax + hi/lo split of S(j) (this sub-step)
// so a wave's matrix-core work runs under its own softmax arithmetic instead of after it.
// K is read one sub-step ahead and V^T one behind, hence separate double buffers with one barrier
// per 64-key tile: after the even sub-step of tile t, K(t+2) and V^T(t+1) are issued.
// P is carried scaled by 2^14 (exp2 argument bias; o and l share the factor, m does not): its fp16
// high plane then leaves the normal range only below 2^-28 of the row maximum, so the split needs
// no subnormal guard (MFMA flushes fp16 subnormal operands).
constexpr float P_BIAS = 14.0f;
// scheduling recipe for one sub-step's basic block: 24 x { 1 MFMA, 8 VALU } (5 / 0 VALU per MFMA and a
// fragment-read-ahead recipe measured within 1 %: profiles/r02_attention_experiments.md)
#define ATTN_INTERLEAVE()                                                     \
    _Pragma("unroll") for (int ig_ = 0; ig_ < 24; ++ig_) {                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    \
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);                    \
    }
typedef float float2v __attribute__((ext_vector_type(2)));


// Deferred rescale (flash-attention "lazy max"): the running maximum - and with it the 64
// accumulator registers of O - is only updated when some query of the wave sees a logit more than
// RESCALE_THR (log2 units) above its current reference; otherwise P is simply taken against the old
// reference (P <= 2^(P_BIAS + RESCALE_THR) = 2^15.5 < 65504, still an fp16 normal) and the O rescale
// (a quarter of the softmax arithmetic of a sub-step, on a kernel that is VALU-issue bound) is
// skipped.  After the first few key tiles almost every sub-step takes the cheap path.  The caller
// applies `alpha` to O when `rescale` is set, AFTER the P.V of the previous sub-step (those P were
// taken against the old reference) - both halves of the wave decide together (wave-uniform branch).
constexpr float RESCALE_THR = 1.5f;

template <bool MASK>
__device__ __forceinline__ void attn_softmax_step(const f32x16& s1, const f32x16& s2, int kbase, int nk, int lane,
                                                  float& m_run, float& l_run, half8 (&ph)[2], half8 (&pl)[2],
                                                  float& alpha, bool& rescale, bool qvalid, bool p_single) {
    float2v sv[8];
    const float2v inv2 = {SPLIT_INV, SPLIT_INV};
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float2v a = {s1[2 * r], s1[2 * r + 1]}, b = {s2[2 * r], s2[2 * r + 1]};
        sv[r] = __builtin_elementwise_fma(b, inv2, a);
    }
    if constexpr (MASK) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (kbase + acc_row(r, lane) >= nk) sv[r >> 1][r & 1] = -INFINITY;
    }
    float tmax = fmaxf(sv[0][0], sv[0][1]);
#pragma unroll
    for (int r = 1; r < 8; ++r) tmax = fmaxf(fmaxf(tmax, sv[r][0]), sv[r][1]);
    {   // the other 16 keys of this query live in lane ^ 32
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
        tmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    // (m_run = -inf on the first sub-step: tmax is finite there - every block's first sub-tile holds at
    // least one valid key - so the comparison is true and the reference is initialised)
    // only lanes that own a real query vote: the rows beyond n hold whatever an earlier, larger problem
    // left there, and the path taken (hence the last bits of the result) must not depend on it
    rescale = __any(qvalid && tmax > m_run + RESCALE_THR);
    if (rescale) {
        const float m_new = fmaxf(m_run, tmax);
        alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
    } else {
        alpha = 1.0f;
    }
    // P = 2^(s - m + P_BIAS).  (m - P_BIAS) is rounded ONCE: for |m| >= 2^23 its rounding error reaches
    // half a unit and larger and 2^(15.5 + error) leaves fp16 (inf in the high plane, then NaN).  Logits
    // that large only come from absurdly scaled inputs, but the answer must still be a softmax: the
    // bias is dropped there (P <= 2^1.5; the common factor cancels in o / l either way).
    const float mb = fabsf(m_run) < 4.0e6f ? m_run - P_BIAS : m_run;
    // The low plane of P is kept UNSCALED: pl = fp16(p - ph), one mixed-precision fma per element
    // (v_fma_mixlo/hi_f16) instead of convert-back, subtract, scale, convert.  An fp16 subnormal
    // operand is flushed by the MFMA, so the low part is lost where p - ph < 2^-14, i.e. for
    // p < 2^-3 = 2^-17 of the 2^14 reference - entries whose low part is below 2^-28 of the row's
    // largest weight, far under fp32 resolution of the sum.  (V^T.pl then accumulates at true scale
    // into the hi.hi accumulator; see `pv`.)
    float psum0 = 0.0f, psum1 = 0.0f;
    typedef unsigned uint4v __attribute__((ext_vector_type(4)));
    uint4v hu[2], lu[2];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float p0 = __builtin_amdgcn_exp2f(sv[r][0] - mb);
        const float p1 = __builtin_amdgcn_exp2f(sv[r][1] - mb);
        const float2v pv2 = {p0, p1};
        const sslam::half2v hv2 = __builtin_convertvector(pv2, sslam::half2v);
        const unsigned h2 = __builtin_bit_cast(unsigned, hv2);
        unsigned l2;
        if (p_single) {
            // precision study (never the product): P is its fp16 rounding alone and the row sum is taken over the ROUNDED
            // weights, so o / l stays an exact softmax of slightly perturbed logits
            psum0 += (float)hv2[0]; psum1 += (float)hv2[1];
            l2 = 0u;
        } else {
            psum0 += p0; psum1 += p1;
            // l2.lo = fp16(h2.lo * -1 + p0) ; l2.hi = fp16(h2.hi * -1 + p1)   (sources: fp16 half, f32, f32)
            asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l2) : "v"(h2), "v"(p0));
            asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l2) : "v"(h2), "v"(p1));
        }
        hu[r >> 2][r & 3] = h2;
        lu[r >> 2][r & 3] = l2;
    }
    ph[0] = __builtin_bit_cast(half8, hu[0]); ph[1] = __builtin_bit_cast(half8, hu[1]);
    pl[0] = __builtin_bit_cast(half8, lu[0]); pl[1] = __builtin_bit_cast(half8, lu[1]);
    l_run = l_run * alpha + (psum0 + psum1);
}

// (r02 measured this kernel with ablation builds - no MFMA / no softmax / no fragment reads / no tile DMA -, a
// pinned-softmax variant, a scaled low plane of P and an 8-wave ping-pong form: all recorded in
// profiles/r02_attention_experiments.md; the switches and the ping-pong kernel now live in scripts/ubench/.)
__global__ __launch_bounds__(256, 2) void lg_attention_p_kernel(AttnArgsH p) {
    __shared__ AttnSmemH sm;
    const int nqb = gridDim.x, nslab = gridDim.y * gridDim.z;
    int slab, qb;
    {
        const int b = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        if ((nslab & 7) == 0) { const int xcd = b & 7, idx = b >> 3; slab = xcd + 8 * (idx / nqb); qb = idx % nqb; }
        else { slab = blockIdx.z * gridDim.y + blockIdx.y; qb = blockIdx.x; }
    }
    const int z = slab / gridDim.y, ih = slab % gridDim.y;
    const int img = ih >> 2, head = ih & 3;
    if (ctrl_of(p.ctrl, img).stop) return;
    const int kimg = p.cross ? (img ^ 1) : img;
    const int nq = n_of(p.ctrl, img), nk = n_of(p.ctrl, kimg);
    const int q0 = qb * AQ;
    if (q0 >= nq) return;
    const int t = threadIdx.x, lane = t & 63, h = lane >> 5, lr = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int ntiles = (nk + AK - 1) / AK;
    const int t0 = (int)((long)z * ntiles / p.KS), t1 = (int)((long)(z + 1) * ntiles / p.KS);
    const bool p_single = p.p_single != 0;

    const size_t qoff = ((size_t)img * NH + head) * p.Kc * DH;
    const size_t koff = ((size_t)kimg * NH + head) * p.Kc * DH;

    const int qi = min(q0 + wave * 32 + lr, p.Kc - 1);
    const bool qvalid = q0 + wave * 32 + lr < nq;
    half8 qh[4], ql[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        qh[s] = *reinterpret_cast<const half8*>(p.Q.hi + qoff + (size_t)qi * DH + 16 * s + 8 * h);
        ql[s] = *reinterpret_cast<const half8*>(p.Q.lo + qoff + (size_t)qi * DH + 16 * s + 8 * h);
    }

    f32x16 o1a, o2a, o1b, o2b;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o1a[r] = 0.0f; o2a[r] = 0.0f; o1b[r] = 0.0f; o2b[r] = 0.0f; }
    float m_run = -INFINITY, l_run = 0.0f;

    // wave w owns plane w (K hi, K lo, V^T hi, V^T lo): 8 DMA instructions of 8 rows per tile
    const _Float16* gplane = (wave == 0 ? p.K.hi : wave == 1 ? p.K.lo : wave == 2 ? p.VT.hi : p.VT.lo) + koff;
    const bool is_v = wave >= 2;
    const int lrow = lane >> 3, lcp = lane & 7;
    auto issue_tile = [&](int tile, int buf) {
        _Float16* dst = wave == 0 ? sm.k_hi[buf] : wave == 1 ? sm.k_lo[buf] : wave == 2 ? sm.vt_hi[buf] : sm.vt_lo[buf];
#pragma unroll
        for (int rg = 0; rg < 8; ++rg) {
            const int row = rg * 8 + lrow;
            const int c = lcp ^ ((row >> 1) & 7);
            const _Float16* src = is_v ? gplane + ((size_t)tile * DH + row) * AK + c * 8
                                       : gplane + (size_t)min(tile * AK + row, p.Kc - 1) * DH + c * 8;
            glds16(src, dst + rg * 8 * DH);
        }
    };

    int koffs[2][4], voffs[2][2][2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
        const int krow = sub * 32 + lr, kswz = (krow >> 1) & 7;
#pragma unroll
        for (int s = 0; s < 4; ++s) koffs[sub][s] = krow * DH + (((2 * s + h) ^ kswz) * 8);
#pragma unroll
        for (int s2i = 0; s2i < 2; ++s2i)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                // V^T keys are stored with the middle quads of every 16 swapped (linear epilogue): the
                // eight k-slots of lane half h are the 16-byte chunk 4 sub + 2 s2i + h of row d
                const int d = db * 32 + lr, vswz = (d >> 1) & 7, c0 = 4 * sub + 2 * s2i + h;
                voffs[sub][s2i][db] = d * AK + ((c0 ^ vswz) * 8);
            }
    }

    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto qk = [&](int buf, int sub, f32x16& s1, f32x16& s2) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const half8 kh = *reinterpret_cast<const half8*>(&sm.k_hi[buf][koffs[sub][s]]);
            const half8 kl = *reinterpret_cast<const half8*>(&sm.k_lo[buf][koffs[sub][s]]);
            s1 = mfma16(kh, qh[s], s == 0 ? zero16 : s1);
            s2 = mfma16(kh, ql[s], s == 0 ? zero16 : s2);
            s2 = mfma16(kl, qh[s], s2);
        }
    };
    auto pv = [&](int buf, int sub, const half8 (&ph)[2], const half8 (&pl)[2]) {
#pragma unroll
        for (int s2i = 0; s2i < 2; ++s2i) {
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int vo = voffs[sub][s2i][db];
                const half8 vh = *reinterpret_cast<const half8*>(&sm.vt_hi[buf][vo]);
                const half8 vl = *reinterpret_cast<const half8*>(&sm.vt_lo[buf][vo]);
                if (db == 0) {          // pl at true scale: with the hi.hi products; V^T lo is the scaled plane
                    o1a = mfma16(vh, ph[s2i], o1a);
                    o1a = mfma16(vh, pl[s2i], o1a);
                    o2a = mfma16(vl, ph[s2i], o2a);
                } else {
                    o1b = mfma16(vh, ph[s2i], o1b);
                    o1b = mfma16(vh, pl[s2i], o1b);
                    o2b = mfma16(vl, ph[s2i], o2b);
                }
            }
        }
    };

    if (t0 < t1) {
        issue_tile(t0, 0);                       // K(t0) and V^T(t0)
        __syncthreads();
        if (!is_v && t0 + 1 < t1) issue_tile(t0 + 1, 1);          // K(t0+1); V^T(t0+1) follows the first even step
        f32x16 s1, s2, n1, n2;                   // logits: current / next sub-step (ping-pong)
        half8 ph[2], pl[2], nh[2], nl[2];        // P: previous / current sub-step (ping-pong)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) { ph[i][e] = (_Float16)0.0f; pl[i][e] = (_Float16)0.0f; }
        qk(0, 0, s1, s2);
        // One tile = two branch-free sub-steps.  First tile: the "previous" P is zero and is
        // multiplied into the (finite) landed V^T buffer; last tile: the look-ahead logits are
        // computed on the current K buffer and dropped.
        auto tile_body = [&](auto mask_c, int tile) {
            constexpr bool MASK = decltype(mask_c)::value;
            const int b = (tile - t0) & 1;
            const bool last = tile + 1 == t1;
            float alpha; bool rescale;
            // even: softmax(S(tile,0)) | S(tile,1) = K.Q^T | O += V^T(tile-1,1) P
            pv(tile > t0 ? b ^ 1 : b, 1, ph, pl);
            qk(b, 1, n1, n2);
            attn_softmax_step<MASK>(s1, s2, tile * AK, nk, lane, m_run, l_run, nh, nl, alpha, rescale, qvalid, p_single);
            ATTN_INTERLEAVE();
            if (rescale) { o1a *= alpha; o2a *= alpha; o1b *= alpha; o2b *= alpha; }
            __syncthreads();     // K(tile+1), V^T(tile) landed; K(tile) and V^T(tile-1) are free
            if (is_v) { if (tile + 1 < t1) issue_tile(tile + 1, b ^ 1); }
            else      { if (tile + 2 < t1) issue_tile(tile + 2, b); }
            // odd: softmax(S(tile,1)) | S(tile+1,0) | O += V^T(tile,0) P
            pv(b, 0, nh, nl);
            qk(last ? b : b ^ 1, 0, s1, s2);
            attn_softmax_step<MASK>(n1, n2, tile * AK + 32, nk, lane, m_run, l_run, ph, pl, alpha, rescale, qvalid, p_single);
            ATTN_INTERLEAVE();
            if (rescale) { o1a *= alpha; o2a *= alpha; o1b *= alpha; o2b *= alpha; }
        };
        const bool ragged = (nk & (AK - 1)) != 0;            // only the last tile of the image can be
        const int tfull = (ragged && t1 == ntiles) ? t1 - 1 : t1;
        for (int tile = t0; tile < tfull; ++tile) tile_body(std::false_type{}, tile);
        if (tfull < t1) tile_body(std::true_type{}, t1 - 1);
        pv((t1 - 1 - t0) & 1, 1, ph, pl);        // the last sub-step's P
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const int qrow = q0 + wave * 32 + lr;
    if (p.KS == 1) {
        // the block saw every key of its queries: normalise, split and write the context planes
        // (k-panel layout: 4 consecutive d of one token = 8 bytes per plane)
        if (qrow < nq) {
            const float inv = 1.0f / l_tot;
            const int prow = img * p.Kc + qrow;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    half4 hh, ll;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = half ? (o1b[4 * g4 + e] + o2b[4 * g4 + e] * SPLIT_INV)
                                             : (o1a[4 * g4 + e] + o2a[4 * g4 + e] * SPLIT_INV);
                        _Float16 a, b;
                        split_f32(v * inv, a, b, range_flag_of(p.ctrl, img));
                        hh[e] = a; ll[e] = b;
                    }
                    const size_t o = panel_index(prow, head * DH + 32 * half + 8 * g4 + 4 * h, p.NIc * p.Kc);
                    *reinterpret_cast<half4*>(p.msg.hi + o) = hh;
                    *reinterpret_cast<half4*>(p.msg.lo + o) = ll;
                }
            }
        }
        return;
    }
    const size_t pbase0 = (((size_t)z * p.NIc + img) * NH + head) * p.Kc + q0 + wave * 32;   // this wave's first row
    if (qrow < nq) {
        const size_t pbase = pbase0 + lr;
        float* op = p.o_part + pbase * DH;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            float4 a, b;
            a.x = o1a[4 * g4] + o2a[4 * g4] * SPLIT_INV; a.y = o1a[4 * g4 + 1] + o2a[4 * g4 + 1] * SPLIT_INV;
            a.z = o1a[4 * g4 + 2] + o2a[4 * g4 + 2] * SPLIT_INV; a.w = o1a[4 * g4 + 3] + o2a[4 * g4 + 3] * SPLIT_INV;
            b.x = o1b[4 * g4] + o2b[4 * g4] * SPLIT_INV; b.y = o1b[4 * g4 + 1] + o2b[4 * g4 + 1] * SPLIT_INV;
            b.z = o1b[4 * g4 + 2] + o2b[4 * g4 + 2] * SPLIT_INV; b.w = o1b[4 * g4 + 3] + o2b[4 * g4 + 3] * SPLIT_INV;
            *reinterpret_cast<float4*>(op + 8 * g4 + 4 * h) = a;
            *reinterpret_cast<float4*>(op + 32 + 8 * g4 + 4 * h) = b;
        }
        if (h == 0) { p.m_part[pbase] = m_run; p.l_part[pbase] = l_tot; }
    }
}

// merge key-split partials -> split planes of msg[img][row][head*64 + d]
__global__ __launch_bounds__(256) void lg_attn_merge_h_kernel(const float* __restrict__ o_part,
                                                              const float* __restrict__ m_part,
                                                              const float* __restrict__ l_part, SplitOut msg,
                                                              int KS, int Kc, int NI, int NIc,
                                                              const LGCtrl* __restrict__ ctrl) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c4 = (int)(gid & 15);
    const long rid = gid >> 4;
    if (rid >= (long)NI * NH * Kc) return;
    const int row = (int)(rid % Kc), ih = (int)(rid / Kc), img = ih >> 2, head = ih & 3;
    if (ctrl_of(ctrl, img).stop || row >= n_of(ctrl, img)) return;
    const size_t zs = (size_t)NIc * NH * Kc;
    float M = -INFINITY;
    for (int z = 0; z < KS; ++z) M = fmaxf(M, m_part[(size_t)z * zs + rid]);
    float4 acc = make_float4(0, 0, 0, 0);
    float L = 0.0f;
    for (int z = 0; z < KS; ++z) {
        const size_t pb = (size_t)z * zs + rid;
        const float mz = m_part[pb];
        const float wz = (mz == -INFINITY) ? 0.0f : exp2f(mz - M);
        const float4 o = *reinterpret_cast<const float4*>(o_part + pb * DH + c4 * 4);
        acc.x += o.x * wz; acc.y += o.y * wz; acc.z += o.z * wz; acc.w += o.w * wz;
        L += l_part[pb] * wz;
    }
    const float inv = 1.0f / L;
    const float v[4] = {acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv};
    half4 hh, ll;
#pragma unroll
    for (int e = 0; e < 4; ++e) { _Float16 a, b; split_f32(v[e], a, b, range_flag_of(ctrl, img)); hh[e] = a; ll[e] = b; }
    const size_t o = panel_index(img * Kc + row, head * DH + c4 * 4, NIc * Kc);
    *reinterpret_cast<half4*>(msg.hi + o) = hh;
    *reinterpret_cast<half4*>(msg.lo + o) = ll;
}

}  // namespace

// ======================================================================== //
//  host side
// ======================================================================== //
// out_proj / to_out are folded into w1 / cw1 at pack time (weights.py _fold_out_proj)
struct LGLayerW {
    const float *wqkv, *bqkv, *w1, *b1, *lnw, *lnb, *w2, *b2;
    const float *cqkv, *cbqkv, *cw1, *cb1, *clnw, *clnb, *cw2, *cb2;
};

struct sslam_lightglue {
    sslam_ctx* ctx = nullptr;
    int Kc = 0;
    int NB = 1;                      // batch capacity in pairs; NIc = 2 NB images
    int NIc = 2;
    int KSmax = 4;                   // key split the attention partial buffers are sized for
    float depth_conf = 0.95f, width_conf = 0.99f, filter_thr = 0.1f;
    int prune_min = -1;
    sslam::Arena arena;
    float* blob = nullptr;
    const float *w_in, *b_in, *w_r;
    LGLayerW L[NL];
    const float *fp_w, *fp_b, *mt_w, *mt_b;   // layer 0; strides below
    long fp_stride = 0, mt_stride = 0;
    const float* tc_w[NL - 1];
    const float* tc_b[NL - 1];
    // workspace
    LGCtrl* ctrl;
    int* range_sticky;               // device word: some pair of some call raised its range flag since the last read
    float *x, *enc_cos, *enc_sin, *q, *k, *v, *msg, *hid, *tx, *tc, *ts;
    float *o_part, *m_part, *l_part, *conf, *mat, *md, *sim, *rmax, *rlog, *cmax, *clog, *best0;
    float *cpmax, *cpsum, *cpval, *bbox;
    int* cparg;
    int *ind, *gmap, *prune, *arg0, *arg1;
    float *in_xy, *in_desc, *up_xy, *up_desc, *out_score;
    int32_t *out_ij, *out_info;
    // split-precision planes (precision == 1)
    int precision = 1;               // 0: fp32 MFMA everywhere; 1: fp16 hi/lo split planes (three MFMAs per product; with p_single two in P.V)
    bool sim_exact = false;          // SSLAM_LG_SIM_EXACT=1 (experiments): the similarity GEMM stays on the fp32 matrix instruction at precision 1
    int dbg_layers = NL;             // test hook: run only the first dbg_layers layers
    int dbg_self_only = 0;           // test hook: stop after the self block of the last executed layer
    int force_ks = 0;                // test hook: key split of the attention launches (0 = by batch size)
    hipError_t launch_error = hipSuccess;   // first failure of a module-API launch (checked with hipGetLastError at the end of an enqueue)
    int p_single = 1;                // precision "f16x3p1" (set_precision 2, the DEFAULT since r05 - profiles/r05_flip_soak.md): P as one fp16 plane in P.V, row sums over the rounded weights
    int study = 0;                   // precision study (sslam_lightglue_debug_split_form): which cross terms of the split products are dropped
    _Float16* zero_plane = nullptr;  // study only: an all-zero fp16 plane standing in for a dropped low plane
    int big_gemm = -1;               // test hook: -1 by batch size, 0 / 1 force the single-pair (ring) / batched form of the linears;
                                     // 2 / 3: batched form with 64- / 32-token FFN tiles whatever the size
    _Float16 *w_hi, *w_lo;           // whole weight blob, split
    _Float16 *ffn_w1f[NL][2], *ffn_w2f[NL][2];   // fused-FFN fragment-order weights, [layer][self / cross]
    _Float16 *xs_hi, *xs_lo, *msgs_hi, *msgs_lo, *hids_hi, *hids_lo;
    _Float16 *qs_hi, *qs_lo, *ks_hi, *ks_lo, *vts_hi, *vts_lo;
    size_t n_blob = 0;
    // optional HIP-event bracketing of the attention launches (bench.py roofline line)
    bool profile = false;
    std::vector<hipEvent_t> ev;     // start/stop pairs
    size_t ev_used = 0;
    int last_pairs = 1;             // pairs of the bracketed launches
    bool use_graphs = false;        // replay the launch sequence of a (batched) device call as a cached hipGraph
    sslam::GraphCache graphs;
    void settings_changed() { if (!graphs.entries.empty()) { (void)hipStreamSynchronize(ctx->stream); graphs.clear(); } }
};

namespace {

size_t pad64(size_t n) { return (n + 63) / 64 * 64; }

int lg_bind_weights(sslam_lightglue* g, size_t n_floats) {
    size_t off = 0;
    auto take = [&](size_t n) { const float* p = g->blob + off; off += pad64(n); return p; };
    g->w_in = take((size_t)D * DIN); g->b_in = take(D); g->w_r = take(ENC * 2);
    for (int i = 0; i < NL; ++i) {
        LGLayerW& l = g->L[i];
        l.wqkv = take(3 * D * D); l.bqkv = take(3 * D);
        l.w1 = take(4 * D * D); l.b1 = take(2 * D); l.lnw = take(2 * D); l.lnb = take(2 * D);
        l.w2 = take(2 * D * D); l.b2 = take(D);
        l.cqkv = take(2 * D * D); l.cbqkv = take(2 * D);
        l.cw1 = take(4 * D * D); l.cb1 = take(2 * D); l.clnw = take(2 * D); l.clnb = take(2 * D);
        l.cw2 = take(2 * D * D); l.cb2 = take(D);
    }
    for (int i = 0; i < NL; ++i) {
        const float* fw = take(D * D); const float* fb = take(D);
        const float* mw = take(D); const float* mb = take(1);
        if (i == 0) { g->fp_w = fw; g->fp_b = fb; g->mt_w = mw; g->mt_b = mb; }
        if (i == 1) { g->fp_stride = fw - g->fp_w; g->mt_stride = mw - g->mt_w; }
    }
    for (int i = 0; i < NL - 1; ++i) { g->tc_w[i] = take(D); g->tc_b[i] = take(1); }
    SSLAM_REQUIRE(off == n_floats, "sslam_lightglue_create: weight blob has %zu floats, expected %zu",
                  n_floats, off);
    return 0;
}

// read and clear THIS instance's sticky range flag of the split-precision path (set by lg_emit_kernel when a
// pair's LGCtrl::range_overflow was raised; instances never share it)
int lg_take_range_flag(sslam_lightglue* g, int* flag_out) {
    hipStream_t s = g->ctx->stream;
    int flag = 0;
    SSLAM_HIP_CHECK(hipMemcpyAsync(&flag, g->range_sticky, sizeof(int), hipMemcpyDeviceToHost, s));
    SSLAM_HIP_CHECK(hipStreamSynchronize(s));
    if (flag) {
        SSLAM_HIP_CHECK(hipMemsetAsync(g->range_sticky, 0, sizeof(int), s));
        SSLAM_HIP_CHECK(hipStreamSynchronize(s));
    }
    *flag_out = flag;
    return 0;
}

float conf_threshold(int layer) {   // np.clip(0.8 + 0.1 * exp(-4 i / n_layers), 0, 1), cast to fp32
    double v = 0.8 + 0.1 * exp(-4.0 * layer / (double)NL);
    v = v < 0 ? 0 : (v > 1 ? 1 : v);
    return (float)v;
}

// key split of one attention launch: enough (image, head, query-block) units to give every CU two
// workgroups without it, otherwise split the keys (partials + merge launch)
int attn_key_split(const sslam_lightglue* g, int NI) {
    int ks = 1;
    if (g->force_ks > 100) ks = g->force_ks - 100;     // (test hook: forced split, assembly kernel)
    else if (g->force_ks > 0) ks = g->force_ks;        // (test hook: forced split, the 4-wave r02 kernel)
    else if (g->force_ks < 0 && g->force_ks > -4) ks = 1;    // (test hooks, no key split: -1 the 4-wave r02 kernel, -3 the assembly kernel)
    else {                                             // 0, -4 = the same policy on the 4-wave r02 kernel, -5 = on the assembly kernel with the merge as a launch
        // one workgroup per CU is the measured optimum of the assembly kernel (2048-keypoint pair, 128 units: no split 1.68 ms
        // per forward, 2 ranges 1.59, 4 ranges 1.65; the 4-wave kernel 1.69 at 2 or 4)
        const int units = NI * NH * (g->Kc / AQ);
        while (ks < 4 && units * ks < 256 && g->Kc / AK >= 2 * ks * 4) ks *= 2;
    }
    return ks > g->KSmax ? g->KSmax : ks;
}

template <int BM, int BN, int TM, int TN, int EPI>
void launch_linear(hipStream_t s, const LinearArgs& a) {
    dim3 grid(a.N / BN, a.NI * sslam::cdiv(a.Kc, BM));
    hipLaunchKernelGGL((lg_linear_kernel<BM, BN, TM, TN, EPI>), grid, dim3(256), 0, s, a);
}

LinearArgs lin(const sslam_lightglue* g, int NI, const float* A0, int lda0, const float* A1, int lda1, int K0,
               int K, const float* W, const float* b, int N) {
    LinearArgs a{};
    a.A0 = A0; a.lda0 = lda0; a.A1 = A1; a.lda1 = lda1; a.K0 = K0; a.K = K;
    a.W = W; a.bias = b; a.N = N; a.out_scale = 1.0f; a.ctrl = g->ctrl; a.Kc = g->Kc; a.NI = NI;
    a.enc_cos = g->enc_cos; a.enc_sin = g->enc_sin;
    return a;
}

void attn_event(sslam_lightglue* g, hipStream_t s, bool start) {
    if (!g->profile) return;
    if (start && g->ev_used + 2 > g->ev.size())
        for (int i = 0; i < 64; ++i) { hipEvent_t e; (void)hipEventCreate(&e); g->ev.push_back(e); }
    (void)hipEventRecord(g->ev[g->ev_used + (start ? 0 : 1)], s);
    if (!start) g->ev_used += 2;
}

void launch_attention(sslam_lightglue* g, hipStream_t s, int NI, const float* Q, const float* K,
                      const float* V, int cross) {
    const int KS = attn_key_split(g, NI);
    AttnArgs a{Q, K, V, cross, g->o_part, g->m_part, g->l_part, KS, g->Kc, g->NIc, g->ctrl};
    dim3 grid(sslam::cdiv(g->Kc, AQ), NI * NH, KS);
    attn_event(g, s, true);
    hipLaunchKernelGGL(lg_attention_kernel, grid, dim3(256), 0, s, a);
    attn_event(g, s, false);
    const long n4 = (long)NI * NH * g->Kc * 16;
    hipLaunchKernelGGL(lg_attn_merge_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s,
                       g->o_part, g->m_part, g->l_part, g->msg, KS, g->Kc, NI, g->NIc, g->ctrl);
}

void launch_ffn(sslam_lightglue* g, hipStream_t s, int NI, const float* message, const float* w1,
                const float* b1, const float* lnw, const float* lnb, const float* w2, const float* b2) {
    // hid = [x | message] W1^T + b1 ; LN + GELU ; x += hid W2^T + b2
    LinearArgs a = lin(g, NI, g->x, D, message, D, D, 2 * D, w1, b1, 2 * D);
    a.out = g->hid; a.ldo = 2 * D;
    launch_linear<64, 128, 1, 2, EPI_PLAIN>(s, a);
    hipLaunchKernelGGL(lg_ln_gelu_kernel, dim3(sslam::cdiv(NI * g->Kc, 4)), dim3(256), 0, s, g->hid, lnw,
                       lnb, g->ctrl, g->Kc, NI);
    LinearArgs c = lin(g, NI, g->hid, 2 * D, nullptr, 0, 2 * D, 2 * D, w2, b2, D);
    c.out = g->x; c.ldo = D;
    launch_linear<64, 64, 1, 1, EPI_RESID>(s, c);
}

SplitPtr wsp(const sslam_lightglue* g, const float* w) {
    const size_t off = (size_t)(w - g->blob);
    return SplitPtr{g->w_hi + off, g->w_lo + off};
}

// big-tile launch (batched token sets): 128 x 128 tiles, 4 waves
template <int BM, int BN, int WM, int WN, int EPI>
void launch_linear_big(hipStream_t s, int NI, const LinearArgsH& a) {
    constexpr int NW = WM * WN;
    constexpr size_t stage = (size_t)sslam::BIG_STAGES * sslam::big_stage_halves<BM, BN>() * sizeof(_Float16);
    constexpr size_t epi = (size_t)BM * (BN + 4) * sizeof(float);
    constexpr size_t lds = stage > epi ? stage : epi;
    static_assert(lds <= 160 * 1024, "LDS budget");
    if (!a.ctrl) {          // configuration call (instance creation): > 64 KiB of dynamic LDS needs the opt-in
        (void)hipFuncSetAttribute((const void*)lg_linear_big_kernel<BM, BN, WM, WN, EPI>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return;
    }
    dim3 grid(a.N / BN, NI * sslam::cdiv(a.Kc, BM));
    hipLaunchKernelGGL((lg_linear_big_kernel<BM, BN, WM, WN, EPI>), grid, dim3(NW * 64), lds, s, a);
}

template <int BM, int BN, int TM, int TN, int EPI>
void launch_linear_h(hipStream_t s, int NI, const LinearArgsH& a) {
    constexpr size_t lds = (size_t)ring_depth<BM, BN>() * sslam::ring_stage_halves<BM, BN>() * sizeof(_Float16);
    if (!a.ctrl) {          // configuration call (instance creation): > 64 KiB of dynamic LDS needs the opt-in
        (void)hipFuncSetAttribute((const void*)lg_linear_h_kernel<BM, BN, TM, TN, EPI>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return;
    }
    dim3 grid(a.N / BN, NI * sslam::cdiv(a.Kc, BM));
    hipLaunchKernelGGL((lg_linear_h_kernel<BM, BN, TM, TN, EPI>), grid, dim3(512), lds, s, a);   // 4 consumer + 4 producer waves
}

LinearArgsH linh(const sslam_lightglue* g, SplitPtr A0, SplitPtr A1, int lda, int K0, int K, const float* W,
                 const float* b, int N) {
    LinearArgsH a{};
    a.A0 = A0; a.A1 = A1; a.lda = lda; a.K0 = K0; a.K = K;
    a.W = wsp(g, W); a.bias = b; a.N = N; a.ctrl = g->ctrl; a.Kc = g->Kc; a.NIc = g->NIc;
    a.enc_cos = g->enc_cos; a.enc_sin = g->enc_sin; a.q_scale = 1.0f; a.k_scale = 1.0f;
    return a;
}

// ---- the hand-scheduled form of the split-precision attention (lg_attention_p_kernel's arithmetic in half-steps): gfx950 assembly printed by csrc/gen_lg_attention_asm.py,
// assembled and embedded by build.py (`sslam_lg_attention_asm_hsaco`), loaded once per device at instance creation
// (never inside a stream capture) and launched through the module API - which a capture records like any launch.
extern "C" const unsigned char sslam_lg_attention_asm_hsaco[];
extern "C" const unsigned char sslam_lg_attention_asm_p1_hsaco[];     // the kernel of precision "f16x3p1" (gen_lg_attention_asm_p1.py)
struct AttnAsmArgs {                // the kernel's argument segment (gen_lg_attention_asm.py: s_load offsets 0x0 .. 0x5c)
    const _Float16 *q_hi, *q_lo, *k_hi, *k_lo, *vt_hi, *vt_lo;
    _Float16 *msg_hi, *msg_lo;
    const LGCtrl* ctrl;
    int cross, Kc, NIc, nqb, nslab; // nslab = key ranges x images x heads = gridDim.y
    unsigned magic;                 // floor(2^32 / nqb) + 1: idx / nqb = mulhi(idx, magic) for idx * nqb < 2^32 (nqb > 1)
    int nih, lks;                   // images x heads ; log2(key ranges): 0 = the kernel normalises and writes the context planes
    float *o_part, *m_part, *l_part;   // lks > 0: the unnormalised (o, m, l) of a key range, lg_attention_p_kernel's layout
};
static_assert(sizeof(AttnAsmArgs) == 128, "kernel argument segment of lg_attention_asm_kernel");
constexpr int ASM_MAX_DEVICES = 64;
hipFunction_t g_attn_asm_fn[ASM_MAX_DEVICES] = {};
hipFunction_t g_attn_asm_p1_fn[ASM_MAX_DEVICES] = {};
std::mutex g_attn_asm_mutex;

int lg_load_attention_asm(int device) {
    SSLAM_REQUIRE(device >= 0 && device < ASM_MAX_DEVICES, "device %d out of range", device);
    std::lock_guard<std::mutex> lock(g_attn_asm_mutex);
    if (g_attn_asm_fn[device]) return 0;
    hipModule_t mod, mod1;
    SSLAM_HIP_CHECK(hipModuleLoadData(&mod, sslam_lg_attention_asm_hsaco));
    SSLAM_HIP_CHECK(hipModuleLoadData(&mod1, sslam_lg_attention_asm_p1_hsaco));
    SSLAM_HIP_CHECK(hipModuleGetFunction(&g_attn_asm_p1_fn[device], mod1, "lg_attention_asm_p1_kernel"));
    SSLAM_HIP_CHECK(hipModuleGetFunction(&g_attn_asm_fn[device], mod, "lg_attention_asm_kernel"));
    return 0;
}

void launch_attention_h(sslam_lightglue* g, hipStream_t s, int NI, SplitPtr Q, SplitPtr K, SplitPtr VT, int cross,
                        bool merge_in_ffn = false) {
    const int KS = attn_key_split(g, NI);
    if (g->study) {       // precision study: a dropped cross term = its low-plane operand replaced by zeros (bit-identical to not issuing the MFMA)
        if (g->study & 0x01) K.lo = g->zero_plane;          // S = kh.qh + kh.ql          (K as one fp16 plane)
        if (g->study & 0x02) Q.lo = g->zero_plane;          // S = kh.qh + kl.qh          (Q as one fp16 plane)
        if (g->study & 0x08) VT.lo = g->zero_plane;         // O = vh.ph + vh.pl          (V as one fp16 plane)
    }
    AttnArgsH a{Q, K, VT, cross, g->o_part, g->m_part, g->l_part, SplitOut{g->msgs_hi, g->msgs_lo}, KS, g->Kc,
                g->NIc, g->ctrl, ((g->study & 0x04) || g->p_single) ? 1 : 0};     // O = vh.ph + vl.ph: study bit 0x04 (4-wave kernel only) or precision "f16x3p1"
    attn_event(g, s, true);
    if (!(g->study & 0x04) && (g->force_ks == 0 || g->force_ks == -3 || g->force_ks == -5 || g->force_ks > 100)) {
        // the hand-scheduled assembly kernel - the arithmetic, LDS images and results of lg_attention_p_kernel (no key split: batched
        // launches, debug_key_split(lg, -3)) and of lg_attention_p_kernel's key ranges (single pairs) bit for bit, 8 - 10 % faster
        // (profiles/r03_attention_experiments.md)
        const int lks = KS == 4 ? 2 : KS == 2 ? 1 : 0;
        AttnAsmArgs k{Q.hi, Q.lo, K.hi, K.lo, VT.hi, VT.lo, g->msgs_hi, g->msgs_lo, g->ctrl, cross, g->Kc, g->NIc,
                      sslam::cdiv(g->Kc, AQ), NI * NH * KS, 0u, NI * NH, lks, g->o_part, g->m_part, g->l_part};
        k.magic = k.nqb > 1 ? (unsigned)((1ull << 32) / (unsigned)k.nqb + 1) : 0u;
        size_t sz = sizeof(k);
        void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &k, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        const hipFunction_t fn = g->p_single ? g_attn_asm_p1_fn[g->ctx->device] : g_attn_asm_fn[g->ctx->device];
        const hipError_t e = hipModuleLaunchKernel(fn, (unsigned)k.nqb, (unsigned)k.nslab, 1,
                                                   256, 1, 1, 0, s, nullptr, cfg);
        if (e != hipSuccess && g->launch_error == hipSuccess) g->launch_error = e;
    } else {
        // key split, or debug_key_split(lg, -1): the r02 4-wave kernel
        dim3 grid(sslam::cdiv(g->Kc, AQ), NI * NH, KS);
        hipLaunchKernelGGL(lg_attention_p_kernel, grid, dim3(256), 0, s, a);
    }
    attn_event(g, s, false);
    if (KS == 1 || merge_in_ffn) return;   // the kernel wrote the context planes itself / the fused FFN merges the partials
    const long n4 = (long)NI * NH * g->Kc * 16;
    hipLaunchKernelGGL(lg_attn_merge_h_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, g->o_part,
                       g->m_part, g->l_part, SplitOut{g->msgs_hi, g->msgs_lo}, KS, g->Kc, NI, g->NIc, g->ctrl);
}

// one transformer layer (self + cross block) on the split-precision path
// `heads`: the layer's cross-block FFN also evaluates the token heads on the new state (batched form only; returns whether
// it did - otherwise lg_token_heads_kernel has to run)
bool lg_layer_h(sslam_lightglue* g, hipStream_t s, int NI, const LGLayerW& l, int layer, bool self_only, bool heads) {
    const SplitPtr xs{g->xs_hi, g->xs_lo}, msgs{g->msgs_hi, g->msgs_lo};
    const SplitPtr hids{g->hids_hi, g->hids_lo}, none{nullptr, nullptr};
    const SplitPtr qs{g->qs_hi, g->qs_lo}, ks{g->ks_hi, g->ks_lo}, vts{g->vts_hi, g->vts_lo};
    // precision study (sslam_lightglue_debug_split_form): W.x products with one cross term dropped - the READ side of the
    // activation (0x10 projections, 0x20 FFN) or weight (0x40 projections, 0x80 FFN) low plane is an all-zero plane
    const int st = g->study;
    auto act = [&](SplitPtr a, int bit) { if (st & bit) a.lo = g->zero_plane; return a; };
    auto wgt = [&](LinearArgsH& a, int bit) { if (st & bit) a.W.lo = g->zero_plane; };
    const unsigned tokblocks = sslam::cdiv(NI * g->Kc, 4);
    const float sm_scale = 0.125f * 1.4426950408889634f;        // 1/sqrt(64) * log2(e)
    // enough token rows for 128-row tiles to fill the chip (a batch of pairs): 128 x 128 projections and the
    // whole FFN as ONE kernel (ffn_fused.hpp); a single pair keeps the 64-row ring kernels (r01 form)
    const bool big = g->big_gemm >= 0 ? g->big_gemm != 0 : ((long)NI * g->Kc >= 4096 && g->Kc % 128 == 0);
    // one pair: the attention's key-range partials are merged by the FFN tiles (ffn_fused.hpp FOLD), not by a launch of their own
    const int KS = attn_key_split(g, NI);
    const bool small_tiles = g->big_gemm == 3 || (g->big_gemm != 2 && NI * (g->Kc / 64) <= 128);
    const bool fold = big && small_tiles && KS > 1 && st == 0 && (g->force_ks == 0 || g->force_ks > 100);   // (test hooks -4 / -5: the merge launch)
    auto ffn = [&](int cross, const float* w1, const float* b1, const float* lnw, const float* lnb, const float* w2,
                   const float* b2) {
        if (big) {
            FfnKArgs k{};
            k.f.xs = act(xs, 0x20); k.f.msgs = act(msgs, 0x20); k.f.plane_rows = g->NIc * g->Kc;   // (study: the hidden planes are internal - FFN-2 keeps 3 terms here)
            k.f.w1f = g->ffn_w1f[layer][cross]; k.f.b1 = b1; k.f.ln_w = lnw; k.f.ln_b = lnb;
            k.f.w2f = g->ffn_w2f[layer][cross]; k.f.b2 = b2;
            k.f.x = g->x; k.f.xo_hi = g->xs_hi; k.f.xo_lo = g->xs_lo; k.f.stamps = nullptr;
            if (cross && heads) {
                const bool do_stop = g->depth_conf > 0.0f;
                k.f.hm = g->mt_w + (size_t)layer * g->mt_stride; k.f.hm_b = g->mt_b + (size_t)layer * g->mt_stride;
                k.f.hc = do_stop ? g->tc_w[layer] : nullptr; k.f.hc_b = do_stop ? g->tc_b[layer] : nullptr;
                k.f.conf_thr = conf_threshold(layer); k.f.conf = g->conf; k.f.mat = g->mat;
            }
            k.ctrl = g->ctrl; k.Kc = g->Kc;
            if (fold) {                            // one pair, keys split into ranges: the tile merges the partials itself
                k.f.o_part = g->o_part; k.f.m_part = g->m_part; k.f.l_part = g->l_part;
                k.f.ks = KS; k.f.part_zs = (long)g->NIc * NH * g->Kc;
                hipLaunchKernelGGL((lg_ffn_fused_kernel<1, true>), dim3(NI * (g->Kc / 32)), dim3(512), sslam::FFN_LDS_BYTES, s, k);
            } else if (small_tiles)                // too few 64-token tiles for the chip: 32-token tiles (same results)
                hipLaunchKernelGGL(lg_ffn_fused_kernel<1>, dim3(NI * (g->Kc / 32)), dim3(512), sslam::FFN_LDS_BYTES, s, k);
            else
                hipLaunchKernelGGL(lg_ffn_fused_kernel<2>, dim3(NI * (g->Kc / 64)), dim3(512), sslam::FFN_LDS_BYTES, s, k);
            return;
        }
        LinearArgsH a = linh(g, act(xs, 0x20), act(msgs, 0x20), D, D, 2 * D, w1, b1, 2 * D);      // [x | attention context]
        wgt(a, 0x80);
        a.out = g->hid; a.ldo = 2 * D;
        launch_linear_h<64, 128, 1, 2, EPH_F32>(s, NI, a);
        hipLaunchKernelGGL(lg_ln_gelu_h_kernel, dim3(tokblocks), dim3(256), 0, s, g->hid,
                           SplitOut{g->hids_hi, g->hids_lo}, lnw, lnb, g->ctrl, g->Kc, NI, g->NIc);
        LinearArgsH c = linh(g, act(hids, 0x20), none, 2 * D, 2 * D, 2 * D, w2, b2, D);
        wgt(c, 0x80);
        c.out = g->x; c.ldo = D; c.outs = SplitOut{g->xs_hi, g->xs_lo};
        launch_linear_h<64, 64, 1, 1, EPH_RESID>(s, NI, c);
    };
    {   // self block
        LinearArgsH a = linh(g, act(xs, 0x10), none, D, D, D, l.wqkv, l.bqkv, 3 * D);
        wgt(a, 0x40);
        a.q = SplitOut{g->qs_hi, g->qs_lo}; a.k = SplitOut{g->ks_hi, g->ks_lo}; a.vt = SplitOut{g->vts_hi, g->vts_lo};
        a.q_scale = sm_scale; a.k_scale = 1.0f;
        if (big) launch_linear_big<128, 128, 2, 2, EPH_QKV>(s, NI, a);
        else launch_linear_h<64, 192, 1, 3, EPH_QKV>(s, NI, a);  // 768 / 192 = 4 column tiles
    }
    launch_attention_h(g, s, NI, qs, ks, vts, 0, fold);
    ffn(0, l.w1, l.b1, l.lnw, l.lnb, l.w2, l.b2);
    if (self_only) return false;
    {   // cross block: the shared qk projection is both query and key -> sqrt(scale) on it
        LinearArgsH a = linh(g, act(xs, 0x10), none, D, D, D, l.cqkv, l.cbqkv, 2 * D);
        wgt(a, 0x40);
        a.q = SplitOut{g->qs_hi, g->qs_lo}; a.vt = SplitOut{g->vts_hi, g->vts_lo};
        a.q_scale = sqrtf(sm_scale);
        if (big) launch_linear_big<128, 128, 2, 2, EPH_CROSS>(s, NI, a);
        else launch_linear_h<64, 128, 1, 2, EPH_CROSS>(s, NI, a);
    }
    launch_attention_h(g, s, NI, qs, qs, vts, 1, fold);
    ffn(1, l.cw1, l.cb1, l.clnw, l.clnb, l.cw2, l.cb2);
    return big && heads;
}

// Enqueue one batch of `pairs` pairs on the context stream.  `src` names the inputs of image
// 2p (query side) and 2p+1 of every pair; outputs of pair p go to ij_out + p*out_stride*2,
// score_out + p*out_stride, info_out + 4p.
int lg_enqueue(sslam_lightglue* g, int pairs, const StageSrc& src, float min_conf, int32_t* ij_out,
               float* score_out, int32_t* info_out, long out_stride) {
    hipStream_t s = g->ctx->stream;
    (void)hipGetLastError();     // (a stale error of another library on this thread - e.g. RCCL's probes - is not ours)
    const int Kc = g->Kc, NI = 2 * pairs;
    hipLaunchKernelGGL(lg_prepare_kernel, dim3(NI, 8), dim3(1024), 0, s, src, Kc, g->in_xy, g->in_desc, g->bbox,
                       g->ind, g->prune, g->ctrl);
    hipLaunchKernelGGL(lg_posenc_kernel, dim3(sslam::cdiv(NI * Kc * ENC, 256)), dim3(256), 0, s, g->in_xy,
                       g->bbox, g->w_r, g->enc_cos, g->enc_sin, Kc, NI, g->ctrl);
    const bool proj_h = g->precision == 1 && !g->sim_exact;       // the projections and the similarity GEMM on the split pipe
    constexpr size_t big_lds = (size_t)sslam::BIG_STAGES * sslam::big_stage_halves<128, 128>() * sizeof(_Float16);
    const dim3 biggrid(D / 128, sslam::cdiv(Kc, 128), NI);
    if (proj_h) {   // input_proj: descriptors split into the (idle) k planes (k-panels), x and its planes from the epilogue
        hipLaunchKernelGGL(lg_split_rows_kernel, dim3(SPLIT_BLOCKS_PER_IMAGE, NI), dim3(256), 0, s, g->in_desc, g->ks_hi, g->ks_lo, DIN, Kc,
                           NI, g->NIc, g->ctrl, 0);
        ProjBigArgs a{};
        a.a_hi = g->ks_hi; a.a_lo = g->ks_lo; a.a_rows = g->NIc * Kc; a.K = DIN;
        a.w_hi = g->w_hi + (g->w_in - g->blob); a.w_lo = g->w_lo + (g->w_in - g->blob); a.bias = g->b_in;
        a.out_scale = 1.0f; a.out = g->x; a.o_hi = g->xs_hi; a.o_lo = g->xs_lo; a.o_rows = g->NIc * Kc;
        a.ctrl = g->ctrl; a.Kc = Kc;
        hipLaunchKernelGGL((lg_proj_big_kernel<128, 128, 2, 2>), biggrid, dim3(256), big_lds, s, a);
    } else {   // input_proj (lightglue.py: desc = self.input_proj(desc))
        LinearArgs a = lin(g, NI, g->in_desc, DIN, nullptr, 0, DIN, DIN, g->w_in, g->b_in, D);
        a.out = g->x; a.ldo = D;
        launch_linear<64, 64, 1, 1, EPI_PLAIN>(s, a);
    }
    const unsigned tokblocks = sslam::cdiv(NI * Kc, 4);
    const dim3 headgrid(sslam::cdiv(Kc, 256), NI);               // one lane per token, one image per grid row
    const dim3 splitblocks(SPLIT_BLOCKS_PER_IMAGE, NI);
    if (g->precision == 1 && !proj_h)
        hipLaunchKernelGGL(lg_split_rows_kernel, splitblocks, dim3(256), 0, s, g->x, g->xs_hi, g->xs_lo, D, Kc,
                           NI, g->NIc, g->ctrl, 0);
    for (int i = 0; i < g->dbg_layers; ++i) {
        const LGLayerW& l = g->L[i];
        const bool self_only = g->dbg_self_only && i == g->dbg_layers - 1;
        // the token heads of this layer (early stop + pruning) ride in the cross block's fused FFN when there is one
        const bool want_heads = !(i == NL - 1 || i == g->dbg_layers - 1) && (g->depth_conf > 0.0f || g->width_conf > 0.0f);
        bool heads_done = false;
        if (g->precision == 1) {
            heads_done = lg_layer_h(g, s, NI, l, i, self_only, want_heads && g->big_gemm != 5);
        } else {
            // ---- self block
            {
                LinearArgs a = lin(g, NI, g->x, D, nullptr, 0, D, D, l.wqkv, l.bqkv, 3 * D);
                a.q = g->q; a.k = g->k; a.v = g->v;
                launch_linear<64, 128, 1, 2, EPI_QKV>(s, a);
            }
            launch_attention(g, s, NI, g->q, g->k, g->v, 0);
            launch_ffn(g, s, NI, g->msg, l.w1, l.b1, l.lnw, l.lnb, l.w2, l.b2);
            if (!self_only) {
                // ---- cross block
                {
                    LinearArgs a = lin(g, NI, g->x, D, nullptr, 0, D, D, l.cqkv, l.cbqkv, 2 * D);
                    a.q = g->q; a.v = g->v;
                    launch_linear<64, 128, 1, 2, EPI_CROSSQKV>(s, a);
                }
                launch_attention(g, s, NI, g->q, g->q, g->v, 1);
                launch_ffn(g, s, NI, g->msg, l.cw1, l.cb1, l.clnw, l.clnb, l.cw2, l.cb2);
            }
        }
        if (i == NL - 1 || i == g->dbg_layers - 1) break;
        // ---- early stop + point pruning (lightglue.py check_if_stop / get_pruning_mask)
        const int do_stop = g->depth_conf > 0.0f;
        const int do_prune = g->width_conf > 0.0f;
        if (!do_stop && !do_prune) continue;
        const float thr = conf_threshold(i);
        if (!heads_done)
            hipLaunchKernelGGL(lg_token_heads_kernel, headgrid, dim3(256), 0, s, g->x,
                               do_stop ? g->tc_w[i] : nullptr, do_stop ? g->tc_b[i] : nullptr,
                               g->mt_w + (size_t)i * g->mt_stride, g->mt_b + (size_t)i * g->mt_stride, 0L, 0,
                               thr, g->conf, g->mat, g->ctrl, Kc, do_stop);
        hipLaunchKernelGGL(lg_decide_kernel, dim3(pairs), dim3(1024), 0, s, i, thr, g->depth_conf,
                           g->width_conf, g->prune_min, do_stop, g->conf, g->mat, g->ind, g->gmap,
                           g->prune, g->ctrl, Kc);
        if (do_prune) {
            const bool sp = g->precision == 1;        // token rows moved: the copy back refreshes their split planes
            hipLaunchKernelGGL(lg_gather_kernel, dim3(tokblocks), dim3(256), 0, s, g->x, g->enc_cos,
                               g->enc_sin, g->gmap, g->tx, g->tc, g->ts, g->ctrl, Kc, NI, 0, nullptr, nullptr, g->NIc);
            hipLaunchKernelGGL(lg_gather_kernel, dim3(tokblocks), dim3(256), 0, s, g->x, g->enc_cos,
                               g->enc_sin, g->gmap, g->tx, g->tc, g->ts, g->ctrl, Kc, NI, 1, sp ? g->xs_hi : nullptr,
                               sp ? g->xs_lo : nullptr, g->NIc);
        }
    }
    // ---- assignment with log_assignment[stop_layer]
    if (proj_h) {   // final_proj of the stop layer: the token state's planes in, md and its planes (the idle q planes) out
        ProjBigArgs a{};
        a.a_hi = g->xs_hi; a.a_lo = g->xs_lo; a.a_rows = g->NIc * Kc; a.K = D;
        a.w_hi = g->w_hi + (g->fp_w - g->blob); a.w_lo = g->w_lo + (g->fp_w - g->blob); a.w_layer_stride = g->fp_stride;
        a.bias = g->fp_b; a.b_layer_stride = g->fp_stride; a.by_stop_layer = 1; a.ignore_stop = 1;
        a.out_scale = 0.25f;                    // 1 / 256^0.25
        a.out = g->md; a.o_hi = g->qs_hi; a.o_lo = g->qs_lo; a.o_rows = g->NIc * Kc; a.ctrl = g->ctrl; a.Kc = Kc;
        hipLaunchKernelGGL((lg_proj_big_kernel<128, 128, 2, 2>), biggrid, dim3(256), big_lds, s, a);
    } else {
        LinearArgs a = lin(g, NI, g->x, D, nullptr, 0, D, D, g->fp_w, g->fp_b, D);
        a.by_stop_layer = 1; a.w_layer_stride = g->fp_stride; a.b_layer_stride = g->fp_stride;
        a.ignore_stop = 1; a.out = g->md; a.ldo = D;
        a.out_scale = 0.25f;                    // 1 / 256^0.25
        launch_linear<64, 64, 1, 1, EPI_PLAIN>(s, a);
    }
    hipLaunchKernelGGL(lg_token_heads_kernel, headgrid, dim3(256), 0, s, g->x, nullptr, nullptr,
                       g->mt_w, g->mt_b, g->mt_stride, 1, 0.0f, g->conf, g->mat, g->ctrl, Kc, 0);
    if (proj_h) {
        SimBigArgs a{g->qs_hi, g->qs_lo, g->sim, Kc, g->NIc * Kc, g->ctrl};
        hipLaunchKernelGGL((lg_sim_big_kernel<128, 128, 2, 2>), dim3(sslam::cdiv(Kc, 128), sslam::cdiv(Kc, 128), pairs), dim3(256), big_lds, s, a);
    } else {
        SimArgs a{g->md, g->sim, Kc, g->ctrl};
        dim3 grid(sslam::cdiv(Kc, 128), sslam::cdiv(Kc, 64), pairs);
        hipLaunchKernelGGL((lg_sim_kernel<64, 128, 1, 2>), grid, dim3(256), 0, s, a);
    }
    hipLaunchKernelGGL(lg_row_stats_kernel, dim3(sslam::cdiv(Kc, 4), pairs), dim3(256), 0, s, g->sim, g->rmax,
                       g->rlog, Kc, g->ctrl);
    hipLaunchKernelGGL(lg_col_stats_kernel, dim3(sslam::cdiv(Kc, 64), CSLAB, pairs), dim3(256), 0, s, g->sim,
                       g->cpmax, g->cpsum, Kc, g->ctrl);
    hipLaunchKernelGGL(lg_col_stats_merge_kernel, dim3(sslam::cdiv(Kc, 256), pairs), dim3(256), 0, s, g->cpmax,
                       g->cpsum, g->cmax, g->clog, Kc, g->ctrl);
    hipLaunchKernelGGL(lg_col_argmax_kernel, dim3(sslam::cdiv(Kc, 64), CSLAB, pairs), dim3(256), 0, s, g->sim,
                       g->rmax, g->rlog, g->cmax, g->clog, g->conf, g->cpval, g->cparg, Kc, g->ctrl);
    hipLaunchKernelGGL(lg_row_argmax_kernel, dim3(sslam::cdiv(Kc, 4), pairs), dim3(256), 0, s, g->sim, g->rmax,
                       g->rlog, g->cmax, g->clog, g->conf, g->best0, g->arg0, g->cpval, g->cparg, g->arg1, Kc, g->ctrl);
    hipLaunchKernelGGL(lg_emit_kernel, dim3(pairs), dim3(1024), 0, s, g->best0, g->arg0, g->arg1, g->ind,
                       g->filter_thr, min_conf, ij_out, score_out, info_out, g->ctrl, Kc, out_stride, g->range_sticky);
    SSLAM_HIP_CHECK(hipGetLastError());
    if (g->launch_error != hipSuccess) {                 // a module-API launch (the assembly attention kernel) failed
        const hipError_t e = g->launch_error;
        g->launch_error = hipSuccess;
        SSLAM_HIP_CHECK(e);
    }
    return 0;
}

// function attributes of every linear instantiation, set once at instance creation (never inside a
// stream capture)
void lg_configure_kernels() {
    static bool done = false;
    if (done) return;
    done = true;
    const LinearArgsH cfg{};            // ctrl == nullptr: the launchers only configure
    hipStream_t s = nullptr;
    launch_linear_h<64, 128, 1, 2, EPH_F32>(s, 0, cfg);   launch_linear_h<64, 64, 1, 1, EPH_RESID>(s, 0, cfg);
    launch_linear_h<64, 192, 1, 3, EPH_QKV>(s, 0, cfg);   launch_linear_h<64, 128, 1, 2, EPH_CROSS>(s, 0, cfg);
    launch_linear_big<128, 128, 2, 2, EPH_QKV>(s, 0, cfg); launch_linear_big<128, 128, 2, 2, EPH_CROSS>(s, 0, cfg);
    {
        constexpr int big_lds = sslam::BIG_STAGES * sslam::big_stage_halves<128, 128>() * (int)sizeof(_Float16);
        (void)hipFuncSetAttribute((const void*)lg_proj_big_kernel<128, 128, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, big_lds);
        (void)hipFuncSetAttribute((const void*)lg_sim_big_kernel<128, 128, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, big_lds);
    }
    (void)hipFuncSetAttribute((const void*)lg_ffn_fused_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, sslam::FFN_LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)lg_ffn_fused_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sslam::FFN_LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)lg_ffn_fused_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, sslam::FFN_LDS_BYTES);
}

// lg_enqueue through the graph cache (device entry points only; never while profiling: the
// bracketing events are host-side records)
int lg_enqueue_cached(sslam_lightglue* g, int pairs, const StageSrc& src, float min_conf, int32_t* ij_out,
                      float* score_out, int32_t* info_out, long out_stride) {
    g->last_pairs = pairs;                 // host-side bookkeeping lives OUTSIDE the captured sequence (a replay skips the lambda)
    if (!g->use_graphs || g->profile)
        return lg_enqueue(g, pairs, src, min_conf, ij_out, score_out, info_out, out_stride);
    std::vector<uint64_t> key{(uint64_t)pairs, (uint64_t)ij_out, (uint64_t)score_out, (uint64_t)info_out,
                              (uint64_t)out_stride, 0};
    memcpy(&key[5], &min_conf, sizeof(float));
    for (int i = 0; i < 2 * pairs; ++i) {
        key.push_back((uint64_t)src.xy[i]); key.push_back((uint64_t)src.desc[i]);
        key.push_back((uint64_t)src.cnt[i]); key.push_back((uint64_t)src.bound[i]);
        uint64_t sz = 0; memcpy(&sz, &src.size_w[i], 4); memcpy((char*)&sz + 4, &src.size_h[i], 4);
        key.push_back(sz);
    }
    return sslam::run_cached(g->graphs, g->ctx->stream, key, [&] {
        return lg_enqueue(g, pairs, src, min_conf, ij_out, score_out, info_out, out_stride);
    });
}

}  // namespace

extern "C" {

int sslam_lightglue_create_batched(sslam_ctx* ctx, const float* weights, size_t n_floats, int max_kpts,
                                   int max_pairs, sslam_lightglue** out) {
    SSLAM_REQUIRE(ctx && weights && out, "sslam_lightglue_create: NULL argument");
    SSLAM_REQUIRE(max_kpts >= 1 && max_kpts <= 8192, "sslam_lightglue_create: max_kpts %d not in [1, 8192]",
                  max_kpts);
    SSLAM_REQUIRE(max_pairs >= 1 && max_pairs <= MAX_PAIRS, "sslam_lightglue_create: max_pairs %d not in [1, %d]",
                  max_pairs, MAX_PAIRS);
    SSLAM_HIP_CHECK(hipSetDevice(ctx->device));
    lg_configure_kernels();
    if (int rc = lg_load_attention_asm(ctx->device)) return rc;
    sslam_lightglue* g = new sslam_lightglue();
    g->ctx = ctx;
    const int Kc = (max_kpts + 127) / 128 * 128;     // whole attention / GEMM row blocks
    g->Kc = Kc;
    g->NB = max_pairs; g->NIc = 2 * max_pairs;
    g->KSmax = Kc >= 1024 ? 4 : (Kc >= 512 ? 2 : 1);
    if (const char* e = getenv("SSLAM_LG_SIM_EXACT")) g->sim_exact = e[0] == '1';
    const size_t K = (size_t)Kc, NI = (size_t)g->NIc, NB = (size_t)max_pairs;
    auto carve = [&](sslam::Arena& A) {
        g->blob = A.take<float>(n_floats);
        g->ctrl = A.take<LGCtrl>(NB);
        g->range_sticky = A.take<int>(4);
        g->x = A.take<float>(NI * K * D); g->msg = A.take<float>(NI * K * D);
        g->tx = A.take<float>(NI * K * D); g->md = A.take<float>(NI * K * D);
        g->hid = A.take<float>(NI * K * 2 * D);
        g->q = A.take<float>(NI * K * D); g->k = A.take<float>(NI * K * D); g->v = A.take<float>(NI * K * D);
        g->enc_cos = A.take<float>(NI * K * ENC); g->enc_sin = A.take<float>(NI * K * ENC);
        g->tc = A.take<float>(NI * K * ENC); g->ts = A.take<float>(NI * K * ENC);
        g->o_part = A.take<float>((size_t)g->KSmax * NI * NH * K * DH);
        g->m_part = A.take<float>((size_t)g->KSmax * NI * NH * K);
        g->l_part = A.take<float>((size_t)g->KSmax * NI * NH * K);
        g->conf = A.take<float>(NI * K); g->mat = A.take<float>(NI * K); g->bbox = A.take<float>(NI * 4);
        g->sim = A.take<float>(NB * K * K);
        g->rmax = A.take<float>(NB * K); g->rlog = A.take<float>(NB * K); g->cmax = A.take<float>(NB * K);
        g->clog = A.take<float>(NB * K); g->best0 = A.take<float>(NB * K);
        g->ind = A.take<int>(NI * K); g->gmap = A.take<int>(NI * K); g->prune = A.take<int>(NI * K);
        g->arg0 = A.take<int>(NB * K); g->arg1 = A.take<int>(NB * K);
        g->cpmax = A.take<float>(NB * CSLAB * K); g->cpsum = A.take<float>(NB * CSLAB * K);
        g->cpval = A.take<float>(NB * CSLAB * K); g->cparg = A.take<int>(NB * CSLAB * K);
        g->in_xy = A.take<float>(NI * K * 2); g->in_desc = A.take<float>(NI * K * DIN);
        g->up_xy = A.take<float>(2 * K * 2); g->up_desc = A.take<float>(2 * K * DIN);
        g->out_ij = A.take<int32_t>(2 * K); g->out_score = A.take<float>(K); g->out_info = A.take<int32_t>(8);
        g->w_hi = A.take<_Float16>(n_floats); g->w_lo = A.take<_Float16>(n_floats);
        for (int i = 0; i < NL; ++i)
            for (int c = 0; c < 2; ++c) {
                g->ffn_w1f[i][c] = A.take<_Float16>((size_t)2 * 4 * D * D);
                g->ffn_w2f[i][c] = A.take<_Float16>((size_t)2 * 2 * D * D);
            }
        g->xs_hi = A.take<_Float16>(NI * K * D); g->xs_lo = A.take<_Float16>(NI * K * D);
        g->msgs_hi = A.take<_Float16>(NI * K * D); g->msgs_lo = A.take<_Float16>(NI * K * D);
        g->hids_hi = A.take<_Float16>(NI * K * 2 * D); g->hids_lo = A.take<_Float16>(NI * K * 2 * D);
        g->qs_hi = A.take<_Float16>(NI * K * D); g->qs_lo = A.take<_Float16>(NI * K * D);
        g->ks_hi = A.take<_Float16>(NI * K * D); g->ks_lo = A.take<_Float16>(NI * K * D);
        g->vts_hi = A.take<_Float16>(NI * K * D); g->vts_lo = A.take<_Float16>(NI * K * D);
    };
    sslam::Arena probe;
    probe.measure();
    carve(probe);
    if (g->arena.init(probe.off + 256)) { delete g; return 1; }
    carve(g->arena);
    SSLAM_REQUIRE(g->vts_lo != nullptr, "sslam_lightglue_create: workspace arena exhausted");
    SSLAM_HIP_CHECK(hipMemcpy(g->blob, weights, n_floats * 4, hipMemcpyHostToDevice));
    if (int rc = lg_bind_weights(g, n_floats)) { g->arena.release(); delete g; return rc; }
    g->n_blob = n_floats;
    {   // split every transformer-layer weight matrix into k-panel fp16 planes (same offsets as the blob)
        auto splitw = [&](const float* w, int N, int K) {
            const size_t off = (size_t)(w - g->blob), n = (size_t)N * K;
            hipLaunchKernelGGL(lg_split_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                               w, g->w_hi + off, g->w_lo + off, N, K, g->range_sticky);
        };
        for (int i = 0; i < NL; ++i) {
            const LGLayerW& l = g->L[i];
            splitw(l.wqkv, 3 * D, D); splitw(l.w1, 2 * D, 2 * D); splitw(l.w2, D, 2 * D);
            splitw(l.cqkv, 2 * D, D); splitw(l.cw1, 2 * D, 2 * D); splitw(l.cw2, D, 2 * D);
            hipLaunchKernelGGL(lg_pack_ffn_kernel, dim3(4 * D * D / 256), dim3(256), 0, ctx->stream, l.w1, l.w2,
                               g->ffn_w1f[i][0], g->ffn_w2f[i][0], g->range_sticky);
            hipLaunchKernelGGL(lg_pack_ffn_kernel, dim3(4 * D * D / 256), dim3(256), 0, ctx->stream, l.cw1, l.cw2,
                               g->ffn_w1f[i][1], g->ffn_w2f[i][1], g->range_sticky);
        }
        splitw(g->w_in, D, DIN);                               // the input / final projections (lg_proj_big_kernel)
        for (int i = 0; i < NL; ++i) splitw(g->fp_w + (size_t)i * g->fp_stride, D, D);
        int wflag = 0;
        if (int rc = lg_take_range_flag(g, &wflag)) { g->arena.release(); delete g; return rc; }
        if (wflag) {
            g->arena.release(); delete g;
            SSLAM_REQUIRE(false, "sslam_lightglue_create: a weight with |value| >= 65520 does not fit the fp16 planes of the "
                                 "split-precision path");
        }
    }
    sslam::ctx_retain(ctx);
    *out = g;
    return 0;
}

int sslam_lightglue_create(sslam_ctx* ctx, const float* weights, size_t n_floats, int max_kpts,
                           sslam_lightglue** out) {
    return sslam_lightglue_create_batched(ctx, weights, n_floats, max_kpts, 1, out);
}

int sslam_lightglue_destroy(sslam_lightglue* g) {
    if (!g) return 0;
    (void)hipStreamSynchronize(g->ctx->stream);
    g->graphs.clear();
    for (hipEvent_t e : g->ev) (void)hipEventDestroy(e);
    if (g->zero_plane) (void)hipFree(g->zero_plane);
    g->arena.release();
    sslam_ctx* ctx = g->ctx;
    delete g;
    sslam::ctx_release(ctx);
    return 0;
}

int sslam_lightglue_set_conf(sslam_lightglue* g, float depth_confidence, float width_confidence,
                             float filter_threshold, int prune_min_kpts) {
    SSLAM_REQUIRE(g != nullptr, "sslam_lightglue_set_conf: NULL instance");
    g->settings_changed();
    g->depth_conf = depth_confidence; g->width_conf = width_confidence;
    g->filter_thr = filter_threshold; g->prune_min = prune_min_kpts;
    return 0;
}

int sslam_lightglue_match_batch_dev(sslam_lightglue* g, int n_pairs, const float* const* xy0,
                                    const float* const* desc0, const int32_t* const* m_dev, const int32_t* M,
                                    const float* const* xy1, const float* const* desc1,
                                    const int32_t* const* n_dev, const int32_t* N, float min_conf,
                                    int32_t* ij_out, float* score_out, int32_t* info_out, int out_stride) {
    SSLAM_REQUIRE(g && xy0 && desc0 && xy1 && desc1 && M && N && ij_out && score_out && info_out,
                  "sslam_lightglue_match_batch_dev: NULL argument");
    SSLAM_REQUIRE(n_pairs >= 1 && n_pairs <= g->NB, "sslam_lightglue_match_batch_dev: %d pairs, capacity %d",
                  n_pairs, g->NB);
    SSLAM_REQUIRE(out_stride >= 1, "sslam_lightglue_match_batch_dev: out_stride %d", out_stride);
    StageSrc src{};
    for (int p = 0; p < n_pairs; ++p) {
        SSLAM_REQUIRE(M[p] >= 0 && N[p] >= 0 && M[p] <= g->Kc && N[p] <= g->Kc,
                      "sslam_lightglue_match_batch_dev: pair %d: M=%d N=%d exceed max_kpts capacity %d", p, M[p],
                      N[p], g->Kc);
        SSLAM_REQUIRE((M[p] == 0 || (xy0[p] && desc0[p])) && (N[p] == 0 || (xy1[p] && desc1[p])),
                      "sslam_lightglue_match_batch_dev: pair %d: NULL input", p);
        SSLAM_REQUIRE((M[p] < N[p] ? M[p] : N[p]) <= out_stride,
                      "sslam_lightglue_match_batch_dev: pair %d can emit %d matches, out_stride is %d", p,
                      M[p] < N[p] ? M[p] : N[p], out_stride);
        src.xy[2 * p] = xy0[p]; src.desc[2 * p] = desc0[p]; src.bound[2 * p] = M[p];
        src.cnt[2 * p] = m_dev ? m_dev[p] : nullptr;
        src.xy[2 * p + 1] = xy1[p]; src.desc[2 * p + 1] = desc1[p]; src.bound[2 * p + 1] = N[p];
        src.cnt[2 * p + 1] = n_dev ? n_dev[p] : nullptr;
    }
    return lg_enqueue_cached(g, n_pairs, src, min_conf, ij_out, score_out, info_out, out_stride);
}

int sslam_lightglue_match_dev(sslam_lightglue* g, const float* xy0, const float* desc0, int M,
                              const float* xy1, const float* desc1, int N, const int32_t* m_dev,
                              const int32_t* n_dev, float min_conf, int32_t* ij_out, float* score_out,
                              int32_t* info_out) {
    SSLAM_REQUIRE(g && ij_out && score_out && info_out, "sslam_lightglue_match_dev: NULL argument");
    SSLAM_REQUIRE(M >= 0 && N >= 0 && M <= g->Kc && N <= g->Kc,
                  "sslam_lightglue_match_dev: M=%d N=%d exceed max_kpts capacity %d", M, N, g->Kc);
    SSLAM_REQUIRE((M == 0 || (xy0 && desc0)) && (N == 0 || (xy1 && desc1)),
                  "sslam_lightglue_match_dev: NULL input");
    StageSrc src{};
    src.xy[0] = xy0; src.desc[0] = desc0; src.cnt[0] = m_dev; src.bound[0] = M;
    src.xy[1] = xy1; src.desc[1] = desc1; src.cnt[1] = n_dev; src.bound[1] = N;
    return lg_enqueue_cached(g, 1, src, min_conf, ij_out, score_out, info_out, g->Kc);
}

int sslam_lightglue_match_host(sslam_lightglue* g, const float* xy0, const float* desc0, int M,
                               const float* xy1, const float* desc1, int N, float min_conf,
                               int32_t* ij_out, float* score_out, int32_t* k_out,
                               int32_t* stop_layer_out) {
    return sslam_lightglue_match_host_sized(g, xy0, desc0, M, nullptr, xy1, desc1, N, nullptr, min_conf, ij_out, score_out,
                                            k_out, stop_layer_out);
}

int sslam_lightglue_match_host_sized(sslam_lightglue* g, const float* xy0, const float* desc0, int M, const float* size0,
                                     const float* xy1, const float* desc1, int N, const float* size1, float min_conf,
                                     int32_t* ij_out, float* score_out, int32_t* k_out, int32_t* stop_layer_out) {
    SSLAM_REQUIRE(g && ij_out && score_out && k_out, "sslam_lightglue_match_host: NULL argument");
    SSLAM_REQUIRE((!size0 || (size0[0] > 0 && size0[1] > 0)) && (!size1 || (size1[0] > 0 && size1[1] > 0)),
                  "sslam_lightglue_match_host_sized: image_size must be positive (W, H)");
    SSLAM_REQUIRE(M >= 0 && N >= 0 && M <= g->Kc && N <= g->Kc,
                  "sslam_lightglue_match_host: M=%d N=%d exceed max_kpts capacity %d", M, N, g->Kc);
    *k_out = 0;
    if (stop_layer_out) *stop_layer_out = 0;
    if (M == 0 || N == 0) return 0;            // features_utils.py:118-124 -> []
    SSLAM_REQUIRE(xy0 && desc0 && xy1 && desc1, "sslam_lightglue_match_host: NULL input");
    SSLAM_HIP_CHECK(hipSetDevice(g->ctx->device));
    hipStream_t s = g->ctx->stream;
    const size_t K = (size_t)g->Kc;
    SSLAM_HIP_CHECK(hipMemcpyAsync(g->up_xy, xy0, (size_t)M * 8, hipMemcpyHostToDevice, s));
    SSLAM_HIP_CHECK(hipMemcpyAsync(g->up_desc, desc0, (size_t)M * DIN * 4, hipMemcpyHostToDevice, s));
    SSLAM_HIP_CHECK(hipMemcpyAsync(g->up_xy + 2 * K, xy1, (size_t)N * 8, hipMemcpyHostToDevice, s));
    SSLAM_HIP_CHECK(hipMemcpyAsync(g->up_desc + K * DIN, desc1, (size_t)N * DIN * 4, hipMemcpyHostToDevice, s));
    StageSrc src{};
    src.xy[0] = g->up_xy; src.desc[0] = g->up_desc; src.bound[0] = M;
    src.xy[1] = g->up_xy + 2 * K; src.desc[1] = g->up_desc + K * DIN; src.bound[1] = N;
    if (size0) { src.size_w[0] = size0[0]; src.size_h[0] = size0[1]; }
    if (size1) { src.size_w[1] = size1[0]; src.size_h[1] = size1[1]; }
    g->last_pairs = 1;
    if (int rc = lg_enqueue(g, 1, src, min_conf, g->out_ij, g->out_score, g->out_info, (long)K)) return rc;
    int32_t info[4];
    SSLAM_HIP_CHECK(hipMemcpyAsync(info, g->out_info, sizeof(info), hipMemcpyDeviceToHost, s));
    SSLAM_HIP_CHECK(hipStreamSynchronize(s));
    if (info[0] == -1) {                       // lg_emit_kernel: this pair's range flag was raised
        int flag = 0;
        (void)lg_take_range_flag(g, &flag);    // reported here: clear the instance's sticky word
        SSLAM_REQUIRE(false, "sslam_lightglue_match_host: an activation left the fp16 range of the split-precision "
                             "path (|value| >= 65520); rescale the descriptors or use sslam_lightglue_set_precision(lg, 0)");
    }
    const int Kn = info[0];
    SSLAM_REQUIRE(Kn >= 0 && Kn <= (M < N ? M : N), "sslam_lightglue_match_host: corrupt match count %d", Kn);
    if (Kn) {
        SSLAM_HIP_CHECK(hipMemcpyAsync(ij_out, g->out_ij, (size_t)Kn * 8, hipMemcpyDeviceToHost, s));
        SSLAM_HIP_CHECK(hipMemcpyAsync(score_out, g->out_score, (size_t)Kn * 4, hipMemcpyDeviceToHost, s));
        SSLAM_HIP_CHECK(hipStreamSynchronize(s));
    }
    *k_out = Kn;
    if (stop_layer_out) *stop_layer_out = info[1];
    return 0;
}

/* Test hook: copy an internal buffer to the host after a match call (pair 0 of the last batch
 * unless `which` has 0x100 * pair added).
 * which: 0 = x (token states) [2][Kc][256], 1 = sim [Kc][Kc], 2 = ind [2][Kc] (int32),
 *        3 = prune counters [2][Kc] (int32), 4 = info {K, stop, n0, n1} (single-pair host call),
 *        5 = enc_cos [2][Kc][32], 6 = FFN hidden fp32 [2][Kc][512], 7 / 8 / 9 = hi planes (fp16) of the
 *        attention context / q / k */
int sslam_lightglue_debug_read(sslam_lightglue* g, int which, void* dst, size_t bytes) {
    SSLAM_REQUIRE(g && dst, "sslam_lightglue_debug_read: NULL argument");
    const size_t K = (size_t)g->Kc;
    const int pair = which >> 8;
    which &= 0xff;
    SSLAM_REQUIRE(pair >= 0 && pair < g->NB, "sslam_lightglue_debug_read: pair %d out of range", pair);
    const char* src = nullptr; size_t cap = 0;
    switch (which) {
        case 0: cap = 2 * K * D * 4; src = (const char*)g->x + pair * cap; break;
        case 1: cap = K * K * 4; src = (const char*)g->sim + pair * cap; break;
        case 2: cap = 2 * K * 4; src = (const char*)g->ind + pair * cap; break;
        case 3: cap = 2 * K * 4; src = (const char*)g->prune + pair * cap; break;
        case 4: cap = 16; src = (const char*)g->out_info; break;
        case 5: cap = 2 * K * ENC * 4; src = (const char*)g->enc_cos + pair * cap; break;
        case 6: cap = 2 * K * 2 * D * 4; src = (const char*)g->hid + pair * cap; break;          // FFN hidden (pre-LN), fp32
        case 7: cap = 2 * K * D * 2; src = (const char*)g->msgs_hi; break;                       // context planes (k-panel layout), pair 0
        case 8: cap = 2 * K * D * 2; src = (const char*)g->qs_hi + pair * cap; break;            // q hi plane [2][4][Kc][64]
        case 9: cap = 2 * K * D * 2; src = (const char*)g->ks_hi + pair * cap; break;
        default: SSLAM_REQUIRE(false, "sslam_lightglue_debug_read: unknown buffer %d", which);
    }
    SSLAM_REQUIRE(bytes <= cap, "sslam_lightglue_debug_read: %zu bytes requested, buffer has %zu", bytes, cap);
    SSLAM_HIP_CHECK(hipStreamSynchronize(g->ctx->stream));
    SSLAM_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return 0;
}

/* Bracket every attention launch (the dominant kernel) with HIP events on the context stream. */
int sslam_lightglue_profile(sslam_lightglue* g, int enable) {
    SSLAM_REQUIRE(g != nullptr, "sslam_lightglue_profile: NULL instance");
    g->profile = enable != 0;
    return 0;
}

/* Synchronise and return the summed duration / number of the bracketed attention launches
 * since the last read. */
int sslam_lightglue_profile_read(sslam_lightglue* g, float* total_ms_out, int32_t* launches_out) {
    SSLAM_REQUIRE(g && total_ms_out && launches_out, "sslam_lightglue_profile_read: NULL argument");
    SSLAM_HIP_CHECK(hipStreamSynchronize(g->ctx->stream));
    float tot = 0.0f;
    for (size_t i = 0; i + 1 < g->ev_used; i += 2) {
        float ms = 0.0f;
        SSLAM_HIP_CHECK(hipEventElapsedTime(&ms, g->ev[i], g->ev[i + 1]));
        tot += ms;
    }
    *total_ms_out = tot;
    *launches_out = (int32_t)(g->ev_used / 2);
    g->ev_used = 0;
    return 0;
}

/* 0: every contraction on the exact-fp32 matrix-core instruction; 1 (default): transformer layers on
 * the fp16 hi/lo split path (3 MFMA per product, ~2^-22 relative error), assignment stays fp32; 2 (opt-in, r04): as 1, but
 * attention carries P as ONE fp16 plane in P.V (2 MFMA per product there, row sums over the rounded weights): -12 % attention
 * time, match indices identical on every parity case, token states 2.4e-5 instead of 4e-6 from exact (profiles/r04_split_study.md). */
int sslam_lightglue_set_precision(sslam_lightglue* g, int mode) {
    SSLAM_REQUIRE(g != nullptr && (mode == 0 || mode == 1 || mode == 2), "sslam_lightglue_set_precision: bad argument");
    g->settings_changed();
    g->precision = mode == 0 ? 0 : 1;
    g->p_single = mode == 2;
    return 0;
}

/* Test hook: execute only the first `layers` transformer layers (and optionally only the self block
 * of the last one) so intermediate token states can be compared with the oracle. */
int sslam_lightglue_debug_layers(sslam_lightglue* g, int layers, int self_only) {
    SSLAM_REQUIRE(g != nullptr && layers >= 1 && layers <= NL, "sslam_lightglue_debug_layers: bad argument");
    g->settings_changed();
    g->dbg_layers = layers; g->dbg_self_only = self_only != 0;
    return 0;
}

/* Test hook: force the key split of the attention launches (0 = chosen by batch size). */
int sslam_lightglue_debug_key_split(sslam_lightglue* g, int ks) {
    SSLAM_REQUIRE(g != nullptr && ((ks >= -5 && ks <= 2 && ks != -2) || ks == 4 || ks == 101 || ks == 102 || ks == 104),
                  "sslam_lightglue_debug_key_split: ks must be -5 (split by size, assembly kernel, the merge as a launch of its own), "
                  "-4 (split by size, r02 4-wave kernel), -3 (no split, assembly kernel), "
                  "-1 (no split, r02 4-wave kernel), 0, 1, 2 or 4 (forced split, r02 4-wave kernel), 101, 102 or 104 (forced split, "
                  "assembly kernel)");
    g->settings_changed();
    g->force_ks = ks;
    return 0;
}

/* Test hook: -1 = linears chosen by batch size, 0 = always the 64-row ring kernels (single-pair form), 1 = always
 * the batched form (128 x 128 projections + the fused FFN kernel). */
int sslam_lightglue_debug_big_gemm(sslam_lightglue* g, int mode) {
    SSLAM_REQUIRE(g != nullptr && mode >= -1 && mode <= 5 && mode != 4, "sslam_lightglue_debug_big_gemm: bad argument");
    g->settings_changed();
    g->big_gemm = mode;
    return 0;
}

/* Precision study hook (profiles/r04_split_study.md; never set by the product): drop cross terms of the three-term split
 * products.  mask bits - attention logits: 0x01 K as one fp16 plane (S = kh.qh + kh.ql), 0x02 Q as one plane; context:
 * 0x04 P as one plane (row sum over the rounded weights; runs on the 4-wave kernel), 0x08 V as one plane; W.x products:
 * activation low plane dropped in the projections (0x10) / the FFN (0x20), weight low plane dropped in the projections
 * (0x40) / the FFN (0x80).  A dropped term is computed against an all-zero plane (same bits as not issuing the MFMA).
 * With the fused FFN kernel (batched form) the hidden activations stay three-term; debug_big_gemm(lg, 0) covers them. */
int sslam_lightglue_debug_split_form(sslam_lightglue* g, int mask) {
    SSLAM_REQUIRE(g != nullptr && mask >= 0 && mask <= 0xff, "sslam_lightglue_debug_split_form: bad argument");
    g->settings_changed();
    if (mask && !g->zero_plane) {
        const size_t act = (size_t)g->NIc * g->Kc * 2 * D, wmax = (size_t)4 * D * D;
        const size_t n = (act > wmax ? act : wmax) * sizeof(_Float16);
        SSLAM_HIP_CHECK(hipSetDevice(g->ctx->device));
        SSLAM_HIP_CHECK(hipMalloc((void**)&g->zero_plane, n));
        SSLAM_HIP_CHECK(hipMemset(g->zero_plane, 0, n));
    }
    g->study = mask;
    return 0;
}

/* Replay the launch sequence of sslam_lightglue_match_dev / _batch_dev as a cached hipGraph (one
 * graph per distinct argument tuple, LRU of 128; ~190 launches become one hipGraphLaunch): for
 * callers that cycle through a fixed set of buffers, as the frame pipeline does.  Results are
 * identical.  Changing any instance setting drops the cache. */
int sslam_lightglue_use_graphs(sslam_lightglue* g, int enable) {
    SSLAM_REQUIRE(g != nullptr, "sslam_lightglue_use_graphs: NULL instance");
    if (!enable) g->settings_changed();
    g->use_graphs = enable != 0;
    return 0;
}

/* Synchronise and report (then clear) the range flag of the split-precision path: 1 if, since the
 * last call, a finite activation with |value| >= 65520 reached an fp16 split (its results are then
 * not fp32-grade).  The _host entry points check it themselves; _dev / batch callers poll here. */
int sslam_lightglue_range_overflow(sslam_lightglue* g, int* flag_out) {
    SSLAM_REQUIRE(g && flag_out, "sslam_lightglue_range_overflow: NULL argument");
    return lg_take_range_flag(g, flag_out);
}

int sslam_lightglue_capacity(sslam_lightglue* g, int* kc_out) {
    SSLAM_REQUIRE(g && kc_out, "sslam_lightglue_capacity: NULL argument");
    *kc_out = g->Kc;
    return 0;
}

int sslam_lightglue_batch_capacity(sslam_lightglue* g, int* pairs_out) {
    SSLAM_REQUIRE(g && pairs_out, "sslam_lightglue_batch_capacity: NULL argument");
    *pairs_out = g->NB;
    return 0;
}

}  // extern "C"
