// aliked_kernels.hip - ALIKED-n16 keypoint + descriptor extraction on gfx950.
//
// Replaces `detector.extract(_bgr_to_tensor(img))` + rbd + the descriptor
// re-normalisation at slam/core/features_utils.py:92-100 (the cvg/LightGlue
// `ALIKED` module: image preprocessor, conv encoder with two deformable stages,
// feature aggregation, score head, DKD keypoint detection, SDDH descriptors).
//
// Numeric type fp32.  Roofline of the dense part: HBM (small-channel 3x3 convs);
// the 128-channel, 162 MB normalised feature map of the reference is NEVER
// materialised: the score head consumes it on the fly, and the descriptor head
// rebuilds the 128-vector only at the <= 73 pixels per keypoint it reads.
//
// HBM layout (planar CHW, padded network size Hp x Wp = multiple of 32):
//   img   [3][Hp][Wp]   preprocessed RGB        x1 [16][Hp][Wp]
//   x2 [32][Hp/2][Wp/2]  x3 [64][Hp/8][Wp/8]    x4 [128][Hp/32][Wp/32]
//   g2,g3,g4 [32][..]    gated 1x1 projections of x2..x4 (x1's is recomputed)
//   s8 [8][Hp][Wp]       first score-head layer  rnorm [Hp][Wp] 1/||F||
//   score [h][w]         un-padded score map     nms [h][w]
#include "common.hpp"
#include "gemm_f32.hpp"
#include <type_traits>
#include "gemm_f16x3.hpp"

namespace {

using sslam::f32x16;
using sslam::acc_row;
using sslam::GemmSmem;
using sslam::GemmA;
using sslam::gemm_mainloop;

constexpr float SELU_SCALE = 1.0507009873554804934193349852946f;
constexpr float SELU_ALPHA = 1.6732632423543772848170429916717f;
// (r04: a ~18-instruction expm1 - Taylor near zero, exact-argument exp2 below - instead of ocml's expm1f changed no kernel's
//  time by more than 4 %: these kernels wait on memory, their vector pipe is not the limit; ocml's stays)
#ifndef AL_FAST_SELU
#define AL_FAST_SELU 1
#endif
// SELU.  r04: exp(x) - 1 on the hardware exponential (v_exp_f32, 1 ulp) instead of expm1f - 7 instead of ~30 vector instructions
// per value; the dense stages are bound by vector-instruction issue (the aggregation evaluates 40 SELUs per pixel, the
// descriptor GEMM epilogue 4 M per frame).  Absolute error <= 1.2e-7 x 1.76 (the subtraction near x = 0): fp32 rounding of an
// O(1) value.  AL_FAST_SELU=0 restores expm1f.
__device__ __forceinline__ float selu(float x) {
#if AL_FAST_SELU
    return SELU_SCALE * (x > 0.0f ? x : SELU_ALPHA * (__expf(x) - 1.0f));
#else
    return SELU_SCALE * (x > 0.0f ? x : SELU_ALPHA * expm1f(x));
#endif
}

// SELU on ocml's expm1f (a polynomial + v_ldexp: no transcendental instruction): al_aggregate_kernel keeps it.  History: with
// the hardware-exponential form in that kernel's tail, 1 / ||F|| came out wrong by 0.2 - 4 % on 16 consecutive pixels about once
// in 150 frames when other streams' kernels shared the GPU (r04: bisected over commits; r05: the exponentials are a switch, not
// the place - profiles/r05_aggregate_selu_hazard.md).  r06 found the place (profiles/r06_aggregate_rnorm_diagnosis.md, section 5):
// the cheap tail tipped the SLP vectoriser into pairing agg_level's bilinear coefficients as
//     v_pk_mul_f32 v[32:33], v[14:15], v[12:13] op_sel:[0,1] op_sel_hi:[1,0]        ; (hy * lx, hx * ly)
// and on gfx950 a packed fp32 instruction with op_sel = [0,1] and two different sources computes its LOW half with src1's high
// half read as 0.0 in lanes 48..63, about once in 10^4 executions, while a wave of another kernel executes a wide-operand MFMA
// on the same SIMD (scripts/ubench/pk_probe.hip reproduces it in a 60-line kernel).  The kernel now carries no packed fp32 at all
// (AL_AGG_TARGET below) and build.py refuses any library that holds the form (isa_guard.py); the polynomial SELU stays because
// the kernel is latency-bound (no time difference) and its outputs are the ones every golden hash was taken with.
#ifndef AL_AGG_FAST_SELU
#define AL_AGG_FAST_SELU 0
#endif
// (AL_AGG_FAST_SELU, experiments only - scripts/ab_stress_aliked.sh: bit 0 = the hardware-exponential form in the kernel's channel
//  loop, bit 1 = in its tail)
template <int SITE> __device__ __forceinline__ float selu_precise(float x) {
    if ((AL_AGG_FAST_SELU >> SITE) & 1) return selu(x);
    return SELU_SCALE * (x > 0.0f ? x : SELU_ALPHA * expm1f(x));
}
constexpr int HBINS = 4096;        // score histogram bins (uniform in score, monotone)

struct ALCtrl {
    int n_cand;       // candidates above threshold
    int n_kp;         // keypoints emitted
    int overflow;     // candidate buffer overflow flag
    int found;        // the detection threshold found candidates (else the second collect launch thresholds on mean(score map))
    int range_overflow;   // a finite activation with |value| >= 65520 reached a split (fp16 hi/lo) operand: the frame's features are void
    int pad[11];
};

// RANGE of the split-precision stages (gemm_f16x3.hpp): split2_fast / split8_fast do not saturate, they report the largest
// magnitude they saw.  Every kernel that splits keeps that maximum per thread over its whole run (`amax` below) and raises the
// frame's flag once at its end; al_finalize_kernel turns a raised flag into keypoint count -1 (and the instance's sticky word),
// so the verdict travels with the result exactly as in the matcher (LGCtrl::range_overflow).
__device__ __forceinline__ void al_range_note(float amax, ALCtrl* ctrl) {
    if (amax >= 65520.0f && amax < INFINITY) ctrl->range_overflow = 1;
}
// a WEIGHT into its (hi, lo) planes, once at instance creation (BN scales already folded where the kernel folds them): a finite
// |v| >= 65520 does not fit and raises the instance's flag - sslam_aliked_create refuses such weights, as the matcher does
__device__ __forceinline__ void al_split_weight(float v, _Float16& hi, _Float16& lo, int* range_flag) {
    if (fabsf(v) >= 65520.0f && fabsf(v) < INFINITY) *range_flag = 1;
    hi = fabsf(v) < 6.103515625e-5f ? (_Float16)0.0f : (_Float16)v;
    lo = (_Float16)((v - (float)hi) * sslam::SPLIT_SCALE);
}

struct Dims {
    int H, W, C;          // input image
    int h, w;             // resized
    int Hp, Wp;           // padded (multiple of 32)
    int pl, pt;           // left / top padding
};

// Batched extraction (sslam_aliked_extract_batch_dev): F frames of one size go through ONE launch sequence.  Every
// per-frame buffer of frame f sits `fs` bytes behind frame 0's (one workspace block per frame, carved identically),
// so a kernel takes frame 0's pointers plus that stride and finds its frame in a grid dimension; weights and the
// blur taps are shared.  The arithmetic of a frame does not depend on F: a batch gives the single-frame results bit
// for bit.
constexpr int MAX_FRAMES = 16;
template <typename T> __device__ __forceinline__ T* fsh(T* p, int f, size_t fs) {
    // byte arithmetic on the POINTER, and no null test: a round trip through an integer, or a phi with a null constant, loses
    // the global address space - every load behind it becomes a flat_load, which also counts against the LDS counter, so
    // LDS reads wait for global loads in flight (r04: 1 175 flat operations in this file before).  A null optional argument
    // becomes a non-null garbage pointer: such arguments are only touched under the template flag / `fsh0` below.
    using B = std::conditional_t<std::is_const_v<T>, const char, char>;
    return reinterpret_cast<T*>(reinterpret_cast<B*>(p) + (size_t)f * fs);
}
// XCD-aware placement of image tiles (r06).  Workgroups are dealt to the 8 XCDs round-robin by their linear id and each XCD has
// an L2 of its own, so with tiles numbered in raster order the eight neighbours of a tile sit behind seven OTHER L2s and every
// halo line is fetched from HBM once per XCD that touches it (PMC r04 - r06: al_score_tail fetched 30.8 MB for 10.5 MB of s8).
// xcd_band() renumbers: inside every frame the workgroups that share an XCD (ids congruent mod 8) get a CONTIGUOUS band of tile
// rows, so a halo is re-read from the L2 that already holds it (30.8 -> 11.9 MB; al_resize_pad 12.4 -> 5.9 MB; kernel times
// unchanged within 2 %: neither kernel is bound by that traffic - scripts/ab_aliked_band.sh; form 1, whole frames per XCD at
// F = 8, measured 2 % slower per call).  Placement only: every tile is computed by exactly one workgroup with the same arithmetic.
// -> (x, y, z) the kernel uses in place of blockIdx.
#ifndef AL_XCD_BAND
#define AL_XCD_BAND 2
#endif
struct Tile3 { int x, y, z; };
__device__ __forceinline__ Tile3 xcd_band() {
#if AL_XCD_BAND == 1
    const unsigned nx = gridDim.x, ny = gridDim.y, total = nx * ny * gridDim.z;
    if ((total & 7u) == 0u) {
        const unsigned L = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
        const unsigned g = (L & 7u) * (total >> 3) + (L >> 3);
        return Tile3{(int)(g % nx), (int)((g / nx) % ny), (int)(g / (nx * ny))};
    }
#elif AL_XCD_BAND == 2          // bands inside every frame (the XCDs work on the same frame at the same time)
    const unsigned nx = gridDim.x, ny = gridDim.y, per = nx * ny;
    if ((per & 7u) == 0u) {
        const unsigned L = blockIdx.x + nx * blockIdx.y;
        const unsigned g = (L & 7u) * (per >> 3) + (L >> 3);
        return Tile3{(int)(g % nx), (int)(g / nx), (int)blockIdx.z};
    }
#endif
    return Tile3{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
}
template <typename T> __device__ __forceinline__ T* fsh0(T* p, int f, size_t fs) {      // null stays null (run-time optional outputs)
    return p ? fsh(p, f, fs) : nullptr;
}
// element at a wave-uniform base + a 32-bit BYTE offset of the lane: the compiler can then use the scalar-base addressing mode
// (an `unsigned` element index is widened to 64 bits before the shift and costs a register pair per distinct offset)
template <typename T> __device__ __forceinline__ T& at_b(T* base, unsigned byte_off) {
    using B = std::conditional_t<std::is_const_v<T>, const char, char>;
    return *reinterpret_cast<T*>(reinterpret_cast<B*>(base) + byte_off);
}
struct FrameIn { const uint8_t* img[MAX_FRAMES]; };
struct FrameOut { float* xy[MAX_FRAMES]; float* desc[MAX_FRAMES]; float* score[MAX_FRAMES]; int32_t* n[MAX_FRAMES]; };

// ------------------------------------------------------------------------ //
//  0. pre-processing: u8 HWC (BGR | gray | BGRA) -> RGB/255 -> [blur] -> resize -> pad
// ------------------------------------------------------------------------ //
__global__ void al_to_float_kernel(FrameIn srcs, float* __restrict__ dst, Dims d,
                                   const float* __restrict__ gx, int kx, int blur, size_t fs) {
    // dst [3][H][W] = horizontally blurred RGB/255 (reflect border), or plain RGB/255
    const uint8_t* __restrict__ src = srcs.img[blockIdx.z];
    dst = fsh(dst, blockIdx.z, fs);
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= d.W) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int sc = d.C == 1 ? 0 : 2 - c;                      // BGR -> RGB
        float v;
        if (!blur) {
            v = (float)src[((size_t)y * d.W + x) * d.C + sc] / 255.0f;
        } else {
            v = 0.0f;
            const int r = kx / 2;
#pragma unroll 8            // (rolled, every tap waited for its own byte load)
            for (int k = 0; k < kx; ++k) {
                int xx = x + k - r;
                xx = xx < 0 ? -xx : (xx >= d.W ? 2 * d.W - 2 - xx : xx);       // reflect
                v += gx[k] * ((float)src[((size_t)y * d.W + xx) * d.C + sc] / 255.0f);
            }
        }
        dst[((size_t)c * d.H + y) * d.W + x] = v;
    }
}

__global__ void al_resize_pad_kernel(const float* __restrict__ src, float* __restrict__ img, Dims d,
                                     const float* __restrict__ gy, int ky, int blur, size_t fs) {
    // img [3][Hp][Wp]: replicate-padded bilinear (align_corners=False) resize of (vertically blurred) src
    const Tile3 tb = xcd_band();                    // (the vertical taps and the bilinear rows of neighbouring output rows overlap)
    src = fsh(src, tb.z, fs); img = fsh(img, tb.z, fs);
    const int xp = tb.x * blockDim.x + threadIdx.x, yp = tb.y;
    if (xp >= d.Wp) return;
    const int y = min(max(yp - d.pt, 0), d.h - 1), x = min(max(xp - d.pl, 0), d.w - 1);
    const float sy_ = (float)d.H / (float)d.h, sx_ = (float)d.W / (float)d.w;
    float fy = sy_ * ((float)y + 0.5f) - 0.5f, fx = sx_ * ((float)x + 0.5f) - 0.5f;
    fy = fy < 0.0f ? 0.0f : fy;
    fx = fx < 0.0f ? 0.0f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < d.H - 1), x1 = x0 + (x0 < d.W - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* p = src + (size_t)c * d.H * d.W;
        float v00, v01, v10, v11;
        if (!blur) {
            v00 = p[(size_t)y0 * d.W + x0]; v01 = p[(size_t)y0 * d.W + x1];
            v10 = p[(size_t)y1 * d.W + x0]; v11 = p[(size_t)y1 * d.W + x1];
        } else {
            v00 = v01 = v10 = v11 = 0.0f;
            const int r = ky / 2;
#pragma unroll 4            // (16 loads in flight per trip group; rolled, each tap row waited for its own four)
            for (int k = 0; k < ky; ++k) {
                int ya = y0 + k - r, yb = y1 + k - r;
                ya = ya < 0 ? -ya : (ya >= d.H ? 2 * d.H - 2 - ya : ya);
                yb = yb < 0 ? -yb : (yb >= d.H ? 2 * d.H - 2 - yb : yb);
                const float g = gy[k];
                v00 += g * p[(size_t)ya * d.W + x0]; v01 += g * p[(size_t)ya * d.W + x1];
                v10 += g * p[(size_t)yb * d.W + x0]; v11 += g * p[(size_t)yb * d.W + x1];
            }
        }
        img[((size_t)c * d.Hp + yp) * d.Wp + xp] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------ //
//  1. the dense stages (block1, block2) - fused rolling-row kernels on the matrix cores (r04).
//      History in one paragraph (the superseded kernels live in scripts/ubench/aliked_superseded_r04.hpp):
//      r03 ran four implicit-GEMM convolutions on the exact-fp32 instruction, one tile per workgroup; phase ablations showed
//      load / MFMA / store never overlapping, the ISA showed why (flat pointers, serial loads in the epilogues, vector-issue
//      bound SELU and address arithmetic).  What remains: split-precision weight fragments prepared at create time
//      (al_conv16_wfrag / al_conv32p_wfrag / al_conv32_wfrag), al_block1_rows_kernel and al_block2_rows_kernel.
//      A fragment copy is [k-step][plane (hi, lo)][lane][8 halves]: the 16 bytes a lane feeds one MFMA with.
// ------------------------------------------------------------------------ //
// block1.conv2 (16 -> 16): k = (tap, channel) in 5 steps of 32 = two taps x 16 channels (v_mfma_f32_16x16x32_f16)
__global__ void al_conv16_wfrag_kernel(const float* __restrict__ w /*[ci 16][tap 9][co 16]*/, _Float16* __restrict__ wf, int* range_flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // (k-step, lane, e)
    if (i >= 5 * 64 * 8) return;
    const int e = i & 7, lane = (i >> 3) & 63, ks = i >> 9;
    const int kk = lane >> 4, co = lane & 15, tap = 2 * ks + (kk >> 1), ci = 8 * (kk & 1) + e;
    const float v = tap < 9 ? w[(ci * 9 + tap) * 16 + co] : 0.0f;
    _Float16 hi, lo;
    al_split_weight(v, hi, lo, range_flag);
    wf[((ks * 2 + 0) * 64 + lane) * 8 + e] = hi;
    wf[((ks * 2 + 1) * 64 + lane) * 8 + e] = lo;
}


// block2.conv1 + the 1 x 1 downsample branch (16 -> 32): k-step = tap (16 channels), the tenth k-step = the 1 x 1 weights
__global__ void al_conv32p_wfrag_kernel(const float* __restrict__ w /*[ci 16][tap 9][co 32]*/, const float* __restrict__ wd /*[ci 16][co 32]*/,
                                        _Float16* __restrict__ wf, int* range_flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // (k-step, lane, e)
    if (i >= 10 * 64 * 8) return;
    const int e = i & 7, lane = (i >> 3) & 63, ks = i >> 9;
    const int co = lane & 31, ci = 8 * (lane >> 5) + e;
    const float v = ks < 9 ? w[(ci * 9 + ks) * 32 + co] : wd[ci * 32 + co];
    _Float16 hi, lo;
    al_split_weight(v, hi, lo, range_flag);
    wf[((ks * 2 + 0) * 64 + lane) * 8 + e] = hi;
    wf[((ks * 2 + 1) * 64 + lane) * 8 + e] = lo;
}


// block2.conv2 (32 -> 32): k = tap x 32 + channel in 18 steps of 16
__global__ void al_conv32_wfrag_kernel(const float* __restrict__ w /*[ci 32][tap 9][co 32]*/, _Float16* __restrict__ wf, int* range_flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // one element (co, k)
    if (i >= 32 * 288) return;
    const int co = i / 288, k = i % 288, tap = k / 32, ci = k % 32;
    const float v = w[(ci * 9 + tap) * 32 + co];
    _Float16 hi, lo;
    al_split_weight(v, hi, lo, range_flag);
    const int ks = k >> 4, hh = (k >> 3) & 1, e = k & 7;
    wf[(((ks * 2 + 0) * 2 + hh) * 32 + co) * 8 + e] = hi;
    wf[(((ks * 2 + 1) * 2 + hh) * 32 + co) * 8 + e] = lo;
}


// ------------------------------------------------------------------------ //
//  1c. block2 FUSED (r04): pooling + conv1 (16 -> 32) + the 1 x 1 branch + conv2 (32 -> 32) + residual, rolling rows.
//      As two kernels the block moved 934 bytes per half-resolution pixel through HBM (t2 and the identity branch written,
//      then re-read with a halo); fused it reads x1 (256 B) and writes x2 (128 B).  A wave owns a strip of 30 output pixels x
//      `hs` rows and keeps two rings in LDS: three pooled input rows (34 pixels, 16 channels) and three rows of t2 (32
//      pixels = the strip + one on each side, 32 channels), both channel-last fp16 (hi, lo).  Step y: prefetch the input of
//      pooled row y + 3; conv1 -> t2 row y + 1 (zero outside the map: conv2's padding) into the t2 ring; the 1 x 1 branch on
//      pooled row y; conv2 row y from the t2 ring; + BN + residual + SELU -> x2 (planar fp32).  Split-precision matrix path
//      throughout (3 x (27 + 3 + 54) v_mfma_f32_32x32x16_f16 per step).  Four independent waves per workgroup share the A
//      fragments in LDS (conv1 lo planes 10 KB, conv2 36 KB; conv1 hi planes in registers): 147 KB, one workgroup per CU,
//      no workgroup barrier after the weights are staged.
// ------------------------------------------------------------------------ //
constexpr int B2_SW = 30;                                   // output pixels per strip
constexpr int B2_PXP = 24, B2_ROWP = 34 * B2_PXP, B2_PLP = 3 * B2_ROWP;          // pooled ring (halves)
constexpr int B2_PXT = 40, B2_ROWT = 32 * B2_PXT, B2_PLT = 3 * B2_ROWT;          // t2 ring (halves)
constexpr int B2_RING = 2 * B2_PLP + 2 * B2_PLT;                                   // halves per wave
#ifndef AL_B2_WREG
#define AL_B2_WREG 1       // 1: every A fragment in registers (224 of the wave's 512; one wave per SIMD either way) - the LDS then only serves B fragments
#endif
constexpr int B2_W1 = AL_B2_WREG ? 0 : 10 * 64 * 8, B2_W2 = AL_B2_WREG ? 0 : 36 * 64 * 8;      // shared A fragments (halves)
constexpr size_t B2_LDS = (size_t)(B2_W1 + B2_W2 + 4 * B2_RING) * 2 + 5 * 32 * 4;

#ifndef AL_B2_ASM
#define AL_B2_ASM 1
#endif
// The 224 registers of A fragments live in the accumulation-register file: written once, read by the MFMAs directly.  Through the
// builtin the compiler copies each fragment to an arch register first (257 v_accvgpr_read per step) and zeroes every chain's
// accumulator with 16 moves; here the first MFMA of a chain takes the literal 0 as C.  An asm MFMA is opaque to the hazard
// recogniser: nothing writes the A registers inside the loop, B fragments arrive by ds_read (waited for by register use), and
// b2_mfma_done(c1, c2) pads the MFMA -> VALU read distance after the last MFMA of a chain (as an in/out of both accumulators).
#if AL_B2_ASM && AL_B2_WREG
#define B2_AREG "a"
#else
#define B2_AREG "v"
#endif
__device__ __forceinline__ void b2_mfma0(f32x16& c, const sslam::half8& a_, const sslam::half8& b_) {
#if AL_B2_ASM
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : B2_AREG(a_), "v"(b_));
#else
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
    c = sslam::mfma16(a_, b_, z);
#endif
}
__device__ __forceinline__ void b2_mfma(f32x16& c, const sslam::half8& a_, const sslam::half8& b_) {
#if AL_B2_ASM
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : B2_AREG(a_), "v"(b_));
#else
    c = sslam::mfma16(a_, b_, c);
#endif
}
__device__ __forceinline__ void b2_mfma_done(f32x16& c1, f32x16& c2) {
#if AL_B2_ASM
    // the 19 wait states between the last MFMA of a chain and the first vector read of its accumulators, TIED to the
    // accumulators: every later read of c1 / c2 depends on this statement's outputs, so the compiler cannot schedule one
    // between the (opaque) MFMA statements and the padding
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(c1), "+v"(c2));
#endif
}
__global__ __launch_bounds__(256, 1) void al_block2_rows_kernel(const float* __restrict__ in /* x1 [16][2 H][2 W] */, float* __restrict__ out /* x2 [32][H][W] */,
                                                               int H, int W, int hs, int nblk, int strips, int n_waves,
                                                               const _Float16* __restrict__ wf1 /*[10][2][64][8]*/, const _Float16* __restrict__ wf2 /*[18][2][64][8]*/,
                                                               const float* __restrict__ a1, const float* __restrict__ b1, const float* __restrict__ bd,
                                                               const float* __restrict__ a2, const float* __restrict__ b2, ALCtrl* ctrl, size_t fs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char b2_lds[];
    _Float16* w1lo = reinterpret_cast<_Float16*>(b2_lds);
    _Float16* w2 = w1lo + B2_W1;
    float* aff = reinterpret_cast<float*>(w2 + B2_W2 + 4 * B2_RING);
    // (the wave index as a SCALAR: strip, block, frame and every row pointer derived from it stay in scalar registers)
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), h = lane >> 5, px = lane & 31;
    // stage the shared A fragments and the affine tables
    for (int i = t; i < B2_W1 / 8; i += 256)
        *reinterpret_cast<uint4*>(w1lo + i * 8) = *reinterpret_cast<const uint4*>(wf1 + (((i >> 6) * 2 + 1) * 64 + (i & 63)) * 8);
    for (int i = t; i < B2_W2 / 8; i += 256) *reinterpret_cast<uint4*>(w2 + i * 8) = *reinterpret_cast<const uint4*>(wf2 + (size_t)i * 8);
    (void)w1lo;
    if (t < 32) { aff[t] = a1[t]; aff[32 + t] = b1[t]; aff[64 + t] = bd[t]; aff[96 + t] = a2[t]; aff[128 + t] = b2[t]; }
    __syncthreads();
    const int gw = blockIdx.x * 4 + wave;
    if (gw >= n_waves) return;                               // (no barrier below)
    const int strip = gw % strips, blk = (gw / strips) % nblk, f = gw / (strips * nblk);
    in = fsh(in, f, fs); out = fsh(out, f, fs); ctrl = fsh(ctrl, f, fs);
    float amax = 0.0f;                                       // largest magnitude this thread split (al_range_note at the end)
    _Float16* P = w2 + B2_W2 + wave * B2_RING;               // [plane][slot][34][24]
    _Float16* T = P + 2 * B2_PLP;                            // [plane][slot][32][40]
    const int x0 = strip * B2_SW, yb = blk * hs, ye = min(yb + hs, H);
    const int inW = 2 * W, inH = 2 * H;
    const unsigned inHW = (unsigned)inH * inW;
    sslam::half8 ah1[10];
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) ah1[ks] = *reinterpret_cast<const sslam::half8*>(wf1 + ((ks * 2 + 0) * 64 + lane) * 8);
#if AL_B2_WREG
    sslam::half8 al1[10], wr2[36];
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) al1[ks] = *reinterpret_cast<const sslam::half8*>(wf1 + ((ks * 2 + 1) * 64 + lane) * 8);
#pragma unroll
    for (int i = 0; i < 36; ++i) wr2[i] = *reinterpret_cast<const sslam::half8*>(wf2 + (i * 64 + lane) * 8);
#define B2_A1LO(ks) al1[ks]
#define B2_A2(i) wr2[i]
#else
    const _Float16* w1l = w1lo + lane * 8;
    const _Float16* w2l = w2 + lane * 8;
#define B2_A1LO(ks) (*reinterpret_cast<const sslam::half8*>(w1l + (ks) * 512))
#define B2_A2(i) (*reinterpret_cast<const sslam::half8*>(w2l + (i) * 512))
#endif
    // a pooled row = 16 channels x 17 float4 pairs (34 pooled pixels from x0 - 2): 272 items, five rounds
    unsigned iofs[5]; bool iok[5]; int ipo[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int idx = min(lane + 64 * j, 271), ch = idx / 17, v4 = idx % 17, x = x0 - 2 + 2 * v4;
        iok[j] = lane + 64 * j < 272 && x >= 0 && x < W;
        iofs[j] = 4u * ((unsigned)ch * inHW + 2u * (unsigned)min(max(x, 0), W - 2));
        ipo[j] = 2 * v4 * B2_PXP + ch;
    }
    float4 ra[5], rb[5];
    auto load_row = [&](int yy) {
        const int yc = min(max(yy, 0), H - 1);
        const float* r0 = in + (size_t)(2 * yc) * inW;
        const float* r1 = r0 + inW;
#pragma unroll
        for (int j = 0; j < 5; ++j) { ra[j] = at_b(reinterpret_cast<const float4*>(r0), iofs[j]); rb[j] = at_b(reinterpret_cast<const float4*>(r1), iofs[j]); }
    };
    auto stash_row = [&](int slot, int yy) {
        const bool rowok = yy >= 0 && yy < H;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            if (lane + 64 * j >= 272) continue;
            const bool ok = rowok && iok[j];
            // 2 x 2 average summed in the order (0,0) (0,1) (1,0) (1,1)
            const float p0 = ok ? (((ra[j].x + ra[j].y) + rb[j].x) + rb[j].y) / 4.0f : 0.0f;
            const float p1 = ok ? (((ra[j].z + ra[j].w) + rb[j].z) + rb[j].w) / 4.0f : 0.0f;
            unsigned h2, l2;
            sslam::split2_fast(p0, p1, h2, l2, amax);
            const int o = slot * B2_ROWP + ipo[j];
            P[o] = __builtin_bit_cast(_Float16, (unsigned short)(h2 & 0xffffu));
            P[o + B2_PXP] = __builtin_bit_cast(_Float16, (unsigned short)(h2 >> 16));
            P[o + B2_PLP] = __builtin_bit_cast(_Float16, (unsigned short)(l2 & 0xffffu));
            P[o + B2_PLP + B2_PXP] = __builtin_bit_cast(_Float16, (unsigned short)(l2 >> 16));
        }
    };
    // pooled rows yb - 2, yb - 1, yb -> slots 0, 1, 2; the first two steps only build t2 rows yb - 1 and yb
#pragma unroll
    for (int r = 0; r < 3; ++r) { load_row(yb - 2 + r); stash_row(r, yb - 2 + r); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    const _Float16* pb = P + px * B2_PXP + 8 * h;            // conv1 B fragments: pooled pixel q + dx
    const _Float16* tb = T + px * B2_PXT + 8 * h;            // conv2 B fragments: t2 pixel j + dx
    const bool tq_ok = x0 - 1 + px >= 0 && x0 - 1 + px < W;  // t2 pixel of this lane inside the map
    const bool live = px < B2_SW && x0 + px < W;
    const unsigned HWb = 4u * (unsigned)H * W, lo = (unsigned)(4 * h) * HWb + 4u * px;
    auto step = [&](auto ph, int y) {
        // pooled rows y, y + 1, y + 2 in slots PH, PH + 1, PH + 2 (mod 3); t2 rows y - 1, y in slots PH, PH + 1; t2 row y + 1 -> slot PH + 2
        constexpr int PH = decltype(ph)::value;
        load_row(y + 3);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 c1, c2;
        {   // conv1 -> t2 row y + 1
#pragma unroll
            for (int ks = 0; ks < 9; ++ks) {
                const int o = ((PH + ks / 3) % 3) * B2_ROWP + (ks % 3) * B2_PXP;
                const sslam::half8 xh = *reinterpret_cast<const sslam::half8*>(pb + o);
                const sslam::half8 xl = *reinterpret_cast<const sslam::half8*>(pb + o + B2_PLP);
                if (ks == 0) { b2_mfma0(c1, ah1[ks], xh); b2_mfma0(c2, ah1[ks], xl); }
                else { b2_mfma(c1, ah1[ks], xh); b2_mfma(c2, ah1[ks], xl); }
                b2_mfma(c2, B2_A1LO(ks), xh);
            }
            b2_mfma_done(c1, c2);
            const bool ok = tq_ok && y + 1 >= 0 && y + 1 < H;
            _Float16* trow = T + ((PH + 2) % 3) * B2_ROWT + px * B2_PXT + 4 * h;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float vv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g4 + e, co = acc_row(r, lane);
                    vv[e] = ok ? selu(fmaf(c1[r] + c2[r] * sslam::SPLIT_INV, aff[co], aff[32 + co])) : 0.0f;
                }
                unsigned h01, l01, h23, l23;
                sslam::split2_fast(vv[0], vv[1], h01, l01, amax);
                sslam::split2_fast(vv[2], vv[3], h23, l23, amax);
                *reinterpret_cast<uint2*>(trow + 8 * g4) = make_uint2(h01, h23);
                *reinterpret_cast<uint2*>(trow + B2_PLT + 8 * g4) = make_uint2(l01, l23);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (y >= yb) {
            float idv[16];
            {   // the 1 x 1 branch: pooled row y (slot PH), pixel j + 2
                const int o = PH * B2_ROWP + 2 * B2_PXP;
                const sslam::half8 xh = *reinterpret_cast<const sslam::half8*>(pb + o);
                const sslam::half8 xl = *reinterpret_cast<const sslam::half8*>(pb + o + B2_PLP);
                b2_mfma0(c1, ah1[9], xh);
                b2_mfma0(c2, ah1[9], xl);
                b2_mfma(c2, B2_A1LO(9), xh);
                b2_mfma_done(c1, c2);
#pragma unroll
                for (int r = 0; r < 16; ++r) idv[r] = (c1[r] + c2[r] * sslam::SPLIT_INV) + aff[64 + acc_row(r, lane)];
            }
#pragma unroll
            for (int ks = 0; ks < 18; ++ks) {
                const int tap = ks >> 1, o = ((PH + tap / 3) % 3) * B2_ROWT + (tap % 3) * B2_PXT + 16 * (ks & 1);
                const sslam::half8 xh = *reinterpret_cast<const sslam::half8*>(tb + o);
                const sslam::half8 xl = *reinterpret_cast<const sslam::half8*>(tb + o + B2_PLT);
                if (ks == 0) { b2_mfma0(c1, B2_A2(0), xh); b2_mfma0(c2, B2_A2(0), xl); }
                else { b2_mfma(c1, B2_A2(ks * 2 + 0), xh); b2_mfma(c2, B2_A2(ks * 2 + 0), xl); }
                b2_mfma(c2, B2_A2(ks * 2 + 1), xh);
            }
            b2_mfma_done(c1, c2);
            if (live) {
                float* orow = out + (size_t)y * W + x0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = acc_row(r, lane);
                    at_b(orow, lo + (8 * (r / 4) + r % 4) * HWb) =
                        selu(fmaf(c1[r] + c2[r] * sslam::SPLIT_INV, aff[96 + co], aff[128 + co]) + idv[r]);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // this step's LDS reads before the overwrite of pooled slot PH
        stash_row(PH, y + 3);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    };
    for (int y = yb - 2; y < ye; y += 3) {
        step(std::integral_constant<int, 0>{}, y);
        if (y + 1 >= ye) break;
        step(std::integral_constant<int, 1>{}, y + 1);
        if (y + 2 >= ye) break;
        step(std::integral_constant<int, 2>{}, y + 2);
    }
    al_range_note(amax, ctrl);
}

// ------------------------------------------------------------------------ //
//  1d. block1 FUSED (r04): conv1 (3 -> 16, exact-fp32 MFMA) + conv2 (16 -> 16, split-precision pipe) in one rolling-rows kernel -
//      the intermediate map (64 bytes per pixel written and read back: two thirds of the block's HBM traffic) stays in an LDS
//      ring, as t2 does in al_block2_rows_kernel.  One wave per workgroup owns 30 output pixels x `hs` rows: a ring of three
//      image rows (planar fp32, 34 pixels from x0 - 2) and a ring of three rows of conv1's output (channel-last fp16 (hi, lo),
//      32 pixels from x0 - 1, zero outside the map = conv2's padding).  Step y: prefetch image row y + 3; conv1 -> ring row
//      y + 1; conv2 row y -> x1 (planar fp32).  Arithmetic per output as in al_conv16_rows_kernel<3, true> + al_conv16h_rows_kernel.
// ------------------------------------------------------------------------ //
constexpr int B1_SW = 30;
__global__ __launch_bounds__(64) void al_block1_rows_kernel(const float* __restrict__ in /* image [3][H][W] */, float* __restrict__ out /* x1 [16][H][W] */,
                                                           int H, int W, int hs, const float* __restrict__ w1 /*[3][9][16]*/,
                                                           const float* __restrict__ a1, const float* __restrict__ b1,
                                                           const _Float16* __restrict__ wf2 /*[5][2][64][8]*/,
                                                           const float* __restrict__ a2, const float* __restrict__ b2, ALCtrl* ctrl, size_t fs) {
    in = fsh(in, blockIdx.z, fs); out = fsh(out, blockIdx.z, fs); ctrl = fsh(ctrl, blockIdx.z, fs);
    float amax = 0.0f;                                                       // (al_range_note at the end)
    constexpr int RS = 48, SLOTI = 4 * RS;                                   // image ring: [slot][4 channels (one zero)][48 floats]
    constexpr int PXS = 24, ROWH = 34 * PXS, PLH = 3 * ROWH;                 // conv1-output ring (halves): [plane][slot][34 pixels][24]
    __shared__ __attribute__((aligned(16))) float iring[3 * SLOTI];
    __shared__ __attribute__((aligned(16))) _Float16 tring[2 * PLH];
    const int lane = threadIdx.x, kk = lane >> 4, n = lane & 15;
    const int x0 = blockIdx.x * B1_SW, yb = blockIdx.y * hs, ye = min(yb + hs, H);
    const size_t HW = (size_t)H * W;
    float aw[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) aw[tap] = kk < 3 ? w1[(kk * 9 + tap) * 16 + n] : 0.0f;
    sslam::half8 ah[5], al[5];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        ah[ks] = *reinterpret_cast<const sslam::half8*>(wf2 + ((ks * 2 + 0) * 64 + lane) * 8);
        al[ks] = *reinterpret_cast<const sslam::half8*>(wf2 + ((ks * 2 + 1) * 64 + lane) * 8);
    }
    float alr1[4], ber1[4], alr2[4], ber2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { alr1[i] = a1[4 * kk + i]; ber1[i] = b1[4 * kk + i]; alr2[i] = a2[4 * kk + i]; ber2[i] = b2[4 * kk + i]; }
    for (int i = lane; i < 3 * SLOTI; i += 64) iring[i] = 0.0f;               // (the fourth channel and the row tails stay zero)
    for (int i = lane; i < 2 * PLH / 2; i += 64) reinterpret_cast<unsigned*>(tring)[i] = 0u;      // (pixels 32, 33 are never written)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    // an image row: 3 channels x 17 pixel pairs (34 pixels from x0 - 2; x0 - 2 and W are even: a pair is inside or outside)
    const int ich = min(lane, 50) / 17, iv2 = min(lane, 50) % 17, ix = x0 - 2 + 2 * iv2;
    const bool iok = lane < 51 && ix >= 0 && ix < W;
    const unsigned iofs = 4u * ((unsigned)ich * (unsigned)HW + (unsigned)min(max(ix, 0), W - 2));
    float2 ri;
    auto load_row = [&](int yy) {
        const int yc = min(max(yy, 0), H - 1);
        ri = at_b(reinterpret_cast<const float2*>(in + (size_t)yc * W), iofs);
    };
    auto stash_row = [&](int slot, int yy) {
        if (lane < 51) {
            const bool ok = iok && yy >= 0 && yy < H;
            *reinterpret_cast<float2*>(&iring[slot * SLOTI + ich * RS + 2 * iv2]) = ok ? ri : make_float2(0.0f, 0.0f);
        }
    };
#pragma unroll
    for (int r = 0; r < 3; ++r) { load_row(yb - 2 + r); stash_row(r, yb - 2 + r); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    const float* ib = iring + kk * RS + n;                                      // conv1 B operand: channel kk, image-ring pixel q + dx
    int boff[5][3];                                                             // conv2 B fragments (as al_conv16h_rows_kernel)
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        const int tap = min(2 * ks + (kk >> 1), 8);
#pragma unroll
        for (int ph = 0; ph < 3; ++ph) boff[ks][ph] = ((ph + tap / 3) % 3) * ROWH + (n + tap % 3) * PXS + 8 * (kk & 1);
    }
    const unsigned HWb = 4u * (unsigned)HW, lo = (unsigned)(4 * kk) * HWb + 4u * n;
    auto step = [&](auto ph, int y) {
        // image rows y, y + 1, y + 2 in slots PH, PH + 1, PH + 2 (mod 3); conv1 rows y - 1, y in slots PH, PH + 1; conv1 row y + 1 -> slot PH + 2
        constexpr int PH = decltype(ph)::value;
        load_row(y + 3);
        __builtin_amdgcn_sched_barrier(0);
        {
            f32x4 acc[2];
            acc[0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; acc[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
                    acc[hf] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[tap], ib[((PH + tap / 3) % 3) * SLOTI + tap % 3 + 16 * hf], acc[hf], 0, 0, 0);
            const bool rowok = y + 1 >= 0 && y + 1 < H;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int q = 16 * hf + n, x = x0 - 1 + q;
                const bool ok = rowok && x >= 0 && x < W;
                float vv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) vv[i] = ok ? selu(fmaf(acc[hf][i], alr1[i], ber1[i])) : 0.0f;
                unsigned h01, l01, h23, l23;
                sslam::split2_fast(vv[0], vv[1], h01, l01, amax);
                sslam::split2_fast(vv[2], vv[3], h23, l23, amax);
                _Float16* tq = tring + ((PH + 2) % 3) * ROWH + q * PXS + 4 * kk;
                *reinterpret_cast<uint2*>(tq) = make_uint2(h01, h23);
                *reinterpret_cast<uint2*>(tq + PLH) = make_uint2(l01, l23);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (y >= yb) {
            f32x4 c1[2], c2[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) { c1[hf] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; c2[hf] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
#pragma unroll
            for (int ks = 0; ks < 5; ++ks)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const sslam::half8 xh = *reinterpret_cast<const sslam::half8*>(&tring[boff[ks][PH] + 16 * hf * PXS]);
                    const sslam::half8 xl = *reinterpret_cast<const sslam::half8*>(&tring[boff[ks][PH] + 16 * hf * PXS + PLH]);
                    c1[hf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], xh, c1[hf], 0, 0, 0);
                    c2[hf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], xl, c2[hf], 0, 0, 0);
                    c2[hf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ks], xh, c2[hf], 0, 0, 0);
                }
            float* orow = out + (size_t)y * W + x0;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int j = 16 * hf + n;
                if (j >= B1_SW || x0 + j >= W) continue;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    at_b(orow, lo + i * HWb + 64u * hf) = selu(fmaf(c1[hf][i] + c2[hf][i] * sslam::SPLIT_INV, alr2[i], ber2[i]));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        stash_row(PH, y + 3);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    };
    for (int y = yb - 2; y < ye; y += 3) {
        step(std::integral_constant<int, 0>{}, y);
        if (y + 1 >= ye) break;
        step(std::integral_constant<int, 1>{}, y + 1);
        if (y + 2 >= ye) break;
        step(std::integral_constant<int, 2>{}, y + 2);
    }
    al_range_note(amax, ctrl);
}

// ------------------------------------------------------------------------ //
//  2. small-map stages (1/8 and 1/32 resolution): pooling, offset conv,
//     deformable conv (torchvision deform_conv2d semantics) + BN + residual + SELU
// ------------------------------------------------------------------------ //
__global__ void al_avgpool_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int H, int W,
                                  int P, size_t fs, float* __restrict__ out_cl /* [pixel][C] copy for al_dcn_col (r04) */) {   // out [C][H/P][W/P]
    in = fsh(in, blockIdx.y, fs); out = fsh(out, blockIdx.y, fs); out_cl = fsh0(out_cl, blockIdx.y, fs);
    const int oh = H / P, ow = W / P;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * oh * ow) return;
    const int c = i / (oh * ow), y = (i / ow) % oh, x = i % ow;
    float s = 0.0f;
    if (P == 4) {                      // (both call sites: all 16 loads in flight; same summation order)
        float v[16];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) v[4 * a + b] = in[((size_t)c * H + y * 4 + a) * W + x * 4 + b];
#pragma unroll
        for (int k = 0; k < 16; ++k) s += v[k];
    } else {
        for (int a = 0; a < P; ++a)
            for (int b = 0; b < P; ++b) s += in[((size_t)c * H + y * P + a) * W + x * P + b];
    }
    out[i] = s / (float)(P * P);
    if (out_cl) out_cl[(size_t)(y * ow + x) * C + c] = s / (float)(P * P);
}

// offset conv: 3x3, zero pad, bias, clamp to +-max_off.  One wave per pixel: lanes stride over the
// CIN*9 (ci, tap) products, each lane keeps 18 partial sums, then 18 wave reductions.  The weights
// are read from the [18][CIN*9] copy made at create time, so each of the 18 loads of an iteration
// is 256 contiguous bytes across the wave (with the packed [k][18] layout every one of them
// walked the same 36 cache lines again: 648 line look-ups per iteration instead of 36).

// ------------------------------------------------------------------------ //
//  2a. the offset convolutions of the deformable blocks (3 x 3, CIN -> 18, + bias, clamp) on the split-precision matrix pipe
//      (r04).  The direct form above spreads k = (channel, tap) over the lanes and reduces 18 sums over 64 lanes per pixel
//      group: 40 us per launch for 53 MFLOP.  Here ONE WAVE computes one output row segment of 32 pixels x all 18 (of 32)
//      channels: it converts its 3 x 34 pixel neighbourhood of the channel-last fp32 input (the copies the im2col kernel reads)
//      to fp16 (hi, lo) planes in LDS once, then runs 9 x CIN / 16 k-steps of three v_mfma_f32_32x32x16_f16; A fragments
//      (weights) stream from a fragment-ordered split copy through L1.  Offsets move by ~1e-6 pixel (2^-22 relative).
// ------------------------------------------------------------------------ //
template <int CIN>
__global__ void al_offc_wfrag_kernel(const float* __restrict__ ow /*[ci*9 + tap][18]*/, _Float16* __restrict__ wf, int* range_flag) {
    constexpr int KS = 9 * CIN / 16;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // (k-step, lane, e)
    if (i >= KS * 64 * 8) return;
    const int e = i & 7, lane = (i >> 3) & 63, ks = i >> 9;
    const int co = lane & 31, tap = ks / (CIN / 16), ci = 16 * (ks % (CIN / 16)) + 8 * (lane >> 5) + e;
    const float v = co < 18 ? ow[(ci * 9 + tap) * 18 + co] : 0.0f;
    _Float16 hi, lo;
    al_split_weight(v, hi, lo, range_flag);
    wf[((ks * 2 + 0) * 64 + lane) * 8 + e] = hi;
    wf[((ks * 2 + 1) * 64 + lane) * 8 + e] = lo;
}

template <int CIN>
__global__ __launch_bounds__(256) void al_offset_conv_h_kernel(const float* __restrict__ in /* channel-last [H W][CIN] */, float* __restrict__ off /*[18][H W]*/,
                                                              int H, int W, const _Float16* __restrict__ wf, const float* __restrict__ b,
                                                              float max_off, ALCtrl* ctrl, size_t fs) {
    // four waves per output tile (32 pixels of one row x 18 channels): they fill the tile together and split the k-steps
    // (wave w takes k-steps w, w + 4, ...); the partial sums meet in LDS in wave order.  The maps are small (40 x 128 and
    // 10 x 32 pixels): with one wave per tile the 1/32 levels were 80 waves of 216 serial MFMAs.
    in = fsh(in, blockIdx.z, fs); off = fsh(off, blockIdx.z, fs); ctrl = fsh(ctrl, blockIdx.z, fs);
    float amax = 0.0f;                                        // (al_range_note below)
    constexpr int CP = CIN + 8, ROWH = 34 * CP, PLH = 3 * ROWH, C8 = CIN / 8, NFR = 3 * 34 * C8, KSC = CIN / 16, KS = 9 * KSC;
    constexpr int TILEH = 2 * PLH > 4 * 16 * 64 * 2 ? 2 * PLH : 4 * 16 * 64 * 2;      // (the partial sums reuse the tile: 4 x 16 x 64 floats)
    __shared__ __attribute__((aligned(16))) _Float16 tile[TILEH];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), h = lane >> 5, px = lane & 31;
    const int x0 = blockIdx.x * 32, y = blockIdx.y;
    // fill: fragments of 8 channels, (row, pixel, c8) with c8 fastest - 32 contiguous bytes per thread, whole pixels per thread group
    constexpr int NR = (NFR + 255) / 256, CH = NR < 4 ? NR : 4;      // rounds; CH of them with their loads in flight together
    for (int j0 = 0; j0 < NR; j0 += CH) {
        float4 va[CH], vb[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int idx = min(t + 256 * (j0 + j), NFR - 1), c8 = idx % C8, q = (idx / C8) % 34, row = idx / (C8 * 34);
            const int yy = min(max(y + row - 1, 0), H - 1), xx = min(max(x0 + q - 1, 0), W - 1);
            const float* p = in + ((size_t)yy * W + xx) * CIN + 8 * c8;
            va[j] = *reinterpret_cast<const float4*>(p); vb[j] = *reinterpret_cast<const float4*>(p + 4);
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int idx = t + 256 * (j0 + j);
            if (idx >= NFR) continue;
            const int c8 = idx % C8, q = (idx / C8) % 34, row = idx / (C8 * 34);
            const int yy = y + row - 1, xx = x0 + q - 1;
            const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
            const float v[8] = {va[j].x, va[j].y, va[j].z, va[j].w, vb[j].x, vb[j].y, vb[j].z, vb[j].w};
            uint4 hi, lo;
            sslam::split8_fast(v, hi, lo, amax);
            if (!ok) { hi = make_uint4(0u, 0u, 0u, 0u); lo = hi; }
            *reinterpret_cast<uint4*>(&tile[row * ROWH + q * CP + 8 * c8]) = hi;
            *reinterpret_cast<uint4*>(&tile[PLH + row * ROWH + q * CP + 8 * c8]) = lo;
        }
    }
    al_range_note(amax, ctrl);
    __syncthreads();
    f32x16 c1, c2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c1[r] = 0.0f; c2[r] = 0.0f; }
    const _Float16* bl = tile + px * CP + 8 * h;
    const _Float16* af = wf + lane * 8;
#pragma unroll
    for (int i = 0; i < (KS + 3) / 4; ++i) {
        const int ks = wave + 4 * i;
        if (ks >= KS) break;
        const int tap = ks / KSC, o = (tap / 3) * ROWH + (tap % 3) * CP + 16 * (ks % KSC);
        const sslam::half8 xh = *reinterpret_cast<const sslam::half8*>(bl + o);
        const sslam::half8 xl = *reinterpret_cast<const sslam::half8*>(bl + o + PLH);
        const sslam::half8 ah = *reinterpret_cast<const sslam::half8*>(af + (size_t)(ks * 2 + 0) * 512);
        const sslam::half8 al = *reinterpret_cast<const sslam::half8*>(af + (size_t)(ks * 2 + 1) * 512);
        c1 = sslam::mfma16(ah, xh, c1);
        c2 = sslam::mfma16(ah, xl, c2);
        c2 = sslam::mfma16(al, xh, c2);
    }
    __syncthreads();                                          // every wave is done reading the tile: it becomes the reduction buffer
    float* red = reinterpret_cast<float*>(tile);             // [wave][r][lane]
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = c1[r] + c2[r] * sslam::SPLIT_INV;
    __syncthreads();
    if (x0 + px < W) {
        const size_t HW = (size_t)H * W;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {                     // wave w finishes registers 4 w .. 4 w + 3
            const int r = 4 * wave + rr, co = acc_row(r, lane);
            const float v = ((red[(0 * 16 + r) * 64 + lane] + red[(1 * 16 + r) * 64 + lane]) + red[(2 * 16 + r) * 64 + lane]) + red[(3 * 16 + r) * 64 + lane];
            if (co < 18) off[(size_t)co * HW + (size_t)y * W + x0 + px] = fminf(fmaxf(v + b[co], -max_off), max_off);
        }
    }
}




// ------------------------------------------------------------------------ //
//  2b. deformable conv FUSED (r04): bilinear sampling + the K = 9 CIN (+ RC) contraction + BN + residual branch + SELU in ONE
//      kernel on the split-precision matrix pipe.  As im2col + split-K GEMM + epilogue the layer wrote and re-read its
//      im2col rows (11.8 MB per frame and layer at 1/8 resolution) and its split-K slabs (as much again) through three
//      launches of latency-bound small kernels (60 us per layer and batch of 8 frames).  Here a workgroup owns 32 pixels of a
//      row x all COUT channels; wave w takes the taps w, w + 4, w + 8 (slot 9 = the 1 x 1 residual branch on the block input):
//      per tap it samples its 32 x CIN values (torchvision deform_conv2d semantics, the arithmetic of al_dcn_col_kernel),
//      splits them into a wave-private LDS buffer and runs CIN / 16 k-steps x COUT / 32 tiles of three MFMAs; the four
//      partial sums meet in LDS in wave order.  The BN scale is folded into the main weights at fragment time
//      (w' = fl(alpha w), one more rounding at 2^-24), so main and residual share one accumulator: v = acc + beta + bd.
// ------------------------------------------------------------------------ //
template <int CIN, int COUT, int RC>
__global__ void al_dcn_wfrag_kernel(const float* __restrict__ w /*[ci][tap][co]*/, const float* __restrict__ alpha,
                                    const float* __restrict__ wd /*[ci RC][co]*/, _Float16* __restrict__ wf, int* range_flag) {
    constexpr int KSC = CIN / 16, MT = COUT / 32, KSR = RC / 16, NF = (9 * KSC + KSR) * MT;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // (fragment, lane, e)
    if (i >= NF * 512) return;
    const int e = i & 7, lane = (i >> 3) & 63, fr = i >> 9;
    const int mt = fr % MT, ksg = fr / MT;                     // global k-step: 9 taps x KSC, then the residual's KSR
    const int co = 32 * mt + (lane & 31);
    float v;
    if (ksg < 9 * KSC) { const int tap = ksg / KSC, ci = 16 * (ksg % KSC) + 8 * (lane >> 5) + e; v = alpha[co] * w[(ci * 9 + tap) * COUT + co]; }
    else { const int ci = 16 * (ksg - 9 * KSC) + 8 * (lane >> 5) + e; v = wd[ci * COUT + co]; }
    _Float16 hi, lo;
    al_split_weight(v, hi, lo, range_flag);
    wf[((size_t)fr * 2 + 0) * 512 + lane * 8 + e] = hi;
    wf[((size_t)fr * 2 + 1) * 512 + lane * 8 + e] = lo;
}

template <int CIN, int COUT, int RC, int CS = 1>      // CS: workgroups per pixel tile, each with COUT / CS output channels (the 1/32 levels: 10 tiles per frame)
__global__ __launch_bounds__(256) void al_dcn_h_kernel(const float* __restrict__ in /* channel-last [H W][CIN] */, const float* __restrict__ off /*[18][H W]*/,
                                                      const float* __restrict__ res_in /* channel-last [H W][RC] */, int H, int W,
                                                      const _Float16* __restrict__ wf, const float* __restrict__ beta, const float* __restrict__ bd,
                                                      float* __restrict__ out /*[COUT][H W]*/, float* __restrict__ out_cl /*[H W][COUT] or null*/,
                                                      ALCtrl* ctrl, size_t fs) {
    in = fsh(in, blockIdx.z, fs); off = fsh(off, blockIdx.z, fs); res_in = fsh(res_in, blockIdx.z, fs);
    out = fsh(out, blockIdx.z, fs); out_cl = fsh0(out_cl, blockIdx.z, fs); ctrl = fsh(ctrl, blockIdx.z, fs);
    float amax = 0.0f;                                        // (al_range_note below)
    constexpr int CP = CIN + 8, KSC = CIN / 16, MTA = COUT / 32, MT = MTA / CS, KSR = RC / 16, C8 = CIN / 8, PLH = 32 * CP;      // MTA: 32-channel tiles of the layer, MT: of this workgroup
    constexpr int BUFH = 2 * PLH;                                              // one wave's tap buffer (halves): [plane][32 px][CP]
    constexpr int REDF = 4 * MT * 16 * 64;                                     // reduction floats
    constexpr int LDSH = 4 * BUFH > 2 * REDF ? 4 * BUFH : 2 * REDF;
    __shared__ __attribute__((aligned(16))) _Float16 lds[LDSH];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), h = lane >> 5, px = lane & 31;
    const int x0 = (blockIdx.x / CS) * 32, y = blockIdx.y, HW = H * W, mt0 = (blockIdx.x % CS) * MT;
    _Float16* buf = lds + wave * BUFH;
    __shared__ float offs[18 * 32];                           // the tile's offsets, once (they were a dependent global load in front of every sample)
    for (int i = t; i < 18 * 32; i += 256) offs[i] = off[(size_t)(i >> 5) * HW + y * W + min(x0 + (i & 31), W - 1)];
    __syncthreads();
    f32x16 c1[MT], c2[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) { c1[m][r] = 0.0f; c2[m][r] = 0.0f; }
    const _Float16* bl = buf + px * CP + 8 * h;
    const _Float16* af = wf + lane * 8;
#pragma unroll 1
    for (int slot = wave; slot < (RC ? 10 : 9); slot += 4) {
        const int nks = slot == 9 ? KSR : KSC;
        // ---- fill: 32 pixels x (channels / 8) fragments of this slot, G items of a lane at a time with all their corner loads in
        //      flight (one item after the other, every fragment waited out two dependent memory latencies: 50 us per workgroup)
        const int c8n = slot == 9 ? RC / 8 : C8;
        constexpr int G = C8 >= 8 ? 4 : C8 / 2;                 // items per lane and group (32 C8 / 64 items per lane in all)
        for (int it0 = 0; it0 < 32 * c8n / 64; it0 += G) {
            float4 qa[G][4], qb[G][4]; float wq[G][4]; int cq[G], pq[G];
#pragma unroll
            for (int g2 = 0; g2 < G; ++g2) {
                const int idx = min(lane + 64 * (it0 + g2), 32 * c8n - 1);
                const int c8 = idx % c8n, q = idx / c8n, xq = min(x0 + q, W - 1), pix = y * W + xq, c = 8 * c8;
                cq[g2] = c; pq[g2] = q;
                const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
                for (int k = 0; k < 4; ++k) { qa[g2][k] = z; qb[g2][k] = z; wq[g2][k] = 0.0f; }
                if (slot == 9) {
                    qa[g2][0] = *reinterpret_cast<const float4*>(res_in + (size_t)pix * RC + c);
                    qb[g2][0] = *reinterpret_cast<const float4*>(res_in + (size_t)pix * RC + c + 4);
                    wq[g2][0] = 1.0f;
                } else {
                    // torchvision deform_conv2d bilinear sample (al_dcn_col_kernel's arithmetic); offsets from the tile's LDS copy
                    const float ys = (float)(y - 1 + slot / 3) + offs[(2 * slot) * 32 + q];
                    const float xs = (float)(xq - 1 + slot % 3) + offs[(2 * slot + 1) * 32 + q];
                    if (!(ys <= -1.0f || ys >= (float)H || xs <= -1.0f || xs >= (float)W)) {
                        const float fy = floorf(ys), fx = floorf(xs);
                        const int y0 = (int)fy, xx0 = (int)fx, y1 = y0 + 1, xx1 = xx0 + 1;
                        const float ly = ys - fy, lx = xs - fx, hy = 1.0f - ly, hx = 1.0f - lx;
                        const bool m1 = y0 >= 0 && xx0 >= 0, m2 = y0 >= 0 && xx1 <= W - 1, m3 = y1 <= H - 1 && xx0 >= 0, m4 = y1 <= H - 1 && xx1 <= W - 1;
                        const int i1 = m1 ? y0 * W + xx0 : 0, i2 = m2 ? y0 * W + xx1 : 0, i3 = m3 ? y1 * W + xx0 : 0, i4 = m4 ? y1 * W + xx1 : 0;
                        wq[g2][0] = hy * hx; wq[g2][1] = hy * lx; wq[g2][2] = ly * hx; wq[g2][3] = ly * lx;
                        const float* p1 = in + (size_t)i1 * CIN + c; const float* p2 = in + (size_t)i2 * CIN + c;
                        const float* p3 = in + (size_t)i3 * CIN + c; const float* p4 = in + (size_t)i4 * CIN + c;
                        if (m1) { qa[g2][0] = *reinterpret_cast<const float4*>(p1); qb[g2][0] = *reinterpret_cast<const float4*>(p1 + 4); }
                        if (m2) { qa[g2][1] = *reinterpret_cast<const float4*>(p2); qb[g2][1] = *reinterpret_cast<const float4*>(p2 + 4); }
                        if (m3) { qa[g2][2] = *reinterpret_cast<const float4*>(p3); qb[g2][2] = *reinterpret_cast<const float4*>(p3 + 4); }
                        if (m4) { qa[g2][3] = *reinterpret_cast<const float4*>(p4); qb[g2][3] = *reinterpret_cast<const float4*>(p4 + 4); }
                    }
                }
            }
#pragma unroll
            for (int g2 = 0; g2 < G; ++g2) {
                if (lane + 64 * (it0 + g2) >= 32 * c8n) continue;
                float o[8];
                if (slot == 9) {
                    o[0] = qa[g2][0].x; o[1] = qa[g2][0].y; o[2] = qa[g2][0].z; o[3] = qa[g2][0].w;
                    o[4] = qb[g2][0].x; o[5] = qb[g2][0].y; o[6] = qb[g2][0].z; o[7] = qb[g2][0].w;
                } else {
                    const float w1 = wq[g2][0], w2 = wq[g2][1], w3 = wq[g2][2], w4 = wq[g2][3];
                    const float v1[8] = {qa[g2][0].x, qa[g2][0].y, qa[g2][0].z, qa[g2][0].w, qb[g2][0].x, qb[g2][0].y, qb[g2][0].z, qb[g2][0].w};
                    const float v2[8] = {qa[g2][1].x, qa[g2][1].y, qa[g2][1].z, qa[g2][1].w, qb[g2][1].x, qb[g2][1].y, qb[g2][1].z, qb[g2][1].w};
                    const float v3[8] = {qa[g2][2].x, qa[g2][2].y, qa[g2][2].z, qa[g2][2].w, qb[g2][2].x, qb[g2][2].y, qb[g2][2].z, qb[g2][2].w};
                    const float v4[8] = {qa[g2][3].x, qa[g2][3].y, qa[g2][3].z, qa[g2][3].w, qb[g2][3].x, qb[g2][3].y, qb[g2][3].z, qb[g2][3].w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = w1 * v1[e] + w2 * v2[e] + w3 * v3[e] + w4 * v4[e];
                }
                uint4 hi, lo;
                sslam::split8_fast(o, hi, lo, amax);
                *reinterpret_cast<uint4*>(&buf[pq[g2] * CP + cq[g2]]) = hi;
                *reinterpret_cast<uint4*>(&buf[PLH + pq[g2] * CP + cq[g2]]) = lo;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        // ---- this slot's k-steps
        const _Float16* afs = af + (size_t)(slot == 9 ? 9 * KSC : slot * KSC) * MTA * 2 * 512;
        for (int s = 0; s < nks; ++s) {
            const sslam::half8 xh = *reinterpret_cast<const sslam::half8*>(bl + 16 * s);
            const sslam::half8 xl = *reinterpret_cast<const sslam::half8*>(bl + 16 * s + PLH);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const sslam::half8 ah = *reinterpret_cast<const sslam::half8*>(afs + (size_t)((s * MTA + mt0 + m) * 2 + 0) * 512);
                const sslam::half8 al = *reinterpret_cast<const sslam::half8*>(afs + (size_t)((s * MTA + mt0 + m) * 2 + 1) * 512);
                c1[m] = sslam::mfma16(ah, xh, c1[m]);
                c2[m] = sslam::mfma16(ah, xl, c2[m]);
                c2[m] = sslam::mfma16(al, xh, c2[m]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");            // the reads above, before the next slot overwrites the buffer
    }
    al_range_note(amax, ctrl);
    __syncthreads();                                          // every wave is done with its buffer: the memory becomes the reduction buffer
    float* red = reinterpret_cast<float*>(lds);              // [wave][tile][r][lane]
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * MT + m) * 16 + r) * 64 + lane] = c1[m][r] + c2[m][r] * sslam::SPLIT_INV;
    __syncthreads();
    if (x0 + px < W) {
        const int pix = y * W + x0 + px;
#pragma unroll
        for (int j = 0; j < MT * 4; ++j) {                   // wave w finishes accumulator registers w MT 4 .. + MT 4 - 1 (of MT x 16)
            const int qi = wave * MT * 4 + j, m = qi / 16, r = qi % 16, co = 32 * (mt0 + m) + acc_row(r, lane);
            const int o = (m * 16 + r) * 64 + lane;
            float v = ((red[o] + red[o + MT * 1024]) + red[o + 2 * MT * 1024]) + red[o + 3 * MT * 1024];
            v += beta[co];
            if (RC) v += bd[co];
            v = selu(v);
            out[(size_t)co * HW + pix] = v;
            if (out_cl) out_cl[(size_t)pix * COUT + co] = v;
        }
    }
}


// conv weights [ci][tap][co] -> [co][tap*CIN + ci]; 1x1 weights [ci][co] -> [co][ci] (taps = 1)

// 1x1 conv (no bias) + SELU: one thread per pixel produces all 32 outputs (inputs read once,
// weights [ci][32] as wave-uniform scalar loads).  r04: the channel loop is unrolled with every input of the pixel loaded up
// front (rolled, each of its 32 trips waited for its own load: 32 serial memory latencies per thread), and the channel-last
// copy leaves through LDS as whole 128-byte lines per 8 lanes (direct, every lane wrote 16-byte pieces of its own line).
template <int CIN>
__global__ __launch_bounds__(256) void al_gate_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                      int HW, const float* __restrict__ w /*[ci][32]*/,
                                                      float* __restrict__ out_cl /*[HW][32]*/, size_t fs) {
    __shared__ float cl[256 * 33];
    const int p0 = blockIdx.x * blockDim.x, p = p0 + threadIdx.x, pc = min(p, HW - 1);
    in = fsh(in, blockIdx.y, fs); out = fsh(out, blockIdx.y, fs); out_cl = fsh(out_cl, blockIdx.y, fs);
    float v[CIN];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) v[ci] = in[(size_t)ci * HW + pc];
    float acc[32];
#pragma unroll
    for (int o = 0; o < 32; ++o) acc[o] = 0.0f;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
        const float* wp = w + ci * 32;
#pragma unroll
        for (int o = 0; o < 32; ++o) acc[o] = fmaf(v[ci], wp[o], acc[o]);
    }
#pragma unroll
    for (int o = 0; o < 32; ++o) { acc[o] = selu(acc[o]); cl[threadIdx.x * 33 + o] = acc[o]; }
    if (p < HW) {
#pragma unroll
        for (int o = 0; o < 32; ++o) out[(size_t)o * HW + p] = acc[o];
    }
    __syncthreads();
    const int npx = min(256, HW - p0);
    float* dst = out_cl + (size_t)p0 * 32;
    for (int i = threadIdx.x; i < npx * 8; i += 256) {
        const int q = i >> 3, c4 = (i & 7) * 4;
        *reinterpret_cast<float4*>(dst + (size_t)q * 32 + c4) = make_float4(cl[q * 33 + c4], cl[q * 33 + c4 + 1], cl[q * 33 + c4 + 2], cl[q * 33 + c4 + 3]);
    }
}

// small-map variant (1/8, 1/32 resolution): one thread per (co, pixel), more parallelism.  r06: both small levels in ONE launch
// (a launch of a few workgroups is ~5 us of a single frame's dependent chain): the first `blocks_a` workgroups gate level a,
// the rest level b - each thread's arithmetic is what its own launch did.
struct GateSmall { const float* in; float* out; int CIN; int HW; const float* w; float* out_cl; };
__global__ void al_gate_small_kernel(GateSmall a, GateSmall b, int blocks_a, size_t fs) {
    const bool second = (int)blockIdx.x >= blocks_a;
    const GateSmall& q = second ? b : a;
    const int i = ((int)blockIdx.x - (second ? blocks_a : 0)) * blockDim.x + threadIdx.x;
    const int CIN = q.CIN, HW = q.HW;
    if (i >= 32 * HW) return;
    const float* __restrict__ in = fsh(q.in, blockIdx.y, fs);
    float* __restrict__ out = fsh(q.out, blockIdx.y, fs);
    float* __restrict__ out_cl = fsh0(q.out_cl, blockIdx.y, fs);
    const float* __restrict__ w = q.w;           /*[ci][32]*/
    const int co = i / HW, p = i % HW;
    float a0 = 0.0f, a1 = 0.0f;
#pragma unroll 8                 // 16 loads in flight per thread: the loop is pure load latency otherwise
    for (int ci = 0; ci < CIN; ci += 2) {
        a0 = fmaf(in[(size_t)ci * HW + p], w[ci * 32 + co], a0);
        a1 = fmaf(in[(size_t)(ci + 1) * HW + p], w[(ci + 1) * 32 + co], a1);
    }
    const float v = selu(a0 + a1);
    out[i] = v;
    out_cl[(size_t)p * 32 + co] = v;
}

// ------------------------------------------------------------------------ //
//  3. feature aggregation.  F(p) = [ selu(W1 x1(p)) | up2(g2)(p) | up8(g3)(p) | up32(g4)(p) ]
//     (bilinear, align_corners=True).  The aggregate kernel turns F into the first
//     score-head layer s8 = selu(Ws0 F) and 1/||F|| without storing F.
// ------------------------------------------------------------------------ //
struct Pyr {
    const float* x1; const float* g2; const float* g3; const float* g4;
    const float* w1;      // conv1 [16][32]
    int Hp, Wp;
    float* g1cl;          // selu(W1 x1) channel-last [Hp][Wp][32], written by the aggregate kernel
    // align_corners=True source steps (ih - 1) / (Hp - 1) of the three upsampled levels (host-computed:
    // the same correctly rounded float quotient the kernels would form, without a division per tap)
    float sy2, sx2, sy8, sx8, sy32, sx32;
    // channel-last copies [pixel][32] of the three gated levels for the descriptor head: a bilinear
    // tap of all 32 channels is then ONE 128-byte line instead of 32 lines of 32 planes
    const float *g2cl, *g3cl, *g4cl;
    // r04: what the aggregate kernel needs of an upsampled level, formed ONCE at the level's own resolution
    // (al_agg_pre_kernel): [AGG_PRE][pixels] planar per level - see there
    const float *pre2, *pre3, *pre4;
};

__device__ __forceinline__ Pyr pyr_at(Pyr P, int f, size_t fs) {      // the pyramid of frame f (weights w1 shared)
    P.x1 = fsh(P.x1, f, fs); P.g2 = fsh(P.g2, f, fs); P.g3 = fsh(P.g3, f, fs); P.g4 = fsh(P.g4, f, fs);
    P.g1cl = fsh(P.g1cl, f, fs); P.g2cl = fsh(P.g2cl, f, fs); P.g3cl = fsh(P.g3cl, f, fs); P.g4cl = fsh(P.g4cl, f, fs);
    P.pre2 = fsh(P.pre2, f, fs); P.pre3 = fsh(P.pre3, f, fs); P.pre4 = fsh(P.pre4, f, fs);
    return P;
}

struct UpTap { int o00, o01, o10, o11; float w00, w01, w10, w11; };

__device__ __forceinline__ UpTap up_tap(int y, int x, int Hp, int Wp, int S, float sy, float sx) {
    const int ih = Hp / S, iw = Wp / S;
    const float fy = sy * (float)y, fx = sx * (float)x;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < ih - 1), x1 = x0 + (x0 < iw - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
    UpTap t;
    t.o00 = y0 * iw + x0; t.o01 = y0 * iw + x1; t.o10 = y1 * iw + x0; t.o11 = y1 * iw + x1;
    t.w00 = hx; t.w01 = lx; t.w10 = hy; t.w11 = ly;     // combined as hy*(hx*a+lx*b)+ly*(hx*c+lx*d)
    return t;
}
__device__ __forceinline__ float up_eval_cl(const float* __restrict__ p, const UpTap& t, int c) {
    return t.w10 * (t.w00 * p[t.o00 * 32 + c] + t.w01 * p[t.o01 * 32 + c]) +
           t.w11 * (t.w00 * p[t.o10 * 32 + c] + t.w01 * p[t.o11 * 32 + c]);
}
__device__ __forceinline__ float up_eval(const float* __restrict__ p, const UpTap& t) {
    return t.w10 * (t.w00 * p[t.o00] + t.w01 * p[t.o01]) + t.w11 * (t.w00 * p[t.o10] + t.w01 * p[t.o11]);
}

// r04: bilinear upsampling is linear, so what the aggregate kernel needs of an upsampled 32-channel level g at a
// full-resolution pixel - its contribution to the first score-head layer, sum_c Ws0[c][o] up(g_c), and to ||F||^2,
// sum_c up(g_c)^2 - are functions of low-resolution maps formed ONCE per level pixel instead of 32 channels x 4 taps per
// full-resolution pixel (r03: 384 gathers and ~2000 vector instructions per pixel for the three levels; now 126 and ~250):
//     proj[o][p] = sum_c Ws0[c][o] g_c[p]                                  (8 maps)   -> contribution = up(proj[o])
//     S[p] = <g[p], g[p]>, H[p] = <g[p], g[p + x]>, V[p] = <g[p], g[p + y]>, D1[p] = <g[p], g[p + x + y]>,
//     D2[p] = <g[p + x], g[p + y]>                                          (5 maps)
//     sum_c up(g_c)^2 = sum_{t, t'} w_t w_t' <g[t], g[t']> over the four taps = a quadratic form in S, H, V, D1, D2
// (neighbours clamped at the border, where their tap weight is exactly zero).  One thread per level pixel, all three levels
// of a frame in one launch; reads the planar levels.
constexpr int AGG_PRE = 13;
#ifndef AL_PRE_UNROLL
#define AL_PRE_UNROLL 32       // all 128 loads of a thread in flight at once: 5.4 us per frame (rounds of 8 / 16 channels: 11.3 / 18.5 - a memory latency per round)
#endif
__global__ __launch_bounds__(256) void al_agg_pre_kernel(const float* __restrict__ g2cl /* planar [32][pixels] */, const float* __restrict__ g3cl,
                                                         const float* __restrict__ g4cl, const float* __restrict__ ws0 /*[128][8]*/,
                                                         float* __restrict__ pre2, float* __restrict__ pre3, float* __restrict__ pre4,
                                                         int Hp, int Wp, size_t fs) {
    // (xcd_band() here halves this kernel's fetch - 23.3 -> 11.2 MB per frame, its threads read their pixel's right and lower
    //  neighbours - and costs it 20 % of its time, 6.1 -> 7.4 us per frame: eight far-apart streams per plane instead of one; not used)
    const int f = blockIdx.y;
    const int n2 = (Hp / 2) * (Wp / 2), n3 = (Hp / 8) * (Wp / 8), n4 = (Hp / 32) * (Wp / 32);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float* g; float* pre; int S, lvl;
    if (i < n2) { g = g2cl; pre = pre2; S = 2; lvl = 1; }
    else if (i < n2 + n3) { i -= n2; g = g3cl; pre = pre3; S = 8; lvl = 2; }
    else if (i < n2 + n3 + n4) { i -= n2 + n3; g = g4cl; pre = pre4; S = 32; lvl = 3; }
    else return;
    g = fsh(g, f, fs); pre = fsh(pre, f, fs);
    const int ih = Hp / S, iw = Wp / S, n = ih * iw;
    const int y = i / iw, x = i % iw;
    const int xr = x + (x < iw - 1), yd = y + (y < ih - 1);
    const int ia = y * iw + x, ib = y * iw + xr, ic = yd * iw + x, id = yd * iw + xr;
    float pr[8] = {}, ss = 0.0f, hh = 0.0f, vv = 0.0f, d1 = 0.0f, d2 = 0.0f;
    const float* w = ws0 + lvl * 32 * 8;
    // planar reads: lane = pixel, so one load instruction is 256 contiguous bytes per channel (the channel-last copies
    // would make every lane fetch its own 128-byte line: 22 us per frame measured, this form ~3)
#pragma unroll AL_PRE_UNROLL
    for (int c = 0; c < 32; ++c) {
        const float* gc = g + (size_t)c * n;
        const float av = gc[ia], bv = gc[ib], cv = gc[ic], dv = gc[id];
        ss = fmaf(av, av, ss); hh = fmaf(av, bv, hh); vv = fmaf(av, cv, vv);
        d1 = fmaf(av, dv, d1); d2 = fmaf(bv, cv, d2);
#pragma unroll
        for (int o = 0; o < 8; ++o) pr[o] = fmaf(av, w[c * 8 + o], pr[o]);
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) pre[(size_t)o * n + i] = pr[o];
    pre[(size_t)8 * n + i] = ss; pre[(size_t)9 * n + i] = hh; pre[(size_t)10 * n + i] = vv;
    pre[(size_t)11 * n + i] = d1; pre[(size_t)12 * n + i] = d2;
}

#if defined(AL_AGG_LOAD_NOP) && AL_AGG_LOAD_NOP == 3
__device__ unsigned al_dbg_words[64];       // experiment only (sslam_aliked_debug_read(99)): see agg_level
#endif
// contribution of one upsampled level at a full-resolution pixel: s[o] += up(proj[o]), n2 += the quadratic form
__device__ __forceinline__ void agg_level(const float* __restrict__ pre, int n, const UpTap& t, float (&s)[8], float& n2) {
#pragma unroll
    for (int o = 0; o < 8; ++o) s[o] += up_eval(pre + (size_t)o * n, t);
    const float w00 = t.w10 * t.w00, w01 = t.w10 * t.w01, w10 = t.w11 * t.w00, w11 = t.w11 * t.w01;     // hy hx, hy lx, ly hx, ly lx
    const float* S = pre + (size_t)8 * n; const float* H = pre + (size_t)9 * n; const float* V = pre + (size_t)10 * n;
    const float* D1 = pre + (size_t)11 * n; const float* D2 = pre + (size_t)12 * n;
#ifdef AL_AGG_LOAD_NOP
    // experiment (scripts/diag_agg_rnorm.sh, profiles/r06_aggregate_rnorm_diagnosis.md): the ten gathers of the quadratic form
    // in named registers and an asm statement that reads them all - the compiler's s_waitcnt vmcnt lands in FRONT of it - with
    // (1) or without (2) idle cycles before the first instruction that consumes a loaded register
    float s00 = S[t.o00], s01 = S[t.o01], s10 = S[t.o10], s11 = S[t.o11], h00 = H[t.o00], h10 = H[t.o10], v00 = V[t.o00],
          v01 = V[t.o01], d1 = D1[t.o00], d2 = D2[t.o00];
#if AL_AGG_LOAD_NOP == 1
    asm volatile("s_nop 7" : "+v"(s00), "+v"(s01), "+v"(s10), "+v"(s11), "+v"(h00), "+v"(h10), "+v"(v00), "+v"(v01), "+v"(d1), "+v"(d2));
#else
    asm volatile("" : "+v"(s00), "+v"(s01), "+v"(s10), "+v"(s11), "+v"(h00), "+v"(h10), "+v"(v00), "+v"(v01), "+v"(d1), "+v"(d2));
#endif
    const float sq = (w00 * w00 * s00 + w01 * w01 * s01) + (w10 * w10 * s10 + w11 * w11 * s11);
#if AL_AGG_LOAD_NOP == 3
    // (3): is it the LOADED REGISTER that holds a wrong value, or the packed instruction that consumes it?  The six cross products
    // once as the compiler forms them (SLP-vectorised: v_pk_mul_f32 on register pairs) and once by single-lane-width v_mul_f32 in
    // inline assembly ON THE SAME REGISTERS (coefficients laundered through an asm so both forms read the same six values);
    // a lane whose two forms disagree records itself in al_dbg_words
    float c0 = w00 * w01, c1 = w10 * w11, c2 = w00 * w10, c3 = w01 * w11, c4 = w00 * w11, c5 = w01 * w10;
    asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5));
    const float p0 = c0 * h00, p1 = c1 * h10, p2 = c2 * v00, p3 = c3 * v01, p4 = c4 * d1, p5 = c5 * d2;
    float q0, q1, q2, q3, q4, q5;
    asm volatile("v_mul_f32 %0, %6, %12\n\tv_mul_f32 %1, %7, %13\n\tv_mul_f32 %2, %8, %14\n\tv_mul_f32 %3, %9, %15\n\t"
                 "v_mul_f32 %4, %10, %16\n\tv_mul_f32 %5, %11, %17"
                 : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5)
                 : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(c4), "v"(c5), "v"(h00), "v"(h10), "v"(v00), "v"(v01), "v"(d1), "v"(d2));
    const float cr = ((p0 + p1) + (p2 + p3)) + (p4 + p5);
    {
        const int differ = (p0 != q0) | ((p1 != q1) << 1) | ((p2 != q2) << 2) | ((p3 != q3) << 3) | ((p4 != q4) << 4) | ((p5 != q5) << 5);
        if (differ) {
            const unsigned k = atomicAdd(&al_dbg_words[0], 1u);
            atomicOr(&al_dbg_words[1], (unsigned)differ);                         // which of the six products ever differed
            atomicOr(&al_dbg_words[2], 1u << ((threadIdx.x & 63) >> 4));          // which 16-lane group of the wave
            if (k < 8) {                                                          // the first few in full: packed, scalar, both operands
                const float pv[6] = {p0, p1, p2, p3, p4, p5}, qv[6] = {q0, q1, q2, q3, q4, q5}, cv[6] = {c0, c1, c2, c3, c4, c5},
                            gv[6] = {h00, h10, v00, v01, d1, d2};
                const int w = __ffs(differ) - 1;
                unsigned* o = al_dbg_words + 8 + 6 * k;
                o[0] = (unsigned)differ | ((threadIdx.x & 63) << 8) | ((unsigned)n << 16);
                o[1] = __float_as_uint(pv[w]); o[2] = __float_as_uint(qv[w]); o[3] = __float_as_uint(cv[w]); o[4] = __float_as_uint(gv[w]);
                o[5] = blockIdx.x | (blockIdx.y << 8) | (blockIdx.z << 24);
            }
        }
    }
#else
    const float cr = ((w00 * w01) * h00 + (w10 * w11) * h10) + ((w00 * w10) * v00 + (w01 * w11) * v01) +
                     ((w00 * w11) * d1 + (w01 * w10) * d2);
#endif
#else
    const float sq = (w00 * w00 * S[t.o00] + w01 * w01 * S[t.o01]) + (w10 * w10 * S[t.o10] + w11 * w11 * S[t.o11]);
    const float cr = ((w00 * w01) * H[t.o00] + (w10 * w11) * H[t.o10]) + ((w00 * w10) * V[t.o00] + (w01 * w11) * V[t.o01]) +
                     ((w00 * w11) * D1[t.o00] + (w01 * w10) * D2[t.o00]);
#endif
    n2 += fmaf(2.0f, cr, sq);
}

// (r03: the kernel is latency-bound - waves parked 65 % of their cycles, SQ counters - so the level loops carry four
//  channels = 16 gathers in flight per iteration: 49 -> 40 us per frame; eight: the weights start to spill to v_readlane)
#ifndef AL_AGG1_UNROLL
#define AL_AGG1_UNROLL 2
#endif
// r06: NO packed-fp32 instructions in this kernel (AL_AGG_PACKED=1 lifts that, for the experiment scripts only).  Its one known
// fault - 1 / ||F|| wrong by 0.2 - 4 % in lanes 48..63 of a wave, once per few hundred frames and only with other streams'
// kernels on the GPU - was n2 missing the D2 term of one level's quadratic form because its coefficient hy lx came out 0.0 of
// `v_pk_mul_f32 ... op_sel:[0,1]` (see selu_precise above; bisected in the compiler's own assembly, one instruction at a time:
// scripts/agg_isa_patch.py).  Whether the vectoriser forms that instruction depended on an unrelated detail (the exponential of
// the tail); the attribute takes the choice away from it, and isa_guard.py checks every kernel of every build for the form.
#ifndef AL_AGG_PACKED
#define AL_AGG_PACKED 0
#endif
#if AL_AGG_PACKED
#define AL_AGG_TARGET
#else
#define AL_AGG_TARGET __attribute__((target("no-packed-fp32-ops")))
#endif
AL_AGG_TARGET __global__ __launch_bounds__(256) void al_aggregate_kernel(Pyr P0, const float* __restrict__ ws0 /*[128][8]*/,
                                                           float* __restrict__ s8, float* __restrict__ rnorm, size_t fs) {
    const Pyr P = pyr_at(P0, blockIdx.z, fs);
    s8 = fsh(s8, blockIdx.z, fs); rnorm = fsh(rnorm, blockIdx.z, fs);
    // the block's 256 pixels x 32 channels of g1 are one contiguous 32 KiB run of the channel-last
    // map: stage them in LDS ([pixel][33], conflict-free) and write them out as coalesced float4
    __shared__ float g1s[256 * 33];
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    const bool live = x < P.Wp;
    const size_t HW = (size_t)P.Hp * P.Wp, pix = (size_t)y * P.Wp + min(x, P.Wp - 1);
    float xv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) xv[k] = P.x1[k * HW + pix];
    float s[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) s[o] = 0.0f;
    float n2 = 0.0f;
    // channel loops stay rolled (AL_AGG1_UNROLL channels per iteration): fully unrolled, their wave-uniform
    // weights overflow the SGPR file and return through v_readlane
#pragma unroll AL_AGG1_UNROLL
    for (int c = 0; c < 32; ++c) {
        float a = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) a = fmaf(xv[k], P.w1[k * 32 + c], a);
        a = selu_precise<0>(a);
        g1s[threadIdx.x * 33 + c] = a;     // the descriptor head gathers this instead of redoing the 16x32 product
        n2 = fmaf(a, a, n2);
#pragma unroll
        for (int o = 0; o < 8; ++o) s[o] = fmaf(a, ws0[c * 8 + o], s[o]);
    }
    const UpTap t2 = up_tap(y, x, P.Hp, P.Wp, 2, P.sy2, P.sx2), t3 = up_tap(y, x, P.Hp, P.Wp, 8, P.sy8, P.sx8),
                t4 = up_tap(y, x, P.Hp, P.Wp, 32, P.sy32, P.sx32);
    agg_level(P.pre2, (int)(HW / 4), t2, s, n2);
    agg_level(P.pre3, (int)(HW / 64), t3, s, n2);
    agg_level(P.pre4, (int)(HW / 1024), t4, s, n2);
    if (live) {
#if (AL_AGG_FAST_SELU >> 2) & 1          // (experiment: the norm first, the eight exponentials of the tail behind it)
        rnorm[pix] = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
#endif
#pragma unroll
        for (int o = 0; o < 8; ++o) s8[o * HW + pix] = selu_precise<1>(s[o]);
#if (AL_AGG_FAST_SELU >> 3) & 1          // (experiment: 32 idle cycles between the last exponential's chain and the square root)
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(n2));
#endif
#if !((AL_AGG_FAST_SELU >> 2) & 1)
        rnorm[pix] = 1.0f / fmaxf(sqrtf(n2), 1e-12f);            // F.normalize eps
#endif
    }
    // (the header's inline functions - __syncthreads, make_float4 - are compiled with the file's target features and would stay
    //  CALLS from a kernel whose features differ: the barrier and the 16-byte store are written with the builtins they wrap)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    const int npx = min(256, P.Wp - (int)(blockIdx.x * blockDim.x));
    float* dst = P.g1cl + ((size_t)y * P.Wp + blockIdx.x * blockDim.x) * 32;
    for (int i = threadIdx.x; i < npx * 8; i += 256) {           // 16-byte pieces, fully coalesced
        const int p = i >> 3, c4 = (i & 7) * 4;
        const f32x4_t v = {g1s[p * 33 + c4], g1s[p * 33 + c4 + 1], g1s[p * 33 + c4 + 2], g1s[p * 33 + c4 + 3]};
        *reinterpret_cast<f32x4_t*>(dst + (size_t)p * 32 + c4) = v;
    }
}

// Normalised feature vector at integer pixel (y,x) of the UN-padded map: one wave per pixel,
// lane l returns channels l and l+64.
__device__ __forceinline__ float2 feat_pair(const Pyr& P, const float* __restrict__ rnorm, int pl, int pt,
                                            int y, int x, int lane) {
    const int yp = y + pt, xp = x + pl;
    const size_t pix = (size_t)yp * P.Wp + xp;
    float a, b;
    const int c = lane & 31;
    if (lane < 32) {
        a = P.g1cl[pix * 32 + c];                      // one coalesced 128 B line per pixel
        b = up_eval_cl(P.g3cl, up_tap(yp, xp, P.Hp, P.Wp, 8, P.sy8, P.sx8), c);
    } else {
        a = up_eval_cl(P.g2cl, up_tap(yp, xp, P.Hp, P.Wp, 2, P.sy2, P.sx2), c);
        b = up_eval_cl(P.g4cl, up_tap(yp, xp, P.Hp, P.Wp, 32, P.sy32, P.sx32), c);
    }
    const float r = rnorm[pix];
    return make_float2(a * r, b * r);        // channels: lane<32: (c, 64+c) ; lane>=32: (32+c, 96+c)
}

// r04: the same value with every gather addressed as (wave-uniform map pointer) + (32-bit byte offset of the lane): one vector
// add per tap instead of a 64-bit multiply-add chain (the descriptor head's kernels are bound by vector-instruction issue,
// and ~250 of al_sample_kernel's 489 vector instructions were address arithmetic).  Maps are < 4 GB.
// The mix itself with its contractions written out, so that every caller rounds the same way whatever the compiler would
// have chosen around it (al_sample_kernel evaluates it on shared and on per-corner taps and must get the same bits).
__device__ __forceinline__ float up_mix(const UpTap& t, float v00, float v01, float v10, float v11) {
    const float top = fmaf(t.w00, v00, __fmul_rn(t.w01, v01)), bot = fmaf(t.w00, v10, __fmul_rn(t.w01, v11));
    return fmaf(t.w10, top, __fmul_rn(t.w11, bot));
}
__device__ __forceinline__ float up_eval_cl32(const float* __restrict__ p, const UpTap& t, unsigned cb) {
    return up_mix(t, at_b(p, (unsigned)t.o00 * 128u + cb), at_b(p, (unsigned)t.o01 * 128u + cb),
                  at_b(p, (unsigned)t.o10 * 128u + cb), at_b(p, (unsigned)t.o11 * 128u + cb));
}
__device__ __forceinline__ float2 feat_pair32(const Pyr& P, const float* __restrict__ rnorm, int pl, int pt,
                                              int y, int x, int lane) {
    const int yp = y + pt, xp = x + pl;
    const unsigned pix = (unsigned)(yp * P.Wp + xp);
    float a, b;
    const unsigned cb = 4u * (lane & 31);
    if (lane < 32) {
        a = at_b(P.g1cl, pix * 128u + cb);
        b = up_eval_cl32(P.g3cl, up_tap(yp, xp, P.Hp, P.Wp, 8, P.sy8, P.sx8), cb);
    } else {
        a = up_eval_cl32(P.g2cl, up_tap(yp, xp, P.Hp, P.Wp, 2, P.sy2, P.sx2), cb);
        b = up_eval_cl32(P.g4cl, up_tap(yp, xp, P.Hp, P.Wp, 32, P.sy32, P.sx32), cb);
    }
    const float r = rnorm[pix];
    return make_float2(a * r, b * r);
}

// ------------------------------------------------------------------------ //
//  4. score head tail: 3x3 (8->4) SELU, 3x3 (4->4) SELU, 3x3 (4->1), sigmoid.
//     One kernel, intermediate layers kept in LDS; zero padding at the padded-map border.
// ------------------------------------------------------------------------ //
#ifndef AL_ST_H
#define AL_ST_H 8
#endif
constexpr int ST_W = 32, ST_H = AL_ST_H;

__global__ __launch_bounds__(256) void al_score_tail_kernel(const float* __restrict__ s8, int Hp, int Wp,
                                                            const float* __restrict__ w2 /*[8][9][4]*/,
                                                            const float* __restrict__ w4 /*[4][9][4]*/,
                                                            const float* __restrict__ w6 /*[4][9][1]*/,
                                                            float* __restrict__ score, int h, int w, int pl,
                                                            int pt, size_t fs) {
    const Tile3 tb = xcd_band();
    s8 = fsh(s8, tb.z, fs); score = fsh(score, tb.z, fs);
    __shared__ float t0[8][ST_H + 6][ST_W + 6];
    __shared__ float t1[4][ST_H + 4][ST_W + 4];
    __shared__ float t2[4][ST_H + 2][ST_W + 2];
    const int x0 = tb.x * ST_W, y0 = tb.y * ST_H;
    const size_t HW = (size_t)Hp * Wp;
    for (int i = threadIdx.x; i < 8 * (ST_H + 6) * (ST_W + 6); i += 256) {
        const int c = i / ((ST_H + 6) * (ST_W + 6)), rem = i % ((ST_H + 6) * (ST_W + 6));
        const int yy = y0 + rem / (ST_W + 6) - 3, xx = x0 + rem % (ST_W + 6) - 3;
        (&t0[0][0][0])[i] = (yy >= 0 && yy < Hp && xx >= 0 && xx < Wp) ? s8[c * HW + (size_t)yy * Wp + xx] : 0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (ST_H + 4) * (ST_W + 4); i += 256) {
        const int ly = i / (ST_W + 4), lx = i % (ST_W + 4);
        const int yy = y0 + ly - 2, xx = x0 + lx - 2;
        float a[4] = {0, 0, 0, 0};
        if (yy >= 0 && yy < Hp && xx >= 0 && xx < Wp) {
            // one input channel per (rolled) iteration: its 36 wave-uniform weights fit the SGPR file.
            // Fully unrolled, the 288 scalar weights overflow it and come back through ~1000
            // v_readlane per thread - more than the kernel's FMAs.
#pragma unroll 1
            for (int ci = 0; ci < 8; ++ci)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const float v = t0[ci][ly + tap / 3][lx + tap % 3];
#pragma unroll
                    for (int o = 0; o < 4; ++o) a[o] = fmaf(v, w2[(ci * 9 + tap) * 4 + o], a[o]);
                }
            for (int o = 0; o < 4; ++o) a[o] = selu(a[o]);
        }
        for (int o = 0; o < 4; ++o) t1[o][ly][lx] = a[o];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (ST_H + 2) * (ST_W + 2); i += 256) {
        const int ly = i / (ST_W + 2), lx = i % (ST_W + 2);
        const int yy = y0 + ly - 1, xx = x0 + lx - 1;
        float a[4] = {0, 0, 0, 0};
        if (yy >= 0 && yy < Hp && xx >= 0 && xx < Wp) {
#pragma unroll 1
            for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const float v = t1[ci][ly + tap / 3][lx + tap % 3];
#pragma unroll
                    for (int o = 0; o < 4; ++o) a[o] = fmaf(v, w4[(ci * 9 + tap) * 4 + o], a[o]);
                }
            for (int o = 0; o < 4; ++o) a[o] = selu(a[o]);
        }
        for (int o = 0; o < 4; ++o) t2[o][ly][lx] = a[o];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ST_H * ST_W; i += 256) {        // (ST_H = 8: one pixel per thread)
        const int lx = i & (ST_W - 1), ly = i / ST_W;
        const int yy = y0 + ly, xx = x0 + lx;
        float a = 0.0f;
        for (int ci = 0; ci < 4; ++ci)
            for (int tap = 0; tap < 9; ++tap) a = fmaf(t2[ci][ly + tap / 3][lx + tap % 3], w6[ci * 9 + tap], a);
        const int uy = yy - pt, ux = xx - pl;
        if (uy >= 0 && uy < h && ux >= 0 && ux < w) score[(size_t)uy * w + ux] = 1.0f / (1.0f + expf(-a));
    }
}

// ------------------------------------------------------------------------ //
//  5. DKD: simple_nms (5x5, two recovery rounds) + border + threshold -> candidates
// ------------------------------------------------------------------------ //
constexpr int NHALO = 10;     // dependency radius of simple_nms: 2 + 4 + 4

// r04: NMS with a WAVE owning a tile and no LDS (the LDS-tile form it replaced: scripts/ubench/aliked_superseded_r04.hpp, al_nms_kernel): lane = column (64 of them, 44 + the 10-pixel halo either side), the
// R rows of the column in registers.  A 5-wide row maximum is four DPP wave shifts fused into v_max (lanes shifted in from
// outside the wave see themselves = an absent neighbour, as the LDS tile's rim was), a 5-tall column maximum three register
// v_max per row (pairs, pairs of pairs, + one), and the 0/1 maps (max_mask, supp) are ONE 64-bit word per lane - bit y = row y
// of the lane's column - so their 5 x 5 dilation is ten shifts and eight DPP ORs for the whole tile instead of a pooled
// float map (r03 form: 3.7 x halo redundancy, ~60 LDS operations per element, 9.6 us per frame; max is exact and
// associative, so the order of the pooling steps changes nothing: nms is bit-identical.  block_sum's partition changes, i.e.
// the rounding of the mean score the no-candidate fallback thresholds on).
#ifndef AL_NMS_ROWS
#define AL_NMS_ROWS 48
#endif
constexpr int NW_R = AL_NMS_ROWS, NW_OW = 64 - 2 * NHALO, NW_OH = NW_R - 2 * NHALO;
static_assert(NW_R <= 64 && NW_OH > 0, "a column's rows are the bits of one 64-bit word");

// lane i <- lane i - 1 / lane i + 1; the wave's first / last lane reads 0 (bound_ctrl): the identity of the mask ORs, and for the
// float maxima one more wrong value in a rim lane, which the halo keeps out of the central columns like the rest of the rim
__device__ __forceinline__ unsigned nw_up(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ unsigned nw_dn(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true); }
__device__ __forceinline__ float nw_up(float v) { return __builtin_bit_cast(float, nw_up(__builtin_bit_cast(unsigned, v))); }
__device__ __forceinline__ float nw_dn(float v) { return __builtin_bit_cast(float, nw_dn(__builtin_bit_cast(unsigned, v))); }
__device__ __forceinline__ float nw_row5(float v) {
    const float l1 = fmaxf(v, nw_up(v)), l2 = fmaxf(l1, nw_up(l1));       // columns x-2 .. x
    const float r1 = fmaxf(v, nw_dn(v)), r2 = fmaxf(r1, nw_dn(r1));       // columns x .. x+2
    return fmaxf(l2, r2);
}
// 5 x 5 maximum of the tile held as in[row] per lane
__device__ __forceinline__ void nw_pool5(const float (&in)[NW_R], float (&v)[NW_R]) {
    float hrow[NW_R], p[NW_R], q[NW_R];
#pragma unroll
    for (int y = 0; y < NW_R; ++y) hrow[y] = nw_row5(in[y]);
#pragma unroll
    for (int y = 0; y + 1 < NW_R; ++y) p[y] = fmaxf(hrow[y], hrow[y + 1]);          // rows y, y+1
#pragma unroll
    for (int y = 0; y + 3 < NW_R; ++y) q[y] = fmaxf(p[y], p[y + 2]);                // rows y .. y+3
#pragma unroll
    for (int y = 0; y < NW_R; ++y) {
        if (y == 0) v[y] = fmaxf(p[0], hrow[2]);
        else if (y == 1) v[y] = q[0];
        else if (y + 2 < NW_R) v[y] = fmaxf(q[y - 2], hrow[y + 2]);                 // rows y-2 .. y+2
        else if (y + 2 == NW_R) v[y] = q[NW_R - 4];
        else v[y] = fmaxf(hrow[NW_R - 3], p[NW_R - 2]);
    }
}
__device__ __forceinline__ unsigned long long nw_dilate5(unsigned long long c) {
    const unsigned long long vv = c | (c << 1) | (c << 2) | (c >> 1) | (c >> 2);
    unsigned lo = (unsigned)vv, hi = (unsigned)(vv >> 32);
    const unsigned l1 = lo | nw_up(lo), l2 = l1 | nw_up(l1), r1 = lo | nw_dn(lo), r2 = r1 | nw_dn(r1);
    const unsigned h1 = hi | nw_up(hi), h2 = h1 | nw_up(h1), g1 = hi | nw_dn(hi), g2 = g1 | nw_dn(g1);
    return ((unsigned long long)(h2 | g2) << 32) | (l2 | r2);
}

// The tile's pixels above the detection threshold go straight into the candidate list (what the first al_collect_kernel
// launch did from the nms map: one more pass over it, 10 us of a single-frame extraction): the list is unordered either way.
// (Four tiles = four independent waves per workgroup, so that the list's counter sees one returning atomic per four tiles:
//  ~300 of them on one address per frame were 10 us.)
__global__ __launch_bounds__(256) void al_nms_wave_kernel(const float* __restrict__ score, int h, int w, int nbx, int n_tiles,
                                                          float* __restrict__ nms, float* __restrict__ block_sum, float thr,
                                                         unsigned long long* __restrict__ cand, int cap,
                                                         ALCtrl* __restrict__ ctrl, unsigned* __restrict__ hist, size_t fs) {
    score = fsh(score, blockIdx.y, fs); nms = fsh(nms, blockIdx.y, fs); block_sum = fsh(block_sum, blockIdx.y, fs);
    cand = fsh(cand, blockIdx.y, fs); ctrl = fsh(ctrl, blockIdx.y, fs); hist = fsh(hist, blockIdx.y, fs);
    __shared__ int wtot[4];
    __shared__ int s_base;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile_raw = (int)blockIdx.x * 4 + wv;
    const bool act = tile_raw < n_tiles;                           // (a spare wave of the last workgroup redoes tile 0 and keeps nothing)
    const int tile = act ? tile_raw : 0, tbx = tile % nbx, tby = tile / nbx;
    const int xx = tbx * NW_OW - NHALO + lane, y0 = tby * NW_OH - NHALO;
    const bool xin = xx >= 0 && xx < w;
    const unsigned xc = (unsigned)min(max(xx, 0), w - 1);
    float s[NW_R];
#pragma unroll
    for (int y = 0; y < NW_R; ++y) {            // unconditional loads at a clamped address: all R in flight
        const int yy = y0 + y;
        const float v = score[(unsigned)min(max(yy, 0), h - 1) * (unsigned)w + xc];
        // a row outside the map: + (-inf) from a scalar register (one lane mask per row would spill the scalar file)
        s[y] = (xin ? v : -INFINITY) + ((yy >= 0 && yy < h) ? 0.0f : -INFINITY);
    }
    auto equal_bits = [](const float (&a)[NW_R], const float (&b)[NW_R]) {          // bit y: a[y] == b[y] and a[y] is a score
        unsigned lo = 0u, hi = 0u;
#pragma unroll
        for (int y = 0; y < NW_R; ++y) {
            const bool e = a[y] == b[y] && a[y] > -INFINITY;
            if (y < 32) lo |= e ? (1u << y) : 0u; else hi |= e ? (1u << (y - 32)) : 0u;
        }
        return ((unsigned long long)hi << 32) | lo;
    };
    float t[NW_R];
    nw_pool5(s, t);
    unsigned long long mask = equal_bits(s, t);                    // max_mask = scores == max_pool(scores)
#pragma unroll 1
    for (int round = 0; round < 2; ++round) {
        const unsigned long long supp = nw_dilate5(mask);         // supp = max_pool(max_mask) > 0
        float q[NW_R];
#pragma unroll
        for (int y = 0; y < NW_R; ++y) {
            const bool sp = (y < 32 ? ((unsigned)supp >> y) : ((unsigned)(supp >> 32) >> (y - 32))) & 1u;
            q[y] = (s[y] == -INFINITY) ? -INFINITY : (sp ? 0.0f : s[y]);
        }
        nw_pool5(q, t);
        mask |= equal_bits(q, t) & ~supp;                          // max_mask |= new_max & ~supp
    }
    float lsum = 0.0f;
    const bool xout = act && lane >= NHALO && lane < NHALO + NW_OW && xx < w;
    const bool xkeep = xout & (xx >= 2) & (xx < w - 2);
    auto kept = [&](int y, int yy) {                               // the nms map's value at row y of this lane's column
        const bool mk = (y < 32 ? ((unsigned)mask >> y) : ((unsigned)(mask >> 32) >> (y - 32))) & 1u;
        const bool rowkeep = (yy >= 2) & (yy < h - 2);             // border of `radius` (bitwise: no short-circuit branches)
        return (mk & xkeep & rowkeep) ? s[y] : 0.0f;
    };
    // order: count, claim the list range (one returning atomic per workgroup), THEN write the nms rows - the atomic's round
    // trip and the stores' are in flight together - and only then use the range
    int cnt = 0;
#pragma unroll
    for (int y = NHALO; y < NHALO + NW_OH; ++y) cnt += kept(y, y0 + y) > thr;
    int incl = cnt;
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
    if (lane == 63) wtot[wv] = incl;
    __syncthreads();
    int woff = 0, total = 0;
    for (int i = 0; i < 4; ++i) { if (i < wv) woff += wtot[i]; total += wtot[i]; }
    int base = 0;
    if (threadIdx.x == 0 && total) base = atomicAdd(&ctrl->n_cand, total);
    if (xout)
#pragma unroll
    for (int y = NHALO; y < NHALO + NW_OH; ++y) {
        const int yy = y0 + y;
        if (yy < h) {
            nms[(unsigned)yy * (unsigned)w + (unsigned)xx] = kept(y, yy);
            lsum += s[y];
        }
    }
    for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o);
    if (lane == 0 && act) block_sum[tile] = lsum;
    if (total == 0) return;                                        // (the same answer in every wave)
    if (threadIdx.x == 0) { s_base = base; ctrl->found = 1; }
    __syncthreads();
    int pos = s_base + woff + incl - cnt;
    // (a wave alone on its SIMD issues an instruction every ~5 cycles: this pass is kept to a dozen per row - the capacity
    //  test is made once for the workgroup)
    const bool fits = s_base + total <= cap;
    if (!fits && threadIdx.x == 0) ctrl->overflow = 1;
#pragma unroll
    for (int y = NHALO; y < NHALO + NW_OH; ++y) {
        const int yy = y0 + y;
        const float v = kept(y, yy);                               // (0 outside the tile's own pixels: rows >= h - 2, columns with !xout)
        const bool c = v > thr;
        if (c & (fits | (pos < cap))) {
            const unsigned i = (unsigned)yy * (unsigned)w + (unsigned)xx;
            cand[pos] = ((unsigned long long)__float_as_uint(v) << 32) | (0xffffffffu - i);
        }
        if (c) {
            atomicAdd(&hist[min((int)(v * (float)HBINS), HBINS - 1)], 1u);     // scores are in (0, 1]
            ++pos;
        }
    }
}

// collect pixels with nms > thr into an (unordered) candidate list of 64-bit keys:
// key = score_bits << 32 | (0xffffffff - index)  -> larger key = better (score desc, index asc)
// (end of r04: launched for the mean-score fallback only - `fallback` = 1, returns at once when the NMS waves found candidates
//  above the detection threshold; the detection-threshold pass itself is the tail of al_nms_wave_kernel)
constexpr int COLLECT_PPT = 16;     // pixels per thread: 4096 per block -> ~80 blocks, one same-address atomic each

__global__ __launch_bounds__(256) void al_collect_kernel(const float* __restrict__ nms, int n_px, float thr,
                                                         int fallback, const float* __restrict__ block_sum,
                                                         int n_blocks, unsigned long long* __restrict__ cand,
                                                         int cap, ALCtrl* __restrict__ ctrl,
                                                         unsigned* __restrict__ hist, size_t fs) {
    __shared__ int wcnt[4];
    __shared__ int s_base;
    nms = fsh(nms, blockIdx.y, fs); block_sum = fsh(block_sum, blockIdx.y, fs); cand = fsh(cand, blockIdx.y, fs);
    ctrl = fsh(ctrl, blockIdx.y, fs); hist = fsh(hist, blockIdx.y, fs);
    if (fallback) {
        if (ctrl->found) return;                    // the normal threshold found keypoints (written by the FIRST launch only:
                                                    // every thread of this one sees one answer)
        float s = 0.0f;                             // mean of the raw score map, fixed summation order
        for (int i = 0; i < n_blocks; ++i) s += block_sum[i];
        thr = s / (float)n_px;
    }
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int base_px = blockIdx.x * 256 * COLLECT_PPT;
    float v[COLLECT_PPT];
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < COLLECT_PPT; ++j) {
        const int i = base_px + j * 256 + t;
        v[j] = i < n_px ? nms[i] : 0.0f;
        cnt += (i < n_px && v[j] > thr);
    }
    // exclusive scan of the per-thread counts over the block
    int incl = cnt;
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
    if (lane == 63) wcnt[wave] = incl;
    __syncthreads();
    int woff = 0, total = 0;
    for (int w = 0; w < 4; ++w) { if (w < wave) woff += wcnt[w]; total += wcnt[w]; }
    if (t == 0) {
        s_base = total ? atomicAdd(&ctrl->n_cand, total) : 0;
        if (total && !fallback) ctrl->found = 1;
    }
    __syncthreads();
    int pos = s_base + woff + incl - cnt;
#pragma unroll
    for (int j = 0; j < COLLECT_PPT; ++j) {
        const int i = base_px + j * 256 + t;
        if (i < n_px && v[j] > thr) {
            if (pos < cap) cand[pos] = ((unsigned long long)__float_as_uint(v[j]) << 32) | (unsigned)(0xffffffffu - (unsigned)i);
            else ctrl->overflow = 1;
            atomicAdd(&hist[min((int)(v[j] * (float)HBINS), HBINS - 1)], 1u);     // scores are in (0, 1]
            ++pos;
        }
    }
}
// kornia get_gaussian_kernel1d in fp32: gk[0..kx) horizontal taps, gk[32..32+ky) vertical taps
__device__ void al_taps(float* __restrict__ gk, int kx, float sx, int ky, float sy) {
    for (int pass = 0; pass < 2; ++pass) {
        const int ks = pass ? ky : kx;
        const float sigma = pass ? sy : sx;
        float* o = gk + 32 * pass;
        float sum = 0.0f;
        for (int i = 0; i < ks; ++i) {
            float x = (float)(i - ks / 2);
            if (ks % 2 == 0) x += 0.5f;
            o[i] = expf(-(x * x) / (2.0f * sigma * sigma));
            sum += o[i];
        }
        for (int i = 0; i < ks; ++i) o[i] /= sum;
    }
}
// first launch of a sequence: per-frame control block and score histogram back to zero (one block per frame); the first block
// also writes the blur taps (r04: three launches - reset, taps, the fallback flag between the collects - were 14 us of a
// single-frame extraction's 405)
__global__ void al_reset_kernel(ALCtrl* __restrict__ ctrl, unsigned* __restrict__ hist, size_t fs, float* __restrict__ gk,
                                int kx, float sx, int ky, float sy) {
    if (blockIdx.x == 0 && threadIdx.x == 64) al_taps(gk, kx, sx, ky, sy);
    ctrl = fsh(ctrl, blockIdx.x, fs); hist = fsh(hist, blockIdx.x, fs);
    if (threadIdx.x < sizeof(ALCtrl) / 4) reinterpret_cast<int*>(ctrl)[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < HBINS; i += blockDim.x) hist[i] = 0u;
}

// ------------------------------------------------------------------------ //
//  6. selection: top-n_limit by (score desc, index asc) or all in raster order.  One block per frame picks the
//     SET of keys (histogram cut + ordered edge bin); their ORDER is a rank count spread over the chip in
//     al_refine_kernel (r03: a 66-pass bitonic sort of the 2048 keys in this one block was 30 of its 45 us).
// ------------------------------------------------------------------------ //
constexpr int SEL_CAP = 8192;      // max keypoints (sort capacity)

__device__ void bitonic_sort_desc(unsigned long long* a, int n_pow2) {
    for (int k = 2; k <= n_pow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n_pow2; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long x = a[i], y = a[ixj];
                    const bool up = (i & k) == 0;         // descending overall
                    if (up ? (x < y) : (x > y)) { a[i] = y; a[ixj] = x; }
                }
            }
            __syncthreads();
        }
}

constexpr int EDGE_CAP = 2048;     // candidates allowed in the cut bin before falling back to radix select

__global__ __launch_bounds__(1024) void al_select_kernel(const unsigned long long* __restrict__ cand, int cap,
                                                         int n_limit, unsigned long long* __restrict__ sel_keys,
                                                         ALCtrl* __restrict__ ctrl, const unsigned* __restrict__ hist_g,
                                                         size_t fs) {
    cand = fsh(cand, blockIdx.x, fs); sel_keys = fsh(sel_keys, blockIdx.x, fs); ctrl = fsh(ctrl, blockIdx.x, fs);
    hist_g = fsh(hist_g, blockIdx.x, fs);                                       // one block per frame
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];   // SEL_CAP + EDGE_CAP
    unsigned long long* edge = keys + SEL_CAP;
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_remaining;
    __shared__ int s_count, s_edge, s_cutbin, s_above;
    __shared__ unsigned wsum[16];
    const int n = min(ctrl->n_cand, cap);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int n_sel;
    if (n <= n_limit) {
        // raster order: sort by index ascending = key with score bits cleared, descending on (~index)
        for (int i = t; i < SEL_CAP; i += blockDim.x) keys[i] = i < n ? (cand[i] & 0xffffffffull) : 0ull;
        n_sel = n;
        __syncthreads();
    } else {
        // 1. cut bin from the score histogram (filled by al_collect): the largest bin b with
        //    count(bins > b) < n_limit <= count(bins >= b).  Thread t owns bins 4t..4t+3 (descending scan).
        unsigned loc[4], tot = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { loc[q] = hist_g[HBINS - 1 - (4 * t + q)]; tot += loc[q]; }
        unsigned incl = tot;
        for (int o = 1; o < 64; o <<= 1) { const unsigned v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        unsigned base = 0;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        unsigned run = base + incl - tot;              // candidates in bins above this thread's first bin
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (run < (unsigned)n_limit && run + loc[q] >= (unsigned)n_limit) { s_cutbin = HBINS - 1 - (4 * t + q); s_above = (int)run; }
            run += loc[q];
        }
        if (t == 0) { s_count = 0; s_edge = 0; }
        for (int i = t; i < SEL_CAP; i += blockDim.x) keys[i] = 0ull;
        __syncthreads();
        const int cutbin = s_cutbin, above = s_above;
        // 2. candidates above the cut bin are in; those in the cut bin go to the edge list
        for (int i = t; i < n; i += blockDim.x) {
            const unsigned long long k = cand[i];
            const int bin = min((int)(__uint_as_float((unsigned)(k >> 32)) * (float)HBINS), HBINS - 1);
            if (bin > cutbin) { const int pos = atomicAdd(&s_count, 1); if (pos < SEL_CAP) keys[pos] = k; }
            else if (bin == cutbin) { const int pos = atomicAdd(&s_edge, 1); if (pos < EDGE_CAP) edge[pos] = k; }
        }
        __syncthreads();
        const int n_edge = s_edge, need = n_limit - above;
        if (n_edge <= EDGE_CAP) {
            // 3. order the edge list, take its best `need` keys
            for (int i = n_edge + t; i < EDGE_CAP; i += blockDim.x) edge[i] = 0ull;
            __syncthreads();
            int p2e = 2;
            while (p2e < n_edge) p2e <<= 1;
            bitonic_sort_desc(edge, p2e);
            for (int i = t; i < need; i += blockDim.x) keys[above + i] = edge[i];
            __syncthreads();
        } else {
            // fallback (degenerate score maps: thousands of ties in one bin): exact 8-pass radix select
            unsigned long long prefix = 0ull;
            int remaining = n_limit;
            for (int shift = 56; shift >= 0; shift -= 8) {
                for (int i = t; i < 256; i += blockDim.x) hist[i] = 0;
                __syncthreads();
                const unsigned long long mask_hi = shift == 56 ? 0ull : (~0ull << (shift + 8));
                for (int i = t; i < n; i += blockDim.x) {
                    const unsigned long long k = cand[i];
                    if ((k & mask_hi) == prefix) atomicAdd(&hist[(k >> shift) & 255], 1u);
                }
                __syncthreads();
                if (t == 0) {
                    int rem = remaining, d = 255;
                    for (; d > 0; --d) {
                        if ((int)hist[d] >= rem) break;
                        rem -= hist[d];
                    }
                    s_prefix = d; s_remaining = rem;
                }
                __syncthreads();
                prefix |= (unsigned long long)s_prefix << shift;
                remaining = s_remaining;
                __syncthreads();
            }
            if (t == 0) s_count = 0;
            for (int i = t; i < SEL_CAP; i += blockDim.x) keys[i] = 0ull;
            __syncthreads();
            for (int i = t; i < n; i += blockDim.x) {
                const unsigned long long k = cand[i];
                if (k >= prefix) { const int pos = atomicAdd(&s_count, 1); if (pos < SEL_CAP) keys[pos] = k; }
            }
            __syncthreads();
        }
        n_sel = n_limit;
    }
    for (int i = t; i < n_sel; i += blockDim.x) sel_keys[i] = keys[i];         // the set; al_refine_kernel ranks it
    if (t == 0) ctrl->n_kp = n_sel;
}

// output position of every selected key = the number of selected keys above it (keys are distinct: score bits over
// ~index, so descending key order is score descending, index ascending - torch.topk's order / raster order), counted by
// 8 lanes per key over 1/8 of the set each; then soft-argmax refinement + score sampling (DKD.forward sub_pixel=True)
// by the first lane of each group, written at that position
constexpr int REFINE_KPB = 32;       // keys per 256-thread block
__global__ __launch_bounds__(256) void al_refine_kernel(const float* __restrict__ score, int h, int w,
                                                        const unsigned long long* __restrict__ sel_keys,
                                                        int* __restrict__ kp_index, float* __restrict__ kp_norm,
                                                        float* __restrict__ kp_score, const ALCtrl* __restrict__ ctrl, size_t fs) {
    score = fsh(score, blockIdx.y, fs); kp_index = fsh(kp_index, blockIdx.y, fs); kp_norm = fsh(kp_norm, blockIdx.y, fs);
    kp_score = fsh(kp_score, blockIdx.y, fs); ctrl = fsh(ctrl, blockIdx.y, fs); sel_keys = fsh(sel_keys, blockIdx.y, fs);
    const int n = ctrl->n_kp;
    const int p = blockIdx.x * REFINE_KPB + (threadIdx.x >> 3), c = threadIdx.x & 7;
    if (blockIdx.x * REFINE_KPB >= n) return;
    const unsigned long long key = sel_keys[p < n ? p : 0];
    // rank = number of larger keys.  r04: the keys pass through LDS in chunks of 2 048 (read straight from global memory each of
    // a thread's n / 8 compares waited for its own load: 28 us for one frame's 2 048 keypoints, now ~6)
    __shared__ unsigned long long sk[2048];
    int above = 0;
    for (int j0 = 0; j0 < n; j0 += 2048) {
        const int m = min(2048, n - j0);
        __syncthreads();
#pragma unroll 8
        for (int j = threadIdx.x; j < m; j += 256) sk[j] = sel_keys[j0 + j];
        __syncthreads();
#pragma unroll 8
        for (int j = c; j < m; j += 8) above += sk[j] > key;
    }
    above += __shfl_xor(above, 1);
    above += __shfl_xor(above, 2);
    above += __shfl_xor(above, 4);
    // the 5 x 5 soft-argmax of a keypoint on its eight lanes: lane c loads and exponentiates elements c, c + 8, c + 16 (, 24); the
    // sums run on lane 0 over the elements in raster order, as one lane computed them before (bit-identical; the 25 dependent
    // loads and 25 expf of one lane in eight were 17 of the kernel's 28 us)
    const bool live = p < n;
    const int i = above;
    const int idx = (int)(0xffffffffu - (unsigned)(key & 0xffffffffull)), x = idx % w, y = idx / w;
    float pe[4], mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = c + 8 * j, yy = y + k / 5 - 2, xx = x + k % 5 - 2;
        pe[j] = (live && k < 25 && yy >= 0 && yy < h && xx >= 0 && xx < w) ? score[(size_t)yy * w + xx] : 0.0f;   // unfold zero pad
        if (k < 25) mx = fmaxf(mx, pe[j]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2)); mx = fmaxf(mx, __shfl_xor(mx, 4));
#pragma unroll
    for (int j = 0; j < 4; ++j) pe[j] = expf((pe[j] - mx) / 0.1f);
    float se = 0.0f, sx = 0.0f, sy = 0.0f;
    const int lane0 = (threadIdx.x & 63) & ~7;
#pragma unroll
    for (int k = 0; k < 25; ++k) {
        const float e = __shfl(pe[k >> 3], lane0 + (k & 7));
        se += e;
        sx += e * (float)(k % 5 - 2);
        sy += e * (float)(k / 5 - 2);
    }
    if (!live || c != 0) return;
    kp_index[i] = idx;
    const float kx = ((float)x + sx / se) / (float)(w - 1) * 2.0f - 1.0f;
    const float ky = ((float)y + sy / se) / (float)(h - 1) * 2.0f - 1.0f;
    kp_norm[2 * i] = kx;
    kp_norm[2 * i + 1] = ky;
    // grid_sample(bilinear, align_corners=True, zeros padding)
    const float ix = (kx + 1.0f) / 2.0f * (float)(w - 1), iy = (ky + 1.0f) / 2.0f * (float)(h - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
    auto at = [&](int yy, int xx) { return (yy >= 0 && yy < h && xx >= 0 && xx < w) ? score[(size_t)yy * w + xx] : 0.0f; };
    kp_score[i] = at(y0, x0) * (wx0 * wy0) + at(y0, x1) * (wx1 * wy0) + at(y1, x0) * (wx0 * wy1) + at(y1, x1) * (wx1 * wy1);
}

// ------------------------------------------------------------------------ //
//  7. SDDH descriptor head
// ------------------------------------------------------------------------ //
// 3x3 patch of the normalised feature map around each keypoint -> patch[n][c*9 + tap]
// (get_patches corner rule); one wave per (keypoint, patch row)
__global__ __launch_bounds__(256) void al_patch_kernel(Pyr P0, const float* __restrict__ rnorm, int pl, int pt,
                                                       int h, int w, const float* __restrict__ kp_norm,
                                                       float* __restrict__ patch, const ALCtrl* __restrict__ ctrl, size_t fs) {
    const Pyr P = pyr_at(P0, blockIdx.y, fs);
    rnorm = fsh(rnorm, blockIdx.y, fs); kp_norm = fsh(kp_norm, blockIdx.y, fs); patch = fsh(patch, blockIdx.y, fs);
    ctrl = fsh(ctrl, blockIdx.y, fs);
    // one wave per (keypoint, patch row): its three taps' 42 gathers are in flight together (one wave per tap was pure
    // latency: three quarters of the wave cycles parked in s_waitcnt)
    const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (a scalar)
    const int n = gw / 3, trow = gw % 3;
    if (n >= ctrl->n_kp) return;
    const float kx = (kp_norm[2 * n] / 2.0f + 0.5f) * (float)(w - 1);
    const float ky = (kp_norm[2 * n + 1] / 2.0f + 0.5f) * (float)(h - 1);
    // corner = (long(kwh) - ps/2 + 1).long(), clamped to [0, w-1-ps] x [0, h-1-ps]
    int cx = (int)((float)(int)kx - 1.5f + 1.0f), cy = (int)((float)(int)ky - 1.5f + 1.0f);
    cx = min(max(cx, 0), w - 1 - 3);
    cy = min(max(cy, 0), h - 1 - 3);
    float2 f[3];
#pragma unroll
    for (int tc = 0; tc < 3; ++tc) f[tc] = feat_pair32(P, rnorm, pl, pt, cy + trow, cx + tc, lane);
    const int c = lane & 31;
    const int ca = lane < 32 ? c : 32 + c, cb = lane < 32 ? 64 + c : 96 + c;
    float* dst = patch + (size_t)n * 1152;
#pragma unroll
    for (int tc = 0; tc < 3; ++tc) {
        dst[ca * 9 + 3 * trow + tc] = f[tc].x;
        dst[cb * 9 + 3 * trow + tc] = f[tc].y;
    }
}

// offsets = clamp(conv1x1(selu(h32)) + b) ; sample positions in un-padded pixel coordinates
constexpr int SDDH_KSPLIT = 4;

__global__ void al_offsets_kernel(const float* __restrict__ h32 /*[KSPLIT][cap][32] partial pre-activations*/,
                                  int cap, const float* __restrict__ b1,
                                  const float* __restrict__ w2 /*[32][32] (o,i)*/, const float* __restrict__ b2,
                                  const float* __restrict__ kp_norm, int h, int w, float max_off,
                                  float* __restrict__ pos /*[n][16][2]*/, const ALCtrl* __restrict__ ctrl, size_t fs) {
    h32 = fsh(h32, blockIdx.y, fs); kp_norm = fsh(kp_norm, blockIdx.y, fs); pos = fsh(pos, blockIdx.y, fs); ctrl = fsh(ctrl, blockIdx.y, fs);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = i / 32, o = i % 32;
    if (n >= ctrl->n_kp) return;
    float acc = 0.0f;
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
        float hv = b1[k];
#pragma unroll
        for (int z = 0; z < SDDH_KSPLIT; ++z) hv += h32[((size_t)z * cap + n) * 32 + k];
        acc = fmaf(selu(hv), w2[o * 32 + k], acc);
    }
    acc = fminf(fmaxf(acc + b2[o], -max_off), max_off);
    // offset[:, :, 0, 0].view(n, 2, M).permute(0, 2, 1): channel o < 16 -> x of position o, else y
    const int p = o & 15, comp = o >> 4;
    const float k0 = (kp_norm[2 * n + comp] / 2.0f + 0.5f) * (float)((comp ? h : w) - 1);
    pos[(n * 16 + p) * 2 + comp] = k0 + acc;
}

// bilinear sample (grid_sample align_corners=True, zeros padding) of the normalised feature map at
// the 16 positions of each keypoint -> sampled[n*16 + p][128]; one wave per (keypoint, position)
__global__ __launch_bounds__(256) void al_sample_kernel(Pyr P0, const float* __restrict__ rnorm, int pl, int pt,
                                                        int h, int w, const float* __restrict__ pos,
                                                        _Float16* __restrict__ sampled /* hi plane [rows][128]; lo plane `lo_off` halves behind */,
                                                        size_t lo_off, ALCtrl* ctrl, size_t fs) {
    const Pyr P = pyr_at(P0, blockIdx.y, fs);
    rnorm = fsh(rnorm, blockIdx.y, fs); pos = fsh(pos, blockIdx.y, fs); sampled = fsh(sampled, blockIdx.y, fs); ctrl = fsh(ctrl, blockIdx.y, fs);
    const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (a scalar)
    if (gw >= ctrl->n_kp * 16) return;
    // pos -> normalised -> back to pixels exactly as grid_sample does
    const float gx = 2.0f * pos[gw * 2] / (float)(w - 1) - 1.0f, gy = 2.0f * pos[gw * 2 + 1] / (float)(h - 1) - 1.0f;
    const float ix = (gx + 1.0f) / 2.0f * (float)(w - 1), iy = (gy + 1.0f) / 2.0f * (float)(h - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
    // all four corners are fetched unconditionally (clamped address, zero weight when outside):
    // 52 independent gathers in flight instead of four dependent groups
    float2 f[4]; float wgt[4];
    // r04: the four corners are neighbours at full resolution, so on the 1/8 and 1/32 levels they almost always (77 % / 94 %) read
    // the SAME four source pixels with different weights: those taps are then loaded once (4 instead of 16 gathers per level;
    // the kernel sits at ~55 % of the texture-address rate; 20 -> 18 us per frame).  Same taps, same weights, same mix per corner;
    // -DAL_SAMPLE_SHARED=0 builds the per-corner loads only (the two builds agree to 1.5e-7 on the unit descriptors, i.e. to the
    // contractions the compiler picks around the mix, and share keypoints and scores bit for bit).
    unsigned pixq[4]; UpTap t2[4], t3[4], t4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int yy = y0 + (q >> 1), xx = x0 + (q & 1);
        const bool inside = yy >= 0 && yy < h && xx >= 0 && xx < w;
        wgt[q] = inside ? ((q & 1) ? wx1 : wx0) * ((q >> 1) ? wy1 : wy0) : 0.0f;
        const int yp = min(max(yy, 0), h - 1) + pt, xp = min(max(xx, 0), w - 1) + pl;
        pixq[q] = (unsigned)(yp * P.Wp + xp);
        t2[q] = up_tap(yp, xp, P.Hp, P.Wp, 2, P.sy2, P.sx2);
        t3[q] = up_tap(yp, xp, P.Hp, P.Wp, 8, P.sy8, P.sx8);
        t4[q] = up_tap(yp, xp, P.Hp, P.Wp, 32, P.sy32, P.sx32);
    }
#ifndef AL_SAMPLE_SHARED
#define AL_SAMPLE_SHARED 1
#endif
    auto same_taps = [](const UpTap (&t)[4]) {
        bool s_ = true;
#pragma unroll
        for (int q = 1; q < 4; ++q) s_ = s_ && t[q].o00 == t[0].o00 && t[q].o01 == t[0].o01 && t[q].o10 == t[0].o10 && t[q].o11 == t[0].o11;
        return s_;
    };
    const unsigned cbo = 4u * (lane & 31);
    auto level_shared = [&](const float* __restrict__ p, const UpTap (&t)[4], float (&o)[4]) {
        if (AL_SAMPLE_SHARED && same_taps(t)) {
            const float v00 = at_b(p, (unsigned)t[0].o00 * 128u + cbo), v01 = at_b(p, (unsigned)t[0].o01 * 128u + cbo);
            const float v10 = at_b(p, (unsigned)t[0].o10 * 128u + cbo), v11 = at_b(p, (unsigned)t[0].o11 * 128u + cbo);
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = up_mix(t[q], v00, v01, v10, v11);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = up_eval_cl32(p, t[q], cbo);
        }
    };
    float aq[4], bq[4];
    if (lane < 32) {
#pragma unroll
        for (int q = 0; q < 4; ++q) aq[q] = at_b(P.g1cl, pixq[q] * 128u + cbo);
        level_shared(P.g3cl, t3, bq);
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) aq[q] = up_eval_cl32(P.g2cl, t2[q], cbo);
        level_shared(P.g4cl, t4, bq);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { const float r = rnorm[pixq[q]]; f[q] = make_float2(aq[q] * r, bq[q] * r); }
    float ax = 0.0f, bx = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) { ax = fmaf(f[q].x, wgt[q], ax); bx = fmaf(f[q].y, wgt[q], bx); }
    const int c = lane & 31;
    const int ca = lane < 32 ? c : 32 + c, cb = lane < 32 ? 64 + c : 96 + c;
    // r04: the sampled features feed the split-precision GEMM (gemm_f16x3.hpp) - written as its (hi, lo) fp16 planes
    // (same bytes as the fp32 row they replace)
    unsigned h2, l2; float amax = 0.0f;
    sslam::split2_fast(ax, bx, h2, l2, amax);
    al_range_note(amax, ctrl);
    const _Float16 hv[2] = {__builtin_bit_cast(sslam::half2v, h2)[0], __builtin_bit_cast(sslam::half2v, h2)[1]};
    const _Float16 lv[2] = {__builtin_bit_cast(sslam::half2v, l2)[0], __builtin_bit_cast(sslam::half2v, l2)[1]};
    sampled[(size_t)gw * 128 + ca] = hv[0]; sampled[(size_t)gw * 128 + cb] = hv[1];
    sampled[lo_off + (size_t)gw * 128 + ca] = lv[0]; sampled[lo_off + (size_t)gw * 128 + cb] = lv[1];
}

// generic row GEMM for the descriptor head: C[M][N] = act(A[M][K] W[N][K]^T + bias)
template <int BM, int BN, int TM, int TN>
__global__ __launch_bounds__(256) void al_gemm_kernel(const float* __restrict__ A, int K, const float* __restrict__ Wt,
                                                      const float* __restrict__ bias, int N, float* __restrict__ C,
                                                      int rows_per_kp, int row_cap, int do_selu,
                                                      const ALCtrl* __restrict__ ctrl, int KS, size_t fs) {
    __shared__ GemmSmem<BM, BN> sm;
    const int zs = blockIdx.z % KS, fr = blockIdx.z / KS;              // grid z = frame * KS + k slice
    A = fsh(A, fr, fs); C = fsh(C, fr, fs); ctrl = fsh(ctrl, fr, fs);
    const int M = ctrl->n_kp * rows_per_kp;
    const int row0 = blockIdx.y * BM, col0 = blockIdx.x * BN;
    if (row0 >= M) return;
    // split-K over blockIdx.z: these GEMMs have few row blocks (M <= 2048) and a long K; slice z
    // writes its partial product to C + z * row_cap * N, the consumer adds the slices in order
    const int kper = K / KS, koff = zs * kper;
    C += (size_t)zs * row_cap * N;
    GemmA ga{A + koff, K, A + koff, K, kper};   // (A1 unused; a null A1 trips an InstCombine crash in ROCm 7.2)
    f32x16 acc[TM][TN];
    gemm_mainloop<BM, BN, TM, TN>(ga, Wt + koff, K, kper, row0, row_cap, col0, N, sm, acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = col0 + wn * 32 * TN + j * 32 + (lane & 31);
            const float b = (bias && col < N) ? bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wm * 32 * TM + i * 32 + acc_row(r, lane);
                float v = acc[i][j][r] + b;
                if (do_selu) v = selu(v);
                if (row < M && col < N) C[(size_t)row * N + col] = v;
            }
        }
}

// r04: the two large GEMMs of the descriptor head (sf_conv 128 -> 128 over 16 sample rows per keypoint, and the
// aggregation, K = 2048) on the split-precision matrix path (gemm_f16x3.hpp: operands as fp16 (hi, lo) planes, three
// v_mfma_f32_32x32x16_f16 per product into two fp32 accumulators, ~2^-22 relative per product) instead of the exact-fp32
// MFMA (1/16 of the f16 rate): their fp32 main loops ran at 46 % of a 157 TFLOP/s ceiling.  A planes row-major
// [rows][K] (lo plane `a_lo` halves behind hi), W planes [N][K] split once at create time.  SPLIT_OUT: SELU, then the
// output goes out as the NEXT GEMM's A planes ([rows][N] == [keypoints][16 N]); else fp32 slabs per k slice.
template <int BM, int BN, int TM, int TN, bool SPLIT_OUT>
__global__ __launch_bounds__(256) void al_gemm_h_kernel(const _Float16* __restrict__ A, size_t a_lo, int K,
                                                        const _Float16* __restrict__ W, size_t w_lo, int N,
                                                        float* __restrict__ C, _Float16* __restrict__ Ch, size_t c_lo,
                                                        int rows_per_kp, int row_cap, ALCtrl* ctrl, int KS,
                                                        size_t fs) {
    __shared__ sslam::GemmSmemH<BM, BN> sm;
    const int zs = blockIdx.z % KS, fr = blockIdx.z / KS;
    A = fsh(A, fr, fs); C = fsh(C, fr, fs); Ch = fsh(Ch, fr, fs); ctrl = fsh(ctrl, fr, fs);
    const int M = ctrl->n_kp * rows_per_kp;
    const int row0 = blockIdx.y * BM, col0 = blockIdx.x * BN;
    if (row0 >= M) return;
    const int kper = K / KS, koff = zs * kper;
    sslam::GemmAH ga{{A + koff, A + a_lo + koff}, {A + koff, A + a_lo + koff}, K, kper};
    f32x16 c1[TM][TN], c2[TM][TN];
    sslam::gemm_mainloop_h<BM, BN, TM, TN>(ga, sslam::SplitPtr{W + koff, W + w_lo + koff}, K, kper, row0, row_cap, col0, N, sm, c1, c2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    if constexpr (!SPLIT_OUT) C += (size_t)zs * row_cap * N;
    float amax = 0.0f;                                        // (SPLIT_OUT: al_range_note at the end)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = col0 + wn * 32 * TN + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wm * 32 * TM + i * 32 + acc_row(r, lane);
                float v = c1[i][j][r] + c2[i][j][r] * sslam::SPLIT_INV;
                if (row >= M || col >= N) continue;
                if constexpr (SPLIT_OUT) {
                    v = selu(v);
                    const float aa = fabsf(v);
                    amax = fmaxf(amax, aa);
                    const _Float16 hi = aa < 6.103515625e-5f ? (_Float16)0.0f : (_Float16)v;      // (as sslam::split_f32)
                    Ch[(size_t)row * N + col] = hi;
                    Ch[c_lo + (size_t)row * N + col] = (_Float16)((v - (float)hi) * sslam::SPLIT_SCALE);
                } else {
                    C[(size_t)row * N + col] = v;
                }
            }
        }
    if constexpr (SPLIT_OUT) al_range_note(amax, ctrl);
}

// fp32 [n] -> (hi, lo) planes (weights of the split GEMMs, once at create time)
__global__ void al_split_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, size_t n, int* range_flag) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    _Float16 hi, lo;
    al_split_weight(src[i], hi, lo, range_flag);
    dst[i] = hi;
    dst[n + i] = lo;
}

// L2 normalise (F.normalize), the reference's second normalisation (features_utils.py:100),
// and keypoints back to input-image pixels; one wave per keypoint
__global__ __launch_bounds__(256) void al_finalize_kernel(const float* __restrict__ raw /*[KSPLIT][cap][128]*/, int cap,
                                                          const float* __restrict__ kp_norm,
                                                          const float* __restrict__ kp_score, int h, int w, float scale_x,
                                                          float scale_y, FrameOut outs, const ALCtrl* __restrict__ ctrl,
                                                          int* range_sticky, size_t fs) {
    raw = fsh(raw, blockIdx.y, fs); kp_norm = fsh(kp_norm, blockIdx.y, fs); kp_score = fsh(kp_score, blockIdx.y, fs);
    ctrl = fsh(ctrl, blockIdx.y, fs);
    float* __restrict__ xy_out = outs.xy[blockIdx.y]; float* __restrict__ desc_out = outs.desc[blockIdx.y];
    float* __restrict__ score_out = outs.score[blockIdx.y]; int32_t* __restrict__ n_out = outs.n[blockIdx.y];
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // a raised range flag voids the frame: count -1 travels with the result (and the instance's sticky word is set for
        // callers that never look at the count); the keypoints below are still written, nobody may use them
        n_out[0] = ctrl->range_overflow ? -1 : ctrl->n_kp;
        if (ctrl->range_overflow) *range_sticky = 1;
    }
    if (n >= ctrl->n_kp) return;
    float a = 0.0f, b = 0.0f;
    for (int z = 0; z < SDDH_KSPLIT; ++z) {
        a += raw[((size_t)z * cap + n) * 128 + lane];
        b += raw[((size_t)z * cap + n) * 128 + 64 + lane];
    }
    float s = a * a + b * b;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
    const float a1 = a * inv, b1 = b * inv;
    float s2 = a1 * a1 + b1 * b1;
    for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
    const float den = sqrtf(s2) + 1e-8f;
    desc_out[(size_t)n * 128 + lane] = a1 / den;
    desc_out[(size_t)n * 128 + 64 + lane] = b1 / den;
    if (lane == 0) {
        // keypoints = wh * (kp + 1) / 2 ; then (kp + 0.5) / scales - 0.5
        const float px = (float)(w - 1) * (kp_norm[2 * n] + 1.0f) / 2.0f;
        const float py = (float)(h - 1) * (kp_norm[2 * n + 1] + 1.0f) / 2.0f;
        xy_out[2 * n] = (px + 0.5f) / scale_x - 0.5f;
        xy_out[2 * n + 1] = (py + 0.5f) / scale_y - 0.5f;
        if (score_out) score_out[n] = kp_score[n];
    }
}

}  // namespace

// ======================================================================== //
//  host side
// ======================================================================== //
struct ALConvW { const float *w, *a, *b; };
struct ALDcnW { const float *ow, *ob, *w, *a, *b; };

struct sslam_aliked {
    sslam_ctx* ctx = nullptr;
    int max_h = 0, max_w = 0, max_kpts = 0;
    int max_frames = 1;               // frames per batched launch sequence (workspace blocks)
    size_t fs = 0;                    // bytes between the workspace blocks of consecutive frames
    int Hp_cap = 0, Wp_cap = 0;
    sslam::Arena arena;
    float* blob = nullptr;
    // weights
    ALConvW b1c1, b1c2, b2c1, b2c2;
    const float *b2dw, *b2db;
    ALDcnW b3c1, b3c2, b4c1, b4c2;
    const float *b3dw, *b3db, *b4dw, *b4db;
    _Float16 *b3c1of, *b3c2of, *b4c1of, *b4c2of;                          // the same, split, fragment order (al_offc_wfrag_kernel)
    _Float16 *b3c1f, *b3c2f, *b4c1f, *b4c2f;                              // deformable-conv weights (BN scale folded, + 1 x 1 branch), split, fragment order
    const float *gw1, *gw2, *gw3, *gw4;
    const float *sh0, *sh2, *sh4, *sh6;
    const float *d_ow, *d_ob, *d_w2, *d_b2, *d_sf, *d_agg;
    // workspace
    ALCtrl* ctrl;
    uint8_t* in_u8;
    float *fsrc, *img, *x1, *x2, *p3, *off, *t3, *x3, *p4, *t4, *x4, *g2, *g3, *g4;
    float *s8, *rnorm, *score, *nms, *bsum, *gk, *g1cl, *g2cl, *g3cl, *g4cl, *pre2, *pre3, *pre4;
    unsigned long long* cand;
    unsigned* hist;
    int cand_cap;
    int* kp_index;
    int* range_sticky;                 // device word (shared by the frames of a batch): a frame of some call raised its range flag since the last read
    unsigned long long* sel_keys;      // the selected keys, unordered (al_select -> al_refine)
    float *kp_norm, *kp_score, *patch, *h32, *pos, *sampled, *feats, *raw;
    _Float16 *d_sf_s, *d_agg_s;          // split (hi | lo) copies of the two large descriptor-head weight matrices
    _Float16* b2c2f;                     // block2.conv2 weights, split, fragment order (al_conv32_wfrag_kernel)
    _Float16* b1c2f;                     // block1.conv2 weights, split, fragment order (al_conv16_wfrag_kernel)
    _Float16* b2c1f;                     // block2.conv1 + downsample weights, split, fragment order (al_conv32p_wfrag_kernel)
    float *p3cl, *t3cl, *p4cl, *t4cl;    // channel-last copies of the deformable layers' inputs (al_dcn_col)
    float *out_xy, *out_desc, *out_score;
    int32_t* out_n;
    Dims last{};
    bool use_graphs = false;          // replay the per-call launch sequence as a cached hipGraph
    sslam::GraphCache graphs;
};

namespace {

size_t al_pad64(size_t n) { return (n + 63) / 64 * 64; }

int al_bind_weights(sslam_aliked* g, size_t n_floats) {
    size_t off = 0;
    auto take = [&](size_t n) { const float* p = g->blob + off; off += al_pad64(n); return p; };
    auto conv = [&](int ci, int co) { ALConvW c; c.w = take((size_t)ci * 9 * co); c.a = take(co); c.b = take(co); return c; };
    auto dcn = [&](int ci, int co) {
        ALDcnW c; c.ow = take((size_t)ci * 9 * 18); c.ob = take(18); c.w = take((size_t)ci * 9 * co);
        c.a = take(co); c.b = take(co); return c;
    };
    g->b1c1 = conv(3, 16); g->b1c2 = conv(16, 16);
    g->b2c1 = conv(16, 32); g->b2c2 = conv(32, 32); g->b2dw = take(16 * 32); g->b2db = take(32);
    g->b3c1 = dcn(32, 64); g->b3c2 = dcn(64, 64); g->b3dw = take(32 * 64); g->b3db = take(64);
    g->b4c1 = dcn(64, 128); g->b4c2 = dcn(128, 128); g->b4dw = take(64 * 128); g->b4db = take(128);
    g->gw1 = take(16 * 32); g->gw2 = take(32 * 32); g->gw3 = take(64 * 32); g->gw4 = take(128 * 32);
    g->sh0 = take(128 * 8); g->sh2 = take(8 * 9 * 4); g->sh4 = take(4 * 9 * 4); g->sh6 = take(4 * 9);
    g->d_ow = take(32 * 1152); g->d_ob = take(32); g->d_w2 = take(32 * 32); g->d_b2 = take(32);
    g->d_sf = take(128 * 128); g->d_agg = take((size_t)128 * 2048);
    SSLAM_REQUIRE(off == n_floats, "sslam_aliked_create: weight blob has %zu floats, expected %zu", n_floats, off);
    return 0;
}

struct ResizePlan { int h, w, blur, ky, kx; float sy, sx; };

ResizePlan resize_plan(int H, int W, int resize) {
    // kornia resize(side='long'): the long side becomes `resize`, the other int(resize / aspect)
    ResizePlan p{};
    const double ar = (double)W / (double)H;
    if (ar > 1.0) { p.h = (int)(resize / ar); p.w = resize; }
    else { p.h = resize; p.w = (int)(resize * ar); }
    const double fy = (double)H / p.h, fx = (double)W / p.w;
    p.blur = (fy > fx ? fy : fx) > 1.0;
    p.sy = (float)((fy - 1.0) / 2.0 > 0.001 ? (fy - 1.0) / 2.0 : 0.001);
    p.sx = (float)((fx - 1.0) / 2.0 > 0.001 ? (fx - 1.0) / 2.0 : 0.001);
    const double ksy = 2.0 * 2 * ((fy - 1.0) / 2.0 > 0.001 ? (fy - 1.0) / 2.0 : 0.001);
    const double ksx = 2.0 * 2 * ((fx - 1.0) / 2.0 > 0.001 ? (fx - 1.0) / 2.0 : 0.001);
    p.ky = (int)(ksy > 3 ? ksy : 3); p.kx = (int)(ksx > 3 ? ksx : 3);
    if (p.ky % 2 == 0) ++p.ky;
    if (p.kx % 2 == 0) ++p.kx;
    return p;
}

// network-size bookkeeping of one image (resize to 1024 on the long side, centred pad to a multiple of 32)
Dims al_dims(int H, int W, int C) {
    const ResizePlan rp = resize_plan(H, W, 1024);
    Dims d{};
    d.H = H; d.W = W; d.C = C; d.h = rp.h; d.w = rp.w;
    const int pad_h = (((d.h / 32) + 1) * 32 - d.h) % 32, pad_w = (((d.w / 32) + 1) * 32 - d.w) % 32;
    d.Hp = d.h + pad_h; d.Wp = d.w + pad_w; d.pl = pad_w / 2; d.pt = pad_h / 2;
    return d;
}

// (host-side state such as g->last is set by the ENTRY POINTS, not here: a graph replay skips this function)
// One launch sequence for F frames of one size: every grid carries the frame in its last used dimension and every
// per-frame pointer is frame 0's plus f * g->fs (see fsh above).  F = 1 is the single-frame entry.
int al_enqueue(sslam_aliked* g, int F, const FrameIn& srcs, int H, int W, int C, int n_limit, const FrameOut& outs) {
    hipStream_t s = g->ctx->stream;
    (void)hipGetLastError();     // (a stale error of another library on this thread - e.g. RCCL's probes - is not ours)
    const ResizePlan rp = resize_plan(H, W, 1024);
    const Dims d = al_dims(H, W, C);
    SSLAM_REQUIRE(d.Hp <= g->Hp_cap && d.Wp <= g->Wp_cap && d.h >= 8 && d.w >= 8,
                  "sslam_aliked: network size %dx%d outside the instance capacity %dx%d", d.Hp, d.Wp,
                  g->Hp_cap, g->Wp_cap);
    SSLAM_REQUIRE(F >= 1 && F <= g->max_frames, "sslam_aliked: %d frames, instance capacity %d", F, g->max_frames);
    const int Hp = d.Hp, Wp = d.Wp;
    const unsigned uF = (unsigned)F;
    const size_t fs = g->fs;
    SSLAM_REQUIRE(rp.kx <= 31 && rp.ky <= 31, "sslam_aliked: blur kernel too large (%d,%d)", rp.kx, rp.ky);
    hipLaunchKernelGGL(al_reset_kernel, dim3(uF), dim3(256), 0, s, g->ctrl, g->hist, fs, g->gk, rp.kx, rp.sx, rp.ky, rp.sy);

    hipLaunchKernelGGL(al_to_float_kernel, dim3(sslam::cdiv(W, 256), H, uF), dim3(256), 0, s, srcs, g->fsrc, d,
                       g->gk, rp.kx, rp.blur, fs);
    hipLaunchKernelGGL(al_resize_pad_kernel, dim3(sslam::cdiv(Wp, 256), Hp, uF), dim3(256), 0, s, g->fsrc, g->img, d,
                       g->gk + 32, rp.ky, rp.blur, fs);
    // block1 (conv1 + conv2 fused): ~3 waves per SIMD when the batch allows; at least four rows per wave (two extra conv1 rows per block:
    // short blocks only when one or two frames have to fill the chip)
    {
        const int strips = sslam::cdiv(Wp, B1_SW);
        const int nblk = std::max(1, std::min(sslam::cdiv(Hp, 4), 3072 / std::max(1, strips * F)));
        const int hs1 = sslam::cdiv(Hp, nblk);
        hipLaunchKernelGGL(al_block1_rows_kernel, dim3(strips, sslam::cdiv(Hp, hs1), uF), dim3(64), 0, s, g->img, g->x1, Hp, Wp, hs1,
                           g->b1c1.w, g->b1c1.a, g->b1c1.b, g->b1c2f, g->b1c2.a, g->b1c2.b, g->ctrl, fs);
    }
    // block2 at 1/2 (pooling + conv1 + 1 x 1 branch + conv2 + residual fused): one wave per SIMD when the batch allows; at least
    // three rows per wave (two extra t2 rows per block).  (A frame's values do not depend on these splits: no sum is re-associated.)
    const int H2 = Hp / 2, W2 = Wp / 2;
    {
        const int strips = sslam::cdiv(W2, B2_SW);
        const int nblk = std::max(1, std::min(sslam::cdiv(H2, 3), 1024 / std::max(1, strips * F)));
        const int hs2 = sslam::cdiv(H2, nblk), nb = sslam::cdiv(H2, hs2), n_waves = strips * nb * F;
        hipLaunchKernelGGL(al_block2_rows_kernel, dim3(sslam::cdiv(n_waves, 4)), dim3(256), B2_LDS, s, g->x1, g->x2, H2, W2, hs2, nb, strips, n_waves,
                           g->b2c1f, g->b2c2f, g->b2c1.a, g->b2c1.b, g->b2db, g->b2c2.a, g->b2c2.b, g->ctrl, fs);
    }
#ifndef AL_DCN4_CS
#define AL_DCN4_CS 4      // workgroups per pixel tile of the 1/32 deformable layers (output channels split: 10 tiles per frame there)
#endif
#ifndef AL_DCN3_CS
#define AL_DCN3_CS 1
#endif
    // block3 at 1/8 (deformable): per layer the offset conv, then sampling + contraction + BN (+ 1 x 1 residual branch) + SELU fused
    const int H3 = Hp / 8, W3 = Wp / 8, HW3 = H3 * W3;
    const dim3 g3(sslam::cdiv(W3, 32), H3, uF);
    hipLaunchKernelGGL(al_avgpool_kernel, dim3(sslam::cdiv(32 * HW3, 256), uF), dim3(256), 0, s, g->x2, g->p3, 32, H2, W2, 4, fs, g->p3cl);
    const float mo3 = (float)(H3 > W3 ? H3 : W3) / 4.0f;
    hipLaunchKernelGGL((al_offset_conv_h_kernel<32>), g3, dim3(256), 0, s, g->p3cl, g->off, H3, W3, g->b3c1of, g->b3c1.ob, mo3, g->ctrl, fs);
    hipLaunchKernelGGL((al_dcn_h_kernel<32, 64, 0, AL_DCN3_CS>), dim3(AL_DCN3_CS * sslam::cdiv(W3, 32), H3, uF), dim3(256), 0, s, g->p3cl, g->off, nullptr, H3, W3, g->b3c1f, g->b3c1.b, nullptr, g->t3, g->t3cl, g->ctrl, fs);
    hipLaunchKernelGGL((al_offset_conv_h_kernel<64>), g3, dim3(256), 0, s, g->t3cl, g->off, H3, W3, g->b3c2of, g->b3c2.ob, mo3, g->ctrl, fs);
    hipLaunchKernelGGL((al_dcn_h_kernel<64, 64, 32, AL_DCN3_CS>), dim3(AL_DCN3_CS * sslam::cdiv(W3, 32), H3, uF), dim3(256), 0, s, g->t3cl, g->off, g->p3cl, H3, W3, g->b3c2f, g->b3c2.b, g->b3db, g->x3, nullptr, g->ctrl, fs);
    // block4 at 1/32
    const int H4 = Hp / 32, W4 = Wp / 32, HW4 = H4 * W4;
    const dim3 g4(sslam::cdiv(W4, 32), H4, uF);
    sslam::graph_cut(s);       // (pieces of 10 - 12 launches: a graph's replay stops for ~25 us after its 15th kernel node, common.hpp)
    hipLaunchKernelGGL(al_avgpool_kernel, dim3(sslam::cdiv(64 * HW4, 256), uF), dim3(256), 0, s, g->x3, g->p4, 64, H3, W3, 4, fs, g->p4cl);
    const float mo4 = (float)(H4 > W4 ? H4 : W4) / 4.0f;
    hipLaunchKernelGGL((al_offset_conv_h_kernel<64>), g4, dim3(256), 0, s, g->p4cl, g->off, H4, W4, g->b4c1of, g->b4c1.ob, mo4, g->ctrl, fs);
    const dim3 g4s(AL_DCN4_CS * sslam::cdiv(W4, 32), H4, uF);  // AL_DCN4_CS workgroups per tile, 128 / AL_DCN4_CS output channels each
    hipLaunchKernelGGL((al_dcn_h_kernel<64, 128, 0, AL_DCN4_CS>), g4s, dim3(256), 0, s, g->p4cl, g->off, nullptr, H4, W4, g->b4c1f, g->b4c1.b, nullptr, g->t4, g->t4cl, g->ctrl, fs);
    hipLaunchKernelGGL((al_offset_conv_h_kernel<128>), g4, dim3(256), 0, s, g->t4cl, g->off, H4, W4, g->b4c2of, g->b4c2.ob, mo4, g->ctrl, fs);
    hipLaunchKernelGGL((al_dcn_h_kernel<128, 128, 64, AL_DCN4_CS>), g4s, dim3(256), 0, s, g->t4cl, g->off, g->p4cl, H4, W4, g->b4c2f, g->b4c2.b, g->b4db, g->x4, nullptr, g->ctrl, fs);
    // gates
    hipLaunchKernelGGL(al_gate_kernel<32>, dim3(sslam::cdiv(H2 * W2, 256), uF), dim3(256), 0, s, g->x2, g->g2, H2 * W2, g->gw2, g->g2cl, fs);
    {
        const int ba = sslam::cdiv(32 * HW3, 256), bb = sslam::cdiv(32 * HW4, 256);
        hipLaunchKernelGGL(al_gate_small_kernel, dim3(ba + bb, uF), dim3(256), 0, s, GateSmall{g->x3, g->g3, 64, HW3, g->gw3, g->g3cl},
                           GateSmall{g->x4, g->g4, 128, HW4, g->gw4, g->g4cl}, ba, fs);
    }
    // aggregation + score head
    Pyr P{g->x1, g->g2, g->g3, g->g4, g->gw1, Hp, Wp, g->g1cl};
    {
        auto step = [](int full, int S) { return (float)(full / S - 1) / (float)(full - 1); };
        P.sy2 = step(Hp, 2); P.sx2 = step(Wp, 2); P.sy8 = step(Hp, 8); P.sx8 = step(Wp, 8);
        P.sy32 = step(Hp, 32); P.sx32 = step(Wp, 32);
        P.g2cl = g->g2cl; P.g3cl = g->g3cl; P.g4cl = g->g4cl;
        P.pre2 = g->pre2; P.pre3 = g->pre3; P.pre4 = g->pre4;
    }
    hipLaunchKernelGGL(al_agg_pre_kernel, dim3(sslam::cdiv(H2 * W2 + HW3 + HW4, 256), uF), dim3(256), 0, s, g->g2, g->g3, g->g4,
                       g->sh0, g->pre2, g->pre3, g->pre4, Hp, Wp, fs);
    hipLaunchKernelGGL(al_aggregate_kernel, dim3(sslam::cdiv(Wp, 256), Hp, uF), dim3(256), 0, s, P, g->sh0, g->s8, g->rnorm, fs);
    hipLaunchKernelGGL(al_score_tail_kernel, dim3(sslam::cdiv(Wp, ST_W), sslam::cdiv(Hp, ST_H), uF), dim3(256), 0, s, g->s8,
                       Hp, Wp, g->sh2, g->sh4, g->sh6, g->score, d.h, d.w, d.pl, d.pt, fs);
    // DKD
    const int nbx = sslam::cdiv(d.w, NW_OW), nby = sslam::cdiv(d.h, NW_OH), npx = d.h * d.w;
    hipLaunchKernelGGL(al_nms_wave_kernel, dim3(sslam::cdiv(nbx * nby, 4), uF), dim3(256), 0, s, g->score, d.h, d.w, nbx, nbx * nby,
                       g->nms, g->bsum, 0.2f, g->cand, g->cand_cap, g->ctrl, g->hist, fs);
    sslam::graph_cut(s);
    hipLaunchKernelGGL(al_collect_kernel, dim3(sslam::cdiv(npx, 256 * COLLECT_PPT), uF), dim3(256), 0, s, g->nms, npx, 0.0f, 1, g->bsum,
                       nbx * nby, g->cand, g->cand_cap, g->ctrl, g->hist, fs);
    hipLaunchKernelGGL(al_select_kernel, dim3(uF), dim3(1024), (SEL_CAP + EDGE_CAP) * 8, s, g->cand, g->cand_cap, n_limit,
                       g->sel_keys, g->ctrl, g->hist, fs);
    const int NK = g->max_kpts;
    hipLaunchKernelGGL(al_refine_kernel, dim3(sslam::cdiv(NK, REFINE_KPB), uF), dim3(256), 0, s, g->score, d.h, d.w, g->sel_keys,
                       g->kp_index, g->kp_norm, g->kp_score, g->ctrl, fs);
    // SDDH
    hipLaunchKernelGGL(al_patch_kernel, dim3(sslam::cdiv(NK * 3, 4), uF), dim3(256), 0, s, P, g->rnorm, d.pl, d.pt, d.h, d.w,
                       g->kp_norm, g->patch, g->ctrl, fs);
    hipLaunchKernelGGL((al_gemm_kernel<64, 64, 1, 1>), dim3(1, sslam::cdiv(NK, 64), SDDH_KSPLIT * uF), dim3(256), 0, s, g->patch,
                       1152, g->d_ow, nullptr, 32, g->h32, 1, NK, 0, g->ctrl, SDDH_KSPLIT, fs);
    const float mo = (float)(d.h > d.w ? d.h : d.w) / 4.0f;
    hipLaunchKernelGGL(al_offsets_kernel, dim3(sslam::cdiv(NK * 32, 256), uF), dim3(256), 0, s, g->h32, NK, g->d_ob, g->d_w2, g->d_b2,
                       g->kp_norm, d.h, d.w, mo, g->pos, g->ctrl, fs);
    // (r04: `sampled` and `feats` hold the (hi, lo) fp16 planes of [rows][128] - the same bytes as the fp32 rows they replaced)
    const size_t plane = (size_t)(NK * 16 + 64) * 128;
    _Float16* sampled_h = reinterpret_cast<_Float16*>(g->sampled);
    _Float16* feats_h = reinterpret_cast<_Float16*>(g->feats);
    hipLaunchKernelGGL(al_sample_kernel, dim3(sslam::cdiv(NK * 16, 4), uF), dim3(256), 0, s, P, g->rnorm, d.pl, d.pt, d.h,
                       d.w, g->pos, sampled_h, plane, g->ctrl, fs);
    hipLaunchKernelGGL((al_gemm_h_kernel<64, 128, 1, 2, true>), dim3(1, sslam::cdiv(NK * 16, 64), uF), dim3(256), 0, s, sampled_h, plane,
                       128, g->d_sf_s, (size_t)128 * 128, 128, nullptr, feats_h, plane, 16, NK * 16, g->ctrl, 1, fs);
    // (batches: 64 x 128 tiles - the 16.8 MB of sampled-feature planes are read once; with two column blocks the PMC counters showed
    //  35.7 MB fetched per frame.  One or two frames: 64 x 64 tiles, twice the workgroups.  Same k-split, same order: bit-identical)
    if (F >= 4)
        hipLaunchKernelGGL((al_gemm_h_kernel<64, 128, 1, 2, false>), dim3(1, sslam::cdiv(NK, 64), SDDH_KSPLIT * uF), dim3(256), 0, s, feats_h,
                           plane, 2048, g->d_agg_s, (size_t)128 * 2048, 128, g->raw, nullptr, 0, 1, NK, g->ctrl, SDDH_KSPLIT, fs);
    else
        hipLaunchKernelGGL((al_gemm_h_kernel<64, 64, 1, 1, false>), dim3(2, sslam::cdiv(NK, 64), SDDH_KSPLIT * uF), dim3(256), 0, s, feats_h,
                           plane, 2048, g->d_agg_s, (size_t)128 * 2048, 128, g->raw, nullptr, 0, 1, NK, g->ctrl, SDDH_KSPLIT, fs);
    const float scale_x = (float)d.w / (float)W, scale_y = (float)d.h / (float)H;
    hipLaunchKernelGGL(al_finalize_kernel, dim3(sslam::cdiv(NK, 4), uF), dim3(256), 0, s, g->raw, NK, g->kp_norm, g->kp_score,
                       d.h, d.w, scale_x, scale_y, outs, g->ctrl, g->range_sticky, fs);
    SSLAM_HIP_CHECK(hipGetLastError());
    return 0;
}

// single-frame form
int al_enqueue(sslam_aliked* g, const uint8_t* img_dev, int H, int W, int C, int n_limit, float* xy_out,
               float* desc_out, float* score_out, int32_t* n_out) {
    FrameIn in{}; FrameOut out{};
    in.img[0] = img_dev; out.xy[0] = xy_out; out.desc[0] = desc_out; out.score[0] = score_out; out.n[0] = n_out;
    return al_enqueue(g, 1, in, H, W, C, n_limit, out);
}

}  // namespace

extern "C" {

int sslam_aliked_create_batched(sslam_ctx* ctx, const float* weights, size_t n_floats, int max_h, int max_w,
                                int max_kpts, int max_frames, sslam_aliked** out) {
    SSLAM_REQUIRE(ctx && weights && out, "sslam_aliked_create: NULL argument");
    SSLAM_REQUIRE(max_frames >= 1 && max_frames <= MAX_FRAMES, "sslam_aliked_create: max_frames %d not in [1, %d]",
                  max_frames, MAX_FRAMES);
    SSLAM_REQUIRE(max_h >= 16 && max_w >= 16 && max_h <= 8192 && max_w <= 8192,
                  "sslam_aliked_create: image size %dx%d unsupported", max_w, max_h);
    SSLAM_REQUIRE(max_kpts >= 1 && max_kpts <= SEL_CAP, "sslam_aliked_create: max_kpts %d not in [1, %d]", max_kpts,
                  SEL_CAP);
    SSLAM_HIP_CHECK(hipSetDevice(ctx->device));
    sslam_aliked* g = new sslam_aliked();
    g->ctx = ctx; g->max_h = max_h; g->max_w = max_w; g->max_kpts = max_kpts; g->max_frames = max_frames;
    // the network never runs larger than 1024 on the long side (+ padding to /32)
    g->Hp_cap = 1024 + 32; g->Wp_cap = 1024 + 32;
    const size_t HWp = (size_t)g->Hp_cap * g->Wp_cap, HWi = (size_t)max_h * max_w, NK = (size_t)max_kpts;
    g->cand_cap = (int)(1024 * 1024);
    // shared by all frames: weights, their re-ordered copies, the blur taps
    auto carve_shared = [&](sslam::Arena& A) {
        g->blob = A.take<float>(n_floats);
        g->d_sf_s = A.take<_Float16>(2 * 128 * 128); g->d_agg_s = A.take<_Float16>((size_t)2 * 128 * 2048);
        g->b2c2f = A.take<_Float16>(2 * 32 * 288);
        g->b1c2f = A.take<_Float16>(5 * 2 * 64 * 8);
        g->b2c1f = A.take<_Float16>(10 * 2 * 64 * 8);
        g->b3c1f = A.take<_Float16>((size_t)(9 * 2 + 0) * 2 * 2 * 512); g->b3c2f = A.take<_Float16>((size_t)(9 * 4 + 2) * 2 * 2 * 512);
        g->b4c1f = A.take<_Float16>((size_t)(9 * 4 + 0) * 4 * 2 * 512); g->b4c2f = A.take<_Float16>((size_t)(9 * 8 + 4) * 4 * 2 * 512);
        g->b3c1of = A.take<_Float16>(18 * 2 * 512); g->b3c2of = A.take<_Float16>(36 * 2 * 512); g->b4c1of = A.take<_Float16>(36 * 2 * 512); g->b4c2of = A.take<_Float16>(72 * 2 * 512);
        g->gk = A.take<float>(64);
        g->range_sticky = A.take<int>(4);
    };
    // one workspace block per frame of a batch (frame f's copy of a buffer = frame 0's + f * g->fs bytes)
    auto carve_frame = [&](sslam::Arena& A) {
        g->ctrl = A.take<ALCtrl>(1);
        g->in_u8 = A.take<uint8_t>(HWi * 4);
        g->fsrc = A.take<float>(3 * HWi); g->img = A.take<float>(3 * HWp);
        g->x1 = A.take<float>(16 * HWp); g->x2 = A.take<float>(32 * HWp / 4);
        g->p3 = A.take<float>(32 * HWp / 64); g->off = A.take<float>(18 * HWp / 64);
        g->t3 = A.take<float>(64 * HWp / 64); g->x3 = A.take<float>(64 * HWp / 64);
        g->p3cl = A.take<float>(32 * HWp / 64); g->t3cl = A.take<float>(64 * HWp / 64);
        g->p4cl = A.take<float>(64 * HWp / 1024); g->t4cl = A.take<float>(128 * HWp / 1024);
        g->p4 = A.take<float>(64 * HWp / 1024); g->t4 = A.take<float>(128 * HWp / 1024); g->x4 = A.take<float>(128 * HWp / 1024);
        g->g2 = A.take<float>(32 * HWp / 4); g->g3 = A.take<float>(32 * HWp / 64); g->g4 = A.take<float>(32 * HWp / 1024);
        g->g2cl = A.take<float>(32 * HWp / 4); g->g3cl = A.take<float>(32 * HWp / 64); g->g4cl = A.take<float>(32 * HWp / 1024);
        g->s8 = A.take<float>(8 * HWp); g->rnorm = A.take<float>(HWp); g->g1cl = A.take<float>(32 * HWp);
        g->pre2 = A.take<float>(AGG_PRE * HWp / 4); g->pre3 = A.take<float>(AGG_PRE * HWp / 64); g->pre4 = A.take<float>(AGG_PRE * HWp / 1024);
        g->score = A.take<float>(HWp); g->nms = A.take<float>(HWp); g->bsum = A.take<float>(4096);
        g->cand = A.take<unsigned long long>(g->cand_cap); g->hist = A.take<unsigned>(HBINS);
        g->kp_index = A.take<int>(SEL_CAP); g->sel_keys = A.take<unsigned long long>(SEL_CAP);
        g->kp_norm = A.take<float>(2 * NK + 64); g->kp_score = A.take<float>(NK + 64);
        g->patch = A.take<float>((NK + 64) * 1152); g->h32 = A.take<float>(SDDH_KSPLIT * (NK + 64) * 32); g->pos = A.take<float>(NK * 32 + 64);
        g->sampled = A.take<float>((NK * 16 + 64) * 128); g->feats = A.take<float>((NK * 16 + 64) * 128);
        g->raw = A.take<float>(SDDH_KSPLIT * (NK + 64) * 128);
        g->out_xy = A.take<float>(2 * NK); g->out_desc = A.take<float>(NK * 128); g->out_score = A.take<float>(NK);
        g->out_n = A.take<int32_t>(16);
    };
    sslam::Arena probe;
    probe.measure();
    carve_shared(probe);
    const size_t shared_bytes = (probe.off + 4095) / 4096 * 4096;
    probe.off = 0;
    carve_frame(probe);
    g->fs = (probe.off + 4095) / 4096 * 4096;
    if (g->arena.init(shared_bytes + g->fs * (size_t)max_frames + 4096)) { delete g; return 1; }
    carve_shared(g->arena);
    g->arena.off = shared_bytes;               // frame 0's block starts on a page boundary; blocks 1.. follow at g->fs
    carve_frame(g->arena);
    if (g->out_n == nullptr) {                 // (release before reporting: the batched workspace is GBs)
        g->arena.release(); delete g;
        SSLAM_REQUIRE(false, "sslam_aliked_create: workspace arena exhausted");
    }
    SSLAM_HIP_CHECK(hipMemcpy(g->blob, weights, n_floats * 4, hipMemcpyHostToDevice));
    if (int rc = al_bind_weights(g, n_floats)) { g->arena.release(); delete g; return rc; }
    {   // split-precision weight fragments of the matrix-core kernels (once)
        hipStream_t s = ctx->stream;
        SSLAM_HIP_CHECK(hipMemsetAsync(g->range_sticky, 0, 4 * sizeof(int), s));
        hipLaunchKernelGGL(al_conv32_wfrag_kernel, dim3(sslam::cdiv(32 * 288, 256)), dim3(256), 0, s, g->b2c2.w, g->b2c2f, g->range_sticky);
        hipLaunchKernelGGL(al_conv16_wfrag_kernel, dim3(sslam::cdiv(5 * 64 * 8, 256)), dim3(256), 0, s, g->b1c2.w, g->b1c2f, g->range_sticky);
        hipLaunchKernelGGL(al_conv32p_wfrag_kernel, dim3(sslam::cdiv(10 * 64 * 8, 256)), dim3(256), 0, s, g->b2c1.w, g->b2dw, g->b2c1f, g->range_sticky);
        hipLaunchKernelGGL(al_split_kernel, dim3(sslam::cdiv(128 * 128, 256)), dim3(256), 0, s, g->d_sf, g->d_sf_s, (size_t)128 * 128, g->range_sticky);
        hipLaunchKernelGGL(al_split_kernel, dim3(sslam::cdiv(128 * 2048, 256)), dim3(256), 0, s, g->d_agg, g->d_agg_s, (size_t)128 * 2048, g->range_sticky);
        hipLaunchKernelGGL((al_dcn_wfrag_kernel<32, 64, 0>), dim3(sslam::cdiv((9 * 2 + 0) * 2 * 512, 256)), dim3(256), 0, s, g->b3c1.w, g->b3c1.a, nullptr, g->b3c1f, g->range_sticky);
        hipLaunchKernelGGL((al_dcn_wfrag_kernel<64, 64, 32>), dim3(sslam::cdiv((9 * 4 + 2) * 2 * 512, 256)), dim3(256), 0, s, g->b3c2.w, g->b3c2.a, g->b3dw, g->b3c2f, g->range_sticky);
        hipLaunchKernelGGL((al_dcn_wfrag_kernel<64, 128, 0>), dim3(sslam::cdiv((9 * 4 + 0) * 4 * 512, 256)), dim3(256), 0, s, g->b4c1.w, g->b4c1.a, nullptr, g->b4c1f, g->range_sticky);
        hipLaunchKernelGGL((al_dcn_wfrag_kernel<128, 128, 64>), dim3(sslam::cdiv((9 * 8 + 4) * 4 * 512, 256)), dim3(256), 0, s, g->b4c2.w, g->b4c2.a, g->b4dw, g->b4c2f, g->range_sticky);
        hipLaunchKernelGGL(al_offc_wfrag_kernel<32>, dim3(sslam::cdiv(18 * 512, 256)), dim3(256), 0, s, g->b3c1.ow, g->b3c1of, g->range_sticky);
        hipLaunchKernelGGL(al_offc_wfrag_kernel<64>, dim3(sslam::cdiv(36 * 512, 256)), dim3(256), 0, s, g->b3c2.ow, g->b3c2of, g->range_sticky);
        hipLaunchKernelGGL(al_offc_wfrag_kernel<64>, dim3(sslam::cdiv(36 * 512, 256)), dim3(256), 0, s, g->b4c1.ow, g->b4c1of, g->range_sticky);
        hipLaunchKernelGGL(al_offc_wfrag_kernel<128>, dim3(sslam::cdiv(72 * 512, 256)), dim3(256), 0, s, g->b4c2.ow, g->b4c2of, g->range_sticky);
        int wflag = 0;
        SSLAM_HIP_CHECK(hipMemcpyAsync(&wflag, g->range_sticky, sizeof(int), hipMemcpyDeviceToHost, s));
        SSLAM_HIP_CHECK(hipStreamSynchronize(s));
        if (wflag) {
            g->arena.release(); delete g;
            SSLAM_REQUIRE(false, "sslam_aliked_create: a weight (BN scale folded) with |value| >= 65520 does not fit the fp16 planes "
                                 "of the split-precision stages");
        }
    }
    SSLAM_HIP_CHECK(hipFuncSetAttribute((const void*)al_block2_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B2_LDS));
    SSLAM_HIP_CHECK(hipFuncSetAttribute((const void*)al_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (SEL_CAP + EDGE_CAP) * 8));
    sslam::ctx_retain(ctx);
    *out = g;
    return 0;
}

int sslam_aliked_create(sslam_ctx* ctx, const float* weights, size_t n_floats, int max_h, int max_w,
                        int max_kpts, sslam_aliked** out) {
    return sslam_aliked_create_batched(ctx, weights, n_floats, max_h, max_w, max_kpts, 1, out);
}

int sslam_aliked_destroy(sslam_aliked* g) {
    if (!g) return 0;
    (void)hipStreamSynchronize(g->ctx->stream);
    g->graphs.clear();
    g->arena.release();
    sslam_ctx* ctx = g->ctx;
    delete g;
    sslam::ctx_release(ctx);
    return 0;
}

static int al_check_image(sslam_aliked* g, int H, int W, int C, int max_kpts) {
    SSLAM_REQUIRE(C == 1 || C == 3 || C == 4, "sslam_aliked_extract: %d channels (want 1 gray, 3 BGR or 4 BGRA)", C);
    SSLAM_REQUIRE(H >= 16 && W >= 16 && H <= g->max_h && W <= g->max_w,
                  "sslam_aliked_extract: image %dx%d outside the instance capacity %dx%d", W, H, g->max_w, g->max_h);
    SSLAM_REQUIRE(max_kpts >= 1 && max_kpts <= g->max_kpts, "sslam_aliked_extract: max_kpts %d exceeds capacity %d",
                  max_kpts, g->max_kpts);
    return 0;
}

int sslam_aliked_extract_dev(sslam_aliked* g, const uint8_t* img, int H, int W, int C, int max_kpts, float* xy_out,
                             float* desc_out, float* score_out, int32_t* n_out) {
    SSLAM_REQUIRE(g && img && xy_out && desc_out && n_out, "sslam_aliked_extract_dev: NULL argument");
    if (int rc = al_check_image(g, H, W, C, max_kpts)) return rc;
    g->last = al_dims(H, W, C);
    if (!g->use_graphs) return al_enqueue(g, img, H, W, C, max_kpts, xy_out, desc_out, score_out, n_out);
    const std::vector<uint64_t> key{(uint64_t)img, (uint64_t)H, (uint64_t)W, (uint64_t)C, (uint64_t)max_kpts,
                                    (uint64_t)xy_out, (uint64_t)desc_out, (uint64_t)score_out, (uint64_t)n_out};
    return sslam::run_cached(g->graphs, g->ctx->stream, key,
                             [&] { return al_enqueue(g, img, H, W, C, max_kpts, xy_out, desc_out, score_out, n_out); });
}

int sslam_aliked_extract_batch_dev(sslam_aliked* g, int n_frames, const uint8_t* const* imgs, int H, int W, int C,
                                   int max_kpts, float* const* xy_out, float* const* desc_out,
                                   float* const* score_out, int32_t* const* n_out) {
    SSLAM_REQUIRE(g && imgs && xy_out && desc_out && n_out, "sslam_aliked_extract_batch_dev: NULL argument");
    SSLAM_REQUIRE(n_frames >= 1 && n_frames <= g->max_frames, "sslam_aliked_extract_batch_dev: %d frames, instance capacity %d",
                  n_frames, g->max_frames);
    if (int rc = al_check_image(g, H, W, C, max_kpts)) return rc;
    FrameIn in{}; FrameOut out{};
    std::vector<uint64_t> key{(uint64_t)n_frames, (uint64_t)H, (uint64_t)W, (uint64_t)C, (uint64_t)max_kpts};
    for (int f = 0; f < n_frames; ++f) {
        SSLAM_REQUIRE(imgs[f] && xy_out[f] && desc_out[f] && n_out[f], "sslam_aliked_extract_batch_dev: NULL pointer for frame %d", f);
        in.img[f] = imgs[f]; out.xy[f] = xy_out[f]; out.desc[f] = desc_out[f];
        out.score[f] = score_out ? score_out[f] : nullptr; out.n[f] = n_out[f];
        key.push_back((uint64_t)imgs[f]); key.push_back((uint64_t)xy_out[f]); key.push_back((uint64_t)desc_out[f]);
        key.push_back((uint64_t)out.score[f]); key.push_back((uint64_t)n_out[f]);
    }
    g->last = al_dims(H, W, C);
    if (!g->use_graphs) return al_enqueue(g, n_frames, in, H, W, C, max_kpts, out);
    return sslam::run_cached(g->graphs, g->ctx->stream, key, [&] { return al_enqueue(g, n_frames, in, H, W, C, max_kpts, out); });
}

/* Replay the launch sequence of sslam_aliked_extract_dev as a cached hipGraph (one graph per
 * distinct argument tuple, LRU of 128): for callers that cycle through a fixed set of buffers,
 * as the frame pipeline does.  Results are identical; only the host cost of a call changes. */
int sslam_aliked_range_overflow(sslam_aliked* g, int* flag_out) {
    SSLAM_REQUIRE(g && flag_out, "sslam_aliked_range_overflow: NULL argument");
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

int sslam_aliked_use_graphs(sslam_aliked* g, int enable) {
    SSLAM_REQUIRE(g != nullptr, "sslam_aliked_use_graphs: NULL instance");
    if (!enable) { (void)hipStreamSynchronize(g->ctx->stream); g->graphs.clear(); }
    g->use_graphs = enable != 0;
    return 0;
}

int sslam_aliked_extract_host(sslam_aliked* g, const uint8_t* img, int H, int W, int C, int max_kpts, float* xy_out,
                              float* desc_out, float* score_out, int32_t* n_out) {
    SSLAM_REQUIRE(g && img && xy_out && desc_out && n_out, "sslam_aliked_extract_host: NULL argument");
    if (int rc = al_check_image(g, H, W, C, max_kpts)) return rc;
    SSLAM_HIP_CHECK(hipSetDevice(g->ctx->device));
    hipStream_t s = g->ctx->stream;
    SSLAM_HIP_CHECK(hipMemcpyAsync(g->in_u8, img, (size_t)H * W * C, hipMemcpyHostToDevice, s));
    g->last = al_dims(H, W, C);
    if (int rc = al_enqueue(g, g->in_u8, H, W, C, max_kpts, g->out_xy, g->out_desc, g->out_score, g->out_n)) return rc;
    int32_t n = 0;
    SSLAM_HIP_CHECK(hipMemcpyAsync(&n, g->out_n, 4, hipMemcpyDeviceToHost, s));
    SSLAM_HIP_CHECK(hipStreamSynchronize(s));
    if (n == -1) {                           // (al_finalize_kernel: the frame's range flag)
        (void)hipMemsetAsync(g->range_sticky, 0, sizeof(int), s);
        (void)hipStreamSynchronize(s);
        SSLAM_REQUIRE(false, "sslam_aliked_extract_host: an activation left the fp16 range of the split-precision stages "
                             "(|value| >= 65520): the frame's features are void");
    }
    SSLAM_REQUIRE(n >= 0 && n <= max_kpts, "sslam_aliked_extract_host: corrupt keypoint count %d", n);
    if (n) {
        SSLAM_HIP_CHECK(hipMemcpyAsync(xy_out, g->out_xy, (size_t)n * 8, hipMemcpyDeviceToHost, s));
        SSLAM_HIP_CHECK(hipMemcpyAsync(desc_out, g->out_desc, (size_t)n * 512, hipMemcpyDeviceToHost, s));
        if (score_out) SSLAM_HIP_CHECK(hipMemcpyAsync(score_out, g->out_score, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        SSLAM_HIP_CHECK(hipStreamSynchronize(s));
    }
    *n_out = n;
    return 0;
}

/* Test hook.  which: 0 = score map [h][w], 1 = kp_index [n] (int32), 2 = dims {h,w,Hp,Wp,pl,pt,n_cand,n_kp},
 * 3 = x1 [16][Hp][Wp], 4 = x2, 5 = x3, 6 = x4, 7 = img [3][Hp][Wp], 8 = nms [h][w], 9 = kp_norm [n][2] */
int sslam_aliked_debug_read(sslam_aliked* g, int which, void* dst, size_t bytes) {
    SSLAM_REQUIRE(g && dst, "sslam_aliked_debug_read: NULL argument");
    SSLAM_HIP_CHECK(hipStreamSynchronize(g->ctx->stream));
    const Dims& d = g->last;
    const size_t HWp = (size_t)d.Hp * d.Wp;
    const void* src = nullptr; size_t cap = 0;
    int32_t dims[8];
    switch (which) {
        case 0: src = g->score; cap = (size_t)d.h * d.w * 4; break;
        case 1: src = g->kp_index; cap = (size_t)SEL_CAP * 4; break;
        case 2: {
            ALCtrl c;
            SSLAM_HIP_CHECK(hipMemcpy(&c, g->ctrl, sizeof(c), hipMemcpyDeviceToHost));
            dims[0] = d.h; dims[1] = d.w; dims[2] = d.Hp; dims[3] = d.Wp; dims[4] = d.pl; dims[5] = d.pt;
            dims[6] = c.n_cand; dims[7] = c.n_kp;
            SSLAM_REQUIRE(bytes <= sizeof(dims), "sslam_aliked_debug_read: dims is 32 bytes");
            memcpy(dst, dims, bytes);
            return 0;
        }
        case 3: src = g->x1; cap = 16 * HWp * 4; break;
        case 4: src = g->x2; cap = 32 * HWp / 4 * 4; break;
        case 5: src = g->x3; cap = 64 * HWp / 64 * 4; break;
        case 6: src = g->x4; cap = 128 * HWp / 1024 * 4; break;
        case 7: src = g->img; cap = 3 * HWp * 4; break;
        case 8: src = g->nms; cap = (size_t)d.h * d.w * 4; break;
        case 9: src = g->kp_norm; cap = (size_t)g->max_kpts * 8; break;
        case 10: src = g->g1cl; cap = 32 * HWp * 4; break;           // (r04: the aggregation's outputs and the gated levels, for determinism checks)
        case 11: src = g->rnorm; cap = HWp * 4; break;
        case 12: src = g->g2; cap = 32 * HWp / 4 * 4; break;
        case 13: src = g->g3; cap = 32 * HWp / 64 * 4; break;
        case 14: src = g->g4; cap = 32 * HWp / 1024 * 4; break;
        case 15: src = g->pre2; cap = AGG_PRE * HWp / 4 * 4; break;
        case 16: src = g->pre3; cap = AGG_PRE * HWp / 64 * 4; break;
        case 17: src = g->pre4; cap = AGG_PRE * HWp / 1024 * 4; break;
        case 18: src = g->s8; cap = 8 * HWp * 4; break;
#if defined(AL_AGG_LOAD_NOP) && AL_AGG_LOAD_NOP == 3
        case 99: {                                                      // experiment: the packed-versus-scalar discrepancy record
            SSLAM_REQUIRE(bytes <= sizeof(unsigned) * 64, "sslam_aliked_debug_read: 256 bytes");
            SSLAM_HIP_CHECK(hipMemcpyFromSymbol(dst, HIP_SYMBOL(al_dbg_words), bytes));
            return 0;
        }
#endif
        default: SSLAM_REQUIRE(false, "sslam_aliked_debug_read: unknown buffer %d", which);
    }
    SSLAM_REQUIRE(bytes <= cap, "sslam_aliked_debug_read: %zu bytes requested, buffer has %zu", bytes, cap);
    SSLAM_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
