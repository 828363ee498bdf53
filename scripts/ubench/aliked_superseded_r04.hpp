// ALIKED kernels superseded during round 4 - kept OUTSIDE the product as the record HISTORY.md section 4 refers to (not compiled by
// build.py; they need the helpers of csrc/aliked_kernels.hip at the commit named below to build again).
//   al_conv3x3_mfma_kernel     r03 implicit-GEMM 3 x 3 convolution on the exact-fp32 MFMA, one tile per workgroup
//   al_conv3x3_sweep_kernel    the same as a four-wave vertical sweep with register prefetch (+ sweep_load / sweep_stash)
//   al_conv16_rows_kernel      block1 convolutions as one wave per workgroup rolling down a strip (exact-fp32 MFMA)
//   al_conv16h_rows_kernel     block1.conv2 on the split-precision pipe, rolling rows
//   al_conv32p_rows_kernel     block2.conv1 (pooling on load + 1 x 1 branch) on the split pipe, rolling rows
//   al_conv32_h_kernel         block2.conv2 on the split pipe, 4-row tiles
//   al_offset_conv_kernel      r03 offset convolution (k over the lanes, 18 reductions over 64 lanes per pixel group)
//   al_dcn_col / _gemm / _epilogue, al_dcn_wt, al_transpose   deformable conv as im2col + split-K fp32 GEMM + epilogue
// Last commit that launched them behind build switches: 14c766b (git show 14c766b:opencv-simpleslam_amd/csrc/aliked_kernels.hip).
// Replaced by al_block1_rows_kernel, al_block2_rows_kernel, al_offset_conv_h_kernel, al_dcn_h_kernel.
#if 0
#ifndef AL_CONV_UNROLL
#define AL_CONV_UNROLL 8
#endif
template <int CIN, int COUT, int POOL, bool DOWN, bool RESID, int RPW, bool CLOUT = false>
__global__ __launch_bounds__(256) void al_conv3x3_mfma_kernel(
    const float* __restrict__ in, int inH, int inW, float* __restrict__ out, int H, int W,
    const float* __restrict__ w /*[ci][tap][COUT]*/, const float* __restrict__ alpha, const float* __restrict__ beta,
    const float* __restrict__ wd /*[ci][COUT]*/, const float* __restrict__ bd, float* __restrict__ idn,
    const float* __restrict__ resid, size_t fs) {
    static_assert(COUT == 16 || COUT == 32, "two matrix-core shapes");
    in = fsh(in, blockIdx.z, fs); out = fsh(out, blockIdx.z, fs); idn = fsh(idn, blockIdx.z, fs); resid = fsh(resid, blockIdx.z, fs);
    static_assert(!(DOWN && COUT == 16), "the 1x1 branch is only built for the 32-row shape");
    constexpr bool M16 = COUT == 16;
    constexpr int KG = M16 ? 4 : 2;                                  // channels per MFMA
    constexpr int CINP = (CIN + KG - 1) / KG * KG;                   // input channels padded with zero channels
    constexpr int CTH = 4 * RPW;                                     // tile height: RPW rows per wave
    constexpr int TH = CTH + 2;
    constexpr int CT_TW = M16 ? CT_W + 8 : CT_W + 2;                 // tile row stride (floats)
    constexpr int IC = M16 ? 4 : 1;                                  // column of the first interior pixel
    constexpr int CHS = M16 ? (TH * CT_TW + 63 - 16) / 64 * 64 + 16 : TH * CT_TW;
    constexpr int K = 9 * CINP;
    constexpr int WLD = COUT;                                        // weight row stride in LDS
    __shared__ __attribute__((aligned(16))) float tile[CINP * CHS];
#ifndef AL_W_GLOBAL
#define AL_W_GLOBAL 1      // 32-row shape: the A operand (weights) straight from global memory / L1 instead of an LDS copy
#endif
    constexpr bool WG = AL_W_GLOBAL && !M16;
    __shared__ float wl[WG ? 1 : (K + (DOWN ? CINP : 0)) * WLD];     // [k = tap*CINP + ci][co] (+ 1x1 rows)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int x0 = blockIdx.x * CT_W, y0 = blockIdx.y * CTH;
#if CONV_ABL & 2
    for (int i = t; i < CINP * CHS; i += 256) tile[i] = 0.5f;
    if (0)
#endif
    {
        constexpr int ROWS = CINP * TH;
        // interior: 8 x 16 bytes per tile row (unrolled: the loads of several trips in flight, then their LDS stores -
        // rolled, every trip waited out its own load: the convs' waves were parked on memory half of their cycles)
#pragma unroll AL_CONV_UNROLL
        for (int idx = t; idx < ROWS * 8; idx += 256) {
            const int row = idx >> 3, v4 = idx & 7;
            const int c = row / TH, rr = row % TH;
            const int yy = y0 + rr - 1, xx = x0 + 4 * v4;
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (c < CIN && yy >= 0 && yy < H && xx < W) {
                if (POOL == 1) {
                    const float* src = in + ((size_t)c * inH + yy) * inW + xx;
                    if (xx + 3 < W) v = *reinterpret_cast<const float4*>(src);
                    else { v.x = src[0]; if (xx + 1 < W) v.y = src[1]; if (xx + 2 < W) v.z = src[2]; }
                } else {
                    // 2x2 average on load, summed in the order (0,0) (0,1) (1,0) (1,1)
                    const float* r0 = in + ((size_t)c * inH + yy * 2) * inW + xx * 2;
                    const float* r1 = r0 + inW;
                    float o[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (xx + 3 < W) {
                        const float4 a0 = *reinterpret_cast<const float4*>(r0), a1 = *reinterpret_cast<const float4*>(r0 + 4);
                        const float4 b0 = *reinterpret_cast<const float4*>(r1), b1 = *reinterpret_cast<const float4*>(r1 + 4);
                        o[0] = (((a0.x + a0.y) + b0.x) + b0.y) / 4.0f; o[1] = (((a0.z + a0.w) + b0.z) + b0.w) / 4.0f;
                        o[2] = (((a1.x + a1.y) + b1.x) + b1.y) / 4.0f; o[3] = (((a1.z + a1.w) + b1.z) + b1.w) / 4.0f;
                    } else {
                        for (int j = 0; j < 4 && xx + j < W; ++j)
                            o[j] = (((r0[2 * j] + r0[2 * j + 1]) + r1[2 * j]) + r1[2 * j + 1]) / 4.0f;
                    }
                    v = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
            float* dst = &tile[c * CHS + rr * CT_TW + IC + 4 * v4];
            if constexpr (M16) *reinterpret_cast<float4*>(dst) = v;
            else { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w; }
        }
        // halo columns x0 - 1 and x0 + 32
#pragma unroll AL_CONV_UNROLL
        for (int idx = t; idx < ROWS * 2; idx += 256) {
            const int row = idx >> 1, side = idx & 1;
            const int c = row / TH, rr = row % TH;
            const int yy = y0 + rr - 1, xx = side ? x0 + CT_W : x0 - 1;
            float v = 0.0f;
            if (c < CIN && yy >= 0 && yy < H && xx >= 0 && xx < W) {
                if (POOL == 1) {
                    v = in[((size_t)c * inH + yy) * inW + xx];
                } else {
                    const float* r0 = in + ((size_t)c * inH + yy * 2) * inW + xx * 2;
                    v = (((r0[0] + r0[1]) + r0[inW]) + r0[inW + 1]) / 4.0f;
                }
            }
            tile[c * CHS + rr * CT_TW + (side ? IC + CT_W : IC - 1)] = v;
        }
    }
    if constexpr (!WG) {
#pragma unroll AL_CONV_UNROLL
        for (int i = t; i < K * WLD; i += 256) {          // global [ci][tap][co] -> LDS [tap][ci][co], zero padded
            const int co = i % WLD, k = i / WLD, tap = k / CINP, ci = k % CINP;
            wl[i] = ci < CIN ? w[(ci * 9 + tap) * COUT + co] : 0.0f;
        }
        if (DOWN)
            for (int i = t; i < CINP * WLD; i += 256) {
                const int co = i % WLD, ci = i / WLD;
                wl[K * WLD + i] = ci < CIN ? wd[ci * COUT + co] : 0.0f;
            }
    }
    __syncthreads();

    if constexpr (M16) {
        // lane = (k = lane >> 4, n = lane & 15): A = wl[k0 + k][co = n], B = tile[ci0 + k][row][px = 16 half + n]
        const int kk = lane >> 4, n = lane & 15;
        f32x4 acc[RPW][2];
#pragma unroll
        for (int q = 0; q < RPW; ++q)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) acc[q][hf] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const float* bbase = tile + kk * CHS + (RPW * wave) * CT_TW + (IC - 1) + n;
        const float* abase = wl + kk * WLD + n;
#if !(CONV_ABL & 1)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int g4 = 0; g4 < CINP / 4; ++g4) {
                const float a = abase[(tap * CINP + 4 * g4) * WLD];
                const int boff = (4 * g4) * CHS + (tap / 3) * CT_TW + (tap % 3);
#pragma unroll
                for (int q = 0; q < RPW; ++q)
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
                        acc[q][hf] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bbase[boff + q * CT_TW + 16 * hf], acc[q][hf], 0, 0, 0);
            }
        }
#else
        for (int q = 0; q < RPW; ++q) acc[q][0][0] = bbase[q * CT_TW] * abase[0];
#endif
        // accumulator register i of lane (kk, n): co = 4 kk + i, pixel = 16 half + n
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int y = y0 + RPW * wave + q;
            if (y >= H) continue;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int x = x0 + 16 * hf + n;
                if (x >= W) continue;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int co = 4 * kk + i;
                    const size_t o = ((size_t)co * H + y) * W + x;
                    float v = fmaf(acc[q][hf][i], alpha[co], beta[co]);
                    if (RESID) v += resid[o];
#if CONV_ABL & 4
                    if (v == 123.456f)
#endif
                    out[o] = selu(v);
                }
            }
        }
    } else {
        const int h = lane >> 5, px = lane & 31;
        f32x16 acc[RPW], dn[DOWN ? RPW : 1];
#pragma unroll
        for (int q = 0; q < RPW; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[q][r] = 0.0f; if (DOWN) dn[q][r] = 0.0f; }
        // lane bases: B operand (pixels) = tile[(ci0 + h)][row + dy][px + dx]; A operand = wl[(k0 + h)][co = px]
        const float* bbase = tile + h * CHS + (RPW * wave) * CT_TW + (IC - 1) + px;
        const float* abase = wl + h * WLD + px;
#if !(CONV_ABL & 1)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int c2 = 0; c2 < CINP / 2; ++c2) {
                const int koff = (tap * CINP + 2 * c2) * WLD;                        // A: rows k0, k0+1
                const int boff = (2 * c2) * CHS + (tap / 3) * CT_TW + (tap % 3);      // B: channels 2c2, 2c2+1
                // (32 consecutive output channels of one (ci, tap) per lane half: two coalesced 128-byte runs of the
                //  [ci][tap][co] weights as they are - no per-workgroup LDS copy, 37 KB of LDS less, three workgroups per CU)
                const float a = WG ? w[((2 * c2 + h) * 9 + tap) * COUT + px] : abase[koff];
#pragma unroll
                for (int q = 0; q < RPW; ++q) acc[q] = sslam::mfma32(a, bbase[boff + q * CT_TW], acc[q]);
            }
        }
#else
        for (int q = 0; q < RPW; ++q) acc[q][0] = bbase[q * CT_TW] * abase[0];
#endif
        if (DOWN) {
#pragma unroll
            for (int c2 = 0; c2 < CINP / 2; ++c2) {
                const float a = WG ? wd[(2 * c2 + h) * COUT + px] : abase[(K + 2 * c2) * WLD];
                const int boff = (2 * c2) * CHS + CT_TW + 1;                          // centre tap
#pragma unroll
                for (int q = 0; q < RPW; ++q) dn[q] = sslam::mfma32(a, bbase[boff + q * CT_TW], dn[q]);
            }
        }
        const int x = x0 + px;
        if (x >= W) return;
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int y = y0 + RPW * wave + q;
            if (y >= H) continue;
            if constexpr (CLOUT) {
                // r04: the only consumer of this map (block2.conv2, al_conv32_h_kernel) runs on the split-precision matrix
                // path: the output leaves channel-last as fp16 (hi, lo) planes [H][W][32] (the bytes of the planar fp32 map);
                // a lane holds four consecutive channels per register quad: 8-byte pieces
                _Float16* oh = reinterpret_cast<_Float16*>(out);
                const size_t plane = (size_t)H * W * 32;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    float vv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g4 + e, co = acc_row(r, lane);
                        vv[e] = selu(fmaf(acc[q][r], alpha[co], beta[co]));
                        if (DOWN) idn[((size_t)co * H + y) * W + x] = dn[q][r] + bd[co];
                    }
                    unsigned h01, l01, h23, l23; float amax = 0.0f;
                    sslam::split2_fast(vv[0], vv[1], h01, l01, amax);
                    sslam::split2_fast(vv[2], vv[3], h23, l23, amax);
                    const size_t o = ((size_t)y * W + x) * 32 + 8 * g4 + 4 * (lane >> 5);
                    *reinterpret_cast<uint2*>(oh + o) = make_uint2(h01, h23);
                    *reinterpret_cast<uint2*>(oh + plane + o) = make_uint2(l01, l23);
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = acc_row(r, lane);
                const size_t o = ((size_t)co * H + y) * W + x;
                float v = fmaf(acc[q][r], alpha[co], beta[co]);
                if (RESID) v += resid[o];
#if CONV_ABL & 4
                if (v == 123.456f)
#endif
                out[o] = selu(v);
                if (DOWN) idn[o] = dn[q][r] + bd[co];
            }
        }
    }
}


#ifndef AL_SWEEP_WPE
#define AL_SWEEP_WPE 2      // 228 registers, two waves per SIMD: 129 us per launch against 135 squeezed into 168 with spills
#endif
template <int CIN, int COUT, int POOL, bool DOWN, int NS, bool CLOUT = false>
__global__ __launch_bounds__(256, COUT == 32 ? AL_SWEEP_WPE : 4) void al_conv3x3_sweep_kernel(
    const float* __restrict__ in, int inH, int inW, float* __restrict__ out, int H, int W,
    const float* __restrict__ w /*[ci][tap][COUT]*/, const float* __restrict__ alpha, const float* __restrict__ beta,
    const float* __restrict__ wd /*[ci][COUT]*/, const float* __restrict__ bd, float* __restrict__ idn, size_t fs) {
    static_assert(COUT == 16 || COUT == 32, "two matrix-core shapes");
    static_assert(!(DOWN && COUT == 16), "the 1x1 branch is only built for the 32-row shape");
    in = fsh(in, blockIdx.z, fs); out = fsh(out, blockIdx.z, fs); idn = fsh(idn, blockIdx.z, fs);
    constexpr bool M16 = COUT == 16;
    constexpr int KG = M16 ? 4 : 2;
    constexpr int CINP = (CIN + KG - 1) / KG * KG;
    constexpr int TH = 6;
    constexpr int CT_TW = M16 ? CT_W + 8 : CT_W + 2;
    constexpr int IC = M16 ? 4 : 1;
    constexpr int CHS = M16 ? (TH * CT_TW + 63 - 16) / 64 * 64 + 16 : TH * CT_TW;
    constexpr int K = 9 * CINP;
    constexpr int WLD = COUT;
    constexpr bool WG = false;       // weights in LDS, staged once per strip (global A operands would queue behind the prefetch: loads return in order)
#ifndef AL_SWEEP_DB
#define AL_SWEEP_DB 1      // 2: two LDS tiles (one barrier per step); 1: one tile, two barriers per step, twice the workgroups per CU
#endif
    constexpr int DB = AL_SWEEP_DB;
    __shared__ __attribute__((aligned(16))) float tile[DB][CINP * CHS];
    __shared__ float wl[WG ? 1 : (K + (DOWN ? CINP : 0)) * WLD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int x0 = blockIdx.x * CT_W, yb = blockIdx.y * (4 * NS);
    constexpr int ROWS = CINP * TH, NI = (ROWS * 8 + 255) / 256, NH = (ROWS * 2 + 255) / 256, NR = POOL == 2 ? 4 : 1;
    float4 ri[NI][NR]; float rh[NH][NR];

    unsigned ok = sweep_load<CIN, CINP, POOL>(in, inH, inW, H, W, x0, yb, t, ri, rh);
    if constexpr (!WG) {
#pragma unroll AL_CONV_UNROLL
        for (int i = t; i < K * WLD; i += 256) {
            const int co = i % WLD, k = i / WLD, tap = k / CINP, ci = k % CINP;
            wl[i] = ci < CIN ? w[(ci * 9 + tap) * COUT + co] : 0.0f;
        }
        if (DOWN)
            for (int i = t; i < CINP * WLD; i += 256) {
                const int co = i % WLD, ci = i / WLD;
                wl[K * WLD + i] = ci < CIN ? wd[ci * COUT + co] : 0.0f;
            }
    }
    sweep_stash<CINP, POOL, M16>(tile[0], t, ok, ri, rh);
    __syncthreads();

    // the BN affine of the output channels, once per strip: in registers for the 16-row shape (4 + 4 per lane), in LDS for the
    // 32-row shape (48 registers there would leave one wave per SIMD).  (r04: inside the epilogue each alpha[co] / beta[co]
    // was a global load with a full s_waitcnt vmcnt(0) behind it - 8 / 16 serial memory round trips per wave and step, and
    // the wait also drained the prefetch and the stores in flight)
    constexpr int NCO = M16 ? 4 : 1;
    float alr[NCO], ber[NCO];
    __shared__ float aff[M16 ? 1 : 3 * 32];
    if constexpr (M16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { alr[r] = alpha[4 * (lane >> 4) + r]; ber[r] = beta[4 * (lane >> 4) + r]; }
    } else {
        if (t < 32) { aff[t] = alpha[t]; aff[32 + t] = beta[t]; aff[64 + t] = DOWN ? bd[t] : 0.0f; }
        __syncthreads();
    }
#pragma unroll 1
    for (int st = 0; st < NS; ++st) {
        const int y0 = yb + 4 * st;
        if (y0 >= H) break;                                    // (uniform)
        const bool more = st + 1 < NS && y0 + 4 < H;
        // (unconditional: under `if (more)` the registers become phis whose copies wait for the loads in front of the matrix loop;
        //  the last step re-reads its own tile from the caches and drops it)
        if (!(AL_SWEEP_ABL & 2)) ok = sweep_load<CIN, CINP, POOL>(in, inH, inW, H, W, x0, more ? y0 + 4 : y0, t, ri, rh);
        __builtin_amdgcn_sched_barrier(0);     // (the scheduler would pull the pooling adds of the stash - and the wait for these loads - up here)
        const float* tl = tile[DB == 2 ? (st & 1) : 0];
        const int y = y0 + wave;
        if constexpr (M16) {
            const int kk = lane >> 4, n = lane & 15;
            f32x4 acc[2];
            acc[0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; acc[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            const float* bbase = tl + kk * CHS + wave * CT_TW + (IC - 1) + n;
            const float* abase = wl + kk * WLD + n;
#pragma unroll
            for (int tap = 0; tap < ((AL_SWEEP_ABL & 1) ? 1 : 9); ++tap) {
#pragma unroll
                for (int g4 = 0; g4 < CINP / 4; ++g4) {
                    const float a = abase[(tap * CINP + 4 * g4) * WLD];
                    const int boff = (4 * g4) * CHS + (tap / 3) * CT_TW + (tap % 3);
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
                        acc[hf] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bbase[boff + 16 * hf], acc[hf], 0, 0, 0);
                }
            }
            if (y < H && !((AL_SWEEP_ABL & 4) && acc[0][0] != 123.456f)) {
                // wave-uniform row pointer + a 32-bit lane offset that does not depend on the step (channel plane + column)
                float* orow = out + (size_t)y * W + x0;
                const unsigned HW = (unsigned)H * W, lo = (unsigned)(4 * kk) * HW + n;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    if (x0 + 16 * hf + n >= W) continue;
#pragma unroll
                    for (int i = 0; i < 4; ++i) orow[lo + i * HW + 16 * hf] = SWEEP_SELU(fmaf(acc[hf][i], alr[i], ber[i]));
                }
            }
        } else {
            const int h = lane >> 5, px = lane & 31;
            f32x16 acc, dn;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[r] = 0.0f; dn[r] = 0.0f; }
            const float* bbase = tl + h * CHS + wave * CT_TW + (IC - 1) + px;
            const float* abase = wl + h * WLD + px;
#pragma unroll
            for (int tap = 0; tap < ((AL_SWEEP_ABL & 1) ? 1 : 9); ++tap) {
#pragma unroll
                for (int c2 = 0; c2 < CINP / 2; ++c2) {
                    const int koff = (tap * CINP + 2 * c2) * WLD;
                    const int boff = (2 * c2) * CHS + (tap / 3) * CT_TW + (tap % 3);
                    const float a = WG ? w[((2 * c2 + h) * 9 + tap) * COUT + px] : abase[koff];
                    acc = sslam::mfma32(a, bbase[boff], acc);
                }
            }
            if (DOWN) {
#pragma unroll
                for (int c2 = 0; c2 < CINP / 2; ++c2) {
                    const float a = WG ? wd[(2 * c2 + h) * COUT + px] : abase[(K + 2 * c2) * WLD];
                    dn = sslam::mfma32(a, bbase[(2 * c2) * CHS + CT_TW + 1], dn);
                }
            }
            const int x = x0 + px;
            if (x < W && y < H && !((AL_SWEEP_ABL & 4) && acc[0] != 123.456f)) {
                // planar outputs: wave-uniform row pointer + 32-bit lane offset (channel plane of the lane half + column);
                // register r adds the plane of acc_row(r, .) = 8 (r / 4) + r % 4 (+ 4 h in the lane part)
                const unsigned HW = (unsigned)H * W, lo = (unsigned)(4 * h) * HW + px;
                float* irow = DOWN ? idn + (size_t)y * W + x0 : nullptr;
                if constexpr (CLOUT) {
                    _Float16* oh = reinterpret_cast<_Float16*>(out);
                    const size_t plane = (size_t)H * W * 32;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        float vv[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int r = 4 * g4 + e, co = acc_row(r, lane);
                            vv[e] = SWEEP_SELU(fmaf(acc[r], aff[co], aff[32 + co]));
                            if (DOWN) irow[lo + (8 * (r / 4) + r % 4) * HW] = dn[r] + aff[64 + co];
                        }
                        unsigned h01, l01, h23, l23; float amax = 0.0f;
                        sslam::split2_fast(vv[0], vv[1], h01, l01, amax);
                        sslam::split2_fast(vv[2], vv[3], h23, l23, amax);
                        const size_t o = ((size_t)y * W + x) * 32 + 8 * g4 + 4 * (lane >> 5);
                        *reinterpret_cast<uint2*>(oh + o) = make_uint2(h01, h23);
                        *reinterpret_cast<uint2*>(oh + plane + o) = make_uint2(l01, l23);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = acc_row(r, lane);
                        float* orow = out + (size_t)y * W + x0;
                        orow[lo + (8 * (r / 4) + r % 4) * HW] = SWEEP_SELU(fmaf(acc[r], aff[co], aff[32 + co]));
                        if (DOWN) irow[lo + (8 * (r / 4) + r % 4) * HW] = dn[r] + aff[64 + co];
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (DB == 1) __syncthreads();          // every wave is done reading the tile
        if (more) sweep_stash<CINP, POOL, M16>(tile[DB == 2 ? ((st + 1) & 1) : 0], t, ok, ri, rh);
        __syncthreads();
    }
}


template <int CIN, bool SPLIT_OUT = false>      // SPLIT_OUT: the map leaves channel-last as fp16 (hi, lo) planes [H][W][16] for al_conv16h_rows_kernel
__global__ __launch_bounds__(64) void al_conv16_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int hs,
                                                           const float* __restrict__ w /*[ci][tap][16]*/,
                                                           const float* __restrict__ alpha, const float* __restrict__ beta, size_t fs) {
    in = fsh(in, blockIdx.z, fs); out = fsh(out, blockIdx.z, fs);
    constexpr int CINP = (CIN + 3) / 4 * 4, G = CINP / 4, RS = 48, SLOT = CINP * RS, NI = (CINP * 8 + 63) / 64;
    __shared__ __attribute__((aligned(16))) float ring[3 * SLOT];
    const int lane = threadIdx.x, kk = lane >> 4, n = lane & 15;
    const int x0 = blockIdx.x * CT_W, yb = blockIdx.y * hs, ye = min(yb + hs, H);
    float a[9][G];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int g = 0; g < G; ++g) a[tap][g] = 4 * g + kk < CIN ? w[((4 * g + kk) * 9 + tap) * 16 + n] : 0.0f;
    float alr[4], ber[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { alr[i] = alpha[4 * kk + i]; ber[i] = beta[4 * kk + i]; }
    // the lane's pieces of an input row: NI float4 of the interior (channel = idx / 8, 4 pixels) + one halo value
    int ich[NI], iv4[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) { const int idx = min(lane + 64 * j, CINP * 8 - 1); ich[j] = idx >> 3; iv4[j] = idx & 7; }
    const int hl = min(lane, CINP * 2 - 1), hch = hl >> 1, hside = hl & 1;
    const int hx = hside ? x0 + CT_W : x0 - 1;
    const bool hok = lane < CINP * 2 && hch < CIN && hx >= 0 && hx < W;
    const size_t HW = (size_t)H * W;
    float4 ri[NI]; float rh;
    auto load_row = [&](int yy) {               // unconditional, clamped (validity applied at stash time)
        const int yc = min(max(yy, 0), H - 1);
#pragma unroll
        for (int j = 0; j < NI; ++j)
            ri[j] = *reinterpret_cast<const float4*>(in + (size_t)min(ich[j], CIN - 1) * HW + (size_t)yc * W + x0 + 4 * iv4[j]);
        rh = in[(size_t)min(hch, CIN - 1) * HW + (size_t)yc * W + min(max(hx, 0), W - 1)];
    };
    auto stash_row = [&](int slot, int yy) {
        const bool rowok = yy >= 0 && yy < H;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            if (lane + 64 * j >= CINP * 8) continue;
            const float4 v = rowok && ich[j] < CIN ? ri[j] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            *reinterpret_cast<float4*>(&ring[slot * SLOT + ich[j] * RS + 4 + 4 * iv4[j]]) = v;
        }
        if (lane < CINP * 2) ring[slot * SLOT + hch * RS + (hside ? 4 + CT_W : 3)] = rowok && hok ? rh : 0.0f;
    };
    // rows yb - 1, yb, yb + 1 -> slots 0, 1, 2
#pragma unroll
    for (int r = 0; r < 3; ++r) { load_row(yb - 1 + r); stash_row(r, yb - 1 + r); }
    __syncthreads();
    const float* bl = ring + kk * RS + 3 + n;
    const unsigned HWu = (unsigned)HW, lo = (unsigned)(4 * kk) * HWu + n;
    auto step = [&](auto ph, int y) {           // rows y - 1, y, y + 1 in slots PH, PH + 1, PH + 2 (mod 3); row y + 2 -> slot PH
        constexpr int PH = decltype(ph)::value;
        load_row(y + 2);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[2];
        acc[0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; acc[1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            constexpr int dummy = 0; (void)dummy;
            const int slot = (PH + tap / 3) % 3;
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
                    acc[hf] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tap][g], bl[slot * SLOT + 4 * g * RS + tap % 3 + 16 * hf], acc[hf], 0, 0, 0);
        }
        if constexpr (SPLIT_OUT) {
            // a lane holds channels 4 kk .. 4 kk + 3 of its pixel: 8 bytes per plane, the four k-lanes of a pixel side by side
            _Float16* oh = reinterpret_cast<_Float16*>(out) + ((size_t)y * W + x0) * 16 + 4 * kk;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                if (x0 + 16 * hf + n >= W) continue;
                float vv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) vv[i] = SWEEP_SELU(fmaf(acc[hf][i], alr[i], ber[i]));
                unsigned h01, l01, h23, l23; float amax = 0.0f;
                sslam::split2_fast(vv[0], vv[1], h01, l01, amax);
                sslam::split2_fast(vv[2], vv[3], h23, l23, amax);
                *reinterpret_cast<uint2*>(oh + (16 * hf + n) * 16) = make_uint2(h01, h23);
                *reinterpret_cast<uint2*>(oh + HW * 16 + (16 * hf + n) * 16) = make_uint2(l01, l23);
            }
        } else {
        float* orow = out + (size_t)y * W + x0;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            if (x0 + 16 * hf + n >= W) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) orow[lo + i * HWu + 16 * hf] = SWEEP_SELU(fmaf(acc[hf][i], alr[i], ber[i]));
        }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                         // (one wave: orders this step's LDS reads before the overwrite of slot PH)
        stash_row(PH, y + 2);
        __syncthreads();
    };
    for (int y = yb; y < ye; y += 3) {
        step(std::integral_constant<int, 0>{}, y);
        if (y + 1 >= ye) break;
        step(std::integral_constant<int, 1>{}, y + 1);
        if (y + 2 >= ye) break;
        step(std::integral_constant<int, 2>{}, y + 2);
    }
}


#ifndef AL_C16H_WPE
#define AL_C16H_WPE 1
#endif
__global__ __launch_bounds__(64, AL_C16H_WPE) void al_conv16h_rows_kernel(const _Float16* __restrict__ in /* hi plane [H][W][16]; lo plane H W 16 halves behind */,
                                                            float* __restrict__ out, int H, int W, int hs, const _Float16* __restrict__ wf,
                                                            const float* __restrict__ alpha, const float* __restrict__ beta, size_t fs) {
    in = fsh(in, blockIdx.z, fs); out = fsh(out, blockIdx.z, fs);
    constexpr int PXS = 24, ROWH = (CT_W + 2) * PXS, PLH = 3 * ROWH;       // halves: pixel stride, row, plane (3 ring rows)
    __shared__ __attribute__((aligned(16))) _Float16 ring[2 * PLH];
    const int lane = threadIdx.x, kk = lane >> 4, n = lane & 15;
    const int x0 = blockIdx.x * CT_W, yb = blockIdx.y * hs, ye = min(yb + hs, H);
    const size_t HW = (size_t)H * W;
    sslam::half8 ah[5], al[5];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        ah[ks] = *reinterpret_cast<const sslam::half8*>(wf + ((ks * 2 + 0) * 64 + lane) * 8);
        al[ks] = *reinterpret_cast<const sslam::half8*>(wf + ((ks * 2 + 1) * 64 + lane) * 8);
    }
    float alr[4], ber[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { alr[i] = alpha[4 * kk + i]; ber[i] = beta[4 * kk + i]; }
    // an input row = 2 planes x 34 pixels x two 16-byte pieces = 136 pieces, three per lane (the last one on 8 lanes)
    int pofs[3], lofs[3]; bool pok[3], pin[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int idx = lane + 64 * j;
        pin[j] = idx < 136;
        const int id = min(idx, 135), pl = id / 68, rem = id % 68, px = rem >> 1, hf8 = rem & 1, xx = x0 - 1 + px;
        pok[j] = pin[j] && xx >= 0 && xx < W;
        pofs[j] = min(max(xx, 0), W - 1) * 16 + 8 * hf8;          // halves inside the row of the plane
        lofs[j] = pl * PLH + px * PXS + 8 * hf8;
        if (pl) pofs[j] += 0;                                      // (plane offset added as a 64-bit term below)
    }
    const size_t plane = HW * 16;
    uint4 ri[3];
    auto load_row = [&](int yy) {
        const int yc = min(max(yy, 0), H - 1);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int id = min(lane + 64 * j, 135);
            ri[j] = *reinterpret_cast<const uint4*>(in + (id >= 68 ? plane : (size_t)0) + (size_t)yc * W * 16 + pofs[j]);
        }
    };
    auto stash_row = [&](int slot, int yy) {
        const bool rowok = yy >= 0 && yy < H;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (pin[j]) *reinterpret_cast<uint4*>(&ring[lofs[j] + slot * ROWH]) = rowok && pok[j] ? ri[j] : make_uint4(0u, 0u, 0u, 0u);
    };
#pragma unroll
    for (int r = 0; r < 3; ++r) { load_row(yb - 1 + r); stash_row(r, yb - 1 + r); }
    __syncthreads();
    // B fragment of lane (kk, n) in k-step ks: tap 2 ks + (kk >> 1), channels 8 (kk & 1) .. + 7, pixel n (+ 16 hf) + dx
    int boff[5][3];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        const int tap = min(2 * ks + (kk >> 1), 8);                // (the tenth tap has zero weights: any valid address)
#pragma unroll
        for (int ph = 0; ph < 3; ++ph) boff[ks][ph] = ((ph + tap / 3) % 3) * ROWH + (n + tap % 3) * PXS + 8 * (kk & 1);
    }
    const unsigned HWu = (unsigned)HW, lo = (unsigned)(4 * kk) * HWu + n;
    auto step = [&](auto ph, int y) {
        constexpr int PH = decltype(ph)::value;
        load_row(y + 2);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 c1[2], c2[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) { c1[hf] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; c2[hf] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
#pragma unroll
        for (int ks = 0; ks < 5; ++ks)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const sslam::half8 xh = *reinterpret_cast<const sslam::half8*>(&ring[boff[ks][PH] + 16 * hf * PXS]);
                const sslam::half8 xl = *reinterpret_cast<const sslam::half8*>(&ring[boff[ks][PH] + 16 * hf * PXS + PLH]);
                c1[hf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], xh, c1[hf], 0, 0, 0);
                c2[hf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], xl, c2[hf], 0, 0, 0);
                c2[hf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ks], xh, c2[hf], 0, 0, 0);
            }
        float* orow = out + (size_t)y * W + x0;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            if (x0 + 16 * hf + n >= W) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                orow[lo + i * HWu + 16 * hf] = SWEEP_SELU(fmaf(c1[hf][i] + c2[hf][i] * sslam::SPLIT_INV, alr[i], ber[i]));
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        stash_row(PH, y + 2);
        __syncthreads();
    };
    for (int y = yb; y < ye; y += 3) {
        step(std::integral_constant<int, 0>{}, y);
        if (y + 1 >= ye) break;
        step(std::integral_constant<int, 1>{}, y + 1);
        if (y + 2 >= ye) break;
        step(std::integral_constant<int, 2>{}, y + 2);
    }
}


#ifndef AL_C32P_WPE
#define AL_C32P_WPE 1
#endif
__global__ __launch_bounds__(64, AL_C32P_WPE) void al_conv32p_rows_kernel(const float* __restrict__ in /*[16][2 H][2 W]*/, float* __restrict__ out /* t2: split planes [H][W][32] */,
                                                            float* __restrict__ idn /*[32][H][W]*/, int H, int W, int hs, const _Float16* __restrict__ wf,
                                                            const float* __restrict__ alpha, const float* __restrict__ beta, const float* __restrict__ bd, size_t fs) {
    in = fsh(in, blockIdx.z, fs); out = fsh(out, blockIdx.z, fs); idn = fsh(idn, blockIdx.z, fs);
    constexpr int PXS = 24, ROWH = (CT_W + 2) * PXS, PLH = 3 * ROWH;
    __shared__ __attribute__((aligned(16))) _Float16 ring[2 * PLH];
    __shared__ float aff[96];
    const int lane = threadIdx.x, h = lane >> 5, px = lane & 31;
    const int x0 = blockIdx.x * CT_W, yb = blockIdx.y * hs, ye = min(yb + hs, H);
    const int inW = 2 * W, inH = 2 * H;
    // A fragments: the hi planes in registers for the whole strip, the lo planes in LDS (both in registers: spills at 256)
    __shared__ __attribute__((aligned(16))) _Float16 wlo[10 * 64 * 8];
    sslam::half8 ah[10];
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) {
        ah[ks] = *reinterpret_cast<const sslam::half8*>(wf + ((ks * 2 + 0) * 64 + lane) * 8);
        *reinterpret_cast<sslam::half8*>(&wlo[(ks * 64 + lane) * 8]) = *reinterpret_cast<const sslam::half8*>(wf + ((ks * 2 + 1) * 64 + lane) * 8);
    }
    const _Float16* wl = wlo + lane * 8;
    if (lane < 32) { aff[lane] = alpha[lane]; aff[32 + lane] = beta[lane]; aff[64 + lane] = bd[lane]; }
    // a ring row = the 2 x 2 averages of two input rows: 16 channels x 16 float4 (= 2 pooled pixels) x 2 rows, four items per lane,
    // + the halo columns (x0 - 1, x0 + 32): 16 channels x 2 sides on lanes 0 .. 31
    const int hch = (lane & 31) >> 1, hside = lane & 1, hx = hside ? x0 + CT_W : x0 - 1;
    const bool hok = lane < 32 && hx >= 0 && hx < W;
    const size_t inHW = (size_t)inH * inW;
    float4 ra[4], rb[4]; float2 ha, hb;
    // (wave-uniform row pointers + 32-bit lane offsets: 64-bit lane pointers per item spill at two waves per SIMD)
    unsigned iofs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int idx = lane + 64 * j; iofs[j] = 4u * ((unsigned)(idx >> 4) * (unsigned)inHW + 2 * x0 + 4 * (idx & 15)); }
    const unsigned hofs = 4u * ((unsigned)hch * (unsigned)inHW + 2 * min(max(hx, 0), W - 1));
    auto load_row = [&](int yy) {
        const int yc = min(max(yy, 0), H - 1);
        const float* r0 = in + (size_t)(2 * yc) * inW;
        const float* r1 = r0 + inW;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ra[j] = at_b(reinterpret_cast<const float4*>(r0), iofs[j]); rb[j] = at_b(reinterpret_cast<const float4*>(r1), iofs[j]);
        }
        ha = at_b(reinterpret_cast<const float2*>(r0), hofs); hb = at_b(reinterpret_cast<const float2*>(r1), hofs);
    };
    auto put = [&](int o, float v0, float v1, bool two) {      // pooled values of one channel at ring pixels o, o + PXS
        unsigned h2, l2; float amax = 0.0f;
        sslam::split2_fast(v0, v1, h2, l2, amax);
        ring[o] = __builtin_bit_cast(_Float16, (unsigned short)(h2 & 0xffffu));
        ring[o + PLH] = __builtin_bit_cast(_Float16, (unsigned short)(l2 & 0xffffu));
        if (two) {
            ring[o + PXS] = __builtin_bit_cast(_Float16, (unsigned short)(h2 >> 16));
            ring[o + PXS + PLH] = __builtin_bit_cast(_Float16, (unsigned short)(l2 >> 16));
        }
    };
    auto stash_row = [&](int slot, int yy) {
        const bool rowok = yy >= 0 && yy < H;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = lane + 64 * j, ch = idx >> 4, v4 = idx & 15;
            // 2 x 2 average summed in the order (0,0) (0,1) (1,0) (1,1)
            const float p0 = rowok ? (((ra[j].x + ra[j].y) + rb[j].x) + rb[j].y) / 4.0f : 0.0f;
            const float p1 = rowok ? (((ra[j].z + ra[j].w) + rb[j].z) + rb[j].w) / 4.0f : 0.0f;
            put(slot * ROWH + (1 + 2 * v4) * PXS + ch, p0, p1, true);
        }
        if (lane < 32) {
            const float p = rowok && hok ? (((ha.x + ha.y) + hb.x) + hb.y) / 4.0f : 0.0f;
            put(slot * ROWH + (hside ? CT_W + 1 : 0) * PXS + hch, p, 0.0f, false);
        }
    };
#pragma unroll
    for (int r = 0; r < 3; ++r) { load_row(yb - 1 + r); stash_row(r, yb - 1 + r); }
    __syncthreads();
    const _Float16* bl = ring + px * PXS + 8 * h;
    const size_t HW = (size_t)H * W;
    const unsigned HWb = 4u * (unsigned)HW, lo = (unsigned)(4 * h) * HWb + 4u * px;      // bytes
    auto step = [&](auto ph, int y) {
        constexpr int PH = decltype(ph)::value;
        load_row(y + 2);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 c1, c2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { c1[r] = 0.0f; c2[r] = 0.0f; }
#pragma unroll
        for (int ks = 0; ks < 9; ++ks) {
            const int o = ((PH + ks / 3) % 3) * ROWH + (ks % 3) * PXS;
            const sslam::half8 xh = *reinterpret_cast<const sslam::half8*>(bl + o);
            const sslam::half8 xl = *reinterpret_cast<const sslam::half8*>(bl + o + PLH);
            c1 = sslam::mfma16(ah[ks], xh, c1);
            c2 = sslam::mfma16(ah[ks], xl, c2);
            c2 = sslam::mfma16(*reinterpret_cast<const sslam::half8*>(wl + ks * 64 * 8), xh, c2);
        }
        const bool live = x0 + px < W;
        if (live) {
            _Float16* orow = reinterpret_cast<_Float16*>(out) + ((size_t)y * W + x0) * 32;
            const unsigned ol = 2u * (px * 32 + 4 * h), opl = 2u * (unsigned)HW * 32;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float vv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g4 + e, co = acc_row(r, lane);
                    vv[e] = SWEEP_SELU(fmaf(c1[r] + c2[r] * sslam::SPLIT_INV, aff[co], aff[32 + co]));
                }
                unsigned h01, l01, h23, l23; float amax = 0.0f;
                sslam::split2_fast(vv[0], vv[1], h01, l01, amax);
                sslam::split2_fast(vv[2], vv[3], h23, l23, amax);
                at_b(reinterpret_cast<uint2*>(orow), ol + 16 * g4) = make_uint2(h01, h23);
                at_b(reinterpret_cast<uint2*>(orow), opl + ol + 16 * g4) = make_uint2(l01, l23);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        {   // the 1 x 1 branch on the centre tap, in the same accumulator registers (they would not fit twice at two waves per SIMD)
            const int o = ((PH + 1) % 3) * ROWH + PXS;
            const sslam::half8 xh = *reinterpret_cast<const sslam::half8*>(bl + o);
            const sslam::half8 xl = *reinterpret_cast<const sslam::half8*>(bl + o + PLH);
#pragma unroll
            for (int r = 0; r < 16; ++r) { c1[r] = 0.0f; c2[r] = 0.0f; }
            c1 = sslam::mfma16(ah[9], xh, c1);
            c2 = sslam::mfma16(ah[9], xl, c2);
            c2 = sslam::mfma16(*reinterpret_cast<const sslam::half8*>(wl + 9 * 64 * 8), xh, c2);
            if (live) {
                float* irow = idn + (size_t)y * W + x0;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    at_b(irow, lo + (8 * (r / 4) + r % 4) * HWb) = (c1[r] + c2[r] * sslam::SPLIT_INV) + aff[64 + acc_row(r, lane)];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        stash_row(PH, y + 2);
        __syncthreads();
    };
    for (int y = yb; y < ye; y += 3) {
        step(std::integral_constant<int, 0>{}, y);
        if (y + 1 >= ye) break;
        step(std::integral_constant<int, 1>{}, y + 1);
        if (y + 2 >= ye) break;
        step(std::integral_constant<int, 2>{}, y + 2);
    }
}


template <int RPW>
__global__ __launch_bounds__(256) void al_conv32_h_kernel(const _Float16* __restrict__ in /* hi plane [H][W][32]; lo `plane` halves behind */,
                                                          float* __restrict__ out, int H, int W, const _Float16* __restrict__ wf,
                                                          const float* __restrict__ alpha, const float* __restrict__ beta,
                                                          const float* __restrict__ resid, size_t fs) {
    in = fsh(in, blockIdx.z, fs); out = fsh(out, blockIdx.z, fs); resid = fsh(resid, blockIdx.z, fs);
    constexpr int CTH = 4 * RPW, TH = CTH + 2, TW = CT_W + 2;
    __shared__ __attribute__((aligned(16))) _Float16 tile[2][TH * TW * C32_PS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int x0 = blockIdx.x * CT_W, y0 = blockIdx.y * CTH;
    const size_t plane = (size_t)H * W * 32;
    // tile fill: 16-byte chunks, (plane, row, pixel, chunk) with the chunk fastest: 64 contiguous bytes per pixel
    constexpr int CHUNKS = 2 * TH * TW * 4;
#pragma unroll AL_C32_FILL
    for (int idx = t; idx < CHUNKS; idx += 256) {
        const int c4 = idx & 3, pxl = (idx >> 2) % TW, rr = ((idx >> 2) / TW) % TH, pl = (idx >> 2) / (TW * TH);
        const int yy = y0 + rr - 1, xx = x0 + pxl - 1;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (yy >= 0 && yy < H && xx >= 0 && xx < W && !(AL_C32_ABL & 2))
            v = *reinterpret_cast<const uint4*>(in + pl * plane + ((size_t)yy * W + xx) * 32 + 8 * c4);
        *reinterpret_cast<uint4*>(&tile[pl][(rr * TW + pxl) * C32_PS + 8 * c4]) = v;
    }
    __syncthreads();
    const int h = lane >> 5, px = lane & 31;
    f32x16 c1[RPW], c2[RPW];
#pragma unroll
    for (int q = 0; q < RPW; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) { c1[q][r] = 0.0f; c2[q][r] = 0.0f; }
    const _Float16* bh = &tile[0][((RPW * wave) * TW + px) * C32_PS + 8 * h];
    const _Float16* bl = &tile[1][((RPW * wave) * TW + px) * C32_PS + 8 * h];
    const _Float16* af = wf + (h * 32 + px) * 8;               // + ((ks * 2 + plane) * 2) * 32 * 8
    // BN affine and residual of the lane's own outputs: loaded here, in flight under the matrix loop (r04: inside the epilogue
    // every alpha[co] / beta[co] / resid[o] was a load with a full s_waitcnt behind it - 16 serial round trips per wave; the
    // epilogue was 73 of the kernel's 125 us per launch)
    float alr[16], ber[16], rsd[RPW][16];
    {
        const int xq = min(x0 + px, W - 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = acc_row(r, lane);
            alr[r] = alpha[co]; ber[r] = beta[co];
#pragma unroll
            for (int q = 0; q < RPW; ++q) rsd[q][r] = resid[((size_t)co * H + min(y0 + RPW * wave + q, H - 1)) * W + xq];
        }
    }
#pragma unroll
    for (int ks = 0; ks < ((AL_C32_ABL & 1) ? 2 : 18); ++ks) {
        const int tap = ks >> 1, boff = ((tap / 3) * TW + (tap % 3)) * C32_PS + 16 * (ks & 1);
        const sslam::half8 ah = *reinterpret_cast<const sslam::half8*>(af + (size_t)(ks * 2 + 0) * 2 * 32 * 8);
        const sslam::half8 al = *reinterpret_cast<const sslam::half8*>(af + (size_t)(ks * 2 + 1) * 2 * 32 * 8);
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const sslam::half8 xh = *reinterpret_cast<const sslam::half8*>(bh + boff + q * TW * C32_PS);
            const sslam::half8 xl = *reinterpret_cast<const sslam::half8*>(bl + boff + q * TW * C32_PS);
            c1[q] = sslam::mfma16(ah, xh, c1[q]);
            c2[q] = sslam::mfma16(ah, xl, c2[q]);
            c2[q] = sslam::mfma16(al, xh, c2[q]);
        }
    }
    const int x = x0 + px;
    if (x >= W) return;
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int y = y0 + RPW * wave + q;
        if (y >= H) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = acc_row(r, lane);
            const size_t o = ((size_t)co * H + y) * W + x;
            const float acc = c1[q][r] + c2[q][r] * sslam::SPLIT_INV;
#if AL_C32_ABL & 4
            if (acc == 123.456f) out[o] = acc;
#else
            out[o] = selu(fmaf(acc, alr[r], ber[r]) + rsd[q][r]);
#endif
        }
    }
}


template <int CIN, int OC_PP>      // OC_PP: pixels per wave (4 for batches of frames, 1 when one frame has to fill the chip)
__global__ __launch_bounds__(256) void al_offset_conv_kernel(const float* __restrict__ in, float* __restrict__ off,
                                                             int H, int W,
                                                             const float* __restrict__ wt /*[18][CIN*9]*/,
                                                             const float* __restrict__ b, float max_off, size_t fs) {
    // r03: a wave takes OC_PP consecutive pixels, so the 18 weight loads of an iteration serve four pixels (one
    // wave per pixel re-read the whole [18][CIN*9] weight block for every pixel: 1.7 GB of L1 / L2 traffic per launch
    // at F = 8).  A pixel's sums run over the same lanes and k's in the same order as before: bit-identical.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pix0 = (blockIdx.x * 4 + wave) * OC_PP;
    if (pix0 >= H * W) return;
    in = fsh(in, blockIdx.y, fs); off = fsh(off, blockIdx.y, fs);
    float part[OC_PP][18];
#pragma unroll
    for (int p = 0; p < OC_PP; ++p)
#pragma unroll
        for (int o = 0; o < 18; ++o) part[p][o] = 0.0f;
    // fully unrolled (CIN is a template parameter): all loads of an iteration are in flight at once;
    // as a rolled loop every iteration waited out a full memory latency
#pragma unroll
    for (int it = 0; it < (CIN * 9 + 63) / 64; ++it) {
        const int k = lane + 64 * it;
        if (CIN * 9 % 64 != 0 && k >= CIN * 9) break;
        const int ci = k / 9, tap = k % 9;
        float v[OC_PP];
#pragma unroll
        for (int p = 0; p < OC_PP; ++p) {
            const int pix = min(pix0 + p, H * W - 1);
            const int y = pix / W, x = pix % W;
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            v[p] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? in[((size_t)ci * H + yy) * W + xx] : 0.0f;
        }
#pragma unroll
        for (int o = 0; o < 18; ++o) {
            const float wv = wt[o * (CIN * 9) + k];
#pragma unroll
            for (int p = 0; p < OC_PP; ++p) part[p][o] = fmaf(v[p], wv, part[p][o]);
        }
    }
    // 18 sums over the 64 lanes: through LDS, lane (o, third) adds a third of row o, two shuffles
    // finish it (18 butterfly reductions = 108 cross-lane steps dominated the kernel)
    __shared__ float red[4][18][65];
    const int o = lane / 3, th = lane % 3;
#pragma unroll
    for (int p = 0; p < OC_PP; ++p) {
        if (p) __builtin_amdgcn_wave_barrier();      // the previous pixel's reads of the slab are done (one wave: program order)
#pragma unroll
        for (int q = 0; q < 18; ++q) red[wave][q][lane] = part[p][q];
        __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wave's own slab
        __builtin_amdgcn_wave_barrier();
        float v = 0.0f;
        if (lane < 54) {
            const float* r = red[wave][o];
            const int j0 = th * 22, j1 = th == 2 ? 64 : j0 + 22;
            for (int j = j0; j < j1; ++j) v += r[j];
        }
        v += __shfl_down(v, 1) + __shfl_down(v, 2);
        const int pix = pix0 + p;
        if (lane < 54 && th == 0 && pix < H * W) off[(size_t)o * H * W + pix] = fminf(fmaxf(v + b[o], -max_off), max_off);
        __builtin_amdgcn_s_waitcnt(0xc07f);          // the reads above, before the next pixel overwrites the slab
    }
}


__global__ __launch_bounds__(256) void al_dcn_col_kernel(const float* __restrict__ in /* channel-last [pixel][CIN] (r04) */,
                                                         const float* __restrict__ off,
                                                         float* __restrict__ col, int CIN, int H, int W,
                                                         const float* __restrict__ res_in /* channel-last [pixel][RC] */, int RC, size_t fs) {
    in = fsh(in, blockIdx.y, fs); off = fsh(off, blockIdx.y, fs); col = fsh(col, blockIdx.y, fs); res_in = fsh(res_in, blockIdx.y, fs);
    // r04: thread = (pixel, tap slot, channel quad) with the QUAD fastest and the input channel-last: the four corner loads of
    // a lane are 16 bytes of a pixel's channel vector (16 - 32 lanes share a 256 - 512-byte run) and the float4 it writes is
    // the next 16 bytes of the pixel's im2col row - loads and stores of a wave are whole runs.  (r03: pixel fastest over
    // planar input - coalesced gathers, but every lane's store went to another row of `col`: 64 scattered 16-byte pieces per
    // instruction, 39 us per launch.)  Same arithmetic per element: bit-identical rows.
    const int K = CIN * 9, KT = K + RC, HW = H * W, slots = RC ? 10 : 9, CQ = CIN / 4;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW * slots * CQ) return;
    const int cq = i % CQ, tap = (i / CQ) % slots, pix = i / (CQ * slots), c = 4 * cq;
    float* dst = col + (size_t)pix * KT;
    if (tap == 9) {                                   // block input for the 1x1 branch
        if (c < RC) *reinterpret_cast<float4*>(dst + K + c) = *reinterpret_cast<const float4*>(res_in + (size_t)pix * RC + c);
        return;
    }
    // torchvision deform_conv2d bilinear sample
    const int py = pix / W, px = pix % W;
    const float y = (float)(py - 1 + tap / 3) + off[(size_t)(2 * tap) * HW + pix];
    const float x = (float)(px - 1 + tap % 3) + off[(size_t)(2 * tap + 1) * HW + pix];
    float o[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (!(y <= -1.0f || y >= (float)H || x <= -1.0f || x >= (float)W)) {
        const float fy = floorf(y), fx = floorf(x);
        const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
        const float ly = y - fy, lx = x - fx, hy = 1.0f - ly, hx = 1.0f - lx;
        const bool m1 = y0 >= 0 && x0 >= 0, m2 = y0 >= 0 && x1 <= W - 1, m3 = y1 <= H - 1 && x0 >= 0, m4 = y1 <= H - 1 && x1 <= W - 1;
        const int i1 = m1 ? y0 * W + x0 : 0, i2 = m2 ? y0 * W + x1 : 0, i3 = m3 ? y1 * W + x0 : 0, i4 = m4 ? y1 * W + x1 : 0;
        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const float4 q1 = m1 ? *reinterpret_cast<const float4*>(in + (size_t)i1 * CIN + c) : z;
        const float4 q2 = m2 ? *reinterpret_cast<const float4*>(in + (size_t)i2 * CIN + c) : z;
        const float4 q3 = m3 ? *reinterpret_cast<const float4*>(in + (size_t)i3 * CIN + c) : z;
        const float4 q4 = m4 ? *reinterpret_cast<const float4*>(in + (size_t)i4 * CIN + c) : z;
        const float v1[4] = {q1.x, q1.y, q1.z, q1.w}, v2[4] = {q2.x, q2.y, q2.z, q2.w};
        const float v3[4] = {q3.x, q3.y, q3.z, q3.w}, v4[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = w1 * v1[e] + w2 * v2[e] + w3 * v3[e] + w4 * v4[e];
    }
    *reinterpret_cast<float4*>(dst + tap * CIN + c) = make_float4(o[0], o[1], o[2], o[3]);
}


__global__ __launch_bounds__(256) void al_dcn_gemm_kernel(const float* __restrict__ wt /*[COUT][K]*/, int K,
                                                          const float* __restrict__ col /*[HW][K + RC]*/, int ldc,
                                                          int HW, int COUT, float* __restrict__ part /*[KS+1][COUT][HW]*/,
                                                          const float* __restrict__ wdt /*[COUT][RC]*/, int RC, int KS,
                                                          size_t fs) {
    __shared__ GemmSmem<64, 64> sm;
    const int pix0 = blockIdx.x * 64, co0 = blockIdx.y * 64, z = blockIdx.z % KS, fr = blockIdx.z / KS;   // grid z = frame * KS + slice
    col = fsh(col, fr, fs); part = fsh(part, fr, fs);
    const int kper = K / KS, koff = z * kper;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    const int pix = pix0 + wn * 32 + (lane & 31);
    f32x16 acc[1][1];
    {
        GemmA ga{wt + koff, K, wt + koff, K, kper};
        gemm_mainloop<64, 64, 1, 1>(ga, col + koff, ldc, kper, co0, COUT, pix0, HW, sm, acc);
        float* dst = part + (size_t)z * COUT * HW;
        if (pix < HW)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(size_t)(co0 + wm * 32 + acc_row(r, lane)) * HW + pix] = acc[0][0][r];
    }
    if (RC && z == 0) {                               // block-uniform
        GemmA gd{wdt, RC, wdt, RC, RC};
        gemm_mainloop<64, 64, 1, 1>(gd, col + K, ldc, RC, co0, COUT, pix0, HW, sm, acc);
        float* dst = part + (size_t)KS * COUT * HW;
        if (pix < HW)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(size_t)(co0 + wm * 32 + acc_row(r, lane)) * HW + pix] = acc[0][0][r];
    }
}


__global__ __launch_bounds__(256) void al_dcn_epilogue_kernel(const float* __restrict__ part, int KS, int HW, int COUT,
                                                              float* __restrict__ out, const float* __restrict__ alpha,
                                                              const float* __restrict__ beta, int resid,
                                                              const float* __restrict__ bd, size_t fs, float* __restrict__ out_cl) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= COUT * HW) return;
    part = fsh(part, blockIdx.y, fs); out = fsh(out, blockIdx.y, fs); out_cl = fsh0(out_cl, blockIdx.y, fs);
    const int co = i / HW;
    float acc = part[i];
#pragma unroll 6
    for (int z = 1; z < KS; ++z) acc += part[(size_t)z * COUT * HW + i];
    float v = fmaf(acc, alpha[co], beta[co]);
    if (resid) v += part[(size_t)KS * COUT * HW + i] + bd[co];
    v = selu(v);
    out[i] = v;
    if (out_cl) out_cl[(size_t)(i % HW) * COUT + co] = v;       // [pixel][COUT] copy for the next layer's al_dcn_col
}


__global__ void al_transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // src [R][C] -> dst [C][R]
    if (i >= R * C) return;
    dst[(size_t)(i % C) * R + i / C] = src[i];
}


__global__ void al_dcn_wt_kernel(const float* __restrict__ src, float* __restrict__ dst, int CIN, int taps, int COUT) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= CIN * taps * COUT) return;
    const int co = i % COUT, tap = (i / COUT) % taps, ci = i / (COUT * taps);
    dst[(size_t)co * (taps * CIN) + tap * CIN + ci] = src[i];
}


// ------------------------------------------------------------------------ //
//  1a. the same convolution as a vertical SWEEP (r04): a workgroup owns a 32-pixel column strip of 4 NS rows and walks it in
//      NS steps of 4 rows.  The one-tile kernel above runs load -> LDS -> MFMA -> store once per workgroup and the phases of
//      the workgroups of a CU do not overlap (ablations, 16 -> 16 at F = 8: 192 us per launch; without the loads 145, without
//      the MFMA loop 147, without the stores 150).  Here the global loads of step s + 1 are issued into registers BEFORE the
//      MFMA loop of step s and land in the second LDS tile after its stores, the [k][co] weights are staged once per strip.
//      Arithmetic per output: identical (same operands, same accumulation order).  Requires W % 4 == 0.
// ------------------------------------------------------------------------ //
// global -> registers; nothing waits on the data here.  Loads are UNCONDITIONAL from clamped addresses and the validity goes
// into a bit mask applied at stash time: with predicated loads the optimiser folds the pooling adds of the stash into the
// load's block (phi of a constant and a load), and the wait for the data lands in front of the matrix loop.
template <int CIN, int CINP, int POOL, int NI, int NH, int NR>
__device__ __forceinline__ unsigned sweep_load(const float* __restrict__ in, int inH, int inW, int H, int W, int x0, int y0, int t,
                                               float4 (&ri)[NI][NR], float (&rh)[NH][NR]) {
    constexpr int TH = 6, ROWS = CINP * TH;
    unsigned ok = 0;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int idx = min(t + 256 * j, ROWS * 8 - 1), row = idx >> 3, v4 = idx & 7;
        const int c = row / TH, rr = row % TH;
        const int yy = y0 + rr - 1, xx = x0 + 4 * v4;
        ok |= (unsigned)(t + 256 * j < ROWS * 8 && c < CIN && yy >= 0 && yy < H && xx < W) << j;
        const int cc = min(c, CIN - 1), yc = min(max(yy, 0), H - 1), xc = min(xx, W - 4);
        if (POOL == 1) {
            ri[j][0] = *reinterpret_cast<const float4*>(in + ((size_t)cc * inH + yc) * inW + xc);
        } else {
            const float* r0 = in + ((size_t)cc * inH + yc * 2) * inW + xc * 2;
            const float* r1 = r0 + inW;
            ri[j][0] = *reinterpret_cast<const float4*>(r0); ri[j][NR > 1 ? 1 : 0] = *reinterpret_cast<const float4*>(r0 + 4);
            ri[j][NR > 2 ? 2 : 0] = *reinterpret_cast<const float4*>(r1); ri[j][NR > 3 ? 3 : 0] = *reinterpret_cast<const float4*>(r1 + 4);
        }
    }
#pragma unroll
    for (int j = 0; j < NH; ++j) {
        const int idx = min(t + 256 * j, ROWS * 2 - 1), row = idx >> 1, side = idx & 1;
        const int c = row / TH, rr = row % TH;
        const int yy = y0 + rr - 1, xx = side ? x0 + CT_W : x0 - 1;
        ok |= (unsigned)(t + 256 * j < ROWS * 2 && c < CIN && yy >= 0 && yy < H && xx >= 0 && xx < W) << (16 + j);
        const int cc = min(c, CIN - 1), yc = min(max(yy, 0), H - 1), xc = min(max(xx, 0), W - 1);
        if (POOL == 1) {
            rh[j][0] = in[((size_t)cc * inH + yc) * inW + xc];
        } else {
            const float* r0 = in + ((size_t)cc * inH + yc * 2) * inW + xc * 2;
            rh[j][0] = r0[0]; rh[j][NR > 1 ? 1 : 0] = r0[1]; rh[j][NR > 2 ? 2 : 0] = r0[inW]; rh[j][NR > 3 ? 3 : 0] = r0[inW + 1];
        }
    }
    return ok;
}


template <int CINP, int POOL, bool M16, int NI, int NH, int NR>
__device__ __forceinline__ void sweep_stash(float* __restrict__ tl, int t, unsigned ok, const float4 (&ri)[NI][NR], const float (&rh)[NH][NR]) {
    // registers -> LDS (2x2 average in the order (0,0) (0,1) (1,0) (1,1)); invalid positions (outside the map, padding channels) = 0
    constexpr int TH = 6, ROWS = CINP * TH, CT_TW = M16 ? CT_W + 8 : CT_W + 2, IC = M16 ? 4 : 1;
    constexpr int CHS = M16 ? (TH * CT_TW + 63 - 16) / 64 * 64 + 16 : TH * CT_TW;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int idx = t + 256 * j, row = idx >> 3, v4 = idx & 7;
        if (idx >= ROWS * 8) continue;
        const int c = row / TH, rr = row % TH;
        float4 v;
        if (POOL == 1) v = ri[j][0];
        else {
            const float4 a0 = ri[j][0], a1 = ri[j][NR > 1 ? 1 : 0], b0 = ri[j][NR > 2 ? 2 : 0], b1 = ri[j][NR > 3 ? 3 : 0];
            v = make_float4((((a0.x + a0.y) + b0.x) + b0.y) / 4.0f, (((a0.z + a0.w) + b0.z) + b0.w) / 4.0f,
                            (((a1.x + a1.y) + b1.x) + b1.y) / 4.0f, (((a1.z + a1.w) + b1.z) + b1.w) / 4.0f);
        }
        if (!((ok >> j) & 1u)) v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        float* dst = &tl[c * CHS + rr * CT_TW + IC + 4 * v4];
        if constexpr (M16) *reinterpret_cast<float4*>(dst) = v;
        else { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w; }
    }
#pragma unroll
    for (int j = 0; j < NH; ++j) {
        const int idx = t + 256 * j, row = idx >> 1, side = idx & 1;
        if (idx >= ROWS * 2) continue;
        const int c = row / TH, rr = row % TH;
        float v = POOL == 1 ? rh[j][0] : (((rh[j][0] + rh[j][NR > 1 ? 1 : 0]) + rh[j][NR > 2 ? 2 : 0]) + rh[j][NR > 3 ? 3 : 0]) / 4.0f;
        if (!((ok >> (16 + j)) & 1u)) v = 0.0f;
        tl[c * CHS + rr * CT_TW + (side ? IC + CT_W : IC - 1)] = v;
    }
}


// ---- the LDS-tile NMS (r01 - r04; superseded by al_nms_wave_kernel at the end of r04: one wave per tile, rows in registers, 0/1 maps
// as 64-bit words; the two forms gave bit-identical nms maps on scripts/ab_hash_aliked.sh).  The first al_collect_kernel launch
// (threshold pass over the nms map) went with it: the NMS waves append the candidates themselves.
#ifndef AL_NT_W
#define AL_NT_W 32      // 32 x 16 tiles: 30 KB of LDS, five workgroups per CU (64 x 16: 48 KB, three; 11.5 -> 9.6 us per frame)
#endif
#ifndef AL_NT_H
#define AL_NT_H 16
#endif
constexpr int NT_W = AL_NT_W, NT_H = AL_NT_H, NHALO = 10;     // dependency radius 2 + 4 + 4
constexpr int NE_W = NT_W + 2 * NHALO, NE_H = NT_H + 2 * NHALO;
constexpr int NE = NE_H * NE_W;
constexpr int NPASS = (NE + 255) / 256;

// 5x5 max-pool of an LDS tile, separable: rows into `tmp`, then columns (-inf outside the tile,
// which is what F.max_pool2d's implicit padding does at the map border)
__device__ __forceinline__ void pool5_rows(const float* __restrict__ a, float* __restrict__ tmp) {
#pragma unroll
    for (int k = 0; k < NPASS; ++k) {
        const int i = threadIdx.x + 256 * k;
        if (i < NE) {
            const int lx = i % NE_W;
            float m = a[i];
            if (lx >= 1) m = fmaxf(m, a[i - 1]);
            if (lx >= 2) m = fmaxf(m, a[i - 2]);
            if (lx + 1 < NE_W) m = fmaxf(m, a[i + 1]);
            if (lx + 2 < NE_W) m = fmaxf(m, a[i + 2]);
            tmp[i] = m;
        }
    }
}
__device__ __forceinline__ float pool5_col(const float* __restrict__ tmp, int i) {
    const int ly = i / NE_W;
    float m = tmp[i];
    if (ly >= 1) m = fmaxf(m, tmp[i - NE_W]);
    if (ly >= 2) m = fmaxf(m, tmp[i - 2 * NE_W]);
    if (ly + 1 < NE_H) m = fmaxf(m, tmp[i + NE_W]);
    if (ly + 2 < NE_H) m = fmaxf(m, tmp[i + 2 * NE_W]);
    return m;
}

__global__ __launch_bounds__(256) void al_nms_kernel(const float* __restrict__ score, int h, int w,
                                                     float* __restrict__ nms, float* __restrict__ block_sum, size_t fs) {
    score = fsh(score, blockIdx.z, fs); nms = fsh(nms, blockIdx.z, fs); block_sum = fsh(block_sum, blockIdx.z, fs);
    // s: scores (-inf outside the map); m: max_mask (0/1); q: suppressed scores; tmp: row-pooled scratch.
    // Values at the LDS-tile rim are wrong (missing neighbours) but the 10-pixel halo keeps them out
    // of the dependency cone of the central NT_H x NT_W outputs.
    __shared__ float s[NE], m[NE], q[NE], tmp[NE];
    const int x0 = blockIdx.x * NT_W - NHALO, y0 = blockIdx.y * NT_H - NHALO;
#pragma unroll
    for (int k = 0; k < NPASS; ++k) {
        const int i = threadIdx.x + 256 * k;
        if (i < NE) {
            const int yy = y0 + i / NE_W, xx = x0 + i % NE_W;
            s[i] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? score[(size_t)yy * w + xx] : -INFINITY;
        }
    }
    __syncthreads();
    pool5_rows(s, tmp);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NPASS; ++k) {
        const int i = threadIdx.x + 256 * k;
        if (i < NE) m[i] = (s[i] == pool5_col(tmp, i) && s[i] > -INFINITY) ? 1.0f : 0.0f;
    }
    __syncthreads();
    for (int round = 0; round < 2; ++round) {
        pool5_rows(m, tmp);
        __syncthreads();
        bool supp_r[NPASS];
#pragma unroll
        for (int k = 0; k < NPASS; ++k) {
            const int i = threadIdx.x + 256 * k;
            supp_r[k] = false;
            if (i < NE) {
                supp_r[k] = pool5_col(tmp, i) > 0.0f;                 // supp = maxpool(max_mask) > 0
                q[i] = (s[i] == -INFINITY) ? -INFINITY : (supp_r[k] ? 0.0f : s[i]);
            }
        }
        __syncthreads();
        pool5_rows(q, tmp);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NPASS; ++k) {
            const int i = threadIdx.x + 256 * k;
            if (i < NE) {
                const bool newmax = q[i] == pool5_col(tmp, i) && q[i] > -INFINITY;
                if (newmax && !supp_r[k]) m[i] = 1.0f;                 // max_mask |= new_max & ~supp
            }
        }
        __syncthreads();
    }
    float lsum = 0.0f;
    for (int i = threadIdx.x; i < NT_H * NT_W; i += 256) {
        const int ly = i / NT_W + NHALO, lx = i % NT_W + NHALO;
        const int yy = y0 + ly, xx = x0 + lx;
        if (yy < h && xx < w) {
            const float sv = s[ly * NE_W + lx];
            float v = m[ly * NE_W + lx] > 0.0f ? sv : 0.0f;
            if (yy < 2 || xx < 2 || yy >= h - 2 || xx >= w - 2) v = 0.0f;      // border of `radius`
            nms[(size_t)yy * w + xx] = v;
            lsum += sv;
        }
    }
    for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o);
    __shared__ float ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) block_sum[blockIdx.y * gridDim.x + blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}


#endif
