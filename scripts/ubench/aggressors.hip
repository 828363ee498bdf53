// aggressors.hip - r06, profiles/r06_aggregate_rnorm_diagnosis.md: synthetic kernels to run BESIDE the unstable code shape of
// al_aggregate_kernel (scripts/diag_agg_rnorm.py <repeats> 1 <kind>), one property each, to find which one the events need.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o scripts/ubench/libaggr.so scripts/ubench/aggressors.hip
#include <hip/hip_runtime.h>

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_trans(float* out, int iters) {                 // v_exp_f32 only
    float x = threadIdx.x * 1e-3f, acc = 0.0f;
    for (int i = 0; i < iters; ++i) {
        float e;
        asm volatile("v_exp_f32 %0, %1" : "=v"(e) : "v"(x));
        acc += e; x = x * 0.999f + 1e-4f;
    }
    if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters) {                  // matrix pipe only
    half4_t a = {1, 2, 3, 4}, b = {4, 3, 2, 1};
    f32x16_t c = {};
    for (int i = 0; i < iters; ++i) c = __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, c, 0, 0, 0);
    if (c[0] == 123.456f) out[0] = c[3];
}
__global__ __launch_bounds__(256) void k_pk(float* out, int iters) {                    // packed fp32 VALU only
    f32x2_t a = {1.0001f, 0.9999f}, b = {threadIdx.x * 1e-6f, 1e-7f}, c = {0, 0};
    for (int i = 0; i < iters; ++i) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(c) : "v"(a), "v"(b));
    if (c[0] == 123.456f) out[0] = c[1];
}
__global__ __launch_bounds__(256) void k_valu(float* out, int iters) {                  // plain fp32 VALU only
    float a = 1.0001f, b = threadIdx.x * 1e-6f, c = 0;
    for (int i = 0; i < iters; ++i) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(c) : "v"(a), "v"(b));
    if (c == 123.456f) out[0] = c;
}
__global__ __launch_bounds__(256) void k_lds(float* out, int iters) {                   // LDS traffic (48 KB allocated)
    __shared__ float buf[12288];
    for (int i = threadIdx.x; i < 12288; i += 256) buf[i] = i;
    __syncthreads();
    float acc = 0; unsigned j = threadIdx.x;
    for (int i = 0; i < iters; ++i) { acc += buf[j % 12288]; buf[(j * 7 + 3) % 12288] = acc; j = j * 5 + 1; }
    if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_gather(const float* __restrict__ tab, float* out, int iters, unsigned mask) {   // L1 / L2-resident vector loads
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    float acc = 0;
    for (int i = 0; i < iters; ++i) { h = h * 1664525u + 1013904223u; acc += tab[(h >> 8) & mask]; }
    if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_store(float* dst, int iters, unsigned mask) {   // vector stores into an L2-resident window
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    for (int i = 0; i < iters; ++i) { h = h * 1664525u + 1013904223u; dst[(h >> 8) & mask] = (float)i; }
}
__global__ __launch_bounds__(256) void k_scalar(const int* __restrict__ tab, float* out, int iters) {                      // scalar loads (wave-uniform)
    int acc = 0;
    for (int i = 0; i < iters; ++i) acc += tab[(blockIdx.x * 131 + i * 17) & 4095];
    if (acc == 123456789) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_ldsdma(const float* __restrict__ tab, float* out, int iters, unsigned mask) {   // LDS-DMA: global_load_lds_dwordx4
    extern __shared__ float dyn[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned h = (blockIdx.x * 4 + wave) * 2654435761u;
    float acc = 0;
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        const float* src = tab + ((((h >> 8) & mask) & ~255u) + lane * 4);            // the wave's 1 KiB piece, 16 B per lane
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dyn + wave * 2048 + (i & 7) * 256), 16, 0, 0);
        if ((i & 7) == 7) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc += dyn[wave * 2048 + lane];
        }
    }
    if (acc == 123.456f) out[0] = acc;
}


// r06, section 5 of the diagnosis: ONE instruction each, in the very form lg_attention_p_kernel (the strongest trigger) uses it - which
// instruction of another wave on the same SIMD makes `v_pk_mul_f32 ... op_sel:[0,1]` (scripts/ubench/pk_probe.hip) return 0.0 in
// its last 16 lanes?  Registers are named outright (nothing is computed: the operands are whatever the registers hold).
#define INSN_KERNEL(NAME, TEXT)                                                                                            \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters) {                                                   \
        for (int i = 0; i < iters; ++i)                                                                                    \
            asm volatile(TEXT "\n\t" TEXT "\n\t" TEXT "\n\t" TEXT ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "s20", "s21");                                             \
    }
INSN_KERNEL(k_i_mixlo, "v_fma_mixlo_f16 v20, v10, -1.0, v11 op_sel_hi:[1,0,0]")
INSN_KERNEL(k_i_mixhi, "v_fma_mixhi_f16 v20, v10, -1.0, v11 op_sel:[1,0,0] op_sel_hi:[1,0,0]")
INSN_KERNEL(k_i_sdwa, "v_cvt_f32_f16_sdwa v20, v10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
INSN_KERNEL(k_i_cvtpk, "v_cvt_pk_f16_f32 v20, v10, v11")
INSN_KERNEL(k_i_perm, "v_perm_b32 v20, v10, v11, v12")
INSN_KERNEL(k_i_permswap, "v_permlane32_swap_b32_e32 v20, v21")
INSN_KERNEL(k_i_bitop3, "v_bitop3_b32 v20, v10, v11, 7 bitop3:0x78")
INSN_KERNEL(k_i_mov64, "v_mov_b64_e32 v[20:21], v[10:11]")
INSN_KERNEL(k_i_max3, "v_max3_f32 v20, v10, v11, v12")
INSN_KERNEL(k_i_pkmul_hi10, "v_pk_mul_f32 v[20:21], v[10:11], v[12:13] op_sel_hi:[1,0]")
INSN_KERNEL(k_i_pkmul_01, "v_pk_mul_f32 v[20:21], v[10:11], v[12:13] op_sel:[0,1] op_sel_hi:[1,0]")
INSN_KERNEL(k_i_pkfma_hi101, "v_pk_fma_f32 v[20:21], v[10:11], v[12:13], v[14:15] op_sel_hi:[1,0,1]")
INSN_KERNEL(k_i_mfma16, "v_mfma_f32_32x32x16_f16 v[32:47], v[10:13], v[14:17], v[32:47]")
INSN_KERNEL(k_i_cvtf16, "v_cvt_f32_f16_e32 v20, v10")
INSN_KERNEL(k_i_fmamk, "v_fmamk_f32 v20, v10, 0x3a000000, v11")
INSN_KERNEL(k_i_bfi, "v_bfi_b32 v20, v10, v11, v12")
INSN_KERNEL(k_i_cmpabs, "v_cmp_nlg_f32_e64 s[20:21], |v10|, v11")
INSN_KERNEL(k_i_lshladd64, "v_lshl_add_u64 v[20:21], v[10:11], 2, v[12:13]")
INSN_KERNEL(k_i_exp, "v_exp_f32_e32 v20, v10")
INSN_KERNEL(k_i_pkmul, "v_pk_mul_f32 v[20:21], v[10:11], v[12:13]")
INSN_KERNEL(k_i_cvt_f16, "v_cvt_f16_f32_e32 v20, v10")
INSN_KERNEL(k_i_shl64, "v_lshlrev_b64 v[20:21], 2, v[10:11]")
// (which MFMA: the gfx950 double-K forms, the older ones, other types)
INSN_KERNEL(k_i_mfma_16x16x32_f16, "v_mfma_f32_16x16x32_f16 v[32:35], v[10:13], v[14:17], v[32:35]")
INSN_KERNEL(k_i_mfma_32x32x16_bf16, "v_mfma_f32_32x32x16_bf16 v[32:47], v[10:13], v[14:17], v[32:47]")
INSN_KERNEL(k_i_mfma_16x16x32_bf16, "v_mfma_f32_16x16x32_bf16 v[32:35], v[10:13], v[14:17], v[32:35]")
INSN_KERNEL(k_i_mfma_32x32x8_f16, "v_mfma_f32_32x32x8_f16 v[32:47], v[10:11], v[14:15], v[32:47]")
INSN_KERNEL(k_i_mfma_16x16x16_f16, "v_mfma_f32_16x16x16_f16 v[32:35], v[10:11], v[14:15], v[32:35]")
INSN_KERNEL(k_i_mfma_32x32x64_f8f6f4, "v_mfma_f32_32x32x64_f8f6f4 v[32:47], v[10:17], v[18:25], v[32:47]")
INSN_KERNEL(k_i_mfma_32x32x2_f32, "v_mfma_f32_32x32x2_f32 v[32:47], v10, v11, v[32:47]")
INSN_KERNEL(k_i_mfma_32x32x32_i8, "v_mfma_i32_32x32x32_i8 v[32:47], v[10:13], v[14:17], v[32:47]")
INSN_KERNEL(k_i_mfma_32x32x16_fp8, "v_mfma_f32_32x32x16_fp8_fp8 v[32:47], v[10:11], v[14:15], v[32:47]")
INSN_KERNEL(k_i_mfma_16x16x128_f8f6f4, "v_mfma_f32_16x16x128_f8f6f4 v[32:35], v[10:17], v[18:25], v[32:35]")
__global__ __launch_bounds__(256) void k_i_all(float* out, int iters) {
    for (int i = 0; i < iters; ++i)
        asm volatile("v_fma_mixlo_f16 v20, v10, -1.0, v11 op_sel_hi:[1,0,0]" "\n\t"
                     "v_fma_mixhi_f16 v20, v10, -1.0, v11 op_sel:[1,0,0] op_sel_hi:[1,0,0]" "\n\t"
                     "v_cvt_f32_f16_sdwa v20, v10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" "\n\t"
                     "v_cvt_pk_f16_f32 v20, v10, v11" "\n\t"
                     "v_perm_b32 v20, v10, v11, v12" "\n\t"
                     "v_permlane32_swap_b32_e32 v20, v21" "\n\t"
                     "v_bitop3_b32 v20, v10, v11, 7 bitop3:0x78" "\n\t"
                     "v_mov_b64_e32 v[20:21], v[10:11]" "\n\t"
                     "v_max3_f32 v20, v10, v11, v12" "\n\t"
                     "v_pk_mul_f32 v[20:21], v[10:11], v[12:13] op_sel_hi:[1,0]" "\n\t"
                     "v_pk_mul_f32 v[20:21], v[10:11], v[12:13] op_sel:[0,1] op_sel_hi:[1,0]" "\n\t"
                     "v_pk_fma_f32 v[20:21], v[10:11], v[12:13], v[14:15] op_sel_hi:[1,0,1]" "\n\t"
                     "v_mfma_f32_32x32x16_f16 v[32:47], v[10:13], v[14:17], v[32:47]" "\n\t"
                     "v_cvt_f32_f16_e32 v20, v10" "\n\t"
                     "v_fmamk_f32 v20, v10, 0x3a000000, v11" "\n\t"
                     "v_bfi_b32 v20, v10, v11, v12" "\n\t"
                     "v_cmp_nlg_f32_e64 s[20:21], |v10|, v11" "\n\t"
                     "v_lshl_add_u64 v[20:21], v[10:11], 2, v[12:13]" "\n\t"
                     "v_exp_f32_e32 v20, v10" "\n\t"
                     "v_pk_mul_f32 v[20:21], v[10:11], v[12:13]" "\n\t"
                     "v_cvt_f16_f32_e32 v20, v10" "\n\t"
                     "v_lshlrev_b64 v[20:21], 2, v[10:11]" ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "s20", "s21");
}

static float* g_buf = nullptr;
extern "C" int aggr_launch(int kind, void* stream, int blocks, int iters) {
    hipStream_t s = (hipStream_t)stream;
    if (!g_buf) { if (hipMalloc(&g_buf, 64 << 20) != hipSuccess) return 1; (void)hipMemset(g_buf, 0, 64 << 20); }
    float* out = g_buf + (60 << 18);
    switch (kind) {
        case 1: hipLaunchKernelGGL(k_trans, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 2: hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 3: hipLaunchKernelGGL(k_pk, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 4: hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 5: hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 6: hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, s, g_buf, out, iters, (1u << 20) - 1); break;   // 4 MB window
        case 7: hipLaunchKernelGGL(k_store, dim3(blocks), dim3(256), 0, s, g_buf, iters, (1u << 20) - 1); break;
        case 8: hipLaunchKernelGGL(k_scalar, dim3(blocks), dim3(256), 0, s, (const int*)g_buf, out, iters); break;
        case 9: hipLaunchKernelGGL(k_ldsdma, dim3(blocks), dim3(256), 32768, s, g_buf, out, iters, (1u << 20) - 1); break;
        case 10: hipLaunchKernelGGL(k_i_mixlo, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 11: hipLaunchKernelGGL(k_i_mixhi, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 12: hipLaunchKernelGGL(k_i_sdwa, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 13: hipLaunchKernelGGL(k_i_cvtpk, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 14: hipLaunchKernelGGL(k_i_perm, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 15: hipLaunchKernelGGL(k_i_permswap, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 16: hipLaunchKernelGGL(k_i_bitop3, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 17: hipLaunchKernelGGL(k_i_mov64, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 18: hipLaunchKernelGGL(k_i_max3, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 19: hipLaunchKernelGGL(k_i_pkmul_hi10, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 20: hipLaunchKernelGGL(k_i_pkmul_01, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 21: hipLaunchKernelGGL(k_i_pkfma_hi101, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 22: hipLaunchKernelGGL(k_i_mfma16, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 23: hipLaunchKernelGGL(k_i_cvtf16, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 24: hipLaunchKernelGGL(k_i_fmamk, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 25: hipLaunchKernelGGL(k_i_bfi, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 26: hipLaunchKernelGGL(k_i_cmpabs, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 27: hipLaunchKernelGGL(k_i_lshladd64, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 28: hipLaunchKernelGGL(k_i_exp, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 30: hipLaunchKernelGGL(k_i_pkmul, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 31: hipLaunchKernelGGL(k_i_cvt_f16, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 32: hipLaunchKernelGGL(k_i_shl64, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 50: hipLaunchKernelGGL(k_i_mfma_16x16x32_f16, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 51: hipLaunchKernelGGL(k_i_mfma_32x32x16_bf16, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 52: hipLaunchKernelGGL(k_i_mfma_16x16x32_bf16, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 53: hipLaunchKernelGGL(k_i_mfma_32x32x8_f16, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 54: hipLaunchKernelGGL(k_i_mfma_16x16x16_f16, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 55: hipLaunchKernelGGL(k_i_mfma_32x32x64_f8f6f4, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 56: hipLaunchKernelGGL(k_i_mfma_32x32x2_f32, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 57: hipLaunchKernelGGL(k_i_mfma_32x32x32_i8, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 58: hipLaunchKernelGGL(k_i_mfma_32x32x16_fp8, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 59: hipLaunchKernelGGL(k_i_mfma_16x16x128_f8f6f4, dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 40: hipLaunchKernelGGL(k_i_all, dim3(blocks), dim3(256), 0, s, out, iters); break;
        default: return 2;
    }
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
