// pk_opsel_hazard_repro.hip - stand-alone reproducer (no library, no Python) of the fault that
// profiles/r06_aggregate_rnorm_diagnosis.md section 5 describes, for whoever owns the hardware:
//
//   on gfx950 (MI355X, ROCm 7.2.0), `v_pk_mul_f32 D, A, B op_sel:[0,1]` (low result = A.lo x B.hi; likewise v_pk_add_f32 and
//   v_pk_fma_f32, any op_sel_hi, A != B) returns its LOW half computed with B.hi read as 0.0 in lanes 48..63, about once per 10^4
//   executions, while ANOTHER wave (of another kernel, or of the same one) issues `v_mfma_f32_16x16x32_f16` (or any MFMA with
//   128-bit or wider A / B operands) on the same SIMD.  The mirrored form op_sel:[1,0], the unswizzled form, and the same kernels run one after the other never fail.
//
// Two streams: stream 0 loops a kernel of nothing but the MFMA, stream 1 launches a kernel that executes the packed multiply on
// lane-dependent operands in [0.5, 1.5) and checks both halves against single-width multiplies.  Output: wrong results per form,
// with and without the MFMA kernel running, the lanes they fell in, and - with the MFMA kernel's waves confined to one SIMD of
// every compute unit - the SIMD of the waves they fell in (only that one).
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_opsel_hazard_repro scripts/ubench/pk_opsel_hazard_repro.hip && /tmp/pk_opsel_hazard_repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int simd_id() {           // HW_ID bits 5:4: which of the compute unit's four SIMDs this wave runs on
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    return (hw >> 4) & 3;
}

// the aggressor: 4 x `iters` MFMAs per wave on whatever the registers hold; only_simd >= 0: waves that find themselves on another
// SIMD leave at once, so that MFMAs issue on that one SIMD of every compute unit only
__global__ __launch_bounds__(256) void mfma_kernel(int iters, int only_simd) {
    if (only_simd >= 0 && simd_id() != only_simd) return;
    for (int i = 0; i < iters; ++i)
        asm volatile("v_mfma_f32_16x16x32_f16 v[32:35], v[10:13], v[14:17], v[32:35]\n\t"
                     "v_mfma_f32_16x16x32_f16 v[32:35], v[10:13], v[14:17], v[32:35]\n\t"
                     "v_mfma_f32_16x16x32_f16 v[32:35], v[10:13], v[14:17], v[32:35]\n\t"
                     "v_mfma_f32_16x16x32_f16 v[32:35], v[10:13], v[14:17], v[32:35]"
                     ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v32", "v33", "v34", "v35");
}

// the victim.  FORM 0: op_sel:[0,1] op_sel_hi:[1,0] (fails); 1: op_sel:[1,0] op_sel_hi:[0,1] (the mirror image: never); 2: no swizzle
// counts[0] wrong low halves, [1] wrong high halves, [2 + g] wrong results in 16-lane group g, [8] results that were exactly 0.0,
// [10 + s] wrong results of waves on SIMD s
template <int FORM>
__global__ __launch_bounds__(256) void pk_kernel(unsigned* __restrict__ counts, int iters) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 a = {0.5f + (float)(tid & 1023) * (1.0f / 1024.0f), 0.5f + (float)((tid * 7u) & 1023) * (1.0f / 1024.0f)};
    f32x2 b = {0.5f + (float)((tid * 13u) & 1023) * (1.0f / 1024.0f), 0.5f + (float)((tid * 29u) & 1023) * (1.0f / 1024.0f)};
    unsigned nlo = 0, nhi = 0, nzero = 0;
    for (int it = 0; it < iters; ++it) {
        f32x2 d;
        float elo, ehi;
        if (FORM == 0) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
            asm volatile("s_nop 1\n\tv_mul_f32 %0, %1, %2" : "=v"(elo) : "v"(a.x), "v"(b.y));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ehi) : "v"(a.y), "v"(b.x));
        } else if (FORM == 1) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
            asm volatile("s_nop 1\n\tv_mul_f32 %0, %1, %2" : "=v"(elo) : "v"(a.y), "v"(b.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ehi) : "v"(a.x), "v"(b.y));
        } else {
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
            asm volatile("s_nop 1\n\tv_mul_f32 %0, %1, %2" : "=v"(elo) : "v"(a.x), "v"(b.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ehi) : "v"(a.y), "v"(b.y));
        }
        asm volatile("s_nop 1" : "+v"(d));
        if (d.x != elo) { ++nlo; nzero += d.x == 0.0f; }
        if (d.y != ehi) ++nhi;
        a.x += 1.0f / 4096.0f; if (a.x >= 1.5f) a.x -= 1.0f;
        b.y += 1.0f / 8192.0f; if (b.y >= 1.5f) b.y -= 1.0f;
    }
    if (nlo) { atomicAdd(&counts[0], nlo); atomicAdd(&counts[2 + ((threadIdx.x & 63) >> 4)], nlo); atomicAdd(&counts[8], nzero); atomicAdd(&counts[10 + simd_id()], nlo); }
    if (nhi) { atomicAdd(&counts[1], nhi); atomicAdd(&counts[2 + ((threadIdx.x & 63) >> 4)], nhi); }
}

// does it take ANOTHER kernel?  One kernel, run alone.  MIX 0: every wave alternates four MFMAs and one checked packed multiply;
// MIX 1: the odd waves of a workgroup issue MFMAs, the even ones the packed multiply
template <int MIX>
__global__ __launch_bounds__(256) void mixed_kernel(unsigned* __restrict__ counts, int iters) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    const bool mfma_wave = MIX == 1 && ((threadIdx.x >> 6) & 1);
    f32x2 a = {0.5f + (float)(tid & 1023) * (1.0f / 1024.0f), 0.5f + (float)((tid * 7u) & 1023) * (1.0f / 1024.0f)};
    f32x2 b = {0.5f + (float)((tid * 13u) & 1023) * (1.0f / 1024.0f), 0.5f + (float)((tid * 29u) & 1023) * (1.0f / 1024.0f)};
    unsigned nlo = 0, nhi = 0, nzero = 0;
    for (int it = 0; it < iters; ++it) {
        if (MIX == 0 || mfma_wave)
            asm volatile("v_mfma_f32_16x16x32_f16 v[32:35], v[10:13], v[14:17], v[32:35]\n\t"
                         "v_mfma_f32_16x16x32_f16 v[32:35], v[10:13], v[14:17], v[32:35]\n\t"
                         "v_mfma_f32_16x16x32_f16 v[32:35], v[10:13], v[14:17], v[32:35]\n\t"
                         "v_mfma_f32_16x16x32_f16 v[32:35], v[10:13], v[14:17], v[32:35]"
                         ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v32", "v33", "v34", "v35");
        if (mfma_wave) continue;
        f32x2 d;
        float elo, ehi;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
        asm volatile("s_nop 1\n\tv_mul_f32 %0, %1, %2" : "=v"(elo) : "v"(a.x), "v"(b.y));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ehi) : "v"(a.y), "v"(b.x));
        asm volatile("s_nop 1" : "+v"(d));
        if (d.x != elo) { ++nlo; nzero += d.x == 0.0f; }
        if (d.y != ehi) ++nhi;
        a.x += 1.0f / 4096.0f; if (a.x >= 1.5f) a.x -= 1.0f;
        b.y += 1.0f / 8192.0f; if (b.y >= 1.5f) b.y -= 1.0f;
    }
    if (nlo) { atomicAdd(&counts[0], nlo); atomicAdd(&counts[2 + ((threadIdx.x & 63) >> 4)], nlo); atomicAdd(&counts[8], nzero); atomicAdd(&counts[10 + simd_id()], nlo); }
    if (nhi) atomicAdd(&counts[1], nhi);
}

template <int MIX> void run_mixed(const char* name, hipStream_t s, unsigned* d_counts) {
    CHECK(hipMemset(d_counts, 0, 64));
    for (int r = 0; r < 200; ++r) hipLaunchKernelGGL(mixed_kernel<MIX>, dim3(2048), dim3(256), 0, s, d_counts, 256);
    CHECK(hipDeviceSynchronize());
    unsigned c[16];
    CHECK(hipMemcpy(c, d_counts, 64, hipMemcpyDeviceToHost));
    printf("%-86s wrong low halves %10u, wrong high halves %u", name, c[0], c[1]);
    if (c[0] + c[1]) printf(";  by 16-lane group: %u %u %u %u;  exactly 0.0: %u", c[2], c[3], c[4], c[5], c[8]);
    printf("\n");
}

template <int FORM> void run(const char* name, bool with_mfma, hipStream_t s_mfma, hipStream_t s_pk, unsigned* d_counts, int only_simd = -1) {
    CHECK(hipMemset(d_counts, 0, 64));
    const int rounds = 100;
    for (int r = 0; r < rounds; ++r) {
        if (with_mfma) hipLaunchKernelGGL(mfma_kernel, dim3(only_simd >= 0 ? 4096 : 1024), dim3(256), 0, s_mfma, 4000, only_simd);
        for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(pk_kernel<FORM>, dim3(2048), dim3(256), 0, s_pk, d_counts, 256);
        if (r % 10 == 9) { CHECK(hipStreamSynchronize(s_pk)); CHECK(hipStreamSynchronize(s_mfma)); }
    }
    CHECK(hipDeviceSynchronize());
    unsigned c[16];
    CHECK(hipMemcpy(c, d_counts, 64, hipMemcpyDeviceToHost));
    const double execs = (double)rounds * 10 * 2048 * 4 * 256;
    char where[64];
    if (only_simd >= 0) snprintf(where, sizeof where, "MFMAs on SIMD %d only", only_simd);
    printf("%-46s %-22s wrong low halves %10u, wrong high halves %u  (of %.1e wave executions)", name,
           !with_mfma ? "alone" : only_simd >= 0 ? where : "beside the MFMA kernel", c[0], c[1], execs);
    if (c[0] + c[1]) printf(";  by 16-lane group: %u %u %u %u;  exactly 0.0: %u;  by the victim wave's SIMD: %u %u %u %u", c[2], c[3], c[4], c[5], c[8],
                            c[10], c[11], c[12], c[13]);
    printf("\n");
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    printf("device: %s (%s), %d CUs\n", p.name, p.gcnArchName, p.multiProcessorCount);
    hipStream_t s0, s1;
    CHECK(hipStreamCreate(&s0)); CHECK(hipStreamCreate(&s1));
    unsigned* d_counts;
    CHECK(hipMalloc(&d_counts, 64));
    for (int with = 0; with < 2; ++with) {
        run<0>("v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]", with, s0, s1, d_counts);
        run<1>("v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1]", with, s0, s1, d_counts);
        run<2>("v_pk_mul_f32 (no swizzle)", with, s0, s1, d_counts);
    }
    // is it the same SIMD?  The MFMA kernel's waves stay on ONE SIMD of every compute unit; the victim's waves say where they ran
    for (int simd = 0; simd < 4; ++simd) run<0>("v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]", true, s0, s1, d_counts, simd);
    // one kernel alone
    run_mixed<0>("ONE kernel alone: every wave alternates 4 MFMAs and the packed multiply", s1, d_counts);
    run_mixed<1>("ONE kernel alone: odd waves of a workgroup issue MFMAs, even waves the packed multiply", s1, d_counts);
    return 0;
}
