// insn_probe.hip - any ONE instruction as a victim: is its result the same beside another queue's MFMA kernels as alone?
//
// pk_probe.hip checks packed fp32 against expected values; this file asks the same of every OTHER modifier-bearing instruction form
// the product's ISA contains (mixed-precision FMAs with op_sel, SDWA, DPP, v_pk_mov_b32 ...: `llvm-objdump -d` of the library,
// profiles/r06_aggregate_rnorm_diagnosis.md section 5) without knowing what each computes: every lane runs the instruction on a
// deterministic operand sequence and folds the results into a hash; the hashes of a launch are compared on the device with those
// of the first launch, taken alone.  Same C entry points as agg_victim.hip (scripts/agg_victim_run.py, scripts/insn_probe_run.sh);
// MODE = index into FORMS below (victim_mode_text(m)).
//   hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -shared -o scripts/ubench/libinsnprobe.so scripts/ubench/insn_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>

namespace {

// operands: v[10:11] = A, v[12:13] = B, v[14:15] = C (every 16-bit half a normal f16 in [0.5, 2), the 32-bit words normal floats);
// result: v[20:21] (zeroed in front of the instruction)
#define FORMS(X)                                                                                          \
    X(0, "v_fma_mixlo_f16 v20, v10, v12, v14 op_sel_hi:[1,0,0]")                                          \
    X(1, "v_fma_mixhi_f16 v20, v10, v12, v14 op_sel:[1,0,0] op_sel_hi:[1,0,0]")                            \
    X(2, "v_fma_mix_f32 v20, v10, 1.0, v14 op_sel_hi:[1,0,0]")                                            \
    X(3, "v_fma_mix_f32 v20, v10, 1.0, v14 op_sel:[1,0,0] op_sel_hi:[1,0,0]")                              \
    X(4, "v_fma_mix_f32 v20, v10, v12, -v14 op_sel_hi:[0,0,1]")                                           \
    X(5, "v_cvt_f32_f16_sdwa v20, v10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")               \
    X(6, "v_add_u32_sdwa v20, v10, v12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0") \
    X(7, "v_add_u32_sdwa v20, v10, v12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1") \
    X(8, "v_mul_u32_u24_sdwa v20, v10, v12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD") \
    X(9, "v_mul_u32_u24_sdwa v20, v10, v12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD") \
    X(10, "v_lshlrev_b32_sdwa v20, v10, v12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0") \
    X(11, "v_or_b32_sdwa v20, v10, v12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1") \
    X(12, "v_pk_mov_b32 v[20:21], v[10:11], v[12:13] op_sel:[1,0]")                                        \
    X(13, "v_mov_b32_dpp v20, v10 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")                    \
    X(14, "v_mov_b32_dpp v20, v10 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")                    \
    X(15, "v_mov_b32_dpp v20, v10 row_shr:8 row_mask:0xf bank_mask:0xf")                                  \
    X(16, "v_mov_b32_dpp v20, v10 row_bcast:15 row_mask:0xf bank_mask:0xf")                               \
    X(17, "v_add_f32_dpp v20, v10, v12 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")                \
    X(18, "v_cvt_pk_f16_f32 v20, v10, v12")                                                               \
    X(19, "v_pk_fma_f32 v[20:21], v[10:11], v[12:13], v[14:15] op_sel_hi:[0,1,1]")                         \
    X(20, "v_pk_mul_f32 v[20:21], v[10:11], v[12:13] op_sel_hi:[0,1]")                                     \
    X(21, "v_pk_mul_f32 v[20:21], v[10:11], v[12:13] op_sel:[0,1] op_sel_hi:[1,0]")                        \
    X(22, "v_pk_mul_f16 v20, v10, v12 op_sel:[0,1] op_sel_hi:[1,0]")                                      \
    X(23, "v_pk_fma_f16 v20, v10, v12, v14 op_sel:[0,1,0] op_sel_hi:[1,0,1]")                              \
    X(24, "v_pk_add_f16 v20, v10, v12 op_sel:[0,1] op_sel_hi:[1,0]")                                      \
    X(25, "v_dot2c_f32_f16 v20, v10, v12")                                                                \
    X(26, "v_pk_mul_lo_u16 v20, v10, v12 op_sel:[0,1] op_sel_hi:[1,0]")                                   \
    X(27, "v_perm_b32 v20, v10, v12, v14")                                                                \
    X(28, "v_permlane32_swap_b32_e32 v20, v21")
constexpr int N_MODES = 29;

struct Probe { unsigned *hash, *ref, *bad; int mode; bool have_ref; };
constexpr int BLOCKS = 2048, THREADS = 256, N = BLOCKS * THREADS;

template <int MODE>
__global__ __launch_bounds__(256) void insn_probe_kernel(unsigned* __restrict__ out, int iters) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned r = tid * 2654435761u + 12345u, h = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned w[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {                       // both 16-bit halves 0x3800 .. 0x3fff: f16 in [0.5, 2); the word a normal float
            r = r * 1664525u + 1013904223u;
            w[k] = 0x38003800u | ((r >> 8) & 0x07ff07ffu);
        }
        unsigned d0, d1;
#define X(M, TEXT)                                                                                                              \
        if (MODE == M)                                                                                                          \
            asm volatile("v_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\tv_mov_b32 v12, %4\n\tv_mov_b32 v13, %5\n\tv_mov_b32 v14, %6\n\t"   \
                         "v_mov_b32 v15, %7\n\tv_mov_b32 v20, 0\n\tv_mov_b32 v21, 0\n\ts_nop 1\n\t" TEXT "\n\ts_nop 1\n\t"           \
                         "v_mov_b32 %0, v20\n\tv_mov_b32 %1, v21"                                                                \
                         : "=v"(d0), "=v"(d1) : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5])                   \
                         : "v10", "v11", "v12", "v13", "v14", "v15", "v20", "v21");
        FORMS(X)
#undef X
        h = (h * 31u + d0) ^ (d1 * 2246822519u);
    }
    out[tid] = h;
}

__global__ void insn_compare(const unsigned* __restrict__ a, const unsigned* __restrict__ b, int n, unsigned* __restrict__ bad, unsigned launch) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (a[i] != b[i]) {
            const unsigned k = atomicAdd(&bad[0], 1u);
            if (k < 15) { bad[4 + 4 * k] = (unsigned)i; bad[5 + 4 * k] = a[i]; bad[6 + 4 * k] = b[i]; bad[7 + 4 * k] = launch; }
        }
}

}  // namespace

extern "C" {

void* victim_create(int, int, int mode, unsigned) {
    if (mode < 0 || mode >= N_MODES) return nullptr;
    Probe* p = new Probe();
    p->mode = mode; p->have_ref = false;
    if (hipMalloc(&p->hash, N * 4) != hipSuccess || hipMalloc(&p->ref, N * 4) != hipSuccess || hipMalloc(&p->bad, 64 * 4) != hipSuccess) return nullptr;
    (void)hipMemset(p->bad, 0, 64 * 4);
    return p;
}

// `iters` launches (2048 workgroups x 256 threads x 256 instructions), each compared with the first launch ever made
int victim_run(void* h, void* stream, int iters, int) {
    Probe* p = (Probe*)h;
    hipStream_t s = (hipStream_t)stream;
    static unsigned launch = 0;
    for (int i = 0; i < iters; ++i) {
        ++launch;
        switch (p->mode) {
#define X(M, TEXT) case M: hipLaunchKernelGGL(insn_probe_kernel<M>, dim3(BLOCKS), dim3(THREADS), 0, s, p->hash, 256); break;
            FORMS(X)
#undef X
        }
        if (!p->have_ref) { (void)hipMemcpyAsync(p->ref, p->hash, N * 4, hipMemcpyDeviceToDevice, s); p->have_ref = true; }
        else hipLaunchKernelGGL(insn_compare, dim3(512), dim3(256), 0, s, p->hash, p->ref, N, p->bad, launch);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int victim_poll(void* h, void* stream, unsigned* out64) {
    Probe* p = (Probe*)h;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
    return hipMemcpy(out64, p->bad, 64 * 4, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}

const char* victim_mode_text(int m) {
    switch (m) {
#define X(M, TEXT) case M: return TEXT;
        FORMS(X)
#undef X
    }
    return nullptr;
}

void victim_destroy(void* h) {
    Probe* p = (Probe*)h;
    (void)hipFree(p->hash); (void)hipFree(p->ref); (void)hipFree(p->bad);
    delete p;
}

}  // extern "C"
