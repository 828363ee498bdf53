// insn_probe.hip - any ONE instruction as a victim: is its result the same beside another queue's MFMA kernels as alone?
//
// pk_probe.hip checks packed fp32 against expected values; this file asks the same of EVERY VALU instruction form the product's ISA
// contains (467 forms of 59 139 instructions: f64 arithmetic, 64-bit integer, mixed-precision FMAs with op_sel, SDWA, DPP, compares
// ...: scripts/gen_insn_probe_forms.py, profiles/r06_aggregate_rnorm_diagnosis.md section 5) without knowing what each computes: every lane runs the instruction on a
// deterministic operand sequence and folds the results into a hash; the hashes of a launch are compared on the device with those
// of the first launch, taken alone.  Same C entry points as agg_victim.hip (scripts/agg_victim_run.py, scripts/insn_probe_run.sh);
// MODE = index into PRODUCT_FORMS, then EXTRA_FORMS (victim_mode_text(m), victim_modes()).
//   hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -shared -o scripts/ubench/libinsnprobe.so scripts/ubench/insn_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>

namespace {

// operands: v[10:13] = A, v[14:17] = B, v[18:21] = C (every 16-bit half a normal f16 in [0.5, 2), the 32-bit words normal floats,
// the 64-bit pairs normal doubles), vcc = a fixed lane pattern; result: v[30:33] (zeroed in front of the instruction; v30 / v31 are hashed).
// PRODUCT_FORMS: generated from the built library (scripts/gen_insn_probe_forms.py); EXTRA_FORMS: forms the product does NOT contain -
// the positive control and the 16-bit packed forms with the swizzle that fails on register pairs.
#include "insn_probe_forms.inc"
#define EXTRA_FORMS(X)                                                                                    \
    X(0, "v_pk_mul_f32 v[30:31], v[10:11], v[14:15] op_sel:[0,1] op_sel_hi:[1,0]")                         \
    X(1, "v_pk_mul_f16 v30, v10, v14 op_sel:[0,1] op_sel_hi:[1,0]")                                       \
    X(2, "v_pk_fma_f16 v30, v10, v14, v18 op_sel:[0,1,0] op_sel_hi:[1,0,1]")                               \
    X(3, "v_pk_add_f16 v30, v10, v14 op_sel:[0,1] op_sel_hi:[1,0]")                                       \
    X(4, "v_pk_mul_lo_u16 v30, v10, v14 op_sel:[0,1] op_sel_hi:[1,0]")                                    \
    X(5, "v_dot2c_f32_f16 v30, v10, v14")
constexpr int N_EXTRA_FORMS = 6;
constexpr int N_MODES = N_PRODUCT_FORMS + N_EXTRA_FORMS;

struct Probe { unsigned *hash, *ref, *bad; int mode; bool have_ref; };
constexpr int BLOCKS = 2048, THREADS = 256, N = BLOCKS * THREADS;

template <int MODE, bool EXTRA>
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
#define PROLOGUE "v_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\tv_mov_b32 v12, %4\n\tv_mov_b32 v13, %5\n\t"                                  \
                 "v_mov_b32 v14, %4\n\tv_mov_b32 v15, %5\n\tv_mov_b32 v16, %6\n\tv_mov_b32 v17, %7\n\t"                                  \
                 "v_mov_b32 v18, %6\n\tv_mov_b32 v19, %7\n\tv_mov_b32 v20, %2\n\tv_mov_b32 v21, %3\n\t"                                  \
                 "v_mov_b32 v30, 0\n\tv_mov_b32 v31, 0\n\tv_mov_b32 v32, 0\n\tv_mov_b32 v33, 0\n\t"                                      \
                 "s_mov_b32 vcc_lo, 0x5a5a5a5a\n\ts_mov_b32 vcc_hi, 0xa5a5a5a5\n\ts_nop 4\n\t"
#define X(M, TEXT)                                                                                                              \
        if (MODE == M)                                                                                                          \
            asm volatile(PROLOGUE TEXT "\n\ts_nop 1\n\tv_mov_b32 %0, v30\n\tv_mov_b32 %1, v31"                                    \
                         : "=v"(d0), "=v"(d1) : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5])                   \
                         : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v30", "v31", "v32", "v33", "vcc");
        if (EXTRA) { EXTRA_FORMS(X) } else { PRODUCT_FORMS(X) }
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
        if (p->mode < N_PRODUCT_FORMS) switch (p->mode) {
#define X(M, TEXT) case M: hipLaunchKernelGGL((insn_probe_kernel<M, false>), dim3(BLOCKS), dim3(THREADS), 0, s, p->hash, 256); break;
            PRODUCT_FORMS(X)
#undef X
        } else switch (p->mode - N_PRODUCT_FORMS) {
#define X(M, TEXT) case M: hipLaunchKernelGGL((insn_probe_kernel<M, true>), dim3(BLOCKS), dim3(THREADS), 0, s, p->hash, 256); break;
            EXTRA_FORMS(X)
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
    if (m >= 0 && m < N_PRODUCT_FORMS) switch (m) {
#define X(M, TEXT) case M: return TEXT;
        PRODUCT_FORMS(X)
#undef X
    } else switch (m - N_PRODUCT_FORMS) {
#define X(M, TEXT) case M: return TEXT;
        EXTRA_FORMS(X)
#undef X
    }
    return nullptr;
}

int victim_modes(int* product) { if (product) *product = N_PRODUCT_FORMS; return N_MODES; }

void victim_destroy(void* h) {
    Probe* p = (Probe*)h;
    (void)hipFree(p->hash); (void)hipFree(p->ref); (void)hipFree(p->bad);
    delete p;
}

}  // extern "C"
