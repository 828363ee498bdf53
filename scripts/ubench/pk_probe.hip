// pk_probe.hip - ONE instruction as the victim: v_pk_{mul,add,fma}_f32 under every op_sel / op_sel_hi.
//
// profiles/r06_aggregate_rnorm_diagnosis.md section 5 bisected the one fault of al_aggregate_kernel's packed-fp32 shape, in the
// compiler's own assembly, to a single instruction - `v_pk_mul_f32 v[32:33], v[14:15], v[12:13] op_sel:[0,1] op_sel_hi:[1,0]` - whose LOW
// result is exactly 0.0 in lanes 48..63 about once per 10^4 wave executions while a kernel built on wide-operand MFMAs runs on
// another queue.  This file takes the instruction out of that kernel: a loop of it in inline assembly on lane-dependent operands
// in [0.5, 1.5), each half of each result checked against a single-width instruction on the same registers, with the same C
// entry points as agg_victim.hip so that scripts/agg_victim_run.py drives it beside the same aggressors (scripts/pk_probe_run.sh).
// MODE (victim_create's `F` argument; victim_mode_text(m) names it):
//     0 ..  15   v_pk_mul_f32  op_sel:[a,b]   op_sel_hi:[c,d]        m = 8a + 4b + 2c + d
//    16 ..  31   v_pk_add_f32  the same
//    32 ..  95   v_pk_fma_f32  op_sel:[a,b,e] op_sel_hi:[c,d,f]      m = 32 + 32a + 16b + 8e + 4c + 2d + f
//    96          v_pk_add_f32 D, A, A op_sel:[0,1] op_sel_hi:[1,0]   (same-source horizontal add: the one instance in the product)
//    97          v_pk_mul_f32 D, A, A op_sel:[0,1] op_sel_hi:[1,0]
//    +256        a global load per iteration in front of the instruction
//   hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -shared -o scripts/ubench/libpkprobe.so scripts/ubench/pk_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Probe { unsigned* bad; float* tab; int mode; };

#define COMBOS2(X) X(0,0,0,0) X(0,0,0,1) X(0,0,1,0) X(0,0,1,1) X(0,1,0,0) X(0,1,0,1) X(0,1,1,0) X(0,1,1,1) X(1,0,0,0) X(1,0,0,1) X(1,0,1,0) X(1,0,1,1) X(1,1,0,0) X(1,1,0,1) X(1,1,1,0) X(1,1,1,1)
#define COMBOS3(X) X(0,0,0,0,0,0) X(0,0,0,0,0,1) X(0,0,0,0,1,0) X(0,0,0,0,1,1) X(0,0,0,1,0,0) X(0,0,0,1,0,1) X(0,0,0,1,1,0) X(0,0,0,1,1,1) X(0,0,1,0,0,0) X(0,0,1,0,0,1) X(0,0,1,0,1,0) X(0,0,1,0,1,1) X(0,0,1,1,0,0) X(0,0,1,1,0,1) X(0,0,1,1,1,0) X(0,0,1,1,1,1) X(0,1,0,0,0,0) X(0,1,0,0,0,1) X(0,1,0,0,1,0) X(0,1,0,0,1,1) X(0,1,0,1,0,0) X(0,1,0,1,0,1) X(0,1,0,1,1,0) X(0,1,0,1,1,1) X(0,1,1,0,0,0) X(0,1,1,0,0,1) X(0,1,1,0,1,0) X(0,1,1,0,1,1) X(0,1,1,1,0,0) X(0,1,1,1,0,1) X(0,1,1,1,1,0) X(0,1,1,1,1,1) X(1,0,0,0,0,0) X(1,0,0,0,0,1) X(1,0,0,0,1,0) X(1,0,0,0,1,1) X(1,0,0,1,0,0) X(1,0,0,1,0,1) X(1,0,0,1,1,0) X(1,0,0,1,1,1) X(1,0,1,0,0,0) X(1,0,1,0,0,1) X(1,0,1,0,1,0) X(1,0,1,0,1,1) X(1,0,1,1,0,0) X(1,0,1,1,0,1) X(1,0,1,1,1,0) X(1,0,1,1,1,1) X(1,1,0,0,0,0) X(1,1,0,0,0,1) X(1,1,0,0,1,0) X(1,1,0,0,1,1) X(1,1,0,1,0,0) X(1,1,0,1,0,1) X(1,1,0,1,1,0) X(1,1,0,1,1,1) X(1,1,1,0,0,0) X(1,1,1,0,0,1) X(1,1,1,0,1,0) X(1,1,1,0,1,1) X(1,1,1,1,0,0) X(1,1,1,1,0,1) X(1,1,1,1,1,0) X(1,1,1,1,1,1)

__device__ __forceinline__ float pick(f32x2 v, int hi) { return hi ? v.y : v.x; }

// one packed instruction and the two single-width ones it must agree with; OP: 0 mul, 1 add, 2 fma
template <int OP, int A, int B, int E, int C, int D, int F>
__device__ __forceinline__ void pk_and_expected(f32x2 a, f32x2 b, f32x2 c, f32x2& d, float& elo, float& ehi);

#define X(A, B, C, D)                                                                                                         \
    template <> __device__ __forceinline__ void pk_and_expected<0, A, B, 0, C, D, 1>(f32x2 a, f32x2 b, f32x2, f32x2& d, float& elo, float& ehi) { \
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[" #A "," #B "] op_sel_hi:[" #C "," #D "]" : "=v"(d) : "v"(a), "v"(b));     \
        asm volatile("s_nop 1\n\tv_mul_f32 %0, %1, %2" : "=v"(elo) : "v"(pick(a, A)), "v"(pick(b, B)));                          \
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ehi) : "v"(pick(a, C)), "v"(pick(b, D)));                                     \
    }                                                                                                                         \
    template <> __device__ __forceinline__ void pk_and_expected<1, A, B, 0, C, D, 1>(f32x2 a, f32x2 b, f32x2, f32x2& d, float& elo, float& ehi) { \
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[" #A "," #B "] op_sel_hi:[" #C "," #D "]" : "=v"(d) : "v"(a), "v"(b));     \
        asm volatile("s_nop 1\n\tv_add_f32 %0, %1, %2" : "=v"(elo) : "v"(pick(a, A)), "v"(pick(b, B)));                          \
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(ehi) : "v"(pick(a, C)), "v"(pick(b, D)));                                     \
    }
COMBOS2(X)
#undef X
#define X(A, B, E, C, D, F)                                                                                                   \
    template <> __device__ __forceinline__ void pk_and_expected<2, A, B, E, C, D, F>(f32x2 a, f32x2 b, f32x2 c, f32x2& d, float& elo, float& ehi) { \
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[" #A "," #B "," #E "] op_sel_hi:[" #C "," #D "," #F "]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); \
        asm volatile("s_nop 1\n\tv_fma_f32 %0, %1, %2, %3" : "=v"(elo) : "v"(pick(a, A)), "v"(pick(b, B)), "v"(pick(c, E)));     \
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(ehi) : "v"(pick(a, C)), "v"(pick(b, D)), "v"(pick(c, F)));                \
    }
COMBOS3(X)
#undef X
// same-source forms (OP 3: add, 4: mul)
template <> __device__ __forceinline__ void pk_and_expected<3, 0, 1, 0, 1, 0, 1>(f32x2 a, f32x2, f32x2, f32x2& d, float& elo, float& ehi) {
    asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a));
    asm volatile("s_nop 1\n\tv_add_f32 %0, %1, %2" : "=v"(elo) : "v"(a.x), "v"(a.y));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(ehi) : "v"(a.y), "v"(a.x));
}
template <> __device__ __forceinline__ void pk_and_expected<4, 0, 1, 0, 1, 0, 1>(f32x2 a, f32x2, f32x2, f32x2& d, float& elo, float& ehi) {
    asm volatile("v_pk_mul_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a));
    asm volatile("s_nop 1\n\tv_mul_f32 %0, %1, %2" : "=v"(elo) : "v"(a.x), "v"(a.y));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ehi) : "v"(a.y), "v"(a.x));
}

// bad[0] low-half mismatches, bad[1] high-half mismatches; bad[4 + 4k ...]: first events - (lane | half << 8 | block << 16), got,
// expected, launch
template <int OP, int A, int B, int E, int C, int D, int F, bool LOADS>
__global__ __launch_bounds__(256) void pk_probe_kernel(unsigned* __restrict__ bad, const float* __restrict__ tab, int iters, unsigned launch) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    // operands in [0.5, 1.5): every product and sum is a normal, non-zero number
    f32x2 a = {0.5f + (float)(tid & 1023) * (1.0f / 1024.0f), 0.5f + (float)((tid * 7u) & 1023) * (1.0f / 1024.0f)};
    f32x2 b = {0.5f + (float)((tid * 13u) & 1023) * (1.0f / 1024.0f), 0.5f + (float)((tid * 29u) & 1023) * (1.0f / 1024.0f)};
    f32x2 c = {0.5f + (float)((tid * 17u) & 1023) * (1.0f / 1024.0f), 0.5f + (float)((tid * 37u) & 1023) * (1.0f / 1024.0f)};
    unsigned nlo = 0, nhi = 0;
    for (int it = 0; it < iters; ++it) {
        if (LOADS) b.x = 0.5f + tab[(tid * 31u + (unsigned)it * 977u) & 0xfffffu];     // (values in [0, 1))
        f32x2 d;
        float elo, ehi;
        pk_and_expected<OP, A, B, E, C, D, F>(a, b, c, d, elo, ehi);
        asm volatile("s_nop 1" : "+v"(d));                    // (a result written with op_sel needs a wait state in front of its reader)
        if (d.x != elo || d.y != ehi) {
            const int half = d.x != elo ? 0 : 1;
            if (half == 0) ++nlo; else ++nhi;
            const unsigned k = atomicAdd(&bad[3], 1u);
            if (k < 15) {
                bad[4 + 4 * k] = (threadIdx.x & 63) | (half << 8) | (blockIdx.x << 16);
                bad[5 + 4 * k] = __float_as_uint(half ? d.y : d.x); bad[6 + 4 * k] = __float_as_uint(half ? ehi : elo); bad[7 + 4 * k] = launch;
            }
        }
        a.x += 1.0f / 4096.0f; if (a.x >= 1.5f) a.x -= 1.0f;      // other operands every iteration
        b.y += 1.0f / 8192.0f; if (b.y >= 1.5f) b.y -= 1.0f;
    }
    if (nlo) atomicAdd(&bad[0], nlo);
    if (nhi) atomicAdd(&bad[1], nhi);
}

template <int OP, int A, int B, int E, int C, int D, int F>
void launch_one(Probe* p, hipStream_t s, bool loads, unsigned launch) {
    if (loads) hipLaunchKernelGGL((pk_probe_kernel<OP, A, B, E, C, D, F, true>), dim3(2048), dim3(256), 0, s, p->bad, p->tab, 256, launch);
    else hipLaunchKernelGGL((pk_probe_kernel<OP, A, B, E, C, D, F, false>), dim3(2048), dim3(256), 0, s, p->bad, p->tab, 256, launch);
}

constexpr int N_MODES = 98;

}  // namespace

extern "C" {

void* victim_create(int, int, int mode, unsigned) {
    Probe* p = new Probe();
    p->mode = mode;
    if (hipMalloc(&p->bad, 64 * sizeof(unsigned)) != hipSuccess) return nullptr;
    (void)hipMemset(p->bad, 0, 64 * sizeof(unsigned));
    std::vector<float> h(1u << 20);
    unsigned s = 12345u;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (float)(s >> 8) * (1.0f / 16777216.0f); }
    if (hipMalloc(&p->tab, h.size() * sizeof(float)) != hipSuccess) return nullptr;
    (void)hipMemcpy(p->tab, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    return p;
}

// `iters` launches of 2048 workgroups x 256 threads x 256 instructions each
int victim_run(void* h, void* stream, int iters, int) {
    Probe* p = (Probe*)h;
    hipStream_t s = (hipStream_t)stream;
    static unsigned launch = 0;
    const int m = p->mode & 0xff; const bool loads = (p->mode >> 8) & 1;
    for (int i = 0; i < iters; ++i) {
        ++launch;
        bool done = false;
#define X(A, B, C, D)                                                                                          \
        if (m == 8 * A + 4 * B + 2 * C + D) { launch_one<0, A, B, 0, C, D, 1>(p, s, loads, launch); done = true; }       \
        if (m == 16 + 8 * A + 4 * B + 2 * C + D) { launch_one<1, A, B, 0, C, D, 1>(p, s, loads, launch); done = true; }
        COMBOS2(X)
#undef X
#define X(A, B, E, C, D, F)                                                                                    \
        if (m == 32 + 32 * A + 16 * B + 8 * E + 4 * C + 2 * D + F) { launch_one<2, A, B, E, C, D, F>(p, s, loads, launch); done = true; }
        COMBOS3(X)
#undef X
        if (m == 96) { launch_one<3, 0, 1, 0, 1, 0, 1>(p, s, loads, launch); done = true; }
        if (m == 97) { launch_one<4, 0, 1, 0, 1, 0, 1>(p, s, loads, launch); done = true; }
        if (!done) return -3;
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int victim_poll(void* h, void* stream, unsigned* out64) {
    Probe* p = (Probe*)h;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
    return hipMemcpy(out64, p->bad, 64 * sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}

// the instruction of mode m (A, B, C: the three operand pairs), or nullptr
const char* victim_mode_text(int m) {
    static char buf[128];
    if (m < 0 || m >= N_MODES) return nullptr;
    if (m == 96) return "v_pk_add_f32 D, A, A op_sel:[0,1] op_sel_hi:[1,0]";
    if (m == 97) return "v_pk_mul_f32 D, A, A op_sel:[0,1] op_sel_hi:[1,0]";
    if (m < 32) snprintf(buf, sizeof buf, "v_pk_%s_f32 D, A, B op_sel:[%d,%d] op_sel_hi:[%d,%d]", m < 16 ? "mul" : "add", (m >> 3) & 1, (m >> 2) & 1, (m >> 1) & 1, m & 1);
    else { const int k = m - 32; snprintf(buf, sizeof buf, "v_pk_fma_f32 D, A, B, C op_sel:[%d,%d,%d] op_sel_hi:[%d,%d,%d]", (k >> 5) & 1, (k >> 4) & 1, (k >> 3) & 1, (k >> 2) & 1, (k >> 1) & 1, k & 1); }
    return buf;
}

void victim_destroy(void* h) {
    Probe* p = (Probe*)h;
    (void)hipFree(p->bad); (void)hipFree(p->tab);
    delete p;
}

}  // extern "C"
