// pk_probe.hip - ONE instruction as the victim: v_pk_mul_f32 with op_sel:[0,1] (low result = src0.lo x src1.HI).
//
// profiles/r06_aggregate_rnorm_diagnosis.md section 5 bisected the one fault of al_aggregate_kernel's packed-fp32 shape, in the
// compiler's own assembly, to a single instruction - `v_pk_mul_f32 v[32:33], v[14:15], v[12:13] op_sel:[0,1] op_sel_hi:[1,0]` - whose LOW
// result is exactly 0.0 in lanes 48..63 about once per 10^4 wave executions while lg_attention_p_kernel runs on another queue; of
// the 16 op_sel / op_sel_hi combinations exactly the four with op_sel = [0,1] fail, always in the low half.  This file takes the
// instruction out of that kernel: a loop of it in inline assembly on lane-dependent operands, each result checked against
// single-width multiplies, with the same C entry points as agg_victim.hip so that scripts/agg_victim_run.py drives it beside
// the same aggressors.  MODE (victim_create's `F` argument): index into a generated table - every op_sel of v_pk_mul_f32, v_pk_add_f32
// and v_pk_fma_f32, each with op_sel_hi straight and [1,0,..], and the same-source horizontal add (victim_mode_text(m) names it);
// +256: a global load per iteration in front of the instruction.
//   hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -shared -o scripts/ubench/libpkprobe.so scripts/ubench/pk_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Probe { unsigned* bad; float* tab; int mode; };

// bad[0] low-half mismatches, bad[1] high-half mismatches; bad[4 + 4k ...]: first events - (lane | half << 8 | block << 16), got,
// expected, launch.  MODE: index into the table below (generated: every op_sel of v_pk_mul / add / fma_f32, op_sel_hi straight and
// [1,0,..]); LOADS: a global load per iteration in front of the instruction.
template <int MODE, bool LOADS>
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
        if (MODE == 0) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b));
            asm volatile("s_nop 1\n\tv_mul_f32 %0, %1, %2" : "=v"(elo) : "v"(a.x), "v"(b.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ehi) : "v"(a.y), "v"(b.y));
        }
        if (MODE == 1) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
            asm volatile("s_nop 1\n\tv_mul_f32 %0, %1, %2" : "=v"(elo) : "v"(a.x), "v"(b.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ehi) : "v"(a.y), "v"(b.x));
        }
        if (MODE == 2) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b));
            asm volatile("s_nop 1\n\tv_mul_f32 %0, %1, %2" : "=v"(elo) : "v"(a.y), "v"(b.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ehi) : "v"(a.y), "v"(b.y));
        }
        if (MODE == 3) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
            asm volatile("s_nop 1\n\tv_mul_f32 %0, %1, %2" : "=v"(elo) : "v"(a.y), "v"(b.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ehi) : "v"(a.y), "v"(b.x));
        }
        if (MODE == 4) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b));
            asm volatile("s_nop 1\n\tv_mul_f32 %0, %1, %2" : "=v"(elo) : "v"(a.x), "v"(b.y));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ehi) : "v"(a.y), "v"(b.y));
        }
        if (MODE == 5) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
            asm volatile("s_nop 1\n\tv_mul_f32 %0, %1, %2" : "=v"(elo) : "v"(a.x), "v"(b.y));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ehi) : "v"(a.y), "v"(b.x));
        }
        if (MODE == 6) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b));
            asm volatile("s_nop 1\n\tv_mul_f32 %0, %1, %2" : "=v"(elo) : "v"(a.y), "v"(b.y));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ehi) : "v"(a.y), "v"(b.y));
        }
        if (MODE == 7) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
            asm volatile("s_nop 1\n\tv_mul_f32 %0, %1, %2" : "=v"(elo) : "v"(a.y), "v"(b.y));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ehi) : "v"(a.y), "v"(b.x));
        }
        if (MODE == 8) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b));
            asm volatile("s_nop 1\n\tv_add_f32 %0, %1, %2" : "=v"(elo) : "v"(a.x), "v"(b.x));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(ehi) : "v"(a.y), "v"(b.y));
        }
        if (MODE == 9) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
            asm volatile("s_nop 1\n\tv_add_f32 %0, %1, %2" : "=v"(elo) : "v"(a.x), "v"(b.x));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(ehi) : "v"(a.y), "v"(b.x));
        }
        if (MODE == 10) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b));
            asm volatile("s_nop 1\n\tv_add_f32 %0, %1, %2" : "=v"(elo) : "v"(a.y), "v"(b.x));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(ehi) : "v"(a.y), "v"(b.y));
        }
        if (MODE == 11) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
            asm volatile("s_nop 1\n\tv_add_f32 %0, %1, %2" : "=v"(elo) : "v"(a.y), "v"(b.x));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(ehi) : "v"(a.y), "v"(b.x));
        }
        if (MODE == 12) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b));
            asm volatile("s_nop 1\n\tv_add_f32 %0, %1, %2" : "=v"(elo) : "v"(a.x), "v"(b.y));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(ehi) : "v"(a.y), "v"(b.y));
        }
        if (MODE == 13) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
            asm volatile("s_nop 1\n\tv_add_f32 %0, %1, %2" : "=v"(elo) : "v"(a.x), "v"(b.y));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(ehi) : "v"(a.y), "v"(b.x));
        }
        if (MODE == 14) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b));
            asm volatile("s_nop 1\n\tv_add_f32 %0, %1, %2" : "=v"(elo) : "v"(a.y), "v"(b.y));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(ehi) : "v"(a.y), "v"(b.y));
        }
        if (MODE == 15) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
            asm volatile("s_nop 1\n\tv_add_f32 %0, %1, %2" : "=v"(elo) : "v"(a.y), "v"(b.y));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(ehi) : "v"(a.y), "v"(b.x));
        }
        if (MODE == 16) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
            asm volatile("s_nop 1\n\tv_fma_f32 %0, %1, %2, %3" : "=v"(elo) : "v"(a.x), "v"(b.x), "v"(c.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(ehi) : "v"(a.y), "v"(b.y), "v"(c.y));
        }
        if (MODE == 17) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
            asm volatile("s_nop 1\n\tv_fma_f32 %0, %1, %2, %3" : "=v"(elo) : "v"(a.x), "v"(b.x), "v"(c.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(ehi) : "v"(a.y), "v"(b.x), "v"(c.y));
        }
        if (MODE == 18) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
            asm volatile("s_nop 1\n\tv_fma_f32 %0, %1, %2, %3" : "=v"(elo) : "v"(a.y), "v"(b.x), "v"(c.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(ehi) : "v"(a.y), "v"(b.y), "v"(c.y));
        }
        if (MODE == 19) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
            asm volatile("s_nop 1\n\tv_fma_f32 %0, %1, %2, %3" : "=v"(elo) : "v"(a.y), "v"(b.x), "v"(c.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(ehi) : "v"(a.y), "v"(b.x), "v"(c.y));
        }
        if (MODE == 20) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
            asm volatile("s_nop 1\n\tv_fma_f32 %0, %1, %2, %3" : "=v"(elo) : "v"(a.x), "v"(b.y), "v"(c.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(ehi) : "v"(a.y), "v"(b.y), "v"(c.y));
        }
        if (MODE == 21) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
            asm volatile("s_nop 1\n\tv_fma_f32 %0, %1, %2, %3" : "=v"(elo) : "v"(a.x), "v"(b.y), "v"(c.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(ehi) : "v"(a.y), "v"(b.x), "v"(c.y));
        }
        if (MODE == 22) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
            asm volatile("s_nop 1\n\tv_fma_f32 %0, %1, %2, %3" : "=v"(elo) : "v"(a.y), "v"(b.y), "v"(c.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(ehi) : "v"(a.y), "v"(b.y), "v"(c.y));
        }
        if (MODE == 23) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
            asm volatile("s_nop 1\n\tv_fma_f32 %0, %1, %2, %3" : "=v"(elo) : "v"(a.y), "v"(b.y), "v"(c.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(ehi) : "v"(a.y), "v"(b.x), "v"(c.y));
        }
        if (MODE == 24) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
            asm volatile("s_nop 1\n\tv_fma_f32 %0, %1, %2, %3" : "=v"(elo) : "v"(a.x), "v"(b.x), "v"(c.y));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(ehi) : "v"(a.y), "v"(b.y), "v"(c.y));
        }
        if (MODE == 25) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
            asm volatile("s_nop 1\n\tv_fma_f32 %0, %1, %2, %3" : "=v"(elo) : "v"(a.x), "v"(b.x), "v"(c.y));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(ehi) : "v"(a.y), "v"(b.x), "v"(c.y));
        }
        if (MODE == 26) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
            asm volatile("s_nop 1\n\tv_fma_f32 %0, %1, %2, %3" : "=v"(elo) : "v"(a.y), "v"(b.x), "v"(c.y));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(ehi) : "v"(a.y), "v"(b.y), "v"(c.y));
        }
        if (MODE == 27) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
            asm volatile("s_nop 1\n\tv_fma_f32 %0, %1, %2, %3" : "=v"(elo) : "v"(a.y), "v"(b.x), "v"(c.y));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(ehi) : "v"(a.y), "v"(b.x), "v"(c.y));
        }
        if (MODE == 28) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
            asm volatile("s_nop 1\n\tv_fma_f32 %0, %1, %2, %3" : "=v"(elo) : "v"(a.x), "v"(b.y), "v"(c.y));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(ehi) : "v"(a.y), "v"(b.y), "v"(c.y));
        }
        if (MODE == 29) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
            asm volatile("s_nop 1\n\tv_fma_f32 %0, %1, %2, %3" : "=v"(elo) : "v"(a.x), "v"(b.y), "v"(c.y));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(ehi) : "v"(a.y), "v"(b.x), "v"(c.y));
        }
        if (MODE == 30) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,1] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
            asm volatile("s_nop 1\n\tv_fma_f32 %0, %1, %2, %3" : "=v"(elo) : "v"(a.y), "v"(b.y), "v"(c.y));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(ehi) : "v"(a.y), "v"(b.y), "v"(c.y));
        }
        if (MODE == 31) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
            asm volatile("s_nop 1\n\tv_fma_f32 %0, %1, %2, %3" : "=v"(elo) : "v"(a.y), "v"(b.y), "v"(c.y));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(ehi) : "v"(a.y), "v"(b.x), "v"(c.y));
        }
        if (MODE == 32) {
            asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a));
            asm volatile("s_nop 1\n\tv_add_f32 %0, %1, %2" : "=v"(elo) : "v"(a.x), "v"(a.y));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(ehi) : "v"(a.y), "v"(a.x));
        }
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

const char* const MODE_TEXT[] = {
    "v_pk_mul_f32 A, B op_sel:[0,0] op_sel_hi:[1,1]",
    "v_pk_mul_f32 A, B op_sel:[0,0] op_sel_hi:[1,0]",
    "v_pk_mul_f32 A, B op_sel:[1,0] op_sel_hi:[1,1]",
    "v_pk_mul_f32 A, B op_sel:[1,0] op_sel_hi:[1,0]",
    "v_pk_mul_f32 A, B op_sel:[0,1] op_sel_hi:[1,1]",
    "v_pk_mul_f32 A, B op_sel:[0,1] op_sel_hi:[1,0]",
    "v_pk_mul_f32 A, B op_sel:[1,1] op_sel_hi:[1,1]",
    "v_pk_mul_f32 A, B op_sel:[1,1] op_sel_hi:[1,0]",
    "v_pk_add_f32 A, B op_sel:[0,0] op_sel_hi:[1,1]",
    "v_pk_add_f32 A, B op_sel:[0,0] op_sel_hi:[1,0]",
    "v_pk_add_f32 A, B op_sel:[1,0] op_sel_hi:[1,1]",
    "v_pk_add_f32 A, B op_sel:[1,0] op_sel_hi:[1,0]",
    "v_pk_add_f32 A, B op_sel:[0,1] op_sel_hi:[1,1]",
    "v_pk_add_f32 A, B op_sel:[0,1] op_sel_hi:[1,0]",
    "v_pk_add_f32 A, B op_sel:[1,1] op_sel_hi:[1,1]",
    "v_pk_add_f32 A, B op_sel:[1,1] op_sel_hi:[1,0]",
    "v_pk_fma_f32 A, B, C op_sel:[0,0,0] op_sel_hi:[1,1,1]",
    "v_pk_fma_f32 A, B, C op_sel:[0,0,0] op_sel_hi:[1,0,1]",
    "v_pk_fma_f32 A, B, C op_sel:[1,0,0] op_sel_hi:[1,1,1]",
    "v_pk_fma_f32 A, B, C op_sel:[1,0,0] op_sel_hi:[1,0,1]",
    "v_pk_fma_f32 A, B, C op_sel:[0,1,0] op_sel_hi:[1,1,1]",
    "v_pk_fma_f32 A, B, C op_sel:[0,1,0] op_sel_hi:[1,0,1]",
    "v_pk_fma_f32 A, B, C op_sel:[1,1,0] op_sel_hi:[1,1,1]",
    "v_pk_fma_f32 A, B, C op_sel:[1,1,0] op_sel_hi:[1,0,1]",
    "v_pk_fma_f32 A, B, C op_sel:[0,0,1] op_sel_hi:[1,1,1]",
    "v_pk_fma_f32 A, B, C op_sel:[0,0,1] op_sel_hi:[1,0,1]",
    "v_pk_fma_f32 A, B, C op_sel:[1,0,1] op_sel_hi:[1,1,1]",
    "v_pk_fma_f32 A, B, C op_sel:[1,0,1] op_sel_hi:[1,0,1]",
    "v_pk_fma_f32 A, B, C op_sel:[0,1,1] op_sel_hi:[1,1,1]",
    "v_pk_fma_f32 A, B, C op_sel:[0,1,1] op_sel_hi:[1,0,1]",
    "v_pk_fma_f32 A, B, C op_sel:[1,1,1] op_sel_hi:[1,1,1]",
    "v_pk_fma_f32 A, B, C op_sel:[1,1,1] op_sel_hi:[1,0,1]",
    "v_pk_add_f32 A, A op_sel:[0,1] op_sel_hi:[1,0]",
};
constexpr int N_MODES = 33;

}  // namespace

extern "C" {

void* victim_create(int, int, int mode, unsigned) {
    Probe* p = new Probe();
    p->mode = mode;
    if (hipMalloc(&p->bad, 64 * sizeof(unsigned)) != hipSuccess) return nullptr;
    hipMemset(p->bad, 0, 64 * sizeof(unsigned));
    std::vector<float> h(1u << 20);
    unsigned s = 12345u;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (float)(s >> 8) * (1.0f / 16777216.0f); }
    if (hipMalloc(&p->tab, h.size() * sizeof(float)) != hipSuccess) return nullptr;
    hipMemcpy(p->tab, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    return p;
}

// `iters` launches of 2048 workgroups x 256 threads x 256 instructions each
int victim_run(void* h, void* stream, int iters, int) {
    Probe* p = (Probe*)h;
    hipStream_t s = (hipStream_t)stream;
    static unsigned launch = 0;
    for (int i = 0; i < iters; ++i) {
        ++launch;
        const int m = p->mode & 0xff; const bool loads = (p->mode >> 8) & 1;
        switch (m) {
#define L(M) case M: if (loads) hipLaunchKernelGGL((pk_probe_kernel<M, true>), dim3(2048), dim3(256), 0, s, p->bad, p->tab, 256, launch); \
                     else hipLaunchKernelGGL((pk_probe_kernel<M, false>), dim3(2048), dim3(256), 0, s, p->bad, p->tab, 256, launch); break;
            L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15) L(16) L(17) L(18) L(19) L(20) L(21) L(22) L(23) L(24) L(25) L(26) L(27) L(28) L(29) L(30) L(31) L(32)
#undef L
            default: return -3;
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int victim_poll(void* h, void* stream, unsigned* out64) {
    Probe* p = (Probe*)h;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
    return hipMemcpy(out64, p->bad, 64 * sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}

// the instruction of mode m (A, B, C: the three operand pairs), or nullptr
const char* victim_mode_text(int m) { return m >= 0 && m < N_MODES ? MODE_TEXT[m] : nullptr; }

void victim_destroy(void* h) {
    Probe* p = (Probe*)h;
    (void)hipFree(p->bad); (void)hipFree(p->tab);
    delete p;
}

}  // extern "C"
