// Stand-alone timing of the LightGlue attention kernel (same TU as the product kernels, synthetic
// operands, no downstream kernels): used for A/B of schedule variants and for ablations that would
// poison a full match.  Variants are separate builds (-D flags) run back to back on the same device.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I opencv-simpleslam_amd/csrc \
//         scripts/ubench/attn_bench.hip -o /tmp/attn_bench && /tmp/attn_bench [N=2048] [pairs=8] [KS=1] [rounds=5]
#include "../../opencv-simpleslam_amd/csrc/lightglue_kernels.hip"
#include "attn_hs_reference.hpp"
#include "attn_pp_experiment.hpp"
#include "attn_w1_experiment.hpp"
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
namespace sslam { void set_error(const char*, ...) {} void ctx_retain(sslam_ctx*) {} void ctx_release(sslam_ctx*) {} }
extern "C" const unsigned char sslam_lg_attention_asm_hsaco[1] = {0};      // (the product embeds the code objects; here one is loaded from a file)
extern "C" const unsigned char sslam_lg_attention_asm_p1_hsaco[1] = {0};

// the hand-scheduled kernel (opencv-simpleslam_amd/csrc/gen_lg_attention_asm.py): loaded from the code object named by ATTN_HSACO
struct AsmArgs {
    const void *q_hi, *q_lo, *k_hi, *k_lo, *vt_hi, *vt_lo; void *msg_hi, *msg_lo; const void* ctrl;
    int cross, Kc, NIc, nqb, nslab; unsigned magic; int nih, lks;      // lks = 0: no key ranges here
    float *o_part, *m_part, *l_part; unsigned* dbg;                    // dbg: the stamp buffer of a diagnostic build (offset 128)
};
static hipFunction_t asm_fn;
static unsigned* g_dbg = nullptr;
static bool asm_load() {
    const char* path = getenv("ATTN_HSACO");
    if (!path) return false;
    hipModule_t m;
    if (hipModuleLoad(&m, path) != hipSuccess) { printf("cannot load %s\n", path); return false; }
    return hipModuleGetFunction(&asm_fn, m, "lg_attention_asm_kernel") == hipSuccess;
}
static void asm_launch(const AttnArgsH& a, int NI) {
    AsmArgs k{a.Q.hi, a.Q.lo, a.K.hi, a.K.lo, a.VT.hi, a.VT.lo, a.msg.hi, a.msg.lo, a.ctrl, a.cross, a.Kc, a.NIc,
              sslam::cdiv(a.Kc, AQ), NI * NH, 0, NI * NH, 0, nullptr, nullptr, nullptr, g_dbg};
    k.magic = k.nqb > 1 ? (unsigned)((1ull << 32) / (unsigned)k.nqb + 1) : 0;
    size_t sz = sizeof(k);
    void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &k, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    hipModuleLaunchKernel(asm_fn, k.nqb, k.nslab, 1, 256, 1, 1, 0, 0, nullptr, cfg);
}

__global__ void fill_half(_Float16* p, size_t n, unsigned seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = (_Float16)(((x & 0xffff) / 65536.0f - 0.5f) * scale);
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 2048;
    const int B = argc > 2 ? atoi(argv[2]) : 8;
    const int KS = argc > 3 ? atoi(argv[3]) : 1;
    const int rounds = argc > 4 ? atoi(argv[4]) : 5;
    const int Kc = (N + 127) / 128 * 128, NI = 2 * B;
    const size_t plane = (size_t)NI * NH * Kc * DH;
    _Float16* buf[8];
    for (int i = 0; i < 8; ++i) {
        hipMalloc(&buf[i], plane * 2);
        const float amp = getenv("ATTN_ZERO") && atoi(getenv("ATTN_ZERO")) ? 0.0f : 1.0f;   // zero operands: the DVFS / power check
        fill_half<<<(plane + 255) / 256, 256>>>(buf[i], plane, 17 * i + 1, amp * ((i & 1) ? 0.01f : 2.0f));
    }
    float *o_part, *m_part, *l_part; LGCtrl* ctrl;
    hipMalloc(&o_part, (size_t)KS * plane * 4); hipMalloc(&m_part, (size_t)KS * NI * NH * Kc * 4);
    hipMalloc(&l_part, (size_t)KS * NI * NH * Kc * 4); hipMalloc(&ctrl, sizeof(LGCtrl) * B);
    std::vector<LGCtrl> h(B); for (auto& c : h) { c = LGCtrl{}; c.n[0] = c.n[1] = N; }
    hipMemcpy(ctrl, h.data(), sizeof(LGCtrl) * B, hipMemcpyHostToDevice);
    AttnArgsH a{{buf[0], buf[1]}, {buf[2], buf[3]}, {buf[4], buf[5]}, 0, o_part, m_part, l_part, SplitOut{buf[6], buf[7]}, KS, Kc, NI, ctrl};
    dim3 grid(sslam::cdiv(Kc, AQ), NI * NH, KS);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int ppm = getenv("ATTN_PP") ? atoi(getenv("ATTN_PP")) : 0;
    const bool pp = ppm == 1 && KS == 1;     // 8-wave ping-pong form
    const bool w1 = ppm == 2 && KS == 1;     // one wave per SIMD, hand-placed gaps
    const bool hs = ppm == 3 && KS == 1;     // half-step form (fragment reads half a sub-step ahead)
    const bool as = ppm == 4 && KS == 1;     // hand-scheduled assembly form of the half-step kernel
    if (as && !asm_load()) { printf("ATTN_PP=4 needs ATTN_HSACO=<code object>\n"); return 1; }
    if (getenv("ATTN_CROSS")) a.cross = atoi(getenv("ATTN_CROSS"));
    if (getenv("ATTN_N1")) {                 // different counts for the two images of a pair
        for (auto& c : h) c.n[1] = atoi(getenv("ATTN_N1"));
        hipMemcpy(ctrl, h.data(), sizeof(LGCtrl) * B, hipMemcpyHostToDevice);
    }
    if (getenv("ATTN_CMP") && KS == 1) {
        // compare the hi / lo context planes of the variant against the 4-wave kernel on the same operands
        std::vector<_Float16> ref_hi(plane), ref_lo(plane), got_hi(plane), got_lo(plane);
        hipMemset(buf[6], 0, plane * 2); hipMemset(buf[7], 0, plane * 2);
        hipLaunchKernelGGL(lg_attention_p_kernel, grid, dim3(256), 0, 0, a);
        hipDeviceSynchronize();
        hipMemcpy(ref_hi.data(), buf[6], plane * 2, hipMemcpyDeviceToHost); hipMemcpy(ref_lo.data(), buf[7], plane * 2, hipMemcpyDeviceToHost);
        hipMemset(buf[6], 0, plane * 2); hipMemset(buf[7], 0, plane * 2);
        if (w1) hipLaunchKernelGGL(lg_attention_w1_kernel, dim3(sslam::cdiv(Kc, AQ2), NI * NH), dim3(256), 0, 0, a);
        else if (hs) hipLaunchKernelGGL(lg_attention_hs_kernel, dim3(sslam::cdiv(Kc, AQ), NI * NH), dim3(256), 0, 0, a);
        else if (as) asm_launch(a, NI);
        else hipLaunchKernelGGL(lg_attention_pp_kernel, dim3(sslam::cdiv(Kc, AQ2), NI * NH), dim3(512), 0, 0, a);
        hipDeviceSynchronize();
        printf("variant launch: %s\n", hipGetErrorString(hipGetLastError()));
        hipMemcpy(got_hi.data(), buf[6], plane * 2, hipMemcpyDeviceToHost); hipMemcpy(got_lo.data(), buf[7], plane * 2, hipMemcpyDeviceToHost);
        // panel layout: plane[k / PANEL_K][row][k % PANEL_K] over NI * Kc rows, k = head * 64 + d
        size_t bad = 0, bits = 0; double worst = 0; int shown = 0;
        for (size_t i = 0; i < plane; ++i)
            bits += memcmp(&ref_hi[i], &got_hi[i], 2) != 0 || memcmp(&ref_lo[i], &got_lo[i], 2) != 0;
        printf("bitwise: %zu of %zu (hi, lo) words differ\n", bits, plane);
        const size_t rows = (size_t)NI * Kc;
        for (size_t pnl = 0; pnl < 4; ++pnl)
            for (size_t row = 0; row < rows; ++row)
                for (int d = 0; d < 64; ++d) {
                    const size_t o = sslam::panel_index((int)row, (int)(pnl * 64 + d), (int)rows);
                    const double r = (double)ref_hi[o] + (double)ref_lo[o] / 2048.0, g = (double)got_hi[o] + (double)got_lo[o] / 2048.0;
                    const double e = fabs(r - g);
                    if (!(e <= 1e-6 + 1e-5 * fabs(r))) {
                        ++bad;
                        if (shown < 24) { printf("  head %zu img %zu row %zu d %d: ref %.7g got %.7g\n", pnl, row / Kc, row % Kc, d, r, g); ++shown; }
                    }
                    if (e == e) worst = fmax(worst, e);
                }
        printf("compare: %zu of %zu values differ, worst |diff| %.3g\n", bad, plane, worst);
        return 0;
    }
    if (as && getenv("ATTN_STAMP")) {
        // diagnostic code object (ATTN_ASM_STAMP=1): per-wave cycle counters {lifetime, barrier wait, ODD wait, EVEN wait, pair cost, tiles}
        const int nw = sslam::cdiv(Kc, AQ) * NI * NH * 4;
        hipMalloc(&g_dbg, (size_t)nw * 32); hipMemset(g_dbg, 0, (size_t)nw * 32);
        for (int i = 0; i < 3; ++i) asm_launch(a, NI);
        hipDeviceSynchronize();
        std::vector<unsigned> hdbg((size_t)nw * 8);
        hipMemcpy(hdbg.data(), g_dbg, hdbg.size() * 4, hipMemcpyDeviceToHost);
        double s[6] = {0, 0, 0, 0, 0, 0}; int live = 0;
        for (int w = 0; w < nw; ++w) if (hdbg[8 * w]) { ++live; for (int k = 0; k < 6; ++k) s[k] += hdbg[8 * w + k]; }
        for (double& x : s) x /= live;
        const double stamps = 4 * s[5];          // stamp pairs per kind: barrier 1 per tile, ODD 2, EVEN 2
        printf("stamps over %d waves: lifetime %.0f cycles, %.0f tiles; per tile: barrier wait %.0f, fragment wait ODD %.0f EVEN %.0f (two each), "
               "empty stamp pair %.0f -> net per tile: barrier %.0f, ODD %.0f, EVEN %.0f of %.0f cycles\n", live, s[0], s[5], s[1] / s[5], s[2] / s[5],
               s[3] / s[5], s[4], s[1] / s[5] - s[4], s[2] / s[5] - 2 * s[4], s[3] / s[5] - 2 * s[4], s[0] / s[5]);
        (void)stamps;
        return 0;
    }
    auto launch = [&] {
        if (w1) hipLaunchKernelGGL(lg_attention_w1_kernel, dim3(sslam::cdiv(Kc, AQ2), NI * NH), dim3(256), 0, 0, a);
        else if (hs) hipLaunchKernelGGL(lg_attention_hs_kernel, dim3(sslam::cdiv(Kc, AQ), NI * NH), dim3(256), 0, 0, a);
        else if (as) asm_launch(a, NI);
        else if (pp) hipLaunchKernelGGL(lg_attention_pp_kernel, dim3(sslam::cdiv(Kc, AQ2), NI * NH), dim3(512), 0, 0, a);
        else hipLaunchKernelGGL(lg_attention_p_kernel, grid, dim3(256), 0, 0, a);
    };
    for (int i = 0; i < 5; ++i) launch();
    hipDeviceSynchronize();
    std::vector<float> t;
    for (int r = 0; r < rounds; ++r) {
        const int R = 20;
        hipEventRecord(e0);
        for (int i = 0; i < R; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        t.push_back(ms / R * 1e3f);
    }
    std::sort(t.begin(), t.end());
    const double fl = 8.0 * N * (double)N * 256 * B;
    printf("%s N=%d pairs=%d KS=%d: median %.1f us/launch (min %.1f) = %.2f us per pair, %.0f TF alg, executed %.1f%% of the f16 peak; err=%s\n",
           w1 ? "[w1]" : hs ? "[hs]" : as ? "[asm]" : pp ? "[pp]" : "[p4]", N, B, KS,
           t[t.size() / 2], t[0], t[t.size() / 2] / B, fl / (t[t.size() / 2] * 1e-6) / 1e12, 3 * fl / (t[t.size() / 2] * 1e-6) / 2.5e15 * 100,
           hipGetErrorString(hipGetLastError()));
    return 0;
}
