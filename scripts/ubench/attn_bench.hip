// Stand-alone timing of the LightGlue attention kernel (same TU as the product kernels, synthetic
// operands, no downstream kernels): used for ablations that would poison a full match.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I opencv-simpleslam_amd/csrc [-DATTN_ABL=..] \
//         scripts/ubench/attn_bench.hip -o /tmp/attn_bench && /tmp/attn_bench [N] [KS] [kernel 0|1]
#include "../../opencv-simpleslam_amd/csrc/lightglue_kernels.hip"
#include <cstdio>
#include <vector>

__global__ void fill_half(_Float16* p, size_t n, unsigned seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = (_Float16)(((x & 0xffff) / 65536.0f - 0.5f) * scale);
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 2048;
    const int KS = argc > 2 ? atoi(argv[2]) : 4;
    const int which = argc > 3 ? atoi(argv[3]) : 1;
    const int batch = argc > 4 ? atoi(argv[4]) : 1;      // needs -DATTN_BATCH_EMU when > 1
    const int Kc = (N + 127) / 128 * 128;
    const size_t plane = (size_t)2 * NH * Kc * DH;
    _Float16* buf[6];
    for (int i = 0; i < 6; ++i) {
        hipMalloc(&buf[i], plane * 2);
        fill_half<<<(plane + 255) / 256, 256>>>(buf[i], plane, 17 * i + 1, (i & 1) ? 0.01f : 2.0f);
    }
    float *o_part, *m_part, *l_part; LGCtrl* ctrl;
    hipMalloc(&o_part, (size_t)KS * plane * 4); hipMalloc(&m_part, (size_t)KS * 2 * NH * Kc * 4);
    hipMalloc(&l_part, (size_t)KS * 2 * NH * Kc * 4); hipMalloc(&ctrl, sizeof(LGCtrl));
    LGCtrl h{}; h.n[0] = h.n[1] = N; hipMemcpy(ctrl, &h, sizeof(h), hipMemcpyHostToDevice);
    AttnArgsH a{{buf[0], buf[1]}, {buf[2], buf[3]}, {buf[4], buf[5]}, 0, o_part, m_part, l_part, KS, Kc, ctrl};
    dim3 grid(sslam::cdiv(Kc, AQ), 2 * NH * batch, KS);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&] {
        (void)which;
        hipLaunchKernelGGL(lg_attention_p_kernel, grid, dim3(256), 0, 0, a);
    };
    for (int i = 0; i < 5; ++i) launch();
    hipDeviceSynchronize();
    const int R = 50;
    hipEventRecord(e0);
    for (int i = 0; i < R; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // isolated launches (event pair around each)
    float iso = 0;
    for (int i = 0; i < 20; ++i) {
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float m; hipEventElapsedTime(&m, e0, e1); iso += m;
    }
    const double fl = 8.0 * N * (double)N * 256;
    printf("N=%d KS=%d kernel=%d abl=%d batch=%d: back-to-back %.2f us/launch = %.2f us per pair (%.0f TF alg), isolated %.2f us; err=%s\n", N, KS, which,
           ATTN_ABL, batch, ms / R * 1e3, ms / R * 1e3 / batch, batch * fl / (ms / R * 1e-3) / 1e12, iso / 20 * 1e3, hipGetErrorString(hipGetLastError()));
    return 0;
}
