// Stand-alone timing of the ALIKED dense 3x3 conv kernels (same TU as the product kernels).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I opencv-simpleslam_amd/csrc [-DCONV_ABL=..] \
//         scripts/ubench/conv_bench.hip opencv-simpleslam_amd/csrc/context.hip -o /tmp/conv_bench
#include "../../opencv-simpleslam_amd/csrc/aliked_kernels.hip"
#include <cstdio>

__global__ void fillf(float* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = ((x & 0xffff) / 65536.0f - 0.5f);
}

template <typename F>
static float time_it(F f, int R = 30) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < R; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / R * 1e3f;
}

int main() {
    const int Hp = 320, Wp = 1024, H2 = 160, W2 = 512;
    float *a, *b, *c, *w, *al, *be, *wd, *bd;
    hipMalloc(&a, (size_t)32 * Hp * Wp * 4); hipMalloc(&b, (size_t)32 * Hp * Wp * 4); hipMalloc(&c, (size_t)32 * Hp * Wp * 4);
    hipMalloc(&w, 32 * 9 * 32 * 4); hipMalloc(&al, 128); hipMalloc(&be, 128); hipMalloc(&wd, 32 * 32 * 4); hipMalloc(&bd, 128);
    fillf<<<(32 * Hp * Wp + 255) / 256, 256>>>(a, (size_t)32 * Hp * Wp, 1);
    fillf<<<(32 * 9 * 32 + 255) / 256, 256>>>(w, 32 * 9 * 32, 2);
    fillf<<<1, 32>>>(al, 32, 3); fillf<<<1, 32>>>(be, 32, 4); fillf<<<4, 256>>>(wd, 1024, 5); fillf<<<1, 32>>>(bd, 32, 6);
    dim3 g1(sslam::cdiv(Wp, CT_W), sslam::cdiv(Hp, 16)), g2(sslam::cdiv(W2, CT_W), sslam::cdiv(H2, 8));
    float t;
    t = time_it([&] { hipLaunchKernelGGL((al_conv3x3_mfma_kernel<3, 16, 1, false, false, 4>), g1, dim3(256), 0, 0, a, Hp, Wp, b, Hp, Wp, w, al, be, nullptr, nullptr, nullptr, nullptr); });
    printf("conv  3->16 full res : %7.1f us\n", t);
    t = time_it([&] { hipLaunchKernelGGL((al_conv3x3_mfma_kernel<16, 16, 1, false, false, 4>), g1, dim3(256), 0, 0, a, Hp, Wp, b, Hp, Wp, w, al, be, nullptr, nullptr, nullptr, nullptr); });
    printf("conv 16->16 full res : %7.1f us  (1.51 GFLOP, 42 MB)\n", t);
    t = time_it([&] { hipLaunchKernelGGL((al_conv3x3_mfma_kernel<16, 32, 2, true, false, 2>), g2, dim3(256), 0, 0, a, Hp, Wp, b, H2, W2, w, al, be, wd, bd, c, nullptr); });
    printf("conv 16->32 pool+down: %7.1f us\n", t);
    t = time_it([&] { hipLaunchKernelGGL((al_conv3x3_mfma_kernel<32, 32, 1, false, true, 2>), g2, dim3(256), 0, 0, a, H2, W2, b, H2, W2, w, al, be, nullptr, nullptr, nullptr, c); });
    printf("conv 32->32 half res : %7.1f us  (1.51 GFLOP, 31 MB)\n", t);
    printf("err=%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
