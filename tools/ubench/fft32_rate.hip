// How fast does the in-register FFT-32 instruction stream itself issue (no LDS, no global traffic)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../beamform_amd/csrc/fft32.hpp"
using namespace bf;
template <int WPS>
__global__ __launch_bounds__(256, WPS) void k(float *out, int iters, float seed) {
    float re[32], im[32];
    for (int i = 0; i < 32; ++i) { re[i] = seed + i * 0.01f + threadIdx.x * 1e-4f; im[i] = seed - i * 0.02f; }
    for (int it = 0; it < iters; ++it) {
        fft32_dif<float, -1>(re, im);
        fft32_dit<float, +1>(re, im);
        for (int i = 0; i < 32; ++i) { re[i] *= 0.03125f; im[i] *= 0.03125f; }
    }
    float acc = 0;
    for (int i = 0; i < 32; ++i) acc += re[i] + im[i];
    if (acc == 1234.5f) out[threadIdx.x] = acc;
}
template <int WPS>
void run(int instr_per_iter) {
    float *out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    const int blocks = 256 * WPS;
    hipLaunchKernelGGL(k<WPS>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<WPS>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double inst_per_simd = (double)iters * instr_per_iter * WPS;
    printf("waves/SIMD=%d: %.3f ms, %.2f ns per VALU instr per SIMD -> %.2f cycles @2.0GHz, %.2f @2.4GHz\n", WPS, ms,
           ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.0, ms * 1e6 / inst_per_simd * 2.4);
}
int main(int argc, char **argv) {
    int n = argc > 1 ? atoi(argv[1]) : 840;
    run<1>(n); run<2>(n); run<4>(n);
    return 0;
}
