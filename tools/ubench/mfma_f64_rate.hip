// mfma_f64_rate.hip -- issue rate of v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64 against v_fma_f64 on gfx950:
// the numbers behind "does the 16 x 16 covariance slide / Cholesky trailing update of lcmv-16 belong on the matrix pipe?"
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k_mfma16(double *out, int iters) {
    double4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3];
}
__global__ void k_mfma4(double *out, int iters) {
    double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0 + acc1 + acc2 + acc3;
}
__global__ void k_fma(double *out, int iters) {
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-9;
    double c0 = 0, c1 = 1, c2 = 2, c3 = 3, c4 = 4, c5 = 5, c6 = 6, c7 = 7;
    for (int i = 0; i < iters; ++i) {
        c0 = fma(a, b, c0); c1 = fma(a, b, c1); c2 = fma(a, b, c2); c3 = fma(a, b, c3);
        c4 = fma(a, b, c4); c5 = fma(a, b, c5); c6 = fma(a, b, c6); c7 = fma(a, b, c7);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}
template <typename K>
static double run(K k, int waves_per_simd, int iters, double flops_per_wave_iter, const char *name) {
    double *d;
    (void)hipMalloc(&d, sizeof(double) * 256 * 4 * 64 * waves_per_simd);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd;  // 256-thread blocks = one wave per SIMD each
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double tf = flops_per_wave_iter * iters * blocks * 4 / (ms * 1e-3) / 1e12;
    printf("%-26s %d wave(s)/SIMD: %.3f ms, %.1f TFLOP/s\n", name, waves_per_simd, ms, tf);
    (void)hipFree(d);
    return tf;
}
int main() {
    const int it = 20000;
    for (int w = 1; w <= 2; ++w) {
        run(k_mfma16, w, it, 4.0 * 16 * 16 * 4 * 2, "v_mfma_f64_16x16x4_f64");
        run(k_mfma4, w, it, 4.0 * 4 * (4 * 4 * 4 * 2), "v_mfma_f64_4x4x4_4b_f64");
        run(k_fma, w, it, 8.0 * 64 * 2, "v_fma_f64");
    }
    return 0;
}
