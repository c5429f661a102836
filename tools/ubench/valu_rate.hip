// Microbenchmark: issue rate of scalar vs packed fp32 VALU ops on gfx950 at 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float *out, int iters, float seed) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    float2v p0 = {seed, seed}, p1 = {seed + 1, seed}, p2 = {seed + 2, seed}, p3 = {seed + 3, seed};
    float2v p4 = {seed + 4, seed}, p5 = {seed + 5, seed}, p6 = {seed + 6, seed}, p7 = {seed + 7, seed};
    const float c = 1.0001f, d = 0.5f;
    const float2v c2 = {1.0001f, 0.9999f}, d2 = {0.5f, 0.25f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) {  // scalar fma, 8 independent chains
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            } else if (MODE == 1) {  // packed fma
                asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                             "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2), "v"(d2));
            } else if (MODE == 2) {  // scalar add
                asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                             "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            } else if (MODE == 3) {  // packed add
                asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                             "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));
            } else if (MODE == 4) {  // v_mov
                asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                             "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (MODE == 5) {  // fp64 fma
                asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c2), "v"(d2));
            } else if (MODE == 6) {  // dependent chain scalar fma (latency)
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             : "+v"(a0) : "v"(c), "v"(d));
            }
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.x + p5.x + p6.x + p7.x + p0.y + p1.y + p2.y + p3.y;
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int MODE>
void run(const char *name, int insts_per_iter) {
    float *out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps : {1, 2, 4, 8}) {
        int threads = 256;                 // 4 waves per block -> 1 wave per SIMD per block
        int blocks = 256 * wps;            // wps blocks per CU
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 100, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double inst_per_wave = (double)iters * 16 * insts_per_iter;
        double total_wave_inst = inst_per_wave * blocks * 4;
        double per_simd_per_s = total_wave_inst / (256.0 * 4) / (ms * 1e-3);
        printf("%-14s waves/SIMD=%d  %.3f ms  %.3f G wave-inst/s/SIMD  (cycles/inst @2.4GHz: %.2f)\n", name, wps, ms,
               per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
    }
}
int main() {
    run<0>("v_fma_f32", 8); run<1>("v_pk_fma_f32", 8); run<2>("v_add_f32", 8); run<3>("v_pk_add_f32", 8);
    run<4>("v_mov_b32", 8); run<5>("v_fma_f64", 4); run<6>("fma_f32 dep", 8);
    return 0;
}
