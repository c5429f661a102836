// Microbenchmark: issue cost of the instructions the fp64 das kernels are made of, at 1 / 2 / 4 wavefronts per SIMD (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(double *out, int iters, double seed) {
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    float f0 = (float)seed, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    const double c = 1.0000001, d = 0.5;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0)
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            else if (MODE == 1)
                asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                             "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            else if (MODE == 2)
                asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                             "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            else if (MODE == 3)
                asm volatile("v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %9\n v_cvt_f64_f32 %2, %10\n v_cvt_f64_f32 %3, %11\n"
                             "v_cvt_f64_f32 %4, %8\n v_cvt_f64_f32 %5, %9\n v_cvt_f64_f32 %6, %10\n v_cvt_f64_f32 %7, %11\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));
            else if (MODE == 4)
                asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n"
                             "v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
            else if (MODE == 5)
                asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane16_swap_b32 %0, %2\n v_permlane16_swap_b32 %1, %3\n"
                             "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane16_swap_b32 %0, %2\n v_permlane16_swap_b32 %1, %3\n"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if (MODE == 6)
                asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %4\n"
                             "v_mov_b64 %4, %5\n v_mov_b64 %5, %6\n v_mov_b64 %6, %7\n v_mov_b64 %7, %0\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            else if (MODE == 8)  // v_fmac_f64 with a DPP row_newbcast source: the only fp64 arithmetic gfx950 encodes with DPP (VOP2), broadcast fused into the FMA
                asm volatile("v_fmac_f64_dpp %0, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                             "v_fmac_f64_dpp %2, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %8, %9 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
                             "v_fmac_f64_dpp %4, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                             "v_fmac_f64_dpp %6, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            else if (MODE == 9)  // v_mov_b64 with DPP row_newbcast (one instruction per double; the b32 form needs two)
                asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b64_dpp %2, %3 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %3, %4 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b64_dpp %4, %5 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %5, %6 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b64_dpp %6, %7 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %7, %0 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            else if (MODE == 10)  // plain v_fmac_f64 (VOP2) for comparison with MODE 8
                asm volatile("v_fmac_f64 %0, %8, %9\n v_fmac_f64 %1, %8, %9\n v_fmac_f64 %2, %8, %9\n v_fmac_f64 %3, %8, %9\n"
                             "v_fmac_f64 %4, %8, %9\n v_fmac_f64 %5, %8, %9\n v_fmac_f64 %6, %8, %9\n v_fmac_f64 %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            else if (MODE == 7)  // fp64 fma with two constant operands from SGPRs, as the fused-twiddle butterflies have them
                asm volatile("v_fma_f64 %0, %0, %8, %1\n v_fma_f64 %1, %1, %8, %2\n v_fma_f64 %2, %2, %8, %3\n v_fma_f64 %3, %3, %8, %4\n"
                             "v_fma_f64 %4, %4, %8, %5\n v_fma_f64 %5, %5, %8, %6\n v_fma_f64 %6, %6, %8, %7\n v_fma_f64 %7, %7, %8, %0\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(c));
        }
    }
    double r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + u0 + u1 + u2 + u3;
    if (r == 12345.678) out[threadIdx.x] = r;
}
template <int MODE>
void run(const char *name) {
    double *out; (void)hipMalloc(&out, 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 8000;
    for (int wps : {1, 2, 4}) {
        int threads = 256, blocks = 256 * wps;
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 100, 1.0);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double per_simd_per_s = (double)iters * 16 * 8 * blocks * 4 / (256.0 * 4) / (ms * 1e-3);
        printf("%-18s waves/SIMD=%d  %.3f ms  cycles per wave-instruction per SIMD @2.4GHz: %.2f\n", name, wps, ms, 2.4e9 / per_simd_per_s);
    }
}
int main() {
    run<0>("v_fma_f64"); run<1>("v_add_f64"); run<2>("v_mul_f64"); run<7>("v_fma_f64 sgpr"); run<3>("v_cvt_f64_f32"); run<4>("v_cvt_f32_f64");
    run<5>("v_permlane_swap"); run<6>("v_mov_b64");
    run<10>("v_fmac_f64"); run<8>("v_fmac_f64_dpp"); run<9>("v_mov_b64_dpp");
    return 0;
}
