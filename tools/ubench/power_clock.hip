// Which instruction classes pull the shader clock down?  Every CU busy with two wavefronts per SIMD of ONE instruction class (operands that keep
// toggling), ~10 ms per launch; the clock the chip holds = s_memtime / s_memrealtime (100 MHz) per wavefront, averaged.  Output: time, clock,
// true cycles per wave-instruction per SIMD.  (das_f64_pair_kernel runs at 2.0-2.1 GHz on the full chip, 2.4 GHz on half of it: DESIGN.md 9.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(512) void k(double *out, unsigned long long *stamps, int iters, double seed, const unsigned long long *stream) {
    // modes 6-10: the mix of mode 4 with the pair kernel's share of LDS traffic (10 LDS instructions per 64 VALU: 4 ds_write_b64, 4 ds_read_b64,
    // 2 ds_read_b128) and / or of HBM reads (one 512-byte row per wavefront per 64 VALU ~ 3.9 TB/s on the chip; every wavefront walks its own 1 MiB)
    constexpr bool kMix = MODE == 4 || (MODE >= 6 && MODE <= 8), kLds = MODE == 6 || MODE == 8 || MODE == 9, kMem = MODE == 7 || MODE == 8 || MODE == 10;
    __shared__ double plane[8][16 * 65 + 64];
    const int lane = threadIdx.x;
    double *pl = &plane[threadIdx.x >> 6][threadIdx.x & 63];
    const unsigned long long *sp = stream + ((size_t)(blockIdx.x * 8 + (threadIdx.x >> 6)) << 17) + (threadIdx.x & 63);
    unsigned acc = 0;
    double a0 = seed + 0.37 * lane, a1 = a0 * 1.3 + 1, a2 = a0 * 0.7 - 2, a3 = a0 + 3.1, a4 = a0 - 4.7, a5 = a0 * 2.1, a6 = a0 + 6.3, a7 = a0 - 7.9;
    unsigned u0 = lane * 2654435761u, u1 = u0 ^ 0x5bd1e995u, u2 = u0 + 77, u3 = ~u0;
    const double c = -0.99999991, d = 0.6180339887 + 0.001 * lane, e = 1.000000119;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0)
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
            else if (MODE == 1)  // adds of registers to each other with alternating sign: values stay bounded, mantissas keep changing
                asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %1, %1, -%2\n v_add_f64 %2, %2, %3\n v_add_f64 %3, %3, -%0\n"
                             "v_add_f64 %4, %4, %5\n v_add_f64 %5, %5, -%6\n v_add_f64 %6, %6, %7\n v_add_f64 %7, %7, -%4\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            else if (MODE == 2)
                asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %9\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %9\n"
                             "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %9\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(e));
            else if (MODE == 3)
                asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane16_swap_b32 %0, %2\n v_permlane16_swap_b32 %1, %3\n"
                             "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane16_swap_b32 %0, %2\n v_permlane16_swap_b32 %1, %3\n"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            else if (kMix)  // the pair kernel's mix per 8: 3 add, 3 fma, 1 mul, 1 swap
                asm volatile("v_add_f64 %0, %0, %1\n v_fma_f64 %1, %1, %8, %9\n v_add_f64 %2, %2, -%3\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_mul_f64 %4, %4, %8\n v_add_f64 %5, %5, -%6\n v_fma_f64 %6, %6, %8, %9\n v_permlane32_swap_b32 %10, %11\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d), "v"(u0), "v"(u1));
            else if (MODE == 5)  // fp32 fma for comparison
                asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                             "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(0xbf7fffffu), "v"(0x3f1e377au));
        }
        if (kLds) {
            pl[0] = a0; pl[65] = a1; pl[130] = a2; pl[195] = a3;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            a4 += pl[260 + (i & 7)]; a5 += pl[325]; a6 += pl[390]; a7 += pl[455];
            typedef double d2 __attribute__((ext_vector_type(2)));
            const d2 g0 = *(const d2 *)((const char *)&plane[threadIdx.x >> 6][0] + (threadIdx.x & 63) * 16), g1 = *(const d2 *)((const char *)&plane[threadIdx.x >> 6][128] + (threadIdx.x & 63) * 16);
            a0 += g0.x * 1e-9; a1 += g0.y * 1e-9; a2 += g1.x * 1e-9; a3 += g1.y * 1e-9;
        }
        if (kMem) {
            const unsigned long long v = __builtin_nontemporal_load(sp + ((size_t)(i & 2047) << 6));
            acc ^= (unsigned)v ^ (unsigned)(v >> 32);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
    double r = acc + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + u0 + u1 + u2 + u3;
    if (r == 12345.678) out[threadIdx.x] = r;
}
template <int MODE>
void run(const char *name, int blocks) {
    double *out; (void)hipMalloc(&out, 8192);
    unsigned long long *st; (void)hipMalloc(&st, sizeof(unsigned long long) * 2 * 256 * 8);
    static unsigned long long *stream = nullptr;
    if (!stream) { (void)hipMalloc(&stream, (size_t)2048 << 20); (void)hipMemset(stream, 1, (size_t)2048 << 20); }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 40000;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, st, iters, 1.0, stream);   // settle
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, st, iters, 1.0, stream);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks * 8);
    (void)hipMemcpy(h.data(), st, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double clk = 0;
    for (int w = 0; w < blocks * 8; ++w) clk += (double)h[2 * w] / (double)h[2 * w + 1] * 0.1;
    clk /= blocks * 8;
    const double inst_per_simd = (double)iters * 8 * 8 * 2;  // two wavefronts per SIMD
    printf("%-28s CUs=%3d  %.3f ms  clock %.3f GHz  %.2f cycles per wave-instruction per SIMD\n", name, blocks, ms, clk, ms * 1e-3 * clk * 1e9 / inst_per_simd);
    (void)hipFree(out); (void)hipFree(st);
}
int main() {
    for (int blocks : {256, 128}) {
        run<0>("v_fma_f64", blocks); run<1>("v_add_f64", blocks); run<2>("v_mul_f64", blocks); run<3>("v_permlane swap", blocks);
        run<4>("mix 3 add 3 fma 1 mul 1 swap", blocks); run<5>("v_fma_f32", blocks);
        run<6>("mix + LDS", blocks); run<7>("mix + HBM reads", blocks); run<8>("mix + LDS + HBM reads", blocks); run<9>("LDS only", blocks); run<10>("HBM reads only", blocks);
    }
    return 0;
}
