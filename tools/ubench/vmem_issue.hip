// Microbenchmark: issue cost of 64 back-to-back dword loads per wave for the addressing forms of global_load / buffer_load
// (L2-resident data), at 1 / 4 / 8 waves per CU.  Cycles from s_memtime around the issue run and around issue + drain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X, base) X(base + 0) X(base + 1) X(base + 2) X(base + 3) X(base + 4) X(base + 5) X(base + 6) X(base + 7) X(base + 8) X(base + 9) X(base + 10) X(base + 11) X(base + 12) X(base + 13) X(base + 14) X(base + 15)

template <int MODE>
__global__ __launch_bounds__(512) void k(const float *x, float *out, unsigned long long *cyc, int reps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *p = x + (size_t)(blockIdx.x * 8 + wave) * 8192 + lane;  // 32 KiB per wave
    float v[64];
    unsigned long long t_issue = 0, t_all = 0;
    float acc = 0.f;
    const unsigned voff = (unsigned)((blockIdx.x * 8 + wave) * 8192 + lane) * 4u;
    for (int r = 0; r < reps; ++r) {
        __builtin_amdgcn_s_barrier();
        const unsigned long long t0 = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 0) {  // 64-bit VGPR address
#pragma unroll
            for (int j = 0; j < 64; ++j) asm volatile("global_load_dword %0, %1, off offset:%c2" : "=v"(v[j]) : "v"(p + (j >> 5) * 4096), "i"((j & 31) * 128));
        } else if (MODE == 1) {  // SGPR base + 32-bit VGPR offset
#pragma unroll
            for (int j = 0; j < 64; ++j) asm volatile("global_load_dword %0, %1, %2 offset:%c3" : "=v"(v[j]) : "v"(voff + (j >> 5) * 16384), "s"(x), "i"((j & 31) * 128));
        } else if (MODE == 2) {  // dwordx4, 64-bit VGPR address, 16 instructions for the same bytes
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 *v4 = reinterpret_cast<f4 *>(v);
            const float *p4 = x + (size_t)(blockIdx.x * 8 + wave) * 8192 + lane * 4;
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("global_load_dwordx4 %0, %1, off offset:%c2" : "=v"(v4[j]) : "v"(p4 + (j >> 2) * 1024), "i"((j & 3) * 1024));
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_readcyclecounter();
#pragma unroll
        for (int j = 0; j < 64; ++j) { asm volatile("" : "+v"(v[j])); acc += v[j]; }
        if (r > 0) { t_issue += t1 - t0; t_all += t2 - t0; }
    }
    if (acc == 1234.5f) out[threadIdx.x] = acc;
    if (lane == 0) { atomicAdd(&cyc[0], t_issue); atomicAdd(&cyc[1], t_all); }
}

template <int MODE>
void run(const char *name, const float *x, float *out, unsigned long long *cyc) {
    for (int waves : {1, 4, 8}) {
        const int reps = 200, blocks = 256;
        (void)hipMemset(cyc, 0, 16);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64 * waves), 0, 0, x, out, cyc, reps);
        (void)hipDeviceSynchronize();
        unsigned long long h[2];
        (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        const double n = (double)blocks * waves * (reps - 1);
        printf("%-28s waves/CU=%d  issue %.0f cyc per 64 loads (%.1f per load)   issue+drain %.0f\n", name, waves, h[0] / n, h[0] / n / 64, h[1] / n);
    }
}
int main() {
    float *x, *out; unsigned long long *cyc;
    (void)hipMalloc(&x, (size_t)256 * 8 * 8192 * 4 + 65536); (void)hipMalloc(&out, 4096); (void)hipMalloc(&cyc, 16);
    (void)hipMemset(x, 0, (size_t)256 * 8 * 8192 * 4 + 65536);
    run<0>("dword, 64-bit vaddr", x, out, cyc);
    run<1>("dword, saddr + voffset", x, out, cyc);
    run<2>("dwordx4, 64-bit vaddr (16)", x, out, cyc);
    return 0;
}
