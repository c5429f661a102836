// Calibrates s_memrealtime against the host's event clock, and s_memtime against it (idle chip, s_sleep loop).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long ticks, unsigned long long *out) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memrealtime() - r0 < ticks) __builtin_amdgcn_s_sleep(8);
    out[0] = __builtin_amdgcn_s_memrealtime() - r0;
    out[1] = __builtin_amdgcn_s_memtime() - t0;
}
int main() {
    unsigned long long *d, h[2];
    (void)hipMalloc(&d, 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (unsigned long long ticks : {100000ull, 1000000ull, 5000000ull}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, 1000ull, d);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, ticks, d);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%llu realtime ticks, %llu memtime ticks in %.3f ms by the events: s_memrealtime %.2f MHz, s_memtime %.3f GHz (idle, sleeping)\n", h[0], h[1], ms,
               h[0] / (ms * 1e3), h[1] / (ms * 1e6));
    }
    return 0;
}
