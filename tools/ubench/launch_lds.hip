// Microbenchmark: event-timed duration of a trivial 256 x 512-thread launch as a function of its static LDS size (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KB>
__global__ __launch_bounds__(512) void k(float *out) {
    __shared__ float lds[KB * 256];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (lds[(threadIdx.x + 1) & 511] == 12345.f) out[0] = 1.f;
}
template <int KB>
void run() {
    float *out; (void)hipMalloc(&out, 64);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<KB>, dim3(256), dim3(512), 0, 0, out);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 20; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<KB>, dim3(256), dim3(512), 0, 0, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // back-to-back: 50 launches between one event pair
    (void)hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k<KB>, dim3(256), dim3(512), 0, 0, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms50; (void)hipEventElapsedTime(&ms50, e0, e1);
    printf("static LDS %3d KB: single launch between two events %.1f us (best of 20); 50 back-to-back: %.1f us each\n", KB, best * 1e3, ms50 * 1e3 / 50);
}
int main() { run<2>(); run<64>(); run<128>(); run<159>(); return 0; }
