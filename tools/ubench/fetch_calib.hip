// Calibration for rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 with the DAS kernel's access shape:
// each half-wavefront reads 32 consecutive dwords (one 128-B line) per instruction and writes likewise.
// Known byte counts: read_dword reads BYTES, copy_dword reads and writes BYTES, copy_x4 the same with 16 B/lane.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read_dword(const float *in, float *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i < n; i += stride) acc += in[i];
    if (acc == 123.456f) out[0] = acc;
}
__global__ void copy_dword(const float *in, float *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}
__global__ void copy_x4(const float4 *in, float4 *out, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) out[i] = in[i];
}
struct x3 { unsigned a, b, c; };  // 12 bytes per lane: the z48 packed-spectrum element of the mvdr / lcmv chain
__global__ void copy_x3(const x3 *in, x3 *out, size_t n3) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n3; i += stride) out[i] = in[i];
}
int main() {
    const size_t bytes = (size_t)1 << 30;  // 1 GiB, well past the 256 MiB Infinity Cache
    float *a, *b;
    (void)hipMalloc(&a, bytes);
    (void)hipMalloc(&b, bytes);
    (void)hipMemset(a, 1, bytes);
    (void)hipMemset(b, 0, bytes);
    const size_t n = bytes / 4;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read_dword, dim3(4096), dim3(256), 0, 0, a, b, n);
        hipLaunchKernelGGL(copy_dword, dim3(4096), dim3(256), 0, 0, a, b, n);
        hipLaunchKernelGGL(copy_x4, dim3(4096), dim3(256), 0, 0, (const float4 *)a, (float4 *)b, n / 4);
        hipLaunchKernelGGL(copy_x3, dim3(4096), dim3(256), 0, 0, (const x3 *)a, (x3 *)b, n / 3);
    }
    (void)hipDeviceSynchronize();
    printf("known bytes per kernel: read %zu, write %zu (copy kernels)\n", bytes, bytes);
    return 0;
}
