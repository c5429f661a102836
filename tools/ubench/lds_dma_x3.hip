// Where does global_load_lds_dwordx3 put lane l's 12 bytes?  Dumps the LDS image after one wave-wide DMA of 64 x 12 bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned *g, unsigned *o) {
    __shared__ __attribute__((aligned(16))) unsigned s[512];
    for (int i = threadIdx.x; i < 512; i += 64) s[i] = 0xdeadbeefu;
    __syncthreads();
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + 3 * threadIdx.x),
                                     (__attribute__((address_space(3))) void *)&s[0], 12, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) o[i] = s[i];
}
int main() {
    std::vector<unsigned> h(192), r(512);
    for (int i = 0; i < 192; ++i) h[i] = ((i / 3) << 8) | (i % 3);  // lane << 8 | dword
    unsigned *g, *o;
    hipMalloc(&g, 192 * 4); hipMalloc(&o, 512 * 4);
    hipMemcpy(g, h.data(), 192 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o);
    hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < 272; ++i) printf("%s%06x", i % 16 ? " " : "\n", r[i]);
    printf("\n");
    return 0;
}
