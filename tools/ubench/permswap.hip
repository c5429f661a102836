// permswap.hip -- lane semantics of v_permlane32_swap / v_permlane16_swap on gfx950 (used by the 64-lane FFT's row transpose)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
    const unsigned l = threadIdx.x;
    unsigned a = 1000 + l, b = 2000 + l;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[l] = r[0];
    out[64 + l] = r[1];
    auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[128 + l] = q[0];
    out[192 + l] = q[1];
}
int main() {
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[4] = {"permlane32_swap vdst'", "permlane32_swap src0'", "permlane16_swap vdst'", "permlane16_swap src0'"};
    for (int v = 0; v < 4; ++v) {
        printf("%s:", names[v]);
        for (int row = 0; row < 4; ++row) printf("  row%d=%u..%u", row, h[64 * v + 16 * row], h[64 * v + 16 * row + 15]);
        printf("\n");
    }
    return 0;
}
