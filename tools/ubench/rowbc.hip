#include <hip/hip_runtime.h>
#include <cstdio>
template <int N>
__device__ __forceinline__ int rowbc(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x150 + N, 0xF, 0xF, false); }
__global__ void k(int *out) {
    const int l = threadIdx.x;
    int v = l * 10;
    out[l] = rowbc<5>(v);
    out[64 + l] = rowbc<15>(v);
    out[128 + l] = rowbc<0>(v);
}
int main() {
    int *d; hipMalloc(&d, 192 * 4); k<<<1, 64>>>(d); int h[192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        int row = l / 16;
        if (h[l] != (row * 16 + 5) * 10) bad++;
        if (h[64 + l] != (row * 16 + 15) * 10) bad++;
        if (h[128 + l] != (row * 16) * 10) bad++;
    }
    printf("row_newbcast bad=%d  sample: %d %d %d %d\n", bad, h[0], h[17], h[64 + 33], h[128 + 50]);
    return bad != 0;
}
