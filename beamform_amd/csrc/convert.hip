// convert.hip -- float32 -> PCM16 of a batch that is resident in HBM: the sample rule of sf_write_float on a PCM_16 file
// (rosjack.cpp:404-409; libsndfile f2s_array: lrintf(x * 32767.0f) stored as short, no clipping -- see wavio.cpp), so that the
// output stage copies 2 bytes per sample to the host instead of 4.  Pure streaming: 4 B in + 2 B out per sample.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "launch_trace.hpp"
#include "../../include/bfcore.h"

namespace {

__device__ __forceinline__ int16_t f2s(float v) {
#pragma clang fp contract(off)
    const float scaled = v * 32767.0f;
    return (int16_t)(uint16_t)(unsigned long long)__float2ll_rn(scaled);  // nearest-even, then the modulo-2^16 cast to short
}

__global__ __launch_bounds__(256) void pcm16_kernel(const float *__restrict__ src, int16_t *__restrict__ dst, size_t n) {
    const size_t n8 = n / 8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const float4 a = reinterpret_cast<const float4 *>(src)[2 * i], b = reinterpret_cast<const float4 *>(src)[2 * i + 1];
        union { int16_t s[8]; uint4 v; } o;
        o.s[0] = f2s(a.x); o.s[1] = f2s(a.y); o.s[2] = f2s(a.z); o.s[3] = f2s(a.w);
        o.s[4] = f2s(b.x); o.s[5] = f2s(b.y); o.s[6] = f2s(b.z); o.s[7] = f2s(b.w);
        reinterpret_cast<uint4 *>(dst)[i] = o.v;
    }
    if (blockIdx.x == 0)
        for (size_t i = n8 * 8 + threadIdx.x; i < n; i += 256) dst[i] = f2s(src[i]);
}

}  // namespace

extern "C" int bf_float_to_pcm16_device(const float *src_dev, int16_t *dst_dev, size_t n, void *hip_stream) {
    if (!src_dev || !dst_dev) return BF_EINVAL;
    if (n == 0) return BF_OK;
    if (((uintptr_t)src_dev & 15) || ((uintptr_t)dst_dev & 15)) return BF_EINVAL;  // 16-byte aligned buffers (hipMalloc gives 256)
    size_t blocks = (n / 8 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    BF_LAUNCH(pcm16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)hip_stream, src_dev, dst_dev, n);
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EIO;
}
