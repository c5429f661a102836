// das_fused_w64.hip -- fused fp32 delay-and-sum, 64-lane x 16-point-per-lane FFT (fft1024_w64.hpp).
//
// Same algorithm, boundary and state handling as das_fused.hip; the difference is the FFT
// factorisation: one full wavefront per frame, 16 complex points per lane, three passes
// (16 x 16 x 4) with one LDS transpose and one in-quad DPP transpose.  About 100 VGPRs instead
// of 213, so a CU holds 16 wavefronts = 4 per SIMD (one 1024-thread block walking 16 consecutive
// frames per iteration) and the VALU no longer idles whenever one of two wavefronts waits.
#include <hip/hip_runtime.h>

#include "fft1024.hpp"
#include "fft1024_w64.hpp"
#include "kernels.hpp"

namespace bf {

namespace {

constexpr int kBlock = 1024;
constexpr int kWaves = kBlock / 64;
constexpr int kHop = 512;
constexpr int kNfft = 1024;
constexpr int kRS = 68;                      // T1 plane row stride (floats): conflict-free 4-byte reads
constexpr int kPlane = 16 * kRS;             // floats per wavefront
constexpr int kWinStride = 20;               // floats per lane row of the window table
constexpr int kLdsTw = 2 * (1024 + 64);      // floats
constexpr int kLdsFixed = kLdsTw + kWaves * kPlane + 64 * kWinStride;

__device__ __forceinline__ float dpp_xor1(float x) {
    const int v = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_xor2(float x) {
    const int v = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));
}
// 4x4 transpose inside a quad of lanes: afterwards lane l holds r'[c] = r[l] of lane c
__device__ __forceinline__ void quad_transpose(float &r0, float &r1, float &r2, float &r3, bool o1, bool o2) {
    // partner values are fetched unconditionally (every lane must execute the DPP moves)
    const float p0 = dpp_xor1(r0), p1 = dpp_xor1(r1), p2 = dpp_xor1(r2), p3 = dpp_xor1(r3);
    const float a0 = o1 ? p1 : r0;
    const float a1 = o1 ? r1 : p0;
    const float a2 = o1 ? p3 : r2;
    const float a3 = o1 ? r3 : p2;
    const float q0 = dpp_xor2(a0), q1 = dpp_xor2(a1), q2 = dpp_xor2(a2), q3 = dpp_xor2(a3);
    r0 = o2 ? q2 : a0;
    r2 = o2 ? a2 : q0;
    r1 = o2 ? q3 : a1;
    r3 = o2 ? a3 : q1;
}
constexpr int brev2c(int i) { return ((i & 1) << 1) | ((i >> 1) & 1); }

// T2: position brev2(g) + 4*brev2(q) (lane field b)  <->  register 4*g + b (lane field q)
template <bool FWD>
__device__ __forceinline__ void w64_T2(float (&re)[16], float (&im)[16], int lane) {
    const bool o1 = lane & 1, o2 = lane & 2;
    float nr[16], ni[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float r[4], s[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int src = FWD ? brev2c(g) + 4 * brev2c(c) : 4 * g + c;
            r[c] = re[src];
            s[c] = im[src];
        }
        quad_transpose(r[0], r[1], r[2], r[3], o1, o2);
        quad_transpose(s[0], s[1], s[2], s[3], o1, o2);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int dst = FWD ? 4 * g + c : brev2c(g) + 4 * brev2c(c);
            nr[dst] = r[c];
            ni[dst] = s[c];
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        re[i] = nr[i];
        im[i] = ni[i];
    }
}

// T1 through one scalar LDS plane (real plane, then imaginary plane)
__device__ __forceinline__ void w64_T1_fwd(float (&re)[16], float (&im)[16], int lane, float *pl) {
    const int rd = (lane >> 2) * kRS + (lane & 3);
#pragma unroll
    for (int i = 0; i < 16; ++i) pl[brev4(i) * kRS + lane] = re[i];
#pragma unroll
    for (int a = 0; a < 16; ++a) re[a] = pl[rd + 4 * a];
#pragma unroll
    for (int i = 0; i < 16; ++i) pl[brev4(i) * kRS + lane] = im[i];
#pragma unroll
    for (int a = 0; a < 16; ++a) im[a] = pl[rd + 4 * a];
}
__device__ __forceinline__ void w64_T1_inv(float (&re)[16], float (&im)[16], int lane, float *pl) {
    const int wr = (lane >> 2) * kRS + (lane & 3);
#pragma unroll
    for (int a = 0; a < 16; ++a) pl[wr + 4 * a] = re[a];
#pragma unroll
    for (int i = 0; i < 16; ++i) re[i] = pl[brev4(i) * kRS + lane];
#pragma unroll
    for (int a = 0; a < 16; ++a) pl[wr + 4 * a] = im[a];
#pragma unroll
    for (int i = 0; i < 16; ++i) im[i] = pl[brev4(i) * kRS + lane];
}

template <int LAYOUT, int NPL>
__global__ __launch_bounds__(kBlock) void das_fused_w64_kernel(DasFusedArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[kLdsFixed + NPL * 2048 + 17 * kHop];
    const cx<float> *s_tw1 = reinterpret_cast<const cx<float> *>(lds);
    const cx<float> *s_tw2 = s_tw1 + 1024;
    float *s_win = lds + kLdsTw + kWaves * kPlane;
    const cx<float> *s_gain = reinterpret_cast<const cx<float> *>(lds + kLdsFixed);
    float *s_tails = lds + kLdsFixed + NPL * 2048;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    float *pl = lds + kLdsTw + w * kPlane;
    const int M = a.n_mics;
    const int n_pairs = (M + 1) >> 1;
    {
        const float *twf = reinterpret_cast<const float *>(a.twiddle);
        for (int i = tid; i < kLdsTw; i += kBlock) lds[i] = twf[i];
        for (int i = tid; i < kNfft; i += kBlock) s_win[(i & 63) * kWinStride + (i >> 6)] = a.window[i];  // [lane][j]
        if (NPL > 0) {
            const float *gf = reinterpret_cast<const float *>(a.gains);
            for (int i = tid; i < n_pairs * 2048; i += kBlock) lds[kLdsFixed + i] = gf[i];
        }
    }
    const int stream = blockIdx.x / a.chunks_per_stream;
    const long c_in_s = blockIdx.x - (long)stream * a.chunks_per_stream;
    const long T0 = c_in_s * a.frames_per_chunk;
    long T1 = T0 + a.frames_per_chunk;
    if (T1 > a.n_frames) T1 = a.n_frames;
    if (T0 == 0)
        for (int i = tid; i < kHop; i += kBlock) s_tails[i] = a.tail_in[(long)stream * kHop + i];
    __syncthreads();
    const float4 *wrow = reinterpret_cast<const float4 *>(s_win + lane * kWinStride);

    const float *xs = a.x + (long)stream * a.stream_stride_x;
    const float *hs = a.hist_in + (long)stream * M * kHop;
    float *ys = a.y + (long)stream * a.n_frames * kHop;

    float re[16], im[16], Sr[16], Si[16];
    const int n_iter = (int)((T1 - T0 + kWaves - 1) / kWaves);

    for (int it = 0; it < n_iter; ++it) {
        const long t = T0 + (long)it * kWaves + w;
        const bool valid = t < T1;
        const long tc = valid ? t : T1 - 1;

        for (int p = 0; p < n_pairs; ++p) {
            const int ma = 2 * p;
            const bool b_ok = (2 * p + 1) < M;
            const int mb = b_ok ? 2 * p + 1 : ma;
            const float bscale = b_ok ? 1.f : 0.f;
            if (LAYOUT == 0) {
                const float *a1 = (tc >= 1 ? xs + (long)ma * a.mic_stride + (tc - 1) * kHop : hs + ma * kHop) + lane;
                const float *b1 = (tc >= 1 ? xs + (long)mb * a.mic_stride + (tc - 1) * kHop : hs + mb * kHop) + lane;
                const float *a2 = xs + (long)ma * a.mic_stride + tc * kHop + lane;
                const float *b2 = xs + (long)mb * a.mic_stride + tc * kHop + lane;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    re[j] = a1[64 * j];
                    im[j] = b1[64 * j];
                    re[j + 8] = a2[64 * j];
                    im[j + 8] = b2[64 * j];
                }
            } else {
                const float *s1 = (tc >= 1 ? xs + (tc - 1) * (long)kHop * M : hs) + (long)lane * M;
                const float *s2 = xs + tc * (long)kHop * M + (long)lane * M;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    re[j] = s1[(long)64 * j * M + ma];
                    im[j] = s1[(long)64 * j * M + mb];
                    re[j + 8] = s2[(long)64 * j * M + ma];
                    im[j + 8] = s2[(long)64 * j * M + mb];
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 hv = wrow[g];
                re[4 * g + 0] *= hv.x; im[4 * g + 0] *= hv.x * bscale;
                re[4 * g + 1] *= hv.y; im[4 * g + 1] *= hv.y * bscale;
                re[4 * g + 2] *= hv.z; im[4 * g + 2] *= hv.z * bscale;
                re[4 * g + 3] *= hv.w; im[4 * g + 3] *= hv.w * bscale;
            }
            w64_fwd_p1<float>(re, im, lane, s_tw1);
            w64_T1_fwd(re, im, lane, pl);
            w64_fwd_p2<float>(re, im, lane, s_tw2);
            w64_T2<true>(re, im, lane);
            w64_fwd_p3<float>(re, im);

            const cx<float> *gp = (NPL > 0 ? s_gain : reinterpret_cast<const cx<float> *>(a.gains)) + (long)p * 1024 + lane;
            if (p == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const cx<float> g = gp[64 * r];
                    Sr[r] = g.x * re[r] - g.y * im[r];
                    Si[r] = g.x * im[r] + g.y * re[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const cx<float> g = gp[64 * r];
                    Sr[r] += g.x * re[r] - g.y * im[r];
                    Si[r] += g.x * im[r] + g.y * re[r];
                }
            }
        }

        if (a.sdump != nullptr && valid) {
            f32x2 *sd = a.sdump + ((long)stream * a.n_frames + t) * kNfft;
#pragma unroll
            for (int r = 0; r < 16; ++r) sd[w64_bin(lane, r)] = f32x2{Sr[r], Si[r]};
        }

        w64_inv_p3<float>(Sr, Si);
        w64_T2<false>(Sr, Si, lane);
        w64_inv_p2<float>(Sr, Si, lane, s_tw2);
        w64_T1_inv(Sr, Si, lane, pl);
        w64_inv_p1<float>(Sr, Si, lane, s_tw1);

        // register j holds sample n = 64*j + lane: j < 8 first half, j >= 8 second half
        float h[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 hv = wrow[g];
            h[4 * g + 0] = hv.x; h[4 * g + 1] = hv.y; h[4 * g + 2] = hv.z; h[4 * g + 3] = hv.w;
        }
        float *my_slot = s_tails + (w + 1) * kHop + lane;
#pragma unroll
        for (int j = 0; j < 8; ++j) my_slot[64 * j] = Sr[j + 8] * h[j + 8];
        __syncthreads();
        if (valid) {
            float *yo = ys + t * kHop + lane;
            const float *prev = s_tails + w * kHop + lane;
            if (t == T0 && T0 > 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) atomicAdd(yo + 64 * j, Sr[j] * h[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
#pragma clang fp contract(off)
                    yo[64 * j] = prev[64 * j] + Sr[j] * h[j];
                }
            }
            if (t == T1 - 1) {
                if (T1 < a.n_frames) {
                    float *yn = ys + T1 * kHop + lane;
#pragma unroll
                    for (int j = 0; j < 8; ++j) atomicAdd(yn + 64 * j, Sr[j + 8] * h[j + 8]);
                } else {
                    float *to = a.tail_out + (long)stream * kHop + lane;
#pragma unroll
                    for (int j = 0; j < 8; ++j) to[64 * j] = Sr[j + 8] * h[j + 8];
                    float *ho = a.hist_out + (long)stream * M * kHop;
                    if (LAYOUT == 0) {
                        for (int m = 0; m < M; ++m)
                            for (int j = 0; j < 8; ++j)
                                ho[m * kHop + 64 * j + lane] = xs[(long)m * a.mic_stride + t * kHop + 64 * j + lane];
                    } else {
                        for (int j = 0; j < 8 * M; ++j) ho[64 * j + lane] = xs[t * (long)kHop * M + 64 * j + lane];
                    }
                }
            }
        }
        __syncthreads();
        if (w == kWaves - 1) {
            const float *src = s_tails + kWaves * kHop + lane;
            float *dst = s_tails + lane;
#pragma unroll
            for (int j = 0; j < 8; ++j) dst[64 * j] = src[64 * j];
        }
    }
}

template <int LAYOUT>
void launch_layout(const DasFusedArgs &a, unsigned blocks, hipStream_t stream) {
    const int np = (a.n_mics + 1) / 2;
    if (np <= 4)
        hipLaunchKernelGGL((das_fused_w64_kernel<LAYOUT, 4>), dim3(blocks), dim3(kBlock), 0, stream, a);
    else
        hipLaunchKernelGGL((das_fused_w64_kernel<LAYOUT, 0>), dim3(blocks), dim3(kBlock), 0, stream, a);
}

}  // namespace

hipError_t launch_das_fused_w64(const DasFusedArgs &a, hipStream_t stream) {
    const unsigned blocks = (unsigned)((long)a.chunks_per_stream * a.n_streams);
    if (a.layout == 0)
        launch_layout<0>(a, blocks, stream);
    else
        launch_layout<1>(a, blocks, stream);
    return hipGetLastError();
}

}  // namespace bf
