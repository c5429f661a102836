// das_fused.hip -- fused fp32 delay-and-sum for gfx950 (MI355X).
//
// One kernel does, per STFT frame, everything the reference's das node does in
// apply_weights() + do_overlap():
//   overlap_and_add_prepare_input  util.h:217-242   window + load (two mics packed as re/im)
//   fftw_execute(x_forward) x M    das.cpp:51-57    -> ceil(M/2) complex FFT-1024
//   weights^H * in_fft / M         das.cpp:60-63    -> S += D_p * Z_p   (geometry.hpp das_pair_gains)
//   fftw_execute(y_inverse)        das.cpp:66       -> one complex IFFT-1024, real part kept
//   overlap_and_add_prepare_output util.h:244-253   1/N (folded into D) and synthesis window
//   out = prev[H+n] + cur[n]       util.h:301-302   overlap-add, tail kept in registers
//
// Mapping: a 32-lane half-wavefront owns a run of consecutive frames of one
// stream; each lane holds 32 complex points in VGPRs (fft1024.hpp).  The two
// halves of a wavefront work on different frame runs, so no cross-lane traffic
// exists outside the per-FFT LDS transpose, and there is no workgroup barrier in
// the main loop.  The first frame of every run is recomputed (not stored) to
// obtain the overlap tail; run 0 of a stream takes it from the carried state.
//
// Bound: HBM stream of M*H*4 B in + H*4 B out per frame at >= 40 % of 8 TB/s
// needs ~50 % of the fp32 VALU peak for the 5 FFTs per frame; LDS carries one
// 8.5 KiB transpose per FFT plus the 8 KiB twiddle table reads.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "fft1024.hpp"
#include "kernels.hpp"

namespace bf {

namespace {

constexpr int kBlock = 512;
constexpr int kHalves = kBlock / 32;
constexpr int kRS = tr_stride<float>::value;
constexpr int kHop = 512;
constexpr int kNfft = 1024;
constexpr int kWinStride = 36;  // floats per lane row: 144 B keeps ds_read_b128 groups conflict-free

// ABL: timing-only ablation bits (outputs are wrong when != 0): 1 = no gain loads, 2 = no input loads, 4 = no LDS transpose
template <int LAYOUT, int ABL = 0>
__global__ __launch_bounds__(kBlock, 2) void das_fused_kernel(DasFusedArgs a) {
    // one LDS object: [twiddles 1024][16 transpose buffers][window 32 lanes x 36]
    __shared__ __attribute__((aligned(16))) cx<float> lds[1024 + kHalves * 32 * kRS + (32 * kWinStride) / 2];
    cx<float> *s_tw = lds;
    float *s_win = reinterpret_cast<float *>(lds + 1024 + kHalves * 32 * kRS);
    const int tid = threadIdx.x;
    const int lane = tid & 31;
    const int hw = tid >> 5;
    cx<float> *buf = lds + 1024 + hw * (32 * kRS);

    for (int i = tid; i < 1024; i += kBlock) {
        const f32x2 w = a.twiddle[i];
        s_tw[i] = cx<float>{w.x, w.y};
    }
    for (int i = tid; i < 1024; i += kBlock) s_win[(i & 31) * kWinStride + (i >> 5)] = a.window[i];  // [lane][j]
    __syncthreads();
    const float4 *wrow = reinterpret_cast<const float4 *>(s_win + lane * kWinStride);

    const long chunk = (long)blockIdx.x * kHalves + hw;
    int stream = (int)(chunk / a.chunks_per_stream);
    const long c_in_s = chunk - (long)stream * a.chunks_per_stream;
    const bool chunk_ok = stream < a.n_streams;
    if (!chunk_ok) stream = a.n_streams - 1;  // keep addresses valid; stores are predicated
    const long t0 = c_in_s * a.frames_per_chunk;
    const int M = a.n_mics;
    const int n_pairs = (M + 1) >> 1;

    const float *xs = a.x + (long)stream * a.stream_stride_x;
    const float *hs = a.hist + (long)stream * M * kHop;
    float *ys = a.y + (long)stream * a.n_frames * kHop;

    float tail[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) tail[q] = 0.f;

    float re[32], im[32], Sr[32], Si[32];

    for (int it = 0; it <= a.frames_per_chunk; ++it) {
        const long t = t0 - 1 + it;  // it == 0: warm-up frame (overlap tail only)
        const bool store = chunk_ok && it > 0 && t < a.n_frames;
        long tc = t < 0 ? 0 : t;
        if (tc > a.n_frames - 1) tc = a.n_frames - 1;

        for (int p = 0; p < n_pairs; ++p) {
            const int ma = 2 * p;
            const bool b_ok = (2 * p + 1) < M;
            const int mb = b_ok ? 2 * p + 1 : ma;
            const float bscale = b_ok ? 1.f : 0.f;
            if (ABL & 2) {
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    re[j] = (float)(j + lane) * 1e-3f + (float)tc;
                    im[j] = (float)(j - lane) * 1e-3f;
                }
            } else if (LAYOUT == 0) {
                const float *a1 = (tc >= 1 ? xs + (long)ma * a.mic_stride + (tc - 1) * kHop : hs + ma * kHop) + lane;
                const float *b1 = (tc >= 1 ? xs + (long)mb * a.mic_stride + (tc - 1) * kHop : hs + mb * kHop) + lane;
                const float *a2 = xs + (long)ma * a.mic_stride + tc * kHop + lane;
                const float *b2 = xs + (long)mb * a.mic_stride + tc * kHop + lane;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    re[j] = a1[32 * j];
                    im[j] = b1[32 * j];
                    re[j + 16] = a2[32 * j];
                    im[j + 16] = b2[32 * j];
                }
            } else {
                const float *s1 = (tc >= 1 ? xs + (tc - 1) * (long)kHop * M : hs) + (long)lane * M;
                const float *s2 = xs + tc * (long)kHop * M + (long)lane * M;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    re[j] = s1[(long)32 * j * M + ma];
                    im[j] = s1[(long)32 * j * M + mb];
                    re[j + 16] = s2[(long)32 * j * M + ma];
                    im[j + 16] = s2[(long)32 * j * M + mb];
                }
            }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float4 hv = wrow[g];
                re[4 * g + 0] *= hv.x; im[4 * g + 0] *= hv.x * bscale;
                re[4 * g + 1] *= hv.y; im[4 * g + 1] *= hv.y * bscale;
                re[4 * g + 2] *= hv.z; im[4 * g + 2] *= hv.z * bscale;
                re[4 * g + 3] *= hv.w; im[4 * g + 3] *= hv.w * bscale;
            }

            if (ABL & 4) {
                fft32_dif<float, -1>(re, im);
                fft32_dif<float, -1>(re, im);
            } else {
                fft1024_fwd_a<float>(re, im, lane, s_tw, buf);
                __builtin_amdgcn_wave_barrier();
                fft1024_fwd_b<float>(re, im, lane, buf);
                __builtin_amdgcn_wave_barrier();
            }

            const f32x2 *gp = a.gains + (long)p * 1024 + lane;
            if (p == 0) {
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const f32x2 g = (ABL & 1) ? f32x2{0.5f + i, 0.25f} : gp[32 * i];
                    Sr[i] = g.x * re[i] - g.y * im[i];
                    Si[i] = g.x * im[i] + g.y * re[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const f32x2 g = (ABL & 1) ? f32x2{0.5f + i, 0.25f} : gp[32 * i];
                    Sr[i] += g.x * re[i] - g.y * im[i];
                    Si[i] += g.x * im[i] + g.y * re[i];
                }
            }
        }

        if (a.sdump != nullptr && store) {
            f32x2 *sd = a.sdump + ((long)stream * a.n_frames + t) * kNfft + lane;
#pragma unroll
            for (int i = 0; i < 32; ++i) sd[32 * brev5(i)] = f32x2{Sr[i], Si[i]};
        }

        if (ABL & 4) {
            fft32_dit<float, +1>(Sr, Si);
            fft32_dif<float, +1>(Sr, Si);
        } else {
            fft1024_inv_a<float>(Sr, Si, lane, s_tw, buf);
            __builtin_amdgcn_wave_barrier();
            fft1024_inv_b<float>(Sr, Si, lane, buf);
            __builtin_amdgcn_wave_barrier();
        }

        // position i holds sample n = 32*brev5(i) + lane; even i -> first half, odd i -> n + 512
        float h[32];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float4 hv = wrow[g];
            h[4 * g + 0] = hv.x; h[4 * g + 1] = hv.y; h[4 * g + 2] = hv.z; h[4 * g + 3] = hv.w;
        }
        if (store) {
            float *yo = ys + t * kHop + lane;
#pragma unroll
            for (int q = 0; q < 16; ++q) yo[32 * brev5(2 * q)] = tail[q] + Sr[2 * q] * h[brev5(2 * q)];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) tail[q] = Sr[2 * q + 1] * h[brev5(2 * q + 1)];

        if (it == 0 && t0 == 0) {  // stream start: overlap tail comes from the carried state
            const float *ti = a.tail_in + (long)stream * kHop + lane;
#pragma unroll
            for (int q = 0; q < 16; ++q) tail[q] = ti[32 * brev5(2 * q)];
        }
        if (store && t == a.n_frames - 1) {
            float *to = a.tail_out + (long)stream * kHop + lane;
#pragma unroll
            for (int q = 0; q < 16; ++q) to[32 * brev5(2 * q)] = tail[q];
        }
    }
}

__global__ void das_hermitian_dump_kernel(const f32x2 *s, f64x2 *out, long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long f = idx / kNfft;
    const int k = (int)(idx - f * kNfft);
    const f32x2 u = s[f * kNfft + k];
    const f32x2 v = s[f * kNfft + ((kNfft - k) & (kNfft - 1))];
    // undo the folded 1/N; Hermitian part (S[k] + conj(S[N-k]))/2
    out[idx] = f64x2{0.5 * kNfft * ((double)u.x + (double)v.x), 0.5 * kNfft * ((double)u.y - (double)v.y)};
}

}  // namespace

hipError_t launch_das_fused(const DasFusedArgs &a, hipStream_t stream) {
    const long chunks = (long)a.chunks_per_stream * a.n_streams;
    const unsigned blocks = (unsigned)((chunks + kHalves - 1) / kHalves);
    static const int abl = getenv("BF_ABLATE") ? atoi(getenv("BF_ABLATE")) : 0;
    if (abl && a.layout == 0) {
        switch (abl) {
            case 1: hipLaunchKernelGGL((das_fused_kernel<0, 1>), dim3(blocks), dim3(kBlock), 0, stream, a); break;
            case 2: hipLaunchKernelGGL((das_fused_kernel<0, 2>), dim3(blocks), dim3(kBlock), 0, stream, a); break;
            case 3: hipLaunchKernelGGL((das_fused_kernel<0, 3>), dim3(blocks), dim3(kBlock), 0, stream, a); break;
            case 4: hipLaunchKernelGGL((das_fused_kernel<0, 4>), dim3(blocks), dim3(kBlock), 0, stream, a); break;
            case 7: hipLaunchKernelGGL((das_fused_kernel<0, 7>), dim3(blocks), dim3(kBlock), 0, stream, a); break;
            default: break;
        }
        return hipGetLastError();
    }
    if (a.layout == 0)
        hipLaunchKernelGGL(das_fused_kernel<0>, dim3(blocks), dim3(kBlock), 0, stream, a);
    else
        hipLaunchKernelGGL(das_fused_kernel<1>, dim3(blocks), dim3(kBlock), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_das_hermitian_dump(const f32x2 *sdump, f64x2 *out, long n_frames_total, hipStream_t stream) {
    const long total = n_frames_total * kNfft;
    hipLaunchKernelGGL(das_hermitian_dump_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, sdump, out,
                       total);
    return hipGetLastError();
}

}  // namespace bf
