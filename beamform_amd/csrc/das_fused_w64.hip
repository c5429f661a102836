// das_fused_w64.hip -- fused fp32 delay-and-sum, 64-lane x 16-point-per-lane FFT (fft1024_w64.hpp).
//
// Same algorithm, boundary and state handling as das_fused.hip (das.cpp:47-70 + util.h:217-314 in one kernel); the
// difference is the FFT factorisation: one full wavefront per frame, 16 complex points per lane, three passes
// (16 x 16 x 4).  64 data + accumulator registers instead of 128, so a CU holds 16 wavefronts = FOUR per SIMD (one
// 1024-thread block walking 16 consecutive frames per iteration).  Why that matters: a wavefront issues at most one VALU
// instruction per ~4.8 cycles and cannot hide its own load / LDS waits (das_fused_kernel: 49 % of a wavefront's cycles
// issue VALU work, 30 % sit in s_waitcnt -- profiles/r02_*_das8_pmc.txt), so with two wavefronts per SIMD the chip is bound
// by the critical path of ONE wavefront; with four the SIMD has something to issue nearly all the time.
//
// What makes the three-pass form affordable (the round-1 version of this file lost 40 % to the 32 x 32 kernel):
//   * the first pass keeps lane = sample (global loads / stores lane-contiguous: a permuted lane order was measured at 8x the
//     TA/TCP accesses and 45 % of the wave cycles in issue stalls);
//   * T1, the one LDS transpose: lane 4a+b writes register k1 to row k1, column 16 b + a (two-way store conflicts, free on
//     ds_write_b32); lane 16 b + k1 reads columns 16 b .. 16 b + 15 of row k1 as four ds_read_b128 (two-way conflicts at a
//     272-B row: 8 instead of 4 LDS cycles) instead of sixteen strided ds_read_b32;
//   * T2, the 4 x 4 transpose across the four 16-lane rows: v_permlane32_swap + v_permlane16_swap, four instructions per
//     four registers and no select (the quad_perm version needed 8 DPP moves + 8 v_cndmask);
//   * no block barrier anywhere in the loop: the overlap-add partner travels through the 17-slot LDS ring with one flag per
//     slot, as in das_fused.hip.
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
constexpr int kRS = 68;                      // T1 plane row stride (floats)
constexpr int kPlane = 16 * kRS;             // floats per wavefront
constexpr int kWinStride = 20;               // floats per lane row of the window table (16 values + pad)
constexpr int kLdsTw = 2 * (1024 + 64);      // floats
constexpr int kLdsFixed = kLdsTw + kWaves * kPlane + 64 * kWinStride;

// ---- T2: 4 x 4 transpose across the 16-lane rows (lane field b = lane >> 4) -----------------------------------------
// afterwards register c of row q holds what register q of row c held.  Lane semantics of the two swaps checked on the
// device by tools/ubench/permswap.hip.
__device__ __forceinline__ void row_transpose4(float &r0, float &r1, float &r2, float &r3) {
    // Inline asm on purpose: with the two-result __builtin_amdgcn_permlane{32,16}_swap hipcc (ROCm 7.2) was seen to treat the
    // second result of a chained swap as a copy of the first (tools/ubench/w64_test.hip caught it: "v_mov v3, v2" in place of
    // the swapped partner).  Operands are read-write; the s_nop 1 pads are the two wait states a VALU write needs before a
    // v_permlane*_swap reads it (guide T21), before the first pair and between the dependent pairs.
    //   permlane32_swap a, b: rows {2,3} of a <-> rows {0,1} of b;   permlane16_swap a, b: odd rows of a <-> even rows of b
    asm volatile(
        "s_nop 1\n\t"
        "v_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\t"
        "s_nop 1\n\t"
        "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3"
        : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
}
constexpr int brev2c(int i) { return ((i & 1) << 1) | ((i >> 1) & 1); }

// T2: position brev2(g) + 4*brev2(q) (row b)  <->  register 4*g + b (row q)
template <bool FWD>
__device__ __forceinline__ void w64_T2(float (&re)[16], float (&im)[16]) {
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
        row_transpose4(r[0], r[1], r[2], r[3]);
        row_transpose4(s[0], s[1], s[2], s[3]);
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

// ---- T1 through one scalar LDS plane (real plane, then imaginary plane) ------------------------------------------------
// forward: position i (k1 = brev4(i)) of lane 4a+b  ->  register a of lane 16 b + k1.
//   wcol = plane + w64_col(lane): this lane's column (rows brev4(i));  row16 = this lane's 16 columns of row (lane & 15).
// LDS operations of one wavefront execute in issue order: only compiler barriers separate the phases.
__device__ __forceinline__ void w64_T1_fwd(float (&re)[16], float (&im)[16], float *wcol, const float *row16) {
    const float4 *r4 = reinterpret_cast<const float4 *>(row16);
#pragma unroll
    for (int i = 0; i < 16; ++i) wcol[brev4(i) * kRS] = re[i];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 q = r4[g];
        re[4 * g + 0] = q.x; re[4 * g + 1] = q.y; re[4 * g + 2] = q.z; re[4 * g + 3] = q.w;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 16; ++i) wcol[brev4(i) * kRS] = im[i];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 q = r4[g];
        im[4 * g + 0] = q.x; im[4 * g + 1] = q.y; im[4 * g + 2] = q.z; im[4 * g + 3] = q.w;
    }
    __builtin_amdgcn_wave_barrier();
}
// backward: register a of lane 16 b + k1  ->  position i (k1 = brev4(i)) of lane 4a+b.
__device__ __forceinline__ void w64_T1_inv(float (&re)[16], float (&im)[16], float *row16, const float *wcol) {
    float4 *w4 = reinterpret_cast<float4 *>(row16);
#pragma unroll
    for (int g = 0; g < 4; ++g) w4[g] = float4{re[4 * g], re[4 * g + 1], re[4 * g + 2], re[4 * g + 3]};
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 16; ++i) re[i] = wcol[brev4(i) * kRS];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int g = 0; g < 4; ++g) w4[g] = float4{im[4 * g], im[4 * g + 1], im[4 * g + 2], im[4 * g + 3]};
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 16; ++i) im[i] = wcol[brev4(i) * kRS];
    __builtin_amdgcn_wave_barrier();
}

template <int LAYOUT, int NPL>
__global__ __launch_bounds__(kBlock) void das_fused_w64_kernel(DasFusedArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[kLdsFixed + NPL * 2048 + 17 * kHop + 32];
    const cx<float> *s_tw1 = reinterpret_cast<const cx<float> *>(lds);
    const cx<float> *s_tw2 = s_tw1 + 1024;
    float *s_win = lds + kLdsTw + kWaves * kPlane;
    const cx<float> *s_gain = reinterpret_cast<const cx<float> *>(lds + kLdsFixed);
    // 17-slot ring of frame tails with one flag per slot: see das_fused.hip
    float *s_tails = lds + kLdsFixed + NPL * 2048;
    // LDS address space spelled out: a volatile access through a generic pointer compiles to flat_load / flat_store, whose wait is
    // vmcnt(0) -- it would drain the next frame's prefetch in front of every flag read
    volatile __attribute__((address_space(3))) int *s_flag = (volatile __attribute__((address_space(3))) int *)(lds + kLdsFixed + NPL * 2048 + 17 * kHop);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int m = lane;                      // first pass: this lane's sample inside every block of 64
    float *pl = lds + kLdsTw + w * kPlane;
    float *row16 = pl + (lane & 15) * kRS + 16 * (lane >> 4);   // T1: second-pass lane (b, k1): its 16 columns of row k1
    float *wcol = pl + w64_col(lane);                           // T1: first-pass lane 4a+b: column 16 b + a of every row
    const int M = a.n_mics;
    const int n_pairs = (M + 1) >> 1;

    const int stream = blockIdx.x / a.chunks_per_stream;  // output stream = input stream * n_dirs + look direction
    const long c_in_s = blockIdx.x - (long)stream * a.chunks_per_stream;
    const int in_stream = stream / a.n_dirs;
    const f32x2 *gains = a.gains + (long)(stream - in_stream * a.n_dirs) * n_pairs * 1024;
    {
        const float *twf = reinterpret_cast<const float *>(a.twiddle);
        for (int i = tid; i < kLdsTw; i += kBlock) lds[i] = twf[i];
        // [lane][j] = window[64 j + m(lane)]
        for (int i = tid; i < kNfft; i += kBlock) s_win[(i & 63) * kWinStride + (i >> 6)] = a.window[i];
        if (NPL > 0) {
            const float *gf = reinterpret_cast<const float *>(gains);
            for (int i = tid; i < n_pairs * 2048; i += kBlock) lds[kLdsFixed + i] = gf[i];
        }
    }
    const long T0 = c_in_s * a.frames_per_chunk;
    long T1 = T0 + a.frames_per_chunk;
    if (T1 > a.n_frames) T1 = a.n_frames;
    if (T0 == 0)  // stream start: the overlap partner of frame 0 is the carried state, stored in sample order
        for (int i = tid; i < kHop; i += kBlock) s_tails[i] = a.tail_in[(long)stream * kHop + i];
    if (tid < 17) s_flag[tid] = (tid == 0) ? (int)(T0 - 1) : -2;
    __syncthreads();
    const float4 *wrow = reinterpret_cast<const float4 *>(s_win + lane * kWinStride);

    const float *xs = a.x + (long)in_stream * a.stream_stride_x;
    const float *hs = a.hist_in + (long)in_stream * M * kHop;
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
            if (LAYOUT == 0) {
                const float *a1 = (tc >= 1 ? xs + (long)ma * a.mic_stride + (tc - 1) * kHop : hs + ma * kHop) + m;
                const float *b1 = (!b_ok ? a.zeros : tc >= 1 ? xs + (long)(ma + 1) * a.mic_stride + (tc - 1) * kHop : hs + (ma + 1) * kHop) + m;
                const float *a2 = xs + (long)ma * a.mic_stride + tc * kHop + m;
                const float *b2 = (!b_ok ? a.zeros : xs + (long)(ma + 1) * a.mic_stride + tc * kHop) + m;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    re[j] = a1[64 * j];
                    im[j] = b1[64 * j];
                    re[j + 8] = a2[64 * j];
                    im[j + 8] = b2[64 * j];
                }
            } else {
                const int mb = b_ok ? ma + 1 : ma;
                const float bscale = b_ok ? 1.f : 0.f;
                const float *s1 = (tc >= 1 ? xs + (tc - 1) * (long)kHop * M : hs) + (long)m * M;
                const float *s2 = xs + tc * (long)kHop * M + (long)m * M;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    re[j] = s1[(long)64 * j * M + ma];
                    im[j] = s1[(long)64 * j * M + mb] * bscale;
                    re[j + 8] = s2[(long)64 * j * M + ma];
                    im[j + 8] = s2[(long)64 * j * M + mb] * bscale;
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 hv = wrow[g];
                re[4 * g + 0] *= hv.x; im[4 * g + 0] *= hv.x;
                re[4 * g + 1] *= hv.y; im[4 * g + 1] *= hv.y;
                re[4 * g + 2] *= hv.z; im[4 * g + 2] *= hv.z;
                re[4 * g + 3] *= hv.w; im[4 * g + 3] *= hv.w;
            }
            w64_fwd_p1<float>(re, im, lane, s_tw1);
            w64_T1_fwd(re, im, wcol, row16);
            w64_fwd_p2<float>(re, im, lane, s_tw2);
            w64_T2<true>(re, im);
            w64_fwd_p3<float>(re, im);

            const cx<float> *gp = (NPL > 0 ? s_gain : reinterpret_cast<const cx<float> *>(gains)) + (long)p * 1024 + lane;
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
        w64_T2<false>(Sr, Si);
        w64_inv_p2<float>(Sr, Si, lane, s_tw2);
        w64_T1_inv(Sr, Si, row16, wcol);
        w64_inv_p1<float>(Sr, Si, lane, s_tw1);

        // register j holds sample n = 64*j + m: j < 8 first half, j >= 8 second half
        float h[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 hv = wrow[g];
            h[4 * g + 0] = hv.x; h[4 * g + 1] = hv.y; h[4 * g + 2] = hv.z; h[4 * g + 3] = hv.w;
        }
        // second half of this frame -> its ring slot (lane order), then publish: LDS operations of a wavefront complete in order
        const int r = (int)(tc - T0);
        const int my = (r + 1) % 17, pv = r % 17;
        if (valid) {
            float *my_slot = s_tails + my * kHop + lane;
#pragma unroll
            for (int j = 0; j < 8; ++j) my_slot[64 * j] = Sr[j + 8] * h[j + 8];
            asm volatile("" ::: "memory");
            if (lane == 0) s_flag[my] = (int)t;
        }
        if (valid) {
            float *yo = ys + t * kHop + m;
            if (t == T0 && T0 > 0) {
                // first hop of the run: the previous run adds its half separately (both into a zeroed hop)
#pragma unroll
                for (int j = 0; j < 8; ++j) atomicAdd(yo + 64 * j, Sr[j] * h[j]);
            } else {
                while (s_flag[pv] != (int)(t - 1)) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
                const float *prev = s_tails + pv * kHop + lane;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    // product and sum rounded separately (util.h:250-252, 302), as on the atomic run-boundary path
#pragma clang fp contract(off)
                    yo[64 * j] = prev[64 * j] + Sr[j] * h[j];
                }
            }
            if (t == T1 - 1) {
                if (T1 < a.n_frames) {  // last frame of the run: its second half belongs to the next run's first hop
                    float *yn = ys + T1 * kHop + m;
#pragma unroll
                    for (int j = 0; j < 8; ++j) atomicAdd(yn + 64 * j, Sr[j + 8] * h[j + 8]);
                } else {
                    // end of the batch: carried state for the next call (OLA tail and the last input hop)
                    float *to = a.tail_out + (long)stream * kHop + m;
#pragma unroll
                    for (int j = 0; j < 8; ++j) to[64 * j] = Sr[j + 8] * h[j + 8];
                    float *ho = a.hist_out + (long)in_stream * M * kHop;  // every direction writes the same values
                    if (LAYOUT == 0) {
                        for (int mm = 0; mm < M; ++mm)
                            for (int j = 0; j < 8; ++j)
                                ho[mm * kHop + 64 * j + lane] = xs[(long)mm * a.mic_stride + t * kHop + 64 * j + lane];
                    } else {
                        for (int j = 0; j < 8 * M; ++j) ho[64 * j + lane] = xs[t * (long)kHop * M + 64 * j + lane];
                    }
                }
            }
        }
    }
}

template <int LAYOUT>
void launch_layout(const DasFusedArgs &a, unsigned blocks, hipStream_t stream) {
    const int np = (a.n_mics + 1) / 2;
    if (np <= 4)
        hipLaunchKernelGGL((das_fused_w64_kernel<LAYOUT, 4>), dim3(blocks), dim3(kBlock), 0, stream, a);
    else  // > 8 mics: gains from L2
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
