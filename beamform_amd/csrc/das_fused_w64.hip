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

#include "launch_trace.hpp"
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

// =====================================================================================================================================
//        JACK period 1024 (FFT 2048) on the 64-lane transform: das_fused_2048.hip's radix-2 split, one full wavefront per frame
// =====================================================================================================================================
// e[n] = w[n] x[n] + w[n + 1024] x[n + 1024] gives the even bins, (w[n] x[n] - w[n + 1024] x[n + 1024]) W2048^n the odd ones; per pass
// ceil(M/2) packed forward FFT-1024, S += D_p Z_p with the pair gains of the even (odd) bins, one backward transform; y[n] = Re A[n] +
// Re(conj(W^n) B[n]), y[n + 1024] = Re A[n] - ... (das_fused_2048.hip).  16 points per lane instead of 32: ~130 registers, THREE
// wavefronts per SIMD (the half-wavefront version needs 350 and runs one).  A wavefront owns (output stream, run of consecutive frames)
// and walks it; the overlap-add tail waits in the output buffer (same lane, same addresses, program order); a run that does not
// start the stream recomputes its previous frame.  LDS: 8.5 KB twiddles + 12 x 4.3 KB exchange planes + 8 KB W^n + 8 KB window +
// 64 KB pair gains in the order the passes read them (one look direction, <= 8 microphones; otherwise from L2 in natural order).
constexpr int kBlk2 = 768, kWaves2 = kBlk2 / 64;
constexpr int kH2 = 1024;  // hop = JACK period
constexpr int o2Pl = kLdsTw, o2W = o2Pl + kWaves2 * kPlane, o2Win = o2W + 2048, o2G = o2Win + 2048, kLds2 = o2G + 2 * 4 * 16 * 64 * 2;

template <int LAYOUT>
__global__ __launch_bounds__(kBlk2) void das_fused_2048_w64_kernel(DasFusedArgs a, const f32x2 *tw_split) {
    __shared__ __attribute__((aligned(16))) float lds[kLds2];
    const cx<float> *s_tw1 = reinterpret_cast<const cx<float> *>(lds);
    const cx<float> *s_tw2 = s_tw1 + 1024;
    const f32x2 *s_w = reinterpret_cast<const f32x2 *>(lds + o2W);
    const float *s_win = lds + o2Win;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *pl = lds + o2Pl + w * kPlane;
    float *row16 = pl + (lane & 15) * kRS + 16 * (lane >> 4);
    float *wcol = pl + w64_col(lane);
    const int M = a.n_mics, NP = (M + 1) >> 1;
    const bool g_lds = NP <= 4 && a.n_dirs == 1;
    {
        const float *tf = reinterpret_cast<const float *>(tw_split);  // [w64 twiddles (1088 complex) | W2048^n, n < 1024]
        for (int i = tid; i < kLdsTw; i += kBlk2) lds[i] = tf[i];
        for (int i = tid; i < 2048; i += kBlk2) lds[o2W + i] = tf[kLdsTw + i];
        for (int i = tid; i < 2048; i += kBlk2) lds[o2Win + i] = a.window[i];
        if (g_lds) {
            f32x2 *lg = reinterpret_cast<f32x2 *>(lds + o2G);
            for (int e = tid; e < NP * 2048; e += kBlk2) {  // e = ((pass * NP + p) * 16 + r) * 64 + lane  <-  bin 2 w64_bin(lane, r) + pass of pair p
                const int l = e & 63, r = (e >> 6) & 15, pp = e >> 10, p = pp % NP, pass = pp / NP;
                lg[e] = a.gains[(long)p * 2048 + 2 * w64_bin(l, r) + pass];
            }
        }
    }
    __syncthreads();
    const long L = a.frames_per_chunk, runs = a.chunks_per_stream;
    const long item = (long)blockIdx.x * kWaves2 + w;
    if (item >= (long)a.n_streams * runs) return;  // no block barrier below
    const int s = (int)(item / runs);              // output stream = input stream * n_dirs + look direction
    const long t0 = (item - (long)s * runs) * L;
    long te = t0 + L;
    if (te > a.n_frames) te = a.n_frames;
    const int in_stream = s / a.n_dirs;
    const f32x2 *gains = a.gains + (long)(s - in_stream * a.n_dirs) * NP * 2048;  // [pair][bin], 1/N folded in
    const float *xs = a.x + (long)in_stream * a.stream_stride_x;
    const float *hs = a.hist_in + (long)in_stream * M * kH2;
    float *ys = a.y + (long)s * a.n_frames * kH2;

    // sample 64 j + lane of hop h (h = -1: the carried hop) of microphone m; jstep = elements between a lane's consecutive registers
    auto hop_ptr = [&](long h, int m) -> const float * {
        if (LAYOUT == 0) return (h >= 0 ? xs + (long)m * a.mic_stride + h * kH2 : hs + (long)m * kH2) + lane;
        return (h >= 0 ? xs + h * (long)kH2 * M : hs) + (long)lane * M + m;
    };
    const long jstep = LAYOUT == 0 ? 64 : (long)64 * M;

    const long tb = t0 == 0 ? 0 : t0 - 1;
    for (long t = tb; t < te; ++t) {  // t0 - 1: warm-up frame, only its second half (the tail) is used
        float v[16];  // Re(conj(W^n) B[n]) of the odd pass
        float Sr[16], Si[16];
        for (int pass = 1; pass >= 0; --pass) {  // odd bins first
            const float sg = pass ? -1.f : 1.f;  // the sign of the second half's term
            for (int p = 0; p < NP; ++p) {
                float re[16], im[16];
                const int ma = 2 * p, mb = 2 * p + 1;
                {   // channel a, then channel b: one channel's two hops in flight at a time
                    float x2[16];
                    const float *q1 = hop_ptr(t - 1, ma), *q2 = hop_ptr(t, ma);
#pragma unroll
                    for (int j = 0; j < 16; ++j) re[j] = q1[j * jstep];
#pragma unroll
                    for (int j = 0; j < 16; ++j) x2[j] = q2[j * jstep];
#pragma unroll
                    for (int j = 0; j < 16; ++j)  // buf[j]*hann_win[i] (util.h:235), both halves of the frame
                        re[j] = bf_fma(x2[j], s_win[kH2 + 64 * j + lane] * sg, re[j] * s_win[64 * j + lane]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (mb < M) {
                        const float *r1 = hop_ptr(t - 1, mb), *r2 = hop_ptr(t, mb);
#pragma unroll
                        for (int j = 0; j < 16; ++j) im[j] = r1[j * jstep];
#pragma unroll
                        for (int j = 0; j < 16; ++j) x2[j] = r2[j * jstep];
#pragma unroll
                        for (int j = 0; j < 16; ++j)
                            im[j] = bf_fma(x2[j], s_win[kH2 + 64 * j + lane] * sg, im[j] * s_win[64 * j + lane]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 16; ++j) im[j] = 0.f;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (pass) {  // o[n] *= W^n
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const f32x2 ww = s_w[64 * j + lane];
                        const float xr = re[j], xi = im[j];
                        re[j] = xr * ww.x - xi * ww.y;
                        im[j] = xr * ww.y + xi * ww.x;
                    }
                }
                w64_fwd_p1<float>(re, im, lane, s_tw1);
                w64_T1_fwd(re, im, wcol, row16);
                w64_fwd_p2<float>(re, im, lane, s_tw2);
                w64_T2<true>(re, im);
                w64_fwd_p3<float>(re, im);
                // register r of lane l holds bin k = w64_bin(l, r) of this pass' transform = bin 2 k + pass of the frame
                if (g_lds) {
                    const f32x2 *gp = reinterpret_cast<const f32x2 *>(lds + o2G) + (pass * NP + p) * 1024 + lane;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const f32x2 g = gp[64 * r];
                        Sr[r] = bf_fma(-g.y, im[r], bf_fma(g.x, re[r], p == 0 ? 0.f : Sr[r]));
                        Si[r] = bf_fma(g.y, re[r], bf_fma(g.x, im[r], p == 0 ? 0.f : Si[r]));
                    }
                } else {
                    const f32x2 *gp = gains + (long)p * 2048 + pass;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const f32x2 g = gp[2 * w64_bin(lane, r)];
                        Sr[r] = bf_fma(-g.y, im[r], bf_fma(g.x, re[r], p == 0 ? 0.f : Sr[r]));
                        Si[r] = bf_fma(g.y, re[r], bf_fma(g.x, im[r], p == 0 ? 0.f : Si[r]));
                    }
                }
            }
            w64_inv_p3<float>(Sr, Si);
            w64_T2<false>(Sr, Si);
            w64_inv_p2<float>(Sr, Si, lane, s_tw2);
            w64_T1_inv(Sr, Si, row16, wcol);
            w64_inv_p1<float>(Sr, Si, lane, s_tw1);
            if (pass) {  // register j <-> n = 64 j + lane: Re(conj(W^n) B[n])
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const f32x2 ww = s_w[64 * j + lane];
                    v[j] = Sr[j] * ww.x + Si[j] * ww.y;
                }
            }
        }
        // Sr = Re A.  First half: sample n, second half: sample n + 1024; synthesis window and overlap-add with the float stores
        float *yo = ys + t * kH2 + lane;
        const float *prev = (t == 0) ? a.tail_in + (long)s * kH2 + lane : yo;  // the tail parked by frame t - 1 (the carried state at t = 0)
        const bool store = t >= t0, park = t + 1 < te;
        float *yn = yo + kH2;
        float o2[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
#pragma clang fp contract(off)
            const float o1 = (Sr[j] + v[j]) * s_win[64 * j + lane];     // (float)(Re / N) [1/N inside the gains] times hann (util.h:249-251)
            o2[j] = (Sr[j] - v[j]) * s_win[kH2 + 64 * j + lane];
            if (store) yo[64 * j] = prev[64 * j] + o1;                  // out = prev[H + n] + cur[n]  (util.h:301-302)
            if (park) yn[64 * j] = o2[j];                               // completed by the run's next frame
        }
        if (t == a.n_frames - 1) {  // end of the batch: carried state for the next call (OLA tail and the last input hop)
            float *to = a.tail_out + (long)s * kH2 + lane;
#pragma unroll
            for (int j = 0; j < 16; ++j) to[64 * j] = o2[j];
            float *ho = a.hist_out + (long)in_stream * M * kH2;  // every look direction writes the same values
            if (LAYOUT == 0) {
                for (int m = 0; m < M; ++m)
                    for (int j = 0; j < 16; ++j) ho[m * kH2 + 64 * j + lane] = xs[(long)m * a.mic_stride + t * kH2 + 64 * j + lane];
            } else {
                for (int j = 0; j < 16 * M; ++j) ho[64 * j + lane] = xs[t * (long)kH2 * M + 64 * j + lane];
            }
        }
    }
}

// =====================================================================================================================================
//     JACK periods below 512 frames on the 64-lane transform: das_fused_small.hip's frame interleaving, one full wavefront per run
// =====================================================================================================================================
// R = 1024 / N consecutive frames interleaved into one 1024-point sequence, the N-point pair gains repeated R times (an LTI identity:
// das_fused_small.hip).  Lane l of the wavefront holds frame l mod R, samples m = (64 / R) j + l / R; first halves are registers j < 8,
// the overlap-add partner of (frame i, m) is one lane to the left, for i = 0 the previous group's last frame R - 1 lanes to the right in the
// previous iteration's values (both inside a 16-lane row: DPP).  16 points per lane: ~130 registers, three wavefronts per SIMD -- the
// half-wavefront version (32 points per lane, 229 registers, two) runs at 0.54 ms for the headline batch's samples at period 256, this
// one at the rate of the 1024-frame-period kernel above.  a.gains = das_pair_gains_interleaved tables (32 x 32 order; re-ordered here).
constexpr int kBlk3 = 768, kWaves3 = kBlk3 / 64;
constexpr int o3Pl = kLdsTw, o3Win = o3Pl + kWaves3 * kPlane, o3G = o3Win + 1024, kLds3 = o3G + 4 * 2048;

template <int CTRL>
__device__ __forceinline__ float dpp_mov_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

template <int LAYOUT, int R>
__global__ __launch_bounds__(kBlk3) void das_fused_small_w64_kernel(DasFusedArgs a, const f32x2 *tw_w64) {
    constexpr int N = 1024 / R, H = N / 2;  // frame length and hop
    constexpr int JS = 64 / R;              // samples between a lane's consecutive registers
    __shared__ __attribute__((aligned(16))) float lds[kLds3];
    const cx<float> *s_tw1 = reinterpret_cast<const cx<float> *>(lds);
    const cx<float> *s_tw2 = s_tw1 + 1024;
    const float *s_win = lds + o3Win;  // window expanded to the interleaved index: w_N[n / R], n < 1024
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *pl = lds + o3Pl + w * kPlane;
    float *row16 = pl + (lane & 15) * kRS + 16 * (lane >> 4);
    float *wcol = pl + w64_col(lane);
    const int M = a.n_mics, NP = (M + 1) >> 1;
    const bool g_lds = NP <= 4 && a.n_dirs == 1;
    {
        const float *tf = reinterpret_cast<const float *>(tw_w64);
        for (int i = tid; i < kLdsTw; i += kBlk3) lds[i] = tf[i];
        for (int i = tid; i < 1024; i += kBlk3) lds[o3Win + i] = a.window[i / R];
        if (g_lds) {
            f32x2 *lg = reinterpret_cast<f32x2 *>(lds + o3G);
            for (int e = tid; e < NP * 1024; e += kBlk3) {  // e = (p * 16 + r) * 64 + lane  <-  bin k = w64_bin(lane, r): position brev5(k >> 5), lane k & 31 of the 32 x 32 table
                const int l = e & 63, r = (e >> 6) & 15, p = e >> 10, k = w64_bin(l, r);
                lg[e] = a.gains[((long)p * 32 + brev5(k >> 5)) * 32 + (k & 31)];
            }
        }
    }
    __syncthreads();
    const long L = a.frames_per_chunk, runs = a.chunks_per_stream;  // L: frames per run, a multiple of R
    const long item = (long)blockIdx.x * kWaves3 + w;
    if (item >= (long)a.n_streams * runs) return;  // no block barrier below
    const int s = (int)(item / runs);              // output stream = input stream * n_dirs + look direction
    const long t0 = (item - (long)s * runs) * L;
    long te = t0 + L;
    if (te > a.n_frames) te = a.n_frames;
    const int in_stream = s / a.n_dirs;
    const f32x2 *gains = a.gains + (long)(s - in_stream * a.n_dirs) * NP * 1024;  // [pair][position][lane] of the 32 x 32 order
    const float *xs = a.x + (long)in_stream * a.stream_stride_x;
    const float *hs = a.hist_in + (long)in_stream * M * H;
    float *ys = a.y + (long)s * a.n_frames * H;
    const int fi = lane % R, c = lane / R;  // this lane's frame inside a group and its sample offset

    auto hop_ptr = [&](long h, int m) -> const float * {  // sample c of hop h (h = -1: the carried hop) of microphone m
        if (LAYOUT == 0) return (h >= 0 ? xs + (long)m * a.mic_stride + h * H : hs + (long)m * H) + c;
        return (h >= 0 ? xs + h * (long)H * M : hs) + (long)c * M + m;
    };
    const long jstep = LAYOUT == 0 ? JS : (long)JS * M;

    float tprev[8];  // second halves of the previous group, windowed; lanes == R - 1 (mod R) feed the next group
    if (t0 == 0) {   // stream start: the carried state (out_buff[0] of the previous call)
        const float *ti = a.tail_in + (long)s * H + c;
#pragma unroll
        for (int j = 0; j < 8; ++j) tprev[j] = ti[JS * j];
    }
    for (long tg = (t0 == 0 ? 0 : t0 - R); tg < te; tg += R) {  // t0 - R: warm-up group, only its last second half is used
        long f = tg + fi;                                         // this lane's frame; past the end: the last frame again, never stored
        const bool f_ok = f < a.n_frames;
        if (!f_ok) f = a.n_frames - 1;
        float Sr[16], Si[16];
        for (int p = 0; p < NP; ++p) {
            float re[16], im[16];
            const int ma = 2 * p, mb = 2 * p + 1;
            {
                const float *q1 = hop_ptr(f - 1, ma), *q2 = hop_ptr(f, ma);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    re[j] = q1[j * jstep];
                    re[j + 8] = q2[j * jstep];
                }
            }
            if (mb < M) {
                const float *r1 = hop_ptr(f - 1, mb), *r2 = hop_ptr(f, mb);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    im[j] = r1[j * jstep];
                    im[j + 8] = r2[j * jstep];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) im[j] = 0.f;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {  // buf[j]*hann_win[i]  (util.h:235); register j <-> interleaved index 64 j + lane
                const float wv = s_win[64 * j + lane];
                re[j] *= wv;
                im[j] *= wv;
            }
            w64_fwd_p1<float>(re, im, lane, s_tw1);
            w64_T1_fwd(re, im, wcol, row16);
            w64_fwd_p2<float>(re, im, lane, s_tw2);
            w64_T2<true>(re, im);
            w64_fwd_p3<float>(re, im);
            if (g_lds) {
                const f32x2 *gp = reinterpret_cast<const f32x2 *>(lds + o3G) + p * 1024 + lane;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const f32x2 g = gp[64 * r];
                    Sr[r] = bf_fma(-g.y, im[r], bf_fma(g.x, re[r], p == 0 ? 0.f : Sr[r]));
                    Si[r] = bf_fma(g.y, re[r], bf_fma(g.x, im[r], p == 0 ? 0.f : Si[r]));
                }
            } else {
                const f32x2 *gp = gains + (long)p * 1024;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k = w64_bin(lane, r);
                    const f32x2 g = gp[brev5(k >> 5) * 32 + (k & 31)];
                    Sr[r] = bf_fma(-g.y, im[r], bf_fma(g.x, re[r], p == 0 ? 0.f : Sr[r]));
                    Si[r] = bf_fma(g.y, re[r], bf_fma(g.x, im[r], p == 0 ? 0.f : Si[r]));
                }
            }
        }
        w64_inv_p3<float>(Sr, Si);
        w64_T2<false>(Sr, Si);
        w64_inv_p2<float>(Sr, Si, lane, s_tw2);
        w64_T1_inv(Sr, Si, row16, wcol);
        w64_inv_p1<float>(Sr, Si, lane, s_tw1);
        // register j <-> interleaved index n = 64 j + lane: j < 8 the first half of this lane's frame, j >= 8 its second half
        const bool store = f_ok && tg >= t0;
        float *yo = ys + f * H + c;
        float tcur[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma clang fp contract(off)
            const float o1 = Sr[j] * s_win[64 * j + lane];              // (float)(Re / N) [inside the gains] times hann (util.h:249-251)
            tcur[j] = Sr[j + 8] * s_win[512 + 64 * j + lane];
            // partner: the second half of the frame before this lane's -- one lane to the left in this group, or (frame 0 of the group) the
            // last frame of the previous group, R - 1 lanes to the right in the previous iteration's values; never across a 16-lane row
            const float pleft = dpp_mov_f<0x111>(tcur[j]);               // row_shr:1
            const float pright = dpp_mov_f<0x100 + (R - 1)>(tprev[j]);   // row_shl:R-1
            const float partner = fi == 0 ? pright : pleft;
            if (store) yo[JS * j] = partner + o1;                        // out = prev[H + n] + cur[n]  (util.h:301-302)
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) tprev[j] = tcur[j];
        if (f_ok && f == a.n_frames - 1 && tg + R >= te) {  // end of the batch: carried state for the next call (OLA tail and the last input hop)
            float *to = a.tail_out + (long)s * H + c;
#pragma unroll
            for (int j = 0; j < 8; ++j) to[JS * j] = tcur[j];
            float *ho = a.hist_out + (long)in_stream * M * H;  // every look direction writes the same values
            if (LAYOUT == 0) {
                for (int m = 0; m < M; ++m)
                    for (int j = 0; j < 8; ++j) ho[m * H + JS * j + c] = xs[(long)m * a.mic_stride + f * H + JS * j + c];
            } else {
                for (int m = 0; m < M; ++m)
                    for (int j = 0; j < 8; ++j) ho[(JS * j + c) * M + m] = xs[(f * (long)H + JS * j + c) * M + m];
            }
        }
    }
}

template <int LAYOUT>
hipError_t launch_small_r(const DasFusedArgs &a, int R, const f32x2 *tw_w64, unsigned blocks, hipStream_t stream) {
    if (R == 2) BF_LAUNCH((das_fused_small_w64_kernel<LAYOUT, 2>), dim3(blocks), dim3(kBlk3), 0, stream, a, tw_w64);
    else if (R == 4) BF_LAUNCH((das_fused_small_w64_kernel<LAYOUT, 4>), dim3(blocks), dim3(kBlk3), 0, stream, a, tw_w64);
    else if (R == 8) BF_LAUNCH((das_fused_small_w64_kernel<LAYOUT, 8>), dim3(blocks), dim3(kBlk3), 0, stream, a, tw_w64);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

template <int LAYOUT>
void launch_layout(const DasFusedArgs &a, unsigned blocks, hipStream_t stream) {
    const int np = (a.n_mics + 1) / 2;
    if (np <= 4)
        BF_LAUNCH((das_fused_w64_kernel<LAYOUT, 4>), dim3(blocks), dim3(kBlock), 0, stream, a);
    else  // > 8 mics: gains from L2
        BF_LAUNCH((das_fused_w64_kernel<LAYOUT, 0>), dim3(blocks), dim3(kBlock), 0, stream, a);
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

// n_fft = 512 / 256 / 128 on the 64-lane transform: a.frames_per_chunk (a multiple of 1024 / n_fft) / a.chunks_per_stream = frames per run / runs
// per OUTPUT stream (one wavefront per run); tw_w64 = twiddle_table_w64(); a.gains = das_pair_gains_interleaved tables; a.window = the n_fft-point
// window; no spectrum dump
hipError_t launch_das_fused_small_w64(const DasFusedArgs &a, int n_fft, const f32x2 *tw_w64, hipStream_t stream) {
    if (a.sdump != nullptr) return hipErrorNotSupported;
    const int R = 1024 / n_fft;
    const long items = (long)a.chunks_per_stream * a.n_streams;
    const unsigned blocks = (unsigned)((items + kWaves3 - 1) / kWaves3);
    return a.layout == 0 ? launch_small_r<0>(a, R, tw_w64, blocks, stream) : launch_small_r<1>(a, R, tw_w64, blocks, stream);
}

// a.frames_per_chunk / a.chunks_per_stream: frames per run and runs per OUTPUT stream (one wavefront per run); tw_split =
// twiddle_table_split2048_w64() (geometry.hpp); a.gains = das_pair_gains_natural tables; no spectrum dump
hipError_t launch_das_fused_2048_w64(const DasFusedArgs &a, const f32x2 *tw_split, hipStream_t stream) {
    if (a.sdump != nullptr) return hipErrorNotSupported;
    const long items = (long)a.chunks_per_stream * a.n_streams;
    const unsigned blocks = (unsigned)((items + kWaves2 - 1) / kWaves2);
    if (a.layout == 0)
        BF_LAUNCH(das_fused_2048_w64_kernel<0>, dim3(blocks), dim3(kBlk2), 0, stream, a, tw_split);
    else
        BF_LAUNCH(das_fused_2048_w64_kernel<1>, dim3(blocks), dim3(kBlk2), 0, stream, a, tw_split);
    return hipGetLastError();
}

}  // namespace bf
