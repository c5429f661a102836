// das_fused_2048.hip -- fused fp32 delay-and-sum at the JACK period of 1024 frames (FFT 2048; rosjack.cpp:131, util.h:261) on the
// in-register 32 x 32 FFT-1024 of the 512-frame period instead of the LDS-staged generic transforms (das_fused_gen.hip).
//
// One radix-2 step splits the 2048-point transforms into 1024-point ones.  With the frame x[n] (n < 2048; first half = hop t - 1,
// second half = hop t), the analysis window w (util.h:235) and W = exp(-2 pi i / 2048):
//   e[n] = w[n] x[n] + w[n + 1024] x[n + 1024]                    FFT1024(e)[k] = X[2k]
//   o[n] = (w[n] x[n] - w[n + 1024] x[n + 1024]) W^n              FFT1024(o)[k] = X[2k + 1]          (n, k < 1024)
// so a frame is two passes of the N = 1024 machinery: per pass ceil(M/2) packed forward transforms, S += D_p Z_p with the pair gains
// of the even (odd) bins, one backward transform.  With A = IFFT1024(S_even), B = IFFT1024(S_odd) (unnormalised; 1/N is inside the
// gains) the frame's output is
//   y[n] = Re A[n] + Re(conj(W^n) B[n]),   y[n + 1024] = Re A[n] - Re(conj(W^n) B[n]),
// then the synthesis window and the float overlap-add (util.h:247-252,301-302), as das_fused_gen.hip does them.  The odd pass runs
// first and leaves 32 reals per lane; the even pass' backward transform ends with the output in registers.
//
// Mapping: a half-wavefront owns (output stream, run of consecutive frames) and walks the run; the overlap-add tail waits in the output
// buffer (see below); a run that does not start the stream recomputes its previous frame for that tail.  A 256-thread block per CU:
// 8 KB twiddles + 8 x 4.5 KB exchange planes + 8 KB W^n + 64 KB pair gains + 8 KB window = 124 KB of LDS, one wavefront per SIMD
// (the two register arrays of a pass plus the odd pass' 32 reals and the loads in flight come to ~350 registers; a 512-thread build
// spills 155 of them and runs slower).  One look direction and up to 8 microphones: the gains sit in LDS in the order the passes
// read them; otherwise they come from L2 in natural bin order (the table das_fused_gen.hip reads).  The input is read twice per
// frame (once per pass), the second time from L2.  No spectrum dump: capi.cpp keeps das_fused_gen_kernel<2048> for that.
// MI355X, 8 microphones, 32 768 frames: 1.03 ms against 1.61 ms for the generic kernel (BF_DAS_SPLIT2048=0).
#include <hip/hip_runtime.h>

#include "launch_trace.hpp"
#include "fft1024.hpp"
#include "kernels.hpp"

namespace bf {

namespace {

constexpr int kBlk = 256, kHalves = kBlk / 32;  // one wavefront per SIMD (a 512-thread block spills 155 registers to scratch)
constexpr int kH = 1024;                            // hop = JACK period
constexpr int kPSf = plane_stride<float>::value;    // 36
// LDS map (floats): 124 KB
constexpr int oTw = 0;                              // 1024 complex: inter-pass twiddles of the 32 x 32 factorisation
constexpr int oPl = 2048;                           // 8 planes of 32 x 36
constexpr int oW = oPl + kHalves * 32 * kPSf;       // W^n, n < 1024, complex
constexpr int oG = oW + 2048;                       // pair gains [pass][pair][position][lane] complex: 4 pairs x 2048 bins
constexpr int oWin = oG + 4 * 2 * 2048;             // 2048 window values
constexpr int kLds = oWin + 2048;

template <int LAYOUT>
__global__ __launch_bounds__(kBlk) void das_fused_2048_kernel(DasFusedArgs a, const f32x2 *tw_split) {
    __shared__ __attribute__((aligned(16))) float lds[kLds];
    const cx<float> *s_tw = reinterpret_cast<const cx<float> *>(lds + oTw);
    const float *s_win = lds + oWin;
    const f32x2 *s_w = reinterpret_cast<const f32x2 *>(lds + oW);
    const int tid = threadIdx.x, lane = tid & 31, hw = tid >> 5;
    float *pbuf = lds + oPl + hw * 32 * kPSf;
    {
        const float *tf = reinterpret_cast<const float *>(tw_split);  // [32 x 32 twiddles | W^n]
        for (int i = tid; i < 2048; i += kBlk) lds[oTw + i] = tf[i];
        for (int i = tid; i < 2048; i += kBlk) lds[oW + i] = tf[2048 + i];
        for (int i = tid; i < 2048; i += kBlk) lds[oWin + i] = a.window[i];
    }
    const int M = a.n_mics, NP = (M + 1) >> 1;
    const bool g_lds = NP <= 4 && a.n_dirs == 1;  // one look direction, up to 8 microphones: the gains fit the LDS in the order the passes read them
    if (g_lds) {
        f32x2 *lg = reinterpret_cast<f32x2 *>(lds + oG);
        for (int e = tid; e < NP * 2048; e += kBlk) {  // e = ((pass * NP + p) * 32 + i) * 32 + lane  <-  bin 2 (lane + 32 brev5(i)) + pass of pair p
            const int l = e & 31, i = (e >> 5) & 31, pp = e >> 10, p = pp % NP, pass = pp / NP;
            lg[e] = a.gains[(long)p * 2048 + 2 * (l + 32 * brev5(i)) + pass];
        }
    }
    __syncthreads();
    const long L = a.frames_per_chunk, runs = a.chunks_per_stream;
    const long item = (long)blockIdx.x * kHalves + hw;
    if (item >= (long)a.n_streams * runs) return;  // no block barrier below
    const int s = (int)(item / runs);              // output stream = input stream * n_dirs + look direction
    const long t0 = (item - (long)s * runs) * L;
    long te = t0 + L;
    if (te > a.n_frames) te = a.n_frames;
    const int in_stream = s / a.n_dirs;
    const f32x2 *gains = a.gains + (long)(s - in_stream * a.n_dirs) * NP * 2048;  // [pair][bin], 1/N folded in
    const float *xs = a.x + (long)in_stream * a.stream_stride_x;
    const float *hs = a.hist_in + (long)in_stream * M * kH;
    float *ys = a.y + (long)s * a.n_frames * kH;

    // sample 32 j + lane of hop h (h = -1: the carried hop) of microphone m
    auto hop_ptr = [&](long h, int m) -> const float * {
        if (LAYOUT == 0) return (h >= 0 ? xs + (long)m * a.mic_stride + h * kH : hs + (long)m * kH) + lane;
        return (h >= 0 ? xs + h * (long)kH * M : hs) + (long)lane * M + m;
    };
    constexpr int kStep = 32;  // samples between a lane's consecutive registers
    const long jstride = LAYOUT == 0 ? kStep : (long)kStep * M;

    // The second half of a frame (its overlap-add tail) waits in the output buffer itself: stored as hop t + 1's content, read back and
    // completed by the next frame of the run (same lane, same addresses, program order).  32 registers less per lane.
    const long tb = t0 == 0 ? 0 : t0 - 1;
    for (long t = tb; t < te; ++t) {  // t0 - 1: warm-up frame, only its second half (the tail) is used
        float v[32];  // Re(conj(W^n) B[n]) of the odd pass
        float Sr[32], Si[32];
        for (int pass = 1; pass >= 0; --pass) {  // odd bins first
#pragma unroll
            for (int i = 0; i < 32; ++i) Sr[i] = Si[i] = 0.f;
            for (int p = 0; p < NP; ++p) {
                float re[32], im[32];
                const int ma = 2 * p, mb = 2 * p + 1;
                const bool b_ok = mb < M;
                const float sg = pass ? -1.f : 1.f;  // the sign of the second half's term
                {   // channel a, then channel b: one channel's two hops in flight at a time
                    float x2[32];
                    const float *q1 = hop_ptr(t - 1, ma), *q2 = hop_ptr(t, ma);
#pragma unroll
                    for (int j = 0; j < 32; ++j) re[j] = q1[j * jstride];
#pragma unroll
                    for (int j = 0; j < 32; ++j) x2[j] = q2[j * jstride];
                    // e / o (the sign of the second half's term is the pass); buf[j]*hann_win[i]  (util.h:235)
#pragma unroll
                    for (int j = 0; j < 32; ++j) {
                        const float w1 = s_win[32 * j + lane], w2 = s_win[kH + 32 * j + lane];
                        re[j] = bf_fma(x2[j], w2 * sg, re[j] * w1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (b_ok) {
                        const float *r1 = hop_ptr(t - 1, mb), *r2 = hop_ptr(t, mb);
#pragma unroll
                        for (int j = 0; j < 32; ++j) im[j] = r1[j * jstride];
#pragma unroll
                        for (int j = 0; j < 32; ++j) x2[j] = r2[j * jstride];
#pragma unroll
                        for (int j = 0; j < 32; ++j) {
                            const float w1 = s_win[32 * j + lane], w2 = s_win[kH + 32 * j + lane];
                            im[j] = bf_fma(x2[j], w2 * sg, im[j] * w1);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 32; ++j) im[j] = 0.f;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (pass) {  // o[n] *= W^n
#pragma unroll
                    for (int j = 0; j < 32; ++j) {
                        const f32x2 w = s_w[32 * j + lane];
                        const float xr = re[j], xi = im[j];
                        re[j] = xr * w.x - xi * w.y;
                        im[j] = xr * w.y + xi * w.x;
                    }
                }
                fft1024p_fwd_A<float>(re, im, lane, s_tw, pbuf);
                __builtin_amdgcn_wave_barrier();
                fft1024p_B<float>(re, lane, pbuf);
                __builtin_amdgcn_wave_barrier();
                fft1024p_C<float, false>(im, lane, pbuf);
                __builtin_amdgcn_wave_barrier();
                fft1024p_D<float, -1>(re, im, lane, pbuf);
                __builtin_amdgcn_wave_barrier();
                // position i holds bin k = lane + 32 brev5(i) of this pass' transform = bin 2 k + pass of the frame
                const f32x2 *gp = g_lds ? reinterpret_cast<const f32x2 *>(lds + oG) + (pass * NP + p) * 1024 + lane : gains + (long)p * 2048 + 2 * lane + pass;
                const int gs = g_lds ? 32 : 64;  // elements between positions
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const f32x2 g = g_lds ? gp[gs * i] : gp[gs * brev5(i)];
                    Sr[i] = bf_fma(-g.y, im[i], bf_fma(g.x, re[i], Sr[i]));
                    Si[i] = bf_fma(g.y, re[i], bf_fma(g.x, im[i], Si[i]));
                }
            }
            fft1024p_inv_A<float>(Sr, Si, lane, s_tw, pbuf);
            __builtin_amdgcn_wave_barrier();
            fft1024p_B<float>(Sr, lane, pbuf);
            __builtin_amdgcn_wave_barrier();
            fft1024p_C<float, true>(Si, lane, pbuf);
            __builtin_amdgcn_wave_barrier();
            fft1024p_D<float, +1>(Sr, Si, lane, pbuf);
            __builtin_amdgcn_wave_barrier();
            if (pass) {  // position i <-> n = 32 brev5(i) + lane: Re(conj(W^n) B[n])
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const f32x2 w = s_w[32 * brev5(i) + lane];
                    v[i] = Sr[i] * w.x + Si[i] * w.y;
                }
            }
        }
        // Sr = Re A.  First half: sample n, second half: sample n + 1024; synthesis window and overlap-add with the float stores
        float *yo = ys + t * kH + lane;
        const float *prev = (t == 0) ? a.tail_in + (long)s * kH + lane : yo;  // the tail parked by frame t - 1 (the carried state at t = 0)
        const bool store = t >= t0, park = t + 1 < te;
        float *yn = yo + kH;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int n = 32 * brev5(i);
            {
#pragma clang fp contract(off)
                const float o1 = (Sr[i] + v[i]) * s_win[n + lane];        // (float)(Re / N) [1/N inside the gains] times hann (util.h:249-251)
                const float o2 = (Sr[i] - v[i]) * s_win[kH + n + lane];
                if (store) yo[n] = prev[n] + o1;                          // out = prev[H + n] + cur[n]  (util.h:301-302)
                if (park) yn[n] = o2;                                     // completed by the run's next frame
                Sr[i] = o2;
            }
        }
        if (t == a.n_frames - 1) {  // end of the batch: carried state for the next call (OLA tail and the last input hop)
            float *to = a.tail_out + (long)s * kH + lane;
#pragma unroll
            for (int i = 0; i < 32; ++i) to[32 * brev5(i)] = Sr[i];
            float *ho = a.hist_out + (long)in_stream * M * kH;  // every look direction writes the same values
            if (LAYOUT == 0) {
                for (int m = 0; m < M; ++m)
                    for (int j = 0; j < 32; ++j) ho[m * kH + 32 * j + lane] = xs[(long)m * a.mic_stride + t * kH + 32 * j + lane];
            } else {
                for (int j = 0; j < 32 * M; ++j) ho[32 * j + lane] = xs[t * (long)kH * M + 32 * j + lane];
            }
        }
    }
}

}  // namespace

// a.frames_per_chunk / a.chunks_per_stream: frames per run and runs per OUTPUT stream (one half-wavefront per run); tw_split =
// twiddle_table_split2048() (geometry.hpp); a.gains = das_pair_gains_natural tables; no spectrum dump
hipError_t launch_das_fused_2048(const DasFusedArgs &a, const f32x2 *tw_split, hipStream_t stream) {
    if (a.sdump != nullptr) return hipErrorNotSupported;
    const long items = (long)a.chunks_per_stream * a.n_streams;
    const unsigned blocks = (unsigned)((items + kHalves - 1) / kHalves);
    if (a.layout == 0)
        BF_LAUNCH(das_fused_2048_kernel<0>, dim3(blocks), dim3(kBlk), 0, stream, a, tw_split);
    else
        BF_LAUNCH(das_fused_2048_kernel<1>, dim3(blocks), dim3(kBlk), 0, stream, a, tw_split);
    return hipGetLastError();
}

}  // namespace bf
