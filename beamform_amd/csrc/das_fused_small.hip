// das_fused_small.hip -- fused fp32 delay-and-sum at the JACK periods below 512 frames (256 / 128 / 64: FFT 512 / 256 / 128;
// rosjack.cpp:131, util.h:261) on the register-resident FFT-1024 of the 512-frame period.
//
// R = 1024 / N consecutive frames are INTERLEAVED into one 1024-point sequence, z[R m + i] = (windowed frame t + i)[m].  Applying the
// same filter to every frame is a circular convolution of z with the zero-stuffed impulse response, whose spectrum is the N-point one
// repeated R times: so the whole per-frame chain -- N-point FFTs of the microphones, S = sum_m conj(w_m) X_m / M per bin
// (das.cpp:60-66), inverse FFT -- of R frames is ONE pass of the 1024-point machinery with the pair gains D_p[k mod N] (1/1024 folded
// in; geometry.hpp das_pair_gains_interleaved), and the small periods run at the speed of the 1024-point transform per SAMPLE instead
// of through LDS-staged generic transforms (das_fused_gen.hip: 3 x slower per sample).  Only the index maps differ from the
// 512-frame kernel: lane l of the half-wavefront holds frame i = l mod R, samples m = (32 / R) j + l / R; first halves are register
// positions n < 512, second halves n >= 512 (as at N = 1024), the overlap-add partner of (frame i, m) is (frame i - 1, m + N/2) =
// one lane to the left, and for i = 0 the last frame of the previous group: R - 1 lanes to the right in the values of the previous
// iteration.  Analysis window util.h:235, synthesis window and float overlap-add util.h:247-252,301-302.
//
// Mapping: a half-wavefront owns (output stream, run of consecutive frame groups) and walks it, the overlap-add tail in 16
// registers; a run that does not start the stream recomputes its previous group.  A 512-thread block per CU: 8 KB twiddles +
// 16 x 4.5 KB exchange planes + 4 KB expanded window + 32 KB pair gains (one look direction, <= 8 microphones; otherwise gains from
// L2).  No spectrum dump: capi.cpp keeps das_fused_gen_kernel for that.
#include <hip/hip_runtime.h>

#include "launch_trace.hpp"
#include "fft1024.hpp"
#include "kernels.hpp"

namespace bf {

namespace {

constexpr int kBlk = 512, kHalves = kBlk / 32;
constexpr int kPSf = plane_stride<float>::value;  // 36
// LDS map (floats)
constexpr int oTw = 0;                           // 1024 complex: inter-pass twiddles of the 32 x 32 factorisation
constexpr int oPl = 2048;                        // 16 planes of 32 x 36
constexpr int oWin = oPl + kHalves * 32 * kPSf;  // window expanded to the interleaved index, w_N[n / R], as [lane][j] rows of 36 floats (ds_read_b128)
constexpr int oG = oWin + 32 * kPSf;              // pair gains [pair][position][lane] complex, 4 pairs
constexpr int kLds = oG + 4 * 2048;

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

template <int LAYOUT, int R>
__global__ __launch_bounds__(kBlk) void das_fused_small_kernel(DasFusedArgs a, const f32x2 *tw1024) {
    constexpr int N = 1024 / R, H = N / 2;  // frame length and hop
    constexpr int JS = 32 / R;              // samples between a lane's consecutive registers
    __shared__ __attribute__((aligned(16))) float lds[kLds];
    const cx<float> *s_tw = reinterpret_cast<const cx<float> *>(lds + oTw);
    const int tid = threadIdx.x, lane = tid & 31, hw = tid >> 5;
    float *pbuf = lds + oPl + hw * 32 * kPSf;
    const int M = a.n_mics, NP = (M + 1) >> 1;
    const bool g_lds = NP <= 4 && a.n_dirs == 1;
    {
        const float *tf = reinterpret_cast<const float *>(tw1024);
        for (int i = tid; i < 2048; i += kBlk) lds[oTw + i] = tf[i];
        for (int i = tid; i < 1024; i += kBlk) lds[oWin + (i & 31) * kPSf + (i >> 5)] = a.window[i / R];  // [lane][j]
        if (g_lds) {
            const float *gf = reinterpret_cast<const float *>(a.gains);
            for (int i = tid; i < NP * 2048; i += kBlk) lds[oG + i] = gf[i];
        }
    }
    __syncthreads();
    const long L = a.frames_per_chunk, runs = a.chunks_per_stream;  // L: frames per run, a multiple of R
    const long item = (long)blockIdx.x * kHalves + hw;
    if (item >= (long)a.n_streams * runs) return;  // no block barrier below
    const int s = (int)(item / runs);              // output stream = input stream * n_dirs + look direction
    const long t0 = (item - (long)s * runs) * L;
    long te = t0 + L;
    if (te > a.n_frames) te = a.n_frames;
    const int in_stream = s / a.n_dirs;
    const f32x2 *gains = a.gains + (long)(s - in_stream * a.n_dirs) * NP * 1024;  // [pair][position][lane], 1/1024 folded in
    const float *xs = a.x + (long)in_stream * a.stream_stride_x;
    const float *hs = a.hist_in + (long)in_stream * M * H;
    float *ys = a.y + (long)s * a.n_frames * H;
    const int fi = lane % R, c = lane / R;  // this lane's frame inside a group and its sample offset
    const float4 *wrow = reinterpret_cast<const float4 *>(lds + oWin + lane * kPSf);  // window of this lane's interleaved indices 32 j + lane

    // sample c of hop h (h = -1: the carried hop) of microphone m; consecutive registers are JS samples apart
    auto hop_ptr = [&](long h, int m) -> const float * {
        if (LAYOUT == 0) return (h >= 0 ? xs + (long)m * a.mic_stride + h * H : hs + (long)m * H) + c;
        return (h >= 0 ? xs + h * (long)H * M : hs) + (long)c * M + m;
    };
    const long jstep = LAYOUT == 0 ? JS : (long)JS * M;

    float tprev[16];  // second halves of the previous group, windowed (position 2 q + 1 -> q); lanes == R - 1 (mod R) feed the next group
    if (t0 == 0) {    // stream start: the carried state (out_buff[0] of the previous call) sits where frame -1 would have left it
        const float *ti = a.tail_in + (long)s * H + c;
#pragma unroll
        for (int q = 0; q < 16; ++q) tprev[q] = ti[JS * brev5(2 * q)];
    }
    for (long tg = (t0 == 0 ? 0 : t0 - R); tg < te; tg += R) {  // t0 - R: warm-up group, only its last second half is used
        long f = tg + fi;                                         // this lane's frame; past the end: the last frame again, never stored
        const bool f_ok = f < a.n_frames;
        if (!f_ok) f = a.n_frames - 1;
        float Sr[32], Si[32];
        for (int p = 0; p < NP; ++p) {
            float re[32], im[32];
            const int ma = 2 * p, mb = 2 * p + 1;
            {
                const float *q1 = hop_ptr(f - 1, ma), *q2 = hop_ptr(f, ma);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    re[j] = q1[j * jstep];
                    re[j + 16] = q2[j * jstep];
                }
            }
            if (mb < M) {
                const float *r1 = hop_ptr(f - 1, mb), *r2 = hop_ptr(f, mb);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    im[j] = r1[j * jstep];
                    im[j + 16] = r2[j * jstep];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 32; ++j) im[j] = 0.f;
            }
#pragma unroll
            for (int g = 0; g < 8; ++g) {  // buf[j]*hann_win[i]  (util.h:235); register j <-> interleaved index 32 j + lane
                const float4 hv = wrow[g];
                re[4 * g + 0] *= hv.x; im[4 * g + 0] *= hv.x;
                re[4 * g + 1] *= hv.y; im[4 * g + 1] *= hv.y;
                re[4 * g + 2] *= hv.z; im[4 * g + 2] *= hv.z;
                re[4 * g + 3] *= hv.w; im[4 * g + 3] *= hv.w;
            }
            fft1024p_fwd_A<float>(re, im, lane, s_tw, pbuf);
            __builtin_amdgcn_wave_barrier();
            fft1024p_B<float>(re, lane, pbuf);
            __builtin_amdgcn_wave_barrier();
            fft1024p_C<float, false>(im, lane, pbuf);
            __builtin_amdgcn_wave_barrier();
            fft1024p_D<float, -1>(re, im, lane, pbuf);
            __builtin_amdgcn_wave_barrier();
            const f32x2 *gp = (g_lds ? reinterpret_cast<const f32x2 *>(lds + oG) : gains) + (long)p * 1024 + lane;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const f32x2 g = gp[32 * i];
                Sr[i] = bf_fma(-g.y, im[i], bf_fma(g.x, re[i], p == 0 ? 0.f : Sr[i]));
                Si[i] = bf_fma(g.y, re[i], bf_fma(g.x, im[i], p == 0 ? 0.f : Si[i]));
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
        // position i <-> interleaved index n = 32 brev5(i) + lane; even i: first half of this lane's frame, odd i: n + 512, its second half
        const bool store = f_ok && tg >= t0;
        float *yo = ys + f * H + c;
        float tcur[16], h[32];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float4 hv = wrow[g];
            h[4 * g + 0] = hv.x; h[4 * g + 1] = hv.y; h[4 * g + 2] = hv.z; h[4 * g + 3] = hv.w;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
#pragma clang fp contract(off)
            const float o1 = Sr[2 * q] * h[brev5(2 * q)];             // (float)(Re / N) [inside the gains] times hann (util.h:249-251)
            tcur[q] = Sr[2 * q + 1] * h[brev5(2 * q + 1)];
            // partner: the second half of the frame before this lane's -- one lane to the left in this group, or (frame 0 of the group) the
            // last frame of the previous group, R - 1 lanes to the right in the previous iteration's values; never across a 16-lane row
            const float pl = dpp_mov<0x111>(tcur[q]);                 // row_shr:1
            const float pr = dpp_mov<0x100 + (R - 1)>(tprev[q]);      // row_shl:R-1
            const float partner = fi == 0 ? pr : pl;
            if (store) yo[JS * brev5(2 * q)] = partner + o1;          // out = prev[H + n] + cur[n]  (util.h:301-302)
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) tprev[q] = tcur[q];
        if (f_ok && f == a.n_frames - 1 && tg + R >= te) {  // end of the batch: carried state for the next call (OLA tail and the last input hop)
            float *to = a.tail_out + (long)s * H + c;
#pragma unroll
            for (int q = 0; q < 16; ++q) to[JS * brev5(2 * q)] = tcur[q];
            float *ho = a.hist_out + (long)in_stream * M * H;  // every look direction writes the same values
            if (LAYOUT == 0) {
                for (int m = 0; m < M; ++m)
                    for (int j = 0; j < 16; ++j) ho[m * H + JS * j + c] = xs[(long)m * a.mic_stride + f * H + JS * j + c];
            } else {
                for (int m = 0; m < M; ++m)
                    for (int j = 0; j < 16; ++j) ho[(JS * j + c) * M + m] = xs[(f * (long)H + JS * j + c) * M + m];
            }
        }
    }
}

template <int LAYOUT>
hipError_t launch_r(const DasFusedArgs &a, int R, const f32x2 *tw1024, unsigned blocks, hipStream_t stream) {
    if (R == 2) BF_LAUNCH((das_fused_small_kernel<LAYOUT, 2>), dim3(blocks), dim3(kBlk), 0, stream, a, tw1024);
    else if (R == 4) BF_LAUNCH((das_fused_small_kernel<LAYOUT, 4>), dim3(blocks), dim3(kBlk), 0, stream, a, tw1024);
    else if (R == 8) BF_LAUNCH((das_fused_small_kernel<LAYOUT, 8>), dim3(blocks), dim3(kBlk), 0, stream, a, tw1024);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace

// n_fft = 512 / 256 / 128; a.frames_per_chunk (a multiple of 1024 / n_fft) / a.chunks_per_stream: frames per run and runs per OUTPUT
// stream (one half-wavefront per run); tw1024 = twiddle_table_32x32<f32x2>(); a.gains = das_pair_gains_interleaved tables; a.window =
// the n_fft-point window; no spectrum dump
hipError_t launch_das_fused_small(const DasFusedArgs &a, int n_fft, const f32x2 *tw1024, hipStream_t stream) {
    if (a.sdump != nullptr) return hipErrorNotSupported;
    const int R = 1024 / n_fft;
    const long items = (long)a.chunks_per_stream * a.n_streams;
    const unsigned blocks = (unsigned)((items + kHalves - 1) / kHalves);
    return a.layout == 0 ? launch_r<0>(a, R, tw1024, blocks, stream) : launch_r<1>(a, R, tw1024, blocks, stream);
}

}  // namespace bf
