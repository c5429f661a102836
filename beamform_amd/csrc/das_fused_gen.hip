// das_fused_gen.hip -- fused fp32 delay-and-sum for the JACK periods the in-register 32 x 32 transform does not cover:
// every power of two from 64 to 4096 frames except 512 (FFT 128 ... 8192); the reference takes whatever period the server
// reports (rosjack.cpp:131-134) and sets fft_win = 2 * period (util.h:261).
//
// Same formulation as das_fused.hip -- per frame ceil(M/2) packed forward transforms, S += D_p Z_p with the pair gains of
// geometry.hpp (natural bin order here), one backward transform, synthesis window, float overlap-add (das.cpp:47-70,
// util.h:217-253,301-302) -- in ONE launch, spectra never leaving the CU.  The transforms are LDS-staged radix-4 Stockham
// passes of one block (256 threads; N / 2 threads below N = 512; one frame at a time per block): a plain design, 3-4 x the
// three-kernel fp64 chain these periods ran before, not the register-resident machinery of the 512-frame period.  A block owns
// a run of consecutive frames and keeps the overlap-add tail in LDS; a run that does not start the stream recomputes its
// previous frame for that tail.  N = 8192 reads twiddles and window through L1 (its two transform buffers fill the LDS).
#include <hip/hip_runtime.h>

#include "launch_trace.hpp"
#include "kernels.hpp"

namespace bf {

namespace {

constexpr int gen_block(int n) { return n >= 512 ? 256 : n / 2; }  // threads per block: at least two points per thread

// autosort (Stockham) passes in LDS: data starts in b0, result ends in the returned buffer.  Radix-4 passes (N = 512: four + one
// radix-2, N = 2048: five + one radix-2; earlier: nine / eleven radix-2 passes: half the barriers, half the LDS traffic).  Twiddles:
// c4 = the per-pass blocks of geometry.hpp stockham_twiddles (the three factors of butterfly k contiguous in k: conflict-free reads),
// tw = W^m, m < N/2, read in order by the closing radix-2 pass.
template <int N, int DIR>
__device__ __forceinline__ float2 *stockham32(float2 *b0, float2 *b1, const float2 *c4, const float2 *tw, int tid) {
    constexpr int kGB = gen_block(N);
    float2 *in = b0, *out = b1;
    auto dirw = [](float2 w) { return float2{w.x, DIR < 0 ? w.y : -w.y}; };  // W (forward) or its conjugate (backward)
    auto cmul = [](float2 v, float2 w) { return float2{v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x}; };
    int ns = 1;
#pragma unroll 1
    for (; ns * 4 <= N; ns <<= 2) {
        const float2 *cp = c4 + (ns - 1);
#pragma unroll
        for (int j = tid; j < N / 4; j += kGB) {
            const int k = j & (ns - 1);
            const float2 a0 = in[j];
            const float2 a1 = cmul(in[j + N / 4], dirw(cp[k]));
            const float2 a2 = cmul(in[j + N / 2], dirw(cp[ns + k]));
            const float2 a3 = cmul(in[j + 3 * N / 4], dirw(cp[2 * ns + k]));
            const float2 s02 = float2{a0.x + a2.x, a0.y + a2.y}, d02 = float2{a0.x - a2.x, a0.y - a2.y};
            const float2 s13 = float2{a1.x + a3.x, a1.y + a3.y}, d13 = float2{a1.x - a3.x, a1.y - a3.y};
            // forward: -i d13 = (d13.y, -d13.x); backward: +i d13 = (-d13.y, d13.x)
            const float2 r13 = DIR < 0 ? float2{d13.y, -d13.x} : float2{-d13.y, d13.x};
            const int j0 = ((j - k) << 2) + k;
            out[j0] = float2{s02.x + s13.x, s02.y + s13.y};
            out[j0 + ns] = float2{d02.x + r13.x, d02.y + r13.y};
            out[j0 + 2 * ns] = float2{s02.x - s13.x, s02.y - s13.y};
            out[j0 + 3 * ns] = float2{d02.x - r13.x, d02.y - r13.y};
        }
        __syncthreads();
        float2 *t = in;
        in = out;
        out = t;
    }
    if (ns < N) {  // one radix-2 pass left (N = 2 * 4^k): ns = N / 2, twiddle W^k
#pragma unroll
        for (int j = tid; j < N / 2; j += kGB) {
            const float2 u = in[j], b = cmul(in[j + N / 2], dirw(tw[j]));
            out[j] = float2{u.x + b.x, u.y + b.y};
            out[j + N / 2] = float2{u.x - b.x, u.y - b.y};
        }
        __syncthreads();
        float2 *t = in;
        in = out;
        out = t;
    }
    return in;
}

template <int N>
__global__ __launch_bounds__(gen_block(N)) void das_fused_gen_kernel(DasFusedArgs a) {
    constexpr int kGB = gen_block(N);
    constexpr int H = N / 2, BPT = N / kGB;  // bins (samples) per thread: 2 .. 32
    constexpr bool kTabLds = N <= 4096;      // N = 8192: 2 x 64 KB of transform buffers + 16 KB of tail leave no room for the tables
    constexpr int kR4 = stockham_r4_entries(N);
    constexpr bool kTw2Lds = kTabLds && N <= 1024;  // the radix-2 pass' table: in order, L1 serves it as well (N = 2048: keeps three blocks per CU)
    __shared__ float2 s_a[N], s_b[N], s_c4l[kTabLds ? kR4 : 1], s_twl[kTw2Lds ? N / 2 : 1];
    __shared__ float s_winl[kTabLds ? N : 1], s_tail[H];
    const float2 *s_c4 = kTabLds ? s_c4l : reinterpret_cast<const float2 *>(a.twiddle) + N / 2;
    const float2 *s_tw = kTw2Lds ? s_twl : reinterpret_cast<const float2 *>(a.twiddle);
    const float *s_win = kTabLds ? s_winl : a.window;
    const int tid = threadIdx.x;
    const int M = a.n_mics, n_pairs = (M + 1) >> 1;
    const int stream = blockIdx.x / a.chunks_per_stream;  // output stream = input stream * n_dirs + look direction
    const long c_in_s = blockIdx.x - (long)stream * a.chunks_per_stream;
    const int in_stream = stream / a.n_dirs;
    const f32x2 *gains = a.gains + (long)(stream - in_stream * a.n_dirs) * n_pairs * N;  // [pair][bin], 1/N folded in
    if (kTabLds) {
        for (int i = tid; i < kR4; i += kGB) s_c4l[i] = float2{a.twiddle[N / 2 + i].x, a.twiddle[N / 2 + i].y};
        if (kTw2Lds)
            for (int i = tid; i < N / 2; i += kGB) s_twl[i] = float2{a.twiddle[i].x, a.twiddle[i].y};
        for (int i = tid; i < N; i += kGB) s_winl[i] = a.window[i];
    }
    const long T0 = c_in_s * a.frames_per_chunk;
    long T1 = T0 + a.frames_per_chunk;
    if (T1 > a.n_frames) T1 = a.n_frames;
    if (T0 == 0)
        for (int i = tid; i < H; i += kGB) s_tail[i] = a.tail_in[(long)stream * H + i];
    __syncthreads();
    const float *xs = a.x + (long)in_stream * a.stream_stride_x;
    const float *hs = a.hist_in + (long)in_stream * M * H;
    float *ys = a.y + (long)stream * a.n_frames * H;

    for (long t = (T0 == 0 ? 0 : T0 - 1); t < T1; ++t) {  // T0 - 1: warm-up frame, only its second half (the tail) is used
        float2 S[BPT];
#pragma unroll
        for (int i = 0; i < BPT; ++i) S[i] = float2{0.f, 0.f};
        for (int p = 0; p < n_pairs; ++p) {
            const int ma = 2 * p, mb = 2 * p + 1;
            const bool b_ok = mb < M;
#pragma unroll
            for (int i = 0; i < BPT; ++i) {
                const int n = tid + kGB * i;
                const bool first = n < H;  // first half of the frame = the hop before hop t (the carried hop at t = 0)
                const int k = first ? n : n - H;
                float va, vb = 0.f;
                if (a.layout == 0) {
                    const float *ba = first ? (t >= 1 ? xs + (long)ma * a.mic_stride + (t - 1) * H : hs + ma * H) : xs + (long)ma * a.mic_stride + t * H;
                    va = ba[k];
                    if (b_ok) {
                        const float *bb = first ? (t >= 1 ? xs + (long)mb * a.mic_stride + (t - 1) * H : hs + mb * H) : xs + (long)mb * a.mic_stride + t * H;
                        vb = bb[k];
                    }
                } else {
                    const float *bs = first ? (t >= 1 ? xs + (t - 1) * (long)H * M : hs) : xs + t * (long)H * M;
                    va = bs[(long)k * M + ma];
                    if (b_ok) vb = bs[(long)k * M + mb];
                }
                const float w = s_win[n];
                s_a[n] = float2{va * w, vb * w};  // buf[j]*hann_win[i]  (util.h:235)
            }
            __syncthreads();
            const float2 *Z = stockham32<N, -1>(s_a, s_b, s_c4, s_tw, tid);
            const f32x2 *gp = gains + (long)p * N;
#pragma unroll
            for (int i = 0; i < BPT; ++i) {
                const int k = tid + kGB * i;
                const f32x2 g = gp[k];
                const float2 z = Z[k];
                S[i].x = __builtin_fmaf(-g.y, z.y, __builtin_fmaf(g.x, z.x, S[i].x));
                S[i].y = __builtin_fmaf(g.y, z.x, __builtin_fmaf(g.x, z.y, S[i].y));
            }
            __syncthreads();  // Z is rewritten by the next pair's samples
        }
        if (a.sdump != nullptr && t >= T0) {
            f32x2 *sd = a.sdump + ((long)stream * a.n_frames + t) * N;
#pragma unroll
            for (int i = 0; i < BPT; ++i) sd[tid + kGB * i] = f32x2{S[i].x, S[i].y};
        }
#pragma unroll
        for (int i = 0; i < BPT; ++i) s_a[tid + kGB * i] = S[i];
        __syncthreads();
        const float2 *Y = stockham32<N, +1>(s_a, s_b, s_c4, s_tw, tid);
        float o[BPT];
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const int n = tid + kGB * i;
            {
#pragma clang fp contract(off)
                o[i] = Y[n].x * s_win[n];  // (float)(Re / N) [1/N inside the gains] times the synthesis window (util.h:249-251)
                if (n < H && t >= T0) ys[t * H + n] = s_tail[n] + o[i];  // out = prev[H + n] + cur[n]  (util.h:301-302)
            }
        }
        __syncthreads();  // every read of the old tail is done
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const int n = tid + kGB * i;
            if (n >= H) s_tail[n - H] = o[i];
        }
        __syncthreads();
        if (t == a.n_frames - 1) {  // end of the batch: carried state for the next call (OLA tail and the last input hop)
            for (int i = tid; i < H; i += kGB) a.tail_out[(long)stream * H + i] = s_tail[i];
            float *ho = a.hist_out + (long)in_stream * M * H;  // every direction writes the same values
            if (a.layout == 0) {
                for (int i = tid; i < M * H; i += kGB) ho[i] = xs[(long)(i / H) * a.mic_stride + t * H + (i % H)];
            } else {
                for (int i = tid; i < M * H; i += kGB) ho[i] = xs[t * (long)H * M + i];
            }
        }
    }
}

__global__ void das_hermitian_dump_gen_kernel(const f32x2 *s, f64x2 *out, long total, int N) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long f = idx / N;
    const int k = (int)(idx - f * N);
    const f32x2 u = s[f * N + k];
    const f32x2 v = s[f * N + ((N - k) & (N - 1))];
    // undo the folded 1/N; Hermitian part (S[k] + conj(S[N-k]))/2
    out[idx] = f64x2{0.5 * N * ((double)u.x + (double)v.x), 0.5 * N * ((double)u.y - (double)v.y)};
}

}  // namespace

hipError_t launch_das_fused_gen(const DasFusedArgs &a, int n_fft, hipStream_t stream) {
    const unsigned blocks = (unsigned)((long)a.chunks_per_stream * a.n_streams);
#define BF_GEN(N_) case N_: BF_LAUNCH(das_fused_gen_kernel<N_>, dim3(blocks), dim3(gen_block(N_)), 0, stream, a); break
    switch (n_fft) {
        BF_GEN(128); BF_GEN(256); BF_GEN(512); BF_GEN(2048); BF_GEN(4096); BF_GEN(8192);
        default: return hipErrorInvalidValue;
    }
#undef BF_GEN
    return hipGetLastError();
}

hipError_t launch_das_hermitian_dump_gen(const f32x2 *sdump, f64x2 *out, long n_frames_total, int n_fft, hipStream_t stream) {
    const long total = n_frames_total * n_fft;
    BF_LAUNCH(das_hermitian_dump_gen_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, sdump, out, total, n_fft);
    return hipGetLastError();
}

}  // namespace bf
