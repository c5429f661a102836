// stft_istft.hip -- the two ends of the fp64 bin pipeline: STFT (window + forward FFTs, packed pair spectra to HBM), ISTFT
// (Hermitian extension, backward FFT, synthesis window, overlap-add), the full-spectrum dump and phasempf's output smoothing.
#include <cstdlib>

#include "launch_trace.hpp"
#include "bins_common.hpp"
#include "fft_small.hpp"
#if BF_NFFT == 1024
#include "w64_f64_dev.hpp"
#endif

namespace bf {
namespace BF_NTAG {

namespace {

#if BF_NFFT == 1024
// ======================================================================================
//                                        STFT
// ======================================================================================
// One 256-thread block per CU (LDS = 16 KB inter-pass twiddles + 8 x 8.5 KB transpose planes = 84 KB; a 512-thread block
// spills at 256 registers).  A half-wavefront owns (stream, microphone pair, run of `run_len` consecutive frames) and walks
// the run: the hop two consecutive frames share (50 % overlap, util.h:217-242) stays in registers as raw float samples, so
// every input sample is fetched ONCE (the item-per-frame version fetched 1.5x: counters, profiles/traffic_mvdr8.json of
// round 2), and the next hop is requested before the current frame is transformed -- a wavefront alone on its SIMD has
// nobody else to hide the load latency behind.  The window is read from LDS (rows per lane).
// Z48: the packed pair spectrum leaves as z48 elements (mvdr / lcmv).
template <int LAYOUT, bool Z48>
__global__ __launch_bounds__(256) void stft_kernel(StftArgs a) {
    constexpr int kStftBlock = 256, kStftHalves = kStftBlock / 32;
    // the window sits in LDS as [lane][j] rows of 34 doubles (272 B: the 16 lanes of a ds_read_b128 group land 4 banks apart): in
    // registers it cost 64 of the 256 VGPRs beside 128 of data and 96 of carried hops, and hipcc moved ~450 values per frame
    // through the accumulator registers to make room
    constexpr int kWinRow = 34;
    __shared__ __attribute__((aligned(16))) double lds[2048 + kStftHalves * 32 * kPSd + 32 * kWinRow];
    const cx<double> *s_tw = reinterpret_cast<const cx<double> *>(lds);
    const int tid = threadIdx.x, lane = tid & 31, hw = tid >> 5;
    double *pbuf = lds + 2048 + hw * 32 * kPSd;
    double *s_win = lds + 2048 + kStftHalves * 32 * kPSd;
    {
        const double *twf = reinterpret_cast<const double *>(a.tw);
        for (int i = tid; i < 2048; i += kStftBlock) lds[i] = twf[i];
        for (int i = tid; i < kN; i += kStftBlock) s_win[(i & 31) * kWinRow + (i >> 5)] = a.win[i] * (a.halve ? 0.5 : 1.0);  // mvdr / lcmv spectra are stored halved (exact): unpacking a pair is then Z[k] +- conj Z[N-k] without the 1/2
        __syncthreads();
    }
    const f64x2 *wrow = reinterpret_cast<const f64x2 *>(s_win + lane * kWinRow);
    const int M = a.n_mics, MF = a.n_fft_mics, NP = (MF + 1) >> 1, L = a.run_len;
    const long runs = (a.n_frames + L - 1) / L;
    const long total = (long)a.n_streams * runs * NP;
    const long stride = (long)gridDim.x * kStftHalves;
    const long rounds = (total + stride - 1) / stride;
    for (long r = 0; r < rounds; ++r) {
        long item = r * stride + (long)blockIdx.x * kStftHalves + hw;
        const bool ok = item < total;
        if (!ok) item = total - 1;
        const int p = (int)(item % NP);
        const long sr = item / NP;
        const long run = sr % runs;
        const int s = (int)(sr / runs);
        const long t0 = run * L;
        const float *xs = a.x + (long)s * a.stream_stride_x;
        const float *hs = a.hist + (long)s * M * kHop;
        const int ma = 2 * p;
        const bool b_ok = 2 * p + 1 < MF;
        const int mb = b_ok ? 2 * p + 1 : ma;
        const double bs = b_ok ? 1.0 : 0.0;
        // hop h of this pair (h = -1: the carried hop in front of the batch) as raw samples, lane l <- sample 32 j + l
        auto load_hop = [&](long h, float (&va)[16], float (&vb)[16]) {
            if (LAYOUT == 0) {
                const float *pa = (h >= 0 ? xs + (long)ma * a.mic_stride + h * kHop : hs + ma * kHop) + lane;
                const float *pb = (h >= 0 ? xs + (long)mb * a.mic_stride + h * kHop : hs + mb * kHop) + lane;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    va[j] = pa[32 * j];
                    vb[j] = pb[32 * j];
                }
            } else {
                const float *ps = (h >= 0 ? xs + h * (long)kHop * M : hs) + (long)lane * M;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    va[j] = ps[(long)32 * j * M + ma];
                    vb[j] = ps[(long)32 * j * M + mb];
                }
            }
        };
        float pa[16], pb[16], ca[16], cb[16], na[16], nb[16];
        load_hop(t0 - 1, pa, pb);
        load_hop(t0, ca, cb);
        for (int it = 0; it < L; ++it) {
            const long t = t0 + it;
            const bool t_ok = ok && t < a.n_frames;
            {  // next hop: in flight during this frame's transform
                long tn = t + 1;
                if (tn >= a.n_frames) tn = a.n_frames - 1;
                load_hop(tn, na, nb);
            }
            double re[32], im[32];
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                const f64x2 w0 = wrow[j >> 1], w1 = wrow[8 + (j >> 1)];  // win[j], win[j+1] / win[j+16], win[j+17]
                re[j] = (double)pa[j] * w0.x;  // buf[j]*hann_win[i]  (util.h:235)
                im[j] = (double)pb[j] * (w0.x * bs);
                re[j + 1] = (double)pa[j + 1] * w0.y;
                im[j + 1] = (double)pb[j + 1] * (w0.y * bs);
                re[j + 16] = (double)ca[j] * w1.x;
                im[j + 16] = (double)cb[j] * (w1.x * bs);
                re[j + 17] = (double)ca[j + 1] * w1.y;
                im[j + 17] = (double)cb[j + 1] * (w1.y * bs);
            }
            fft1024p_fwd_A<double>(re, im, lane, s_tw, pbuf);
            __builtin_amdgcn_wave_barrier();
            fft1024p_B<double>(re, lane, pbuf);
            __builtin_amdgcn_wave_barrier();
            fft1024p_C<double, false>(im, lane, pbuf);
            __builtin_amdgcn_wave_barrier();
            fft1024p_D<double, -1>(re, im, lane, pbuf);
            __builtin_amdgcn_wave_barrier();
            if (t_ok) {
                const long zoff = (((long)s * a.frames_ws + a.frame_off + t) * NP + p) * kN + lane;
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const int row = 32 * brev5(i);  // bins row .. row+31 of this store
                    if (row > a.skip_lo && row + 31 < a.skip_hi) continue;  // band-limited nodes never read these bins
                    if (Z48)
                        reinterpret_cast<z48 *>(a.Z)[zoff + row] = enc48(re[i], im[i]);
                    else
                        a.Z[zoff + row] = f64x2{re[i], im[i]};
                }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                pa[j] = ca[j];
                pb[j] = cb[j];
                ca[j] = na[j];
                cb[j] = nb[j];
            }
        }
    }
}

// ======================================================================================
//                 ISTFT in double on the 64-lane transform (BF_ISTFT_F64=1, and gsc always)
// ======================================================================================
// The back half of das_f64_pair_kernel as a kernel of its own: one full wavefront per transform (64 lanes x 16 points, fft1024_w64.hpp),
// two frames per complex transform (Ya + i Yb -> real part = frame a, imaginary part = frame b), a run of consecutive frames per
// wavefront with the overlap-add tail carried in registers (one warm-up frame per run), two wavefronts per SIMD.  Register r = 4 g + k3 of
// lane l holds bin l + 64 g + 256 k3 (w64_bin): k3 < 2 reads row[l + 64 g + 256 k3], k3 >= 2 the conjugate of row[(64 - l) + 64 (3 - g) +
// 256 (3 - k3)]: 1 KiB per wave-instruction either way.  util.h:244-253, 301-302.  0.157 ms per 65 536 frames (round 4's half-wavefront
// kernel, one 32 x 32 transform per 32 lanes at one wavefront per SIMD: 0.55; the fp32 istft32_kernel below: 0.148).
// (hi_abs / wave_max_u32: w64_f64_dev.hpp)
constexpr int kIw64Block = 256;
constexpr int kIw64Waves = kIw64Block / 64;
constexpr int kIw64TwD = 2 * (960 + 4 * kTw2RowW64Rot);  // tw1 rows k1 = 1..15 + tw2' (a.tw_w64 + 64), in doubles
constexpr int kIw64WinRow = 18;                          // window as [lane][j] rows of 16 doubles + 2
constexpr int oIwPlane = kIw64TwD;
constexpr int oIwWin = oIwPlane + kIw64Waves * kPlaneD;
constexpr int kIw64Lds = oIwWin + 64 * kIw64WinRow;

// BAND: the rows hold problem 0 and the problems a.yh_lo .. a.yh_hi only (mvdr / lcmv with a band that ends below the Nyquist problems: everything
// else is zero by definition, mvdr.cpp:103, was never written and is never read: no zero-fill of 0.54 GB per 65 536 frames in front of this kernel)
template <bool BAND>
__global__ __launch_bounds__(kIw64Block, 2) void istft_w64_kernel(IstftArgs a, int pairs_per_run, int runs_per_stream) {
    __shared__ __attribute__((aligned(16))) double lds[kIw64Lds];
    const cx<double> *s_tw1 = reinterpret_cast<const cx<double> *>(lds) - 64;  // row k1 starts at 64 (k1 - 1)
    const cx<double> *s_tw2 = reinterpret_cast<const cx<double> *>(lds) + 960;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *plane = lds + oIwPlane + w * kPlaneD;
    double *wcol = plane + w64_col_rot(lane);
    double *row16 = plane + (lane & 15) * kRS + 16 * (lane >> 4);
    const double *wrow = lds + oIwWin + lane * kIw64WinRow;
    {
        const f64x2 *tw2 = a.tw_w64 + 64;
        f64x2 *ltw = reinterpret_cast<f64x2 *>(lds);
        for (int i = tid; i < kIw64TwD / 2; i += kIw64Block) ltw[i] = tw2[i];
        for (int i = tid; i < 1024; i += kIw64Block) lds[oIwWin + (i & 63) * kIw64WinRow + (i >> 6)] = a.win[i];
    }
    __syncthreads();
    const long run = (long)blockIdx.x * kIw64Waves + w;
    int s = (int)(run / runs_per_stream);
    const long r_in_s = run - (long)s * runs_per_stream;
    const bool run_ok = s < a.n_streams;  // wavefront-uniform; no block barrier below
    if (!run_ok) return;
    const long t0 = r_in_s * 2L * pairs_per_run;  // first frame of this run (even)
    if (t0 >= a.n_frames) return;
    long t1 = t0 + 2L * pairs_per_run;
    if (t1 > a.n_frames) t1 = a.n_frames;
    const f64x2 *Ys = a.Yh + (long)s * a.n_frames * kYhStride;
    float *ys = a.y + (long)s * a.n_frames * kHop;

    // Each step takes frames (t, t + 1) through one transform -- or frame t alone: the warm-up frame t0 - 1 (only its second half, the
    // overlap-add tail, is used), the last frame of an odd run, and any pair with a non-finite value in it (a frame the reference turns
    // into NaN / Inf -- mvdr / lcmv: inverse of an all-zero covariance -- would poison its partner through the shared transform).
    float tail[8];  // second half of the previous frame, as float (out_buff[0], util.h:302): register j <- sample 512 + 64 j + lane
    long t = t0;
    if (t0 == 0) {  // stream start: the tail is the carried state
        const float *ti = a.tail_in + (long)s * kHop;
#pragma unroll
        for (int j = 0; j < 8; ++j) tail[j] = ti[(unsigned)(64 * j + lane)];
    } else {
        t = t0 - 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) tail[j] = 0.f;
    }
    // Hermitian extension of a row (register r of lane l <- bin l + 64 g + 256 k3); the irregular bins 0 / 511 / 512 / 513 (quirk Q1):
    // Y[511] and Y[513] averaged with each other's conjugate, Y[0] and Y[512] real, by selects.  Every load is unconditional.
    const int ylo = a.yh_lo, yhi = a.yh_hi;
    auto load_row = [&](const f64x2 *row, cd (&u)[16]) {
        if constexpr (BAND) {  // problems 511 .. 513 are out of band: the plain Hermitian extension, addresses clamped to problem 0, zeros by select
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int g = r >> 2, k3 = r & 3;
                const int qd = k3 < 2 ? lane + 64 * g + 256 * k3 : 64 * (3 - g) + 256 * (3 - k3) + 64 - lane;
                const bool in = qd == 0 || (qd >= ylo && qd <= yhi);
                const cd v = ld(row + (unsigned)(in ? qd : 0));
                u[r] = cd{in ? v.x : 0.0, in ? (k3 < 2 ? v.y : -v.y) : 0.0};
            }
            u[0].y = lane == 0 ? 0.0 : u[0].y;  // bin 0: real part only
            return;
        }
        const cd y513 = ld(row + 513);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int g = r >> 2, k3 = r & 3;
            if (k3 < 2) u[r] = ld(row + (unsigned)(lane + 64 * g + 256 * k3));
            else u[r] = conj(ld(row + (unsigned)(64 * (3 - g) + 256 * (3 - k3) + 64 - lane)));
        }
        u[0].y = lane == 0 ? 0.0 : u[0].y;  // bins 0 and 512: real part only
        const cd m513 = (y513 + u[2]) * 0.5, m511 = (u[13] + conj(y513)) * 0.5;
        u[2].x = lane == 1 ? m513.x : u[2].x;  // bin 513: (Y[513] + conj Y[511]) / 2
        u[2].y = lane == 1 ? m513.y : lane == 0 ? 0.0 : u[2].y;
        u[13].x = lane == 63 ? m511.x : u[13].x;  // bin 511 = 63 + 64 * 3 + 256: (Y[511] + conj Y[513]) / 2
        u[13].y = lane == 63 ? m511.y : u[13].y;
    };
    // register j: sample n = 64 j + lane.  overlap_and_add_prepare_output (util.h:247-252) with the reference's float stores
    auto window = [&](const double (&x)[16], float (&o)[16]) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float f = (float)(x[j] * (1.0 / 1024.0));
            f = (float)((double)f * wrow[j]);
            if (a.use_post_amp) f = (float)((double)f * a.post_amp);  // mvdr.cpp:112-114
            o[j] = f;
        }
    };
    while (t < t1) {
        const bool warm = t < t0;
        bool pair = !warm && t + 1 < t1;
        const f64x2 *ra = Ys + t * kYhStride, *rb = pair ? ra + kYhStride : ra;
        double re[16], im[16];
        {
            cd u[16];
            load_row(ra, u);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                re[r] = u[r].x;
                im[r] = u[r].y;
            }
        }
        BF_STAGE();
        if (pair) {
            cd v[16];
            load_row(rb, v);
            // Two frames share a transform only if neither holds a non-finite value and their largest magnitudes are within 2^20 of each
            // other: the transform's rounding error (1e-16 of the LOUDER frame) lands in both outputs.  A frame of 1e300s beside an
            // ordinary one (gss / lcmv on a rank-deficient covariance), or any frame beside an all-zero one, goes alone.
            unsigned ha = 0, hb = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                ha = max(ha, max(hi_abs(re[r]), hi_abs(im[r])));
                hb = max(hb, max(hi_abs(v[r].x), hi_abs(v[r].y)));
            }
            ha = wave_max_u32(ha);
            hb = wave_max_u32(hb);
            const int ea = (int)(ha >> 20), eb = (int)(hb >> 20);  // biased exponents; 0x7FF: NaN / Inf
            pair = ea < 0x7FF && eb < 0x7FF && ea - eb <= 20 && eb - ea <= 20;
            if (pair) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {  // Ya + i Yb
                    re[r] -= v[r].y;
                    im[r] += v[r].x;
                }
            }
        }
        cx<double> tw[15];
        BF_STAGE();
        load_tw2<1, 16>(tw, s_tw2, lane);
        BF_STAGE();
        w64_inv_p3<double>(re, im);
        w64_T2_any<false>(re, im, row16 - 16 * (lane >> 4), lane >> 4);
        BF_STAGE();
        mul_tw<true, 1, 16>(re, im, tw);
        BF_STAGE();
        load_tw1<1, 16>(tw, s_tw1, lane);
        BF_STAGE();
        fft16_core<double, +1, false>(re, im);
        BF_STAGE();
        T1_inv(re, im, row16, wcol);
        BF_STAGE();
        mul_tw<true, 1, 16>(re, im, tw);
        fft16_core<double, +1, false>(re, im);
        BF_STAGE();
        float oa[16];
        window(re, oa);  // real part: frame t
        if (!warm) {
            float *yo = ys + t * kHop;
#pragma unroll
            for (int j = 0; j < 8; ++j) yo[(unsigned)(64 * j + lane)] = tail[j] + oa[j];
        }
        if (pair) {
            float ob[16];
            window(im, ob);  // imaginary part: frame t + 1
            float *yo = ys + (t + 1) * kHop;
#pragma unroll
            for (int j = 0; j < 8; ++j) yo[(unsigned)(64 * j + lane)] = oa[j + 8] + ob[j];
#pragma unroll
            for (int j = 0; j < 8; ++j) tail[j] = ob[j + 8];
            t += 2;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) tail[j] = oa[j + 8];
            t += 1;
        }
    }
    if (t1 == a.n_frames) {  // the batch ends in this run: out_buff[0] of the next call
        float *to = a.tail_out + (long)s * kHop;
#pragma unroll
        for (int j = 0; j < 8; ++j) to[(unsigned)(64 * j + lane)] = tail[j];
    }
}

// ======================================================================================
//                          ISTFT, single-precision transform
// ======================================================================================
// The inverse path has no error amplification (the output is a float32 sample; the reference rounds Re(ifft)/N to float,
// util.h:249), so the backward FFT runs in fp32: 64 data registers instead of 128, three 256-thread blocks per CU instead
// of one (the fp64 kernel above spends 70 % of its wave cycles in s_waitcnt at one wavefront per SIMD), a third of the
// VALU cycles.  ONE frame per transform: the two-frames-per-IFFT packing of the fp64 kernel would leak a loud frame's
// rounding error (1e-7 of ITS level) into a quiet partner.  Observed against the oracle: <= 2e-7 relative L2 (bound 1e-5).
constexpr int kI32Block = 256;
constexpr int kI32Halves = kI32Block / 32;
constexpr int kPSf = plane_stride<float>::value;  // 36

// (three blocks per CU -- __launch_bounds__(256, 3): 168 registers, 20-32 of them spilled -- 0.148 -> 0.202 ms for phase: round 5)
__global__ __launch_bounds__(kI32Block) void istft32_kernel(IstftArgs a, int frames_per_chunk, int chunks_per_stream) {
    __shared__ __attribute__((aligned(16))) float lds[2048 + kI32Halves * 32 * kPSf];
    const cx<float> *s_tw = reinterpret_cast<const cx<float> *>(lds);
    const int tid = threadIdx.x, lane = tid & 31, hw = tid >> 5;
    float *pbuf = lds + 2048 + hw * 32 * kPSf;
    {
        const float *twf = reinterpret_cast<const float *>(a.tw32);
        for (int i = tid; i < 2048; i += kI32Block) lds[i] = twf[i];
    }
    __syncthreads();
    const double *gwin = a.win + lane;
    const long chunk = (long)blockIdx.x * kI32Halves + hw;
    int s = (int)(chunk / chunks_per_stream);
    const long c_in_s = chunk - (long)s * chunks_per_stream;
    const bool chunk_ok = s < a.n_streams;
    if (!chunk_ok) s = a.n_streams - 1;
    const long t0 = c_in_s * (long)frames_per_chunk;
    long t1 = t0 + frames_per_chunk;
    if (t1 > a.n_frames) t1 = a.n_frames;
    float *ys = a.y + (long)s * a.n_frames * kHop;

    float tail[16];  // second half of the previous frame (out_buff[0], util.h:302)
    if (t0 == 0) {   // stream start: carried state
        const float *ti = a.tail_in + (long)s * kHop + lane;
#pragma unroll
        for (int q = 0; q < 16; ++q) tail[q] = ti[32 * brev5(2 * q)];
    }
    float re[32], im[32];
    for (long t = (t0 == 0 ? 0 : t0 - 1); t < t1; ++t) {  // t0 - 1: warm-up frame, only its second half is used
        // Hermitian extension of the stored row: position i holds bin k = lane + 32*brev5(i); even i are bins < 512, odd i
        // bins >= 512 (conjugate of row[1024 - k]); bins 0 / 511 / 512 / 513 are irregular (quirk Q1)
        // f32x2 rows of the band-limited covariance nodes: only problems 0 and yh_lo..yh_hi exist, the rest is zero
        // (mvdr.cpp:103) and was never written -- groups of 32 bins outside the band cost no load at all
        const f32x2 *row32 = reinterpret_cast<const f32x2 *>(a.Yh) + ((long)s * a.n_frames + t) * kYhStride;
        const int ylo = a.yh_lo, yhi = a.yh_hi;
        auto ldr = [&](int k) -> f32x2 {
            if (k == 0 || (k >= ylo && k <= yhi)) return row32[k];
            return f32x2{0.f, 0.f};
        };
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int kb = 32 * brev5(i);
            const int glo = kb < 512 ? kb : kN - kb - 31;  // rows this position reads: glo .. glo + 31
            f32x2 u{0.f, 0.f};
            if (!(glo > yhi || (glo + 31 < ylo && glo > 0))) u = ldr(kb < 512 ? kb + lane : (kN - kb) - lane);
            if (kb >= 512) u.y = -u.y;
            if (i == 0 && lane == 0) u.y = 0.f;
            if (i == 1) {
                if (lane == 0) {
                    u.y = 0.f;
                } else if (lane == 1 && yhi >= 511) {
                    const f32x2 v = ldr(513);
                    u = f32x2{(v.x + u.x) * 0.5f, (v.y + u.y) * 0.5f};
                }
            }
            if (i == 30 && lane == 31 && yhi >= 511) {
                const f32x2 v = ldr(513);
                u = f32x2{(u.x + v.x) * 0.5f, (u.y - v.y) * 0.5f};
            }
            re[i] = u.x;
            im[i] = u.y;
        }
        fft1024p_inv_A<float>(re, im, lane, s_tw, pbuf);
        __builtin_amdgcn_wave_barrier();
        fft1024p_B<float>(re, lane, pbuf);
        __builtin_amdgcn_wave_barrier();
        fft1024p_C<float, true>(im, lane, pbuf);
        __builtin_amdgcn_wave_barrier();
        fft1024p_D<float, +1>(re, im, lane, pbuf);
        __builtin_amdgcn_wave_barrier();
        // position i: sample n = 32*brev5(i) + lane (even i: first half, odd i: second half).
        // overlap_and_add_prepare_output (util.h:247-252) with the reference's float stores.
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const double h = gwin[32 * brev5(i)];
            float f = re[i] * (1.0f / 1024.0f);
            f = (float)((double)f * h);
            if (a.use_post_amp) f = (float)((double)f * a.post_amp);  // mvdr.cpp:112-114
            re[i] = f;
        }
        if (chunk_ok && t >= t0) {
            float *yo = ys + t * kHop + lane;
#pragma unroll
            for (int q = 0; q < 16; ++q) yo[32 * brev5(2 * q)] = tail[q] + re[2 * q];
            if (t == a.n_frames - 1) {
                float *to = a.tail_out + (long)s * kHop + lane;
#pragma unroll
                for (int q = 0; q < 16; ++q) to[32 * brev5(2 * q)] = re[2 * q + 1];
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) tail[q] = re[2 * q + 1];
    }
}

#else

// ======================================================================================
//          generic sizes (N = 128 ... 8192 except 1024): LDS-staged transforms
// ======================================================================================
// JACK periods other than 512 frames (rosjack.cpp:131-134 takes whatever the server reports; fft_win = 2 * period, util.h:261).
// One 256-thread block per transform, the whole signal in LDS: two N-point complex double buffers and radix-4 autosort (Stockham)
// passes with a barrier each up to N = 4096; at N = 8192 one buffer fills the LDS (128 KB), so that size runs radix-2 passes
// in place on bit-reversed input.  Twiddles exp(-2 pi i m / N) from a double-precision table built on the host (in LDS up to
// N = 2048, through L1 above).  Correctness first: these sizes are not the benchmark shape; the 1024-point path keeps its
// in-register kernels.
constexpr int kGenBlock = 256;
constexpr bool kGenInPlace = kN >= 8192;
constexpr bool kGenTwLds = kN <= 2048;  // 4 KB at N = 512, 16 KB at N = 2048: measured better than L1 there (5.52 vs 5.68 ms per mvdr batch)
constexpr int gen_log2(int n) { return n <= 1 ? 0 : 1 + gen_log2(n >> 1); }
// where sample / bin n goes before the transform: its own slot (autosort) or the bit-reversed one (in-place passes)
__device__ __forceinline__ int gen_slot(int n) { return kGenInPlace ? (int)(__brev((unsigned)n) >> (32 - gen_log2(kN))) : n; }
// radix-2 decimation-in-time passes in place: input at its bit-reversed slot, result in natural order
template <int DIR>
__device__ __forceinline__ void fft_inplace(cd *buf, const f64x2 *tw, int tid) {
    for (int half = 1; half < kN; half <<= 1) {
        for (int b = tid; b < kN / 2; b += kGenBlock) {
            const int k = b & (half - 1);
            const int j = ((b - k) << 1) + k;
            const f64x2 w = tw[k * (kN / (2 * half))];  // W^(k N / (2 half)): index < N / 2
            const cd u = buf[j], v = buf[j + half] * cd{w.x, DIR < 0 ? w.y : -w.y};
            buf[j] = u + v;
            buf[j + half] = u - v;
        }
        __syncthreads();
    }
}

// in-place-looking wrapper: data starts in buf0, result ends in the returned buffer.  DIR = -1 forward, +1 backward.
// Radix-4 autosort passes and one closing radix-2 pass (N = 2 * 4^k: four + one at 512, five + one at 2048; the first version
// ran nine / eleven radix-2 passes with the twiddles read from global memory), tw = the half-length table in LDS,
// W^(m + N/2) = -W^m.
template <int DIR>
__device__ __forceinline__ cd *stockham(cd *buf0, cd *buf1, const f64x2 *c4, const f64x2 *tw, int tid) {
    cd *in = buf0, *out = buf1;
    auto dirw = [](f64x2 w) { return cd{w.x, DIR < 0 ? w.y : -w.y}; };  // W (forward) or its conjugate (backward)
    int ns = 1;
    for (; ns * 4 <= kN; ns <<= 2) {
        const f64x2 *cp = c4 + (ns - 1);  // this pass' block: the three twiddles of butterfly k contiguous in k (geometry.hpp stockham_twiddles)
        for (int j = tid; j < kN / 4; j += kGenBlock) {
            const int k = j & (ns - 1);
            const cd a0 = in[j];
            const cd a1 = in[j + kN / 4] * dirw(cp[k]);
            const cd a2 = in[j + kN / 2] * dirw(cp[ns + k]);
            const cd a3 = in[j + 3 * kN / 4] * dirw(cp[2 * ns + k]);
            const cd s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
            const cd r13 = DIR < 0 ? cd{d13.y, -d13.x} : cd{-d13.y, d13.x};  // -i d13 (forward) / +i d13 (backward)
            const int j0 = ((j - k) << 2) + k;
            out[j0] = s02 + s13;
            out[j0 + ns] = d02 + r13;
            out[j0 + 2 * ns] = s02 - s13;
            out[j0 + 3 * ns] = d02 - r13;
        }
        __syncthreads();
        cd *t = in; in = out; out = t;
    }
    if (ns < kN) {  // closing radix-2 pass (N = 2 * 4^k): ns = N / 2, twiddle W^k in order
        for (int j = tid; j < kN / 2; j += kGenBlock) {
            const cd a = in[j], b = in[j + kN / 2] * dirw(tw[j]);
            out[j] = a + b;
            out[j + kN / 2] = a - b;
        }
        __syncthreads();
        cd *t = in; in = out; out = t;
    }
    return in;
}

template <int LAYOUT>
__global__ __launch_bounds__(kGenBlock) void stft_generic_kernel(StftArgs a) {
    __shared__ cd s_a[kN], s_b[kGenInPlace ? 1 : kN];
    constexpr bool kTwLds = kGenTwLds;
    constexpr int kR4 = stockham_r4_entries(kN);
    __shared__ f64x2 s_twl[kTwLds ? kR4 : 1];  // the radix-4 passes' blocks; the closing radix-2 pass reads W^m in order through L1
    const int tid = threadIdx.x;
    if (kTwLds) {
        for (int i = tid; i < kR4; i += kGenBlock) s_twl[i] = a.tw[kN / 2 + i];
        __syncthreads();
    }
    const f64x2 *s_c4 = kTwLds ? s_twl : a.tw + kN / 2, *s_tw = a.tw;
    const int M = a.n_mics, MF = a.n_fft_mics, NP = (MF + 1) >> 1;
    const long total = (long)a.n_streams * a.n_frames * NP;
    for (long item = blockIdx.x; item < total; item += gridDim.x) {
        const int p = (int)(item % NP);
        const long st = item / NP;
        const long t = st % a.n_frames;
        const int s = (int)(st / a.n_frames);
        const float *xs = a.x + (long)s * a.stream_stride_x;
        const float *hs = a.hist + (long)s * M * kHop;
        const int ma = 2 * p;
        const bool b_ok = 2 * p + 1 < MF;
        const int mb = b_ok ? 2 * p + 1 : ma;
        for (int n = tid; n < kN; n += kGenBlock) {
            const bool first = n < kHop;          // first half of the frame = the hop before hop t
            const int i = first ? n : n - kHop;
            float va, vb;
            if (LAYOUT == 0) {
                const float *ba = first ? (t >= 1 ? xs + (long)ma * a.mic_stride + (t - 1) * kHop : hs + ma * kHop)
                                        : xs + (long)ma * a.mic_stride + t * kHop;
                const float *bb = first ? (t >= 1 ? xs + (long)mb * a.mic_stride + (t - 1) * kHop : hs + mb * kHop)
                                        : xs + (long)mb * a.mic_stride + t * kHop;
                va = ba[i];
                vb = bb[i];
            } else {
                const float *bs = first ? (t >= 1 ? xs + (t - 1) * (long)kHop * M : hs) : xs + t * (long)kHop * M;
                va = bs[(long)i * M + ma];
                vb = bs[(long)i * M + mb];
            }
            const double h = a.win[n] * (a.halve ? 0.5 : 1.0);  // mvdr / lcmv spectra are stored halved (exact)
            s_a[gen_slot(n)] = cd{(double)va * h, b_ok ? (double)vb * h : 0.0};   // buf[j]*hann_win[i]  (util.h:235)
        }
        __syncthreads();
        const cd *res = s_a;
        if (kGenInPlace)
            fft_inplace<-1>(s_a, s_tw, tid);
        else
            res = stockham<-1>(s_a, s_b, s_c4, s_tw, tid);
        const long zoff = (((long)s * a.frames_ws + a.frame_off + t) * NP + p) * kN;
        for (int k = tid; k < kN; k += kGenBlock) {
            if (a.z48)
                reinterpret_cast<z48 *>(a.Z)[zoff + k] = enc48(res[k].x, res[k].y);
            else
                a.Z[zoff + k] = f64x2{res[k].x, res[k].y};
        }
        __syncthreads();
    }
}

#if BF_NFFT == 128 || BF_NFFT == 256 || BF_NFFT == 512
// ---- N = 512 / 256 / 128 in registers: 32 points per lane x N / 32 lanes, 1024 / N frames per half-wavefront ---------------------------
// The generic kernel above costs 2.1 ms per 131 072 frames of 8 microphones at N = 512 (the N = 1024 kernel: 0.69 ms for the same samples).
// N = 32 x NL: lane (g, n2) of a half-wavefront -- g = lane / NL the frame, n2 = lane mod NL -- holds x_g[NL j + n2] in register j.  The first
// pass is fft1024.hpp's (32-point DIF over j, twiddle W_N^(n2 k1), plane transpose: the G = 32 / NL frames of the half-wavefront travel through
// the same 32 x 32 plane side by side); after it lane k1 holds, for every frame g, the NL values n2 = 0..NL-1 in registers g NL .. g NL + NL - 1,
// and the second pass is one NL-point DIF per frame: position i' of frame g = bin k1 + 32 brev(i').  A store instruction writes 32 consecutive
// bins of one frame.  The lanes of a frame sit side by side, so a load instruction touches G runs of NL consecutive samples.
constexpr int kNL = kN / 32, kG = 32 / kNL, kLogNL = gen_log2(kNL);
template <int LAYOUT, bool Z48>
__global__ __launch_bounds__(256) void stft_small_kernel(StftArgs a) {
    constexpr int kBlock = 256, kHalves = kBlock / 32, kWinRow = 34;
    // the twiddle row k1 = 0 (all ones) is not stored: at N = 512 the block then takes exactly half of the CU's LDS and two blocks (two wavefronts per
    // SIMD) are resident
    __shared__ __attribute__((aligned(16))) double lds[2 * 31 * kNL + kHalves * 32 * kPSd + kNL * kWinRow];
    cx<double> *s_tw = reinterpret_cast<cx<double> *>(lds);  // [k1 - 1][n2] = W_N^(k1 n2), k1 = 1..31
    const int tid = threadIdx.x, lane = tid & 31, hw = tid >> 5;
    double *pbuf = lds + 2 * 31 * kNL + hw * 32 * kPSd;
    double *s_win = lds + 2 * 31 * kNL + kHalves * 32 * kPSd;  // [n2][j] = win[NL j + n2]
    {
        for (int i = tid; i < 31 * kNL; i += kBlock) {
            const int m = ((i / kNL + 1) * (i % kNL)) % kN;  // a.tw[m] = exp(-2 pi i m / N) for m < N / 2; W^(m + N/2) = -W^m
            const f64x2 w = a.tw[m % (kN / 2)];
            s_tw[i] = m < kN / 2 ? cx<double>{w.x, w.y} : cx<double>{-w.x, -w.y};
        }
        for (int i = tid; i < kN; i += kBlock) s_win[(i % kNL) * kWinRow + i / kNL] = a.win[i] * (a.halve ? 0.5 : 1.0);  // mvdr / lcmv spectra are stored halved (exact): unpacking a pair is then Z[k] +- conj Z[N-k] without the 1/2
        __syncthreads();
    }
    const int g = lane / kNL, n2 = lane % kNL;
    const f64x2 *wrow = reinterpret_cast<const f64x2 *>(s_win + n2 * kWinRow);
    const int M = a.n_mics, MF = a.n_fft_mics, NP = (MF + 1) >> 1, L = a.run_len;  // L: a multiple of G
    const long runs = (a.n_frames + L - 1) / L;
    const long total = (long)a.n_streams * runs * NP;
    const long stride = (long)gridDim.x * kHalves;
    for (long item = (long)blockIdx.x * kHalves + hw; item < total; item += stride) {  // no block barrier below
        const int p = (int)(item % NP);
        const long sr = item / NP;
        const long run = sr % runs;
        const int s = (int)(sr / runs);
        const float *xs = a.x + (long)s * a.stream_stride_x;
        const float *hs = a.hist + (long)s * M * kHop;
        const int ma = 2 * p;
        const bool b_ok = 2 * p + 1 < MF;
        const int mb = b_ok ? 2 * p + 1 : ma;
        const double bs = b_ok ? 1.0 : 0.0;
        long te = (run + 1) * L;
        if (te > a.n_frames) te = a.n_frames;
        for (long t = run * L; t < te; t += kG) {
            long f = t + g;  // this lane's frame; past the end: the last frame again, never stored
            if (f >= a.n_frames) f = a.n_frames - 1;
            double re[32], im[32];
            {
                float va[32], vb[32];  // hop f - 1 (f = 0: the carried hop) and hop f as raw samples, register j <- sample NL j + n2
                if (LAYOUT == 0) {
                    const float *pa = (f >= 1 ? xs + (long)ma * a.mic_stride + (f - 1) * kHop : hs + ma * kHop) + n2;
                    const float *pb = (f >= 1 ? xs + (long)mb * a.mic_stride + (f - 1) * kHop : hs + mb * kHop) + n2;
                    const float *ca = xs + (long)ma * a.mic_stride + f * kHop + n2, *cb = xs + (long)mb * a.mic_stride + f * kHop + n2;
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        va[j] = pa[kNL * j];
                        vb[j] = pb[kNL * j];
                        va[j + 16] = ca[kNL * j];
                        vb[j + 16] = cb[kNL * j];
                    }
                } else {
                    const float *ps = (f >= 1 ? xs + (f - 1) * (long)kHop * M : hs) + (long)n2 * M, *cs = xs + f * (long)kHop * M + (long)n2 * M;
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        va[j] = ps[(long)kNL * j * M + ma];
                        vb[j] = ps[(long)kNL * j * M + mb];
                        va[j + 16] = cs[(long)kNL * j * M + ma];
                        vb[j + 16] = cs[(long)kNL * j * M + mb];
                    }
                }
#pragma unroll
                for (int j = 0; j < 32; j += 2) {
                    const f64x2 w = wrow[j >> 1];  // win[NL j + n2], win[NL (j + 1) + n2]
                    re[j] = (double)va[j] * w.x;   // buf[j]*hann_win[i]  (util.h:235)
                    im[j] = (double)vb[j] * (w.x * bs);
                    re[j + 1] = (double)va[j + 1] * w.y;
                    im[j + 1] = (double)vb[j + 1] * (w.y * bs);
                }
            }
            fft32_dif<double, -1>(re, im);
#pragma unroll
            for (int i = 1; i < 32; ++i) {
                const cx<double> w = s_tw[(brev5(i) - 1) * kNL + n2];
                const double xr = re[i], xi = im[i];
                re[i] = xr * w.x - xi * w.y;
                im[i] = xr * w.y + xi * w.x;
            }
#pragma unroll
            for (int i = 0; i < 32; ++i) pbuf[brev5(i) * kPSd + lane] = re[i];
            __builtin_amdgcn_wave_barrier();
            fft1024p_B<double>(re, lane, pbuf);
            __builtin_amdgcn_wave_barrier();
            fft1024p_C<double, false>(im, lane, pbuf);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int cc = 0; cc < 32; ++cc) im[cc] = pbuf[lane * kPSd + cc];
            __builtin_amdgcn_wave_barrier();
            fftn_dif_all<double, -1, kLogNL>(re, im);
            // lane = k1; register g' NL + i' = bin k1 + 32 brev(i') of frame t + g'
#pragma unroll
            for (int gg = 0; gg < kG; ++gg) {
                if (t + gg >= te) continue;
                const long zoff = (((long)s * a.frames_ws + a.frame_off + t + gg) * NP + p) * kN + lane;
#pragma unroll
                for (int i = 0; i < kNL; ++i) {
                    const int row = 32 * brevn(i, kLogNL);  // bins row .. row + 31 of this store
                    if (row > a.skip_lo && row + 31 < a.skip_hi) continue;  // band-limited nodes never read these bins
                    if (Z48)
                        reinterpret_cast<z48 *>(a.Z)[zoff + row] = enc48(re[gg * kNL + i], im[gg * kNL + i]);
                    else
                        a.Z[zoff + row] = f64x2{re[gg * kNL + i], im[gg * kNL + i]};
                }
            }
        }
    }
}
#endif


// Hermitian part of y_fft at bin k (0..N-1) from the per-bin kernels' output row.
__device__ __forceinline__ cd herm_gen(const f64x2 *row, int k) {
    if (k == 0 || k == kN / 2) return cd{row[k].x, 0.0};
    if (k == kN / 2 - 1) return (ld(row + k) + conj(ld(row + kQX))) * 0.5;
    if (k == kQX) return (ld(row + kQX) + conj(ld(row + kN / 2 - 1))) * 0.5;
    if (k < kN / 2) return ld(row + k);
    return conj(ld(row + (kN - k)));
}

// one frame per block: backward transform + synthesis window, windowed frame to a.frames (float, reference rounding)
__global__ __launch_bounds__(kGenBlock) void istft_generic_kernel(IstftArgs a) {
    __shared__ cd s_a[kN], s_b[kGenInPlace ? 1 : kN];
    constexpr bool kTwLds = kGenTwLds;
    constexpr int kR4 = stockham_r4_entries(kN);
    __shared__ f64x2 s_twl[kTwLds ? kR4 : 1];  // the radix-4 passes' blocks; the closing radix-2 pass reads W^m in order through L1
    const int tid = threadIdx.x;
    if (kTwLds) {
        for (int i = tid; i < kR4; i += kGenBlock) s_twl[i] = a.tw[kN / 2 + i];
        __syncthreads();
    }
    const f64x2 *s_c4 = kTwLds ? s_twl : a.tw + kN / 2, *s_tw = a.tw;
    const long total = (long)a.n_streams * a.n_frames;
    for (long f = blockIdx.x; f < total; f += gridDim.x) {
        const f64x2 *row = a.Yh + f * kYhStride;
        for (int k = tid; k < kN; k += kGenBlock) s_a[gen_slot(k)] = herm_gen(row, k);
        __syncthreads();
        const cd *res = s_a;
        if (kGenInPlace)
            fft_inplace<+1>(s_a, s_tw, tid);
        else
            res = stockham<+1>(s_a, s_b, s_c4, s_tw, tid);
        float *fo = a.frames + f * kN;
        for (int n = tid; n < kN; n += kGenBlock) {
            float v = (float)(res[n].x / (double)kN);            // util.h:249
            v = (float)((double)v * a.win[n]);                    // util.h:250
            if (a.use_post_amp) v = (float)((double)v * a.post_amp);  // mvdr.cpp:112-114
            fo[n] = v;
        }
        __syncthreads();
    }
}

#if BF_NFFT == 128 || BF_NFFT == 256 || BF_NFFT == 512
// ---- backward side of stft_small_kernel: Hermitian extension, N-point backward transform, window, overlap-add in ONE kernel -----------------
// y[NL n1 + n2] = sum_k1 W32^(-n1 k1) [ W_N^(-n2 k1) sum_k2 Y[k1 + 32 k2] W_NL^(-n2 k2) ]: an NL-point DIT per frame in the registers of lane k1
// (positions g NL + brev(k2) in, g NL + n2 out), the conjugate twiddle, the plane transpose, and fft32_dif<+1> over k1 in lane (g, n2).  Position i
// of lane (g, n2) then holds sample NL brev5(i) + n2 of frame g: even positions the first half, odd positions the second.  The overlap-add partner
// (second half of the frame before) sits NL lanes to the left -- or, for g = 0, in the last NL lanes of the previous iteration: both through a
// small LDS buffer.  A run that does not start the stream recomputes the group in front of it.  Same float roundings as istft_generic_kernel +
// ola_generic_kernel (util.h:249-250, 301-302), which wrote every windowed frame to HBM and added the halves in a second kernel: 0.46 ms per
// 131 072 frames at N = 512.
__global__ __launch_bounds__(256) void istft_small_kernel(IstftArgs a, int L) {
    constexpr int kBlock = 256, kHalves = kBlock / 32, kWinRow = 34, kTS = 17;
    __shared__ __attribute__((aligned(16))) double lds[2 * 32 * kNL + kHalves * 32 * kPSd + kNL * kWinRow + kHalves * 32 * kTS / 2 + 8];
    cx<double> *s_tw = reinterpret_cast<cx<double> *>(lds);  // [n2][k1] = W_N^(k1 n2)
    const int tid = threadIdx.x, lane = tid & 31, hw = tid >> 5;
    double *pbuf = lds + 2 * 32 * kNL + hw * 32 * kPSd;
    double *s_win = lds + 2 * 32 * kNL + kHalves * 32 * kPSd;  // [n2][j] = win[NL j + n2]
    float *tb = reinterpret_cast<float *>(lds + 2 * 32 * kNL + kHalves * 32 * kPSd + kNL * kWinRow) + hw * 32 * kTS;  // second halves: [lane][q]
    {
        for (int i = tid; i < 32 * kNL; i += kBlock) {
            const int m = ((i / 32) * (i % 32)) % kN;  // a.tw[m] = exp(-2 pi i m / N) for m < N / 2; W^(m + N/2) = -W^m
            const f64x2 w = a.tw[m % (kN / 2)];
            s_tw[i] = m < kN / 2 ? cx<double>{w.x, w.y} : cx<double>{-w.x, -w.y};
        }
        for (int i = tid; i < kN; i += kBlock) s_win[(i % kNL) * kWinRow + i / kNL] = a.win[i];
        __syncthreads();
    }
    const int g = lane / kNL, n2 = lane % kNL;
    const long runs = (a.n_frames + L - 1) / L;
    const long total = (long)a.n_streams * runs;
    const long stride = (long)gridDim.x * kHalves;
    for (long item = (long)blockIdx.x * kHalves + hw; item < total; item += stride) {  // no block barrier below
        const int s = (int)(item / runs);
        const long t0 = (item - (long)s * runs) * L;
        long te = t0 + L;
        if (te > a.n_frames) te = a.n_frames;
        float *ys = a.y + (long)s * a.n_frames * kHop;
        if (t0 == 0) {  // stream start: the carried second half (out_buff[0], util.h:302) stands in for "the last frame of the group before"
            if (g == kG - 1) {
#pragma unroll
                for (int q = 0; q < 16; ++q) tb[lane * kTS + q] = a.tail_in[(long)s * kHop + kNL * brev5(2 * q) + n2];
            }
            __builtin_amdgcn_wave_barrier();
        }
        for (long t = (t0 == 0 ? 0 : t0 - kG); t < te; t += kG) {  // t0 - G: warm-up group, only its last frame's second half is used
            double re[32], im[32];
            // lane = k1: position gg NL + i' <- bin k1 + 32 brev(i') of frame t + gg (past the end: the last frame again, never stored)
#pragma unroll
            for (int gg = 0; gg < kG; ++gg) {
                long fr = t + gg;
                if (fr >= a.n_frames) fr = a.n_frames - 1;
                const f64x2 *row = a.Yh + ((long)s * a.n_frames + fr) * kYhStride;
#pragma unroll
                for (int i = 0; i < kNL; ++i) {
                    const cd v = herm_gen(row, lane + 32 * brevn(i, kLogNL));
                    re[gg * kNL + i] = v.x;
                    im[gg * kNL + i] = v.y;
                }
            }
            fftn_dit_all<double, +1, kLogNL>(re, im);
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                if (r % kNL == 0) continue;  // n2 = 0: no twiddle
                const cx<double> w = s_tw[(r % kNL) * 32 + lane];  // conj applied
                const double xr = re[r], xi = im[r];
                re[r] = xr * w.x + xi * w.y;
                im[r] = xi * w.x - xr * w.y;
            }
#pragma unroll
            for (int r = 0; r < 32; ++r) pbuf[r * kPSd + lane] = re[r];
            __builtin_amdgcn_wave_barrier();
            fft1024p_B<double>(re, lane, pbuf);
            __builtin_amdgcn_wave_barrier();
            fft1024p_C<double, true>(im, lane, pbuf);
            __builtin_amdgcn_wave_barrier();
            fft1024p_D<double, +1>(re, im, lane, pbuf);
            __builtin_amdgcn_wave_barrier();
            // lane = (g, n2): position i = sample NL brev5(i) + n2 of frame t + g
            const double *wr = s_win + n2 * kWinRow;
            float first[16], tail[16];
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                float v = (float)(re[i] / (double)kN);                        // util.h:249
                v = (float)((double)v * wr[brev5(i)]);                        // util.h:250
                if (a.use_post_amp) v = (float)((double)v * a.post_amp);      // mvdr.cpp:112-114
                if (i & 1) tail[i >> 1] = v; else first[i >> 1] = v;
            }
            float prev0[16];  // g = 0: the last frame of the group before, parked by the previous iteration
#pragma unroll
            for (int q = 0; q < 16; ++q) prev0[q] = tb[((lane + 32 - kNL) & 31) * kTS + q];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 16; ++q) tb[lane * kTS + q] = tail[q];
            __builtin_amdgcn_wave_barrier();
            const long f = t + g;
            const bool st_ok = t >= t0 && f < te;
            float *yo = ys + f * kHop + n2;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float left = tb[((lane + 32 - kNL) & 31) * kTS + q];  // g >= 1: the frame before, NL lanes to the left
                const float partner = g == 0 ? prev0[q] : left;
                if (st_ok) yo[kNL * brev5(2 * q)] = partner + first[q];  // out = prev[H + n] + cur[n]  (util.h:301-302)
            }
            if (f == a.n_frames - 1 && t >= t0) {  // carried state for the next call
#pragma unroll
                for (int q = 0; q < 16; ++q) a.tail_out[(long)s * kHop + kNL * brev5(2 * q) + n2] = tail[q];
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}
#endif

#if BF_NFFT == 2048
// ---- N = 2048 as ONE transform per FULL wavefront: 32 registers x 64 lanes (das_fused.hip das_fused_wave2048_kernel has the scheme) -----------
// First pass over the registers (n = 64 j + lane64), twiddle W2048^(k1 lane64), a 32 x 64 plane transpose through LDS (row = register position,
// 64 columns; lane l of half h reads row l, columns 32 h .. 32 h + 31), one radix-2 stage between lane l of half 0 and of half 1
// (v_permlane32_swap_b32 on the two dwords of a double), a second 32-point pass: position i of lane64 = bin lane64 + 64 brev5(i), a store writes 64
// consecutive bins.  No even / odd passes, no second read of the samples, 128 data registers.  Round 4's stft_split_kernel (two FFT-1024 and a radix-2
// step, the first transform waiting in the accumulator registers): 1.27 ms z48 / 1.46 ms c128 per 32 768 frames of 8 microphones.
__device__ __forceinline__ void halves_pair_d(double v, double &lo, double &hi) {  // the value of v in lane l of half 0 / of half 1
    unsigned a0 = (unsigned)__double2loint(v), a1 = (unsigned)__double2hiint(v), b0 = a0, b1 = a1;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
    lo = __hiloint2double((int)a1, (int)a0);
    hi = __hiloint2double((int)b1, (int)b0);
}

template <int LAYOUT, bool Z48>
__global__ __launch_bounds__(256) void stft_wave2048_kernel(StftArgs a) {
    constexpr int kBlock = 256, kWaves = kBlock / 64, kRS = 66, kWinRow = 34;  // 66-double rows: 16-lane ds_read_b128 groups on distinct banks
    __shared__ __attribute__((aligned(16))) double lds[2 * 32 * 64 + 64 + kWaves * 32 * kRS + 64 * kWinRow];
    cx<double> *s_tw = reinterpret_cast<cx<double> *>(lds);              // [k1][lane64] = W2048^(k1 lane64)
    cx<double> *s_w64 = reinterpret_cast<cx<double> *>(lds + 2 * 32 * 64);  // [c] = W64^c
    const int tid = threadIdx.x, lane64 = tid & 63, l = tid & 31, h = (tid >> 5) & 1;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *pl = lds + 2 * 32 * 64 + 64 + w * 32 * kRS;
    double *s_win = lds + 2 * 32 * 64 + 64 + kWaves * 32 * kRS;  // [lane64][j] = win[64 j + lane64]
    {
        for (int i = tid; i < 32 * 64; i += kBlock) {
            const int m = ((i >> 6) * (i & 63)) % kN;  // a.tw[m] = W2048^m for m < 1024, W^(m + 1024) = -W^m
            const f64x2 t = a.tw[m % 1024];
            s_tw[i] = m < 1024 ? cx<double>{t.x, t.y} : cx<double>{-t.x, -t.y};
        }
        if (tid < 32) {
            const f64x2 t = a.tw[32 * tid];  // W64^c = W2048^(32 c)
            s_w64[tid] = cx<double>{t.x, t.y};
        }
        for (int i = tid; i < kN; i += kBlock) s_win[(i & 63) * kWinRow + (i >> 6)] = a.win[i] * (a.halve ? 0.5 : 1.0);  // mvdr / lcmv spectra are stored halved (exact): unpacking a pair is then Z[k] +- conj Z[N-k] without the 1/2
        __syncthreads();
    }
    const f64x2 *wrow = reinterpret_cast<const f64x2 *>(s_win + lane64 * kWinRow);
    const int M = a.n_mics, MF = a.n_fft_mics, NP = (MF + 1) >> 1, L = a.run_len;
    const long runs = (a.n_frames + L - 1) / L;
    const long total = (long)a.n_streams * runs * NP;
    const long stride = (long)gridDim.x * kWaves;
    for (long item = (long)blockIdx.x * kWaves + w; item < total; item += stride) {  // no block barrier below
        const int p = (int)(item % NP);
        const long sr = item / NP;
        const long run = sr % runs;
        const int s = (int)(sr / runs);
        const float *xs = a.x + (long)s * a.stream_stride_x;
        const float *hs = a.hist + (long)s * M * kHop;
        const int ma = 2 * p;
        const bool b_ok = 2 * p + 1 < MF;
        const int mb = b_ok ? 2 * p + 1 : ma;
        const double bs = b_ok ? 1.0 : 0.0;
        long te = (run + 1) * L;
        if (te > a.n_frames) te = a.n_frames;
        for (long t = run * L; t < te; ++t) {
            double re[32], im[32];
            if (LAYOUT == 0) {
                const float *pa = (t >= 1 ? xs + (long)ma * a.mic_stride + (t - 1) * kHop : hs + ma * kHop) + lane64;
                const float *pb = (t >= 1 ? xs + (long)mb * a.mic_stride + (t - 1) * kHop : hs + mb * kHop) + lane64;
                const float *ca = xs + (long)ma * a.mic_stride + t * kHop + lane64, *cb = xs + (long)mb * a.mic_stride + t * kHop + lane64;
#pragma unroll
                for (int j = 0; j < 16; j += 2) {
                    const f64x2 w1 = wrow[j >> 1], w2 = wrow[8 + (j >> 1)];  // win[64 j + lane64], win[64 (j + 1) + lane64] / the same for j + 16
                    re[j] = (double)pa[64 * j] * w1.x;  // buf[j]*hann_win[i]  (util.h:235)
                    im[j] = (double)pb[64 * j] * (w1.x * bs);
                    re[j + 1] = (double)pa[64 * (j + 1)] * w1.y;
                    im[j + 1] = (double)pb[64 * (j + 1)] * (w1.y * bs);
                    re[j + 16] = (double)ca[64 * j] * w2.x;
                    im[j + 16] = (double)cb[64 * j] * (w2.x * bs);
                    re[j + 17] = (double)ca[64 * (j + 1)] * w2.y;
                    im[j + 17] = (double)cb[64 * (j + 1)] * (w2.y * bs);
                }
            } else {
                const float *ps = (t >= 1 ? xs + (t - 1) * (long)kHop * M : hs) + (long)lane64 * M, *cs = xs + t * (long)kHop * M + (long)lane64 * M;
#pragma unroll
                for (int j = 0; j < 16; j += 2) {
                    const f64x2 w1 = wrow[j >> 1], w2 = wrow[8 + (j >> 1)];
                    re[j] = (double)ps[(long)64 * j * M + ma] * w1.x;
                    im[j] = (double)ps[(long)64 * j * M + mb] * (w1.x * bs);
                    re[j + 1] = (double)ps[(long)64 * (j + 1) * M + ma] * w1.y;
                    im[j + 1] = (double)ps[(long)64 * (j + 1) * M + mb] * (w1.y * bs);
                    re[j + 16] = (double)cs[(long)64 * j * M + ma] * w2.x;
                    im[j + 16] = (double)cs[(long)64 * j * M + mb] * (w2.x * bs);
                    re[j + 17] = (double)cs[(long)64 * (j + 1) * M + ma] * w2.y;
                    im[j + 17] = (double)cs[(long)64 * (j + 1) * M + mb] * (w2.y * bs);
                }
            }
            fft32_dif<double, -1>(re, im);
#pragma unroll
            for (int i = 1; i < 32; ++i) {
                const cx<double> tw = s_tw[brev5(i) * 64 + lane64];
                const double xr = re[i], xi = im[i];
                re[i] = xr * tw.x - xi * tw.y;
                im[i] = xr * tw.y + xi * tw.x;
            }
            const double *rowp = pl + l * kRS + 32 * h;
#pragma unroll
            for (int i = 0; i < 32; ++i) pl[brev5(i) * kRS + lane64] = re[i];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < 32; ++c) re[c] = rowp[c];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 32; ++i) pl[brev5(i) * kRS + lane64] = im[i];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < 32; ++c) im[c] = rowp[c];
            __builtin_amdgcn_wave_barrier();
            // lane (k1 = l, h): register c = n2 = 32 h + c.  Radix-2 DIF stage over h: half 0 keeps a + b, half 1 keeps (a - b) W64^c
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                double ar, br, ai, bi;
                halves_pair_d(re[c], ar, br);
                halves_pair_d(im[c], ai, bi);
                const cx<double> tw = s_w64[c];
                const double dr = ar - br, di = ai - bi;
                re[c] = h ? dr * tw.x - di * tw.y : ar + br;
                im[c] = h ? dr * tw.y + di * tw.x : ai + bi;
            }
            fft32_dif<double, -1>(re, im);
            const long zoff = (((long)s * a.frames_ws + a.frame_off + t) * NP + p) * kN + lane64;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int row = 64 * brev5(i);  // bins row .. row + 63 of this store
                if (row > a.skip_lo && row + 63 < a.skip_hi) continue;  // band-limited nodes never read these bins
                if (Z48)
                    reinterpret_cast<z48 *>(a.Z)[zoff + row] = enc48(re[i], im[i]);
                else
                    a.Z[zoff + row] = f64x2{re[i], im[i]};
            }
        }
    }
}
#endif

#if BF_NFFT == 2048
// ---- backward side at N = 2048: ONE FFT-1024 per frame ------------------------------------------------------------------------------
// y real: its even samples have the spectrum A[k] = Y[k] + Y[k + 1024], its odd samples B[k] = (Y[k] - Y[k + 1024]) conj(W2048^k) (one radix-2
// decimation-in-frequency step of the backward transform), both Hermitian over 1024 bins -- so IFFT1024(A + i B) = y_even + i y_odd: one complex
// transform on fft1024.hpp's machinery returns the whole frame, sample pair (2 m, 2 m + 1) in (re, im) of one register.  Window, the reference's
// float roundings (util.h:249-250), overlap-add against the previous frame's second half carried in registers, 8-byte stores.  A half-wavefront
// per (stream, run of frames); a run that does not start the stream recomputes the frame in front of it.  istft_generic_kernel + ola_generic_kernel:
// 0.77 ms per 32 768 frames.
__global__ __launch_bounds__(256) void istft_split_kernel(IstftArgs a, int L) {
    constexpr int kBlock = 256, kHalves = kBlock / 32, kWinRow = 66;
    __shared__ __attribute__((aligned(16))) double lds[2048 + 2048 + kHalves * 32 * kPSd + 32 * kWinRow];
    cx<double> *s_tw = reinterpret_cast<cx<double> *>(lds);         // [n2][k1] = W1024^(k1 n2) (symmetric)
    cx<double> *s_w2 = reinterpret_cast<cx<double> *>(lds + 2048);  // [k] = W2048^k, k < 1024
    const int tid = threadIdx.x, lane = tid & 31, hw = tid >> 5;
    double *pbuf = lds + 4096 + hw * 32 * kPSd;
    double *s_win = lds + 4096 + kHalves * 32 * kPSd;  // [lane][2 j + q] = win[2 (32 j + lane) + q], j < 32
    {
        for (int i = tid; i < 1024; i += kBlock) {
            const int m = (2 * (i >> 5) * (i & 31)) % kN;
            const f64x2 w = a.tw[m % 1024];
            s_tw[i] = m < 1024 ? cx<double>{w.x, w.y} : cx<double>{-w.x, -w.y};
            const f64x2 v = a.tw[i];
            s_w2[i] = cx<double>{v.x, v.y};
        }
        for (int i = tid; i < kN; i += kBlock) {
            const int m = i >> 1;
            s_win[(m & 31) * kWinRow + 2 * (m >> 5) + (i & 1)] = a.win[i];
        }
        __syncthreads();
    }
    const f64x2 *wrow = reinterpret_cast<const f64x2 *>(s_win + lane * kWinRow);
    const long runs = (a.n_frames + L - 1) / L;
    const long total = (long)a.n_streams * runs;
    const long stride = (long)gridDim.x * kHalves;
    for (long item = (long)blockIdx.x * kHalves + hw; item < total; item += stride) {  // no block barrier below
        const int s = (int)(item / runs);
        const long t0 = (item - (long)s * runs) * L;
        long te = t0 + L;
        if (te > a.n_frames) te = a.n_frames;
        float2 *ys = reinterpret_cast<float2 *>(a.y + (long)s * a.n_frames * kHop);
        float2 tail[16];  // second half of the frame before: sample pair 2 (32 brev5(2 q) + lane) + {0, 1} of its last 1024 samples
        if (t0 == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) tail[q] = reinterpret_cast<const float2 *>(a.tail_in + (long)s * kHop)[32 * brev5(2 * q) + lane];
        }
        for (long t = (t0 == 0 ? 0 : t0 - 1); t < te; ++t) {  // t0 - 1: warm-up frame, only its second half is used
            const f64x2 *row = a.Yh + ((long)s * a.n_frames + t) * kYhStride;
            double re[32], im[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) {  // lane = k1: position i <- k = k1 + 32 brev5(i)
                const int k = lane + 32 * brev5(i);
                const cd y0 = herm_gen(row, k), y1 = herm_gen(row, k + 1024);
                const cx<double> w = s_w2[k];
                const double dr = y0.x - y1.x, di = y0.y - y1.y;
                const double br = dr * w.x + di * w.y, bi = di * w.x - dr * w.y;  // (Y[k] - Y[k + 1024]) conj(W2048^k)
                re[i] = (y0.x + y1.x) - bi;  // A + i B
                im[i] = (y0.y + y1.y) + br;
            }
            fft1024p_inv_A<double>(re, im, lane, s_tw, pbuf);
            __builtin_amdgcn_wave_barrier();
            fft1024p_B<double>(re, lane, pbuf);
            __builtin_amdgcn_wave_barrier();
            fft1024p_C<double, true>(im, lane, pbuf);
            __builtin_amdgcn_wave_barrier();
            fft1024p_D<double, +1>(re, im, lane, pbuf);
            __builtin_amdgcn_wave_barrier();
            // lane = n2: position i holds samples 2 m, 2 m + 1 of the frame, m = 32 brev5(i) + n2
            const bool st_ok = t >= t0;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                float2 fh, sh;
                {
                    const f64x2 w = wrow[brev5(2 * q)];
                    float v0 = (float)(re[2 * q] / (double)kN), v1 = (float)(im[2 * q] / (double)kN);  // util.h:249
                    v0 = (float)((double)v0 * w.x);                                                     // util.h:250
                    v1 = (float)((double)v1 * w.y);
                    if (a.use_post_amp) { v0 = (float)((double)v0 * a.post_amp); v1 = (float)((double)v1 * a.post_amp); }  // mvdr.cpp:112-114
                    fh = float2{v0, v1};
                }
                {
                    const f64x2 w = wrow[brev5(2 * q + 1)];
                    float v0 = (float)(re[2 * q + 1] / (double)kN), v1 = (float)(im[2 * q + 1] / (double)kN);
                    v0 = (float)((double)v0 * w.x);
                    v1 = (float)((double)v1 * w.y);
                    if (a.use_post_amp) { v0 = (float)((double)v0 * a.post_amp); v1 = (float)((double)v1 * a.post_amp); }
                    sh = float2{v0, v1};
                }
                if (st_ok) ys[t * (kHop / 2) + 32 * brev5(2 * q) + lane] = float2{tail[q].x + fh.x, tail[q].y + fh.y};  // out = prev[H + n] + cur[n]  (util.h:301-302)
                tail[q] = sh;
            }
            if (t == a.n_frames - 1) {  // carried state for the next call
#pragma unroll
                for (int q = 0; q < 16; ++q) reinterpret_cast<float2 *>(a.tail_out + (long)s * kHop)[32 * brev5(2 * q) + lane] = tail[q];
            }
        }
    }
}
#endif

// do_overlap's overlap-add (util.h:301-302): out hop t = second half of frame t-1 + first half of frame t
__global__ void ola_generic_kernel(IstftArgs a) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per = a.n_frames * kHop;
    if (idx >= per * a.n_streams) return;
    const int s = (int)(idx / per);
    const long r = idx - (long)s * per;
    const long t = r / kHop;
    const int n = (int)(r - t * kHop);
    const float *fr = a.frames + ((long)s * a.n_frames) * kN;
    const float prev = t >= 1 ? fr[(t - 1) * kN + kHop + n] : a.tail_in[(long)s * kHop + n];
    a.y[idx] = prev + fr[t * kN + n];
    if (t == a.n_frames - 1) a.tail_out[(long)s * kHop + n] = fr[t * kN + kHop + n];
}

#endif  // BF_NFFT == 1024

// full N-bin y_fft dump from the per-problem rows
__global__ void expand_spectrum_kernel(const f64x2 *Yh, f64x2 *out, long frames_total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= frames_total * kN) return;
    const long f = idx / kN;
    const int j = (int)(idx - f * kN);
    const f64x2 *row = Yh + f * kYhStride;
    f64x2 v;
    if (j <= kQX) {
        v = row[j];
    } else {
        v = row[kN - j];
        v.y = -v.y;
    }
    out[idx] = v;
}

__global__ void smooth_kernel(const float *yraw, float *y, const double *state, long n, int n_streams, int sz) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * n_streams) return;
    const int s = (int)(idx / n);
    const long i = idx - (long)s * n;
    const float *yr = yraw + (long)s * n;
    const double *st = state + (long)s * 64;  // st[63] = most recent raw sample before this batch
    double acc = 0.0;
    for (int k = sz - 1; k >= 0; --k) {  // oldest first, as get_mean() sums past_samples[0..]
        const long src = i - k;
        const double v = src >= 0 ? (double)yr[src] : st[64 + src];
        acc += v;
    }
    y[idx] = (float)(acc / (double)sz);
}
// the same for windows of up to 8 samples (the launch files use 3), four consecutive outputs per thread: one 16-byte load brings the
// four newest samples, the SZ - 1 older ones come singly; every output is still the double sum of its window, oldest first
template <int SZ>
__global__ __launch_bounds__(256) void smooth4_kernel(const float *yraw, float *y, const double *state, long n, int n_streams) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;  // quad index; n is a multiple of 4 (whole hops)
    const long nq = n >> 2;
    if (q >= nq * n_streams) return;
    const int s = (int)(q / nq);
    const long i0 = (q - (long)s * nq) << 2;
    const float *yr = yraw + (long)s * n;
    const double *st = state + (long)s * 64;
    double w[SZ + 3];
#pragma unroll
    for (int m = 0; m < SZ - 1; ++m) {
        const long src = i0 - (SZ - 1) + m;
        w[m] = src >= 0 ? (double)yr[src] : st[64 + src];
    }
    const float4 cur = *reinterpret_cast<const float4 *>(yr + i0);
    w[SZ - 1] = (double)cur.x; w[SZ] = (double)cur.y; w[SZ + 1] = (double)cur.z; w[SZ + 2] = (double)cur.w;
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double acc = 0.0;
#pragma unroll
        for (int m = 0; m < SZ; ++m) acc += w[j + m];
        o[j] = (float)(acc / (double)SZ);
    }
    *reinterpret_cast<float4 *>(y + (long)s * n + i0) = float4{o[0], o[1], o[2], o[3]};
}
__global__ void smooth_state_kernel(const float *yraw, double *state, long n, int n_streams) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 64 * n_streams) return;
    const int s = idx / 64, k = idx % 64;
    const long src = n - 64 + k;
    // n >= 512 always (one hop), so the new state is entirely inside this batch
    state[(long)s * 64 + k] = (double)yraw[(long)s * n + src];
}

}  // namespace

#if BF_NFFT == 1024
hipError_t launch_stft(const StftArgs &a, int n_cus, hipStream_t s) {
    const long np = (a.n_fft_mics + 1) / 2;
    // 256 threads, twiddles in LDS: a 512-thread block spills (44 VGPRs) and twiddles read from global memory cost 40 % (measured)
    constexpr int nb = 256, halves = nb / 32;
    // frames per run: one run per half-wavefront slot (one block per CU) when the batch is long enough -- the first frame of a
    // run fetches its leading hop a second time (1/run_len of the input)
    StftArgs b = a;
    const long slots = (long)n_cus * halves;  // (1 / 2 / 4 / 8 runs per slot measured: 0.713 / 0.720 / 0.732 / 0.739 ms)
    long L = ((long)a.n_streams * a.n_frames * np + slots - 1) / slots;
    if (L < 1) L = 1;
    if (L > 256) L = 256;
    b.run_len = (int)L;
    const long total = (long)a.n_streams * ((a.n_frames + L - 1) / L) * np;
    long blocks = (total + halves - 1) / halves;
    const long cap = (long)n_cus * 4;
    if (blocks > cap) blocks = cap;
    if (a.layout == 0) {
        if (a.z48) BF_LAUNCH((stft_kernel<0, true>), dim3((unsigned)blocks), dim3(nb), 0, s, b);
        else BF_LAUNCH((stft_kernel<0, false>), dim3((unsigned)blocks), dim3(nb), 0, s, b);
    } else {
        if (a.z48) BF_LAUNCH((stft_kernel<1, true>), dim3((unsigned)blocks), dim3(nb), 0, s, b);
        else BF_LAUNCH((stft_kernel<1, false>), dim3((unsigned)blocks), dim3(nb), 0, s, b);
    }
    return hipGetLastError();
}

hipError_t launch_istft(const IstftArgs &a, int n_cus, hipStream_t s) {
    if (a.tw32 != nullptr) {  // fp32 backward transform, one frame per FFT
        // two blocks per CU are resident (213-228 VGPRs): one chunk per half-wavefront slot = one round; every chunk recomputes one
        // warm-up frame.  Measured per 65 536 frames: x1 0.181, x2 0.127, x3 0.160 (round 2's value), x4 0.134, x8 0.148 ms
        long slots = (long)n_cus * kI32Halves * 2 / a.n_streams;
        if (slots < 1) slots = 1;
        long cps = slots < a.n_frames ? slots : a.n_frames;
        const long fpc = (a.n_frames + cps - 1) / cps;
        cps = (a.n_frames + fpc - 1) / fpc;
        const long chunks = cps * a.n_streams;
        const dim3 grid((unsigned)((chunks + kI32Halves - 1) / kI32Halves));
        if (!a.yh32) return hipErrorInvalidValue;  // (f64x2 rows go through istft_w64_kernel)
        BF_LAUNCH(istft32_kernel, grid, dim3(kI32Block), 0, s, a, (int)fpc, (int)cps);
        return hipGetLastError();
    }
    const long pairs = (a.n_frames + 1) / 2;
    if (a.tw_w64 == nullptr) return hipErrorInvalidValue;
    // one run per wavefront slot (2 blocks x 4 wavefronts per CU), every run recomputes one warm-up frame
    long slots = (long)n_cus * kIw64Waves * 2 / a.n_streams;
    if (slots < 1) slots = 1;
    long rps = slots < pairs ? slots : pairs;
    const long ppr = (pairs + rps - 1) / rps;
    rps = (pairs + ppr - 1) / ppr;
    const long runs = rps * a.n_streams;
    const dim3 grid((unsigned)((runs + kIw64Waves - 1) / kIw64Waves));
    if (a.yh_lo > 0 || a.yh_hi < a.yh_lo) BF_LAUNCH(istft_w64_kernel<true>, grid, dim3(kIw64Block), 0, s, a, (int)ppr, (int)rps);  // band-limited rows (mvdr / lcmv)
    else BF_LAUNCH(istft_w64_kernel<false>, grid, dim3(kIw64Block), 0, s, a, (int)ppr, (int)rps);
    return hipGetLastError();
}

#else
hipError_t launch_stft(const StftArgs &a, int n_cus, hipStream_t s) {
#if BF_NFFT == 128 || BF_NFFT == 256 || BF_NFFT == 512
    // the in-register kernel (BF_STFT_SMALL=0: the generic one, for A/B runs)
    static const bool small_on = !(getenv("BF_STFT_SMALL") && atoi(getenv("BF_STFT_SMALL")) == 0);
    if (small_on) {
        constexpr int halves = 8;
        const long np = (a.n_fft_mics + 1) / 2;
        StftArgs b = a;
        const long slots = (long)n_cus * halves * 2;  // two rounds of runs per half-wavefront slot
        long L = ((long)a.n_streams * a.n_frames * np + slots - 1) / slots;
        L = ((L + kG - 1) / kG) * kG;  // whole groups of frames
        if (L > 256) L = 256;
        if (L < kG) L = kG;
        b.run_len = (int)L;
        const long items = (long)a.n_streams * ((a.n_frames + L - 1) / L) * np;
        long blocks = (items + halves - 1) / halves;
        if (blocks > (long)n_cus * 4) blocks = (long)n_cus * 4;
        if (a.layout == 0) {
            if (a.z48) BF_LAUNCH((stft_small_kernel<0, true>), dim3((unsigned)blocks), dim3(256), 0, s, b);
            else BF_LAUNCH((stft_small_kernel<0, false>), dim3((unsigned)blocks), dim3(256), 0, s, b);
        } else {
            if (a.z48) BF_LAUNCH((stft_small_kernel<1, true>), dim3((unsigned)blocks), dim3(256), 0, s, b);
            else BF_LAUNCH((stft_small_kernel<1, false>), dim3((unsigned)blocks), dim3(256), 0, s, b);
        }
        return hipGetLastError();
    }
#endif
#if BF_NFFT == 2048
    // one 2048-point transform per full wavefront (BF_STFT_SPLIT=0: the generic kernel, for cross-checks)
    static const bool split_on = !(getenv("BF_STFT_SPLIT") && atoi(getenv("BF_STFT_SPLIT")) == 0);
    if (split_on) {
        constexpr int waves = 4;
        const long np = (a.n_fft_mics + 1) / 2;
        StftArgs b = a;
        const long slots = (long)n_cus * waves * 2;
        long L = ((long)a.n_streams * a.n_frames * np + slots - 1) / slots;
        if (L > 256) L = 256;
        if (L < 1) L = 1;
        b.run_len = (int)L;
        const long items = (long)a.n_streams * ((a.n_frames + L - 1) / L) * np;
        long blocks = (items + waves - 1) / waves;
        if (blocks > (long)n_cus * 4) blocks = (long)n_cus * 4;
        if (a.layout == 0) {
            if (a.z48) BF_LAUNCH((stft_wave2048_kernel<0, true>), dim3((unsigned)blocks), dim3(256), 0, s, b);
            else BF_LAUNCH((stft_wave2048_kernel<0, false>), dim3((unsigned)blocks), dim3(256), 0, s, b);
        } else {
            if (a.z48) BF_LAUNCH((stft_wave2048_kernel<1, true>), dim3((unsigned)blocks), dim3(256), 0, s, b);
            else BF_LAUNCH((stft_wave2048_kernel<1, false>), dim3((unsigned)blocks), dim3(256), 0, s, b);
        }
        return hipGetLastError();
    }
#endif
    const long total = (long)a.n_streams * a.n_frames * ((a.n_fft_mics + 1) / 2);
    long blocks = total < (long)n_cus * 8 ? total : (long)n_cus * 8;
    if (blocks < 1) blocks = 1;
    if (a.layout == 0)
        BF_LAUNCH(stft_generic_kernel<0>, dim3((unsigned)blocks), dim3(kGenBlock), 0, s, a);
    else
        BF_LAUNCH(stft_generic_kernel<1>, dim3((unsigned)blocks), dim3(kGenBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_istft(const IstftArgs &a, int n_cus, hipStream_t s) {
#if BF_NFFT == 128 || BF_NFFT == 256 || BF_NFFT == 512
    // backward transform, window and overlap-add in registers (BF_STFT_SMALL=0: the generic pair of kernels, for A/B runs)
    static const bool small_on = !(getenv("BF_STFT_SMALL") && atoi(getenv("BF_STFT_SMALL")) == 0);
    if (small_on && !a.yh32) {
        constexpr int halves = 8;
        const long slots = (long)n_cus * halves;
        long L = ((long)a.n_streams * a.n_frames + slots - 1) / slots;
        L = ((L + kG - 1) / kG) * kG;  // whole groups of frames
        if (L < 4 * kG) L = 4 * kG;    // a run recomputes one group
        const long items = (long)a.n_streams * ((a.n_frames + L - 1) / L);
        long blocks = (items + halves - 1) / halves;
        BF_LAUNCH(istft_small_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, (int)L);
        return hipGetLastError();
    }
#endif
#if BF_NFFT == 2048
    // one FFT-1024 per frame, window and overlap-add in registers (BF_STFT_SPLIT=0: the generic pair of kernels, for A/B runs)
    static const bool split_on = !(getenv("BF_STFT_SPLIT") && atoi(getenv("BF_STFT_SPLIT")) == 0);
    if (split_on && !a.yh32) {
        constexpr int halves = 8;
        const long slots = (long)n_cus * halves;
        long L = ((long)a.n_streams * a.n_frames + slots - 1) / slots;
        if (L < 8) L = 8;  // a run recomputes one frame
        const long items = (long)a.n_streams * ((a.n_frames + L - 1) / L);
        BF_LAUNCH(istft_split_kernel, dim3((unsigned)((items + halves - 1) / halves)), dim3(256), 0, s, a, (int)L);
        return hipGetLastError();
    }
#endif
    if (a.frames == nullptr) return hipErrorInvalidValue;
    const long total = (long)a.n_streams * a.n_frames;
    long blocks = total < (long)n_cus * 8 ? total : (long)n_cus * 8;
    if (blocks < 1) blocks = 1;
    BF_LAUNCH(istft_generic_kernel, dim3((unsigned)blocks), dim3(kGenBlock), 0, s, a);
    const long samples = total * kHop;
    BF_LAUNCH(ola_generic_kernel, dim3((unsigned)((samples + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

#endif

hipError_t launch_smooth(const float *yraw, float *y, double *state, long n_frames, int n_streams, int smooth_size,
                         hipStream_t s) {
    const long n = n_frames * kHop;
    const long total = n * n_streams;
    const bool al = ((reinterpret_cast<size_t>(yraw) | reinterpret_cast<size_t>(y)) & 15) == 0 && (n & 3) == 0;
    const dim3 g4((unsigned)((total / 4 + 255) / 256));
#define BF_SM4(SZ_) BF_LAUNCH((smooth4_kernel<SZ_>), g4, dim3(256), 0, s, yraw, y, state, n, n_streams)
    if (al && smooth_size == 1) BF_SM4(1);
    else if (al && smooth_size == 2) BF_SM4(2);
    else if (al && smooth_size == 3) BF_SM4(3);
    else if (al && smooth_size == 4) BF_SM4(4);
    else if (al && smooth_size == 5) BF_SM4(5);
    else if (al && smooth_size == 6) BF_SM4(6);
    else if (al && smooth_size == 7) BF_SM4(7);
    else if (al && smooth_size == 8) BF_SM4(8);
    else
        BF_LAUNCH(smooth_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, yraw, y, state, n, n_streams,
                           smooth_size);
#undef BF_SM4
    BF_LAUNCH(smooth_state_kernel, dim3((unsigned)((64 * n_streams + 255) / 256)), dim3(256), 0, s, yraw, state, n,
                       n_streams);
    return hipGetLastError();
}

hipError_t launch_expand_spectrum(const f64x2 *Yh, f64x2 *spectrum, long frames, hipStream_t s) {
    BF_LAUNCH(expand_spectrum_kernel, dim3((unsigned)((frames * kN + 255) / 256)), dim3(256), 0, s, Yh, spectrum, frames);
    return hipGetLastError();
}

}  // namespace BF_NTAG
}  // namespace bf
