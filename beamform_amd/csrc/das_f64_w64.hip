// das_f64_w64.hip -- das at the reference's precision (double arithmetic, das.cpp:47-70 + util.h:217-314), one launch,
// one full wavefront per transform (fft1024_w64.hpp: 64 lanes x 16 points, 16 x 16 x 4).
//
// Two kernels share the transform machinery:
//   das_f64_pair_kernel  planar input (the bench headline).  A complex transform carries frames t and t + 1 of ONE microphone;
//                        U = sum_m ce_m Z_m with the per-microphone Hermitian gains, one backward transform returns both frames
//                        (real / imaginary part): 4.5 transforms per frame.  Frame pairs are handed out to the eight wavefronts of
//                        a block through an LDS counter, the overlap-add between pairs is first come first served (below).
//   das_f64_w64_kernel   [sample][mic] input.  A transform carries two microphones of one frame (one 8-byte
//                        load per sample), S += D_p Z_p with the Hermitian-part pair gains (das_pair_gains_t), Re of the backward
//                        transform: 5 transforms per frame; wavefront w of a block takes frame T0 + 8 it + w.
// Both: (float)Re, float x double window, float overlap-add (util.h:247-252,301-302), 1/N inside the gains.
//
// Mapping onto the chip (what differs from round 3's 32 x 32 half-wavefront kernel):
//   * 64 lanes x 16 points per lane: 64 data + 64 accumulator registers, so TWO wavefronts share a SIMD (the 32 x 32 kernel needs
//     256 + 206 registers, one wavefront per SIMD, ~1 000 v_accvgpr moves per frame).
//   * a 512-thread block per CU walks a run of consecutive frames.  First-pass lane = sample, so every global load is one
//     contiguous 256-byte row; the hop two consecutive frames (pairs) share is fetched by two wavefronts of the same CU within
//     microseconds (L1 / L2): each sample leaves HBM once.
//   * no block barrier after the table copy.  Run boundaries: two float atomic adds into a hop zeroed beforehand
//     (prepare_das_f64_w64), bit-exact because a + b == b + a.
//
// LDS (155-159 KB): twiddles W1024^(k1 lane) + tw2' (16 KB), 8 x 8.1 KB exchange planes (16 rows x 65 doubles, one scalar plane per
// wavefront: real parts, then imaginary parts), 64-65 KB gains, 9 KB window rows, flags.  Exchange layout: fft1024_w64.hpp
// w64_col_rot (every ds_read_b64 / ds_write_b64 group lands on distinct bank pairs).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "launch_trace.hpp"
#include "das_f64_plan.hpp"
#include "pipeline_kernels.hpp"
#include "w64_f64_dev.hpp"

namespace bf {

namespace {

constexpr int kBlock = 512;
constexpr int kWaves = kBlock / 64;
constexpr int kHop = 512;
// LDS map, in doubles
constexpr int kTwD = 2 * (1024 + 4 * kTw2RowW64Rot);  // 1024 + 4 x 17 complex (geometry.hpp twiddle_table_w64_rot)
constexpr int oTw = 0;
constexpr int oPlane = kTwD;
constexpr int oGain = oPlane + kWaves * kPlaneD;
constexpr int oCross = oGain + 4 * 2048;        // 2 x 512 floats
constexpr int oFlag = oCross + 512;             // 2 x 8 ints
constexpr int kWinRow = 18;                     // window as [lane][j] rows of 16 doubles + 2: the 16 lanes of a ds_read_b128 group start 9 bank quads apart
constexpr int oWin = oFlag + 8;
constexpr int kLdsD = oWin + 64 * kWinRow;

typedef volatile __attribute__((address_space(3))) int *lds_flag_t;

// LAYOUT 0: planar [mic][sample]; 1: interleaved [sample][mic] (bf_layout) -- a pair's two microphones are then one 8-byte load per
// sample, the four pairs of a frame read the same 128-byte lines one after the other (from L2: each sample still leaves HBM once)
template <int LAYOUT>
__global__ __launch_bounds__(kBlock) void das_f64_w64_kernel(DasF64Args a, int frames_per_chunk, int chunks_per_stream) {
    __shared__ __attribute__((aligned(16))) double lds[kLdsD];
    const cx<double> *s_tw1 = reinterpret_cast<const cx<double> *>(lds + oTw);
    const cx<double> *s_tw2 = s_tw1 + 1024;
    const cx<double> *s_gain = reinterpret_cast<const cx<double> *>(lds + oGain);
    float *s_cross = reinterpret_cast<float *>(lds + oCross);
    lds_flag_t s_ready = (lds_flag_t)(lds + oFlag);
    lds_flag_t s_cons = s_ready + kWaves;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: frame index, pointers and the flag protocol stay in SGPRs
    const int M = a.n_mics, NP = (M + 1) >> 1;
    double *plane = lds + oPlane + w * kPlaneD;
    double *wcol = plane + w64_col_rot(lane);                       // first-pass lane 4a+b: its column of every row
    double *row16 = plane + (lane & 15) * kRS + 16 * (lane >> 4);   // second-pass lane (b, k1): its segment of row k1

    const int stream = blockIdx.x / chunks_per_stream;
    const long c_in_s = blockIdx.x - (long)stream * chunks_per_stream;
    // window[64 j + lane], j = 0..15: both windows of the frame (util.h:235,250) touch the same 16 values of this lane's row
    const double *wrow = lds + oWin + lane * kWinRow;

    const long T0 = c_in_s * frames_per_chunk;
    long T1 = T0 + frames_per_chunk;
    if (T1 > a.n_frames) T1 = a.n_frames;
    const float *xs = a.x + (long)stream * a.stream_stride_x;
    const float *hs = a.hist + (long)stream * M * kHop;
    float *ys = a.y + (long)stream * a.n_frames * kHop;

    // raw samples of pair p of frame tt (hop tt-1 | hop tt; hop -1 = the carried hop): register j <- sample 64 j + lane
    float na[16], nb[16];
    auto request = [&](long tt, int p) {
        const int ma = 2 * p, mb = (2 * p + 1 < M) ? 2 * p + 1 : ma;
        if (LAYOUT == 0) {
            const float *a1 = tt >= 1 ? xs + (long)ma * a.mic_stride + (tt - 1) * kHop : hs + ma * kHop;
            const float *b1 = tt >= 1 ? xs + (long)mb * a.mic_stride + (tt - 1) * kHop : hs + mb * kHop;
            const float *a2 = xs + (long)ma * a.mic_stride + tt * kHop;
            const float *b2 = xs + (long)mb * a.mic_stride + tt * kHop;
#pragma unroll
            for (int j = 0; j < 8; ++j) {  // scalar base + this lane's 32-bit offset
                na[j] = a1[(unsigned)(64 * j + lane)];
                nb[j] = b1[(unsigned)(64 * j + lane)];
                na[j + 8] = a2[(unsigned)(64 * j + lane)];
                nb[j + 8] = b2[(unsigned)(64 * j + lane)];
            }
        } else {
            const float *s1 = tt >= 1 ? xs + (tt - 1) * (long)kHop * M : hs;  // the carried hop is kept in the same layout
            const float *s2 = xs + tt * (long)kHop * M;
            if ((M & 1) == 0) {  // even microphone count: the pair is 8 bytes, 8-byte aligned
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float2 v1 = *reinterpret_cast<const float2 *>(s1 + (unsigned)((64 * j + lane) * M + ma));
                    const float2 v2 = *reinterpret_cast<const float2 *>(s2 + (unsigned)((64 * j + lane) * M + ma));
                    na[j] = v1.x;
                    nb[j] = v1.y;
                    na[j + 8] = v2.x;
                    nb[j + 8] = v2.y;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    na[j] = s1[(unsigned)((64 * j + lane) * M + ma)];
                    nb[j] = s1[(unsigned)((64 * j + lane) * M + mb)];
                    na[j + 8] = s2[(unsigned)((64 * j + lane) * M + ma)];
                    nb[j + 8] = s2[(unsigned)((64 * j + lane) * M + mb)];
                }
            }
        }
    };

    const int n_iter = (int)((T1 - T0 + kWaves - 1) / kWaves);
    if (T0 + w < T1) request(T0 + w, 0);  // the first frame's samples travel while the tables are copied into LDS
    {
        const f64x2 *tw2 = a.tw, *g2 = a.gains;  // 16-byte copies (kTwD, oGain are even)
        f64x2 *ltw = reinterpret_cast<f64x2 *>(lds + oTw), *lg = reinterpret_cast<f64x2 *>(lds + oGain);
#pragma unroll 4
        for (int i = tid; i < kTwD / 2; i += kBlock) ltw[i] = tw2[i];
#pragma unroll 8
        for (int i = tid; i < NP * 1024; i += kBlock) lg[i] = g2[i];
#pragma unroll 2
        for (int i = tid; i < 1024; i += kBlock) lds[oWin + (i & 63) * kWinRow + (i >> 6)] = a.win[i];
        if (tid < 2 * kWaves) s_ready[tid] = -2;
    }
    __syncthreads();
    for (int it = 0; it < n_iter; ++it) {
        const long t = T0 + (long)it * kWaves + w;
        if (t >= T1) break;  // wavefront-uniform; no block barrier below

        double Sr[16], Si[16];
        for (int p = 0; p < NP; ++p) {
            double re[16], im[16];
            const bool b_ok = 2 * p + 1 < M;
            // buf[j]*hann_win[i] (util.h:235) and the first butterfly stage of the transform in one: x_p w_p +- x_(p+8) w_(p+8) as one product
            // and two FMAs (the second product is not rounded on its own: <= 1 ulp from the separate form)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double w0 = wrow[j], w1 = wrow[j + 8];
                const double t = (double)na[j] * w0, u = (double)na[j + 8];
                re[j] = fma(u, w1, t);
                re[j + 8] = fma(-u, w1, t);
            }
            if (b_ok) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const double w0 = wrow[j], w1 = wrow[j + 8];
                    const double t = (double)nb[j] * w0, u = (double)nb[j + 8];
                    im[j] = fma(u, w1, t);
                    im[j + 8] = fma(-u, w1, t);
                }
            } else {  // odd microphone count: the last pair's second channel is silence
#pragma unroll
                for (int j = 0; j < 16; ++j) im[j] = 0.0;
            }
            {  // next pair (or the next frame's first one; past the run's end: this frame's again, unused): in flight during this transform
                long tn = t;
                int pn = p + 1;
                if (pn == NP) {
                    pn = 0;
                    if (t + kWaves < T1) tn = t + kWaves;
                }
                request(tn, pn);
            }
            cx<double> tw[15];
            BF_STAGE();
            load_tw1<1, 9>(tw, s_tw1, lane);  // lands during the first 16-point transform
            BF_STAGE();
            fft16_core<double, -1, true, 1>(re, im);  // stage 0 is done
            BF_STAGE();
            load_tw1<9, 16>(tw, s_tw1, lane);
            BF_STAGE();
            mul_tw<false, 1, 9>(re, im, tw);
            BF_STAGE();
            mul_tw<false, 9, 16>(re, im, tw);
            if (p == 0 && it > 0 && w < kWaves - 1)  // the plane's head still holds the last frame's tail until w + 1 has taken it
                while (s_cons[w] < (int)(t - kWaves)) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            BF_STAGE();
            T1_fwd(re, im, wcol, row16);
            load_tw2<1, 9>(tw, s_tw2, lane);  // queued behind the exchange's reads: there when the second transform ends
            BF_STAGE();
            fft16_core<double, -1, true>(re, im);
            BF_STAGE();
            load_tw2<9, 16>(tw, s_tw2, lane);
            BF_STAGE();
            mul_tw<false, 1, 9>(re, im, tw);
            BF_STAGE();
            mul_tw<false, 9, 16>(re, im, tw);
            BF_STAGE();
            w64_T2<true>(re, im);
            cx<double> g[16];
            const cx<double> *gp = s_gain + p * 1024 + lane;
            BF_STAGE();
#pragma unroll
            for (int r = 0; r < 8; ++r) g[r] = gp[64 * r];  // lands during the 4-point transforms
            BF_STAGE();
            w64_fwd_p3<double>(re, im);
            BF_STAGE();
#pragma unroll
            for (int r = 8; r < 16; ++r) g[r] = gp[64 * r];
            BF_STAGE();
            if (p == 0) {  // the first pair starts the sum (no zero fill, a multiplication instead of the inner FMA)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (r == 8) BF_STAGE();
                    Sr[r] = fma(-g[r].y, im[r], g[r].x * re[r]);
                    Si[r] = fma(g[r].y, re[r], g[r].x * im[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (r == 8) BF_STAGE();
                    Sr[r] = fma(-g[r].y, im[r], fma(g[r].x, re[r], Sr[r]));
                    Si[r] = fma(g[r].y, re[r], fma(g[r].x, im[r], Si[r]));
                }
            }
        }
        cx<double> tw[15];
        BF_STAGE();
        load_tw2<1, 16>(tw, s_tw2, lane);  // the backward half has registers to spare: re / im are dead
        BF_STAGE();
        w64_inv_p3<double>(Sr, Si);
        w64_T2<false>(Sr, Si);
        BF_STAGE();
        mul_tw<true, 1, 16>(Sr, Si, tw);
        BF_STAGE();
        load_tw1<1, 16>(tw, s_tw1, lane);
        BF_STAGE();
        fft16_core<double, +1, false>(Sr, Si);
        BF_STAGE();
        T1_inv(Sr, Si, row16, wcol);
        BF_STAGE();
        mul_tw<true, 1, 16>(Sr, Si, tw);
        fft16_core<double, +1, false>(Sr, Si);

        // register j holds sample n = 64 j + lane (j < 8: first half, j >= 8: second half); util.h:247-252 with the float stores
        float o[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float f = (float)Sr[j];               // (float)(Re / N): 1/N is inside the gains
            o[j] = (float)((double)f * wrow[j]);        // o *= hann_win[n]
        }
        const int pw = (w + kWaves - 1) & (kWaves - 1);
        if (t + 1 < T1) {  // the next frame of this run takes my second half
            float *area = (w < kWaves - 1) ? reinterpret_cast<float *>(plane) : s_cross + (it & 1) * kHop;
            if (w == kWaves - 1 && it >= 2)  // the slot's previous tail (two steps back) must have been taken: it has, long ago
                while (s_cons[w] < (int)(t - 2 * kWaves)) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) area[64 * j + lane] = o[j + 8];
            asm volatile("" ::: "memory");
            if (lane == 0) s_ready[w] = (int)t;  // LDS operations of a wavefront complete in order: the tail is there first
        }
        float *yo = ys + t * kHop;
        if (t == T0) {
            if (T0 == 0) {  // stream start: the partner is the carried state (out_buff[0] of the previous call)
                const float *ti = a.tail_in + (long)stream * kHop;
#pragma unroll
                for (int j = 0; j < 8; ++j) yo[(unsigned)(64 * j + lane)] = ti[(unsigned)(64 * j + lane)] + o[j];
            } else {  // first hop of a run: the previous run adds its half separately, both into a zeroed hop
#pragma unroll
                for (int j = 0; j < 8; ++j) atomicAdd(yo + (unsigned)(64 * j + lane), o[j]);
            }
        } else {
            const float *parea = (pw < kWaves - 1) ? reinterpret_cast<const float *>(lds + oPlane + pw * kPlaneD) : s_cross + ((it + 1) & 1) * kHop;
            while (s_ready[pw] < (int)(t - 1)) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            float prev[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) prev[j] = parea[64 * j + lane];
            asm volatile("" ::: "memory");
            if (lane == 0) s_cons[pw] = (int)(t - 1);
#pragma unroll
            for (int j = 0; j < 8; ++j) yo[(unsigned)(64 * j + lane)] = prev[j] + o[j];  // out_buff[0][j] + out_buff[1][j] as floats (util.h:302)
        }
        if (t == T1 - 1) {
            if (T1 < a.n_frames) {  // last frame of the run: its second half belongs to the next run's first hop
                float *yn = ys + T1 * kHop;
#pragma unroll
                for (int j = 0; j < 8; ++j) atomicAdd(yn + (unsigned)(64 * j + lane), o[j + 8]);
            } else {  // end of the batch: carried state for the next call
                float *to = a.tail_out + (long)stream * kHop;
#pragma unroll
                for (int j = 0; j < 8; ++j) to[(unsigned)(64 * j + lane)] = o[j + 8];
            }
        }
    }
}


// =====================================================================================================================================
//                 frame-pair kernel: two consecutive frames of ONE microphone per complex transform (planar input)
// =====================================================================================================================================
// The kernel above packs two microphones of one frame into a transform and takes Re of the backward transform: half of that
// transform's output is thrown away.  Packing the SAME microphone's frames t and t + 1 instead (z = x_t + i x_(t+1)) keeps every
// spectrum Hermitian-separable:  U = sum_m ce_m (X_m(t) + i X_m(t+1)) = H_t + i H_(t+1)  with ce_m the Hermitian part of conj(w_m) / M
// (geometry.hpp das_mic_gains_w64_f64; ce_m[N - k] = conj ce_m[k]) and H real-output spectra, so ONE backward transform returns y_t in
// its real and y_(t+1) in its imaginary part: 4.5 transforms per frame instead of 5, three hops loaded per microphone and frame pair
// instead of four, one overlap-add across wavefronts per two frames (the inner one is a register add: samples n and n + 512 share a lane).
//
// Scheduling.  The two wavefronts of a SIMD do not share it evenly: the older one issues whenever it can (it runs at the lone-wavefront
// rate), the younger one gets what is left (tools/stats_w64.py: with a fixed frame -> wavefront map and a ring of hand-offs, wavefronts
// 0-3 slept a third of the kernel waiting for 4-7, i.e. a third of the time each SIMD ran ONE wavefront).  So (i) frame pairs are taken
// from an LDS counter -- the fast wavefronts end up with ~72 % of them and all eight finish together -- and (ii) the overlap-add
// between pairs never waits for the other side's arithmetic (claim_boundary below).
// Gains: ce_m is Hermitian, so only bins 0 .. 512 are kept: [mic][row 2 g + k3, k3 < 2][65] = ce_m[64 g + 256 k3 + c], c = 0 .. 64.
// Register 4 g + k3 of lane l is bin l + 64 g + 256 k3: k3 < 2 reads row (g, k3) column l; k3 >= 2 reads row (3 - g, 3 - k3) column
// 64 - l (N - k = (64 - l) + 64 (3 - g) + 256 (3 - k3)) and conjugates through the FMA signs.  The 65th column is the bin behind the row.
// read-back of a parked half: non-temporal (used once); a plain load moved 4 % more bytes across the fabric (-DBF_PARK_PLAIN for A/B runs)
#ifdef BF_PARK_PLAIN
#define BF_PARK_LD(p) (*(const volatile float *)(p))
#else
#define BF_PARK_LD(p) __builtin_nontemporal_load(p)
#endif
constexpr int kGRow = 65;                                  // complex entries per gain row
constexpr int kGMic = 8 * kGRow;                           // per microphone
constexpr int kSlots = 64;                                 // boundary states in flight (<= 2 pairs per wavefront are): a ring
constexpr int kTwP = 2 * (960 + 4 * kTw2RowW64Rot);        // tw1 rows k1 = 1 .. 15 (row 0 is all ones and never read) + tw2'
constexpr int pTw = 0;
constexpr int pPlane = kTwP;
constexpr int pGain = pPlane + kWaves * kPlaneD;
constexpr int pFlag = pGain + 7 * kGMic * 2;       // microphones 1 .. 7 (microphone 0 has no gains: below)
constexpr int kRingR = 32;                                 // [sample][mic] input: shared hop slots of the block's ring (16 pairs), + one private slot per wavefront
constexpr int kRingSlots = kRingR + kWaves;
constexpr int pRing = pFlag + (kSlots + 4) / 2;          // kSlots ints + the 64-bit work word, padded to 16 bytes; then kRingR slot states
constexpr int pWin = pRing + kRingR / 2;
constexpr int kLdsP = pWin + 64 * kWinRow;
static_assert(kLdsP * 8 <= 160 * 1024, "LDS");
static_assert((pGain & 1) == 0 && (pWin & 1) == 0, "16-byte alignment");


typedef volatile __attribute__((address_space(3))) int *lds_int_t;
typedef volatile __attribute__((address_space(3))) unsigned long long *lds_u64_t;
__device__ __forceinline__ int lds_fetch_add(lds_int_t p, int v, int lane) {
    int old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add((__attribute__((address_space(3))) int *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __builtin_amdgcn_readfirstlane(old);
}
__device__ __forceinline__ unsigned long long lds_fetch_add64(lds_u64_t p, int lane) {
    unsigned long long old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add((__attribute__((address_space(3))) unsigned long long *)p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)old), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(old >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
struct DasSched {           // the work queue of das_f64_pair_kernel (written by das_f64_sched_kernel on the launch stream)
    const int4 *chunks;     // [n_chunks] {stream, first frame, frames, 0}: runs of consecutive frames of one stream, biggest first
    unsigned *counter;      // next chunk to hand out (starts at the grid size: block b begins with chunk b)
    int n_chunks;
};
// ---- work queue ------------------------------------------------------------------------------------------------------------------
// The batch is cut into CHUNKS of consecutive frames of one stream (das_f64_sched_kernel writes the table, biggest chunks first).
// Blocks are persistent: block b starts with chunk b and takes further chunks from one global counter, so a block on a fast XCD (the
// eight XCDs finish equal static runs 4-5 % apart: profiles/r05_finish_hist_static.txt) simply ends up with more of the small chunks
// at the table's end.  Inside a block the frame pairs of the current chunk are handed to the eight wavefronts through ONE 64-bit LDS
// word  [ virtual index of the chunk's first pair : 24 | chunk : 20 | pairs in the chunk : 10 | next pair : 10 ],  advanced by an LDS
// atomic add: whoever draws pair == pairs (the first one past the end) fetches the next chunk from the global counter and installs
// it; later arrivals spin on the word until the chunk field changes.  The virtual index numbers the pairs a block has drawn 0, 1, 2, ...
// across chunks: boundary state slot = virtual index mod kSlots, whatever the chunks' sizes.
// (kChunkEnd = 0xFFFFF in the chunk field: the table is exhausted; kMaxChunkPairs = 1000 pairs per chunk: 10 bits, and up to 8 draws past the
// end before the word is replaced -- das_f64_plan.hpp)
__device__ __forceinline__ unsigned long long pack_work(unsigned vbase, int chunk, int pairs, int next) {
    return ((unsigned long long)(vbase & 0xFFFFFFu) << 40) | ((unsigned long long)(unsigned)chunk << 20) | ((unsigned long long)(unsigned)pairs << 10) | (unsigned)next;
}
struct PairWork {   // wavefront-uniform
    int4 d;         // the chunk: {stream, first frame, frames, index in the table}
    int pos;        // this pair inside the chunk
    int len;        // pairs of the chunk
    unsigned u;     // virtual index
    bool have;
};
__device__ __forceinline__ int4 load_chunk(const int4 *chunks, int k) {
    const int4 d = chunks[k];
    return int4{__builtin_amdgcn_readfirstlane(d.x), __builtin_amdgcn_readfirstlane(d.y), __builtin_amdgcn_readfirstlane(d.z), k};
}
// `held`: the chunk the wavefront is working on.  Nearly every draw stays inside it (104 pairs at the headline size): its row of the table is then
// not fetched again (a global load + s_waitcnt vmcnt(0) with nothing else in flight: one exposed L2 round trip per pair until round 6)
__device__ __forceinline__ PairWork draw_pair(lds_u64_t work, const DasSched &sc, int lane, const int4 held) {
    PairWork r;
    r.have = false;
    r.d = int4{0, 0, 0, 0};
    r.pos = r.len = 0;
    r.u = 0;
    for (;;) {
        const unsigned long long old = lds_fetch_add64(work, lane);
        const int pos = (int)(old & 1023u), len = (int)((old >> 10) & 1023u), k = (int)((old >> 20) & 0xFFFFFu);
        const unsigned vb = (unsigned)(old >> 40);
        if (k == kChunkEnd) return r;
        if (pos < len) {
            if (k == held.w) r.d = held;
            else r.d = load_chunk(sc.chunks, k);
            r.pos = pos; r.len = len; r.u = vb + (unsigned)pos; r.have = true;
            return r;
        }
        if (pos == len) {  // first past the end: install the next chunk (the others spin below meanwhile)
            unsigned k2 = 0;
            if (lane == 0) k2 = __hip_atomic_fetch_add(sc.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            k2 = (unsigned)__builtin_amdgcn_readfirstlane((int)k2);
            if (k2 >= (unsigned)sc.n_chunks) {
                if (lane == 0) *work = pack_work(vb + (unsigned)len, kChunkEnd, 0, 0);
                return r;
            }
            r.d = load_chunk(sc.chunks, (int)k2);
            r.len = (r.d.z + 1) >> 1;
            r.pos = 0; r.u = vb + (unsigned)len; r.have = true;
            if (lane == 0) *work = pack_work(r.u, (int)k2, r.len, 1);   // (a plain store: adds that land before it belong to spinners, who draw again)
            return r;
        }
        while ((int)((*work >> 20) & 0xFFFFFu) == k) __builtin_amdgcn_s_sleep(2);
    }
}
// Boundary b (between the pairs of virtual index b and b + 1) lives in slot b mod kSlots; over its life the slot gains 4 (two claims of 1,
// one publication of 2 -- or 4 at once where b + 1 belongs to another chunk and there is nothing to hand over), so generation b / kSlots
// starts at 4 (b / kSlots).  claim: 0 = first to arrive; otherwise the other side has claimed (1) or already published (3).  A claim waits
// until the slot's previous generation is complete (a wavefront 64 pairs behind its siblings: never seen, but nothing else forbids it).
__device__ __forceinline__ int claim_boundary(lds_int_t st, unsigned b, int lane) {
    const int base = 4 * (int)(b / kSlots);
    while (st[b & (kSlots - 1)] < base) __builtin_amdgcn_s_sleep(1);
    return lds_fetch_add(st + (b & (kSlots - 1)), 1, lane) - base;
}
__device__ __forceinline__ void publish_boundary(lds_int_t st, unsigned b, int lane) { (void)lds_fetch_add(st + (b & (kSlots - 1)), 2, lane); }
__device__ __forceinline__ void skip_boundary(lds_int_t st, unsigned b, int lane) {
    const int base = 4 * (int)(b / kSlots);
    while (st[b & (kSlots - 1)] < base) __builtin_amdgcn_s_sleep(1);
    (void)lds_fetch_add(st + (b & (kSlots - 1)), 4, lane);
}
__device__ __forceinline__ void await_boundary(lds_int_t st, unsigned b) {
    while (st[b & (kSlots - 1)] - 4 * (int)(b / kSlots) < 3) __builtin_amdgcn_s_sleep(1);
}

#ifdef BF_W64_STATS  // debug build (tools/ab_w64.sh stats -DBF_W64_STATS): how the hand-offs of das_f64_pair_kernel went, per wavefront
// [block][wavefront][partner there before the backward transform / at the epilogue / had to wait / 10 ns ticks waited / ticks in the kernel]
__device__ unsigned long long g_stats[256 * 8 * 5];
#define BF_STATS_DECL unsigned long long st_[5] = {0, 0, 0, 0, 0}; const unsigned long long st_t0_ = __builtin_amdgcn_s_memrealtime()
#define BF_STAT(i) (++st_[i])
#define BF_STAT_WAIT_BEGIN const unsigned long long st_w0_ = __builtin_amdgcn_s_memrealtime()
#define BF_STAT_WAIT_END st_[3] += __builtin_amdgcn_s_memrealtime() - st_w0_
#define BF_STATS_FLUSH do { st_[4] = __builtin_amdgcn_s_memrealtime() - st_t0_; if (lane == 0 && blockIdx.x < 256) for (int i_ = 0; i_ < 5; ++i_) g_stats[(blockIdx.x * 8 + w) * 5 + i_] = st_[i_]; } while (0)
#else
#define BF_STATS_DECL
#define BF_STAT(i) ((void)0)
#define BF_STAT_WAIT_BEGIN
#define BF_STAT_WAIT_END
#define BF_STATS_FLUSH
#endif
#ifdef BF_W64_STAMPS  // debug build (tools/ab_w64.sh stamps -DBF_W64_STAMPS): s_memrealtime (100 MHz) per wavefront: entry, tables done, exit
__device__ unsigned long long g_stamps[256 * 8 * 64];
#define BF_STAMP(slot) do { if (lane == 0 && blockIdx.x < 256 && (slot) < 64) g_stamps[(blockIdx.x * 8 + w) * 64 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define BF_STAMP(slot) do { } while (0)
#endif

// One tile of [sample][mic] input -- 1024 floats = 1024 / M samples of M = 2^LG microphones, lane l of load i holding floats 4 (64 i + l) ... + 3 --
// through a wavefront-private LDS tile [sample][M + 1] and out as [mic][sample]: 16-byte stores of 4 consecutive samples of one microphone
// into dst + mic * 512 + t * (1024 / M).  With M a template parameter every LDS address is one per-lane base plus an immediate.
typedef float il_f4 __attribute__((ext_vector_type(4)));
template <int LG>
__device__ __forceinline__ void il_tile_through_plane(const il_f4 (&v)[4], float *tile, float *dst, int t, int lane) {
    // row pitch 10 (8 microphones) / 5 / 3 floats and, on the way out, lane -> (microphone = lane mod M, sample quad = lane / M + (64 / M) j): the 64
    // lanes of a read then cover all 64 banks once (pitch 9 with the microphone as the slow index: 4 R1 q mod 64 takes 16 values: 4-way conflicts)
    constexpr int M = 1 << LG, R1 = LG == 3 ? 10 : M + 1, spt = 1024 >> LG;
    asm volatile("" : "+v"(lane)::"memory");  // (the bases depend on the lane only: hoisted out of the pair loop they would live, or spill, through every transform)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int e = 256 * i + 4 * lane + c;  // float index in the tile: sample e >> LG, microphone e & (M - 1)
            tile[(e >> LG) * R1 + (e & (M - 1))] = v[i][c];
        }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();  // (LDS operations of one wavefront execute in issue order)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int mic = lane & (M - 1), q4 = (lane >> LG) + (64 >> LG) * j;
        il_f4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = tile[(4 * q4 + c) * R1 + mic];
        reinterpret_cast<il_f4 *>(dst + mic * kHop + t * spt)[q4] = o;
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// LAYOUT 0: planar input.  LAYOUT 1: [sample][mic] input (round 6, second session): the kernel walks ONE microphone at a time, which on that layout
// makes every 128-byte line cross L2 -> L1 eight times (1.34 ms), and a transposition kernel in front costs 0.35 ms at the copy ceiling while
// the vector pipes idle.  Here the wavefront that draws a pair transposes the pair's two NEW hops (2 x 512 samples x M microphones, 16-byte
// loads, through its idle exchange plane) into the block's ring of planar hop slots in global memory and reads its three hops per microphone
// from there, coalesced, as in the planar case; the hop it shares with the previous pair is that pair's second slot.  Slot states in LDS (as
// the boundary states): a slot's life is worth 4 -- published 2, readers 1 + 1 (the first hop of a pair has one reader: 2) -- so generation g
// of a slot starts at 4 g; a writer waits for 4 g (every reader of the previous content is done), a reader for 4 g + 2.  The first pair of
// a chunk puts the hop in front of it (or the carried hop) into its wavefront's private slot.  The memory side of the transposition
// overlaps the other wavefronts' transforms: measured with a plain 32 KB + 32 KB copy per pair inside the planar kernel: + 0.19 ms.
template <int LAYOUT>
__device__ __forceinline__ void das_f64_pair_body(const DasF64Args &a, const DasSched &sc) {
    __shared__ __attribute__((aligned(16))) double lds[kLdsP];
    const cx<double> *s_tw1 = reinterpret_cast<const cx<double> *>(lds + pTw) - 64;  // row k1 starts at 64 (k1 - 1)
    const cx<double> *s_tw2 = reinterpret_cast<const cx<double> *>(lds + pTw) + 960;
    const cx<double> *s_gain = reinterpret_cast<const cx<double> *>(lds + pGain);
    lds_int_t s_state = (lds_int_t)(lds + pFlag);              // kSlots boundary states
    lds_u64_t s_work = (lds_u64_t)(lds + pFlag + kSlots / 2);  // the block's work word (above)
    lds_int_t s_ring = (lds_int_t)(lds + pRing);               // LAYOUT 1: states of the ring's shared hop slots
    (void)s_ring;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = a.n_mics;
    double *plane = lds + pPlane + w * kPlaneD;
    double *wcol = plane + w64_col_rot(lane);
    double *row16 = plane + (lane & 15) * kRS + 16 * (lane >> 4);
    const double *wrow = lds + pWin + lane * kWinRow;
    // the gain rows lie beyond the 64 KB a DS offset field reaches: a per-lane base register that hipcc cannot split back into
    // "lane + constant" (it then spent 16 v_add_u32 per microphone on the sixteen row addresses)
    typedef double cxa __attribute__((ext_vector_type(2)));  // 16-byte aligned (re, im): one ds_read_b128
    typedef const __attribute__((address_space(3))) cxa *lds_gain_t;
    unsigned gd0 = (unsigned)(size_t)(lds_gain_t)(s_gain + lane);
    asm volatile("" : "+v"(gd0));
    gd0 &= ~15u;  // (what the asm hid: 16-byte alignment)
    const unsigned g_mirror = 2u * (unsigned)(size_t)(lds_gain_t)s_gain + 1024u;  // byte address of column 64 - lane = g_mirror - that of column lane

    // hops tA - 1 (hop -1 = the carried hop), tA, tA + 1 of one microphone: register j <- sample 64 j + lane of the hop.  h1 = hop tA.
    float n0[8], n1[8], n2[8];
#ifndef BF_PAIR_NT
#define BF_PAIR_NT 0
#endif
    auto request = [&](const float *h0, const float *h1, const float *h2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            n0[j] = (BF_PAIR_NT & 2) ? __builtin_nontemporal_load(h0 + (unsigned)(64 * j + lane)) : h0[(unsigned)(64 * j + lane)];
            n1[j] = (BF_PAIR_NT & 1) ? __builtin_nontemporal_load(h1 + (unsigned)(64 * j + lane)) : h1[(unsigned)(64 * j + lane)];
            n2[j] = (BF_PAIR_NT & 4) ? __builtin_nontemporal_load(h2 + (unsigned)(64 * j + lane)) : h2[(unsigned)(64 * j + lane)];
        }
    };
    // the three hop pointers of microphone m of the pair that starts with frame tA of `stream`
    auto request_pair_mic = [&](int stream, long tA, int m) {
        const float *h1 = a.x + (long)stream * a.stream_stride_x + (long)m * a.mic_stride + tA * kHop;
        const float *h0 = tA >= 1 ? h1 - kHop : a.hist + ((long)stream * M + m) * kHop;
        const float *h2 = tA + 1 < a.n_frames ? h1 + kHop : h1;  // a lone last frame: any readable hop, unused
        request(h0, h1, h2);
    };
    // Microphone 0 is never transformed.  Its weight row is identically 1 (das.cpp:33-38: written once, tau_0 = 0), so its term of
    // das.cpp:60-63 is X_0[j] / M and FFT -> IFFT / N of it is the windowed frame itself: h[n] x_0[n] / M (util.h:235,247-252), added in
    // double to Re(u[n]) / N in front of the float cast.  4.0 transforms per frame instead of 4.5.  Its three hops are requested when
    // the backward transform starts (re / im are dead then: registers to spare) and are there when it ends.
    float q0[8], q1[8], q2[8];
    const double inv_m = 1.0 / (double)M;
    // Microphones with IDENTICAL weight rows share a transform (the reference drops z -- util.h:82-92 -- so microphones 1 and 7 of its aira16
    // array, beamform_config.yaml:21,27, have the same delays for every look direction): sum_m conj(w_m) X_m takes conj(w) FFT(h (x_a + x_b))
    // for such a pair, x_a + x_b formed in double (exact).  The host puts ONE such pair into slot 0 (DasF64Args::slot_mic, extra_mic): the
    // second microphone's three hops travel beside the first one's while the previous pair's backward transform runs (re / im dead).
    float e0[8], e1[8], e2[8];
    const int NT = a.n_tr, XM = a.extra_mic;
    // (called on every path, `on` false = zeros: a conditional definition would keep the 24 registers live through the forward transforms)
    auto request_extra = [&](bool on, int stream, long tA) {
        if (!on) {
#pragma unroll
            for (int j = 0; j < 8; ++j) e0[j] = e1[j] = e2[j] = 0.f;
            return;
        }
        const float *h1 = a.x + (long)stream * a.stream_stride_x + (long)XM * a.mic_stride + tA * kHop;
        const float *h0 = tA >= 1 ? h1 - kHop : a.hist + ((long)stream * M + XM) * kHop;
        const float *h2 = tA + 1 < a.n_frames ? h1 + kHop : h1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            e0[j] = h0[(unsigned)(64 * j + lane)];
            e1[j] = h1[(unsigned)(64 * j + lane)];
            e2[j] = h2[(unsigned)(64 * j + lane)];
        }
    };

    // ---- LAYOUT 1: the block's hop ring ---------------------------------------------------------------------------------------------------
    struct RingHops { const float *h, *a, *b; };  // slot bases [mic][512] of hops tA - 1, tA, tA + 1 (wavefront-uniform)
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int lgM = 31 - __builtin_clz((unsigned)M);  // M is 2, 4 or 8 here (the host checks)
    float *ringb = LAYOUT == 1 ? a.ring + (size_t)blockIdx.x * kRingSlots * M * kHop : nullptr;
    auto ring_slot = [&](unsigned j) { return ringb + (size_t)(j & (kRingR - 1)) * M * kHop; };
    // one hop = 512 M floats in M / 2 tiles of 1024 floats (1024 / M samples), moved in UNITS of two tiles (32 registers): 16-byte loads,
    // lane l of load i holds floats 4 (64 i + l) ... of its tile
    auto il_load2 = [&](const float *src, int unit, f4 (&v)[2][4]) {
        const f4 *s4 = reinterpret_cast<const f4 *>(src);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int t = 2 * unit + tt;
            if (2 * t < M) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[tt][i] = s4[t * 256 + i * 64 + lane];
            } else {  // (2 microphones: half a unit; defined on every path)
#pragma unroll
                for (int i = 0; i < 4; ++i) v[tt][i] = f4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    // ... through the wavefront's exchange plane as [sample][M + 1] floats, out as 16-byte stores of 4 consecutive samples of one microphone
    // (il_tile_through_plane below: the microphone count as a template parameter turns every LDS address into lane base + immediate)
    auto il_store2 = [&](const f4 (&v)[2][4], float *dst, int unit) {
        float *tile = reinterpret_cast<float *>(plane);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int t = 2 * unit + tt;
            if (2 * t < M) {
                if (lgM == 3) il_tile_through_plane<3>(v[tt], tile, dst, t, lane);
                else if (lgM == 2) il_tile_through_plane<2>(v[tt], tile, dst, t, lane);
                else il_tile_through_plane<1>(v[tt], tile, dst, t, lane);
            }
        }
    };
    // the pair `pw` is about to become this wavefront's: its hops into the ring, then the three slot bases
    auto prepare_pair = [&](const PairWork &pw) {
        RingHops r;
        const unsigned u2 = 2u * pw.u;
        const int sA = (int)(u2 & (kRingR - 1)), gen4 = 4 * (int)(u2 / kRingR);
        float *dA = ring_slot(u2), *dB = ring_slot(u2 + 1), *dH = ringb + (size_t)(kRingR + w) * M * kHop;
        const long tA = pw.d.y + 2L * pw.pos;
        const float *xs = a.x + (long)pw.d.x * a.stream_stride_x;
        while (s_ring[sA] < gen4 || s_ring[sA + 1] < gen4) __builtin_amdgcn_s_sleep(1);  // every reader of the slots' previous hops is done
        f4 v0[2][4], v1[2][4];  // two units in flight: the loads of one travel while the other goes through the plane
        const bool two = tA + 1 < a.n_frames, first = pw.pos == 0, wide = M > 4;  // wide: a hop is two units
        const float *srcA = xs + tA * (long)kHop * M, *srcB = srcA + (long)kHop * M;
        const float *srcH = tA >= 1 ? srcA - (long)kHop * M : a.hist + (long)pw.d.x * M * kHop;
        il_load2(srcA, 0, v0);
        if (wide) il_load2(srcA, 1, v1);
        il_store2(v0, dA, 0);
        if (two) il_load2(srcB, 0, v0);
        if (wide) il_store2(v1, dA, 1);
        if (two && wide) il_load2(srcB, 1, v1);
        if (two) il_store2(v0, dB, 0);
        if (first) il_load2(srcH, 0, v0);
        if (two && wide) il_store2(v1, dB, 1);
        if (first && wide) il_load2(srcH, 1, v1);
        if (first) il_store2(v0, dH, 0);
        if (first && wide) il_store2(v1, dH, 1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        (void)lds_fetch_add(s_ring + sA, 2, lane);
        (void)lds_fetch_add(s_ring + sA + 1, 2, lane);
        r.a = dA;
        r.b = two ? dB : dA;  // a lone last frame: any readable hop, unused
        r.h = dH;
        if (!first) {  // the previous pair's second hop: published by whoever drew that pair
            const unsigned jp = u2 - 1;
            const int sP = (int)(jp & (kRingR - 1)), genp = 4 * (int)(jp / kRingR);
            while (s_ring[sP] - genp < 2) __builtin_amdgcn_s_sleep(1);
            r.h = ring_slot(jp);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        return r;
    };
    // ... and has read the last of them (has_t: the pair behind it reads its second hop too)
    auto release_pair = [&](const PairWork &pw, bool has_t) {
        const unsigned u2 = 2u * pw.u;
        (void)lds_fetch_add(s_ring + (u2 & (kRingR - 1)), 2, lane);
        (void)lds_fetch_add(s_ring + ((u2 + 1) & (kRingR - 1)), has_t ? 1 : 2, lane);
        if (pw.pos > 0) (void)lds_fetch_add(s_ring + ((u2 - 1) & (kRingR - 1)), 1, lane);
    };
    auto request_ring_mic = [&](const RingHops &r, int m) { request(r.h + m * kHop, r.a + m * kHop, r.b + m * kHop); };
    auto request_ring_extra = [&](bool on, const RingHops &r) {
        if (!on) {
#pragma unroll
            for (int j = 0; j < 8; ++j) e0[j] = e1[j] = e2[j] = 0.f;
            return;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            e0[j] = r.h[XM * kHop + (unsigned)(64 * j + lane)];
            e1[j] = r.a[XM * kHop + (unsigned)(64 * j + lane)];
            e2[j] = r.b[XM * kHop + (unsigned)(64 * j + lane)];
        }
    };
    RingHops curR{nullptr, nullptr, nullptr};
    (void)curR;

    BF_STAMP(0);
#ifdef BF_W64_STAMPS
    if (lane == 0 && blockIdx.x < 256) g_stamps[(blockIdx.x * 8 + w) * 64 + 59] = __builtin_amdgcn_s_memtime();
#endif
    // the first kWaves pairs of the block's first chunk are handed out statically: their samples travel during the table copy
    PairWork cur;
    cur.d = load_chunk(sc.chunks, (int)blockIdx.x);
    cur.len = (cur.d.z + 1) >> 1;
    const int n_static = cur.len < kWaves ? cur.len : kWaves;
    cur.pos = w;
    cur.u = (unsigned)w;
    cur.have = w < n_static;
    if (LAYOUT == 0 && cur.have) {
        request_pair_mic(cur.d.x, cur.d.y + 2L * w, a.slot_mic[0]);
        request_extra(XM >= 0, cur.d.x, cur.d.y + 2L * w);
    } else {
        request_extra(false, 0, 0);
    }
    {
        const f64x2 *tw2 = a.tw + 64;
        f64x2 *ltw = reinterpret_cast<f64x2 *>(lds + pTw), *lg = reinterpret_cast<f64x2 *>(lds + pGain);
#pragma unroll 4
        for (int i = tid; i < kTwP / 2; i += kBlock) ltw[i] = tw2[i];
        for (int k = 0; k < NT; ++k) {  // the gains of slot k's microphone (microphone 0 has none: below)
            const f64x2 *g2 = a.gains_mic + a.slot_mic[k] * kGMic;
#pragma unroll 2
            for (int i = tid; i < kGMic; i += kBlock) lg[k * kGMic + i] = g2[i];
        }
#pragma unroll 2
        for (int i = tid; i < 1024; i += kBlock) lds[pWin + (i & 63) * kWinRow + (i >> 6)] = a.win[i];
        if (tid < kSlots) s_state[tid] = 0;
        if (tid < kRingR) s_ring[tid] = 0;
        if (tid == 0) *s_work = pack_work(0u, (int)blockIdx.x, cur.len, n_static);
    }
    __syncthreads();
    BF_STAMP(1);
    BF_STATS_DECL;
    int it = 0;  // pairs this wavefront has done (debug stamps)
    if (!cur.have) {
        cur = draw_pair(s_work, sc, lane, cur.d);
        if (LAYOUT == 0 && cur.have) {
            request_pair_mic(cur.d.x, cur.d.y + 2L * cur.pos, a.slot_mic[0]);
            request_extra(XM >= 0, cur.d.x, cur.d.y + 2L * cur.pos);
        }
    }
    while (cur.have) {  // wavefront-uniform; no block barrier below
        if constexpr (LAYOUT == 1) {  // this pair's hops into the ring (ONE inlined copy of the transposition: here), its first microphone out of it
            curR = prepare_pair(cur);
            request_ring_mic(curR, a.slot_mic[0]);
            request_ring_extra(XM >= 0, curR);
        }
        const int stream = cur.d.x;
        const long T0 = cur.d.y, T1 = T0 + cur.d.z;  // the chunk
        const long tA = T0 + 2L * cur.pos;
        const bool pair = tA + 1 < T1;       // false: the odd last frame of a stream on its own (imaginary input zero)
        const long tL = pair ? tA + 1 : tA;  // the frame whose second half leaves this wavefront
        const unsigned u = cur.u;
        const bool has_t = cur.pos + 1 < cur.len, has_h = cur.pos > 0;  // boundaries u (behind this pair) and u - 1 (in front of it) inside the chunk
        float *ys = a.y + (long)stream * a.n_frames * kHop;
        PairWork nxt;
        nxt.have = false;
        // hop tA of the microphone whose loads are in flight (request_pair_mic asked for microphone 1 of this pair); the carried hop
        // stands in for hop -1, the last frame of a stream has no hop behind it
        const float *hp0 = a.x + (long)stream * a.stream_stride_x + tA * kHop;  // microphone 0's
        const float *hist0 = a.hist + (long)stream * M * kHop;
        const bool first_hop = tA < 1, last_hop = !(tA + 1 < a.n_frames);
        (void)hp0; (void)hist0; (void)first_hop; (void)last_hop;

        double Sr[16], Si[16];
        // one microphone: forward transform of (frame tA, frame tA + 1) and S += ce_m Z_m.  Instantiated twice (FIRST: the microphone
        // that starts the sum with a multiplication): a run-time test per accumulator cost 32 scalar branches per microphone
        auto one_mic = [&](const int k, auto first_tag) {  // slot k of the pair
            constexpr bool FIRST = decltype(first_tag)::value;
            double re[16], im[16];
#ifndef BF_PAIR_NO_VMWAIT
            // this microphone's 24 loads were requested a transform ago and nothing younger is in flight: ONE wait instead of the sixteen
            // counted ones hipcc places between the conversions (not for the first microphone: the previous pair's stores are still in flight)
            if constexpr (!FIRST) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), lgkmcnt / expcnt untouched
#endif
            // buf[j]*hann_win[i] (util.h:235) and the first butterfly stage in one (see the kernel above); hop tA is the second half of
            // frame tA and the first half of frame tA + 1
            if (FIRST && XM >= 0) {  // slot 0 carries two microphones with the same weight row: their samples are added in double (exact) in front of the window
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const double w0 = wrow[j], w1 = wrow[j + 8];
                    const double c = (double)n1[j] + (double)e1[j];
                    const double t = ((double)n0[j] + (double)e0[j]) * w0;
                    re[j] = fma(c, w1, t);
                    re[j + 8] = fma(-c, w1, t);
                    const double t2 = c * w0, u2 = (double)n2[j] + (double)e2[j];
                    im[j] = fma(u2, w1, t2);
                    im[j + 8] = fma(-u2, w1, t2);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const double w0 = wrow[j], w1 = wrow[j + 8];
                    const double c = (double)n1[j];
                    const double t = (double)n0[j] * w0;
                    re[j] = fma(c, w1, t);
                    re[j + 8] = fma(-c, w1, t);
                    const double t2 = c * w0, u2 = (double)n2[j];
                    im[j] = fma(u2, w1, t2);
                    im[j + 8] = fma(-u2, w1, t2);
                }
            }
            if (__builtin_expect(!pair, 0)) {  // a wavefront-uniform BRANCH (the asm keeps hipcc from turning it into 32 v_cndmask per microphone)
#ifndef BF_PAIR_CNDMASK
                asm volatile("" ::: "memory");
#endif
#pragma unroll
                for (int j = 0; j < 16; ++j) im[j] = 0.0;
            }
            // the next microphone, or the first one of this wavefront's next pair (none left: this pair's first again, unused)
            if (k + 1 < NT) {
                if constexpr (LAYOUT == 1) request_ring_mic(curR, a.slot_mic[k + 1]);
                else request_pair_mic(stream, tA, a.slot_mic[k + 1]);
            } else {
                nxt = draw_pair(s_work, sc, lane, cur.d);
                if constexpr (LAYOUT == 0) {
                    if (nxt.have) request_pair_mic(nxt.d.x, nxt.d.y + 2L * nxt.pos, a.slot_mic[0]);
                    else request_pair_mic(stream, tA, a.slot_mic[0]);
                }  // (LAYOUT 1: the next pair's hops are transposed, and then requested, behind this pair's epilogue)
            }
            cx<double> tw[15];
            BF_STAGE();
            load_tw1<1, 9>(tw, s_tw1, lane);
            BF_STAGE();
            fft16_core<double, -1, true, 1>(re, im);
            BF_STAGE();
            load_tw1<9, 16>(tw, s_tw1, lane);
            BF_STAGE();
            mul_tw<false, 1, 9>(re, im, tw);
            BF_STAGE();
            mul_tw<false, 9, 16>(re, im, tw);
            BF_STAGE();
            T1_fwd(re, im, wcol, row16);
            load_tw2<1, 9>(tw, s_tw2, lane);
            BF_STAGE();
            fft16_core<double, -1, true>(re, im);
            BF_STAGE();
            load_tw2<9, 16>(tw, s_tw2, lane);
            BF_STAGE();
            mul_tw<false, 1, 9>(re, im, tw);
            BF_STAGE();
            mul_tw<false, 9, 16>(re, im, tw);
            BF_STAGE();
            w64_T2_any<true>(re, im, row16 - 16 * (lane >> 4), lane >> 4);
            cxa g[16];
            const lds_gain_t gd = (lds_gain_t)(size_t)(gd0 + (unsigned)(k * kGMic * 16));  // k3 < 2: row (g, k3), column lane
            const lds_gain_t gm = (lds_gain_t)(size_t)((g_mirror + (unsigned)(2 * k * kGMic * 16)) - (gd0 + (unsigned)(k * kGMic * 16)));  // k3 >= 2: row (3 - g, 3 - k3), column 64 - lane, conjugated
            BF_STAGE();
#pragma unroll
            for (int r = 0; r < 8; ++r)
                g[r] = (r & 3) < 2 ? gd[(2 * (r >> 2) + (r & 3)) * kGRow] : gm[(2 * (3 - (r >> 2)) + 3 - (r & 3)) * kGRow];
            BF_STAGE();
            w64_fwd_p3<double>(re, im);
            BF_STAGE();
#pragma unroll
            for (int r = 8; r < 16; ++r)
                g[r] = (r & 3) < 2 ? gd[(2 * (r >> 2) + (r & 3)) * kGRow] : gm[(2 * (3 - (r >> 2)) + 3 - (r & 3)) * kGRow];
            BF_STAGE();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r == 8) BF_STAGE();
                const double gx = g[r].x, gy = (r & 3) < 2 ? g[r].y : -g[r].y;
                if constexpr (FIRST) {  // the first microphone starts the sum
                    Sr[r] = fma(-gy, im[r], gx * re[r]);
                    Si[r] = fma(gy, re[r], gx * im[r]);
                } else {
                    Sr[r] = fma(-gy, im[r], fma(gx, re[r], Sr[r]));
                    Si[r] = fma(gy, re[r], fma(gx, im[r], Si[r]));
                }
            }
        };
        one_mic(0, std::true_type{});
        for (int k = 1; k < NT; ++k) one_mic(k, std::false_type{});
        float *yo = ys + tA * kHop;
        cx<double> tw[15];
        BF_STAGE();
        {  // microphone 0's hops tA - 1, tA, tA + 1: in flight during the backward transform
            const float *h0 = LAYOUT == 1 ? curR.h : (first_hop ? hist0 : hp0 - kHop), *h1 = LAYOUT == 1 ? curR.a : hp0;
            const float *h2 = LAYOUT == 1 ? curR.b : (last_hop ? hp0 : hp0 + kHop);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                q0[j] = h0[(unsigned)(64 * j + lane)];
                q1[j] = h1[(unsigned)(64 * j + lane)];
                q2[j] = h2[(unsigned)(64 * j + lane)];
            }
            if constexpr (LAYOUT == 0) request_extra(XM >= 0 && nxt.have, nxt.d.x, nxt.d.y + 2L * nxt.pos);  // ... and the next pair's second slot-0 microphone
        }
        BF_STAGE();
        load_tw2<1, 16>(tw, s_tw2, lane);
        BF_STAGE();
        w64_inv_p3<double>(Sr, Si);
        w64_T2_any<false>(Sr, Si, row16 - 16 * (lane >> 4), lane >> 4);
        BF_STAGE();
        mul_tw<true, 1, 16>(Sr, Si, tw);
        BF_STAGE();
        load_tw1<1, 16>(tw, s_tw1, lane);
        BF_STAGE();
        fft16_core<double, +1, false>(Sr, Si);
        BF_STAGE();
        T1_inv(Sr, Si, row16, wcol);
        BF_STAGE();
        mul_tw<true, 1, 16>(Sr, Si, tw);
        fft16_core<double, +1, false>(Sr, Si);

        // register j: sample n = 64 j + lane of frame tA (real part) and of frame tA + 1 (imaginary part); util.h:247-252, float stores
        // Overlap-add across wavefronts, first come first served: the hop between two frame pairs is the float sum of the second half of
        // the earlier pair's last frame and the first half of the later pair's first frame (util.h:302; a + b == b + a bit for bit).
        // Whichever side gets there first claims the boundary (LDS atomic), parks its half in the output hop itself and publishes it
        // once the stores are acknowledged; the other side finds the claim, reads the hop back, adds its half and stores the hop for
        // good.  Nobody waits for anybody's arithmetic -- at most for a store acknowledgement when both arrive within a microsecond.
        // microphone 0: h[n] x_0[n] / M joins Re / N in double (frame tA in the real, frame tA + 1 in the imaginary part; a lone frame's
        // imaginary part is not used)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double w0 = wrow[j], w1 = wrow[j + 8];
            const double c = (double)q1[j];
            Sr[j] = fma((double)q0[j] * w0, inv_m, Sr[j]);
            Sr[j + 8] = fma(c * w1, inv_m, Sr[j + 8]);
            Si[j] = fma(c * w0, inv_m, Si[j]);
            Si[j + 8] = fma((double)q2[j] * w1, inv_m, Si[j + 8]);
        }
        float oA[16], tl[8];
        if (pair) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float f1 = (float)Si[j + 8];
                tl[j] = (float)((double)f1 * wrow[j + 8]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float f1 = (float)Sr[j + 8];
                tl[j] = (float)((double)f1 * wrow[j + 8]);
            }
        }
        float *yn = ys + (tL + 1) * kHop;
        int oT = -1, oH = -1;                            // what the claim found: 0 = nobody yet
        if (has_t) {
            oT = claim_boundary(s_state, u, lane);
            if (oT == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) yn[(unsigned)(64 * j + lane)] = tl[j];
            }
        } else {
            skip_boundary(s_state, u, lane);             // the chunk ends here: nobody else will touch this slot's generation
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float f = (float)Sr[j];               // (float)(Re / N): 1/N is inside the gains
            oA[j] = (float)((double)f * wrow[j]);       // o *= hann_win[n]
        }
        if (pair) {
            float *y1 = ys + (tA + 1) * kHop;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float f0 = (float)Si[j];
                const float b0 = (float)((double)f0 * wrow[j]);
                y1[(unsigned)(64 * j + lane)] = oA[j + 8] + b0;  // out_buff[0][j] + out_buff[1][j] as floats (util.h:302), both halves in this lane
            }
        }
        if (has_h) {
            oH = claim_boundary(s_state, u - 1, lane);
            if (oH == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) yo[(unsigned)(64 * j + lane)] = oA[j];
            }
        } else if (T0 == 0) {  // stream start: the partner is the carried state (out_buff[0] of the previous call)
            const float *ti = a.tail_in + (long)stream * kHop;
#pragma unroll
            for (int j = 0; j < 8; ++j) yo[(unsigned)(64 * j + lane)] = ti[(unsigned)(64 * j + lane)] + oA[j];
        } else {               // first hop of a chunk: the previous chunk adds its half separately, both into a hop zeroed beforehand
#pragma unroll
            for (int j = 0; j < 8; ++j) atomicAdd(yo + (unsigned)(64 * j + lane), oA[j]);
        }
        if (oT == 0 || oH == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the parked halves have reached L2 (one CU: one L1) before anyone is told
            if (oT == 0) publish_boundary(s_state, u, lane);
            if (oH == 0) publish_boundary(s_state, u - 1, lane);
        }
        if (oT > 0) {
            BF_STAT(2);
            BF_STAT_WAIT_BEGIN;
            await_boundary(s_state, u);
            BF_STAT_WAIT_END;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = BF_PARK_LD(yn + (unsigned)(64 * j + lane));
#pragma unroll
            for (int j = 0; j < 8; ++j) yn[(unsigned)(64 * j + lane)] = tl[j] + o[j];
        } else {
            BF_STAT(0);
        }
        if (oH > 0) {
            BF_STAT(2);
            BF_STAT_WAIT_BEGIN;
            await_boundary(s_state, u - 1);
            BF_STAT_WAIT_END;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = BF_PARK_LD(yo + (unsigned)(64 * j + lane));
#pragma unroll
            for (int j = 0; j < 8; ++j) yo[(unsigned)(64 * j + lane)] = o[j] + oA[j];
        } else {
            BF_STAT(1);
        }
        if (!has_t) {  // the chunk's last frame: its second half belongs to the next chunk's first hop -- or to the next call
            if (T1 < a.n_frames) {
                float *ynx = ys + T1 * kHop;
#pragma unroll
                for (int j = 0; j < 8; ++j) atomicAdd(ynx + (unsigned)(64 * j + lane), tl[j]);
            } else {
                float *to = a.tail_out + (long)stream * kHop;
#pragma unroll
                for (int j = 0; j < 8; ++j) to[(unsigned)(64 * j + lane)] = tl[j];
                // ... and the ring-buffer carry (util.h:305-308): the last input hop, into the OTHER hist buffer (frame 0 of this launch may
                // still be reading the current one)
                if (a.hist_out != nullptr) {
                    float *ho = a.hist_out + (long)stream * M * kHop;
                    const float *xs = a.x + (long)stream * a.stream_stride_x;
                    for (int m = 0; m < M; ++m) {
                        const float *xl = xs + (long)m * a.mic_stride + tL * kHop;
#pragma unroll
                        for (int j = 0; j < 8; ++j) ho[m * kHop + 64 * j + lane] = xl[(unsigned)(64 * j + lane)];
                    }
                }
            }
        }
        if constexpr (LAYOUT == 1) {
            // every load of this pair's hops has been consumed -- their values are in the output stores above, which the compiler may not move
            // across this barrier -- so the slots may be handed on; the next pair's hops go in at the top of the loop
            asm volatile("" ::: "memory");
            release_pair(cur, has_t);
        }
        cur = nxt;
        ++it;
    }
    BF_STATS_FLUSH;
#ifdef BF_W64_STAMPS  // when each wavefront left the kernel and how many frame pairs it took (tools/finish_hist.py)
    if (lane == 0 && blockIdx.x < 256) {
        g_stamps[(blockIdx.x * 8 + w) * 64 + 63] = __builtin_amdgcn_s_memrealtime();
        g_stamps[(blockIdx.x * 8 + w) * 64 + 60] = __builtin_amdgcn_s_memtime();
        g_stamps[(blockIdx.x * 8 + w) * 64 + 62] = (unsigned long long)it;
        g_stamps[(blockIdx.x * 8 + w) * 64 + 61] = (unsigned long long)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
    }
#endif
    (void)it;
}

// (two kernels, one body: the planar kernel keeps its name in every profile and traffic file)
__global__ __launch_bounds__(kBlock) void das_f64_pair_kernel(DasF64Args a, DasSched sc) { das_f64_pair_body<0>(a, sc); }
__global__ __launch_bounds__(kBlock) void das_f64_ring_kernel(DasF64Args a, DasSched sc) { das_f64_pair_body<1>(a, sc); }

// The chunk table of one launch + the zeroing of every hop that two chunks complete by atomic adds + the reset of the counter: one
// small launch in front of the kernel (it replaces the hipMemset2DAsync of the static-run version).  32 threads per chunk.
// Levels: per stream, level i holds cnt[i] chunks of size[i] pairs (the last chunk of a stream may be shorter); level-major order over
// all streams, so the table starts with every block's big first chunk and ends with the small ones that level the finishing times.
// (eight chunks per 256-thread block, 32 threads each: 2 304 blocks of 128 threads took 5.3 us at the headline size, most of it dispatch)
constexpr int kSchedPerBlock = 8;
__global__ __launch_bounds__(32 * kSchedPerBlock) void das_f64_sched_kernel(DasSchedPlan p, int4 *chunks, unsigned *counter, float *y, long n_frames, int n_streams) {
    const int k = blockIdx.x * kSchedPerBlock + (int)(threadIdx.x >> 5), l = (int)(threadIdx.x & 31);
    if (k >= p.n_chunks) return;
    int stream;
    long t0, n;
    das_f64_chunk(p, n_frames, n_streams, k, &stream, &t0, &n);
    if (l == 0) {
        chunks[k] = int4{stream, (int)t0, (int)n, 0};
        if (k == 0) *counter = (unsigned)p.grid;
    }
    if (t0 > 0) {
        float4 *h = reinterpret_cast<float4 *>(y + ((long)stream * n_frames + t0) * kHop);
        for (int i = l; i < kHop / 4; i += 32) h[i] = float4{0.f, 0.f, 0.f, 0.f};
    }
}

// [sample][mic] -> planar, in front of das_f64_pair_kernel (round 6).  The pair kernel walks ONE microphone at a time, which on
// [sample][mic] input makes every 128-byte line cross L2 -> L1 eight times (1.34 ms, EXPERIMENTS round 5); das_f64_w64_kernel<1> (two
// microphones per transform, 5 transforms per frame, no microphone-0 / identical-row savings) takes 1.03 ms.  A plain transposition through
// LDS -- 256 samples x M microphones per block, 16-byte accesses on both sides -- moves 2 x the input once (1.07 GB in, 1.07 GB out at
// 8 microphones) and hands the pair kernel what it is fast on.  Tile rows are padded to M + 1 floats (bank conflicts of the gather).
__global__ __launch_bounds__(256) void interleaved_to_planar_kernel(const float *x, float *out, long tiles_per_stream, int M, long in_stream_stride,
                                                                    long out_mic_stride, long out_stream_stride) {
    __shared__ float s_t[256 * 9];
    const int tid = threadIdx.x;
    const long s = blockIdx.x / tiles_per_stream, tile = blockIdx.x - s * tiles_per_stream;
    const float4 *src = reinterpret_cast<const float4 *>(x + s * in_stream_stride + tile * 256 * M);
    const int R = M + 1;
    for (int i = tid; i < 64 * M; i += 256) {  // 256 M floats, 16 bytes per lane, contiguous
        const float4 v = src[i];
        const int e = 4 * i;
        s_t[(e / M) * R + e % M] = v.x;
        s_t[((e + 1) / M) * R + (e + 1) % M] = v.y;
        s_t[((e + 2) / M) * R + (e + 2) % M] = v.z;
        s_t[((e + 3) / M) * R + (e + 3) % M] = v.w;
    }
    __syncthreads();
    float *dst = out + s * out_stream_stride + tile * 256;
    for (int j = tid; j < 64 * M; j += 256) {  // per microphone 64 float4 = 256 consecutive samples
        const int m = j >> 6, q = j & 63;
        const float4 v{s_t[(4 * q) * R + m], s_t[(4 * q + 1) * R + m], s_t[(4 * q + 2) * R + m], s_t[(4 * q + 3) * R + m]};
        reinterpret_cast<float4 *>(dst + (long)m * out_mic_stride)[q] = v;
    }
}

}  // namespace

// planar input: the frame-pair kernel; [sample][mic] input: the microphone-pair kernel
// (the frame-pair kernel never transforms microphone 0: it needs the reference's unit weight row there -- das.cpp:33-38, always true for das
// on a handle that started cold -- and a second microphone; anything else goes through the chain)
static bool ring_mics(int m) { return m == 2 || m == 4 || m == 8; }  // (the ring's transposition addresses by shifts)
static bool use_pair_kernel(const DasF64Args &a) {
    if (!(a.gains_mic != nullptr && a.sched_ws != nullptr && a.mic0_unit != 0 && a.n_mics >= 2 && a.n_tr >= 1)) return false;
    if (a.layout == 0) return true;
    return a.ring != nullptr && ring_mics(a.n_mics) && a.hist_out == nullptr;  // [sample][mic]: through the blocks' hop rings
}
size_t das_f64_ring_bytes(int n_mics, int n_cus) {
    return ring_mics(n_mics) ? (size_t)n_cus * kRingSlots * n_mics * kHop * sizeof(float) : 0;
}

bool das_f64_writes_hist(const DasF64Args &a) { return use_pair_kernel(a) && a.hist_out != nullptr; }

// ---- the frame-pair kernel's work queue ------------------------------------------------------------------------------------------
// Per stream: nb = blocks per stream (n_cus / n_streams, at least 1).  Level 0 gives every block one long chunk (kSchedFirst of its
// equal share: consecutive pairs on one CU share their input hop through L1 / L2 and hand over their output hop through LDS flags),
// the following levels halve the chunk until kSchedLast pairs; what is left goes out in chunks of kSchedLast pairs.  A chunk edge costs
// one input hop read twice and two atomic adds per output sample, so the small chunks are kept to the last ~12 % of the batch.
// BF_DAS_F64_SCHED=0: one level of equal chunks (the static runs of round 4); BF_DAS_F64_SCHED="88,16,8,4,2": explicit chunk sizes in pairs.
constexpr size_t kSchedCounterBytes = 256;
size_t das_f64_sched_ws_bytes() { return kSchedCounterBytes + (size_t)kSchedMaxChunks * sizeof(int4); }

// frames per run of the microphone-pair kernel: a multiple of one step of the block (8 frames), about one run per CU
static void das_f64_w64_runs(const DasF64Args &a, int n_cus, long *fpc, long *cps) {
    const long step = kWaves;
    long runs = (long)n_cus / a.n_streams;
    if (runs < 1) runs = 1;
    long f = (a.n_frames + runs - 1) / runs;
    f = ((f + step - 1) / step) * step;
    *fpc = f;
    *cps = (a.n_frames + f - 1) / f;
}

// what has to happen on `s` before the kernel: the frame-pair kernel's chunk table, counter and zeroed chunk-boundary hops (one small
// launch); the microphone-pair kernel: the first hop of every run but the first of a stream zeroed (it is completed by atomic adds)
hipError_t prepare_das_f64_w64(const DasF64Args &a, int n_cus, hipStream_t s) {
    if (a.n_mics > 8) return hipErrorNotSupported;  // the gain tables fill the LDS
    if (use_pair_kernel(a)) {
        if (a.sched_ws_bytes < das_f64_sched_ws_bytes() || (long)a.n_streams * ((a.n_frames + 1) / 2) >= (1L << 31)) return hipErrorNotSupported;
        const DasSchedPlan p = das_f64_plan(a.n_frames, a.n_streams, n_cus, getenv("BF_DAS_F64_SCHED"));
        if (p.n_chunks < 1 || p.n_chunks > kSchedMaxChunks) return hipErrorNotSupported;  // (more streams than the table has rows: the chain serves them)
        unsigned *counter = reinterpret_cast<unsigned *>(a.sched_ws);
        int4 *chunks = reinterpret_cast<int4 *>(reinterpret_cast<char *>(a.sched_ws) + kSchedCounterBytes);
        BF_LAUNCH(das_f64_sched_kernel, dim3((unsigned)((p.n_chunks + kSchedPerBlock - 1) / kSchedPerBlock)), dim3(32 * kSchedPerBlock), 0, s, p, chunks, counter, a.y, a.n_frames, a.n_streams);
        return hipGetLastError();
    }
    if (a.layout == 0) return hipErrorNotSupported;  // (planar input without the pair kernel's tables or with a non-unit row 0: the chain serves it)
    long fpc, cps;
    das_f64_w64_runs(a, n_cus, &fpc, &cps);
    if (cps > 1)
        for (int st = 0; st < a.n_streams; ++st) {
            hipError_t e = hipMemset2DAsync(a.y + ((long)st * a.n_frames + fpc) * kHop, (size_t)fpc * kHop * sizeof(float), 0,
                                            kHop * sizeof(float), (size_t)cps - 1, s);
            if (e != hipSuccess) return e;
        }
    return hipSuccess;
}

// x = [stream][n][M] -> out = [stream][M][n] (n a multiple of 256 samples, M <= 8, both 16-byte aligned)
hipError_t launch_interleaved_to_planar(const float *x, float *out, long n, int n_mics, int n_streams, hipStream_t s) {
    if (n_mics < 1 || n_mics > 8 || (n & 255) != 0) return hipErrorNotSupported;
    const long tiles = n / 256;
    BF_LAUNCH(interleaved_to_planar_kernel, dim3((unsigned)(tiles * n_streams)), dim3(256), 0, s, x, out, tiles, n_mics, (long)n_mics * n, n, (long)n_mics * n);
    return hipGetLastError();
}

hipError_t launch_das_f64_w64(const DasF64Args &a, int n_cus, hipStream_t s) {
    if (a.n_mics > 8) return hipErrorNotSupported;
    if (use_pair_kernel(a)) {
        const DasSchedPlan p = das_f64_plan(a.n_frames, a.n_streams, n_cus, getenv("BF_DAS_F64_SCHED"));
        DasSched sc;
        sc.counter = reinterpret_cast<unsigned *>(a.sched_ws);
        sc.chunks = reinterpret_cast<const int4 *>(reinterpret_cast<char *>(a.sched_ws) + kSchedCounterBytes);
        sc.n_chunks = p.n_chunks;
        if (a.layout == 0) {
            BF_LAUNCH(das_f64_pair_kernel, dim3((unsigned)p.grid), dim3(kBlock), 0, s, a, sc);
        } else {
            if (a.ring_bytes < das_f64_ring_bytes(a.n_mics, p.grid)) return hipErrorNotSupported;
            BF_LAUNCH(das_f64_ring_kernel, dim3((unsigned)p.grid), dim3(kBlock), 0, s, a, sc);
        }
        return hipGetLastError();
    }
    long fpc, cps;
    das_f64_w64_runs(a, n_cus, &fpc, &cps);
    if (a.layout == 0) return hipErrorNotSupported;  // (planar input without the pair kernel's tables: the chain serves it)
    BF_LAUNCH(das_f64_w64_kernel<1>, dim3((unsigned)(cps * a.n_streams)), dim3(kBlock), 0, s, a, (int)fpc, (int)cps);
    return hipGetLastError();
}

}  // namespace bf

#ifdef BF_W64_STATS
extern "C" int bf_dbg_stats(unsigned long long *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(bf::g_stats), sizeof(unsigned long long) * 256 * 8 * 5);
}
#endif
#ifdef BF_W64_STAMPS
extern "C" int bf_dbg_stamps(unsigned long long *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(bf::g_stamps), sizeof(unsigned long long) * 256 * 8 * 64);
}
#endif
