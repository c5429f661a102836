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
//   ring advance / buffer swap     util.h:305-313   last hop + tail written back as carried state
//
// Mapping: a 32-lane half-wavefront owns a run of consecutive frames of one
// stream; each lane holds 32 complex points in VGPRs (fft1024.hpp).  The two
// halves of a wavefront work on different frame runs, so no cross-lane traffic
// exists outside the per-FFT LDS transpose and there is no workgroup barrier in
// the main loop.  The first frame of every run is recomputed (not stored) to
// obtain the overlap tail; run 0 of a stream takes it from the carried state.
//
// LDS (one 512-thread block per CU, 16 half-wavefronts):
//   8 KiB inter-pass twiddles + 16 x 4.5 KiB plane-split transpose buffers
//   + 4.5 KiB window ([lane][j]) + 8 KiB x ceil(M/2) pair gains  (118 KiB at M = 8)
// so the only global traffic in the loop is the input stream and the output hop.
//
// Bound: HBM stream of M*H*4 B in + H*4 B out per frame; at >= 40 % of 8 TB/s the
// 5 FFT-1024 per frame need about half of the fp32 VALU peak.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "launch_trace.hpp"
#include "fft1024.hpp"
#include "kernels.hpp"

namespace bf {

namespace {

// Debug build only (-DBF_DAS_STAMPS): per-phase wall-clock sums of the unrolled scalar kernel, printed after every launch.
#ifdef BF_DAS_STAMPS
constexpr int kStampBlocks = 4096;
__device__ unsigned long long g_stamps[kStampBlocks][20];  // per block (wave 0 reports; plain stores, summed on the host)
#if BF_DAS_STAMPS >= 2
#define BF_STAMP(slot)                                                                  \
    do {                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                              \
        const unsigned long long now_ = __builtin_readcyclecounter();                   \
        st_acc[slot] += now_ - st_prev;                                                 \
        st_prev = now_;                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                              \
    } while (0)
#else  // 1: whole-loop cycles and clock only (no stamp inside the loop)
#define BF_STAMP(slot) ((void)0)
#endif
#define BF_STAMP_PARAMS , unsigned long long (&st_acc)[16], unsigned long long &st_prev
#define BF_STAMP_ARGS , st_acc, st_prev
#else
#define BF_STAMP(slot) ((void)0)
#define BF_STAMP_PARAMS
#define BF_STAMP_ARGS
#endif
#ifndef BF_DAS_CHUNK
#define BF_DAS_CHUNK 8
#endif
constexpr int kBlock = 512;
constexpr int kHalves = kBlock / 32;
constexpr int kPS = plane_stride<float>::value;  // 36
constexpr int kHop = 512;
constexpr int kNfft = 1024;
constexpr int kLdsTw = 2048;                 // floats
constexpr int kLdsPlane = 32 * kPS;          // floats per half-wavefront
constexpr int kLdsWin = 32 * kPS;            // floats
constexpr int kLdsFixed = kLdsTw + kHalves * kLdsPlane + kLdsWin;


// ---- wave-interleaved transposes (WT variant) ---------------------------------------------------------------------------
// The plane-split transpose of fft1024.hpp writes one dword per lane with ds_write(2)_b32 (64 B/clk/CU) -- 58 % of this
// kernel's LDS cycles.  ds_write_addtid_b32 (address = M0 + offset + 4*lane, no address VGPR) stores at twice that rate, but
// its address pattern is fixed: 64 consecutive dwords per instruction.  So the two half-wavefronts of a wavefront share one
// plane whose rows are [half 0: 32 floats | half 1: 32 floats | 4 pad] (272 B): one store per register position writes the
// same row of both halves; lane l of half h then reads row l, columns 32 h .. 32 h + 31, as eight ds_read_b128 (the 16-lane
// read groups stay inside one half, lanes 4 banks apart: conflict-free).  Same arithmetic, bit-identical output.
constexpr int kWRow = 68;                 // floats per row of the shared plane
constexpr int kWPlane = 32 * kWRow;       // floats per wavefront

template <bool NATURAL>
__device__ constexpr int wt_row_off(int i) { return (NATURAL ? i : brev5(i)) * kWRow * 4; }

// eight register positions I0 .. I0+7 of `v` -> rows of the plane at LDS byte address `base` (wave-uniform).
// M0 is compiler-reserved: saved and restored inside the statement that uses it.
template <int I0, bool NATURAL>
__device__ __forceinline__ void wt_store8(const float (&v)[32], unsigned base) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %9\n\ts_nop 0\n\t"
        "ds_write_addtid_b32 %1 offset:%c10\n\tds_write_addtid_b32 %2 offset:%c11\n\t"
        "ds_write_addtid_b32 %3 offset:%c12\n\tds_write_addtid_b32 %4 offset:%c13\n\t"
        "ds_write_addtid_b32 %5 offset:%c14\n\tds_write_addtid_b32 %6 offset:%c15\n\t"
        "ds_write_addtid_b32 %7 offset:%c16\n\tds_write_addtid_b32 %8 offset:%c17\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(v[I0]), "v"(v[I0 + 1]), "v"(v[I0 + 2]), "v"(v[I0 + 3]), "v"(v[I0 + 4]), "v"(v[I0 + 5]), "v"(v[I0 + 6]),
          "v"(v[I0 + 7]), "s"(base), "i"(wt_row_off<NATURAL>(I0)), "i"(wt_row_off<NATURAL>(I0 + 1)),
          "i"(wt_row_off<NATURAL>(I0 + 2)), "i"(wt_row_off<NATURAL>(I0 + 3)), "i"(wt_row_off<NATURAL>(I0 + 4)),
          "i"(wt_row_off<NATURAL>(I0 + 5)), "i"(wt_row_off<NATURAL>(I0 + 6)), "i"(wt_row_off<NATURAL>(I0 + 7))
        : "memory");
}
template <bool NATURAL>
__device__ __forceinline__ void wt_store_plane(const float (&v)[32], unsigned base) {
    wt_store8<0, NATURAL>(v, base);
    wt_store8<8, NATURAL>(v, base);
    wt_store8<16, NATURAL>(v, base);
    wt_store8<24, NATURAL>(v, base);
}
// row `lane` of this half: rp = plane + lane * kWRow + 32 * half.  LDS operations of one wavefront execute in issue order,
// so the reads see the stores above without a wait (the "memory" clobber keeps the compiler from moving them).
__device__ __forceinline__ void wt_load_row(float (&v)[32], const float *rp) {
    const float4 *r4 = reinterpret_cast<const float4 *>(rp);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const float4 q = r4[g];
        v[4 * g + 0] = q.x; v[4 * g + 1] = q.y; v[4 * g + 2] = q.z; v[4 * g + 3] = q.w;
    }
}
//
// R > 1 (frame groups, see das_fused_kernel): the lanes of a half-wavefront carry the 32 residues n mod 32 in the order perm_lane<R> --
// physical lane p holds logical lane (p mod (32 / R)) R + p / (32 / R), so that the 32 / R lanes of one frame sit side by side and a load
// instruction touches one run of consecutive samples per frame.  The transform does not care which physical lane holds a residue: the
// first pass is per lane (its twiddles are looked up under the logical lane, `lane` below), and the plane transpose turns the physical
// lane into the register index -- a renaming of registers at compile time, no instruction.
template <int R>
__device__ constexpr int perm_lane(int p) { return (p % (32 / R)) * R + p / (32 / R); }
template <int R>
__device__ __forceinline__ void perm_regs_fwd(float (&v)[32]) {  // v[p] (column p of the plane: physical lane p) -> position perm_lane(p)
    if constexpr (R > 1) {
        float t[32];
#pragma unroll
        for (int p = 0; p < 32; ++p) t[perm_lane<R>(p)] = v[p];
#pragma unroll
        for (int p = 0; p < 32; ++p) v[p] = t[p];
    }
}
// forward / backward FFT-1024 of the half-wavefront's 32 x 32 points (same arithmetic as fft1024p_{fwd,inv}_A .. D)
template <int R = 1>
__device__ __forceinline__ void wt_fft_fwd(float (&re)[32], float (&im)[32], int lane, const cx<float> *tw, unsigned base,
                                           const float *rp) {
    fft32_dif<float, -1>(re, im);
#pragma unroll
    for (int i = 1; i < 32; ++i) {
        const cx<float> w = tw[brev5(i) * 32 + lane];
        const float xr = re[i], xi = im[i];
        re[i] = xr * w.x - xi * w.y;
        im[i] = xr * w.y + xi * w.x;
    }
    wt_store_plane<false>(re, base);
    wt_load_row(re, rp);
    wt_store_plane<false>(im, base);
    wt_load_row(im, rp);
    perm_regs_fwd<R>(re);
    perm_regs_fwd<R>(im);
    fft32_dif<float, -1>(re, im);
}
template <int R = 1>
__device__ __forceinline__ void wt_fft_inv(float (&re)[32], float (&im)[32], int lane, const cx<float> *tw, unsigned base,
                                           const float *rp) {
    fft32_dit<float, +1>(re, im);
#pragma unroll
    for (int n2 = 1; n2 < 32; ++n2) {
        const cx<float> w = tw[n2 * 32 + lane];
        const float xr = re[n2], xi = im[n2];
        re[n2] = xr * w.x + xi * w.y;
        im[n2] = xi * w.x - xr * w.y;
    }
    if constexpr (R > 1) {  // row i of the plane receives register perm_lane(i): see wt_fft_inv_p2
        float t[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) t[i] = re[perm_lane<R>(i)];
        wt_store_plane<true>(t, base);
        wt_load_row(re, rp);
#pragma unroll
        for (int i = 0; i < 32; ++i) t[i] = im[perm_lane<R>(i)];
        wt_store_plane<true>(t, base);
        wt_load_row(im, rp);
    } else {
        wt_store_plane<true>(re, base);
        wt_load_row(re, rp);
        wt_store_plane<true>(im, base);
        wt_load_row(im, rp);
    }
    fft32_dif<float, +1>(re, im);
}

// ---- paired tables (unrolled kernels) -------------------------------------------------------------------------------------
// hipcc merges two ds_read_b64 of one table (twiddle rows k1 and k1', gains of positions i and i') into one ds_read2_b64, which
// the LDS serves at HALF the rate of the two separate reads (8 cycles per 1 KiB instead of 2 x 2: MI355X guide, LDS table).
// So the unrolled kernels re-pack both tables while filling the LDS: two entries a lane needs back to back become 16
// contiguous bytes -> one ds_read_b128 at the full 256 B/clk.
//   twiddles: [k < 16][lane][2] = { W1024^(k lane), W1024^((k + 16) lane) }   (forward: positions brev5(k), brev5(k) + 1;
//                                                                             backward: positions k, k + 16)
//   gains:    [pair][m < 16][lane][2] = { D[2m][lane], D[2m + 1][lane] }
// SPLIT: the second 32-point pass stops after its stages 0..2 (fft32_dif_head); the caller finishes it four positions at a time
// with fft32_dif_tail and takes every finished group straight into the weight-and-sum
template <bool SPLIT = false, int R = 1>
__device__ __forceinline__ void wt_fft_fwd_p2(float (&re)[32], float (&im)[32], int lane, const float4 *tw2, unsigned base,
                                              const float *rp BF_STAMP_PARAMS) {
    fft32_dif<float, -1>(re, im);
    BF_STAMP(1);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float4 w = tw2[k * 32 + lane];
        const int i = brev5(k);  // even position: k1 = k; i + 1: k1 = k + 16
        if (k > 0) {
            const float xr = re[i], xi = im[i];
            re[i] = xr * w.x - xi * w.y;
            im[i] = xr * w.y + xi * w.x;
        }
        const float yr = re[i + 1], yi = im[i + 1];
        re[i + 1] = yr * w.z - yi * w.w;
        im[i + 1] = yr * w.w + yi * w.z;
    }
    BF_STAMP(2);
    wt_store_plane<false>(re, base);
    wt_load_row(re, rp);
    wt_store_plane<false>(im, base);
    wt_load_row(im, rp);
    perm_regs_fwd<R>(re);
    perm_regs_fwd<R>(im);
    BF_STAMP(3);
    if (SPLIT) {
#pragma unroll
        for (int g = 0; g < 4; ++g) fft32_dif_head<float, -1>(re, im, g);
    } else {
        fft32_dif<float, -1>(re, im);
    }
    BF_STAMP(4);
}
// R > 1: row i of the plane (the lane that reads it: physical lane i) receives register perm_lane(i) -- the output lanes carry the residues
// in the same order as the forward transform's input lanes
template <int R = 1>
__device__ __forceinline__ void wt_fft_inv_p2(float (&re)[32], float (&im)[32], int lane, const float4 *tw2, unsigned base,
                                              const float *rp BF_STAMP_PARAMS) {
    fft32_dit<float, +1>(re, im);
    BF_STAMP(6);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float4 w = tw2[k * 32 + lane];
        if (k > 0) {
            const float xr = re[k], xi = im[k];
            re[k] = xr * w.x + xi * w.y;
            im[k] = xi * w.x - xr * w.y;
        }
        const float yr = re[k + 16], yi = im[k + 16];
        re[k + 16] = yr * w.z + yi * w.w;
        im[k + 16] = yi * w.z - yr * w.w;
    }
    BF_STAMP(7);
    if constexpr (R > 1) {
        float t[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) t[i] = re[perm_lane<R>(i)];
        wt_store_plane<true>(t, base);
        wt_load_row(re, rp);
#pragma unroll
        for (int i = 0; i < 32; ++i) t[i] = im[perm_lane<R>(i)];
        wt_store_plane<true>(t, base);
        wt_load_row(im, rp);
    } else {
        wt_store_plane<true>(re, base);
        wt_load_row(re, rp);
        wt_store_plane<true>(im, base);
        wt_load_row(im, rp);
    }
    BF_STAMP(8);
    fft32_dif<float, +1>(re, im);
    BF_STAMP(9);
}

// NPL = number of pair-gain tables held in LDS (0: read gains from global memory)
//
// Work split: one 512-thread block (16 half-wavefronts) owns a run of consecutive frames of one
// stream and walks it 16 frames at a time -- half-wavefront hw transforms frame t0 + 16*it + hw.
// Neighbouring half-wavefronts therefore read the shared hop of the 50 % overlap within the same
// few microseconds (L1/L2 hit): HBM sees every input sample once (the per-run variant this
// replaces fetched every hop twice: profiles/r01_c_traffic_das8_v2kernel.json).  The overlap-add
// partner (second half of frame t-1) comes from the neighbour through a 17-slot LDS ring; only the
// first hop of a run needs the previous run's last frame, and that one hop is completed by two
// float atomic adds into a pre-zeroed hop (sum of two terms: order-independent, bit-exact).
//
// R > 1 (JACK periods 256 / 128 / 64, frames of N = 1024 / R samples): the unit of work is a GROUP of R consecutive frames
// interleaved into one 1024-point sequence (z[R m + i] = frame_{R g + i}[m]; the N-point pair gains repeated R times act on every
// frame separately -- an LTI identity: DESIGN.md 3.3).  Lane p holds frame p / (32 / R), samples (32 / R) j + p mod (32 / R): the
// lanes of a frame side by side (perm_lane above; with the frames alternating from lane to lane the texture addresser served 4x the
// cache accesses of R = 1 and was 75 % busy -- profiles/r04_e_pmc_small.txt).  A group yields 512 output samples like a frame does at
// R = 1, parks the same 512-float second half in the ring, and differs only in where the overlap-add partner sits: in the group's own
// slot, 32 / R lanes to the left, for the frames inside a group; in the previous group's slot, (R - 1) 32 / R lanes to the right, for
// its first frame; the run boundaries' atomics touch the hop of that one frame.  a.gains = das_pair_gains_interleaved
// tables, a.window = the N-point window, a.frames_per_chunk a multiple of 16 R; T0 / T1 / t below count groups.
template <int LAYOUT, int NPL, int UNR = 0, int R = 1>
__global__ __launch_bounds__(kBlock, 2) void das_fused_kernel(DasFusedArgs a) {
    constexpr int H = kHop / R, JS = 32 / R;  // hop of a frame; samples between a lane's consecutive registers
    __shared__ __attribute__((aligned(16))) float lds[kLdsFixed + NPL * 2048 + 17 * kHop + 32];
    const cx<float> *s_tw = reinterpret_cast<const cx<float> *>(lds);
    float *s_win = lds + kLdsTw + kHalves * kLdsPlane;
    const cx<float> *s_gain = reinterpret_cast<const cx<float> *>(lds + kLdsFixed);
    // 17-slot ring of frame tails: frame t (run-relative r = t - T0) parks its second half in slot (r + 1) % 17 and
    // frame t+1 picks it up there.  With 16 frames in flight the reader of a slot is always its next writer, so the
    // only hazard is read-after-write: one flag per slot (= the frame whose tail it holds) replaces block barriers.
    float *s_tails = lds + kLdsFixed + NPL * 2048;
    // LDS address space spelled out: a volatile access through a generic pointer compiles to flat_load / flat_store, whose wait is
    // vmcnt(0) -- it would drain the next frame's prefetch in front of every flag read
    volatile __attribute__((address_space(3))) int *s_flag = (volatile __attribute__((address_space(3))) int *)(lds + kLdsFixed + NPL * 2048 + 17 * kHop);
#ifdef BF_DAS_STAMPS
    const unsigned long long st_k0 = __builtin_readcyclecounter();
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 31;
    const int hw = tid >> 5;
    const int fi = lane / JS, c = lane % JS;  // frame inside the group and sample offset (R = 1: 0 and the lane)
    const int llane = perm_lane<R>(lane);     // the residue n mod 32 this lane carries (R = 1: the lane)
    // the two halves of a wavefront share one interleaved plane (8 x 8.5 KiB inside the same 72 KiB region)
    float *wplane = lds + kLdsTw + (hw >> 1) * kWPlane;
    const unsigned wbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float *)wplane);
    const float *wrowp = wplane + lane * kWRow + 32 * (hw & 1);
    const int M = a.n_mics;
    const int n_pairs = (M + 1) >> 1;

    const int stream = blockIdx.x / a.chunks_per_stream;  // output stream = input stream * n_dirs + look direction
    const long c_in_s = blockIdx.x - (long)stream * a.chunks_per_stream;
    const int in_stream = stream / a.n_dirs;
    const f32x2 *gains = a.gains + (long)(stream - in_stream * a.n_dirs) * n_pairs * 1024;
    constexpr bool kP2 = UNR > 0 && LAYOUT == 0 && NPL > 0;  // paired tables (see wt_fft_fwd_p2)
    {
        const f32x2 *twc = a.twiddle;
        f32x2 *ltw = reinterpret_cast<f32x2 *>(lds);
        for (int i = tid; i < kLdsTw / 2; i += kBlock) {  // i = k1 * 32 + lane
            const int k1 = i >> 5, l = i & 31;
            ltw[kP2 ? (((k1 & 15) * 32 + l) * 2 + (k1 >> 4)) : i] = twc[i];
        }
        for (int i = tid; i < kNfft; i += kBlock) s_win[(i & 31) * kPS + (i >> 5)] = a.window[i / R];  // [lane][j]; interleaved index i <-> sample i / R
        if (NPL > 0) {
            f32x2 *lg = reinterpret_cast<f32x2 *>(lds + kLdsFixed);
            for (int i = tid; i < n_pairs * 1024; i += kBlock) {  // i = (pair * 32 + pos) * 32 + lane
                const int l = i & 31, pos = (i >> 5) & 31, pr = i >> 10;
                lg[kP2 ? (((pr * 16 + (pos >> 1)) * 32 + l) * 2 + (pos & 1)) : i] = gains[i];
            }
        }
    }
    const long n_units = (a.n_frames + R - 1) / R;  // groups per stream (the last one may be partial)
    const long T0 = c_in_s * (a.frames_per_chunk / R);
    long T1 = T0 + a.frames_per_chunk / R;
    if (T1 > n_units) T1 = n_units;
    if (T0 == 0) {  // stream start: the overlap partner of frame 0 is the carried state (out_buff[0], util.h:302), parked as frame R - 1 of "group -1"
        for (int i = tid; i < H; i += kBlock) s_tails[32 * (i / JS) + (R - 1) * JS + (i % JS)] = a.tail_in[(long)stream * H + i];
    }
    if (tid < 17) s_flag[tid] = (tid == 0) ? (int)(T0 - 1) : -2;  // slot 0 holds "frame T0-1" (state, or unused when T0 > 0)
    __syncthreads();
    const float4 *wrow = reinterpret_cast<const float4 *>(s_win + llane * kPS);

    const float *xs = a.x + (long)in_stream * a.stream_stride_x;
    const float *hs = a.hist_in + (long)in_stream * M * H;
    float *ys = a.y + (long)stream * a.n_frames * H;
    // the frame this lane holds of unit tc (R = 1: the unit itself); past the end of the stream: the last frame again, never stored
    auto lane_frame = [&](long tc) -> long {
        const long f = tc * R + fi;
        return (R > 1 && f >= a.n_frames) ? a.n_frames - 1 : f;
    };

    float re[32], im[32], Sr[32], Si[32];
    const int n_iter = (int)((T1 - T0 + kHalves - 1) / kHalves);
#ifdef BF_DAS_STAMPS
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_prev = __builtin_readcyclecounter();
    const unsigned long long st_c0 = st_prev, st_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    // frame handled by this half-wavefront at iteration `it` (clamped into the run: invalid slots redo its last frame)
    auto frame_of = [&](int it) -> long {
        const long t = T0 + (long)it * kHalves + hw;
        return t < T1 ? t : T1 - 1;
    };
    // planar layout: raw samples of mic pair p of frame tc into re (mic 2p) / im (mic 2p+1), natural order
    // j <-> sample 32*j + lane
    auto issue_loads = [&](long tc, int p) {
        const int ma = 2 * p, mb = 2 * p + 1;
        const bool b_ok = mb < M;  // odd microphone count: the last pair's partner channel reads the zero buffer
        {
            const long f = lane_frame(tc);
            const float *a1 = (f >= 1 ? xs + (long)ma * a.mic_stride + (f - 1) * H : hs + ma * H) + c;
            const float *b1 = (!b_ok ? a.zeros : f >= 1 ? xs + (long)mb * a.mic_stride + (f - 1) * H : hs + mb * H) + c;
            const float *a2 = xs + (long)ma * a.mic_stride + f * H + c;
            const float *b2 = (!b_ok ? a.zeros : xs + (long)mb * a.mic_stride + f * H) + c;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                re[j] = a1[JS * j];
                im[j] = b1[JS * j];
                re[j + 16] = a2[JS * j];
                im[j + 16] = b2[JS * j];
            }
        }
    };

    // planar: pair 0 of a frame is requested before the previous frame's inverse FFT (one round of latency hidden);
    // interleaved: address arithmetic per sample is register-hungry, loads stay inside the round
    if (LAYOUT == 0) issue_loads(frame_of(0), 0);

    for (int it = 0; it < n_iter; ++it) {
        const long t = T0 + (long)it * kHalves + hw;
        const bool valid = t < T1;
        const long tc = valid ? t : T1 - 1;

        if constexpr (UNR > 0 && LAYOUT == 0 && NPL > 0) {
            // exact pair count, planar input: the pair loop is unrolled and the loads of pair p + 1 are issued from inside
            // pair p's gain loop, eight register positions at a time, into the registers that loop has just consumed
            auto rows = [&](int p, int i0, int n) {
                const int ma = 2 * p, mb = 2 * p + 1;
                const bool b_ok = mb < M;  // odd microphone count: the last pair's partner channel reads the zero buffer
                const long f = lane_frame(tc);
                const float *a1 = (f >= 1 ? xs + (long)ma * a.mic_stride + (f - 1) * H : hs + ma * H) + c;
                const float *b1 = (!b_ok ? a.zeros : f >= 1 ? xs + (long)mb * a.mic_stride + (f - 1) * H : hs + mb * H) + c;
                const float *a2 = xs + (long)ma * a.mic_stride + f * H + c;
                const float *b2 = (!b_ok ? a.zeros : xs + (long)mb * a.mic_stride + f * H) + c;
#pragma unroll
                for (int jx = i0; jx < i0 + n; ++jx) {
                    re[jx] = jx < 16 ? a1[JS * jx] : a2[JS * (jx - 16)];
                    im[jx] = jx < 16 ? b1[JS * jx] : b2[JS * (jx - 16)];
                }
            };
#pragma unroll
            for (int p = 0; p < UNR; ++p) {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const float4 hv = wrow[g];
                    re[4 * g + 0] *= hv.x; im[4 * g + 0] *= hv.x;
                    re[4 * g + 1] *= hv.y; im[4 * g + 1] *= hv.y;
                    re[4 * g + 2] *= hv.z; im[4 * g + 2] *= hv.z;
                    re[4 * g + 3] *= hv.w; im[4 * g + 3] *= hv.w;
                }
                BF_STAMP(0);
                wt_fft_fwd_p2<true, R>(re, im, llane, reinterpret_cast<const float4 *>(lds), wbase, wrowp BF_STAMP_ARGS);
                const float4 *gp2 = reinterpret_cast<const float4 *>(lds + kLdsFixed) + p * 512 + lane;
                // the last two stages of the second pass run four register positions at a time; every finished group goes straight
                // into the weight-and-sum and its registers to the next pair's loads, so those are spread over stages 3-4 AND the
                // weight-and-sum (same butterflies, bit-identical output; -0.4 % at 8 microphones, -1.5 % at 4).  Taking the FIRST
                // pass set by set in the order of those loads as well (so that the last-issued ones are needed last) needs a second
                // register set for the incoming pair: built and measured, 256 VGPRs + 17 scratch operations, 0.349 vs 0.334 ms.
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    fft32_dif_tail<float, -1>(re, im, m);
#pragma unroll
                    for (int i = 4 * m; i < 4 * m + 4; i += 2) {
                        const float4 g = gp2[16 * i];  // gains of positions i and i + 1
                        // accumulate with two chained FMAs per component (4 instructions per position instead of 6)
                        Sr[i] = bf_fma(-g.y, im[i], bf_fma(g.x, re[i], (p == 0) ? 0.f : Sr[i]));
                        Si[i] = bf_fma(g.y, re[i], bf_fma(g.x, im[i], (p == 0) ? 0.f : Si[i]));
                        Sr[i + 1] = bf_fma(-g.w, im[i + 1], bf_fma(g.z, re[i + 1], (p == 0) ? 0.f : Sr[i + 1]));
                        Si[i + 1] = bf_fma(g.w, re[i + 1], bf_fma(g.z, im[i + 1], (p == 0) ? 0.f : Si[i + 1]));
                    }
                    if (p + 1 < UNR) {
                        __builtin_amdgcn_sched_barrier(0);
                        rows(p + 1, 4 * m, 4);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                BF_STAMP(5);
            }
        } else
        for (int p = 0; p < n_pairs; ++p) {
            const bool b_ok = (2 * p + 1) < M;
            if (LAYOUT != 0) {
                // interleaved [sample][mic]: the two mics of a pair are adjacent -> one 8-byte load per sample
                const int ma = 2 * p;
                const int mb = b_ok ? 2 * p + 1 : ma;
                const long f = lane_frame(tc);
                const float *s1 = (f >= 1 ? xs + (f - 1) * (long)H * M : hs) + (long)c * M;
                const float *s2 = xs + f * (long)H * M + (long)c * M;
                if (b_ok && (M & 1) == 0) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const float2 u = *reinterpret_cast<const float2 *>(s1 + (long)JS * j * M + ma);
                        const float2 w = *reinterpret_cast<const float2 *>(s2 + (long)JS * j * M + ma);
                        re[j] = u.x;
                        im[j] = u.y;
                        re[j + 16] = w.x;
                        im[j + 16] = w.y;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        re[j] = s1[(long)JS * j * M + ma];
                        im[j] = b_ok ? s1[(long)JS * j * M + mb] : 0.f;
                        re[j + 16] = s2[(long)JS * j * M + ma];
                        im[j + 16] = b_ok ? s2[(long)JS * j * M + mb] : 0.f;
                    }
                }
            } else if (p > 0) {
                issue_loads(tc, p);
            }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float4 hv = wrow[g];
                re[4 * g + 0] *= hv.x; im[4 * g + 0] *= hv.x;
                re[4 * g + 1] *= hv.y; im[4 * g + 1] *= hv.y;
                re[4 * g + 2] *= hv.z; im[4 * g + 2] *= hv.z;
                re[4 * g + 3] *= hv.w; im[4 * g + 3] *= hv.w;
            }

            wt_fft_fwd<R>(re, im, llane, s_tw, wbase, wrowp);

            const cx<float> *gp = (NPL > 0 ? s_gain : reinterpret_cast<const cx<float> *>(gains)) + (long)p * 1024 + lane;
            if (p == 0) {
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const cx<float> g = gp[32 * i];
                    Sr[i] = g.x * re[i] - g.y * im[i];
                    Si[i] = g.x * im[i] + g.y * re[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const cx<float> g = gp[32 * i];
                    Sr[i] += g.x * re[i] - g.y * im[i];
                    Si[i] += g.x * im[i] + g.y * re[i];
                }
            }
        }

        // (R > 1 never gets a dump -- launch_das_fused refuses it -- but the branch stays in those instantiations: without this block boundary
        // the scheduler pulls the next unit's 64 prefetch registers up into the pair loop and the kernel spills 1.6 KB per lane)
        if (a.sdump != nullptr && valid) {
            f32x2 *sd = a.sdump + ((long)stream * a.n_frames + t) * kNfft + lane;
#pragma unroll
            for (int i = 0; i < 32; ++i) sd[32 * brev5(i)] = f32x2{Sr[i], Si[i]};
        }

        // the next frame's first pair streams in while the inverse transform runs on (Sr, Si)
        if (LAYOUT == 0 && it + 1 < n_iter) issue_loads(frame_of(it + 1), 0);
        BF_STAMP(13);

        if (kP2) {
            wt_fft_inv_p2<R>(Sr, Si, lane, reinterpret_cast<const float4 *>(lds), wbase, wrowp BF_STAMP_ARGS);
        } else {
            wt_fft_inv<R>(Sr, Si, lane, s_tw, wbase, wrowp);
        }

        // position i holds sample n = 32*brev5(i) + lane; even i -> first half, odd i -> n + 512
        float h[32];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float4 hv = wrow[g];
            h[4 * g + 0] = hv.x; h[4 * g + 1] = hv.y; h[4 * g + 2] = hv.z; h[4 * g + 3] = hv.w;
        }
        // second half of this frame -> its ring slot, then publish (LDS operations of a wavefront complete in order)
        const int r = (int)(tc - T0);
        const int my = (r + 1) % 17, pv = r % 17;
        if (valid) {
            float *my_slot = s_tails + my * kHop + lane;
#pragma unroll
            for (int q = 0; q < 16; ++q) my_slot[32 * brev5(2 * q)] = Sr[2 * q + 1] * h[brev5(2 * q + 1)];
            // data then flag: LDS operations of one wavefront execute in issue order, so a compiler barrier suffices
            asm volatile("" ::: "memory");
            if (lane == 0) s_flag[my] = (int)t;
        }
        BF_STAMP(10);
        if (R == 1 && valid) {
            float *yo = ys + t * kHop + lane;
            if (t == T0 && T0 > 0) {
                // first hop of the run: the previous run adds its half separately (both into a zeroed hop)
#pragma unroll
                for (int q = 0; q < 16; ++q) atomicAdd(yo + 32 * brev5(2 * q), Sr[2 * q] * h[brev5(2 * q)]);
            } else {
                // wait for frame t-1's tail (neighbouring half-wavefront, or the last one of the previous iteration)
                while (s_flag[pv] != (int)(t - 1)) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
                BF_STAMP(11);
                const float *prev = s_tails + pv * kHop + lane;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    // product and sum rounded separately, as the reference's two float stores (util.h:250-252, 302) and as the
                    // atomic run-boundary path does: results do not depend on how a stream is cut into batches and runs
#pragma clang fp contract(off)
                    yo[32 * brev5(2 * q)] = prev[32 * brev5(2 * q)] + Sr[2 * q] * h[brev5(2 * q)];
                }
            }
            if (t == T1 - 1) {
                if (T1 < n_units) {  // last frame of the run: its second half belongs to the next run's first hop
                    float *yn = ys + T1 * kHop + lane;
#pragma unroll
                    for (int q = 0; q < 16; ++q) atomicAdd(yn + 32 * brev5(2 * q), Sr[2 * q + 1] * h[brev5(2 * q + 1)]);
                } else {
                    // end of the batch: carried state for the next call (OLA tail and the last input hop)
                    float *to = a.tail_out + (long)stream * kHop + lane;
#pragma unroll
                    for (int q = 0; q < 16; ++q) to[32 * brev5(2 * q)] = Sr[2 * q + 1] * h[brev5(2 * q + 1)];
                    float *ho = a.hist_out + (long)in_stream * M * kHop;  // every direction writes the same values
                    if (LAYOUT == 0) {
                        for (int m = 0; m < M; ++m)
                            for (int j = 0; j < 16; ++j)
                                ho[m * kHop + 32 * j + lane] = xs[(long)m * a.mic_stride + t * kHop + 32 * j + lane];
                    } else {
                        for (int j = 0; j < 16 * M; ++j) ho[32 * j + lane] = xs[t * (long)kHop * M + 32 * j + lane];
                    }
                }
            }
        }
        if (R > 1 && valid) {
            // groups of R frames: the partner of a frame inside the group is the second half of the frame 32 / R lanes to the left, parked
            // in this group's own slot a moment ago; the partner of the group's first frame is the last frame of the previous group's slot
            const long f = t * R + fi;           // this lane's frame
            const bool f_ok = f < a.n_frames;    // the last group of a stream may be partial
            float *yo = ys + f * H + c;
            const bool run_head = t == T0 && T0 > 0;
            if (!run_head) {
                while (s_flag[pv] != (int)(t - 1)) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
            }
            const float *src = fi == 0 ? s_tails + pv * kHop + lane + (R - 1) * JS : s_tails + my * kHop + lane - JS;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
#pragma clang fp contract(off)
                const float o1 = Sr[2 * q] * h[brev5(2 * q)];
                if (run_head && fi == 0) {
                    // first hop of the run: the previous run adds its half separately (both into a zeroed hop)
                    if (f_ok) atomicAdd(yo + JS * brev5(2 * q), o1);
                } else {
                    const float partner = src[32 * brev5(2 * q)];
                    if (f_ok) yo[JS * brev5(2 * q)] = partner + o1;
                }
            }
            if (t == T1 - 1) {
                if (T1 < n_units) {  // last group of the run: its last frame's second half belongs to the next run's first hop
                    if (fi == R - 1) {
                        float *yn = ys + T1 * (long)R * H + c;
#pragma unroll
                        for (int q = 0; q < 16; ++q) atomicAdd(yn + JS * brev5(2 * q), Sr[2 * q + 1] * h[brev5(2 * q + 1)]);
                    }
                } else if (f == a.n_frames - 1) {
                    // end of the batch: carried state for the next call (OLA tail and the last input hop)
                    float *to = a.tail_out + (long)stream * H + c;
#pragma unroll
                    for (int q = 0; q < 16; ++q) to[JS * brev5(2 * q)] = Sr[2 * q + 1] * h[brev5(2 * q + 1)];
                    float *ho = a.hist_out + (long)in_stream * M * H;  // every direction writes the same values
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
        BF_STAMP(12);
#ifdef BF_DAS_STAMPS
        st_acc[15] += 1;
#endif
    }
#ifdef BF_DAS_STAMPS
    if (tid == 0 && blockIdx.x < kStampBlocks) {
        unsigned long long *g = g_stamps[blockIdx.x];
        const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 14; ++i) g[i] += st_acc[i];
        g[14] += c1 - st_c0;   // shader cycles of the main loop
        g[15] += st_acc[15];   // iterations
        g[16] += r1 - st_r0;   // 100 MHz ticks of the main loop
        g[17] += c1 - st_k0;   // kernel entry to end of loop
        g[18] = st_r0;         // absolute 100 MHz time at loop start / end of the LAST launch
        g[19] = r1;
    }
#endif
}



// ---- the 1024-frame JACK period: one FULL wavefront per 2048-sample frame on the same plane ------------------------------------------------
// N = 2048 = 32 (registers) x 64 (lanes).  The shared transpose plane above was built for two independent 1024-point transforms, one per
// half-wavefront: one store per register position writes a row [half 0: 32 columns | half 1: 32 columns], lane l of half h reads row l,
// columns 32 h .. 32 h + 31.  Read the 64 columns of a row as ONE transform's n2 = 0..63 instead and the same stores / loads are the
// transpose of a 32 x 64 decomposition; what is missing is one radix-2 stage over bit 5 of n2, between lane l of half 0 and lane l of half 1:
// v_permlane32_swap hands each half the other's value (VALU, no LDS).
//   forward:  fft32_dif over j (n = 64 j + lane64), twiddle W2048^(k1 lane64), transpose; now lane (k1, h) holds n2 = 32 h + c in register c.
//             Cross-half DIF stage: h = 0 keeps a + b, h = 1 keeps (a - b) W64^c; fft32_dif over c: position i = bin k1 + 32 h + 64 brev5(i).
//   backward: fft32_dit over i, cross-half DIT stage (h = 0: a + conj(W64^c) b, h = 1: a - conj(W64^c) b), transpose, conjugate twiddle,
//             fft32_dif<+1>: position i of lane64 = sample 64 brev5(i) + lane64.
// A frame costs about two period-512 frames; eight frames (one per wavefront) are in flight per block, second halves travel through a
// 9-slot LDS ring of 1024 floats, run boundaries are completed by atomic adds into a zeroed hop -- das_fused_kernel's scheme with a
// wavefront where it has a half-wavefront.  Gains (64 KB per direction at 8 microphones) come from L2: [pair][bin], natural order
// (das_pair_gains_natural), a.twiddle = exp(-2 pi i m / 2048), m < 1024, a.window = 2048 floats (the generic kernel's tables).
// (Round 4's two-pass kernels -- two FFT-1024 per frame -- took 0.74 ms per headline batch of samples: removed in round 5.)
constexpr int kHop2 = 1024, kN2 = 2048, kWaves2 = kBlock / 64;
constexpr int kWinRow2 = 36;  // floats per lane row of the window (32 + pad: float4 reads of 16-lane groups on distinct banks)
constexpr int o2Tw = 0, o2Pl = o2Tw + 2 * 32 * 64, o2Win = o2Pl + kWaves2 * kWPlane, o2W64 = o2Win + 64 * kWinRow2, o2Tail = o2W64 + 64,
              o2Flag = o2Tail + (kWaves2 + 1) * kHop2, kLds2048 = o2Flag + 32;

// (lo, hi) = the value of `v` in this lane's twin of half 0 / half 1 (lanes l and l + 32).  v_permlane32_swap_b32 vdst, vsrc exchanges
// vdst[32..63] with vsrc[0..31]: with both operands holding v, vdst becomes half 0's value in both halves and vsrc half 1's.  Written as
// inline assembly: through __builtin_amdgcn_permlane32_swap this compiler (ROCm 7.2) uses the first result for both (lo - hi came out as
// v_sub v, v).  The s_nop covers the VALU-write -> permlane-read wait states the compiler would have inserted.
// Four values per statement: the two wait states between the copies and the first swap are paid once.
__device__ __forceinline__ void halves_quad(const float (&v)[4], float (&lo)[4], float (&hi)[4]) {
    float a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3], b0 = v[0], b1 = v[1], b2 = v[2], b3 = v[3];
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\tv_permlane32_swap_b32 %2, %6\n\tv_permlane32_swap_b32 %3, %7"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
    lo[0] = a0; lo[1] = a1; lo[2] = a2; lo[3] = a3;
    hi[0] = b0; hi[1] = b1; hi[2] = b2; hi[3] = b3;
}

template <int LAYOUT>
__global__ __launch_bounds__(kBlock, 2) void das_fused_wave2048_kernel(DasFusedArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[kLds2048];
    cx<float> *s_tw = reinterpret_cast<cx<float> *>(lds + o2Tw);      // [k1][lane64] = W2048^(k1 lane64)
    cx<float> *s_w64 = reinterpret_cast<cx<float> *>(lds + o2W64);    // [c] = W64^c, c < 32
    float *s_win = lds + o2Win;                                        // [lane64][j] = window[64 j + lane64]
    float *s_tails = lds + o2Tail;                                     // 9 slots of 1024 floats
    volatile __attribute__((address_space(3))) int *s_flag = (volatile __attribute__((address_space(3))) int *)(lds + o2Flag);
    const int tid = threadIdx.x, lane64 = tid & 63, l = tid & 31, h = (tid >> 5) & 1;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *wplane = lds + o2Pl + w * kWPlane;
    const unsigned wbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float *)wplane);
    const float *wrowp = wplane + l * kWRow + 32 * h;
    const int M = a.n_mics, n_pairs = (M + 1) >> 1;
    const int stream = blockIdx.x / a.chunks_per_stream;  // output stream = input stream * n_dirs + look direction
    const long c_in_s = blockIdx.x - (long)stream * a.chunks_per_stream;
    const int in_stream = stream / a.n_dirs;
    const f32x2 *gains = a.gains + (long)(stream - in_stream * a.n_dirs) * n_pairs * kN2;  // [pair][bin], 1/N folded in
    {
        for (int i = tid; i < 32 * 64; i += kBlock) {
            const int m = ((i >> 6) * (i & 63)) % kN2;  // a.twiddle[m] = W2048^m for m < 1024; W^(m + 1024) = -W^m
            const f32x2 t = a.twiddle[m % 1024];
            s_tw[i] = m < 1024 ? cx<float>{t.x, t.y} : cx<float>{-t.x, -t.y};
        }
        if (tid < 32) {
            const f32x2 t = a.twiddle[32 * tid];  // W64^c = W2048^(32 c)
            s_w64[tid] = cx<float>{t.x, t.y};
        }
        for (int i = tid; i < kN2; i += kBlock) s_win[(i & 63) * kWinRow2 + (i >> 6)] = a.window[i];
    }
    const long T0 = c_in_s * a.frames_per_chunk;
    long T1 = T0 + a.frames_per_chunk;
    if (T1 > a.n_frames) T1 = a.n_frames;
    if (T0 == 0) {  // stream start: the overlap partner of frame 0 is the carried state (out_buff[0], util.h:302)
        for (int i = tid; i < kHop2; i += kBlock) s_tails[i] = a.tail_in[(long)stream * kHop2 + i];
    }
    if (tid <= kWaves2) s_flag[tid] = (tid == 0) ? (int)(T0 - 1) : -2;  // slot 0 holds "frame T0-1" (state, or unused when T0 > 0)
    __syncthreads();
    const float4 *wrow = reinterpret_cast<const float4 *>(s_win + lane64 * kWinRow2);
    const float *xs = a.x + (long)in_stream * a.stream_stride_x;
    const float *hs = a.hist_in + (long)in_stream * M * kHop2;
    float *ys = a.y + (long)stream * a.n_frames * kHop2;
    const cx<float> w64c = cx<float>{0.f, 0.f};
    (void)w64c;

    const int n_iter = (int)((T1 - T0 + kWaves2 - 1) / kWaves2);
    float re[32], im[32], Sr[32], Si[32];
    auto frame_of = [&](int it) -> long {  // the frame of this wavefront at iteration `it` (clamped into the run: spare slots redo its last frame)
        const long t = T0 + (long)it * kWaves2 + w;
        return t < T1 ? t : T1 - 1;
    };
    // raw samples of microphone pair p of frame tc into re (mic 2p) / im (mic 2p+1): register j <-> sample 64 j + lane64
    auto issue_loads = [&](long tc, int p) {
        const int ma = 2 * p, mb = 2 * p + 1;
        const bool b_ok = mb < M;  // odd microphone count: the last pair's partner channel reads the zero buffer
        if (LAYOUT == 0) {
            const float *a1 = (tc >= 1 ? xs + (long)ma * a.mic_stride + (tc - 1) * kHop2 : hs + ma * kHop2) + lane64;
            const float *b1 = (!b_ok ? a.zeros : tc >= 1 ? xs + (long)mb * a.mic_stride + (tc - 1) * kHop2 : hs + mb * kHop2) + lane64;
            const float *a2 = xs + (long)ma * a.mic_stride + tc * kHop2 + lane64;
            const float *b2 = (!b_ok ? a.zeros : xs + (long)mb * a.mic_stride + tc * kHop2) + lane64;
            const int bstep = b_ok ? 64 : 0;  // a.zeros holds 2048 floats
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                re[j] = a1[64 * j];
                im[j] = b1[bstep * j];
                re[j + 16] = a2[64 * j];
                im[j + 16] = b2[bstep * j];
            }
        } else {
            const float *s1 = (tc >= 1 ? xs + (tc - 1) * (long)kHop2 * M : hs) + (long)lane64 * M;
            const float *s2 = xs + tc * (long)kHop2 * M + (long)lane64 * M;
            if (b_ok && (M & 1) == 0) {  // the two microphones of a pair are adjacent: one 8-byte load per sample
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float2 u = *reinterpret_cast<const float2 *>(s1 + (long)64 * j * M + ma);
                    const float2 v = *reinterpret_cast<const float2 *>(s2 + (long)64 * j * M + ma);
                    re[j] = u.x;
                    im[j] = u.y;
                    re[j + 16] = v.x;
                    im[j + 16] = v.y;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    re[j] = s1[(long)64 * j * M + ma];
                    im[j] = b_ok ? s1[(long)64 * j * M + mb] : 0.f;
                    re[j + 16] = s2[(long)64 * j * M + ma];
                    im[j + 16] = b_ok ? s2[(long)64 * j * M + mb] : 0.f;
                }
            }
        }
    };
    if (n_iter > 0) issue_loads(frame_of(0), 0);
    for (int it = 0; it < n_iter; ++it) {
        const long t = T0 + (long)it * kWaves2 + w;
        const bool valid = t < T1;
        const long tc = valid ? t : T1 - 1;
        for (int p = 0; p < n_pairs; ++p) {
            if (p > 0) issue_loads(tc, p);
#pragma unroll
            for (int g = 0; g < 8; ++g) {  // buf[j]*hann_win[i]  (util.h:235)
                const float4 hv = wrow[g];
                re[4 * g + 0] *= hv.x; im[4 * g + 0] *= hv.x;
                re[4 * g + 1] *= hv.y; im[4 * g + 1] *= hv.y;
                re[4 * g + 2] *= hv.z; im[4 * g + 2] *= hv.z;
                re[4 * g + 3] *= hv.w; im[4 * g + 3] *= hv.w;
            }
            fft32_dif<float, -1>(re, im);
#pragma unroll
            for (int i = 1; i < 32; ++i) {
                const cx<float> tw = s_tw[brev5(i) * 64 + lane64];
                const float xr = re[i], xi = im[i];
                re[i] = xr * tw.x - xi * tw.y;
                im[i] = xr * tw.y + xi * tw.x;
            }
            wt_store_plane<false>(re, wbase);
            wt_load_row(re, wrowp);
            wt_store_plane<false>(im, wbase);
            wt_load_row(im, wrowp);
            // lane (k1 = l, h): register c = n2 = 32 h + c.  Radix-2 DIF stage over h: half 0 keeps a + b, half 1 keeps (a - b) W64^c
#pragma unroll
            for (int c = 0; c < 32; c += 2) {
                const float v[4] = {re[c], im[c], re[c + 1], im[c + 1]};
                float lo[4], hi[4];
                halves_quad(v, lo, hi);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float ar = lo[2 * e], ai = lo[2 * e + 1], br = hi[2 * e], bi = hi[2 * e + 1];
                    const cx<float> tw = s_w64[c + e];
                    const float dr = ar - br, di = ai - bi;
                    re[c + e] = h ? dr * tw.x - di * tw.y : ar + br;
                    im[c + e] = h ? dr * tw.y + di * tw.x : ai + bi;
                }
            }
            fft32_dif<float, -1>(re, im);
            const f32x2 *gp = gains + (long)p * kN2 + lane64;  // position i = bin lane64 + 64 brev5(i)
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const f32x2 g = gp[64 * brev5(i)];
                Sr[i] = bf_fma(-g.y, im[i], bf_fma(g.x, re[i], p == 0 ? 0.f : Sr[i]));
                Si[i] = bf_fma(g.y, re[i], bf_fma(g.x, im[i], p == 0 ? 0.f : Si[i]));
            }
        }
        // the next frame's first pair streams in while the backward transform runs on (Sr, Si)
        if (it + 1 < n_iter) issue_loads(frame_of(it + 1), 0);
        fft32_dit<float, +1>(Sr, Si);
#pragma unroll
        for (int c = 0; c < 32; c += 2) {  // DIT stage over h: n2 = c + 32 h <- a + conj(W64^c) b (h = 0), a - conj(W64^c) b (h = 1)
            const float v[4] = {Sr[c], Si[c], Sr[c + 1], Si[c + 1]};
            float lo[4], hi[4];
            halves_quad(v, lo, hi);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float ar = lo[2 * e], ai = lo[2 * e + 1], br = hi[2 * e], bi = hi[2 * e + 1];
                const cx<float> tw = s_w64[c + e];
                const float tr = br * tw.x + bi * tw.y, ti = bi * tw.x - br * tw.y;
                Sr[c + e] = h ? ar - tr : ar + tr;
                Si[c + e] = h ? ai - ti : ai + ti;
            }
        }
        wt_store_plane<true>(Sr, wbase);
        wt_load_row(Sr, wrowp);
        wt_store_plane<true>(Si, wbase);
        wt_load_row(Si, wrowp);
        // lane64 = n2 (l + 32 h: row l, columns of half h ... see below), register c = k1
#pragma unroll
        for (int c = 1; c < 32; ++c) {
            const cx<float> tw = s_tw[c * 64 + lane64];  // conj applied
            const float xr = Sr[c], xi = Si[c];
            Sr[c] = xr * tw.x + xi * tw.y;
            Si[c] = xi * tw.x - xr * tw.y;
        }
        fft32_dif<float, +1>(Sr, Si);
        // position i holds sample n = 64 brev5(i) + lane64; even i -> first half, odd i -> n + 1024
        float hh[32];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float4 hv = wrow[g];
            hh[4 * g + 0] = hv.x; hh[4 * g + 1] = hv.y; hh[4 * g + 2] = hv.z; hh[4 * g + 3] = hv.w;
        }
        const int r = (int)(tc - T0);
        const int my = (r + 1) % (kWaves2 + 1), pv = r % (kWaves2 + 1);
        if (valid) {
            float *my_slot = s_tails + my * kHop2 + lane64;
#pragma unroll
            for (int q = 0; q < 16; ++q) my_slot[64 * brev5(2 * q)] = Sr[2 * q + 1] * hh[brev5(2 * q + 1)];
            asm volatile("" ::: "memory");  // data then flag: LDS operations of one wavefront execute in issue order
            if (lane64 == 0) s_flag[my] = (int)t;
            float *yo = ys + t * kHop2 + lane64;
            if (t == T0 && T0 > 0) {
                // first hop of the run: the previous run adds its half separately (both into a zeroed hop)
#pragma unroll
                for (int q = 0; q < 16; ++q) atomicAdd(yo + 64 * brev5(2 * q), Sr[2 * q] * hh[brev5(2 * q)]);
            } else {
                while (s_flag[pv] != (int)(t - 1)) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
                const float *prev = s_tails + pv * kHop2 + lane64;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
#pragma clang fp contract(off)
                    yo[64 * brev5(2 * q)] = prev[64 * brev5(2 * q)] + Sr[2 * q] * hh[brev5(2 * q)];  // out = prev[H + n] + cur[n]  (util.h:301-302)
                }
            }
            if (t == T1 - 1) {
                if (T1 < a.n_frames) {  // last frame of the run: its second half belongs to the next run's first hop
                    float *yn = ys + T1 * kHop2 + lane64;
#pragma unroll
                    for (int q = 0; q < 16; ++q) atomicAdd(yn + 64 * brev5(2 * q), Sr[2 * q + 1] * hh[brev5(2 * q + 1)]);
                } else {
                    // end of the batch: carried state for the next call (OLA tail and the last input hop)
                    float *to = a.tail_out + (long)stream * kHop2 + lane64;
#pragma unroll
                    for (int q = 0; q < 16; ++q) to[64 * brev5(2 * q)] = Sr[2 * q + 1] * hh[brev5(2 * q + 1)];
                    float *ho = a.hist_out + (long)in_stream * M * kHop2;  // every direction writes the same values
                    if (LAYOUT == 0) {
                        for (int m = 0; m < M; ++m)
                            for (int j = 0; j < 16; ++j) ho[m * kHop2 + 64 * j + lane64] = xs[(long)m * a.mic_stride + t * kHop2 + 64 * j + lane64];
                    } else {
                        for (int j = 0; j < 16 * M; ++j) ho[64 * j + lane64] = xs[t * (long)kHop2 * M + 64 * j + lane64];
                    }
                }
            }
        }
    }
}

// ---- interleaved input [sample][mic], 4 or 8 microphones -----------------------------------------------------------------------
// The generic kernel reads one pair (8 bytes) of every 4 M-byte sample per pass: at M = 8 a wave-instruction touches sixteen
// 128-byte lines and uses a quarter of each, and the four passes of a frame fetch every line four times through the TCP
// (0.69 ms per 65 536-frame batch against 0.35 planar).  Here one 16-byte load brings the two pairs of a group (microphones
// 4g .. 4g+3) of a sample: pair 2g lands in (ar, ai), pair 2g+1 waits in (br, bi) and is transformed in place there, so every
// line is touched twice per frame at M = 8 and once at M = 4.  Register budget: two data sets + the accumulator = 192.
// Same transform, tables and ring as das_fused_kernel<0, NPL, true, UNR>; NG = number of groups (M / 4).
template <int NPL, int NG>
__global__ __launch_bounds__(kBlock, 2) void das_fused_il_kernel(DasFusedArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[kLdsFixed + NPL * 2048 + 17 * kHop + 32];
    float *s_win = lds + kLdsTw + kHalves * kLdsPlane;
    float *s_tails = lds + kLdsFixed + NPL * 2048;
    // LDS address space spelled out: a volatile access through a generic pointer compiles to flat_load / flat_store, whose wait is
    // vmcnt(0) -- it would drain the next frame's prefetch in front of every flag read
    volatile __attribute__((address_space(3))) int *s_flag = (volatile __attribute__((address_space(3))) int *)(lds + kLdsFixed + NPL * 2048 + 17 * kHop);
    constexpr int M = 4 * NG, UNR = 2 * NG;

    const int tid = threadIdx.x;
    const int lane = tid & 31;
    const int hw = tid >> 5;
    float *wplane = lds + kLdsTw + (hw >> 1) * kWPlane;
    const unsigned wbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float *)wplane);
    const float *wrowp = wplane + lane * kWRow + 32 * (hw & 1);

    const int stream = blockIdx.x / a.chunks_per_stream;
    const long c_in_s = blockIdx.x - (long)stream * a.chunks_per_stream;
    const int in_stream = stream / a.n_dirs;
    const f32x2 *gains = a.gains + (long)(stream - in_stream * a.n_dirs) * UNR * 1024;
    {
        const f32x2 *twc = a.twiddle;
        f32x2 *ltw = reinterpret_cast<f32x2 *>(lds);
        for (int i = tid; i < kLdsTw / 2; i += kBlock) {  // paired tables, as the unrolled planar kernel
            const int k1 = i >> 5, l = i & 31;
            ltw[((k1 & 15) * 32 + l) * 2 + (k1 >> 4)] = twc[i];
        }
        for (int i = tid; i < kNfft; i += kBlock) s_win[(i & 31) * kPS + (i >> 5)] = a.window[i];
        f32x2 *lg = reinterpret_cast<f32x2 *>(lds + kLdsFixed);
        for (int i = tid; i < UNR * 1024; i += kBlock) {
            const int l = i & 31, pos = (i >> 5) & 31, pr = i >> 10;
            lg[((pr * 16 + (pos >> 1)) * 32 + l) * 2 + (pos & 1)] = gains[i];
        }
    }
    const long T0 = c_in_s * a.frames_per_chunk;
    long T1 = T0 + a.frames_per_chunk;
    if (T1 > a.n_frames) T1 = a.n_frames;
    if (T0 == 0) {
        for (int i = tid; i < kHop; i += kBlock) s_tails[i] = a.tail_in[(long)stream * kHop + i];
    }
    if (tid < 17) s_flag[tid] = (tid == 0) ? (int)(T0 - 1) : -2;
    __syncthreads();
    const float4 *wrow = reinterpret_cast<const float4 *>(s_win + lane * kPS);
    const float4 *tw2 = reinterpret_cast<const float4 *>(lds);

    const float *xs = a.x + (long)in_stream * a.stream_stride_x;
    const float *hs = a.hist_in + (long)in_stream * M * kHop;
    float *ys = a.y + (long)stream * a.n_frames * kHop;

    float ar[32], ai[32], br[32], bi[32], Sr[32], Si[32];
    const int n_iter = (int)((T1 - T0 + kHalves - 1) / kHalves);

    // the two pairs of group g of frame tc: register position j <-> sample 32 j + lane of the frame
    auto load_group = [&](long tc, int g) {
        const float *s1 = (tc >= 1 ? xs + (tc - 1) * (long)kHop * M : hs) + (long)lane * M + 4 * g;
        const float *s2 = xs + tc * (long)kHop * M + (long)lane * M + 4 * g;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float4 u = *reinterpret_cast<const float4 *>(s1 + 32 * j * M);
            const float4 w = *reinterpret_cast<const float4 *>(s2 + 32 * j * M);
            ar[j] = u.x; ai[j] = u.y; br[j] = u.z; bi[j] = u.w;
            ar[j + 16] = w.x; ai[j + 16] = w.y; br[j + 16] = w.z; bi[j + 16] = w.w;
        }
    };
    // window -> FFT-1024 -> weight-and-sum of the pair held in (re, im)
    auto do_pair = [&](float (&re)[32], float (&im)[32], int p) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float4 hv = wrow[g];
            re[4 * g + 0] *= hv.x; im[4 * g + 0] *= hv.x;
            re[4 * g + 1] *= hv.y; im[4 * g + 1] *= hv.y;
            re[4 * g + 2] *= hv.z; im[4 * g + 2] *= hv.z;
            re[4 * g + 3] *= hv.w; im[4 * g + 3] *= hv.w;
        }
#ifdef BF_DAS_STAMPS
        unsigned long long st_acc[16], st_prev = 0;
#endif
        wt_fft_fwd_p2(re, im, lane, tw2, wbase, wrowp BF_STAMP_ARGS);
        const float4 *gp2 = reinterpret_cast<const float4 *>(lds + kLdsFixed) + p * 512 + lane;
#pragma unroll
        for (int i = 0; i < 32; i += 2) {
            const float4 g = gp2[16 * i];  // gains of positions i and i + 1
            Sr[i] = bf_fma(-g.y, im[i], bf_fma(g.x, re[i], (p == 0) ? 0.f : Sr[i]));
            Si[i] = bf_fma(g.y, re[i], bf_fma(g.x, im[i], (p == 0) ? 0.f : Si[i]));
            Sr[i + 1] = bf_fma(-g.w, im[i + 1], bf_fma(g.z, re[i + 1], (p == 0) ? 0.f : Sr[i + 1]));
            Si[i + 1] = bf_fma(g.w, re[i + 1], bf_fma(g.z, im[i + 1], (p == 0) ? 0.f : Si[i + 1]));
        }
    };

    {
        const long t0 = T0 + hw;
        load_group(t0 < T1 ? t0 : T1 - 1, 0);
    }
    for (int it = 0; it < n_iter; ++it) {
        const long t = T0 + (long)it * kHalves + hw;
        const bool valid = t < T1;
        const long tc = valid ? t : T1 - 1;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            // group 1: the same lines group 0 brought in (a 32-byte sample holds both), so these are L1 / L2 hits.  Issuing them
            // from inside pair 1's weight-and-sum was measured: the 16-byte register tuples fragment the file (153 scratch
            // operations per iteration), 1.06 ms instead of 0.50.
            if (g > 0) load_group(tc, g);
            do_pair(ar, ai, 2 * g);
            do_pair(br, bi, 2 * g + 1);
        }

        if (a.sdump != nullptr && valid) {
            f32x2 *sd = a.sdump + ((long)stream * a.n_frames + t) * kNfft + lane;
#pragma unroll
            for (int i = 0; i < 32; ++i) sd[32 * brev5(i)] = f32x2{Sr[i], Si[i]};
        }
        // the next frame's first group streams in while the inverse transform runs on (Sr, Si)
        if (it + 1 < n_iter) {
            const long tn = T0 + (long)(it + 1) * kHalves + hw;
            load_group(tn < T1 ? tn : T1 - 1, 0);
        }
        {
#ifdef BF_DAS_STAMPS
            unsigned long long st_acc[16], st_prev = 0;
#endif
            wt_fft_inv_p2(Sr, Si, lane, tw2, wbase, wrowp BF_STAMP_ARGS);
        }

        float h[32];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float4 hv = wrow[g];
            h[4 * g + 0] = hv.x; h[4 * g + 1] = hv.y; h[4 * g + 2] = hv.z; h[4 * g + 3] = hv.w;
        }
        const int r = (int)(tc - T0);
        const int my = (r + 1) % 17, pv = r % 17;
        if (valid) {
            float *my_slot = s_tails + my * kHop + lane;
#pragma unroll
            for (int q = 0; q < 16; ++q) my_slot[32 * brev5(2 * q)] = Sr[2 * q + 1] * h[brev5(2 * q + 1)];
            asm volatile("" ::: "memory");
            if (lane == 0) s_flag[my] = (int)t;
        }
        if (valid) {
            float *yo = ys + t * kHop + lane;
            if (t == T0 && T0 > 0) {
#pragma unroll
                for (int q = 0; q < 16; ++q) atomicAdd(yo + 32 * brev5(2 * q), Sr[2 * q] * h[brev5(2 * q)]);
            } else {
                while (s_flag[pv] != (int)(t - 1)) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
                const float *prev = s_tails + pv * kHop + lane;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
#pragma clang fp contract(off)
                    yo[32 * brev5(2 * q)] = prev[32 * brev5(2 * q)] + Sr[2 * q] * h[brev5(2 * q)];
                }
            }
            if (t == T1 - 1) {
                if (T1 < a.n_frames) {
                    float *yn = ys + T1 * kHop + lane;
#pragma unroll
                    for (int q = 0; q < 16; ++q) atomicAdd(yn + 32 * brev5(2 * q), Sr[2 * q + 1] * h[brev5(2 * q + 1)]);
                } else {
                    float *to = a.tail_out + (long)stream * kHop + lane;
#pragma unroll
                    for (int q = 0; q < 16; ++q) to[32 * brev5(2 * q)] = Sr[2 * q + 1] * h[brev5(2 * q + 1)];
                    float *ho = a.hist_out + (long)in_stream * M * kHop;
                    for (int j = 0; j < 16 * M; ++j) ho[32 * j + lane] = xs[t * (long)kHop * M + 32 * j + lane];
                }
            }
        }
    }
}

// ---- interleaved input [sample][mic], 8 microphones: one WAVEFRONT per frame -----------------------------------------------
// A 32-byte sample holds all eight microphones.  Lane l of half-wavefront h loads bytes 16 h .. 16 h + 15 of sample 32 j + l
// (microphones 4h .. 4h+3 = pairs 2h, 2h+1), so ONE wave-instruction covers 32 whole samples = 1 KiB of contiguous memory:
// every 128-byte line is requested once, fully used, by a quarter of the load instructions the planar kernel issues (its
// loads are 4 bytes per lane).  Each half transforms its two pairs and accumulates a partial S over them; the halves exchange
// their partial sums with v_permlane32_swap (copy + swap + add per register, no LDS) and then BOTH hold S.  The backward
// transform therefore runs on the whole wavefront for one frame (half of it redundant: 3 wavefront-passes per frame where the
// planar kernel needs 2.5); the output work is split: half 0 stores the frame's output hop, half 1 parks the tail for the next
// frame.  The frame's single load phase is issued in front of the previous frame's backward transform and lands behind it.
// Same tables and tail ring as das_fused_kernel; the accumulation order is (p0 + p1) + (p2 + p3): last-bit differences from the
// planar kernel, none between batch / run cuts.
template <int I0>
__device__ __forceinline__ void halves_sum8(float (&v)[32]) {
    float c0 = v[I0], c1 = v[I0 + 1], c2 = v[I0 + 2], c3 = v[I0 + 3], c4 = v[I0 + 4], c5 = v[I0 + 5], c6 = v[I0 + 6], c7 = v[I0 + 7];
    // swap a, b: lanes 32..63 of a <-> lanes 0..31 of b.  With b a copy of a: a = (lo, lo), b = (hi, hi); a + b = lo + hi in both halves.
    // s_nop 1: the two wait states a VALU write needs before v_permlane*_swap reads it.
    asm volatile(
        "s_nop 1\n\t"
        "v_permlane32_swap_b32 %0, %8\n\tv_permlane32_swap_b32 %1, %9\n\tv_permlane32_swap_b32 %2, %10\n\tv_permlane32_swap_b32 %3, %11\n\t"
        "v_permlane32_swap_b32 %4, %12\n\tv_permlane32_swap_b32 %5, %13\n\tv_permlane32_swap_b32 %6, %14\n\tv_permlane32_swap_b32 %7, %15"
        : "+v"(v[I0]), "+v"(v[I0 + 1]), "+v"(v[I0 + 2]), "+v"(v[I0 + 3]), "+v"(v[I0 + 4]), "+v"(v[I0 + 5]), "+v"(v[I0 + 6]), "+v"(v[I0 + 7]),
          "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7));
    v[I0] += c0; v[I0 + 1] += c1; v[I0 + 2] += c2; v[I0 + 3] += c3;
    v[I0 + 4] += c4; v[I0 + 5] += c5; v[I0 + 6] += c6; v[I0 + 7] += c7;
}
__device__ __forceinline__ void halves_sum(float (&v)[32]) {
    halves_sum8<0>(v);
    halves_sum8<8>(v);
    halves_sum8<16>(v);
    halves_sum8<24>(v);
}

constexpr int kIl8Waves = kBlock / 64;  // frames in flight per block
#ifndef BF_IL8_SPLIT
#define BF_IL8_SPLIT 1
#endif
constexpr bool kIl8Split = BF_IL8_SPLIT != 0;
__global__ __launch_bounds__(kBlock, 2) void das_fused_il8_kernel(DasFusedArgs a) {
    constexpr int NPL = 4, M = 8;
    __shared__ __attribute__((aligned(16))) float lds[kLdsFixed + NPL * 2048 + (kIl8Waves + 1) * kHop + 32];
    float *s_win = lds + kLdsTw + kHalves * kLdsPlane;
    float *s_tails = lds + kLdsFixed + NPL * 2048;
    // LDS address space spelled out: a volatile access through a generic pointer compiles to flat_load / flat_store, whose wait is
    // vmcnt(0) -- it would drain the next frame's prefetch in front of every flag read
    volatile __attribute__((address_space(3))) int *s_flag = (volatile __attribute__((address_space(3))) int *)(lds + kLdsFixed + NPL * 2048 + (kIl8Waves + 1) * kHop);
    constexpr int kSlots = kIl8Waves + 1;

    const int tid = threadIdx.x;
    const int lane = tid & 31;
    const int hw = tid >> 5, half = hw & 1;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform: the frame, its addresses and its ring slots live in SGPRs
    float *wplane = lds + kLdsTw + wv * kWPlane;
    const unsigned wbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float *)wplane);
    const float *wrowp = wplane + lane * kWRow + 32 * half;

    const int stream = blockIdx.x / a.chunks_per_stream;
    const long c_in_s = blockIdx.x - (long)stream * a.chunks_per_stream;
    const int in_stream = stream / a.n_dirs;
    const f32x2 *gains = a.gains + (long)(stream - in_stream * a.n_dirs) * NPL * 1024;
    {
        const f32x2 *twc = a.twiddle;
        f32x2 *ltw = reinterpret_cast<f32x2 *>(lds);
        for (int i = tid; i < kLdsTw / 2; i += kBlock) {  // paired tables, as the unrolled planar kernel
            const int k1 = i >> 5, l = i & 31;
            ltw[((k1 & 15) * 32 + l) * 2 + (k1 >> 4)] = twc[i];
        }
        for (int i = tid; i < kNfft; i += kBlock) s_win[(i & 31) * kPS + (i >> 5)] = a.window[i];
        f32x2 *lg = reinterpret_cast<f32x2 *>(lds + kLdsFixed);
        for (int i = tid; i < NPL * 1024; i += kBlock) {
            const int l = i & 31, pos = (i >> 5) & 31, pr = i >> 10;
            lg[((pr * 16 + (pos >> 1)) * 32 + l) * 2 + (pos & 1)] = gains[i];
        }
    }
    const long T0 = c_in_s * a.frames_per_chunk;
    long T1 = T0 + a.frames_per_chunk;
    if (T1 > a.n_frames) T1 = a.n_frames;
    if (T0 == 0) {
        for (int i = tid; i < kHop; i += kBlock) s_tails[i] = a.tail_in[(long)stream * kHop + i];
    }
    if (tid < kSlots) s_flag[tid] = (tid == 0) ? (int)(T0 - 1) : -2;
    __syncthreads();
    const float4 *wrow = reinterpret_cast<const float4 *>(s_win + lane * kPS);
    const float4 *tw2 = reinterpret_cast<const float4 *>(lds);

    const float *xs = a.x + (long)in_stream * a.stream_stride_x;
    const float *hs = a.hist_in + (long)in_stream * M * kHop;
    float *ys = a.y + (long)stream * a.n_frames * kHop;

    float ar[32], ai[32], br[32], bi[32], Sr[32], Si[32];
    const int n_iter = (int)((T1 - T0 + kIl8Waves - 1) / kIl8Waves);

    // this half's two pairs of frame tc: register position j <-> sample 32 j + lane of the frame
    const unsigned voff = (unsigned)(lane * M + 4 * half);  // this lane's 16 bytes inside a 32-sample row (floats)
    auto load_frame = [&](long tc) {
        const float *s1 = (tc >= 1 ? xs + (tc - 1) * (long)kHop * M : hs);  // wave-uniform bases: SGPR base + VGPR offset addressing
        const float *s2 = xs + tc * (long)kHop * M;
        if (kIl8Split) {
            // the half's first pair in the first 32 requests, its second pair behind them: the first transform starts when half of the
            // frame has arrived (vmcnt(32)) and the other half gets that transform's time on top
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float2 u = *reinterpret_cast<const float2 *>(s1 + 32 * j * M + voff);
                const float2 w = *reinterpret_cast<const float2 *>(s2 + 32 * j * M + voff);
                ar[j] = u.x; ai[j] = u.y;
                ar[j + 16] = w.x; ai[j + 16] = w.y;
            }
            unsigned voff2 = voff + 2;
            asm volatile("" : "+v"(voff2));  // opaque: the load vectorizer would fuse the two halves of a sample back into one 16-byte load
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float2 u = *reinterpret_cast<const float2 *>(s1 + 32 * j * M + voff2);
                const float2 w = *reinterpret_cast<const float2 *>(s2 + 32 * j * M + voff2);
                br[j] = u.x; bi[j] = u.y;
                br[j + 16] = w.x; bi[j + 16] = w.y;
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float4 u = *reinterpret_cast<const float4 *>(s1 + 32 * j * M + voff);
            const float4 w = *reinterpret_cast<const float4 *>(s2 + 32 * j * M + voff);
            ar[j] = u.x; ai[j] = u.y; br[j] = u.z; bi[j] = u.w;
            ar[j + 16] = w.x; ai[j + 16] = w.y; br[j + 16] = w.z; bi[j + 16] = w.w;
        }
    };
    // window -> FFT-1024 -> weight-and-sum of the pair held in (re, im); FIRST: the half's first pair starts its partial sum
    auto do_pair = [&](float (&re)[32], float (&im)[32], int p, bool first) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float4 hv = wrow[g];
            re[4 * g + 0] *= hv.x; im[4 * g + 0] *= hv.x;
            re[4 * g + 1] *= hv.y; im[4 * g + 1] *= hv.y;
            re[4 * g + 2] *= hv.z; im[4 * g + 2] *= hv.z;
            re[4 * g + 3] *= hv.w; im[4 * g + 3] *= hv.w;
        }
#ifdef BF_DAS_STAMPS
        unsigned long long st_acc[16], st_prev = 0;
#endif
        wt_fft_fwd_p2(re, im, lane, tw2, wbase, wrowp BF_STAMP_ARGS);
        const float4 *gp2 = reinterpret_cast<const float4 *>(lds + kLdsFixed) + p * 512 + lane;
#pragma unroll
        for (int i = 0; i < 32; i += 2) {
            const float4 g = gp2[16 * i];  // gains of positions i and i + 1
            Sr[i] = bf_fma(-g.y, im[i], bf_fma(g.x, re[i], first ? 0.f : Sr[i]));
            Si[i] = bf_fma(g.y, re[i], bf_fma(g.x, im[i], first ? 0.f : Si[i]));
            Sr[i + 1] = bf_fma(-g.w, im[i + 1], bf_fma(g.z, re[i + 1], first ? 0.f : Sr[i + 1]));
            Si[i + 1] = bf_fma(g.w, re[i + 1], bf_fma(g.z, im[i + 1], first ? 0.f : Si[i + 1]));
        }
    };

    {
        const long t0 = T0 + wv;
        load_frame(t0 < T1 ? t0 : T1 - 1);
    }
    for (int it = 0; it < n_iter; ++it) {
        const long t = T0 + (long)it * kIl8Waves + wv;
        const bool valid = t < T1;
        const long tc = valid ? t : T1 - 1;
        do_pair(ar, ai, 2 * half, true);
        do_pair(br, bi, 2 * half + 1, false);
        halves_sum(Sr);
        halves_sum(Si);

        if (a.sdump != nullptr && valid && half == 0) {
            f32x2 *sd = a.sdump + ((long)stream * a.n_frames + t) * kNfft + lane;
#pragma unroll
            for (int i = 0; i < 32; ++i) sd[32 * brev5(i)] = f32x2{Sr[i], Si[i]};
        }
        // the next frame streams in while the backward transform runs on (Sr, Si)
        if (it + 1 < n_iter) {
            const long tn = T0 + (long)(it + 1) * kIl8Waves + wv;
            load_frame(tn < T1 ? tn : T1 - 1);
        }
        {
#ifdef BF_DAS_STAMPS
            unsigned long long st_acc[16], st_prev = 0;
#endif
            wt_fft_inv_p2(Sr, Si, lane, tw2, wbase, wrowp BF_STAMP_ARGS);
        }

        // position i holds sample n = 32*brev5(i) + lane; even i -> first half of the frame, odd i -> second half.  Half-wavefront 0
        // finishes the output hop (first half + the previous frame's tail), half-wavefront 1 parks this frame's tail: each needs
        // 16 of the 32 positions and 16 of the 32 window values (brev5(2q + 1) = brev5(2q) + 16: a contiguous half of the row)
        float pq[16];
        {
            const float4 *wh = wrow + 4 * half;
            float hq[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 hv = wh[g];
                hq[4 * g + 0] = hv.x; hq[4 * g + 1] = hv.y; hq[4 * g + 2] = hv.z; hq[4 * g + 3] = hv.w;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float ev = Sr[2 * q], od = Sr[2 * q + 1];  // select on VALUES (v_cndmask), not on addresses (a scratch array)
                pq[q] = (half != 0 ? od : ev) * hq[brev5(2 * q)];
            }
        }
        const int r = (int)(tc - T0);
        const int my = (r + 1) % kSlots, pv = r % kSlots;
        if (valid && half == 1) {  // second half of this frame -> its ring slot, then publish
            float *my_slot = s_tails + my * kHop + lane;
#pragma unroll
            for (int q = 0; q < 16; ++q) my_slot[32 * brev5(2 * q)] = pq[q];
            asm volatile("" ::: "memory");
            if (lane == 0) s_flag[my] = (int)t;
            if (t == T1 - 1) {
                if (T1 < a.n_frames) {  // last frame of the run: its second half belongs to the next run's first hop
                    float *yn = ys + T1 * kHop + lane;
#pragma unroll
                    for (int q = 0; q < 16; ++q) atomicAdd(yn + 32 * brev5(2 * q), pq[q]);
                } else {  // end of the batch: carried state for the next call (OLA tail and the last input hop)
                    float *to = a.tail_out + (long)stream * kHop + lane;
#pragma unroll
                    for (int q = 0; q < 16; ++q) to[32 * brev5(2 * q)] = pq[q];
                    float *ho = a.hist_out + (long)in_stream * M * kHop;
                    for (int j = 0; j < 16 * M; ++j) ho[32 * j + lane] = xs[t * (long)kHop * M + 32 * j + lane];
                }
            }
        }
        if (valid && half == 0) {
            float *yo = ys + t * kHop + lane;
            if (t == T0 && T0 > 0) {  // first hop of the run: the previous run adds its half separately (both into a zeroed hop)
#pragma unroll
                for (int q = 0; q < 16; ++q) atomicAdd(yo + 32 * brev5(2 * q), pq[q]);
            } else {
                while (s_flag[pv] != (int)(t - 1)) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
                const float *prev = s_tails + pv * kHop + lane;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
#pragma clang fp contract(off)
                    yo[32 * brev5(2 * q)] = prev[32 * brev5(2 * q)] + pq[q];
                }
            }
        }
    }
}

// ---- several look directions from ONE set of forward transforms -----------------------------------------------------------
// The reference transforms a frame once and applies its weights (das.cpp:51-63); a controller that scans D candidate
// directions (scripts/energy2theta.py:62-101) runs D nodes on the same input.  das_fused_kernel gives every direction its own
// block, which redoes the forward transforms: D directions cost D launches' worth.  Here a block walks its run two frames at
// a time:
//   forward phase   wavefronts 0-3 = eight half-wavefronts = (frame A | B) x (pair 0..3): window, packed forward transform,
//                   spectrum parked in LDS -- [frame][pair][position pair][lane] float4, 64 KiB;
//   direction phase wavefront w takes directions 2w and 2w+1 one after the other, its half h the frame A + h: weight-and-sum
//                   out of the parked spectra (gains straight from global memory: both halves read the SAME direction's table,
//                   one 256-byte segment per wave-instruction), backward transform, window, overlap-add.
// Overlap-add partner: frame A's second half goes to half 1 by v_permlane32_swap in the same step; frame B's second half goes
// the other way and waits in half 0's registers for the next round (16 registers per direction) -- no ring, no flags.  Run
// boundaries as in das_fused_kernel (two float atomic adds into a pre-zeroed hop).  Same transforms, same separately rounded
// overlap-add, the weight-and-sum accumulates pairs in the same order: the output equals das_fused_kernel<0, NPL, true, UNR>'s
// to the last bits (hipcc fuses the analysis window's products into the first butterflies differently in the two kernels)
// and does not depend on how a stream is cut into batches and runs.
__device__ __forceinline__ float halves_exchange(float v, int half) {
    float lo = v, hi = v;  // swap: lanes 32..63 of lo <-> lanes 0..31 of hi  ->  lo = (v.lo, v.lo), hi = (v.hi, v.hi)
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
    return half ? lo : hi;
}

constexpr int kDirsZ = 2 * 4 * 16 * 32 * 4;  // floats
__global__ __launch_bounds__(kBlock, 2) void das_fused_dirs_kernel(DasFusedArgs a, int dir0, int n_here) {
    __shared__ __attribute__((aligned(16))) float lds[kLdsTw + 8 * kWPlane + kLdsWin + kDirsZ];
    float *s_win = lds + kLdsTw + 8 * kWPlane;
    float4 *s_z = reinterpret_cast<float4 *>(lds + kLdsTw + 8 * kWPlane + kLdsWin);

    const int tid = threadIdx.x;
    const int lane = tid & 31;
    const int hw = tid >> 5, half = hw & 1;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *wplane = lds + kLdsTw + wv * kWPlane;
    const unsigned wbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float *)wplane);
    const float *wrowp = wplane + lane * kWRow + 32 * half;
    const int M = a.n_mics, np = (M + 1) >> 1;

    const int in_stream = blockIdx.x / a.chunks_per_stream;
    const long c_in_s = blockIdx.x - (long)in_stream * a.chunks_per_stream;
    {
        const f32x2 *twc = a.twiddle;
        f32x2 *ltw = reinterpret_cast<f32x2 *>(lds);
        for (int i = tid; i < kLdsTw / 2; i += kBlock) {  // paired tables, as the unrolled planar kernel
            const int k1 = i >> 5, l = i & 31;
            ltw[((k1 & 15) * 32 + l) * 2 + (k1 >> 4)] = twc[i];
        }
        for (int i = tid; i < kNfft; i += kBlock) s_win[(i & 31) * kPS + (i >> 5)] = a.window[i];
    }
    const long T0 = c_in_s * a.frames_per_chunk;
    long T1 = T0 + a.frames_per_chunk;
    if (T1 > a.n_frames) T1 = a.n_frames;
    __syncthreads();
    const float4 *wrow = reinterpret_cast<const float4 *>(s_win + lane * kPS);
    const float4 *tw2 = reinterpret_cast<const float4 *>(lds);
    const float *xs = a.x + (long)in_stream * a.stream_stride_x;
    const float *hs = a.hist_in + (long)in_stream * M * kHop;

    float re[32], im[32];
    float keep[2][16];  // half 0: the second half of the frame before this round's frame A, per direction of this wavefront
#pragma unroll
    for (int dd = 0; dd < 2; ++dd) {
        const int dl = dd * 8 + (7 - wv);
        const long so = (long)in_stream * a.n_dirs + dir0 + (dl < n_here ? dl : 0);
#pragma unroll
        for (int q = 0; q < 16; ++q) keep[dd][q] = (T0 == 0) ? a.tail_in[so * kHop + 32 * brev5(2 * q) + lane] : 0.f;
    }

    // raw samples of this half-wavefront's (frame, pair) of the round that starts at frame tA -> (re, im), natural order
    auto issue_inputs = [&](long tA) {
        const int p = hw & 3;
        const int pc = p < np ? p : np - 1;  // a pair that does not exist redoes the last one (no divergence around the transposes)
        long tc = tA + (hw >> 2);
        if (tc >= T1) tc = T1 - 1;
        const int ma = 2 * pc, mb = 2 * pc + 1;
        const bool b_ok = mb < M;
        const float *a1 = (tc >= 1 ? xs + (long)ma * a.mic_stride + (tc - 1) * kHop : hs + ma * kHop) + lane;
        const float *b1 = (!b_ok ? a.zeros : tc >= 1 ? xs + (long)mb * a.mic_stride + (tc - 1) * kHop : hs + mb * kHop) + lane;
        const float *a2 = xs + (long)ma * a.mic_stride + tc * kHop + lane;
        const float *b2 = (!b_ok ? a.zeros : xs + (long)mb * a.mic_stride + tc * kHop) + lane;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            re[j] = a1[32 * j];
            im[j] = b1[32 * j];
            re[j + 16] = a2[32 * j];
            im[j + 16] = b2[32 * j];
        }
    };
    const int n_rounds = (int)((T1 - T0 + 1) / 2);
    if (wv < 4) issue_inputs(T0);
    for (int r = 0; r < n_rounds; ++r) {
        const long tA = T0 + 2L * r;
        if (wv < 4) {  // ---- forward phase: (frame, pair) = (hw >> 2, hw & 3); the samples were requested at the end of the previous round
            const int p = hw & 3;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float4 hv = wrow[g];
                re[4 * g + 0] *= hv.x; im[4 * g + 0] *= hv.x;
                re[4 * g + 1] *= hv.y; im[4 * g + 1] *= hv.y;
                re[4 * g + 2] *= hv.z; im[4 * g + 2] *= hv.z;
                re[4 * g + 3] *= hv.w; im[4 * g + 3] *= hv.w;
            }
#ifdef BF_DAS_STAMPS
            unsigned long long st_acc[16], st_prev = 0;
#endif
            wt_fft_fwd_p2(re, im, lane, tw2, wbase, wrowp BF_STAMP_ARGS);
            if (p < np) {
                float4 *zp = s_z + ((hw >> 2) * 4 + p) * 16 * 32 + lane;
#pragma unroll
                for (int m = 0; m < 16; ++m) zp[m * 32] = float4{re[2 * m], im[2 * m], re[2 * m + 1], im[2 * m + 1]};
            }
        }
        // ---- direction phase: wavefront w takes directions 7 - w and 15 - w (the wavefronts without forward work get the second
        // direction first: 9-12 directions balance forward + one against two).  Gains come straight from global memory, eight
        // position pairs (16 entries, 32 registers) at a time, the next group requested before the current one is used; a
        // direction's first group is requested ahead of the barrier / of the previous direction's backward transform.  The table
        // offset is made opaque per (round, direction): otherwise LICM hoists the loop-invariant per-position addresses out of
        // the round loop as 64-bit VGPR pairs, they spill, and every gain load waits on a scratch reload (5.8 ms per 16
        // directions instead of 3.1)
        f32x2 ga[16], gb[16];
        auto gptr = [&](int dd) -> const f32x2 * {
            long goff = (long)(dir0 + dd * 8 + (7 - wv)) * np * 1024;
            asm volatile("" : "+s"(goff));
            return a.gains + goff + lane;
        };
        auto ldg = [&](const f32x2 *gd, f32x2(&g)[16], int p, int m0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                g[2 * k] = gd[(p * 32 + 2 * (m0 + k)) * 32];
                g[2 * k + 1] = gd[(p * 32 + 2 * (m0 + k) + 1) * 32];
            }
        };
        if (7 - wv < n_here) ldg(gptr(0), ga, 0, 0);
        __syncthreads();
        const long t = tA + half;
        const bool valid = t < T1;
#pragma unroll
        for (int dd = 0; dd < 2; ++dd) {
            const int dl = dd * 8 + (7 - wv);
            if (dl < n_here) {  // wave-uniform
                const int d = dir0 + dl;
                const f32x2 *gd = gptr(dd);
                const float4 *zf = s_z + half * 4 * 16 * 32 + lane;
#pragma unroll
                for (int i = 0; i < 32; ++i) re[i] = im[i] = 0.f;  // fma(g, z, 0) on the first pair, as das_fused_kernel
                for (int p = 0; p < np; ++p) {
                    ldg(gd, gb, p, 8);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float4 z = zf[(p * 16 + k) * 32];
                        const f32x2 g0 = ga[2 * k], g1 = ga[2 * k + 1];
                        re[2 * k] = bf_fma(-g0.y, z.y, bf_fma(g0.x, z.x, re[2 * k]));
                        im[2 * k] = bf_fma(g0.y, z.x, bf_fma(g0.x, z.y, im[2 * k]));
                        re[2 * k + 1] = bf_fma(-g1.y, z.w, bf_fma(g1.x, z.z, re[2 * k + 1]));
                        im[2 * k + 1] = bf_fma(g1.y, z.z, bf_fma(g1.x, z.w, im[2 * k + 1]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (p + 1 < np) ldg(gd, ga, p + 1, 0);
                    else if (dd == 0 && 8 + (7 - wv) < n_here) ldg(gptr(1), ga, 0, 0);  // the next direction's first group rides out the backward transform
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float4 z = zf[(p * 16 + 8 + k) * 32];
                        const f32x2 g0 = gb[2 * k], g1 = gb[2 * k + 1];
                        re[16 + 2 * k] = bf_fma(-g0.y, z.y, bf_fma(g0.x, z.x, re[16 + 2 * k]));
                        im[16 + 2 * k] = bf_fma(g0.y, z.x, bf_fma(g0.x, z.y, im[16 + 2 * k]));
                        re[16 + 2 * k + 1] = bf_fma(-g1.y, z.w, bf_fma(g1.x, z.z, re[16 + 2 * k + 1]));
                        im[16 + 2 * k + 1] = bf_fma(g1.y, z.z, bf_fma(g1.x, z.w, im[16 + 2 * k + 1]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                {
#ifdef BF_DAS_STAMPS
                    unsigned long long st_acc[16], st_prev = 0;
#endif
                    wt_fft_inv_p2(re, im, lane, tw2, wbase, wrowp BF_STAMP_ARGS);
                }
                // position i holds sample n = 32*brev5(i) + lane; even i -> first half of the frame, odd i -> second half
                float h[32];
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const float4 hv = wrow[g];
                    h[4 * g + 0] = hv.x; h[4 * g + 1] = hv.y; h[4 * g + 2] = hv.z; h[4 * g + 3] = hv.w;
                }
                float first[16], second[16], partner[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    first[q] = re[2 * q] * h[brev5(2 * q)];
                    second[q] = re[2 * q + 1] * h[brev5(2 * q + 1)];
                    partner[q] = halves_exchange(second[q], half);
                }
                float *ys = a.y + ((long)in_stream * a.n_dirs + d) * a.n_frames * kHop;
                if (valid) {
                    float *yo = ys + t * kHop + lane;
                    if (t == T0 && T0 > 0) {  // first hop of the run: the previous run adds its half separately (both into a zeroed hop)
#pragma unroll
                        for (int q = 0; q < 16; ++q) atomicAdd(yo + 32 * brev5(2 * q), first[q]);
                    } else {
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
#pragma clang fp contract(off)
                            yo[32 * brev5(2 * q)] = (half ? partner[q] : keep[dd][q]) + first[q];
                        }
                    }
                    if (t == T1 - 1) {
                        if (T1 < a.n_frames) {  // last frame of the run: its second half belongs to the next run's first hop
                            float *yn = ys + T1 * kHop + lane;
#pragma unroll
                            for (int q = 0; q < 16; ++q) atomicAdd(yn + 32 * brev5(2 * q), second[q]);
                        } else {  // end of the batch: carried state for the next call (OLA tail and the last input hop)
                            float *to = a.tail_out + ((long)in_stream * a.n_dirs + d) * kHop + lane;
#pragma unroll
                            for (int q = 0; q < 16; ++q) to[32 * brev5(2 * q)] = second[q];
                            if (dl == 0) {
                                float *ho = a.hist_out + (long)in_stream * M * kHop;
                                for (int m = 0; m < M; ++m)
                                    for (int j = 0; j < 16; ++j)
                                        ho[m * kHop + 32 * j + lane] = xs[(long)m * a.mic_stride + t * kHop + 32 * j + lane];
                            }
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) keep[dd][q] = partner[q];  // half 0: frame B's second half, for the next round
            }
        }
        // the next round's samples: in flight across the barrier (the wavefronts with one direction less wait there anyway)
        if (wv < 4 && r + 1 < n_rounds) issue_inputs(tA + 2);
        __syncthreads();  // the parked spectra are rewritten by the next round
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

template <int LAYOUT>
void launch_layout(const DasFusedArgs &a, unsigned blocks, hipStream_t stream) {
    const int np = (a.n_mics + 1) / 2;
    // planar input: pair loop unrolled for the exact pair count, next pair's loads issued from inside the gain loop
    if (a.group > 1) {  // frame groups (periods below 512): launch_das_fused checked the shape
#define BF_DAS_GRP(NPL_, UNR_)                                                                                                              \
    do {                                                                                                                                    \
        if (a.group == 2) BF_LAUNCH((das_fused_kernel<LAYOUT, NPL_, UNR_, 2>), dim3(blocks), dim3(kBlock), 0, stream, a);                   \
        else if (a.group == 4) BF_LAUNCH((das_fused_kernel<LAYOUT, NPL_, UNR_, 4>), dim3(blocks), dim3(kBlock), 0, stream, a);              \
        else BF_LAUNCH((das_fused_kernel<LAYOUT, NPL_, UNR_, 8>), dim3(blocks), dim3(kBlock), 0, stream, a);                                \
    } while (0)
        if constexpr (LAYOUT == 0) {  // planar: the unrolled pair loop up to 8 microphones
            if (np == 1) BF_DAS_GRP(1, 1); else if (np == 2) BF_DAS_GRP(2, 2); else if (np == 3) BF_DAS_GRP(4, 3); else if (np == 4) BF_DAS_GRP(4, 4);
            else BF_DAS_GRP(0, 0);    // > 8 microphones: gains from L2
        } else {
            if (np == 1) BF_DAS_GRP(1, 0); else if (np == 2) BF_DAS_GRP(2, 0); else if (np <= 4) BF_DAS_GRP(4, 0); else BF_DAS_GRP(0, 0);
        }
#undef BF_DAS_GRP
        return;
    }
    if (LAYOUT == 1 && (a.n_mics == 4 || a.n_mics == 8)) {  // 16-byte loads: two pairs per sample access
        if (a.n_mics == 4)  // one 16-byte load per sample = the whole sample: two frames per wavefront
            BF_LAUNCH((das_fused_il_kernel<2, 1>), dim3(blocks), dim3(kBlock), 0, stream, a);
        else                // 8 microphones: one frame per wavefront, each half-wavefront loads its 16 bytes of the 32-byte sample
            BF_LAUNCH(das_fused_il8_kernel, dim3(blocks), dim3(kBlock), 0, stream, a);
        return;
    }
    constexpr bool unr = LAYOUT == 0;
#define BF_DAS_GO(NPL_, UNR_) BF_LAUNCH((das_fused_kernel<LAYOUT, NPL_, (unr ? UNR_ : 0)>), dim3(blocks), dim3(kBlock), 0, stream, a)
    if (np <= 1) BF_DAS_GO(1, 1);
    else if (np <= 2) BF_DAS_GO(2, 2);
    else if (np == 3) BF_DAS_GO(4, 3);
    else if (np == 4) BF_DAS_GO(4, 4);
    else BF_DAS_GO(0, 0);  // > 8 mics: the gain tables no longer fit beside the transpose buffers and the tail ring
#undef BF_DAS_GO
}

// Sum of squares of every output stream (double accumulation): bf_stream_rms.
__global__ __launch_bounds__(256) void stream_sumsq_kernel(const float *y, long n, double *sumsq) {
    const float *ys = y + (long)blockIdx.y * n;
    double acc = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const double v = (double)ys[i];
        acc += v * v;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sumsq + blockIdx.y, part[0] + part[1] + part[2] + part[3]);
}

}  // namespace

// Runs are multiples of 16 frames; the first hop of every run but the first of a stream is completed by
// atomic adds and must be zero beforehand (prepare_das_fused).
hipError_t prepare_das_fused(const DasFusedArgs &a, hipStream_t stream) {
    if (a.chunks_per_stream > 1) {
        const int H = a.group > 1 ? kHop / a.group : kHop;  // frame groups: the hop of one frame of the period
        for (int s = 0; s < a.n_streams; ++s) {
            hipError_t e = hipMemset2DAsync(a.y + ((long)s * a.n_frames + a.frames_per_chunk) * H,
                                            (size_t)a.frames_per_chunk * H * sizeof(float), 0, H * sizeof(float),
                                            (size_t)a.chunks_per_stream - 1, stream);
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

// a.group = 2 / 4 / 8 (periods 256 / 128 / 64 as groups of interleaved frames): either layout, any microphone count,
// no spectrum dump; a.frames_per_chunk a multiple of 16 * group
bool das_fused_takes_groups(const DasFusedArgs &a) {
    return a.sdump == nullptr && (a.group == 2 || a.group == 4 || a.group == 8);
}

hipError_t launch_das_fused(const DasFusedArgs &a, hipStream_t stream) {
    if (a.group > 1 && !(das_fused_takes_groups(a) && a.frames_per_chunk % (16 * a.group) == 0)) return hipErrorInvalidValue;
    const unsigned blocks = (unsigned)((long)a.chunks_per_stream * a.n_streams);
    if (a.layout == 0) launch_layout<0>(a, blocks, stream);
    else launch_layout<1>(a, blocks, stream);
#ifdef BF_DAS_STAMPS
    {
        static int n_launch = 0;  // sums accumulate over back-to-back launches; read out only when asked (no sync otherwise)
        ++n_launch;
        if (getenv("BF_DAS_STAMPS_PRINT")) {
            (void)hipStreamSynchronize(stream);
            static unsigned long long hb[kStampBlocks][20];
            (void)hipMemcpyFromSymbol(hb, HIP_SYMBOL(g_stamps), sizeof(hb));
            unsigned long long h[18] = {0};
            const int nb = (int)(a.chunks_per_stream * a.n_streams) < kStampBlocks ? (int)(a.chunks_per_stream * a.n_streams) : kStampBlocks;
            for (int b = 0; b < nb; ++b)
                for (int i = 0; i < 18; ++i) h[i] += hb[b][i];
            void *sym = nullptr;
            (void)hipGetSymbolAddress(&sym, HIP_SYMBOL(g_stamps));
            (void)hipMemset(sym, 0, sizeof(hb));
            fprintf(stderr, "stamps over %d launches (cycles per wave-iteration, wave 0 of each block):", n_launch);
            for (int i = 0; i < 14; ++i) fprintf(stderr, " [%d]=%.0f", i, (double)h[i] / (double)(h[15] ? h[15] : 1));
            {
                unsigned long long r0min = ~0ull, r1max = 0, dmin = ~0ull, dmax = 0, dsum = 0, r0max = 0, r1min = ~0ull;
                for (int b = 0; b < nb; ++b) {
                    const unsigned long long d = hb[b][19] - hb[b][18];
                    r0min = hb[b][18] < r0min ? hb[b][18] : r0min; r0max = hb[b][18] > r0max ? hb[b][18] : r0max;
                    r1max = hb[b][19] > r1max ? hb[b][19] : r1max; r1min = hb[b][19] < r1min ? hb[b][19] : r1min;
                    dmin = d < dmin ? d : dmin; dmax = d > dmax ? d : dmax; dsum += d;
                }
                fprintf(stderr, "\n  last launch, loop of wave 0 per block (us): min %.1f mean %.1f max %.1f; first start -> last end %.1f; start spread %.1f, end spread %.1f",
                        dmin * 0.01, dsum * 0.01 / nb, dmax * 0.01, (r1max - r0min) * 0.01, (r0max - r0min) * 0.01, (r1max - r1min) * 0.01);
            }
            if (getenv("BF_DAS_STAMPS_BLOCKS")) {
                fprintf(stderr, "\n  per-block loop us:");
                for (int b = 0; b < nb; ++b) fprintf(stderr, " %.1f", (hb[b][19] - hb[b][18]) * 0.01);
            }
            const double nw = (double)n_launch * nb;
            fprintf(stderr, "  clock %.3f GHz  loop %.0f cycles/wave  kernel %.0f cycles/wave\n",
                    h[16] ? 0.1 * (double)h[14] / (double)h[16] : 0.0, (double)h[14] / nw, (double)h[17] / nw);
            n_launch = 0;
        }
    }
#endif
    return hipGetLastError();
}

// The 1024-frame period, a wavefront per frame (das_fused_wave2048_kernel): a.gains = das_pair_gains_natural tables [dir][pair][2048],
// a.twiddle = exp(-2 pi i m / 2048) for m < 1024, a.window = 2048 floats, a.zeros >= 1024 floats; a.frames_per_chunk a multiple of 8; the first
// hop of every run but the first of a stream must be zero beforehand (prepare_das_fused_wave2048); no spectrum dump
hipError_t prepare_das_fused_wave2048(const DasFusedArgs &a, hipStream_t stream) {
    if (a.chunks_per_stream > 1) {
        for (int s = 0; s < a.n_streams; ++s) {
            hipError_t e = hipMemset2DAsync(a.y + ((long)s * a.n_frames + a.frames_per_chunk) * kHop2, (size_t)a.frames_per_chunk * kHop2 * sizeof(float), 0,
                                            kHop2 * sizeof(float), (size_t)a.chunks_per_stream - 1, stream);
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}
hipError_t launch_das_fused_wave2048(const DasFusedArgs &a, hipStream_t stream) {
    if (a.sdump != nullptr || a.frames_per_chunk % kWaves2 != 0) return hipErrorInvalidValue;
    const unsigned blocks = (unsigned)((long)a.chunks_per_stream * a.n_streams);
    if (a.layout == 0) BF_LAUNCH(das_fused_wave2048_kernel<0>, dim3(blocks), dim3(kBlock), 0, stream, a);
    else BF_LAUNCH(das_fused_wave2048_kernel<1>, dim3(blocks), dim3(kBlock), 0, stream, a);
    return hipGetLastError();
}

// Look directions dir0 .. dir0 + n_here - 1 (n_here <= 16) of every input stream from one set of forward transforms; planar
// input, <= 8 microphones, no spectrum dump.  chunks_per_stream counts runs per INPUT stream here.
hipError_t launch_das_fused_dirs(const DasFusedArgs &a, int dir0, int n_here, hipStream_t stream) {
    if (a.layout != 0 || a.n_mics > 8 || n_here < 1 || n_here > 16 || a.sdump != nullptr) return hipErrorInvalidValue;
    const unsigned blocks = (unsigned)((long)a.chunks_per_stream * (a.n_streams / a.n_dirs));
    BF_LAUNCH(das_fused_dirs_kernel, dim3(blocks), dim3(kBlock), 0, stream, a, dir0, n_here);
    return hipGetLastError();
}

hipError_t launch_stream_rms(const float *y, long n_samples, int n_streams, double *sumsq, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(sumsq, 0, sizeof(double) * n_streams, stream);
    if (e != hipSuccess) return e;
    long bx = (n_samples + 256L * 64 - 1) / (256L * 64);
    if (bx < 1) bx = 1;
    if (bx > 1024) bx = 1024;
    BF_LAUNCH(stream_sumsq_kernel, dim3((unsigned)bx, (unsigned)n_streams), dim3(256), 0, stream, y, n_samples, sumsq);
    return hipGetLastError();
}

hipError_t launch_das_hermitian_dump(const f32x2 *sdump, f64x2 *out, long n_frames_total, hipStream_t stream) {
    const long total = n_frames_total * kNfft;
    BF_LAUNCH(das_hermitian_dump_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, sdump, out,
                       total);
    return hipGetLastError();
}

}  // namespace bf
