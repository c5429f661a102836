// pipeline_kernels.hpp -- argument blocks and launchers of the fp64 bin-pipeline kernels.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/bfcore.h"
#include "geometry.hpp"

namespace bf {

// FFT size of a translation unit.  The kernel files (stft_istft / mask / cov / gsc_gss / pipeline_kernels .hip) are compiled
// once per supported hop -- -DBF_NFFT=128 ... 8192, i.e. JACK periods of 64 ... 4096 frames (rosjack.cpp:131,
// util.h:261 fft_win = 2 * window) -- into namespaces bf::n128 ... bf::n8192; the host side (pipeline.hip) is
// compiled once and picks a KernelSet by the configured hop.
#ifdef BF_NFFT
#define BF_CAT2_(a, b) a##b
#define BF_CAT2(a, b) BF_CAT2_(a, b)
#define BF_NTAG BF_CAT2(n, BF_NFFT)
#endif

// Problems per frame handed to the per-bin kernels: q = 0..N/2 is FFT bin q, q = N/2+1 is bin
// N/2+1.  Bins N/2+2..N-1 are the exact conjugates of bins N/2-2..1 (real input, conjugate-
// symmetric steering) and are never computed; N/2-1 and N/2+1 are NOT conjugates of each other
// because of the reference's frequency-vector quirk Q1 (util.h:198), hence the extra slot.
constexpr int problems_per_frame(int nfft) { return nfft / 2 + 2; }   // 514 at N = 1024
constexpr int yh_stride(int nfft) { return nfft / 2 + 4; }            // f64x2 per frame in the per-bin output buffer: 516
constexpr int kMpfVecs = 7;     // S_prev, S_tmp, S_min, lambda_noise, Z, rev0, rev1 (phasempf.cpp:67-76)
#ifdef BF_NFFT
namespace BF_NTAG {
constexpr int kN = BF_NFFT;
constexpr int kHop = kN / 2;
constexpr int kNQ = problems_per_frame(kN);
constexpr int kYhStride = yh_stride(kN);
constexpr int kQX = kN / 2 + 1;   // the extra problem: bin N/2+1, whose X is the conjugate of bin N/2-1's
}  // namespace BF_NTAG
#endif

// Packed pair spectrum element of the covariance nodes (mvdr / lcmv): the top 48 bits of each component of a complex double
// (sign, exponent, 36 mantissa bits; round to nearest), 12 bytes instead of 16.  Every in-band spectrum crosses the fabric three
// times (stored once, read as the newest and as the oldest frame of the sliding covariance), so the element size sets the
// chain's traffic; 2^-37 relative on X moves the solved spectrum by < 1e-10 relative L2 at cond(R) = 3e4 (tools/z48_precision.py),
// and the magnitude gate (mvdr.cpp:85) flips with probability ~1e-11 per bin-frame.  Decoding is two 32-bit operations.
// Stored HALVED like the c128 spectra of these nodes (StftArgs::halve: the stft kernel scales its window by 1/2: exact), so that unpacking a microphone pair is X_a = Z[k] + conj Z[N-k],
// X_b = -i (Z[k] - conj Z[N-k]) without the two multiplications.
struct z48 {
    unsigned lo, re_hi, im_hi;  // lo = (re's low dword & 0xFFFF0000) | (im's low dword >> 16)
};

struct StftArgs {
    const float *x;
    const float *hist;  // hop before frame 0, layout as x
    f64x2 *Z;           // [stream][frames_ws][NP][N] f64x2 -- or z48 elements (reinterpret) when `z48` is set
    const f64x2 *tw;
    const double *win;
    long n_frames, frames_ws, frame_off, mic_stride, stream_stride_x;
    int n_streams, n_mics, layout;
    int n_fft_mics;        // channels actually transformed (= n_mics; 1 for the single-channel mcra node)
    int skip_lo, skip_hi;  // packed-spectrum bins in (skip_lo, skip_hi) are never read by the per-bin kernel: not stored
    int z48;               // store z48 elements (mvdr / lcmv)
    int halve = 0;         // store the packed pair spectra halved (exact; mvdr / lcmv, z48 or c128): the per-bin kernels unpack X_a = Z[k] + conj Z[N-k],
                           // X_b = -i (Z[k] - conj Z[N-k]) without the two multiplications
    int run_len;           // consecutive frames one half-wavefront walks (N = 1024): the shared hop stays in registers
    const f64x2 *tw_w64 = nullptr;  // twiddle_table_w64_rot (N = 1024): stft_bins_w64_kernel
};

struct BinsArgs {
    const f64x2 *Z;      // [stream][frames_ws][NP][N] f64x2 (z48 elements for mvdr / lcmv: reinterpret)
    f64x2 *Yh;           // [stream][n_frames][kYhStride]: y_fft of problems q = 0..N/2+1 (f32x2 rows when yh32)
    int yh32;            // mvdr / lcmv in front of the fp32 backward transform: rows are f32x2 and only the problems
                         // 0 and yh_lo..yh_hi are written (the rest is zero by definition: mvdr.cpp:103) -- istft32 knows
    int yh_lo, yh_hi;
    f64x2 *spectrum;     // nullable: [stream][n_frames][N] full y_fft dump
    const f64x2 *steer;  // [dir][col][mic][N]
    const double *freqs; // [N]
    long n_frames, frames_ws, frame_off;
    int n_streams, n_mics, kp1;  // n_streams = OUTPUT streams (input streams * n_dirs)
    int n_dirs;                  // look directions per input stream: Z is indexed by stream / n_dirs, steer by stream % n_dirs
    long steer_dir_stride;       // f64x2 elements between the steering tables of two look directions
    bf_config cfg;
    f64x2 *gssW;         // [stream][N][kp1][n_mics]
    double *mpf;         // [stream][kMpfVecs*N + 8] (the mcra node uses vectors 0..3 and the two scalars)
    unsigned long long gss_reset_mask;  // bit d: look direction d re-initialises W = C^H (gss.cpp:90-93) in this batch
    int z48;             // mvdr / lcmv: Z holds z48 elements (the default); 0 = full f64x2 spectra (BF_Z48=0: parity debugging on
                         // ill-conditioned scenes; only the group-per-problem kernel reads them)
    // phasempf with many streams (N = 1024, default precision, no dump): the recursion kernel runs the backward transform too
    // (mpf_rec_istft_kernel: one block per output stream; y_fft rows never reach HBM).  Set by the pipeline, which then skips its ISTFT launch.
    int rec_istft = 0;
    float *rec_y = nullptr;               // [stream][n_frames * hop]
    const float *rec_tail_in = nullptr;   // [stream][hop]
    float *rec_tail_out = nullptr;
    const f64x2 *rec_tw_w64 = nullptr;    // twiddle_table_w64_rot
    const double *rec_win = nullptr;
    int mpf32;           // phasempf in front of the fp32 backward transform: the recursion leaves y_fft as f32x2 rows in the slots of its
                         // |out_int|^2 input (8 bytes each, same [stream][frame][kYhStride] layout, behind the f64x2 rows)
};
struct IstftArgs {
    const f64x2 *Yh;
    int yh32;              // rows are f32x2; problems outside {0} + [yh_lo, yh_hi] are zero and were never written
    int yh_lo, yh_hi;
    float *y;              // [stream][n_frames*hop]
    const float *tail_in;  // [stream][hop]
    float *tail_out;
    const f64x2 *tw;
    const f32x2 *tw32;     // non-null: backward FFT in fp32, one frame per transform (istft32_kernel; N = 1024 only)
    const f64x2 *tw_w64 = nullptr;  // non-null (N = 1024): twiddle_table_w64_rot, the fp64 backward transform runs istft_w64_kernel
    float *frames;         // N != 1024: [stream][n_frames][N] windowed frames (generic kernel), overlap-added by a second pass
    const double *win;
    long n_frames;
    int n_streams;
    double post_amp;
    int use_post_amp;
};
// das at the reference's precision in ONE launch (N = 1024, <= 8 microphones, one look direction, no spectrum dump): das_f64_w64.hip
struct DasF64Args {
    const float *x;
    const float *hist;     // hop before frame 0 [stream][mic][hop]
    float *y;              // [stream][n_frames*hop]
    const float *tail_in;  // [stream][hop]
    float *tail_out;
    const f64x2 *gains;    // das_pair_gains_w64_f64: Hermitian-part pair gains in the 64-lane kernel's register / lane order, 1/N folded in
    const f64x2 *tw;
    const double *win;
    long n_frames, mic_stride, stream_stride_x;
    int n_streams, n_mics, run_len;
    int layout = 0;        // bf_layout of x and hist; 1 (interleaved) only with launch_das_f64_w64
    float *hist_out = nullptr;         // das_f64_pair_kernel: receives the last hop of the batch (the ring-buffer carry), layout as hist
    const f64x2 *gains_mic = nullptr;  // das_mic_gains_w64_f64: per-microphone Hermitian gains of the frame-pair kernel (planar input)
    // das_f64_pair_kernel: the microphones that get a forward transform, in the order the kernel walks them (slot k -> microphone slot_mic[k],
    // never microphone 0), and at most one more microphone whose weight row is bitwise identical to slot 0's (extra_mic, -1 = none): it
    // shares slot 0's transform (das.cpp:60-63 is linear in the microphones; the reference drops z, so aira16's microphones 1 and 7 coincide)
    int n_tr = 0, extra_mic = -1;
    int slot_mic[8] = {1, 2, 3, 4, 5, 6, 7, 0};
    int mic0_unit = 0;                 // the weight row of microphone 0 is identically 1 (das.cpp:33-38): das_f64_pair_kernel adds h x_0 / M in the time domain
    void *sched_ws = nullptr;          // das_f64_pair_kernel: device workspace of its work queue (das_f64_sched_ws_bytes())
    size_t sched_ws_bytes = 0;
    float *ring = nullptr;             // das_f64_pair_kernel on [sample][mic] input: the blocks' hop rings (das_f64_ring_bytes())
    size_t ring_bytes = 0;
};

// the same node on one full wavefront per frame (das_f64_w64.hip; N = 1024 only): `gains` = das_pair_gains_w64_f64, `tw` =
// twiddle_table_w64_rot.  prepare_ zeroes the run-boundary hops of y on `s` (they are completed by atomic adds) and must precede the
// launch.  hipErrorNotSupported above 8 microphones.
hipError_t prepare_das_f64_w64(const DasF64Args &a, int n_cus, hipStream_t s);
hipError_t launch_das_f64_w64(const DasF64Args &a, int n_cus, hipStream_t s);
size_t das_f64_sched_ws_bytes();
size_t das_f64_ring_bytes(int n_mics, int n_cus);  // 0: this microphone count has no ring kernel (the transposition in front of the planar kernel serves it)
// [stream][n][M] -> [stream][M][n] in front of the frame-pair kernel ([sample][mic] handles); n a multiple of 256, M <= 8
hipError_t launch_interleaved_to_planar(const float *x, float *out, long n, int n_mics, int n_streams, hipStream_t s);
bool das_f64_writes_hist(const DasF64Args &a);  // the kernel launch_das_f64_w64 picks stores a.hist_out itself (no copy behind it)

#ifdef BF_NFFT
namespace BF_NTAG {
hipError_t launch_stft(const StftArgs &a, int n_cus, hipStream_t s);
hipError_t launch_bins(const BinsArgs &a, int n_cus, hipStream_t s);
// stft + per-bin stage in one launch for the nodes without a frame history (das fp64, phase, phasempf; N = 1024, <= 8 mics,
// one look direction): the spectra stay in LDS.  hipErrorNotSupported = run launch_stft + launch_bins instead.  mask_kernels.hip
hipError_t launch_stft_bins_fused(const StftArgs &a, const BinsArgs &b, int n_cus, hipStream_t s);
// per-node launchers behind launch_bins (one translation unit per kernel family)
hipError_t launch_pointwise(const BinsArgs &a, hipStream_t s);              // das (fp64), phase: mask_kernels.hip
hipError_t launch_phasempf(const BinsArgs &a, int n_cus, hipStream_t s);    // mask_kernels.hip
hipError_t launch_mcra_node(const BinsArgs &a, hipStream_t s);              // mask_kernels.hip
hipError_t launch_mvdr_lcmv(const BinsArgs &a, int n_cus, hipStream_t s);   // cov_kernels.hip
hipError_t launch_gss(const BinsArgs &a, int n_cus, hipStream_t s);         // gsc_gss_kernels.hip
hipError_t launch_gsc_align(const BinsArgs &a, hipStream_t s);              // gsc_gss_kernels.hip
hipError_t launch_expand_spectrum(const f64x2 *Yh, f64x2 *spectrum, long frames, hipStream_t s);  // stft_istft.hip
hipError_t launch_istft(const IstftArgs &a, int n_cus, hipStream_t s);
// gsc.cpp:120-181: the sample-serial float32 NLMS sidelobe canceller over the phase-aligned microphone signals.
// aligned = [stream][mic][n_samples] (ISTFT output of the align pass), y = [stream][n_samples],
// state = [stream][(2*(M-1) + 1) * filter_size] floats: block_matrix rows, filter rows, last_outputs (reference order).
hipError_t launch_gsc_nlms(const float *aligned, float *y, float *state, long n_samples, int n_streams, int n_mics,
                           const bf_config &cfg, hipStream_t s);
// phasempf.cpp:331-334: moving average over the output samples, state = last 63 raw samples
hipError_t launch_smooth(const float *yraw, float *y, double *state, long n_frames, int n_streams, int smooth_size,
                         hipStream_t s);
}  // namespace BF_NTAG
#endif

// What the host pipeline calls: the launchers of one FFT size.
struct KernelSet {
    int nfft;
    hipError_t (*stft)(const StftArgs &, int, hipStream_t);
    hipError_t (*bins)(const BinsArgs &, int, hipStream_t);
    hipError_t (*stft_bins)(const StftArgs &, const BinsArgs &, int, hipStream_t);
    hipError_t (*istft)(const IstftArgs &, int, hipStream_t);
    hipError_t (*smooth)(const float *, float *, double *, long, int, int, hipStream_t);
    hipError_t (*gsc_nlms)(const float *, float *, float *, long, int, int, const bf_config &, hipStream_t);
};
const KernelSet *kernel_set_n128();
const KernelSet *kernel_set_n256();
const KernelSet *kernel_set_n512();
const KernelSet *kernel_set_n1024();
const KernelSet *kernel_set_n2048();
const KernelSet *kernel_set_n4096();
const KernelSet *kernel_set_n8192();
inline const KernelSet *kernel_set(int nfft) {
    switch (nfft) {
        case 128: return kernel_set_n128();
        case 256: return kernel_set_n256();
        case 512: return kernel_set_n512();
        case 1024: return kernel_set_n1024();
        case 2048: return kernel_set_n2048();
        case 4096: return kernel_set_n4096();
        case 8192: return kernel_set_n8192();
        default: return nullptr;
    }
}

}  // namespace bf
