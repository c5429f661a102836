// pipeline_kernels.hpp -- argument blocks and launchers of the fp64 bin-pipeline kernels.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/bfcore.h"
#include "geometry.hpp"

namespace bf {

// Problems per frame handed to the per-bin kernels: q = 0..512 is FFT bin q, q = 513 is bin
// 513.  Bins 514..1023 are the exact conjugates of bins 510..1 (real input, conjugate-
// symmetric steering) and are never computed; 511/513 are NOT conjugates of each other
// because of the reference's frequency-vector quirk Q1 (util.h:198), hence the extra slot.
constexpr int kNQ = 514;
constexpr int kYhStride = 516;  // f64x2 per frame in the per-bin output buffer (16-byte friendly)
constexpr int kMpfVecs = 7;     // S_prev, S_tmp, S_min, lambda_noise, Z, rev0, rev1 (phasempf.cpp:67-76)

struct StftArgs {
    const float *x;
    const float *hist;  // hop before frame 0, layout as x
    f64x2 *Z;           // [stream][frames_ws][NP][1024]
    const f64x2 *tw;
    const double *win;
    long n_frames, frames_ws, frame_off, mic_stride, stream_stride_x;
    int n_streams, n_mics, layout;
    int n_fft_mics;        // channels actually transformed (= n_mics; 1 for the single-channel mcra node)
    int skip_lo, skip_hi;  // packed-spectrum bins in (skip_lo, skip_hi) are never read by the per-bin kernel: not stored
};
hipError_t launch_stft(const StftArgs &a, int n_cus, hipStream_t s);

struct BinsArgs {
    const f64x2 *Z;      // [stream][frames_ws][NP][1024]
    f64x2 *Yh;           // [stream][n_frames][kYhStride]: y_fft of problems q = 0..513
    f64x2 *spectrum;     // nullable: [stream][n_frames][1024] full y_fft dump
    const f64x2 *steer;  // [dir][col][mic][1024]
    const double *freqs; // [1024]
    long n_frames, frames_ws, frame_off;
    int n_streams, n_mics, kp1;  // n_streams = OUTPUT streams (input streams * n_dirs)
    int n_dirs;                  // look directions per input stream: Z is indexed by stream / n_dirs, steer by stream % n_dirs
    long steer_dir_stride;       // f64x2 elements between the steering tables of two look directions
    bf_config cfg;
    f64x2 *gssW;         // [stream][1024][kp1][n_mics]
    double *mpf;         // [stream][kMpfVecs*1024 + 8] (the mcra node uses vectors 0..3 and the two scalars)
    unsigned long long gss_reset_mask;  // bit d: look direction d re-initialises W = C^H (gss.cpp:90-93) in this batch
};
hipError_t launch_bins(const BinsArgs &a, int n_cus, hipStream_t s);
// per-node launchers behind launch_bins (one translation unit per kernel family)
hipError_t launch_pointwise(const BinsArgs &a, hipStream_t s);              // das (fp64), phase: mask_kernels.hip
hipError_t launch_phasempf(const BinsArgs &a, int n_cus, hipStream_t s);    // mask_kernels.hip
hipError_t launch_mcra_node(const BinsArgs &a, hipStream_t s);              // mask_kernels.hip
hipError_t launch_mvdr_lcmv(const BinsArgs &a, int n_cus, hipStream_t s);   // cov_kernels.hip
hipError_t launch_gss(const BinsArgs &a, int n_cus, hipStream_t s);         // gsc_gss_kernels.hip
hipError_t launch_gsc_align(const BinsArgs &a, hipStream_t s);              // gsc_gss_kernels.hip
hipError_t launch_expand_spectrum(const f64x2 *Yh, f64x2 *spectrum, long frames, hipStream_t s);  // stft_istft.hip

struct IstftArgs {
    const f64x2 *Yh;
    float *y;              // [stream][n_frames*512]
    const float *tail_in;  // [stream][512]
    float *tail_out;
    const f64x2 *tw;
    const f32x2 *tw32;     // non-null: backward FFT in fp32, one frame per transform (istft32_kernel)
    const double *win;
    long n_frames;
    int n_streams;
    double post_amp;
    int use_post_amp;
};
hipError_t launch_istft(const IstftArgs &a, int n_cus, hipStream_t s);

// phasempf.cpp:331-334: moving average over the output samples, state = last 63 raw samples
// gsc.cpp:120-181: the sample-serial float32 NLMS sidelobe canceller over the phase-aligned microphone signals.
// aligned = [stream][mic][n_samples] (ISTFT output of the align pass), y = [stream][n_samples],
// state = [stream][(2*(M-1) + 1) * filter_size] floats: block_matrix rows, filter rows, last_outputs (reference order).
hipError_t launch_gsc_nlms(const float *aligned, float *y, float *state, long n_samples, int n_streams, int n_mics,
                           const bf_config &cfg, hipStream_t s);

hipError_t launch_smooth(const float *yraw, float *y, double *state, long n_frames, int n_streams, int smooth_size,
                         hipStream_t s);

}  // namespace bf
