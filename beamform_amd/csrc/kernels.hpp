// kernels.hpp -- host-visible launchers of the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>

#include "geometry.hpp"

namespace bf {

// ---- fused fp32 DAS (das_fused.hip) ------------------------------------------
struct DasFusedArgs {
    const float *x;        // input samples, layout per `layout`
    const float *hist_in;  // [stream][mic][hop] (planar) or [stream][hop][mic]: the hop before frame 0
    float *hist_out;       // same layout: the last hop of this batch (ring-buffer carry, util.h:305-308)
    float *y;              // [stream][n_frames*hop]
    const float *tail_in;  // [stream][hop] second half of the frame before frame 0 (out_buff[0], util.h:302)
    float *tail_out;       // [stream][hop] second half of the last frame of this batch
    const f32x2 *gains;    // das_pair_gains() tables, one per look direction: [dir][pair][1024]
    const f32x2 *twiddle;  // twiddle_table_32x32<f32x2>()
    const float *window;   // sqrt-Hann, fp32, natural order [1024]
    const float *zeros;    // >= 1024 zero floats: partner channel of the last mic when n_mics is odd
    f32x2 *sdump;          // nullable: [stream][frame][1024] accumulated pair spectrum S (1/N folded in)
    long n_frames;         // frames per stream
    long mic_stride;       // planar: samples between mics of one stream
    long stream_stride_x;  // samples between streams in x
    int n_streams;         // OUTPUT streams = input streams * n_dirs (stream = input * n_dirs + dir)
    int n_dirs;            // look directions per input stream (>= 1); x / hist are indexed by the input stream
    int n_mics;
    int frames_per_chunk;
    int chunks_per_stream;
    int layout;            // bf_layout
    int group = 1;         // launch_das_fused only: R = 1024 / n_fft frames of a period below 512 interleaved per unit of work (1: period 512)
};
hipError_t prepare_das_fused(const DasFusedArgs &a, hipStream_t stream);  // zero the atomically-completed hops
hipError_t launch_das_fused(const DasFusedArgs &a, hipStream_t stream);
bool das_fused_takes_groups(const DasFusedArgs &a);
// the 1024-frame period on a full wavefront per 2048-point frame (das_fused.hip das_fused_wave2048_kernel): the generic kernel's tables
hipError_t prepare_das_fused_wave2048(const DasFusedArgs &a, hipStream_t stream);
hipError_t launch_das_fused_wave2048(const DasFusedArgs &a, hipStream_t stream);  // whether launch_das_fused runs this shape with a.group > 1
// look directions dir0 .. dir0 + n_here - 1 (n_here <= 16) from ONE set of forward transforms per frame (planar, <= 8 microphones, no
// spectrum dump); a.chunks_per_stream / frames_per_chunk describe the runs of an INPUT stream; output equal to launch_das_fused's within float rounding, not bit for bit
// (the window products are fused differently: a beam's low-order bits may change when n_dirs crosses BF_DAS_SHARED_DIRS)
hipError_t launch_das_fused_dirs(const DasFusedArgs &a, int dir0, int n_here, hipStream_t stream);

// S dump -> Hermitian part of the reference's y_fft as double2 [frames][1024]
// per-stream root-mean-square of y [n_streams][n_samples] -> rms[n_streams] (double); `sumsq` = n_streams doubles of scratch
hipError_t launch_stream_rms(const float *y, long n_samples, int n_streams, double *sumsq, hipStream_t stream);

hipError_t launch_das_hermitian_dump(const f32x2 *sdump, f64x2 *out, long n_frames_total, hipStream_t stream);

// JACK periods 256 / 1024 (FFT 512 / 2048), das_fused_gen.hip: same argument block; `gains` = das_pair_gains_natural tables [dir][pair][n_fft],
// `twiddle` = exp(-2 pi i m / n_fft) for m < n_fft / 2, `window` n_fft floats; frames_per_chunk is free (no multiple of 16), no
// prepare step (a run that does not start the stream recomputes its previous frame); sdump rows are n_fft long, natural order
hipError_t launch_das_fused_gen(const DasFusedArgs &a, int n_fft, hipStream_t stream);
hipError_t launch_das_hermitian_dump_gen(const f32x2 *sdump, f64x2 *out, long n_frames_total, int n_fft, hipStream_t stream);

}  // namespace bf
