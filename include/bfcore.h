/*
 * bfcore.h -- C ABI of the MI355X beamforming core (libbfcore.so).
 *
 * Drop-in boundary for the per-callback hot path of balkce/beamform:
 *   STFT (sqrt-Hann, 50 % hop) -> per-bin complex weighting -> ISTFT + overlap-add
 * as driven by every node's jack_callback (das, mvdr, lcmv, gss, phase, phasempf).
 * Plain C types only; device buffers are raw HIP device pointers.
 *
 * Each entry point names the reference interface it replaces (file:line relative
 * to the reference root).  All functions return 0 on success or a negative
 * BF_E* code; nothing here exits the process, prints, or keeps global state
 * (the reference keeps everything in file-scope globals, util.h:24-50).
 *
 * There is NO CPU fallback: bf_create fails with BF_ENODEV when no HIP device is
 * usable.
 */
#ifndef BFCORE_H
#define BFCORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BF_MAX_MICS 32
#define BF_MAX_INTERF 15 /* beamform_config.yaml:43-57 lists angle_interf1..15 */

/* Which node's apply_weights() runs between the STFT and the ISTFT. */
enum bf_algo {
    BF_DAS = 0,      /* das.cpp:47-70 */
    BF_MVDR = 1,     /* mvdr.cpp:62-115 */
    BF_LCMV = 2,     /* lcmv.cpp:88-140 */
    BF_GSS = 3,      /* gss.cpp:96-156 */
    BF_PHASE = 4,    /* phase.cpp:70-134 */
    BF_PHASEMPF = 5, /* phasempf.cpp:193-302 + :331-334 */
    BF_GSC = 7,      /* gsc.cpp:54-197: per-microphone phase alignment (STFT -> conj(w_m) -> ISTFT, do_overlap_bymic) followed by
                        the sample-serial float32 NLMS sidelobe canceller; uses the gsc_* fields */
    BF_MCRA = 6      /* mcra.cpp:64-155: single-channel MCRA noise subtraction (channel 0 only; uses mcra_*, out_amp,
                        out_only_noise) */
};

enum bf_error {
    BF_OK = 0,
    BF_EINVAL = -22,   /* bad argument / inconsistent config */
    BF_ENOMEM = -12,   /* host or device allocation failed */
    BF_ENODEV = -19,   /* no usable HIP device (no CPU fallback exists) */
    BF_ENOSYS = -38,   /* algorithm/variant not built into this library */
    BF_EIO = -5,       /* HIP runtime error; see bf_last_error() */
    BF_ENOENT = -2     /* config file not found */
};

/* How bf_process_batch* lays out multichannel input. */
enum bf_layout {
    BF_PLANAR = 0,      /* [stream][mic][sample]  -- what JACK hands the callback (rosjack.cpp:538-547) */
    BF_INTERLEAVED = 1  /* [stream][sample][mic]  -- interleaved frame buffer.  das in double (BF_DAS_F64, period 512): with 2, 4 or 8 microphones
                           the kernel transposes hop by hop into per-block rings (160 MB of device scratch on a 256-CU chip, allocated on first
                           use, kept); with 3, 5, 6 or 7 the batch is transposed on the device into a planar scratch of the batch's size in front
                           of the planar kernel; every other node reads the layout directly */
};

/* Arithmetic of the das node.  The reference computes in std::complex<double> end to end (das.cpp:16-24,47-70): BF_DAS_F64 is the
 * default (bf_config_init) and what the bench headline measures.  Other algorithms always compute in double. */
enum bf_das_impl {
    BF_DAS_FUSED_F32 = 0, /* opt-in: one fused kernel in fp32 arithmetic (narrower than the reference's; 1.5e-7 from it, inside the 1e-5
                             bar); also the only das path with register-resident kernels at JACK periods other than 512 and with
                             look directions that share their forward transforms */
    BF_DAS_F64 = 1        /* default: double arithmetic like das.cpp.  Period 512, <= 8 microphones, one look direction, no spectrum
                             dump: ONE launch (das_f64_pair_kernel: planar input; das_f64_w64_kernel: [sample][mic]); anything else
                             runs STFT -> per-bin kernel -> ISTFT and can dump the full N-bin spectrum */
};

/* Arithmetic of what lies between the transforms (every node computes its per-bin stage in double).
 * BF_PRECISION_REFERENCE (bf_config_init's default): nothing is narrower than the reference's std::complex<double>: spectra cross HBM as
 * complex doubles, the backward transform runs in double (mvdr.cpp:76-115 between fftw_execute(x_forward) and fftw_execute(y_inverse)).
 * BF_PRECISION_MIXED (opt-in, throughput): mvdr / lcmv park their spectra as 12-byte elements (36-bit mantissa: < 1e-10 on the solved
 * spectrum at cond(R) = 3e4) and every node whose per-bin stage can emit float rows runs the backward transform in fp32 (1.1e-7 on the
 * output instead of bit-identity with the double path).  Both are inside the 1e-5 parity bar; only REFERENCE is the reference's arithmetic. */
enum bf_precision {
    BF_PRECISION_REFERENCE = 0,
    BF_PRECISION_MIXED = 1
};

/*
 * Everything a node reads from the ROS parameter server
 * (handle_params util.h:52-134 + each node's *_handle_params) plus the two
 * JACK-server facts rosjack_create() records (rosjack.cpp:131-134).
 * bf_config_init() fills the launch-file values.
 */
typedef struct bf_config {
    int algo;                      /* enum bf_algo */
    int n_mics;                    /* number_of_microphones (util.h:122) */
    int hop;                       /* rosjack_window_size = the JACK period (rosjack.cpp:131); fft_win = 2*hop (util.h:261).
                                      Any power of two from 64 to 4096 (what jackd -p accepts in that range).  512 is the tuned
                                      shape (in-register FFT-1024 kernels); 64 ... 256 and 1024 run on the same register-resident
                                      machinery (several short frames per transform, or two FFT-1024 per long frame) for das
                                      (with the BF_DAS_FUSED_F32 opt-in: one fused fp32 kernel) and for the fp64 bin pipeline of das in double and of every other node;
                                      2048 and 4096 on LDS-staged transforms (radix-4 autosort; radix-2 in place at 4096) */
    double sample_rate;            /* rosjack_sample_rate */
    double mic_x[BF_MAX_MICS];     /* RAW mic<i>.x / .y from beamform_config.yaml (util.h:82-92) */
    double mic_y[BF_MAX_MICS];
    double theta;                  /* initial_angle, degrees (util.h:68-73) */
    int n_interf;                  /* number of angle_interf<k> with |angle| <= 180 (util.h:101-113) */
    double interf_angle[BF_MAX_INTERF];
    int verbose;
    /* mvdr / lcmv / gss */
    int past_windows;              /* mvdr.cpp:151-157 */
    double freq_mag_threshold;     /* mvdr.cpp:159-164 */
    double freq_max, freq_min;     /* mvdr.cpp:166-178 */
    double out_amp;                /* mvdr.cpp:180-185 (also phasempf.cpp:448-453) */
    double interf_angle_threshold; /* lcmv.cpp:212-217 */
    double mu, lambda_;            /* gss.cpp:219-231 */
    /* phase */
    double min_phase;              /* phase.cpp:169-175, phasempf.cpp:359-365 */
    double mag_mult, mag_threshold;/* phase.cpp:177-189 */
    /* phasempf */
    double min_mag;                /* phasempf.cpp:367-372 */
    int smooth_size;               /* phasempf.cpp:374-383 */
    double mcra_alphaS, mcra_alphaD, mcra_alphaD2, mcra_delta; /* phasempf.cpp:385-411 */
    int mcra_L;                    /* phasempf.cpp:413-418 */
    double mpf_alphaS, mpf_eta, mpf_rev_gamma, mpf_rev_delta;  /* phasempf.cpp:420-446 */
    double noise_floor;            /* phasempf.cpp:455-460 */
    int out_only_noise, out_only_mcra; /* phasempf.cpp:462-474 */
    /* build-specific (no reference counterpart) */
    int device;                    /* HIP device ordinal */
    int n_streams;                 /* independent audio streams per batch (each = one reference node's state) */
    int layout;                    /* enum bf_layout for bf_process_batch* input */
    int das_impl;                  /* enum bf_das_impl; bf_config_init: BF_DAS_F64 (the reference's arithmetic) */
    int precision;                 /* enum bf_precision; bf_config_init: BF_PRECISION_REFERENCE */
    int n_dirs;                    /* look directions evaluated per input stream from the SAME samples (0/1 = one, the
                                      reference node).  Output stream index = stream * n_dirs + dir.  Every node except mcra
                                      (no look direction) and gsc; gss / phasempf keep their recursive state per beam.
                                      das (fp32, planar input, <= 8 microphones): from 6 directions on the forward transforms
                                      of a frame are computed once and shared by the beams (das.cpp:51-63 transforms once).
                                      SURVEY 8(e) "look directions" / 8(f) row 4 */
    /* gsc (gsc.cpp:199-256, launch/gsc.launch:6-11); write_mu is file I/O and not part of the path */
    int gsc_use_vad;
    double gsc_vad_threshold, gsc_mu0, gsc_mu_max;
    int gsc_filter_size;           /* 1..256 */
} bf_config;
#define BF_MAX_DIRS 64

typedef struct bf_handle bf_handle;

/* Library identification / diagnostics. */
const char *bf_version(void);
const char *bf_strerror(int code);
/* Last HIP/runtime error text recorded on this handle (or globally when h==NULL). */
const char *bf_last_error(const bf_handle *h);
/* Number of HIP devices visible; <= 0 means the product cannot run. */
int bf_device_count(void);

/* Launch-file defaults for `algo` and the uncommented geometry of
 * beamform_config.yaml:15-17 (aira3).  Replaces the getParam fallbacks. */
int bf_config_init(bf_config *cfg, int algo);
/* Parse a beamform_config.yaml-style file (util.h:61-113 keys: verbose,
 * initial_angle, mic<i>: {id,x,y[,z]}, angle_interf<k>) plus the flat
 * per-node keys the launch files set (past_windows, freq_max, ...). */
int bf_config_load_yaml(bf_config *cfg, const char *path);
/* Same parser on an in-memory string. */
int bf_config_parse_yaml(bf_config *cfg, const char *text);

/* main(): prepare_overlap_and_add + buffer/plan allocation + update_weights(true)
 * (das.cpp:119-140 and the same block in every node). */
int bf_create(const bf_config *cfg, bf_handle **out);
void bf_destroy(bf_handle *h);

/* theta_roscallback: angle = msg->data; update_weights() (das.cpp:94-99).
 * Thread-safe against a concurrent bf_process_*: takes effect at the next
 * hop/batch (the reference updates in place with no lock). */
int bf_set_theta(bf_handle *h, double degrees);
/* Look-direction batch: direction `dir` (0 <= dir < n_dirs) of every input stream gets its own /theta;
 * bf_set_theta() is bf_set_theta_dir(h, 0, deg).  bf_set_thetas sets directions 0..n-1 with one table rebuild.
 * Every direction starts at bf_config.theta. */
int bf_set_theta_dir(bf_handle *h, int dir, double degrees);
int bf_set_thetas(bf_handle *h, const double *degrees, int n);
/* Root-mean-square of each output stream of a batch that is resident on the device (y as written by
 * bf_process_batch_device): rms_host[n_streams * n_dirs].  This is the quantity the reference's theta controllers
 * steer on (scripts/energy2theta.py:23-27 get_energy_from_list), so "publish theta, wait, measure" becomes
 * "evaluate n_dirs candidates in one batch and pick". */
int bf_stream_rms(bf_handle *h, const float *y_dev, size_t n_frames, double *rms_host, void *hip_stream);
/* interf_theta_roscallback (lcmv.cpp:258-309, gss.cpp:288-339): id is 1-based.  id <= current count updates that
 * interferer (and removes it when it lands within interf_angle_threshold of another one); id > count appends a
 * new interferer unless it is that close to an existing one.  As in the reference, a structural change rebuilds
 * the weight matrices zeroed and re-runs update_weights() without ini, so the reference-mic row is 0 afterwards
 * (quirk Q3).  Up to BF_MAX_INTERF interferers.  With more constraints than microphones (K + 1 > n_mics) the constraint
 * Gram matrix is singular and the reference's inverse() returns rounding noise: no parity is claimed there.
 * Takes effect at the next hop/batch. */
int bf_set_interference(bf_handle *h, unsigned id, double degrees);
/* Current number of interferers (interference_angles.size()). */
int bf_n_interferers(bf_handle *h);

/* Page-locked host memory for the host-buffer entry points: bf_process_batch copies from / to pageable memory at
 * about 30 GB/s (the runtime stages it), from / to these buffers at the PCIe rate (~55 GB/s) -- 34 vs 22 ms per
 * 65 536-frame 8-microphone batch.  No reference counterpart (JACK owns the reference's buffers). */
void *bf_host_alloc(size_t bytes);
void bf_host_free(void *p);

/* jack_callback body: do_overlap(in, out, nframes, apply_weights)
 * (das.cpp:72-92, util.h:289-314).  in = n_mics pointers to nframes float32
 * (host memory, as input_from_rosjack returns), out = nframes float32 (host).
 * nframes must equal cfg.hop.  Single-stream handles only; with n_dirs > 1 `out` receives [n_dirs][nframes]. */
int bf_process_hop(bf_handle *h, const float *const *in, float *out, uint32_t nframes);

/* n_frames consecutive callbacks per stream in one call, host buffers.
 * x: layout per cfg.layout with n_frames*hop samples per mic;
 * y: [n_streams][n_frames*hop].  State carries over to the next call exactly
 * as consecutive callbacks would.  Rounding: the das kernels that pack several frames of a batch into one transform (das in double on
 * planar input: two; fp32 das at periods below 512 frames: 1024 / (2 hop)) choose the frames by their position in the batch, so the same
 * stream cut differently agrees to the last bits of the float output (<= 1e-6 of its scale), not bit for bit; every cut is within the
 * stated tolerance of the reference arithmetic.  Frames that share a transform also share its rounding error, 1e-16 of the LOUDER
 * one: beside an ordinary frame a frame of exact zeros comes out as 1e-17s where the reference writes zeros, and a frame 2^-40 below
 * its neighbour loses its own last 40 bits (das in double only: the other nodes' kernels pair microphones in the forward transform
 * and, in the backward one, only frames of comparable scale).  gss: from 57 streams on a batch runs one lane per (stream, bin) problem
 * (sums over the microphones as one FMA chain), below that a group of lanes per problem (pairwise sums): the demixing recursion of a
 * stream rounds differently in a small and in a large batch, 1e-16 per step on a state with long memory -- the same stream alone and
 * inside a 64-stream batch agrees to ~1e-12 on the spectrum (float output: equal but for rare last-bit flips), not bit for bit. */
int bf_process_batch(bf_handle *h, const float *x_host, size_t n_frames, float *y_host);

/* Same, buffers already resident in HBM; enqueued on `hip_stream`
 * (a hipStream_t, NULL = default stream) without host synchronisation.
 * spectrum_dev (nullable): receives y_fft per frame as double2
 * [n_streams][n_frames][2*hop] (see DESIGN.md for the DAS fused variant). */
int bf_process_batch_device(bf_handle *h, const float *x_dev, size_t n_frames, float *y_dev,
                            void *spectrum_dev, void *hip_stream);

/* Steering / constraint matrices as the reference's update_weights builds them:
 * [2*hop][n_mics][n_interf+1] complex double (re,im). */
int bf_get_weights(bf_handle *h, double *w_host);

/* Checkpoint of all per-stream state (ring hop, OLA tail, covariance history,
 * GSS demixing matrices, MCRA/MPF vectors, smoothing tail).  The reference has
 * no counterpart (state lives in process globals). */
size_t bf_state_size(const bf_handle *h);
int bf_get_state(bf_handle *h, void *blob_host, size_t size);
int bf_set_state(bf_handle *h, const void *blob_host, size_t size);
/* Back to the reference's cold start (zeroed rings/tails/history; prepare_overlap_and_add, util.h:272-286).  Host-synchronous:
 * waits for the device before and after. */
int bf_reset(bf_handle *h);
/* The same, enqueued on `hip_stream` without host synchronisation: ordered against the batches the caller runs on that
 * stream (what a per-step cold start in a stream-driven loop needs: beamform_amd/shard.py run_shard, bf_shard_run). */
int bf_reset_async(bf_handle *h, void *hip_stream);

/* ---- frame-range sharding of one long stream across processes (one per GPU) -- SURVEY 8(e) ---------------------------------
 * The reference runs one node per process on one stream (das.cpp:101-145); frames are independent except for the overlap-add
 * neighbour (util.h:301-302) and mvdr / lcmv's covariance of the previous P frames (mvdr.cpp:87,100-101), so a rank that owns
 * the output hops [lo, hi) feeds a COLD node with `lead` hop (seeds the ring buffer, util.h:272-277) + `warm` recomputed frames
 * (output dropped) + its owned frames and needs no data-path collective; the one collective is the final gather of the
 * owned hops (RCCL over xGMI: examples/shard_node.cpp calls it directly, beamform_amd/shard.py through torch.distributed). */
/* bf_process_batch_device on a column range of a longer planar buffer: microphone m starts at x_dev + m * mic_stride
 * (samples), so one resident slice can be walked in pieces without copies (state carries from piece to piece as always). */
int bf_process_batch_device_strided(bf_handle *h, const float *x_dev, size_t n_frames, float *y_dev, void *hip_stream,
                                    long mic_stride);
typedef struct bf_shard {
    long long lo, hi; /* output hops this rank owns */
    int warm;         /* frames recomputed in front of lo (clipped at the stream start) */
    int lead;         /* hops fed in front of the first recomputed frame (0 at the stream start) */
} bf_shard;
/* Frames in front of its first frame a shard must recompute: 1 (das, phase), past_windows + 1 (mvdr, lcmv);
 * -1 for the nodes that recurse over frames or samples (gss, phasempf, mcra, gsc): those shard by stream only. */
int bf_shard_halo(const bf_config *cfg);
/* Contiguous, near-equal frame ranges; rank 0 starts from the true stream state (no warm-up, no lead). */
int bf_shard_plan(size_t n_frames, int world, int rank, int halo, bf_shard *out);
long long bf_shard_first_feed(const bf_shard *s); /* first hop of the global stream fed to the rank's cold node: lo - warm - lead */
long long bf_shard_n_feed(const bf_shard *s);     /* hops fed: lead + warm + owned */
long long bf_shard_n_drop(const bf_shard *s);     /* output hops in front of the owned range that are discarded: warm + lead */
/* One rank's step: cold start + the fed hops, enqueued on `hip_stream` without host synchronisation.  x_feed_dev holds hops
 * [first_feed, hi) of the global stream in the handle's layout, y_feed_dev n_feed * hop floats; the owned hops start at
 * element n_drop * hop of y_feed_dev.  One input stream, one look direction. */
int bf_shard_run(bf_handle *h, const float *x_feed_dev, const bf_shard *shard, float *y_feed_dev, void *hip_stream);

/* Timing hook for bench.py: runs bf_process_batch_device `iters` times between
 * two hipEvents recorded on `hip_stream` (mean ms per call -> *ms_per_call) and,
 * when ms_kernel != NULL, brackets every launch of the dominant kernel with its
 * own event pair on the same stream (mean ms per launch -> *ms_kernel). */
int bf_time_batch_device(bf_handle *h, const float *x_dev, size_t n_frames, float *y_dev, void *hip_stream,
                         int iters, float *ms_per_call, float *ms_kernel);
/* The same per-launch event pairs around the launches the CALLER issues between begin and end (bench.py takes its
 * roofline duration from the very launches its step time covers).  end: mean ms per launch, number of launches. */
int bf_kernel_timing_begin(bf_handle *h);
int bf_kernel_timing_end(bf_handle *h, float *ms_mean, int *n_launches);
/* Launch trace (diagnostics; no counterpart in the reference): the names of the kernels the CALLING THREAD launches through this library
 * between begin and end, in launch order, '\n'-separated, spelled as rocprofv3 spells them (demangled, no parameter list, e.g.
 * "bf::das_f64_pair_kernel", "bf::n1024::stft_kernel<0, true>").  end returns the length of the full list (snprintf convention: the
 * list is truncated to cap - 1 bytes) or a negative error when no trace is open.  tools/dispatch_table.py prints DESIGN.md's
 * dispatch table from it; bench.py prints a `traffic` figure only beside the kernels the counter file was taken on. */
int bf_trace_begin(void);
long bf_trace_end(char *buf, size_t cap);

/* ---- rosjack output stage, file half, and the batch front-end ----------------------------------------
 * rosjack.cpp:189-210: sf_open(audio_file_path, SFM_WRITE, {WAV | PCM_16, 1 channel, JACK or resampled rate});
 * rosjack.cpp:404-409: sf_write_float(audio_file, write_file_buffer, data_length) once per callback; closed with the node.
 * libsndfile (1.0.28 on the reference's Ubuntu 20.04) is neither vendored in the reference nor installed here; its header
 * layout and its float -> short rule on this call path (lrintf(x * 32767.0f), no clipping: beyond +-1.0 it wraps) are
 * restated in csrc/wavio.cpp.  Host functions: no HIP device needed. */
typedef struct bf_wav_writer bf_wav_writer;
int bf_wav_writer_open(const char *path, int sample_rate, bf_wav_writer **out);        /* sf_open(..., SFM_WRITE, ...) */
int bf_wav_writer_write(bf_wav_writer *w, const float *samples, size_t n);             /* sf_write_float */
int bf_wav_writer_write_pcm16(bf_wav_writer *w, const int16_t *pcm, size_t n);         /* samples already converted (device) */
int bf_wav_writer_close(bf_wav_writer *w);                                             /* sf_close: patches the lengths */
/* The sample rule of sf_write_float on a PCM_16 file, on the host and on a batch resident in HBM (16-byte aligned buffers). */
void bf_float_to_pcm16(const float *src, int16_t *dst, size_t n);
int bf_float_to_pcm16_device(const float *src_dev, int16_t *dst_dev, size_t n, void *hip_stream);
/* Front-end: a WAV file (PCM 16/24/32 or float32, any channel count; sf_read_float's scaling) or a raw planar float32 file
 * -> planar float32 [channel][sample], the BF_PLANAR layout of bf_process_batch.  Release with bf_wav_free. */
int bf_wav_read(const char *path, float **planar, int *n_channels, size_t *n_samples, int *sample_rate);
int bf_planar_f32_read(const char *path, int n_channels, float **planar, size_t *n_samples);
void bf_wav_free(float *planar);

/* ---- rosjack output stage, sample-rate half ----------------------------------------------------------------------------------
 * rosjack.cpp:159-184: samplerate_conv = src_new(SRC_SINC_FASTEST, 1, ..), src_ratio = ros_output_sample_rate / rosjack_sample_rate,
 * refused when src_is_valid_ratio fails; rosjack.cpp:311-338 (convert_to_sample_rate): one src_process per JACK period with
 * end_of_input = 0, the generated samples queued and emitted one period at a time (:340-349).  bf_resampler_* is that converter
 * as a stream operator on the GPU: every call takes the next piece of the mono stream and returns the output samples that
 * have become available (an output exists once bf_resampler_latency input samples beyond its position have been seen; what
 * libsamplerate holds back in the same way).  Results do not depend on how the stream is cut into calls.
 * libsamplerate (un-vendored; 0.1.9 on the reference's Ubuntu 20.04) is not in this image: csrc/resample.hip restates
 * src_sinc.c's mono converter; the built-in coefficient table is a Kaiser-windowed sinc of SINC_FASTEST's geometry, NOT
 * libsamplerate's fastest_coeffs.h (parity unpinned) -- bf_resampler_set_table installs any table of that form, e.g. the
 * original one.
 * Two ways of driving it (bf_resampler_set_mode):
 *   BF_RS_STREAM (default)  every input sample is consumed and every output sample returned (bf_resampler_process*): the
 *                           mathematically complete conversion of the stream, independent of how it is cut into calls;
 *   BF_RS_ROSJACK           the stage exactly as rosjack runs it, one bf_resampler_callback* per JACK period: src_process may
 *                           return at most one period of output (output_frames = rosjack_window_size, rosjack.cpp:176-183), a
 *                           period is copied into the converter only when input_frames == 0 and DROPPED otherwise (:311-338),
 *                           libsamplerate pulls input in lazily (src_sinc.c prepare_data), and a block is published only when a
 *                           full period of output is queued, at most one per callback, the tail never (:340-349, :416-436).
 *                           With out_rate > in_rate the reference therefore loses whole periods (16 -> 48 kHz keeps about every
 *                           third one); with out_rate <= in_rate the two modes differ only in the blocking of the output.
 *                           examples/file_node: trailing argument "rosjack". */
typedef struct bf_resampler bf_resampler;
int bf_resampler_create(int in_rate, int out_rate, bf_resampler **out);               /* src_new + src_ratio; BF_EINVAL outside 1/256..256 */
int bf_resampler_set_table(bf_resampler *r, const float *coeffs, int n_coeffs, int index_inc); /* half table incl. 2 guard entries; resets */
enum bf_resampler_mode { BF_RS_STREAM = 0, BF_RS_ROSJACK = 1 };
int bf_resampler_set_mode(bf_resampler *r, int mode, int period);                     /* period = rosjack_window_size (BF_RS_ROSJACK); resets */
/* BF_RS_ROSJACK: one output_to_rosjack(data_out, period).  *emitted = 1 when a block of `period` samples was published into
 * `block`, *accepted = 0 when the reference would have dropped this period.  Device or host buffers of `period` floats. */
int bf_resampler_callback_device(bf_resampler *r, const float *period_dev, float *block_dev, int *emitted, int *accepted, void *hip_stream);
int bf_resampler_callback(bf_resampler *r, const float *period, float *block, int *emitted, int *accepted);
int bf_resampler_reset(bf_resampler *r);                                              /* src_reset */
size_t bf_resampler_out_count(bf_resampler *r, size_t n_in);                          /* outputs the next call with n_in samples yields */
int bf_resampler_latency(bf_resampler *r);                                            /* look-ahead in input samples */
int bf_resampler_process_device(bf_resampler *r, const float *in_dev, size_t n_in, float *out_dev, size_t out_cap, size_t *n_out,
                                void *hip_stream);                                    /* src_process on buffers resident in HBM */
int bf_resampler_process(bf_resampler *r, const float *in, size_t n_in, float *out, size_t out_cap, size_t *n_out); /* host buffers */
void bf_resampler_destroy(bf_resampler *r);                                           /* src_delete */
int bf_resampler_default_table(float *dst, int cap, int *index_inc);                  /* copy of the built-in table; returns its length */

#ifdef __cplusplus
}
#endif
#endif /* BFCORE_H */
