/*
 * bf_oracle.h -- C interface of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * The oracle is a double-precision CPU restatement of the hot path of
 * balkce/beamform (STFT -> per-bin weighting -> ISTFT + overlap-add) used
 * only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * The product library (beamform_amd/lib/libbfcore.so) never links, loads or
 * calls anything in this directory.
 *
 * PARITY UNPINNED: the reference has no tests, fixtures or golden vectors and
 * cannot be built in this image (ROS, JACK, FFTW3 and Eigen3 headers are
 * absent), so this restatement is checked only against an independent numpy
 * restatement (oracle/np_oracle.py) and against known-answer properties that
 * follow from the reference source.  See DESIGN.md "Oracle".
 */
#ifndef BF_ORACLE_H
#define BF_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_MICS 32
#define ORC_MAX_INTERF 16

enum { ORC_DAS = 0, ORC_MVDR = 1, ORC_LCMV = 2, ORC_GSS = 3, ORC_PHASE = 4, ORC_PHASEMPF = 5,
       ORC_MCRA = 6 /* single-channel mcra node, mcra.cpp */,
       ORC_GSC = 7  /* generalized sidelobe canceller, gsc.cpp */ };

typedef struct orc_params {
    int algo;
    int n_mics;
    int hop;                          /* rosjack_window_size; fft_win = 2*hop (util.h:261) */
    double sample_rate;               /* rosjack_sample_rate */
    double mic_x[ORC_MAX_MICS];       /* RAW beamform_config.yaml coordinates (util.h:82-92) */
    double mic_y[ORC_MAX_MICS];
    double theta;                     /* initial_angle, degrees */
    int n_interf;                     /* interference_angles.size() */
    double interf_angle[ORC_MAX_INTERF];
    /* mvdr / lcmv / gss (mvdr.cpp:27-31, lcmv.cpp:29-34, gss.cpp:35-41) */
    int past_windows;
    double freq_mag_threshold, freq_max, freq_min, out_amp;
    double mu, lambda_;
    /* phase (phase.cpp:24-27) */
    double min_phase, mag_mult, mag_threshold;
    /* phasempf (phasempf.cpp:30-59); the mcra node (mcra.cpp:40-48) uses the mcra_* fields, out_amp and out_only_noise */
    double min_mag;
    int smooth_size;
    double mcra_alphaS, mcra_alphaD, mcra_alphaD2, mcra_delta;
    int mcra_L;
    double mpf_alphaS, mpf_eta, mpf_rev_gamma, mpf_rev_delta;
    double noise_floor;
    int out_only_noise, out_only_mcra;
    /* gsc (gsc.cpp:17-21, launch/gsc.launch) */
    int gsc_use_vad;
    double gsc_vad_threshold, gsc_mu0, gsc_mu_max;
    int gsc_filter_size;
} orc_params;

typedef struct orc_node orc_node;

/* main(): params -> prepare_overlap_and_add -> buffers -> update_weights(true) */
orc_node *orc_create(const orc_params *p);
void orc_destroy(orc_node *n);

/* theta_roscallback: angle = msg; update_weights() (das.cpp:94-99) */
void orc_set_theta(orc_node *n, double deg);

/* interf_theta_roscallback (lcmv.cpp:258-309, gss.cpp:288-339): id is 1-based; updates an existing interferer,
 * removes it when it lands within interf_angle_threshold of another one, or appends a new one.  Structural
 * changes re-allocate the (zeroed) weight matrices and call update_weights() WITHOUT ini, so the reference-mic
 * row stays 0 afterwards (quirk Q3).  Returns the new number of interferers. */
int orc_set_interference(orc_node *n, unsigned id, double angle, double interf_angle_threshold);

/* One jack_callback worth of work: do_overlap(in, out, nframes, apply_weights)
 * (+ phasempf output smoothing).  in = [n_mics][hop] planar float32, out = [hop].
 * If Y != NULL it receives y_fft of this frame (fft_win complex doubles, re/im
 * interleaved) as it stands just before fftw_execute(y_inverse). */
int orc_process_hop(orc_node *n, const float *in, float *out, double *Y);

/* n_frames consecutive callbacks.  x = [n_mics][n_frames*hop] planar,
 * y = [n_frames*hop], Y = [n_frames][fft_win][2] or NULL. */
int orc_process(orc_node *n, const float *x, long n_frames, float *y, double *Y);

/* Introspection for known-answer tests. */
void orc_get_freqs(const orc_node *n, double *f /* fft_win */);
void orc_get_delays(const orc_node *n, double *d /* n_mics */);
void orc_get_hann(const orc_node *n, double *h /* fft_win */);
/* steering matrix: [fft_win][n_mics][n_interf+1] complex (re,im) */
void orc_get_weights(const orc_node *n, double *w);
/* forward FFT used by the oracle (unnormalised, exp(-i...)), for self-test */
void orc_fft(const double *in, double *out, int n, int sign);

#ifdef __cplusplus
}
#endif
#endif
