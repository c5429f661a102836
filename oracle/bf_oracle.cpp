/*
 * bf_oracle.cpp -- CPU oracle (TEST INFRASTRUCTURE ONLY; see bf_oracle.h).
 *
 * A double-precision restatement of the per-callback hot path of
 * balkce/beamform, written from the reference's behaviour, function by
 * function.  Each routine cites the reference file:line it follows
 * (paths relative to the reference root, beamform/src/...).
 *
 * Third-party arithmetic the reference delegates to libraries that are NOT
 * vendored in it and NOT installed in this image:
 *   - FFTW3 (un-pinned; ROS Noetic => 3.3.8 by inference): fftw_plan_dft_1d
 *     forward = sum x[n] exp(-2 pi i n k / N), backward = unnormalised
 *     exp(+...).  Restated here as an iterative radix-2 transform with a
 *     double-precision twiddle table (dft_pow2).
 *   - Eigen3 (un-pinned; 3.3.7 by inference): MatrixXcd products, adjoint,
 *     cwiseProduct and inverse(); inverse() of a Dynamic-size matrix is
 *     PartialPivLU followed by a solve against the identity.  Restated here
 *     as lu_inverse().
 *
 * PARITY UNPINNED (no reference tests / golden vectors exist and the reference
 * cannot be built here without writing stand-ins for ROS/JACK/FFTW/Eigen).
 *
 * Defined behaviour where the reference reads uninitialised memory
 * (documented in DESIGN.md "Reference quirks"):
 *   Q1   freqs[fft_win/2] is never written (util.h:190-199)        -> 0.0
 *   Q15d phasempf y_fft[0] is never written (phasempf.cpp:274)     -> 0.0
 *        phasempf out_soi_square[0]/out_int_square[0] never written -> 0.0
 *   Q16  mcra node: "y_fft[j] = in_fft(0,0)" runs with j == fft_win (mcra.cpp:127, one element past
 *        the buffer); y_fft[0] is never written                      -> 0.0
 */
#include "bf_oracle.h"

#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef std::complex<double> cd;

/* util.h:23-27 */
#define ORC_PI 3.141592653589793238462643383279502884
static const cd ORC_I(0, 1);
static const double v_sound = 343;
static const double rad2deg = 180.0 / ORC_PI;
static const double deg2rad = ORC_PI / 180.0;

/* ---------------------------------------------------------------- FFT -- */
/* Stand-in for fftw_plan_dft_1d + fftw_execute (das.cpp:127-128): like an FFTW plan, the bit-reversal permutation and
 * the twiddle table of a (size, direction) pair are built ONCE (first use, then cached for the life of the process --
 * the reference plans once in main()) and every execute only runs the butterflies.
 * sign = -1: FFTW_FORWARD, +1: FFTW_BACKWARD (unnormalised). */
struct DftPlan {
    int n, sign;
    std::vector<int> rev;
    std::vector<cd> tw;
    DftPlan(int n_, int sign_) : n(n_), sign(sign_), rev(n_), tw(n_ / 2 > 0 ? n_ / 2 : 1) {
        int lg = 0;
        while ((1 << lg) < n) lg++;
        for (int i = 0; i < n; i++) {
            int r = 0;
            for (int b = 0; b < lg; b++)
                if (i & (1 << b)) r |= 1 << (lg - 1 - b);
            rev[i] = r;
        }
        for (int k = 0; k < n / 2; k++) {
            double a = sign * 2.0 * M_PI * (double)k / (double)n;
            tw[k] = cd(std::cos(a), std::sin(a));
        }
    }
};

static const DftPlan &dft_plan(int n, int sign) {
    /* one plan per (size, direction); thread_local so concurrent oracle nodes (bench.py's all-cores baseline uses
     * processes, tests may use threads) never share a half-built table */
    static thread_local std::vector<DftPlan *> plans;
    for (DftPlan *p : plans)
        if (p->n == n && p->sign == sign) return *p;
    plans.push_back(new DftPlan(n, sign));
    return *plans.back();
}

static void dft_pow2(const cd *in, cd *out, int n, int sign) {
    const DftPlan &pl = dft_plan(n, sign);
    const int *rev = pl.rev.data();
    const cd *tw = pl.tw.data();
    for (int i = 0; i < n; i++) out[rev[i]] = in[i];
    for (int len = 2; len <= n; len <<= 1) {
        int half = len >> 1, step = n / len;
        for (int i = 0; i < n; i += len)
            for (int k = 0; k < half; k++) {
                cd w = tw[k * step];
                cd b = out[i + k + half];
                cd t(w.real() * b.real() - w.imag() * b.imag(), w.real() * b.imag() + w.imag() * b.real());
                cd a = out[i + k];
                out[i + k] = a + t;
                out[i + k + half] = a - t;
            }
    }
}

/* ------------------------------------------------- small dense complex -- */
/* Column-major dense matrix, the subset of Eigen::MatrixXcd the path uses. */
struct Mat {
    int r, c;
    std::vector<cd> a;
    Mat() : r(0), c(0) {}
    Mat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, cd(0, 0)) {}
    cd &operator()(int i, int j) { return a[(size_t)j * r + i]; }
    const cd &operator()(int i, int j) const { return a[(size_t)j * r + i]; }
};

static inline cd cmul(const cd &x, const cd &y) {
    return cd(x.real() * y.real() - x.imag() * y.imag(), x.real() * y.imag() + x.imag() * y.real());
}

static Mat matmul(const Mat &A, const Mat &B) {
    Mat C(A.r, B.c);
    for (int j = 0; j < B.c; j++)
        for (int k = 0; k < A.c; k++) {
            cd b = B(k, j);
            for (int i = 0; i < A.r; i++) C(i, j) += cmul(A(i, k), b);
        }
    return C;
}

static Mat adjoint(const Mat &A) {
    Mat H(A.c, A.r);
    for (int i = 0; i < A.r; i++)
        for (int j = 0; j < A.c; j++) H(j, i) = std::conj(A(i, j));
    return H;
}

/* Eigen MatrixXcd::inverse() for Dynamic sizes = PartialPivLU + solve(I)
 * (mvdr.cpp:88, lcmv.cpp:113,116).  Pivot = largest |.| in the column, as
 * Eigen's partial_lu_impl does (it compares abs, scalar_score_coeff_op). A zero
 * pivot is NOT trapped: like Eigen, the division produces inf/nan
 * (reference frame-0 behaviour). */
static Mat lu_inverse(const Mat &Ain) {
    int n = Ain.r;
    Mat A = Ain;
    std::vector<int> perm(n);
    for (int i = 0; i < n; i++) perm[i] = i;
    for (int k = 0; k < n; k++) {
        int piv = k;
        double best = std::abs(A(k, k));
        for (int i = k + 1; i < n; i++) {
            double v = std::abs(A(i, k));
            if (v > best) { best = v; piv = i; }
        }
        if (piv != k) {
            for (int j = 0; j < n; j++) std::swap(A(k, j), A(piv, j));
            std::swap(perm[k], perm[piv]);
        }
        cd d = A(k, k);
        for (int i = k + 1; i < n; i++) A(i, k) = A(i, k) / d;
        for (int j = k + 1; j < n; j++) {
            cd u = A(k, j);
            for (int i = k + 1; i < n; i++) A(i, j) -= cmul(A(i, k), u);
        }
    }
    Mat X(n, n);
    for (int col = 0; col < n; col++) {
        std::vector<cd> b(n);
        for (int i = 0; i < n; i++) b[i] = (perm[i] == col) ? cd(1, 0) : cd(0, 0);
        for (int i = 0; i < n; i++)
            for (int k = 0; k < i; k++) b[i] -= cmul(A(i, k), b[k]);
        for (int i = n - 1; i >= 0; i--) {
            for (int k = i + 1; k < n; k++) b[i] -= cmul(A(i, k), b[k]);
            b[i] = b[i] / A(i, i);
        }
        for (int i = 0; i < n; i++) X(i, col) = b[i];
    }
    return X;
}

/* ------------------------------------------------------------- the node -- */
struct orc_node {
    orc_params p;
    int M, H, N, S; /* mics, hop, fft_win, n_interf+1 */
    double angle;

    /* util.h:35-36,82-92: dist/angle from the RAW coordinates (Q2) */
    std::vector<double> mic_dist, mic_ang;
    std::vector<double> interference_angles;

    /* util.h:40-50 */
    std::vector<double> hann_win;
    std::vector<std::vector<float> > in_ring; /* M rings, 2H float32 oldest-first */
    std::vector<float> out_buff[2];

    std::vector<double> freqs, delays;
    std::vector<std::vector<double> > interf_delays;
    /* weights[j] is M x S (lcmv.cpp:25 layout; das/mvdr/phase use column 0) */
    std::vector<Mat> weights, weights_h;
    std::vector<cd> x_time, x_fft, y_fft, y_time;
    Mat in_fft; /* M x N */

    /* mvdr / lcmv */
    std::vector<Mat> past_ffts; /* N of M x P */
    Mat whiteR;
    /* gss */
    std::vector<Mat> sep_matrix;
    /* phase */
    std::vector<double> phases_aligned;
    /* phasempf (phasempf.cpp:62-76) */
    std::vector<double> past_samples;
    std::vector<cd> out_soi, out_int;
    std::vector<double> soi2, int2, S_, S_prev, S_f, S_tmp, S_min, Z, lambda_noise, lambda_leak, lambda_rev0, lambda_rev1,
        lambda_mpf;
    int current_L;
    bool first_L;
    /* gsc (gsc.cpp:24-27, util.h:316-347) */
    std::vector<std::vector<float> > out_buff_mic[2]; /* [2][M][N] */
    std::vector<std::vector<float> > block_matrix, filter;
    std::vector<float> last_outputs;
};

/* util.h:136-161 calculate_delays and util.h:163-188 calculate_interf_delays */
static void calc_delays(const orc_node *n, double ang, double *delay_buffer) {
    for (int i = 0; i < n->M; ++i) {
        if (i == 0) {
            delay_buffer[i] = 0.0;
        } else {
            double this_dist = n->mic_dist[i];
            double this_angle = n->mic_ang[i] - ang;
            if (this_angle > 180) {
                this_angle -= 360;
            } else if (this_angle < -180) {
                this_angle += 360;
            }
            delay_buffer[i] = this_dist * cos(this_angle * deg2rad) / (-v_sound);
        }
    }
}

/* util.h:190-199 (Q1: index N/2 left at the defined value 0.0) */
static void calc_freqs(double *freq_buffer, unsigned int size, double sample_rate) {
    for (unsigned int i = 0; i < size; i++) freq_buffer[i] = 0.0;
    freq_buffer[0] = 0.0;
    for (int i = 0; i < (int)(size / 2) - 1; ++i) {
        freq_buffer[i + 1] = ((double)(i + 1) / (double)size) * sample_rate;
        freq_buffer[size - 1 - i] = -((double)(i + 1) / (double)size) * sample_rate;
    }
    freq_buffer[(size / 2) - 1] = sample_rate / 2;
}

/* das.cpp:27-45, mvdr.cpp:41-60, lcmv.cpp:44-86, gss.cpp:51-94,
 * phase.cpp:33-51, phasempf.cpp:85-103 */
static void update_weights(orc_node *n, bool ini) {
    calc_delays(n, n->angle, n->delays.data());
    for (int i = 0; i < n->M; i++) {
        if (i == 0) {
            if (ini)
                for (int j = 0; j < n->N; j++) n->weights[j](i, 0) = 1.0;
        } else {
            for (int j = 0; j < n->N; j++)
                n->weights[j](i, 0) = std::exp(-ORC_I * (double)2 * ORC_PI * n->freqs[j] * n->delays[i]);
        }
    }
    for (int k = 0; k < n->S - 1; k++) {
        calc_delays(n, n->interference_angles[k], n->interf_delays[k].data());
        for (int i = 0; i < n->M; i++) {
            if (i == 0) {
                if (ini)
                    for (int j = 0; j < n->N; j++) n->weights[j](i, k + 1) = 1.0;
            } else {
                for (int j = 0; j < n->N; j++)
                    n->weights[j](i, k + 1) = std::exp(-ORC_I * (double)2 * ORC_PI * n->freqs[j] * n->interf_delays[k][i]);
            }
        }
    }
    for (int j = 0; j < n->N; j++) {
        n->weights_h[j] = adjoint(n->weights[j]);
        if (n->p.algo == ORC_GSS) n->sep_matrix[j] = adjoint(n->weights[j]); /* gss.cpp:90-93 */
    }
}

/* util.h:217-242 (ring read vector flattened: ring holds N float32 oldest first) */
static void prepare_input(const orc_node *n, int mic, cd *x) {
    const float *buf = n->in_ring[mic].data();
    for (int i = 0; i < n->N; i++) x[i] = buf[i] * n->hann_win[i];
}

/* util.h:244-253 */
static void prepare_output(const orc_node *n, const cd *y, float *out) {
    for (int j = 0; j < n->N; j++) {
        out[j] = real(y[j]) / (double)(unsigned int)n->N;
        out[j] *= n->hann_win[j];
    }
}

/* common head of every apply_weights (das.cpp:51-57) */
static void forward_all(orc_node *n) {
    for (int i = 0; i < n->M; i++) {
        prepare_input(n, i, n->x_time.data());
        dft_pow2(n->x_time.data(), n->x_fft.data(), n->N, -1);
        for (int j = 0; j < n->N; j++) n->in_fft(i, j) = n->x_fft[j];
    }
}

/* das.cpp:47-70 */
static void apply_das(orc_node *n, float *out) {
    forward_all(n);
    for (int j = 0; j < n->N; j++) {
        cd acc(0, 0);
        for (int i = 0; i < n->M; i++) acc += cmul(std::conj(n->weights[j](i, 0)), n->in_fft(i, j));
        n->y_fft[j] = acc;
        n->y_fft[j] /= n->M;
    }
}

/* mvdr.cpp:62-115 and lcmv.cpp:88-140 share everything but the bin-0 rule and
 * the constraint matrix. */
static void apply_mvdr_lcmv(orc_node *n, bool lcmv) {
    forward_all(n);
    const int M = n->M, P = n->p.past_windows;
    int j0 = 0;
    if (!lcmv) { /* mvdr.cpp:76-77 */
        n->y_fft[0] = n->in_fft(0, 0);
        j0 = 1;
    }
    for (int j = j0; j < n->N; j++) {
        double this_freq = std::abs(n->freqs[j]);
        double this_mag = 0.0;
        for (int i = 0; i < M; i++) this_mag += std::abs(n->in_fft(i, j));
        this_mag /= (unsigned int)M * (unsigned int)n->N;

        if (this_freq >= n->p.freq_min && this_freq <= n->p.freq_max) {
            if (this_mag > n->p.freq_mag_threshold) {
                Mat &Pst = n->past_ffts[j];
                Mat R = matmul(Pst, adjoint(Pst));
                for (int a = 0; a < M; a++)
                    for (int b = 0; b < M; b++) R(a, b) = cmul(R(a, b), n->whiteR(a, b));
                Mat invR = lu_inverse(R);
                if (!lcmv) {
                    /* mvdr.cpp:91-94 */
                    Mat a(M, 1);
                    for (int i = 0; i < M; i++) a(i, 0) = n->weights[j](i, 0);
                    Mat num = matmul(invR, a);
                    Mat ah(1, M);
                    for (int i = 0; i < M; i++) ah(0, i) = n->weights_h[j](0, i);
                    cd den = matmul(matmul(ah, invR), a)(0, 0);
                    cd acc(0, 0);
                    for (int i = 0; i < M; i++) acc += cmul(std::conj(num(i, 0) / den), n->in_fft(i, j));
                    n->y_fft[j] = acc;
                } else {
                    /* lcmv.cpp:116-119 */
                    const Mat &C = n->weights[j];
                    const Mat &Ch = n->weights_h[j];
                    Mat W = matmul(matmul(invR, C), lu_inverse(matmul(matmul(Ch, invR), C)));
                    cd acc(0, 0);
                    for (int i = 0; i < M; i++) acc += cmul(std::conj(W(i, 0)), n->in_fft(i, j));
                    n->y_fft[j] = acc;
                }
            } else {
                n->y_fft[j] = n->in_fft(0, j) * 0.01;
            }
            /* mvdr.cpp:100-101: shift history, append this window */
            Mat &Pst = n->past_ffts[j];
            for (int c = 0; c < P - 1; c++)
                for (int i = 0; i < M; i++) Pst(i, c) = Pst(i, c + 1);
            for (int i = 0; i < M; i++) Pst(i, P - 1) = n->in_fft(i, j);
        } else {
            n->y_fft[j] = 0.0;
        }
    }
}

/* gss.cpp:96-156 */
static void apply_gss(orc_node *n) {
    forward_all(n);
    const int M = n->M, S = n->S;
    for (int j = 0; j < n->N; j++) {
        double this_freq = std::abs(n->freqs[j]);
        double this_mag = 0.0;
        for (int i = 0; i < M; i++) this_mag += std::abs(n->in_fft(i, j));
        this_mag /= (unsigned int)M * (unsigned int)n->N;

        if (this_freq >= n->p.freq_min && this_freq <= n->p.freq_max) {
            if (this_mag > n->p.freq_mag_threshold) {
                Mat x(M, 1);
                for (int i = 0; i < M; i++) x(i, 0) = n->in_fft(i, j);
                Mat yf = matmul(n->sep_matrix[j], x); /* gss.cpp:120 */
                n->y_fft[j] = yf(0, 0);
                Mat E = matmul(yf, adjoint(yf)); /* gss.cpp:124-125 */
                for (int d = 0; d < S; d++) E(d, d) -= E(d, d);
                double alpha = 0.0; /* gss.cpp:128-129 */
                for (int i = 0; i < M; i++) alpha += std::norm(x(i, 0));
                alpha *= alpha;
                /* gss.cpp:132 */
                double c1 = (double)(4 * (size_t)S) * (1 / alpha);
                Mat dj1 = matmul(matmul(E, yf), adjoint(x));
                for (size_t q = 0; q < dj1.a.size(); q++) dj1.a[q] = dj1.a[q] * c1;
                /* gss.cpp:133: 2 * (1/(S)) is INTEGER arithmetic (Q13) */
                size_t c2i = 2 * (1 / (size_t)S);
                double c2 = (double)c2i;
                Mat WA = matmul(n->sep_matrix[j], n->weights[j]);
                for (int d = 0; d < S; d++) WA(d, d) -= cd(1, 0);
                Mat dj2 = matmul(WA, n->weights_h[j]);
                for (size_t q = 0; q < dj2.a.size(); q++) dj2.a[q] = dj2.a[q] * c2;
                /* gss.cpp:136 */
                double keep = 1 - n->p.lambda_ * n->p.mu;
                Mat &W = n->sep_matrix[j];
                for (size_t q = 0; q < W.a.size(); q++) W.a[q] = (keep * W.a[q]) - n->p.mu * (dj1.a[q] + dj2.a[q]);
            } else {
                n->y_fft[j] = n->in_fft(0, j) * 0.01;
            }
        } else {
            n->y_fft[j] = 0.0;
        }
    }
}

/* phase.cpp:53-68 / phasempf.cpp:105-120 (recursive pair sum, same association).
 * Toolchain-dependent reading: phase.cpp:58 calls an UNQUALIFIED abs() on a double.  It resolves to the floating overload only because
 * libstdc++'s <cmath> / <stdlib.h> wrappers put std::abs(double) into the global namespace (true for the GCC 9 that ROS Noetic implies and
 * for the GCC 11 of this image: SURVEY.md App. C probe 2 prints 2.7 for -2.7); with a C library's int abs(int) alone the difference would be
 * truncated to an integer.  The oracle assumes the floating overload (std::abs below). */
static double overall_phase_diff(const orc_node *n, int min_i, int *num_i) {
    if (min_i < n->M - 1) {
        double this_diff = 0;
        for (int i = min_i + 1; i < n->M; i++) {
            double this_diff_raw = std::abs(n->phases_aligned[min_i] - n->phases_aligned[i]);
            if (this_diff_raw > M_PI) this_diff_raw = 2 * M_PI - this_diff_raw;
            this_diff += this_diff_raw;
            (*num_i)++;
        }
        return this_diff + overall_phase_diff(n, min_i + 1, num_i);
    }
    return 0;
}

/* phase.cpp:70-134 */
static void apply_phase(orc_node *n) {
    forward_all(n);
    const double min_phase_diff_mean = n->p.min_phase * M_PI / 180; /* phase.cpp:175 */
    n->y_fft[0] = n->in_fft(0, 0);
    for (int j = 1; j < n->N; j++) {
        double mag_mean = 0;
        for (int i = 0; i < n->M; i++) mag_mean += std::abs(n->in_fft(i, j));
        mag_mean /= n->M;
        double pha_mean = std::arg(n->in_fft(0, j));
        if (mag_mean / (unsigned int)n->N > n->p.mag_threshold) {
            for (int i = 0; i < n->M; i++)
                n->phases_aligned[i] = std::arg(cmul(std::conj(n->weights[j](i, 0)), n->in_fft(i, j)));
            int num = 0;
            double sum = overall_phase_diff(n, 0, &num);
            double mean = sum / (double)num;
            pha_mean = std::arg(n->in_fft(0, j));
            if (mean < min_phase_diff_mean) {
                n->y_fft[j] = cd(mag_mean * cos(pha_mean), mag_mean * sin(pha_mean));
            } else {
                mag_mean *= n->p.mag_mult;
                n->y_fft[j] = cd(mag_mean * cos(pha_mean), mag_mean * sin(pha_mean));
            }
        } else {
            mag_mean *= n->p.mag_mult;
            n->y_fft[j] = cd(mag_mean * cos(pha_mean), mag_mean * sin(pha_mean));
        }
    }
}

static inline double dmin(double a, double b) { return a > b ? b : a; } /* phasempf.cpp:132-138 */

/* phasempf.cpp:140-191 */
static void mcra(orc_node *n) {
    static const double win[3] = {0.25, 0.5, 0.25}; /* phasempf.cpp:43-45 */
    static const int pos[3] = {-1, 0, 1};
    const int N = n->N;
    n->S_f[0] = std::abs(n->out_soi[0]);
    for (int j = 1; j < N; j++) {
        n->S_f[j] = 0.0;
        for (int i = 0; i < 3; i++) {
            int this_j = j + pos[i];
            if (this_j >= 1 && this_j < N) n->S_f[j] += win[i] * n->soi2[j]; /* Q15e: [j], not [this_j] */
        }
    }
    for (int j = 0; j < N; j++) n->S_[j] = (n->p.mcra_alphaS * n->S_prev[j]) + ((1 - n->p.mcra_alphaS) * n->S_f[j]);
    if (n->current_L > n->p.mcra_L) {
        for (int j = 0; j < N; j++) {
            n->S_min[j] = dmin(n->S_tmp[j], n->S_[j]);
            n->S_tmp[j] = n->S_[j];
        }
        n->current_L = 1;
        n->first_L = false;
    } else {
        for (int j = 0; j < N; j++) {
            n->S_min[j] = dmin(n->S_min[j], n->S_[j]);
            n->S_tmp[j] = dmin(n->S_tmp[j], n->S_[j]);
        }
        n->current_L++;
    }
    for (int j = 0; j < N; j++) {
        if (n->first_L || n->S_[j] < n->S_min[j] * n->p.mcra_delta || n->lambda_noise[j] > n->soi2[j]) {
            if (n->first_L && ((1.0f / (double)n->current_L) > n->p.mcra_alphaD)) {
                n->lambda_noise[j] = (1.0f / (double)n->current_L) * n->lambda_noise[j] +
                                     (1.0f - (1.0f / (double)n->current_L)) * n->soi2[j];
            } else {
                n->lambda_noise[j] = n->p.mcra_alphaD2 * n->lambda_noise[j] + (1.0f - n->p.mcra_alphaD) * n->soi2[j];
            }
        }
    }
    for (int j = 0; j < N; j++) n->S_prev[j] = n->S_[j];
}


/* mcra.cpp:64-155 -- the single-channel MCRA spectral-subtraction node (only channel 0 is used, mcra.cpp:72-73) */
static void apply_mcra_node(orc_node *n) {
    static const double win[3] = {0.25, 0.5, 0.25}; /* mcra.cpp:29-31 */
    static const int pos[3] = {-1, 0, 1};
    const int N = n->N;
    prepare_input(n, 0, n->x_time.data());
    dft_pow2(n->x_time.data(), n->x_fft.data(), N, -1);
    for (int j = 0; j < N; j++) { /* mcra.cpp:75-78 */
        n->in_fft(0, j) = n->x_fft[j];
        n->soi2[j] = std::norm(n->x_fft[j]);
    }
    n->S_f[0] = std::abs(n->in_fft(0, 0)); /* magnitude, not power: mcra.cpp:83 */
    for (int j = 1; j < N; j++) {          /* mcra.cpp:84-92: 3-tap smoothing over the neighbouring bins, bin 0 excluded */
        n->S_f[j] = 0.0;
        for (int i = 0; i < 3; i++) {
            int this_j = j + pos[i];
            if (this_j >= 1 && this_j < N) n->S_f[j] += win[i] * n->soi2[this_j];
        }
    }
    for (int j = 0; j < N; j++) n->S_[j] = (n->p.mcra_alphaS * n->S_prev[j]) + ((1 - n->p.mcra_alphaS) * n->S_f[j]);
    if (n->current_L > n->p.mcra_L) { /* mcra.cpp:100-113 */
        for (int j = 0; j < N; j++) {
            n->S_min[j] = dmin(n->S_tmp[j], n->S_[j]);
            n->S_tmp[j] = n->S_[j];
        }
        n->current_L = 1;
        n->first_L = false;
    } else {
        for (int j = 0; j < N; j++) {
            n->S_min[j] = dmin(n->S_min[j], n->S_[j]);
            n->S_tmp[j] = dmin(n->S_tmp[j], n->S_[j]);
        }
        n->current_L++;
    }
    for (int j = 0; j < N; j++) { /* mcra.cpp:116-124 */
        if (n->first_L || n->S_[j] < n->S_min[j] * n->p.mcra_delta || n->lambda_noise[j] > n->soi2[j]) {
            if (n->first_L && ((1.0f / (double)n->current_L) > n->p.mcra_alphaD)) {
                n->lambda_noise[j] = (1.0f / (double)n->current_L) * n->lambda_noise[j] +
                                     (1.0f - (1.0f / (double)n->current_L)) * n->soi2[j];
            } else {
                n->lambda_noise[j] = n->p.mcra_alphaD2 * n->lambda_noise[j] + (1.0f - n->p.mcra_alphaD) * n->soi2[j];
            }
        }
    }
    /* mcra.cpp:127: the store of in_fft(0,0) lands one past the end of y_fft; y_fft[0] keeps its value 0 (Q16) */
    for (int j = 1; j < N; j++) { /* mcra.cpp:128-143 */
        double this_pha = std::arg(n->in_fft(0, j));
        double this_mag;
        if (n->p.out_only_noise) {
            this_mag = (sqrt(n->lambda_noise[j])) * n->p.out_amp;
        } else {
            this_mag = (std::abs(n->in_fft(0, j)) - sqrt(n->lambda_noise[j])) * n->p.out_amp;
            if (this_mag < 0) this_mag = 0.0;
        }
        n->y_fft[j] = cd(this_mag * cos(this_pha), this_mag * sin(this_pha));
    }
    for (int j = 0; j < N; j++) n->S_prev[j] = n->S_[j]; /* mcra.cpp:146-148 */
}

/* phasempf.cpp:193-302 */
static void apply_phasempf(orc_node *n) {
    forward_all(n);
    const int N = n->N;
    const double min_phase_diff_mean = n->p.min_phase * M_PI / 180; /* phasempf.cpp:365 */
    n->out_soi[0] = n->in_fft(0, 0);
    n->out_int[0] = n->in_fft(0, 0);
    for (int j = 1; j < N; j++) {
        for (int i = 0; i < n->M; i++)
            n->phases_aligned[i] = std::arg(cmul(std::conj(n->weights[j](i, 0)), n->in_fft(i, j)));
        int num = 0;
        double sum = overall_phase_diff(n, 0, &num);
        double mean = sum / (double)num;
        double mag_mean = 0;
        for (int i = 0; i < n->M; i++) mag_mean += std::abs(n->in_fft(i, j));
        mag_mean /= n->M;
        double pha_mean = std::arg(n->in_fft(0, j));
        if (mean < min_phase_diff_mean) {
            n->out_soi[j] = cd(mag_mean * cos(pha_mean), mag_mean * sin(pha_mean));
            mag_mean *= n->p.min_mag;
            n->out_int[j] = cd(mag_mean * cos(pha_mean), mag_mean * sin(pha_mean));
        } else {
            n->out_int[j] = cd(mag_mean * cos(pha_mean), mag_mean * sin(pha_mean));
            mag_mean *= n->p.min_mag;
            n->out_soi[j] = cd(mag_mean * cos(pha_mean), mag_mean * sin(pha_mean));
        }
        n->soi2[j] = std::norm(n->out_soi[j]);
        n->int2[j] = std::norm(n->out_int[j]);
    }
    mcra(n);
    for (int j = 0; j < N; j++) { /* phasempf.cpp:255-271 */
        n->Z[j] = n->p.mpf_alphaS * n->Z[j] + (1 - n->p.mpf_alphaS) * n->int2[j];
        n->lambda_leak[j] = n->p.mpf_eta * n->Z[j];
        n->lambda_rev0[j] = n->p.mpf_rev_gamma * n->lambda_rev0[j] + ((1 - n->p.mpf_rev_gamma / n->p.mpf_rev_delta)) * n->soi2[j];
        n->lambda_rev1[j] = n->p.mpf_rev_gamma * n->lambda_rev1[j] + ((1 - n->p.mpf_rev_gamma / n->p.mpf_rev_delta)) * n->int2[j];
        n->lambda_mpf[j] = n->lambda_noise[j] + n->lambda_leak[j] + n->lambda_rev0[j] + n->lambda_rev1[j];
        n->lambda_mpf[j] = sqrt(n->lambda_mpf[j]);
    }
    n->y_fft[0] = 0.0; /* Q15d: reference leaves y_fft[0] unwritten; defined as 0 */
    for (int j = 1; j < N; j++) { /* phasempf.cpp:275-295 */
        double pha_mean = std::arg(n->out_soi[j]);
        double mag_mean;
        if (n->p.out_only_noise) {
            mag_mean = n->lambda_mpf[j] * n->p.out_amp;
        } else {
            if (n->p.out_only_mcra) {
                mag_mean = (std::abs(n->out_soi[j]) - sqrt(n->lambda_noise[j])) * n->p.out_amp;
            } else {
                mag_mean = (std::abs(n->out_soi[j]) - n->lambda_mpf[j]) * n->p.out_amp;
            }
            if (mag_mean < 0) mag_mean = n->p.noise_floor;
        }
        n->y_fft[j] = cd(mag_mean * cos(pha_mean), mag_mean * sin(pha_mean));
    }
}


/* ------------------------------------------------------------------ gsc -- */
/* gsc.cpp:54-75: per-microphone phase alignment, weight_func of do_overlap_bymic */
static void gsc_apply_weights(orc_node *n, int mic, float *out) {
    prepare_input(n, mic, n->x_time.data());
    dft_pow2(n->x_time.data(), n->x_fft.data(), n->N, -1);
    for (int i = 0; i < n->N; i++) n->x_fft[i] *= std::conj(n->weights[i](mic, 0));
    for (int i = 0; i < n->N; i++) n->y_fft[i] = n->x_fft[i];
    dft_pow2(n->y_fft.data(), n->y_time.data(), n->N, +1);
    prepare_output(n, n->y_time.data(), out);
}

/* gsc.cpp:77-82 */
static void gsc_shift_data(float data, float *buf, int size) {
    for (int i = 1; i < size; i++) buf[i - 1] = buf[i];
    buf[size - 1] = data;
}

/* gsc.cpp:84-91 (rosjack_data = float: every product and sum is rounded to float) */
static float gsc_calculate_power(const float *buf, int size) {
    float p = 0.0;
    for (int i = 0; i < size; i++) p += buf[i] * buf[i];
    p /= (float)size;
    return sqrt(p);
}

/* gsc.cpp:93-197: one jack_callback.  in = [M][H] planar, out = [H] */
static void gsc_callback(orc_node *n, const float *in, float *out) {
    const int M = n->M, H = n->H, fs = n->p.gsc_filter_size;
    std::vector<std::vector<float> > overlap_out(M, std::vector<float>(H, 0.0f));
    /* do_overlap_bymic (util.h:353-379) */
    for (int i = 0; i < M; ++i) {
        memcpy(n->in_ring[i].data() + H, in + (size_t)i * H, sizeof(float) * H);
        gsc_apply_weights(n, i, n->out_buff_mic[1][i].data());
        for (int j = 0; j < H; ++j) overlap_out[i][j] = n->out_buff_mic[0][i][j + H] + n->out_buff_mic[1][i][j];
        n->out_buff_mic[0][i].swap(n->out_buff_mic[1][i]);
    }
    for (int i = 0; i < M; ++i) memmove(n->in_ring[i].data(), n->in_ring[i].data() + H, sizeof(float) * H);

    for (int j = 0; j < H; ++j) { /* gsc.cpp:120-181 */
        float das_out = 0.0;
        for (int i = 0; i < M; ++i) das_out += overlap_out[i][j];
        das_out /= M;
        out[j] = das_out;
        for (int i = 0; i < M - 1; ++i) {
            gsc_shift_data(overlap_out[i + 1][j] - overlap_out[i][j], n->block_matrix[i].data(), fs);
            float block_out = 0.0;
            for (int k = 0; k < fs; ++k) block_out += n->filter[i][k] * n->block_matrix[i][k];
            out[j] -= block_out;
        }
        gsc_shift_data(out[j], n->last_outputs.data(), fs);
        float last_out_power = gsc_calculate_power(n->last_outputs.data(), fs);
        if (last_out_power < n->p.gsc_vad_threshold || (!n->p.gsc_use_vad)) {
            for (int i = 0; i < M - 1; i++) {
                float block_power = gsc_calculate_power(n->block_matrix[i].data(), fs);
                float this_mu;
                if (n->p.gsc_mu0 * block_power / last_out_power < n->p.gsc_mu_max) {
                    this_mu = n->p.gsc_mu0 / last_out_power;
                } else {
                    this_mu = n->p.gsc_mu0 / block_power;
                }
                if (std::isnan(this_mu) || std::isinf(this_mu)) this_mu = 0.0;
                for (int k = 0; k < fs; ++k) {
                    n->filter[i][k] += this_mu * out[j] * n->block_matrix[i][k];
                    if (std::isnan(n->filter[i][k])) n->filter[i][k] = 0.0;
                }
            }
        }
    }
}

/* the node's apply_weights(in_buff, out_buff[1]) */
static void apply_weights(orc_node *n, float *out, double *Ydump) {
    switch (n->p.algo) {
        case ORC_DAS: apply_das(n, out); break;
        case ORC_MVDR: apply_mvdr_lcmv(n, false); break;
        case ORC_LCMV: apply_mvdr_lcmv(n, true); break;
        case ORC_GSS: apply_gss(n); break;
        case ORC_PHASE: apply_phase(n); break;
        case ORC_MCRA: apply_mcra_node(n); break;
        default: apply_phasempf(n); break;
    }
    if (Ydump) memcpy(Ydump, n->y_fft.data(), sizeof(cd) * n->N);
    dft_pow2(n->y_fft.data(), n->y_time.data(), n->N, +1); /* fftw_execute(y_inverse) */
    prepare_output(n, n->y_time.data(), out);
    if (n->p.algo == ORC_MVDR || n->p.algo == ORC_LCMV || n->p.algo == ORC_GSS) /* mvdr.cpp:112-114 */
        for (int j = 0; j < n->N; j++) out[j] *= n->p.out_amp;
}

extern "C" {

orc_node *orc_create(const orc_params *p) {
    orc_node *n = new orc_node();
    n->p = *p;
    n->M = p->n_mics;
    n->H = p->hop;
    n->N = 2 * p->hop; /* util.h:261 */
    bool multi = (p->algo == ORC_LCMV || p->algo == ORC_GSS);
    n->S = multi ? p->n_interf + 1 : 1;
    n->angle = p->theta;
    /* handle_params, util.h:82-92 */
    for (int i = 0; i < n->M; i++) {
        n->mic_dist.push_back(sqrt(p->mic_x[i] * p->mic_x[i] + p->mic_y[i] * p->mic_y[i]));
        n->mic_ang.push_back(atan2(p->mic_y[i], p->mic_x[i]) * rad2deg);
    }
    for (int k = 0; k < n->S - 1; k++) n->interference_angles.push_back(p->interf_angle[k]);
    /* prepare_overlap_and_add, util.h:257-287 */
    n->hann_win.resize(n->N);
    for (int i = 0; i < n->N; ++i)
        n->hann_win[i] = sqrt(0.5 - 0.5 * cos(2 * ORC_PI * (unsigned int)i / ((unsigned int)n->N))); /* util.h:201-211 */
    n->in_ring.assign(n->M, std::vector<float>(n->N, 0.0f)); /* one hop of zeros pre-written */
    n->out_buff[0].assign(n->N, 0.0f);
    n->out_buff[1].assign(n->N, 0.0f);
    n->x_time.resize(n->N); n->x_fft.resize(n->N); n->y_fft.assign(n->N, cd(0, 0)); n->y_time.resize(n->N);
    n->freqs.resize(n->N);
    calc_freqs(n->freqs.data(), n->N, p->sample_rate);
    n->delays.resize(n->M);
    n->interf_delays.assign(n->S - 1 > 0 ? n->S - 1 : 0, std::vector<double>(n->M));
    n->weights.assign(n->N, Mat(n->M, n->S));
    n->weights_h.assign(n->N, Mat(n->S, n->M));
    n->in_fft = Mat(n->M, n->N);
    if (p->algo == ORC_MVDR || p->algo == ORC_LCMV) {
        n->past_ffts.assign(n->N, Mat(n->M, p->past_windows)); /* setZero, mvdr.cpp:228-232 */
        n->whiteR = Mat(n->M, n->M);
        for (int a = 0; a < n->M; a++)
            for (int b = 0; b < n->M; b++) n->whiteR(a, b) = (a == b) ? 1.001 : 1.0; /* mvdr.cpp:239-243 */
    }
    if (p->algo == ORC_GSS) n->sep_matrix.assign(n->N, Mat(n->S, n->M));
    n->phases_aligned.resize(n->M);
    if (p->algo == ORC_GSC) { /* gsc.cpp:278-285, util.h:341-346: everything calloc'ed */
        for (int b = 0; b < 2; b++) n->out_buff_mic[b].assign(n->M, std::vector<float>(n->N, 0.0f));
        n->block_matrix.assign(n->M > 1 ? n->M - 1 : 0, std::vector<float>(p->gsc_filter_size, 0.0f));
        n->filter = n->block_matrix;
        n->last_outputs.assign(p->gsc_filter_size, 0.0f);
    }
    if (p->algo == ORC_MCRA) { /* mcra.cpp:257-271 */
        std::vector<double> z(n->N, 0.0);
        n->soi2 = z; n->S_ = z; n->S_prev = z; n->S_f = z; n->S_tmp = z; n->S_min = z; n->lambda_noise = z;
    }
    if (p->algo == ORC_PHASEMPF) {
        n->past_samples.assign(p->smooth_size, 0.0); /* calloc, phasempf.cpp:510 */
        n->out_soi.assign(n->N, cd(0, 0)); n->out_int.assign(n->N, cd(0, 0));
        std::vector<double> z(n->N, 0.0);
        n->soi2 = z; n->int2 = z; n->S_ = z; n->S_prev = z; n->S_f = z; n->S_tmp = z; n->S_min = z; n->Z = z;
        n->lambda_noise = z; n->lambda_leak = z; n->lambda_rev0 = z; n->lambda_rev1 = z; n->lambda_mpf = z;
    }
    n->current_L = 0;  /* phasempf.cpp:46-47 */
    n->first_L = true;
    update_weights(n, true);
    return n;
}

void orc_destroy(orc_node *n) { delete n; }

void orc_set_theta(orc_node *n, double deg) {
    n->angle = deg;
    update_weights(n, false);
}

/* free_interf_buffers + allocate_interf_buffers (lcmv.cpp:221-256, gss.cpp:242-286): everything sized by the
 * interferer count is rebuilt zeroed */
static void realloc_interf(orc_node *n) {
    n->S = (int)n->interference_angles.size() + 1;
    n->interf_delays.assign(n->S - 1, std::vector<double>(n->M));
    n->weights.assign(n->N, Mat(n->M, n->S));
    n->weights_h.assign(n->N, Mat(n->S, n->M));
    if (n->p.algo == ORC_GSS) n->sep_matrix.assign(n->N, Mat(n->S, n->M));
}

int orc_set_interference(orc_node *n, unsigned id, double angle, double thr) {
    std::vector<double> &ia = n->interference_angles;
    if (id >= 1 && id <= ia.size()) {
        ia[id - 1] = angle;
        for (size_t i = 0; i < ia.size(); i++) {
            if (i != (id - 1) && std::abs(ia[i] - angle) < thr) {
                ia.erase(ia.begin() + id - 1);
                realloc_interf(n);
                break;
            }
        }
        update_weights(n, false);
    } else if (id > ia.size()) {
        size_t i;
        for (i = 0; i < ia.size(); i++)
            if (std::abs(ia[i] - angle) < thr) break;
        if (i == ia.size()) {
            ia.push_back(angle);
            realloc_interf(n);
            update_weights(n, false);
        }
    }
    return (int)ia.size();
}

/* jack_callback -> do_overlap (util.h:289-314) */
int orc_process_hop(orc_node *n, const float *in, float *out, double *Y) {
    const int H = n->H;
    if (n->p.algo == ORC_GSC) { /* time-domain output: there is no single y_fft to report */
        gsc_callback(n, in, out);
        if (Y) memset(Y, 0, sizeof(double) * 2 * n->N);
        return 0;
    }
    for (int i = 0; i < n->M; ++i) /* jack_ringbuffer_write: ring now holds [prev hop, this hop] */
        memcpy(n->in_ring[i].data() + H, in + (size_t)i * H, sizeof(float) * H);
    apply_weights(n, n->out_buff[1].data(), Y);
    for (int j = 0; j < H; ++j) out[j] = n->out_buff[0][j + H] + n->out_buff[1][j];
    for (int i = 0; i < n->M; ++i) /* jack_ringbuffer_read_advance */
        memmove(n->in_ring[i].data(), n->in_ring[i].data() + H, sizeof(float) * H);
    n->out_buff[0].swap(n->out_buff[1]);
    if (n->p.algo == ORC_PHASEMPF) { /* phasempf.cpp:331-334 */
        const int sz = n->p.smooth_size;
        for (int j = 0; j < H; j++) {
            for (int i = 1; i < sz; i++) n->past_samples[i - 1] = n->past_samples[i];
            n->past_samples[sz - 1] = out[j];
            double s = 0.0;
            for (int i = 0; i < sz; i++) s += n->past_samples[i];
            out[j] = s / (double)sz;
        }
    }
    return 0;
}

int orc_process(orc_node *n, const float *x, long n_frames, float *y, double *Y) {
    const int H = n->H;
    std::vector<float> in((size_t)n->M * H);
    for (long t = 0; t < n_frames; t++) {
        for (int m = 0; m < n->M; m++)
            memcpy(in.data() + (size_t)m * H, x + (size_t)m * n_frames * H + (size_t)t * H, sizeof(float) * H);
        orc_process_hop(n, in.data(), y + (size_t)t * H, Y ? Y + (size_t)t * n->N * 2 : 0);
    }
    return 0;
}

void orc_get_freqs(const orc_node *n, double *f) { memcpy(f, n->freqs.data(), sizeof(double) * n->N); }
void orc_get_delays(const orc_node *n, double *d) { memcpy(d, n->delays.data(), sizeof(double) * n->M); }
void orc_get_hann(const orc_node *n, double *h) { memcpy(h, n->hann_win.data(), sizeof(double) * n->N); }
void orc_get_weights(const orc_node *n, double *w) {
    size_t q = 0;
    for (int j = 0; j < n->N; j++)
        for (int i = 0; i < n->M; i++)
            for (int s = 0; s < n->S; s++) {
                w[q++] = n->weights[j](i, s).real();
                w[q++] = n->weights[j](i, s).imag();
            }
}
void orc_fft(const double *in, double *out, int nn, int sign) {
    dft_pow2((const cd *)in, (cd *)out, nn, sign);
}

} /* extern "C" */
