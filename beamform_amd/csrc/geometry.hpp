// geometry.hpp -- host-side (double precision) array geometry, frequency vector,
// steering weights and the derived device tables.
//
// Mirrors, for this path only, what the reference computes at start-up and on
// every /theta message:
//   handle_params            util.h:52-134   (dist/angle from RAW yaml xy: quirk Q2)
//   calculate_delays         util.h:136-161
//   calculate_frequency_vector util.h:190-199 (quirk Q1: f[N/2-1]=sr/2, f[N/2] undefined -> 0)
//   hann / create_hann_winn  util.h:201-211
//   update_weights           das.cpp:27-45, lcmv.cpp:44-86
// Header-only so the CPU-side unit tests can compile it with g++.
#pragma once

#include <cmath>
#include <complex>
#include <vector>

#include "fft1024_w64.hpp"
#include "fft32.hpp"

namespace bf {

typedef std::complex<double> cplxd;

constexpr double kPi = 3.141592653589793238462643383279502884;
constexpr double kSoundSpeed = 343.0;  // util.h:25

struct ArrayGeometry {
    int n_mics = 0;
    std::vector<double> dist;     // sqrt(x^2+y^2) of the raw coordinates
    std::vector<double> ang_deg;  // atan2(y,x) in degrees of the raw coordinates

    void set(const double *x, const double *y, int n) {
        n_mics = n;
        dist.resize(n);
        ang_deg.resize(n);
        for (int i = 0; i < n; ++i) {
            dist[i] = std::sqrt(x[i] * x[i] + y[i] * y[i]);
            ang_deg[i] = std::atan2(y[i], x[i]) * (180.0 / kPi);
        }
    }

    // Far-field delay of every mic relative to mic 0 for a source at `angle` degrees.
    void delays(double angle, double *tau) const {
        for (int i = 0; i < n_mics; ++i) {
            if (i == 0) {
                tau[i] = 0.0;
                continue;
            }
            double d = ang_deg[i] - angle;
            if (d > 180)
                d -= 360;
            else if (d < -180)
                d += 360;
            tau[i] = dist[i] * std::cos(d * (kPi / 180.0)) / (-kSoundSpeed);
        }
    }
};

// FFT-bin -> Hz including the reference's two off-by-one quirks (Q1).
inline std::vector<double> frequency_vector(int n_fft, double sample_rate) {
    std::vector<double> f(n_fft, 0.0);
    for (int k = 1; k < n_fft / 2; ++k) {
        f[k] = ((double)k / (double)n_fft) * sample_rate;
        f[n_fft - k] = -((double)k / (double)n_fft) * sample_rate;
    }
    f[n_fft / 2 - 1] = sample_rate / 2;
    return f;
}

// Periodic sqrt-Hann analysis/synthesis window.
inline std::vector<double> sqrt_hann(int n_fft) {
    std::vector<double> h(n_fft);
    for (int i = 0; i < n_fft; ++i) h[i] = std::sqrt(0.5 - 0.5 * std::cos(2 * kPi * i / n_fft));
    return h;
}

// Steering / constraint matrices, [bin][mic][col] with col 0 = look direction and
// col k>=1 = interferer k.  `first` = the reference's update_weights(ini=true):
// row 0 (reference mic) is only ever written then (quirk Q3), so after a
// structural re-allocation it stays 0 until the next cold start.
struct SteeringSet {
    int n_fft = 0, n_mics = 0, n_cols = 1;
    std::vector<cplxd> w;  // [n_fft][n_mics][n_cols]

    cplxd &at(int j, int m, int c) { return w[((size_t)j * n_mics + m) * n_cols + c]; }
    const cplxd &at(int j, int m, int c) const { return w[((size_t)j * n_mics + m) * n_cols + c]; }

    void allocate(int nfft, int mics, int cols) {
        n_fft = nfft;
        n_mics = mics;
        n_cols = cols;
        w.assign((size_t)nfft * mics * cols, cplxd(0, 0));
    }

    void update_column(const ArrayGeometry &g, const std::vector<double> &freqs, int col, double angle, bool first) {
        std::vector<double> tau(n_mics);
        g.delays(angle, tau.data());
        const cplxd minus_i(0, -1);
        for (int m = 0; m < n_mics; ++m) {
            if (m == 0) {
                if (first)
                    for (int j = 0; j < n_fft; ++j) at(j, 0, col) = 1.0;
            } else {
                for (int j = 0; j < n_fft; ++j) at(j, m, col) = std::exp(minus_i * (double)2 * kPi * freqs[j] * tau[m]);
            }
        }
    }
};

struct f32x2 {
    float x, y;
};
struct f64x2 {
    double x, y;
};

// exp(-2 pi i k l / 1024) for k,l in [0,32): the inter-pass twiddles of the 32x32
// decomposition, laid out [k][l] so that lanes (l) read consecutive addresses.
template <typename V>
inline std::vector<V> twiddle_table_32x32() {
    std::vector<V> t(1024);
    for (int k = 0; k < 32; ++k)
        for (int l = 0; l < 32; ++l) {
            double a = -2.0 * kPi * (double)(k * l) / 1024.0;
            t[k * 32 + l].x = (decltype(t[0].x))std::cos(a);
            t[k * 32 + l].y = (decltype(t[0].x))std::sin(a);
        }
    return t;
}

// Per-pair complex gains of the fused DAS kernel.
//
// The reference computes y = Re IFFT( sum_m conj(w[m,k]) X_m[k] / M ) (das.cpp:60-66,
// util.h:249).  Two real mics (a,b) are transformed as one complex signal
// Z = FFT(a + i b); with c_m = conj(w_m)/M and its Hermitian part
// ce_m[k] = (c_m[k] + conj(c_m[N-k]))/2 the same real output is
//   y = Re IFFT( sum_pairs D_p[k] Z_p[k] ),   D_p = ce_a - i ce_b
// (derivation in DESIGN.md).  1/N of util.h:249 is folded in.  Odd mic counts
// get a zero b-channel.  Layout: [pair][pos i][lane l] = D_p[l + 32*brev5(i)],
// the register/lane order in which fft1024 leaves the spectrum.
template <typename V>
inline std::vector<V> das_pair_gains_t(const SteeringSet &s, int n_pairs_alloc) {
    const int N = s.n_fft, M = s.n_mics;
    std::vector<V> D((size_t)n_pairs_alloc * N, V{0, 0});
    auto ce = [&](int m, int k) -> cplxd {
        if (m >= M) return cplxd(0, 0);
        cplxd c1 = std::conj(s.at(k, m, 0)) / (double)M;
        cplxd c2 = std::conj(s.at((N - k) % N, m, 0)) / (double)M;
        return 0.5 * (c1 + std::conj(c2));
    };
    for (int p = 0; p < (M + 1) / 2; ++p)
        for (int i = 0; i < 32; ++i)
            for (int l = 0; l < 32; ++l) {
                int k = l + 32 * brev5(i);
                cplxd d = (ce(2 * p, k) - cplxd(0, 1) * ce(2 * p + 1, k)) / (double)N;
                D[((size_t)p * 32 + i) * 32 + l] = V{(decltype(V{}.x))d.real(), (decltype(V{}.x))d.imag()};
            }
    return D;
}
inline std::vector<f32x2> das_pair_gains(const SteeringSet &s, int n_pairs_alloc) { return das_pair_gains_t<f32x2>(s, n_pairs_alloc); }
// The same gains in natural bin order [pair][k], any FFT size (das_fused_gen.hip: JACK periods 256 and 1024).
inline std::vector<f32x2> das_pair_gains_natural(const SteeringSet &s, int n_pairs_alloc) {
    const int N = s.n_fft, M = s.n_mics;
    std::vector<f32x2> D((size_t)n_pairs_alloc * N, f32x2{0.f, 0.f});
    auto ce = [&](int m, int k) -> cplxd {
        if (m >= M) return cplxd(0, 0);
        const cplxd c1 = std::conj(s.at(k, m, 0)) / (double)M, c2 = std::conj(s.at((N - k) % N, m, 0)) / (double)M;
        return 0.5 * (c1 + std::conj(c2));
    };
    for (int p = 0; p < (M + 1) / 2; ++p)
        for (int k = 0; k < N; ++k) {
            const cplxd d = (ce(2 * p, k) - cplxd(0, 1) * ce(2 * p + 1, k)) / (double)N;
            D[(size_t)p * N + k] = f32x2{(float)d.real(), (float)d.imag()};
        }
    return D;
}


// Frame-interleaving (das_fused.hip, group mode; N = 512 / 256 / 128): the N-point pair gains repeated 1024 / N times over the bins of
// the 1024-point transform, 1/1024 folded in, in the register / lane order of fft1024: [pair][position i][lane l] = D_p[(l + 32 brev5(i)) mod N]
inline std::vector<f32x2> das_pair_gains_interleaved(const SteeringSet &s, int n_pairs_alloc) {
    const int N = s.n_fft, M = s.n_mics;
    std::vector<f32x2> D((size_t)n_pairs_alloc * 1024, f32x2{0.f, 0.f});
    auto ce = [&](int m, int k) -> cplxd {
        if (m >= M) return cplxd(0, 0);
        const cplxd c1 = std::conj(s.at(k, m, 0)) / (double)M, c2 = std::conj(s.at((N - k) % N, m, 0)) / (double)M;
        return 0.5 * (c1 + std::conj(c2));
    };
    for (int p = 0; p < (M + 1) / 2; ++p)
        for (int i = 0; i < 32; ++i)
            for (int l = 0; l < 32; ++l) {
                const int k = (l + 32 * brev5(i)) % N;
                const cplxd d = (ce(2 * p, k) - cplxd(0, 1) * ce(2 * p + 1, k)) / 1024.0;
                D[((size_t)p * 32 + i) * 32 + l] = f32x2{(float)d.real(), (float)d.imag()};
            }
    return D;
}

// Twiddles of the LDS-staged autosort (Stockham) transforms of the generic FFT sizes (das_fused_gen.hip, stft_istft.hip):
//   [0, N/2)            W^m = exp(-2 pi i m / N): the closing radix-2 pass of N = 2 * 4^k reads it in order (and the in-place
//                       transform of N = 8192 with its own strides);
//   [N/2, N/2 + R4)     one block per radix-4 pass ns = 1, 4, 16, ... (4 ns <= N), at offset N/2 + (ns - 1): [q][k] = W^((q + 1) k N / (4 ns)),
//                       q = 0..2, k < ns -- the three twiddles of butterfly k CONTIGUOUS in k.  Read out of the plain W^m table the lanes of
//                       a wavefront (consecutive k) are N / (4 ns) entries apart: up to 16 of them on one LDS bank (measured: half of the
//                       LDS cycles of das_fused_gen_kernel<2048> were bank conflicts).
constexpr int stockham_r4_entries(int n) {
    int ns = 1, tot = 0;
    for (; ns * 4 <= n; ns <<= 2) tot += 3 * ns;
    return tot;
}
template <typename V>
inline std::vector<V> stockham_twiddles(int N) {
    std::vector<V> t((size_t)N / 2 + stockham_r4_entries(N));
    typedef decltype(V{}.x) T;
    for (int m = 0; m < N / 2; ++m) {
        const double a = -2.0 * kPi * (double)m / (double)N;
        t[m] = V{(T)std::cos(a), (T)std::sin(a)};
    }
    for (int ns = 1; ns * 4 <= N; ns <<= 2)
        for (int q = 0; q < 3; ++q)
            for (int k = 0; k < ns; ++k) {
                const double a = -2.0 * kPi * (double)((long)(q + 1) * k * (N / (4 * ns))) / (double)N;
                t[(size_t)N / 2 + (ns - 1) + (size_t)q * ns + k] = V{(T)std::cos(a), (T)std::sin(a)};
            }
    return t;
}

// fp64 tables of the 64-lane factorisation with the rotated exchange (fft1024_w64.hpp w64_col_rot; das_f64_w64.hip):
// [0, 1024) = W1024^(k1*lane) as [k1][lane]; [1024, 1092) = tw2'[b][k2] = exp(2 pi i 15 b k2 / 64) in rows of 17 (one element of
// padding: the four rows a wavefront reads at once then start 4 LDS banks apart instead of on the same one)
constexpr int kTw2RowW64Rot = 17;
inline std::vector<f64x2> twiddle_table_w64_rot() {
    std::vector<f64x2> t(1024 + 4 * kTw2RowW64Rot, f64x2{0.0, 0.0});
    for (int k = 0; k < 16; ++k)
        for (int l = 0; l < 64; ++l) {
            const double a = -2.0 * kPi * (double)(k * l) / 1024.0;
            t[k * 64 + l] = f64x2{std::cos(a), std::sin(a)};
        }
    for (int b = 0; b < 4; ++b)
        for (int k = 0; k < 16; ++k) {
            const double a = 2.0 * kPi * (double)((15 * b * k) % 64) / 64.0;
            t[1024 + b * kTw2RowW64Rot + k] = f64x2{std::cos(a), std::sin(a)};
        }
    return t;
}
// das_pair_gains_t<f64x2> (32 x 32 order) -> [pair][register r][lane] = D_p[w64_bin(lane, r)]
inline std::vector<f64x2> das_pair_gains_w64_f64(const std::vector<f64x2> &D32, int n_pairs) {
    std::vector<f64x2> D((size_t)n_pairs * 1024);
    std::vector<f64x2> nat(1024);
    for (int p = 0; p < n_pairs; ++p) {
        for (int i = 0; i < 32; ++i)
            for (int l = 0; l < 32; ++l) nat[l + 32 * brev5(i)] = D32[((size_t)p * 32 + i) * 32 + l];
        for (int r = 0; r < 16; ++r)
            for (int l = 0; l < 64; ++l) D[((size_t)p * 16 + r) * 64 + l] = nat[w64_bin(l, r)];
    }
    return D;
}

// Per-microphone Hermitian-part gains of the frame-pair kernel (das_f64_w64.hip das_f64_pair_kernel): ce_m[k] / N for the bins 0 .. 512
// only (ce_m[N - k] = conj ce_m[k]), [mic][row 2 g + k3, k3 < 2][65]: entry c of row (g, k3) = bin 64 g + 256 k3 + c, c = 0 .. 64.
constexpr int kDasMicGainRow = 65, kDasMicGainRows = 8;
inline std::vector<f64x2> das_mic_gains_w64_f64(const SteeringSet &s, int n_mics_alloc) {
    const int N = s.n_fft, M = s.n_mics;
    std::vector<f64x2> T((size_t)n_mics_alloc * kDasMicGainRows * kDasMicGainRow, f64x2{0, 0});
    for (int m = 0; m < M && m < n_mics_alloc; ++m)
        for (int g = 0; g < 4; ++g)
            for (int k3 = 0; k3 < 2; ++k3)
                for (int c = 0; c <= 64; ++c) {
                    const int k = 64 * g + 256 * k3 + c;
                    const cplxd c1 = std::conj(s.at(k, m, 0)) / (double)M, c2 = std::conj(s.at((N - k) % N, m, 0)) / (double)M;
                    const cplxd ce = 0.5 * (c1 + std::conj(c2)) / (double)N;
                    T[((size_t)m * kDasMicGainRows + 2 * g + k3) * kDasMicGainRow + c] = f64x2{ce.real(), ce.imag()};
                }
    return T;
}

}  // namespace bf
