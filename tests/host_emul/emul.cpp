// emul.cpp -- CPU emulation of the half-wavefront algorithms in beamform_amd/csrc
// (test infrastructure).  The very same templates that hipcc compiles for gfx950
// are instantiated here with the 32 lanes of a half-wavefront run as a loop, so
// index maps, twiddle/gain tables and the pair-packing algebra are checked on the
// CPU before any GPU time is spent.
#include <cstring>
#include <vector>

#include "../../beamform_amd/csrc/fft1024.hpp"
#include "../../beamform_amd/csrc/geometry.hpp"

using namespace bf;

template <typename T>
static void fft1024_emul(const double *in, double *out, int dir) {
    constexpr int RS = tr_stride<T>::value;
    std::vector<cx<T>> tw(1024), buf(32 * RS);
    for (int k = 0; k < 32; ++k)
        for (int l = 0; l < 32; ++l) {
            double a = -2.0 * kPi * (k * l) / 1024.0;
            tw[k * 32 + l] = cx<T>{(T)std::cos(a), (T)std::sin(a)};
        }
    static T re[32][32], im[32][32];
    if (dir < 0) {
        for (int l = 0; l < 32; ++l) {
            for (int j = 0; j < 32; ++j) {
                re[l][j] = (T)in[2 * (32 * j + l)];
                im[l][j] = (T)in[2 * (32 * j + l) + 1];
            }
            fft1024_fwd_a<T>(re[l], im[l], l, tw.data(), buf.data());
        }
        for (int l = 0; l < 32; ++l) {
            fft1024_fwd_b<T>(re[l], im[l], l, buf.data());
            for (int i = 0; i < 32; ++i) {
                int k = l + 32 * brev5(i);
                out[2 * k] = re[l][i];
                out[2 * k + 1] = im[l][i];
            }
        }
    } else {
        for (int l = 0; l < 32; ++l) {
            for (int i = 0; i < 32; ++i) {
                int k = l + 32 * brev5(i);
                re[l][i] = (T)in[2 * k];
                im[l][i] = (T)in[2 * k + 1];
            }
            fft1024_inv_a<T>(re[l], im[l], l, tw.data(), buf.data());
        }
        for (int l = 0; l < 32; ++l) {
            fft1024_inv_b<T>(re[l], im[l], l, buf.data());
            for (int i = 0; i < 32; ++i) {
                int n = 32 * brev5(i) + l;
                out[2 * n] = re[l][i];
                out[2 * n + 1] = im[l][i];
            }
        }
    }
}

extern "C" {

void emul_fft32(const double *in, double *out, int dir, int dit) {
    double re[32], im[32];
    for (int i = 0; i < 32; ++i) {
        int src = dit ? brev5(i) : i;
        re[i] = in[2 * src];
        im[i] = in[2 * src + 1];
    }
    if (dit) {
        if (dir < 0) fft32_dit<double, -1>(re, im); else fft32_dit<double, +1>(re, im);
    } else {
        if (dir < 0) fft32_dif<double, -1>(re, im); else fft32_dif<double, +1>(re, im);
    }
    for (int i = 0; i < 32; ++i) {
        int dst = dit ? i : brev5(i);
        out[2 * dst] = re[i];
        out[2 * dst + 1] = im[i];
    }
}

void emul_fft1024(const double *in, double *out, int dir, int use_float) {
    if (use_float) fft1024_emul<float>(in, out, dir); else fft1024_emul<double>(in, out, dir);
}

// Fused-DAS algorithm (pair packing + D gains + unpaired inverse), float arithmetic
// exactly as the kernel does it.  x planar [M][F*H], y [F*H].
void emul_das_fused(int M, int H, double sr, const double *mx, const double *my, double theta,
                    const float *x, long F, float *y) {
    const int N = 2 * H;
    ArrayGeometry g;
    g.set(mx, my, M);
    std::vector<double> freqs = frequency_vector(N, sr);
    SteeringSet st;
    st.allocate(N, M, 1);
    st.update_column(g, freqs, 0, theta, true);
    const int NP = (M + 1) / 2;
    std::vector<f32x2> D = das_pair_gains(st, NP);
    std::vector<double> hd = sqrt_hann(N);
    std::vector<float> h(N);
    for (int i = 0; i < N; ++i) h[i] = (float)hd[i];
    constexpr int RS = tr_stride<float>::value;
    std::vector<cx<float>> tw(1024), buf(32 * RS);
    {
        std::vector<f32x2> t = twiddle_table_32x32<f32x2>();
        for (int i = 0; i < 1024; ++i) tw[i] = cx<float>{t[i].x, t[i].y};
    }
    static float re[32][32], im[32][32], Sr[32][32], Si[32][32];
    std::vector<float> tail(H, 0.f);
    for (long t = 0; t < F; ++t) {
        for (int p = 0; p < NP; ++p) {
            int a = 2 * p, b = 2 * p + 1;
            for (int l = 0; l < 32; ++l) {
                for (int j = 0; j < 32; ++j) {
                    int n = 32 * j + l;
                    long s = (t - 1) * (long)H + n;
                    float va = s < 0 ? 0.f : x[(size_t)a * F * H + s];
                    float vb = (b < M) ? (s < 0 ? 0.f : x[(size_t)b * F * H + s]) : 0.f;
                    re[l][j] = va * h[n];
                    im[l][j] = vb * h[n];
                }
                fft1024_fwd_a<float>(re[l], im[l], l, tw.data(), buf.data());
            }
            for (int l = 0; l < 32; ++l) {
                fft1024_fwd_b<float>(re[l], im[l], l, buf.data());
                for (int i = 0; i < 32; ++i) {
                    f32x2 d = D[((size_t)p * 32 + i) * 32 + l];
                    float pr = d.x * re[l][i] - d.y * im[l][i];
                    float pi = d.x * im[l][i] + d.y * re[l][i];
                    if (p == 0) { Sr[l][i] = pr; Si[l][i] = pi; } else { Sr[l][i] += pr; Si[l][i] += pi; }
                }
            }
        }
        for (int l = 0; l < 32; ++l) fft1024_inv_a<float>(Sr[l], Si[l], l, tw.data(), buf.data());
        for (int l = 0; l < 32; ++l) {
            fft1024_inv_b<float>(Sr[l], Si[l], l, buf.data());
            for (int i = 0; i < 32; ++i) {
                int n = 32 * brev5(i) + l;
                float o = Sr[l][i] * h[n];
                if (n < H) y[(size_t)t * H + n] = tail[n] + o; 
            }
            for (int i = 0; i < 32; ++i) {
                int n = 32 * brev5(i) + l;
                if (n >= H) tail[n - H] = Sr[l][i] * h[n];
            }
        }
    }
}

// Host geometry accessors for known-answer tests.
void emul_freqs(int N, double sr, double *f) {
    std::vector<double> v = frequency_vector(N, sr);
    memcpy(f, v.data(), sizeof(double) * N);
}
void emul_delays(int M, const double *mx, const double *my, double theta, double *tau) {
    ArrayGeometry g;
    g.set(mx, my, M);
    g.delays(theta, tau);
}
void emul_hann(int N, double *h) {
    std::vector<double> v = sqrt_hann(N);
    memcpy(h, v.data(), sizeof(double) * N);
}
}
