// emul.cpp -- CPU emulation of the half-wavefront algorithms in beamform_amd/csrc
// (test infrastructure).  The very same templates that hipcc compiles for gfx950
// are instantiated here with the 32 lanes of a half-wavefront run as a loop, so
// index maps, twiddle/gain tables and the pair-packing algebra are checked on the
// CPU before any GPU time is spent.
#include <cstring>
#include <vector>

#include "../../beamform_amd/csrc/fft1024.hpp"
#include "../../beamform_amd/csrc/fft1024_w64.hpp"
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


// ---- 64-lane x 16-point factorisation (fft1024_w64.hpp) -----------------------------------------
namespace {
constexpr int kRS64 = 68;
inline int brev2(int i) { return ((i & 1) << 1) | ((i >> 1) & 1); }

// T1 forward: reg-position i (k1 = brev4(i)), lane 4a+b  ->  reg a, lane 16b+k1.  The writer stores at row k1, column
// w64_col(lane) = 16 b + a; the reader takes columns 16 b .. 16 b + 15 of row (lane & 15).
// ROT: the fp64 kernel's exchange (w64_col_rot: segment b rotated by 4 b columns; the reader still takes its segment as it lies)
template <typename T, bool ROT = false>
void w64_T1_fwd(T (*re)[16], T (*im)[16]) {
    static T br[16 * kRS64], bi[16 * kRS64];
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 16; ++i) {
            const int c = ROT ? w64_col_rot(l) : w64_col(l);
            br[brev4(i) * kRS64 + c] = re[l][i];
            bi[brev4(i) * kRS64 + c] = im[l][i];
        }
    for (int l = 0; l < 64; ++l)
        for (int a = 0; a < 16; ++a) { re[l][a] = br[(l & 15) * kRS64 + 16 * (l >> 4) + a]; im[l][a] = bi[(l & 15) * kRS64 + 16 * (l >> 4) + a]; }
}
template <typename T, bool ROT = false>
void w64_T1_inv(T (*re)[16], T (*im)[16]) {
    static T br[16 * kRS64], bi[16 * kRS64];
    for (int l = 0; l < 64; ++l)
        for (int a = 0; a < 16; ++a) { br[(l & 15) * kRS64 + 16 * (l >> 4) + a] = re[l][a]; bi[(l & 15) * kRS64 + 16 * (l >> 4) + a] = im[l][a]; }
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 16; ++i) {
            const int c = ROT ? w64_col_rot(l) : w64_col(l);
            re[l][i] = br[brev4(i) * kRS64 + c];
            im[l][i] = bi[brev4(i) * kRS64 + c];
        }
}
// T2 forward: position g' + 4*brev2(q) at lane (row b, ..)  ->  register 4*g + b at lane (row q, ..); g' = brev2(g)
template <typename T>
void w64_T2_fwd(T (*re)[16], T (*im)[16]) {
    static T nr[64][16], ni[64][16];
    for (int l = 0; l < 64; ++l)
        for (int g = 0; g < 4; ++g)
            for (int q = 0; q < 4; ++q) {
                const int b = l >> 4, src_pos = brev2(g) + 4 * brev2(q), dl = (l & 15) | (q << 4);
                nr[dl][4 * g + b] = re[l][src_pos];
                ni[dl][4 * g + b] = im[l][src_pos];
            }
    memcpy(re, nr, sizeof(nr));
    memcpy(im, ni, sizeof(ni));
}
template <typename T>
void w64_T2_inv(T (*re)[16], T (*im)[16]) {
    static T nr[64][16], ni[64][16];
    for (int l = 0; l < 64; ++l)
        for (int g = 0; g < 4; ++g)
            for (int b = 0; b < 4; ++b) {
                const int q = l >> 4, dst_pos = brev2(g) + 4 * brev2(q), dl = (l & 15) | (b << 4);
                nr[dl][dst_pos] = re[l][4 * g + b];
                ni[dl][dst_pos] = im[l][4 * g + b];
            }
    memcpy(re, nr, sizeof(nr));
    memcpy(im, ni, sizeof(ni));
}

template <typename T, bool ROT = false>
void fft1024_w64_emul(const double *in, double *out, int dir) {
    std::vector<cx<T>> tw1(16 * 64), tw2(4 * 16);
    for (int k = 0; k < 16; ++k)
        for (int l = 0; l < 64; ++l) {
            double a = -2.0 * kPi * (k * l) / 1024.0;
            tw1[k * 64 + l] = cx<T>{(T)std::cos(a), (T)std::sin(a)};
        }
    for (int b = 0; b < 4; ++b)
        for (int k = 0; k < 16; ++k) {
            double a = ROT ? 2.0 * kPi * ((15 * b * k) % 64) / 64.0 : -2.0 * kPi * (b * k) / 64.0;  // twiddle_table_w64_rot
            tw2[b * 16 + k] = cx<T>{(T)std::cos(a), (T)std::sin(a)};
        }
    static T re[64][16], im[64][16];
    if (dir < 0) {
        for (int l = 0; l < 64; ++l) {
            for (int j = 0; j < 16; ++j) { re[l][j] = (T)in[2 * (64 * j + l)]; im[l][j] = (T)in[2 * (64 * j + l) + 1]; }
            w64_fwd_p1<T>(re[l], im[l], l, tw1.data());
        }
        w64_T1_fwd<T, ROT>(re, im);
        for (int l = 0; l < 64; ++l) w64_fwd_p2<T>(re[l], im[l], l, tw2.data());
        w64_T2_fwd<T>(re, im);
        for (int l = 0; l < 64; ++l) {
            w64_fwd_p3<T>(re[l], im[l]);
            for (int r = 0; r < 16; ++r) { out[2 * w64_bin(l, r)] = re[l][r]; out[2 * w64_bin(l, r) + 1] = im[l][r]; }
        }
    } else {
        for (int l = 0; l < 64; ++l) {
            for (int r = 0; r < 16; ++r) { re[l][r] = (T)in[2 * w64_bin(l, r)]; im[l][r] = (T)in[2 * w64_bin(l, r) + 1]; }
            w64_inv_p3<T>(re[l], im[l]);
        }
        w64_T2_inv<T>(re, im);
        for (int l = 0; l < 64; ++l) w64_inv_p2<T>(re[l], im[l], l, tw2.data());
        w64_T1_inv<T, ROT>(re, im);
        for (int l = 0; l < 64; ++l) {
            w64_inv_p1<T>(re[l], im[l], l, tw1.data());
            for (int j = 0; j < 16; ++j) { out[2 * (64 * j + l)] = re[l][j]; out[2 * (64 * j + l) + 1] = im[l][j]; }
        }
    }
}
}  // namespace

extern "C" void emul_fft1024_w64(const double *in, double *out, int dir, int use_float) {
    if (use_float) fft1024_w64_emul<float>(in, out, dir); else fft1024_w64_emul<double>(in, out, dir);
}
// the fp64 kernel's variant: rotated exchange + tw2' table (das_f64_w64.hip)
extern "C" void emul_fft1024_w64_rot(const double *in, double *out, int dir) { fft1024_w64_emul<double, true>(in, out, dir); }

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
