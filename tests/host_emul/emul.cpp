// emul.cpp -- CPU emulation of the half-wavefront algorithms in beamform_amd/csrc
// (test infrastructure).  The very same templates that hipcc compiles for gfx950
// are instantiated here with the 32 lanes of a half-wavefront run as a loop, so
// index maps, twiddle/gain tables and the pair-packing algebra are checked on the
// CPU before any GPU time is spent.
#include <cstring>
#include <vector>

#include "../../beamform_amd/csrc/fft1024.hpp"
#include "../../beamform_amd/csrc/fft1024_w64.hpp"
#include "../../beamform_amd/csrc/fft_small.hpp"
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

// N = 32 x NL (fft_small.hpp): G = 32 / NL transforms side by side through one 32 x 32 plane, as stft_small_kernel / istft_small_kernel run
// them.  in / out: [G][N] complex doubles.
template <int LOGNL>
static void fft_small_emul(const double *in, double *out, int dir) {
    constexpr int NL = 1 << LOGNL, G = 32 / NL, N = 32 * NL, PS = plane_stride<double>::value;
    std::vector<double> plane(32 * PS);
    static double re[32][32], im[32][32];
    auto tw = [&](int k1, int n2) {
        const double a = -2.0 * kPi * (k1 * n2) / (double)N;
        return cx<double>{std::cos(a), std::sin(a)};
    };
    auto transpose = [&](double (*v)[32], bool natural_rows) {  // register r of lane l -> row (natural_rows ? r : brev5(r)), column l; lane q reads row q
        for (int l = 0; l < 32; ++l)
            for (int r = 0; r < 32; ++r) plane[(natural_rows ? r : brev5(r)) * PS + l] = v[l][r];
        for (int q = 0; q < 32; ++q)
            for (int c = 0; c < 32; ++c) v[q][c] = plane[q * PS + c];
    };
    if (dir < 0) {
        for (int l = 0; l < 32; ++l) {  // lane (g, n2): register j <- x_g[NL j + n2]
            const int g = l / NL, n2 = l % NL;
            for (int j = 0; j < 32; ++j) {
                re[l][j] = in[2 * (g * N + NL * j + n2)];
                im[l][j] = in[2 * (g * N + NL * j + n2) + 1];
            }
            fft32_dif<double, -1>(re[l], im[l]);
            for (int i = 1; i < 32; ++i) {
                const cx<double> w = tw(brev5(i), n2);
                const double xr = re[l][i], xi = im[l][i];
                re[l][i] = xr * w.x - xi * w.y;
                im[l][i] = xr * w.y + xi * w.x;
            }
        }
        transpose(re, false);
        transpose(im, false);
        for (int k1 = 0; k1 < 32; ++k1) {
            fftn_dif_all<double, -1, LOGNL>(re[k1], im[k1]);
            for (int g = 0; g < G; ++g)
                for (int i = 0; i < NL; ++i) {
                    const int k = k1 + 32 * brevn(i, LOGNL);
                    out[2 * (g * N + k)] = re[k1][g * NL + i];
                    out[2 * (g * N + k) + 1] = im[k1][g * NL + i];
                }
        }
    } else {
        for (int k1 = 0; k1 < 32; ++k1) {  // lane k1: position g NL + i' <- Y_g[k1 + 32 brev(i')]
            for (int g = 0; g < G; ++g)
                for (int i = 0; i < NL; ++i) {
                    const int k = k1 + 32 * brevn(i, LOGNL);
                    re[k1][g * NL + i] = in[2 * (g * N + k)];
                    im[k1][g * NL + i] = in[2 * (g * N + k) + 1];
                }
            fftn_dit_all<double, +1, LOGNL>(re[k1], im[k1]);
            for (int r = 0; r < 32; ++r) {
                const cx<double> w = tw(k1, r % NL);  // conj applied
                const double xr = re[k1][r], xi = im[k1][r];
                re[k1][r] = xr * w.x + xi * w.y;
                im[k1][r] = xi * w.x - xr * w.y;
            }
        }
        transpose(re, true);
        transpose(im, true);
        for (int l = 0; l < 32; ++l) {
            const int g = l / NL, n2 = l % NL;
            fft32_dif<double, +1>(re[l], im[l]);
            for (int i = 0; i < 32; ++i) {
                out[2 * (g * N + NL * brev5(i) + n2)] = re[l][i];
                out[2 * (g * N + NL * brev5(i) + n2) + 1] = im[l][i];
            }
        }
    }
}
extern "C" int emul_fft_small(int n, const double *in, double *out, int dir) {
    if (n == 512) fft_small_emul<4>(in, out, dir);
    else if (n == 256) fft_small_emul<3>(in, out, dir);
    else if (n == 128) fft_small_emul<2>(in, out, dir);
    else return -1;
    return 0;
}

// N = 2048 around FFT-1024 (istft_split_kernel, stft_bins_split_kernel).  Forward: in = 2048 complex samples, E / O = FFT-1024 of the even / odd ones,
// X[k] = E[k] + W^k O[k], X[k + 1024] = E[k] - W^k O[k].  Backward (the spectrum of a REAL frame, 2048 complex bins in): A = Y[k] + Y[k + 1024],
// B = (Y[k] - Y[k + 1024]) conj(W^k), one backward FFT-1024 of A + i B returns sample 2 m in its real and 2 m + 1 in its imaginary part; out = 2048 reals.
extern "C" void emul_fft2048_split(const double *in, double *out, int dir) {
    std::vector<double> a(2048), b(2048), fa(2048), fb(2048);
    if (dir < 0) {
        for (int m = 0; m < 1024; ++m) {
            a[2 * m] = in[2 * (2 * m)];
            a[2 * m + 1] = in[2 * (2 * m) + 1];
            b[2 * m] = in[2 * (2 * m + 1)];
            b[2 * m + 1] = in[2 * (2 * m + 1) + 1];
        }
        fft1024_emul<double>(a.data(), fa.data(), -1);
        fft1024_emul<double>(b.data(), fb.data(), -1);
        for (int k = 0; k < 1024; ++k) {
            const double ang = -2.0 * kPi * k / 2048.0, wx = std::cos(ang), wy = std::sin(ang);
            const double tr = fb[2 * k] * wx - fb[2 * k + 1] * wy, ti = fb[2 * k] * wy + fb[2 * k + 1] * wx;
            out[2 * k] = fa[2 * k] + tr;
            out[2 * k + 1] = fa[2 * k + 1] + ti;
            out[2 * (k + 1024)] = fa[2 * k] - tr;
            out[2 * (k + 1024) + 1] = fa[2 * k + 1] - ti;
        }
    } else {
        for (int k = 0; k < 1024; ++k) {
            const double ang = -2.0 * kPi * k / 2048.0, wx = std::cos(ang), wy = std::sin(ang);
            const double y0r = in[2 * k], y0i = in[2 * k + 1], y1r = in[2 * (k + 1024)], y1i = in[2 * (k + 1024) + 1];
            const double dr = y0r - y1r, di = y0i - y1i;
            const double br = dr * wx + di * wy, bi = di * wx - dr * wy;
            a[2 * k] = (y0r + y1r) - bi;
            a[2 * k + 1] = (y0i + y1i) + br;
        }
        fft1024_emul<double>(a.data(), fa.data(), +1);
        for (int m = 0; m < 1024; ++m) {
            out[2 * m] = fa[2 * m];
            out[2 * m + 1] = fa[2 * m + 1];
        }
    }
}

// N = 2048 = 32 (registers) x 64 (lanes) as das_fused_wave2048_kernel runs it: first pass over the registers, twiddle, the two-half plane
// transpose (row = register position, 64 columns; lane l of half h reads row l, columns 32 h ..), one radix-2 stage between the halves, a
// 32-point pass per lane.  in / out: 2048 complex doubles.
template <typename T>
static void fft2048_wave_emul(const double *in, double *out, int dir) {
    static T re[64][32], im[64][32], pr[32][64], pi_[32][64];
    auto tw2048 = [&](long m) {
        const double a = -2.0 * kPi * (double)(m % 2048) / 2048.0;
        return cx<T>{(T)std::cos(a), (T)std::sin(a)};
    };
    auto transpose = [&](bool natural_rows) {
        for (int ln = 0; ln < 64; ++ln)
            for (int r = 0; r < 32; ++r) {
                pr[natural_rows ? r : brev5(r)][ln] = re[ln][r];
                pi_[natural_rows ? r : brev5(r)][ln] = im[ln][r];
            }
        for (int ln = 0; ln < 64; ++ln) {
            const int l = ln & 31, h = ln >> 5;
            for (int c = 0; c < 32; ++c) {
                re[ln][c] = pr[l][32 * h + c];
                im[ln][c] = pi_[l][32 * h + c];
            }
        }
    };
    if (dir < 0) {
        for (int ln = 0; ln < 64; ++ln) {
            for (int j = 0; j < 32; ++j) {
                re[ln][j] = (T)in[2 * (64 * j + ln)];
                im[ln][j] = (T)in[2 * (64 * j + ln) + 1];
            }
            fft32_dif<T, -1>(re[ln], im[ln]);
            for (int i = 1; i < 32; ++i) {
                const cx<T> w = tw2048((long)brev5(i) * ln);
                const T xr = re[ln][i], xi = im[ln][i];
                re[ln][i] = xr * w.x - xi * w.y;
                im[ln][i] = xr * w.y + xi * w.x;
            }
        }
        transpose(false);
        for (int l = 0; l < 32; ++l)
            for (int c = 0; c < 32; ++c) {  // radix-2 DIF stage between lane l of half 0 (a) and of half 1 (b)
                const T ar = re[l][c], ai = im[l][c], br = re[l + 32][c], bi = im[l + 32][c];
                const cx<T> w = tw2048(32L * c);  // W64^c
                const T dr = ar - br, di = ai - bi;
                re[l][c] = ar + br;
                im[l][c] = ai + bi;
                re[l + 32][c] = dr * w.x - di * w.y;
                im[l + 32][c] = dr * w.y + di * w.x;
            }
        for (int ln = 0; ln < 64; ++ln) {
            fft32_dif<T, -1>(re[ln], im[ln]);
            for (int i = 0; i < 32; ++i) {
                const int k = ln + 64 * brev5(i);
                out[2 * k] = re[ln][i];
                out[2 * k + 1] = im[ln][i];
            }
        }
    } else {
        for (int ln = 0; ln < 64; ++ln) {
            for (int i = 0; i < 32; ++i) {
                const int k = ln + 64 * brev5(i);
                re[ln][i] = (T)in[2 * k];
                im[ln][i] = (T)in[2 * k + 1];
            }
            fft32_dit<T, +1>(re[ln], im[ln]);
        }
        for (int l = 0; l < 32; ++l)
            for (int c = 0; c < 32; ++c) {  // radix-2 DIT stage: n2 = c (half 0) <- a + conj(W64^c) b, n2 = c + 32 (half 1) <- a - conj(W64^c) b
                const T ar = re[l][c], ai = im[l][c], br = re[l + 32][c], bi = im[l + 32][c];
                const cx<T> w = tw2048(32L * c);
                const T tr = br * w.x + bi * w.y, ti = bi * w.x - br * w.y;
                re[l][c] = ar + tr;
                im[l][c] = ai + ti;
                re[l + 32][c] = ar - tr;
                im[l + 32][c] = ai - ti;
            }
        transpose(true);
        for (int ln = 0; ln < 64; ++ln) {
            for (int c = 1; c < 32; ++c) {
                const cx<T> w = tw2048((long)c * ln);  // conj applied
                const T xr = re[ln][c], xi = im[ln][c];
                re[ln][c] = xr * w.x + xi * w.y;
                im[ln][c] = xi * w.x - xr * w.y;
            }
            fft32_dif<T, +1>(re[ln], im[ln]);
            for (int i = 0; i < 32; ++i) {
                const int n = 64 * brev5(i) + ln;
                out[2 * n] = re[ln][i];
                out[2 * n + 1] = im[ln][i];
            }
        }
    }
}
extern "C" void emul_fft2048_wave(const double *in, double *out, int dir, int use_float) {
    use_float ? fft2048_wave_emul<float>(in, out, dir) : fft2048_wave_emul<double>(in, out, dir);
}

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

// das_f64_pair_kernel's formulation in double with the kernel's own gain table (geometry.hpp das_mic_gains_w64_f64: bins 0 .. 512 in rows
// of 65) and its addressing: frames t, t + 1 of one microphone per transform, register 4 g + k3 of lane l = bin l + 64 g + 256 k3; k3 < 2
// reads row (g, k3) column l, k3 >= 2 row (3 - g, 3 - k3) column 64 - l, conjugated.  Real part of the backward transform = frame t,
// imaginary part = frame t + 1.  x planar [M][F*512], y [F*512].
void emul_das_pair_f64(int M, double sr, const double *mx, const double *my, double theta, const float *x, long F, float *y) {
    const int N = 1024, H = 512;
    ArrayGeometry g;
    g.set(mx, my, M);
    std::vector<double> freqs = frequency_vector(N, sr);
    SteeringSet st;
    st.allocate(N, M, 1);
    st.update_column(g, freqs, 0, theta, true);
    const std::vector<f64x2> T = das_mic_gains_w64_f64(st, 8);
    const std::vector<double> h = sqrt_hann(N);
    std::vector<float> tail(H, 0.f);
    std::vector<double> z(2 * N), Z(2 * N), U(2 * N), u(2 * N);
    auto sample = [&](int m, long s) -> double { return s < 0 ? 0.0 : (double)x[(size_t)m * F * H + s]; };
    for (long t = 0; t < F; t += 2) {
        const bool two = t + 1 < F;
        std::fill(U.begin(), U.end(), 0.0);
        for (int m = 0; m < M; ++m) {
            for (int n = 0; n < N; ++n) {
                z[2 * n] = h[n] * sample(m, (t - 1) * (long)H + n);
                z[2 * n + 1] = two ? h[n] * sample(m, t * (long)H + n) : 0.0;
            }
            fft1024_emul<double>(z.data(), Z.data(), -1);
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 16; ++r) {
                    const int k = w64_bin(lane, r), gg = r >> 2, k3 = r & 3;
                    f64x2 G;
                    if (k3 < 2) {
                        G = T[((size_t)m * kDasMicGainRows + 2 * gg + k3) * kDasMicGainRow + lane];
                    } else {
                        G = T[((size_t)m * kDasMicGainRows + 2 * (3 - gg) + (3 - k3)) * kDasMicGainRow + (64 - lane)];
                        G.y = -G.y;
                    }
                    U[2 * k] += G.x * Z[2 * k] - G.y * Z[2 * k + 1];
                    U[2 * k + 1] += G.x * Z[2 * k + 1] + G.y * Z[2 * k];
                }
        }
        fft1024_emul<double>(U.data(), u.data(), +1);  // unnormalised: 1/N is inside the gains
        for (int f = 0; f < (two ? 2 : 1); ++f)
            for (int n = 0; n < H; ++n) {
                const float o1 = (float)((double)(float)u[2 * n + f] * h[n]);
                const float o2 = (float)((double)(float)u[2 * (n + H) + f] * h[n + H]);
                y[(size_t)(t + f) * H + n] = tail[n] + o1;
                tail[n] = o2;
            }
    }
}

// das_fused_small_kernel's formulation (periods below 512 frames): R = 1024 / N consecutive frames interleaved sample by sample into one
// 1024-point sequence, the N-point pair gains repeated R times (das_pair_gains_interleaved, fp32 table in the 32 x 32 order), one
// 1024-point transform pair per microphone pair and GROUP of frames.  Arithmetic in double (the kernel: float).  x planar [M][F*H].
void emul_das_small(int M, int H, double sr, const double *mx, const double *my, double theta, const float *x, long F, float *y) {
    const int N = 2 * H, R = 1024 / N;
    ArrayGeometry g;
    g.set(mx, my, M);
    std::vector<double> freqs = frequency_vector(N, sr);
    SteeringSet st;
    st.allocate(N, M, 1);
    st.update_column(g, freqs, 0, theta, true);
    const int NP = (M + 1) / 2;
    const std::vector<f32x2> D = das_pair_gains_interleaved(st, NP);
    const std::vector<double> hd = sqrt_hann(N);
    std::vector<float> h(N);
    for (int i = 0; i < N; ++i) h[i] = (float)hd[i];
    std::vector<std::vector<float>> tails(1, std::vector<float>(H, 0.f));
    std::vector<float> tail(H, 0.f);
    std::vector<double> z(2 * 1024), Z(2 * 1024), S(2 * 1024), u(2 * 1024);
    auto sample = [&](int m, long s) -> double { return (m >= M || s < 0 || s >= F * (long)H) ? 0.0 : (double)x[(size_t)m * F * H + s]; };
    for (long tg = 0; tg < F; tg += R) {
        std::fill(S.begin(), S.end(), 0.0);
        for (int p = 0; p < NP; ++p) {
            for (int n = 0; n < 1024; ++n) {  // interleaved index n = R m + i: sample m of frame tg + i
                const int i = n % R, m = n / R;
                const long s0 = (tg + i - 1) * (long)H + m;
                const bool ok = tg + i < F;
                z[2 * n] = ok ? (double)h[m] * sample(2 * p, s0) : 0.0;
                z[2 * n + 1] = ok ? (double)h[m] * sample(2 * p + 1, s0) : 0.0;
            }
            fft1024_emul<double>(z.data(), Z.data(), -1);
            for (int i = 0; i < 32; ++i)
                for (int l = 0; l < 32; ++l) {
                    const int k = l + 32 * brev5(i);
                    const f32x2 d = D[((size_t)p * 32 + i) * 32 + l];
                    S[2 * k] += (double)d.x * Z[2 * k] - (double)d.y * Z[2 * k + 1];
                    S[2 * k + 1] += (double)d.x * Z[2 * k + 1] + (double)d.y * Z[2 * k];
                }
        }
        fft1024_emul<double>(S.data(), u.data(), +1);
        for (int i = 0; i < R && tg + i < F; ++i)
            for (int m = 0; m < H; ++m) {
                const float o1 = (float)u[2 * (R * m + i)] * h[m];
                const float o2 = (float)u[2 * (R * (m + H) + i)] * h[m + H];
                y[(size_t)(tg + i) * H + m] = tail[m] + o1;
                tail[m] = o2;
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

// ---- das_f64_pair_kernel's chunk plan (csrc/das_f64_plan.hpp): the table as das_f64_sched_kernel writes it --------------------------------
#include "../../beamform_amd/csrc/das_f64_plan.hpp"
// fills stream / t0 / n (capacity `cap` rows) and returns the number of chunks (negative: more than `cap`); *grid = persistent blocks
extern "C" int emul_das_plan(long n_frames, int n_streams, int n_cus, const char *env, int cap, int *stream, long *t0, long *n, int *grid) {
    const bf::DasSchedPlan p = bf::das_f64_plan(n_frames, n_streams, n_cus, (env && *env) ? env : nullptr);
    *grid = p.grid;
    if (p.n_chunks > cap) return -p.n_chunks;
    for (int k = 0; k < p.n_chunks; ++k) bf::das_f64_chunk(p, n_frames, n_streams, k, &stream[k], &t0[k], &n[k]);
    return p.n_chunks;
}
