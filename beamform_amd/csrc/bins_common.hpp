// bins_common.hpp -- device helpers shared by the fp64 bin-pipeline translation units (complex doubles, packed-
// spectrum unpacking, problem indexing).  Internal linkage: every .hip that includes it gets its own copy.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "fft1024.hpp"
#include "pipeline_kernels.hpp"

#ifndef BF_NFFT
#error "kernel translation units are compiled per FFT size: -DBF_NFFT=512|1024|2048"
#endif

namespace bf {
namespace BF_NTAG {
namespace {

constexpr int kPSd = plane_stride<double>::value;  // 34

// ---- tiny complex helpers (double) ---------------------------------------------------
struct cd {
    double x, y;
};
__device__ __forceinline__ cd mk(double x, double y) { return cd{x, y}; }
__device__ __forceinline__ cd operator+(cd a, cd b) { return cd{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cd operator-(cd a, cd b) { return cd{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cd operator*(cd a, cd b) { return cd{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cd operator*(cd a, double s) { return cd{a.x * s, a.y * s}; }
__device__ __forceinline__ cd conj(cd a) { return cd{a.x, -a.y}; }
__device__ __forceinline__ double norm2(cd a) { return a.x * a.x + a.y * a.y; }
// |a|: spectra of [-1,1] audio are far from the double range limits, so no hypot-style rescaling is needed
__device__ __forceinline__ double cabs(cd a) { return sqrt(a.x * a.x + a.y * a.y); }
// mag * (cos, sin)(arg z) without trigonometry: mag * z/|z|; arg(0) = 0 as std::arg does
__device__ __forceinline__ cd with_phase_of(double mag, cd z) {
    const double r = cabs(z);
    if (r == 0.0) return cd{mag, 0.0};
    return cd{mag * (z.x / r), mag * (z.y / r)};
}
__device__ __forceinline__ cd cdiv(cd a, cd b) {
    // Smith's algorithm, as libstdc++/libgcc __divdc3 do for finite operands
    if (fabs(b.x) >= fabs(b.y)) {
        const double r = b.y / b.x, d = b.x + b.y * r;
        return cd{(a.x + a.y * r) / d, (a.y - a.x * r) / d};
    }
    const double r = b.x / b.y, d = b.x * r + b.y;
    return cd{(a.x * r + a.y) / d, (a.y * r - a.x) / d};
}
// 1/sqrt(x), sqrt(x) and 1/x for x well inside the double range (sums of squared spectra of [-1,1] audio): the hardware estimate
// (v_rsq_f64 / v_rcp_f64, ~2^-26 relative) and two Newton steps, without the subnormal rescaling and class tests of the library
// versions (x = 0 gives inf / NaN like they do: a zero covariance still yields the reference's NaN frame).  Relative error < 1e-15.
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = fma(y, fma(-hx * y, y, 0.5), y);
    y = fma(y, fma(-hx * y, y, 0.5), y);
    return y;
}
// one Newton step: 4.3e-15 relative (v_rsq_f64 alone: 5.2e-8; two steps: 2.6e-16 -- tools/ubench/rsq_test.hip, 4 M values over 24 decades).
// For the Cholesky pivots only: their error reaches y_fft times the condition number, next to the 7e-12 of the packed spectra and
// a 1e-5 tolerance; the step is three dependent fp64 operations on the critical path of every elimination step.
__device__ __forceinline__ double fast_rsqrt1(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    return fma(y, fma(-0.5 * x * y, y, 0.5), y);
}
__device__ __forceinline__ double fast_sqrt(double x) { return x == 0.0 ? 0.0 : x * fast_rsqrt(x); }
__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = fma(y, fma(-x, y, 1.0), y);
    y = fma(y, fma(-x, y, 1.0), y);
    return y;
}
// atan2 for the phase masks (phase.cpp:101, phasempf.cpp:221: arg of conj(w) X), about half the instructions of the library
// routine: one division instead of two, no special-case ladder.  With mx = max(|x|,|y|), mn = min(|x|,|y|) the octant angle is
//   atan(mn / mx) = atan(c) + atan(z),  z = (mn - c mx) / (mx + c mn),  c in {0, 1/2, 1} chosen so that |z| <= 1/4,
// atan(z) = z + z^3 Q(z^2) with a degree-8 Q (least-squares fit in long double at 600 Chebyshev nodes, max relative error of the
// evaluated polynomial 1.7e-16); atan(c), pi/4, pi/2 and pi enter as hi + lo pairs.  Measured on the device over 1.7e7 points spanning 18 decades
// (tools/ubench/atan2_test.hip): within 1.82 ulp of the exact value (the library routine: 1.62 ulp).  Signed zeros follow the IEEE rules (a silent channel gives arg(0) as std::arg does);
// magnitudes outside 1e-280 .. 1e280 are rescaled by an exact power of two first; NaN in, NaN out.
__device__ __forceinline__ double atan2_fast(double y, double x) {
    const double ax = fabs(x), ay = fabs(y);
    double mx = fmax(ax, ay), mn = fmin(ax, ay);
    // keep the quotient's reciprocal estimate in range: exact power-of-two rescaling of both magnitudes (no branch, no library call)
    const double sc = mx < 1e-280 ? 0x1p600 : (mx > 1e280 ? 0x1p-600 : 1.0);
    mx *= sc;
    mn *= sc;
    const bool b1 = mn > 0.25 * mx, b2 = mn > 0.75 * mx;
    const double c = b2 ? 1.0 : (b1 ? 0.5 : 0.0);
    const double chi = b2 ? 0.7853981633974483 : (b1 ? 0.4636476090008061 : 0.0);
    const double clo = b2 ? 3.061616997868383e-17 : (b1 ? 2.268693045925918e-17 : 0.0);
    const double num = fma(-c, mx, mn), den = fma(c, mn, mx);
    const double r = fast_rcp(den);
    double z = num * r;
    z = fma(fma(-den, z, num), r, z);  // one correction step on the quotient
    const double s = z * z;
    double q = -0.040008015363946665;
    q = fma(q, s, 0.05724708565692529);
    q = fma(q, s, -0.066554543402443);
    q = fma(q, s, 0.07691819902065834);
    q = fma(q, s, -0.09090895876111414);
    q = fma(q, s, 0.1111111089272362);
    q = fma(q, s, -0.14285714283649867);
    q = fma(q, s, 0.19999999999990273);
    q = fma(q, s, -0.33333333333333315);
    double a = fma(z * s, q, z);
    a = chi + (a + clo);
    if (mx == 0.0) a = 0.0;  // atan2(+-0, +-0): 0 in the right half plane, pi in the left (decided by the sign of x below)
    if (ay > ax) a = 1.5707963267948966 - (a - 6.123233995736766e-17);
    if (__builtin_signbit(x)) a = 3.141592653589793 - (a - 1.2246467991473532e-16);
    return copysign(a, y);
}
// x / d with a reciprocal that was formed once (rd ~ 1/d) and one correction step: the quotient is within half an ulp of the
// correctly rounded one for finite x and d (exact for d = 0: x * inf, 0 * inf = NaN as the division gives).
__device__ __forceinline__ double div_rcp(double x, double d, double rd) {
    const double q = x * rd;
    const double r = fma(-d, q, x);
    return (d == 0.0 || r != r) ? q : fma(r, rd, q);
}
// |X_m| of N spectra, stage by stage across the N values (same reason as atan2_fast_n below); each value = fast_sqrt(norm2(X_m)).
template <int N>
__device__ __forceinline__ void cabs_n(const cd (&X)[N], double (&out)[N]) {
    double n2[N], y[N];
#pragma unroll
    for (int i = 0; i < N; ++i) n2[i] = fma(X[i].y, X[i].y, X[i].x * X[i].x);
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = __builtin_amdgcn_rsq(n2[i]);
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < N; ++i) y[i] = fma(y[i], fma(-0.5 * n2[i] * y[i], y[i], 0.5), y[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = n2[i] == 0.0 ? 0.0 : n2[i] * y[i];
}
// atan2_fast of N independent arguments, written stage by stage across the N values: a per-bin kernel that runs one wavefront
// per SIMD has no second wavefront to cover the ~10-cycle latency of dependent fp64 operations, and one atan2 is a single chain
// of ~45 of them; interleaving the N chains keeps the pipe busy.  Element for element the same operations as atan2_fast.
template <int N>
__device__ __forceinline__ void atan2_fast_n(const double (&y)[N], const double (&x)[N], double (&out)[N]) {
    double mx[N], mn[N], chi[N], clo[N], num[N], den[N], r[N], z[N], s[N], q[N];
    bool swap_[N], zero_[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double ax = fabs(x[i]), ay = fabs(y[i]);
        swap_[i] = ay > ax;
        double a = fmax(ax, ay), b = fmin(ax, ay);
        const double sc = a < 1e-280 ? 0x1p600 : (a > 1e280 ? 0x1p-600 : 1.0);
        a *= sc;
        b *= sc;
        zero_[i] = a == 0.0;
        mx[i] = a;
        mn[i] = b;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const bool b1 = mn[i] > 0.25 * mx[i], b2 = mn[i] > 0.75 * mx[i];
        const double c = b2 ? 1.0 : (b1 ? 0.5 : 0.0);
        chi[i] = b2 ? 0.7853981633974483 : (b1 ? 0.4636476090008061 : 0.0);
        clo[i] = b2 ? 3.061616997868383e-17 : (b1 ? 2.268693045925918e-17 : 0.0);
        num[i] = fma(-c, mx[i], mn[i]);
        den[i] = fma(c, mn[i], mx[i]);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = __builtin_amdgcn_rcp(den[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = fma(r[i], fma(-den[i], r[i], 1.0), r[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = fma(r[i], fma(-den[i], r[i], 1.0), r[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) z[i] = num[i] * r[i];
#pragma unroll
    for (int i = 0; i < N; ++i) z[i] = fma(fma(-den[i], z[i], num[i]), r[i], z[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        s[i] = z[i] * z[i];
        q[i] = -0.040008015363946665;
    }
    constexpr double kC[8] = {0.05724708565692529, -0.066554543402443, 0.07691819902065834, -0.09090895876111414,
                              0.1111111089272362, -0.14285714283649867, 0.19999999999990273, -0.33333333333333315};
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int i = 0; i < N; ++i) q[i] = fma(q[i], s[i], kC[k]);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double a = fma(z[i] * s[i], q[i], z[i]);
        a = chi[i] + (a + clo[i]);
        if (zero_[i]) a = 0.0;
        if (swap_[i]) a = 1.5707963267948966 - (a - 6.123233995736766e-17);
        if (__builtin_signbit(x[i])) a = 3.141592653589793 - (a - 1.2246467991473532e-16);
        out[i] = copysign(a, y[i]);
    }
}
// fp32 atan2 of N independent arguments for the phase-mask PRE-decision (phase_is_close below): octant reduction as atan2_fast
// (c in {0, 1/2, 1}, |z| <= 1/4), atan(z) = z (1 - s/3 + s^2/5 - s^3/7), s = z^2 (next term z^9/9 < 5e-7), hardware reciprocal.
// Absolute error < 2e-6 rad over the float range; both components zero gives 0 like atan2f.
template <int N>
__device__ __forceinline__ void atan2f_fast_n(const float (&y)[N], const float (&x)[N], float (&out)[N]) {
    float mx[N], mn[N], ch[N], z[N];
    bool swap_[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float ax = __builtin_fabsf(x[i]), ay = __builtin_fabsf(y[i]);
        swap_[i] = ay > ax;
        mx[i] = __builtin_fmaxf(ax, ay);
        mn[i] = __builtin_fminf(ax, ay);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const bool b1 = mn[i] > 0.25f * mx[i], b2 = mn[i] > 0.75f * mx[i];
        const float c = b2 ? 1.0f : (b1 ? 0.5f : 0.0f);
        ch[i] = b2 ? 0.78539816f : (b1 ? 0.46364761f : 0.0f);
        const float num = __builtin_fmaf(-c, mx[i], mn[i]), den = __builtin_fmaf(c, mn[i], mx[i]);
        z[i] = num * __builtin_amdgcn_rcpf(den);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float s = z[i] * z[i];
        float q = -0.14285714f;
        q = __builtin_fmaf(q, s, 0.2f);
        q = __builtin_fmaf(q, s, -0.33333333f);
        float a = ch[i] + __builtin_fmaf(z[i] * s, q, z[i]);
        if (mx[i] == 0.0f) a = 0.0f;
        if (swap_[i]) a = 1.57079633f - a;
        if (__builtin_signbit(x[i])) a = 3.14159265f - a;
        out[i] = __builtin_copysignf(a, y[i]);
    }
}
// 1 / g for a pivot of the small Hermitian systems (|g|^2 well inside the double range: Gram entries of whitened spectra):
// conj(g) / |g|^2 through the reciprocal estimate + two Newton steps, ~13 instructions against ~100 for Smith's algorithm with its
// three library divisions.  Relative error ~2e-16.
__device__ __forceinline__ cd crcp(cd g) {
    const double r = fast_rcp(fma(g.x, g.x, g.y * g.y));
    return cd{g.x * r, -g.y * r};
}
// acc - a * conj(b) and acc + a * conj(b), four FMAs each
__device__ __forceinline__ cd cfms_conj(cd acc, cd a, cd b) {
    return cd{fma(-a.y, b.y, fma(-a.x, b.x, acc.x)), fma(a.x, b.y, fma(-a.y, b.x, acc.y))};
}
__device__ __forceinline__ cd cfma_conj(cd acc, cd a, cd b) {
    return cd{fma(a.y, b.y, fma(a.x, b.x, acc.x)), fma(-a.x, b.y, fma(a.y, b.x, acc.y))};
}
// acc - a * b
__device__ __forceinline__ cd cfms(cd acc, cd a, cd b) {
    return cd{fma(a.y, b.y, fma(-a.x, b.x, acc.x)), fma(-a.y, b.x, fma(-a.x, b.y, acc.y))};
}
__device__ __forceinline__ cd ld(const f64x2 *p) {
    const f64x2 v = *p;
    return cd{v.x, v.y};
}
// z48 <-> complex double (pipeline_kernels.hpp): round to nearest on the 36-bit mantissa (a 64-bit integer add that may carry
// into the exponent, as rounding does), decode = the stored bits with a zero tail
__device__ __forceinline__ z48 enc48(double re, double im) {
    const unsigned long long ur = (unsigned long long)__double_as_longlong(re) + 0x8000ull;
    const unsigned long long ui = (unsigned long long)__double_as_longlong(im) + 0x8000ull;
    return z48{(((unsigned)ur) & 0xFFFF0000u) | (((unsigned)ui) >> 16), (unsigned)(ur >> 32), (unsigned)(ui >> 32)};
}
__device__ __forceinline__ cd dec48(z48 v) {
    return cd{__hiloint2double((int)v.re_hi, (int)(v.lo & 0xFFFF0000u)), __hiloint2double((int)v.im_hi, (int)(v.lo << 16))};
}
__device__ __forceinline__ cd ld(const z48 *p) { return dec48(*p); }
// y_fft of one problem into its row: f64x2 rows, or f32x2 rows in front of the fp32 backward transform (BinsArgs::yh32)
// (then only problem 0 and the in-band problems yh_lo..yh_hi exist: the others are zero by definition and nobody reads them)
__device__ __forceinline__ void st_y(const BinsArgs &a, long idx, int q, cd y) {
    if (a.yh32) {
        if (q == 0 || (q >= a.yh_lo && q <= a.yh_hi)) reinterpret_cast<f32x2 *>(a.Yh)[idx] = f32x2{(float)y.x, (float)y.y};
    } else {
        a.Yh[idx] = f64x2{y.x, y.y};
    }
}

// problem index -> FFT bin whose packed spectrum is read, and whether X must be conjugated
__device__ __forceinline__ int q_src_bin(int q) { return q == kQX ? kN / 2 - 1 : q; }
__device__ __forceinline__ int q_bin(int q) { return q; }
inline int q_bin_host(int q) { return q; }

// X_m for problem q out of the packed pair spectra of one frame (Zf = [NP][1024]).
template <int MP, typename ZT>
__device__ __forceinline__ void load_X(const ZT *Zf, int q, int M, cd (&X)[MP]) {
    const int k = q_src_bin(q);
    const int kn = (kN - k) & (kN - 1);
#pragma unroll
    for (int p = 0; p < MP / 2; ++p) {
        if (2 * p < M) {
            const cd z = ld(Zf + p * kN + k);
            const cd zc = conj(ld(Zf + p * kN + kn));
            cd xa = (z + zc) * 0.5;                 // (Z[k] + conj Z[N-k]) / 2
            const cd d = z - zc;                    // (Z[k] - conj Z[N-k]) / (2i) = -i/2 * d
            cd xb = cd{0.5 * d.y, -0.5 * d.x};
            if (q == kQX) {
                xa = conj(xa);
                xb = conj(xb);
            }
            X[2 * p] = xa;
            X[2 * p + 1] = xb;
        } else {
            X[2 * p] = cd{0, 0};
            X[2 * p + 1] = cd{0, 0};
        }
    }
}

}  // namespace
}  // namespace BF_NTAG
}  // namespace bf
