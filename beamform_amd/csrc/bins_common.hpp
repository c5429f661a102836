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
__device__ __forceinline__ double fast_sqrt(double x) { return x == 0.0 ? 0.0 : x * fast_rsqrt(x); }
__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = fma(y, fma(-x, y, 1.0), y);
    y = fma(y, fma(-x, y, 1.0), y);
    return y;
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

// problem index -> FFT bin whose packed spectrum is read, and whether X must be conjugated
__device__ __forceinline__ int q_src_bin(int q) { return q == kQX ? kN / 2 - 1 : q; }
__device__ __forceinline__ int q_bin(int q) { return q; }

// X_m for problem q out of the packed pair spectra of one frame (Zf = [NP][1024]).
template <int MP>
__device__ __forceinline__ void load_X(const f64x2 *Zf, int q, int M, cd (&X)[MP]) {
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
