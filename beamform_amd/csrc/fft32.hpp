// fft32.hpp -- in-register 32-point complex FFT building block.
//
// The 1024-point transforms of the STFT path (das.cpp:127-128 plans) are done as
// 32 x 32: every lane of a 32-lane half-wavefront runs one 32-point FFT entirely
// in VGPRs, a twiddle multiply, one LDS transpose, and a second 32-point FFT
// (see fft1024.hpp).  This header is plain C++ so the same code is compiled by
// g++ for the host-side emulation tests and by hipcc for gfx950.
//
// All loops have compile-time trip counts and are fully unrolled so that the
// re[]/im[] arrays live in registers (no runtime indexing -> no scratch).
#pragma once

#if defined(__HIPCC__)
#define BF_HD __host__ __device__ __forceinline__
#else
#define BF_HD inline __attribute__((always_inline))
#endif

// Compiler scheduling fence (no instruction): keeps the scheduler from hoisting a whole table's worth of
// LDS/global loads above the arithmetic that frees their registers.
#if defined(__HIP_DEVICE_COMPILE__)
#define BF_SCHED_FENCE() ((void)0)  // disabled: see DESIGN.md (fences cost more in exposed LDS latency than they saved)
#else
#define BF_SCHED_FENCE() ((void)0)
#endif

namespace bf {

constexpr int brev5(int i) {
    return ((i & 1) << 4) | ((i & 2) << 2) | (i & 4) | ((i & 8) >> 2) | ((i & 16) >> 4);
}

// cos(2*pi*k/32), sin(2*pi*k/32) for k = 0..15 (only the first half-turn is needed)
constexpr double cos32(int k) {
    constexpr double t[16] = {1.0,
                              0.98078528040323044912618223613424,
                              0.92387953251128675612818318939679,
                              0.83146961230254523707878837761791,
                              0.70710678118654752440084436210485,
                              0.55557023301960222474283081394853,
                              0.38268343236508977172845998403040,
                              0.19509032201612826784828486847702,
                              0.0,
                              -0.19509032201612826784828486847702,
                              -0.38268343236508977172845998403040,
                              -0.55557023301960222474283081394853,
                              -0.70710678118654752440084436210485,
                              -0.83146961230254523707878837761791,
                              -0.92387953251128675612818318939679,
                              -0.98078528040323044912618223613424};
    return t[k];
}
constexpr double sin32(int k) {
    constexpr double t[16] = {0.0,
                              0.19509032201612826784828486847702,
                              0.38268343236508977172845998403040,
                              0.55557023301960222474283081394853,
                              0.70710678118654752440084436210485,
                              0.83146961230254523707878837761791,
                              0.92387953251128675612818318939679,
                              0.98078528040323044912618223613424,
                              1.0,
                              0.98078528040323044912618223613424,
                              0.92387953251128675612818318939679,
                              0.83146961230254523707878837761791,
                              0.70710678118654752440084436210485,
                              0.55557023301960222474283081394853,
                              0.38268343236508977172845998403040,
                              0.19509032201612826784828486847702};
    return t[k];
}

BF_HD float bf_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
BF_HD double bf_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// Decimation-in-time butterfly with the twiddle fused into the additions (Linzer-Feig form):
//   (a, b) -> (a + w b, a - w b),  w = exp(DIR * 2*pi*i * k / 32),  k in [0,16) known after unrolling.
// With rho = Im(w)/Re(w) (or its reciprocal when |Re w| < |Im w|) the general case is 6 FMAs instead
// of a 4-op complex multiply plus 4 additions; trivial twiddles stay pure additions.
// DIR = -1: forward (FFTW_FORWARD sign), +1: backward.
template <typename T, int DIR>
BF_HD void bfly_dit(int k, T &ar, T &ai, T &br, T &bi) {
    const T xr = br, xi = bi, pr = ar, pi = ai;
    if (k == 0) {
        ar = pr + xr;
        ai = pi + xi;
        br = pr - xr;
        bi = pi - xi;
    } else if (k == 8) {  // w = DIR * i
        if (DIR < 0) {
            ar = pr + xi;
            ai = pi - xr;
            br = pr - xi;
            bi = pi + xr;
        } else {
            ar = pr - xi;
            ai = pi + xr;
            br = pr + xi;
            bi = pi - xr;
        }
    } else if (k <= 4 || k >= 12) {  // |Re w| >= |Im w|
        const T c = (T)cos32(k);
        const T rho = (T)(DIR * sin32(k) / cos32(k));
        const T u = bf_fma(-rho, xi, xr);
        const T v = bf_fma(rho, xr, xi);
        ar = bf_fma(c, u, pr);
        ai = bf_fma(c, v, pi);
        br = bf_fma(-c, u, pr);
        bi = bf_fma(-c, v, pi);
    } else {  // |Re w| < |Im w|: factor Im w instead
        const T sn = (T)(DIR * sin32(k));
        const T rho = (T)(cos32(k) / (DIR * sin32(k)));
        const T u = bf_fma(rho, xr, -xi);
        const T v = bf_fma(rho, xi, xr);
        ar = bf_fma(sn, u, pr);
        ai = bf_fma(sn, v, pi);
        br = bf_fma(-sn, u, pr);
        bi = bf_fma(-sn, v, pi);
    }
}

// In-place radix-2 DIT on logical positions; PERM maps a logical position to the physical
// register.  Logical input L[i] = x[brev5(i)], logical output L[k] = X[k].
//   PERM = identity: physical in = bit-reversed, out = natural          (fft32_dit)
//   PERM = brev5   : physical in = natural,      out = X[brev5(i)] at i (fft32_dif's contract)
template <typename T, int DIR, bool PERM_BREV>
BF_HD void fft32_core(T (&re)[32], T (&im)[32]) {
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int half = 1 << s;
        const int tstep = 16 >> s;
#pragma unroll
        for (int blk = 0; blk < 32; blk += 2 * half) {
#pragma unroll
            for (int j = 0; j < half; ++j) {
                const int la = blk + j, lb = la + half;
                const int a = PERM_BREV ? brev5(la) : la;
                const int b = PERM_BREV ? brev5(lb) : lb;
                bfly_dit<T, DIR>(j * tstep, re[a], im[a], re[b], im[b]);
            }
        }
    }
}

// fft32_dif (= fft32_core<T, DIR, true>) cut into pieces that touch disjoint register sets, so that a caller can interleave other work
// (loads into registers that have just been released, gains on outputs that have just been finished) between them.  In that core
// stage s pairs physical positions a and a + (16 >> s) (a's bit 4 - s clear) with twiddle (brev5(a) & ((1 << s) - 1)) * (16 >> s):
//   stages 0..2 stay inside the four sets { g, g + 4, ..., g + 28 }   -> fft32_dif_head(g)
//   stages 3..4 stay inside the eight sets { 4 m, ..., 4 m + 3 }      -> fft32_dif_tail(m)
// head(0..3) followed by tail(0..7) is fft32_dif: the same butterflies with the same operands, only their order differs.
template <typename T, int DIR>
BF_HD void fft32_dif_piece(T (&re)[32], T (&im)[32], int s_lo, int s_hi, int first, int count, int stride) {
#pragma unroll
    for (int s = s_lo; s <= s_hi; ++s) {
#pragma unroll
        for (int n = 0; n < count; ++n) {
            const int a = first + n * stride;
            if (((a >> (4 - s)) & 1) == 0) {
                const int b = a + (16 >> s);
                bfly_dit<T, DIR>((brev5(a) & ((1 << s) - 1)) * (16 >> s), re[a], im[a], re[b], im[b]);
            }
        }
    }
}
template <typename T, int DIR>
BF_HD void fft32_dif_head(T (&re)[32], T (&im)[32], int g) { fft32_dif_piece<T, DIR>(re, im, 0, 2, g, 8, 4); }
template <typename T, int DIR>
BF_HD void fft32_dif_tail(T (&re)[32], T (&im)[32], int m) { fft32_dif_piece<T, DIR>(re, im, 3, 4, 4 * m, 4, 1); }

// natural-order input, output X[brev5(i)] at position i
template <typename T, int DIR>
BF_HD void fft32_dif(T (&re)[32], T (&im)[32]) {
    fft32_core<T, DIR, true>(re, im);
}

// input x[brev5(i)] at position i, natural-order output
template <typename T, int DIR>
BF_HD void fft32_dit(T (&re)[32], T (&im)[32]) {
    fft32_core<T, DIR, false>(re, im);
}

}  // namespace bf
