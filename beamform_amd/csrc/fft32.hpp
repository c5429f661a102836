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
#define BF_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
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

// (xr + i xi) * exp(DIR * 2*pi*i * k / 32), k in [0,16) known after unrolling.
// DIR = -1: forward (FFTW_FORWARD sign), +1: backward.
template <typename T, int DIR>
BF_HD void mul_w32(int k, T xr, T xi, T &yr, T &yi) {
    if (k == 0) {
        yr = xr;
        yi = xi;
    } else if (k == 8) {  // exp(DIR*i*pi/2) = DIR*i
        if (DIR < 0) {
            yr = xi;
            yi = -xr;
        } else {
            yr = -xi;
            yi = xr;
        }
    } else if (k == 4) {  // (1 + DIR*i)/sqrt2
        const T r = (T)0.70710678118654752440084436210485;
        if (DIR < 0) {
            yr = (xr + xi) * r;
            yi = (xi - xr) * r;
        } else {
            yr = (xr - xi) * r;
            yi = (xr + xi) * r;
        }
    } else if (k == 12) {  // (-1 + DIR*i)/sqrt2
        const T r = (T)0.70710678118654752440084436210485;
        if (DIR < 0) {
            yr = (xi - xr) * r;
            yi = -(xr + xi) * r;
        } else {
            yr = -(xr + xi) * r;
            yi = (xr - xi) * r;
        }
    } else {
        const T c = (T)cos32(k);
        const T s = (T)(DIR * sin32(k));
        yr = xr * c - xi * s;
        yi = xr * s + xi * c;
    }
}

// Decimation in frequency: natural-order input, output X[brev5(i)] at position i.
template <typename T, int DIR>
BF_HD void fft32_dif(T (&re)[32], T (&im)[32]) {
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int half = 16 >> s;
        const int tstep = 1 << s;
#pragma unroll
        for (int blk = 0; blk < 32; blk += 2 * half) {
#pragma unroll
            for (int j = 0; j < half; ++j) {
                const int a = blk + j, b = a + half;
                const T ar = re[a], ai = im[a], br = re[b], bi = im[b];
                re[a] = ar + br;
                im[a] = ai + bi;
                mul_w32<T, DIR>(j * tstep, ar - br, ai - bi, re[b], im[b]);
            }
        }
    }
}

// Decimation in time: input x[brev5(i)] at position i, natural-order output.
template <typename T, int DIR>
BF_HD void fft32_dit(T (&re)[32], T (&im)[32]) {
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int half = 1 << s;
        const int tstep = 16 >> s;
#pragma unroll
        for (int blk = 0; blk < 32; blk += 2 * half) {
#pragma unroll
            for (int j = 0; j < half; ++j) {
                const int a = blk + j, b = a + half;
                T tr, ti;
                mul_w32<T, DIR>(j * tstep, re[b], im[b], tr, ti);
                const T ar = re[a], ai = im[a];
                re[a] = ar + tr;
                im[a] = ai + ti;
                re[b] = ar - tr;
                im[b] = ai - ti;
            }
        }
    }
}

}  // namespace bf
