// fft1024.hpp -- 1024-point complex FFT of one 32-lane half-wavefront, 32 x 32.
//
// Replaces fftw_execute() on the reference's two fftw_plan_dft_1d plans
// (das.cpp:53,66,127-128): forward = sum x[n] exp(-2 pi i n k/N), backward =
// unnormalised exp(+...).
//
// Index maps (n = 32*n1 + n2 input, k = k1 + 32*k2 output):
//   forward  a: lane n2 holds x[32*j + n2] in reg j -> 32-pt DIF over j, twiddle
//               W1024^(n2*k1), write to the transpose buffer row k1, column n2
//            b: lane k1 reads row k1 (all n2) -> 32-pt DIF over n2
//               => position i of lane k1 holds X[k1 + 32*brev5(i)]
//   backward a: lane k1 holds S[k1 + 32*brev5(i)] at position i -> 32-pt DIT over
//               k2 (bit-reversed in, natural out), twiddle conj(W1024^(n2*k1)),
//               write row n2, column k1
//            b: lane n2 reads row n2 (all k1) -> 32-pt DIF over k1
//               => position i of lane n2 holds y[32*brev5(i) + n2]
// so the forward output layout IS the backward input layout: spectra are
// weighted in place, with no data movement between the two transforms.
//
// The transpose buffer is per half-wavefront; one wavefront executes LDS
// operations in order, so inside a wave only a compiler barrier separates the
// write phase from the read phase.
#pragma once

#include "fft32.hpp"

namespace bf {

template <typename T>
struct cx {
    T x, y;
};

// Row stride (in complex elements) of the transpose buffer.  Reads are one row
// per lane as 16-byte accesses; a lane-to-lane stride of 16 B mod 256 B keeps the
// 16-lane ds_read_b128 groups on distinct banks (MI355X guide, LDS table).
template <typename T>
struct tr_stride;
template <>
struct tr_stride<float> {
    static constexpr int value = 34;  // 272 B
};
template <>
struct tr_stride<double> {
    static constexpr int value = 33;  // 528 B
};

template <typename T>
BF_HD void fft1024_fwd_a(T (&re)[32], T (&im)[32], int lane, const cx<T> *tw, cx<T> *buf) {
    constexpr int RS = tr_stride<T>::value;
    fft32_dif<T, -1>(re, im);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int k1 = brev5(i);
        cx<T> o;
        if (k1 == 0) {
            o.x = re[i];
            o.y = im[i];
        } else {
            const cx<T> w = tw[k1 * 32 + lane];
            o.x = re[i] * w.x - im[i] * w.y;
            o.y = re[i] * w.y + im[i] * w.x;
        }
        buf[k1 * RS + lane] = o;
    }
}

template <typename T>
BF_HD void fft1024_fwd_b(T (&re)[32], T (&im)[32], int lane, const cx<T> *buf) {
    constexpr int RS = tr_stride<T>::value;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        const cx<T> v = buf[lane * RS + c];
        re[c] = v.x;
        im[c] = v.y;
    }
    fft32_dif<T, -1>(re, im);
}

template <typename T>
BF_HD void fft1024_inv_a(T (&re)[32], T (&im)[32], int lane, const cx<T> *tw, cx<T> *buf) {
    constexpr int RS = tr_stride<T>::value;
    fft32_dit<T, +1>(re, im);
#pragma unroll
    for (int n2 = 0; n2 < 32; ++n2) {
        cx<T> o;
        if (n2 == 0) {
            o.x = re[n2];
            o.y = im[n2];
        } else {
            const cx<T> w = tw[n2 * 32 + lane];  // conj(w) applied
            o.x = re[n2] * w.x + im[n2] * w.y;
            o.y = im[n2] * w.x - re[n2] * w.y;
        }
        buf[n2 * RS + lane] = o;
    }
}

template <typename T>
BF_HD void fft1024_inv_b(T (&re)[32], T (&im)[32], int lane, const cx<T> *buf) {
    constexpr int RS = tr_stride<T>::value;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        const cx<T> v = buf[lane * RS + c];
        re[c] = v.x;
        im[c] = v.y;
    }
    fft32_dif<T, +1>(re, im);
}

// ---- plane-split variant ------------------------------------------------------
// Same transform, but the transpose moves the real plane and then the imaginary
// plane through ONE scalar buffer of 32 x plane_stride elements: half the LDS
// footprint per half-wavefront (which is what lets the gain/twiddle/window tables
// live in LDS as well).  Four phases; on the GPU a wavefront runs them back to
// back (LDS ops of one wave execute in order), the CPU emulation runs each phase
// for all 32 lanes before the next.
template <typename T>
struct plane_stride;
template <>
struct plane_stride<float> {
    static constexpr int value = 36;  // 144 B rows: 16-lane ds_read_b128 groups hit distinct banks
};
template <>
struct plane_stride<double> {
    static constexpr int value = 34;  // 272 B rows
};

// phase A (forward): 32-pt DIF, inter-pass twiddle, real plane out
template <typename T>
BF_HD void fft1024p_fwd_A(T (&re)[32], T (&im)[32], int lane, const cx<T> *tw, T *pbuf) {
    constexpr int PS = plane_stride<T>::value;
    fft32_dif<T, -1>(re, im);
#pragma unroll
    for (int i = 1; i < 32; ++i) {
        if ((i & 7) == 0) BF_SCHED_FENCE();
        const int k1 = brev5(i);
        const cx<T> w = tw[k1 * 32 + lane];
        const T xr = re[i], xi = im[i];
        re[i] = xr * w.x - xi * w.y;
        im[i] = xr * w.y + xi * w.x;
    }
    BF_SCHED_FENCE();
#pragma unroll
    for (int i = 0; i < 32; ++i) pbuf[brev5(i) * PS + lane] = re[i];
}
// phase A (backward): 32-pt DIT (bit-reversed in, natural out), conj twiddle, real plane out
template <typename T>
BF_HD void fft1024p_inv_A(T (&re)[32], T (&im)[32], int lane, const cx<T> *tw, T *pbuf) {
    constexpr int PS = plane_stride<T>::value;
    fft32_dit<T, +1>(re, im);
#pragma unroll
    for (int n2 = 1; n2 < 32; ++n2) {
        if ((n2 & 7) == 0) BF_SCHED_FENCE();
        const cx<T> w = tw[n2 * 32 + lane];
        const T xr = re[n2], xi = im[n2];
        re[n2] = xr * w.x + xi * w.y;
        im[n2] = xi * w.x - xr * w.y;
    }
#pragma unroll
    for (int n2 = 0; n2 < 32; ++n2) pbuf[n2 * PS + lane] = re[n2];
}
// phase B: real plane in (row `lane`)
template <typename T>
BF_HD void fft1024p_B(T (&re)[32], int lane, const T *pbuf) {
    constexpr int PS = plane_stride<T>::value;
#pragma unroll
    for (int c = 0; c < 32; ++c) re[c] = pbuf[lane * PS + c];
}
// phase C: imaginary plane out; `natural` selects the register->row map of the phase-A that ran
template <typename T, bool NATURAL>
BF_HD void fft1024p_C(const T (&im)[32], int lane, T *pbuf) {
    constexpr int PS = plane_stride<T>::value;
#pragma unroll
    for (int i = 0; i < 32; ++i) pbuf[(NATURAL ? i : brev5(i)) * PS + lane] = im[i];
}
// phase D: imaginary plane in, second 32-pt DIF (DIR = -1 forward, +1 backward)
template <typename T, int DIR>
BF_HD void fft1024p_D(T (&re)[32], T (&im)[32], int lane, const T *pbuf) {
    constexpr int PS = plane_stride<T>::value;
#pragma unroll
    for (int c = 0; c < 32; ++c) im[c] = pbuf[lane * PS + c];
    fft32_dif<T, DIR>(re, im);
}

}  // namespace bf
