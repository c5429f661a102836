// fft1024_w64.hpp -- 1024-point complex FFT of one full wavefront: 64 lanes x 16 points per lane.
//
// Why a second factorisation: the 32 x 32 half-wavefront scheme (fft1024.hpp) keeps 64 data +
// 64 accumulator registers per lane, which caps the fused DAS kernel at two wavefronts per SIMD,
// and gfx950 issues at most one VALU instruction per wavefront every ~4.8 cycles
// (tools/ubench/valu_rate.hip), so two wavefronts must both be runnable all the time to fill the
// VALU.  16 points per lane halves the register footprint (4 wavefronts per SIMD) at the price
// of a third pass:
//
//   n = 64*n1 + m,  m = 4*a + b   (n1 in [0,16) registers; a in [0,16), b in [0,4))
//   first pass: lane = m (global loads and stores stay lane-contiguous: a permuted lane order costs 8x the TA/TCP work);
//   after T1:   lane = 16*b + k1: the 16-lane DPP row is b, the lane inside the row is k1
//   k = k1 + 16*k2 + 256*k3
//   P1  16-point FFT over n1 (registers)            -> k1          lane 4a+b,    reg k1
//   TW1 multiply by W1024^(m*k1)
//   T1  LDS transpose: reg k1 <-> lane field a                      lane (b,k1),  reg a
//       (writer 4a+b stores register k1 at row k1, column 16 b + a; the reader takes columns 16 b .. 16 b + 15 of row
//        k1 = its lane & 15 as four ds_read_b128)
//   P2  16-point FFT over a (registers)             -> k2          lane (b,k1),  reg k2
//   TW2 multiply by W64^(b*k2)
//   T2  4x4 transpose across the four 16-lane rows (v_permlane32_swap, v_permlane16_swap), reg field k2&3 <-> row b
//                                                                   lane (k2&3,k1), reg (k2>>2, b)
//   P3  four 4-point FFTs over b (registers)        -> k3          lane (k2&3,k1), reg (k2>>2, k3)
//
// The backward transform runs the exact mirror (P3^-1, T2, conj TW2, P2^-1, T1^-1, conj TW1, P1^-1),
// so the forward output layout is the backward input layout: spectra are weighted and summed in
// place.  Register index of bin k after the forward transform:  r = 4*(k2>>2) + k3,
// lane = 16*(k2&3) + k1.
//
// This header holds the per-lane arithmetic (host + device); the two cross-lane steps are
// supplied by the caller (LDS + DPP on the GPU, plain index permutations in the CPU emulation).
#pragma once

#include "fft32.hpp"

namespace bf {

constexpr int brev4(int i) { return ((i & 1) << 3) | ((i & 2) << 1) | ((i & 4) >> 1) | ((i & 8) >> 3); }

// 16-point in-place radix-2 DIT with fused twiddles; PERM_BREV as in fft32_core:
//   false: physical in = bit-reversed (x[brev4(i)] at i), out natural
//   true : physical in = natural, out X[brev4(i)] at i
// S0 = first stage to run: 1 when the caller has done stage 0 itself (with PERM_BREV that stage is the plain butterfly of the
// physical positions p and p + 8, p < 8 -- das_f64_w64.hip folds the analysis window into it)
template <typename T, int DIR, bool PERM_BREV, int S0 = 0>
BF_HD void fft16_core(T (&re)[16], T (&im)[16]) {
#pragma unroll
    for (int s = S0; s < 4; ++s) {
        const int half = 1 << s;
        const int tstep = 16 >> s;  // in units of the 32-point twiddle table: W16^j = W32^(2j)
#pragma unroll
        for (int blk = 0; blk < 16; blk += 2 * half) {
#pragma unroll
            for (int j = 0; j < half; ++j) {
                const int la = blk + j, lb = la + half;
                const int a = PERM_BREV ? brev4(la) : la;
                const int b = PERM_BREV ? brev4(lb) : lb;
                bfly_dit<T, DIR>(j * tstep, re[a], im[a], re[b], im[b]);
            }
        }
    }
}

// 4-point DFT on registers (i0..i3) in natural order in and out.
template <typename T, int DIR>
BF_HD void fft4(T &r0, T &i0, T &r1, T &i1, T &r2, T &i2, T &r3, T &i3) {
    const T ar = r0 + r2, ai = i0 + i2, br = r0 - r2, bi = i0 - i2;
    const T cr = r1 + r3, ci = i1 + i3, dr = r1 - r3, di = i1 - i3;
    r0 = ar + cr;
    i0 = ai + ci;
    r2 = ar - cr;
    i2 = ai - ci;
    if (DIR < 0) {  // X1 = b - i d, X3 = b + i d
        r1 = br + di;
        i1 = bi - dr;
        r3 = br - di;
        i3 = bi + dr;
    } else {
        r1 = br - di;
        i1 = bi + dr;
        r3 = br + di;
        i3 = bi - dr;
    }
}

// ---- per-lane phases (T = float in the kernel; double in the emulation for index checks) ----------
// Register conventions: all 16-point passes take natural input and leave bit-reversed output
// (fft16_core<.., true>) or the reverse; the maps below say where each logical index lives.

// column of the T1 plane that first-pass lane 4a+b exchanges with the second-pass lanes: 16 b + a
constexpr int w64_col(int lane) { return 16 * (lane & 3) + (lane >> 2); }

// fp64 kernel (das_f64_w64.hip): the same exchange with segment b of every row rotated by 4 b columns, col = 16 b + ((a + 4 b) & 15).
// With 8-byte elements the plain map puts the 16 lanes of a ds_write_b64 group (a = 4 g .. 4 g + 3, every b) on 4 bank pairs
// (4-way conflict); rotated, they cover all 16.  The second-pass lane (b, k1) then holds first-pass lane a at register position
// (a + 4 b) & 15: a circular shift of the input of its 16-point transform, i.e. a factor W16^(4 b k2) on output k2, which the
// TW2 table absorbs: tw2'[b][k2] = W64^(b k2) conj(W16^(4 b k2)) = exp(2 pi i 15 b k2 / 64) (geometry.hpp twiddle_table_w64_rot).
// The backward transform multiplies by conj(tw2') and so leaves its output shifted the same way: the same table serves both.
constexpr int w64_col_rot(int lane) { return 16 * (lane & 3) + (((lane >> 2) + 4 * (lane & 3)) & 15); }

// forward P1 + TW1.  in: reg j = x[64*j + lane].  out: position i holds A[k1 = brev4(i)] * W1024^(lane*k1)
template <typename T, typename TW>
BF_HD void w64_fwd_p1(T (&re)[16], T (&im)[16], int lane, const TW *tw1 /* [k1][lane] = W1024^(lane*k1) */) {
    fft16_core<T, -1, true>(re, im);
#pragma unroll
    for (int i = 1; i < 16; ++i) {
        const int k1 = brev4(i);
        const TW w = tw1[k1 * 64 + lane];
        const T xr = re[i], xi = im[i];
        re[i] = xr * w.x - xi * w.y;
        im[i] = xr * w.y + xi * w.x;
    }
}
// after T1 (reg a natural): forward P2 + TW2.  out: position i holds C[k2 = brev4(i)] * W64^(b*k2), b = lane >> 4
// TW2S = row stride of the tw2 table in elements (16, or 17 where the four rows must start on different LDS banks)
template <typename T, typename TW, int TW2S = 16>
BF_HD void w64_fwd_p2(T (&re)[16], T (&im)[16], int lane, const TW *tw2 /* [b][TW2S] = W64^(b*k2) */) {
    fft16_core<T, -1, true>(re, im);
    const int b = lane >> 4;
#pragma unroll
    for (int i = 1; i < 16; ++i) {
        const int k2 = brev4(i);
        const TW w = tw2[b * TW2S + k2];
        const T xr = re[i], xi = im[i];
        re[i] = xr * w.x - xi * w.y;
        im[i] = xr * w.y + xi * w.x;
    }
}
// after T2 the caller has arranged: register 4*g + b  (g = k2>>2, b = old lane field).  forward P3.
template <typename T>
BF_HD void w64_fwd_p3(T (&re)[16], T (&im)[16]) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
        fft4<T, -1>(re[4 * g], im[4 * g], re[4 * g + 1], im[4 * g + 1], re[4 * g + 2], im[4 * g + 2], re[4 * g + 3], im[4 * g + 3]);
}

// backward mirror
template <typename T>
BF_HD void w64_inv_p3(T (&re)[16], T (&im)[16]) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
        fft4<T, +1>(re[4 * g], im[4 * g], re[4 * g + 1], im[4 * g + 1], re[4 * g + 2], im[4 * g + 2], re[4 * g + 3], im[4 * g + 3]);
}
// after T2 (back): position i holds element k2 = brev4(i) (same register map the forward P2 left), b = lane >> 4.
// conj TW2 then inverse 16-point over k2 (bit-reversed in, natural out): reg a natural.
template <typename T, typename TW, int TW2S = 16>
BF_HD void w64_inv_p2(T (&re)[16], T (&im)[16], int lane, const TW *tw2) {
    const int b = lane >> 4;
#pragma unroll
    for (int i = 1; i < 16; ++i) {
        const int k2 = brev4(i);
        const TW w = tw2[b * TW2S + k2];
        const T xr = re[i], xi = im[i];
        re[i] = xr * w.x + xi * w.y;
        im[i] = xi * w.x - xr * w.y;
    }
    fft16_core<T, +1, false>(re, im);
}
// after T1 (back): position i holds element k1 = brev4(i); conj TW1 then inverse 16-point over k1: reg n1 natural.
template <typename T, typename TW>
BF_HD void w64_inv_p1(T (&re)[16], T (&im)[16], int lane, const TW *tw1) {
#pragma unroll
    for (int i = 1; i < 16; ++i) {
        const int k1 = brev4(i);
        const TW w = tw1[k1 * 64 + lane];
        const T xr = re[i], xi = im[i];
        re[i] = xr * w.x + xi * w.y;
        im[i] = xi * w.x - xr * w.y;
    }
    fft16_core<T, +1, false>(re, im);
}

// Bin held by (lane, register r) after the forward transform, and its inverse map.
constexpr int w64_bin(int lane, int r) { return (lane & 15) + 16 * ((lane >> 4) + 4 * (r >> 2)) + 256 * (r & 3); }

}  // namespace bf
