// fft_small.hpp -- N = 512 / 256 / 128 on the register-resident machinery of fft1024.hpp: N = 32 x NL.
//
// Lane (g, n2) of a 32-lane half-wavefront -- g = lane / NL one of G = 32 / NL frames, n2 = lane mod NL -- holds x_g[NL j + n2] in
// register j.  Forward: fft32_dif over j (position i <-> k1 = brev5(i)), twiddle W_N^(n2 k1), the 32 x 32 plane transpose of
// fft1024.hpp (row k1, column lane), then in lane k1 one NL-point DIF per frame on registers g NL .. g NL + NL - 1:
//   position g NL + i' of lane k1 = X_g[k1 + 32 brev(i')].
// Backward: the same steps in reverse (NL-point DIT per frame, conjugate twiddle, transpose, fft32_dif<+1>):
//   position i of lane (g, n2) = y_g[NL brev5(i) + n2].
// Plain C++ like fft32.hpp: compiled by hipcc for gfx950 and by g++ for the host emulation (tests/host_emul).
#pragma once

#include "fft32.hpp"

namespace bf {

constexpr int brevn(int i, int logn) { return logn == 0 ? 0 : ((i & 1) << (logn - 1)) | brevn(i >> 1, logn - 1); }

// 2^LOGN-point DIF on registers OFF .. OFF + 2^LOGN - 1: natural order in, X[brev(i)] at position OFF + i (fft32_core's butterflies on a
// shorter block: stage s pairs logical positions 2^s apart with twiddle exp(DIR 2 pi i j / 2^(s+1)) = the 32nd root to the power j 16 / 2^s)
template <typename T, int DIR, int LOGN, int OFF>
BF_HD void fftn_dif(T (&re)[32], T (&im)[32]) {
#pragma unroll
    for (int st = 0; st < LOGN; ++st) {
        const int half = 1 << st, tstep = 16 >> st;
#pragma unroll
        for (int blk = 0; blk < (1 << LOGN); blk += 2 * half) {
#pragma unroll
            for (int j = 0; j < half; ++j) {
                const int pa = OFF + brevn(blk + j, LOGN), pb = OFF + brevn(blk + j + half, LOGN);
                bfly_dit<T, DIR>(j * tstep, re[pa], im[pa], re[pb], im[pb]);
            }
        }
    }
}
// the same butterflies on physical positions: X[brev(i)]-ordered in, natural order out
template <typename T, int DIR, int LOGN, int OFF>
BF_HD void fftn_dit(T (&re)[32], T (&im)[32]) {
#pragma unroll
    for (int st = 0; st < LOGN; ++st) {
        const int half = 1 << st, tstep = 16 >> st;
#pragma unroll
        for (int blk = 0; blk < (1 << LOGN); blk += 2 * half) {
#pragma unroll
            for (int j = 0; j < half; ++j)
                bfly_dit<T, DIR>(j * tstep, re[OFF + blk + j], im[OFF + blk + j], re[OFF + blk + j + half], im[OFF + blk + j + half]);
        }
    }
}
// every frame of the half-wavefront: G = 32 >> LOGN blocks
template <typename T, int DIR, int LOGN, int G0 = 0>
BF_HD void fftn_dif_all(T (&re)[32], T (&im)[32]) {
    if constexpr (G0 < (32 >> LOGN)) {
        fftn_dif<T, DIR, LOGN, (G0 << LOGN)>(re, im);
        fftn_dif_all<T, DIR, LOGN, G0 + 1>(re, im);
    }
}
template <typename T, int DIR, int LOGN, int G0 = 0>
BF_HD void fftn_dit_all(T (&re)[32], T (&im)[32]) {
    if constexpr (G0 < (32 >> LOGN)) {
        fftn_dit<T, DIR, LOGN, (G0 << LOGN)>(re, im);
        fftn_dit_all<T, DIR, LOGN, G0 + 1>(re, im);
    }
}

}  // namespace bf
