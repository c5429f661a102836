// w64_f64_dev.hpp -- device-side pieces of the 64-lane x 16-point FFT-1024 in double (fft1024_w64.hpp) that more than one kernel uses:
// the exchange through a per-wavefront LDS plane (T1), the 4 x 4 row transpose by permlane swaps (T2), staged twiddle access.
// Users: das_f64_w64.hip (das in double), mask_kernels.hip (stft_bins_w64_kernel: phase / phasempf / das with a spectrum dump).
#pragma once

#include <hip/hip_runtime.h>

#include "fft1024.hpp"
#include "fft1024_w64.hpp"
#include "geometry.hpp"

namespace bf {
namespace {

constexpr int kRS = 65;             // doubles per exchange-plane row (odd: rows k1 = 0..15 start on distinct bank pairs)
constexpr int kPlaneD = 16 * kRS;   // doubles per wavefront

// ---- T2: 4 x 4 transpose across the four 16-lane rows, for doubles (two dwords each) --------------------------------------
// v_permlane32_swap a, b: rows {2,3} of a <-> rows {0,1} of b;  v_permlane16_swap a, b: odd rows of a <-> even rows of b (lane
// semantics checked on the device by tools/ubench/permswap.hip; the builtins, not inline asm: tied 32-bit halves cost ~4 v_mov per block).  One block
// moves the low and the high dwords of four doubles: the four independent swaps between a register's two swaps cover the wait
// states a swap needs behind the instruction that wrote its operand; the leading s_nop covers the VALU in front of the block.
__device__ __forceinline__ void swap32(unsigned &a, unsigned &b) {
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
__device__ __forceinline__ void swap16(unsigned &a, unsigned &b) {
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
__device__ __forceinline__ void row_transpose4(double &d0, double &d1, double &d2, double &d3) {
    unsigned l0 = (unsigned)__double2loint(d0), l1 = (unsigned)__double2loint(d1), l2 = (unsigned)__double2loint(d2), l3 = (unsigned)__double2loint(d3);
    unsigned h0 = (unsigned)__double2hiint(d0), h1 = (unsigned)__double2hiint(d1), h2 = (unsigned)__double2hiint(d2), h3 = (unsigned)__double2hiint(d3);
    swap32(l0, l2); swap32(l1, l3); swap32(h0, h2); swap32(h1, h3);
    swap16(l0, l1); swap16(l2, l3); swap16(h0, h1); swap16(h2, h3);
    d0 = __hiloint2double((int)h0, (int)l0);
    d1 = __hiloint2double((int)h1, (int)l1);
    d2 = __hiloint2double((int)h2, (int)l2);
    d3 = __hiloint2double((int)h3, (int)l3);
}
constexpr int brev2c(int i) { return ((i & 1) << 1) | ((i >> 1) & 1); }

// T2: position brev2(g) + 4*brev2(q) (row b)  <->  register 4*g + b (row q)   
template <bool FWD>
__device__ __forceinline__ void w64_T2(double (&re)[16], double (&im)[16]) {
    double nr[16], ni[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        double r[4], s[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int src = FWD ? brev2c(g) + 4 * brev2c(c) : 4 * g + c;
            r[c] = re[src];
            s[c] = im[src];
        }
        row_transpose4(r[0], r[1], r[2], r[3]);
        row_transpose4(s[0], s[1], s[2], s[3]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int dst = FWD ? 4 * g + c : brev2c(g) + 4 * brev2c(c);
            nr[dst] = r[c];
            ni[dst] = s[c];
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        re[i] = nr[i];
        im[i] = ni[i];
    }
}

#ifdef BF_T1_MERGED_READS
#define BF_T1_RD(p) (*(p))
#else
#define BF_T1_RD(p) (*(const volatile __attribute__((address_space(3))) double *)(p))
#endif

// (The same transpose through the exchange plane instead of the swaps -- 64 ds_write_b64 + 64 ds_read_b64 per transform on a conflict-free
// column map -- was measured in round 4: 0.575 ms against 0.545 for das_f64_pair_kernel; the two extra LDS round trips per transform cost more
// than the 1 000 vector cycles they free.  EXPERIMENTS.md, round 4; the code is gone.)
template <bool FWD>
__device__ __forceinline__ void w64_T2_any(double (&re)[16], double (&im)[16], double *prow, int hi) {
    (void)prow; (void)hi;
    w64_T2<FWD>(re, im);
}

// ---- T1 through one scalar plane (real parts, then imaginary parts) -----------------------------------------------------------
// forward: position i (k1 = brev4(i)) of lane 4a+b -> register position (a + 4 b) & 15 of lane 16 b + k1.  LDS operations of one
// wavefront execute in issue order: only compiler barriers separate the phases.
// exchange reads as single ds_read_b64 (2 LDS cycles per 512 B): merged into ds_read2_b64 by the compiler they take 8 cycles per 1 KB
__device__ __forceinline__ void T1_fwd(double (&re)[16], double (&im)[16], double *wcol, const double *row16) {
#pragma unroll
    for (int i = 0; i < 16; ++i) wcol[brev4(i) * kRS] = re[i];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < 16; ++c) re[c] = BF_T1_RD(row16 + c);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 16; ++i) wcol[brev4(i) * kRS] = im[i];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < 16; ++c) im[c] = BF_T1_RD(row16 + c);
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void T1_inv(double (&re)[16], double (&im)[16], double *row16, const double *wcol) {
#pragma unroll
    for (int c = 0; c < 16; ++c) row16[c] = re[c];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 16; ++i) re[i] = BF_T1_RD(wcol + brev4(i) * kRS);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < 16; ++c) row16[c] = im[c];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 16; ++i) im[i] = BF_T1_RD(wcol + brev4(i) * kRS);
    __builtin_amdgcn_wave_barrier();
}

// ---- frame-pairing test of the backward transforms (istft_w64_kernel, mpf_rec_istft_kernel) ---------------------------------------
// |x| as an unsigned integer that orders like the magnitude (high word without the sign; NaN / Inf >= 0x7FF00000)
__device__ __forceinline__ unsigned hi_abs(double x) { return (unsigned)((unsigned long long)__double_as_longlong(x) >> 32) & 0x7fffffffu; }
// the largest value over the 64 lanes, in every lane (prefix maxima along the rows by DPP, lane 15 of each row to the next rows, lane 63)
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true));  // row_shr:1
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true));  // row_shr:2
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true));  // row_shr:4
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true));  // row_shr:8
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true));  // row_bcast:15 into rows 1 and 3
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true));  // row_bcast:31 into rows 2 and 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}


// ---- staged twiddle / gain access ------------------------------------------------------------------------------------------
// The per-lane passes of fft1024_w64.hpp fetch each twiddle right where it is multiplied in; at 256 registers hipcc then keeps one or
// two ds_read_b128 in flight and every fourth instruction waits a full LDS round trip (36 % of the wave cycles in s_waitcnt,
// profiles/r04_a_das8_f64_w64_pmc.txt).  Here a pass's 15 twiddles (60 registers) are requested as a block BEFORE the 16-point
// transform that precedes their use and are all there when it ends; sched_barrier keeps hipcc from sinking them back.
#define BF_STAGE() __builtin_amdgcn_sched_barrier(0)

// (two batches: 8 twiddles ahead of the transform, the other 7 requested when the multiplication starts and consumed last)
template <int LO, int HI>
__device__ __forceinline__ void load_tw1(cx<double> (&tw)[15], const cx<double> *s_tw1, int lane) {
#pragma unroll
    for (int i = LO; i < HI; ++i) tw[i - 1] = s_tw1[brev4(i) * 64 + lane];  // W1024^(lane * k1), k1 = brev4(i)
}
template <int LO, int HI>
__device__ __forceinline__ void load_tw2(cx<double> (&tw)[15], const cx<double> *s_tw2, int lane) {
    const cx<double> *row = s_tw2 + (lane >> 4) * kTw2RowW64Rot;
#pragma unroll
    for (int i = LO; i < HI; ++i) tw[i - 1] = row[brev4(i)];
}
template <bool CONJ, int LO, int HI>
__device__ __forceinline__ void mul_tw(double (&re)[16], double (&im)[16], const cx<double> (&tw)[15]) {
#pragma unroll
    for (int i = LO; i < HI; ++i) {
        const cx<double> w = tw[i - 1];
        const double xr = re[i], xi = im[i];
        if (!CONJ) {
            re[i] = xr * w.x - xi * w.y;
            im[i] = xr * w.y + xi * w.x;
        } else {
            re[i] = xr * w.x + xi * w.y;
            im[i] = xi * w.x - xr * w.y;
        }
    }
}

}  // namespace
}  // namespace bf
