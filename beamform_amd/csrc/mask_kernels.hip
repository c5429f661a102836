// mask_kernels.hip -- the pointwise nodes (das in fp64, phase), phasempf (mask + MCRA/MPF recursion) and the mcra node.
#include <cstdlib>

#include "launch_trace.hpp"
#include "bins_common.hpp"
#include "fft_small.hpp"
#if BF_NFFT == 1024
#include "w64_f64_dev.hpp"
#endif

namespace bf {
namespace BF_NTAG {

namespace {

// ======================================================================================
//                         pointwise per-bin kernels: das, phase
// ======================================================================================
struct BinCtx {
    const f64x2 *Zf;   // packed spectra of this frame [NP][1024]
    const f64x2 *steer;
    int M, q;
};

// das.cpp:60-63 on the unpacked spectra X[m] and steering entries w[m] of one bin
template <int MP>
__device__ __forceinline__ cd das_core(const cd (&X)[MP], const cd (&w)[MP], int M) {
    cd acc{0, 0};
#pragma unroll
    for (int m = 0; m < MP; ++m)
        if (m < M) acc = acc + conj(w[m]) * X[m];
    const double dM = (double)M, rM = 1.0 / dM;  // M is uniform: one division per thread, hoisted out of the item loop
    return cd{div_rcp(acc.x, dM, rM), div_rcp(acc.y, dM, rM)};
}
template <int MP>
__device__ __forceinline__ void load_steer(const f64x2 *steer, int j, int M, cd (&w)[MP]) {
#pragma unroll
    for (int m = 0; m < MP; ++m) w[m] = (m < M) ? ld(steer + (long)m * kN + j) : cd{0, 0};
}
template <int MP>
__device__ __forceinline__ cd das_bin(const BinCtx &c) {
    cd X[MP], w[MP];
    load_X<MP>(c.Zf, c.q, c.M, X);
    load_steer<MP>(c.steer, q_bin(c.q), c.M, w);
    return das_core<MP>(X, w, c.M);
}

// mean over mic pairs of the wrapped |p_m - p_m'| with the reference's summation order
// (get_overall_phase_diff, phase.cpp:53-68)
template <int MP>
__device__ __forceinline__ double pair_phase_mean(const double (&ph)[MP], int M) {
    double d[MP];
#pragma unroll
    for (int i = 0; i < MP; ++i) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < MP; ++k) {
            if (k > i && k < M) {
                double r = fabs(ph[i] - ph[k]);
                if (r > M_PI) r = 2 * M_PI - r;
                acc += r;
            }
        }
        d[i] = acc;
    }
    double tot = 0.0;
#pragma unroll
    for (int i = MP - 1; i >= 0; --i)
        if (i < M - 1) tot = d[i] + tot;
    const int num = M * (M - 1) / 2;
    const double dn = (double)num;
    return div_rcp(tot, dn, 1.0 / dn);  // 0/0 = NaN when M == 1, as the reference
}

// The mask decision `mean wrapped phase difference < min_phase` (phase.cpp:110, phasempf.cpp:229) is 8 atan2 + 28 wrapped
// differences per bin-frame in the reference's double arithmetic -- most of these nodes' per-bin time.  The decision (not
// the output) is first formed in fp32: arguments rounded to float, atan2f_fast_n (error < 2e-6 rad), the same pairwise mean
// (fp32 summation error < 1e-5).  Only when that mean lies within 1e-4 rad of the threshold -- or an argument is outside the
// float range -- does the wavefront redo the test in double (~6e-5 of the bin-frames, 0.4 % of the wavefronts), so every
// decision equals the double-precision one.
template <int MP>
__device__ __forceinline__ float pair_phase_mean_f(const float (&ph)[MP], int M) {
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < MP; ++i)
#pragma unroll
        for (int k = i + 1; k < MP; ++k)
            if (k < M) {
                float r = __builtin_fabsf(ph[i] - ph[k]);
                if (r > 3.14159265f) r = 6.28318531f - r;
                tot += r;
            }
    return tot / (float)(M * (M - 1) / 2);  // 0/0 = NaN when M == 1: falls through to the double path
}
template <int MP>
__device__ __forceinline__ bool phase_is_close(const double (&uy)[MP], const double (&ux)[MP], int M, double thr) {
    float fy[MP], fx[MP], ph[MP];
    bool odd = false;  // an argument the float range cannot carry (|u| of [-1,1] audio spectra sits around 1e-3 .. 1e3)
#pragma unroll
    for (int m = 0; m < MP; ++m) {
        fy[m] = (float)uy[m];
        fx[m] = (float)ux[m];
        const float big = __builtin_fmaxf(__builtin_fabsf(fx[m]), __builtin_fabsf(fy[m]));
        if (m < M) odd = odd || !(big > 1e-30f && big < 1e30f);
    }
    atan2f_fast_n<MP>(fy, fx, ph);
    const float mean = pair_phase_mean_f<MP>(ph, M), thrf = (float)thr;
    const bool unsure = odd || !(__builtin_fabsf(mean - thrf) > 1e-4f);
    if (__builtin_amdgcn_ballot_w64(unsure) != 0) {  // rare: the reference's arithmetic for this wavefront
        double pd[MP];
        atan2_fast_n<MP>(uy, ux, pd);
        return pair_phase_mean<MP>(pd, M) < thr;
    }
    return mean < thrf;
}

// phase.cpp:87-127 on the unpacked spectra X[m] and steering entries w[m] of bin j
template <int MP>
__device__ __forceinline__ cd phase_core(const cd (&X)[MP], const cd (&w)[MP], int M, int j, const bf_config &cfg) {
    if (j == 0) return X[0];
    double ab[MP], mag = 0.0;
    cabs_n<MP>(X, ab);
#pragma unroll
    for (int m = 0; m < MP; ++m)
        if (m < M) mag += ab[m];
    mag = div_rcp(mag, (double)M, 1.0 / (double)M);
    bool keep = false;
    if (mag / (double)kN > cfg.mag_threshold) {
        double uy[MP], ux[MP];
#pragma unroll
        for (int m = 0; m < MP; ++m) {
            const cd u = conj(w[m]) * X[m];  // padding channels: w = X = 0 -> their phase is never read
            uy[m] = u.y;
            ux[m] = u.x;
        }
        keep = phase_is_close<MP>(uy, ux, M, cfg.min_phase * M_PI / 180);
    }
    if (!keep) mag *= cfg.mag_mult;
    // mag * (cos, sin)(arg X_0) = mag * X_0 / |X_0|  (phase.cpp:115-122); arg(0) = 0
    if (ab[0] == 0.0) return cd{mag, 0.0};
    const double sc = mag * fast_rcp(ab[0]);
    return cd{sc * X[0].x, sc * X[0].y};
}
// phase_core with the steering entries fetched only where the magnitude gate is open (stft_bins_w64_kernel re-reads them per item from
// L1 / L2: on a frame below the gate that is 128 bytes per bin for nothing); the same arithmetic, bit for bit
template <int MP>
__device__ __forceinline__ cd phase_core_lazy(const cd (&X)[MP], const f64x2 *steer, int M, int j, const bf_config &cfg) {
    if (j == 0) return X[0];
    double ab[MP], mag = 0.0;
    cabs_n<MP>(X, ab);
#pragma unroll
    for (int m = 0; m < MP; ++m)
        if (m < M) mag += ab[m];
    mag = div_rcp(mag, (double)M, 1.0 / (double)M);
    bool keep = false;
    const bool open = mag / (double)kN > cfg.mag_threshold;
    if (__builtin_amdgcn_ballot_w64(open) != 0) {  // wavefront-uniform: nobody loads when every lane is below the gate
        cd w[MP];
        load_steer<MP>(steer, j, M, w);
        if (open) {
            double uy[MP], ux[MP];
#pragma unroll
            for (int m = 0; m < MP; ++m) {
                const cd u = conj(w[m]) * X[m];
                uy[m] = u.y;
                ux[m] = u.x;
            }
            keep = phase_is_close<MP>(uy, ux, M, cfg.min_phase * M_PI / 180);
        }
    }
    if (!keep) mag *= cfg.mag_mult;
    if (ab[0] == 0.0) return cd{mag, 0.0};
    const double sc = mag * fast_rcp(ab[0]);
    return cd{sc * X[0].x, sc * X[0].y};
}
template <int MP>
__device__ __forceinline__ cd phase_bin(const BinCtx &c, const bf_config &cfg) {
    cd X[MP], w[MP];
    load_X<MP>(c.Zf, c.q, c.M, X);
    const int j = q_bin(c.q);
    if (j == 0) return X[0];
    load_steer<MP>(c.steer, j, c.M, w);
    return phase_core<MP>(X, w, c.M, j, cfg);
}

// the binary phase mask of phasempf.cpp:210-248 for bin j >= 1: out_soi and |out_int|^2
template <int MP>
__device__ __forceinline__ void mpf_mask_core(const cd (&X)[MP], const cd (&w)[MP], int M, const bf_config &cfg, cd &soi_out,
                                              double &int2_out) {
    double uy[MP], ux[MP], ab[MP];
    double mag = 0.0;
    cabs_n<MP>(X, ab);
#pragma unroll
    for (int m = 0; m < MP; ++m) {
        const cd u = conj(w[m]) * X[m];  // padding channels: w = X = 0 -> their phase is never read
        uy[m] = u.y;
        ux[m] = u.x;
        if (m < M) mag += ab[m];
    }
    const bool is_soi = phase_is_close<MP>(uy, ux, M, cfg.min_phase * M_PI / 180);
    mag = div_rcp(mag, (double)M, 1.0 / (double)M);
    const double lo = mag * cfg.min_mag;
    const double msoi = is_soi ? mag : lo, mint = is_soi ? lo : mag;
    // m * (cos, sin)(arg X_0) = m * X_0 / |X_0| for both magnitudes; arg(0) = 0
    cd soi, in;
    if (ab[0] == 0.0) {
        soi = cd{msoi, 0.0};
        in = cd{mint, 0.0};
    } else {
        const double inv = fast_rcp(ab[0]);
        const cd unit = cd{X[0].x * inv, X[0].y * inv};
        soi = cd{msoi * unit.x, msoi * unit.y};
        in = cd{mint * unit.x, mint * unit.y};
    }
    soi_out = soi;
    int2_out = norm2(in);
}

template <int MP, int ALGO>
__global__ __launch_bounds__(256) void pointwise_bins_kernel(BinsArgs a) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)a.n_streams * a.n_frames * kNQ;
    if (idx >= total) return;
    const int q = (int)(idx % kNQ);
    const long st = idx / kNQ;
    const long t = st % a.n_frames;
    const int s = (int)(st / a.n_frames);
    const int NP = (a.n_mics + 1) >> 1;
    BinCtx c;
    c.Zf = a.Z + (((long)(s / a.n_dirs) * a.frames_ws + a.frame_off + t) * NP) * kN;
    c.steer = a.steer + (long)(s % a.n_dirs) * a.steer_dir_stride;
    c.M = a.n_mics;
    c.q = q;
    cd y;
    if (ALGO == BF_DAS)
        y = das_bin<MP>(c);
    else
        y = phase_bin<MP>(c, a.cfg);
    st_y(a, ((long)s * a.n_frames + t) * kYhStride + q, q, y);
}


// ======================================================================================
//                   phasempf: phase mask (parallel) + MCRA / MPF recursion (sequential)
// ======================================================================================
// Pass 1, one thread per (stream, frame, problem): the binary phase mask of phasempf.cpp:210-248.
// out_soi goes to Yh (complex), |out_int|^2 to aux.
template <int MP>
__global__ __launch_bounds__(256) void mpf_mask_kernel(BinsArgs a, double *aux) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)a.n_streams * a.n_frames * kNQ;
    if (idx >= total) return;
    const int q = (int)(idx % kNQ);
    const long st = idx / kNQ;
    const long t = st % a.n_frames;
    const int s = (int)(st / a.n_frames);
    const int M = a.n_mics, NP = (M + 1) >> 1;
    const f64x2 *Zf = a.Z + (((long)(s / a.n_dirs) * a.frames_ws + a.frame_off + t) * NP) * kN;
    const f64x2 *steer = a.steer + (long)(s % a.n_dirs) * a.steer_dir_stride;
    cd X[MP];
    load_X<MP>(Zf, q, M, X);
    const int j = q_bin(q);
    const long o = ((long)s * a.n_frames + t) * kYhStride + q;
    if (j == 0) {  // out_soi[0] = out_int[0] = in_fft(0,0); the squares at index 0 are never written: defined 0
        a.Yh[o] = f64x2{X[0].x, X[0].y};
        aux[o] = 0.0;
        return;
    }
    cd w[MP], soi;
    double int2;
    load_steer<MP>(steer, j, M, w);
    mpf_mask_core<MP>(X, w, M, a.cfg, soi, int2);
    a.Yh[o] = f64x2{soi.x, soi.y};
    aux[o] = int2;
}

struct MpfState {
    double Sprev, Stmp, Smin, lam, Z, rev0, rev1;
};

// one frame of mcra() + the MPF block + spectral subtraction for one bin (phasempf.cpp:140-191,254-295)
__device__ __forceinline__ cd mpf_step(MpfState &st, const bf_config &c, int j, cd soi, double int2, bool search_reset,
                                       bool firstL, int cL) {
    const double soi2 = (j == 0) ? 0.0 : norm2(soi);
    double Sf;
    if (j == 0) {
        Sf = cabs(soi);
    } else {
        Sf = 0.0;
        if (j - 1 >= 1) Sf += 0.25 * soi2;  // quirk Q15e: every tap multiplies soi2[j]
        Sf += 0.5 * soi2;
        if (j + 1 < kN) Sf += 0.25 * soi2;
    }
    const double S = (c.mcra_alphaS * st.Sprev) + ((1 - c.mcra_alphaS) * Sf);
    if (search_reset) {
        st.Smin = st.Stmp > S ? S : st.Stmp;
        st.Stmp = S;
    } else {
        st.Smin = st.Smin > S ? S : st.Smin;
        st.Stmp = st.Stmp > S ? S : st.Stmp;
    }
    {
        // phasempf.cpp:172-183.  firstL and current_L are per stream, i.e. wavefront-uniform: the division 1 / current_L is formed only while
        // the first search period runs (its result is read only then), and the update itself is a select instead of a divergent branch
        const bool upd = firstL || S < st.Smin * c.mcra_delta || st.lam > soi2;
        double lam_new = c.mcra_alphaD2 * st.lam + (1.0 - c.mcra_alphaD) * soi2;  // quirk Q15g
        if (firstL) {
            const double ic = 1.0 / (double)cL;
            if (ic > c.mcra_alphaD) lam_new = ic * st.lam + (1.0 - ic) * soi2;
        }
        st.lam = upd ? lam_new : st.lam;
    }
    st.Sprev = S;
    st.Z = c.mpf_alphaS * st.Z + (1 - c.mpf_alphaS) * int2;
    const double leak = c.mpf_eta * st.Z;
    const double kq = 1 - c.mpf_rev_gamma / c.mpf_rev_delta;  // quirk Q15i
    st.rev0 = c.mpf_rev_gamma * st.rev0 + kq * soi2;
    st.rev1 = c.mpf_rev_gamma * st.rev1 + kq * int2;
    const double Lam = fast_sqrt(st.lam + leak + st.rev0 + st.rev1);
    if (j == 0) return cd{0, 0};  // quirk Q15d: y_fft[0] is never written; defined 0
    const double as = fast_sqrt(soi2);
    double mg;
    if (c.out_only_noise) {
        mg = Lam * c.out_amp;
    } else {
        mg = (as - (c.out_only_mcra ? fast_sqrt(st.lam) : Lam)) * c.out_amp;
        if (mg < 0) mg = c.noise_floor;
    }
    // mag * (cos, sin)(arg(soi)) == mag * soi/|soi|; arg(0) = 0
    if (as == 0.0) return cd{mg, 0.0};
    const double sc = mg * fast_rcp(as);
    return cd{sc * soi.x, sc * soi.y};
}

// Pass 2, one thread per (stream, problem), sequential over frames.
__global__ __launch_bounds__(64) void mpf_recursion_kernel(BinsArgs a, double *aux) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.n_streams * kNQ) return;
    const int s = idx / kNQ, q = idx % kNQ;
    const int j = q_bin(q);
    double *sv = a.mpf + (long)s * (kMpfVecs * kN + 8);
    MpfState st{sv[0 * kN + j], sv[1 * kN + j], sv[2 * kN + j], sv[3 * kN + j], sv[4 * kN + j], sv[5 * kN + j], sv[6 * kN + j]};
    int cL = (int)sv[kMpfVecs * kN + 0];
    bool firstL = sv[kMpfVecs * kN + 1] == 0.0;  // stored as "first_L is over" flag so a zeroed state = cold start
    f64x2 *row = a.Yh + ((long)s * a.n_frames) * kYhStride + q;
    double *arow = aux + ((long)s * a.n_frames) * kYhStride + q;
    // 2 wavefronts per SIMD at config 4's size and a recursion per thread: the next frames' inputs are requested four frames ahead
    // (a ring of four register sets), otherwise every step waits out its own two loads
    constexpr int kAhead = 4;
    cd soi_r[kAhead];
    double int_r[kAhead];
#pragma unroll
    for (int k = 0; k < kAhead; ++k) {
        const long tt = k < a.n_frames ? k : a.n_frames - 1;
        soi_r[k] = ld(row + tt * kYhStride);
        int_r[k] = arow[tt * kYhStride];
    }
    for (long t0 = 0; t0 < a.n_frames; t0 += kAhead) {
#pragma unroll
      for (int k = 0; k < kAhead; ++k) {
        const long t = t0 + k;
        if (t >= a.n_frames) break;
        const cd soi = soi_r[k];
        const double int2 = int_r[k];
        {
            const long tn = t + kAhead < a.n_frames ? t + kAhead : a.n_frames - 1;
            soi_r[k] = ld(row + tn * kYhStride);
            int_r[k] = arow[tn * kYhStride];
        }
        const bool reset = cL > a.cfg.mcra_L;  // phasempf.cpp:161
        if (reset) {
            cL = 1;
            firstL = false;
        } else {
            cL++;
        }
        const cd y = mpf_step(st, a.cfg, j, soi, int2, reset, firstL, cL);
        // mpf32: the fp32 backward transform is the only reader -- its (float) conversion happens here and the row element takes the
        // 8-byte slot this thread has just read |out_int|^2 from (half the bytes written, half the bytes read back)
        if (a.mpf32)
            reinterpret_cast<f32x2 *>(arow)[t * kYhStride] = f32x2{(float)y.x, (float)y.y};
        else
            row[t * kYhStride] = f64x2{y.x, y.y};
      }
    }
    sv[0 * kN + j] = st.Sprev; sv[1 * kN + j] = st.Stmp; sv[2 * kN + j] = st.Smin; sv[3 * kN + j] = st.lam;
    sv[4 * kN + j] = st.Z; sv[5 * kN + j] = st.rev0; sv[6 * kN + j] = st.rev1;
    if (q == 0) {
        sv[kMpfVecs * kN + 0] = (double)cL;
        sv[kMpfVecs * kN + 1] = firstL ? 0.0 : 1.0;
    }
}

#if BF_NFFT == 1024
// ---- phasempf with many streams: the recursion and the backward transform in one kernel ---------------------------------------------
// With as many streams as CUs the recursion (one lane per (stream, problem), serial over the frames) and the backward transform (one
// wavefront per frame pair) can share a block per stream: the y_fft rows go from the recursion's lanes to the transform through LDS and
// never reach HBM (0.54 GB written + read per 65 536 frames; mpf_recursion_kernel + istft_w64_kernel: 0.34 + 0.195 ms at 256 x 256).
// Block = 7 wavefronts: 0..3 run the recursion of problems r and r + 256, wavefront 6 that of problems 512 and 513; wavefronts 4 and 5
// transform the PREVIOUS batch of four frames (one frame pair each, the code of istft_w64_kernel) while the recursion works on the next:
// two halves of four rows in LDS, one block barrier per batch.  Tails between the two pairs of a batch and between batches travel
// through parity-double-buffered LDS slots (written in one phase, read behind the next barrier).  phasempf.cpp:140-191,254-295 +
// util.h:244-253,301-302.
constexpr int kRiNB = 4;
constexpr int kRiThreads = 448;
constexpr int kRiTwD = 2 * (960 + 4 * kTw2RowW64Rot);   // tw1 rows k1 = 1..15 + tw2' (a.rec_tw_w64 + 64), in doubles
constexpr int kRiWinRow = 18;
constexpr int kRiRowD = 2 * kYhStride;                  // doubles per y_fft row
constexpr int oRiPlane = kRiTwD;
constexpr int oRiWin = oRiPlane + 2 * kPlaneD;
constexpr int oRiRows = oRiWin + 64 * kRiWinRow;        // [half][frame of the batch][kYhStride] c128
constexpr int oRiTail = oRiRows + 2 * kRiNB * kRiRowD;  // floats: [parity][wavefront 4's tail | carried tail][512]
constexpr int kRiLds = oRiTail + 2 * 2 * 512 / 2;
static_assert(kRiLds * 8 <= 160 * 1024 && (oRiRows & 1) == 0 && (oRiWin & 1) == 0, "LDS");

__global__ __launch_bounds__(kRiThreads) void mpf_rec_istft_kernel(BinsArgs a, double *aux) {
    __shared__ __attribute__((aligned(16))) double lds[kRiLds];
    const cx<double> *s_tw1 = reinterpret_cast<const cx<double> *>(lds) - 64;  // row k1 starts at 64 (k1 - 1)
    const cx<double> *s_tw2 = reinterpret_cast<const cx<double> *>(lds) + 960;
    float *s_tail = reinterpret_cast<float *>(lds + oRiTail);                  // [parity][0: wavefront 4's tail, 1: carry][512]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = blockIdx.x;
    const long F = a.n_frames;
    {
        const f64x2 *tw2 = a.rec_tw_w64 + 64;
        f64x2 *ltw = reinterpret_cast<f64x2 *>(lds);
        for (int i = tid; i < kRiTwD / 2; i += kRiThreads) ltw[i] = tw2[i];
        for (int i = tid; i < 1024; i += kRiThreads) lds[oRiWin + (i & 63) * kRiWinRow + (i >> 6)] = a.rec_win[i];
        for (int i = tid; i < kHop; i += kRiThreads) s_tail[(0 * 2 + 1) * 512 + i] = a.rec_tail_in[(long)s * kHop + i];  // carry into batch 0
    }
    const int n_batches = (int)((F + kRiNB - 1) / kRiNB);
    const bool worker = w == 4 || w == 5;
    const int wk = w - 4;  // which pair of the batch
    // ---- transform lanes ---------------------------------------------------------------------------------------------------------------
    double *plane = lds + oRiPlane + (worker ? wk : 0) * kPlaneD;
    double *wcol = plane + w64_col_rot(lane);
    double *row16 = plane + (lane & 15) * kRS + 16 * (lane >> 4);
    const double *wrow = lds + oRiWin + lane * kRiWinRow;
    float *ys = a.rec_y + (long)s * F * kHop;
    __syncthreads();

    // The two roles run their own loops (wavefront-uniform branch; the same number of block barriers on either side), so that the
    // registers of the recursion (input ring, two states) and of the transform are never live together.
    if (!worker) {
        // ---- recursion lanes: problems q0 (and q1) of this stream --------------------------------------------------------------------------
        const int nq = w < 4 ? 2 : (w == 6 && lane < 2) ? 1 : 0;
        const int q0 = w < 4 ? tid : 512 + lane, q1 = tid + 256;
        double *sv = a.mpf + (long)s * (kMpfVecs * kN + 8);
        MpfState st0{}, st1{};
        if (nq >= 1) st0 = MpfState{sv[0 * kN + q_bin(q0)], sv[1 * kN + q_bin(q0)], sv[2 * kN + q_bin(q0)], sv[3 * kN + q_bin(q0)], sv[4 * kN + q_bin(q0)], sv[5 * kN + q_bin(q0)], sv[6 * kN + q_bin(q0)]};
        if (nq >= 2) st1 = MpfState{sv[0 * kN + q_bin(q1)], sv[1 * kN + q_bin(q1)], sv[2 * kN + q_bin(q1)], sv[3 * kN + q_bin(q1)], sv[4 * kN + q_bin(q1)], sv[5 * kN + q_bin(q1)], sv[6 * kN + q_bin(q1)]};
        int cL = (int)sv[kMpfVecs * kN + 0];
        bool firstL = sv[kMpfVecs * kN + 1] == 0.0;
        const f64x2 *row = a.Yh + ((long)s * F) * kYhStride;
        const double *arow = aux + ((long)s * F) * kYhStride;
        cd soi0[kRiNB], soi1[kRiNB];   // the next four frames' inputs: a ring of register sets, refilled as they are consumed
        double in0[kRiNB], in1[kRiNB];
    #pragma unroll
        for (int k = 0; k < kRiNB; ++k) {
            const long tt = k < F ? k : F - 1;
            soi0[k] = soi1[k] = cd{0, 0};
            in0[k] = in1[k] = 0.0;
            if (nq >= 1) { soi0[k] = ld(row + tt * kYhStride + q0); in0[k] = arow[tt * kYhStride + q0]; }
            if (nq >= 2) { soi1[k] = ld(row + tt * kYhStride + q1); in1[k] = arow[tt * kYhStride + q1]; }
        }
        for (int it = 0; it <= n_batches; ++it) {
            const int par = it & 1;
            if (nq > 0 && it < n_batches) {  // ---- the recursion of batch `it` into half `par` ----
                f64x2 *rows = reinterpret_cast<f64x2 *>(lds + oRiRows) + (long)par * kRiNB * kYhStride;
#pragma unroll
                for (int k = 0; k < kRiNB; ++k) {
                    const long t = (long)it * kRiNB + k;
                    if (t >= F) break;
                    const cd sa = soi0[k], sb = soi1[k];
                    const double ia = in0[k], ib = in1[k];
                    {
                        const long tn = t + kRiNB < F ? t + kRiNB : F - 1;
                        soi0[k] = ld(row + tn * kYhStride + q0);
                        in0[k] = arow[tn * kYhStride + q0];
                        if (nq >= 2) { soi1[k] = ld(row + tn * kYhStride + q1); in1[k] = arow[tn * kYhStride + q1]; }
                    }
                    const bool reset = cL > a.cfg.mcra_L;  // phasempf.cpp:161
                    if (reset) {
                        cL = 1;
                        firstL = false;
                    } else {
                        cL++;
                    }
                    const cd ya = mpf_step(st0, a.cfg, q_bin(q0), sa, ia, reset, firstL, cL);
                    rows[k * kYhStride + q0] = f64x2{ya.x, ya.y};
                    if (nq >= 2) {
                        const cd yb = mpf_step(st1, a.cfg, q_bin(q1), sb, ib, reset, firstL, cL);
                        rows[k * kYhStride + q1] = f64x2{yb.x, yb.y};
                    }
                }
            }
            __syncthreads();
        }
        if (nq >= 1) {
            const int j0 = q_bin(q0);
            sv[0 * kN + j0] = st0.Sprev; sv[1 * kN + j0] = st0.Stmp; sv[2 * kN + j0] = st0.Smin; sv[3 * kN + j0] = st0.lam;
            sv[4 * kN + j0] = st0.Z; sv[5 * kN + j0] = st0.rev0; sv[6 * kN + j0] = st0.rev1;
        }
        if (nq >= 2) {
            const int j1 = q_bin(q1);
            sv[0 * kN + j1] = st1.Sprev; sv[1 * kN + j1] = st1.Stmp; sv[2 * kN + j1] = st1.Smin; sv[3 * kN + j1] = st1.lam;
            sv[4 * kN + j1] = st1.Z; sv[5 * kN + j1] = st1.rev0; sv[6 * kN + j1] = st1.rev1;
        }
        if (tid == 0) {
            sv[kMpfVecs * kN + 0] = (double)cL;
            sv[kMpfVecs * kN + 1] = firstL ? 0.0 : 1.0;
        }
    } else {
        for (int it = 0; it <= n_batches; ++it) {
            const int par = it & 1;
            float head[8], tl[8];
            int nb_prev = 0;
            long b0 = 0;
            bool active = false, last = false;
            if (it >= 1) {  // ---- the backward transforms of batch `it - 1` out of half `par ^ 1` ----
                b0 = (long)(it - 1) * kRiNB;
                nb_prev = (int)(F - b0 < kRiNB ? F - b0 : kRiNB);
                const int fa = 2 * wk;
                active = fa < nb_prev;
                last = active && fa + 2 >= nb_prev;  // this wavefront holds the batch's last frame: its tail is the next batch's carry
                if (active) {
                    const f64x2 *rows = reinterpret_cast<const f64x2 *>(lds + oRiRows) + (long)(par ^ 1) * kRiNB * kYhStride;
                    // Hermitian extension of a row (istft_w64_kernel): register r of lane l <- bin l + 64 g + 256 k3; bins 0 / 511 / 512 / 513 by selects
                    auto load_row = [&](const f64x2 *rw, cd (&u)[16]) {
                        const cd y513 = ld(rw + 513);
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int g = r >> 2, k3 = r & 3;
                            if (k3 < 2) u[r] = ld(rw + (unsigned)(lane + 64 * g + 256 * k3));
                            else u[r] = conj(ld(rw + (unsigned)(64 * (3 - g) + 256 * (3 - k3) + 64 - lane)));
                        }
                        u[0].y = lane == 0 ? 0.0 : u[0].y;
                        const cd m513 = (y513 + u[2]) * 0.5, m511 = (u[13] + conj(y513)) * 0.5;
                        u[2].x = lane == 1 ? m513.x : u[2].x;
                        u[2].y = lane == 1 ? m513.y : lane == 0 ? 0.0 : u[2].y;
                        u[13].x = lane == 63 ? m511.x : u[13].x;
                        u[13].y = lane == 63 ? m511.y : u[13].y;
                    };
                    auto window = [&](const double (&x)[16], float (&o)[16]) {
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            const float f = (float)(x[j] * (1.0 / 1024.0));
                            o[j] = (float)((double)f * wrow[j]);
                        }
                    };
                    int t = fa;
                    const int t1 = fa + 2 < nb_prev ? fa + 2 : nb_prev;
                    bool first = true;
                    while (t < t1) {
                        bool pair = t + 1 < t1;
                        const f64x2 *ra = rows + t * kYhStride, *rb = pair ? ra + kYhStride : ra;
                        double re[16], im[16];
                        {
                            cd u[16];
                            load_row(ra, u);
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                re[r] = u[r].x;
                                im[r] = u[r].y;
                            }
                        }
                        BF_STAGE();
                        if (pair) {
                            cd v[16];
                            load_row(rb, v);
                            unsigned ha = 0, hb = 0;  // frames share a transform only if both are finite and of comparable scale (istft_w64_kernel)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                ha = max(ha, max(hi_abs(re[r]), hi_abs(im[r])));
                                hb = max(hb, max(hi_abs(v[r].x), hi_abs(v[r].y)));
                            }
                            ha = wave_max_u32(ha);
                            hb = wave_max_u32(hb);
                            const int ea = (int)(ha >> 20), eb = (int)(hb >> 20);
                            pair = ea < 0x7FF && eb < 0x7FF && ea - eb <= 20 && eb - ea <= 20;
                            if (pair) {
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    re[r] -= v[r].y;
                                    im[r] += v[r].x;
                                }
                            }
                        }
                        cx<double> tw[15];
                        BF_STAGE();
                        load_tw2<1, 16>(tw, s_tw2, lane);
                        BF_STAGE();
                        w64_inv_p3<double>(re, im);
                        w64_T2_any<false>(re, im, row16 - 16 * (lane >> 4), lane >> 4);
                        BF_STAGE();
                        mul_tw<true, 1, 16>(re, im, tw);
                        BF_STAGE();
                        load_tw1<1, 16>(tw, s_tw1, lane);
                        BF_STAGE();
                        fft16_core<double, +1, false>(re, im);
                        BF_STAGE();
                        T1_inv(re, im, row16, wcol);
                        BF_STAGE();
                        mul_tw<true, 1, 16>(re, im, tw);
                        fft16_core<double, +1, false>(re, im);
                        BF_STAGE();
                        float oa[16];
                        window(re, oa);
                        if (first) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) head[j] = oa[j];  // completed behind the barrier with the tail in front of it
                        } else {
                            float *yo = ys + (b0 + t) * kHop;
#pragma unroll
                            for (int j = 0; j < 8; ++j) yo[(unsigned)(64 * j + lane)] = tl[j] + oa[j];
                        }
                        if (pair) {
                            float ob[16];
                            window(im, ob);
                            float *yo = ys + (b0 + t + 1) * kHop;
#pragma unroll
                            for (int j = 0; j < 8; ++j) yo[(unsigned)(64 * j + lane)] = oa[j + 8] + ob[j];
#pragma unroll
                            for (int j = 0; j < 8; ++j) tl[j] = ob[j + 8];
                            t += 2;
                        } else {
#pragma unroll
                            for (int j = 0; j < 8; ++j) tl[j] = oa[j + 8];
                            t += 1;
                        }
                        first = false;
                    }
                    if (!last) {  // wavefront 4 with a second pair behind it: its tail meets wavefront 5's head behind the barrier
#pragma unroll
                        for (int j = 0; j < 8; ++j) s_tail[(par * 2 + 0) * 512 + 64 * j + lane] = tl[j];
                    }
                }
            }
            __syncthreads();
            if (active) {  // the hop in front of this wavefront's first frame: out_buff[0][j] + out_buff[1][j] as floats (util.h:302)
                const float *tin = wk == 0 ? s_tail + ((par ^ 1) * 2 + 1) * 512 : s_tail + (par * 2 + 0) * 512;
                float *yo = ys + (b0 + 2 * wk) * kHop;
#pragma unroll
                for (int j = 0; j < 8; ++j) yo[(unsigned)(64 * j + lane)] = tin[64 * j + lane] + head[j];
                if (last) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) s_tail[(par * 2 + 1) * 512 + 64 * j + lane] = tl[j];
                }
            }
        }
    }
    __syncthreads();
    {  // out_buff[0] of the next call: the carry the last batch left (written in iteration n_batches: parity n_batches & 1)
        const float *c = s_tail + ((n_batches & 1) * 2 + 1) * 512;
        for (int i = tid; i < kHop; i += kRiThreads) a.rec_tail_out[(long)s * kHop + i] = c[i];
    }
}
#endif

// ======================================================================================
//                  mcra node: single-channel MCRA noise subtraction (mcra.cpp:64-155)
// ======================================================================================
// One thread per (stream, problem), sequential over frames (S, S_min, S_tmp and lambda recurse over time).
// Only channel 0 is transformed (mcra.cpp:72-73), its pair partner is zero, so the packed spectrum IS X and
// the neighbouring bins of the 3-tap frequency smoothing are plain loads.
__global__ __launch_bounds__(64) void mcra_node_kernel(BinsArgs a) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.n_streams * kNQ) return;
    const int s = idx / kNQ, q = idx % kNQ;
    const int j = q_bin(q);
    double *sv = a.mpf + (long)s * (kMpfVecs * kN + 8);
    double Sprev = sv[0 * kN + j], Stmp = sv[1 * kN + j], Smin = sv[2 * kN + j], lam = sv[3 * kN + j];
    int cL = (int)sv[kMpfVecs * kN + 0];
    bool firstL = sv[kMpfVecs * kN + 1] == 0.0;  // stored inverted: a zeroed state is a cold start
    const f64x2 *Zs = a.Z + ((long)s * a.frames_ws + a.frame_off) * kN;
    f64x2 *row = a.Yh + ((long)s * a.n_frames) * kYhStride + q;
    const double aS = a.cfg.mcra_alphaS, aD = a.cfg.mcra_alphaD, aD2 = a.cfg.mcra_alphaD2, delta = a.cfg.mcra_delta;
    // the three spectrum values of a step are requested four frames ahead (as mpf_recursion_kernel): two wavefronts per SIMD at
    // 256 streams and a recursion per thread, nobody else covers the load latency
    constexpr int kAhead = 4;
    const int jl = j - 1 >= 1 ? j - 1 : j;  // bin 1 has no left neighbour inside [1, N): its slot re-reads bin j and is not used
    cd xr_[kAhead], xl_[kAhead], xc_[kAhead];
#pragma unroll
    for (int k = 0; k < kAhead; ++k) {
        const f64x2 *Zf = Zs + (k < a.n_frames ? k : a.n_frames - 1) * kN;
        xc_[k] = ld(Zf + j);
        xl_[k] = ld(Zf + jl);
        xr_[k] = ld(Zf + j + 1);
    }
    for (long t0 = 0; t0 < a.n_frames; t0 += kAhead) {
#pragma unroll
      for (int k = 0; k < kAhead; ++k) {
        const long t = t0 + k;
        if (t >= a.n_frames) break;
        const cd x = xc_[k], xl = xl_[k], xr = xr_[k];
        {
            const f64x2 *Zn = Zs + (t + kAhead < a.n_frames ? t + kAhead : a.n_frames - 1) * kN;
            xc_[k] = ld(Zn + j);
            xl_[k] = ld(Zn + jl);
            xr_[k] = ld(Zn + j + 1);
        }
        const double x2 = norm2(x);  // in_fft_square (mcra.cpp:77)
        double Sf;
        if (j == 0) {
            Sf = cabs(x);  // magnitude, not power (mcra.cpp:83)
        } else {           // 0.25 / 0.5 / 0.25 over bins j-1, j, j+1 inside [1, N) (mcra.cpp:84-92); j+1 <= 514 < N here
            Sf = 0.0;
            if (j - 1 >= 1) Sf += 0.25 * norm2(xl);
            Sf += 0.5 * x2;
            Sf += 0.25 * norm2(xr);
        }
        const double S = (aS * Sprev) + ((1 - aS) * Sf);
        if (cL > a.cfg.mcra_L) {  // mcra.cpp:100-113
            Smin = Stmp > S ? S : Stmp;
            Stmp = S;
            cL = 1;
            firstL = false;
        } else {
            Smin = Smin > S ? S : Smin;
            Stmp = Stmp > S ? S : Stmp;
            cL++;
        }
        {  // mcra.cpp:116-124 (as mpf_step: the division only while the first search period runs, the update by select)
            const bool upd = firstL || S < Smin * delta || lam > x2;
            double lam_new = aD2 * lam + (1.0 - aD) * x2;
            if (firstL) {
                const double invL = 1.0 / (double)cL;
                if (invL > aD) lam_new = invL * lam + (1.0 - invL) * x2;
            }
            lam = upd ? lam_new : lam;
        }
        cd y{0, 0};  // bin 0 is never written by the node (quirk Q16, mcra.cpp:127)
        if (j != 0) {
            double mag;
            if (a.cfg.out_only_noise) {
                mag = sqrt(lam) * a.cfg.out_amp;
            } else {
                mag = (cabs(x) - sqrt(lam)) * a.cfg.out_amp;
                if (mag < 0) mag = 0.0;
            }
            y = with_phase_of(mag, x);
        }
        row[t * kYhStride] = f64x2{y.x, y.y};
        Sprev = S;
      }
    }
    sv[0 * kN + j] = Sprev; sv[1 * kN + j] = Stmp; sv[2 * kN + j] = Smin; sv[3 * kN + j] = lam;
    if (q == 0) {
        sv[kMpfVecs * kN + 0] = (double)cL;
        sv[kMpfVecs * kN + 1] = firstL ? 0.0 : 1.0;
    }
}


#if BF_NFFT == 1024
// ======================================================================================
//            STFT + pointwise per-bin stage in one kernel (das fp64, phase, phasempf mask)
// ======================================================================================
// The nodes without a frame history need a frame's spectra exactly once, so they never have to reach HBM: a block transforms the
// microphone pairs of a frame into LDS -- 16 KB of packed pair spectrum per pair -- and then turns to the per-bin arithmetic of that
// frame, reading the spectra back from LDS through the same load_X / *_core functions the unfused kernels use.  HBM sees the input
// samples, the per-bin output rows (8 KB per frame) and nothing else: 2.3 GB instead of 11.6 GB per 65 536-frame batch at 8 microphones.
// (Round 2 / 3 ran this on the 32 x 32 half-wavefront transform with two frames per round between block barriers -- stft_bins_fused_kernel,
// 1.145 ms for phase against 1.03 for the team kernel below: EXPERIMENTS.md, round 4; removed in round 5.)
//
// On the 64-lane x 16-point transform (fft1024_w64.hpp, w64_f64_dev.hpp) with no block barrier in the loop: a 512-thread
// block is 8 / NPc TEAMS of NPc wavefronts; a team owns one frame at a time -- one full wavefront per microphone pair transforms it into
// the team's LDS slots (64 data registers per lane instead of 128: two wavefronts per SIMD), a team barrier (LDS counter), the team's
// threads run the per-bin stage of that frame out of LDS (the code above, bit-identical per bin; the transform's rounding differs from
// the 32 x 32 one at 1e-16), another team barrier.  Teams take frames from an LDS counter and drift apart, so one team's global loads and
// LDS round trips are covered by the others' arithmetic, and the wavefronts that win the issue arbitration of their SIMD (the older
// ones: das_f64_w64.hip) simply take more frames.  The exchange plane of a transform (16 x 65 doubles) lies inside the 16 KB slot that
// receives its spectrum; spectra are stored in natural bin order.  LDS: 17 KB twiddles + 128 KB slots + 9 KB window rows + counters.
typedef volatile __attribute__((address_space(3))) int *lds_cnt_t;
__device__ __forceinline__ int lds_cnt_add(lds_cnt_t p, int lane) {
    int old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add((__attribute__((address_space(3))) int *)p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __builtin_amdgcn_readfirstlane(old);
}
// all NPc wavefronts of a team arrive; LDS operations of a wavefront complete in issue order, so what it wrote before arriving is there
__device__ __forceinline__ void team_barrier(lds_cnt_t cnt, int target, int lane) {
    asm volatile("" ::: "memory");
    (void)lds_cnt_add(cnt, lane);
    while (*cnt < target) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

template <int LAYOUT, int MP, int ALGO>
__global__ __launch_bounds__(512) void stft_bins_w64_kernel(StftArgs a, BinsArgs b, long frames_per_block, long total_frames, double *aux,
                                                            f64x2 *xtail) {
    constexpr int NPc = MP / 2;            // wavefronts per team = pair slots per frame
    constexpr int NT = 8 / NPc;            // teams per block
    constexpr int TT = 64 * NPc;           // threads per team
    constexpr int kQMain = kN / 2;
    constexpr int NIT = kQMain / TT;       // (frame, bin) items per thread
    constexpr int kTwD = 2 * (1024 + 4 * kTw2RowW64Rot);
    constexpr int kWinRow = 18;            // [lane][j] rows of 16 doubles + 2 (das_f64_w64.hip)
    constexpr int oWin = kTwD + 8 * 2048, oCnt = oWin + 64 * kWinRow;
    __shared__ __attribute__((aligned(16))) double lds[oCnt + 10];  // 1 + 2 NT <= 17 counters
    const cx<double> *s_tw1 = reinterpret_cast<const cx<double> *>(lds);
    const cx<double> *s_tw2 = s_tw1 + 1024;
    lds_cnt_t s_next = (lds_cnt_t)(lds + oCnt);  // next frame of the block; then one arrival counter per team
    lds_cnt_t s_arr = s_next + 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = w / NPc, p = w % NPc, tt = tid - team * TT;  // thread index inside the team
    double *slot = lds + kTwD + w * 2048;  // this wavefront's spectrum [1024] c128; its head is the exchange plane while it transforms
    double *wcol = slot + w64_col_rot(lane);
    double *row16 = slot + (lane & 15) * kRS + 16 * (lane >> 4);
    const double *wrow = lds + oWin + lane * kWinRow;  // window[64 j + lane], j = 0..15
    {
        const f64x2 *tw2 = a.tw_w64;
        f64x2 *ltw = reinterpret_cast<f64x2 *>(lds);
        for (int i = tid; i < kTwD / 2; i += 512) ltw[i] = tw2[i];
        for (int i = tid; i < kN; i += 512) lds[oWin + (i & 63) * kWinRow + (i >> 6)] = a.win[i];
        if (tid <= NT) s_next[tid] = tid == 0 ? 2 * NT : 0;  // the first 2 NT frames are handed out statically (a team holds two: below)
    }
    const int M = a.n_mics, NP = (M + 1) >> 1;
    const bool has_pair = p < NP;
    const long gf0 = (long)blockIdx.x * frames_per_block;  // frames are numbered stream * n_frames + frame
    long gf1 = gf0 + frames_per_block;
    if (gf1 > total_frames) gf1 = total_frames;
    const int n_local = (int)(gf1 - gf0);

    float na[16], nb[16];  // raw samples of this wavefront's pair of the team's next frame: register j <- sample 64 j + lane of (hop t - 1 | hop t)
    auto request = [&](int lf) {
        const long gf = gf0 + lf;
        const int s = (int)(gf / a.n_frames);
        const long t = gf - (long)s * a.n_frames;
        if (!has_pair) return;
        const float *xs = a.x + (long)s * a.stream_stride_x;
        const float *hs = a.hist + (long)s * M * kHop;
        const int ma = 2 * p;
        const int mb = (2 * p + 1 < M) ? 2 * p + 1 : ma;
        if (LAYOUT == 0) {
            const float *a1 = t >= 1 ? xs + (long)ma * a.mic_stride + (t - 1) * kHop : hs + ma * kHop;
            const float *b1 = t >= 1 ? xs + (long)mb * a.mic_stride + (t - 1) * kHop : hs + mb * kHop;
            const float *a2 = xs + (long)ma * a.mic_stride + t * kHop;
            const float *b2 = xs + (long)mb * a.mic_stride + t * kHop;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                na[j] = a1[(unsigned)(64 * j + lane)];
                nb[j] = b1[(unsigned)(64 * j + lane)];
                na[j + 8] = a2[(unsigned)(64 * j + lane)];
                nb[j + 8] = b2[(unsigned)(64 * j + lane)];
            }
        } else {
            const float *s1 = t >= 1 ? xs + (t - 1) * (long)kHop * M : hs;
            const float *s2 = xs + t * (long)kHop * M;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                na[j] = s1[(unsigned)((64 * j + lane) * M + ma)];
                nb[j] = s1[(unsigned)((64 * j + lane) * M + mb)];
                na[j + 8] = s2[(unsigned)((64 * j + lane) * M + ma)];
                nb[j + 8] = s2[(unsigned)((64 * j + lane) * M + mb)];
            }
        }
    };
    // A team holds TWO frames: lf, whose samples are in na / nb, and lf_next, whose samples are requested as soon as the window stage has
    // consumed lf's (in flight during lf's transform AND its per-bin pass).  Until round 6 the request went out behind the team barrier,
    // in front of the per-bin pass -- whose steering loads return in order behind it: every frame's first item waited for the next frame's
    // samples to arrive from HBM (s_waitcnt vmcnt(0) a few hundred instructions behind 32 HBM loads).
    int lf = team, lf_next = NT + team;  // local indices; every wavefront of the team follows the same sequence
    if (lf < n_local) request(lf);
    __syncthreads();  // twiddles, window, counters
    int arrivals = 0;
    lds_cnt_t my_arr = s_arr + team;
    lds_cnt_t my_nxt = s_arr + NT + team;  // the frame after lf_next, published by the team's first wavefront
    while (lf < n_local) {
        const long gf = gf0 + lf;
        const int s = (int)(gf / a.n_frames);
        const long t = gf - (long)s * a.n_frames;
        // ---- pass 1: window + forward FFT of (frame, pair p) into this wavefront's slot -------------------------------------------
        if (has_pair) {
            double re[16], im[16];
            const bool b_ok = 2 * p + 1 < M;
            // buf[j]*hann_win[i] (util.h:235) and the first butterfly stage of the transform in one (das_f64_w64.hip)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double w0 = wrow[j], w1 = wrow[j + 8];
                const double t0 = (double)na[j] * w0, u = (double)na[j + 8];
                re[j] = fma(u, w1, t0);
                re[j + 8] = fma(-u, w1, t0);
            }
            if (b_ok) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const double w0 = wrow[j], w1 = wrow[j + 8];
                    const double t0 = (double)nb[j] * w0, u = (double)nb[j + 8];
                    im[j] = fma(u, w1, t0);
                    im[j + 8] = fma(-u, w1, t0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) im[j] = 0.0;
            }
            cx<double> tw[15];
            BF_STAGE();
            if (lf_next < n_local) request(lf_next);  // na / nb are consumed: the next frame's samples travel during this transform and the per-bin pass
            BF_STAGE();
            load_tw1<1, 9>(tw, s_tw1, lane);
            BF_STAGE();
            fft16_core<double, -1, true, 1>(re, im);  // stage 0 is done
            BF_STAGE();
            load_tw1<9, 16>(tw, s_tw1, lane);
            BF_STAGE();
            mul_tw<false, 1, 9>(re, im, tw);
            BF_STAGE();
            mul_tw<false, 9, 16>(re, im, tw);
            BF_STAGE();
            T1_fwd(re, im, wcol, row16);
            load_tw2<1, 9>(tw, s_tw2, lane);
            BF_STAGE();
            fft16_core<double, -1, true>(re, im);
            BF_STAGE();
            load_tw2<9, 16>(tw, s_tw2, lane);
            BF_STAGE();
            mul_tw<false, 1, 9>(re, im, tw);
            BF_STAGE();
            mul_tw<false, 9, 16>(re, im, tw);
            BF_STAGE();
            w64_T2<true>(re, im);
            w64_fwd_p3<double>(re, im);
            // register 4 g + k3 of lane l holds bin l + 64 g + 256 k3: natural order in the slot, one contiguous 1 KB row per store
            f64x2 *zo = reinterpret_cast<f64x2 *>(slot) + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) zo[64 * (r >> 2) + 256 * (r & 3)] = f64x2{re[r], im[r]};
        }
        // the frame after lf_next: taken by the team's first wavefront, read by the others behind the barrier
        if (p == 0) {
            const int nx = lds_cnt_add(s_next, lane);
            if (lane == 0) *my_nxt = nx;
        }
        arrivals += NPc;
        team_barrier(my_arr, arrivals, lane);
        const int lf_next2 = __builtin_amdgcn_readfirstlane(*my_nxt);
        // ---- pass 2: the per-bin stage of the frame, spectra read back from LDS -------------------------------------------------
        const f64x2 *zs = reinterpret_cast<const f64x2 *>(lds + kTwD) + (long)team * NPc * kN;
#pragma nounroll  // one item's registers at a time (two interleaved items spill 28 registers in the phasempf build)
        for (int n = 0; n < NIT; ++n) {
            const int q = tt + TT * n, j = q_bin(q);
            cd wst[MP];
            if (ALGO != BF_PHASE) load_steer<MP>(b.steer, j, M, wst);  // from L1 / L2, in flight while the spectra come out of LDS
            cd X[MP];
            load_X<MP>(zs, q, M, X);
            const long o = ((long)s * b.n_frames + t) * kYhStride + q;
            if (ALGO == BF_DAS) {
                const cd y = das_core<MP>(X, wst, M);
                st_y(b, o, q, y);
            } else if (ALGO == BF_PHASE) {
                const cd y = phase_core_lazy<MP>(X, b.steer, M, j, b.cfg);  // steering only where the magnitude gate is open
                st_y(b, o, q, y);
            } else {  // phasempf mask
                if (j == 0) {
                    b.Yh[o] = f64x2{X[0].x, X[0].y};
                    aux[o] = 0.0;
                } else {
                    cd soi;
                    double int2;
                    mpf_mask_core<MP>(X, wst, M, b.cfg, soi, int2);
                    b.Yh[o] = f64x2{soi.x, soi.y};
                    aux[o] = int2;
                }
            }
        }
        if (tt < 2) {  // bins N/2 and N/2+1 of the frame: X only (fused_tail_kernel finishes them)
            const int q = kQMain + tt;
            cd X[MP];
            load_X<MP>(zs, q, M, X);
            f64x2 *xt = xtail + (((long)s * b.n_frames + t) * 2 + tt) * MP;
#pragma unroll
            for (int m = 0; m < MP; ++m) xt[m] = f64x2{X[m].x, X[m].y};
        }
        arrivals += NPc;
        team_barrier(my_arr, arrivals, lane);  // the slots are rewritten by the next frame
        lf = lf_next;
        lf_next = lf_next2;
    }
}

#endif

// the two deferred problems per frame of the fused STFT + per-bin kernels: one thread per (stream, frame, q in {N/2, N/2+1})
template <int MP, int ALGO>
__global__ __launch_bounds__(256) void fused_tail_kernel(BinsArgs b, const f64x2 *xtail, double *aux) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)b.n_streams * b.n_frames * 2;
    if (idx >= total) return;
    const int q = kN / 2 + (int)(idx & 1), j = q_bin(q);
    const long sf = idx >> 1;  // stream * n_frames + frame
    cd X[MP], w[MP];
#pragma unroll
    for (int m = 0; m < MP; ++m) X[m] = ld(xtail + idx * MP + m);
    load_steer<MP>(b.steer, j, b.n_mics, w);
    const long o = sf * kYhStride + q;
    if (ALGO == BF_DAS) {
        const cd y = das_core<MP>(X, w, b.n_mics);
        st_y(b, o, q, y);
    } else if (ALGO == BF_PHASE) {
        const cd y = phase_core<MP>(X, w, b.n_mics, j, b.cfg);
        st_y(b, o, q, y);
    } else {
        cd soi;
        double int2;
        mpf_mask_core<MP>(X, w, b.n_mics, b.cfg, soi, int2);
        b.Yh[o] = f64x2{soi.x, soi.y};
        aux[o] = int2;
    }
}

#if BF_NFFT == 128 || BF_NFFT == 256 || BF_NFFT == 512
// ---- the same fusion at the JACK periods 64 / 128 / 256: stft_small_kernel's transform in front of the per-bin arithmetic ----------------------
// A half-wavefront transforms the G = 1024 / N frames of one group for one microphone pair (fft_small.hpp: N = 32 x NL, the frames side by side
// through one transpose plane) into its 16 KB LDS slot -- [g][N] packed pair spectra, the plane inside the slot while it transforms -- and the
// block then runs the per-bin stage of the 8 / NPc groups = FPR = (8 / NPc) G frames of the round out of LDS with the functions of the unfused
// kernels (bit-identical per bin).  HBM sees the samples, the per-bin rows and the two deferred bins per frame; the c128 spectra (1 ms of
// traffic each way per headline batch of samples at 8 microphones) stay on the CU.
template <int MP, typename ZT>
__device__ __forceinline__ void load_X_stride(const ZT *Zf, long pair_stride, int q, int M, cd (&X)[MP]) {  // bins_common.hpp load_X with the pairs `pair_stride` apart
    const int k = q_src_bin(q);
    const int kn = (kN - k) & (kN - 1);
#pragma unroll
    for (int p = 0; p < MP / 2; ++p) {
        if (2 * p < M) {
            const cd z = ld(Zf + p * pair_stride + k);
            const cd zc = conj(ld(Zf + p * pair_stride + kn));
            cd xa = (z + zc) * 0.5;
            const cd d = z - zc;
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

template <int LAYOUT, int MP, int ALGO>
__global__ __launch_bounds__(256, 1) void stft_bins_small_kernel(StftArgs a, BinsArgs b, long rounds_per_stream, long total_rounds,
                                                                 long rounds_per_block, double *aux, f64x2 *xtail) {
    constexpr int kNL = kN / 32, kG = 32 / kNL, kLogNL = kNL == 16 ? 4 : kNL == 8 ? 3 : 2;
    constexpr int NPc = MP / 2;        // pair slots per group of frames
    constexpr int FS = 8 / NPc;        // groups per round
    constexpr int FPR = FS * kG;       // frames per round
    constexpr int kQMain = kN / 2;     // problems 0 .. N/2-1 in the main passes; N/2 and N/2+1 go to fused_tail_kernel
    constexpr int NIT = FPR * kQMain / 256;
    __shared__ __attribute__((aligned(16))) double lds[2 * 32 * kNL + 8 * 2048 + kNL * 33];
    cx<double> *s_tw = reinterpret_cast<cx<double> *>(lds);  // [k1][n2] = W_N^(k1 n2)
    const int tid = threadIdx.x, lane = tid & 31, hw = tid >> 5;
    double *slot = lds + 2 * 32 * kNL + hw * 2048;  // this half-wavefront's spectra [g][N] c128; transpose plane while it transforms
    double *s_win = lds + 2 * 32 * kNL + 8 * 2048;  // [n2][j] = win[NL j + n2], rows of 33
    {
        for (int i = tid; i < 32 * kNL; i += 256) {
            const int m = ((i / kNL) * (i % kNL)) % kN;  // a.tw[m] = exp(-2 pi i m / N) for m < N / 2; W^(m + N/2) = -W^m
            const f64x2 w = a.tw[m % (kN / 2)];
            s_tw[i] = m < kN / 2 ? cx<double>{w.x, w.y} : cx<double>{-w.x, -w.y};
        }
        for (int i = tid; i < kN; i += 256) s_win[(i % kNL) * 33 + i / kNL] = a.win[i];
    }
    const int M = a.n_mics, NP = (M + 1) >> 1;
    const int fs = hw / NPc, p = hw % NPc;
    const bool has_pair = p < NP;
    const int gl = lane / kNL, n2 = lane % kNL;  // first pass: this lane's frame inside the group and its sample offset
    const double *hwin = s_win + n2 * 33;

    int it_f[NIT], it_q[NIT];  // the (frame of the round, problem) items of this thread and their steering entries: the same in every round
    cd st[NIT][MP];
#pragma unroll
    for (int n = 0; n < NIT; ++n) {
        const int idx = tid + 256 * n;
        it_f[n] = idx / kQMain;
        it_q[n] = idx % kQMain;
        load_steer<MP>(b.steer, q_bin(it_q[n]), M, st[n]);
    }

    const long r0 = (long)blockIdx.x * rounds_per_block;
    long r1 = r0 + rounds_per_block;
    if (r1 > total_rounds) r1 = total_rounds;

    float fr[32], fi[32];  // raw samples of this half-wavefront's next (group, pair): register j <- sample NL j + n2 of frame gl
    auto request = [&](long r) {
        const int s = (int)(r / rounds_per_stream);
        const long tg = (r % rounds_per_stream) * FPR + fs * kG;  // first frame of this half-wavefront's group
        if (!has_pair || tg >= a.n_frames) return;
        long t = tg + gl;
        if (t >= a.n_frames) t = a.n_frames - 1;  // past the end: the last frame again, never used
        const float *xs = a.x + (long)s * a.stream_stride_x;
        const float *hs = a.hist + (long)s * M * kHop;
        const int ma = 2 * p;
        const int mb = (2 * p + 1 < M) ? 2 * p + 1 : ma;
        if (LAYOUT == 0) {
            const float *a1 = (t >= 1 ? xs + (long)ma * a.mic_stride + (t - 1) * kHop : hs + ma * kHop) + n2;
            const float *b1 = (t >= 1 ? xs + (long)mb * a.mic_stride + (t - 1) * kHop : hs + mb * kHop) + n2;
            const float *a2 = xs + (long)ma * a.mic_stride + t * kHop + n2;
            const float *b2 = xs + (long)mb * a.mic_stride + t * kHop + n2;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                fr[j] = a1[kNL * j];
                fi[j] = b1[kNL * j];
                fr[j + 16] = a2[kNL * j];
                fi[j + 16] = b2[kNL * j];
            }
        } else {
            const float *s1 = (t >= 1 ? xs + (t - 1) * (long)kHop * M : hs) + (long)n2 * M;
            const float *s2 = xs + t * (long)kHop * M + (long)n2 * M;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                fr[j] = s1[(long)kNL * j * M + ma];
                fi[j] = s1[(long)kNL * j * M + mb];
                fr[j + 16] = s2[(long)kNL * j * M + ma];
                fi[j + 16] = s2[(long)kNL * j * M + mb];
            }
        }
    };
    if (r0 < r1) request(r0);
    __syncthreads();  // twiddles, window

    for (long r = r0; r < r1; ++r) {
        const int s = (int)(r / rounds_per_stream);
        const long f0 = (r % rounds_per_stream) * FPR;
        // ---- pass 1: window + forward transforms of (group fs, pair p) into this half-wavefront's slot -------------------------------------
        if (has_pair && f0 + fs * kG < a.n_frames) {
            double re[32], im[32];
            const double bs = (2 * p + 1 < M) ? 1.0 : 0.0;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                re[j] = (double)fr[j] * hwin[j];          // buf[j]*hann_win[i]  (util.h:235)
                im[j] = (double)fi[j] * (hwin[j] * bs);
            }
            fft32_dif<double, -1>(re, im);
#pragma unroll
            for (int i = 1; i < 32; ++i) {
                const cx<double> w = s_tw[brev5(i) * kNL + n2];
                const double xr = re[i], xi = im[i];
                re[i] = xr * w.x - xi * w.y;
                im[i] = xr * w.y + xi * w.x;
            }
#pragma unroll
            for (int i = 0; i < 32; ++i) slot[brev5(i) * kPSd + lane] = re[i];
            __builtin_amdgcn_wave_barrier();
            fft1024p_B<double>(re, lane, slot);
            __builtin_amdgcn_wave_barrier();
            fft1024p_C<double, false>(im, lane, slot);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int cc = 0; cc < 32; ++cc) im[cc] = slot[lane * kPSd + cc];
            __builtin_amdgcn_wave_barrier();
            fftn_dif_all<double, -1, kLogNL>(re, im);
            f64x2 *zo = reinterpret_cast<f64x2 *>(slot) + lane;  // lane = k1: position g NL + i' = bin k1 + 32 brev(i') of frame g
#pragma unroll
            for (int gg = 0; gg < kG; ++gg)
#pragma unroll
                for (int i = 0; i < kNL; ++i) zo[gg * kN + 32 * brevn(i, kLogNL)] = f64x2{re[gg * kNL + i], im[gg * kNL + i]};
        }
        __syncthreads();
        if (r + 1 < r1) request(r + 1);  // lands while the per-bin pass runs
        // ---- pass 2: the per-bin stage of the FPR frames, spectra read back from LDS ----------------------------------------------------------
        const f64x2 *zs = reinterpret_cast<const f64x2 *>(lds + 2 * 32 * kNL);
        auto frame_spectra = [&](int f) { return zs + (long)(f / kG) * NPc * 1024 + (f % kG) * kN; };  // pair p of that frame: + p * 1024
#pragma unroll
        for (int n = 0; n < NIT; ++n) {
            const int f = it_f[n];
            if (f0 + f >= a.n_frames) continue;
            const int q = it_q[n], j = q_bin(q);
            cd X[MP];
            load_X_stride<MP>(frame_spectra(f), 1024, q, M, X);
            const long o = ((long)s * b.n_frames + f0 + f) * kYhStride + q;
            if (ALGO == BF_DAS) {
                const cd y = das_core<MP>(X, st[n], M);
                st_y(b, o, q, y);
            } else if (ALGO == BF_PHASE) {
                const cd y = phase_core<MP>(X, st[n], M, j, b.cfg);
                st_y(b, o, q, y);
            } else {  // phasempf mask
                if (j == 0) {
                    b.Yh[o] = f64x2{X[0].x, X[0].y};
                    aux[o] = 0.0;
                } else {
                    cd soi;
                    double int2;
                    mpf_mask_core<MP>(X, st[n], M, b.cfg, soi, int2);
                    b.Yh[o] = f64x2{soi.x, soi.y};
                    aux[o] = int2;
                }
            }
        }
        if (tid < 2 * FPR && f0 + (tid >> 1) < a.n_frames) {  // bins N/2 and N/2+1 of every frame of the round: X only
            const int f = tid >> 1, q = kQMain + (tid & 1);
            cd X[MP];
            load_X_stride<MP>(frame_spectra(f), 1024, q, M, X);
            f64x2 *xt = xtail + (((long)s * b.n_frames + f0 + f) * 2 + (tid & 1)) * MP;
#pragma unroll
            for (int m = 0; m < MP; ++m) xt[m] = f64x2{X[m].x, X[m].y};
        }
        __syncthreads();  // the slots are rewritten by the next round
    }
}
#endif

#if BF_NFFT == 2048
// ---- the same fusion at the 1024-frame JACK period ------------------------------------------------------------------------------------------------
// X[k] = E[k] + W^k O[k], X[k + 1024] = E[k] - W^k O[k] with E, O = FFT-1024 of the even / odd samples (as round 4's stft_split_kernel had it).  Here the two transforms
// of a (frame, microphone pair) run on TWO half-wavefronts side by side, each into its own 16 KB LDS slot, and the radix-2 step is folded into
// the per-bin stage's spectrum read: eight half-wavefronts = the eight transforms of one frame at 8 microphones (two frames up to 4), 128 KB of
// slots + 16 KB of inter-pass twiddles; the window and W2048^k come through L1 (the LDS is full).  The c128 spectra (4.3 GB each way per headline
// batch of samples at 8 microphones) stay on the CU.
template <int MP, typename ZT>
__device__ __forceinline__ void load_X_split(const ZT *Zf, const f64x2 *w2048, int q, int M, cd (&X)[MP]) {  // Zf = [pair][E | O][1024]
    const int k = q_src_bin(q);
    const int kn = (kN - k) & (kN - 1);
    const int k1 = k & 1023, kn1 = kn & 1023;
    const cd wk = ld(w2048 + k1), wn = ld(w2048 + kn1);
    const double sk = k < 1024 ? 1.0 : -1.0, sn = kn < 1024 ? 1.0 : -1.0;
#pragma unroll
    for (int p = 0; p < MP / 2; ++p) {
        if (2 * p < M) {
            const cd z = ld(Zf + p * 2048 + k1) + (ld(Zf + p * 2048 + 1024 + k1) * wk) * sk;
            const cd zc = conj(ld(Zf + p * 2048 + kn1) + (ld(Zf + p * 2048 + 1024 + kn1) * wn) * sn);
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

template <int LAYOUT, int MP, int ALGO>
__global__ __launch_bounds__(256, 1) void stft_bins_split_kernel(StftArgs a, BinsArgs b, long rounds_per_stream, long total_rounds,
                                                                 long rounds_per_block, double *aux, f64x2 *xtail) {
    constexpr int NPc = MP / 2;           // pair slots per frame; two half-wavefronts (even / odd samples) per pair
    constexpr int FPR = 8 / (2 * NPc);    // frames per round
    constexpr int kQMain = kN / 2;        // problems 0 .. N/2-1 in the main passes; N/2 and N/2+1 go to fused_tail_kernel
    constexpr int NIT = FPR * kQMain / 256;
    __shared__ __attribute__((aligned(16))) double lds[2048 + 8 * 2048];
    cx<double> *s_tw = reinterpret_cast<cx<double> *>(lds);  // [k1][n2] = W1024^(k1 n2)
    const int tid = threadIdx.x, lane = tid & 31, hw = tid >> 5;
    double *slot = lds + 2048 + hw * 2048;  // this half-wavefront's E or O spectrum [1024] c128; transpose plane while it transforms
    for (int i = tid; i < 1024; i += 256) {
        const int m = (2 * (i >> 5) * (i & 31)) % kN;  // W1024^(k1 n2) = W2048^(2 k1 n2); a.tw[m] = W2048^m for m < 1024, W^(m + 1024) = -W^m
        const f64x2 w = a.tw[m % 1024];
        s_tw[i] = m < 1024 ? cx<double>{w.x, w.y} : cx<double>{-w.x, -w.y};
    }
    const int M = a.n_mics, NP = (M + 1) >> 1;
    const int fs = hw / (2 * NPc), p = (hw % (2 * NPc)) >> 1, eo = hw & 1;
    const bool has_pair = p < NP;

    int it_f[NIT], it_q[NIT];  // the (frame of the round, problem) items of this thread and their steering entries: the same in every round
    cd st[NIT][MP];
#pragma unroll
    for (int n = 0; n < NIT; ++n) {
        const int idx = tid + 256 * n;
        it_f[n] = idx / kQMain;
        it_q[n] = idx % kQMain;
        load_steer<MP>(b.steer, q_bin(it_q[n]), M, st[n]);
    }

    const long r0 = (long)blockIdx.x * rounds_per_block;
    long r1 = r0 + rounds_per_block;
    if (r1 > total_rounds) r1 = total_rounds;

    float fr[32], fi[32];  // raw samples of this half-wavefront's next (frame, pair): register j <- sample 2 (32 j + lane) + eo
    auto request = [&](long r) {
        const int s = (int)(r / rounds_per_stream);
        const long t = (r % rounds_per_stream) * FPR + fs;
        if (!has_pair || t >= a.n_frames) return;
        const float *xs = a.x + (long)s * a.stream_stride_x;
        const float *hs = a.hist + (long)s * M * kHop;
        const int ma = 2 * p;
        const int mb = (2 * p + 1 < M) ? 2 * p + 1 : ma;
        if (LAYOUT == 0) {
            const float *a1 = (t >= 1 ? xs + (long)ma * a.mic_stride + (t - 1) * kHop : hs + ma * kHop) + 2 * lane + eo;
            const float *b1 = (t >= 1 ? xs + (long)mb * a.mic_stride + (t - 1) * kHop : hs + mb * kHop) + 2 * lane + eo;
            const float *a2 = xs + (long)ma * a.mic_stride + t * kHop + 2 * lane + eo;
            const float *b2 = xs + (long)mb * a.mic_stride + t * kHop + 2 * lane + eo;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                fr[j] = a1[64 * j];
                fi[j] = b1[64 * j];
                fr[j + 16] = a2[64 * j];
                fi[j + 16] = b2[64 * j];
            }
        } else {
            const float *s1 = (t >= 1 ? xs + (t - 1) * (long)kHop * M : hs) + (long)(2 * lane + eo) * M;
            const float *s2 = xs + t * (long)kHop * M + (long)(2 * lane + eo) * M;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                fr[j] = s1[(long)64 * j * M + ma];
                fi[j] = s1[(long)64 * j * M + mb];
                fr[j + 16] = s2[(long)64 * j * M + ma];
                fi[j + 16] = s2[(long)64 * j * M + mb];
            }
        }
    };
    if (r0 < r1) request(r0);
    __syncthreads();  // twiddles

    for (long r = r0; r < r1; ++r) {
        const int s = (int)(r / rounds_per_stream);
        const long f0 = (r % rounds_per_stream) * FPR;
        // ---- pass 1: window + FFT-1024 of the even or the odd samples of (frame f0 + fs, pair p) into this half-wavefront's slot ---------------
        if (has_pair && f0 + fs < a.n_frames) {
            double re[32], im[32];
            const double bs = (2 * p + 1 < M) ? 1.0 : 0.0;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const double h = a.win[2 * (32 * j + lane) + eo];  // through L1: the LDS is full
                re[j] = (double)fr[j] * h;                           // buf[j]*hann_win[i]  (util.h:235)
                im[j] = (double)fi[j] * (h * bs);
            }
            fft1024p_fwd_A<double>(re, im, lane, s_tw, slot);
            __builtin_amdgcn_wave_barrier();
            fft1024p_B<double>(re, lane, slot);
            __builtin_amdgcn_wave_barrier();
            fft1024p_C<double, false>(im, lane, slot);
            __builtin_amdgcn_wave_barrier();
            fft1024p_D<double, -1>(re, im, lane, slot);
            __builtin_amdgcn_wave_barrier();
            f64x2 *zo = reinterpret_cast<f64x2 *>(slot) + lane;
#pragma unroll
            for (int i = 0; i < 32; ++i) zo[32 * brev5(i)] = f64x2{re[i], im[i]};
        }
        __syncthreads();
        if (r + 1 < r1) request(r + 1);  // lands while the per-bin pass runs
        // ---- pass 2: the per-bin stage of the FPR frames; X[k] = E[k mod 1024] +- W^(k mod 1024) O[k mod 1024] formed on the read ------------
        const f64x2 *zs = reinterpret_cast<const f64x2 *>(lds + 2048);
#pragma unroll
        for (int n = 0; n < NIT; ++n) {
            const int f = it_f[n];
            if (f0 + f >= a.n_frames) continue;
            const int q = it_q[n], j = q_bin(q);
            cd X[MP];
            load_X_split<MP>(zs + (long)f * NPc * 2048, a.tw, q, M, X);
            const long o = ((long)s * b.n_frames + f0 + f) * kYhStride + q;
            if (ALGO == BF_DAS) {
                const cd y = das_core<MP>(X, st[n], M);
                st_y(b, o, q, y);
            } else if (ALGO == BF_PHASE) {
                const cd y = phase_core<MP>(X, st[n], M, j, b.cfg);
                st_y(b, o, q, y);
            } else {  // phasempf mask
                if (j == 0) {
                    b.Yh[o] = f64x2{X[0].x, X[0].y};
                    aux[o] = 0.0;
                } else {
                    cd soi;
                    double int2;
                    mpf_mask_core<MP>(X, st[n], M, b.cfg, soi, int2);
                    b.Yh[o] = f64x2{soi.x, soi.y};
                    aux[o] = int2;
                }
            }
        }
        if (tid < 2 * FPR && f0 + (tid >> 1) < a.n_frames) {  // bins N/2 and N/2+1 of every frame of the round: X only
            const int f = tid >> 1, q = kQMain + (tid & 1);
            cd X[MP];
            load_X_split<MP>(zs + (long)f * NPc * 2048, a.tw, q, M, X);
            f64x2 *xt = xtail + (((long)s * b.n_frames + f0 + f) * 2 + (tid & 1)) * MP;
#pragma unroll
            for (int m = 0; m < MP; ++m) xt[m] = f64x2{X[m].x, X[m].y};
        }
        __syncthreads();  // the slots are rewritten by the next round
    }
}
#endif

}  // namespace

// stft + per-bin stage of the history-free nodes in one launch; hipErrorNotSupported = use the two-kernel chain
hipError_t launch_stft_bins_fused(const StftArgs &a, const BinsArgs &b, int n_cus, hipStream_t s) {
#if BF_NFFT == 1024
    const int algo = b.cfg.algo;
    if (!(algo == BF_DAS || algo == BF_PHASE || algo == BF_PHASEMPF)) return hipErrorNotSupported;
    if (a.n_mics > 8 || a.n_fft_mics != a.n_mics || b.n_dirs != 1 || a.frame_off != 0 || b.n_streams != a.n_streams)
        return hipErrorNotSupported;
    double *aux = reinterpret_cast<double *>(b.Yh + (long)b.n_streams * b.n_frames * kYhStride);
    f64x2 *xtail = a.Z;  // [stream][frame][2][MP]: the caller sizes the Z workspace for it (fused_tail_elems)
    if (!xtail) return hipErrorInvalidValue;
    const long tail_items = (long)b.n_streams * b.n_frames * 2;
    const unsigned tail_blocks = (unsigned)((tail_items + 255) / 256);
    if (a.tw_w64 == nullptr) return hipErrorNotSupported;
    const long wtotal = a.n_frames * a.n_streams;  // frames, numbered stream * n_frames + frame; one contiguous range per block
    long wblocks = wtotal < n_cus ? wtotal : n_cus;
    if (wblocks < 1) wblocks = 1;
    const long wfpb = (wtotal + wblocks - 1) / wblocks;
    wblocks = (wtotal + wfpb - 1) / wfpb;
#define BF_FUSED_GO(L_, MP_, A_)                                                                                             \
    do {                                                                                                                      \
        BF_LAUNCH((stft_bins_w64_kernel<L_, MP_, A_>), dim3((unsigned)wblocks), dim3(512), 0, s, a, b, wfpb, wtotal, aux, xtail);   \
        BF_LAUNCH((fused_tail_kernel<MP_, A_>), dim3(tail_blocks), dim3(256), 0, s, b, (const f64x2 *)xtail, aux);   \
    } while (0)
#define BF_FUSED_ALGO(L_, MP_)                                   \
    do {                                                          \
        if (algo == BF_DAS) BF_FUSED_GO(L_, MP_, BF_DAS);         \
        else if (algo == BF_PHASE) BF_FUSED_GO(L_, MP_, BF_PHASE); \
        else BF_FUSED_GO(L_, MP_, BF_PHASEMPF);                   \
    } while (0)
    if (a.layout == 0) {
        if (a.n_mics <= 4) BF_FUSED_ALGO(0, 4); else BF_FUSED_ALGO(0, 8);
    } else {
        if (a.n_mics <= 4) BF_FUSED_ALGO(1, 4); else BF_FUSED_ALGO(1, 8);
    }
#undef BF_FUSED_ALGO
#undef BF_FUSED_GO
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (algo == BF_PHASEMPF) {  // second pass of launch_phasempf: the recursion over frames (with many streams: and the backward transform)
        if (b.rec_istft) {
            BF_LAUNCH(mpf_rec_istft_kernel, dim3((unsigned)b.n_streams), dim3(kRiThreads), 0, s, b, aux);
        } else {
            const int nthr = b.n_streams * kNQ;
            BF_LAUNCH(mpf_recursion_kernel, dim3((nthr + 63) / 64), dim3(64), 0, s, b, aux);
        }
        e = hipGetLastError();
    }
    if (e == hipSuccess && b.spectrum) e = launch_expand_spectrum(b.Yh, b.spectrum, (long)b.n_streams * b.n_frames, s);
    return e;
#elif BF_NFFT == 128 || BF_NFFT == 256 || BF_NFFT == 512 || BF_NFFT == 2048
    const int algo = b.cfg.algo;
    if (!(algo == BF_DAS || algo == BF_PHASE || algo == BF_PHASEMPF)) return hipErrorNotSupported;
    if (a.n_mics > 8 || a.n_fft_mics != a.n_mics || b.n_dirs != 1 || a.frame_off != 0 || b.n_streams != a.n_streams)
        return hipErrorNotSupported;
#if BF_NFFT == 2048
    const int fpr = a.n_mics <= 4 ? 2 : 1;  // frames per round
#define BF_FUSED_KERNEL stft_bins_split_kernel
#else
    const int fpr = (a.n_mics <= 4 ? 4 : 2) * (1024 / kN);  // frames per round
#define BF_FUSED_KERNEL stft_bins_small_kernel
#endif
    const long rps = (a.n_frames + fpr - 1) / fpr;
    const long total = rps * a.n_streams;
    long blocks = total < n_cus ? total : n_cus;
    if (blocks < 1) blocks = 1;
    const long rpb = (total + blocks - 1) / blocks;
    blocks = (total + rpb - 1) / rpb;
    double *aux = reinterpret_cast<double *>(b.Yh + (long)b.n_streams * b.n_frames * kYhStride);
    f64x2 *xtail = a.Z;  // [stream][frame][2][MP]: the caller sizes the Z workspace for it
    if (!xtail) return hipErrorInvalidValue;
    const long tail_items = (long)b.n_streams * b.n_frames * 2;
    const unsigned tail_blocks = (unsigned)((tail_items + 255) / 256);
#define BF_FUSED_GO(L_, MP_, A_)                                                                                                                  \
    do {                                                                                                                                          \
        BF_LAUNCH((BF_FUSED_KERNEL<L_, MP_, A_>), dim3((unsigned)blocks), dim3(256), 0, s, a, b, rps, total, rpb, aux, xtail);           \
        BF_LAUNCH((fused_tail_kernel<MP_, A_>), dim3(tail_blocks), dim3(256), 0, s, b, (const f64x2 *)xtail, aux);                       \
    } while (0)
#define BF_FUSED_ALGO(L_, MP_)                                   \
    do {                                                          \
        if (algo == BF_DAS) BF_FUSED_GO(L_, MP_, BF_DAS);         \
        else if (algo == BF_PHASE) BF_FUSED_GO(L_, MP_, BF_PHASE); \
        else BF_FUSED_GO(L_, MP_, BF_PHASEMPF);                   \
    } while (0)
    if (a.layout == 0) {
        if (a.n_mics <= 4) BF_FUSED_ALGO(0, 4); else BF_FUSED_ALGO(0, 8);
    } else {
        if (a.n_mics <= 4) BF_FUSED_ALGO(1, 4); else BF_FUSED_ALGO(1, 8);
    }
#undef BF_FUSED_ALGO
#undef BF_FUSED_GO
#undef BF_FUSED_KERNEL
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (algo == BF_PHASEMPF) {  // second pass of launch_phasempf: the recursion over frames
        const int nthr = b.n_streams * kNQ;
        BF_LAUNCH(mpf_recursion_kernel, dim3((nthr + 63) / 64), dim3(64), 0, s, b, aux);
        e = hipGetLastError();
    }
    if (e == hipSuccess && b.spectrum) e = launch_expand_spectrum(b.Yh, b.spectrum, (long)b.n_streams * b.n_frames, s);
    return e;
#else
    (void)a; (void)b; (void)n_cus; (void)s;
    return hipErrorNotSupported;
#endif
}

hipError_t launch_phasempf(const BinsArgs &a, int n_cus, hipStream_t s) {
    // aux (|out_int|^2 per problem) lives behind the Yh rows: Yh was allocated with 2x room by the pipeline
    double *aux = reinterpret_cast<double *>(a.Yh + (long)a.n_streams * a.n_frames * kYhStride);
    const long total = (long)a.n_streams * a.n_frames * kNQ;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (a.n_mics <= 4)
        BF_LAUNCH((mpf_mask_kernel<4>), dim3(blocks), dim3(256), 0, s, a, aux);
    else if (a.n_mics <= 8)
        BF_LAUNCH((mpf_mask_kernel<8>), dim3(blocks), dim3(256), 0, s, a, aux);
    else if (a.n_mics <= 16)
        BF_LAUNCH((mpf_mask_kernel<16>), dim3(blocks), dim3(256), 0, s, a, aux);
    else
        BF_LAUNCH((mpf_mask_kernel<32>), dim3(blocks), dim3(256), 0, s, a, aux);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
#if BF_NFFT == 1024
    if (a.rec_istft) {
        BF_LAUNCH(mpf_rec_istft_kernel, dim3((unsigned)a.n_streams), dim3(kRiThreads), 0, s, a, aux);
        return hipGetLastError();
    }
#endif
    const int nthr = a.n_streams * kNQ;
    BF_LAUNCH(mpf_recursion_kernel, dim3((nthr + 63) / 64), dim3(64), 0, s, a, aux);
    return hipGetLastError();
}

template <int ALGO>
static void launch_pointwise_t(const BinsArgs &a, hipStream_t s) {
    const long total = (long)a.n_streams * a.n_frames * kNQ;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (a.n_mics <= 4)
        BF_LAUNCH((pointwise_bins_kernel<4, ALGO>), dim3(blocks), dim3(256), 0, s, a);
    else if (a.n_mics <= 8)
        BF_LAUNCH((pointwise_bins_kernel<8, ALGO>), dim3(blocks), dim3(256), 0, s, a);
    else if (a.n_mics <= 16)
        BF_LAUNCH((pointwise_bins_kernel<16, ALGO>), dim3(blocks), dim3(256), 0, s, a);
    else
        BF_LAUNCH((pointwise_bins_kernel<32, ALGO>), dim3(blocks), dim3(256), 0, s, a);
}

hipError_t launch_pointwise(const BinsArgs &a, hipStream_t s) {
    if (a.cfg.algo == BF_DAS)
        launch_pointwise_t<BF_DAS>(a, s);
    else
        launch_pointwise_t<BF_PHASE>(a, s);
    return hipGetLastError();
}

hipError_t launch_mcra_node(const BinsArgs &a, hipStream_t s) {
    BF_LAUNCH(mcra_node_kernel, dim3((a.n_streams * kNQ + 63) / 64), dim3(64), 0, s, a);
    return hipGetLastError();
}

}  // namespace BF_NTAG
}  // namespace bf
