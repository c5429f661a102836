// mask_kernels.hip -- the pointwise nodes (das in fp64, phase), phasempf (mask + MCRA/MPF recursion) and the mcra node.
#include "bins_common.hpp"

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

// das.cpp:60-63
template <int MP>
__device__ __forceinline__ cd das_bin(const BinCtx &c) {
    cd X[MP];
    load_X<MP>(c.Zf, c.q, c.M, X);
    const int j = q_bin(c.q);
    cd acc{0, 0};
#pragma unroll
    for (int m = 0; m < MP; ++m)
        if (m < c.M) acc = acc + conj(ld(c.steer + (long)m * kN + j)) * X[m];
    return cd{acc.x / (double)c.M, acc.y / (double)c.M};
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
    return tot / (double)num;  // 0/0 = NaN when M == 1, as the reference
}

// phase.cpp:87-127
template <int MP>
__device__ __forceinline__ cd phase_bin(const BinCtx &c, const bf_config &cfg) {
    cd X[MP];
    load_X<MP>(c.Zf, c.q, c.M, X);
    const int j = q_bin(c.q);
    if (j == 0) return X[0];
    double mag = 0.0;
#pragma unroll
    for (int m = 0; m < MP; ++m)
        if (m < c.M) mag += cabs(X[m]);
    mag /= (double)c.M;
    bool keep = false;
    if (mag / (double)kN > cfg.mag_threshold) {
        double ph[MP];
#pragma unroll
        for (int m = 0; m < MP; ++m) {
            if (m < c.M) {
                const cd u = conj(ld(c.steer + (long)m * kN + j)) * X[m];
                ph[m] = atan2(u.y, u.x);
            } else {
                ph[m] = 0.0;
            }
        }
        const double mean = pair_phase_mean<MP>(ph, c.M);
        keep = mean < cfg.min_phase * M_PI / 180;
    }
    if (!keep) mag *= cfg.mag_mult;
    return with_phase_of(mag, X[0]);  // mag * (cos, sin)(arg X_0)  (phase.cpp:115-122)
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
    a.Yh[((long)s * a.n_frames + t) * kYhStride + q] = f64x2{y.x, y.y};
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
    double ph[MP];
    double mag = 0.0;
#pragma unroll
    for (int m = 0; m < MP; ++m) {
        if (m < M) {
            const cd u = conj(ld(steer + (long)m * kN + j)) * X[m];
            ph[m] = atan2(u.y, u.x);
            mag += cabs(X[m]);
        } else {
            ph[m] = 0.0;
        }
    }
    const double mean = pair_phase_mean<MP>(ph, M);
    mag /= (double)M;
    const bool is_soi = mean < a.cfg.min_phase * M_PI / 180;
    const double lo = mag * a.cfg.min_mag;
    const double msoi = is_soi ? mag : lo, mint = is_soi ? lo : mag;
    const cd soi = with_phase_of(msoi, X[0]), in = with_phase_of(mint, X[0]);
    a.Yh[o] = f64x2{soi.x, soi.y};
    aux[o] = norm2(in);
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
    if (firstL || S < st.Smin * c.mcra_delta || st.lam > soi2) {
        const double ic = 1.0 / (double)cL;
        if (firstL && ic > c.mcra_alphaD)
            st.lam = ic * st.lam + (1.0 - ic) * soi2;
        else
            st.lam = c.mcra_alphaD2 * st.lam + (1.0 - c.mcra_alphaD) * soi2;  // quirk Q15g
    }
    st.Sprev = S;
    st.Z = c.mpf_alphaS * st.Z + (1 - c.mpf_alphaS) * int2;
    const double leak = c.mpf_eta * st.Z;
    const double kq = 1 - c.mpf_rev_gamma / c.mpf_rev_delta;  // quirk Q15i
    st.rev0 = c.mpf_rev_gamma * st.rev0 + kq * soi2;
    st.rev1 = c.mpf_rev_gamma * st.rev1 + kq * int2;
    const double Lam = sqrt(st.lam + leak + st.rev0 + st.rev1);
    if (j == 0) return cd{0, 0};  // quirk Q15d: y_fft[0] is never written; defined 0
    const double as = cabs(soi);
    double mg;
    if (c.out_only_noise) {
        mg = Lam * c.out_amp;
    } else {
        mg = (as - (c.out_only_mcra ? sqrt(st.lam) : Lam)) * c.out_amp;
        if (mg < 0) mg = c.noise_floor;
    }
    // mag * (cos, sin)(arg(soi)) == mag * soi/|soi|; arg(0) = 0
    if (as == 0.0) return cd{mg, 0.0};
    return cd{mg * (soi.x / as), mg * (soi.y / as)};
}

// Pass 2, one thread per (stream, problem), sequential over frames.
__global__ __launch_bounds__(64) void mpf_recursion_kernel(BinsArgs a, const double *aux) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.n_streams * kNQ) return;
    const int s = idx / kNQ, q = idx % kNQ;
    const int j = q_bin(q);
    double *sv = a.mpf + (long)s * (kMpfVecs * kN + 8);
    MpfState st{sv[0 * kN + j], sv[1 * kN + j], sv[2 * kN + j], sv[3 * kN + j], sv[4 * kN + j], sv[5 * kN + j], sv[6 * kN + j]};
    int cL = (int)sv[kMpfVecs * kN + 0];
    bool firstL = sv[kMpfVecs * kN + 1] == 0.0;  // stored as "first_L is over" flag so a zeroed state = cold start
    f64x2 *row = a.Yh + ((long)s * a.n_frames) * kYhStride + q;
    const double *arow = aux + ((long)s * a.n_frames) * kYhStride + q;
    for (long t = 0; t < a.n_frames; ++t) {
        const cd soi = ld(row + t * kYhStride);
        const double int2 = arow[t * kYhStride];
        const bool reset = cL > a.cfg.mcra_L;  // phasempf.cpp:161
        if (reset) {
            cL = 1;
            firstL = false;
        } else {
            cL++;
        }
        const cd y = mpf_step(st, a.cfg, j, soi, int2, reset, firstL, cL);
        row[t * kYhStride] = f64x2{y.x, y.y};
    }
    sv[0 * kN + j] = st.Sprev; sv[1 * kN + j] = st.Stmp; sv[2 * kN + j] = st.Smin; sv[3 * kN + j] = st.lam;
    sv[4 * kN + j] = st.Z; sv[5 * kN + j] = st.rev0; sv[6 * kN + j] = st.rev1;
    if (q == 0) {
        sv[kMpfVecs * kN + 0] = (double)cL;
        sv[kMpfVecs * kN + 1] = firstL ? 0.0 : 1.0;
    }
}

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
    for (long t = 0; t < a.n_frames; ++t) {
        const f64x2 *Zf = Zs + t * kN;
        const cd x = ld(Zf + j);
        const double x2 = norm2(x);  // in_fft_square (mcra.cpp:77)
        double Sf;
        if (j == 0) {
            Sf = cabs(x);  // magnitude, not power (mcra.cpp:83)
        } else {           // 0.25 / 0.5 / 0.25 over bins j-1, j, j+1 inside [1, N) (mcra.cpp:84-92); j+1 <= 514 < N here
            Sf = 0.0;
            if (j - 1 >= 1) Sf += 0.25 * norm2(ld(Zf + j - 1));
            Sf += 0.5 * x2;
            Sf += 0.25 * norm2(ld(Zf + j + 1));
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
        if (firstL || S < Smin * delta || lam > x2) {  // mcra.cpp:116-124
            const double invL = 1.0 / (double)cL;
            if (firstL && invL > aD)
                lam = invL * lam + (1.0 - invL) * x2;
            else
                lam = aD2 * lam + (1.0 - aD) * x2;
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
    sv[0 * kN + j] = Sprev; sv[1 * kN + j] = Stmp; sv[2 * kN + j] = Smin; sv[3 * kN + j] = lam;
    if (q == 0) {
        sv[kMpfVecs * kN + 0] = (double)cL;
        sv[kMpfVecs * kN + 1] = firstL ? 0.0 : 1.0;
    }
}

}  // namespace

hipError_t launch_phasempf(const BinsArgs &a, int n_cus, hipStream_t s) {
    // aux (|out_int|^2 per problem) lives behind the Yh rows: Yh was allocated with 2x room by the pipeline
    double *aux = reinterpret_cast<double *>(a.Yh + (long)a.n_streams * a.n_frames * kYhStride);
    const long total = (long)a.n_streams * a.n_frames * kNQ;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (a.n_mics <= 4)
        hipLaunchKernelGGL((mpf_mask_kernel<4>), dim3(blocks), dim3(256), 0, s, a, aux);
    else if (a.n_mics <= 8)
        hipLaunchKernelGGL((mpf_mask_kernel<8>), dim3(blocks), dim3(256), 0, s, a, aux);
    else if (a.n_mics <= 16)
        hipLaunchKernelGGL((mpf_mask_kernel<16>), dim3(blocks), dim3(256), 0, s, a, aux);
    else
        hipLaunchKernelGGL((mpf_mask_kernel<32>), dim3(blocks), dim3(256), 0, s, a, aux);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int nthr = a.n_streams * kNQ;
    hipLaunchKernelGGL(mpf_recursion_kernel, dim3((nthr + 63) / 64), dim3(64), 0, s, a, (const double *)aux);
    return hipGetLastError();
}

template <int ALGO>
static void launch_pointwise_t(const BinsArgs &a, hipStream_t s) {
    const long total = (long)a.n_streams * a.n_frames * kNQ;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (a.n_mics <= 4)
        hipLaunchKernelGGL((pointwise_bins_kernel<4, ALGO>), dim3(blocks), dim3(256), 0, s, a);
    else if (a.n_mics <= 8)
        hipLaunchKernelGGL((pointwise_bins_kernel<8, ALGO>), dim3(blocks), dim3(256), 0, s, a);
    else if (a.n_mics <= 16)
        hipLaunchKernelGGL((pointwise_bins_kernel<16, ALGO>), dim3(blocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((pointwise_bins_kernel<32, ALGO>), dim3(blocks), dim3(256), 0, s, a);
}

hipError_t launch_pointwise(const BinsArgs &a, hipStream_t s) {
    if (a.cfg.algo == BF_DAS)
        launch_pointwise_t<BF_DAS>(a, s);
    else
        launch_pointwise_t<BF_PHASE>(a, s);
    return hipGetLastError();
}

hipError_t launch_mcra_node(const BinsArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(mcra_node_kernel, dim3((a.n_streams * kNQ + 63) / 64), dim3(64), 0, s, a);
    return hipGetLastError();
}

}  // namespace BF_NTAG
}  // namespace bf
