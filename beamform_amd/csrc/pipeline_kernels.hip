// pipeline_kernels.hip -- gfx950 kernels of the fp64 bin pipeline (see pipeline.hip).
#include "pipeline_kernels.hpp"

#include <hip/hip_runtime.h>

#include <type_traits>

#include <cstdlib>

#include "fft1024.hpp"

namespace bf {

namespace {

constexpr int kHop = 512;
constexpr int kN = 1024;
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
__device__ __forceinline__ int q_src_bin(int q) { return q == 513 ? 511 : q; }
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
            if (q == 513) {
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

// ======================================================================================
//                                        STFT
// ======================================================================================
constexpr int kStftBlock = 256;
constexpr int kStftHalves = kStftBlock / 32;

template <int LAYOUT>
__global__ __launch_bounds__(kStftBlock) void stft_kernel(StftArgs a) {
    __shared__ __attribute__((aligned(16))) double lds[2048 + kStftHalves * 32 * kPSd + 32 * kPSd];
    const cx<double> *s_tw = reinterpret_cast<const cx<double> *>(lds);
    double *s_win = lds + 2048 + kStftHalves * 32 * kPSd;
    const int tid = threadIdx.x, lane = tid & 31, hw = tid >> 5;
    double *pbuf = lds + 2048 + hw * 32 * kPSd;
    {
        const double *twf = reinterpret_cast<const double *>(a.tw);
        for (int i = tid; i < 2048; i += kStftBlock) lds[i] = twf[i];
        for (int i = tid; i < kN; i += kStftBlock) s_win[(i & 31) * kPSd + (i >> 5)] = a.win[i];
    }
    __syncthreads();
    const int M = a.n_mics, MF = a.n_fft_mics, NP = (MF + 1) >> 1;
    const long total = (long)a.n_streams * a.n_frames * NP;
    const long stride = (long)gridDim.x * kStftHalves;
    const long rounds = (total + stride - 1) / stride;
    double re[32], im[32];
    for (long r = 0; r < rounds; ++r) {
        long item = r * stride + (long)blockIdx.x * kStftHalves + hw;
        const bool ok = item < total;
        if (!ok) item = total - 1;
        const int p = (int)(item % NP);
        const long st = item / NP;
        const long t = st % a.n_frames;
        const int s = (int)(st / a.n_frames);
        const float *xs = a.x + (long)s * a.stream_stride_x;
        const float *hs = a.hist + (long)s * M * kHop;
        const int ma = 2 * p;
        const bool b_ok = 2 * p + 1 < MF;
        const int mb = b_ok ? 2 * p + 1 : ma;
        if (LAYOUT == 0) {
            const float *a1 = (t >= 1 ? xs + (long)ma * a.mic_stride + (t - 1) * kHop : hs + ma * kHop) + lane;
            const float *b1 = (t >= 1 ? xs + (long)mb * a.mic_stride + (t - 1) * kHop : hs + mb * kHop) + lane;
            const float *a2 = xs + (long)ma * a.mic_stride + t * kHop + lane;
            const float *b2 = xs + (long)mb * a.mic_stride + t * kHop + lane;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                re[j] = (double)a1[32 * j];
                im[j] = (double)b1[32 * j];
                re[j + 16] = (double)a2[32 * j];
                im[j + 16] = (double)b2[32 * j];
            }
        } else {
            const float *s1 = (t >= 1 ? xs + (t - 1) * (long)kHop * M : hs) + (long)lane * M;
            const float *s2 = xs + t * (long)kHop * M + (long)lane * M;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                re[j] = (double)s1[(long)32 * j * M + ma];
                im[j] = (double)s1[(long)32 * j * M + mb];
                re[j + 16] = (double)s2[(long)32 * j * M + ma];
                im[j + 16] = (double)s2[(long)32 * j * M + mb];
            }
        }
        const double bs = b_ok ? 1.0 : 0.0;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const double h = s_win[lane * kPSd + j];
            re[j] *= h;          // buf[j]*hann_win[i]  (util.h:235)
            im[j] *= h * bs;
        }
        fft1024p_fwd_A<double>(re, im, lane, s_tw, pbuf);
        __builtin_amdgcn_wave_barrier();
        fft1024p_B<double>(re, lane, pbuf);
        __builtin_amdgcn_wave_barrier();
        fft1024p_C<double, false>(im, lane, pbuf);
        __builtin_amdgcn_wave_barrier();
        fft1024p_D<double, -1>(re, im, lane, pbuf);
        __builtin_amdgcn_wave_barrier();
        if (ok) {
            f64x2 *zo = a.Z + (((long)s * a.frames_ws + a.frame_off + t) * NP + p) * kN + lane;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int row = 32 * brev5(i);  // bins row .. row+31 of this store
                if (row > a.skip_lo && row + 31 < a.skip_hi) continue;  // band-limited nodes never read these bins
                zo[row] = f64x2{re[i], im[i]};
            }
        }
    }
}

// ======================================================================================
//                                        ISTFT
// ======================================================================================
constexpr int kIstftBlock = 256;
constexpr int kIstftHalves = kIstftBlock / 32;

// Hermitian part of y_fft at bin k (0..1023) from the per-bin kernels' output row.
__device__ __forceinline__ cd herm_at(const f64x2 *row, int k) {
    if (k == 0 || k == 512) return cd{row[k].x, 0.0};
    if (k == 511) {
        const cd u = ld(row + 511), v = conj(ld(row + 513));
        return (u + v) * 0.5;
    }
    if (k == 513) {
        const cd u = ld(row + 513), v = conj(ld(row + 511));
        return (u + v) * 0.5;
    }
    if (k < 512) return ld(row + k);
    return conj(ld(row + (kN - k)));
}

__global__ __launch_bounds__(kIstftBlock) void istft_kernel(IstftArgs a, int pairs_per_chunk, int chunks_per_stream) {
    __shared__ __attribute__((aligned(16))) double lds[2048 + kIstftHalves * 32 * kPSd + 32 * kPSd];
    const cx<double> *s_tw = reinterpret_cast<const cx<double> *>(lds);
    double *s_win = lds + 2048 + kIstftHalves * 32 * kPSd;
    const int tid = threadIdx.x, lane = tid & 31, hw = tid >> 5;
    double *pbuf = lds + 2048 + hw * 32 * kPSd;
    {
        const double *twf = reinterpret_cast<const double *>(a.tw);
        for (int i = tid; i < 2048; i += kIstftBlock) lds[i] = twf[i];
        for (int i = tid; i < kN; i += kIstftBlock) s_win[(i & 31) * kPSd + (i >> 5)] = a.win[i];
    }
    __syncthreads();
    const long chunk = (long)blockIdx.x * kIstftHalves + hw;
    int s = (int)(chunk / chunks_per_stream);
    const long c_in_s = chunk - (long)s * chunks_per_stream;
    const bool chunk_ok = s < a.n_streams;
    if (!chunk_ok) s = a.n_streams - 1;
    const long t0 = c_in_s * 2L * pairs_per_chunk;  // first frame of this run (even)
    const f64x2 *Ys = a.Yh + (long)s * a.n_frames * kYhStride;
    float *ys = a.y + (long)s * a.n_frames * kHop;

    float tail[16];  // second half of the previous frame, as float (out_buff[0], util.h:302)
#pragma unroll
    for (int q = 0; q < 16; ++q) tail[q] = 0.f;
    double re[32], im[32];

    for (int it = 0; it <= pairs_per_chunk; ++it) {
        const long ta = t0 - 2 + 2L * it;  // frames (ta, ta+1); it == 0 is the warm-up pair
        const bool va = ta >= 0 && ta < a.n_frames;
        const bool vb = ta + 1 >= 0 && ta + 1 < a.n_frames;
        const f64x2 *ra = Ys + (va ? ta : 0) * kYhStride;
        const f64x2 *rb = Ys + (vb ? ta + 1 : 0) * kYhStride;
        float oa[32], ob[32];
        // Two frames share one complex IFFT (Ya + i*Yb -> re = frame a, im = frame b).  A frame the
        // reference turns into NaN/Inf (mvdr/lcmv: inverse of an all-zero covariance) would
        // poison its partner through the shared transform, so such pairs are transformed one at a time.
        // Hermitian extension of both rows straight into Ya + i*Yb.  Position i holds bin k = lane + 32*brev5(i): even i
        // are bins < 512 (the stored row), odd i are bins >= 512 (conjugate of row[1024 - k]); only three positions touch
        // the irregular bins 0 / 511 / 512 / 513 (quirk Q1), so the rest is branch-free.
        auto load_pair = [&](bool useA, bool useB) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                constexpr int dummy = 0;
                (void)dummy;
                const int kb = 32 * brev5(i);
                cd u{0, 0}, v{0, 0};
                if (kb < 512) {
                    if (useA) u = ld(ra + kb + lane);
                    if (useB) v = ld(rb + kb + lane);
                } else {
                    if (useA) u = conj(ld(ra + (kN - kb) - lane));
                    if (useB) v = conj(ld(rb + (kN - kb) - lane));
                }
                if (i == 0 && lane == 0) {  // bin 0: real part only
                    u.y = 0.0;
                    v.y = 0.0;
                }
                if (i == 1) {  // bins 512 (lane 0: real part only) and 513 (lane 1: (Y[513] + conj Y[511]) / 2)
                    if (lane == 0) {
                        u.y = 0.0;
                        v.y = 0.0;
                    } else if (lane == 1) {
                        if (useA) u = (ld(ra + 513) + u) * 0.5;
                        if (useB) v = (ld(rb + 513) + v) * 0.5;
                    }
                }
                if (i == 30 && lane == 31) {  // bin 511: (Y[511] + conj Y[513]) / 2
                    if (useA) u = (u + conj(ld(ra + 513))) * 0.5;
                    if (useB) v = (v + conj(ld(rb + 513))) * 0.5;
                }
                re[i] = u.x - v.y;  // Ya + i*Yb
                im[i] = u.y + v.x;
            }
        };
        load_pair(va, vb);
        bool bad = false;  // a non-finite value in either frame makes the combination non-finite
#pragma unroll
        for (int i = 0; i < 32; ++i) bad = bad || !(isfinite(re[i]) && isfinite(im[i]));
        const bool split = __any(bad ? 1 : 0) != 0;
        for (int pass = 0; pass < (split ? 2 : 1); ++pass) {
            const bool useA = va && (!split || pass == 0);
            const bool useB = vb && (!split || pass == 1);
            if (split) load_pair(useA, useB);  // rare: one frame at a time
            fft1024p_inv_A<double>(re, im, lane, s_tw, pbuf);
            __builtin_amdgcn_wave_barrier();
            fft1024p_B<double>(re, lane, pbuf);
            __builtin_amdgcn_wave_barrier();
            fft1024p_C<double, true>(im, lane, pbuf);
            __builtin_amdgcn_wave_barrier();
            fft1024p_D<double, +1>(re, im, lane, pbuf);
            __builtin_amdgcn_wave_barrier();

            // position i: sample n = 32*brev5(i) + lane.  re -> frame ta, im -> frame ta+1.
            // overlap_and_add_prepare_output (util.h:247-252) with the reference's float stores.
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const double h = s_win[lane * kPSd + brev5(i)];
                float fa = (float)(re[i] / 1024.0);
                fa = (float)((double)fa * h);
                float fb = (float)(im[i] / 1024.0);
                fb = (float)((double)fb * h);
                if (a.use_post_amp) {  // mvdr.cpp:112-114
                    fa = (float)((double)fa * a.post_amp);
                    fb = (float)((double)fb * a.post_amp);
                }
                if (useA || (!va && pass == 0)) oa[i] = fa;
                if (useB || (!vb && pass == 0)) ob[i] = fb;
            }
        }
        const bool st_a = chunk_ok && it > 0 && va;
        const bool st_b = chunk_ok && it > 0 && vb;
        if (it == 0 && t0 == 0) {  // stream start: tail comes from the carried state, not from frame -1
            const float *ti = a.tail_in + (long)s * kHop + lane;
#pragma unroll
            for (int q = 0; q < 16; ++q) tail[q] = ti[32 * brev5(2 * q)];
        } else if (it == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) tail[q] = ob[2 * q + 1];
        }
        if (it > 0) {
            if (st_a) {
                float *yo = ys + ta * kHop + lane;
#pragma unroll
                for (int q = 0; q < 16; ++q) yo[32 * brev5(2 * q)] = tail[q] + oa[2 * q];
            }
            if (st_b) {
                float *yo = ys + (ta + 1) * kHop + lane;
#pragma unroll
                for (int q = 0; q < 16; ++q) yo[32 * brev5(2 * q)] = oa[2 * q + 1] + ob[2 * q];
            }
            if (st_a && ta == a.n_frames - 1) {  // odd frame count: the batch ends on frame a
                float *to = a.tail_out + (long)s * kHop + lane;
#pragma unroll
                for (int q = 0; q < 16; ++q) to[32 * brev5(2 * q)] = oa[2 * q + 1];
            }
            if (st_b && ta + 1 == a.n_frames - 1) {
                float *to = a.tail_out + (long)s * kHop + lane;
#pragma unroll
                for (int q = 0; q < 16; ++q) to[32 * brev5(2 * q)] = ob[2 * q + 1];
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) tail[q] = ob[2 * q + 1];
        }
    }
}

// full 1024-bin y_fft dump from the per-problem rows
__global__ void expand_spectrum_kernel(const f64x2 *Yh, f64x2 *out, long frames_total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= frames_total * kN) return;
    const long f = idx / kN;
    const int j = (int)(idx - f * kN);
    const f64x2 *row = Yh + f * kYhStride;
    f64x2 v;
    if (j <= 513) {
        v = row[j];
    } else {
        v = row[kN - j];
        v.y = -v.y;
    }
    out[idx] = v;
}

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
    if (mag / 1024.0 > cfg.mag_threshold) {
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

__global__ void smooth_kernel(const float *yraw, float *y, const double *state, long n, int n_streams, int sz) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * n_streams) return;
    const int s = (int)(idx / n);
    const long i = idx - (long)s * n;
    const float *yr = yraw + (long)s * n;
    const double *st = state + (long)s * 64;  // st[63] = most recent raw sample before this batch
    double acc = 0.0;
    for (int k = sz - 1; k >= 0; --k) {  // oldest first, as get_mean() sums past_samples[0..]
        const long src = i - k;
        const double v = src >= 0 ? (double)yr[src] : st[64 + src];
        acc += v;
    }
    y[idx] = (float)(acc / (double)sz);
}
__global__ void smooth_state_kernel(const float *yraw, double *state, long n, int n_streams) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 64 * n_streams) return;
    const int s = idx / 64, k = idx % 64;
    const long src = n - 64 + k;
    // n >= 512 always (one hop), so the new state is entirely inside this batch
    state[(long)s * 64 + k] = (double)yraw[(long)s * n + src];
}


// ======================================================================================
//                     mvdr / lcmv: covariance, Cholesky solve, constraints
// ======================================================================================
// One group of MP lanes per (stream, bin) problem, lane i <-> microphone i; the group walks a
// tile of consecutive frames so the sample covariance of the previous P frames
//   R = past_ffts[j] * past_ffts[j]^H                     (mvdr.cpp:87, lcmv.cpp:112)
// is slid by one rank-1 update and one downdate per frame (recomputed from scratch at the tile
// start).  The reference inverts R o whiteR with PartialPivLU and forms
//   mvdr:  w = R^-1 a / (a^H R^-1 a),   y = w^H x                     (mvdr.cpp:88-94)
//   lcmv:  W = R^-1 C (C^H R^-1 C)^-1,  y = W(:,0)^H x                (lcmv.cpp:113-119)
// R o whiteR is Hermitian positive definite whenever every mic has history energy, so with
// R = L L^H, U = L^-1 [C | x]:   G = U_C^H U_C,  g = U_C^H u_x,  y = (G^-1 g)_0
// (mvdr is the KP1 = 1 case: y = u_a^H u_x / u_a^H u_a).  Lane i owns row i of the
// factorisation; columns are exchanged through LDS (one wavefront executes its LDS operations
// in order, so only compiler barriers separate the phases).  A zero covariance (frame 0 of a
// cold start) yields 0 * inf = NaN, the same NaN frame the reference emits.
// ---- mvdr fast path: one thread per (stream, bin), whole problem in registers -----------------
// For M <= 8 the lower triangle of R (36 complex) and of its working copy fit the 512-entry
// register file of a wavefront that has a SIMD to itself (fp64 FMA issues every 4 cycles, so one
// wavefront per SIMD already keeps the fp64 pipe busy).  Lanes are consecutive bins: spectra
// loads are coalesced, no LDS, no idle lanes.  Same maths as mvdr_lcmv_kernel with KP1 = 1:
//   R o whiteR = L L^H,  u = L^-1 a,  v = L^-1 x,  y = u^H v / u^H u.
template <int MP>
__global__ __launch_bounds__(64, 1) void mvdr_fast_kernel(BinsArgs a, int tile, int tiles_per_stream) {
    constexpr int NT = MP * (MP + 1) / 2;
    const int q = blockIdx.y * 64 + threadIdx.x;
    const int s = blockIdx.x / tiles_per_stream;
    const long tA = (long)(blockIdx.x % tiles_per_stream) * tile;
    long tB = tA + tile;
    if (tB > a.n_frames) tB = a.n_frames;
    const int M = a.n_mics, NP = (M + 1) >> 1, P = a.cfg.past_windows;
    const bool live = q < kNQ;
    const int qq = live ? q : kNQ - 1;
    const int j = q_bin(qq);
    const double f = fabs(a.freqs[j]);
    const bool inband = live && f >= a.cfg.freq_min && f <= a.cfg.freq_max;
    f64x2 *yout = a.Yh + ((long)s * a.n_frames) * kYhStride + qq;
    const f64x2 *Zs = a.Z + ((long)(s / a.n_dirs) * a.frames_ws + a.frame_off) * NP * kN;
    const f64x2 *steer = a.steer + (long)(s % a.n_dirs) * a.steer_dir_stride;
    if (__builtin_amdgcn_ballot_w64(inband) == 0) {  // whole wavefront out of band (mvdr.cpp:103) or bin 0 (:76)
        if (live)
            for (long t = tA; t < tB; ++t) {
                cd y{0, 0};
                if (j == 0) {
                    cd X[MP];
                    load_X<MP>(Zs + t * NP * kN, qq, M, X);
                    y = X[0];
                }
                yout[t * kYhStride] = f64x2{y.x, y.y};
            }
        return;
    }
    cd st[MP];
#pragma unroll
    for (int m = 0; m < MP; ++m) st[m] = (m < M) ? ld(steer + (long)m * kN + j) : cd{0, 0};

    // Spectra are prefetched one frame ahead by global->LDS DMA (global_load_lds_dwordx4: no VGPRs, 1 KB per
    // instruction, lane l lands at row base + 16 l): a wavefront that owns its SIMD has nobody to hide HBM latency
    // behind, and the PMC profile of the register-only version showed 57 % of its cycles in s_waitcnt.
    // Row 2p / 2p+1 = Z_t[p][k] / Z_t[p][N-k]; rows MP + 2p, MP + 2p + 1 = the same of frame t - P (leaving the window).
    __shared__ __attribute__((aligned(16))) f64x2 s_pf[2][2 * MP][64];
    const int lane = threadIdx.x;
    const int ksrc = q_src_bin(qq), kneg = (kN - ksrc) & (kN - 1);
    auto dma_frame = [&](long t, bool with_old, int buf) {
        const f64x2 *Zn = Zs + t * NP * kN, *Zo = Zs + (t - P) * NP * kN;
#pragma unroll
        for (int p = 0; p < MP / 2; ++p)
            if (2 * p < M) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Zn + p * kN + ksrc),
                                                 (__attribute__((address_space(3))) void *)&s_pf[buf][2 * p][0], 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Zn + p * kN + kneg),
                                                 (__attribute__((address_space(3))) void *)&s_pf[buf][2 * p + 1][0], 16, 0, 0);
                if (with_old) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Zo + p * kN + ksrc),
                                                     (__attribute__((address_space(3))) void *)&s_pf[buf][MP + 2 * p][0], 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Zo + p * kN + kneg),
                                                     (__attribute__((address_space(3))) void *)&s_pf[buf][MP + 2 * p + 1][0], 16, 0, 0);
                }
            }
    };
    auto unpack = [&](int buf, int base, cd (&X)[MP]) {  // load_X out of the prefetched rows
#pragma unroll
        for (int p = 0; p < MP / 2; ++p) {
            if (2 * p < M) {
                const cd z = ld(&s_pf[buf][base + 2 * p][lane]);
                const cd zc = conj(ld(&s_pf[buf][base + 2 * p + 1][lane]));
                cd xa = (z + zc) * 0.5;
                const cd d = z - zc;
                cd xb = cd{0.5 * d.y, -0.5 * d.x};
                if (qq == 513) {
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
    };
#define BF_DMA_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

    cd R[NT];  // lower triangle, row-major: R[i*(i+1)/2 + c], c <= i
#pragma unroll
    for (int e = 0; e < NT; ++e) R[e] = cd{0, 0};
    int pb = 0;  // buffer the next consumer reads
    dma_frame(tA - 1, false, pb);
    for (int p = 1; p <= P; ++p) {  // covariance of the P frames in front of the tile
        BF_DMA_WAIT();
        __builtin_amdgcn_wave_barrier();
        if (p < P)
            dma_frame(tA - p - 1, false, pb ^ 1);
        else
            dma_frame(tA, true, pb ^ 1);  // first frame of the tile
        cd X[MP];
        unpack(pb, 0, X);
#pragma unroll
        for (int i = 0; i < MP; ++i)
#pragma unroll
            for (int c = 0; c <= i; ++c) R[i * (i + 1) / 2 + c] = cfma_conj(R[i * (i + 1) / 2 + c], X[i], X[c]);
        pb ^= 1;
    }
    for (long t = tA; t < tB; ++t) {
        BF_DMA_WAIT();  // frame t (and t - P) have landed in s_pf[pb]
        __builtin_amdgcn_wave_barrier();
        if (t + 1 < tB) dma_frame(t + 1, true, pb ^ 1);
        cd X[MP];
        unpack(pb, 0, X);
        double mag = 0.0;
#pragma unroll
        for (int m = 0; m < MP; ++m)
            if (m < M) mag += sqrt(norm2(X[m]));  // |X| well inside double range: no hypot scaling needed
        mag /= (double)((unsigned)M * 1024u);
        cd A[NT], ua[MP], ux[MP];
#pragma unroll
        for (int i = 0; i < MP; ++i) {
#pragma unroll
            for (int c = 0; c <= i; ++c) {
                cd v = R[i * (i + 1) / 2 + c];
                if (c == i) v = (i < M) ? v * 1.001 : cd{1.0, 0.0};  // whiteR diagonal (mvdr.cpp:239-243); padding = identity
                A[i * (i + 1) / 2 + c] = v;
            }
            ua[i] = st[i];
            ux[i] = X[i];
        }
#pragma unroll
        for (int jj = 0; jj < MP; ++jj) {
            const double inv = rsqrt(A[jj * (jj + 1) / 2 + jj].x);  // 1/L_jj; L_jj itself is never needed
            ua[jj] = ua[jj] * inv;
            ux[jj] = ux[jj] * inv;
#pragma unroll
            for (int i = jj + 1; i < MP; ++i) {
                const cd Lij = A[i * (i + 1) / 2 + jj] * inv;
                A[i * (i + 1) / 2 + jj] = Lij;
                ua[i] = cfms(ua[i], Lij, ua[jj]);
                ux[i] = cfms(ux[i], Lij, ux[jj]);
            }
#pragma unroll
            for (int c = jj + 1; c < MP; ++c) {
                const cd Lc = A[c * (c + 1) / 2 + jj];
#pragma unroll
                for (int i = c; i < MP; ++i)
                    A[i * (i + 1) / 2 + c] = cfms_conj(A[i * (i + 1) / 2 + c], A[i * (i + 1) / 2 + jj], Lc);
            }
        }
        cd num{0, 0};
        double den = 0.0;
#pragma unroll
        for (int i = 0; i < MP; ++i) {
            num = cfma_conj(num, ux[i], ua[i]);
            den += norm2(ua[i]);
        }
        cd y = cd{num.x / den, num.y / den};
        if (!(mag > a.cfg.freq_mag_threshold)) y = X[0] * 0.01;  // mvdr.cpp:96
        if (!inband) y = cd{0, 0};
        if (j == 0) y = X[0];
        if (live) yout[t * kYhStride] = f64x2{y.x, y.y};
        // slide the covariance window (mvdr.cpp:100-101)
        cd Xo[MP];
        unpack(pb, MP, Xo);
#pragma unroll
        for (int i = 0; i < MP; ++i)
#pragma unroll
            for (int c = 0; c <= i; ++c)
                R[i * (i + 1) / 2 + c] = cfms_conj(cfma_conj(R[i * (i + 1) / 2 + c], X[i], X[c]), Xo[i], Xo[c]);
        pb ^= 1;
    }
#undef BF_DMA_WAIT
}

template <int KM>
struct GramIdx {  // entries of the Hermitian upper triangle of G followed by g
    static constexpr int NG = KM * (KM + 1) / 2;
    static constexpr int NE = NG + KM;
};

template <int MP, int KM>
__global__ __launch_bounds__(256) void mvdr_lcmv_kernel(BinsArgs a, int tile, int tiles_per_stream) {
    constexpr int GPB = 256 / MP;          // problem groups per block
    constexpr int NB = KM + 1;             // right-hand sides: constraints + current frame
    constexpr int NE = GramIdx<KM>::NE;
    // +1 element of padding per row: the groups of a wavefront read the same [row][k] at the same time (broadcast
    // inside a group), and with a group stride that is a multiple of 128 B all of them would hit the same 4 banks
    // (PMC before the padding: SQ_LDS_BANK_CONFLICT = 1.7x SQ_ACTIVE_INST_LDS; lcmv 16-mic 37.5 -> 34.0 ms)
    __shared__ cd s_col[GPB][MP + 1];
    __shared__ cd s_x[GPB][MP + 1];
    __shared__ cd s_xo[GPB][MP + 1];
    __shared__ cd s_u[GPB][NB][MP + 1];
    __shared__ cd s_e[GPB][NE + 1];

    const int grp = threadIdx.x / MP;
    const int i = threadIdx.x % MP;
    const int q = blockIdx.y * GPB + grp;
    if (q >= kNQ) return;
    const int s = blockIdx.x / tiles_per_stream;
    const long tA = (long)(blockIdx.x % tiles_per_stream) * tile;
    long tB = tA + tile;
    if (tB > a.n_frames) tB = a.n_frames;
    const int M = a.n_mics, NP = (M + 1) >> 1, KP1 = a.kp1, P = a.cfg.past_windows;
    const int j = q_bin(q);
    const bool lcmv = a.cfg.algo == BF_LCMV;
    f64x2 *yout = a.Yh + ((long)s * a.n_frames) * kYhStride + q;
    const f64x2 *Zs = a.Z + ((long)(s / a.n_dirs) * a.frames_ws + a.frame_off) * NP * kN;  // frame 0 of this batch
    const f64x2 *steer = a.steer + (long)(s % a.n_dirs) * a.steer_dir_stride;

    // one microphone's spectrum at this problem's bin, frame t (may be negative: history)
    const int ksrc = q_src_bin(q), kneg = (kN - ksrc) & (kN - 1);
    auto load_xi = [&](long t) -> cd {
        if (i >= M) return cd{0, 0};
        const f64x2 *Zf = Zs + t * NP * kN + (i >> 1) * kN;
        const cd z = ld(Zf + ksrc), zc = conj(ld(Zf + kneg));
        cd x;
        if ((i & 1) == 0) {
            x = (z + zc) * 0.5;
        } else {
            const cd d = z - zc;
            x = cd{0.5 * d.y, -0.5 * d.x};
        }
        return q == 513 ? conj(x) : x;
    };

    const double f = fabs(a.freqs[j]);
    const bool inband = f >= a.cfg.freq_min && f <= a.cfg.freq_max;
    if (!inband || (!lcmv && j == 0)) {
        // mvdr.cpp:76 y_fft[0] = in_fft(0,0); out of band: y_fft[j] = 0 (mvdr.cpp:103)
        for (long t = tA; t < tB; ++t) {
            cd y{0, 0};
            if (!lcmv && j == 0) {
                const cd x = load_xi(t);
                s_x[grp][i] = x;
                __builtin_amdgcn_wave_barrier();
                y = s_x[grp][0];
                __builtin_amdgcn_wave_barrier();
            }
            if (i == 0) yout[t * kYhStride] = f64x2{y.x, y.y};
        }
        return;
    }

    cd cst[KM];  // this mic's entries of the constraint columns (weights[j](i, r))
#pragma unroll
    for (int r = 0; r < KM; ++r)
        cst[r] = (r < KP1 && i < M) ? ld(steer + ((long)r * M + i) * kN + j) : cd{0, 0};

    // R row i (lower triangle c <= i is what the factorisation reads)
    cd R[MP];
#pragma unroll
    for (int c = 0; c < MP; ++c) R[c] = cd{0, 0};
    for (int p = 1; p <= P; ++p) {
        const cd x = load_xi(tA - p);
        s_x[grp][i] = x;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < MP; ++c) R[c] = cfma_conj(R[c], x, s_x[grp][c]);
        __builtin_amdgcn_wave_barrier();
    }

    for (long t = tA; t < tB; ++t) {
        const cd x = load_xi(t);
        const cd xo = load_xi(t - P);
        s_x[grp][i] = x;
        s_xo[grp][i] = xo;
        __builtin_amdgcn_wave_barrier();
        double mag = 0.0;
        for (int m = 0; m < M; ++m) mag += sqrt(norm2(s_x[grp][m]));
        mag /= (double)((unsigned)M * 1024u);
        cd y;
        if (mag > a.cfg.freq_mag_threshold) {
            cd A[MP], b[NB];
#pragma unroll
            for (int c = 0; c < MP; ++c) A[c] = R[c];
            if (i < M) {
                // cwiseProduct(whiteR): diagonal * 1.001 (mvdr.cpp:239-243)
#pragma unroll
                for (int c = 0; c < MP; ++c)
                    if (c == i) A[c] = A[c] * 1.001;
            } else {
#pragma unroll
                for (int c = 0; c < MP; ++c) A[c] = cd{c == i ? 1.0 : 0.0, 0.0};
            }
#pragma unroll
            for (int r = 0; r < KM; ++r) b[r] = cst[r];
            b[KM] = x;
#pragma unroll
            for (int jj = 0; jj < MP; ++jj) {
                s_col[grp][i] = A[jj];  // raw column jj, row i
                if (i == jj) {
#pragma unroll
                    for (int r = 0; r < NB; ++r) s_u[grp][r][0] = b[r];
                }
                __builtin_amdgcn_wave_barrier();
                const double inv = rsqrt(s_col[grp][jj].x);  // 1 / L_jj
                const cd Lij = A[jj] * inv;
                if (i > jj) {
#pragma unroll
                    for (int c = jj + 1; c < MP; ++c)
                        if (c <= i) A[c] = cfms_conj(A[c], Lij, s_col[grp][c] * inv);
#pragma unroll
                    for (int r = 0; r < NB; ++r) b[r] = cfms(b[r], Lij, s_u[grp][r][0] * inv);
                } else if (i == jj) {
#pragma unroll
                    for (int r = 0; r < NB; ++r) b[r] = b[r] * inv;
                }
                __builtin_amdgcn_wave_barrier();
            }
            // b[r] is now row i of U = L^-1 [C | x]
#pragma unroll
            for (int r = 0; r < NB; ++r) s_u[grp][r][i] = (i < M) ? b[r] : cd{0, 0};
            __builtin_amdgcn_wave_barrier();
            // Gram entries: e < NG: G(r,r2) with r <= r2; e >= NG: g(r)
            for (int e = i; e < NE; e += MP) {
                int r = 0, r2 = 0;
                if (e < GramIdx<KM>::NG) {
                    int rem = e;
                    while (rem >= KM - r) { rem -= KM - r; ++r; }
                    r2 = r + rem;
                } else {
                    r = e - GramIdx<KM>::NG;
                    r2 = KM;
                }
                cd acc{0, 0};
                for (int m = 0; m < M; ++m) acc = cfma_conj(acc, s_u[grp][r2][m], s_u[grp][r][m]);
                s_e[grp][e] = acc;
            }
            __builtin_amdgcn_wave_barrier();
            // every lane solves the (KP1 x KP1) system G y = g redundantly; padding rows are identity
            cd Gm[KM][KM], gv[KM];
            {
                int e = 0;
#pragma unroll
                for (int r = 0; r < KM; ++r)
#pragma unroll
                    for (int r2 = r; r2 < KM; ++r2) {
                        const cd v = s_e[grp][e++];
                        Gm[r][r2] = v;
                        Gm[r2][r] = conj(v);
                    }
#pragma unroll
                for (int r = 0; r < KM; ++r) gv[r] = s_e[grp][GramIdx<KM>::NG + r];
#pragma unroll
                for (int r = 0; r < KM; ++r)
                    if (r >= KP1) {
#pragma unroll
                        for (int r2 = 0; r2 < KM; ++r2) {
                            Gm[r][r2] = cd{r == r2 ? 1.0 : 0.0, 0.0};
                            Gm[r2][r] = cd{r == r2 ? 1.0 : 0.0, 0.0};
                        }
                        gv[r] = cd{0, 0};
                    }
            }
#pragma unroll
            for (int k = 0; k < KM; ++k) {  // Gaussian elimination (G is Hermitian positive definite)
                const cd pinv = cdiv(cd{1, 0}, Gm[k][k]);
#pragma unroll
                for (int r = k + 1; r < KM; ++r) {
                    const cd fct = Gm[r][k] * pinv;
#pragma unroll
                    for (int c = k + 1; c < KM; ++c) Gm[r][c] = Gm[r][c] - fct * Gm[k][c];
                    gv[r] = gv[r] - fct * gv[k];
                }
            }
#pragma unroll
            for (int k = KM - 1; k >= 0; --k) {
                cd acc = gv[k];
#pragma unroll
                for (int c = k + 1; c < KM; ++c) acc = acc - Gm[k][c] * gv[c];
                gv[k] = cdiv(acc, Gm[k][k]);
            }
            y = gv[0];
        } else {
            y = s_x[grp][0] * 0.01;  // in_fft(0,j)*0.01 (mvdr.cpp:96)
        }
        if (i == 0) yout[t * kYhStride] = f64x2{y.x, y.y};
        // slide the covariance window: + x_t x_t^H - x_{t-P} x_{t-P}^H (mvdr.cpp:100-101)
#pragma unroll
        for (int c = 0; c < MP; ++c) R[c] = cfms_conj(cfma_conj(R[c], x, s_x[grp][c]), xo, s_xo[grp][c]);
        __builtin_amdgcn_wave_barrier();
    }
}


// ---- mvdr / lcmv "lanes" kernel: L lanes per problem, rows dealt cyclically, exchange by DPP -------------
// For 9..16 microphones (and lcmv with up to 8) a problem does not fit one lane's registers.  Instead of one lane per
// row (mvdr_lcmv_kernel: 16 lanes per problem, half of them idle on average, every column through LDS) lane q of an
// L-lane group (L = 4 for M <= 16, 2 for M <= 8; a group never straddles a quad) owns rows i = r*L + q of R and of its
// Cholesky factor.  Cyclic rows keep every lane busy until the last column, the column / pivot / right-hand-side
// exchange is a quad_perm DPP broadcast, and a wavefront carries 64/L problems.  R (its stored rows: 40 complex per
// lane at M = 16) lives in LDS, lane-contiguous, so the working copy and the right-hand sides fit the register file of
// a wavefront that owns its SIMD.  Maths identical to mvdr_lcmv_kernel:
//   R o whiteR = L L^H,  U = L^-1 [C | x],  G = U_C^H U_C,  g = U_C^H u_x,  y = (G^-1 g)_0.
template <int L>
struct LaneGrp;
template <>
struct LaneGrp<4> {
    template <int SRC>
    static __device__ __forceinline__ int bc(int v) { return __builtin_amdgcn_update_dpp(v, v, SRC * 0x55, 0xF, 0xF, false); }
    static __device__ __forceinline__ int x1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false); }
    static __device__ __forceinline__ int x2(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false); }
};
template <>
struct LaneGrp<2> {
    template <int SRC>
    static __device__ __forceinline__ int bc(int v) {
        return __builtin_amdgcn_update_dpp(v, v, SRC | (SRC << 2) | ((2 + SRC) << 4) | ((2 + SRC) << 6), 0xF, 0xF, false);
    }
    static __device__ __forceinline__ int x1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false); }
    static __device__ __forceinline__ int x2(int v) { return v; }
};
template <int L, int SRC>
__device__ __forceinline__ double bcast_d(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = LaneGrp<L>::template bc<SRC>((int)(b & 0xffffffffLL)), hi = LaneGrp<L>::template bc<SRC>((int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <int L>
__device__ __forceinline__ double bcast_from(int src, double v) {  // src is a compile-time constant after unrolling
    if (L == 2) return src == 0 ? bcast_d<L, 0>(v) : bcast_d<L, 1>(v);
    return src == 0 ? bcast_d<L, 0>(v) : src == 1 ? bcast_d<L, 1>(v) : src == 2 ? bcast_d<L, 2>(v) : bcast_d<L, 3>(v);
}
template <int L>
__device__ __forceinline__ cd bcast_from(int src, cd v) { return cd{bcast_from<L>(src, v.x), bcast_from<L>(src, v.y)}; }
template <int L>
__device__ __forceinline__ double grp_sum(double v) {
    auto sh = [](double x, bool second) {
        const long long b = __builtin_bit_cast(long long, x);
        const int lo = second ? LaneGrp<L>::x2((int)(b & 0xffffffffLL)) : LaneGrp<L>::x1((int)(b & 0xffffffffLL));
        const int hi = second ? LaneGrp<L>::x2((int)(b >> 32)) : LaneGrp<L>::x1((int)(b >> 32));
        return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
    };
    v += sh(v, false);
    if (L == 4) v += sh(v, true);
    return v;
}

template <int MP, int L, int KM>
__global__ __launch_bounds__(64, 1) void mvdr_lcmv_lanes_kernel(BinsArgs a, int tile, int tiles_per_stream) {
    constexpr int RPL = MP / L;               // rows per lane
    constexpr int PPW = 64 / L;               // problems per wavefront
    constexpr int NB = KM + 1;
    constexpr int NG = GramIdx<KM>::NG, NE = GramIdx<KM>::NE;
    constexpr int NT = L * RPL * (RPL + 1) / 2;  // stored entries per lane: slot r keeps columns 0 .. r*L+L-1
#define TIX(r, c) (L * (r) * ((r) + 1) / 2 + (c))
    __shared__ __attribute__((aligned(16))) f64x2 s_R[NT][64];  // R rows of this lane: s_R[TIX(r, c)][lane]
    const int lane = threadIdx.x;
    const int q = lane % L;
    const int pq = blockIdx.y * PPW + lane / L;  // problem (bin) index
    const int s = blockIdx.x / tiles_per_stream;
    const long tA = (long)(blockIdx.x % tiles_per_stream) * tile;
    long tB = tA + tile;
    if (tB > a.n_frames) tB = a.n_frames;
    const int M = a.n_mics, NP = (M + 1) >> 1, KP1 = a.kp1, P = a.cfg.past_windows;
    const bool live = pq < kNQ;
    const int qq = live ? pq : kNQ - 1;
    const int j = q_bin(qq);
    const bool lcmv = a.cfg.algo == BF_LCMV;
    const double f = fabs(a.freqs[j]);
    const bool inband = live && f >= a.cfg.freq_min && f <= a.cfg.freq_max && !(j == 0 && !lcmv);
    f64x2 *yout = a.Yh + ((long)s * a.n_frames) * kYhStride + qq;
    const f64x2 *Zs = a.Z + ((long)(s / a.n_dirs) * a.frames_ws + a.frame_off) * NP * kN;
    const f64x2 *steer = a.steer + (long)(s % a.n_dirs) * a.steer_dir_stride;
    const int ksrc = q_src_bin(qq), kneg = (kN - ksrc) & (kN - 1);
    auto load_mic = [&](long t, int m) -> cd {  // spectrum of microphone m at this bin, frame t
        if (m >= M) return cd{0, 0};
        const f64x2 *Zf = Zs + t * NP * kN + (m >> 1) * kN;
        const cd z = ld(Zf + ksrc), zc = conj(ld(Zf + kneg));
        cd x;
        if ((m & 1) == 0) {
            x = (z + zc) * 0.5;
        } else {
            const cd d = z - zc;
            x = cd{0.5 * d.y, -0.5 * d.x};
        }
        return qq == 513 ? conj(x) : x;
    };
    if (__builtin_amdgcn_ballot_w64(inband) == 0) {  // nothing to solve in this wavefront
        if (live && q == 0)
            for (long t = tA; t < tB; ++t) {
                cd y{0, 0};
                if (j == 0 && !lcmv) y = load_mic(t, 0);  // mvdr.cpp:76
                yout[t * kYhStride] = f64x2{y.x, y.y};
            }
        return;
    }
    // this lane's entries of the constraint columns (weights[j](i, c)) are re-read every frame (L2-resident table)
    // straight into the right-hand sides: keeping them would cost 64 more registers at M = 16, K + 1 = 4
    auto load_cst = [&](int r, int c) -> cd {
        const int i = r * L + q;
        return (c < KP1 && i < M) ? ld(steer + ((long)c * M + i) * kN + j) : cd{0, 0};
    };
#pragma unroll
    for (int e = 0; e < NT; ++e) s_R[e][lane] = f64x2{0, 0};
    // R[i][c] += x_i conj(x_c) (- xo_i conj(xo_c)) on the stored rows; x_c comes from its owner lane by DPP
    auto rank1 = [&](const cd (&xl)[RPL], const cd (&xo)[RPL], bool with_old) {
#pragma unroll
        for (int c = 0; c < MP; ++c) {
            const cd xc = bcast_from<L>(c % L, xl[c / L]);
            const cd xoc = with_old ? bcast_from<L>(c % L, xo[c / L]) : cd{0, 0};
#pragma unroll
            for (int r = c / L; r < RPL; ++r) {
                const int i = r * L + q;
                if (c <= i) {
                    cd v = ld(&s_R[TIX(r, c)][lane]);
                    v = cfma_conj(v, xl[r], xc);
                    if (with_old) v = cfms_conj(v, xo[r], xoc);
                    s_R[TIX(r, c)][lane] = f64x2{v.x, v.y};
                }
            }
        }
    };
    for (int p = 1; p <= P; ++p) {
        cd xl[RPL];
#pragma unroll
        for (int r = 0; r < RPL; ++r) xl[r] = load_mic(tA - p, r * L + q);
        rank1(xl, xl, false);
    }

    for (long t = tA; t < tB; ++t) {
        cd xl[RPL], xo[RPL];
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            xl[r] = load_mic(t, r * L + q);
            xo[r] = load_mic(t - P, r * L + q);
        }
        double mag = 0.0;
#pragma unroll
        for (int r = 0; r < RPL; ++r) mag += sqrt(norm2(xl[r]));  // padded rows are 0
        mag = grp_sum<L>(mag) / (double)((unsigned)M * 1024u);
        const cd x0 = bcast_from<L>(0, xl[0]);
        cd y;
        if (mag > a.cfg.freq_mag_threshold) {
            cd A[NT], b[RPL][NB];
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
#pragma unroll
                for (int c = 0; c < KM; ++c) b[r][c] = load_cst(r, c);
                b[r][KM] = xl[r];
            }
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                const int i = r * L + q;
#pragma unroll
                for (int c = 0; c < (r + 1) * L; ++c) {
                    cd v = ld(&s_R[TIX(r, c)][lane]);
                    if (c == i) v = (i < M) ? v * 1.001 : cd{1.0, 0.0};  // whiteR diagonal; padding rows = identity
                    if (i >= M && c != i) v = cd{0, 0};
                    A[TIX(r, c)] = v;
                }
            }
#pragma unroll
            for (int jj = 0; jj < MP; ++jj) {
                const int ro = jj / L, qo = jj % L;  // owner slot / lane of row jj
                const double inv = rsqrt(bcast_from<L>(qo, A[TIX(ro, jj)].x));
                cd Lc[RPL];  // scaled column jj of the local rows (meaningful where row > jj)
#pragma unroll
                for (int r = ro; r < RPL; ++r) Lc[r] = A[TIX(r, jj)] * inv;
#pragma unroll
                for (int col = 0; col < NB; ++col) {  // right-hand sides: u_jj = b_jj / L_jj, then b_i -= L_ij u_jj
                    const cd u = bcast_from<L>(qo, b[ro][col] * inv);
                    if (q == qo) b[ro][col] = u;
#pragma unroll
                    for (int r = ro; r < RPL; ++r) {
                        const int i = r * L + q;
                        if (i > jj) b[r][col] = cfms(b[r][col], Lc[r], u);
                    }
                }
#pragma unroll
                for (int c = jj + 1; c < MP; ++c) {  // trailing update A_ic -= L_ij conj(L_cj), jj < c <= i
                    const cd Lcj = bcast_from<L>(c % L, Lc[c / L]);
#pragma unroll
                    for (int r = c / L; r < RPL; ++r) {
                        const int i = r * L + q;
                        if (c <= i) A[TIX(r, c)] = cfms_conj(A[TIX(r, c)], Lc[r], Lcj);
                    }
                }
            }
            // b holds the local rows of U = L^-1 [C | x]; Gram entries, reduced over the group
            cd ge[NE];
            {
                int e = 0;
#pragma unroll
                for (int r1 = 0; r1 < KM; ++r1)
#pragma unroll
                    for (int r2 = r1; r2 < KM; ++r2) {
                        cd acc{0, 0};
#pragma unroll
                        for (int r = 0; r < RPL; ++r)
                            if (r * L + q < M) acc = cfma_conj(acc, b[r][r2], b[r][r1]);
                        ge[e++] = cd{grp_sum<L>(acc.x), grp_sum<L>(acc.y)};
                    }
#pragma unroll
                for (int r1 = 0; r1 < KM; ++r1) {
                    cd acc{0, 0};
#pragma unroll
                    for (int r = 0; r < RPL; ++r)
                        if (r * L + q < M) acc = cfma_conj(acc, b[r][KM], b[r][r1]);
                    ge[NG + r1] = cd{grp_sum<L>(acc.x), grp_sum<L>(acc.y)};
                }
            }
            cd Gm[KM][KM], gv[KM];
            {
                int e = 0;
#pragma unroll
                for (int r1 = 0; r1 < KM; ++r1)
#pragma unroll
                    for (int r2 = r1; r2 < KM; ++r2) {
                        const cd v = ge[e++];
                        Gm[r1][r2] = v;
                        Gm[r2][r1] = conj(v);
                    }
#pragma unroll
                for (int r1 = 0; r1 < KM; ++r1) gv[r1] = ge[NG + r1];
#pragma unroll
                for (int r1 = 0; r1 < KM; ++r1)
                    if (r1 >= KP1) {
#pragma unroll
                        for (int r2 = 0; r2 < KM; ++r2) {
                            Gm[r1][r2] = cd{r1 == r2 ? 1.0 : 0.0, 0.0};
                            Gm[r2][r1] = cd{r1 == r2 ? 1.0 : 0.0, 0.0};
                        }
                        gv[r1] = cd{0, 0};
                    }
            }
#pragma unroll
            for (int k = 0; k < KM; ++k) {  // Gaussian elimination (G is Hermitian positive definite)
                const cd pinv = cdiv(cd{1, 0}, Gm[k][k]);
#pragma unroll
                for (int r1 = k + 1; r1 < KM; ++r1) {
                    const cd fct = Gm[r1][k] * pinv;
#pragma unroll
                    for (int c = k + 1; c < KM; ++c) Gm[r1][c] = Gm[r1][c] - fct * Gm[k][c];
                    gv[r1] = gv[r1] - fct * gv[k];
                }
            }
#pragma unroll
            for (int k = KM - 1; k >= 0; --k) {
                cd acc = gv[k];
#pragma unroll
                for (int c = k + 1; c < KM; ++c) acc = acc - Gm[k][c] * gv[c];
                gv[k] = cdiv(acc, Gm[k][k]);
            }
            y = gv[0];
        } else {
            y = x0 * 0.01;  // mvdr.cpp:96
        }
        if (!inband) y = (j == 0 && !lcmv) ? x0 : cd{0, 0};
        if (live && q == 0) yout[t * kYhStride] = f64x2{y.x, y.y};
        rank1(xl, xo, true);  // slide the covariance window (mvdr.cpp:100-101)
    }
#undef TIX
}


// ---- lcmv / mvdr, 9..16 microphones: one problem per 16-lane DPP row, exchange by row_newbcast ----------
// Same row-per-lane factorisation as mvdr_lcmv_kernel<16, KM>, but a problem occupies exactly one DPP row, so the
// pivot, the scaled column and the right-hand sides travel by `v_mov_b32_dpp row_newbcast:n` (lane n of every row to
// the whole row, one instruction per dword, VALU latency) instead of an LDS write -> s_waitcnt -> read round trip per
// column, and the Gram sums are row reductions (quad_perm xor 1/2, row_half_mirror, row_mirror).  No LDS at all.
template <int N>
__device__ __forceinline__ double rowbc(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), 0x150 + N, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x150 + N, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <int N>
__device__ __forceinline__ cd rowbc(cd v) { return cd{rowbc<N>(v.x), rowbc<N>(v.y)}; }
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double row_sum(double v) {  // every lane of the 16-lane row gets the row total
    v += dpp_d<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_d<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_d<0x141>(v);  // row_half_mirror
    v += dpp_d<0x140>(v);  // row_mirror
    return v;
}
template <int C, int MP>
struct RowStep {  // compile-time loops over the broadcast source lane
    template <typename F>
    static __device__ __forceinline__ void run(F &&f) {
        f(std::integral_constant<int, C>{});
        RowStep<C + 1, MP>::run(f);
    }
};
template <int MP>
struct RowStep<MP, MP> {
    template <typename F>
    static __device__ __forceinline__ void run(F &&) {}
};

template <int KM>
__global__ __launch_bounds__(256, 2) void mvdr_lcmv_row_kernel(BinsArgs a, int tile, int tiles_per_stream) {
    constexpr int MP = 16, GPB = 256 / MP, NB = KM + 1;
    constexpr int NG = GramIdx<KM>::NG, NE = GramIdx<KM>::NE;
    const int grp = threadIdx.x / MP;
    const int i = threadIdx.x % MP;
    const int q = blockIdx.y * GPB + grp;
    if (q >= kNQ) return;
    const int s = blockIdx.x / tiles_per_stream;
    const long tA = (long)(blockIdx.x % tiles_per_stream) * tile;
    long tB = tA + tile;
    if (tB > a.n_frames) tB = a.n_frames;
    const int M = a.n_mics, NP = (M + 1) >> 1, KP1 = a.kp1, P = a.cfg.past_windows;
    const int j = q_bin(q);
    const bool lcmv = a.cfg.algo == BF_LCMV;
    f64x2 *yout = a.Yh + ((long)s * a.n_frames) * kYhStride + q;
    const f64x2 *Zs = a.Z + ((long)(s / a.n_dirs) * a.frames_ws + a.frame_off) * NP * kN;
    const f64x2 *steer = a.steer + (long)(s % a.n_dirs) * a.steer_dir_stride;
    const int ksrc = q_src_bin(q), kneg = (kN - ksrc) & (kN - 1);
    auto load_xi = [&](long t) -> cd {
        if (i >= M) return cd{0, 0};
        const f64x2 *Zf = Zs + t * NP * kN + (i >> 1) * kN;
        const cd z = ld(Zf + ksrc), zc = conj(ld(Zf + kneg));
        cd x;
        if ((i & 1) == 0) {
            x = (z + zc) * 0.5;
        } else {
            const cd d = z - zc;
            x = cd{0.5 * d.y, -0.5 * d.x};
        }
        return q == 513 ? conj(x) : x;
    };
    const double f = fabs(a.freqs[j]);
    const bool inband = f >= a.cfg.freq_min && f <= a.cfg.freq_max;
    if (!inband || (!lcmv && j == 0)) {  // uniform per row
        for (long t = tA; t < tB; ++t) {
            cd y{0, 0};
            if (!lcmv && j == 0) y = rowbc<0>(load_xi(t));  // mvdr.cpp:76
            if (i == 0) yout[t * kYhStride] = f64x2{y.x, y.y};
        }
        return;
    }
    // this microphone's entries of the constraint columns are re-read per frame (L2-resident) straight into the
    // right-hand sides: holding them costs the 16 registers that decide between one and two wavefronts per SIMD
    auto load_cst = [&](int r) -> cd { return (r < KP1 && i < M) ? ld(steer + ((long)r * M + i) * kN + j) : cd{0, 0}; };

    cd R[MP];  // row i of R
#pragma unroll
    for (int c = 0; c < MP; ++c) R[c] = cd{0, 0};
    for (int p = 1; p <= P; ++p) {
        const cd x = load_xi(tA - p);
        RowStep<0, MP>::run([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            R[c] = cfma_conj(R[c], x, rowbc<c>(x));
        });
    }
    for (long t = tA; t < tB; ++t) {
        const cd x = load_xi(t);
        const double mag = row_sum(sqrt(norm2(x))) / (double)((unsigned)M * 1024u);
        cd y;
        if (mag > a.cfg.freq_mag_threshold) {  // uniform per row
            cd A[MP], b[NB];
#pragma unroll
            for (int c = 0; c < MP; ++c) A[c] = R[c];
            if (i < M) {
#pragma unroll
                for (int c = 0; c < MP; ++c)
                    if (c == i) A[c] = A[c] * 1.001;  // cwiseProduct(whiteR) (mvdr.cpp:239-243)
            } else {
#pragma unroll
                for (int c = 0; c < MP; ++c) A[c] = cd{c == i ? 1.0 : 0.0, 0.0};  // padding rows = identity
            }
#pragma unroll
            for (int r = 0; r < KM; ++r) b[r] = load_cst(r);
            b[KM] = x;
            RowStep<0, MP>::run([&](auto jc) {
                constexpr int jj = decltype(jc)::value;
                const double inv = rsqrt(rowbc<jj>(A[jj].x));  // 1 / L_jj from the owner's diagonal
                const cd Lij = A[jj] * inv;                   // my row's entry of the scaled column (valid for i > jj)
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    const cd bs = b[r] * inv;
                    const cd ujj = rowbc<jj>(bs);  // u_jj = b_jj / L_jj
                    if (i > jj)
                        b[r] = cfms(b[r], Lij, ujj);
                    else if (i == jj)
                        b[r] = bs;
                }
                RowStep<jj + 1, MP>::run([&](auto cc) {  // trailing update A_ic -= L_ij conj(L_cj), jj < c <= i
                    constexpr int c = decltype(cc)::value;
                    const cd Lcj = rowbc<c>(Lij);
                    if (i >= c) A[c] = cfms_conj(A[c], Lij, Lcj);
                });
            });
            // b[r] = row i of U = L^-1 [C | x]; Gram entries by row reduction
            cd ge[NE];
            {
                int e = 0;
#pragma unroll
                for (int r1 = 0; r1 < KM; ++r1)
#pragma unroll
                    for (int r2 = r1; r2 < KM; ++r2) {
                        cd pr = (i < M) ? cfma_conj(cd{0, 0}, b[r2], b[r1]) : cd{0, 0};
                        ge[e++] = cd{row_sum(pr.x), row_sum(pr.y)};
                    }
#pragma unroll
                for (int r1 = 0; r1 < KM; ++r1) {
                    cd pr = (i < M) ? cfma_conj(cd{0, 0}, b[KM], b[r1]) : cd{0, 0};
                    ge[NG + r1] = cd{row_sum(pr.x), row_sum(pr.y)};
                }
            }
            // (K+1) x (K+1) system G y = g on the upper triangle only (G and every Schur complement are Hermitian):
            // U[r][c], r <= c, is ge[] itself; rows / columns beyond the live constraints are the identity
            cd gv[KM];
            auto UI = [](int r, int c) { return r * KM - r * (r - 1) / 2 + (c - r); };
#pragma unroll
            for (int r1 = 0; r1 < KM; ++r1) gv[r1] = ge[NG + r1];
#pragma unroll
            for (int r1 = 0; r1 < KM; ++r1)
                if (r1 >= KP1) {
#pragma unroll
                    for (int r2 = 0; r2 < r1; ++r2) ge[UI(r2, r1)] = cd{0, 0};
                    ge[UI(r1, r1)] = cd{1.0, 0.0};
#pragma unroll
                    for (int c = r1 + 1; c < KM; ++c) ge[UI(r1, c)] = cd{0, 0};
                    gv[r1] = cd{0, 0};
                }
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                const cd pinv = cdiv(cd{1, 0}, ge[UI(k, k)]);
#pragma unroll
                for (int r1 = k + 1; r1 < KM; ++r1) {
                    const cd fct = conj(ge[UI(k, r1)]) * pinv;  // G[r1][k] / G[k][k]
#pragma unroll
                    for (int c = r1; c < KM; ++c) ge[UI(r1, c)] = ge[UI(r1, c)] - fct * ge[UI(k, c)];
                    gv[r1] = gv[r1] - fct * gv[k];
                }
            }
#pragma unroll
            for (int k = KM - 1; k >= 0; --k) {
                cd acc = gv[k];
#pragma unroll
                for (int c = k + 1; c < KM; ++c) acc = acc - ge[UI(k, c)] * gv[c];
                gv[k] = cdiv(acc, ge[UI(k, k)]);
            }
            y = gv[0];
        } else {
            y = rowbc<0>(x) * 0.01;  // in_fft(0,j)*0.01 (mvdr.cpp:96)
        }
        if (i == 0) yout[t * kYhStride] = f64x2{y.x, y.y};
        // slide the covariance window: + x_t x_t^H - x_{t-P} x_{t-P}^H (mvdr.cpp:100-101)
        const cd xo = load_xi(t - P);  // loaded late: 4 registers less across the factorisation
        RowStep<0, MP>::run([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            R[c] = cfms_conj(cfma_conj(R[c], x, rowbc<c>(x)), xo, rowbc<c>(xo));
        });
    }
}

}  // namespace

hipError_t launch_mvdr_lcmv(const BinsArgs &a, int n_cus, hipStream_t s) {
    int tile = 64;
    if (a.n_frames < tile) tile = (int)a.n_frames;
    const int tps = (int)((a.n_frames + tile - 1) / tile);
    const int M = a.n_mics, km = a.kp1 <= 1 ? 1 : 4;
    static const bool no_fast = getenv("BF_MVDR_GROUP") && atoi(getenv("BF_MVDR_GROUP")) != 0;
    // lcmv with 9..16 microphones: one problem per DPP row, row_newbcast exchange (34.2 -> 27.6 ms per 32 768 frames at 16)
    if (!no_fast && a.cfg.algo == BF_LCMV && M > 8) {
        const dim3 grid(tps * a.n_streams, (kNQ + 15) / 16);  // x: tile * stream (can exceed 65535), y: bin blocks
        if (km == 1)
            hipLaunchKernelGGL((mvdr_lcmv_row_kernel<1>), grid, dim3(256), 0, s, a, tile, tps);
        else
            hipLaunchKernelGGL((mvdr_lcmv_row_kernel<4>), grid, dim3(256), 0, s, a, tile, tps);
        return hipGetLastError();
    }
    // lanes kernel: mvdr with 9..16 microphones (12.7 vs 26 ms per 32 768 frames at 16) and lcmv with up to 8
    // (9.4 vs 15.4 ms per 65 536 frames).  lcmv with 9..16 microphones keeps the row-per-lane kernel: 40 complex of
    // working copy + 5 right-hand sides x 4 rows do not fit 512 registers (736 B of scratch, 5x slower).
    if (!no_fast && ((a.cfg.algo == BF_MVDR && M > 8) || (a.cfg.algo == BF_LCMV && M <= 8))) {
        int lt = 32;
        if (a.n_frames < lt) lt = (int)a.n_frames;
        const int ltps = (int)((a.n_frames + lt - 1) / lt);
        if (M <= 4) {
            const dim3 grid(ltps * a.n_streams, (kNQ + 31) / 32);  // x: tile * stream (can exceed 65535), y: bin blocks
            hipLaunchKernelGGL((mvdr_lcmv_lanes_kernel<4, 2, 4>), grid, dim3(64), 0, s, a, lt, ltps);
        } else if (M <= 8) {
            const dim3 grid(ltps * a.n_streams, (kNQ + 31) / 32);  // x: tile * stream (can exceed 65535), y: bin blocks
            hipLaunchKernelGGL((mvdr_lcmv_lanes_kernel<8, 2, 4>), grid, dim3(64), 0, s, a, lt, ltps);
        } else {
            const dim3 grid(ltps * a.n_streams, (kNQ + 15) / 16);  // x: tile * stream (can exceed 65535), y: bin blocks
            hipLaunchKernelGGL((mvdr_lcmv_lanes_kernel<16, 4, 1>), grid, dim3(64), 0, s, a, lt, ltps);
        }
        return hipGetLastError();
    }
    if (a.cfg.algo == BF_MVDR && M <= 8 && !no_fast) {
        int ft = 32;
        if (a.n_frames < ft) ft = (int)a.n_frames;
        const int ftps = (int)((a.n_frames + ft - 1) / ft);
        const dim3 grid(ftps * a.n_streams, (kNQ + 63) / 64);  // x: tile * stream (can exceed 65535), y: bin blocks
        if (M <= 4)
            hipLaunchKernelGGL((mvdr_fast_kernel<4>), grid, dim3(64), 0, s, a, ft, ftps);
        else
            hipLaunchKernelGGL((mvdr_fast_kernel<8>), grid, dim3(64), 0, s, a, ft, ftps);
        return hipGetLastError();
    }
#define BF_LAUNCH_ML(MP_, KM_)                                                                                   \
    hipLaunchKernelGGL((mvdr_lcmv_kernel<MP_, KM_>), dim3(tps * a.n_streams, (kNQ + (256 / MP_) - 1) / (256 / MP_)), \
                       dim3(256), 0, s, a, tile, tps)
    if (M <= 4) {
        if (km == 1) BF_LAUNCH_ML(4, 1); else BF_LAUNCH_ML(4, 4);
    } else if (M <= 8) {
        if (km == 1) BF_LAUNCH_ML(8, 1); else BF_LAUNCH_ML(8, 4);
    } else {
        if (km == 1) BF_LAUNCH_ML(16, 1); else BF_LAUNCH_ML(16, 4);
    }
#undef BF_LAUNCH_ML
    return hipGetLastError();
}

namespace {

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

// ======================================================================================
//                 gsc: generalized sidelobe canceller (gsc.cpp:54-197)
// ======================================================================================
// Pass 1 (align): output stream s*M + m carries microphone m of input stream s steered to the look direction,
// y_fft = x_fft * conj(weights[m]) over all bins (gsc.cpp:62-70); the ISTFT then does the per-microphone
// overlap-add of do_overlap_bymic (util.h:353-379).
__global__ __launch_bounds__(256) void gsc_align_kernel(BinsArgs a) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)a.n_streams * a.n_frames * kNQ;
    if (idx >= total) return;
    const int q = (int)(idx % kNQ);
    const long st = idx / kNQ;
    const long t = st % a.n_frames;
    const int so = (int)(st / a.n_frames);
    const int M = a.n_mics, NP = (M + 1) >> 1;
    const int si = so / M, m = so - si * M;
    const f64x2 *Zf = a.Z + (((long)si * a.frames_ws + a.frame_off + t) * NP + (m >> 1)) * kN;
    const int k = q_src_bin(q), kn = (kN - k) & (kN - 1);
    const cd z = ld(Zf + k), zc = conj(ld(Zf + kn));
    cd x;
    if ((m & 1) == 0) {
        x = (z + zc) * 0.5;
    } else {
        const cd d = z - zc;
        x = cd{0.5 * d.y, -0.5 * d.x};
    }
    if (q == 513) x = conj(x);
    const cd y = x * conj(ld(a.steer + (long)m * kN + q_bin(q)));
    a.Yh[((long)so * a.n_frames + t) * kYhStride + q] = f64x2{y.x, y.y};
}

// Pass 2 (NLMS): one wavefront per stream, strictly sample by sample.  The reference does this arithmetic in
// float32 (rosjack_data) with every product and sum rounded separately and the 128-tap sums taken in order, and
// it branches on the results (mu selection, NaN guards), so the kernel keeps exactly that order: lane i owns
// blocking branch i and walks its taps sequentially (__fmul_rn/__fadd_rn: no FMA contraction), lane M-1 does
// the same for the output-power window.  Only what is elementwise is spread over the lanes: the upper beamformer
// and the neighbour differences of a 64-sample tile (lane = sample), and the filter update (lane = tap), whose
// coefficients live in registers with a write-through copy in LDS for the serial walk.
// Windows are mirrored rings in LDS (each sample stored at p and p + fs) so a window is always contiguous.
// A block is one wavefront: its LDS operations complete in issue order, so phases are separated by compiler
// fences (wave_barrier), not s_barrier.
template <int NBM, int KPL>  // NBM >= blocking branches (M - 1), KPL >= ceil(filter_size / 64)
__global__ __launch_bounds__(64) void gsc_nlms_kernel(const float *aligned, float *y, float *state, long n, int M, int fs,
                                                      int use_vad, double vad_threshold, double mu0, double mu_max) {
    extern __shared__ float gl[];
    const int lane = threadIdx.x;
    const int nb = M - 1;                 // blocking branches
    const int nbr = nb > 0 ? nb : 1;
    // row lengths are padded to a whole number of 64-tap lane groups (+8 for the staged loads of the serial walk) and
    // made odd, so neither the walk nor the update needs a per-lane bounds guard and lane i / row i hit distinct banks
    const int bstride = (fs + 64 * KPL + 8) | 1;  // mirrored ring: a window starts at h1 < fs
    const int fstride = (64 * KPL + 8) | 1;
    float *s_bm = gl;                     // [nb][bstride]
    float *s_f = s_bm + nbr * bstride;    // [nb][fstride]
    float *s_lo = s_f + nbr * fstride;    // [2*fs]
    float *s_d = s_lo + 2 * fs + 16;      // [nb][64] neighbour differences of the current tile (16 words of slack first)
    float *s_das = s_d + nbr * 64;        // [64] upper beamformer of the current tile
    float *s_c = s_das + 64;              // [16] mu_i * out
    float *s_out = s_c + 16;              // [64]
    const int s = blockIdx.x;
    const float *as = aligned + (long)s * M * n;
    float *ys = y + (long)s * n;
    float *sv = state + (long)s * (2 * nb + 1) * fs;
    float freg[NBM][KPL];  // filter taps k = lane + 64 c of every branch
#pragma unroll
    for (int i = 0; i < NBM; ++i)
#pragma unroll
        for (int c = 0; c < KPL; ++c) {
            const int k = lane + 64 * c;
            float v = 0.f;
            if (i < nb && k < fs) {
                const float b = sv[i * fs + k];
                s_bm[i * bstride + k] = b;
                s_bm[i * bstride + k + fs] = b;
                v = sv[nb * fs + i * fs + k];
                s_f[i * fstride + k] = v;
            }
            freg[i][c] = v;
        }
    for (int k = lane; k < fs; k += 64) {
        const float v = sv[2 * nb * fs + k];
        s_lo[k] = v;
        s_lo[k + fs] = v;
    }
    __builtin_amdgcn_wave_barrier();
    int h = 0;  // ring position of the oldest element (same for every window: all advance once per sample)
    const float fsz = (float)fs;
    const bool is_branch = lane < nb, is_lo = lane == nb;
    const float *myrow = is_branch ? s_bm + lane * bstride : s_lo;  // lane nb (= M-1) walks the output window
    const float *myflt = is_branch ? s_f + lane * fstride : s_f;
    for (long n0 = 0; n0 < n; n0 += 64) {
        {   // tile prologue, lane = sample: das_out (gsc.cpp:122-127) and the blocking-matrix inputs (gsc.cpp:131)
            const bool ok = n0 + lane < n;
            float prev = ok ? as[n0 + lane] : 0.f, das = 0.f;
            das = __fadd_rn(das, prev);
            for (int m = 1; m < M; ++m) {
                const float cur = ok ? as[(long)m * n + n0 + lane] : 0.f;
                das = __fadd_rn(das, cur);
                s_d[(m - 1) * 64 + lane] = __fsub_rn(cur, prev);
                prev = cur;
            }
            s_das[lane] = __fdiv_rn(das, (float)M);
        }
        __builtin_amdgcn_wave_barrier();
        const int cnt = (n - n0) < 64 ? (int)(n - n0) : 64;
        for (int jj = 0; jj < cnt; ++jj) {
            const float das = s_das[jj];
            if (is_branch) {
                const float d = s_d[lane * 64 + jj];
                s_bm[lane * bstride + h] = d;
                s_bm[lane * bstride + h + fs] = d;
            }
            const int h1 = (h + 1 == fs) ? 0 : h + 1;  // window = [h1, h1 + fs)
            __builtin_amdgcn_wave_barrier();
            // lane i: block_out_i and the sum of squares of its window; lane nb: sum of squares of the output
            // window WITHOUT its newest element (added below, last, as the reference's loop order has it).
            // Loads are unconditional (every row is fs words long) and staged one group of 8 taps ahead of the
            // two dependent add chains.
            float bo = 0.f, pw = 0.f;
            {
                const float *u = myrow + h1;
                float un[8], wn[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    un[r] = u[r];
                    wn[r] = myflt[r];
                }
                // taps 0 .. fs-2 are common to all lanes; tap fs-1 belongs to the branch lanes only (the output
                // window's newest element is not known yet)
                int k = 0;
                for (; k + 8 <= fs - 1; k += 8) {
                    float uv[8], wv[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        uv[r] = un[r];
                        wv[r] = wn[r];
                    }
#pragma unroll
                    for (int r = 0; r < 8; ++r) {  // next group (reads past the window end land in the row padding)
                        un[r] = u[k + 8 + r];
                        wn[r] = myflt[k + 8 + r];
                    }
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        bo = __fadd_rn(bo, __fmul_rn(wv[r], uv[r]));
                        pw = __fadd_rn(pw, __fmul_rn(uv[r], uv[r]));
                    }
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) {  // leftover taps, already staged
                    if (k + r < fs - 1 || (k + r == fs - 1 && !is_lo)) {
                        bo = __fadd_rn(bo, __fmul_rn(wn[r], un[r]));
                        pw = __fadd_rn(pw, __fmul_rn(un[r], un[r]));
                    }
                }
            }
            float out = das;
            for (int i = 0; i < nb; ++i)
                out = __fsub_rn(out, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bo), i)));
            const float pwl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pw), nb));
            const float lop = __fsqrt_rn(__fdiv_rn(__fadd_rn(pwl, __fmul_rn(out, out)), fsz));  // calculate_power(last_outputs)
            if (lane == 0) {
                s_lo[h] = out;
                s_lo[h + fs] = out;
                s_out[jj] = out;
            }
            const bool adapt = ((double)lop < vad_threshold) || !use_vad;  // gsc.cpp:147
            if (adapt && nb > 0) {
                if (is_branch) {
                    const float bp = __fsqrt_rn(__fdiv_rn(pw, fsz));
                    float mu;
                    if (mu0 * (double)bp / (double)lop < mu_max)  // gsc.cpp:153-157 (double arithmetic: mu0 is a double)
                        mu = (float)(mu0 / (double)lop);
                    else
                        mu = (float)(mu0 / (double)bp);
                    if (isnan(mu) || isinf(mu)) mu = 0.f;
                    s_c[lane] = __fmul_rn(mu, out);
                }
                __builtin_amdgcn_wave_barrier();
                // filter[i][k] += this_mu*out[j]*block_matrix[i][k] (gsc.cpp:163-170), taps over the lanes:
                // all loads first, then the arithmetic and the write-through stores
                float bmv[NBM][KPL], cv[NBM];
#pragma unroll
                for (int i = 0; i < NBM; ++i) {
                    const int ic = i < nb ? i : 0;
                    cv[i] = s_c[ic];
#pragma unroll
                    for (int c = 0; c < KPL; ++c) bmv[i][c] = s_bm[ic * bstride + h1 + lane + 64 * c];
                }
#pragma unroll
                for (int i = 0; i < NBM; ++i)
                    if (i < nb) {  // uniform
#pragma unroll
                        for (int c = 0; c < KPL; ++c) {  // lanes past filter_size work on row padding nobody reads
                            float fv = __fadd_rn(freg[i][c], __fmul_rn(cv[i], bmv[i][c]));
                            if (isnan(fv)) fv = 0.f;
                            freg[i][c] = fv;
                            s_f[i * fstride + lane + 64 * c] = fv;
                        }
                    }
            }
            __builtin_amdgcn_wave_barrier();
            h = h1;
        }
        if (lane < cnt) ys[n0 + lane] = s_out[lane];
        __builtin_amdgcn_wave_barrier();
    }
    // carried state in the reference's (shifted, oldest-first) order
    for (int e = lane; e < nb * fs; e += 64) {
        const int i = e / fs, k = e - i * fs;
        sv[e] = s_bm[i * bstride + h + k];
        sv[nb * fs + e] = s_f[i * fstride + k];
    }
    for (int k = lane; k < fs; k += 64) sv[2 * nb * fs + k] = s_lo[h + k];
}

// ======================================================================================
//                              gss: geometric source separation
// ======================================================================================
// One group of MP lanes per (stream, problem), lane m owns column m of the demixing matrix
// W_j (S x M) and walks the frames in order (the update is recursive, gss.cpp:136).
template <int MP, int KM>
__global__ __launch_bounds__(256) void gss_kernel(BinsArgs a) {
    constexpr int GPB = 256 / MP;
    __shared__ cd s_x[GPB][MP + 1];   // padded rows: see mvdr_lcmv_kernel (gss 256x256: 6.5 -> 5.8 ms)
    __shared__ cd s_p[GPB][KM][MP + 1];
    const int grp = threadIdx.x / MP, m = threadIdx.x % MP;
    const int gq = blockIdx.x * GPB + grp;
    if (gq >= a.n_streams * kNQ) return;
    const int s = gq / kNQ, q = gq % kNQ;
    const int j = q_bin(q);
    const int M = a.n_mics, NP = (M + 1) >> 1, S = a.kp1;
    f64x2 *yout = a.Yh + ((long)s * a.n_frames) * kYhStride + q;
    const f64x2 *Zs = a.Z + ((long)(s / a.n_dirs) * a.frames_ws + a.frame_off) * NP * kN;
    const f64x2 *steer = a.steer + (long)(s % a.n_dirs) * a.steer_dir_stride;
    const double f = fabs(a.freqs[j]);
    const bool inband = f >= a.cfg.freq_min && f <= a.cfg.freq_max;
    if (!inband) {
        if (m == 0)
            for (long t = 0; t < a.n_frames; ++t) yout[t * kYhStride] = f64x2{0, 0};
        return;
    }
    const int ksrc = q_src_bin(q), kneg = (kN - ksrc) & (kN - 1);
    cd C[KM], W[KM];
    f64x2 *Wg = a.gssW + (((long)s * kN + j) * S) * M;
#pragma unroll
    for (int r = 0; r < KM; ++r) {
        C[r] = (r < S && m < M) ? ld(steer + ((long)r * M + m) * kN + j) : cd{0, 0};
        if ((a.gss_reset_mask >> (s % a.n_dirs)) & 1ull)
            W[r] = conj(C[r]);  // sep_matrix[j] = weights[j].adjoint() (gss.cpp:92)
        else
            W[r] = (r < S && m < M) ? ld(Wg + (long)r * M + m) : cd{0, 0};
    }
    const double mu = a.cfg.mu, keep = 1 - a.cfg.lambda_ * a.cfg.mu;
    const double c2 = (double)(size_t)(2 * (1 / (size_t)S));  // integer arithmetic, quirk Q13
    for (long t = 0; t < a.n_frames; ++t) {
        cd x{0, 0};
        if (m < M) {
            const f64x2 *Zf = Zs + t * NP * kN + (m >> 1) * kN;
            const cd z = ld(Zf + ksrc), zc = conj(ld(Zf + kneg));
            if ((m & 1) == 0) {
                x = (z + zc) * 0.5;
            } else {
                const cd d = z - zc;
                x = cd{0.5 * d.y, -0.5 * d.x};
            }
            if (q == 513) x = conj(x);
        }
        s_x[grp][m] = x;
#pragma unroll
        for (int r = 0; r < KM; ++r) s_p[grp][r][m] = W[r] * x;
        __builtin_amdgcn_wave_barrier();
        double mag = 0.0, alpha = 0.0;
        for (int k = 0; k < M; ++k) {
            const cd v = s_x[grp][k];
            mag += cabs(v);
            alpha += norm2(v);
        }
        mag /= (double)((unsigned)M * 1024u);
        cd y;
        if (mag > a.cfg.freq_mag_threshold) {
            cd yf[KM];
#pragma unroll
            for (int r = 0; r < KM; ++r) {
                cd acc{0, 0};
                for (int k = 0; k < M; ++k) acc = acc + s_p[grp][r][k];
                yf[r] = acc;
            }
            y = yf[0];
            alpha *= alpha;
            const double c1 = (double)(4 * (size_t)S) * (1 / alpha);
            cd Ey[KM];
#pragma unroll
            for (int r = 0; r < KM; ++r) {
                cd acc{0, 0};
#pragma unroll
                for (int r2 = 0; r2 < KM; ++r2)
                    if (r2 != r && r < S && r2 < S) acc = acc + (yf[r] * conj(yf[r2])) * yf[r2];
                Ey[r] = acc;
            }
            cd d2[KM];
#pragma unroll
            for (int r = 0; r < KM; ++r) d2[r] = cd{0, 0};
            if (c2 != 0.0) {  // only S == 1: dj2 = 2 (W C - I) C^H
                __builtin_amdgcn_wave_barrier();
                s_p[grp][0][m] = W[0] * C[0];
                __builtin_amdgcn_wave_barrier();
                cd wc{0, 0};
                for (int k = 0; k < M; ++k) wc = wc + s_p[grp][0][k];
                wc.x -= 1.0;
                d2[0] = (wc * conj(C[0])) * c2;
            }
#pragma unroll
            for (int r = 0; r < KM; ++r)
                if (r < S) W[r] = (W[r] * keep) - ((Ey[r] * conj(x)) * c1 + d2[r]) * mu;
        } else {
            y = s_x[grp][0] * 0.01;
        }
        if (m == 0) yout[t * kYhStride] = f64x2{y.x, y.y};
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int r = 0; r < KM; ++r)
        if (r < S && m < M) Wg[(long)r * M + m] = f64x2{W[r].x, W[r].y};
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

hipError_t launch_gss(const BinsArgs &a, int n_cus, hipStream_t s) {
    const int M = a.n_mics, km = a.kp1 <= 1 ? 1 : 4;
    const int groups = a.n_streams * kNQ;
#define BF_LAUNCH_GSS(MP_, KM_) \
    hipLaunchKernelGGL((gss_kernel<MP_, KM_>), dim3((groups + (256 / MP_) - 1) / (256 / MP_)), dim3(256), 0, s, a)
    if (M <= 4) {
        if (km == 1) BF_LAUNCH_GSS(4, 1); else BF_LAUNCH_GSS(4, 4);
    } else if (M <= 8) {
        if (km == 1) BF_LAUNCH_GSS(8, 1); else BF_LAUNCH_GSS(8, 4);
    } else {
        if (km == 1) BF_LAUNCH_GSS(16, 1); else BF_LAUNCH_GSS(16, 4);
    }
#undef BF_LAUNCH_GSS
    return hipGetLastError();
}

// ---- launchers ---------------------------------------------------------------------------
hipError_t launch_stft(const StftArgs &a, int n_cus, hipStream_t s) {
    const long total = (long)a.n_streams * a.n_frames * ((a.n_fft_mics + 1) / 2);
    long blocks = (total + kStftHalves - 1) / kStftHalves;
    const long cap = (long)n_cus * 4;
    if (blocks > cap) blocks = cap;
    if (a.layout == 0)
        hipLaunchKernelGGL(stft_kernel<0>, dim3((unsigned)blocks), dim3(kStftBlock), 0, s, a);
    else
        hipLaunchKernelGGL(stft_kernel<1>, dim3((unsigned)blocks), dim3(kStftBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_istft(const IstftArgs &a, int n_cus, hipStream_t s) {
    const long pairs = (a.n_frames + 1) / 2;
    long slots = (long)n_cus * kIstftHalves * 2 / a.n_streams;
    if (slots < 1) slots = 1;
    long cps = slots < pairs ? slots : pairs;
    const long ppc = (pairs + cps - 1) / cps;
    cps = (pairs + ppc - 1) / ppc;
    const long chunks = cps * a.n_streams;
    hipLaunchKernelGGL(istft_kernel, dim3((unsigned)((chunks + kIstftHalves - 1) / kIstftHalves)), dim3(kIstftBlock), 0, s, a,
                       (int)ppc, (int)cps);
    return hipGetLastError();
}

hipError_t launch_gsc_nlms(const float *aligned, float *y, float *state, long n_samples, int n_streams, int n_mics,
                           const bf_config &cfg, hipStream_t s) {
    const int fs = cfg.gsc_filter_size, nb = n_mics - 1, nbr = nb > 0 ? nb : 1;
    const int kpl = (fs + 63) / 64, kp = kpl <= 1 ? 1 : kpl <= 2 ? 2 : 4;
    const size_t lds = sizeof(float) * ((size_t)nbr * ((fs + 64 * kp + 8) | 1) + (size_t)nbr * ((64 * kp + 8) | 1) + 2 * fs + 16 +
                                        (size_t)nbr * 64 + 64 + 16 + 64);
#define BF_NLMS(NBM_, KPL_)                                                                                              \
    hipLaunchKernelGGL((gsc_nlms_kernel<NBM_, KPL_>), dim3((unsigned)n_streams), dim3(64), lds, s, aligned, y, state,    \
                       n_samples, n_mics, fs, cfg.gsc_use_vad, cfg.gsc_vad_threshold, cfg.gsc_mu0, cfg.gsc_mu_max)
#define BF_NLMS_K(NBM_)                     \
    do {                                    \
        if (kpl <= 1) BF_NLMS(NBM_, 1);     \
        else if (kpl <= 2) BF_NLMS(NBM_, 2);\
        else BF_NLMS(NBM_, 4);              \
    } while (0)
    if (nb <= 1) BF_NLMS_K(1);
    else if (nb <= 3) BF_NLMS_K(3);
    else if (nb <= 7) BF_NLMS_K(7);
    else BF_NLMS_K(15);
#undef BF_NLMS_K
#undef BF_NLMS
    return hipGetLastError();
}

hipError_t launch_smooth(const float *yraw, float *y, double *state, long n_frames, int n_streams, int smooth_size,
                         hipStream_t s) {
    const long n = n_frames * kHop;
    const long total = n * n_streams;
    hipLaunchKernelGGL(smooth_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, yraw, y, state, n, n_streams,
                       smooth_size);
    hipLaunchKernelGGL(smooth_state_kernel, dim3((unsigned)((64 * n_streams + 255) / 256)), dim3(256), 0, s, yraw, state, n,
                       n_streams);
    return hipGetLastError();
}

hipError_t launch_phasempf(const BinsArgs &a, int n_cus, hipStream_t s);
hipError_t launch_gss(const BinsArgs &a, int n_cus, hipStream_t s);

template <int ALGO>
static void launch_pointwise(const BinsArgs &a, hipStream_t s) {
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

hipError_t launch_bins(const BinsArgs &a, int n_cus, hipStream_t s) {
    hipError_t e = hipSuccess;
    switch (a.cfg.algo) {
        case BF_DAS: launch_pointwise<BF_DAS>(a, s); e = hipGetLastError(); break;
        case BF_PHASE: launch_pointwise<BF_PHASE>(a, s); e = hipGetLastError(); break;
        case BF_MVDR:
        case BF_LCMV: e = launch_mvdr_lcmv(a, n_cus, s); break;
        case BF_PHASEMPF: e = launch_phasempf(a, n_cus, s); break;
        case BF_GSS: e = launch_gss(a, n_cus, s); break;
        case BF_GSC: {
            const long total = (long)a.n_streams * a.n_frames * kNQ;
            hipLaunchKernelGGL(gsc_align_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
            e = hipGetLastError();
            break;
        }
        case BF_MCRA:
            hipLaunchKernelGGL(mcra_node_kernel, dim3((a.n_streams * kNQ + 63) / 64), dim3(64), 0, s, a);
            e = hipGetLastError();
            break;
        default: e = hipErrorInvalidValue; break;
    }
    if (e != hipSuccess) return e;
    if (a.spectrum) {
        const long frames = (long)a.n_streams * a.n_frames;
        hipLaunchKernelGGL(expand_spectrum_kernel, dim3((unsigned)((frames * kN + 255) / 256)), dim3(256), 0, s, a.Yh,
                           a.spectrum, frames);
        e = hipGetLastError();
    }
    return e;
}


}  // namespace bf
