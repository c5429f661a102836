// resample.hip -- the rosjack output stage's sample-rate converter as a batch operator.
//
// Reference call sites: rosjack.cpp:159-184 (src_new(SRC_SINC_FASTEST, 1 channel), src_ratio = ros_output_sample_rate /
// rosjack_sample_rate), rosjack.cpp:311-338 (convert_to_sample_rate: one src_process per JACK period, end_of_input = 0, output
// queued in a ring and emitted 512 samples at a time, :340-349, :416-427).  The converter itself is a third-party dependency
// that the reference does not vendor: libsamplerate (libsamplerate0-dev; 0.1.9 on the reference's Ubuntu 20.04), src_sinc.c,
// mono path "sinc_mono_vari_process" + "calc_output_single".  Its published algorithm is restated here:
//   * a half table of a windowed sinc sampled `index_inc` times per zero crossing of the unit-rate sinc, read with linear
//     interpolation at a 12-bit fixed-point fractional index;
//   * for every output sample at input position `cur + frac`: left wing = taps cur, cur-1, ... ; right wing = cur+1, cur+2, ...;
//     filter index step `increment = fixed(index_inc * min(ratio, 1))` (the filter is stretched by 1/ratio when downsampling),
//     start index `fixed(frac * index_inc * min(ratio, 1))`; the sum is scaled by min(ratio, 1);
//   * accumulation and coefficient interpolation in double, output rounded to float;
//   * history before the first sample is zero; with end_of_input = 0 an output is produced once the input reaches
//     `half_taps = lrint((coeff_half_len + 2) / index_inc / min(ratio, 1)) + 1` samples beyond its position.
// PARITY UNPINNED: libsamplerate's SINC_FASTEST coefficient table (fastest_coeffs.h, 2464 entries, index_inc 128) is not in the
// image, so the built-in table is a Kaiser-windowed sinc of the same geometry (cut-off 0.8315 of Nyquist = the published first
// coefficient, beta 9.73 for the converter's stated 97 dB) -- the same kind of filter, not the same numbers.  A caller that has
// libsamplerate's table loads it with bf_resampler_set_table and then runs libsamplerate's arithmetic on it.
// One deliberate difference: the input position of output K is K * in_rate / out_rate in exact integer arithmetic, where
// libsamplerate accumulates 1 / ratio in a double (its drift is ~1e-16 per sample); results do not depend on how the stream is
// cut into calls.
//
// Kernels: one output sample per thread, taps read through L1/L2 (neighbouring outputs share all but one or two taps).  When
// the rates give few distinct fractional positions (48000 -> 16000: 1, 48000 -> 44100: 147) the interpolated coefficients are
// tabulated per position once (polyphase form, tables in LDS, one FMA per tap); otherwise the table is interpolated per tap.
// Streaming bound: 4 B in / ratio + 4 B out per output sample; fp64 FMA work 2 * half_taps per output.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "launch_trace.hpp"
#include "../../include/bfcore.h"

namespace {

constexpr int kShift = 12;              // src_sinc.c SHIFT_BITS
constexpr int kMaxTable = 8192;         // floats of LDS for the table
constexpr int kDefaultLen = 2464;       // fastest_coeffs.h: ARRAY_LEN
constexpr int kDefaultInc = 128;        // fastest_coeffs.increment

struct ResampleArgs {
    const float *hist;   // the hist_len input samples in front of `in`
    const float *in;     // new input, n_in samples
    float *out;
    long hist_len, n_in;
    long first_out;      // global index of out[0]
    long in_base;        // global index of in[0]
    long n_out;
    long in_rate, out_rate;
    int coeff_half_len;  // table entries - 2
    int index_inc;
    int table_len;
    const float *table;
};

__device__ __forceinline__ double tap(const ResampleArgs &a, long g) {  // input sample with global index g
    const long r = g - a.in_base;
    if (r >= 0) return r < a.n_in ? (double)a.in[r] : 0.0;
    const long h = r + a.hist_len;
    return h >= 0 ? (double)a.hist[h] : 0.0;
}

__global__ __launch_bounds__(256) void sinc_resample_kernel(ResampleArgs a) {
    __shared__ float s_tab[kMaxTable];
    for (int i = threadIdx.x; i < a.table_len; i += 256) s_tab[i] = a.table[i];
    __syncthreads();
    const double ratio = (double)a.out_rate / (double)a.in_rate;
    const double float_increment = (double)a.index_inc * (ratio < 1.0 ? ratio : 1.0);
    const int increment = (int)llrint(float_increment * (double)(1 << kShift));
    const int max_filter_index = a.coeff_half_len << kShift;
    const double inv_fp = 1.0 / (double)(1 << kShift);
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < a.n_out; k += (long)gridDim.x * 256) {
        const long K = a.first_out + k;
        // position of output K on the input axis: K * in_rate / out_rate = cur + frac
        const unsigned long long num = (unsigned long long)K * (unsigned long long)a.in_rate;
        const long cur = (long)(num / (unsigned long long)a.out_rate);
        const double frac = (double)(num % (unsigned long long)a.out_rate) / (double)a.out_rate;
        const int start = (int)llrint(frac * float_increment * (double)(1 << kShift));
        // left wing: cur, cur - 1, ...
        int fi = start;
        int cc = (max_filter_index - fi) / increment;
        fi += cc * increment;
        long di = cur - cc;
        double left = 0.0;
        do {
            const double fr = (double)(fi & ((1 << kShift) - 1)) * inv_fp;
            const int ix = fi >> kShift;
            const double c0 = (double)s_tab[ix], c1 = (double)s_tab[ix + 1];
            left += (c0 + fr * (c1 - c0)) * tap(a, di);
            fi -= increment;
            ++di;
        } while (fi >= 0);
        // right wing: cur + 1, cur + 2, ...
        fi = increment - start;
        cc = (max_filter_index - fi) / increment;
        fi += cc * increment;
        di = cur + 1 + cc;
        double right = 0.0;
        do {
            const double fr = (double)(fi & ((1 << kShift) - 1)) * inv_fp;
            const int ix = fi >> kShift;
            const double c0 = (double)s_tab[ix], c1 = (double)s_tab[ix + 1];
            right += (c0 + fr * (c1 - c0)) * tap(a, di);
            fi -= increment;
            --di;
        } while (fi > 0);
        a.out[k] = (float)((float_increment / (double)a.index_inc) * (left + right));
    }
}

// new history = the last hist_len samples of [hist | in]
__global__ void hist_roll_kernel(const float *hist, const float *in, long hist_len, long n_in, float *hist_new) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hist_len) return;
    const long r = n_in - hist_len + i;  // index relative to in[0]
    hist_new[i] = r >= 0 ? in[r] : hist[hist_len + r];
}

// ---- polyphase form ---------------------------------------------------------------------------------------------------------------
// With integer rates the fractional position of output K takes only L = out_rate / gcd values, so the interpolated coefficients
// of every tap can be tabulated once per converter: row m holds, in libsamplerate's order of evaluation, the left wing (far tap
// first, ending at the tap on `cur`) and then the right wing (far tap first, ending at `cur + 1`).  The per-output loop is then
// one LDS read, one sample and one FMA per tap instead of two table reads, two conversions and the interpolation.
struct PolyArgs {
    double *poly;     // [L][row] doubles: left wing at 0, right wing at `right_off`
    int2 *count;      // [L] taps per wing
    int L, row, right_off;
    long in_rate, out_rate, g;
    int coeff_half_len, index_inc;
    const float *table;
};

__global__ void poly_build_kernel(PolyArgs a) {
    const int m = blockIdx.x;
    const double ratio = (double)a.out_rate / (double)a.in_rate;
    const double float_increment = (double)a.index_inc * (ratio < 1.0 ? ratio : 1.0);
    const int increment = (int)llrint(float_increment * (double)(1 << kShift));
    const int max_filter_index = a.coeff_half_len << kShift;
    const double inv_fp = 1.0 / (double)(1 << kShift);
    const double frac = (double)((long)m * a.g) / (double)a.out_rate;
    const int start = (int)llrint(frac * float_increment * (double)(1 << kShift));
    const int ccl = (max_filter_index - start) / increment;
    const int fr0 = increment - start;
    const int ccr = (max_filter_index - fr0) / increment;
    if (threadIdx.x == 0) a.count[m] = int2{ccl + 1, ccr + 1};
    for (int e = threadIdx.x; e <= ccl; e += blockDim.x) {
        const int fi = start + (ccl - e) * increment;
        const double fr = (double)(fi & ((1 << kShift) - 1)) * inv_fp;
        const int ix = fi >> kShift;
        const double c0 = (double)a.table[ix], c1 = (double)a.table[ix + 1];
        a.poly[(long)m * a.row + e] = c0 + fr * (c1 - c0);
    }
    for (int e = threadIdx.x; e <= ccr; e += blockDim.x) {
        const int fi = fr0 + (ccr - e) * increment;
        const double fr = (double)(fi & ((1 << kShift) - 1)) * inv_fp;
        const int ix = fi >> kShift;
        const double c0 = (double)a.table[ix], c1 = (double)a.table[ix + 1];
        a.poly[(long)m * a.row + a.right_off + e] = c0 + fr * (c1 - c0);
    }
}

constexpr int kPolyLdsDoubles = 7936;  // 62 KiB of coefficients per block

__global__ __launch_bounds__(256) void sinc_poly_kernel(ResampleArgs a, PolyArgs q) {
    __shared__ double s_poly[kPolyLdsDoubles];
    __shared__ int2 s_cnt[512];
    for (int i = threadIdx.x; i < q.L * q.row; i += 256) s_poly[i] = q.poly[i];
    for (int i = threadIdx.x; i < q.L; i += 256) s_cnt[i] = q.count[i];
    __syncthreads();
    const double ratio = (double)a.out_rate / (double)a.in_rate;
    const double scale = ((double)a.index_inc * (ratio < 1.0 ? ratio : 1.0)) / (double)a.index_inc;
    for (long k0 = (long)blockIdx.x * 256; k0 < a.n_out; k0 += (long)gridDim.x * 256) {
        const long k = k0 + threadIdx.x;
        const bool live = k < a.n_out;
        const long K = a.first_out + (live ? k : a.n_out - 1);
        const unsigned long long num = (unsigned long long)K * (unsigned long long)a.in_rate;
        const long cur = (long)(num / (unsigned long long)a.out_rate);
        const int m = (int)((num % (unsigned long long)a.out_rate) / (unsigned long long)q.g);
        const int2 cnt = s_cnt[m];
        const double *rowp = s_poly + m * q.row;
        // does every tap of this block's outputs lie inside the new input?  (block-uniform: first and last output of the block)
        const long Kf = a.first_out + k0, Kl = a.first_out + (k0 + 255 < a.n_out ? k0 + 255 : a.n_out - 1);
        const long lo = (long)((unsigned long long)Kf * (unsigned long long)a.in_rate / (unsigned long long)a.out_rate) - q.right_off - a.in_base;
        const long hi = (long)((unsigned long long)Kl * (unsigned long long)a.in_rate / (unsigned long long)a.out_rate) + (q.row - q.right_off) + 1 - a.in_base;
        double left = 0.0, right = 0.0;
        if (lo >= 0 && hi < a.n_in) {
            const float *xl = a.in + (cur - a.in_base) - (cnt.x - 1);
            for (int e = 0; e < cnt.x; ++e) left += rowp[e] * (double)xl[e];
            const float *xr = a.in + (cur - a.in_base) + cnt.y;
            const double *rr = rowp + q.right_off;
            for (int e = 0; e < cnt.y; ++e) right += rr[e] * (double)xr[-e];
        } else {
            for (int e = 0; e < cnt.x; ++e) left += rowp[e] * tap(a, cur - (cnt.x - 1) + e);
            const double *rr = rowp + q.right_off;
            for (int e = 0; e < cnt.y; ++e) right += rr[e] * tap(a, cur + cnt.y - e);
        }
        if (live) a.out[k] = (float)(scale * (left + right));
    }
}

// Polyphase form with the block's input window staged in LDS: the 256 outputs of a block read 256 * in/out + taps consecutive
// input samples (886 floats at 48 -> 16 kHz), fetched once, coalesced, through tap() -- history, new input and the zeros beyond
// the data all look the same afterwards, so there is ONE summation path whatever the position of the block in the stream
// (results stay independent of how the stream is cut into calls).  Dynamic LDS: coefficients + tap counts + window.
__global__ __launch_bounds__(256) void sinc_poly_lds_kernel(ResampleArgs a, PolyArgs q, int win_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    double *s_poly = reinterpret_cast<double *>(dyn_lds);
    int2 *s_cnt = reinterpret_cast<int2 *>(s_poly + q.L * q.row);
    float *s_x = reinterpret_cast<float *>(s_cnt + q.L);
    for (int i = threadIdx.x; i < q.L * q.row; i += 256) s_poly[i] = q.poly[i];
    for (int i = threadIdx.x; i < q.L; i += 256) s_cnt[i] = q.count[i];
    const double ratio = (double)a.out_rate / (double)a.in_rate;
    const double scale = ((double)a.index_inc * (ratio < 1.0 ? ratio : 1.0)) / (double)a.index_inc;
    const int max_l = q.right_off, max_r = q.row - q.right_off;
    for (long k0 = (long)blockIdx.x * 256; k0 < a.n_out; k0 += (long)gridDim.x * 256) {
        const long Kf = a.first_out + k0, Kl = a.first_out + (k0 + 255 < a.n_out ? k0 + 255 : a.n_out - 1);
        const long win0 = (long)((unsigned long long)Kf * (unsigned long long)a.in_rate / (unsigned long long)a.out_rate) - (max_l - 1);
        const long win1 = (long)((unsigned long long)Kl * (unsigned long long)a.in_rate / (unsigned long long)a.out_rate) + max_r;
        const int W = (int)(win1 - win0 + 1);  // <= win_cap by the launcher's choice of this kernel
        __syncthreads();                        // previous window fully consumed (and, first time, the tables are in place)
        for (int i = threadIdx.x; i < W && i < win_cap; i += 256) s_x[i] = (float)tap(a, win0 + i);
        __syncthreads();
        const long k = k0 + threadIdx.x;
        if (k < a.n_out) {
            const long K = a.first_out + k;
            const unsigned long long num = (unsigned long long)K * (unsigned long long)a.in_rate;
            const long cur = (long)(num / (unsigned long long)a.out_rate);
            const int m = (int)((num % (unsigned long long)a.out_rate) / (unsigned long long)q.g);
            const int2 cnt = s_cnt[m];
            const double *rowp = s_poly + m * q.row;
            const float *xl = s_x + (cur - win0) - (cnt.x - 1);
            double left = 0.0, right = 0.0;
            for (int e = 0; e < cnt.x; ++e) left += rowp[e] * (double)xl[e];
            const float *xr = s_x + (cur - win0) + cnt.y;
            const double *rr = rowp + q.right_off;
            for (int e = 0; e < cnt.y; ++e) right += rr[e] * (double)xr[-e];
            a.out[k] = (float)(scale * (left + right));
        }
    }
}

double bessel_i0(double x) {
    double sum = 1.0, term = 1.0;
    for (int k = 1; k < 200; ++k) {
        term *= (x / (2.0 * k)) * (x / (2.0 * k));
        sum += term;
        if (term < 1e-18 * sum) break;
    }
    return sum;
}

std::vector<float> kaiser_sinc_table(int len, int inc) {
    std::vector<float> t((size_t)len);
    const double fc = 0.83147237295484055508;  // libsamplerate's published coeffs[0] for SRC_SINC_FASTEST
    const double beta = 0.1102 * (97.0 - 8.7);
    const double i0b = bessel_i0(beta);
    for (int i = 0; i < len; ++i) {
        const double x = (double)i / (double)inc * fc;
        const double s = i == 0 ? 1.0 : sin(M_PI * x) / (M_PI * x);
        const double u = (double)i / (double)(len - 1);
        const double w = u < 1.0 ? bessel_i0(beta * sqrt(1.0 - u * u)) / i0b : 0.0;
        t[(size_t)i] = (float)(fc * s * w);
    }
    t[(size_t)len - 1] = 0.0f;  // the guard entry read by the interpolation at the end of the wing
    return t;
}

}  // namespace

struct bf_resampler {
    std::mutex mu;
    long in_rate = 0, out_rate = 0;
    int table_len = 0, index_inc = 0;
    float *d_table = nullptr;
    float *d_hist[2] = {nullptr, nullptr};
    int hist_cur = 0;
    long hist_len = 0;
    long consumed = 0;   // input samples taken so far
    long generated = 0;  // output samples produced so far
    // polyphase tables (built when the rates give few enough phases to fit the LDS; otherwise the generic kernel runs)
    double *d_poly = nullptr;
    int2 *d_count = nullptr;
    int poly_L = 0, poly_row = 0, poly_right_off = 0;
    long poly_g = 1;
    float *d_in = nullptr, *d_out = nullptr;  // staging for the host entry point
    size_t cap_in = 0, cap_out = 0;
    // BF_RS_ROSJACK (bf_resampler_set_mode): the stage as rosjack drives it, one call per JACK period
    int mode = BF_RS_STREAM;
    long period = 0;                 // rosjack_window_size: the input block, the cap on one src_process' output, the published block
    long b_len = 0, b_current = 0, b_end = 0;  // libsamplerate's buffer bookkeeping: decides input_frames_used (prepare_data)
    float *d_pend = nullptr;         // samplerate_buff_in: the last accepted period
    long pend_off = 0, pend_n = 0;   // its part not yet pulled into the converter (data_in / input_frames)
    float *d_fifo[2] = {nullptr, nullptr};  // every output computed so far and not yet published (samplerate_circbuff and what the
    int fifo_cur = 0;                       // converter would still hold back), linear; compacted into the other buffer when full
    long fifo_cap = 0, fifo_base = 0;       // global output index of d_fifo[fifo_cur][0]
    long released = 0, emitted = 0;         // outputs src_process has handed over / rosjack has published

    double ratio() const { return (double)out_rate / (double)in_rate; }
    long half_taps() const {
        double count = ((double)(table_len - 2) + 2.0) / (double)index_inc;
        const double r = ratio();
        if (r < 1.0) count /= r;
        return lrint(count) + 1;
    }
    // outputs that exist once `avail` input samples have been seen: K with cur(K) + half_taps < avail
    long outputs_for(long avail) const {
        const long lim = avail - half_taps();  // cur(K) < lim
        if (lim <= 0) return 0;
        // cur(K) = floor(K in / out) < lim  <=>  K in < lim out  <=>  K <= (lim out - 1) / in
        return (long)(((unsigned long long)lim * (unsigned long long)out_rate - 1ull) / (unsigned long long)in_rate) + 1;
    }
};

static long gcd_l(long a, long b) {
    while (b) {
        const long t = a % b;
        a = b;
        b = t;
    }
    return a;
}

// Polyphase tables for the current rates and coefficient table (none when they would not fit the LDS).
static int build_poly(bf_resampler *r) {
    if (r->d_poly) (void)hipFree(r->d_poly);
    if (r->d_count) (void)hipFree(r->d_count);
    r->d_poly = nullptr;
    r->d_count = nullptr;
    r->poly_L = 0;
    const long g = gcd_l(r->in_rate, r->out_rate);
    const long L = r->out_rate / g;
    if (L > 512) return BF_OK;
    const double ratio = r->ratio();
    const double float_increment = (double)r->index_inc * (ratio < 1.0 ? ratio : 1.0);
    const int increment = (int)llrint(float_increment * (double)(1 << kShift));
    const int max_filter_index = (r->table_len - 2) << kShift;
    int max_l = 0, max_r = 0;
    for (long m = 0; m < L; ++m) {  // the tap counts of poly_build_kernel
        const double frac = (double)(m * g) / (double)r->out_rate;
        const int start = (int)llrint(frac * float_increment * (double)(1 << kShift));
        const int nl = (max_filter_index - start) / increment + 1;
        const int nr = (max_filter_index - (increment - start)) / increment + 1;
        max_l = nl > max_l ? nl : max_l;
        max_r = nr > max_r ? nr : max_r;
    }
    const int row = max_l + max_r;
    if (L * row > kPolyLdsDoubles) return BF_OK;
    if (hipMalloc(&r->d_poly, sizeof(double) * (size_t)(L * row)) != hipSuccess) return BF_ENOMEM;
    if (hipMalloc(&r->d_count, sizeof(int2) * (size_t)L) != hipSuccess) return BF_ENOMEM;
    if (hipMemset(r->d_poly, 0, sizeof(double) * (size_t)(L * row)) != hipSuccess) return BF_EIO;
    PolyArgs q;
    q.poly = r->d_poly;
    q.count = r->d_count;
    q.L = (int)L;
    q.row = row;
    q.right_off = max_l;
    q.in_rate = r->in_rate;
    q.out_rate = r->out_rate;
    q.g = g;
    q.coeff_half_len = r->table_len - 2;
    q.index_inc = r->index_inc;
    q.table = r->d_table;
    BF_LAUNCH(poly_build_kernel, dim3((unsigned)L), dim3(64), 0, nullptr, q);
    if (hipDeviceSynchronize() != hipSuccess) return BF_EIO;
    r->poly_L = (int)L;
    r->poly_row = row;
    r->poly_right_off = max_l;
    r->poly_g = g;
    return BF_OK;
}

static int upload_table(bf_resampler *r, const float *coeffs, int len, int inc) {
    if (len < 4 || len > kMaxTable || inc < 1) return BF_EINVAL;
    float *d = nullptr;
    if (hipMalloc(&d, sizeof(float) * (size_t)len) != hipSuccess) return BF_ENOMEM;
    if (hipMemcpy(d, coeffs, sizeof(float) * (size_t)len, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        return BF_EIO;
    }
    const long old_hist = r->hist_len;
    if (r->d_table) (void)hipFree(r->d_table);
    r->d_table = d;
    r->table_len = len;
    r->index_inc = inc;
    const long hl = 2 * r->half_taps() + 8;
    if (hl != old_hist) {
        r->hist_len = 0;
        for (int b = 0; b < 2; ++b) {
            if (r->d_hist[b]) (void)hipFree(r->d_hist[b]);
            r->d_hist[b] = nullptr;
        }
        for (int b = 0; b < 2; ++b)
            if (hipMalloc(&r->d_hist[b], sizeof(float) * (size_t)hl) != hipSuccess) {
                r->d_hist[b] = nullptr;
                return BF_ENOMEM;  // process_* refuses to run until a set_table succeeds
            }
        r->hist_len = hl;
    }
    for (int b = 0; b < 2; ++b)
        if (hipMemset(r->d_hist[b], 0, sizeof(float) * (size_t)r->hist_len) != hipSuccess) return BF_EIO;
    r->hist_cur = 0;
    r->consumed = r->generated = 0;
    return build_poly(r);
}

extern "C" int bf_resampler_create(int in_rate, int out_rate, bf_resampler **out) {
    if (!out || in_rate <= 0 || out_rate <= 0) return BF_EINVAL;
    const double ratio = (double)out_rate / (double)in_rate;
    if (ratio < 1.0 / 256.0 || ratio > 256.0) return BF_EINVAL;  // src_is_valid_ratio
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return BF_ENODEV;
    bf_resampler *r = new (std::nothrow) bf_resampler;
    if (!r) return BF_ENOMEM;
    r->in_rate = in_rate;
    r->out_rate = out_rate;
    const std::vector<float> t = kaiser_sinc_table(kDefaultLen, kDefaultInc);
    const int rc = upload_table(r, t.data(), kDefaultLen, kDefaultInc);
    if (rc != BF_OK) {
        bf_resampler_destroy(r);
        return rc;
    }
    *out = r;
    return BF_OK;
}

extern "C" int bf_resampler_set_table(bf_resampler *r, const float *coeffs, int n_coeffs, int index_inc) {
    if (!r || !coeffs) return BF_EINVAL;
    std::lock_guard<std::mutex> lk(r->mu);
    return upload_table(r, coeffs, n_coeffs, index_inc);
}

static int rj_reset_locked(bf_resampler *r);

extern "C" int bf_resampler_reset(bf_resampler *r) {
    if (!r) return BF_EINVAL;
    std::lock_guard<std::mutex> lk(r->mu);
    for (int b = 0; b < 2; ++b)
        if (hipMemset(r->d_hist[b], 0, sizeof(float) * (size_t)r->hist_len) != hipSuccess) return BF_EIO;
    r->hist_cur = 0;
    r->consumed = r->generated = 0;
    return rj_reset_locked(r);
}

extern "C" size_t bf_resampler_out_count(bf_resampler *r, size_t n_in) {
    if (!r) return 0;
    std::lock_guard<std::mutex> lk(r->mu);
    return (size_t)(r->outputs_for(r->consumed + (long)n_in) - r->generated);
}

extern "C" int bf_resampler_latency(bf_resampler *r) {
    if (!r) return 0;
    std::lock_guard<std::mutex> lk(r->mu);
    return (int)r->half_taps();
}

static int process_locked(bf_resampler *r, const float *d_in, size_t n_in, float *d_out, size_t out_cap, size_t *n_out, hipStream_t s) {
    if (!r->d_table || !r->d_hist[0] || !r->d_hist[1]) return BF_ENOMEM;  // an earlier set_table ran out of memory half way
    const long total = r->outputs_for(r->consumed + (long)n_in);
    const long want = total - r->generated;
    if (n_out) *n_out = (size_t)want;
    if ((size_t)want > out_cap || (want > 0 && !d_out)) return BF_EINVAL;  // *n_out says how much room the call needs
    if (want > 0) {
        ResampleArgs a;
        a.hist = r->d_hist[r->hist_cur];
        a.in = d_in;
        a.out = d_out;
        a.hist_len = r->hist_len;
        a.n_in = (long)n_in;
        a.first_out = r->generated;
        a.in_base = r->consumed;
        a.n_out = want;
        a.in_rate = r->in_rate;
        a.out_rate = r->out_rate;
        a.coeff_half_len = r->table_len - 2;
        a.index_inc = r->index_inc;
        a.table_len = r->table_len;
        a.table = r->d_table;
        long blocks = (want + 255) / 256;
        if (r->poly_L > 0) {
            PolyArgs q;
            q.poly = r->d_poly;
            q.count = r->d_count;
            q.L = r->poly_L;
            q.row = r->poly_row;
            q.right_off = r->poly_right_off;
            q.in_rate = r->in_rate;
            q.out_rate = r->out_rate;
            q.g = r->poly_g;
            q.coeff_half_len = r->table_len - 2;
            q.index_inc = r->index_inc;
            q.table = r->d_table;
            if (blocks > 256 * 8) blocks = 256 * 8;  // every block copies the tables into its LDS first
            // input samples under one block's 256 outputs (+ both wings): staged in LDS when everything fits 64 KB of dynamic LDS
            const long win = (255 * r->in_rate) / r->out_rate + 2 + r->poly_row;
            const size_t lds_bytes = (size_t)r->poly_L * r->poly_row * sizeof(double) + (size_t)r->poly_L * sizeof(int2) + (size_t)win * sizeof(float);
            if (lds_bytes <= 64 * 1024)
                BF_LAUNCH(sinc_poly_lds_kernel, dim3((unsigned)blocks), dim3(256), lds_bytes, s, a, q, (int)win);
            else
                BF_LAUNCH(sinc_poly_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, q);
        } else {
            if (blocks > 256 * 32) blocks = 256 * 32;
            BF_LAUNCH(sinc_resample_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
        }
    }
    if (n_in > 0) {
        BF_LAUNCH(hist_roll_kernel, dim3((unsigned)((r->hist_len + 255) / 256)), dim3(256), 0, s, r->d_hist[r->hist_cur], d_in,
                           r->hist_len, (long)n_in, r->d_hist[r->hist_cur ^ 1]);
        r->hist_cur ^= 1;
    }
    r->consumed += (long)n_in;
    r->generated = total;
    return hipGetLastError() == hipSuccess ? BF_OK : BF_EIO;
}

extern "C" int bf_resampler_process_device(bf_resampler *r, const float *in_dev, size_t n_in, float *out_dev, size_t out_cap, size_t *n_out,
                                           void *hip_stream) {
    if (!r || (n_in && !in_dev)) return BF_EINVAL;
    std::lock_guard<std::mutex> lk(r->mu);
    if (r->mode != BF_RS_STREAM) return BF_EINVAL;  // BF_RS_ROSJACK: one bf_resampler_callback* per JACK period
    return process_locked(r, in_dev, n_in, out_dev, out_cap, n_out, (hipStream_t)hip_stream);
}

extern "C" int bf_resampler_process(bf_resampler *r, const float *in, size_t n_in, float *out, size_t out_cap, size_t *n_out) {
    if (!r || (n_in && !in)) return BF_EINVAL;
    std::lock_guard<std::mutex> lk(r->mu);
    if (r->mode != BF_RS_STREAM) return BF_EINVAL;
    const size_t want = (size_t)(r->outputs_for(r->consumed + (long)n_in) - r->generated);
    if (n_out) *n_out = want;
    if (want > out_cap || (want && !out)) return BF_EINVAL;
    if (n_in > r->cap_in) {
        if (r->d_in) (void)hipFree(r->d_in);
        r->d_in = nullptr;
        r->cap_in = 0;
        if (hipMalloc(&r->d_in, sizeof(float) * n_in) != hipSuccess) return BF_ENOMEM;
        r->cap_in = n_in;
    }
    if (want > r->cap_out) {
        if (r->d_out) (void)hipFree(r->d_out);
        r->d_out = nullptr;
        r->cap_out = 0;
        if (hipMalloc(&r->d_out, sizeof(float) * want) != hipSuccess) return BF_ENOMEM;
        r->cap_out = want;
    }
    if (n_in && hipMemcpy(r->d_in, in, sizeof(float) * n_in, hipMemcpyHostToDevice) != hipSuccess) return BF_EIO;
    const int rc = process_locked(r, r->d_in, n_in, r->d_out, want, nullptr, nullptr);
    if (rc != BF_OK) return rc;
    if (hipDeviceSynchronize() != hipSuccess) return BF_EIO;
    if (want && hipMemcpy(out, r->d_out, sizeof(float) * want, hipMemcpyDeviceToHost) != hipSuccess) return BF_EIO;
    return BF_OK;
}

// ---- BF_RS_ROSJACK: rosjack.cpp:311-349,416-436 around libsamplerate 0.1.9's src_process -----------------------------------------
// What the reference's stage does that a stream converter does not: (1) samplerate_data.output_frames = rosjack_window_size
// (rosjack.cpp:176-183), so one src_process returns at most one period of output; (2) the callback's period is copied into the
// converter's input only when input_frames == 0 (:314-320), otherwise it is DROPPED; (3) src_process pulls input into its buffer
// only when fewer than half_filter_chan_len + 1 samples are in hand and then as much as fits (src_sinc.c prepare_data), so that
// when the output rate is the higher one input_frames_used is often 0 and whole periods never reach the converter
// (16 -> 48 kHz keeps about every third period); (4) a block is published only when a full period of output is queued, at most
// one per callback (:340-349, :416-436).  The sample VALUES are the stream converter's: the output at position K of the stream of
// ACCEPTED periods does not depend on when it is computed, so the outputs are computed when their input is pulled in and parked
// in a device FIFO; the bookkeeping below only decides which periods enter and when blocks leave.
static long rj_prepare_len(bf_resampler *r) {  // prepare_data (libsamplerate 0.1.9), one channel
    const long half = r->half_taps();
    long len;
    if (r->b_current == 0) {  // initial state: zeros in front, then data
        len = r->b_len - 2 * half;
        r->b_current = r->b_end = half;
    } else if (r->b_end + half + 1 < r->b_len) {
        len = r->b_len - r->b_current - half;
    } else {  // what is left moves to the start of the buffer
        const long keep = r->b_end - r->b_current;
        r->b_current = half;
        r->b_end = half + keep;
        len = r->b_len - r->b_current - half;
    }
    return len > 0 ? len : 0;
}

static int rj_reset_locked(bf_resampler *r) {
    r->b_current = r->b_end = 0;
    r->pend_off = r->pend_n = 0;
    r->fifo_cur = 0;
    r->fifo_base = 0;
    r->released = r->emitted = 0;
    return BF_OK;
}

extern "C" int bf_resampler_set_mode(bf_resampler *r, int mode, int period) {
    if (!r || (mode != BF_RS_STREAM && mode != BF_RS_ROSJACK) || (mode == BF_RS_ROSJACK && (period < 1 || period > 65536))) return BF_EINVAL;
    {
        std::lock_guard<std::mutex> lk(r->mu);
        if (r->d_pend) (void)hipFree(r->d_pend);
        for (int b = 0; b < 2; ++b) {
            if (r->d_fifo[b]) (void)hipFree(r->d_fifo[b]);
            r->d_fifo[b] = nullptr;
        }
        r->d_pend = nullptr;
        r->mode = mode;
        r->period = period;
        if (mode == BF_RS_ROSJACK) {
            // sinc_set_converter (0.1.9): b_len = max(lrint(2.5 * coeff_half_len / index_inc * SRC_MAX_RATIO), 4096) * channels
            const long bl = lrint(2.5 * (double)(r->table_len - 2) / (double)r->index_inc * 256.0);
            r->b_len = bl > 4096 ? bl : 4096;
            const double up = r->ratio() > 1.0 ? r->ratio() : 1.0;
            r->fifo_cap = 2 * ((long)std::ceil((double)(r->b_len + period) * up) + 2 * (long)period);
            if (hipMalloc(&r->d_pend, sizeof(float) * (size_t)period) != hipSuccess) return BF_ENOMEM;
            for (int b = 0; b < 2; ++b)
                if (hipMalloc(&r->d_fifo[b], sizeof(float) * (size_t)r->fifo_cap) != hipSuccess) return BF_ENOMEM;
        }
    }
    return bf_resampler_reset(r);
}

// one output_to_rosjack(data_out, data_length = period) on device buffers
static int rj_callback_locked(bf_resampler *r, const float *d_period, float *d_block, int *emitted, int *accepted, hipStream_t s) {
    if (r->mode != BF_RS_ROSJACK || !r->d_pend || !r->d_fifo[0] || !r->d_fifo[1]) return BF_EINVAL;
    const long P = r->period, half = r->half_taps();
    int acc = 0;
    if (r->pend_n == 0) {  // rosjack.cpp:314-320
        if (hipMemcpyAsync(r->d_pend, d_period, sizeof(float) * (size_t)P, hipMemcpyDeviceToDevice, s) != hipSuccess) return BF_EIO;
        r->pend_off = 0;
        r->pend_n = P;
        acc = 1;
    }
    // src_process with output_frames = P (sinc_mono_vari_process' loop; positions in the exact arithmetic of this converter)
    long out_gen = 0;
    while (out_gen < P) {
        if (r->b_end - r->b_current <= half) {  // samples_in_hand <= half_filter_chan_len: reload
            long n = rj_prepare_len(r);
            if (n > r->pend_n) n = r->pend_n;
            if (n > 0) {
                // pull n samples in: every output they complete is computed now and parked behind the ones already there
                const long total = r->outputs_for(r->consumed + n), want = total - r->generated;
                long used = r->generated - r->fifo_base;  // floats in the FIFO
                if (used + want > r->fifo_cap) {          // compact: drop what has been published
                    const long live0 = r->emitted - r->fifo_base, live = used - live0;
                    if (live + want > r->fifo_cap) return BF_ENOMEM;  // cannot happen: fifo_cap covers a full converter buffer
                    if (live > 0 && hipMemcpyAsync(r->d_fifo[r->fifo_cur ^ 1], r->d_fifo[r->fifo_cur] + live0, sizeof(float) * (size_t)live,
                                                   hipMemcpyDeviceToDevice, s) != hipSuccess)
                        return BF_EIO;
                    r->fifo_cur ^= 1;
                    r->fifo_base = r->emitted;
                    used = live;
                }
                const int rc = process_locked(r, r->d_pend + r->pend_off, (size_t)n, r->d_fifo[r->fifo_cur] + used, (size_t)(r->fifo_cap - used),
                                              nullptr, s);
                if (rc != BF_OK) return rc;
                r->pend_off += n;
                r->pend_n -= n;
                r->b_end += n;
            }
            if (r->b_end - r->b_current <= half) break;
        }
        // one output sample; b_current advances by the integer step of the input position: cur(K + 1) - cur(K)
        const unsigned long long K = (unsigned long long)r->released;
        const long step = (long)(((K + 1) * (unsigned long long)r->in_rate) / (unsigned long long)r->out_rate -
                                 (K * (unsigned long long)r->in_rate) / (unsigned long long)r->out_rate);
        r->b_current += step;
        ++r->released;
        ++out_gen;
    }
    int em = 0;
    if (r->released - r->emitted >= P) {  // convert_to_sample_rate_ready (:340-349): one block per callback (:416-436)
        if (d_block && hipMemcpyAsync(d_block, r->d_fifo[r->fifo_cur] + (r->emitted - r->fifo_base), sizeof(float) * (size_t)P,
                                      hipMemcpyDeviceToDevice, s) != hipSuccess)
            return BF_EIO;
        r->emitted += P;
        em = 1;
    }
    if (emitted) *emitted = em;
    if (accepted) *accepted = acc;
    return BF_OK;
}

extern "C" int bf_resampler_callback_device(bf_resampler *r, const float *period_dev, float *block_dev, int *emitted, int *accepted,
                                            void *hip_stream) {
    if (!r || !period_dev || !block_dev) return BF_EINVAL;
    std::lock_guard<std::mutex> lk(r->mu);
    return rj_callback_locked(r, period_dev, block_dev, emitted, accepted, (hipStream_t)hip_stream);
}

extern "C" int bf_resampler_callback(bf_resampler *r, const float *period, float *block, int *emitted, int *accepted) {
    if (!r || !period || !block) return BF_EINVAL;
    std::lock_guard<std::mutex> lk(r->mu);
    if (r->mode != BF_RS_ROSJACK) return BF_EINVAL;
    const size_t P = (size_t)r->period;
    for (int k = 0; k < 2; ++k) {  // staging: d_in holds the period, d_out the block
        float **buf = k ? &r->d_out : &r->d_in;
        size_t *cap = k ? &r->cap_out : &r->cap_in;
        if (P > *cap) {
            if (*buf) (void)hipFree(*buf);
            *buf = nullptr;
            *cap = 0;
            if (hipMalloc(buf, sizeof(float) * P) != hipSuccess) return BF_ENOMEM;
            *cap = P;
        }
    }
    if (hipMemcpy(r->d_in, period, sizeof(float) * P, hipMemcpyHostToDevice) != hipSuccess) return BF_EIO;
    int em = 0;
    const int rc = rj_callback_locked(r, r->d_in, r->d_out, &em, accepted, nullptr);
    if (rc != BF_OK) return rc;
    if (hipDeviceSynchronize() != hipSuccess) return BF_EIO;
    if (em && hipMemcpy(block, r->d_out, sizeof(float) * P, hipMemcpyDeviceToHost) != hipSuccess) return BF_EIO;
    if (emitted) *emitted = em;
    return BF_OK;
}

extern "C" void bf_resampler_destroy(bf_resampler *r) {
    if (!r) return;
    if (r->d_pend) (void)hipFree(r->d_pend);
    for (int b = 0; b < 2; ++b)
        if (r->d_fifo[b]) (void)hipFree(r->d_fifo[b]);
    if (r->d_table) (void)hipFree(r->d_table);
    for (int b = 0; b < 2; ++b)
        if (r->d_hist[b]) (void)hipFree(r->d_hist[b]);
    if (r->d_in) (void)hipFree(r->d_in);
    if (r->d_out) (void)hipFree(r->d_out);
    if (r->d_poly) (void)hipFree(r->d_poly);
    if (r->d_count) (void)hipFree(r->d_count);
    delete r;
}

// The built-in table, for callers (and the tests' oracle) that want to look at it: returns the entry count.
extern "C" int bf_resampler_default_table(float *dst, int cap, int *index_inc) {
    if (index_inc) *index_inc = kDefaultInc;
    if (dst && cap >= kDefaultLen) {
        const std::vector<float> t = kaiser_sinc_table(kDefaultLen, kDefaultInc);
        memcpy(dst, t.data(), sizeof(float) * (size_t)kDefaultLen);
    }
    return kDefaultLen;
}
