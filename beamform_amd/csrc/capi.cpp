// capi.cpp -- the C ABI of libbfcore.so (include/bfcore.h) and the host-side
// state a reference node keeps in file-scope globals (util.h:24-50, das.cpp:15-25).
//
// Host work done here is start-up / control-plane only (geometry, steering
// weights in double precision, table upload, launch sizing); every per-frame
// operation runs in the gfx950 kernels.  There is no CPU compute fallback.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include <cxxabi.h>

#include "../../include/bfcore.h"
#include "geometry.hpp"
#include "kernels.hpp"
#include "launch_trace.hpp"
#include "pipeline.hpp"

using namespace bf;

static thread_local std::string g_last_error;

// ---- launch trace (launch_trace.hpp): the kernels the calling thread launches between bf_trace_begin and bf_trace_end ----------
static thread_local std::vector<const void *> *g_trace = nullptr;
void bf::trace_note(const void *host_fn) {
    if (g_trace) g_trace->push_back(host_fn);
}

struct bf_handle {
    bf_config cfg;
    int M = 0, H = 0, N = 0, S = 1, n_streams = 1;  // n_streams: input streams
    int n_dirs = 1, n_out = 1;                      // look directions per input stream; output streams = n_streams * n_dirs
    int device = 0, n_cus = 256;
    std::string err;

    // control plane (guarded by mu): what update_weights() owns in the reference
    std::mutex mu;
    ArrayGeometry geo;
    std::vector<double> freqs;
    std::vector<SteeringSet> steer;  // one per look direction
    std::vector<double> angle;       // /theta of every look direction
    std::vector<double> interf;
    bool tables_dirty = true;

    // fused-DAS device state
    f32x2 *d_gains[2] = {nullptr, nullptr};
    int gains_cur = 0;
    f32x2 *d_twiddle = nullptr;
    f32x2 *d_gains_il[2] = {nullptr, nullptr};  // hop < 512: das_pair_gains_interleaved tables (das_fused.hip, group mode), double-buffered with d_gains
    f32x2 *d_twiddle_1024 = nullptr;            // hop < 512: twiddle_table_32x32 (the frame-interleaving kernel runs the 1024-point machinery)
    float *d_window = nullptr;
    float *d_zeros = nullptr;
    float *d_hist[2] = {nullptr, nullptr};  // the hop before the next frame (the reference's ring buffer content)
    float *d_tail[2] = {nullptr, nullptr};
    int tail_cur = 0;  // index of the valid hist/tail pair; the kernel writes the other one
    f32x2 *d_sdump = nullptr;
    size_t sdump_cap = 0;
    double *d_sumsq = nullptr;  // bf_stream_rms scratch

    // bin pipeline (mvdr/lcmv/gss/phase/phasempf and DAS_BINS_F64)
    BinPipeline *pipe = nullptr;

    // staging for the host-buffer entry points
    float *d_x = nullptr, *d_y = nullptr;
    size_t d_x_cap = 0, d_y_cap = 0;
    // bf_process_batch pipelining: copies on their own streams, chunk by chunk, around the compute stream
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    std::vector<hipEvent_t> ev_in, ev_out;
    float *h_pin = nullptr;  // pinned [n_mics*hop | n_out*hop]: bf_process_hop stages through it (no pageable-copy detour)
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    // per-launch timing of the dominant kernel: a pool of event pairs, created on demand (bf_kernel_timing_begin reserves a batch so
    // that the caller's timed loop only records), reused by every session and destroyed with the handle
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    int timing = 0;  // 0: off; 1: bf_kernel_timing_begin .. _end; 2: inside bf_time_batch_device
    bool row0_written = true;  // reference-mic weight row: written by the cold start's update_weights(true), left zero by a
                               // structural interferer change (quirk Q3, lcmv.cpp:243-252,281,304)
};

namespace {

int fail(bf_handle *h, int code, const char *what, hipError_t e = hipSuccess) {
    std::string msg = what ? what : "";
    if (e != hipSuccess) {
        msg += ": ";
        msg += hipGetErrorString(e);
    }
    if (h) h->err = msg;
    g_last_error = msg;
    return code;
}

#define BF_HIP(h, call)                                                  \
    do {                                                                 \
        hipError_t e_ = (call);                                          \
        if (e_ != hipSuccess) return fail((h), BF_EIO, #call, e_);       \
    } while (0)

// fused fp32 das: the 512-frame period has the register-resident kernels (32 x 32 in-register FFT-1024, das_fused.hip); every
// other period (64 ... 4096 frames) one fused kernel on LDS-staged transforms (das_fused_gen.hip)
bool uses_fused_das(const bf_handle *h) { return h->cfg.algo == BF_DAS && h->cfg.das_impl == BF_DAS_FUSED_F32; }
bool fused_das_gen(const bf_handle *h) { return h->cfg.hop != 512; }

// update_weights(): recompute every steering column from the current angles.
void rebuild_steering(bf_handle *h, bool first, int only_dir = -1) {
    for (int d = 0; d < h->n_dirs; ++d) {
        if (only_dir >= 0 && d != only_dir) continue;
        h->steer[d].update_column(h->geo, h->freqs, 0, h->angle[d], first);
        for (int k = 0; k < h->S - 1; ++k) h->steer[d].update_column(h->geo, h->freqs, k + 1, h->interf[k], first);
    }
    h->tables_dirty = true;
}

// Upload whatever set_theta changed; stream-ordered, double-buffered so a batch
// already in flight keeps reading the table it was launched with.
int sync_tables(bf_handle *h, hipStream_t s, RunSnapshot *snap) {
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->tables_dirty) {
        if (h->pipe) *snap = h->pipe->snapshot_for_run();
        return BF_OK;
    }
    if (uses_fused_das(h)) {
        const int np = (h->M + 1) / 2;
        std::vector<f32x2> g, gil;  // [dir][pair][1024]
        for (int d = 0; d < h->n_dirs; ++d) {
            if (h->d_gains_il[0]) {
                const std::vector<f32x2> gi = das_pair_gains_interleaved(h->steer[d], np);
                gil.insert(gil.end(), gi.begin(), gi.end());
            }
            const std::vector<f32x2> gd = fused_das_gen(h) ? das_pair_gains_natural(h->steer[d], np) : das_pair_gains(h->steer[d], np);
            g.insert(g.end(), gd.begin(), gd.end());
        }
        const int nxt = h->gains_cur ^ 1;
        BF_HIP(h, hipMemcpyAsync(h->d_gains[nxt], g.data(), g.size() * sizeof(f32x2), hipMemcpyHostToDevice, s));
        if (h->d_gains_il[0])
            BF_HIP(h, hipMemcpyAsync(h->d_gains_il[nxt], gil.data(), gil.size() * sizeof(f32x2), hipMemcpyHostToDevice, s));
        BF_HIP(h, hipStreamSynchronize(s));  // pageable staging vectors go out of scope
        h->gains_cur = nxt;
    }
    if (h->pipe) {
        int rc = h->pipe->upload_steering(h->steer, s);
        if (rc != BF_OK) return fail(h, rc, h->pipe->error().c_str());
    }
    h->tables_dirty = false;
    // one consistent view of {columns, table, pending gss resets} for this batch, taken under the same lock
    if (h->pipe) *snap = h->pipe->snapshot_for_run();
    return BF_OK;
}

// Event pair for the next launch of the dominant kernel (both null when no timing session is active).
int timing_reserve(bf_handle *h, size_t n) {
    while (h->ev_pool.size() < n) {
        hipEvent_t k0 = nullptr, k1 = nullptr;
        BF_HIP(h, hipEventCreate(&k0));
        hipError_t e = hipEventCreate(&k1);
        if (e != hipSuccess) {
            (void)hipEventDestroy(k0);
            return fail(h, BF_EIO, "hipEventCreate", e);
        }
        h->ev_pool.push_back(std::make_pair(k0, k1));
    }
    return BF_OK;
}
int timing_acquire(bf_handle *h, hipEvent_t *k0, hipEvent_t *k1) {
    *k0 = *k1 = nullptr;
    if (!h->timing) return BF_OK;
    int rc = timing_reserve(h, h->ev_used + 1);
    if (rc != BF_OK) return rc;
    *k0 = h->ev_pool[h->ev_used].first;
    *k1 = h->ev_pool[h->ev_used].second;
    ++h->ev_used;
    return BF_OK;
}
// mean duration of the pairs recorded since the session began; ends the session
int timing_collect(bf_handle *h, float *ms_mean, int *n_launches) {
    float sum = 0.f;
    int n = 0;
    hipError_t e = hipSuccess;
    for (size_t i = 0; i < h->ev_used; ++i) {
        float t = 0.f;
        if (e == hipSuccess) e = hipEventSynchronize(h->ev_pool[i].second);
        if (e == hipSuccess) e = hipEventElapsedTime(&t, h->ev_pool[i].first, h->ev_pool[i].second);
        if (e == hipSuccess) { sum += t; ++n; }
    }
    h->ev_used = 0;
    h->timing = 0;
    if (e != hipSuccess) return fail(h, BF_EIO, "kernel timing", e);
    *ms_mean = n ? sum / (float)n : 0.f;
    if (n_launches) *n_launches = n;
    return BF_OK;
}

int run_das_fused(bf_handle *h, const float *x_dev, size_t n_frames, float *y_dev, void *spectrum_dev, hipStream_t s,
                  int layout, long mic_stride) {
    const long F = (long)n_frames;
    const int S = h->n_out;
    // one block (16 half-wavefronts) per run of consecutive frames; runs are multiples of 16 frames and
    // there are about as many runs as CUs
    const bool gen = fused_das_gen(h);
    // (periods 256 / 1024: several 256-thread blocks share a CU -- 13 / 52 KB of LDS each -- and a run costs one recomputed frame)
    // several look directions, planar input, <= 8 microphones, no dump: one set of forward transforms per frame serves up to 16
    // directions (das_fused_dirs_kernel); BF_DAS_SHARED_DIRS = the smallest direction count that takes it (0: never)
    static const int shared_min = getenv("BF_DAS_SHARED_DIRS") ? atoi(getenv("BF_DAS_SHARED_DIRS")) : 6;
    const bool shared = !gen && layout == BF_PLANAR && h->M <= 8 && !spectrum_dev && shared_min > 0 && h->n_dirs >= shared_min;
    // (generic periods: blocks of 13 N bytes of LDS -- 26 N at N = 8192 -- share a CU: 8 at N <= 512, 3 at 2048, 1 from 4096 on)
    const int gen_per_cu = h->N <= 512 ? 8 : h->N <= 1024 ? 6 : h->N <= 2048 ? 3 : 1;
    // period 1024 without a dump: ONE 2048-point transform per frame on a full wavefront, eight frames in flight per block and the tails
    // through an LDS ring (das_fused.hip das_fused_wave2048_kernel); BF_DAS_SPLIT2048=0: the generic kernel (cross-checks)
    static const int split_env = getenv("BF_DAS_SPLIT2048") ? atoi(getenv("BF_DAS_SPLIT2048")) : 3;
    const bool wave2048 = gen && h->N == 2048 && !spectrum_dev && split_env != 0;
    // periods below 512 without a dump: 1024 / N frames interleaved into one pass of the 1024-point machinery -- the period-512 kernel
    // itself in group mode: one block per run, tails through its LDS ring, HBM sees every hop once; BF_DAS_INTERLEAVE=0: the generic
    // kernel (cross-checks)
    static const int il_env = getenv("BF_DAS_INTERLEAVE") ? atoi(getenv("BF_DAS_INTERLEAVE")) : 1;
    const bool small_ring = gen && h->N < 1024 && !spectrum_dev && il_env != 0 && h->d_gains_il[0] != nullptr && h->d_twiddle_1024 != nullptr;
    const long Rg = small_ring ? 1024 / h->N : 1;
    long runs = (small_ring || wave2048 ? (long)h->n_cus : gen ? (long)h->n_cus * gen_per_cu : (long)h->n_cus) / (shared ? h->n_streams : S);
    if (runs < 1) runs = 1;
    long fpc = (F + runs - 1) / runs;
    if (!gen) fpc = ((fpc + 15) / 16) * 16;
    if (wave2048) fpc = ((fpc + 7) / 8) * 8;                              // eight frames per pass of a block
    else if (small_ring) fpc = ((fpc + 16 * Rg - 1) / (16 * Rg)) * (16 * Rg);  // sixteen groups per pass of a block
    const long cps = (F + fpc - 1) / fpc;

    if (spectrum_dev) {
        const size_t need = (size_t)S * F * h->N;
        if (need > h->sdump_cap) {
            if (h->d_sdump) (void)hipFree(h->d_sdump);
            h->d_sdump = nullptr;
            h->sdump_cap = 0;
            BF_HIP(h, hipMalloc((void **)&h->d_sdump, need * sizeof(f32x2)));
            h->sdump_cap = need;
        }
    }

    DasFusedArgs a;
    a.x = x_dev;
    a.hist_in = h->d_hist[h->tail_cur];
    a.hist_out = h->d_hist[h->tail_cur ^ 1];
    a.y = y_dev;
    a.tail_in = h->d_tail[h->tail_cur];
    a.tail_out = h->d_tail[h->tail_cur ^ 1];
    a.gains = small_ring ? h->d_gains_il[h->gains_cur] : h->d_gains[h->gains_cur];
    a.twiddle = small_ring ? h->d_twiddle_1024 : h->d_twiddle;
    a.window = h->d_window;
    a.zeros = h->d_zeros;
    a.sdump = spectrum_dev ? h->d_sdump : nullptr;
    a.n_frames = F;
    a.mic_stride = mic_stride;
    a.stream_stride_x = (long)h->M * F * h->H;
    a.n_streams = S;
    a.n_dirs = h->n_dirs;
    a.n_mics = h->M;
    a.frames_per_chunk = (int)fpc;
    a.chunks_per_stream = (int)cps;
    a.layout = layout;
    a.group = small_ring ? (int)Rg : 1;
    if (wave2048) BF_HIP(h, prepare_das_fused_wave2048(a, s));
    else if (!gen || small_ring) BF_HIP(h, prepare_das_fused(a, s));
    hipEvent_t k0 = nullptr, k1 = nullptr;
    {
        int trc = timing_acquire(h, &k0, &k1);
        if (trc != BF_OK) return trc;
    }
    // (a launch that fails between the two records must not leave a half-recorded pair in the session: hand it back)
    auto launch_all = [&]() -> hipError_t {
        hipError_t e = hipSuccess;
        if (k0) e = hipEventRecord(k0, s);
        if (e != hipSuccess) return e;
        if (shared) {
            for (int d0 = 0; d0 < h->n_dirs && e == hipSuccess; d0 += 16)
                e = launch_das_fused_dirs(a, d0, h->n_dirs - d0 < 16 ? h->n_dirs - d0 : 16, s);
        } else {
            e = wave2048 ? launch_das_fused_wave2048(a, s)
                : small_ring ? launch_das_fused(a, s)
                : gen     ? launch_das_fused_gen(a, h->N, s)
                : launch_das_fused(a, s);
        }
        if (e == hipSuccess && k1) e = hipEventRecord(k1, s);
        return e;
    };
    {
        const hipError_t e = launch_all();
        if (e != hipSuccess) {
            if (k0 && h->ev_used > 0) --h->ev_used;
            BF_HIP(h, e);
        }
    }
    h->tail_cur ^= 1;

    if (spectrum_dev)
        BF_HIP(h, gen ? launch_das_hermitian_dump_gen(h->d_sdump, (f64x2 *)spectrum_dev, (long)S * F, h->N, s)
                      : launch_das_hermitian_dump(h->d_sdump, (f64x2 *)spectrum_dev, (long)S * F, s));
    return BF_OK;
}

int run_batch_device(bf_handle *h, const float *x_dev, size_t n_frames, float *y_dev, void *spectrum_dev, hipStream_t s,
                     int layout, long mic_stride) {
    if (n_frames == 0) return BF_OK;
    RunSnapshot snap;
    int rc = sync_tables(h, s, &snap);
    if (rc != BF_OK) return rc;
    if (uses_fused_das(h)) return run_das_fused(h, x_dev, n_frames, y_dev, spectrum_dev, s, layout, mic_stride);
    // the pipeline brackets its dominant kernel with this pair when it has one (das fp64 in one launch); otherwise the pair stays
    // unrecorded and the session's mean is taken over nothing (callers fall back to the call time)
    hipEvent_t k0 = nullptr, k1 = nullptr;
    rc = timing_acquire(h, &k0, &k1);
    if (rc != BF_OK) return rc;
    h->pipe->kev0 = k0;
    h->pipe->kev1 = k1;
    h->pipe->kev_recorded = false;
    rc = h->pipe->run(x_dev, (long)n_frames, y_dev, (f64x2 *)spectrum_dev, s, layout, mic_stride, snap);
    h->pipe->kev0 = h->pipe->kev1 = nullptr;
    if (k0 && !h->pipe->kev_recorded) --h->ev_used;  // nothing was bracketed: hand the pair back
    if (rc != BF_OK) return fail(h, rc, h->pipe->error().c_str());
    return BF_OK;
}

int ensure_staging(bf_handle *h, size_t x_elems, size_t y_elems) {
    if (x_elems > h->d_x_cap) {
        if (h->d_x) (void)hipFree(h->d_x);
        h->d_x = nullptr;
        h->d_x_cap = 0;
        BF_HIP(h, hipMalloc((void **)&h->d_x, x_elems * sizeof(float)));
        h->d_x_cap = x_elems;
    }
    if (y_elems > h->d_y_cap) {
        if (h->d_y) (void)hipFree(h->d_y);
        h->d_y = nullptr;
        h->d_y_cap = 0;
        BF_HIP(h, hipMalloc((void **)&h->d_y, y_elems * sizeof(float)));
        h->d_y_cap = y_elems;
    }
    return BF_OK;
}

}  // namespace

extern "C" {

const char *bf_version(void) { return "bfcore-mi355x 0.1 (gfx950)"; }

const char *bf_strerror(int code) {
    switch (code) {
        case BF_OK: return "ok";
        case BF_EINVAL: return "invalid argument or configuration";
        case BF_ENOMEM: return "out of memory";
        case BF_ENODEV: return "no usable HIP device (this library has no CPU fallback)";
        case BF_ENOSYS: return "algorithm or variant not available in this build";
        case BF_EIO: return "HIP runtime error";
        case BF_ENOENT: return "file not found";
        default: return "unknown error";
    }
}

const char *bf_last_error(const bf_handle *h) { return h ? h->err.c_str() : g_last_error.c_str(); }

int bf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int bf_create(const bf_config *cfg, bf_handle **out) {
    if (!cfg || !out) return BF_EINVAL;
    *out = nullptr;
    if (cfg->algo < BF_DAS || cfg->algo > BF_GSC) return fail(nullptr, BF_EINVAL, "algo out of range");
    if (cfg->n_mics < 1 || cfg->n_mics > BF_MAX_MICS) return fail(nullptr, BF_EINVAL, "n_mics out of range");
    if (cfg->hop < 64 || cfg->hop > 4096 || (cfg->hop & (cfg->hop - 1)) != 0)  // jackd -p takes powers of two; rosjack.cpp:131-134 whatever it reports
        return fail(nullptr, BF_ENOSYS, "hop (JACK period) must be a power of two from 64 to 4096 frames (fft_win 128 ... 8192)");
    if (cfg->n_streams < 1) return fail(nullptr, BF_EINVAL, "n_streams < 1");
    if (cfg->n_dirs < 0 || cfg->n_dirs > BF_MAX_DIRS) return fail(nullptr, BF_EINVAL, "n_dirs out of range");
    if (cfg->n_dirs > 1 && (cfg->algo == BF_MCRA || cfg->algo == BF_GSC))
        return fail(nullptr, BF_ENOSYS, "look-direction batches: mcra has no look direction, gsc is not built for them");
    if (cfg->n_interf < 0 || cfg->n_interf > BF_MAX_INTERF) return fail(nullptr, BF_EINVAL, "n_interf out of range");
    if (cfg->layout != BF_PLANAR && cfg->layout != BF_INTERLEAVED) return fail(nullptr, BF_EINVAL, "layout");
    if (cfg->precision != BF_PRECISION_REFERENCE && cfg->precision != BF_PRECISION_MIXED) return fail(nullptr, BF_EINVAL, "precision");
    if (cfg->das_impl != BF_DAS_FUSED_F32 && cfg->das_impl != BF_DAS_F64) return fail(nullptr, BF_EINVAL, "das_impl");
    int ndev = bf_device_count();
    if (ndev <= 0) return fail(nullptr, BF_ENODEV, "no HIP device visible; libbfcore has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, BF_ENODEV, "device ordinal out of range");

    bf_handle *h = new bf_handle();
    h->cfg = *cfg;
    h->M = cfg->n_mics;
    h->H = cfg->hop;
    h->N = 2 * cfg->hop;
    const bool multi = (cfg->algo == BF_LCMV || cfg->algo == BF_GSS);
    h->S = multi ? cfg->n_interf + 1 : 1;
    h->n_streams = cfg->n_streams;
    h->n_dirs = cfg->n_dirs > 1 ? cfg->n_dirs : 1;
    h->cfg.n_dirs = h->n_dirs;
    h->n_out = h->n_streams * h->n_dirs;
    h->device = cfg->device;
    h->angle.assign(h->n_dirs, cfg->theta);
    for (int k = 0; k < h->S - 1; ++k) h->interf.push_back(cfg->interf_angle[k]);

#define BF_CREATE_HIP(call)                                   \
    do {                                                      \
        hipError_t e_ = (call);                               \
        if (e_ != hipSuccess) {                               \
            int rc_ = fail(nullptr, BF_EIO, #call, e_);       \
            bf_destroy(h);                                    \
            return rc_;                                       \
        }                                                     \
    } while (0)

    BF_CREATE_HIP(hipSetDevice(h->device));
    hipDeviceProp_t prop;
    BF_CREATE_HIP(hipGetDeviceProperties(&prop, h->device));
    h->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;

    // handle_params + calculate_frequency_vector + update_weights(true)
    h->geo.set(cfg->mic_x, cfg->mic_y, h->M);
    h->freqs = frequency_vector(h->N, cfg->sample_rate);
    h->steer.resize(h->n_dirs);
    for (auto &st : h->steer) st.allocate(h->N, h->M, h->S);
    rebuild_steering(h, true);
    h->row0_written = true;

    BF_CREATE_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    BF_CREATE_HIP(hipEventCreate(&h->ev0));
    BF_CREATE_HIP(hipEventCreate(&h->ev1));

    const size_t S = h->n_streams, So = h->n_out;
    if (uses_fused_das(h)) {
        const size_t gsz = (size_t)((h->M + 1) / 2) * h->N * h->n_dirs;
        BF_CREATE_HIP(hipMalloc((void **)&h->d_gains[0], gsz * sizeof(f32x2)));
        BF_CREATE_HIP(hipMalloc((void **)&h->d_gains[1], gsz * sizeof(f32x2)));
        std::vector<f32x2> tw = twiddle_table_32x32<f32x2>();
        if (fused_das_gen(h)) tw = stockham_twiddles<f32x2>(h->N);  // W^m, m < N/2, + the per-pass radix-4 blocks (geometry.hpp)
        if (h->N < 1024) {
            const std::vector<f32x2> t32 = twiddle_table_32x32<f32x2>();
            BF_CREATE_HIP(hipMalloc((void **)&h->d_twiddle_1024, t32.size() * sizeof(f32x2)));
            BF_CREATE_HIP(hipMemcpy(h->d_twiddle_1024, t32.data(), t32.size() * sizeof(f32x2), hipMemcpyHostToDevice));
            const size_t gil = (size_t)((h->M + 1) / 2) * 1024 * h->n_dirs;
            for (int i = 0; i < 2; ++i) BF_CREATE_HIP(hipMalloc((void **)&h->d_gains_il[i], gil * sizeof(f32x2)));
        }
        BF_CREATE_HIP(hipMalloc((void **)&h->d_twiddle, tw.size() * sizeof(f32x2)));
        BF_CREATE_HIP(hipMemcpy(h->d_twiddle, tw.data(), tw.size() * sizeof(f32x2), hipMemcpyHostToDevice));
        std::vector<double> hd = sqrt_hann(h->N);
        std::vector<float> hf(h->N);
        for (int i = 0; i < h->N; ++i) hf[i] = (float)hd[i];
        BF_CREATE_HIP(hipMalloc((void **)&h->d_window, hf.size() * sizeof(float)));
        BF_CREATE_HIP(hipMemcpy(h->d_window, hf.data(), hf.size() * sizeof(float), hipMemcpyHostToDevice));
        BF_CREATE_HIP(hipMalloc((void **)&h->d_zeros, 2048 * sizeof(float)));
        BF_CREATE_HIP(hipMemset(h->d_zeros, 0, 2048 * sizeof(float)));
        BF_CREATE_HIP(hipMalloc((void **)&h->d_hist[0], S * h->M * h->H * sizeof(float)));
        BF_CREATE_HIP(hipMalloc((void **)&h->d_hist[1], S * h->M * h->H * sizeof(float)));
        BF_CREATE_HIP(hipMalloc((void **)&h->d_tail[0], So * h->H * sizeof(float)));
        BF_CREATE_HIP(hipMalloc((void **)&h->d_tail[1], So * h->H * sizeof(float)));
    } else {
        h->pipe = BinPipeline::create(h->cfg, h->n_cus);
        if (!h->pipe) {
            int rc = fail(nullptr, BF_ENOSYS, "bin pipeline for this algorithm is not built");
            bf_destroy(h);
            return rc;
        }
        int rc = h->pipe->init();
        if (rc != BF_OK) {
            fail(nullptr, rc, h->pipe->error().c_str());
            bf_destroy(h);
            return rc;
        }
    }
#undef BF_CREATE_HIP
    int rc = bf_reset(h);
    if (rc != BF_OK) {
        bf_destroy(h);
        return rc;
    }
    *out = h;
    return BF_OK;
}

void bf_destroy(bf_handle *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (int i = 0; i < 2; ++i) {
        if (h->d_gains[i]) (void)hipFree(h->d_gains[i]);
        if (h->d_tail[i]) (void)hipFree(h->d_tail[i]);
        if (h->d_hist[i]) (void)hipFree(h->d_hist[i]);
    }
    if (h->d_twiddle) (void)hipFree(h->d_twiddle);
    if (h->d_twiddle_1024) (void)hipFree(h->d_twiddle_1024);
    for (int i = 0; i < 2; ++i)
        if (h->d_gains_il[i]) (void)hipFree(h->d_gains_il[i]);
    if (h->d_window) (void)hipFree(h->d_window);
    if (h->d_zeros) (void)hipFree(h->d_zeros);
    if (h->d_sdump) (void)hipFree(h->d_sdump);
    if (h->d_sumsq) (void)hipFree(h->d_sumsq);
    if (h->d_x) (void)hipFree(h->d_x);
    if (h->d_y) (void)hipFree(h->d_y);
    if (h->h_pin) (void)hipHostFree(h->h_pin);
    for (auto &pr : h->ev_pool) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    for (hipEvent_t e : h->ev_in) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->ev_out) (void)hipEventDestroy(e);
    if (h->s_h2d) (void)hipStreamDestroy(h->s_h2d);
    if (h->s_d2h) (void)hipStreamDestroy(h->s_d2h);
    delete h->pipe;
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

// The clears are enqueued on `s` (hipMemsetAsync): they order against the batches the caller runs on the same stream -- a plain
// hipMemset goes to the NULL stream, which does not order against non-blocking streams (every torch.cuda.Stream is one).
int bf_reset_async(bf_handle *h, void *hip_stream) {
    if (!h) return BF_EINVAL;
    BF_HIP(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)hip_stream;
    const size_t S = h->n_streams;
    if (uses_fused_das(h)) {
        // prepare_overlap_and_add: ring pre-filled with one hop of zeros, out_buff calloc'ed (util.h:272-286)
        BF_HIP(h, hipMemsetAsync(h->d_hist[0], 0, S * h->M * h->H * sizeof(float), s));
        BF_HIP(h, hipMemsetAsync(h->d_hist[1], 0, S * h->M * h->H * sizeof(float), s));
        BF_HIP(h, hipMemsetAsync(h->d_tail[0], 0, (size_t)h->n_out * h->H * sizeof(float), s));
        BF_HIP(h, hipMemsetAsync(h->d_tail[1], 0, (size_t)h->n_out * h->H * sizeof(float), s));
        h->tail_cur = 0;
    } else if (h->pipe) {
        std::lock_guard<std::mutex> lk(h->mu);  // reset() re-arms the gss demixing reset, which /theta also writes
        int rc = h->pipe->reset(s);
        if (rc != BF_OK) return fail(h, rc, h->pipe->error().c_str());
    }
    return BF_OK;
}

// Host-synchronous form: the device is idle with respect to this handle before and after.
int bf_reset(bf_handle *h) {
    if (!h) return BF_EINVAL;
    BF_HIP(h, hipSetDevice(h->device));
    BF_HIP(h, hipDeviceSynchronize());
    const int rc = bf_reset_async(h, nullptr);
    if (rc != BF_OK) return rc;
    BF_HIP(h, hipStreamSynchronize(nullptr));
    return BF_OK;
}

int bf_set_theta_dir(bf_handle *h, int dir, double degrees) {
    if (!h) return BF_EINVAL;
    if (dir < 0 || dir >= h->n_dirs) return fail(h, BF_EINVAL, "look direction index out of range");
    std::lock_guard<std::mutex> lk(h->mu);
    h->angle[dir] = degrees;
    rebuild_steering(h, false, dir);
    if (h->pipe) h->pipe->on_theta_changed(dir);  // gss resets that beam's demixing matrices (gss.cpp:90-93)
    return BF_OK;
}

int bf_set_theta(bf_handle *h, double degrees) { return bf_set_theta_dir(h, 0, degrees); }

int bf_set_thetas(bf_handle *h, const double *degrees, int n) {
    if (!h || !degrees) return BF_EINVAL;
    if (n < 1 || n > h->n_dirs) return fail(h, BF_EINVAL, "more angles than look directions");
    std::lock_guard<std::mutex> lk(h->mu);
    for (int d = 0; d < n; ++d) {
        h->angle[d] = degrees[d];
        rebuild_steering(h, false, d);
        if (h->pipe) h->pipe->on_theta_changed(d);
    }
    return BF_OK;
}

int bf_stream_rms(bf_handle *h, const float *y_dev, size_t n_frames, double *rms_host, void *hip_stream) {
    if (!h || !y_dev || !rms_host || n_frames == 0) return BF_EINVAL;
    BF_HIP(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)hip_stream;
    if (!h->d_sumsq) BF_HIP(h, hipMalloc((void **)&h->d_sumsq, sizeof(double) * h->n_out));
    const long n = (long)n_frames * h->H;
    BF_HIP(h, launch_stream_rms(y_dev, n, h->n_out, h->d_sumsq, s));
    BF_HIP(h, hipMemcpyAsync(rms_host, h->d_sumsq, sizeof(double) * h->n_out, hipMemcpyDeviceToHost, s));
    BF_HIP(h, hipStreamSynchronize(s));
    for (int i = 0; i < h->n_out; ++i) rms_host[i] = std::sqrt(rms_host[i] / (double)n);  // energy2theta.py:23-27
    return BF_OK;
}

int bf_set_interference(bf_handle *h, unsigned id, double degrees) {
    if (!h) return BF_EINVAL;
    if (h->cfg.algo != BF_LCMV && h->cfg.algo != BF_GSS) return fail(h, BF_EINVAL, "node has no interferers");
    std::lock_guard<std::mutex> lk(h->mu);
    std::vector<double> &ia = h->interf;
    const double thr = h->cfg.interf_angle_threshold;
    bool structural = false;
    if (id >= 1 && id <= ia.size()) {  // lcmv.cpp:259-281
        ia[id - 1] = degrees;
        for (size_t i = 0; i < ia.size(); ++i) {
            if (i != (id - 1) && std::abs(ia[i] - degrees) < thr) {
                ia.erase(ia.begin() + id - 1);
                structural = true;
                break;
            }
        }
    } else if (id > ia.size()) {  // lcmv.cpp:282-305
        size_t i;
        for (i = 0; i < ia.size(); ++i)
            if (std::abs(ia[i] - degrees) < thr) break;
        if (i != ia.size()) return BF_OK;  // too close to an existing interferer: ignored, as the reference does
        if (ia.size() + 1 > BF_MAX_INTERF) return fail(h, BF_ENOSYS, "per-bin kernels are built for up to 15 interferers");
        ia.push_back(degrees);
        structural = true;
    } else {
        return fail(h, BF_EINVAL, "interference id must be >= 1");
    }
    if (structural) {
        // free_interf_buffers + allocate_interf_buffers: weights come back zeroed, row 0 is not rewritten (Q3)
        h->S = (int)ia.size() + 1;
        for (auto &st : h->steer) st.allocate(h->N, h->M, h->S);
        h->row0_written = false;
        if (h->pipe) h->pipe->set_columns(h->S);
    }
    rebuild_steering(h, false);
    if (h->pipe) h->pipe->on_theta_changed();  // gss: sep_matrix = weights^H (gss.cpp:90-93)
    return BF_OK;
}

int bf_n_interferers(bf_handle *h) {
    if (!h) return BF_EINVAL;
    std::lock_guard<std::mutex> lk(h->mu);
    return (int)h->interf.size();
}

int bf_get_weights(bf_handle *h, double *w_host) {
    if (!h || !w_host) return BF_EINVAL;
    std::lock_guard<std::mutex> lk(h->mu);
    memcpy(w_host, h->steer[0].w.data(), h->steer[0].w.size() * sizeof(cplxd));  // look direction 0
    return BF_OK;
}

int bf_process_batch_device(bf_handle *h, const float *x_dev, size_t n_frames, float *y_dev, void *spectrum_dev,
                            void *hip_stream) {
    if (!h || !x_dev || !y_dev) return BF_EINVAL;
    BF_HIP(h, hipSetDevice(h->device));
    return run_batch_device(h, x_dev, n_frames, y_dev, spectrum_dev, (hipStream_t)hip_stream, h->cfg.layout,
                            (long)n_frames * h->H);
}

void *bf_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void bf_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int bf_process_batch(bf_handle *h, const float *x_host, size_t n_frames, float *y_host) {
    if (!h || !x_host || !y_host) return BF_EINVAL;
    if (n_frames == 0) return BF_OK;
    BF_HIP(h, hipSetDevice(h->device));
    const size_t xe = (size_t)h->n_streams * h->M * n_frames * h->H;
    const size_t ye = (size_t)h->n_out * n_frames * h->H;
    int rc = ensure_staging(h, xe, ye);
    if (rc != BF_OK) return rc;
    constexpr int kChunks = 8;
    if (h->n_out == 1 && n_frames >= 8192) {
        // One stream of a long batch: H2D of chunk c+1, compute of chunk c and D2H of chunk c-1 overlap (three HIP streams,
        // events in between).  Consecutive chunks are consecutive batches of the same stream, so the carried state does the
        // rest.  (Only page-locked host buffers -- bf_host_alloc -- make the copies truly asynchronous.)
        if (!h->s_h2d) {
            // all-or-nothing: a half-built set must not survive a failed create (later calls would record on null events)
            hipStream_t a = nullptr, b = nullptr;
            std::vector<hipEvent_t> ei(kChunks, nullptr), eo(kChunks, nullptr);
            hipError_t e = hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
            for (int c = 0; c < kChunks && e == hipSuccess; ++c) {
                e = hipEventCreateWithFlags(&ei[c], hipEventDisableTiming);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&eo[c], hipEventDisableTiming);
            }
            if (e != hipSuccess) {
                for (hipEvent_t v : ei) if (v) (void)hipEventDestroy(v);
                for (hipEvent_t v : eo) if (v) (void)hipEventDestroy(v);
                if (a) (void)hipStreamDestroy(a);
                if (b) (void)hipStreamDestroy(b);
                return fail(h, BF_EIO, "bf_process_batch: copy streams / events", e);
            }
            h->s_h2d = a;
            h->s_d2h = b;
            h->ev_in.swap(ei);
            h->ev_out.swap(eo);
        }
        const size_t F = n_frames, H = (size_t)h->H, M = (size_t)h->M;
        const size_t cf = (F + kChunks - 1) / kChunks;
        hipError_t e = hipSuccess;
        rc = BF_OK;
        const char *what = "";
#define BF_CHUNK(call)                                   \
        if (e == hipSuccess && rc == BF_OK) {            \
            e = (call);                                  \
            if (e != hipSuccess) what = #call;           \
        }
        for (int c = 0; c < kChunks && e == hipSuccess && rc == BF_OK; ++c) {
            const size_t c0 = (size_t)c * cf;
            if (c0 >= F) break;
            const size_t n = (c0 + cf <= F) ? cf : F - c0;
            const float *xd;
            if (h->cfg.layout == BF_PLANAR) {  // M rows of F*H samples: a chunk is a column block
                BF_CHUNK(hipMemcpy2DAsync(h->d_x + c0 * H, F * H * sizeof(float), x_host + c0 * H, F * H * sizeof(float),
                                          n * H * sizeof(float), M, hipMemcpyHostToDevice, h->s_h2d));
                xd = h->d_x + c0 * H;
            } else {
                BF_CHUNK(hipMemcpyAsync(h->d_x + c0 * H * M, x_host + c0 * H * M, n * H * M * sizeof(float), hipMemcpyHostToDevice,
                                        h->s_h2d));
                xd = h->d_x + c0 * H * M;
            }
            BF_CHUNK(hipEventRecord(h->ev_in[c], h->s_h2d));
            BF_CHUNK(hipStreamWaitEvent(h->stream, h->ev_in[c], 0));
            if (e == hipSuccess)
                rc = run_batch_device(h, xd, n, h->d_y + c0 * H, nullptr, h->stream, h->cfg.layout, (long)(F * H));
            BF_CHUNK(hipEventRecord(h->ev_out[c], h->stream));
            BF_CHUNK(hipStreamWaitEvent(h->s_d2h, h->ev_out[c], 0));
            BF_CHUNK(hipMemcpyAsync(y_host + c0 * H, h->d_y + c0 * H, n * H * sizeof(float), hipMemcpyDeviceToHost, h->s_d2h));
        }
#undef BF_CHUNK
        // success or not: nothing may still be copying from / into the caller's buffers when this returns
        const hipError_t e1 = hipStreamSynchronize(h->s_h2d), e2 = hipStreamSynchronize(h->stream), e3 = hipStreamSynchronize(h->s_d2h);
        if (rc != BF_OK) return rc;
        if (e != hipSuccess) return fail(h, BF_EIO, what, e);
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess)
            return fail(h, BF_EIO, "bf_process_batch: stream synchronisation", e1 != hipSuccess ? e1 : (e2 != hipSuccess ? e2 : e3));
        return BF_OK;
    }
    BF_HIP(h, hipMemcpyAsync(h->d_x, x_host, xe * sizeof(float), hipMemcpyHostToDevice, h->stream));
    rc = run_batch_device(h, h->d_x, n_frames, h->d_y, nullptr, h->stream, h->cfg.layout, (long)n_frames * h->H);
    if (rc != BF_OK) return rc;
    BF_HIP(h, hipMemcpyAsync(y_host, h->d_y, ye * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    BF_HIP(h, hipStreamSynchronize(h->stream));
    return BF_OK;
}

int bf_process_hop(bf_handle *h, const float *const *in, float *out, uint32_t nframes) {
    if (!h || !in || !out) return BF_EINVAL;
    if ((int)nframes != h->H) return fail(h, BF_EINVAL, "nframes must equal the configured hop");
    if (h->n_streams != 1) return fail(h, BF_EINVAL, "bf_process_hop drives stream 0 of a single-stream handle");
    BF_HIP(h, hipSetDevice(h->device));
    int rc = ensure_staging(h, (size_t)h->M * h->H, (size_t)h->n_out * h->H);
    if (rc != BF_OK) return rc;
    const size_t n_in = (size_t)h->M * h->H, n_outv = (size_t)h->n_out * h->H;
    if (!h->h_pin) BF_HIP(h, hipHostMalloc((void **)&h->h_pin, (n_in + n_outv) * sizeof(float), hipHostMallocDefault));
    float *packed = h->h_pin, *res = h->h_pin + n_in;
    if (h->cfg.layout == BF_PLANAR) {
        for (int m = 0; m < h->M; ++m) memcpy(packed + (size_t)m * h->H, in[m], sizeof(float) * h->H);
    } else {
        for (int m = 0; m < h->M; ++m)
            for (int n = 0; n < h->H; ++n) packed[(size_t)n * h->M + m] = in[m][n];
    }
    BF_HIP(h, hipMemcpyAsync(h->d_x, packed, n_in * sizeof(float), hipMemcpyHostToDevice, h->stream));
    rc = run_batch_device(h, h->d_x, 1, h->d_y, nullptr, h->stream, h->cfg.layout, (long)h->H);
    if (rc != BF_OK) return rc;
    BF_HIP(h, hipMemcpyAsync(res, h->d_y, n_outv * sizeof(float), hipMemcpyDeviceToHost, h->stream));  // [dir][hop]
    BF_HIP(h, hipStreamSynchronize(h->stream));
    memcpy(out, res, n_outv * sizeof(float));
    return BF_OK;
}

// ---- frame-range sharding (beamform_amd/shard.py is the same plan in Python) ------------------------------------------------
// A column range of a longer planar buffer: microphone m starts at x_dev + m * mic_stride (samples).  One input stream.
int bf_process_batch_device_strided(bf_handle *h, const float *x_dev, size_t n_frames, float *y_dev, void *hip_stream,
                                    long mic_stride) {
    if (!h || !x_dev || !y_dev || mic_stride < (long)(n_frames * (size_t)h->H)) return BF_EINVAL;
    // (several look directions write [dir][n_frames * hop] rows: a piece of a longer output buffer would need its own row stride)
    if (h->cfg.layout != BF_PLANAR || h->n_streams != 1 || h->n_dirs != 1)
        return fail(h, BF_EINVAL, "strided batches: planar layout, one input stream, one look direction");
    BF_HIP(h, hipSetDevice(h->device));
    return run_batch_device(h, x_dev, n_frames, y_dev, nullptr, (hipStream_t)hip_stream, h->cfg.layout, mic_stride);
}

int bf_shard_halo(const bf_config *cfg) {
    if (!cfg) return -1;
    switch (cfg->algo) {
        case BF_DAS:
        case BF_PHASE: return 1;                       // overlap-add neighbour only (util.h:301-302)
        case BF_MVDR:
        case BF_LCMV: return cfg->past_windows + 1;    // covariance history of frame lo-1, plus that frame (mvdr.cpp:87,100-101)
        default: return -1;                            // gss / phasempf / mcra recurse over frames, gsc over samples
    }
}

int bf_shard_plan(size_t n_frames, int world, int rank, int halo, bf_shard *out) {
    if (!out || world < 1 || rank < 0 || rank >= world || halo < 0) return BF_EINVAL;
    const long long n = (long long)n_frames, base = n / world, rem = n % world;
    out->lo = rank * base + (rank < rem ? rank : rem);
    out->hi = out->lo + base + (rank < rem ? 1 : 0);
    out->warm = (int)(halo < out->lo ? halo : out->lo);
    out->lead = (halo > 0 && out->lo - out->warm > 0) ? 1 : 0;
    return BF_OK;
}

long long bf_shard_first_feed(const bf_shard *s) { return s ? s->lo - s->warm - s->lead : 0; }
long long bf_shard_n_feed(const bf_shard *s) { return s ? s->hi - bf_shard_first_feed(s) : 0; }
long long bf_shard_n_drop(const bf_shard *s) { return s ? (long long)s->warm + s->lead : 0; }

int bf_shard_run(bf_handle *h, const float *x_feed_dev, const bf_shard *sh, float *y_feed_dev, void *hip_stream) {
    if (!h || !sh) return BF_EINVAL;
    if (h->n_streams != 1 || h->n_dirs != 1)
        return fail(h, BF_EINVAL, "frame-range sharding drives one input stream and one look direction per handle");
    const long long n_feed = bf_shard_n_feed(sh);
    if (n_feed < 0 || sh->warm < 0 || sh->lead < 0) return fail(h, BF_EINVAL, "malformed shard");
    int rc = bf_reset_async(h, hip_stream);  // the rank's node knows nothing about the frames before its slice
    if (rc != BF_OK || n_feed == 0) return rc;
    return bf_process_batch_device(h, x_feed_dev, (size_t)n_feed, y_feed_dev, nullptr, hip_stream);
}

// Per-launch timing of the dominant kernel for launches the CALLER issues (bench.py's timed loop): between begin and end every
// launch of the fused kernel is bracketed by its own event pair on the stream it is launched on.
int bf_kernel_timing_begin(bf_handle *h) {
    if (!h) return BF_EINVAL;
    if (h->timing) return fail(h, BF_EINVAL, "kernel timing already active");
    BF_HIP(h, hipSetDevice(h->device));
    int rc = timing_reserve(h, 256);  // the caller's timed loop then only records
    if (rc != BF_OK) return rc;
    h->ev_used = 0;
    h->timing = 1;
    return BF_OK;
}

int bf_trace_begin(void) {
    if (g_trace) return BF_EINVAL;
    g_trace = new std::vector<const void *>();
    return BF_OK;
}

// rocprofv3's spelling of a kernel name: the demangled name without its parameter list, "void" prefix and anonymous namespaces
static std::string trace_kernel_name(const void *fn) {
    const char *m = hipKernelNameRefByPtr(fn, nullptr);
    if (!m) return "?";
    int st = 0;
    char *d = abi::__cxa_demangle(m, nullptr, nullptr, &st);
    std::string n = (st == 0 && d) ? d : m;
    free(d);
    int depth = 0;   // cut the parameter list: the last '(' at template depth 0
    size_t cut = std::string::npos;
    for (size_t i = 0; i < n.size(); ++i) {
        if (n[i] == '<') ++depth;
        else if (n[i] == '>') --depth;
        else if (n[i] == '(' && depth == 0 && n.compare(i, 21, "(anonymous namespace)") != 0) { cut = i; break; }
    }
    if (cut != std::string::npos) n.resize(cut);
    if (n.compare(0, 5, "void ") == 0) n.erase(0, 5);
    for (size_t p; (p = n.find("(anonymous namespace)::")) != std::string::npos;) n.erase(p, 23);
    return n;
}

long bf_trace_end(char *buf, size_t cap) {
    if (!g_trace) return BF_EINVAL;
    std::string out;
    for (const void *fn : *g_trace) {
        if (!out.empty()) out += '\n';
        out += trace_kernel_name(fn);
    }
    delete g_trace;
    g_trace = nullptr;
    if (buf && cap) {
        const size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return (long)out.size();
}

int bf_kernel_timing_end(bf_handle *h, float *ms_mean, int *n_launches) {
    if (!h || !ms_mean) return BF_EINVAL;
    if (h->timing != 1) return fail(h, BF_EINVAL, "kernel timing is not active");
    return timing_collect(h, ms_mean, n_launches);
}

int bf_time_batch_device(bf_handle *h, const float *x_dev, size_t n_frames, float *y_dev, void *hip_stream, int iters,
                         float *ms_per_call, float *ms_kernel) {
    if (!h || !x_dev || !y_dev || iters < 1 || !ms_per_call) return BF_EINVAL;
    if (h->timing) return fail(h, BF_EINVAL, "bf_time_batch_device inside a bf_kernel_timing_begin session");
    BF_HIP(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)hip_stream;
    if (ms_kernel) {
        int rc0 = timing_reserve(h, (size_t)iters);
        if (rc0 != BF_OK) return rc0;
        h->ev_used = 0;
        h->timing = 2;
    }
    int rc = BF_OK;
    hipError_t e = hipEventRecord(h->ev0, s);
    for (int i = 0; i < iters && rc == BF_OK && e == hipSuccess; ++i)
        rc = run_batch_device(h, x_dev, n_frames, y_dev, nullptr, s, h->cfg.layout, (long)n_frames * h->H);
    if (e == hipSuccess) e = hipEventRecord(h->ev1, s);
    if (e == hipSuccess) e = hipEventSynchronize(h->ev1);
    float ms = 0.f, msk = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, h->ev0, h->ev1);
    int nk = 0;
    if (h->timing == 2) {
        int trc = (e == hipSuccess && rc == BF_OK) ? timing_collect(h, &msk, &nk) : BF_OK;
        h->timing = 0;
        h->ev_used = 0;
        if (rc == BF_OK) rc = trc;
    }
    if (rc != BF_OK) return rc;
    if (e != hipSuccess) return fail(h, BF_EIO, "bf_time_batch_device", e);
    *ms_per_call = ms / (float)iters;
    if (ms_kernel) *ms_kernel = nk ? msk : 0.f;
    return BF_OK;
}

// ---- checkpoint -------------------------------------------------------------
struct bf_state_header {
    uint32_t magic, algo, n_mics, n_streams, hop, das_impl;
    uint64_t payload;
};
static const uint32_t kStateMagic = 0x42465333;  // "BFS3"
// control-plane part of a checkpoint: what /theta and /theta_interference have made of the node since start-up
struct bf_state_control {
    uint32_t kp1;              // constraint columns (1 + interferers)
    uint32_t row0_written;     // reference-mic weight row: 1 after a cold start, 0 after a structural change (quirk Q3)
    uint64_t gss_pending;      // look directions whose demixing matrices restart at the next run (gss.cpp:90-93)
    uint64_t cfg_hash;         // the bf_config fields that give the state its meaning (geometry, rate, band, history length, ...)
    double theta[BF_MAX_DIRS];
    double interf[BF_MAX_INTERF];
};

// FNV-1a over the configuration a checkpoint is only valid under: restoring covariance history or demixing matrices next to
// another geometry, sample rate, band or window count would silently pair them with the wrong steering
static uint64_t state_cfg_hash(const bf_handle *h) {
    uint64_t x = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t n) {
        const unsigned char *b = (const unsigned char *)p;
        for (size_t i = 0; i < n; ++i) { x ^= b[i]; x *= 1099511628211ull; }
    };
    const bf_config &c = h->cfg;
    mix(c.mic_x, sizeof(double) * h->M);
    mix(c.mic_y, sizeof(double) * h->M);
    const double d[] = {c.sample_rate, c.freq_mag_threshold, c.freq_max, c.freq_min, c.mu, c.lambda_, c.min_phase, c.min_mag,
                        c.mcra_alphaS, c.mcra_alphaD, c.mcra_alphaD2, c.mcra_delta, c.mpf_alphaS, c.mpf_eta, c.mpf_rev_gamma,
                        c.mpf_rev_delta, c.gsc_mu0, c.gsc_mu_max};
    mix(d, sizeof(d));
    const int i[] = {c.past_windows, c.smooth_size, c.mcra_L, c.layout, c.gsc_filter_size, h->n_dirs, c.precision};  // (precision: the stored element of the covariance history)
    mix(i, sizeof(i));
    return x;
}

size_t bf_state_size(const bf_handle *h) {
    if (!h) return 0;
    size_t payload;
    if (uses_fused_das(h))
        payload = ((size_t)h->n_streams * h->M * h->H + (size_t)h->n_out * h->H) * sizeof(float);
    else
        payload = h->pipe ? h->pipe->state_bytes() : 0;
    return sizeof(bf_state_header) + sizeof(bf_state_control) + payload;
}

int bf_get_state(bf_handle *h, void *blob, size_t size) {
    if (!h || !blob) return BF_EINVAL;
    {
        std::lock_guard<std::mutex> lk(h->mu);  // the payload size follows the interferer count, which /theta_interference changes
        if (size < bf_state_size(h)) return BF_EINVAL;
    }
    BF_HIP(h, hipSetDevice(h->device));
    BF_HIP(h, hipDeviceSynchronize());
    bf_state_header hd = {kStateMagic, (uint32_t)h->cfg.algo, (uint32_t)h->M, (uint32_t)h->n_streams | ((uint32_t)h->n_dirs << 20),
                          (uint32_t)h->H, (uint32_t)h->cfg.das_impl, (uint64_t)(bf_state_size(h) - sizeof(bf_state_header))};
    memcpy(blob, &hd, sizeof(hd));
    bf_state_control ct;
    memset(&ct, 0, sizeof(ct));
    {
        std::lock_guard<std::mutex> lk(h->mu);
        ct.kp1 = (uint32_t)h->S;
        ct.row0_written = h->row0_written ? 1u : 0u;
        ct.cfg_hash = state_cfg_hash(h);
        ct.gss_pending = h->pipe ? h->pipe->pending_resets() : 0ull;
        for (int d = 0; d < h->n_dirs; ++d) ct.theta[d] = h->angle[d];
        for (size_t k = 0; k < h->interf.size(); ++k) ct.interf[k] = h->interf[k];
    }
    memcpy((char *)blob + sizeof(hd), &ct, sizeof(ct));
    char *p = (char *)blob + sizeof(hd) + sizeof(ct);
    if (uses_fused_das(h)) {
        const size_t hb = (size_t)h->n_streams * h->M * h->H * sizeof(float), tb = (size_t)h->n_out * h->H * sizeof(float);
        BF_HIP(h, hipMemcpy(p, h->d_hist[h->tail_cur], hb, hipMemcpyDeviceToHost));
        BF_HIP(h, hipMemcpy(p + hb, h->d_tail[h->tail_cur], tb, hipMemcpyDeviceToHost));
        return BF_OK;
    }
    int rc = h->pipe->get_state(p);
    return rc == BF_OK ? BF_OK : fail(h, rc, h->pipe->error().c_str());
}

int bf_set_state(bf_handle *h, const void *blob, size_t size) {
    if (!h || !blob) return BF_EINVAL;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        if (size < bf_state_size(h)) return BF_EINVAL;
    }
    bf_state_header hd;
    memcpy(&hd, blob, sizeof(hd));
    if (hd.magic != kStateMagic || hd.algo != (uint32_t)h->cfg.algo || hd.n_mics != (uint32_t)h->M ||
        hd.n_streams != ((uint32_t)h->n_streams | ((uint32_t)h->n_dirs << 20)) || hd.hop != (uint32_t)h->H || hd.das_impl != (uint32_t)h->cfg.das_impl)
        return fail(h, BF_EINVAL, "state blob does not match this handle");
    bf_state_control ct;
    memcpy(&ct, (const char *)blob + sizeof(hd), sizeof(ct));
    BF_HIP(h, hipSetDevice(h->device));
    BF_HIP(h, hipDeviceSynchronize());
    {
        // the per-bin state (covariance history, demixing matrices) only means something together with the steering it
        // was built under: restore the angles, the interferer list and the pending demixing resets with it
        std::lock_guard<std::mutex> lk(h->mu);
        if ((int)ct.kp1 != h->S) return fail(h, BF_EINVAL, "state blob was taken with a different number of interferers");
        if (ct.cfg_hash != state_cfg_hash(h))
            return fail(h, BF_EINVAL, "state blob was taken under another configuration (geometry, rate, band, window count or node parameters)");
        for (int d = 0; d < h->n_dirs; ++d)
            if (!std::isfinite(ct.theta[d])) return fail(h, BF_EINVAL, "state blob holds a non-finite look angle");
        for (size_t k = 0; k < h->interf.size(); ++k)
            if (!std::isfinite(ct.interf[k])) return fail(h, BF_EINVAL, "state blob holds a non-finite interferer angle");
        h->row0_written = ct.row0_written != 0;
        for (int d = 0; d < h->n_dirs; ++d) h->angle[d] = ct.theta[d];
        for (size_t k = 0; k < h->interf.size(); ++k) h->interf[k] = ct.interf[k];
        for (auto &st : h->steer)
            for (int c = 0; c < st.n_cols; ++c)
                for (int j = 0; j < st.n_fft; ++j) st.at(j, 0, c) = ct.row0_written ? cplxd(1.0, 0.0) : cplxd(0.0, 0.0);
        rebuild_steering(h, false);
        if (h->pipe) h->pipe->set_pending_resets(ct.gss_pending);
    }
    const char *p = (const char *)blob + sizeof(hd) + sizeof(ct);
    if (uses_fused_das(h)) {
        const size_t hb = (size_t)h->n_streams * h->M * h->H * sizeof(float), tb = (size_t)h->n_out * h->H * sizeof(float);
        BF_HIP(h, hipMemcpy(h->d_hist[h->tail_cur], p, hb, hipMemcpyHostToDevice));
        BF_HIP(h, hipMemcpy(h->d_tail[h->tail_cur], p + hb, tb, hipMemcpyHostToDevice));
        return BF_OK;
    }
    int rc = h->pipe->set_state(p);
    return rc == BF_OK ? BF_OK : fail(h, rc, h->pipe->error().c_str());
}

}  // extern "C"
