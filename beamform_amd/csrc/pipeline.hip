// pipeline.hip -- fp64 bin pipeline for gfx950: every node except fused DAS.
//
//   stft_kernel   : overlap_and_add_prepare_input + fftw_execute(x_forward) for all mics
//                   (util.h:217-242, das.cpp:51-57); two real mics per complex FFT-1024,
//                   packed spectra Z_p = FFT(a + i b) go to an HBM workspace
//   *_bins_kernel : the node's per-bin loop of apply_weights(); unpacks
//                   X_a[k] = (Z[k] + conj Z[N-k])/2, X_b[k] = (Z[k] - conj Z[N-k])/(2i) on load
//   istft_w64_kernel / istft32_kernel : fftw_execute(y_inverse) + overlap_and_add_prepare_output + do_overlap's
//                   overlap-add (das.cpp:66, util.h:244-253,301-302), two frames per complex
//                   IFFT (Hermitian half-spectra in, real frames out as re / im)
//
// fp64 throughout: the covariance solves of mvdr/lcmv amplify input error by the
// condition number (1e3..1e4 on coherent scenes), and the phase / MCRA threshold
// decisions flip on fp32 noise; 1e-5 on the complex spectrum needs double
// (DESIGN.md "Precision").  The reference itself is double everywhere.
#include "pipeline.hpp"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fft1024.hpp"
#include "pipeline_kernels.hpp"

namespace bf {

// ============================================================================
//                                   host side
// ============================================================================
namespace {

#define PIPE_HIP(call)                                                        \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) {                                               \
            err_ = std::string(#call) + ": " + hipGetErrorString(e_);         \
            return BF_EIO;                                                    \
        }                                                                     \
    } while (0)

class BinPipelineImpl : public BinPipeline {
    static constexpr int kMaxCols = BF_MAX_INTERF + 1;  // look direction + up to 15 interferers (per-bin kernels: KM <= 16)
   public:
    BinPipelineImpl(const bf_config &c, int n_cus) : cfg_(c), n_cus_(n_cus) {
        M_ = c.n_mics;
        H_ = c.hop;
        N_ = 2 * c.hop;
        NQ_ = problems_per_frame(N_);
        YS_ = yh_stride(N_);
        ks_ = kernel_set(N_);
        MF_ = (c.algo == BF_MCRA) ? 1 : M_;  // the mcra node only transforms channel 0 (mcra.cpp:72-73)
        NP_ = (MF_ + 1) / 2;
        S_ = c.n_streams;                                   // input streams
        D_ = c.n_dirs > 1 ? c.n_dirs : 1;                   // look directions per input stream
        if (c.algo == BF_GSC) D_ = M_;                      // gsc: one aligned output per microphone (do_overlap_bymic)
        So_ = S_ * D_;                                      // output streams
        const bool multi = (c.algo == BF_LCMV || c.algo == BF_GSS);
        KP1_ = multi ? c.n_interf + 1 : 1;
        Phist_ = (c.algo == BF_MVDR || c.algo == BF_LCMV) ? c.past_windows : 0;
        // BF_PRECISION_MIXED: mvdr / lcmv park their spectra as 12-byte z48 elements (pipeline_kernels.hpp); the default keeps complex doubles
        z48_ = (c.algo == BF_MVDR || c.algo == BF_LCMV) && c.precision == BF_PRECISION_MIXED;
        zsz_ = z48_ ? sizeof(z48) : sizeof(f64x2);
    }
    ~BinPipelineImpl() override { free_all(); }

    int init() override {
        if (ks_ == nullptr) {
            err_ = "hop (JACK period) must be a power of two from 64 to 4096 frames";
            return BF_ENOSYS;
        }
        if (M_ > 32 && (cfg_.algo == BF_MVDR || cfg_.algo == BF_LCMV || cfg_.algo == BF_GSS)) {
            err_ = "mvdr/lcmv/gss kernels are built for up to 32 microphones";
            return BF_ENOSYS;
        }
        if (KP1_ > kMaxCols && (cfg_.algo == BF_LCMV || cfg_.algo == BF_GSS)) {
            err_ = "lcmv/gss kernels are built for up to 15 interferers";
            return BF_ENOSYS;
        }
        if ((cfg_.algo == BF_MVDR || cfg_.algo == BF_LCMV) && (Phist_ < 1 || Phist_ > 64)) {
            err_ = "past_windows must be in 1..64";
            return BF_EINVAL;
        }
        if (cfg_.algo == BF_GSC && (M_ > 16 || cfg_.gsc_filter_size < 1 || cfg_.gsc_filter_size > 256)) {
            err_ = "gsc is built for up to 16 microphones and filter_size 1..256";
            return M_ > 16 ? BF_ENOSYS : BF_EINVAL;
        }
        if (cfg_.algo == BF_PHASEMPF && (cfg_.smooth_size < 1 || cfg_.smooth_size > 64)) {
            err_ = "smooth_size must be in 1..64";
            return BF_EINVAL;
        }
        // N = 1024: inter-pass twiddles of the 32 x 32 factorisation; other sizes: exp(-2 pi i m / N), m < N/2 (Stockham passes)
        std::vector<f64x2> tw = twiddle_table_32x32<f64x2>();
        if (N_ != 1024) tw = stockham_twiddles<f64x2>(N_);
        PIPE_HIP(hipMalloc((void **)&d_tw_, tw.size() * sizeof(f64x2)));
        PIPE_HIP(hipMemcpy(d_tw_, tw.data(), tw.size() * sizeof(f64x2), hipMemcpyHostToDevice));
        std::vector<f32x2> tw32 = twiddle_table_32x32<f32x2>();
        PIPE_HIP(hipMalloc((void **)&d_tw32_, tw32.size() * sizeof(f32x2)));
        PIPE_HIP(hipMemcpy(d_tw32_, tw32.data(), tw32.size() * sizeof(f32x2), hipMemcpyHostToDevice));
        std::vector<double> h = sqrt_hann(N_);
        PIPE_HIP(hipMalloc((void **)&d_win_, h.size() * sizeof(double)));
        PIPE_HIP(hipMemcpy(d_win_, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
        freqs_ = frequency_vector(N_, cfg_.sample_rate);
        PIPE_HIP(hipMalloc((void **)&d_freq_, N_ * sizeof(double)));
        PIPE_HIP(hipMemcpy(d_freq_, freqs_.data(), N_ * sizeof(double), hipMemcpyHostToDevice));
        for (int i = 0; i < 2; ++i)
            PIPE_HIP(hipMalloc((void **)&d_steer_[i], steer_bytes()));
        if (das_one_launch_shape()) {
            for (int i = 0; i < 2; ++i) PIPE_HIP(hipMalloc((void **)&d_dasg_w64_[i], (size_t)4 * 1024 * sizeof(f64x2)));
            for (int i = 0; i < 2; ++i) PIPE_HIP(hipMalloc((void **)&d_dasg_mic_[i], (size_t)8 * kDasMicGainRows * kDasMicGainRow * sizeof(f64x2)));
            PIPE_HIP(hipMalloc(&d_das_sched_, das_f64_sched_ws_bytes()));  // das_f64_pair_kernel's work queue (chunk table + counter)
        }
        if (N_ == 1024) {
            const std::vector<f64x2> tw64 = twiddle_table_w64_rot();
            PIPE_HIP(hipMalloc((void **)&d_tw_w64_, tw64.size() * sizeof(f64x2)));
            PIPE_HIP(hipMemcpy(d_tw_w64_, tw64.data(), tw64.size() * sizeof(f64x2), hipMemcpyHostToDevice));
        }
        for (int i = 0; i < 2; ++i) PIPE_HIP(hipMalloc((void **)&d_hist2_[i], (size_t)S_ * M_ * H_ * sizeof(float)));
        PIPE_HIP(hipMalloc((void **)&d_tail_[0], (size_t)So_ * H_ * sizeof(float)));
        PIPE_HIP(hipMalloc((void **)&d_tail_[1], (size_t)So_ * H_ * sizeof(float)));
        if (Phist_ > 0) PIPE_HIP(hipMalloc((void **)&d_zhist_, zhist_bytes()));
        if (cfg_.algo == BF_GSS) {
            PIPE_HIP(hipMalloc((void **)&d_gssW_, gss_bytes()));
            PIPE_HIP(hipMemset(d_gssW_, 0, gss_bytes()));  // defined content until the first run applies W = C^H
        }
        if (cfg_.algo == BF_PHASEMPF || cfg_.algo == BF_MCRA) PIPE_HIP(hipMalloc((void **)&d_mpf_, mpf_bytes()));
        if (cfg_.algo == BF_PHASEMPF) PIPE_HIP(hipMalloc((void **)&d_smooth_, smooth_bytes()));
        if (cfg_.algo == BF_GSC) PIPE_HIP(hipMalloc((void **)&d_nlms_, nlms_bytes()));
        return BF_OK;
    }

    int reset(hipStream_t st) override {
        PIPE_HIP(hipMemsetAsync(d_hist2_[0], 0, (size_t)S_ * M_ * H_ * sizeof(float), st));
        hist_cur_ = 0;
        PIPE_HIP(hipMemsetAsync(d_tail_[0], 0, (size_t)So_ * H_ * sizeof(float), st));
        PIPE_HIP(hipMemsetAsync(d_tail_[1], 0, (size_t)So_ * H_ * sizeof(float), st));
        tail_cur_ = 0;
        if (d_zhist_) PIPE_HIP(hipMemsetAsync(d_zhist_, 0, zhist_bytes(), st));  // past_ffts setZero (mvdr.cpp:228-232); zero bits = z48 zero
        if (d_mpf_) PIPE_HIP(hipMemsetAsync(d_mpf_, 0, mpf_bytes(), st));          // phasempf.cpp:535-545, current_L=0/first_L
        if (d_smooth_) PIPE_HIP(hipMemsetAsync(d_smooth_, 0, smooth_bytes(), st)); // calloc past_samples (phasempf.cpp:510)
        if (d_nlms_) PIPE_HIP(hipMemsetAsync(d_nlms_, 0, nlms_bytes(), st));       // calloc block_matrix/filter/last_outputs (gsc.cpp:278-285)
        gss_reset_mask_ = ~0ull;  // sep_matrix = weights^H (gss.cpp:90-93), done on the stream at next run
        return BF_OK;
    }

    int upload_steering(const std::vector<SteeringSet> &dirs, hipStream_t stream) override {
        // device layout [dir][col][mic][bin] so that lanes (bins) read consecutive addresses
        const int nc = dirs[0].n_cols, nd = (int)dirs.size();  // nd = look directions (gsc: 1, although D_ = M outputs)
        if ((size_t)nd * nc * M_ * N_ * sizeof(f64x2) > steer_bytes()) {
            err_ = "steering table larger than the device buffer";
            return BF_EINVAL;
        }
        std::vector<f64x2> t((size_t)nd * N_ * M_ * nc);
        for (int d = 0; d < nd; ++d)
            for (int c = 0; c < nc; ++c)
                for (int m = 0; m < M_; ++m)
                    for (int j = 0; j < N_; ++j) {
                        const cplxd w = dirs[d].at(j, m, c);
                        t[(((size_t)d * nc + c) * M_ + m) * N_ + j] = f64x2{w.real(), w.imag()};
                    }
        const int nxt = steer_cur_ ^ 1;
        std::vector<f64x2> dg, dg64, dgm;  // das fp64 in one launch: the pair gains of the (single) look direction, same double buffering
        if (das_one_launch_shape()) {
            dg = das_pair_gains_t<f64x2>(dirs[0], 4);
            dg64 = das_pair_gains_w64_f64(dg, 4);
            PIPE_HIP(hipMemcpyAsync(d_dasg_w64_[nxt], dg64.data(), dg64.size() * sizeof(f64x2), hipMemcpyHostToDevice, stream));
            dgm = das_mic_gains_w64_f64(dirs[0], 8);
            PIPE_HIP(hipMemcpyAsync(d_dasg_mic_[nxt], dgm.data(), dgm.size() * sizeof(f64x2), hipMemcpyHostToDevice, stream));
            bool unit = true;  // das.cpp:33-38 writes weights(0, j) = 1 once; only then may the pair kernel skip microphone 0's transform
            for (int j = 0; j < N_ && unit; ++j) unit = dirs[0].at(j, 0, 0) == cplxd(1.0, 0.0);
            das_mic0_unit_[nxt] = unit;
            // Microphones 1 .. M - 1 whose weight rows are bitwise identical (same delay: the reference drops z, util.h:82-92, so e.g. aira16's
            // microphones 1 and 7 -- beamform_config.yaml:21,27 -- coincide for every look direction) share a forward transform in
            // das_f64_pair_kernel: the first such pair goes into slot 0, the other microphones follow in ascending order.
            DasSlots sl;
            int pa = -1, pb = -1;
            for (int m1 = 1; m1 < M_ && pa < 0; ++m1)
                for (int m2 = m1 + 1; m2 < M_ && pa < 0; ++m2) {
                    bool same = true;
                    for (int j = 0; j < N_ && same; ++j) same = dirs[0].at(j, m1, 0) == dirs[0].at(j, m2, 0);
                    if (same) { pa = m1; pb = m2; }
                }
            sl.n_tr = 0;
            if (pa >= 0) sl.slot_mic[sl.n_tr++] = pa;
            for (int m = 1; m < M_; ++m)
                if (m != pa && m != pb) sl.slot_mic[sl.n_tr++] = m;
            sl.extra_mic = pb;
            das_slots_[nxt] = sl;
        }
        PIPE_HIP(hipMemcpyAsync(d_steer_[nxt], t.data(), t.size() * sizeof(f64x2), hipMemcpyHostToDevice, stream));
        PIPE_HIP(hipStreamSynchronize(stream));  // `t` is pageable and about to go out of scope
        steer_cur_ = nxt;
        steer_dir_stride_ = (long)nc * M_ * N_;
        return BF_OK;
    }

    void on_theta_changed(int dir) override { gss_reset_mask_ |= dir < 0 ? ~0ull : (1ull << dir); }
    void set_columns(int kp1) override {
        KP1_ = kp1;
        gss_reset_mask_ = ~0ull;
    }

    RunSnapshot snapshot_for_run() override {
        RunSnapshot sn;
        sn.kp1 = KP1_;
        sn.gss_reset_mask = gss_reset_mask_;
        sn.steer = d_steer_[steer_cur_];
        sn.steer_dir_stride = steer_dir_stride_;
        sn.das_gains_w64 = d_dasg_w64_[steer_cur_];
        sn.das_gains_mic = d_dasg_mic_[steer_cur_];
        sn.das_mic0_unit = das_mic0_unit_[steer_cur_];
        sn.das_slots = das_slots_[steer_cur_];
        gss_reset_mask_ = 0;
        return sn;
    }
    int columns() const override { return KP1_; }
    unsigned long long pending_resets() const override { return gss_reset_mask_; }
    void set_pending_resets(unsigned long long mask) override { gss_reset_mask_ = mask; }

    int run(const float *x, long F, float *y, f64x2 *spectrum, hipStream_t stream, int layout, long mic_stride,
            const RunSnapshot &snap) override;
    int run_one(const float *x, long F, float *y, f64x2 *spectrum, hipStream_t stream, int layout, long mic_stride,
                const RunSnapshot &snap);

    size_t state_bytes() const override {
        return (size_t)S_ * M_ * H_ * 4 + (size_t)So_ * H_ * 4 + zhist_bytes() + gss_bytes() + mpf_bytes() + smooth_bytes() +
               nlms_bytes();
    }
    int get_state(void *host) override { return copy_state((char *)host, true); }
    int set_state(const void *host) override { return copy_state((char *)host, false); }

   private:
    // das through this pipeline on the tuned shape: eligible for the one-launch kernels of das_f64_w64.hip (run_one decides per batch)
    bool das_one_launch_shape() const { return cfg_.algo == BF_DAS && N_ == 1024 && M_ <= 8 && D_ == 1; }
    size_t steer_bytes() const { return (size_t)D_ * N_ * M_ * kMaxCols * sizeof(f64x2); }
    size_t zhist_bytes() const { return Phist_ ? (size_t)S_ * Phist_ * NP_ * N_ * zsz_ : 0; }
    // recursive per-beam state is sized by OUTPUT streams (input streams x look directions)
    size_t gss_bytes() const { return cfg_.algo == BF_GSS ? (size_t)So_ * N_ * kMaxCols * M_ * sizeof(f64x2) : 0; }
    size_t mpf_bytes() const {
        return (cfg_.algo == BF_PHASEMPF || cfg_.algo == BF_MCRA) ? (size_t)So_ * (kMpfVecs * N_ + 8) * sizeof(double) : 0;
    }
    size_t smooth_bytes() const { return cfg_.algo == BF_PHASEMPF ? (size_t)So_ * 64 * sizeof(double) : 0; }
    size_t nlms_bytes() const {
        return cfg_.algo == BF_GSC ? (size_t)S_ * (2 * (M_ - 1) + 1) * cfg_.gsc_filter_size * sizeof(float) : 0;
    }

    int copy_state(char *p, bool to_host) {
        PIPE_HIP(hipDeviceSynchronize());
        struct Seg { void *d; size_t n; } segs[] = {
            {d_hist2_[hist_cur_], (size_t)S_ * M_ * H_ * 4}, {d_tail_[tail_cur_], (size_t)So_ * H_ * 4}, {d_zhist_, zhist_bytes()},
            {d_gssW_, gss_bytes()}, {d_mpf_, mpf_bytes()}, {d_smooth_, smooth_bytes()}, {d_nlms_, nlms_bytes()}};
        for (auto &s : segs) {
            if (!s.n) continue;
            if (to_host)
                PIPE_HIP(hipMemcpy(p, s.d, s.n, hipMemcpyDeviceToHost));
            else
                PIPE_HIP(hipMemcpy(s.d, p, s.n, hipMemcpyHostToDevice));
            p += s.n;
        }
        return BF_OK;  // the pending-reset mask travels in the blob's control-plane section (capi.cpp)
    }

    int ensure(void **ptr, size_t *cap, size_t need) {
        if (need <= *cap) return BF_OK;
        if (*ptr) (void)hipFree(*ptr);
        *ptr = nullptr;
        *cap = 0;
        PIPE_HIP(hipMalloc(ptr, need));
        *cap = need;
        return BF_OK;
    }

    void free_all() {
        void *ptrs[] = {d_dasg_w64_[0], d_dasg_w64_[1], d_dasg_mic_[0], d_dasg_mic_[1], d_das_sched_, d_tw_w64_, d_tw32_, d_tw_, d_win_, d_freq_, d_steer_[0], d_steer_[1], d_hist2_[0], d_hist2_[1], d_tail_[0], d_tail_[1], d_zhist_,
                        d_gssW_, d_mpf_, d_smooth_, d_nlms_, d_Z_, d_Yh_, d_yraw_, d_frames_, d_planar_};
        for (void *p : ptrs)
            if (p) (void)hipFree(p);
    }

    bf_config cfg_;
    int n_cus_, M_, MF_, NP_, S_, D_, So_, KP1_, Phist_;
    bool z48_ = false;
    size_t zsz_ = sizeof(f64x2);  // bytes per packed-spectrum element
    int H_ = 512, N_ = 1024, NQ_ = 514, YS_ = 516;  // hop, FFT size, problems per frame, row stride of Yh
    const KernelSet *ks_ = nullptr;                 // launchers compiled for N_
    long steer_dir_stride_ = 0;
    std::vector<double> freqs_;
    f64x2 *d_tw_ = nullptr;
    f32x2 *d_tw32_ = nullptr;
    double *d_win_ = nullptr, *d_freq_ = nullptr;
    f64x2 *d_steer_[2] = {nullptr, nullptr};
    f64x2 *d_dasg_w64_[2] = {nullptr, nullptr};  // das_pair_gains_w64_f64 of the same
    f64x2 *d_tw_w64_ = nullptr;                  // twiddle_table_w64_rot
    void *d_das_sched_ = nullptr;                // das_f64_pair_kernel: chunk table + counter (das_f64_sched_ws_bytes())
    f64x2 *d_dasg_mic_[2] = {nullptr, nullptr};  // das_mic_gains_w64_f64 (frame-pair kernel)
    bool das_mic0_unit_[2] = {false, false};     // ... and whether microphone 0's weight row in that table is identically 1
    DasSlots das_slots_[2];                      // ... and which microphones get a forward transform in which order (identical rows merged)
    int steer_cur_ = 0;
    float *d_hist2_[2] = {nullptr, nullptr};  // ring hop in front of the next batch; two buffers: das_f64_pair_kernel writes the carry itself
    int hist_cur_ = 0;
    float *d_tail_[2] = {nullptr, nullptr};
    int tail_cur_ = 0;
    f64x2 *d_zhist_ = nullptr;   // [stream][Phist][NP][1024] elements of zsz_ bytes: packed spectra of the previous Phist frames
    f64x2 *d_gssW_ = nullptr;    // [stream][bin][KP1][M]
    double *d_mpf_ = nullptr;    // [stream][kMpfVecs*1024 + 8]
    double *d_smooth_ = nullptr; // [stream][64]
    float *d_nlms_ = nullptr;    // gsc: [stream][(2(M-1)+1) * filter_size]
    unsigned long long gss_reset_mask_ = ~0ull;  // look directions whose demixing matrices restart at the next run
    // workspaces (grown on demand)
    f64x2 *d_Z_ = nullptr;   size_t Z_cap_ = 0;   // [stream][Phist+F][NP][1024]
    f64x2 *d_Yh_ = nullptr;  size_t Yh_cap_ = 0;  // [stream][F][kYhStride]
    float *d_yraw_ = nullptr; size_t yraw_cap_ = 0;
    float *d_planar_ = nullptr; size_t planar_cap_ = 0;  // das in double on [sample][mic] input: the batch (+ carried hop) transposed for das_f64_pair_kernel
    float *d_frames_ = nullptr; size_t frames_cap_ = 0;  // N != 1024: windowed frames between the generic ISTFT and its overlap-add
};

int BinPipelineImpl::run(const float *x, long F, float *y, f64x2 *spectrum, hipStream_t stream, int layout,
                         long mic_stride, const RunSnapshot &snap) {
    // one pass over the whole batch: cutting it into Infinity-Cache-sized frame tiles was measured (3.9-12 ms for mvdr instead of
    // 3.0: per-tile launches underfill the chip and the per-bin kernels lose their parallelism over time) -- DESIGN.md 3.2
    return run_one(x, F, y, spectrum, stream, layout, mic_stride, snap);
}

int BinPipelineImpl::run_one(const float *x, long F, float *y, f64x2 *spectrum, hipStream_t stream, int layout,
                             long mic_stride, const RunSnapshot &snap) {
    // das at the reference's precision on the tuned shape, no spectrum dump: ONE launch, spectra never leave the CU (das_f64_w64.hip:
    // planar input das_f64_pair_kernel, [sample][mic] input das_f64_w64_kernel<1>); BF_FUSED_BINS=0 keeps the chain below (cross-checks)
    static const int fuse_env0 = getenv("BF_FUSED_BINS") ? atoi(getenv("BF_FUSED_BINS")) : 1;
    if (das_one_launch_shape() && fuse_env0 == 1 && spectrum == nullptr && snap.das_gains_w64 != nullptr) {
        DasF64Args da;
        da.x = x; da.hist = d_hist2_[hist_cur_]; da.hist_out = d_hist2_[hist_cur_ ^ 1]; da.y = y; da.tail_in = d_tail_[tail_cur_]; da.tail_out = d_tail_[tail_cur_ ^ 1];
        da.win = d_win_; da.n_frames = F; da.mic_stride = mic_stride;
        da.stream_stride_x = (long)M_ * F * H_; da.n_streams = S_; da.n_mics = M_; da.run_len = 1;
        da.layout = layout;
        // [sample][mic] input: transposed into a planar scratch (batch + carried hop) in front of the frame-pair kernel, which then runs with its
        // microphone-0 and identical-row savings (das_f64_w64.hip interleaved_to_planar_kernel); the carried hop stays in the handle's layout
        static const int il_ring_env = getenv("BF_DAS_IL_RING") ? atoi(getenv("BF_DAS_IL_RING")) : 1;
        const size_t ring_bytes = il_ring_env ? das_f64_ring_bytes(M_, n_cus_) : 0;
        if (layout == BF_INTERLEAVED && snap.das_mic0_unit && M_ >= 2 && snap.das_slots.n_tr >= 1 && d_das_sched_ != nullptr && ring_bytes > 0) {
            // 2, 4 or 8 microphones: the frame-pair kernel transposes hop by hop into its blocks' rings (das_f64_ring_kernel): 160 MB of
            // scratch instead of a planar copy of the batch, and the transposition's memory traffic runs under the other wavefronts' transforms
            const int rc0 = ensure((void **)&d_planar_, &planar_cap_, ring_bytes);
            if (rc0 != BF_OK) return rc0;
            da.ring = d_planar_; da.ring_bytes = ring_bytes; da.hist_out = nullptr;
            da.stream_stride_x = (long)M_ * F * H_;
        } else if (layout == BF_INTERLEAVED && snap.das_mic0_unit && M_ >= 2 && snap.das_slots.n_tr >= 1 && d_das_sched_ != nullptr) {
            const size_t nb = (size_t)S_ * M_ * F * H_, nh = (size_t)S_ * M_ * H_;
            const int rc0 = ensure((void **)&d_planar_, &planar_cap_, (nb + nh) * sizeof(float));
            if (rc0 != BF_OK) return rc0;
            hipError_t te = launch_interleaved_to_planar(x, d_planar_, F * H_, M_, S_, stream);
            if (te == hipSuccess) te = launch_interleaved_to_planar(d_hist2_[hist_cur_], d_planar_ + nb, H_, M_, S_, stream);
            if (te == hipSuccess) {
                da.x = d_planar_; da.hist = d_planar_ + nb; da.hist_out = nullptr; da.mic_stride = F * H_; da.layout = BF_PLANAR;
            } else if (te != hipErrorNotSupported) {
                PIPE_HIP(te);
            } else {
                (void)hipGetLastError();
            }
        }
        da.gains = snap.das_gains_w64; da.gains_mic = snap.das_gains_mic; da.tw = d_tw_w64_;
        da.mic0_unit = snap.das_mic0_unit ? 1 : 0;
        da.n_tr = snap.das_slots.n_tr; da.extra_mic = snap.das_slots.extra_mic;
        for (int k = 0; k < 8; ++k) da.slot_mic[k] = snap.das_slots.slot_mic[k];
        da.sched_ws = d_das_sched_; da.sched_ws_bytes = d_das_sched_ ? das_f64_sched_ws_bytes() : 0;
        hipError_t de = prepare_das_f64_w64(da, n_cus_, stream);
        if (de == hipSuccess) {
            if (kev0) PIPE_HIP(hipEventRecord(kev0, stream));
            de = launch_das_f64_w64(da, n_cus_, stream);
            if (kev1 && de == hipSuccess) {
                PIPE_HIP(hipEventRecord(kev1, stream));
                kev_recorded = true;
            }
        }
        if (de == hipSuccess) {  // ring-buffer carry (util.h:305-308)
            if (das_f64_writes_hist(da)) {
                hist_cur_ ^= 1;  // das_f64_pair_kernel stored the last hop into the other buffer
            } else if (layout == BF_PLANAR)
                PIPE_HIP(hipMemcpy2DAsync(d_hist2_[hist_cur_], H_ * sizeof(float), x + (F - 1) * H_, (size_t)mic_stride * sizeof(float),
                                          H_ * sizeof(float), (size_t)S_ * M_, hipMemcpyDeviceToDevice, stream));
            else
                PIPE_HIP(hipMemcpy2DAsync(d_hist2_[hist_cur_], (size_t)H_ * M_ * sizeof(float), x + (F - 1) * (long)H_ * M_,
                                          (size_t)F * H_ * M_ * sizeof(float), (size_t)H_ * M_ * sizeof(float), (size_t)S_,
                                          hipMemcpyDeviceToDevice, stream));
            tail_cur_ ^= 1;
            return BF_OK;
        }
        if (de != hipErrorNotSupported) PIPE_HIP(de);
        (void)hipGetLastError();
    }
    const long FT = Phist_ + F;  // frames in the Z workspace per stream
    // nodes without a frame history: STFT and per-bin stage in one launch, spectra never leave the CU (launch_stft_bins_fused;
    // BF_FUSED_BINS=0 selects the two-kernel chain) -- then the Z workspace (64 KB per frame at 8 microphones) is not needed at all
    const bool try_fused = fuse_env0 != 0 && Phist_ == 0 && N_ <= 2048 && M_ <= 8 && MF_ == M_ && D_ == 1 &&  // (N = 128 / 256 / 512: stft_bins_small_kernel, 2048: stft_bins_split_kernel)
                           (cfg_.algo == BF_DAS || cfg_.algo == BF_PHASE || cfg_.algo == BF_PHASEMPF);
    int rc = BF_OK;
    if (!try_fused)
        // (+ 512 frames of slack behind the last stream: mvdr_fast_kernel's lanes of a short last tile prefetch up to one tile
        // length past the end of their stream and never use what they fetched)
        rc = ensure((void **)&d_Z_, &Z_cap_, ((size_t)S_ * FT + ((cfg_.algo == BF_MVDR || cfg_.algo == BF_LCMV) ? 512 : 0)) * NP_ * N_ * zsz_);
    else  // the fused kernel parks the unpacked spectra of two bins per frame here (stream x frame x 2 x 8 microphones)
        rc = ensure((void **)&d_Z_, &Z_cap_, (size_t)S_ * F * 2 * 8 * sizeof(f64x2));
    if (rc != BF_OK) return rc;
    // phasempf keeps |out_int|^2 (one double per problem) behind the spectrum rows
    rc = ensure((void **)&d_Yh_, &Yh_cap_,
                (size_t)So_ * F * YS_ * (sizeof(f64x2) + (cfg_.algo == BF_PHASEMPF ? sizeof(double) : 0)));
    if (rc != BF_OK) return rc;
    if (cfg_.algo == BF_PHASEMPF || cfg_.algo == BF_GSC) {
        rc = ensure((void **)&d_yraw_, &yraw_cap_, (size_t)So_ * F * H_ * sizeof(float));
        if (rc != BF_OK) return rc;
    }
    if (N_ != 1024) {
        rc = ensure((void **)&d_frames_, &frames_cap_, (size_t)So_ * F * N_ * sizeof(float));
        if (rc != BF_OK) return rc;
    }
    const size_t frame_elems = (size_t)NP_ * N_;

    // covariance history in front of the new frames
    if (Phist_ > 0)
        PIPE_HIP(hipMemcpy2DAsync(d_Z_, (size_t)FT * frame_elems * zsz_, d_zhist_, (size_t)Phist_ * frame_elems * zsz_,
                                  (size_t)Phist_ * frame_elems * zsz_, (size_t)S_, hipMemcpyDeviceToDevice, stream));

    StftArgs sa;
    sa.x = x; sa.hist = d_hist2_[hist_cur_]; sa.Z = d_Z_; sa.tw = d_tw_; sa.win = d_win_;
    sa.n_frames = F; sa.frames_ws = FT; sa.frame_off = Phist_; sa.mic_stride = mic_stride;
    sa.stream_stride_x = (long)M_ * F * H_; sa.n_streams = S_; sa.n_mics = M_; sa.n_fft_mics = MF_; sa.layout = layout;
    sa.skip_lo = N_; sa.skip_hi = 0;  // store everything ...
    sa.z48 = z48_ ? 1 : 0; sa.run_len = 1; sa.tw_w64 = d_tw_w64_;
    sa.halve = (cfg_.algo == BF_MVDR || cfg_.algo == BF_LCMV) ? 1 : 0;
    if (cfg_.algo == BF_MVDR || cfg_.algo == BF_LCMV || cfg_.algo == BF_GSS) {
        // ... except, for the band-limited nodes, the bins between the highest in-band bin k and its mirror N-k
        // (quirk Q1 makes bins 511..513 irregular: only skip when the band ends below them)
        int kmax = 0;
        for (int k = 0; k <= N_ / 2 + 1; ++k) {
            const double f = std::fabs(freqs_[k]);
            if (f >= cfg_.freq_min && f <= cfg_.freq_max) kmax = k;
        }
        if (kmax < N_ / 2 - 2) { sa.skip_lo = kmax; sa.skip_hi = N_ - kmax; }
    }
    if (cfg_.algo == BF_GSC && spectrum) {  // time-domain node: there is no single y_fft; the dump reads as zeros
        PIPE_HIP(hipMemsetAsync(spectrum, 0, (size_t)S_ * F * N_ * sizeof(f64x2), stream));
        spectrum = nullptr;
    }
    BinsArgs ba;
    ba.Z = d_Z_; ba.Yh = d_Yh_; ba.spectrum = spectrum; ba.steer = snap.steer; ba.freqs = d_freq_;
    ba.n_frames = F; ba.frames_ws = FT; ba.frame_off = Phist_; ba.n_streams = So_; ba.n_mics = MF_; ba.kp1 = snap.kp1;
    ba.n_dirs = D_; ba.steer_dir_stride = snap.steer_dir_stride;
    ba.z48 = z48_ ? 1 : 0;
    ba.cfg = cfg_; ba.gssW = d_gssW_; ba.mpf = d_mpf_; ba.gss_reset_mask = snap.gss_reset_mask;
    // Backward transform.  BF_PRECISION_REFERENCE (the default): in double behind every node (istft_w64_kernel at N = 1024: with it the float
    // output of das, phase, phasempf, gss and mcra equals the oracle's bit for bit).  BF_PRECISION_MIXED: in fp32 (istft32_kernel) wherever
    // the per-bin stage can emit f32x2 rows: mvdr / lcmv (band-limited rows: half the row traffic, no zero-fill), das / phase through the
    // bin pipeline, phasempf.  gsc (its sample-serial NLMS branches on the aligned signals), a spectrum dump and the other FFT sizes: always
    // in double.
    const bool cov_node = cfg_.algo == BF_MVDR || cfg_.algo == BF_LCMV;
    const bool want32 = cfg_.precision == BF_PRECISION_MIXED && cfg_.algo != BF_GSC && N_ == 1024 && spectrum == nullptr;
    // mvdr / lcmv hand the fp32 transform f32x2 rows holding only problem 0 and the in-band problems (everything else is zero,
    // mvdr.cpp:103); das / phase through the bin pipeline: f32x2 rows too (every problem written)
    const bool pointwise32 = (cfg_.algo == BF_DAS || cfg_.algo == BF_PHASE) && want32;
    ba.yh32 = ((z48_ && want32) || pointwise32) ? 1 : 0;
    // phasempf: the recursion's y_fft has no reader but the fp32 backward transform either: f32x2 rows in the |out_int|^2 slots
    ba.mpf32 = (cfg_.algo == BF_PHASEMPF && want32) ? 1 : 0;
    const bool istft32 = ba.yh32 != 0 || ba.mpf32 != 0;
    ba.yh_lo = 0; ba.yh_hi = NQ_ - 1;
    // mvdr / lcmv rows in front of a backward transform (no dump): only problem 0 and the band's problems exist (everything else is zero,
    // mvdr.cpp:103, and is neither written nor read): f32x2 rows into istft32_kernel, f64x2 rows into istft_w64_kernel<true>
    if (cov_node && spectrum == nullptr && N_ == 1024) {
        int klo = N_, khi = 0;
        for (int q = 1; q < NQ_; ++q) {
            const double f = std::fabs(freqs_[q]);  // problem q = bin q for q <= N/2 + 1
            if (f >= cfg_.freq_min && f <= cfg_.freq_max) { if (q < klo) klo = q; if (q > khi) khi = q; }
        }
        if (khi < N_ / 2 - 1 && klo <= khi) { ba.yh_lo = klo; ba.yh_hi = khi; }
        else if (klo > khi) { ba.yh_lo = 1; ba.yh_hi = 0; }  // empty band: only problem 0
    }
    // phasempf with at least a quarter as many streams as CUs, default precision, no dump: the recursion kernel runs the backward transform too
    // (mask_kernels.hip mpf_rec_istft_kernel: a block per stream, the y_fft rows stay in LDS); with fewer streams its blocks leave the chip empty
    const bool rec_istft = cfg_.algo == BF_PHASEMPF && N_ == 1024 && spectrum == nullptr && !ba.mpf32 && (long)So_ * 4 >= n_cus_;
    if (rec_istft) {
        ba.rec_istft = 1;
        ba.rec_y = d_yraw_; ba.rec_tail_in = d_tail_[tail_cur_]; ba.rec_tail_out = d_tail_[tail_cur_ ^ 1];
        ba.rec_tw_w64 = d_tw_w64_; ba.rec_win = d_win_;
    }
    bool fused = false;
    if (try_fused) {
        const hipError_t fe = ks_->stft_bins(sa, ba, n_cus_, stream);
        if (fe == hipSuccess) {
            fused = true;
        } else if (fe != hipErrorNotSupported) {
            PIPE_HIP(fe);
        } else {  // the launcher declined: fall back to the two-kernel chain
            (void)hipGetLastError();
            rc = ensure((void **)&d_Z_, &Z_cap_, (size_t)S_ * FT * NP_ * N_ * sizeof(f64x2));
            if (rc != BF_OK) return rc;
            sa.Z = d_Z_;
            ba.Z = d_Z_;
        }
    }
    if (!fused) PIPE_HIP(ks_->stft(sa, n_cus_, stream));

    // ring-buffer carry (util.h:305-308)
    if (layout == BF_PLANAR) {
        PIPE_HIP(hipMemcpy2DAsync(d_hist2_[hist_cur_], H_ * sizeof(float), x + (F - 1) * H_, (size_t)mic_stride * sizeof(float),
                                  H_ * sizeof(float), (size_t)S_ * M_, hipMemcpyDeviceToDevice, stream));
    } else {
        PIPE_HIP(hipMemcpy2DAsync(d_hist2_[hist_cur_], (size_t)H_ * M_ * sizeof(float), x + (F - 1) * (long)H_ * M_,
                                  (size_t)F * H_ * M_ * sizeof(float), (size_t)H_ * M_ * sizeof(float), (size_t)S_,
                                  hipMemcpyDeviceToDevice, stream));
    }

    if (!fused) PIPE_HIP(ks_->bins(ba, n_cus_, stream));

    if (Phist_ > 0)  // keep the last Phist frames' spectra for the next call
        PIPE_HIP(hipMemcpy2DAsync(d_zhist_, (size_t)Phist_ * frame_elems * zsz_, (const char *)d_Z_ + (size_t)F * frame_elems * zsz_,
                                  (size_t)FT * frame_elems * zsz_, (size_t)Phist_ * frame_elems * zsz_, (size_t)S_,
                                  hipMemcpyDeviceToDevice, stream));

    IstftArgs ia;
    ia.Yh = d_Yh_; ia.y = (cfg_.algo == BF_PHASEMPF || cfg_.algo == BF_GSC) ? d_yraw_ : y; ia.tail_in = d_tail_[tail_cur_];
    ia.tail_out = d_tail_[tail_cur_ ^ 1]; ia.tw = d_tw_; ia.win = d_win_; ia.n_frames = F; ia.n_streams = So_;
    ia.tw32 = istft32 ? d_tw32_ : nullptr;
    ia.tw_w64 = d_tw_w64_;
    ia.yh32 = ba.yh32; ia.yh_lo = ba.yh_lo; ia.yh_hi = ba.yh_hi;
    if (ba.mpf32) {  // rows of 8-byte elements behind the f64x2 rows (where aux lives), every problem written
        ia.Yh = d_Yh_ + (size_t)So_ * F * YS_;
        ia.yh32 = 1; ia.yh_lo = 0; ia.yh_hi = NQ_ - 1;
    }
    ia.frames = d_frames_;
    ia.post_amp = (cfg_.algo == BF_MVDR || cfg_.algo == BF_LCMV || cfg_.algo == BF_GSS) ? cfg_.out_amp : 1.0;
    ia.use_post_amp = (cfg_.algo == BF_MVDR || cfg_.algo == BF_LCMV || cfg_.algo == BF_GSS) ? 1 : 0;
    if (!rec_istft) PIPE_HIP(ks_->istft(ia, n_cus_, stream));
    tail_cur_ ^= 1;

    if (cfg_.algo == BF_PHASEMPF)
        PIPE_HIP(ks_->smooth(d_yraw_, y, d_smooth_, F, So_, cfg_.smooth_size, stream));
    if (cfg_.algo == BF_GSC)
        PIPE_HIP(ks_->gsc_nlms(d_yraw_, y, d_nlms_, F * H_, S_, M_, cfg_, stream));
    return BF_OK;
}

}  // namespace

BinPipeline *BinPipeline::create(const bf_config &cfg, int n_cus) { return new BinPipelineImpl(cfg, n_cus); }

}  // namespace bf
