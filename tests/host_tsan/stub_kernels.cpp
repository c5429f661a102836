// Kernel launchers of libbfcore stubbed for the host ThreadSanitizer harness (TEST INFRASTRUCTURE).
// No arithmetic of the hot path happens here: every launcher only (1) reads the tables it was handed, on the calling
// (processing) thread, so that ThreadSanitizer sees the same host-memory accesses a real launch would order on the
// stream, and (2) checks that the arguments of ONE batch are consistent with each other.
#include <atomic>
#include <cmath>

#include "../../beamform_amd/csrc/kernels.hpp"
#include "../../beamform_amd/csrc/pipeline_kernels.hpp"

std::atomic<long> g_inconsistent{0};
std::atomic<long> g_launches{0};

namespace bf {

static hipError_t launch_stft(const StftArgs &, int, hipStream_t) { return hipSuccess; }
static hipError_t launch_istft(const IstftArgs &, int, hipStream_t) { return hipSuccess; }
static hipError_t launch_smooth(const float *, float *, double *, long, int, int, hipStream_t) { return hipSuccess; }
static hipError_t launch_gsc_nlms(const float *, float *, float *, long, int, int, const bf_config &, hipStream_t) { return hipSuccess; }

// The per-bin stage: the batch must see ONE consistent {column count, steering table}.  The table is laid out
// [dir][col][mic][bin] with kp1 columns; every uploaded steering entry of mic >= 1 has modulus 1, memory that was never
// uploaded reads 0 (the stub's hipMalloc zero-fills).
static hipError_t launch_bins(const BinsArgs &a, int, hipStream_t) {
    g_launches++;
    const int M = a.n_mics, kp1 = a.kp1;
    if (a.steer_dir_stride != (long)kp1 * M * 1024) g_inconsistent++;
    if (M > 1)
        for (int d = 0; d < a.n_dirs; ++d)
            for (int c = 0; c < kp1; ++c)
                for (int j = 1; j < 1024; j += 97) {
                    const f64x2 w = a.steer[d * a.steer_dir_stride + ((long)c * M + 1) * 1024 + j];
                    if (std::fabs(w.x * w.x + w.y * w.y - 1.0) > 1e-9) g_inconsistent++;
                }
    return hipSuccess;
}

// the fused STFT + per-bin launcher declines: the harness exercises the two-kernel chain's host logic
static hipError_t launch_stft_bins(const StftArgs &, const BinsArgs &, int, hipStream_t) { return hipErrorNotSupported; }

// the harness runs at hop 512; the other sizes share the same host code
static const KernelSet g_stub_set = {1024, &launch_stft, &launch_bins, &launch_stft_bins, &launch_istft, &launch_smooth, &launch_gsc_nlms};
const KernelSet *kernel_set_n128() { return &g_stub_set; }
const KernelSet *kernel_set_n256() { return &g_stub_set; }
const KernelSet *kernel_set_n512() { return &g_stub_set; }
const KernelSet *kernel_set_n1024() { return &g_stub_set; }
const KernelSet *kernel_set_n2048() { return &g_stub_set; }
const KernelSet *kernel_set_n4096() { return &g_stub_set; }
const KernelSet *kernel_set_n8192() { return &g_stub_set; }
hipError_t prepare_das_f64_w64(const DasF64Args &, int, hipStream_t) { return hipSuccess; }
hipError_t launch_das_f64_w64(const DasF64Args &, int, hipStream_t) { return hipSuccess; }
hipError_t launch_interleaved_to_planar(const float *, float *, long, int, int, hipStream_t) { return hipSuccess; }
bool das_f64_writes_hist(const DasF64Args &) { return false; }
size_t das_f64_sched_ws_bytes() { return 256; }
size_t das_f64_ring_bytes(int, int) { return 0; }

hipError_t prepare_das_fused(const DasFusedArgs &, hipStream_t) { return hipSuccess; }
hipError_t launch_das_fused(const DasFusedArgs &a, hipStream_t) {
    g_launches++;
    const int np = (a.n_mics + 1) / 2;
    double acc = 0;  // read the whole gain table this launch was given
    for (long i = 0; i < (long)a.n_dirs * np * 1024; ++i) acc += a.gains[i].x;
    if (!(acc == acc)) g_inconsistent++;
    return hipSuccess;
}
hipError_t prepare_das_fused_wave2048(const DasFusedArgs &, hipStream_t) { return hipSuccess; }
hipError_t launch_das_fused_wave2048(const DasFusedArgs &a, hipStream_t s) { return launch_das_fused(a, s); }
hipError_t launch_das_fused_dirs(const DasFusedArgs &a, int, int, hipStream_t s) { return launch_das_fused(a, s); }
hipError_t launch_stream_rms(const float *, long, int, double *, hipStream_t) { return hipSuccess; }
hipError_t launch_das_hermitian_dump(const f32x2 *, f64x2 *, long, hipStream_t) { return hipSuccess; }
hipError_t launch_das_fused_gen(const DasFusedArgs &a, int, hipStream_t s) { return launch_das_fused(a, s); }
hipError_t launch_das_hermitian_dump_gen(const f32x2 *, f64x2 *, long, int, hipStream_t) { return hipSuccess; }

}  // namespace bf
