// pipeline_kernels.hip -- dispatch of the per-bin stage of the fp64 bin pipeline.  The kernels live in
// stft_istft.hip, mask_kernels.hip (das / phase / phasempf / mcra), cov_kernels.hip (mvdr / lcmv) and
// gsc_gss_kernels.hip.
#include "pipeline_kernels.hpp"

#include <hip/hip_runtime.h>

namespace bf {
namespace BF_NTAG {

hipError_t launch_bins(const BinsArgs &a, int n_cus, hipStream_t s) {
    hipError_t e = hipSuccess;
    switch (a.cfg.algo) {
        case BF_DAS:
        case BF_PHASE: e = launch_pointwise(a, s); break;
        case BF_MVDR:
        case BF_LCMV: e = launch_mvdr_lcmv(a, n_cus, s); break;
        case BF_PHASEMPF: e = launch_phasempf(a, n_cus, s); break;
        case BF_GSS: e = launch_gss(a, n_cus, s); break;
        case BF_GSC: e = launch_gsc_align(a, s); break;
        case BF_MCRA: e = launch_mcra_node(a, s); break;
        default: e = hipErrorInvalidValue; break;
    }
    if (e != hipSuccess) return e;
    if (a.spectrum) e = launch_expand_spectrum(a.Yh, a.spectrum, (long)a.n_streams * a.n_frames, s);
    return e;
}

}  // namespace BF_NTAG

// the launchers of this FFT size, as the host pipeline sees them
const KernelSet *BF_CAT2(kernel_set_n, BF_NFFT)() {
    static const KernelSet ks = {BF_NFFT, &BF_NTAG::launch_stft, &BF_NTAG::launch_bins, &BF_NTAG::launch_stft_bins_fused,
                                 &BF_NTAG::launch_istft,
                                 &BF_NTAG::launch_smooth, &BF_NTAG::launch_gsc_nlms};
    return &ks;
}

namespace BF_NTAG {

}  // namespace BF_NTAG
}  // namespace bf
