// pipeline.hpp -- the fp64 "bin pipeline" used by every node except fused DAS:
//   STFT kernel (window + forward FFTs, spectra to HBM)
//   -> per-bin kernel (the node's apply_weights loop body)
//   -> ISTFT kernel (backward FFT, synthesis window, overlap-add).
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/bfcore.h"
#include "geometry.hpp"

namespace bf {

// What one run() reads of the state the control plane (bf_set_theta / bf_set_interference, another thread) may change:
// taken under the handle's mutex right after the table upload, so a batch sees ONE consistent
// {column count, steering table, pending demixing resets} (the reference guards the same window with READY=false
// plus a sleep, lcmv.cpp:262-307).
// das_f64_pair_kernel's walk over the microphones (pipeline_kernels.hpp DasF64Args::slot_mic / extra_mic)
struct DasSlots {
    int n_tr = 0, extra_mic = -1;
    int slot_mic[8] = {1, 2, 3, 4, 5, 6, 7, 0};
};

struct RunSnapshot {
    int kp1 = 1;
    unsigned long long gss_reset_mask = 0;
    const f64x2 *steer = nullptr;
    long steer_dir_stride = 0;
    const f64x2 *das_gains_mic = nullptr;  // per-microphone Hermitian gains of the frame-pair kernel (das_f64_pair_kernel)
    DasSlots das_slots;
    bool das_mic0_unit = false;            // row 0 of the das weights is identically 1 in that table (das.cpp:33-38; quirk Q3 can leave it 0)
    const f64x2 *das_gains_w64 = nullptr;  // the same gains in the register / lane order of the 64-lane kernel (das_f64_w64.hip)
};

class BinPipeline {
   public:
    // nullptr when the algorithm is not built
    static BinPipeline *create(const bf_config &cfg, int n_cus);
    virtual ~BinPipeline() {}
    virtual int init() = 0;
    virtual int reset(hipStream_t stream) = 0;  // clears enqueued on `stream`
    virtual int upload_steering(const std::vector<SteeringSet> &dirs, hipStream_t stream) = 0;  // one set per look direction
    virtual void on_theta_changed(int dir = -1) = 0;  // dir < 0: every look direction
    virtual void set_columns(int kp1) = 0;  // interferer added/removed (lcmv.cpp:266-305)
    // caller holds the control-plane mutex: hands the pending demixing resets to this run and clears them
    virtual RunSnapshot snapshot_for_run() = 0;
    virtual int run(const float *x_dev, long n_frames, float *y_dev, f64x2 *spectrum_dev, hipStream_t stream, int layout,
                    long mic_stride, const RunSnapshot &snap) = 0;
    // checkpoint of the control-plane side (caller holds the mutex)
    virtual int columns() const = 0;
    virtual unsigned long long pending_resets() const = 0;
    virtual void set_pending_resets(unsigned long long mask) = 0;
    virtual size_t state_bytes() const = 0;
    virtual int get_state(void *host) = 0;
    virtual int set_state(const void *host) = 0;
    const std::string &error() const { return err_; }
    // set by the caller around run(): when non-null and the run has ONE dominant kernel (das fp64 in one launch), the two events are
    // recorded on the run's stream right before and after that launch and kev_recorded is raised (bf_kernel_timing_begin / _end)
    hipEvent_t kev0 = nullptr, kev1 = nullptr;
    bool kev_recorded = false;

   protected:
    std::string err_;
};

}  // namespace bf
