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

class BinPipeline {
   public:
    // nullptr when the algorithm is not built
    static BinPipeline *create(const bf_config &cfg, int n_cus);
    virtual ~BinPipeline() {}
    virtual int init() = 0;
    virtual int reset() = 0;
    virtual int upload_steering(const std::vector<SteeringSet> &dirs, hipStream_t stream) = 0;  // one set per look direction
    virtual void on_theta_changed(int dir = -1) = 0;  // dir < 0: every look direction
    virtual void set_columns(int kp1) = 0;  // interferer added/removed (lcmv.cpp:266-305)
    virtual int run(const float *x_dev, long n_frames, float *y_dev, f64x2 *spectrum_dev, hipStream_t stream, int layout,
                    long mic_stride) = 0;
    virtual size_t state_bytes() const = 0;
    virtual int get_state(void *host) = 0;
    virtual int set_state(const void *host) = 0;
    const std::string &error() const { return err_; }

   protected:
    std::string err_;
};

}  // namespace bf
