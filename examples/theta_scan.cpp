// theta_scan.cpp -- the reference's theta controllers as one batch per decision.
//
//   theta_scan <das|phase|mvdr|lcmv> <beamform_config.yaml> <in.f32> <n_dirs> [windows_per_block=50]
//
// scripts/energy2theta.py listens to the beamformer output, takes the RMS of the last num_win = 50 windows
// (get_energy_from_list, energy2theta.py:23-27), publishes a new /theta and waits for the next 50 windows -- one
// candidate direction per 0.5 s.  With bfcore the same 50 windows are beamformed towards n_dirs candidates at once
// (bf_config.n_dirs, bf_set_thetas) and bf_stream_rms gives every candidate's energy: one line per block
//   "<block> <best_theta_deg> <best_rms>  | <rms of every direction>"
// in.f32: planar float32 [n_mics][n_samples].  State carries across blocks exactly as in a running node.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/bfcore.h"

int main(int argc, char **argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: %s <das|phase|mvdr|lcmv> <config.yaml> <in.f32> <n_dirs> [windows_per_block]\n", argv[0]);
        return 2;
    }
    const char *names[] = {"das", "mvdr", "lcmv", "gss", "phase"};
    int algo = -1;
    for (int i = 0; i < 5; ++i)
        if (!strcmp(argv[1], names[i])) algo = i;
    bf_config cfg;
    if (algo < 0 || bf_config_init(&cfg, algo) != BF_OK || bf_config_load_yaml(&cfg, argv[2]) != BF_OK) {
        fprintf(stderr, "bad algo or config\n");
        return 2;
    }
    const int D = atoi(argv[4]);
    const long W = argc > 5 ? atol(argv[5]) : 50;  // num_win, energy2theta.py:11
    cfg.n_dirs = D;
    FILE *fi = fopen(argv[3], "rb");
    if (!fi) return 2;
    fseek(fi, 0, SEEK_END);
    const size_t bytes = ftell(fi);
    fseek(fi, 0, SEEK_SET);
    const int M = cfg.n_mics, H = cfg.hop;
    const size_t per_mic = bytes / sizeof(float) / M;
    std::vector<float> in((size_t)M * per_mic);
    if (fread(in.data(), sizeof(float), in.size(), fi) != in.size()) return 2;
    fclose(fi);

    bf_handle *bf = nullptr;
    if (bf_create(&cfg, &bf) != BF_OK) {
        fprintf(stderr, "bf_create: %s\n", bf_last_error(nullptr));
        return 1;
    }
    std::vector<double> thetas(D), rms(D);
    for (int d = 0; d < D; ++d) thetas[d] = -180.0 + 360.0 * d / D;
    if (bf_set_thetas(bf, thetas.data(), D) != BF_OK) return 1;

    float *x_dev = nullptr, *y_dev = nullptr;
    if (hipMalloc((void **)&x_dev, (size_t)M * W * H * sizeof(float)) != hipSuccess) return 1;
    if (hipMalloc((void **)&y_dev, (size_t)D * W * H * sizeof(float)) != hipSuccess) return 1;
    const long periods = (long)(per_mic / H);
    for (long b = 0; (b + 1) * W <= periods; ++b) {
        // planar block [M][W*H] out of the planar file [M][per_mic]
        if (hipMemcpy2D(x_dev, (size_t)W * H * sizeof(float), in.data() + (size_t)b * W * H, per_mic * sizeof(float),
                        (size_t)W * H * sizeof(float), M, hipMemcpyHostToDevice) != hipSuccess)
            return 1;
        if (bf_process_batch_device(bf, x_dev, (size_t)W, y_dev, nullptr, nullptr) != BF_OK ||
            bf_stream_rms(bf, y_dev, (size_t)W, rms.data(), nullptr) != BF_OK) {
            fprintf(stderr, "bfcore: %s\n", bf_last_error(bf));
            return 1;
        }
        int best = 0;
        for (int d = 1; d < D; ++d)
            if (rms[d] > rms[best]) best = d;
        printf("%ld %.3f %.9g |", b, thetas[best], rms[best]);
        for (int d = 0; d < D; ++d) printf(" %.9g", rms[d]);
        printf("\n");
    }
    (void)hipFree(x_dev);
    (void)hipFree(y_dev);
    bf_destroy(bf);
    return 0;
}
