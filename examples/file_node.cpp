// file_node.cpp -- a beamformer "node" with files in place of JACK/ROS.
//
//   file_node <das|mvdr|lcmv|gss|phase|phasempf|mcra|gsc> <beamform_config.yaml> <in.f32|in.wav> <out.f32|out.wav> [theta_script|-] [out_rate]
//
// in.f32: planar float32 [n_mics][n_samples]; in.wav: a multichannel WAV file (bf_wav_read: PCM 16/24/32 or float32).
// The node is driven exactly as JACK drives the reference: one jack_callback(period, 0) per period, planar per-mic
// pointers in, one period of samples out.  out.wav is what rosjack's write_file option produces (rosjack.cpp:189-210,
// 404-409: mono PCM16, one sf_write_float per callback); out.f32 keeps the raw float32 samples.
// theta_script (optional, "-" for none): lines "<callback_index> <degrees>" = /theta messages.
// out_rate (optional) = rosjack's ros_output_sample_rate: when it differs from the input rate every period goes through the
// sample-rate converter before it is written (rosjack.cpp:410-427 convert_to_sample_rate + the resampled write_file branch).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../include/bf_node_shim.hpp"

static std::vector<float> g_in;
static std::vector<float *> g_ptrs;
static size_t g_samples_per_mic = 0, g_pos = 0;
static int g_mics = 0;
static FILE *g_out = nullptr;
static bf_wav_writer *g_wav = nullptr;
static bf_resampler *g_rs = nullptr;
static std::vector<float> g_rs_buf;
static bool g_rosjack_stage = false;  // argv[7] = "rosjack": BF_RS_ROSJACK (bfcore.h) instead of the complete stream conversion

static float **input_from_files(int n) {            // stands in for rosjack.cpp:538-547
    for (int m = 0; m < g_mics; ++m) g_ptrs[m] = g_in.data() + (size_t)m * g_samples_per_mic + g_pos;
    g_pos += n;
    return g_ptrs.data();
}
static void output_to_file(float *data, int n, int) {  // rosjack.cpp:356 output_to_rosjack, write_file branch :404-409
    if (g_rs && g_rosjack_stage) {  // the stage as rosjack runs it: a period may be dropped, at most one block leaves per callback
        int emitted = 0;
        g_rs_buf.resize((size_t)n);
        if (bf_resampler_callback(g_rs, data, g_rs_buf.data(), &emitted, nullptr) != BF_OK || !emitted) return;
        data = g_rs_buf.data();
    } else if (g_rs) {  // convert_to_sample_rate (rosjack.cpp:311-338) as a complete stream conversion: whatever the converter releases
        size_t got = 0;
        g_rs_buf.resize(bf_resampler_out_count(g_rs, (size_t)n) + 1);
        if (bf_resampler_process(g_rs, data, (size_t)n, g_rs_buf.data(), g_rs_buf.size(), &got) != BF_OK) return;
        data = g_rs_buf.data();
        n = (int)got;
    }
    if (g_wav)
        bf_wav_writer_write(g_wav, data, (size_t)n);
    else
        fwrite(data, sizeof(float), n, g_out);
}
static bool ends_with(const char *s, const char *suf) {
    const size_t a = strlen(s), b = strlen(suf);
    return a >= b && !strcmp(s + a - b, suf);
}

int main(int argc, char **argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: %s <algo> <config.yaml> <in.f32|.wav> <out.f32|.wav> [theta_script|-] [out_rate] [rosjack]\n", argv[0]);
        return 2;
    }
    const char *names[] = {"das", "mvdr", "lcmv", "gss", "phase", "phasempf", "mcra", "gsc"};
    int algo = -1;
    for (int i = 0; i < 8; ++i)
        if (!strcmp(argv[1], names[i])) algo = i;
    bf_config cfg;
    if (algo < 0 || bf_config_init(&cfg, algo) != BF_OK || bf_config_load_yaml(&cfg, argv[2]) != BF_OK) {
        fprintf(stderr, "bad algo or config\n");
        return 2;
    }
    g_mics = cfg.n_mics;
    float *planar = nullptr;
    int ch = 0, rate = (int)cfg.sample_rate;
    int rc = ends_with(argv[3], ".wav") ? bf_wav_read(argv[3], &planar, &ch, &g_samples_per_mic, &rate)
                                        : bf_planar_f32_read(argv[3], g_mics, &planar, &g_samples_per_mic);
    if (rc != BF_OK || (ends_with(argv[3], ".wav") && ch < g_mics)) {
        fprintf(stderr, "cannot read %s (%s)\n", argv[3], bf_strerror(rc));
        bf_wav_free(planar);  // a readable file with too few channels
        return 2;
    }
    if (ends_with(argv[3], ".wav")) cfg.sample_rate = rate;  // rosjack_sample_rate = what the "server" runs at
    g_in.assign(planar, planar + (size_t)g_mics * g_samples_per_mic);  // the first n_mics channels
    bf_wav_free(planar);
    int out_rate = argc > 6 ? atoi(argv[6]) : (int)cfg.sample_rate;
    if (out_rate != (int)cfg.sample_rate && bf_resampler_create((int)cfg.sample_rate, out_rate, &g_rs) != BF_OK) {
        fprintf(stderr, "invalid output sample rate %d: keeping %d\n", out_rate, (int)cfg.sample_rate);  // rosjack.cpp:170-172
        out_rate = (int)cfg.sample_rate;
    }
    g_rosjack_stage = g_rs && argc > 7 && !strcmp(argv[7], "rosjack");
    if (g_rosjack_stage && bf_resampler_set_mode(g_rs, BF_RS_ROSJACK, cfg.hop) != BF_OK) return 2;
    if (ends_with(argv[4], ".wav")) {
        if (bf_wav_writer_open(argv[4], out_rate, &g_wav) != BF_OK) return 2;
    } else {
        g_out = fopen(argv[4], "wb");
        if (!g_out) return 2;
    }
    g_ptrs.resize(g_mics);
    std::map<long, float> thetas;
    if (argc > 5 && strcmp(argv[5], "-")) {
        FILE *ft = fopen(argv[5], "r");
        long k;
        float d;
        while (ft && fscanf(ft, "%ld %f", &k, &d) == 2) thetas[k] = d;
        if (ft) fclose(ft);
    }
    bfshim::Node node;
    if (node.start(cfg, input_from_files, output_to_file)) return 1;
    const long periods = (long)(g_samples_per_mic / cfg.hop);
    for (long k = 0; k < periods; ++k) {
        if (thetas.count(k)) node.theta_roscallback(thetas[k]);
        node.jack_callback((uint32_t)cfg.hop, nullptr);
    }
    node.stop();
    if (g_wav) bf_wav_writer_close(g_wav);
    if (g_out) fclose(g_out);
    if (g_rs) bf_resampler_destroy(g_rs);
    fprintf(stderr, "%s: %ld callbacks, %d mics\n", names[algo], periods, g_mics);
    return 0;
}
