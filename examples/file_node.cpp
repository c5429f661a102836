// file_node.cpp -- a beamformer "node" with files in place of JACK/ROS.
//
//   file_node <das|mvdr|lcmv|gss|phase|phasempf|mcra|gsc> <beamform_config.yaml> <in.f32> <out.f32> [theta_script]
//
// in.f32: planar float32 [n_mics][n_samples]; the node is driven exactly as JACK drives the
// reference: one jack_callback(512, 0) per period, planar per-mic pointers in, 512 samples out.
// theta_script (optional): lines "<callback_index> <degrees>" = /theta messages.
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

static float **input_from_files(int n) {            // stands in for rosjack.cpp:538-547
    for (int m = 0; m < g_mics; ++m) g_ptrs[m] = g_in.data() + (size_t)m * g_samples_per_mic + g_pos;
    g_pos += n;
    return g_ptrs.data();
}
static void output_to_file(float *data, int n, int) { fwrite(data, sizeof(float), n, g_out); }  // rosjack.cpp:356

int main(int argc, char **argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: %s <algo> <config.yaml> <in.f32> <out.f32> [theta_script]\n", argv[0]);
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
    FILE *fi = fopen(argv[3], "rb");
    g_out = fopen(argv[4], "wb");
    if (!fi || !g_out) return 2;
    fseek(fi, 0, SEEK_END);
    const size_t bytes = ftell(fi);
    fseek(fi, 0, SEEK_SET);
    g_mics = cfg.n_mics;
    g_samples_per_mic = bytes / sizeof(float) / g_mics;
    g_in.resize((size_t)g_mics * g_samples_per_mic);
    if (fread(g_in.data(), sizeof(float), g_in.size(), fi) != g_in.size()) return 2;
    fclose(fi);
    g_ptrs.resize(g_mics);
    std::map<long, float> thetas;
    if (argc > 5) {
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
    fclose(g_out);
    fprintf(stderr, "%s: %ld callbacks, %d mics\n", names[algo], periods, g_mics);
    return 0;
}
