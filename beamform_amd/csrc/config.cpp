// config.cpp -- bf_config defaults and the beamform_config.yaml reader.
//
// Replaces, for the parameters the hot path depends on, the ROS parameter-server
// reads of handle_params (util.h:52-134) and of every node's *_handle_params
// (mvdr.cpp:146-187, lcmv.cpp:171-219, gss.cpp:187-240, phase.cpp:165-191,
// phasempf.cpp:355-475).  Defaults are the launch-file values
// (launch/*.launch), not the getParam fallbacks.
//
// The reader understands the YAML subset those files use: `key: scalar`,
// `key: {k: v, k: v}` flow maps, `#` comments, blank lines.
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/bfcore.h"

namespace {

std::string trim(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) ++a;
    while (b > a && isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}

bool parse_scalar(const std::string &v, double *out) {
    std::string t = trim(v);
    if (t.empty()) return false;
    if (t == "true" || t == "True") { *out = 1; return true; }
    if (t == "false" || t == "False") { *out = 0; return true; }
    char *end = nullptr;
    double d = strtod(t.c_str(), &end);
    if (end == t.c_str()) return false;
    while (*end && isspace((unsigned char)*end)) ++end;
    if (*end) return false;
    *out = d;
    return true;
}

bool parse_flow_map(const std::string &v, std::map<std::string, double> *m) {
    std::string t = trim(v);
    if (t.size() < 2 || t.front() != '{' || t.back() != '}') return false;
    t = t.substr(1, t.size() - 2);
    size_t pos = 0;
    while (pos < t.size()) {
        size_t comma = t.find(',', pos);
        std::string item = t.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
        size_t colon = item.find(':');
        if (colon != std::string::npos) {
            double d;
            if (parse_scalar(item.substr(colon + 1), &d)) (*m)[trim(item.substr(0, colon))] = d;
        }
        if (comma == std::string::npos) break;
        pos = comma + 1;
    }
    return true;
}

}  // namespace

extern "C" {

int bf_config_init(bf_config *c, int algo) {
    if (!c || algo < BF_DAS || algo > BF_GSC) return BF_EINVAL;
    memset(c, 0, sizeof(*c));
    c->algo = algo;
    c->hop = 512;              // JACK period behind "1024-pt FFT" (util.h:261)
    c->sample_rate = 48000.0;  // rosjack_config.yaml:9
    // beamform_config.yaml:15-17 (aira3, the uncommented geometry)
    c->n_mics = 3;
    c->mic_x[0] = 0.000; c->mic_y[0] = 0.000;
    c->mic_x[1] = 0.000; c->mic_y[1] = -0.180;
    c->mic_x[2] = -0.156; c->mic_y[2] = -0.090;
    c->theta = 0.0;  // beamform_config.yaml:2
    c->n_interf = 0; // all angle_interf* are 181 in beamform_config.yaml:43-57
    c->verbose = 1;
    // launch/mvdr.launch:6-10, launch/lcmv.launch:6-11
    c->past_windows = 10;
    c->freq_mag_threshold = 0.001;
    c->freq_max = 16000.0;
    c->freq_min = 100.0;
    c->out_amp = 1.0;
    c->interf_angle_threshold = 1.0;
    c->mu = 0.001;  // launch/gss.launch:11-12
    c->lambda_ = 0.0;
    if (algo == BF_GSS) c->out_amp = 0.1;  // launch/gss.launch:9
    // phase.launch:6 + phase.cpp:180,187 (launch keys min_mag/smooth_size are never read by phase: Q14)
    c->min_phase = 10.0;
    c->mag_mult = 0.1;
    c->mag_threshold = 0.05;
    // launch/phasempf.launch:6-21
    c->min_mag = 0.05;
    c->smooth_size = 3;
    c->mcra_alphaS = 0.95; c->mcra_alphaD = 0.95; c->mcra_alphaD2 = 0.98; c->mcra_delta = 0.001; c->mcra_L = 50;
    c->mpf_alphaS = 0.7; c->mpf_eta = 0.3; c->mpf_rev_gamma = 0.9; c->mpf_rev_delta = 1.0;
    c->noise_floor = 0.001;
    c->out_only_noise = 0; c->out_only_mcra = 0;
    if (algo == BF_PHASEMPF) { c->min_phase = 30.0; c->out_amp = 2.5; }
    if (algo == BF_MCRA) { c->mcra_L = 300; c->out_amp = 3.5; }  // launch/mcra.launch:6-12
    // launch/gsc.launch:6-11
    c->gsc_use_vad = 0; c->gsc_vad_threshold = 0.1; c->gsc_mu0 = 0.0001; c->gsc_mu_max = 0.1; c->gsc_filter_size = 128;
    c->device = 0;
    c->n_streams = 1;
    c->layout = BF_PLANAR;
    c->das_impl = BF_DAS_F64;  // the reference's arithmetic (das.cpp:16-24); BF_DAS_FUSED_F32 is the opt-in
    c->precision = BF_PRECISION_REFERENCE;  // ... between the transforms too; BF_PRECISION_MIXED is the opt-in
    return BF_OK;
}

int bf_config_parse_yaml(bf_config *c, const char *text) {
    if (!c || !text) return BF_EINVAL;
    std::map<int, std::map<std::string, double>> mics;
    std::map<int, double> interf;
    std::string all(text);
    size_t pos = 0;
    while (pos <= all.size()) {
        size_t nl = all.find('\n', pos);
        std::string line = all.substr(pos, nl == std::string::npos ? std::string::npos : nl - pos);
        pos = (nl == std::string::npos) ? all.size() + 1 : nl + 1;
        size_t hash = line.find('#');
        if (hash != std::string::npos) line = line.substr(0, hash);
        line = trim(line);
        if (line.empty()) continue;
        size_t colon = line.find(':');
        if (colon == std::string::npos) continue;
        std::string key = trim(line.substr(0, colon));
        std::string val = trim(line.substr(colon + 1));
        double d = 0;
        if (key.compare(0, 3, "mic") == 0 && key.size() > 3 && isdigit((unsigned char)key[3])) {
            std::map<std::string, double> m;
            if (!parse_flow_map(val, &m)) return BF_EINVAL;
            mics[atoi(key.c_str() + 3)] = m;
            continue;
        }
        if (key.compare(0, 12, "angle_interf") == 0) {
            if (!parse_scalar(val, &d)) return BF_EINVAL;
            interf[atoi(key.c_str() + 12)] = d;
            continue;
        }
        if (!parse_scalar(val, &d)) continue;  // strings (write_file_path: '') etc. are not ours
#define BF_KEY_D(name) if (key == #name) { c->name = d; continue; }
#define BF_KEY_I(name) if (key == #name) { c->name = (int)d; continue; }
        BF_KEY_I(verbose)
        if (key == "initial_angle") { c->theta = d; continue; }
        BF_KEY_I(past_windows) BF_KEY_D(freq_mag_threshold) BF_KEY_D(freq_max) BF_KEY_D(freq_min) BF_KEY_D(out_amp)
        BF_KEY_D(interf_angle_threshold) BF_KEY_D(mu)
        if (key == "lambda") { c->lambda_ = d; continue; }
        BF_KEY_D(min_phase) BF_KEY_D(mag_mult) BF_KEY_D(mag_threshold) BF_KEY_D(min_mag) BF_KEY_I(smooth_size)
        // the mcra node reads the same quantities without the MCRA_ prefix (mcra.cpp:180-217)
        if (c->algo == BF_MCRA) {
            if (key == "alphaS") { c->mcra_alphaS = d; continue; }
            if (key == "alphaD") { c->mcra_alphaD = d; continue; }
            if (key == "alphaD2") { c->mcra_alphaD2 = d; continue; }
            if (key == "delta") { c->mcra_delta = d; continue; }
            if (key == "L") { c->mcra_L = (int)d; continue; }
        }
        if (key == "MCRA_alphaS") { c->mcra_alphaS = d; continue; }
        if (key == "MCRA_alphaD") { c->mcra_alphaD = d; continue; }
        if (key == "MCRA_alphaD2") { c->mcra_alphaD2 = d; continue; }
        if (key == "MCRA_delta") { c->mcra_delta = d; continue; }
        if (key == "MCRA_L") { c->mcra_L = (int)d; continue; }
        if (key == "MPF_alphaS") { c->mpf_alphaS = d; continue; }
        if (key == "MPF_eta") { c->mpf_eta = d; continue; }
        if (key == "MPF_rev_gamma") { c->mpf_rev_gamma = d; continue; }
        if (key == "MPF_rev_delta") { c->mpf_rev_delta = d; continue; }
        BF_KEY_D(noise_floor) BF_KEY_I(out_only_noise) BF_KEY_I(out_only_mcra)
        if (key == "use_vad") { c->gsc_use_vad = (int)d; continue; }  // gsc.cpp:203-256
        if (key == "vad_threshold") { c->gsc_vad_threshold = d; continue; }
        if (key == "mu0") { c->gsc_mu0 = d; continue; }
        if (key == "mu_max") { c->gsc_mu_max = d; continue; }
        if (key == "filter_size") { c->gsc_filter_size = (int)d; continue; }
        if (key == "rosjack_window_size") { c->hop = (int)d; continue; }
        if (key == "rosjack_sample_rate") { c->sample_rate = d; continue; }
#undef BF_KEY_D
#undef BF_KEY_I
    }
    // mic0, mic1, ... consumed until the first missing index (util.h:82-92)
    if (!mics.empty()) {
        int n = 0;
        while (mics.count(n) && n < BF_MAX_MICS) {
            c->mic_x[n] = mics[n].count("x") ? mics[n]["x"] : 0.0;
            c->mic_y[n] = mics[n].count("y") ? mics[n]["y"] : 0.0;
            ++n;
        }
        c->n_mics = n;
    }
    // angle_interf1.. consumed until the first |angle| > 180 or missing index (util.h:101-113)
    if (!interf.empty()) {
        int k = 0;
        while (interf.count(k + 1) && k < BF_MAX_INTERF) {
            double a = interf[k + 1];
            if (!(std::abs(a) <= 180)) break;
            c->interf_angle[k++] = a;
        }
        c->n_interf = k;
    }
    return BF_OK;
}

int bf_config_load_yaml(bf_config *c, const char *path) {
    if (!c || !path) return BF_EINVAL;
    FILE *f = fopen(path, "rb");
    if (!f) return BF_ENOENT;
    std::string text;
    char buf[4096];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, n);
    fclose(f);
    return bf_config_parse_yaml(c, text.c_str());
}

}  // extern "C"
