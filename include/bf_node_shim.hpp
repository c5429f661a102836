// bf_node_shim.hpp -- C++ host-side mirror of the reference node skeleton, on top of the C ABI.
//
// A balkce/beamform node is (das.cpp:72-145):
//     handle_params(); rosjack_create(..., jack_callback); prepare_overlap_and_add();
//     <alloc + FFTW plans>; update_weights(true); READY = true; ros::spin();
//     int jack_callback(jack_nframes_t nframes, void*) { in = input_from_rosjack(nframes);
//         do_overlap(in, out, nframes, apply_weights); output_to_rosjack(out, nframes, output_type); }
//     void theta_roscallback(msg) { angle = msg->data; update_weights(); }
// This header keeps those names and call shapes so a node's source changes by a few lines
// (INTEGRATION.md shows the diff); everything between input_from_rosjack and output_to_rosjack
// runs on the MI355X behind bf_process_hop.  Header-only, no ROS/JACK dependency: the two I/O
// functions are supplied by the embedding program (rosjack in the reference, files in the example).
#pragma once

#include <cstdint>
#include <cstdio>
#include <vector>

#include "bfcore.h"

namespace bfshim {

typedef float rosjack_data;        // rosjack.h:36 (jack_default_audio_sample_t)
typedef uint32_t jack_nframes_t;   // <jack/types.h>

// rosjack.h:98,100 -- provided by the embedding program
typedef rosjack_data **(*input_fn)(int data_length);
typedef void (*output_fn)(rosjack_data *data, int data_length, int output_type);

struct Node {
    bf_handle *handle = nullptr;
    bool READY = false;            // das.cpp:15
    int output_type = 0;           // rosjack.h:27-30
    input_fn input_from_rosjack = nullptr;
    output_fn output_to_rosjack = nullptr;
    std::vector<rosjack_data> out;

    // main(): everything between rosjack_create() and READY = true (das.cpp:119-140)
    int start(const bf_config &cfg, input_fn in, output_fn outp) {
        input_from_rosjack = in;
        output_to_rosjack = outp;
        out.assign(cfg.hop, 0.0f);
        int rc = bf_create(&cfg, &handle);
        if (rc != BF_OK) {
            fprintf(stderr, "bf_create: %s (%s)\n", bf_strerror(rc), bf_last_error(nullptr));
            return 1;               // the reference's "JACK agent could not be created" convention: 1 = fail
        }
        READY = true;
        return 0;
    }

    // int jack_callback(jack_nframes_t nframes, void *arg)  (das.cpp:72-92); 0 = keep running
    int jack_callback(jack_nframes_t nframes, void * /*arg*/) {
        if (out.size() < nframes) out.resize(nframes);
        if (READY) {
            rosjack_data **in = input_from_rosjack((int)nframes);
            if (bf_process_hop(handle, in, out.data(), nframes) != BF_OK) {
                fprintf(stderr, "bf_process_hop: %s\n", bf_last_error(handle));
                for (jack_nframes_t i = 0; i < nframes; i++) out[i] = 0.0f;
            }
        } else {
            for (jack_nframes_t i = 0; i < nframes; i++) out[i] = 0.0f;   // das.cpp:81-85
        }
        output_to_rosjack(out.data(), (int)nframes, output_type);
        return 0;
    }

    // void theta_roscallback(const std_msgs::Float32::ConstPtr&)  (das.cpp:94-99)
    void theta_roscallback(float degrees) { bf_set_theta(handle, (double)degrees); }
    // void interf_theta_roscallback(const beamform::InterfTheta::ConstPtr&)  (lcmv.cpp:258-309)
    void interf_theta_roscallback(unsigned id, float degrees) { bf_set_interference(handle, id, (double)degrees); }

    void stop() {
        READY = false;
        bf_destroy(handle);
        handle = nullptr;
    }
};

}  // namespace bfshim
