// ThreadSanitizer harness around libbfcore's host control plane (TEST INFRASTRUCTURE).
// The situation of das.cpp:94-99 (ROS callback thread: angle = msg->data; update_weights()) against das.cpp:72-92 (JACK
// thread: jack_callback) and of lcmv.cpp:258-309 against lcmv.cpp:142-162: one thread hammers bf_set_theta /
// bf_set_interference / bf_get_weights while another runs bf_process_hop.  Built with -fsanitize=thread against the
// host HIP stand-in (stub/hip/hip_runtime.h): any unsynchronised access to the handle's tables makes TSAN fail the
// run (exit code 66), and the stubbed per-bin launcher checks every batch for a torn {column count, table} pair.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../../include/bfcore.h"

extern std::atomic<long> g_inconsistent, g_launches;

static int run_case(int algo, int n_mics, int n_interf, int hops) {
    bf_config cfg;
    bf_config_init(&cfg, algo);
    cfg.n_mics = n_mics;
    for (int m = 0; m < n_mics; ++m) {
        cfg.mic_x[m] = 0.05 * (m % 4) - 0.02 * m;
        cfg.mic_y[m] = 0.04 * (m / 4) + 0.01 * m;
    }
    cfg.n_interf = n_interf;
    for (int k = 0; k < n_interf; ++k) cfg.interf_angle[k] = -60.0 + 70.0 * k;
    bf_handle *h = nullptr;
    int rc = bf_create(&cfg, &h);
    if (rc != BF_OK) {
        fprintf(stderr, "bf_create(%d): %d %s\n", algo, rc, bf_last_error(nullptr));
        return 1;
    }
    std::vector<float> buf((size_t)n_mics * 512, 0.25f), out(512);
    std::vector<const float *> in(n_mics);
    for (int m = 0; m < n_mics; ++m) in[m] = buf.data() + (size_t)m * 512;
    std::atomic<bool> stop{false};
    std::atomic<long> updates{0};
    int bad = 0;
    std::thread ctl([&] {
        unsigned n = 0;
        std::vector<double> w((size_t)1024 * n_mics * 16 * 2);
        while (!stop.load()) {
            bf_set_theta(h, -170.0 + (double)(n * 37 % 340));
            if (algo == BF_LCMV || algo == BF_GSS) {
                // move, append (structural: the tables are re-allocated), remove by moving next to another one
                bf_set_interference(h, 1, -80.0 + (double)(n % 40));
                if (n % 5 == 0) bf_set_interference(h, 9, 120.0 + (double)(n % 7));
                if (n % 5 == 3 && bf_n_interferers(h) > 1) bf_set_interference(h, 2, -80.0 + (double)(n % 40) + 0.25);
                bf_get_weights(h, w.data());
            }
            ++n;
            updates++;
            std::this_thread::sleep_for(std::chrono::microseconds(150));  // a topic, not a spin: std::mutex is not fair
        }
    });
    for (int t = 0; t < hops; ++t) {
        rc = bf_process_hop(h, in.data(), out.data(), 512);
        if (rc != BF_OK) {
            fprintf(stderr, "bf_process_hop: %d %s\n", rc, bf_last_error(h));
            bad = 1;
            break;
        }
        if (t % 64 == 0) {  // checkpoint from the processing thread while /theta keeps arriving
            std::vector<char> blob(bf_state_size(h));
            if (bf_get_state(h, blob.data(), blob.size()) != BF_OK) bad = 1;
        }
    }
    stop.store(true);
    ctl.join();
    bf_destroy(h);
    printf("algo %d: %d hops, %ld control updates, %ld launches, %ld inconsistent batches\n", algo, hops, updates.load(),
           g_launches.load(), g_inconsistent.load());
    return bad || g_inconsistent.load() != 0 || updates.load() < 10;
}

int main(int argc, char **argv) {
    const int hops = argc > 1 ? atoi(argv[1]) : 2000;
    int bad = 0;
    bad |= run_case(BF_DAS, 8, 0, hops);
    bad |= run_case(BF_MVDR, 8, 0, hops);
    bad |= run_case(BF_LCMV, 8, 1, hops);
    bad |= run_case(BF_GSS, 4, 1, hops);
    return bad;
}
