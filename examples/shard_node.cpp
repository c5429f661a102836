// shard_node.cpp -- one long stream cut into frame ranges across processes (one per GPU), the C way:
// bf_shard_plan / bf_reset_async / bf_process_batch_device from include/bfcore.h and RCCL called directly for the final gather
// (north_star: "Host C++ calls a thin C-ABI ... RCCL over xGMI only for the final gather").  The Python counterpart is
// beamform_amd/shard.py over torch.distributed.
//
//   shard_node <das|mvdr|lcmv|phase> <n_mics> <total_frames> <world> <rank> <id_file> [chunks=4] [steps=3]
//       one process per rank; rank 0 writes the ncclUniqueId into <id_file>, the others wait for it.
//       Every rank materialises its slice of ONE global counter-noise stream (lead hop + warm-up frames + owned frames),
//       starts from a cold handle, and walks its slice in `chunks` pieces: while piece c is computed on the compute stream,
//       piece c-1's owned hops travel to rank 0 on a second stream (grouped ncclSend / ncclRecv = point-to-point over the
//       direct xGMI link of each peer).  Rank 0 prints the step time with and without the gather and a checksum.
//       BF_SHARD_STUB=1: the same schedule with the transfers carried by named pipes through host memory instead of RCCL (which
//       refuses two ranks on one device): every ncclSend / ncclRecv of the real run has its counterpart, sizes are checked at the
//       receiving end and an unmatched transfer blocks -- the way to run all ranks of the gather on a one-GPU box.
//       BF_SHARD_SELF=1: rank 0 sends its own pieces to itself through RCCL as well (grouped ncclSend + ncclRecv): `world` = 1 then
//       executes the communicator set-up and the point-to-point path of the gather on one GPU.
//   shard_node <algo> <n_mics> <total_frames> <world> logical
//       no RCCL: one process plays all `world` ranks one after the other on one GPU (device-to-device copies instead of the
//       gather) and compares the assembled output with the unsharded run of the same stream -- the check that the plan,
//       the cold start and the chunked walk reproduce the single-node result.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cerrno>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <csignal>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../include/bfcore.h"

#define CK(call)                                                                              \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)
#define CKN(call)                                                                              \
    do {                                                                                       \
        ncclResult_t r_ = (call);                                                              \
        if (r_ != ncclSuccess) {                                                               \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #call, ncclGetErrorString(r_)); \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)
#define CKB(call)                                                                             \
    do {                                                                                      \
        int r_ = (call);                                                                      \
        if (r_ != BF_OK) {                                                                    \
            fprintf(stderr, "%s:%d %s: %d %s\n", __FILE__, __LINE__, #call, r_, bf_last_error(nullptr)); \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

// samples [s0, s1) of microphone m of the global stream: the same counter-based hash as beamform_amd.synth.stream_noise, so
// the halo a rank re-reads is bit-identical to what its neighbour owns (no exchange of input)
__global__ void stream_noise_kernel(float *out, long long s0, long long n, int m, int seed) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long h = ((unsigned long long)(s0 + i) + (unsigned long long)(m + 1) * 0x632BE5ABull +
                            (unsigned long long)(seed + 1) * 0x85157AF5ull) & 0xFFFFFFFFull;
    h = ((h ^ (h >> 16)) * 0x45D9F3Bull) & 0xFFFFFFFFull;
    h = ((h ^ (h >> 16)) * 0x45D9F3Bull) & 0xFFFFFFFFull;
    h = h ^ (h >> 16);
    out[i] = (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f;
}

struct Rank {
    bf_shard sh;
    float *x = nullptr, *y = nullptr;  // fed hops (planar [M][n_feed * hop]) and their output
    long long n_feed = 0, n_drop = 0, n_own = 0;
};

static int make_slice(Rank &r, int M, int H, hipStream_t s) {
    r.n_feed = bf_shard_n_feed(&r.sh);
    r.n_drop = bf_shard_n_drop(&r.sh);
    r.n_own = r.sh.hi - r.sh.lo;
    CK(hipMalloc((void **)&r.x, (size_t)M * r.n_feed * H * sizeof(float)));
    CK(hipMalloc((void **)&r.y, (size_t)r.n_feed * H * sizeof(float)));
    const long long n = r.n_feed * H, s0 = bf_shard_first_feed(&r.sh) * H;
    for (int m = 0; m < M; ++m)
        hipLaunchKernelGGL(stream_noise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, r.x + (size_t)m * n, s0, n, m, 1234);
    CK(hipGetLastError());
    return 0;
}

// owned hops of piece c when a slice of n_feed fed hops (the first n_drop of them warm-up) is cut into `chunks` pieces:
// [*own0, *own0 + return value) in owned-hop indices.  Every rank evaluates this for every rank, so that both ends of a transfer
// agree on its size (the pieces of two ranks differ: rank 0 feeds no halo, and a short slice has fewer pieces than `chunks`).
static long long piece_owned(long long n_feed, long long n_drop, int chunks, int c, long long *own0) {
    const long long per = (n_feed + chunks - 1) / chunks;
    const long long f0 = c * per, f1 = (f0 + per < n_feed) ? f0 + per : n_feed;
    const long long o0 = f0 > n_drop ? f0 : n_drop;
    *own0 = o0 - n_drop;
    return f1 > o0 ? f1 - o0 : 0;
}

// the rank's slice in `chunks` pieces on `cs`; in EVERY round c = 0 .. chunks-1, `on_piece(c, first owned hop of the piece, its hops)`
// is called with the compute stream positioned right behind that piece -- with 0 hops when this rank's piece c is empty or
// all warm-up: the gather's receiving side has to post the peers' piece c in that round all the same
template <typename F>
static int walk(bf_handle *bf, Rank &r, int M, int H, int chunks, hipStream_t cs, F &&on_piece) {
    CKB(bf_reset_async(bf, cs));  // cold start, ordered on the compute stream
    const long long per = (r.n_feed + chunks - 1) / chunks;
    // planar input with n_feed * H samples per microphone: a piece is a column range, described by the handle's mic stride
    for (int c = 0; c < chunks; ++c) {
        const long long f0 = c * per, f1 = (f0 + per < r.n_feed) ? f0 + per : r.n_feed;
        if (f1 > f0)
            CKB(bf_process_batch_device_strided(bf, r.x + f0 * H, (size_t)(f1 - f0), r.y + f0 * H, cs, (long)(r.n_feed * H)));
        long long own0 = 0;
        const long long n = piece_owned(r.n_feed, r.n_drop, chunks, c, &own0);
        int rc = on_piece(c, own0, n);
        if (rc) return rc;
    }
    return 0;
}

// ---- the gather's transport: RCCL point-to-point, or (BF_SHARD_STUB=1) named pipes through host memory --------------------------
struct Transport {
    bool stub = false;
    ncclComm_t comm = nullptr;
    int rank = 0;
    std::string base;  // stub: pipes <base>.<src>.<dst>
    std::vector<float> host;
    // stub: every pipe is opened once and stays open for the whole run.  (Opening and closing one per message loses data: a sender that
    // reopens while the receiver still holds the previous message's descriptor writes into a pipe whose last reader then closes --
    // EPIPE / SIGPIPE on the sender, and the other ranks wait for ever.)
    std::map<int, int> wfd, rfd;
    int pipe_fd(std::map<int, int> &m, int peer, const std::string &path, int flags) {
        auto it = m.find(peer);
        if (it != m.end()) return it->second;
        const int fd = open(path.c_str(), flags);  // O_WRONLY blocks until the receiver opens its end, and the reverse
        if (fd >= 0) m[peer] = fd;
        return fd;
    }

    int init(int world, int rank_, const char *id_file, bool stub_) {
        stub = stub_;
        rank = rank_;
        base = id_file;
        if (stub) {
            signal(SIGPIPE, SIG_IGN);  // a vanished peer is a write error here, not a silent death
            for (int r = 1; r < world; ++r) {  // every rank may create them; EEXIST is fine
                const std::string p = base + "." + std::to_string(r) + ".0";
                if (mkfifo(p.c_str(), 0600) != 0 && errno != EEXIST) return 1;
            }
            return 0;
        }
        ncclUniqueId id;
        if (rank == 0) {
            CKN(ncclGetUniqueId(&id));
            const std::string tmp = base + ".tmp";
            FILE *f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(&id, sizeof(id), 1, f) != 1) return 1;
            fclose(f);
            rename(tmp.c_str(), base.c_str());
        } else {
            FILE *f = nullptr;
            for (int i = 0; i < 600 && !(f = fopen(base.c_str(), "rb")); ++i) std::this_thread::sleep_for(std::chrono::milliseconds(100));
            if (!f || fread(&id, sizeof(id), 1, f) != 1) return 1;
            fclose(f);
        }
        CKN(ncclCommInitRank(&comm, world, id, rank));
        return 0;
    }
    int group_start() { if (!stub) CKN(ncclGroupStart()); return 0; }
    int group_end() { if (!stub) CKN(ncclGroupEnd()); return 0; }
    int send(const float *dev, size_t n, int peer, hipStream_t s) {
        if (!stub) { CKN(ncclSend(dev, n, ncclFloat, peer, comm, s)); return 0; }
        host.resize(n);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(host.data(), dev, n * sizeof(float), hipMemcpyDeviceToHost));
        const std::string p = base + "." + std::to_string(rank) + "." + std::to_string(peer);
        const int fd = pipe_fd(wfd, peer, p, O_WRONLY);
        if (fd < 0) return 1;
        unsigned long long hdr = n;
        bool ok = write(fd, &hdr, sizeof(hdr)) == (ssize_t)sizeof(hdr);
        for (size_t off = 0; ok && off < n * sizeof(float);) {
            const ssize_t w = write(fd, (const char *)host.data() + off, n * sizeof(float) - off);
            if (w <= 0) ok = false; else off += (size_t)w;
        }
        return ok ? 0 : 1;
    }
    int recv(float *dev, size_t n, int peer, hipStream_t s) {
        if (!stub) { CKN(ncclRecv(dev, n, ncclFloat, peer, comm, s)); return 0; }
        const std::string p = base + "." + std::to_string(peer) + "." + std::to_string(rank);
        const int fd = pipe_fd(rfd, peer, p, O_RDONLY);
        if (fd < 0) return 1;
        unsigned long long hdr = 0;
        bool ok = true;
        for (size_t off = 0; ok && off < sizeof(hdr);) {  // (a read on a pipe may return less than asked)
            const ssize_t r = read(fd, (char *)&hdr + off, sizeof(hdr) - off);
            if (r <= 0) ok = false; else off += (size_t)r;
        }
        if (ok && hdr != n) {
            fprintf(stderr, "stub transport: rank %d expects %zu floats from rank %d, which sends %llu\n", rank, n, peer, hdr);
            ok = false;
        }
        host.resize(n);
        for (size_t off = 0; ok && off < n * sizeof(float);) {
            const ssize_t r = read(fd, (char *)host.data() + off, n * sizeof(float) - off);
            if (r <= 0) ok = false; else off += (size_t)r;
        }
        if (!ok) return 1;
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(dev, host.data(), n * sizeof(float), hipMemcpyHostToDevice));
        return 0;
    }
    void destroy() {
        if (comm) ncclCommDestroy(comm);
        for (auto &kv : wfd) close(kv.second);
        for (auto &kv : rfd) close(kv.second);
    }
};

int main(int argc, char **argv) {
    if (argc < 6) {
        fprintf(stderr, "usage: %s <das|mvdr|lcmv|phase> <n_mics> <total_frames> <world> <rank|logical> [id_file] [chunks] [steps]\n", argv[0]);
        return 2;
    }
    const char *names[] = {"das", "mvdr", "lcmv", "gss", "phase"};
    int algo = -1;
    for (int i = 0; i < 5; ++i)
        if (!strcmp(argv[1], names[i])) algo = i;
    const int M = atoi(argv[2]);
    const long long F = atoll(argv[3]);
    const int world = atoi(argv[4]);
    const bool logical = !strcmp(argv[5], "logical");
    const int rank = logical ? 0 : atoi(argv[5]);
    const char *id_file = argc > 6 ? argv[6] : "/tmp/bf_shard_node.id";
    const int chunks = argc > 7 ? atoi(argv[7]) : 4;
    const int steps = argc > 8 ? atoi(argv[8]) : 3;
    bf_config cfg;
    if (algo < 0 || algo == BF_GSS || bf_config_init(&cfg, algo) != BF_OK || M < 1 || M > 16 || world < 1 || F < world) {
        fprintf(stderr, "bad arguments\n");
        return 2;
    }
    static const double ax[16] = {0.158, 0.158, -0.045, -0.050, -0.195, -0.057, 0.180, 0.158, 0.056, -0.050, -0.128, -0.195, -0.132, -0.057, 0.056, 0.158};
    static const double ay[16] = {0.115, -0.115, 0.000, -0.188, 0.000, 0.186, 0.000, -0.115, -0.171, -0.188, -0.098, 0.000, 0.098, 0.186, 0.171, 0.115};
    cfg.n_mics = M;  // beamform_config.yaml:20-35 ("aira16"), first M entries
    for (int m = 0; m < M; ++m) { cfg.mic_x[m] = ax[m]; cfg.mic_y[m] = ay[m]; }
    if (algo == BF_LCMV) {
        cfg.n_interf = 3;
        cfg.interf_angle[0] = -60.0; cfg.interf_angle[1] = 90.0; cfg.interf_angle[2] = 150.0;
    }
    const int halo = bf_shard_halo(&cfg);
    if (halo < 0) {
        fprintf(stderr, "this node recurses over frames: it shards by stream only\n");
        return 2;
    }
    int ndev = 0;
    CK(hipGetDeviceCount(&ndev));
    cfg.device = logical ? 0 : rank % ndev;  // (a one-GPU box puts every rank on device 0: RCCL refuses that, BF_SHARD_STUB=1 does not)
    CK(hipSetDevice(cfg.device));
    const int H = cfg.hop;
    hipStream_t cs, gs;  // compute / gather
    CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&gs, hipStreamNonBlocking));
    bf_handle *bf = nullptr;
    CKB(bf_create(&cfg, &bf));

    if (logical) {
        // unsharded reference on the same device, then every rank's slice in turn
        Rank whole;
        CKB(bf_shard_plan((size_t)F, 1, 0, halo, &whole.sh));
        if (make_slice(whole, M, H, cs)) return 1;
        CKB(bf_shard_run(bf, whole.x, &whole.sh, whole.y, cs));
        float *out = nullptr;
        CK(hipMalloc((void **)&out, (size_t)F * H * sizeof(float)));
        for (int r = 0; r < world; ++r) {
            Rank rk;
            CKB(bf_shard_plan((size_t)F, world, r, halo, &rk.sh));
            if (make_slice(rk, M, H, cs)) return 1;
            int rc = walk(bf, rk, M, H, chunks, cs, [&](int, long long own0, long long n) -> int {
                if (n == 0) return 0;
                CK(hipMemcpyAsync(out + (rk.sh.lo + own0) * H, rk.y + (rk.n_drop + own0) * H, (size_t)n * H * sizeof(float),
                                  hipMemcpyDeviceToDevice, cs));
                return 0;
            });
            if (rc) return rc;
            CK(hipStreamSynchronize(cs));
            CK(hipFree(rk.x));
            CK(hipFree(rk.y));
        }
        std::vector<float> a((size_t)F * H), b((size_t)F * H);
        CK(hipMemcpy(a.data(), whole.y, a.size() * sizeof(float), hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), out, b.size() * sizeof(float), hipMemcpyDeviceToHost));
        double num = 0, den = 0;
        long long bad = 0, differ = 0;
        for (size_t i = (size_t)(halo + 1) * H; i < a.size(); ++i) {  // mvdr / lcmv: the first P+1 frames are the reference's NaN frames
            if (!std::isfinite(a[i]) || !std::isfinite(b[i])) { bad += (std::isfinite(a[i]) != std::isfinite(b[i])); continue; }
            num += ((double)a[i] - b[i]) * ((double)a[i] - b[i]);
            den += (double)a[i] * a[i];
            differ += a[i] != b[i];
        }
        const double rel = std::sqrt(num / (den + 1e-300));
        printf("logical %d ranks x %d pieces, %s %d-mic %lld frames: rel L2 vs unsharded %.3e, %lld samples differ, %lld finite-mask mismatches\n",
               world, chunks, names[algo], M, F, rel, differ, bad);
        bf_destroy(bf);
        return (rel < 1e-5 && bad == 0) ? 0 : 1;
    }

    // ---- one process per rank, RCCL for the gather -----------------------------------------------------------------------
    const bool stub = getenv("BF_SHARD_STUB") && atoi(getenv("BF_SHARD_STUB")) != 0;
    // BF_SHARD_SELF=1 (RCCL only): rank 0 moves its own pieces through the communicator too -- with world = 1 the whole grouped
    // send / recv path of the gather runs on a one-GPU box (the execution check in front of the first real multi-GPU run)
    const bool self_loop = !stub && getenv("BF_SHARD_SELF") && atoi(getenv("BF_SHARD_SELF")) != 0;
    Transport tr;
    if (tr.init(world, rank, id_file, stub)) {
        fprintf(stderr, "rank %d: transport set-up failed\n", rank);
        return 1;
    }
    Rank me;
    CKB(bf_shard_plan((size_t)F, world, rank, halo, &me.sh));
    if (make_slice(me, M, H, cs)) return 1;
    float *out = nullptr;  // rank 0: the whole output, hops in stream order
    if (rank == 0) CK(hipMalloc((void **)&out, (size_t)F * H * sizeof(float)));
    std::vector<bf_shard> plan(world);
    for (int r = 0; r < world; ++r) CKB(bf_shard_plan((size_t)F, world, r, halo, &plan[r]));
    hipEvent_t piece_done, gather_done;
    CK(hipEventCreateWithFlags(&piece_done, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&gather_done, hipEventDisableTiming));
    CK(hipEventRecord(gather_done, gs));

    auto step = [&](bool gather) -> int {
        // the walk below rewrites me.y on the compute stream: not before the previous step's last transfer has read it
        if (gather) CK(hipStreamWaitEvent(cs, gather_done, 0));
        int rc = walk(bf, me, M, H, chunks, cs, [&](int c, long long own0, long long n) -> int {
            if (!gather) return 0;
            // piece c of every rank moves while piece c+1 is computed: the gather stream waits for this piece only
            CK(hipEventRecord(piece_done, cs));
            CK(hipStreamWaitEvent(gs, piece_done, 0));
            if (rank == 0 && n > 0 && !self_loop)
                CK(hipMemcpyAsync(out + own0 * H, me.y + (me.n_drop + own0) * H, (size_t)n * H * sizeof(float), hipMemcpyDeviceToDevice, gs));
            if (tr.group_start()) return 1;
            if (rank == 0 && n > 0 && self_loop) {  // BF_SHARD_SELF=1: rank 0's own piece through ncclSend + ncclRecv to itself, inside the group
                if (tr.send(me.y + (me.n_drop + own0) * H, (size_t)n * H, 0, gs)) return 1;
                if (tr.recv(out + own0 * H, (size_t)n * H, 0, gs)) return 1;
            }
            if (rank != 0) {
                if (n > 0 && tr.send(me.y + (me.n_drop + own0) * H, (size_t)n * H, 0, gs)) return 1;
            } else {
                for (int r = 1; r < world; ++r) {  // piece c of rank r, cut by rank r's own feed length (piece_owned): posted in round c
                    long long po0 = 0;               // whether or not rank 0 has a piece of its own in this round
                    const long long pn = piece_owned(bf_shard_n_feed(&plan[r]), bf_shard_n_drop(&plan[r]), chunks, c, &po0);
                    if (pn > 0 && tr.recv(out + (plan[r].lo + po0) * H, (size_t)pn * H, r, gs)) return 1;
                }
            }
            if (tr.group_end()) return 1;
            return 0;
        });
        if (rc) return rc;
        if (gather) CK(hipEventRecord(gather_done, gs));
        return 0;
    };
    auto timed = [&](bool gather, double *ms) -> int {
        if (step(gather)) return 1;  // warm-up (first gather also builds the channels)
        CK(hipStreamSynchronize(cs));
        CK(hipStreamSynchronize(gs));
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < steps; ++i)
            if (step(gather)) return 1;
        CK(hipStreamSynchronize(cs));
        CK(hipStreamSynchronize(gs));
        *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / steps;
        return 0;
    };
    double ms_compute = 0, ms_gather = 0;
    if (timed(false, &ms_compute) || timed(true, &ms_gather)) return 1;
    if (rank == 0) {
        std::vector<float> o((size_t)F * H);
        CK(hipMemcpy(o.data(), out, o.size() * sizeof(float), hipMemcpyDeviceToHost));
        double cs_ = 0;
        for (size_t i = (size_t)(halo + 1) * H; i < o.size(); ++i)
            if (std::isfinite(o[i])) cs_ += std::fabs((double)o[i]);
        printf("{\"world\": %d, \"algo\": \"%s\", \"mics\": %d, \"total_frames\": %lld, \"pieces\": %d, \"ms_per_step_compute\": %.4f, "
               "\"ms_per_step_with_overlapped_gather\": %.4f, \"frames_per_s_with_gather\": %.4e, \"checksum\": %.6f}\n",
               world, names[algo], M, F, chunks, ms_compute, ms_gather, (double)F / (ms_gather * 1e-3), cs_);
    }
    tr.destroy();
    bf_destroy(bf);
    return 0;
}
