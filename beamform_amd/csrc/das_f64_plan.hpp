// das_f64_plan.hpp -- the work queue of das_f64_pair_kernel (das_f64_w64.hip), host arithmetic: how a batch is cut into CHUNKS of
// consecutive frame pairs and where chunk k lies.  Plain C++ (the CPU suite checks it through tests/host_emul: every plan must tile every
// stream exactly once, start every chunk on an even frame, stay inside the table and inside the 10-bit fields of the kernel's work word).
//
// Per stream: nb = blocks per stream (n_cus / n_streams, at least 1).  The first level gives every block one long chunk (81 % of its equal
// share: consecutive pairs on one CU share their input hop through L1 / L2 and hand over their output hop through LDS flags), the following
// levels halve what is left, the last ones are chunks of 4 and 2 pairs.  A chunk edge costs one input hop read twice and two atomic adds per
// output sample, so the small chunks are kept to the last ~12 % of the batch.  Level-major order over all streams: the table starts with
// every block's long chunk and ends with the small ones that level the finishing times.
// env = BF_DAS_F64_SCHED: "0" = one level of equal chunks (the static runs of round 4); "88,16,8,4,2" = explicit chunk sizes in pairs.
#pragma once

#include <cstdlib>
#include <cstring>

#if defined(__HIPCC__)
#define BF_PLAN_HD __host__ __device__ inline
#else
#define BF_PLAN_HD inline
#endif

namespace bf {

constexpr int kChunkEnd = 0xFFFFF;       // chunk field of the kernel's work word: the table is exhausted
constexpr int kMaxChunkPairs = 1000;     // pairs per chunk (10 bits, and up to 8 draws past the end before the word is replaced)
constexpr int kSchedMaxChunks = 16384;   // rows of the chunk table

struct DasSchedPlan {
    int n_levels;
    int cnt[8], size[8];   // chunks per stream / pairs per chunk of each level
    int n_chunks;          // over all streams
    int grid;              // persistent blocks
};

// chunk k of the table: level-major, inside a level stream-major; the last chunk of a stream may be shorter (and end on a lone frame)
BF_PLAN_HD void das_f64_chunk(const DasSchedPlan &p, long n_frames, int n_streams, int k, int *stream, long *t0, long *n) {
    int lvl = 0, kk = k;
    long pair0 = 0;  // pairs of a stream in front of level lvl
    while (lvl < p.n_levels - 1 && kk >= p.cnt[lvl] * n_streams) {
        kk -= p.cnt[lvl] * n_streams;
        pair0 += (long)p.cnt[lvl] * p.size[lvl];
        ++lvl;
    }
    const int s = kk / p.cnt[lvl], j = kk - s * p.cnt[lvl];
    *stream = s;
    *t0 = 2 * (pair0 + (long)j * p.size[lvl]);
    long len = 2L * p.size[lvl];
    if (*t0 + len > n_frames) len = n_frames - *t0;
    *n = len;
}

inline DasSchedPlan das_f64_plan(long n_frames, int n_streams, int n_cus, const char *env) {
    DasSchedPlan p{};
    const long pairs = (n_frames + 1) / 2;
    long nb = (long)n_cus / n_streams;
    if (nb < 1) nb = 1;
    int sizes[8], n_sizes = 0;
    const long share = (pairs + nb - 1) / nb;  // pairs per block if they were dealt out evenly
    if (env && strchr(env, ',')) {
        const char *c = env;
        while (*c && n_sizes < 8) {
            sizes[n_sizes++] = atoi(c);
            c = strchr(c, ',');
            if (!c) break;
            ++c;
        }
    } else if (!(env && atoi(env) == 0) && share >= 48) {
        long front = (share * 13 / 16) & ~7L;             // 81 % of the share in long chunks, a multiple of 8 pairs
        long rest = share - front;
        while (front > kMaxChunkPairs && n_sizes < 3) {    // (a chunk holds at most kMaxChunkPairs pairs: very long batches get several)
            sizes[n_sizes++] = kMaxChunkPairs;
            front -= kMaxChunkPairs;
        }
        if (front > kMaxChunkPairs) { rest += front - kMaxChunkPairs; front = kMaxChunkPairs; }
        sizes[n_sizes++] = (int)front;
        while (n_sizes < 6 && rest >= 24) {                // the rest in chunks that halve it level by level: 8 at the headline size
            long sz = (rest / 2) & ~7L;
            if (sz > kMaxChunkPairs) sz = kMaxChunkPairs;
            sizes[n_sizes++] = (int)sz;
            rest -= sz;
        }
        if (rest >= 12) { sizes[n_sizes++] = 4; rest -= 4; }
        sizes[n_sizes++] = 2;
    }
    if (n_sizes == 0) {  // equal chunks
        long sz = ((share + 7) / 8) * 8;
        if (sz > kMaxChunkPairs) sz = kMaxChunkPairs & ~7;
        sizes[n_sizes++] = (int)sz;
    }
    long left = pairs;
    for (int i = 0; i < n_sizes && left > 0; ++i) {
        int sz = sizes[i] < 1 ? 1 : (sizes[i] > kMaxChunkPairs ? kMaxChunkPairs : sizes[i]);
        long cnt = (i == n_sizes - 1) ? (left + sz - 1) / sz : nb;
        if (cnt * sz > left) cnt = (left + sz - 1) / sz;
        p.size[p.n_levels] = sz;
        p.cnt[p.n_levels] = (int)cnt;
        ++p.n_levels;
        left -= cnt * sz;
    }
    long total = 0;
    for (int i = 0; i < p.n_levels; ++i) total += (long)p.cnt[i] * n_streams;
    if (total > kSchedMaxChunks || total >= kChunkEnd) {  // too fine for the table: one level of the largest chunks that fit
        long sz = kMaxChunkPairs & ~7;
        p.n_levels = 1;
        p.size[0] = (int)sz;
        p.cnt[0] = (int)((pairs + sz - 1) / sz);
        total = (long)p.cnt[0] * n_streams;
    }
    p.n_chunks = (int)total;
    p.grid = (int)(total < n_cus ? total : n_cus);
    return p;
}


}  // namespace bf
