// cov_kernels.hip -- mvdr / lcmv: sliding per-bin covariance + Cholesky-based constrained solve, four mappings of a problem
// onto lanes (thread-per-problem, row-per-lane over LDS, cyclic rows over DPP quads, row-per-lane over DPP rows).
#include "launch_trace.hpp"
#include "bins_common.hpp"

#include <cmath>
#include <vector>

namespace bf {
namespace BF_NTAG {

namespace {

// ======================================================================================
//                     mvdr / lcmv: covariance, Cholesky solve, constraints
// ======================================================================================
// One group of MP lanes per (stream, bin) problem, lane i <-> microphone i; the group walks a
// tile of consecutive frames so the sample covariance of the previous P frames
//   R = past_ffts[j] * past_ffts[j]^H                     (mvdr.cpp:87, lcmv.cpp:112)
// is slid by one rank-1 update and one downdate per frame (recomputed from scratch at the tile
// start).  The reference inverts R o whiteR with PartialPivLU and forms
//   mvdr:  w = R^-1 a / (a^H R^-1 a),   y = w^H x                     (mvdr.cpp:88-94)
//   lcmv:  W = R^-1 C (C^H R^-1 C)^-1,  y = W(:,0)^H x                (lcmv.cpp:113-119)
// R o whiteR is Hermitian positive definite whenever every mic has history energy, so with
// R = L L^H, U = L^-1 [C | x]:   G = U_C^H U_C,  g = U_C^H u_x,  y = (G^-1 g)_0
// (mvdr is the KP1 = 1 case: y = u_a^H u_x / u_a^H u_a).  Lane i owns row i of the
// factorisation; columns are exchanged through LDS (one wavefront executes its LDS operations
// in order, so only compiler barriers separate the phases).  A zero covariance (frame 0 of a
// cold start) yields 0 * inf = NaN, the same NaN frame the reference emits.
// ---- mvdr fast path: one thread per (stream, tile, in-band problem), whole problem in registers -------------
// For M <= 8 the lower triangle of R (36 complex) and of its working copy fit the 512-entry register file of a
// wavefront that has a SIMD to itself (fp64 FMA issues every 4 cycles, so one wavefront per SIMD already keeps the
// fp64 pipe busy).  No LDS exchange, no idle lanes.  Same maths as mvdr_lcmv_kernel with KP1 = 1:
//   R o whiteR = L L^H,  u = L^-1 a,  v = L^-1 x,  y = u^H v / u^H u.
// Lanes enumerate (time tile, problem) pairs of ONE output stream, problem fastest, over the problems that need a solve
// only: k = 0 is problem 0 (y = X_0, mvdr.cpp:76), then the in-band problems (FastPlan).  Out-of-band problems (34 % of
// the rows at the launch-file band) occupy no lane at all, and the host picks the tile length so that the wavefronts
// fill the chip's 4 x CUs slots a whole number of times (65 536 frames, 340 problems: 2 040 wavefronts of 171 frames
// = two rounds, where one wavefront per (128-frame tile, 64 consecutive problems) took three).
// MP = 2 x microphone pairs (2, 4, 6, 8): an odd count's last pair has a zero partner channel, whose row is pinned to the
// identity (its spectrum is the transform's ~1e-16 rounding residue, not an exact zero).
struct FastPlan {
    int q_lo, n_main;  // problems q_lo .. q_lo + n_main - 1 are in band (a contiguous run inside 1 .. N/2-1)
    int n_extra;       // the irregular problems of quirk Q1 that are in band: N/2 (f := 0) when 0 Hz is, N/2+1 (|f| = (N/2-1) sr/N)
    int extra_q[2];
    int solve0;        // lcmv with 0 Hz in band: problem 0 is an ordinary problem (mvdr passes X_0 through, mvdr.cpp:76)
    int nb;            // 1 + n_main + n_extra
    int tile, tiles, waves_per_stream;
};

// one row of 64 stored elements, global -> LDS (lane l's 12 or 16 bytes land at row + 16 l)
template <bool Z128>
__device__ __forceinline__ void lds_dma(const char *g, void *l) {
    if constexpr (Z128)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 0, 0);
    else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 12, 0, 0);
}

// KC = constraint columns: 1 = mvdr (the steering vector); 2..4 = lcmv with up to KC - 1 interferers (lcmv.cpp:102-130): the
// columns C ride through the factorisation like the steering vector does (U = L^-1 C, v = L^-1 x), then G = U^H U, g = U^H v and
// y = (G^-1 g)_0 -- the first row of W^H x with W = R^-1 C (C^H R^-1 C)^-1.  Unused columns are padded with the identity.
// Z128: the spectra are complex doubles (the default, BF_PRECISION_REFERENCE: 16 bytes per element) instead of z48 (BF_PRECISION_MIXED: 12 bytes); both stored halved
template <int MP, int KC, bool Z128>
__global__ __launch_bounds__(64, 1) void mvdr_fast_kernel(BinsArgs a, FastPlan fp) {
    constexpr int NT = MP * (MP + 1) / 2;
    constexpr long kZB = Z128 ? (long)sizeof(f64x2) : (long)sizeof(z48);  // bytes per stored element
    const int lane = threadIdx.x;
    const int s = blockIdx.x / fp.waves_per_stream;  // output stream: uniform per wavefront
    const int id0 = (blockIdx.x - s * fp.waves_per_stream) * 64;
    int id = id0 + lane;
    const bool live = id < fp.tiles * fp.nb;
    if (!live) id = fp.tiles * fp.nb - 1;
    const int tl0 = id0 / fp.nb, tl = id / fp.nb;  // time tile of lane 0 / of this lane
    const int k = id - tl * fp.nb;
    const int q = k == 0 ? 0 : k <= fp.n_main ? fp.q_lo + k - 1 : fp.extra_q[k - 1 - fp.n_main];
    const bool solve_q = q != 0 || fp.solve0 != 0;
    const int j = q_bin(q);
    const long tA0 = (long)tl0 * fp.tile;          // first frame of lane 0's tile: the wavefront's time origin (uniform)
    const int dt = (tl - tl0) * fp.tile;           // this lane's tile starts dt frames later
    long n_it = a.n_frames - tA0;                  // frames the wavefront walks (lane 0 has the longest tile)
    if (n_it > fp.tile) n_it = fp.tile;
    long cnt = a.n_frames - (tA0 + dt);            // frames this lane owns
    if (cnt > fp.tile) cnt = fp.tile;
    if (!live) cnt = 0;
    const int M = a.n_mics, NP = (M + 1) >> 1, P = a.cfg.past_windows;
    // frame tA0 of this stream's packed spectra (uniform) + a 32-bit per-lane byte offset: SGPR base + VGPR offset addressing
    const char *Zu = reinterpret_cast<const char *>(a.Z) + ((long)(s / a.n_dirs) * a.frames_ws + a.frame_off + tA0) * NP * kN * kZB;
    const int ksrc = q_src_bin(q), kneg = (kN - ksrc) & (kN - 1);
    const unsigned vk = (unsigned)(((long)dt * NP * kN + ksrc) * kZB);
    const unsigned vn = (unsigned)(((long)dt * NP * kN + kneg) * kZB);
    const long frame_bytes = (long)NP * kN * kZB;
    const f64x2 *steer = a.steer + (long)(s % a.n_dirs) * a.steer_dir_stride + j;
    const long yidx = ((long)s * a.n_frames + tA0 + dt) * kYhStride + q;

    // Spectra are prefetched one frame ahead by global->LDS DMA (global_load_lds_dwordx3: no VGPRs; lane l's 12 bytes
    // land at row base + 16 l -- measured, tools/ubench/lds_dma_x3.hip -- so a row is 64 slots of 16 bytes): a wavefront
    // that owns its SIMD has nobody to hide HBM latency behind (PMC of the register-only version: 57 % of its cycles in
    // s_waitcnt).  Row 2p / 2p+1 = Z_t[p][k] / Z_t[p][N-k]; rows MP + 2p, MP + 2p + 1 = the same of frame t - P.
    struct alignas(16) z48slot {
        z48 v;
        unsigned pad;
    };
    __shared__ z48slot s_pf[2][2 * MP][64];
    auto dma_frame = [&](long i, bool with_old, int buf) {  // frame tA0 + i (+ dt per lane) into s_pf[buf]
        const char *bn = Zu + i * frame_bytes, *bo = Zu + (i - P) * frame_bytes;
#pragma unroll
        for (int p = 0; p < MP / 2; ++p) {
                const long po = (long)p * kN * kZB;
                lds_dma<Z128>(bn + po + vk, &s_pf[buf][2 * p][0]);
                lds_dma<Z128>(bn + po + vn, &s_pf[buf][2 * p + 1][0]);
                if (with_old) {
                    lds_dma<Z128>(bo + po + vk, &s_pf[buf][MP + 2 * p][0]);
                    lds_dma<Z128>(bo + po + vn, &s_pf[buf][MP + 2 * p + 1][0]);
                }
            }
    };
    auto unpack = [&](int buf, int base, cd (&X)[MP]) {  // microphone spectra out of the prefetched rows (z48 is stored halved)
#pragma unroll
        for (int p = 0; p < MP / 2; ++p) {
            cd z, c;  // Z[k] / 2 and Z[N-k] / 2 (stored halved: StftArgs::halve; the latter conjugated on the fly)
            if constexpr (Z128) {
                z = ld(reinterpret_cast<const f64x2 *>(&s_pf[buf][base + 2 * p][lane]));
                c = ld(reinterpret_cast<const f64x2 *>(&s_pf[buf][base + 2 * p + 1][lane]));
            } else {
                z = dec48(s_pf[buf][base + 2 * p][lane].v);
                c = dec48(s_pf[buf][base + 2 * p + 1][lane].v);
            }
            X[2 * p] = cd{z.x + c.x, z.y - c.y};      // (Z[k] + conj Z[N-k]) / 2
            X[2 * p + 1] = cd{z.y + c.y, c.x - z.x};  // (Z[k] - conj Z[N-k]) / (2i)
        }
        if (q == kQX) {  // the extra problem: bin N/2+1 holds the conjugate of bin N/2-1's spectrum
#pragma unroll
            for (int m = 0; m < MP; ++m) X[m].y = -X[m].y;
        }
    };
#define BF_DMA_WAIT() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_waitcnt(0x0F70); asm volatile("" ::: "memory"); } while (0)  /* vmcnt(0): the builtin, so that hipcc's own counter model sees the drain */
    // mvdr (KC == 1), round 6: the same 32 KB seen as FOUR half-buffers of MP rows -- a ring of three for the NEWEST frames, requested TWO frames
    // ahead, and one for the frame that leaves the window, requested one frame ahead (it used to travel two ahead and the newest one: the rows
    // fresh from the STFT kernel are the ones that come from HBM).  With the steering vector in registers the loop has no load that returns
    // to a register, so hipcc places no vector-memory wait of its own and the counted s_waitcnt vmcnt(MP) at the loop's top is the only one.
    constexpr bool DEEP = KC == 1;
    auto hrow = [&](int h, int row) { return &s_pf[h >> 1][(h & 1) * MP + row][0]; };
    auto dma_rows = [&](const char *b, int h) {  // the MP rows of the frame at byte base b (+ dt per lane) into half-buffer h
#pragma unroll
        for (int p = 0; p < MP / 2; ++p) {
            const long po = (long)p * kN * kZB;
            lds_dma<Z128>(b + po + vk, hrow(h, 2 * p));
            lds_dma<Z128>(b + po + vn, hrow(h, 2 * p + 1));
        }
    };
    auto unpack_h = [&](int h, cd (&X)[MP]) { unpack(h >> 1, (h & 1) * MP, X); };

    cd R[NT];       // strict lower triangle, row-major: R[i*(i+1)/2 + c], c < i (the diagonal slots are unused)
    double Rd[MP];  // the diagonal is real: kept and updated as such (two FMAs per rank-1 term instead of four)
#pragma unroll
    for (int e = 0; e < NT; ++e) R[e] = cd{0, 0};
#pragma unroll
    for (int i = 0; i < MP; ++i) Rd[i] = 0.0;
    int pb = 0;  // buffer the next consumer reads
    cd Uk[MP];   // mvdr: the steering vector stays in registers for the whole tile (the factorisation scales its copy in place)
#pragma unroll
    for (int m = 0; m < MP; ++m) Uk[m] = (DEEP && m < M) ? ld(steer + (long)m * kN) : cd{0, 0};
    if constexpr (DEEP) {  // the tile's first two frames into ring slots 0 and 1; the warm-up below walks half-buffers 2 and 3
        dma_rows(Zu, 0);
        dma_rows(Zu + (n_it > 1 ? 1 : 0) * frame_bytes, 1);
        dma_rows(Zu - frame_bytes, 2);
    } else {
        dma_frame(-1, false, pb);
    }
    for (int p = 1; p <= P; ++p) {  // covariance of the P frames in front of the tile
        BF_DMA_WAIT();
        __builtin_amdgcn_wave_barrier();
        if constexpr (DEEP) {
            if (p < P) dma_rows(Zu - (long)(p + 1) * frame_bytes, 2 + (p & 1));
            cd X[MP];
            unpack_h(2 + ((p - 1) & 1), X);
#pragma unroll
            for (int i = 0; i < MP; ++i) {
#pragma unroll
                for (int c = 0; c < i; ++c) R[i * (i + 1) / 2 + c] = cfma_conj(R[i * (i + 1) / 2 + c], X[i], X[c]);
                Rd[i] = fma(X[i].y, X[i].y, fma(X[i].x, X[i].x, Rd[i]));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the half-buffer is rewritten two steps on)
            continue;
        }
        if (p < P)
            dma_frame(-p - 1, false, pb ^ 1);
        else
            dma_frame(0, true, pb ^ 1);  // first frame of the tile
        cd X[MP];
        unpack(pb, 0, X);
#pragma unroll
        for (int i = 0; i < MP; ++i) {
#pragma unroll
            for (int c = 0; c < i; ++c) R[i * (i + 1) / 2 + c] = cfma_conj(R[i * (i + 1) / 2 + c], X[i], X[c]);
            Rd[i] = fma(X[i].y, X[i].y, fma(X[i].x, X[i].x, Rd[i]));
        }
        pb ^= 1;
    }
    const float thr32 = (float)(a.cfg.freq_mag_threshold * (double)((unsigned)M * (unsigned)kN));
    cd y_prev{0, 0};  // frame it - 1's output: stored one iteration late, behind the next frame's DMA requests, so that no s_waitcnt vmcnt(0)
                      // of an iteration waits for a store issued a few instructions earlier
    if constexpr (DEEP) BF_DMA_WAIT();  // (the steering vector and the ring's first two frames are there whatever P is: hipcc's model enters the loop with nothing pending)
    int sn = 0;  // DEEP: ring slot of frame it (it mod 3)
    for (long it = 0; it < n_it; ++it) {
        if constexpr (DEEP) {
            // frame it (requested two iterations ago) and frame it - 1 - P (one ago) have landed, frame it - 2's output has been stored; the MP
            // requests of frame it + 1 may still travel (vector loads return in order: at most MP outstanding = only those)
            if constexpr (MP == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (MP == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if constexpr (MP == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            BF_DMA_WAIT();  // frame it (and it - P) have landed in s_pf[pb]; frame it - 2's output has been stored
        }
        __builtin_amdgcn_wave_barrier();
        const int sp = sn == 0 ? 2 : sn - 1;  // DEEP: slot of frame it - 1 = the slot frame it + 2 goes into
        // The constraint columns: re-read per frame (L2-resident, consecutive lanes = consecutive bins) instead of living in registers
        // that the factorisation needs.  Requested HERE, in front of the slide that covers their L2 round trip, and drained (vmcnt(0),
        // the builtin: hipcc's counter model sees it) in front of the next frame's DMA: hipcc prices any vector-memory wait with an
        // LDS-DMA request in flight at vmcnt(0), so until round 6 -- DMA first, then these loads -- the factorisation's first use of U
        // waited for the rows requested a few hundred instructions earlier (SQ_WAIT_ANY 17 % of the wave cycles).
        cd U[KC][MP];
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
            for (int m = 0; m < MP; ++m) {
                if constexpr (DEEP) U[c][m] = Uk[m];
                else U[c][m] = (m < M && c < a.kp1) ? ld(steer + ((long)c * M + m) * kN) : cd{0, 0};
            }
        cd(&ua)[MP] = U[0];
        __builtin_amdgcn_sched_barrier(0);
        if (it > 0) {
            // slide the covariance window over the PREVIOUS frame (mvdr.cpp:100-101), whose rows still sit in the other buffer:
            // done here, not behind the solve, so that the updated R is at once the factorisation's working copy -- R itself
            // then rests (in the accumulator half of the register file) while the solve has the 256 arithmetic registers
            cd Xn[MP], Xo[MP];
#pragma unroll
            for (int m = 0; m < MP; ++m)  // parked by the previous iteration
                Xn[m] = ld(reinterpret_cast<const f64x2 *>(DEEP ? reinterpret_cast<z48slot *>(hrow(sp, m)) + lane : &s_pf[pb ^ 1][m][lane]));
            if constexpr (DEEP) unpack_h(3, Xo);
            else unpack(pb ^ 1, MP, Xo);
#pragma unroll
            for (int i = 0; i < MP; ++i) {
#pragma unroll
                for (int c = 0; c < i; ++c)
                    R[i * (i + 1) / 2 + c] = cfms_conj(cfma_conj(R[i * (i + 1) / 2 + c], Xn[i], Xn[c]), Xo[i], Xo[c]);
                Rd[i] = fma(-Xo[i].y, Xo[i].y, fma(-Xo[i].x, Xo[i].x, fma(Xn[i].y, Xn[i].y, fma(Xn[i].x, Xn[i].x, Rd[i]))));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the rows have been read: the next DMA may overwrite them
        }
        // the next frame's rows: requested on every path (the last iteration re-requests its own frame into the idle buffer), with
        // nothing older in flight; the previous frame's output goes out behind them (the wait at the top of the loop covers it)
        if constexpr (DEEP) {
            // the frame that leaves the window at the NEXT slide, the previous frame's output, then frame it + 2 into the slot the slide has just
            // emptied -- in this order: the next top-of-loop wait lets exactly the youngest MP requests stand
            __builtin_amdgcn_sched_barrier(0);
            dma_rows(Zu + (it - P) * frame_bytes, 3);
            if (it > 0 && it - 1 < cnt) st_y(a, yidx + (it - 1) * kYhStride, q, y_prev);
            dma_rows(Zu + (it + 2 < n_it ? it + 2 : n_it - 1) * frame_bytes, sp);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            BF_DMA_WAIT();
            __builtin_amdgcn_sched_barrier(0);
            dma_frame(it + 1 < n_it ? it + 1 : it, true, pb ^ 1);
            if (it > 0 && it - 1 < cnt) st_y(a, yidx + (it - 1) * kYhStride, q, y_prev);
            __builtin_amdgcn_sched_barrier(0);
        }
        cd X[MP];
        if constexpr (DEEP) unpack_h(sn, X);
        else unpack(pb, 0, X);
        // park the unpacked spectra in the slots their packed form came from (a 16-byte slot holds one complex double): the
        // next iteration's slide reads them back with one LDS read each instead of unpacking the frame again
#pragma unroll
        for (int m = 0; m < MP; ++m)
            *reinterpret_cast<f64x2 *>(DEEP ? reinterpret_cast<z48slot *>(hrow(sn, m)) + lane : &s_pf[pb][m][lane]) = f64x2{X[m].x, X[m].y};
        // magnitude gate (mvdr.cpp:85,95): sum |X_m| / (M N) > threshold.  Decided in fp32 unless the fp32 sum is within 1e-4
        // of the threshold (its own error is < 1e-6): then, for that wavefront and frame, in the reference's double arithmetic.
        float m32 = 0.f;  // (a zero partner channel of an odd microphone count adds its ~1e-16 |X| rounding residue: irrelevant here)
#pragma unroll
        for (int m = 0; m < MP; ++m) m32 += __builtin_amdgcn_sqrtf((float)norm2(X[m]));
        bool open = m32 > thr32;
        if (__builtin_amdgcn_ballot_w64(__builtin_fabsf(m32 - thr32) <= 1e-4f * thr32) != 0) {
            double mag = 0.0;
#pragma unroll
            for (int m = 0; m < MP; ++m)
                if (m < M) mag += fast_sqrt(norm2(X[m]));  // |X| well inside double range: no hypot scaling needed
            mag /= (double)((unsigned)M * (unsigned)kN);
            open = mag > a.cfg.freq_mag_threshold;
        }
        cd y = X[0] * 0.01;  // gate closed: mvdr.cpp:96
        if (q == 0 && !fp.solve0) y = a.cfg.algo == BF_LCMV ? cd{0, 0} : X[0];  // mvdr.cpp:76; lcmv has no such rule: problem 0 out of band reads as zero
        if (__builtin_amdgcn_ballot_w64(open && solve_q) != 0) {
            cd A[NT];
#pragma unroll
            for (int i = 0; i < MP; ++i) {
#pragma unroll
                for (int c = 0; c < i; ++c) A[i * (i + 1) / 2 + c] = R[i * (i + 1) / 2 + c];
                A[i * (i + 1) / 2 + i] = cd{(i < M) ? Rd[i] * 1.001 : 1.0, 0.0};  // whiteR diagonal (mvdr.cpp:239-243); padding = identity
            }
#pragma unroll
            for (int jj = 0; jj < MP; ++jj) {
                const double inv = fast_rsqrt1(A[jj * (jj + 1) / 2 + jj].x);  // 1/L_jj; L_jj itself is never needed
#pragma unroll
                for (int c = 0; c < KC; ++c) U[c][jj] = U[c][jj] * inv;
                X[jj] = X[jj] * inv;  // X turns into v = L^-1 x in place
#pragma unroll
                for (int i = jj + 1; i < MP; ++i) {
                    const cd Lij = A[i * (i + 1) / 2 + jj] * inv;
                    A[i * (i + 1) / 2 + jj] = Lij;
#pragma unroll
                    for (int c = 0; c < KC; ++c) U[c][i] = cfms(U[c][i], Lij, U[c][jj]);
                    X[i] = cfms(X[i], Lij, X[jj]);
                }
#pragma unroll
                for (int c = jj + 1; c < MP; ++c) {
                    const cd Lc = A[c * (c + 1) / 2 + jj];
#pragma unroll
                    for (int i = c; i < MP; ++i) {
                        if (i == c)  // diagonal: only the real part is ever read
                            A[i * (i + 1) / 2 + c].x = fma(-Lc.y, Lc.y, fma(-Lc.x, Lc.x, A[i * (i + 1) / 2 + c].x));
                        else
                            A[i * (i + 1) / 2 + c] = cfms_conj(A[i * (i + 1) / 2 + c], A[i * (i + 1) / 2 + jj], Lc);
                    }
                }
            }
            if constexpr (KC == 1) {
                cd num{0, 0};
                double den = 0.0;
#pragma unroll
                for (int i = 0; i < MP; ++i) {
                    num = cfma_conj(num, X[i], ua[i]);
                    den += norm2(ua[i]);
                }
                const double rden = fast_rcp(den);
                if (open && solve_q) y = cd{num.x * rden, num.y * rden};
            } else {
                // G (Hermitian, upper triangle row-major) and g, then the KC x KC system by elimination (as cov2d_kernel)
                auto UI = [](int r, int c) { return r * KC - r * (r - 1) / 2 + (c - r); };
                cd ge[KC * (KC + 1) / 2], gv[KC];
#pragma unroll
                for (int r1 = 0; r1 < KC; ++r1) {
#pragma unroll
                    for (int r2 = r1; r2 < KC; ++r2) {
                        cd acc{0, 0};
#pragma unroll
                        for (int i = 0; i < MP; ++i) acc = cfma_conj(acc, U[r2][i], U[r1][i]);
                        ge[UI(r1, r2)] = acc;
                    }
                    cd acc{0, 0};
#pragma unroll
                    for (int i = 0; i < MP; ++i) acc = cfma_conj(acc, X[i], U[r1][i]);
                    gv[r1] = acc;
                }
#pragma unroll
                for (int r1 = 0; r1 < KC; ++r1)
                    if (r1 >= a.kp1) {
#pragma unroll
                        for (int r2 = 0; r2 < r1; ++r2) ge[UI(r2, r1)] = cd{0, 0};
                        ge[UI(r1, r1)] = cd{1.0, 0.0};
#pragma unroll
                        for (int c = r1 + 1; c < KC; ++c) ge[UI(r1, c)] = cd{0, 0};
                        gv[r1] = cd{0, 0};
                    }
                cd pinvs[KC];
#pragma unroll
                for (int k = 0; k < KC; ++k) {
                    const cd pinv = crcp(ge[UI(k, k)]);
                    pinvs[k] = pinv;
#pragma unroll
                    for (int r1 = k + 1; r1 < KC; ++r1) {
                        const cd fct = conj(ge[UI(k, r1)]) * pinv;
#pragma unroll
                        for (int c = r1; c < KC; ++c) ge[UI(r1, c)] = ge[UI(r1, c)] - fct * ge[UI(k, c)];
                        gv[r1] = gv[r1] - fct * gv[k];
                    }
                }
#pragma unroll
                for (int k = KC - 1; k >= 0; --k) {
                    cd acc = gv[k];
#pragma unroll
                    for (int c = k + 1; c < KC; ++c) acc = acc - ge[UI(k, c)] * gv[c];
                    gv[k] = acc * pinvs[k];
                }
                if (open && solve_q) y = gv[0];
            }
        }
        y_prev = y;
        pb ^= 1;
        sn = sn == 2 ? 0 : sn + 1;
    }
    BF_DMA_WAIT();  // (the last iterations' idle requests)
    if (n_it > 0 && n_it - 1 < cnt) st_y(a, yidx + (n_it - 1) * kYhStride, q, y_prev);
#undef BF_DMA_WAIT
}

template <int KM>
struct GramIdx {  // entries of the Hermitian upper triangle of G followed by g
    static constexpr int NG = KM * (KM + 1) / 2;
    static constexpr int NE = NG + KM;
};

template <int MP, int KM>
__global__ __launch_bounds__(256) void mvdr_lcmv_kernel(BinsArgs a, int tile, int tiles_per_stream) {
    constexpr int GPB = 256 / MP;          // problem groups per block
    constexpr int NB = KM + 1;             // right-hand sides: constraints + current frame
    constexpr int NE = GramIdx<KM>::NE;
    // +1 element of padding per row: the groups of a wavefront read the same [row][k] at the same time (broadcast
    // inside a group), and with a group stride that is a multiple of 128 B all of them would hit the same 4 banks
    // (PMC before the padding: SQ_LDS_BANK_CONFLICT = 1.7x SQ_ACTIVE_INST_LDS; lcmv 16-mic 37.5 -> 34.0 ms)
    __shared__ cd s_col[GPB][MP + 1];
    __shared__ cd s_x[GPB][MP + 1];
    __shared__ cd s_xo[GPB][MP + 1];
    __shared__ cd s_u[GPB][NB][MP + 1];
    __shared__ cd s_e[GPB][NE + 1];

    const int grp = threadIdx.x / MP;
    const int i = threadIdx.x % MP;
    const int q = blockIdx.y * GPB + grp;
    if (q >= kNQ) return;
    const int s = blockIdx.x / tiles_per_stream;
    const long tA = (long)(blockIdx.x % tiles_per_stream) * tile;
    long tB = tA + tile;
    if (tB > a.n_frames) tB = a.n_frames;
    const int M = a.n_mics, NP = (M + 1) >> 1, KP1 = a.kp1, P = a.cfg.past_windows;
    const int j = q_bin(q);
    const bool lcmv = a.cfg.algo == BF_LCMV;
    const long yidx = ((long)s * a.n_frames) * kYhStride + q;
    const long z0 = ((long)(s / a.n_dirs) * a.frames_ws + a.frame_off) * NP * kN;  // element index of frame 0 of this batch
    const f64x2 *steer = a.steer + (long)(s % a.n_dirs) * a.steer_dir_stride;

    // one microphone's spectrum at this problem's bin, frame t (may be negative: history); z48 elements (stored halved) or,
    // (the default) full doubles; stored halved either way
    const int ksrc = q_src_bin(q), kneg = (kN - ksrc) & (kN - 1);
    auto ldz = [&](long e) -> cd { return a.z48 ? ld(reinterpret_cast<const z48 *>(a.Z) + e) : ld(a.Z + e); };  // (either way stored halved)
    auto load_xi = [&](long t) -> cd {
        if (i >= M) return cd{0, 0};
        const long ef = z0 + t * NP * kN + (long)(i >> 1) * kN;
        const cd z = ldz(ef + ksrc), zc = conj(ldz(ef + kneg));
        cd x;
        if ((i & 1) == 0) {
            x = z + zc;  // z48 spectra are stored halved
        } else {
            const cd d = z - zc;
            x = cd{d.y, -d.x};
        }
        return q == kQX ? conj(x) : x;
    };

    const double f = fabs(a.freqs[j]);
    const bool inband = f >= a.cfg.freq_min && f <= a.cfg.freq_max;
    if (!inband || (!lcmv && j == 0)) {
        // mvdr.cpp:76 y_fft[0] = in_fft(0,0); out of band: y_fft[j] = 0 (mvdr.cpp:103)
        for (long t = tA; t < tB; ++t) {
            cd y{0, 0};
            if (!lcmv && j == 0) {
                const cd x = load_xi(t);
                s_x[grp][i] = x;
                __builtin_amdgcn_wave_barrier();
                y = s_x[grp][0];
                __builtin_amdgcn_wave_barrier();
            }
            if (i == 0) st_y(a, yidx + t * kYhStride, q, y);
        }
        return;
    }

    cd cst[KM];  // this mic's entries of the constraint columns (weights[j](i, r))
#pragma unroll
    for (int r = 0; r < KM; ++r)
        cst[r] = (r < KP1 && i < M) ? ld(steer + ((long)r * M + i) * kN + j) : cd{0, 0};

    // R row i (lower triangle c <= i is what the factorisation reads)
    cd R[MP];
#pragma unroll
    for (int c = 0; c < MP; ++c) R[c] = cd{0, 0};
    for (int p = 1; p <= P; ++p) {
        const cd x = load_xi(tA - p);
        s_x[grp][i] = x;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < MP; ++c) R[c] = cfma_conj(R[c], x, s_x[grp][c]);
        __builtin_amdgcn_wave_barrier();
    }

    for (long t = tA; t < tB; ++t) {
        const cd x = load_xi(t);
        const cd xo = load_xi(t - P);
        s_x[grp][i] = x;
        s_xo[grp][i] = xo;
        __builtin_amdgcn_wave_barrier();
        double mag = 0.0;
        for (int m = 0; m < M; ++m) mag += fast_sqrt(norm2(s_x[grp][m]));
        mag /= (double)((unsigned)M * (unsigned)kN);
        cd y;
        if (mag > a.cfg.freq_mag_threshold) {
            cd A[MP], b[NB];
#pragma unroll
            for (int c = 0; c < MP; ++c) A[c] = R[c];
            if (i < M) {
                // cwiseProduct(whiteR): diagonal * 1.001 (mvdr.cpp:239-243)
#pragma unroll
                for (int c = 0; c < MP; ++c)
                    if (c == i) A[c] = A[c] * 1.001;
            } else {
#pragma unroll
                for (int c = 0; c < MP; ++c) A[c] = cd{c == i ? 1.0 : 0.0, 0.0};
            }
#pragma unroll
            for (int r = 0; r < KM; ++r) b[r] = cst[r];
            b[KM] = x;
#pragma unroll
            for (int jj = 0; jj < MP; ++jj) {
                s_col[grp][i] = A[jj];  // raw column jj, row i
                if (i == jj) {
#pragma unroll
                    for (int r = 0; r < NB; ++r) s_u[grp][r][0] = b[r];
                }
                __builtin_amdgcn_wave_barrier();
                const double inv = fast_rsqrt1(s_col[grp][jj].x);  // 1 / L_jj
                const cd Lij = A[jj] * inv;
                if (i > jj) {
#pragma unroll
                    for (int c = jj + 1; c < MP; ++c)
                        if (c <= i) A[c] = cfms_conj(A[c], Lij, s_col[grp][c] * inv);
#pragma unroll
                    for (int r = 0; r < NB; ++r) b[r] = cfms(b[r], Lij, s_u[grp][r][0] * inv);
                } else if (i == jj) {
#pragma unroll
                    for (int r = 0; r < NB; ++r) b[r] = b[r] * inv;
                }
                __builtin_amdgcn_wave_barrier();
            }
            // b[r] is now row i of U = L^-1 [C | x]
#pragma unroll
            for (int r = 0; r < NB; ++r) s_u[grp][r][i] = (i < M) ? b[r] : cd{0, 0};
            __builtin_amdgcn_wave_barrier();
            // Gram entries: e < NG: G(r,r2) with r <= r2; e >= NG: g(r)
            for (int e = i; e < NE; e += MP) {
                int r = 0, r2 = 0;
                if (e < GramIdx<KM>::NG) {
                    int rem = e;
                    while (rem >= KM - r) { rem -= KM - r; ++r; }
                    r2 = r + rem;
                } else {
                    r = e - GramIdx<KM>::NG;
                    r2 = KM;
                }
                cd acc{0, 0};
                for (int m = 0; m < M; ++m) acc = cfma_conj(acc, s_u[grp][r2][m], s_u[grp][r][m]);
                s_e[grp][e] = acc;
            }
            __builtin_amdgcn_wave_barrier();
            // every lane solves the (KP1 x KP1) system G y = g redundantly; padding rows are identity
            cd Gm[KM][KM], gv[KM];
            {
                int e = 0;
#pragma unroll
                for (int r = 0; r < KM; ++r)
#pragma unroll
                    for (int r2 = r; r2 < KM; ++r2) {
                        const cd v = s_e[grp][e++];
                        Gm[r][r2] = v;
                        Gm[r2][r] = conj(v);
                    }
#pragma unroll
                for (int r = 0; r < KM; ++r) gv[r] = s_e[grp][GramIdx<KM>::NG + r];
#pragma unroll
                for (int r = 0; r < KM; ++r)
                    if (r >= KP1) {
#pragma unroll
                        for (int r2 = 0; r2 < KM; ++r2) {
                            Gm[r][r2] = cd{r == r2 ? 1.0 : 0.0, 0.0};
                            Gm[r2][r] = cd{r == r2 ? 1.0 : 0.0, 0.0};
                        }
                        gv[r] = cd{0, 0};
                    }
            }
            cd pinvs[KM];  // reciprocals of the pivots: formed once, used by the elimination and by the back substitution
#pragma unroll
            for (int k = 0; k < KM; ++k) {  // Gaussian elimination (G is Hermitian positive definite)
                const cd pinv = crcp(Gm[k][k]);
                pinvs[k] = pinv;
#pragma unroll
                for (int r = k + 1; r < KM; ++r) {
                    const cd fct = Gm[r][k] * pinv;
#pragma unroll
                    for (int c = k + 1; c < KM; ++c) Gm[r][c] = Gm[r][c] - fct * Gm[k][c];
                    gv[r] = gv[r] - fct * gv[k];
                }
            }
#pragma unroll
            for (int k = KM - 1; k >= 0; --k) {
                cd acc = gv[k];
#pragma unroll
                for (int c = k + 1; c < KM; ++c) acc = acc - Gm[k][c] * gv[c];
                gv[k] = acc * pinvs[k];
            }
            y = gv[0];
        } else {
            y = s_x[grp][0] * 0.01;  // in_fft(0,j)*0.01 (mvdr.cpp:96)
        }
        if (i == 0) st_y(a, yidx + t * kYhStride, q, y);
        // slide the covariance window: + x_t x_t^H - x_{t-P} x_{t-P}^H (mvdr.cpp:100-101)
#pragma unroll
        for (int c = 0; c < MP; ++c) R[c] = cfms_conj(cfma_conj(R[c], x, s_x[grp][c]), xo, s_xo[grp][c]);
        __builtin_amdgcn_wave_barrier();
    }
}


// ---- 16-lane DPP row helpers (row_newbcast exchange) of cov2d_kernel below; round 2's one-problem-per-row kernel is gone (EXPERIMENTS.md) ----------
// Same row-per-lane factorisation as mvdr_lcmv_kernel<16, KM>, but a problem occupies exactly one DPP row, so the
// pivot, the scaled column and the right-hand sides travel by `v_mov_b32_dpp row_newbcast:n` (lane n of every row to
// the whole row, one instruction per dword, VALU latency) instead of an LDS write -> s_waitcnt -> read round trip per
// column, and the Gram sums are row reductions (quad_perm xor 1/2, row_half_mirror, row_mirror).  No LDS at all.
// DPP note: every pattern in this file (quad_perm, row_newbcast, row mirrors) reads a valid source lane for every destination
// lane, so the `old` operand is dead.  With bound_ctrl = false hipcc still materialises it (v_mov_b32 old, 0 in front of every
// v_mov_b32_dpp: 812 + 812 instructions in the lcmv-16 solve block = 44 % of it); bound_ctrl = true drops the initialisation.
// (row_newbcast is the one DPP control gfx950 encodes for 64-bit operands: one v_mov_b64_dpp per double instead of two v_mov_b32_dpp)
template <int N>
__device__ __forceinline__ double rowbc(double v) {
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + N, 0xF, 0xF, true);
}
template <int N>
__device__ __forceinline__ cd rowbc(cd v) { return cd{rowbc<N>(v.x), rowbc<N>(v.y)}; }
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double row_sum(double v) {  // every lane of the 16-lane row gets the row total
    v += dpp_d<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_d<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_d<0x141>(v);  // row_half_mirror
    v += dpp_d<0x140>(v);  // row_mirror
    return v;
}
template <int C, int MP>
struct RowStep {  // compile-time loops over the broadcast source lane
    template <typename F>
    static __device__ __forceinline__ void run(F &&f) {
        f(std::integral_constant<int, C>{});
        RowStep<C + 1, MP>::run(f);
    }
};
template <int MP>
struct RowStep<MP, MP> {
    template <typename F>
    static __device__ __forceinline__ void run(F &&) {}
};

// ---- lcmv / mvdr, 9..16 microphones: 2-D cyclic 4 x 4 lanes per problem -------------------------------------------------
// One problem (stream, bin) per 16-lane DPP row (as round 2's row-per-lane kernel had it), but the lanes form a 4 x 4 grid (p = lane >> 2,
// q = lane & 3 inside the row) and matrix entry (i, c) lives in lane (i mod 4, c mod 4) at local index (i / 4, c / 4):
// every lane owns a 4 x 4 block of R and of the working copy (its lower triangle + diagonal: 10 entries), so
//   * the triangular trailing update keeps all 16 lanes busy until the last 4 x 4 block (the row-per-lane kernel idles half
//     of them on average: 1 020 fp64 instructions per wavefront-frame where 455 would do),
//   * a lane carries 10 + 10 matrix entries and 4 x 2 right-hand-side entries instead of 16 + 16 + 5: ~130 VGPRs, four
//     wavefronts per SIMD instead of two.
// Exchange per elimination step jj (column jj = 4 bj + qj):
//   pivot            lane (qj, qj)        -> whole row        DPP row_newbcast
//   scaled column    lanes (p, qj)        -> their quad       DPP quad_perm broadcast          Lrow[a] = L(4a+p, jj)
//   its transpose    lane (q, p)          -> lane (p, q)      ds_bpermute, fixed address       Lcol[b] = L(4b+q, jj)
//   right-hand side  lane (qj, q)         -> lanes (., q)     ds_bpermute                      u = b(jj, col) / L(jj, jj)
// Right-hand-side column r belongs to the lanes with q = r mod 4 (slot r / 4): constraints 0..3 in slot 0, the frame's x in
// slot 1 of the q = 0 lanes.  Rows that are already final (i <= jj) are switched off by zeroing their Lrow entry, so the
// updates run unconditionally.  Maths as everywhere in this file: R o whiteR = L L^H, U = L^-1 [C | x], G = U_C^H U_C,
// g = U_C^H u_x, y = (G^-1 g)_0.
template <int SRC>
__device__ __forceinline__ double quadbc(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), SRC * 0x55, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), SRC * 0x55, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <int SRC>
__device__ __forceinline__ cd quadbc(cd v) { return cd{quadbc<SRC>(v.x), quadbc<SRC>(v.y)}; }
__device__ __forceinline__ double bperm_d(int addr, double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_ds_bpermute(addr, (int)(b & 0xffffffffLL));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, (int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ cd bperm_c(int addr, cd v) { return cd{bperm_d(addr, v.x), bperm_d(addr, v.y)}; }
// sum over the four quads of a 16-lane row (same q): row_ror 4 and 8
__device__ __forceinline__ double quads_sum(double v) {
    v += dpp_d<0x124>(v);  // row_ror:4
    v += dpp_d<0x128>(v);  // row_ror:8
    return v;
}
// local lower triangle (a >= b) of a 4 x 4 block, row-major
__device__ constexpr int LT(int a, int b) { return a * (a + 1) / 2 + b; }

// Z128: the spectra are complex doubles (the default, BF_PRECISION_REFERENCE: 16 bytes per element, stored halved like z48) instead of z48
template <int KM, int WPS, bool Z128>
__global__ __launch_bounds__(256, WPS) void cov2d_kernel(BinsArgs a, int tile, int tiles_per_stream) {
    constexpr int NB = KM + 1, NS = (NB + 3) / 4;  // right-hand sides, slots per lane
    constexpr int NG = GramIdx<KM>::NG, NE = GramIdx<KM>::NE;
    // Row paddings against bank conflicts (round-3 PMC, profiles/r03_lcmv16_chain_pmc.txt: 53 % of this kernel's LDS cycles were
    // conflicts, 16 % of its wave cycles waited to issue an LDS instruction).  A ds_read_b128 serves 16 lanes at a time, drawn
    // from two of the wavefront's four problems; with 256-byte rows per problem both read the same banks:
    //   s_x: 320-byte rows (problem stride = 64 B mod 256): the four column entries x(4k+q) of two problems land 64 B apart;
    //   s_u: 272-byte rows: the Gram sums read the SAME row index i of up to five different columns at once (16 B apart now),
    //        and a problem's block of NB rows is 80 B mod 256 (NB = 5) / 32 B (NB = 2) from its neighbour's;
    //   s_g: 272-byte rows: every lane of a problem reads the same entry, two problems per access.
    __shared__ __attribute__((aligned(16))) f64x2 s_x[4][2][4][20];   // [wave][new / old][problem][mic]
    __shared__ __attribute__((aligned(16))) f64x2 s_u[4][4][NB][17];  // [wave][problem][column][row]: U = L^-1 [C | x]
    __shared__ __attribute__((aligned(16))) f64x2 s_g[4][4][4][17];   // [wave][problem][frame slot][Gram entry]: four frames' systems wait here
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l16 = lane & 15, grp = lane >> 4, p = l16 >> 2, q = l16 & 3;
    // Workgroup b runs on XCD b % 8 (round-robin dispatch), each XCD behind its own L2.  A block's 16 problems are 192 contiguous
    // bytes of every spectrum row -- one and a half cache lines -- so neighbouring problem groups share lines: XCD x takes the
    // (stream, tile) units x, x + 8, ... and walks the problem groups of a unit back to back, which puts the blocks that share
    // lines into the same L2 at the same time (a 2-D grid had them 1.5 residency windows apart: every shared line came twice).
    const int n_grp = (kNQ + 15) / 16;
    const int unit = (int)(blockIdx.x >> 3) / n_grp * 8 + (int)(blockIdx.x & 7);
    if (unit >= tiles_per_stream * a.n_streams) return;
    const int pq = ((int)(blockIdx.x >> 3) % n_grp) * 16 + wv * 4 + grp;
    const int s = unit / tiles_per_stream;
    const long tA = (long)(unit % tiles_per_stream) * tile;
    long tB = tA + tile;
    if (tB > a.n_frames) tB = a.n_frames;
    const int M = a.n_mics, NP = (M + 1) >> 1, KP1 = a.kp1, P = a.cfg.past_windows;
    const bool live = pq < kNQ;
    const int qq = live ? pq : kNQ - 1;
    const int j = q_bin(qq);
    const bool lcmv = a.cfg.algo == BF_LCMV;
    const long yidx = ((long)s * a.n_frames) * kYhStride + qq;
    typedef typename std::conditional<Z128, f64x2, z48>::type zel;  // stored element
    const zel *Zs = reinterpret_cast<const zel *>(a.Z) + ((long)(s / a.n_dirs) * a.frames_ws + a.frame_off) * NP * kN;
    const f64x2 *steer = a.steer + (long)(s % a.n_dirs) * a.steer_dir_stride;
    const int ksrc = q_src_bin(qq), kneg = (kN - ksrc) & (kN - 1);
    const int addrT = 4 * ((lane & ~15) | (q << 2) | p);  // my transpose partner (q, p)

    // microphone l16's spectrum at this bin, frame t (may be negative: history).  Requested a frame ahead (load_raw) and unpacked
    // where it is needed (finish_mic): a load issued where its value is used costs its full latency every frame -- the compiler cannot
    // move it above the st_y in between -- and two wavefronts per SIMD do not hide that
    struct RawMic {
        zel z, zc;
    };
    const int mic_pair = (l16 < M ? l16 : 0) >> 1;
    auto load_raw = [&](long t) -> RawMic {
        const zel *Zf = Zs + t * NP * kN + mic_pair * kN;
        return RawMic{Zf[ksrc], Zf[kneg]};
    };
    auto dec = [](const zel &v) -> cd {
        if constexpr (Z128) return cd{v.x, v.y};  // (stored halved, like z48)
        else return dec48(v);
    };
    auto finish_mic = [&](const RawMic &r) -> cd {
        if (l16 >= M) return cd{0, 0};
        const cd z = dec(r.z), zc = conj(dec(r.zc));
        cd x;
        if ((l16 & 1) == 0) {
            x = z + zc;  // z48 spectra are stored halved
        } else {
            const cd d = z - zc;
            x = cd{d.y, -d.x};
        }
        return qq == kQX ? conj(x) : x;
    };
    auto load_mic = [&](long t) -> cd { return finish_mic(load_raw(t)); };
    // one microphone per lane -> LDS -> this lane's four row entries x(4a+p) and four column entries x(4b+q).
    // LDS operations of one wavefront execute in issue order: compiler barriers only.
    auto spread = [&](cd x, int slot, cd (&xr)[4], cd (&xc)[4]) {
        s_x[wv][slot][grp][l16] = f64x2{x.x, x.y};
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            xr[k] = ld(&s_x[wv][slot][grp][4 * k + p]);
            xc[k] = ld(&s_x[wv][slot][grp][4 * k + q]);
        }
        __builtin_amdgcn_wave_barrier();
    };
    const double f = fabs(a.freqs[j]);
    const bool inband = live && f >= a.cfg.freq_min && f <= a.cfg.freq_max && !(j == 0 && !lcmv);
    if (__builtin_amdgcn_ballot_w64(inband) == 0) {  // nothing to solve in this wavefront
        for (long t = tA; t < tB; ++t) {
            cd y{0, 0};
            if (j == 0 && !lcmv) y = rowbc<0>(load_mic(t));  // mvdr.cpp:76
            if (live && l16 == 0) st_y(a, yidx + t * kYhStride, qq, y);
        }
        return;
    }
    // Right-hand sides: lane (p, q) holds rows 4a + q (a = 0..3) of column p + 4 sl -- rows by q, columns by p -- so that
    // the per-step u = b(jj, col) / L_jj comes from lane (p, qj) of the SAME quad (one DPP broadcast) and the update uses
    // Lcol.  Columns 0..KM-1 = constraints, column KM = the frame's x.
    auto rhs_init = [&](int arow, int col, const cd (&xc)[4]) -> cd {
        const int i = 4 * arow + q;
        if (col == KM) return xc[arow];
        return (col < KP1 && col < KM && i < M) ? ld(steer + ((long)col * M + i) * kN + j) : cd{0, 0};
    };

    // The constraint columns do not change over the tile: the two-wavefront build (195 registers of 256) keeps this lane's four
    // entries of column p instead of fetching them again every frame; the three-wavefront build has no register to spare.
    constexpr bool kKeepC = WPS == 2;
    cd cst[4];
    {
        const cd none[4] = {};
#pragma unroll
        for (int ar = 0; ar < 4; ++ar) cst[ar] = (kKeepC && p < KM) ? rhs_init(ar, p, none) : cd{0, 0};
    }

    cd R[10];
#pragma unroll
    for (int e = 0; e < 10; ++e) R[e] = cd{0, 0};
    // (the three-wavefront build has no registers for the 12 raw dwords in flight: 132 bytes of scratch, mvdr 16-mic 9.4 -> 10.0 ms; it
    // keeps the loads where the values are used.  Two-wavefront build, lcmv 16-mic K = 3: 10.7 -> 10.2 ms)
    constexpr bool kAhead = WPS == 2;
    RawMic nxt = load_raw(kAhead ? (P > 0 ? tA - 1 : tA) : tA);
    for (int pp = 1; pp <= P; ++pp) {  // covariance of the P frames in front of the tile
        const RawMic cur = kAhead ? nxt : load_raw(tA - pp);
        if (kAhead) nxt = load_raw(pp < P ? tA - pp - 1 : tA);  // (the last one: the tile's first frame)
        cd xr[4], xc[4];
        spread(finish_mic(cur), 0, xr, xc);
#pragma unroll
        for (int ar = 0; ar < 4; ++ar)
#pragma unroll
            for (int bc = 0; bc <= ar; ++bc) R[LT(ar, bc)] = cfma_conj(R[LT(ar, bc)], xr[ar], xc[bc]);
    }
    // The (K+1) x (K+1) system G y = g of a frame has no successor in the recursion (only R slides on), and every lane of a
    // problem's row would solve it redundantly: the Gram entries of four consecutive frames are parked in LDS instead and
    // lanes 0..3 of the row solve one frame each -- one pass of the small solve per four frames.
    unsigned open_mask = 0;  // frame slots of this row whose system waits (uniform per row)
    for (long t = tA; t < tB; ++t) {
        const int slot = (int)(t - tA) & 3;
        const cd xm = finish_mic(kAhead ? nxt : load_raw(t));
        RawMic old = nxt;
        if (kAhead) {
            old = load_raw(t - P);                       // needed behind the factorisation
            nxt = load_raw(t + 1 < tB ? t + 1 : t);      // next iteration's frame
        }
        cd xr[4], xc[4];
        spread(xm, 0, xr, xc);
        const double mag = row_sum(fast_sqrt(norm2(xm))) / (double)((unsigned)M * (unsigned)kN);
        const cd x0 = rowbc<0>(xm);
        cd y{0, 0};
        bool deferred = false;
        if (mag > a.cfg.freq_mag_threshold) {  // uniform per row
            cd A[10], b[4][NS];
#pragma unroll
            for (int ar = 0; ar < 4; ++ar) {
#pragma unroll
                for (int bc = 0; bc <= ar; ++bc) {
                    cd v = R[LT(ar, bc)];
                    if (ar == bc) {
                        const int i = 4 * ar + p;
                        if (p == q) v = (i < M) ? v * 1.001 : cd{1.0, 0.0};  // whiteR diagonal (mvdr.cpp:239-243); padding = identity
                    }
                    A[LT(ar, bc)] = v;
                }
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) {
                    if (kKeepC && sl == 0 && KM <= 4)
                        b[ar][0] = p == KM ? xc[ar] : (p < KM ? cst[ar] : cd{0, 0});
                    else
                        b[ar][sl] = (p + 4 * sl < NB) ? rhs_init(ar, p + 4 * sl, xc) : cd{0, 0};
                }
            }
            RowStep<0, 16>::run([&](auto jc) {
                constexpr int jj = decltype(jc)::value, bj = jj >> 2, qj = jj & 3;
                const double inv = fast_rsqrt1(rowbc<5 * qj>(A[LT(bj, bj)].x));  // 1 / L_jj from lane (qj, qj)
                cd Lrow[4], Lcol[4], u[NS];
#pragma unroll
                for (int ar = bj; ar < 4; ++ar) Lrow[ar] = quadbc<qj>(A[LT(ar, bj)] * inv);  // L(4ar+p, jj) from lane (p, qj)
                if (p <= qj) Lrow[bj] = cd{0, 0};                                              // rows i <= jj are final
#pragma unroll
                for (int bc = bj; bc < 4; ++bc) Lcol[bc] = bperm_c(addrT, Lrow[bc]);            // L(4bc+q, jj), 0 for rows <= jj
                const double inv_q = q == qj ? inv : 1.0;  // row jj itself becomes U(jj, col); the other rows of the block keep their value (x 1.0 is exact)
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) {  // meanwhile: u_jj of my column(s), from lane (p, qj) of my quad
                    b[bj][sl] = b[bj][sl] * inv_q;
                    u[sl] = quadbc<qj>(b[bj][sl]);
                }
#pragma unroll
                for (int ar = bj; ar < 4; ++ar)
#pragma unroll
                    for (int bc = bj; bc <= ar; ++bc) A[LT(ar, bc)] = cfms_conj(A[LT(ar, bc)], Lrow[ar], Lcol[bc]);
#pragma unroll
                for (int sl = 0; sl < NS; ++sl)
#pragma unroll
                    for (int ar = bj; ar < 4; ++ar) b[ar][sl] = cfms(b[ar][sl], Lcol[ar], u[sl]);
            });
            // b = rows 4a + q of U for column(s) p + 4 sl.  Through LDS: [column][row]; Gram entry e is summed by lane e of
            // the row over the 16 rows, published, and read back by every lane (the small solve runs redundantly).
#pragma unroll
            for (int sl = 0; sl < NS; ++sl)
                if (p + 4 * sl < NB) {
#pragma unroll
                    for (int ar = 0; ar < 4; ++ar) {
                        const cd v = (4 * ar + q < M) ? b[ar][sl] : cd{0, 0};
                        s_u[wv][grp][p + 4 * sl][4 * ar + q] = f64x2{v.x, v.y};
                    }
                }
            __builtin_amdgcn_wave_barrier();
            {
                // entry l16 < NG: G(r1, r2), r1 <= r2, row-major upper triangle; NG <= l16 < NE: g(r1) = U(:, r1)^H u_x
                int r1 = 0, r2 = KM;
                if (l16 < NG) {
                    int rem = l16;
                    while (rem >= KM - r1) { rem -= KM - r1; ++r1; }
                    r2 = r1 + rem;
                } else {
                    r1 = l16 - NG;
                }
                cd acc{0, 0};
                if (l16 < NE) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc = cfma_conj(acc, ld(&s_u[wv][grp][r2][i]), ld(&s_u[wv][grp][r1][i]));
                }
                s_g[wv][grp][slot][l16] = f64x2{acc.x, acc.y};
            }
            deferred = true;
        } else {
            y = x0 * 0.01;  // in_fft(0,j)*0.01 (mvdr.cpp:96)
        }
        if (!inband) {
            y = (j == 0 && !lcmv) ? x0 : cd{0, 0};
            deferred = false;
        }
        if (deferred)
            open_mask |= 1u << slot;
        else if (live && l16 == 0)
            st_y(a, yidx + t * kYhStride, qq, y);
        if (slot == 3 || t == tB - 1) {  // uniform: the parked systems, one frame per lane 0..3 of the row
            __builtin_amdgcn_wave_barrier();
            const int fs = l16 & 3;
            cd ge[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) ge[e] = ld(&s_g[wv][grp][fs][e]);
            __builtin_amdgcn_wave_barrier();
            // (K+1) x (K+1) system G y = g on the upper triangle (one system per lane)
            cd gv[KM];
            auto UI = [](int r, int c) { return r * KM - r * (r - 1) / 2 + (c - r); };
#pragma unroll
            for (int r1 = 0; r1 < KM; ++r1) gv[r1] = ge[NG + r1];
#pragma unroll
            for (int r1 = 0; r1 < KM; ++r1)
                if (r1 >= KP1) {
#pragma unroll
                    for (int r2 = 0; r2 < r1; ++r2) ge[UI(r2, r1)] = cd{0, 0};
                    ge[UI(r1, r1)] = cd{1.0, 0.0};
#pragma unroll
                    for (int c = r1 + 1; c < KM; ++c) ge[UI(r1, c)] = cd{0, 0};
                    gv[r1] = cd{0, 0};
                }
            cd pinvs[KM];  // reciprocals of the pivots: formed once, used by the elimination and by the back substitution
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                const cd pinv = crcp(ge[UI(k, k)]);
                pinvs[k] = pinv;
#pragma unroll
                for (int r1 = k + 1; r1 < KM; ++r1) {
                    const cd fct = conj(ge[UI(k, r1)]) * pinv;
#pragma unroll
                    for (int c = r1; c < KM; ++c) ge[UI(r1, c)] = ge[UI(r1, c)] - fct * ge[UI(k, c)];
                    gv[r1] = gv[r1] - fct * gv[k];
                }
            }
#pragma unroll
            for (int k = KM - 1; k >= 0; --k) {
                cd acc = gv[k];
#pragma unroll
                for (int c = k + 1; c < KM; ++c) acc = acc - ge[UI(k, c)] * gv[c];
                gv[k] = acc * pinvs[k];
            }
            // lane fs of the row holds frame (t - slot + fs)
            if (live && l16 < 4 && fs <= slot && ((open_mask >> fs) & 1u)) st_y(a, yidx + (t - slot + fs) * kYhStride, qq, gv[0]);
            open_mask = 0;
        }
        // slide the covariance window: + x_t x_t^H - x_{t-P} x_{t-P}^H (mvdr.cpp:100-101).  x_t is read back from its LDS slot
        // (still there) instead of being kept in 32 registers across the factorisation.
        cd xor_[4], xoc[4];
        spread(finish_mic(kAhead ? old : load_raw(t - P)), 1, xor_, xoc);
#pragma unroll
        for (int ar = 0; ar < 4; ++ar) {
            const cd xra = kKeepC ? xr[ar] : ld(&s_x[wv][0][grp][4 * ar + p]);
#pragma unroll
            for (int bc = 0; bc <= ar; ++bc) {
                const cd xcb = kKeepC ? xc[bc] : ld(&s_x[wv][0][grp][4 * bc + q]);
                R[LT(ar, bc)] = cfms_conj(cfma_conj(R[LT(ar, bc)], xra, xcb), xor_[ar], xoc[bc]);
            }
        }
    }
}

}  // namespace

hipError_t launch_mvdr_lcmv(const BinsArgs &a, int n_cus, hipStream_t s) {
    int tile = 64;
    if (a.n_frames < tile) tile = (int)a.n_frames;
    const int tps = (int)((a.n_frames + tile - 1) / tile);
    const int M = a.n_mics, km = a.kp1 <= 1 ? 1 : 4;
    static const bool no_fast_env = getenv("BF_MVDR_GROUP") && atoi(getenv("BF_MVDR_GROUP")) != 0;
    const bool no_fast = no_fast_env;
    const bool no_2d = no_fast_env;
    // frequencies of the irregular problems (quirk Q1, util.h:190-199): f[N/2] = 0, f[N/2+1] = -(N/2-1) sr/N
    const double f_qx = (double)(kN / 2 - 1) * a.cfg.sample_rate / (double)kN;
    const bool band_hits_nyquist = (0.0 >= a.cfg.freq_min && 0.0 <= a.cfg.freq_max) || (f_qx >= a.cfg.freq_min && f_qx <= a.cfg.freq_max);
    // lcmv with up to 8 microphones rides mvdr_fast_kernel while its columns fit the register file beside R and its working copy:
    // any K <= 3 up to 6 microphones, K <= 2 at 7-8 (K = 3 at 7-8 spills 36 registers and still beats every other kernel)
    const bool lcmv_fast = !no_fast && a.cfg.algo == BF_LCMV && M <= 8 && a.kp1 <= 4;
    (void)band_hits_nyquist;
#define BF_LAUNCH_ML(MP_, KM_)                                                                                   \
    BF_LAUNCH((mvdr_lcmv_kernel<MP_, KM_>), dim3(tps * a.n_streams, (kNQ + (256 / MP_) - 1) / (256 / MP_)), \
                       dim3(256), 0, s, a, tile, tps)
    // beyond the tuned shapes -- more than 3 interferers (lcmv.cpp:258-309 appends without a cap; the yaml lists
    // angle_interf1..15) or more than 16 microphones -- the row-per-lane group kernel runs with the next larger
    // (lanes per problem, constraint columns) instantiation; padding rows / columns are identity
    if (a.kp1 > 4 || M > 16) {
        if (a.kp1 > 16 || M > 32) return hipErrorInvalidValue;
        if (a.kp1 <= 1) {
            BF_LAUNCH_ML(32, 1);
        } else if (a.kp1 <= 4) {
            BF_LAUNCH_ML(32, 4);
        } else if (a.kp1 <= 8) {
            if (M <= 8) BF_LAUNCH_ML(8, 8);
            else if (M <= 16) BF_LAUNCH_ML(16, 8);
            else BF_LAUNCH_ML(32, 8);
        } else {
            if (M <= 16) BF_LAUNCH_ML(16, 16);
            else BF_LAUNCH_ML(32, 16);
        }
        return hipGetLastError();
    }
    // 9..16 microphones, up to 3 interferers: 2-D cyclic 4 x 4 lanes per problem.  Wavefronts per SIMD: with constraint columns (lcmv) the
    // three-wavefront build spills 30 registers per lane and the spill traffic alone is 6 GB per 32 768 frames of 16 microphones: two
    // wavefronts at 211 registers are 7 % faster; without constraints (mvdr) three wavefronts win by 12 %
    if (!no_2d && M > 8 && M <= 16) {
        const dim3 grid((unsigned)(((long)tps * a.n_streams + 7) / 8 * 8 * ((kNQ + 15) / 16)));  // (unit, problem group) -> XCD-aware order in the kernel
        if (km == 1) {
            if (a.z48) BF_LAUNCH((cov2d_kernel<1, 3, false>), grid, dim3(256), 0, s, a, tile, tps);
            else BF_LAUNCH((cov2d_kernel<1, 3, true>), grid, dim3(256), 0, s, a, tile, tps);
        } else {
            if (a.z48) BF_LAUNCH((cov2d_kernel<4, 2, false>), grid, dim3(256), 0, s, a, tile, tps);
            else BF_LAUNCH((cov2d_kernel<4, 2, true>), grid, dim3(256), 0, s, a, tile, tps);
        }
        return hipGetLastError();
    }
    if ((a.cfg.algo == BF_MVDR || lcmv_fast) && M <= 8 && !no_fast) {
        // the problems that need a solve (FastPlan): 0, the in-band run inside 1 .. N/2-1, and N/2 / N/2+1 when in band
        const std::vector<double> fr = frequency_vector(kN, a.cfg.sample_rate);  // the table the other kernels read (a.freqs)
        auto inb = [&](int q) {
            const double f = std::fabs(fr[q_bin_host(q)]);
            return f >= a.cfg.freq_min && f <= a.cfg.freq_max;
        };
        FastPlan fp;
        fp.q_lo = 1;
        while (fp.q_lo < kN / 2 && !inb(fp.q_lo)) ++fp.q_lo;
        fp.n_main = 0;
        while (fp.q_lo + fp.n_main < kN / 2 && inb(fp.q_lo + fp.n_main)) ++fp.n_main;
        for (int q = fp.q_lo + fp.n_main; q < kN / 2; ++q)
            if (inb(q)) return hipErrorInvalidValue;  // cannot happen: |f| rises with q below N/2
        fp.n_extra = 0;
        fp.extra_q[0] = fp.extra_q[1] = 0;
        if (inb(kN / 2)) fp.extra_q[fp.n_extra++] = kN / 2;          // f[N/2] := 0 (quirk Q1): in band when 0 Hz is
        if (inb(kN / 2 + 1)) fp.extra_q[fp.n_extra++] = kN / 2 + 1;  // the conjugate problem
        fp.solve0 = (a.cfg.algo == BF_LCMV && inb(0)) ? 1 : 0;
        fp.nb = 1 + fp.n_main + fp.n_extra;
        // tile length: the wavefronts (64 lanes = 64 (tile, problem) pairs) should fill the 4 x CUs slots a whole number of times;
        // cost of a choice = rounds x (frames walked + P warm-up frames); BF_MVDR_TILE forces a length (tests: lanes that straddle tiles,
        // a short last tile)
        static const int ft_env = getenv("BF_MVDR_TILE") ? atoi(getenv("BF_MVDR_TILE")) : 0;
        // resident wavefronts per CU: the prefetch buffers (2 x 2 MP rows of 1 KiB in LDS) and the register count of the instantiation
        // that will run decide -- 8 microphones: one per SIMD (512 registers); fewer microphones: more
        const int mp_ = M <= 2 ? 2 : M <= 4 ? 4 : M <= 6 ? 6 : 8;
        const int kc_ = (!lcmv_fast || a.kp1 <= 1) ? 1 : a.kp1;
        const int wpc = mp_ == 2 ? 16 : mp_ == 4 ? (kc_ <= 2 ? 9 : 8) : mp_ == 6 ? (kc_ == 1 ? 6 : 4) : 4;
        const long slots = (long)n_cus * wpc, F = a.n_frames;
        long best_t = 1;
        double best_c = 1e300;
        for (long t = 1; t <= 512 && t <= F; ++t) {
            const long tiles = (F + t - 1) / t;
            const long waves = (long)a.n_streams * ((tiles * fp.nb + 63) / 64);
            const long rounds = (waves + slots - 1) / slots;
            const double c = (double)rounds * (double)(t + a.cfg.past_windows + 2);
            if (c <= best_c) { best_c = c; best_t = t; }
        }
        if (ft_env > 0) best_t = ft_env < F ? ft_env : F;
        fp.tile = (int)best_t;
        fp.tiles = (int)((F + best_t - 1) / best_t);
        fp.waves_per_stream = (int)(((long)fp.tiles * fp.nb + 63) / 64);
        const bool band_rows = a.yh_lo > 0 || a.yh_hi < a.yh_lo;  // the consumer reads problem 0 and yh_lo..yh_hi only (a band below the Nyquist problems, no dump)
        if (!a.yh32 && !band_rows)  // f64x2 rows read in full (spectrum dump, other FFT sizes, a band up to Nyquist): the rows nobody solves read as zero (mvdr.cpp:103)
            (void)hipMemsetAsync(a.Yh, 0, (size_t)a.n_streams * a.n_frames * kYhStride * sizeof(f64x2), s);
        else if (a.yh32 && a.yh_lo == 0 && fp.nb < kNQ)  // f32x2 rows that the backward transform reads in full (a band up to the Nyquist problems) while part of them is out of band
            (void)hipMemsetAsync(a.Yh, 0, (size_t)a.n_streams * a.n_frames * kYhStride * sizeof(f32x2), s);
        const dim3 grid((unsigned)((long)fp.waves_per_stream * a.n_streams));
#define BF_FAST_GO(MP_, KC_)                                                                     \
    do {                                                                                         \
        if (a.z48) BF_LAUNCH((mvdr_fast_kernel<MP_, KC_, false>), grid, dim3(64), 0, s, a, fp);  \
        else BF_LAUNCH((mvdr_fast_kernel<MP_, KC_, true>), grid, dim3(64), 0, s, a, fp);         \
    } while (0)
        if (!lcmv_fast || a.kp1 <= 1) {  // lcmv without interferers = mvdr except problem 0
            if (M <= 2) BF_FAST_GO(2, 1);
            else if (M <= 4) BF_FAST_GO(4, 1);
            else if (M <= 6) BF_FAST_GO(6, 1);
            else BF_FAST_GO(8, 1);
        } else if (M <= 2) {
            BF_FAST_GO(2, 2);
        } else if (M <= 4) {
            if (a.kp1 <= 2) BF_FAST_GO(4, 2); else BF_FAST_GO(4, 4);
        } else if (M <= 6) {
            if (a.kp1 <= 2) BF_FAST_GO(6, 2); else BF_FAST_GO(6, 4);
        } else {
            if (a.kp1 <= 2) BF_FAST_GO(8, 2); else if (a.kp1 == 3) BF_FAST_GO(8, 3); else BF_FAST_GO(8, 4);
        }
#undef BF_FAST_GO
        return hipGetLastError();
    }
    if (M <= 4) {
        if (km == 1) BF_LAUNCH_ML(4, 1); else BF_LAUNCH_ML(4, 4);
    } else if (M <= 8) {
        if (km == 1) BF_LAUNCH_ML(8, 1); else BF_LAUNCH_ML(8, 4);
    } else {
        if (km == 1) BF_LAUNCH_ML(16, 1); else BF_LAUNCH_ML(16, 4);
    }
#undef BF_LAUNCH_ML
    return hipGetLastError();
}

}  // namespace BF_NTAG
}  // namespace bf
