// gsc_gss_kernels.hip -- the two adaptive nodes: gss (per-bin demixing-matrix recursion) and gsc (per-microphone alignment
// + sample-serial float32 NLMS).
#include <cstdlib>

#include "launch_trace.hpp"
#include "bins_common.hpp"

namespace bf {
namespace BF_NTAG {

namespace {

// ======================================================================================
//                 gsc: generalized sidelobe canceller (gsc.cpp:54-197)
// ======================================================================================
// Pass 1 (align): output stream s*M + m carries microphone m of input stream s steered to the look direction,
// y_fft = x_fft * conj(weights[m]) over all bins (gsc.cpp:62-70); the ISTFT then does the per-microphone
// overlap-add of do_overlap_bymic (util.h:353-379).
__global__ __launch_bounds__(256) void gsc_align_kernel(BinsArgs a) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)a.n_streams * a.n_frames * kNQ;
    if (idx >= total) return;
    const int q = (int)(idx % kNQ);
    const long st = idx / kNQ;
    const long t = st % a.n_frames;
    const int so = (int)(st / a.n_frames);
    const int M = a.n_mics, NP = (M + 1) >> 1;
    const int si = so / M, m = so - si * M;
    const f64x2 *Zf = a.Z + (((long)si * a.frames_ws + a.frame_off + t) * NP + (m >> 1)) * kN;
    const int k = q_src_bin(q), kn = (kN - k) & (kN - 1);
    const cd z = ld(Zf + k), zc = conj(ld(Zf + kn));
    cd x;
    if ((m & 1) == 0) {
        x = (z + zc) * 0.5;
    } else {
        const cd d = z - zc;
        x = cd{0.5 * d.y, -0.5 * d.x};
    }
    if (q == kQX) x = conj(x);
    const cd y = x * conj(ld(a.steer + (long)m * kN + q_bin(q)));
    a.Yh[((long)so * a.n_frames + t) * kYhStride + q] = f64x2{y.x, y.y};
}

// Pass 2 (NLMS): one wavefront per stream, strictly sample by sample.  The reference does this arithmetic in
// float32 (rosjack_data) with every product and sum rounded separately and the 128-tap sums taken in order, and
// it branches on the results (mu selection, NaN guards), so the kernel keeps exactly that order: lane i owns
// blocking branch i and walks its taps sequentially (plain * and + under `#pragma clang fp contract(off)`: no FMA contraction), lane M-1 does
// the same for the output-power window.  Only what is elementwise is spread over the lanes: the upper beamformer
// and the neighbour differences of a 64-sample tile (lane = sample), and the filter update (lane = tap), whose
// coefficients live in registers with a write-through copy in LDS for the serial walk.
// Windows are mirrored rings in LDS (each sample stored at p and p + fs) so a window is always contiguous.
// A block is one wavefront: its LDS operations complete in issue order, so phases are separated by compiler
// fences (wave_barrier), not s_barrier.
template <int NBM, int KPL>  // NBM >= blocking branches (M - 1), KPL >= ceil(filter_size / 64)
__global__ __launch_bounds__(64) void gsc_nlms_kernel(const float *aligned, float *y, float *state, long n, int M, int fs,
                                                      int use_vad, double vad_threshold, double mu0, double mu_max) {
    // every float product and sum below must round on its own, as the reference's x86 build does; the __fmul_rn / __fadd_rn
    // intrinsics do NOT guarantee that (they inline to a*b / a+b with the caller's contraction allowed), this pragma does
#pragma clang fp contract(off)
    extern __shared__ float gl[];
    const int lane = threadIdx.x;
    const int nb = M - 1;                 // blocking branches
    const int nbr = nb > 0 ? nb : 1;
    // row lengths are padded to a whole number of 64-tap lane groups (+8 for the staged loads of the serial walk) and
    // made odd, so neither the walk nor the update needs a per-lane bounds guard and lane i / row i hit distinct banks
    const int bstride = (fs + 64 * KPL + 8) | 1;  // mirrored ring: a window starts at h1 < fs
    const int fstride = (64 * KPL + 8) | 1;
    float *s_bm = gl;                     // [nb][bstride]
    float *s_f = s_bm + nbr * bstride;    // [nb][fstride]
    float *s_lo = s_f + nbr * fstride;    // [2*fs]
    float *s_d = s_lo + 2 * fs + 16;      // [nb][64] neighbour differences of the current tile (16 words of slack first)
    float *s_das = s_d + nbr * 64;        // [64] upper beamformer of the current tile
    float *s_c = s_das + 64;              // [16] mu_i * out
    float *s_out = s_c + 16;              // [64]
    const int s = blockIdx.x;
    const float *as = aligned + (long)s * M * n;
    float *ys = y + (long)s * n;
    float *sv = state + (long)s * (2 * nb + 1) * fs;
    float freg[NBM][KPL];  // filter taps k = lane + 64 c of every branch
#pragma unroll
    for (int i = 0; i < NBM; ++i)
#pragma unroll
        for (int c = 0; c < KPL; ++c) {
            const int k = lane + 64 * c;
            float v = 0.f;
            if (i < nb && k < fs) {
                const float b = sv[i * fs + k];
                s_bm[i * bstride + k] = b;
                s_bm[i * bstride + k + fs] = b;
                v = sv[nb * fs + i * fs + k];
                s_f[i * fstride + k] = v;
            }
            freg[i][c] = v;
        }
    for (int k = lane; k < fs; k += 64) {
        const float v = sv[2 * nb * fs + k];
        s_lo[k] = v;
        s_lo[k + fs] = v;
    }
    __builtin_amdgcn_wave_barrier();
    int h = 0;  // ring position of the oldest element (same for every window: all advance once per sample)
    const float fsz = (float)fs;
    const bool is_branch = lane < nb, is_lo = lane == nb;
    const float *myrow = is_branch ? s_bm + lane * bstride : s_lo;  // lane nb (= M-1) walks the output window
    const float *myflt = is_branch ? s_f + lane * fstride : s_f;
    for (long n0 = 0; n0 < n; n0 += 64) {
        {   // tile prologue, lane = sample: das_out (gsc.cpp:122-127) and the blocking-matrix inputs (gsc.cpp:131)
            const bool ok = n0 + lane < n;
            float prev = ok ? as[n0 + lane] : 0.f, das = 0.f;
            das = (das + prev);
            for (int m = 1; m < M; ++m) {
                const float cur = ok ? as[(long)m * n + n0 + lane] : 0.f;
                das = (das + cur);
                s_d[(m - 1) * 64 + lane] = (cur - prev);
                prev = cur;
            }
            s_das[lane] = __fdiv_rn(das, (float)M);
        }
        __builtin_amdgcn_wave_barrier();
        const int cnt = (n - n0) < 64 ? (int)(n - n0) : 64;
        for (int jj = 0; jj < cnt; ++jj) {
            const float das = s_das[jj];
            if (is_branch) {
                const float d = s_d[lane * 64 + jj];
                s_bm[lane * bstride + h] = d;
                s_bm[lane * bstride + h + fs] = d;
            }
            const int h1 = (h + 1 == fs) ? 0 : h + 1;  // window = [h1, h1 + fs)
            __builtin_amdgcn_wave_barrier();
            // lane i: block_out_i and the sum of squares of its window; lane nb: sum of squares of the output
            // window WITHOUT its newest element (added below, last, as the reference's loop order has it).
            // Loads are unconditional (every row is fs words long) and staged one group of 8 taps ahead of the
            // two dependent add chains.
            float bo = 0.f, pw = 0.f;
            {
                const float *u = myrow + h1;
                float un[8], wn[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    un[r] = u[r];
                    wn[r] = myflt[r];
                }
                // taps 0 .. fs-2 are common to all lanes; tap fs-1 belongs to the branch lanes only (the output
                // window's newest element is not known yet)
                int k = 0;
                for (; k + 8 <= fs - 1; k += 8) {
                    float uv[8], wv[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        uv[r] = un[r];
                        wv[r] = wn[r];
                    }
#pragma unroll
                    for (int r = 0; r < 8; ++r) {  // next group (reads past the window end land in the row padding)
                        un[r] = u[k + 8 + r];
                        wn[r] = myflt[k + 8 + r];
                    }
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        bo = (bo + (wv[r] * uv[r]));
                        pw = (pw + (uv[r] * uv[r]));
                    }
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) {  // leftover taps, already staged
                    if (k + r < fs - 1 || (k + r == fs - 1 && !is_lo)) {
                        bo = (bo + (wn[r] * un[r]));
                        pw = (pw + (un[r] * un[r]));
                    }
                }
            }
            float out = das;
            for (int i = 0; i < nb; ++i)
                out = out - __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bo), i));
            const float pwl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pw), nb));
            const float lop = __fsqrt_rn(__fdiv_rn((pwl + (out * out)), fsz));  // calculate_power(last_outputs)
            if (lane == 0) {
                s_lo[h] = out;
                s_lo[h + fs] = out;
                s_out[jj] = out;
            }
            const bool adapt = ((double)lop < vad_threshold) || !use_vad;  // gsc.cpp:147
            if (adapt && nb > 0) {
                if (is_branch) {
                    const float bp = __fsqrt_rn(__fdiv_rn(pw, fsz));
                    float mu;
                    if (mu0 * (double)bp / (double)lop < mu_max)  // gsc.cpp:153-157 (double arithmetic: mu0 is a double)
                        mu = (float)(mu0 / (double)lop);
                    else
                        mu = (float)(mu0 / (double)bp);
                    if (isnan(mu) || isinf(mu)) mu = 0.f;
                    s_c[lane] = (mu * out);
                }
                __builtin_amdgcn_wave_barrier();
                // filter[i][k] += this_mu*out[j]*block_matrix[i][k] (gsc.cpp:163-170), taps over the lanes:
                // all loads first, then the arithmetic and the write-through stores
                float bmv[NBM][KPL], cv[NBM];
#pragma unroll
                for (int i = 0; i < NBM; ++i) {
                    const int ic = i < nb ? i : 0;
                    cv[i] = s_c[ic];
#pragma unroll
                    for (int c = 0; c < KPL; ++c) bmv[i][c] = s_bm[ic * bstride + h1 + lane + 64 * c];
                }
#pragma unroll
                for (int i = 0; i < NBM; ++i)
                    if (i < nb) {  // uniform
#pragma unroll
                        for (int c = 0; c < KPL; ++c) {  // lanes past filter_size work on row padding nobody reads
                            float fv = (freg[i][c] + (cv[i] * bmv[i][c]));
                            if (isnan(fv)) fv = 0.f;
                            freg[i][c] = fv;
                            s_f[i * fstride + lane + 64 * c] = fv;
                        }
                    }
            }
            __builtin_amdgcn_wave_barrier();
            h = h1;
        }
        if (lane < cnt) ys[n0 + lane] = s_out[lane];
        __builtin_amdgcn_wave_barrier();
    }
    // carried state in the reference's (shifted, oldest-first) order
    for (int e = lane; e < nb * fs; e += 64) {
        const int i = e / fs, k = e - i * fs;
        sv[e] = s_bm[i * bstride + h + k];
        sv[nb * fs + e] = s_f[i * fstride + k];
    }
    for (int k = lane; k < fs; k += 64) sv[2 * nb * fs + k] = s_lo[h + k];
}

// Pass 2, taps over the lanes (the default; BF_GSC_SERIAL=1 selects the kernel above).  The sample loop stays serial -- the filter that
// produces sample n was updated with sample n - 1 -- but inside a sample the filter_size-tap dot products of the nb blocking branches, their
// window powers and the output window's power are spread over the 64 lanes (tap k = lane + 64 c) and summed by DPP butterflies inside the
// 16-lane rows plus four v_readlane: ~200 vector instructions per sample instead of a dependent chain of filter_size additions per lane
// with an LDS round trip every eight taps (3 us per sample: 99 ms per 256 streams x 64 frames).  Every product and partial sum still
// rounds on its own (no FMA contraction); what changes against the reference is the ORDER of the float additions inside a sum
// (pairwise instead of tap 0, 1, 2, ...): results agree to float rounding of the sums (1e-7 relative per sample, observed <= 1e-6 on
// the output against the oracle), not bit for bit; the comparisons the reference branches on see the same quantities.
template <int CTRL>
__device__ __forceinline__ float gsc_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the 64 lanes, the same value in every lane (separately rounded float additions, pairwise order): butterflies inside the
// 16-lane rows, then row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3, and lane 63 holds the total
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float gsc_dpp_rows(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float gsc_wave_sum(float v) {
#pragma clang fp contract(off)
    v = v + gsc_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
    v = v + gsc_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
    v = v + gsc_dpp<0x141>(v);  // row_half_mirror
    v = v + gsc_dpp<0x140>(v);  // row_mirror: every lane of a 16-lane row holds the row's sum
    v = v + gsc_dpp_rows<0x142, 0xA>(v);  // rows 1, 3 += lane 15 of rows 0, 2
    v = v + gsc_dpp_rows<0x143, 0xC>(v);  // rows 2, 3 += lane 31
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

template <int NBM, int KPL>  // NBM >= blocking branches (M - 1), KPL >= ceil(filter_size / 64)
__global__ __launch_bounds__(64) void gsc_nlms_par_kernel(const float *aligned, float *y, float *state, long n, int M, int fs,
                                                          int use_vad, double vad_threshold, double mu0, double mu_max) {
#pragma clang fp contract(off)
    extern __shared__ float gl[];
    const int lane = threadIdx.x;
    const int nb = M - 1;                 // blocking branches
    const int nbr = nb > 0 ? nb : 1;
    const int bstride = (fs + 64 * KPL + 8) | 1;  // mirrored ring: a window starts at h1 < fs; reads past it land in the row padding
    float *s_bm = gl;                     // [nb][bstride]
    float *s_lo = s_bm + nbr * bstride;   // [2*fs + 64*KPL]
    float *s_d = s_lo + 2 * fs + 64 * KPL + 16;  // [nb][64] neighbour differences of the current tile
    float *s_das = s_d + nbr * 64;        // [64] upper beamformer of the current tile
    float *s_out = s_das + 64;            // [64]
    const int s = blockIdx.x;
    const float *as = aligned + (long)s * M * n;
    float *ys = y + (long)s * n;
    float *sv = state + (long)s * (2 * nb + 1) * fs;
    float freg[NBM][KPL];  // filter taps k = lane + 64 c of every branch
    bool tap_ok[KPL];
#pragma unroll
    for (int c = 0; c < KPL; ++c) tap_ok[c] = lane + 64 * c < fs;
#pragma unroll
    for (int i = 0; i < NBM; ++i)
#pragma unroll
        for (int c = 0; c < KPL; ++c) {
            const int k = lane + 64 * c;
            float v = 0.f;
            if (i < nb && k < fs) {
                const float b = sv[i * fs + k];
                s_bm[i * bstride + k] = b;
                s_bm[i * bstride + k + fs] = b;
                v = sv[nb * fs + i * fs + k];
            }
            freg[i][c] = v;
        }
    for (int k = lane; k < fs; k += 64) {
        const float v = sv[2 * nb * fs + k];
        s_lo[k] = v;
        s_lo[k + fs] = v;
    }
    __builtin_amdgcn_wave_barrier();
    int h = 0;  // ring position of the oldest element (same for every window: all advance once per sample)
    const float fsz = (float)fs;
    for (long n0 = 0; n0 < n; n0 += 64) {
        {   // tile prologue, lane = sample: das_out (gsc.cpp:122-127) and the blocking-matrix inputs (gsc.cpp:131)
            const bool ok = n0 + lane < n;
            float prev = ok ? as[n0 + lane] : 0.f, das = 0.f;
            das = (das + prev);
            for (int m = 1; m < M; ++m) {
                const float cur = ok ? as[(long)m * n + n0 + lane] : 0.f;
                das = (das + cur);
                s_d[(m - 1) * 64 + lane] = (cur - prev);
                prev = cur;
            }
            s_das[lane] = __fdiv_rn(das, (float)M);
        }
        __builtin_amdgcn_wave_barrier();
        const int cnt = (n - n0) < 64 ? (int)(n - n0) : 64;
        for (int jj = 0; jj < cnt; ++jj) {
            const float das = s_das[jj];
            if (lane < nb) {
                const float d = s_d[lane * 64 + jj];
                s_bm[lane * bstride + h] = d;
                s_bm[lane * bstride + h + fs] = d;
            }
            const int h1 = (h + 1 == fs) ? 0 : h + 1;  // window = [h1, h1 + fs)
            __builtin_amdgcn_wave_barrier();
            // block_out_i = sum_k filter_i[k] u_i[k] and the power sum_k u_i[k]^2 of every branch: this lane's taps, then the wave sums
            // (unconditional loads -- rows are padded to 64 KPL taps -- and a select: a load inside the condition compiles to a branch per tap)
            float bmv[NBM][KPL];
#pragma unroll
            for (int i = 0; i < NBM; ++i)
#pragma unroll
                for (int c = 0; c < KPL; ++c) bmv[i][c] = s_bm[(i < nb ? i : 0) * bstride + h1 + lane + 64 * c];
            // the output window WITHOUT its newest element (known only below): taps 0 .. fs - 2
            float lov[KPL];
#pragma unroll
            for (int c = 0; c < KPL; ++c) lov[c] = s_lo[h1 + lane + 64 * c];
#pragma unroll
            for (int c = 0; c < KPL; ++c) {
#pragma unroll
                for (int i = 0; i < NBM; ++i) bmv[i][c] = tap_ok[c] ? bmv[i][c] : 0.f;
                lov[c] = (lane + 64 * c < fs - 1) ? lov[c] : 0.f;
            }
            float out = das, pws[NBM];
#pragma unroll
            for (int i = 0; i < NBM; ++i) {
                float pb = 0.f, pp = 0.f;
#pragma unroll
                for (int c = 0; c < KPL; ++c) {
                    pb = (pb + (freg[i][c] * bmv[i][c]));
                    pp = (pp + (bmv[i][c] * bmv[i][c]));
                }
                pws[i] = 0.f;
                if (i < nb) {  // uniform
                    out = out - gsc_wave_sum(pb);
                    pws[i] = gsc_wave_sum(pp);
                }
            }
            float pl = 0.f;
#pragma unroll
            for (int c = 0; c < KPL; ++c) pl = (pl + (lov[c] * lov[c]));
            const float pwl = gsc_wave_sum(pl);
            const float lop = __fsqrt_rn(__fdiv_rn((pwl + (out * out)), fsz));  // calculate_power(last_outputs)
            if (lane == 0) {
                s_lo[h] = out;
                s_lo[h + fs] = out;
                s_out[jj] = out;
            }
            const bool adapt = ((double)lop < vad_threshold) || !use_vad;  // gsc.cpp:147
            if (adapt && nb > 0) {
                // filter[i][k] += this_mu*out[j]*block_matrix[i][k] (gsc.cpp:163-170).  Lane i works out branch i's step size (two double
                // divisions: once per branch, not once per lane and branch), the products mu_i * out travel by v_readlane
                float mypw = 0.f;
#pragma unroll
                for (int i = 0; i < NBM; ++i) mypw = lane == i ? pws[i] : mypw;
                const float bp = __fsqrt_rn(__fdiv_rn(mypw, fsz));
                float mu;
                if (mu0 * (double)bp / (double)lop < mu_max)  // gsc.cpp:153-157 (double arithmetic: mu0 is a double)
                    mu = (float)(mu0 / (double)lop);
                else
                    mu = (float)(mu0 / (double)bp);
                if (isnan(mu) || isinf(mu)) mu = 0.f;
                const float cvl = (mu * out);
#pragma unroll
                for (int i = 0; i < NBM; ++i)
                    if (i < nb) {  // uniform
                        const float cv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cvl), i));
#pragma unroll
                        for (int c = 0; c < KPL; ++c) {
                            float fv = (freg[i][c] + (cv * bmv[i][c]));
                            if (isnan(fv)) fv = 0.f;
                            freg[i][c] = tap_ok[c] ? fv : 0.f;
                        }
                    }
            }
            __builtin_amdgcn_wave_barrier();
            h = h1;
        }
        if (lane < cnt) ys[n0 + lane] = s_out[lane];
        __builtin_amdgcn_wave_barrier();
    }
    // carried state in the reference's (shifted, oldest-first) order
    for (int e = lane; e < nb * fs; e += 64) {
        const int i = e / fs, k = e - i * fs;
        sv[e] = s_bm[i * bstride + h + k];
    }
#pragma unroll
    for (int i = 0; i < NBM; ++i)
#pragma unroll
        for (int c = 0; c < KPL; ++c)
            if (i < nb && tap_ok[c]) sv[nb * fs + i * fs + lane + 64 * c] = freg[i][c];
    for (int k = lane; k < fs; k += 64) sv[2 * nb * fs + k] = s_lo[h + k];
}

// The same with the blocking branches dealt out to NW wavefronts of one block (branch i belongs to wavefront i mod NW): 256 streams
// of 8 microphones keep 1 024 wavefronts busy instead of 256.  Per sample ONE block barrier, behind the branch sums (every wavefront
// then forms `out`, the output power and the adapt decision from the same numbers, in the same order); each wavefront moves the rings
// of its own branches on (round 4: a second barrier behind ring updates done by wavefront 0: 36.3 ms per 256 x 64 frames).
// Arithmetic and summation order are gsc_nlms_par_kernel's: the results are the same bit for bit.
template <int NW, int NBL, int KPL>  // NBL >= ceil((M - 1) / NW) branches per wavefront, KPL >= ceil(filter_size / 64)
__global__ __launch_bounds__(64 * NW) void gsc_nlms_mw_kernel(const float *aligned, float *y, float *state, long n, int M, int fs,
                                                              int use_vad, double vad_threshold, double mu0, double mu_max) {
#pragma clang fp contract(off)
    extern __shared__ float gl[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = M - 1;
    const int nbr = nb > 0 ? nb : 1;
    const int bstride = (fs + 64 * KPL + 8) | 1;
    float *s_bm = gl;                                // [nb][bstride] mirrored rings of the blocking-matrix inputs
    float *s_lo = s_bm + nbr * bstride;              // [2*fs + 64*KPL + 16] mirrored ring of the outputs
    float *s_d = s_lo + 2 * fs + 64 * KPL + 16;      // [nb][64] neighbour differences of the current tile
    float *s_das = s_d + nbr * 64;                   // [64] upper beamformer of the current tile
    float *s_out = s_das + 64;                       // [64]
    float *s_bo = s_out + 64;                        // [2][16] block_out_i of the current sample (two samples in flight: parity of jj)
    float *s_pw = s_bo + 32;                         // [2][16] window power of branch i; [15]: of the output window
    const int s = blockIdx.x;
    const float *as = aligned + (long)s * M * n;
    float *ys = y + (long)s * n;
    float *sv = state + (long)s * (2 * nb + 1) * fs;
    float freg[NBL][KPL];  // filter taps k = lane + 64 c of this wavefront's branches wv, wv + NW, ...
    bool tap_ok[KPL];
#pragma unroll
    for (int c = 0; c < KPL; ++c) tap_ok[c] = lane + 64 * c < fs;
#pragma unroll
    for (int l = 0; l < NBL; ++l) {
        const int i = wv + NW * l;
#pragma unroll
        for (int c = 0; c < KPL; ++c) {
            const int k = lane + 64 * c;
            freg[l][c] = (i < nb && k < fs) ? sv[nb * fs + i * fs + k] : 0.f;
        }
    }
    for (int e = tid; e < nb * fs; e += 64 * NW) {
        const int i = e / fs, k = e - i * fs;
        const float b = sv[e];
        s_bm[i * bstride + k] = b;
        s_bm[i * bstride + k + fs] = b;
    }
    for (int k = tid; k < fs; k += 64 * NW) {
        const float v = sv[2 * nb * fs + k];
        s_lo[k] = v;
        s_lo[k + fs] = v;
    }
    int h = 0;
    const float fsz = (float)fs;
    for (long n0 = 0; n0 < n; n0 += 64) {
        __syncthreads();  // the previous tile's s_out has been stored; rings initialised
        if (wv == 0) {    // tile prologue, lane = sample: das_out (gsc.cpp:122-127) and the blocking-matrix inputs (gsc.cpp:131)
            const bool ok = n0 + lane < n;
            float prev = ok ? as[n0 + lane] : 0.f, das = 0.f;
            das = (das + prev);
            for (int m = 1; m < M; ++m) {
                const float cur = ok ? as[(long)m * n + n0 + lane] : 0.f;
                das = (das + cur);
                s_d[(m - 1) * 64 + lane] = (cur - prev);
                prev = cur;
            }
            s_das[lane] = __fdiv_rn(das, (float)M);
            __builtin_amdgcn_wave_barrier();
            if (lane < nb) {  // the tile's first sample enters the rings
                const float d = s_d[lane * 64];
                s_bm[lane * bstride + h] = d;
                s_bm[lane * bstride + h + fs] = d;
            }
        }
        __syncthreads();
        const int cnt = (n - n0) < 64 ? (int)(n - n0) : 64;
        for (int jj = 0; jj < cnt; ++jj) {
            const float das = s_das[jj];
            const int h1 = (h + 1 == fs) ? 0 : h + 1;  // window = [h1, h1 + fs)
            float *bo = s_bo + 16 * (jj & 1), *pw = s_pw + 16 * (jj & 1);
            // ---- this wavefront's branches: block_out_i and the window power --------------------------------------------------------
            float bmv[NBL][KPL];
#pragma unroll
            for (int l = 0; l < NBL; ++l) {
                const int i = wv + NW * l, ir = i < nb ? i : 0;
#pragma unroll
                for (int c = 0; c < KPL; ++c) bmv[l][c] = s_bm[ir * bstride + h1 + lane + 64 * c];
            }
            float lov[KPL];
            if (wv == 0) {
#pragma unroll
                for (int c = 0; c < KPL; ++c) lov[c] = s_lo[h1 + lane + 64 * c];
            }
#pragma unroll
            for (int l = 0; l < NBL; ++l) {
                const int i = wv + NW * l;
                float pb = 0.f, pp = 0.f;
#pragma unroll
                for (int c = 0; c < KPL; ++c) {
                    bmv[l][c] = tap_ok[c] ? bmv[l][c] : 0.f;
                    pb = (pb + (freg[l][c] * bmv[l][c]));
                    pp = (pp + (bmv[l][c] * bmv[l][c]));
                }
                if (i < nb) {  // uniform
                    const float sb = gsc_wave_sum(pb), sp = gsc_wave_sum(pp);
                    if (lane == 0) {
                        bo[i] = sb;
                        pw[i] = sp;
                    }
                }
            }
            if (wv == 0) {  // the output window WITHOUT its newest element (known only below): taps 0 .. fs - 2
                float pl = 0.f;
#pragma unroll
                for (int c = 0; c < KPL; ++c) {
                    const float v = (lane + 64 * c < fs - 1) ? lov[c] : 0.f;
                    pl = (pl + (v * v));
                }
                const float sl = gsc_wave_sum(pl);
                if (lane == 0) pw[15] = sl;
            }
            __syncthreads();
            // ---- every wavefront: out, the output power, the adapt decision (same numbers, same order) ---------------------------------
            float out = das;
            for (int i = 0; i < nb; ++i) out = out - bo[i];
            const float lop = __fsqrt_rn(__fdiv_rn((pw[15] + (out * out)), fsz));  // calculate_power(last_outputs)
            const bool adapt = ((double)lop < vad_threshold) || !use_vad;          // gsc.cpp:147
            if (adapt && nb > 0) {
                // lane i works out branch i's step size (gsc.cpp:153-157, double arithmetic); this wavefront's branches take theirs by v_readlane
                const float mypw = lane < nb ? pw[lane] : 0.f;
                const float bp = __fsqrt_rn(__fdiv_rn(mypw, fsz));
                float mu;
                if (mu0 * (double)bp / (double)lop < mu_max)
                    mu = (float)(mu0 / (double)lop);
                else
                    mu = (float)(mu0 / (double)bp);
                if (isnan(mu) || isinf(mu)) mu = 0.f;
                const float cvl = (mu * out);
#pragma unroll
                for (int l = 0; l < NBL; ++l) {
                    const int i = wv + NW * l;
                    if (i < nb) {  // uniform
                        const float cv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cvl), i));
#pragma unroll
                        for (int c = 0; c < KPL; ++c) {  // filter[i][k] += this_mu*out[j]*block_matrix[i][k]  (gsc.cpp:163-170)
                            float fv = (freg[l][c] + (cv * bmv[l][c]));
                            if (isnan(fv)) fv = 0.f;
                            freg[l][c] = tap_ok[c] ? fv : 0.f;
                        }
                    }
                }
            }
            // The rings move on.  The output ring has one reader and one writer, wavefront 0; the ring of branch i is read by the wavefront
            // that owns the branch, and that wavefront feeds it: no second block barrier (LDS operations of one wavefront execute in
            // issue order).  A wavefront that runs ahead writes the next sample's sums into the other half of s_bo / s_pw and meets the
            // others at that sample's barrier.
            if (wv == 0 && lane == 0) {
                s_lo[h] = out;
                s_lo[h + fs] = out;
                s_out[jj] = out;
            }
            if (jj + 1 < cnt && lane < NBL) {
                const int i = wv + NW * lane;
                if (i < nb) {
                    const float d = s_d[i * 64 + jj + 1];
                    s_bm[i * bstride + h1] = d;
                    s_bm[i * bstride + h1 + fs] = d;
                }
            }
            __builtin_amdgcn_wave_barrier();
            h = h1;
        }
        if (wv == 0 && lane < cnt) ys[n0 + lane] = s_out[lane];
    }
    __syncthreads();
    // carried state in the reference's (shifted, oldest-first) order
    for (int e = tid; e < nb * fs; e += 64 * NW) {
        const int i = e / fs, k = e - i * fs;
        sv[e] = s_bm[i * bstride + h + k];
    }
#pragma unroll
    for (int l = 0; l < NBL; ++l) {
        const int i = wv + NW * l;
#pragma unroll
        for (int c = 0; c < KPL; ++c)
            if (i < nb && tap_ok[c]) sv[nb * fs + i * fs + lane + 64 * c] = freg[l][c];
    }
    for (int k = tid; k < fs; k += 64 * NW) sv[2 * nb * fs + k] = s_lo[h + k];
}

// ======================================================================================
//                              gss: geometric source separation
// ======================================================================================
// One group of MP lanes per (stream, problem), lane m owns column m of the demixing matrix
// W_j (S x M) and walks the frames in order (the update is recursive, gss.cpp:136).
// sum over the MP lanes of a group (MP = 4, 8, 16: inside one 16-lane DPP row), every lane gets the total
template <int CTRL>
__device__ __forceinline__ double gss_dpp(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <int MP>
__device__ __forceinline__ double grp_sum(double v) {
    v += gss_dpp<0xB1>(v);                 // quad_perm [1,0,3,2]
    v += gss_dpp<0x4E>(v);                 // quad_perm [2,3,0,1]
    if (MP >= 8) v += gss_dpp<0x141>(v);   // row_half_mirror
    if (MP >= 16) v += gss_dpp<0x140>(v);  // row_mirror
    return v;
}
template <int MP>
__device__ __forceinline__ cd grp_sum(cd v) { return cd{grp_sum<MP>(v.x), grp_sum<MP>(v.y)}; }

template <int MP, int KM>
__global__ __launch_bounds__(256) void gss_kernel(BinsArgs a) {
    constexpr bool kDpp = MP <= 16;  // sums over the group's lanes by DPP butterflies (pairwise order) instead of serial walks over LDS
    constexpr int GPB = 256 / MP;
    __shared__ cd s_x[GPB][MP + 1];   // padded rows: see mvdr_lcmv_kernel (gss 256x256: 6.5 -> 5.8 ms)
    __shared__ cd s_p[GPB][KM][MP + 1];
    const int grp = threadIdx.x / MP, m = threadIdx.x % MP;
    const int gq = blockIdx.x * GPB + grp;
    if (gq >= a.n_streams * kNQ) return;
    const int s = gq / kNQ, q = gq % kNQ;
    const int j = q_bin(q);
    const int M = a.n_mics, NP = (M + 1) >> 1, S = a.kp1;
    f64x2 *yout = a.Yh + ((long)s * a.n_frames) * kYhStride + q;
    const f64x2 *Zs = a.Z + ((long)(s / a.n_dirs) * a.frames_ws + a.frame_off) * NP * kN;
    const f64x2 *steer = a.steer + (long)(s % a.n_dirs) * a.steer_dir_stride;
    const double f = fabs(a.freqs[j]);
    const bool inband = f >= a.cfg.freq_min && f <= a.cfg.freq_max;
    if (!inband) {
        if (m == 0)
            for (long t = 0; t < a.n_frames; ++t) yout[t * kYhStride] = f64x2{0, 0};
        return;
    }
    const int ksrc = q_src_bin(q), kneg = (kN - ksrc) & (kN - 1);
    cd C[KM], W[KM];
    f64x2 *Wg = a.gssW + (((long)s * kN + j) * S) * M;
#pragma unroll
    for (int r = 0; r < KM; ++r) {
        C[r] = (r < S && m < M) ? ld(steer + ((long)r * M + m) * kN + j) : cd{0, 0};
        if ((a.gss_reset_mask >> (s % a.n_dirs)) & 1ull)
            W[r] = conj(C[r]);  // sep_matrix[j] = weights[j].adjoint() (gss.cpp:92)
        else
            W[r] = (r < S && m < M) ? ld(Wg + (long)r * M + m) : cd{0, 0};
    }
    const double mu = a.cfg.mu, keep = 1 - a.cfg.lambda_ * a.cfg.mu;
    const double c2 = (double)(size_t)(2 * (1 / (size_t)S));  // integer arithmetic, quirk Q13
    // this lane's two packed-spectrum values of a step are requested four frames ahead: the recursion leaves a wavefront nothing
    // else to cover the load latency with (256 x 256: see EXPERIMENTS)
    constexpr int kAhead = 4;
    const int mz = m < M ? m : 0;
    cd zr_[kAhead], zcr_[kAhead];
#pragma unroll
    for (int k = 0; k < kAhead; ++k) {
        const f64x2 *Zf = Zs + (k < a.n_frames ? k : a.n_frames - 1) * NP * kN + (mz >> 1) * kN;
        zr_[k] = ld(Zf + ksrc);
        zcr_[k] = ld(Zf + kneg);
    }
    for (long t0 = 0; t0 < a.n_frames; t0 += kAhead) {
#pragma unroll
      for (int kk = 0; kk < kAhead; ++kk) {
        const long t = t0 + kk;
        if (t >= a.n_frames) break;  // uniform
        cd x{0, 0};
        const cd z = zr_[kk], zc = conj(zcr_[kk]);
        {
            const f64x2 *Zn = Zs + (t + kAhead < a.n_frames ? t + kAhead : a.n_frames - 1) * NP * kN + (mz >> 1) * kN;
            zr_[kk] = ld(Zn + ksrc);
            zcr_[kk] = ld(Zn + kneg);
        }
        if (m < M) {
            if ((m & 1) == 0) {
                x = (z + zc) * 0.5;
            } else {
                const cd d = z - zc;
                x = cd{0.5 * d.y, -0.5 * d.x};
            }
            if (q == kQX) x = conj(x);
        }
        s_x[grp][m] = x;
        if (!kDpp) {
#pragma unroll
            for (int r = 0; r < KM; ++r) s_p[grp][r][m] = W[r] * x;
        }
        __builtin_amdgcn_wave_barrier();
        double mag = 0.0, alpha = 0.0;
        if (kDpp) {  // lanes m >= M hold x = 0
            mag = grp_sum<MP>(cabs(x));
            alpha = grp_sum<MP>(norm2(x));
        } else {
            for (int k = 0; k < M; ++k) {
                const cd v = s_x[grp][k];
                mag += cabs(v);
                alpha += norm2(v);
            }
        }
        mag /= (double)((unsigned)M * (unsigned)kN);
        cd y;
        if (mag > a.cfg.freq_mag_threshold) {
            cd yf[KM];
#pragma unroll
            for (int r = 0; r < KM; ++r) {
                if (kDpp) {
                    yf[r] = grp_sum<MP>(W[r] * x);
                } else {
                    cd acc{0, 0};
                    for (int k = 0; k < M; ++k) acc = acc + s_p[grp][r][k];
                    yf[r] = acc;
                }
            }
            y = yf[0];
            alpha *= alpha;
            const double c1 = (double)(4 * (size_t)S) * (1 / alpha);
            cd Ey[KM];
#pragma unroll
            for (int r = 0; r < KM; ++r) {
                cd acc{0, 0};
#pragma unroll
                for (int r2 = 0; r2 < KM; ++r2)
                    if (r2 != r && r < S && r2 < S) acc = acc + (yf[r] * conj(yf[r2])) * yf[r2];
                Ey[r] = acc;
            }
            cd d2[KM];
#pragma unroll
            for (int r = 0; r < KM; ++r) d2[r] = cd{0, 0};
            if (c2 != 0.0) {  // only S == 1: dj2 = 2 (W C - I) C^H
                cd wc{0, 0};
                if (kDpp) {
                    wc = grp_sum<MP>(W[0] * C[0]);
                } else {
                    __builtin_amdgcn_wave_barrier();
                    s_p[grp][0][m] = W[0] * C[0];
                    __builtin_amdgcn_wave_barrier();
                    for (int k = 0; k < M; ++k) wc = wc + s_p[grp][0][k];
                }
                wc.x -= 1.0;
                d2[0] = (wc * conj(C[0])) * c2;
            }
#pragma unroll
            for (int r = 0; r < KM; ++r)
                if (r < S) W[r] = (W[r] * keep) - ((Ey[r] * conj(x)) * c1 + d2[r]) * mu;
        } else {
            y = s_x[grp][0] * 0.01;
        }
        if (m == 0) yout[t * kYhStride] = f64x2{y.x, y.y};
        __builtin_amdgcn_wave_barrier();
      }
    }
#pragma unroll
    for (int r = 0; r < KM; ++r)
        if (r < S && m < M) Wg[(long)r * M + m] = f64x2{W[r].x, W[r].y};
}


// ---- gss, one LANE per (stream, problem): the tuned shapes (<= 8 microphones, <= 4 sources) ------------------------------------------
// The group kernel above spends most of its instructions on DPP sums over the MP lanes of a problem (ten 64-bit group sums per frame).
// Here the whole S x M demixing matrix of a problem lives in one lane's registers (4 x 8 complex = 128 VGPRs: two wavefronts per SIMD),
// every sum is a chain of FMAs inside the lane, the 64 lanes of a wavefront are 64 consecutive problems of one stream (their spectrum rows
// are contiguous: 1 KiB per wave-instruction) and the rows of the next frame travel by global -> LDS DMA while this one is worked on
// (as in mvdr_fast_kernel).  The recursion over the frames (gss.cpp:136) stays serial per problem; 256 streams x 341 in-band problems
// are 1 364 wavefronts, all resident at once.  Sums over the microphones run m = 0, 1, ... (the group kernel: pairwise).
template <int MP, int KM>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void gss_lane_kernel(BinsArgs a) {
    const int lane = threadIdx.x;
    constexpr int wps = (kNQ + 63) / 64;  // wavefronts per stream
    const int s = blockIdx.x / wps;
    int q = (blockIdx.x - s * wps) * 64 + lane;
    const bool live = q < kNQ;
    if (!live) q = kNQ - 1;
    const int j = q_bin(q);
    const int M = a.n_mics, NP = (M + 1) >> 1, S = a.kp1;
    f64x2 *yout = a.Yh + ((long)s * a.n_frames) * kYhStride + q;
    const double f = fabs(a.freqs[j]);
    const bool inband = live && f >= a.cfg.freq_min && f <= a.cfg.freq_max;
    if (__builtin_amdgcn_ballot_w64(inband) == 0) {  // nothing to separate in this wavefront (gss.cpp:150: y_fft = 0 out of band)
        if (live)
            for (long t = 0; t < a.n_frames; ++t) yout[t * kYhStride] = f64x2{0, 0};
        return;
    }
    // frame 0 of this stream's spectra (uniform) + a per-lane byte offset; rows 2p / 2p + 1 of a buffer = Z_t[p][k] / Z_t[p][N - k]
    const char *Zu = reinterpret_cast<const char *>(a.Z + ((long)(s / a.n_dirs) * a.frames_ws + a.frame_off) * NP * kN);
    const int ksrc = q_src_bin(q), kneg = (kN - ksrc) & (kN - 1);
    const unsigned vk = (unsigned)ksrc * 16u, vn = (unsigned)kneg * 16u;
    const long frame_bytes = (long)NP * kN * 16;
    __shared__ __attribute__((aligned(16))) f64x2 s_pf[2][MP][64];
    auto dma_frame = [&](long t, int buf) {
        const char *b = Zu + t * frame_bytes;
#pragma unroll
        for (int p = 0; p < MP / 2; ++p)
            if (p < NP) {  // uniform
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(b + (long)p * kN * 16 + vk),
                                                 (__attribute__((address_space(3))) void *)&s_pf[buf][2 * p][0], 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(b + (long)p * kN * 16 + vn),
                                                 (__attribute__((address_space(3))) void *)&s_pf[buf][2 * p + 1][0], 16, 0, 0);
            }
    };
    const f64x2 *steer = a.steer + (long)(s % a.n_dirs) * a.steer_dir_stride;
    f64x2 *Wg = a.gssW + (((long)s * kN + j) * S) * M;
    cd W[KM][MP], C0[MP];
    const bool reset = ((a.gss_reset_mask >> (s % a.n_dirs)) & 1ull) != 0;
#pragma unroll
    for (int r = 0; r < KM; ++r)
#pragma unroll
        for (int m = 0; m < MP; ++m) {
            cd w{0, 0};
            if (r < S && m < M) w = reset ? conj(ld(steer + ((long)r * M + m) * kN + j)) : ld(Wg + (long)r * M + m);  // sep_matrix[j] = weights[j].adjoint() (gss.cpp:92)
            W[r][m] = w;
        }
#pragma unroll
    for (int m = 0; m < MP; ++m) C0[m] = (KM == 1 && m < M) ? ld(steer + (long)m * kN + j) : cd{0, 0};  // only S == 1 reads the constraint in the loop
    const double mu = a.cfg.mu, keep = 1 - a.cfg.lambda_ * a.cfg.mu;
    const double c2 = (double)(size_t)(2 * (1 / (size_t)S));  // integer arithmetic, quirk Q13
    const float thr32 = (float)(a.cfg.freq_mag_threshold * (double)((unsigned)M * (unsigned)kN));
    int pb = 0;
    if (a.n_frames > 0) dma_frame(0, pb);
    cd y_prev{0, 0};  // frame t - 1's output: stored one iteration late (behind the wait, in front of the next DMA), so that the wait at the
                      // top of an iteration never waits for a store issued a few instructions earlier
    for (long t = 0; t < a.n_frames; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // frame t has landed in s_pf[pb]
        __builtin_amdgcn_wave_barrier();
        if (t > 0 && live) yout[(t - 1) * kYhStride] = f64x2{y_prev.x, y_prev.y};
        if (t + 1 < a.n_frames) dma_frame(t + 1, pb ^ 1);
        cd X[MP];
#pragma unroll
        for (int p = 0; p < MP / 2; ++p) {
            const cd z = ld(&s_pf[pb][2 * p][lane]), zc = conj(ld(&s_pf[pb][2 * p + 1][lane]));
            const cd d = z - zc;
            X[2 * p] = (2 * p < M) ? (z + zc) * 0.5 : cd{0, 0};
            X[2 * p + 1] = (2 * p + 1 < M) ? cd{0.5 * d.y, -0.5 * d.x} : cd{0, 0};  // (an odd count's partner channel is rounding residue, not zero)
        }
        if (q == kQX) {
#pragma unroll
            for (int m = 0; m < MP; ++m) X[m].y = -X[m].y;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the rows have been read: the DMA after next may overwrite them
        // magnitude gate (gss.cpp:118): decided in fp32 unless the fp32 sum is within 1e-4 of the threshold
        float m32 = 0.f;
#pragma unroll
        for (int m = 0; m < MP; ++m) m32 += __builtin_amdgcn_sqrtf((float)norm2(X[m]));
        bool open = m32 > thr32;
        if (__builtin_amdgcn_ballot_w64(__builtin_fabsf(m32 - thr32) <= 1e-4f * thr32) != 0) {
            double mag = 0.0;
#pragma unroll
            for (int m = 0; m < MP; ++m) mag += cabs(X[m]);
            mag /= (double)((unsigned)M * (unsigned)kN);
            open = mag > a.cfg.freq_mag_threshold;
        }
        open = open && inband;
        cd y = X[0] * 0.01;  // gate closed (gss.cpp:152)
        if (__builtin_amdgcn_ballot_w64(open) != 0) {
            double alpha = 0.0;
#pragma unroll
            for (int m = 0; m < MP; ++m) alpha += norm2(X[m]);
            cd yf[KM];
#pragma unroll
            for (int r = 0; r < KM; ++r) {
                cd acc{0, 0};
#pragma unroll
                for (int m = 0; m < MP; ++m) acc = acc + W[r][m] * X[m];
                yf[r] = acc;
            }
            alpha *= alpha;
            // a lane whose gate is closed keeps its matrix: W * 1 - (.. * 0 + d2) * 0 is W bit for bit -- three selects per frame instead of
            // four per matrix entry
            const double c1 = open ? (double)(4 * (size_t)S) * (1 / alpha) : 0.0;
            const double keep_l = open ? keep : 1.0, mu_l = open ? mu : 0.0;
            cd Ey[KM];
#pragma unroll
            for (int r = 0; r < KM; ++r) {
                cd acc{0, 0};
#pragma unroll
                for (int r2 = 0; r2 < KM; ++r2)
                    if (r2 != r && r < S && r2 < S) acc = acc + (yf[r] * conj(yf[r2])) * yf[r2];
                Ey[r] = acc;
            }
            cd wc{0, 0};
            if (KM == 1 && c2 != 0.0) {  // only S == 1: dj2 = 2 (W C - I) C^H
#pragma unroll
                for (int m = 0; m < MP; ++m) wc = wc + W[0][m] * C0[m];
                wc.x -= 1.0;
            }
#pragma unroll
            for (int r = 0; r < KM; ++r)
                if (r < S) {  // uniform
#pragma unroll
                    for (int m = 0; m < MP; ++m) {
                        cd d2{0, 0};
                        if (KM == 1 && r == 0 && c2 != 0.0) d2 = (wc * conj(C0[m])) * c2;
                        W[r][m] = (W[r][m] * keep_l) - ((Ey[r] * conj(X[m])) * c1 + d2) * mu_l;
                    }
                }
            if (open) y = yf[0];
        }
        if (!inband) y = cd{0, 0};
        y_prev = y;
        pb ^= 1;
    }
    if (live && a.n_frames > 0) yout[(a.n_frames - 1) * kYhStride] = f64x2{y_prev.x, y_prev.y};
#pragma unroll
    for (int r = 0; r < KM; ++r)
#pragma unroll
        for (int m = 0; m < MP; ++m)
            if (inband && r < S && m < M) Wg[(long)r * M + m] = f64x2{W[r][m].x, W[r][m].y};
}

}  // namespace

hipError_t launch_gss(const BinsArgs &a, int n_cus, hipStream_t s) {
    const int M = a.n_mics, km = a.kp1 <= 1 ? 1 : 4;
    const int groups = a.n_streams * kNQ;
    // One lane per problem pays once the lanes fill the chip: 256 streams x 256 frames 3.33 -> 2.10 ms (4 microphones 1.76 -> 1.24), but ONE
    // stream of 65 536 frames 69 -> 145 ms (9 wavefronts; a frame step is 2.2 us in one lane, 1.06 us spread over 8): lane kernel from two
    // wavefronts per CU on.  BF_GSS_GROUP=1 / 0 force the group / the lane kernel (tests, A/B).
    static const int group_env = getenv("BF_GSS_GROUP") ? atoi(getenv("BF_GSS_GROUP")) : -1;
    const bool lane_kernel = group_env >= 0 ? group_env == 0 : (long)a.n_streams * ((kNQ + 63) / 64) >= 2L * n_cus;
#define BF_LAUNCH_GSS(MP_, KM_) \
    BF_LAUNCH((gss_kernel<MP_, KM_>), dim3((groups + (256 / MP_) - 1) / (256 / MP_)), dim3(256), 0, s, a)
    if (a.kp1 > 4 || M > 16) {  // beyond the tuned shapes: more interferers (up to 15) or microphones (up to 32)
        if (a.kp1 > 16 || M > 32) return hipErrorInvalidValue;
        if (a.kp1 <= 1) BF_LAUNCH_GSS(32, 1);
        else if (a.kp1 <= 4) BF_LAUNCH_GSS(32, 4);
        else if (a.kp1 <= 8) { if (M <= 8) BF_LAUNCH_GSS(8, 8); else if (M <= 16) BF_LAUNCH_GSS(16, 8); else BF_LAUNCH_GSS(32, 8); }
        else { if (M <= 16) BF_LAUNCH_GSS(16, 16); else BF_LAUNCH_GSS(32, 16); }
    } else if (M <= 8 && lane_kernel) {  // one lane per problem
        const dim3 grid((unsigned)(a.n_streams * ((kNQ + 63) / 64)));
        if (M <= 4) {
            if (km == 1) BF_LAUNCH((gss_lane_kernel<4, 1>), grid, dim3(64), 0, s, a); else BF_LAUNCH((gss_lane_kernel<4, 4>), grid, dim3(64), 0, s, a);
        } else {
            if (km == 1) BF_LAUNCH((gss_lane_kernel<8, 1>), grid, dim3(64), 0, s, a); else BF_LAUNCH((gss_lane_kernel<8, 4>), grid, dim3(64), 0, s, a);
        }
    } else if (M <= 4) {
        if (km == 1) BF_LAUNCH_GSS(4, 1); else BF_LAUNCH_GSS(4, 4);
    } else if (M <= 8) {
        if (km == 1) BF_LAUNCH_GSS(8, 1); else BF_LAUNCH_GSS(8, 4);
    } else {
        if (km == 1) BF_LAUNCH_GSS(16, 1); else BF_LAUNCH_GSS(16, 4);
    }
#undef BF_LAUNCH_GSS
    return hipGetLastError();
}

hipError_t launch_gsc_nlms(const float *aligned, float *y, float *state, long n_samples, int n_streams, int n_mics,
                           const bf_config &cfg, hipStream_t s) {
    const int fs = cfg.gsc_filter_size, nb = n_mics - 1, nbr = nb > 0 ? nb : 1;
    const int kpl = (fs + 63) / 64, kp = kpl <= 1 ? 1 : kpl <= 2 ? 2 : 4;
    // BF_GSC_SERIAL=1: the sums in the reference's tap order, one branch per lane (gsc_nlms_kernel); default: taps over the lanes, the
    // branches dealt out to 8 wavefronts per stream from five branches on, 4 from three, 2 at two, one branch: gsc_nlms_par_kernel
    // (8 microphones, 256 streams x 64 frames: 52.6 / 54.2 / 43.1 / 36.8 ms at 1 / 2 / 4 / 8 wavefronts)
    static const bool serial = getenv("BF_GSC_SERIAL") && atoi(getenv("BF_GSC_SERIAL")) == 1;
    const int nw = serial ? 1 : (nb >= 5 ? 8 : nb >= 3 ? 4 : nb >= 2 ? 2 : 1);
    const size_t lds_serial = sizeof(float) * ((size_t)nbr * ((fs + 64 * kp + 8) | 1) + (size_t)nbr * ((64 * kp + 8) | 1) + 2 * fs + 16 +
                                               (size_t)nbr * 64 + 64 + 16 + 64);
    const size_t lds_par = sizeof(float) * ((size_t)nbr * ((fs + 64 * kp + 8) | 1) + 2 * fs + 64 * kp + 16 + (size_t)nbr * 64 + 64 + 64 + 64);
    const size_t lds = serial ? lds_serial : lds_par;
    if (nw > 1 && nb <= 15) {
        const int nbl = (nb + nw - 1) / nw;  // <= 4 at nw = 4, <= 8 at nw = 2
#define BF_MW(NW_, NBL_, KPL_)                                                                                                     \
    BF_LAUNCH((gsc_nlms_mw_kernel<NW_, NBL_, KPL_>), dim3((unsigned)n_streams), dim3(64 * NW_), lds, s, aligned, y, state, \
                       n_samples, n_mics, fs, cfg.gsc_use_vad, cfg.gsc_vad_threshold, cfg.gsc_mu0, cfg.gsc_mu_max)
#define BF_MW_K(NW_, NBL_)                     \
    do {                                       \
        if (kpl <= 1) BF_MW(NW_, NBL_, 1);     \
        else if (kpl <= 2) BF_MW(NW_, NBL_, 2);\
        else BF_MW(NW_, NBL_, 4);              \
    } while (0)
        if (nw == 8) {
            if (nbl <= 1) BF_MW_K(8, 1); else BF_MW_K(8, 2);
        } else if (nw == 4) {
            if (nbl <= 1) BF_MW_K(4, 1); else if (nbl <= 2) BF_MW_K(4, 2); else BF_MW_K(4, 4);
        } else {
            if (nbl <= 1) BF_MW_K(2, 1); else if (nbl <= 2) BF_MW_K(2, 2); else if (nbl <= 4) BF_MW_K(2, 4); else BF_MW_K(2, 8);
        }
#undef BF_MW_K
#undef BF_MW
        return hipGetLastError();
    }
#define BF_NLMS(NBM_, KPL_)                                                                                                        \
    do {                                                                                                                            \
        if (serial)                                                                                                                 \
            BF_LAUNCH((gsc_nlms_kernel<NBM_, KPL_>), dim3((unsigned)n_streams), dim3(64), lds, s, aligned, y, state,       \
                               n_samples, n_mics, fs, cfg.gsc_use_vad, cfg.gsc_vad_threshold, cfg.gsc_mu0, cfg.gsc_mu_max);         \
        else                                                                                                                        \
            BF_LAUNCH((gsc_nlms_par_kernel<NBM_, KPL_>), dim3((unsigned)n_streams), dim3(64), lds, s, aligned, y, state,   \
                               n_samples, n_mics, fs, cfg.gsc_use_vad, cfg.gsc_vad_threshold, cfg.gsc_mu0, cfg.gsc_mu_max);         \
    } while (0)
#define BF_NLMS_K(NBM_)                     \
    do {                                    \
        if (kpl <= 1) BF_NLMS(NBM_, 1);     \
        else if (kpl <= 2) BF_NLMS(NBM_, 2);\
        else BF_NLMS(NBM_, 4);              \
    } while (0)
    if (nb <= 1) BF_NLMS_K(1);
    else if (nb <= 3) BF_NLMS_K(3);
    else if (nb <= 7) BF_NLMS_K(7);
    else BF_NLMS_K(15);
#undef BF_NLMS_K
#undef BF_NLMS
    return hipGetLastError();
}

hipError_t launch_gsc_align(const BinsArgs &a, hipStream_t s) {
    const long total = (long)a.n_streams * a.n_frames * kNQ;
    BF_LAUNCH(gsc_align_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace BF_NTAG
}  // namespace bf
