// Device unit test of the cross-lane pieces of das_fused_w64.hip (row transpose, T1, forward FFT stage by stage, round trip).
#include "../../beamform_amd/csrc/das_fused_w64.hip"
#include <cstdio>
#include <complex>
#include <vector>
using namespace bf;
__global__ void k_rows(float *out) {
    int lane = threadIdx.x;
    float r0 = lane * 10 + 0, r1 = lane * 10 + 1, r2 = lane * 10 + 2, r3 = lane * 10 + 3;
    row_transpose4(r0, r1, r2, r3);
    out[lane * 4 + 0] = r0; out[lane * 4 + 1] = r1; out[lane * 4 + 2] = r2; out[lane * 4 + 3] = r3;
}
__global__ void k_fft(const float2 *in, float2 *out, const f32x2 *twg, int stage) {
    __shared__ __attribute__((aligned(16))) float lds[kLdsTw + kPlane];
    for (int i = threadIdx.x; i < kLdsTw; i += 64) lds[i] = reinterpret_cast<const float *>(twg)[i];
    __syncthreads();
    const cx<float> *tw1 = reinterpret_cast<const cx<float> *>(lds);
    const cx<float> *tw2 = tw1 + 1024;
    float *pl = lds + kLdsTw;
    int lane = threadIdx.x;
    float *row16 = pl + (lane & 15) * kRS + 16 * (lane >> 4);
    float *wcol = pl + w64_col(lane);
    float re[16], im[16];
    for (int j = 0; j < 16; ++j) { re[j] = in[64 * j + lane].x; im[j] = in[64 * j + lane].y; }
    w64_fwd_p1<float>(re, im, lane, tw1);
    w64_T1_fwd(re, im, wcol, row16);
    w64_fwd_p2<float>(re, im, lane, tw2);
    w64_T2<true>(re, im);
    w64_fwd_p3<float>(re, im);
    if (stage == 0)
        for (int r = 0; r < 16; ++r) out[lane * 16 + r] = float2{re[r], im[r]};
    w64_inv_p3<float>(re, im);
    w64_T2<false>(re, im);
    w64_inv_p2<float>(re, im, lane, tw2);
    w64_T1_inv(re, im, row16, wcol);
    w64_inv_p1<float>(re, im, lane, tw1);
    if (stage == 1)
        for (int j = 0; j < 16; ++j) out[64 * j + lane] = float2{re[j] / 1024.f, im[j] / 1024.f};
}
int main() {
    float *d; (void)hipMalloc(&d, 1024); float h[256];
    hipLaunchKernelGGL(k_rows, dim3(1), dim3(64), 0, 0, d); (void)hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;  // register c of row q must hold what register q of row c held
    for (int l = 0; l < 64; ++l) for (int c = 0; c < 4; ++c) { float exp = ((l & 15) | (c << 4)) * 10 + (l >> 4); if (h[l * 4 + c] != exp) bad++; }
    printf("row_transpose4 mismatches: %d\n", bad);
    std::vector<float2> x(1024); for (int i = 0; i < 1024; ++i) x[i] = float2{(float)sin(0.37 * i) + 0.1f * (i % 7), (float)cos(0.11 * i * i)};
    std::vector<f32x2> tw = twiddle_table_w64();
    float2 *dx, *dy; f32x2 *dt; (void)hipMalloc(&dx, 8192); (void)hipMalloc(&dy, 8192); (void)hipMalloc(&dt, tw.size() * 8);
    (void)hipMemcpy(dx, x.data(), 8192, hipMemcpyHostToDevice); (void)hipMemcpy(dt, tw.data(), tw.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_fft, dim3(1), dim3(64), 0, 0, dx, dy, dt, 0);
    std::vector<float2> y(1024); (void)hipMemcpy(y.data(), dy, 8192, hipMemcpyDeviceToHost);
    double num = 0, den = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
        int k = w64_bin(l, r); std::complex<double> acc = 0;
        for (int n = 0; n < 1024; ++n) acc += std::complex<double>(x[n].x, x[n].y) * std::polar(1.0, -2 * M_PI * (double)((long)n * k % 1024) / 1024.0);
        std::complex<double> g(y[l * 16 + r].x, y[l * 16 + r].y);
        num += std::norm(g - acc); den += std::norm(acc);
    }
    printf("forward fft rel err: %.3e\n", sqrt(num / den));
    hipLaunchKernelGGL(k_fft, dim3(1), dim3(64), 0, 0, dx, dy, dt, 1);
    (void)hipMemcpy(y.data(), dy, 8192, hipMemcpyDeviceToHost);
    num = den = 0;
    for (int n = 0; n < 1024; ++n) { num += (y[n].x - x[n].x) * (y[n].x - x[n].x) + (y[n].y - x[n].y) * (y[n].y - x[n].y); den += x[n].x * x[n].x + x[n].y * x[n].y; }
    printf("forward + backward round trip rel err: %.3e\n", sqrt(num / den));
    return 0;
}
