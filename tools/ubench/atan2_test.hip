// Device check of atan2_fast (beamform_amd/csrc/bins_common.hpp) against the library atan2 and against a long-double host value.
#define BF_NFFT 1024
#include "../../beamform_amd/csrc/bins_common.hpp"
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
using namespace bf::n1024;
__global__ void k(const double *y, const double *x, double *fast, double *lib, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    fast[i] = atan2_fast(y[i], x[i]);
    lib[i] = atan2(y[i], x[i]);
}
int main() {
    const long n = 1 << 24;
    std::vector<double> y(n), x(n), f(n), l(n);
    std::mt19937_64 g(5);
    std::uniform_real_distribution<double> u(-1.0, 1.0), e(-12.0, 6.0);
    for (long i = 0; i < n; ++i) {
        const double sc = std::pow(10.0, e(g));
        y[i] = u(g) * sc;
        x[i] = u(g) * ((i & 7) == 0 ? sc * std::pow(10.0, e(g) / 3) : sc);
        if (i < 64) {  // axes, diagonals, octant boundaries, zeros
            const double v[8] = {0.0, -0.0, 1.0, -1.0, 0.25, 0.75, -0.25, 3.0};
            y[i] = v[i & 7];
            x[i] = v[(i >> 3) & 7];
        }
    }
    double *dy, *dx, *df, *dl;
    (void)hipMalloc(&dy, n * 8); (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&df, n * 8); (void)hipMalloc(&dl, n * 8);
    (void)hipMemcpy(dy, y.data(), n * 8, hipMemcpyHostToDevice); (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dy, dx, df, dl, n);
    (void)hipMemcpy(f.data(), df, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(l.data(), dl, n * 8, hipMemcpyDeviceToHost);
    double worst_lib = 0, worst_true = 0, worst_lib_true = 0; long bad = 0;
    for (long i = 0; i < n; ++i) {
        const long double t = atan2l((long double)y[i], (long double)x[i]);
        const double ulp = std::fabs(l[i]) > 0 ? std::nextafter(std::fabs(l[i]), 1e300) - std::fabs(l[i]) : 4.9e-324;
        const double d1 = std::fabs(f[i] - l[i]) / ulp, d2 = (double)(fabsl((long double)f[i] - t) / ulp), d3 = (double)(fabsl((long double)l[i] - t) / ulp);
        if (d1 > worst_lib) worst_lib = d1;
        if (d2 > worst_true) worst_true = d2;
        if (d3 > worst_lib_true) worst_lib_true = d3;
        if (std::signbit(f[i]) != std::signbit(l[i]) || std::isnan(f[i]) != std::isnan(l[i])) ++bad;
    }
    printf("n = %ld: max |fast - lib| = %.2f ulp, max |fast - true| = %.2f ulp, max |lib - true| = %.2f ulp, sign/nan mismatches %ld\n", n, worst_lib, worst_true, worst_lib_true, bad);
    return 0;
}
