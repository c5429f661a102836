// accuracy of v_rsq_f64 followed by 0 / 1 / 2 Newton steps against 1 / sqrt(x) in double (correctly rounded division of the correctly rounded root)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *x, double *e0, double *e1, double *e2, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    const double ref = 1.0 / sqrt(v);
    double y = __builtin_amdgcn_rsq(v);
    e0[i] = fabs(y - ref) / ref;
    const double hx = 0.5 * v;
    y = fma(y, fma(-hx * y, y, 0.5), y);
    e1[i] = fabs(y - ref) / ref;
    y = fma(y, fma(-hx * y, y, 0.5), y);
    e2[i] = fabs(y - ref) / ref;
}
int main() {
    const int n = 1 << 22;
    std::vector<double> h(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) / 9007199254740992.0;
        h[i] = std::pow(10.0, -12.0 + 24.0 * u) * (1.0 + u);
    }
    double *x, *e0, *e1, *e2;
    hipMalloc(&x, n * 8); hipMalloc(&e0, n * 8); hipMalloc(&e1, n * 8); hipMalloc(&e2, n * 8);
    hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, e0, e1, e2, n);
    std::vector<double> a(n), b(n), c(n);
    hipMemcpy(a.data(), e0, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), e1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), e2, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0;
    for (int i = 0; i < n; ++i) { m0 = std::fmax(m0, a[i]); m1 = std::fmax(m1, b[i]); m2 = std::fmax(m2, c[i]); }
    printf("max relative error over %d values spanning 24 decades: v_rsq_f64 %.3e, +1 Newton step %.3e, +2 steps %.3e (eps = 1.1e-16)\n", n, m0, m1, m2);
    return 0;
}
