// exp_neg (csrc/kernels.hpp) against the device library's exp, bit for bit, over 2^25 arguments in [-1100, 0] (the range
// the covariance functions and the normal density use), every exponent boundary region included, and the special values.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tests/probes/exp_neg_probe tests/probes/exp_neg_probe.hip && tests/probes/exp_neg_probe   (build() does the first half)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "../../bayesian_optimization_amd/csrc/kernels.hpp"

__global__ void k_both(const double* x, double* a, double* b, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        a[i] = exp(x[i]);
        b[i] = gpbo::exp_neg(x[i]);
    }
}

int main() {
    const long n = 1L << 25;
    std::vector<double> x(n);
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> wide(-1100.0, 0.0), tail(-760.0, -700.0), small(-1e-3, 0.0);
    for (long i = 0; i < n; ++i) x[i] = (i % 4 == 0) ? tail(rng) : (i % 4 == 1 ? small(rng) : wide(rng));
    const double special[] = {0.0, -0.0, -INFINITY, NAN, -1075.0, -1074.9999, -745.2, -745.13321910194122, -708.4, -709.78, -1e-300, -5e-324, -1e308};
    for (size_t i = 0; i < sizeof(special) / sizeof(double); ++i) x[i] = special[i];
    double *dx, *da, *db;
    if (hipMalloc((void**)&dx, n * 8) != hipSuccess || hipMalloc((void**)&da, n * 8) != hipSuccess || hipMalloc((void**)&db, n * 8) != hipSuccess) return 2;
    (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k_both<<<(unsigned)((n + 255) / 256), 256>>>(dx, da, db, n);
    std::vector<double> a(n), b(n);
    if (hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    long bad = 0;
    for (long i = 0; i < n; ++i)
        if (std::memcmp(&a[i], &b[i], 8) != 0 && !(std::isnan(a[i]) && std::isnan(b[i]))) {
            if (bad < 10) std::printf("x = %.17g: exp %.17g exp_neg %.17g\n", x[i], a[i], b[i]);
            ++bad;
        }
    std::printf("arguments %ld  differing %ld\n", n, bad);
    return bad ? 1 : 0;
}
