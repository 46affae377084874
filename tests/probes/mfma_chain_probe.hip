// Is an element of the fp64 MFMA tile product the k-ordered chain of fused multiply-adds?  A 64 x 64 tile of A B (k = 16 ... 512)
// on gemm_tile_64x64 against v = fma(A[i][k], B[k][j], v), k ascending from v = 0, bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tests/probes/mfma_chain_probe tests/probes/mfma_chain_probe.hip && tests/probes/mfma_chain_probe   (build() does the first half)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "../../bayesian_optimization_amd/csrc/kernels.hpp"
using namespace gpbo;

__global__ __launch_bounds__(GEMM_THREADS) void k_tile(const double* A, long lda, const double* B, long ldb, int kpad, double* C) {
    __shared__ double lds[QUARTER_LDS_DOUBLES];
    f64x4 acc[2][2];
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
    gemm_tile_64x64<false>(A, lda, B, ldb, 0, kpad, acc, lds);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int ta = 0; ta < 2; ++ta)
        for (int r = 0; r < 4; ++r)
            for (int tb = 0; tb < 2; ++tb)
                C[((w >> 1) * 32 + ta * 16 + (lane >> 4) + 4 * r) * 64 + (w & 1) * 32 + tb * 16 + (lane & 15)] = acc[ta][tb][r];
}

int main() {
    std::mt19937_64 rng(11);
    std::normal_distribution<double> g(0.0, 1.0);
    long bad = 0, total = 0;
    for (int kpad : {16, 48, 224, 512}) {
        std::vector<double> A(64 * kpad), B((size_t)kpad * 64), C(64 * 64);
        for (auto& x : A) x = g(rng) * std::exp(3 * g(rng));
        for (auto& x : B) x = g(rng);
        double *dA, *dB, *dC;
        if (hipMalloc((void**)&dA, A.size() * 8) != hipSuccess || hipMalloc((void**)&dB, B.size() * 8) != hipSuccess || hipMalloc((void**)&dC, C.size() * 8) != hipSuccess) return 2;
        (void)hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
        (void)hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
        k_tile<<<1, GEMM_THREADS>>>(dA, kpad, dB, 64, kpad, dC);
        if (hipMemcpy(C.data(), dC, C.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        for (int i = 0; i < 64; ++i)
            for (int j = 0; j < 64; ++j) {
                double v = 0.0;
                for (int k = 0; k < kpad; ++k) v = std::fma(A[(size_t)i * kpad + k], B[(size_t)k * 64 + j], v);
                ++total;
                if (std::memcmp(&v, &C[i * 64 + j], 8) != 0) {
                    if (bad < 5) std::printf("k = %d (%d, %d): chain %.17g tile %.17g\n", kpad, i, j, v, C[i * 64 + j]);
                    ++bad;
                }
            }
        (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
    }
    std::printf("elements %ld  differing %ld\n", total, bad);
    return bad ? 1 : 0;
}
