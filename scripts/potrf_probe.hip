// Debug probe: in-kernel cycle stamps of k_potrf_diag (load / factor / inverse+store) and the clock it runs at.
#define GPBO_STAMPS 1
#include "../bayesian_optimization_amd/csrc/kernels.hpp"
#include <cstdio>
#include <vector>
using namespace gpbo;
int run(int defer) {
    const int threads = POTRF_MFMA_THREADS;
    const int n = 128;
    std::vector<double> A(n * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) A[i * n + j] = exp(-0.5 * (i - j) * (i - j) / 400.0) + (i == j ? 0.01 : 0.0);
    double *dA, *dinv; int* info;
    (void)hipMalloc(&dA, n * n * 8); (void)hipMalloc(&dinv, n * n * 8); (void)hipMalloc(&info, 64);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
        (void)hipMemset(info, 0, 64);
        { long long z[64] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)); }
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        k_potrf_diag<<<1, threads>>>(dA, n, 0, dinv, info, BatchStride{}, defer, n);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<double> L(n * n), W(n * n);
        (void)hipMemcpy(L.data(), dA, n * n * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(W.data(), dinv, n * n * 8, hipMemcpyDeviceToHost);
        double e1m = 0, e2m = 0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double s1 = 0, s2 = 0;
                for (int k = 0; k < n; ++k) { s1 += L[i * n + k] * L[j * n + k]; s2 += W[i * n + k] * L[k * n + j]; }
                e1m = fmax(e1m, fabs(s1 - A[i * n + j])); if (!defer) e2m = fmax(e2m, fabs(s2 - (i == j)));
            }
        { long long z[64]; (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_stamps), sizeof(z)); printf("ticks(100MHz): load %lld potrf16 %lld trsm %lld update %lld storeL %lld inverse %lld storeW %lld\n", z[4], z[1], z[2], z[3], z[6], z[7], z[9]); }
        printf("event %.1f us  |LL^T - A| %.2e  |WL - I| %.2e\n", ms * 1e3, e1m, e2m);
    }
    return 0;
}
int main() { for (int defer : {0, 1}) { printf("--- defer = %d (1: factor + tile inverses only, as inside the blocked factorisation)\n", defer); run(defer); } return 0; }
