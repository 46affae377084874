// Probe: rate of the Cholesky trailing update (C -= P P^T on lower tiles) for several launch shapes.
#include "../bayesian_optimization_amd/csrc/kernels.hpp"
#include <cstdio>
#include <vector>
using namespace gpbo;

// the square grid whose upper half exits at once (the form the library used first)
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_syrk_square(double* __restrict__ A, long ld, int ib0, int jb0, int kcol0, int klen) {
    const int ib = ib0 + blockIdx.y, jb = jb0 + blockIdx.x;
    if (jb > ib) return;
    __shared__ double lds[GEMM_LDS_DOUBLES];
    const double* Pi = A + (long)ib * PB * ld + kcol0;
    const double* Pj = A + (long)jb * PB * ld + kcol0;
    f64x4 acc[4][4];
    acc_zero(acc);
    gemm_tile_128<true, true>(Pi, ld, Pj, ld, 0, klen, acc, lds);
    double* C = A + (long)ib * PB * ld + (long)jb * PB;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double* c = C + (long)acc_row(ta, r) * ld + acc_col(tb);
                *c -= acc[ta][tb][r];
            }
}

// same without the read-modify-write (stores only)
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_syrk_norw(double* __restrict__ A, long ld, int ib0, int jb0, int kcol0, int klen, double* sink) {
    const int ib = ib0 + blockIdx.y, jb = jb0 + blockIdx.x;
    if (jb > ib) return;
    __shared__ double lds[GEMM_LDS_DOUBLES];
    const double* Pi = A + (long)ib * PB * ld + kcol0;
    const double* Pj = A + (long)jb * PB * ld + kcol0;
    f64x4 acc[4][4];
    acc_zero(acc);
    gemm_tile_128<true, true>(Pi, ld, Pj, ld, 0, klen, acc, lds);
    double s = 0;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[ta][tb][r];
    if (s == 1.2345e300) *sink = s;
}

int main() {
    const long n = 16384, ld = n;
    double* A; double* sink;
    (void)hipMalloc(&A, n * n * 8); (void)hipMalloc(&sink, 8);
    (void)hipMemset(A, 0, n * n * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto report = [&](const char* name, int T, int klen, float ms) {
        const double tiles = 0.5 * T * (T + 1);
        printf("%-28s T=%3d K=%4d  %.3f ms  %.1f TFLOP/s\n", name, T, klen, ms, tiles * 128 * 128 * klen * 2 / (ms * 1e-3) / 1e12);
    };
    for (int klen : {512, 1024, 2048}) {
        const int kb = klen / 128;
        for (int T : {124, 64, 32}) {
            const int ib0 = 128 - T;
            float ms;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                k_syrk_square<<<dim3(T, T), GEMM_THREADS>>>(A, ld, ib0, ib0, 0, klen);
                (void)hipEventRecord(e1); (void)hipDeviceSynchronize(); (void)hipEventElapsedTime(&ms, e0, e1);
            }
            report("2-D square grid", T, klen, ms);
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                k_syrk_update<true><<<dim3(T * (T + 1) / 2), GEMM_THREADS>>>(A, ld, ib0, ib0, 0, klen, BatchStride{});
                (void)hipEventRecord(e1); (void)hipDeviceSynchronize(); (void)hipEventElapsedTime(&ms, e0, e1);
            }
            report("1-D lower tiles (library)", T, klen, ms);
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                k_syrk_norw<<<dim3(T, T), GEMM_THREADS>>>(A, ld, ib0, ib0, 0, klen, sink);
                (void)hipEventRecord(e1); (void)hipDeviceSynchronize(); (void)hipEventElapsedTime(&ms, e0, e1);
            }
            report("2-D grid, no C traffic", T, klen, ms);
            (void)kb;
        }
    }
    return 0;
}
