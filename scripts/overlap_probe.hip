// Probe: duration of the diagonal-block kernel on a high-priority stream while a trailing update runs on
// (a) an unmasked stream, (b) a stream whose CU mask leaves out 1 CU, (c) 8 CUs, (d) 32 CUs.
#include "../bayesian_optimization_amd/csrc/kernels.hpp"
#include <cstdio>
#include <vector>
using namespace gpbo;
int main() {
    const long n = 16384, ld = n;
    double *A, *Dinv; int* info;
    (void)hipMalloc(&A, n * n * 8); (void)hipMalloc(&Dinv, n * PB * 8); (void)hipMalloc(&info, 64);
    // a well-conditioned SPD-ish matrix: diagonal 4, off-diagonals small
    std::vector<double> blk(PB * PB);
    for (int i = 0; i < PB; ++i) for (int j = 0; j < PB; ++j) blk[i * PB + j] = (i == j) ? 4.0 : 0.01 / (1 + abs(i - j));
    (void)hipMemset(A, 0, n * n * 8);
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    int lo, hi; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStream_t P; (void)hipStreamCreateWithPriority(&P, hipStreamNonBlocking, hi);
    hipEvent_t e0, e1, b0, b1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&b0); (void)hipEventCreate(&b1);
    for (int leave_out : {-1, 0, 1, 8, 32}) {
        hipStream_t B;
        if (leave_out <= 0) (void)hipStreamCreateWithFlags(&B, hipStreamNonBlocking);
        else {
            std::vector<uint32_t> mask((pr.multiProcessorCount + 31) / 32, 0xffffffffu);
            for (int i = 0; i < leave_out; ++i) mask[i / 32] &= ~(1u << (i % 32));
            hipError_t e = hipExtStreamCreateWithCUMask(&B, (uint32_t)mask.size(), mask.data());
            if (e != hipSuccess) { printf("mask create failed: %s\n", hipGetErrorString(e)); return 1; }
        }
        for (int rep = 0; rep < 2; ++rep) {
            for (int p = 0; p < 4; ++p)
                (void)hipMemcpy2D(A + (long)p * PB * ld + p * PB, ld * 8, blk.data(), PB * 8, PB * 8, PB, hipMemcpyHostToDevice);
            (void)hipDeviceSynchronize();
            const int T = 100;
            if (leave_out >= 0) {
                (void)hipEventRecord(b0, B);
                k_syrk_update<true><<<dim3(T * (T + 1) / 2), GEMM_THREADS, 0, B>>>(A, ld, 28, 28, 4 * PB, 512, BatchStride{});
                (void)hipEventRecord(b1, B);
            }
            // let the bulk kernel fill the GPU, then run four diagonal blocks back to back on P
            for (volatile int spin = 0; spin < 200000; ++spin) {}
            float ms[4];
            hipEvent_t ev[5]; for (auto& e : ev) (void)hipEventCreate(&e);
            (void)hipEventRecord(ev[0], P);
            for (int p = 0; p < 4; ++p) { k_potrf_diag<<<1, POTRF_MFMA_THREADS, 0, P>>>(A, ld, p, Dinv, info, BatchStride{}, 0, 1 << 20); (void)hipEventRecord(ev[p + 1], P); }
            (void)hipDeviceSynchronize();
            for (int p = 0; p < 4; ++p) (void)hipEventElapsedTime(&ms[p], ev[p], ev[p + 1]);
            float bms = 0; if (leave_out >= 0) (void)hipEventElapsedTime(&bms, b0, b1);
            if (rep) printf("bulk CUs left out %3d: bulk %.3f ms; diagonal blocks %.0f %.0f %.0f %.0f us\n", leave_out, bms, ms[0] * 1e3, ms[1] * 1e3, ms[2] * 1e3, ms[3] * 1e3);
        }
        (void)hipStreamDestroy(B);
    }
    return 0;
}
