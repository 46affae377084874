// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 (cycles per MFMA per SIMD, chip TFLOP/s).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, double a0, double b0) {
    f64x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f64x4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[gridDim.x * 256] = (double)(t1 - t0);
}

template <int NACC>
void run(const char* name, int blocks_per_cu) {
    int iters = 2000;
    int grid = 256 * blocks_per_cu;
    double* out;
    hipMalloc(&out, (grid * 256 + 8) * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k_mfma<NACC><<<grid, 256>>>(out, iters, 1.0, 2.0);
    hipEventRecord(e0);
    k_mfma<NACC><<<grid, 256>>>(out, iters, 1.0, 2.0);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double cyc;
    hipMemcpy(&cyc, out + grid * 256, 8, hipMemcpyDeviceToHost);
    double n_mfma_per_wave = (double)iters * NACC;
    double flops = (double)grid * 4 * n_mfma_per_wave * 2048.0;
    printf("%s NACC=%d blocks/CU=%d: %.3f ms, %.2f TFLOP/s, memtime ticks per MFMA (one wave) %.2f\n", name, NACC, blocks_per_cu, ms,
           flops / (ms * 1e-3) / 1e12, cyc / n_mfma_per_wave);
    hipFree(out);
}

int main() {
    run<1>("dependent chain", 1);
    run<4>("4 acc", 1);
    run<16>("16 acc", 1);
    run<16>("16 acc", 2);
    run<8>("8 acc", 2);
    return 0;
}
