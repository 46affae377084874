// Debug probe: where the cycles of a register-resident 16x16 Cholesky tile go (one wave, pivots through v_readlane).
// Variants: 0 full; 1 one Newton step; 2 pivot chain only (no trailing update except the next diagonal);
// 3 trailing update with lane-local multipliers (no readlane); 4 readlane traffic only (no fma).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ double readlane_f64(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}
template <int VAR> __global__ __launch_bounds__(1024) void k_leaf(const double* __restrict__ A, double* __restrict__ out, long long* __restrict__ cyc) {
    const int lane = threadIdx.x & 63, i = lane & 31;
    double v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = (i < 16) ? ((c <= i) ? A[(i & 15) * 16 + c] : 0.0) : ((c == i - 16) ? 1.0 : 0.0);
    double v0[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v0[c] = v[c];
    long long t0, t1;
    const long long w0 = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory");
  for (int rep = 0; rep < 64; ++rep) {
#pragma unroll
    for (int c = 0; c < 16; ++c) asm volatile("" : "+v"(v[c]));
    double pivp = readlane_f64(v[0], 0);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double piv = (VAR == 5 || VAR == 6) ? pivp : readlane_f64(v[j], j);
        double rd = __builtin_amdgcn_rsq(piv);
        if (VAR == 6) {
            const double e = fma(-piv * rd, rd, 1.0);
            rd = fma(rd * e, fma(0.375, e, 0.5), rd);
        } else {
            rd = rd * fma(-0.5 * piv * rd, rd, 1.5);
            if (VAR != 1) rd = rd * fma(-0.5 * piv * rd, rd, 1.5);
        }
        double d = piv * rd;
        d = fma(0.5 * rd, fma(-d, d, piv), d);
        if ((VAR == 5 || VAR == 6) && j < 15) {
            const double l = readlane_f64(v[j], j + 1) * rd;
            pivp = fma(-l, l, readlane_f64(v[j + 1], j + 1));
        }
        const double lij = (i > j) ? v[j] * rd : 0.0;
        v[j] = (i == j) ? d : lij;
#pragma unroll
        for (int c = j + 1; c < 16; ++c) {
            if (VAR == 2 && c > j + 1) continue;
            if (VAR == 3) {
                v[c] = fma(-lij, v[j], v[c]);
            } else if (VAR == 4) {
                const double lcj = readlane_f64(v[j], c);
                if (c == j + 1) v[c] = fma(-lij, lcj, v[c]);
                else v[c] += __longlong_as_double(__double_as_longlong(lcj) & 1);
            } else {
                const double lcj = readlane_f64(v[j], c);
                v[c] = fma(-lij, lcj, v[c]);
            }
        }
        if (VAR == 7) {
#pragma unroll
            for (int c = j + 1; c < 16; ++c) asm volatile("" : "+v"(v[c]));
        }
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) { v[c] = fma(1e-9, v[c], v0[c]); asm volatile("" : "+v"(v[c])); }
  }
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) s += v[c];
    asm volatile("s_nop 0\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(s) : "memory");
    out[lane] = s;
    if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = wall_clock64() - w0; }
}

// 32x32 tile: lanes 0..31 rows of A, lanes 32..63 rows of the identity
__global__ __launch_bounds__(1024) void k_leaf32(const double* __restrict__ A, double* __restrict__ out, long long* __restrict__ cyc) {
    const int lane = threadIdx.x & 63, i = lane;
    double v[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) v[c] = (i < 32) ? ((c <= i) ? A[(i & 31) * 32 + c] : 0.0) : ((c == i - 32) ? 1.0 : 0.0);
    double v0[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) v0[c] = v[c];
    long long t0, t1;
    const long long w0 = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory");
  for (int rep = 0; rep < 64; ++rep) {
#pragma unroll
    for (int c = 0; c < 32; ++c) asm volatile("" : "+v"(v[c]));
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const double piv = readlane_f64(v[j], j);
        double rd = __builtin_amdgcn_rsq(piv);
        rd = rd * fma(-0.5 * piv * rd, rd, 1.5);
        rd = rd * fma(-0.5 * piv * rd, rd, 1.5);
        double d = piv * rd;
        d = fma(0.5 * rd, fma(-d, d, piv), d);
        const double lij = (i > j) ? v[j] * rd : 0.0;
        v[j] = (i == j) ? d : lij;
#pragma unroll
        for (int c = j + 1; c < 32; ++c) {
            const double lcj = readlane_f64(v[j], c);
            v[c] = fma(-lij, lcj, v[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < 32; ++c) { v[c] = fma(1e-9, v[c], v0[c]); asm volatile("" : "+v"(v[c])); }
  }
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < 32; ++c) s += v[c];
    asm volatile("s_nop 0\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(s) : "memory");
    out[lane] = s;
    if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = wall_clock64() - w0; }
}

template <int VAR> void run(const double* dA, double* dout, long long* dcyc) {
    long long best = 1ll << 60, wall = 0;
    for (int rep = 0; rep < 5; ++rep) {
        k_leaf<VAR><<<1, 64>>>(dA, dout, dcyc);
        long long c[2];
        (void)hipMemcpy(c, dcyc, 16, hipMemcpyDeviceToHost);
        if (c[0] < best) { best = c[0]; wall = c[1]; }
    }
    printf("variant %d: %lld s_memtime ticks, %lld wall ticks (100 MHz) for 64 tiles: %.2f us per tile\n", VAR, best, wall, wall / 100.0 / 64);
}
int main() {
    std::vector<double> A(256);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) A[i * 16 + j] = exp(-0.5 * (i - j) * (i - j) / 9.0) + (i == j ? 0.1 : 0.0);
    double *dA, *dout; long long* dcyc;
    (void)hipMalloc(&dA, 256 * 8); (void)hipMalloc(&dout, 64 * 8); (void)hipMalloc(&dcyc, 16);
    (void)hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice);
    run<0>(dA, dout, dcyc); run<1>(dA, dout, dcyc); run<2>(dA, dout, dcyc); run<3>(dA, dout, dcyc); run<4>(dA, dout, dcyc); run<5>(dA, dout, dcyc); run<6>(dA, dout, dcyc); run<7>(dA, dout, dcyc);
    {
        std::vector<double> B(1024);
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) B[i * 32 + j] = exp(-0.5 * (i - j) * (i - j) / 9.0) + (i == j ? 0.1 : 0.0);
        double* dB; (void)hipMalloc(&dB, 1024 * 8);
        (void)hipMemcpy(dB, B.data(), 1024 * 8, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 3; ++rep) {
            k_leaf32<<<1, 64>>>(dB, dout, dcyc);
            long long c[2];
            (void)hipMemcpy(c, dcyc, 16, hipMemcpyDeviceToHost);
            printf("32x32 tile: %lld ticks, %.2f us per tile\n", c[0] / 64, c[1] / 100.0 / 64);
        }
    }
    return 0;
}
