// A/B of the two ways to build a K(X*, X) panel (the operand of the variance product):
//   direct : D = sum_t (x*_t - x_t)^2 on scaled coordinates (what the library's k_ks_build does), one thread per candidate;
//   gemm   : D = |x*|^2 + |x|^2 - 2 x . x*  with the inner products on v_mfma_f64_16x16x4_f64 (K dimension = d) and the
//            training row norms staged in LDS (the form of GP/BayesOpt_v3.m:445 and GP/pdist_kg.m:15).
// Prints time per panel, achieved write bandwidth, and the element-wise error of both against a long-double host
// evaluation on a sample, for the two benchmark shapes.  Build: hipcc --offload-arch=gfx950 -O3 -o ks_gemm_ab ks_gemm_ab.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double kern_eval(double D, int kernel) {
    if (kernel == 0) return exp(-0.5 * D);
    const double r = sqrt(D);
    const double s5r = 2.23606797749978969641 * r;
    return (1.0 + s5r + (5.0 / 3.0) * D) * exp(-s5r);
}

// direct: thread per candidate, training point wave-uniform.  Xts: [d][n] scaled, Xs: [m][d] scaled.
template <int D>
__global__ __launch_bounds__(256) void ks_direct(const double* __restrict__ Xs, const double* __restrict__ Xts, int n,
                                                 double* __restrict__ B, long ldb, int kernel) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    double xs[D];
#pragma unroll
    for (int t = 0; t < D; ++t) xs[t] = Xs[j * D + t];
    const int k_lo = blockIdx.y * 512, k_hi = min(k_lo + 512, n);
    for (int k = k_lo; k < k_hi; ++k) {
        double Dq = 0.0;
#pragma unroll
        for (int t = 0; t < D; ++t) {
            const double df = xs[t] - Xts[(long)t * n + k];
            Dq += df * df;
        }
        B[(long)k * ldb + j] = kern_eval(Dq, kernel);
    }
}

// direct, training rows staged in LDS (row-major [512][D]: one broadcast ds_read_b128 per two coordinates), loop unrolled by 4
template <int D>
__global__ __launch_bounds__(256) void ks_direct_lds(const double* __restrict__ Xs, const double* __restrict__ Xtr, int n,
                                                     double* __restrict__ B, long ldb, int kernel) {
    __shared__ double xr[512 * D];
    const int tid = threadIdx.x;
    const int k_lo = blockIdx.y * 512;
    for (int idx = tid; idx < 512 * D; idx += 256) xr[idx] = Xtr[(long)k_lo * D + idx];
    __syncthreads();
    const long j = (long)blockIdx.x * 256 + tid;
    double xs[D];
#pragma unroll
    for (int t = 0; t < D; ++t) xs[t] = Xs[j * D + t];
#pragma unroll 4
    for (int k = 0; k < 512; ++k) {
        double Dq = 0.0;
#pragma unroll
        for (int t = 0; t < D; ++t) {
            const double df = xs[t] - xr[k * D + t];
            Dq += df * df;
        }
        B[(long)(k_lo + k) * ldb + j] = kern_eval(Dq, kernel);
    }
}

// gemm: a wave owns 16 candidates (operand B of the MFMA, loaded once) and walks the training rows 16 at a time;
// the rows' coordinates and norms come from LDS (512 rows per workgroup).  Xtr: [n][D] scaled (row-major).
template <int D>
__global__ __launch_bounds__(256) void ks_gemm(const double* __restrict__ Xs, const double* __restrict__ Xtr, int n,
                                               double* __restrict__ B, long ldb, int kernel) {
    static_assert(D % 4 == 0, "inner dimension in MFMA steps of 4");
    __shared__ double xr[512 * D];
    __shared__ double nr[512];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int k_lo = blockIdx.y * 512;
    for (int idx = tid; idx < 512 * D; idx += 256) xr[idx] = Xtr[(long)k_lo * D + idx];
    __syncthreads();
    for (int r = tid; r < 512; r += 256) {
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < D; ++t) s = fma(xr[r * D + t], xr[r * D + t], s);
        nr[r] = s;
    }
    __syncthreads();
    const long j0 = (long)blockIdx.x * 64 + 16 * w;
    double bfrag[D / 4];
    double ns = 0.0;                                   // |x*_j|^2 for column j0 + li
#pragma unroll
    for (int q = 0; q < D / 4; ++q) bfrag[q] = Xs[(j0 + li) * D + 4 * q + lk];
#pragma unroll
    for (int t = 0; t < D; ++t) {
        const double v = Xs[(j0 + li) * D + t];
        ns = fma(v, v, ns);
    }
    for (int kt = 0; kt < 32; ++kt) {
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < D / 4; ++q)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xr[(16 * kt + li) * D + 4 * q + lk], bfrag[q], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * kt + lk + 4 * r;
            double Dq = nr[row] + ns - 2.0 * acc[r];
            Dq = Dq < 0.0 ? 0.0 : Dq;
            B[(long)(k_lo + row) * ldb + j0 + li] = kern_eval(Dq, kernel);
        }
    }
}

static double u01(uint64_t seed, uint64_t i) {
    uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

template <int D>
void run(const char* name, int n, int m, double theta, int kernel) {
    std::vector<double> X((size_t)n * D), Xt((size_t)D * n), Xs((size_t)m * D);
    for (size_t i = 0; i < X.size(); ++i) X[i] = u01(3, i) / theta;
    for (size_t i = 0; i < Xs.size(); ++i) Xs[i] = u01(33, i) / theta;
    for (int k = 0; k < n; ++k)
        for (int t = 0; t < D; ++t) Xt[(size_t)t * n + k] = X[(size_t)k * D + t];
    double *dX, *dXt, *dXs, *dB;
    (void)hipMalloc(&dX, X.size() * 8); (void)hipMalloc(&dXt, Xt.size() * 8); (void)hipMalloc(&dXs, Xs.size() * 8);
    (void)hipMalloc(&dB, (size_t)n * m * 8);
    (void)hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dXt, Xt.data(), Xt.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dXs, Xs.data(), Xs.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    std::vector<double> out((size_t)n * m);
    for (int form = 0; form < 3; ++form) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0);
            if (form == 0) hipLaunchKernelGGL(ks_direct<D>, dim3(m / 256, n / 512), dim3(256), 0, 0, dXs, dXt, n, dB, (long)m, kernel);
            else if (form == 1) hipLaunchKernelGGL(ks_gemm<D>, dim3(m / 64, n / 512), dim3(256), 0, 0, dXs, dX, n, dB, (long)m, kernel);
            else hipLaunchKernelGGL(ks_direct_lds<D>, dim3(m / 256, n / 512), dim3(256), 0, 0, dXs, dX, n, dB, (long)m, kernel);
            (void)hipEventRecord(e1);
            (void)hipDeviceSynchronize();
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        (void)hipMemcpy(out.data(), dB, out.size() * 8, hipMemcpyDeviceToHost);
        double emax = 0, erel = 0;
        for (int s = 0; s < 20000; ++s) {
            const int k = (int)(u01(7, s) * n), j = (int)(u01(8, s) * m);
            long double Dq = 0;
            for (int t = 0; t < D; ++t) {
                const long double df = (long double)Xs[(size_t)j * D + t] - (long double)X[(size_t)k * D + t];
                Dq += df * df;
            }
            long double ref;
            if (kernel == 0) ref = expl(-0.5L * Dq);
            else { const long double s5r = sqrtl(5.0L * Dq); ref = (1.0L + s5r + (5.0L / 3.0L) * Dq) * expl(-s5r); }
            const double err = fabs((double)(out[(size_t)k * m + j] - ref));
            emax = fmax(emax, err);
            erel = fmax(erel, err / (double)ref);
        }
        printf("%s %-6s: %8.3f ms per %d x %d panel, %7.1f GB/s written, max abs err %.2e, max rel err %.2e (%.1f ulp of 1)\n", name,
               form == 0 ? "direct" : form == 1 ? "gemm" : "dirlds", best, n, m, 8.0 * n * m / (best * 1e-3) / 1e9, emax, erel, emax / 1.11e-16);
    }
    (void)hipFree(dX); (void)hipFree(dXt); (void)hipFree(dXs); (void)hipFree(dB);
}

int main() {
    run<4>("cfg2 (n=4096, d=4, RBF, theta=0.35)       ", 4096, 32768, 0.35, 0);
    run<8>("cfg3 (n=32768, d=8, Matern-5/2, theta=0.6)", 32768, 4096, 0.6, 1);
    return 0;
}
