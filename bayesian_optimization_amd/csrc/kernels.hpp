// Device kernels of libgpbo (gfx950): the shared definitions are here, the kernels in the k_*.hpp parts included at the
// end (covariance, Cholesky, inverse, triangular solves, posterior, acquisition, likelihood sweep, append).
// Everything is fp64.  Matrices are row-major, padded to a multiple of 128 rows/columns (pad = identity), so no kernel
// needs edge handling in the training dimension; candidate chunks are padded to 128 columns and the padded columns are
// computed but never written to the outputs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_f64.hpp"

namespace gpbo {

constexpr int PB = 128;    // Cholesky panel / diagonal block

// params block on the device (doubles)
enum { P_THETA1 = 0, P_CB = 1, P_C0 = 2, P_C1 = 3, P_NLL = 4, P_YA = 5, P_SUMLOG = 6, P_YMAX = 7, P_INVTHETA = 8,
       P_BETA = 40, P_NBETA = 52, P_DEG = 53, P_CLAMP = 54, P_REFINE = 55, P_COND = 56 };
constexpr int PARAMS_DOUBLES = 64;

// A batched context holds S problems of the same shape side by side ("slots": every buffer is S times as long) and
// runs them in lockstep: the kernels of the fit take the slot from blockIdx.z and move their pointers by these element
// strides.  All zero (and gridDim.z = 1) for an ordinary context.
struct BatchStride {
    long mat = 0;    // n_pad * n_pad        (K / L, the scratch matrix, W)
    long dinv = 0;   // n_pad * 128          (inverted diagonal blocks)
    long xts = 0;    // MAX_D * n_pad        (scaled inputs)
    long vec = 0;    // n_pad                (alpha, y, work vectors)
    long par = 0;    // PARAMS_DOUBLES
    long gp = 0;     // gradient tile partials
    long slot = 0;   // sizeof(LmlSlot) / 8  (per-slot setting record, addressed as doubles)
};
constexpr int MAX_TREND = 10;
constexpr int MAX_D = 32;

// Lower-triangle tile index t = i (i + 1) / 2 + j, 0 <= j <= i  ->  (i, j).  Launching exactly the lower tiles
// instead of a square grid whose upper half exits at once is worth 10-35 % on the triangular updates: an empty
// workgroup still occupies a dispatch slot and its 72 KiB of LDS for a moment.
__device__ __forceinline__ void tri_decode(int t, int& i, int& j) {
    if (t < (1 << 22)) {
        // single precision puts the root within one of i here (one instruction instead of the ~40 of a double square
        // root: this sits at the head of every triangular-grid workgroup and inside the diagonal-block kernel's phases);
        // the two loops make it exact
        i = (int)((__fsqrt_rn(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while ((i + 1) * (i + 2) / 2 <= t) ++i;
        while (i * (i + 1) / 2 > t) --i;
        j = t - i * (i + 1) / 2;
        return;
    }
    i = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((long)(i + 1) * (i + 2) / 2 <= t) ++i;
    while ((long)i * (i + 1) / 2 > t) --i;
    j = t - (int)((long)i * (i + 1) / 2);
}

// The same lower triangle of T block rows walked in bands of R rows: inside a band the tiles come in chunks of eight
// columns, the R rows of a chunk one after the other, so that the R tiles that read the same column panel are eight
// workgroup indices apart -- workgroups go round-robin over the eight XCDs, so they meet in one L2 while they run.
// Band b (rows R b .. R b + R - 1) holds the R x R b rectangle left of the diagonal and its R x R tip; the rows of an
// incomplete last band follow in row-major order.
template <int R>
__device__ __forceinline__ void band_decode(int t, int T, int& i, int& j) {
    auto first = [](int b) { return R * R * (b * (b - 1) / 2) + b * (R * (R + 1) / 2); };
    const int nfull = T / R;
    int b = min((int)(__fsqrt_rn(2.0f * (float)t) * (1.0f / R)), nfull);
    while (b < nfull && first(b + 1) <= t) ++b;
    while (first(b) > t) --b;
    int u = t - first(b);
    if (b == nfull) {
        i = R * b;
        while (u >= i + 1) {
            u -= i + 1;
            ++i;
        }
        j = u;
        return;
    }
    const int rect = R * R * b;
    if (u < rect) {
        const int full = (R * b) >> 3;
        if (u < 8 * R * full) {
            const int w = u % (8 * R);
            i = R * b + (w >> 3);
            j = 8 * (u / (8 * R)) + (w & 7);
        } else {
            const int cr = R * b - 8 * full, w = u - 8 * R * full;
            i = R * b + w / cr;
            j = 8 * full + w % cr;
        }
        return;
    }
    int ri, rj;
    tri_decode(u - rect, ri, rj);
    i = R * b + ri;
    j = R * b + rj;
}

// a * b + c with the constant c held in a scalar register pair.  For a 64-bit literal addend the compiler otherwise builds
// the constant in a vector register pair (two vector moves) in front of every v_fmac_f64: three vector instructions per
// Horner step instead of one.
__device__ __forceinline__ double fma_sc(double a, double b, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}

// exp(x) for x <= 0 (or NaN).  Every exponential on this path is of that kind: a covariance, the normal density.  The
// algorithm and the coefficients are those of the device library's exp (range reduction by ln 2 in two parts, degree-11
// polynomial, ldexp), the results identical to exp(x) bit for bit (tests/probes/exp_neg_probe.hip: 2^25 arguments and the special
// values), but the polynomial's constants sit in scalar registers (fma_sc): 20 vector instructions fewer per value.
__device__ __forceinline__ double exp_neg(double x) {
    constexpr double LOG2E = __builtin_bit_cast(double, 0x3ff71547652b82feULL);
    constexpr double LN2_HI = __builtin_bit_cast(double, 0x3fe62e42fefa39efULL);
    constexpr double LN2_LO = __builtin_bit_cast(double, 0x3c7abc9e3b39803fULL);
    const double n = __builtin_rint(x * LOG2E);
    double r = fma(-LN2_HI, n, x);
    r = fma(-LN2_LO, n, r);
    double p = fma_sc(r, __builtin_bit_cast(double, 0x3e5ade156a5dcb37ULL), __builtin_bit_cast(double, 0x3e928af3fca7ab0cULL));
    p = fma_sc(r, p, __builtin_bit_cast(double, 0x3ec71dee623fde64ULL));
    p = fma_sc(r, p, __builtin_bit_cast(double, 0x3efa01997c89e6b0ULL));
    p = fma_sc(r, p, __builtin_bit_cast(double, 0x3f2a01a014761f6eULL));
    p = fma_sc(r, p, __builtin_bit_cast(double, 0x3f56c16c1852b7b0ULL));
    p = fma_sc(r, p, __builtin_bit_cast(double, 0x3f81111111122322ULL));
    p = fma_sc(r, p, __builtin_bit_cast(double, 0x3fa55555555502a1ULL));
    p = fma_sc(r, p, __builtin_bit_cast(double, 0x3fc5555555555511ULL));
    p = fma_sc(r, p, __builtin_bit_cast(double, 0x3fe000000000000bULL));
    p = fma(r, p, 1.0);
    p = fma(r, p, 1.0);
    double z = ldexp(p, (int)n);
    if (x < -1075.0) z = 0.0;
    return z;
}

__device__ __forceinline__ double kern_eval(double D, int kernel) {
    if (kernel == 0) return exp_neg(-0.5 * D);
    const double r = sqrt(D);
    const double s5r = 2.23606797749978969641 * r;
    return (1.0 + s5r + (5.0 / 3.0) * D) * exp_neg(-s5r);
}

// -2 dk/d(D) as a function of the scaled squared distance D: the factor F in dK/dtheta_t = F q_t / theta_t.
__device__ __forceinline__ double kern_dfac(double D, int kernel) {
    if (kernel == 0) return exp_neg(-0.5 * D);
    const double s5r = 2.23606797749978969641 * sqrt(D);
    return (5.0 / 3.0) * (1.0 + s5r) * exp_neg(-s5r);
}

}  // namespace gpbo

#include "k_covariance.hpp"
#include "k_cholesky.hpp"
#include "k_inverse.hpp"
#include "k_trsv.hpp"
#include "k_posterior.hpp"
#include "k_acquisition.hpp"
#include "k_lml.hpp"
#include "k_append.hpp"
