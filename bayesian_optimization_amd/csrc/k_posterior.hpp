// Device kernels of libgpbo (gfx950), part of kernels.hpp: posterior mean and variance over a chunk of candidates.
#pragma once

namespace gpbo {

// ---------------------------------------------------------------------------------------
// K(X*,X) chunk: Bmat[k][j] = cb * k(D(x_k, xs_j)) for k < n (0 for the pad), j over the chunk
// (j contiguous), plus partial means  mupart[kc][j] = sum_{k in chunk kc} Bmat[k][j] * alpha[k].
// One thread per candidate; a workgroup covers KS_ROWS training rows, whose scaled coordinates and alpha are staged in
// LDS once (row-major, read back as broadcasts).  Measured against wave-uniform scalar loads of the same values
// (scripts/ks_gemm_ab.hip): 0.218 vs 0.243 ms per 4096 x 32768 panel at d = 4, 0.371 vs 0.547 ms per 32768 x 4096 panel at
// d = 8 with the Matern kernel -- all of what the GEMM form of the distance gains (0.225 / 0.357 ms), without its
// cancellation error (18-28 ulp against 1-2 ulp).  D = 0: any d up to MAX_D, coordinates straight from global memory.
constexpr int KS_ROWS = 512;
template <int D>
__global__ __launch_bounds__(256) void k_ks_build(const double* __restrict__ Xs, long m, long j0, int d_rt,
                                                  const double* __restrict__ Xts, int n, int n_pad,
                                                  const double* __restrict__ alpha, const double* __restrict__ params,
                                                  double* __restrict__ Bmat, long ldb, double* __restrict__ mupart, int kc,
                                                  int kernel) {
    const int d = (D > 0) ? D : d_rt;
    const long jl = (long)blockIdx.x * 256 + threadIdx.x;   // column in the chunk
    long j = j0 + jl;
    if (j >= m) j = m - 1;                                   // padded columns: recompute the last candidate
    const double cb = params[P_CB];
    const int k_lo = blockIdx.y * kc;
    const int k_hi = min(k_lo + kc, n_pad);
    double xs[(D > 0) ? D : MAX_D];
#pragma unroll
    for (int t = 0; t < ((D > 0) ? D : MAX_D); ++t)
        if (t < d) xs[t] = Xs[j * d + t] * params[P_INVTHETA + t];
    double mu = 0.0;
    if (D > 0) {
        __shared__ double xr[KS_ROWS * ((D > 0) ? D : 1)];
        __shared__ double al[KS_ROWS];
        const int rows = k_hi - k_lo;                        // kc == KS_ROWS for this form (the launch guarantees it)
        for (int idx = threadIdx.x; idx < rows * D; idx += 256) {
            const int t = idx / rows, k = idx - t * rows;    // coalesced over k for a fixed dimension
            xr[k * D + t] = Xts[(long)t * n_pad + k_lo + k];
        }
        for (int k = threadIdx.x; k < rows; k += 256) al[k] = alpha[k_lo + k];
        __syncthreads();
        const int live = min(rows, max(n - k_lo, 0));        // rows below n hold data, the rest is the pad
        double* Bp = Bmat + (long)k_lo * ldb + jl;
        auto entry = [&](int k) {
            double Dq = 0.0;
#pragma unroll
            for (int t = 0; t < D; ++t) {
                const double df = xs[t] - xr[k * D + t];
                Dq += df * df;
            }
            const double v = cb * kern_eval(Dq, kernel);
            mu += v * al[k];
            Bp[(long)k * ldb] = v;
        };
        if (live == KS_ROWS) {                               // the common case: a constant trip count the compiler can pipeline
#pragma unroll 4
            for (int k = 0; k < KS_ROWS; ++k) entry(k);
        } else {
            for (int k = 0; k < live; ++k) entry(k);
            for (int k = live; k < rows; ++k) Bp[(long)k * ldb] = 0.0;
        }
    } else {
        for (int k = k_lo; k < k_hi; ++k) {
            double v = 0.0;
            if (k < n) {
                double Dq = 0.0;
                for (int t = 0; t < d; ++t) {
                    const double df = xs[t] - Xts[(long)t * n_pad + k];
                    Dq += df * df;
                }
                v = cb * kern_eval(Dq, kernel);
                mu += v * alpha[k];
            }
            Bmat[(long)k * ldb + jl] = v;
        }
    }
    mupart[(long)blockIdx.y * ldb + jl] = mu;
}

// V = W * Bmat (W lower triangular), fused epilogue: per-column sum of squares of the
// 128 rows of the tile -> partial[ib][j].
// STORE_V: also keep V rows < v_rows of the tile in the resident buffer Vres[row][j0 + col]
// (row pitch ldv = number of candidates) for the incremental append path.
//
// Tile order: groups of cg column blocks; inside a group row blocks heaviest first (longest k range), the group's column
// blocks of a row block next to each other, so the resident tiles share few W row panels (served from L2) and the tail
// of a launch consists of one-k-block tiles.  Groups of 128 column blocks against one group of 256 (the whole chunk):
// +2.4 % on two machines, -0.2 % on a third that ran both at the higher rate; narrower groups lose (48: -3 %, 8: -9 %);
// alternating heavy and light row blocks loses 1-2 %.  An XCD-aware 8x8 super-tile order that also shares the Ks panels
// measured 9 % slower (tiles marching through k in lockstep expose the miss latency); bands of 2 / 4 / 8 row blocks per
// column block, dispatched together: -3 / -10 / -20 % at 32 row blocks, -0.3 / -1.2 / -2.8 % at 256.  Persistent workgroups that chain
// their tiles (no prologue) and draw them from a counter: +0.7 % with one counter but 20 GB instead of 12 GB fetched per
// launch (the row blocks of a column block no longer meet in one L2), -0.5 % with a counter per XCD; with a static
// assignment -4 % (workgroups do not progress at equal rates).  The hardware dispatcher's order stays.
// MODE: the tile engine's scheduling mode; 2 (priority raise + fragment prefetch) is never worse than 0 or 1 here: equal to
// the better of them at 32 row blocks, whichever that is on the machine at hand, and +0.9 % at 256.
template <bool STORE_V, int MODE>
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_trmm_sumsq(const double* __restrict__ W, long ld,
                                                             const double* __restrict__ Bmat, long ldb,
                                                             double* __restrict__ partial, int nb, int nj,
                                                             double* __restrict__ Vres, long ldv, long j0, long m, int v_rows,
                                                             int cg) {
    const int per = nb * cg;
    const int grp = blockIdx.x / per, r = blockIdx.x % per;
    const int cw = min(cg, nj - grp * cg);              // the last group may be narrower
    const int ib = nb - 1 - r / cw;
    const int jb = grp * cg + r % cw;
    __shared__ double lds[GEMM_LDS_DOUBLES];
    f64x4 acc[4][4];
    acc_zero(acc);
    gemm_tile_128<true, false, MODE>(W + (long)ib * PB * ld, ld, Bmat + (long)jb * TN, ldb, 0, (ib + 1) * PB, acc, lds, ib * PB);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;
    if (STORE_V) {
#pragma unroll
        for (int ta = 0; ta < 4; ++ta)
#pragma unroll
            for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = ib * PB + acc_row(ta, r);
                    const long col = j0 + (long)jb * TN + acc_col(tb);
                    if (row < v_rows && col < m) Vres[(long)row * ldv + col] = acc[ta][tb][r];
                }
    }
    double cs[4];
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) {
        double s = 0.0;
#pragma unroll
        for (int ta = 0; ta < 4; ++ta)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[ta][tb][r] * acc[ta][tb][r];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        cs[tb] = s;
    }
    double* red = lds;   // the tile engine ended on a barrier: LDS is free
    if ((lane >> 4) == 0) {
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) red[wr * 128 + wc * 64 + tb * 16 + lane] = cs[tb];
    }
    __syncthreads();
    if (tid < 128) partial[(long)ib * ldb + (long)jb * TN + tid] = red[tid] + red[128 + tid];
}

// Trend basis of the kriging flavour (polymat, Gaussian_Process_Regression.ipynb:113-140): columns for
// C = 1..3 inputs and degree 0..3, in the reference's column order.  Returns f(x) . beta.
__device__ __forceinline__ double trend_eval(const double* __restrict__ x, int C, int deg, const double* __restrict__ beta) {
    double f = beta[0];
    if (deg == 0) return f;
    if (C == 1) {
        f += beta[1] * x[0];
        if (deg >= 2) f += beta[2] * x[0] * x[0];
    } else if (C == 2) {
        f += beta[1] * x[0] + beta[2] * x[1];
        if (deg >= 2) f += beta[3] * x[0] * x[0] + beta[4] * x[1] * x[1];
        if (deg >= 3) f += beta[5] * x[0] * x[1];
    } else {
        f += beta[1] * x[0] + beta[2] * x[1] + beta[3] * x[2];
        if (deg >= 2) f += beta[4] * x[0] * x[0] + beta[5] * x[1] * x[1] + beta[6] * x[2] * x[2];
        if (deg >= 3) f += beta[7] * x[0] * x[1] + beta[8] * x[0] * x[2] + beta[9] * x[1] * x[2];
    }
    return f;
}

// var = c0 - c1 * sum_ib partial[ib][j] (clamped: < 1e-60 -> 0 unless the model says otherwise),
// mu = sum_kc mupart[kc][j] (+ trend for the kriging flavour).
__global__ __launch_bounds__(256) void k_finalize(const double* __restrict__ partial, const double* __restrict__ mupart,
                                                  long ldb, int nb, int nkc, const double* __restrict__ params,
                                                  double* __restrict__ mu, double* __restrict__ var, long j0, long m,
                                                  double* __restrict__ sres, const double* __restrict__ Xs, int d) {
    const long jl = (long)blockIdx.x * 256 + threadIdx.x;
    const long j = j0 + jl;
    if (jl >= ldb || j >= m) return;
    double s = 0.0;
    for (int ib = 0; ib < nb; ++ib) s += partial[(long)ib * ldb + jl];
    double mm = 0.0;
    for (int c = 0; c < nkc; ++c) mm += mupart[(long)c * ldb + jl];
    double v = params[P_C0] - params[P_C1] * s;
    if (params[P_CLAMP] != 0.0 && v < 1e-60) v = 0.0;
    const int nbeta = (int)params[P_NBETA];
    if (nbeta > 0) {
        double x[3] = {0.0, 0.0, 0.0};
        for (int t = 0; t < d && t < 3; ++t) x[t] = Xs[j * d + t];
        mm += trend_eval(x, d, (int)params[P_DEG], params + P_BETA);
    }
    mu[j] = mm;
    var[j] = v;
    if (sres) sres[j] = s;
}

}  // namespace gpbo
