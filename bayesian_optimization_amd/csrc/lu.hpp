// LU with partial pivoting for the trend flavour ("kriging"): the reference solves with np.linalg.solve on the
// covariance matrix WITHOUT jitter (Gaussian_Process_Regression.ipynb:201-202, 232-233), which returns numbers for
// matrices that are not numerically positive definite (the notebook's own 3-D cell: 800 grid points, cond 4e19), where a
// Cholesky factorisation stops at a non-positive pivot.  When that happens the fit falls back to this path:
//   P K = L U (blocked right-looking, panels of 128 columns, trailing update on the fp64 MFMA tile engine),
//   Kinv = U^-1 L^-1 P explicitly (n^3, once per fit),
//   per candidate r^T Kinv r on the tile engine (k_gemm_dot: the full product Kinv R with a fused column dot against R).
// All matrices row-major n_pad x n_pad, identity padded, like everywhere else in the library.
#pragma once
#include "kernels.hpp"

namespace gpbo {

// The kernel-matrix build writes the 128-blocks on and below the block diagonal only: mirror them into the rest.
__global__ __launch_bounds__(256) void k_mirror_upper(double* __restrict__ A, int n_pad) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i < n_pad && j < n_pad && (j / PB) > (i / PB)) A[(long)i * n_pad + j] = A[(long)j * n_pad + i];
}

// Panel factorisation: columns c0 .. c0 + 127, rows c0 .. n_pad - 1, by ONE workgroup (1024 threads) in global memory,
// column by column: pivot = the entry of largest modulus on or below the diagonal (lowest row on ties, as LAPACK's
// idamax), rows swapped inside the panel, multipliers = column / pivot, rank-1 update of the panel's remaining columns.
// ipiv[c] = row exchanged with row c at step c.  info: first column (1-based) whose pivot is exactly zero.
__global__ __launch_bounds__(1024) void k_lu_panel(double* __restrict__ A, long ld, int n_pad, int c0, int* __restrict__ ipiv,
                                                   int* __restrict__ info) {
    __shared__ double rv[1024];
    __shared__ int ri[1024];
    __shared__ double prow[PB];
    const int tid = threadIdx.x;
    for (int j = 0; j < PB; ++j) {
        const int col = c0 + j;
        double bv = -1.0;
        int bi = col;
        for (int i = col + tid; i < n_pad; i += 1024) {
            const double v = fabs(A[(long)i * ld + col]);
            if (v > bv) {
                bv = v;
                bi = i;
            }
        }
        rv[tid] = bv;
        ri[tid] = bi;
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) {
            if (tid < o && (rv[tid + o] > rv[tid] || (rv[tid + o] == rv[tid] && ri[tid + o] < ri[tid]))) {
                rv[tid] = rv[tid + o];
                ri[tid] = ri[tid + o];
            }
            __syncthreads();
        }
        const int p = ri[0];
        if (tid == 0) {
            ipiv[col] = p;
            if (!(rv[0] > 0.0)) atomicCAS(info, 0, col + 1);      // an exactly zero (or NaN) column: singular, as numpy reports
        }
        if (tid < PB && p != col) {
            const double a = A[(long)col * ld + c0 + tid], b = A[(long)p * ld + c0 + tid];
            A[(long)col * ld + c0 + tid] = b;
            A[(long)p * ld + c0 + tid] = a;
        }
        __syncthreads();
        if (tid < PB) prow[tid] = A[(long)col * ld + c0 + tid];
        __syncthreads();
        const double piv = prow[j];
        for (int i = col + 1 + tid; i < n_pad; i += 1024) A[(long)i * ld + col] /= piv;
        __syncthreads();
        // eight rows per pass, one thread per (row, panel column)
        const int r = tid >> 7, t = tid & 127;
        if (t > j) {
            const double u = prow[t];
            for (int i = col + 1 + r; i < n_pad; i += 8) A[(long)i * ld + c0 + t] -= A[(long)i * ld + col] * u;
        }
        __syncthreads();
    }
}

// The panel's row exchanges applied to the columns outside the panel (left: the multipliers of earlier panels; right:
// the part of the matrix still to be eliminated).  One thread per column, the 128 exchanges in order.
__global__ __launch_bounds__(256) void k_lu_swap_rows(double* __restrict__ A, long ld, int n_pad, int c0, const int* __restrict__ ipiv) {
    int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= c0) x += PB;
    if (x >= n_pad) return;
    for (int j = 0; j < PB; ++j) {
        const int row = c0 + j, p = ipiv[row];
        if (p != row) {
            const double a = A[(long)row * ld + x], b = A[(long)p * ld + x];
            A[(long)row * ld + x] = b;
            A[(long)p * ld + x] = a;
        }
    }
}

// U12 = L11^-1 A12: forward substitution with the unit lower triangle of the panel's diagonal block, one thread per
// column right of the panel (the multipliers are the same for every thread: uniform loads).
__global__ __launch_bounds__(256) void k_lu_trsm_u(double* __restrict__ A, long ld, int n_pad, int c0) {
    const int x = c0 + PB + blockIdx.x * 256 + threadIdx.x;
    if (x >= n_pad) return;
    for (int i = 1; i < PB; ++i) {
        double s = A[(long)(c0 + i) * ld + x];
        const double* Li = A + (long)(c0 + i) * ld + c0;
        for (int k = 0; k < i; ++k) s -= Li[k] * A[(long)(c0 + k) * ld + x];
        A[(long)(c0 + i) * ld + x] = s;
    }
}

// Trailing update A22 -= A21 U12 on the tile engine: tile (ib, jb) of the blocks below / right of panel p.
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_lu_gemm_update(double* __restrict__ A, long ld, int p) {
    __shared__ double lds[GEMM_LDS_DOUBLES];
    const int ib = p + 1 + blockIdx.y, jb = p + 1 + blockIdx.x;
    const double* A21 = A + (long)ib * PB * ld + (long)p * PB;      // rows of the tile, k contiguous
    const double* U12 = A + (long)p * PB * ld + (long)jb * PB;      // element (k, j) at U12[k * ld + j]
    double* C = A + (long)ib * PB * ld + (long)jb * PB;
    f64x4 acc[4][4];
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[ta][tb][r] = -C[(long)acc_row(ta, r) * ld + acc_col(tb)];
    gemm_tile_128<true, false, 1>(A21, ld, U12, ld, 0, PB, acc, lds);
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(long)acc_row(ta, r) * ld + acc_col(tb)] = -acc[ta][tb][r];
}

// Column c of Kinv = U^-1 L^-1 P e_c, one thread per column: the exchanges move the 1 of e_c to row `pos`; forward
// substitution with the unit lower factor from there on, backward substitution with U.  Each thread reads back only what it
// wrote itself; for a fixed row the threads of a wave touch consecutive addresses.
__global__ __launch_bounds__(256) void k_lu_inverse_cols(const double* __restrict__ LU, long ld, int n_pad, const int* __restrict__ ipiv,
                                                         double* __restrict__ X) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_pad) return;
    int pos = c;
    for (int j = 0; j < n_pad; ++j) {
        const int p = ipiv[j];
        if (pos == j) pos = p;
        else if (pos == p) pos = j;
    }
    for (int i = 0; i < n_pad; ++i) {
        double s = (i == pos) ? 1.0 : 0.0;
        if (i > pos) {
            const double* Li = LU + (long)i * ld;
            for (int k = pos; k < i; ++k) s -= Li[k] * X[(long)k * ld + c];
        }
        X[(long)i * ld + c] = s;
    }
    for (int i = n_pad - 1; i >= 0; --i) {
        const double* Ui = LU + (long)i * ld;
        double s = X[(long)i * ld + c];
        for (int k = i + 1; k < n_pad; ++k) s -= Ui[k] * X[(long)k * ld + c];
        X[(long)i * ld + c] = s / Ui[i];
    }
}

// x = M v for q vectors stored column after column with pitch n_pad (M a full n x n matrix): wave per row.
__global__ __launch_bounds__(256) void k_full_gemv(const double* __restrict__ M, long ld, int n, const double* __restrict__ v,
                                                   double* __restrict__ x, int n_pad) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.y;
    if (row >= n) return;
    double s = 0.0;
    for (int j = lane; j < n; j += 64) s += M[(long)row * ld + j] * v[(long)q * n_pad + j];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) x[(long)q * n_pad + row] = s;
}

// sum of log(diag) of a factor, rows < n.  One workgroup, fixed order.
__global__ __launch_bounds__(256) void k_sumlog_diag(const double* __restrict__ L, long ld, int n, double* __restrict__ out) {
    __shared__ double r[256];
    const int tid = threadIdx.x;
    double s = 0.0;
    for (int i = tid; i < n; i += 256) s += log(L[(long)i * ld + i]);
    r[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) r[tid] += r[tid + o];
        __syncthreads();
    }
    if (tid == 0) out[0] = r[0];
}

// The candidate sweep of the LU path: V = Kinv * Bmat over the full k range (Kinv is a full matrix), epilogue
// partial[ib][j] = sum over the tile's 128 rows of V[row][j] * Bmat[row][j], so that sum_ib partial[ib][j] = r_j^T Kinv r_j.
// Same tile engine, tile order and reduction shape as k_trmm_sumsq.
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_gemm_dot(const double* __restrict__ Kinv, long ld, const double* __restrict__ Bmat,
                                                            long ldb, double* __restrict__ partial, int nb, int nj, int cg) {
    const int per = nb * cg;
    const int grp = blockIdx.x / per, r0 = blockIdx.x % per;
    const int cw = min(cg, nj - grp * cg);
    const int ib = r0 / cw;
    const int jb = grp * cg + r0 % cw;
    __shared__ double lds[GEMM_LDS_DOUBLES];
    f64x4 acc[4][4];
    acc_zero(acc);
    gemm_tile_128<true, false, 2>(Kinv + (long)ib * PB * ld, ld, Bmat + (long)jb * TN, ldb, 0, nb * PB, acc, lds);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;
    double cs[4];
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) {
        double s = 0.0;
#pragma unroll
        for (int ta = 0; ta < 4; ++ta)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                s += acc[ta][tb][r] * Bmat[(long)(ib * PB + acc_row(ta, r)) * ldb + (long)jb * TN + acc_col(tb)];
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

}  // namespace gpbo
