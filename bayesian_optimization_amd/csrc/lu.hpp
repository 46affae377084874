// LU with partial pivoting for the trend flavour ("kriging"): the reference solves with np.linalg.solve on the
// covariance matrix WITHOUT jitter (Gaussian_Process_Regression.ipynb:201-202, 232-233), which returns numbers for
// matrices that are not numerically positive definite (the notebook's own 3-D cell: 800 grid points, cond 4e19), where a
// Cholesky factorisation stops at a non-positive pivot.  When that happens the fit falls back to this path:
//   P K = L U (blocked right-looking, panels of 128 columns, trailing update on the fp64 MFMA tile engine);
//   every solve -- the trend's right-hand sides at the fit, the correlation vectors of a chunk of candidates in the sweep --
//   is a pair of blocked triangular solves with those factors (block row by block row: one product on the tile engine with
//   the rows already solved, one substitution inside the 128-row diagonal block), as the reference's LU solve does it.
// No explicit inverse: at cond(K) > 1 / eps only the backward stability of the substitution keeps r^T K^-1 r and the
// interpolant meaningful; a product with an explicitly formed inverse does not have it.
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

// ---- solves with the factors, for a k-major block of right-hand sides V[k][j] (row pitch ldv, `cols` columns)

// V <- P V: the factorisation's row exchanges in order, one thread per column.
__global__ __launch_bounds__(256) void k_lu_permute_rows(double* __restrict__ V, long ldv, int cols, int n_pad, const int* __restrict__ ipiv) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= cols) return;
    for (int j = 0; j < n_pad; ++j) {
        const int p = ipiv[j];
        if (p != j) {
            const double a = V[(long)j * ldv + x], b = V[(long)p * ldv + x];
            V[(long)j * ldv + x] = b;
            V[(long)p * ldv + x] = a;
        }
    }
}

// V[block row ib] -= LU[block row ib][k0 .. k1) * V[k0 .. k1): the part of a block row's substitution that involves rows
// solved earlier, on the tile engine.  Grid: one workgroup per 128 columns of V.
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_lu_rect_update(const double* __restrict__ LU, long ld, double* __restrict__ V, long ldv,
                                                                  int ib, int k0, int k1) {
    __shared__ double lds[GEMM_LDS_DOUBLES];
    const int jb = blockIdx.x;
    const double* Arow = LU + (long)ib * PB * ld;                  // rows of the tile, k contiguous
    const double* Bk = V + (long)jb * TN;                           // element (k, j) at Bk[k * ldv + j]
    double* C = V + (long)ib * PB * ldv + (long)jb * TN;
    f64x4 acc[4][4];
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[ta][tb][r] = -C[(long)acc_row(ta, r) * ldv + acc_col(tb)];
    gemm_tile_128<true, false, 1>(Arow, ld, Bk, ldv, k0, k1, acc, lds);
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(long)acc_row(ta, r) * ldv + acc_col(tb)] = -acc[ta][tb][r];
}

// Substitution inside diagonal block ib, in place, one thread per column of V.  UPPER = false: the unit lower triangle
// (forward); UPPER = true: the upper triangle with its diagonal (backward).
template <bool UPPER>
__global__ __launch_bounds__(256) void k_lu_diag_solve(const double* __restrict__ LU, long ld, int ib, double* __restrict__ V, long ldv,
                                                       int cols) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= cols) return;
    const int r0 = ib * PB;
    if (!UPPER) {
        for (int i = 1; i < PB; ++i) {
            const double* Li = LU + (long)(r0 + i) * ld + r0;
            double s = V[(long)(r0 + i) * ldv + x];
            for (int k = 0; k < i; ++k) s -= Li[k] * V[(long)(r0 + k) * ldv + x];
            V[(long)(r0 + i) * ldv + x] = s;
        }
    } else {
        for (int i = PB - 1; i >= 0; --i) {
            const double* Ui = LU + (long)(r0 + i) * ld + r0;
            double s = V[(long)(r0 + i) * ldv + x];
            for (int k = i + 1; k < PB; ++k) s -= Ui[k] * V[(long)(r0 + k) * ldv + x];
            V[(long)(r0 + i) * ldv + x] = s / Ui[i];
        }
    }
}

// partial[0][j] = sum_k R[k][j] Z[k][j] (= r_j^T K^-1 r_j once Z holds the solves), the other nb - 1 rows of partial zero:
// the shape k_finalize sums.  One thread per column.
__global__ __launch_bounds__(256) void k_col_dot(const double* __restrict__ R, const double* __restrict__ Z, long ldv, int n_pad, int cols,
                                                 double* __restrict__ partial, int nb) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= cols) return;
    double s = 0.0;
    for (int k = 0; k < n_pad; ++k) s += R[(long)k * ldv + x] * Z[(long)k * ldv + x];
    partial[x] = s;
    for (int ib = 1; ib < nb; ++ib) partial[(long)ib * ldv + x] = 0.0;
}

}  // namespace gpbo
