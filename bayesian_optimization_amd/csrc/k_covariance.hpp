// Device kernels of libgpbo (gfx950), part of kernels.hpp: the covariance matrix of the training points.
#pragma once

namespace gpbo {

// ---------------------------------------------------------------------------------------
// K(X,X): the 128-blocks on and below the block diagonal of the n_pad x n_pad matrix (all the factorisation reads; the
// blocks above are never touched), scale*k(D) + eps on the diagonal, identity pad.
// Xts is [d][n_pad] = X / theta.  scale_ptr == nullptr -> 1.  eps: eps_ptr[0] if eps_ptr, else eps_val.
// A workgroup covers 32 rows x 64 columns; a thread owns one column j and KM_ROWS rows, so x_j[t] is loaded once per
// KM_ROWS elements and the rows' coordinates are wave-uniform (scalar loads).
constexpr int KM_ROWS = 8;
__global__ __launch_bounds__(256) void k_kernel_matrix(const double* __restrict__ Xts, int n, int n_pad, int d,
                                                       double* __restrict__ K, const double* __restrict__ scale_ptr,
                                                       double eps_val, const double* __restrict__ eps_ptr, int kernel,
                                                       BatchStride bs) {
    if ((blockIdx.x * 64) / PB > (blockIdx.y * 32) / PB) return;      // strictly above the block diagonal
    Xts += blockIdx.z * bs.xts;
    K += blockIdx.z * bs.mat;
    if (scale_ptr) scale_ptr += blockIdx.z * bs.par;
    if (eps_ptr) eps_ptr += blockIdx.z * bs.slot;
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i0 = blockIdx.y * 32 + (threadIdx.x >> 6) * KM_ROWS;      // n_pad is a multiple of 128: no ragged edge
    const double scale = scale_ptr ? *scale_ptr : 1.0;
    const double eps = eps_ptr ? *eps_ptr : eps_val;      // from device memory when the launch is replayed from a graph
    double D[KM_ROWS];
#pragma unroll
    for (int r = 0; r < KM_ROWS; ++r) D[r] = 0.0;
    for (int t = 0; t < d; ++t) {
        const double xj = Xts[(long)t * n_pad + j];
        const double* xi = Xts + (long)t * n_pad + i0;
#pragma unroll
        for (int r = 0; r < KM_ROWS; ++r) {
            const double df = xi[r] - xj;
            D[r] += df * df;
        }
    }
#pragma unroll
    for (int r = 0; r < KM_ROWS; ++r) {
        const int i = i0 + r;
        double v;
        if (i == j) v = (i < n) ? scale + eps : 1.0;
        else if (i < n && j < n) v = scale * kern_eval(D[r], kernel);
        else v = 0.0;
        K[(long)i * n_pad + j] = v;
    }
}

}  // namespace gpbo
