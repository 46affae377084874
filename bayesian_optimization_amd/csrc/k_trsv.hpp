// Device kernels of libgpbo (gfx950), part of kernels.hpp: triangular solves for single right-hand sides and the refinement step of alpha.
#pragma once

namespace gpbo {

// ---------------------------------------------------------------------------------------
// Triangular solves for one right-hand side, one launch per 128-block step.
// Forward step p: z_p = inv(L_pp) w_p ; w_ib -= L[ib,p] z_p for ib > p.   grid = nb - p.
constexpr int TRSV_FWD_THREADS = 512;              // eight waves, 16 rows each: both operands fit the registers without spills
__global__ __launch_bounds__(TRSV_FWD_THREADS) void k_trsv_fwd_step(const double* __restrict__ L, long ld,
                                                       const double* __restrict__ Dinv, double* __restrict__ w,
                                                       double* __restrict__ z, int p, BatchStride bs,
                                                       const double* __restrict__ gate) {
    if (gate && gate[blockIdx.z * bs.par] == 0.0) return;      // a refinement solve the model does not need
    __shared__ double yp[PB];
    L += blockIdx.z * bs.mat;
    Dinv += blockIdx.z * bs.dinv;
    w += blockIdx.z * bs.vec;
    z += blockIdx.z * bs.vec;
    __shared__ double zp[PB];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // Both operands of the step (this wave's 16 rows of the inverted diagonal block and of the block of L below) are
    // requested before anything is waited for: the step is a chain of memory round trips and little else, and this
    // makes it one round trip instead of eight.  Row of value u: wv + 8 u.
    const double* dinv = Dinv + (long)p * PB * PB;
    const int ib = p + blockIdx.x;
    const double* Lp = L + (long)ib * PB * ld + (long)p * PB;
    double d0[16], d1[16], l0[16], l1[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int r = wv + 8 * u;
        d0[u] = dinv[r * PB + lane];
        d1[u] = dinv[r * PB + 64 + lane];
    }
    if (blockIdx.x != 0) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int r = wv + 8 * u;
            l0[u] = Lp[(long)r * ld + lane];
            l1[u] = Lp[(long)r * ld + 64 + lane];
        }
    }
    if (tid < PB) yp[tid] = w[p * PB + tid];
    __syncthreads();
    const double y0 = yp[lane], y1 = yp[64 + lane];
    double sv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) sv[u] = d0[u] * y0 + d1[u] * y1;
    const double zsum = wave_sum<16>(sv);           // value lane >> 2
    if ((lane & 3) == 0) zp[wv + 8 * (lane >> 2)] = zsum;
    __syncthreads();
    if (blockIdx.x == 0) {
        if (tid < PB) z[p * PB + tid] = zp[tid];
        return;
    }
    const double z0 = zp[lane], z1 = zp[64 + lane];
#pragma unroll
    for (int u = 0; u < 16; ++u) sv[u] = l0[u] * z0 + l1[u] * z1;
    const double wsum = wave_sum<16>(sv);
    if ((lane & 3) == 0) w[ib * PB + wv + 8 * (lane >> 2)] -= wsum;
}

// Backward step p (L^T x = z): x_p = inv(L_pp)^T z_p ; z_jb -= L[p,jb]^T x_p for jb < p.  grid = p + 1.
__global__ __launch_bounds__(256) void k_trsv_bwd_step(const double* __restrict__ L, long ld,
                                                       const double* __restrict__ Dinv, double* __restrict__ z,
                                                       double* __restrict__ x, int p, BatchStride bs,
                                                       const double* __restrict__ gate) {
    if (gate && gate[blockIdx.z * bs.par] == 0.0) return;
    __shared__ double zp[PB];
    L += blockIdx.z * bs.mat;
    Dinv += blockIdx.z * bs.dinv;
    z += blockIdx.z * bs.vec;
    x += blockIdx.z * bs.vec;
    __shared__ double xp[PB];
    __shared__ double red[256];
    const int tid = threadIdx.x;
    // as in the forward step: both operands are requested up front, one memory round trip per step
    const double* dinv = Dinv + (long)p * PB * PB;
    const int c = tid & 127, hh = tid >> 7;
    const int jb = (int)blockIdx.x - 1;
    const double* Lp = L + (long)p * PB * ld + (long)(jb < 0 ? 0 : jb) * PB;
    double dv[64], lv[64];
#pragma unroll
    for (int u = 0; u < 64; ++u) dv[u] = dinv[(hh * 64 + u) * PB + c];
    if (blockIdx.x != 0) {
#pragma unroll
        for (int u = 0; u < 64; ++u) lv[u] = Lp[(long)(hh * 64 + u) * ld + c];
    }
    if (tid < PB) zp[tid] = z[p * PB + tid];
    __syncthreads();
    {
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < 64; ++u) s += dv[u] * zp[hh * 64 + u];
        red[tid] = s;
    }
    __syncthreads();
    if (tid < PB) xp[tid] = red[tid] + red[tid + 128];
    __syncthreads();
    if (blockIdx.x == 0) {
        if (tid < PB) x[p * PB + tid] = xp[tid];
        return;
    }
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < 64; ++u) s += lv[u] * xp[hh * 64 + u];
    red[tid] = s;
    __syncthreads();
    if (tid < PB) z[jb * PB + tid] -= red[tid] + red[tid + 128];
}

// One step of iterative refinement of alpha1 = (K0 + eps I)^-1 y against the covariance matrix itself (rebuilt entry by
// entry, the factor is not used for the residual): r = y - (K0 + eps I) alpha1, delta = L^-T L^-1 r, alpha1 += delta.
// It removes what the blocked substitution (explicitly inverted 128x128 diagonal blocks) and the factor's backward error
// add on top of a LAPACK-style solve when K0 + eps I is badly conditioned (sigmaY = 0: eps = 1e-15 n): theta1 = y.alpha1/n
// is then off by ~cond * 1e-17 (1e-6 at cond 4e11), and with it the variance.  Gated on the device: k_refine_flag sets
// params[P_REFINE] from (max diag L / min diag L)^2, and every kernel of the step returns at once when it is 0.
constexpr double REFINE_COND = 1e7;
__global__ __launch_bounds__(256) void k_refine_flag(const double* __restrict__ L, long ld, int n, double* __restrict__ params,
                                                     BatchStride bs) {
    __shared__ double lo[256], hi[256];
    L += blockIdx.z * bs.mat;
    params += blockIdx.z * bs.par;
    const int tid = threadIdx.x;
    double a = INFINITY, b = -INFINITY;
    for (int i = tid; i < n; i += 256) {
        const double v = L[(long)i * ld + i];
        a = fmin(a, v);
        b = fmax(b, v);
    }
    lo[tid] = a;
    hi[tid] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            lo[tid] = fmin(lo[tid], lo[tid + o]);
            hi[tid] = fmax(hi[tid], hi[tid + o]);
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double c = (hi[0] / lo[0]) * (hi[0] / lo[0]);
        params[P_COND] = c;
        params[P_REFINE] = (c > REFINE_COND) ? 1.0 : 0.0;      // NaN compares false: a failed factorisation is not refined
    }
}

// r_i = y_i - sum_j k(x_i, x_j) a_j - eps a_i for REF_ROWS rows per workgroup; the row's products are summed over
// j = tid, tid + 256, ... and then over the 256 threads in a fixed tree (deterministic).
constexpr int REF_ROWS = 8;
__global__ __launch_bounds__(256) void k_refine_residual(const double* __restrict__ Xts, int n, int n_pad, int d,
                                                         const double* __restrict__ y, const double* __restrict__ a,
                                                         double eps_val, const double* __restrict__ eps_ptr, int kernel,
                                                         const double* __restrict__ params, double* __restrict__ r, BatchStride bs) {
    if (params[blockIdx.z * bs.par + P_REFINE] == 0.0) return;
    Xts += blockIdx.z * bs.xts;
    y += blockIdx.z * bs.vec;
    a += blockIdx.z * bs.vec;
    r += blockIdx.z * bs.vec;
    const double eps = eps_ptr ? eps_ptr[blockIdx.z * bs.slot] : eps_val;
    __shared__ double red[REF_ROWS][256];
    const int tid = threadIdx.x;
    const int i0 = blockIdx.x * REF_ROWS;
    double acc[REF_ROWS];
#pragma unroll
    for (int q = 0; q < REF_ROWS; ++q) acc[q] = 0.0;
    for (int j = tid; j < n; j += 256) {
        const double aj = a[j];
        double D[REF_ROWS];
#pragma unroll
        for (int q = 0; q < REF_ROWS; ++q) D[q] = 0.0;
        for (int t = 0; t < d; ++t) {
            const double xj = Xts[(long)t * n_pad + j];
            const double* xi = Xts + (long)t * n_pad + i0;
#pragma unroll
            for (int q = 0; q < REF_ROWS; ++q) {
                const double df = xi[q] - xj;
                D[q] += df * df;
            }
        }
#pragma unroll
        for (int q = 0; q < REF_ROWS; ++q) {
            const int i = i0 + q;
            const double kij = (i == j) ? 1.0 + eps : kern_eval(D[q], kernel);
            acc[q] = fma(kij, aj, acc[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < REF_ROWS; ++q) red[q][tid] = acc[q];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
#pragma unroll
            for (int q = 0; q < REF_ROWS; ++q) red[q][tid] += red[q][tid + o];
        }
        __syncthreads();
    }
    if (tid < REF_ROWS) {
        const int i = i0 + tid;
        r[i] = (i < n) ? y[i] - red[tid][0] : 0.0;
    }
}

__global__ __launch_bounds__(256) void k_refine_add(double* __restrict__ a, const double* __restrict__ delta, int n_pad,
                                                    const double* __restrict__ params, BatchStride bs) {
    if (params[blockIdx.z * bs.par + P_REFINE] == 0.0) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_pad) a[blockIdx.z * bs.vec + i] += delta[blockIdx.z * bs.vec + i];
}

// theta1 = y.alpha1/n, sum log diag L, -logL, scale factors and the alpha used for the mean.
// One workgroup (deterministic order).
__global__ __launch_bounds__(256) void k_fit_scalars(const double* __restrict__ L, long ld, const double* __restrict__ y,
                                                     const double* __restrict__ alpha1, double* __restrict__ alpha_out,
                                                     int n, int n_pad, int variant, double* __restrict__ params,
                                                     BatchStride bs) {
    __shared__ double r1[256];
    L += blockIdx.z * bs.mat;
    y += blockIdx.z * bs.vec;
    alpha1 += blockIdx.z * bs.vec;
    alpha_out += blockIdx.z * bs.vec;
    params += blockIdx.z * bs.par;
    __shared__ double r2[256];
    __shared__ double r3[256];
    const int tid = threadIdx.x;
    double ya = 0.0, sl = 0.0, ym = -INFINITY;
    for (int i = tid; i < n; i += 256) {
        ya += y[i] * alpha1[i];
        sl += log(L[(long)i * ld + i]);
        ym = fmax(ym, y[i]);
    }
    r1[tid] = ya;
    r2[tid] = sl;
    r3[tid] = ym;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            r1[tid] += r1[tid + o];
            r2[tid] += r2[tid + o];
            r3[tid] = fmax(r3[tid], r3[tid + o]);
        }
        __syncthreads();
    }
    const double theta1 = r1[0] / n;
    if (tid == 0) {
        params[P_YA] = r1[0];
        params[P_SUMLOG] = r2[0];
        params[P_THETA1] = theta1;
        params[P_YMAX] = r3[0];
        params[P_NLL] = 0.5 * n * (log(2.0 * M_PI) + 1.0) + 0.5 * n * log(theta1) + r2[0];
        params[P_CLAMP] = 1.0;
        params[P_NBETA] = 0.0;
        params[P_DEG] = 0.0;
        if (variant != 1) {      // two-factor forms (v4 and the MATLAB form): theta1-scaled covariances, eps unscaled
            params[P_CB] = theta1;
            params[P_C0] = theta1;
            params[P_C1] = 1.0;
        } else {
            params[P_CB] = 1.0;
            params[P_C0] = theta1;
            params[P_C1] = theta1;
        }
    }
    for (int i = tid; i < n_pad; i += 256) alpha_out[i] = (i < n) ? (variant != 1 ? alpha1[i] / theta1 : alpha1[i]) : 0.0;
}

// -logL as the MATLAB two-factor form reports it (GP/BayesOpt_v3.m:362): n/2 (log 2 pi + 1) + n/2 log(y.alpha2 / n) +
// sum log diag L2, with alpha2 and L2 of the SECOND factorisation (theta1 K0 + eps).  One workgroup, fixed order.
__global__ __launch_bounds__(256) void k_nll_second_factor(const double* __restrict__ L2, long ld, const double* __restrict__ y,
                                                           const double* __restrict__ alpha2, int n, double* __restrict__ params) {
    __shared__ double r1[256];
    __shared__ double r2[256];
    const int tid = threadIdx.x;
    double ya = 0.0, sl = 0.0;
    for (int i = tid; i < n; i += 256) {
        ya += y[i] * alpha2[i];
        sl += log(L2[(long)i * ld + i]);
    }
    r1[tid] = ya;
    r2[tid] = sl;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            r1[tid] += r1[tid + o];
            r2[tid] += r2[tid + o];
        }
        __syncthreads();
    }
    if (tid == 0) params[P_NLL] = 0.5 * n * (log(2.0 * M_PI) + 1.0) + 0.5 * n * log(r1[0] / n) + r2[0];
}

}  // namespace gpbo
