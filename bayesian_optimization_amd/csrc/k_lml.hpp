// Device kernels of libgpbo (gfx950), part of kernels.hpp: the likelihood sweep: -log L and its gradient, one setting or many in lockstep.
#pragma once

namespace gpbo {

// ---------------------------------------------------------------------------------------
// Batched likelihood sweep helpers.
// Per setting: Xts[t][i] = Xraw[t][i] / theta[t], params[P_INVTHETA + t] = 1 / theta[t].
__global__ __launch_bounds__(256) void k_setup_setting(const double* __restrict__ Xraw, const double* __restrict__ theta,
                                                       int d, int n_pad, double* __restrict__ Xts, double* __restrict__ params,
                                                       BatchStride bs) {
    theta += blockIdx.z * bs.slot;
    Xts += blockIdx.z * bs.xts;
    params += blockIdx.z * bs.par;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int t = blockIdx.y;
    const double it = 1.0 / theta[t];
    if (i < n_pad) Xts[(long)t * n_pad + i] = Xraw[(long)t * n_pad + i] * it;
    if (i == 0) params[P_INVTHETA + t] = it;
}

// Gradient of -logL w.r.t. theta:  g_t = -1/2 sum_{j,k} c_jk dK_t[j][k],
//   c_jk = n a_j a_k / (y.a) - Kinv_jk,  Kinv = W^T W,  dK_t[j][k] = q_t F_jk / theta_t  (F = K0 for RBF, see kern_dfac),
//   q_t = ((x_jt - x_kt)/theta_t)^2.
// One workgroup = one lower tile (jb >= kb) of Kinv on the MFMA tile engine, reduced against dK_t
// in the epilogue without ever storing Kinv.  gpart[(jb*nb + kb)*GRAD_D + t] = sum_tile c_jk q_t K0_jk
// (x2 for off-diagonal tiles); the caller zero-fills gpart.
constexpr int GRAD_D = 16;
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_lml_grad_tiles(const double* __restrict__ W, long ld,
                                                                    const double* __restrict__ Xts, int n, int n_pad, int d,
                                                                    const double* __restrict__ alpha1,
                                                                    const double* __restrict__ params,
                                                                    double* __restrict__ gpart, int nb, int kernel, BatchStride bs) {
    W += blockIdx.z * bs.mat;
    Xts += blockIdx.z * bs.xts;
    alpha1 += blockIdx.z * bs.vec;
    params += blockIdx.z * bs.par;
    gpart += blockIdx.z * bs.gp;
    int jb, kb;                    // 1-D grid over the lower tiles; jb ascending = longest contraction first
    tri_decode(blockIdx.x, jb, kb);
    __shared__ double lds[GEMM_LDS_DOUBLES];
    f64x4 acc[4][4];
    acc_zero(acc);
    // A(j, r) = W[r][jb*128 + j], B(r, k) = W[r][kb*128 + k]; W[r][j] = 0 for r < j
    gemm_tile_128<false, false>(W + (long)jb * PB, ld, W + (long)kb * PB, ld, jb * PB, n_pad, acc, lds);
    // The epilogue keeps nothing but the 64 accumulators live (two workgroups per CU): the tile's scaled coordinates
    // and alpha go through LDS (the tile engine ended on a barrier), c_jk K0_jk overwrites the accumulator, and the
    // d sums are formed one dimension at a time.
    const int tid = threadIdx.x;
    double* xj = lds;                        // [GRAD_D][128]
    double* xk = lds + GRAD_D * PB;          // [GRAD_D][128]
    double* aj = lds + 2 * GRAD_D * PB;      // [128]
    double* ak = aj + PB;                    // [128]
    double* red = ak + PB;                   // [4][GRAD_D]
    for (int idx = tid; idx < d * PB; idx += GEMM_THREADS) {
        const int t = idx >> 7, i = idx & 127;
        xj[idx] = Xts[(long)t * n_pad + jb * PB + i];
        xk[idx] = Xts[(long)t * n_pad + kb * PB + i];
    }
    if (tid < PB) aj[tid] = alpha1[jb * PB + tid];
    else ak[tid - PB] = alpha1[kb * PB + tid - PB];
    __syncthreads();
    const double scale_c = (double)n / params[P_YA];
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) {
        const int kc = acc_col(tb);
        const int k = kb * PB + kc;
#pragma unroll
        for (int ta = 0; ta < 4; ++ta)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jr = acc_row(ta, r);
                const int j = jb * PB + jr;
                double Dq = 0.0;
                for (int t = 0; t < d; ++t) {
                    const double df = xj[t * PB + jr] - xk[t * PB + kc];
                    Dq += df * df;
                }
                const double c = (scale_c * aj[jr] * ak[kc] - acc[ta][tb][r]) * kern_dfac(Dq, kernel);
                acc[ta][tb][r] = (j < n && k < n && j != k) ? c : 0.0;
            }
    }
    for (int t = 0; t < d; ++t) {
        double g = 0.0;
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
            const double xkt = xk[t * PB + acc_col(tb)];
#pragma unroll
            for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double df = xj[t * PB + acc_row(ta, r)] - xkt;
                    g += acc[ta][tb][r] * (df * df);
                }
        }
        for (int o = 32; o > 0; o >>= 1) g += __shfl_xor(g, o);
        if ((tid & 63) == 0) red[(tid >> 6) * GRAD_D + t] = g;
    }
    __syncthreads();
    const double wgt = (jb == kb) ? 1.0 : 2.0;
    if (tid < GRAD_D)
        gpart[((long)jb * nb + kb) * GRAD_D + tid] =
            (tid < d) ? wgt * (red[tid] + red[GRAD_D + tid] + red[2 * GRAD_D + tid] + red[3 * GRAD_D + tid]) : 0.0;
}

// Whole -logL (+ gradient) evaluation of ONE hyper-parameter setting in ONE workgroup, for n <= 128: covariance
// tiles built in registers, factor and inverse by the diagonal-block kernel's MFMA routines, z = W y, alpha = W^T z, the
// scalars, then Kinv = W^T W tile by tile out of LDS and the gradient sums.  grid = number of settings: a
// multi-start fit on ~100 points is one launch instead of ten launches per setting.
//   nll = n/2 (log 2pi + 1) + n/2 log(theta1) + sum log diag L,  theta1 = y.alpha / n       (mlefcn)
//   g_t = -1/(2 theta_t) sum_{j != k} (n a_j a_k / (y.a) - Kinv_jk) q_t F_jk   (jacmlefun; F = K0 for RBF, kern_dfac)
// LDS: W with pitch 129 (the packed copy of L lives in the same space while the inverse runs), scaled inputs, vectors.
constexpr int SMALL_WP = PB + 1;
constexpr int SMALL_WIDE_THREADS = 1024;       // workgroup of the one-workgroup routines for larger n (256 threads for small n)
constexpr int SMALL_LDS_DOUBLES = PB * SMALL_WP + GRAD_D * PB + 6 * PB + 16 * GRAD_D + 8;
// The evaluation itself, for one setting: th (d length scales, any address space), sg; results to *out_nll, out_grad[0..d),
// *out_info (written by thread 0 / threads < d).  All POTRF_THREADS threads must call it; lds: SMALL_LDS_DOUBLES doubles.
// The caller separates two calls (and its own reads of the outputs) by a barrier.
// TG x TG threads hold the matrix as TS x TS register tiles: 16 x 16 threads with TS = 2 (n <= 32; TS = 4 up to 64 if the
// caller asks for 256 threads there), 32 x 32 threads with TS = 2 / 4 for n <= 64 / 128 (WIDE: a 1024-thread workgroup,
// which also gives the factor and inverse routines their 16 waves; the host uses it above 32 points).  With 8 x 8 tiles on 256 threads a 36-point problem kept 15 threads busy in the covariance build and the
// gradient, a 94-point one 144.
struct SmallLds {
    double *Wl, *xs, *dg, *yv, *zv, *al, *r1, *r2, *red, *sc;
    __device__ __forceinline__ explicit SmallLds(double* lds) {
        Wl = lds;                         // [128][129]; first the packed L
        xs = Wl + PB * SMALL_WP;          // [GRAD_D][128] inputs / theta
        dg = xs + GRAD_D * PB;            // [128] diag L
        yv = dg + PB;
        zv = yv + PB;
        al = zv + PB;
        r1 = al + PB;                     // [128] reduction scratch: y_i alpha_i
        r2 = r1 + PB;                     // [128] reduction scratch: log diag L
        red = r2 + PB;                    // [16][GRAD_D] per-wave gradient partials
        sc = red + 16 * GRAD_D;           // [0] y.alpha  [1] sum log diag L
    }
};

// covariance tiles (identity pad beyond n) into the packed block, rows and columns < 16 TS
template <int TS, int TG>
__device__ __forceinline__ void lml_small_build(const SmallLds& m, int n, int d, double eps, int kernel) {
    const int tid = threadIdx.x, tr = tid / TG, tc = tid % TG;
#pragma unroll
    for (int rr = 0; rr < TS; ++rr)
#pragma unroll
        for (int cc = 0; cc < TS; ++cc) {
            const int r = tr * TS + rr, c = tc * TS + cc;
            if (c > r) continue;
            double v = 0.0;
            if (r == c) {
                v = (r < n) ? 1.0 + eps : 1.0;
            } else if (r < n) {
                double D = 0.0;
                for (int t = 0; t < d; ++t) {
                    const double df = m.xs[t * PB + r] - m.xs[t * PB + c];
                    D += df * df;
                }
                v = kern_eval(D, kernel);
            }
            m.Wl[tri_off(r) + c] = v;
        }
}

// everything after the inverse: W out of the packed block, z, alpha, the scalars, Kinv = W^T W tile by tile, gradient sums
template <int TS, int TG>
__device__ __forceinline__ void lml_small_finish(const SmallLds& m, int n, int d, const double* th, int kernel, int want_grad,
                                                 int fail, double* out_nll, double* out_grad, int* out_info) {
    constexpr int MS = TG * TS;                // rows and columns covered by the register tiles
    constexpr int NW = TG * TG / 64;           // waves
    const int tid = threadIdx.x, tr = tid / TG, tc = tid % TG;
    double* Wl = m.Wl;
    const int nt = (n + TS - 1) / TS;          // tile columns that hold data; the rest of the block is the identity pad
    double a[TS][TS];
#pragma unroll
    for (int rr = 0; rr < TS; ++rr)
#pragma unroll
        for (int cc = 0; cc < TS; ++cc) {
            const int r = tr * TS + rr, c = tc * TS + cc;
            a[rr][cc] = (c <= r) ? Wl[tri_off(r) + (c <= r ? c : r)] : 0.0;
        }
    __syncthreads();                           // every read of the packed W is done
#pragma unroll
    for (int rr = 0; rr < TS; ++rr)
#pragma unroll
        for (int cc = 0; cc < TS; ++cc) Wl[(tr * TS + rr) * SMALL_WP + tc * TS + cc] = a[rr][cc];
    __syncthreads();
    if (tid < PB) {                            // z = W y (row tid)
        double z = 0.0;
        if (tid < MS)
            for (int c = 0; c <= min(tid, nt * TS - 1); ++c) z = fma(Wl[tid * SMALL_WP + c], m.yv[c], z);      // y is zero in the pad
        m.zv[tid] = z;
    }
    __syncthreads();
    if (tid < PB) {                            // alpha = W^T z (column tid)
        double s = 0.0;
        if (tid < MS)                          // beyond the register tiles the W space holds no matrix
            for (int r = tid; r < nt * TS; ++r) s = fma(Wl[r * SMALL_WP + tid], m.zv[r], s);               // z is zero in the pad
        m.al[tid] = s;
        m.r1[tid] = m.yv[tid] * s;
        m.r2[tid] = (tid < n) ? log(m.dg[tid]) : 0.0;
    }
    __syncthreads();
    for (int o = 64; o > 0; o >>= 1) {
        if (tid < o) {
            m.r1[tid] += m.r1[tid + o];
            m.r2[tid] += m.r2[tid + o];
        }
        __syncthreads();
    }
    if (tid == 0) {
        m.sc[0] = m.r1[0];
        m.sc[1] = m.r2[0];
    }
    __syncthreads();
    const double ya = m.sc[0], sumlog = m.sc[1];
    const double theta1 = ya / n;
    const bool bad = fail >= 0;
    if (tid == 0) {
        const double nll = 0.5 * n * (log(2.0 * M_PI) + 1.0) + 0.5 * n * log(theta1) + sumlog;
        *out_nll = bad ? NAN : nll;
        *out_info = bad ? fail + 1 : 0;
    }
    if (!want_grad) return;
    // Kinv tile (tr, tc), tr >= tc: sum over r >= TS tr of W[r][TS tr + rr] W[r][TS tc + cc] (W is zero above its diagonal)
    const bool lower = tr >= tc;
#pragma unroll
    for (int rr = 0; rr < TS; ++rr)
#pragma unroll
        for (int cc = 0; cc < TS; ++cc) a[rr][cc] = 0.0;
    if (lower && tr < nt) {                    // pad rows of W are rows of the identity: they add nothing for j, k < n
        for (int r = tr * TS; r < nt * TS; ++r) {
            double wj[TS], wk[TS];
#pragma unroll
            for (int t = 0; t < TS; ++t) {
                wj[t] = Wl[r * SMALL_WP + tr * TS + t];
                wk[t] = Wl[r * SMALL_WP + tc * TS + t];
            }
#pragma unroll
            for (int rr = 0; rr < TS; ++rr)
#pragma unroll
                for (int cc = 0; cc < TS; ++cc) a[rr][cc] = fma(wj[rr], wk[cc], a[rr][cc]);
        }
        const double scale_c = (double)n / ya;
        const double wgt = (tr == tc) ? 1.0 : 2.0;
#pragma unroll
        for (int rr = 0; rr < TS; ++rr)
#pragma unroll
            for (int cc = 0; cc < TS; ++cc) {
                const int j = tr * TS + rr, k = tc * TS + cc;
                double c = 0.0;
                if (j < n && k < n && j != k) {
                    double D = 0.0;
                    for (int t = 0; t < d; ++t) {
                        const double df = m.xs[t * PB + j] - m.xs[t * PB + k];
                        D += df * df;
                    }
                    c = wgt * (scale_c * m.al[j] * m.al[k] - a[rr][cc]) * kern_dfac(D, kernel);
                }
                a[rr][cc] = c;
            }
    }
    for (int t = 0; t < d; ++t) {
        double g = 0.0;
        if (lower) {
#pragma unroll
            for (int rr = 0; rr < TS; ++rr)
#pragma unroll
                for (int cc = 0; cc < TS; ++cc) {
                    const double df = m.xs[t * PB + tr * TS + rr] - m.xs[t * PB + tc * TS + cc];
                    g = fma(a[rr][cc], df * df, g);
                }
        }
        for (int o = 32; o > 0; o >>= 1) g += __shfl_xor(g, o);
        if ((tid & 63) == 0) m.red[(tid >> 6) * GRAD_D + t] = g;
    }
    __syncthreads();
    if (tid < d) {
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < NW; w += 4)
            sum += (m.red[w * GRAD_D + tid] + m.red[(w + 1) * GRAD_D + tid]) + (m.red[(w + 2) * GRAD_D + tid] + m.red[(w + 3) * GRAD_D + tid]);
        out_grad[tid] = bad ? NAN : -0.5 * sum * (1.0 / th[tid]);
    }
}

template <bool WIDE>
__device__ __forceinline__ void lml_small_body(double* __restrict__ lds, const double* __restrict__ Xraw, int xpitch,
                                               const double* __restrict__ y, int n, int d, const double* th, double sg, int kernel,
                                               int want_grad, double* out_nll, double* out_grad, int* out_info) {
    const SmallLds m(lds);
    const int tid = threadIdx.x;
    const double eps = (sg == 0.0) ? 1e-15 * (double)n : sg * sg;
    for (int idx = tid; idx < d * PB; idx += (WIDE ? SMALL_WIDE_THREADS : POTRF_THREADS)) {
        const int t = idx >> 7, i = idx & 127;
        m.xs[idx] = (i < n) ? Xraw[(long)t * xpitch + i] * (1.0 / th[t]) : 0.0;
    }
    if (tid < PB) m.yv[tid] = (tid < n) ? y[tid] : 0.0;
    __syncthreads();
    // factor and inverse as in the diagonal-block kernel (register tile factorisation of the 16x16 tiles, everything
    // else on MFMA), on the packed copy; the product buffer and the tile inverses live in the rest of the W space
    __shared__ int fail_s;
    double* Tbuf = m.Wl + POTRF_LDS_PACKED;
    double* Wi8 = Tbuf + TRINV_T_DOUBLES;
    static_assert(POTRF_LDS_PACKED + TRINV_T_DOUBLES + 8 * LEAF_DOUBLES <= PB * SMALL_WP, "scratch fits behind the packed block");
    if (tid == 0) fail_s = -1;
    if (WIDE && n > 64) lml_small_build<4, 32>(m, n, d, eps, kernel);
    else if (WIDE) lml_small_build<2, 32>(m, n, d, eps, kernel);
    else if (n <= 32) lml_small_build<2, 16>(m, n, d, eps, kernel);
    else lml_small_build<4, 16>(m, n, d, eps, kernel);
    __syncthreads();
    const int ntl = (n + 15) >> 4;
    potrf128_mfma(m.Wl, Wi8, &fail_s, ntl);
    const int fail = fail_s;
    if (tid < n) m.dg[tid] = m.Wl[tri_off(tid) + tid];
    __syncthreads();                           // diag L is saved before the inverse overwrites it
    trinv128_mfma(m.Wl, Wi8, Tbuf, ntl);
    if (WIDE && n > 64) lml_small_finish<4, 32>(m, n, d, th, kernel, want_grad, fail, out_nll, out_grad, out_info);
    else if (WIDE) lml_small_finish<2, 32>(m, n, d, th, kernel, want_grad, fail, out_nll, out_grad, out_info);
    else if (n <= 32) lml_small_finish<2, 16>(m, n, d, th, kernel, want_grad, fail, out_nll, out_grad, out_info);
    else lml_small_finish<4, 16>(m, n, d, th, kernel, want_grad, fail, out_nll, out_grad, out_info);
}

template <bool WIDE>
__global__ __launch_bounds__(WIDE ? SMALL_WIDE_THREADS : POTRF_THREADS) void k_lml_small(const double* __restrict__ Xraw, int xpitch,
                                                             const double* __restrict__ y, int n, int d,
                                                             const double* __restrict__ thetas,
                                                             const double* __restrict__ sigmas, int kernel, int want_grad,
                                                             double* __restrict__ out_nll, double* __restrict__ out_grad,
                                                             int* __restrict__ out_info) {
    __shared__ double lds[SMALL_LDS_DOUBLES];
    const int b = blockIdx.x;
    lml_small_body<WIDE>(lds, Xraw, xpitch, y, n, d, thetas + (long)b * d, sigmas[b], kernel, want_grad, out_nll + b,
                   out_grad + (long)b * d, out_info + b);
}

// The reference's descent on -logL (conj_gradient_descent, BayesOpt_v4.ipynb:31-87, steprule other than 'backtrack') for
// n <= 128, one workgroup per start, every iteration inside the kernel: a multi-start fit is ONE launch instead of one
// launch, two copies and a host round trip per iteration.  Per iteration k = 1 .. max_its, as the reference:
//   dstep = (1 - beta) dtemp + beta dir(grad),  dir = sign(grad) sqrt(d) (norm 0, 'component'), grad / |grad| (1, 'full'), grad (2)
//   wn = w - alpha dstep;  steprule 1 ('diminishing'): alpha = 1 / k from the next iteration on
//   evaluate at wn: not positive definite -> record (wn, inf) and stop; else accept (w, grad, cost, dtemp = dstep), record,
//   and stop with exitflag 1 once |wn - w| < 1e-10.
// hist_w [(r (max_its + 1) + i) d + t], hist_c [r (max_its + 1) + i]: the iterates of start r and their costs (inf where the
// matrix was not positive definite), count[r] of them; flag[r] the exit flag.  Update arithmetic without fused
// multiply-adds, operation for operation as NumPy evaluates the reference's expressions (the host loop of the larger sizes).
template <bool WIDE>
__global__ __launch_bounds__(WIDE ? SMALL_WIDE_THREADS : POTRF_THREADS) void k_descent_small(const double* __restrict__ Xraw, int xpitch,
                                                                 const double* __restrict__ y, int n, int d,
                                                                 const double* __restrict__ W0, double sg, int kernel, double alpha,
                                                                 double beta, int norm, int steprule, int max_its,
                                                                 double* __restrict__ hist_w, double* __restrict__ hist_c,
                                                                 int* __restrict__ count, int* __restrict__ flag) {
    __shared__ double lds[SMALL_LDS_DOUBLES];
    __shared__ double w_s[GRAD_D], wn_s[GRAD_D], g_s[GRAD_D], gn_s[GRAD_D], dtemp_s[GRAD_D], dstep_s[GRAD_D];
    __shared__ double nll_s, moved_s;
    __shared__ int info_s, active_s, unusable_s;
    const int tid = threadIdx.x, r = blockIdx.x;
    const long H = (long)max_its + 1;
    if (tid < d) {
        w_s[tid] = W0[(long)r * d + tid];
        dtemp_s[tid] = 0.0;
    }
    __syncthreads();
    lml_small_body<WIDE>(lds, Xraw, xpitch, y, n, d, w_s, sg, kernel, 1, &nll_s, g_s, &info_s);
    __syncthreads();
    if (tid < d) hist_w[(r * H) * d + tid] = w_s[tid];
    if (tid == 0) {
        hist_c[r * H] = (info_s == 0) ? nll_s : INFINITY;
        active_s = (info_s == 0);
        count[r] = 1;
        flag[r] = 0;
    }
    __syncthreads();
    double alpha_r = alpha;
    int cnt = 1;
    for (int k = 1; k <= max_its && active_s; ++k) {
        if (tid == 0) {
#pragma clang fp contract(off)                      // no fused multiply-adds here: every product and sum rounds as NumPy's does
            // sums of squares in the order of NumPy's add.reduce over a contiguous axis (what np.linalg.norm(., axis=1) runs):
            // below eight terms one after the other; from eight on, eight running sums over blocks of eight, joined as
            // ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)), then the leftover terms one by one
            double sqv[GRAD_D];
            auto numpy_sum = [&](int m) {
                if (m < 8) {
                    double acc = 0.0;
                    for (int t = 0; t < m; ++t) acc = acc + sqv[t];
                    return acc;
                }
                double r8[8];
                for (int j = 0; j < 8; ++j) r8[j] = sqv[j];
                int i = 8;
                for (; i < m - (m % 8); i += 8)
                    for (int j = 0; j < 8; ++j) r8[j] = r8[j] + sqv[i + j];
                double acc = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
                for (; i < m; ++i) acc = acc + sqv[i];
                return acc;
            };
            double gnorm = 0.0;
            if (norm == 1) {
                for (int t = 0; t < d; ++t) sqv[t] = g_s[t] * g_s[t];
                gnorm = sqrt(numpy_sum(d));
            }
            const double sqd = sqrt((double)d), omb = 1.0 - beta;
            int unusable = 0;
            for (int t = 0; t < d; ++t) {
                const double g = g_s[t];
                double dir;
                if (norm == 0) {
                    const double bs = beta * ((g != g) ? g : (double)((g > 0.0) - (g < 0.0)));
                    dir = bs * sqd;
                } else if (norm == 1) {
                    const double gn = g / gnorm;
                    dir = beta * gn;
                } else {
                    dir = beta * g;
                }
                const double mom = omb * dtemp_s[t];
                const double ds = mom + dir;
                const double stepv = alpha_r * ds;
                const double wn = w_s[t] - stepv;
                dstep_s[t] = ds;
                wn_s[t] = wn;
                const double df = wn - w_s[t];
                sqv[t] = df * df;
                unusable |= !(fabs(wn) < INFINITY) || wn == 0.0;
            }
            moved_s = sqrt(numpy_sum(d));
            unusable_s = unusable;
            if (unusable) info_s = 1;
        }
        if (steprule == 1) alpha_r = 1.0 / (double)k;
        __syncthreads();
        // an iterate that is not a set of finite, non-zero length scales (0/0 from a zero gradient under the 'full' norm, an
        // overflow) is not evaluated: the start ends there like one whose matrix is not positive definite
        if (!unusable_s) lml_small_body<WIDE>(lds, Xraw, xpitch, y, n, d, wn_s, sg, kernel, 1, &nll_s, gn_s, &info_s);
        __syncthreads();
        const bool ok = info_s == 0;
        if (tid < d) {
            hist_w[(r * H + cnt) * d + tid] = wn_s[tid];
            if (ok) {
                w_s[tid] = wn_s[tid];
                g_s[tid] = gn_s[tid];
                dtemp_s[tid] = dstep_s[tid];
            }
        }
        if (tid == 0) {
            hist_c[r * H + cnt] = ok ? nll_s : INFINITY;
            count[r] = cnt + 1;
            if (!ok) active_s = 0;
            else if (moved_s < 1e-10) {
                flag[r] = 1;
                active_s = 0;
            }
        }
        cnt += 1;
        __syncthreads();
    }
}

// Sum of m <= GRAD_D terms in the order of NumPy's add.reduce over a contiguous axis: below eight terms one after the other;
// from eight on, eight running sums over blocks of eight, joined as a tree, then the leftover terms one by one.
__device__ __forceinline__ double numpy_order_sum(const double* v, int m) {
#pragma clang fp contract(off)
    if (m < 8) {
        double acc = 0.0;
        for (int t = 0; t < m; ++t) acc = acc + v[t];
        return acc;
    }
    double r8[8];
    for (int j = 0; j < 8; ++j) r8[j] = v[j];
    int i = 8;
    for (; i < m - (m % 8); i += 8)
        for (int j = 0; j < 8; ++j) r8[j] = r8[j] + v[i + j];
    double acc = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
    for (; i < m; ++i) acc = acc + v[i];
    return acc;
}

// The 'backtrack' step rule of the same descent (BayesOpt_v4.ipynb:62-71), one workgroup per start, every trial point
// evaluated inside the kernel.  Per outer iteration, as the reference:
//   wtemp = w;  w = w - alpha d;  h = g(wtemp) + grad(wtemp).(w - wtemp) + |w - wtemp|^2 / (2 alpha)
//   while g(w) > h (at most 200 rounds):  alpha *= 0.8;  w = w - alpha d;  h = (the same expression: old wtemp, new alpha);  wtemp = w
//   w = w - alpha d            (one more step after the loop; alpha stays shrunk for the following iterations)
// A trial point whose matrix is not positive definite, or that is no longer a set of finite non-zero length scales, ends the
// start: the iterate after the closing step is recorded with cost +inf.  The walk is a three-phase state machine (0 first
// trial, 1 a round of the shrinking loop, 2 the closing step) around ONE evaluation site; thread 0 keeps the books without
// fused multiply-adds, in the order NumPy evaluates the expressions above over a batch of starts.
template <bool WIDE>
__global__ __launch_bounds__(WIDE ? SMALL_WIDE_THREADS : POTRF_THREADS) void k_descent_backtrack_small(
    const double* __restrict__ Xraw, int xpitch, const double* __restrict__ y, int n, int d, const double* __restrict__ W0, double sg,
    int kernel, double alpha, double beta, int norm, int max_its, double* __restrict__ hist_w, double* __restrict__ hist_c,
    int* __restrict__ count, int* __restrict__ flag) {
    __shared__ double lds[SMALL_LDS_DOUBLES];
    __shared__ double w_s[GRAD_D], wn_s[GRAD_D], g_s[GRAD_D], gn_s[GRAD_D], dtemp_s[GRAD_D], dstep_s[GRAD_D], wt_s[GRAD_D], gt_s[GRAD_D];
    __shared__ double nll_s, cost_s, ft_s, h_s, alpha_s;
    __shared__ int info_s, active_s, eval_s, phase_s, failed_s, rounds_s, k_s, cnt_s;
    const int tid = threadIdx.x, r = blockIdx.x;
    const long H = (long)max_its + 1;
    if (tid < d) {
        w_s[tid] = W0[(long)r * d + tid];
        dtemp_s[tid] = 0.0;
    }
    __syncthreads();
    lml_small_body<WIDE>(lds, Xraw, xpitch, y, n, d, w_s, sg, kernel, 1, &nll_s, g_s, &info_s);
    __syncthreads();
    if (tid < d) hist_w[(r * H) * d + tid] = w_s[tid];
    if (tid == 0) {
        cost_s = (info_s == 0) ? nll_s : INFINITY;
        hist_c[r * H] = cost_s;
        active_s = (info_s == 0) && max_its >= 1;
        count[r] = 1;
        flag[r] = 0;
        alpha_s = alpha;
        phase_s = 0;
        k_s = 1;
        cnt_s = 1;
    }
    __syncthreads();
    while (active_s) {
        if (tid == 0) {
#pragma clang fp contract(off)
            double sqv[GRAD_D];
            const int phase = phase_s;
            if (phase == 0) {
                double gnorm = 0.0;
                if (norm == 1) {
                    for (int t = 0; t < d; ++t) sqv[t] = g_s[t] * g_s[t];
                    gnorm = sqrt(numpy_order_sum(sqv, d));
                }
                const double sqd = sqrt((double)d), omb = 1.0 - beta;
                for (int t = 0; t < d; ++t) {
                    const double g = g_s[t];
                    double dir;
                    if (norm == 0) {
                        const double bs = beta * ((g != g) ? g : (double)((g > 0.0) - (g < 0.0)));
                        dir = bs * sqd;
                    } else if (norm == 1) {
                        const double gn = g / gnorm;
                        dir = beta * gn;
                    } else {
                        dir = beta * g;
                    }
                    const double mom = omb * dtemp_s[t];
                    dstep_s[t] = mom + dir;
                    wt_s[t] = w_s[t];
                    gt_s[t] = g_s[t];
                    wn_s[t] = w_s[t];
                }
                ft_s = cost_s;
                failed_s = 0;
                rounds_s = 0;
            } else if (phase == 1) {
                alpha_s = alpha_s * 0.8;
            }
            const double a = alpha_s;
            int unusable = 0;
            for (int t = 0; t < d; ++t) {
                const double stepv = a * dstep_s[t];
                const double wn = wn_s[t] - stepv;
                wn_s[t] = wn;
                unusable |= !(fabs(wn) < INFINITY) || wn == 0.0;
            }
            if (phase == 1) {                   // the bound the coming evaluation is held against: old wtemp, its cost and gradient
                for (int t = 0; t < d; ++t) {
                    const double df = wn_s[t] - wt_s[t];
                    sqv[t] = gt_s[t] * df;
                }
                const double s1 = numpy_order_sum(sqv, d);
                for (int t = 0; t < d; ++t) {
                    const double df = wn_s[t] - wt_s[t];
                    sqv[t] = df * df;
                }
                const double s2 = numpy_order_sum(sqv, d);
                const double two_a = 2.0 * a;
                h_s = (ft_s + s1) + s2 / two_a;
            }
            const int skip = unusable || (phase == 2 && failed_s);
            if (unusable) failed_s = 1;
            eval_s = !skip;
            info_s = skip ? 1 : 0;
        }
        __syncthreads();
        // the gradient at the first trial point is never used (either the loop replaces it or the closing step moves on)
        if (eval_s) lml_small_body<WIDE>(lds, Xraw, xpitch, y, n, d, wn_s, sg, kernel, phase_s != 0, &nll_s, gn_s, &info_s);
        __syncthreads();
        if (tid == 0) {
#pragma clang fp contract(off)
            double sqv[GRAD_D];
            const int phase = phase_s;
            const bool bad = info_s != 0;
            if (phase == 0) {
                if (bad) {
                    failed_s = 1;
                    phase_s = 2;
                } else {
                    for (int t = 0; t < d; ++t) {
                        const double df = wn_s[t] - wt_s[t];
                        sqv[t] = gt_s[t] * df;
                    }
                    const double s1 = numpy_order_sum(sqv, d);
                    for (int t = 0; t < d; ++t) {
                        const double df = wn_s[t] - wt_s[t];
                        sqv[t] = df * df;
                    }
                    const double s2 = numpy_order_sum(sqv, d);
                    const double two_a = 2.0 * alpha_s;
                    h_s = (ft_s + s1) + s2 / two_a;
                    phase_s = (nll_s > h_s) ? 1 : 2;
                }
            } else if (phase == 1) {
                rounds_s += 1;
                if (bad) {
                    failed_s = 1;
                    phase_s = 2;
                } else {
                    for (int t = 0; t < d; ++t) {
                        wt_s[t] = wn_s[t];
                        gt_s[t] = gn_s[t];
                    }
                    ft_s = nll_s;
                    phase_s = (nll_s > h_s && rounds_s < 200) ? 1 : 2;
                }
            } else {
                const int cnt = cnt_s;
                for (int t = 0; t < d; ++t) hist_w[(r * H + cnt) * d + t] = wn_s[t];
                if (failed_s || bad) {
                    hist_c[r * H + cnt] = INFINITY;
                    active_s = 0;
                } else {
                    for (int t = 0; t < d; ++t) {
                        const double df = wn_s[t] - w_s[t];
                        sqv[t] = df * df;
                    }
                    const double moved = sqrt(numpy_order_sum(sqv, d));
                    for (int t = 0; t < d; ++t) {
                        w_s[t] = wn_s[t];
                        g_s[t] = gn_s[t];
                        dtemp_s[t] = dstep_s[t];
                    }
                    cost_s = nll_s;
                    hist_c[r * H + cnt] = nll_s;
                    if (moved < 1e-10) {
                        flag[r] = 1;
                        active_s = 0;
                    }
                }
                count[r] = cnt + 1;
                cnt_s = cnt + 1;
                k_s += 1;
                if (k_s > max_its) active_s = 0;
                phase_s = 0;
            }
        }
        __syncthreads();
    }
}

// Per-lane device slot of the likelihood sweep: everything that changes from one setting to the next reaches the
// kernels through it, so the launch sequence of a setting has constant arguments and can be replayed from a graph.
struct LmlSlot {
    double theta[MAX_D];
    double eps;
    int cur;      // index of the setting in the batch
    int info;     // factorisation status of this setting
};
static_assert(sizeof(LmlSlot) % sizeof(double) == 0, "slots are strided in doubles");

__global__ void k_pick_setting(int b0, int B, const double* __restrict__ thetas, const double* __restrict__ sigmas, int d, int n,
                               LmlSlot* __restrict__ slots) {
    // slot blockIdx.x takes setting b0 + blockIdx.x; past the end of the batch it recomputes the last setting and is
    // marked inactive (cur = -1: nothing is stored for it)
    LmlSlot* slot = slots + blockIdx.x;
    const int bq = b0 + blockIdx.x;
    const int b = bq < B ? bq : B - 1;
    const int t = threadIdx.x;
    if (t < d) slot->theta[t] = thetas[(long)b * d + t];
    if (t == 0) {
        const double sg = sigmas[b];
        slot->eps = (sg == 0.0) ? 1e-15 * (double)n : sg * sg;
        slot->cur = bq < B ? bq : -1;
        slot->info = 0;
    }
}

// every slot's observation vector is the same padded y: dst[s*n_pad + i] = src[i], s = blockIdx.y
__global__ __launch_bounds__(256) void k_replicate_vec(const double* __restrict__ src, double* __restrict__ dst, int n_pad) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_pad) dst[(long)blockIdx.y * n_pad + i] = src[i];
}

// b = slot->cur: out_nll[b] = -logL (NaN when the factorisation failed), out_grad[b*d + t] = -1/2 * sum_tiles gpart / theta_t,
// out_info[b] = the factorisation status
__global__ __launch_bounds__(256) void k_lml_store(const double* __restrict__ params, const LmlSlot* __restrict__ slot,
                                                   const double* __restrict__ gpart, int ntiles, int d, int want_grad,
                                                   double* __restrict__ out_nll, double* __restrict__ out_grad,
                                                   int* __restrict__ out_info, BatchStride bs) {
    __shared__ double r[256];
    params += blockIdx.z * bs.par;
    slot += blockIdx.z;
    gpart += blockIdx.z * bs.gp;
    const int tid = threadIdx.x;
    const int b = slot->cur;
    if (b < 0) return;
    const bool bad = (slot->info != 0);
    if (tid == 0) {
        out_nll[b] = bad ? NAN : params[P_NLL];
        out_info[b] = slot->info;
    }
    if (!want_grad) return;
    for (int t = 0; t < d; ++t) {
        double s = 0.0;
        for (int i = tid; i < ntiles; i += 256) s += gpart[(long)i * GRAD_D + t];
        r[tid] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) r[tid] += r[tid + o];
            __syncthreads();
        }
        if (tid == 0) out_grad[(long)b * d + t] = bad ? NAN : -0.5 * r[0] * params[P_INVTHETA + t];
        __syncthreads();
    }
}

}  // namespace gpbo
