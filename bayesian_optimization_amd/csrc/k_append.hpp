// Device kernels of libgpbo (gfx950), part of kernels.hpp: incremental append of one training point.
#pragma once

namespace gpbo {

// ---------------------------------------------------------------------------------------
// Incremental append of one training point (single-factor variant, theta frozen).
// scal (device doubles): [0] d_new  [1] z_new  [2] ll = l.l  [3] lz = l.z
enum { A_DNEW = 0, A_ZNEW = 1, A_LL = 2, A_LZ = 3 };

// kvec[i] = k(x_i, x_new) for i < n (0 for the pad); also records the new scaled point as column n of Xts.
// First launch of a loop step: the new scaled point arrives as a kernel argument and is left in device memory (xnew_out)
// for the kernels that follow, and the status word is cleared -- no launch, copy or fill of their own for either.
__device__ __forceinline__ void append_kvec_at(int i, double* __restrict__ Xts, int n, int n_pad, int d, const PointArg& p,
                                               double* __restrict__ xnew_out, int* __restrict__ info, double* __restrict__ kvec,
                                               int kernel) {
    if (i < MAX_D) xnew_out[i] = p.x[i];
    if (i == 0) *info = 0;
    if (i >= n_pad) return;
    double v = 0.0;
    if (i < n) {
        double Dq = 0.0;
        for (int t = 0; t < d; ++t) {
            const double df = Xts[(long)t * n_pad + i] - p.x[t];
            Dq += df * df;
        }
        v = kern_eval(Dq, kernel);
    }
    kvec[i] = v;
    if (i == n)
        for (int t = 0; t < d; ++t) Xts[(long)t * n_pad + n] = p.x[t];
}

__global__ __launch_bounds__(256) void k_append_kvec(double* __restrict__ Xts, int n, int n_pad, int d, PointArg p,
                                                     double* __restrict__ xnew_out, int* __restrict__ info,
                                                     double* __restrict__ kvec, int kernel) {
    append_kvec_at(blockIdx.x * 256 + threadIdx.x, Xts, n, n_pad, d, p, xnew_out, info, kvec, kernel);
}

// l = W[0:n,0:n] * kvec  (one wave per row, lanes along the row)
__device__ __forceinline__ void append_lvec_row(int row, int lane, const double* __restrict__ W, long ld, int n,
                                                const double* __restrict__ kvec, double* __restrict__ lvec) {
    if (row >= n) return;
    double s = 0.0;
    for (int j = lane; j <= row; j += 64) s += W[(long)row * ld + j] * kvec[j];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) lvec[row] = s;
}

// Four rows (row0, row0 + 16, row0 + 32, row0 + 48) by one wave, their loads in flight together: for a single workgroup walking
// over all rows the time is the number of memory round trips.  Each row's sum is formed exactly as in append_lvec_row.
__device__ __forceinline__ void append_lvec_rows4(int row0, int lane, const double* __restrict__ W, long ld, int n,
                                                  const double* __restrict__ kvec, double* __restrict__ lvec) {
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    const int last = (row0 + 48 < n ? row0 + 48 : n - 1);
    for (int j = lane; j <= last; j += 64) {
        const double kj = kvec[j];
        double wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = row0 + 16 * u;
            wv[u] = (row < n && j <= row) ? W[(long)row * ld + j] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = row0 + 16 * u;
            if (row < n && j <= row) s[u] += wv[u] * kj;
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        double t = s[u];
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if (lane == 0 && row0 + 16 * u < n) lvec[row0 + 16 * u] = t;
    }
}

__global__ __launch_bounds__(256) void k_append_lvec(const double* __restrict__ W, long ld, int n, const double* __restrict__ kvec,
                                                     double* __restrict__ lvec) {
    append_lvec_row(blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63, W, ld, n, kvec, lvec);
}

// One workgroup: d_new = sqrt(1 + eps - l.l), z_new = (y_new - l.z)/d_new, scalars of the enlarged model,
// new rows of L (optional) and the diagonal of W.  n is the size BEFORE the append.
// (tid < 256 and `active`: the 256 threads that do the work; every thread of the workgroup calls it, for the barriers)
__device__ __forceinline__ void append_scalars_phase(int tid, bool active, double* r1, double* r2, const double* __restrict__ lvec,
                                                     double* __restrict__ zfw, double* __restrict__ yv, int n, double y_new, double eps,
                                                     double* __restrict__ scal, double* __restrict__ params, double* __restrict__ L,
                                                     double* __restrict__ W, long ld, int* __restrict__ info) {
    double ll = 0.0, lz = 0.0;
    if (active) {
        for (int i = tid; i < n; i += 256) {
            ll += lvec[i] * lvec[i];
            lz += lvec[i] * zfw[i];
        }
        r1[tid] = ll;
        r2[tid] = lz;
    }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (active && tid < o) {
            r1[tid] += r1[tid + o];
            r2[tid] += r2[tid + o];
        }
        __syncthreads();
    }
    if (!active) return;
    const double dd = 1.0 + eps - r1[0];
    const double dnew = sqrt(dd);
    const double znew = (y_new - r2[0]) / dnew;
    if (L)
        for (int i = tid; i < n; i += 256) L[(long)n * ld + i] = lvec[i];
    if (tid == 0) {
        if (!(dd > 0.0)) atomicCAS(info, 0, n + 1);
        scal[A_DNEW] = dnew;
        scal[A_ZNEW] = znew;
        scal[A_LL] = r1[0];
        scal[A_LZ] = r2[0];
        zfw[n] = znew;
        yv[n] = y_new;
        if (L) L[(long)n * ld + n] = dnew;
        W[(long)n * ld + n] = 1.0 / dnew;
        const int n1 = n + 1;
        const double ya = params[P_YA] + znew * znew;          // y^T K^-1 y = |L^-1 y|^2
        const double sumlog = params[P_SUMLOG] + log(dnew);
        const double theta1 = ya / n1;
        params[P_YA] = ya;
        params[P_SUMLOG] = sumlog;
        params[P_THETA1] = theta1;
        params[P_C0] = theta1;
        params[P_C1] = theta1;
        params[P_YMAX] = fmax(params[P_YMAX], y_new);
        params[P_NLL] = 0.5 * n1 * (log(2.0 * M_PI) + 1.0) + 0.5 * n1 * log(theta1) + sumlog;
    }
}

__global__ __launch_bounds__(256) void k_append_scalars(const double* __restrict__ lvec, double* __restrict__ zfw,
                                                        double* __restrict__ yv, int n, double y_new, double eps,
                                                        double* __restrict__ scal, double* __restrict__ params,
                                                        double* __restrict__ L, double* __restrict__ W, long ld,
                                                        int* __restrict__ info) {
    __shared__ double r1[256];
    __shared__ double r2[256];
    append_scalars_phase(threadIdx.x, true, r1, r2, lvec, zfw, yv, n, y_new, eps, scal, params, L, W, ld, info);
}

// New row of W and the enlarged alpha, per column j <= n (n = size before the append):
//   W[n][j] = -(1/d_new) * sum_{i=j..n-1} l_i W[i][j]  (j < n; W[n][n] is already in place),
//   alpha1[j] = alpha[j] = sum_{i=j..n} W[i][j] z_i   (W^T z over the enlarged model; the mean of this variant uses alpha = alpha1).
// Both sums are always formed as eight partial sums over (i - j) mod 8, joined pairwise -- by one thread here, by eight
// threads in k_append_wrow: the same arithmetic, so the two routes of an append update the factor bit for bit alike.
__device__ __forceinline__ double join8(const double (&p)[8]) { return ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7])); }

// Neither result is read by the sweep over the candidates, so the first workgroup(s) of the streaming sweep's launch run
// this routine beside it instead of launches of their own in front of it.
__device__ __forceinline__ void append_wrow_alpha(int blk, double* __restrict__ W, long ld, int n, const double* __restrict__ lvec,
                                                  const double* __restrict__ scal, const double* __restrict__ zfw,
                                                  double* __restrict__ alpha1, double* __restrict__ alpha) {
    const int j = blk * 256 + threadIdx.x;
    if (j > n) return;
    double a = 0.0, wn;
    if (j < n) {
        double ps[8], pa[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) ps[q] = pa[q] = 0.0;
        int i = j;
        for (; i + 8 <= n; i += 8) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double w = W[(long)(i + q) * ld + j];
                ps[q] = fma(lvec[i + q], w, ps[q]);
                pa[q] = fma(w, zfw[i + q], pa[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (i + q < n) {
                const double w = W[(long)(i + q) * ld + j];
                ps[q] = fma(lvec[i + q], w, ps[q]);
                pa[q] = fma(w, zfw[i + q], pa[q]);
            }
        wn = -join8(ps) / scal[A_DNEW];
        a = join8(pa);
        W[(long)n * ld + j] = wn;
    } else {
        wn = W[(long)n * ld + n];
    }
    a = fma(wn, zfw[n], a);
    alpha1[j] = a;
    alpha[j] = a;
}

// The same update in a launch of its own (the lattice route of an append has no streaming sweep to ride along with): 32
// columns x 8 threads per workgroup, thread q of a column forms partial sum q.
// (blk < 0: a thread group with no columns of its own, there for the barrier only)
__device__ __forceinline__ void append_wrow_block(int blk, int t256, double (*ss)[33], double (*sa)[33], double* __restrict__ W, long ld,
                                                  int n, const double* __restrict__ lvec, const double* __restrict__ scal,
                                                  const double* __restrict__ zfw, double* __restrict__ alpha1,
                                                  double* __restrict__ alpha) {
    const int c = t256 & 31, q = t256 >> 5;
    const int j = blk < 0 ? n + 1 : blk * 32 + c;
    double ps = 0.0, pa = 0.0;
    if (j < n) {
        int i = j + q;
        for (; i + 24 < n; i += 32) {        // four loads in flight; the sums in the same order as one at a time
            double w4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) w4[u] = W[(long)(i + 8 * u) * ld + j];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ps = fma(lvec[i + 8 * u], w4[u], ps);
                pa = fma(w4[u], zfw[i + 8 * u], pa);
            }
        }
        for (; i < n; i += 8) {
            const double w = W[(long)i * ld + j];
            ps = fma(lvec[i], w, ps);
            pa = fma(w, zfw[i], pa);
        }
    }
    ss[q][c] = ps;
    sa[q][c] = pa;
    __syncthreads();
    if (q != 0 || j > n) return;
    double a = 0.0, wn;
    if (j < n) {
        const double s8[8] = {ss[0][c], ss[1][c], ss[2][c], ss[3][c], ss[4][c], ss[5][c], ss[6][c], ss[7][c]};
        const double a8[8] = {sa[0][c], sa[1][c], sa[2][c], sa[3][c], sa[4][c], sa[5][c], sa[6][c], sa[7][c]};
        wn = -join8(s8) / scal[A_DNEW];
        a = join8(a8);
        W[(long)n * ld + j] = wn;
    } else {
        wn = W[(long)n * ld + n];
    }
    a = fma(wn, zfw[n], a);
    alpha1[j] = a;
    alpha[j] = a;
}

__global__ __launch_bounds__(256) void k_append_wrow(double* __restrict__ W, long ld, int n, const double* __restrict__ lvec,
                                                     const double* __restrict__ scal, const double* __restrict__ zfw,
                                                     double* __restrict__ alpha1, double* __restrict__ alpha) {
    __shared__ double ss[8][33], sa[8][33];
    append_wrow_block((int)blockIdx.x, threadIdx.x, ss, sa, W, ld, n, lvec, scal, zfw, alpha1, alpha);
}

// The HBM-streaming sweep: per candidate j
//   v = (k(xs_j, x_new) - sum_{k<n} l_k V[k][j]) / d_new ;  V[n][j] = v ;  s_j += v^2 ;  mu_j += v z_new ;
//   var_j = theta1 (1 - s_j), < 1e-60 -> 0.
// A thread owns VEC adjacent candidates (VEC = 2: 16-byte loads, 1 KiB per wave instruction; needs m even) and keeps
// eight rows of V in flight; V is read once and never again, so the loads are non-temporal.  The dot product of a
// candidate is always eight partial sums over k mod 8, combined pairwise: the same arithmetic in both forms, so a
// candidate's result does not depend on how the set is split over ranks.
template <int D, int VEC>
__global__ __launch_bounds__(256) void k_append_sweep(const double* __restrict__ Xs, long m, int d_rt,
                                                      const double* __restrict__ xnew, const double* __restrict__ params,
                                                      const double* __restrict__ lvec, const double* __restrict__ scal,
                                                      double* __restrict__ V, long ldv, int n, double* __restrict__ sres,
                                                      double* __restrict__ mu, double* __restrict__ var, int kernel,
                                                      double* __restrict__ bmax, int extra, double* __restrict__ W, long ldw,
                                                      const double* __restrict__ zfw, double* __restrict__ alpha1,
                                                      double* __restrict__ alpha) {
    // the first `extra` workgroups complete the factor update (new row of W, alpha); the others stream V
    if ((int)blockIdx.x < extra) {
        append_wrow_alpha((int)blockIdx.x, W, ldw, n, lvec, scal, zfw, alpha1, alpha);
        return;
    }
    const long blk = (long)blockIdx.x - extra;
    const long j = (blk * 256 + threadIdx.x) * VEC;
    double mx = -INFINITY;           // the largest updated mean among this thread's candidates
    if (j < m) {
    const int d = (D > 0) ? D : d_rt;
    double dot[VEC][8];
#pragma unroll
    for (int c = 0; c < VEC; ++c)
#pragma unroll
        for (int u = 0; u < 8; ++u) dot[c][u] = 0.0;
    const double* Vj = V + j;
    int k = 0;
    for (; k + 8 <= n; k += 8) {
        double vv[8][VEC];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (VEC == 2) {
                const f64x2 t = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(Vj + (long)(k + u) * ldv));
                vv[u][0] = t.x;
                vv[u][VEC - 1] = t.y;
            } else {
                vv[u][0] = __builtin_nontemporal_load(Vj + (long)(k + u) * ldv);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double l = lvec[k + u];
#pragma unroll
            for (int c = 0; c < VEC; ++c) dot[c][u] = fma(l, vv[u][c], dot[c][u]);
        }
    }
    for (int u = 0; k < n; ++k, ++u) {
        const double l = lvec[k];
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            const double t = Vj[(long)k * ldv + c];
#pragma unroll
            for (int w = 0; w < 8; ++w)
                if (w == u) dot[c][w] = fma(l, t, dot[c][w]);
        }
    }
    const double dnew = scal[A_DNEW], znew = scal[A_ZNEW];
    const double c0 = params[P_C0], c1 = params[P_C1];
    double vout[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        double Dq = 0.0;
#pragma unroll
        for (int t = 0; t < ((D > 0) ? D : MAX_D); ++t)
            if (t < d) {
                const double df = Xs[(j + c) * d + t] * params[P_INVTHETA + t] - xnew[t];
                Dq += df * df;
            }
        const double sum = ((dot[c][0] + dot[c][1]) + (dot[c][2] + dot[c][3])) + ((dot[c][4] + dot[c][5]) + (dot[c][6] + dot[c][7]));
        const double v = (kern_eval(Dq, kernel) - sum) / dnew;
        vout[c] = v;
        const double s = sres[j + c] + v * v;
        sres[j + c] = s;
        const double mnew = mu[j + c] + v * znew;
        mu[j + c] = mnew;
        mx = max_nan(mx, mnew);
        double vr = c0 - c1 * s;
        if (vr < 1e-60) vr = 0.0;
        var[j + c] = vr;
    }
    if (VEC == 2) {
        f64x2 t;
        t.x = vout[0];
        t.y = vout[VEC - 1];
        *reinterpret_cast<f64x2*>(V + (long)n * ldv + j) = t;
    } else {
        V[(long)n * ldv + j] = vout[0];
    }
    }
    // max(mu) of the workgroup's candidates (stage 1 of the reduction every iteration needs, free of a pass over mu)
    __shared__ double wm[4];
    for (int o = 32; o > 0; o >>= 1) mx = max_nan(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) bmax[blk] = max_nan(max_nan(wm[0], wm[1]), max_nan(wm[2], wm[3]));
}


// u = W v  (lower-triangular W, wave per row) and x = W^T u (thread per column), for q vectors stored
// column-after-column with pitch n_pad:  K^-1 v = W^T (W v).
__global__ __launch_bounds__(256) void k_tri_gemv_n(const double* __restrict__ W, long ld, int n, const double* __restrict__ v,
                                                    double* __restrict__ u, int n_pad) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.y;
    if (row >= n) return;
    double s = 0.0;
    for (int j = lane; j <= row; j += 64) s += W[(long)row * ld + j] * v[(long)q * n_pad + j];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) u[(long)q * n_pad + row] = s;
}
__global__ __launch_bounds__(256) void k_tri_gemv_t(const double* __restrict__ W, long ld, int n, const double* __restrict__ u,
                                                    double* __restrict__ x, int n_pad) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int q = blockIdx.y;
    if (j >= n) return;
    double s = 0.0;
    for (int i = j; i < n; ++i) s += W[(long)i * ld + j] * u[(long)q * n_pad + i];
    x[(long)q * n_pad + j] = s;
}


// K[i][j] = k(U_i, V_j) for two arbitrary point sets (row-major (p,d), (q,d)); direct differences.
__global__ __launch_bounds__(256) void k_cross_kernel(const double* __restrict__ U, long p, const double* __restrict__ V, long q,
                                                      int d, const double* __restrict__ invtheta, int kernel,
                                                      double* __restrict__ K) {
    const long j = (long)blockIdx.x * 64 + (threadIdx.x & 63);
    const long i = (long)blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= p || j >= q) return;
    double D = 0.0;
    for (int t = 0; t < d; ++t) {
        const double df = (U[i * d + t] - V[j * d + t]) * invtheta[t];
        D += df * df;
    }
    K[i * q + j] = kern_eval(D, kernel);
}

// kernfcn_v2 on explicit per-dimension |difference| matrices (the notebooks' Daa list): Dabs is [d][p*q].
__global__ __launch_bounds__(256) void k_kernel_from_absdiff(const double* __restrict__ Dabs, long pq, int d,
                                                             const double* __restrict__ invtheta, int kernel,
                                                             double* __restrict__ K) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= pq) return;
    double D = 0.0;
    for (int t = 0; t < d; ++t) {
        const double df = Dabs[(long)t * pq + e] * invtheta[t];
        D += df * df;
    }
    K[e] = kern_eval(D, kernel);
}

// out[0] = min, out[1] = max of diag(L), rows < n.  One workgroup.
__global__ __launch_bounds__(256) void k_diag_minmax(const double* __restrict__ L, long ld, int n, double* __restrict__ out) {
    __shared__ double lo[256], hi[256];
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
        out[0] = lo[0];
        out[1] = hi[0];
    }
}

// Synthetic candidates: element e of the row-major (m, d) set is the counter-based splitmix64 stream
//   u(seed, e0 + e) = (mix(seed, e0 + e) >> 11) * 2^-53  in [0, 1),
// integer arithmetic and an exact conversion, so a shard generated here is bit-identical to the same rows generated
// on the host.  count = m * d elements.
__global__ __launch_bounds__(256) void k_uniform_splitmix(double* __restrict__ out, long count, unsigned long long seed,
                                                          unsigned long long e0) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= count) return;
    unsigned long long z = (e0 + (unsigned long long)e + 1ull) * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    out[e] = (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

__global__ void k_set_scalar(double* dst, double v) { *dst = v; }
__global__ void k_copy_scalar(double* dst, const double* src) { *dst = *src; }
__global__ void k_copy4(double* dst, const double* src) {
    if (threadIdx.x < 4) dst[threadIdx.x] = src[threadIdx.x];
}

}  // namespace gpbo
