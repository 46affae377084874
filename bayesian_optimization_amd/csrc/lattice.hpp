// Incremental append on a candidate LATTICE (the grid the notebooks build with nested loops,
// BayesOpt/BayesOpt_v4.ipynb:705-714: every combination of per-axis coordinates, last axis fastest) with the RBF kernel.
//
// The new row of V = L^-1 K(X, X*) after appending training point n is
//     v_j = sum_{i <= n} W[n][i] k(x_i, x*_j),        W = L^-1 (its new row is part of the factor update anyway),
// and the RBF kernel is a product over the axes, k(x_i, x*_j) = prod_t e_t(i, g_t(j)) with
// e_t(i, g) = exp(-(axis_t[g] / theta_t - x_it / theta_t)^2 / 2).  Splitting the axes into a prefix (the slow ones) and a
// suffix, candidate j = p * Q + q, gives
//     v_(p,q) = sum_i A[p][i] B[i][q],   A[p][i] = W[n][i] prod_{t < split} e_t(i, g_t(p)),   B[i][q] = prod_{t >= split} e_t(i, g_t(q)):
// one P x Q x (n + 1) matrix product on the fp64 MFMA tile engine (P = Q = 4096, n <= 216 on a 64^4 lattice: 7 GFLOP)
// in place of streaming the resident V (8 n m bytes: 29 GB at n = 216), and V need not be kept at all.  The epilogue of
// the product is the rest of the update: s_j += v^2, mu_j += v z_new, max(mu) per workgroup (the variance theta1 (1 - s_j) is
// formed from s_j where it is used).
// Every candidate's sum runs over i in the same order whatever tile it sits in, so a lattice cut into slabs along its
// first axis (several contexts or ranks) reproduces the one-context numbers bit for bit.
#pragma once

namespace gpbo {

struct LatticeDesc {
    int d, split;          // axes 0 .. split-1 index the rows of A, axes split .. d-1 the columns of B
    int size[MAX_D];       // points per axis
    int off[MAX_D];        // start of axis t in the concatenated coordinate array
    long P, Q;             // number of prefix / suffix combinations (m = P Q)
};

// Candidate rows of the lattice, lexicographic, last axis fastest.
__global__ __launch_bounds__(256) void k_lattice_points(LatticeDesc L, const double* __restrict__ axes, double* __restrict__ Xs, long m) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    long idx = j;
    for (int t = L.d - 1; t >= 0; --t) {
        const int g = (int)(idx % L.size[t]);
        idx /= L.size[t];
        Xs[j * L.d + t] = axes[L.off[t] + g];
    }
}

// xts: the training point's coordinate on axis t, already divided by theta_t
__device__ __forceinline__ double lattice_factor(const LatticeDesc& L, const double* __restrict__ axes, const double* __restrict__ params,
                                                 int t, int g, double xts) {
    const double df = axes[L.off[t] + g] * params[P_INVTHETA + t] - xts;
    return exp_neg(-0.5 * df * df);
}

// Row i of the per-axis factor table of the prefix axes, E[(t * smax + g) * kcap + i], and of B (k-major: B[i * ldb + q],
// columns q >= Q zero), element idx of max(ldb, split * smax); x[t]: the scaled coordinates of training point i.
template <typename XT>
__device__ __forceinline__ void lattice_row_at(long idx, int i, const LatticeDesc& L, const double* __restrict__ axes,
                                               const double* __restrict__ params, const XT& x, double* __restrict__ E, int kcap, int smax,
                                               double* __restrict__ B, long ldb) {
    if (idx < ldb) {
        double v = 0.0;
        if (idx < L.Q) {
            v = 1.0;
            long r = idx;
            for (int t = L.d - 1; t >= L.split; --t) {
                const int g = (int)(r % L.size[t]);
                r /= L.size[t];
                v *= lattice_factor(L, axes, params, t, g, x(t));
            }
        }
        B[(long)i * ldb + idx] = v;
    }
    if (idx < (long)L.split * smax) {
        const int t = (int)(idx / smax), g = (int)(idx % smax);
        if (g < L.size[t]) E[((long)t * smax + g) * kcap + i] = lattice_factor(L, axes, params, t, g, x(t));
    }
}

// Rows i0 + blockIdx.y: at the first posterior for all training points, afterwards one row per append.
__global__ __launch_bounds__(256) void k_lattice_rows(LatticeDesc L, const double* __restrict__ axes, const double* __restrict__ params,
                                                      const double* __restrict__ Xts, long n_pad, int i0, double* __restrict__ E,
                                                      int kcap, int smax, double* __restrict__ B, long ldb) {
    const int i = i0 + (int)blockIdx.y;
    auto x = [&](int t) { return Xts[(long)t * n_pad + i]; };
    lattice_row_at((long)blockIdx.x * 256 + threadIdx.x, i, L, axes, params, x, E, kcap, smax, B, ldb);
}

// The incumbent an acquisition inside the update launch is to use, formed BEFORE that launch (extra workgroup of k_lattice_A).
// mode 0: the largest posterior mean after the update, which only the update itself can know -- but its location moves
// rarely (195 of 200 iterations of the 64^4 loop keep it), so the new mean of the candidate that held the maximum so far
// is formed here, bit for bit as the update will form it (an element of the fp64 MFMA product is the k-ordered chain of
// fused multiply-adds from zero: tests/probes/mfma_chain_probe.hip), and used as the incumbent; the launch that ends the step
// compares it with the maximum the update found and, where they differ, the step falls back to the separate acquisition
// pass.  mode 1: the largest observation (params[P_YMAX]).  mode 2: a given value.
struct SpecArgs {
    double* spec;            // [0] the incumbent (out); [1] bit pattern of an int64: index of the largest mean so far (in)
    int mode, have;          // have: spec[1] holds the index for the current posterior
    double given;
    const double* B;         // the k-major factor matrix of the suffix axes
    long ldb;
    const double* mu;
    const double* scal;
    const double* params;
    int kpad;
};

// A[p][i] = W[n][i] prod_{t < split} e_t(i, g_t(p)) for i <= n, zero for n < i < kpad (row-major, i contiguous).  With
// sp.spec != nullptr the launch has one workgroup more, which forms the incumbent (SpecArgs).
__global__ __launch_bounds__(256) void k_lattice_A(LatticeDesc L, const double* __restrict__ E, int kcap, int smax,
                                                   const double* __restrict__ wrow, int n, double* __restrict__ A, long lda, int kpad,
                                                   SpecArgs sp) {
    if (sp.spec && blockIdx.x == gridDim.x - 1) {
        __shared__ double sa[256], sb[256];
        __shared__ double chain;
        const int tid = threadIdx.x;
        if (sp.mode != 0 || !sp.have) {
            if (tid == 0) sp.spec[0] = sp.mode == 1 ? sp.params[P_YMAX] : (sp.mode == 2 ? sp.given : NAN);
            return;
        }
        const long js = reinterpret_cast<const long*>(sp.spec)[1];
        const long ps = js / L.Q, qs = js % L.Q;
        int g[MAX_D];
        long r = ps;
        for (int t = L.split - 1; t >= 0; --t) {
            g[t] = (int)(r % L.size[t]);
            r /= L.size[t];
        }
        if (tid == 0) chain = 0.0;
        for (int i0 = 0; i0 <= n; i0 += 256) {
            const int i = i0 + tid;
            double a = 0.0, b = 0.0;
            if (i <= n) {
                a = wrow[i];
                for (int t = 0; t < L.split; ++t) a *= E[((long)t * smax + g[t]) * kcap + i];
                b = sp.B[(long)i * sp.ldb + qs];
            }
            __syncthreads();
            sa[tid] = a;
            sb[tid] = b;
            __syncthreads();
            if (tid == 0) {
                double v = chain;
                const int cnt = (n + 1 - i0) < 256 ? (n + 1 - i0) : 256;
                for (int k = 0; k < cnt; ++k) v = fma(sa[k], sb[k], v);
                chain = v;
            }
        }
        if (tid == 0) {
            double v = chain;
            if (sp.kpad > n + 1) v = fma(0.0, 0.0, v);         // the zero padding of the product's k range (only a -0 would notice)
            sp.spec[0] = fma(v, sp.scal[A_ZNEW], sp.mu[js]);
        }
        return;
    }
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long p = idx / kpad;
    const int i = (int)(idx % kpad);
    if (p >= L.P) return;
    double v = 0.0;
    if (i <= n) {
        v = wrow[i];
        int g[MAX_D];
        long r = p;
        for (int t = L.split - 1; t >= 0; --t) {
            g[t] = (int)(r % L.size[t]);
            r /= L.size[t];
        }
        for (int t = 0; t < L.split; ++t) v *= E[((long)t * smax + g[t]) * kcap + i];
    }
    A[p * lda + i] = v;
}

// v = A B on 64 x 64 quarter tiles (four workgroups per CU, so that the product phase of one tile runs beside the
// read-modify-write phase of another: 0.174 against 0.206 ms per update on the 64^4 lattice with 128 x 128 tiles), then per
// candidate j = row * Q + col:
//   s_j += v^2 ; mu_j += v z_new ; the workgroup's max(mu) -> bmax[blockIdx.x].
// The variance theta1 (1 - s_j) is NOT written (8 of 40 bytes per candidate): the acquisition pass forms it from s_j, and
// the array is filled when somebody reads it (k_var_from_s).
__global__ __launch_bounds__(GEMM_THREADS, 4) void k_lattice_gemm_update(const double* __restrict__ A, long lda, const double* __restrict__ B,
                                                                           long ldb, int kpad, long P, long Q, int tiles_q,
                                                                           const double* __restrict__ scal,
                                                                           double* __restrict__ sres, double* __restrict__ mu,
                                                                           double* __restrict__ bmax) {
    const int ib = (int)blockIdx.x / tiles_q, jb = (int)blockIdx.x % tiles_q;
    __shared__ double lds[QUARTER_LDS_DOUBLES];
    f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
    // The old sums of this tile are requested BEFORE the product, so that their HBM latency runs under it (-4.5 % on the
    // kernel).  The old means are not: 16 more doubles per lane do not fit the 128 registers of four workgroups per CU
    // (40 spilled, +20 % per update), and three workgroups per CU with both prefetched lose 3 % against this form.
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double s_old[2][4][2];
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long row = (long)ib * 64 + (w >> 1) * 32 + ta * 16 + (lane >> 4) + 4 * r;
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                const long col = (long)jb * 64 + (w & 1) * 32 + tb * 16 + (lane & 15);
                const bool in = row < P && col < Q;
                const long j = in ? row * Q + col : 0;
                s_old[ta][r][tb] = in ? sres[j] : 0.0;
            }
        }
    gemm_tile_64x64<false>(A + (long)ib * QUARTER_M * lda, lda, B + (long)jb * 64, ldb, 0, kpad, acc, lds);
    const double znew = scal[A_ZNEW];
    double mx = -INFINITY;
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long row = (long)ib * 64 + (w >> 1) * 32 + ta * 16 + (lane >> 4) + 4 * r;
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                const long col = (long)jb * 64 + (w & 1) * 32 + tb * 16 + (lane & 15);
                if (row < P && col < Q) {
                    const long j = row * Q + col;
                    const double v = acc[ta][tb][r];
                    sres[j] = s_old[ta][r][tb] + v * v;
                    const double mnew = mu[j] + v * znew;
                    mu[j] = mnew;
                    mx = max_nan(mx, mnew);
                }
            }
        }
    double* wm = lds;
    for (int o = 32; o > 0; o >>= 1) mx = max_nan(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) bmax[blockIdx.x] = max_nan(max_nan(wm[0], wm[1]), max_nan(wm[2], wm[3]));
}

// The update with the acquisition in its epilogue: every candidate's value under the incumbent *inc_ptr, the tile's best
// (value, index) and the tile's (max, argmax) of the new means beside the read-modify-write.  The arithmetic per candidate is
// that of k_lattice_gemm_update followed by k_acquire_argmax (variance from the new sum, acq_value), so the values are the
// ones the separate pass would find -- provided the incumbent handed in is the one that pass would use.
struct TileBest {
    double* best_v;     // per tile: largest acquisition value ...
    long* best_i;       // ... and its candidate index (ties: lowest index; NaN first, as better())
    double* mu_v;       // per tile: largest new mean ...
    long* mu_i;         // ... and its candidate index
};
template <int ACQ>
__global__ __launch_bounds__(GEMM_THREADS, 3) void k_lattice_gemm_update_acq(const double* __restrict__ A, long lda, const double* __restrict__ B,
                                                                               long ldb, int kpad, long P, long Q, int tiles_q,
                                                                               const double* __restrict__ scal, const double* __restrict__ params,
                                                                               const double* __restrict__ inc_ptr, double kappa,
                                                                               double* __restrict__ sres, double* __restrict__ mu, TileBest out) {
    const int ib = (int)blockIdx.x / tiles_q, jb = (int)blockIdx.x % tiles_q;
    __shared__ double lds[QUARTER_LDS_DOUBLES];
    f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double s_old[2][4][2];
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long row = (long)ib * 64 + (w >> 1) * 32 + ta * 16 + (lane >> 4) + 4 * r;
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                const long col = (long)jb * 64 + (w & 1) * 32 + tb * 16 + (lane & 15);
                const bool in = row < P && col < Q;
                const long j = in ? row * Q + col : 0;
                s_old[ta][r][tb] = in ? sres[j] : 0.0;
            }
        }
    // ... and the old means with them: three workgroups per CU (170 registers) hold both beside the accumulators, and with the
    // acquisition's arithmetic in the epilogue that beats four per CU loading the means after the product (0.2388 against
    // 0.2415-0.2436 ms per iteration; the update alone prefers four, see above).  Inside the epilogue loop each load would
    // wait behind the stores to the same array.
    double m_old[2][4][2];
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long row = (long)ib * 64 + (w >> 1) * 32 + ta * 16 + (lane >> 4) + 4 * r;
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                const long col = (long)jb * 64 + (w & 1) * 32 + tb * 16 + (lane & 15);
                const bool in = row < P && col < Q;
                m_old[ta][r][tb] = in ? mu[row * Q + col] : 0.0;
            }
        }
    gemm_tile_64x64<false>(A + (long)ib * QUARTER_M * lda, lda, B + (long)jb * 64, ldb, 0, kpad, acc, lds);
    const double znew = scal[A_ZNEW], c0 = params[P_C0], c1 = params[P_C1], inc = *inc_ptr;
    double bv = -INFINITY, mv = -INFINITY;
    long bi = INT64_MAX, mi = INT64_MAX;
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long row = (long)ib * 64 + (w >> 1) * 32 + ta * 16 + (lane >> 4) + 4 * r;
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                const long col = (long)jb * 64 + (w & 1) * 32 + tb * 16 + (lane & 15);
                if (row < P && col < Q) {
                    const long j = row * Q + col;
                    const double v = acc[ta][tb][r];
                    const double s = fma(v, v, s_old[ta][r][tb]);
                    sres[j] = s;
                    const double mnew = fma(v, znew, m_old[ta][r][tb]);
                    mu[j] = mnew;
                    if (better(mnew, j, mv, mi)) {
                        mv = mnew;
                        mi = j;
                    }
                    double vr = fma(-c1, s, c0);
                    if (vr < 1e-60) vr = 0.0;
                    const double a = acq_value(mnew, vr, inc, ACQ, kappa);
                    if (better(a, j, bv, bi)) {
                        bv = a;
                        bi = j;
                    }
                }
            }
        }
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(bv, o), om = __shfl_xor(mv, o);
        const long oi = __shfl_xor(bi, o), omi = __shfl_xor(mi, o);
        if (better(ov, oi, bv, bi)) {
            bv = ov;
            bi = oi;
        }
        if (better(om, omi, mv, mi)) {
            mv = om;
            mi = omi;
        }
    }
    double* wv = lds;                                   // [0..3] best values, [4..7] max means; indices behind them
    long* wi = reinterpret_cast<long*>(lds + 8);
    __syncthreads();                                    // (every wave is done with the operand buffers)
    if (lane == 0) {
        wv[w] = bv;
        wv[4 + w] = mv;
        wi[w] = bi;
        wi[4 + w] = mi;
    }
    __syncthreads();
    if (tid == 0) {
        for (int q = 1; q < 4; ++q) {
            if (better(wv[q], wi[q], bv, bi)) {
                bv = wv[q];
                bi = wi[q];
            }
            if (better(wv[4 + q], wi[4 + q], mv, mi)) {
                mv = wv[4 + q];
                mi = wi[4 + q];
            }
        }
        out.best_v[blockIdx.x] = bv;
        out.best_i[blockIdx.x] = bi;
        out.mu_v[blockIdx.x] = mv;
        out.mu_i[blockIdx.x] = mi;
    }
}

// End of a step whose update launch carried the acquisition: the last stage of both reductions over the per-tile results, the
// decision on the incumbent (hit = the acquisition values are those of the separate pass), the end-of-step block (layout of
// k_step_end, plus res[12] = hit), the index of the largest mean for the next step's incumbent, and the incumbent slots.
__global__ __launch_bounds__(1024) void k_spec_finish(TileBest in, long tiles, const double* __restrict__ params, const int* __restrict__ info,
                                                      double* __restrict__ spec, int need_match, int have, double* __restrict__ inc_out,
                                                      double* __restrict__ res) {
    __shared__ double rv[1024], rm[1024];
    __shared__ long ri[1024], rj[1024];
    const int tid = threadIdx.x;
    double bv = -INFINITY, mv = -INFINITY;
    long bi = INT64_MAX, mi = INT64_MAX;
    for (long t = tid; t < tiles; t += 1024) {
        if (better(in.best_v[t], in.best_i[t], bv, bi)) {
            bv = in.best_v[t];
            bi = in.best_i[t];
        }
        if (better(in.mu_v[t], in.mu_i[t], mv, mi)) {
            mv = in.mu_v[t];
            mi = in.mu_i[t];
        }
    }
    rv[tid] = bv;
    ri[tid] = bi;
    rm[tid] = mv;
    rj[tid] = mi;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) {
            if (better(rv[tid + o], ri[tid + o], rv[tid], ri[tid])) {
                rv[tid] = rv[tid + o];
                ri[tid] = ri[tid + o];
            }
            if (better(rm[tid + o], rj[tid + o], rm[tid], rj[tid])) {
                rm[tid] = rm[tid + o];
                rj[tid] = rj[tid + o];
            }
        }
        __syncthreads();
    }
    if (tid < 8) res[tid] = params[tid];
    if (tid == 8) res[8] = (double)*info;
    if (tid == 9) res[9] = rm[0];
    if (tid == 10) res[10] = rv[0];
    if (tid == 11) reinterpret_cast<long*>(res)[11] = ri[0];
    if (tid == 12) {
        const bool same = __double_as_longlong(rm[0]) == __double_as_longlong(spec[0]);
        res[12] = (!need_match || (have && same)) ? 1.0 : 0.0;
        reinterpret_cast<long*>(spec)[1] = rj[0];
        inc_out[0] = rm[0];
        inc_out[1] = rm[0];
    }
}

// var_j = c0 - c1 s_j (< 1e-60 -> 0) for all candidates: the variance array after lattice-route appends, on demand.
__global__ __launch_bounds__(256) void k_var_from_s(const double* __restrict__ sres, const double* __restrict__ params, double* __restrict__ var,
                                                    long m) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    double vr = params[P_C0] - params[P_C1] * sres[j];
    if (vr < 1e-60) vr = 0.0;
    var[j] = vr;
}

}  // namespace gpbo
