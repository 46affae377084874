// Device kernels of libgpbo (gfx950), part of kernels.hpp: reductions (max, first-occurrence argmax) and the acquisition functions.
#pragma once

namespace gpbo {

// ---------------------------------------------------------------------------------------
// Reductions: max and (value, first index) argmax, two stages, deterministic.
// Both follow NumPy on NaN: a maximum that meets a NaN is NaN (np.max, BayesOpt_v4.ipynb:193), and the argmax returns the
// first NaN if there is one (np.argmax, :647), so a poisoned posterior surfaces instead of being masked.
__device__ __forceinline__ double max_nan(double a, double b) { return (a > b || a != a) ? a : b; }
// out2 (optional, single-workgroup launches): a second destination for the result (the incumbent slot of the acquisition)
__global__ __launch_bounds__(256) void k_max_stage(const double* __restrict__ a, long m, double* __restrict__ out,
                                                   double* __restrict__ out2 = nullptr) {
    __shared__ double r[256];
    const int tid = threadIdx.x;
    double v = -INFINITY;
    for (long i = (long)blockIdx.x * 256 + tid; i < m; i += (long)gridDim.x * 256) v = max_nan(v, a[i]);
    r[tid] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) r[tid] = max_nan(r[tid], r[tid + o]);
        __syncthreads();
    }
    if (tid == 0) {
        out[blockIdx.x] = r[0];
        if (out2) out2[blockIdx.x] = r[0];
    }
}

// The new scaled point of a loop step, passed to the step's first kernel by value.
struct PointArg {
    double x[MAX_D];
};

// End of a loop step: everything the host reads back, gathered into one block (pinned host memory mapped into the
// device's address space: the kernel's stores are the transfer).  res: [0..7] params head, [8] info, [9] max(mu),
// [10] best acquisition value, [11] its index (bit pattern of an int64).
constexpr int STEP_RES_DOUBLES = 16;
__global__ void k_step_end(const double* __restrict__ params, const int* __restrict__ info, const double* __restrict__ mu_max,
                           const double* __restrict__ best_v, const long* __restrict__ best_i, double* __restrict__ res) {
    const int t = threadIdx.x;
    if (t < 8) res[t] = params[t];
    if (t == 8) res[8] = (double)*info;
    if (t == 9) res[9] = *mu_max;
    if (t == 10) res[10] = *best_v;
    if (t == 11) reinterpret_cast<long*>(res)[11] = *best_i;
}

// (v, i) goes ahead of (bv, bi): a NaN ranks above every number, equal ranks go to the lower index.  A total order, so
// the winner does not depend on the shape of the reduction.
__device__ __forceinline__ bool better(double v, long i, double bv, long bi) {
    const bool vn = v != v, bn = bv != bv;
    if (vn || bn) return vn && (!bn || i < bi);
    return (v > bv) || (v == bv && i < bi);
}

// The last stage of the argmax (over the per-workgroup results of k_acquire_argmax) and the end-of-step block in one
// single-workgroup launch: res[10] / res[11] receive the winner, the rest as k_step_end.  Which of several equal values wins
// is decided by the index alone, so the result does not depend on the shape of the reduction.
__global__ __launch_bounds__(1024) void k_argmax_step_end(const double* __restrict__ pv, const long* __restrict__ pi, long m,
                                                          const double* __restrict__ params, const int* __restrict__ info,
                                                          const double* __restrict__ mu_max, double* __restrict__ res) {
    __shared__ double rv[1024];
    __shared__ long ri[1024];
    const int tid = threadIdx.x;
    double bv = -INFINITY;
    long bi = INT64_MAX;
    for (long i0 = tid; i0 < m; i0 += 4096) {
        double v[4];
        long g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = i0 + u * 1024;
            v[u] = (i < m) ? pv[i] : -INFINITY;
            g[u] = (i < m) ? pi[i] : INT64_MAX;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (better(v[u], g[u], bv, bi)) {
                bv = v[u];
                bi = g[u];
            }
    }
    rv[tid] = bv;
    ri[tid] = bi;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o && better(rv[tid + o], ri[tid + o], rv[tid], ri[tid])) {
            rv[tid] = rv[tid + o];
            ri[tid] = ri[tid + o];
        }
        __syncthreads();
    }
    if (tid < 8) res[tid] = params[tid];
    if (tid == 8) res[8] = (double)*info;
    if (tid == 9) res[9] = *mu_max;
    if (tid == 10) res[10] = rv[0];
    if (tid == 11) reinterpret_cast<long*>(res)[11] = ri[0];
}

__global__ __launch_bounds__(256) void k_argmax_stage(const double* __restrict__ a, const long* __restrict__ idx_in, long m,
                                                      double* __restrict__ out_v, long* __restrict__ out_i) {
    __shared__ double rv[256];
    __shared__ long ri[256];
    const int tid = threadIdx.x;
    double bv = -INFINITY;
    long bi = INT64_MAX;
    for (long i = (long)blockIdx.x * 256 + tid; i < m; i += (long)gridDim.x * 256) {
        const double v = a[i];
        const long gi = idx_in ? idx_in[i] : i;
        if (better(v, gi, bv, bi)) {
            bv = v;
            bi = gi;
        }
    }
    rv[tid] = bv;
    ri[tid] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o && better(rv[tid + o], ri[tid + o], rv[tid], ri[tid])) {
            rv[tid] = rv[tid + o];
            ri[tid] = ri[tid + o];
        }
        __syncthreads();
    }
    if (tid == 0) {
        out_v[blockIdx.x] = rv[0];
        out_i[blockIdx.x] = ri[0];
    }
}

// Lowest index i with a[i] >= thr (INT64_MAX if none); stage 1 over the array, stage 2 over the per-workgroup results.
__global__ __launch_bounds__(256) void k_first_ge_stage(const double* __restrict__ a, const long* __restrict__ idx_in, long m,
                                                        double thr, long* __restrict__ out_i) {
    __shared__ long ri[256];
    const int tid = threadIdx.x;
    long bi = INT64_MAX;
    for (long i = (long)blockIdx.x * 256 + tid; i < m; i += (long)gridDim.x * 256) {
        if (idx_in) {
            if (idx_in[i] < bi) bi = idx_in[i];
        } else if (a[i] >= thr && i < bi) {
            bi = i;
        }
    }
    ri[tid] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o && ri[tid + o] < ri[tid]) ri[tid] = ri[tid + o];
        __syncthreads();
    }
    if (tid == 0) out_i[blockIdx.x] = ri[0];
}

// erfc(z) for z >= 0 as  t exp(-z^2) G(2t - 1),  t = 2 / (2 + z):  G(u) = exp(z^2) erfc(z) / t is smooth on [-1, 1]
// (1 / (2 sqrt(pi)) at u = -1, i.e. z = inf; 1 at u = 1, i.e. z = 0) and its Chebyshev series converges like 0.2^k.  G was
// projected onto T_0 .. T_95 in 60-digit arithmetic, truncated after T_27 (within 9e-18, relative, of G over the whole
// range) and re-expanded in powers of u: the 28 coefficients below (scripts/erfcx_chebyshev.py).  They fall off like 0.4^k
// and their absolute values sum to 1.03 where G(1) = 1, so Horner's rule -- one FMA per term, half the arithmetic of
// Clenshaw's recurrence on the Chebyshev form -- loses nothing: 2.0e-16 from G in binary64 against 3.8e-16.  One formula for
// every z: no range branches for a wave to diverge on, and the exponential is the one the normal density needs as well.
// The table sits in constant memory, not in the instruction stream: a wave reads it once into scalar registers and every
// term is one v_fma_f64 with a scalar addend (as literals each coefficient costs two more vector moves per use).
__constant__ double ERFCX_POW[28] = {
    0.5107913526210114891,      0.34358034220690527304,      0.13973604627802689131,      0.018221115753442496732,
    -0.010755330944716363879,   -0.0032395925339865131725,   0.0015453060290576436895,    0.0004186480332964868087,
    -0.00033153674761946044252, -0.000017185737594548939107, 0.000072235350614385433482,  -0.000017074336530735700106,
    -0.000010848057268891095768, 8.0600404910281462479e-6,   -5.6281669351929246327e-7,   -1.8208078384029821516e-6,
    1.0009185862654101138e-6,   -3.0587196456747097464e-9,   -2.8554818238341171664e-7,   1.6491163111184632914e-7,
    -1.6982724012935666188e-9,  -5.7601466971756245011e-8,   2.9041985444209085665e-8,    6.3538646695051392166e-9,
    -9.744756625932257797e-9,   1.22845363501865661e-9,      1.2051665995883636701e-9,    -3.3888695546308688491e-10};
__device__ __forceinline__ double erfcx_over_t(double u) {
    double p = ERFCX_POW[27];
#pragma unroll
    for (int k = 26; k >= 0; --k) p = fma(p, u, ERFCX_POW[k]);
    return p;
}

// One candidate's acquisition value: EI (BayesOpt_v4.ipynb:188-197: imp Phi(Z) + sigma phi(Z), ei[sigma == 0] = 0), UCB
// (GP/BayesOpt_v1.m:248-251) or PI (Gaussian_Process_Regression.ipynb:265-268).  Phi(Z) = erfc(-Z / sqrt 2) / 2 for Z < 0 and
// 1 - erfc(Z / sqrt 2) / 2 otherwise, phi(Z) = exp(-Z^2 / 2) / sqrt(2 pi): one exponential serves both.  Every kernel that
// evaluates an acquisition goes through this function, so their results agree bit for bit.  sigma = 0: Z = +-inf gives
// Phi = 0 or 1 (t = 0, exp(-inf) = 0); 0 / 0 gives NaN under PI, as norm.cdf(nan) does.
__device__ __forceinline__ double acq_value(double mj, double vj, double inc, int acq, double kappa) {
    const double sigma = sqrt(vj);
    if (acq == 1) return mj + kappa * sigma;
    const double imp = (acq == 0) ? mj - inc - kappa : mj - inc;
    const double Z = imp / sigma;
    const double z = fabs(Z) * 0.70710678118654752440;
    const double t = 2.0 / (2.0 + z);
    const double ZZ = Z * Z;
    const double E = exp_neg(-ZZ / 2.0);         // the density's exponential, formed as the reference forms it
    // for the distribution function the rounding of Z * Z (worth Z^2 ulp in the far tail) is put back: exp(-(ZZ + e) / 2) = E (1 - e / 2)
    const double Ec = (ZZ < INFINITY) ? fma(E, -0.5 * fma(Z, Z, -ZZ), E) : E;
    const double half_erfc = 0.5 * t * Ec * erfcx_over_t(2.0 * t - 1.0);
    const double Phi = (Z < 0.0) ? half_erfc : 1.0 - half_erfc;
    if (acq == 2) return Phi;
    double r = imp * Phi + sigma * (E * 0.39894228040143267794);
    if (sigma == 0.0) r = 0.0;
    return r;
}

// Acquisition over the resident posterior.  inc_ptr: device scalar incumbent (EI/PI).
__global__ __launch_bounds__(256) void k_acquire(const double* __restrict__ mu, const double* __restrict__ var, long m, int acq,
                                                 double kappa, const double* __restrict__ inc_ptr,
                                                 double* __restrict__ out) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    out[j] = acq_value(mu[j], var[j], *inc_ptr, acq, kappa);
}

// Acquisition and stage 1 of the first-occurrence argmax in one pass over (mu, var): a workgroup covers ACQ_SPAN
// consecutive candidates, a thread ACQ_PER of them at stride 256 (coalesced), ascending, so "strictly greater" keeps the
// lowest index; the workgroup's (value, index) goes to part_v / part_i.  STORE: also write the values to out.
constexpr int ACQ_PER = 8;
constexpr int ACQ_SPAN = 256 * ACQ_PER;
// n_parts > 0: the incumbent is the maximum of inc_parts[0 .. n_parts) (n_parts <= 256: the second-stage results of the
// max(mu) reduction), which every workgroup reduces for itself instead of waiting for a launch that does it once; workgroup 0
// leaves it in inc_out[0] (incumbent) and inc_out[1] (max(mu)).  A maximum does not depend on the order it is taken in.
// The maximum of up to 256 partial maxima, by every thread of a 256-thread workgroup (rv: 4 doubles of shared memory).
__device__ __forceinline__ double block_max_of_parts(const double* __restrict__ parts, int n_parts, double* rv) {
    const int tid = threadIdx.x;
    double v = (tid < n_parts) ? parts[tid] : -INFINITY;
    for (int o = 32; o > 0; o >>= 1) v = max_nan(v, __shfl_xor(v, o));
    if ((tid & 63) == 0) rv[tid >> 6] = v;
    __syncthreads();
    const double r = max_nan(max_nan(rv[0], rv[1]), max_nan(rv[2], rv[3]));
    __syncthreads();
    return r;
}

template <bool STORE, int ACQ = -1>
__global__ __launch_bounds__(256) void k_acquire_argmax(const double* __restrict__ mu, const double* __restrict__ var, long m, int acq,
                                                        double kappa, const double* __restrict__ inc_ptr, double* __restrict__ out,
                                                        double* __restrict__ part_v, long* __restrict__ part_i,
                                                        const double* __restrict__ inc_parts = nullptr, int n_parts = 0,
                                                        double* __restrict__ inc_out = nullptr,
                                                        const double* __restrict__ c01 = nullptr) {
    // c01 != nullptr: `var` holds sum v^2 per candidate and the variance is c01[0] - c01[1] * var[j] (< 1e-60 -> 0), as the
    // lattice route of an append leaves it
    __shared__ double rv[256];
    __shared__ long ri[256];
    const int tid = threadIdx.x;
    const long base = (long)blockIdx.x * ACQ_SPAN;
    double inc;
    if (n_parts > 0) {
        inc = block_max_of_parts(inc_parts, n_parts, rv);
        if (blockIdx.x == 0 && tid == 0) {
            inc_out[0] = inc;
            inc_out[1] = inc;
        }
    } else {
        inc = *inc_ptr;
    }
    double bv = -INFINITY;
    long bi = INT64_MAX;
    // all of the thread's candidates are requested before the first is evaluated (inside the bounds check of each one the
    // loads would wait in turn: eight memory round trips per thread instead of one)
    double mj[ACQ_PER], vj[ACQ_PER];
#pragma unroll
    for (int u = 0; u < ACQ_PER; ++u) {
        const long j = base + u * 256 + tid;
        mj[u] = (j < m) ? mu[j] : 0.0;
        vj[u] = (j < m) ? var[j] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < ACQ_PER; ++u) {
        const long j = base + u * 256 + tid;
        if (j < m) {
            double v = vj[u];
            if (c01) {
                v = c01[0] - c01[1] * v;
                if (v < 1e-60) v = 0.0;
            }
            const double r = acq_value(mj[u], v, inc, ACQ >= 0 ? ACQ : acq, kappa);       // (ACQ >= 0: the acquisition is fixed at compile time)
            if (STORE) out[j] = r;
            if (better(r, j, bv, bi)) {
                bv = r;
                bi = j;
            }
        }
    }
    rv[tid] = bv;
    ri[tid] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o && better(rv[tid + o], ri[tid + o], rv[tid], ri[tid])) {
            rv[tid] = rv[tid + o];
            ri[tid] = ri[tid + o];
        }
        __syncthreads();
    }
    if (tid == 0) {
        part_v[blockIdx.x] = rv[0];
        part_i[blockIdx.x] = ri[0];
    }
}

}  // namespace gpbo
