// libgpbo host side: context, memory, launch sequences and the extern "C" entry points
// declared in include/gpbo.h.  Single GPU per context, one HIP stream, no CPU compute path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gpbo.h"
#include "kernels.hpp"
#include "lu.hpp"
#include "lattice.hpp"
#include "append_fused.hpp"

#define GPBO_VARIANT_KRIGING 2      /* trend GP of Gaussian_Process_Regression.ipynb:197-234 (internal tag) */
#define GPBO_VARIANT_KRIGING_LU 4   /* the same model solved through LU with partial pivoting: bufW holds P K = L U, ipiv the row exchanges */

using namespace gpbo;

namespace {

enum TimeCat { T_KBUILD, T_CHOL, T_TRSV, T_INV, T_KS, T_TRMM, T_FIN, T_ACQ, T_TOTAL, T_SWEEP, T_NCAT };

struct EvRec {
    int cat;
    hipEvent_t a, b;
};

}  // namespace

struct gpbo_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    int64_t info = 0;

    // model
    bool fitted = false;
    int64_t n = 0, n_pad = 0, nb = 0, cap_pad = 0;
    int cap_batch = 1;
    int d = 0, kernel = 0, variant = 0;
    double sigmaY = 0, eps = 0, theta1 = 0, nll = 0, y_max = 0;
    std::vector<double> theta;
    double *bufA = nullptr, *bufB = nullptr, *bufW = nullptr, *Dinv = nullptr;
    double *Xts = nullptr, *alpha = nullptr, *alpha1 = nullptr, *yv = nullptr, *work = nullptr, *zv = nullptr;
    double* params = nullptr;
    int* info_dev = nullptr;
    bool have_factor = false;   // bufL valid (false on ranks that only received W)
    double* bufL = nullptr;     // points at bufA or bufB

    // incremental append (single-factor variant)
    int64_t reserve = 0;        // rows reserved for appended points (requested before fit)
    int64_t reserve_left = 0;   // rows still free in the fitted model
    double *zfw = nullptr, *kvec = nullptr, *lvec = nullptr, *scal = nullptr, *xnew_dev = nullptr;
    double *Vres = nullptr, *sres = nullptr;
    int64_t v_cap_rows = 0, v_cap_m = 0;
    bool inc_valid = false;     // Vres/sres describe the current model and candidates
    std::vector<double> h_xnew;

    // candidates given as a lattice (gpbo_set_candidates_lattice): with the RBF kernel an append takes the new row of V
    // as a matrix product of per-axis factors (lattice.hpp) and V is not kept
    bool lattice = false;       // the resident candidates are the full lattice of lat (lexicographic order)
    bool inc_lattice = false;   // the current incremental state (sres, latE, latB) is the lattice route's
    bool var_stale = false;     // lattice-route appends since `var` was last written: the variance is theta1 (1 - sres[j])
    LatticeDesc lat{};
    double *lat_axes = nullptr, *latA = nullptr, *latB = nullptr, *latE = nullptr, *lat_bmax = nullptr;
    int64_t lat_axes_cap = 0, lat_kcap = 0, lat_ppad = 0, lat_qpad = 0, lat_smax = 0, lat_a_cap = 0, lat_b_cap = 0, lat_e_cap = 0,
            lat_bmax_cap = 0, sres_cap = 0;
    // acquisition inside the lattice route's update launch (append_acquire): per-tile results, the incumbent slot, and whether
    // the device knows where the largest mean of the current posterior sits (lattice.hpp: SpecArgs)
    double *lat_tile_v = nullptr, *spec_dev = nullptr;
    long *lat_tile_i = nullptr, *lat_mu_i = nullptr;
    int64_t lat_tile_cap = 0;
    bool mu_argmax_valid = false;
    long sweep_parts = 0;       // per-workgroup max(mu) partials the last append left in its first-stage buffer
    double* sweep_part_buf = nullptr;

    // workspace of the one-workgroup likelihood kernel (n <= 128), kept between calls
    double* sm_buf = nullptr;
    size_t sm_cap = 0;          // doubles
    double* sm_host = nullptr;  // pinned staging of the same size (pageable memory would be copied through a driver buffer)

    // batched likelihood sweep, general path: batch arrays (caller context) and per-lane state, all kept between calls
    double *lb_xraw = nullptr, *lb_th = nullptr, *lb_sg = nullptr, *lb_nll = nullptr, *lb_grad = nullptr;
    int* lb_info = nullptr;
    size_t lb_xraw_cap = 0, lb_b_cap = 0, lb_bd_cap = 0;
    double* lane_gpart = nullptr;
    size_t lane_gpart_cap = 0;
    LmlSlot* lane_slot = nullptr;
    hipGraphExec_t lane_graph = nullptr;
    std::vector<int64_t> lane_graph_key;
    size_t params_cap = PARAMS_DOUBLES, slot_cap = 0;
    int64_t alloc_gen = 0;      // bumped whenever a buffer a captured launch sequence names is (re)allocated
    bool no_timing = false;     // set while a launch sequence is captured into a graph
    bool timing_off = false;    // gpbo_set_timing(ctx, 0)
    int batch = 1;              // > 1: a batched context, every model buffer holds `batch` slots (see BatchStride)
    BatchStride bs;
    gpbo_ctx* lml_lock = nullptr;   // the batched work context of the likelihood sweep (caller context only)
    gpbo_ctx* kr_work = nullptr;    // private work context of gpbo_kriging_lml (caller context only)
    int* ipiv = nullptr;            // row exchanges of the LU fallback of the trend flavour (n_pad ints)
    int64_t ipiv_cap = 0;

    // work contexts of the batched likelihood sweep (created on first use, kept for the next call)
    std::vector<gpbo_ctx*> lml_lanes;

    // candidates + outputs
    double* Xs = nullptr;
    int64_t m = 0, m_cap = 0, xs_cap = 0;   // m_cap: outputs (doubles per array); xs_cap: doubles in Xs
    int64_t xs_rows = 0;                    // candidate rows resident in Xs (m may differ after gpbo_set_posterior)
    double *mu = nullptr, *var = nullptr, *acq = nullptr;
    double* res_host = nullptr;                  // pinned, device-mapped result block of a loop step (STEP_RES_DOUBLES)
    double* res_dev = nullptr;                   // the device's address of res_host
    double *acq_pv = nullptr, *bmax = nullptr;   // per-workgroup partials: fused acquisition/argmax, max(mu) of the append sweep
    long* acq_pi = nullptr;
    bool have_posterior = false;
    bool have_acq = false;      // acq holds the acquisition values of the current posterior

    // chunk scratch
    int64_t chunk_req = 0, chunk = 0, scratch_npad = 0;
    double *Bmat = nullptr, *partial = nullptr, *mupart = nullptr;
    double* Zmat = nullptr;     // LU path of the trend flavour only: the solves of a chunk's correlation vectors (same shape as Bmat)
    int64_t z_chunk = 0, z_npad = 0;
    int kc = KS_ROWS;   // training rows per workgroup of the Ks build = rows per partial mean

    // reductions
    double *red_v = nullptr, *red_v2 = nullptr, *inc_dev = nullptr;
    long *red_i = nullptr, *red_i2 = nullptr;

    // timing
    std::vector<EvRec> recs;
    std::vector<hipEvent_t> pool;
    size_t pool_used = 0;
    gpbo_timings tm{};
};

namespace {

#define HIPCHK(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            ctx->err = std::string(#expr) + ": " + hipGetErrorString(e_);                        \
            return GPBO_ERR_HIP;                                                                 \
        }                                                                                        \
    } while (0)

#define BADARG(msg)              \
    do {                         \
        ctx->err = (msg);        \
        return GPBO_ERR_BAD_ARG; \
    } while (0)

#define CHK(expr)                      \
    do {                               \
        int s_ = (expr);               \
        if (s_ != GPBO_OK) return s_;  \
    } while (0)

int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// Largest padded size for which the trend flavour falls back to LU when its matrix is not positive definite (its panel
// factorisation runs in one workgroup per 128 columns: a fallback, not a fast path).
constexpr int64_t KRIGING_LU_MAX = 8192;

hipEvent_t ev_get(gpbo_ctx* ctx) {
    if (ctx->pool_used == ctx->pool.size()) {
        hipEvent_t e;
        hipEventCreate(&e);
        ctx->pool.push_back(e);
    }
    return ctx->pool[ctx->pool_used++];
}

struct Scope {
    gpbo_ctx* ctx;
    EvRec r;
    Scope(gpbo_ctx* c, int cat) : ctx(c) {
        if (c->no_timing || c->timing_off) return;
        r.cat = cat;
        r.a = ev_get(c);
        r.b = ev_get(c);
        hipEventRecord(r.a, c->stream);
    }
    ~Scope() {
        if (ctx->no_timing || ctx->timing_off) return;
        hipEventRecord(r.b, ctx->stream);
        ctx->recs.push_back(r);
    }
};

void timing_begin(gpbo_ctx* ctx) {
    ctx->recs.clear();
    ctx->pool_used = 0;
}

// after a stream sync: accumulate per-category milliseconds
void timing_collect(gpbo_ctx* ctx, double* ms, double* count) {
    for (int i = 0; i < T_NCAT; ++i) ms[i] = 0, count[i] = 0;
    for (auto& r : ctx->recs) {
        float t = 0;
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            ms[r.cat] += t;
            count[r.cat] += 1;
        }
    }
}

template <typename T>
int dev_alloc(gpbo_ctx* ctx, T** p, size_t count) {
    if (*p) {
        hipFree(*p);
        *p = nullptr;
    }
    HIPCHK(hipMalloc((void**)p, std::max<size_t>(count, 1) * sizeof(T)));
    return GPBO_OK;
}

void free_model(gpbo_ctx* ctx) {
    double** ps[] = {&ctx->bufA, &ctx->bufB, &ctx->bufW, &ctx->Dinv, &ctx->Xts, &ctx->alpha, &ctx->alpha1, &ctx->yv, &ctx->work, &ctx->zv,
                     &ctx->zfw, &ctx->kvec, &ctx->lvec};
    for (auto p : ps) {
        if (*p) hipFree(*p);
        *p = nullptr;
    }
    ctx->cap_pad = 0;
    ctx->fitted = false;
    ctx->have_factor = false;
    ctx->alloc_gen += 1;
    if (ctx->lane_graph) {       // its launches name the buffers just freed
        hipGraphExecDestroy(ctx->lane_graph);
        ctx->lane_graph = nullptr;
    }
    ctx->lane_graph_key.clear();
}

// (re)allocate model buffers for n_pad; full = also the factor buffers (fit ranks)
int ensure_model(gpbo_ctx* ctx, int64_t n_pad, int d, bool full) {
    const bool need = (n_pad != ctx->cap_pad) || (full && !ctx->bufA) || d > MAX_D || ctx->batch != ctx->cap_batch;
    if (d > MAX_D) BADARG("d exceeds 32");
    if (!need) return GPBO_OK;
    free_model(ctx);
    const size_t S = (size_t)ctx->batch;
    const size_t nn = (size_t)n_pad * n_pad * S, nv = (size_t)n_pad * S;
    CHK(dev_alloc(ctx, &ctx->bufW, nn));
    if (full) {
        CHK(dev_alloc(ctx, &ctx->bufA, nn));
        CHK(dev_alloc(ctx, &ctx->bufB, nn));
        CHK(dev_alloc(ctx, &ctx->Dinv, (size_t)n_pad * PB * S));
        CHK(dev_alloc(ctx, &ctx->alpha1, nv));
        CHK(dev_alloc(ctx, &ctx->yv, nv));
        CHK(dev_alloc(ctx, &ctx->work, nv));
        CHK(dev_alloc(ctx, &ctx->zv, nv));
        CHK(dev_alloc(ctx, &ctx->zfw, nv));
        CHK(dev_alloc(ctx, &ctx->kvec, nv));
        CHK(dev_alloc(ctx, &ctx->lvec, nv));
    }
    CHK(dev_alloc(ctx, &ctx->Xts, (size_t)MAX_D * n_pad * S));
    CHK(dev_alloc(ctx, &ctx->alpha, nv));
    ctx->cap_pad = n_pad;
    ctx->cap_batch = ctx->batch;
    return GPBO_OK;
}

int ensure_scratch(gpbo_ctx* ctx) {
    // chunk: multiple of 256 candidates; default keeps Bmat <= ~4 GiB, between 4096 and 131072 columns (longer launches of
    // the variance product lose less to their ramp and tail: +0.8 % at n = 4096 over 32768 columns, +0.5 % at n = 32768
    // over 4096 columns)
    int64_t chunk = ctx->chunk_req;
    if (chunk <= 0) {
        const char* e = getenv("GPBO_CHUNK");
        if (e) chunk = atoll(e);
    }
    if (chunk <= 0) {
        chunk = (int64_t)(4ull << 30) / (8 * ctx->n_pad);
        chunk = std::max<int64_t>(chunk, 4096);
        chunk = std::min<int64_t>(chunk, 131072);
    }
    chunk = round_up(chunk, 256);
    chunk = std::min<int64_t>(chunk, round_up(std::max<int64_t>(ctx->m, 1), 256));
    if (chunk == ctx->chunk && ctx->scratch_npad == ctx->n_pad && ctx->Bmat) return GPBO_OK;
    const int nkc = (int)((ctx->n_pad + ctx->kc - 1) / ctx->kc);
    CHK(dev_alloc(ctx, &ctx->Bmat, (size_t)ctx->n_pad * chunk));
    CHK(dev_alloc(ctx, &ctx->partial, (size_t)ctx->nb * chunk));
    CHK(dev_alloc(ctx, &ctx->mupart, (size_t)nkc * chunk));
    ctx->chunk = chunk;
    ctx->scratch_npad = ctx->n_pad;
    return GPBO_OK;
}

// ---------------------------------------------------------------- launch sequences
int launch_kernel_matrix(gpbo_ctx* ctx, double* K, const double* scale_ptr, double eps, const double* eps_ptr = nullptr) {
    Scope s(ctx, T_KBUILD);
    dim3 grid((unsigned)(ctx->n_pad / 64), (unsigned)(ctx->n_pad / 32), (unsigned)ctx->batch);
    hipLaunchKernelGGL(k_kernel_matrix, grid, dim3(256), 0, ctx->stream, ctx->Xts, (int)ctx->n, (int)ctx->n_pad, ctx->d, K,
                       scale_ptr, eps, eps_ptr, ctx->kernel, ctx->bs);
    HIPCHK(hipGetLastError());
    return GPBO_OK;
}

// Right-looking blocked Cholesky, panels of 128 columns processed in groups of CHOL_GROUP:
// inside a group each panel is factored (diagonal block in one workgroup, then the panel solve) and the group's own
// block columns are brought up to date recursively (chol_factor_group); after the group the whole trailing matrix is
// updated once with K = 128*group.
int chol_group(int nb, int batch) {
    if (const char* e = getenv("GPBO_CHOL_GROUP")) return std::max(1, atoi(e));
    if (batch >= 8 && nb >= 64) return 16;      // slots in lockstep fill the chip between them: the chain matters less than K
                                                // (n = 16384, 16 slots: 14.30 / 14.45 / 14.50 settings/s with 4 / 8 / 16)
    if (nb <= 56) return 1;        // small n: every launch is latency bound, one update per panel is the shortest chain
    if (nb <= 80) return 2;        // measured: n = 8192: 7.70 / 7.65 / 7.92 ms, n = 12288: 19.1 / 17.7 / 17.5 ms with 1 / 2 / 4
    if (nb < 112) return 4;
    return nb >= 192 ? 16 : 8;     // large n, bulk update in bands: n = 16384: 32.29 / 32.10 / 32.13 ms, 20480: 56.9 / 56.0 / 55.9,
                                   // 32768: - / 195.3 / 194.5 (and 195.5 with 32) for 4 / 8 / 16
}

// Panel solve: one wave per 16 rows (k_trsm_rows) while every wave gets a SIMD of its own (8 waves per block row, 1024
// SIMDs); beyond that the 128-tile engine, which shares the inverted block through LDS, is faster (measured at
// n = 32768).  The row form needs only the inverses of the 16x16 diagonal tiles, so for those panels the diagonal-block
// kernel stops after the factor ("deferred") and k_trinv_diag forms the block inverses after the factorisation, all in
// one launch, off the critical path.  Panels from chol_first_deferred(nb, slots) on are deferred.
int chol_first_deferred(int nb, int batch) {
    static const int trsm_rows_max = getenv("GPBO_TRSM_ROWS_MAX") ? atoi(getenv("GPBO_TRSM_ROWS_MAX")) : 128;
    if (nb == 1) return nb;                        // a single block has no panel solve: inverting in place saves a launch
    return std::max(0, nb - 1 - trsm_rows_max / std::max(1, batch));      // block rows below the panel, over all slots
}

// The one-workgroup likelihood and descent routines use 1024 threads above this many training points, 256 up to it.
// The 256-thread form holds the covariance matrix as 2 x 2 register tiles, i.e. at most 32 points: larger settings are
// clamped to 32.
int small_wide_min() {
    static const int v = std::max(0, std::min(32, getenv("GPBO_SMALL_WIDE_MIN") ? atoi(getenv("GPBO_SMALL_WIDE_MIN")) : 32));
    return v;
}

// Trailing updates with at most this many 128 x 128 tiles (over all slots) run on half tiles (k_syrk_update_half).
long syrk_half_max() {
    static const long v = getenv("GPBO_SYRK_HALF_MAX") ? atol(getenv("GPBO_SYRK_HALF_MAX")) : 1024;
    return v;
}

// ... and with at most this many on quarter tiles (k_syrk_update_quarter).  0 = never.
long syrk_quarter_max() {
    static const long v = getenv("GPBO_SYRK_QUARTER_MAX") ? atol(getenv("GPBO_SYRK_QUARTER_MAX")) : 4096;
    return v;
}

// F(group): the latency-bound part.  Needs the group's block columns up to date.
// Inside the group the panels are applied to the group's own block columns recursively: factor the left half, update
// the right half's columns with the whole left half at once (K = 128 * half), factor the right half.  The number of
// launches on the critical path is that of one K = 128 update per panel, but a tile of the group's columns is read and
// written log2(g) times instead of up to g - 1 times, and most of the flops run at K = 256 and up.
void chol_factor_group(hipStream_t st, gpbo_ctx* ctx, double* A, int* info_dev, int p0, int g) {
    const long ld = ctx->n_pad;
    const int nb = (int)ctx->nb;
    const unsigned S = (unsigned)ctx->batch;
    if (g > 1) {
        const int h = g / 2, c0 = p0 + h;          // columns c0 .. p0 + g - 1 take panels p0 .. c0 - 1
        chol_factor_group(st, ctx, A, info_dev, p0, h);
        static const long rect_quarter_max = getenv("GPBO_SYRK_RECT_QUARTER_MAX") ? atol(getenv("GPBO_SYRK_RECT_QUARTER_MAX")) : 4096;
        if ((long)(g - h) * (nb - c0) * S <= rect_quarter_max)
            hipLaunchKernelGGL(k_syrk_update_quarter<false>, dim3(4 * (g - h), nb - c0, S), dim3(GEMM_THREADS), 0, st, A, ld, c0, c0,
                               p0 * PB, h * PB, ctx->bs);
        else if ((long)(g - h) * (nb - c0) * S <= syrk_half_max())
            hipLaunchKernelGGL(k_syrk_update_half<false>, dim3(2 * (g - h), nb - c0, S), dim3(GEMM_THREADS), 0, st, A, ld, c0, c0,
                               p0 * PB, h * PB, ctx->bs);
        else
            hipLaunchKernelGGL(k_syrk_update<false>, dim3(g - h, nb - c0, S), dim3(GEMM_THREADS), 0, st, A, ld, c0, c0, p0 * PB, h * PB,
                               ctx->bs);
        chol_factor_group(st, ctx, A, info_dev, c0, g - h);
        return;
    }
    const int q = p0;
    const int defer = q >= chol_first_deferred(nb, ctx->batch);
    hipLaunchKernelGGL(k_potrf_diag, dim3(1, 1, S), dim3(POTRF_MFMA_THREADS), 0, st, A, ld, q, ctx->Dinv, info_dev, ctx->bs, defer, (int)ctx->n);
    if (q + 1 < nb) {
        if (defer)
            hipLaunchKernelGGL(k_trsm_rows, dim3(8 * (nb - q - 1), 1, S), dim3(64), 0, st, A, ld, q, (const double*)ctx->Dinv, ctx->bs);
        else
            hipLaunchKernelGGL(k_trsm_panel, dim3(nb - q - 1, 1, S), dim3(GEMM_THREADS), 0, st, A, ld, q, (const double*)ctx->Dinv, ctx->bs);
    }
}

int launch_cholesky(gpbo_ctx* ctx, double* A, int* info_dev) {
    Scope s(ctx, T_CHOL);
    const long ld = ctx->n_pad;
    const int nb = (int)ctx->nb;
    const int G = chol_group(nb, ctx->batch);
    {
        for (int p0 = 0; p0 < nb; p0 += G) {
            const int g = std::min(G, nb - p0);
            chol_factor_group(ctx->stream, ctx, A, info_dev, p0, g);
            const int T = nb - (p0 + g);
            const long tiles = (long)T * (T + 1) / 2;
            if (T > 0 && tiles * ctx->batch <= syrk_quarter_max())
                hipLaunchKernelGGL(k_syrk_update_quarter<true>, dim3((unsigned)(4 * tiles), 1, (unsigned)ctx->batch), dim3(GEMM_THREADS), 0,
                                   ctx->stream, A, ld, p0 + g, p0 + g, p0 * PB, g * PB, ctx->bs);
            else if (T > 0 && tiles * ctx->batch <= syrk_half_max())
                hipLaunchKernelGGL(k_syrk_update_half<true>, dim3((unsigned)(2 * tiles), 1, (unsigned)ctx->batch), dim3(GEMM_THREADS), 0,
                                   ctx->stream, A, ld, p0 + g, p0 + g, p0 * PB, g * PB, ctx->bs);
            else if (T > 0) {
                // tiles in bands of eight block rows (band_decode): at n = 32768 the group's panel (T x 128 rows x K columns,
                // up to 500 MB) no longer stays in the 256 MB memory-side cache between the rows of tiles of the row-major
                // order, which re-reads all of it once per row; a band of eight rows re-reads it once per band, and the
                // eight tiles of a column meet in one XCD's L2.  GPBO_SYRK_BAND=0 selects the row-major order.
                static const int band = getenv("GPBO_SYRK_BAND") ? atoi(getenv("GPBO_SYRK_BAND")) : 8;
                if (band == 4)
                    hipLaunchKernelGGL((k_syrk_update<true, 1, 4>), dim3((unsigned)tiles, 1, (unsigned)ctx->batch), dim3(GEMM_THREADS), 0,
                                       ctx->stream, A, ld, p0 + g, T, p0 * PB, g * PB, ctx->bs);
                else if (band == 8)
                    hipLaunchKernelGGL((k_syrk_update<true, 1, 8>), dim3((unsigned)tiles, 1, (unsigned)ctx->batch), dim3(GEMM_THREADS), 0,
                                       ctx->stream, A, ld, p0 + g, T, p0 * PB, g * PB, ctx->bs);
                else
                hipLaunchKernelGGL(k_syrk_update<true>, dim3((unsigned)tiles, 1, (unsigned)ctx->batch), dim3(GEMM_THREADS), 0,
                                   ctx->stream, A, ld, p0 + g, p0 + g, p0 * PB, g * PB, ctx->bs);
            }
        }
    }
    const int first_deferred = chol_first_deferred(nb, ctx->batch);
    if (first_deferred < nb)
        hipLaunchKernelGGL(k_trinv_diag, dim3((unsigned)(nb - first_deferred), 1, (unsigned)ctx->batch), dim3(1024), 0, ctx->stream,
                           (const double*)A, ld, first_deferred, ctx->Dinv, ctx->bs, (int)ctx->n);
    dim3 grid((unsigned)(ctx->n_pad / 64), (unsigned)(ctx->n_pad / 4), (unsigned)ctx->batch);
    hipLaunchKernelGGL(k_zero_upper, grid, dim3(256), 0, ctx->stream, A, (int)ctx->n_pad, ctx->bs);
    HIPCHK(hipGetLastError());
    return GPBO_OK;
}

// dst = L^-T L^-1 src  (uses Dinv of the factor that was computed last).  keep_z: also keep z = L^-1 src in zfw.
// gate: device flag per slot (stride bs.par); the solve is skipped where it is 0 (nullptr: always solve).
int launch_trsv_vec(gpbo_ctx* ctx, const double* L, const double* src, double* dst, bool keep_z, const double* gate) {
    const long ld = ctx->n_pad;
    const int nb = (int)ctx->nb;
    const unsigned S = (unsigned)ctx->batch;
    const size_t vbytes = (size_t)ctx->n_pad * S * sizeof(double);
    HIPCHK(hipMemcpyAsync(ctx->work, src, vbytes, hipMemcpyDeviceToDevice, ctx->stream));
    for (int p = 0; p < nb; ++p)
        hipLaunchKernelGGL(k_trsv_fwd_step, dim3(nb - p, 1, S), dim3(TRSV_FWD_THREADS), 0, ctx->stream, L, ld, (const double*)ctx->Dinv, ctx->work,
                           ctx->zv, p, ctx->bs, gate);
    if (keep_z) HIPCHK(hipMemcpyAsync(ctx->zfw, ctx->zv, vbytes, hipMemcpyDeviceToDevice, ctx->stream));   // z = L^-1 y
    for (int p = nb - 1; p >= 0; --p)
        hipLaunchKernelGGL(k_trsv_bwd_step, dim3(p + 1, 1, S), dim3(256), 0, ctx->stream, L, ld, (const double*)ctx->Dinv, ctx->zv,
                           dst, p, ctx->bs, gate);
    HIPCHK(hipGetLastError());
    return GPBO_OK;
}

// alpha1 = (K0 + eps I)^-1 y from the factor L, plus one gated step of iterative refinement against the covariance
// matrix itself (see k_refine_residual).  eps: eps_ptr per slot if given, else eps_val.
int launch_trsv(gpbo_ctx* ctx, const double* L, double eps_val, const double* eps_ptr = nullptr) {
    Scope s(ctx, T_TRSV);
    const unsigned S = (unsigned)ctx->batch;
    CHK(launch_trsv_vec(ctx, L, ctx->yv, ctx->alpha1, true, nullptr));
    // Two steps.  Measured at cond 4e11 (golden case g3_d4_n512_s0): theta1 is 1.05e-6 from the reference's without
    // refinement and 1-2.5e-7 with one to five steps -- the level at which the reference's own value is defined: the same
    // LAPACK algorithm with another block size moves theta1 by 2-6e-7, one-ulp changes of K's entries by 4-5e-7
    // (scripts/illcond_noise_floor.py).  GPBO_REFINE=k: k steps, 0: none.
    int steps = 2;
    if (const char* e = getenv("GPBO_REFINE")) steps = std::max(0, atoi(e));
    if (steps == 0) return GPBO_OK;
    hipLaunchKernelGGL(k_refine_flag, dim3(1, 1, S), dim3(256), 0, ctx->stream, L, (long)ctx->n_pad, (int)ctx->n, ctx->params, ctx->bs);
    if (!ctx->no_timing) {
        // not under graph capture: read the flags back (one small copy) and skip the gated launches when no slot needs them
        // (one contiguous copy spanning the slots' parameter blocks: a strided hipMemcpy2DAsync to pageable memory left
        // device memory allocated after the context was destroyed, one matrix's worth per context lifetime)
        const size_t stride = S > 1 ? (size_t)ctx->bs.par : 1;
        std::vector<double> flags((S - 1) * stride + 1, 0.0);
        HIPCHK(hipMemcpyAsync(flags.data(), ctx->params + P_REFINE, flags.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        bool any = false;
        for (unsigned sl = 0; sl < S; ++sl) any = any || flags[sl * stride] != 0.0;
        if (!any) return GPBO_OK;
    }
    for (int it = 0; it < steps; ++it) {
        hipLaunchKernelGGL(k_refine_residual, dim3((unsigned)(ctx->n_pad / REF_ROWS), 1, S), dim3(256), 0, ctx->stream,
                           (const double*)ctx->Xts, (int)ctx->n, (int)ctx->n_pad, ctx->d, (const double*)ctx->yv, (const double*)ctx->alpha1,
                           eps_val, eps_ptr, ctx->kernel, (const double*)ctx->params, ctx->kvec, ctx->bs);
        CHK(launch_trsv_vec(ctx, L, ctx->kvec, ctx->lvec, false, ctx->params + P_REFINE));
        hipLaunchKernelGGL(k_refine_add, dim3((unsigned)((ctx->n_pad + 255) / 256), 1, S), dim3(256), 0, ctx->stream, ctx->alpha1,
                           (const double*)ctx->lvec, (int)ctx->n_pad, (const double*)ctx->params, ctx->bs);
    }
    HIPCHK(hipGetLastError());
    return GPBO_OK;
}

// W = inv(L): diagonal blocks from Dinv, then log2(nb) doubling levels
int launch_inverse(gpbo_ctx* ctx, const double* L, double* W, double* T) {
    Scope s(ctx, T_INV);
    const long ld = ctx->n_pad;
    const int nb = (int)ctx->nb;
    const unsigned S = (unsigned)ctx->batch;
    HIPCHK(hipMemsetAsync(W, 0, (size_t)ctx->n_pad * ctx->n_pad * S * sizeof(double), ctx->stream));
    hipLaunchKernelGGL(k_inv_copy_diag, dim3(nb, 1, S), dim3(256), 0, ctx->stream, W, ld, (const double*)ctx->Dinv, ctx->bs);
    static const long inv_half_max = getenv("GPBO_INV_HALF_MAX") ? atol(getenv("GPBO_INV_HALF_MAX")) : 1024;
    for (int sb = 1; sb < nb; sb *= 2) {
        const int pairs = (nb + 2 * sb - 1) / (2 * sb);
        dim3 grid(sb, sb, pairs * S);  // step1: (ib, jb), step2: (jb, reversed ib): the slow index orders tiles longest first
        static const long inv_quarter_max = getenv("GPBO_INV_QUARTER_MAX") ? atol(getenv("GPBO_INV_QUARTER_MAX")) : 1024;
        if ((long)sb * sb * pairs * S <= inv_quarter_max) {
            dim3 gq(4 * sb, sb, pairs * S);
            hipLaunchKernelGGL(k_inv_step1_quarter, gq, dim3(GEMM_THREADS), 0, ctx->stream, L, (const double*)W, T, ld, sb, nb, ctx->bs);
            hipLaunchKernelGGL(k_inv_step2_quarter, gq, dim3(GEMM_THREADS), 0, ctx->stream, W, (const double*)T, ld, sb, nb, ctx->bs);
            continue;
        }
        if ((long)sb * sb * pairs * S <= inv_half_max) {      // too few tiles for the chip: half tiles
            dim3 gh(2 * sb, sb, pairs * S);
            hipLaunchKernelGGL(k_inv_step1<true>, gh, dim3(GEMM_THREADS), 0, ctx->stream, L, (const double*)W, T, ld, sb, nb, ctx->bs);
            hipLaunchKernelGGL(k_inv_step2<true>, gh, dim3(GEMM_THREADS), 0, ctx->stream, W, (const double*)T, ld, sb, nb, ctx->bs);
            continue;
        }
        hipLaunchKernelGGL(k_inv_step1<false>, grid, dim3(GEMM_THREADS), 0, ctx->stream, L, (const double*)W, T, ld, sb, nb, ctx->bs);
        hipLaunchKernelGGL(k_inv_step2<false>, grid, dim3(GEMM_THREADS), 0, ctx->stream, W, (const double*)T, ld, sb, nb, ctx->bs);
    }
    HIPCHK(hipGetLastError());
    return GPBO_OK;
}

template <int D>
void launch_ks_t(gpbo_ctx* ctx, int64_t j0, dim3 grid) {
    hipLaunchKernelGGL(k_ks_build<D>, grid, dim3(256), 0, ctx->stream, ctx->Xs, (long)ctx->m, (long)j0, ctx->d, ctx->Xts,
                       (int)ctx->n, (int)ctx->n_pad, ctx->alpha, ctx->params, ctx->Bmat, (long)ctx->chunk, ctx->mupart, ctx->kc,
                       ctx->kernel);
}

void launch_ks(gpbo_ctx* ctx, int64_t j0, dim3 grid) {
    switch (ctx->d) {
        case 1: launch_ks_t<1>(ctx, j0, grid); break;
        case 2: launch_ks_t<2>(ctx, j0, grid); break;
        case 3: launch_ks_t<3>(ctx, j0, grid); break;
        case 4: launch_ks_t<4>(ctx, j0, grid); break;
        case 5: launch_ks_t<5>(ctx, j0, grid); break;
        case 6: launch_ks_t<6>(ctx, j0, grid); break;
        case 7: launch_ks_t<7>(ctx, j0, grid); break;
        case 8: launch_ks_t<8>(ctx, j0, grid); break;
        default: launch_ks_t<0>(ctx, j0, grid); break;
    }
}

int reduce_max(gpbo_ctx* ctx, const double* a, int64_t m, double* dst_dev) {
    const int blocks = (int)std::min<int64_t>(1024, (m + 255) / 256);
    hipLaunchKernelGGL(k_max_stage, dim3(blocks), dim3(256), 0, ctx->stream, a, (long)m, ctx->red_v);
    hipLaunchKernelGGL(k_max_stage, dim3(1), dim3(256), 0, ctx->stream, (const double*)ctx->red_v, (long)blocks, dst_dev);
    HIPCHK(hipGetLastError());
    return GPBO_OK;
}

// Z <- K^-1 R for a k-major block R[k][j] (row pitch ldv, `cols` columns, a multiple of 128) with the factors P K = L U
// held in LU (n_pad x n_pad) and ipiv: Z = R, row exchanges, forward substitution with the unit lower factor and backward
// substitution with U, block row by block row.
int launch_lu_solve(gpbo_ctx* ctx, const double* LU, const int* ipiv, const double* R, double* Z, long ldv, int64_t cols) {
    const long ld = ctx->n_pad;
    const int nb = (int)ctx->nb, n_pad = (int)ctx->n_pad;
    HIPCHK(hipMemcpy2DAsync(Z, (size_t)ldv * 8, R, (size_t)ldv * 8, (size_t)cols * 8, (size_t)n_pad, hipMemcpyDeviceToDevice, ctx->stream));
    const unsigned cblocks = (unsigned)((cols + 255) / 256), ctiles = (unsigned)(cols / TN);
    hipLaunchKernelGGL(k_lu_permute_rows, dim3(cblocks), dim3(256), 0, ctx->stream, Z, ldv, (int)cols, n_pad, ipiv);
    for (int ib = 0; ib < nb; ++ib) {
        if (ib > 0) hipLaunchKernelGGL(k_lu_rect_update, dim3(ctiles), dim3(GEMM_THREADS), 0, ctx->stream, LU, ld, Z, ldv, ib, 0, ib * PB);
        hipLaunchKernelGGL(k_lu_diag_solve<false>, dim3(cblocks), dim3(256), 0, ctx->stream, LU, ld, ib, Z, ldv, (int)cols);
    }
    for (int ib = nb - 1; ib >= 0; --ib) {
        if (ib < nb - 1)
            hipLaunchKernelGGL(k_lu_rect_update, dim3(ctiles), dim3(GEMM_THREADS), 0, ctx->stream, LU, ld, Z, ldv, ib, (ib + 1) * PB, n_pad);
        hipLaunchKernelGGL(k_lu_diag_solve<true>, dim3(cblocks), dim3(256), 0, ctx->stream, LU, ld, ib, Z, ldv, (int)cols);
    }
    HIPCHK(hipGetLastError());
    return GPBO_OK;
}

// GPBO_LATTICE=0 keeps the streaming route (resident V) on lattice candidates too: same-machine comparison of the two.
bool lattice_route_enabled() {
    static const bool on = [] {
        const char* e = getenv("GPBO_LATTICE");
        return !(e && atoi(e) == 0);
    }();
    return on;
}

// Buffers of the lattice route for the fitted model's capacity: A (P_pad x kcap, row-major), B (kcap x Q_pad, k-major), the
// factor table of the prefix axes, the per-tile max(mu) partials.  Rows and columns beyond what has been written stay zero.
int lattice_ensure(gpbo_ctx* ctx) {
    const int64_t kcap = round_up(ctx->n + ctx->reserve_left + 1, TK);
    const int64_t ppad = round_up(ctx->lat.P, QUARTER_M), qpad = round_up(ctx->lat.Q, QUARTER_M);
    int64_t smax = 1;
    for (int t = 0; t < ctx->lat.d; ++t) smax = std::max<int64_t>(smax, ctx->lat.size[t]);
    const int64_t a_need = ppad * kcap, b_need = kcap * qpad, e_need = (int64_t)std::max(ctx->lat.split, 1) * smax * kcap;
    const int64_t tiles = (ppad / QUARTER_M) * (qpad / QUARTER_M);
    if (a_need > ctx->lat_a_cap) {
        CHK(dev_alloc(ctx, &ctx->latA, (size_t)a_need));
        ctx->lat_a_cap = a_need;
    }
    if (b_need > ctx->lat_b_cap) {
        CHK(dev_alloc(ctx, &ctx->latB, (size_t)b_need));
        ctx->lat_b_cap = b_need;
    }
    if (e_need > ctx->lat_e_cap) {
        CHK(dev_alloc(ctx, &ctx->latE, (size_t)e_need));
        ctx->lat_e_cap = e_need;
    }
    if (tiles > ctx->lat_bmax_cap) {
        CHK(dev_alloc(ctx, &ctx->lat_bmax, (size_t)tiles));
        ctx->lat_bmax_cap = tiles;
    }
    if (tiles > ctx->lat_tile_cap) {
        CHK(dev_alloc(ctx, &ctx->lat_tile_v, (size_t)tiles));
        CHK(dev_alloc(ctx, &ctx->lat_tile_i, (size_t)tiles));
        CHK(dev_alloc(ctx, &ctx->lat_mu_i, (size_t)tiles));
        ctx->lat_tile_cap = tiles;
    }
    if (!ctx->spec_dev) CHK(dev_alloc(ctx, &ctx->spec_dev, 8));
    HIPCHK(hipMemsetAsync(ctx->latA, 0, (size_t)a_need * sizeof(double), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->latB, 0, (size_t)b_need * sizeof(double), ctx->stream));
    ctx->lat_kcap = kcap;
    ctx->lat_ppad = ppad;
    ctx->lat_qpad = qpad;
    ctx->lat_smax = smax;
    return GPBO_OK;
}

int posterior_impl(gpbo_ctx* ctx, double* mu_max_out) {
    if (!ctx->fitted) BADARG("posterior: no fitted model");
    if (ctx->xs_rows <= 0 || !ctx->Xs) BADARG("posterior: no candidates");
    ctx->m = ctx->xs_rows;       // a posterior uploaded with gpbo_set_posterior (any length) is replaced by the one of the resident candidates
    CHK(ensure_scratch(ctx));
    if (ctx->variant == GPBO_VARIANT_KRIGING_LU && (!ctx->Zmat || ctx->z_chunk != ctx->chunk || ctx->z_npad != ctx->n_pad)) {
        CHK(dev_alloc(ctx, &ctx->Zmat, (size_t)ctx->n_pad * ctx->chunk));
        ctx->z_chunk = ctx->chunk;
        ctx->z_npad = ctx->n_pad;
    }
    const bool keep_inc = (ctx->variant == GPBO_VARIANT_SINGLE && ctx->reserve > 0 && ctx->have_factor);   // also after the last append
    // lattice candidates + RBF: appends take the new row of V from a product of per-axis factors, V itself is not kept
    // (not where the fast axes have a single combination -- one axis, or trailing axes of one point: A would be as large as V.
    //  The test must not look at axis 0: slabs of one lattice on several contexts have to take the same route.)
    const bool lat = keep_inc && ctx->lattice && ctx->kernel == GPBO_KERNEL_RBF && ctx->lat.Q >= 2 && lattice_route_enabled();
    const bool keep_v = keep_inc && !lat;
    if (keep_v) {
        const int64_t rows = ctx->n + ctx->reserve_left;
        if (rows > ctx->v_cap_rows || ctx->m > ctx->v_cap_m || !ctx->Vres) {
            CHK(dev_alloc(ctx, &ctx->Vres, (size_t)rows * ctx->m));
            ctx->v_cap_rows = rows;
            ctx->v_cap_m = ctx->m;
        }
    }
    if (keep_inc && (ctx->m > ctx->sres_cap || !ctx->sres)) {
        CHK(dev_alloc(ctx, &ctx->sres, (size_t)ctx->m));
        ctx->sres_cap = ctx->m;
    }
    if (lat) CHK(lattice_ensure(ctx));
    ctx->inc_valid = false;
    timing_begin(ctx);
    const int nkc = (int)((ctx->n_pad + ctx->kc - 1) / ctx->kc);
    const int nb = (int)ctx->nb;
    double trmm_launches = 0;
    double ms[T_NCAT], cnt[T_NCAT], ms_acc[T_NCAT];
    for (int i = 0; i < T_NCAT; ++i) ms_acc[i] = 0;
    // Long sweeps are cut into segments of SEG chunks: at a segment boundary the stream is drained, the event pool
    // recycled (six events per chunk otherwise pile up: 16384 chunks at n = 32768 and 64 M candidates) and, with
    // GPBO_PROGRESS=1, one line goes to stderr.  A boundary costs one idle gap of a few microseconds.
    const int64_t SEG = 512;
    const bool progress = getenv("GPBO_PROGRESS") && atoi(getenv("GPBO_PROGRESS")) != 0;
    const int64_t nchunks = (ctx->m + ctx->chunk - 1) / ctx->chunk;
    const auto wall0 = std::chrono::steady_clock::now();
    {
        std::unique_ptr<Scope> tot(new Scope(ctx, T_TOTAL));      // start-to-end time of a segment, launch gaps included
        int64_t ci = 0;
        for (int64_t j0 = 0; j0 < ctx->m; j0 += ctx->chunk, ++ci) {
            if (ci > 0 && ci % SEG == 0) {
                tot.reset();
                HIPCHK(hipStreamSynchronize(ctx->stream));
                timing_collect(ctx, ms, cnt);
                for (int i = 0; i < T_NCAT; ++i) ms_acc[i] += ms[i];
                timing_begin(ctx);
                if (progress)
                    fprintf(stderr, "[gpbo] posterior: chunk %lld of %lld, %.1f s\n", (long long)ci, (long long)nchunks,
                            std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count());
                tot.reset(new Scope(ctx, T_TOTAL));
            }
            const int64_t cols = std::min<int64_t>(ctx->chunk, round_up(ctx->m - j0, 256));   // multiple of 256
            {
                Scope s(ctx, T_KS);
                launch_ks(ctx, j0, dim3((unsigned)(cols / 256), (unsigned)nkc));
            }
            {
                Scope s(ctx, T_TRMM);
                const int nj = (int)(cols / TN);
                int cg = std::min(nj, 128);   // column blocks per tile-order group
                if (const char* e = getenv("GPBO_TRMM_COLGROUP")) cg = std::max(1, std::min(nj, atoi(e)));
                // tile engine mode 2: priority raised over each MFMA burst, LDS fragments requested one k-substep ahead
                if (ctx->variant == GPBO_VARIANT_KRIGING_LU) {    // bufW holds P K = L U: r^T K^-1 r per candidate through the factors
                    CHK(launch_lu_solve(ctx, ctx->bufW, ctx->ipiv, ctx->Bmat, ctx->Zmat, (long)ctx->chunk, cols));
                    hipLaunchKernelGGL(k_col_dot, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)ctx->Bmat,
                                       (const double*)ctx->Zmat, (long)ctx->chunk, (int)ctx->n_pad, (int)cols, ctx->partial, nb);
                }
                else if (keep_v)
                    hipLaunchKernelGGL((k_trmm_sumsq<true, 2>), dim3((unsigned)(nb * nj)), dim3(GEMM_THREADS), 0, ctx->stream,
                                       (const double*)ctx->bufW, (long)ctx->n_pad, (const double*)ctx->Bmat, (long)ctx->chunk,
                                       ctx->partial, nb, nj, ctx->Vres, (long)ctx->m, (long)j0, (long)ctx->m, (int)ctx->n, cg);
                else
                    hipLaunchKernelGGL((k_trmm_sumsq<false, 2>), dim3((unsigned)(nb * nj)), dim3(GEMM_THREADS), 0, ctx->stream,
                                       (const double*)ctx->bufW, (long)ctx->n_pad, (const double*)ctx->Bmat, (long)ctx->chunk,
                                       ctx->partial, nb, nj, (double*)nullptr, 0L, (long)j0, (long)ctx->m, 0, cg);
                trmm_launches += 1;
            }
            {
                Scope s(ctx, T_FIN);
                hipLaunchKernelGGL(k_finalize, dim3((unsigned)(cols / 256)), dim3(256), 0, ctx->stream,
                                   (const double*)ctx->partial, (const double*)ctx->mupart, (long)ctx->chunk, nb, nkc,
                                   (const double*)ctx->params, ctx->mu, ctx->var, (long)j0, (long)ctx->m,
                                   keep_inc ? ctx->sres : (double*)nullptr, (const double*)ctx->Xs, ctx->d);
            }
        }
        {
            Scope s(ctx, T_FIN);
            CHK(reduce_max(ctx, ctx->mu, ctx->m, ctx->inc_dev + 1));
        }
        if (lat)      // factor tables and B for the training points of the fit; every append adds one row
            hipLaunchKernelGGL(k_lattice_rows, dim3((unsigned)((std::max<int64_t>(ctx->lat_qpad, ctx->lat.split * ctx->lat_smax) + 255) / 256), (unsigned)ctx->n),
                               dim3(256), 0, ctx->stream, ctx->lat, (const double*)ctx->lat_axes, (const double*)ctx->params,
                               (const double*)ctx->Xts, (long)ctx->n_pad, 0, ctx->latE, (int)ctx->lat_kcap, (int)ctx->lat_smax, ctx->latB,
                               (long)ctx->lat_qpad);
    }
    HIPCHK(hipGetLastError());
    double mu_max = 0;
    HIPCHK(hipMemcpyAsync(&mu_max, ctx->inc_dev + 1, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    timing_collect(ctx, ms, cnt);
    for (int i = 0; i < T_NCAT; ++i) ms_acc[i] += ms[i];
    ctx->tm.posterior_total_ms = ms_acc[T_TOTAL];
    ctx->tm.ksbuild_ms = ms_acc[T_KS];
    ctx->tm.trmm_ms = ms_acc[T_TRMM];
    ctx->tm.trmm_launches = trmm_launches;
    ctx->tm.finalize_ms = ms_acc[T_FIN];
    ctx->tm.trmm_flops = (double)ctx->n * (double)ctx->n * (double)ctx->m;
    ctx->have_posterior = true;
    ctx->have_acq = false;
    ctx->inc_valid = keep_inc;
    ctx->inc_lattice = lat;
    ctx->mu_argmax_valid = false;
    ctx->var_stale = false;
    if (mu_max_out) *mu_max_out = mu_max;
    return GPBO_OK;
}

template <int D>
void launch_sweep_t(gpbo_ctx* ctx) {
    // two candidates per thread when every row of V starts 16-byte aligned (hipMalloc aligns the base; pitch = m)
    // the launch's first `extra` workgroups write the new row of W and the enlarged alpha (columns 0 .. n)
    const int extra = (int)((ctx->n + 1 + 255) / 256);
    if (ctx->m % 2 == 0)
        hipLaunchKernelGGL((k_append_sweep<D, 2>), dim3((unsigned)((ctx->m / 2 + 255) / 256 + extra)), dim3(256), 0, ctx->stream,
                           (const double*)ctx->Xs, (long)ctx->m, ctx->d, (const double*)ctx->xnew_dev, (const double*)ctx->params,
                           (const double*)ctx->lvec, (const double*)ctx->scal, ctx->Vres, (long)ctx->m, (int)ctx->n, ctx->sres, ctx->mu,
                           ctx->var, ctx->kernel, ctx->bmax, extra, ctx->bufW, (long)ctx->n_pad, (const double*)ctx->zfw, ctx->alpha1,
                           ctx->alpha);
    else
        hipLaunchKernelGGL((k_append_sweep<D, 1>), dim3((unsigned)((ctx->m + 255) / 256 + extra)), dim3(256), 0, ctx->stream,
                           (const double*)ctx->Xs, (long)ctx->m, ctx->d, (const double*)ctx->xnew_dev, (const double*)ctx->params,
                           (const double*)ctx->lvec, (const double*)ctx->scal, ctx->Vres, (long)ctx->m, (int)ctx->n, ctx->sres, ctx->mu,
                           ctx->var, ctx->kernel, ctx->bmax, extra, ctx->bufW, (long)ctx->n_pad, (const double*)ctx->zfw, ctx->alpha1,
                           ctx->alpha);
}

// ---- one BO-loop step in two halves: enqueue (launches only) and finish (one copy + one synchronisation), so that the
// append and the acquisition that follows it can share a single host round trip (gpbo_append_acquire).
int append_check(gpbo_ctx* ctx, const double* x_new) {
    if (!ctx->fitted || !ctx->have_factor) BADARG("append: no fitted model with a factor on this context");
    if (ctx->variant != GPBO_VARIANT_SINGLE) BADARG("append: only the single-factor variant can be updated incrementally");
    if (!x_new) BADARG("append: null x_new");
    if (!ctx->inc_valid || ctx->m != ctx->xs_rows) BADARG("append: call gpbo_reserve_appends, gpbo_fit, gpbo_set_candidates and gpbo_posterior first");
    if (ctx->reserve_left <= 0 || ctx->n + 1 > ctx->n_pad || ctx->n + 1 > (ctx->inc_lattice ? ctx->lat_kcap : ctx->v_cap_rows))
        BADARG("append: reserved rows exhausted (gpbo_reserve_appends)");
    return GPBO_OK;
}

// factor update (O(n^2)) + the HBM-streaming sweep over the resident V with max(mu) folded into its epilogue.
// Leaves max(mu) in inc_dev[1].  Call timing_begin first.
// What the acquisition that follows an append wants, for the routes that can carry it inside the update launch.
struct AcqRequest {
    int acq, mode;
    double kappa, incumbent;
    bool carried = false;        // out: the update launch evaluated the acquisition and the step's last launch was enqueued
};

// GPBO_LATTICE_SPEC=0: the lattice route keeps the acquisition in a launch of its own.
bool lattice_spec_enabled() {
    static const bool on = !(getenv("GPBO_LATTICE_SPEC") && atoi(getenv("GPBO_LATTICE_SPEC")) == 0);
    return on;
}

template <int ACQ>
void launch_update_acq(gpbo_ctx* ctx, long tiles, int tiles_q, int kpad, double kappa) {
    TileBest tb{ctx->lat_tile_v, ctx->lat_tile_i, ctx->lat_bmax, ctx->lat_mu_i};
    hipLaunchKernelGGL(k_lattice_gemm_update_acq<ACQ>, dim3((unsigned)tiles), dim3(GEMM_THREADS), 0, ctx->stream, (const double*)ctx->latA,
                       (long)ctx->lat_kcap, (const double*)ctx->latB, (long)ctx->lat_qpad, kpad, ctx->lat.P, ctx->lat.Q, tiles_q,
                       (const double*)ctx->scal, (const double*)ctx->params, (const double*)ctx->spec_dev, kappa, ctx->sres, ctx->mu, tb);
}

int append_enqueue(gpbo_ctx* ctx, const double* x_new, double y_new, bool defer_last_max = false, AcqRequest* rq = nullptr) {
    const int n = (int)ctx->n;
    const long ld = ctx->n_pad;
    PointArg pa;
    for (int t = 0; t < MAX_D; ++t) pa.x[t] = (t < ctx->d) ? x_new[t] / ctx->theta[t] : 0.0;
    Scope tot(ctx, T_TOTAL);
    // small models: the chain of the factor update (and the lattice route's new table row) is one launch (append_fused.hpp);
    // GPBO_APPEND_FUSED=0 keeps the separate launches
    static const bool fused_on = !(getenv("GPBO_APPEND_FUSED") && atoi(getenv("GPBO_APPEND_FUSED")) == 0);
    const bool fused = fused_on && n < APPEND_FUSED_MAX_N;
    const long rows_elems = ctx->inc_lattice ? (long)std::max<int64_t>(ctx->lat_qpad, ctx->lat.split * ctx->lat_smax) : 0;
    if (fused) {
        AppendFusedArgs a{};
        a.Xts = ctx->Xts;
        a.n = n;
        a.n_pad = (int)ctx->n_pad;
        a.d = ctx->d;
        a.kernel = ctx->kernel;
        a.xnew_out = ctx->xnew_dev;
        a.info = ctx->info_dev;
        a.kvec = ctx->kvec;
        a.lvec = ctx->lvec;
        a.zfw = ctx->zfw;
        a.yv = ctx->yv;
        a.scal = ctx->scal;
        a.params = ctx->params;
        a.L = ctx->bufL;
        a.W = ctx->bufW;
        a.ld = ld;
        a.y_new = y_new;
        a.eps = ctx->eps;
        a.with_wrow = ctx->inc_lattice ? 1 : 0;
        a.alpha1 = ctx->alpha1;
        a.alpha = ctx->alpha;
        a.rows_elems = rows_elems;
        a.axes = ctx->lat_axes;
        a.E = ctx->latE;
        a.B = ctx->latB;
        a.kcap = (int)ctx->lat_kcap;
        a.smax = (int)ctx->lat_smax;
        a.ldb = (long)ctx->lat_qpad;
        const unsigned extra = (unsigned)((rows_elems + APPEND_FUSED_THREADS - 1) / APPEND_FUSED_THREADS);
        hipLaunchKernelGGL(k_append_fused, dim3(1 + extra), dim3(APPEND_FUSED_THREADS), 0, ctx->stream, a, pa, ctx->lat);
    } else {
        hipLaunchKernelGGL(k_append_kvec, dim3((unsigned)((ctx->n_pad + 255) / 256)), dim3(256), 0, ctx->stream, ctx->Xts, n,
                           (int)ctx->n_pad, ctx->d, pa, ctx->xnew_dev, ctx->info_dev, ctx->kvec, ctx->kernel);
        hipLaunchKernelGGL(k_append_lvec, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, (const double*)ctx->bufW, ld, n,
                           (const double*)ctx->kvec, ctx->lvec);
        hipLaunchKernelGGL(k_append_scalars, dim3(1), dim3(256), 0, ctx->stream, (const double*)ctx->lvec, ctx->zfw, ctx->yv, n,
                           y_new, ctx->eps, ctx->scal, ctx->params, ctx->bufL, ctx->bufW, ld, ctx->info_dev);
    }
    if (ctx->inc_lattice) {
        Scope sw(ctx, T_SWEEP);
        const int kpad = (int)round_up(n + 1, TK);
        const int tiles_q = (int)(ctx->lat_qpad / QUARTER_M);
        const long tiles = (long)(ctx->lat_ppad / QUARTER_M) * tiles_q;
        if (!fused) {
            hipLaunchKernelGGL(k_append_wrow, dim3((unsigned)((n + 1 + 31) / 32)), dim3(256), 0, ctx->stream, ctx->bufW, ld, n,
                               (const double*)ctx->lvec, (const double*)ctx->scal, (const double*)ctx->zfw, ctx->alpha1, ctx->alpha);
            hipLaunchKernelGGL(k_lattice_rows, dim3((unsigned)((rows_elems + 255) / 256), 1u), dim3(256), 0, ctx->stream, ctx->lat,
                               (const double*)ctx->lat_axes, (const double*)ctx->params, (const double*)ctx->Xts, (long)ctx->n_pad, n,
                               ctx->latE, (int)ctx->lat_kcap, (int)ctx->lat_smax, ctx->latB, (long)ctx->lat_qpad);
        }
        // with an acquisition to follow, the update launch evaluates it as well (lattice.hpp: SpecArgs); the incumbent it
        // needs is formed by one more workgroup of the launch that builds A
        const bool carry = rq && lattice_spec_enabled();
        SpecArgs sp{};
        if (carry) {
            sp.spec = ctx->spec_dev;
            sp.mode = rq->acq == GPBO_ACQ_UCB ? 2 : (rq->mode == GPBO_INCUMBENT_MAX_MU ? 0 : (rq->mode == GPBO_INCUMBENT_MAX_Y ? 1 : 2));
            sp.have = ctx->mu_argmax_valid ? 1 : 0;
            sp.given = rq->acq == GPBO_ACQ_UCB ? 0.0 : rq->incumbent;
            sp.B = ctx->latB;
            sp.ldb = (long)ctx->lat_qpad;
            sp.mu = ctx->mu;
            sp.scal = ctx->scal;
            sp.params = ctx->params;
            sp.kpad = kpad;
        }
        hipLaunchKernelGGL(k_lattice_A, dim3((unsigned)((ctx->lat.P * kpad + 255) / 256) + (carry ? 1u : 0u)), dim3(256), 0, ctx->stream, ctx->lat,
                           (const double*)ctx->latE, (int)ctx->lat_kcap, (int)ctx->lat_smax, (const double*)(ctx->bufW + (long)n * ld), n,
                           ctx->latA, (long)ctx->lat_kcap, kpad, sp);
        if (carry) {
            switch (rq->acq) {
                case GPBO_ACQ_EI: launch_update_acq<0>(ctx, tiles, tiles_q, kpad, rq->kappa); break;
                case GPBO_ACQ_UCB: launch_update_acq<1>(ctx, tiles, tiles_q, kpad, rq->kappa); break;
                default: launch_update_acq<2>(ctx, tiles, tiles_q, kpad, rq->kappa); break;
            }
            TileBest tb{ctx->lat_tile_v, ctx->lat_tile_i, ctx->lat_bmax, ctx->lat_mu_i};
            hipLaunchKernelGGL(k_spec_finish, dim3(1), dim3(1024), 0, ctx->stream, tb, tiles, (const double*)ctx->params, (const int*)ctx->info_dev,
                               ctx->spec_dev, sp.mode == 0 ? 1 : 0, sp.have, ctx->inc_dev, ctx->res_dev);
            ctx->var_stale = true;
            ctx->mu_argmax_valid = true;      // (k_spec_finish leaves the index of the largest mean behind)
            rq->carried = true;
            HIPCHK(hipGetLastError());
            return GPBO_OK;
        }
        hipLaunchKernelGGL(k_lattice_gemm_update, dim3((unsigned)tiles), dim3(GEMM_THREADS), 0, ctx->stream, (const double*)ctx->latA,
                           (long)ctx->lat_kcap, (const double*)ctx->latB, (long)ctx->lat_qpad, kpad, ctx->lat.P, ctx->lat.Q, tiles_q,
                           (const double*)ctx->scal, ctx->sres, ctx->mu, ctx->lat_bmax);
        ctx->var_stale = true;
        ctx->mu_argmax_valid = false;
        ctx->sweep_parts = tiles;
        ctx->sweep_part_buf = ctx->lat_bmax;
    } else {
        Scope sw(ctx, T_SWEEP);
        switch (ctx->d) {
            case 1: launch_sweep_t<1>(ctx); break;
            case 2: launch_sweep_t<2>(ctx); break;
            case 3: launch_sweep_t<3>(ctx); break;
            case 4: launch_sweep_t<4>(ctx); break;
            case 5: launch_sweep_t<5>(ctx); break;
            case 6: launch_sweep_t<6>(ctx); break;
            case 7: launch_sweep_t<7>(ctx); break;
            case 8: launch_sweep_t<8>(ctx); break;
            default: launch_sweep_t<0>(ctx); break;
        }
        ctx->sweep_parts = (ctx->m % 2 == 0) ? (ctx->m / 2 + 255) / 256 : (ctx->m + 255) / 256;
        ctx->sweep_part_buf = ctx->bmax;
    }
    // stages 2 and 3 of max(mu) over the sweep's per-workgroup maxima; the last one also fills the incumbent slot
    // (used when the acquisition that follows takes max(mu) as its incumbent)
    const long sweep_blocks = ctx->sweep_parts;
    const int b2 = (int)std::min<long>(256, (sweep_blocks + 255) / 256);
    hipLaunchKernelGGL(k_max_stage, dim3(b2), dim3(256), 0, ctx->stream, (const double*)ctx->sweep_part_buf, sweep_blocks, ctx->red_v, (double*)nullptr);
    // (deferred: the acquisition kernel that follows takes the maximum of the b2 second-stage results itself)
    if (!defer_last_max)
        hipLaunchKernelGGL(k_max_stage, dim3(1), dim3(256), 0, ctx->stream, (const double*)ctx->red_v, (long)b2, ctx->inc_dev + 1, ctx->inc_dev);
    HIPCHK(hipGetLastError());
    return GPBO_OK;
}

// acquisition + argmax stage 1 fused; the per-workgroup results stay in acq_pv / acq_pi.  store: keep the acquisition array.
int acquire_enqueue(gpbo_ctx* ctx, int acq, double kappa, int mode, double incumbent, bool store, bool inc_ready = false) {
    if (acq < 0 || acq > 2) BADARG("acquire: unknown acquisition");
    Scope s(ctx, T_ACQ);
    const double* inc_parts = nullptr;
    int n_parts = 0;
    if (mode == GPBO_INCUMBENT_MAX_MU && inc_ready) {
        // the append sweep that was just enqueued left the second-stage maxima of mu in red_v: the acquisition kernel reduces
        // them itself and fills the incumbent and max(mu) slots (the third stage has no launch of its own)
        n_parts = (int)std::min<long>(256, (ctx->sweep_parts + 255) / 256);
        inc_parts = ctx->red_v;
    } else if (mode == GPBO_INCUMBENT_MAX_MU)
        hipLaunchKernelGGL(k_copy_scalar, dim3(1), dim3(1), 0, ctx->stream, ctx->inc_dev, (const double*)(ctx->inc_dev + 1));
    else if (mode == GPBO_INCUMBENT_MAX_Y)
        hipLaunchKernelGGL(k_copy_scalar, dim3(1), dim3(1), 0, ctx->stream, ctx->inc_dev, (const double*)(ctx->params + P_YMAX));
    else if (mode == GPBO_INCUMBENT_GIVEN)
        hipLaunchKernelGGL(k_set_scalar, dim3(1), dim3(1), 0, ctx->stream, ctx->inc_dev, incumbent);
    else
        BADARG("acquire: unknown incumbent mode");
    const unsigned blocks = (unsigned)((ctx->m + ACQ_SPAN - 1) / ACQ_SPAN);
    // after lattice-route appends the variance array has not been written: the pass forms the variance from sum v^2
    const double* vsrc = ctx->var_stale ? (const double*)ctx->sres : (const double*)ctx->var;
    const double* c01 = ctx->var_stale ? (const double*)(ctx->params + P_C0) : (const double*)nullptr;
    if (store)
        hipLaunchKernelGGL(k_acquire_argmax<true>, dim3(blocks), dim3(256), 0, ctx->stream, (const double*)ctx->mu, vsrc,
                           (long)ctx->m, acq, kappa, (const double*)ctx->inc_dev, ctx->acq, ctx->acq_pv, ctx->acq_pi, inc_parts, n_parts,
                           ctx->inc_dev, c01);
    else if (acq == GPBO_ACQ_EI)      // the loop's pass, one instantiation per acquisition: no branches on it inside the eight evaluations of a thread
        hipLaunchKernelGGL((k_acquire_argmax<false, 0>), dim3(blocks), dim3(256), 0, ctx->stream, (const double*)ctx->mu, vsrc,
                           (long)ctx->m, acq, kappa, (const double*)ctx->inc_dev, ctx->acq, ctx->acq_pv, ctx->acq_pi, inc_parts, n_parts,
                           ctx->inc_dev, c01);
    else if (acq == GPBO_ACQ_UCB)
        hipLaunchKernelGGL((k_acquire_argmax<false, 1>), dim3(blocks), dim3(256), 0, ctx->stream, (const double*)ctx->mu, vsrc,
                           (long)ctx->m, acq, kappa, (const double*)ctx->inc_dev, ctx->acq, ctx->acq_pv, ctx->acq_pi, inc_parts, n_parts,
                           ctx->inc_dev, c01);
    else
        hipLaunchKernelGGL((k_acquire_argmax<false, 2>), dim3(blocks), dim3(256), 0, ctx->stream, (const double*)ctx->mu, vsrc,
                           (long)ctx->m, acq, kappa, (const double*)ctx->inc_dev, ctx->acq, ctx->acq_pv, ctx->acq_pi, inc_parts, n_parts,
                           ctx->inc_dev, c01);
    // (the last argmax stage, over the `blocks` per-workgroup results, runs inside the end-of-step launch: step_fetch)
    HIPCHK(hipGetLastError());
    ctx->have_acq = store;       // (not stored: the array still holds the values of an earlier posterior)
    return GPBO_OK;
}

struct StepResult {      // what one synchronisation brings back
    double par[8];
    double mu_max, best;
    long best_i;
    int info;
};

int step_fetch(gpbo_ctx* ctx, StepResult* r, bool want_append, bool want_acquire) {
    (void)want_append;
    // one small kernel stores everything into the pinned result block; the synchronisation is the whole transfer.  After an
    // acquisition the same launch first finishes the argmax over the per-workgroup results.
    if (want_acquire)
        hipLaunchKernelGGL(k_argmax_step_end, dim3(1), dim3(1024), 0, ctx->stream, (const double*)ctx->acq_pv, (const long*)ctx->acq_pi,
                           (long)((ctx->m + ACQ_SPAN - 1) / ACQ_SPAN), (const double*)ctx->params, (const int*)ctx->info_dev,
                           (const double*)(ctx->inc_dev + 1), ctx->res_dev);
    else
        hipLaunchKernelGGL(k_step_end, dim3(1), dim3(64), 0, ctx->stream, (const double*)ctx->params, (const int*)ctx->info_dev,
                           (const double*)(ctx->inc_dev + 1), (const double*)ctx->red_v2, (const long*)ctx->red_i2, ctx->res_dev);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    memcpy(r->par, ctx->res_host, sizeof(r->par));
    r->info = (int)ctx->res_host[8];
    r->mu_max = ctx->res_host[9];
    r->best = ctx->res_host[10];
    memcpy(&r->best_i, ctx->res_host + 11, sizeof(long));
    return GPBO_OK;
}

int append_finish(gpbo_ctx* ctx, const StepResult& r, double* mu_max_out) {
    double ms[T_NCAT], cnt[T_NCAT];
    timing_collect(ctx, ms, cnt);
    ctx->tm.append_total_ms = ms[T_TOTAL];
    ctx->tm.append_sweep_ms = ms[T_SWEEP];
    ctx->tm.append_route = ctx->inc_lattice ? 1.0 : 0.0;
    ctx->tm.acquire_ms = ms[T_ACQ];
    if (r.info != 0) {
        ctx->info = r.info;
        ctx->fitted = false;     // the factor row that was written is not usable
        ctx->inc_valid = false;
        ctx->have_posterior = false;
        ctx->have_acq = false;
        ctx->err = "append: enlarged matrix is not positive definite";
        return GPBO_ERR_NOT_PD;
    }
    ctx->n += 1;
    ctx->reserve_left -= 1;
    ctx->theta1 = r.par[P_THETA1];
    ctx->nll = r.par[P_NLL];
    ctx->y_max = r.par[P_YMAX];
    if (mu_max_out) *mu_max_out = r.mu_max;
    return GPBO_OK;
}

int append_impl(gpbo_ctx* ctx, const double* x_new, double y_new, double* mu_max_out) {
    CHK(append_check(ctx, x_new));
    ctx->have_acq = false;       // the posterior changes under the stored acquisition values
    timing_begin(ctx);
    CHK(append_enqueue(ctx, x_new, y_new));
    StepResult r{};
    CHK(step_fetch(ctx, &r, true, false));
    return append_finish(ctx, r, mu_max_out);
}

int acquire_impl(gpbo_ctx* ctx, int acq, double kappa, int mode, double incumbent, int64_t* argmax_out, double* best_out) {
    if (!ctx->have_posterior) BADARG("acquire: no posterior (call gpbo_posterior first)");
    timing_begin(ctx);
    CHK(acquire_enqueue(ctx, acq, kappa, mode, incumbent, true));
    StepResult r{};
    CHK(step_fetch(ctx, &r, false, true));
    double ms[T_NCAT], cnt[T_NCAT];
    timing_collect(ctx, ms, cnt);
    ctx->tm.acquire_ms = ms[T_ACQ];
    if (r.best_i == INT64_MAX) r.best_i = 0;   // (no candidate at all; a NaN or -inf value is picked by its index like any other)
    if (argmax_out) *argmax_out = r.best_i;
    if (best_out) *best_out = r.best;
    return GPBO_OK;
}

// append + the acquisition over the updated posterior, one synchronisation.  The acquisition array is not stored
// (gpbo_read(acq) and gpbo_first_at_least are refused until the next gpbo_acquire: the values they would see are those of
// an earlier posterior).
int append_acquire_impl(gpbo_ctx* ctx, const double* x_new, double y_new, int acq, double kappa, int mode, double incumbent,
                        double* mu_max_out, int64_t* argmax_out, double* best_out) {
    CHK(append_check(ctx, x_new));
    if (acq < 0 || acq > 2) BADARG("append_acquire: unknown acquisition");
    if (mode < 0 || mode > 2) BADARG("append_acquire: unknown incumbent mode");
    timing_begin(ctx);
    AcqRequest rq{acq, mode, kappa, incumbent};
    CHK(append_enqueue(ctx, x_new, y_new, mode == GPBO_INCUMBENT_MAX_MU, ctx->inc_lattice ? &rq : nullptr));
    StepResult r{};
    bool settled = false, spec_hit = false;
    if (rq.carried) {
        // the update launch evaluated the acquisition and the step's last launch has written the result block; what is left
        // to find out is whether the incumbent it worked with was the right one
        HIPCHK(hipStreamSynchronize(ctx->stream));
        settled = spec_hit = ctx->res_host[12] != 0.0;
        if (settled) {
            memcpy(r.par, ctx->res_host, sizeof(r.par));
            r.info = (int)ctx->res_host[8];
            r.mu_max = ctx->res_host[9];
            r.best = ctx->res_host[10];
            memcpy(&r.best_i, ctx->res_host + 11, sizeof(long));
        } else if ((int)ctx->res_host[8] != 0) {
            memcpy(r.par, ctx->res_host, sizeof(r.par));       // the append itself failed: no acquisition to repeat
            r.info = (int)ctx->res_host[8];
            r.mu_max = ctx->res_host[9];
            settled = true;
        }
    }
    if (!settled) {
        // (after a carried update: max(mu) is in the incumbent slots, k_spec_finish put it there)
        CHK(acquire_enqueue(ctx, acq, kappa, mode, incumbent, false, !rq.carried));
        CHK(step_fetch(ctx, &r, true, true));
    }
    CHK(append_finish(ctx, r, mu_max_out));
    if (spec_hit) ctx->tm.append_route = 2.0;      // lattice route, acquisition carried by the update launch and kept
    if (r.best_i == INT64_MAX) r.best_i = 0;
    if (argmax_out) *argmax_out = r.best_i;
    if (best_out) *best_out = r.best;
    return GPBO_OK;
}

// mu / var / acq for m candidates and the per-workgroup partial arrays sized by m
int ensure_outputs(gpbo_ctx* ctx, int64_t m) {
    if (m <= ctx->m_cap) return GPBO_OK;
    CHK(dev_alloc(ctx, &ctx->mu, m));
    CHK(dev_alloc(ctx, &ctx->var, m));
    CHK(dev_alloc(ctx, &ctx->acq, m));
    CHK(dev_alloc(ctx, &ctx->acq_pv, (m + ACQ_SPAN - 1) / ACQ_SPAN));
    CHK(dev_alloc(ctx, &ctx->acq_pi, (m + ACQ_SPAN - 1) / ACQ_SPAN));
    CHK(dev_alloc(ctx, &ctx->bmax, (m + 255) / 256));
    ctx->m_cap = m;
    return GPBO_OK;
}

int reserve_candidates(gpbo_ctx* ctx, int64_t m) {
    if (m <= 0) BADARG("set_candidates: empty candidate set");
    if (!ctx->fitted) BADARG("set_candidates: fit first (the candidate width is the model's d)");
    if (m * ctx->d > ctx->xs_cap) {
        CHK(dev_alloc(ctx, &ctx->Xs, (size_t)m * ctx->d));
        ctx->xs_cap = m * ctx->d;
    }
    CHK(ensure_outputs(ctx, m));
    ctx->m = m;
    ctx->xs_rows = m;
    ctx->have_posterior = false;
    ctx->have_acq = false;
    ctx->inc_valid = false;
    ctx->lattice = false;        // (gpbo_set_candidates_lattice sets it again)
    return GPBO_OK;
}

int set_candidates_impl(gpbo_ctx* ctx, const double* Xs, int64_t m, int on_device) {
    if (!Xs) BADARG("set_candidates: null candidate pointer");
    CHK(reserve_candidates(ctx, m));
    HIPCHK(hipMemcpyAsync(ctx->Xs, Xs, (size_t)m * ctx->d * sizeof(double),
                          on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return GPBO_OK;
}

}  // namespace

// ======================================================================== extern "C"
extern "C" {

int gpbo_create(int device, gpbo_ctx** out) {
    if (!out) return GPBO_ERR_BAD_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return GPBO_ERR_HIP;
    gpbo_ctx* ctx = new gpbo_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&ctx->stream) != hipSuccess) {
        delete ctx;
        return GPBO_ERR_HIP;
    }
    bool ok = hipMalloc((void**)&ctx->params, PARAMS_DOUBLES * sizeof(double)) == hipSuccess &&
              hipMalloc((void**)&ctx->info_dev, sizeof(int)) == hipSuccess &&
              hipMalloc((void**)&ctx->red_v, 1024 * sizeof(double)) == hipSuccess &&
              hipMalloc((void**)&ctx->red_i, 1024 * sizeof(long)) == hipSuccess &&
              hipMalloc((void**)&ctx->red_v2, 8 * sizeof(double)) == hipSuccess &&
              hipMalloc((void**)&ctx->red_i2, 8 * sizeof(long)) == hipSuccess &&
              hipMalloc((void**)&ctx->inc_dev, 8 * sizeof(double)) == hipSuccess &&
              hipMalloc((void**)&ctx->scal, 8 * sizeof(double)) == hipSuccess &&
              hipMalloc((void**)&ctx->xnew_dev, MAX_D * sizeof(double)) == hipSuccess &&
              hipHostMalloc((void**)&ctx->res_host, STEP_RES_DOUBLES * sizeof(double), hipHostMallocMapped) == hipSuccess &&
              hipHostGetDevicePointer((void**)&ctx->res_dev, ctx->res_host, 0) == hipSuccess;
    if (!ok) {
        gpbo_destroy(ctx);
        return GPBO_ERR_HIP;
    }
    hipMemset(ctx->params, 0, PARAMS_DOUBLES * sizeof(double));
    *out = ctx;
    return GPBO_OK;
}

void gpbo_destroy(gpbo_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    for (gpbo_ctx* l : ctx->lml_lanes) gpbo_destroy(l);
    ctx->lml_lanes.clear();
    if (ctx->lml_lock) gpbo_destroy(ctx->lml_lock);
    ctx->lml_lock = nullptr;
    if (ctx->kr_work) gpbo_destroy(ctx->kr_work);
    ctx->kr_work = nullptr;
    if (ctx->ipiv) hipFree(ctx->ipiv);
    ctx->ipiv = nullptr;
    free_model(ctx);
    void* ps[] = {ctx->params, ctx->info_dev, ctx->red_v, ctx->red_i, ctx->red_v2, ctx->red_i2, ctx->inc_dev,
                  ctx->Xs,     ctx->mu,       ctx->var,   ctx->acq,   ctx->Bmat,   ctx->partial, ctx->mupart,
                  ctx->scal,   ctx->xnew_dev, ctx->Vres,  ctx->sres,     ctx->acq_pv,  ctx->acq_pi, ctx->bmax, ctx->Zmat,
                  ctx->lat_axes, ctx->latA,   ctx->latB,  ctx->latE,     ctx->lat_bmax, ctx->lat_tile_v, ctx->lat_tile_i, ctx->lat_mu_i, ctx->spec_dev};
    for (void* p : ps)
        if (p) hipFree(p);
    if (ctx->res_host) hipHostFree(ctx->res_host);
    for (auto e : ctx->pool) hipEventDestroy(e);
    if (ctx->sm_buf) hipFree(ctx->sm_buf);
    if (ctx->sm_host) hipHostFree(ctx->sm_host);
    if (ctx->lane_graph) hipGraphExecDestroy(ctx->lane_graph);
    for (void* p : {(void*)ctx->lb_xraw, (void*)ctx->lb_th, (void*)ctx->lb_sg, (void*)ctx->lb_nll, (void*)ctx->lb_grad,
                    (void*)ctx->lb_info, (void*)ctx->lane_gpart, (void*)ctx->lane_slot})
        if (p) hipFree(p);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* gpbo_last_error(gpbo_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
int64_t gpbo_last_info(gpbo_ctx* ctx) { return ctx ? ctx->info : 0; }
void* gpbo_stream(gpbo_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int gpbo_set_chunk(gpbo_ctx* ctx, int64_t chunk) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    if (chunk < 0) BADARG("set_chunk: negative");
    ctx->chunk_req = chunk;
    return GPBO_OK;
}

int gpbo_fit(gpbo_ctx* ctx, const double* X, const double* y, int64_t n, int32_t d, const double* theta, double sigmaY,
             int32_t kernel, int32_t variant, double* theta1_out, double* nll_out) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    if (!X || !y || !theta) BADARG("fit: null input");
    if (n <= 0 || d <= 0 || d > MAX_D) BADARG("fit: need n >= 1 and 1 <= d <= 32");
    if (kernel != GPBO_KERNEL_RBF && kernel != GPBO_KERNEL_MATERN52) BADARG("fit: unknown kernel");
    if (variant != GPBO_VARIANT_V4 && variant != GPBO_VARIANT_SINGLE && variant != GPBO_VARIANT_M3) BADARG("fit: unknown variant");
    if (!(sigmaY >= 0)) BADARG("fit: sigmaY must be >= 0");
    for (int t = 0; t < d; ++t)
        if (!(theta[t] != 0.0) || !std::isfinite(theta[t])) BADARG("fit: theta entries must be finite and non-zero");
    HIPCHK(hipSetDevice(ctx->device));
    const int64_t n_pad = round_up(n + (variant == GPBO_VARIANT_SINGLE ? ctx->reserve : 0), PB);
    CHK(ensure_model(ctx, n_pad, d, true));
    if (d != ctx->d) ctx->xs_rows = 0;            // resident candidates have the old column count: dropped
    ctx->m = ctx->xs_rows;                        // (an uploaded posterior of another length goes with the old model)
    ctx->fitted = false;
    ctx->have_posterior = false;
    ctx->have_acq = false;
    ctx->inc_valid = false;
    ctx->n = n;
    ctx->n_pad = n_pad;
    ctx->nb = n_pad / PB;
    ctx->d = d;
    ctx->kernel = kernel;
    ctx->variant = variant;
    ctx->sigmaY = sigmaY;
    // BayesOpt_v4.ipynb:138-141; the MATLAB two-factor form uses eps * length(K) (GP/BayesOpt_v3.m:343,352)
    ctx->eps = (sigmaY == 0) ? (variant == GPBO_VARIANT_M3 ? 2.220446049250313e-16 : 1e-15) * (double)n : sigmaY * sigmaY;
    ctx->theta.assign(theta, theta + d);
    ctx->info = 0;

    // host staging: X/theta dimension-major, padded y, 1/theta
    std::vector<double> xts((size_t)d * n_pad, 0.0), yp(n_pad, 0.0), par(PARAMS_DOUBLES, 0.0);
    for (int t = 0; t < d; ++t) {
        const double it = 1.0 / theta[t];
        par[P_INVTHETA + t] = it;
        for (int64_t i = 0; i < n; ++i) xts[(size_t)t * n_pad + i] = X[i * d + t] * it;
    }
    for (int64_t i = 0; i < n; ++i) yp[i] = y[i];
    HIPCHK(hipMemcpyAsync(ctx->Xts, xts.data(), xts.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->yv, yp.data(), yp.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->params, par.data(), par.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->info_dev, 0, sizeof(int), ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));   // staging vectors die at scope end

    timing_begin(ctx);
    double nchol = 0;
    {
        Scope tot(ctx, T_TOTAL);
        CHK(launch_kernel_matrix(ctx, ctx->bufA, nullptr, ctx->eps));
        CHK(launch_cholesky(ctx, ctx->bufA, ctx->info_dev));
        nchol += 1;
        CHK(launch_trsv(ctx, ctx->bufA, ctx->eps));
        hipLaunchKernelGGL(k_fit_scalars, dim3(1), dim3(256), 0, ctx->stream, (const double*)ctx->bufA, (long)n_pad,
                           (const double*)ctx->yv, (const double*)ctx->alpha1, ctx->alpha, (int)n, (int)n_pad, variant, ctx->params, ctx->bs);
        if (variant == GPBO_VARIANT_V4 || variant == GPBO_VARIANT_M3) {
            CHK(launch_kernel_matrix(ctx, ctx->bufB, ctx->params + P_THETA1, ctx->eps));   // theta1*K0 + eps, :146-147
            CHK(launch_cholesky(ctx, ctx->bufB, ctx->info_dev));
            nchol += 1;
            if (variant == GPBO_VARIANT_M3) {
                // the MATLAB form takes alpha from the SECOND factor (GP/BayesOpt_v3.m:352-356): Mu = (theta1 Ks0) (theta1 K0 + eps)^-1 y
                Scope s2(ctx, T_TRSV);
                CHK(launch_trsv_vec(ctx, ctx->bufB, ctx->yv, ctx->alpha, false, nullptr));
                // ... and reports -logL from that alpha and the second factor's diagonal (:362)
                hipLaunchKernelGGL(k_nll_second_factor, dim3(1), dim3(256), 0, ctx->stream, (const double*)ctx->bufB, (long)n_pad,
                                   (const double*)ctx->yv, (const double*)ctx->alpha, (int)n, ctx->params);
            }
            CHK(launch_inverse(ctx, ctx->bufB, ctx->bufW, ctx->bufA));
            ctx->bufL = ctx->bufB;
        } else {
            CHK(launch_inverse(ctx, ctx->bufA, ctx->bufW, ctx->bufB));
            ctx->bufL = ctx->bufA;
        }
    }
    HIPCHK(hipGetLastError());
    double par_out[8];
    int info = 0;
    HIPCHK(hipMemcpyAsync(par_out, ctx->params, sizeof(par_out), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(&info, ctx->info_dev, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    double ms[T_NCAT], cnt[T_NCAT];
    timing_collect(ctx, ms, cnt);
    ctx->tm.fit_total_ms = ms[T_TOTAL];
    ctx->tm.kbuild_ms = ms[T_KBUILD];
    ctx->tm.chol_ms = ms[T_CHOL];
    ctx->tm.chol_count = nchol;
    ctx->tm.trsv_ms = ms[T_TRSV];
    ctx->tm.inverse_ms = ms[T_INV];
    if (info != 0) {
        ctx->info = info;
        ctx->err = "fit: matrix is not positive definite (pivot <= 0 at row " + std::to_string(info) + ")";
        return GPBO_ERR_NOT_PD;
    }
    ctx->theta1 = par_out[P_THETA1];
    ctx->nll = par_out[P_NLL];
    ctx->y_max = par_out[P_YMAX];
    if (!(ctx->theta1 > 0) || !std::isfinite(ctx->theta1)) {
        // the second factor of a non-positive theta1 would have failed above; NaN means garbage in
        ctx->err = "fit: theta1 = y^T alpha / n is not positive/finite";
        ctx->info = 0;
        return GPBO_ERR_NOT_PD;
    }
    ctx->fitted = true;
    ctx->have_factor = true;
    ctx->reserve_left = (variant == GPBO_VARIANT_SINGLE) ? ctx->reserve : 0;
    if (theta1_out) *theta1_out = ctx->theta1;
    if (nll_out) *nll_out = ctx->nll;
    return GPBO_OK;
}

int gpbo_set_candidates(gpbo_ctx* ctx, const double* Xs, int64_t m, int32_t on_device) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    return set_candidates_impl(ctx, Xs, m, on_device);
}

int gpbo_set_candidates_uniform(gpbo_ctx* ctx, uint64_t seed, int64_t row_offset, int64_t m) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    if (row_offset < 0) BADARG("set_candidates_uniform: negative row offset");
    HIPCHK(hipSetDevice(ctx->device));
    CHK(reserve_candidates(ctx, m));
    const long count = (long)m * ctx->d;
    hipLaunchKernelGGL(k_uniform_splitmix, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, ctx->Xs, count,
                       (unsigned long long)seed, (unsigned long long)row_offset * (unsigned long long)ctx->d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return GPBO_OK;
}

int gpbo_set_candidates_lattice(gpbo_ctx* ctx, const double* axes, const int32_t* sizes, int32_t d) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    if (!axes || !sizes) BADARG("set_candidates_lattice: null input");
    if (!ctx->fitted) BADARG("set_candidates: fit first (the candidate width is the model's d)");
    if (d != ctx->d) BADARG("set_candidates_lattice: the number of axes must equal the model's d");
    LatticeDesc L{};
    L.d = d;
    L.split = (d + 1) / 2;       // a fixed rule: the grouping of the factors must not depend on how a lattice was cut into slabs
    int64_t total = 0, m = 1;
    L.P = 1;
    L.Q = 1;
    for (int t = 0; t < d; ++t) {
        if (sizes[t] < 1) BADARG("set_candidates_lattice: every axis needs at least one point");
        L.size[t] = sizes[t];
        L.off[t] = (int)total;
        total += sizes[t];
        if (m > (int64_t)1 << 40) BADARG("set_candidates_lattice: lattice too large");
        m *= sizes[t];
        if (t < L.split) L.P *= sizes[t];
        else L.Q *= sizes[t];
    }
    for (int64_t i = 0; i < total; ++i)
        if (!std::isfinite(axes[i])) BADARG("set_candidates_lattice: coordinates must be finite");
    HIPCHK(hipSetDevice(ctx->device));
    CHK(reserve_candidates(ctx, m));
    if (total > ctx->lat_axes_cap) {
        CHK(dev_alloc(ctx, &ctx->lat_axes, (size_t)total));
        ctx->lat_axes_cap = total;
    }
    HIPCHK(hipMemcpyAsync(ctx->lat_axes, axes, (size_t)total * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_lattice_points, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, L, (const double*)ctx->lat_axes, ctx->Xs,
                       (long)m);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));      // axes is the caller's memory
    ctx->lat = L;
    ctx->lattice = true;
    return GPBO_OK;
}

int gpbo_posterior(gpbo_ctx* ctx, double* mu_max_out) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    return posterior_impl(ctx, mu_max_out);
}

int gpbo_acquire(gpbo_ctx* ctx, int32_t acq, double kappa, int32_t incumbent_mode, double incumbent, int64_t* argmax_out,
                 double* best_out) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    return acquire_impl(ctx, acq, kappa, incumbent_mode, incumbent, argmax_out, best_out);
}

int gpbo_first_at_least(gpbo_ctx* ctx, double threshold, int64_t* index_out) {
    if (!ctx || !index_out) return GPBO_ERR_BAD_ARG;
    if (!ctx->have_posterior || !ctx->have_acq) BADARG("first_at_least: no acquisition values for the current posterior (call gpbo_acquire first)");
    HIPCHK(hipSetDevice(ctx->device));
    const int blocks = (int)std::min<int64_t>(1024, (ctx->m + 255) / 256);
    hipLaunchKernelGGL(k_first_ge_stage, dim3(blocks), dim3(256), 0, ctx->stream, (const double*)ctx->acq, (const long*)nullptr,
                       (long)ctx->m, threshold, ctx->red_i);
    hipLaunchKernelGGL(k_first_ge_stage, dim3(1), dim3(256), 0, ctx->stream, (const double*)nullptr, (const long*)ctx->red_i,
                       (long)blocks, threshold, ctx->red_i2);
    HIPCHK(hipGetLastError());
    long bi = 0;
    HIPCHK(hipMemcpyAsync(&bi, ctx->red_i2, sizeof(long), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *index_out = (bi == INT64_MAX) ? -1 : bi;
    return GPBO_OK;
}

int gpbo_read(gpbo_ctx* ctx, double* mu, double* var, double* acq, int64_t offset, int64_t count) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    if (offset < 0 || count < 0 || offset + count > ctx->m) BADARG("read: range outside the candidate set");
    if (!ctx->have_posterior) BADARG("read: no posterior");
    if (acq && !ctx->have_acq) BADARG("read: no acquisition values for the current posterior (call gpbo_acquire first)");
    HIPCHK(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)count * sizeof(double);
    if (mu) HIPCHK(hipMemcpyAsync(mu, ctx->mu + offset, bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (var) {
        if (ctx->var_stale) {      // lattice-route appends left the variance as sum v^2: fill the array now
            hipLaunchKernelGGL(k_var_from_s, dim3((unsigned)((ctx->m + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)ctx->sres,
                               (const double*)ctx->params, ctx->var, (long)ctx->m);
            HIPCHK(hipGetLastError());
            ctx->var_stale = false;
        }
        HIPCHK(hipMemcpyAsync(var, ctx->var + offset, bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (acq) HIPCHK(hipMemcpyAsync(acq, ctx->acq + offset, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return GPBO_OK;
}

int gpbo_predict_ei(gpbo_ctx* ctx, const double* Xs, int64_t m, double kappa, int32_t incumbent_mode, double incumbent,
                    double* mu_out, double* var_out, double* ei_out, int64_t* argmax_out, double* ei_max_out,
                    double* mu_max_out) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    CHK(set_candidates_impl(ctx, Xs, m, 0));
    CHK(posterior_impl(ctx, mu_max_out));
    CHK(acquire_impl(ctx, GPBO_ACQ_EI, kappa, incumbent_mode, incumbent, argmax_out, ei_max_out));
    return gpbo_read(ctx, mu_out, var_out, ei_out, 0, m);
}

int gpbo_get_factor(gpbo_ctx* ctx, double* L_out, double* alpha_out) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    if (!ctx->fitted) BADARG("get_factor: no fitted model");
    HIPCHK(hipSetDevice(ctx->device));
    if (L_out) {
        if (!ctx->have_factor) BADARG("get_factor: this context holds a received model without the factor");
        HIPCHK(hipMemcpy2DAsync(L_out, ctx->n * sizeof(double), ctx->bufL, ctx->n_pad * sizeof(double), ctx->n * sizeof(double),
                                ctx->n, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (alpha_out) HIPCHK(hipMemcpyAsync(alpha_out, ctx->alpha, ctx->n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return GPBO_OK;
}

int gpbo_model_scalars(gpbo_ctx* ctx, double* theta1_out, double* nll_out, double* y_max_out, int64_t* n_out) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    if (!ctx->fitted) BADARG("model_scalars: no fitted model");
    if (theta1_out) *theta1_out = ctx->theta1;
    if (nll_out) *nll_out = ctx->nll;
    if (y_max_out) *y_max_out = ctx->y_max;
    if (n_out) *n_out = ctx->n;
    return GPBO_OK;
}

int gpbo_shape(gpbo_ctx* ctx, int64_t* n_out, int32_t* d_out, int64_t* m_out, int64_t* candidate_rows_out) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    if (n_out) *n_out = ctx->n;
    if (d_out) *d_out = ctx->d;
    if (m_out) *m_out = ctx->m;
    if (candidate_rows_out) *candidate_rows_out = ctx->xs_rows;
    return GPBO_OK;
}

int gpbo_cond_estimate(gpbo_ctx* ctx, double* cond_out) {
    if (!ctx || !cond_out) return GPBO_ERR_BAD_ARG;
    if (!ctx->fitted || !ctx->have_factor || !ctx->bufL) BADARG("cond_estimate: no factor on this context");
    HIPCHK(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_diag_minmax, dim3(1), dim3(256), 0, ctx->stream, (const double*)ctx->bufL, (long)ctx->n_pad, (int)ctx->n,
                       ctx->red_v);
    double mm[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(mm, ctx->red_v, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *cond_out = (mm[1] / mm[0]) * (mm[1] / mm[0]);
    return GPBO_OK;
}

int gpbo_descent_small(gpbo_ctx* ctx, const double* X, const double* y, int64_t n, int32_t d, const double* W0, int32_t nruns,
                       double sigmaY, int32_t kernel, double alpha, double beta, int32_t norm, int32_t steprule, int32_t max_its,
                       double* w_hist_out, double* c_hist_out, int32_t* count_out, int32_t* flag_out) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    if (!X || !y || !W0 || !w_hist_out || !c_hist_out || !count_out || !flag_out) BADARG("descent_small: null argument");
    if (n <= 0 || n > PB || d <= 0 || d > GRAD_D) BADARG("descent_small: need 1 <= n <= 128 and 1 <= d <= 16");
    if (nruns <= 0 || max_its < 0 || max_its > (1 << 24)) BADARG("descent_small: need nruns >= 1 and 0 <= max_its <= 2^24");
    if (kernel != GPBO_KERNEL_RBF && kernel != GPBO_KERNEL_MATERN52) BADARG("descent_small: unknown kernel");
    if (norm < 0 || norm > 2 || steprule < 0 || steprule > 2) BADARG("descent_small: unknown norm or step rule");
    if (!(sigmaY >= 0)) BADARG("descent_small: sigmaY must be >= 0");
    for (int64_t i = 0; i < (int64_t)nruns * d; ++i)
        if (!(W0[i] != 0.0) || !std::isfinite(W0[i])) BADARG("descent_small: starting points must be finite and non-zero");
    HIPCHK(hipSetDevice(ctx->device));
    // one device block: [Xraw d*128][y 128][W0 nruns*d] in, [hist_w][hist_c][count, flag as int32] out; staged through
    // pinned memory like the one-workgroup likelihood path, whose buffers it shares
    const size_t H = (size_t)max_its + 1;
    const size_t n_in = (size_t)d * PB + PB + (size_t)nruns * d;
    const size_t n_hw = (size_t)nruns * H * d, n_hc = (size_t)nruns * H, n_ints = (size_t)nruns;      // 2 int32 per double slot
    const size_t n_out = n_hw + n_hc + n_ints;
    if (ctx->sm_cap < n_in + n_out) {
        if (ctx->sm_buf) hipFree(ctx->sm_buf);
        if (ctx->sm_host) hipHostFree(ctx->sm_host);
        ctx->sm_buf = nullptr;
        ctx->sm_host = nullptr;
        ctx->sm_cap = 0;
        const size_t cap = n_in + n_out + 4096;
        HIPCHK(hipMalloc((void**)&ctx->sm_buf, cap * 8));
        HIPCHK(hipHostMalloc((void**)&ctx->sm_host, cap * 8, hipHostMallocDefault));
        ctx->sm_cap = cap;
    }
    double* h = ctx->sm_host;
    std::fill(h, h + (size_t)d * PB + PB, 0.0);
    for (int t = 0; t < d; ++t)
        for (int64_t i = 0; i < n; ++i) h[(size_t)t * PB + i] = X[i * d + t];
    for (int64_t i = 0; i < n; ++i) h[(size_t)d * PB + i] = y[i];
    memcpy(h + (size_t)d * PB + PB, W0, (size_t)nruns * d * 8);
    double* dX = ctx->sm_buf;
    double* dy = dX + (size_t)d * PB;
    double* dW0 = dy + PB;
    double* dhw = ctx->sm_buf + n_in;
    double* dhc = dhw + n_hw;
    int* dcount = reinterpret_cast<int*>(dhc + n_hc);
    int* dflag = dcount + nruns;
    HIPCHK(hipMemcpyAsync(ctx->sm_buf, h, n_in * 8, hipMemcpyHostToDevice, ctx->stream));
    if (steprule == 2 && n > small_wide_min())
        hipLaunchKernelGGL(k_descent_backtrack_small<true>, dim3((unsigned)nruns), dim3(SMALL_WIDE_THREADS), 0, ctx->stream,
                           (const double*)dX, (int)PB, (const double*)dy, (int)n, (int)d, (const double*)dW0, sigmaY, (int)kernel, alpha,
                           beta, (int)norm, (int)max_its, dhw, dhc, dcount, dflag);
    else if (steprule == 2)
        hipLaunchKernelGGL(k_descent_backtrack_small<false>, dim3((unsigned)nruns), dim3(POTRF_THREADS), 0, ctx->stream,
                           (const double*)dX, (int)PB, (const double*)dy, (int)n, (int)d, (const double*)dW0, sigmaY, (int)kernel, alpha,
                           beta, (int)norm, (int)max_its, dhw, dhc, dcount, dflag);
    else if (n > small_wide_min())
        hipLaunchKernelGGL(k_descent_small<true>, dim3((unsigned)nruns), dim3(SMALL_WIDE_THREADS), 0, ctx->stream, (const double*)dX,
                           (int)PB, (const double*)dy, (int)n, (int)d, (const double*)dW0, sigmaY, (int)kernel, alpha, beta, (int)norm,
                           (int)steprule, (int)max_its, dhw, dhc, dcount, dflag);
    else
        hipLaunchKernelGGL(k_descent_small<false>, dim3((unsigned)nruns), dim3(POTRF_THREADS), 0, ctx->stream, (const double*)dX,
                           (int)PB, (const double*)dy, (int)n, (int)d, (const double*)dW0, sigmaY, (int)kernel, alpha, beta, (int)norm,
                           (int)steprule, (int)max_its, dhw, dhc, dcount, dflag);
    HIPCHK(hipGetLastError());
    // the counts first: only the written part of the histories is copied back
    HIPCHK(hipMemcpyAsync(h + n_in + n_hw + n_hc, dcount, n_ints * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const int* hcount = reinterpret_cast<const int*>(h + n_in + n_hw + n_hc);
    int maxc = 0;
    for (int r = 0; r < nruns; ++r) {
        count_out[r] = hcount[r];
        flag_out[r] = hcount[nruns + r];
        maxc = std::max(maxc, hcount[r]);
    }
    if (maxc == (int)H || nruns == 1) {
        HIPCHK(hipMemcpyAsync(h + n_in, dhw, (n_hw + n_hc) * 8, hipMemcpyDeviceToHost, ctx->stream));
    } else {
        for (int r = 0; r < nruns; ++r) {
            HIPCHK(hipMemcpyAsync(h + n_in + (size_t)r * H * d, dhw + (size_t)r * H * d, (size_t)hcount[r] * d * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipMemcpyAsync(h + n_in + n_hw + (size_t)r * H, dhc + (size_t)r * H, (size_t)hcount[r] * 8, hipMemcpyDeviceToHost, ctx->stream));
        }
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int r = 0; r < nruns; ++r) {
        memcpy(w_hist_out + (size_t)r * H * d, h + n_in + (size_t)r * H * d, (size_t)count_out[r] * d * 8);
        memcpy(c_hist_out + (size_t)r * H, h + n_in + n_hw + (size_t)r * H, (size_t)count_out[r] * 8);
    }
    return GPBO_OK;
}

int gpbo_set_timing(gpbo_ctx* ctx, int32_t enabled) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    ctx->timing_off = (enabled == 0);
    return GPBO_OK;
}

int gpbo_last_timings(gpbo_ctx* ctx, gpbo_timings* out) {
    if (!ctx || !out) return GPBO_ERR_BAD_ARG;
    *out = ctx->tm;
    return GPBO_OK;
}

int gpbo_state_describe(gpbo_ctx* ctx, gpbo_state_desc* out) {
    if (!ctx || !out) return GPBO_ERR_BAD_ARG;
    if (!ctx->bufW || ctx->n <= 0) BADARG("state_describe: no model buffers");
    memset(out, 0, sizeof(*out));
    out->n = ctx->n;
    out->n_pad = ctx->n_pad;
    out->d = ctx->d;
    out->kernel = ctx->kernel;
    out->variant = ctx->variant;
    out->sigmaY = ctx->sigmaY;
    out->W = ctx->bufW;
    out->W_bytes = (int64_t)ctx->n_pad * ctx->n_pad * 8;
    out->alpha = ctx->alpha;
    out->alpha_bytes = ctx->n_pad * 8;
    out->Xts = ctx->Xts;
    out->Xts_bytes = (int64_t)MAX_D * ctx->n_pad * 8;
    out->params = ctx->params;
    out->params_bytes = PARAMS_DOUBLES * 8;
    out->y_max = ctx->y_max;
    return GPBO_OK;
}

int gpbo_state_alloc(gpbo_ctx* ctx, int64_t n, int64_t n_pad_in, int32_t d, int32_t kernel, int32_t variant, double sigmaY) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    if (n <= 0 || d <= 0 || d > MAX_D) BADARG("state_alloc: bad shape");
    if (n_pad_in != 0 && (n_pad_in < n || n_pad_in % PB != 0)) BADARG("state_alloc: n_pad must be 0 or a multiple of 128 that is >= n");
    HIPCHK(hipSetDevice(ctx->device));
    const int64_t n_pad = n_pad_in ? n_pad_in : round_up(n, PB);   // the sender's padding (it may hold rows reserved for appends)
    if (n_pad != ctx->cap_pad) CHK(ensure_model(ctx, n_pad, d, false));
    if (d != ctx->d) ctx->xs_rows = 0;
    ctx->m = ctx->xs_rows;
    ctx->fitted = false;
    ctx->have_posterior = false;
    ctx->have_acq = false;
    ctx->have_factor = false;
    ctx->n = n;
    ctx->n_pad = n_pad;
    ctx->nb = n_pad / PB;
    ctx->d = d;
    ctx->kernel = kernel;
    ctx->variant = variant;
    ctx->sigmaY = sigmaY;
    ctx->eps = (sigmaY == 0) ? (variant == GPBO_VARIANT_M3 ? 2.220446049250313e-16 : 1e-15) * (double)n : sigmaY * sigmaY;
    return GPBO_OK;
}

int gpbo_state_commit(gpbo_ctx* ctx, double y_max) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    if (!ctx->bufW || ctx->n <= 0) BADARG("state_commit: call gpbo_state_alloc first");
    HIPCHK(hipSetDevice(ctx->device));
    double par_out[8];
    HIPCHK(hipMemcpy(par_out, ctx->params, sizeof(par_out), hipMemcpyDeviceToHost));
    ctx->theta1 = par_out[P_THETA1];
    ctx->nll = par_out[P_NLL];
    ctx->y_max = y_max;
    ctx->fitted = true;
    return GPBO_OK;
}

int gpbo_append(gpbo_ctx* ctx, const double* x_new, double y_new, double* mu_max_out) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    return append_impl(ctx, x_new, y_new, mu_max_out);
}

int gpbo_append_acquire(gpbo_ctx* ctx, const double* x_new, double y_new, int32_t acq, double kappa, int32_t incumbent_mode,
                        double incumbent, double* mu_max_out, int64_t* argmax_out, double* best_out) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    return append_acquire_impl(ctx, x_new, y_new, acq, kappa, incumbent_mode, incumbent, mu_max_out, argmax_out, best_out);
}

// gpr_poly on the device (see gpbo_kriging_fit).  mle_epsilon != nullptr: also the quantity gpr_mle returns
// (Gaussian_Process_Regression.ipynb:255-263) into *mle_out, which needs a second factorisation, of K + epsilon I.
static int kriging_fit_impl(gpbo_ctx* ctx, const double* X, const double* y, int64_t n, int32_t d, double theta, int32_t deg,
                            double* beta_out, int32_t* nbeta_out, double* cov_out, const double* mle_epsilon, double* mle_out) {
    if (!X || !y) BADARG("kriging_fit: null input");
    if (n <= 0 || d <= 0 || d > 3) BADARG("kriging_fit: the trend basis exists for 1..3 input columns");
    if (deg < 0) BADARG("kriging_fit: negative degree");
    if (!(theta != 0.0) || !std::isfinite(theta)) BADARG("kriging_fit: theta must be finite and non-zero");
    if (mle_epsilon && !(*mle_epsilon >= 0.0)) BADARG("kriging_lml: epsilon must be >= 0");
    // number of trend columns, as polymat builds them
    int p = 1;
    if (deg >= 1) p = 1 + d;
    if (deg >= 2) p = 1 + 2 * d;
    if (deg >= 3 && d == 2) p = 6;
    if (deg >= 3 && d == 3) p = 10;
    if (n < p) BADARG("kriging_fit: fewer training points than trend columns");
    HIPCHK(hipSetDevice(ctx->device));
    const int64_t n_pad = round_up(n, PB);
    CHK(ensure_model(ctx, n_pad, d, true));
    if (d != ctx->d) ctx->xs_rows = 0;
    ctx->m = ctx->xs_rows;
    ctx->fitted = false;
    ctx->have_posterior = false;
    ctx->have_acq = false;
    ctx->inc_valid = false;
    ctx->n = n;
    ctx->n_pad = n_pad;
    ctx->nb = n_pad / PB;
    ctx->d = d;
    ctx->kernel = GPBO_KERNEL_RBF;
    ctx->variant = GPBO_VARIANT_KRIGING;
    ctx->sigmaY = 0;
    ctx->eps = 0.0;                          // gpr_poly uses K as built, no jitter (:199-202)
    ctx->theta.assign(d, theta);
    ctx->info = 0;
    const int q = p + 1;                     // right-hand sides: the p trend columns and y
    std::vector<double> xts((size_t)d * n_pad, 0.0), par(PARAMS_DOUBLES, 0.0), rhs((size_t)q * n_pad, 0.0);
    for (int t = 0; t < d; ++t) {
        par[P_INVTHETA + t] = 1.0 / theta;
        for (int64_t i = 0; i < n; ++i) xts[(size_t)t * n_pad + i] = X[i * d + t] / theta;
    }
    auto trend_col = [&](int64_t i, int c) -> double {
        const double* x = X + i * d;
        if (c == 0) return 1.0;
        if (c <= d) return x[c - 1];
        if (c <= 2 * d) return x[c - 1 - d] * x[c - 1 - d];
        if (d == 2) return x[0] * x[1];
        const int pair = c - 1 - 2 * d;     // d == 3: xy, xz, yz
        return pair == 0 ? x[0] * x[1] : (pair == 1 ? x[0] * x[2] : x[1] * x[2]);
    };
    for (int c = 0; c < p; ++c)
        for (int64_t i = 0; i < n; ++i) rhs[(size_t)c * n_pad + i] = trend_col(i, c);
    for (int64_t i = 0; i < n; ++i) rhs[(size_t)p * n_pad + i] = y[i];
    double *rhs_dev = nullptr, *u_dev = nullptr, *x_dev = nullptr;
    struct Guard {      // the three work vectors are released on every way out
        double *&a, *&b, *&c;
        ~Guard() {
            for (double* ptr : {a, b, c})
                if (ptr) hipFree(ptr);
        }
    } guard{rhs_dev, u_dev, x_dev};
    HIPCHK(hipMalloc((void**)&rhs_dev, rhs.size() * 8));
    HIPCHK(hipMalloc((void**)&u_dev, rhs.size() * 8));
    HIPCHK(hipMalloc((void**)&x_dev, rhs.size() * 8));
    HIPCHK(hipMemcpyAsync(ctx->Xts, xts.data(), xts.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->params, par.data(), par.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(rhs_dev, rhs.data(), rhs.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->info_dev, 0, sizeof(int), ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    timing_begin(ctx);
    int info = 0;
    bool use_lu = false;
    std::vector<double> lu_sol;
    {
        Scope tot(ctx, T_TOTAL);
        CHK(launch_kernel_matrix(ctx, ctx->bufA, nullptr, 0.0));
        CHK(launch_cholesky(ctx, ctx->bufA, ctx->info_dev));
        HIPCHK(hipMemcpyAsync(&info, ctx->info_dev, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (info == 0) {
            CHK(launch_inverse(ctx, ctx->bufA, ctx->bufW, ctx->bufB));
            hipLaunchKernelGGL(k_tri_gemv_n, dim3((unsigned)((n + 3) / 4), (unsigned)q), dim3(256), 0, ctx->stream,
                               (const double*)ctx->bufW, (long)n_pad, (int)n, (const double*)rhs_dev, u_dev, (int)n_pad);
            hipLaunchKernelGGL(k_tri_gemv_t, dim3((unsigned)((n + 255) / 256), (unsigned)q), dim3(256), 0, ctx->stream,
                               (const double*)ctx->bufW, (long)n_pad, (int)n, (const double*)u_dev, x_dev, (int)n_pad);
        } else if (n_pad <= KRIGING_LU_MAX) {
            // K without jitter is not numerically positive definite: the reference's np.linalg.solve (LU with partial
            // pivoting) still returns numbers there.  Same route: P K = L U in bufW, every solve by substitution with the factors.
            use_lu = true;
            ctx->variant = GPBO_VARIANT_KRIGING_LU;
            if (ctx->ipiv_cap < n_pad) {
                if (ctx->ipiv) hipFree(ctx->ipiv);
                ctx->ipiv = nullptr;
                ctx->ipiv_cap = 0;
                HIPCHK(hipMalloc((void**)&ctx->ipiv, (size_t)n_pad * sizeof(int)));
                ctx->ipiv_cap = n_pad;
            }
            HIPCHK(hipMemsetAsync(ctx->info_dev, 0, sizeof(int), ctx->stream));
            double* LU = ctx->bufW;          // the matrix the sweep reads (and a multi-context handle hands over) is bufW on every route
            CHK(launch_kernel_matrix(ctx, LU, nullptr, 0.0));
            hipLaunchKernelGGL(k_mirror_upper, dim3((unsigned)(n_pad / 64), (unsigned)(n_pad / 4)), dim3(256), 0, ctx->stream, LU, (int)n_pad);
            const int nb = (int)ctx->nb;
            for (int pp = 0; pp < nb; ++pp) {
                const int c0 = pp * PB;
                hipLaunchKernelGGL(k_lu_panel, dim3(1), dim3(1024), 0, ctx->stream, LU, (long)n_pad, (int)n_pad, c0, ctx->ipiv, ctx->info_dev);
                if (nb > 1)
                    hipLaunchKernelGGL(k_lu_swap_rows, dim3((unsigned)((n_pad - PB + 255) / 256)), dim3(256), 0, ctx->stream, LU,
                                       (long)n_pad, (int)n_pad, c0, (const int*)ctx->ipiv);
                if (pp + 1 < nb) {
                    hipLaunchKernelGGL(k_lu_trsm_u, dim3((unsigned)((n_pad - c0 - PB + 255) / 256)), dim3(256), 0, ctx->stream, LU,
                                       (long)n_pad, (int)n_pad, c0);
                    hipLaunchKernelGGL(k_lu_gemm_update, dim3((unsigned)(nb - pp - 1), (unsigned)(nb - pp - 1)), dim3(GEMM_THREADS), 0,
                                       ctx->stream, LU, (long)n_pad, pp);
                }
            }
            // the q right-hand sides as one k-major block of 128 columns (u_dev -> x_dev would be column-major: stage through
            // two small matrices of the sweep's layout)
            double *Rk = nullptr, *Zk = nullptr;
            struct G2 {
                double *&a, *&b;
                ~G2() {
                    if (a) hipFree(a);
                    if (b) hipFree(b);
                }
            } g2{Rk, Zk};
            HIPCHK(hipMalloc((void**)&Rk, (size_t)n_pad * TN * 8));
            HIPCHK(hipMalloc((void**)&Zk, (size_t)n_pad * TN * 8));
            std::vector<double> rk((size_t)n_pad * TN, 0.0);
            for (int c = 0; c < q; ++c)
                for (int64_t i = 0; i < n; ++i) rk[(size_t)i * TN + c] = rhs[(size_t)c * n_pad + i];
            HIPCHK(hipMemcpyAsync(Rk, rk.data(), rk.size() * 8, hipMemcpyHostToDevice, ctx->stream));
            CHK(launch_lu_solve(ctx, LU, ctx->ipiv, Rk, Zk, (long)TN, TN));
            HIPCHK(hipMemcpyAsync(rk.data(), Zk, rk.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipMemcpyAsync(&info, ctx->info_dev, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            lu_sol.assign((size_t)q * n_pad, 0.0);
            for (int c = 0; c < q; ++c)
                for (int64_t i = 0; i < n; ++i) lu_sol[(size_t)c * n_pad + i] = rk[(size_t)i * TN + c];
            HIPCHK(hipGetLastError());
        }
    }
    HIPCHK(hipGetLastError());
    std::vector<double> sol((size_t)q * n_pad);
    if (use_lu) sol = lu_sol;
    else if (info == 0) HIPCHK(hipMemcpyAsync(sol.data(), x_dev, sol.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    double ms[T_NCAT], cnt[T_NCAT];
    timing_collect(ctx, ms, cnt);
    ctx->tm.fit_total_ms = ms[T_TOTAL];
    ctx->tm.kbuild_ms = ms[T_KBUILD];
    ctx->tm.chol_ms = ms[T_CHOL];
    ctx->tm.chol_count = 1;
    ctx->tm.inverse_ms = ms[T_INV];
    if (info != 0) {
        ctx->info = info;
        ctx->err = use_lu ? "kriging_fit: K is singular (an exactly zero pivot column in the LU factorisation)"
                          : "kriging_fit: K is not positive definite (no jitter is added in this flavour) and too large for the LU fallback";
        return GPBO_ERR_NOT_PD;
    }
    // GLS trend on the host (p <= 10): beta = (F^T K^-1 F)^-1 F^T K^-1 y, :201
    std::vector<double> A((size_t)p * p), b(p), beta(p);
    for (int r = 0; r < p; ++r) {
        for (int c = 0; c < p; ++c) {
            double s2 = 0;
            for (int64_t i = 0; i < n; ++i) s2 += rhs[(size_t)r * n_pad + i] * sol[(size_t)c * n_pad + i];
            A[(size_t)r * p + c] = s2;
        }
        double s3 = 0;
        for (int64_t i = 0; i < n; ++i) s3 += rhs[(size_t)r * n_pad + i] * sol[(size_t)p * n_pad + i];
        b[r] = s3;
    }
    {   // Gaussian elimination with partial pivoting (what np.linalg.solve does for this p x p system)
        for (int c = 0; c < p; ++c) {
            int best = c;
            for (int r = c + 1; r < p; ++r)
                if (std::fabs(A[(size_t)r * p + c]) > std::fabs(A[(size_t)best * p + c])) best = r;
            if (A[(size_t)best * p + c] == 0.0) {
                ctx->err = "kriging_fit: singular trend system F^T K^-1 F";
                return GPBO_ERR_NOT_PD;
            }
            if (best != c) {
                for (int k2 = 0; k2 < p; ++k2) std::swap(A[(size_t)c * p + k2], A[(size_t)best * p + k2]);
                std::swap(b[c], b[best]);
            }
            for (int r = c + 1; r < p; ++r) {
                const double f = A[(size_t)r * p + c] / A[(size_t)c * p + c];
                for (int k2 = c; k2 < p; ++k2) A[(size_t)r * p + k2] -= f * A[(size_t)c * p + k2];
                b[r] -= f * b[c];
            }
        }
        for (int r = p - 1; r >= 0; --r) {
            double s2 = b[r];
            for (int c = r + 1; c < p; ++c) s2 -= A[(size_t)r * p + c] * beta[c];
            beta[r] = s2 / A[(size_t)r * p + r];
        }
    }
    // a = K^-1 (y - F beta), cov = (y - F beta)^T a / n  (:202)
    std::vector<double> a(n_pad, 0.0);
    double cov = 0, rr = 0;
    for (int64_t i = 0; i < n; ++i) {
        double ai = sol[(size_t)p * n_pad + i], ri = y[i];
        for (int c = 0; c < p; ++c) {
            ai -= sol[(size_t)c * n_pad + i] * beta[c];
            ri -= rhs[(size_t)c * n_pad + i] * beta[c];
        }
        a[i] = ai;
        cov += ri * ai;
        rr += ri * ri;
    }
    cov /= (double)n;
    par[P_THETA1] = cov;
    par[P_CB] = 1.0;
    par[P_C0] = cov * cov;                   // mse = cov**2 * (1 - r^T K^-1 r), :233
    par[P_C1] = cov * cov;
    par[P_CLAMP] = 0.0;                      // y_kriging_v2 does not clamp
    par[P_NBETA] = p;
    par[P_DEG] = deg;
    double ymax = y[0];
    for (int64_t i = 1; i < n; ++i) ymax = std::max(ymax, y[i]);
    par[P_YMAX] = ymax;
    for (int c = 0; c < p; ++c) par[P_BETA + c] = beta[c];
    HIPCHK(hipMemcpy(ctx->params, par.data(), par.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->alpha, a.data(), a.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->alpha1, a.data(), a.size() * 8, hipMemcpyHostToDevice));
    ctx->theta1 = cov;
    ctx->nll = NAN;
    ctx->y_max = ymax;
    ctx->bufL = ctx->bufA;
    ctx->fitted = true;
    ctx->have_factor = !use_lu;              // the LU path has no Cholesky factor to hand out
    ctx->reserve_left = 0;
    if (beta_out) memcpy(beta_out, beta.data(), p * sizeof(double));
    if (nbeta_out) *nbeta_out = p;
    if (cov_out) *cov_out = cov;
    if (mle_epsilon) {
        // gpr_mle (:255-263): 1/2 log det chol(K + epsilon I) + r^T r / (2 cov^2) + N/2 log(2 pi cov^2); the determinant of the
        // triangular factor as the sum of the logs of its diagonal (the reference multiplies the diagonal out, which
        // underflows for a few hundred points).  K + epsilon I is factored in bufB; a pivot <= 0 is the LinAlgError
        // np.linalg.cholesky raises there.
        HIPCHK(hipMemsetAsync(ctx->info_dev, 0, sizeof(int), ctx->stream));
        CHK(launch_kernel_matrix(ctx, ctx->bufB, nullptr, *mle_epsilon));
        CHK(launch_cholesky(ctx, ctx->bufB, ctx->info_dev));
        hipLaunchKernelGGL(k_sumlog_diag, dim3(1), dim3(256), 0, ctx->stream, (const double*)ctx->bufB, (long)n_pad, (int)n, ctx->red_v);
        HIPCHK(hipGetLastError());
        double sumlog = 0;
        int info2 = 0;
        HIPCHK(hipMemcpyAsync(&sumlog, ctx->red_v, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(&info2, ctx->info_dev, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (info2 != 0) {
            ctx->info = info2;
            ctx->err = "kriging_lml: K + epsilon I is not positive definite";
            return GPBO_ERR_NOT_PD;
        }
        if (mle_out) *mle_out = 0.5 * sumlog + rr / (2.0 * cov * cov) + 0.5 * (double)n * std::log(2.0 * M_PI * cov * cov);
    }
    return GPBO_OK;
}

int gpbo_kriging_fit(gpbo_ctx* ctx, const double* X, const double* y, int64_t n, int32_t d, double theta, int32_t deg,
                     double* beta_out, int32_t* nbeta_out, double* cov_out) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    return kriging_fit_impl(ctx, X, y, n, d, theta, deg, beta_out, nbeta_out, cov_out, nullptr, nullptr);
}

int gpbo_kriging_cross_scale(gpbo_ctx* ctx, const double* X, double theta_r) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    if (!X) BADARG("kriging_cross_scale: null input");
    if (!ctx->fitted || (ctx->variant != GPBO_VARIANT_KRIGING && ctx->variant != GPBO_VARIANT_KRIGING_LU))
        BADARG("kriging_cross_scale: call gpbo_kriging_fit first");
    if (!(theta_r != 0.0) || !std::isfinite(theta_r)) BADARG("kriging_cross_scale: theta_r must be finite and non-zero");
    HIPCHK(hipSetDevice(ctx->device));
    const int d = ctx->d;
    const int64_t n = ctx->n, n_pad = ctx->n_pad;
    std::vector<double> xts((size_t)d * n_pad, 0.0), it((size_t)d, 1.0 / theta_r);
    for (int t = 0; t < d; ++t)
        for (int64_t i = 0; i < n; ++i) xts[(size_t)t * n_pad + i] = X[i * d + t] / theta_r;
    HIPCHK(hipMemcpyAsync(ctx->Xts, xts.data(), xts.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->params + P_INVTHETA, it.data(), it.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->theta.assign((size_t)d, theta_r);
    ctx->have_posterior = false;
    ctx->have_acq = false;
    return GPBO_OK;
}

int gpbo_kriging_lml(gpbo_ctx* ctx, const double* X, const double* y, int64_t n, int32_t d, const double* thetas, int32_t B,
                     int32_t deg, double epsilon, double* nll_out, int32_t* status_out) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    if (!thetas || !nll_out || B <= 0) BADARG("kriging_lml: need B >= 1 and non-null arrays");
    if (!ctx->kr_work && gpbo_create(ctx->device, &ctx->kr_work) != GPBO_OK) {
        ctx->err = "kriging_lml: could not create the work context";
        return GPBO_ERR_HIP;
    }
    gpbo_ctx* w = ctx->kr_work;      // private: the caller's model, candidates and posterior are untouched
    w->timing_off = true;
    for (int b = 0; b < B; ++b) {
        double v = NAN;
        const int st = kriging_fit_impl(w, X, y, n, d, thetas[b], deg, nullptr, nullptr, nullptr, &epsilon, &v);
        if (st == GPBO_ERR_BAD_ARG || st == GPBO_ERR_HIP) {
            ctx->err = w->err;
            return st;
        }
        nll_out[b] = (st == GPBO_OK) ? v : NAN;
        if (status_out) status_out[b] = st;
    }
    return GPBO_OK;
}

int gpbo_set_posterior(gpbo_ctx* ctx, const double* mu, const double* var, int64_t m) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    if (!mu || !var || m <= 0) BADARG("set_posterior: empty input");
    HIPCHK(hipSetDevice(ctx->device));
    CHK(ensure_outputs(ctx, m));
    ctx->m = m;
    ctx->inc_valid = false;
    HIPCHK(hipMemcpyAsync(ctx->mu, mu, (size_t)m * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->var, var, (size_t)m * 8, hipMemcpyHostToDevice, ctx->stream));
    ctx->var_stale = false;
    ctx->mu_argmax_valid = false;
    CHK(reduce_max(ctx, ctx->mu, m, ctx->inc_dev + 1));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->have_posterior = true;
    ctx->have_acq = false;
    return GPBO_OK;
}

int gpbo_kernel_matrix(gpbo_ctx* ctx, const double* U, int64_t p, const double* V, int64_t q, int32_t d, const double* theta,
                       int32_t kernel, double* K_out) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    if (!U || !V || !theta || !K_out || p <= 0 || q <= 0 || d <= 0 || d > MAX_D) BADARG("kernel_matrix: bad arguments");
    if (kernel != GPBO_KERNEL_RBF && kernel != GPBO_KERNEL_MATERN52) BADARG("kernel_matrix: unknown kernel");
    for (int t = 0; t < d; ++t)
        if (!(theta[t] != 0.0) || !std::isfinite(theta[t])) BADARG("kernel_matrix: theta entries must be finite and non-zero");
    HIPCHK(hipSetDevice(ctx->device));
    double *du = nullptr, *dv = nullptr, *dk = nullptr, *dth = nullptr;
    auto cleanup = [&]() {
        for (void* ptr : {(void*)du, (void*)dv, (void*)dk, (void*)dth})
            if (ptr) hipFree(ptr);
    };
    std::vector<double> ith(d);
    for (int t = 0; t < d; ++t) ith[t] = 1.0 / theta[t];
    if (hipMalloc((void**)&du, (size_t)p * d * 8) != hipSuccess || hipMalloc((void**)&dv, (size_t)q * d * 8) != hipSuccess ||
        hipMalloc((void**)&dk, (size_t)p * q * 8) != hipSuccess || hipMalloc((void**)&dth, (size_t)d * 8) != hipSuccess) {
        cleanup();
        ctx->err = "kernel_matrix: hipMalloc failed";
        return GPBO_ERR_HIP;
    }
    hipError_t up = hipMemcpyAsync(du, U, (size_t)p * d * 8, hipMemcpyHostToDevice, ctx->stream);
    if (up == hipSuccess) up = hipMemcpyAsync(dv, V, (size_t)q * d * 8, hipMemcpyHostToDevice, ctx->stream);
    if (up == hipSuccess) up = hipMemcpyAsync(dth, ith.data(), (size_t)d * 8, hipMemcpyHostToDevice, ctx->stream);
    if (up != hipSuccess) {
        hipStreamSynchronize(ctx->stream);
        cleanup();
        ctx->err = std::string("kernel_matrix: upload failed: ") + hipGetErrorString(up);
        return GPBO_ERR_HIP;
    }
    hipLaunchKernelGGL(k_cross_kernel, dim3((unsigned)((q + 63) / 64), (unsigned)((p + 3) / 4)), dim3(256), 0, ctx->stream,
                       (const double*)du, (long)p, (const double*)dv, (long)q, (int)d, (const double*)dth, kernel, dk);
    hipError_t e1 = hipMemcpyAsync(K_out, dk, (size_t)p * q * 8, hipMemcpyDeviceToHost, ctx->stream);
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    cleanup();
    if (e1 != hipSuccess || e2 != hipSuccess) {
        ctx->err = std::string("kernel_matrix: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2);
        return GPBO_ERR_HIP;
    }
    return GPBO_OK;
}

int gpbo_kernel_from_distances(gpbo_ctx* ctx, const double* Dabs, int32_t d, int64_t p, int64_t q, const double* theta,
                               int32_t kernel, double* K_out) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    if (!Dabs || !theta || !K_out || p <= 0 || q <= 0 || d <= 0 || d > MAX_D) BADARG("kernel_from_distances: bad arguments");
    if (kernel != GPBO_KERNEL_RBF && kernel != GPBO_KERNEL_MATERN52) BADARG("kernel_from_distances: unknown kernel");
    for (int t = 0; t < d; ++t)
        if (!(theta[t] != 0.0) || !std::isfinite(theta[t])) BADARG("kernel_from_distances: theta entries must be finite and non-zero");
    HIPCHK(hipSetDevice(ctx->device));
    const size_t pq = (size_t)p * q;
    double *dd = nullptr, *dk = nullptr, *dth = nullptr;
    auto cleanup = [&]() {
        for (void* ptr : {(void*)dd, (void*)dk, (void*)dth})
            if (ptr) hipFree(ptr);
    };
    std::vector<double> ith(d);
    for (int t = 0; t < d; ++t) ith[t] = 1.0 / theta[t];
    if (hipMalloc((void**)&dd, pq * d * 8) != hipSuccess || hipMalloc((void**)&dk, pq * 8) != hipSuccess ||
        hipMalloc((void**)&dth, (size_t)d * 8) != hipSuccess) {
        cleanup();
        ctx->err = "kernel_from_distances: hipMalloc failed";
        return GPBO_ERR_HIP;
    }
    hipError_t up = hipMemcpyAsync(dd, Dabs, pq * d * 8, hipMemcpyHostToDevice, ctx->stream);
    if (up == hipSuccess) up = hipMemcpyAsync(dth, ith.data(), (size_t)d * 8, hipMemcpyHostToDevice, ctx->stream);
    if (up != hipSuccess) {
        hipStreamSynchronize(ctx->stream);
        cleanup();
        ctx->err = std::string("kernel_from_distances: upload failed: ") + hipGetErrorString(up);
        return GPBO_ERR_HIP;
    }
    hipLaunchKernelGGL(k_kernel_from_absdiff, dim3((unsigned)((pq + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)dd,
                       (long)pq, (int)d, (const double*)dth, kernel, dk);
    hipError_t e1 = hipMemcpyAsync(K_out, dk, pq * 8, hipMemcpyDeviceToHost, ctx->stream);
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    cleanup();
    if (e1 != hipSuccess || e2 != hipSuccess) {
        ctx->err = std::string("kernel_from_distances: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2);
        return GPBO_ERR_HIP;
    }
    return GPBO_OK;
}

int gpbo_reserve_appends(gpbo_ctx* ctx, int64_t max_appends) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    if (max_appends < 0) BADARG("reserve_appends: negative");
    ctx->reserve = max_appends;
    return GPBO_OK;
}

int gpbo_lml_batch(gpbo_ctx* ctx, const double* X, const double* y, int64_t n, int32_t d, const double* thetas,
                   const double* sigmas, int32_t B, int32_t kernel, double* nll_out, double* grad_out, int32_t* status_out) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    if (!X || !y || !thetas || !sigmas || !nll_out) BADARG("lml_batch: null input");
    if (n <= 0 || d <= 0 || d > MAX_D || B <= 0) BADARG("lml_batch: need n >= 1, 1 <= d <= 32, B >= 1");
    if (kernel != GPBO_KERNEL_RBF && kernel != GPBO_KERNEL_MATERN52) BADARG("lml_batch: unknown kernel");
    if (grad_out && d > GRAD_D) BADARG("lml_batch: gradient supports d <= 16");
    for (int64_t i = 0; i < (int64_t)B * d; ++i)
        if (!(thetas[i] != 0.0) || !std::isfinite(thetas[i])) BADARG("lml_batch: theta entries must be finite and non-zero");
    for (int b = 0; b < B; ++b)
        if (!(sigmas[b] >= 0)) BADARG("lml_batch: sigmas must be >= 0");
    HIPCHK(hipSetDevice(ctx->device));
    const int64_t n_pad = round_up(n, PB);
    const int nb = (int)(n_pad / PB);
    const int ntiles = nb * nb;

    // Up to 128 points: one workgroup evaluates one setting from start to end (k_lml_small), one launch for all B.
    bool small = (n_pad == PB && d <= GRAD_D);
    if (const char* e = getenv("GPBO_LML_SMALL")) small = small && atoi(e) != 0;
    if (small) {
        // one persistent device workspace, one upload, one launch, one download:
        //   in : [Xraw d*128][y 128][thetas B*d][sigmas B]      out: [nll B][grad B*d][info B (as doubles)]
        const size_t n_in = (size_t)d * PB + PB + (size_t)B * d + B;
        const size_t n_out = (size_t)B + (size_t)B * d + B;
        if (ctx->sm_cap < n_in + n_out) {
            if (ctx->sm_buf) hipFree(ctx->sm_buf);
            if (ctx->sm_host) hipHostFree(ctx->sm_host);
            ctx->sm_buf = nullptr;
            ctx->sm_host = nullptr;
            ctx->sm_cap = 0;
            const size_t cap = 2 * (n_in + n_out);         // room to grow: a multi-start fit calls again and again
            HIPCHK(hipMalloc((void**)&ctx->sm_buf, cap * 8));
            HIPCHK(hipHostMalloc((void**)&ctx->sm_host, cap * 8, hipHostMallocDefault));
            ctx->sm_cap = cap;
        }
        double* h = ctx->sm_host;
        std::fill(h, h + (size_t)d * PB + PB, 0.0);
        for (int t = 0; t < d; ++t)
            for (int64_t i = 0; i < n; ++i) h[(size_t)t * PB + i] = X[i * d + t];
        double* hy = h + (size_t)d * PB;
        for (int64_t i = 0; i < n; ++i) hy[i] = y[i];
        memcpy(hy + PB, thetas, (size_t)B * d * 8);
        memcpy(hy + PB + (size_t)B * d, sigmas, (size_t)B * 8);
        double* dX = ctx->sm_buf;
        double* dy = dX + (size_t)d * PB;
        double* dth = dy + PB;
        double* dsg = dth + (size_t)B * d;
        double* dnll = ctx->sm_buf + n_in;
        double* dgrad = dnll + B;
        int* dinfo = reinterpret_cast<int*>(dgrad + (size_t)B * d);
        const auto t0 = std::chrono::steady_clock::now();
        HIPCHK(hipMemcpyAsync(ctx->sm_buf, h, n_in * 8, hipMemcpyHostToDevice, ctx->stream));
        if (n > small_wide_min())
            hipLaunchKernelGGL(k_lml_small<true>, dim3((unsigned)B), dim3(SMALL_WIDE_THREADS), 0, ctx->stream, (const double*)dX, (int)PB,
                               (const double*)dy, (int)n, (int)d, (const double*)dth, (const double*)dsg, (int)kernel, grad_out ? 1 : 0,
                               dnll, dgrad, dinfo);
        else
            hipLaunchKernelGGL(k_lml_small<false>, dim3((unsigned)B), dim3(POTRF_THREADS), 0, ctx->stream, (const double*)dX, (int)PB,
                               (const double*)dy, (int)n, (int)d, (const double*)dth, (const double*)dsg, (int)kernel, grad_out ? 1 : 0,
                               dnll, dgrad, dinfo);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h + n_in, dnll, n_out * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        const double* onll = h + n_in;
        const double* ograd = onll + B;
        const int* hinfo = reinterpret_cast<const int*>(ograd + (size_t)B * d);
        ctx->tm.fit_total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        ctx->tm.chol_count = B;
        ctx->tm.kbuild_ms = ctx->tm.chol_ms = ctx->tm.trsv_ms = ctx->tm.inverse_ms = 0;
        for (int b = 0; b < B; ++b) {
            const bool bad = hinfo[b] != 0 || !std::isfinite(onll[b]);
            if (status_out) status_out[b] = bad ? GPBO_ERR_NOT_PD : GPBO_OK;
            nll_out[b] = bad ? NAN : onll[b];
            if (grad_out)
                for (int t = 0; t < d; ++t) grad_out[(size_t)b * d + t] = bad ? NAN : ograd[(size_t)b * d + t];
        }
        return GPBO_OK;
    }

    // The settings are independent.  Each lane = a context of its own (work matrices + stream); settings are dealt
    // round-robin over the lanes so that the latency-bound parts of one factorisation (diagonal blocks, panel
    // solves, triangular solves) overlap the MFMA-bound updates of the others.  Every lane is a private work context:
    // the caller's own model, candidates and posterior are never touched by a likelihood sweep.
    // Eight lanes measure 5-6 % faster than four at n = 16384; a lane holds three n_pad^2 matrices, so large problems
    // take as many as fit in half of the free memory.
    int lanes = 8;
    {
        size_t free_b = 0, total_b = 0;
        const double per_lane = 3.0 * 8.0 * (double)n_pad * (double)n_pad;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            lanes = (int)std::max(1.0, std::min(8.0, std::floor(0.5 * (double)free_b / per_lane)));
    }
    if (const char* e = getenv("GPBO_LML_STREAMS")) lanes = atoi(e);
    lanes = std::max(1, std::min(lanes, std::min<int>(B, 8)));
    // Up to 16 panels a setting is ~30-150 launches of a few microseconds of GPU work each and the issuing host thread
    // is the bottleneck (~6 us per launch; issuing from one thread per lane does not help, the launches serialise in
    // the runtime): the sequence of one setting is captured once per lane into a graph and replayed.
    bool use_graph = nb <= 16;
    if (const char* e = getenv("GPBO_LML_GRAPH")) use_graph = atoi(e) != 0;
    // At least two settings: ONE batched work context runs up to 64 settings (8 from n = 8320 up) in lockstep, every
    // kernel of the sequence covering all of them (slot = blockIdx.z).  The chain of small launches that bounds a
    // single factorisation at these sizes is then paid once per round instead of once per setting; streams mapped on
    // a handful of hardware queues overlap only 2-3 settings.  Measured against eight stream lanes: 10-12x less time
    // per setting at n = 200-1000, 5x at 2000, 2.3x at 4096, 1.4x at 8192 (16 slots), 1.13x at 16384 (8 slots).
    bool lockstep = B >= 2;
    if (const char* e = getenv("GPBO_LML_LOCKSTEP")) lockstep = atoi(e) != 0 && B >= 2;
    int slots = 1;
    if (lockstep) {
        size_t free_b = 0, total_b = 0;
        const double per_slot = 3.2 * 8.0 * (double)n_pad * (double)n_pad;
        slots = std::min<int>(B, nb <= 16 ? 64 : (nb <= 32 ? 32 : (nb <= 128 ? 16 : 8)));      // n = 16384: 16 slots 13.8 settings/s, 8: 13.6, 32: 13.0
        if (const char* e = getenv("GPBO_LML_SLOTS")) slots = std::min<int>(B, std::max(1, atoi(e)));
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            slots = (int)std::max(1.0, std::min((double)slots, std::floor(0.4 * (double)free_b / per_slot)));
        if (slots < 2) lockstep = false;
    }
    if (lockstep) {
        lanes = 1;
        use_graph = false;
    }
    std::vector<gpbo_ctx*> lane(lanes, nullptr);
    if (lockstep) {
        if (!ctx->lml_lock && gpbo_create(ctx->device, &ctx->lml_lock) != GPBO_OK) {
            ctx->err = "lml_batch: could not create the batched work context";
            return GPBO_ERR_HIP;
        }
        lane[0] = ctx->lml_lock;
        lane[0]->batch = slots;
        lane[0]->bs.mat = (long)n_pad * n_pad;
        lane[0]->bs.dinv = (long)n_pad * PB;
        lane[0]->bs.xts = (long)MAX_D * n_pad;
        lane[0]->bs.vec = (long)n_pad;
        lane[0]->bs.par = PARAMS_DOUBLES;
        lane[0]->bs.gp = (long)ntiles * GRAD_D;
        lane[0]->bs.slot = (long)(sizeof(LmlSlot) / sizeof(double));
    }
    if (!lockstep)
        for (int l = 0; l < lanes && l < (int)ctx->lml_lanes.size(); ++l) lane[l] = ctx->lml_lanes[l];
    auto fail_hip = [&](const char* what, hipError_t e) {
        ctx->err = std::string(what) + ": " + hipGetErrorString(e);
        return GPBO_ERR_HIP;
    };
#define LCHK(expr)                                         \
    do {                                                   \
        hipError_t e_ = (expr);                            \
        if (e_ != hipSuccess) return fail_hip(#expr, e_);  \
    } while (0)
    // batch arrays on the caller context, grown on demand (their addresses are part of the lanes' graph keys)
    const size_t need_x = (size_t)(d + 1) * n_pad, need_b = (size_t)B, need_bd = (size_t)B * d;
    if (ctx->lb_xraw_cap < need_x) {
        if (ctx->lb_xraw) hipFree(ctx->lb_xraw);
        ctx->lb_xraw = nullptr;
        ctx->lb_xraw_cap = 0;
        LCHK(hipMalloc((void**)&ctx->lb_xraw, need_x * 8));
        ctx->lb_xraw_cap = need_x;
    }
    if (ctx->lb_b_cap < need_b) {
        for (void* p : {(void*)ctx->lb_sg, (void*)ctx->lb_nll, (void*)ctx->lb_info})
            if (p) hipFree(p);
        ctx->lb_sg = ctx->lb_nll = nullptr;
        ctx->lb_info = nullptr;
        ctx->lb_b_cap = 0;
        const size_t cap = std::max<size_t>(64, need_b * 2);
        LCHK(hipMalloc((void**)&ctx->lb_sg, cap * 8));
        LCHK(hipMalloc((void**)&ctx->lb_nll, cap * 8));
        LCHK(hipMalloc((void**)&ctx->lb_info, cap * sizeof(int)));
        ctx->lb_b_cap = cap;
    }
    if (ctx->lb_bd_cap < need_bd) {
        for (void* p : {(void*)ctx->lb_th, (void*)ctx->lb_grad})
            if (p) hipFree(p);
        ctx->lb_th = ctx->lb_grad = nullptr;
        ctx->lb_bd_cap = 0;
        const size_t cap = std::max<size_t>(64 * MAX_D, need_bd * 2);
        LCHK(hipMalloc((void**)&ctx->lb_th, cap * 8));
        LCHK(hipMalloc((void**)&ctx->lb_grad, cap * 8));
        ctx->lb_bd_cap = cap;
    }
    for (int l = 0; l < lanes; ++l) {
        if (!lane[l]) {
            if (gpbo_create(ctx->device, &lane[l]) != GPBO_OK) {
                ctx->err = "lml_batch: could not create a work context";
                        return GPBO_ERR_HIP;
            }
            ctx->lml_lanes.push_back(lane[l]);
        }
        gpbo_ctx* c = lane[l];
        int st = ensure_model(c, n_pad, d, true);
        if (st != GPBO_OK) {
            ctx->err = c->err;
            return st;
        }
        c->fitted = false;                           // the model buffers are work space here
        c->have_posterior = false;
        c->have_acq = false;
        c->have_factor = false;
        c->inc_valid = false;
        c->n = n;
        c->n_pad = n_pad;
        c->nb = nb;
        c->d = d;
        c->kernel = kernel;
        c->variant = GPBO_VARIANT_SINGLE;
        const size_t S = (size_t)c->batch;
        if (c->slot_cap < S) {
            if (c->lane_slot) hipFree(c->lane_slot);
            c->lane_slot = nullptr;
            c->slot_cap = 0;
            LCHK(hipMalloc((void**)&c->lane_slot, S * sizeof(LmlSlot)));
            c->slot_cap = S;
            c->alloc_gen += 1;
        }
        if (c->params_cap < S * PARAMS_DOUBLES) {
            if (c->params) hipFree(c->params);
            c->params = nullptr;
            c->params_cap = 0;
            LCHK(hipMalloc((void**)&c->params, S * PARAMS_DOUBLES * sizeof(double)));
            c->params_cap = S * PARAMS_DOUBLES;
            c->alloc_gen += 1;
        }
        if (grad_out && c->lane_gpart_cap < (size_t)ntiles * GRAD_D * S) {
            if (c->lane_gpart) hipFree(c->lane_gpart);
            c->lane_gpart = nullptr;
            c->lane_gpart_cap = 0;
            LCHK(hipMalloc((void**)&c->lane_gpart, (size_t)ntiles * GRAD_D * 8 * S));
            c->lane_gpart_cap = (size_t)ntiles * GRAD_D * S;
            c->alloc_gen += 1;
        }
    }
    {
        // one staged upload: the d padded coordinate rows, then the padded y as row d; the slots' copies of y are
        // made on the device
        std::vector<double> xr((size_t)(d + 1) * n_pad, 0.0);
        for (int t = 0; t < d; ++t)
            for (int64_t i = 0; i < n; ++i) xr[(size_t)t * n_pad + i] = X[i * d + t];
        for (int64_t i = 0; i < n; ++i) xr[(size_t)d * n_pad + i] = y[i];
        LCHK(hipMemcpy(ctx->lb_xraw, xr.data(), xr.size() * 8, hipMemcpyHostToDevice));
        LCHK(hipMemcpy(ctx->lb_th, thetas, (size_t)B * d * 8, hipMemcpyHostToDevice));
        LCHK(hipMemcpy(ctx->lb_sg, sigmas, (size_t)B * 8, hipMemcpyHostToDevice));
        LCHK(hipDeviceSynchronize());
        for (int l = 0; l < lanes; ++l)
            hipLaunchKernelGGL(k_replicate_vec, dim3((unsigned)((n_pad + 255) / 256), (unsigned)lane[l]->batch), dim3(256), 0,
                               lane[l]->stream, (const double*)(ctx->lb_xraw + (size_t)d * n_pad), lane[l]->yv, (int)n_pad);
        LCHK(hipGetLastError());
    }
    // the launch sequence of one setting per slot of context c; every argument is the same for every setting
    auto enqueue_setting = [&](gpbo_ctx* c) -> int {
        const unsigned S = (unsigned)c->batch;
        hipLaunchKernelGGL(k_setup_setting, dim3((unsigned)((n_pad + 255) / 256), (unsigned)d, S), dim3(256), 0, c->stream,
                           (const double*)ctx->lb_xraw, (const double*)c->lane_slot->theta, (int)d, (int)n_pad, c->Xts, c->params, c->bs);
        int st = launch_kernel_matrix(c, c->bufA, nullptr, 0.0, &c->lane_slot->eps);
        if (st == GPBO_OK) st = launch_cholesky(c, c->bufA, &c->lane_slot->info);
        if (st == GPBO_OK) st = launch_trsv(c, c->bufA, 0.0, &c->lane_slot->eps);
        if (st != GPBO_OK) return st;
        hipLaunchKernelGGL(k_fit_scalars, dim3(1, 1, S), dim3(256), 0, c->stream, (const double*)c->bufA, (long)n_pad,
                           (const double*)c->yv, (const double*)c->alpha1, c->alpha, (int)n, (int)n_pad,
                           (int)GPBO_VARIANT_SINGLE, c->params, c->bs);
        if (grad_out) {
            st = launch_inverse(c, c->bufA, c->bufW, c->bufB);
            if (st != GPBO_OK) return st;
            if (hipMemsetAsync(c->lane_gpart, 0, (size_t)ntiles * GRAD_D * 8 * S, c->stream) != hipSuccess) return GPBO_ERR_HIP;
            hipLaunchKernelGGL(k_lml_grad_tiles, dim3((unsigned)(nb * (nb + 1) / 2), 1, S), dim3(GEMM_THREADS), 0, c->stream,
                               (const double*)c->bufW, (long)n_pad, (const double*)c->Xts, (int)n, (int)n_pad, (int)d,
                               (const double*)c->alpha1, (const double*)c->params, c->lane_gpart, nb, (int)kernel, c->bs);
        }
        hipLaunchKernelGGL(k_lml_store, dim3(1, 1, S), dim3(256), 0, c->stream, (const double*)c->params, (const LmlSlot*)c->lane_slot,
                           (const double*)c->lane_gpart, ntiles, (int)d, grad_out ? 1 : 0, ctx->lb_nll, ctx->lb_grad, ctx->lb_info, c->bs);
        return GPBO_OK;
    };
    if (use_graph) {
        const std::vector<int64_t> key = {n, n_pad, d, kernel, grad_out ? 1 : 0, (int64_t)(intptr_t)ctx->lb_xraw,
                                          (int64_t)(intptr_t)ctx->lb_nll, (int64_t)(intptr_t)ctx->lb_grad,
                                          (int64_t)(intptr_t)ctx->lb_info, (int64_t)(intptr_t)ctx->lb_th,
                                          (int64_t)(intptr_t)ctx->lb_sg, 0};
        for (int l = 0; l < lanes; ++l) {
            gpbo_ctx* c = lane[l];
            std::vector<int64_t> k = key;
            k.back() = c->alloc_gen;      // every buffer of the lane that the captured launches name (model, slot, params, partials)
            if (c->lane_graph && c->lane_graph_key == k) continue;
            if (c->lane_graph) {
                hipGraphExecDestroy(c->lane_graph);
                c->lane_graph = nullptr;
            }
            hipGraph_t g = nullptr;
            c->no_timing = true;
            hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed);
            int st = GPBO_OK;
            if (e == hipSuccess) {
                st = enqueue_setting(c);
                e = hipStreamEndCapture(c->stream, &g);
            }
            c->no_timing = false;
            if (e == hipSuccess && st == GPBO_OK) e = hipGraphInstantiate(&c->lane_graph, g, nullptr, nullptr, 0);
            if (g) hipGraphDestroy(g);
            if (st != GPBO_OK) {
                if (c != ctx) ctx->err = c->err;
                        return st;
            }
            if (e != hipSuccess) return fail_hip("lml_batch: graph capture", e);
            c->lane_graph_key = k;
        }
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (int l = 0; l < lanes; ++l) timing_begin(lane[l]);
    for (int b = 0, r = 0; b < B; b += lane[r % lanes]->batch, ++r) {
        gpbo_ctx* c = lane[r % lanes];
        hipLaunchKernelGGL(k_pick_setting, dim3((unsigned)c->batch), dim3(64), 0, c->stream, b, (int)B, (const double*)ctx->lb_th,
                           (const double*)ctx->lb_sg, (int)d, (int)n, c->lane_slot);
        if (use_graph) {
            LCHK(hipGraphLaunch(c->lane_graph, c->stream));
        } else {
            const int st = enqueue_setting(c);
            if (st != GPBO_OK) {
                if (c != ctx) ctx->err = c->err;
                for (int l = 0; l < lanes; ++l) hipStreamSynchronize(lane[l]->stream);
                        return st;
            }
            // the event pool of a lane grows with every launch sequence; recycle it when the lane's stream is idle enough
            if (c->pool_used > 4096) {
                hipStreamSynchronize(c->stream);
                timing_begin(c);
            }
        }
    }
    LCHK(hipGetLastError());
    for (int l = 0; l < lanes; ++l) LCHK(hipStreamSynchronize(lane[l]->stream));
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::vector<int> hinfo(B);
    LCHK(hipMemcpy(nll_out, ctx->lb_nll, (size_t)B * 8, hipMemcpyDeviceToHost));
    if (grad_out) LCHK(hipMemcpy(grad_out, ctx->lb_grad, (size_t)B * d * 8, hipMemcpyDeviceToHost));
    LCHK(hipMemcpy(hinfo.data(), ctx->lb_info, (size_t)B * sizeof(int), hipMemcpyDeviceToHost));
#undef LCHK
    ctx->tm.fit_total_ms = wall_ms;      // host wall time of the whole sweep (several streams run concurrently)
    ctx->tm.chol_count = B;
    ctx->tm.kbuild_ms = ctx->tm.chol_ms = ctx->tm.trsv_ms = ctx->tm.inverse_ms = 0;
    for (int b = 0; b < B; ++b) {
        const bool bad = hinfo[b] != 0 || !std::isfinite(nll_out[b]);
        if (status_out) status_out[b] = bad ? GPBO_ERR_NOT_PD : GPBO_OK;
        if (bad) nll_out[b] = NAN;
    }
    return GPBO_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- acquisition without any allocation
// gpbo_acquire_static: the kernels of gpbo_acquire (k_max_stage, k_acquire, k_argmax_stage) on caller-owned device
// arrays, with the reduction partials in a file-scope device array instead of buffers a context allocates.
namespace {
constexpr int STATIC_BLOCKS = 1024;                        // first-stage workgroups of the two reductions
constexpr int64_t STATIC_MAX_M = (int64_t)1 << 28;         // candidates per call (268,435,456)
// layout (doubles): [0,1024) max partials | [1024,2048) argmax values | [2048,3072) argmax indices (int64)
//                   [3072] incumbent  [3073] max(mu)  [3074] best value  [3075] best index (int64)
constexpr int STATIC_SCRATCH_DOUBLES = 4096;               // 32 KiB per GPU, held until the process ends
static __device__ __align__(256) double scratch[STATIC_SCRATCH_DOUBLES];
constexpr int STATIC_MAX_DEVICES = 64;
double* g_scratch_ptr[STATIC_MAX_DEVICES] = {};           // per device: the address of that GPU's copy of `scratch`
std::mutex g_scratch_mutex;

// The array's name is a host-side handle, not a device address: the address is looked up once per device, with that
// device current, and only that pointer is ever given to kernels and copies.
hipError_t static_scratch(int device, double** out) {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        double* p = nullptr;
        hipError_t e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}
}  // namespace

extern "C" int gpbo_acquire_static(int32_t device, void* stream, const double* mu_dev, const double* var_dev, int64_t m,
                                   int32_t acq, double kappa, int32_t incumbent_mode, double incumbent, double* acq_out_dev,
                                   double* result_dev, int64_t* argmax_out, double* best_out, double* mu_max_out) {
    if (!mu_dev || !var_dev || !acq_out_dev) return GPBO_ERR_BAD_ARG;
    if (m <= 0 || m > STATIC_MAX_M) return GPBO_ERR_BAD_ARG;                 // beyond the stated maximum: rejected, nothing launched
    if (acq < 0 || acq > 2) return GPBO_ERR_BAD_ARG;
    if (incumbent_mode != GPBO_INCUMBENT_MAX_MU && incumbent_mode != GPBO_INCUMBENT_GIVEN) return GPBO_ERR_BAD_ARG;   // no model here: no max(yTrain)
    if (device < 0 || device >= STATIC_MAX_DEVICES) return GPBO_ERR_BAD_ARG;
    // the caller's current device is put back on every way out (a torch caller may be working on another GPU)
    struct DeviceGuard {
        int prev = -1;
        ~DeviceGuard() {
            if (prev >= 0) (void)hipSetDevice(prev);
        }
    } guard;
    if (hipGetDevice(&guard.prev) != hipSuccess) guard.prev = -1;
    if (hipSetDevice(device) != hipSuccess) return GPBO_ERR_HIP;
    double* sp = nullptr;
    if (static_scratch(device, &sp) != hipSuccess) return GPBO_ERR_HIP;
    hipStream_t st = (hipStream_t)stream;
    double* part_max = sp;
    double* part_v = sp + STATIC_BLOCKS;
    long* part_i = reinterpret_cast<long*>(sp + 2 * STATIC_BLOCKS);
    double* inc = sp + 3 * STATIC_BLOCKS;
    double* mu_max = inc + 1;
    double* best_v = inc + 2;
    long* best_i = reinterpret_cast<long*>(inc + 3);
    const int blocks = (int)std::min<int64_t>(STATIC_BLOCKS, (m + 255) / 256);
    // every scratch word read below is written first by this call (the stages write exactly `blocks` partials and the
    // four scalars): nothing of an earlier call's contents is consumed, nothing needs clearing
    hipLaunchKernelGGL(k_max_stage, dim3(blocks), dim3(256), 0, st, mu_dev, (long)m, part_max);
    hipLaunchKernelGGL(k_max_stage, dim3(1), dim3(256), 0, st, (const double*)part_max, (long)blocks, mu_max);
    if (incumbent_mode == GPBO_INCUMBENT_MAX_MU)
        hipLaunchKernelGGL(k_copy_scalar, dim3(1), dim3(1), 0, st, inc, (const double*)mu_max);
    else
        hipLaunchKernelGGL(k_set_scalar, dim3(1), dim3(1), 0, st, inc, incumbent);
    hipLaunchKernelGGL(k_acquire, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, mu_dev, var_dev, (long)m, (int)acq, kappa,
                       (const double*)inc, acq_out_dev);
    hipLaunchKernelGGL(k_argmax_stage, dim3(blocks), dim3(256), 0, st, (const double*)acq_out_dev, (const long*)nullptr, (long)m,
                       part_v, part_i);
    hipLaunchKernelGGL(k_argmax_stage, dim3(1), dim3(256), 0, st, (const double*)part_v, (const long*)part_i, (long)blocks, best_v,
                       best_i);
    if (result_dev) hipLaunchKernelGGL(k_copy4, dim3(1), dim3(64), 0, st, result_dev, (const double*)inc);
    if (hipGetLastError() != hipSuccess) return GPBO_ERR_HIP;
    if (!argmax_out && !best_out && !mu_max_out) return GPBO_OK;      // launches only: legal on a capturing stream
    double h[4] = {0, 0, 0, 0};      // incumbent, max(mu), best value, best index (bit pattern of an int64)
    if (hipMemcpyAsync(h, inc, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess) return GPBO_ERR_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return GPBO_ERR_HIP;
    long bi;
    memcpy(&bi, &h[3], sizeof(bi));
    if (bi == INT64_MAX) bi = 0;     // (no candidate at all; a NaN or -inf value is picked by its index like any other)
    if (argmax_out) *argmax_out = bi;
    if (best_out) *best_out = h[2];
    if (mu_max_out) *mu_max_out = h[1];
    return GPBO_OK;
}

// ---------------------------------------------------------------- self test of the tile engine
namespace {
template <bool AK, bool BKC>
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_selftest_gemm(const double* A, const double* B, double* C, int n, int reps) {
    __shared__ double lds[GEMM_LDS_DOUBLES];
    f64x4 acc[4][4];
    acc_zero(acc);
    const int ib = blockIdx.y, jb = blockIdx.x;
    const double* Ap = AK ? A + (long)ib * 128 * n : A + (long)ib * 128;
    const double* Bp = BKC ? B + (long)jb * 128 * n : B + (long)jb * 128;
    (void)reps;
    gemm_tile_128<AK, BKC>(Ap, n, Bp, n, 0, n, acc, lds);
    double* Ct = C + (long)ib * 128 * n + (long)jb * 128;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ct[(long)acc_row(ta, r) * n + acc_col(tb)] = acc[ta][tb][r];
}

void launch_selftest(hipStream_t st, int a_kc, int b_kc, const double* A, const double* B, double* C, int n, int reps) {
    dim3 grid(n / 128, n / 128), blk(GEMM_THREADS);
    if (a_kc && b_kc) hipLaunchKernelGGL((k_selftest_gemm<true, true>), grid, blk, 0, st, A, B, C, n, reps);
    else if (a_kc) hipLaunchKernelGGL((k_selftest_gemm<true, false>), grid, blk, 0, st, A, B, C, n, reps);
    else if (b_kc) hipLaunchKernelGGL((k_selftest_gemm<false, true>), grid, blk, 0, st, A, B, C, n, reps);
    else hipLaunchKernelGGL((k_selftest_gemm<false, false>), grid, blk, 0, st, A, B, C, n, reps);
}
}  // namespace

extern "C" {

int gpbo_selftest_gemm(gpbo_ctx* ctx, int32_t n, double* err_out, double* rate_out) {
    if (!ctx) return GPBO_ERR_BAD_ARG;
    if (n <= 0 || n % 128) BADARG("selftest: n must be a positive multiple of 128");
    HIPCHK(hipSetDevice(ctx->device));
    const size_t nn = (size_t)n * n;
    std::vector<double> A(nn), B(nn), C(nn);
    uint64_t s = 12345;
    auto rnd = [&]() {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        return (double)(int)((s >> 40) % 17) - 8.0;   // small integers: products are exact
    };
    for (auto& v : A) v = rnd();
    for (auto& v : B) v = rnd();
    double *dA = nullptr, *dB = nullptr, *dC = nullptr;
    struct Guard {      // the three device matrices are released on every way out
        double *&a, *&b, *&c;
        ~Guard() {
            for (double* p : {a, b, c})
                if (p) hipFree(p);
        }
    } guard{dA, dB, dC};
    HIPCHK(hipMalloc((void**)&dA, nn * 8));
    HIPCHK(hipMalloc((void**)&dB, nn * 8));
    HIPCHK(hipMalloc((void**)&dC, nn * 8));
    HIPCHK(hipMemcpy(dA, A.data(), nn * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dB, B.data(), nn * 8, hipMemcpyHostToDevice));
    double worst = 0;
    const int nchk = std::min(n, 256);
    for (int mode = 0; mode < 4; ++mode) {
        const int a_kc = mode & 1, b_kc = (mode >> 1) & 1;
        launch_selftest(ctx->stream, a_kc, b_kc, dA, dB, dC, n, 1);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(C.data(), dC, nn * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        // element (i,k) of A: a_kc ? A[i*n+k] : A[k*n+i];  (k,j) of B: b_kc ? B[j*n+k] : B[k*n+j]
        for (int i = 0; i < nchk; ++i)
            for (int j = 0; j < nchk; ++j) {
                const int ii = (i * 37) % n, jj = (j * 91 + 5) % n;
                double ref = 0;
                for (int k = 0; k < n; ++k)
                    ref += (a_kc ? A[(size_t)ii * n + k] : A[(size_t)k * n + ii]) * (b_kc ? B[(size_t)jj * n + k] : B[(size_t)k * n + jj]);
                worst = std::max(worst, std::fabs(ref - C[(size_t)ii * n + jj]));
            }
    }
    if (err_out) *err_out = worst;
    if (rate_out) {
        const int reps = 1;
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        launch_selftest(ctx->stream, 1, 0, dA, dB, dC, n, reps);
        hipEventRecord(a, ctx->stream);
        launch_selftest(ctx->stream, 1, 0, dA, dB, dC, n, reps);
        hipEventRecord(b, ctx->stream);
        HIPCHK(hipStreamSynchronize(ctx->stream));
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        hipEventDestroy(a);
        hipEventDestroy(b);
        *rate_out = 2.0 * (double)n * n * n * reps / (ms * 1e-3) / 1e12;
    }
    return GPBO_OK;
}

}  // extern "C"

#include "multi.hpp"
