/* libgpbo — C ABI of the MI355X (gfx950) GP posterior + acquisition hot path.
 *
 * The reference (kgururangan/Bayesian-Optimization) has no FFI or plugin interface;
 * the boundary this ABI sits behind is the set of Python function signatures in the
 * function cell of BayesOpt/BayesOpt_v4.ipynb (gprfcn :127-152, expected_improvement
 * :188-197, mlefcn :154-165) and, for the trend-GP flavour,
 * BayesOpt/Gaussian_Process_Regression.ipynb (gpr_poly :197-203, y_kriging_v2 :228-234).
 * Each entry point below names the reference lines whose arithmetic it replaces.
 *
 * Conventions
 *   - all arrays are C-order float64; pointers are HOST pointers unless the name says
 *     "device"; the caller owns every buffer it passes and the library keeps no host
 *     pointer after a call returns;
 *   - every call returns an int status: GPBO_OK, GPBO_ERR_NOT_PD (Cholesky pivot <= 0,
 *     what numpy.linalg.cholesky reports as LinAlgError; gpbo_last_info() gives the
 *     1-based failing row), GPBO_ERR_BAD_ARG, GPBO_ERR_HIP (gpbo_last_error() has text);
 *   - a context owns one GPU's state (device buffers, one HIP stream).  Calls block
 *     until the results they return are complete.  One context per process and GPU;
 *     a context is not re-entrant.
 */
#ifndef GPBO_H
#define GPBO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPBO_OK 0
#define GPBO_ERR_NOT_PD 1
#define GPBO_ERR_BAD_ARG 2
#define GPBO_ERR_HIP 3

#define GPBO_KERNEL_RBF 0      /* exp(-D/2), BayesOpt_v4.ipynb:119-124 */
#define GPBO_KERNEL_MATERN52 1 /* (1+sqrt5 r+5r^2/3)exp(-sqrt5 r); no reference implementation */

#define GPBO_VARIANT_V4 0      /* two-factor posterior, BayesOpt_v4.ipynb:142-151 */
#define GPBO_VARIANT_SINGLE 1  /* single factor, theta1-scaled variance, GP/BayesOptFcn_v1.m:73-83; the scaled-factor form of
                                * GP/BayesOpt_v2.m:177-189 (L <- sqrt(theta1) L) is the same posterior analytically */
#define GPBO_VARIANT_M3 3      /* MATLAB two-factor form, GP/BayesOpt_v3.m:343-362: variance as v4, mean with alpha taken from
                                * the second factor, Mu = (theta1 Ks0)(theta1 K0 + eps)^-1 y (differs from v4 when sigmaY != 0);
                                * jitter eps*n with MATLAB's eps = 2^-52 when sigmaY == 0 (:343,352); -logL from y'alpha and
                                * diag(L) of the second factor (:362) */

#define GPBO_ACQ_EI 0          /* BayesOpt_v4.ipynb:188-197 */
#define GPBO_ACQ_UCB 1         /* GP/BayesOpt_v1.m:248-251 */
#define GPBO_ACQ_PI 2          /* Gaussian_Process_Regression.ipynb:265-268 */

#define GPBO_INCUMBENT_MAX_MU 0 /* max posterior mean over the candidates, BayesOpt_v4.ipynb:193 */
#define GPBO_INCUMBENT_MAX_Y 1  /* max(yTrain), GP/GPFR.m:48 */
#define GPBO_INCUMBENT_GIVEN 2  /* caller-supplied (multi-GPU: the all-reduced max) */

typedef struct gpbo_ctx gpbo_ctx;

/* Milliseconds measured with HIP events on the context's own stream, for the most
 * recent call of each kind.  *_launches count kernel launches inside that total. */
typedef struct gpbo_timings {
    double fit_total_ms;
    double kbuild_ms;      /* K(X,X) builds */
    double chol_ms;        /* all Cholesky factorisations of the fit */
    double chol_count;
    double trsv_ms;        /* forward + backward solves for alpha */
    double inverse_ms;     /* triangular inverse W = L^-1 */
    double posterior_total_ms;
    double ksbuild_ms;     /* K(X*,X) tile build + mean partials */
    double trmm_ms;        /* W * Ks^T with fused column sum of squares */
    double trmm_launches;
    double finalize_ms;    /* variance/mean finalisation + max(mu) */
    double acquire_ms;     /* acquisition + argmax */
    double trmm_flops;     /* algorithmic flops of the trmm launches: n_pad-free count n^2 per candidate */
    double append_total_ms;  /* last gpbo_append: factor update + sweep + max(mu) */
    double append_sweep_ms;  /* the update of the candidates' posterior inside the last gpbo_append */
    double append_route;     /* how that update ran: 0 = pass over the resident V, 1 = product of per-axis factors (lattice),
                                2 = the same with the acquisition of gpbo_append_acquire evaluated inside the update launch */
} gpbo_timings;

/* Device-side description of a fitted model, for broadcasting it to other ranks
 * (every pointer is a DEVICE pointer owned by the context). */
typedef struct gpbo_state_desc {
    int64_t n, n_pad;
    int32_t d, kernel, variant, pad_;
    double sigmaY;
    void* W;       int64_t W_bytes;       /* n_pad x n_pad, inverse Cholesky factor, lower */
    void* alpha;   int64_t alpha_bytes;   /* n_pad */
    void* Xts;     int64_t Xts_bytes;     /* d x n_pad, training inputs / theta, dimension-major */
    void* params;  int64_t params_bytes;  /* theta1, scale factors, 1/theta */
    double y_max;
} gpbo_state_desc;

int gpbo_create(int device, gpbo_ctx** out);
void gpbo_destroy(gpbo_ctx* ctx);
const char* gpbo_last_error(gpbo_ctx* ctx);
int64_t gpbo_last_info(gpbo_ctx* ctx);
/* The HIP stream (hipStream_t) every kernel of this context is launched on. */
void* gpbo_stream(gpbo_ctx* ctx);

/* Fit: everything of gprfcn that does not depend on xTest — K(X,X) (BayesOpt_v4.ipynb:131,134),
 * jitter rule (:138-141), Cholesky (:142, and :147 for the v4 variant), alpha (:143),
 * theta1 (:144), and -logL as mlefcn computes it (:154-165).  X is (n,d), y is (n,),
 * theta is (d,).  theta1_out / nll_out may be NULL. */
int gpbo_fit(gpbo_ctx* ctx, const double* X, const double* y, int64_t n, int32_t d,
             const double* theta, double sigmaY, int32_t kernel, int32_t variant,
             double* theta1_out, double* nll_out);

/* Make a candidate set (m,d) resident in HBM.  on_device != 0: Xs is a device pointer
 * and is copied device-to-device.  Replaces distmat_v2(xTest,xTrain) inputs (:132). */
int gpbo_set_candidates(gpbo_ctx* ctx, const double* Xs, int64_t m, int32_t on_device);

/* A synthetic candidate set generated in HBM: rows [row_offset, row_offset + m) of the (.., d) set whose element e
 * (row-major) is the counter-based splitmix64 uniform u(seed, e) in [0,1) -- the same numbers, bit for bit, as
 * bayesian_optimization_amd/synth.py:uniform on the host.  Stands where the notebooks build their candidate grid on
 * the host (BayesOpt_v4.ipynb:705-714) for benchmark sets too large to generate and upload (64 M x 8 = 4.3 GB). */
int gpbo_set_candidates_uniform(gpbo_ctx* ctx, uint64_t seed, int64_t row_offset, int64_t m);

/* The candidate grid the notebooks build with nested loops (BayesOpt_v4.ipynb:705-714, the 400-point grid at :248): every
 * combination of per-axis coordinates, in lexicographic order with the LAST axis fastest, generated in HBM.  axes: the
 * coordinates of axis 0, then axis 1, ... (sizes[t] of them for axis t; d = the model's d); m = prod sizes[t].  Everything
 * that works on a candidate set works on this one.  In addition, with the RBF kernel and a model fitted after
 * gpbo_reserve_appends (single-factor variant), gpbo_append / gpbo_append_acquire take the new row of L^-1 K(X, X*) as ONE
 * matrix product of per-axis kernel factors on the fp64 MFMA pipe (the kernel is a product over the axes) instead of a
 * pass over a resident (n x m) matrix, which is then not allocated at all: same posterior to rounding, a fraction of the
 * memory traffic (lattices whose axes ceil(d / 2) .. d-1 have at least two combinations: with one axis there is nothing to
 * factorise and the pass over the resident matrix stays).  (GPBO_LATTICE=0 in the environment keeps the streaming route, for comparison.) */
int gpbo_set_candidates_lattice(gpbo_ctx* ctx, const double* axes, const int32_t* sizes, int32_t d);

/* Pass 1 over the resident candidates: posterior mean and variance
 * (BayesOpt_v4.ipynb:135,146,149-151; only diag(Cov) is formed) kept in HBM, and the
 * maximum posterior mean of this context's candidates (:193). */
int gpbo_posterior(gpbo_ctx* ctx, double* mu_max_out);

/* Pass 2: acquisition over the resident posterior + first-occurrence argmax
 * (expected_improvement :188-197, np.argmax :647).  incumbent is used when
 * incumbent_mode == GPBO_INCUMBENT_GIVEN.  argmax_out is the index into this
 * context's candidate set (lowest index on ties); best_out the acquisition there. */
int gpbo_acquire(gpbo_ctx* ctx, int32_t acq, double kappa, int32_t incumbent_mode,
                 double incumbent, int64_t* argmax_out, double* best_out);

/* Lowest candidate index whose acquisition value (as the last gpbo_acquire left it) is >= threshold; -1 if there is none.
 * With threshold = best - rtol*|best| this is a tie-tolerant argmax: on an equispaced grid such as the notebooks'
 * (BayesOpt_v4.ipynb:248) candidates mirrored about a training point have EI equal to the last bit or two, and which
 * of them np.argmax (:647) returns is round-off of the evaluation order -- the reference's own included. */
int gpbo_first_at_least(gpbo_ctx* ctx, double threshold, int64_t* index_out);

/* The same acquisition pass for callers that cannot allocate: it has NO counterpart in the reference and exists for
 * code under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc.  It needs no context and calls
 * no hipMalloc / hipFree or any allocating routine: the partial results of its two reductions live in a 32 KiB
 * file-scope device array that every GPU the library runs on holds until the process ends.
 *   mu_dev, var_dev (m,) and acq_out_dev (m,) are DEVICE pointers owned by the caller; stream is a hipStream_t (NULL =
 *   the default stream).  result_dev (DEVICE pointer to 4 doubles, may be NULL) receives {incumbent, max(mu), best
 *   acquisition value, its index as the bit pattern of an int64}.  argmax_out / best_out / mu_max_out are host pointers
 *   (any may be NULL): when at least one is given the call copies the results back and synchronises the stream; when all
 *   three are NULL it only enqueues kernels -- no copy, no synchronisation -- which is the form to use on a capturing
 *   stream (read the results from result_dev after the graph has run).  incumbent_mode: GPBO_INCUMBENT_MAX_MU or
 *   GPBO_INCUMBENT_GIVEN (there is no model here, so no max(yTrain)).  Same arithmetic and tie rule as gpbo_acquire:
 *   results are bit-identical.  The caller's current device is restored before returning.
 * Limits: 1 <= m <= 2^28 (268,435,456) candidates per call, larger sizes return GPBO_ERR_BAD_ARG and launch nothing;
 * device < 64.  All calls on one GPU share that GPU's array: calls that overlap in time on one GPU (two host threads,
 * two streams) race and must be serialised by the caller. */
int gpbo_acquire_static(int32_t device, void* stream, const double* mu_dev, const double* var_dev, int64_t m,
                        int32_t acq, double kappa, int32_t incumbent_mode, double incumbent, double* acq_out_dev,
                        double* result_dev, int64_t* argmax_out, double* best_out, double* mu_max_out);

/* Copy count results starting at candidate offset to host; any pointer may be NULL.  acq: the values the last gpbo_acquire
 * stored; GPBO_ERR_BAD_ARG if the posterior has changed since (new candidates, model, appended point, uploaded posterior). */
int gpbo_read(gpbo_ctx* ctx, double* mu, double* var, double* acq, int64_t offset, int64_t count);

/* One call = set_candidates + posterior + acquire + read: the whole of
 * gprfcn(xTest,...) followed by expected_improvement and argmax for host buffers. */
int gpbo_predict_ei(gpbo_ctx* ctx, const double* Xs, int64_t m, double kappa,
                    int32_t incumbent_mode, double incumbent,
                    double* mu_out, double* var_out, double* ei_out,
                    int64_t* argmax_out, double* ei_max_out, double* mu_max_out);

/* Border-append one training point with theta frozen (single-factor variant only)
 * and update the resident posterior of the candidate set in place.  The reference
 * refits from scratch each iteration (BayesOpt_v4.ipynb:642-645); this is the same
 * posterior for frozen theta (GP/BayesOptFcn_v1.m:73-83). */
int gpbo_append(gpbo_ctx* ctx, const double* x_new, double y_new, double* mu_max_out);

/* One iteration of the frozen-theta loop in one call and one host synchronisation: gpbo_append(x_new, y_new) followed by
 * gpbo_acquire over the updated posterior (the body of BayesOpt_v4.ipynb:645-655 for the point picked last time and the
 * pick of the next one).  Same results, bit for bit, as the two calls; the acquisition array itself is not stored:
 * gpbo_read with an acq pointer and gpbo_first_at_least return GPBO_ERR_BAD_ARG until the next gpbo_acquire. */
int gpbo_append_acquire(gpbo_ctx* ctx, const double* x_new, double y_new, int32_t acq, double kappa, int32_t incumbent_mode,
                        double incumbent, double* mu_max_out, int64_t* argmax_out, double* best_out);

/* Trend-GP ("kriging") flavour.  gpr_poly (Gaussian_Process_Regression.ipynb:197-203): isotropic
 * K = exp(-D^2/(2 theta^2)) WITHOUT jitter, trend basis polymat(deg) (:113-140; d <= 3), GLS
 * beta (:201), cov = r^T K^-1 r / n (:202).  beta_out must hold 10 doubles.  After it,
 * gpbo_set_candidates + gpbo_posterior give y_kriging_v2 (:228-234) for every candidate:
 * mu = f beta + r^T K^-1 (y - F beta) and, in the variance slot, mse = cov^2 (1 - r^T K^-1 r)
 * (not clamped, as in the reference).
 * K is factored by Cholesky; when a pivot is not positive (K without jitter is often not numerically positive definite:
 * the notebook's own 3-D cell, :811-874, has cond(K) = 4e19) the call falls back, for n <= 8192, to LU with partial pivoting
 * and the explicit inverse -- the route of the reference's np.linalg.solve -- and returns numbers wherever the reference
 * does (GPBO_ERR_NOT_PD only for an exactly singular matrix, numpy's "Singular matrix").  What those numbers are worth
 * at such a condition number is another matter: the reference's own are not reproducible there either. */
int gpbo_kriging_fit(gpbo_ctx* ctx, const double* X, const double* y, int64_t n, int32_t d, double theta,
                     int32_t deg, double* beta_out, int32_t* nbeta_out, double* cov_out);

/* gpr_mle (Gaussian_Process_Regression.ipynb:255-263), the function the GPR notebook's descent minimises (:476, :622), for B
 * values of theta:  1/2 log det chol(K + epsilon I) + r^T r / (2 cov^2) + n/2 log(2 pi cov^2),  r = y - F beta, with beta and
 * cov from gpr_poly on the un-jittered K (as above, LU fallback included); epsilon only enters the factor whose
 * determinant is taken.  nll_out (B,); status_out (B,) may be NULL: GPBO_ERR_NOT_PD (and nll = NaN) where K + epsilon I has a
 * non-positive pivot (np.linalg.cholesky raises there) or K is exactly singular.  Runs in a private work context:
 * the model, candidates and posterior of ctx are untouched. */
int gpbo_kriging_lml(gpbo_ctx* ctx, const double* X, const double* y, int64_t n, int32_t d, const double* thetas, int32_t B,
                     int32_t deg, double epsilon, double* nll_out, int32_t* status_out);

/* y_kriging / mse_kriging (Gaussian_Process_Regression.ipynb:205-226, :236-253, 'gaussian' branch) evaluate the correlation
 * vector of a new point with a length scale of their own, r = exp(-d^2 / (2 w0)), against the K that gpr_poly built with
 * its theta.  After gpbo_kriging_fit, this call makes the candidate sweep use theta_r = sqrt(w0) for r (X: the training
 * inputs of the fit, (n,d)); the factor of K, beta and cov stay.  The next gpbo_kriging_fit resets it. */
int gpbo_kriging_cross_scale(gpbo_ctx* ctx, const double* X, double theta_r);

/* Make host arrays (m,) the resident posterior, so gpbo_acquire can run expected_improvement
 * (BayesOpt_v4.ipynb:188-197) / acquisition (Gaussian_Process_Regression.ipynb:265-268) on any Mu, diag(Cov).
 * Only the posterior arrays are replaced: the resident candidates stay as they are, and the next gpbo_posterior (or a
 * refit with the same column count followed by it) recomputes the posterior over exactly those candidates, whatever m
 * was here.  gpbo_append returns GPBO_ERR_BAD_ARG until gpbo_posterior has run again. */
int gpbo_set_posterior(gpbo_ctx* ctx, const double* mu, const double* var, int64_t m);

/* kernfcn_v2(distmat_v2(U,V), theta) (BayesOpt_v4.ipynb:100-105,119-124): the (p,q) covariance matrix
 * of two point sets, to host memory.  A data-format helper for callers that want K, Ks or Kss themselves. */
int gpbo_kernel_matrix(gpbo_ctx* ctx, const double* U, int64_t p, const double* V, int64_t q, int32_t d,
                       const double* theta, int32_t kernel, double* K_out);

/* kernfcn_v2(Daa, theta) on an explicit list of per-dimension |u_i - v_j| matrices (the notebooks' Daa,
 * BayesOpt_v4.ipynb:119-124): Dabs is d stacked (p,q) matrices, K_out is (p,q). */
int gpbo_kernel_from_distances(gpbo_ctx* ctx, const double* Dabs, int32_t d, int64_t p, int64_t q,
                               const double* theta, int32_t kernel, double* K_out);

/* Room for max_appends appended points; call BEFORE gpbo_fit (single-factor variant).  The next
 * gpbo_posterior then keeps V = L^-1 Ks^T (rows x candidates, 8 bytes each) resident in HBM. */
int gpbo_reserve_appends(gpbo_ctx* ctx, int64_t max_appends);

/* -logL (mlefcn, BayesOpt_v4.ipynb:154-165) and, when grad_out != NULL, its gradient
 * w.r.t. theta (jacmlefun, GP/BayesOpt_v3.m:392-417; the Matern-5/2 form has no reference counterpart) for B settings.
 * thetas is (B,d), sigmas (B,), nll_out (B,), grad_out (B,d).  status_out (B,) may be
 * NULL; a non-PD setting yields nll = NaN and status GPBO_ERR_NOT_PD for that entry.
 * The sweep runs in private work contexts owned by ctx: the model, candidates and posterior of ctx are untouched. */
int gpbo_lml_batch(gpbo_ctx* ctx, const double* X, const double* y, int64_t n, int32_t d,
                   const double* thetas, const double* sigmas, int32_t B, int32_t kernel,
                   double* nll_out, double* grad_out, int32_t* status_out);

/* conj_gradient_descent (BayesOpt_v4.ipynb:31-87) on -logL with the analytic gradient, for up to 128 training points and
 * every step rule: the nruns starts of hypopt (:167-178) run to the end on the device, one workgroup per
 * start, one launch for the whole multi-start fit.  W0 (nruns,d) starting points; norm: 0 'component' (sign(grad) sqrt(d)),
 * 1 'full' (grad / |grad|), 2 the raw gradient; steprule: 0 constant step alpha, 1 'diminishing' (alpha = 1/k),
 * 2 'backtrack' (:62-71: the step shrinks by 0.8 while -logL at the trial point exceeds the quadratic bound, at most 200
 * times per iteration; every trial point is evaluated inside the launch, one history entry per outer iteration).
 * w_hist_out (nruns, max_its+1, d) and c_hist_out (nruns, max_its+1) receive weight_history and cost_history of every
 * start (cost +inf where the covariance matrix is not positive definite or the iterate is no longer a set of finite
 * non-zero length scales: that start stops there, the iterate is still recorded), count_out (nruns) how
 * many entries each start wrote, flag_out (nruns) the exit flag (1: the step fell below 1e-10).
 * GPBO_ERR_BAD_ARG for n > 128 or d > 16: use gpbo_lml_batch and a host-side loop there. */
int gpbo_descent_small(gpbo_ctx* ctx, const double* X, const double* y, int64_t n, int32_t d, const double* W0,
                       int32_t nruns, double sigmaY, int32_t kernel, double alpha, double beta, int32_t norm,
                       int32_t steprule, int32_t max_its, double* w_hist_out, double* c_hist_out, int32_t* count_out,
                       int32_t* flag_out);

/* The factor gprfcn returns as L (the second one for the v4 variant, :147) as a dense
 * (n,n) lower-triangular matrix, and alpha as returned at :148,:152.  Either may be NULL. */
int gpbo_get_factor(gpbo_ctx* ctx, double* L_out, double* alpha_out);

int gpbo_last_timings(gpbo_ctx* ctx, gpbo_timings* out);

/* Per-phase event timing of the calls on this context on (1, the default) or off (0).  No counterpart in the reference:
 * the events behind gpbo_last_timings cost a few microseconds each, which shows in a loop of gpbo_append_acquire calls
 * (about 1 % of an iteration); a caller that does not read the timings switches them off, or on for a sample of calls. */
int gpbo_set_timing(gpbo_ctx* ctx, int32_t enabled);

/* Scalars of the current model: theta1 (BayesOpt_v4.ipynb:144), -logL (:164), max(yTrain), n.  Any may be NULL. */
int gpbo_model_scalars(gpbo_ctx* ctx, double* theta1_out, double* nll_out, double* y_max_out, int64_t* n_out);

/* Shapes the context currently holds: training points n, columns d, length m of the current posterior and the number of
 * candidate rows resident in HBM.  Valid after a failed call as well (a fit that ends in GPBO_ERR_NOT_PD has already
 * replaced n and d, and dropped resident candidates of another width): host wrappers re-read their bookkeeping from here. */
int gpbo_shape(gpbo_ctx* ctx, int64_t* n_out, int32_t* d_out, int64_t* m_out, int64_t* candidate_rows_out);

/* Conditioning diagnostic of the factored matrix (the one np.linalg.cholesky receives at BayesOpt_v4.ipynb:142, or
 * :147 for the v4 variant): (max diag L / min diag L)^2, a lower bound of its 2-norm condition number.  Needs the
 * factor on this context (the rank that called gpbo_fit). */
int gpbo_cond_estimate(gpbo_ctx* ctx, double* cond_out);

/* Multi-GPU: rank 0 fits, gpbo_state_describe() exposes the device buffers that the
 * other ranks need; they call gpbo_state_alloc() with the same shape, receive the
 * bytes into the described buffers (RCCL broadcast), then gpbo_state_commit(). */
int gpbo_state_describe(gpbo_ctx* ctx, gpbo_state_desc* out);
/* n_pad: the n_pad of the sender's gpbo_state_desc (a sender that reserved rows for appends pads further than
 * round_up(n, 128), and every broadcast buffer is sized by it); 0 = round_up(n, 128). */
int gpbo_state_alloc(gpbo_ctx* ctx, int64_t n, int64_t n_pad, int32_t d, int32_t kernel, int32_t variant, double sigmaY);
int gpbo_state_commit(gpbo_ctx* ctx, double y_max);

/* ---- Several GPUs behind one handle, in ONE process (the notebooks are one Jupyter process).  No counterpart in the
 * reference, which runs on one CPU: the loop body that is split is BayesOpt_v4.ipynb:642-647.  A gpbo_multi owns one
 * context per entry of `devices` (an entry may repeat: several contexts on one GPU) and one host thread per context.
 * Candidates are split into contiguous blocks in context order (the first m % N blocks one row longer), so a global
 * index is block offset + local index and the first-occurrence argmax (:647) survives the split.  gpbo_multi_fit fits on
 * context 0 and hands W = L^-1, alpha, the scaled inputs and the scalar block to the other contexts device to device: W
 * in N - 1 slices, slice s to context s + 1, then exchanged among the receivers, so that every link carries 2 / (N - 1)
 * of W.  (After gpbo_multi_reserve_appends a single-factor fit is instead carried out on every context: the frozen-theta
 * loop appends on all of them and never exchanges the model.)  gpbo_multi_posterior / _acquire run all blocks side by
 * side; max(Mu) (the EI incumbent, :193) and the (value, lowest index) pair of the argmax are reduced on the host, with
 * numpy's NaN rules.  Every result is bit-identical to one context sweeping the whole set.
 * Status codes, argument meaning and ownership as for the single-context calls of the same name. */
typedef struct gpbo_multi gpbo_multi;
int gpbo_device_count(void);
int gpbo_multi_create(const int32_t* devices, int32_t n_contexts, gpbo_multi** out);
void gpbo_multi_destroy(gpbo_multi* mg);
const char* gpbo_multi_last_error(gpbo_multi* mg);
int64_t gpbo_multi_last_info(gpbo_multi* mg);
int32_t gpbo_multi_size(gpbo_multi* mg);
/* Context i (owned by the handle; 0 holds the factor): for gpbo_last_timings, gpbo_get_factor, gpbo_kernel_matrix ... */
gpbo_ctx* gpbo_multi_context(gpbo_multi* mg, int32_t i);
/* Rows [lo, hi) of the current posterior that context i holds. */
int gpbo_multi_block(gpbo_multi* mg, int32_t i, int64_t* lo_out, int64_t* hi_out);
int gpbo_multi_shape(gpbo_multi* mg, int64_t* n_out, int32_t* d_out, int64_t* m_out, int64_t* candidate_rows_out);
int gpbo_multi_reserve_appends(gpbo_multi* mg, int64_t max_appends);
int gpbo_multi_set_chunk(gpbo_multi* mg, int64_t chunk);
int gpbo_multi_set_timing(gpbo_multi* mg, int32_t enabled);
int gpbo_multi_fit(gpbo_multi* mg, const double* X, const double* y, int64_t n, int32_t d, const double* theta, double sigmaY,
                   int32_t kernel, int32_t variant, double* theta1_out, double* nll_out);
int gpbo_multi_kriging_fit(gpbo_multi* mg, const double* X, const double* y, int64_t n, int32_t d, double theta, int32_t deg,
                           double* beta_out, int32_t* nbeta_out, double* cov_out);
int gpbo_multi_set_candidates(gpbo_multi* mg, const double* Xs, int64_t m, int32_t on_device);
/* Context i generates rows [row_offset + lo_i, row_offset + hi_i) of the synthetic set itself. */
int gpbo_multi_set_candidates_uniform(gpbo_multi* mg, uint64_t seed, int64_t row_offset, int64_t m);
/* gpbo_set_candidates_lattice over the contexts: slabs along axis 0, each again a lattice (contexts beyond the number of axis-0
 * points get no candidates). */
int gpbo_multi_set_candidates_lattice(gpbo_multi* mg, const double* axes, const int32_t* sizes, int32_t d);
int gpbo_multi_posterior(gpbo_multi* mg, double* mu_max_out);
int gpbo_multi_acquire(gpbo_multi* mg, int32_t acq, double kappa, int32_t incumbent_mode, double incumbent, int64_t* argmax_out,
                       double* best_out);
int gpbo_multi_first_at_least(gpbo_multi* mg, double threshold, int64_t* index_out);
int gpbo_multi_read(gpbo_multi* mg, double* mu, double* var, double* acq, int64_t offset, int64_t count);
int gpbo_multi_set_posterior(gpbo_multi* mg, const double* mu, const double* var, int64_t m);
int gpbo_multi_predict_ei(gpbo_multi* mg, const double* Xs, int64_t m, double kappa, int32_t incumbent_mode, double incumbent,
                          double* mu_out, double* var_out, double* ei_out, int64_t* argmax_out, double* ei_max_out,
                          double* mu_max_out);
int gpbo_multi_append(gpbo_multi* mg, const double* x_new, double y_new, double* mu_max_out);
/* With more than one context the incumbent max(Mu) needs the host reduction between the two halves: two synchronisations
 * per iteration instead of the one of gpbo_append_acquire. */
int gpbo_multi_append_acquire(gpbo_multi* mg, const double* x_new, double y_new, int32_t acq, double kappa, int32_t incumbent_mode,
                              double incumbent, double* mu_max_out, int64_t* argmax_out, double* best_out);
/* The B settings are independent: context i evaluates a contiguous block of them, nothing is exchanged. */
int gpbo_multi_lml_batch(gpbo_multi* mg, const double* X, const double* y, int64_t n, int32_t d, const double* thetas,
                         const double* sigmas, int32_t B, int32_t kernel, double* nll_out, double* grad_out, int32_t* status_out);
/* Wall time (ms) and bytes moved by the model hand-over of the last gpbo_multi_fit / gpbo_multi_kriging_fit. */
int gpbo_multi_last_distribute(gpbo_multi* mg, double* ms_out, double* bytes_out);

/* Tuning knob: candidates per posterior chunk (multiple of 128; 0 = default). */
int gpbo_set_chunk(gpbo_ctx* ctx, int64_t chunk);

/* Self-test of the fp64 MFMA tile engine against a host triple loop (max abs error
 * returned in err_out) and its sustained rate in TFLOP/s (rate_out may be NULL). */
int gpbo_selftest_gemm(gpbo_ctx* ctx, int32_t n, double* err_out, double* rate_out);

#ifdef __cplusplus
}
#endif
#endif /* GPBO_H */
