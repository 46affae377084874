// Several GPUs behind one handle, in one process: the candidate set of the BO loop body
// (BayesOpt/BayesOpt_v4.ipynb:642-647) is split into contiguous blocks, one per context; context 0 fits the model and
// hands the inverse factor, alpha, the scaled inputs and the scalars to the others device-to-device; every sweep runs
// on all contexts at once, one host thread per context, and the two scalars that tie the blocks together -- max(Mu),
// the EI incumbent (:193), and the (value, lowest index) of the argmax (:647) -- are reduced on the host.
// Included by gpbo.hip after the single-context entry points (it drives them).
#pragma once

#include <condition_variable>
#include <functional>

namespace {

// One host thread that runs the jobs of one context, so that the contexts' calls (launch sequences and the
// synchronisations inside them) proceed side by side.  Context 0 is driven by the calling thread.
class Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has = false, done = true, quit = false;

    void loop() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return has || quit; });
                if (quit && !has) return;
                f = std::move(job);
                has = false;
            }
            f();
            {
                std::lock_guard<std::mutex> lk(mu);
                done = true;
            }
            cv.notify_all();
        }
    }

public:
    Worker() : th([this] { loop(); }) {}
    ~Worker() {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cv.notify_all();
        th.join();
    }
    void submit(std::function<void()> f) {
        {
            std::lock_guard<std::mutex> lk(mu);
            job = std::move(f);
            has = true;
            done = false;
        }
        cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
    }
};

}  // namespace

struct gpbo_multi {
    std::vector<gpbo_ctx*> ctx;
    std::vector<std::unique_ptr<Worker>> workers;      // workers[i - 1] drives ctx[i]
    std::vector<int64_t> xoff;                         // candidate block of context i: rows [xoff[i], xoff[i + 1])
    std::vector<int64_t> poff;                         // the same for the current posterior (an uploaded one may differ)
    std::string err;
    int64_t info = 0;
    int64_t reserve = 0;
    bool replicated = false;      // every context holds its own factor (the frozen-theta loop appends on all of them)
    bool fitted = false;
    double mu_max = NAN;          // max(Mu) over all blocks of the current posterior
    double distribute_ms = 0, distribute_bytes = 0;
};

namespace {

#define MBADARG(msg)             \
    do {                         \
        mg->err = (msg);         \
        return GPBO_ERR_BAD_ARG; \
    } while (0)

// rows [lo, hi) of an m-row set for block i of N: contiguous, the first m % N blocks one row longer
void block_range(int64_t m, int i, int N, int64_t* lo, int64_t* hi) {
    const int64_t q = m / N, r = m % N;
    *lo = i * q + std::min<int64_t>(i, r);
    *hi = *lo + q + (i < r ? 1 : 0);
}

void make_offsets(std::vector<int64_t>& off, int64_t m, int N) {
    off.assign((size_t)N + 1, 0);
    for (int i = 0; i < N; ++i) {
        int64_t lo, hi;
        block_range(m, i, N, &lo, &hi);
        off[i] = lo;
        off[i + 1] = hi;
    }
}

// f(i) for every context i for which use(i) holds, side by side; the first failing context (lowest i) decides the
// status and the error text.
int run_all(gpbo_multi* mg, const std::function<bool(int)>& use, const std::function<int(int)>& f) {
    const int N = (int)mg->ctx.size();
    std::vector<int> st((size_t)N, GPBO_OK);
    for (int i = 1; i < N; ++i)
        if (use(i)) mg->workers[(size_t)i - 1]->submit([&st, &f, i] { st[(size_t)i] = f(i); });
    if (use(0)) st[0] = f(0);
    for (int i = 1; i < N; ++i)
        if (use(i)) mg->workers[(size_t)i - 1]->wait();
    for (int i = 0; i < N; ++i)
        if (st[(size_t)i] != GPBO_OK) {
            mg->err = "context " + std::to_string(i) + " (device " + std::to_string(mg->ctx[(size_t)i]->device) + "): " + mg->ctx[(size_t)i]->err;
            mg->info = mg->ctx[(size_t)i]->info;
            return st[(size_t)i];
        }
    return GPBO_OK;
}

bool every(int) { return true; }

// The fitted model of context 0 to every other context, device to device.  alpha, the scaled inputs and the scalars are a
// few hundred kilobytes and go straight from the source.  W (n_pad^2 doubles, 8.6 GB at n = 32768) is cut into N - 1
// slices of whole block rows: slice s goes to context s + 1 (stage 1: the source's N - 1 links carry one slice each),
// then every receiver fetches the other slices from the receivers that hold them (stage 2: an all-gather among the
// peers, every peer-to-peer link carries one slice).  Each link moves 2 / (N - 1) of W instead of all of it.
int multi_distribute(gpbo_multi* mg) {
    const int N = (int)mg->ctx.size();
    mg->distribute_ms = 0;
    mg->distribute_bytes = 0;
    if (N == 1) return GPBO_OK;
    gpbo_ctx* src = mg->ctx[0];
    const auto t0 = std::chrono::steady_clock::now();
    CHK(run_all(mg, [](int i) { return i >= 1; }, [&](int i) {
        return gpbo_state_alloc(mg->ctx[(size_t)i], src->n, src->n_pad, src->d, src->kernel, src->variant, src->sigmaY);
    }));
    const int P = N - 1;
    const int64_t nb = src->nb, n_pad = src->n_pad;
    std::vector<hipEvent_t> ev((size_t)N, nullptr);
    auto fail = [&](const char* what, hipError_t e) {
        for (int i = 0; i < N; ++i) (void)hipStreamSynchronize(mg->ctx[(size_t)i]->stream);
        for (hipEvent_t x : ev)
            if (x) (void)hipEventDestroy(x);
        mg->err = std::string("distribute: ") + what + ": " + hipGetErrorString(e);
        return GPBO_ERR_HIP;
    };
#define DCHK(expr)                                    \
    do {                                              \
        hipError_t e_ = (expr);                       \
        if (e_ != hipSuccess) return fail(#expr, e_); \
    } while (0)
    auto slice = [&](int s, int64_t* first, int64_t* bytes) {      // block rows [nb s / P, nb (s + 1) / P) of W
        const int64_t b0 = nb * s / P, b1 = nb * (s + 1) / P;
        *first = b0 * PB * n_pad;
        *bytes = (b1 - b0) * PB * n_pad * (int64_t)sizeof(double);
    };
    for (int r = 1; r < N; ++r) {
        gpbo_ctx* c = mg->ctx[(size_t)r];
        DCHK(hipSetDevice(c->device));
        DCHK(hipMemcpyPeerAsync(c->alpha, c->device, src->alpha, src->device, (size_t)n_pad * 8, c->stream));
        DCHK(hipMemcpyPeerAsync(c->Xts, c->device, src->Xts, src->device, (size_t)MAX_D * n_pad * 8, c->stream));
        DCHK(hipMemcpyPeerAsync(c->params, c->device, src->params, src->device, PARAMS_DOUBLES * 8, c->stream));
        if (src->variant == GPBO_VARIANT_KRIGING_LU) {      // the LU route of the trend flavour: bufW holds the factors, the row exchanges go with them
            if (c->ipiv_cap < n_pad) {
                if (c->ipiv) (void)hipFree(c->ipiv);
                c->ipiv = nullptr;
                c->ipiv_cap = 0;
                DCHK(hipMalloc((void**)&c->ipiv, (size_t)n_pad * sizeof(int)));
                c->ipiv_cap = n_pad;
            }
            DCHK(hipMemcpyPeerAsync(c->ipiv, c->device, src->ipiv, src->device, (size_t)n_pad * sizeof(int), c->stream));
        }
        int64_t first, bytes;
        slice(r - 1, &first, &bytes);
        if (bytes > 0) DCHK(hipMemcpyPeerAsync(c->bufW + first, c->device, src->bufW + first, src->device, (size_t)bytes, c->stream));
        DCHK(hipEventCreateWithFlags(&ev[(size_t)r], hipEventDisableTiming));
        DCHK(hipEventRecord(ev[(size_t)r], c->stream));
        mg->distribute_bytes += (double)bytes;
    }
    for (int k = 1; k < N; ++k) {
        gpbo_ctx* c = mg->ctx[(size_t)k];
        DCHK(hipSetDevice(c->device));
        for (int s = 0; s < P; ++s) {
            if (s == k - 1) continue;
            const gpbo_ctx* owner = mg->ctx[(size_t)s + 1];
            int64_t first, bytes;
            slice(s, &first, &bytes);
            if (bytes <= 0) continue;
            DCHK(hipStreamWaitEvent(c->stream, ev[(size_t)s + 1], 0));
            DCHK(hipMemcpyPeerAsync(c->bufW + first, c->device, owner->bufW + first, owner->device, (size_t)bytes, c->stream));
            mg->distribute_bytes += (double)bytes;
        }
    }
    for (int r = 1; r < N; ++r) {
        DCHK(hipSetDevice(mg->ctx[(size_t)r]->device));
        DCHK(hipStreamSynchronize(mg->ctx[(size_t)r]->stream));
    }
    for (hipEvent_t& x : ev)
        if (x) {
            (void)hipEventDestroy(x);
            x = nullptr;
        }
#undef DCHK
    for (int r = 1; r < N; ++r) {
        gpbo_ctx* c = mg->ctx[(size_t)r];
        c->theta = src->theta;
        const int st = gpbo_state_commit(c, src->y_max);
        if (st != GPBO_OK) {
            mg->err = "distribute: " + c->err;
            return st;
        }
    }
    (void)hipSetDevice(src->device);
    mg->distribute_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return GPBO_OK;
}

// value/index pairs of the blocks -> the overall winner: a NaN ranks above every number, equal ranks go to the lower
// global index (numpy.argmax over the concatenated array)
void pick_global(const std::vector<double>& v, const std::vector<int64_t>& gi, const std::vector<char>& used, double* best,
                 int64_t* idx) {
    double bv = -INFINITY;
    int64_t bi = INT64_MAX;
    for (size_t i = 0; i < v.size(); ++i) {
        if (!used[i]) continue;
        const bool vn = v[i] != v[i], bn = bv != bv;
        bool take;
        if (vn || bn) take = vn && (!bn || gi[i] < bi);
        else take = v[i] > bv || (v[i] == bv && gi[i] < bi);
        if (take) {
            bv = v[i];
            bi = gi[i];
        }
    }
    *best = bv;
    *idx = bi == INT64_MAX ? 0 : bi;
}

double host_max_nan(double a, double b) { return (a > b || a != a) ? a : b; }

int multi_posterior(gpbo_multi* mg, double* mu_max_out) {
    const int N = (int)mg->ctx.size();
    if (!mg->fitted) MBADARG("posterior: no fitted model");
    if (mg->xoff.empty() || mg->xoff[(size_t)N] <= 0) MBADARG("posterior: no candidates");
    std::vector<double> mm((size_t)N, -INFINITY);
    auto use = [&](int i) { return mg->xoff[(size_t)i + 1] > mg->xoff[(size_t)i]; };
    CHK(run_all(mg, use, [&](int i) { return gpbo_posterior(mg->ctx[(size_t)i], &mm[(size_t)i]); }));
    double g = -INFINITY;
    for (int i = 0; i < N; ++i)
        if (use(i)) g = host_max_nan(g, mm[(size_t)i]);
    mg->mu_max = g;
    mg->poff = mg->xoff;
    if (mu_max_out) *mu_max_out = g;
    return GPBO_OK;
}

int multi_acquire(gpbo_multi* mg, int acq, double kappa, int mode, double incumbent, int64_t* argmax_out, double* best_out) {
    const int N = (int)mg->ctx.size();
    if (mg->poff.empty() || mg->poff[(size_t)N] <= 0) MBADARG("acquire: no posterior (call gpbo_multi_posterior first)");
    if (mode < 0 || mode > 2) MBADARG("acquire: unknown incumbent mode");
    // the notebook's incumbent is the maximum of the posterior mean over ALL candidates (:193): every block gets the global one
    const int cmode = (mode == GPBO_INCUMBENT_MAX_MU) ? GPBO_INCUMBENT_GIVEN : mode;
    const double cinc = (mode == GPBO_INCUMBENT_MAX_MU) ? mg->mu_max : incumbent;
    std::vector<double> best((size_t)N, -INFINITY);
    std::vector<int64_t> idx((size_t)N, 0);
    std::vector<char> used((size_t)N, 0);
    auto use = [&](int i) { return mg->poff[(size_t)i + 1] > mg->poff[(size_t)i]; };
    CHK(run_all(mg, use, [&](int i) { return gpbo_acquire(mg->ctx[(size_t)i], acq, kappa, cmode, cinc, &idx[(size_t)i], &best[(size_t)i]); }));
    for (int i = 0; i < N; ++i) {
        used[(size_t)i] = use(i);
        idx[(size_t)i] += mg->poff[(size_t)i];
    }
    double b;
    int64_t gi;
    pick_global(best, idx, used, &b, &gi);
    if (argmax_out) *argmax_out = gi;
    if (best_out) *best_out = b;
    return GPBO_OK;
}

}  // namespace

extern "C" {

int gpbo_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

int gpbo_multi_create(const int32_t* devices, int32_t n_contexts, gpbo_multi** out) {
    if (!out) return GPBO_ERR_BAD_ARG;
    *out = nullptr;
    if (!devices || n_contexts <= 0 || n_contexts > 64) return GPBO_ERR_BAD_ARG;
    gpbo_multi* mg = new gpbo_multi();
    for (int i = 0; i < n_contexts; ++i) {
        gpbo_ctx* c = nullptr;
        const int st = gpbo_create(devices[i], &c);
        if (st != GPBO_OK) {
            for (gpbo_ctx* x : mg->ctx) gpbo_destroy(x);
            delete mg;
            return st;
        }
        mg->ctx.push_back(c);
        if (i >= 1) mg->workers.emplace_back(new Worker());
    }
    // direct copies between distinct devices where the hardware allows them (otherwise the runtime stages through the host)
    for (int a = 0; a < n_contexts; ++a)
        for (int b = 0; b < n_contexts; ++b) {
            const int da = devices[a], db = devices[b];
            if (da == db) continue;
            int can = 0;
            if (hipSetDevice(da) == hipSuccess && hipDeviceCanAccessPeer(&can, da, db) == hipSuccess && can) {
                const hipError_t e = hipDeviceEnablePeerAccess(db, 0);
                if (e != hipSuccess) (void)hipGetLastError();      // already enabled by an earlier handle (or by torch): fine
            }
        }
    (void)hipSetDevice(devices[0]);
    mg->xoff.assign((size_t)n_contexts + 1, 0);
    mg->poff.assign((size_t)n_contexts + 1, 0);
    *out = mg;
    return GPBO_OK;
}

void gpbo_multi_destroy(gpbo_multi* mg) {
    if (!mg) return;
    mg->workers.clear();      // joins the threads
    for (gpbo_ctx* c : mg->ctx) gpbo_destroy(c);
    delete mg;
}

const char* gpbo_multi_last_error(gpbo_multi* mg) { return mg ? mg->err.c_str() : "null handle"; }
int64_t gpbo_multi_last_info(gpbo_multi* mg) { return mg ? mg->info : 0; }
int32_t gpbo_multi_size(gpbo_multi* mg) { return mg ? (int32_t)mg->ctx.size() : 0; }
gpbo_ctx* gpbo_multi_context(gpbo_multi* mg, int32_t i) {
    return (mg && i >= 0 && i < (int32_t)mg->ctx.size()) ? mg->ctx[(size_t)i] : nullptr;
}

int gpbo_multi_block(gpbo_multi* mg, int32_t i, int64_t* lo_out, int64_t* hi_out) {
    if (!mg) return GPBO_ERR_BAD_ARG;
    if (i < 0 || i >= (int32_t)mg->ctx.size()) MBADARG("block: no such context");
    if (lo_out) *lo_out = mg->poff[(size_t)i];
    if (hi_out) *hi_out = mg->poff[(size_t)i + 1];
    return GPBO_OK;
}

int gpbo_multi_shape(gpbo_multi* mg, int64_t* n_out, int32_t* d_out, int64_t* m_out, int64_t* candidate_rows_out) {
    if (!mg) return GPBO_ERR_BAD_ARG;
    if (n_out) *n_out = mg->ctx[0]->n;
    if (d_out) *d_out = mg->ctx[0]->d;
    if (m_out) *m_out = mg->poff.back();
    if (candidate_rows_out) *candidate_rows_out = mg->xoff.back();
    return GPBO_OK;
}

int gpbo_multi_reserve_appends(gpbo_multi* mg, int64_t max_appends) {
    if (!mg) return GPBO_ERR_BAD_ARG;
    if (max_appends < 0) MBADARG("reserve_appends: negative");
    mg->reserve = max_appends;
    for (gpbo_ctx* c : mg->ctx) c->reserve = max_appends;
    return GPBO_OK;
}

int gpbo_multi_set_chunk(gpbo_multi* mg, int64_t chunk) {
    if (!mg) return GPBO_ERR_BAD_ARG;
    if (chunk < 0) MBADARG("set_chunk: negative");
    for (gpbo_ctx* c : mg->ctx) c->chunk_req = chunk;
    return GPBO_OK;
}

int gpbo_multi_set_timing(gpbo_multi* mg, int32_t enabled) {
    if (!mg) return GPBO_ERR_BAD_ARG;
    for (gpbo_ctx* c : mg->ctx) c->timing_off = (enabled == 0);
    return GPBO_OK;
}

int gpbo_multi_fit(gpbo_multi* mg, const double* X, const double* y, int64_t n, int32_t d, const double* theta, double sigmaY,
                   int32_t kernel, int32_t variant, double* theta1_out, double* nll_out) {
    if (!mg) return GPBO_ERR_BAD_ARG;
    const int N = (int)mg->ctx.size();
    const int32_t d_before = mg->ctx[0]->d;
    mg->fitted = false;
    mg->info = 0;
    mg->poff.assign((size_t)N + 1, 0);
    // a model that will be appended to needs its factor on every context: n is small there, each context fits its own
    // copy (deterministic kernels: the copies are bit-identical) and nothing is exchanged.  Otherwise context 0 fits
    // and the others receive what the sweep reads.
    mg->replicated = (variant == GPBO_VARIANT_SINGLE && mg->reserve > 0);
    int st;
    if (mg->replicated) {
        st = run_all(mg, every, [&](int i) {
            return gpbo_fit(mg->ctx[(size_t)i], X, y, n, d, theta, sigmaY, kernel, variant, nullptr, nullptr);
        });
        mg->distribute_ms = 0;
        mg->distribute_bytes = 0;
    } else {
        st = gpbo_fit(mg->ctx[0], X, y, n, d, theta, sigmaY, kernel, variant, nullptr, nullptr);
        if (st != GPBO_OK) {
            mg->err = mg->ctx[0]->err;
            mg->info = mg->ctx[0]->info;
        } else {
            st = multi_distribute(mg);
        }
    }
    if (mg->ctx[0]->d != d_before) mg->xoff.assign((size_t)N + 1, 0);      // candidates of another width were dropped
    if (st != GPBO_OK) return st;
    mg->fitted = true;
    if (theta1_out) *theta1_out = mg->ctx[0]->theta1;
    if (nll_out) *nll_out = mg->ctx[0]->nll;
    return GPBO_OK;
}

int gpbo_multi_kriging_fit(gpbo_multi* mg, const double* X, const double* y, int64_t n, int32_t d, double theta, int32_t deg,
                           double* beta_out, int32_t* nbeta_out, double* cov_out) {
    if (!mg) return GPBO_ERR_BAD_ARG;
    const int N = (int)mg->ctx.size();
    const int32_t d_before = mg->ctx[0]->d;
    mg->fitted = false;
    mg->replicated = false;
    mg->info = 0;
    mg->poff.assign((size_t)N + 1, 0);
    int st = gpbo_kriging_fit(mg->ctx[0], X, y, n, d, theta, deg, beta_out, nbeta_out, cov_out);
    if (st != GPBO_OK) {
        mg->err = mg->ctx[0]->err;
        mg->info = mg->ctx[0]->info;
    } else {
        st = multi_distribute(mg);
    }
    if (mg->ctx[0]->d != d_before) mg->xoff.assign((size_t)N + 1, 0);
    if (st != GPBO_OK) return st;
    mg->fitted = true;
    return GPBO_OK;
}

int gpbo_multi_set_candidates(gpbo_multi* mg, const double* Xs, int64_t m, int32_t on_device) {
    if (!mg) return GPBO_ERR_BAD_ARG;
    if (!Xs || m <= 0) MBADARG("set_candidates: empty candidate set");
    if (!mg->fitted) MBADARG("set_candidates: fit first (the candidate width is the model's d)");
    const int N = (int)mg->ctx.size();
    const int d = mg->ctx[0]->d;
    std::vector<int64_t> off;
    make_offsets(off, m, N);
    mg->xoff.assign((size_t)N + 1, 0);
    mg->poff.assign((size_t)N + 1, 0);
    CHK(run_all(mg, [&](int i) { return off[(size_t)i + 1] > off[(size_t)i]; }, [&](int i) {
        return gpbo_set_candidates(mg->ctx[(size_t)i], Xs + off[(size_t)i] * d, off[(size_t)i + 1] - off[(size_t)i], on_device);
    }));
    for (int i = 0; i < N; ++i)
        if (off[(size_t)i + 1] == off[(size_t)i]) mg->ctx[(size_t)i]->xs_rows = 0;      // an empty block holds no candidates
    mg->xoff = off;
    return GPBO_OK;
}

int gpbo_multi_set_candidates_uniform(gpbo_multi* mg, uint64_t seed, int64_t row_offset, int64_t m) {
    if (!mg) return GPBO_ERR_BAD_ARG;
    if (m <= 0 || row_offset < 0) MBADARG("set_candidates_uniform: need m >= 1 and row_offset >= 0");
    if (!mg->fitted) MBADARG("set_candidates_uniform: fit first (the candidate width is the model's d)");
    const int N = (int)mg->ctx.size();
    std::vector<int64_t> off;
    make_offsets(off, m, N);
    mg->xoff.assign((size_t)N + 1, 0);
    mg->poff.assign((size_t)N + 1, 0);
    CHK(run_all(mg, [&](int i) { return off[(size_t)i + 1] > off[(size_t)i]; }, [&](int i) {
        return gpbo_set_candidates_uniform(mg->ctx[(size_t)i], seed, row_offset + off[(size_t)i], off[(size_t)i + 1] - off[(size_t)i]);
    }));
    for (int i = 0; i < N; ++i)
        if (off[(size_t)i + 1] == off[(size_t)i]) mg->ctx[(size_t)i]->xs_rows = 0;
    mg->xoff = off;
    return GPBO_OK;
}

int gpbo_multi_set_candidates_lattice(gpbo_multi* mg, const double* axes, const int32_t* sizes, int32_t d) {
    if (!mg) return GPBO_ERR_BAD_ARG;
    if (!axes || !sizes || d < 1 || d > MAX_D) MBADARG("set_candidates_lattice: null input or bad number of axes");
    if (!mg->fitted) MBADARG("set_candidates_lattice: fit first (the candidate width is the model's d)");
    for (int t = 0; t < d; ++t)
        if (sizes[t] < 1) MBADARG("set_candidates_lattice: every axis needs at least one point");
    const int N = (int)mg->ctx.size();
    // slabs along axis 0: every context's block is again a lattice, contiguous in the lexicographic order
    int64_t total = 0, inner = 1;
    for (int t = 0; t < d; ++t) total += sizes[t];
    for (int t = 1; t < d; ++t) inner *= sizes[t];
    std::vector<int64_t> off((size_t)N + 1, 0), lo((size_t)N), hi((size_t)N);
    for (int i = 0; i < N; ++i) {
        block_range(sizes[0], i, N, &lo[(size_t)i], &hi[(size_t)i]);
        off[(size_t)i] = lo[(size_t)i] * inner;
        off[(size_t)i + 1] = hi[(size_t)i] * inner;
    }
    mg->xoff.assign((size_t)N + 1, 0);
    mg->poff.assign((size_t)N + 1, 0);
    std::vector<std::vector<double>> ax((size_t)N);
    std::vector<std::vector<int32_t>> sz((size_t)N);
    for (int i = 0; i < N; ++i) {
        if (hi[(size_t)i] <= lo[(size_t)i]) continue;
        ax[(size_t)i].assign(axes + lo[(size_t)i], axes + hi[(size_t)i]);
        ax[(size_t)i].insert(ax[(size_t)i].end(), axes + sizes[0], axes + total);
        sz[(size_t)i].assign(sizes, sizes + d);
        sz[(size_t)i][0] = (int32_t)(hi[(size_t)i] - lo[(size_t)i]);
    }
    CHK(run_all(mg, [&](int i) { return hi[(size_t)i] > lo[(size_t)i]; }, [&](int i) {
        return gpbo_set_candidates_lattice(mg->ctx[(size_t)i], ax[(size_t)i].data(), sz[(size_t)i].data(), d);
    }));
    for (int i = 0; i < N; ++i)
        if (hi[(size_t)i] <= lo[(size_t)i]) mg->ctx[(size_t)i]->xs_rows = 0;      // an empty block holds no candidates
    mg->xoff = off;
    return GPBO_OK;
}

int gpbo_multi_posterior(gpbo_multi* mg, double* mu_max_out) {
    if (!mg) return GPBO_ERR_BAD_ARG;
    return multi_posterior(mg, mu_max_out);
}

int gpbo_multi_acquire(gpbo_multi* mg, int32_t acq, double kappa, int32_t incumbent_mode, double incumbent, int64_t* argmax_out,
                       double* best_out) {
    if (!mg) return GPBO_ERR_BAD_ARG;
    return multi_acquire(mg, acq, kappa, incumbent_mode, incumbent, argmax_out, best_out);
}

int gpbo_multi_first_at_least(gpbo_multi* mg, double threshold, int64_t* index_out) {
    if (!mg || !index_out) return GPBO_ERR_BAD_ARG;
    const int N = (int)mg->ctx.size();
    if (mg->poff[(size_t)N] <= 0) MBADARG("first_at_least: no posterior");
    std::vector<int64_t> li((size_t)N, -1);
    auto use = [&](int i) { return mg->poff[(size_t)i + 1] > mg->poff[(size_t)i]; };
    CHK(run_all(mg, use, [&](int i) { return gpbo_first_at_least(mg->ctx[(size_t)i], threshold, &li[(size_t)i]); }));
    *index_out = -1;
    for (int i = 0; i < N; ++i)
        if (use(i) && li[(size_t)i] >= 0) {
            *index_out = mg->poff[(size_t)i] + li[(size_t)i];      // blocks are in index order: the first hit is the lowest
            break;
        }
    return GPBO_OK;
}

int gpbo_multi_read(gpbo_multi* mg, double* mu, double* var, double* acq, int64_t offset, int64_t count) {
    if (!mg) return GPBO_ERR_BAD_ARG;
    const int N = (int)mg->ctx.size();
    if (offset < 0 || count < 0 || offset + count > mg->poff[(size_t)N]) MBADARG("read: range outside the candidate set");
    auto lo_of = [&](int i) { return std::max(offset, mg->poff[(size_t)i]); };
    auto hi_of = [&](int i) { return std::min(offset + count, mg->poff[(size_t)i + 1]); };
    return run_all(mg, [&](int i) { return hi_of(i) > lo_of(i); }, [&](int i) {
        const int64_t lo = lo_of(i), cnt = hi_of(i) - lo, dst = lo - offset;
        return gpbo_read(mg->ctx[(size_t)i], mu ? mu + dst : nullptr, var ? var + dst : nullptr, acq ? acq + dst : nullptr,
                         lo - mg->poff[(size_t)i], cnt);
    });
}

int gpbo_multi_set_posterior(gpbo_multi* mg, const double* mu, const double* var, int64_t m) {
    if (!mg) return GPBO_ERR_BAD_ARG;
    if (!mu || !var || m <= 0) MBADARG("set_posterior: empty input");
    const int N = (int)mg->ctx.size();
    std::vector<int64_t> off;
    make_offsets(off, m, N);
    mg->poff.assign((size_t)N + 1, 0);
    auto use = [&](int i) { return off[(size_t)i + 1] > off[(size_t)i]; };
    CHK(run_all(mg, use, [&](int i) {
        return gpbo_set_posterior(mg->ctx[(size_t)i], mu + off[(size_t)i], var + off[(size_t)i], off[(size_t)i + 1] - off[(size_t)i]);
    }));
    double g = -INFINITY;
    for (int64_t j = 0; j < m; ++j) g = host_max_nan(g, mu[j]);
    mg->mu_max = g;
    mg->poff = off;
    return GPBO_OK;
}

int gpbo_multi_predict_ei(gpbo_multi* mg, const double* Xs, int64_t m, double kappa, int32_t incumbent_mode, double incumbent,
                          double* mu_out, double* var_out, double* ei_out, int64_t* argmax_out, double* ei_max_out,
                          double* mu_max_out) {
    if (!mg) return GPBO_ERR_BAD_ARG;
    CHK(gpbo_multi_set_candidates(mg, Xs, m, 0));
    CHK(multi_posterior(mg, mu_max_out));
    CHK(multi_acquire(mg, GPBO_ACQ_EI, kappa, incumbent_mode, incumbent, argmax_out, ei_max_out));
    return gpbo_multi_read(mg, mu_out, var_out, ei_out, 0, m);
}

int gpbo_multi_append(gpbo_multi* mg, const double* x_new, double y_new, double* mu_max_out) {
    if (!mg) return GPBO_ERR_BAD_ARG;
    const int N = (int)mg->ctx.size();
    if (!mg->replicated) MBADARG("append: call gpbo_multi_reserve_appends before a single-factor gpbo_multi_fit");
    for (int i = 0; i < N; ++i)
        if (mg->poff[(size_t)i + 1] <= mg->poff[(size_t)i]) MBADARG("append: fewer candidates than contexts, or no posterior yet");
    std::vector<double> mm((size_t)N, -INFINITY);
    const int st = run_all(mg, every, [&](int i) { return gpbo_append(mg->ctx[(size_t)i], x_new, y_new, &mm[(size_t)i]); });
    if (st != GPBO_OK) {
        mg->fitted = mg->ctx[0]->fitted;
        return st;
    }
    double g = -INFINITY;
    for (int i = 0; i < N; ++i) g = host_max_nan(g, mm[(size_t)i]);
    mg->mu_max = g;
    if (mu_max_out) *mu_max_out = g;
    return GPBO_OK;
}

int gpbo_multi_append_acquire(gpbo_multi* mg, const double* x_new, double y_new, int32_t acq, double kappa, int32_t incumbent_mode,
                              double incumbent, double* mu_max_out, int64_t* argmax_out, double* best_out) {
    if (!mg) return GPBO_ERR_BAD_ARG;
    if (mg->ctx.size() == 1) {      // one block: the incumbent needs no exchange, the fused single-context call does both halves
        if (!mg->replicated) MBADARG("append: call gpbo_multi_reserve_appends before a single-factor gpbo_multi_fit");
        const int st = gpbo_append_acquire(mg->ctx[0], x_new, y_new, acq, kappa, incumbent_mode, incumbent, &mg->mu_max, argmax_out, best_out);
        if (st != GPBO_OK) {
            mg->err = mg->ctx[0]->err;
            mg->info = mg->ctx[0]->info;
            mg->fitted = mg->ctx[0]->fitted;
            return st;
        }
        if (mu_max_out) *mu_max_out = mg->mu_max;
        return GPBO_OK;
    }
    CHK(gpbo_multi_append(mg, x_new, y_new, mu_max_out));
    return multi_acquire(mg, acq, kappa, incumbent_mode, incumbent, argmax_out, best_out);
}

int gpbo_multi_lml_batch(gpbo_multi* mg, const double* X, const double* y, int64_t n, int32_t d, const double* thetas,
                         const double* sigmas, int32_t B, int32_t kernel, double* nll_out, double* grad_out, int32_t* status_out) {
    if (!mg) return GPBO_ERR_BAD_ARG;
    if (B <= 0 || !thetas || !sigmas || !nll_out) MBADARG("lml_batch: need B >= 1 and non-null arrays");
    const int N = (int)mg->ctx.size();
    std::vector<int64_t> off;
    make_offsets(off, B, N);      // the settings are independent: a contiguous block per context, nothing exchanged
    return run_all(mg, [&](int i) { return off[(size_t)i + 1] > off[(size_t)i]; }, [&](int i) {
        const int64_t lo = off[(size_t)i], cnt = off[(size_t)i + 1] - lo;
        return gpbo_lml_batch(mg->ctx[(size_t)i], X, y, n, d, thetas + lo * d, sigmas + lo, (int32_t)cnt, kernel, nll_out + lo,
                              grad_out ? grad_out + lo * d : nullptr, status_out ? status_out + lo : nullptr);
    });
}

int gpbo_multi_last_distribute(gpbo_multi* mg, double* ms_out, double* bytes_out) {
    if (!mg) return GPBO_ERR_BAD_ARG;
    if (ms_out) *ms_out = mg->distribute_ms;
    if (bytes_out) *bytes_out = mg->distribute_bytes;
    return GPBO_OK;
}

}  // extern "C"
