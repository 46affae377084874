/* A caller of libgpbo written in C99 against include/gpbo.h alone: the body of the reference's BO loop
 * (BayesOpt/BayesOpt_v4.ipynb:645-647: gprfcn -> expected_improvement -> argmax) and one likelihood value (mlefcn, :154-165)
 * through the C ABI, with no Python in the process.  Test infrastructure: tests/test_gpu_c_caller.py writes the inputs,
 * runs this program and compares what it wrote with the ctypes route on the same inputs.
 *
 *   bo_body <in.bin> <out.bin>
 * in.bin : int64 n, d, m; double sigmaY, kappa; then X (n*d), y (n), theta (d), Xs (m*d) as doubles
 * out.bin: int64 status, argmax; double theta1, nll, ei_max, mu_max, nll_sweep; then mu (m), var (m), ei (m)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "gpbo.h"

static void die(const char* what, gpbo_ctx* ctx) {
    fprintf(stderr, "bo_body: %s%s%s\n", what, ctx ? ": " : "", ctx ? gpbo_last_error(ctx) : "");
    exit(1);
}

static void* xmalloc(size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    if (!p) die("out of memory", NULL);
    return p;
}

int main(int argc, char** argv) {
    if (argc != 3) die("usage: bo_body in.bin out.bin", NULL);
    FILE* f = fopen(argv[1], "rb");
    if (!f) die("cannot open the input file", NULL);
    int64_t hdr[3];
    double par[2];
    if (fread(hdr, sizeof(int64_t), 3, f) != 3 || fread(par, sizeof(double), 2, f) != 2) die("short input header", NULL);
    const int64_t n = hdr[0], d = hdr[1], m = hdr[2];
    if (n < 1 || d < 1 || d > 32 || m < 1) die("bad sizes", NULL);
    double* X = xmalloc((size_t)(n * d) * sizeof(double));
    double* y = xmalloc((size_t)n * sizeof(double));
    double* theta = xmalloc((size_t)d * sizeof(double));
    double* Xs = xmalloc((size_t)(m * d) * sizeof(double));
    if (fread(X, sizeof(double), (size_t)(n * d), f) != (size_t)(n * d) || fread(y, sizeof(double), (size_t)n, f) != (size_t)n ||
        fread(theta, sizeof(double), (size_t)d, f) != (size_t)d || fread(Xs, sizeof(double), (size_t)(m * d), f) != (size_t)(m * d))
        die("short input body", NULL);
    fclose(f);

    gpbo_ctx* ctx = NULL;
    if (gpbo_create(0, &ctx) != GPBO_OK || !ctx) die("gpbo_create failed (no GPU?)", NULL);
    double theta1 = 0.0, nll = 0.0, ei_max = 0.0, mu_max = 0.0, nll_sweep = 0.0;
    int64_t argmax = -1;
    int32_t sweep_status = -1;
    double* mu = xmalloc((size_t)m * sizeof(double));
    double* var = xmalloc((size_t)m * sizeof(double));
    double* ei = xmalloc((size_t)m * sizeof(double));
    int st = gpbo_fit(ctx, X, y, n, (int32_t)d, theta, par[0], GPBO_KERNEL_RBF, GPBO_VARIANT_V4, &theta1, &nll);
    if (st == GPBO_OK)
        st = gpbo_predict_ei(ctx, Xs, m, par[1], GPBO_INCUMBENT_MAX_MU, 0.0, mu, var, ei, &argmax, &ei_max, &mu_max);
    if (st == GPBO_OK)
        st = gpbo_lml_batch(ctx, X, y, n, (int32_t)d, theta, &par[0], 1, GPBO_KERNEL_RBF, &nll_sweep, NULL, &sweep_status);
    if (st != GPBO_OK && st != GPBO_ERR_NOT_PD) die("library call failed", ctx);

    f = fopen(argv[2], "wb");
    if (!f) die("cannot open the output file", NULL);
    const int64_t head[2] = {st, argmax};
    const double sc[5] = {theta1, nll, ei_max, mu_max, nll_sweep};
    fwrite(head, sizeof(int64_t), 2, f);
    fwrite(sc, sizeof(double), 5, f);
    if (st == GPBO_OK) {
        fwrite(mu, sizeof(double), (size_t)m, f);
        fwrite(var, sizeof(double), (size_t)m, f);
        fwrite(ei, sizeof(double), (size_t)m, f);
    }
    fclose(f);
    gpbo_destroy(ctx);
    free(X); free(y); free(theta); free(Xs); free(mu); free(var); free(ei);
    return 0;
}
