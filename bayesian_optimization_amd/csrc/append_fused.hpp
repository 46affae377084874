// The factor update of an append as ONE launch for small models (n < APPEND_FUSED_MAX_N).
//
// The update is a chain -- k(x_i, x_new) for all i, l = W k, the scalars of the enlarged model, the new row of W and
// alpha -- and at a few hundred training points each link is a launch of 3-4 us with nothing to do but wait for the one
// before it.  Here workgroup 0 runs the links as phases separated by barriers (the same device functions the separate
// kernels call, the same arithmetic), and on the lattice route workgroups 1 .. add the new training point's row of the
// factor tables, which depends on the new point alone.  W's triangle is read twice by that one workgroup, hence the limit.
#pragma once

namespace gpbo {

constexpr int APPEND_FUSED_MAX_N = 512;
constexpr int APPEND_FUSED_THREADS = 1024;

struct AppendFusedArgs {
    // model
    double* Xts;
    int n, n_pad, d, kernel;
    double* xnew_out;
    int* info;
    double *kvec, *lvec, *zfw, *yv, *scal, *params, *L, *W;
    long ld;
    double y_new, eps;
    // new row of W and alpha in this launch (the lattice route; the streaming sweep's launch does them otherwise)
    int with_wrow;
    double *alpha1, *alpha;
    // lattice rows (workgroups 1 ..; rows_elems = 0: none)
    long rows_elems;
    const double* axes;
    double *E, *B;
    int kcap, smax;
    long ldb;
};

__global__ __launch_bounds__(APPEND_FUSED_THREADS) void k_append_fused(AppendFusedArgs a, PointArg p, LatticeDesc lat) {
    const int tid = threadIdx.x;
    if (blockIdx.x > 0) {
        auto x = [&](int t) { return p.x[t]; };
        lattice_row_at((long)(blockIdx.x - 1) * APPEND_FUSED_THREADS + tid, a.n, lat, a.axes, a.params, x, a.E, a.kcap, a.smax, a.B, a.ldb);
        return;
    }
    __shared__ double r1[256];
    __shared__ double r2[256];
    __shared__ double ss[4][8][33], sa[4][8][33];
    const int top = a.n_pad > MAX_D ? a.n_pad : MAX_D;
    for (int i = tid; i < top; i += APPEND_FUSED_THREADS) append_kvec_at(i, a.Xts, a.n, a.n_pad, a.d, p, a.xnew_out, a.info, a.kvec, a.kernel);
    __syncthreads();
    for (int row0 = tid >> 6; row0 < a.n; row0 += 64) append_lvec_rows4(row0, tid & 63, a.W, a.ld, a.n, a.kvec, a.lvec);
    __syncthreads();
    append_scalars_phase(tid, tid < 256, r1, r2, a.lvec, a.zfw, a.yv, a.n, a.y_new, a.eps, a.scal, a.params, a.L, a.W, a.ld, a.info);
    if (!a.with_wrow) return;
    __syncthreads();
    const int nblk = (a.n + 1 + 31) / 32, g = tid >> 8;
    for (int b0 = 0; b0 < nblk; b0 += 4) {
        const int blk = b0 + g < nblk ? b0 + g : -1;
        append_wrow_block(blk, tid & 255, ss[g], sa[g], a.W, a.ld, a.n, a.lvec, a.scal, a.zfw, a.alpha1, a.alpha);
        __syncthreads();      // (the shared partial sums are reused by the next pass)
    }
}

}  // namespace gpbo
