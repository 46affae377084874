// Device kernels of libgpbo (gfx950), part of kernels.hpp: blocked Cholesky factorisation: diagonal block, row solve, trailing update.
#pragma once

namespace gpbo {

// ---------------------------------------------------------------------------------------

// Diagonal block p of the blocked factorisation: Cholesky factor of the 128x128 block and its inverse in one
// workgroup (1024 threads), both on a packed copy of the lower triangle in LDS (row r starts at r (r + 1) / 2), on the
// MFMA routines below (potrf128_mfma / trinv128_mfma), which the one-workgroup likelihood kernel uses as well.
// Writes L (lower, zeros above) back into A and inv(L) (lower, zeros above) into Dinv.
// info: first failing 1-based global row (pivot <= 0 or NaN), set once.
constexpr int POTRF_THREADS = 256;      // workgroup of the one-workgroup likelihood / descent routines for small n
__device__ __forceinline__ constexpr int tri_off(int r) { return r * (r + 1) / 2; }
constexpr int POTRF_LDS_PACKED = PB * (PB + 1) / 2;     // 8256 doubles
#ifdef GPBO_STAMPS
__device__ long long g_stamps[64];
#define GPBO_STAMP(i, acc) do { if (threadIdx.x == 0) { const long long t_ = wall_clock64(); if (acc) g_stamps[i] += t_ - g_stamps[63]; g_stamps[63] = t_; } } while (0)
#else
#define GPBO_STAMP(i, acc) do { } while (0)
#endif
constexpr int TRINV_T_DOUBLES = 64 * 65;

// ---------------------------------------------------------------------------------------
// MFMA form of the diagonal block (potrf128_mfma + trinv128_mfma): the block lives packed in LDS as before, but
//   * a 16x16 diagonal tile is factored AND inverted by one wave entirely in registers: lane i < 16 holds row i, the pivot
//     and the multipliers of other rows arrive through v_readlane (uniform, no LDS round trip, no barrier inside);
//     its inverse is built column-wise (lane c holds column c) from the same registers;
//   * the rows below a tile are multiplied by that inverse, and the trailing triangle is updated, on v_mfma_f64_16x16x4;
//   * the inverse of the whole block is assembled from the eight tile inverses by recursive doubling on MFMA.
// Fragment conventions of v_mfma_f64_16x16x4 (as in gemm_f64.hpp): operand A: lane l holds A[l & 15][l >> 4],
// operand B: B[l >> 4][l & 15], result: element r of lane l is D[(l >> 4) + 4 r][l & 15].
__device__ __forceinline__ double readlane_f64(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

constexpr int LEAF_P = 17;                        // pitch of a 16x16 tile inverse in LDS
constexpr int LEAF_DOUBLES = 16 * LEAF_P;

// Wave-level: Cholesky factor of the 16x16 tile at (c0, c0) of the packed block M, written back in place, and its
// inverse into Wi[r * LEAF_P + c].  Returns the first failing column (pivot <= 0 or NaN) or -1.
// The elimination runs on the 32x16 matrix [A; I]: lanes 0..15 hold the rows of A, lanes 16..31 the rows of the identity.
// A row below the diagonal tile leaves an elimination as row * inv(L)^T, so lane 16 + m ends with row m of inv(L)^T,
// i.e. column m of the inverse, from the very instructions that factor A.  Lanes >= 32 idle along.
__device__ __forceinline__ int leaf_potrf_inv(double* __restrict__ M, int c0, double* __restrict__ Wi) {
    const int lane = threadIdx.x & 63;
    int i = lane & 31;                             // row of [A; I]
    // opaque per call: otherwise the 32 lane masks (i > j, i == j) are hoisted out of the caller's loop over the tiles,
    // do not fit the scalar registers and come back through v_readlane from a spill register inside the pivot chain
    asm volatile("" : "+v"(i));
    double v[16];
    {
        // all sixteen reads are issued before the first is used (every lane reads a valid address; the empty asm keeps
        // the compiler from sinking the reads into the selects below, which would serialise sixteen LDS round trips)
        const int r = i & 15;
        const double* row = M + tri_off(c0 + r) + c0;
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = row[c <= r ? c : r];
#pragma unroll
        for (int c = 0; c < 16; ++c) asm volatile("" : "+v"(v[c]));
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = (i < 16) ? ((c <= i) ? v[c] : 0.0) : ((c == i - 16) ? 1.0 : 0.0);
    }
    int fail = -1;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double piv = readlane_f64(v[j], j);
        if (!(piv > 0.0) && fail < 0) fail = c0 + j;
        // 1/sqrt from the hardware estimate and two Newton steps, sqrt as piv * (1/sqrt) with one correction
        double rd = __builtin_amdgcn_rsq(piv);
        rd = rd * fma(-0.5 * piv * rd, rd, 1.5);
        rd = rd * fma(-0.5 * piv * rd, rd, 1.5);
        double d = piv * rd;
        d = fma(0.5 * rd, fma(-d, d, piv), d);
        const double lij = (i > j) ? v[j] * rd : 0.0;
        v[j] = (i == j) ? d : lij;                 // rows above the diagonal hold zero in this column
#pragma unroll
        for (int c = j + 1; c < 16; ++c) {
            const double lcj = readlane_f64(v[j], c);          // L[c][j], uniform
            v[c] = fma(-lij, lcj, v[c]);           // entries right of a row's diagonal pick up values nobody reads
        }
    }
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c <= i) M[tri_off(c0 + i) + c0 + c] = v[c];
    } else if (lane < 32) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Wi[r * LEAF_P + (i - 16)] = v[r];
    }
    return fail;
}

// Factor of the packed 128x128 block M in place; Wi8: [8][LEAF_DOUBLES] receives the inverses of the eight diagonal tiles.
// Per 16-column panel: the rows below the diagonal tile are multiplied by its inverse, then the trailing triangle is
// updated.  The next diagonal tile is updated first, by wave 0, which goes on to factor it while the other waves
// update the rest of the triangle: the tile factorisation (one wave, a chain of 16 pivots) is the critical path, and
// this way only the small row-scaling phase stands between two of them.  All threads must call it; ends on a barrier.
// ntl: number of leading 16x16 tile rows that hold data; from there on the block must be the identity pad (zero off the
// diagonal, one on it), which is left as it is and whose tile inverses are set to the identity.
__device__ __forceinline__ void potrf128_mfma(double* __restrict__ M, double* __restrict__ Wi8, int* __restrict__ failp,
                                              int ntl = 8) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    for (int idx = tid; idx < (8 - ntl) * 256; idx += blockDim.x) {
        const int b = ntl + (idx >> 8), r = (idx >> 4) & 15, c = idx & 15;
        Wi8[b * LEAF_DOUBLES + r * LEAF_P + c] = (r == c) ? 1.0 : 0.0;
    }
    auto leaf = [&](int jb) {
        const int fail = leaf_potrf_inv(M, 16 * jb, Wi8 + jb * LEAF_DOUBLES);
        if (lane == 0 && fail >= 0 && *failp < 0) *failp = fail;
    };
    GPBO_STAMP(0, 0);
    if (w == 0) leaf(0);
    GPBO_STAMP(1, 1);
    __syncthreads();
    for (int jb = 0; jb < ntl - 1; ++jb) {
        const int c0 = 16 * jb;
        const double* Wi = Wi8 + jb * LEAF_DOUBLES;
        const int nt = ntl - 1 - jb;                 // 16-row tiles below the diagonal tile (the pad has zeros there)
        GPBO_STAMP(0, 0);
        // rows below: X = A[rows, c0:c0+16] * inv(L11)^T, one tile per wave and turn
        for (int t = w; t < nt; t += nw) {
            const int r0 = c0 + 16 + 16 * t;
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double a = M[tri_off(r0 + li) + c0 + 4 * q + lk];
                const double b = Wi[li * LEAF_P + 4 * q + lk];          // B[k][j] = inv(L11)[j][k]
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) M[tri_off(r0 + lk + 4 * r) + c0 + li] = acc[r];
        }
        __syncthreads();
        GPBO_STAMP(2, 1);
        // trailing triangle: A[I, J] -= X_I X_J^T over the tile pairs I >= J; pair 0 is the next diagonal tile
        // With more than four waves, those that share wave 0's SIMD (w = 4, 8, ...) sit this phase out: the tile
        // factorisation is a chain of dependent instructions and runs faster with its SIMD to itself.
        const int npair = nt * (nt + 1) / 2;
        const bool spare = nw > 4;
        const int helpers = spare ? nw - nw / 4 : nw - 1;                // waves working on pairs 1 ..
        const int hidx = spare ? w - 1 - w / 4 : w - 1;                  // this wave's index among them
        const bool idle = spare && w != 0 && (w & 3) == 0;
        const int first = (w == 0) ? 0 : (idle ? npair : 1 + hidx), step = helpers > 0 ? helpers : 1;
        for (int pr = first; pr < npair; pr = (pr == 0 && nw > 1) ? npair : pr + step) {
            int ti, tj;
            tri_decode(pr, ti, tj);
            const int rI = c0 + 16 + 16 * ti, rJ = c0 + 16 + 16 * tj;
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double a = M[tri_off(rI + li) + c0 + 4 * q + lk];
                const double b = M[tri_off(rJ + li) + c0 + 4 * q + lk];  // B[k][j] = X_J[j][k]
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rI + lk + 4 * r, col = rJ + li;
                if (col <= row) M[tri_off(row) + col] -= acc[r];
            }
        }
        if (w == 0) {
            // the tile this wave just updated is read back by the same wave: LDS serves a wave's requests in order
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            GPBO_STAMP(0, 0);
            leaf(jb + 1);
            GPBO_STAMP(1, 1);
        }
        __syncthreads();
        GPBO_STAMP(3, 1);
    }
}

// One level of the block inverse on MFMA.  Before: the diagonal blocks of size S hold their inverses, the blocks below
// them still hold L.  Pair p joins A = [2pS, 2pS+S) and B = A + S:  W[B,A] = -W[B,B] (L[B,A] W[A,A]).
// ntl: tile rows from ntl on are the identity pad (see potrf128_mfma): their rows of W[B,A] are zero and stay untouched.
template <int S>
__device__ __forceinline__ void trinv_level_mfma(double* __restrict__ M, double* __restrict__ T, int ntl) {
    constexpr int P = PB / (2 * S), NT = S / 16, TP = S + 1;
    constexpr int TILES = P * NT * NT;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    for (int t = w; t < TILES; t += nw) {            // T = L[B,A] W[A,A]; W[A,A] lower: k-tiles kt >= jt
        const int p = t / (NT * NT), it = (t / NT) % NT, jt = t % NT;
        const int a0 = 2 * p * S, b0 = a0 + S;
        if (b0 + 16 * it >= 16 * ntl) continue;
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
        for (int kt = jt; kt < NT; ++kt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = 16 * kt + 4 * q + lk, j = 16 * jt + li;
                const double a = M[tri_off(b0 + 16 * it + li) + a0 + k];
                const double bv = M[tri_off(a0 + k) + a0 + (j <= k ? j : k)];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, (j <= k) ? bv : 0.0, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) T[p * S * TP + (16 * it + lk + 4 * r) * TP + 16 * jt + li] = acc[r];
    }
    __syncthreads();
    for (int t = w; t < TILES; t += nw) {            // W[B,A] = -W[B,B] T; W[B,B] lower: k-tiles kt <= it
        const int p = t / (NT * NT), it = (t / NT) % NT, jt = t % NT;
        const int a0 = 2 * p * S, b0 = a0 + S;
        if (b0 + 16 * it >= 16 * ntl) continue;
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
        for (int kt = 0; kt <= it; ++kt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = 16 * kt + 4 * q + lk, i = 16 * it + li;
                const double av = M[tri_off(b0 + i) + b0 + (k <= i ? k : i)];
                const double b = T[p * S * TP + k * TP + 16 * jt + li];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64((k <= i) ? av : 0.0, b, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) M[tri_off(b0 + 16 * it + lk + 4 * r) + a0 + 16 * jt + li] = -acc[r];
    }
    __syncthreads();
}

// Inverse of the packed lower-triangular block M in place, given the inverses of its eight diagonal tiles.
// T: 64 * 65 doubles.  All 256 threads; ends on a barrier.
__device__ __forceinline__ void trinv128_mfma(double* __restrict__ M, const double* __restrict__ Wi8, double* __restrict__ T,
                                              int ntl = 8) {
    const int tid = threadIdx.x;
    for (int idx = tid; idx < 8 * 256; idx += blockDim.x) {      // tile inverses into the diagonal tiles
        const int b = idx >> 8, r = (idx >> 4) & 15, c = idx & 15;
        if (c <= r) M[tri_off(16 * b + r) + 16 * b + c] = Wi8[b * LEAF_DOUBLES + r * LEAF_P + c];
    }
    __syncthreads();
    trinv_level_mfma<16>(M, T, ntl);
    trinv_level_mfma<32>(M, T, ntl);
    trinv_level_mfma<64>(M, T, ntl);
}

// 16-row tiles of diagonal block p that hold data when the matrix has n real rows (the rest of the block, padding up to
// a multiple of 128, is the identity and is skipped by the MFMA routines)
__device__ __forceinline__ int data_tiles(int n, int p) {
    const int t = (n - p * PB + 15) >> 4;
    return t < 1 ? 1 : (t > 8 ? 8 : t);
}

// defer: stop after the factor; the inverses of the eight 16x16 diagonal tiles go into the diagonal
// tiles of the block's Dinv (all k_trsm_rows needs), the inverse of the whole block is left to k_trinv_diag.
__device__ __forceinline__ void potrf_diag_body(double* __restrict__ A, long ld, int p, double* __restrict__ Dinv,
                                                int* __restrict__ info, double* __restrict__ S, bool defer, int ntl) {
    __shared__ int fail_s;
    const int tid = threadIdx.x;
    double* blk = A + (long)p * PB * ld + (long)p * PB;
    double* Tbuf = S + POTRF_LDS_PACKED;
    double* Wi8 = Tbuf + TRINV_T_DOUBLES;
    GPBO_STAMP(8, 0);
    if (tid == 0) fail_s = -1;
    // lower triangle -> packed LDS.  Eight requests per thread go out before the first is waited for (with 1024 threads
    // that is the whole block in one memory round trip; a load inside the "c <= r" branch is waited for where it is
    // issued, eight round trips in a row on the factorisation's critical path).  The pairs right of the diagonal are
    // read as well (they lie inside the block) and dropped.
    for (int base = 0; base < PB * PB / 2; base += 8 * (int)blockDim.x) {
        f64x2 v[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = base + it * (int)blockDim.x + tid;
            const int ii = idx < PB * PB / 2 ? idx : PB * PB / 2 - 1;
            v[it] = *reinterpret_cast<const f64x2*>(blk + (long)(ii >> 6) * ld + (ii & 63) * 2);
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = base + it * (int)blockDim.x + tid;
            const int r = idx >> 6, c = (idx & 63) * 2;
            if (idx < PB * PB / 2 && c <= r) {
                S[tri_off(r) + c] = v[it].x;
                if (c + 1 <= r) S[tri_off(r) + c + 1] = v[it].y;
            }
        }
    }
    __syncthreads();
    GPBO_STAMP(4, 1);
    potrf128_mfma(S, Wi8, &fail_s, ntl);
    GPBO_STAMP(5, 1);
    if (tid == 0 && fail_s >= 0) atomicCAS(info, 0, p * PB + fail_s + 1);
    for (int idx = tid; idx < PB * PB / 2; idx += blockDim.x) {      // L -> global, zeros above the diagonal
        const int r = idx >> 6, c = (idx & 63) * 2;
        f64x2 v;
        v.x = (c <= r) ? S[tri_off(r) + c] : 0.0;
        v.y = (c + 1 <= r) ? S[tri_off(r) + c + 1] : 0.0;
        *reinterpret_cast<f64x2*>(blk + (long)r * ld + c) = v;
    }
    double* dinv = Dinv + (long)p * PB * PB;
    if (defer) {
        for (int idx = tid; idx < 8 * 256; idx += blockDim.x) {
            const int b = idx >> 8, r = (idx >> 4) & 15, c = idx & 15;
            dinv[(16 * b + r) * PB + 16 * b + c] = (c <= r) ? Wi8[b * LEAF_DOUBLES + r * LEAF_P + c] : 0.0;
        }
        GPBO_STAMP(6, 1);
        return;
    }
    __syncthreads();
    GPBO_STAMP(6, 1);
    trinv128_mfma(S, Wi8, Tbuf, ntl);
    GPBO_STAMP(7, 1);
    for (int idx = tid; idx < PB * PB / 2; idx += blockDim.x) {
        const int r = idx >> 6, c = (idx & 63) * 2;
        f64x2 v;
        v.x = (c <= r) ? S[tri_off(r) + c] : 0.0;
        v.y = (c + 1 <= r) ? S[tri_off(r) + c + 1] : 0.0;
        *reinterpret_cast<f64x2*>(dinv + r * PB + c) = v;
    }
    GPBO_STAMP(9, 1);
}

// The LDS request (packed L, the product buffer of the inverse and the tile inverses: 116 KB) keeps every
// tile-engine workgroup (72 KB of 160) off the CU: the kernel is latency bound and runs 4x slower when its waves share
// SIMDs with MFMA waves.
constexpr int POTRF_MFMA_THREADS = 1024;        // 16 waves: the MFMA phases are LDS-latency bound, four waves per SIMD hide it
__global__ __launch_bounds__(POTRF_MFMA_THREADS) void k_potrf_diag(double* __restrict__ A, long ld, int p,
                                                              double* __restrict__ Dinv, int* __restrict__ info, BatchStride bs,
                                                              int defer, int n) {
    __shared__ double S[POTRF_LDS_PACKED + TRINV_T_DOUBLES + 8 * LEAF_DOUBLES];
    potrf_diag_body(A + blockIdx.z * bs.mat, ld, p, Dinv + blockIdx.z * bs.dinv, info + blockIdx.z * bs.slot * 2, S, defer != 0,
                    data_tiles(n, p));
}

// Panel solve: rows below diagonal block p, in place:  A[ib, p] <- A[ib, p] * inv(L_pp)^T
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_trsm_panel(double* __restrict__ A, long ld, int p,
                                                             const double* __restrict__ Dinv, BatchStride bs) {
    __shared__ double lds[GEMM_LDS_DOUBLES];
    A += blockIdx.z * bs.mat;
    Dinv += blockIdx.z * bs.dinv;
    __builtin_amdgcn_s_setprio(2);
    const int ib = p + 1 + blockIdx.x;
    double* Ap = A + (long)ib * PB * ld + (long)p * PB;
    f64x4 acc[4][4];
    acc_zero(acc);
    // (k,j) of the B operand = inv(L)[j][k]  -> k contiguous, ld 128
    gemm_tile_128<true, true>(Ap, ld, Dinv + (long)p * PB * PB, PB, 0, PB, acc, lds);
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ap[(long)acc_row(ta, r) * ld + acc_col(tb)] = acc[ta][tb][r];
}

// Panel solve without LDS, one wave per 16 rows of the panel: X L_pp^T = A[rows, p], solved as L_pp X^T = A^T by block
// forward substitution over the eight 16x16 tiles of L_pp,
//   X^T_j = inv(L_jj) (A^T_j - sum_{k<j} L_jk X^T_k),
// every operand fragment read straight from global memory (L2) into registers, all reads of a wave in flight at once.
// The 128-tile engine pays one memory round trip per 16 k-values, eight in a row at K = 128, and the panel solve sits on
// the factorisation's critical path with at most nb - 1 workgroups; here eight times as many SIMDs work, nothing waits
// on LDS or barriers, and only the inverses of the 16x16 diagonal tiles are needed (the diagonal tiles of Dinv), so the
// inverse of the whole block can be formed off the critical path.
// v_mfma_f64_16x16x4: A operand lane l = A[l & 15][kk], B operand = B[kk][l & 15], kk = l >> 4; result element r of lane
// l = D[(l >> 4) + 4 r][l & 15].  A result feeds the next product as B operand without leaving its registers: in
// sub-step r lane l offers row (l >> 4) + 4 r, so the A operand of that sub-step is read at k = (l >> 4) + 4 r (the
// order of the k-sum is free).  The tiles are kept negated (nx = -X^T) so that the sum accumulates onto A^T.
__global__ __launch_bounds__(64) void k_trsm_rows(double* __restrict__ A, long ld, int p, const double* __restrict__ Dinv,
                                                 BatchStride bs) {
    A += blockIdx.z * bs.mat;
    const double* Wd = Dinv + blockIdx.z * bs.dinv + (long)p * PB * PB;
    const int lane = threadIdx.x, li = lane & 15, lk = lane >> 4;
    const double* Lb = A + (long)p * PB * ld + (long)p * PB;
    double* rows = A + ((long)(p + 1) * PB + 16 * (long)blockIdx.x) * ld + (long)p * PB;
    f64x4 st[8];                                   // A^T tiles: element r = A[row li][16 j + lk + 4 r]
    double lf[28][4], wf[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            st[j][r] = rows[(long)li * ld + 16 * j + lk + 4 * r];
            wf[j][r] = Wd[(16 * j + li) * PB + 16 * j + lk + 4 * r];
        }
#pragma unroll
    for (int j = 1; j < 8; ++j)
#pragma unroll
        for (int k = 0; k < j; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) lf[j * (j - 1) / 2 + k][r] = Lb[(long)(16 * j + li) * ld + 16 * k + lk + 4 * r];
    f64x4 x[8], nx[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        f64x4 sacc = st[j];
#pragma unroll
        for (int k = 0; k < j; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                sacc = __builtin_amdgcn_mfma_f64_16x16x4f64(lf[j * (j - 1) / 2 + k][r], nx[k][r], sacc, 0, 0, 0);
        f64x4 xa = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) xa = __builtin_amdgcn_mfma_f64_16x16x4f64(wf[j][r], sacc[r], xa, 0, 0, 0);
        x[j] = xa;
        nx[j] = -xa;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) rows[(long)li * ld + 16 * j + lk + 4 * r] = x[j][r];
}

// Inverse of factored diagonal blocks p0 + blockIdx.x, from L (in A) and the tile inverses (diagonal tiles of Dinv),
// for the blocks whose factorisation kernel left it out (it is not needed by k_trsm_rows).
__global__ __launch_bounds__(1024) void k_trinv_diag(const double* __restrict__ A, long ld, int p0, double* __restrict__ Dinv,
                                                    BatchStride bs, int n) {
    __shared__ double S[POTRF_LDS_PACKED + TRINV_T_DOUBLES + 8 * LEAF_DOUBLES];
    const int tid = threadIdx.x, p = p0 + blockIdx.x;
    const double* blk = A + blockIdx.z * bs.mat + (long)p * PB * ld + (long)p * PB;
    double* dinv = Dinv + blockIdx.z * bs.dinv + (long)p * PB * PB;
    double* Tbuf = S + POTRF_LDS_PACKED;
    double* Wi8 = Tbuf + TRINV_T_DOUBLES;
    for (int idx = tid; idx < PB * PB / 2; idx += blockDim.x) {
        const int r = idx >> 6, c = (idx & 63) * 2;
        if (c <= r) {
            const f64x2 v = *reinterpret_cast<const f64x2*>(blk + (long)r * ld + c);
            S[tri_off(r) + c] = v.x;
            if (c + 1 <= r) S[tri_off(r) + c + 1] = v.y;
        }
    }
    for (int idx = tid; idx < 8 * 256; idx += blockDim.x) {
        const int b = idx >> 8, r = (idx >> 4) & 15, c = idx & 15;
        Wi8[b * LEAF_DOUBLES + r * LEAF_P + c] = dinv[(16 * b + r) * PB + 16 * b + c];
    }
    __syncthreads();
    trinv128_mfma(S, Wi8, Tbuf, data_tiles(n, p));
    for (int idx = tid; idx < PB * PB / 2; idx += blockDim.x) {
        const int r = idx >> 6, c = (idx & 63) * 2;
        f64x2 v;
        v.x = (c <= r) ? S[tri_off(r) + c] : 0.0;
        v.y = (c + 1 <= r) ? S[tri_off(r) + c + 1] : 0.0;
        *reinterpret_cast<f64x2*>(dinv + r * PB + c) = v;
    }
}

// Trailing update with a group of solved panels: A[ib, jb] -= P[ib] * P[jb]^T,
// P[x] = A[x, kcol0 : kcol0 + klen] (klen = 128 .. 512: several adjacent panels at once, so the
// read-modify-write of a tile of A is paid once per group instead of once per panel).
// TRI: 1-D grid over the lower tiles of the square of blocks starting at b0 (ib = b0 + i, jb = b0 + j, j <= i).
// !TRI: a rectangle of blocks, ib = b0 + blockIdx.y, jb = jb0 + blockIdx.x (every tile is computed).
template <bool TRI, int MODE = 1, int BAND = 0>
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_syrk_update(double* __restrict__ A, long ld, int b0, int jb0, int kcol0,
                                                              int klen, BatchStride bs) {
    A += blockIdx.z * bs.mat;
    int ib, jb;
    if (TRI) {
        int i, j;
        if constexpr (BAND > 0)
            band_decode<BAND>(blockIdx.x, jb0, i, j);       // jb0 carries the number of block rows here
        else
            tri_decode(blockIdx.x, i, j);
        ib = b0 + i;
        jb = b0 + j;
    } else {
        ib = b0 + blockIdx.y;
        jb = jb0 + blockIdx.x;
        __builtin_amdgcn_s_setprio(2);       // the rectangular form is only used on the factorisation's critical path
    }
    __shared__ double lds[GEMM_LDS_DOUBLES];
    const double* Pi = A + (long)ib * PB * ld + kcol0;
    const double* Pj = A + (long)jb * PB * ld + kcol0;
    // The accumulators start at -C and the tile is stored as -(acc): the read of C then travels with the first operand
    // loads instead of standing as a second memory round trip after the product (sign flips are exact).
    f64x4 acc[4][4];
    double* C = A + (long)ib * PB * ld + (long)jb * PB;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[ta][tb][r] = -C[(long)acc_row(ta, r) * ld + acc_col(tb)];
    gemm_tile_128<true, true, MODE>(Pi, ld, Pj, ld, 0, klen, acc, lds);
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(long)acc_row(ta, r) * ld + acc_col(tb)] = -acc[ta][tb][r];
}

// The same update on 64 x 128 half tiles (gemm_tile_64x128), for launches whose 128 x 128 tiles would leave most of the
// chip idle: twice the workgroups, each with half the MFMA time.  Grid: 2 x the tiles of k_syrk_update, blockIdx.x =
// 2 * tile + half (TRI) or blockIdx.x = 2 * column + half, blockIdx.y = row (!TRI).
template <bool TRI>
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_syrk_update_half(double* __restrict__ A, long ld, int b0, int jb0, int kcol0,
                                                                   int klen, BatchStride bs) {
    A += blockIdx.z * bs.mat;
    const int h = blockIdx.x & 1;
    int ib, jb;
    if (TRI) {
        int i, j;
        tri_decode(blockIdx.x >> 1, i, j);
        ib = b0 + i;
        jb = b0 + j;
    } else {
        ib = b0 + blockIdx.y;
        jb = jb0 + (blockIdx.x >> 1);
        __builtin_amdgcn_s_setprio(2);
    }
    __shared__ double lds[HALF_LDS_DOUBLES];
    const double* Pi = A + ((long)ib * PB + h * HALF_M) * ld + kcol0;
    const double* Pj = A + (long)jb * PB * ld + kcol0;
    double* C = A + ((long)ib * PB + h * HALF_M) * ld + (long)jb * PB;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int row0 = (w >> 1) * 32 + (lane >> 4), col0 = (w & 1) * 64 + (lane & 15);
    f64x4 acc[2][4];            // start at -C, store -(acc): see k_syrk_update
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[ta][tb][r] = -C[(long)(row0 + ta * 16 + 4 * r) * ld + col0 + tb * 16];
    gemm_tile_64x128<true>(Pi, ld, Pj, ld, 0, klen, acc, lds);
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(long)(row0 + ta * 16 + 4 * r) * ld + col0 + tb * 16] = -acc[ta][tb][r];
}

// The same update on 64 x 64 quarter tiles (gemm_tile_64x64), for launches with so few tiles that the half tiles still leave
// CUs idle or double up on some: four workgroups per 128 x 128 tile, blockIdx.x = 4 * tile + 2 * (row half) + (column half),
// over the lower tiles of the square of blocks starting at b0.
// !TRI: a rectangle of blocks, ib = b0 + blockIdx.y, jb = jb0 + (blockIdx.x >> 2) (the updates inside a group of panels).
template <bool TRI>
__global__ __launch_bounds__(GEMM_THREADS, 4) void k_syrk_update_quarter(double* __restrict__ A, long ld, int b0, int jb0, int kcol0,
                                                                         int klen, BatchStride bs) {
    A += blockIdx.z * bs.mat;
    const int hr = (blockIdx.x >> 1) & 1, hc = blockIdx.x & 1;
    int ib, jb;
    if (TRI) {
        int i, j;
        tri_decode(blockIdx.x >> 2, i, j);
        ib = b0 + i;
        jb = b0 + j;
    } else {
        ib = b0 + blockIdx.y;
        jb = jb0 + (blockIdx.x >> 2);
        __builtin_amdgcn_s_setprio(2);
    }
    __shared__ double lds[QUARTER_LDS_DOUBLES];
    const double* Pi = A + ((long)ib * PB + hr * QUARTER_M) * ld + kcol0;
    const double* Pj = A + ((long)jb * PB + hc * QUARTER_M) * ld + kcol0;
    double* C = A + ((long)ib * PB + hr * QUARTER_M) * ld + (long)jb * PB + hc * QUARTER_M;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int row0 = (w >> 1) * 32 + (lane >> 4), col0 = (w & 1) * 32 + (lane & 15);
    f64x4 acc[2][2];            // start at -C, store -(acc): see k_syrk_update
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[ta][tb][r] = -C[(long)(row0 + ta * 16 + 4 * r) * ld + col0 + tb * 16];
    gemm_tile_64x64(Pi, ld, Pj, ld, 0, klen, acc, lds);
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(long)(row0 + ta * 16 + 4 * r) * ld + col0 + tb * 16] = -acc[ta][tb][r];
}

// Zero the strict upper triangle (the factorisation leaves stale values there).
__global__ __launch_bounds__(256) void k_zero_upper(double* __restrict__ A, int n_pad, BatchStride bs) {
    A += blockIdx.z * bs.mat;
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i < n_pad && j < n_pad && j > i) A[(long)i * n_pad + j] = 0.0;
}

}  // namespace gpbo
