// fp64 MFMA tile engine for gfx950 (MI355X): one 256-thread workgroup computes a
// 128x128 output tile  acc(i,j) = sum_{k in [k_begin,k_end)} A(i,k) * B(k,j)
// with v_mfma_f64_16x16x4_f64.  Four waves, 2x2, each owning a 64x64 sub-tile as
// 4x4 MFMA tiles (64 accumulator doubles per lane).
//
// Operands are staged global -> registers -> LDS, 16 k-values per step, double
// buffered with one barrier per step; the LDS stores of the next tile are issued
// in the middle of the MFMA phase so only the barrier itself sits between steps.
//
// Layout flags: KCONTIG=true  -> element (x,k) at base[x*ld + k]  (row-major, k contiguous)
//               KCONTIG=false -> element (x,k) at base[k*ld + x]  (x contiguous)
// LDS image per operand (18,432 B either way):
//   KCONTIG=true : row-major [128 x][18]: 8 lanes fetch one row's 128 contiguous bytes and
//                  store them with ds_write_b128; the MFMA fragment read (lane l: x = l&15,
//                  k = l>>4) walks rows at pitch 18 doubles = 36 banks: the 16 rows of a lane
//                  group fall on 16 distinct even banks and the k+1 group on the odd ones.
//   KCONTIG=false: k-major [16 k][144]: one wave-load is one 1 KiB k-row; fragment reads take
//                  16 consecutive doubles per lane group, row pitch 1152 B == 128 mod 256.
// Both fragment reads are conflict-free ds_read_b64.  (A lane map that gathers 16 rows per
// wave-load costs ~13 % of the MFMA rate: the vector memory pipe, not latency, is the limit.)
// All tile origins, leading dimensions and k ranges must be multiples of 16 doubles
// (callers pad every matrix to a multiple of 128).
#pragma once
#include <hip/hip_runtime.h>

namespace gpbo {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int TM = 128;
constexpr int TN = 128;
constexpr int TK = 16;
constexpr int LDS_LD = 144;   // k-major image: doubles per k-row
constexpr int LDS_RK = 18;    // row-major image: doubles per x-row
constexpr int TILE_DOUBLES = TK * LDS_LD;            // == 128 * LDS_RK
constexpr int GEMM_LDS_DOUBLES = 4 * TILE_DOUBLES;   // 2 buffers x 2 operands = 73,728 B
constexpr int GEMM_THREADS = 256;
static_assert(TK * LDS_LD == 128 * LDS_RK, "both LDS images have the same size");

struct Stage {
    f64x2 v[4];
};

template <bool KCONTIG>
__device__ __forceinline__ void load_tile(Stage& s, const double* __restrict__ base, long ld, int k0, int tid) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int q = tid + 256 * p;
        if (KCONTIG) {
            const int x = q >> 3, kp = q & 7;        // 8 lanes = one row's 16 k = 128 contiguous bytes
            s.v[p] = *reinterpret_cast<const f64x2*>(base + (long)x * ld + k0 + 2 * kp);
        } else {
            const int k = q >> 6, c2 = q & 63;       // one wave-load = one 1 KiB k-row
            s.v[p] = *reinterpret_cast<const f64x2*>(base + (long)(k0 + k) * ld + 2 * c2);
        }
    }
}

template <bool KCONTIG>
__device__ __forceinline__ void store_tile(const Stage& s, double* lds, int tid) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int q = tid + 256 * p;
        if (KCONTIG) {
            const int x = q >> 3, kp = q & 7;
            *reinterpret_cast<f64x2*>(lds + x * LDS_RK + 2 * kp) = s.v[p];
        } else {
            const int k = q >> 6, c2 = q & 63;
            *reinterpret_cast<f64x2*>(lds + k * LDS_LD + 2 * c2) = s.v[p];
        }
    }
}

// Offset of this lane's fragment element for 16-row block t = 0 and k-substep 0, and the
// strides to the next 16-row block / next k-substep (4 k).
template <bool KCONTIG>
__device__ __forceinline__ int frag_base(int lane, int x0) {
    return KCONTIG ? (x0 + (lane & 15)) * LDS_RK + (lane >> 4) : x0 + (lane & 15) + (lane >> 4) * LDS_LD;
}
template <bool KCONTIG>
__device__ __forceinline__ constexpr int frag_tstride() { return KCONTIG ? 16 * LDS_RK : 16; }
template <bool KCONTIG>
__device__ __forceinline__ constexpr int frag_kstride() { return KCONTIG ? 4 : 4 * LDS_LD; }

// One k-step: global loads of the next tile (one behind each of the first eight MFMAs), 64 MFMAs out of the current
// LDS buffer, LDS stores of the next tile in the middle.  TRI: A(i, k) = 0 for k > tri_k0 + i (the diagonal block of a
// lower-triangular operand); a 16-row block of A whose last row lies left of a 4-wide k-substep contributes nothing and
// its four MFMAs are skipped (uniform over the wave).
template <bool A_KCONTIG, bool B_KCONTIG, bool TRI, int MODE>
__device__ __forceinline__ void gemm_k_step(const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb,
                                            int kn, int k_rel, int row0, Stage& sa, Stage& sb, const double* a_l,
                                            const double* b_l, double* nxt, f64x4 (&acc)[4][4], int tid) {
    load_tile<A_KCONTIG>(sa, A, lda, kn, tid);
    load_tile<B_KCONTIG>(sb, B, ldb, kn, tid);
    constexpr bool PRIO = MODE >= 1, PREF = MODE == 2;
    // PREF: the fragments of k-substep kk + 1 are requested before the MFMA burst of kk (the priority switches fence the
    // scheduler, so the reads stay in front of the burst and have a whole burst to land)
    double af[2][4], bf[2][4];
    if (PREF) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            af[0][t] = a_l[t * frag_tstride<A_KCONTIG>()];
            bf[0][t] = b_l[t * frag_tstride<B_KCONTIG>()];
        }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int kr = PREF ? kk + 1 : kk, slot = PREF ? (kk + 1) & 1 : kk & 1;
        if (kr < 4) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                af[slot][t] = a_l[kr * frag_kstride<A_KCONTIG>() + t * frag_tstride<A_KCONTIG>()];
                bf[slot][t] = b_l[kr * frag_kstride<B_KCONTIG>() + t * frag_tstride<B_KCONTIG>()];
            }
        }
        if (PRIO) __builtin_amdgcn_s_setprio(1);   // a wave in its MFMA burst goes ahead of waves issuing loads and stores
#pragma unroll
        for (int ta = 0; ta < 4; ++ta) {
            if (TRI && k_rel + kk * 4 > row0 + ta * 16 + 15) continue;
#pragma unroll
            for (int tb = 0; tb < 4; ++tb)
                acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk & 1][ta], bf[kk & 1][tb], acc[ta][tb], 0, 0, 0);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        if (kk == 0 && !TRI) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        if (kk == 2) {   // the global data has had three k-substeps to arrive
            store_tile<A_KCONTIG>(sa, nxt, tid);
            store_tile<B_KCONTIG>(sb, nxt + TILE_DOUBLES, tid);
        }
    }
}

// acc must be zero-initialised (or hold the value to accumulate onto) by the caller.
// The loads and LDS stores of the next tile are unconditional (the last step re-reads its own tile into the buffer
// nobody reads any more), so a k-step is one basic block.  (Refilling the register stage right after its LDS store, a
// whole k-step ahead, measured 10 % slower.)
// tri_k0 (multiple of 16; default: none): from this k on A is the diagonal block of a lower-triangular matrix.
// MODE 1 raises the wave priority over each MFMA burst: +2 % on the trailing update of the factorisation, a loss on the
// short tiles of the triangular inverse (MODE 0).  MODE 2 also requests the LDS fragments one k-substep ahead: +2 % on the
// candidate sweep with short row panels; without the priority switches (which fence the instruction scheduler) the
// compiler sinks the reads back to their use, and on the factorisation's short products it loses 4 %.
template <bool A_KCONTIG, bool B_KCONTIG, int MODE = 1>
__device__ __forceinline__ void gemm_tile_128(const double* __restrict__ A, long lda,
                                              const double* __restrict__ B, long ldb,
                                              int k_begin, int k_end, f64x4 (&acc)[4][4], double* lds,
                                              int tri_k0 = 0x7fffffff) {
    if (k_begin >= k_end) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;
    Stage sa, sb;
    load_tile<A_KCONTIG>(sa, A, lda, k_begin, tid);
    load_tile<B_KCONTIG>(sb, B, ldb, k_begin, tid);
    store_tile<A_KCONTIG>(sa, lds, tid);
    store_tile<B_KCONTIG>(sb, lds + TILE_DOUBLES, tid);
    __syncthreads();
    int buf = 0;
    const int a_off = frag_base<A_KCONTIG>(lane, wr * 64);
    const int b_off = TILE_DOUBLES + frag_base<B_KCONTIG>(lane, wc * 64);
    const int k_split = tri_k0 < k_end ? (tri_k0 > k_begin ? tri_k0 : k_begin) : k_end;
    int k0 = k_begin;
    for (; k0 < k_split; k0 += TK) {
        const int kn = (k0 + TK) < k_end ? k0 + TK : k0;
        gemm_k_step<A_KCONTIG, B_KCONTIG, false, MODE>(A, lda, B, ldb, kn, 0, 0, sa, sb, lds + buf * 2 * TILE_DOUBLES + a_off,
                                                 lds + buf * 2 * TILE_DOUBLES + b_off, lds + (buf ^ 1) * 2 * TILE_DOUBLES, acc, tid);
        __syncthreads();
        buf ^= 1;
    }
    for (; k0 < k_end; k0 += TK) {
        const int kn = (k0 + TK) < k_end ? k0 + TK : k0;
        gemm_k_step<A_KCONTIG, B_KCONTIG, true, MODE>(A, lda, B, ldb, kn, k0 - tri_k0, wr * 64, sa, sb, lds + buf * 2 * TILE_DOUBLES + a_off,
                                                lds + buf * 2 * TILE_DOUBLES + b_off, lds + (buf ^ 1) * 2 * TILE_DOUBLES, acc, tid);
        __syncthreads();
        buf ^= 1;
    }
}

// ---------------------------------------------------------------------------------------
// Half-height form: one 256-thread workgroup computes a 64 x 128 tile (waves 2 x 2, each 32 x 64 = 2 x 4 MFMA tiles), both
// operands row-major with k contiguous.  For products whose 128 x 128 tiles are too few to fill the chip (the trailing
// updates of a small factorisation): a 128 x 128 x 128 tile is 13.7 us of MFMA time on one CU whatever the grid, twice
// as many half tiles finish in half of that.  Per flop it moves 1.5x the operand bytes of the square tile, so it is not
// for large grids.  LDS: 2 x (64 x 18 + 2304) doubles = 55,296 B (the B image is 128 x 18 row-major or 16 x 144 k-major).
constexpr int HALF_M = 64;
constexpr int HALF_A_DOUBLES = HALF_M * LDS_RK;             // 1152
constexpr int HALF_BUF_DOUBLES = HALF_A_DOUBLES + TILE_DOUBLES;
constexpr int HALF_LDS_DOUBLES = 2 * HALF_BUF_DOUBLES;      // 6912 doubles

struct StageHalf {
    f64x2 a[2];
    f64x2 b[4];
};

// B_KCONTIG as in load_tile: true = B row-major with k contiguous, false = k-major (element (j, k) at B[k * ldb + j]).
template <bool B_KCONTIG>
__device__ __forceinline__ void load_half(StageHalf& s, const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb,
                                          int k0, int tid) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int q = tid + 256 * p;
        s.a[p] = *reinterpret_cast<const f64x2*>(A + (long)(q >> 3) * lda + k0 + 2 * (q & 7));
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int q = tid + 256 * p;
        if (B_KCONTIG) s.b[p] = *reinterpret_cast<const f64x2*>(B + (long)(q >> 3) * ldb + k0 + 2 * (q & 7));
        else s.b[p] = *reinterpret_cast<const f64x2*>(B + (long)(k0 + (q >> 6)) * ldb + 2 * (q & 63));
    }
}

template <bool B_KCONTIG>
__device__ __forceinline__ void store_half(const StageHalf& s, double* lds, int tid) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int q = tid + 256 * p;
        *reinterpret_cast<f64x2*>(lds + (q >> 3) * LDS_RK + 2 * (q & 7)) = s.a[p];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int q = tid + 256 * p;
        if (B_KCONTIG) *reinterpret_cast<f64x2*>(lds + HALF_A_DOUBLES + (q >> 3) * LDS_RK + 2 * (q & 7)) = s.b[p];
        else *reinterpret_cast<f64x2*>(lds + HALF_A_DOUBLES + (q >> 6) * LDS_LD + 2 * (q & 63)) = s.b[p];
    }
}

// acc[ta][tb][r] of this lane is element (row, col) of the 64 x 128 tile with
//   row = (w >> 1) * 32 + ta * 16 + (lane >> 4) + 4 r,  col = (w & 1) * 64 + tb * 16 + (lane & 15).
template <bool B_KCONTIG = true>
__device__ __forceinline__ void gemm_tile_64x128(const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb,
                                                 int k_begin, int k_end, f64x4 (&acc)[2][4], double* lds) {
    if (k_begin >= k_end) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;
    StageHalf st;
    load_half<B_KCONTIG>(st, A, lda, B, ldb, k_begin, tid);
    store_half<B_KCONTIG>(st, lds, tid);
    __syncthreads();
    int buf = 0;
    const int a_off = (wr * 32 + (lane & 15)) * LDS_RK + (lane >> 4);
    const int b_off = HALF_A_DOUBLES + frag_base<B_KCONTIG>(lane, wc * 64);
    for (int k0 = k_begin; k0 < k_end; k0 += TK) {
        const int kn = (k0 + TK) < k_end ? k0 + TK : k0;
        load_half<B_KCONTIG>(st, A, lda, B, ldb, kn, tid);
        const double* a_l = lds + buf * HALF_BUF_DOUBLES + a_off;
        const double* b_l = lds + buf * HALF_BUF_DOUBLES + b_off;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            double af[2], bf[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) af[t] = a_l[kk * 4 + t * 16 * LDS_RK];
#pragma unroll
            for (int t = 0; t < 4; ++t) bf[t] = b_l[kk * frag_kstride<B_KCONTIG>() + t * frag_tstride<B_KCONTIG>()];
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                for (int tb = 0; tb < 4; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ta], bf[tb], acc[ta][tb], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            if (kk == 2) store_half<B_KCONTIG>(st, lds + (buf ^ 1) * HALF_BUF_DOUBLES, tid);
        }
        __syncthreads();
        buf ^= 1;
    }
}

// ---------------------------------------------------------------------------------------
// Quarter form: one 256-thread workgroup computes a 64 x 64 tile (waves 2 x 2, each 32 x 32 = 2 x 2 MFMA tiles); A row-major
// with k contiguous, B either way.  For products with so few 128 x 128 tiles that even the half tiles leave CUs idle
// or give some of them two: a 64 x 64 x 128 tile is 3.4 us of MFMA time, and four workgroups fit a CU (LDS 38,912 B each).
// Per flop it moves twice the operand bytes of the square tile.  Same sums in the same order as the other forms.
constexpr int QUARTER_M = 64;
constexpr int QUARTER_A_DOUBLES = QUARTER_M * LDS_RK;         // 1152: row-major image [64 x][18]
constexpr int QUARTER_LD = 80;                                // k-major image of a 64-wide B tile: 640 B == 128 mod 256
constexpr int QUARTER_B_DOUBLES = TK * QUARTER_LD;            // 1280 (the row-major image needs 1152)
constexpr int QUARTER_BUF_DOUBLES = QUARTER_A_DOUBLES + QUARTER_B_DOUBLES;
constexpr int QUARTER_LDS_DOUBLES = 2 * QUARTER_BUF_DOUBLES;  // 4864 doubles = 38,912 B

struct StageQuarter {
    f64x2 a[2];
    f64x2 b[2];
};

// B_KCONTIG as in load_tile: true = B row-major with k contiguous, false = k-major (element (j, k) at B[k * ldb + j]).
template <bool B_KCONTIG>
__device__ __forceinline__ void load_quarter(StageQuarter& s, const double* __restrict__ A, long lda, const double* __restrict__ B,
                                             long ldb, int k0, int tid) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int q = tid + 256 * p;                          // 8 lanes = one row's 16 k = 128 contiguous bytes
        s.a[p] = *reinterpret_cast<const f64x2*>(A + (long)(q >> 3) * lda + k0 + 2 * (q & 7));
        if (B_KCONTIG) s.b[p] = *reinterpret_cast<const f64x2*>(B + (long)(q >> 3) * ldb + k0 + 2 * (q & 7));
        else s.b[p] = *reinterpret_cast<const f64x2*>(B + (long)(k0 + (q >> 5)) * ldb + 2 * (q & 31));   // 32 lanes = one 512 B k-row
    }
}

template <bool B_KCONTIG>
__device__ __forceinline__ void store_quarter(const StageQuarter& s, double* lds, int tid) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int q = tid + 256 * p;
        *reinterpret_cast<f64x2*>(lds + (q >> 3) * LDS_RK + 2 * (q & 7)) = s.a[p];
        if (B_KCONTIG) *reinterpret_cast<f64x2*>(lds + QUARTER_A_DOUBLES + (q >> 3) * LDS_RK + 2 * (q & 7)) = s.b[p];
        else *reinterpret_cast<f64x2*>(lds + QUARTER_A_DOUBLES + (q >> 5) * QUARTER_LD + 2 * (q & 31)) = s.b[p];
    }
}

// acc[ta][tb][r] of this lane is element (row, col) of the 64 x 64 tile with
//   row = (w >> 1) * 32 + ta * 16 + (lane >> 4) + 4 r,  col = (w & 1) * 32 + tb * 16 + (lane & 15).
template <bool B_KCONTIG = true>
__device__ __forceinline__ void gemm_tile_64x64(const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb,
                                                int k_begin, int k_end, f64x4 (&acc)[2][2], double* lds) {
    if (k_begin >= k_end) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;
    StageQuarter st;
    load_quarter<B_KCONTIG>(st, A, lda, B, ldb, k_begin, tid);
    store_quarter<B_KCONTIG>(st, lds, tid);
    __syncthreads();
    int buf = 0;
    const int a_off = (wr * 32 + (lane & 15)) * LDS_RK + (lane >> 4);
    const int b_off = QUARTER_A_DOUBLES + (B_KCONTIG ? (wc * 32 + (lane & 15)) * LDS_RK + (lane >> 4)
                                                     : wc * 32 + (lane & 15) + (lane >> 4) * QUARTER_LD);
    constexpr int b_kstride = B_KCONTIG ? 4 : 4 * QUARTER_LD, b_tstride = B_KCONTIG ? 16 * LDS_RK : 16;
    for (int k0 = k_begin; k0 < k_end; k0 += TK) {
        const int kn = (k0 + TK) < k_end ? k0 + TK : k0;
        load_quarter<B_KCONTIG>(st, A, lda, B, ldb, kn, tid);
        const double* a_l = lds + buf * QUARTER_BUF_DOUBLES + a_off;
        const double* b_l = lds + buf * QUARTER_BUF_DOUBLES + b_off;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            double af[2], bf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[t] = a_l[kk * 4 + t * 16 * LDS_RK];
                bf[t] = b_l[kk * b_kstride + t * b_tstride];
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                for (int tb = 0; tb < 2; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ta], bf[tb], acc[ta][tb], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            if (kk == 2) store_quarter<B_KCONTIG>(st, lds + (buf ^ 1) * QUARTER_BUF_DOUBLES, tid);
        }
        __syncthreads();
        buf ^= 1;
    }
}

// Accumulator element acc[ta][tb][r] of this lane is output (row, col) of the tile:
__device__ __forceinline__ int acc_row(int ta, int r) {
    const int tid = threadIdx.x;
    return ((tid >> 7) & 1) * 64 + ta * 16 + ((tid & 63) >> 4) + 4 * r;
}
__device__ __forceinline__ int acc_col(int tb) {
    const int tid = threadIdx.x;
    return ((tid >> 6) & 1) * 64 + tb * 16 + (tid & 15);
}

__device__ __forceinline__ void acc_zero(f64x4 (&acc)[4][4]) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
}

}  // namespace gpbo
