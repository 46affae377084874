// Experiment harness: variants of the fp64 MFMA tile loop on a 4096^3 NN GEMM (A row-major k-contiguous, B row-major).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr int LDS_LD = 144;

template <int TK> struct Stage { f64x2 a[TK / 4]; f64x2 b[TK / 4]; };

template <int TK>
__device__ __forceinline__ void load_stage(Stage<TK>& s, const double* A, long lda, const double* B, long ldb, int k0, int tid) {
#pragma unroll
    for (int p = 0; p < TK / 4; ++p) {
        const int q = tid + 256 * p;
        const int l = q & 63, wb = q >> 6;
        const int x = (l & 15) + 16 * (wb & 7);
        const int kp = (l >> 4) + 4 * (wb >> 3);
        s.a[p] = *reinterpret_cast<const f64x2*>(A + (long)x * lda + k0 + 2 * kp);
        const int k = q >> 6, c2 = q & 63;
        s.b[p] = *reinterpret_cast<const f64x2*>(B + (long)(k0 + k) * ldb + 2 * c2);
    }
}
template <int TK>
__device__ __forceinline__ void store_stage(const Stage<TK>& s, double* la, double* lb, int tid) {
#pragma unroll
    for (int p = 0; p < TK / 4; ++p) {
        const int q = tid + 256 * p;
        const int l = q & 63, wb = q >> 6;
        const int x = (l & 15) + 16 * (wb & 7);
        const int kp = (l >> 4) + 4 * (wb >> 3);
        la[(2 * kp) * LDS_LD + x] = s.a[p].x;
        la[(2 * kp + 1) * LDS_LD + x] = s.a[p].y;
        const int k = q >> 6, c2 = q & 63;
        *reinterpret_cast<f64x2*>(lb + k * LDS_LD + 2 * c2) = s.b[p];
    }
}

// MODE 0: baseline (stores after all MFMAs). MODE 1: setprio around MFMAs. MODE 2: stores placed mid-phase.
template <int TK, int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void k_gemm(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, int n) {
    __shared__ double lds[4 * TK * LDS_LD];
    constexpr int TILE = TK * LDS_LD;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
    const double* Ap = A + (long)blockIdx.y * 128 * n;
    const double* Bp = B + (long)blockIdx.x * 128;
    f64x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f64x4{0, 0, 0, 0};
    Stage<TK> st;
    load_stage<TK>(st, Ap, n, Bp, n, 0, tid);
    store_stage<TK>(st, lds, lds + TILE, tid);
    __syncthreads();
    int buf = 0;
    const int frag_off = (lane & 15) + (lane >> 4) * LDS_LD;
    for (int k0 = 0; k0 < n; k0 += TK) {
        const bool has_next = (k0 + TK) < n;
        if (has_next && MODE != 4 && MODE != 5) load_stage<TK>(st, Ap, n, Bp, n, k0 + TK, tid);
        const double* a_l = lds + buf * 2 * TILE + wr * 64 + frag_off;
        const double* b_l = lds + buf * 2 * TILE + TILE + wc * 64 + frag_off;
        if (MODE == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < TK / 4; ++kk) {
            double af[4], bf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                af[t] = a_l[kk * 4 * LDS_LD + t * 16];
                bf[t] = b_l[kk * 4 * LDS_LD + t * 16];
            }
#pragma unroll
            for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                for (int tb = 0; tb < 4; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ta], bf[tb], acc[ta][tb], 0, 0, 0);
            if (MODE == 2 && kk == TK / 4 - 2 && has_next) {
                double* nxt = lds + (buf ^ 1) * 2 * TILE;
                store_stage<TK>(st, nxt, nxt + TILE, tid);
            }
        }
        if (MODE == 1) __builtin_amdgcn_s_setprio(0);
        if (MODE != 2 && MODE != 4 && MODE != 5 && MODE != 6 && has_next) {
            double* nxt = lds + (buf ^ 1) * 2 * TILE;
            store_stage<TK>(st, nxt, nxt + TILE, tid);
        }
        if (MODE == 6) asm volatile("" :: "v"(st.a[0].x), "v"(st.a[1].x), "v"(st.a[2].x), "v"(st.a[3].x), "v"(st.b[0].x), "v"(st.b[1].x), "v"(st.b[2].x), "v"(st.b[3].x));
        if (MODE != 4) __syncthreads();
        if (MODE != 4 && MODE != 5 && MODE != 6) buf ^= 1;
    }
    double* Ct = C + (long)blockIdx.y * 128 * n + (long)blockIdx.x * 128;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Ct[(long)(wr * 64 + ta * 16 + (lane >> 4) + 4 * r) * n + wc * 64 + tb * 16 + (lane & 15)] = acc[ta][tb][r];
}

template <int TK, int MODE, int WPS>
void run(const char* name, const double* A, const double* B, double* C, int n) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    dim3 grid(n / 128, n / 128);
    k_gemm<TK, MODE, WPS><<<grid, 256>>>(A, B, C, n);
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
        (void)hipEventRecord(e0);
        k_gemm<TK, MODE, WPS><<<grid, 256>>>(A, B, C, n);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    double sum = 0;
    std::vector<double> h(1024);
    (void)hipMemcpy(h.data(), C, 1024 * 8, hipMemcpyDeviceToHost);
    for (double v : h) sum += v;
    printf("%-28s TK=%d MODE=%d WPS=%d: %.3f ms  %.2f TFLOP/s  (chk %.6e)\n", name, TK, MODE, WPS, best, 2.0 * n * n * n / (best * 1e-3) / 1e12, sum);
}

// MODE 3: the k-step boundary (LDS stores of the next tile, barrier, first fragment reads of the next buffer, global
// loads of the tile after) sits between "fragments of the last k-substep are in registers" and "their MFMAs are issued",
// so 16 MFMAs (1024 cycles) cover the boundary.
template <int WPS>
__global__ __launch_bounds__(256, WPS) void k_gemm3(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, int n) {
    constexpr int TK = 16;
    __shared__ double lds[4 * TK * LDS_LD];
    constexpr int TILE = TK * LDS_LD;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
    const double* Ap = A + (long)blockIdx.y * 128 * n;
    const double* Bp = B + (long)blockIdx.x * 128;
    f64x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f64x4{0, 0, 0, 0};
    Stage<TK> st;
    load_stage<TK>(st, Ap, n, Bp, n, 0, tid);
    store_stage<TK>(st, lds, lds + TILE, tid);
    __syncthreads();
    if (TK < n) load_stage<TK>(st, Ap, n, Bp, n, TK, tid);
    const int frag_off = (lane & 15) + (lane >> 4) * LDS_LD;
    int buf = 0;
    double ca[4], cb[4];
    {
        const double* a_l = lds + wr * 64 + frag_off;
        const double* b_l = lds + TILE + wc * 64 + frag_off;
#pragma unroll
        for (int t = 0; t < 4; ++t) { ca[t] = a_l[t * 16]; cb[t] = b_l[t * 16]; }
    }
    for (int k0 = 0; k0 < n; k0 += TK) {
        const bool has_next = (k0 + TK) < n;
        const bool has_next2 = (k0 + 2 * TK) < n;
        const double* a_l = lds + buf * 2 * TILE + wr * 64 + frag_off;
        const double* b_l = lds + buf * 2 * TILE + TILE + wc * 64 + frag_off;
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            double na[4], nb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { na[t] = a_l[(kk + 1) * 4 * LDS_LD + t * 16]; nb[t] = b_l[(kk + 1) * 4 * LDS_LD + t * 16]; }
#pragma unroll
            for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                for (int tb = 0; tb < 4; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[ta], cb[tb], acc[ta][tb], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) { ca[t] = na[t]; cb[t] = nb[t]; }
        }
        double* nxt = lds + (buf ^ 1) * 2 * TILE;
        if (has_next) store_stage<TK>(st, nxt, nxt + TILE, tid);
        __syncthreads();
        if (has_next2) load_stage<TK>(st, Ap, n, Bp, n, k0 + 2 * TK, tid);
        double na[4], nb[4];
        if (has_next) {
            const double* a_n = nxt + wr * 64 + frag_off;
            const double* b_n = nxt + TILE + wc * 64 + frag_off;
#pragma unroll
            for (int t = 0; t < 4; ++t) { na[t] = a_n[t * 16]; nb[t] = b_n[t * 16]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ta = 0; ta < 4; ++ta)
#pragma unroll
            for (int tb = 0; tb < 4; ++tb)
                acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[ta], cb[tb], acc[ta][tb], 0, 0, 0);
        if (has_next) {
#pragma unroll
            for (int t = 0; t < 4; ++t) { ca[t] = na[t]; cb[t] = nb[t]; }
        }
        buf ^= 1;
    }
    double* Ct = C + (long)blockIdx.y * 128 * n + (long)blockIdx.x * 128;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Ct[(long)(wr * 64 + ta * 16 + (lane >> 4) + 4 * r) * n + wc * 64 + tb * 16 + (lane & 15)] = acc[ta][tb][r];
}

template <int WPS>
void run3(const char* name, const double* A, const double* B, double* C, int n) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    dim3 grid(n / 128, n / 128);
    k_gemm3<WPS><<<grid, 256>>>(A, B, C, n);
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
        (void)hipEventRecord(e0);
        k_gemm3<WPS><<<grid, 256>>>(A, B, C, n);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    double sum = 0;
    std::vector<double> h(1024);
    (void)hipMemcpy(h.data(), C, 1024 * 8, hipMemcpyDeviceToHost);
    for (double v : h) sum += v;
    printf("%-28s MODE=3 WPS=%d: %.3f ms  %.2f TFLOP/s  (chk %.6e)\n", name, WPS, best, 2.0 * n * n * n / (best * 1e-3) / 1e12, sum);
}

// MODE 7/8: A operand kept ROW-major in LDS ([128 rows][18 doubles]): global loads are 128 contiguous bytes per 8 lanes,
// stored with ds_write_b128, fragments read with ds_read_b64 at row pitch 18 (conflict-free).  lda = row pitch of A in
// global memory (n, or n + 16 to break the power-of-two stride).
constexpr int LDA_K = 18;
template <int WPS, bool MID>
__global__ __launch_bounds__(256, WPS) void k_gemm7(const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb,
                                                    double* __restrict__ C, int n) {
    constexpr int TK = 16;
    constexpr int TILE_A = 128 * LDA_K, TILE_B = TK * LDS_LD;
    __shared__ double lds[2 * (TILE_A + TILE_B)];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
    const double* Ap = A + (long)blockIdx.y * 128 * lda;
    const double* Bp = B + (long)blockIdx.x * 128;
    f64x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f64x4{0, 0, 0, 0};
    f64x2 sa[4], sb[4];
    auto load = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int q = tid + 256 * p;
            const int x = q >> 3, kp = q & 7;
            sa[p] = *reinterpret_cast<const f64x2*>(Ap + (long)x * lda + k0 + 2 * kp);
            const int k = q >> 6, c2 = q & 63;
            sb[p] = *reinterpret_cast<const f64x2*>(Bp + (long)(k0 + k) * ldb + 2 * c2);
        }
    };
    auto store = [&](double* la, double* lb) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int q = tid + 256 * p;
            const int x = q >> 3, kp = q & 7;
            *reinterpret_cast<f64x2*>(la + x * LDA_K + 2 * kp) = sa[p];
            const int k = q >> 6, c2 = q & 63;
            *reinterpret_cast<f64x2*>(lb + k * LDS_LD + 2 * c2) = sb[p];
        }
    };
    load(0);
    store(lds, lds + TILE_A);
    __syncthreads();
    int buf = 0;
    const int a_off = (wr * 64 + (lane & 15)) * LDA_K + (lane >> 4);
    const int b_off = wc * 64 + (lane & 15) + (lane >> 4) * LDS_LD;
    for (int k0 = 0; k0 < n; k0 += TK) {
        const bool has_next = (k0 + TK) < n;
        if (has_next) load(k0 + TK);
        const double* a_l = lds + buf * (TILE_A + TILE_B) + a_off;
        const double* b_l = lds + buf * (TILE_A + TILE_B) + TILE_A + b_off;
        double* nxt = lds + (buf ^ 1) * (TILE_A + TILE_B);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            double af[4], bf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                af[t] = a_l[t * 16 * LDA_K + kk * 4];
                bf[t] = b_l[kk * 4 * LDS_LD + t * 16];
            }
#pragma unroll
            for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                for (int tb = 0; tb < 4; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ta], bf[tb], acc[ta][tb], 0, 0, 0);
            if (MID && kk == 2 && has_next) store(nxt, nxt + TILE_A);
        }
        if (!MID && has_next) store(nxt, nxt + TILE_A);
        __syncthreads();
        buf ^= 1;
    }
    double* Ct = C + (long)blockIdx.y * 128 * n + (long)blockIdx.x * 128;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Ct[(long)(wr * 64 + ta * 16 + (lane >> 4) + 4 * r) * n + wc * 64 + tb * 16 + (lane & 15)] = acc[ta][tb][r];
}

template <int WPS, bool MID>
void run7(const char* name, const double* A, long lda, const double* B, long ldb, double* C, int n) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    dim3 grid(n / 128, n / 128);
    k_gemm7<WPS, MID><<<grid, 256>>>(A, lda, B, ldb, C, n);
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
        (void)hipEventRecord(e0);
        k_gemm7<WPS, MID><<<grid, 256>>>(A, lda, B, ldb, C, n);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    double sum = 0;
    std::vector<double> h(1024);
    (void)hipMemcpy(h.data(), C, 1024 * 8, hipMemcpyDeviceToHost);
    for (double v : h) sum += v;
    printf("%-28s rowmajorA WPS=%d MID=%d lda=%ld: %.3f ms  %.2f TFLOP/s  (chk %.6e)\n", name, WPS, (int)MID, lda, best,
           2.0 * n * n * n / (best * 1e-3) / 1e12, sum);
}

// Packed operands: every (tile, k-step) slab of 128 x 16 doubles is one contiguous 16 KB block in global memory
// (A: [ib][ks][x][16 k];  B: [jb][ks][16 k][128 j]), so each wave-load is 1 KiB contiguous for both operands.
template <int WPS>
__global__ __launch_bounds__(256, WPS) void k_gemm9(const double* __restrict__ A, const double* __restrict__ B,
                                                    double* __restrict__ C, int n) {
    constexpr int TK = 16;
    constexpr int TILE_A = 128 * LDA_K, TILE_B = TK * LDS_LD;
    __shared__ double lds[2 * (TILE_A + TILE_B)];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
    const double* Ap = A + (long)blockIdx.y * (n / 16) * 2048;
    const double* Bp = B + (long)blockIdx.x * (n / 16) * 2048;
    f64x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f64x4{0, 0, 0, 0};
    f64x2 sa[4], sb[4];
    auto load = [&](int ks) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int q = tid + 256 * p;
            sa[p] = *reinterpret_cast<const f64x2*>(Ap + (long)ks * 2048 + 2 * q);
            sb[p] = *reinterpret_cast<const f64x2*>(Bp + (long)ks * 2048 + 2 * q);
        }
    };
    auto store = [&](double* la, double* lb) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int q = tid + 256 * p;
            const int x = q >> 3, kp = q & 7;
            *reinterpret_cast<f64x2*>(la + x * LDA_K + 2 * kp) = sa[p];
            const int k = q >> 6, c2 = q & 63;
            *reinterpret_cast<f64x2*>(lb + k * LDS_LD + 2 * c2) = sb[p];
        }
    };
    load(0);
    store(lds, lds + TILE_A);
    __syncthreads();
    int buf = 0;
    const int a_off = (wr * 64 + (lane & 15)) * LDA_K + (lane >> 4);
    const int b_off = wc * 64 + (lane & 15) + (lane >> 4) * LDS_LD;
    const int nks = n / 16;
    for (int ks = 0; ks < nks; ++ks) {
        const bool has_next = (ks + 1) < nks;
        if (has_next) load(ks + 1);
        const double* a_l = lds + buf * (TILE_A + TILE_B) + a_off;
        const double* b_l = lds + buf * (TILE_A + TILE_B) + TILE_A + b_off;
        double* nxt = lds + (buf ^ 1) * (TILE_A + TILE_B);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            double af[4], bf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                af[t] = a_l[t * 16 * LDA_K + kk * 4];
                bf[t] = b_l[kk * 4 * LDS_LD + t * 16];
            }
#pragma unroll
            for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                for (int tb = 0; tb < 4; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ta], bf[tb], acc[ta][tb], 0, 0, 0);
            if (kk == 2 && has_next) store(nxt, nxt + TILE_A);
        }
        __syncthreads();
        buf ^= 1;
    }
    double* Ct = C + (long)blockIdx.y * 128 * n + (long)blockIdx.x * 128;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Ct[(long)(wr * 64 + ta * 16 + (lane >> 4) + 4 * r) * n + wc * 64 + tb * 16 + (lane & 15)] = acc[ta][tb][r];
}

template <int WPS>
void run9(const char* name, const double* A, const double* B, double* C, int n) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    dim3 grid(n / 128, n / 128);
    k_gemm9<WPS><<<grid, 256>>>(A, B, C, n);
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
        (void)hipEventRecord(e0);
        k_gemm9<WPS><<<grid, 256>>>(A, B, C, n);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    double sum = 0;
    std::vector<double> h(1024);
    (void)hipMemcpy(h.data(), C, 1024 * 8, hipMemcpyDeviceToHost);
    for (double v : h) sum += v;
    printf("%-28s packed WPS=%d: %.3f ms  %.2f TFLOP/s  (chk %.6e)\n", name, WPS, best, 2.0 * n * n * n / (best * 1e-3) / 1e12, sum);
}

// LDS-DMA staging (global_load_lds, 16 B per lane): no VGPR staging and no ds_write.  B image k-major [16][144]
// (one wave-instruction = one 1 KiB k-row); A image row-major UNPADDED [128][16] with the 16-byte chunks of row x
// XOR-swizzled by (x>>1)&7 (applied on the global source address and on the fragment read).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glob_void_t;
template <int WPS>
__global__ __launch_bounds__(256, WPS) void k_gemm_glds(const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb,
                                                        double* __restrict__ C, int n) {
    constexpr int TK = 16;
    constexpr int TILE_A = 128 * 16, TILE_B = TK * LDS_LD;
    __shared__ __attribute__((aligned(16))) double lds[2 * (TILE_A + TILE_B)];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
    const double* Ap = A + (long)blockIdx.y * 128 * lda;
    const double* Bp = B + (long)blockIdx.x * 128;
    f64x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f64x4{0, 0, 0, 0};
    auto stage = [&](int k0, double* la, double* lb) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            // A: wave w, pass p covers rows xb..xb+7
            const int xb = (w * 4 + p) * 8;
            const int x = xb + (lane >> 3), pc = lane & 7;
            const int lc = pc ^ ((x >> 1) & 7);
            const double* ga = Ap + (long)x * lda + k0 + 2 * lc;
            __builtin_amdgcn_global_load_lds((glob_void_t*)ga, (lds_void_t*)(la + xb * 16), 16, 0, 0);
            // B: k-row r
            const int r = w * 4 + p;
            const double* gb = Bp + (long)(k0 + r) * ldb + 2 * lane;
            __builtin_amdgcn_global_load_lds((glob_void_t*)gb, (lds_void_t*)(lb + r * LDS_LD), 16, 0, 0);
        }
    };
    stage(0, lds, lds + TILE_A);
    __syncthreads();
    int buf = 0;
    const int ax = wr * 64 + (lane & 15);
    const int b_off = wc * 64 + (lane & 15) + (lane >> 4) * LDS_LD;
    for (int k0 = 0; k0 < n; k0 += TK) {
        const bool has_next = (k0 + TK) < n;
        double* cur = lds + buf * (TILE_A + TILE_B);
        double* nxt = lds + (buf ^ 1) * (TILE_A + TILE_B);
        if (has_next) stage(k0 + TK, nxt, nxt + TILE_A);
        const double* b_l = cur + TILE_A + b_off;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            double af[4], bf[4];
            const int k = kk * 4 + (lane >> 4);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int x = ax + t * 16;
                af[t] = cur[x * 16 + (((k >> 1) ^ ((x >> 1) & 7)) << 1) + (k & 1)];
                bf[t] = b_l[kk * 4 * LDS_LD + t * 16];
            }
#pragma unroll
            for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                for (int tb = 0; tb < 4; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ta], bf[tb], acc[ta][tb], 0, 0, 0);
        }
        __syncthreads();
        buf ^= 1;
    }
    double* Ct = C + (long)blockIdx.y * 128 * n + (long)blockIdx.x * 128;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Ct[(long)(wr * 64 + ta * 16 + (lane >> 4) + 4 * r) * n + wc * 64 + tb * 16 + (lane & 15)] = acc[ta][tb][r];
}

template <int WPS>
void run_glds(const char* name, const double* A, const double* B, double* C, int n) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    dim3 grid(n / 128, n / 128);
    k_gemm_glds<WPS><<<grid, 256>>>(A, n, B, n, C, n);
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
        (void)hipEventRecord(e0);
        k_gemm_glds<WPS><<<grid, 256>>>(A, n, B, n, C, n);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    double sum = 0;
    std::vector<double> h(1024);
    (void)hipMemcpy(h.data(), C, 1024 * 8, hipMemcpyDeviceToHost);
    for (double v : h) sum += v;
    printf("%-28s glds WPS=%d: %.3f ms  %.2f TFLOP/s  (chk %.6e)\n", name, WPS, best, 2.0 * n * n * n / (best * 1e-3) / 1e12, sum);
}

// Producer/consumer split: waves 0-3 only read LDS and issue MFMAs; wave 4 issues every LDS-DMA load.
template <int WPS>
__global__ __launch_bounds__(320, WPS) void k_gemm_spec(const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb,
                                                        double* __restrict__ C, int n) {
    constexpr int TK = 16;
    constexpr int TILE_A = 128 * 16, TILE_B = TK * LDS_LD;
    __shared__ __attribute__((aligned(16))) double lds[2 * (TILE_A + TILE_B)];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double* Ap = A + (long)blockIdx.y * 128 * lda;
    const double* Bp = B + (long)blockIdx.x * 128;
    auto stage = [&](int k0, double* la, double* lb) {
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int xb = p * 8;
            const int x = xb + (lane >> 3), pc = lane & 7;
            const int lc = pc ^ ((x >> 1) & 7);
            const double* ga = Ap + (long)x * lda + k0 + 2 * lc;
            __builtin_amdgcn_global_load_lds((glob_void_t*)ga, (lds_void_t*)(la + xb * 16), 16, 0, 0);
            const double* gb = Bp + (long)(k0 + p) * ldb + 2 * lane;
            __builtin_amdgcn_global_load_lds((glob_void_t*)gb, (lds_void_t*)(lb + p * LDS_LD), 16, 0, 0);
        }
    };
    if (w == 4) {
        stage(0, lds, lds + TILE_A);
        __syncthreads();
        int buf = 0;
        for (int k0 = 0; k0 < n; k0 += TK) {
            double* nxt = lds + (buf ^ 1) * (TILE_A + TILE_B);
            if (k0 + TK < n) stage(k0 + TK, nxt, nxt + TILE_A);
            __syncthreads();
            buf ^= 1;
        }
        return;
    }
    const int wr = w >> 1, wc = w & 1;
    f64x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f64x4{0, 0, 0, 0};
    __syncthreads();
    int buf = 0;
    const int ax = wr * 64 + (lane & 15);
    const int b_off = wc * 64 + (lane & 15) + (lane >> 4) * LDS_LD;
    for (int k0 = 0; k0 < n; k0 += TK) {
        double* cur = lds + buf * (TILE_A + TILE_B);
        const double* b_l = cur + TILE_A + b_off;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            double af[4], bf[4];
            const int k = kk * 4 + (lane >> 4);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int x = ax + t * 16;
                af[t] = cur[x * 16 + (((k >> 1) ^ ((x >> 1) & 7)) << 1) + (k & 1)];
                bf[t] = b_l[kk * 4 * LDS_LD + t * 16];
            }
#pragma unroll
            for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                for (int tb = 0; tb < 4; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ta], bf[tb], acc[ta][tb], 0, 0, 0);
        }
        __syncthreads();
        buf ^= 1;
    }
    double* Ct = C + (long)blockIdx.y * 128 * n + (long)blockIdx.x * 128;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Ct[(long)(wr * 64 + ta * 16 + (lane >> 4) + 4 * r) * n + wc * 64 + tb * 16 + (lane & 15)] = acc[ta][tb][r];
}

template <int WPS>
void run_spec(const char* name, const double* A, const double* B, double* C, int n) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    dim3 grid(n / 128, n / 128);
    k_gemm_spec<WPS><<<grid, 320>>>(A, n, B, n, C, n);
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
        (void)hipEventRecord(e0);
        k_gemm_spec<WPS><<<grid, 320>>>(A, n, B, n, C, n);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    double sum = 0;
    std::vector<double> h(1024);
    (void)hipMemcpy(h.data(), C, 1024 * 8, hipMemcpyDeviceToHost);
    for (double v : h) sum += v;
    printf("%-28s spec WPS=%d: %.3f ms  %.2f TFLOP/s  (chk %.6e)\n", name, WPS, best, 2.0 * n * n * n / (best * 1e-3) / 1e12, sum);
}

int main() {
    const int n = 4096;
    std::vector<double> hA((size_t)n * n), hB((size_t)n * n);
    unsigned long long s = 1;
    for (auto& v : hA) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (double)(s >> 11) / 9007199254740992.0 - 0.5; }
    for (auto& v : hB) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (double)(s >> 11) / 9007199254740992.0 - 0.5; }
    double *A, *B, *C;
    (void)hipMalloc(&A, (size_t)n * n * 8); (void)hipMalloc(&B, (size_t)n * n * 8); (void)hipMalloc(&C, (size_t)n * n * 8);
    (void)hipMemcpy(A, hA.data(), (size_t)n * n * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(B, hB.data(), (size_t)n * n * 8, hipMemcpyHostToDevice);
    run<16, 0, 2>("baseline", A, B, C, n);
    run_glds<2>("LDS-DMA staging", A, B, C, n);
    run_spec<2>("loader wave + 4 MFMA waves", A, B, C, n);
    run_spec<1>("loader wave + 4 MFMA waves", A, B, C, n);
    run_glds<1>("LDS-DMA staging", A, B, C, n);
    run7<2, false>("A row-major in LDS", A, n, B, n, C, n);
    run7<1, false>("A row-major in LDS", A, n, B, n, C, n);
    run7<2, true>("A row-major in LDS", A, n, B, n, C, n);
    {
        // padded leading dimensions (n + 16)
        double *Ap2, *Bp2;
        (void)hipMalloc(&Ap2, (size_t)n * (n + 16) * 8); (void)hipMalloc(&Bp2, (size_t)n * (n + 16) * 8);
        (void)hipMemcpy2D(Ap2, (n + 16) * 8, A, n * 8, n * 8, n, hipMemcpyDeviceToDevice);
        (void)hipMemcpy2D(Bp2, (n + 16) * 8, B, n * 8, n * 8, n, hipMemcpyDeviceToDevice);
        run7<2, false>("padded ld", Ap2, n + 16, Bp2, n + 16, C, n);
        run7<1, false>("padded ld", Ap2, n + 16, Bp2, n + 16, C, n);
        run7<2, true>("padded ld", Ap2, n + 16, Bp2, n + 16, C, n);
    }
    run3<2>("boundary under MFMA", A, B, C, n);
    run3<1>("boundary under MFMA", A, B, C, n);
    run<16, 2, 1>("stores mid 1WG", A, B, C, n);
    {
        std::vector<double> pA((size_t)n * n), pB((size_t)n * n);
        for (int ib = 0; ib < n / 128; ++ib)
            for (int ks = 0; ks < n / 16; ++ks)
                for (int x = 0; x < 128; ++x)
                    for (int kk = 0; kk < 16; ++kk) {
                        pA[(((size_t)ib * (n / 16) + ks) * 128 + x) * 16 + kk] = hA[(size_t)(ib * 128 + x) * n + ks * 16 + kk];
                        pB[(((size_t)ib * (n / 16) + ks) * 16 + kk) * 128 + x] = hB[(size_t)(ks * 16 + kk) * n + ib * 128 + x];
                    }
        double *dpA, *dpB;
        (void)hipMalloc(&dpA, (size_t)n * n * 8); (void)hipMalloc(&dpB, (size_t)n * n * 8);
        (void)hipMemcpy(dpA, pA.data(), (size_t)n * n * 8, hipMemcpyHostToDevice);
        (void)hipMemcpy(dpB, pB.data(), (size_t)n * n * 8, hipMemcpyHostToDevice);
        run9<2>("packed tiles", dpA, dpB, C, n);
        run9<1>("packed tiles", dpA, dpB, C, n);
    }
    run<16, 4, 1>("LDS reads + MFMA only 1WG", A, B, C, n);
    run<16, 4, 2>("LDS reads + MFMA only 2WG", A, B, C, n);
    run<16, 5, 1>("+ barrier 1WG", A, B, C, n);
    run<16, 5, 2>("+ barrier 2WG", A, B, C, n);
    run<16, 6, 1>("+ global loads 1WG", A, B, C, n);
    run<16, 6, 2>("+ global loads 2WG", A, B, C, n);
    run<16, 1, 2>("setprio", A, B, C, n);
    run<16, 2, 2>("stores mid-phase", A, B, C, n);
    run<8, 0, 2>("TK=8", A, B, C, n);
    run<8, 2, 2>("TK=8 stores mid", A, B, C, n);
    run<16, 0, 1>("1 WG/CU", A, B, C, n);
    run<32, 0, 1>("TK=32 1 WG/CU", A, B, C, n);
    run<32, 2, 1>("TK=32 mid 1 WG/CU", A, B, C, n);
    return 0;
}
