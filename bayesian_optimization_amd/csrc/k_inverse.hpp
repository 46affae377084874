// Device kernels of libgpbo (gfx950), part of kernels.hpp: triangular inverse W = L^-1 by recursive doubling.
#pragma once

namespace gpbo {

// ---------------------------------------------------------------------------------------
// Triangular inverse W = inv(L) by recursive doubling over 128-blocks.
// Level with half-size sb (in blocks); pair q joins A = blocks [a0, a0+sb) and
// B = blocks [b0, b0+rows) below it:  W[B,A] = -inv(L_BB) * L[B,A] * inv(L_AA).
__global__ __launch_bounds__(256) void k_inv_copy_diag(double* __restrict__ W, long ld, const double* __restrict__ Dinv,
                                                       BatchStride bs) {
    W += blockIdx.z * bs.mat;
    Dinv += blockIdx.z * bs.dinv;
    const int p = blockIdx.x;
    double* blk = W + (long)p * PB * ld + (long)p * PB;
    const double* src = Dinv + (long)p * PB * PB;
    // pairs of doubles, eight requests per thread in flight (one after the other the copy is a chain of 64 round trips)
    for (int base = 0; base < PB * PB / 2; base += 8 * 256) {
        f64x2 v[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) v[it] = *reinterpret_cast<const f64x2*>(src + 2 * (base + it * 256 + (int)threadIdx.x));
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = base + it * 256 + (int)threadIdx.x;
            *reinterpret_cast<f64x2*>(blk + (long)(idx >> 6) * ld + (idx & 63) * 2) = v[it];
        }
    }
}

// step 1: T[B,A] = L[B,A] * W[A,A]      (W[A,A] lower triangular: k >= j)
// HALF (both steps): 64 x 128 half tiles, blockIdx.x = 2 * (x index) + half, for the levels whose 128 x 128 tiles are too
// few to fill the chip (see gemm_tile_64x128); same sums in the same order.
template <bool HALF>
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_inv_step1(const double* __restrict__ L, const double* __restrict__ W,
                                                            double* __restrict__ T, long ld, int sb, int nb, BatchStride bs) {
    const int pairs = (nb + 2 * sb - 1) / (2 * sb);      // gridDim.z = pairs * slots
    const int q = blockIdx.z % pairs, slot = blockIdx.z / pairs;
    L += slot * bs.mat;
    W += slot * bs.mat;
    T += slot * bs.mat;
    const int a0 = 2 * q * sb, b0 = a0 + sb;
    const int rows = min(sb, nb - b0);
    const int ib = HALF ? blockIdx.x >> 1 : blockIdx.x, jb = blockIdx.y;      // jb slow: the long tiles (small jb) are dispatched first
    if (ib >= rows) return;
    if (HALF) {
        __shared__ double ldh[HALF_LDS_DOUBLES];
        const int h = blockIdx.x & 1;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const int row0 = (w >> 1) * 32 + (lane >> 4), col0 = (w & 1) * 64 + (lane & 15);
        f64x4 acc[2][4];
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = f64x4{0.0, 0.0, 0.0, 0.0};
        const double* Ap = L + ((long)(b0 + ib) * PB + h * HALF_M) * ld + (long)a0 * PB;
        const double* Bp = W + (long)a0 * PB * ld + (long)(a0 + jb) * PB;
        gemm_tile_64x128<false>(Ap, ld, Bp, ld, jb * PB, sb * PB, acc, ldh);
        double* C = T + ((long)(b0 + ib) * PB + h * HALF_M) * ld + (long)(a0 + jb) * PB;
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                for (int r = 0; r < 4; ++r) C[(long)(row0 + ta * 16 + 4 * r) * ld + col0 + tb * 16] = acc[ta][tb][r];
        return;
    }
    __shared__ double lds[GEMM_LDS_DOUBLES];
    f64x4 acc[4][4];
    acc_zero(acc);
    const double* Ap = L + (long)(b0 + ib) * PB * ld + (long)a0 * PB;
    const double* Bp = W + (long)a0 * PB * ld + (long)(a0 + jb) * PB;
    gemm_tile_128<true, false, 0>(Ap, ld, Bp, ld, jb * PB, sb * PB, acc, lds);
    double* C = T + (long)(b0 + ib) * PB * ld + (long)(a0 + jb) * PB;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(long)acc_row(ta, r) * ld + acc_col(tb)] = acc[ta][tb][r];
}

// step 2: W[B,A] = - W[B,B] * T[B,A]    (W[B,B] lower triangular: k <= i)
template <bool HALF>
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_inv_step2(double* __restrict__ W, const double* __restrict__ T, long ld,
                                                            int sb, int nb, BatchStride bs) {
    const int pairs = (nb + 2 * sb - 1) / (2 * sb);
    const int q = blockIdx.z % pairs, slot = blockIdx.z / pairs;
    W += slot * bs.mat;
    T += slot * bs.mat;
    const int a0 = 2 * q * sb, b0 = a0 + sb;
    const int rows = min(sb, nb - b0);
    const int ib = rows - 1 - (int)blockIdx.y, jb = HALF ? blockIdx.x >> 1 : blockIdx.x;      // long tiles (large ib) first
    if (ib < 0) return;
    if (HALF) {
        __shared__ double ldh[HALF_LDS_DOUBLES];
        const int h = blockIdx.x & 1;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const int row0 = (w >> 1) * 32 + (lane >> 4), col0 = (w & 1) * 64 + (lane & 15);
        f64x4 acc[2][4];
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = f64x4{0.0, 0.0, 0.0, 0.0};
        const double* Ap = W + ((long)(b0 + ib) * PB + h * HALF_M) * ld + (long)b0 * PB;
        const double* Bp = T + (long)b0 * PB * ld + (long)(a0 + jb) * PB;
        // W[B,B] is lower triangular: the upper half of the rows ends 64 columns earlier (those products are exact zeros)
        gemm_tile_64x128<false>(Ap, ld, Bp, ld, 0, ib * PB + (h + 1) * HALF_M, acc, ldh);
        double* C = W + ((long)(b0 + ib) * PB + h * HALF_M) * ld + (long)(a0 + jb) * PB;
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int tb = 0; tb < 4; ++tb)
#pragma unroll
                for (int r = 0; r < 4; ++r) C[(long)(row0 + ta * 16 + 4 * r) * ld + col0 + tb * 16] = -acc[ta][tb][r];
        return;
    }
    __shared__ double lds[GEMM_LDS_DOUBLES];
    f64x4 acc[4][4];
    acc_zero(acc);
    const double* Ap = W + (long)(b0 + ib) * PB * ld + (long)b0 * PB;
    const double* Bp = T + (long)b0 * PB * ld + (long)(a0 + jb) * PB;
    gemm_tile_128<true, false, 0>(Ap, ld, Bp, ld, 0, (ib + 1) * PB, acc, lds);
    double* C = W + (long)(b0 + ib) * PB * ld + (long)(a0 + jb) * PB;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(long)acc_row(ta, r) * ld + acc_col(tb)] = -acc[ta][tb][r];
}

// Both steps on 64 x 64 quarter tiles (gemm_tile_64x64, k-major second operand): blockIdx.x = 4 * (x index) + 2 * (row half)
// + (column half).  A level of the doubling is as long as its longest tile (the products shrink with the distance from the
// diagonal), and four small workgroups per CU even that out better than two halves; same sums in the same order.
__global__ __launch_bounds__(GEMM_THREADS, 4) void k_inv_step1_quarter(const double* __restrict__ L, const double* __restrict__ W,
                                                                       double* __restrict__ T, long ld, int sb, int nb, BatchStride bs) {
    const int pairs = (nb + 2 * sb - 1) / (2 * sb);
    const int q = blockIdx.z % pairs, slot = blockIdx.z / pairs;
    L += slot * bs.mat;
    W += slot * bs.mat;
    T += slot * bs.mat;
    const int a0 = 2 * q * sb, b0 = a0 + sb;
    const int rows = min(sb, nb - b0);
    const int ib = blockIdx.x >> 2, jb = blockIdx.y;
    if (ib >= rows) return;
    const int hr = (blockIdx.x >> 1) & 1, hc = blockIdx.x & 1;
    __shared__ double ldq[QUARTER_LDS_DOUBLES];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row0 = (w >> 1) * 32 + (lane >> 4), col0 = (w & 1) * 32 + (lane & 15);
    f64x4 acc[2][2];
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) acc[ta][tb] = f64x4{0.0, 0.0, 0.0, 0.0};
    const double* Ap = L + ((long)(b0 + ib) * PB + hr * QUARTER_M) * ld + (long)a0 * PB;
    const double* Bp = W + (long)a0 * PB * ld + (long)(a0 + jb) * PB + hc * QUARTER_M;
    gemm_tile_64x64<false>(Ap, ld, Bp, ld, jb * PB, sb * PB, acc, ldq);
    double* C = T + ((long)(b0 + ib) * PB + hr * QUARTER_M) * ld + (long)(a0 + jb) * PB + hc * QUARTER_M;
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(long)(row0 + ta * 16 + 4 * r) * ld + col0 + tb * 16] = acc[ta][tb][r];
}

__global__ __launch_bounds__(GEMM_THREADS, 4) void k_inv_step2_quarter(double* __restrict__ W, const double* __restrict__ T, long ld,
                                                                       int sb, int nb, BatchStride bs) {
    const int pairs = (nb + 2 * sb - 1) / (2 * sb);
    const int q = blockIdx.z % pairs, slot = blockIdx.z / pairs;
    W += slot * bs.mat;
    T += slot * bs.mat;
    const int a0 = 2 * q * sb, b0 = a0 + sb;
    const int rows = min(sb, nb - b0);
    const int ib = rows - 1 - (int)blockIdx.y, jb = blockIdx.x >> 2;      // long tiles (large ib) first
    if (ib < 0) return;
    const int hr = (blockIdx.x >> 1) & 1, hc = blockIdx.x & 1;
    __shared__ double ldq[QUARTER_LDS_DOUBLES];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row0 = (w >> 1) * 32 + (lane >> 4), col0 = (w & 1) * 32 + (lane & 15);
    f64x4 acc[2][2];
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) acc[ta][tb] = f64x4{0.0, 0.0, 0.0, 0.0};
    const double* Ap = W + ((long)(b0 + ib) * PB + hr * QUARTER_M) * ld + (long)b0 * PB;
    const double* Bp = T + (long)b0 * PB * ld + (long)(a0 + jb) * PB + hc * QUARTER_M;
    // W[B,B] is lower triangular: the upper half of the rows ends 64 columns earlier (those products are exact zeros)
    gemm_tile_64x64<false>(Ap, ld, Bp, ld, 0, ib * PB + (hr + 1) * QUARTER_M, acc, ldq);
    double* C = W + ((long)(b0 + ib) * PB + hr * QUARTER_M) * ld + (long)(a0 + jb) * PB + hc * QUARTER_M;
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(long)(row0 + ta * 16 + 4 * r) * ld + col0 + tb * 16] = -acc[ta][tb][r];
}

// Sums of N (a power of two, <= 32) per-lane values over the 64 lanes of a wave: on return every lane holds the total of
// value lane / (64 / N).  Each exchange halves the number of values a lane carries (the half it gives up travels to the
// partner lane): N - 1 exchanges plus a butterfly on the one value left, instead of N butterflies of six exchanges, and
// none of them waits for another value's turn.  The pairing tree (partners at distance 32, 16, 8, 4, 2, 1) and
// therefore every rounding is that of   for (o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o)   applied to each value.
template <int N, int H>
__device__ __forceinline__ void wave_sum_level(double (&s)[N], int lane) {
    // H values stay, partner distance 64 H / N; compile-time recursion keeps every index of s a constant (a loop over
    // the levels is not unrolled by the compiler, and s then turns into chains of selects)
    constexpr int dist = 64 * H / N;
    const bool up = (lane & dist) != 0;
    double give[H];
#pragma unroll
    for (int u = 0; u < H; ++u) {
        give[u] = up ? s[u] : s[u + H];
        s[u] = up ? s[u + H] : s[u];
    }
#pragma unroll
    for (int u = 0; u < H; ++u) give[u] = __shfl_xor(give[u], dist);     // all of a level's exchanges are in flight together
#pragma unroll
    for (int u = 0; u < H; ++u) s[u] += give[u];
    if constexpr (H > 1) wave_sum_level<N, H / 2>(s, lane);
}
template <int N>
__device__ __forceinline__ double wave_sum(double (&s)[N]) {
    static_assert(N >= 2 && N <= 32 && (N & (N - 1)) == 0, "a power of two between 2 and 32");
    const int lane = threadIdx.x & 63;
    wave_sum_level<N, N / 2>(s, lane);
    double r = s[0];
#pragma unroll
    for (int dist = 32 / N; dist >= 1; dist >>= 1) r += __shfl_xor(r, dist);
    return r;
}

}  // namespace gpbo
