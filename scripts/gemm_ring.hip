// Experiment: LDS ring of four k-step-8 buffers filled by LDS-DMA (global_load_lds), against the tile engine of
// csrc/gemm_f64.hpp, on a 4096^3 NN product (A row-major k-contiguous, B row-major).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/gemm_ring scripts/gemm_ring.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../bayesian_optimization_amd/csrc/gemm_f64.hpp"
using namespace gpbo;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glob_void_t;

template <int MODE>
__global__ __launch_bounds__(256, 2) void k_engine(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, int n) {
    __shared__ double lds[GEMM_LDS_DOUBLES];
    f64x4 acc[4][4];
    acc_zero(acc);
    gemm_tile_128<true, false, MODE>(A + (long)blockIdx.y * 128 * n, n, B + (long)blockIdx.x * 128, n, 0, n, acc, lds);
    double* Ct = C + (long)blockIdx.y * 128 * n + (long)blockIdx.x * 128;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ct[(long)acc_row(ta, r) * n + acc_col(tb)] = acc[ta][tb][r];
}

constexpr int RK = 8;                       // k per step
constexpr int RB_LD = 144;                  // B image: doubles per k-row
constexpr int R_A = 128 * RK;               // A image [128][8], 16-byte chunks of row x at position chunk ^ ((x >> 2) & 3)
constexpr int R_STAGE = R_A + RK * RB_LD;   // 2176 doubles = 17,408 B
constexpr int R_NB = 4;
#ifndef NXT_READS
#define NXT_READS 5
#endif

// vmcnt(v), no wait on lgkmcnt / expcnt (gfx9 encoding)
#define WAIT_VM(v) __builtin_amdgcn_s_waitcnt(((v) & 15) | (7 << 4) | (15 << 8) | (((v) >> 4) << 14))
// lgkmcnt(v), no wait on vmcnt / expcnt
#define WAIT_LGKM(v) __builtin_amdgcn_s_waitcnt(15 | (7 << 4) | ((v) << 8) | (3 << 14))

// ABL bit 0: no DMA loads in the loop (stale LDS), bit 1: no barrier in the loop, bit 2: no priority switches -- timing only
template <int ABL>
__global__ __launch_bounds__(256, 2) void k_ring(const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb,
                                                 double* __restrict__ C, int n) {
    __shared__ __attribute__((aligned(16))) double lds[R_NB * R_STAGE + 128];     // + the prefetch dump area
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), wr = w >> 1, wc = w & 1;
    const double* Ap = A + (long)blockIdx.y * 128 * lda;
    const double* Bp = B + (long)blockIdx.x * 128;
    f64x4 acc[4][4];
    acc_zero(acc);
    // this wave's four DMA instructions per k-step: A rows [32 w, 32 w + 32) in two instructions, B k-rows 2 w, 2 w + 1
    const int xa0 = w * 32 + (lane >> 2), xa1 = xa0 + 16;
    const int pc = lane & 3;
    const double* ga0 = Ap + (long)xa0 * lda + 2 * (pc ^ ((xa0 >> 2) & 3));
    const double* ga1 = Ap + (long)xa1 * lda + 2 * (pc ^ ((xa1 >> 2) & 3));
    const double* gb0 = Bp + (long)(2 * w) * ldb + 2 * lane;
    const double* gb1 = gb0 + ldb;
    auto issue = [&](int step) {
        double* st = lds + (step & (R_NB - 1)) * R_STAGE;
        const long ko = ((ABL & 255) & 8) ? 0 : (long)step * RK;     // bit 3: every step re-reads k-step 0 (hot in L2)
        __builtin_amdgcn_global_load_lds((glob_void_t*)(ga0 + ko), (lds_void_t*)(st + (w * 32) * RK), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glob_void_t*)(ga1 + ko), (lds_void_t*)(st + (w * 32 + 16) * RK), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glob_void_t*)(gb0 + ko * ldb), (lds_void_t*)(st + R_A + (2 * w) * RB_LD), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glob_void_t*)(gb1 + ko * ldb), (lds_void_t*)(st + R_A + (2 * w + 1) * RB_LD), 16, 0, 0);
    };
    const int g = lane >> 4, c = lane & 15;
    int aoff[2][4];          // A fragment offsets for k-substep kk and row block t
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int x = wr * 64 + t * 16 + c, k = kk * 4 + g;
            aoff[kk][t] = x * RK + (((k >> 1) ^ ((x >> 2) & 3)) << 1) + (k & 1);
        }
    const int boff = R_A + g * RB_LD + wc * 64 + c;
    auto readf = [&](const double* st, int kk, double (&af)[4], double (&bf)[4]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            af[t] = st[aoff[kk][t]];
            bf[t] = st[boff + kk * 4 * RB_LD + t * 16];
        }
    };
    constexpr int PF = ABL >> 8;      // software L2 prefetch distance in k-steps (0: off)
    // one touch per wave and k-step: waves 0/1 a 4-byte read from each of the 128 A rows' 64-byte segments, waves 2/3 from
    // the 64 128-byte lines of the 8 B rows
    const double* pf_ptr = (w < 2) ? Ap + (long)(w * 64 + lane) * lda : Bp + (long)(lane >> 3) * ldb + 16 * (lane & 7);
    const long pf_step = (w < 2) ? (long)RK : (long)RK * ldb;
    // the touch is an LDS-DMA load of 4 bytes per lane into a per-wave dump area: no register is written behind the compiler's back
    auto prefetch = [&](int step) {
        const double* q = pf_ptr + (long)step * pf_step;
        __builtin_amdgcn_global_load_lds((glob_void_t*)q, (lds_void_t*)(lds + R_NB * R_STAGE + w * 32), 4, 0, 0);
    };
    const int nsteps = n / RK;
    issue(0);
    issue(1);
    issue(2);
    WAIT_VM(4);                 // tiles 0 and 1 have landed
    __builtin_amdgcn_s_barrier();
    double a0[4], b0[4], a1[4], b1[4], an[4], bn[4];
    readf(lds, 0, a0, b0);
    for (int s = 0; s < nsteps; ++s) {
        const double* cur = lds + (s & (R_NB - 1)) * R_STAGE;
        const double* nx = lds + ((s + 1) & (R_NB - 1)) * R_STAGE;
        const bool more = s + 3 < nsteps;
        if (PF > 0 && s + PF < nsteps) prefetch(s + PF);     // issued before this step's loads: one more op may stay in flight
        if (more && !(ABL & 1)) issue(s + 3);
        // both fragment sets needed next are requested here, a whole burst before their first use: substep 1 of this
        // k-step and substep 0 of the next one (complete in LDS since the previous barrier; garbage past the last step)
        if (!(ABL & 4)) __builtin_amdgcn_s_setprio(1);
        readf(cur, 1, a1, b1);
        readf(nx, 0, an, bn);
#pragma unroll
        for (int ta = 0; ta < 4; ++ta)
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[ta], b0[tb], acc[ta][tb], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one MFMA, then every fragment read, then the rest of the burst
        __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 15, 0);
        if (!(ABL & 4)) __builtin_amdgcn_s_setprio(0);
        if (!(ABL & 4)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ta = 0; ta < 4; ++ta)
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[ta], b1[tb], acc[ta][tb], 0, 0, 0);
        if (!(ABL & 4)) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a0[t] = an[t];
            b0[t] = bn[t];
        }
        if (more && !(ABL & 1)) { if (PF > 0 && s + PF < nsteps) WAIT_VM(5); else WAIT_VM(4); }   // tile s + 2 has landed; tile s + 3 may still be in flight
        else WAIT_VM(0);
        if (!(ABL & 2)) __builtin_amdgcn_s_barrier();
    }
    double* Ct = C + (long)blockIdx.y * 128 * n + (long)blockIdx.x * 128;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ct[(long)acc_row(ta, r) * n + acc_col(tb)] = acc[ta][tb][r];
}

template <typename F>
void run(const char* name, F launch, double* C, int n, const std::vector<double>& ref) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    launch();
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
        (void)hipEventRecord(e0);
        launch();
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    std::vector<double> h((size_t)n * 4);
    (void)hipMemcpy(h.data(), C + (size_t)777 * n, h.size() * 8, hipMemcpyDeviceToHost);
    double err = 0;
    for (size_t i = 0; i < ref.size(); ++i) err = fmax(err, fabs(h[i] - ref[i]));
    printf("%-34s %.3f ms  %.2f TFLOP/s  max|diff| vs first %.3e (%s)\n", name, best, 2.0 * n * n * n / (best * 1e-3) / 1e12, err,
           hipGetErrorString(hipGetLastError()));
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
    dim3 grid(n / 128, n / 128);
    std::vector<double> ref;
    k_engine<2><<<grid, 256>>>(A, B, C, n);
    (void)hipDeviceSynchronize();
    ref.resize((size_t)n * 4);
    (void)hipMemcpy(ref.data(), C + (size_t)777 * n, ref.size() * 8, hipMemcpyDeviceToHost);
    for (int rep = 0; rep < 2; ++rep) {
        run("engine mode 0", [&] { k_engine<0><<<grid, 256>>>(A, B, C, n); }, C, n, ref);
        run("engine mode 1", [&] { k_engine<1><<<grid, 256>>>(A, B, C, n); }, C, n, ref);
        run("engine mode 2", [&] { k_engine<2><<<grid, 256>>>(A, B, C, n); }, C, n, ref);
        (void)hipMemset(C, 0, (size_t)n * n * 8);
        run("ring 4 x k8, LDS-DMA", [&] { k_ring<0><<<grid, 256>>>(A, n, B, n, C, n); }, C, n, ref);
        run("ring, no priority", [&] { k_ring<4><<<grid, 256>>>(A, n, B, n, C, n); }, C, n, ref);
        run("ring, no loads (timing only)", [&] { k_ring<1><<<grid, 256>>>(A, n, B, n, C, n); }, C, n, ref);
        run("ring, no barrier (timing only)", [&] { k_ring<2><<<grid, 256>>>(A, n, B, n, C, n); }, C, n, ref);
        run("ring, neither (timing only)", [&] { k_ring<3><<<grid, 256>>>(A, n, B, n, C, n); }, C, n, ref);
        run("ring + L2 prefetch 8 steps", [&] { k_ring<(8 << 8)><<<grid, 256>>>(A, n, B, n, C, n); }, C, n, ref);
        run("ring + L2 prefetch 16 steps", [&] { k_ring<(16 << 8)><<<grid, 256>>>(A, n, B, n, C, n); }, C, n, ref);
        run("ring + L2 prefetch 32 steps", [&] { k_ring<(32 << 8)><<<grid, 256>>>(A, n, B, n, C, n); }, C, n, ref);
        run("ring, loads of k-step 0 only", [&] { k_ring<8><<<grid, 256>>>(A, n, B, n, C, n); }, C, n, ref);
        run("ring, k-step 0 loads, no barrier", [&] { k_ring<10><<<grid, 256>>>(A, n, B, n, C, n); }, C, n, ref);
    }
    return 0;
}
