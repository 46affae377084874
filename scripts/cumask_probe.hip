// Probe: does hipExtStreamCreateWithCUMask restrict a stream's kernels to the masked CUs here?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
__global__ void k_where(unsigned* out, int spin) {
    unsigned xcc = 0, hw = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xf) << 16) | (hw & 0xffff);
}
int main() {
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    printf("CUs %d\n", pr.multiProcessorCount);
    const int nwords = (pr.multiProcessorCount + 31) / 32;
    unsigned* out; (void)hipMalloc(&out, 4096 * 4);
    for (int variant = 0; variant < 3; ++variant) {
        std::vector<uint32_t> mask(nwords, 0xffffffffu);
        if (variant == 1) { for (auto& w : mask) w = 0; mask[0] = 0xff; }          // only bits 0..7
        if (variant == 2) { mask[0] &= ~1u; }                                         // all but bit 0
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, nwords, mask.data());
        printf("variant %d create: %s\n", variant, hipGetErrorString(e));
        if (e != hipSuccess) continue;
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        (void)hipEventRecord(a, s);
        k_where<<<2048, 256, 0, s>>>(out, 20000);
        (void)hipEventRecord(b, s);
        (void)hipStreamSynchronize(s);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        std::vector<unsigned> h(2048);
        (void)hipMemcpy(h.data(), out, 2048 * 4, hipMemcpyDeviceToHost);
        std::set<unsigned> cus, xccs;
        for (unsigned v : h) { cus.insert(((v >> 16) << 16) | (v & 0xff00)); xccs.insert(v >> 16); }   // xcc | (se, cu) bits 8..15
        printf("variant %d: %.3f ms, distinct (xcc, se/cu) ids %zu, xccs %zu\n", variant, ms, cus.size(), xccs.size());
        (void)hipStreamDestroy(s);
    }
    return 0;
}
