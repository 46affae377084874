// Probe: does a kernel launched with hipExtAnyOrderLaunch overlap its predecessor in the same stream on this GPU?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long long ticks, int* sink) {      // wall_clock64 runs at 100 MHz
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (sink && threadIdx.x == 9999) *sink = 1;
}
int main() {
    hipStream_t st;
    (void)hipStreamCreate(&st);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0, st);
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, 20000LL, (int*)nullptr);             // 200 us, one workgroup
            if (mode == 0)
                hipLaunchKernelGGL(spin, dim3(512), dim3(256), 0, st, 15000LL, (int*)nullptr);      // 150 us, 512 workgroups
            else
                hipExtLaunchKernelGGL(spin, dim3(512), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, 15000LL, (int*)nullptr);
            (void)hipEventRecord(e1, st);
            (void)hipStreamSynchronize(st);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.1f us (serial would be ~350, overlapped ~200)\n", mode ? "any-order" : "in-order ", ms * 1e3);
        }
    }
    return 0;
}
