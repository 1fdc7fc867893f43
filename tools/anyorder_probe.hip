// Does hipExtAnyOrderLaunch let two kernels of ONE stream run side by side on gfx950?
//   build: hipcc --offload-arch=gfx950 -O2 -o tools/anyorder_probe tools/anyorder_probe.hip
// Kernel A and B each spin for ~200 us in one workgroup.  Serial: ~400 us per pair; overlapped: ~200 us.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__global__ void spin_kernel(long long ticks, int* sink) {
    const long long t0 = wall_clock64();
    int x = 0;
    while (wall_clock64() - t0 < ticks) x += 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) *sink = x;
}

int main() {
    hipStream_t s;
    hipStreamCreate(&s);
    int* sink;
    hipMalloc(&sink, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const long long ticks = 20000;      // 100 MHz constant clock: 200 us
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, s);
            for (int i = 0; i < 8; ++i) {
                hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ticks, sink);
                if (mode == 0) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ticks, sink + 1);
                else if (mode == 1) hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, sink + 1);
                else { hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, sink + 1);
                       hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, sink + 2); }
            }
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("mode %d (%s): 8 groups in %.3f ms = %.1f us per group (one kernel spins 200 us)\n", mode,
                   mode == 0 ? "A, B plain" : (mode == 1 ? "A plain, B any-order" : "A plain, B and C any-order"), ms, ms * 1000 / 8);
        }
    }
    return 0;
}
