// What a tiny device-to-host copy costs on the stream of a latency-bound step: pageable against pinned destination.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/memcpy_probe tools/memcpy_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k(double* p) { if (threadIdx.x == 0) p[0] += 1.0; }
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    double* d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
    double* pin; CK(hipHostMalloc(&pin, 4096));
    double* pag = (double*)malloc(4096);
    const int reps = 2000;
    for (int variant = 0; variant < 6; ++variant) {
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r) {
            // a "step": 4 kernels, and 0 / 2 / 4 small read-backs in between, one synchronisation at the end
            for (int q = 0; q < 4; ++q) {
                hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d);
                if (variant == 1 && q < 2) CK(hipMemcpyAsync(pag + 8 * q, d, 16, hipMemcpyDeviceToHost, s));
                if (variant == 2 && q < 2) CK(hipMemcpyAsync(pin + 8 * q, d, 16, hipMemcpyDeviceToHost, s));
                if (variant == 3) CK(hipMemcpyAsync(pag + 8 * q, d, 16, hipMemcpyDeviceToHost, s));
                if (variant == 4) CK(hipMemcpyAsync(pin + 8 * q, d, 16, hipMemcpyDeviceToHost, s));
                if (variant == 5 && q == 0) { CK(hipMemcpyAsync(d + 64, pag, 800, hipMemcpyHostToDevice, s)); CK(hipMemcpyAsync(d + 256, pag + 128, 200, hipMemcpyHostToDevice, s)); }
            }
            CK(hipStreamSynchronize(s));
        }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        const char* names[] = {"4 kernels + sync", "+ 2 D2H of 16 B to pageable memory", "+ 2 D2H of 16 B to pinned memory", "+ 4 D2H to pageable", "+ 4 D2H to pinned",
                               "+ 2 H2D (800 B, 200 B) from pageable memory"};
        printf("%-48s %.1f us per step\n", names[variant], us);
    }
    return 0;
}
