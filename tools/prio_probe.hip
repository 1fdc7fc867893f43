// Does a HIGH-priority stream get workgroup slots while a low-priority stream's kernel keeps every CU full (gfx950)?
//   build: hipcc --offload-arch=gfx950 -O2 -o tools/prio_probe tools/prio_probe.hip
// LO: a "bulk" kernel of many workgroups (256 threads, 80 KB LDS: two per CU), each spinning `wg_us`: G generations deep.
// HI: a chain of `nchain` small dependent kernels (1 workgroup of 256 threads, `lds_hi` bytes of LDS, spinning 20 us each).
// Reported: the chain's duration alone, and while the bulk kernel is running — with equal and with unequal stream priorities.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(256) spin_kernel(long long ticks, int* sink) {
    extern __shared__ char lds[];
    const long long t0 = wall_clock64();
    int x = 0;
    while (wall_clock64() - t0 < ticks) x += 1;
    if (threadIdx.x == 0) { lds[0] = (char)x; if (blockIdx.x == 0) *sink = x + lds[0]; }
}

int main(int argc, char** argv) {
    const int wg_us = argc > 1 ? atoi(argv[1]) : 150, gens = argc > 2 ? atoi(argv[2]) : 8, nchain = 16;
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);           // lo = least priority (numerically greatest), hi = greatest
    printf("stream priority range: least %d, greatest %d\n", lo, hi);
    int* sink;
    hipMalloc(&sink, 64);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&spin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 140000);
    for (int lds_hi : {66 * 1024, 132 * 1024}) {
        for (int mode = 0; mode < 3; ++mode) {             // 0: chain alone; 1: both default priority; 2: chain high, bulk low
            hipStream_t sb, sc;
            hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, mode == 2 ? lo : 0);
            hipStreamCreateWithPriority(&sc, hipStreamNonBlocking, mode == 2 ? hi : 0);
            hipEvent_t b0, b1, c0, c1;
            hipEventCreate(&b0); hipEventCreate(&b1); hipEventCreate(&c0); hipEventCreate(&c1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(b0, sb);
                if (mode > 0) hipLaunchKernelGGL(spin_kernel, dim3(512 * gens), dim3(256), 80 * 1024, sb, (long long)wg_us * 100, sink + 1);
                hipEventRecord(b1, sb);
                hipEventRecord(c0, sc);
                for (int i = 0; i < nchain; ++i) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(256), lds_hi, sc, 2000ll, sink);
                hipEventRecord(c1, sc);
                hipDeviceSynchronize();
                float mb = 0, mc = 0;
                hipEventElapsedTime(&mb, b0, b1);
                hipEventElapsedTime(&mc, c0, c1);
                printf("chain LDS %3d KB, mode %d (%s): chain of %d x 20 us took %.3f ms; bulk (%d generations of %d us) %.3f ms\n", lds_hi / 1024, mode,
                       mode == 0 ? "chain alone" : (mode == 1 ? "equal priorities" : "chain HIGH, bulk LOW"), nchain, mc, gens, wg_us, mb);
            }
            hipStreamDestroy(sb); hipStreamDestroy(sc);
        }
    }
    return 0;
}
