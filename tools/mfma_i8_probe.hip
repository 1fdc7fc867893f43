// Issue rate of the gfx950 int8 MFMAs (v_mfma_i32_16x16x64_i8 / v_mfma_i32_32x32x32_i8): the number behind the
// "int8-sliced fp64" note in DESIGN.md §7 — how much faster than the fp64 matrix pipe (78.6 TFLOP/s) the integer pipe is.
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o tools/mfma_i8_probe tools/mfma_i8_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i4_t __attribute__((ext_vector_type(4)));
typedef int i16_t __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void __launch_bounds__(256) rate16(int* out, int iters, int a0) {
    i4_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = i4_t{0, 0, 0, 0};
    i4_t a = {a0 + (int)threadIdx.x, a0, 3, 4}, b = {a0, 7, (int)threadIdx.x, 1};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
    }
    int s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) rate32(int* out, int iters, int a0) {
    i16_t acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    i4_t a = {a0 + (int)threadIdx.x, a0, 3, 4}, b = {a0, 7, (int)threadIdx.x, 1};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
    }
    int s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    int* out; CK(hipMalloc(&out, 4ull * 256 * 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 40000, cus = p.multiProcessorCount;
    for (int bpc = 1; bpc <= 2; ++bpc) {
        const int grid = cus * bpc;
        for (int which = 0; which < 2; ++which) {
            auto launch = [&]() { if (which == 0) rate16<<<grid, 256>>>(out, iters, 1); else rate32<<<grid, 256>>>(out, iters, 1); };
            launch(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double ops = (which == 0 ? 2.0 * 16 * 16 * 64 * 8 : 2.0 * 32 * 32 * 32 * 4) * iters * 4.0 * grid;   // 4 waves per block
            printf("%s, %d block(s)/CU: %8.3f ms  %8.1f TOP/s\n", which == 0 ? "v_mfma_i32_16x16x64_i8" : "v_mfma_i32_32x32x32_i8", bpc, ms, ops / ms * 1e-9);
        }
    }
    return 0;
}
