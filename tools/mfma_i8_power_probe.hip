// Sustained rate of the gfx950 int8 MFMAs on RANDOM operands (the chip lowers its clock under this load): what the int8-residue
// GEMM of csrc/ozaki.hip can at most reach.  Launches back to back for ≈3 s per shape, reports the last launches' rate and the
// in-kernel clock (Δs_memtime / Δs_memrealtime × 100 MHz).
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o tools/mfma_i8_power_probe tools/mfma_i8_power_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i4_t __attribute__((ext_vector_type(4)));
typedef int i16_t __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int SHAPE>
__global__ void __launch_bounds__(256) rate(int* out, unsigned long long* clk, int iters, int zero) {
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    i4_t a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int q = 0; q < 4; ++q) {
            a[i][q] = zero ? 0 : (int)hash32(t * 64 + i * 8 + q);
            b[i][q] = zero ? 0 : (int)hash32(t * 64 + i * 8 + q + 4);
        }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
    if (SHAPE == 32) {
        i16_t acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[(i + it) & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    } else {
        i4_t acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = i4_t{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i & 3], b[(i + it) & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[t] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount, grid = cus * 2;
    int* out; CK(hipMalloc(&out, 4ull * 256 * grid));
    unsigned long long* clk; CK(hipMalloc(&clk, 16ull * grid));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 200000;
    for (int zero = 1; zero >= 0; --zero)
        for (int shape : {32, 16}) {
            float ms = 0;
            for (int rep = 0; rep < 60; ++rep) {      // ≈ 50 ms per launch → 3 s
                CK(hipEventRecord(e0));
                if (shape == 32) rate<32><<<grid, 256>>>(out, clk, iters, zero); else rate<16><<<grid, 256>>>(out, clk, iters, zero);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
            const double ops = (shape == 32 ? 2.0 * 32 * 32 * 32 * 4 : 2.0 * 16 * 16 * 64 * 8) * iters * 4.0 * grid;
            printf("%s operands, v_mfma_i32_%s_i8, 2 blocks/CU: %8.3f ms  %7.1f TOP/s  in-kernel clock %.2f GHz\n", zero ? "zero  " : "random",
                   shape == 32 ? "32x32x32" : "16x16x64", ms, ops / ms * 1e-9, (double)h[0] / (double)h[1] * 0.1);
        }
    return 0;
}
