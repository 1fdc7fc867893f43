// Dependent-issue latency of the instructions on the factorisation's serial spine, ONE wave on its SIMD (the situation of the register
// potf2 step and of the 16×16 inverse in chol.hip): chains of dependent v_fma_f64, 1 / 2 / 4 independent chains, the
// VGPR → v_readlane → SGPR → VALU round trip, v_rsq_f64, an LDS broadcast read.  Times from the 100 MHz constant clock and the
// shader clock (s_memtime), per instruction.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/dp_latency_probe tools/dp_latency_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int REP = 64, UNR = 64;

__device__ __forceinline__ double readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

template <int MODE>
__global__ void __launch_bounds__(64) probe(double* out, long long* clk, double a, double b) {
    __shared__ double lds[64];
    lds[threadIdx.x] = a;
    __syncthreads();
    double x0 = a + threadIdx.x, x1 = a + 1, x2 = a + 2, x3 = a + 3;
    const long long w0 = wall_clock64(), c0 = clock64();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if constexpr (MODE == 0) x0 = fma(x0, a, b);
            if constexpr (MODE == 1) { x0 = fma(x0, a, b); x1 = fma(x1, a, b); }
            if constexpr (MODE == 2) { x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b); }
            if constexpr (MODE == 3) x0 = fma(x0, readlane_f64(x0, 5), b);                   // VALU → readlane → SGPR → VALU
            if constexpr (MODE == 4) x0 = __builtin_amdgcn_rsq(x0) + b;                      // rsq + add
            if constexpr (MODE == 5) { x0 = fma(x0, lds[(__double2loint(x0) & 7)], b); }     // LDS read whose address depends on the chain
            if constexpr (MODE == 6) x0 = x0 * a;                                            // v_mul_f64
            if constexpr (MODE == 7) { float f = (float)x0; f = fmaf(f, 1.0001f, 0.5f); x0 = f; }   // cvt + fma_f32 + cvt
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[threadIdx.x] = x0 + x1 + x2 + x3;
    if (threadIdx.x == 0) { clk[0] = w1 - w0; clk[1] = c1 - c0; }
}

template <int MODE>
void run(const char* what, int per) {
    double* out; long long* clk;
    CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&clk, 16));
    long long h[2];
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(64), 0, 0, out, clk, 1.0000001, 1e-9);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    }
    const double n = (double)REP * UNR * per;
    printf("%-58s %7.2f ns  %6.1f shader-clock ticks per instruction of the chain\n", what, h[0] * 10.0 / n, (double)h[1] / n);
    CK(hipFree(out)); CK(hipFree(clk));
}

int main() {
    run<0>("v_fma_f64, one dependent chain", 1);
    run<1>("v_fma_f64, two independent chains (per instruction)", 2);
    run<2>("v_fma_f64, four independent chains (per instruction)", 4);
    run<6>("v_mul_f64, one dependent chain", 1);
    run<3>("v_fma_f64 <- 2 x v_readlane_b32 of its own result (per round)", 1);
    run<4>("v_rsq_f64 + v_add_f64 (per round)", 1);
    run<5>("v_fma_f64 <- ds_read_b64 at an address from the chain (round)", 1);
    run<7>("cvt f64->f32, v_fma_f32, cvt f32->f64 (per round)", 1);
    return 0;
}
