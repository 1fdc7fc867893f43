// Issue cost of the vector instructions the kernel-matrix generator is made of, gfx950.
//   build: hipcc --offload-arch=gfx950 -O2 -o tools/valu_rate_probe tools/valu_rate_probe.hip
// One workgroup of 256 threads (one wave per SIMD) per CU; each wave runs ITER × 32 independent instructions of one kind
// (32 accumulator chains: no dependency stalls) and reads s_memrealtime / the shader clock around them.
// Printed: shader cycles per wave-instruction with 1 wave per SIMD and with 2 waves per SIMD (512 threads).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP32(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) \
                 X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31)

template <int KIND>
__global__ void probe(int iters, long long* cyc, float* sink) {
    float f[32];
    double d[32];
    int n[32];
    const float fa = 1.0000001f + threadIdx.x * 1e-9f, fb = 1e-7f;
    const double da = 1.0000000001 + threadIdx.x * 1e-12, db = 1e-11;
#pragma unroll
    for (int i = 0; i < 32; ++i) { f[i] = i + threadIdx.x; d[i] = i + threadIdx.x; n[i] = i * 77 + threadIdx.x; }
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fa), "v"(fb));
            REP32(X)
#undef X
        } else if constexpr (KIND == 1) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(da), "v"(db));
            REP32(X)
#undef X
        } else if constexpr (KIND == 2) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fa));
            REP32(X)
#undef X
        } else if constexpr (KIND == 3) {
#define X(i) asm volatile("v_rndne_f32 %0, %0" : "+v"(f[i]));
            REP32(X)
#undef X
        } else if constexpr (KIND == 4) {
#define X(i) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(n[i]) : "v"(f[i]));
            REP32(X)
#undef X
        } else if constexpr (KIND == 5) {
#define X(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(n[i]) : "v"(n[(i + 1) & 31]), "v"(0x05010400));
            REP32(X)
#undef X
        } else if constexpr (KIND == 6) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
            REP32(X)
#undef X
        } else if constexpr (KIND == 7) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(da));
            REP32(X)
#undef X
        } else if constexpr (KIND == 8) {
#define X(i) asm volatile("v_rndne_f64 %0, %0" : "+v"(d[i]));
            REP32(X)
#undef X
        } else if constexpr (KIND == 9) {
#define X(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(n[i]) : "v"(0x00ffffff));
            REP32(X)
#undef X
        } else if constexpr (KIND == 10) {
#define X(i) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 31]));
            REP32(X)
#undef X
        } else if constexpr (KIND == 11) {
#define X(i) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[i]) : "v"(n[i]));
            REP32(X)
#undef X
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += f[i] + (float)d[i] + (float)n[i];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void run(const char* name) {
    const int iters = 100000, blocks = 256;
    long long* cyc;
    float* sink;
    hipMalloc(&cyc, sizeof(long long) * blocks * 8);
    hipMalloc(&sink, 64);
    for (int threads = 256; threads <= 512; threads += 256) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(threads), 0, 0, iters, cyc, sink);   // warm-up
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(threads), 0, 0, iters, cyc, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        // wave-instructions per SIMD = iters · 32 · (waves per SIMD); time per wave-instruction per SIMD
        const double per_simd = (double)iters * 32 * (threads / 256);
        printf("%-14s %d wave(s)/SIMD: %.3f ns per wave-instruction per SIMD (kernel %.3f ms)\n", name, threads / 256,
               ms * 1e6 / per_simd, ms);
    }
    hipFree(cyc); hipFree(sink);
}

int main() {
    run<0>("v_fma_f32");
    run<1>("v_fma_f64");
    run<2>("v_mul_f32");
    run<3>("v_rndne_f32");
    run<4>("v_cvt_i32_f32");
    run<5>("v_perm_b32");
    run<6>("v_add_f64");
    run<7>("v_mul_f64");
    run<8>("v_rndne_f64");
    run<9>("v_and_b32");
    run<10>("v_lshl_or_b32");
    run<11>("v_cvt_f32_i32");
    return 0;
}
