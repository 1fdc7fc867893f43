// Issue cost of the generator's per-residue instruction sequence (kgen_core.h, RES path), gfx950: candidates for the 14-modulus
// unrolled body on a lane's pair of values, timed with 1 … 4 waves per SIMD on every CU.
//   build: hipcc --offload-arch=gfx950 -O2 -o tools/residue_seq_probe tools/residue_seq_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../abstractbayesopt.jl_amd/csrc/abo_oz_dev.h"

using namespace abo;

// VARIANT 0: shipped (3 fma, mul, rndne, fma, cvt; pack two bytes with and/shl/or as the compiler pleases)
// VARIANT 1: quotient and residue through the rounding constant 1.5·2^23 (no rndne, no cvt), constants as literals
// VARIANT 2: as 1, the rounding constant and its negation passed in registers (kernel arguments → SGPRs)
template <int VARIANT>
__global__ void __launch_bounds__(256) probe(int iters, float magic, float* sink, unsigned short* out) {
    OzLimbs x0{(float)(threadIdx.x & 8191), (float)((threadIdx.x * 7) & 8191), (float)((threadIdx.x * 13) & 8191), (float)(threadIdx.x & 2047)};
    OzLimbs x1{(float)((threadIdx.x * 3) & 8191), (float)((threadIdx.x * 5) & 8191), (float)((threadIdx.x * 11) & 8191), (float)((threadIdx.x * 17) & 2047)};
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int l = 0; l < 14; ++l) {
            const float pf = (float)oz_mod_p(l), invp = 1.0f / (float)oz_mod_p(l);
            int r0, r1;
            if constexpr (VARIANT == 0) {
                r0 = sym_residue_f32(x0, oz_mod_c14(l, 1), oz_mod_c14(l, 2), oz_mod_c14(l, 3), invp, pf);
                r1 = sym_residue_f32(x1, oz_mod_c14(l, 1), oz_mod_c14(l, 2), oz_mod_c14(l, 3), invp, pf);
            } else {
                const float M = VARIANT == 1 ? 12582912.0f : magic;
                const float t0 = __builtin_fmaf(x0.a3, oz_mod_c14(l, 3), __builtin_fmaf(x0.a2, oz_mod_c14(l, 2), __builtin_fmaf(x0.a1, oz_mod_c14(l, 1), x0.a0)));
                const float t1 = __builtin_fmaf(x1.a3, oz_mod_c14(l, 3), __builtin_fmaf(x1.a2, oz_mod_c14(l, 2), __builtin_fmaf(x1.a1, oz_mod_c14(l, 1), x1.a0)));
                const float q0 = __builtin_fmaf(t0, invp, M) - M;
                const float q1 = __builtin_fmaf(t1, invp, M) - M;
                const float s0 = __builtin_fmaf(-q0, pf, t0) + M;
                const float s1 = __builtin_fmaf(-q1, pf, t1) + M;
                r0 = __float_as_int(s0);
                r1 = __float_as_int(s1);
            }
            const unsigned pk = (unsigned)((r0 & 0xff) | ((r1 & 0xff) << 8));
            acc += pk;                               // stands for the 2-byte store
            asm volatile("" : "+v"(acc));
        }
        // new values next round (cheap, keeps the compiler from hoisting the bodies)
        x0.a0 += 1.0f; x1.a0 += 1.0f;
        asm volatile("" : "+v"(x0.a0), "+v"(x1.a0));
    }
    if (acc == 0x12345678u) sink[0] = 1.0f;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned short)acc;
}

template <int VARIANT>
static void run(const char* name) {
    const int iters = 20000, blocks = 256;
    float* sink;
    unsigned short* out;
    hipMalloc(&sink, 64);
    hipMalloc(&out, sizeof(unsigned short) * blocks * 4 * 256);
    for (int w = 1; w <= 4; ++w) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(probe<VARIANT>, dim3(blocks * w), dim3(256), 0, 0, iters, 12582912.0f, sink, out);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe<VARIANT>, dim3(blocks * w), dim3(256), 0, 0, iters, 12582912.0f, sink, out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        // residue pairs per SIMD = iters · 14 · w
        printf("%-44s %d wave(s)/SIMD: %.2f ns per residue PAIR per SIMD (kernel %.2f ms)\n", name, w, ms * 1e6 / ((double)iters * 14 * w), ms);
    }
    unsigned short h[4];
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    printf("   check %u %u %u %u\n", h[0], h[1], h[2], h[3]);
    hipFree(sink); hipFree(out);
}

int main() {
    run<0>("shipped: mul, rndne, fma, cvt");
    run<1>("rounding constant, literal");
    run<2>("rounding constant, register");
    return 0;
}
