// Probe for v_mfma_f64_16x16x4_f64 on gfx950: operand/accumulator lane maps and issue rate.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_f64_probe tools/mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// one wave: D(16x16) = A(16x4) * B(4x16); lane l supplies A[l&15][l>>4], B[l>>4][l&15]
__global__ void layout_kernel(const double* A, const double* B, double* D) {
    int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];
    double b = B[(l >> 4) * 16 + (l & 15)];
    double4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}

template <int NACC>
__global__ void __launch_bounds__(256) rate_kernel(double* out, int iters, double a0, double b0) {
    double4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = double4_t{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) fma_kernel(double* out, int iters, double a0, double b0) {
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = i;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fma(acc[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) cyc_kernel(double* out, unsigned long long* cyc, int iters, double a0, double b0) {
    double4_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = double4_t{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x * 2] = t1 - t0; cyc[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    // ---- layout
    std::vector<double> A(64), B(64), D(256), ref(256, 0.0);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1.0 + i * 0.5 + k * 7.0;
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 0.25 * j * j - 3.0 * k + 0.125;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) ref[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dD; CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dD, 256 * 8));
    CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
    layout_kernel<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize());
    CK(hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
    // hypothesis H1: lane l reg r -> row (l>>4)+4r, col l&15 ; H2: row 4*(l>>4)+r, col l&15
    int bad1 = 0, bad2 = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        double v = D[l * 4 + r];
        if (fabs(v - ref[((l >> 4) + 4 * r) * 16 + (l & 15)]) > 1e-9) bad1++;
        if (fabs(v - ref[(4 * (l >> 4) + r) * 16 + (l & 15)]) > 1e-9) bad2++;
    }
    printf("layout: H1(row=(l>>4)+4r) mismatches %d ; H2(row=4(l>>4)+r) mismatches %d\n", bad1, bad2);
    // ---- rate
    double* out; CK(hipMalloc(&out, 8ull * 256 * 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch, double flop) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-40s %8.3f ms  %8.2f TFLOP/s\n", name, ms, flop / ms * 1e-9);
    };
    {
        unsigned long long* dc; CK(hipMalloc(&dc, 16 * 4096));
        for (int rep = 0; rep < 3; ++rep) {
            int it = 40000;
            cyc_kernel<<<p.multiProcessorCount, 256>>>(out, dc, it, 1.0, 0.5); CK(hipDeviceSynchronize());
            std::vector<unsigned long long> hc(2 * p.multiProcessorCount);
            CK(hipMemcpy(hc.data(), dc, hc.size() * 8, hipMemcpyDeviceToHost));
            double cyc = 0, rt = 0; for (int i = 0; i < p.multiProcessorCount; ++i) { cyc += hc[2*i]; rt += hc[2*i+1]; }
            cyc /= p.multiProcessorCount; rt /= p.multiProcessorCount;
            printf("cyc_kernel: %.1f shader cycles per MFMA (1 wave/SIMD, 8 acc); clock %.3f GHz\n", cyc / (it * 8.0), cyc / rt * 0.1);
        }
    }
    int iters = 40000;
    int cus = p.multiProcessorCount;
    for (int bpc = 1; bpc <= 2; ++bpc) {
        int grid = cus * bpc;
        char nm[128];
        snprintf(nm, sizeof nm, "mfma_f64_16x16x4 16acc %d blk/CU", bpc);
        run(nm, [&] { rate_kernel<16><<<grid, 256>>>(out, iters, 1.0, 0.5); }, 2048.0 * 16 * iters * 4.0 * grid);
        snprintf(nm, sizeof nm, "mfma_f64_16x16x4 4acc %d blk/CU", bpc);
        run(nm, [&] { rate_kernel<4><<<grid, 256>>>(out, iters * 4, 1.0, 0.5); }, 2048.0 * 16 * iters * 4.0 * grid);
        snprintf(nm, sizeof nm, "mfma_f64_16x16x4 1acc %d blk/CU", bpc);
        run(nm, [&] { rate_kernel<1><<<grid, 256>>>(out, iters * 4, 1.0, 0.5); }, 2048.0 * 4 * iters * 4.0 * grid);
    }
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
        int grid = cus * bpc; char nm[128];
        snprintf(nm, sizeof nm, "v_fma_f64 16 chains %d blk/CU", bpc);
        run(nm, [&] { fma_kernel<<<grid, 256>>>(out, iters * 8, 1.0000001, 1e-9); }, 2.0 * 16 * iters * 8 * 256.0 * grid);
    }
    return 0;
}
