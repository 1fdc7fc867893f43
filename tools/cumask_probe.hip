// Does a CU-masked stream (hipExtStreamCreateWithCUMask) keep CUs free for the latency-critical chain of the factorisation?
//  1. a spin kernel of 2-per-CU workgroups (80 KB LDS each, like gemm_nt_kernel) runs on a masked stream;
//  2. meanwhile a single 132-KB-LDS workgroup (like chol_diag_kernel) is launched on an unmasked stream: how long until
//     it runs, and on which CU?  Compared with the same experiment without a mask.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)

__global__ void __launch_bounds__(256, 2) busy(unsigned* out, long long spin) {
    __shared__ double pad[10240];   // 80 KB
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    pad[threadIdx.x] = hw;
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { }
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
    if (pad[threadIdx.x] < 0) out[0] = 0;
}

__global__ void __launch_bounds__(1024) fat(unsigned* out, long long* t) {
    __shared__ double pad[16896];   // 132 KB
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    pad[threadIdx.x] = hw;
    if (threadIdx.x == 0) { out[0] = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15); t[0] = wall_clock64(); }
    if (pad[threadIdx.x] < 0) out[0] = 0;
}

int run(bool masked, int reserve) {
    hipStream_t sbig, schain;
    if (masked) {
        std::vector<uint32_t> mask(8, 0xffffffffu);            // 256 bits
        // clear `reserve` bits: which CUs those are is what the probe prints
        for (int b = 0; b < reserve; ++b) mask[b / 32] &= ~(1u << (b % 32));
        CK(hipExtStreamCreateWithCUMask(&sbig, 8, mask.data()));
    } else {
        CK(hipStreamCreateWithFlags(&sbig, hipStreamNonBlocking));
    }
    CK(hipStreamCreateWithFlags(&schain, hipStreamNonBlocking));
    const int nb = 2048;                                       // 4 generations of 512
    unsigned *d, *df; long long* dt;
    CK(hipMalloc(&d, nb * 4)); CK(hipMalloc(&df, 4)); CK(hipMalloc(&dt, 8));
    busy<<<nb, 256, 0, sbig>>>(d, 10000);                      // 100 us per workgroup at 100 MHz wall clock
    CK(hipStreamSynchronize(sbig));
    auto h0 = std::chrono::steady_clock::now();
    busy<<<nb, 256, 0, sbig>>>(d, 10000);
    double lat[8];
    for (int r = 0; r < 8; ++r) {
        auto a = std::chrono::steady_clock::now();
        fat<<<1, 1024, 0, schain>>>(df, dt);
        CK(hipStreamSynchronize(schain));
        lat[r] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count();
    }
    CK(hipStreamSynchronize(sbig));
    double big_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
    std::vector<unsigned> h(nb); unsigned hf;
    CK(hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hf, df, 4, hipMemcpyDeviceToHost));
    std::set<unsigned> cus(h.begin(), h.end());
    printf("%s reserve=%d: busy kernel used %zu distinct CUs, took %.0f us (4 generations x 100 us = 400 if unhindered);\n"
           "   fat workgroup latencies (launch+run+sync, us):", masked ? "MASKED" : "plain ", reserve, cus.size(), big_us);
    for (int r = 0; r < 8; ++r) printf(" %.0f", lat[r]);
    printf("\n   fat ran on xcc %u se %u sh %u cu %u; that CU %s used by the busy kernel\n", hf >> 16, (hf >> 8) & 7, (hf >> 4) & 1, hf & 15,
           cus.count(hf) ? "WAS" : "was NOT");
    if (masked) {
        std::set<unsigned> xccs; for (unsigned v : cus) xccs.insert(v >> 16);
        int per[8] = {0}; for (unsigned v : cus) per[v >> 16]++;
        printf("   CUs per XCC used by the busy kernel:"); for (int x = 0; x < 8; ++x) printf(" %d", per[x]); printf("\n");
    }
    CK(hipStreamDestroy(sbig)); CK(hipStreamDestroy(schain));
    CK(hipFree(d)); CK(hipFree(df)); CK(hipFree(dt));
    return 0;
}

int main() {
    if (run(false, 0)) return 1;
    if (run(true, 8)) return 1;
    if (run(true, 32)) return 1;
    if (run(true, 64)) return 1;
    return 0;
}
