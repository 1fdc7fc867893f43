// Which CUs does a stream created with hipExtStreamCreateWithCUMask run on?  (round 6: a look-ahead of the factorisation needs the panel
// chain and the trailing update on DISJOINT sets of CUs.)  Launches many workgroups that record HW_REG_XCC_ID and HW_REG_HW_ID, on a
// stream whose mask has bits [lo, hi) set, and prints the set of (xcc, se, cu) seen.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/cumask_probe tools/cumask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void where_kernel(unsigned* out, int spin) {
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf;          // HW_REG_XCC_ID[3:0]
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);                 // HW_REG_HW_ID (all 32 bits)
        out[blockIdx.x] = (xcc << 28) | (hw & 0x0fffffff);
    }
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { }
}

int main(int argc, char** argv) {
    const int lo = argc > 1 ? atoi(argv[1]) : 0, hi = argc > 2 ? atoi(argv[2]) : 224;
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    printf("device: %s, %d CUs\n", pr.name, pr.multiProcessorCount);
    const int words = (pr.multiProcessorCount + 31) / 32;
    std::vector<uint32_t> mask(words, 0);
    for (int b = lo; b < hi && b < pr.multiProcessorCount; ++b) mask[b / 32] |= 1u << (b % 32);
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, words, mask.data());
    printf("hipExtStreamCreateWithCUMask(bits %d..%d): %s\n", lo, hi - 1, hipGetErrorString(e));
    if (e != hipSuccess) return 1;
    const int nb = 4096;
    unsigned* d; CK(hipMalloc(&d, nb * 4));
    hipLaunchKernelGGL(where_kernel, dim3(nb), dim3(64), 0, s, d, 2000);       // 20 µs of spinning: the launch spreads over every CU it may use
    CK(hipStreamSynchronize(s));
    std::vector<unsigned> h(nb);
    CK(hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost));
    std::set<unsigned> cus;
    int per_xcc[16] = {0};
    for (unsigned v : h) {
        const unsigned xcc = v >> 28, cu = (v >> 8) & 0xf, sh = (v >> 12) & 1, se = (v >> 13) & 0x7;       // HW_ID: [11:8] CU, [12] SH, [15:13] SE
        const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        if (cus.insert(key).second) per_xcc[xcc]++;
    }
    printf("distinct (xcc, se, sh, cu) seen: %zu; per XCC:", cus.size());
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
    printf("\n");
    return 0;
}
