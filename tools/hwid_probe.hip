// Which HW_ID fields tell two co-resident workgroups of a CU apart? (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <set>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)
__global__ void __launch_bounds__(256, 2) k(unsigned* out, int spin) {
    __shared__ double pad[10240];   // 80 KB -> 2 workgroups per CU
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID full
    unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);     // HW_REG_XCC_ID[3:0]
    pad[threadIdx.x] = hw;
    long long t0 = clock64();
    while (clock64() - t0 < spin) { }
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
    if (pad[threadIdx.x] < 0) out[0] = 0;
}
int main() {
    const int nb = 512;
    unsigned* d; CK(hipMalloc(&d, nb * 4 * 2 * 4));
    k<<<nb, 256>>>(d, 2000000); CK(hipDeviceSynchronize());
    std::vector<unsigned> h(nb * 8); CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
    std::map<unsigned, std::set<unsigned>> tg_per_cu;   // key: (xcc, se, sh, cu)
    std::map<unsigned, int> tgcount;
    for (int b = 0; b < nb; ++b) for (int w = 0; w < 4; ++w) {
        unsigned hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1];
        unsigned wave = hw & 15, simd = (hw >> 4) & 3, pipe = (hw >> 6) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, tg = (hw >> 16) & 15;
        if (b < 6) printf("blk %d wave %d: hw=%08x xcc=%u wave_id=%u simd=%u pipe=%u cu=%u sh=%u se=%u tg=%u vm=%u queue=%u state=%u me=%u\n", b, w, hw, xcc, wave, simd, pipe, cu, sh, se, tg, (hw>>20)&15, (hw>>24)&7, (hw>>27)&7, hw>>30);
        unsigned key = (xcc << 16) | (se << 8) | (sh << 4) | cu;
        if (w == 0) { tg_per_cu[key].insert(tg); tgcount[tg]++; }
    }
    printf("distinct CUs seen: %zu\n", tg_per_cu.size());
    int two = 0; for (auto& kv : tg_per_cu) if (kv.second.size() == 2) two++;
    printf("CUs with 2 distinct TG ids: %d\n", two);
    for (auto& kv : tgcount) printf("tg=%u count=%d\n", kv.first, kv.second);
    return 0;
}
