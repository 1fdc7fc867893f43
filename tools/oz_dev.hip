// Development harness of the int8-residue contraction (csrc/ozaki.hip): layout / exactness checks of every stage against the
// host at a small size, then kernel timings at the C3 chunk shape (N = 8192, Mc = 16384).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -o tools/oz_dev tools/oz_dev.hip
// Run:   tools/oz_dev [nmod] [Nbig] [Mcbig]
#define OZ_PROBE 1
#include "../abstractbayesopt.jl_amd/csrc/ozaki.hip"
#include <cstdio>
#include <algorithm>
#include <random>
#include <vector>
using namespace abo;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

static int host_symres(long double x, int p) {
    long double q = rintl(x / p);
    long long r = (long long)(x - q * p);
    const int h = (p - 1) / 2;
    if (p & 1) { if (r < -h) r += p; if (r > h) r -= p; }
    return (int)r;
}

int main(int argc, char** argv) {
    const int nmod = argc > 1 ? atoi(argv[1]) : 14;
    const int NB = argc > 2 ? atoi(argv[2]) : 8192;
    const int MB = argc > 3 ? atoi(argv[3]) : 16384;
    OzPlan pl;
    if (!oz_make_plan(nmod, &pl)) { printf("bad plan\n"); return 1; }
    printf("plan: n=%d eP=%d moduli", pl.n, pl.eP);
    for (int l = 0; l < pl.n; ++l) printf(" %d", pl.p[l]);
    printf("\n");
    hipStream_t s; CK(hipStreamCreate(&s));

    // ---------------- correctness at Np = 640 (pads to 768), Mc = 384 (pads to 512), nvalid = 600 ----------------
    {
        const int Np = 640, Mc = 384, nvalid = 600;
        const int Np256 = (int)pad_up(Np, 256), Mc256 = (int)pad_up(Mc, 256);
        std::mt19937_64 rng(7);
        std::normal_distribution<double> nd;
        std::uniform_real_distribution<double> ud(0.0, 1.0);
        std::vector<double> W((size_t)Np * Np, 0.0), K((size_t)Mc * Np, 0.0);
        for (int i = 0; i < Np; ++i)
            for (int k = 0; k <= i; ++k) W[(size_t)i * Np + k] = (i < nvalid && k < nvalid) ? nd(rng) * std::exp(-0.02 * (i - k)) * (1 + i % 7) : (i == k ? 1.0 : 0.0);
        for (int j = 0; j < Mc; ++j)
            for (int k = 0; k < nvalid; ++k) K[(size_t)j * Np + k] = std::pow(ud(rng), 3.0);
        double *dW, *dK, *dP; int8_t *WR, *KR, *U; int *sexp, *badr, *badc;
        CK(hipMalloc(&dW, sizeof(double) * W.size())); CK(hipMalloc(&dK, sizeof(double) * K.size()));
        CK(hipMalloc(&dP, sizeof(double) * (Np / 128) * Mc));
        CK(hipMalloc(&WR, oz_w_bytes(nmod, Np))); CK(hipMalloc(&KR, oz_k_bytes(nmod, Np, Mc))); CK(hipMalloc(&U, oz_k_bytes(nmod, Np, Mc)));
        CK(hipMalloc(&sexp, 4 * Np256)); CK(hipMalloc(&badr, 4 * Np256)); CK(hipMalloc(&badc, 4 * (Mc256 + OZ_CTR_INTS)));
        CK(hipMemcpy(dW, W.data(), sizeof(double) * W.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(dK, K.data(), sizeof(double) * K.size(), hipMemcpyHostToDevice));
        CK(hipMemset(U, 0x55, oz_k_bytes(nmod, Np, Mc)));
        CK(oz_prepare_w(pl, dW, Np, Np, nvalid, WR, sexp, badr, s, 1, 0));
        OzVarArgs v{};
        v.plan = &pl; v.Kxz = dK; v.ldk = Np; v.WR = WR; v.sexp = sexp; v.bad_row = badr; v.KR = KR; v.U = U; v.bad_col = badc;
        v.partial = dP; v.ldp = Mc; v.Np = Np; v.Mc = Mc; v.nvalid = nvalid; v.sK = oz_k_scale(1.0);
        CK(launch_var_ozaki(v, s));
        CK(hipStreamSynchronize(s));
        std::vector<int> hs(Np256);
        std::vector<int8_t> hWR(oz_w_bytes(nmod, Np)), hKR(oz_k_bytes(nmod, Np, Mc)), hU(oz_k_bytes(nmod, Np, Mc));
        std::vector<double> hP((size_t)(Np / 128) * Mc);
        CK(hipMemcpy(hs.data(), sexp, 4 * Np256, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hWR.data(), WR, hWR.size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(hKR.data(), KR, hKR.size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(hU.data(), U, hU.size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(hP.data(), dP, sizeof(double) * hP.size(), hipMemcpyDeviceToHost));
        // 1. residues of W and K
        long bad_w = 0, bad_k = 0, bad_u = 0;
        for (int l = 0; l < nmod; ++l) {
            for (int i = 0; i < Np256; ++i)
                for (int k = 0; k < Np256; ++k) {
                    int want = 0;
                    if (i < nvalid && k <= i) want = host_symres(rintl(ldexpl((long double)W[(size_t)i * Np + k], hs[i])), pl.p[l]);
                    const int got = hWR[(size_t)l * Np256 * Np256 + (size_t)oz_plane_off(i, k, Np256 / 64)];
                    if ((int8_t)want != (int8_t)got) { if (bad_w < 5) printf("WR mismatch l=%d i=%d k=%d want %d got %d\n", l, i, k, want, got); ++bad_w; }
                }
            for (int j = 0; j < Mc256; ++j)
                for (int k = 0; k < Np256; ++k) {
                    int want = 0;
                    if (j < Mc && k < Np) want = host_symres(rintl(ldexpl((long double)K[(size_t)j * Np + k], v.sK)), pl.p[l]);
                    const int got = hKR[(size_t)l * Mc256 * Np256 + (size_t)oz_plane_off(j, k, Np256 / 64)];
                    if ((int8_t)want != (int8_t)got) { if (bad_k < 5) printf("KR mismatch l=%d j=%d k=%d want %d got %d\n", l, j, k, want, got); ++bad_k; }
                }
        }
        // 2. U from the device's own residues (exact integer GEMM on the host)
        for (int l = 0; l < nmod; ++l)
            for (int i = 0; i < Np256; ++i)
                for (int j = 0; j < Mc256; ++j) {
                    long long acc = 0;
                    const int8_t* wr = &hWR[(size_t)l * Np256 * Np256];
                    const int8_t* kr = &hKR[(size_t)l * Mc256 * Np256];
                    const int kend = 256 * (i / 256 + 1);
                    for (int k = 0; k < kend; ++k) acc += (int)wr[oz_plane_off(i, k, Np256 / 64)] * (int)kr[oz_plane_off(j, k, Np256 / 64)];
                    const int want = host_symres((long double)acc, pl.p[l]);
                    const int got = hU[(size_t)l * Np256 * Mc256 + (size_t)i * Mc256 + j];
                    if ((int8_t)want != (int8_t)got) { if (bad_u < 8) printf("U mismatch l=%d i=%d j=%d want %d got %d\n", l, i, j, want, got); ++bad_u; }
                }
        // 3. partial sums against a long-double product of the fp64 operands
        double worst = 0, worst_q = 0;
        for (int tb = 0; tb < Np / 128; ++tb)
            for (int j = 0; j < Mc; ++j) {
                long double sum = 0, sumq = 0;
                for (int i = tb * 128; i < tb * 128 + 128 && i < nvalid; ++i) {
                    long double vv = 0, vq = 0;
                    for (int k = 0; k <= i; ++k) {
                        vv += (long double)W[(size_t)i * Np + k] * K[(size_t)j * Np + k];
                        vq += rintl(ldexpl((long double)W[(size_t)i * Np + k], hs[i])) * rintl(ldexpl((long double)K[(size_t)j * Np + k], v.sK));
                    }
                    vq = ldexpl(vq, -(hs[i] + v.sK));
                    sum += vv * vv; sumq += vq * vq;
                }
                const double got = hP[(size_t)tb * Mc + j];
                const double e = (double)(fabsl(got - sum) / (1.0L + sum)), eq = (double)(fabsl(got - sumq) / (1.0L + sumq));
                if (e > worst) worst = e;
                if (eq > worst_q) worst_q = eq;
            }
        printf("check: residue mismatches W %ld K %ld U %ld; partial rel err vs long double %.3e, vs quantised-exact %.3e\n", bad_w, bad_k, bad_u, worst, worst_q);
        hipFree(dW); hipFree(dK); hipFree(dP); hipFree(WR); hipFree(KR); hipFree(U); hipFree(sexp); hipFree(badr); hipFree(badc);
    }

    // ---------------- timings at the C3 chunk shape ----------------
    {
        const int Np = NB, Mc = MB;
        const int Np256 = (int)pad_up(Np, 256), Mc256 = (int)pad_up(Mc, 256);
        std::vector<double> W((size_t)Np * Np, 0.0), K((size_t)Mc * Np);
        std::mt19937_64 rng(11);
        std::normal_distribution<double> nd;
        std::uniform_real_distribution<double> ud(0.0, 1.0);
        for (int i = 0; i < Np; ++i) for (int k = 0; k <= i; ++k) W[(size_t)i * Np + k] = nd(rng);
        for (size_t q = 0; q < K.size(); ++q) K[q] = ud(rng);
        double *dW, *dK, *dP; int8_t *WR, *KR, *U; int *sexp, *badr, *badc;
        CK(hipMalloc(&dW, sizeof(double) * W.size())); CK(hipMalloc(&dK, sizeof(double) * K.size()));
        CK(hipMalloc(&dP, sizeof(double) * (Np / 128) * Mc));
        CK(hipMalloc(&WR, oz_w_bytes(nmod, Np))); CK(hipMalloc(&KR, oz_k_bytes(nmod, Np, Mc))); CK(hipMalloc(&U, oz_k_bytes(nmod, Np, Mc)));
        CK(hipMalloc(&sexp, 4 * Np256)); CK(hipMalloc(&badr, 4 * Np256)); CK(hipMalloc(&badc, 4 * (Mc256 + OZ_CTR_INTS)));
        CK(hipMemcpy(dW, W.data(), sizeof(double) * W.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(dK, K.data(), sizeof(double) * K.size(), hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s));
        CK(oz_prepare_w(pl, dW, Np, Np, Np, WR, sexp, badr, s, 1, 0));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("prepare_w N=%d: %.3f ms\n", Np, ms);
        OzVarArgs v{};
        v.plan = &pl; v.Kxz = dK; v.ldk = Np; v.WR = WR; v.sexp = sexp; v.bad_row = badr; v.KR = KR; v.U = U; v.bad_col = badc;
        v.partial = dP; v.ldp = Mc; v.Np = Np; v.Mc = Mc; v.nvalid = Np; v.sK = oz_k_scale(1.0);
        hipEvent_t eq, eg; CK(hipEventCreate(&eq)); CK(hipEventCreate(&eg));
        v.ev_quant = eq; v.ev_gemm = eg;
        const int reps = argc > 4 ? atoi(argv[4]) : 40;
        std::vector<float> tq, tg, tc, tt;
        for (int rep = 0; rep < reps; ++rep) {
            CK(hipEventRecord(e0, s));
            CK(launch_var_ozaki(v, s));
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float a, b, c, d;
            CK(hipEventElapsedTime(&a, e0, eq)); CK(hipEventElapsedTime(&b, eq, eg)); CK(hipEventElapsedTime(&c, eg, e1)); CK(hipEventElapsedTime(&d, e0, e1));
            if (rep >= reps / 2) { tq.push_back(a); tg.push_back(b); tc.push_back(c); tt.push_back(d); }
        }
        {
            long long pa[16];
            CK(hipMemcpyFromSymbol(pa, HIP_SYMBOL(oz_probe_acc), sizeof(pa)));
            if (pa[2] > 0)
                printf("persistent GEMM, workgroup 0 over %lld tiles (%lld half-stages): k loop %.2f us per tile (%.3f us per half-stage), epilogue %.2f us per tile\n",
                       pa[2], pa[3], pa[0] / 100.0 / pa[2], pa[0] / 100.0 / pa[3], pa[1] / 100.0 / pa[2]);
            if (pa[2] > 0) printf("   of the epilogue: residue arithmetic %.2f us per tile; to the barrier behind pass 0..4: %.2f %.2f %.2f %.2f %.2f us\n", (pa[4] - pa[5]) / 100.0 / pa[2],
                                  (pa[8] - pa[4]) / 100.0 / pa[2], (pa[9] - pa[8]) / 100.0 / pa[2], (pa[10] - pa[9]) / 100.0 / pa[2], (pa[11] - pa[10]) / 100.0 / pa[2], (pa[12] - pa[11]) / 100.0 / pa[2]);
        }
        auto med = [](std::vector<float> x) { std::sort(x.begin(), x.end()); return x[x.size() / 2]; };
        const double macs = (double)nmod * Mc256 * 256.0 * 256.0 * (Np256 / 256) * (Np256 / 256 + 1) / 2.0;
        printf("var_ozaki N=%d Mc=%d n=%d (median of the last %d of %d back-to-back calls): total %.3f ms = quant %.3f + gemm %.3f + crt %.3f;"
               " gemm %.1f TOP/s; fp64-equivalent of the call %.1f TFLOP/s\n", Np, Mc, nmod, (int)tt.size(), reps, med(tt), med(tq), med(tg), med(tc),
               2.0 * macs / med(tg) * 1e-9, (double)Np * Np * Mc / med(tt) * 1e-9);
    }
    return 0;
}
