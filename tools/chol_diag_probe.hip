// Phase timing of chol_diag_kernel (128×128 diagonal block: Cholesky + inverse in one workgroup).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I abstractbayesopt.jl_amd/csrc -I include \
//              -o tools/chol_diag_probe tools/chol_diag_probe.hip
#define ABO_CHOL_PROBE 1
#ifndef ABO_CHOL_SRC
#define ABO_CHOL_SRC "../abstractbayesopt.jl_amd/csrc/chol.hip"
#endif
#include ABO_CHOL_SRC
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main(int argc, char** argv) {
    // mode 0: chol_diag_kernel<0> (factor + inverse);  1: the potf2 step of the panel chain (launch_potf2_diag: potf2_pipe_kernel)
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const int n = 128;
    std::vector<double> K(n * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const double d = (i - j) / 16.0;
            K[i * n + j] = exp(-0.5 * d * d) + (i == j ? 0.1 : 0.0);
        }
    double *dK, *dK0, *dW, *dWT, *dP; int64_t* info;
    CK(hipMalloc(&dP, abo::TRSM_STREAM_BYTES));
    CK(hipMalloc(&dK, n * n * 8)); CK(hipMalloc(&dK0, n * n * 8)); CK(hipMalloc(&dW, n * n * 8)); CK(hipMalloc(&dWT, n * n * 8));
    CK(hipMalloc(&info, 8)); CK(hipMemset(info, 0, 8));
    CK(hipMemcpy(dK0, K.data(), n * n * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemcpy(dK, dK0, n * n * 8, hipMemcpyDeviceToDevice));
        CK(hipMemset(dW, 0, n * n * 8)); CK(hipMemset(dWT, 0, n * n * 8));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        if (mode == 0) CK(abo::launch_chol_diag(dK, dW, dWT, n, 0, info, 0));
        else CK(abo::launch_potf2_diag(dK, dW, dWT, n, 0, info, 0, dP));
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        long long c[16];
        CK(hipMemcpyFromSymbol(c, HIP_SYMBOL(abo_probe_clk), sizeof(c)));
        int64_t inf; CK(hipMemcpy(&inf, info, 8, hipMemcpyDeviceToHost));
        printf("rep %d info=%ld event %.1f us | load %.2f  chol %.2f  inv-diag %.2f  inv-offdiag %.2f  store %.2f  total %.2f us\n",
               rep, (long)inf, ms * 1e3, (c[1] - c[0]) / 100.0, (c[2] - c[1]) / 100.0, (c[3] - c[2]) / 100.0,
               (c[4] - c[3]) / 100.0, (c[5] - c[4]) / 100.0, (c[5] - c[0]) / 100.0);
        printf("      sub-steps 0..3: potf2+solve / trailing update (us):");
        long long prev = c[1];
        for (int p = 0; p < 4; ++p) { printf("  %.2f / %.2f", (c[6 + 2 * p] - prev) / 100.0, (c[7 + 2 * p] - c[6 + 2 * p]) / 100.0); prev = c[7 + 2 * p]; }
        printf("\n");
    }
    // FNV-1a over L, W and WT of the last repetition: two builds of the kernel that print the same value produced the same bits
    std::vector<double> out(3 * n * n);
    CK(hipMemcpy(out.data(), dK, n * n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(out.data() + n * n, dW, n * n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(out.data() + 2 * n * n, dWT, n * n * 8, hipMemcpyDeviceToHost));
    unsigned long long hsh = 1469598103934665603ull;
    const unsigned char* b = reinterpret_cast<const unsigned char*>(out.data());
    for (size_t i = 0; i < out.size() * 8; ++i) { hsh ^= b[i]; hsh *= 1099511628211ull; }
    printf("hash of L | W | WT: %016llx\n", hsh);
    // accuracy of the factor against a long-double Cholesky of the same block (round 6: the register step's pivot chain computes only
    // 1/sqrt(d), two Goldschmidt steps on the v_rsq_f64 seed)
    {
        std::vector<long double> R(n * n);
        for (int i = 0; i < n * n; ++i) R[i] = K[i];
        for (int j = 0; j < n; ++j) {
            long double dj = R[j * n + j];
            for (int k = 0; k < j; ++k) dj -= R[j * n + k] * R[j * n + k];
            const long double l = sqrtl(dj);
            R[j * n + j] = l;
            for (int i = j + 1; i < n; ++i) {
                long double v = R[i * n + j];
                for (int k = 0; k < j; ++k) v -= R[i * n + k] * R[j * n + k];
                R[i * n + j] = v / l;
            }
        }
        double emax = 0.0, ediag = 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j) {
                const double e = fabs((double)((long double)out[i * n + j] - R[i * n + j]));
                if (e > emax) emax = e;
                if (i == j && e > ediag) ediag = e;
            }
        printf("max |L - L_longdouble| = %.3e (diagonal %.3e); L entries are O(1)\n", emax, ediag);
    }
    return 0;
}
