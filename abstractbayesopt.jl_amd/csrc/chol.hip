// Serial pieces of the factorisation: the 128×128 diagonal-block Cholesky + triangular inverse,
// the triangular mat-vecs for alpha, and the NLML reductions.
//
// Reference arithmetic being replaced: [upstream AbstractGPs] posterior(FiniteGP, y) —
// C = cholesky(K + σ²I) (LAPACK potrf), α = C \ (y − m)  — called from update(),
// src/surrogates/StandardGP.jl:79-83; −logpdf at :99-114.  The blocked driver lives in api.hip:
// right-looking, 128-wide panels; this kernel is its potf2 + trtri step, the MFMA GEMM core
// (gemm.hip) does the panel solve, the trailing SYRK and the blocked L⁻¹.
#include "abo_kernels.h"

namespace abo {

constexpr int NB = 128;
constexpr int LDA = NB + 1;   // odd stride: column walks are conflict-free

// One workgroup (256 threads), block resident in LDS (129 KB of the CU's 160 KB).
//   phase 1: unblocked right-looking Cholesky on the lower triangle (pivot check per column)
//   phase 2: Linv by forward substitution, one column per thread; column c of Linv is written
//            into ROW c of the (now free) strict upper triangle, so no second LDS image is needed
//   phase 3: write L (lower, zeros above) back to K, Linv to W (lower) and Linvᵀ to WT (upper)
__global__ void __launch_bounds__(256) chol_diag_kernel(double* K, double* W, double* WT, int64_t ld, int r0,
                                                        int64_t* info) {
    __shared__ double a[NB * LDA];
    __shared__ double col[NB];
    __shared__ double dinv[NB];
    __shared__ int fail;
    const int t = threadIdx.x;
    if (*info != 0) return;
    if (t == 0) fail = 0;
    double* Kb = K + (int64_t)r0 * ld + r0;
    for (int idx = t; idx < NB * NB; idx += 256) {
        const int i = idx >> 7, j = idx & 127;
        a[i * LDA + j] = Kb[(int64_t)i * ld + j];
    }
    __syncthreads();

    for (int j = 0; j < NB; ++j) {
        const double d = a[j * LDA + j];
        if (!(d > 0.0)) {          // also catches NaN; uniform across the block
            if (t == 0) { fail = 1; *info = (int64_t)r0 + j + 1; }
            break;
        }
        const double piv = sqrt(d);
        __syncthreads();           // everyone has read a[j][j] before it is overwritten
        if (t == 0) { a[j * LDA + j] = piv; dinv[j] = 1.0 / piv; }
        for (int i = j + 1 + t; i < NB; i += 256) {
            const double l = a[i * LDA + j] / piv;
            a[i * LDA + j] = l;
            col[i] = l;
        }
        __syncthreads();
        // trailing update of the lower triangle: two threads per row, interleaved columns
        {
            const int i = j + 1 + (t >> 1);
            if (i < NB) {
                const double li = col[i];
                for (int k = j + 1 + (t & 1); k <= i; k += 2) a[i * LDA + k] = fma(-li, col[k], a[i * LDA + k]);
            }
        }
        __syncthreads();
    }
    __syncthreads();
    if (fail) return;

    // Linv[:, c] for c = t: x_c = 1/L_cc ; x_i = −(Σ_{k=c}^{i−1} L[i][k]·x_k)/L_ii  stored at a[c][i]
    if (t < NB) {
        const int c = t;
        for (int i = c + 1; i < NB; ++i) {
            double s = a[i * LDA + c] * dinv[c];
            for (int k = c + 1; k < i; ++k) s = fma(a[i * LDA + k], a[c * LDA + k], s);
            a[c * LDA + i] = -s * dinv[i];
        }
    }
    __syncthreads();

    double* Wb = W + (int64_t)r0 * ld + r0;
    double* WTb = WT + (int64_t)r0 * ld + r0;
    for (int idx = t; idx < NB * NB; idx += 256) {
        const int i = idx >> 7, j = idx & 127;
        double l, w, wt;
        if (i > j) { l = a[i * LDA + j]; w = a[j * LDA + i]; wt = 0.0; }
        else if (i == j) { l = a[i * LDA + i]; w = dinv[i]; wt = dinv[i]; }
        else { l = 0.0; w = 0.0; wt = a[i * LDA + j]; }
        Kb[(int64_t)i * ld + j] = l;
        Wb[(int64_t)i * ld + j] = w;
        WTb[(int64_t)i * ld + j] = wt;
    }
}

hipError_t launch_chol_diag(double* K, double* W, double* WT, int64_t ld, int r0, int64_t* info, hipStream_t s) {
    hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(256), 0, s, K, W, WT, ld, r0, info);
    return hipGetLastError();
}

// one wave per row; lanes stride the row in 16-byte pieces; xor-tree reduction (fixed order)
__global__ void __launch_bounds__(256) trmv_kernel(const double* __restrict__ Wm, int64_t ld, const double* __restrict__ v,
                                                   double* __restrict__ out, int Np, int lower) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= Np) return;
    int kb = lower ? 0 : (row & ~1);
    int ke = lower ? row + 1 : Np;
    const double* wr = Wm + (int64_t)row * ld;
    double s = 0.0;
    for (int k = kb + 2 * lane; k < ke; k += 128) {
        s = fma(wr[k], v[k], s);
        if (k + 1 < ke) s = fma(wr[k + 1], v[k + 1], s);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[row] = s;
}

hipError_t launch_trmv(const double* Wm, int64_t ld, const double* v, double* out, int Np, int lower, hipStream_t s) {
    hipLaunchKernelGGL(trmv_kernel, dim3((Np + 3) / 4), dim3(256), 0, s, Wm, ld, v, out, Np, lower);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) nlml_terms_kernel(const double* L, int64_t ld, const double* delta,
                                                         const double* alpha, int N, double* out) {
    __shared__ double r0[256], r1[256];
    double s0 = 0.0, s1 = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) {
        s0 += 2.0 * log(L[(int64_t)i * ld + i]);
        s1 = fma(delta[i], alpha[i], s1);
    }
    r0[threadIdx.x] = s0;
    r1[threadIdx.x] = s1;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (threadIdx.x < o) { r0[threadIdx.x] += r0[threadIdx.x + o]; r1[threadIdx.x] += r1[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = r0[0]; out[1] = r1[0]; }
}

hipError_t launch_nlml_terms(const double* L, int64_t ld, const double* delta, const double* alpha, int N, double* out,
                             hipStream_t s) {
    hipLaunchKernelGGL(nlml_terms_kernel, dim3(1), dim3(256), 0, s, L, ld, delta, alpha, N, out);
    return hipGetLastError();
}

__global__ void center_kernel(const double* y, double* delta, int N, int Np, double c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Np) delta[i] = (i < N) ? y[i] - c : 0.0;
}

hipError_t launch_center(const double* y, double* delta, int N, int Np, double c, hipStream_t s) {
    hipLaunchKernelGGL(center_kernel, dim3((Np + 255) / 256), dim3(256), 0, s, y, delta, N, Np, c);
    return hipGetLastError();
}

}  // namespace abo
