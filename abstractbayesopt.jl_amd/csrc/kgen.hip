// Kernel-matrix generation: K[j][k] = sigma_f2 · kappa(||s·x_k − s·z_j||²), s = 1/ell.
//
// Reference arithmetic being replaced: [upstream KernelFunctions] kernelmatrix of
// ScaledKernel(inner ∘ ScaleTransform(1/ell), sigma_f2) (normal form built at
// src/surrogates/StandardGP.jl:41-64), evaluated for K_XX in update() (:79-83) and for K_XZ in
// posterior_mean / posterior_var (:361-379).  Inputs are multiplied by s first, then the squared
// differences are summed directly (no ||x||²+||z||²−2x·z expansion: it loses digits and buys
// nothing at d ≤ 64).
//
// Layout / mapping (HBM-write-bound for SE, VALU-bound for Matérn):
//  * output is candidate-major, k contiguous — exactly the "NT" B-operand the fp64 MFMA tile core
//    reads, so the contraction kernel needs no transpose;
//  * a workgroup owns 16 candidates and sweeps all k; lane ↔ k (two consecutive k per lane, one
//    16-byte coalesced store per candidate row), the lane's two training points live in registers,
//    the 16 scaled candidates sit in LDS and are read as wave-wide broadcasts;
//  * the posterior mean falls out of the same pass: mu_j = mean_c + Σ_k K[j][k]·alpha[k]
//    (per-lane partials in registers, fixed-order block reduction — deterministic).
#include "kgen_core.h"
#include "kgen_grad_core.h"

namespace abo {

template <int FAM>
static hipError_t launch_grad_kgen_fam(const KgenArgs& a, hipStream_t s) {
    if (a.res) {                                       // residue planes in the same pass (posterior chunks only)
        if (a.res_n != 14 || a.dlogell) return hipErrorInvalidValue;
        return launch_kgen_grad_res14(a, s);          // (writes every res_bad entry of the chunk itself)
    }
    return a.dlogell ? launch_grad_kgen_dp<FAM, true, 0>(a, s) : launch_grad_kgen_dp<FAM, false, 0>(a, s);
}

// ∂k/∂log ℓ on the squared scaled distance d2 (∂r/∂log ℓ = −r, so dK = −σ_f²·r·κ'(r)):
//   SE   σ_f² d2 e^{−d2/2}            M32  σ_f² 3 d2 e^{−√3 d}
//   M52  σ_f² (5/3) d2 (1+√5 d) e^{−√5 d}     M72  σ_f² d2 (7/5 + (7√7/5) d + (49/15) d2) e^{−√7 d}
template <int FAM>
__device__ __forceinline__ double dkappa_dlogell(double d2) {
    if constexpr (FAM == ABO_KERNEL_SE) {
        return d2 * exp_nonpos(-0.5 * d2);
    } else if constexpr (FAM == ABO_KERNEL_MATERN52) {
        const double s5 = 2.23606797749978969640917366873128;
        const double d = sqrt_pos(d2);
        return (5.0 / 3.0) * d2 * fma(s5, d, 1.0) * exp_nonpos(-s5 * d);
    } else if constexpr (FAM == ABO_KERNEL_MATERN72) {
        const double s7 = 2.64575131106459059050161575363926;
        const double d = sqrt_pos(d2);
        return d2 * fma(d2, 49.0 / 15.0, fma(d, 7.0 * s7 / 5.0, 7.0 / 5.0)) * exp_nonpos(-s7 * d);
    } else {
        const double s3 = 1.73205080756887729352744634150587;
        return 3.0 * d2 * exp_nonpos(-s3 * sqrt_pos(d2));
    }
}

// partial[block] = Σ over the block's pairs of w_ij·(Kinv[i][j] − α_i α_j)·∂K_ij/∂log ℓ, lower 128×128 tiles
// of the training set only (w = 2 for strictly-lower tiles, 1 inside diagonal tiles, which are complete).
// Same lane ↔ column mapping as kgen_kernel: 16 rows per workgroup, lanes sweep two columns each.
template <int FAM, int DP>
__global__ void __launch_bounds__(256) nlml_grad_kernel(NlmlGradArgs p) {
    __shared__ double zs[JT][DP];
    __shared__ double ai[JT];
    __shared__ double red[4];
    const int t = threadIdx.x;
    const int ib = blockIdx.x * JT;                    // first row of this workgroup
    for (int idx = t; idx < JT * DP; idx += 256) zs[idx / DP][idx % DP] = p.Xs[(int64_t)(ib + idx / DP) * DP + idx % DP];
    if (t < JT) ai[t] = p.alpha[ib + t];
    __syncthreads();
    const int kend = (ib / 128 + 1) * 128;             // columns up to the end of the row's diagonal tile
    const int diag0 = (ib / 128) * 128;
    double acc = 0.0;
    for (int k0 = 0; k0 < kend; k0 += KSTEP) {
        const int k = k0 + 2 * t;
        if (k < kend) {
            double x0[DP], x1[DP];
            const double* xp = p.Xs + (int64_t)k * DP;
#pragma unroll
            for (int c = 0; c < DP; ++c) { x0[c] = xp[c]; x1[c] = xp[DP + c]; }
            const double a0 = p.alpha[k], a1 = p.alpha[k + 1];
            const double w = (k >= diag0) ? 1.0 : 2.0;
#pragma unroll
            for (int jj = 0; jj < JT; ++jj) {
                asm volatile("" ::: "memory");
                const int i = ib + jj;
                double r0 = 0.0, r1 = 0.0;
#pragma unroll
                for (int c = 0; c < DP; ++c) {
                    const double z = zs[jj][c];
                    const double e0 = x0[c] - z, e1 = x1[c] - z;
                    r0 = fma(e0, e0, r0);
                    r1 = fma(e1, e1, r1);
                }
                const double* kr = p.Kinv + (int64_t)i * p.ld + k;
                const double m0 = kr[0] - ai[jj] * a0, m1 = kr[1] - ai[jj] * a1;
                const double g0 = (i < p.N && k < p.N) ? m0 * dkappa_dlogell<FAM>(r0) : 0.0;
                const double g1 = (i < p.N && k + 1 < p.N) ? m1 * dkappa_dlogell<FAM>(r1) : 0.0;
                acc = fma(w, g0 + g1, acc);
            }
        }
    }
    const int lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (t == 0) p.partial[blockIdx.x] = p.sigma_f2 * (((red[0] + red[1]) + red[2]) + red[3]);
}

// out[0] = Σ partial (fixed order), out[1] = Σ_{i<N} Kinv[i][i], out[2] = αᵀα, out[3] = αᵀδ
__global__ void __launch_bounds__(256) nlml_grad_finish_kernel(NlmlGradArgs p, int nblocks) {
    __shared__ double r[4][256];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) s0 += p.partial[i];
    for (int i = (int)threadIdx.x; i < p.N; i += 256) {
        s1 += p.Kinv[(int64_t)i * p.ld + i];
        s2 = fma(p.alpha[i], p.alpha[i], s2);
        s3 = fma(p.alpha[i], p.delta[i], s3);
    }
    r[0][threadIdx.x] = s0; r[1][threadIdx.x] = s1; r[2][threadIdx.x] = s2; r[3][threadIdx.x] = s3;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (threadIdx.x < o)
            for (int q = 0; q < 4; ++q) r[q][threadIdx.x] += r[q][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 4) p.out[threadIdx.x] = r[threadIdx.x][0];
}


// d > 32: the same reduction with the coordinates in slabs of 32 (see kgen_wide_kernel)
template <int FAM>
__global__ void __launch_bounds__(256) nlml_grad_wide_kernel(NlmlGradArgs p) {
    __shared__ double zs[JT][32];
    __shared__ double ai[JT];
    __shared__ double red[4];
    const int t = threadIdx.x;
    const int ib = blockIdx.x * JT;
    if (t < JT) ai[t] = p.alpha[ib + t];
    const int kend = (ib / 128 + 1) * 128;
    const int diag0 = (ib / 128) * 128;
    double acc = 0.0;
    for (int k0 = 0; k0 < kend; k0 += KSTEP) {
        const int k = k0 + 2 * t;
        const bool live = k < kend;
        double r0[JT], r1[JT];
#pragma unroll
        for (int jj = 0; jj < JT; ++jj) { r0[jj] = 0.0; r1[jj] = 0.0; }
        for (int c0 = 0; c0 < p.dp; c0 += 32) {
            __syncthreads();
            for (int idx = t; idx < JT * 32; idx += 256) zs[idx / 32][idx % 32] = p.Xs[(int64_t)(ib + idx / 32) * p.dp + c0 + idx % 32];
            __syncthreads();
            if (live) {
                double x0[32], x1[32];
                const double* xp = p.Xs + (int64_t)k * p.dp + c0;
#pragma unroll
                for (int c = 0; c < 32; ++c) { x0[c] = xp[c]; x1[c] = xp[p.dp + c]; }
#pragma unroll
                for (int jj = 0; jj < JT; ++jj) {
                    asm volatile("" ::: "memory");
                    double a = r0[jj], b = r1[jj];
#pragma unroll
                    for (int c = 0; c < 32; ++c) {
                        const double z = zs[jj][c];
                        const double e0 = x0[c] - z, e1 = x1[c] - z;
                        a = fma(e0, e0, a);
                        b = fma(e1, e1, b);
                    }
                    r0[jj] = a; r1[jj] = b;
                }
            }
        }
        if (live) {
            const double a0 = p.alpha[k], a1 = p.alpha[k + 1];
            const double w = (k >= diag0) ? 1.0 : 2.0;
#pragma unroll
            for (int jj = 0; jj < JT; ++jj) {
                const int i = ib + jj;
                const double* kr = p.Kinv + (int64_t)i * p.ld + k;
                const double m0 = kr[0] - ai[jj] * a0, m1 = kr[1] - ai[jj] * a1;
                const double g0 = (i < p.N && k < p.N) ? m0 * dkappa_dlogell<FAM>(r0[jj]) : 0.0;
                const double g1 = (i < p.N && k + 1 < p.N) ? m1 * dkappa_dlogell<FAM>(r1[jj]) : 0.0;
                acc = fma(w, g0 + g1, acc);
            }
        }
    }
    const int lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (t == 0) p.partial[blockIdx.x] = p.sigma_f2 * (((red[0] + red[1]) + red[2]) + red[3]);
}

template <int FAM>
static hipError_t launch_grad_fam(const NlmlGradArgs& a, hipStream_t s) {
    dim3 grid(a.Np / JT), block(256);
    if (a.dp > 32) {
        if (a.dp % 32) return hipErrorInvalidValue;
        hipLaunchKernelGGL((nlml_grad_wide_kernel<FAM>), grid, block, 0, s, a);
        return hipGetLastError();
    }
    switch (a.dp) {
        case 1: hipLaunchKernelGGL((nlml_grad_kernel<FAM, 1>), grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL((nlml_grad_kernel<FAM, 2>), grid, block, 0, s, a); break;
        case 4: hipLaunchKernelGGL((nlml_grad_kernel<FAM, 4>), grid, block, 0, s, a); break;
        case 8: hipLaunchKernelGGL((nlml_grad_kernel<FAM, 8>), grid, block, 0, s, a); break;
        case 16: hipLaunchKernelGGL((nlml_grad_kernel<FAM, 16>), grid, block, 0, s, a); break;
        case 32: hipLaunchKernelGGL((nlml_grad_kernel<FAM, 32>), grid, block, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_nlml_grad(const NlmlGradArgs& a, hipStream_t s) {
    hipError_t e;
    switch (a.family) {
        case ABO_KERNEL_SE: e = launch_grad_fam<ABO_KERNEL_SE>(a, s); break;
        case ABO_KERNEL_MATERN52: e = launch_grad_fam<ABO_KERNEL_MATERN52>(a, s); break;
        case ABO_KERNEL_MATERN72: e = launch_grad_fam<ABO_KERNEL_MATERN72>(a, s); break;
        case ABO_KERNEL_MATERN32: e = launch_grad_fam<ABO_KERNEL_MATERN32>(a, s); break;
        default: return hipErrorInvalidValue;
    }
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(nlml_grad_finish_kernel, dim3(1), dim3(256), 0, s, a, a.Np / JT);
    return hipGetLastError();
}

// partial[block] = sum of w (Kinv[i][k] - alpha_i alpha_k) D[i][k] over the block's 16 rows and the columns up to the end
// of their diagonal 128-tile (w = 2 left of the diagonal tile, 1 inside it): the dK/dlog(ell) term with D given as a matrix.
__global__ void __launch_bounds__(256) nlml_grad_matrix_kernel(NlmlGradArgs p, const double* __restrict__ D, int64_t ldd) {
    __shared__ double red[4];
    const int t = threadIdx.x;
    const int ib = blockIdx.x * JT;
    const int kend = (ib / 128 + 1) * 128, diag0 = (ib / 128) * 128;
    double acc = 0.0;
    for (int k = t; k < kend; k += 256) {
        if (k >= p.N) continue;
        const double ak = p.alpha[k], w = k >= diag0 ? 1.0 : 2.0;
        for (int jj = 0; jj < JT; ++jj) {
            const int i = ib + jj;
            if (i < p.N) acc = fma(w * (p.Kinv[(int64_t)i * p.ld + k] - p.alpha[i] * ak), D[(int64_t)i * ldd + k], acc);
        }
    }
    const int lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (t == 0) p.partial[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

hipError_t launch_nlml_grad_matrix(const NlmlGradArgs& a, const double* D, int64_t ldd, hipStream_t s) {
    hipLaunchKernelGGL(nlml_grad_matrix_kernel, dim3(a.Np / JT), dim3(256), 0, s, a, D, ldd);
    hipLaunchKernelGGL(nlml_grad_finish_kernel, dim3(1), dim3(256), 0, s, a, a.Np / JT);
    return hipGetLastError();
}

// test hook: out[i] = kappa(family, d2[i])
__global__ void kappa_test_kernel(int family, const double* d2, double* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = d2[i];
    double v;
    if (family == ABO_KERNEL_SE) v = kappa_eval<ABO_KERNEL_SE>(x);
    else if (family == ABO_KERNEL_MATERN52) v = kappa_eval<ABO_KERNEL_MATERN52>(x);
    else if (family == ABO_KERNEL_MATERN72) v = kappa_eval<ABO_KERNEL_MATERN72>(x);
    else v = kappa_eval<ABO_KERNEL_MATERN32>(x);
    out[i] = v;
}

hipError_t launch_kappa_test(int family, const double* d2, double* out, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(kappa_test_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, family, d2, out, n);
    return hipGetLastError();
}

// d > 32 (dp = d rounded up to a multiple of 32): the same lane ↔ k map and the same c = 0..d−1 summation order as
// kgen_kernel, with the coordinates taken in slabs of 32 — the lane's two training points hold one slab in registers
// (64 doubles), the 16 candidates' slab sits in LDS, the 2·16 squared distances are carried across slabs.  The
// reference is dimension-agnostic (src/surrogates/StandardGP.jl:79-83); this path keeps the library so.
constexpr int SLAB = 32;
template <int FAM>
__global__ void __launch_bounds__(256) kgen_wide_kernel(KgenArgs p) {
    __shared__ double zs[JT][SLAB];
    __shared__ double red[4][JT];
    const int t = threadIdx.x;
    const int jb = blockIdx.x * JT;
    double mu[JT];
#pragma unroll
    for (int jj = 0; jj < JT; ++jj) mu[jj] = 0.0;
    for (int k0 = 0; k0 < p.Np; k0 += KSTEP) {
        const int k = k0 + 2 * t;
        const bool live = k < p.Np;                      // the barriers below are taken by every thread
        double r0[JT], r1[JT];
#pragma unroll
        for (int jj = 0; jj < JT; ++jj) { r0[jj] = 0.0; r1[jj] = 0.0; }
        for (int c0 = 0; c0 < p.dp; c0 += SLAB) {
            __syncthreads();
            for (int idx = t; idx < JT * SLAB; idx += 256) {
                const int jj = idx / SLAB, c = c0 + idx % SLAB;
                const int64_t gj = p.j0 + jb + jj;
                zs[jj][idx % SLAB] = (c < p.d && gj < p.M) ? p.Z[gj * p.d + c] * p.s : 0.0;
            }
            __syncthreads();
            if (live) {
                double x0[SLAB], x1[SLAB];
                const double* xp = p.Xs + (int64_t)k * p.dp + c0;
#pragma unroll
                for (int c = 0; c < SLAB; c += 2) {
                    const d2_t v0 = *reinterpret_cast<const d2_t*>(xp + c);
                    const d2_t v1 = *reinterpret_cast<const d2_t*>(xp + p.dp + c);
                    x0[c] = v0[0]; x0[c + 1] = v0[1];
                    x1[c] = v1[0]; x1[c + 1] = v1[1];
                }
#pragma unroll
                for (int jj = 0; jj < JT; ++jj) {
                    asm volatile("" ::: "memory");
                    double a = r0[jj], b = r1[jj];
#pragma unroll
                    for (int c = 0; c < SLAB; ++c) {
                        const double z = zs[jj][c];
                        const double e0 = x0[c] - z, e1 = x1[c] - z;
                        a = fma(e0, e0, a);
                        b = fma(e1, e1, b);
                    }
                    r0[jj] = a; r1[jj] = b;
                }
            }
        }
        if (live) {
            const double s0 = k < p.N ? p.sigma_f2 : 0.0, s1 = (k + 1) < p.N ? p.sigma_f2 : 0.0;
            double a0 = 0.0, a1 = 0.0;
            if (p.alpha) { a0 = p.alpha[k]; a1 = p.alpha[k + 1]; }
#pragma unroll
            for (int jj = 0; jj < JT; ++jj) {
                const bool okj = (p.j0 + jb + jj) < p.M;
                const double v0 = okj ? s0 * kappa_eval<FAM>(r0[jj]) : 0.0;
                const double v1 = okj ? s1 * kappa_eval<FAM>(r1[jj]) : 0.0;
                if (p.Kout) *reinterpret_cast<d2_t*>(p.Kout + (int64_t)(jb + jj) * p.ldk + k) = d2_t{v0, v1};
                mu[jj] = fma(v1, a1, fma(v0, a0, mu[jj]));
            }
        }
    }
    if (p.mu == nullptr) return;
    const int lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int jj = 0; jj < JT; ++jj) {
        double v = mu[jj];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave][jj] = v;
    }
    __syncthreads();
    if (t < JT) p.mu[jb + t] = p.mean_c + (((red[0][t] + red[1][t]) + red[2][t]) + red[3][t]);
}

template <int FAM>
static hipError_t launch_fam(const KgenArgs& a, hipStream_t s) {
    dim3 grid(a.Mc / JT), block(256);
    if (a.dp > 32) {
        if (a.dp % SLAB) return hipErrorInvalidValue;
        hipLaunchKernelGGL((kgen_wide_kernel<FAM>), grid, block, 0, s, a);
        return hipGetLastError();
    }
    if (a.res) {
        if (a.res_n != 14) return hipErrorInvalidValue;          // kgen_writes_residues() told the caller not to ask
        return launch_kgen_res14(a, s);               // (writes every res_bad entry of the chunk itself)
    }
    return launch_kgen_dp<FAM, 0>(a, s);
}

hipError_t launch_kgen(const KgenArgs& a, hipStream_t s) {
    if (a.Mc <= 0) return hipSuccess;
    if (a.pt > 1) {                                    // gradient-enhanced GP
        switch (a.family) {
            case ABO_KERNEL_SE: return launch_grad_kgen_fam<ABO_KERNEL_SE>(a, s);
            case ABO_KERNEL_MATERN52: return launch_grad_kgen_fam<ABO_KERNEL_MATERN52>(a, s);
            case ABO_KERNEL_MATERN72: return launch_grad_kgen_fam<ABO_KERNEL_MATERN72>(a, s);
            default: return hipErrorInvalidValue;
        }
    }
    switch (a.family) {
        case ABO_KERNEL_SE: return launch_fam<ABO_KERNEL_SE>(a, s);
        case ABO_KERNEL_MATERN52: return launch_fam<ABO_KERNEL_MATERN52>(a, s);
        case ABO_KERNEL_MATERN72: return launch_fam<ABO_KERNEL_MATERN72>(a, s);
        case ABO_KERNEL_MATERN32: return launch_fam<ABO_KERNEL_MATERN32>(a, s);
        default: return hipErrorInvalidValue;
    }
}

// Column `col` of a resident K_ZX: one kernel evaluation per candidate (the training point just appended).
template <int FAM>
__global__ void __launch_bounds__(256) cand_newcol_kernel(const double* __restrict__ Xs, const double* __restrict__ Z,
                                                           double* __restrict__ Kzx, int64_t ld, int64_t M, int col, int d,
                                                           int dp, double s, double sigma_f2) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const double* x = Xs + (int64_t)col * dp;
    const double* z = Z + j * d;
    double r = 0.0;
    for (int c = 0; c < d; ++c) {
        const double e = x[c] - z[c] * s;
        r = fma(e, e, r);
    }
    Kzx[j * ld + col] = sigma_f2 * kappa_eval<FAM>(r);
}

hipError_t launch_cand_newcol(const double* Xs, const double* Z, double* Kzx, int64_t ld, int64_t M, int col, int d, int dp,
                              int family, double s, double sigma_f2, hipStream_t st) {
    if (M <= 0) return hipSuccess;
    dim3 grid((unsigned)((M + 255) / 256)), block(256);
    switch (family) {
        case ABO_KERNEL_SE: hipLaunchKernelGGL((cand_newcol_kernel<ABO_KERNEL_SE>), grid, block, 0, st, Xs, Z, Kzx, ld, M, col, d, dp, s, sigma_f2); break;
        case ABO_KERNEL_MATERN52: hipLaunchKernelGGL((cand_newcol_kernel<ABO_KERNEL_MATERN52>), grid, block, 0, st, Xs, Z, Kzx, ld, M, col, d, dp, s, sigma_f2); break;
        case ABO_KERNEL_MATERN72: hipLaunchKernelGGL((cand_newcol_kernel<ABO_KERNEL_MATERN72>), grid, block, 0, st, Xs, Z, Kzx, ld, M, col, d, dp, s, sigma_f2); break;
        case ABO_KERNEL_MATERN32: hipLaunchKernelGGL((cand_newcol_kernel<ABO_KERNEL_MATERN32>), grid, block, 0, st, Xs, Z, Kzx, ld, M, col, d, dp, s, sigma_f2); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// The same column for a gradient-enhanced model: training row (point `pt`, output q) against the FUNCTION value of every
// candidate — cov(f(z), f(x)) = σ_f²φ for q = 0, cov(f(z), ∂f(x)/∂x_c) = σ_f²φ'·2e_c/ℓ, e = (x − z)/ℓ, for q = c + 1
// (the qc < 0 branch of kgen_grad_kernel; same arithmetic).
template <int FAM>
__global__ void __launch_bounds__(256) cand_newcol_grad_kernel(const double* __restrict__ Xs, const double* __restrict__ Z,
                                                                double* __restrict__ Kzx, int64_t ld, int64_t M, int col, int pt,
                                                                int q, int d, int dp, double s, double sigma_f2) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const double* x = Xs + (int64_t)pt * dp;
    const double* z = Z + j * d;
    double u = 0.0, ep = 0.0;
    for (int c = 0; c < d; ++c) {
        const double e = x[c] - z[c] * s;
        u = fma(e, e, u);
        if (c == q - 1) ep = e;
    }
    double f0, g0, h0;
    phi_derivs<FAM>(u, f0, g0, h0);
    Kzx[j * ld + col] = sigma_f2 * (q == 0 ? f0 : 2.0 * s * g0 * ep);
}

hipError_t launch_cand_newcol_grad(const double* Xs, const double* Z, double* Kzx, int64_t ld, int64_t M, int col, int pt, int q,
                                   int d, int dp, int family, double s, double sigma_f2, hipStream_t st) {
    if (M <= 0) return hipSuccess;
    dim3 grid((unsigned)((M + 255) / 256)), block(256);
    switch (family) {
        case ABO_KERNEL_SE: hipLaunchKernelGGL((cand_newcol_grad_kernel<ABO_KERNEL_SE>), grid, block, 0, st, Xs, Z, Kzx, ld, M, col, pt, q, d, dp, s, sigma_f2); break;
        case ABO_KERNEL_MATERN52: hipLaunchKernelGGL((cand_newcol_grad_kernel<ABO_KERNEL_MATERN52>), grid, block, 0, st, Xs, Z, Kzx, ld, M, col, pt, q, d, dp, s, sigma_f2); break;
        case ABO_KERNEL_MATERN72: hipLaunchKernelGGL((cand_newcol_grad_kernel<ABO_KERNEL_MATERN72>), grid, block, 0, st, Xs, Z, Kzx, ld, M, col, pt, q, d, dp, s, sigma_f2); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

__global__ void diag_fix_kernel(double* K, int64_t ld, int N, int Np, double noise, int64_t* info_reset) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && info_reset) *info_reset = 0;          // the factorisation's status word: the chain behind this launch reads it
    if (i >= Np) return;
    if (i < N) K[(int64_t)i * ld + i] += noise;
    else K[(int64_t)i * ld + i] = 1.0;
}

hipError_t launch_diag_fix(double* K, int64_t ld, int N, int Np, double noise, hipStream_t s, int64_t* info_reset) {
    hipLaunchKernelGGL(diag_fix_kernel, dim3((Np + 255) / 256), dim3(256), 0, s, K, ld, N, Np, noise, info_reset);
    return hipGetLastError();
}

__global__ void set_diag_kernel(double* A, int64_t ld, int lo, int hi, double v) {
    const int i = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < hi) A[(int64_t)i * ld + i] = v;
}

hipError_t launch_set_diag(double* A, int64_t ld, int lo, int hi, double v, hipStream_t s) {
    if (hi <= lo) return hipSuccess;
    hipLaunchKernelGGL(set_diag_kernel, dim3((hi - lo + 255) / 256), dim3(256), 0, s, A, ld, lo, hi, v);
    return hipGetLastError();
}

__global__ void scale_points_kernel(const double* X, double* Xs, int N, int Np, int d, int dp, double s) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)Np * dp) return;
    const int i = (int)(idx / dp), c = (int)(idx % dp);
    Xs[idx] = (i < N && c < d) ? X[(int64_t)i * d + c] * s : 0.0;
}

__global__ void fit_prep_kernel(const double* Xsrc, const double* ysrc, double* Xraw, double* ybuf, double* Xs, double* delta, double* alpha,
                                int N, int Np, int d, int dp, double s, double mean_c) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (int64_t)Np * dp) {
        const int i = (int)(idx / dp), c = (int)(idx % dp);
        double v = 0.0;
        if (i < N && c < d) {
            v = Xsrc[(int64_t)i * d + c];
            if (Xsrc != Xraw) Xraw[(int64_t)i * d + c] = v;
        }
        Xs[idx] = v * s;
    }
    if (idx < Np) {
        double v = 0.0;
        if (idx < N) {
            v = ysrc[idx];
            if (ysrc != ybuf) ybuf[idx] = v;
            v -= mean_c;
        }
        delta[idx] = v;
        alpha[idx] = 0.0;
    }
}

hipError_t launch_fit_prep(const double* Xsrc, const double* ysrc, double* Xraw, double* ybuf, double* Xs, double* delta, double* alpha,
                           int N, int Np, int d, int dp, double s, double mean_c, hipStream_t st) {
    const int64_t n = (int64_t)Np * dp;
    hipLaunchKernelGGL(fit_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, Xsrc, ysrc, Xraw, ybuf, Xs, delta, alpha, N, Np,
                       d, dp, s, mean_c);
    return hipGetLastError();
}

// the observation a bordered append adds, handed over in the kernel arguments: raw and scaled coordinates, target, centred target and
// the reset status word in ONE launch (instead of two pageable host-to-device copies, two one-thread kernels and a memset)
struct AppendPoint { double x[APPEND_POINT_MAXD]; };
__global__ void append_point_kernel(AppendPoint pt, int d, int dp, double s, double y, double mean_c, double* Xraw_row, double* Xs_row,
                                    double* y_at, double* delta_at, int64_t* info) {
    const int c = threadIdx.x;
    if (c < d) Xraw_row[c] = pt.x[c];
    if (c < dp) Xs_row[c] = c < d ? pt.x[c] * s : 0.0;
    if (c == 0) { *y_at = y; *delta_at = y - mean_c; *info = 0; }
}

hipError_t launch_append_point(const double* x_host, int d, int dp, double s, double y, double mean_c, double* Xraw_row, double* Xs_row,
                               double* y_at, double* delta_at, int64_t* info, hipStream_t st) {
    if (d > APPEND_POINT_MAXD || dp > APPEND_POINT_MAXD) return hipErrorInvalidValue;
    AppendPoint pt{};
    for (int c = 0; c < d; ++c) pt.x[c] = x_host[c];
    hipLaunchKernelGGL(append_point_kernel, dim3(1), dim3(APPEND_POINT_MAXD), 0, st, pt, d, dp, s, y, mean_c, Xraw_row, Xs_row, y_at, delta_at, info);
    return hipGetLastError();
}

hipError_t launch_scale_points(const double* X, double* Xs, int N, int Np, int d, int dp, double s, hipStream_t st) {
    const int64_t n = (int64_t)Np * dp;
    hipLaunchKernelGGL(scale_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, X, Xs, N, Np, d, dp, s);
    return hipGetLastError();
}

}  // namespace abo
