// The gradient-enhanced generator (kgen_grad_kernel), shared by kgen.hip (fp64 output, d/dlog ℓ variant) and kgen_grad_res.hip (the same
// pass also writing the int8-residue engine's planes) — two translation units so that the instantiations compile in parallel.
#pragma once
#include "kgen_core.h"

namespace abo {

// ---- gradient-enhanced GP (GradientGP): multi-output kernel with analytic derivatives ----------------------
// Reference: gradKernel (src/surrogates/GradientGP.jl:573-606) evaluates these blocks with nested
// ForwardDiff.derivative calls; here φ, φ', φ'' of k = σ_f²φ(u), u = ‖x−z‖²/ℓ², are closed forms
// (GradientGP.jl:176-182, :400-407):   with e = (x_i − z)/ℓ  (second argument minus first, scaled)
//     cov(f(z),      f(x_i))        = σ_f² φ
//     cov(f(z),      ∂f(x_i)/∂x_c') = σ_f² φ'·2e_c'/ℓ
//     cov(∂f(z)/∂z_c, f(x_i))       = −σ_f² φ'·2e_c/ℓ
//     cov(∂f(z)/∂z_c, ∂f(x_i)/∂x_c') = −σ_f² (4φ''e_c e_c' + 2φ'δ_cc')/ℓ²
template <int FAM>
__device__ __forceinline__ void phi_derivs(double u, double& p0, double& p1, double& p2) {
    if constexpr (FAM == ABO_KERNEL_SE) {
        const double e = exp_nonpos(-0.5 * u);
        p0 = e; p1 = -0.5 * e; p2 = 0.25 * e;
    } else if constexpr (FAM == ABO_KERNEL_MATERN52) {
        const double a = 2.23606797749978969640917366873128;
        const double r = sqrt_pos(u), e = exp_nonpos(-a * r), t = fma(a, r, 1.0);
        p0 = fma(u, 5.0 / 3.0, t) * e; p1 = (-5.0 / 6.0) * t * e; p2 = (25.0 / 12.0) * e;
    } else {
        const double a = 2.64575131106459059050161575363926;
        const double r = sqrt_pos(u), e = exp_nonpos(-a * r), t = fma(a, r, 1.0);
        p0 = fma(u * r, 7.0 * a / 15.0, fma(u, 14.0 / 5.0, t)) * e;
        p1 = (-7.0 / 10.0) * fma(u, 7.0 / 3.0, t) * e;
        p2 = (49.0 / 60.0) * t * e;
    }
}

// d/dlog(ell) of the same blocks.  With du/dlog(ell) = -2u, de/dlog(ell) = -e, d(1/ell)/dlog(ell) = -1/ell every block
// keeps its form with (phi, phi1, phi2) = (φ, φ', φ'') replaced by
//     phi  -> -2u phi1          phi1 -> -2u phi2 - 2 phi1          phi2 -> -(2u phi3 + 4 phi2)
// (u·phi3 is finite at u = 0 for every family: the Matérn third derivative grows like 1/sqrt(u) there).
template <int FAM>
__device__ __forceinline__ void phi_derivs_dlogell(double u, double& p0, double& p1, double& p2) {
    if constexpr (FAM == ABO_KERNEL_SE) {
        const double e = exp_nonpos(-0.5 * u);
        p0 = u * e; p1 = fma(-0.5, u, 1.0) * e; p2 = fma(0.25, u, -1.0) * e;
    } else if constexpr (FAM == ABO_KERNEL_MATERN52) {
        const double a = 2.23606797749978969640917366873128;
        const double r = sqrt_pos(u), e = exp_nonpos(-a * r), t = fma(a, r, 1.0);
        p0 = (5.0 / 3.0) * u * t * e;
        p1 = fma(5.0 / 3.0, t, (-25.0 / 6.0) * u) * e;
        p2 = fma(25.0 / 12.0 * a, r, -25.0 / 3.0) * e;
    } else {
        const double a = 2.64575131106459059050161575363926;
        const double r = sqrt_pos(u), e = exp_nonpos(-a * r), t = fma(a, r, 1.0);
        p0 = (7.0 / 5.0) * u * fma(u, 7.0 / 3.0, t) * e;
        p1 = fma(7.0 / 5.0, t, fma(49.0 / 15.0, u, (-49.0 / 30.0) * u * t)) * e;
        p2 = fma(343.0 / 60.0, u, (-49.0 / 15.0) * t) * e;
    }
}

// Rows: candidate-output pairs.  Row index g = j0 + (row in chunk); with point_major == 0 it is by outputs
// (q = g / M, j = g % M — MOInputIsotopicByOutputs, what the reference's posterior_grad_* return), with
// point_major == 1 it is j = g / pc, q = g % pc (all outputs of a point adjacent: per-point covariance blocks).
// Columns: training rows in the library's POINT-MAJOR factor order r = i·pt + q' (all outputs of training point i
// adjacent), so that a new observation appends pt rows at the END of the factor (bordered updates, abo_append_grad);
// the reference's by-outputs order (prep_output, GradientGP.jl:893-895) stays at the ABI.  rvalid (when > 0) is the number of
// valid training rows — a point whose outputs are only partly appended yet.
// RES (0 or 14): also write the int8-residue engine's planes of the chunk (kgen_core.h: the same byte layout as the StandardGP
// generator) — images of D·K·E with the engine's exact power-of-two scalings: a derivative training row (k % pt != 0) and a
// derivative candidate output each take 2^-res_ktg on top of 2^res_sK (ozaki.hip: oz_quant_kernel's kper / rmode arguments), so the
// fp64 chunk and the separate quantiser pass over it are not needed when only the contraction reads it.
template <int FAM, int DP, bool DLOGELL, int RES>
__global__ void __launch_bounds__(256) kgen_grad_kernel(KgenArgs p) {
    __shared__ double zs[JT][DP];
    __shared__ int zq[JT];
    __shared__ double red[4][JT];
    __shared__ int bad[JT];          // RES: rows with a non-finite value (kgen_core.h)
    const int t = threadIdx.x;
    const int jb = blockIdx.x * JT;
    if (t < JT) bad[t] = 0;
    const int64_t rows_total = p.M * p.pc;              // candidate rows overall
    for (int idx = t; idx < JT * DP; idx += 256) {
        const int jj = idx / DP, c = idx % DP;
        const int64_t g = p.j0 + jb + jj;
        const int64_t j = p.point_major ? g / p.pc : g % p.M;
        zs[jj][c] = (c < p.d && g < rows_total) ? p.Z[j * p.d + c] * p.s : 0.0;
    }
    if (t < JT) {
        const int64_t g = p.j0 + jb + t;
        zq[t] = (g < rows_total) ? (int)(p.point_major ? g % p.pc : g / p.M) : -1;
    }
    __syncthreads();
    const double il = p.s;                              // 1/ℓ
    double mu[JT];
#pragma unroll
    for (int jj = 0; jj < JT; ++jj) mu[jj] = 0.0;
    const int R = p.rvalid > 0 ? p.rvalid : p.N * p.pt;  // valid training rows
    // scale of a value's fixed-point image by how many of (training row, candidate output) are derivatives: 0, 1 or 2
    const double rsc0 = RES != 0 ? __builtin_ldexp(1.0, p.res_sK) : 0.0;
    const double rsc1 = RES != 0 ? __builtin_ldexp(1.0, p.res_sK - p.res_ktg) : 0.0;
    const double rsc2 = RES != 0 ? __builtin_ldexp(1.0, p.res_sK - 2 * p.res_ktg) : 0.0;
    for (int k0 = 0; k0 < p.Np; k0 += KSTEP) {
        const int k = k0 + 2 * t;
        if (k < p.Np) {
            double x0[DP], x1[DP];
            const int i0 = k / p.pt, i1 = (k + 1) / p.pt;
            const int qp0 = k % p.pt - 1, qp1 = (k + 1) % p.pt - 1;    // training output's coordinate (−1: f itself)
            const bool ok0 = k < R, ok1 = (k + 1) < R;
#pragma unroll
            for (int c = 0; c < DP; ++c) {
                x0[c] = ok0 ? p.Xs[(int64_t)i0 * DP + c] : 0.0;
                x1[c] = ok1 ? p.Xs[(int64_t)i1 * DP + c] : 0.0;
            }
            double a0 = 0.0, a1 = 0.0;
            if (p.alpha) { a0 = p.alpha[k]; a1 = p.alpha[k + 1]; }
#pragma unroll
            for (int jj = 0; jj < JT; ++jj) {
                asm volatile("" ::: "memory");
                const int q = zq[jj];                    // row's output: −1 invalid, 0 f, c+1 gradient
                const int qc = q - 1;
                double u0 = 0.0, u1 = 0.0, ec0 = 0.0, ec1 = 0.0, ep0 = 0.0, ep1 = 0.0;
#pragma unroll
                for (int c = 0; c < DP; ++c) {
                    const double z = zs[jj][c];
                    const double e0 = x0[c] - z, e1 = x1[c] - z;
                    u0 = fma(e0, e0, u0);
                    u1 = fma(e1, e1, u1);
                    if (c == qc) { ec0 = e0; ec1 = e1; }
                    if (c == qp0) ep0 = e0;
                    if (c == qp1) ep1 = e1;
                }
                double f0, g0, h0, f1, g1, h1;
                if constexpr (DLOGELL) {
                    phi_derivs_dlogell<FAM>(u0, f0, g0, h0);
                    phi_derivs_dlogell<FAM>(u1, f1, g1, h1);
                } else {
                    phi_derivs<FAM>(u0, f0, g0, h0);
                    phi_derivs<FAM>(u1, f1, g1, h1);
                }
                double v0, v1;
                if (qc < 0) {
                    v0 = qp0 < 0 ? f0 : 2.0 * il * g0 * ep0;
                    v1 = qp1 < 0 ? f1 : 2.0 * il * g1 * ep1;
                } else {
                    v0 = qp0 < 0 ? -2.0 * il * g0 * ec0 : -il * il * fma(4.0 * h0, ec0 * ep0, qp0 == qc ? 2.0 * g0 : 0.0);
                    v1 = qp1 < 0 ? -2.0 * il * g1 * ec1 : -il * il * fma(4.0 * h1, ec1 * ep1, qp1 == qc ? 2.0 * g1 : 0.0);
                }
                v0 = (ok0 && q >= 0) ? p.sigma_f2 * v0 : 0.0;
                v1 = (ok1 && q >= 0) ? p.sigma_f2 * v1 : 0.0;
                if (p.Kout) *reinterpret_cast<d2_t*>(p.Kout + (int64_t)(jb + jj) * p.ldk + k) = d2_t{v0, v1};
                if constexpr (RES != 0) {
                    if (!(__builtin_fabs(v0) < 1.0e300) || !(__builtin_fabs(v1) < 1.0e300)) bad[jj] = 1;
                    const unsigned koff = (unsigned)(((k >> 6) << 14) + (k & 63));
                    const int64_t rowoff = ((int64_t)((jb + jj) >> 8) * (p.res_ld >> 6)) * 16384 + ((jb + jj) & 255) * 64;
                    const int nd0 = (qc >= 0 ? 1 : 0) + (qp0 >= 0 ? 1 : 0), nd1 = (qc >= 0 ? 1 : 0) + (qp1 >= 0 ? 1 : 0);
                    oz_residue_pair<RES>(__builtin_rint(v0 * (nd0 == 0 ? rsc0 : (nd0 == 1 ? rsc1 : rsc2))),
                                         __builtin_rint(v1 * (nd1 == 0 ? rsc0 : (nd1 == 1 ? rsc1 : rsc2))), [&](int l, unsigned two) {
                        int8_t* plane = p.res + (int64_t)l * p.res_plane + rowoff;        // uniform
                        *reinterpret_cast<unsigned short*>(plane + koff) = (unsigned short)two;
                    });
                }
                mu[jj] = fma(v1, a1, fma(v0, a0, mu[jj]));
            }
        }
    }
    if constexpr (RES != 0) {
        __syncthreads();
        if (t < JT) p.res_bad[jb + t] = bad[t];
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
    if (t < JT) {
        const int q = zq[t];
        p.mu[jb + t] = (q >= 0 ? p.mean_vec[q] : 0.0) + (((red[0][t] + red[1][t]) + red[2][t]) + red[3][t]);
    }
}


// d > 32 (dp = d rounded up to a multiple of 32; GradientGP.jl:617-639 has no dimension limit): the same blocks with the coordinates
// taken in slabs of 32, as kgen_wide_kernel does for the StandardGP — per (row, training row) pair the squared distance and the two
// coordinate differences a derivative block needs (e_c of the row's output, e_c' of the training row's) are carried across the slabs;
// same c = 0 … d−1 summation order as kgen_grad_kernel.  fp64 output only (the int8 engine then quantises the chunk in a pass of
// its own: oz_quant_kernel).
template <int FAM, bool DLOGELL>
__global__ void __launch_bounds__(256) kgen_grad_wide_kernel(KgenArgs p) {
    constexpr int GSLAB = 32;                            // coordinates per slab (dp is a multiple of it)
    constexpr int JH = 8;                                // rows taken together: their carried state (6 doubles per row and training row
                                                         // pair) and one slab of the lane's two training points fill the register file
    __shared__ double zs[JT][GSLAB];
    __shared__ int zq[JT];
    __shared__ double red[4][JT];
    const int t = threadIdx.x;
    const int jb = blockIdx.x * JT;
    const int64_t rows_total = p.M * p.pc;
    if (t < JT) {
        const int64_t g = p.j0 + jb + t;
        zq[t] = (g < rows_total) ? (int)(p.point_major ? g % p.pc : g / p.M) : -1;
    }
    const double il = p.s;
    double mu[JT];
#pragma unroll
    for (int jj = 0; jj < JT; ++jj) mu[jj] = 0.0;
    const int R = p.rvalid > 0 ? p.rvalid : p.N * p.pt;
    for (int k0 = 0; k0 < p.Np; k0 += KSTEP) {
        const int k = k0 + 2 * t;
        const bool live = k < p.Np;                      // the barriers below are taken by every thread
        const int i0 = live ? k / p.pt : 0, i1 = live ? (k + 1) / p.pt : 0;
        const int qp0 = k % p.pt - 1, qp1 = (k + 1) % p.pt - 1;
        const bool ok0 = live && k < R, ok1 = live && (k + 1) < R;
        double a0 = 0.0, a1 = 0.0;
        if (live && p.alpha) { a0 = p.alpha[k]; a1 = p.alpha[k + 1]; }
#pragma unroll
        for (int h = 0; h < JT / JH; ++h) {
            double u0[JH], u1[JH], ec0[JH], ec1[JH], ep0[JH], ep1[JH];
#pragma unroll
            for (int jj = 0; jj < JH; ++jj) { u0[jj] = u1[jj] = ec0[jj] = ec1[jj] = ep0[jj] = ep1[jj] = 0.0; }
            for (int c0 = 0; c0 < p.dp; c0 += GSLAB) {
                __syncthreads();
                for (int idx = t; idx < JH * GSLAB; idx += 256) {
                    const int jj = idx / GSLAB, c = c0 + idx % GSLAB;
                    const int64_t g = p.j0 + jb + JH * h + jj;
                    const int64_t j = p.point_major ? g / p.pc : g % p.M;
                    zs[jj][idx % GSLAB] = (c < p.d && g < rows_total) ? p.Z[j * p.d + c] * p.s : 0.0;
                }
                __syncthreads();
                if (live) {
                    double x0[GSLAB], x1[GSLAB];
#pragma unroll
                    for (int c = 0; c < GSLAB; ++c) {
                        x0[c] = ok0 ? p.Xs[(int64_t)i0 * p.dp + c0 + c] : 0.0;
                        x1[c] = ok1 ? p.Xs[(int64_t)i1 * p.dp + c0 + c] : 0.0;
                    }
#pragma unroll
                    for (int jj = 0; jj < JH; ++jj) {
                        asm volatile("" ::: "memory");
                        const int qc = zq[JH * h + jj] - 1 - c0;      // the row's derivative coordinate, relative to this slab
                        const int q0 = qp0 - c0, q1 = qp1 - c0;
                        double a = u0[jj], b = u1[jj];
#pragma unroll
                        for (int c = 0; c < GSLAB; ++c) {
                            const double z = zs[jj][c];
                            const double e0 = x0[c] - z, e1 = x1[c] - z;
                            a = fma(e0, e0, a);
                            b = fma(e1, e1, b);
                            if (c == qc) { ec0[jj] = e0; ec1[jj] = e1; }
                            if (c == q0) ep0[jj] = e0;
                            if (c == q1) ep1[jj] = e1;
                        }
                        u0[jj] = a; u1[jj] = b;
                    }
                }
            }
            if (live) {
#pragma unroll
                for (int jj = 0; jj < JH; ++jj) {
                    const int q = zq[JH * h + jj];
                    const int qc = q - 1;
                    double f0, g0, h0, f1, g1, h1;
                    if constexpr (DLOGELL) {
                        phi_derivs_dlogell<FAM>(u0[jj], f0, g0, h0);
                        phi_derivs_dlogell<FAM>(u1[jj], f1, g1, h1);
                    } else {
                        phi_derivs<FAM>(u0[jj], f0, g0, h0);
                        phi_derivs<FAM>(u1[jj], f1, g1, h1);
                    }
                    double v0, v1;
                    if (qc < 0) {
                        v0 = qp0 < 0 ? f0 : 2.0 * il * g0 * ep0[jj];
                        v1 = qp1 < 0 ? f1 : 2.0 * il * g1 * ep1[jj];
                    } else {
                        v0 = qp0 < 0 ? -2.0 * il * g0 * ec0[jj] : -il * il * fma(4.0 * h0, ec0[jj] * ep0[jj], qp0 == qc ? 2.0 * g0 : 0.0);
                        v1 = qp1 < 0 ? -2.0 * il * g1 * ec1[jj] : -il * il * fma(4.0 * h1, ec1[jj] * ep1[jj], qp1 == qc ? 2.0 * g1 : 0.0);
                    }
                    v0 = (ok0 && q >= 0) ? p.sigma_f2 * v0 : 0.0;
                    v1 = (ok1 && q >= 0) ? p.sigma_f2 * v1 : 0.0;
                    if (p.Kout) *reinterpret_cast<d2_t*>(p.Kout + (int64_t)(jb + JH * h + jj) * p.ldk + k) = d2_t{v0, v1};
                    mu[JH * h + jj] = fma(v1, a1, fma(v0, a0, mu[JH * h + jj]));
                }
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
    if (t < JT) {
        const int q = zq[t];
        p.mu[jb + t] = (q >= 0 ? p.mean_vec[q] : 0.0) + (((red[0][t] + red[1][t]) + red[2][t]) + red[3][t]);
    }
}

// launch kgen_grad_kernel<FAM, dp, DLOGELL, RES> for the run-time dp
template <int FAM, bool DLOGELL, int RES>
static hipError_t launch_grad_kgen_dp(const KgenArgs& a, hipStream_t s) {
    dim3 grid(a.Mc / JT), block(256);
    switch (a.dp) {
        case 1: hipLaunchKernelGGL((kgen_grad_kernel<FAM, 1, DLOGELL, RES>), grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL((kgen_grad_kernel<FAM, 2, DLOGELL, RES>), grid, block, 0, s, a); break;
        case 4: hipLaunchKernelGGL((kgen_grad_kernel<FAM, 4, DLOGELL, RES>), grid, block, 0, s, a); break;
        case 8: hipLaunchKernelGGL((kgen_grad_kernel<FAM, 8, DLOGELL, RES>), grid, block, 0, s, a); break;
        case 16: hipLaunchKernelGGL((kgen_grad_kernel<FAM, 16, DLOGELL, RES>), grid, block, 0, s, a); break;
        case 32: hipLaunchKernelGGL((kgen_grad_kernel<FAM, 32, DLOGELL, RES>), grid, block, 0, s, a); break;
        default:
            if (RES != 0 || a.dp <= 32 || a.dp % 32) return hipErrorInvalidValue;
            hipLaunchKernelGGL((kgen_grad_wide_kernel<FAM, DLOGELL>), grid, block, 0, s, a);
            break;
    }
    return hipGetLastError();
}

// kgen_grad_res.hip: the RES = 14 instantiations
hipError_t launch_kgen_grad_res14(const KgenArgs& a, hipStream_t s);

}  // namespace abo
