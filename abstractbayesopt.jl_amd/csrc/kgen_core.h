// The kernel-matrix generator's core (kgen_kernel / kgen_rows), shared by kgen.hip (plain fp64 output) and kgen_res.hip (the same
// pass also writing the int8-residue engine's planes) — two translation units so that the per-family / per-dimension instantiations
// compile in parallel.  See kgen.hip for the design notes.
#pragma once
#include "abo_kernels.h"
#include "abo_kappa.h"
#include "abo_oz_dev.h"
#include "../../include/abo_hip.h"

namespace abo {

typedef double d2_t __attribute__((ext_vector_type(2)));
constexpr int JT = 16;        // candidates per workgroup
constexpr int KSTEP = 512;    // k per sweep step (256 threads × 2)

// the 16 candidate rows of a workgroup against this lane's two training points
// RES: 0 = no residue output; n > 0 = residue planes for exactly n moduli, unrolled over the compile-time tables of abo_oz_dev.h
// (instantiated for the default plan, n = 14; other moduli counts go through the separate quantiser of ozaki.hip)
template <int FAM, int DP, bool FULL, int RES>
__device__ __forceinline__ void kgen_rows(const KgenArgs& p, const double (*zs)[DP], const double (&x0)[DP], const double (&x1)[DP],
                                          double s0, double s1, double a0, double a1, int jb, int k, double (&mu)[JT], int* lds_bad) {
    const double rsc = RES != 0 ? __builtin_ldexp(1.0, p.res_sK) : 0.0;
#pragma unroll
    for (int jj = 0; jj < JT; ++jj) {
        // rows past the last candidate are written as zeros (K_XX relies on it for its identity padding; for K_XZ
        // they are padding candidates nobody reads); wave-uniform
        const bool okj = FULL || (p.j0 + jb + jj) < p.M;
        // keep the candidate coordinates in LDS: without this hipcc hoists all JT·DP broadcast reads out of the
        // k sweep and pins them in (up to 512) registers
        asm volatile("" ::: "memory");
        double v0 = 0.0, v1 = 0.0;
        if (okj) {
            double r0 = 0.0, r1 = 0.0;
#pragma unroll
            for (int c = 0; c < DP; ++c) {
                const double z = zs[jj][c];
                const double e0 = x0[c] - z, e1 = x1[c] - z;
                r0 = fma(e0, e0, r0);
                r1 = fma(e1, e1, r1);
            }
            v0 = s0 * kappa_eval<FAM>(r0);
            v1 = s1 * kappa_eval<FAM>(r1);
        }
        if (p.Kout) *reinterpret_cast<d2_t*>(p.Kout + (int64_t)(jb + jj) * p.ldk + k) = d2_t{v0, v1};
        if constexpr (RES != 0) {
            // the int8-residue engine's image of the pair: two bytes per modulus (a wave writes one 128-byte line per row and plane)
            if (!(__builtin_fabs(v0) < 1.0e300) || !(__builtin_fabs(v1) < 1.0e300)) lds_bad[jj] = 1;
            // plane offset = (uniform) row part + (per-lane, row-independent) k part: the stores take a scalar base and a 32-bit lane offset
            const unsigned koff = (unsigned)(((k >> 6) << 14) + (k & 63));
            const int64_t rowoff = ((int64_t)((jb + jj) >> 8) * (p.res_ld >> 6)) * 16384 + ((jb + jj) & 255) * 64;
            oz_residue_pair<RES>(__builtin_rint(v0 * rsc), __builtin_rint(v1 * rsc), [&](int l, unsigned two) {
                int8_t* plane = p.res + (int64_t)l * p.res_plane + rowoff;        // uniform
                *reinterpret_cast<unsigned short*>(plane + koff) = (unsigned short)two;
            });
        }
        mu[jj] = fma(v1, a1, fma(v0, a0, mu[jj]));
    }
}

template <int FAM, int DP, int RES>
__global__ void __launch_bounds__(256) kgen_kernel(KgenArgs p) {
    __shared__ double zs[JT][DP];
    __shared__ double red[4][JT];
    __shared__ int bad[JT];          // RES: a candidate row with a non-finite kernel value (any lane sets it; written out once per row below —
                                     // every res_bad entry of the chunk is written by its workgroup: no memset in front of the launch)
    const int t = threadIdx.x;
    const int jb = blockIdx.x * JT;
    if (t < JT) bad[t] = 0;
    for (int idx = t; idx < JT * DP; idx += 256) {
        const int jj = idx / DP, c = idx % DP;
        const int64_t gj = p.j0 + jb + jj;
        zs[jj][c] = (c < p.d && gj < p.M) ? p.Z[gj * p.d + c] * p.s : 0.0;
    }
    __syncthreads();

    double mu[JT];
#pragma unroll
    for (int jj = 0; jj < JT; ++jj) mu[jj] = 0.0;

    for (int k0 = 0; k0 < p.Np; k0 += KSTEP) {
        const int k = k0 + 2 * t;
        if (k < p.Np) {
            double x0[DP], x1[DP];
            const double* xp = p.Xs + (int64_t)k * DP;
            if constexpr (DP >= 2) {
#pragma unroll
                for (int c = 0; c < DP; c += 2) {
                    const d2_t v0 = *reinterpret_cast<const d2_t*>(xp + c);
                    const d2_t v1 = *reinterpret_cast<const d2_t*>(xp + DP + c);
                    x0[c] = v0[0]; x0[c + 1] = v0[1];
                    x1[c] = v1[0]; x1[c + 1] = v1[1];
                }
            } else {
                x0[0] = xp[0]; x1[0] = xp[1];
            }
            // masks as multipliers (κ is finite everywhere): a select around kappa_eval is compiled into a
            // branch per candidate row, which cuts the sweep into 16 basic blocks of two dependent chains each
            const double s0 = k < p.N ? p.sigma_f2 : 0.0, s1 = (k + 1) < p.N ? p.sigma_f2 : 0.0;
            double a0 = 0.0, a1 = 0.0;
            if (p.alpha) { a0 = p.alpha[k]; a1 = p.alpha[k + 1]; }
            // a workgroup whose 16 rows are all real candidates runs the branch-free body; an edge workgroup (the one-row
            // launch of a bordered append above all) skips the kernel evaluations of its padding rows
            if (p.j0 + jb + JT <= p.M) kgen_rows<FAM, DP, true, RES>(p, zs, x0, x1, s0, s1, a0, a1, jb, k, mu, bad);
            else kgen_rows<FAM, DP, false, RES>(p, zs, x0, x1, s0, s1, a0, a1, jb, k, mu, bad);
        }
    }
    if constexpr (RES != 0) {
        __syncthreads();
        if (t < JT) p.res_bad[jb + t] = bad[t];
    }
    if (p.mu == nullptr) return;
    // fixed-order reduction: lanes (xor tree) → 4 waves (serial)
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


// launch kgen_kernel<FAM, dp, RES> for the run-time dp ∈ {1, 2, 4, 8, 16, 32}
template <int FAM, int RES>
static hipError_t launch_kgen_dp(const KgenArgs& a, hipStream_t s) {
    dim3 grid(a.Mc / JT), block(256);
    switch (a.dp) {
        case 1: hipLaunchKernelGGL((kgen_kernel<FAM, 1, RES>), grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL((kgen_kernel<FAM, 2, RES>), grid, block, 0, s, a); break;
        case 4: hipLaunchKernelGGL((kgen_kernel<FAM, 4, RES>), grid, block, 0, s, a); break;
        case 8: hipLaunchKernelGGL((kgen_kernel<FAM, 8, RES>), grid, block, 0, s, a); break;
        case 16: hipLaunchKernelGGL((kgen_kernel<FAM, 16, RES>), grid, block, 0, s, a); break;
        case 32: hipLaunchKernelGGL((kgen_kernel<FAM, 32, RES>), grid, block, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// kgen_res.hip: the RES = 14 instantiations
hipError_t launch_kgen_res14(const KgenArgs& a, hipStream_t s);

}  // namespace abo
