// Greedy (Kriging-believer) q-EI on a resident candidate set WITHOUT one pass over the resident K_ZX per pick
// (BASELINE config 5; SURVEY.md §8 a13 (ii)/(iii) — no reference counterpart: the reference always refits,
// src/surrogates/StandardGP.jl:79-83, and its EI is single-point, src/acquisition_functions/ExpectedImprovement.jl:40-66).
//
// A pick conditions the grid's posterior on a fantasy observation at x_j.  What that needs of the 17 GB K_ZX is ONE column of
// posterior covariances, c_j(z) = Cov_{j−1}(z, x_j).  Under the BASE model (the N training points the set is synced with)
//     Cov₀(z, x) = k(z, x) − k_zᵀ K⁻¹ k_x                                                    (1)
// and every later conditioning is a rank-1 correction of it,
//     Cov_{j−1}(z, x_j) = Cov₀(z, x_j) − Σ_{i<j} c_i(z)·c_i(x_j)/s_i,      s_i = σ²_{i−1}(x_i) + σ²_n,   σ²_j = σ²_{j−1} − c_j²/s_j.   (2)
// (1) for a BLOCK of T points at once is one product  C₀ = K_ZT − K_ZX·(K⁻¹K_XT):  K_ZX is streamed ONCE for T columns
// (2·N·M·T flop on the fp64 matrix pipe under an 8·N·M-byte stream), and (2) is O(M·j) vector work.  The block is the T best
// candidates of the current scores: the arg-max of the next picks is (measured: always, at the benchmark's configuration) one of
// them; a pick outside every block builds a new block around the current scores.
//
// Kernels of this file (the products run on gemm.hip's split-k / skinny MFMA kernels):
//   qei_kxt_kernel     K_XT[t][i] = σ_f²κ(‖x_i − p_t‖/ℓ)                      the block points against the training set
//   qei_zero_tail      rows of a [T][Np] matrix beyond the view's N (a shared factor may hold a discarded appended branch there)
//   qei_cov_kernel     C₀[t][z] = k(z, p_t) + (−K_ZX·K⁻¹k_t)[z]               finishes (1) on the product's output
//   qei_pick_kernel    c_j = C₀[slot] − Σ_i γ_i·c_i ;  σ² −= c_j²/s_j          (2), the chain vector c_j is kept
//   qei_record_kernel  {score, index, μ, σ², x[d], c_1(x) … c_n(x)} of the k best candidates — what a pick (k = 1) or a block
//                      (k = T) exchanges between shards: every quantity of (2) that belongs to ONE candidate travels with it,
//                      so all shards apply the same numbers and a sharded set repeats the single set's arithmetic bit for bit
#include "abo_kernels.h"
#include "../../include/abo_hip.h"
#include "abo_kappa.h"
#include "abo_acq_dev.h"

namespace abo {

template <int FAM>
__global__ void __launch_bounds__(256) qei_kxt_kernel(const double* __restrict__ Xs, int dp, int N, int Np, const double* __restrict__ P,
                                                       int d, int T, double s, double sigma_f2, double* __restrict__ KXT) {
    const int i = blockIdx.x * 256 + threadIdx.x, t = blockIdx.y;
    if (i >= Np) return;
    double k = 0.0;
    if (t < T && i < N) {
        const double* x = Xs + (int64_t)i * dp;
        const double* p = P + (int64_t)t * d;
        double r = 0.0;
        for (int c = 0; c < d; ++c) {                     // the arithmetic of cand_newcol_kernel with the block point as the candidate
            const double e = x[c] - p[c] * s;
            r = fma(e, e, r);
        }
        k = sigma_f2 * kappa_eval<FAM>(r);
    }
    KXT[(int64_t)t * Np + i] = k;
}

__global__ void qei_zero_tail_kernel(double* V, int64_t ld, int N, int Np, int rows) {
    const int w = Np - N;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < w * rows) V[(int64_t)(e / w) * ld + N + e % w] = 0.0;
}

// one thread per candidate, the block points (pre-scaled) in LDS: z is read once per thread and stays in L1 over the T points; the
// writes of a wave are 64 consecutive candidates of one block row
template <int FAM>
__global__ void __launch_bounds__(256) qei_cov_kernel(const double* __restrict__ Ps, int dp, const double* __restrict__ Z, int64_t M,
                                                       int64_t Mp, int d, int T, double s, double sigma_f2, double* __restrict__ C) {
    extern __shared__ double ps[];                        // [T][d]
    for (int e = threadIdx.x; e < T * d; e += 256) ps[e] = Ps[(int64_t)(e / d) * dp + e % d];
    __syncthreads();
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= M) return;
    const double* z = Z + j * d;
    for (int t = 0; t < T; ++t) {
        const double* p = ps + t * d;                     // pre-scaled, as a training row of Xs is
        double r = 0.0;
        for (int c = 0; c < d; ++c) {                     // as cand_newcol_kernel writes the column of p_t once it is a training row
            const double e = p[c] - z[c] * s;
            r = fma(e, e, r);
        }
        C[(int64_t)t * Mp + j] += sigma_f2 * kappa_eval<FAM>(r);
    }
}

__global__ void __launch_bounds__(256) qei_pick_kernel(QeiPickArgs a) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= a.M) return;
    double c = a.blk[j];
    for (int i = a.first; i < a.nchain; ++i) c = fma(-a.gam[i], a.chain[(int64_t)i * a.Mp + j], c);     // fixed order: entry first, first + 1, …
    a.out[j] = c;
    if (a.var) a.var[j] = a.var[j] - c * c / a.s;         // the expression of downdate_kernel
}

__global__ void qei_record_kernel(const double* __restrict__ tv, const int64_t* __restrict__ ti, int64_t idx_base, const double* __restrict__ Z,
                                  const double* __restrict__ mu, const double* __restrict__ var, const double* __restrict__ chain,
                                  int64_t Mp, int nchain, int d, int words, double* __restrict__ rec) {
    const int e = blockIdx.x, t = threadIdx.x;
    const int64_t gi = ti[e], li = gi - idx_base;
    const bool ok = gi >= 0;
    double* r = rec + (int64_t)e * words;
    if (t == 0) {
        r[0] = tv[e];
        r[1] = (double)gi;                                // exact below 2^53
        r[2] = ok ? mu[li] : 0.0;
        r[3] = ok ? var[li] : 0.0;
    }
    for (int c = t; c < d; c += blockDim.x) r[4 + c] = ok ? Z[li * d + c] : 0.0;
    for (int i = t; i < nchain; i += blockDim.x) r[4 + d + i] = ok ? chain[(int64_t)i * Mp + li] : 0.0;
}

// ---- the pick loop on the device (abo_kernels.h: QeiStepArgs) --------------------------------------------------------------------------
// wave-then-workgroup arg-max of (key, idx) in the total order of `before`; the winner's (mu, var) travel with it.  All threads return the
// winner (broadcast through LDS).
__device__ __forceinline__ void qei_wg_argmax(uint64_t& bk, int64_t& bi, double& bm, double& bv, uint64_t* sk, int64_t* si, double* sm,
                                              double* sv) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const uint64_t ok = (uint64_t)__shfl_xor((long long)bk, o);
        const int64_t oi = (int64_t)__shfl_xor((long long)bi, o);
        const double om = __shfl_xor(bm, o), ov = __shfl_xor(bv, o);
        if (before(ok, oi, bk, bi)) { bk = ok; bi = oi; bm = om; bv = ov; }
    }
    if (lane == 0) { sk[wave] = bk; si[wave] = bi; sm[wave] = bm; sv[wave] = bv; }
    __syncthreads();
    bk = sk[0]; bi = si[0]; bm = sm[0]; bv = sv[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (before(sk[w], si[w], bk, bi)) { bk = sk[w]; bi = si[w]; bm = sm[w]; bv = sv[w]; }
    __syncthreads();                                          // (the LDS slots are reused by the caller's next reduction)
}

constexpr uint64_t QEI_KEY_PAD = 0ull;
constexpr int64_t QEI_IDX_PAD = 0x7fffffffffffffffll;

__global__ void __launch_bounds__(256) qei_step_kernel(const QeiStepArgs a) {
    __shared__ uint64_t sk[4];
    __shared__ int64_t si[4];
    __shared__ double sm[4], sv[4];
    __shared__ double gam[QEI_MAXQ];
    __shared__ int s_slot;
    if (a.st->stop) return;                                   // an earlier launch of the batch asked for the host
    const int t = threadIdx.x;
    const int k = a.k;
    bool apply = false;
    int n = 0, slot = -1, first = 0;
    int64_t li = -1;
    double s = 1.0;
    if (k > 0) {
        // (B) the winner of launch k − 1: pick k − 1 of the batch, selected on a chain of n entries
        const QeiStepPartial* pp = a.part + (size_t)((k - 1) & 1) * QEI_STEP_MAXWG;
        uint64_t bk = QEI_KEY_PAD;
        int64_t bi = QEI_IDX_PAD;
        double bm = 0.0, bv = 0.0;
        for (int e = t; e < a.nwg_prev; e += 256) {
            const QeiStepPartial p = pp[e];
            if (before(p.key, p.idx, bk, bi)) { bk = p.key; bi = p.idx; bm = p.mu; bv = p.var; }
        }
        qei_wg_argmax(bk, bi, bm, bv, sk, si, sm, sv);
        li = bi;
        n = a.n0 + (k - 1);
        const int64_t gidx = li + a.idx_base;
        if (blockIdx.x == 0) {
            double* r = a.rec + (size_t)(k - 1) * a.wmax;
            if (t == 0) { r[0] = score_of_key(bk); r[1] = (double)gidx; r[2] = bm; r[3] = bv; }
            for (int c = t; c < a.d; c += 256) r[4 + c] = a.Z[li * a.d + c];
            for (int i = t; i < n; i += 256) r[4 + a.d + i] = a.chain[(int64_t)i * a.Mp + li];
        }
        if (k == a.q) {                                       // the tail launch: the last pick conditions nothing; the batch is rolled back
            for (int64_t z = (int64_t)blockIdx.x * 256 + t; z < a.M; z += (int64_t)gridDim.x * 256) {
                a.var[z] = a.snap_var[z];
                if (a.distinct) a.mu[z] = a.snap_mu[z];
            }
            return;
        }
        // the pick's block row: the FIRST slot holding its index (qei_find_slot's order)
        if (t == 0) s_slot = 0x7fffffff;
        __syncthreads();
        for (int e = t; e < a.nslots; e += 256)
            if (a.slot_gidx[e] == gidx) atomicMin(&s_slot, e);
        s = bv + a.noise;
        for (int i = t; i < n; i += 256) {
            const double si_ = i < a.n0 ? a.chain_s0[i] : a.st->s_batch[i - a.n0];
            gam[i] = a.chain[(int64_t)i * a.Mp + li] / si_;
        }
        __syncthreads();
        slot = s_slot == 0x7fffffff ? -1 : s_slot;
        if (slot < 0 || !(s > 0.0)) {
            if (blockIdx.x == 0 && t == 0) { a.st->stop_at = k - 1; a.st->stop = slot < 0 ? 1 : 2; }
            return;
        }
        if (blockIdx.x == 0 && t == 0) a.st->s_batch[k - 1] = s;
        first = a.blk_base[slot / a.T16];
        apply = true;
    }
    // (B, continued) condition this workgroup's candidates on the pick, then (C) score them: EI and the partial arg-max
    const double* blk = a.blk + (int64_t)(slot < 0 ? 0 : slot) * a.Mp;
    double* out = a.chain + (int64_t)n * a.Mp;
    uint64_t bk = QEI_KEY_PAD;
    int64_t bi = QEI_IDX_PAD;
    double bm = 0.0, bv = 0.0;
    for (int64_t z = (int64_t)blockIdx.x * 256 + t; z < a.M; z += (int64_t)gridDim.x * 256) {
        double m = a.mu[z], v = a.var[z];
        if (k == 0) { a.snap_mu[z] = m; a.snap_var[z] = v; }  // (a resumed batch never re-runs launch 0)
        if (apply) {
            double c = blk[z];
            for (int i = first; i < n; ++i) c = fma(-gam[i], a.chain[(int64_t)i * a.Mp + z], c);     // fixed order: entry first, first + 1, …
            out[z] = c;
            v = v - c * c / s;                                // the expression of qei_pick_kernel / downdate_kernel
            if (a.distinct && z == li) { m = HUGE_VAL; v = 0.0; a.mu[z] = m; }     // abo_cand_exclude: μ = +Inf, σ² = 0
            a.var[z] = v;
        }
        const uint64_t key = score_key(acq_score(ABO_ACQ_EI, m, v, a.xi, a.best_y));
        if (before(key, z, bk, bi)) { bk = key; bi = z; bm = m; bv = v; }
    }
    qei_wg_argmax(bk, bi, bm, bv, sk, si, sm, sv);
    if (t == 0) {
        QeiStepPartial p;
        p.key = bk; p.idx = bi; p.mu = bm; p.var = bv;
        a.part[(size_t)(k & 1) * QEI_STEP_MAXWG + blockIdx.x] = p;
    }
}

hipError_t launch_qei_step(const QeiStepArgs& a, int nwg, hipStream_t st) {
    if (nwg < 1 || nwg > QEI_STEP_MAXWG || a.M < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(qei_step_kernel, dim3((unsigned)nwg), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_qei_kxt(const double* Xs, int dp, int N, int Np, const double* P, int d, int T, int rows, int family, double s,
                          double sigma_f2, double* KXT, hipStream_t st) {
    if (rows <= 0) return hipSuccess;
    dim3 grid((unsigned)((Np + 255) / 256), (unsigned)rows), block(256);
    switch (family) {
        case ABO_KERNEL_SE: hipLaunchKernelGGL((qei_kxt_kernel<ABO_KERNEL_SE>), grid, block, 0, st, Xs, dp, N, Np, P, d, T, s, sigma_f2, KXT); break;
        case ABO_KERNEL_MATERN52: hipLaunchKernelGGL((qei_kxt_kernel<ABO_KERNEL_MATERN52>), grid, block, 0, st, Xs, dp, N, Np, P, d, T, s, sigma_f2, KXT); break;
        case ABO_KERNEL_MATERN72: hipLaunchKernelGGL((qei_kxt_kernel<ABO_KERNEL_MATERN72>), grid, block, 0, st, Xs, dp, N, Np, P, d, T, s, sigma_f2, KXT); break;
        case ABO_KERNEL_MATERN32: hipLaunchKernelGGL((qei_kxt_kernel<ABO_KERNEL_MATERN32>), grid, block, 0, st, Xs, dp, N, Np, P, d, T, s, sigma_f2, KXT); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_qei_zero_tail(double* V, int64_t ld, int N, int Np, int rows, hipStream_t st) {
    if (Np <= N || rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(qei_zero_tail_kernel, dim3((unsigned)(((Np - N) * rows + 255) / 256)), dim3(256), 0, st, V, ld, N, Np, rows);
    return hipGetLastError();
}

hipError_t launch_qei_cov(const double* Ps, int dp, const double* Z, int64_t M, int64_t Mp, int d, int T, int family, double s,
                          double sigma_f2, double* C, hipStream_t st) {
    if (M <= 0 || T <= 0) return hipSuccess;
    dim3 grid((unsigned)((M + 255) / 256)), block(256);
    const size_t lds = sizeof(double) * (size_t)T * d;
    if (lds > 65536) return hipErrorInvalidValue;         // (T ≤ 64 points of d ≤ 128 coordinates)
    switch (family) {
        case ABO_KERNEL_SE: hipLaunchKernelGGL((qei_cov_kernel<ABO_KERNEL_SE>), grid, block, lds, st, Ps, dp, Z, M, Mp, d, T, s, sigma_f2, C); break;
        case ABO_KERNEL_MATERN52: hipLaunchKernelGGL((qei_cov_kernel<ABO_KERNEL_MATERN52>), grid, block, lds, st, Ps, dp, Z, M, Mp, d, T, s, sigma_f2, C); break;
        case ABO_KERNEL_MATERN72: hipLaunchKernelGGL((qei_cov_kernel<ABO_KERNEL_MATERN72>), grid, block, lds, st, Ps, dp, Z, M, Mp, d, T, s, sigma_f2, C); break;
        case ABO_KERNEL_MATERN32: hipLaunchKernelGGL((qei_cov_kernel<ABO_KERNEL_MATERN32>), grid, block, lds, st, Ps, dp, Z, M, Mp, d, T, s, sigma_f2, C); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_qei_pick(const QeiPickArgs& a, hipStream_t st) {
    if (a.M <= 0) return hipSuccess;
    hipLaunchKernelGGL(qei_pick_kernel, dim3((unsigned)((a.M + 255) / 256)), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_qei_record(const double* tv, const int64_t* ti, int k, int64_t idx_base, const double* Z, const double* mu,
                             const double* var, const double* chain, int64_t Mp, int nchain, int d, double* rec, hipStream_t st) {
    if (k <= 0) return hipSuccess;
    hipLaunchKernelGGL(qei_record_kernel, dim3((unsigned)k), dim3(64), 0, st, tv, ti, idx_base, Z, mu, var, chain, Mp, nchain, d,
                       4 + d + nchain, rec);
    return hipGetLastError();
}

}  // namespace abo
