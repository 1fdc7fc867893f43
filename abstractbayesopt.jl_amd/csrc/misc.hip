// Posterior epilogue (variance assembly + EI / UCB / PI) and the top-k selection.
//
// Reference arithmetic being replaced:
//   var = k(z,z) − colsum((L⁻¹K_XZ)²) + 1e-18   [upstream AbstractGPs var(::FiniteGP) with the default
//         1e-18 jitter] via src/surrogates/StandardGP.jl:377-379
//   EI   src/acquisition_functions/ExpectedImprovement.jl:40-66   (Normal cdf = erfc(−z/√2)/2)
//   UCB  src/acquisition_functions/UpperConfidenceBound.jl:38-45
//   PI   src/acquisition_functions/ProbabilityImprovement.jl:38-63 (incl. the σ²≤1e-12 → max(Δ,0) quirk)
//   top-k  sortperm(scores; rev=true)[1:k]   src/acquisition_functions/acq_utils.jl:51-52
// All of it is O(M) byte-moving work: one coalesced pass, no atomics, fixed summation order.
#include "abo_kernels.h"
#include <cstdlib>
#include "../../include/abo_hip.h"
#include "abo_acq_dev.h"

namespace abo {

__global__ void __launch_bounds__(256) finalize_kernel(FinalizeArgs p) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= p.Mc) return;
    const int64_t gj = p.j0 + j;
    if (gj >= p.M) return;
    double q = 0.0;
#pragma unroll 16
    for (int t = 0; t < p.T; ++t) q += p.partial[(int64_t)t * p.ldp + j];   // row blocks in order (loads hoisted, adds in order)
    double prior = p.sigma_f2;
    if (p.pc > 1) {                                      // which output this row is
        const int64_t o = p.point_major ? gj % p.pc : gj / p.Mpts;
        if (o > 0) prior = p.prior_grad;
    }
    const double var = prior - q + 1e-18;
    const double mu = p.mu_in[j];
    if (p.mu_out) p.mu_out[gj] = mu;
    if (p.var_out) p.var_out[gj] = var;
    if (p.score_out) p.score_out[gj] = acq_score(p.kind, mu, var, p.p0, p.best_y);
}

hipError_t launch_finalize(const FinalizeArgs& a, hipStream_t s) {
    if (a.Mc <= 0) return hipSuccess;
    hipLaunchKernelGGL(finalize_kernel, dim3((a.Mc + 255) / 256), dim3(256), 0, s, a);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) downdate_kernel(double* mu, double* var, const double* c, int64_t M, double beta,
                                                        double s2) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const double cj = c[j];
    mu[j] = fma(cj, beta, mu[j]);
    var[j] = var[j] - cj * cj / s2;
}

hipError_t launch_downdate(double* mu, double* var, const double* c, int64_t M, double beta, double s2, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    hipLaunchKernelGGL(downdate_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, mu, var, c, M, beta, s2);
    return hipGetLastError();
}

// c[j] = Σ_{k<n} Kzx[j][k]·v[k]: HBM-streaming mat-vec over the resident K_ZX.  One wave per 4 candidate rows (the
// 16-byte v loads are shared by the four rows), lanes stride k by 128; per-lane partial sums, then the fixed xor
// tree: deterministic.  Columns ≥ n hold zeros or stale finite values and meet v = 0.
__global__ void __launch_bounds__(256) cand_gemv_kernel(const double* __restrict__ Kzx, int64_t ld, const double* __restrict__ v,
                                                         int n, int64_t M, double* __restrict__ c) {
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t j0 = ((int64_t)blockIdx.x * 4 + wave) * 4;
    if (j0 >= M) return;
    const double* r0 = Kzx + j0 * ld;
    const double* r1 = r0 + (j0 + 1 < M ? ld : 0);
    const double* r2 = r0 + (j0 + 2 < M ? 2 * ld : 0);
    const double* r3 = r0 + (j0 + 3 < M ? 3 * ld : 0);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const int npair = (n + 1) >> 1;                       // ld is even and ≥ n+1: the pair load stays inside the row
    for (int q = lane; q < npair; q += 64) {
        const int k = 2 * q;
        d2_t vv = *reinterpret_cast<const d2_t*>(v + k);
        if (k + 1 >= n) vv[1] = 0.0;
        // K_ZX is read exactly once per pass and is far larger than L2 + MALL: stream it past the caches
        const d2_t x0 = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(r0 + k));
        const d2_t x1 = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(r1 + k));
        const d2_t x2 = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(r2 + k));
        const d2_t x3 = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(r3 + k));
        a0 = fma(x0[1], vv[1], fma(x0[0], vv[0], a0));
        a1 = fma(x1[1], vv[1], fma(x1[0], vv[0], a1));
        a2 = fma(x2[1], vv[1], fma(x2[0], vv[0], a2));
        a3 = fma(x3[1], vv[1], fma(x3[0], vv[0], a3));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        a0 += __shfl_xor(a0, o);
        a1 += __shfl_xor(a1, o);
        a2 += __shfl_xor(a2, o);
        a3 += __shfl_xor(a3, o);
    }
    if (lane == 0) {
        c[j0] = a0;
        if (j0 + 1 < M) c[j0 + 1] = a1;
        if (j0 + 2 < M) c[j0 + 2] = a2;
        if (j0 + 3 < M) c[j0 + 3] = a3;
    }
}

hipError_t launch_cand_gemv(const double* Kzx, int64_t ld, const double* v, int n, int64_t M, double* c, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    hipLaunchKernelGGL(cand_gemv_kernel, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, s, Kzx, ld, v, n, M, c);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) score_kernel(const double* mu, const double* var, double* score, int64_t M, int kind,
                                                     double p0, double best_y) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < M) score[j] = acq_score(kind, mu[j], var[j], p0, best_y);
}

hipError_t launch_score(const double* mu, const double* var, double* score, int64_t M, int kind, double p0, double best_y,
                        hipStream_t s) {
    if (M <= 0) return hipSuccess;
    hipLaunchKernelGGL(score_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, mu, var, score, M, kind, p0, best_y);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) score_terms_kernel(const double* mu, const double* var, double* score, int64_t M, AcqTerms t) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < M) score[j] = terms_score(t, mu[j], var[j]);
}

hipError_t launch_score_terms(const double* mu, const double* var, double* score, int64_t M, const AcqTerms& t, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    hipLaunchKernelGGL(score_terms_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, mu, var, score, M, t);
    return hipGetLastError();
}

// gradient-enhanced model: one thread per point over its mean mu[j][p] and covariance block cov[j][p][p] — function-value terms on
// (mu[0], cov[0][0]), GRADNORM_UCB terms on the gradient block
__global__ void __launch_bounds__(128) score_terms_grad_kernel(const double* mu, const double* cov, double* score, int64_t M, int p,
                                                                AcqTerms t) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const double* m = mu + j * p;
    const double* C = cov + j * p * p;
    score[j] = terms_score(t, m[0], C[0]) + terms_gradnorm(t, m, C, p);
}

hipError_t launch_score_terms_grad(const double* mu, const double* cov, double* score, int64_t M, int p, const AcqTerms& t, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    hipLaunchKernelGGL(score_terms_grad_kernel, dim3((unsigned)((M + 127) / 128)), dim3(128), 0, s, mu, cov, score, M, p, t);
    return hipGetLastError();
}

// Greedy q-EI, multi-device: one record per device and pick — {score, global index, μ(z), z[0..d)} of the device's best
// candidate — assembled on the device so that it can go straight into the RCCL all-gather (or one D2H copy).
__global__ void pick_record_kernel(const double* tv, const int64_t* ti, int64_t idx_base, const double* Z, const double* mu,
                                   int d, double* rec) {
    const int t = threadIdx.x;
    const int64_t gi = ti[0];
    const int64_t li = gi - idx_base;
    if (t == 0) {
        rec[0] = tv[0];
        rec[1] = (double)gi;                             // exact below 2^53
        rec[2] = gi >= 0 ? mu[li] : 0.0;
    }
    for (int c = t; c < d; c += blockDim.x) rec[3 + c] = gi >= 0 ? Z[li * d + c] : 0.0;
}

hipError_t launch_pick_record(const double* tv, const int64_t* ti, int64_t idx_base, const double* Z, const double* mu, int d,
                              double* rec, hipStream_t s) {
    hipLaunchKernelGGL(pick_record_kernel, dim3(1), dim3(64), 0, s, tv, ti, idx_base, Z, mu, d, rec);
    return hipGetLastError();
}

// ---- per-point posterior covariance of all outputs of a gradient-enhanced GP ----------------------------
// posterior_grad_cov(model, [x]) (src/surrogates/GradientGP.jl:966-971) and the GradientNormUCB epilogue
// (src/acquisition_functions/gradNormUCB.jl:43-51).  V rows are point-major: row j·p + q holds L⁻¹k for output q
// of point j.  One workgroup per point; wave w reduces the (q,q') pairs w, w+4, … (lanes stride the R entries).
__global__ void __launch_bounds__(256) grad_cov_kernel(GradCovArgs a) {
    extern __shared__ double gc_lds[];                         // C[p][p], m[p]: 9 KB at p = 33, 134 KB at p = 129
    double* C = gc_lds;
    double* m = gc_lds + a.p * a.p;
    const int j = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int p = a.p;
    const double* Vj = a.V + (int64_t)j * p * a.ldv;
    for (int pr = wave; pr < p * p; pr += 4) {
        const int q = pr / p, q2 = pr % p;
        if (q2 > q) continue;                                  // symmetric: lower pairs only
        const double* v1 = Vj + (int64_t)q * a.ldv;
        const double* v2 = Vj + (int64_t)q2 * a.ldv;
        double s = 0.0;
        for (int i = lane; i < a.R; i += 64) s = fma(v1[i], v2[i], s);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) {
            const double prior = (q == q2) ? ((q == 0 ? a.prior0 : a.prior_g) + 1e-18) : 0.0;
            C[q * p + q2] = prior - s;
            C[q2 * p + q] = prior - s;
        }
    }
    if (t < p) m[t] = a.mu_rows[(int64_t)j * p + t];
    __syncthreads();
    const int64_t gp = a.pt0 + j;
    if (a.cov_out) for (int e = t; e < p * p; e += 256) a.cov_out[gp * p * p + e] = C[e];
    if (a.mu_out && t < p) a.mu_out[gp * p + t] = m[t];
    if (a.score_out && t == 0) {
        // −(mᵀm + trΣ) + β·sqrt(max(4mᵀΣm + 2‖Σ‖_F², 1e-12)) on the gradient block (outputs 1..p−1)
        double ms, vs;
        gradnorm_moments(m, C, p, ms, vs);
        a.score_out[gp] = gradnorm_ucb(ms, vs, a.beta);
    }
}

hipError_t launch_grad_cov(const GradCovArgs& a, int npoints, hipStream_t s) {
    if (npoints <= 0) return hipSuccess;
    const size_t lds = sizeof(double) * ((size_t)a.p * a.p + a.p);
    if (lds > 160 * 1024 - 512) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {                                      // beyond the default cap of dynamic LDS: raised once per process
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(grad_cov_kernel),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        if (attr != hipSuccess) return attr;
    }
    hipLaunchKernelGGL(grad_cov_kernel, dim3(npoints), dim3(256), lds, s, a);
    return hipGetLastError();
}

// ---- Latin hypercube grid on the device (acq_utils.jl:44-47 without the PCIe trip) ----------------
// z[j][c] = lower_c + (π_c(j) + u_jc)/n · (upper_c − lower_c): one point per stratum in every coordinate.
// π_c is a keyed bijection of [0, n): 4-round Feistel network on ⌈log2 n⌉ bits with cycle walking
// (counter-based, so any shard of the grid can be generated independently); u from SplitMix64.
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ uint64_t feistel_perm(uint64_t x, uint64_t n, int bits, uint64_t key) {
    const int hb = (bits + 1) / 2;                       // half width (the domain is 2^(2·hb) ≥ n)
    const uint64_t hm = (1ull << hb) - 1;
    do {
        uint64_t l = x >> hb, r = x & hm;
#pragma unroll
        for (int round = 0; round < 4; ++round) {
            const uint64_t f = mix64(r ^ (key + 0x9E3779B97F4A7C15ull * (round + 1))) & hm;
            const uint64_t nl = r;
            r = l ^ f;
            l = nl;
        }
        x = (l << hb) | r;
    } while (x >= n);                                    // cycle walking: stays a bijection on [0, n)
    return x;
}

__global__ void __launch_bounds__(256) lhs_kernel(double* Z, int64_t n, int d, const double* lower, const double* upper,
                                                   uint64_t seed, int64_t j0, int64_t count, int bits) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count * d) return;
    const int64_t j = j0 + idx / d;
    const int c = (int)(idx % d);
    const uint64_t key = mix64(seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(c + 1)));
    const uint64_t stratum = feistel_perm((uint64_t)j, (uint64_t)n, bits, key);
    const double u = (double)(mix64(key ^ (uint64_t)j * 0x2545F4914F6CDD1Dull) >> 11) * (1.0 / 9007199254740992.0);
    Z[idx] = lower[c] + ((double)stratum + u) / (double)n * (upper[c] - lower[c]);
}

hipError_t launch_lhs(double* Z, int64_t n, int d, const double* lower, const double* upper, uint64_t seed, int64_t j0,
                      int64_t count, hipStream_t s) {
    if (count <= 0) return hipSuccess;
    int bits = 1;
    while ((1ull << bits) < (uint64_t)n) ++bits;
    if (bits & 1) ++bits;                                // even width: two equal Feistel halves
    const int64_t total = count * d;
    hipLaunchKernelGGL(lhs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, Z, n, d, lower, upper, seed, j0,
                       count, bits);
    return hipGetLastError();
}

// ---- top-k ------------------------------------------------------------------------------------
// Total order of Julia's stable `sortperm(scores; rev=true)`: isless-descending (NaN first, then
// +Inf … −Inf, with 0.0 before −0.0), equal scores by ascending index.  Scores are mapped to
// order-preserving u64 keys; each workgroup bitonic-sorts 2048 (key, idx) pairs in LDS and keeps its
// first KP; passes repeat until one block is left.  No atomics: bit-reproducible.
constexpr int TK_E = 2048;   // entries per workgroup
constexpr int TK_T = 1024;   // threads: one compare-exchange per thread and stage (the 66 stages are latency-bound)
constexpr uint64_t KEY_PAD = 0ull;
constexpr int64_t IDX_PAD = 0x7fffffffffffffffll;

// (score_key / before: abo_acq_dev.h — shared with the pick loop of greedy q-EI, qei.hip)

// ordering point for LDS traffic between the lanes of ONE wave (in-order LDS queue: drained counter + compiler fence)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// thr (FIRST pass only, may be null): {key, idx} of the last entry already selected by an earlier round of a k > 1024
// selection — only entries strictly AFTER it in the total order take part (the order is strict, so "after the last
// one taken" is exactly "not taken yet").
template <bool FIRST>
__global__ void __launch_bounds__(TK_T) topk_pass_kernel(const double* __restrict__ scores, const uint64_t* __restrict__ kin,
                                                        const int64_t* __restrict__ iin, int64_t n, int kp,
                                                        uint64_t* __restrict__ kout, int64_t* __restrict__ iout,
                                                        const uint64_t* __restrict__ thr) {
    __shared__ uint64_t sk[TK_E];
    __shared__ int64_t si[TK_E];
    const int t = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * TK_E;
    for (int e = t; e < TK_E; e += TK_T) {
        const int64_t g = base + e;
        uint64_t k = KEY_PAD;
        int64_t i = IDX_PAD;
        if (g < n) {
            if (FIRST) {
                k = score_key(scores[g]); i = g;
                if (thr != nullptr && !before(thr[0], (int64_t)thr[1], k, i)) { k = KEY_PAD; i = IDX_PAD; }
            } else { k = kin[g]; i = iin[g]; }
        }
        sk[e] = k;
        si[e] = i;
    }
    __syncthreads();
    for (int size = 2; size <= TK_E; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int e = t; e < TK_E / 2; e += TK_T) {
                const int lo = 2 * e - (e & (stride - 1));   // index with the `stride` bit clear
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;            // this run sorts "before-first"
                const uint64_t k0 = sk[lo], k1 = sk[hi];
                const int64_t i0 = si[lo], i1 = si[hi];
                const bool swap = up ? before(k1, i1, k0, i0) : before(k0, i0, k1, i1);
                if (swap) { sk[lo] = k1; si[lo] = i1; sk[hi] = k0; si[hi] = i0; }
            }
            // a stage with stride ≤ 64 stays inside the 128 entries its wave owns: only stages that cross waves (or
            // are followed by one that does) need the workgroup barrier — 18 of the 66
            const int next = stride > 1 ? (stride >> 1) : size;          // first stride of the next size is `size`
            if (stride > 64 || next > 64) __syncthreads();
            else wave_lds_sync();
        }
    }
    __syncthreads();
    for (int e = t; e < kp; e += TK_T) {
        kout[(int64_t)blockIdx.x * kp + e] = sk[e];
        iout[(int64_t)blockIdx.x * kp + e] = si[e];
    }
}

// entries e0 .. e0+kc−1 of the selection; thr_out (may be null) receives {key, idx} of the round's last entry
__global__ void topk_emit_kernel(const double* scores, const uint64_t* keys, const int64_t* idx, int64_t M, int kc, int e0,
                                 int64_t idx_base, double* top_val, int64_t* top_idx, uint64_t* thr_out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= kc) return;
    const int64_t i = idx[e];
    if (i == IDX_PAD || i >= M) {
        top_val[e0 + e] = __longlong_as_double(0x7ff8000000000000ll);
        top_idx[e0 + e] = -1;
    } else {
        top_val[e0 + e] = scores[i];
        top_idx[e0 + e] = i + idx_base;
    }
    if (thr_out != nullptr && e == kc - 1) { thr_out[0] = keys[e]; thr_out[1] = (uint64_t)i; }
}

__global__ void topk_fill_kernel(double* top_val, int64_t* top_idx, int lo, int hi) {
    const int e = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (e < hi) { top_val[e] = __longlong_as_double(0x7ff8000000000000ll); top_idx[e] = -1; }
}

// ---- top-k of a small batch in ONE launch ------------------------------------------------------------------------------------------
// The reference's own grid (acq_utils.jl:37: 10 000 points) is a small batch: the block-sort path above takes three launches and two
// full bitonic sorts of 2048 entries for it (50 µs — a third of a whole BO step at those sizes).  One workgroup instead, M ≤ 16384,
// k ≤ 1024:
//   1. keys (the order-preserving u64 image of the scores) into LDS, coalesced;
//   2. radix select of the k-th largest key, most significant byte first: per pass a 256-bin histogram of the entries that still
//      match the prefix (LDS integer atomics: counts are exact whatever the order), a descending scan of the bins by one wave, the
//      pass stops the search as soon as the k-th key is pinned down;
//   3. every entry above the threshold is selected; of the entries EQUAL to it the first `need` in index order (each thread owns a
//      contiguous index range, one block-wide exclusive scan of the per-thread tie counts gives the order) — Julia's stable reverse
//      sort keeps ties in index order;
//   4. the ≤ k selected (key, index) pairs are bitonic-sorted (descending key, ascending index) and written out.
// Same total order as the block-sort path (NaN first, +Inf … −Inf, 0.0 before −0.0, ties → lowest index): bit-identical results.
constexpr int TKS_T = 1024;
constexpr int TKS_MAXM = 16384;
constexpr int TKS_SPLIT_M = TKS_MAXM; // above: slices by many workgroups + one merge (measured at the reference's 10 000-point grid: two launches buy nothing there)

// MODE 0: the whole batch (above).  Larger batches, two launches instead of the block-sort chain's four (M = 65 536, k = 100: 78 → 30 µs):
// MODE 1: workgroup b selects the first min(k, its size) entries of slice b (`slice` ≤ TKS_MAXM scores from b·slice on) in the total order and
//         leaves them SORTED as (key, global index) pairs at kout / iout + b·k — dense: only the last slice can be short, and it is the last;
// MODE 2: one workgroup selects k of those n ≤ TKS_MAXM pairs (kin / iin).  Ties keep index order for free: equal keys stand in
//         ascending global index inside a slice's sorted run, and the runs stand in slice order.
template <int MODE>
__global__ void __launch_bounds__(TKS_T) topk_small_kernel(const double* __restrict__ scores, int64_t Mtot, int k, int kp, int64_t idx_base,
                                                           double* __restrict__ top_val, int64_t* __restrict__ top_idx,
                                                           const uint64_t* __restrict__ kin, const int64_t* __restrict__ iin,
                                                           uint64_t* __restrict__ kout, int64_t* __restrict__ iout, int slice) {
    __shared__ uint64_t keys[TKS_MAXM];          // 128 KB
    __shared__ uint64_t sk[1024];
    __shared__ int si[1024];
    __shared__ int hist[256];
    __shared__ int wsum[TKS_T / 64];
    __shared__ uint64_t s_prefix;
    __shared__ int s_need, s_ngt, s_done;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t base = MODE == 1 ? (int64_t)blockIdx.x * slice : 0;
    const int M = (int)((Mtot - base) < slice ? (Mtot - base) : slice);      // entries of this workgroup (MODE 2: Mtot = n pairs)
    const int kfull = k;
    if (MODE == 1 && k > M) { k = M; kp = 2; while (kp < k) kp <<= 1; }             // a short last slice gives all it has
    for (int e = t; e < M; e += TKS_T) keys[e] = MODE == 2 ? kin[e] : score_key(scores[base + e]);
    if (t == 0) { s_prefix = 0; s_need = k; s_ngt = 0; s_done = 0; }
    __syncthreads();
    // contiguous index range of this thread
    const int per = (M + TKS_T - 1) / TKS_T;
    const int e0 = t * per, e1 = min(M, e0 + per);
    // ---- 2. radix select: after the loop T = s_prefix is the k-th largest key, s_ngt = #keys > T, s_need = k − s_ngt ties to take
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 56 - 8 * pass;
        for (int b = t; b < 256; b += TKS_T) hist[b] = 0;
        __syncthreads();
        const uint64_t prefix = s_prefix;
        const uint64_t mask = pass == 0 ? 0ull : (~0ull << (shift + 8));
        for (int e = e0; e < e1; ++e) {
            const uint64_t key = keys[e];
            if ((key & mask) == prefix) atomicAdd(&hist[(int)((key >> shift) & 0xff)], 1);
        }
        __syncthreads();
        if (wave == 0) {                           // descending scan of the 256 bins: four bins per lane, lane 0 owns the top ones
            int c[4], tot = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { c[q] = hist[255 - (4 * lane + q)]; tot += c[q]; }
            int incl = tot;                        // inclusive prefix over lanes (counts of bins above and including this lane's)
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
            int above = incl - tot;                // entries in bins strictly above this lane's four
            const int need = s_need;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (above < need && above + c[q] >= need) {      // the k-th largest lies in this bin (exactly one lane and q)
                    s_prefix = prefix | ((uint64_t)(255 - (4 * lane + q)) << shift);
                    s_ngt += above;
                    s_need = need - above;
                }
                above += c[q];
            }
        }
        __syncthreads();
    }
    const uint64_t T = s_prefix;
    const int need = s_need, ngt = s_ngt;
    // ---- 3. selection: slots [0, ngt) for keys > T (any order: they are sorted below), slots [ngt, ngt + need) for the first ties
    int my_gt = 0, my_eq = 0;
    for (int e = e0; e < e1; ++e) { const uint64_t key = keys[e]; my_gt += key > T; my_eq += key == T; }
    // block-wide exclusive scans of (my_gt, my_eq): lanes, then waves
    int ig = my_gt, ie = my_eq;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int vg = __shfl_up(ig, o), ve = __shfl_up(ie, o);
        if (lane >= o) { ig += vg; ie += ve; }
    }
    if (lane == 63) { wsum[wave] = ig; hist[wave] = ie; }        // (hist is free again)
    __syncthreads();
    int og = 0, oe = 0;
    for (int w = 0; w < wave; ++w) { og += wsum[w]; oe += hist[w]; }
    int pg = og + ig - my_gt, pe = oe + ie - my_eq;               // exclusive positions of this thread's first entries
    for (int e = t; e < kp; e += TKS_T) { sk[e] = KEY_PAD; si[e] = 0x7fffffff; }
    __syncthreads();
    for (int e = e0; e < e1; ++e) {
        const uint64_t key = keys[e];
        if (key > T) { sk[pg] = key; si[pg] = e; ++pg; }
        else if (key == T) { if (pe < need) { sk[ngt + pe] = key; si[ngt + pe] = e; } ++pe; }
    }
    __syncthreads();
    // ---- 4. bitonic sort of kp ≤ 1024 (key, index) pairs: "before" = larger key, then smaller index
    for (int size = 2; size <= kp; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (t < kp / 2) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint64_t k0 = sk[lo], k1 = sk[hi];
                const int i0 = si[lo], i1 = si[hi];
                const bool b10 = (k1 > k0) || (k1 == k0 && i1 < i0);
                const bool b01 = (k0 > k1) || (k0 == k1 && i0 < i1);
                if (up ? b10 : b01) { sk[lo] = k1; si[lo] = i1; sk[hi] = k0; si[hi] = i0; }
            }
            __syncthreads();
        }
    }
    if constexpr (MODE == 1) {
        for (int e = t; e < k; e += TKS_T) {
            kout[(int64_t)blockIdx.x * kfull + e] = sk[e];
            iout[(int64_t)blockIdx.x * kfull + e] = base + si[e];
        }
        return;
    }
    for (int e = t; e < k; e += TKS_T) {
        const int i = si[e];
        if (i == 0x7fffffff) { top_val[e] = __longlong_as_double(0x7ff8000000000000ll); top_idx[e] = -1; }
        else {
            const int64_t gi = MODE == 2 ? iin[i] : (int64_t)i;
            top_val[e] = scores[gi]; top_idx[e] = gi + idx_base;
        }
    }
}

// k = 1 over a large batch (a greedy q-EI pick over a resident grid of 131 072 candidates: eight of them per config-5 step): the
// arg-max in the same total order (larger key first, ties → lowest index; NaN first as everywhere) by two plain reductions —
// 2048 entries per workgroup into the selection's workspace, then one workgroup over the partial results — instead of three
// launches of block-wide bitonic sorts (53 → 12 µs per pick).
constexpr int AM_T = 256;
__device__ __forceinline__ void argmax_wg_reduce(uint64_t& bk, int64_t& bi, uint64_t* sk, int64_t* si) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const uint64_t ok = (uint64_t)__shfl_xor((long long)bk, o);
        const int64_t oi = (int64_t)__shfl_xor((long long)bi, o);
        if (before(ok, oi, bk, bi)) { bk = ok; bi = oi; }
    }
    if (lane == 0) { sk[wave] = bk; si[wave] = bi; }
    __syncthreads();
    if (t == 0) {
        for (int w = 1; w < AM_T / 64; ++w)
            if (before(sk[w], si[w], bk, bi)) { bk = sk[w]; bi = si[w]; }
    }
}

__global__ void __launch_bounds__(AM_T) argmax_partial_kernel(const double* __restrict__ scores, int64_t n, uint64_t* __restrict__ kout,
                                                              int64_t* __restrict__ iout) {
    __shared__ uint64_t sk[AM_T / 64];
    __shared__ int64_t si[AM_T / 64];
    const int64_t base = (int64_t)blockIdx.x * TK_E;
    uint64_t bk = KEY_PAD;
    int64_t bi = IDX_PAD;
#pragma unroll
    for (int u = 0; u < TK_E / AM_T; ++u) {
        const int64_t g = base + u * AM_T + threadIdx.x;
        if (g < n) {
            const uint64_t k = score_key(scores[g]);
            if (before(k, g, bk, bi)) { bk = k; bi = g; }
        }
    }
    argmax_wg_reduce(bk, bi, sk, si);
    if (threadIdx.x == 0) { kout[blockIdx.x] = bk; iout[blockIdx.x] = bi; }
}

__global__ void __launch_bounds__(AM_T) argmax_final_kernel(const double* __restrict__ scores, const uint64_t* __restrict__ keys,
                                                            const int64_t* __restrict__ idx, int nb, int64_t M, int64_t idx_base,
                                                            double* __restrict__ top_val, int64_t* __restrict__ top_idx) {
    __shared__ uint64_t sk[AM_T / 64];
    __shared__ int64_t si[AM_T / 64];
    uint64_t bk = KEY_PAD;
    int64_t bi = IDX_PAD;
    for (int e = threadIdx.x; e < nb; e += AM_T)
        if (before(keys[e], idx[e], bk, bi)) { bk = keys[e]; bi = idx[e]; }
    argmax_wg_reduce(bk, bi, sk, si);
    if (threadIdx.x == 0) {
        if (bi == IDX_PAD || bi >= M) { top_val[0] = __longlong_as_double(0x7ff8000000000000ll); top_idx[0] = -1; }
        else { top_val[0] = scores[bi]; top_idx[0] = bi + idx_base; }
    }
}

static int pow2_at_least(int k) { int p = 1; while (p < k) p <<= 1; return p; }

constexpr int TK_KMAX = TK_E / 2;   // entries one round can select

int64_t topk_workspace_entries(int64_t M, int k) {
    const int kp = pow2_at_least(k < 1 ? 1 : (k > TK_KMAX ? TK_KMAX : k));
    const int64_t blocks = (M + TK_E - 1) / TK_E;
    return (blocks < 1 ? 1 : blocks) * (int64_t)kp + 2;   // + {key, idx} of the last entry taken (k > 1024 rounds)
}

// k ≤ 1024: one round (block-wise bitonic sort, keep the first kp per block, repeat until one block is left).
// k > 1024 (n_local is a free Int in the reference, acq_utils.jl:33-52): rounds of 1024 — each round selects, among the
// entries strictly after the last one taken so far, the next 1024 in order; ⌈k/1024⌉ passes over the scores, no host
// round trip (the threshold stays on the device).
hipError_t launch_topk(const double* scores, int64_t M, int k, int64_t idx_base, TopkWork w, double* top_val,
                       int64_t* top_idx, hipStream_t s) {
    if (k <= 0) return hipSuccess;
    uint64_t* thr = w.keys[0] + (topk_workspace_entries(M, k) - 2);
    // sortperm(...)[1:min(n_local, M)] (acq_utils.jl:52): at most M real entries; the tail of a longer request is (NaN, −1)
    const int kreal = (int64_t)k < M ? k : (int)(M < 1 ? 1 : M);
    if (kreal < k) hipLaunchKernelGGL(topk_fill_kernel, dim3((k - kreal + 255) / 256), dim3(256), 0, s, top_val, top_idx, kreal, k);
    k = kreal;
    if (M >= 1 && k <= 1024 && (M <= TKS_SPLIT_M || (M <= TKS_MAXM && (k == 1 || 4 * (int64_t)k > M)))) {
        int kp = pow2_at_least(k);
        if (kp < 2) kp = 2;
        hipLaunchKernelGGL(topk_small_kernel<0>, dim3(1), dim3(TKS_T), 0, s, scores, M, k, kp, idx_base, top_val, top_idx, nullptr, nullptr,
                           nullptr, nullptr, TKS_MAXM);
        return hipGetLastError();
    }
    if (k > 1 && k <= 1024 && M > TKS_SPLIT_M) {
        // slices, then one merge — two launches while the slices' picks fit one workgroup.  One workgroup's selection costs ≈ 3.5 µs per
        // 1024 entries: the slices are as short as keeps the merge at ≤ 8192 pairs (M = 65 536, k = 100: 32 slices of 2048, 3200 pairs)
        int slice = 2048;
        auto pairs = [&](int sl) { const int64_t G = (M + sl - 1) / sl, last = M - (G - 1) * sl; return (G - 1) * k + (last < k ? last : k); };
        while (slice < TKS_MAXM && (pairs(slice) > 8192 || slice < 2 * k)) slice *= 2;
        const int64_t n = pairs(slice);
        if (n <= TKS_MAXM && slice >= k) {
            const int64_t G = (M + slice - 1) / slice;
            int kp = pow2_at_least(k);
            if (kp < 2) kp = 2;
            hipLaunchKernelGGL(topk_small_kernel<1>, dim3((unsigned)G), dim3(TKS_T), 0, s, scores, M, k, kp, idx_base, nullptr, nullptr, nullptr,
                               nullptr, w.keys[0], w.idx[0], slice);
            hipLaunchKernelGGL(topk_small_kernel<2>, dim3(1), dim3(TKS_T), 0, s, scores, n, k, kp, idx_base, top_val, top_idx, w.keys[0],
                               w.idx[0], nullptr, nullptr, TKS_MAXM);
            return hipGetLastError();
        }
    }
    if (k == 1 && M > TKS_MAXM) {       // one workspace entry per block of TK_E scores: topk_workspace_entries(M, 1) holds them
        const int64_t nb = (M + TK_E - 1) / TK_E;
        hipLaunchKernelGGL(argmax_partial_kernel, dim3((unsigned)nb), dim3(AM_T), 0, s, scores, M, w.keys[0], w.idx[0]);
        hipLaunchKernelGGL(argmax_final_kernel, dim3(1), dim3(AM_T), 0, s, scores, w.keys[0], w.idx[0], (int)nb, M, idx_base, top_val,
                           top_idx);
        return hipGetLastError();
    }
    for (int e0 = 0; e0 < k; e0 += TK_KMAX) {
        const int kc = (k - e0) < TK_KMAX ? (k - e0) : TK_KMAX;
        const int kp = pow2_at_least(kc);
        int64_t n = M;
        int cur = 0;
        int64_t blocks = (n + TK_E - 1) / TK_E;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL((topk_pass_kernel<true>), dim3((unsigned)blocks), dim3(TK_T), 0, s, scores, nullptr, nullptr, n, kp,
                           w.keys[0], w.idx[0], e0 > 0 ? thr : nullptr);
        n = blocks * kp;
        while (blocks > 1) {
            blocks = (n + TK_E - 1) / TK_E;
            hipLaunchKernelGGL((topk_pass_kernel<false>), dim3((unsigned)blocks), dim3(TK_T), 0, s, nullptr, w.keys[cur],
                               w.idx[cur], n, kp, w.keys[cur ^ 1], w.idx[cur ^ 1], nullptr);
            cur ^= 1;
            n = blocks * kp;
        }
        hipLaunchKernelGGL(topk_emit_kernel, dim3((kc + 255) / 256), dim3(256), 0, s, scores, w.keys[cur], w.idx[cur], M, kc, e0,
                           idx_base, top_val, top_idx, e0 + kc < k ? thr : nullptr);
    }
    return hipGetLastError();
}

// ---- Monte-Carlo fill distance (src/BO_utils.jl:140-159): h = max over sample points of the distance to the nearest training point ----
// One wave per sample: lanes stride the training points (squared distance summed over the coordinates in index order), min by the xor
// tree, sqrt, then max over the samples through the order-preserving integer image of a non-negative double (an integer max: exact
// and order-independent — deterministic).
__global__ void __launch_bounds__(256) fill_distance_kernel(const double* __restrict__ X, int64_t N, int d, const double* __restrict__ S,
                                                             int64_t ns, unsigned long long* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t si = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (si >= ns) return;
    const double* sp = S + si * d;
    double best = HUGE_VAL;
    for (int64_t i = lane; i < N; i += 64) {
        const double* x = X + i * d;
        double r = 0.0;
        for (int c = 0; c < d; ++c) {
            const double e = x[c] - sp[c];
            r = fma(e, e, r);
        }
        best = fmin(best, r);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) best = fmin(best, __shfl_xor(best, o));
    if (lane == 0) atomicMax(out, (unsigned long long)__double_as_longlong(sqrt(best)));
}

hipError_t launch_fill_distance(const double* X, int64_t N, int d, const double* S, int64_t ns, double* out, hipStream_t s) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(double), s);            // +0.0: below every distance
    if (e != hipSuccess || ns <= 0 || N <= 0) return e;
    hipLaunchKernelGGL(fill_distance_kernel, dim3((unsigned)((ns + 3) / 4)), dim3(256), 0, s, X, N, d, S, ns,
                       reinterpret_cast<unsigned long long*>(out));
    return hipGetLastError();
}

__global__ void gather_points_kernel(const double* Z, const int64_t* idx, int64_t idx_base, int k, int d, double* out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= k * d) return;
    const int j = e / d, c = e % d;
    const int64_t gi = idx[j];
    out[e] = gi >= 0 ? Z[(gi - idx_base) * d + c] : 0.0;
}

hipError_t launch_gather_points(const double* Z, const int64_t* idx, int64_t idx_base, int k, int d, double* out, hipStream_t s) {
    if (k <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_points_kernel, dim3((unsigned)((k * d + 255) / 256)), dim3(256), 0, s, Z, idx, idx_base, k, d, out);
    return hipGetLastError();
}

}  // namespace abo
