// Device arithmetic of the acquisition epilogues, shared by the posterior epilogue (misc.hip) and the refinement stage (refine.hip).
//   EI   src/acquisition_functions/ExpectedImprovement.jl:40-66   (Normal cdf = erfc(−z/√2)/2)
//   UCB  src/acquisition_functions/UpperConfidenceBound.jl:38-45
//   PI   src/acquisition_functions/ProbabilityImprovement.jl:38-63 (incl. the σ² ≤ 1e-12 → max(Δ,0) quirk)
//   GradientNormUCB  src/acquisition_functions/gradNormUCB.jl:43-51
//   EnsembleAcquisition  src/acquisition_functions/EnsembleAcq.jl:53-55  (Σ wᵢ·acqᵢ on one posterior)
#pragma once
#include "abo_kernels.h"
#include "../../include/abo_hip.h"

namespace abo {

// Total order of Julia's stable `sortperm(scores; rev=true)` (acq_utils.jl:51): isless-descending (NaN first, then +Inf … −Inf, with
// 0.0 before −0.0), equal scores by ascending index.  Scores map to order-preserving u64 keys.
__device__ __forceinline__ uint64_t score_key(double s) {
    if (s != s) return 0xffffffffffffffffull;
    const uint64_t b = (uint64_t)__double_as_longlong(s);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
// the score a key came from (a NaN comes back as the canonical quiet NaN)
__device__ __forceinline__ double score_of_key(uint64_t k) {
    if (k == 0xffffffffffffffffull) return __longlong_as_double(0x7ff8000000000000ll);
    return __longlong_as_double((long long)((k >> 63) ? (k ^ 0x8000000000000000ull) : ~k));
}
// true if entry (ka, ia) must come before (kb, ib)
__device__ __forceinline__ bool before(uint64_t ka, int64_t ia, uint64_t kb, int64_t ib) {
    return (ka > kb) || (ka == kb && ia < ib);
}

__device__ __forceinline__ double norm_cdf(double z) { return 0.5 * erfc(-z * 0.70710678118654752440084436210485); }
__device__ __forceinline__ double norm_pdf(double z) { return exp(-0.5 * z * z) * 0.39894228040143267793994605993438; }

__device__ __forceinline__ double acq_score(int kind, double mu, double var, double p0, double best_y) {
    if (kind == ABO_ACQ_UCB) return -mu + p0 * sqrt(fmax(var, 0.0));
    if (kind == ABO_ACQ_MEAN) return -mu;
    const double delta = (best_y - p0) - mu;
    if (var <= 1e-12) return fmax(delta, 0.0);
    const double sg = sqrt(var);
    const double z = delta / sg;
    if (kind == ABO_ACQ_EI) return delta * norm_cdf(z) + sg * norm_pdf(z);
    return norm_cdf(z);
}

// acquisition value (the arithmetic of acq_score) and its partial derivatives with respect to μ and σ²
__device__ __forceinline__ double acq_value_and_partials(int kind, double mu, double var, double p0, double best_y, double& dmu,
                                                         double& dvar) {
    if (kind == ABO_ACQ_UCB) {
        const double sg = sqrt(fmax(var, 0.0));
        dmu = -1.0; dvar = var > 0.0 ? 0.5 * p0 / sg : 0.0;
        return -mu + p0 * sg;
    }
    if (kind == ABO_ACQ_MEAN) { dmu = -1.0; dvar = 0.0; return -mu; }
    const double delta = (best_y - p0) - mu;
    if (var <= 1e-12) { dmu = delta > 0.0 ? -1.0 : 0.0; dvar = 0.0; return fmax(delta, 0.0); }
    const double sg = sqrt(var), z = delta / sg, cdf = norm_cdf(z), pdf = norm_pdf(z);
    if (kind == ABO_ACQ_EI) { dmu = -cdf; dvar = 0.5 * pdf / sg; return delta * cdf + sg * pdf; }
    dmu = -pdf / sg; dvar = -0.5 * pdf * z / var;                 // PI = Φ(z)
    return cdf;
}

// Σ_t w_t·acq_t(μ, σ²) over the function-value terms (a GRADNORM_UCB term contributes nothing here).  One term of weight 1 is
// returned as is — the same bits as acq_score.
__device__ __forceinline__ double terms_score(const AcqTerms& t, double mu, double var) {
    if (t.n == 1 && t.w[0] == 1.0) return t.kind[0] == ACQ_GRADNORM_UCB ? 0.0 : acq_score(t.kind[0], mu, var, t.p0[0], t.best_y[0]);
    double f = 0.0;
    for (int i = 0; i < t.n; ++i)
        if (t.kind[i] != ACQ_GRADNORM_UCB) f = fma(t.w[i], acq_score(t.kind[i], mu, var, t.p0[i], t.best_y[i]), f);
    return f;
}

__device__ __forceinline__ double terms_value_and_partials(const AcqTerms& t, double mu, double var, double& dmu, double& dvar) {
    if (t.n == 1 && t.w[0] == 1.0) {
        if (t.kind[0] == ACQ_GRADNORM_UCB) { dmu = 0.0; dvar = 0.0; return 0.0; }
        return acq_value_and_partials(t.kind[0], mu, var, t.p0[0], t.best_y[0], dmu, dvar);
    }
    double f = 0.0;
    dmu = 0.0; dvar = 0.0;
    for (int i = 0; i < t.n; ++i) {
        if (t.kind[i] == ACQ_GRADNORM_UCB) continue;
        double a, b;
        const double v = acq_value_and_partials(t.kind[i], mu, var, t.p0[i], t.best_y[i], a, b);
        f = fma(t.w[i], v, f); dmu = fma(t.w[i], a, dmu); dvar = fma(t.w[i], b, dvar);
    }
    return f;
}

// −(mᵀm + trΣ) + β·sqrt(max(4mᵀΣm + 2‖Σ‖_F², 1e-12)) on the gradient block (outputs 1..p−1) of one point's mean m[p] and
// covariance C[p][p] (gradNormUCB.jl:43-51); the two moments are returned so that several β share them
__device__ __forceinline__ void gradnorm_moments(const double* m, const double* C, int p, double& mean_sq, double& var_sq) {
    double mm = 0.0, tr = 0.0, msm = 0.0, fro = 0.0;
    for (int q = 1; q < p; ++q) {
        mm = fma(m[q], m[q], mm);
        tr += C[q * p + q];
        double row = 0.0;
        for (int q2 = 1; q2 < p; ++q2) {
            row = fma(C[q * p + q2], m[q2], row);
            fro = fma(C[q * p + q2], C[q * p + q2], fro);
        }
        msm = fma(m[q], row, msm);
    }
    mean_sq = mm + tr;
    var_sq = 4.0 * msm + 2.0 * fro;
}
__device__ __forceinline__ double gradnorm_ucb(double mean_sq, double var_sq, double beta) {
    return -mean_sq + beta * sqrt(fmax(var_sq, 1e-12));
}

// Σ over the GRADNORM_UCB terms of w_t·gradNormUCB_{β_t}
__device__ __forceinline__ double terms_gradnorm(const AcqTerms& t, const double* m, const double* C, int p) {
    double ms, vs;
    gradnorm_moments(m, C, p, ms, vs);
    if (t.n == 1 && t.w[0] == 1.0) return t.kind[0] == ACQ_GRADNORM_UCB ? gradnorm_ucb(ms, vs, t.p0[0]) : 0.0;
    double f = 0.0;
    for (int i = 0; i < t.n; ++i)
        if (t.kind[i] == ACQ_GRADNORM_UCB) f = fma(t.w[i], gradnorm_ucb(ms, vs, t.p0[i]), f);
    return f;
}

}  // namespace abo
