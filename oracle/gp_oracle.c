/*
 * Plain-C CPU oracle for the StandardGP update / posterior / acquisition path (IEEE fp64).
 *
 * TEST INFRASTRUCTURE ONLY: an independent, loop-level restatement used to cross-check
 * oracle/gp_oracle.py and the HIP library.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load the shared object built from this file; the product
 * (abstractbayesopt.jl_amd/) never links or calls it.
 *
 * Parity status: pinned by the reference tests' closed-form identities (SURVEY.md §8(c),
 * KAT-1..6) via tests/golden/*.json.  All file:line citations are relative to /root/reference.
 *
 * Build: `make -C oracle` -> oracle/_build/libgp_oracle.so
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { K_SE = 0, K_MATERN52 = 1, K_MATERN72 = 2, K_MATERN32 = 3 };
enum { ACQ_EI = 0, ACQ_UCB = 1, ACQ_PI = 2, ACQ_MEAN = 3 };

/* kappa on the squared scaled distance.  SE: [upstream SqExponentialKernel] exp(-d2/2);
 * Matern-5/2: src/surrogates/GradientGP.jl:94-101 (closed form; the Taylor branch differs by
 * <=1e-16); Matern-7/2: GradientGP.jl:320-327; Matern-3/2: [upstream Matern32Kernel]. */
static double kappa(int family, double d2) {
    if (family == K_SE) return exp(-0.5 * d2);
    double d = sqrt(d2);
    if (family == K_MATERN52) {
        double s5 = sqrt(5.0);
        return (1.0 + s5 * d + 5.0 * d2 / 3.0) * exp(-s5 * d);
    }
    if (family == K_MATERN72) {
        double s7 = sqrt(7.0);
        return (1.0 + s7 * d + 14.0 / 5.0 * d2 + 7.0 * s7 / 15.0 * d2 * d) * exp(-s7 * d);
    }
    double s3 = sqrt(3.0);
    return (1.0 + s3 * d) * exp(-s3 * d);
}

/* sigma_f2 * kappa(||s x - s z||^2), s = 1/ell: ScaledKernel(inner ∘ ScaleTransform(1/ell), sigma_f2)
 * (src/surrogates/StandardGP.jl:41-64).  Points are point-major: x[i*d + c]. */
static double kval(int family, double s, double sf2, const double *x, const double *z, int d) {
    double d2 = 0.0;
    for (int c = 0; c < d; ++c) {
        double t = x[c] * s - z[c] * s;
        d2 += t * t;
    }
    return sf2 * kappa(family, d2);
}

/* update(model::StandardGP, xs, ys) (src/surrogates/StandardGP.jl:79-83) -> AbstractGPs.posterior
 * [upstream]: L = chol(K + noise I) (unblocked, row-oriented), delta = y - c, alpha = K^{-1} delta.
 * Returns LAPACK-style info: 0 ok, k>0 = order of the first non-positive leading minor
 * (the PosDefException the driver catches at src/bayesian_opt.jl:126-141).
 * L is N x N row-major, strictly-upper part zeroed. */
int64_t oracle_fit(int family, double ell, double sf2, double noise, double mean_c,
                   const double *X, int64_t N, int d, const double *y, double *L, double *alpha) {
    double s = 1.0 / ell;
    for (int64_t i = 0; i < N; ++i) {
        for (int64_t j = 0; j <= i; ++j)
            L[i * N + j] = kval(family, s, sf2, X + i * d, X + j * d, d) + (i == j ? noise : 0.0);
        for (int64_t j = i + 1; j < N; ++j) L[i * N + j] = 0.0;
    }
    for (int64_t j = 0; j < N; ++j) {
        double dj = L[j * N + j];
        for (int64_t k = 0; k < j; ++k) dj -= L[j * N + k] * L[j * N + k];
        if (!(dj > 0.0)) return j + 1;
        dj = sqrt(dj);
        L[j * N + j] = dj;
        for (int64_t i = j + 1; i < N; ++i) {
            double v = L[i * N + j];
            for (int64_t k = 0; k < j; ++k) v -= L[i * N + k] * L[j * N + k];
            L[i * N + j] = v / dj;
        }
    }
    /* alpha = L^{-T} (L^{-1} (y - c)) */
    for (int64_t i = 0; i < N; ++i) {
        double v = y[i] - mean_c;
        for (int64_t k = 0; k < i; ++k) v -= L[i * N + k] * alpha[k];
        alpha[i] = v / L[i * N + i];
    }
    for (int64_t i = N - 1; i >= 0; --i) {
        double v = alpha[i];
        for (int64_t k = i + 1; k < N; ++k) v -= L[k * N + i] * alpha[k];
        alpha[i] = v / L[i * N + i];
    }
    return 0;
}

/* posterior_mean / posterior_var (src/surrogates/StandardGP.jl:361-363,:377-379):
 * mu = c + k_z^T alpha;  var = sigma_f2 - ||L^{-1} k_z||^2 + 1e-18 (AbstractGPs FiniteGP default). */
void oracle_predict(int family, double ell, double sf2, double mean_c, const double *X, int64_t N,
                    int d, const double *L, const double *alpha, const double *Z, int64_t M,
                    double *mu, double *var) {
    double s = 1.0 / ell;
    double *v = (double *)malloc((size_t)(N > 0 ? N : 1) * sizeof(double));
    for (int64_t j = 0; j < M; ++j) {
        double m = 0.0, q = 0.0;
        for (int64_t i = 0; i < N; ++i) {
            double k = kval(family, s, sf2, X + i * d, Z + j * d, d);
            m += k * alpha[i];
            double t = k;
            for (int64_t p = 0; p < i; ++p) t -= L[i * N + p] * v[p];
            v[i] = t / L[i * N + i];
            q += v[i] * v[i];
        }
        mu[j] = mean_c + m;
        var[j] = sf2 - q + 1e-18;
    }
    free(v);
}

/* 0.5*(N log 2pi + logdet + delta^T alpha) (src/surrogates/StandardGP.jl:99-114). */
double oracle_nlml(const double *L, const double *alpha, const double *y, double mean_c, int64_t N) {
    double logdet = 0.0, quad = 0.0;
    for (int64_t i = 0; i < N; ++i) {
        logdet += 2.0 * log(L[i * N + i]);
        quad += (y[i] - mean_c) * alpha[i];
    }
    return 0.5 * ((double)N * log(2.0 * M_PI) + logdet + quad);
}

static double ncdf(double z) { return 0.5 * erfc(-z / sqrt(2.0)); }
static double npdf(double z) { return exp(-0.5 * z * z) / sqrt(2.0 * M_PI); }

/* EI: src/acquisition_functions/ExpectedImprovement.jl:40-66; UCB: UpperConfidenceBound.jl:38-45;
 * PI: ProbabilityImprovement.jl:38-63. */
void oracle_acq(int kind, const double *mu, const double *var, int64_t M, double p0, double best_y,
                double *score) {
    for (int64_t j = 0; j < M; ++j) {
        double m = mu[j], v = var[j];
        if (kind == ACQ_UCB) {
            score[j] = -m + p0 * sqrt(v > 0.0 ? v : 0.0);
        } else if (kind == ACQ_MEAN) {
            score[j] = -m;
        } else {
            double delta = (best_y - p0) - m;
            if (v <= 1e-12) {
                score[j] = delta > 0.0 ? delta : 0.0;
            } else {
                double sg = sqrt(v), z = delta / sg;
                score[j] = (kind == ACQ_EI) ? delta * ncdf(z) + sg * npdf(z) : ncdf(z);
            }
        }
    }
}
