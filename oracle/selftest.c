/* Sanitizer run of the plain-C oracle (test infrastructure): KAT-1 / KAT-2 / KAT-3 of SURVEY.md §8(c) — the closed
 * forms the reference's own tests assert (test/test_surrogates.jl:59-105,:145-170; test/test_acquisition.jl:27-38) —
 * plus the not-positive-definite case (test/test_bayesian_opt.jl:759-779) and a ragged random case.
 * Built by `make -C oracle asan-check` with -fsanitize=address,undefined; GPU sanitizers do not exist on this pool. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int64_t oracle_fit(int family, double ell, double sf2, double noise, double mean_c, const double *X, int64_t N, int d,
                   const double *y, double *L, double *alpha);
void oracle_predict(int family, double ell, double sf2, double mean_c, const double *X, int64_t N, int d, const double *L,
                    const double *alpha, const double *Z, int64_t M, double *mu, double *var);
double oracle_nlml(const double *L, const double *alpha, const double *y, double mean_c, int64_t N);
void oracle_acq(int kind, const double *mu, const double *var, int64_t M, double p0, double best_y, double *score);

static int fails = 0;
static void near(const char *what, double got, double want, double tol) {
    if (!(fabs(got - want) <= tol)) { printf("FAIL %s: got %.17g want %.17g\n", what, got, want); ++fails; }
}

int main(void) {
    const double X[3] = {0.0, 0.5, 1.0}, y1[3] = {0.0, 0.25, 1.0}, y3[3] = {2.0, 1.0, 0.5}, z[1] = {0.25};
    double L[9], a[3], mu, var, s;
    if (oracle_fit(0, 1.0, 1.0, 0.1, 0.0, X, 3, 1, y1, L, a) != 0) { printf("FAIL kat1 fit\n"); return 1; }
    oracle_predict(0, 1.0, 1.0, 0.0, X, 3, 1, L, a, z, 1, &mu, &var);
    near("kat1 mu", mu, 0.1771247751991296, 1e-14);
    near("kat1 var", var, 0.050320225208722924, 1e-14);
    near("kat2 nlml", oracle_nlml(L, a, y1, 0.0, 3), 2.6769327097262567, 1e-13);
    oracle_fit(0, 1.0, 1.0, 0.1, 0.0, X, 3, 1, y3, L, a);
    oracle_predict(0, 1.0, 1.0, 0.0, X, 3, 1, L, a, z, 1, &mu, &var);
    near("kat3 mu", mu, 1.467255970550952, 1e-14);
    oracle_acq(0, &mu, &var, 1, 0.01, 0.5, &s);
    near("kat3 ei", s, 3.1134583241154294e-07, 1e-18);
    oracle_acq(1, &mu, &var, 1, 2.0, 0.0, &s);
    near("kat3 ucb", s, -1.0186125700256665, 1e-14);
    oracle_acq(2, &mu, &var, 1, 0.01, 0.5, &s);
    near("kat3 pi", s, 6.60813867902732e-06, 1e-17);
    /* KAT-6: a 1e-12 duplicate with zero noise must fail with the LAPACK order of the bad minor */
    const double X6[6] = {-1.0, -1.0, 5.0, -5.0, -1.0 + 1e-12, -1.0 + 1e-12}, y6[3] = {1.0, 2.0, 1.0};
    near("kat6 info", (double)oracle_fit(0, 1.0, 1.0, 0.0, 0.0, X6, 3, 2, y6, L, a), 3.0, 0.0);
    /* ragged random case on the heap: N = 37, d = 5, M = 11, every family */
    const int N = 37, d = 5, M = 11;
    double *Xr = malloc(sizeof(double) * N * d), *yr = malloc(sizeof(double) * N), *Zr = malloc(sizeof(double) * M * d);
    double *Lr = malloc(sizeof(double) * N * N), *ar = malloc(sizeof(double) * N), *mr = malloc(sizeof(double) * M),
           *vr = malloc(sizeof(double) * M), *sr = malloc(sizeof(double) * M);
    uint64_t st = 12345;
    for (int i = 0; i < N * d; ++i) { st = st * 6364136223846793005ull + 1442695040888963407ull; Xr[i] = (double)(st >> 11) / 9007199254740992.0; }
    for (int i = 0; i < M * d; ++i) { st = st * 6364136223846793005ull + 1442695040888963407ull; Zr[i] = (double)(st >> 11) / 9007199254740992.0; }
    for (int i = 0; i < N; ++i) yr[i] = sin(6.0 * Xr[i * d]);
    for (int fam = 0; fam < 4; ++fam) {
        if (oracle_fit(fam, 0.7, 1.3, 1e-3, 0.1, Xr, N, d, yr, Lr, ar) != 0) { printf("FAIL random fit %d\n", fam); ++fails; continue; }
        oracle_predict(fam, 0.7, 1.3, 0.1, Xr, N, d, Lr, ar, Zr, M, mr, vr);
        for (int k = 0; k < 4; ++k) oracle_acq(k, mr, vr, M, 0.01, -1.0, sr);
        for (int j = 0; j < M; ++j)
            if (!(vr[j] > -1e-9 && vr[j] <= 1.3 + 1e-9 && isfinite(mr[j]))) { printf("FAIL random predict %d %d\n", fam, j); ++fails; }
        if (!isfinite(oracle_nlml(Lr, ar, yr, 0.1, N))) { printf("FAIL random nlml %d\n", fam); ++fails; }
    }
    oracle_predict(0, 1.0, 1.0, 0.0, Xr, 0, d, Lr, ar, Zr, M, mr, vr);      /* empty training set: the prior */
    near("prior var", vr[0], 1.0 + 1e-18, 1e-15);
    free(Xr); free(yr); free(Zr); free(Lr); free(ar); free(mr); free(vr); free(sr);
    printf(fails ? "selftest: %d failure(s)\n" : "selftest ok\n", fails);
    return fails != 0;
}
