/*
 * c_abi_harness.c — a torch-free, Python-free host of libabo_hip.so: what a `ccall` host (the Julia shim of
 * integration/julia/HipStandardGP.jl) does, written in plain C99.  Host arrays only, linked against
 * libabo_hip.so, which itself pulls in the SYSTEM ROCm runtime (/opt/rocm/lib/libamdhip64.so.7) — no PyTorch in the
 * process.  tests/test_gpu_c_abi.py writes the fixture (inputs + expected values: the reference's closed-form cases
 * from tests/golden/kat.json and a seeded problem answered by the CPU oracle), compiles this file with gcc and runs it
 * as a fresh child process; exit status 0 = every check passed.
 *
 * Test infrastructure, not product.  Usage: c_abi_harness <fixture.txt> [device]
 *
 * Fixture format: whitespace-separated tokens; a record is  <name> <count> <count numbers>  (%.17g doubles).
 */
#define _POSIX_C_SOURCE 200112L   /* setenv, clock_gettime */
#include <math.h>
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "abo_hip.h"

typedef struct { char name[48]; long n; double* v; } rec_t;
static rec_t recs[256];
static int nrecs = 0;
static int failures = 0;
static int device = 0;

static void load(const char* path) {
    FILE* f = fopen(path, "r");
    if (!f) { perror(path); exit(2); }
    while (nrecs < 256 && fscanf(f, "%47s %ld", recs[nrecs].name, &recs[nrecs].n) == 2) {
        rec_t* r = &recs[nrecs];
        r->v = (double*)malloc(sizeof(double) * (size_t)(r->n > 0 ? r->n : 1));
        for (long i = 0; i < r->n; ++i)
            if (fscanf(f, "%lf", &r->v[i]) != 1) { fprintf(stderr, "fixture: short record %s\n", r->name); exit(2); }
        ++nrecs;
    }
    fclose(f);
}

static const rec_t* get(const char* name) {
    for (int i = 0; i < nrecs; ++i)
        if (!strcmp(recs[i].name, name)) return &recs[i];
    fprintf(stderr, "fixture: record %s missing\n", name);
    exit(2);
}

static const rec_t* getf(const char* prefix, const char* field) {
    char b[96];
    snprintf(b, sizeof b, "%s.%s", prefix, field);
    return get(b);
}

static int has(const char* prefix, const char* field) {
    char b[96];
    snprintf(b, sizeof b, "%s.%s", prefix, field);
    for (int i = 0; i < nrecs; ++i)
        if (!strcmp(recs[i].name, b)) return 1;
    return 0;
}

#define CHECK(cond, ...)                                                     \
    do {                                                                     \
        if (!(cond)) {                                                       \
            ++failures;                                                      \
            fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__);             \
            fprintf(stderr, __VA_ARGS__);                                    \
            fprintf(stderr, "\n");                                           \
        }                                                                    \
    } while (0)

static void ok_or_die(int32_t st, const char* what) {
    if (st == ABO_OK) return;
    char msg[512];
    abo_last_error(msg, sizeof msg);
    fprintf(stderr, "FATAL %s: status %d: %s\n", what, st, msg);
    exit(1);
}

static double maxabs(const double* a, const double* b, long n) {
    double m = 0.0;
    for (long i = 0; i < n; ++i) {
        const double e = fabs(a[i] - b[i]);
        if (!(e <= m)) m = e;      /* NaN-propagating */
    }
    return m;
}

static abo_params params_of(const char* c) {
    abo_params p;
    memset(&p, 0, sizeof p);
    const double* h = getf(c, "hyper")->v;      /* family ell sigma_f2 noise mean_c */
    p.family = (int32_t)h[0]; p.device = device; p.ell = h[1]; p.sigma_f2 = h[2]; p.noise_var = h[3]; p.mean_c = h[4];
    return p;
}

/* the reference's closed-form cases: test/test_surrogates.jl:59-105,:145-170; test/test_acquisition.jl:20-43,:74-95;
 * test/test_bayesian_opt.jl:461-487,:512-559 */
static void run_kat(const char* c) {
    abo_params p = params_of(c);
    const rec_t *X = getf(c, "X"), *y = getf(c, "y"), *Z = getf(c, "Z"), *mu_e = getf(c, "mu"), *var_e = getf(c, "var");
    const int64_t N = y->n, M = mu_e->n;
    const int32_t d = (int32_t)(X->n / N);
    abo_gp* g = NULL;
    int64_t info = -1;
    ok_or_die(abo_create(&p, &g), "abo_create");
    ok_or_die(abo_fit(g, X->v, N, d, y->v, ABO_HOST, &info), "abo_fit");
    CHECK(info == 0, "%s: info = %lld", c, (long long)info);
    double* mu = (double*)malloc(sizeof(double) * M);
    double* var = (double*)malloc(sizeof(double) * M);
    ok_or_die(abo_predict(g, Z->v, M, d, ABO_HOST, mu, var, ABO_HOST), "abo_predict");
    CHECK(maxabs(mu, mu_e->v, M) <= 1e-12, "%s: mu off by %.3e", c, maxabs(mu, mu_e->v, M));
    CHECK(maxabs(var, var_e->v, M) <= 1e-12, "%s: var off by %.3e", c, maxabs(var, var_e->v, M));
    double nl = 0.0;
    ok_or_die(abo_nlml(g, &nl), "abo_nlml");
    CHECK(fabs(nl - getf(c, "nlml")->v[0]) <= 1e-11, "%s: nlml %.17g vs %.17g", c, nl, getf(c, "nlml")->v[0]);
    if (has(c, "ei")) {
        const double* a = getf(c, "acq")->v;    /* xi best_y beta */
        double* s = (double*)malloc(sizeof(double) * M);
        double tv[1];
        int64_t ti[1];
        ok_or_die(abo_acq(g, Z->v, M, d, ABO_HOST, ABO_ACQ_EI, a[0], a[1], 0, s, 1, tv, ti, ABO_HOST), "abo_acq EI");
        for (int64_t j = 0; j < M; ++j)
            CHECK(fabs(s[j] - getf(c, "ei")->v[j]) <= 1e-8 * fabs(getf(c, "ei")->v[j]) + 1e-15, "%s: EI[%lld] %.17g", c, (long long)j, s[j]);
        ok_or_die(abo_acq(g, Z->v, M, d, ABO_HOST, ABO_ACQ_UCB, a[2], 0.0, 0, s, 0, NULL, NULL, ABO_HOST), "abo_acq UCB");
        CHECK(maxabs(s, getf(c, "ucb")->v, M) <= 1e-12, "%s: UCB off by %.3e", c, maxabs(s, getf(c, "ucb")->v, M));
        ok_or_die(abo_acq(g, Z->v, M, d, ABO_HOST, ABO_ACQ_PI, a[0], a[1], 0, s, 0, NULL, NULL, ABO_HOST), "abo_acq PI");
        for (int64_t j = 0; j < M; ++j)
            CHECK(fabs(s[j] - getf(c, "pi")->v[j]) <= 1e-8 * fabs(getf(c, "pi")->v[j]) + 1e-15, "%s: PI[%lld] %.17g", c, (long long)j, s[j]);
        free(s);
    }
    /* DimensionMismatch is a status, never a crash (test/test_bayesian_opt.jl:788-817) */
    CHECK(abo_predict(g, Z->v, 1, d + 1, ABO_HOST, mu, NULL, ABO_HOST) == ABO_EDIM, "%s: wrong-dimension predict not refused", c);
    free(mu); free(var);
    ok_or_die(abo_destroy(g), "abo_destroy");
    printf("ok %s (N=%lld d=%d M=%lld)\n", c, (long long)N, d, (long long)M);
}

/* test/test_bayesian_opt.jl:749-786: a duplicated point with zero noise must fail as PosDefException(3) */
static void run_kat6(void) {
    const char* c = "kat6";
    abo_params p = params_of(c);
    const rec_t *X = getf(c, "X"), *y = getf(c, "y");
    abo_gp* g = NULL;
    int64_t info = 0;
    ok_or_die(abo_create(&p, &g), "abo_create");
    const int32_t st = abo_fit(g, X->v, y->n, (int32_t)(X->n / y->n), y->v, ABO_HOST, &info);
    char msg[512];
    abo_last_error(msg, sizeof msg);
    CHECK(st == ABO_ENOTPD, "kat6: status %d instead of ABO_ENOTPD", st);
    CHECK(info == 3, "kat6: info = %lld instead of 3", (long long)info);
    CHECK(strstr(msg, "PosDefException") != NULL, "kat6: message '%s'", msg);
    double m1;
    CHECK(abo_predict(g, X->v, 1, 2, ABO_HOST, &m1, NULL, ABO_HOST) == ABO_EINVAL, "kat6: failed handle still predicts");
    ok_or_die(abo_destroy(g), "abo_destroy");
    printf("ok kat6 (ABO_ENOTPD, info=3)\n");
}

/* abo_nlml_grad against central differences of abo_nlml in (log ell, log sigma_f2): the arithmetic behind the Julia shim's
 * Dual-typed nlml / nlml_ls (integration/julia/HipStandardGP.jl), which is what lets the STOCK optimize_hyperparameters
 * (src/bayesian_opt.jl:253-285, autodiff = :forward) run unmodified.  Each evaluation is a refit, as in the reference. */
static double nlml_at(const char* c, double log_ell, double log_sf2) {
    abo_params p = params_of(c);
    const rec_t *X = getf(c, "X"), *y = getf(c, "y");
    p.ell = exp(log_ell); p.sigma_f2 = exp(log_sf2);
    abo_gp* g = NULL;
    int64_t info = 0;
    double v = 0.0;
    ok_or_die(abo_create(&p, &g), "abo_create (nlml_at)");
    ok_or_die(abo_fit(g, X->v, y->n, (int32_t)(X->n / y->n), y->v, ABO_HOST, &info), "abo_fit (nlml_at)");
    ok_or_die(abo_nlml(g, &v), "abo_nlml (nlml_at)");
    ok_or_die(abo_destroy(g), "abo_destroy (nlml_at)");
    return v;
}

static void run_nlml_grad(const char* c, double log_ell, double log_sf2) {
    abo_params p = params_of(c);
    const rec_t *X = getf(c, "X"), *y = getf(c, "y");
    p.ell = exp(log_ell); p.sigma_f2 = exp(log_sf2);
    abo_gp* g = NULL;
    int64_t info = 0;
    double v = 0.0, g1 = 0.0, g2 = 0.0;
    ok_or_die(abo_create(&p, &g), "abo_create (nlml_grad)");
    ok_or_die(abo_fit(g, X->v, y->n, (int32_t)(X->n / y->n), y->v, ABO_HOST, &info), "abo_fit (nlml_grad)");
    ok_or_die(abo_nlml_grad(g, &v, &g1, &g2), "abo_nlml_grad");
    ok_or_die(abo_destroy(g), "abo_destroy (nlml_grad)");
    const double h = 1e-5;
    const double f1 = (nlml_at(c, log_ell + h, log_sf2) - nlml_at(c, log_ell - h, log_sf2)) / (2 * h);
    const double f2 = (nlml_at(c, log_ell, log_sf2 + h) - nlml_at(c, log_ell, log_sf2 - h)) / (2 * h);
    const double scale = fabs(v) > 1.0 ? fabs(v) : 1.0;
    CHECK(fabs(v - nlml_at(c, log_ell, log_sf2)) <= 1e-12 * scale, "%s: abo_nlml_grad value differs from abo_nlml", c);
    CHECK(fabs(g1 - f1) <= 5e-6 * fabs(f1) + 1e-6 * scale, "%s: d/dlog(ell) %.12g vs central difference %.12g", c, g1, f1);
    CHECK(fabs(g2 - f2) <= 5e-6 * fabs(f2) + 1e-6 * scale, "%s: d/dlog(sigma_f2) %.12g vs central difference %.12g", c, g2, f2);
    /* the chain rule the shim applies for Dual parameters p = A t: d nlml / dt = A^T g */
    const double A[2][2] = {{0.5, -1.0}, {3.0, 0.25}};
    const double dt0 = A[0][0] * g1 + A[1][0] * g2;
    const double fd0 = (nlml_at(c, log_ell + h * A[0][0], log_sf2 + h * A[1][0]) - nlml_at(c, log_ell - h * A[0][0], log_sf2 - h * A[1][0])) / (2 * h);
    CHECK(fabs(dt0 - fd0) <= 5e-6 * fabs(fd0) + 1e-6 * scale, "%s: chain rule %.12g vs directional difference %.12g", c, dt0, fd0);
    printf("ok nlml_grad %s (nlml %.10g, grad %.8g %.8g, central differences %.8g %.8g)\n", c, v, g1, g2, f1, f2);
}

/* which shared objects the process really mapped: the HIP runtime must be the system one, nothing of PyTorch or Python */
static void report_runtime(void) {
    FILE* f = fopen("/proc/self/maps", "r");
    char line[1024], hip[512] = "";
    int foreign = 0;
    while (f && fgets(line, sizeof line, f)) {
        char* path = strchr(line, '/');
        if (!path) continue;
        path[strcspn(path, "\n")] = 0;
        if (strstr(path, "libamdhip64") && !hip[0]) snprintf(hip, sizeof hip, "%s", path);
        if (strstr(path, "torch") || strstr(path, "libpython")) foreign = 1;
    }
    if (f) fclose(f);
    printf("hip_runtime=%s\n", hip[0] ? hip : "(none)");
    CHECK(!foreign, "PyTorch / Python objects are mapped into this process");
    CHECK(hip[0] != 0, "no libamdhip64 mapped");
}

/* Leaves the process the way a host that never finalises its handles leaves it: a live model, a live multi-device group with
 * an RCCL communicator, buffers parked in the library's pool, a worker thread parked in the multi-device driver — and returns
 * from main.  Exit handlers, the library's static destructors and the HIP runtime's own teardown must get along: the child
 * has to end with status 0 (a SIGSEGV under __cxa_finalize in an experimental build). */
static int run_exit_live(void) {
    const char* c = "acq";
    abo_params p = params_of(c);
    const rec_t *X = getf(c, "X"), *y = getf(c, "y"), *Z = getf(c, "Z");
    const int64_t N = y->n;
    const int32_t d = (int32_t)(X->n / N);
    abo_gp *g = NULL, *tmp = NULL;
    int64_t info = 0;
    double mu[64], var[64], tv[8];
    int64_t ti[8];
    ok_or_die(abo_create(&p, &g), "abo_create");
    ok_or_die(abo_fit(g, X->v, N, d, y->v, ABO_HOST, &info), "abo_fit");
    ok_or_die(abo_predict(g, Z->v, 64, d, ABO_HOST, mu, var, ABO_HOST), "abo_predict");
    ok_or_die(abo_create(&p, &tmp), "abo_create (pooled)");
    ok_or_die(abo_fit(tmp, X->v, N, d, y->v, ABO_HOST, &info), "abo_fit (pooled)");
    ok_or_die(abo_destroy(tmp), "abo_destroy (its buffers, stream and events go to the pool)");
    int32_t devs[2] = {device, device};
    abo_mgpu *mg2 = NULL, *mg1 = NULL;
    ok_or_die(abo_mgpu_create(&p, 2, devs, &mg2), "abo_mgpu_create (2 shards: worker threads)");
    ok_or_die(abo_mgpu_fit(mg2, X->v, N, d, y->v, &info), "abo_mgpu_fit");
    ok_or_die(abo_mgpu_acq(mg2, Z->v, 4096, d, ABO_ACQ_EI, 0.01, 0.0, NULL, 8, tv, ti), "abo_mgpu_acq");
    setenv("ABO_MGPU_EXCHANGE", "rccl", 1);
    ok_or_die(abo_mgpu_create(&p, 1, devs, &mg1), "abo_mgpu_create (rccl)");
    ok_or_die(abo_mgpu_fit(mg1, X->v, N, d, y->v, &info), "abo_mgpu_fit (rccl)");
    ok_or_die(abo_mgpu_acq(mg1, Z->v, 4096, d, ABO_ACQ_EI, 0.01, 0.0, NULL, 8, tv, ti), "abo_mgpu_acq (rccl)");
    int32_t nd = 0, ex = -1;
    ok_or_die(abo_mgpu_info(mg1, &nd, NULL, &ex), "abo_mgpu_info");
    printf("exit_live: leaving with a live model, a 2-shard group, a %s communicator and pooled buffers\n",
           ex == ABO_XCHG_RCCL ? "live RCCL" : "host-exchange (RCCL unavailable)");
    printf("exchange=%s\n", ex == ABO_XCHG_RCCL ? "rccl" : "host");
    fflush(stdout);
    return failures ? 1 : 0;                               /* nothing destroyed */
}

/* Per-step latency of the path from a host with no interpreter in the way (what a Julia `ccall` host pays): the reference's own
 * loop size — N = 25 points, d = 1, 10 000 grid points, EI, top-100 — as abo_create + abo_fit + abo_acq + abo_destroy per step (the
 * two calls of the stock driver) and as abo_create + abo_fit_acq + abo_destroy; host arrays in, top-100 out. */
static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static int run_latency(void) {
    enum { N = 25, M = 10000, K = 100, STEPS = 300, WARM = 30 };
    static double X[N], y[N], Z[M], tv[K];
    static int64_t ti[K];
    for (int i = 0; i < N; ++i) { X[i] = (i + 0.37) / N; y[i] = sin(10.0 * X[i]); }
    for (int j = 0; j < M; ++j) Z[j] = (j + 0.5) / M;
    abo_params p;
    memset(&p, 0, sizeof p);
    p.family = ABO_KERNEL_MATERN52; p.device = device; p.ell = 0.3; p.sigma_f2 = 1.0; p.noise_var = 1e-6;
    double best = y[0];
    for (int i = 1; i < N; ++i) if (y[i] < best) best = y[i];
    for (int fused = 0; fused < 2; ++fused) {
        double t0 = 0.0;
        for (int s = 0; s < WARM + STEPS; ++s) {
            if (s == WARM) t0 = now_ms();
            abo_gp* g = NULL;
            int64_t info = 0;
            ok_or_die(abo_create(&p, &g), "abo_create");
            if (fused) {
                ok_or_die(abo_fit_acq(g, X, N, 1, y, ABO_HOST, &info, Z, M, ABO_HOST, ABO_ACQ_EI, 0.0, best, 0, NULL, K, tv, ti, ABO_HOST), "abo_fit_acq");
            } else {
                ok_or_die(abo_fit(g, X, N, 1, y, ABO_HOST, &info), "abo_fit");
                ok_or_die(abo_acq(g, Z, M, 1, ABO_HOST, ABO_ACQ_EI, 0.0, best, 0, NULL, K, tv, ti, ABO_HOST), "abo_acq");
            }
            ok_or_die(abo_destroy(g), "abo_destroy");
        }
        printf("latency %s: %.4f ms per step (N=%d, d=1, M=%d, EI, top-%d; host arrays, %d steps; ABO_PHASE_EVENTS=%s)\n",
               fused ? "abo_fit_acq" : "abo_fit + abo_acq", (now_ms() - t0) / STEPS, N, M, K, STEPS,
               getenv("ABO_PHASE_EVENTS") ? getenv("ABO_PHASE_EVENTS") : "automatic");
    }
    return failures ? 1 : 0;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s fixture.txt [device] [exit_live | latency]\n", argv[0]); return 2; }
    load(argv[1]);
    report_runtime();
    if (argc > 2) device = atoi(argv[2]);
    if (argc > 3 && !strcmp(argv[3], "exit_live")) return run_exit_live();
    if (argc > 3 && !strcmp(argv[3], "latency")) return run_latency();
    CHECK(abo_abi_version() == ABO_ABI_VERSION, "library ABI %d, header %d", abo_abi_version(), ABO_ABI_VERSION);

    run_kat("kat1"); run_kat("kat3"); run_kat("kat4"); run_kat("kat5");
    run_kat6();
    run_nlml_grad("kat1", 0.0, 0.0);                       /* KAT-2's configuration (test_surrogates.jl:151-169) */
    run_nlml_grad("kat1", log(0.4), log(2.5));
    run_nlml_grad("acq", log(0.6), log(1.3));              /* N = 300, d = 4, Matern-5/2 */

    /* --- seeded problem: abo_acq with k = 100 against the CPU oracle's selection ------------------------------- */
    const char* c = "acq";
    abo_params p = params_of(c);
    const rec_t *X = getf(c, "X"), *y = getf(c, "y"), *Z = getf(c, "Z"), *sc_e = getf(c, "scores"), *idx_e = getf(c, "top_idx");
    const double* a = getf(c, "acq")->v;         /* kind p0 best_y */
    const int64_t N = y->n, M = sc_e->n;
    const int32_t d = (int32_t)(X->n / N), K = (int32_t)idx_e->n;
    p.n_max = N + 8;
    abo_gp* g = NULL;
    int64_t info = 0;
    ok_or_die(abo_create(&p, &g), "abo_create");
    ok_or_die(abo_fit(g, X->v, N, d, y->v, ABO_HOST, &info), "abo_fit");
    double* s = (double*)malloc(sizeof(double) * M);
    double* tv = (double*)malloc(sizeof(double) * K);
    int64_t* ti = (int64_t*)malloc(sizeof(int64_t) * K);
    ok_or_die(abo_acq(g, Z->v, M, d, ABO_HOST, (int32_t)a[0], a[1], a[2], 0, s, K, tv, ti, ABO_HOST), "abo_acq");
    CHECK(maxabs(s, sc_e->v, M) <= 1e-10, "acq: scores off by %.3e", maxabs(s, sc_e->v, M));
    int same = 1;
    for (int e = 0; e < K; ++e) same = same && ti[e] == (int64_t)idx_e->v[e] && tv[e] == s[ti[e]];
    CHECK(same, "acq: top-%d differs from the oracle's sortperm(scores; rev=true)[1:%d]", K, K);
    printf("ok acq (N=%lld d=%d M=%lld top-%d, max|dscore| %.2e)\n", (long long)N, d, (long long)M, K, maxabs(s, sc_e->v, M));

    /* --- optimize_acquisition in one call (acq_utils.jl:33-73): device grid, top-k, on-device refinement, arg-max -------- */
    {
        enum { KL = 20 };
        double lower[8], upper[8], bx[8], bval = 0.0, sx[KL * 8], sv[KL], rx[KL * 8], rv[KL];
        for (int cdim = 0; cdim < d; ++cdim) { lower[cdim] = 0.0; upper[cdim] = 1.0; }
        ok_or_die(abo_optimize_acquisition(g, (int32_t)a[0], a[1], a[2], lower, upper, d, 4000, KL, 3u, NULL, bx, &bval, sx, sv, rx, rv),
                  "abo_optimize_acquisition");
        int inbox = 1, noloss = 1;
        for (int cdim = 0; cdim < d; ++cdim) inbox = inbox && bx[cdim] >= 0.0 && bx[cdim] <= 1.0;
        for (int e = 0; e < KL; ++e) noloss = noloss && rv[e] >= sv[e] - 1e-15 && (e == 0 || sv[e] <= sv[e - 1]);
        CHECK(inbox, "optimize_acquisition: result outside the box");
        CHECK(noloss, "optimize_acquisition: a refined start lost against its start, or the starts are not sorted");
        CHECK(bval >= sv[0], "optimize_acquisition: best value %.17g below the best grid score %.17g", bval, sv[0]);
        double sb = 0.0;
        ok_or_die(abo_acq(g, bx, 1, d, ABO_HOST, (int32_t)a[0], a[1], a[2], 0, &sb, 0, NULL, NULL, ABO_HOST), "abo_acq (best point)");
        CHECK(fabs(sb - bval) <= 1e-9 * fmax(1.0, fabs(bval)), "optimize_acquisition: value %.17g, score of the returned point %.17g", bval, sb);
        abo_timings tm;
        ok_or_die(abo_get_timings(g, &tm), "abo_get_timings");
        CHECK(tm.refine_starts == KL && tm.refine_evals >= KL && tm.refine_ms > 0.0, "optimize_acquisition: timings not filled");
        /* the sharded group: same grid (counter-based generator), same selection, same refinement → same bits */
        int32_t devs2[2] = {device, device};
        abo_mgpu* mg = NULL;
        double bx2[8], bval2 = 0.0, rv2[KL];
        ok_or_die(abo_mgpu_create(&p, 2, devs2, &mg), "abo_mgpu_create");
        ok_or_die(abo_mgpu_fit(mg, X->v, N, d, y->v, &info), "abo_mgpu_fit");
        ok_or_die(abo_mgpu_optimize_acquisition(mg, (int32_t)a[0], a[1], a[2], lower, upper, d, 4000, KL, 3u, NULL, bx2, &bval2, NULL, NULL, NULL, rv2),
                  "abo_mgpu_optimize_acquisition");
        CHECK(bval2 == bval && memcmp(bx, bx2, sizeof(double) * d) == 0 && memcmp(rv, rv2, sizeof rv) == 0,
              "mgpu optimize_acquisition differs from the single-device call");
        ok_or_die(abo_mgpu_destroy(mg), "abo_mgpu_destroy");
        /* ABI 5: the same call as a one-term objective is the same bits; a two-member ensemble (EnsembleAcq.jl:53-55) scores as the
         * weighted sum of its members' scores; the grid stage on one handle (abo_acq_lhs) returns the starts of the one call */
        {
            abo_acq_term one = {(int32_t)a[0], 0, a[1], a[2], 1.0};
            double bx3[8], bval3 = 0.0, rv3[KL];
            ok_or_die(abo_optimize_acquisition_terms(g, &one, 1, lower, upper, d, 4000, KL, 3u, NULL, bx3, &bval3, NULL, NULL, NULL, rv3),
                      "abo_optimize_acquisition_terms");
            CHECK(bval3 == bval && memcmp(bx, bx3, sizeof(double) * d) == 0 && memcmp(rv, rv3, sizeof rv) == 0,
                  "one-term abo_optimize_acquisition_terms differs from abo_optimize_acquisition");
            double lv[KL], lx[KL * 8];
            int64_t li[KL];
            ok_or_die(abo_acq_lhs(g, 4000, d, lower, upper, 3u, (int32_t)a[0], a[1], a[2], KL, lv, li, lx), "abo_acq_lhs");
            CHECK(memcmp(lv, sv, sizeof sv) == 0 && memcmp(lx, sx, sizeof(double) * KL * d) == 0, "abo_acq_lhs: starts differ from the one call's");
            abo_acq_term two[2] = {{ABO_ACQ_EI, 0, a[1], a[2], 0.25}, {ABO_ACQ_UCB, 0, 2.0, 0.0, 0.75}};
            double s_ei[64], s_ucb[64], s_ens[64];
            ok_or_die(abo_acq(g, Z->v, 64, d, ABO_HOST, ABO_ACQ_EI, a[1], a[2], 0, s_ei, 0, NULL, NULL, ABO_HOST), "abo_acq (EI)");
            ok_or_die(abo_acq(g, Z->v, 64, d, ABO_HOST, ABO_ACQ_UCB, 2.0, 0.0, 0, s_ucb, 0, NULL, NULL, ABO_HOST), "abo_acq (UCB)");
            ok_or_die(abo_acq_terms(g, Z->v, 64, d, ABO_HOST, two, 2, 0, s_ens, 0, NULL, NULL, ABO_HOST), "abo_acq_terms");
            double worst = 0.0;
            for (int e = 0; e < 64; ++e) worst = fmax(worst, fabs(s_ens[e] - (0.25 * s_ei[e] + 0.75 * s_ucb[e])));
            CHECK(worst <= 1e-14, "abo_acq_terms: ensemble off the weighted sum of its members by %.3e", worst);
            double bx4[8], bval4 = 0.0, sb4 = 0.0;
            ok_or_die(abo_optimize_acquisition_terms(g, two, 2, lower, upper, d, 4000, KL, 3u, NULL, bx4, &bval4, NULL, NULL, NULL, NULL),
                      "abo_optimize_acquisition_terms (ensemble)");
            ok_or_die(abo_acq_terms(g, bx4, 1, d, ABO_HOST, two, 2, 0, &sb4, 0, NULL, NULL, ABO_HOST), "abo_acq_terms (best point)");
            CHECK(fabs(sb4 - bval4) <= 1e-9 * fmax(1.0, fabs(bval4)), "ensemble optimize_acquisition: value %.17g, score there %.17g", bval4, sb4);
            abo_acq_term bad = {ABO_ACQ_GRADNORM_UCB, 0, 2.0, 0.0, 1.0};
            CHECK(abo_acq_terms(g, Z->v, 4, d, ABO_HOST, &bad, 1, 0, s_ens, 0, NULL, NULL, ABO_HOST) == ABO_EINVAL,
                  "GradientNormUCB accepted on a model without gradient outputs");
            printf("ok objectives as terms (one term = the plain call; ensemble = weighted sum to %.1e; abo_acq_lhs = the one call's starts)\n", worst);
        }
        CHECK(abo_refine(g, 9, 0.0, 0.0, lower, upper, d, sx, 1, NULL, rx, rv, NULL) == ABO_EINVAL, "abo_refine accepts an unknown acquisition");
        CHECK(abo_refine(g, 0, 0.0, 0.0, lower, upper, d + 1, sx, 1, NULL, rx, rv, NULL) == ABO_EDIM, "abo_refine accepts a wrong dimension");
        printf("ok optimize_acquisition (n_grid=4000 n_local=%d: best %.6g vs best grid score %.6g, %lld evaluations, refinement %.3f ms)\n",
               KL, bval, sv[0], (long long)tm.refine_evals, tm.refine_ms);
    }

    /* --- gradient-enhanced model (GradientGP.jl:617-668) from this host: the reference's closed-form case (test_surrogates.jl:289-348),
     * and optimize_acquisition with GradientNormUCB (gradNormUCB.jl:39-51) / EI as ONE call on it (ABI 5) ----------------------- */
    {
        abo_params pg = p;
        pg.family = ABO_KERNEL_SE; pg.ell = 1.0; pg.sigma_f2 = 1.0; pg.noise_var = 0.1; pg.mean_c = 0.0; pg.n_max = 0;
        const double meang[3] = {0.0, 0.0, 0.0};
        const double Xg[6] = {0.0, 0.0, 0.5, 0.5, 1.0, 1.0};                       /* three points, point-major */
        const double Yg[9] = {1.0, 0.5, 0.0, 0.1, 0.0, -0.1, 0.1, 0.0, -0.1};      /* by outputs: f, then d/dx1, then d/dx2 */
        abo_gp* gg = NULL;
        ok_or_die(abo_create_grad(&pg, 3, meang, &gg), "abo_create_grad");
        ok_or_die(abo_fit(gg, Xg, 3, 2, Yg, ABO_HOST, &info), "abo_fit (gradient-enhanced)");
        const double zq[2] = {0.25, 0.25};
        double mg[3], vg[3];
        ok_or_die(abo_predict_grad(gg, zq, 1, 2, ABO_HOST, mg, vg, ABO_HOST), "abo_predict_grad");
        CHECK(vg[0] >= 0.0 && vg[1] > 0.0 && vg[2] > 0.0 && mg[0] > 0.3 && mg[0] < 1.0, "gradient-enhanced posterior out of range (mu %.4f var %.4f)", mg[0], vg[0]);
        double lo2[2] = {-0.5, -0.5}, up2[2] = {1.5, 1.5}, bxg[2], bvg = 0.0, sxg[16 * 2], svg[16], scg = 0.0;
        abo_acq_term tg[2] = {{ABO_ACQ_GRADNORM_UCB, 0, 2.0, 0.0, 1.0}, {ABO_ACQ_EI, 0, 0.01, 0.0, 1.0}};
        for (int which = 0; which < 2; ++which) {
            ok_or_die(abo_optimize_acquisition_terms(gg, &tg[which], 1, lo2, up2, 2, 2000, 16, 7u, NULL, bxg, &bvg, sxg, svg, NULL, NULL),
                      "abo_optimize_acquisition_terms (gradient-enhanced)");
            ok_or_die(abo_acq_terms(gg, bxg, 1, 2, ABO_HOST, &tg[which], 1, 0, &scg, 0, NULL, NULL, ABO_HOST), "abo_acq_terms (gradient-enhanced)");
            CHECK(bxg[0] >= -0.5 && bxg[0] <= 1.5 && bxg[1] >= -0.5 && bxg[1] <= 1.5, "gradient-enhanced optimize_acquisition: result outside the box");
            CHECK(bvg >= svg[0] - 1e-12, "gradient-enhanced optimize_acquisition: best %.17g below the best grid score %.17g", bvg, svg[0]);
            CHECK(fabs(scg - bvg) <= 1e-9 * fmax(1.0, fabs(bvg)), "gradient-enhanced optimize_acquisition: value %.17g, score there %.17g", bvg, scg);
        }
        printf("ok gradient-enhanced model (closed-form case fitted; optimize_acquisition with GradientNormUCB and EI in one call each)\n");
        ok_or_die(abo_destroy(gg), "abo_destroy (gradient-enhanced)");
    }

    /* --- the int8-residue contraction engine from this host: same scores (to fp64 rounding), same selection --------- */
    {
        abo_gp* g8 = NULL;
        ok_or_die(abo_create(&p, &g8), "abo_create (int8 engine)");
        ok_or_die(abo_set_contraction(g8, ABO_CONTRACT_INT8, 0), "abo_set_contraction");
        CHECK(abo_set_contraction(g8, 7, 0) == ABO_EINVAL && abo_set_contraction(g8, ABO_CONTRACT_INT8, 3) == ABO_EINVAL,
              "abo_set_contraction accepts an unknown engine / moduli count");
        ok_or_die(abo_fit(g8, X->v, N, d, y->v, ABO_HOST, &info), "abo_fit (int8 engine)");
        double* s8 = (double*)malloc(sizeof(double) * M);
        double* tv8 = (double*)malloc(sizeof(double) * K);
        int64_t* ti8 = (int64_t*)malloc(sizeof(int64_t) * K);
        ok_or_die(abo_acq(g8, Z->v, M, d, ABO_HOST, (int32_t)a[0], a[1], a[2], 0, s8, K, tv8, ti8, ABO_HOST), "abo_acq (int8 engine)");
        abo_timings tm;
        ok_or_die(abo_get_timings(g8, &tm), "abo_get_timings");
        CHECK(tm.contraction_engine == ABO_CONTRACT_INT8 && tm.oz_nmod == 14, "int8 engine: timings say engine %lld, %lld moduli",
              (long long)tm.contraction_engine, (long long)tm.oz_nmod);
        ok_or_die(abo_get_timings(g, &tm), "abo_get_timings");
        CHECK(tm.contraction_engine == ABO_CONTRACT_FP64, "N = %lld ran on engine %lld by default", (long long)N, (long long)tm.contraction_engine);
        CHECK(maxabs(s8, sc_e->v, M) <= 1e-10, "int8 engine: scores off the oracle by %.3e", maxabs(s8, sc_e->v, M));
        CHECK(maxabs(s8, s, M) <= 1e-12, "int8 engine: scores off the fp64 engine by %.3e", maxabs(s8, s, M));
        int same8 = 1;
        for (int e = 0; e < K; ++e) same8 = same8 && ti8[e] == (int64_t)idx_e->v[e];
        CHECK(same8, "int8 engine: top-%d differs from the oracle's", K);
        printf("ok int8-residue contraction engine (max|dscore| vs oracle %.2e, vs fp64 engine %.2e)\n", maxabs(s8, sc_e->v, M), maxabs(s8, s, M));
        free(s8); free(tv8); free(ti8);
        ok_or_die(abo_destroy(g8), "abo_destroy (int8 engine)");
    }

    /* --- copy = shared reference; append leaves the parent untouched (rollback); refit equals append ------------ */
    ok_or_die(abo_retain(g), "abo_retain");
    ok_or_die(abo_destroy(g), "abo_destroy (one of two references)");
    double* mu0 = (double*)malloc(sizeof(double) * 256);
    double* var0 = (double*)malloc(sizeof(double) * 256);
    double* mu1 = (double*)malloc(sizeof(double) * 256);
    double* var1 = (double*)malloc(sizeof(double) * 256);
    ok_or_die(abo_predict(g, Z->v, 256, d, ABO_HOST, mu0, var0, ABO_HOST), "abo_predict (retained handle)");
    const rec_t *xn = getf(c, "x_new"), *mu_a = getf(c, "mu_appended"), *var_a = getf(c, "var_appended");
    abo_gp* g2 = NULL;
    ok_or_die(abo_append(g, xn->v, d, getf(c, "y_new")->v[0], &info, &g2), "abo_append");
    int64_t n2 = 0;
    ok_or_die(abo_get_n(g2, &n2, NULL), "abo_get_n");
    CHECK(n2 == N + 1, "append: N = %lld", (long long)n2);
    ok_or_die(abo_predict(g2, Z->v, 256, d, ABO_HOST, mu1, var1, ABO_HOST), "abo_predict (appended)");
    CHECK(maxabs(mu1, mu_a->v, 256) <= 1e-9, "append: mu off the oracle's N+1 refit by %.3e", maxabs(mu1, mu_a->v, 256));
    CHECK(maxabs(var1, var_a->v, 256) <= 1e-9, "append: var off the oracle's N+1 refit by %.3e", maxabs(var1, var_a->v, 256));
    ok_or_die(abo_predict(g, Z->v, 256, d, ABO_HOST, mu1, var1, ABO_HOST), "abo_predict (parent after append)");
    CHECK(memcmp(mu0, mu1, sizeof(double) * 256) == 0 && memcmp(var0, var1, sizeof(double) * 256) == 0,
          "rollback: the parent model changed under an append");
    ok_or_die(abo_destroy(g2), "abo_destroy (appended)");
    printf("ok retain / append / rollback\n");

    /* --- greedy q-EI on a resident candidate set (ABI 6): the block form against the plain loop, then the first pick appended for
     * real — its down-date column comes from the batch's chain (abo_timings.downdate_from_chain) ------------------------------ */
    {
        enum { Q = 5 };
        abo_cand* cs = NULL;
        ok_or_die(abo_cand_create(g, Z->v, M, d, ABO_HOST, &cs), "abo_cand_create");
        double xb[Q * 64], xp[Q * 64], eb[Q], ep[Q];
        int64_t ib[Q], ip[Q];
        abo_qei_stats qs;
        CHECK(d <= 64, "q-EI section: d = %d", d);
        ok_or_die(abo_cand_qei(g, cs, Q, a[1], a[2], 0, 0, 0, xb, ib, eb, &qs), "abo_cand_qei (block form)");
        CHECK(qs.block == 32 && qs.block_builds >= 1 && qs.block_builds + qs.block_hits == Q - 1 && qs.picks == Q,
              "q-EI stats: block %d builds %d hits %d picks %d", qs.block, qs.block_builds, qs.block_hits, qs.picks);
        ok_or_die(abo_cand_qei(g, cs, Q, a[1], a[2], 0, 0, -1, xp, ip, ep, &qs), "abo_cand_qei (plain loop)");
        CHECK(qs.block == 0, "q-EI: block = -1 must run the plain loop (stats say block %d)", qs.block);
        int sameq = 1;
        double dei = 0.0;
        for (int j = 0; j < Q; ++j) {
            sameq = sameq && ib[j] == ip[j] && memcmp(xb + j * d, xp + j * d, sizeof(double) * d) == 0 && ib[j] >= 0 && ib[j] < M;
            const double den = fabs(ep[j]) > 1e-3 * fabs(ep[0]) ? fabs(ep[j]) : 1e-3 * fabs(ep[0]);
            if (fabs(eb[j] - ep[j]) / den > dei) dei = fabs(eb[j] - ep[j]) / den;
        }
        CHECK(sameq, "q-EI: the block form and the plain loop picked different candidates");
        CHECK(dei <= 1e-9, "q-EI: EI values of the two forms differ by %.3e (relative)", dei);
        abo_gp* g3 = NULL;
        ok_or_die(abo_cand_qei(g, cs, Q, a[1], a[2], 0, 0, 0, xb, ib, eb, NULL), "abo_cand_qei (block form, again)");
        ok_or_die(abo_append(g, xb, d, 0.25, &info, &g3), "abo_append (pick 1, real value)");
        ok_or_die(abo_cand_downdate(g3, cs), "abo_cand_downdate (from the chain)");
        abo_timings tq;
        ok_or_die(abo_get_timings(g3, &tq), "abo_get_timings");
        CHECK(tq.downdate_from_chain == 1 && tq.downdate_bytes == 0.0, "down-date after the batch: from_chain %lld, bytes %.0f",
              (long long)tq.downdate_from_chain, tq.downdate_bytes);
        double* muc = (double*)malloc(sizeof(double) * M);
        double* vac = (double*)malloc(sizeof(double) * M);
        double* mur = (double*)malloc(sizeof(double) * M);
        double* var_ = (double*)malloc(sizeof(double) * M);
        ok_or_die(abo_cand_get(g3, cs, muc, vac, ABO_HOST), "abo_cand_get");
        ok_or_die(abo_predict(g3, Z->v, M, d, ABO_HOST, mur, var_, ABO_HOST), "abo_predict (appended, all candidates)");
        CHECK(maxabs(muc, mur, M) <= 1e-9 && maxabs(vac, var_, M) <= 1e-9, "chain down-date off the appended model's own posterior: mu %.3e var %.3e",
              maxabs(muc, mur, M), maxabs(vac, var_, M));
        printf("ok greedy q-EI (block form = plain loop: %d picks, max rel dEI %.2e; chain down-date vs re-evaluation: mu %.2e var %.2e)\n", Q, dei,
               maxabs(muc, mur, M), maxabs(vac, var_, M));
        free(muc); free(vac); free(mur); free(var_);
        ok_or_die(abo_destroy(g3), "abo_destroy (appended)");
        ok_or_die(abo_cand_destroy(cs), "abo_cand_destroy");
    }

    /* --- multi-device handle: two shards on this device = the single handle, bit for bit ------------------------- */
    {
        int32_t devs[2] = {device, device};
        abo_mgpu* mg = NULL;
        ok_or_die(abo_mgpu_create(&p, 2, devs, &mg), "abo_mgpu_create");
        ok_or_die(abo_mgpu_fit(mg, X->v, N, d, y->v, &info), "abo_mgpu_fit");
        double* s2 = (double*)malloc(sizeof(double) * M);
        double* tv2 = (double*)malloc(sizeof(double) * K);
        int64_t* ti2 = (int64_t*)malloc(sizeof(int64_t) * K);
        ok_or_die(abo_mgpu_acq(mg, Z->v, M, d, (int32_t)a[0], a[1], a[2], s2, K, tv2, ti2), "abo_mgpu_acq");
        CHECK(memcmp(s, s2, sizeof(double) * M) == 0, "mgpu: sharded scores differ from the single-device ones");
        CHECK(memcmp(tv, tv2, sizeof(double) * K) == 0 && memcmp(ti, ti2, sizeof(int64_t) * K) == 0,
              "mgpu: merged top-%d differs from the single-device selection", K);
        int32_t nd = 0, ex = -1;
        ok_or_die(abo_mgpu_info(mg, &nd, NULL, &ex), "abo_mgpu_info");
        CHECK(nd == 2 && ex == ABO_XCHG_HOST, "mgpu: ndev %d exchange %d (two shards on one device exchange through the host)", nd, ex);
        ok_or_die(abo_mgpu_destroy(mg), "abo_mgpu_destroy");
        /* one shard with the RCCL transport forced: ncclCommInitAll + ncclAllGather at world size 1 */
        setenv("ABO_MGPU_EXCHANGE", "rccl", 1);
        ok_or_die(abo_mgpu_create(&p, 1, devs, &mg), "abo_mgpu_create (rccl)");
        ok_or_die(abo_mgpu_fit(mg, X->v, N, d, y->v, &info), "abo_mgpu_fit (rccl)");
        ok_or_die(abo_mgpu_acq(mg, Z->v, M, d, (int32_t)a[0], a[1], a[2], NULL, K, tv2, ti2), "abo_mgpu_acq (rccl)");
        CHECK(memcmp(tv, tv2, sizeof(double) * K) == 0 && memcmp(ti, ti2, sizeof(int64_t) * K) == 0,
              "mgpu/rccl: top-%d differs from the single-device selection", K);
        ok_or_die(abo_mgpu_info(mg, &nd, NULL, &ex), "abo_mgpu_info");
        char why[512];
        abo_last_error(why, sizeof why);
        printf("ok mgpu (2 shards via host; 1 shard via %s%s%s)\n", ex == ABO_XCHG_RCCL ? "RCCL" : "host [RCCL unavailable: ",
               ex == ABO_XCHG_RCCL ? "" : why, ex == ABO_XCHG_RCCL ? "" : "]");
        printf("exchange=%s\n", ex == ABO_XCHG_RCCL ? "rccl" : "host");
        ok_or_die(abo_mgpu_destroy(mg), "abo_mgpu_destroy");
        unsetenv("ABO_MGPU_EXCHANGE");
        free(s2); free(tv2); free(ti2);
    }

    ok_or_die(abo_destroy(g), "abo_destroy (last reference)");
    ok_or_die(abo_pool_trim(device), "abo_pool_trim");
    if (failures) { fprintf(stderr, "%d check(s) failed\n", failures); return 1; }
    printf("all checks passed\n");
    return 0;
}
