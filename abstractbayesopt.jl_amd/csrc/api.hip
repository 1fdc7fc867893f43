// C-ABI of libabo_hip.so (declared in include/abo_hip.h): handle lifetime, the blocked fp64
// factorisation driver, the chunked posterior/acquisition driver.  Host-side orchestration only —
// all arithmetic is in the HIP kernels of gemm.hip / kgen.hip / chol.hip / misc.hip.
//
// Device-resident state of a fitted handle (Np = N rounded up to 128, identity-padded):
//   Xs  [Np][dp]   training points × 1/ell, zero padded (dp = d rounded up to 1,2,4,8,16,32)
//   K   [Np][Np]   K + noise·I, overwritten by its lower Cholesky factor L
//   W   [Np][Np]   L⁻¹ (lower, explicit zeros above)       WT = Wᵀ (upper)
//   alpha, delta [Np]
// Posterior workspace (cached across calls): K_XZ chunk [Mc][Np] candidate-major, partial
// [Np/128][Mc], mu chunk [Mc], full-length mu/var/score arrays, top-k scratch.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/abo_hip.h"
#include "abo_kernels.h"

using namespace abo;

namespace {

thread_local char g_err[512] = "";

int32_t fail(int32_t code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(e_ == hipErrorOutOfMemory ? ABO_ENOMEM : ABO_EHIP, "%s failed: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                    \
    } while (0)

// Device-memory pool.  `update` returns a NEW model every BO step (src/surrogates/StandardGP.jl:82)
// and the previous one dies right after, so without a pool every step pays hipMalloc/hipFree for
// ~4 GB of factor + workspace (and the implicit device synchronisations of hipFree).  Freed blocks
// are kept per device (up to ABO_POOL_LIMIT_MB, default 32 GiB) and handed back to the next handle.
struct Pool {
    std::mutex mu;
    std::multimap<size_t, void*> blocks;
    size_t held = 0;
};
Pool g_pool[16];

size_t pool_limit() {
    static size_t lim = [] {
        const char* e = getenv("ABO_POOL_LIMIT_MB");
        return (e ? (size_t)atoll(e) : (size_t)32768) << 20;
    }();
    return lim;
}

size_t round_size(size_t bytes) {
    const size_t g = bytes < ((size_t)1 << 20) ? 4096 : ((size_t)2 << 20);
    return (bytes + g - 1) / g * g;
}

hipError_t pool_alloc(int dev, size_t bytes, void** p, size_t* cap) {
    const size_t want = round_size(bytes);
    Pool& pl = g_pool[dev & 15];
    {
        std::lock_guard<std::mutex> lk(pl.mu);
        auto it = pl.blocks.lower_bound(want);
        if (it != pl.blocks.end() && it->first <= want + want / 4 + ((size_t)1 << 20)) {
            *p = it->second; *cap = it->first;
            pl.held -= it->first;
            pl.blocks.erase(it);
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) {           // out of memory: give the pool back to the driver and retry once
        std::vector<void*> drop;
        {
            std::lock_guard<std::mutex> lk(pl.mu);
            for (auto& kv : pl.blocks) drop.push_back(kv.second);
            pl.blocks.clear(); pl.held = 0;
        }
        for (void* q : drop) (void)hipFree(q);
        (void)hipGetLastError();
        e = hipMalloc(p, want);
    }
    if (e == hipSuccess) *cap = want;
    return e;
}

void pool_free(int dev, void* p, size_t cap) {
    Pool& pl = g_pool[dev & 15];
    {
        std::lock_guard<std::mutex> lk(pl.mu);
        if (pl.held + cap <= pool_limit()) {
            pl.blocks.emplace(cap, p);
            pl.held += cap;
            return;
        }
    }
    (void)hipFree(p);
}

void pool_trim(int dev) {
    Pool& pl = g_pool[dev & 15];
    std::vector<void*> drop;
    {
        std::lock_guard<std::mutex> lk(pl.mu);
        for (auto& kv : pl.blocks) drop.push_back(kv.second);
        pl.blocks.clear(); pl.held = 0;
    }
    for (void* q : drop) (void)hipFree(q);
}

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int dev = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        release();
        return pool_alloc(dev, bytes, &p, &cap);
    }
    void release() { if (p) pool_free(dev, p, cap); p = nullptr; cap = 0; }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// stream + events of a handle, recycled the same way (hipStreamCreate / ~400 hipEventCreate per step otherwise)
struct ExecCtx {
    hipStream_t stream = nullptr;
    std::vector<hipEvent_t> ev;
};
std::mutex g_ctx_mu;
std::vector<ExecCtx*> g_ctx_free[16];

int dp_for(int d) {
    int p = 1;
    while (p < d) p <<= 1;
    return p;
}

}  // namespace

struct abo_gp {
    std::atomic<int> refs{1};
    abo_params prm{};
    ExecCtx* ctx = nullptr;
    hipStream_t stream = nullptr;      // == ctx->stream
    bool fitted = false;
    int64_t N = 0, Np = 0;
    int d = 0, dp = 0;
    double noise_used = 0.0;
    double logdet = 0.0, quad = 0.0;
    DevBuf Xraw, Xs, ybuf, delta, alpha, tvec, K, W, WT, T, info, scal;
    // posterior workspace
    DevBuf Zdev, Kxz, partial, mu_c, mu_all, var_all, score_all, tk_keys0, tk_keys1, tk_idx0, tk_idx1, top_val, top_idx;
    abo_timings tm{};

    std::vector<hipEvent_t>& evs() { return ctx->ev; }

    void set_device(int dev) {
        DevBuf* all[] = {&Xraw, &Xs, &ybuf, &delta, &alpha, &tvec, &K, &W, &WT, &T, &info, &scal, &Zdev, &Kxz,
                         &partial, &mu_c, &mu_all, &var_all, &score_all, &tk_keys0, &tk_keys1, &tk_idx0, &tk_idx1,
                         &top_val, &top_idx};
        for (DevBuf* b : all) b->dev = dev;
    }

    void free_all() {
        DevBuf* all[] = {&Xraw, &Xs, &ybuf, &delta, &alpha, &tvec, &K, &W, &WT, &T, &info, &scal, &Zdev, &Kxz,
                         &partial, &mu_c, &mu_all, &var_all, &score_all, &tk_keys0, &tk_keys1, &tk_idx0, &tk_idx1,
                         &top_val, &top_idx};
        for (DevBuf* b : all) b->release();
        if (ctx) {
            std::lock_guard<std::mutex> lk(g_ctx_mu);
            g_ctx_free[prm.device & 15].push_back(ctx);
            ctx = nullptr; stream = nullptr;
        }
    }
    hipError_t events(size_t n) {
        while (ctx->ev.size() < n) {
            hipEvent_t e;
            hipError_t r = hipEventCreate(&e);
            if (r != hipSuccess) return r;
            ctx->ev.push_back(e);
        }
        return hipSuccess;
    }
};

namespace {

int32_t copy_in(void* dst, const void* src, size_t bytes, int32_t space, hipStream_t s) {
    if (bytes == 0) return ABO_OK;
    HIPCHK(hipMemcpyAsync(dst, src, bytes, space == ABO_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    return ABO_OK;
}

int32_t copy_out(void* dst, const void* src, size_t bytes, int32_t space, hipStream_t s) {
    if (bytes == 0) return ABO_OK;
    HIPCHK(hipMemcpyAsync(dst, src, bytes, space == ABO_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    return ABO_OK;
}

float ev_ms(hipEvent_t a, hipEvent_t b) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}

// Right-looking blocked Cholesky, 128-wide panels, then L⁻¹ by recursive doubling.
int32_t factorise(abo_gp* g, double noise, int64_t* info_host) {
    hipStream_t s = g->stream;
    const int Np = (int)g->Np, N = (int)g->N;
    const int64_t ld = g->Np;
    double* K = g->K.as<double>();
    double* W = g->W.as<double>();
    double* WT = g->WT.as<double>();
    int64_t* info = g->info.as<int64_t>();

    HIPCHK(hipEventRecord(g->evs()[0], s));
    HIPCHK(hipMemsetAsync(info, 0, sizeof(int64_t), s));
    KgenArgs ka{};
    ka.Xs = g->Xs.as<double>(); ka.Z = g->Xraw.as<double>(); ka.alpha = nullptr; ka.Kout = K; ka.mu = nullptr;
    ka.ldk = ld; ka.M = N; ka.j0 = 0; ka.Mc = Np; ka.N = N; ka.Np = Np; ka.d = g->d; ka.dp = g->dp;
    ka.family = g->prm.family; ka.s = 1.0 / g->prm.ell; ka.sigma_f2 = g->prm.sigma_f2; ka.mean_c = 0.0;
    HIPCHK(launch_kgen(ka, s));
    HIPCHK(launch_diag_fix(K, ld, N, Np, noise, s));
    HIPCHK(hipMemsetAsync(W, 0, sizeof(double) * ld * ld, s));
    HIPCHK(hipMemsetAsync(WT, 0, sizeof(double) * ld * ld, s));
    HIPCHK(hipEventRecord(g->evs()[1], s));

    const int T = Np / TB;
    for (int p = 0; p < T; ++p) {
        const int r0 = p * TB;
        HIPCHK(launch_chol_diag(K, W, WT, ld, r0, info, s));
        const int rem = Np - r0 - TB;
        if (rem <= 0) break;
        GemmArgs a{};
        // panel solve  L[r,p] = A[r,p] · Linv_ppᵀ   (in place; each workgroup owns its 128 rows)
        a.A = K + (int64_t)(r0 + TB) * ld + r0; a.lda = ld;
        a.B = W + (int64_t)r0 * ld + r0; a.ldb = ld;
        a.C = K + (int64_t)(r0 + TB) * ld + r0; a.ldc = ld;
        a.M = rem; a.N = TB; a.K = TB; a.kmode = K_FULL; a.lower_only = 0; a.batch = 1;
        a.alpha = 1.0; a.beta = 0.0; a.info = info;
        HIPCHK(launch_gemm_nt(a, s));
        // trailing update  A[r,c] −= L[r,p]·L[c,p]ᵀ  on the lower triangle
        GemmArgs u{};
        u.A = K + (int64_t)(r0 + TB) * ld + r0; u.lda = ld;
        u.B = u.A; u.ldb = ld;
        u.C = K + (int64_t)(r0 + TB) * ld + (r0 + TB); u.ldc = ld;
        u.M = rem; u.N = rem; u.K = TB; u.kmode = K_FULL; u.lower_only = 1; u.batch = 1;
        u.alpha = -1.0; u.beta = 1.0; u.info = info;
        HIPCHK(launch_gemm_nt(u, s));
    }
    HIPCHK(hipEventRecord(g->evs()[2], s));
    HIPCHK(hipMemcpyAsync(info_host, info, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (*info_host != 0) return ABO_OK;   // caller decides (retry with jitter or ENOTPD)

    // W = L⁻¹: [[W11,0],[−W22·L21·W11, W22]] level by level (block size s doubles each level)
    double* Tt = g->T.as<double>();
    for (int64_t sz = TB; sz < Np; sz *= 2) {
        const int64_t two = 2 * sz;
        const int nfull = (int)(Np / two);
        const int64_t rrem = Np - (int64_t)nfull * two;
        for (int pass = 0; pass < 2; ++pass) {
            const bool ragged = pass == 1;
            if (ragged && rrem <= sz) break;
            if (!ragged && nfull == 0) continue;
            const int64_t r1 = ragged ? (int64_t)nfull * two : 0;
            const int64_t r2 = r1 + sz;
            const int64_t s2 = ragged ? rrem - sz : sz;
            const int batch = ragged ? 1 : nfull;
            const int64_t bstride = two * (ld + 1);
            double* Tp = Tt + (ragged ? (int64_t)nfull * sz * sz : 0);
            GemmArgs a{};   // Tt[j][i] = Σ_k WT11[j][k]·L21[i][k]
            a.A = WT + r1 * ld + r1; a.lda = ld; a.sA = bstride;
            a.B = K + r2 * ld + r1; a.ldb = ld; a.sB = bstride;
            a.C = Tp; a.ldc = sz; a.sC = sz * sz;
            a.M = (int)sz; a.N = (int)s2; a.K = (int)sz; a.kmode = K_A_UPPER; a.batch = batch;
            a.alpha = 1.0; a.beta = 0.0;
            HIPCHK(launch_gemm_nt(a, s));
            GemmArgs b{};   // W21[i][j] = −Σ_k W22[i][k]·Tt[j][k]   (+ transposed copy into WT)
            b.A = W + r2 * ld + r2; b.lda = ld; b.sA = bstride;
            b.B = Tp; b.ldb = sz; b.sB = sz * sz;
            b.C = W + r2 * ld + r1; b.ldc = ld; b.sC = bstride;
            b.Ct = WT + r1 * ld + r2; b.ldct = ld; b.sCt = bstride;
            b.M = (int)s2; b.N = (int)sz; b.K = (int)s2; b.kmode = K_A_LOWER; b.batch = batch;
            b.alpha = -1.0; b.beta = 0.0;
            HIPCHK(launch_gemm_nt(b, s));
        }
    }
    HIPCHK(hipEventRecord(g->evs()[3], s));
    // alpha = Wᵀ(W·delta)
    HIPCHK(launch_trmv(W, ld, g->delta.as<double>(), g->tvec.as<double>(), Np, 1, s));
    HIPCHK(launch_trmv(WT, ld, g->tvec.as<double>(), g->alpha.as<double>(), Np, 0, s));
    HIPCHK(launch_nlml_terms(K, ld, g->delta.as<double>(), g->alpha.as<double>(), N, g->scal.as<double>(), s));
    HIPCHK(hipEventRecord(g->evs()[4], s));
    double sc[2];
    HIPCHK(hipMemcpyAsync(sc, g->scal.as<double>(), sizeof sc, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    g->logdet = sc[0];
    g->quad = sc[1];
    g->tm.fit_kernel_matrix_ms = ev_ms(g->evs()[0], g->evs()[1]);
    g->tm.fit_cholesky_ms = ev_ms(g->evs()[1], g->evs()[2]);
    g->tm.fit_inverse_ms = ev_ms(g->evs()[2], g->evs()[3]);
    g->tm.fit_alpha_ms = ev_ms(g->evs()[3], g->evs()[4]);
    g->tm.fit_total_ms = ev_ms(g->evs()[0], g->evs()[4]);
    return ABO_OK;
}

int64_t pick_chunk(const abo_gp* g, int64_t M) {
    int64_t mc = g->prm.chunk;
    if (mc <= 0) {
        // ~512 MiB of K_XZ per chunk (measured optimum at N = 8192: tools/chunk_sweep.sh — larger
        // chunks lose L2/MALL reuse of the candidate panels, smaller ones pay launch tails), at least
        // 2048 and at most 65536 candidates
        mc = ((int64_t)1 << 29) / (g->Np * (int64_t)sizeof(double));
        if (mc < 2048) mc = 2048;
        if (mc > 65536) mc = 65536;
    }
    mc = pad_up(mc, TB);
    const int64_t mp = pad_up(M, TB);
    return mc < mp ? mc : mp;
}

// mu / var / score for M candidates into device arrays (any may be null)
int32_t posterior(abo_gp* g, const double* Zd, int64_t M, int kind, double p0, double best_y, double* mu_out,
                  double* var_out, double* score_out) {
    hipStream_t s = g->stream;
    const int64_t Np = g->Np;
    const int T = (int)(Np / TB);
    const int64_t Mc = pick_chunk(g, M);
    HIPCHK(g->Kxz.ensure(sizeof(double) * Mc * Np));
    HIPCHK(g->partial.ensure(sizeof(double) * T * Mc));
    HIPCHK(g->mu_c.ensure(sizeof(double) * Mc));
    const int64_t nchunk = (M + Mc - 1) / Mc;
    HIPCHK(g->events(8 + 6 * (size_t)nchunk));
    g->tm.var_gemm_launches = 0;
    for (int64_t c = 0; c < nchunk; ++c) {
        const int64_t j0 = c * Mc;
        const int64_t m = (M - j0) < Mc ? (M - j0) : Mc;
        const int mcp = (int)pad_up(m, TB);
        hipEvent_t* e = &g->evs()[8 + 6 * c];
        KgenArgs ka{};
        ka.Xs = g->Xs.as<double>(); ka.Z = Zd; ka.alpha = g->alpha.as<double>(); ka.Kout = g->Kxz.as<double>();
        ka.mu = g->mu_c.as<double>(); ka.ldk = Np; ka.M = M; ka.j0 = j0; ka.Mc = mcp; ka.N = (int)g->N;
        ka.Np = (int)Np; ka.d = g->d; ka.dp = g->dp; ka.family = g->prm.family; ka.s = 1.0 / g->prm.ell;
        ka.sigma_f2 = g->prm.sigma_f2; ka.mean_c = g->prm.mean_c;
        HIPCHK(hipEventRecord(e[0], s));
        HIPCHK(launch_kgen(ka, s));
        HIPCHK(hipEventRecord(e[1], s));
        if (var_out || score_out) {
            VarGemmArgs va{};
            va.W = g->W.as<double>(); va.Kxz = g->Kxz.as<double>(); va.partial = g->partial.as<double>();
            va.ldw = Np; va.ldk = Np; va.ldp = Mc; va.Np = (int)Np; va.Mc = mcp;
            HIPCHK(hipEventRecord(e[2], s));
            HIPCHK(launch_var_gemm(va, s));
            HIPCHK(hipEventRecord(e[3], s));
            g->tm.var_gemm_launches += 1;
        }
        FinalizeArgs fa{};
        fa.partial = g->partial.as<double>(); fa.mu_in = g->mu_c.as<double>(); fa.mu_out = mu_out;
        fa.var_out = var_out; fa.score_out = score_out; fa.ldp = Mc; fa.j0 = j0; fa.M = M;
        fa.T = (var_out || score_out) ? T : 0; fa.Mc = mcp; fa.kind = kind; fa.sigma_f2 = g->prm.sigma_f2;
        fa.p0 = p0; fa.best_y = best_y;
        HIPCHK(hipEventRecord(e[4], s));
        HIPCHK(launch_finalize(fa, s));
        HIPCHK(hipEventRecord(e[5], s));
    }
    return ABO_OK;
}

void collect_posterior_timings(abo_gp* g, int64_t M, bool with_var) {
    const int64_t Mc = pick_chunk(g, M);
    const int64_t nchunk = (M + Mc - 1) / Mc;
    double kx = 0, vg = 0, fi = 0;
    for (int64_t c = 0; c < nchunk; ++c) {
        hipEvent_t* e = &g->evs()[8 + 6 * c];
        kx += ev_ms(e[0], e[1]);
        if (with_var) vg += ev_ms(e[2], e[3]);
        fi += ev_ms(e[4], e[5]);
    }
    g->tm.acq_kxz_ms = kx;
    g->tm.acq_var_gemm_ms = vg;
    g->tm.acq_finalize_ms = fi;
    // algorithmic (triangular) flop of the contraction: N²·M, N = true training size
    g->tm.var_gemm_flop = with_var ? (double)g->N * (double)g->N * (double)M : 0.0;
}

int32_t check_fitted(abo_gp* g, int32_t d) {
    if (!g) return fail(ABO_EINVAL, "null handle");
    if (!g->fitted) return fail(ABO_EINVAL, "surrogate is not conditioned on data yet (call abo_fit first)");
    if (d != g->d) return fail(ABO_EDIM, "DimensionMismatch: candidate dimension %d, model dimension %d", d, g->d);
    return ABO_OK;
}

int32_t stage_candidates(abo_gp* g, const double* Z, int64_t M, int32_t z_space, const double** Zd) {
    if (z_space == ABO_DEVICE) { *Zd = Z; return ABO_OK; }
    HIPCHK(g->Zdev.ensure(sizeof(double) * M * g->d));
    HIPCHK(hipMemcpyAsync(g->Zdev.p, Z, sizeof(double) * M * g->d, hipMemcpyHostToDevice, g->stream));
    *Zd = g->Zdev.as<double>();
    return ABO_OK;
}

}  // namespace

extern "C" {

int32_t abo_abi_version(void) { return ABO_ABI_VERSION; }

int32_t abo_last_error(char* buf, size_t cap) {
    if (!buf || cap == 0) return ABO_EINVAL;
    strncpy(buf, g_err, cap - 1);
    buf[cap - 1] = '\0';
    return ABO_OK;
}

int32_t abo_create(const abo_params* params, abo_gp** out) {
    if (!params || !out) return fail(ABO_EINVAL, "abo_create: null argument");
    if (params->family < ABO_KERNEL_SE || params->family > ABO_KERNEL_MATERN32)
        return fail(ABO_EINVAL, "abo_create: unknown kernel family %d", params->family);
    if (!(params->ell > 0.0) || !(params->sigma_f2 > 0.0) || !(params->noise_var >= 0.0) || !(params->jitter >= 0.0) ||
        !std::isfinite(params->mean_c))
        return fail(ABO_EINVAL, "abo_create: need ell > 0, sigma_f2 > 0, noise_var >= 0, jitter >= 0, finite mean");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (params->device < 0 || params->device >= ndev)
        return fail(ABO_EINVAL, "abo_create: device %d not present (%d devices)", params->device, ndev);
    HIPCHK(hipSetDevice(params->device));
    abo_gp* g = new (std::nothrow) abo_gp();
    if (!g) return fail(ABO_ENOMEM, "abo_create: host allocation failed");
    g->prm = *params;
    g->set_device(params->device);
    {
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        auto& fl = g_ctx_free[params->device & 15];
        if (!fl.empty()) { g->ctx = fl.back(); fl.pop_back(); }
    }
    hipError_t e = hipSuccess;
    if (!g->ctx) {
        g->ctx = new (std::nothrow) ExecCtx();
        if (!g->ctx) { delete g; return fail(ABO_ENOMEM, "abo_create: host allocation failed"); }
        e = hipStreamCreateWithFlags(&g->ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete g->ctx; delete g; return fail(ABO_EHIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    }
    g->stream = g->ctx->stream;
    e = g->events(8);
    if (e != hipSuccess) { g->free_all(); delete g; return fail(ABO_EHIP, "hipEventCreate: %s", hipGetErrorString(e)); }
    *out = g;
    return ABO_OK;
}

int32_t abo_retain(abo_gp* gp) {
    if (!gp) return fail(ABO_EINVAL, "null handle");
    gp->refs.fetch_add(1);
    return ABO_OK;
}

int32_t abo_destroy(abo_gp* gp) {
    if (!gp) return ABO_OK;
    if (gp->refs.fetch_sub(1) == 1) {
        (void)hipSetDevice(gp->prm.device);
        if (gp->stream) (void)hipStreamSynchronize(gp->stream);
        gp->free_all();
        delete gp;
    }
    return ABO_OK;
}

int32_t abo_fit(abo_gp* g, const double* X, int64_t N, int32_t d, const double* y, int32_t space, int64_t* info) {
    if (info) *info = 0;
    if (!g || !X || !y) return fail(ABO_EINVAL, "abo_fit: null argument");
    if (N < 1) return fail(ABO_EINVAL, "abo_fit: need at least one training point");
    if (d < 1 || d > 32) return fail(ABO_EINVAL, "abo_fit: input dimension %d outside the supported 1..32", d);
    if (N > (int64_t)1 << 20) return fail(ABO_EINVAL, "abo_fit: N = %lld too large", (long long)N);
    HIPCHK(hipSetDevice(g->prm.device));
    hipStream_t s = g->stream;
    g->fitted = false;
    g->N = N; g->d = d; g->dp = dp_for(d);
    int64_t cap = g->prm.n_max > N ? g->prm.n_max : N;
    (void)cap;
    g->Np = pad_up(N, TB);
    const int64_t Np = g->Np;
    HIPCHK(g->Xraw.ensure(sizeof(double) * N * d));
    HIPCHK(g->Xs.ensure(sizeof(double) * Np * g->dp));
    HIPCHK(g->ybuf.ensure(sizeof(double) * N));
    HIPCHK(g->delta.ensure(sizeof(double) * Np));
    HIPCHK(g->alpha.ensure(sizeof(double) * Np));
    HIPCHK(g->tvec.ensure(sizeof(double) * Np));
    HIPCHK(g->K.ensure(sizeof(double) * Np * Np));
    HIPCHK(g->W.ensure(sizeof(double) * Np * Np));
    HIPCHK(g->WT.ensure(sizeof(double) * Np * Np));
    HIPCHK(g->T.ensure(sizeof(double) * Np * Np));
    HIPCHK(g->info.ensure(sizeof(int64_t)));
    HIPCHK(g->scal.ensure(sizeof(double) * 2));
    int32_t rc = copy_in(g->Xraw.p, X, sizeof(double) * N * d, space, s);
    if (rc) return rc;
    rc = copy_in(g->ybuf.p, y, sizeof(double) * N, space, s);
    if (rc) return rc;
    HIPCHK(launch_scale_points(g->Xraw.as<double>(), g->Xs.as<double>(), (int)N, (int)Np, d, g->dp, 1.0 / g->prm.ell, s));
    HIPCHK(launch_center(g->ybuf.as<double>(), g->delta.as<double>(), (int)N, (int)Np, g->prm.mean_c, s));

    int64_t inf = 0;
    double noise = g->prm.noise_var;
    for (int attempt = 0;; ++attempt) {
        rc = factorise(g, noise, &inf);
        if (rc) return rc;
        if (inf == 0) break;
        if (!(g->prm.jitter > 0.0) || attempt >= 4) {
            if (info) *info = inf;
            return fail(ABO_ENOTPD, "PosDefException: matrix is not positive definite; Cholesky factorization failed at %lld",
                        (long long)inf);
        }
        noise = g->prm.noise_var + g->prm.jitter * std::pow(10.0, attempt);
    }
    g->noise_used = noise;
    g->fitted = true;
    return ABO_OK;
}

int32_t abo_predict(abo_gp* g, const double* Z, int64_t M, int32_t d, int32_t z_space, double* mu, double* var,
                    int32_t out_space) {
    int32_t rc = check_fitted(g, d);
    if (rc) return rc;
    if (M < 0 || (M > 0 && !Z)) return fail(ABO_EINVAL, "abo_predict: bad candidate buffer");
    if (M == 0) return ABO_OK;
    HIPCHK(hipSetDevice(g->prm.device));
    hipStream_t s = g->stream;
    const double* Zd = nullptr;
    rc = stage_candidates(g, Z, M, z_space, &Zd);
    if (rc) return rc;
    double* mu_d = nullptr;
    double* var_d = nullptr;
    if (mu) {
        if (out_space == ABO_DEVICE) mu_d = mu;
        else { HIPCHK(g->mu_all.ensure(sizeof(double) * M)); mu_d = g->mu_all.as<double>(); }
    }
    if (var) {
        if (out_space == ABO_DEVICE) var_d = var;
        else { HIPCHK(g->var_all.ensure(sizeof(double) * M)); var_d = g->var_all.as<double>(); }
    }
    HIPCHK(hipEventRecord(g->evs()[5], s));
    rc = posterior(g, Zd, M, -1, 0.0, 0.0, mu_d, var_d, nullptr);
    if (rc) return rc;
    HIPCHK(hipEventRecord(g->evs()[6], s));
    if (mu && out_space == ABO_HOST) { rc = copy_out(mu, mu_d, sizeof(double) * M, ABO_HOST, s); if (rc) return rc; }
    if (var && out_space == ABO_HOST) { rc = copy_out(var, var_d, sizeof(double) * M, ABO_HOST, s); if (rc) return rc; }
    HIPCHK(hipStreamSynchronize(s));
    collect_posterior_timings(g, M, var != nullptr);
    g->tm.acq_topk_ms = 0.0;
    g->tm.acq_total_ms = ev_ms(g->evs()[5], g->evs()[6]);
    return ABO_OK;
}

int32_t abo_acq(abo_gp* g, const double* Z, int64_t M, int32_t d, int32_t z_space, int32_t kind, double p0,
                double best_y, int64_t idx_base, double* scores, int32_t k, double* top_val, int64_t* top_idx,
                int32_t out_space) {
    int32_t rc = check_fitted(g, d);
    if (rc) return rc;
    if (kind < ABO_ACQ_EI || kind > ABO_ACQ_MEAN) return fail(ABO_EINVAL, "abo_acq: unknown acquisition kind %d", kind);
    if (M < 0 || (M > 0 && !Z)) return fail(ABO_EINVAL, "abo_acq: bad candidate buffer");
    if (k < 0 || k > 1024) return fail(ABO_EINVAL, "abo_acq: k = %d outside 0..1024", k);
    if (k > 0 && (!top_val || !top_idx)) return fail(ABO_EINVAL, "abo_acq: k > 0 needs top_val and top_idx");
    HIPCHK(hipSetDevice(g->prm.device));
    hipStream_t s = g->stream;
    double* sc_d = nullptr;
    if (M > 0) {
        const double* Zd = nullptr;
        rc = stage_candidates(g, Z, M, z_space, &Zd);
        if (rc) return rc;
        if (scores && out_space == ABO_DEVICE) sc_d = scores;
        else { HIPCHK(g->score_all.ensure(sizeof(double) * M)); sc_d = g->score_all.as<double>(); }
        HIPCHK(hipEventRecord(g->evs()[5], s));
        const bool need_var = kind != ABO_ACQ_MEAN;
        if (need_var) {
            rc = posterior(g, Zd, M, kind, p0, best_y, nullptr, nullptr, sc_d);
        } else {
            // −mu only: skip the contraction (scores come from the mean pass)
            HIPCHK(g->mu_all.ensure(sizeof(double) * M));
            rc = posterior(g, Zd, M, -1, 0.0, 0.0, g->mu_all.as<double>(), nullptr, nullptr);
            if (rc) return rc;
            FinalizeArgs fa{};
            fa.partial = nullptr; fa.mu_in = g->mu_all.as<double>(); fa.score_out = sc_d; fa.ldp = 0; fa.j0 = 0; fa.M = M;
            fa.T = 0; fa.Mc = (int)M; fa.kind = ABO_ACQ_MEAN; fa.sigma_f2 = g->prm.sigma_f2;
            HIPCHK(launch_finalize(fa, s));
        }
        if (rc) return rc;
        HIPCHK(hipEventRecord(g->evs()[6], s));
    } else {
        HIPCHK(hipEventRecord(g->evs()[5], s));
        HIPCHK(hipEventRecord(g->evs()[6], s));
    }
    if (k > 0) {
        const int64_t we = topk_workspace_entries(M, k);
        HIPCHK(g->tk_keys0.ensure(sizeof(uint64_t) * we));
        HIPCHK(g->tk_keys1.ensure(sizeof(uint64_t) * we));
        HIPCHK(g->tk_idx0.ensure(sizeof(int64_t) * we));
        HIPCHK(g->tk_idx1.ensure(sizeof(int64_t) * we));
        TopkWork w{{g->tk_keys0.as<uint64_t>(), g->tk_keys1.as<uint64_t>()}, {g->tk_idx0.as<int64_t>(), g->tk_idx1.as<int64_t>()}};
        double* tv = top_val;
        int64_t* ti = top_idx;
        if (out_space == ABO_HOST) {
            HIPCHK(g->top_val.ensure(sizeof(double) * k));
            HIPCHK(g->top_idx.ensure(sizeof(int64_t) * k));
            tv = g->top_val.as<double>();
            ti = g->top_idx.as<int64_t>();
        }
        HIPCHK(launch_topk(sc_d, M, k, idx_base, w, tv, ti, s));
        if (out_space == ABO_HOST) {
            rc = copy_out(top_val, tv, sizeof(double) * k, ABO_HOST, s); if (rc) return rc;
            rc = copy_out(top_idx, ti, sizeof(int64_t) * k, ABO_HOST, s); if (rc) return rc;
        }
    }
    HIPCHK(hipEventRecord(g->evs()[7], s));
    if (scores && out_space == ABO_HOST && M > 0) {
        rc = copy_out(scores, sc_d, sizeof(double) * M, ABO_HOST, s);
        if (rc) return rc;
    }
    HIPCHK(hipStreamSynchronize(s));
    if (M > 0) collect_posterior_timings(g, M, kind != ABO_ACQ_MEAN);
    g->tm.acq_topk_ms = ev_ms(g->evs()[6], g->evs()[7]);
    g->tm.acq_total_ms = ev_ms(g->evs()[5], g->evs()[7]);
    return ABO_OK;
}

int32_t abo_nlml(abo_gp* g, double* out) {
    if (!g || !out) return fail(ABO_EINVAL, "abo_nlml: null argument");
    if (!g->fitted) return fail(ABO_EINVAL, "surrogate is not conditioned on data yet (call abo_fit first)");
    *out = 0.5 * ((double)g->N * std::log(2.0 * M_PI) + g->logdet + g->quad);
    return ABO_OK;
}

int32_t abo_get_n(abo_gp* g, int64_t* N, int32_t* d) {
    if (!g) return fail(ABO_EINVAL, "null handle");
    if (N) *N = g->fitted ? g->N : 0;
    if (d) *d = g->fitted ? g->d : 0;
    return ABO_OK;
}

int32_t abo_get_timings(abo_gp* g, abo_timings* out) {
    if (!g || !out) return fail(ABO_EINVAL, "abo_get_timings: null argument");
    *out = g->tm;
    return ABO_OK;
}

int32_t abo_get_factor(abo_gp* g, double* L, double* alpha, double* Linv) {
    if (!g) return fail(ABO_EINVAL, "null handle");
    if (!g->fitted) return fail(ABO_EINVAL, "surrogate is not conditioned on data yet (call abo_fit first)");
    HIPCHK(hipSetDevice(g->prm.device));
    const int64_t N = g->N, Np = g->Np;
    hipStream_t s = g->stream;
    if (L) {
        HIPCHK(hipMemcpy2DAsync(L, sizeof(double) * N, g->K.p, sizeof(double) * Np, sizeof(double) * N, N,
                                hipMemcpyDeviceToHost, s));
    }
    if (Linv) {
        HIPCHK(hipMemcpy2DAsync(Linv, sizeof(double) * N, g->W.p, sizeof(double) * Np, sizeof(double) * N, N,
                                hipMemcpyDeviceToHost, s));
    }
    if (alpha) HIPCHK(hipMemcpyAsync(alpha, g->alpha.p, sizeof(double) * N, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (L)   // off-diagonal upper blocks of the in-place factor still hold K: present a clean L
        for (int64_t i = 0; i < N; ++i)
            for (int64_t j = i + 1; j < N; ++j) L[i * N + j] = 0.0;
    return ABO_OK;
}

int32_t abo_pool_trim(int32_t device) {
    if (device < 0 || device > 15) return fail(ABO_EINVAL, "abo_pool_trim: bad device %d", device);
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipDeviceSynchronize());
    pool_trim(device);
    return ABO_OK;
}

int32_t abo_test_gemm_nt(int32_t device, const double* A, const double* B, double* C, int32_t M, int32_t N, int32_t K,
                         int64_t lda, int64_t ldb, int64_t ldc, double alpha, double beta) {
    if (!A || !B || !C) return fail(ABO_EINVAL, "abo_test_gemm_nt: null argument");
    if (M <= 0 || N <= 0 || K <= 0 || M % 128 || N % 128 || K % 16 || (lda & 1) || (ldb & 1))
        return fail(ABO_EINVAL, "abo_test_gemm_nt: M, N must be multiples of 128, K of 16, lda/ldb even");
    HIPCHK(hipSetDevice(device));
    GemmArgs a{};
    a.A = A; a.B = B; a.C = C; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
    a.kmode = K_FULL; a.batch = 1; a.alpha = alpha; a.beta = beta;
    HIPCHK(launch_gemm_nt(a, nullptr));
    HIPCHK(hipStreamSynchronize(nullptr));
    return ABO_OK;
}

}  // extern "C"
