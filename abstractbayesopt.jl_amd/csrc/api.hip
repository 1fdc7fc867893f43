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

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <limits>
#include <map>
#include <mutex>
#include <set>
#include <new>
#include <string>
#include <vector>

#include "../../include/abo_hip.h"
#include "abo_internal.h"
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

// The end of a call waits for its stream.  hipStreamSynchronize blocks the host thread on an interrupt as soon as the work is not done at
// once: 20 – 40 µs from the last kernel's end to the return, three times a BO step on the incremental path (0.7 ms) and once per 0.15 ms
// step at the reference's own sizes.  Short waits are therefore polled (hipStreamQuery reads the queue's completion signal, ≈ 1 µs a
// call); a wait that outlasts ABO_SPIN_WAIT_US (default 400, 0 = always block) falls back to the blocking call — a 450 ms
// posterior does not spin a core.
hipError_t wait_stream(hipStream_t s) {
    static const long spin_us = [] { const char* e = getenv("ABO_SPIN_WAIT_US"); return e ? atol(e) : 400L; }();
    if (spin_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        bool pending = false;
        for (;;) {
            const hipError_t e = hipStreamQuery(s);
            if (e != hipErrorNotReady) {
                if (pending && e == hipSuccess) (void)hipGetLastError();      // hipErrorNotReady must not stay in the runtime's last-error slot:
                return e;                                                     // the launch wrappers return hipGetLastError()
            }
            pending = true;
            if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > spin_us) break;
        }
        (void)hipGetLastError();
    }
    return hipStreamSynchronize(s);
}

// ---- process teardown guard (abo_internal.h) -------------------------------------------------------------------------------
std::atomic<bool> g_exiting{false};
std::atomic<bool> g_exit_armed{false};
std::mutex g_exit_mu;
std::vector<void (*)()> g_exit_hooks;
void exit_hook() {
    g_exiting.store(true);
    std::vector<void (*)()> hooks;
    { std::lock_guard<std::mutex> lk(g_exit_mu); hooks.swap(g_exit_hooks); }
    for (auto f : hooks) f();
}
// the library's own unload (dlclose, or static destruction at exit if the atexit hook was never armed)
__attribute__((destructor)) void lib_unload() { g_exiting.store(true); }

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

size_t oz_scratch_limit() {
    const char* e = getenv("ABO_OZ_SCRATCH_LIMIT_MB");
    return e ? (size_t)atoll(e) << 20 : ~(size_t)0;
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
    if (g_exiting.load()) return;        // the process is going away: the driver reclaims device memory, the runtime may be gone
    Pool& pl = g_pool[dev & 15];
    std::vector<void*> evict;
    bool kept = false;
    {
        std::lock_guard<std::mutex> lk(pl.mu);
        // A full pool gives up its LARGEST blocks to keep a smaller one (round 5: after a config-5 model — 17 GB of K_ZX and its
        // factors held — the few-KB buffers of a small model that followed it were hipFree'd and hipMalloc'ed on every BO step,
        // 0.17 → 0.35 ms per step; a large block is the one whose next hipMalloc is amortised over the most work)
        while (pl.held + cap > pool_limit() && !pl.blocks.empty() && std::prev(pl.blocks.end())->first > cap) {
            auto it = std::prev(pl.blocks.end());
            evict.push_back(it->second);
            pl.held -= it->first;
            pl.blocks.erase(it);
        }
        if (pl.held + cap <= pool_limit()) {
            pl.blocks.emplace(cap, p);
            pl.held += cap;
            kept = true;
        }
    }
    for (void* q : evict) (void)hipFree(q);
    if (!kept) (void)hipFree(p);
}

void pool_trim(int dev) {
    if (g_exiting.load()) return;
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
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }            // every early return (HIPCHK) hands its buffers back to the pool
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        release();
        return pool_alloc(dev, bytes, &p, &cap);
    }
    void release() { if (p) pool_free(dev, p, cap); p = nullptr; cap = 0; }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// stream + events of a handle, recycled the same way (hipStreamCreate / ~400 hipEventCreate per step otherwise)
// pin: a small page-locked host block that lives with the context.  A device-to-host copy of a few bytes into PAGEABLE memory blocks
// the host until it has happened (12 – 14 µs each on this runtime, tools/memcpy_probe.hip; 2.5 – 3.7 µs and asynchronous into
// pinned memory) — four of them were a quarter of a BO step at the reference's own sizes.  The fit's scalars and the selected
// (score, index) pairs land here and are copied on by the host after the call's one synchronisation.
constexpr size_t PIN_BYTES = 32768, PIN_OUT = 64;      // [0,16) fit scalars, [16,24) info, [PIN_OUT, …) selected pairs
struct ExecCtx {
    hipStream_t stream = nullptr;
    std::vector<hipEvent_t> ev;
    char* pin = nullptr;
};
std::mutex g_ctx_mu;
std::vector<ExecCtx*> g_ctx_free[16];

// the small read-backs of ONE call: each goes into the context's pinned block (asynchronously) while it fits, straight to its
// destination otherwise; flush() — behind the call's stream synchronisation — copies the staged ones on
struct PinStage {
    ExecCtx* c;
    size_t off = PIN_OUT;
    struct Item { void* dst; size_t off, n; } it[8];
    int n = 0;
    explicit PinStage(ExecCtx* ctx) : c(ctx) {}
    hipError_t d2h(void* dst, const void* src, size_t bytes, hipStream_t s) {
        if (bytes == 0) return hipSuccess;
        const size_t a = (bytes + 15) & ~(size_t)15;
        if (c && c->pin && n < 8 && off + a <= PIN_BYTES) {
            it[n++] = Item{dst, off, bytes};
            const hipError_t e = hipMemcpyAsync(c->pin + off, src, bytes, hipMemcpyDeviceToHost, s);
            off += a;
            return e;
        }
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s);
    }
    void flush() {
        for (int i = 0; i < n; ++i) memcpy(it[i].dst, c->pin + it[i].off, it[i].n);
        n = 0;
    }
};

// padded coordinate count: 1, 2, 4, 8, 16, 32 (register-resident kernels), beyond that a multiple of 32 (slab kernels)
int dp_for(int d) {
    if (d > 32) return (d + 31) / 32 * 32;
    int p = 1;
    while (p < d) p <<= 1;
    return p;
}

// a local scratch buffer of one call: on every exit path the stream is drained before the block goes back to the pool
// (an error return must not hand memory that queued work still touches to the next handle)
struct ScratchBuf {
    DevBuf b;
    hipStream_t s;
    ScratchBuf(int dev, hipStream_t st) : s(st) { b.dev = dev; }
    ~ScratchBuf() { if (b.p) { (void)wait_stream(s); b.release(); } }
};

std::atomic<uint64_t> g_storage_gen{1};

// contraction engine of the posterior variance: process default (ABO_CONTRACTION = auto | fp64 | int8 | int8:<moduli>),
// overridable per handle (abo_set_contraction)
constexpr int OZ_DEFAULT_NMOD = 14;   // P ≈ 2^110: the fixed-point images keep 50+ bits of W per row and 52–53 bits of K_XZ
constexpr int OZ_AUTO_MIN_NP = 1280;  // below this the fp64 kernels win (tools/engine_crossover.py, profiles/r06_engine_crossover.txt: fp64/int8 = 0.97 at
                                      // 1024, 1.09 at 1280, 1.21 at 1536, 1.90 at 4096; round 2's engine crossed at 1536)
std::atomic<int> g_oz_engine{-1}, g_oz_nmod{OZ_DEFAULT_NMOD};
void oz_defaults(int* engine, int* nmod) {
    int e = g_oz_engine.load();
    if (e < 0) {
        e = ABO_CONTRACT_AUTO;
        int n = OZ_DEFAULT_NMOD;
        if (const char* v = getenv("ABO_CONTRACTION")) {
            if (!strncmp(v, "fp64", 4)) e = ABO_CONTRACT_FP64;
            else if (!strncmp(v, "int8", 4)) {
                e = ABO_CONTRACT_INT8;
                if (v[4] == ':') { const int q = atoi(v + 5); if (q >= 8 && q <= OZ_MAXMOD) n = q; }
            }
        }
        g_oz_nmod.store(n);
        g_oz_engine.store(e);
    }
    *engine = e;
    *nmod = g_oz_nmod.load();
}

}  // namespace

// Heavy device state, shared by every view (copy / appended model) that descends from one fit.
// Rows < N of Xs / delta / L / W / WT are immutable for every live view of size N; an append only
// writes row N of the largest live view, so older views stay valid and rollback is free.
struct Storage {
    std::atomic<int> refs{1};
    // process-wide unique id: what a candidate set remembers of the factor it is synced with (a freed Storage's address
    // is readily reused by the next refit of the same size; its id never is)
    const uint64_t gen = g_storage_gen.fetch_add(1);
    int dev = 0;
    int d = 0, dp = 0;
    int64_t cap = 0;        // padded capacity = leading dimension of K/W/WT (multiple of 128)
    double noise_used = 0.0;
    // sizes N of the live views on this storage.  A view may append in place only if no live view
    // is larger (rows ≥ its N are then dead: stale rows of discarded fantasy branches are masked
    // everywhere by the view's own N, and the append overwrites row N).
    std::mutex mu;
    std::multiset<int64_t> live;
    void add_view(int64_t n) { std::lock_guard<std::mutex> lk(mu); live.insert(n); }
    void drop_view(int64_t n) { std::lock_guard<std::mutex> lk(mu); auto it = live.find(n); if (it != live.end()) live.erase(it); }
    int64_t max_live() { std::lock_guard<std::mutex> lk(mu); return live.empty() ? 0 : *live.rbegin(); }
    // An append in place writes rows [n_old, n_new): allowed only while no live view reaches beyond n_old — decided AND claimed
    // (the new view registered) under one lock, so that two host threads appending to copies of one model cannot both take the
    // rows (the loser refits into storage of its own: copy-on-write).  A failed append gives the claim back (drop_view).
    bool claim_rows(int64_t n_old, int64_t n_new) {
        std::lock_guard<std::mutex> lk(mu);
        if (!live.empty() && *live.rbegin() > n_old) return false;
        live.insert(n_new);
        return true;
    }
    DevBuf Xraw, Xs, ybuf, delta, K, W, WT;
    void set_device(int dv) {
        dev = dv;
        DevBuf* all[] = {&Xraw, &Xs, &ybuf, &delta, &K, &W, &WT};
        for (DevBuf* b : all) b->dev = dv;
    }
    void release_all() {
        DevBuf* all[] = {&Xraw, &Xs, &ybuf, &delta, &K, &W, &WT};
        for (DevBuf* b : all) b->release();
    }
};

void storage_unref(Storage* st) {
    if (st && st->refs.fetch_sub(1) == 1) { st->release_all(); delete st; }
}

struct abo_gp {
    std::atomic<int> refs{1};
    abo_params prm{};
    ExecCtx* ctx = nullptr;
    hipStream_t stream = nullptr;      // == ctx->stream
    Storage* st = nullptr;             // null until conditioned on data
    bool fitted = false;
    int64_t N = 0, Np = 0;             // this view's number of factor rows and its 128-padded size
    int64_t npts = 0;                  // training points (== N unless gradient-enhanced: N = p_out·npts)
    int d = 0, dp = 0;
    int p_out = 1;                     // outputs per point: 1 = StandardGP, d+1 = GradientGP (f + gradient)
    double mean_vec[MAX_P] = {0};      // prior mean per output (gradConstMean; [0] = mean_c for p_out == 1)
    double logdet = 0.0, quad = 0.0;
    // bordered-append bookkeeping (valid when this view was produced by abo_append)
    bool from_append = false;
    double ap_s2 = 0.0, ap_beta = 0.0; // Schur complement l_nn² and (y* − μ(x*))/l_nn²
    std::vector<double> ap_x;          // the point a one-row append added, as the host handed it over (abo_cand_downdate matches it against the q-EI chain)
    double ap_s2v[MAX_P] = {0}, ap_betav[MAX_P] = {0};   // gradient-enhanced append: the same per appended row (p_out of them)
    DevBuf alpha, vext, tvec, T, info, scal;
    // host landing zone of the fit's scalars ({log det, δᵀα} and the LAPACK-style info): read back by one asynchronous copy at
    // the end of the fit's launches and looked at after the NEXT stream synchronisation — the fit's own (abo_fit) or, for
    // abo_fit_acq, the one behind the acquisition launches that were queued right behind the fit
    double h_sc_[2] = {0.0, 0.0};      // (fallback landing zone when the context has no pinned block)
    int64_t h_info_ = 0;
    double* h_sc() { return ctx && ctx->pin ? reinterpret_cast<double*>(ctx->pin) : h_sc_; }
    int64_t& h_info() { return ctx && ctx->pin ? *reinterpret_cast<int64_t*>(ctx->pin + 16) : h_info_; }
    bool fit_small_path = false;
    const double* in_x = nullptr;      // the caller's device arrays while a one-launch fit reads them itself (abo_fit: no staging copies);
    const double* in_y = nullptr;      // null: the model's own copies
    // posterior workspace
    DevBuf Zdev, Kxz, partial, mu_c, mu_all, var_all, score_all, tk_keys0, tk_keys1, tk_idx0, tk_idx1, top_val, top_idx;
    // int8-residue contraction (ozaki.hip): engine choice, the residue planes of this view's W (valid for oz_gen / oz_N /
    // oz_plan.n) and the per-chunk scratch
    int oz_engine = ABO_CONTRACT_AUTO, oz_nmod = 0;
    int* oz_ctr_clean = nullptr;       // the tile-counter block of oz_badc known to hold zeros (OzVarArgs::ctr_clean)
    OzPlan oz_plan{};
    bool oz_prepare_pending = false;  // events 8/9 of the current call bracket a rebuild of the residue planes of W (read with its timings)
    uint64_t oz_gen = 0;
    int64_t oz_N = -1;
    int64_t last_chunk = 0;          // candidates per chunk of the last posterior call (its events are read back with it)
    DevBuf oz_WR, oz_sexp, oz_badr, oz_KR, oz_U, oz_badc;
    abo_timings tm{};

    std::vector<hipEvent_t>& evs() { return ctx->ev; }
    int64_t ld() const { return st->cap; }

    void set_device(int dev) {
        DevBuf* all[] = {&alpha, &vext, &tvec, &T, &info, &scal, &Zdev, &Kxz, &partial, &mu_c, &mu_all, &var_all,
                         &score_all, &tk_keys0, &tk_keys1, &tk_idx0, &tk_idx1, &top_val, &top_idx,
                         &oz_WR, &oz_sexp, &oz_badr, &oz_KR, &oz_U, &oz_badc};
        for (DevBuf* b : all) b->dev = dev;
    }

    void free_all() {
        DevBuf* all[] = {&alpha, &vext, &tvec, &T, &info, &scal, &Zdev, &Kxz, &partial, &mu_c, &mu_all, &var_all,
                         &score_all, &tk_keys0, &tk_keys1, &tk_idx0, &tk_idx1, &top_val, &top_idx,
                         &oz_WR, &oz_sexp, &oz_badr, &oz_KR, &oz_U, &oz_badc};
        for (DevBuf* b : all) b->release();
        oz_N = -1;
        oz_ctr_clean = nullptr;
        if (st && fitted) st->drop_view(N);
        storage_unref(st);
        st = nullptr;
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

// candidate set resident in HBM with its posterior (C5: O(N·M) down-dates instead of re-evaluation)
struct abo_cand {
    int device = 0, d = 0;
    int64_t M = 0;
    uint64_t synced_gen = 0;              // Storage::gen of the factor the mu/var belong to (0 = none)
    int64_t synced_N = -1;
    uint64_t bak_gen = 0;                 // abo_cand_save snapshot
    int64_t bak_N = -1;
    DevBuf Z, mu, var, score, cdot, tk_keys0, tk_keys1, tk_idx0, tk_idx1, top_val, top_idx, mu_bak, var_bak;
    // resident K_ZX (candidate-major, kzx_ld doubles per candidate, column k = training row k of synced_st): kept when
    // it fits the budget (ABO_CAND_KZX_GIB, default 64), so that a down-date streams it once instead of re-evaluating
    // N·M kernel values; kzx_ld = 0 → not resident, the down-date recomputes
    DevBuf Kzx;
    int64_t kzx_ld = 0;
    // block form of greedy q-EI (qei.hip).  Base = the model the set was synced with when the state was last reset (gen, N).
    //   qblk   [nblk_cap][T16][Mp]  covariances Cov(z, x_t) of the block points under the model of the block's build (ring of blocks)
    //   qchain [rows][Mp]           c_i(z) = Cov_{i−1}(z, x_i) of every conditioning since the base, in order: entries [0, nreal) are
    //                               REAL appends (abo_cand_downdate), entries [nreal, nchain) the fantasies of the open / last batch
    // Blocks and real entries OUTLIVE a batch: the next batch on the appended model keeps them (a BO step's picks are mostly the
    // previous step's runners-up: their columns exist already), a block column is corrected by the chain entries made since the
    // block's build (slot_base).  The fantasies of the last batch stay until the next one begins: appending its picks for real, in
    // order, finds each down-date column there (c_i does not depend on the observed value).
    struct Qei {
        bool open = false;
        uint64_t gen = 0;
        int64_t N = -1, Mp = 0;
        int64_t idx_base = -1;                    // the global index of the shard's first candidate the slots below were keyed with (−1: none yet)
        int T16 = 0, nblk_cap = 0, next_blk = 0, qmax = 0;
        std::vector<int64_t> slot_gidx;           // global candidate index per block row (−1: empty)
        std::vector<double> slot_x;               // [rows][d] the block points
        std::vector<int> blk_base;                // [nblk_cap] chain entries that existed when the block was built (already in its columns)
        int nreal = 0, nchain = 0, chain_rows = 0;
        std::vector<double> chain_x;              // [nchain][d] the conditioned points
        std::vector<double> chain_s;              // [nchain] s_i = σ²_{i−1}(x_i) + σ²_n
        // statistics of the current / last batch
        int builds = 0, batch0 = 0;               // batch0: nchain when the batch began
        double block_ms = 0.0, pass_ms = 0.0, pass_bytes = 0.0, pass_flop = 0.0;
    } qei;
    DevBuf qblk, qchain, qwork, qrec, qmu, qvar, qdev;
    void set_device(int dev) {
        device = dev;
        DevBuf* all[] = {&Z, &mu, &var, &score, &cdot, &tk_keys0, &tk_keys1, &tk_idx0, &tk_idx1, &top_val, &top_idx, &mu_bak, &var_bak, &Kzx,
                         &qblk, &qchain, &qwork, &qrec, &qmu, &qvar, &qdev};
        for (DevBuf* b : all) b->dev = dev;
    }
    void free_all() {
        DevBuf* all[] = {&Z, &mu, &var, &score, &cdot, &tk_keys0, &tk_keys1, &tk_idx0, &tk_idx1, &top_val, &top_idx, &mu_bak, &var_bak, &Kzx,
                         &qblk, &qchain, &qwork, &qrec, &qmu, &qvar, &qdev};
        for (DevBuf* b : all) b->release();
        kzx_ld = 0;
        qei = Qei();
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

int32_t factorise(abo_gp* g, double noise, int64_t* info_host);
int32_t fit_small(abo_gp* g, double noise, int64_t* info_host);
void fit_collect(abo_gp* g);

float ev_ms(hipEvent_t a, hipEvent_t b) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) { (void)hipGetLastError(); ms = 0.f; }   // (an event that was never recorded)
    return ms;
}

// Phase events.  Every call is bracketed by two events (its total); the events INSIDE a call — fit phases, per-chunk kernel
// brackets of the posterior — are profiling instrumentation: each costs a host call and a marker packet between two kernels,
// ≈ 40 of them per BO step, which at the reference's own sizes (a step of 0.2 ms) is a fifth of the step.  ABO_PHASE_EVENTS=0
// leaves them out (the phase fields of abo_timings then read 0, the totals stay); default on.
// ABO_PHASE_EVENTS: 1 = always, 0 = never, unset = automatic — on, except while the calling thread works on a model of one row block
// (N ≤ 128: the reference's own loops, where the step is 0.2 ms and the instrumentation a fifth of it).
thread_local int64_t tl_phase_np = (int64_t)1 << 40;     // padded factor rows of the handle the current call works on
bool phase_events() {
    static const int mode = [] { const char* e = getenv("ABO_PHASE_EVENTS"); return !e ? -1 : (e[0] == '0' ? 0 : 1); }();
    return mode < 0 ? tl_phase_np > TB : mode == 1;
}
#define PHASE_EVENT(ev, s) do { if (phase_events()) HIPCHK(hipEventRecord((ev), (s))); } while (0)

// the scalars and phase timings of a finished fit (h_sc / h_info have landed: the stream has been synchronised since)
void fit_collect(abo_gp* g) {
    g->logdet = g->h_sc()[0];
    g->quad = g->h_sc()[1];
    g->tm.fit_total_ms = ev_ms(g->evs()[0], g->evs()[4]);
    if (!phase_events()) {
        g->tm.fit_kernel_matrix_ms = g->tm.fit_cholesky_ms = g->tm.fit_inverse_ms = g->tm.fit_alpha_ms = 0.0;
        return;
    }
    if (g->fit_small_path) {
        g->tm.fit_kernel_matrix_ms = 0.0;
        g->tm.fit_cholesky_ms = ev_ms(g->evs()[1], g->evs()[2]);      // the one launch
        g->tm.fit_inverse_ms = 0.0;
        g->tm.fit_alpha_ms = 0.0;
    } else {
        g->tm.fit_kernel_matrix_ms = ev_ms(g->evs()[0], g->evs()[1]);
        g->tm.fit_cholesky_ms = ev_ms(g->evs()[1], g->evs()[2]);
        g->tm.fit_inverse_ms = ev_ms(g->evs()[2], g->evs()[3]);
        g->tm.fit_alpha_ms = ev_ms(g->evs()[3], g->evs()[4]);
    }
    g->tm.fit_total_ms = ev_ms(g->evs()[0], g->evs()[4]);
}

constexpr size_t EV_BASE = 10;        // 0-4 fit phases, 5-7 acquisition call, 8-9 residue planes of W; from EV_BASE: per-chunk events of a
                                      // posterior call, or (inside a fit) the strip events of the factorisation's look-ahead
constexpr size_t EV_PER_CHUNK = 8;   // kgen 0-1, contraction 2-3, epilogue 4-5, int8 pipeline: end of quantisation 6, end of GEMM 7

// Right-looking blocked Cholesky, 128-wide panels, then L⁻¹ by recursive doubling.
// info_host == nullptr: launches only — the caller synchronises later and then looks at g->h_info / calls fit_collect().
int32_t factorise(abo_gp* g, double noise, int64_t* info_host) {
    hipStream_t s = g->stream;
    Storage* st = g->st;
    const int Np = (int)g->Np, N = (int)g->N;
    const int64_t ld = st->cap;
    double* K = st->K.as<double>();
    double* W = st->W.as<double>();
    double* WT = st->WT.as<double>();
    HIPCHK(g->info.ensure(sizeof(int64_t)));
    int64_t* info = g->info.as<int64_t>();              // the LAPACK-style status

    HIPCHK(hipEventRecord(g->evs()[0], s));
    // whole capacity region: zeros, identity on the padded diagonal (rows ≥ N), K on the active part.  With a panel chain (more than
    // one block) rows < Np of W / WT are zeroed by the chain's own launches — the idle workgroups of each diagonal-block launch take
    // that panel's rows (chol.hip: potf2_pipe_kernel) — instead of two whole-matrix memsets in front of the fit (148 µs at N = 8192).
    const bool split = Np > TB;                            // a single block has no chain: factor + inverse in one launch
    if (ld > Np) HIPCHK(hipMemsetAsync(K, 0, sizeof(double) * ld * ld, s));
    if (!split) {
        HIPCHK(hipMemsetAsync(W, 0, sizeof(double) * ld * ld, s));
        HIPCHK(hipMemsetAsync(WT, 0, sizeof(double) * ld * ld, s));
    } else if (ld > Np) {
        HIPCHK(hipMemsetAsync(W + (int64_t)Np * ld, 0, sizeof(double) * (ld - Np) * ld, s));
        HIPCHK(hipMemsetAsync(WT + (int64_t)Np * ld, 0, sizeof(double) * (ld - Np) * ld, s));
    }
    KgenArgs ka{};
    ka.Xs = st->Xs.as<double>(); ka.Z = st->Xraw.as<double>(); ka.alpha = nullptr; ka.Kout = K; ka.mu = nullptr;
    ka.ldk = ld; ka.M = g->npts; ka.j0 = 0; ka.Mc = Np; ka.N = (int)g->npts; ka.Np = Np; ka.d = g->d; ka.dp = g->dp;
    ka.family = g->prm.family; ka.s = 1.0 / g->prm.ell; ka.sigma_f2 = g->prm.sigma_f2; ka.mean_c = 0.0;
    ka.pt = g->p_out; ka.pc = g->p_out; ka.point_major = 1;      // K_XX: rows and columns point-major → symmetric
    HIPCHK(launch_kgen(ka, s));
    HIPCHK(launch_diag_fix(K, ld, N, (int)ld, noise, s, info));      // also resets the status word
    if (ld > Np) {
        HIPCHK(launch_set_diag(W, ld, Np, (int)ld, 1.0, s));
        HIPCHK(launch_set_diag(WT, ld, Np, (int)ld, 1.0, s));
    }
    PHASE_EVENT(g->evs()[1], s);

    // Two-level blocking: inside a strip of SW columns the 128-wide panels update only the rest of the strip
    // (K = 128 products on a tall, narrow block); the matrix behind the strip is updated once per strip with
    // K = SW — a quarter of the read-modify-write passes over the trailing matrix and four times longer k loops
    // per tile than a panel-by-panel SYRK.
    constexpr int SW = 512;
    // Third level: strips grouped into super-strips of 1024 columns — behind a strip only the rest of its super-strip is updated
    // (K = SW), the matrix behind the super-strip once per super-strip with K = SS: half the passes over the trailing matrix and
    // tiles twice as long (Cholesky at N = 16384 35.0 → 32.8 ms, at N = 8192 8.20 → 8.23: profiles/r03_chol_super_sweep.txt).  A
    // super-strip pays while the square update behind it is large — its in-super-strip update is a launch of 4 × T tiles that lasts
    // ≈ 150 µs whatever T is, against K = 1024 tiles at 70 TFLOP/s instead of 52 (profiles/r04_fit_trace_summary.txt) — so
    // super-strips are used while at least 6144 rows remain, plain strips from there on.
    constexpr int SSmax = 1024, super_rows = 6144;
    // Panel chain: potf2 of the diagonal block (potf2_pipe_kernel) → triangular solve of the rows below it from the operand stream that
    // kernel leaves (trsm_stream_kernel; the stream lives in T, which the factorisation does not use — the blocked inverse behind it
    // does) → in-strip update.  The 128×128 inverses of the diagonal blocks (the seeds of the blocked L⁻¹ below) are not on that chain:
    // all of them are formed by ONE batched launch behind the factorisation.  Measured and not adopted (profiles/r05_notes.md D, removed
    // from the library in round 6): a two-stream look-ahead, one launch per panel with in-launch hand-overs, 256 × 128 trailing tiles;
    // round 6: the same look-ahead on CU-MASKED streams (hipExtStreamCreateWithCUMask: chain and trailing update on disjoint compute
    // units, so the chain never waits for a slot) — 7.4 – 9.1 ms against 7.0 at N = 8192 (profiles/r06_chol_lookahead_cumask_ab.txt).
    if (split) HIPCHK(g->T.ensure(TRSM_STREAM_BYTES));
    double* trsm_ops = g->T.as<double>();
    for (int S0 = 0, SS = SW; S0 < Np; S0 += SS) {
        SS = (Np - S0) >= super_rows ? SSmax : SW;
        const int ss = (Np - S0) < SS ? (Np - S0) : SS;
        for (int s0 = S0; s0 < S0 + ss; s0 += SW) {
            const int sw = (S0 + ss - s0) < SW ? (S0 + ss - s0) : SW;
            for (int r0 = s0; r0 < s0 + sw; r0 += TB) {
                const int rem = Np - r0 - TB;
                if (!split) { HIPCHK(launch_chol_diag(K, W, WT, ld, r0, info, s)); break; }
                HIPCHK(launch_potf2_diag(K, W, WT, ld, r0, info, s, trsm_ops, true));
                if (rem <= 0) break;
                HIPCHK(launch_trsm_stream(K, trsm_ops, ld, r0, rem, info, s));
                // in-strip update  A[r,c] −= L[r,p]·L[c,p]ᵀ  for the strip's remaining columns c, lower tiles only
                const int ncol = s0 + sw - r0 - TB;
                if (ncol > 0) {
                    GemmArgs u{};
                    u.A = K + (int64_t)(r0 + TB) * ld + r0; u.lda = ld;
                    u.B = u.A; u.ldb = ld;
                    u.C = K + (int64_t)(r0 + TB) * ld + (r0 + TB); u.ldc = ld;
                    u.M = rem; u.N = ncol; u.K = TB; u.kmode = K_FULL; u.lower_only = 1; u.batch = 1;
                    u.alpha = -1.0; u.beta = 1.0; u.info = info;
                    HIPCHK(launch_gemm_nt(u, s));
                }
            }
            // update behind the strip, inside its super-strip:  A[r,c] −= L[r,strip]·L[c,strip]ᵀ  for all rows r below the strip and
            // the super-strip's remaining columns c (lower tiles only), K = sw
            const int rows = Np - s0 - sw;
            const int cols = S0 + ss - s0 - sw;
            if (rows > 0 && cols > 0) {
                GemmArgs u{};
                u.A = K + (int64_t)(s0 + sw) * ld + s0; u.lda = ld;
                u.B = u.A; u.ldb = ld;
                u.C = K + (int64_t)(s0 + sw) * ld + (s0 + sw); u.ldc = ld;
                u.M = rows; u.N = cols; u.K = sw; u.kmode = K_FULL; u.lower_only = 1; u.batch = 1;
                u.alpha = -1.0; u.beta = 1.0; u.info = info;
                HIPCHK(launch_gemm_nt(u, s));
            }
        }
        const int rest = Np - S0 - ss;
        if (rest > 0) {
            // trailing update behind the super-strip  A[r,c] −= L[r,super]·L[c,super]ᵀ  on the lower triangle, K = ss
            GemmArgs u{};
            u.A = K + (int64_t)(S0 + ss) * ld + S0; u.lda = ld;
            u.B = u.A; u.ldb = ld;
            u.C = K + (int64_t)(S0 + ss) * ld + (S0 + ss); u.ldc = ld;
            u.M = rest; u.N = rest; u.K = ss; u.kmode = K_FULL; u.lower_only = 1; u.batch = 1;
            u.alpha = -1.0; u.beta = 1.0; u.info = info;
            HIPCHK(launch_gemm_nt(u, s));
        }
    }
    if (split) HIPCHK(launch_trtri_diag_batched(K, W, WT, ld, Np / TB, info, s));
    PHASE_EVENT(g->evs()[2], s);
    // no host round trip here: after a failed pivot every later kernel of the fit either exits on `info` (the GEMMs) or
    // works on finite leftovers whose results are discarded; `info` is read once, with the scalars, at the end

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
            a.alpha = 1.0; a.beta = 0.0; a.info = info;
            HIPCHK(launch_gemm_nt(a, s));
            GemmArgs b{};   // W21[i][j] = −Σ_k W22[i][k]·Tt[j][k]   (+ transposed copy into WT)
            b.A = W + r2 * ld + r2; b.lda = ld; b.sA = bstride;
            b.B = Tp; b.ldb = sz; b.sB = sz * sz;
            b.C = W + r2 * ld + r1; b.ldc = ld; b.sC = bstride;
            b.Ct = WT + r1 * ld + r2; b.ldct = ld; b.sCt = bstride;
            b.M = (int)s2; b.N = (int)sz; b.K = (int)s2; b.kmode = K_A_LOWER; b.batch = batch;
            b.alpha = -1.0; b.beta = 0.0; b.info = info;
            HIPCHK(launch_gemm_nt(b, s));
        }
    }
    PHASE_EVENT(g->evs()[3], s);
    // alpha = Wᵀ(W·delta)
    HIPCHK(launch_trmv(W, ld, st->delta.as<double>(), g->tvec.as<double>(), Np, 1, s));
    HIPCHK(launch_trmv(WT, ld, g->tvec.as<double>(), g->alpha.as<double>(), Np, 0, s));
    HIPCHK(launch_nlml_terms(K, ld, st->delta.as<double>(), g->alpha.as<double>(), N, g->scal.as<double>(), s));
    HIPCHK(hipEventRecord(g->evs()[4], s));
    g->fit_small_path = false;
    HIPCHK(hipMemcpyAsync(g->h_sc(), g->scal.as<double>(), 2 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&g->h_info(), info, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    if (!info_host) return ABO_OK;        // deferred: fit_collect() after the caller's synchronisation
    HIPCHK(wait_stream(s));
    *info_host = g->h_info();
    if (g->h_info() == 0) fit_collect(g);   // else the caller decides (retry with jitter or ENOTPD)
    return ABO_OK;
}

// The whole fit of a small model (N ≤ 128, d ≤ 16, capacity 128) in one launch; same outputs and timing slots as factorise().
int32_t fit_small(abo_gp* g, double noise, int64_t* info_host) {
    hipStream_t s = g->stream;
    Storage* st = g->st;
    int64_t* info = g->info.as<int64_t>();
    HIPCHK(hipEventRecord(g->evs()[0], s));
    PHASE_EVENT(g->evs()[1], s);                            // (the launch resets the status word itself)
    FitSmallArgs fs{};
    fs.Xraw = st->Xraw.as<double>(); fs.y = st->ybuf.as<double>(); fs.Xs = st->Xs.as<double>(); fs.delta = st->delta.as<double>();
    if (g->in_x) { fs.Xraw = g->in_x; fs.y = g->in_y; fs.Xkeep = st->Xraw.as<double>(); fs.ykeep = st->ybuf.as<double>(); }
    fs.alpha = g->alpha.as<double>(); fs.scal = g->scal.as<double>();
    fs.N = (int)g->N; fs.d = g->d; fs.dp = g->dp; fs.family = g->prm.family;
    fs.s = 1.0 / g->prm.ell; fs.sigma_f2 = g->prm.sigma_f2; fs.noise = noise; fs.mean_c = g->prm.mean_c;
    HIPCHK(launch_fit_small(st->K.as<double>(), st->W.as<double>(), st->WT.as<double>(), info, fs, s));
    PHASE_EVENT(g->evs()[2], s);
    HIPCHK(hipEventRecord(g->evs()[4], s));
    g->fit_small_path = true;
    HIPCHK(hipMemcpyAsync(g->h_sc(), g->scal.as<double>(), 2 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&g->h_info(), info, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    if (!info_host) return ABO_OK;        // deferred: fit_collect() after the caller's synchronisation
    HIPCHK(wait_stream(s));
    *info_host = g->h_info();
    if (g->h_info() == 0) fit_collect(g);
    return ABO_OK;
}

// k((z,q),(z,q)) of a gradient output: −2σ_f²φ'(0)/ℓ²  (φ'(0) = −1/2 SE, −5/6 Matérn-5/2, −7/10 Matérn-7/2)
double grad_prior_var(const abo_gp* g) {
    const double c = g->prm.family == ABO_KERNEL_SE ? 1.0 : (g->prm.family == ABO_KERNEL_MATERN52 ? 5.0 / 3.0 : 7.0 / 5.0);
    return c * g->prm.sigma_f2 / (g->prm.ell * g->prm.ell);
}


// exponent of the row scaling of a gradient-enhanced model's derivative outputs: 2^t ≈ √(prior variance of a derivative output / σ_f²)
int oz_grad_exp(const abo_gp* g) {
    if (g->p_out <= 1) return 0;
    return (int)std::lrint(0.5 * std::log2(grad_prior_var(g) / g->prm.sigma_f2));
}

// which engine runs the contraction of a posterior call on this handle (int8 = true), and with how many moduli
bool wants_int8(const abo_gp* g, bool want_var, int pc, int* nmod) {
    if (!want_var || (pc != 1 && pc != g->p_out)) return false;
    int eng = g->oz_engine, nm = g->oz_nmod, de, dn;
    oz_defaults(&de, &dn);
    if (eng == ABO_CONTRACT_AUTO) eng = de;
    if (!nm) nm = dn;
    *nmod = nm;
    // int32 accumulation of the residue GEMMs is exact for k·2^14 < 2^31: beyond 65536 padded training points the engine is not
    // offered (the fp64 kernels take over, whatever was asked for)
    if (g->Np > 65536) return false;
    return eng == ABO_CONTRACT_INT8 || (eng == ABO_CONTRACT_AUTO && g->Np >= OZ_AUTO_MIN_NP);
}

int64_t pick_chunk(const abo_gp* g, int64_t M, bool int8) {
    int64_t mc = g->prm.chunk;
    if (mc <= 0 && int8) {
        // int8 engine: 28 bytes of residue planes and residue products per (candidate, training point); measured at N = 8192:
        // 8192 candidates per chunk 597 ms, 16384 572, 32768 565, 65536 560 per C3 step — 65536 at N = 8192 (15 GB of scratch of the
        // 288), 32768 at N = 16384
        mc = ((int64_t)1 << 32) / (g->Np * (int64_t)sizeof(double));
        if (mc < 2048) mc = 2048;
        if (mc > 65536) mc = 65536;
    } else if (mc <= 0) {
        // ~1 GiB of K_XZ per chunk (measured at N = 8192, tools/chunk_sweep.sh: 2048 candidates per chunk 1072 ms,
        // 4096 1011, 8192 997, 16384 992-995, 32768 998, 65536 996 — small chunks pay launch tails in every kernel,
        // larger ones lose a little L2/MALL reuse of the candidate panels), at least 2048 and at most 65536 candidates
        mc = ((int64_t)1 << 30) / (g->Np * (int64_t)sizeof(double));
        if (mc < 2048) mc = 2048;
        if (mc > 65536) mc = 65536;
    }
    mc = pad_up(mc, TB);
    if (int8) {        // byte offsets inside one residue plane of a chunk stay below 2^31 (32-bit lane offsets in the generator and the DMA)
        const int64_t cap = (((int64_t)1 << 31) / pad_up(g->Np, 256)) / 256 * 256 - 256;
        if (mc > cap) mc = cap;
    }
    const int64_t mp = pad_up(M, TB);
    return mc < mp ? mc : mp;
}

// The int8 engine's scratch — 2 × n bytes per (candidate, factor row) of a chunk, and the n residue planes of W: when the device
// cannot give it (a shared or nearly full GPU), the chunk *Mc is halved down to 4096 candidates, and below that *oz comes back false:
// the call then runs on the fp64 kernels (8 bytes per pair of a chunk four times smaller) instead of failing.
int32_t oz_acquire(abo_gp* g, int nm, int64_t M, int64_t* Mc_io, bool* oz_io) {
    const int64_t Np = g->Np;
    int64_t Mc = *Mc_io;
    bool oz = *oz_io;
    if (oz) {
        if (g->oz_plan.n != nm) {                               // another moduli count: the cached planes of W belong to the old plan
            if (!oz_make_plan(nm, &g->oz_plan)) return fail(ABO_EINVAL, "contraction: %d moduli not supported", nm);
            g->oz_N = -1;
        }
        const int64_t q = pad_up(Np, 256);
        hipError_t e = g->oz_WR.ensure(oz_w_bytes(nm, (int)Np));
        if (e == hipSuccess) e = g->oz_sexp.ensure(sizeof(int) * q);
        if (e == hipSuccess) e = g->oz_badr.ensure(sizeof(int) * q);
        while (e == hipSuccess) {
            // ABO_OZ_SCRATCH_LIMIT_MB caps what the two chunk buffers may take together (a knob for shared devices; the tests use it
            // to walk this very path)
            const size_t kb = oz_k_bytes(nm, (int)Np, (int)Mc);
            e = 2 * kb > oz_scratch_limit() ? hipErrorOutOfMemory : g->oz_KR.ensure(kb);
            if (e == hipSuccess) e = g->oz_U.ensure(kb);
            if (e == hipSuccess) e = g->oz_badc.ensure(sizeof(int) * (pad_up(Mc, 256) + OZ_CTR_INTS));   // + the GEMM's tile counters
            if (e != hipErrorOutOfMemory || Mc <= 4096) break;
            (void)hipGetLastError();
            g->oz_KR.release(); g->oz_U.release();
            Mc = pad_up(Mc / 2, 256);
            e = hipSuccess;
        }
        if (e == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            g->oz_KR.release(); g->oz_U.release(); g->oz_WR.release();
            g->oz_N = -1;
            oz = false;
            Mc = pick_chunk(g, M, false);
        } else HIPCHK(e);
    }
    *Mc_io = Mc;
    *oz_io = oz;
    return ABO_OK;
}

// residue planes of this view's W, once per model (cached on the handle until the storage generation or the view changes)
int32_t oz_planes_of_w(abo_gp* g) {
    hipStream_t s = g->stream;
    if (g->oz_gen != g->st->gen || g->oz_N != g->N) {
        PHASE_EVENT(g->evs()[8], s);
        HIPCHK(oz_prepare_w(g->oz_plan, g->st->W.as<double>(), g->st->cap, (int)g->Np, (int)g->N, g->oz_WR.as<int8_t>(),
                            g->oz_sexp.as<int>(), g->oz_badr.as<int>(), s, g->p_out, oz_grad_exp(g)));
        PHASE_EVENT(g->evs()[9], s);
        g->oz_gen = g->st->gen; g->oz_N = g->N;
        g->oz_prepare_pending = true;                               // both events recorded in THIS call: read with the posterior timings
    }
    return ABO_OK;
}

// mu / var / score for M candidates into device arrays (any may be null)
// pc outputs per candidate (1 = function value; p_out = all outputs of a gradient-enhanced GP), rows by outputs
// unless point_major; mu/var/score arrays then have pc·M entries.
// kstore / ldstore: write K_XZ into a caller-owned candidate-major matrix (pad_up(M,128) rows of ldstore ≥ Np doubles)
// instead of the per-chunk scratch — the resident K_ZX of a candidate set.
int32_t posterior(abo_gp* g, const double* Zd, int64_t Mpts, int kind, double p0, double best_y, double* mu_out,
                  double* var_out, double* score_out, int pc = 1, int point_major = 0, double* kstore = nullptr,
                  int64_t ldstore = 0) {
    const int64_t M = Mpts * pc;                         // candidate rows
    hipStream_t s = g->stream;
    const int64_t Np = g->Np;
    const int T = (int)(Np / TB);
    const bool want_var = var_out || score_out;
    int nm = 0;
    bool oz = wants_int8(g, want_var, pc, &nm);
    int64_t Mc = pick_chunk(g, M, oz);
    { int32_t rc = oz_acquire(g, nm, M, &Mc, &oz); if (rc) return rc; }
    g->last_chunk = Mc;
    {
        KgenArgs probe{};
        probe.pt = g->p_out; probe.dp = g->dp;
        // the fp64 chunk of K_XZ is not materialised when the generator writes the residue planes itself
        if (!kstore && !(oz && kgen_writes_residues(probe, nm))) HIPCHK(g->Kxz.ensure(sizeof(double) * Mc * Np));
    }
    HIPCHK(g->partial.ensure(sizeof(double) * T * Mc));
    HIPCHK(g->mu_c.ensure(sizeof(double) * Mc));
    const int64_t nchunk = (M + Mc - 1) / Mc;
    HIPCHK(g->events(EV_BASE + EV_PER_CHUNK * (size_t)nchunk));
    g->tm.var_gemm_launches = 0;
    g->oz_prepare_pending = false;
    g->tm.oz_prepare_ms = 0.0;                                      // planes cached from an earlier call: nothing spent in this one
    if (oz) { int32_t rc = oz_planes_of_w(g); if (rc) return rc; }
    g->tm.contraction_engine = want_var ? (oz ? ABO_CONTRACT_INT8 : ABO_CONTRACT_FP64) : 0;
    g->tm.oz_nmod = oz ? g->oz_plan.n : 0;
    for (int64_t c = 0; c < nchunk; ++c) {
        const int64_t j0 = c * Mc;
        const int64_t m = (M - j0) < Mc ? (M - j0) : Mc;
        const int mcp = (int)pad_up(m, TB);
        hipEvent_t* e = &g->evs()[EV_BASE + EV_PER_CHUNK * c];
        KgenArgs ka{};
        double* kchunk = kstore ? kstore + j0 * ldstore : g->Kxz.as<double>();
        const int64_t ldk = kstore ? ldstore : Np;
        ka.Xs = g->st->Xs.as<double>(); ka.Z = Zd; ka.alpha = g->alpha.as<double>(); ka.Kout = kchunk;
        ka.mu = g->mu_c.as<double>(); ka.ldk = ldk; ka.M = Mpts; ka.j0 = j0; ka.Mc = mcp; ka.N = (int)g->npts;
        ka.pt = g->p_out; ka.pc = pc; ka.point_major = point_major;
        for (int q = 0; q < MAX_P; ++q) ka.mean_vec[q] = g->mean_vec[q];
        ka.Np = (int)Np; ka.d = g->d; ka.dp = g->dp; ka.family = g->prm.family; ka.s = 1.0 / g->prm.ell;
        ka.sigma_f2 = g->prm.sigma_f2; ka.mean_c = g->prm.mean_c;
        // int8 engine: the generator writes the residue planes of the chunk itself (the fp64 K_XZ is then only materialised for a
        // caller that keeps it — the resident K_ZX of a candidate set)
        const bool fused = oz && kgen_writes_residues(ka, g->oz_plan.n);
        if (fused) {
            const int64_t q = pad_up(Np, 256);
            ka.res = g->oz_KR.as<int8_t>(); ka.res_ld = q; ka.res_plane = pad_up(mcp, 256) * q;
            // (a gradient-enhanced model's scaled chunk is bounded by σ_f²·√2, its all-output chunk by 2σ_f²: the exponent the
            // contraction below is told, oa.sK)
            ka.res_bad = g->oz_badc.as<int>(); ka.res_n = g->oz_plan.n;
            ka.res_sK = oz_k_scale(g->p_out > 1 ? (pc > 1 ? 2.0 : 1.5) * g->prm.sigma_f2 : g->prm.sigma_f2);
            ka.res_ktg = oz_grad_exp(g);
            if (!kstore) ka.Kout = nullptr;
        }
        PHASE_EVENT(e[0], s);
        HIPCHK(launch_kgen(ka, s));
        PHASE_EVENT(e[1], s);
        if (oz) {
            OzVarArgs oa{};
            oa.plan = &g->oz_plan; oa.Kxz = kchunk; oa.ldk = ldk; oa.WR = g->oz_WR.as<int8_t>(); oa.sexp = g->oz_sexp.as<int>();
            oa.bad_row = g->oz_badr.as<int>(); oa.KR = g->oz_KR.as<int8_t>(); oa.U = g->oz_U.as<int8_t>();
            oa.bad_col = g->oz_badc.as<int>(); oa.ctr_clean = &g->oz_ctr_clean; oa.partial = g->partial.as<double>(); oa.ldp = Mc; oa.Np = (int)Np; oa.Mc = mcp;
            // a gradient-enhanced model's scaled chunk is bounded by σ_f²·√2 (oz_prepare_w), a StandardGP's by σ_f²
            // (derivative candidates against derivative training rows, both scaled: 2σ_f²)
            oa.nvalid = (int)g->N; oa.sK = oz_k_scale(g->p_out > 1 ? (pc > 1 ? 2.0 : 1.5) * g->prm.sigma_f2 : g->prm.sigma_f2);
            oa.kper = g->p_out; oa.ktg = oz_grad_exp(g);
            if (pc > 1) { oa.rmode = point_major ? 1 : 2; oa.rper = pc; oa.r0 = j0; oa.rpts = Mpts; }
            oa.ev_quant = phase_events() ? e[6] : nullptr; oa.ev_gemm = phase_events() ? e[7] : nullptr; oa.planes_ready = fused ? 1 : 0;
            PHASE_EVENT(e[2], s);
            HIPCHK(launch_var_ozaki(oa, s));
            PHASE_EVENT(e[3], s);
            g->tm.var_gemm_launches += 1;
        } else if (want_var) {
            VarGemmArgs va{};
            va.W = g->st->W.as<double>(); va.Kxz = kchunk; va.partial = g->partial.as<double>();
            va.ldw = g->st->cap; va.ldk = ldk; va.ldp = Mc; va.Np = (int)Np; va.Mc = mcp; va.nvalid = (int)g->N;
            PHASE_EVENT(e[2], s);
            HIPCHK(launch_var_gemm(va, s));
            PHASE_EVENT(e[3], s);
            g->tm.var_gemm_launches += 1;
        }
        FinalizeArgs fa{};
        fa.partial = g->partial.as<double>(); fa.mu_in = g->mu_c.as<double>(); fa.mu_out = mu_out;
        fa.var_out = var_out; fa.score_out = score_out; fa.ldp = Mc; fa.j0 = j0; fa.M = M;
        fa.T = (var_out || score_out) ? T : 0; fa.Mc = mcp; fa.kind = kind; fa.sigma_f2 = g->prm.sigma_f2;
        fa.p0 = p0; fa.best_y = best_y;
        fa.prior_grad = grad_prior_var(g); fa.pc = pc; fa.point_major = point_major; fa.Mpts = Mpts;
        PHASE_EVENT(e[4], s);
        HIPCHK(launch_finalize(fa, s));
        PHASE_EVENT(e[5], s);
    }
    return ABO_OK;
}

// Gradient-enhanced model: per-point posterior mean of all P outputs (mu_d [M][P]), P×P covariance block (cov_d [M][P][P]) and
// GradientNormUCB score (sc_d [M]) of M points in DEVICE memory (any output may be null) — queued on the handle's stream, no
// synchronisation.  The arithmetic behind abo_predict_grad_cov (GradientGP.jl:936-971, gradNormUCB.jl:43-51) and behind the
// refinement rounds of a gradient-enhanced handle (refine.hip: launch_refine_lockstep_grad).
int32_t grad_eval_device(abo_gp* g, const double* Zd, int64_t M, double beta, double* mu_d, double* cov_d, double* sc_d) {
    hipStream_t s = g->stream;
    const int P = g->p_out;
    const int64_t Np = g->Np;
    // points per chunk: V chunk (rows·Np doubles) ≤ 256 MiB, rows padded to 128
    int64_t pts = (((int64_t)1 << 28) / (Np * (int64_t)sizeof(double))) / P;
    if (pts < 1) pts = 1;
    if (pts > M) pts = M;
    const int64_t rows_pad = pad_up(pts * P, TB);
    // V = L⁻¹K_XZ on the int8-residue engine when the handle's contraction says so (same rule as the variance calls): the residue
    // GEMMs and a reconstruction that writes V itself; a chunk whose scratch the device cannot give runs on the fp64 GEMM
    int nm = 0;
    bool oz = wants_int8(g, true, P, &nm);
    if (oz) {
        int64_t mc = rows_pad;
        int32_t rc = oz_acquire(g, nm, rows_pad, &mc, &oz);
        if (rc) return rc;
        if (oz && mc != rows_pad) oz = false;
    }
    HIPCHK(g->events(EV_BASE));
    g->oz_prepare_pending = false;
    if (oz) { int32_t rc = oz_planes_of_w(g); if (rc) return rc; }
    g->tm.contraction_engine = oz ? ABO_CONTRACT_INT8 : ABO_CONTRACT_FP64;
    g->tm.oz_nmod = oz ? g->oz_plan.n : 0;
    // int8 engine: the generator writes the residue planes of the chunk itself; the fp64 chunk is then not materialised
    bool fused = false;
    if (oz) {
        KgenArgs probe{};
        probe.pt = P; probe.dp = g->dp;
        fused = kgen_writes_residues(probe, g->oz_plan.n);
    }
    if (!fused) HIPCHK(g->Kxz.ensure(sizeof(double) * rows_pad * Np));
    HIPCHK(g->partial.ensure(sizeof(double) * rows_pad * Np));       // V
    HIPCHK(g->mu_c.ensure(sizeof(double) * rows_pad));
    for (int64_t p0 = 0; p0 < M; p0 += pts) {
        const int64_t np = (M - p0) < pts ? (M - p0) : pts;
        const int rows = (int)pad_up(np * P, TB);
        KgenArgs ka{};
        ka.Xs = g->st->Xs.as<double>(); ka.Z = Zd; ka.alpha = g->alpha.as<double>(); ka.Kout = g->Kxz.as<double>();
        ka.mu = g->mu_c.as<double>(); ka.ldk = Np; ka.M = M; ka.j0 = p0 * P; ka.Mc = rows; ka.N = (int)g->npts;
        ka.Np = (int)Np; ka.d = g->d; ka.dp = g->dp; ka.family = g->prm.family; ka.s = 1.0 / g->prm.ell;
        ka.sigma_f2 = g->prm.sigma_f2; ka.pt = P; ka.pc = P; ka.point_major = 1;
        for (int q = 0; q < MAX_P; ++q) ka.mean_vec[q] = g->mean_vec[q];
        if (fused) {
            const int64_t q256 = pad_up(Np, 256);
            ka.Kout = nullptr;
            ka.res = g->oz_KR.as<int8_t>(); ka.res_ld = q256; ka.res_plane = pad_up(rows, 256) * q256;
            ka.res_bad = g->oz_badc.as<int>(); ka.res_n = g->oz_plan.n; ka.res_sK = oz_k_scale(2.0 * g->prm.sigma_f2);
            ka.res_ktg = oz_grad_exp(g);
        }
        HIPCHK(launch_kgen(ka, s));
        if (oz) {
            OzVarArgs oa{};
            oa.plan = &g->oz_plan; oa.Kxz = g->Kxz.as<double>(); oa.ldk = Np; oa.WR = g->oz_WR.as<int8_t>(); oa.sexp = g->oz_sexp.as<int>();
            oa.bad_row = g->oz_badr.as<int>(); oa.KR = g->oz_KR.as<int8_t>(); oa.U = g->oz_U.as<int8_t>();
            oa.bad_col = g->oz_badc.as<int>(); oa.ctr_clean = &g->oz_ctr_clean; oa.partial = nullptr; oa.ldp = 0; oa.Np = (int)Np; oa.Mc = rows;
            oa.nvalid = (int)g->N; oa.sK = oz_k_scale(2.0 * g->prm.sigma_f2);
            oa.kper = P; oa.ktg = oz_grad_exp(g); oa.planes_ready = fused ? 1 : 0;
            oa.rmode = 1; oa.rper = P; oa.r0 = p0 * P; oa.rpts = M;
            oa.Vout = g->partial.as<double>(); oa.ldv = Np;
            HIPCHK(launch_var_ozaki(oa, s));
        } else {
            GemmArgs a{};           // V[row][i] = Σ_k Kxz[row][k]·W[i][k]
            a.A = g->Kxz.as<double>(); a.lda = Np; a.B = g->st->W.as<double>(); a.ldb = g->st->cap;
            a.C = g->partial.as<double>(); a.ldc = Np; a.M = rows; a.N = (int)Np; a.K = (int)Np;
            a.kmode = K_B_LOWER; a.batch = 1; a.alpha = 1.0; a.beta = 0.0;       // W[i][k] = 0 for k > i: half the product
            HIPCHK(launch_gemm_nt(a, s));
        }
        GradCovArgs ca{};
        ca.V = g->partial.as<double>(); ca.mu_rows = g->mu_c.as<double>(); ca.ldv = Np; ca.R = (int)g->N; ca.p = P;
        ca.pt0 = p0; ca.prior0 = g->prm.sigma_f2; ca.prior_g = grad_prior_var(g); ca.beta = beta;
        ca.cov_out = cov_d; ca.mu_out = mu_d; ca.score_out = sc_d;
        HIPCHK(launch_grad_cov(ca, (int)np, s));
    }
    return ABO_OK;
}

void collect_posterior_timings(abo_gp* g, int64_t M, bool with_var) {
    const int64_t Mc = g->last_chunk;
    const int64_t nchunk = (M + Mc - 1) / Mc;
    double kx = 0, vg = 0, fi = 0, oq = 0, og = 0, oc = 0;
    const bool oz = with_var && g->tm.contraction_engine == ABO_CONTRACT_INT8;
    for (int64_t c = 0; phase_events() && c < nchunk; ++c) {
        hipEvent_t* e = &g->evs()[EV_BASE + EV_PER_CHUNK * c];
        kx += ev_ms(e[0], e[1]);
        if (with_var) vg += ev_ms(e[2], e[3]);
        if (oz) { oq += ev_ms(e[2], e[6]); og += ev_ms(e[6], e[7]); oc += ev_ms(e[7], e[3]); }
        fi += ev_ms(e[4], e[5]);
    }
    g->tm.acq_kxz_ms = kx;
    g->tm.acq_var_gemm_ms = vg;
    g->tm.acq_finalize_ms = fi;
    g->tm.oz_quant_ms = oq; g->tm.oz_gemm_ms = og; g->tm.oz_crt_ms = oc;
    if (g->oz_prepare_pending) { g->tm.oz_prepare_ms = phase_events() ? ev_ms(g->evs()[8], g->evs()[9]) : 0.0; g->oz_prepare_pending = false; }
    // ALGORITHMIC int8 operations of the residue GEMMs: n moduli × the triangular product N²·M (N(N+1)/2 multiply-adds per
    // candidate, 2 operations each ≈ N²).  What the kernel issues beyond that — the upper halves of its 256-wide diagonal blocks
    // (of which it skips 6 of 16 units), padding of N and M to 256 — is not credited.
    g->tm.oz_gemm_ops = oz ? (double)g->tm.oz_nmod * (double)g->N * (double)g->N * (double)M : 0.0;
    // algorithmic (triangular) flop of the contraction: N²·M, N = true training size
    g->tm.var_gemm_flop = with_var ? (double)g->N * (double)g->N * (double)M : 0.0;
}

int32_t check_fitted(abo_gp* g, int32_t d) {
    if (!g) return fail(ABO_EINVAL, "null handle");
    tl_phase_np = g->Np;                             // (every posterior / acquisition entry point passes here first)
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

// Full refit into a fresh Storage of capacity max(N, n_max) points.  X/y: caller buffers (host or device).
// Gradient-enhanced models (p_out > 1): y holds p_out values per point, by outputs at the ABI (y[q·N + i]) or — internal
// callers, y_point_major — already in the factor's point-major order (y[i·p + q]).
// defer: queue the fit's launches and return without waiting (no jitter retries then): the handle is provisionally marked
// fitted so that posterior launches can be queued behind it; the caller synchronises and calls fit_finish().
int32_t fit_impl(abo_gp* g, const double* X, int64_t N, int d, const double* y, int32_t space, int64_t* info,
                 int y_point_major = 0, bool defer = false) {
    hipStream_t s = g->stream;
    g->from_append = false;
    g->ap_x.clear();
    Storage* st = new (std::nothrow) Storage();
    if (!st) return fail(ABO_ENOMEM, "abo_fit: host allocation failed");
    st->set_device(g->prm.device);
    st->d = d; st->dp = dp_for(d);
    const int P = g->p_out;                              // N below = number of training POINTS
    const int64_t R = (int64_t)P * N;                    // factor rows
    const int64_t want = (int64_t)P * (g->prm.n_max > N ? g->prm.n_max : N);
    st->cap = pad_up(want, TB);
    const int64_t cap = st->cap;
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = st->Xraw.ensure(sizeof(double) * cap * d);
    if (e == hipSuccess) e = st->Xs.ensure(sizeof(double) * cap * st->dp);
    if (e == hipSuccess) e = st->ybuf.ensure(sizeof(double) * cap);
    if (e == hipSuccess) e = st->delta.ensure(sizeof(double) * cap);
    if (e == hipSuccess) e = st->K.ensure(sizeof(double) * cap * cap);
    if (e == hipSuccess) e = st->W.ensure(sizeof(double) * cap * cap);
    if (e == hipSuccess) e = st->WT.ensure(sizeof(double) * cap * cap);
    if (e != hipSuccess) {
        storage_unref(st);
        return fail(e == hipErrorOutOfMemory ? ABO_ENOMEM : ABO_EHIP, "abo_fit: device allocation failed: %s", hipGetErrorString(e));
    }
    const int64_t Np = pad_up(R, TB);
    {
        hipError_t eh = g->alpha.ensure(sizeof(double) * cap);      // the handle's own buffers, BEFORE the previous storage goes back to
        if (eh == hipSuccess) eh = g->tvec.ensure(sizeof(double) * cap);      // the pool: nothing handed out below may be memory the
        if (eh == hipSuccess) eh = g->T.ensure(sizeof(double) * Np * Np);     // caller's X / y still live in (refit after append)
        if (eh == hipSuccess) eh = g->info.ensure(sizeof(int64_t));
        if (eh == hipSuccess) eh = g->scal.ensure(sizeof(double) * 8);
        if (eh != hipSuccess) {
            storage_unref(st);
            return fail(eh == hipErrorOutOfMemory ? ABO_ENOMEM : ABO_EHIP, "abo_fit: device allocation failed: %s", hipGetErrorString(eh));
        }
    }
    // N ≤ 128, d ≤ 16, no spare capacity: the whole fit is one launch (chol.hip, mode 3 of the diagonal-block kernel)
    const bool fused_small = P == 1 && cap == TB && st->dp <= 16;
    // Inputs.  Device arrays of a StandardGP are read by the fit's first launch itself, which also writes the model's own copies — the
    // prep launch (larger models; issued right here), or the one-launch fit of a fresh handle — instead of two staging copies and three
    // tiny launches in front of the kernel matrix.  Everything else is staged as before.  All of it BEFORE the previous storage is
    // dropped: X / y may alias it (refit after append).
    g->in_x = g->in_y = nullptr;
    const bool direct = P == 1 && space == ABO_DEVICE && N > 0;
    int32_t rc = ABO_OK;
    if (direct && fused_small && g->st == nullptr) {        // (a handle that holds a storage stages: its launch runs behind the drop)
        g->in_x = X; g->in_y = y;
    } else if (!direct || fused_small) {
        rc = copy_in(st->Xraw.p, X, sizeof(double) * N * d, space, s);
        // (gradient-enhanced: staged in K — overwritten by the kernel matrix later — and reordered into ybuf below)
        if (!rc) rc = copy_in(P == 1 ? st->ybuf.p : st->K.p, y, sizeof(double) * R, space, s);
        if (rc) { storage_unref(st); return rc; }
    }
    if (P == 1 && !fused_small)
        HIPCHK(launch_fit_prep(direct ? X : st->Xraw.as<double>(), direct ? y : st->ybuf.as<double>(), st->Xraw.as<double>(),
                               st->ybuf.as<double>(), st->Xs.as<double>(), st->delta.as<double>(), g->alpha.as<double>(), (int)N, (int)cap,
                               d, st->dp, 1.0 / g->prm.ell, g->prm.mean_c, s));
    if (g->st) HIPCHK(wait_stream(s));         // a fresh handle (what update() makes) has nothing the inputs could alias
    if (g->st && g->fitted) g->st->drop_view(g->N);
    g->fitted = false;
    storage_unref(g->st);
    g->st = st;
    g->npts = N; g->N = R; g->d = d; g->dp = st->dp; g->Np = Np;
    tl_phase_np = Np;
    if (defer) {
        rc = fused_small ? fit_small(g, g->prm.noise_var, nullptr) : factorise(g, g->prm.noise_var, nullptr);
        if (rc) return rc;
        st->noise_used = g->prm.noise_var;
        st->add_view(R);
        g->fitted = true;                 // provisional until fit_finish()
        return ABO_OK;
    }
    if (fused_small) {
        int64_t inf = 0;
        double noise = g->prm.noise_var;
        for (int attempt = 0;; ++attempt) {
            rc = fit_small(g, noise, &inf);
            if (rc) return rc;
            if (inf == 0) break;
            if (!(g->prm.jitter > 0.0) || attempt >= 4) {
                if (info) *info = inf;
                return fail(ABO_ENOTPD, "PosDefException: matrix is not positive definite; Cholesky factorization failed at %lld",
                            (long long)inf);
            }
            noise = g->prm.noise_var + g->prm.jitter * std::pow(10.0, attempt);
        }
        st->noise_used = noise;
        st->add_view(R);
        g->fitted = true;
        return ABO_OK;
    }
    if (P > 1) {                                         // point-major rows i·p + q, output q centred by mean_vec[q]
        HIPCHK(launch_scale_points(st->Xraw.as<double>(), st->Xs.as<double>(), (int)N, (int)cap, d, st->dp, 1.0 / g->prm.ell, s));
        MeanVec mv{};
        for (int q = 0; q < P; ++q) mv.c[q] = g->mean_vec[q];
        HIPCHK(launch_center_grad(st->K.as<double>(), st->ybuf.as<double>(), st->delta.as<double>(), (int)N, P, (int)cap, mv,
                                  y_point_major, s));
        HIPCHK(hipMemsetAsync(g->alpha.p, 0, sizeof(double) * cap, s));
    }

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
    st->noise_used = noise;
    st->add_view(R);
    g->fitted = true;
    return ABO_OK;
}

// Bordered ("rank-1 append") update: view `g` (N points) → new view `n` (N+1 points) sharing g's storage.
//   k = k(X, x*),  l = W·k,  l_nn² = k** + noise − ‖l‖²,  v = Wᵀ·l = K⁻¹k
//   L[N] = [lᵀ, l_nn]      W[N] = [−vᵀ/l_nn, 1/l_nn]      β = (δ* − kᵀα)/l_nn²      α' = [α − βv ; β]
// Falls back to a full refit (new storage, doubled capacity) when the capacity is used up or when
// another view already appended to the shared storage (copy-on-write for diverging histories).
int32_t append_impl(abo_gp* g, abo_gp* n, const double* x, double y, int64_t* info) {
    Storage* st = g->st;
    const int d = g->d;
    const int64_t N = g->N;
    hipStream_t s = n->stream;
    if (N + 1 > st->cap || !st->claim_rows(N, N + 1)) {
        // gather this view's data on the device and refit with room to grow
        ScratchBuf xs(g->prm.device, s), ys(g->prm.device, s);
        DevBuf& xb = xs.b;
        DevBuf& yb = ys.b;
        HIPCHK(xb.ensure(sizeof(double) * (N + 1) * d));
        HIPCHK(yb.ensure(sizeof(double) * (N + 1)));
        HIPCHK(hipMemcpyAsync(xb.p, st->Xraw.p, sizeof(double) * N * d, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(yb.p, st->ybuf.p, sizeof(double) * N, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(xb.as<double>() + N * d, x, sizeof(double) * d, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(yb.as<double>() + N, &y, sizeof(double), hipMemcpyHostToDevice, s));
        HIPCHK(wait_stream(s));
        if (n->prm.n_max < 2 * (N + 1)) n->prm.n_max = 2 * (N + 1);
        return fit_impl(n, xb.as<double>(), N + 1, d, yb.as<double>(), ABO_DEVICE, info);
    }
    // rows [N, N + 1) are claimed: every failure below gives them back
    struct Claim { Storage* st; int64_t n; bool keep = false; ~Claim() { if (!keep) st->drop_view(n); } } claim{st, N + 1};
    const int64_t ld = st->cap;
    const int64_t Np = g->Np;                       // padded size of the OLD view
    const int64_t Np1 = pad_up(N + 1, TB);
    HIPCHK(n->alpha.ensure(sizeof(double) * ld));
    HIPCHK(n->vext.ensure(sizeof(double) * ld));
    HIPCHK(n->tvec.ensure(sizeof(double) * ld * 2));
    HIPCHK(n->Kxz.ensure(sizeof(double) * 16 * Np));
    HIPCHK(n->info.ensure(sizeof(int64_t)));
    HIPCHK(n->scal.ensure(sizeof(double) * 8));
    double* Xraw = st->Xraw.as<double>();
    // new point: raw coordinates, scaled coordinates, centred target (rows N — beyond every older view)
    if (d <= APPEND_POINT_MAXD && st->dp <= APPEND_POINT_MAXD) {
        HIPCHK(launch_append_point(x, d, st->dp, 1.0 / g->prm.ell, y, g->prm.mean_c, Xraw + N * d, st->Xs.as<double>() + N * st->dp,
                                   st->ybuf.as<double>() + N, st->delta.as<double>() + N, n->info.as<int64_t>(), s));
    } else {
        HIPCHK(hipMemcpyAsync(Xraw + N * d, x, sizeof(double) * d, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(st->ybuf.as<double>() + N, &y, sizeof(double), hipMemcpyHostToDevice, s));
        HIPCHK(launch_scale_points(Xraw + N * d, st->Xs.as<double>() + N * st->dp, 1, 1, d, st->dp, 1.0 / g->prm.ell, s));
        HIPCHK(launch_center(st->ybuf.as<double>() + N, st->delta.as<double>() + N, 1, 1, g->prm.mean_c, s));
        HIPCHK(hipMemsetAsync(n->info.p, 0, sizeof(int64_t), s));
    }
    // k = k(X, x*): one kernel evaluation per training point (the training points play the candidates of the
    // column kernel; same scaled differences and the same kappa as kgen, so the values are bit-identical to a refit's)
    HIPCHK(launch_cand_newcol(st->Xs.as<double>() + N * st->dp, Xraw, n->Kxz.as<double>(), 1, N, 0, d, st->dp, g->prm.family,
                              1.0 / g->prm.ell, g->prm.sigma_f2, s));
    double* krow = n->Kxz.as<double>();
    double* lvec = n->tvec.as<double>();
    double* vvec = n->tvec.as<double>() + ld;
    // only rows/columns < N: anything beyond may be the stale remains of a discarded branch
    tl_phase_np = Np1;
    const bool timed = phase_events();
    if (timed) { HIPCHK(n->events(EV_BASE + 2)); HIPCHK(hipEventRecord(n->evs()[EV_BASE], s)); }
    HIPCHK(launch_trmv(st->W.as<double>(), ld, krow, lvec, (int)N, 1, s));
    HIPCHK(launch_trmv(st->WT.as<double>(), ld, lvec, vvec, (int)N, 0, s));
    if (timed) HIPCHK(hipEventRecord(n->evs()[EV_BASE + 1], s));
    AppendArgs aa{};
    aa.L = st->K.as<double>(); aa.W = st->W.as<double>(); aa.WT = st->WT.as<double>(); aa.ld = ld;
    aa.krow = krow; aa.lvec = lvec; aa.vvec = vvec; aa.alpha_old = g->alpha.as<double>(); aa.alpha_new = n->alpha.as<double>();
    aa.vext = n->vext.as<double>(); aa.delta = st->delta.as<double>(); aa.N = (int)N; aa.cap = (int)ld;
    aa.kss = g->prm.sigma_f2 + st->noise_used; aa.scal = n->scal.as<double>(); aa.info = n->info.as<int64_t>();
    HIPCHK(launch_append(aa, s));
    double sc[4];
    int64_t inf = 0;
    PinStage pin(n->ctx);                            // the NEW handle's staging block: `g` may be in use by another thread (a copy of it)
    HIPCHK(pin.d2h(sc, n->scal.p, sizeof sc, s));
    HIPCHK(pin.d2h(&inf, n->info.p, sizeof inf, s));
    HIPCHK(wait_stream(s));
    pin.flush();
    if (inf != 0) {
        if (info) *info = inf;
        return fail(ABO_ENOTPD, "PosDefException: matrix is not positive definite; Cholesky factorization failed at %lld",
                    (long long)inf);
    }
    storage_unref(n->st);
    st->refs.fetch_add(1);
    n->st = st;
    claim.keep = true;                               // (registered by claim_rows)
    n->N = N + 1; n->npts = N + 1; n->Np = Np1; n->d = d; n->dp = st->dp;
    n->from_append = true;
    // the two triangular mat-vecs l = L⁻¹k, v = L⁻ᵀl: each streams one triangle of its matrix once (8·N²/2 bytes)
    n->tm.append_trmv_ms = timed ? ev_ms(n->evs()[EV_BASE], n->evs()[EV_BASE + 1]) : 0.0;
    n->tm.append_trmv_bytes = 8.0 * (double)N * (double)N;
    n->ap_s2 = sc[0]; n->ap_beta = sc[1];
    n->ap_x.assign(x, x + d);
    n->logdet = g->logdet + 2.0 * std::log(sc[2]);
    n->quad = g->quad + sc[1] * sc[1] * sc[0];
    n->fitted = true;
    return ABO_OK;
}

// Gradient-enhanced model: one more observation (f and its gradient at x) = p_out more rows at the END of the point-major
// factor, appended one row at a time with the bordered update above; row (N, q) sees the rows (N, q' < q) appended just
// before it.  The kernel row comes from the multi-output generator (analytic derivative blocks), everything else is the
// single-row machinery.  yv: p_out host values {f, ∂f/∂x_1 …}.
int32_t append_grad_impl(abo_gp* g, abo_gp* n, const double* x, const double* yv, int64_t* info) {
    Storage* st = g->st;
    const int d = g->d, P = g->p_out;
    const int64_t R = g->N, npts = g->npts;
    hipStream_t s = n->stream;
    if (R + P > st->cap || !st->claim_rows(R, R + P)) {
        ScratchBuf xs(g->prm.device, s), ys(g->prm.device, s);
        DevBuf& xb = xs.b;
        DevBuf& yb = ys.b;
        HIPCHK(xb.ensure(sizeof(double) * (npts + 1) * d));
        HIPCHK(yb.ensure(sizeof(double) * (R + P)));
        HIPCHK(hipMemcpyAsync(xb.p, st->Xraw.p, sizeof(double) * npts * d, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(yb.p, st->ybuf.p, sizeof(double) * R, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(xb.as<double>() + npts * d, x, sizeof(double) * d, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(yb.as<double>() + R, yv, sizeof(double) * P, hipMemcpyHostToDevice, s));
        HIPCHK(wait_stream(s));
        if (n->prm.n_max < 2 * (npts + 1)) n->prm.n_max = 2 * (npts + 1);
        return fit_impl(n, xb.as<double>(), npts + 1, d, yb.as<double>(), ABO_DEVICE, info, /*y_point_major=*/1);
    }
    struct Claim { Storage* st; int64_t n; bool keep = false; ~Claim() { if (!keep) st->drop_view(n); } } claim{st, R + P};
    const int64_t ld = st->cap;
    HIPCHK(n->alpha.ensure(sizeof(double) * ld));
    HIPCHK(n->T.ensure(sizeof(double) * ld));              // second alpha buffer (the rows alternate between the two)
    HIPCHK(n->vext.ensure(sizeof(double) * ld * P));
    HIPCHK(n->tvec.ensure(sizeof(double) * ld * 2));
    HIPCHK(n->Kxz.ensure(sizeof(double) * 16 * ld));
    HIPCHK(n->info.ensure(sizeof(int64_t)));
    HIPCHK(n->scal.ensure(sizeof(double) * 4 * MAX_P));
    double* Xraw = st->Xraw.as<double>();
    HIPCHK(hipMemcpyAsync(Xraw + npts * d, x, sizeof(double) * d, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(st->ybuf.as<double>() + R, yv, sizeof(double) * P, hipMemcpyHostToDevice, s));
    HIPCHK(launch_scale_points(Xraw + npts * d, st->Xs.as<double>() + npts * st->dp, 1, 1, d, st->dp, 1.0 / g->prm.ell, s));
    for (int q = 0; q < P; ++q)
        HIPCHK(launch_center(st->ybuf.as<double>() + R + q, st->delta.as<double>() + R + q, 1, 1, g->mean_vec[q], s));
    HIPCHK(hipMemsetAsync(n->info.p, 0, sizeof(int64_t), s));
    double* krow = n->Kxz.as<double>();
    double* lvec = n->tvec.as<double>();
    double* vvec = n->tvec.as<double>() + ld;
    const double* alpha_cur = g->alpha.as<double>();
    for (int q = 0; q < P; ++q) {
        const int64_t Rq = R + q;                          // rows in front of the one being appended
        KgenArgs ka{};
        ka.Xs = st->Xs.as<double>(); ka.Z = Xraw + npts * d; ka.alpha = nullptr; ka.Kout = krow; ka.mu = nullptr;
        ka.ldk = ld; ka.M = 1; ka.j0 = q; ka.Mc = 16; ka.N = (int)(npts + 1); ka.Np = (int)pad_up(Rq, TB); ka.d = d; ka.dp = st->dp;
        ka.family = g->prm.family; ka.s = 1.0 / g->prm.ell; ka.sigma_f2 = g->prm.sigma_f2; ka.mean_c = 0.0;
        ka.pt = P; ka.pc = P; ka.point_major = 1; ka.rvalid = (int)Rq;
        HIPCHK(launch_kgen(ka, s));
        HIPCHK(launch_trmv(st->W.as<double>(), ld, krow, lvec, (int)Rq, 1, s));
        HIPCHK(launch_trmv(st->WT.as<double>(), ld, lvec, vvec, (int)Rq, 0, s));
        double* alpha_new = (q & 1) == ((P - 1) & 1) ? n->alpha.as<double>() : n->T.as<double>();   // the last row lands in n->alpha
        AppendArgs aa{};
        aa.L = st->K.as<double>(); aa.W = st->W.as<double>(); aa.WT = st->WT.as<double>(); aa.ld = ld;
        aa.krow = krow; aa.lvec = lvec; aa.vvec = vvec; aa.alpha_old = alpha_cur; aa.alpha_new = alpha_new;
        aa.vext = n->vext.as<double>() + (int64_t)q * ld; aa.delta = st->delta.as<double>(); aa.N = (int)Rq; aa.cap = (int)ld;
        aa.kss = (q == 0 ? g->prm.sigma_f2 : grad_prior_var(g)) + st->noise_used;
        aa.scal = n->scal.as<double>() + 4 * q; aa.info = n->info.as<int64_t>();
        HIPCHK(launch_append(aa, s));
        alpha_cur = alpha_new;
    }
    double sc[4 * MAX_P];
    int64_t inf = 0;
    PinStage pin(n->ctx);
    HIPCHK(pin.d2h(sc, n->scal.p, sizeof(double) * 4 * P, s));
    HIPCHK(pin.d2h(&inf, n->info.p, sizeof inf, s));
    HIPCHK(wait_stream(s));
    pin.flush();
    if (inf != 0) {
        if (info) *info = inf;
        return fail(ABO_ENOTPD, "PosDefException: matrix is not positive definite; Cholesky factorization failed at %lld",
                    (long long)inf);
    }
    storage_unref(n->st);
    st->refs.fetch_add(1);
    n->st = st;
    claim.keep = true;
    n->N = R + P; n->npts = npts + 1; n->Np = pad_up(R + P, TB); n->d = d; n->dp = st->dp;
    n->from_append = true;
    n->logdet = g->logdet; n->quad = g->quad;
    for (int q = 0; q < P; ++q) {
        n->ap_s2v[q] = sc[4 * q]; n->ap_betav[q] = sc[4 * q + 1];
        n->logdet += 2.0 * std::log(sc[4 * q + 2]);
        n->quad += sc[4 * q + 1] * sc[4 * q + 1] * sc[4 * q];
    }
    n->fitted = true;
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
    abo::arm_exit_guard();
    abo_gp* g = new (std::nothrow) abo_gp();
    if (!g) return fail(ABO_ENOMEM, "abo_create: host allocation failed");
    g->prm = *params;
    g->mean_vec[0] = params->mean_c;
    g->set_device(params->device);
    oz_defaults(&g->oz_engine, &g->oz_nmod);
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
        void* pin = nullptr;                               // optional: without it the small read-backs go to pageable memory as before
        static const bool no_pin = getenv("ABO_NO_PINNED") != nullptr;         // A/B and the test of the fallback
        if (!no_pin && hipHostMalloc(&pin, PIN_BYTES, hipHostMallocDefault) == hipSuccess) g->ctx->pin = static_cast<char*>(pin);
        else (void)hipGetLastError();
    }
    g->stream = g->ctx->stream;
    e = g->events(8);
    if (e != hipSuccess) { g->free_all(); delete g; return fail(ABO_EHIP, "hipEventCreate: %s", hipGetErrorString(e)); }
    *out = g;
    return ABO_OK;
}

int32_t abo_create_grad(const abo_params* params, int32_t p, const double* mean_c, abo_gp** out) {
    if (!params || !out) return fail(ABO_EINVAL, "abo_create_grad: null argument");
    if (p < 2 || p > MAX_P)
        return fail(ABO_EINVAL, "abo_create_grad: p = %d outputs outside 2..%d (library limit: a gradient-enhanced GP has "
                                "d = p − 1 ≤ %d inputs)", p, MAX_P, MAX_P - 1);
    if (params->family != ABO_KERNEL_SE && params->family != ABO_KERNEL_MATERN52 && params->family != ABO_KERNEL_MATERN72)
        return fail(ABO_EINVAL, "abo_create_grad: the gradient-enhanced GP needs a twice-differentiable kernel "
                                "(SE, Matern-5/2, Matern-7/2)");
    int32_t rc = abo_create(params, out);
    if (rc) return rc;
    (*out)->p_out = p;
    for (int q = 0; q < p; ++q) {
        const double c = mean_c ? mean_c[q] : 0.0;
        if (!std::isfinite(c)) { abo_destroy(*out); *out = nullptr; return fail(ABO_EINVAL, "abo_create_grad: non-finite mean"); }
        (*out)->mean_vec[q] = c;
    }
    return ABO_OK;
}

int32_t abo_set_contraction(abo_gp* gp, int32_t engine, int32_t nmod) {
    if (engine < ABO_CONTRACT_AUTO || engine > ABO_CONTRACT_INT8) return fail(ABO_EINVAL, "abo_set_contraction: unknown engine %d", engine);
    if (nmod != 0 && (nmod < 8 || nmod > OZ_MAXMOD)) return fail(ABO_EINVAL, "abo_set_contraction: 8 to %d moduli (0 = default)", OZ_MAXMOD);
    if (!gp) {
        int e, n;
        oz_defaults(&e, &n);
        g_oz_nmod.store(nmod ? nmod : OZ_DEFAULT_NMOD);
        g_oz_engine.store(engine);
        return ABO_OK;
    }
    gp->oz_engine = engine;
    gp->oz_nmod = nmod;                                            // 0: whatever the process default is when the posterior runs
    return ABO_OK;
}

int32_t abo_retain(abo_gp* gp) {
    if (!gp) return fail(ABO_EINVAL, "null handle");
    gp->refs.fetch_add(1);
    return ABO_OK;
}

int32_t abo_destroy(abo_gp* gp) {
    if (!gp) return ABO_OK;
    // a finaliser that fires while the process exits (or after the runtime reported itself gone): the handle's device state
    // is reclaimed with the process — nothing here may call into HIP any more, and nothing is worth freeing
    if (g_exiting.load()) return ABO_OK;
    if (gp->refs.fetch_sub(1) == 1) {
        hipError_t e = hipSetDevice(gp->prm.device);
        if (e == hipSuccess && gp->stream) e = wait_stream(gp->stream);
        if (abo::gone(e)) { g_exiting.store(true); return ABO_OK; }      // the runtime says it has been torn down: the process is exiting
        // any other error of these two calls concerns this handle only: its buffers still go back to the pool, nothing is latched
        (void)hipGetLastError();
        gp->free_all();
        delete gp;
    }
    return ABO_OK;
}

int32_t abo_fit(abo_gp* g, const double* X, int64_t N, int32_t d, const double* y, int32_t space, int64_t* info) {
    if (info) *info = 0;
    if (!g || !X || !y) return fail(ABO_EINVAL, "abo_fit: null argument");
    if (N < 1) return fail(ABO_EINVAL, "abo_fit: need at least one training point");
    if (d < 1 || d > 65536) return fail(ABO_EINVAL, "abo_fit: input dimension %d outside 1..65536", d);
    if (N > (int64_t)1 << 20) return fail(ABO_EINVAL, "abo_fit: N = %lld too large", (long long)N);
    if (g->p_out > 1 && d + 1 != g->p_out)
        return fail(ABO_EDIM, "DimensionMismatch: gradient-enhanced model with p = %d outputs needs d = %d inputs, got %d",
                    g->p_out, g->p_out - 1, d);
    HIPCHK(hipSetDevice(g->prm.device));
    return fit_impl(g, X, N, d, y, space, info);
}

int32_t abo_append(abo_gp* g, const double* x, int32_t d, double y, int64_t* info, abo_gp** out) {
    if (info) *info = 0;
    if (!g || !x || !out) return fail(ABO_EINVAL, "abo_append: null argument");
    int32_t rc = check_fitted(g, d);
    if (rc) return rc;
    if (g->p_out > 1) return fail(ABO_EINVAL, "abo_append: a gradient-enhanced model takes p values per observation (abo_append_grad)");
    HIPCHK(hipSetDevice(g->prm.device));
    abo_gp* n = nullptr;
    rc = abo_create(&g->prm, &n);
    if (rc) return rc;
    n->oz_engine = g->oz_engine; n->oz_nmod = g->oz_nmod;
    rc = append_impl(g, n, x, y, info);
    if (rc) { abo_destroy(n); return rc; }
    *out = n;
    return ABO_OK;
}

int32_t abo_append_grad(abo_gp* g, const double* x, int32_t d, const double* y, int64_t* info, abo_gp** out) {
    if (info) *info = 0;
    if (!g || !x || !y || !out) return fail(ABO_EINVAL, "abo_append_grad: null argument");
    int32_t rc = check_fitted(g, d);
    if (rc) return rc;
    if (g->p_out < 2) return fail(ABO_EINVAL, "abo_append_grad: the model has no gradient outputs (abo_append)");
    HIPCHK(hipSetDevice(g->prm.device));
    abo_gp* n = nullptr;
    rc = abo_create(&g->prm, &n);
    if (rc) return rc;
    n->p_out = g->p_out;
    n->oz_engine = g->oz_engine; n->oz_nmod = g->oz_nmod;
    for (int q = 0; q < MAX_P; ++q) n->mean_vec[q] = g->mean_vec[q];
    rc = append_grad_impl(g, n, x, y, info);
    if (rc) { abo_destroy(n); return rc; }
    *out = n;
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
    HIPCHK(wait_stream(s));
    collect_posterior_timings(g, M, var != nullptr);
    g->tm.acq_topk_ms = 0.0;
    g->tm.acq_total_ms = ev_ms(g->evs()[5], g->evs()[6]);
    return ABO_OK;
}

int32_t abo_predict_grad(abo_gp* g, const double* Z, int64_t M, int32_t d, int32_t z_space, double* mu, double* var,
                         int32_t out_space) {
    int32_t rc = check_fitted(g, d);
    if (rc) return rc;
    if (g->p_out < 2) return fail(ABO_EINVAL, "abo_predict_grad: the model has no gradient outputs");
    if (M < 0 || (M > 0 && !Z)) return fail(ABO_EINVAL, "abo_predict_grad: bad candidate buffer");
    if (M == 0) return ABO_OK;
    HIPCHK(hipSetDevice(g->prm.device));
    hipStream_t s = g->stream;
    const int P = g->p_out;
    const double* Zd = nullptr;
    rc = stage_candidates(g, Z, M, z_space, &Zd);
    if (rc) return rc;
    double* mu_d = nullptr;
    double* var_d = nullptr;
    if (mu) { if (out_space == ABO_DEVICE) mu_d = mu; else { HIPCHK(g->mu_all.ensure(sizeof(double) * M * P)); mu_d = g->mu_all.as<double>(); } }
    if (var) { if (out_space == ABO_DEVICE) var_d = var; else { HIPCHK(g->var_all.ensure(sizeof(double) * M * P)); var_d = g->var_all.as<double>(); } }
    rc = posterior(g, Zd, M, -1, 0.0, 0.0, mu_d, var_d, nullptr, P, 0);
    if (rc) return rc;
    if (mu && out_space == ABO_HOST) { rc = copy_out(mu, mu_d, sizeof(double) * M * P, ABO_HOST, s); if (rc) return rc; }
    if (var && out_space == ABO_HOST) { rc = copy_out(var, var_d, sizeof(double) * M * P, ABO_HOST, s); if (rc) return rc; }
    HIPCHK(wait_stream(s));
    return ABO_OK;
}

int32_t abo_predict_grad_cov(abo_gp* g, const double* Z, int64_t M, int32_t d, int32_t z_space, double beta, double* mu,
                             double* cov, double* score, int32_t out_space) {
    int32_t rc = check_fitted(g, d);
    if (rc) return rc;
    if (g->p_out < 2) return fail(ABO_EINVAL, "abo_predict_grad_cov: the model has no gradient outputs");
    if (M < 0 || (M > 0 && !Z)) return fail(ABO_EINVAL, "abo_predict_grad_cov: bad candidate buffer");
    if (M == 0) return ABO_OK;
    HIPCHK(hipSetDevice(g->prm.device));
    hipStream_t s = g->stream;
    const int P = g->p_out;
    const double* Zd = nullptr;
    rc = stage_candidates(g, Z, M, z_space, &Zd);
    if (rc) return rc;
    double* mu_d = nullptr; double* cov_d = nullptr; double* sc_d = nullptr;
    if (mu) { if (out_space == ABO_DEVICE) mu_d = mu; else { HIPCHK(g->mu_all.ensure(sizeof(double) * M * P)); mu_d = g->mu_all.as<double>(); } }
    if (cov) { if (out_space == ABO_DEVICE) cov_d = cov; else { HIPCHK(g->var_all.ensure(sizeof(double) * M * P * P)); cov_d = g->var_all.as<double>(); } }
    if (score) { if (out_space == ABO_DEVICE) sc_d = score; else { HIPCHK(g->score_all.ensure(sizeof(double) * M)); sc_d = g->score_all.as<double>(); } }
    rc = grad_eval_device(g, Zd, M, beta, mu_d, cov_d, sc_d);
    if (rc) return rc;
    if (mu && out_space == ABO_HOST) { rc = copy_out(mu, mu_d, sizeof(double) * M * P, ABO_HOST, s); if (rc) return rc; }
    if (cov && out_space == ABO_HOST) { rc = copy_out(cov, cov_d, sizeof(double) * M * P * P, ABO_HOST, s); if (rc) return rc; }
    if (score && out_space == ABO_HOST) { rc = copy_out(score, sc_d, sizeof(double) * M, ABO_HOST, s); if (rc) return rc; }
    HIPCHK(wait_stream(s));
    return ABO_OK;
}

int32_t abo_acq(abo_gp* g, const double* Z, int64_t M, int32_t d, int32_t z_space, int32_t kind, double p0,
                double best_y, int64_t idx_base, double* scores, int32_t k, double* top_val, int64_t* top_idx,
                int32_t out_space) {
    return abo::acq_ex(g, Z, M, d, z_space, kind, p0, best_y, idx_base, scores, out_space, k, top_val, top_idx, out_space);
}

}  // extern "C"

namespace {

// C-ABI terms → the kernels' form; validates kinds against the handle (GRADNORM_UCB needs gradient outputs)
int32_t make_terms(const abo_gp* g, const abo_acq_term* terms, int32_t n, AcqTerms* out, const char* fn) {
    if (!terms || n < 1) return fail(ABO_EINVAL, "%s: an objective needs at least one term", fn);
    if (n > MAX_TERMS) return fail(ABO_EINVAL, "%s: %d terms, the library takes at most %d", fn, n, MAX_TERMS);
    out->n = n;
    for (int i = 0; i < n; ++i) {
        const int k = terms[i].kind;
        if (k < ABO_ACQ_EI || k > ABO_ACQ_GRADNORM_UCB) return fail(ABO_EINVAL, "%s: unknown acquisition kind %d (term %d)", fn, k, i);
        if (k == ABO_ACQ_GRADNORM_UCB && (!g || g->p_out < 2))
            return fail(ABO_EINVAL, "%s: GradientNormUCB (term %d) needs a gradient-enhanced model", fn, i);
        if (!(terms[i].weight == terms[i].weight)) return fail(ABO_EINVAL, "%s: weight of term %d is not a number", fn, i);
        out->kind[i] = k; out->p0[i] = terms[i].p0; out->best_y[i] = terms[i].best_y; out->w[i] = terms[i].weight;
    }
    return ABO_OK;
}

AcqTerms one_term(int32_t kind, double p0, double best_y) {
    AcqTerms t{};
    t.n = 1; t.kind[0] = kind; t.p0[0] = p0; t.best_y[0] = best_y; t.w[0] = 1.0;
    return t;
}

// scores of M device-resident candidates under the objective `t` into sc_d (device), queued on the handle's stream:
//   one plain term            the fused posterior + epilogue pass (what abo_acq has always run: same bits)
//   function-value terms      ONE posterior pass, then every member's epilogue on that μ, σ² (EnsembleAcq.jl:53-55)
//   a GRADNORM_UCB term       the all-output posterior per point (mean[p], covariance block) in slabs, epilogue on those
int32_t score_terms_device(abo_gp* g, const double* Zd, int64_t M, const AcqTerms& t, double* sc_d) {
    hipStream_t s = g->stream;
    if (terms_plain(t)) {
        if (t.kind[0] != ABO_ACQ_MEAN) return posterior(g, Zd, M, t.kind[0], t.p0[0], t.best_y[0], nullptr, nullptr, sc_d);
        // −mu only: skip the contraction (scores come from the mean pass)
        HIPCHK(g->mu_all.ensure(sizeof(double) * M));
        int32_t rc = posterior(g, Zd, M, -1, 0.0, 0.0, g->mu_all.as<double>(), nullptr, nullptr);
        if (rc) return rc;
        FinalizeArgs fa{};
        fa.partial = nullptr; fa.mu_in = g->mu_all.as<double>(); fa.score_out = sc_d; fa.ldp = 0; fa.j0 = 0; fa.M = M;
        fa.T = 0; fa.Mc = (int)M; fa.kind = ABO_ACQ_MEAN; fa.sigma_f2 = g->prm.sigma_f2;
        HIPCHK(launch_finalize(fa, s));
        return ABO_OK;
    }
    if (!terms_have_gradnorm(t)) {
        HIPCHK(g->mu_all.ensure(sizeof(double) * M));
        HIPCHK(g->var_all.ensure(sizeof(double) * M));
        int32_t rc = posterior(g, Zd, M, -1, 0.0, 0.0, g->mu_all.as<double>(), g->var_all.as<double>(), nullptr);
        if (rc) return rc;
        HIPCHK(launch_score_terms(g->mu_all.as<double>(), g->var_all.as<double>(), sc_d, M, t, s));
        return ABO_OK;
    }
    const int P = g->p_out;
    int64_t slab = ((int64_t)64 << 20) / ((int64_t)sizeof(double) * P * P);       // ≤ 64 MiB of covariance blocks at a time
    if (slab < 1) slab = 1;
    if (slab > M) slab = M;
    HIPCHK(g->mu_all.ensure(sizeof(double) * slab * P));
    HIPCHK(g->var_all.ensure(sizeof(double) * slab * P * P));
    for (int64_t j0 = 0; j0 < M; j0 += slab) {
        const int64_t np = (M - j0) < slab ? (M - j0) : slab;
        int32_t rc = grad_eval_device(g, Zd + j0 * g->d, np, 0.0, g->mu_all.as<double>(), g->var_all.as<double>(), nullptr);
        if (rc) return rc;
        HIPCHK(launch_score_terms_grad(g->mu_all.as<double>(), g->var_all.as<double>(), sc_d + j0, np, P, t, s));
    }
    return ABO_OK;
}

// scores + selection for host or device candidates; separate memory spaces for the M scores and for the k selected pairs
int32_t acq_terms_impl(abo_gp* g, const double* Z, int64_t M, int32_t d, int32_t z_space, const AcqTerms& t, int64_t idx_base,
                       double* scores, int32_t out_space, int32_t k, double* top_val, int64_t* top_idx, int32_t top_space) {
    if (M < 0 || (M > 0 && !Z)) return fail(ABO_EINVAL, "abo_acq: bad candidate buffer");
    if (k < 0) return fail(ABO_EINVAL, "abo_acq: k = %d is negative", k);
    if (k > 0 && (!top_val || !top_idx)) return fail(ABO_EINVAL, "abo_acq: k > 0 needs top_val and top_idx");
    HIPCHK(hipSetDevice(g->prm.device));
    hipStream_t s = g->stream;
    double* sc_d = nullptr;
    PinStage pin(g->ctx);
    const bool with_var = !(terms_plain(t) && t.kind[0] == ABO_ACQ_MEAN);
    const bool fused_timings = !terms_have_gradnorm(t);           // the slab path does not keep per-chunk events
    if (M > 0) {
        const double* Zd = nullptr;
        int32_t rc = stage_candidates(g, Z, M, z_space, &Zd);
        if (rc) return rc;
        if (scores && out_space == ABO_DEVICE) sc_d = scores;
        else { HIPCHK(g->score_all.ensure(sizeof(double) * M)); sc_d = g->score_all.as<double>(); }
        HIPCHK(hipEventRecord(g->evs()[5], s));
        rc = score_terms_device(g, Zd, M, t, sc_d);
        if (rc) return rc;
        PHASE_EVENT(g->evs()[6], s);
    } else {
        HIPCHK(hipEventRecord(g->evs()[5], s));
        PHASE_EVENT(g->evs()[6], s);
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
        if (top_space == ABO_HOST) {
            HIPCHK(g->top_val.ensure(sizeof(double) * k));
            HIPCHK(g->top_idx.ensure(sizeof(int64_t) * k));
            tv = g->top_val.as<double>();
            ti = g->top_idx.as<int64_t>();
        }
        HIPCHK(launch_topk(sc_d, M, k, idx_base, w, tv, ti, s));
        if (top_space == ABO_HOST) {
            HIPCHK(pin.d2h(top_val, tv, sizeof(double) * k, s));             // through the pinned block, copied on after the sync
            HIPCHK(pin.d2h(top_idx, ti, sizeof(int64_t) * k, s));
        }
    }
    HIPCHK(hipEventRecord(g->evs()[7], s));
    if (scores && out_space == ABO_HOST && M > 0) {
        int32_t rc = copy_out(scores, sc_d, sizeof(double) * M, ABO_HOST, s);
        if (rc) return rc;
    }
    HIPCHK(wait_stream(s));
    pin.flush();
    if (M > 0 && fused_timings) collect_posterior_timings(g, M, with_var);
    g->tm.acq_topk_ms = phase_events() ? ev_ms(g->evs()[6], g->evs()[7]) : 0.0;
    g->tm.acq_total_ms = ev_ms(g->evs()[5], g->evs()[7]);
    return ABO_OK;
}

}  // namespace

// abo_acq with separate memory spaces for the M scores and for the k selected pairs (the multi-device driver keeps the
// pairs on the device for the RCCL exchange while the scores, when asked for, go to the caller's host array)
int32_t abo::acq_ex(abo_gp* g, const double* Z, int64_t M, int32_t d, int32_t z_space, int32_t kind, double p0,
                    double best_y, int64_t idx_base, double* scores, int32_t out_space, int32_t k, double* top_val,
                    int64_t* top_idx, int32_t top_space) {
    int32_t rc = check_fitted(g, d);
    if (rc) return rc;
    if (kind < ABO_ACQ_EI || kind > ABO_ACQ_MEAN) return fail(ABO_EINVAL, "abo_acq: unknown acquisition kind %d", kind);
    return acq_terms_impl(g, Z, M, d, z_space, one_term(kind, p0, best_y), idx_base, scores, out_space, k, top_val, top_idx, top_space);
}

int32_t abo::acq_terms_ex(abo_gp* g, const double* Z, int64_t M, int32_t d, int32_t z_space, const abo_acq_term* terms, int32_t nterms,
                          int64_t idx_base, double* scores, int32_t out_space, int32_t k, double* top_val, int64_t* top_idx,
                          int32_t top_space) {
    int32_t rc = check_fitted(g, d);
    if (rc) return rc;
    AcqTerms t{};
    rc = make_terms(g, terms, nterms, &t, "abo_acq_terms");
    if (rc) return rc;
    return acq_terms_impl(g, Z, M, d, z_space, t, idx_base, scores, out_space, k, top_val, top_idx, top_space);
}

extern "C" int32_t abo_acq_terms(abo_gp* g, const double* Z, int64_t M, int32_t d, int32_t z_space, const abo_acq_term* terms,
                                 int32_t nterms, int64_t idx_base, double* scores, int32_t k, double* top_val, int64_t* top_idx,
                                 int32_t out_space) {
    return abo::acq_terms_ex(g, Z, M, d, z_space, terms, nterms, idx_base, scores, out_space, k, top_val, top_idx, out_space);
}

namespace {
// after the synchronisation that followed a deferred fit: the verdict on the factorisation
int32_t fit_finish(abo_gp* g, int64_t* info) {
    if (g->h_info() != 0) {
        if (g->st && g->fitted) g->st->drop_view(g->N);
        g->fitted = false;
        if (info) *info = g->h_info();
        return fail(ABO_ENOTPD, "PosDefException: matrix is not positive definite; Cholesky factorization failed at %lld",
                    (long long)g->h_info());
    }
    fit_collect(g);
    return ABO_OK;
}
}  // namespace

extern "C" {

int32_t abo_fit_acq(abo_gp* g, const double* X, int64_t N, int32_t d, const double* y, int32_t space, int64_t* info, const double* Z,
                    int64_t M, int32_t z_space, int32_t kind, double p0, double best_y, int64_t idx_base, double* scores, int32_t k,
                    double* top_val, int64_t* top_idx, int32_t out_space) {
    if (info) *info = 0;
    if (!g || !X || !y) return fail(ABO_EINVAL, "abo_fit_acq: null argument");
    // an opt-in jitter retry needs the verdict on the factorisation before anything else is queued, and a gradient-enhanced
    // model's targets are reordered on the way in: both take the two calls this entry point otherwise fuses
    if (g->prm.jitter > 0.0 || g->p_out > 1) {
        const int32_t rc = abo_fit(g, X, N, d, y, space, info);
        return rc ? rc : abo_acq(g, Z, M, d, z_space, kind, p0, best_y, idx_base, scores, k, top_val, top_idx, out_space);
    }
    if (N < 1) return fail(ABO_EINVAL, "abo_fit_acq: need at least one training point");
    if (d < 1 || d > 65536) return fail(ABO_EINVAL, "abo_fit_acq: input dimension %d outside 1..65536", d);
    if (N > (int64_t)1 << 20) return fail(ABO_EINVAL, "abo_fit_acq: N = %lld too large", (long long)N);
    HIPCHK(hipSetDevice(g->prm.device));
    int32_t rc = fit_impl(g, X, N, d, y, space, info, 0, /*defer=*/true);
    if (rc) return rc;
    // the acquisition's launches go straight behind the fit's: after a failed pivot they work on finite leftovers or exit on
    // `info`, and their results are discarded below
    rc = abo::acq_ex(g, Z, M, d, z_space, kind, p0, best_y, idx_base, scores, out_space, k, top_val, top_idx, out_space);
    if (rc) { (void)wait_stream(g->stream); (void)fit_finish(g, info); return rc; }
    return fit_finish(g, info);           // acq_ex returned behind its stream synchronisation: the fit's scalars have landed
}

int32_t abo_nlml(abo_gp* g, double* out) {
    if (!g || !out) return fail(ABO_EINVAL, "abo_nlml: null argument");
    if (!g->fitted) return fail(ABO_EINVAL, "surrogate is not conditioned on data yet (call abo_fit first)");
    *out = 0.5 * ((double)g->N * std::log(2.0 * M_PI) + g->logdet + g->quad);
    return ABO_OK;
}

int32_t abo_nlml_grad(abo_gp* g, double* nlml, double* d_log_ell, double* d_log_sigma_f2) {
    if (!g) return fail(ABO_EINVAL, "abo_nlml_grad: null handle");
    if (!g->fitted) return fail(ABO_EINVAL, "surrogate is not conditioned on data yet (call abo_fit first)");
    if (g->from_append || g->st->max_live() != g->N)
        return fail(ABO_EINVAL, "abo_nlml_grad needs a freshly fitted model (hyper-parameter search refits anyway)");
    HIPCHK(hipSetDevice(g->prm.device));
    hipStream_t s = g->stream;
    const int64_t Np = g->Np, ld = g->st->cap;
    const int N = (int)g->N;
    HIPCHK(g->T.ensure(sizeof(double) * Np * Np));
    HIPCHK(g->partial.ensure(sizeof(double) * (Np / 16 + 8)));
    HIPCHK(g->scal.ensure(sizeof(double) * 8));
    // K⁻¹ = WᵀW on the lower 128-tiles: Kinv[i][j] = Σ_{k ≥ i} WT[i][k]·WT[j][k]
    GemmArgs a{};
    a.A = g->st->WT.as<double>(); a.lda = ld; a.B = g->st->WT.as<double>(); a.ldb = ld;
    a.C = g->T.as<double>(); a.ldc = Np; a.M = (int)Np; a.N = (int)Np; a.K = (int)Np;
    a.kmode = K_A_UPPER; a.lower_only = 1; a.batch = 1; a.alpha = 1.0; a.beta = 0.0;
    HIPCHK(g->events(EV_BASE + 3));
    hipEvent_t* ev = &g->evs()[EV_BASE];
    HIPCHK(hipEventRecord(ev[0], s));
    HIPCHK(launch_gemm_nt(a, s));
    HIPCHK(hipEventRecord(ev[1], s));
    NlmlGradArgs ga{};
    ga.Xs = g->st->Xs.as<double>(); ga.Kinv = g->T.as<double>(); ga.alpha = g->alpha.as<double>();
    ga.delta = g->st->delta.as<double>(); ga.partial = g->partial.as<double>(); ga.out = g->scal.as<double>() + 4;
    ga.ld = Np; ga.N = N; ga.Np = (int)Np; ga.dp = g->dp; ga.family = g->prm.family; ga.sigma_f2 = g->prm.sigma_f2;
    if (g->p_out > 1) {
        // gradient-enhanced GP: dK/dlog(ell) of the multi-output system as a matrix (same generator as K_XX with the
        // derivative triple), then the weighted sum against K^-1 - alpha alpha^T
        HIPCHK(g->Kxz.ensure(sizeof(double) * Np * Np));
        KgenArgs ka{};
        ka.Xs = g->st->Xs.as<double>(); ka.Z = g->st->Xraw.as<double>(); ka.alpha = nullptr; ka.Kout = g->Kxz.as<double>();
        ka.mu = nullptr; ka.ldk = Np; ka.M = g->npts; ka.j0 = 0; ka.Mc = (int)Np; ka.N = (int)g->npts; ka.Np = (int)Np;
        ka.d = g->d; ka.dp = g->dp; ka.family = g->prm.family; ka.s = 1.0 / g->prm.ell; ka.sigma_f2 = g->prm.sigma_f2;
        ka.mean_c = 0.0; ka.pt = g->p_out; ka.pc = g->p_out; ka.point_major = 1; ka.dlogell = 1;
        HIPCHK(launch_kgen(ka, s));
        HIPCHK(launch_nlml_grad_matrix(ga, g->Kxz.as<double>(), Np, s));
    } else {
        HIPCHK(launch_nlml_grad(ga, s));
    }
    HIPCHK(hipEventRecord(ev[2], s));
    double o[4];
    HIPCHK(hipMemcpyAsync(o, g->scal.as<double>() + 4, sizeof o, hipMemcpyDeviceToHost, s));
    HIPCHK(wait_stream(s));
    g->tm.nlml_kinv_ms = ev_ms(ev[0], ev[1]);
    g->tm.nlml_trace_ms = ev_ms(ev[1], ev[2]);
    // ∂NLML/∂θ = ½ tr((K⁻¹ − ααᵀ) ∂K/∂θ);  ∂K/∂log σ_f² = K − noise·I  and  K α = δ
    const double noise = g->st->noise_used;
    if (nlml) *nlml = 0.5 * ((double)g->N * std::log(2.0 * M_PI) + g->logdet + g->quad);
    if (d_log_ell) *d_log_ell = 0.5 * o[0];
    if (d_log_sigma_f2) *d_log_sigma_f2 = 0.5 * ((double)N - noise * o[1] - o[3] + noise * o[2]);
    return ABO_OK;
}

// host vector of a gradient-enhanced model: the library's point-major order v[i·p + q] → the ABI's by-outputs order v[q·N + i]
static void to_by_outputs(double* v, int64_t N, int p) {
    std::vector<double> t(v, v + N * p);
    for (int64_t i = 0; i < N; ++i)
        for (int q = 0; q < p; ++q) v[(int64_t)q * N + i] = t[i * p + q];
}

int32_t abo_get_n(abo_gp* g, int64_t* N, int32_t* d) {
    if (!g) return fail(ABO_EINVAL, "null handle");
    if (N) *N = g->fitted ? g->npts : 0;
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
    const int64_t N = g->N;
    hipStream_t s = g->stream;
    if (L) {
        HIPCHK(hipMemcpy2DAsync(L, sizeof(double) * N, g->st->K.p, sizeof(double) * g->st->cap, sizeof(double) * N, N,
                                hipMemcpyDeviceToHost, s));
    }
    if (Linv) {
        HIPCHK(hipMemcpy2DAsync(Linv, sizeof(double) * N, g->st->W.p, sizeof(double) * g->st->cap, sizeof(double) * N, N,
                                hipMemcpyDeviceToHost, s));
    }
    if (alpha) HIPCHK(hipMemcpyAsync(alpha, g->alpha.p, sizeof(double) * N, hipMemcpyDeviceToHost, s));
    HIPCHK(wait_stream(s));
    if (alpha && g->p_out > 1) to_by_outputs(alpha, g->npts, g->p_out);
    if (L)   // off-diagonal upper blocks of the in-place factor still hold K: present a clean L
        for (int64_t i = 0; i < N; ++i)
            for (int64_t j = i + 1; j < N; ++j) L[i * N + j] = 0.0;
    return ABO_OK;
}

int32_t abo_get_data(abo_gp* g, double* X, double* y) {
    if (!g) return fail(ABO_EINVAL, "null handle");
    if (!g->fitted) return fail(ABO_EINVAL, "surrogate is not conditioned on data yet (call abo_fit first)");
    HIPCHK(hipSetDevice(g->prm.device));
    hipStream_t s = g->stream;
    if (X) HIPCHK(hipMemcpyAsync(X, g->st->Xraw.p, sizeof(double) * g->npts * g->d, hipMemcpyDeviceToHost, s));
    if (y) HIPCHK(hipMemcpyAsync(y, g->st->ybuf.p, sizeof(double) * g->N, hipMemcpyDeviceToHost, s));
    HIPCHK(wait_stream(s));
    if (y && g->p_out > 1) to_by_outputs(y, g->npts, g->p_out);
    return ABO_OK;
}

// ---- resident candidate sets (C5: greedy q-EI on a fixed grid with O(N·M) down-dates) ------------
// Bring the set's block-form q-EI state (abo_cand::Qei) in line with a model of `rows_now` factor rows on the same storage: the rows
// appended since the state's base must be the chain's first points, bit for bit.  A model with FEWER appended rows than the chain
// has real entries (the caller rolled the set back — abo_cand_restore — and continues from an earlier point of the same lineage)
// keeps the chain: the entries beyond become fantasies again (a conditioning sequence in that order), and the blocks built on the
// longer model (their columns hold those conditionings) are dropped.  Anything else resets the state.  *ok: the state is usable.
static int32_t qei_resync(abo_gp* g, abo_cand* c, int64_t rows_now, bool* ok) {
    abo_cand::Qei& Q = c->qei;
    *ok = false;
    if (Q.N < 0) return ABO_OK;
    const int64_t i = rows_now - Q.N;
    if (Q.open || Q.gen != g->st->gen || i < 0 || i > Q.nreal) { Q = abo_cand::Qei(); return ABO_OK; }
    if (i > 0) {
        std::vector<double> rows((size_t)i * g->d);
        HIPCHK(hipMemcpyAsync(rows.data(), g->st->Xraw.as<double>() + Q.N * g->d, sizeof(double) * rows.size(), hipMemcpyDeviceToHost, g->stream));
        HIPCHK(wait_stream(g->stream));
        if (memcmp(rows.data(), Q.chain_x.data(), sizeof(double) * rows.size())) { Q = abo_cand::Qei(); return ABO_OK; }
    }
    if (i < Q.nreal) {
        Q.nreal = (int)i;
        for (int b = 0; b < Q.nblk_cap; ++b)
            if (Q.blk_base[b] > Q.nreal)
                for (int t = 0; t < Q.T16; ++t) Q.slot_gidx[(size_t)b * Q.T16 + t] = -1;
    }
    *ok = true;
    return ABO_OK;
}

static int32_t cand_topk(abo_gp* g, abo_cand* c, const double* sc_d, int32_t k, int64_t idx_base, double* top_val,
                         int64_t* top_idx, int32_t out_space) {
    hipStream_t s = g->stream;
    const int64_t we = topk_workspace_entries(c->M, k);
    HIPCHK(c->tk_keys0.ensure(sizeof(uint64_t) * we));
    HIPCHK(c->tk_keys1.ensure(sizeof(uint64_t) * we));
    HIPCHK(c->tk_idx0.ensure(sizeof(int64_t) * we));
    HIPCHK(c->tk_idx1.ensure(sizeof(int64_t) * we));
    TopkWork w{{c->tk_keys0.as<uint64_t>(), c->tk_keys1.as<uint64_t>()}, {c->tk_idx0.as<int64_t>(), c->tk_idx1.as<int64_t>()}};
    double* tv = top_val;
    int64_t* ti = top_idx;
    if (out_space == ABO_HOST) {
        HIPCHK(c->top_val.ensure(sizeof(double) * k));
        HIPCHK(c->top_idx.ensure(sizeof(int64_t) * k));
        tv = c->top_val.as<double>();
        ti = c->top_idx.as<int64_t>();
    }
    HIPCHK(launch_topk(sc_d, c->M, k, idx_base, w, tv, ti, s));
    if (out_space == ABO_HOST) {
        int32_t rc = copy_out(top_val, tv, sizeof(double) * k, ABO_HOST, s); if (rc) return rc;
        rc = copy_out(top_idx, ti, sizeof(int64_t) * k, ABO_HOST, s); if (rc) return rc;
    }
    return ABO_OK;
}

int32_t abo_cand_refresh(abo_gp* g, abo_cand* c) {
    if (!c) return fail(ABO_EINVAL, "abo_cand_refresh: null candidate set");
    int32_t rc = check_fitted(g, c->d);
    if (rc) return rc;
    if (g->prm.device != c->device) return fail(ABO_EINVAL, "candidate set lives on device %d, model on %d", c->device, g->prm.device);
    HIPCHK(hipSetDevice(g->prm.device));
    if (c->M > 0) {
        // keep K_ZX resident when it fits the budget
        const char* lim = getenv("ABO_CAND_KZX_GIB");
        const double budget = (lim ? atof(lim) : 64.0) * 1073741824.0;
        const int64_t rows = pad_up(c->M, TB), ldz = g->st->cap;
        const double need = (double)rows * (double)ldz * sizeof(double);
        c->kzx_ld = 0;
        if (need <= budget) {
            const void* before = c->Kzx.p;
            const hipError_t e = c->Kzx.ensure((size_t)need);
            if (e == hipSuccess) {
                // columns ≥ Np are never written by the refresh: zero a new allocation once (later appends fill them)
                if (c->Kzx.p != before) HIPCHK(hipMemsetAsync(c->Kzx.p, 0, c->Kzx.cap, g->stream));
                c->kzx_ld = ldz;
            } else {
                (void)hipGetLastError();              // does not fit next to the model: fall back to recomputation
            }
        } else {
            c->Kzx.release();
        }
        HIPCHK(hipEventRecord(g->evs()[5], g->stream));
        rc = posterior(g, c->Z.as<double>(), c->M, -1, 0.0, 0.0, c->mu.as<double>(), c->var.as<double>(), nullptr, 1, 0,
                       c->kzx_ld ? c->Kzx.as<double>() : nullptr, c->kzx_ld);
        if (rc) return rc;
        HIPCHK(hipEventRecord(g->evs()[6], g->stream));
        HIPCHK(wait_stream(g->stream));
        collect_posterior_timings(g, c->M, true);
        g->tm.acq_topk_ms = 0.0;
        g->tm.acq_total_ms = ev_ms(g->evs()[5], g->evs()[6]);
    }
    c->synced_gen = g->st->gen;
    c->synced_N = g->N;
    c->qei = abo_cand::Qei();                             // blocks and chain of an earlier q-EI batch belong to the old posterior
    return ABO_OK;
}

int32_t abo_cand_create(abo_gp* g, const double* Z, int64_t M, int32_t d, int32_t z_space, abo_cand** out) {
    if (!out) return fail(ABO_EINVAL, "abo_cand_create: null argument");
    int32_t rc = check_fitted(g, d);
    if (rc) return rc;
    if (M < 0 || (M > 0 && !Z)) return fail(ABO_EINVAL, "abo_cand_create: bad candidate buffer");
    HIPCHK(hipSetDevice(g->prm.device));
    abo_cand* c = new (std::nothrow) abo_cand();
    if (!c) return fail(ABO_ENOMEM, "abo_cand_create: host allocation failed");
    c->set_device(g->prm.device);
    c->d = d; c->M = M;
    hipError_t e = c->Z.ensure(sizeof(double) * (M > 0 ? M : 1) * d);
    if (e == hipSuccess) e = c->mu.ensure(sizeof(double) * (M > 0 ? M : 1));
    if (e == hipSuccess) e = c->var.ensure(sizeof(double) * (M > 0 ? M : 1));
    if (e != hipSuccess) { c->free_all(); delete c; return fail(ABO_ENOMEM, "abo_cand_create: %s", hipGetErrorString(e)); }
    rc = copy_in(c->Z.p, Z, sizeof(double) * M * d, z_space, g->stream);
    if (!rc) rc = abo_cand_refresh(g, c);
    if (rc) { (void)wait_stream(g->stream); c->free_all(); delete c; return rc; }
    *out = c;
    return ABO_OK;
}

int32_t abo_cand_destroy(abo_cand* c) {
    if (!c || g_exiting.load()) return ABO_OK;
    if (abo::gone(hipSetDevice(c->device))) { g_exiting.store(true); return ABO_OK; }
    c->free_all();
    delete c;
    return ABO_OK;
}

int32_t abo_cand_downdate(abo_gp* g, abo_cand* c) {
    if (!c) return fail(ABO_EINVAL, "abo_cand_downdate: null candidate set");
    int32_t rc = check_fitted(g, c->d);
    if (rc) return rc;
    const int P = g->p_out;                                // rows the append added: 1, or p for a gradient-enhanced model
    if (!g->from_append || g->st->gen != c->synced_gen || g->N != c->synced_N + P)
        return fail(ABO_EINVAL, "abo_cand_downdate: the model is not the one-point append of the model this candidate set "
                                "was last evaluated with (call abo_cand_refresh)");
    HIPCHK(hipSetDevice(g->prm.device));
    hipStream_t s = g->stream;
    if (c->M > 0) {
        // c(z) = k(z,x*) − k_zᵀ K⁻¹ k_* = Σ_{k ≤ N} k(z, x_k)·vext[k]: one kernel-evaluation pass, nothing stored
        HIPCHK(c->cdot.ensure(sizeof(double) * pad_up(c->M, 16)));
        const bool resident = c->kzx_ld > 0 && c->kzx_ld == g->st->cap;
        HIPCHK(g->events(8));
        // The set's block-form q-EI state (blocks + chain) follows the model through real appends.  Entry i = number of rows appended
        // since the state's base.  If the appended point is the next FANTASY of the last batch (the picks appended for real, in
        // order), its column c_i(z) is in the chain already (it does not depend on the observed value): no pass over K_ZX.  Otherwise
        // the pass below runs and its column joins the chain as real entry i (the fantasies behind it are dropped).  The model's rows
        // since the base must be the chain's points, bit for bit; anything else resets the state.
        int chain_i = -1, chain_put = -1;
        abo_cand::Qei& Q = c->qei;
        std::vector<double> newx;
        if (Q.N >= 0 && (P != 1 || !resident || getenv("ABO_QEI_NO_CHAIN"))) Q = abo_cand::Qei();
        bool qok = false;
        { const int32_t r = qei_resync(g, c, c->synced_N, &qok); if (r) return r; }
        if (qok) {
            const int i = Q.nreal;
            std::vector<double> row(g->d);
            if ((int)g->ap_x.size() == g->d && Q.N + i == g->N - 1) {
                row = g->ap_x;                                     // the appended point as the host handed it over: no read-back, no wait
            } else {
                HIPCHK(hipMemcpyAsync(row.data(), g->st->Xraw.as<double>() + (Q.N + i) * g->d, sizeof(double) * g->d, hipMemcpyDeviceToHost, s));
                HIPCHK(wait_stream(s));
            }
            if (i < Q.nchain && !memcmp(row.data(), &Q.chain_x[(size_t)i * g->d], sizeof(double) * g->d)) chain_i = i;
            else if (i < Q.chain_rows) { chain_put = i; newx = row; }
            else Q = abo_cand::Qei();                                                                                  // no room: start over
        }
        g->tm.downdate_from_chain = chain_i >= 0 ? 1 : 0;
        HIPCHK(hipEventRecord(g->evs()[5], s));
        double pass_ms = 0.0;
        if (chain_i >= 0) {
            // the appended row's column keeps the resident K_ZX current for later passes
            HIPCHK(launch_cand_newcol(g->st->Xs.as<double>(), c->Z.as<double>(), c->Kzx.as<double>(), c->kzx_ld, c->M, (int)(g->N - 1),
                                      g->d, g->dp, g->prm.family, 1.0 / g->prm.ell, g->prm.sigma_f2, s));
            HIPCHK(hipEventRecord(g->evs()[6], s));
            HIPCHK(launch_downdate(c->mu.as<double>(), c->var.as<double>(), c->qchain.as<double>() + (size_t)chain_i * Q.Mp, c->M,
                                   g->ap_beta, g->ap_s2, s));
            Q.chain_s[chain_i] = g->ap_s2;                         // what the stored variance was down-dated with
            Q.nreal = chain_i + 1;                                 // (the fantasies behind it stay: the next picks of the batch)
        }
        for (int q = 0; chain_i < 0 && q < P; ++q) {       // one rank-1 down-date per appended row, in append order
            const int64_t Rq = g->N - P + q;               // index of the appended row
            const double* vext = g->vext.as<double>() + (P > 1 ? (int64_t)q * g->st->cap : 0);
            const double s2 = P > 1 ? g->ap_s2v[q] : g->ap_s2, beta = P > 1 ? g->ap_betav[q] : g->ap_beta;
            if (resident) {
                // the appended row's column, then one streaming mat-vec over the resident K_ZX
                if (P == 1) {
                    HIPCHK(launch_cand_newcol(g->st->Xs.as<double>(), c->Z.as<double>(), c->Kzx.as<double>(), c->kzx_ld, c->M,
                                              (int)Rq, g->d, g->dp, g->prm.family, 1.0 / g->prm.ell, g->prm.sigma_f2, s));
                } else {
                    HIPCHK(launch_cand_newcol_grad(g->st->Xs.as<double>(), c->Z.as<double>(), c->Kzx.as<double>(), c->kzx_ld, c->M,
                                                   (int)Rq, (int)(g->npts - 1), q, g->d, g->dp, g->prm.family, 1.0 / g->prm.ell,
                                                   g->prm.sigma_f2, s));
                }
                HIPCHK(launch_cand_gemv(c->Kzx.as<double>(), c->kzx_ld, vext, (int)(Rq + 1), c->M, c->cdot.as<double>(), s));
            }
            const int64_t step = 65536;
            for (int64_t j0 = 0; !resident && j0 < c->M; j0 += step) {
                const int64_t m = (c->M - j0) < step ? (c->M - j0) : step;
                KgenArgs ka{};
                ka.Xs = g->st->Xs.as<double>(); ka.Z = c->Z.as<double>(); ka.alpha = vext; ka.Kout = nullptr;
                ka.mu = c->cdot.as<double>() + j0; ka.ldk = 0; ka.M = c->M; ka.j0 = j0; ka.Mc = (int)pad_up(m, 16);
                ka.N = (int)g->npts; ka.Np = (int)pad_up(Rq + 1, TB); ka.d = g->d; ka.dp = g->dp; ka.family = g->prm.family;
                ka.s = 1.0 / g->prm.ell; ka.sigma_f2 = g->prm.sigma_f2; ka.mean_c = 0.0;
                if (P > 1) { ka.pt = P; ka.pc = 1; ka.point_major = 0; ka.rvalid = (int)(Rq + 1); }
                else { ka.N = (int)(Rq + 1); }
                HIPCHK(launch_kgen(ka, s));
            }
            if (q == P - 1) HIPCHK(hipEventRecord(g->evs()[6], s));
            HIPCHK(launch_downdate(c->mu.as<double>(), c->var.as<double>(), c->cdot.as<double>(), c->M, beta, s2, s));
            if (chain_put >= 0) {                                  // the pass's column joins the chain as real entry chain_put
                HIPCHK(hipMemcpyAsync(c->qchain.as<double>() + (size_t)chain_put * Q.Mp, c->cdot.p, sizeof(double) * c->M, hipMemcpyDeviceToDevice, s));
                Q.chain_x.resize((size_t)chain_put * g->d);
                Q.chain_x.insert(Q.chain_x.end(), newx.begin(), newx.end());
                Q.chain_s.resize(chain_put);
                Q.chain_s.push_back(s2);
                Q.nreal = Q.nchain = chain_put + 1;
            }
        }
        HIPCHK(wait_stream(s));
        pass_ms = ev_ms(g->evs()[5], g->evs()[6]);
        g->tm.downdate_ms = pass_ms;
        g->tm.downdate_bytes = (resident && chain_i < 0) ? 8.0 * (double)g->N * (double)c->M * P : 0.0;
    }
    c->synced_N = g->N;
    return ABO_OK;
}

int32_t abo_cand_acq(abo_gp* g, abo_cand* c, int32_t kind, double p0, double best_y, int64_t idx_base, double* scores,
                     int32_t k, double* top_val, int64_t* top_idx, int32_t out_space) {
    return abo::cand_acq_ex(g, c, kind, p0, best_y, idx_base, scores, out_space, k, top_val, top_idx, out_space);
}

}  // extern "C"

int32_t abo::cand_acq_ex(abo_gp* g, abo_cand* c, int32_t kind, double p0, double best_y, int64_t idx_base, double* scores,
                         int32_t out_space, int32_t k, double* top_val, int64_t* top_idx, int32_t top_space) {
    if (!g || !c) return fail(ABO_EINVAL, "abo_cand_acq: null argument");
    if (kind < ABO_ACQ_EI || kind > ABO_ACQ_MEAN) return fail(ABO_EINVAL, "abo_cand_acq: unknown acquisition kind %d", kind);
    if (k < 0) return fail(ABO_EINVAL, "abo_cand_acq: k = %d is negative", k);
    if (k > 0 && (!top_val || !top_idx)) return fail(ABO_EINVAL, "abo_cand_acq: k > 0 needs top_val and top_idx");
    if (g->prm.device != c->device) return fail(ABO_EINVAL, "candidate set lives on device %d, model on %d", c->device, g->prm.device);
    HIPCHK(hipSetDevice(g->prm.device));
    hipStream_t s = g->stream;
    double* sc_d = (scores && out_space == ABO_DEVICE) ? scores : nullptr;
    if (!sc_d) { HIPCHK(c->score.ensure(sizeof(double) * (c->M > 0 ? c->M : 1))); sc_d = c->score.as<double>(); }
    HIPCHK(launch_score(c->mu.as<double>(), c->var.as<double>(), sc_d, c->M, kind, p0, best_y, s));
    if (k > 0) { int32_t rc = cand_topk(g, c, sc_d, k, idx_base, top_val, top_idx, top_space); if (rc) return rc; }
    if (scores && out_space == ABO_HOST && c->M > 0) {
        int32_t rc = copy_out(scores, sc_d, sizeof(double) * c->M, ABO_HOST, s);
        if (rc) return rc;
    }
    HIPCHK(wait_stream(s));
    return ABO_OK;
}

// ---- greedy q-EI, block form (qei.hip; include/abo_hip.h "block form"): per-shard steps + the driver ------------------------------
namespace {

// Julia's isless-descending order on scores (misc.hip: score_key): NaN first, then +Inf … −Inf with 0.0 before −0.0
uint64_t host_score_key(double v) {
    if (v != v) return ~0ull;
    uint64_t b;
    memcpy(&b, &v, 8);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

std::atomic<int> g_qei_block{-1};
int qei_default_block() {
    int v = g_qei_block.load();
    if (v < 0) {
        const char* e = getenv("ABO_QEI_BLOCK");
        // 16 columns ride entirely under the K_ZX stream (6.3 TB/s), 32 cost 10 % more per pass — and halve how often a batch has to
        // build a block: over a 512-step cycle of config 5 with noisy observations one step in 8 rebuilt at T = 16, one in 16 at 32,
        // one in 27 at 64 (whose pass is MFMA-bound); mean step 1.13 / 0.98 / 0.99 ms (tools/c5_cycle.py, profiles/r06_c5_cycle.txt)
        v = e ? atoi(e) : 32;
        if (v < 0) v = 0;
        if (v > QEI_MAXT) v = QEI_MAXT;
        g_qei_block.store(v);
    }
    return v;
}

// chunk length of the split-k products against L⁻¹ (K⁻¹K_XT): about 16 chunks per column tile, so that the (tile, chunk) workgroups
// of the triangular range are a few times the device's CUs — the product streams L⁻¹ once and is HBM-bound
int qei_ksplit(int Np) {
    int ks = Np / 16 / 128 * 128;
    return ks < 128 ? 128 : ks;
}

struct QeiWork { double *P, *Ps, *KXT, *V, *U, *Ct; int ks, nz; size_t bytes; };
QeiWork qei_carve(void* base, int T16, int d, int dp, int Np) {
    QeiWork w{};
    w.ks = qei_ksplit(Np);
    w.nz = (Np + w.ks - 1) / w.ks;
    char* p = static_cast<char*>(base);
    auto take = [&](size_t n) { double* r = reinterpret_cast<double*>(p); p += (n * sizeof(double) + 255) / 256 * 256; return r; };
    w.P = take((size_t)T16 * d);
    w.Ps = take((size_t)T16 * dp);
    w.KXT = take((size_t)T16 * Np);
    w.V = take((size_t)T16 * Np);
    w.U = take((size_t)T16 * Np);
    w.Ct = take((size_t)w.nz * T16 * Np);
    w.bytes = (size_t)(p - static_cast<char*>(base));
    return w;
}

// The block slots are keyed by GLOBAL candidate index = idx_base + local index.  A caller that passes another idx_base than the one
// the slots were keyed with (a continuation from an earlier batch, or a change inside a batch) would match a pick against another
// candidate's covariance column: the slots are dropped (the chain is per local candidate and stays); the next pick builds a block.
int32_t qei_rebase(abo_cand* c, int64_t idx_base) {
    abo_cand::Qei& Q = c->qei;
    if (idx_base < 0) return fail(ABO_EINVAL, "q-EI: idx_base = %lld", (long long)idx_base);
    if (Q.idx_base != idx_base) {
        if (Q.idx_base >= 0) Q.slot_gidx.assign(Q.slot_gidx.size(), -1);
        Q.idx_base = idx_base;
    }
    return ABO_OK;
}

int qei_find_slot(const abo_cand* c, int64_t gidx) {
    const std::vector<int64_t>& v = c->qei.slot_gidx;
    for (size_t i = 0; i < v.size(); ++i) if (v[i] == gidx) return (int)i;
    return -1;
}

}  // namespace

int32_t abo::qei_eligible(abo_gp* g, abo_cand* c, int q) {
    if (!g || !c) return fail(ABO_EINVAL, "q-EI: null argument");
    int32_t rc = check_fitted(g, c->d);
    if (rc) return rc;
    if (g->prm.device != c->device) return fail(ABO_EINVAL, "candidate set lives on device %d, model on %d", c->device, g->prm.device);
    if (g->st->gen != c->synced_gen || g->N != c->synced_N)
        return fail(ABO_EINVAL, "q-EI: the candidate set is not in sync with this model (abo_cand_refresh / abo_cand_downdate)");
    if (g->p_out > 1) return fail(ABO_EINVAL, "q-EI, block form: gradient-enhanced models take the plain loop");
    if (c->M > 0 && !(c->kzx_ld > 0 && c->kzx_ld == g->st->cap)) return fail(ABO_EINVAL, "q-EI, block form: K_ZX of the set is not resident");
    if (q < 1 || q > QEI_MAXQ) return fail(ABO_EINVAL, "q-EI, block form: q = %d outside 1..%d", q, QEI_MAXQ);
    if ((size_t)QEI_MAXT * c->d * sizeof(double) > 65536) return fail(ABO_EINVAL, "q-EI, block form: d = %d > 128 takes the plain loop", c->d);
    return ABO_OK;
}

int32_t abo::qei_begin(abo_gp* g, abo_cand* c, int q, int T, bool snapshot) {
    int32_t rc = abo::qei_eligible(g, c, q);
    if (rc) return rc;
    if (T <= 0) T = qei_default_block();
    if (T < 1) return fail(ABO_EINVAL, "q-EI, block form: block size 0 (the plain loop is a different call)");
    if (T > QEI_MAXT) T = QEI_MAXT;
    HIPCHK(hipSetDevice(g->prm.device));
    abo_cand::Qei& Q = c->qei;
    const int T16 = (int)pad_up(T, 16);
    const int64_t Mp = pad_up(c->M > 0 ? c->M : 1, TB);
    hipStream_t s = g->stream;
    // a continuation: the state's base plus (some of) its real entries IS this model (rows compared bit for bit: qei_resync), same
    // shapes, room for q more
    bool cont = false;
    if (Q.N >= 0 && (Q.T16 != T16 || Q.Mp != Mp || getenv("ABO_QEI_NO_REUSE"))) Q = abo_cand::Qei();
    { const int32_t r = qei_resync(g, c, g->N, &cont); if (r) return r; }
    if (cont && Q.nreal + q > QEI_MAXQ) cont = false;
    if (!cont) {
        Q = abo_cand::Qei();
        Q.gen = g->st->gen; Q.N = g->N; Q.Mp = Mp; Q.T16 = T16;
        Q.nblk_cap = 4;                                    // ring of blocks: a pick outside all of them rebuilds the oldest
        Q.slot_gidx.assign((size_t)Q.nblk_cap * T16, -1);
        Q.slot_x.assign((size_t)Q.nblk_cap * T16 * c->d, 0.0);
        Q.blk_base.assign(Q.nblk_cap, 0);
    }
    HIPCHK(c->qblk.ensure(sizeof(double) * (size_t)Q.nblk_cap * T16 * Mp));
    // chain rows: the real entries so far + this batch's picks, and room for the real appends that follow it
    const int want = Q.nreal + q + 8 < QEI_MAXQ ? Q.nreal + q + 8 : QEI_MAXQ;
    if (want > Q.chain_rows) {
        if (Q.nreal > 0) {                                 // grow, keeping the real entries
            ScratchBuf keep(g->prm.device, s);
            HIPCHK(keep.b.ensure(sizeof(double) * (size_t)Q.nreal * Mp));
            HIPCHK(hipMemcpyAsync(keep.b.p, c->qchain.p, sizeof(double) * (size_t)Q.nreal * Mp, hipMemcpyDeviceToDevice, s));
            HIPCHK(wait_stream(s));
            HIPCHK(c->qchain.ensure(sizeof(double) * (size_t)want * Mp));
            HIPCHK(hipMemcpyAsync(c->qchain.p, keep.b.p, sizeof(double) * (size_t)Q.nreal * Mp, hipMemcpyDeviceToDevice, s));
        } else {
            HIPCHK(c->qchain.ensure(sizeof(double) * (size_t)want * Mp));
        }
        Q.chain_rows = want;
    }
    const QeiWork w = qei_carve(nullptr, T16, g->d, g->dp, (int)g->Np);
    HIPCHK(c->qwork.ensure(w.bytes));
    // snapshot of the stored posterior, in buffers of its own (the caller's abo_cand_save snapshot stays what it is): the batch is
    // rolled back at _end
    const size_t bytes = sizeof(double) * (c->M > 0 ? c->M : 1);
    HIPCHK(c->qmu.ensure(bytes));
    HIPCHK(c->qvar.ensure(bytes));
    if (snapshot) {                                        // (the device pick loop takes the snapshot in its first launch)
        HIPCHK(hipMemcpyAsync(c->qmu.p, c->mu.p, sizeof(double) * c->M, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(c->qvar.p, c->var.p, sizeof(double) * c->M, hipMemcpyDeviceToDevice, s));
    }
    // the fantasies of the last batch go; blocks and real entries stay
    Q.nchain = Q.nreal;
    Q.chain_x.resize((size_t)Q.nreal * c->d);
    Q.chain_s.resize(Q.nreal);
    Q.open = true; Q.qmax = q; Q.batch0 = Q.nreal; Q.builds = 0;
    Q.block_ms = Q.pass_ms = Q.pass_bytes = Q.pass_flop = 0.0;
    return ABO_OK;
}

// EI over the shard, its k best as records in DEVICE memory rec_d (k × (4 + d + picks so far) doubles); nothing is waited for
int32_t abo::qei_top(abo_gp* g, abo_cand* c, double xi, double best_y, int64_t idx_base, int k, double* rec_d) {
    if (!g || !c || !rec_d) return fail(ABO_EINVAL, "abo_cand_qei_top: null argument");
    if (!c->qei.open || c->qei.gen != g->st->gen || c->qei.N + c->qei.nreal != g->N) return fail(ABO_EINVAL, "abo_cand_qei_top: no batch open on this model (abo_cand_qei_begin)");
    if (k < 1 || k > 1024) return fail(ABO_EINVAL, "abo_cand_qei_top: k = %d outside 1..1024", k);
    if (int32_t r = qei_rebase(c, idx_base)) return r;
    HIPCHK(hipSetDevice(g->prm.device));
    hipStream_t s = g->stream;
    HIPCHK(c->score.ensure(sizeof(double) * (c->M > 0 ? c->M : 1)));
    HIPCHK(c->top_val.ensure(sizeof(double) * k));
    HIPCHK(c->top_idx.ensure(sizeof(int64_t) * k));
    HIPCHK(launch_score(c->mu.as<double>(), c->var.as<double>(), c->score.as<double>(), c->M, ABO_ACQ_EI, xi, best_y, s));
    int32_t rc = cand_topk(g, c, c->score.as<double>(), k, idx_base, c->top_val.as<double>(), c->top_idx.as<int64_t>(), ABO_DEVICE);
    if (rc) return rc;
    HIPCHK(launch_qei_record(c->top_val.as<double>(), c->top_idx.as<int64_t>(), k, idx_base, c->Z.as<double>(), c->mu.as<double>(),
                             c->var.as<double>(), c->qchain.as<double>(), c->qei.Mp, c->qei.nchain, c->d, rec_d, s));
    return ABO_OK;
}

// Cov₀(z, p_t) of T points for every candidate of the shard: K⁻¹K_XT by two split-k products against L⁻¹ / L⁻ᵀ, ONE product over the
// resident K_ZX, the kernel values k(z, p_t) on top.  pts: T × d HOST doubles, gidx their global candidate indices.
int32_t abo::qei_block(abo_gp* g, abo_cand* c, const double* pts, const int64_t* gidx, int T) {
    if (!g || !c || !pts || !gidx) return fail(ABO_EINVAL, "abo_cand_qei_block: null argument");
    abo_cand::Qei& Q = c->qei;
    if (!Q.open || Q.gen != g->st->gen || Q.N + Q.nreal != g->N) return fail(ABO_EINVAL, "abo_cand_qei_block: no batch open on this model (abo_cand_qei_begin)");
    if (T < 1 || T > Q.T16) return fail(ABO_EINVAL, "abo_cand_qei_block: T = %d outside 1..%d", T, Q.T16);
    HIPCHK(hipSetDevice(g->prm.device));
    hipStream_t s = g->stream;
    const int T16 = Q.T16, d = g->d, dp = g->dp, N = (int)g->N, Np = (int)g->Np;
    const int64_t ld = g->st->cap, Mp = Q.Mp;
    const int blk = Q.next_blk;
    Q.next_blk = (Q.next_blk + 1) % Q.nblk_cap;
    for (int t = 0; t < T16; ++t) Q.slot_gidx[(size_t)blk * T16 + t] = t < T ? gidx[t] : -1;
    memcpy(&Q.slot_x[(size_t)blk * T16 * g->d], pts, sizeof(double) * (size_t)T * g->d);
    Q.blk_base[blk] = Q.nreal;                             // the model of this build holds the real entries so far: they are in its columns
    ++Q.builds;
    if (c->M == 0) return ABO_OK;
    const QeiWork w = qei_carve(c->qwork.p, T16, d, dp, Np);
    double* C = c->qblk.as<double>() + (size_t)blk * T16 * Mp;
    HIPCHK(g->events(EV_BASE + 4));
    hipEvent_t* ev = &g->evs()[EV_BASE];
    HIPCHK(hipEventRecord(ev[0], s));
    HIPCHK(hipMemsetAsync(w.P, 0, sizeof(double) * (size_t)T16 * d, s));
    HIPCHK(hipMemcpyAsync(w.P, pts, sizeof(double) * (size_t)T * d, hipMemcpyHostToDevice, s));
    HIPCHK(launch_scale_points(w.P, w.Ps, T, T16, d, dp, 1.0 / g->prm.ell, s));
    HIPCHK(launch_qei_kxt(g->st->Xs.as<double>(), dp, N, Np, w.P, d, T, T16, g->prm.family, 1.0 / g->prm.ell, g->prm.sigma_f2, w.KXT, s));
    // V[t][i] = Σ_{k ≤ i} K_XT[t][k]·W[i][k] = (L⁻¹k_t)[i];  U[t][i] = Σ_{k ≥ i} V[t][k]·WT[i][k] = (K⁻¹k_t)[i]
    const bool via_skinny = getenv("ABO_QEI_PASS_SKINNY") != nullptr;      // A/B runs and the bit-equality test: gemm.hip's split-k kernel
    const int64_t sC = (int64_t)T16 * Np;
    GemmArgs g1{};
    g1.A = w.KXT; g1.lda = Np; g1.B = g->st->W.as<double>(); g1.ldb = ld; g1.C = w.Ct; g1.ldc = Np; g1.sC = sC;
    g1.M = TB; g1.N = Np; g1.K = Np; g1.kmode = K_B_LOWER; g1.lower_only = 0; g1.batch = 1; g1.alpha = 1.0; g1.beta = 0.0;
    g1.ksplit = w.ks; g1.mrows = T16;
    if (via_skinny) HIPCHK(launch_gemm_nt(g1, s));
    else HIPCHK(launch_qei_pass(w.KXT, Np, T16, g->st->W.as<double>(), ld, Np, Np, 1.0, w.Ct, Np, s, K_B_LOWER, w.ks, sC));
    HIPCHK(launch_splitk_reduce(w.Ct, Np, sC, w.nz, T16, Np, Np, w.ks, K_B_LOWER, w.V, Np, s));
    HIPCHK(launch_qei_zero_tail(w.V, Np, N, Np, T16, s));      // rows ≥ N of a shared factor may hold a discarded appended branch
    GemmArgs g2 = g1;
    g2.A = w.V; g2.B = g->st->WT.as<double>(); g2.kmode = K_B_UPPER;
    if (via_skinny) HIPCHK(launch_gemm_nt(g2, s));
    else HIPCHK(launch_qei_pass(w.V, Np, T16, g->st->WT.as<double>(), ld, Np, Np, 1.0, w.Ct, Np, s, K_B_UPPER, w.ks, sC));
    HIPCHK(launch_splitk_reduce(w.Ct, Np, sC, w.nz, T16, Np, Np, w.ks, K_B_UPPER, w.U, Np, s));
    HIPCHK(launch_qei_zero_tail(w.U, Np, N, Np, T16, s));
    // C[t][z] = −Σ_k U[t][k]·K_ZX[z][k]: one pass over the resident K_ZX for all T columns (columns ≥ N of K_ZX meet U = 0)
    HIPCHK(hipEventRecord(ev[1], s));
    if (via_skinny) {
        GemmArgs g3{};
        g3.A = w.U; g3.lda = Np; g3.B = c->Kzx.as<double>(); g3.ldb = c->kzx_ld; g3.C = C; g3.ldc = Mp; g3.sC = 0;
        g3.M = TB; g3.N = (int)Mp; g3.K = Np; g3.kmode = K_FULL; g3.lower_only = 0; g3.batch = 1; g3.alpha = -1.0; g3.beta = 0.0;
        g3.ksplit = Np; g3.mrows = T16;
        HIPCHK(launch_gemm_nt(g3, s));
    } else {
        HIPCHK(launch_qei_pass(w.U, Np, T16, c->Kzx.as<double>(), c->kzx_ld, Mp, Np, -1.0, C, Mp, s));
    }
    HIPCHK(hipEventRecord(ev[2], s));
    HIPCHK(launch_qei_cov(w.Ps, dp, c->Z.as<double>(), c->M, Mp, d, T, g->prm.family, 1.0 / g->prm.ell, g->prm.sigma_f2, C, s));
    HIPCHK(hipEventRecord(ev[3], s));
    HIPCHK(wait_stream(s));                          // (pts is the caller's; the events are read)
    Q.block_ms += ev_ms(ev[0], ev[3]);
    Q.pass_ms = ev_ms(ev[1], ev[2]);
    Q.pass_bytes = 8.0 * (double)g->N * (double)c->M;
    Q.pass_flop = 2.0 * (double)g->N * (double)c->M * (double)T16;
    return ABO_OK;
}

int abo::qei_block_default() { return qei_default_block(); }

size_t abo::qei_max_words(int d, int q, int T) {
    if (T <= 0) T = qei_default_block();
    if (T > QEI_MAXT) T = QEI_MAXT;
    if (T < 1) T = 1;
    (void)q;
    return (size_t)T * (size_t)(4 + d + QEI_MAXQ);        // a record carries one value per chain entry: real ones carried over + the batch's
}

int32_t abo::qei_has(const abo_cand* c, int64_t gidx) { return c && qei_find_slot(c, gidx) >= 0 ? 1 : 0; }

// condition the shard's stored variance on the pick `gidx` (a point of a block): chain vector n + 1, σ² −= c²/s with
// s = var_x + σ²_n (var_x: the stored σ² at the pick, from the winner's record), γ_i = cx[i]/s_i (cx: c_1(x) … c_n(x) from the same
// record).  s ≤ 0 is what the plain loop's bordered append reports as a failed pivot: ABO_ENOTPD, *info = N + n + 1.
int32_t abo::qei_pick(abo_gp* g, abo_cand* c, int64_t gidx, double var_x, const double* cx, int n, int64_t excl, int64_t* info) {
    if (info) *info = 0;
    if (!g || !c || (n > 0 && !cx)) return fail(ABO_EINVAL, "abo_cand_qei_pick: null argument");
    abo_cand::Qei& Q = c->qei;
    if (!Q.open || Q.gen != g->st->gen || Q.N + Q.nreal != g->N) return fail(ABO_EINVAL, "abo_cand_qei_pick: no batch open on this model (abo_cand_qei_begin)");
    if (n != Q.nchain) return fail(ABO_EINVAL, "abo_cand_qei_pick: %d chain values for a chain of %d picks", n, Q.nchain);
    if (Q.nchain - Q.batch0 >= Q.qmax || Q.nchain >= Q.chain_rows) return fail(ABO_EINVAL, "abo_cand_qei_pick: the batch was opened for %d picks", Q.qmax);
    if (excl >= c->M) return fail(ABO_EINVAL, "abo_cand_qei_pick: exclusion index %lld outside the shard", (long long)excl);
    const int slot = qei_find_slot(c, gidx);
    if (slot < 0) return fail(ABO_EINVAL, "abo_cand_qei_pick: candidate %lld is in no block (abo_cand_qei_block)", (long long)gidx);
    const double sj = var_x + g->st->noise_used;
    if (!(sj > 0.0)) {
        const long long at = (long long)(g->N + (Q.nchain - Q.nreal) + 1);
        if (info) *info = at;
        return fail(ABO_ENOTPD, "PosDefException: matrix is not positive definite; Cholesky factorization failed at %lld", at);
    }
    HIPCHK(hipSetDevice(g->prm.device));
    hipStream_t s = g->stream;
    QeiPickArgs a{};
    a.blk = c->qblk.as<double>() + (size_t)slot * Q.Mp;
    a.chain = c->qchain.as<double>();
    a.out = c->qchain.as<double>() + (size_t)Q.nchain * Q.Mp;
    a.var = c->var.as<double>();
    a.M = c->M; a.Mp = Q.Mp; a.nchain = Q.nchain; a.s = sj;
    a.first = Q.blk_base[slot / Q.T16];                    // entries before it are already in the block's columns
    for (int i = 0; i < n; ++i) a.gam[i] = cx[i] / Q.chain_s[i];
    HIPCHK(launch_qei_pick(a, s));
    if (excl >= 0) {
        const double ex[2] = {HUGE_VAL, 0.0};                // abo_cand_exclude: μ = +Inf, σ² = 0
        HIPCHK(hipMemcpyAsync(c->mu.as<double>() + excl, &ex[0], sizeof(double), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(c->var.as<double>() + excl, &ex[1], sizeof(double), hipMemcpyHostToDevice, s));
        HIPCHK(wait_stream(s));                      // (ex lives on this frame)
    }
    ++Q.nchain;
    Q.chain_s.push_back(sj);
    Q.chain_x.insert(Q.chain_x.end(), &Q.slot_x[(size_t)slot * c->d], &Q.slot_x[(size_t)slot * c->d] + c->d);
    return ABO_OK;
}

int32_t abo::qei_end(abo_gp* g, abo_cand* c) {
    if (!g || !c) return fail(ABO_EINVAL, "abo_cand_qei_end: null argument");
    if (!c->qei.open) return ABO_OK;
    HIPCHK(hipSetDevice(g->prm.device));
    HIPCHK(hipMemcpyAsync(c->mu.p, c->qmu.p, sizeof(double) * c->M, hipMemcpyDeviceToDevice, g->stream));
    HIPCHK(hipMemcpyAsync(c->var.p, c->qvar.p, sizeof(double) * c->M, hipMemcpyDeviceToDevice, g->stream));
    HIPCHK(wait_stream(g->stream));
    c->qei.open = false;
    return ABO_OK;
}

void abo::qei_get_stats(const abo_cand* c, int picks, double total_ms, abo_qei_stats* out) {
    if (!out) return;
    const abo_cand::Qei& Q = c->qei;
    out->picks = picks; out->block = Q.T16; out->block_builds = Q.builds;
    const int cond = Q.nchain - Q.batch0;                                  // picks of the batch that were conditioned on
    out->block_hits = cond > Q.builds ? cond - Q.builds : 0;               // each of them either found its point in a block or had one built
    out->total_ms = total_ms; out->block_ms = Q.block_ms; out->pass_ms = Q.pass_ms; out->pass_bytes = Q.pass_bytes; out->pass_flop = Q.pass_flop;
}


// A new block around the current scores: the best Tk candidates over all shards (the pick `gidx` is the first of them), their
// covariance columns from ONE pass over every shard's K_ZX.  words = 4 + d + chain entries (the width of a record right now).
static int32_t qei_build_block(const abo::QeiShards& S, double xi, double best_y, int Tk, int words, int64_t gidx) {
    const int n = S.n, d = gp_dim(S.gp[0]);
    const int wt = Tk * words;
    int32_t rc = S.run([&](int i) -> int32_t { return abo::qei_top(S.gp[i], S.cd[i], xi, best_y, S.lo[i], Tk, S.rec(i)); });
    if (rc) return rc;
    std::vector<double> blocks((size_t)n * wt);
    rc = S.gather((size_t)wt, blocks.data());
    if (rc) return rc;
    struct Ent { uint64_t key; int64_t idx; const double* r; };
    std::vector<Ent> ents;
    for (int e = 0; e < n * Tk; ++e) {
        const double* r = &blocks[(size_t)e * words];
        if (r[1] >= 0) ents.push_back({host_score_key(r[0]), (int64_t)r[1], r});
    }
    std::sort(ents.begin(), ents.end(), [](const Ent& a, const Ent& b) { return a.key > b.key || (a.key == b.key && a.idx < b.idx); });
    const int Tb = (int)ents.size() < Tk ? (int)ents.size() : Tk;
    std::vector<double> pts((size_t)Tk * d);
    std::vector<int64_t> gix(Tk);
    bool found = false;
    for (int t = 0; t < Tb; ++t) {
        gix[t] = ents[t].idx;
        memcpy(&pts[(size_t)t * d], ents[t].r + 4, sizeof(double) * d);
        found = found || ents[t].idx == gidx;
    }
    if (!found) return fail(ABO_EINVAL, "q-EI: internal error: the pick is not among the best %d", Tb);
    return S.run([&](int i) -> int32_t { return abo::qei_block(S.gp[i], S.cd[i], pts.data(), gix.data(), Tb); });
}

// The batch on ONE handle with the pick loop on the device (qei.hip: qei_step_kernel): launch k finishes pick k − 1 (record, slot
// look-up, γ, the rank-1 correction) and selects pick k — q + 1 launches and ONE read-back per batch where qei_drive makes five launches,
// a copy and a host synchronisation per pick (the kernels of a pick total ≈ 19 µs, the host round trip made it 47: VERDICT r05).  Only
// a pick outside every block (or a failed pivot) comes back to the host early: the remaining launches return at once, the host builds
// the block and resumes.  Same picks, same values, same chain as qei_drive with one shard, bit for bit (the step calls — what
// abstractbayesopt.jl_amd/incremental.py drives — are compared with this in tests/test_gpu_incremental.py).
static int32_t qei_drive_device(abo_gp* g, abo_cand* c, int q, double xi, double best_y, int distinct, int T, int64_t idx_base,
                                double* x_out, int64_t* idx_out, double* ei_out, int64_t* info) {
    if (T <= 0) T = qei_default_block();
    if (T > QEI_MAXT) T = QEI_MAXT;
    const int d = c->d;
    const int Tk = (int64_t)T < c->M ? T : (int)c->M;
    int32_t rc = abo::qei_begin(g, c, q, T, /*snapshot=*/false);
    if (rc) return rc;
    abo_cand::Qei& Q = c->qei;
    hipStream_t s = g->stream;
    std::string keep;
    bool launched = false, restored = false;               // launch 0 snapshots (μ, σ²), the tail launch rolls them back
    const int wmax = 4 + d + QEI_MAXQ;
    const size_t off_part = (sizeof(QeiStepState) + 255) / 256 * 256;
    const size_t off_rec = off_part + sizeof(QeiStepPartial) * 2 * QEI_STEP_MAXWG;
    const size_t rec_bytes = sizeof(double) * (size_t)q * wmax;
    std::vector<double> rec((size_t)q * wmax);
    QeiStepState hst{};
    do {
        if ((rc = qei_rebase(c, idx_base))) break;
        hipError_t e = c->qdev.ensure(off_rec + rec_bytes);
        if (e == hipSuccess) e = hipMemsetAsync(c->qdev.p, 0, sizeof(QeiStepState), s);
        if (e != hipSuccess) { rc = fail(ABO_EHIP, "abo_cand_qei: %s", hipGetErrorString(e)); break; }
        char* base = static_cast<char*>(c->qdev.p);
        QeiStepArgs a{};
        a.mu = c->mu.as<double>(); a.var = c->var.as<double>(); a.Z = c->Z.as<double>();
        a.snap_mu = c->qmu.as<double>(); a.snap_var = c->qvar.as<double>();
        a.blk = c->qblk.as<double>(); a.chain = c->qchain.as<double>();
        a.st = reinterpret_cast<QeiStepState*>(base);
        a.part = reinterpret_cast<QeiStepPartial*>(base + off_part);
        a.rec = reinterpret_cast<double*>(base + off_rec);
        a.M = c->M; a.Mp = Q.Mp; a.idx_base = idx_base; a.d = d; a.T16 = Q.T16; a.nslots = Q.nblk_cap * Q.T16; a.wmax = wmax;
        a.q = q; a.n0 = Q.nchain; a.distinct = distinct; a.xi = xi; a.best_y = best_y; a.noise = g->st->noise_used;
        for (int i = 0; i < Q.nchain; ++i) a.chain_s0[i] = Q.chain_s[i];
        const int nwg = (int)std::min<int64_t>((c->M + 255) / 256, QEI_STEP_MAXWG);
        a.nwg_prev = nwg;
        int k_from = 0, done = 0;                              // done: picks of the batch the chain holds already
        while (true) {
            for (int b = 0; b < 4; ++b) a.blk_base[b] = b < Q.nblk_cap ? Q.blk_base[b] : 0;
            for (int e2 = 0; e2 < 4 * QEI_MAXT; ++e2) a.slot_gidx[e2] = e2 < a.nslots ? Q.slot_gidx[e2] : -1;
            for (int k = k_from; k <= q && e == hipSuccess; ++k) { a.k = k; e = launch_qei_step(a, nwg, s); launched = true; }
            PinStage pin(g->ctx);
            if (e == hipSuccess) e = pin.d2h(&hst, a.st, sizeof(QeiStepState), s);
            if (e == hipSuccess) e = pin.d2h(rec.data(), a.rec, rec_bytes, s);
            if (e == hipSuccess) e = wait_stream(s);
            if (e != hipSuccess) { rc = fail(ABO_EHIP, "abo_cand_qei: %s", hipGetErrorString(e)); break; }
            pin.flush();
            // the picks conditioned on since the last read-back join the host's image of the chain (abo_cand_downdate compares a real
            // append's point with chain_x to find its column)
            const int cond = hst.stop ? hst.stop_at : q - 1;
            for (int t = done; t < cond; ++t) {
                const double* r = &rec[(size_t)t * wmax];
                ++Q.nchain;
                Q.chain_s.push_back(hst.s_batch[t]);
                Q.chain_x.insert(Q.chain_x.end(), r + 4, r + 4 + d);
            }
            done = cond;
            if (!hst.stop) { restored = true; break; }    // the tail launch ran: the stored posterior is the batch's starting point again
            const double* r = &rec[(size_t)hst.stop_at * wmax];
            if (hst.stop == 2) {                               // s = σ²(x) + σ²_n ≤ 0: the failed pivot of the plain loop's bordered append
                const long long at = (long long)(g->N + (Q.nchain - Q.nreal) + 1);
                if (info) *info = at;
                rc = fail(ABO_ENOTPD, "PosDefException: matrix is not positive definite; Cholesky factorization failed at %lld", at);
                break;
            }
            // pick stop_at lies in no block: a new block around the scores of this moment, then the launches from stop_at + 1 again
            abo_gp* gp1[1] = {g};
            abo_cand* cd1[1] = {c};
            const int64_t lo1[1] = {idx_base};
            abo::QeiShards S;
            S.n = 1; S.gp = gp1; S.cd = cd1; S.lo = lo1;
            S.run = [](const std::function<int32_t(int)>& f) { return f(0); };
            S.rec = [c](int) { return c->qrec.as<double>(); };
            S.gather = [g, c](size_t words, double* out) -> int32_t {
                HIPCHK(hipMemcpyAsync(out, c->qrec.p, sizeof(double) * words, hipMemcpyDeviceToHost, g->stream));
                HIPCHK(wait_stream(g->stream));
                return ABO_OK;
            };
            if ((rc = qei_build_block(S, xi, best_y, Tk, 4 + d + Q.nchain, (int64_t)r[1]))) break;
            e = hipMemsetAsync(&a.st->stop, 0, 2 * sizeof(int32_t), s);
            k_from = hst.stop_at + 1;
        }
        if (rc) break;
        for (int j = 0; j < q; ++j) {
            const double* r = &rec[(size_t)j * wmax];
            ei_out[j] = r[0];
            idx_out[j] = (int64_t)r[1];
            memcpy(x_out + (size_t)j * d, r + 4, sizeof(double) * d);
        }
    } while (false);
    if (rc) keep = g_err;
    int32_t r2 = ABO_OK;
    if (restored || !launched) c->qei.open = false;        // nothing to roll back (or nothing was touched)
    else r2 = abo::qei_end(g, c);                          // an error after launch 0: σ², μ back from the snapshot it took
    if (rc) return fail(rc, "%s", keep.c_str());
    return r2;
}

// The batch over n shards of one set.  S.run(f): f(i) on every shard (mgpu: the shard's worker thread); S.rec(i): the shard's
// record block in DEVICE memory (room for S.max_words doubles); S.gather(words, out): all shards' blocks → host, n × words doubles.
int32_t abo::qei_drive(const QeiShards& S, int q, double xi, double best_y, int distinct, int T, double* x_out, int64_t* idx_out,
                       double* ei_out, int64_t* info) {
    const int n = S.n, d = gp_dim(S.gp[0]);
    if (T <= 0) T = qei_default_block();
    if (T > QEI_MAXT) T = QEI_MAXT;
    int64_t Mtot = 0;
    for (int i = 0; i < n; ++i) Mtot += S.cd[i]->M;
    if (Mtot < 1) return fail(ABO_EINVAL, "q-EI: the candidate set is empty");
    const int Tk = (int64_t)T < Mtot ? T : (int)Mtot;
    int32_t rc = S.run([&](int i) -> int32_t { return abo::qei_begin(S.gp[i], S.cd[i], q, T); });
    std::string keep;
    if (rc) keep = g_err;
    std::vector<double> blocks;
    int nch_done = rc ? 0 : S.cd[0]->qei.nchain;           // the real entries the state carries over
    for (int j = 0; j < q && !rc; ++j) {
        const int nch = nch_done, words = 4 + d + nch;
        rc = S.run([&](int i) -> int32_t { return abo::qei_top(S.gp[i], S.cd[i], xi, best_y, S.lo[i], 1, S.rec(i)); });
        if (rc) break;
        blocks.resize((size_t)n * words);
        rc = S.gather((size_t)words, blocks.data());
        if (rc) break;
        int win = -1;
        for (int i = 0; i < n; ++i) {                          // sortperm(scores; rev=true)[1] over the shards' winners
            const double* r = &blocks[(size_t)i * words];
            if (r[1] < 0) continue;                            // empty shard
            if (win < 0) { win = i; continue; }
            const double* w = &blocks[(size_t)win * words];
            const uint64_t kr = host_score_key(r[0]), kw = host_score_key(w[0]);
            if (kr > kw || (kr == kw && r[1] < w[1])) win = i;
        }
        if (win < 0) { rc = fail(ABO_EINVAL, "q-EI: the candidate set is empty"); break; }
        const std::vector<double> w(blocks.begin() + (size_t)win * words, blocks.begin() + (size_t)(win + 1) * words);
        const int64_t gidx = (int64_t)w[1];
        ei_out[j] = w[0];
        idx_out[j] = gidx;
        memcpy(x_out + (size_t)j * d, &w[4], sizeof(double) * d);
        if (j == q - 1) break;                                 // the last pick conditions nothing (the batch is rolled back)
        if (!abo::qei_has(S.cd[0], gidx)) {
            rc = qei_build_block(S, xi, best_y, Tk, words, gidx);
            if (rc) break;
        }
        rc = S.run([&](int i) -> int32_t {
            const int64_t ex = (distinct && gidx >= S.lo[i] && gidx < S.lo[i] + S.cd[i]->M) ? gidx - S.lo[i] : -1;
            int64_t inf = 0;
            const int32_t r = abo::qei_pick(S.gp[i], S.cd[i], gidx, w[3], &w[4 + d], nch, ex, &inf);
            if (inf && info && i == 0) *info = inf;
            return r;
        });
        if (rc) break;
        ++nch_done;
    }
    if (rc && keep.empty()) keep = g_err;
    const int32_t r2 = S.run([&](int i) -> int32_t { return abo::qei_end(S.gp[i], S.cd[i]); });
    if (rc) return fail(rc, "%s", keep.c_str());
    return r2;
}

// ---- optimize_acquisition on the device (acq_utils.jl:33-73): refinement launch + the one-call driver -----------------
namespace {

struct RefineOpts { int max_iter, ls_max, history; double g_tol, f_abstol, x_abstol; };

RefineOpts refine_defaults(const abo_refine_opts* o) {
    RefineOpts r{100, 20, 10, 1e-5, 2.2e-9, 1e-4};        // acq_utils.jl:10, :62; Optim's LBFGS keeps m = 10 pairs
    if (o) {
        // clamped: "no limit" spelled as INT32_MAX must not overflow the round budget max_iter × linesearch_max (refine.hip) nor
        // size the curvature-pair storage
        if (o->max_iter > 0) r.max_iter = o->max_iter < 10000 ? o->max_iter : 10000;
        if (o->linesearch_max > 0) r.ls_max = o->linesearch_max < 64 ? o->linesearch_max : 64;
        if (o->history > 0) r.history = o->history < 64 ? o->history : 64;
        if (o->g_tol > 0.0) r.g_tol = o->g_tol;
        if (o->f_abstol > 0.0) r.f_abstol = o->f_abstol;
        if (o->x_abstol > 0.0) r.x_abstol = o->x_abstol;
    }
    return r;
}

int32_t check_refinable(abo_gp* g, int32_t d, const char* fn) {
    int32_t rc = check_fitted(g, d);
    if (rc) return rc;
    if (g->p_out > 1 && d + 1 != g->p_out) return fail(ABO_EINVAL, "%s: gradient-enhanced handle with p = %d outputs and d = %d", fn, g->p_out, d);
    return ABO_OK;
}

hipError_t grad_eval_cb(void* ctx, const double* pts, int npts, double* mu, double* cov, hipStream_t) {
    abo_gp* g = static_cast<abo_gp*>(ctx);
    // (a failure inside is reported through the calling thread's error slot; the launcher only needs "not hipSuccess")
    return grad_eval_device(g, pts, npts, 0.0, mu, cov, nullptr) == ABO_OK ? hipSuccess : hipErrorUnknown;
}

// bounds (2·d doubles: lower, upper), starts (S·d) and the outputs all in DEVICE memory; asynchronous on the handle's stream
// between events 5 and 6
int32_t refine_device(abo_gp* g, const AcqTerms& terms, const double* bounds_d, const double* starts_d, int S,
                      RefineOpts o, double* x_out_d, double* f_out_d, int* iters_d, int grad_only) {
    hipStream_t s = g->stream;
    RefineArgs ra{};
    ra.Xs = g->st->Xs.as<double>(); ra.W = g->st->W.as<double>(); ra.WT = g->st->WT.as<double>(); ra.alpha = g->alpha.as<double>();
    ra.ld = g->st->cap; ra.N = (int)g->N; ra.Np = (int)g->Np; ra.d = g->d; ra.dp = g->dp; ra.family = g->prm.family;
    ra.s = 1.0 / g->prm.ell; ra.sigma_f2 = g->prm.sigma_f2; ra.mean_c = g->prm.mean_c;
    ra.terms = terms;
    ra.lower = bounds_d; ra.upper = bounds_d ? bounds_d + g->d : nullptr; ra.starts = starts_d;
    ra.x_out = x_out_d; ra.f_out = f_out_d; ra.iters_out = iters_d;
    // the L-BFGS state of a start lives in its workgroup's LDS (64 KiB without opting into more): fewer pairs for very wide inputs
    int m = o.history;
    while (m > 0 && refine_lds_bytes(g->d, g->dp, m) > 65536) --m;
    if (refine_lds_bytes(g->d, g->dp, m) > 65536)
        return fail(ABO_EINVAL, "abo_refine: input dimension %d too large for the on-device refinement (library limit: %d)", g->d, 700);
    ra.max_iter = o.max_iter; ra.ls_max = o.ls_max; ra.history = m; ra.g_tol = o.g_tol; ra.f_abstol = o.f_abstol; ra.x_abstol = o.x_abstol;
    if (g->p_out > 1) {
        // gradient-enhanced model: lockstep rounds over the all-output posterior (refine.hip: launch_refine_lockstep_grad) — value
        // and gradient of the function-value terms come out of ONE mean / covariance-block evaluation per point
        const bool stencil = terms_have_gradnorm(terms);
        const size_t wb = refine_lockstep_grad_bytes(S, g->d, m, stencil);
        HIPCHK(g->T.ensure(wb + sizeof(double) * MAX_P));
        double* mean_g = reinterpret_cast<double*>(static_cast<char*>(g->T.p) + wb);
        HIPCHK(hipMemcpyAsync(mean_g, g->mean_vec + 1, sizeof(double) * g->d, hipMemcpyHostToDevice, s));
        GradEval ev{g, grad_eval_cb};
        const hipError_t e = grad_only ? launch_acq_grad_via_eval(ra, S, mean_g, g->T.p, ev, s)
                                       : launch_refine_lockstep_grad(ra, S, mean_g, g->T.p, ev, s);
        if (e == hipErrorUnknown) return ABO_EHIP;            // grad_eval_device has set the error text
        HIPCHK(e);
        return ABO_OK;
    }
    // from 1024 factor rows on the starts advance in lockstep rounds whose evaluations are batched on the MFMA tile core (L⁻¹ read once
    // per round instead of once per start and evaluation; measured crossover, profiles/r03_optimize_acquisition_latency.txt: N = 500
    // 2.56 ms one launch / 2.96 lockstep, N = 1024 6.98 / 4.16); ABO_REFINE_LOCKSTEP_NP moves the switch (0 = never)
    const char* lke = getenv("ABO_REFINE_LOCKSTEP_NP");
    const long lock_np = lke ? atol(lke) : 1024L;
    if (!grad_only && lock_np > 0 && g->Np >= lock_np) {
        HIPCHK(g->T.ensure(refine_lockstep_bytes(S, (int)g->Np, g->d, m)));
        HIPCHK(launch_refine_lockstep(ra, S, g->T.p, s));
        return ABO_OK;
    }
    HIPCHK(g->T.ensure(sizeof(double) * 4 * (size_t)g->Np * (size_t)S));      // per-start scratch (k, κ', v, u); T is free after the fit
    ra.scratch = g->T.as<double>();
    HIPCHK(launch_refine(ra, S, grad_only, s));
    return ABO_OK;
}

// the best refined point, the reference's way (acq_utils.jl:66-72: strict `>` keeps the first maximum); the best grid point if no
// refined value reaches its score
void pick_best(const double* starts_x, const double* starts_val, const double* rx, const double* rf, int k, int d, double* best_x,
               double* best_val) {
    int j = -1;
    for (int e = 0; e < k; ++e) {
        const double v = rf[e];
        if (!(v == v) || std::fabs(v) > 1.0e300) continue;
        if (j < 0 || v > rf[j]) j = e;
    }
    const double* src = starts_x;
    double val = k > 0 ? starts_val[0] : std::numeric_limits<double>::quiet_NaN();
    if (j >= 0 && (rf[j] >= val || !(val == val))) { src = rx + (size_t)j * d; val = rf[j]; }
    if (best_x) for (int c = 0; c < d; ++c) best_x[c] = k > 0 ? src[c] : std::numeric_limits<double>::quiet_NaN();
    if (best_val) *best_val = val;
}

}  // namespace

void abo::pick_best_point(const double* starts_x, const double* starts_val, const double* rx, const double* rf, int k, int d,
                          double* best_x, double* best_val) {
    pick_best(starts_x, starts_val, rx, rf, k, d, best_x, best_val);
}

extern "C" {

static int32_t refine_terms_impl(abo_gp* g, const AcqTerms& terms, const double* lower, const double* upper, int32_t d,
                                 const double* starts, int32_t S, const abo_refine_opts* opts, double* x_out, double* f_out,
                                 int32_t* iters_out) {
    if (S < 0 || (S > 0 && (!starts || !x_out || !f_out)) || !lower || !upper) return fail(ABO_EINVAL, "abo_refine: bad argument");
    for (int c = 0; c < d; ++c)
        if (!(lower[c] <= upper[c])) return fail(ABO_EINVAL, "abo_refine: lower[%d] > upper[%d]", c, c);
    if (S == 0) return ABO_OK;
    HIPCHK(hipSetDevice(g->prm.device));
    hipStream_t s = g->stream;
    const size_t nb = 2 * (size_t)d, ns = (size_t)S * d;
    // one staging block: bounds | starts | x_out | f_out | iters
    HIPCHK(g->Zdev.ensure(sizeof(double) * (nb + 2 * ns + S) + sizeof(int) * 2 * S));
    double* base = g->Zdev.as<double>();
    double *bd = base, *sd = base + nb, *xd = sd + ns, *fd = xd + ns;
    int* id = reinterpret_cast<int*>(fd + S);
    HIPCHK(hipMemcpyAsync(bd, lower, sizeof(double) * d, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(bd + d, upper, sizeof(double) * d, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(sd, starts, sizeof(double) * ns, hipMemcpyHostToDevice, s));
    HIPCHK(g->events(8));
    HIPCHK(hipEventRecord(g->evs()[5], s));
    int32_t rc = refine_device(g, terms, bd, sd, S, refine_defaults(opts), xd, fd, id, 0);
    if (rc) return rc;
    HIPCHK(hipEventRecord(g->evs()[6], s));
    std::vector<int> it(2 * (size_t)S);
    PinStage pin(g->ctx);
    HIPCHK(pin.d2h(x_out, xd, sizeof(double) * ns, s));
    HIPCHK(pin.d2h(f_out, fd, sizeof(double) * S, s));
    HIPCHK(pin.d2h(it.data(), id, sizeof(int) * 2 * S, s));
    HIPCHK(wait_stream(s));
    pin.flush();
    g->tm.refine_ms = ev_ms(g->evs()[5], g->evs()[6]);
    g->tm.refine_starts = S;
    g->tm.refine_evals = 0;
    for (int e = 0; e < S; ++e) g->tm.refine_evals += it[2 * e + 1];
    if (iters_out) for (int e = 0; e < 2 * S; ++e) iters_out[e] = it[e];
    return ABO_OK;
}

int32_t abo_refine_terms(abo_gp* g, const abo_acq_term* terms, int32_t nterms, const double* lower, const double* upper, int32_t d,
                         const double* starts, int32_t S, const abo_refine_opts* opts, double* x_out, double* f_out, int32_t* iters_out) {
    if (!g) return fail(ABO_EINVAL, "abo_refine_terms: null handle");
    int32_t rc = check_refinable(g, d, "abo_refine_terms");
    if (rc) return rc;
    AcqTerms t{};
    rc = make_terms(g, terms, nterms, &t, "abo_refine_terms");
    if (rc) return rc;
    return refine_terms_impl(g, t, lower, upper, d, starts, S, opts, x_out, f_out, iters_out);
}

int32_t abo_refine(abo_gp* g, int32_t kind, double p0, double best_y, const double* lower, const double* upper, int32_t d,
                   const double* starts, int32_t S, const abo_refine_opts* opts, double* x_out, double* f_out, int32_t* iters_out) {
    if (!g) return fail(ABO_EINVAL, "abo_refine: null handle");
    int32_t rc = check_refinable(g, d, "abo_refine");
    if (rc) return rc;
    const abo_acq_term one{kind, 0, p0, best_y, 1.0};
    AcqTerms t{};
    rc = make_terms(g, &one, 1, &t, "abo_refine");
    if (rc) return rc;
    return refine_terms_impl(g, t, lower, upper, d, starts, S, opts, x_out, f_out, iters_out);
}

#ifdef ABO_TEST_HOOKS
int32_t abo_test_acq_grad_terms(abo_gp* g, const abo_acq_term* terms, int32_t nterms, const double* Z, int64_t M, int32_t d, double* f,
                                double* grad) {
    if (!g) return fail(ABO_EINVAL, "abo_test_acq_grad: null handle");
    int32_t rc = check_refinable(g, d, "abo_test_acq_grad");
    if (rc) return rc;
    AcqTerms t{};
    rc = make_terms(g, terms, nterms, &t, "abo_test_acq_grad");
    if (rc) return rc;
    if (M < 0 || M > 65535 || (M > 0 && (!Z || !f || !grad))) return fail(ABO_EINVAL, "abo_test_acq_grad: bad argument (M ≤ 65535)");
    if (M == 0) return ABO_OK;
    HIPCHK(hipSetDevice(g->prm.device));
    hipStream_t s = g->stream;
    const size_t ns = (size_t)M * d;
    HIPCHK(g->Zdev.ensure(sizeof(double) * (2 * ns + M)));
    double *sd = g->Zdev.as<double>(), *xd = sd + ns, *fd = xd + ns;
    HIPCHK(hipMemcpyAsync(sd, Z, sizeof(double) * ns, hipMemcpyHostToDevice, s));
    rc = refine_device(g, t, nullptr, sd, (int)M, refine_defaults(nullptr), xd, fd, nullptr, 1);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(grad, xd, sizeof(double) * ns, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(f, fd, sizeof(double) * M, hipMemcpyDeviceToHost, s));
    HIPCHK(wait_stream(s));
    return ABO_OK;
}

int32_t abo_test_acq_grad(abo_gp* g, int32_t kind, double p0, double best_y, const double* Z, int64_t M, int32_t d, double* f,
                          double* grad) {
    const abo_acq_term one{kind, 0, p0, best_y, 1.0};
    return abo_test_acq_grad_terms(g, &one, 1, Z, M, d, f, grad);
}
#endif  // ABO_TEST_HOOKS

}  // extern "C"

// The grid stage on one handle (acq_utils.jl:44-52): n-point Latin hypercube generated on the device (rows j0 … j0+count−1 of it),
// scored under `t`, the k best (score, global index) pairs left in DEVICE memory (tv, ti) and — when sd is given — their
// coordinates gathered to sd [k][d].  bounds_d: device {lower[d], upper[d]}.  grid: scratch of count·d doubles.  Synchronises.
static int32_t grid_stage_device(abo_gp* g, const AcqTerms& t, const double* bounds_d, int d, int64_t n, int64_t j0, int64_t count,
                                 uint64_t seed, int k, double* grid, double* tv, int64_t* ti, double* sd) {
    hipStream_t s = g->stream;
    HIPCHK(launch_lhs(grid, n, d, bounds_d, bounds_d + d, seed, j0, count, s));
    int32_t rc = acq_terms_impl(g, grid, count, d, ABO_DEVICE, t, j0, nullptr, ABO_DEVICE, k, tv, ti, ABO_DEVICE);
    if (rc) return rc;
    if (sd) HIPCHK(launch_gather_points(grid, ti, j0, k, d, sd, s));
    return ABO_OK;
}

static int32_t optimize_terms_impl(abo_gp* g, const AcqTerms& t, const double* lower, const double* upper, int32_t d, int64_t n_grid,
                                   int32_t n_local, uint64_t seed, const abo_refine_opts* opts, double* best_x, double* best_val,
                                   double* starts_x, double* starts_val, double* refined_x, double* refined_val) {
    if (!lower || !upper || !best_x) return fail(ABO_EINVAL, "abo_optimize_acquisition: null argument");
    if (n_grid < 1 || n_local < 1) return fail(ABO_EINVAL, "abo_optimize_acquisition: n_grid and n_local must be positive");
    for (int c = 0; c < d; ++c)
        if (!(lower[c] <= upper[c])) return fail(ABO_EINVAL, "abo_optimize_acquisition: lower[%d] > upper[%d]", c, c);
    HIPCHK(hipSetDevice(g->prm.device));
    hipStream_t s = g->stream;
    const int k = (int)(n_local < n_grid ? n_local : n_grid);
    const size_t nb = 2 * (size_t)d, ns = (size_t)k * d;
    // device block of this call: bounds | starts | refined x | refined f | start scores | start indices | iteration counts
    ScratchBuf blk(g->prm.device, s), grid(g->prm.device, s);
    HIPCHK(blk.b.ensure(sizeof(double) * (nb + 2 * ns + 2 * (size_t)k) + sizeof(int64_t) * k + sizeof(int) * 2 * k));
    HIPCHK(grid.b.ensure(sizeof(double) * (size_t)n_grid * d));
    double* bd = blk.b.as<double>();
    double *sd = bd + nb, *xd = sd + ns, *fd = xd + ns, *tv = fd + k;
    int64_t* ti = reinterpret_cast<int64_t*>(tv + k);
    int* id = reinterpret_cast<int*>(ti + k);
    HIPCHK(hipMemcpyAsync(bd, lower, sizeof(double) * d, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(bd + d, upper, sizeof(double) * d, hipMemcpyHostToDevice, s));
    // grid stage (acq_utils.jl:44-52): Latin hypercube on the device, fused scores, stable reverse sort's first k
    int32_t rc = grid_stage_device(g, t, bd, d, n_grid, 0, n_grid, seed, k, grid.b.as<double>(), tv, ti, sd);
    if (rc) return rc;
    const double grid_ms = g->tm.acq_total_ms;
    // refinement stage (:55-71): one launch
    HIPCHK(g->events(10));
    HIPCHK(hipEventRecord(g->evs()[8], s));
    rc = refine_device(g, t, bd, sd, k, refine_defaults(opts), xd, fd, id, 0);
    if (rc) return rc;
    HIPCHK(hipEventRecord(g->evs()[9], s));
    std::vector<double> hs(ns), hv(k), hx(ns), hf(k);
    std::vector<int> it(2 * (size_t)k);
    PinStage pin(g->ctx);
    HIPCHK(pin.d2h(hs.data(), sd, sizeof(double) * ns, s));
    HIPCHK(pin.d2h(hv.data(), tv, sizeof(double) * k, s));
    HIPCHK(pin.d2h(hx.data(), xd, sizeof(double) * ns, s));
    HIPCHK(pin.d2h(hf.data(), fd, sizeof(double) * k, s));
    HIPCHK(pin.d2h(it.data(), id, sizeof(int) * 2 * k, s));
    HIPCHK(wait_stream(s));
    pin.flush();
    g->tm.acq_total_ms = grid_ms;
    g->tm.refine_ms = ev_ms(g->evs()[8], g->evs()[9]);
    g->tm.refine_starts = k;
    g->tm.refine_evals = 0;
    for (int e = 0; e < k; ++e) g->tm.refine_evals += it[2 * e + 1];
    pick_best(hs.data(), hv.data(), hx.data(), hf.data(), k, d, best_x, best_val);
    if (starts_x) memcpy(starts_x, hs.data(), sizeof(double) * ns);
    if (starts_val) memcpy(starts_val, hv.data(), sizeof(double) * k);
    if (refined_x) memcpy(refined_x, hx.data(), sizeof(double) * ns);
    if (refined_val) memcpy(refined_val, hf.data(), sizeof(double) * k);
    return ABO_OK;
}

// internal (mgpu.hip): a shard's part of the grid stage with the selection left on the device
int32_t abo::acq_lhs_shard(abo_gp* g, const abo_acq_term* terms, int32_t nterms, int64_t n, int32_t d, const double* lower,
                           const double* upper, uint64_t seed, int64_t j0, int64_t count, int32_t k, double* grid_d, double* tv_d,
                           int64_t* ti_d) {
    int32_t rc = check_fitted(g, d);
    if (rc) return rc;
    AcqTerms t{};
    rc = make_terms(g, terms, nterms, &t, "abo_mgpu_acq_lhs");
    if (rc) return rc;
    HIPCHK(hipSetDevice(g->prm.device));
    hipStream_t s = g->stream;
    ScratchBuf bnd(g->prm.device, s);
    HIPCHK(bnd.b.ensure(sizeof(double) * 2 * (size_t)d));
    double* bd = bnd.b.as<double>();
    HIPCHK(hipMemcpyAsync(bd, lower, sizeof(double) * d, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(bd + d, upper, sizeof(double) * d, hipMemcpyHostToDevice, s));
    if (count > 0) HIPCHK(launch_lhs(grid_d, n, d, bd, bd + d, seed, j0, count, s));
    return acq_terms_impl(g, grid_d, count, d, ABO_DEVICE, t, j0, nullptr, ABO_DEVICE, k, tv_d, ti_d, ABO_DEVICE);
}

int32_t abo::refine_terms(abo_gp* g, const abo_acq_term* terms, int32_t nterms, const double* lower, const double* upper, int32_t d,
                          const double* starts, int32_t S, const abo_refine_opts* opts, double* x_out, double* f_out) {
    return abo_refine_terms(g, terms, nterms, lower, upper, d, starts, S, opts, x_out, f_out, nullptr);
}

extern "C" {

int32_t abo_optimize_acquisition_terms(abo_gp* g, const abo_acq_term* terms, int32_t nterms, const double* lower, const double* upper,
                                       int32_t d, int64_t n_grid, int32_t n_local, uint64_t seed, const abo_refine_opts* opts,
                                       double* best_x, double* best_val, double* starts_x, double* starts_val, double* refined_x,
                                       double* refined_val) {
    if (!g) return fail(ABO_EINVAL, "abo_optimize_acquisition_terms: null handle");
    int32_t rc = check_refinable(g, d, "abo_optimize_acquisition_terms");
    if (rc) return rc;
    AcqTerms t{};
    rc = make_terms(g, terms, nterms, &t, "abo_optimize_acquisition_terms");
    if (rc) return rc;
    return optimize_terms_impl(g, t, lower, upper, d, n_grid, n_local, seed, opts, best_x, best_val, starts_x, starts_val, refined_x,
                               refined_val);
}

int32_t abo_optimize_acquisition(abo_gp* g, int32_t kind, double p0, double best_y, const double* lower, const double* upper,
                                 int32_t d, int64_t n_grid, int32_t n_local, uint64_t seed, const abo_refine_opts* opts,
                                 double* best_x, double* best_val, double* starts_x, double* starts_val, double* refined_x,
                                 double* refined_val) {
    const abo_acq_term one{kind, 0, p0, best_y, 1.0};
    return abo_optimize_acquisition_terms(g, &one, 1, lower, upper, d, n_grid, n_local, seed, opts, best_x, best_val, starts_x,
                                          starts_val, refined_x, refined_val);
}

// the grid stage alone on ONE handle: what abo_mgpu_acq_lhs is for a group (the Julia shim's grid_stage on a single-device model)
int32_t abo_acq_lhs(abo_gp* g, int64_t n, int32_t d, const double* lower, const double* upper, uint64_t seed, int32_t kind, double p0,
                    double best_y, int32_t k, double* top_val, int64_t* top_idx, double* top_x) {
    if (!g) return fail(ABO_EINVAL, "abo_acq_lhs: null handle");
    int32_t rc = check_fitted(g, d);
    if (rc) return rc;
    if (n < 1 || !lower || !upper) return fail(ABO_EINVAL, "abo_acq_lhs: bad grid");
    if (k < 1 || !top_val || !top_idx) return fail(ABO_EINVAL, "abo_acq_lhs: needs k >= 1, top_val and top_idx");
    const abo_acq_term one{kind, 0, p0, best_y, 1.0};
    AcqTerms t{};
    rc = make_terms(g, &one, 1, &t, "abo_acq_lhs");
    if (rc) return rc;
    HIPCHK(hipSetDevice(g->prm.device));
    hipStream_t s = g->stream;
    ScratchBuf blk(g->prm.device, s), grid(g->prm.device, s);
    const size_t nb = 2 * (size_t)d, ns = (size_t)k * d;
    HIPCHK(blk.b.ensure(sizeof(double) * (nb + ns + (size_t)k) + sizeof(int64_t) * k));
    HIPCHK(grid.b.ensure(sizeof(double) * (size_t)n * d));
    double* bd = blk.b.as<double>();
    double *sd = bd + nb, *tv = sd + ns;
    int64_t* ti = reinterpret_cast<int64_t*>(tv + k);
    HIPCHK(hipMemcpyAsync(bd, lower, sizeof(double) * d, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(bd + d, upper, sizeof(double) * d, hipMemcpyHostToDevice, s));
    rc = grid_stage_device(g, t, bd, d, n, 0, n, seed, k, grid.b.as<double>(), tv, ti, top_x ? sd : nullptr);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(top_val, tv, sizeof(double) * k, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(top_idx, ti, sizeof(int64_t) * k, hipMemcpyDeviceToHost, s));
    if (top_x) HIPCHK(hipMemcpyAsync(top_x, sd, sizeof(double) * ns, hipMemcpyDeviceToHost, s));
    HIPCHK(wait_stream(s));
    if (top_x) {
        const double nan = std::numeric_limits<double>::quiet_NaN();
        for (int e = 0; e < k; ++e)
            if (top_idx[e] < 0) for (int c = 0; c < d; ++c) top_x[(size_t)e * d + c] = nan;
    }
    return ABO_OK;
}

}  // extern "C"

bool abo::exiting() { return g_exiting.load(); }
void abo::arm_exit_guard() {
    // registered AFTER the first successful device call: atexit handlers run in reverse order of registration, so this one
    // runs before the HIP runtime's own teardown (registered when the runtime initialised)
    if (!g_exit_armed.exchange(true)) std::atexit(exit_hook);
}
void abo::at_exit(void (*f)()) { std::lock_guard<std::mutex> lk(g_exit_mu); g_exit_hooks.push_back(f); }
// only the two codes the runtime reserves for "torn down": hipErrorNotInitialized / hipErrorInvalidContext can be spurious mid-run
// errors of one call, and latching the process-wide flag on those would turn every later destroy into a leak
bool abo::gone(hipError_t e) { return e == hipErrorDeinitialized || e == hipErrorContextIsDestroyed; }

// internal accessors for the multi-device driver (mgpu.hip)
hipStream_t abo::gp_stream(abo_gp* g) { return g->stream; }
int abo::gp_device(const abo_gp* g) { return g->prm.device; }
const abo_params& abo::gp_params(const abo_gp* g) { return g->prm; }
int abo::gp_dim(const abo_gp* g) { return g->fitted ? g->d : 0; }
const double* abo::cand_points(const abo_cand* c) { return c->Z.as<double>(); }
const double* abo::cand_mu(const abo_cand* c) { return c->mu.as<double>(); }
int64_t abo::cand_size(const abo_cand* c) { return c->M; }
int32_t abo::set_error(int32_t code, const char* text) { return fail(code, "%s", text); }
const char* abo::last_error_text() { return g_err; }

extern "C" {

int32_t abo_cand_save(abo_gp* g, abo_cand* c) {
    if (!g || !c) return fail(ABO_EINVAL, "abo_cand_save: null argument");
    HIPCHK(hipSetDevice(g->prm.device));
    const size_t bytes = sizeof(double) * (c->M > 0 ? c->M : 1);
    HIPCHK(c->mu_bak.ensure(bytes));
    HIPCHK(c->var_bak.ensure(bytes));
    HIPCHK(hipMemcpyAsync(c->mu_bak.p, c->mu.p, sizeof(double) * c->M, hipMemcpyDeviceToDevice, g->stream));
    HIPCHK(hipMemcpyAsync(c->var_bak.p, c->var.p, sizeof(double) * c->M, hipMemcpyDeviceToDevice, g->stream));
    HIPCHK(wait_stream(g->stream));
    c->bak_gen = c->synced_gen; c->bak_N = c->synced_N;
    return ABO_OK;
}

int32_t abo_cand_restore(abo_gp* g, abo_cand* c) {
    if (!g || !c) return fail(ABO_EINVAL, "abo_cand_restore: null argument");
    if (c->bak_N < 0) return fail(ABO_EINVAL, "abo_cand_restore: nothing saved");
    HIPCHK(hipSetDevice(g->prm.device));
    HIPCHK(hipMemcpyAsync(c->mu.p, c->mu_bak.p, sizeof(double) * c->M, hipMemcpyDeviceToDevice, g->stream));
    HIPCHK(hipMemcpyAsync(c->var.p, c->var_bak.p, sizeof(double) * c->M, hipMemcpyDeviceToDevice, g->stream));
    HIPCHK(wait_stream(g->stream));
    c->synced_gen = c->bak_gen; c->synced_N = c->bak_N;
    return ABO_OK;
}

int32_t abo_cand_get(abo_gp* g, abo_cand* c, double* mu, double* var, int32_t out_space) {
    if (!g || !c) return fail(ABO_EINVAL, "abo_cand_get: null argument");
    HIPCHK(hipSetDevice(g->prm.device));
    if (mu) { int32_t rc = copy_out(mu, c->mu.p, sizeof(double) * c->M, out_space, g->stream); if (rc) return rc; }
    if (var) { int32_t rc = copy_out(var, c->var.p, sizeof(double) * c->M, out_space, g->stream); if (rc) return rc; }
    HIPCHK(wait_stream(g->stream));
    return ABO_OK;
}

int32_t abo_cand_point(abo_gp* g, abo_cand* c, int64_t idx, double* x, double* mu, double* var) {
    if (!g || !c) return fail(ABO_EINVAL, "abo_cand_point: null argument");
    if (idx < 0 || idx >= c->M) return fail(ABO_EINVAL, "abo_cand_point: index %lld outside 0..%lld", (long long)idx, (long long)c->M - 1);
    HIPCHK(hipSetDevice(g->prm.device));
    hipStream_t s = g->stream;
    PinStage pin(g->ctx);
    if (x) HIPCHK(pin.d2h(x, c->Z.as<double>() + idx * c->d, sizeof(double) * c->d, s));
    if (mu) HIPCHK(pin.d2h(mu, c->mu.as<double>() + idx, sizeof(double), s));
    if (var) HIPCHK(pin.d2h(var, c->var.as<double>() + idx, sizeof(double), s));
    HIPCHK(wait_stream(s));
    pin.flush();
    return ABO_OK;
}

int32_t abo_cand_exclude(abo_gp* g, abo_cand* c, int64_t idx) {
    if (!g || !c) return fail(ABO_EINVAL, "abo_cand_exclude: null argument");
    if (idx < 0 || idx >= c->M) return fail(ABO_EINVAL, "abo_cand_exclude: index %lld outside 0..%lld", (long long)idx, (long long)c->M - 1);
    HIPCHK(hipSetDevice(g->prm.device));
    hipStream_t s = g->stream;
    const double excl[2] = {HUGE_VAL, 0.0};                 // μ = +Inf, σ² = 0: EI = PI = 0, UCB = −Inf
    HIPCHK(hipMemcpyAsync(c->mu.as<double>() + idx, &excl[0], sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->var.as<double>() + idx, &excl[1], sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(wait_stream(s));
    return ABO_OK;
}

// ---- greedy q-EI on one handle (include/abo_hip.h: "block form") ----------------------------------------------------------------
int32_t abo_set_qei_block(int32_t block) {
    if (block < 0 || block > QEI_MAXT) return fail(ABO_EINVAL, "abo_set_qei_block: block = %d outside 0..%d", block, QEI_MAXT);
    g_qei_block.store(block);
    return ABO_OK;
}

int32_t abo_cand_qei_begin(abo_gp* g, abo_cand* c, int32_t q, int32_t block) { return abo::qei_begin(g, c, q, block); }

int32_t abo_cand_qei_eligible(abo_gp* g, abo_cand* c, int32_t q, int32_t block, int32_t* ok) {
    if (!g || !c || !ok) return fail(ABO_EINVAL, "abo_cand_qei_eligible: null argument");
    const int T = block <= 0 ? qei_default_block() : block;
    *ok = 0;
    if (T < 1) { (void)fail(ABO_EINVAL, "q-EI, block form: block size 0 (the plain loop is a different call)"); return ABO_OK; }
    *ok = abo::qei_eligible(g, c, q) == ABO_OK ? 1 : 0;                                   // (the reason stays in abo_last_error)
    return ABO_OK;
}

int32_t abo_cand_qei_top(abo_gp* g, abo_cand* c, double xi, double best_y, int64_t idx_base, int32_t k, double* rec, int64_t cap_words) {
    if (!g || !c || !rec) return fail(ABO_EINVAL, "abo_cand_qei_top: null argument");
    if (k < 1 || k > 1024) return fail(ABO_EINVAL, "abo_cand_qei_top: k = %d outside 1..1024", k);
    const size_t words = (size_t)k * (4 + c->d + c->qei.nchain);
    if (cap_words < 0 || (size_t)cap_words < words)
        return fail(ABO_EINVAL, "abo_cand_qei_top: rec holds %lld doubles, %d records of 4 + d + %d chain values need %zu", (long long)cap_words, k,
                    c->qei.nchain, words);
    HIPCHK(hipSetDevice(g->prm.device));
    HIPCHK(c->qrec.ensure(sizeof(double) * words));
    int32_t rc = abo::qei_top(g, c, xi, best_y, idx_base, k, c->qrec.as<double>());
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(rec, c->qrec.p, sizeof(double) * words, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(wait_stream(g->stream));
    return ABO_OK;
}

int32_t abo_cand_qei_block(abo_gp* g, abo_cand* c, const double* pts, const int64_t* gidx, int32_t T) {
    return abo::qei_block(g, c, pts, gidx, T);
}

int32_t abo_cand_qei_pick(abo_gp* g, abo_cand* c, int64_t gidx, double var_x, const double* cx, int32_t n, int64_t excl, int64_t* info) {
    return abo::qei_pick(g, c, gidx, var_x, cx, n, excl, info);
}

int32_t abo_cand_qei_end(abo_gp* g, abo_cand* c) { return abo::qei_end(g, c); }

int32_t abo_cand_qei_has(abo_gp* g, abo_cand* c, int64_t gidx, int32_t* has, int32_t* nchain) {
    if (!g || !c) return fail(ABO_EINVAL, "abo_cand_qei_has: null argument");
    if (has) *has = (c->qei.N >= 0 && gidx >= 0) ? abo::qei_has(c, gidx) : 0;
    if (nchain) *nchain = c->qei.N >= 0 ? c->qei.nchain : 0;
    return ABO_OK;
}

int32_t abo_cand_qei_stats(abo_gp* g, abo_cand* c, abo_qei_stats* out) {
    if (!g || !c || !out) return fail(ABO_EINVAL, "abo_cand_qei_stats: null argument");
    abo::qei_get_stats(c, c->qei.nchain - c->qei.batch0 + 1, 0.0, out);
    return ABO_OK;
}

// the plain loop on one handle: q × [EI + arg-max, fantasy append, O(N·M) down-date], rolled back (what abo_mgpu_cand_qei's plain
// loop does per device)
static int32_t qei_plain(abo_gp* g, abo_cand* c, int q, double xi, double best_y, int distinct, int64_t idx_base, double* x_out,
                         int64_t* idx_out, double* ei_out, int64_t* info) {
    int32_t rc = check_fitted(g, c->d);
    if (rc) return rc;
    if (g->st->gen != c->synced_gen || g->N != c->synced_N)
        return fail(ABO_EINVAL, "abo_cand_qei: the candidate set is not in sync with this model (abo_cand_refresh / abo_cand_downdate)");
    HIPCHK(hipSetDevice(g->prm.device));
    const int d = c->d;
    const size_t bytes = sizeof(double) * (c->M > 0 ? c->M : 1);
    HIPCHK(c->qmu.ensure(bytes));
    HIPCHK(c->qvar.ensure(bytes));
    HIPCHK(hipMemcpyAsync(c->qmu.p, c->mu.p, sizeof(double) * c->M, hipMemcpyDeviceToDevice, g->stream));
    HIPCHK(hipMemcpyAsync(c->qvar.p, c->var.p, sizeof(double) * c->M, hipMemcpyDeviceToDevice, g->stream));
    HIPCHK(wait_stream(g->stream));
    const uint64_t gen0 = c->synced_gen;
    const int64_t N0 = c->synced_N;
    c->qei = abo_cand::Qei();
    abo_gp* cur = g;
    std::string keep;
    for (int j = 0; j < q && !rc; ++j) {
        double tv = 0.0, mu = 0.0;
        int64_t ti = -1;
        rc = abo::cand_acq_ex(cur, c, ABO_ACQ_EI, xi, best_y, idx_base, nullptr, ABO_DEVICE, 1, &tv, &ti, ABO_HOST);
        if (rc) break;
        if (ti < 0) { rc = fail(ABO_EINVAL, "abo_cand_qei: the candidate set is empty"); break; }
        double* x = x_out + (size_t)j * d;
        rc = abo_cand_point(cur, c, ti - idx_base, x, &mu, nullptr);
        if (rc) break;
        ei_out[j] = tv; idx_out[j] = ti;
        if (j == q - 1) break;
        abo_gp* nw = nullptr;
        if (cur->p_out > 1) {
            double yv[MAX_P];
            rc = abo_predict_grad(cur, x, 1, d, ABO_HOST, yv, nullptr, ABO_HOST);
            if (!rc) rc = abo_append_grad(cur, x, d, yv, info, &nw);
        } else {
            rc = abo_append(cur, x, d, mu, info, &nw);
        }
        if (rc) break;
        rc = abo_cand_downdate(nw, c);
        if (!rc && distinct) rc = abo_cand_exclude(nw, c, ti - idx_base);
        if (cur != g) abo_destroy(cur);
        cur = nw;
    }
    if (rc) keep = g_err;
    if (cur != g) abo_destroy(cur);
    hipError_t e = hipMemcpyAsync(c->mu.p, c->qmu.p, sizeof(double) * c->M, hipMemcpyDeviceToDevice, g->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(c->var.p, c->qvar.p, sizeof(double) * c->M, hipMemcpyDeviceToDevice, g->stream);
    if (e == hipSuccess) e = wait_stream(g->stream);
    c->synced_gen = gen0; c->synced_N = N0;
    if (rc) return fail(rc, "%s", keep.c_str());
    HIPCHK(e);
    return ABO_OK;
}

int32_t abo_cand_qei(abo_gp* g, abo_cand* c, int32_t q, double xi, double best_y, int32_t distinct, int64_t idx_base, int32_t block,
                     double* x_out, int64_t* idx_out, double* ei_out, abo_qei_stats* stats) {
    if (!g || !c || !x_out || !idx_out || !ei_out) return fail(ABO_EINVAL, "abo_cand_qei: null argument");
    if (q < 1) return fail(ABO_EINVAL, "abo_cand_qei: q = %d", q);
    const auto t0 = std::chrono::steady_clock::now();
    int64_t info = 0;
    int T = block == 0 ? qei_default_block() : block;
    int32_t rc;
    if (T > 0 && abo::qei_eligible(g, c, q) == ABO_OK) {
        abo_gp* gp1[1] = {g};
        abo_cand* cd1[1] = {c};
        const int64_t lo1[1] = {idx_base};
        HIPCHK(hipSetDevice(g->prm.device));
        HIPCHK(c->qrec.ensure(sizeof(double) * abo::qei_max_words(c->d, q, T)));
        abo::QeiShards S;
        S.n = 1; S.gp = gp1; S.cd = cd1; S.lo = lo1;
        S.run = [](const std::function<int32_t(int)>& f) { return f(0); };
        S.rec = [c](int) { return c->qrec.as<double>(); };
        S.gather = [g, c](size_t words, double* out) -> int32_t {
            HIPCHK(hipMemcpyAsync(out, c->qrec.p, sizeof(double) * words, hipMemcpyDeviceToHost, g->stream));
            HIPCHK(wait_stream(g->stream));
            return ABO_OK;
        };
        // one handle, at least one candidate: the pick loop stays on the device; an empty set goes through the shard driver (its
        // "the candidate set is empty" is the contract)
        if (c->M >= 1) rc = qei_drive_device(g, c, q, xi, best_y, distinct, T, idx_base, x_out, idx_out, ei_out, &info);
        else rc = abo::qei_drive(S, q, xi, best_y, distinct, T, x_out, idx_out, ei_out, &info);
    } else {
        (void)hipGetLastError();
        rc = qei_plain(g, c, q, xi, best_y, distinct, idx_base, x_out, idx_out, ei_out, &info);
    }
    if (stats) {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        abo::qei_get_stats(c, q, ms, stats);
    }
    return rc;
}

int32_t abo_pool_trim(int32_t device) {
    if (device < 0 || device > 15) return fail(ABO_EINVAL, "abo_pool_trim: bad device %d", device);
    if (g_exiting.load()) return ABO_OK;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipDeviceSynchronize());
    pool_trim(device);
    return ABO_OK;
}

int32_t abo_fill_distance(int32_t device, const double* X, int64_t N, int32_t d, int32_t x_space, const double* S, int64_t n_samples,
                          int32_t s_space, double* out) {
    if (!X || !S || !out) return fail(ABO_EINVAL, "abo_fill_distance: null argument");
    if (N < 1 || n_samples < 1 || d < 1 || d > 65536) return fail(ABO_EINVAL, "abo_fill_distance: bad sizes");
    if (device < 0 || device > 15) return fail(ABO_EINVAL, "abo_fill_distance: bad device %d", device);
    HIPCHK(hipSetDevice(device));
    ScratchBuf xb(device, nullptr), sb(device, nullptr), ob(device, nullptr);
    const double *Xd = X, *Sd = S;
    if (x_space != ABO_DEVICE) {
        HIPCHK(xb.b.ensure(sizeof(double) * (size_t)N * d));
        HIPCHK(hipMemcpyAsync(xb.b.p, X, sizeof(double) * (size_t)N * d, hipMemcpyHostToDevice, nullptr));
        Xd = xb.b.as<double>();
    }
    if (s_space != ABO_DEVICE) {
        HIPCHK(sb.b.ensure(sizeof(double) * (size_t)n_samples * d));
        HIPCHK(hipMemcpyAsync(sb.b.p, S, sizeof(double) * (size_t)n_samples * d, hipMemcpyHostToDevice, nullptr));
        Sd = sb.b.as<double>();
    }
    HIPCHK(ob.b.ensure(sizeof(double)));
    HIPCHK(launch_fill_distance(Xd, N, d, Sd, n_samples, ob.b.as<double>(), nullptr));
    HIPCHK(hipMemcpyAsync(out, ob.b.p, sizeof(double), hipMemcpyDeviceToHost, nullptr));
    HIPCHK(wait_stream(nullptr));
    return ABO_OK;
}

int32_t abo_lhs(int32_t device, int64_t n, int32_t d, const double* lower, const double* upper, uint64_t seed, int64_t j0,
                int64_t count, double* Z_dev) {
    if (!lower || !upper || !Z_dev) return fail(ABO_EINVAL, "abo_lhs: null argument");
    if (n < 1 || d < 1 || d > 65536 || j0 < 0 || count < 0 || j0 + count > n) return fail(ABO_EINVAL, "abo_lhs: bad sizes");
    HIPCHK(hipSetDevice(device));
    ScratchBuf sb(device, nullptr);
    DevBuf& b = sb.b;
    HIPCHK(b.ensure(sizeof(double) * 2 * d));
    HIPCHK(hipMemcpyAsync(b.p, lower, sizeof(double) * d, hipMemcpyHostToDevice, nullptr));
    HIPCHK(hipMemcpyAsync(b.as<double>() + d, upper, sizeof(double) * d, hipMemcpyHostToDevice, nullptr));
    HIPCHK(launch_lhs(Z_dev, n, d, b.as<double>(), b.as<double>() + d, seed, j0, count, nullptr));
    HIPCHK(wait_stream(nullptr));
    return ABO_OK;
}

int32_t abo_score(int32_t device, const double* mu, const double* var, int64_t M, int32_t kind, double p0, double best_y,
                  double* scores) {
    if (M < 0 || (M > 0 && (!mu || !var || !scores))) return fail(ABO_EINVAL, "abo_score: bad argument");
    if (kind < ABO_ACQ_EI || kind > ABO_ACQ_MEAN) return fail(ABO_EINVAL, "abo_score: unknown acquisition kind %d", kind);
    HIPCHK(hipSetDevice(device));
    HIPCHK(launch_score(mu, var, scores, M, kind, p0, best_y, nullptr));
    HIPCHK(wait_stream(nullptr));
    return ABO_OK;
}

#ifdef ABO_TEST_HOOKS
int32_t abo_test_oz_plan(int32_t n, int32_t* p, double* tables, double* scal, int32_t* eP) {
    if (!p || !tables || !scal || !eP) return fail(ABO_EINVAL, "abo_test_oz_plan: null argument");
    OzPlan pl;
    if (!oz_make_plan(n, &pl)) return fail(ABO_EINVAL, "abo_test_oz_plan: %d moduli not supported (8 … %d)", n, OZ_MAXMOD);
    for (int l = 0; l < n; ++l) {
        p[l] = pl.p[l];
        tables[0 * OZ_MAXMOD + l] = pl.invp[l];
        tables[1 * OZ_MAXMOD + l] = pl.c26[l];
        tables[2 * OZ_MAXMOD + l] = pl.s1[l];
        tables[3 * OZ_MAXMOD + l] = pl.s2[l];
    }
    scal[0] = pl.P1; scal[1] = pl.P2; scal[2] = pl.invP;
    *eP = pl.eP;
    return ABO_OK;
}

int32_t abo_test_oz_contract(int32_t device, const double* W, int64_t ldw, int32_t Np, int32_t nvalid, const double* Kxz, int64_t ldk,
                             int32_t Mc, double kmax, int32_t nmod, double* partial, int64_t ldp) {
    if (!W || !Kxz || !partial) return fail(ABO_EINVAL, "abo_test_oz_contract: null argument");
    if (Np <= 0 || Np % TB || Mc <= 0 || Mc % TB || nvalid < 0 || nvalid > Np || ldw < Np || ldk < Np || ldp < Mc || !(kmax > 0.0))
        return fail(ABO_EINVAL, "abo_test_oz_contract: Np and Mc multiples of 128, nvalid <= Np, leading dimensions >= Np / Mc, kmax > 0");
    OzPlan pl;
    if (!oz_make_plan(nmod, &pl)) return fail(ABO_EINVAL, "abo_test_oz_contract: %d moduli not supported", nmod);
    HIPCHK(hipSetDevice(device));
    const int64_t q = pad_up(Np, 256), mq = pad_up(Mc, 256);
    ScratchBuf wr(device, nullptr), kr(device, nullptr), u(device, nullptr), ints(device, nullptr);
    HIPCHK(wr.b.ensure(oz_w_bytes(nmod, Np)));
    HIPCHK(kr.b.ensure(oz_k_bytes(nmod, Np, Mc)));
    HIPCHK(u.b.ensure(oz_k_bytes(nmod, Np, Mc)));
    HIPCHK(ints.b.ensure(sizeof(int) * (2 * q + mq + OZ_CTR_INTS)));
    int* sexp = ints.b.as<int>();
    HIPCHK(oz_prepare_w(pl, W, ldw, Np, nvalid, wr.b.as<int8_t>(), sexp, sexp + q, nullptr));
    OzVarArgs oa{};
    oa.plan = &pl; oa.Kxz = Kxz; oa.ldk = ldk; oa.WR = wr.b.as<int8_t>(); oa.sexp = sexp; oa.bad_row = sexp + q;
    oa.KR = kr.b.as<int8_t>(); oa.U = u.b.as<int8_t>(); oa.bad_col = sexp + 2 * q; oa.partial = partial; oa.ldp = ldp;
    oa.Np = Np; oa.Mc = Mc; oa.nvalid = nvalid; oa.sK = oz_k_scale(kmax);
    HIPCHK(launch_var_ozaki(oa, nullptr));
    HIPCHK(wait_stream(nullptr));
    return ABO_OK;
}

int32_t abo_test_kappa(int32_t device, int32_t family, const double* d2, double* out, int64_t n) {
    if (!d2 || !out || n < 0) return fail(ABO_EINVAL, "abo_test_kappa: bad argument");
    if (family < ABO_KERNEL_SE || family > ABO_KERNEL_MATERN32) return fail(ABO_EINVAL, "abo_test_kappa: unknown family");
    HIPCHK(hipSetDevice(device));
    HIPCHK(launch_kappa_test(family, d2, out, n, nullptr));
    HIPCHK(wait_stream(nullptr));
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
    HIPCHK(wait_stream(nullptr));
    return ABO_OK;
}
#endif  // ABO_TEST_HOOKS

}  // extern "C"
