// Multi-device driver (abo_mgpu_*, include/abo_hip.h): one host process, one host thread per shard, replicated
// deterministic fit, candidates sharded contiguously, selection merged after ONE all-gather of k × (score, index)
// per device (RCCL over xGMI; host copies when RCCL cannot be used).
//
// Reference behaviour being served: `scores = acqf(surrogate, grid_points)` + `sortperm(scores; rev=true)[1:n_local]`
// (src/acquisition_functions/acq_utils.jl:50-52) — every candidate's score depends only on (model, that candidate)
// (src/surrogates/StandardGP.jl:361-379), so the batch shards with no data-path collective; the merge reproduces the
// stable reverse sort globally (descending score, ties → lowest index, NaN first).  The reference has no multi-device
// code of its own; BASELINE configs 4 and 5 define this path.
//
// No arithmetic lives here: every device runs the single-device entry points of api.hip on its own handle.
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: the functions are resolved with dlsym (no link-time dependency)

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <limits>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "abo_exchange.h"
#include "abo_internal.h"
#include "abo_kernels.h"

namespace {

constexpr int MAXDEV = 16;

int32_t failf(int32_t code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int32_t failf(int32_t code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return abo::set_error(code, buf);
}

// ---- RCCL, loaded at run time ---------------------------------------------------------------------------------
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;        // why it is not available
};

const Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        // the copy the process already has (a PyTorch host brings its own) comes first: two RCCLs in one process would
        // each spin up their own proxy threads and IPC state
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            x.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (x.handle) break;
        }
        for (int i = 0; !x.handle && i < 3; ++i) x.handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (!x.handle) { x.why = std::string("librccl.so.1 not loadable: ") + (dlerror() ? dlerror() : "?"); return x; }
        x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(dlsym(x.handle, "ncclCommInitAll"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(x.handle, "ncclCommDestroy"));
        x.CommAbort = reinterpret_cast<decltype(x.CommAbort)>(dlsym(x.handle, "ncclCommAbort"));
        x.AllGather = reinterpret_cast<decltype(x.AllGather)>(dlsym(x.handle, "ncclAllGather"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(x.handle, "ncclGetErrorString"));
        x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(dlsym(x.handle, "ncclGroupStart"));
        x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(dlsym(x.handle, "ncclGroupEnd"));
        if (!x.CommInitAll || !x.CommDestroy || !x.CommAbort || !x.AllGather || !x.GetErrorString || !x.GroupStart || !x.GroupEnd) {
            x.why = "librccl.so.1 lacks ncclCommInitAll / ncclCommAbort / ncclAllGather";
            x.handle = nullptr;
        }
        return x;
    }();
    return r;
}

// ---- worker threads: one per shard of a device list, owned by that list's CommSet ---------------------------------------
// Groups on the same device list (every `update` makes a new group) share the list's threads; groups on DIFFERENT lists share
// nothing, so two models on disjoint device lists run their calls — collectives included — side by side.  A worker is parked
// on its condition variable when idle.  At process exit (abo::at_exit, before the HIP runtime tears down) idle workers are
// told to stop and joined; one that is still inside a job — the host is exiting in the middle of a call — is left alone.
struct Worker {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool stop = false, busy = false;
    std::thread th;
    Worker() {
        th = std::thread([this] {
            for (;;) {
                std::function<void()> job;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [this] { return stop || !q.empty(); });
                    if (q.empty()) return;                 // stop requested and nothing left to run
                    job = std::move(q.front());
                    q.pop_front();
                    busy = true;
                }
                job();
                { std::lock_guard<std::mutex> lk(mu); busy = false; }
            }
        });
    }
    void post(std::function<void()> f) {
        { std::lock_guard<std::mutex> lk(mu); q.push_back(std::move(f)); }
        cv.notify_one();
    }
    // exit path: returns true when the thread has been joined
    bool stop_if_idle() {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (busy || !q.empty()) return false;
            stop = true;
        }
        cv.notify_one();
        if (th.joinable()) th.join();
        return true;
    }
};

std::mutex g_workers_mu;
std::vector<Worker*> g_workers;          // every worker of the process (never freed: a busy one may outlive the exit hook)

void stop_workers_at_exit() {
    std::vector<Worker*> all;
    { std::lock_guard<std::mutex> lk(g_workers_mu); all = g_workers; }
    for (Worker* w : all)
        if (!w->stop_if_idle()) w->th.detach();    // mid-call at exit: never join a thread that may be blocked in the runtime
}

Worker* new_worker() {
    Worker* w = new Worker();
    std::lock_guard<std::mutex> lk(g_workers_mu);
    if (g_workers.empty()) abo::at_exit(stop_workers_at_exit);
    g_workers.push_back(w);
    return w;
}

// f(i) for every shard i concurrently on the given workers; the first failure (lowest shard) becomes the caller's status
// and error text
int32_t run_all(Worker* const* wk, int n, const std::function<int32_t(int)>& f) {
    if (n == 1) return f(0);
    std::mutex mu;
    std::condition_variable cv;
    int left = n;
    std::vector<int32_t> rc(n, 0);
    std::vector<std::string> err(n);
    for (int i = 0; i < n; ++i) {
        wk[i]->post([&, i] {
            rc[i] = f(i);
            if (rc[i]) err[i] = abo::last_error_text();
            std::lock_guard<std::mutex> lk(mu);
            if (--left == 0) cv.notify_one();
        });
    }
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return left == 0; });
    }
    for (int i = 0; i < n; ++i)
        if (rc[i]) return abo::set_error(rc[i], err[i].c_str());
    return ABO_OK;
}

// ---- per-device-list exchange state: communicators and device buffers, created once per process -----------------
struct CommSet {
    std::vector<int> dev;
    bool rccl_ok = false;
    std::string why;
    ncclComm_t comm[MAXDEV] = {nullptr};
    void* pack[MAXDEV] = {nullptr};     // this shard's contribution
    void* gath[MAXDEV] = {nullptr};     // the gathered contributions of all shards
    size_t pack_cap[MAXDEV] = {0}, gath_cap[MAXDEV] = {0};
    std::mutex mu;                      // one exchange at a time per device list
    Worker* wk[MAXDEV] = {nullptr};     // the list's worker threads, one per shard (more than one shard only)
    std::vector<int> lock_order;        // the list's distinct devices, ascending: the order their collective locks are taken in
    bool rccl_ever = false;             // a communicator set existed (abo_mgpu_info's text after a fall-back)
};

// One collective lock per DEVICE: an exchange holds the locks of every device of its list (taken in ascending device order, so
// two lists can never wait for each other).  Two communicators enqueueing on one device from different threads can wait for
// each other forever; lists that share no device share no lock and run their collectives side by side.  (Round 3 kept a
// per-list `overlaps` flag sampled at entry: an exchange already running on list A without the lock could interleave with the
// first collective of a newly created overlapping list B.)
constexpr int MAXDEVID = 64;
std::mutex g_dev_mu[MAXDEVID];

struct DeviceLocks {
    const std::vector<int>& order;
    explicit DeviceLocks(const std::vector<int>& o) : order(o) { for (int dv : order) g_dev_mu[dv & (MAXDEVID - 1)].lock(); }
    ~DeviceLocks() { for (auto it = order.rbegin(); it != order.rend(); ++it) g_dev_mu[*it & (MAXDEVID - 1)].unlock(); }
    DeviceLocks(const DeviceLocks&) = delete;
    DeviceLocks& operator=(const DeviceLocks&) = delete;
};

hipError_t ensure_dev(void** p, size_t* cap, size_t bytes) {
    if (bytes <= *cap) return hipSuccess;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    size_t want = 4096;
    while (want < bytes) want <<= 1;
    hipError_t e = hipMalloc(p, want);
    if (e == hipSuccess) *cap = want;
    return e;
}

long exchange_timeout_ms();
int wait_stream_bounded(hipStream_t s, std::atomic<int>& abort, std::chrono::steady_clock::time_point deadline);

CommSet* comm_set(const int* dev, int ndev) {
    static std::mutex mu;
    static std::map<std::vector<int>, CommSet*> sets;
    std::vector<int> key(dev, dev + ndev);
    const char* ex = getenv("ABO_MGPU_EXCHANGE");
    const bool force_host = ex && !strcmp(ex, "host");
    const bool force_rccl = ex && !strcmp(ex, "rccl");
    key.push_back(force_host ? -1 : (force_rccl ? -2 : 0));      // the override is part of the identity (tests flip it)
    std::lock_guard<std::mutex> lk(mu);
    auto it = sets.find(key);
    if (it != sets.end()) return it->second;
    CommSet* cs = new CommSet();
    cs->dev.assign(dev, dev + ndev);
    bool distinct = true;
    for (int i = 0; i < ndev; ++i)
        for (int j = 0; j < i; ++j) distinct = distinct && dev[i] != dev[j];
    // the list's distinct devices, ascending: the order their collective locks are taken in — needed NOW: communicator creation and the
    // warm-up below enqueue on these devices, and a collective of another list that shares one of them must not interleave
    cs->lock_order.assign(dev, dev + ndev);
    std::sort(cs->lock_order.begin(), cs->lock_order.end());
    cs->lock_order.erase(std::unique(cs->lock_order.begin(), cs->lock_order.end()), cs->lock_order.end());
    if (force_host) cs->why = "ABO_MGPU_EXCHANGE=host";
    else if (!distinct) cs->why = "a device is listed twice (RCCL needs distinct devices)";
    else if (ndev == 1 && !force_rccl) cs->why = "one shard: nothing to exchange";
    else if (!rccl().handle) cs->why = rccl().why;
    else {
        DeviceLocks dl(cs->lock_order);
        const ncclResult_t r = rccl().CommInitAll(cs->comm, ndev, dev);
        if (r == ncclSuccess) cs->rccl_ok = cs->rccl_ever = true;
        else cs->why = std::string("ncclCommInitAll: ") + rccl().GetErrorString(r);
        (void)hipGetLastError();
        // One tiny all-gather over every rank NOW, issued as one group from this thread: recent NCCL / RCCL set their transport
        // connections up lazily inside the first collective — a host-side exchange between the ranks — and the bounded wait of
        // exchange() only bounds what has been ENQUEUED.  After this warm-up a collective call is an asynchronous kernel launch, so a
        // rank that never arrives later costs the others a time-out, not a host call that never returns.
        // The warm-up itself runs on a non-blocking stream of its own per device (the NULL stream would order it against every
        // other stream of the process) and is waited for WITH the exchange's bound: on expiry the communicators are aborted and
        // the list falls back to the host exchange, like any later collective that does not complete.
        if (cs->rccl_ok) {
            bool ok = true;
            hipStream_t ws[MAXDEV] = {nullptr};
            for (int i = 0; ok && i < ndev; ++i)
                ok = hipSetDevice(dev[i]) == hipSuccess && ensure_dev(&cs->pack[i], &cs->pack_cap[i], 8) == hipSuccess &&
                     ensure_dev(&cs->gath[i], &cs->gath_cap[i], 8 * (size_t)ndev) == hipSuccess &&
                     hipStreamCreateWithFlags(&ws[i], hipStreamNonBlocking) == hipSuccess;
            ncclResult_t w = ok ? rccl().GroupStart() : ncclSystemError;
            for (int i = 0; w == ncclSuccess && i < ndev; ++i) {
                (void)hipSetDevice(dev[i]);
                w = rccl().AllGather(cs->pack[i], cs->gath[i], 1, ncclUint64, cs->comm[i], ws[i]);
            }
            if (ok) { const ncclResult_t e = rccl().GroupEnd(); if (w == ncclSuccess) w = e; }
            bool timed_out = false;
            if (w == ncclSuccess) {
                std::atomic<int> abort{0};
                const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(exchange_timeout_ms());
                for (int i = 0; w == ncclSuccess && i < ndev; ++i) {
                    const int q = hipSetDevice(dev[i]) == hipSuccess ? wait_stream_bounded(ws[i], abort, deadline) : -1;
                    if (q != 0) { w = ncclUnhandledCudaError; timed_out = q == 1; }
                }
            }
            if (w != ncclSuccess) {
                for (int i = 0; i < ndev; ++i)
                    if (cs->comm[i]) { (void)hipSetDevice(dev[i]); (void)rccl().CommAbort(cs->comm[i]); cs->comm[i] = nullptr; }
                cs->rccl_ok = false;
                cs->why = timed_out ? std::string("RCCL warm-up all-gather timed out: communicators released, host exchange")
                                    : std::string("RCCL warm-up all-gather failed: ") + (ok ? rccl().GetErrorString(w) : "device buffers");
            }
            for (int i = 0; i < ndev; ++i)
                if (ws[i]) { (void)hipSetDevice(dev[i]); (void)hipStreamSynchronize(ws[i]); (void)hipStreamDestroy(ws[i]); }   // (after an abort the kernel has exited)
            (void)hipGetLastError();
        }
    }
    if (ndev > 1) for (int i = 0; i < ndev; ++i) cs->wk[i] = new_worker();
    sets[key] = cs;
    return cs;
}

void shard_range(int64_t M, int i, int n, int64_t* lo, int64_t* hi) {
    const int64_t base = M / n, extra = M % n;
    *lo = i * base + (i < extra ? i : extra);
    *hi = *lo + base + (i < extra ? 1 : 0);
}

// Julia's isless-descending order on scores: NaN first, then +Inf … −Inf with 0.0 before −0.0 (misc.hip: score_key)
uint64_t score_key(double s) {
    if (s != s) return ~0ull;
    uint64_t b;
    memcpy(&b, &s, 8);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

struct Pair { uint64_t key; int64_t idx; double val; };

// merge ndev × k gathered pairs (layout per shard: k values then k indices) into the global top-k
void merge_pairs(const double* vals, const int64_t* idx, int n, int k, double* top_val, int64_t* top_idx) {
    std::vector<Pair> v;
    v.reserve(n);
    for (int e = 0; e < n; ++e)
        if (idx[e] >= 0) v.push_back({score_key(vals[e]), idx[e], vals[e]});
    std::sort(v.begin(), v.end(), [](const Pair& a, const Pair& b) { return a.key > b.key || (a.key == b.key && a.idx < b.idx); });
    const double nan = std::numeric_limits<double>::quiet_NaN();
    for (int e = 0; e < k; ++e) {
        if (e < (int)v.size()) { top_val[e] = v[e].val; top_idx[e] = v[e].idx; }
        else { top_val[e] = nan; top_idx[e] = -1; }
    }
}

}  // namespace

struct abo_mgpu {
    int ndev = 0;
    int dev[MAXDEV] = {0};
    abo_gp* gp[MAXDEV] = {nullptr};
    abo_params prm{};
    CommSet* cs = nullptr;
    // gradient-enhanced group (abo_mgpu_create_grad): p = d + 1 outputs per point and their prior means; p_out = 1: StandardGP
    int p_out = 1;
    double mean_vec[abo::MAX_P] = {0};
};

struct abo_mcand {
    int ndev = 0, d = 0;
    int64_t M = 0;
    int64_t lo[MAXDEV + 1] = {0};
    abo_cand* c[MAXDEV] = {nullptr};
    abo_qei_stats qei_last{};           // abo_mgpu_cand_qei_stats
};

namespace {

// a fresh un-conditioned handle of the group's kind on device dv
int32_t new_shard_handle(const abo_mgpu* mg, int dv, abo_gp** out) {
    abo_params p = mg->prm;
    p.device = dv;
    return mg->p_out > 1 ? abo_create_grad(&p, mg->p_out, mg->mean_vec, out) : abo_create(&p, out);
}

int32_t check_group(abo_mgpu* mg, const char* fn) {
    if (!mg) return failf(ABO_EINVAL, "%s: null group", fn);
    return ABO_OK;
}

// ---- fault injection (test hook, tests/test_gpu_multigpu.py): ABO_MGPU_FAULT = "<where>:<shard>" -------------------------------
//   ready:<i>       shard i reports "not ready" in the vote that precedes the collective
//   collective:<i>  shard i passes the vote and then fails BEFORE enqueueing its all-gather (the other ranks' all-gathers would wait
//                   for it forever: the case the abort path exists for)
//   stall:<i>       shard i passes the vote and enqueues, in place of its all-gather, a kernel that waits for a peer that never
//                   arrives (bounded: it gives up by itself after 20 s) — a collective that does not complete
// read per exchange, so a test can set and clear it around one call
struct Fault { int where = 0, shard = -1; };      // where: 1 ready, 2 collective, 3 stall
Fault fault_from_env() {
    Fault f;
    const char* e = getenv("ABO_MGPU_FAULT");
    if (!e || !*e) return f;
    const char* c = strchr(e, ':');
    if (!c) return f;
    const size_t n = (size_t)(c - e);
    if (n == 5 && !strncmp(e, "ready", 5)) f.where = 1;
    else if (n == 10 && !strncmp(e, "collective", 10)) f.where = 2;
    else if (n == 5 && !strncmp(e, "stall", 5)) f.where = 3;
    f.shard = atoi(c + 1);
    return f;
}

// the stand-in for an all-gather whose peer never enqueues: spins on a device word until it is set, or for at most `limit`
// ticks of the 100 MHz wall clock — every wave reaches the exit either way
__global__ void stall_kernel(const int* release, long long limit) {
    const long long t0 = wall_clock64();
    while (__atomic_load_n(release, __ATOMIC_RELAXED) == 0 && wall_clock64() - t0 < limit) __builtin_amdgcn_s_sleep(64);
}

long exchange_timeout_ms() {
    const char* e = getenv("ABO_MGPU_TIMEOUT_MS");
    const long v = e ? atol(e) : 0;
    return v > 0 ? v : 60000;      // the collective moves 16·k bytes per device behind work that has already been waited for
}

// wait for everything queued on s, giving up when `abort` is raised by another shard or the deadline passes (then raises it).
// 0 = completed, 1 = aborted / timed out, −1 = the stream reports an error
int wait_stream_bounded(hipStream_t s, std::atomic<int>& abort, std::chrono::steady_clock::time_point deadline) {
    for (unsigned spin = 0;; ++spin) {
        const hipError_t q = hipStreamQuery(s);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) { (void)hipGetLastError(); return -1; }
        if (abort.load(std::memory_order_acquire)) return 1;
        if (std::chrono::steady_clock::now() > deadline) { abort.store(2, std::memory_order_release); return 1; }
        if (spin < 4000) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

// host exchange: every shard copies its own block
int32_t exchange_host(abo_mgpu* mg, size_t words, uint64_t* out) {
    CommSet* cs = mg->cs;
    return run_all(cs->wk, mg->ndev, [&](int i) -> int32_t {
        if (hipSetDevice(mg->dev[i]) != hipSuccess) return failf(ABO_EHIP, "hipSetDevice(%d) failed", mg->dev[i]);
        hipStream_t s = abo::gp_stream(mg->gp[i]);
        if (hipMemcpyAsync(out + (size_t)i * words, cs->pack[i], words * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)
            return failf(ABO_EHIP, "exchange: device-to-host copy failed");
        return ABO_OK;
    });
}

// All shards hold k pairs (k values + k indices) in cs->pack[i] on their device.  out: ndev blocks of `words` 8-byte words.
// RCCL, in three steps none of which can leave a thread blocked in a collective its peers never entered:
//   1. vote      every shard answers "ready?" through run_all (device selectable, communicator and buffers present, its stream
//                healthy).  One "no" and nobody enqueues anything: the status goes back to the caller.
//   2. gather    every shard enqueues its ncclAllGather (shard 0 also the copy of the gathered block to the host) and then WAITS
//                WITH A BOUND: it polls its stream, a shared abort word and a deadline (ABO_MGPU_TIMEOUT_MS, default 60 s).  A
//                shard that fails after the vote raises the abort word; every shard that sees it (or the deadline) calls
//                ncclCommAbort on its own communicator — which makes a collective kernel waiting for a missing peer exit — and
//                returns.
//   3. fall-back after an abort the device list's communicators are gone for the rest of the process (abo_mgpu_info says so and
//                why) and THIS exchange, like all later ones, is done through the host: the blocks in cs->pack are untouched by
//                a failed all-gather, so the call still returns the merged selection if the shards themselves are healthy.
// Host exchange (no RCCL, a device listed twice, ABO_MGPU_EXCHANGE=host, or after a fall-back): every shard copies its own block.
int32_t exchange(abo_mgpu* mg, size_t words, uint64_t* out) {
    CommSet* cs = mg->cs;
    const int n = mg->ndev;
    // the caller holds cs->mu (one exchange at a time per device list); the per-device collective locks order this list's
    // collectives against those of every other list that names one of its devices
    DeviceLocks dl(cs->lock_order);
    if (!cs->rccl_ok) return exchange_host(mg, words, out);
    const Fault fault = fault_from_env();
    // 1. vote
    int32_t rc = run_all(cs->wk, n, [&](int i) -> int32_t {
        if (fault.where == 1 && fault.shard == i) return failf(ABO_EHIP, "exchange: shard %d is not ready (injected: ABO_MGPU_FAULT)", i);
        if (hipSetDevice(mg->dev[i]) != hipSuccess) return failf(ABO_EHIP, "exchange: shard %d is not ready: hipSetDevice(%d) failed", i, mg->dev[i]);
        if (!cs->comm[i] || !cs->pack[i] || !cs->gath[i] || cs->pack_cap[i] < words * 8 || cs->gath_cap[i] < (size_t)n * words * 8)
            return failf(ABO_EHIP, "exchange: shard %d is not ready: communicator or exchange buffers missing", i);
        const hipError_t q = hipStreamQuery(abo::gp_stream(mg->gp[i]));
        if (q != hipSuccess && q != hipErrorNotReady) {
            (void)hipGetLastError();
            return failf(ABO_EHIP, "exchange: shard %d is not ready: its stream reports %s", i, hipGetErrorString(q));
        }
        return ABO_OK;
    });
    if (rc) return rc;                       // nothing was enqueued anywhere
    // 2. gather, bounded: the protocol of abo_exchange.h (driven on a CPU with a stubbed transport by tests/exchange_stub.cpp) over
    // RCCL and the shards' streams
    struct RcclTransport {
        abo_mgpu* mg; CommSet* cs; size_t words; uint64_t* out; Fault fault; int* stall_word = nullptr;
        bool enqueue(int i, std::string* err) {
            hipStream_t s = abo::gp_stream(mg->gp[i]);
            if (hipSetDevice(mg->dev[i]) != hipSuccess) { *err = "hipSetDevice failed after the vote"; return false; }
            if (fault.where == 2 && fault.shard == i) { *err = "failed before its all-gather (injected: ABO_MGPU_FAULT)"; return false; }
            if (fault.where == 3 && fault.shard == i) {
                if (hipMalloc(reinterpret_cast<void**>(&stall_word), sizeof(int)) == hipSuccess && hipMemsetAsync(stall_word, 0, sizeof(int), s) == hipSuccess) {
                    hipLaunchKernelGGL(stall_kernel, dim3(1), dim3(64), 0, s, stall_word, 20ll * 100000000ll);
                    return true;
                }
                *err = "stall injection could not allocate";
                return false;
            }
            const ncclResult_t r = rccl().AllGather(cs->pack[i], cs->gath[i], words, ncclUint64, cs->comm[i], s);
            if (r != ncclSuccess) { *err = rccl().GetErrorString(r); return false; }
            if (i == 0 && hipMemcpyAsync(out, cs->gath[0], mg->ndev * words * 8, hipMemcpyDeviceToHost, s) != hipSuccess) {
                *err = "device-to-host copy of the gathered block failed";
                return false;
            }
            return true;
        }
        int poll(int i) {
            const hipError_t q = hipStreamQuery(abo::gp_stream(mg->gp[i]));
            if (q == hipSuccess) return 0;
            if (q != hipErrorNotReady) { (void)hipGetLastError(); return -1; }
            return 1;
        }
        void abort(int i) {
            if (cs->comm[i]) { (void)rccl().CommAbort(cs->comm[i]); cs->comm[i] = nullptr; }
            if (stall_word && fault.shard == i) {        // release the stand-in kernel (on a stream of its own: the shard's is busy with it)
                const int one = 1;
                hipStream_t t = nullptr;
                if (hipStreamCreateWithFlags(&t, hipStreamNonBlocking) == hipSuccess) {
                    (void)hipMemcpyAsync(stall_word, &one, sizeof one, hipMemcpyHostToDevice, t);
                    (void)hipStreamSynchronize(t);
                    (void)hipStreamDestroy(t);
                }
            }
        }
        void drain(int i) {
            (void)hipStreamSynchronize(abo::gp_stream(mg->gp[i]));   // returns: the collective kernel exits on the abort, the stand-in on its word or its clock
            (void)hipGetLastError();
        }
    } tr{mg, cs, words, out, fault};
    const abo::GatherVerdict gv = abo::bounded_gather(tr, n, exchange_timeout_ms(), [&](const std::function<void(int)>& f) {
        (void)run_all(cs->wk, n, [&](int i) -> int32_t { f(i); return ABO_OK; });
    });
    if (tr.stall_word) { (void)hipFree(tr.stall_word); tr.stall_word = nullptr; }
    if (!gv.aborted) return ABO_OK;
    const std::string first_fault = gv.first_fault;
    // 3. fall-back: no thread is inside RCCL any more; the set's remaining communicators go too
    for (int i = 0; i < n; ++i)
        if (cs->comm[i]) { (void)hipSetDevice(mg->dev[i]); (void)rccl().CommAbort(cs->comm[i]); cs->comm[i] = nullptr; }
    (void)hipGetLastError();
    cs->rccl_ok = false;
    cs->why = std::string(gv.aborted == 2 ? "RCCL exchange timed out" : "RCCL exchange aborted") +
              (first_fault.empty() ? "" : " (" + first_fault + ")") + ": communicators released, host exchange from now on";
    return exchange_host(mg, words, out);
}

int32_t ensure_exchange_buffers(abo_mgpu* mg, int i, size_t words) {
    CommSet* cs = mg->cs;
    if (hipSetDevice(mg->dev[i]) != hipSuccess) return failf(ABO_EHIP, "hipSetDevice(%d) failed", mg->dev[i]);
    if (ensure_dev(&cs->pack[i], &cs->pack_cap[i], words * 8) != hipSuccess ||
        (cs->rccl_ok && ensure_dev(&cs->gath[i], &cs->gath_cap[i], (size_t)mg->ndev * words * 8) != hipSuccess))
        return failf(ABO_ENOMEM, "exchange buffers: device allocation failed");
    return ABO_OK;
}

// gathered blocks (k values, k indices per shard) → merged global top-k
void merge_blocks(const uint64_t* blocks, int ndev, int k, double* top_val, int64_t* top_idx) {
    std::vector<double> vals((size_t)ndev * k);
    std::vector<int64_t> idx((size_t)ndev * k);
    for (int i = 0; i < ndev; ++i) {
        memcpy(&vals[(size_t)i * k], blocks + (size_t)i * 2 * k, sizeof(double) * k);
        memcpy(&idx[(size_t)i * k], blocks + (size_t)i * 2 * k + k, sizeof(int64_t) * k);
    }
    merge_pairs(vals.data(), idx.data(), ndev * k, k, top_val, top_idx);
}

}  // namespace

extern "C" {

static int32_t mgpu_create_impl(const abo_params* params, int32_t p_out, const double* mean_c, int32_t ndev, const int32_t* dev,
                                abo_mgpu** out, const char* fn) {
    if (!params || !dev || !out) return failf(ABO_EINVAL, "%s: null argument", fn);
    if (ndev < 1 || ndev > MAXDEV) return failf(ABO_EINVAL, "%s: ndev = %d outside 1..%d", fn, ndev, MAXDEV);
    if (p_out > abo::MAX_P) return failf(ABO_EINVAL, "%s: p = %d outputs outside 2..%d", fn, p_out, abo::MAX_P);
    abo_mgpu* mg = new (std::nothrow) abo_mgpu();
    if (!mg) return failf(ABO_ENOMEM, "%s: host allocation failed", fn);
    mg->ndev = ndev;
    mg->prm = *params;
    mg->p_out = p_out;
    for (int q = 0; q < p_out && p_out > 1; ++q) mg->mean_vec[q] = mean_c ? mean_c[q] : 0.0;
    for (int i = 0; i < ndev; ++i) {
        mg->dev[i] = dev[i];
        const int32_t rc = new_shard_handle(mg, dev[i], &mg->gp[i]);      // (validates the parameters once per device)
        if (rc) { abo_mgpu_destroy(mg); return rc; }
    }
    mg->cs = comm_set(mg->dev, ndev);
    *out = mg;
    return ABO_OK;
}

int32_t abo_mgpu_create(const abo_params* params, int32_t ndev, const int32_t* dev, abo_mgpu** out) {
    return mgpu_create_impl(params, 1, nullptr, ndev, dev, out, "abo_mgpu_create");
}

int32_t abo_mgpu_create_grad(const abo_params* params, int32_t p, const double* mean_c, int32_t ndev, const int32_t* dev, abo_mgpu** out) {
    if (p < 2) return failf(ABO_EINVAL, "abo_mgpu_create_grad: p = %d outputs (a gradient-enhanced model has p = d + 1 >= 2)", p);
    return mgpu_create_impl(params, p, mean_c, ndev, dev, out, "abo_mgpu_create_grad");
}

int32_t abo_mgpu_clone(abo_mgpu* mg, abo_mgpu** out) {
    if (!mg || !out) return failf(ABO_EINVAL, "abo_mgpu_clone: null argument");
    abo_mgpu* n = new (std::nothrow) abo_mgpu(*mg);
    if (!n) return failf(ABO_ENOMEM, "abo_mgpu_clone: host allocation failed");
    for (int i = 0; i < n->ndev; ++i) abo_retain(n->gp[i]);
    *out = n;
    return ABO_OK;
}

int32_t abo_mgpu_destroy(abo_mgpu* mg) {
    if (!mg || abo::exiting()) return ABO_OK;          // late finaliser: the device state goes with the process (abo_internal.h)
    for (int i = 0; i < mg->ndev; ++i) abo_destroy(mg->gp[i]);
    delete mg;
    return ABO_OK;
}

int32_t abo_mgpu_info(abo_mgpu* mg, int32_t* ndev, int32_t* dev, int32_t* exch) {
    int32_t rc = check_group(mg, "abo_mgpu_info");
    if (rc) return rc;
    if (ndev) *ndev = mg->ndev;
    if (dev) for (int i = 0; i < MAXDEV; ++i) dev[i] = i < mg->ndev ? mg->dev[i] : -1;
    if (exch) *exch = mg->cs->rccl_ok ? ABO_XCHG_RCCL : ABO_XCHG_HOST;
    if (!mg->cs->rccl_ok) abo::set_error(ABO_OK, mg->cs->why.c_str());     // why not RCCL: readable through abo_last_error
    return ABO_OK;
}

int32_t abo_mgpu_get(abo_mgpu* mg, int32_t i, abo_gp** out) {
    int32_t rc = check_group(mg, "abo_mgpu_get");
    if (rc) return rc;
    if (!out || i < 0 || i >= mg->ndev) return failf(ABO_EINVAL, "abo_mgpu_get: shard %d outside 0..%d", i, mg->ndev - 1);
    *out = mg->gp[i];
    return ABO_OK;
}

int32_t abo_mgpu_fit(abo_mgpu* mg, const double* X, int64_t N, int32_t d, const double* y, int64_t* info) {
    if (info) *info = 0;
    int32_t rc = check_group(mg, "abo_mgpu_fit");
    if (rc) return rc;
    // update() returns a NEW model (StandardGP.jl:82) and a clone shares its handles with the group it came from: fit
    // fresh handles and swap them in on success; on failure (PosDefException …) the group keeps its previous state,
    // which is what the driver's rollback restores anyway (src/bayesian_opt.jl:126-141)
    int64_t inf[MAXDEV] = {0};
    abo_gp* nw[MAXDEV] = {nullptr};
    rc = run_all(mg->cs->wk, mg->ndev, [&](int i) -> int32_t {
        const int32_t r = new_shard_handle(mg, mg->dev[i], &nw[i]);
        return r ? r : abo_fit(nw[i], X, N, d, y, ABO_HOST, &inf[i]);      // gradient-enhanced: y holds p·N values, by outputs
    });
    if (info) for (int i = 0; i < mg->ndev; ++i) if (inf[i]) { *info = inf[i]; break; }
    if (rc) {
        const std::string keep = abo::last_error_text();
        for (int i = 0; i < mg->ndev; ++i) if (nw[i]) abo_destroy(nw[i]);
        return abo::set_error(rc, keep.c_str());
    }
    for (int i = 0; i < mg->ndev; ++i) { abo_destroy(mg->gp[i]); mg->gp[i] = nw[i]; }
    return ABO_OK;
}

int32_t abo_mgpu_predict(abo_mgpu* mg, const double* Z, int64_t M, int32_t d, double* mu, double* var) {
    int32_t rc = check_group(mg, "abo_mgpu_predict");
    if (rc) return rc;
    if (M < 0 || (M > 0 && !Z)) return failf(ABO_EINVAL, "abo_mgpu_predict: bad candidate buffer");
    return run_all(mg->cs->wk, mg->ndev, [&](int i) -> int32_t {
        int64_t lo, hi;
        shard_range(M, i, mg->ndev, &lo, &hi);
        // an empty shard still checks the dimension (same status on every device)
        return abo_predict(mg->gp[i], Z + lo * d, hi - lo, d, ABO_HOST, mu ? mu + lo : nullptr, var ? var + lo : nullptr, ABO_HOST);
    });
}

int32_t abo_mgpu_acq(abo_mgpu* mg, const double* Z, int64_t M, int32_t d, int32_t kind, double p0, double best_y,
                     double* scores, int32_t k, double* top_val, int64_t* top_idx) {
    int32_t rc = check_group(mg, "abo_mgpu_acq");
    if (rc) return rc;
    if (M < 0 || (M > 0 && !Z)) return failf(ABO_EINVAL, "abo_mgpu_acq: bad candidate buffer");
    if (k < 0) return failf(ABO_EINVAL, "abo_mgpu_acq: k = %d is negative", k);
    if (k > 0 && (!top_val || !top_idx)) return failf(ABO_EINVAL, "abo_mgpu_acq: k > 0 needs top_val and top_idx");
    CommSet* cs = mg->cs;
    std::lock_guard<std::mutex> lk(cs->mu);
    const size_t words = 2 * (size_t)k;
    rc = run_all(mg->cs->wk, mg->ndev, [&](int i) -> int32_t {
        int64_t lo, hi;
        shard_range(M, i, mg->ndev, &lo, &hi);
        double* tv = nullptr;
        int64_t* ti = nullptr;
        if (k > 0) {
            const int32_t r = ensure_exchange_buffers(mg, i, words);
            if (r) return r;
            tv = static_cast<double*>(cs->pack[i]);
            ti = static_cast<int64_t*>(cs->pack[i]) + k;
        }
        return abo::acq_ex(mg->gp[i], Z + lo * d, hi - lo, d, ABO_HOST, kind, p0, best_y, lo, scores ? scores + lo : nullptr,
                           ABO_HOST, k, tv, ti, ABO_DEVICE);
    });
    if (rc || k == 0) return rc;
    std::vector<uint64_t> blocks((size_t)mg->ndev * words);
    rc = exchange(mg, words, blocks.data());
    if (rc) return rc;
    merge_blocks(blocks.data(), mg->ndev, k, top_val, top_idx);
    return ABO_OK;
}

static int32_t mgpu_acq_lhs_terms(abo_mgpu* mg, const abo_acq_term* terms, int32_t nterms, int64_t n, int32_t d, const double* lower,
                                  const double* upper, uint64_t seed, int32_t k, double* top_val, int64_t* top_idx, double* top_x) {
    int32_t rc = check_group(mg, "abo_mgpu_acq_lhs");
    if (rc) return rc;
    if (n < 1 || !lower || !upper) return failf(ABO_EINVAL, "abo_mgpu_acq_lhs: bad grid");
    if (k < 1 || !top_val || !top_idx) return failf(ABO_EINVAL, "abo_mgpu_acq_lhs: needs k >= 1, top_val and top_idx");
    CommSet* cs = mg->cs;
    std::lock_guard<std::mutex> lk(cs->mu);
    const size_t words = 2 * (size_t)k;
    double* zdev[MAXDEV] = {nullptr};
    auto free_grids = [&] {
        for (int i = 0; i < mg->ndev; ++i)
            if (zdev[i]) { (void)hipSetDevice(mg->dev[i]); (void)hipFree(zdev[i]); zdev[i] = nullptr; }
    };
    rc = run_all(mg->cs->wk, mg->ndev, [&](int i) -> int32_t {
        int64_t lo, hi;
        shard_range(n, i, mg->ndev, &lo, &hi);
        int32_t r = ensure_exchange_buffers(mg, i, words);
        if (r) return r;
        if (hi > lo && hipMalloc(reinterpret_cast<void**>(&zdev[i]), sizeof(double) * (hi - lo) * d) != hipSuccess)
            return failf(ABO_ENOMEM, "abo_mgpu_acq_lhs: device allocation of the grid shard failed");
        return abo::acq_lhs_shard(mg->gp[i], terms, nterms, n, d, lower, upper, seed, lo, hi - lo, k, zdev[i],
                                  static_cast<double*>(cs->pack[i]), static_cast<int64_t*>(cs->pack[i]) + k);
    });
    std::vector<uint64_t> blocks((size_t)mg->ndev * words);
    if (!rc) rc = exchange(mg, words, blocks.data());
    if (rc) { const std::string keep = abo::last_error_text(); free_grids(); return abo::set_error(rc, keep.c_str()); }
    merge_blocks(blocks.data(), mg->ndev, k, top_val, top_idx);
    if (top_x) {
        // coordinates of the winners, fetched from the shard that generated them
        rc = run_all(mg->cs->wk, mg->ndev, [&](int i) -> int32_t {
            int64_t lo, hi;
            shard_range(n, i, mg->ndev, &lo, &hi);
            if (hipSetDevice(mg->dev[i]) != hipSuccess) return failf(ABO_EHIP, "hipSetDevice(%d) failed", mg->dev[i]);
            hipStream_t s = abo::gp_stream(mg->gp[i]);
            for (int e = 0; e < k; ++e) {
                const int64_t g = top_idx[e];
                if (g < lo || g >= hi) continue;
                if (hipMemcpyAsync(top_x + (size_t)e * d, zdev[i] + (g - lo) * d, sizeof(double) * d, hipMemcpyDeviceToHost, s) != hipSuccess)
                    return failf(ABO_EHIP, "abo_mgpu_acq_lhs: copy of a selected point failed");
            }
            if (hipStreamSynchronize(s) != hipSuccess) return failf(ABO_EHIP, "abo_mgpu_acq_lhs: stream synchronisation failed");
            return ABO_OK;
        });
        const double nan = std::numeric_limits<double>::quiet_NaN();
        for (int e = 0; e < k; ++e)
            if (top_idx[e] < 0) for (int c = 0; c < d; ++c) top_x[(size_t)e * d + c] = nan;
    }
    free_grids();
    return rc;
}

int32_t abo_mgpu_acq_lhs(abo_mgpu* mg, int64_t n, int32_t d, const double* lower, const double* upper, uint64_t seed,
                         int32_t kind, double p0, double best_y, int32_t k, double* top_val, int64_t* top_idx, double* top_x) {
    const abo_acq_term one{kind, 0, p0, best_y, 1.0};
    return mgpu_acq_lhs_terms(mg, &one, 1, n, d, lower, upper, seed, k, top_val, top_idx, top_x);
}

// optimize_acquisition (acq_utils.jl:33-73) across the group: grid stage sharded (abo_mgpu_acq_lhs), the selected starts dealt
// out contiguously, every device refines its share (abo_refine_terms), the host keeps the best
int32_t abo_mgpu_optimize_acquisition_terms(abo_mgpu* mg, const abo_acq_term* terms, int32_t nterms, const double* lower,
                                            const double* upper, int32_t d, int64_t n_grid, int32_t n_local, uint64_t seed,
                                            const abo_refine_opts* opts, double* best_x, double* best_val, double* starts_x,
                                            double* starts_val, double* refined_x, double* refined_val) {
    int32_t rc = check_group(mg, "abo_mgpu_optimize_acquisition");
    if (rc) return rc;
    if (!lower || !upper || !best_x || !terms) return failf(ABO_EINVAL, "abo_mgpu_optimize_acquisition: null argument");
    if (n_grid < 1 || n_local < 1) return failf(ABO_EINVAL, "abo_mgpu_optimize_acquisition: n_grid and n_local must be positive");
    const int k = (int)(n_local < n_grid ? n_local : n_grid);
    std::vector<double> tv(k), tx((size_t)k * d), rx((size_t)k * d), rf(k);
    std::vector<int64_t> ti(k);
    rc = mgpu_acq_lhs_terms(mg, terms, nterms, n_grid, d, lower, upper, seed, k, tv.data(), ti.data(), tx.data());
    if (rc) return rc;
    // a selection shorter than k cannot happen (k ≤ n_grid), but a NaN score can: such a start is returned unchanged by abo_refine
    rc = run_all(mg->cs->wk, mg->ndev, [&](int i) -> int32_t {
        int64_t lo, hi;
        shard_range(k, i, mg->ndev, &lo, &hi);
        if (hi == lo) return ABO_OK;
        return abo::refine_terms(mg->gp[i], terms, nterms, lower, upper, d, tx.data() + lo * d, (int32_t)(hi - lo), opts,
                                 rx.data() + lo * d, rf.data() + lo);
    });
    if (rc) return rc;
    abo::pick_best_point(tx.data(), tv.data(), rx.data(), rf.data(), k, d, best_x, best_val);
    if (starts_x) memcpy(starts_x, tx.data(), sizeof(double) * k * d);
    if (starts_val) memcpy(starts_val, tv.data(), sizeof(double) * k);
    if (refined_x) memcpy(refined_x, rx.data(), sizeof(double) * k * d);
    if (refined_val) memcpy(refined_val, rf.data(), sizeof(double) * k);
    return ABO_OK;
}

int32_t abo_mgpu_optimize_acquisition(abo_mgpu* mg, int32_t kind, double p0, double best_y, const double* lower, const double* upper,
                                      int32_t d, int64_t n_grid, int32_t n_local, uint64_t seed, const abo_refine_opts* opts,
                                      double* best_x, double* best_val, double* starts_x, double* starts_val, double* refined_x,
                                      double* refined_val) {
    const abo_acq_term one{kind, 0, p0, best_y, 1.0};
    return abo_mgpu_optimize_acquisition_terms(mg, &one, 1, lower, upper, d, n_grid, n_local, seed, opts, best_x, best_val, starts_x,
                                               starts_val, refined_x, refined_val);
}

// ---- config 5 across devices -----------------------------------------------------------------------------------
static int32_t mgpu_append_impl(abo_mgpu* mg, const double* x, int32_t d, double y, const double* yv, int64_t* info, abo_mcand* mc,
                                const char* fn) {
    if (info) *info = 0;
    int32_t rc = check_group(mg, fn);
    if (rc) return rc;
    if (mc && mc->ndev != mg->ndev) return failf(ABO_EINVAL, "%s: candidate set has %d shards, group %d", fn, mc->ndev, mg->ndev);
    abo_gp* nw[MAXDEV] = {nullptr};
    int64_t inf[MAXDEV] = {0};
    rc = run_all(mg->cs->wk, mg->ndev, [&](int i) -> int32_t {
        int32_t r = yv ? abo_append_grad(mg->gp[i], x, d, yv, &inf[i], &nw[i]) : abo_append(mg->gp[i], x, d, y, &inf[i], &nw[i]);
        if (r) return r;
        if (mc) r = abo_cand_downdate(nw[i], mc->c[i]);
        return r;
    });
    if (rc) {
        const std::string keep = abo::last_error_text();
        for (int i = 0; i < mg->ndev; ++i)
            if (inf[i] && info && !*info) *info = inf[i];
        // a shard that had already down-dated its candidates is out of step with its (unchanged) model now
        bool any = false;
        for (int i = 0; i < mg->ndev; ++i) any = any || nw[i] != nullptr;
        for (int i = 0; i < mg->ndev; ++i) if (nw[i]) { abo_destroy(nw[i]); nw[i] = nullptr; }
        if (mc && any) (void)abo_mgpu_cand_refresh(mg, mc);
        return abo::set_error(rc, keep.c_str());
    }
    for (int i = 0; i < mg->ndev; ++i) { abo_destroy(mg->gp[i]); mg->gp[i] = nw[i]; }
    return ABO_OK;
}

int32_t abo_mgpu_append(abo_mgpu* mg, const double* x, int32_t d, double y, int64_t* info, abo_mcand* mc) {
    if (mg && mg->p_out > 1) return failf(ABO_EINVAL, "abo_mgpu_append: a gradient-enhanced group takes p values per observation (abo_mgpu_append_grad)");
    return mgpu_append_impl(mg, x, d, y, nullptr, info, mc, "abo_mgpu_append");
}

int32_t abo_mgpu_append_grad(abo_mgpu* mg, const double* x, int32_t d, const double* y, int64_t* info, abo_mcand* mc) {
    if (!y) return failf(ABO_EINVAL, "abo_mgpu_append_grad: null argument");
    if (mg && mg->p_out < 2) return failf(ABO_EINVAL, "abo_mgpu_append_grad: the group's model has no gradient outputs (abo_mgpu_append)");
    return mgpu_append_impl(mg, x, d, 0.0, y, info, mc, "abo_mgpu_append_grad");
}

int32_t abo_mgpu_cand_destroy(abo_mcand* mc) {
    if (!mc || abo::exiting()) return ABO_OK;
    for (int i = 0; i < mc->ndev; ++i) abo_cand_destroy(mc->c[i]);
    delete mc;
    return ABO_OK;
}

static int32_t mcand_new(abo_mgpu* mg, int64_t M, int32_t d, abo_mcand** out) {
    abo_mcand* mc = new (std::nothrow) abo_mcand();
    if (!mc) return failf(ABO_ENOMEM, "candidate set: host allocation failed");
    mc->ndev = mg->ndev; mc->d = d; mc->M = M;
    for (int i = 0; i < mg->ndev; ++i) shard_range(M, i, mg->ndev, &mc->lo[i], &mc->lo[i + 1]);
    *out = mc;
    return ABO_OK;
}

int32_t abo_mgpu_cand_create(abo_mgpu* mg, const double* Z, int64_t M, int32_t d, abo_mcand** out) {
    int32_t rc = check_group(mg, "abo_mgpu_cand_create");
    if (rc) return rc;
    if (!out || M < 0 || (M > 0 && !Z)) return failf(ABO_EINVAL, "abo_mgpu_cand_create: bad argument");
    abo_mcand* mc = nullptr;
    rc = mcand_new(mg, M, d, &mc);
    if (rc) return rc;
    rc = run_all(mg->cs->wk, mg->ndev, [&](int i) -> int32_t {
        return abo_cand_create(mg->gp[i], Z + mc->lo[i] * d, mc->lo[i + 1] - mc->lo[i], d, ABO_HOST, &mc->c[i]);
    });
    if (rc) { const std::string keep = abo::last_error_text(); abo_mgpu_cand_destroy(mc); return abo::set_error(rc, keep.c_str()); }
    *out = mc;
    return ABO_OK;
}

int32_t abo_mgpu_cand_create_lhs(abo_mgpu* mg, int64_t n, int32_t d, const double* lower, const double* upper, uint64_t seed,
                                 abo_mcand** out) {
    int32_t rc = check_group(mg, "abo_mgpu_cand_create_lhs");
    if (rc) return rc;
    if (!out || n < 1 || !lower || !upper) return failf(ABO_EINVAL, "abo_mgpu_cand_create_lhs: bad argument");
    abo_mcand* mc = nullptr;
    rc = mcand_new(mg, n, d, &mc);
    if (rc) return rc;
    rc = run_all(mg->cs->wk, mg->ndev, [&](int i) -> int32_t {
        const int64_t m = mc->lo[i + 1] - mc->lo[i];
        double* z = nullptr;
        if (hipSetDevice(mg->dev[i]) != hipSuccess) return failf(ABO_EHIP, "hipSetDevice(%d) failed", mg->dev[i]);
        if (hipMalloc(reinterpret_cast<void**>(&z), sizeof(double) * (m > 0 ? m : 1) * d) != hipSuccess)
            return failf(ABO_ENOMEM, "abo_mgpu_cand_create_lhs: device allocation of the grid shard failed");
        int32_t r = m > 0 ? abo_lhs(mg->dev[i], n, d, lower, upper, seed, mc->lo[i], m, z) : ABO_OK;
        if (!r) r = abo_cand_create(mg->gp[i], z, m, d, ABO_DEVICE, &mc->c[i]);   // copies the points into the set
        (void)hipFree(z);
        return r;
    });
    if (rc) { const std::string keep = abo::last_error_text(); abo_mgpu_cand_destroy(mc); return abo::set_error(rc, keep.c_str()); }
    *out = mc;
    return ABO_OK;
}

int32_t abo_mgpu_cand_refresh(abo_mgpu* mg, abo_mcand* mc) {
    int32_t rc = check_group(mg, "abo_mgpu_cand_refresh");
    if (rc) return rc;
    if (!mc || mc->ndev != mg->ndev) return failf(ABO_EINVAL, "abo_mgpu_cand_refresh: candidate set does not belong to this group");
    return run_all(mg->cs->wk, mg->ndev, [&](int i) -> int32_t { return abo_cand_refresh(mg->gp[i], mc->c[i]); });
}

int32_t abo_mgpu_cand_get(abo_mgpu* mg, abo_mcand* mc, double* mu, double* var) {
    int32_t rc = check_group(mg, "abo_mgpu_cand_get");
    if (rc) return rc;
    if (!mc || mc->ndev != mg->ndev) return failf(ABO_EINVAL, "abo_mgpu_cand_get: candidate set does not belong to this group");
    return run_all(mg->cs->wk, mg->ndev, [&](int i) -> int32_t {
        if (mc->lo[i + 1] == mc->lo[i]) return ABO_OK;
        return abo_cand_get(mg->gp[i], mc->c[i], mu ? mu + mc->lo[i] : nullptr, var ? var + mc->lo[i] : nullptr, ABO_HOST);
    });
}

int32_t abo_mgpu_cand_acq(abo_mgpu* mg, abo_mcand* mc, int32_t kind, double p0, double best_y, int32_t k, double* top_val,
                          int64_t* top_idx) {
    int32_t rc = check_group(mg, "abo_mgpu_cand_acq");
    if (rc) return rc;
    if (!mc || mc->ndev != mg->ndev) return failf(ABO_EINVAL, "abo_mgpu_cand_acq: candidate set does not belong to this group");
    if (k < 1 || !top_val || !top_idx) return failf(ABO_EINVAL, "abo_mgpu_cand_acq: needs k >= 1, top_val and top_idx");
    CommSet* cs = mg->cs;
    std::lock_guard<std::mutex> lk(cs->mu);
    const size_t words = 2 * (size_t)k;
    rc = run_all(mg->cs->wk, mg->ndev, [&](int i) -> int32_t {
        const int32_t r = ensure_exchange_buffers(mg, i, words);
        if (r) return r;
        return abo::cand_acq_ex(mg->gp[i], mc->c[i], kind, p0, best_y, mc->lo[i], nullptr, ABO_DEVICE, k,
                                static_cast<double*>(cs->pack[i]), static_cast<int64_t*>(cs->pack[i]) + k, ABO_DEVICE);
    });
    if (rc) return rc;
    std::vector<uint64_t> blocks((size_t)mg->ndev * words);
    rc = exchange(mg, words, blocks.data());
    if (rc) return rc;
    merge_blocks(blocks.data(), mg->ndev, k, top_val, top_idx);
    return ABO_OK;
}

int32_t abo_mgpu_cand_qei(abo_mgpu* mg, abo_mcand* mc, int32_t q, double xi, double best_y, int32_t distinct, double* x_out,
                          int64_t* idx_out, double* ei_out) {
    int32_t rc = check_group(mg, "abo_mgpu_cand_qei");
    if (rc) return rc;
    if (!mc || mc->ndev != mg->ndev) return failf(ABO_EINVAL, "abo_mgpu_cand_qei: candidate set does not belong to this group");
    if (q < 1) return failf(ABO_EINVAL, "abo_mgpu_cand_qei: q = %d", q);
    const int n = mg->ndev, d = mc->d;
    CommSet* cs = mg->cs;
    std::lock_guard<std::mutex> lk(cs->mu);
    const auto t0 = std::chrono::steady_clock::now();
    mc->qei_last = abo_qei_stats{};
    mc->qei_last.picks = q;
    // Block form (include/abo_hip.h, api.hip: qei_drive): the covariance columns of the T best candidates from ONE pass over each
    // shard's resident K_ZX, no fantasy appends; what the shards exchange are pick records {EI, index, μ, σ², x[d], c_1(x) … c_n(x)}
    // (one per pick) and, when a block is built, T of them per shard.  Every shard applies the same exchanged numbers, so the
    // sharded batch repeats the single handle's arithmetic bit for bit.  Falls back to the plain loop below when a shard does not
    // qualify (gradient-enhanced model, K_ZX not resident, q > 64, block size 0).
    if (abo::qei_max_words(d, q, 0) > 0 && abo::qei_block_default() > 0 &&
        run_all(cs->wk, n, [&](int i) -> int32_t { return abo::qei_eligible(mg->gp[i], mc->c[i], q); }) == ABO_OK) {
        const size_t maxw = abo::qei_max_words(d, q, 0);
        rc = run_all(cs->wk, n, [&](int i) -> int32_t { return ensure_exchange_buffers(mg, i, maxw); });
        if (rc) return rc;
        abo::QeiShards S;
        S.n = n; S.gp = mg->gp; S.cd = mc->c; S.lo = mc->lo;
        S.run = [&](const std::function<int32_t(int)>& f) { return run_all(cs->wk, n, f); };
        S.rec = [&](int i) { return static_cast<double*>(cs->pack[i]); };
        S.gather = [&](size_t words, double* out) { return exchange(mg, words, reinterpret_cast<uint64_t*>(out)); };
        int64_t info = 0;
        rc = abo::qei_drive(S, q, xi, best_y, distinct, 0, x_out, idx_out, ei_out, &info);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        abo::qei_get_stats(mc->c[0], q, ms, &mc->qei_last);
        return rc;
    }
    (void)hipGetLastError();
    // exchange block per shard: the pick record {score, index, μ, x[d]}; the (score, index) pair of the device top-1 is
    // parked behind it in the same buffer
    const size_t words = 3 + (size_t)d;
    rc = run_all(mg->cs->wk, n, [&](int i) -> int32_t {
        const int32_t r = ensure_exchange_buffers(mg, i, words + 2);
        if (r) return r;
        return abo_cand_save(mg->gp[i], mc->c[i]);
    });
    if (rc) return rc;
    abo_gp* cur[MAXDEV];
    for (int i = 0; i < n; ++i) cur[i] = mg->gp[i];
    std::vector<double> recs((size_t)n * words);
    int32_t status = ABO_OK;
    std::string keep;
    for (int j = 0; j < q && !status; ++j) {
        status = run_all(mg->cs->wk, n, [&](int i) -> int32_t {
            double* rec = static_cast<double*>(cs->pack[i]);
            double* tv = rec + words;
            int64_t* ti = reinterpret_cast<int64_t*>(rec + words + 1);
            int32_t r = abo::cand_acq_ex(cur[i], mc->c[i], ABO_ACQ_EI, xi, best_y, mc->lo[i], nullptr, ABO_DEVICE, 1, tv, ti, ABO_DEVICE);
            if (r) return r;
            if (abo::launch_pick_record(tv, ti, mc->lo[i], abo::cand_points(mc->c[i]), abo::cand_mu(mc->c[i]), d, rec,
                                        abo::gp_stream(cur[i])) != hipSuccess)
                return failf(ABO_EHIP, "abo_mgpu_cand_qei: pick-record launch failed");
            // the exchange runs on the stream of the group's base handle, the record was written on the stream of `cur`
            if (hipStreamSynchronize(abo::gp_stream(cur[i])) != hipSuccess) return failf(ABO_EHIP, "stream synchronisation failed");
            return ABO_OK;
        });
        if (status) break;
        status = exchange(mg, words, reinterpret_cast<uint64_t*>(recs.data()));
        if (status) break;
        int win = -1;
        for (int i = 0; i < n; ++i) {
            const double* r = &recs[(size_t)i * words];
            if (r[1] < 0) continue;                                   // empty shard
            if (win < 0) { win = i; continue; }
            const double* w = &recs[(size_t)win * words];
            const uint64_t kr = score_key(r[0]), kw = score_key(w[0]);
            if (kr > kw || (kr == kw && r[1] < w[1])) win = i;
        }
        if (win < 0) { status = failf(ABO_EINVAL, "abo_mgpu_cand_qei: the candidate set is empty"); break; }
        const double* w = &recs[(size_t)win * words];
        const int64_t gidx = (int64_t)w[1];
        const double mu = w[2];
        const double* x = w + 3;
        ei_out[j] = w[0];
        idx_out[j] = gidx;
        memcpy(x_out + (size_t)j * d, x, sizeof(double) * d);
        // the last pick conditions nothing: set and models are rolled back below, so its fantasy append and the O(N·M)
        // down-date pass behind it (one ninth of a config-5 step) would only be thrown away
        if (j == q - 1) break;
        abo_gp* nw[MAXDEV] = {nullptr};
        status = run_all(mg->cs->wk, n, [&](int i) -> int32_t {
            int64_t inf = 0;
            int32_t r;
            if (mg->p_out > 1) {
                // gradient-enhanced model: the fantasy observation is the posterior mean of ALL p outputs at x — evaluated on
                // every device from its own (identical) model, so nothing more has to be exchanged
                double yv[abo::MAX_P];
                r = abo_predict_grad(cur[i], x, 1, d, ABO_HOST, yv, nullptr, ABO_HOST);
                if (!r) r = abo_append_grad(cur[i], x, d, yv, &inf, &nw[i]);
            } else {
                r = abo_append(cur[i], x, d, mu, &inf, &nw[i]);          // fantasy observation y = μ(x): β = 0
            }
            if (r) return r;
            r = abo_cand_downdate(nw[i], mc->c[i]);
            if (!r && distinct && gidx >= mc->lo[i] && gidx < mc->lo[i + 1]) r = abo_cand_exclude(nw[i], mc->c[i], gidx - mc->lo[i]);
            return r;
        });
        if (status) keep = abo::last_error_text();
        for (int i = 0; i < n; ++i) {
            if (nw[i]) {
                if (cur[i] != mg->gp[i]) abo_destroy(cur[i]);
                cur[i] = nw[i];
            }
        }
    }
    if (status && keep.empty()) keep = abo::last_error_text();
    for (int i = 0; i < n; ++i)
        if (cur[i] != mg->gp[i]) abo_destroy(cur[i]);
    rc = run_all(mg->cs->wk, n, [&](int i) -> int32_t { return abo_cand_restore(mg->gp[i], mc->c[i]); });
    mc->qei_last.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (status) return abo::set_error(status, keep.c_str());
    return rc;
}

int32_t abo_mgpu_cand_qei_stats(abo_mgpu* mg, abo_mcand* mc, abo_qei_stats* out) {
    if (!mg || !mc || !out) return failf(ABO_EINVAL, "abo_mgpu_cand_qei_stats: null argument");
    *out = mc->qei_last;
    return ABO_OK;
}

}  // extern "C"
