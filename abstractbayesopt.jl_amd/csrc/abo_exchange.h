// The bounded gather of the multi-device exchange (mgpu.hip: exchange, step 2) as a protocol over an abstract transport — no HIP,
// no RCCL in this header, so that the part that must never hang can be driven on a CPU with a stubbed transport
// (tests/exchange_stub.cpp: a peer that never enqueues, a collective that never completes, n > 1 shards, one thread per shard).
//
// Every shard, on its own thread:
//     enqueue(i)          its part of the collective (false: it failed before enqueueing — the peers' parts would wait for it forever)
//     poll(i) until done  0 = completed, 1 = not yet, −1 = its stream reports an error;
//                         between polls it looks at a shared ABORT WORD and at a DEADLINE
//     on failure / abort word / deadline: raise the abort word, abort(i) — take this shard's communicator down, which makes whatever
//                         it has in flight exit — and drain(i)
// Returns 0 when every shard completed; 1 when the gather was aborted (why: the first fault seen, or the time-out) — the caller then
// releases the remaining communicators and completes the exchange through the host.  No shard can be left blocked: a shard waits
// only in poll loops that watch the abort word and the deadline, and abort(i) is called by the shard itself, on its own communicator.
#pragma once
#include <atomic>
#include <chrono>
#include <cstdio>
#include <functional>
#include <mutex>
#include <string>
#include <thread>

namespace abo {

struct GatherVerdict {
    int aborted = 0;          // 0 completed; 1 a shard failed after the vote; 2 the deadline passed
    std::string first_fault;  // "shard i: what"
};

// run(f): f(i) for every shard i < n concurrently, returns when all have returned (mgpu.hip: run_all on the list's workers)
template <class Transport>
GatherVerdict bounded_gather(Transport& T, int n, long timeout_ms,
                             const std::function<void(const std::function<void(int)>&)>& run) {
    GatherVerdict v;
    std::atomic<int> abort_word{0};
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
    std::mutex fmu;
    auto give_up = [&](int i, const char* what) {
        abort_word.store(1, std::memory_order_release);
        std::lock_guard<std::mutex> lk(fmu);
        if (v.first_fault.empty()) { char b[256]; snprintf(b, sizeof b, "shard %d: %s", i, what); v.first_fault = b; }
    };
    std::atomic<int> failed{0};
    run([&](int i) {
        std::string err;
        bool enq = T.enqueue(i, &err);
        if (!enq) give_up(i, err.empty() ? "failed before its all-gather" : err.c_str());
        int w = 1;                                            // 0 completed, 1 aborted / timed out, −1 stream error
        if (enq) {
            for (unsigned spin = 0;; ++spin) {
                const int q = T.poll(i);
                if (q == 0) { w = 0; break; }
                if (q < 0) { w = -1; break; }
                if (abort_word.load(std::memory_order_acquire)) { w = 1; break; }
                if (std::chrono::steady_clock::now() > deadline) { abort_word.store(2, std::memory_order_release); w = 1; break; }
                if (spin < 4000) std::this_thread::yield();
                else std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
        }
        if (w == 0) return;
        if (w < 0) give_up(i, "its stream reported an error during the collective");
        failed.fetch_add(1);
        T.abort(i);                                           // this shard's communicator: whatever it has in flight exits
        T.drain(i);
    });
    if (failed.load()) v.aborted = abort_word.load() == 2 ? 2 : 1;
    return v;
}

}  // namespace abo
