// CPU unit test of the multi-device exchange's bounded gather (abstractbayesopt.jl_amd/csrc/abo_exchange.h) with a STUBBED transport:
// the case the design exists for and the one-GPU box can never run — n > 1 shards, one host thread each, a peer blocked inside the
// collective being released by the abort of its own communicator, several shards calling abort at once.
// The stub models a collective: shard i's part completes only once EVERY shard has enqueued and none is stuck; abort(i) makes
// shard i's part exit (what ncclCommAbort does to a collective kernel waiting for a missing peer).
//   build: g++ -std=c++17 -O1 -pthread -I abstractbayesopt.jl_amd/csrc tests/exchange_stub.cpp -o tests/_build/exchange_stub
#include <atomic>
#include <chrono>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

#include "abo_exchange.h"

struct Stub {
    int n;
    int never_enqueues = -1;         // this shard fails before enqueueing
    int stuck = -1;                  // this shard's part never completes (everybody enqueues; nobody can finish)
    int stream_error = -1;           // this shard's stream reports an error while polling
    std::atomic<int> enqueued{0};
    std::vector<std::atomic<int>> aborted, drained, polls;
    explicit Stub(int n_) : n(n_), aborted(n_), drained(n_), polls(n_) {}
    bool enqueue(int i, std::string* err) {
        if (i == never_enqueues) { *err = "stub: failed before its all-gather"; return false; }
        enqueued.fetch_add(1);
        return true;
    }
    int poll(int i) {
        polls[i].fetch_add(1);
        if (i == stream_error && polls[i].load() > 3) return -1;
        if (aborted[i].load()) return 0;                                   // an aborted communicator's kernel has exited
        const bool all_in = enqueued.load() == n && stuck < 0;
        return all_in ? 0 : 1;
    }
    void abort(int i) { aborted[i].fetch_add(1); }
    void drain(int i) { drained[i].fetch_add(1); }
};

static void run_threads(int n, const std::function<void(int)>& f) {
    std::vector<std::thread> th;
    for (int i = 0; i < n; ++i) th.emplace_back([&, i] { f(i); });
    for (auto& t : th) t.join();
}

static int failures = 0;
#define CHECK(c, ...) do { if (!(c)) { ++failures; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

int main() {
    for (int n : {2, 3, 8}) {
        auto run = [n](const std::function<void(int)>& f) { run_threads(n, f); };
        {   // clean: everybody completes, nobody aborts
            Stub s(n);
            const auto t0 = std::chrono::steady_clock::now();
            const abo::GatherVerdict v = abo::bounded_gather(s, n, 5000, run);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            CHECK(v.aborted == 0 && v.first_fault.empty(), "n=%d clean: aborted %d (%s)", n, v.aborted, v.first_fault.c_str());
            for (int i = 0; i < n; ++i) CHECK(s.aborted[i].load() == 0 && s.drained[i].load() == 0, "n=%d clean: shard %d aborted", n, i);
            CHECK(ms < 2000.0, "n=%d clean: took %.0f ms", n, ms);
        }
        {   // a shard fails after the vote, before enqueueing: its peers sit in the collective until THEIR OWN abort releases them
            Stub s(n);
            s.never_enqueues = n - 1;
            const auto t0 = std::chrono::steady_clock::now();
            const abo::GatherVerdict v = abo::bounded_gather(s, n, 5000, run);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            CHECK(v.aborted == 1, "n=%d missing peer: verdict %d", n, v.aborted);
            CHECK(v.first_fault.find("shard") == 0 && v.first_fault.find("before its all-gather") != std::string::npos, "n=%d missing peer: fault text '%s'", n, v.first_fault.c_str());
            for (int i = 0; i < n; ++i)
                CHECK(s.aborted[i].load() == 1 && s.drained[i].load() == 1, "n=%d missing peer: shard %d aborted %d drained %d (each exactly once, by itself)", n, i,
                      s.aborted[i].load(), s.drained[i].load());
            CHECK(ms < 2000.0, "n=%d missing peer: released after %.0f ms (the abort word, not the 5 s deadline)", n, ms);
        }
        {   // a collective that never completes: the deadline raises the abort word, every shard aborts its own communicator
            Stub s(n);
            s.stuck = 0;
            const auto t0 = std::chrono::steady_clock::now();
            const abo::GatherVerdict v = abo::bounded_gather(s, n, 300, run);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            CHECK(v.aborted == 2, "n=%d stuck: verdict %d", n, v.aborted);
            for (int i = 0; i < n; ++i) CHECK(s.aborted[i].load() == 1 && s.drained[i].load() == 1, "n=%d stuck: shard %d aborted %d", n, i, s.aborted[i].load());
            CHECK(ms >= 250.0 && ms < 3000.0, "n=%d stuck: returned after %.0f ms for a 300 ms bound", n, ms);
        }
        {   // a stream error on one shard while the others wait: abort word, everybody out
            Stub s(n);
            s.stuck = 1 % n;
            s.stream_error = 0;
            const abo::GatherVerdict v = abo::bounded_gather(s, n, 5000, run);
            CHECK(v.aborted == 1 && v.first_fault.find("stream reported an error") != std::string::npos, "n=%d stream error: verdict %d '%s'", n, v.aborted,
                  v.first_fault.c_str());
            for (int i = 0; i < n; ++i) CHECK(s.aborted[i].load() == 1, "n=%d stream error: shard %d aborted %d", n, i, s.aborted[i].load());
        }
    }
    if (failures) { printf("%d check(s) failed\n", failures); return 1; }
    printf("exchange stub: all checks passed (n = 2, 3, 8: clean, missing peer, stuck collective, stream error)\n");
    return 0;
}
