// loadgen.cpp — concurrent callers for the request coalescer (measurement tool, part of libpairec_host.so).
//
// Stands in for what a pairec process does to its plug-ins: many goroutines, each blocked in ONE request at a time
// (service/recall.go:129-145, service/rank/rank_service.go:264-289).  `callers` host threads run closed loops of
// single-request calls through the C ABI (pg_coalescer_recommend / pg_coalescer_recall), cycling through the given
// user vectors; per-request latencies are collected per thread and merged.  bench.py's "concurrent_callers" line
// and tests/test_gpu_coalescer.py use it; nothing here is on the product path.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pairec_gpu.h"

extern "C" {

typedef struct {
    uint64_t requests;            // completed requests
    uint64_t errors;              // calls that returned non-zero
    double   seconds;             // wall time of the measured window
    double   p50_ms, p90_ms, p99_ms, max_ms, mean_ms;
    uint64_t checksum;            // sum over requests of (first row id + count): run-to-run comparable
} ph_loadgen_result;

// mode 0: pg_coalescer_recommend(top_n); mode 1: pg_coalescer_recall.
// Every caller issues `warmup` unmeasured requests, waits at a barrier, then loops until `seconds` have elapsed.
int ph_loadgen_run(pg_coalescer* c, int mode, const float* user_vecs, uint32_t n_users, uint32_t dim, uint32_t k,
                   uint32_t top_n, uint32_t callers, uint32_t warmup, double seconds, ph_loadgen_result* out) {
    if (!c || !user_vecs || !out || n_users == 0 || callers == 0) return -1;
    using Clock = std::chrono::steady_clock;
    std::atomic<uint32_t> ready{0};
    std::atomic<bool> go{false}, stop{false};
    std::vector<std::vector<float>> lat(callers);
    std::vector<uint64_t> sums(callers, 0), errs(callers, 0);
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < callers; ++t) {
        th.emplace_back([&, t]() {
            const uint32_t n_out = mode == 0 ? top_n : k;
            std::vector<uint64_t> rows(n_out);
            std::vector<float> rec(n_out), rnk(n_out);
            std::vector<double> fus(n_out);
            uint32_t u = t % n_users;
            auto one = [&]() -> int {
                uint32_t cnt = 0;
                const float* v = user_vecs + (size_t)u * dim;
                u = (u + callers) % n_users;
                int rc = mode == 0 ? pg_coalescer_recommend(c, v, top_n, rows.data(), rec.data(), rnk.data(), fus.data(), &cnt)
                                   : pg_coalescer_recall(c, v, rows.data(), rec.data(), &cnt);
                if (rc == 0) sums[t] += rows[0] + cnt;
                else errs[t]++;
                return rc;
            };
            for (uint32_t i = 0; i < warmup; ++i) one();
            sums[t] = 0;
            errs[t] = 0;
            ready.fetch_add(1);
            while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
            lat[t].reserve(4096);
            while (!stop.load(std::memory_order_relaxed)) {
                const auto t0 = Clock::now();
                one();
                lat[t].push_back(std::chrono::duration<float, std::milli>(Clock::now() - t0).count());
            }
        });
    }
    while (ready.load() < callers) std::this_thread::yield();
    const auto t0 = Clock::now();
    go.store(true, std::memory_order_release);
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    stop.store(true);
    for (auto& x : th) x.join();
    const double el = std::chrono::duration<double>(Clock::now() - t0).count();
    std::vector<float> all;
    uint64_t cs = 0, er = 0;
    for (uint32_t t = 0; t < callers; ++t) {
        all.insert(all.end(), lat[t].begin(), lat[t].end());
        cs += sums[t];
        er += errs[t];
    }
    std::sort(all.begin(), all.end());
    memset(out, 0, sizeof *out);
    out->requests = all.size();
    out->errors = er;
    out->seconds = el;
    out->checksum = cs;
    if (!all.empty()) {
        auto pct = [&](double p) { return (double)all[std::min(all.size() - 1, (size_t)(p * all.size()))]; };
        out->p50_ms = pct(0.50);
        out->p90_ms = pct(0.90);
        out->p99_ms = pct(0.99);
        out->max_ms = all.back();
        double s = 0;
        for (float x : all) s += x;
        out->mean_ms = s / all.size();
    }
    return 0;
}

}  // extern "C"
