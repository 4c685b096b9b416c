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

// What a caller issues (one call = one "request" in the result):
//   mode 0  pg_coalescer_recommend(top_n)            user vector u
//   mode 1  pg_coalescer_recall                      user vector u
//   mode 2  pg_coalescer_rank_fm2t                   user vector u + its field ids, `rank_items` candidate rows drawn from cand_pool
//   mode 3  pg_coalescer_dpp                         `rank_items` candidate rows from cand_pool, relevance rel_pool (descending), *dpp
//   mode 4  pg_coalescer_i2i_recall                  trigger row cand_pool[u]
//   mode 5  pg_coalescer_online_recall               user vector u (width dim = the query model's d_user)
//   mode 6  pg_coalescer_rank_dnn3                   user vector u, `rank_items` candidate rows from cand_pool
typedef struct {
    int mode;
    const float* user_vecs;       // [n_users][dim]
    uint32_t n_users, dim, k, top_n;
    const int32_t* user_field_ids;   // [n_users][n_user_fields] (mode 2)
    uint32_t n_user_fields;
    const uint32_t* cand_pool;    // candidate rows; caller t's request i uses a window starting at a pseudo-random offset
    uint32_t pool_size, rank_items;
    const double* rel_pool;       // [rank_items] relevance scores, descending (mode 3)
    const pg_dpp_options* dpp;    // mode 3
} ph_loadgen_spec;

// Every caller issues `warmup` unmeasured requests, waits at a barrier, then loops until `seconds` have elapsed — or, with
// max_requests > 0, until the callers together have STARTED that many requests (a fixed amount of work: bench.py's router
// mode times exactly steps x 256 x replicas requests).  `r` != NULL sends modes 0 / 1 through the replica router instead of `c`.
int ph_loadgen_run_target(pg_coalescer* c, pg_router* r, const ph_loadgen_spec* sp, uint32_t callers, uint32_t warmup,
                          double seconds, uint64_t max_requests, ph_loadgen_result* out) {
    if ((!c && !r) || !sp || !out || callers == 0) return -1;
    if (r && sp->mode != 0 && sp->mode != 1) return -1;
    const int mode = sp->mode;
    const float* user_vecs = sp->user_vecs;
    const uint32_t n_users = sp->n_users, dim = sp->dim, k = sp->k, top_n = sp->top_n;
    if ((mode != 3 && (!user_vecs || n_users == 0)) || ((mode == 2 || mode == 3 || mode == 4 || mode == 6) && (!sp->cand_pool || sp->pool_size < sp->rank_items + 1)))
        return -1;
    using Clock = std::chrono::steady_clock;
    std::atomic<uint32_t> ready{0};
    std::atomic<bool> go{false}, stop{false};
    std::atomic<uint64_t> started{0};
    std::vector<std::vector<float>> lat(callers);
    std::vector<uint64_t> sums(callers, 0), errs(callers, 0);
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < callers; ++t) {
        th.emplace_back([&, t]() {
            const uint32_t n_out = std::max<uint32_t>(1, mode == 0 ? top_n : (mode == 2 || mode == 3 || mode == 6 ? sp->rank_items : k));
            std::vector<uint64_t> rows(n_out);
            std::vector<float> rec(n_out), rnk(n_out);
            std::vector<double> fus(n_out);
            std::vector<uint32_t> idx(n_out);
            uint32_t u = n_users ? t % n_users : 0;
            uint64_t rng = 0x9E3779B97F4A7C15ull * (t + 1);
            auto one = [&]() -> int {
                uint32_t cnt = 0;
                const float* v = user_vecs ? user_vecs + (size_t)u * dim : nullptr;
                const uint32_t u_now = u;
                if (n_users) u = (u + callers) % n_users;
                rng = rng * 6364136223846793005ull + 1442695040888963407ull;
                const uint32_t off = sp->pool_size > sp->rank_items ? (uint32_t)((rng >> 33) % (sp->pool_size - sp->rank_items)) : 0u;
                int rc;
                switch (mode) {
                    case 0:
                        rc = r ? pg_router_recommend(r, v, top_n, rows.data(), rec.data(), rnk.data(), fus.data(), &cnt)
                               : pg_coalescer_recommend(c, v, top_n, rows.data(), rec.data(), rnk.data(), fus.data(), &cnt);
                        break;
                    case 1:
                        rc = r ? pg_router_recall(r, v, rows.data(), rec.data(), &cnt) : pg_coalescer_recall(c, v, rows.data(), rec.data(), &cnt);
                        break;
                    case 2:
                        rc = pg_coalescer_rank_fm2t(c, v, sp->user_field_ids + (size_t)u_now * sp->n_user_fields, sp->cand_pool + off, sp->rank_items,
                                                    rec.data());
                        rows[0] = (uint64_t)(rec[0] * 1e6f);
                        cnt = sp->rank_items;
                        break;
                    case 3:
                        rc = pg_coalescer_dpp(c, sp->cand_pool + off, sp->rel_pool, sp->rank_items, sp->dpp, nullptr, idx.data(), &cnt, nullptr);
                        rows[0] = idx[0];
                        break;
                    case 4: rc = pg_coalescer_i2i_recall(c, sp->cand_pool[off], rows.data(), rec.data(), &cnt); break;
                    case 5: rc = pg_coalescer_online_recall(c, v, rows.data(), rec.data(), &cnt); break;
                    default:
                        rc = pg_coalescer_rank_dnn3(c, v, sp->cand_pool + off, sp->rank_items, rec.data());
                        rows[0] = (uint64_t)(rec[0] * 1e6f);
                        cnt = sp->rank_items;
                }
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
                if (max_requests && started.fetch_add(1, std::memory_order_relaxed) >= max_requests) break;
                const auto t0 = Clock::now();
                one();
                lat[t].push_back(std::chrono::duration<float, std::milli>(Clock::now() - t0).count());
            }
        });
    }
    while (ready.load() < callers) std::this_thread::yield();
    const auto t0 = Clock::now();
    go.store(true, std::memory_order_release);
    if (max_requests) {
        // fixed work: the callers stop by themselves; `seconds` is only a bound against a stuck device
        const auto deadline = t0 + std::chrono::duration<double>(seconds > 0 ? seconds : 600.0);
        while (started.load(std::memory_order_relaxed) < max_requests && Clock::now() < deadline)
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        if (started.load() < max_requests) stop.store(true);
    } else {
        std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
        stop.store(true);
    }
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

int ph_loadgen_run_ex(pg_coalescer* c, const ph_loadgen_spec* sp, uint32_t callers, uint32_t warmup, double seconds,
                      ph_loadgen_result* out) {
    return ph_loadgen_run_target(c, nullptr, sp, callers, warmup, seconds, 0, out);
}

// the two original modes (0 recommend, 1 recall)
int ph_loadgen_run(pg_coalescer* c, int mode, const float* user_vecs, uint32_t n_users, uint32_t dim, uint32_t k,
                   uint32_t top_n, uint32_t callers, uint32_t warmup, double seconds, ph_loadgen_result* out) {
    if (mode != 0 && mode != 1) return -1;
    ph_loadgen_spec sp;
    memset(&sp, 0, sizeof sp);
    sp.mode = mode;
    sp.user_vecs = user_vecs;
    sp.n_users = n_users;
    sp.dim = dim;
    sp.k = k;
    sp.top_n = top_n;
    return ph_loadgen_run_ex(c, &sp, callers, warmup, seconds, out);
}

}  // extern "C"
