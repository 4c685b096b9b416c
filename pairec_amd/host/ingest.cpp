// ingest.cpp — real item ids and table ingestion for the host mirror (part of libpairec_host.so).
//
// pairec's items are strings (module.ItemId, module/item.go:13): the FAISS reply carries `labels string[K]` beside the
// row numbers (algorithm/faiss/vectorretrieval.proto:11-20), VectorRecall builds its items from them
// (service/recall/vector_recall.go:93-102), and the vector tables themselves arrive as (item_id, embedding) rows of a
// Hologres partition that is replaced wholesale (module/vector_hologres_dao.go:40-61).  The device side works on row
// numbers; this file is the other half:
//   * IdDict     row ↔ item id for up to 2^32 - 2 items: the ids back to back in one arena, an offset per row, and an
//                open-addressing table of row numbers (linear probing, load <= 0.5) built by several threads with CAS;
//   * ingestion  a new table generation is streamed in chunks of (ids, fp32 rows) into a STAGING table while the live one
//                keeps serving — through the coalescer, nothing waits — then committed: the staging table's shadow is
//                built, pg_table_swap exchanges the two tables (exclusive against every enqueue, device drained) and the
//                dictionaries change over in the same critical section of the engine's version lock, which every
//                request holds shared from its first plug-in call to its last label lookup: a request sees ONE
//                generation of rows and ids.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <thread>

#include "pairec_host.hpp"

namespace pairec {

static inline uint64_t hash_bytes(const char* p, size_t n) {
    uint64_t h = 0xcbf29ce484222325ull;                       // FNV-1a, then a finaliser (ids share long prefixes)
    for (size_t i = 0; i < n; ++i) h = (h ^ (unsigned char)p[i]) * 0x100000001b3ull;
    h ^= h >> 32;
    h *= 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    return h;
}

void IdDict::Reserve(uint64_t rows, uint64_t id_bytes) {
    off_.reserve(rows + 1);
    arena_.reserve(id_bytes);
    if (off_.empty()) off_.push_back(0);
}

void IdDict::Append(const char* id, size_t len) {
    if (off_.empty()) off_.push_back(0);
    arena_.insert(arena_.end(), id, id + len);
    off_.push_back(arena_.size());
}

bool IdDict::BuildIndex(std::string* err) {
    const uint64_t n = size();
    if (n >= 0xFFFFFFFEull) { if (err) *err = "IdDict: too many ids"; return false; }
    uint64_t cap = 16;
    while (cap < 2 * n) cap <<= 1;
    mask_ = cap - 1;
    slots_.reset(new std::atomic<uint32_t>[cap]);
    const unsigned hw = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    const unsigned nt = n < 100000 ? 1 : hw;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&, t]() { for (uint64_t i = cap * t / nt; i < cap * (t + 1) / nt; ++i) slots_[i].store(kEmpty, std::memory_order_relaxed); });
    for (auto& x : th) x.join();
    th.clear();
    std::atomic<uint64_t> dup{kEmpty};
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&, t]() {
            for (uint64_t r = n * t / nt; r < n * (t + 1) / nt; ++r) {
                const char* p = arena_.data() + off_[r];
                const size_t len = (size_t)(off_[r + 1] - off_[r]);
                uint64_t s = hash_bytes(p, len) & mask_;
                for (;;) {
                    uint32_t cur = slots_[s].load(std::memory_order_acquire);
                    if (cur == kEmpty) {
                        if (slots_[s].compare_exchange_strong(cur, (uint32_t)r, std::memory_order_acq_rel)) break;
                    }
                    // occupied (possibly by the CAS we just lost): the same id twice is an error, anything else probes on
                    const size_t olen = (size_t)(off_[cur + 1] - off_[cur]);
                    if (olen == len && !memcmp(arena_.data() + off_[cur], p, len)) {
                        dup.store(r, std::memory_order_relaxed);
                        break;
                    }
                    s = (s + 1) & mask_;
                }
            }
        });
    for (auto& x : th) x.join();
    if (dup.load() != kEmpty) {
        if (err) *err = "IdDict: duplicate item id \"" + std::string(IdOf(dup.load())) + "\"";
        return false;
    }
    return true;
}

bool IdDict::RowOf(const char* id, size_t len, uint32_t* row) const {
    if (!slots_) return false;
    uint64_t s = hash_bytes(id, len) & mask_;
    for (;;) {
        const uint32_t cur = slots_[s].load(std::memory_order_relaxed);
        if (cur == kEmpty) return false;
        const size_t olen = (size_t)(off_[cur + 1] - off_[cur]);
        if (olen == len && !memcmp(arena_.data() + off_[cur], id, len)) {
            *row = cur;
            return true;
        }
        s = (s + 1) & mask_;
    }
}

// ---- engine: id mapping -----------------------------------------------------------------------------------------
bool Engine::RowOfId(const std::string& id, uint32_t* row) const {
    if (std::shared_ptr<const IdDict> d = std::atomic_load(&dict)) return d->RowOf(id.data(), id.size(), row) && *row < table_rows;
    if (id.compare(0, id_prefix.size(), id_prefix) != 0) return false;
    char* e = nullptr;
    const unsigned long long r = strtoull(id.c_str() + id_prefix.size(), &e, 10);
    if (e == id.c_str() + id_prefix.size() || *e != '\0' || r >= table_rows) return false;
    *row = (uint32_t)r;
    return true;
}

std::string Engine::IdOfRow(uint64_t row) const {
    if (std::shared_ptr<const IdDict> d = std::atomic_load(&dict))
        if (row < d->size()) return std::string(d->IdOf(row));
    return id_prefix + std::to_string(row);
}

// ---- engine: ingestion --------------------------------------------------------------------------------------------
bool Engine::IngestBegin(std::string* err) {
    std::lock_guard<std::mutex> g(ingest_mu);
    if (!staging && pg_table_create(ctx, table_rows, dim, 0, &staging) != PG_OK) {
        if (err) *err = std::string("pg_table_create (staging): ") + pg_last_error();
        staging = nullptr;
        return false;
    }
    staging_dict.reset(new IdDict());
    staging_dict->Reserve(table_rows, table_rows * 16);
    staging_filled = 0;
    if (staging_feats) { pg_features_destroy(ctx, staging_feats); staging_feats = nullptr; }    // an abandoned load's columns
    return true;
}

// One int32 item feature column of the generation being loaded (between IngestBegin and IngestCommit), keyed by the NEW
// rows.  The commit changes the columns over together with rows and ids; a generation that replaces a table with
// feature columns must bring its own (IngestCommit refuses otherwise: the old columns describe the old rows).
bool Engine::IngestFeatureColumn(const std::string& name, const int32_t* values, uint64_t n, std::string* err) {
    std::lock_guard<std::mutex> g(ingest_mu);
    if (!staging || !staging_dict) { if (err) *err = "ingest: no load in progress (IngestBegin first)"; return false; }
    if (n != table_rows || !values) { if (err) *err = "ingest: a feature column has one value per table row"; return false; }
    if (!staging_feats && pg_features_create(ctx, table_rows, &staging_feats) != PG_OK) {
        if (err) *err = std::string("pg_features_create (staging): ") + pg_last_error();
        staging_feats = nullptr;
        return false;
    }
    if (pg_features_set_column(ctx, staging_feats, name.c_str(), PG_F_I32, values, 0.0) != PG_OK) {
        if (err) *err = std::string("pg_features_set_column: ") + pg_last_error();
        return false;
    }
    return true;
}

bool Engine::IngestChunk(const char* ids, size_t ids_bytes, const float* rows, uint64_t n, std::string* err) {
    std::lock_guard<std::mutex> g(ingest_mu);
    if (!staging || !staging_dict) { if (err) *err = "ingest: no load in progress (IngestBegin first)"; return false; }
    if (staging_filled + n > table_rows) { if (err) *err = "ingest: more rows than the table holds"; return false; }
    // n NUL-terminated ids back to back — exactly n: an id with a NUL inside would shift every later id by a row
    const char* p = ids;
    const char* end = ids + ids_bytes;
    for (uint64_t i = 0; i < n; ++i) {
        const char* z = (const char*)memchr(p, 0, (size_t)(end - p));
        if (!z) { if (err) *err = "ingest: fewer ids than rows in the chunk"; return false; }
        p = z + 1;
    }
    if (p != end) { if (err) *err = "ingest: the chunk's id buffer holds more than one id per row (an id with a NUL byte inside?)"; return false; }
    // the rows go straight into the staging table; the live table keeps serving (its lock is not touched)
    if (pg_table_upload(ctx, staging, staging_filled, n, rows) != PG_OK) {
        if (err) *err = std::string("pg_table_upload: ") + pg_last_error();
        return false;
    }
    // the dictionary follows the rows only once they are in place (a failed upload leaves both where they were)
    p = ids;
    for (uint64_t i = 0; i < n; ++i) {
        const char* z = (const char*)memchr(p, 0, (size_t)(end - p));
        staging_dict->Append(p, (size_t)(z - p));
        p = z + 1;
    }
    staging_filled += n;
    return true;
}

bool Engine::IngestCommit(std::string* err) {
    std::lock_guard<std::mutex> g(ingest_mu);
    if (!staging || !staging_dict) { if (err) *err = "ingest: no load in progress"; return false; }
    if (staging_filled != table_rows) {
        if (err) *err = "ingest: " + std::to_string(staging_filled) + " of " + std::to_string(table_rows) + " rows loaded (a partition is replaced whole)";
        return false;
    }
    if (feats && pg_features_num_columns(feats) > 0 && !staging_feats) {
        if (err) *err = "ingest: the engine serves row-keyed feature columns (a WhereClause column / FM item fields); the new generation "
                        "must bring its own (ph_engine_ingest_feature_column between begin and commit) — the old ones describe the old rows";
        return false;
    }
    if (!staging_dict->BuildIndex(err)) return false;
    // the new generation's shadow, outside the critical section (the live table keeps serving meanwhile)
    if (pg_table_screen_info(ctx, staging, nullptr, nullptr, nullptr) != PG_OK) {
        if (err) *err = std::string("pg_table_screen_info: ") + pg_last_error();
        return false;
    }
    {
        VersionLock::Write w(version);          // no request is between its first plug-in call and its last label lookup
        if (pg_table_swap(ctx, table, staging) != PG_OK) {
            if (err) *err = std::string("pg_table_swap: ") + pg_last_error();
            return false;
        }
        std::atomic_store(&dict, std::shared_ptr<const IdDict>(staging_dict.release()));
        if (staging_feats) {
            // rows, ids and columns change over together; whatever was built over the old columns (the scene coalescer's FM
            // algorithm, filtered views) goes with them — nothing is in flight inside this section
            DropCoalescers();
            std::swap(feats, staging_feats);
        }
        generation++;
    }
    if (staging_feats) { pg_features_destroy(ctx, staging_feats); staging_feats = nullptr; }   // the previous generation's columns
    staging_filled = 0;                         // `staging` now holds the previous generation's rows: the next load overwrites them
    return true;
}

// Text: one item per line, `item_id<TAB>v1,v2,...,vD` (the Hologres "{v1,...}" braces are accepted).
// Binary: "PGT1" u32 dim u64 rows, then per item u16 id length, the id bytes, dim fp32 values.
bool Engine::IngestFile(const std::string& path, std::string* err) {
    std::ifstream in(path, std::ios::binary);
    if (!in) { if (err) *err = "ingest: cannot open " + path; return false; }
    char magic[4] = {0, 0, 0, 0};
    in.read(magic, 4);
    const bool binary = in.gcount() == 4 && !memcmp(magic, "PGT1", 4);
    if (!IngestBegin(err)) return false;
    const uint64_t chunk = 65536;
    std::string ids;
    std::vector<float> rows;
    rows.reserve((size_t)chunk * dim);
    uint64_t in_chunk = 0;
    auto flush = [&]() -> bool {
        if (!in_chunk) return true;
        const bool ok = IngestChunk(ids.data(), ids.size(), rows.data(), in_chunk, err);
        ids.clear();
        rows.clear();
        in_chunk = 0;
        return ok;
    };
    if (binary) {
        uint32_t fdim = 0;
        uint64_t frows = 0;
        in.read((char*)&fdim, 4);
        in.read((char*)&frows, 8);
        if (!in || fdim != dim || frows != table_rows) { if (err) *err = "ingest: " + path + ": header does not match the table (dim / rows)"; return false; }
        std::string id;
        for (uint64_t r = 0; r < frows; ++r) {
            uint16_t len = 0;
            in.read((char*)&len, 2);
            id.resize(len);
            in.read(&id[0], len);
            const size_t o = rows.size();
            rows.resize(o + dim);
            in.read((char*)&rows[o], (std::streamsize)dim * 4);
            if (!in) { if (err) *err = "ingest: " + path + ": truncated at row " + std::to_string(r); return false; }
            if (id.find('\0') != std::string::npos) { if (err) *err = "ingest: " + path + ": the id of row " + std::to_string(r) + " contains a NUL byte"; return false; }
            ids.append(id);
            ids.push_back('\0');
            if (++in_chunk == chunk && !flush()) return false;
        }
    } else {
        in.clear();
        in.seekg(0);
        std::string line;
        uint64_t lineno = 0;
        while (std::getline(in, line)) {
            ++lineno;
            if (line.empty()) continue;
            const size_t tab = line.find('\t');
            if (tab == std::string::npos) { if (err) *err = "ingest: " + path + ":" + std::to_string(lineno) + ": no TAB between id and vector"; return false; }
            const char* p = line.c_str() + tab + 1;
            uint32_t got = 0;
            while (*p && got < dim) {
                while (*p == '{' || *p == ',' || *p == ' ') ++p;
                if (!*p || *p == '}') break;
                char* e = nullptr;
                const float v = strtof(p, &e);
                if (e == p) break;
                rows.push_back(v);
                ++got;
                p = e;
            }
            while (*p == ',' || *p == ' ' || *p == '}' || *p == '\r') ++p;
            if (got != dim || *p) { if (err) *err = "ingest: " + path + ":" + std::to_string(lineno) + ": " + (got != dim ? std::to_string(got) : "more than " + std::to_string(dim)) + " values, the table has dim " + std::to_string(dim); return false; }
            ids.append(line, 0, tab);
            ids.push_back('\0');
            if (++in_chunk == chunk && !flush()) return false;
        }
    }
    if (!flush()) return false;
    return IngestCommit(err);
}

}  // namespace pairec

extern "C" {

extern thread_local std::string g_ph_err_ingest;
thread_local std::string g_ph_err_ingest;
const char* ph_ingest_last_error(void) { return g_ph_err_ingest.c_str(); }

int ph_engine_ingest_begin(void* h) {
    auto* e = (pairec::Engine*)h;
    return e && e->IngestBegin(&g_ph_err_ingest) ? 0 : -1;
}
int ph_engine_ingest_chunk(void* h, const char* ids, size_t ids_bytes, const float* rows, uint64_t n) {
    auto* e = (pairec::Engine*)h;
    return e && ids && rows && e->IngestChunk(ids, ids_bytes, rows, n, &g_ph_err_ingest) ? 0 : -1;
}
int ph_engine_ingest_feature_column(void* h, const char* name, const int32_t* values, uint64_t n) {
    auto* e = (pairec::Engine*)h;
    return e && name && e->IngestFeatureColumn(name, values, n, &g_ph_err_ingest) ? 0 : -1;
}
int ph_engine_ingest_commit(void* h) {
    auto* e = (pairec::Engine*)h;
    return e && e->IngestCommit(&g_ph_err_ingest) ? 0 : -1;
}
int ph_engine_ingest_file(void* h, const char* path) {
    auto* e = (pairec::Engine*)h;
    return e && path && e->IngestFile(path, &g_ph_err_ingest) ? 0 : -1;
}
// ids only (the rows are already in the table, e.g. generated on the device): n NUL-terminated ids back to back
int ph_engine_set_ids(void* h, const char* ids, size_t ids_bytes, uint64_t n) {
    auto* e = (pairec::Engine*)h;
    if (!e || !ids || n != e->table_rows) { g_ph_err_ingest = "set_ids: one id per table row"; return -1; }
    std::unique_ptr<pairec::IdDict> d(new pairec::IdDict());
    d->Reserve(n, ids_bytes);
    const char* p = ids;
    const char* end = ids + ids_bytes;
    for (uint64_t i = 0; i < n; ++i) {
        const char* z = (const char*)memchr(p, 0, (size_t)(end - p));
        if (!z) { g_ph_err_ingest = "set_ids: fewer ids than rows"; return -1; }
        d->Append(p, (size_t)(z - p));
        p = z + 1;
    }
    if (p != end) { g_ph_err_ingest = "set_ids: the buffer holds more than one id per row (an id with a NUL byte inside?)"; return -1; }
    if (!d->BuildIndex(&g_ph_err_ingest)) return -1;
    pairec::VersionLock::Write w(e->version);
    std::atomic_store(&e->dict, std::shared_ptr<const pairec::IdDict>(d.release()));
    e->generation++;
    return 0;
}
uint64_t ph_engine_generation(void* h) { return h ? ((pairec::Engine*)h)->generation.load() : 0; }
// row ↔ id through the engine's dictionary (tests): returns the id's length, or -1
int ph_engine_id_of_row(void* h, uint64_t row, char* out, int cap) {
    auto* e = (pairec::Engine*)h;
    if (!e || !out) return -1;
    const std::string s = e->IdOfRow(row);
    if ((int)s.size() >= cap) return -1;
    memcpy(out, s.c_str(), s.size() + 1);
    return (int)s.size();
}
long long ph_engine_row_of_id(void* h, const char* id) {
    auto* e = (pairec::Engine*)h;
    uint32_t row = 0;
    return e && id && e->RowOfId(id, &row) ? (long long)row : -1;
}

}  // extern "C"

// CPU self-test of the dictionary (tests/test_gpu_ingest.py::test_id_dictionary_on_cpu): n ids with long shared
// prefixes; every id maps to its row and back, absent ids are absent, a duplicate is refused.  0 = ok.
extern "C" int ph_id_dict_selftest(uint64_t n) {
    pairec::IdDict d;
    d.Reserve(n, n * 24);
    char buf[64];
    for (uint64_t i = 0; i < n; ++i) {
        const int len = snprintf(buf, sizeof buf, "shop/42/item-%llu-%c", (unsigned long long)(i * 7919 % n + n * (i & 1)), "ab"[i & 1]);
        d.Append(buf, (size_t)len);
    }
    std::string err;
    if (!d.BuildIndex(&err)) return 1;
    for (uint64_t i = 0; i < n; i += 97) {
        const std::string id = d.IdOf(i);
        uint32_t row = 0;
        if (!d.RowOf(id.data(), id.size(), &row) || row != i) return 2;
    }
    uint32_t row = 0;
    if (d.RowOf("shop/42/item-x", 14, &row) || d.RowOf("", 0, &row)) return 3;
    pairec::IdDict dup;
    dup.Append("a", 1);
    dup.Append("b", 1);
    dup.Append("a", 1);
    if (dup.BuildIndex(&err)) return 4;
    return 0;
}
