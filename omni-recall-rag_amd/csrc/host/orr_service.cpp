// orr_service.cpp -- C++ mirrors of InMemoryIngestionStore and of
// RecallSearchService.SearchAsync with the scoring replaced by the HIP library
// (include/omnirecall_host.h).  What a .NET host does in C#  in the
// language available here.
#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../../include/omnirecall_hip.h"
#include "../../../include/omnirecall_host.h"

#include "orr_store.h"

namespace orrh_detail {
thread_local std::string g_err;
int fail(int code, const std::string &msg) { g_err = msg; return code; }
}  // namespace orrh_detail

namespace orrh_detail {

void json_string(const std::string &s, std::string &out)
{
    out.push_back('"');
    for (unsigned char c : s) {
        switch (c) {
        case '"': out += "\\\""; break;
        case '\\': out += "\\\\"; break;
        case '\n': out += "\\n"; break;
        case '\r': out += "\\r"; break;
        case '\t': out += "\\t"; break;
        case '\b': out += "\\b"; break;
        case '\f': out += "\\f"; break;
        default:
            if (c < 0x20) { char b[8]; snprintf(b, sizeof(b), "\\u%04X", c); out += b; }
            else out.push_back((char)c);
        }
    }
    out.push_back('"');
}

// DateTime (Kind=Utc) as System.Text.Json writes it: yyyy-MM-ddTHH:mm:ss[.fffffff]Z
std::string iso_utc(int64_t ticks)
{
    const int64_t tps = 10000000;
    int64_t secs = ticks / tps, frac = ticks % tps;
    int64_t days = secs / 86400, sod = secs % 86400;
    // days since 0001-01-01 -> civil date (proleptic Gregorian)
    int64_t z = days + 306;                       // shift so the era starts on 0000-03-01
    int64_t era = z / 146097, doe = z % 146097;
    int64_t yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
    int64_t y = yoe + era * 400;
    int64_t doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
    int64_t mp = (5 * doy + 2) / 153;
    int64_t d = doy - (153 * mp + 2) / 5 + 1;
    int64_t m = mp < 10 ? mp + 3 : mp - 9;
    if (m <= 2) y += 1;
    char buf[64];
    int n = snprintf(buf, sizeof(buf), "%04lld-%02lld-%02lldT%02lld:%02lld:%02lld", (long long)y, (long long)m,
                     (long long)d, (long long)(sod / 3600), (long long)((sod / 60) % 60), (long long)(sod % 60));
    std::string out(buf, (size_t)n);
    if (frac) {
        char f[16];
        snprintf(f, sizeof(f), ".%07lld", (long long)frac);
        std::string fs(f);
        while (fs.back() == '0') fs.pop_back();
        out += fs;
    }
    out.push_back('Z');
    return out;
}

}  // namespace orrh_detail

using orrh_detail::Chunk;
using orrh_detail::Document;
using orrh_detail::fail;
using orrh_detail::g_err;
using orrh_detail::iso_utc;
using orrh_detail::json_string;

struct Shard {                       // one sealed orr_index and the chunks behind its row ids
    orr_index *index = nullptr;
    std::vector<Chunk> chunks;       // row id = id_base + position in this vector
    int64_t id_base = 0;
    int64_t min_created = 0, max_created = 0;
    std::map<std::string, uint64_t> doc_stamps;      // documents with live rows here -> version of their chunk list
    std::vector<char> dead;          // rows deleted in place (orr_index_delete_rows), by position in `chunks`
    int64_t n_dead = 0;
    int64_t n_dead_compacted = 0;    // of those, rows that have left the device arrays (orr_index_compact)
};

struct orrh_service {
    orrh_store *store = nullptr;
    int32_t device = 0;
    int64_t candidate_limit = 300;
    std::mutex mu;
    std::vector<Shard> shards;       // newest first = global candidate order
    int32_t dim = 0;
    uint64_t built_version = ~0ull;
    int64_t next_id = 0;
    int64_t full_rebuilds = 0, delta_builds = 0, tombstoned_rows = 0, compactions = 0, delta_merges = 0;
};

namespace {

std::string lower(const std::string &s)
{
    std::string out(4 * s.size() + 4, '\0');
    int64_t m = orrh_lower_invariant(reinterpret_cast<const uint8_t *>(s.data()), (int64_t)s.size(),
                                     reinterpret_cast<uint8_t *>(&out[0]), (int64_t)out.size());
    out.resize(m < 0 ? 0 : (size_t)m);
    return out;
}

void json_double(double v, std::string &out)
{
    if (std::isnan(v) || std::isinf(v)) { out += "null"; return; }     // System.Text.Json would throw
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf), v);                 // shortest round-trip, like .NET "R"
    out.append(buf, r.ptr);
}

void free_shards(orrh_service *svc)
{
    for (auto &sh : svc->shards)
        if (sh.index) orr_index_destroy(sh.index);
    svc->shards.clear();
}

// One sealed shard from chunks given in store enumeration order.
int build_shard(orrh_service *svc, std::vector<Chunk> &&chunks, int32_t dim, Shard *out)
{
    Shard sh;
    sh.chunks = std::move(chunks);
    sh.id_base = svc->next_id;
    svc->next_id += (int64_t)sh.chunks.size();
    orr_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = (int32_t)sizeof(cfg);
    cfg.device = svc->device;
    cfg.dim = dim;
    int r = orr_index_create(&cfg, &sh.index);
    if (r != ORR_OK) return fail(r, orr_last_error());
    const size_t n = sh.chunks.size();
    sh.min_created = n ? sh.chunks[0].created_ticks : 0;
    sh.max_created = sh.min_created;
    size_t i = 0;
    while (i < n) {                                   // runs of rows with / without a usable embedding
        const bool has = dim > 0 && (int32_t)sh.chunks[i].embedding.size() == dim;
        size_t e = i;
        std::vector<float> emb;
        std::vector<int64_t> created, ids;
        std::vector<uint64_t> off{0};
        std::string pool;
        while (e < n && (dim > 0 && (int32_t)sh.chunks[e].embedding.size() == dim) == has && e - i < 65536) {
            const Chunk &c = sh.chunks[e];
            if (has) emb.insert(emb.end(), c.embedding.begin(), c.embedding.end());
            created.push_back(c.created_ticks);
            ids.push_back(sh.id_base + (int64_t)e);
            sh.min_created = std::min(sh.min_created, c.created_ticks);
            sh.max_created = std::max(sh.max_created, c.created_ticks);
            pool += lower(c.content);                 // Content.ToLowerInvariant(), :110 hoisted to ingest
            off.push_back(pool.size());
            ++e;
        }
        r = orr_index_append(sh.index, (int64_t)(e - i), has ? dim : 0, has ? emb.data() : nullptr, created.data(),
                             reinterpret_cast<const uint8_t *>(pool.data()), off.data(), ids.data());
        if (r != ORR_OK) { orr_index_destroy(sh.index); return fail(r, orr_last_error()); }
        i = e;
    }
    r = orr_index_seal(sh.index);
    if (r != ORR_OK) { orr_index_destroy(sh.index); return fail(r, orr_last_error()); }
    for (const auto &c : sh.chunks) sh.doc_stamps[c.document_id] = 0;
    *out = std::move(sh);
    return ORR_OK;
}

int32_t majority_dim(const std::vector<Chunk> &chunks)
{
    std::map<int32_t, int64_t> hist;
    for (const auto &c : chunks) if (!c.embedding.empty()) hist[(int32_t)c.embedding.size()]++;
    int32_t dim = 0;
    int64_t best = 0;
    for (auto &kv : hist) if (kv.second > best) { best = kv.second; dim = kv.first; }
    return dim;
}

void assign_row_bases(orrh_service *svc)
{
    int64_t base = 0, dead = 0;
    for (auto &sh : svc->shards) {                    // newest shard first
        orr_index_set_row_base(sh.index, base);
        orr_index_set_option(sh.index, "dead_rows_before", dead);      // candidate_limit counts live rows
        base += orr_index_rows(sh.index);                              // (rows on the device: compaction removed some of the mirror's)
        dead += orr_index_rows(sh.index) - orr_index_live_rows(sh.index);
    }
}

// Bring the device shards up to date with the store (the :26 data source).  Chunk lists are
// flattened in enumeration order, as GetRecentChunksAsync does (InMemoryIngestionStore.cs:59-60).
int ensure_index(orrh_service *svc)
{
    std::lock_guard<std::mutex> sl(svc->store->mu);
    orrh_store *st = svc->store;
    if (!svc->shards.empty() && svc->built_version == st->chunks_version) return ORR_OK;

    // what is indexed vs what the store holds now.  Documents that were deleted, or whose chunk list was
    // replaced (InMemoryIngestionStore.cs:17-25, 50-55), lose their rows in place: orr_index_delete_rows,
    // no reseal; a replaced list then counts as new below.
    std::map<std::string, uint64_t> indexed;
    bool changed = false;
    for (auto &sh : svc->shards) {
        std::vector<std::string> stale;
        for (const auto &kv : sh.doc_stamps) {
            auto it = st->chunk_stamp.find(kv.first);
            if (it == st->chunk_stamp.end() || it->second != kv.second) stale.push_back(kv.first);
        }
        if (!stale.empty() && !changed) {
            if (sh.dead.empty()) sh.dead.assign(sh.chunks.size(), 0);
            std::vector<int64_t> ids;
            std::vector<size_t> marked;
            for (size_t p = 0; p < sh.chunks.size(); ++p)
                if (!sh.dead[p] && std::binary_search(stale.begin(), stale.end(), sh.chunks[p].document_id)) {
                    ids.push_back(sh.id_base + (int64_t)p);
                    marked.push_back(p);
                }
            int64_t done = 0;
            int r = orr_index_delete_rows(sh.index, (int64_t)ids.size(), ids.data(), &done);
            if (r == ORR_ESTATE) {
                // more than a quarter of the shard would be tombstones: compact it in place (the rows deleted so far leave the
                // device arrays; ids are kept, so the mirror's id -> chunk map stands) and delete again
                int64_t removed = 0;
                if (orr_index_compact(sh.index, &removed) == ORR_OK) {
                    svc->compactions++;
                    sh.n_dead_compacted += removed;
                    r = orr_index_delete_rows(sh.index, (int64_t)ids.size(), ids.data(), &done);
                }
            }
            if (r == ORR_ESTATE) changed = true;              // still too much of the shard is gone: rebuild below
            else if (r != ORR_OK) return fail(r, orr_last_error());     // (nothing was marked: the mirror still matches the index)
            else {
                for (size_t p : marked) sh.dead[p] = 1;       // only once the index has dropped them
                sh.n_dead += done;
                svc->tombstoned_rows += done;
                for (const auto &d : stale) sh.doc_stamps.erase(d);
            }
        }
        for (const auto &kv : sh.doc_stamps) indexed[kv.first] = kv.second;
    }
    std::vector<Chunk> added;
    std::map<std::string, uint64_t> added_stamps;
    if (!changed)
        for (const auto &doc : st->doc_order) {
            if (indexed.count(doc)) continue;
            auto it = st->chunks_by_document.find(doc);
            if (it == st->chunks_by_document.end()) continue;
            for (const auto &c : it->second) added.push_back(c);
            added_stamps[doc] = st->chunk_stamp[doc];
        }
    bool delta_ok = !changed && !svc->shards.empty() && !added.empty();
    if (delta_ok) {
        int64_t newest = svc->shards[0].max_created;
        for (const auto &sh : svc->shards) newest = std::max(newest, sh.max_created);
        const int32_t d = majority_dim(added);
        for (const auto &c : added) delta_ok = delta_ok && c.created_ticks > newest;      // strictly newer: ties keep enumeration order
        delta_ok = delta_ok && (d == svc->dim || d == 0);
    }
    if (delta_ok && svc->shards.size() >= 8) {
        // Eight shards already: the DELTA shards (all but the oldest) and the new chunks become ONE shard; the oldest -- the
        // large one of a corpus that grows by uploads -- stays on the device untouched.  The delta shards are strictly newer
        // than one another in the order they were added, so their live chunks concatenated newest shard first (each in its own
        // enumeration order) are what a rebuild would enumerate for those documents.  Only worth it while the deltas together
        // are smaller than that oldest shard; otherwise everything is rebuilt below.
        size_t delta_rows = added.size();
        for (size_t i = 0; i + 1 < svc->shards.size(); ++i) delta_rows += svc->shards[i].chunks.size() - (size_t)svc->shards[i].n_dead;
        const Shard &base = svc->shards.back();
        if (delta_rows < base.chunks.size() - (size_t)base.n_dead) {
            std::vector<Chunk> merged = std::move(added);
            std::map<std::string, uint64_t> stamps = added_stamps;
            for (size_t i = 0; i + 1 < svc->shards.size(); ++i) {
                Shard &sh = svc->shards[i];
                for (size_t p = 0; p < sh.chunks.size(); ++p)
                    if (sh.dead.empty() || !sh.dead[p]) merged.push_back(std::move(sh.chunks[p]));
                for (const auto &kv : sh.doc_stamps) stamps[kv.first] = kv.second;
            }
            Shard one;
            int r = build_shard(svc, std::move(merged), svc->dim, &one);
            if (r != ORR_OK) {                              // (the mirror's chunks were moved out: nothing to fall back on but a rebuild)
                free_shards(svc);
                svc->built_version = ~0ull;
                return r;
            }
            one.doc_stamps = stamps;
            Shard keep = std::move(svc->shards.back());
            svc->shards.pop_back();
            free_shards(svc);
            svc->shards.push_back(std::move(one));
            svc->shards.push_back(std::move(keep));
            svc->delta_merges++;
            assign_row_bases(svc);
            svc->built_version = st->chunks_version;
            return ORR_OK;
        }
        delta_ok = false;
    }
    if (delta_ok) {
        Shard sh;
        int r = build_shard(svc, std::move(added), svc->dim, &sh);
        if (r != ORR_OK) return r;
        sh.doc_stamps = added_stamps;
        svc->shards.insert(svc->shards.begin(), std::move(sh));
        svc->delta_builds++;
    } else if (changed || svc->shards.empty() || !added.empty()) {
        free_shards(svc);
        std::vector<Chunk> all;
        std::map<std::string, uint64_t> stamps;
        for (const auto &doc : st->doc_order) {
            auto it = st->chunks_by_document.find(doc);
            if (it == st->chunks_by_document.end()) continue;
            for (const auto &c : it->second) all.push_back(c);
            stamps[doc] = st->chunk_stamp[doc];
        }
        svc->dim = majority_dim(all);
        Shard sh;
        int r = build_shard(svc, std::move(all), svc->dim, &sh);
        if (r != ORR_OK) return r;
        sh.doc_stamps = stamps;
        svc->shards.push_back(std::move(sh));
        svc->full_rebuilds++;
    }
    assign_row_bases(svc);
    svc->built_version = st->chunks_version;
    return ORR_OK;
}

const Chunk *chunk_of(const orrh_service *svc, int64_t row_id)
{
    for (const auto &sh : svc->shards)
        if (row_id >= sh.id_base && row_id < sh.id_base + (int64_t)sh.chunks.size()) return &sh.chunks[(size_t)(row_id - sh.id_base)];
    return nullptr;
}

// RecallSearchService.cs:26-37 over one or several shards.
int search_shards(orrh_service *svc, int32_t qdim, const float *qvec, const uint8_t *terms, const uint32_t *term_off,
                  const uint32_t *qoff, int64_t now_ticks, int32_t topk, int64_t *rows, double *scores, int32_t *count)
{
    if (svc->shards.size() == 1) {
        int r = orr_search_batch(svc->shards[0].index, 1, qdim, qdim > 0 ? qvec : nullptr, terms, term_off, qoff, now_ticks, topk,
                                 svc->candidate_limit, rows, scores, count);
        return r == ORR_OK ? ORR_OK : fail(r, orr_last_error());
    }
    const int32_t n_sh = (int32_t)svc->shards.size();
    int64_t total = 0;
    for (const auto &sh : svc->shards) total += (int64_t)sh.chunks.size();
    int64_t kprime = std::min<int64_t>(std::max<int64_t>(total, 1), std::max<int64_t>(std::max(1, topk) + 22, 32));
    for (;;) {
        std::vector<orr_candidate> recs((size_t)n_sh * ((size_t)kprime + 1));
        for (int32_t i = 0; i < n_sh; ++i) {
            int r = orr_search_shard(svc->shards[i].index, 1, qdim, qdim > 0 ? qvec : nullptr, terms, term_off, qoff, now_ticks,
                                     (int32_t)kprime, svc->candidate_limit, recs.data() + (size_t)i * (kprime + 1));
            if (r != ORR_OK) return fail(r, orr_last_error());
        }
        int32_t unc = 0;
        int r = orr_merge_candidates(n_sh, 1, (int32_t)kprime, recs.data(), svc->dim, qdim, qvec, qoff, now_ticks, topk, rows, scores,
                                     count, &unc);
        if (r != ORR_OK) return fail(r, orr_last_error());
        if (unc == 0 || kprime >= total) return ORR_OK;
        kprime = std::min<int64_t>(total, kprime * 4);
    }
}

}  // namespace

extern "C" {

const char *orrh_last_error(void) { return g_err.c_str(); }

orrh_store *orrh_store_create(void) { return new orrh_store(); }
void orrh_store_destroy(orrh_store *s) { delete s; }

int orrh_store_upsert_document(orrh_store *s, const char *id, const char *file_name, int64_t created_ticks)
{
    if (!s || !id || !file_name) return fail(ORR_EINVAL, "orrh_store_upsert_document: null argument");
    std::lock_guard<std::mutex> l(s->mu);
    Document d;
    d.id = id; d.file_name = file_name; d.created_ticks = created_ticks;
    s->documents[d.id] = d;
    s->version++;
    return ORR_OK;
}

int orrh_store_upsert_chunks(orrh_store *s, const char *document_id, int32_t n, const char *const *chunk_ids,
                             const int32_t *chunk_index, const char *const *contents, const float *emb,
                             const int32_t *emb_len, const int64_t *created_ticks)
{
    if (!s || !document_id) return fail(ORR_EINVAL, "orrh_store_upsert_chunks: null argument");
    if (n <= 0) return ORR_OK;                                        // :19-20
    if (!chunk_ids || !chunk_index || !contents || !created_ticks) return fail(ORR_EINVAL, "orrh_store_upsert_chunks: null array");
    std::vector<Chunk> list((size_t)n);
    size_t eoff = 0;
    for (int32_t i = 0; i < n; ++i) {
        Chunk &c = list[i];
        c.id = chunk_ids[i]; c.document_id = document_id; c.content = contents[i] ? contents[i] : "";
        c.chunk_index = chunk_index[i]; c.created_ticks = created_ticks[i];
        const int32_t len = emb_len ? emb_len[i] : 0;
        if (len > 0 && emb) c.embedding.assign(emb + eoff, emb + eoff + len);
        eoff += (size_t)std::max(len, 0);
    }
    std::stable_sort(list.begin(), list.end(), [](const Chunk &a, const Chunk &b) { return a.chunk_index < b.chunk_index; });  // :23
    std::lock_guard<std::mutex> l(s->mu);
    if (!s->chunks_by_document.count(document_id)) s->doc_order.push_back(document_id);
    s->chunks_by_document[document_id] = std::move(list);
    s->version++;
    s->chunks_version++;
    s->chunk_stamp[document_id] = s->chunks_version;
    return ORR_OK;
}

int orrh_store_delete_document(orrh_store *s, const char *document_id)
{
    if (!s || !document_id) return fail(ORR_EINVAL, "orrh_store_delete_document: null argument");
    std::lock_guard<std::mutex> l(s->mu);
    s->documents.erase(document_id);
    if (s->chunks_by_document.erase(document_id)) {
        s->doc_order.erase(std::remove(s->doc_order.begin(), s->doc_order.end(), std::string(document_id)), s->doc_order.end());
        s->chunk_stamp.erase(document_id);
        s->chunks_version++;
    }
    s->version++;
    return ORR_OK;
}

int64_t orrh_store_chunk_count(const orrh_store *s)
{
    if (!s) return 0;
    int64_t n = 0;
    for (const auto &kv : s->chunks_by_document) n += (int64_t)kv.second.size();
    return n;
}

orrh_service *orrh_service_create(orrh_store *s, int32_t device, int64_t candidate_limit)
{
    if (!s) { g_err = "orrh_service_create: null store"; return nullptr; }
    orrh_service *svc = new orrh_service();
    svc->store = s; svc->device = device; svc->candidate_limit = candidate_limit;
    return svc;
}

void orrh_service_destroy(orrh_service *svc)
{
    if (!svc) return;
    free_shards(svc);
    delete svc;
}

void orrh_service_stats(orrh_service *svc, int32_t *n_shards, int64_t *full_rebuilds, int64_t *delta_builds)
{
    if (!svc) return;
    std::lock_guard<std::mutex> l(svc->mu);
    if (n_shards) *n_shards = (int32_t)svc->shards.size();
    if (full_rebuilds) *full_rebuilds = svc->full_rebuilds;
    if (delta_builds) *delta_builds = svc->delta_builds;
}

int64_t orrh_service_delta_merges(orrh_service *svc)
{
    if (!svc) return 0;
    std::lock_guard<std::mutex> lock(svc->mu);
    return svc->delta_merges;
}

int64_t orrh_service_compactions(orrh_service *svc)
{
    if (!svc) return 0;
    std::lock_guard<std::mutex> l(svc->mu);
    return svc->compactions;
}

int64_t orrh_service_tombstoned_rows(orrh_service *svc)
{
    if (!svc) return 0;
    std::lock_guard<std::mutex> l(svc->mu);
    return svc->tombstoned_rows;
}

void orrh_free(void *p) { free(p); }

int orrh_service_search_json(orrh_service *svc, const char *query_utf8, const float *qvec, int32_t qdim, int32_t topk,
                             int64_t now_ticks, char **out_json, int64_t *out_len)
{
    if (!svc || !out_json) return fail(ORR_EINVAL, "orrh_service_search_json: null argument");
    *out_json = nullptr;
    const std::string query = query_utf8 ? query_utf8 : "";
    if (orrh_is_blank(reinterpret_cast<const uint8_t *>(query.data()), (int64_t)query.size()))
        return fail(ORR_EINVAL, "Query is required.");                                       // :22-23
    std::lock_guard<std::mutex> l(svc->mu);
    int r = ensure_index(svc);                                                                // the :26 data source
    if (r != ORR_OK) return r;

    // queryTerms (:95-108), once per query
    std::vector<uint8_t> terms(4 * query.size() + 16);
    std::vector<uint32_t> term_off(query.size() + 2);
    const int32_t T = orrh_query_terms(reinterpret_cast<const uint8_t *>(query.data()), (int64_t)query.size(), terms.data(),
                                       (int64_t)terms.size(), term_off.data(), (int32_t)term_off.size());
    if (T < 0) return fail(ORR_EINVAL, "query tokenisation failed");
    const uint32_t qoff[2] = {0, (uint32_t)T};

    const int32_t k = std::max(1, topk);                                                      // :36
    std::vector<int64_t> rows((size_t)k, -1);
    std::vector<double> scores((size_t)k, 0.0);
    int32_t count = 0;
    r = search_shards(svc, qdim, qvec, terms.data(), term_off.data(), qoff, now_ticks, topk, rows.data(), scores.data(),
                      &count);                                                                // replaces :26-37
    if (r != ORR_OK) return r;

    std::string js = "{\"query\":";
    json_string(query, js);
    js += ",\"citations\":[";
    {
        std::lock_guard<std::mutex> sl(svc->store->mu);
        for (int32_t i = 0; i < count; ++i) {
            const Chunk *cp = chunk_of(svc, rows[i]);
            if (!cp) return fail(ORR_ECOMM, "search returned an unknown row id");
            const Chunk &c = *cp;
            auto d = svc->store->documents.find(c.document_id);                               // :39,44
            const std::string file = d == svc->store->documents.end() ? "unknown" : d->second.file_name;   // :47
            std::string snip(4 * c.content.size() + 16, '\0');
            int64_t m = orrh_build_snippet(reinterpret_cast<const uint8_t *>(c.content.data()), (int64_t)c.content.size(), 180,
                                           reinterpret_cast<uint8_t *>(&snip[0]), (int64_t)snip.size());   // :50
            snip.resize(m < 0 ? 0 : (size_t)m);
            if (i) js.push_back(',');
            js += "{\"documentId\":"; json_string(c.document_id, js);
            js += ",\"fileName\":"; json_string(file, js);
            js += ",\"chunkId\":"; json_string(c.id, js);
            js += ",\"chunkIndex\":" + std::to_string(c.chunk_index);
            js += ",\"snippet\":"; json_string(snip, js);
            js += ",\"score\":"; json_double(orrh_round4(scores[i]), js);                     // :51
            js += ",\"createdAtUtc\":"; json_string(iso_utc(c.created_ticks), js);
            js.push_back('}');
        }
    }
    js += "]}";
    char *buf = static_cast<char *>(malloc(js.size() + 1));
    if (!buf) return fail(ORR_ENOMEM, "out of memory");
    memcpy(buf, js.data(), js.size() + 1);
    *out_json = buf;
    if (out_len) *out_len = (int64_t)js.size();
    return ORR_OK;
}

}  // extern "C"
