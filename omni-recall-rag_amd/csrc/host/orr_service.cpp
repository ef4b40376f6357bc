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

namespace {

thread_local std::string g_err;
int fail(int code, const std::string &msg) { g_err = msg; return code; }

struct Chunk {                      // CosmosChunkRecord (CosmosIngestionRecords.cs:19-30)
    std::string id, document_id, content;
    int32_t chunk_index = 0;
    std::vector<float> embedding;   // empty = null
    int64_t created_ticks = 0;
};
struct Document {                   // CosmosDocumentRecord, the fields the path reads
    std::string id, file_name;
    int64_t created_ticks = 0;
};

}  // namespace

struct orrh_store {
    std::mutex mu;
    std::vector<std::string> doc_order;                 // enumeration order of _chunksByDocument
    std::map<std::string, Document> documents;
    std::map<std::string, std::vector<Chunk>> chunks_by_document;
    uint64_t version = 0;
};

struct orrh_service {
    orrh_store *store = nullptr;
    int32_t device = 0;
    int64_t candidate_limit = 300;
    std::mutex mu;
    orr_index *index = nullptr;
    uint64_t built_version = ~0ull;
    std::vector<const Chunk *> rows;                    // row id -> chunk
    std::vector<Chunk> snapshot;
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

void json_double(double v, std::string &out)
{
    if (std::isnan(v) || std::isinf(v)) { out += "null"; return; }     // System.Text.Json would throw
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf), v);                 // shortest round-trip, like .NET "R"
    out.append(buf, r.ptr);
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

// (Re)build the device index from the store: chunk lists in enumeration order, as
// GetRecentChunksAsync flattens them (InMemoryIngestionStore.cs:59-60).
int ensure_index(orrh_service *svc)
{
    std::lock_guard<std::mutex> sl(svc->store->mu);
    if (svc->index && svc->built_version == svc->store->version) return ORR_OK;
    if (svc->index) { orr_index_destroy(svc->index); svc->index = nullptr; }
    svc->snapshot.clear();
    for (const auto &doc : svc->store->doc_order) {
        auto it = svc->store->chunks_by_document.find(doc);
        if (it == svc->store->chunks_by_document.end()) continue;
        for (const auto &c : it->second) svc->snapshot.push_back(c);
    }
    int32_t dim = 0;
    {   // index dimension = the most common non-empty embedding length
        std::map<int32_t, int64_t> hist;
        for (const auto &c : svc->snapshot) if (!c.embedding.empty()) hist[(int32_t)c.embedding.size()]++;
        int64_t best = 0;
        for (auto &kv : hist) if (kv.second > best) { best = kv.second; dim = kv.first; }
    }
    orr_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = (int32_t)sizeof(cfg);
    cfg.device = svc->device;
    cfg.dim = dim;
    int r = orr_index_create(&cfg, &svc->index);
    if (r != ORR_OK) return fail(r, orr_last_error());
    svc->rows.clear();
    const size_t n = svc->snapshot.size();
    size_t i = 0;
    while (i < n) {                                   // runs of rows with / without a usable embedding
        const bool has = dim > 0 && (int32_t)svc->snapshot[i].embedding.size() == dim;
        size_t e = i;
        std::vector<float> emb;
        std::vector<int64_t> created;
        std::vector<uint64_t> off{0};
        std::string pool;
        while (e < n && (dim > 0 && (int32_t)svc->snapshot[e].embedding.size() == dim) == has && e - i < 65536) {
            const Chunk &c = svc->snapshot[e];
            if (has) emb.insert(emb.end(), c.embedding.begin(), c.embedding.end());
            created.push_back(c.created_ticks);
            pool += lower(c.content);                 // Content.ToLowerInvariant(), :110 hoisted to ingest
            off.push_back(pool.size());
            ++e;
        }
        r = orr_index_append(svc->index, (int64_t)(e - i), has ? dim : 0, has ? emb.data() : nullptr, created.data(),
                             reinterpret_cast<const uint8_t *>(pool.data()), off.data(), nullptr);
        if (r != ORR_OK) return fail(r, orr_last_error());
        i = e;
    }
    for (const auto &c : svc->snapshot) svc->rows.push_back(&c);
    r = orr_index_seal(svc->index);
    if (r != ORR_OK) return fail(r, orr_last_error());
    svc->built_version = svc->store->version;
    return ORR_OK;
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
    return ORR_OK;
}

int orrh_store_delete_document(orrh_store *s, const char *document_id)
{
    if (!s || !document_id) return fail(ORR_EINVAL, "orrh_store_delete_document: null argument");
    std::lock_guard<std::mutex> l(s->mu);
    s->documents.erase(document_id);
    if (s->chunks_by_document.erase(document_id))
        s->doc_order.erase(std::remove(s->doc_order.begin(), s->doc_order.end(), std::string(document_id)), s->doc_order.end());
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
    if (svc->index) orr_index_destroy(svc->index);
    delete svc;
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
    r = orr_search_batch(svc->index, 1, qdim, qdim > 0 ? qvec : nullptr, terms.data(), term_off.data(), qoff, now_ticks,
                         topk, svc->candidate_limit, rows.data(), scores.data(), &count);     // replaces :26-37
    if (r != ORR_OK) return fail(r, orr_last_error());

    std::string js = "{\"query\":";
    json_string(query, js);
    js += ",\"citations\":[";
    {
        std::lock_guard<std::mutex> sl(svc->store->mu);
        for (int32_t i = 0; i < count; ++i) {
            const Chunk &c = *svc->rows[(size_t)rows[i]];
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
