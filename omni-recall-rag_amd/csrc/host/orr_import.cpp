// orr_import.cpp -- the reference's only durable corpus format, in and out of the store mirror
// (include/omnirecall_host.h).
//
// Cosmos items are CosmosDocumentRecord / CosmosChunkRecord (Data/Models/CosmosIngestionRecords.cs:5-30)
// serialised by System.Text.Json with JsonNamingPolicy.CamelCase (Services/CosmosIngestionStore.cs:34-40):
//   chunk     {"id","PartitionKey","type":"chunk","documentId","chunkIndex","content","embedding":[..]|null,"createdAtUtc"}
//   document  {"id","PartitionKey","type":"document","fileName","sourceType","blobPath","contentHash","chunkCount","createdAtUtc"}
// plus whatever system properties the service adds on the way out (_rid, _self, _etag, _attachments, _ts),
// which are skipped like any unknown property.  Property names are matched case-sensitively and the last
// duplicate wins, as System.Text.Json does.  Numbers of "embedding" are parsed straight to binary32 with
// correct rounding (what Utf8JsonReader.GetSingle does), never through a double.
#include <charconv>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../../include/omnirecall_hip.h"
#include "../../../include/omnirecall_host.h"

#include "orr_store.h"

using orrh_detail::fail;

namespace {

struct Record {
    std::string id, type, document_id, content, file_name, created;
    bool has_type = false, has_document_id = false, has_file_name = false, has_created = false;
    int64_t chunk_index = 0;
    std::vector<float> embedding;
};

struct Parser {
    const uint8_t *p, *end;
    std::string err;
    int depth = 0;

    bool failp(const std::string &m)
    {
        if (err.empty()) err = m + " at byte " + std::to_string((long long)(p - start));
        return false;
    }
    const uint8_t *start;
    Parser(const uint8_t *b, const uint8_t *e) : p(b), end(e), start(b) {}

    void ws()
    {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
    }
    bool eat(char c)
    {
        ws();
        if (p < end && *p == (uint8_t)c) { ++p; return true; }
        return false;
    }
    bool peek(char c)
    {
        ws();
        return p < end && *p == (uint8_t)c;
    }
    bool literal(const char *lit)
    {
        const size_t n = strlen(lit);
        if ((size_t)(end - p) >= n && memcmp(p, lit, n) == 0) { p += n; return true; }
        return false;
    }
    static void put_utf8(uint32_t cp, std::string &out)
    {
        if (cp < 0x80) out.push_back((char)cp);
        else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
        else if (cp < 0x10000) {
            out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F)));
        } else {
            out.push_back((char)(0xF0 | (cp >> 18))); out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
            out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F)));
        }
    }
    bool hex4(uint32_t *v)
    {
        if (end - p < 4) return failp("truncated \\u escape");
        uint32_t x = 0;
        for (int i = 0; i < 4; ++i) {
            const uint8_t c = p[i];
            x <<= 4;
            if (c >= '0' && c <= '9') x |= c - '0';
            else if (c >= 'a' && c <= 'f') x |= c - 'a' + 10;
            else if (c >= 'A' && c <= 'F') x |= c - 'A' + 10;
            else return failp("bad \\u escape");
        }
        p += 4;
        *v = x;
        return true;
    }
    bool string(std::string *out)       // out == nullptr: skip
    {
        ws();
        if (p >= end || *p != '"') return failp("expected a string");
        ++p;
        for (;;) {
            if (p >= end) return failp("unterminated string");
            const uint8_t c = *p++;
            if (c == '"') return true;
            if (c < 0x20) return failp("control character inside a string");
            if (c != '\\') { if (out) out->push_back((char)c); continue; }
            if (p >= end) return failp("unterminated escape");
            const uint8_t e = *p++;
            char plain = 0;
            switch (e) {
            case '"': plain = '"'; break;
            case '\\': plain = '\\'; break;
            case '/': plain = '/'; break;
            case 'b': plain = '\b'; break;
            case 'f': plain = '\f'; break;
            case 'n': plain = '\n'; break;
            case 'r': plain = '\r'; break;
            case 't': plain = '\t'; break;
            case 'u': {
                uint32_t cp = 0;
                if (!hex4(&cp)) return false;
                if (cp >= 0xD800 && cp < 0xDC00) {                       // high surrogate: needs its partner
                    uint32_t lo = 0;
                    if (end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
                        const uint8_t *save = p;
                        p += 2;
                        if (!hex4(&lo)) return false;
                        if (lo >= 0xDC00 && lo < 0xE000) cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        else { p = save; cp = 0xFFFD; }
                    } else cp = 0xFFFD;
                } else if (cp >= 0xDC00 && cp < 0xE000) cp = 0xFFFD;
                if (out) put_utf8(cp, *out);
                continue;
            }
            default: return failp("unknown escape");
            }
            if (out) out->push_back(plain);
        }
    }
    // one JSON number token [b, e)
    bool number(const char **b, const char **e)
    {
        ws();
        const uint8_t *s = p;
        if (p < end && *p == '-') ++p;
        if (p >= end || *p < '0' || *p > '9') return failp("expected a number");
        if (*p == '0') ++p; else while (p < end && *p >= '0' && *p <= '9') ++p;
        if (p < end && *p == '.') {
            ++p;
            if (p >= end || *p < '0' || *p > '9') return failp("digits must follow the decimal point");
            while (p < end && *p >= '0' && *p <= '9') ++p;
        }
        if (p < end && (*p == 'e' || *p == 'E')) {
            ++p;
            if (p < end && (*p == '+' || *p == '-')) ++p;
            if (p >= end || *p < '0' || *p > '9') return failp("digits must follow the exponent");
            while (p < end && *p >= '0' && *p <= '9') ++p;
        }
        *b = reinterpret_cast<const char *>(s);
        *e = reinterpret_cast<const char *>(p);
        return true;
    }
    bool skip_value()
    {
        ws();
        if (p >= end) return failp("unexpected end of input");
        if (++depth > 64) return failp("nesting deeper than 64 levels");     // System.Text.Json's default MaxDepth
        bool ok = true;
        if (*p == '"') ok = string(nullptr);
        else if (*p == '{') {
            ++p;
            if (!eat('}')) {
                do {
                    if (!string(nullptr)) { ok = false; break; }
                    if (!eat(':')) { ok = failp("expected ':'"); break; }
                    if (!skip_value()) { ok = false; break; }
                } while (eat(','));
                if (ok && !eat('}')) ok = failp("expected '}'");
            }
        } else if (*p == '[') {
            ++p;
            if (!eat(']')) {
                do { if (!skip_value()) { ok = false; break; } } while (eat(','));
                if (ok && !eat(']')) ok = failp("expected ']'");
            }
        } else if (literal("true") || literal("false") || literal("null")) {
        } else {
            const char *b, *e;
            ok = number(&b, &e);
        }
        --depth;
        return ok;
    }
    bool string_or_null(std::string *out, bool *present)
    {
        ws();
        if (literal("null")) { out->clear(); *present = false; return true; }
        out->clear();
        *present = true;
        return string(out);
    }
    bool float_array_or_null(std::vector<float> *out)
    {
        out->clear();
        ws();
        if (literal("null")) return true;
        if (!eat('[')) return failp("\"embedding\" must be an array of numbers or null");
        if (eat(']')) return true;
        do {
            ws();
            float v = 0.f;
            if (p < end && *p == '"') {        // JsonNumberHandling.AllowNamedFloatingPointLiterals
                std::string name;
                if (!string(&name)) return false;
                if (name == "NaN") v = std::numeric_limits<float>::quiet_NaN();
                else if (name == "Infinity") v = std::numeric_limits<float>::infinity();
                else if (name == "-Infinity") v = -std::numeric_limits<float>::infinity();
                else return failp("\"embedding\" holds a string that is not a floating-point literal");
            } else {
                const char *b, *e;
                if (!number(&b, &e)) return false;
                auto r = std::from_chars(b, e, v);                       // correctly rounded binary32
                if (r.ec == std::errc::result_out_of_range) v = strtof(std::string(b, e).c_str(), nullptr);   // +-inf / 0 / subnormal
                else if (r.ec != std::errc() || r.ptr != e) return failp("bad number in \"embedding\"");
            }
            out->push_back(v);
        } while (eat(','));
        if (!eat(']')) return failp("expected ']' after the embedding");
        return true;
    }
    bool integer(int64_t *out)
    {
        const char *b, *e;
        if (!number(&b, &e)) return false;
        auto r = std::from_chars(b, e, *out);
        if (r.ec != std::errc() || r.ptr != e) return failp("\"chunkIndex\" must be an integer");
        return true;
    }

    bool records_array(std::vector<Record> *out);
    // One top-level object: a record, or a query page {"Documents":[...], "_count":..}.
    bool object(std::vector<Record> *out)
    {
        if (!eat('{')) return failp("expected '{'");
        Record r;
        bool page = false;
        if (!eat('}')) {
            do {
                std::string key;
                if (!string(&key)) return false;
                if (!eat(':')) return failp("expected ':'");
                bool present = false;
                if (key == "id") { if (!string_or_null(&r.id, &present)) return false; }
                else if (key == "type") { if (!string_or_null(&r.type, &r.has_type)) return false; }
                else if (key == "documentId") { if (!string_or_null(&r.document_id, &r.has_document_id)) return false; }
                else if (key == "content") { if (!string_or_null(&r.content, &present)) return false; }
                else if (key == "fileName") { if (!string_or_null(&r.file_name, &r.has_file_name)) return false; }
                else if (key == "createdAtUtc") { if (!string_or_null(&r.created, &r.has_created)) return false; }
                else if (key == "chunkIndex") { if (!integer(&r.chunk_index)) return false; }
                else if (key == "embedding") { if (!float_array_or_null(&r.embedding)) return false; }
                else if (key == "Documents" && peek('[')) { page = true; if (!records_array(out)) return false; }
                else if (!skip_value()) return false;
            } while (eat(','));
            if (!eat('}')) return failp("expected '}'");
        }
        if (!page) out->push_back(std::move(r));
        return true;
    }
};

bool Parser::records_array(std::vector<Record> *out)
{
    if (!eat('[')) return failp("expected '['");
    if (eat(']')) return true;
    do { if (!object(out)) return false; } while (eat(','));
    if (!eat(']')) return failp("expected ']'");
    return true;
}

// ISO 8601 extended profile that System.Text.Json reads into a DateTime -> UTC ticks (100 ns since 0001-01-01).
bool parse_iso_ticks(const std::string &s, int64_t *ticks)
{
    auto num = [&](size_t at, int n, int64_t *v) {
        if (at + (size_t)n > s.size()) return false;
        int64_t x = 0;
        for (int i = 0; i < n; ++i) {
            const char c = s[at + (size_t)i];
            if (c < '0' || c > '9') return false;
            x = x * 10 + (c - '0');
        }
        *v = x;
        return true;
    };
    int64_t y, mo, d, h = 0, mi = 0, sec = 0, frac = 0, off_min = 0;
    if (!num(0, 4, &y) || s.size() < 10 || s[4] != '-' || !num(5, 2, &mo) || s[7] != '-' || !num(8, 2, &d)) return false;
    size_t at = 10;
    if (at < s.size()) {
        if (s[at] != 'T') return false;
        if (!num(at + 1, 2, &h) || at + 3 >= s.size() || s[at + 3] != ':' || !num(at + 4, 2, &mi)) return false;
        at += 6;
        if (at < s.size() && s[at] == ':') {
            if (!num(at + 1, 2, &sec)) return false;
            at += 3;
            if (at < s.size() && s[at] == '.') {
                ++at;
                int digits = 0;
                while (at < s.size() && s[at] >= '0' && s[at] <= '9') {
                    if (digits < 7) frac = frac * 10 + (s[at] - '0');        // beyond 100 ns: truncated
                    ++digits; ++at;
                }
                if (digits == 0 || digits > 16) return false;
                for (; digits < 7; ++digits) frac *= 10;
            }
        }
        if (at < s.size()) {
            if (s[at] == 'Z') ++at;
            else if (s[at] == '+' || s[at] == '-') {
                int64_t oh, om;
                if (!num(at + 1, 2, &oh) || at + 3 >= s.size() || s[at + 3] != ':' || !num(at + 4, 2, &om) || oh > 14 || om > 59) return false;
                off_min = (oh * 60 + om) * (s[at] == '-' ? -1 : 1);
                at += 6;
            }
            if (at != s.size()) return false;
        }
    }
    if (y < 1 || mo < 1 || mo > 12 || d < 1 || h > 23 || mi > 59 || sec > 59) return false;
    static const int mdays[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
    const bool leap = (y % 4 == 0 && y % 100 != 0) || y % 400 == 0;
    if (d > mdays[mo - 1] + ((mo == 2 && leap) ? 1 : 0)) return false;
    // days since 0001-01-01 (proleptic Gregorian)
    const int64_t yy = mo <= 2 ? y - 1 : y;
    const int64_t era = yy / 400, yoe = yy - era * 400;
    const int64_t doy = (153 * (mo + (mo > 2 ? -3 : 9)) + 2) / 5 + d - 1;
    const int64_t doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
    const int64_t days = era * 146097 + doe - 306;            // 0000-03-01 based -> 0001-01-01 based
    const int64_t tps = 10000000;
    int64_t t = ((days * 24 + h) * 60 + mi) * 60 + sec;
    t = t * tps + frac - off_min * 60 * tps;
    if (t < 0) return false;
    *ticks = t;
    return true;
}

void json_float(float v, std::string &out)
{
    if (std::isnan(v)) { out += "\"NaN\""; return; }
    if (std::isinf(v)) { out += v > 0 ? "\"Infinity\"" : "\"-Infinity\""; return; }
    char buf[48];
    auto r = std::to_chars(buf, buf + sizeof(buf), v);        // shortest text that reads back as the same binary32
    out.append(buf, r.ptr);
}

}  // namespace

extern "C" {

int orrh_store_import_cosmos_json(orrh_store *s, const uint8_t *json, int64_t len, int64_t *out_documents,
                                  int64_t *out_chunks)
{
    if (out_documents) *out_documents = 0;
    if (out_chunks) *out_chunks = 0;
    if (!s || (len > 0 && !json) || len < 0) return fail(ORR_EINVAL, "orrh_store_import_cosmos_json: bad argument");
    Parser ps(json, json + len);
    if (len >= 3 && json[0] == 0xEF && json[1] == 0xBB && json[2] == 0xBF) ps.p += 3;      // UTF-8 BOM
    std::vector<Record> recs;
    ps.ws();
    if (ps.peek('[')) {
        if (!ps.records_array(&recs)) return fail(ORR_EINVAL, "Cosmos JSON: " + ps.err);
        ps.ws();
        if (ps.p != ps.end) return fail(ORR_EINVAL, "Cosmos JSON: text after the closing ']'");
    } else {
        for (;;) {                     // objects one after another: JSON lines, or query pages back to back
            ps.ws();
            if (ps.p >= ps.end) break;
            if (!ps.object(&recs)) return fail(ORR_EINVAL, "Cosmos JSON: " + ps.err);
            ps.eat(',');
        }
    }
    // everything is parsed and checked before the store changes
    struct DocIn { std::string id, file; int64_t ticks; };
    std::vector<DocIn> docs;
    std::vector<std::string> chunk_docs;                               // first-seen order
    std::map<std::string, std::vector<const Record *>> by_doc;
    std::vector<int64_t> ticks(recs.size(), 0);
    for (size_t i = 0; i < recs.size(); ++i) {
        const Record &r = recs[i];
        const bool is_chunk = r.has_type ? r.type == "chunk" : r.has_document_id;
        const bool is_doc = r.has_type ? r.type == "document" : (!r.has_document_id && r.has_file_name);
        if (!is_chunk && !is_doc) continue;                            // some other item type in the container
        if (r.id.empty()) return fail(ORR_EINVAL, "Cosmos JSON: record " + std::to_string(i) + " has no id");
        if (r.has_created && !parse_iso_ticks(r.created, &ticks[i]))
            return fail(ORR_EINVAL, "Cosmos JSON: record " + std::to_string(i) + " has a createdAtUtc that is not an ISO 8601 date-time: " + r.created);
        if (is_doc) docs.push_back({r.id, r.file_name, ticks[i]});
        else {
            if (r.document_id.empty()) return fail(ORR_EINVAL, "Cosmos JSON: chunk " + r.id + " has no documentId");
            if (r.chunk_index < INT32_MIN || r.chunk_index > INT32_MAX) return fail(ORR_EINVAL, "Cosmos JSON: chunkIndex out of range");
            auto &v = by_doc[r.document_id];
            if (v.empty()) chunk_docs.push_back(r.document_id);
            bool replaced = false;                                     // an item id is unique: a later one replaces it (UpsertItemAsync)
            for (auto &prev : v)
                if (prev->id == r.id) { prev = &r; replaced = true; break; }
            if (!replaced) v.push_back(&r);
        }
    }
    for (const auto &d : docs) {
        const int r = orrh_store_upsert_document(s, d.id.c_str(), d.file.c_str(), d.ticks);
        if (r != ORR_OK) return r;
    }
    int64_t n_chunks = 0;
    for (const auto &doc : chunk_docs) {
        const auto &v = by_doc[doc];
        std::vector<const char *> ids, contents;
        std::vector<int32_t> index, emb_len;
        std::vector<int64_t> created;
        std::vector<float> emb;
        for (const Record *r : v) {
            ids.push_back(r->id.c_str());
            contents.push_back(r->content.c_str());
            index.push_back((int32_t)r->chunk_index);
            emb_len.push_back((int32_t)r->embedding.size());
            emb.insert(emb.end(), r->embedding.begin(), r->embedding.end());
            created.push_back(ticks[(size_t)(r - recs.data())]);
        }
        const int r = orrh_store_upsert_chunks(s, doc.c_str(), (int32_t)v.size(), ids.data(), index.data(), contents.data(),
                                               emb.empty() ? nullptr : emb.data(), emb_len.data(), created.data());
        if (r != ORR_OK) return r;
        n_chunks += (int64_t)v.size();
    }
    if (out_documents) *out_documents = (int64_t)docs.size();
    if (out_chunks) *out_chunks = n_chunks;
    return ORR_OK;
}

int orrh_store_export_cosmos_json(orrh_store *s, uint8_t **out_json, int64_t *out_len)
{
    if (!s || !out_json) return fail(ORR_EINVAL, "orrh_store_export_cosmos_json: null argument");
    *out_json = nullptr;
    std::string js = "[";
    bool first = true;
    {
        std::lock_guard<std::mutex> l(s->mu);
        for (const auto &kv : s->documents) {
            const auto &d = kv.second;
            auto it = s->chunks_by_document.find(d.id);
            js += first ? "\n" : ",\n";
            first = false;
            js += "{\"id\":"; orrh_detail::json_string(d.id, js);
            js += ",\"PartitionKey\":\"user:default\",\"type\":\"document\",\"fileName\":"; orrh_detail::json_string(d.file_name, js);
            js += ",\"sourceType\":\"file\",\"blobPath\":\"\",\"contentHash\":\"\",\"chunkCount\":";
            js += std::to_string(it == s->chunks_by_document.end() ? 0 : (long long)it->second.size());
            js += ",\"createdAtUtc\":"; orrh_detail::json_string(orrh_detail::iso_utc(d.created_ticks), js);
            js += "}";
        }
        for (const auto &doc : s->doc_order) {
            auto it = s->chunks_by_document.find(doc);
            if (it == s->chunks_by_document.end()) continue;
            for (const auto &c : it->second) {
                js += first ? "\n" : ",\n";
                first = false;
                js += "{\"id\":"; orrh_detail::json_string(c.id, js);
                js += ",\"PartitionKey\":\"user:default\",\"type\":\"chunk\",\"documentId\":"; orrh_detail::json_string(c.document_id, js);
                js += ",\"chunkIndex\":" + std::to_string(c.chunk_index);
                js += ",\"content\":"; orrh_detail::json_string(c.content, js);
                js += ",\"embedding\":";
                if (c.embedding.empty()) js += "null";
                else {
                    js.push_back('[');
                    for (size_t i = 0; i < c.embedding.size(); ++i) { if (i) js.push_back(','); json_float(c.embedding[i], js); }
                    js.push_back(']');
                }
                js += ",\"createdAtUtc\":"; orrh_detail::json_string(orrh_detail::iso_utc(c.created_ticks), js);
                js += "}";
            }
        }
    }
    js += "\n]\n";
    uint8_t *buf = static_cast<uint8_t *>(malloc(js.size() + 1));
    if (!buf) return fail(ORR_ENOMEM, "out of memory");
    memcpy(buf, js.data(), js.size() + 1);
    *out_json = buf;
    if (out_len) *out_len = (int64_t)js.size();
    return ORR_OK;
}

}  // extern "C"
