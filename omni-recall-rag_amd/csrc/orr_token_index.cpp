// orr_token_index.cpp -- seal-time build of the shard's token index (host side).
//
// Why it is exact: the query terms of RecallSearchService.cs:95 come out of a split on
// char.IsWhiteSpace, so no term contains whitespace; therefore a term is a substring of
// a chunk's (lowercased) content (:111) iff it is a substring of ONE maximal
// whitespace-free run ("token") of that content.  The index maps every distinct token of
// the shard to the ascending list of rows (candidate positions) that contain it; at query
// time the device scans the vocabulary for tokens containing the term and ORs their
// posting lists.
//
// Tokenisation here must use the same whitespace set as the host's query split
// (char.IsWhiteSpace), on the UTF-8 bytes of the already lowercased content.
#include "orr_token_index.h"

#include <algorithm>
#include <cstring>
#include <string_view>
#include <thread>
#include <unordered_map>

namespace orr {

// Length in bytes of the whitespace character starting at p (0 if none).  UTF-8 forms of
// U+0009-000D, 0020, 0085, 00A0, 1680, 2000-200A, 2028, 2029, 202F, 205F, 3000.
static inline int ws_len(const uint8_t *p, const uint8_t *end)
{
    const uint8_t b = p[0];
    if (b < 0x80) return ((b >= 0x09 && b <= 0x0D) || b == 0x20) ? 1 : 0;
    if (b == 0xC2) return (p + 1 < end && (p[1] == 0x85 || p[1] == 0xA0)) ? 2 : 0;
    if (b == 0xE1) return (p + 2 < end && p[1] == 0x9A && p[2] == 0x80) ? 3 : 0;
    if (b == 0xE2) {
        if (p + 2 >= end) return 0;
        if (p[1] == 0x80) return ((p[2] >= 0x80 && p[2] <= 0x8A) || p[2] == 0xA8 || p[2] == 0xA9 || p[2] == 0xAF) ? 3 : 0;
        if (p[1] == 0x81) return p[2] == 0x9F ? 3 : 0;
        return 0;
    }
    if (b == 0xE3) return (p + 2 < end && p[1] == 0x80 && p[2] == 0x80) ? 3 : 0;
    return 0;
}

namespace {

struct Partial {                       // one worker's view of a contiguous row range
    int64_t row_begin = 0, row_end = 0;
    std::unordered_map<std::string_view, uint32_t> ids;   // token -> local id
    std::vector<std::string_view> tokens;                 // local id -> token
    std::vector<uint32_t> occ;                            // per row: distinct local ids, concatenated
    std::vector<uint32_t> occ_rows;                       // row of each occ entry (relative to row_begin)
};

void tokenize_range(const uint8_t *pool, const uint64_t *cstart, const uint32_t *clen, Partial &w)
{
    std::vector<uint32_t> row_ids;
    w.ids.reserve(1 << 16);
    for (int64_t r = w.row_begin; r < w.row_end; ++r) {
        const uint8_t *p = pool + cstart[r], *end = p + clen[r];
        row_ids.clear();
        while (p < end) {
            int wl;
            while (p < end && (wl = ws_len(p, end)) > 0) p += wl;
            const uint8_t *t0 = p;
            while (p < end && ws_len(p, end) == 0) ++p;
            if (p > t0) {
                std::string_view tok(reinterpret_cast<const char *>(t0), (size_t)(p - t0));
                auto it = w.ids.find(tok);
                uint32_t id;
                if (it == w.ids.end()) {
                    id = (uint32_t)w.tokens.size();
                    w.ids.emplace(tok, id);
                    w.tokens.push_back(tok);
                } else {
                    id = it->second;
                }
                row_ids.push_back(id);
            }
        }
        std::sort(row_ids.begin(), row_ids.end());
        row_ids.erase(std::unique(row_ids.begin(), row_ids.end()), row_ids.end());
        for (uint32_t id : row_ids) {
            w.occ.push_back(id);
            w.occ_rows.push_back((uint32_t)(r - w.row_begin));
        }
    }
}

}  // namespace

void build_token_index(const uint8_t *pool, const uint64_t *cstart, const uint32_t *clen, int64_t n_rows,
                       int n_threads, TokenIndexHost &out)
{
    if (n_threads < 1) n_threads = 1;
    if ((int64_t)n_threads > n_rows / 4096 + 1) n_threads = (int)(n_rows / 4096 + 1);
    std::vector<Partial> parts((size_t)n_threads);
    for (int t = 0; t < n_threads; ++t) {
        parts[t].row_begin = n_rows * t / n_threads;
        parts[t].row_end = n_rows * (t + 1) / n_threads;
    }
    {
        std::vector<std::thread> th;
        for (int t = 1; t < n_threads; ++t)
            th.emplace_back(tokenize_range, pool, cstart, clen, std::ref(parts[t]));
        tokenize_range(pool, cstart, clen, parts[0]);
        for (auto &x : th) x.join();
    }
    // global vocabulary in order of first appearance (worker 0's tokens first)
    std::unordered_map<std::string_view, uint32_t> gids;
    std::vector<std::string_view> vocab;
    std::vector<std::vector<uint32_t>> remap((size_t)n_threads);
    for (int t = 0; t < n_threads; ++t) {
        remap[t].resize(parts[t].tokens.size());
        for (size_t i = 0; i < parts[t].tokens.size(); ++i) {
            auto it = gids.find(parts[t].tokens[i]);
            if (it == gids.end()) {
                const uint32_t id = (uint32_t)vocab.size();
                gids.emplace(parts[t].tokens[i], id);
                vocab.push_back(parts[t].tokens[i]);
                remap[t][i] = id;
            } else {
                remap[t][i] = it->second;
            }
        }
    }
    const size_t V = vocab.size();
    // postings in CSR form; rows ascend because workers cover ascending row ranges in order
    out.post_off.assign(V + 1, 0);
    for (int t = 0; t < n_threads; ++t)
        for (uint32_t id : parts[t].occ) out.post_off[remap[t][id] + 1]++;
    for (size_t v = 0; v < V; ++v) out.post_off[v + 1] += out.post_off[v];
    out.post_rows.resize(out.post_off[V]);
    std::vector<uint64_t> cursor(out.post_off.begin(), out.post_off.end() - 1);
    for (int t = 0; t < n_threads; ++t) {
        const Partial &w = parts[t];
        for (size_t i = 0; i < w.occ.size(); ++i)
            out.post_rows[cursor[remap[t][w.occ[i]]]++] = (uint32_t)(w.row_begin + w.occ_rows[i]);
    }
    // vocabulary pool in the scan kernel's layout: 16-byte aligned starts, space padding
    out.vstart.resize(V);
    out.vlen.resize(V);
    uint64_t cur = 0;
    for (size_t v = 0; v < V; ++v) {
        out.vstart[v] = cur;
        out.vlen[v] = (uint32_t)vocab[v].size();
        cur += (vocab[v].size() / 16 + 1) * 16;
    }
    out.vpool.assign(cur, 0x20);
    for (size_t v = 0; v < V; ++v) memcpy(out.vpool.data() + out.vstart[v], vocab[v].data(), vocab[v].size());
}

}  // namespace orr
