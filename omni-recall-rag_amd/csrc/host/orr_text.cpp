// orr_text.cpp -- host-side string semantics of the recall-search path, mirroring
// the .NET calls the reference makes (see include/omnirecall_host.h).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/omnirecall_host.h"

namespace orrh {

using u32s = std::u32string;

struct CaseRun { char32_t first, last; int32_t delta, step; };
static const CaseRun kLowerRuns[] = {
#include "case_table.inc"
};

// char.ToLowerInvariant
static char32_t to_lower_invariant(char32_t c)
{
    if (c < 0x80) return (c >= U'A' && c <= U'Z') ? c + 32 : c;
    const CaseRun *b = kLowerRuns, *e = kLowerRuns + sizeof(kLowerRuns) / sizeof(kLowerRuns[0]);
    const CaseRun *it = std::upper_bound(b, e, c, [](char32_t v, const CaseRun &r) { return v < r.first; });
    if (it == b) return c;
    --it;
    if (c > it->last || (c - it->first) % it->step != 0) return c;
    return (char32_t)((int32_t)c + it->delta);
}

// char.IsWhiteSpace
static bool is_white_space(char32_t c)
{
    switch (c) {
    case 0x09: case 0x0A: case 0x0B: case 0x0C: case 0x0D: case 0x20: case 0x85: case 0xA0:
    case 0x1680: case 0x2028: case 0x2029: case 0x202F: case 0x205F: case 0x3000:
        return true;
    default:
        return c >= 0x2000 && c <= 0x200A;
    }
}

// UTF-8 -> code points; malformed bytes become U+FFFD one at a time.
static u32s decode(const uint8_t *s, int64_t n)
{
    u32s out;
    out.reserve((size_t)std::max<int64_t>(n, 0));
    for (int64_t i = 0; i < n;) {
        const uint8_t b = s[i];
        int extra = b < 0x80 ? 0 : (b >> 5) == 6 ? 1 : (b >> 4) == 14 ? 2 : (b >> 3) == 30 ? 3 : -1;
        char32_t cp = 0xFFFD;
        int used = 1;
        if (extra == 0) {
            cp = b;
        } else if (extra > 0 && i + extra < n) {
            char32_t v = b & (0x3F >> extra);
            bool ok = true;
            for (int k = 1; k <= extra && ok; ++k) {
                ok = (s[i + k] & 0xC0) == 0x80;
                v = (v << 6) | (s[i + k] & 0x3F);
            }
            static const char32_t min_for[4] = {0, 0x80, 0x800, 0x10000};
            if (ok && v >= min_for[extra] && v <= 0x10FFFF && !(v >= 0xD800 && v <= 0xDFFF)) {
                cp = v;
                used = extra + 1;
            }
        }
        out.push_back(cp);
        i += used;
    }
    return out;
}

static void encode(char32_t c, std::string &out)
{
    if (c < 0x80) {
        out.push_back((char)c);
    } else if (c < 0x800) {
        out.push_back((char)(0xC0 | (c >> 6)));
        out.push_back((char)(0x80 | (c & 0x3F)));
    } else if (c < 0x10000) {
        out.push_back((char)(0xE0 | (c >> 12)));
        out.push_back((char)(0x80 | ((c >> 6) & 0x3F)));
        out.push_back((char)(0x80 | (c & 0x3F)));
    } else {
        out.push_back((char)(0xF0 | (c >> 18)));
        out.push_back((char)(0x80 | ((c >> 12) & 0x3F)));
        out.push_back((char)(0x80 | ((c >> 6) & 0x3F)));
        out.push_back((char)(0x80 | (c & 0x3F)));
    }
}

static std::string encode(const u32s &s)
{
    std::string out;
    out.reserve(s.size());
    for (char32_t c : s) encode(c, out);
    return out;
}

static const char *const kStopWords[] = {   // RecallSearchService.cs:13-18
    "a", "an", "and", "are", "as", "at", "be", "by", "for", "from", "how", "in", "is", "it",
    "of", "on", "or", "that", "the", "to", "was", "what", "when", "where", "which", "who", "why", "with"};

static bool is_stop_word(const u32s &t)
{
    for (const char *w : kStopWords) {
        const size_t n = strlen(w);
        if (n != t.size()) continue;
        bool eq = true;
        for (size_t i = 0; i < n && eq; ++i) eq = t[i] == (char32_t)(unsigned char)w[i];
        if (eq) return true;
    }
    return false;
}

bool is_blank(const uint8_t *s, int64_t len)
{
    if (!s || len <= 0) return true;
    for (char32_t c : decode(s, len))
        if (!is_white_space(c)) return false;
    return true;
}

std::string lower_invariant(const uint8_t *s, int64_t len)
{
    u32s cps = decode(s, len);
    for (auto &c : cps) c = to_lower_invariant(c);
    return encode(cps);
}

// RecallSearchService.cs:95-108
std::vector<std::string> query_terms(const uint8_t *query, int64_t len)
{
    std::vector<u32s> raw;
    u32s cur;
    auto flush = [&]() {
        if (cur.empty()) return;
        if (std::find(raw.begin(), raw.end(), cur) == raw.end()) raw.push_back(cur);   // Distinct, first occurrence
        cur.clear();
    };
    for (char32_t c : decode(query, std::max<int64_t>(len, 0))) {
        if (is_white_space(c)) flush();
        else cur.push_back(to_lower_invariant(c));
    }
    flush();
    std::vector<u32s> kept;
    for (const auto &t : raw)
        if (!is_stop_word(t)) kept.push_back(t);
    const std::vector<u32s> &use = kept.empty() ? raw : kept;
    std::vector<std::string> out;
    for (const auto &t : use) out.push_back(encode(t));
    return out;
}

// TextSnippetHelper.cs:5-11; lengths are UTF-16 code units like string.Length.
std::string build_snippet(const uint8_t *content, int64_t len, int32_t max_chars)
{
    u32s cps = decode(content, std::max<int64_t>(len, 0));
    for (auto &c : cps)
        if (c == U'\n' || c == U'\r') c = U' ';
    size_t b = 0, e = cps.size();
    while (b < e && is_white_space(cps[b])) ++b;
    while (e > b && is_white_space(cps[e - 1])) --e;
    int64_t units = 0;
    for (size_t i = b; i < e; ++i) units += cps[i] >= 0x10000 ? 2 : 1;
    std::string out;
    if (units <= max_chars) {
        for (size_t i = b; i < e; ++i) encode(cps[i], out);
        return out;
    }
    int64_t used = 0;
    for (size_t i = b; i < e; ++i) {
        const int w = cps[i] >= 0x10000 ? 2 : 1;
        if (used + w > max_chars) {
            if (used < max_chars) encode(0xFFFD, out);   // a cut through a surrogate pair
            break;
        }
        encode(cps[i], out);
        used += w;
    }
    out += "...";
    return out;
}

double round4(double x)
{
    if (std::fabs(x) < 1e16) {
        x *= 1e4;
        x = std::nearbyint(x);   // MidpointRounding.ToEven under the default rounding mode
        x /= 1e4;
    }
    return x;
}

}  // namespace orrh

extern "C" {

int32_t orrh_is_blank(const uint8_t *s, int64_t len) { return orrh::is_blank(s, len) ? 1 : 0; }

int64_t orrh_lower_invariant(const uint8_t *s, int64_t len, uint8_t *out, int64_t out_cap)
{
    const std::string r = orrh::lower_invariant(s, std::max<int64_t>(len, 0));
    if ((int64_t)r.size() > out_cap) return -1;
    memcpy(out, r.data(), r.size());
    return (int64_t)r.size();
}

int32_t orrh_query_terms(const uint8_t *query, int64_t query_len, uint8_t *terms, int64_t terms_cap,
                         uint32_t *term_off, int32_t term_off_cap)
{
    if (orrh::is_blank(query, query_len)) return 0;
    const auto ts = orrh::query_terms(query, query_len);
    if ((int32_t)ts.size() + 1 > term_off_cap) return -1;
    int64_t m = 0;
    term_off[0] = 0;
    for (size_t i = 0; i < ts.size(); ++i) {
        if (m + (int64_t)ts[i].size() > terms_cap) return -1;
        memcpy(terms + m, ts[i].data(), ts[i].size());
        m += (int64_t)ts[i].size();
        term_off[i + 1] = (uint32_t)m;
    }
    return (int32_t)ts.size();
}

int64_t orrh_build_snippet(const uint8_t *content, int64_t content_len, int32_t max_chars, uint8_t *out, int64_t out_cap)
{
    const std::string r = orrh::build_snippet(content, content_len, max_chars);
    if ((int64_t)r.size() > out_cap) return -1;
    memcpy(out, r.data(), r.size());
    return (int64_t)r.size();
}

double orrh_round4(double x) { return orrh::round4(x); }

// ChatOrchestrationService.cs:58-65
int32_t orrh_has_sufficient_evidence(const double *citation_scores, int32_t n_citations, int32_t minimum_citation_count,
                                     double minimum_strong_citation_score)
{
    if (n_citations < 0 || (n_citations > 0 && !citation_scores)) return 0;
    if (n_citations < std::max(1, minimum_citation_count)) return 0;                 // :60-61
    const double threshold = std::max(0.0, minimum_strong_citation_score);           // :63  (Math.Max(0d, NaN) is NaN: nothing passes)
    if (threshold != threshold) return 0;
    for (int32_t i = 0; i < n_citations; ++i)
        if (citation_scores[i] >= threshold) return 1;                               // :64
    return 0;
}

// ChatOrchestrationService.cs:85: fixed-point with four decimals.  The value was rounded to four
// decimals before (RecallSearchService.cs:51), so the nearest four-decimal string is unambiguous and
// the C library's correctly rounded "%.4f" prints the digits .NET prints.
int32_t orrh_format_score_f4(double rounded_score, char *out, int32_t out_cap)
{
    if (!out || out_cap <= 0) return -1;
    char buf[64];
    int n;
    if (rounded_score != rounded_score) n = snprintf(buf, sizeof buf, "NaN");
    else if (rounded_score == __builtin_inf()) n = snprintf(buf, sizeof buf, "\xE2\x88\x9E");      // U+221E, .NET's InvariantCulture symbol
    else if (rounded_score == -__builtin_inf()) n = snprintf(buf, sizeof buf, "-\xE2\x88\x9E");
    else n = snprintf(buf, sizeof buf, "%.4f", rounded_score == 0.0 ? 0.0 : rounded_score);   // "-0.0000" is "-0.0000" in .NET Core 3.0+ too, but a rounded score of -0 never reaches here
    if (n < 0 || n > out_cap) return -1;
    memcpy(out, buf, (size_t)n);
    return n;
}

}  // extern "C"
