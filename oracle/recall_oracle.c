/*
 * recall_oracle.c -- CPU restatement of the reference scorer.
 * TEST INFRASTRUCTURE ONLY (see recall_oracle.h).
 *
 * Follows /root/reference/src/OmniRecall.Api/Services/RecallSearchService.cs
 * line by line in *behaviour* (written from scratch in C; the reference is C#
 * and cannot be built in this image).  Deliberately scalar and
 * reference-shaped: the query norm is recomputed per chunk, the query is
 * re-tokenised and the content re-lowercased per chunk, exactly like the C#.
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off, no -ffast-math: the
 * arithmetic below must not be re-associated or fused).
 *
 * Arithmetic that lives in the .NET BCL rather than in the reference tree
 * (net10.0, not vendored) and how it is restated here:
 *   float*float            -> IEEE binary32 multiply (RyuJIT mulss)
 *   double += float        -> widen then IEEE binary64 add
 *   Math.Sqrt              -> sqrt()  (IEEE exact)
 *   Math.Exp               -> libm exp() (what the CLR PAL calls on Linux)
 *   Math.Round(x, 4)       -> rint(x * 1e4) / 1e4 for |x| < 1e16
 *   TimeSpan.TotalDays     -> (double)ticks / 864000000000
 *   OrderByDescending/ThenByDescending -> stable merge sort
 *   double.CompareTo       -> NaN sorts below everything, -0 == +0
 *   char.IsWhiteSpace      -> the explicit set in is_ws()
 *   ToLowerInvariant       -> simple case mapping table (lower_table.inc)
 */
#include "recall_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ text */

static const struct { uint32_t from, to; } k_lower_pairs[] = {
#include "lower_table.inc"
};
#define N_LOWER_PAIRS ((int)(sizeof(k_lower_pairs) / sizeof(k_lower_pairs[0])))

/* char.ToLowerInvariant, one code point */
static uint32_t cp_lower(uint32_t cp)
{
    if (cp < 0x80)
        return (cp >= 'A' && cp <= 'Z') ? cp + 32 : cp;
    int lo = 0, hi = N_LOWER_PAIRS - 1;
    while (lo <= hi) {
        int mid = (lo + hi) >> 1;
        if (k_lower_pairs[mid].from == cp)
            return k_lower_pairs[mid].to;
        if (k_lower_pairs[mid].from < cp)
            lo = mid + 1;
        else
            hi = mid - 1;
    }
    return cp;
}

/* char.IsWhiteSpace */
static int is_ws(uint32_t cp)
{
    if (cp >= 0x09 && cp <= 0x0D) return 1;
    if (cp == 0x20 || cp == 0x85 || cp == 0xA0 || cp == 0x1680) return 1;
    if (cp >= 0x2000 && cp <= 0x200A) return 1;
    if (cp == 0x2028 || cp == 0x2029 || cp == 0x202F || cp == 0x205F || cp == 0x3000) return 1;
    return 0;
}

/* Decode one UTF-8 scalar; malformed input yields U+FFFD and consumes one byte. */
static int64_t utf8_next(const uint8_t *s, int64_t i, int64_t n, uint32_t *out)
{
    uint8_t b0 = s[i];
    if (b0 < 0x80) { *out = b0; return i + 1; }
    int need = 0; uint32_t cp = 0, min = 0;
    if ((b0 & 0xE0) == 0xC0) { need = 1; cp = b0 & 0x1F; min = 0x80; }
    else if ((b0 & 0xF0) == 0xE0) { need = 2; cp = b0 & 0x0F; min = 0x800; }
    else if ((b0 & 0xF8) == 0xF0) { need = 3; cp = b0 & 0x07; min = 0x10000; }
    else { *out = 0xFFFD; return i + 1; }
    if (i + need >= n) { *out = 0xFFFD; return i + 1; }   /* truncated sequence */
    for (int k = 1; k <= need; k++) {
        uint8_t b = s[i + k];
        if ((b & 0xC0) != 0x80) { *out = 0xFFFD; return i + 1; }
        cp = (cp << 6) | (b & 0x3F);
    }
    if (cp < min || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) { *out = 0xFFFD; return i + 1; }
    *out = cp;
    return i + 1 + need;
}

static int64_t utf8_decode(const uint8_t *s, int64_t n, uint32_t *out)
{
    int64_t i = 0, m = 0;
    while (i < n)
        i = utf8_next(s, i, n, &out[m++]);
    return m;
}

static int64_t utf8_put(uint32_t cp, uint8_t *out)
{
    if (cp < 0x80) { out[0] = (uint8_t)cp; return 1; }
    if (cp < 0x800) { out[0] = 0xC0 | (cp >> 6); out[1] = 0x80 | (cp & 0x3F); return 2; }
    if (cp < 0x10000) { out[0] = 0xE0 | (cp >> 12); out[1] = 0x80 | ((cp >> 6) & 0x3F); out[2] = 0x80 | (cp & 0x3F); return 3; }
    out[0] = 0xF0 | (cp >> 18); out[1] = 0x80 | ((cp >> 12) & 0x3F); out[2] = 0x80 | ((cp >> 6) & 0x3F); out[3] = 0x80 | (cp & 0x3F);
    return 4;
}

int32_t orc_is_blank(const uint8_t *s, int64_t len)
{
    if (!s || len <= 0) return 1;
    int64_t i = 0;
    while (i < len) {
        uint32_t cp;
        i = utf8_next(s, i, len, &cp);
        if (!is_ws(cp)) return 0;
    }
    return 1;
}

int64_t orc_lower_invariant(const uint8_t *s, int64_t len, uint8_t *out, int64_t out_cap)
{
    int64_t i = 0, m = 0;
    while (i < len) {
        uint32_t cp;
        i = utf8_next(s, i, len, &cp);
        if (m + 4 > out_cap) return -1;
        m += utf8_put(cp_lower(cp), out + m);
    }
    return m;
}

/* RecallSearchService.cs:13-18 */
static const char *const k_stop_words[] = {
    "a", "an", "and", "are", "as", "at", "be", "by", "for", "from", "how", "in", "is",
    "it", "of", "on", "or", "that", "the", "to", "was", "what", "when", "where", "which",
    "who", "why", "with"
};
#define N_STOP ((int)(sizeof(k_stop_words) / sizeof(k_stop_words[0])))

static int is_stop_word(const uint32_t *t, int64_t n)
{
    for (int w = 0; w < N_STOP; w++) {
        const char *sw = k_stop_words[w];
        int64_t L = (int64_t)strlen(sw);
        if (L != n) continue;
        int64_t k = 0;
        while (k < n && t[k] == (uint32_t)(unsigned char)sw[k]) k++;
        if (k == n) return 1;
    }
    return 0;
}

/* A tokenised query: code points in `cps`, term t is cps[off[t] .. off[t+1]). */
typedef struct {
    uint32_t *cps;
    int64_t *off;
    int32_t n_terms;
} term_list;

static void term_list_free(term_list *tl)
{
    free(tl->cps); free(tl->off);
    tl->cps = NULL; tl->off = NULL; tl->n_terms = 0;
}

/* RecallSearchService.cs:95-108 */
static int build_terms(const uint8_t *query, int64_t query_len, term_list *out)
{
    out->cps = NULL; out->off = NULL; out->n_terms = 0;
    if (query_len <= 0) return 0;
    uint32_t *q = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)query_len);
    uint32_t *raw = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)query_len);
    int64_t *raw_off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(query_len + 2));
    if (!q || !raw || !raw_off) { free(q); free(raw); free(raw_off); return -1; }
    int64_t nq = utf8_decode(query, query_len, q);

    /* :95 Split(null, RemoveEmptyEntries | TrimEntries); :96 ToLowerInvariant; :97 Distinct */
    int32_t n_raw = 0; int64_t w = 0;
    raw_off[0] = 0;
    int64_t i = 0;
    while (i < nq) {
        while (i < nq && is_ws(q[i])) i++;
        if (i >= nq) break;
        int64_t start = w;
        while (i < nq && !is_ws(q[i])) raw[w++] = cp_lower(q[i++]);
        int64_t len = w - start;
        int dup = 0;
        for (int32_t t = 0; t < n_raw && !dup; t++) {
            int64_t tl = raw_off[t + 1] - raw_off[t];
            if (tl == len && memcmp(raw + raw_off[t], raw + start, sizeof(uint32_t) * (size_t)len) == 0)
                dup = 1;
        }
        if (dup) { w = start; continue; }
        n_raw++;
        raw_off[n_raw] = w;
    }
    free(q);
    if (n_raw == 0) { free(raw); free(raw_off); return 0; }           /* :100-101 */

    /* :103-108 drop stop words unless that empties the list */
    uint32_t *kept = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(w > 0 ? w : 1));
    int64_t *kept_off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_raw + 1));
    if (!kept || !kept_off) { free(raw); free(raw_off); free(kept); free(kept_off); return -1; }
    int32_t n_kept = 0; int64_t kw = 0;
    kept_off[0] = 0;
    for (int32_t t = 0; t < n_raw; t++) {
        int64_t len = raw_off[t + 1] - raw_off[t];
        if (is_stop_word(raw + raw_off[t], len)) continue;
        memcpy(kept + kw, raw + raw_off[t], sizeof(uint32_t) * (size_t)len);
        kw += len;
        kept_off[++n_kept] = kw;
    }
    if (n_kept == 0) {
        free(kept); free(kept_off);
        out->cps = raw; out->off = raw_off; out->n_terms = n_raw;
    } else {
        free(raw); free(raw_off);
        out->cps = kept; out->off = kept_off; out->n_terms = n_kept;
    }
    return 0;
}

int32_t orc_query_terms(const uint8_t *query, int64_t query_len,
                        uint8_t *terms, int64_t terms_cap,
                        int32_t *term_off, int32_t term_off_cap)
{
    term_list tl;
    if (orc_is_blank(query, query_len)) return 0;                      /* :92 */
    if (build_terms(query, query_len, &tl) != 0) return -1;
    if (tl.n_terms + 1 > term_off_cap) { term_list_free(&tl); return -1; }
    int64_t m = 0;
    term_off[0] = 0;
    for (int32_t t = 0; t < tl.n_terms; t++) {
        for (int64_t k = tl.off[t]; k < tl.off[t + 1]; k++) {
            if (m + 4 > terms_cap) { term_list_free(&tl); return -1; }
            m += utf8_put(tl.cps[k], terms + m);
        }
        term_off[t + 1] = (int32_t)m;
    }
    int32_t n = tl.n_terms;
    term_list_free(&tl);
    return n;
}

/* string.Contains(term, StringComparison.Ordinal) on code points */
static int contains_ordinal(const uint32_t *hay, int64_t nh, const uint32_t *needle, int64_t nn)
{
    if (nn == 0) return 1;
    if (nn > nh) return 0;
    for (int64_t p = 0; p + nn <= nh; p++) {
        if (hay[p] != needle[0]) continue;
        int64_t k = 1;
        while (k < nn && hay[p + k] == needle[k]) k++;
        if (k == nn) return 1;
    }
    return 0;
}

/* RecallSearchService.cs:90-113 */
double orc_keyword_score(const uint8_t *query, int64_t query_len,
                         const uint8_t *content, int64_t content_len)
{
    if (orc_is_blank(query, query_len) || orc_is_blank(content, content_len))
        return 0.0;                                                    /* :92-93 */
    term_list tl;
    if (build_terms(query, query_len, &tl) != 0 || tl.n_terms == 0) {  /* :95-101 */
        term_list_free(&tl);
        return 0.0;
    }
    /* :110 content.ToLowerInvariant() -- re-done per chunk, as in the reference */
    uint32_t *lower = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)content_len);
    if (!lower) { term_list_free(&tl); return 0.0; }
    int64_t nc = utf8_decode(content, content_len, lower);
    for (int64_t i = 0; i < nc; i++) lower[i] = cp_lower(lower[i]);
    /* :111 */
    int32_t matches = 0;
    for (int32_t t = 0; t < tl.n_terms; t++)
        matches += contains_ordinal(lower, nc, tl.cps + tl.off[t], tl.off[t + 1] - tl.off[t]);
    double score = (double)matches / (double)tl.n_terms;              /* :112 */
    free(lower);
    term_list_free(&tl);
    return score;
}

/* ------------------------------------------------------------ arithmetic */

/* float*float must be evaluated in binary32 (SSE scalar), never in a wider
 * format, or the "product rounded to fp32" step of the reference is lost. */
#if !defined(FLT_EVAL_METHOD) || FLT_EVAL_METHOD != 0
#error "oracle needs FLT_EVAL_METHOD == 0 (x86-64 SSE or equivalent)"
#endif

/* RecallSearchService.cs:69-88.  a[i]*b[i] is a float*float product rounded to
 * binary32, then widened and added to a binary64 accumulator, in index order. */
double orc_cosine(const float *a, int64_t na, const float *b, int64_t nb)
{
    if (na == 0 || b == NULL || nb == 0 || na != nb)                   /* :71-72 */
        return 0.0;
    double dot = 0.0, norm_a = 0.0, norm_b = 0.0;
    for (int64_t i = 0; i < na; i++) {                                 /* :77-82 */
        float p_ab = a[i] * b[i];
        float p_aa = a[i] * a[i];
        float p_bb = b[i] * b[i];
        dot += (double)p_ab;
        norm_a += (double)p_aa;
        norm_b += (double)p_bb;
    }
    if (norm_a <= 0.0 || norm_b <= 0.0)                                /* :84-85 */
        return 0.0;
    return dot / (sqrt(norm_a) * sqrt(norm_b));                        /* :87 */
}

double orc_dot(const float *a, const float *b, int64_t n)
{
    double dot = 0.0;
    for (int64_t i = 0; i < n; i++) {
        float p = a[i] * b[i];
        dot += (double)p;
    }
    return dot;
}

/* RecallSearchService.cs:115-119; TimeSpan.TotalDays = (double)ticks / TicksPerDay */
double orc_recency(int64_t created_ticks, int64_t now_ticks)
{
    double total_days = (double)(now_ticks - created_ticks) / 864000000000.0;
    double age_days = total_days > 0.0 ? total_days : 0.0;   /* Math.Max(0d, x); NaN impossible here */
    return exp(-age_days / 30.0);
}

/* RecallSearchService.cs:59-67 */
double orc_score_chunk(const orc_corpus *c, int64_t row,
                       const float *qvec, int64_t qdim,
                       const uint8_t *query, int64_t query_len,
                       int64_t now_ticks)
{
    const float *b = c->emb_len[row] > 0 ? c->emb + c->emb_off[row] : NULL;
    double embedding_score = orc_cosine(qvec, qdim, b, c->emb_len[row]);
    double keyword_score = orc_keyword_score(query, query_len,
                                             c->content + c->content_off[row],
                                             c->content_off[row + 1] - c->content_off[row]);
    double recency_score = orc_recency(c->created_ticks[row], now_ticks);
    return (embedding_score * 0.7) + (keyword_score * 0.2) + (recency_score * 0.1);   /* :66 */
}

double orc_round4(double x)
{
    if (fabs(x) < 1e16) {
        x *= 1e4;
        x = rint(x);           /* MidpointRounding.ToEven */
        x /= 1e4;
    }
    return x;
}

/* ------------------------------------------------------- stable ordering */

/* double.CompareTo */
static int cmp_double(double a, double b)
{
    if (a < b) return -1;
    if (a > b) return 1;
    if (a == b) return 0;
    if (isnan(a)) return isnan(b) ? 0 : -1;
    return 1;
}

typedef int (*before_fn)(const void *ctx, int64_t x, int64_t y); /* <0: x first; 0: tie */

static void merge_sort_idx(int64_t *idx, int64_t *tmp, int64_t n, before_fn f, const void *ctx)
{
    if (n < 2) return;
    int64_t h = n / 2;
    merge_sort_idx(idx, tmp, h, f, ctx);
    merge_sort_idx(idx + h, tmp, n - h, f, ctx);
    int64_t i = 0, j = h, k = 0;
    while (i < h && j < n) {
        if (f(ctx, idx[j], idx[i]) < 0) tmp[k++] = idx[j++];   /* strict: ties keep the left run */
        else tmp[k++] = idx[i++];
    }
    while (i < h) tmp[k++] = idx[i++];
    while (j < n) tmp[k++] = idx[j++];
    memcpy(idx, tmp, sizeof(int64_t) * (size_t)n);
}

static int created_desc(const void *ctx, int64_t x, int64_t y)
{
    const int64_t *created = (const int64_t *)ctx;
    if (created[x] > created[y]) return -1;
    if (created[x] < created[y]) return 1;
    return 0;
}

/* InMemoryIngestionStore.cs:57-65 */
int64_t orc_recent_chunks(const orc_corpus *c, int64_t max_count, int64_t *out_order)
{
    int64_t n = c->n_chunks;
    if (n <= 0) return 0;
    int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    if (!tmp) return -1;
    for (int64_t i = 0; i < n; i++) out_order[i] = i;
    merge_sort_idx(out_order, tmp, n, created_desc, c->created_ticks);
    free(tmp);
    int64_t take = max_count > 1 ? max_count : 1;
    return take < n ? take : n;
}

typedef struct {
    const double *score;       /* indexed by candidate position */
    const int64_t *created;    /* indexed by candidate position */
} rank_ctx;

/* RecallSearchService.cs:34-35 */
static int score_then_created_desc(const void *vctx, int64_t x, int64_t y)
{
    const rank_ctx *r = (const rank_ctx *)vctx;
    int cs = cmp_double(r->score[x], r->score[y]);
    if (cs != 0) return -cs;
    if (r->created[x] > r->created[y]) return -1;
    if (r->created[x] < r->created[y]) return 1;
    return 0;
}

/* ---------------------------------------------------------------- search */

typedef struct {
    const orc_corpus *c;
    const int64_t *order;
    double *scores;
    const float *qvec; int64_t qdim;
    const uint8_t *query; int64_t query_len;
    int64_t now_ticks;
    int64_t begin, end;
} score_job;

static void *score_range(void *arg)
{
    score_job *j = (score_job *)arg;
    for (int64_t p = j->begin; p < j->end; p++)
        j->scores[p] = orc_score_chunk(j->c, j->order[p], j->qvec, j->qdim,
                                       j->query, j->query_len, j->now_ticks);
    return NULL;
}

int64_t orc_score_all(const orc_corpus *c, int64_t candidate_limit,
                      const float *qvec, int64_t qdim,
                      const uint8_t *query, int64_t query_len,
                      int64_t now_ticks, int32_t n_threads,
                      int64_t *out_order, double *out_scores)
{
    int64_t n_cand = orc_recent_chunks(c, candidate_limit, out_order);   /* :26 */
    if (n_cand <= 0) return n_cand;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    if (n_threads == 1 || n_cand < 2 * n_threads) {
        score_job j = { c, out_order, out_scores, qvec, qdim, query, query_len, now_ticks, 0, n_cand };
        score_range(&j);
        return n_cand;
    }
    pthread_t th[256];
    score_job jobs[256];
    for (int32_t t = 0; t < n_threads; t++) {
        score_job j = { c, out_order, out_scores, qvec, qdim, query, query_len, now_ticks,
                        n_cand * t / n_threads, n_cand * (t + 1) / n_threads };
        jobs[t] = j;
        pthread_create(&th[t], NULL, score_range, &jobs[t]);
    }
    for (int32_t t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    return n_cand;
}

int64_t orc_search(const orc_corpus *c, int64_t candidate_limit,
                   const float *qvec, int64_t qdim,
                   const uint8_t *query, int64_t query_len,
                   int64_t now_ticks, int32_t topk, int32_t n_threads,
                   int64_t *out_rows, double *out_scores, double *out_rounded)
{
    int64_t n = c->n_chunks;
    if (n <= 0) return 0;
    int64_t *order = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    double *scores = (double *)malloc(sizeof(double) * (size_t)n);
    int64_t *pos = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    int64_t *created = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    int64_t n_out = -1;
    if (order && scores && pos && tmp && created) {
        int64_t n_cand = orc_score_all(c, candidate_limit, qvec, qdim, query, query_len,
                                       now_ticks, n_threads, order, scores);   /* :26-33 */
        for (int64_t p = 0; p < n_cand; p++) { pos[p] = p; created[p] = c->created_ticks[order[p]]; }
        rank_ctx ctx = { scores, created };
        merge_sort_idx(pos, tmp, n_cand, score_then_created_desc, &ctx);      /* :34-35 */
        int64_t take = topk > 1 ? topk : 1;                                   /* :36 */
        n_out = take < n_cand ? take : n_cand;
        for (int64_t k = 0; k < n_out; k++) {
            out_rows[k] = order[pos[k]];
            out_scores[k] = scores[pos[k]];
            if (out_rounded) out_rounded[k] = orc_round4(scores[pos[k]]);     /* :51 */
        }
    }
    free(order); free(scores); free(pos); free(tmp); free(created);
    return n_out;
}

/* TextSnippetHelper.cs:5-11: Replace('\n',' ').Replace('\r',' ').Trim(); cut at
 * maxLength UTF-16 code units and append "...". */
int64_t orc_snippet(const uint8_t *content, int64_t content_len, int32_t max_chars,
                    uint8_t *out, int64_t out_cap)
{
    uint32_t *cps = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(content_len > 0 ? content_len : 1));
    if (!cps) return -1;
    int64_t n = utf8_decode(content, content_len, cps);
    for (int64_t i = 0; i < n; i++)
        if (cps[i] == '\n' || cps[i] == '\r') cps[i] = ' ';
    int64_t b = 0, e = n;
    while (b < e && is_ws(cps[b])) b++;
    while (e > b && is_ws(cps[e - 1])) e--;
    /* length in UTF-16 code units */
    int64_t units = 0;
    for (int64_t i = b; i < e; i++) units += cps[i] >= 0x10000 ? 2 : 1;
    int64_t m = 0;
    int truncated = units > max_chars;
    int64_t budget = truncated ? max_chars : units;
    int64_t used = 0;
    for (int64_t i = b; i < e; i++) {
        int w = cps[i] >= 0x10000 ? 2 : 1;
        if (used + w > budget) {
            /* a cut inside a surrogate pair leaves a lone high surrogate in C#;
             * it serialises as U+FFFD.  Emit that. */
            if (used < budget) {
                if (m + 4 > out_cap) { free(cps); return -1; }
                m += utf8_put(0xFFFD, out + m);
            }
            break;
        }
        if (m + 4 > out_cap) { free(cps); return -1; }
        m += utf8_put(cps[i], out + m);
        used += w;
    }
    if (truncated) {
        if (m + 3 > out_cap) { free(cps); return -1; }
        out[m++] = '.'; out[m++] = '.'; out[m++] = '.';
    }
    free(cps);
    return m;
}
