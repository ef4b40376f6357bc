/*
 * recall_oracle.h -- CPU restatement of the reference's hybrid recall-search
 * scorer.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
 * link or call this library.  The product path (omni-recall-rag_amd/) never does.
 *
 * Reference (C#, read as text; it cannot be compiled in this image -- no .NET
 * SDK): /root/reference/src/OmniRecall.Api/Services/RecallSearchService.cs and
 * friends.  Every function cites the file:line it follows.
 *
 * Pinning status: the reference's own tests pin only rank-1 identity on five
 * cases (tests/golden/reference_kats.json replays all of them through this
 * oracle).  No reference test asserts a numeric score, so NUMERIC parity is
 * unpinned by the reference; the hand-derived values in the KAT file
 * (1.0 / 0.3 / 0.1) are the only numeric anchors.
 */
#ifndef RECALL_ORACLE_H
#define RECALL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One stored chunk table (CosmosChunkRecord rows,
 * Data/Models/CosmosIngestionRecords.cs:19-30) in the store's enumeration
 * order.  Embeddings are ragged so that null / empty / wrong-dimension rows
 * can be expressed: row r owns emb[emb_off[r] .. emb_off[r]+emb_len[r]);
 * emb_len[r] == 0 stands for a null or empty list. */
typedef struct orc_corpus {
    int64_t n_chunks;
    const float *emb;
    const int64_t *emb_off;      /* [n_chunks]   offset in floats              */
    const int32_t *emb_len;      /* [n_chunks]   number of floats (0 = null)   */
    const int64_t *created_ticks;/* [n_chunks]   DateTime.Ticks (100 ns)       */
    const uint8_t *content;      /* UTF-8 pool, original case                  */
    const int64_t *content_off;  /* [n_chunks+1]                               */
} orc_corpus;

/* RecallSearchService.cs:69-88 */
double orc_cosine(const float *a, int64_t na, const float *b, int64_t nb);

/* RecallSearchService.cs:95-108: Split / ToLowerInvariant / Distinct / stop
 * words.  Writes the surviving terms (lowercased UTF-8, concatenated) to
 * `terms`, their boundaries to `term_off[0..T]`; returns T (>= 0), or -1 if a
 * buffer is too small. */
int32_t orc_query_terms(const uint8_t *query, int64_t query_len,
                        uint8_t *terms, int64_t terms_cap,
                        int32_t *term_off, int32_t term_off_cap);

/* RecallSearchService.cs:90-113 */
double orc_keyword_score(const uint8_t *query, int64_t query_len,
                         const uint8_t *content, int64_t content_len);

/* RecallSearchService.cs:115-119 with a frozen clock */
double orc_recency(int64_t created_ticks, int64_t now_ticks);

/* RecallSearchService.cs:59-67 */
double orc_score_chunk(const orc_corpus *c, int64_t row,
                       const float *qvec, int64_t qdim,
                       const uint8_t *query, int64_t query_len,
                       int64_t now_ticks);

/* InMemoryIngestionStore.cs:57-65: stable OrderByDescending(CreatedAtUtc),
 * Take(max(1,max_count)).  out_order must hold n_chunks entries; returns the
 * number of candidates. */
int64_t orc_recent_chunks(const orc_corpus *c, int64_t max_count,
                          int64_t *out_order);

/* RecallSearchService.cs:26-37 + :51.  candidate_limit is the
 * GetRecentChunksAsync argument (300 in the reference).  Outputs hold
 * max(1,topk) entries; returns the number of citations.  out_scores are the
 * unrounded doubles, out_rounded is Math.Round(score, 4).  n_threads <= 1 runs
 * the scoring loop on the calling thread; > 1 splits the candidate list across
 * that many pthreads (same arithmetic per chunk, so results are identical). */
int64_t orc_search(const orc_corpus *c, int64_t candidate_limit,
                   const float *qvec, int64_t qdim,
                   const uint8_t *query, int64_t query_len,
                   int64_t now_ticks, int32_t topk, int32_t n_threads,
                   int64_t *out_rows, double *out_scores, double *out_rounded);

/* Raw per-candidate scores in candidate order (for kernel-level parity tests):
 * fills out_order[n_cand] and out_scores[n_cand]; returns n_cand. */
int64_t orc_score_all(const orc_corpus *c, int64_t candidate_limit,
                      const float *qvec, int64_t qdim,
                      const uint8_t *query, int64_t query_len,
                      int64_t now_ticks, int32_t n_threads,
                      int64_t *out_order, double *out_scores);

/* Sequential fp64 sum of fp32-rounded products, RecallSearchService.cs:74-82
 * (one accumulator). */
double orc_dot(const float *a, const float *b, int64_t n);

/* Math.Round(x, 4), RecallSearchService.cs:51 */
double orc_round4(double x);

/* TextSnippetHelper.cs:5-11.  Returns bytes written (UTF-8), or -1. */
int64_t orc_snippet(const uint8_t *content, int64_t content_len, int32_t max_chars,
                    uint8_t *out, int64_t out_cap);

/* string.IsNullOrWhiteSpace, RecallSearchService.cs:22,92 */
int32_t orc_is_blank(const uint8_t *s, int64_t len);

/* ToLowerInvariant on UTF-8 (same length or shorter/longer by re-encoding);
 * returns bytes written or -1. */
int64_t orc_lower_invariant(const uint8_t *s, int64_t len, uint8_t *out, int64_t out_cap);

#ifdef __cplusplus
}
#endif
#endif
