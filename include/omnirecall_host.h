/*
 * omnirecall_host.h -- C API of libomnirecall_host.so: the host side of the
 * recall-search path written in C++ because the reference's own host language
 * (C#, net10.0) has no toolchain in this image.  It mirrors, name for name, the
 * pieces of the reference that stay on the host around the one native call:
 *
 *   query tokenisation      RecallSearchService.cs:92-108  (KeywordScore prologue)
 *   ToLowerInvariant        RecallSearchService.cs:96,110
 *   IsNullOrWhiteSpace      RecallSearchService.cs:22,92-93
 *   BuildSnippet            TextSnippetHelper.cs:5-11
 *   Math.Round(score, 4)    RecallSearchService.cs:51
 *
 * A .NET host does not need this library: it calls the BCL for these and binds
 * include/omnirecall_hip.h directly.  Pure host code, no GPU.
 */
#ifndef OMNIRECALL_HOST_H
#define OMNIRECALL_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* string.IsNullOrWhiteSpace on UTF-8 */
int32_t orrh_is_blank(const uint8_t *s, int64_t len);

/* string.ToLowerInvariant on UTF-8; returns bytes written, or -1 if out_cap is
 * too small (4*len+4 always suffices). */
int64_t orrh_lower_invariant(const uint8_t *s, int64_t len, uint8_t *out, int64_t out_cap);

/* The queryTerms array of RecallSearchService.cs:95-108: Split on whitespace,
 * ToLowerInvariant, Distinct, drop the 28 stop words unless that leaves nothing.
 * Terms are written back to back into `terms`, boundaries into term_off[0..T].
 * Returns T >= 0, or -1 when a buffer is too small. */
int32_t orrh_query_terms(const uint8_t *query, int64_t query_len, uint8_t *terms, int64_t terms_cap,
                         uint32_t *term_off, int32_t term_off_cap);

/* TextSnippetHelper.BuildSnippet(content, max_chars); returns bytes written or -1. */
int64_t orrh_build_snippet(const uint8_t *content, int64_t content_len, int32_t max_chars,
                           uint8_t *out, int64_t out_cap);

/* Math.Round(x, 4): banker's rounding of x*1e4 */
double orrh_round4(double x);

/* ---- the second consumer of the scores -----------------------
 * ChatOrchestrationService.HasSufficientEvidence (ChatOrchestrationService.cs:58-65): at least
 * max(1, minimum_citation_count) citations and one whose (4-decimal) score reaches
 * max(0, minimum_strong_citation_score).  Returns 1 or 0. */
int32_t orrh_has_sufficient_evidence(const double *citation_scores, int32_t n_citations, int32_t minimum_citation_count,
                                     double minimum_strong_citation_score);
/* The "score={c.Score:F4}" text of BuildGroundedPrompt (ChatOrchestrationService.cs:85) for a score that
 * already went through Math.Round(x, 4); returns bytes written (no terminator) or -1. */
int32_t orrh_format_score_f4(double rounded_score, char *out, int32_t out_cap);

/* ---- store + service mirrors ------------------------------------------------
 * orrh_store mirrors the parts of InMemoryIngestionStore the path touches
 * (InMemoryIngestionStore.cs:11-25,50-76); orrh_service mirrors
 * RecallSearchService.SearchAsync (RecallSearchService.cs:20-57) with lines :26-37
 * replaced by orr_search_batch, and returns the /api/recall/search response body
 * (RecallDtos.cs:3-16, camelCase) as JSON.  Enumeration order of documents is
 * insertion order (the reference's ConcurrentDictionary order is unspecified).
 * Status codes are those of omnirecall_hip.h; orrh_last_error() gives the text. */
typedef struct orrh_store orrh_store;
typedef struct orrh_service orrh_service;

const char *orrh_last_error(void);
orrh_store *orrh_store_create(void);
void        orrh_store_destroy(orrh_store *s);
/* UpsertDocumentAsync (:11-15) */
int orrh_store_upsert_document(orrh_store *s, const char *id, const char *file_name, int64_t created_ticks);
/* UpsertChunksAsync (:17-25): replaces the chunk list of chunks[0]'s document, ordered by
 * chunk_index.  emb holds the chunks' vectors back to back; emb_len[i] = 0 means null. */
int orrh_store_upsert_chunks(orrh_store *s, const char *document_id, int32_t n, const char *const *chunk_ids,
                             const int32_t *chunk_index, const char *const *contents, const float *emb,
                             const int32_t *emb_len, const int64_t *created_ticks);
/* DeleteDocumentAsync (:50-55) */
int orrh_store_delete_document(orrh_store *s, const char *document_id);
int64_t orrh_store_chunk_count(const orrh_store *s);

/* ---- the reference's durable corpus format -------------------------------------
 * Cosmos items as System.Text.Json writes them under JsonNamingPolicy.CamelCase
 * (CosmosIngestionRecords.cs:5-30, CosmosIngestionStore.cs:34-40):
 *   {"id","PartitionKey","type":"chunk","documentId","chunkIndex","content","embedding":[..]|null,"createdAtUtc"}
 *   {"id","PartitionKey","type":"document","fileName",...,"createdAtUtc"}
 * Input: a JSON array of items, or items one after another (JSON lines), or query pages
 * {"Documents":[...],"_count":n} back to back; system properties (_rid, _etag, _ts ...) and unknown ones
 * are ignored, names are case-sensitive, the last duplicate property wins.  "embedding" numbers are read
 * straight to binary32, correctly rounded; "createdAtUtc" is ISO 8601 (Z or +-hh:mm, up to 100 ns).
 * Documents are upserted (UpsertDocumentAsync); the chunk items of one documentId become that document's
 * chunk list (UpsertChunksAsync: ordered by chunkIndex, replacing what the store held; a repeated item id
 * replaces the earlier item).  The input is parsed and validated completely before the store changes:
 * ORR_EINVAL leaves it untouched.  Export writes the store in the same form (documents, then chunks in
 * enumeration order; floats in their shortest round-trip text; release with orrh_free): import(export(s))
 * rebuilds an identical store. */
int orrh_store_import_cosmos_json(orrh_store *s, const uint8_t *json, int64_t len, int64_t *out_documents,
                                  int64_t *out_chunks);
int orrh_store_export_cosmos_json(orrh_store *s, uint8_t **out_json, int64_t *out_len);

/* candidate_limit = GetRecentChunksAsync(maxCount) (300 in the reference). */
orrh_service *orrh_service_create(orrh_store *s, int32_t device, int64_t candidate_limit);
void          orrh_service_destroy(orrh_service *svc);
/* Index maintenance: documents uploaded after the last build whose chunks are
 * all strictly newer than everything indexed become a small DELTA shard placed in front of the
 * existing ones (candidate order is CreatedAt-descending); searches then run per shard and are
 * merged exactly like a multi-GPU search (orr_search_shard + orr_merge_candidates).  A deleted document
 * and a document whose chunk list was replaced (InMemoryIngestionStore.cs:17-25, 50-55) lose their rows in
 * place (orr_index_delete_rows: no reseal, row ids of the others unchanged); a replaced list then counts
 * as new.  With eight shards in place the next upload MERGES the delta shards (all but the oldest) and the
 * new chunks into one shard, the oldest -- the large one of a corpus that grows by uploads -- staying on the
 * device untouched (while the deltas together are smaller than it).  Older timestamps, a different embedding
 * dimension or more than a quarter of a shard deleted (after compaction) trigger a full rebuild.  Counters
 * for tests/metrics: */
void orrh_service_stats(orrh_service *svc, int32_t *n_shards, int64_t *full_rebuilds, int64_t *delta_builds);
int64_t orrh_service_tombstoned_rows(orrh_service *svc);    /* rows dropped in place so far */
int64_t orrh_service_delta_merges(orrh_service *svc);       /* times the delta shards were merged into one (above) */
int64_t orrh_service_compactions(orrh_service *svc);        /* shards compacted in place (orr_index_compact) instead of rebuilt, when more than a
                                                               quarter of a shard's rows had been dropped */
/* SearchAsync(query, topK) with the query embedding supplied by the caller (the
 * IEmbeddingClient result; qdim 0 = empty vector) and a frozen clock.  *out_json is
 * malloc'd; release it with orrh_free.  A blank query is ORR_EINVAL "Query is required." */
int  orrh_service_search_json(orrh_service *svc, const char *query_utf8, const float *qvec, int32_t qdim,
                              int32_t topk, int64_t now_ticks, char **out_json, int64_t *out_len);
void orrh_free(void *p);

/* ---- request micro-batcher -----------------------------------------
 * The reference serves one SearchAsync per request thread (scoped service, Program.cs:59);
 * the GPU's batched paths need tens to hundreds of queries per call.  The batcher
 * coalesces concurrent single-query calls into one orr_search_batch: a worker thread takes
 * the first waiting request, keeps collecting compatible ones (same dim and
 * candidate_limit) until max_batch are there or max_wait_us have passed, runs the batch
 * with the largest topk among them (a smaller topk is a prefix of a larger one) at ONE
 * clock -- the latest now_ticks of its requests: callers pass DateTime.UtcNow.Ticks, which
 * never repeats, and a batch's requests are at most max_wait_us apart (rule F3: one frozen
 * clock per search) -- and wakes the callers.  index is the orr_index* the requests are
 * answered from (not owned). */
typedef struct orrh_batcher orrh_batcher;
orrh_batcher *orrh_batcher_create(void *index, int32_t max_batch, int32_t max_wait_us);
void          orrh_batcher_destroy(orrh_batcher *b);
/* Blocking, thread-safe.  Same arguments as one query of orr_search_batch; out_rows and
 * out_scores hold max(1,topk) entries. */
int orrh_batcher_search(orrh_batcher *b, int32_t dim, const float *q, const uint8_t *terms_utf8,
                        const uint32_t *term_off, int32_t n_terms, int64_t now_ticks, int32_t topk,
                        int64_t candidate_limit, int64_t *out_rows, double *out_scores, int32_t *out_count);
/* The same; *out_batch_now (may be NULL) receives the clock the request's batch was answered at. */
int orrh_batcher_search_at(orrh_batcher *b, int32_t dim, const float *q, const uint8_t *terms_utf8,
                           const uint32_t *term_off, int32_t n_terms, int64_t now_ticks, int32_t topk,
                           int64_t candidate_limit, int64_t *out_rows, double *out_scores, int32_t *out_count,
                           int64_t *out_batch_now);
/* Batches dispatched and requests served so far (for tests and metrics). */
void orrh_batcher_stats(orrh_batcher *b, int64_t *batches, int64_t *requests, int32_t *largest_batch);

#ifdef __cplusplus
}
#endif
#endif
