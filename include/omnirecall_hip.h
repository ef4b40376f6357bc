/*
 * omnirecall_hip.h -- C ABI of libomnirecall_hip.so, the MI355X (gfx950) scorer
 * behind the reference's IRecallSearchService seam.
 *
 * The reference has no FFI of its own: its only seam for this path is the DI
 * interface
 *     IRecallSearchService.SearchAsync(string query, int topK, CancellationToken)
 *         src/OmniRecall.Api/Services/RecallSearchService.cs:6-9   (registered Program.cs:59)
 * A drop-in GpuRecallSearchService keeps RecallSearchService.cs:22-25 (validate,
 * embed) and :39-56 (file names, snippet, Math.Round, DTO) in C#, and replaces
 * :26-37 -- GetRecentChunksAsync(300) + Select(ScoreChunk) + OrderByDescending /
 * ThenByDescending / Take -- with ONE call into this library.  INTEGRATION.md
 * shows the P/Invoke declarations that bind exactly these symbols.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++ / torch types, no exceptions.
 *   - every function returning int returns ORR_OK (0) or a negative orr_status;
 *     orr_last_error() gives the thread-local message.  Numeric guard cases of
 *     the reference (empty vector, dimension mismatch, zero norm;
 *     RecallSearchService.cs:71-72,84-85) are NOT errors: they score cosine 0.
 *   - the caller owns every buffer it passes; the library reads inputs only for
 *     the duration of the call (append copies) and owns device memory behind
 *     the opaque handle.  Strings cross as UTF-8 bytes with explicit offsets;
 *     nothing relies on NUL termination; nothing is freed across the ABI.
 *   - pointers marked "host or device" may be either; the library copies with
 *     hipMemcpyDefault.  The library works on its own non-blocking HIP streams: device-resident
 *     inputs must be COMPLETE (their producing stream synchronised, or an event waited for)
 *     before the call, and device-resident outputs are complete when the call returns.
 *   - threads: searches on a sealed index may be issued from any thread and run CONCURRENTLY on one
 *     handle (RecallSearchService is scoped per request, Program.cs:59; the store behind it is
 *     lock-free, InMemoryIngestionStore.cs:8-9): each takes a search lane of the index -- its own
 *     workspaces, or those of an internal view created on demand, up to the "max_lanes" option
 *     (default 4; corpus and shadows are shared, a lane costs its workspaces); further callers wait
 *     for a lane.  append / seal / delete / options wait until no search is in flight.
 */
#ifndef OMNIRECALL_HIP_H
#define OMNIRECALL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORR_ABI_VERSION 1

typedef enum orr_status {
    ORR_OK      = 0,
    ORR_EINVAL  = -1,   /* bad argument -> ArgumentException (RecallSearchService.cs:22-23)   */
    ORR_ENOMEM  = -2,   /* host or device allocation failed                                   */
    ORR_EDEVICE = -3,   /* HIP runtime / kernel failure, or no gfx950 device                   */
    ORR_ECOMM   = -4,   /* shard exchange inconsistent (merge input malformed)                 */
    ORR_EDIM    = -5,   /* appended embedding dimension differs from the index dimension       */
    ORR_ESTATE  = -6    /* call not valid in this state (append after seal, search before)     */
} orr_status;

typedef struct orr_index orr_index;      /* opaque: one corpus shard resident on one GPU */

typedef struct orr_config {
    int32_t struct_size;      /* sizeof(orr_config), for forward compatibility                  */
    int32_t device;           /* HIP device ordinal                                             */
    int32_t dim;              /* embedding dimension D of every stored vector; 0 = corpus
                                 without embeddings (NoOpEmbeddingClient.cs:7): cosine is 0     */
    int32_t flags;            /* reserved, 0                                                    */
    int64_t capacity_rows;    /* rows to reserve up front; 0 = grow on demand                   */
    int64_t row_base;         /* position of this shard's first row in the GLOBAL
                                 CreatedAt-descending candidate order (0 on a single GPU)       */
} orr_config;

/* One ranked-candidate record as exchanged between shards (56 bytes).  A shard
 * emits kprime of these per query plus one trailer (see orr_search_shard). */
typedef struct orr_candidate {
    double  approx_score;   /* device-side fused score used for selection                      */
    double  dot;            /* sum_i (double)fl32(q_i * e_i), reference order (…cs:77-82)      */
    double  norm_b;         /* sum_i (double)fl32(e_i * e_i); 0 for rows without an embedding  */
    int64_t created_ticks;  /* CosmosChunkRecord.CreatedAtUtc.Ticks                            */
    int64_t row_id;         /* caller's id of the row (see orr_index_append); -1 = no record   */
    int64_t order_key;      /* global candidate position (row_base + position in the shard)    */
    int32_t matches;        /* query terms found in the content (…cs:111)                      */
    int32_t flags;          /* ORR_CAND_* bits                                                 */
    /* trailer record (index kprime of each query): approx_score = score of the
     * worst kept candidate, or -inf when the shard kept every participating row;
     * row_id = -1; order_key = rows that took part on this shard; matches =
     * number of valid records in front of it; flags = ORR_CAND_TRAILER.       */
} orr_candidate;

#define ORR_CAND_TRAILER   1
#define ORR_CAND_DOT_EXACT 2   /* `dot` is already the reference-order fp64 sum */
#define ORR_CAND_OVERFLOW  4   /* trailer only: the shard's candidate buffer overflowed; repeat unfused */
#define ORR_CAND_TWO_STAGE 8   /* trailer only: norm_b holds L; every row not offered has an exact score < L */
#define ORR_CAND_DEAD      16  /* the row was deleted (orr_index_delete_rows): the host finish drops the record */

/* Per-kernel timing collected with HIP events on the index's own stream. */
typedef struct orr_kernel_stat {
    char    name[48];
    int64_t launches;
    double  total_ms;        /* sum of hipEventElapsedTime over those launches                 */
    double  algo_bytes;      /* algorithmic bytes summed over those launches       */
} orr_kernel_stat;

/* Counters of one index since the last reset (orr_index_search_stats): how often the cheap passes had to be
 * repeated, and what the screening pass of the two-stage search kept.  128 bytes. */
typedef struct orr_search_stats {
    int64_t searches;            /* orr_search_batch calls                                                    */
    int64_t queries;             /* queries in them                                                           */
    int64_t passes;              /* device passes run for them (= searches when nothing had to be repeated)   */
    int64_t requeried;           /* queries that went through another pass (summed over repeats)              */
    int64_t overflowed_queries;  /* queries whose survivors did not fit their buffer in some pass             */
    int64_t buffer_growths;      /* times the survivors' buffers were enlarged from the measured counts       */
    int64_t exact_pass_queries;  /* queries that ended in the reference-arithmetic pass over every row        */
    int64_t survivors_total;     /* (query,row) pairs kept by the screening pass, summed over queries         */
    int64_t survivor_samples;    /* queries counted in survivors_total                                        */
    int64_t survivors_max;       /* largest count of one query                                                */
    int64_t survivor_capacity;   /* buffer entries per query the index uses now                               */
    int64_t vocab_tokens;        /* distinct whitespace-free tokens of the shard's contents (the keyword index)  */
    int64_t kw_hits_total;       /* (distinct query term, vocabulary token containing it) pairs, summed over passes */
    int64_t kw_passes;           /* passes that had query terms                                                */
    int64_t pass_mode;           /* what the LAST device pass ran: 0 no two-stage pass (exact kernel, unfused batched pass,
                                    large-k sort); 1 two-stage on the int8 shadow; 2 two-stage on the bf16 shadow; 3 two-stage
                                    WITHOUT a shadow (fp32 rows converted inside the kernel: "two_stage" = 2, or the shadow
                                    did not fit in device memory and "two_stage" = 1 fell back)                          */
    int64_t reserved[1];         /* orr_cluster_search_stats: record exchanges done by RCCL all-gather ("exchange" = 1)      */
} orr_search_stats;

int         orr_abi_version(void);
int         orr_device_count(void);                 /* gfx950 devices visible; 0 if none        */
const char *orr_last_error(void);                   /* thread-local, never NULL                 */

/* ---- corpus shard -------------------------------------------------------
 * Replaces the data side of InMemoryIngestionStore.GetRecentChunksAsync
 * (InMemoryIngestionStore.cs:57-65): rows are CosmosChunkRecord projections
 * (Data/Models/CosmosIngestionRecords.cs:19-30). */
int  orr_index_create(const orr_config *cfg, orr_index **out);
void orr_index_destroy(orr_index *idx);

/* Appends n rows in the store's enumeration order (copies; host or device
 * pointers).
 *   dim            == cfg.dim with emb = [n][dim] row-major fp32, or 0 with
 *                  emb = NULL for rows whose Embedding is null/empty.  Any other
 *                  dim is ORR_EDIM (a mixed-dimension corpus is not supported;
 *                  in the reference such rows always score cosine 0 unless the
 *                  query has that same odd dimension).
 *   created_ticks  [n]    DateTime.Ticks of CreatedAtUtc.
 *   content_lower  UTF-8 of Content.ToLowerInvariant() (RecallSearchService.cs:110
 *                  is hoisted to ingest; the C# shim calls ToLowerInvariant itself).
 *   content_off    [n+1]  byte offsets into content_lower.
 *   row_ids        [n] ids returned by searches, or NULL for
 *                  row_base + (rows appended so far) + i.                        */
int orr_index_append(orr_index *idx, int64_t n, int32_t dim, const float *emb,
                     const int64_t *created_ticks, const uint8_t *content_lower,
                     const uint64_t *content_off, const int64_t *row_ids);

/* Puts rows into candidate order -- stable CreatedAt-descending, i.e. what
 * OrderByDescending(c => c.CreatedAtUtc) yields (InMemoryIngestionStore.cs:61) --
 * and precomputes the exact row norms.  Required before searching. */
int     orr_index_seal(orr_index *idx);
/* Moves the shard within the global candidate order (a newer shard was put in front of it):
 * only order keys and the clipping of candidate_limit depend on it. */
int     orr_index_set_row_base(orr_index *idx, int64_t row_base);
int64_t orr_index_rows(const orr_index *idx);
int32_t orr_index_dim(const orr_index *idx);

/* ---- search -------------------------------------------------------------
 * One batch of B queries against a sealed single-GPU index; replaces
 * RecallSearchService.cs:26-37 for each query.
 *   dim, q          query vectors [B][dim] (host or device); dim 0 = empty
 *                   vector (EmbeddingResult.Vector = []), q may be NULL.
 *   terms_utf8, term_off, query_term_off
 *                   the queryTerms of RecallSearchService.cs:95-108, already split,
 *                   lowercased, de-duplicated and stop-word filtered by the host:
 *                   term t is terms_utf8[term_off[t] .. term_off[t+1]); query b owns
 *                   terms query_term_off[b] .. query_term_off[b+1).  A query with no
 *                   terms has keyword score 0 (:100-101).
 *                   These three arrays are HOST memory.
 *   now_ticks       the frozen DateTime.UtcNow.Ticks for :117.
 *   topk            Take(Math.Max(1, topK)) (:36).
 *   candidate_limit GetRecentChunksAsync(maxCount) (:26): 300 reproduces the
 *                   reference, >= rows scores the whole corpus.
 *   out_rows, out_scores  [B][max(1,topk)]: row ids and UNROUNDED fused scores in
 *                   rank order (score desc, CreatedAt desc, candidate order).
 *   out_counts      [B]: citations actually produced (< topk on a small corpus). */
int orr_search_batch(orr_index *idx, int32_t B, int32_t dim, const float *q,
                     const uint8_t *terms_utf8, const uint32_t *term_off,
                     const uint32_t *query_term_off, int64_t now_ticks, int32_t topk,
                     int64_t candidate_limit, int64_t *out_rows, double *out_scores,
                     int32_t *out_counts);

/* Row-sharded corpus, step 1 (runs on every shard's GPU): the shard's best
 * kprime candidates per query, selected on (score desc, global candidate
 * position asc).  out: [B][kprime+1] records (host or device), the last one of
 * each query being the trailer.  candidate_limit is GLOBAL; the shard clips it
 * with its row_base. */
int orr_search_shard(orr_index *idx, int32_t B, int32_t dim, const float *q,
                     const uint8_t *terms_utf8, const uint32_t *term_off,
                     const uint32_t *query_term_off, int64_t now_ticks, int32_t kprime,
                     int64_t candidate_limit, orr_candidate *out);

/* The same with the two per-call choices as ARGUMENTS instead of sticky index options ("shard_topk", "shard_pass", which
 * orr_search_shard reads): topk = the caller's k when > 0 (the two-stage floor then comes from the k-th best score of the
 * sampled prefix, not the k'-th: fewer survivors; valid across shards, the global k-th best is at least every shard's);
 * pass = 0 the library's choice, 1 the unfused batched pass, 2 the reference-arithmetic pass over every row (the repeat of
 * queries orr_merge_candidates could not certify).  Passes 1 and 2 keep one number per (query,row): they run over slices
 * of the batch so that their workspace stays below 4 GiB whatever B. */
int orr_search_shard_ex(orr_index *idx, int32_t B, int32_t dim, const float *q,
                        const uint8_t *terms_utf8, const uint32_t *term_off,
                        const uint32_t *query_term_off, int64_t now_ticks, int32_t kprime,
                        int64_t candidate_limit, int32_t topk, int32_t pass, orr_candidate *out);

/* Row-sharded corpus, step 2 (host only, no GPU needed): merges the gathered
 * records of n_shards shards ([n_shards][B][kprime+1], host memory), rescoring
 * every candidate in the reference's exact arithmetic and ranking with the exact
 * key.  index_dim is the shards' embedding dimension (cosine applies only when
 * dim == index_dim > 0, RecallSearchService.cs:71).  *out_uncertified (may be NULL) receives the number of queries whose
 * top-k could not be certified against the shards' cut-off scores -- the caller
 * should repeat both steps with a larger kprime for those. */
int orr_merge_candidates(int32_t n_shards, int32_t B, int32_t kprime, const orr_candidate *all,
                         int32_t index_dim, int32_t dim, const float *q_host,
                         const uint32_t *query_term_off, int64_t now_ticks, int32_t topk,
                         int64_t *out_rows, double *out_scores, int32_t *out_counts,
                         int32_t *out_uncertified);

/* The same, also telling WHICH queries were certified: out_certified[B] (may be NULL), 1 = the query's top-k is final.  The
 * caller repeats only the others (as a compacted sub-batch, through orr_search_shard_ex with pass = 2, then a larger kprime):
 * every rank of a multi-process job holds identical gathered bytes, so all ranks pick the same sub-batch without a collective. */
int orr_merge_candidates_ex(int32_t n_shards, int32_t B, int32_t kprime, const orr_candidate *all,
                            int32_t index_dim, int32_t dim, const float *q_host,
                            const uint32_t *query_term_off, int64_t now_ticks, int32_t topk,
                            int64_t *out_rows, double *out_scores, int32_t *out_counts,
                            int32_t *out_uncertified, uint8_t *out_certified);

/* ---- shard file ---------------------------------------------
 * A sealed shard as one binary file (embeddings, exact norms, timestamps, row ids and the
 * token index), so that a corpus does not have to be re-ingested and re-sealed per run.
 * The reference's only durable form is Cosmos JSON (CosmosIngestionRecords.cs:19-30).
 * load: cfg->device and cfg->row_base are taken from cfg; dim comes from the file
 * (cfg->dim must be 0 or equal).  The loaded index is sealed. */
int orr_index_save(orr_index *idx, const char *path);
int orr_index_load(const orr_config *cfg, const char *path, orr_index **out);

/* ---- a second search lane ---------------------------------------------------
 * orr_index_view: another handle over the same SEALED shard with its own streams and workspaces.  It
 * borrows the corpus (and the bf16 shadow, built first if the two-stage pass is on): nothing is copied.
 * Searches on the index and on its views may run concurrently from different threads, which lets the
 * keyword chain, the ranking pass and the host finish of one batch overlap the screening pass of another
 * (the request path of Program.cs:59 is concurrent by nature).  A view must be destroyed before its
 * parent; it cannot be appended to, sealed again or saved. */
int orr_index_view(orr_index *parent, orr_index **view);

/* ---- deletes without a reseal -------------------------------------------------
 * Replaces InMemoryIngestionStore.DeleteDocumentAsync (InMemoryIngestionStore.cs:50-55) and the
 * "replace the chunk list" half of UpsertChunksAsync (:17-25) on a SEALED shard: the rows with the given
 * ids (the ids of orr_index_append) stop taking part in every later search, exactly as if the shard had
 * been rebuilt without them -- they are not ranked, and they do not count towards candidate_limit.  The
 * rows keep their positions (order_key of the others is unchanged, so is every row id), nothing is moved
 * in HBM: the cost is one small upload.  On the device a deleted row's norm and timestamp are overwritten
 * (its score drops to the keyword part, <= 0.2), its records are flagged ORR_CAND_DEAD and the host finish
 * (orr_search_batch, orr_merge_candidates) drops them; the certificate logic is unchanged, so results stay
 * exact.  Unknown and already deleted ids are skipped; *out_deleted (may be NULL) = rows newly deleted.
 * Exclusive like append/seal: no search may be in flight on the index or its views.  Views see the
 * deletes of their parent.  ORR_ESTATE once more than a quarter of the shard is deleted: rebuild it.
 * orr_index_live_rows = rows - deleted rows.  A shard behind others in the global order is told how many
 * deleted rows lie in front of it with the "dead_rows_before" option, so that candidate_limit keeps
 * counting live rows only.  orr_index_save / orr_index_load keep the deleted set. */
int     orr_index_delete_rows(orr_index *idx, int64_t n, const int64_t *row_ids, int64_t *out_deleted);
int64_t orr_index_live_rows(const orr_index *idx);

/* ---- compaction ------------------------------------------------------------------
 * Rebuilds a sealed shard IN PLACE without its deleted rows -- the other half of "replace the chunk list"
 * (InMemoryIngestionStore.cs:17-25, 50-55), where the reference simply drops the old list: embeddings move up
 * chunk by chunk through a 256 MiB bounce buffer (no second copy of the shard), norms / timestamps / ids are
 * gathered, every posting list of the token index loses the deleted positions and is renumbered, the shadows are
 * dropped and rebuilt at the next search that wants them.  Row ids are kept; positions (order keys) close up, so a
 * shard BEHIND this one in a global order moves up by *out_removed (may be NULL) rows: give it its new row_base
 * (orr_index_set_row_base) and "dead_rows_before" -- orr_cluster_compact does both for a cluster.  Lifts the
 * quarter-of-the-shard limit of orr_index_delete_rows.  Exclusive like delete; ORR_ESTATE while views made with
 * orr_index_view are alive (they borrow the arrays that move).  A failure half way (ORR_EDEVICE / ORR_ENOMEM)
 * leaves the shard unusable: rebuild it. */
int orr_index_compact(orr_index *idx, int64_t *out_removed);

/* ---- tuning knobs ----------------------------------------------------------
 * Integer options of one index; unknown names are ORR_EINVAL.
 *   "dead_rows_before"  deleted rows in the shards in front of this one (default 0), see above.
 *   "max_lanes"      1..16 (default 4): searches that may run at once on this handle (see "threads" above); lanes that
 *                    exist are kept.
 *   "kw_hits_cap"    entries of the keyword chain's hit list, one per (distinct query term, vocabulary token containing it)
 *                    (default 16M = 384 MB at most).  A batch that needs more grows the list to the measured count and
 *                    repeats its pass; the option exists to pre-size it (or, in tests, to force that path).
 *   "shard_pass"     0/1/2 (default 0): which pass orr_search_shard runs -- 0 the library's choice, 1 the unfused
 *                    batched pass, 2 the reference-arithmetic pass over every row.  The caller of
 *                    orr_merge_candidates sets 2 for the repeat of a batch some query of which could not be certified.
 *   "shard_topk"     the caller's topK for orr_search_shard (default 0: unknown).  When set, the two-stage pass takes its
 *                    floor from the k-th best score of the sampled prefix instead of the k'-th: fewer survivors.
 *   "fuse_epilogue"  0/1 (default 0): batches > 64 queries over >= 196,608 rows score and filter
 *                    inside the GEMM epilogue instead of writing the dots to HBM.
 *   "two_stage"      0/1/2 (default 1): searches over >= 196,608 rows screen ALL rows with ONE low-precision
 *                    product, keep every (query,row) pair that could reach a lower bound of the query's k-th
 *                    best score, and re-score those in the reference arithmetic on the device.
 *                    1: the product reads a shadow copy of the embeddings, built at the first such search (or
 *                    now if the index is sealed and the option is set explicitly): int8 with one scale per row
 *                    and a per-pair error bound when dim % 128 == 0 (+25 % HBM; a stream for 1..4 queries, an
 *                    int8 MFMA GEMM for more), bf16 otherwise (+50 % HBM, bound 2^-7 |q||e|); silently falls
 *                    back to 2 when the shadow does not fit.  2: no shadow: 5+ queries convert the fp32 rows to
 *                    bf16 inside the kernel, fewer run the exact kernel.  0: exact kernel (1..4 queries) /
 *                    streaming or split-bf16 MFMA pass over all rows. */
int orr_index_set_option(orr_index *idx, const char *name, int64_t value);

/* Diagnostic: out[B][orr_index_rows] = the screening dots of the two-stage pass (fp32, host or device
 * memory), i.e. sum_k bf16(q_k) bf16(e_k) accumulated in fp32 on the matrix cores.  Lets a test check the
 * bound the pass relies on.  Needs the bf16 shadow (ORR_ENOMEM when it does not fit); dim % 64 == 0. */
int orr_index_screen_dots(orr_index *idx, int32_t B, int32_t dim, const float *q, float *out);

/* Diagnostic: the RAW int32 accumulators of the int8 screening GEMM (K2j), out_dots[B][orr_index_rows] (host or device), as
 * one FORM of the kernel computes them -- 0: eight-wave 32x32x32 tile (what batches of up to 64 queries, the sampled prefix
 * and the bf16 shadow run), 1: four-wave 32x32x32 tile (65..128 queries, very large shards), 2: four-wave 16x16x64 tile
 * (129+ queries: the dominant kernel of config C3/C5) -- with the same K loop, operand rings, request streams and persistent
 * walk of the output tiles as the fused launches of a search; only the scoring epilogue is replaced by a store.  nt_rows:
 * rows requested non-temporal (what a search does for batches of one query tile).  The integer work the screen does in
 * place of RecallSearchService.cs:77-82 is checkable bit for bit this way: out_iq[B][dim] / out_ie[rows][dim] (either may be
 * NULL) receive the quantised int8 images the product multiplies (queries: one level; rows: the shard's int8 shadow, untiled),
 * and out_dots must equal out_iq x out_ie^T exactly.  dim % 128 == 0; forms 1 and 2 need dim >= 448.  ORR_ENOMEM without room
 * for the shadow. */
int orr_index_screen_i8_dots(orr_index *idx, int32_t B, int32_t dim, const float *q, int32_t form, int32_t nt_rows,
                             int32_t *out_dots, int8_t *out_iq, int8_t *out_ie);

/* ---- measurement ---------------------------------------------------------*/
/* enabled: 0 off; 1 an event pair around every kernel; 2 only around the one launch per search that streams every row
 * (each pair costs a few microseconds of stream time, which a one-query search notices).  Also resets the counters. */
int orr_index_set_profiling(orr_index *idx, int32_t enabled);
int orr_index_kernel_stats(orr_index *idx, orr_kernel_stat *out, int32_t cap);  /* returns count */

int orr_index_search_stats(orr_index *idx, orr_search_stats *out, int32_t reset);  /* out may be NULL (reset only) */

/* ---- several GPUs behind one handle, in ONE process ------------------------------
 * The reference host is a single process with a singleton store (Program.cs:59,
 * IngestionServiceCollectionExtensions.cs:22-23); north_star keeps that host in C#.  An orr_cluster owns one
 * shard per entry of `devices` (the same ordinal may appear twice: two shards on one GPU) and answers
 * orr_cluster_search_batch exactly as one orr_index over all the rows would: every shard scores the whole batch on
 * its own device at once (one host thread per shard), the per-shard [B][k'+1] candidate records come back through
 * pinned host memory, the host merges and certifies them as orr_merge_candidates does and repeats only the queries
 * that could not be certified.  (Between PROCESSES the same records travel by one RCCL all-gather: sharded.py.)
 *   create            dim as in orr_config; capacity_rows_per_shard reserves device memory per shard (0: grow).
 *   shard(i)          borrowed handle for orr_index_append / orr_index_set_option / orr_index_delete_rows.  Shard i
 *                     must receive rows that are all at least as new as every row of shard i + 1 (partition the
 *                     store's rows by CreatedAtUtc, newest first; orr_cluster_seal checks it): the global candidate
 *                     order (InMemoryIngestionStore.cs:61) is then shard 0's rows, shard 1's, ...
 *   seal              seals every shard (concurrently) and places them in the global order (row_base).
 *   search_batch      arguments as orr_search_batch; q must be HOST memory (each device uploads it).  Row ids are
 *                     the ids given at append (default: position in the shard + its row_base at append time, i.e.
 *                     pass explicit row_ids when appending to a cluster).
 * Thread-safe: cluster searches from different threads run side by side (each shard half on a search lane of its shard,
 * host halves on persistent pool threads); seal and destroy are exclusive. */
typedef struct orr_cluster orr_cluster;
int        orr_cluster_create(const int32_t *devices, int32_t n_shards, int32_t dim, int64_t capacity_rows_per_shard,
                              orr_cluster **out);
void       orr_cluster_destroy(orr_cluster *c);
int32_t    orr_cluster_shards(const orr_cluster *c);
orr_index *orr_cluster_shard(orr_cluster *c, int32_t i);
int        orr_cluster_seal(orr_cluster *c);
int64_t    orr_cluster_rows(const orr_cluster *c);
int        orr_cluster_search_batch(orr_cluster *c, int32_t B, int32_t dim, const float *q_host,
                                    const uint8_t *terms_utf8, const uint32_t *term_off,
                                    const uint32_t *query_term_off, int64_t now_ticks, int32_t topk,
                                    int64_t candidate_limit, int64_t *out_rows, double *out_scores,
                                    int32_t *out_counts);
int        orr_cluster_search_stats(orr_cluster *c, orr_search_stats *out, int32_t reset);
/* Integer options of a cluster; unknown names are ORR_EINVAL.
 *   "exchange"   0 (default): the per-shard [B][k'+1] candidate records come back through pinned host memory (every record is
 *                wanted in ONE address space, so nothing needs a collective); 1: every shard writes its records into a
 *                device buffer and ONE RCCL all-gather over xGMI (ncclAllGather, one communicator per shard device, grouped
 *                from one thread) brings all shards' records to every device; the merge reads device 0's gathered copy --
 *                the literal exchange of BASELINE.json's north_star in the form a single-process host can load.  librccl.so is
 *                bound at run time (dlopen): ORR_ECOMM when it cannot be loaded, ORR_EINVAL when two shards share a device
 *                (a communicator needs distinct devices).  A failure of a later collective switches the cluster back to 0
 *                and returns ORR_ECOMM for that search. */
int        orr_cluster_set_option(orr_cluster *c, const char *name, int64_t value);
/* orr_index_compact on every shard (concurrently), then the shards are placed in the global order again. */
int        orr_cluster_compact(orr_cluster *c, int64_t *out_removed);

#ifdef __cplusplus
}
#endif
#endif /* OMNIRECALL_HIP_H */
