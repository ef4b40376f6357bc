// orr_kernels.h -- launch wrappers of the gfx950 kernels behind libomnirecall_hip.so.
// All launches are asynchronous on the given stream; none allocates or synchronises.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/omnirecall_hip.h"

#include <atomic>

namespace orr {

// Per-DEVICE launch state (orr_cluster drives several devices from one process, one host thread per shard): the
// max-dynamic-LDS attribute of a kernel and a device's CU count are established on every device they are used on, not on
// whichever device happened to be current at the first call of the process.
inline int current_device_ordinal()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
    return d;
}
template <auto KERNEL>
hipError_t ensure_max_dynamic_lds(int bytes)
{
    static std::atomic<uint64_t> done[4] = {};            // one bit per device ordinal (256 devices)
    const int dev = current_device_ordinal() & 255;
    if ((done[dev >> 6].load(std::memory_order_acquire) >> (dev & 63)) & 1ull) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done[dev >> 6].fetch_or(1ull << (dev & 63), std::memory_order_release);
    return e;
}
inline int device_cu_count()
{
    static std::atomic<int> cus[256] = {};
    const int dev = current_device_ordinal() & 255;
    int n = cus[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = -1;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n > 0 ? n : 0;
}

constexpr int kSelWidth = 64;        // entries a wave keeps while selecting (one per lane)
constexpr int kSelSegRows = 4096;    // rows one workgroup of fuse_select scans
constexpr int kMaxExactQ = 8;        // queries one launch of the exact dot kernel carries
constexpr int kMaxScanTerms = 64;    // query terms one launch of the keyword scan carries
constexpr int kMaxGemvScreenQ = 8;   // queries one launch of the streaming screen (K2g) carries
constexpr int kMaxI8ScreenQ = 4;     // ... of its int8 form (K2i): two accumulators per query and row
constexpr int kCountPlanes = 4;      // 32-bit words per (32 queries, row) of keyword match counts: four bits per query (saturating at 15)

// One selection entry: `key` orders scores (see score_key in the .hip), `pos` is the
// row's position in the shard's candidate order.  key 0 = empty slot.
struct SelEntry {
    unsigned long long key;
    uint32_t pos;
    uint32_t pad;
};

// Per-term metadata for the keyword scan (built on the host per batch).
struct ScanTerm {
    uint32_t off;      // byte offset of the term in the term pool
    uint32_t len;      // bytes
    uint32_t prefix;   // first min(4,len) bytes, little-endian packed
    uint32_t mask;     // byte mask of those bytes
};

// K0 / K1e: out[q][r] = sum_i (double)fl32(Q[q][i] * E[r][i]) in index order (exact
// reference arithmetic, RecallSearchService.cs:77-82).  self_norm: Q is ignored and
// out[0][r] = sum_i (double)fl32(E[r][i]^2).  nq <= kMaxExactQ.
hipError_t launch_dot_exact(const float *E, int64_t n_rows, int32_t D, const float *Q, int32_t nq,
                            bool self_norm, double *out, int64_t out_stride, hipStream_t s);

// K3: matches[b][r] = number of terms of query b that occur in row r's content
// (RecallSearchService.cs:111).  n_terms <= kMaxScanTerms; q_term_off[B+1] indexes
// into terms[0..n_terms).  accumulate != 0 adds to matches instead of overwriting (a
// query with more than kMaxScanTerms terms is scanned in several launches).
// Pool layout: row r occupies [cstart[r], cstart[r]+clen[r]), cstart 16-byte aligned,
// followed by 1..16 space bytes; the pool is over-allocated by kScanPoolSlack bytes.
// Terms must not contain whitespace bytes.
constexpr size_t kScanPoolSlack = 2048;
constexpr uint64_t kRowAlign = 16;
inline uint64_t padded_row_bytes(uint64_t len) { return (len / kRowAlign + 1) * kRowAlign; }
hipError_t launch_keyword_scan(const uint8_t *pool, const uint64_t *cstart, const uint32_t *clen, int64_t n_rows,
                               const uint8_t *term_pool, const ScanTerm *terms, int32_t n_terms,
                               const uint32_t *q_term_off, int32_t B, uint16_t *matches,
                               int64_t matches_stride, int32_t accumulate, hipStream_t s);

// The same scan over the shard's VOCABULARY for every distinct query term at once (one launch).
hipError_t launch_vocab_scan(const uint8_t *vpool, const uint64_t *vstart, const uint32_t *vlen, int64_t n_tokens,
                             const uint8_t *term_pool, const ScanTerm *terms, int32_t n_terms, const uint32_t *identity,
                             uint16_t *vmatch, hipStream_t s);

// Keyword side of a batch as the scoring kernels see it: one row bitmap per DISTINCT
// query term (bit r of bitmaps[t*words_per_term + r/32] = term t occurs in row r) and
// each query's list of distinct-term indices.  bitmaps == nullptr: no query has terms.
struct KwView {
    const uint32_t *bitmaps;
    int64_t words_per_term;
    const uint32_t *q_term_idx;    // [sum of query term counts]
    const uint32_t *q_term_off;    // [B+1]
    // Where distinct term t's bitmap starts, in words from `bitmaps` (null: t * words_per_term).  A term whose only hit is a
    // vocabulary token with a STORED bitmap (orr_index: built once per sealed shard for the frequent tokens) points straight at
    // that bitmap -- the offset then leads out of the batch's own bitmaps into the token store, nothing is expanded or copied.
    const int64_t *term_word_off;
};
__host__ __device__ inline int64_t kw_term_base(const KwView &kw, uint32_t t)
{
    return kw.term_word_off ? kw.term_word_off[t] : (int64_t)t * kw.words_per_term;
}

// One (term, token) match: the token's posting run and where its 1024-posting chunks start.
struct KwHit {
    uint64_t post_begin;
    uint32_t post_len;      // postings of the token (a shard holds fewer than 2^32 rows)
    uint32_t chunk_base;
    uint32_t term;
    uint32_t token;         // vocabulary token of the hit (launch_kw_alias: a term whose ONLY hit is a token with a stored bitmap needs no expansion)
};
constexpr uint32_t kPostChunk = 1024;

// token_ids != nullptr: the scanned rows were the vocabulary tokens token_ids[0..n_tokens) (the long ones)
hipError_t launch_vocab_hits(const uint16_t *vmatch, int64_t n_tokens, int32_t n_terms, const uint32_t *token_ids,
                             const uint64_t *post_off, unsigned long long *counter, KwHit *hits, uint32_t max_hits, hipStream_t s);

// A query term for the lane-per-token matcher: its first 16 bytes as dwords with byte masks.
struct MatchTerm {
    uint32_t w[4], m[4];
    uint32_t len;       // bytes; terms longer than 16 bytes cannot occur in the tokens this kernel covers
    uint32_t pad[3];
};
struct MatchTerm8 {                   // the same with eight dwords: terms of up to 32 bytes against tokens of 17..32 bytes
    uint32_t w[8], m[8];
    uint32_t len;
    uint32_t pad[3];
};
constexpr int kMatchGroup = 32;       // terms per workgroup (grid.y)
// Every token of at most 16 bytes against every term, hits reserved as by launch_vocab_hits.
hipError_t launch_vocab_match_short(const uint8_t *vpool, const uint64_t *vstart, const uint32_t *vlen, int64_t n_tokens,
                                    const MatchTerm *terms, int32_t n_terms, const uint64_t *post_off,
                                    unsigned long long *counter, KwHit *hits, uint32_t max_hits, hipStream_t s);
// Tokens of at most 16 bytes against MANY terms: the terms (of at most 16 bytes) sorted by their length-masked first dword in
// four classes (lk[0..4] = class boundaries in keys / tidx; 1, 2, 3, 4+ bytes), every window of a token looked up.
// bloom: kVocabBloomBits bits, bit vocab_bloom_hash(key, class) set for every (class, key) of the tables.
constexpr int kVocabBloomBits = 1 << 17;
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t vocab_bloom_hash(uint32_t key, uint32_t cls)
{
    uint32_t h = (key ^ (cls * 0x9E3779B9u)) * 2654435761u;
    h ^= h >> 15;
    return h & (uint32_t)(kVocabBloomBits - 1);
}
hipError_t launch_vocab_match_lookup(const uint8_t *vpool, const uint64_t *vstart, const uint32_t *vlen, int64_t n_tokens,
                                     const MatchTerm *terms, const uint32_t *lk, const uint32_t *keys, const uint32_t *tidx,
                                     const uint32_t *bloom, const uint64_t *post_off, unsigned long long *counter, KwHit *hits,
                                     uint32_t max_hits, hipStream_t s);
// The tokens of 17..32 bytes of a list (start, length, token number) against every term (longer terms cannot occur in them),
// one lane per token.
hipError_t launch_vocab_match_mid(const uint8_t *vpool, const uint64_t *starts, const uint32_t *lens, const uint32_t *ids, int64_t n_list,
                                  const MatchTerm8 *terms, int32_t n_terms, const uint64_t *post_off, unsigned long long *counter,
                                  KwHit *hits, uint32_t max_hits, hipStream_t s);
// counter_host (optional, pinned host memory): receives *counter (hits << 32 | chunks) for the caller's statistics.
// skip_term (optional, [terms]): hits of terms flagged there are left out (their bitmaps are aliases, launch_kw_alias).
hipError_t launch_expand_hits(const KwHit *hits, const unsigned long long *counter, uint32_t max_hits,
                              const uint32_t *post_rows, uint32_t *bitmaps, int64_t words_per_term, hipStream_t s,
                              unsigned long long *counter_host = nullptr, const uint8_t *skip_term = nullptr);
// Decides per distinct term of a batch whether its bitmap can be an ALIAS of a stored token bitmap: exactly one (term, token)
// hit, and that token has a stored bitmap (tok_bm_index[token] >= 0: bitmap number in a store that begins bm_store_delta
// words from the batch's bitmaps).  term_cnt [n_terms] must be zero on entry.  Out: term_word_off [n_terms], alias [n_terms].
hipError_t launch_kw_alias(const KwHit *hits, const unsigned long long *counter, uint32_t max_hits, int32_t n_terms,
                           const int32_t *tok_bm_index, int64_t bm_store_delta, int64_t words_per_term, uint32_t *term_cnt,
                           uint32_t *term_tok, int64_t *term_word_off, uint8_t *alias, hipStream_t s);

// Per-query constants of the fused score.
struct QueryConst {
    double norm_a;        // exact sum_i (double)fl32(q_i^2); unused when use_cos == 0
    double inv_sqrt_na;   // 1/sqrt(norm_a) for the batched selection score
    double inv_n_terms;   // 1/queryTerms.Length, 0 without terms
    int32_t n_terms;      // queryTerms.Length (RecallSearchService.cs:112)
    int32_t use_cos;      // query dim == index dim, dim > 0 (and norm_a > 0 in the batched form)
};

// qc[b].norm_a / inv_sqrt_na / use_cos from norms computed on the device (launch_dot_exact, self_norm over the queries).
hipError_t launch_patch_query_norms(QueryConst *qc, const double *norm_a, int32_t B, bool batched, hipStream_t s,
                                    double *norm_host = nullptr,         // norm_host (optional, pinned): the norms for the host as well
                                    const QueryConst *qc_host = nullptr);   // qc_host (optional, pinned): the constants come from there, not from qc

// Per-row selection constants for a batch: out[r] = {1/sqrt(normB) or 0, recency * 0.1}.
hipError_t launch_row_consts(const double *norm_b, const int64_t *created, int64_t now_ticks, int64_t n_rows,
                             double2 *out, hipStream_t s);

// K4+K5a: fused score per (query,row) and per-workgroup top-64.
//   score = (cos*0.7) + (kw*0.2) + (rec*0.1)  in fp64, left to right (…cs:66)
// out_sel: [B][n_seg][kSelWidth] entries, best first.  n_seg = ceil(n_rows / kSelSegRows).
// dot (fp64, exact pass) or dotf (fp32, K2 candidate pass): exactly one is non-null when cosine applies.
// row_consts != nullptr selects the batched (reciprocal-multiply) form of the score.
// Scans segments [seg_first, seg_first+seg_count); tau != nullptr: per-query key a row must
// exceed to be considered (the k'-th best key of an already scanned prefix).
// i8.rowf != nullptr (with row_consts): dotf holds the INTEGER dots of the int8 screening GEMM over these rows
// (launch_screen_i8_dots); the key is then a LOWER bound of the score: score(dot^) minus the per-pair bound.
struct I8Prefix {
    const float4 *rowf = nullptr;    // launch_i8_rowf
    const float *qs1 = nullptr;      // [B] query scales
    const double *qerr2 = nullptr;   // [B] |q - q^|^2 of the one-level queries
};
hipError_t launch_fuse_select(const double *dot, const float *dotf, int64_t dot_stride, const double *norm_b,
                              const int64_t *created, const double2 *row_consts, KwView kw,
                              const QueryConst *qc, int64_t now_ticks, int64_t n_rows, int32_t B,
                              int32_t seg_first, int32_t seg_count, const unsigned long long *tau,
                              SelEntry *out_sel, int32_t n_seg_stride, hipStream_t s, I8Prefix i8 = I8Prefix(),
                              int floor_only = 0);          // floor_only 1: lists of per-lane maxima (enough for a floor key; FAST rows, no tau);
                                                            // 2: 16 wave maxima per list, nothing sorted (launch_select_final_sample wave_maxima = 1)

// K5b: merges the per-workgroup lists of each query and writes kprime candidate
// records plus the trailer ([B][kprime+1], see orr_candidate).
// tau_out != nullptr: write only the k'-th best key per query (sampling pass), no records.
// approx_eps goes into the trailer's `dot` field: the bound on |selection score - exact score|
// of the pass that produced the records (0 for the exact pass).
hipError_t launch_select_final(const SelEntry *sel, int32_t n_seg, int32_t B, int32_t kprime,
                               int64_t n_rows, int64_t row_base, const double *dot, const float *dotf,
                               int64_t dot_stride, const double *norm_b, const int64_t *created,
                               const int64_t *row_ids, KwView kw, int32_t dot_exact, double approx_eps,
                               unsigned long long *tau_out, const uint32_t *fused_cnt, uint32_t fused_cap,
                               const double *two_stage_L, orr_candidate *out, hipStream_t s);

// Fused epilogue of the batched candidate pass: score, compare with the query's floor key,
// append survivors to the query's buffer (cap entries; cnt keeps counting past it).
struct FusedEpilogue {
    const double2 *rowc;               // per-row {1/sqrt(normB) or 0, recency*0.1}
    const QueryConst *qc;
    KwView kw;
    const unsigned long long *tau;     // [B] floor keys
    const float4 *qf;                  // [B] fp32 pre-filter constants (launch_fused_query_consts)
    const uint32_t *count_planes;      // [kCountPlanes][ceil(B/32)][plane_stride] match counts, word k = queries 8k..8k+7 of the group, 4 bits each; or null
    int64_t plane_stride;
    // int8 screening GEMM (K2j) only, else null: per row {se_r, 0.7 (rel_err + 2^-22), rel_hat, 0}; per query s1.
    // The accumulator then holds the integer dot I: dot^ = s1 se_r I, and qf.x / qf.w carry s1 0.7/sqrt(normA)
    // and 0.7 |q - q^| / sqrt(normA): score bound = score(dot^) + rowf.y + qf.w rowf.z.
    const float4 *i8_rowf;
    const float *i8_qs1;
    uint32_t *cnt;                     // [B]
    SelEntry *buf;                     // [B][cap]
    uint32_t cap;
    unsigned long long *stamps;        // diagnostic (ORR_SCREEN_STAMPS=file): s_memtime at the phases of every output tile, else null
    int32_t count_bits;                // 4 (or 0: the same): count words as described above; 2: TWO bits per (query,row) -- word kk = queries 16 kk .. 16 kk + 15 of the group, two words per (32 queries, row) -- for batches whose queries all have at most three terms (the 16 x 16 x 64 form only)
    const float4 *qf16;                // [B] the 16 x 16 x 64 form's staged constants: qf with everything finite and .w = the batch's largest query bound term (launch_fused_query_consts)
    uint32_t *tickets;                 // [8] zeroed counters of ONE launch of the 16 x 16 x 64 screening GEMM (output tiles beyond a workgroup's first two are drawn from them), or null: static assignment
};
// K2b: S (or the fused epilogue) from three bf16 MFMA products of hi/lo splits (see orr_gemm.hip
// for the error bound); D % 64 == 0.  q_split_ws: 4*B*D bytes filled by launch_split_queries.
hipError_t launch_split_queries(const float *Q, int32_t B, int32_t D, void *q_split_ws, hipStream_t s);
hipError_t launch_gemm_dot_bf16x3(const void *q_split_ws, int32_t B, const float *E, int64_t row_first, int64_t n_rows, int32_t D,
                                  float *S, int64_t s_stride, const FusedEpilogue *epi, int32_t products, hipStream_t s);
// rows [row_first, row_end) only (row_first % 256 == 0; row_end < 0: to the end)
// bits = 2: two-bit counts (every query of the batch has at most three terms), half the words (see FusedEpilogue::count_bits)
hipError_t launch_query_count_planes(KwView kw, int32_t B, int64_t n_rows, int64_t plane_stride, uint32_t *planes, hipStream_t s,
                                     int64_t row_first = 0, int64_t row_end = -1, int32_t bits = 4);
// whether launch_screen_i8 runs its 16 x 16 x 64 form for this shape (the only form that reads two-bit count words)
bool screen_i8_uses_tile16(int32_t B, int64_t n_rows, int32_t D, int64_t plane_stride);
// qf16 (optional, [B]): the same constants made safe for a NaN-dropping test (fused_epilogue16) -- finite qx / qz, a query with
// anything non-finite turned into "every pair passes" (qx = qz = 0, floor -inf) -- with .w = the largest finite query bound
// term of the batch in EVERY entry.
// zero_a / zero_b (optional): words this launch clears as well (the pass's counters, the screening launches' tickets).
hipError_t launch_fused_query_consts(const QueryConst *qc, const unsigned long long *tau, int32_t B, float4 *qf, hipStream_t s,
                                     const float *i8_qs1 = nullptr, const double *i8_qerr2 = nullptr, float4 *qf16 = nullptr,
                                     uint32_t *zero_a = nullptr, int32_t n_a = 0, uint32_t *zero_b = nullptr, int32_t n_b = 0);
// K2c (orr_screen.hip): plain-bf16 screening GEMM (256 x 256 x 64 tiles, LDS-DMA staging) over TILED bf16
// images of the embeddings (the shard's shadow) and of the batch's queries; S or the fused epilogue as above.
size_t bf16_tiled_bytes(int64_t n_rows, int32_t D);
hipError_t launch_bf16_tiled(const float *X, int64_t n_rows, int32_t D, void *out, hipStream_t s);
hipError_t launch_screen_bf16(const void *q_tiled, int32_t B, const void *e_shadow, int64_t row_first, int64_t n_rows,
                              int32_t D, float *S, int64_t s_stride, const FusedEpilogue *epi, hipStream_t s);
// K2g: the screening pass for 1..kMaxGemvScreenQ queries as a stream over the tiled shadow (no matrix core).
hipError_t launch_screen_gemv_bf16(const void *q_hi, int32_t B, const void *e_shadow, int64_t n_rows, int32_t D,
                                   const FusedEpilogue &epi, hipStream_t s);
// K2i: the streaming screen over an INT8 shadow (per-row scale, two-level int8 queries, exact integer dots,
// a per-pair Cauchy-Schwarz bound from the quantisation norms) -- see orr_screen.hip.  D % 128 == 0.
size_t i8_tiled_bytes(int64_t n_rows, int32_t D);
hipError_t launch_i8_shadow(const float *E, const double *norm_b, int64_t n_rows, int32_t D, void *tiled, float *scale,
                            float *rel_err, float *rel_hat, hipStream_t s);
// zero (optional): n_zero words the kernel clears as well (the pass's counters: saves the call a memset).
hipError_t launch_i8_queries(const float *Q, int32_t B, int32_t D, void *q12, float *s1, double *err2, hipStream_t s,
                             double *err2_level1 = nullptr, uint32_t *zero = nullptr, int32_t n_zero = 0);
// norm_b / created / now_ticks: not null -> the kernel forms each row's scoring constants itself (epi.rowc unused).
// lower_bound with screen_gemv_i8_prefix_makes_lists(D): epi.tau must be all 0 and the keys leave as sorted lists of 64
// in epi.buf[b][0 .. ceil(n_rows / 64) * 64) (epi.cnt unused); otherwise survivors are appended to epi.buf through epi.cnt.
bool screen_gemv_i8_prefix_makes_lists(int32_t D);
hipError_t launch_screen_gemv_i8(const void *q12, const float *s1, const double *err2, int32_t B, const void *tiled,
                                 const float *scale, const float *rel_err, const float *rel_hat, const double *norm_b,
                                 const int64_t *created, int64_t now_ticks, int64_t n_rows, int32_t D,
                                 const FusedEpilogue &epi, bool lower_bound, hipStream_t s);
// K2j: the screening GEMM on the int8 shadow (v_mfma_i32_32x32x32_i8: twice the bf16 rate, half its bytes).
// Queries: ONE int8 level (q1 of launch_i8_queries, err2_l1), tiled like the rows; epi must carry i8_rowf / i8_qs1.
hipError_t launch_i8_rowf(const float *scale, const float *rel_err, const float *rel_hat, int64_t n_rows, float4 *rowf, hipStream_t s);
hipError_t launch_i8_tile_queries(const void *q1_linear, int32_t B, int32_t D, void *tiled, hipStream_t s);
// rows [row_first, n_rows) (row_first % 256 == 0): a launch per row range lets the count words of later ranges be formed
// while earlier ranges are multiplied
hipError_t launch_screen_i8(const void *q_tiled, int32_t B, const void *e_tiled, int64_t n_rows, int32_t D,
                            const FusedEpilogue &epi, hipStream_t s, int64_t row_first = 0);
// The same product over rows [0, n_rows) with the integer dots written out: S[b][r] = (float)I (the sampled prefix).
hipError_t launch_screen_i8_dots(const void *q_tiled, int32_t B, const void *e_tiled, int64_t n_rows, int32_t D, float *S,
                                 int64_t s_stride, hipStream_t s);
// Diagnostic: the raw int32 accumulators of one FORM of the kernel (0 eight-wave, 1 four-wave 32 x 32 x 32, 2 four-wave
// 16 x 16 x 64; forms 1 and 2 need D / 64 > 6), rows requested non-temporal or not; and the shadow untiled again.
hipError_t launch_screen_i8_dots_raw(const void *q_tiled, int32_t B, const void *e_tiled, int64_t n_rows, int32_t D, int32_t *S,
                                     int64_t s_stride, int32_t form, bool nt_rows, hipStream_t s, uint32_t *tickets = nullptr);
hipError_t launch_i8_untile(const void *tiled, int64_t n_rows, int32_t D, void *out_linear, hipStream_t s);
// Two-stage pass helpers (orr_gemm.hip).
hipError_t launch_rescore_buffer_exact(const float *E, int32_t D, const float *Q, int32_t B, const double *norm_b,
                                       const int64_t *created, KwView kw, const QueryConst *qc, int64_t now_ticks,
                                       const uint32_t *cnt, uint32_t cap, SelEntry *buf, double *buf_dot, hipStream_t s);
// Up to 256 queries (the caller's choice), D % 256 == 0: re-score + lists + final selection + records with exact dots in one launch (done: [B]
// zeroed counters; recs / cnt_host may be pinned host memory).
hipError_t launch_finish_survivors(const float *E, int32_t D, const float *Q, int32_t B, const double *norm_b, const int64_t *created,
                                   const int64_t *row_ids, KwView kw, const QueryConst *qc, int64_t now_ticks, const uint32_t *cnt,
                                   uint32_t *done, uint32_t cap, SelEntry *buf, double *buf_dot, SelEntry *lists, int32_t kprime,
                                   int64_t n_rows, int64_t row_base, const double *two_stage_L, orr_candidate *recs, uint32_t *cnt_host,
                                   hipStream_t s);
// survivors per group of that launch (its `lists`: B x cap / group x kSelWidth entries); 0 = groups of 64
int32_t finish_survivors_group(int32_t B, int32_t D);
hipError_t launch_records_dot_from_buffer(const SelEntry *buf, const double *buf_dot, const uint32_t *cnt, uint32_t cap, int32_t B,
                                          int32_t kprime, int64_t row_base, orr_candidate *recs, hipStream_t s);
hipError_t launch_buffer_to_lists(const SelEntry *buf, const uint32_t *cnt, uint32_t cap, int32_t B, int32_t seg_first,
                                  int32_t n_seg_total, SelEntry *out_sel, hipStream_t s);
// K2s: the same for B <= 32 queries, streaming (HBM-bound) structure.
hipError_t launch_gemv_mfma(const float *Q, int32_t B, const float *E, int64_t n_rows, int32_t D, float *S,
                            int64_t s_stride, hipStream_t s);
// K6: recomputes records[b][c].dot in the reference's exact arithmetic for every valid record
// (row = order_key - row_base) and sets ORR_CAND_DOT_EXACT.
hipError_t launch_rescore_exact(const float *E, int32_t D, const float *Q, int32_t B, int32_t kprime, int64_t row_base,
                                orr_candidate *recs, hipStream_t s);

// Sampling pass of the batched selection: tau_out[b] = k'-th best key among the first sample_seg
// lists of query b (0 if there are fewer).
// floor (optional): the two-stage floor of that key in the same launch (floor_key[b], L[b]; two_stage_floor_of, orr_device.h).
struct FloorOut {
    unsigned long long *floor_key = nullptr;
    double *L = nullptr;
    double eps3 = 0.0, eps1 = 0.0;
};
hipError_t launch_select_final_sample(const SelEntry *sel, int32_t n_seg_total, int32_t sample_seg, int32_t B,
                                      int32_t kprime, unsigned long long *tau_out, hipStream_t s, FloorOut floor = FloorOut(),
                                      int32_t wave_maxima = 0);    // 1: the lists hold 16 unsorted wave maxima each (fuse_select floor_only = 2)

// Generic path for large k: keys[r] = score key of (query b,row r), vals[r] = r.
hipError_t launch_score_keys(const double *dot, const double *norm_b, const int64_t *created,
                             KwView kw, int32_t b, QueryConst qc, int64_t now_ticks, int64_t n_rows,
                             unsigned long long *keys, uint32_t *vals, hipStream_t s);
// Device radix sort (descending, stable) of (keys, vals); temp sizing by query.
hipError_t sort_pairs_desc(void *temp, size_t &temp_bytes, const unsigned long long *keys_in,
                           unsigned long long *keys_out, const uint32_t *vals_in, uint32_t *vals_out,
                           int64_t n, hipStream_t s);
// Builds candidate records for the first K sorted entries (+ trailer at index K).
hipError_t launch_records_from_sorted(const unsigned long long *keys, const uint32_t *vals, int32_t K,
                                      int64_t n_rows, int64_t row_base, const double *dot, int64_t dot_stride,
                                      const double *norm_b, const int64_t *created, const int64_t *row_ids,
                                      KwView kw, int32_t b, int32_t dot_exact, orr_candidate *out,
                                      hipStream_t s);

// Seal-time row permutation (dst[p] = src[perm[p]]).
hipError_t launch_gather_rows_f32(const float *src, float *dst, const int64_t *perm, int64_t n, int32_t D,
                                  hipStream_t s);
hipError_t launch_gather_i64(const int64_t *src, int64_t *dst, const int64_t *perm, int64_t n, hipStream_t s);
// Content rows: dst row r <- src row perm[r] (perm == nullptr: identity).
hipError_t launch_gather_content(const uint8_t *src_pool, const uint64_t *src_start, const uint32_t *src_len,
                                 uint8_t *dst_pool, const uint64_t *dst_start, const int64_t *perm, int64_t n,
                                 hipStream_t s);
hipError_t launch_gather_u32(const uint32_t *src, uint32_t *dst, const int64_t *perm, int64_t n, hipStream_t s);
hipError_t launch_iota_i64(int64_t *dst, int64_t n, int64_t base, hipStream_t s);
// Deleted rows: overwrite norm and timestamp at the given positions; flag the records of deleted positions.
hipError_t launch_tombstone_rows(const int64_t *pos, int32_t n, double *norm_b, int64_t *created, hipStream_t s);
hipError_t launch_mark_dead_records(orr_candidate *recs, int32_t B, int32_t kprime, const int64_t *dead, int32_t n_dead,
                                    int64_t row_base, hipStream_t s);

}  // namespace orr
