// orr_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the hybrid recall-search scorer.
//
// Reference arithmetic being reproduced (src/OmniRecall.Api/Services/RecallSearchService.cs):
//   :77-82  dot/norm sums: float*float rounded to binary32, widened, added to a
//           binary64 accumulator in index order
//   :84-87  guards, dot / (sqrt(normA) * sqrt(normB))
//   :111-112 keyword matches / terms
//   :117-118 exp(-max(0, ageDays) / 30)
//   :66     (cos*0.7) + (kw*0.2) + (rec*0.1), left to right
//   :34-36  order by score desc, CreatedAt desc, stable  == (score desc, candidate
//           position asc) because rows are stored CreatedAt-descending
//
// Built with -ffp-contract=off: nothing here may be fused or re-associated.
#include "orr_kernels.h"
#include "orr_device.h"

#include <hipcub/hipcub.hpp>

#include <cstdlib>

namespace orr {

__global__ __launch_bounds__(256) void row_consts_kernel(const double *__restrict__ norm_b,
                                                         const int64_t *__restrict__ created, int64_t now_ticks,
                                                         int64_t n_rows, double2 *__restrict__ out)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x) {
        out[r] = row_consts_of(norm_b[r], created[r], now_ticks);
    }
}

// Query norms computed on the device (queries that already live there): fills the norm-dependent fields of the
// per-query constants exactly as the host does for host-resident queries.
__global__ void patch_query_norms_kernel(QueryConst *__restrict__ qc, const double *__restrict__ norm_a, int32_t B, int32_t batched,
                                         double *__restrict__ norm_host, const QueryConst *__restrict__ qc_host)
{
    const int32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double na = norm_a[b];
    QueryConst c = qc_host ? qc_host[b] : qc[b];          // (qc_host: the host's copy in pinned memory, read in place -- no upload command)
    c.norm_a = na;
    if (norm_host) norm_host[b] = na;                     // (pinned host memory: no copy command behind this launch)
    if (batched && c.use_cos) {
        if (na <= 0.0) c.use_cos = 0;                     // guard :84 -> cosine 0 for every row
        else c.inv_sqrt_na = 1.0 / sqrt(na);              // NaN stays NaN
    }
    qc[b] = c;
}

hipError_t launch_patch_query_norms(QueryConst *qc, const double *norm_a, int32_t B, bool batched, hipStream_t s, double *norm_host,
                                    const QueryConst *qc_host)
{
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(patch_query_norms_kernel, dim3((B + 255) / 256), dim3(256), 0, s, qc, norm_a, B, batched ? 1 : 0, norm_host, qc_host);
    return hipGetLastError();
}

hipError_t launch_row_consts(const double *norm_b, const int64_t *created, int64_t now_ticks, int64_t n_rows,
                             double2 *out, hipStream_t s)
{
    if (n_rows <= 0) return hipSuccess;
    int64_t blocks = (n_rows + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(row_consts_kernel, dim3((unsigned)blocks), dim3(256), 0, s, norm_b, created, now_ticks, n_rows, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K0 / K1e  exact dot: one row per lane, rows staged through a wave-private
// LDS tile so that HBM reads stay coalesced (4 rows x 256 B per wave
// instruction) while each lane walks its own row in index order.
//
// Tile: [64 rows][64 floats] = 16 KiB per wave, 16-byte chunks XOR-swizzled by
// (row & 15) so that the 16-lane groups of ds_read_b128 hit 16 distinct slots.
// ---------------------------------------------------------------------------
template <int NQ, bool SELF, bool PREFETCH, bool NT>
__global__ __launch_bounds__(256) void dot_exact_tiled(const float *__restrict__ E, int64_t n_rows, int32_t D,
                                                       const float *__restrict__ Q, double *__restrict__ out,
                                                       int64_t out_stride)
{
    __shared__ __attribute__((aligned(16))) float tile_all[4][64 * 64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float *tile = tile_all[wave];
    const int64_t n_groups = (n_rows + 63) >> 6;
    const int ld_row = lane >> 4;     // 0..3: row within a 4-row load
    const int ld_ch = lane & 15;      // 16-byte chunk within the 256-byte row piece

    for (int64_t g = (int64_t)blockIdx.x * 4 + wave; g < n_groups; g += (int64_t)gridDim.x * 4) {
        const int64_t row0 = g << 6;
        double acc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = 0.0;

        // global -> registers: 16 wave instructions of 4 rows x 256 B each
        auto load_stage = [&](float4 (&st)[16], int c0) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int r = it * 4 + ld_row;
                const int64_t row = row0 + r;
                st[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < n_rows) {
                    const float *src = E + row * (int64_t)D + c0 + ld_ch * 4;
                    if (NT) {   // streamed once: keep it out of the caches
                        typedef float f32x4 __attribute__((ext_vector_type(4)));
                        const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src));
                        st[it] = make_float4(v.x, v.y, v.z, v.w);
                    } else {
                        st[it] = *reinterpret_cast<const float4 *>(src);
                    }
                }
            }
        };
        float4 stage[16];
        if (PREFETCH) load_stage(stage, 0);

        for (int c0 = 0; c0 < D; c0 += 64) {
            if (!PREFETCH) load_stage(stage, c0);
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int r = it * 4 + ld_row;
                *reinterpret_cast<float4 *>(tile + r * 64 + ((ld_ch ^ (r & 15)) << 2)) = stage[it];
            }
            // next piece of the rows goes in flight while this one is consumed from LDS
            if (PREFETCH && c0 + 64 < D) load_stage(stage, c0 + 64);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float4 e = *reinterpret_cast<const float4 *>(tile + lane * 64 + ((j ^ (lane & 15)) << 2));
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    float x0, x1, x2, x3;
                    if (SELF) {
                        x0 = e.x; x1 = e.y; x2 = e.z; x3 = e.w;
                    } else {
                        const float *qp = Q + (int64_t)q * D + c0 + j * 4;   // wave-uniform: scalar loads
                        x0 = qp[0]; x1 = qp[1]; x2 = qp[2]; x3 = qp[3];
                    }
                    float p0 = x0 * e.x;
                    acc[q] += (double)p0;
                    float p1 = x1 * e.y;
                    acc[q] += (double)p1;
                    float p2 = x2 * e.z;
                    acc[q] += (double)p2;
                    float p3 = x3 * e.w;
                    acc[q] += (double)p3;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (row0 + lane < n_rows) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) out[(int64_t)q * out_stride + row0 + lane] = acc[q];
        }
    }
}

// Any D (tests use 2, 3, ...): one thread per row, plain loads.  Correctness path
// for dimensions the tiled kernel does not take (D % 64 != 0).
template <bool SELF>
__global__ __launch_bounds__(256) void dot_exact_generic(const float *__restrict__ E, int64_t n_rows, int32_t D,
                                                         const float *__restrict__ Q, int32_t nq,
                                                         double *__restrict__ out, int64_t out_stride)
{
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    const float *e = E + row * (int64_t)D;
    const int nqq = SELF ? 1 : nq;
    for (int q = 0; q < nqq; ++q) {
        const float *x = SELF ? e : Q + (int64_t)q * D;
        double acc = 0.0;
        for (int i = 0; i < D; ++i) {
            float p = x[i] * e[i];
            acc += (double)p;
        }
        out[(int64_t)q * out_stride + row] = acc;
    }
}

hipError_t launch_dot_exact(const float *E, int64_t n_rows, int32_t D, const float *Q, int32_t nq,
                            bool self_norm, double *out, int64_t out_stride, hipStream_t s)
{
    if (n_rows <= 0 || D <= 0) return hipSuccess;
    if (nq < 1 || nq > kMaxExactQ) return hipErrorInvalidValue;
    const bool tiled = (D % 64 == 0) && ((reinterpret_cast<uintptr_t>(E) & 15) == 0);
    if (!tiled) {
        const int64_t blocks = (n_rows + 255) / 256;
        if (self_norm)
            hipLaunchKernelGGL(dot_exact_generic<true>, dim3((unsigned)blocks), dim3(256), 0, s, E, n_rows, D, Q, 1, out, out_stride);
        else
            hipLaunchKernelGGL(dot_exact_generic<false>, dim3((unsigned)blocks), dim3(256), 0, s, E, n_rows, D, Q, nq, out, out_stride);
        return hipGetLastError();
    }
    const int64_t n_groups = (n_rows + 63) / 64;
    int64_t blocks = (n_groups + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;                               // 16 workgroups per CU, grid-stride beyond that (measured best of 2..16)
    dim3 grid((unsigned)blocks), block(256);
    // (register prefetch of the next piece + non-temporal loads: the fastest of the three forms measured in round 1)
#define ORR_LAUNCH_DOT(NQ_, SELF_) hipLaunchKernelGGL((dot_exact_tiled<NQ_, SELF_, true, true>), grid, block, 0, s, E, n_rows, D, Q, out, out_stride)
    if (self_norm) {
        ORR_LAUNCH_DOT(1, true);
    } else {
        switch (nq) {
        case 1: ORR_LAUNCH_DOT(1, false); break;
        case 2: ORR_LAUNCH_DOT(2, false); break;
        case 3: ORR_LAUNCH_DOT(3, false); break;
        case 4: ORR_LAUNCH_DOT(4, false); break;
        case 5: ORR_LAUNCH_DOT(5, false); break;
        case 6: ORR_LAUNCH_DOT(6, false); break;
        case 7: ORR_LAUNCH_DOT(7, false); break;
        default: ORR_LAUNCH_DOT(8, false); break;
        }
    }
#undef ORR_LAUNCH_DOT
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K3  keyword scan.  Device content pool layout (built at append): every row
// starts on a 16-byte boundary and is followed by 1..16 space bytes up to the next
// boundary.  Query terms never contain whitespace (they come out of a whitespace
// split, RecallSearchService.cs:95), so no term can match across a row boundary
// and only whole lanes past the row end have to be masked.
//
// One wave per row; per step each lane owns 16 consecutive content bytes and the
// 16 four-byte windows starting in them (v_alignbyte), computed once and reused
// for every term.  A term costs one v_cmp per start position (its lane mask is
// OR-ed in scalar registers); terms longer than 4 bytes are verified byte-wise
// only where their 4-byte prefix hit.
// ---------------------------------------------------------------------------
// Tells the compiler a value is the same in every lane (it then lives in SGPRs).
__device__ __forceinline__ int uniform32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t uniform64(int64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

template <bool FULL_MASK>
__device__ __forceinline__ unsigned long long prefix_hits(const uint32_t (&win)[16], uint32_t prefix, uint32_t mask)
{
    unsigned long long any = 0ull;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t v = FULL_MASK ? win[i] : (win[i] & mask);
        any |= __ballot(v == prefix);
    }
    return any;
}

__global__ __launch_bounds__(256) void keyword_scan_kernel(const uint8_t *__restrict__ pool,
                                                           const uint64_t *__restrict__ cstart,
                                                           const uint32_t *__restrict__ clen, int64_t n_rows,
                                                           const uint8_t *__restrict__ term_pool,
                                                           const ScanTerm *__restrict__ terms, int32_t n_terms,
                                                           const uint32_t *__restrict__ q_term_off, int32_t B,
                                                           uint16_t *__restrict__ matches, int64_t matches_stride,
                                                           int32_t accumulate, int32_t group_terms)
{
    // group_terms > 0 (vocabulary form): blockIdx.y selects a group of single-term "queries"; n_terms is the
    // total, q_term_off the identity table 0..group_terms
    if (group_terms > 0) {
        const int32_t first = (int32_t)blockIdx.y * group_terms;
        terms += first;
        matches += (int64_t)first * matches_stride;
        n_terms = n_terms - first < group_terms ? n_terms - first : group_terms;
        B = n_terms;
    }
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + uniform32((int)(threadIdx.x >> 6));
    const int64_t n_waves = (int64_t)gridDim.x * 4;

    for (int64_t row = wave_id; row < n_rows; row += n_waves) {
        const uint64_t start = (uint64_t)uniform64((int64_t)cstart[row]);    // 16-byte aligned
        const int len = uniform32((int)clen[row]);
        unsigned long long found = 0ull;                      // wave-uniform bit per term

        for (int cb = 0; cb < len; cb += 1024) {
            const uint8_t *p = pool + start + cb + lane * 16;
            const uint4 w = *reinterpret_cast<const uint4 *>(p);
            const uint32_t w4 = *reinterpret_cast<const uint32_t *>(p + 16);
            uint32_t win[16];
            win[0] = w.x;  win[1] = __builtin_amdgcn_alignbyte(w.y, w.x, 1);
            win[2] = __builtin_amdgcn_alignbyte(w.y, w.x, 2);  win[3] = __builtin_amdgcn_alignbyte(w.y, w.x, 3);
            win[4] = w.y;  win[5] = __builtin_amdgcn_alignbyte(w.z, w.y, 1);
            win[6] = __builtin_amdgcn_alignbyte(w.z, w.y, 2);  win[7] = __builtin_amdgcn_alignbyte(w.z, w.y, 3);
            win[8] = w.z;  win[9] = __builtin_amdgcn_alignbyte(w.w, w.z, 1);
            win[10] = __builtin_amdgcn_alignbyte(w.w, w.z, 2); win[11] = __builtin_amdgcn_alignbyte(w.w, w.z, 3);
            win[12] = w.w; win[13] = __builtin_amdgcn_alignbyte(w4, w.w, 1);
            win[14] = __builtin_amdgcn_alignbyte(w4, w.w, 2);  win[15] = __builtin_amdgcn_alignbyte(w4, w.w, 3);
            // lanes whose 16 bytes begin inside the row (later lanes hold padding or the next row)
            const int live = (len - cb + 15) >> 4;
            const unsigned long long live_mask = live >= 64 ? ~0ull : ((1ull << live) - 1ull);

            for (int t = 0; t < n_terms; ++t) {
                if ((found >> t) & 1ull) continue;
                const ScanTerm tm = terms[t];
                if (tm.len == 0 || (int)tm.len > len) continue;
                unsigned long long any = (tm.mask == 0xFFFFFFFFu) ? prefix_hits<true>(win, tm.prefix, tm.mask)
                                                                  : prefix_hits<false>(win, tm.prefix, tm.mask);
                any &= live_mask;
                if (any == 0ull) continue;
                if (tm.len <= 4) { found |= (1ull << t); continue; }
                // rare: the 4-byte prefix hit somewhere in this step; verify the tail where it hit
                bool hit = false;
                const int pos0 = cb + lane * 16;
                for (int i = 0; i < 16; ++i) {
                    const int pos = pos0 + i;
                    if (win[i] == tm.prefix && pos + (int)tm.len <= len) {
                        bool ok = true;
                        for (uint32_t k = 4; k < tm.len; ++k)
                            ok = ok && (pool[start + pos + k] == term_pool[tm.off + k]);
                        hit = hit || ok;
                    }
                }
                if (__any(hit)) found |= (1ull << t);
            }
        }
        if (lane < B) {
            const uint32_t t0 = q_term_off[lane], t1 = q_term_off[lane + 1];
            unsigned long long m = found >> t0;
            const uint32_t nt = t1 - t0;
            if (nt < 64) m &= ((1ull << nt) - 1ull);
            uint16_t *dst = matches + (int64_t)lane * matches_stride + row;
            *dst = (uint16_t)((accumulate ? *dst : 0) + __popcll(m));
        }
    }
}

hipError_t launch_keyword_scan(const uint8_t *pool, const uint64_t *cstart, const uint32_t *clen, int64_t n_rows,
                               const uint8_t *term_pool, const ScanTerm *terms, int32_t n_terms,
                               const uint32_t *q_term_off, int32_t B, uint16_t *matches,
                               int64_t matches_stride, int32_t accumulate, hipStream_t s)
{
    if (n_rows <= 0 || B <= 0) return hipSuccess;
    if (n_terms > kMaxScanTerms || B > 64) return hipErrorInvalidValue;
    int64_t blocks = (n_rows + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(keyword_scan_kernel, dim3((unsigned)blocks), dim3(256), 0, s, pool, cstart, clen, n_rows, term_pool,
                       terms, n_terms, q_term_off, B, matches, matches_stride, accumulate, 0);
    return hipGetLastError();
}

// Vocabulary form: every one of n_terms distinct terms is its own single-term query; vmatch[t][v] = 1 iff
// term t occurs inside token v.  All groups of kMaxScanTerms terms run in ONE launch (grid.y), which matters
// when the vocabulary is small and a launch is mostly latency.  identity = {0, 1, ..., kMaxScanTerms}.
hipError_t launch_vocab_scan(const uint8_t *vpool, const uint64_t *vstart, const uint32_t *vlen, int64_t n_tokens,
                             const uint8_t *term_pool, const ScanTerm *terms, int32_t n_terms, const uint32_t *identity,
                             uint16_t *vmatch, hipStream_t s)
{
    if (n_tokens <= 0 || n_terms <= 0) return hipSuccess;
    const int32_t groups = (n_terms + kMaxScanTerms - 1) / kMaxScanTerms;
    if (groups > 65535) return hipErrorInvalidValue;
    int64_t blocks = (n_tokens + 3) / 4;
    const int64_t cap = std::max<int64_t>(64, (256 * 8) / groups);
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(keyword_scan_kernel, dim3((unsigned)blocks, (unsigned)groups), dim3(256), 0, s, vpool, vstart, vlen, n_tokens,
                       term_pool, terms, n_terms, identity, 0, vmatch, n_tokens, 0, kMaxScanTerms);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Token index, query side.  The scan kernel above is run over the VOCABULARY
// (one "row" per distinct token of the shard) with one single-term "query" per
// distinct query term: vmatch[t][v] = 1 iff term t occurs inside token v.  A
// whitespace-free term occurs in a row's content iff it occurs inside one of the
// row's whitespace-delimited tokens, so OR-ing the posting lists of the matching
// tokens gives exactly `contentLower.Contains(term)` (RecallSearchService.cs:111)
// for every row.
//
// vocab_hits: one thread per (token, term); every hit reserves its slot in the hit
// list and its run of 1024-posting chunks with ONE 64-bit atomic (count in the high
// word, chunks in the low word), so hit order and chunk order agree.
// expand_hits: grid-stride over chunks; binary search for the owning hit, then 64
// postings per step, one atomicOr each into the term's row bitmap.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vocab_hits_kernel(const uint16_t *__restrict__ vmatch, int64_t n_tokens,
                                                         int32_t n_terms, const uint32_t *__restrict__ token_ids,
                                                         const uint64_t *__restrict__ post_off,
                                                         unsigned long long *__restrict__ counter,
                                                         KwHit *__restrict__ hits, uint32_t max_hits)
{
    const int64_t total = n_tokens * (int64_t)n_terms;
    for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
        if (vmatch[id] == 0) continue;                       // vmatch is [term][token]
        const int64_t t = id / n_tokens;
        int64_t v = id - t * n_tokens;
        if (token_ids) v = token_ids[v];                     // the scanned rows were a subset of the vocabulary
        const uint64_t p0 = post_off[v], p1 = post_off[v + 1];
        const uint32_t chunks = (uint32_t)((p1 - p0 + kPostChunk - 1) / kPostChunk);
        const unsigned long long old = atomicAdd(counter, (1ull << 32) | chunks);
        const uint32_t slot = (uint32_t)(old >> 32);
        if (slot < max_hits) {
            KwHit h;
            h.post_begin = p0; h.post_len = (uint32_t)(p1 - p0); h.chunk_base = (uint32_t)old; h.term = (uint32_t)t; h.token = (uint32_t)v;
            hits[slot] = h;
        }
    }
}

// One thread per hit: how many hits each distinct term has, and (for the terms with one) which token it was.
__global__ __launch_bounds__(256) void kw_term_hits_kernel(const KwHit *__restrict__ hits, const unsigned long long *__restrict__ counter,
                                                           uint32_t max_hits, uint32_t *__restrict__ term_cnt, uint32_t *__restrict__ term_tok)
{
    uint32_t n_hits = (uint32_t)(*counter >> 32);
    if (n_hits > max_hits) n_hits = max_hits;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_hits; i += gridDim.x * blockDim.x) {
        const KwHit h = hits[i];
        atomicAdd(&term_cnt[h.term], 1u);
        term_tok[h.term] = h.token;                    // (the only writer where the count ends up 1)
    }
}

__global__ __launch_bounds__(256) void kw_alias_kernel(const uint32_t *__restrict__ term_cnt, const uint32_t *__restrict__ term_tok,
                                                       const int32_t *__restrict__ tok_bm_index, int32_t n_terms, int64_t bm_store_delta,
                                                       int64_t words_per_term, int64_t *__restrict__ term_word_off, uint8_t *__restrict__ alias)
{
    const int32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_terms) return;
    int32_t bm = -1;
    if (term_cnt[t] == 1u && tok_bm_index) bm = tok_bm_index[term_tok[t]];
    alias[t] = bm >= 0 ? 1 : 0;
    term_word_off[t] = bm >= 0 ? bm_store_delta + (int64_t)bm * words_per_term : (int64_t)t * words_per_term;
}

// Both steps in ONE workgroup for the few terms of a small batch (a one-query call pays every launch boundary of the keyword
// chain in front of its stream): the counts live in LDS.
constexpr int kKwAliasSmall = 512;
__global__ __launch_bounds__(256) void kw_alias_small_kernel(const KwHit *__restrict__ hits, const unsigned long long *__restrict__ counter,
                                                             uint32_t max_hits, const int32_t *__restrict__ tok_bm_index, int32_t n_terms,
                                                             int64_t bm_store_delta, int64_t words_per_term,
                                                             int64_t *__restrict__ term_word_off, uint8_t *__restrict__ alias)
{
    __shared__ uint32_t cnt[kKwAliasSmall], tok[kKwAliasSmall];
    for (int t = threadIdx.x; t < n_terms; t += blockDim.x) { cnt[t] = 0u; tok[t] = 0u; }
    __syncthreads();
    uint32_t n_hits = (uint32_t)(*counter >> 32);
    if (n_hits > max_hits) n_hits = max_hits;
    for (uint32_t i = threadIdx.x; i < n_hits; i += blockDim.x) {
        const KwHit h = hits[i];
        atomicAdd(&cnt[h.term], 1u);
        tok[h.term] = h.token;                         // (the only writer where the count ends up 1)
    }
    __syncthreads();
    for (int t = threadIdx.x; t < n_terms; t += blockDim.x) {
        int32_t bm = -1;
        if (cnt[t] == 1u && tok_bm_index) bm = tok_bm_index[tok[t]];
        alias[t] = bm >= 0 ? 1 : 0;
        term_word_off[t] = bm >= 0 ? bm_store_delta + (int64_t)bm * words_per_term : (int64_t)t * words_per_term;
    }
}

hipError_t launch_kw_alias(const KwHit *hits, const unsigned long long *counter, uint32_t max_hits, int32_t n_terms,
                           const int32_t *tok_bm_index, int64_t bm_store_delta, int64_t words_per_term, uint32_t *term_cnt,
                           uint32_t *term_tok, int64_t *term_word_off, uint8_t *alias, hipStream_t s)
{
    if (n_terms <= 0) return hipSuccess;
    if (n_terms <= 64) {                               // (a handful of queries: beyond, many workgroups count the hits faster)
        hipLaunchKernelGGL(kw_alias_small_kernel, dim3(1), dim3(256), 0, s, hits, counter, max_hits, tok_bm_index, n_terms, bm_store_delta,
                           words_per_term, term_word_off, alias);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(kw_term_hits_kernel, dim3(64), dim3(256), 0, s, hits, counter, max_hits, term_cnt, term_tok);
    hipLaunchKernelGGL(kw_alias_kernel, dim3((unsigned)((n_terms + 255) / 256)), dim3(256), 0, s, term_cnt, term_tok, tok_bm_index, n_terms,
                       bm_store_delta, words_per_term, term_word_off, alias);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void expand_hits_kernel(const KwHit *__restrict__ hits,
                                                          const unsigned long long *__restrict__ counter,
                                                          uint32_t max_hits, const uint32_t *__restrict__ post_rows,
                                                          uint32_t *__restrict__ bitmaps, int64_t words_per_term,
                                                          unsigned long long *__restrict__ counter_host, const uint8_t *__restrict__ skip_term)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long cnt = *counter;
    if (counter_host && blockIdx.x == 0 && threadIdx.x == 0) *counter_host = cnt;      // statistics (pinned host memory)
    uint32_t n_hits = (uint32_t)(cnt >> 32);
    if (n_hits > max_hits) n_hits = max_hits;       // overflow is detected and retried by the host
    const uint32_t n_chunks = (uint32_t)cnt;
    const uint32_t wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    // a wave takes a CONTIGUOUS run of chunks: one binary search for the hit of its first chunk, a walk along the list for the
    // others (chunk numbers ascend with the hits) -- a fresh search per chunk was fifteen dependent loads each
    const uint32_t per_wave = (n_chunks + n_waves - 1) / n_waves;
    const uint32_t c_first = wave_id * per_wave, c_end = c_first + per_wave < n_chunks ? c_first + per_wave : n_chunks;
    uint32_t lo = 0;
    if (c_first < c_end) {
        uint32_t hi = n_hits;                        // last hit with chunk_base <= c_first
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (hits[mid].chunk_base <= c_first) lo = mid; else hi = mid;
        }
    }
    for (uint32_t c = c_first; c < c_end; ++c) {
        while (lo + 1 < n_hits && hits[lo + 1].chunk_base <= c) ++lo;
        const KwHit h = hits[lo];
        if (skip_term && skip_term[h.term]) continue;   // the term's bitmap is a stored token bitmap: nothing to expand
        const uint64_t post_end = h.post_begin + h.post_len;
        const uint64_t b0 = h.post_begin + (uint64_t)(c - h.chunk_base) * kPostChunk;
        const uint64_t b1 = b0 + kPostChunk < post_end ? b0 + kPostChunk : post_end;
        uint32_t *bm = bitmaps + (int64_t)h.term * words_per_term;
        // posting rows ascend, so the lanes that hit one bitmap word are neighbours: OR their bits
        // together (segmented, 5 steps: a word has 32 bits) and let the last lane of each run do the atomic
        // (four groups of 64 postings in flight: a one-query search has a handful of chunks, each a chain of
        // load -> atomic round trips otherwise)
        for (uint64_t p0 = b0; p0 < b1; p0 += 256) {
            uint32_t rows4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint64_t p = p0 + (uint64_t)g * 64 + lane;
                rows4[g] = p < b1 ? post_rows[p] : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint32_t row = rows4[g];
                const bool live = row != 0xFFFFFFFFu;
                const uint32_t word = row >> 5;
                uint32_t bits = live ? 1u << (row & 31) : 0u;
#pragma unroll
                for (int d = 1; d < 32; d <<= 1) {
                    const uint32_t ow = (uint32_t)__shfl_up((int)word, d, 64), ob = (uint32_t)__shfl_up((int)bits, d, 64);
                    if (lane >= d && ow == word) bits |= ob;
                }
                const uint32_t nw = (uint32_t)__shfl_down((int)word, 1, 64);
                if (live && (lane == 63 || nw != word)) atomicOr(&bm[word], bits);
            }
        }
    }
}

hipError_t launch_vocab_hits(const uint16_t *vmatch, int64_t n_tokens, int32_t n_terms, const uint32_t *token_ids,
                             const uint64_t *post_off, unsigned long long *counter, KwHit *hits, uint32_t max_hits, hipStream_t s)
{
    if (n_tokens <= 0 || n_terms <= 0) return hipSuccess;
    int64_t blocks = (n_tokens * (int64_t)n_terms + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(vocab_hits_kernel, dim3((unsigned)blocks), dim3(256), 0, s, vmatch, n_tokens, n_terms, token_ids, post_off,
                       counter, hits, max_hits);
    return hipGetLastError();
}

// Vocabulary tokens of at most 16 bytes (all but URLs and the like): ONE LANE per token instead of one wave.
// The lane keeps its token's 16 four-byte windows in registers; the terms of the block's group are uniform
// (scalar loads), each as up to four dwords with byte masks.  A term occurs in the token iff at some start i
// with i + len(term) <= len(token) every dword of the term equals the window at i + 4j under its mask; bytes
// past the token are padding and never take part.  A hit reserves its slot like vocab_hits does.
__global__ __launch_bounds__(256) void vocab_match_short_kernel(const uint8_t *__restrict__ vpool,
                                                                const uint64_t *__restrict__ vstart,
                                                                const uint32_t *__restrict__ vlen, int64_t n_tokens,
                                                                const MatchTerm *__restrict__ terms, int32_t n_terms,
                                                                const uint64_t *__restrict__ post_off,
                                                                unsigned long long *__restrict__ counter,
                                                                KwHit *__restrict__ hits, uint32_t max_hits)
{
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int32_t t0 = (int32_t)blockIdx.y * kMatchGroup;
    const int32_t t1 = n_terms < t0 + kMatchGroup ? n_terms : t0 + kMatchGroup;
    int32_t len = 0;
    uint4 w = make_uint4(0x20202020u, 0x20202020u, 0x20202020u, 0x20202020u);
    if (v < n_tokens) {
        const uint32_t l = vlen[v];
        if (l >= 1 && l <= 16) { len = (int32_t)l; w = *reinterpret_cast<const uint4 *>(vpool + vstart[v]); }   // rows start 16-byte aligned
    }
    const uint32_t w4 = 0x20202020u;
    uint32_t win[16];
    win[0] = w.x;  win[1] = __builtin_amdgcn_alignbyte(w.y, w.x, 1);
    win[2] = __builtin_amdgcn_alignbyte(w.y, w.x, 2);  win[3] = __builtin_amdgcn_alignbyte(w.y, w.x, 3);
    win[4] = w.y;  win[5] = __builtin_amdgcn_alignbyte(w.z, w.y, 1);
    win[6] = __builtin_amdgcn_alignbyte(w.z, w.y, 2);  win[7] = __builtin_amdgcn_alignbyte(w.z, w.y, 3);
    win[8] = w.z;  win[9] = __builtin_amdgcn_alignbyte(w.w, w.z, 1);
    win[10] = __builtin_amdgcn_alignbyte(w.w, w.z, 2); win[11] = __builtin_amdgcn_alignbyte(w.w, w.z, 3);
    win[12] = w.w; win[13] = __builtin_amdgcn_alignbyte(w4, w.w, 1);
    win[14] = __builtin_amdgcn_alignbyte(w4, w.w, 2);  win[15] = __builtin_amdgcn_alignbyte(w4, w.w, 3);

    for (int32_t t = t0; t < t1; ++t) {
        const uint32_t tlen = (uint32_t)uniform32((int)terms[t].len);
        if (tlen == 0 || tlen > 16) continue;                     // longer terms only fit longer tokens
        const uint32_t w0 = (uint32_t)uniform32((int)terms[t].w[0]), m0 = (uint32_t)uniform32((int)terms[t].m[0]);
        uint32_t cand = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) cand |= ((win[i] & m0) == w0) ? (1u << i) : 0u;
        const int32_t last = len - (int32_t)tlen;                 // last start that keeps the term inside the token
        cand = last < 0 ? 0u : (cand & ((2u << last) - 1u));
        if (__ballot(cand != 0u) == 0ull) continue;
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            if (tlen <= 4u * j) break;
            const uint32_t wj = (uint32_t)uniform32((int)terms[t].w[j]), mj = (uint32_t)uniform32((int)terms[t].m[j]);
#pragma unroll
            for (int i = 0; i + 4 * j < 16; ++i)
                if ((win[i + 4 * j] & mj) != wj) cand &= ~(1u << i);
        }
        if (cand == 0u) continue;
        const uint64_t p0 = post_off[v], p1 = post_off[v + 1];
        const uint32_t chunks = (uint32_t)((p1 - p0 + kPostChunk - 1) / kPostChunk);
        const unsigned long long old = atomicAdd(counter, (1ull << 32) | chunks);
        const uint32_t slot = (uint32_t)(old >> 32);
        if (slot < max_hits) {
            KwHit h;
            h.post_begin = p0; h.post_len = (uint32_t)(p1 - p0); h.chunk_base = (uint32_t)old; h.term = (uint32_t)t; h.token = (uint32_t)v;
            hits[slot] = h;
        }
    }
}

// Many terms (a batch of 256+ queries against a vocabulary of 10^5..10^6 tokens): the all-pairs form above costs tokens x terms
// first-dword compares.  Here the terms are LOOKED UP instead: the host sorts them by their (length-masked) first dword in four
// classes -- 1, 2, 3 and 4+ bytes -- and a lane searches each of its token's 16 windows in each non-empty class (binary search,
// then the run of equal keys), verifying the few candidates like the all-pairs form does.  One hit per (token, term): a start
// only counts if no earlier start of the token matches the term too.
//   lk[0..4] = class boundaries in keys / tidx (class c holds the terms of c + 1 bytes, class 3 those of 4..16 bytes).
__device__ __forceinline__ bool term_at(const uint32_t (&win)[16], int i, const MatchTerm &mt)
{
    // (bytes behind the token are spaces, which no term contains: a start too close to the end fails by itself)
    bool ok = (win[i] & mt.m[0]) == mt.w[0];
#pragma unroll
    for (int j = 1; j < 4; ++j)
        if (mt.len > 4u * j) {
            const uint32_t wv = i + 4 * j < 16 ? win[(i + 4 * j) & 15] : 0x20202020u;
            ok = ok && (wv & mt.m[j]) == mt.w[j];
        }
    return ok;
}

__global__ __launch_bounds__(256) void vocab_match_lookup_kernel(const uint8_t *__restrict__ vpool, const uint64_t *__restrict__ vstart,
                                                                 const uint32_t *__restrict__ vlen, int64_t n_tokens,
                                                                 const MatchTerm *__restrict__ terms, const uint32_t *__restrict__ lk,
                                                                 const uint32_t *__restrict__ keys, const uint32_t *__restrict__ tidx,
                                                                 const uint32_t *__restrict__ bloom,
                                                                 const uint64_t *__restrict__ post_off, unsigned long long *__restrict__ counter,
                                                                 KwHit *__restrict__ hits, uint32_t max_hits)
{
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int32_t len = 0;
    uint4 w = make_uint4(0x20202020u, 0x20202020u, 0x20202020u, 0x20202020u);
    if (v < n_tokens) {
        const uint32_t l = vlen[v];
        if (l >= 1 && l <= 16) { len = (int32_t)l; w = *reinterpret_cast<const uint4 *>(vpool + vstart[v]); }   // rows start 16-byte aligned
    }
    const uint32_t w4 = 0x20202020u;
    uint32_t win[16];
    win[0] = w.x;  win[1] = __builtin_amdgcn_alignbyte(w.y, w.x, 1);
    win[2] = __builtin_amdgcn_alignbyte(w.y, w.x, 2);  win[3] = __builtin_amdgcn_alignbyte(w.y, w.x, 3);
    win[4] = w.y;  win[5] = __builtin_amdgcn_alignbyte(w.z, w.y, 1);
    win[6] = __builtin_amdgcn_alignbyte(w.z, w.y, 2);  win[7] = __builtin_amdgcn_alignbyte(w.z, w.y, 3);
    win[8] = w.z;  win[9] = __builtin_amdgcn_alignbyte(w.w, w.z, 1);
    win[10] = __builtin_amdgcn_alignbyte(w.w, w.z, 2); win[11] = __builtin_amdgcn_alignbyte(w.w, w.z, 3);
    win[12] = w.w; win[13] = __builtin_amdgcn_alignbyte(w4, w.w, 1);
    win[14] = __builtin_amdgcn_alignbyte(w4, w.w, 2);  win[15] = __builtin_amdgcn_alignbyte(w4, w.w, 3);
    // a Bloom filter of the (class, key) pairs in LDS (built by the host, kVocabBloomBits bits): sixteen independent reads per class
    // instead of sixteen binary searches of dependent loads; a window goes to the search only when its bit is set
    __shared__ uint32_t bloom_lds[kVocabBloomBits / 32];
    // The hits of a workgroup are staged in LDS and reserved in the global list with ONE atomic (hit slots and posting chunks come
    // out of one 64-bit counter and must stay jointly ascending: expand_hits searches the list by chunk number): a batch of 256
    // queries against 250,000 tokens leaves 23,000 hits, and as many atomics on one address were the whole cost of the all-pairs
    // form (0.27 ms of which 0.02 is comparing).
    constexpr uint32_t kStage = 512;
    __shared__ KwHit staged[kStage];
    __shared__ unsigned long long local_ctr, base_ctr;
    for (int i = threadIdx.x; i < kVocabBloomBits / 32; i += 256) bloom_lds[i] = bloom[i];
    if (threadIdx.x == 0) local_ctr = 0ull;
    __syncthreads();
    if (len != 0) {
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {                                 // (not unrolled: sixteen copies of the search are code enough)
        const uint32_t c0 = lk[c], c1 = lk[c + 1];
        if (c0 == c1) continue;                                   // (uniform: no term of this class in the batch)
        const uint32_t mask = c == 0 ? 0xFFu : c == 1 ? 0xFFFFu : c == 2 ? 0xFFFFFFu : 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i + c + 1 > len) continue;                        // the shortest term of this class does not fit behind this start
            const uint32_t key = win[i] & mask;
            const uint32_t hb = vocab_bloom_hash(key, (uint32_t)c);
            if (!((bloom_lds[hb >> 5] >> (hb & 31u)) & 1u)) continue;
            uint32_t lo = c0, hi = c1;                            // lower bound of key in keys[c0, c1)
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (keys[mid] < key) lo = mid + 1; else hi = mid;
            }
            for (uint32_t pos = lo; pos < c1 && keys[pos] == key; ++pos) {
                const uint32_t t = tidx[pos];
                const MatchTerm mt = terms[t];
                if ((int32_t)mt.len > len - i) continue;          // the term would run past the token's end
                if (!term_at(win, i, mt)) continue;
                bool earlier = false;                             // one hit per (token, term)
                for (int e = 0; e < i; ++e) earlier = earlier || term_at(win, e, mt);
                if (earlier) continue;
                const uint64_t p0 = post_off[v], p1 = post_off[v + 1];
                const uint32_t chunks = (uint32_t)((p1 - p0 + kPostChunk - 1) / kPostChunk);
                KwHit h;
                h.post_begin = p0; h.post_len = (uint32_t)(p1 - p0); h.term = t; h.token = (uint32_t)v;
                const unsigned long long mine = atomicAdd(&local_ctr, (1ull << 32) | chunks);
                if ((uint32_t)(mine >> 32) < kStage) {
                    h.chunk_base = (uint32_t)mine;                // (relative to the workgroup's block: rebased below)
                    staged[(uint32_t)(mine >> 32)] = h;
                } else {                                          // (more hits than the stage holds: straight into the list)
                    const unsigned long long old = atomicAdd(counter, (1ull << 32) | chunks);
                    const uint32_t slot = (uint32_t)(old >> 32);
                    h.chunk_base = (uint32_t)old;
                    if (slot < max_hits) hits[slot] = h;
                }
            }
        }
    }
    }
    __syncthreads();
    // the staged hits: everything up to the first one that did not fit (hits and chunks counted in the order of the local counter)
    const uint32_t n_local = (uint32_t)(local_ctr >> 32) < kStage ? (uint32_t)(local_ctr >> 32) : kStage;
    if (n_local == 0) return;
    if (threadIdx.x == 0) {
        // chunks of the staged hits = the local chunk counter where hit number n_local began (or the total when all fit)
        uint32_t staged_chunks = (uint32_t)local_ctr;
        if ((uint32_t)(local_ctr >> 32) > kStage) {
            const KwHit &last = staged[kStage - 1];
            staged_chunks = last.chunk_base + (uint32_t)((last.post_len + kPostChunk - 1) / kPostChunk);
        }
        base_ctr = atomicAdd(counter, ((unsigned long long)n_local << 32) | staged_chunks);
    }
    __syncthreads();
    const uint32_t base_slot = (uint32_t)(base_ctr >> 32), base_chunk = (uint32_t)base_ctr;
    for (uint32_t i = threadIdx.x; i < n_local; i += 256) {
        KwHit h = staged[i];
        h.chunk_base += base_chunk;
        if (base_slot + i < max_hits) hits[base_slot + i] = h;
    }
}

hipError_t launch_vocab_match_lookup(const uint8_t *vpool, const uint64_t *vstart, const uint32_t *vlen, int64_t n_tokens,
                                     const MatchTerm *terms, const uint32_t *lk, const uint32_t *keys, const uint32_t *tidx,
                                     const uint32_t *bloom, const uint64_t *post_off, unsigned long long *counter, KwHit *hits,
                                     uint32_t max_hits, hipStream_t s)
{
    if (n_tokens <= 0) return hipSuccess;
    const int64_t blocks = (n_tokens + 255) / 256;
    if (blocks > 0x7FFFFFFF) return hipErrorInvalidValue;
    hipLaunchKernelGGL(vocab_match_lookup_kernel, dim3((unsigned)blocks), dim3(256), 0, s, vpool, vstart, vlen, n_tokens, terms, lk, keys, tidx,
                       bloom, post_off, counter, hits, max_hits);
    return hipGetLastError();
}

// The same for tokens of 17..32 bytes, given as a list (start, length, token number): 32 windows per lane, terms of up to 32
// bytes (eight dwords); longer terms cannot occur in these tokens.
__global__ __launch_bounds__(256) void vocab_match_mid_kernel(const uint8_t *__restrict__ vpool, const uint64_t *__restrict__ starts,
                                                              const uint32_t *__restrict__ lens, const uint32_t *__restrict__ ids,
                                                              int64_t n_list, const MatchTerm8 *__restrict__ terms, int32_t n_terms,
                                                              const uint64_t *__restrict__ post_off, unsigned long long *__restrict__ counter,
                                                              KwHit *__restrict__ hits, uint32_t max_hits)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int32_t t0 = (int32_t)blockIdx.y * kMatchGroup;
    const int32_t t1 = n_terms < t0 + kMatchGroup ? n_terms : t0 + kMatchGroup;
    int32_t len = 0;
    uint32_t v = 0u;
    uint32_t w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = 0x20202020u;
    if (i < n_list) {
        const uint32_t l = lens[i];
        if (l >= 17 && l <= 32) {
            len = (int32_t)l;
            v = ids[i];
            const uint4 a = *reinterpret_cast<const uint4 *>(vpool + starts[i]);            // rows start 16-byte aligned; a token
            const uint4 b = *reinterpret_cast<const uint4 *>(vpool + starts[i] + 16);       // of 17..32 bytes owns at least 32
            w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
        }
    }
    uint32_t win[32];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        win[4 * k] = w[k];
        win[4 * k + 1] = __builtin_amdgcn_alignbyte(w[k + 1], w[k], 1);
        win[4 * k + 2] = __builtin_amdgcn_alignbyte(w[k + 1], w[k], 2);
        win[4 * k + 3] = __builtin_amdgcn_alignbyte(w[k + 1], w[k], 3);
    }
    for (int32_t t = t0; t < t1; ++t) {
        const uint32_t tlen = (uint32_t)uniform32((int)terms[t].len);
        if (tlen == 0 || tlen > 32) continue;
        const uint32_t w0 = (uint32_t)uniform32((int)terms[t].w[0]), m0 = (uint32_t)uniform32((int)terms[t].m[0]);
        uint32_t cand = 0;
#pragma unroll
        for (int k = 0; k < 32; ++k) cand |= ((win[k] & m0) == w0) ? (1u << k) : 0u;
        const int32_t last = len - (int32_t)tlen;                 // last start that keeps the term inside the token (<= 31)
        cand = last < 0 ? 0u : (cand & ((2u << last) - 1u));
        if (__ballot(cand != 0u) == 0ull) continue;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            if (tlen <= 4u * j) break;
            const uint32_t wj = (uint32_t)uniform32((int)terms[t].w[j]), mj = (uint32_t)uniform32((int)terms[t].m[j]);
#pragma unroll
            for (int k = 0; k + 4 * j < 32; ++k)
                if ((win[k + 4 * j] & mj) != wj) cand &= ~(1u << k);
        }
        if (cand == 0u) continue;
        const uint64_t p0 = post_off[v], p1 = post_off[v + 1];
        const uint32_t chunks = (uint32_t)((p1 - p0 + kPostChunk - 1) / kPostChunk);
        const unsigned long long old = atomicAdd(counter, (1ull << 32) | chunks);
        const uint32_t slot = (uint32_t)(old >> 32);
        if (slot < max_hits) {
            KwHit h;
            h.post_begin = p0; h.post_len = (uint32_t)(p1 - p0); h.chunk_base = (uint32_t)old; h.term = (uint32_t)t; h.token = v;
            hits[slot] = h;
        }
    }
}

hipError_t launch_vocab_match_mid(const uint8_t *vpool, const uint64_t *starts, const uint32_t *lens, const uint32_t *ids, int64_t n_list,
                                  const MatchTerm8 *terms, int32_t n_terms, const uint64_t *post_off, unsigned long long *counter,
                                  KwHit *hits, uint32_t max_hits, hipStream_t s)
{
    if (n_list <= 0 || n_terms <= 0) return hipSuccess;
    const int64_t blocks = (n_list + 255) / 256;
    const int32_t groups = (n_terms + kMatchGroup - 1) / kMatchGroup;
    if (groups > 65535 || blocks > 0x7FFFFFFF) return hipErrorInvalidValue;
    hipLaunchKernelGGL(vocab_match_mid_kernel, dim3((unsigned)blocks, (unsigned)groups), dim3(256), 0, s, vpool, starts, lens, ids, n_list,
                       terms, n_terms, post_off, counter, hits, max_hits);
    return hipGetLastError();
}

hipError_t launch_vocab_match_short(const uint8_t *vpool, const uint64_t *vstart, const uint32_t *vlen, int64_t n_tokens,
                                    const MatchTerm *terms, int32_t n_terms, const uint64_t *post_off,
                                    unsigned long long *counter, KwHit *hits, uint32_t max_hits, hipStream_t s)
{
    if (n_tokens <= 0 || n_terms <= 0) return hipSuccess;
    const int64_t blocks = (n_tokens + 255) / 256;
    const int32_t groups = (n_terms + kMatchGroup - 1) / kMatchGroup;
    if (groups > 65535 || blocks > 0x7FFFFFFF) return hipErrorInvalidValue;
    hipLaunchKernelGGL(vocab_match_short_kernel, dim3((unsigned)blocks, (unsigned)groups), dim3(256), 0, s, vpool, vstart, vlen,
                       n_tokens, terms, n_terms, post_off, counter, hits, max_hits);
    return hipGetLastError();
}

hipError_t launch_expand_hits(const KwHit *hits, const unsigned long long *counter, uint32_t max_hits,
                              const uint32_t *post_rows, uint32_t *bitmaps, int64_t words_per_term, hipStream_t s,
                              unsigned long long *counter_host, const uint8_t *skip_term)
{
    hipLaunchKernelGGL(expand_hits_kernel, dim3(1024), dim3(256), 0, s, hits, counter, max_hits, post_rows, bitmaps,
                       words_per_term, counter_host, skip_term);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K4 + K5a  fused score and per-workgroup selection.  One workgroup scans
// kSelSegRows rows of one query: each wave walks a quarter of them 64 at a time
// (lane = row, coalesced 8-byte reads), skips batches that cannot enter its
// list, and the four lists are merged through LDS at the end.
// ---------------------------------------------------------------------------
// LANEMAX (the sampled prefix of a two-stage pass, whose lists only serve the floor): every lane keeps the best of its U rows
// and the wave's list is ONE sort of those -- the k-th best of such maxima belongs to k distinct rows, so it is a valid floor,
// and with thousands of lanes per query it is the k-th best row's key itself unless two of the k best rows share a lane
// (k^2 / (2 x 13,000) of the time at 10M rows): a quarter of the sorts and none of the merges of the full ranking.
// LANEMAX = 2 (enough segments for 8 k wave maxima per query): nothing is sorted at all -- every wave writes the best of its 256 rows
// into slot `wave` of the segment's list (16 entries per list, the rest unused), and the floor is the k-th best of those
// (select_floor_heads_kernel with 16 entries per list).
template <bool FAST, int LANEMAX>
__global__ __launch_bounds__(1024) void fuse_select_kernel(const double *__restrict__ dot,
                                                           const float *__restrict__ dotf, int64_t dot_stride,
                                                           const double *__restrict__ norm_b,
                                                           const int64_t *__restrict__ created,
                                                           const double2 *__restrict__ row_consts, KwView kw,
                                                           const QueryConst *__restrict__ qcs,
                                                           int64_t now_ticks, int64_t n_rows, int32_t seg_first,
                                                           int32_t n_seg_total,
                                                           const unsigned long long *__restrict__ tau,
                                                           SelEntry *__restrict__ out_sel, I8Prefix i8)
{
    __shared__ SelEntry lists[16][kSelWidth];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int b = blockIdx.x;                        // queries vary fastest: neighbouring workgroups share a segment
    const int seg = seg_first + blockIdx.y;
    const QueryConst qc = qcs[b];
    // Rows after the sampled prefix only matter if they beat the k'-th best key of the prefix
    // (equal keys lose the position tie-break to the prefix rows): most batches are skipped.
    const unsigned long long floor_key = tau ? tau[b] : 0ull;
    const int64_t seg0 = (int64_t)seg * kSelSegRows;
    const int64_t seg1 = (seg0 + kSelSegRows < n_rows) ? seg0 + kSelSegRows : n_rows;
    // int8 prefix: the query's part of the per-pair bound, as the screening GEMM's epilogue forms it
    double i8_qs1 = 0.0, i8_qw = 0.0;
    if (FAST && i8.rowf) {
        i8_qs1 = (double)i8.qs1[b];
        i8_qw = qc.use_cos ? (double)__double2float_ru(0.7 * 1.000001 * sqrt(i8.qerr2[b]) * qc.inv_sqrt_na) : 0.0;
    }

    // each of the 16 waves scores kSelSegRows/16 = 256 rows: 4 batches of 64 whose loads are
    // all issued before the first use
    constexpr int U = kSelSegRows / (16 * 64);
    unsigned long long k = 0ull;
    uint32_t p = 0xFFFFFFFFu;
    unsigned long long nk[U];
    uint32_t np[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t r = seg0 + (int64_t)(wave * U + u) * 64 + lane;
        nk[u] = 0ull;
        np[u] = 0xFFFFFFFFu;
        if (r < seg1) {
            const double d = !qc.use_cos ? 0.0
                             : dotf  ? (double)dotf[(int64_t)b * dot_stride + r]     // K2 candidate pass
                                     : dot[(int64_t)b * dot_stride + r];             // K1e exact
            const uint32_t m = qc.n_terms > 0 ? kw_matches(kw, b, (uint32_t)r) : 0u;
            if (FAST && i8.rowf) {
                const double2 rc = row_consts[r];
                const float4 rf = i8.rowf[r];
                const double bound = qc.use_cos ? (double)rf.y + i8_qw * (double)rf.z : 0.0;
                const double sc = fused_score_fast(d * ((double)rf.x * i8_qs1), rc.x, rc.y, m, qc) - bound;
                nk[u] = score_key(sc);
                if (!(fabs(sc) <= 1.7976931348623157e308)) nk[u] = 1ull;          // non-finite: no floor from it
            } else if (FAST) {
                const double2 rc = row_consts[r];
                nk[u] = score_key(fused_score_fast(d, rc.x, rc.y, m, qc));
            } else {
                nk[u] = score_key(fused_score(d, norm_b[r], created[r], m, qc, now_ticks));
            }
            np[u] = (uint32_t)r;
        }
    }
    if (LANEMAX) {
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (better(nk[u], np[u], k, p)) { k = nk[u]; p = np[u]; }
        if (LANEMAX == 2) {
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                const unsigned long long ok = __shfl_xor(k, d, 64);
                const uint32_t op = __shfl_xor(p, d, 64);
                if (better(ok, op, k, p)) { k = ok; p = op; }
            }
            if (lane == 0) {
                SelEntry e;
                e.key = k; e.pos = p; e.pad = 0;
                out_sel[((int64_t)b * n_seg_total + seg) * kSelWidth + wave] = e;
            }
            return;
        }
        wave_sort(k, p, lane);
    }
#pragma unroll
    for (int u = 0; u < U && LANEMAX == 0; ++u) {
        if (seg0 + (int64_t)(wave * U + u) * 64 >= seg1) break;
        if (floor_key) {
            if (!__any(nk[u] > floor_key)) continue;
            if (nk[u] <= floor_key) { nk[u] = 0ull; np[u] = 0xFFFFFFFFu; }
        }
        const unsigned long long tk = __shfl(k, 63, 64);
        const uint32_t tp = __shfl(p, 63, 64);
        if (!__any(better(nk[u], np[u], tk, tp))) continue;
        wave_sort(nk[u], np[u], lane);
        wave_merge_sorted(k, p, nk[u], np[u], lane);
    }
    lists[wave][lane].key = k;
    lists[wave][lane].pos = p;
    __syncthreads();
#pragma unroll
    for (int stride = 8; stride > 0; stride >>= 1) {
        if (wave < stride) {
            wave_merge_sorted(k, p, lists[wave + stride][lane].key, lists[wave + stride][lane].pos, lane);
            lists[wave][lane].key = k;
            lists[wave][lane].pos = p;
        }
        __syncthreads();
    }
    if (wave == 0) {
        SelEntry e;
        e.key = k; e.pos = p; e.pad = 0;
        out_sel[((int64_t)b * n_seg_total + seg) * kSelWidth + lane] = e;
    }
}

hipError_t launch_fuse_select(const double *dot, const float *dotf, int64_t dot_stride, const double *norm_b,
                              const int64_t *created, const double2 *row_consts, KwView kw,
                              const QueryConst *qc, int64_t now_ticks, int64_t n_rows, int32_t B,
                              int32_t seg_first, int32_t seg_count, const unsigned long long *tau,
                              SelEntry *out_sel, int32_t n_seg_stride, hipStream_t s, I8Prefix i8, int floor_only)
{
    if (n_rows <= 0 || B <= 0 || seg_count <= 0) return hipSuccess;
    const int64_t n_seg = n_seg_stride > 0 ? n_seg_stride : (n_rows + kSelSegRows - 1) / kSelSegRows;
    if (n_seg > 65535) return hipErrorInvalidValue;
    if (floor_only && (tau || !row_consts)) return hipErrorInvalidValue;
    if (floor_only == 2)
        hipLaunchKernelGGL((fuse_select_kernel<true, 2>), dim3((unsigned)B, (unsigned)seg_count), dim3(1024), 0, s, dot, dotf, dot_stride,
                           norm_b, created, row_consts, kw, qc, now_ticks, n_rows, seg_first, (int32_t)n_seg, tau, out_sel, i8);
    else if (floor_only)
        hipLaunchKernelGGL((fuse_select_kernel<true, 1>), dim3((unsigned)B, (unsigned)seg_count), dim3(1024), 0, s, dot, dotf, dot_stride,
                           norm_b, created, row_consts, kw, qc, now_ticks, n_rows, seg_first, (int32_t)n_seg, tau, out_sel, i8);
    else if (row_consts)
        hipLaunchKernelGGL((fuse_select_kernel<true, 0>), dim3((unsigned)B, (unsigned)seg_count), dim3(1024), 0, s, dot, dotf, dot_stride,
                           norm_b, created, row_consts, kw, qc, now_ticks, n_rows, seg_first, (int32_t)n_seg, tau, out_sel, i8);
    else
        hipLaunchKernelGGL((fuse_select_kernel<false, 0>), dim3((unsigned)B, (unsigned)seg_count), dim3(1024), 0, s, dot, dotf, dot_stride,
                           norm_b, created, row_consts, kw, qc, now_ticks, n_rows, seg_first, (int32_t)n_seg, tau, out_sel, I8Prefix());
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K5b  final merge per query: 16 waves fold the n_seg sorted lists, then a
// 4-level tree through LDS; wave 0 writes kprime records and the trailer.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void select_final_kernel(const SelEntry *__restrict__ sel, int32_t n_seg,
                                                            int32_t seg_stride, int32_t kprime, int64_t n_rows, int64_t row_base,
                                                            const double *__restrict__ dot,
                                                            const float *__restrict__ dotf, int64_t dot_stride,
                                                            const double *__restrict__ norm_b,
                                                            const int64_t *__restrict__ created,
                                                            const int64_t *__restrict__ row_ids, KwView kw,
                                                            int32_t dot_exact, double approx_eps,
                                                            unsigned long long *__restrict__ tau_out,
                                                            const uint32_t *__restrict__ fused_cnt, uint32_t fused_cap,
                                                            const double *__restrict__ two_stage_L,
                                                            orr_candidate *__restrict__ out, FloorOut floor)
{
    __shared__ SelEntry lists[16][kSelWidth];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int b = blockIdx.x;
    const SelEntry *mine = sel + (int64_t)b * seg_stride * kSelWidth;

    unsigned long long k = 0ull;
    uint32_t p = 0xFFFFFFFFu;
    constexpr int U = 8;     // lists fetched together
    for (int sg0 = wave; sg0 < n_seg; sg0 += 16 * U) {
        SelEntry e[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int sgi = sg0 + 16 * u;
            e[u].key = 0ull; e[u].pos = 0xFFFFFFFFu;
            if (sgi < n_seg) e[u] = mine[(int64_t)sgi * kSelWidth + lane];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (sg0 + 16 * u >= n_seg) break;
            wave_merge_sorted(k, p, e[u].key, e[u].pos, lane);
        }
    }
    lists[wave][lane].key = k;
    lists[wave][lane].pos = p;
    __syncthreads();
#pragma unroll
    for (int stride = 8; stride > 0; stride >>= 1) {
        if (wave < stride) {
            wave_merge_sorted(k, p, lists[wave + stride][lane].key, lists[wave + stride][lane].pos, lane);
            lists[wave][lane].key = k;
            lists[wave][lane].pos = p;
        }
        __syncthreads();
    }
    if (wave == 0 && tau_out) {      // sampling pass: only the k'-th best key of these lists is wanted
        const unsigned long long kth = __shfl(k, kprime - 1, 64);
        if (lane == 0) {
            tau_out[b] = kth;                     // 0 (empty slot) if fewer than k' rows: no filtering
            if (floor.floor_key) two_stage_floor_of(kth, floor.eps3, floor.eps1, floor.floor_key + b, floor.L + b);
        }
        return;
    }
    if (wave == 0) {
        orr_candidate *o = out + (int64_t)b * (kprime + 1);
        if (lane < kprime)
            write_record(o + lane, k, p, b, row_base, dot, dotf, dot_stride, norm_b, created, row_ids, kw, dot_exact);
        const unsigned long long valid_mask = __ballot(k != 0ull && lane < kprime);
        const int n_valid = __popcll(valid_mask);
        const unsigned long long worst_key = __shfl(k, (n_valid > 0 ? n_valid - 1 : 0), 64);
        if (lane == 0) {
            orr_candidate t;
            // two-stage: every buffered row became a record -> the only rows left out are below L
            const bool kept_all = n_rows <= (int64_t)kprime || (two_stage_L && fused_cnt && fused_cnt[b] <= (uint32_t)kprime);
            t.approx_score = (kept_all || n_valid == 0) ? -__builtin_huge_val() : key_score(worst_key);
            t.dot = approx_eps; t.norm_b = 0.0; t.created_ticks = 0;
            t.row_id = -1; t.order_key = n_rows; t.matches = n_valid; t.flags = ORR_CAND_TRAILER;
            if (fused_cnt && fused_cnt[b] > fused_cap) t.flags |= ORR_CAND_OVERFLOW;
            if (two_stage_L) { t.norm_b = two_stage_L[b]; t.flags |= ORR_CAND_TWO_STAGE; }
            o[kprime] = t;
        }
    }
}

hipError_t launch_select_final(const SelEntry *sel, int32_t n_seg, int32_t B, int32_t kprime,
                               int64_t n_rows, int64_t row_base, const double *dot, const float *dotf,
                               int64_t dot_stride, const double *norm_b, const int64_t *created,
                               const int64_t *row_ids, KwView kw, int32_t dot_exact, double approx_eps,
                               unsigned long long *tau_out, const uint32_t *fused_cnt, uint32_t fused_cap,
                               const double *two_stage_L, orr_candidate *out, hipStream_t s)
{
    if (B <= 0) return hipSuccess;
    if (kprime < 1 || kprime > kSelWidth) return hipErrorInvalidValue;
    hipLaunchKernelGGL(select_final_kernel, dim3((unsigned)B), dim3(1024), 0, s, sel, n_seg, n_seg, kprime, n_rows, row_base,
                       dot, dotf, dot_stride, norm_b, created, row_ids, kw, dot_exact, approx_eps, tau_out, fused_cnt, fused_cap, two_stage_L, out, FloorOut());
    return hipGetLastError();
}

// The floor from MANY lists (a one-query call's sampled stream leaves 128 of them): the k-th best of the lists' HEADS belongs to
// k distinct rows, so it is a valid floor, and it is the k-th best row's key unless two of the k best rows share a list
// (k^2 / (2 lists) of the time: the floor is then the (k+1)-th best).  One wave per query instead of sixteen merging every list.
__global__ __launch_bounds__(64) void select_floor_heads_kernel(const SelEntry *__restrict__ sel, int32_t n_seg, int32_t seg_stride,
                                                                int32_t kprime, unsigned long long *__restrict__ tau_out, FloorOut floor,
                                                                int32_t per_list)
{
    const int lane = threadIdx.x, b = blockIdx.x;
    const SelEntry *mine = sel + (int64_t)b * seg_stride * kSelWidth;
    unsigned long long k = 0ull;
    uint32_t p = 0xFFFFFFFFu;
    // per_list = 1: the heads of sorted lists; 16: the 16 wave maxima a floor-only ranking left in each list's first slots
    for (int i = lane; i < n_seg * per_list; i += 64) {
        const SelEntry e = mine[(int64_t)(i / per_list) * kSelWidth + (i % per_list)];
        if (better(e.key, e.pos, k, p)) { k = e.key; p = e.pos; }
    }
    wave_sort(k, p, lane);
    const unsigned long long kth = __shfl(k, kprime - 1, 64);
    if (lane == 0) {
        tau_out[b] = kth;
        if (floor.floor_key) two_stage_floor_of(kth, floor.eps3, floor.eps1, floor.floor_key + b, floor.L + b);
    }
}

hipError_t launch_select_final_sample(const SelEntry *sel, int32_t n_seg_total, int32_t sample_seg, int32_t B,
                                      int32_t kprime, unsigned long long *tau_out, hipStream_t s, FloorOut floor, int32_t wave_maxima)
{
    if (B <= 0 || sample_seg <= 0) return hipSuccess;
    if (kprime < 1 || kprime > kSelWidth) return hipErrorInvalidValue;
    if (wave_maxima) {                                                 // (the lists hold 16 wave maxima each: fuse_select's LANEMAX = 2)
        if (!floor.floor_key) return hipErrorInvalidValue;
        hipLaunchKernelGGL(select_floor_heads_kernel, dim3((unsigned)B), dim3(64), 0, s, sel, sample_seg, n_seg_total, kprime, tau_out, floor, 16);
        return hipGetLastError();
    }
    static const bool full = getenv("ORR_FLOOR_FULL") != nullptr;      // (A/B)
    if (floor.floor_key && sample_seg >= 8 * kprime && !full) {        // (a two-stage pass's floor: any k distinct rows' k-th key serves)
        hipLaunchKernelGGL(select_floor_heads_kernel, dim3((unsigned)B), dim3(64), 0, s, sel, sample_seg, n_seg_total, kprime, tau_out, floor, 1);
        return hipGetLastError();
    }
    const KwView nokw{nullptr, 0, nullptr, nullptr};
    hipLaunchKernelGGL(select_final_kernel, dim3((unsigned)B), dim3(1024), 0, s, sel, sample_seg, n_seg_total, kprime,
                       (int64_t)0, (int64_t)0, nullptr, nullptr, (int64_t)0, nullptr, nullptr, nullptr, nokw, 0, 0.0,
                       tau_out, nullptr, 0u, nullptr, nullptr, floor);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Generic large-k path: all keys -> stable descending radix sort -> records.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void score_keys_kernel(const double *__restrict__ dot,
                                                         const double *__restrict__ norm_b,
                                                         const int64_t *__restrict__ created, KwView kw, int32_t b,
                                                         QueryConst qc, int64_t now_ticks, int64_t n_rows,
                                                         unsigned long long *__restrict__ keys,
                                                         uint32_t *__restrict__ vals)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x) {
        const double d = qc.use_cos ? dot[r] : 0.0;
        const uint32_t m = qc.n_terms > 0 ? kw_matches(kw, b, (uint32_t)r) : 0u;
        keys[r] = score_key(fused_score(d, norm_b[r], created[r], m, qc, now_ticks));
        vals[r] = (uint32_t)r;
    }
}

hipError_t launch_score_keys(const double *dot, const double *norm_b, const int64_t *created,
                             KwView kw, int32_t b, QueryConst qc, int64_t now_ticks, int64_t n_rows,
                             unsigned long long *keys, uint32_t *vals, hipStream_t s)
{
    if (n_rows <= 0) return hipSuccess;
    int64_t blocks = (n_rows + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(score_keys_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dot, norm_b, created, kw, b, qc,
                       now_ticks, n_rows, keys, vals);
    return hipGetLastError();
}

hipError_t sort_pairs_desc(void *temp, size_t &temp_bytes, const unsigned long long *keys_in,
                           unsigned long long *keys_out, const uint32_t *vals_in, uint32_t *vals_out,
                           int64_t n, hipStream_t s)
{
    return hipcub::DeviceRadixSort::SortPairsDescending(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out,
                                                        (int)n, 0, 64, s);
}

__global__ __launch_bounds__(256) void records_from_sorted_kernel(const unsigned long long *__restrict__ keys,
                                                                  const uint32_t *__restrict__ vals, int32_t K,
                                                                  int64_t n_rows, int64_t row_base,
                                                                  const double *__restrict__ dot, int64_t dot_stride,
                                                                  const double *__restrict__ norm_b,
                                                                  const int64_t *__restrict__ created,
                                                                  const int64_t *__restrict__ row_ids, KwView kw,
                                                                  int32_t b, int32_t dot_exact,
                                                                  orr_candidate *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K) {
        if ((int64_t)i < n_rows)
            write_record(out + i, keys[i], vals[i], b, row_base, dot, nullptr, dot_stride, norm_b, created, row_ids, kw, dot_exact);
        else
            write_record(out + i, 0ull, 0u, b, row_base, dot, nullptr, dot_stride, norm_b, created, row_ids, kw, dot_exact);
    } else if (i == K) {
        orr_candidate t;
        const int64_t n_valid = n_rows < (int64_t)K ? n_rows : (int64_t)K;
        const bool kept_all = n_rows <= (int64_t)K;
        t.approx_score = (kept_all || n_valid == 0) ? -__builtin_huge_val() : key_score(keys[n_valid - 1]);
        t.dot = 0.0; t.norm_b = 0.0; t.created_ticks = 0;
        t.row_id = -1; t.order_key = n_rows; t.matches = (int32_t)n_valid; t.flags = ORR_CAND_TRAILER;
        out[K] = t;
    }
}

hipError_t launch_records_from_sorted(const unsigned long long *keys, const uint32_t *vals, int32_t K,
                                      int64_t n_rows, int64_t row_base, const double *dot, int64_t dot_stride,
                                      const double *norm_b, const int64_t *created, const int64_t *row_ids,
                                      KwView kw, int32_t b, int32_t dot_exact, orr_candidate *out,
                                      hipStream_t s)
{
    const int blocks = (K + 1 + 255) / 256;
    hipLaunchKernelGGL(records_from_sorted_kernel, dim3(blocks), dim3(256), 0, s, keys, vals, K, n_rows, row_base, dot,
                       dot_stride, norm_b, created, row_ids, kw, b, dot_exact, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Seal-time permutation into candidate order.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_f32_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                              const int64_t *__restrict__ perm, int64_t n, int32_t D)
{
    for (int64_t r = blockIdx.x; r < n; r += gridDim.x) {
        const float *s = src + perm[r] * (int64_t)D;
        float *d = dst + r * (int64_t)D;
        for (int i = threadIdx.x; i < D; i += blockDim.x) d[i] = s[i];
    }
}

__global__ __launch_bounds__(256) void gather_i64_kernel(const int64_t *__restrict__ src, int64_t *__restrict__ dst,
                                                         const int64_t *__restrict__ perm, int64_t n)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x)
        dst[r] = src[perm[r]];
}

// dst row r <- src row perm[r] (perm == nullptr: identity).  Source rows are
// (src_start, len) byte ranges; destination rows start at dst_start[r].
__global__ __launch_bounds__(256) void gather_content_kernel(const uint8_t *__restrict__ src_pool,
                                                             const uint64_t *__restrict__ src_start,
                                                             const uint32_t *__restrict__ src_len,
                                                             uint8_t *__restrict__ dst_pool,
                                                             const uint64_t *__restrict__ dst_start,
                                                             const int64_t *__restrict__ perm, int64_t n)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave_id; r < n; r += n_waves) {
        const int64_t sr = perm ? perm[r] : r;
        const uint64_t s0 = src_start[sr], d0 = dst_start[r];
        const uint32_t len = src_len[sr];
        for (uint32_t i = lane; i < len; i += 64) dst_pool[d0 + i] = src_pool[s0 + i];
    }
}

__global__ __launch_bounds__(256) void gather_u32_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst,
                                                         const int64_t *__restrict__ perm, int64_t n)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x)
        dst[r] = src[perm[r]];
}

// ---- deleted rows (orr_index_delete_rows).  A deleted row keeps its position; its norm and timestamp are
// overwritten so that it scores at most 0.2 (keyword part only) and its records are flagged for the host
// finish, which drops them.
__global__ void tombstone_rows_kernel(const int64_t *pos, int32_t n, double *norm_b, int64_t *created)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    norm_b[pos[i]] = 0.0;          // cosine 0 through the norm guard (:84)
    created[pos[i]] = 0;           // year 1: exp(-age / 30) underflows to 0
}

__global__ void mark_dead_records_kernel(orr_candidate *recs, int64_t count, int32_t stride, const int64_t *dead,
                                         int32_t n_dead, int64_t row_base)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count || (int32_t)(i % stride) == stride - 1) return;     // trailers stay as they are
    const int64_t key = recs[i].order_key;
    if (recs[i].row_id < 0 && key < 0) return;
    const int64_t p = key - row_base;
    int32_t lo = 0, hi = n_dead;                                       // dead[] ascending
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (dead[mid] < p) lo = mid + 1; else hi = mid;
    }
    if (lo < n_dead && dead[lo] == p) recs[i].flags |= ORR_CAND_DEAD;
}

__global__ __launch_bounds__(256) void iota_i64_kernel(int64_t *__restrict__ dst, int64_t n, int64_t base)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x)
        dst[r] = base + r;
}

static inline unsigned capped_blocks(int64_t work_items, int per_block)
{
    int64_t b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > 256 * 8) b = 256 * 8;
    return (unsigned)b;
}

hipError_t launch_gather_rows_f32(const float *src, float *dst, const int64_t *perm, int64_t n, int32_t D, hipStream_t s)
{
    if (n <= 0 || D <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_rows_f32_kernel, dim3(capped_blocks(n, 1)), dim3(256), 0, s, src, dst, perm, n, D);
    return hipGetLastError();
}

hipError_t launch_gather_i64(const int64_t *src, int64_t *dst, const int64_t *perm, int64_t n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_i64_kernel, dim3(capped_blocks(n, 256)), dim3(256), 0, s, src, dst, perm, n);
    return hipGetLastError();
}

hipError_t launch_gather_content(const uint8_t *src_pool, const uint64_t *src_start, const uint32_t *src_len,
                                 uint8_t *dst_pool, const uint64_t *dst_start, const int64_t *perm, int64_t n,
                                 hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_content_kernel, dim3(capped_blocks(n, 4)), dim3(256), 0, s, src_pool, src_start, src_len,
                       dst_pool, dst_start, perm, n);
    return hipGetLastError();
}

hipError_t launch_gather_u32(const uint32_t *src, uint32_t *dst, const int64_t *perm, int64_t n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_u32_kernel, dim3(capped_blocks(n, 256)), dim3(256), 0, s, src, dst, perm, n);
    return hipGetLastError();
}

hipError_t launch_iota_i64(int64_t *dst, int64_t n, int64_t base, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(iota_i64_kernel, dim3(capped_blocks(n, 256)), dim3(256), 0, s, dst, n, base);
    return hipGetLastError();
}

hipError_t launch_tombstone_rows(const int64_t *pos, int32_t n, double *norm_b, int64_t *created, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(tombstone_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pos, n, norm_b, created);
    return hipGetLastError();
}

hipError_t launch_mark_dead_records(orr_candidate *recs, int32_t B, int32_t kprime, const int64_t *dead, int32_t n_dead,
                                    int64_t row_base, hipStream_t s)
{
    const int64_t count = (int64_t)B * (kprime + 1);
    if (count <= 0 || n_dead <= 0) return hipSuccess;
    hipLaunchKernelGGL(mark_dead_records_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, recs, count, kprime + 1,
                       dead, n_dead, row_base);
    return hipGetLastError();
}

}  // namespace orr
