// orr_epilogue.h -- scoring epilogue of the batched MFMA kernels (orr_gemm.hip, orr_screen.hip).
#pragma once

#include "orr_kernels.h"
#include "orr_device.h"

#include <type_traits>

#include "orr_screen_tile16_asm.inc"      // ORR_T16_ACC_CLOBBERS

namespace orr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4e __attribute__((ext_vector_type(4)));

// Scoring epilogue shared by the batched kernels.  acc[i][j] are 32 x 32 accumulator tiles of
// v_mfma_f32_32x32x16_bf16 / v_mfma_i32_32x32x32_i8: element e of lane (fr = lane & 31, fh = lane >> 5) belongs
// to query qbase + 32 i + (e & 3) + 8 (e >> 2) + 4 fh and to row colbase + 32 j + fr.
//
// Pass 1 (fp32): an upper bound of the pair's score with the exact keyword term -- match counts (saturating at 15) come
// from the count words, four bits per (query,row), four 32-bit words per (32 queries, row) -- against the query's floor
// minus a margin.  The
// few elements that pass are parked in a per-thread queue in LDS (the operand images are dead by now; the caller has
// put a workgroup barrier in between).
// Pass 2 (one copy of the code): fp64 score from the term bitmaps, compare with the floor key, atomic append to the
// query's buffer.  A non-finite accumulator (fp32 overflow) is never filtered.  A thread whose queue is full (a "hot"
// tile: e.g. the newest rows pass for every query) appends further elements straight to the buffers when RESCORED says
// that every buffer entry gets its exact key afterwards anyway (two-stage pass); otherwise it bumps the query's counter
// past the buffer capacity, which the host treats like any other overflow (the batch is repeated unfused).
//
// What the structure below is shaped by (in-kernel stamps of the int8 screening GEMM at 1M x 3072 rows x 256 queries,
// ORR_SCREEN_STAMPS; a tile's K loop is 77,000 cycles):
//   * with the query constants loaded from global memory inside the element loop, every element ended in a (rarely
//     taken) branch the compiler moves no load across: 64 dependent trips to L2 per tile;
//   * with the loads hoisted but a branch per element kept, a block of 32 queries still took 6,500 cycles (31,000 per
//     tile): each element is a serial chain plus scalar bookkeeping plus the branch, 5.6 cycles per instruction with
//     both waves of the SIMD running.
// Hence: ONE global trip per tile (EpiTileLoads, issued by the caller before its barrier), the tile's query constants
// in LDS (staged by the caller; LDS reads need no vmcnt wait and the compiler pipelines them freely), and pass 1 in
// two forms: 1a BRANCH-FREE over a block of 32 queries, keeping only the wave-wide OR of the answers (the v_cmp
// results OR-ed in scalar registers); 1b, the element-by-element form with the parking code, runs only for the rare
// block in which 1a found something.
constexpr int kEpiQueue = 8;            // parked elements per thread (default; the screening GEMM keeps 3: its LDS belongs to the operand rings)
struct EpiParked { float a; uint32_t idx; };

// Everything pass 1 reads from global memory for one output tile and one lane, as loaded (no arithmetic on it yet, so
// that the loads stay in flight across the caller's barrier): the rows' constants and all their count-plane words.
template <int NI, int NJ>
struct EpiTileLoads {
    double2 rc[NJ];
    float4 rf[NJ];
    uint32_t w[NI][NJ][kCountPlanes];
};

// (Addresses: every load is a wave-uniform base -- the plane of (p, query group), the constants' arrays -- plus the lane's
// row as a 32-bit offset, so the base stays in scalar registers and no lane computes a 64-bit address: with the lanes
// doing that for each of the 18 loads per row tile the requests alone were 3,500 cycles of a one-wave SIMD.  Rows per
// launch < 2^28: the callers' shards are far below.)
template <int NI, int NJ>
__device__ __forceinline__ void epilogue_issue_loads(EpiTileLoads<NI, NJ> &L, int qbase, int64_t colbase, int32_t B, int64_t n_rows,
                                                     const FusedEpilogue &epi, int lane)
{
    const int fr = lane & 31;
    const int32_t n_qg = (B + 31) >> 5;
    // No branch in here: an absent array (no int8 row constants: the bf16 kernels; no count words: a batch without query
    // terms) is replaced by the rows' constants as a readable stand-in and its values by their defaults afterwards.  (As
    // "pointer ? load : default" the compiler built a branch around each load with an s_waitcnt vmcnt(0) inside: the
    // tile's loads went out one row tile at a time, 8,000-10,000 cycles per output tile by the stamps.)
    const bool has_rf = epi.i8_rowf != nullptr, has_cp = epi.count_planes != nullptr;
    const float4 *rowf = has_rf ? epi.i8_rowf : reinterpret_cast<const float4 *>(epi.rowc);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int64_t col = colbase + j * 32 + fr;
        const uint32_t colc = (uint32_t)(col < n_rows ? col : n_rows - 1);     // clamped, never branched around
        L.rc[j] = load_global(epi.rowc, colc);
        const float4 rf = load_global(rowf, colc);
        L.rf[j] = has_rf ? rf : make_float4(1.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int qg = (qbase + i * 32) >> 5;
            const int qgc = qg < n_qg ? qg : n_qg - 1;
#pragma unroll
            for (int p = 0; p < kCountPlanes; ++p) {
                // wave-uniform base (the plane of (p, query group)) + the lane's row
                const gptr<uint32_t> plane = as_global(has_cp ? epi.count_planes + ((int64_t)p * n_qg + qgc) * epi.plane_stride
                                                              : reinterpret_cast<const uint32_t *>(epi.rowc));
                const uint32_t wv = plane[colc];
                L.w[i][j][p] = has_cp ? wv : 0u;
            }
        }
    }
}

// Integer accumulators (ACC = i32x16, the int8 screening GEMM) go in as they are: |I| <= 3072 * 127^2, the conversion
// to fp32 costs at most 2^-24, inside the margin of the int8 row bound (i8_rowf.y).  A coarser gate in front of pass 1
// (two integer thresholds per 32 x 32 block from the extremes of the block's 32 queries, v_max3_i32 over a lane's 16
// dots) was built and measured this round: it closes for 99 % of the blocks of cosine-only batches and for none of a
// hybrid batch (the floor sits within the keyword credit, 0.2, of the bulk), and next to the branch-free pass 1a it
// bought 2 % in the first case and cost 2.5 % in the second; removed.
//
// pre / qf_lds (STAGED): the caller's pre-issued loads and the tile's query constants in LDS, indexed by
// [query - qbase]; otherwise (orr_gemm.hip) everything is loaded here and only pass 1b runs.
// AGPR: the caller's accumulators live in accumulation registers (a wave with 256 of them); each element is then taken
// out with an explicit v_accvgpr_read where it is used -- left to itself the register allocator parks all 256 in
// scratch memory behind the K loop and reads them back one by one.
template <bool AGPR, typename T>
__device__ __forceinline__ float acc_as_float(T v)
{
    if constexpr (AGPR) {
        T r;
        asm("v_accvgpr_read_b32 %0, %1" : "=v"(r) : "a"(v));
        return (float)r;
    } else {
        return (float)v;
    }
}

struct EpiNoHook { __device__ __forceinline__ void operator()(int) const {} };

// between_blocks(i): called once per block of 32 queries, in front of it (the 4-wave screening kernel spreads the requests
// that refill its operand rings over the epilogue this way: issued in one piece they were 3,000 cycles of a CU's address
// path with nothing else going on).
template <int NI, int NJ, bool RESCORED, typename ACC = f32x16, bool STAGED = false, int QDEPTH = kEpiQueue, bool AGPR = false,
          typename HOOK = EpiNoHook>
__device__ __forceinline__ void fused_epilogue(const ACC (&acc)[NI][NJ], int qbase, int64_t colbase, int32_t B,
                                               int64_t n_rows, const FusedEpilogue &epi, int lane, EpiParked *queue,
                                               int queue_stride, uint32_t idx_salt = 0, unsigned long long *st = nullptr,
                                               const EpiTileLoads<NI, NJ> *pre = nullptr, const float4 *qf_lds = nullptr,
                                               HOOK between_blocks = HOOK())
{
    constexpr bool INT_ACC = !__is_same(ACC, f32x16);
    // st (diagnostic, ORR_SCREEN_STAMPS): s_memtime of lane 0 at the phases of this call -- [4] row constants in registers,
    // [5] first block of 32 queries done, [6] pass 1 done (the caller stamps the end of pass 2)
#define ORR_EPI_STAMP(k) if (st && lane == 0) st[k] = __builtin_amdgcn_s_memtime()
    // idx_salt: zero, but opaque to the compiler when the caller runs this inside a loop over output tiles -- it
    // keeps the 128 constant tags of the parked entries from being hoisted out of that loop into registers.
    const int fr = lane & 31, fh = lane >> 5;
    EpiTileLoads<NI, NJ> own;
    if constexpr (!STAGED) epilogue_issue_loads(own, qbase, colbase, B, n_rows, epi, lane);
    const EpiTileLoads<NI, NJ> &L = STAGED ? *pre : own;
    float rb[NJ], cj[NJ], eb[NJ];                                          // cj: the row's share of the bound: recency + (int8) its own quantisation error
    int64_t cols[NJ];
    bool ok[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        cols[j] = colbase + j * 32 + fr;                                   // this lane's row of E
        ok[j] = cols[j] < n_rows;
        rb[j] = (float)L.rc[j].x;
        const float rr = ok[j] ? (float)L.rc[j].y : -__builtin_huge_valf();   // rows past the end never pass
        float ea = 0.f;
        eb[j] = 0.f;
        if (epi.i8_rowf) {                                                 // int8 GEMM: acc is the integer dot
            rb[j] *= L.rf[j].x;
            ea = L.rf[j].y; eb[j] = L.rf[j].z;
        }
        cj[j] = rr + ea;
    }
    if (st) { asm volatile("" :: "v"(cj[0]), "v"(rb[0]), "v"(eb[0])); ORR_EPI_STAMP(4); }
    int parked = 0;
    // the query constants of (block i, element e) for this lane: {0.7/sqrt(normA) [* s1], floor - margin, 0.2/terms, int8: query part of the bound}
    auto qf_of = [&](int i, int e) -> float4 {
        const int qi = qbase + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        if constexpr (STAGED) return qf_lds[i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh];     // (the caller clamped past B when staging)
        else return epi.qf[qi < B ? qi : B - 1];
    };
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        between_blocks(i);
        // Count words (four bits per query, word k = queries 8 k .. 8 k + 7 of the block) shifted so that element e's query
        // sits at the COMPILE-TIME nibble e & 3 of word e >> 2: the count is one v_bfe with immediates.
        uint32_t w[NJ][kCountPlanes];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int p = 0; p < kCountPlanes; ++p) w[j][p] = L.w[i][j][p] >> (16 * fh);
        const int q_left = B - (qbase + i * 32 + 4 * fh);                   // elements with (e & 3) + 8 (e >> 2) >= q_left have no query
        if constexpr (STAGED) {
            // Pass 1a, branch-free: could any of the block's 16 x NJ elements of any lane reach its query's floor?  Nothing but
            // the bound and one compare per element: a row past the end has cj = -inf (its bound is below every floor) and a
            // query past the batch was staged with a floor of +inf.
            unsigned long long wave_any = 0ull;
            // the constants of four queries are requested together and ONE GROUP AHEAD of their use (left to itself the
            // compiler issues each LDS read right in front of its use; with one wave on the SIMD nobody covers that trip)
            float4 qfg[2][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) qfg[0][e] = qf_of(i, e);
#pragma unroll
            for (int e0 = 0; e0 < 16; e0 += 4) {
                if (e0 + 4 < 16) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) qfg[((e0 >> 2) + 1) & 1][e] = qf_of(i, e0 + 4 + e);
                }
                __builtin_amdgcn_sched_barrier(0);
                // (the answers of a group are collected first and OR-ed afterwards: OR-ed one by one, every scalar OR waits for the
                // vector compare in front of it, and one wave has nobody to fill those waits)
                unsigned long long hit[4][NJ];
#pragma unroll
                for (int e1 = 0; e1 < 4; ++e1) {
                    const int e = e0 + e1;
                    const float4 qf = qfg[(e0 >> 2) & 1][e1];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const uint32_t m = (w[j][e >> 2] >> (4 * (e & 3))) & 15u;
                        const float a = acc_as_float<AGPR>(acc[i][j][e]);
                        const float upper = __builtin_fmaf((float)m, qf.z, __builtin_fmaf(a, qf.x * rb[j], __builtin_fmaf(qf.w, eb[j], cj[j])));
                        hit[e1][j] = __builtin_amdgcn_ballot_w64(!(upper < qf.y));      // NaN bounds (float accumulators that overflowed) are kept
                    }
                }
#pragma unroll
                for (int e1 = 0; e1 < 4; ++e1)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) wave_any |= hit[e1][j];
            }
            if (wave_any == 0ull) {
                if (i == 0) { ORR_EPI_STAMP(5); }
                continue;
            }
        }
        // Pass 1b: the same tests element by element, parking what passes.  With staged inputs it only runs for a block in
        // which 1a found something (about one block in fifteen on the bench corpus).
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int cb = (e & 3) + 8 * (e >> 2);
            const int qi = qbase + i * 32 + cb + 4 * fh;
            const float4 qf = qf_of(i, e);
            const bool has_query = cb < q_left;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const uint32_t m = (w[j][e >> 2] >> (4 * (e & 3))) & 15u;
                const float a = acc_as_float<AGPR>(acc[i][j][e]);          // int8 GEMM: |I| <= 3072 * 127^2, the conversion costs at most 2^-24
                // score bound: a (qf.x rb) + recency + keyword credit + (int8) the pair's quantisation bound
                const float upper = __builtin_fmaf((float)m, qf.z, __builtin_fmaf(a, qf.x * rb[j], __builtin_fmaf(qf.w, eb[j], cj[j])));
                // NaN and overflowed sums (float accumulators only) are never dropped here
                bool drop = upper < qf.y;
                if constexpr (!INT_ACC) drop = drop && __builtin_fabsf(a) <= 3.4028234663852886e38f;
                drop = drop || !has_query || !ok[j];
                if (!drop) {
                    if (parked < QDEPTH) {
                        EpiParked pk;
                        pk.a = a; pk.idx = (uint32_t)((i * 16 + e) * NJ + j) + idx_salt;
                        queue[parked * queue_stride] = pk;
                    } else if (RESCORED) {
                        const uint32_t slot = atomicAdd(&epi.cnt[qi], 1u);
                        if (slot < epi.cap) {
                            SelEntry en;
                            en.key = ~0ull; en.pos = (uint32_t)cols[j]; en.pad = 0;   // the exact re-score sets the key
                            epi.buf[(int64_t)qi * epi.cap + slot] = en;
                        }
                    } else {
                        atomicAdd(&epi.cnt[qi], epi.cap + 1u);             // forces the overflow route for this query
                    }
                    ++parked;
                }
            }
        }
        if (i == 0) { ORR_EPI_STAMP(5); }
    }
    ORR_EPI_STAMP(6);
#undef ORR_EPI_STAMP
    if (parked > QDEPTH) parked = QDEPTH;
    for (int s = 0; s < parked; ++s) {
        EpiParked pk = queue[s * queue_stride];
        pk.idx -= idx_salt;
        const int j = (int)(pk.idx % NJ), ie = (int)(pk.idx / NJ), e = ie & 15, i = ie >> 4;
        const int qi = qbase + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        const int64_t col = colbase + j * 32 + fr;
        const QueryConst qc = epi.qc[qi];
        const double2 rc = epi.rowc[col];
        const uint32_t mm = qc.n_terms > 0 ? kw_matches(epi.kw, qi, (uint32_t)col) : 0u;
        double dot = (double)pk.a, bound = 0.0;
        if (epi.i8_rowf) {
            const float4 rf = epi.i8_rowf[col];
            dot *= (double)rf.x * (double)epi.i8_qs1[qi];
            bound = (double)rf.y + (double)epi.qf[qi].w * (double)rf.z;
        }
        unsigned long long key = score_key(fused_score_fast(dot, rc.x, rc.y, mm, qc) + bound);
        if (!(__builtin_fabsf(pk.a) <= 3.4028234663852886e38f) || !(bound <= 1.7976931348623157e308)) key = ~0ull;   // kept whatever the floor: re-scored exactly later
        if (key > epi.tau[qi]) {
            const uint32_t slot = atomicAdd(&epi.cnt[qi], 1u);
            if (slot < epi.cap) {
                SelEntry en;
                en.key = key; en.pos = (uint32_t)col; en.pad = 0;
                epi.buf[(int64_t)qi * epi.cap + slot] = en;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// The same epilogue for 16 x 16 accumulator tiles (v_mfma_i32_16x16x64_i8; the int8 screening GEMM's 128 x 128 wave tile as
// 8 x 8 tiles of four registers).  Element e of lane (c, g; tile16_c_of, tile16_g_of below) of tile (i, j) belongs to query
// qbase + 16 i + 4 g + e and to row colbase + 16 j + c: a lane holds 8 rows x 32 queries (the 32 x 32 form: 4 x 64).
// A block of 32 queries is the tile pair (2 b, 2 b + 1); within it the lane's queries are 16 t + 4 g + e (t = 0, 1), whose
// counts sit in count word 2 t + (g >> 1) at nibble 4 (g & 1) + e -- so a lane needs two of a row's four words per block,
// shifted by 16 (g & 1) once, and the nibble index is the compile-time e.  Everything else is fused_epilogue's.
// ---------------------------------------------------------------------------
// Which 16 rows a fragment holds: fragment row m of a 16-row tile is the tile's row tile16_row(m) -- the row quads 1 and 3
// change places.  With the stored swizzle (chunk c of row r in slot c ^ ((r >> 2) & 3), made for 32-row fragments) the
// straight order puts rows m and m + 4 of a ds_read_b128 lane group on the same banks (2-way: half of the LDS array's cycles
// by the counters); in this order every group covers the 64 banks once.  Queries and rows are read the same way, so
// accumulator element e of lane (c, g) belongs to query 16 i + 4 ((-g) & 3) + e and row 16 j + tile16_row(c).
__device__ __forceinline__ int tile16_row(int m) { return (((0 - (m >> 2)) & 3) << 2) | (m & 3); }
__device__ __forceinline__ int tile16_c_of(int lane) { return tile16_row(lane & 15); }
__device__ __forceinline__ int tile16_g_of(int lane) { return (0 - (lane >> 4)) & 3; }

struct EpiTileLoads16 {
    double2 rc[8];
    float4 rf[8];
    uint32_t w[4][8][2];           // this lane's count words: [block of 32 queries][row tile][t]
};

// Word offsets of this lane's count words inside a plane pair: its row (clamped) plus, where g >> 1 is set, the distance to the
// next plane (word 2 t + 1 instead of 2 t).  32 bits: the launcher sends shards whose planes exceed 2^30 words elsewhere.
// (BITS = 2: two-bit count words -- word t of a block holds queries 16 t .. 16 t + 15, so every lane reads words 0 and 1 and its
// four counts of tile t sit in byte g of word t; no half-plane offset.)
template <int BITS = 4>
__device__ __forceinline__ void epilogue_word_offsets16(uint32_t (&at)[8], int64_t colbase, int32_t B, int64_t n_rows, const FusedEpilogue &epi,
                                                        int lane)
{
    const int c = tile16_c_of(lane), g = tile16_g_of(lane);
    const int32_t n_qg = (B + 31) >> 5;
    const uint32_t plane_dist = epi.count_planes && BITS == 4 ? (uint32_t)((int64_t)n_qg * epi.plane_stride) : 0u;
    // (32-bit arithmetic throughout: the launcher sends shards of 2^28 rows and more elsewhere)
    const uint32_t col0 = (uint32_t)colbase + (uint32_t)c, last = (uint32_t)(n_rows - 1), half = (uint32_t)(g >> 1) * plane_dist;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t col = col0 + 16u * j;
        at[j] = (col < last ? col : last) + half;                                                    // clamped, never branched around
    }
}

// this lane's two count words per row tile for block b of 32 queries, already shifted to its half (nibble e = query 4 g + e)
template <int BITS = 4>
__device__ __forceinline__ void epilogue_load_words16(uint32_t (&w)[8][2], const uint32_t (&at)[8], int b, int qbase, int32_t B,
                                                      const FusedEpilogue &epi)
{
    const int32_t n_qg = (B + 31) >> 5;
    const bool has_cp = epi.count_planes != nullptr;
    const int qg = (qbase >> 5) + b;
    const int qgc = qg < n_qg ? qg : n_qg - 1;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        // the plane of word 2 t of (query group): a wave-uniform base; the lane's offset picks word 2 t or 2 t + 1
        const uint32_t *plane = has_cp ? epi.count_planes + ((int64_t)(BITS == 2 ? t : 2 * t) * n_qg + qgc) * epi.plane_stride
                                       : reinterpret_cast<const uint32_t *>(epi.rowc);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t wv = load_global_at<uint32_t>(plane, at[j] * 4u);
            w[j][t] = has_cp ? wv : 0u;
        }
    }
}

template <int BITS = 4>
__device__ __forceinline__ void epilogue_issue_loads16(EpiTileLoads16 &L, int qbase, int64_t colbase, int32_t B, int64_t n_rows,
                                                       const FusedEpilogue &epi, int lane)
{
    const uint32_t col0 = (uint32_t)colbase + (uint32_t)tile16_c_of(lane), last = (uint32_t)(n_rows - 1);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t col = col0 + 16u * j;
        const uint32_t off = (col < last ? col : last) * 16u;                  // clamped, never branched around; rows < 2^28
        L.rc[j] = load_global_at_d2(epi.rowc, off);
        L.rf[j] = load_global_at_f4(epi.i8_rowf, off);
    }
    uint32_t at[8];
    epilogue_word_offsets16<BITS>(at, colbase, B, n_rows, epi, lane);
    epilogue_load_words16<BITS>(L.w[0], at, 0, qbase, B, epi);
}

// the count words of the blocks 1..3 (behind the K loop: beside the fragments their 48 landing registers did not fit)
template <int BITS = 4>
__device__ __forceinline__ void epilogue_issue_later_words16(EpiTileLoads16 &L, int qbase, int64_t colbase, int32_t B, int64_t n_rows,
                                                             const FusedEpilogue &epi, int lane)
{
    uint32_t at[8];
    epilogue_word_offsets16<BITS>(at, colbase, B, n_rows, epi, lane);
#pragma unroll
    for (int b = 1; b < 4; ++b) epilogue_load_words16<BITS>(L.w[b], at, b, qbase, B, epi);
}

// Registers: the 256 accumulators fill the accumulation file, so everything here has to fit the 256 vector registers with
// room to spare -- a spilled value comes back through scratch memory behind an s_waitcnt vmcnt(0), which in this kernel also
// waits for the ring refill in flight.  Hence: three floats per row, one block's count words plus the next block's in
// flight, the query constants read from LDS one query ahead, 32-bit offsets.
//
// The accumulators are not C++ objects here: the K loop is assembler text on fixed registers (tile (i, j), element e =
// a[4 (8 i + j) + e], orr_screen_tile16_asm.inc), and this epilogue reads them by NUMBER (v_accvgpr_read_b32 vN, aM with M a
// template constant) -- so every loop over tiles and elements is a compile-time loop.
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// (Not volatile -- the compiler may pair and schedule these freely -- but each takes `token`, a value the K loop's last
// statement produces: that is the only ordering they need.  a[N]: the immediate prints in hex from 64 up, which the
// bracket form takes.)
template <int IDX>
__device__ __forceinline__ float acc16_as_float(int token)
{
    int r;
    // (the clobber list: on gfx950 the allocator treats accumulation registers as ordinary ones and, under pressure, parks
    // vector registers there -- over the accumulators it does not know about.  Declaring every read a clobber of all of
    // them leaves it nothing to keep there across the epilogue.)
    asm("v_accvgpr_read_b32 %0, a[%1]" : "=v"(r) : "n"(IDX), "v"(token) : ORR_T16_ACC_CLOBBERS);
    return (float)r;
}

// the same, volatile: for the rare pass 1b -- a non-volatile asm has no side effects the compiler knows of, so it hoisted all
// 256 reads of pass 1b (and what hangs on them) out of the `if (block flagged)` around them: 7,000 cycles per tile, always
template <int IDX>
__device__ __forceinline__ float acc16_as_float_here(int token)
{
    int r;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(r) : "n"(IDX), "v"(token) : ORR_T16_ACC_CLOBBERS);
    return (float)r;
}

// the raw accumulator (diagnostic dots form of the kernel)
template <int IDX>
__device__ __forceinline__ int acc16_as_int_here(int token)
{
    int r;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(r) : "n"(IDX), "v"(token) : ORR_T16_ACC_CLOBBERS);
    return r;
}

typedef float f32x2e __attribute__((ext_vector_type(2)));

// byte E of the pair (x, y) = (even nibbles, odd nibbles of a count word's low half, one per byte): nibble e of the word
template <int E>
__device__ __forceinline__ float count_of_nibble(uint32_t x, uint32_t y)
{
    const uint32_t v = (E & 1) ? y : x;
    float r;
    if constexpr ((E >> 1) == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(r) : "v"(v));
    else asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

template <int BYTE>
__device__ __forceinline__ float ubyte_as_float(uint32_t v)
{
    float r;
    if constexpr (BYTE == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(r) : "v"(v));
    else if constexpr (BYTE == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(r) : "v"(v));
    else if constexpr (BYTE == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(r) : "v"(v));
    else asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

// Pass 1 of this epilogue, round 3.  The test of one (query, row) pair is
//     upper = a (qx rb_j) + m qz + cjw_j  >=  qy            cjw_j = recency_j + row bound_j + QW eb_j
// with QW = the LARGEST query-side quantisation term of the batch's (finite) queries instead of the pair's own (eb_j >= 0, so
// the bound only grows -- by (QW - qw_q) eb_j, nothing for queries quantised alike; pass 2 keeps the pair's own bound in fp64).
// That makes cjw a per-ROW constant, and the pair's test three packed instructions for two rows (v_pk_mul_f32, two
// v_pk_fma_f32) plus its share of a v_pk_add_f32 (- qy) and a v_max3_f32 that folds two differences into the running maximum
// of the group (8 rows x one query per lane): no compare and no scalar OR per pair -- one compare per group.  The match count
// of a pair comes out of its count word with v_cvt_f32_ubyteN after the word's nibbles were spread to bytes once for four
// queries.  Round 2's form: 9.3 vector + 1 scalar instruction per pair (3,800 in all per wave and tile); this one: 6.4.
//
// NaN: v_max3_f32 drops a NaN operand, so no NaN may stand for "keep".  The inputs are made safe instead: a row whose
// constants are not finite (embedding with a non-finite value: "never screened out") enters as rb = 0, cjw = +inf (every
// pair passes), a row past the end as cjw = -inf; launch_fused_query_consts hands over finite qx, qz (and QW) and turns a
// query with anything non-finite into qx = qz = 0, qy = -inf (every pair passes); a slot without a query has qy = +inf.
// a is an integer dot (|a| <= 3072 * 127^2), rb <= 1e30 and qx <= 1: no product overflows; inf - inf only arises for pairs
// that are to be dropped (row past the end and keep-everything query; no query and keep-everything row).
// BITS = 2: two-bit count words (a batch whose queries all have at most three terms): half the words to load, the lane's four
// counts of a tile are one byte of the word, spread to four bytes with one multiplication.
template <int QDEPTH, typename HOOK = EpiNoHook, int BITS = 4>
__device__ __forceinline__ void fused_epilogue16(int acc_token, int qbase, int64_t colbase, int32_t B, int64_t n_rows,
                                                 const FusedEpilogue &epi, int lane, EpiParked *queue, int queue_stride, uint32_t idx_salt,
                                                 unsigned long long *st, const EpiTileLoads16 &L, const float4 *qf_lds,
                                                 HOOK between_blocks = HOOK())
{
#define ORR_EPI_STAMP(k) if (st && lane == 0) st[k] = __builtin_amdgcn_s_memtime()
    const int c = tile16_c_of(lane), g = tile16_g_of(lane);
    const int sh = 16 * (g & 1);
    const float qw_max = qf_lds[0].w;                                       // (the staged constants carry QW in every .w: launch_fused_query_consts)
    f32x2e rb2[4], cjw2[4];
    const int64_t col0 = colbase + c;                                      // this lane's row of row tile 0 (tile j: + 16 j)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float rbv = (float)L.rc[j].x * L.rf[j].x;                     // the accumulator is the integer dot
        const float cj = (float)L.rc[j].y + L.rf[j].y, eb = L.rf[j].z;
        const bool keep_all = !(rbv <= 1e30f) || !(eb <= 1e30f) || !(cj <= 3.0e38f) || !(cj >= -3.0e38f);     // (also true for NaNs)
        const float cjw = keep_all ? __builtin_huge_valf() : __builtin_fmaf(qw_max, eb, cj);
        rb2[j >> 1][j & 1] = keep_all ? 0.f : rbv;
        cjw2[j >> 1][j & 1] = col0 + j * 16 < n_rows ? cjw : -__builtin_huge_valf();   // rows past the end never pass
    }
    if (st) { asm volatile("" :: "v"(cjw2[0][0]), "v"(rb2[0][0])); ORR_EPI_STAMP(4); }
    int parked = 0;
    const float4 *qf_lane = qf_lds + 4 * g;                                // query 16 t + 4 g + e of block b: qf_lane[32 b + 16 t + e]
    // Pass 1a for all four blocks first, pass 1b (rare, and twenty times the code) for the flagged groups behind them: with
    // 1b's code between the blocks' 1a, every block began with a jump over 21 KiB and a cold instruction cache (6,500 cycles
    // per output tile by the stamps; the accumulators stay where they are, so 1b can read them again later).
    // What 1a hands to 1b: one bit per (block, group of 8 rows x one query per lane) -- 1b then runs for the flagged GROUPS only
    // (a flagged block is nearly always one element: with the whole block re-tested, 256 tests per lane, the wave that had it
    // came 10,000 cycles late to the next barrier, and two tiles in three have such a wave).
    uint32_t group_flags = 0u;
    static_for<4>([&](auto b_c) {
        constexpr int b = decltype(b_c)::value;
        between_blocks(b);
        static_for<2>([&](auto t_c) {
            constexpr int t = decltype(t_c)::value;
            // this lane's four queries of tile (2 b + t), splatted for the packed instructions
            f32x2e qx2[4], qz2[4], nqy2[4];
            float mx[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float4 qf = qf_lane[32 * b + 16 * t + e];
                qx2[e] = f32x2e{qf.x, qf.x}; qz2[e] = f32x2e{qf.z, qf.z}; nqy2[e] = f32x2e{-qf.y, -qf.y};
                mx[e] = -__builtin_huge_valf();
            }
            static_for<4>([&](auto jp_c) {
                constexpr int jp = decltype(jp_c)::value, j0 = 2 * jp;
                // the count words of rows j0, j0 + 1: nibble e -> byte (e >> 1) of x (even e) or y (odd e); two-bit words: field e -> byte e of x
                uint32_t x0, y0, x1, y1;
                if constexpr (BITS == 2) {
                    x0 = (__builtin_amdgcn_ubfe(L.w[b][j0][t], 8u * (uint32_t)g, 8u) * 0x41041u) & 0x03030303u;
                    x1 = (__builtin_amdgcn_ubfe(L.w[b][j0 + 1][t], 8u * (uint32_t)g, 8u) * 0x41041u) & 0x03030303u;
                    y0 = y1 = 0u;
                } else {
                    const uint32_t w0 = L.w[b][j0][t] >> sh, w1 = L.w[b][j0 + 1][t] >> sh;
                    x0 = w0 & 0x0F0F0F0Fu; y0 = (w0 >> 4) & 0x0F0F0F0Fu; x1 = w1 & 0x0F0F0F0Fu; y1 = (w1 >> 4) & 0x0F0F0F0Fu;
                }
                static_for<4>([&](auto e_c) {
                    constexpr int e = decltype(e_c)::value;
                    const f32x2e a2 = {acc16_as_float<4 * (8 * (2 * b + t) + j0) + e>(acc_token), acc16_as_float<4 * (8 * (2 * b + t) + j0 + 1) + e>(acc_token)};
                    f32x2e m2;
                    if constexpr (BITS == 2) m2 = f32x2e{ubyte_as_float<e>(x0), ubyte_as_float<e>(x1)};
                    else m2 = f32x2e{count_of_nibble<e>(x0, y0), count_of_nibble<e>(x1, y1)};
                    const f32x2e t1 = a2 * rb2[jp];
                    const f32x2e t2 = __builtin_elementwise_fma(t1, qx2[e], cjw2[jp]);
                    const f32x2e t3 = __builtin_elementwise_fma(m2, qz2[e], t2);
                    const f32x2e d = t3 + nqy2[e];
                    mx[e] = __builtin_fmaxf(__builtin_fmaxf(mx[e], d[0]), d[1]);
                });
            });
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned long long group_any = __builtin_amdgcn_ballot_w64(mx[e] >= 0.f);
                group_flags |= group_any != 0ull ? 1u << (8 * b + 4 * t + e) : 0u;
            }
            // (pinned: the flags are only looked at behind all four blocks; left alone the compiler postpones the ORs)
            asm volatile("" :: "s"(group_flags));
        });
        if constexpr (b == 0) { ORR_EPI_STAMP(5); }
    });
    int token_1b = acc_token;
    asm volatile("" : "+v"(token_1b));            // (its own token: pass 1b reads the accumulators again instead of keeping 1a's 256 values)
    static_for<4>([&](auto b_c) {
        constexpr int b = decltype(b_c)::value;
        if (((group_flags >> (8 * b)) & 0xffu) != 0u) {
            // Pass 1b: the same test element by element, parking what passes (about one block in fifteen, one group of it).
            // Its count words come from memory again: kept in registers since pass 1a they were 64 more live values for a rare
            // path.
            uint32_t at1[8], w1[8][2];
            epilogue_word_offsets16<BITS>(at1, colbase, B, n_rows, epi, lane);
            epilogue_load_words16<BITS>(w1, at1, b, qbase, B, epi);
            static_for<8>([&](auto te_c) {
                constexpr int te = decltype(te_c)::value, t = te >> 2, e = te & 3, i = 2 * b + t;
                if ((group_flags & (1u << (8 * b + te))) == 0u) return;
                const int qi = qbase + 16 * i + 4 * g + e;
                const float4 q1 = qf_lane[32 * b + 16 * t + e];
                const bool has_query = qi < B;
                static_for<8>([&](auto j_c) {
                    constexpr int j = decltype(j_c)::value;
                    const uint32_t m = BITS == 2 ? (w1[j][t] >> (8 * g + 2 * e)) & 3u : ((w1[j][t] >> sh) >> (4 * e)) & 15u;
                    const float a = acc16_as_float_here<4 * (8 * i + j) + e>(token_1b);
                    const float upper = __builtin_fmaf((float)m, q1.z, __builtin_fmaf(a * rb2[j >> 1][j & 1], q1.x, cjw2[j >> 1][j & 1]));
                    const bool drop = upper < q1.y || !has_query || !(col0 + j * 16 < n_rows);
                    if (!drop) {
                        if (parked < QDEPTH) {
                            EpiParked pk;
                            pk.a = a; pk.idx = (uint32_t)((i * 4 + e) * 8 + j) + idx_salt;
                            queue[parked * queue_stride] = pk;
                        } else {                                            // every buffer entry gets its exact key afterwards anyway
                            const uint32_t slot = atomicAdd(&epi.cnt[qi], 1u);
                            if (slot < epi.cap) {
                                SelEntry en;
                                en.key = ~0ull; en.pos = (uint32_t)(col0 + j * 16); en.pad = 0;
                                epi.buf[(int64_t)qi * epi.cap + slot] = en;
                            }
                        }
                        ++parked;
                    }
                });
            });
        }
    });
    ORR_EPI_STAMP(6);
#undef ORR_EPI_STAMP
    if (parked > QDEPTH) parked = QDEPTH;
    for (int s = 0; s < parked; ++s) {
        EpiParked pk = queue[s * queue_stride];
        pk.idx -= idx_salt;
        const int j = (int)(pk.idx & 7u), ie = (int)(pk.idx >> 3), e = ie & 3, i = ie >> 2;
        const int qi = qbase + 16 * i + 4 * g + e;
        const int64_t col = colbase + j * 16 + c;
        const QueryConst qc = epi.qc[qi];
        const double2 rc = epi.rowc[col];
        const uint32_t mm = qc.n_terms > 0 ? kw_matches(epi.kw, qi, (uint32_t)col) : 0u;
        const float4 rf = epi.i8_rowf[col];
        const double dot = (double)pk.a * ((double)rf.x * (double)epi.i8_qs1[qi]);
        const double bound = (double)rf.y + (double)epi.qf[qi].w * (double)rf.z;
        unsigned long long key = score_key(fused_score_fast(dot, rc.x, rc.y, mm, qc) + bound);
        if (!(bound <= 1.7976931348623157e308)) key = ~0ull;               // kept whatever the floor: re-scored exactly later
        if (key > epi.tau[qi]) {
            const uint32_t slot = atomicAdd(&epi.cnt[qi], 1u);
            if (slot < epi.cap) {
                SelEntry en;
                en.key = key; en.pos = (uint32_t)col; en.pad = 0;
                epi.buf[(int64_t)qi * epi.cap + slot] = en;
            }
        }
    }
}

}  // namespace orr
