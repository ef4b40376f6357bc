// orr_screen.hip -- K2c: the wide first stage of the two-stage batched pass.
//
// S[b][r] = sum_k hi(Q[b][k]) * hi(E[r][k]) with hi(x) = bf16(x), from a bf16 SHADOW of the
// embedding matrix (built once per sealed shard, half the bytes of the fp32 master) and the hi
// halves of the pre-split queries.  bf16 x bf16 products are exact in the fp32 accumulator, so
//   |S - sum q_k e_k| <= [2^-7 (1 + 2^-9) + 1.02 D 2^-23] sum|q_k e_k|
// (two bf16 roundings per product, u = 2^-8; every fp32 addition charged one full ulp).  The
// pass never decides a result: it only drops (query,row) pairs whose score cannot reach a lower
// bound of the query's k-th best score; everything it keeps is re-scored in the reference's own
// arithmetic (RecallSearchService.cs:77-87) by rescore_buffer_exact (orr_gemm.hip).
//
// Two forms of the same 256 x 256 tile live here: the eight-wave one described next (up to 64 queries, the unfused
// prefix product, the bf16 shadow) and, further down, the four-wave one (screen_tile4_kernel: one wave per SIMD, from 65
// queries on the int8 shadow, where the matrix cores carry the launch).
//
// Structure (gfx950): one 256 (queries) x 256 (rows) tile per workgroup, 8 waves as 2 x 4, each
// wave 128 x 64 = 4 x 2 accumulator tiles of v_mfma_f32_32x32x16_bf16 (128 accumulator VGPRs).
// Both operands are bf16 with k contiguous, so both go global -> LDS with
// global_load_lds_dwordx4 (no VGPR staging, no ds_write) into a ring of five 32 KiB stages
// (all 160 KiB of LDS), one K-tile of 32 per stage.  One workgroup barrier per K-tile; the
// LDS-DMA pieces of three K-tiles (96 KiB) stay in flight across it behind a counted
// s_waitcnt vmcnt -- with one 64 KiB tile in flight the loads of a CU were latency-bound at
// 29 GB/s, HBM stream and L2-resident queries together.  The fragments of K-tile t+1 are read
// into a second register set while K-tile t is multiplied, so no MFMA waits for LDS.
//
// Memory layout: the shadow (and the hi halves of the queries) are stored TILED in exactly the
// order of the LDS image -- [row tile of 256][K-tile of 32][256 rows][32 k] bf16, 16 KiB per
// (row tile, K-tile) -- so a K-tile of an operand is ONE contiguous 16 KiB run of HBM and every
// LDS-DMA piece (1 KiB per wave instruction, lane-linear) reads 1 KiB of consecutive addresses.
// Within a row's 64 bytes the 16-byte chunk c sits in slot c ^ ((row >> 2) & 3): that makes the
// ds_read_b128 of an A/B fragment (32 rows x one chunk) conflict-free for the instruction's
// four 16-lane groups (MI355X_MICROARCH.md, LDS), and because the swizzle is part of the stored
// layout the copy stays linear.  Rows past the end of a tile are zero.  Workgroups that share a
// row tile (B > 256) sit back to back on one XCD, so later reads of the tile come from that
// XCD's L2.
#include "orr_kernels.h"
#include "orr_device.h"
#include "orr_epilogue.h"

#include <algorithm>
#include <cstdio>
#include <vector>
#include <type_traits>
#include <cstdlib>

namespace orr {

namespace {

constexpr int kScBM = 256, kScBN = 256, kScBK = 32;
constexpr int kScImage = kScBM * kScBK * 2;             // bytes of one operand image of one K-tile (16 KiB)
// LDS: a ring of kScNA query images, a ring of kScNB row images, and 32 KiB that belong to the epilogue.  The two rings
// are separate so that their depths can differ (the rows come from HBM, the queries from L2) and so that a wave
// requests from ONE ring only (see the kernel).  Measured at 1M x 3072 rows x 256 queries (ORR_SC_NA / ORR_SC_NB builds
// on one box): row depth 4 = 5 = 6 stages (0.850 / 0.855 / 0.850 ms), query depth 2 instead of 3: 0.905 ms -- the K
// loop is not waiting for memory; what bounds it is the waves' own instruction streams around the one barrier per
// K-tile (in-kernel stamps: 1,560 cycles per K-tile against 1,024 of MFMA issue; the first form of the new request
// code, ~45 scalar instructions per piece, took 2,240).
#ifndef ORR_SC_NA
#define ORR_SC_NA 3
#define ORR_SC_NB 5
#endif
constexpr int kScNA = ORR_SC_NA, kScNB = ORR_SC_NB;
constexpr int kScEpiBytes = 32768;                      // parking queue (7 entries x 8 B x 512 threads) + the tile's query constants (4 KiB)
constexpr int kScEpiQueue = 7;
constexpr int kScLds = (kScNA + kScNB) * kScImage + kScEpiBytes;       // 160 KiB
static_assert(kScLds <= 160 * 1024, "the rings and the epilogue's region must fit a CU's LDS");

typedef int i32x4v __attribute__((ext_vector_type(4)));
typedef int i32x16v __attribute__((ext_vector_type(16)));

// I8 = true (K2j): the same kernel on the int8 shadow and int8 queries -- an image row's 64 bytes are 64 k,
// v_mfma_i32_32x32x32_i8 consumes the same 16-byte fragments at twice the rate, the accumulator is the exact
// integer dot, half as many K-tiles and half the bytes per row.
//
// LIVE = 32-query tiles of the 256-query tile that hold queries (8: all).  Batches of at most 128 queries only
// fill the first LIVE <= 4 of them: waves 4..7 (query rows 128..255) then only keep requesting row pieces,
// waves 0..3 multiply their first LIVE query tiles, the upper half of the query image is never requested
// (LIVE <= 2: nor its second quarter) -- the kernel turns from latency- into HBM-bound.
//
// Who requests what: waves 0..3 the query image (4 / 2 / 1 pieces of 1 KiB each per K-tile for LIVE = 8 / 4 / <= 2),
// waves 4..7 the row image (4 pieces each).  s_waitcnt vmcnt counts a wave's requests in issue order, so a wave that
// requested both could not keep five row tiles in flight behind two query tiles; with the roles split every wave
// waits for exactly the oldest tile of its own ring.
template <bool FUSED, bool I8, int LIVE = 8>
__global__ __launch_bounds__(512, 1) void screen_bf16_kernel(const __bf16 *__restrict__ Qh, int32_t B,
                                                             const __bf16 *__restrict__ Eh, int64_t row_first, int64_t n_rows,
                                                             int32_t D, float *__restrict__ S, int64_t s_stride,
                                                             int32_t n_ntiles, int32_t n_mtiles, int32_t flags, FusedEpilogue epi)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char *const lds_a = lds, *const lds_b = lds + kScNA * kScImage, *const lds_epi = lds + (kScNA + kScNB) * kScImage;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    constexpr int LI = LIVE >= 4 ? 4 : LIVE;                               // query tiles a multiplying wave owns
    const bool active = LIVE == 8 || wr == 0;
    const bool row_loader = wr != 0;                                        // waves 4..7 request rows, waves 0..3 queries
    constexpr int kPA = LIVE == 8 ? 4 : LIVE == 4 ? 2 : 1;                  // query pieces per K-tile and requesting wave
    constexpr int kPB = 4;                                                  // row pieces per K-tile and requesting wave
    // PERSISTENT: a workgroup walks the output tiles id = blockIdx.x, + gridDim.x, ... (the launcher starts one per
    // CU).  The rings run on across output tiles: while the last K-tiles of one are multiplied, the requests fetch
    // the first K-tiles of the next, which then arrive during the epilogue.
    // ids that differ by 8 share an XCD; the query tiles of one row tile are consecutive there
    const int total_ids = ((n_ntiles + 7) / 8) * 8 * n_mtiles;
    auto valid_from = [&](int id) {                                         // first id >= the given one (in this workgroup's walk) with a row tile
        while (id < total_ids && ((id >> 3) / n_mtiles) * 8 + (id & 7) >= n_ntiles) id += gridDim.x;
        return id;
    };
    const int T = I8 ? D / 64 : D / kScBK;
    const bool stream_rows = n_mtiles == 1 && (flags & 1);
    const int fr = lane & 31, fh = lane >> 5;
    // ---- this wave's request stream.  It runs AHEAD of the multiplication by the depth of its ring and on across output
    // tiles, so it keeps its own position: (s_id, s_k) = output tile and K-tile requested next, s_src = global address of
    // this wave's part of that K-tile's image (rows: KiB 4 w .. 4 w + 3 of the row image; queries: kPA KiB of the query
    // image), s_lds = LDS byte address of the same part in the ring stage it goes to.  A request is three instructions
    // (M0, a wait state, global_load_lds with the piece's KiB as the immediate offset, which applies to both addresses);
    // advancing the stream a handful of scalar ones.  (The first form of this loop recomputed stage, source and the
    // end-of-tile cases per piece, ~45 scalar instructions each: the K loop was then bound by the waves' own
    // instruction streams, 2,240 cycles per K-tile however the memory behaved.)
    const uint32_t lane_off = (uint32_t)(lane * 16);
    const uint32_t part_off = row_loader ? (uint32_t)((wave & 3) * kPB * 1024) : (uint32_t)(wave * kPA * 1024);
    auto stream_src = [&](int sid) -> const unsigned char * {
        const int smt = (sid >> 3) % n_mtiles, snt = ((sid >> 3) / n_mtiles) * 8 + (sid & 7);
        return row_loader ? reinterpret_cast<const unsigned char *>(Eh) + ((int64_t)(row_first / kScBN + snt) * T) * kScImage + part_off
                          : reinterpret_cast<const unsigned char *>(Qh) + ((int64_t)smt * T) * kScImage + part_off;
    };
    int s_id = valid_from(blockIdx.x), s_k = 0, s_stage = 0;
    const unsigned char *s_src = s_id < total_ids ? stream_src(s_id) : nullptr;
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(row_loader ? lds_b : lds_a) + part_off;     // LDS byte address of this wave's part in stage 0
    const int ring_n = row_loader ? kScNB : kScNA;
    // piece `slot` of the stream's current K-tile
    auto issue_piece = [&](int slot) {
        if (!row_loader && slot >= kPA) return;
        const uint32_t m0v = ring_lds + (uint32_t)s_stage * kScImage;
        const uint64_t src = (uint64_t)(uintptr_t)s_src;
#define ORR_GLDS(OFF, POLICY) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:" #OFF POLICY \
                                           :: "v"(lane_off), "s"(src), "s"(m0v) : "memory")
        // (M0 is not on the clobber list: the compiler reserves it and loads it itself right before each of its own uses.)
        // rows are streamed once when the whole batch fits one query tile: non-temporal, so they do not push
        // the query images (re-read by every workgroup) out of L2
        if (row_loader && stream_rows) {
            switch (slot) { case 0: ORR_GLDS(0, " nt"); break; case 1: ORR_GLDS(1024, " nt"); break; case 2: ORR_GLDS(2048, " nt"); break; default: ORR_GLDS(3072, " nt"); }
        } else {
            switch (slot) { case 0: ORR_GLDS(0, ""); break; case 1: ORR_GLDS(1024, ""); break; case 2: ORR_GLDS(2048, ""); break; default: ORR_GLDS(3072, ""); }
        }
#undef ORR_GLDS
    };
    // the stream moves on to its next K-tile (past the last output tile it keeps re-requesting the last K-tile: spare copies
    // into stages nobody reads any more)
    auto advance_stream = [&]() {
        s_stage = s_stage + 1 == ring_n ? 0 : s_stage + 1;
        if (s_k + 1 < T) { ++s_k; s_src += kScImage; return; }
        const int nid = valid_from(s_id + gridDim.x);
        if (nid < total_ids) { s_id = nid; s_k = 0; s_src = stream_src(nid); }
    };
    // until this wave's pieces of the tiles that may still be in flight behind the awaited one are all that is outstanding
    auto await_own = [&](bool first_of_tile) {
        if (row_loader) {
            if (!first_of_tile) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kPB * (kScNB - 2)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kPB * (kScNB - 1)) : "memory");
        } else {
            if (!first_of_tile) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kPA * (kScNA - 2)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kPA * (kScNA - 1)) : "memory");
        }
    };
    int ring_a = 0, ring_b = 0;                                             // stages of the multiplied output tile's K-tile 0 in the two rings
    bool first = true;
    int tile_seq = 0;
#define ORR_STAMP(k) if (FUSED && epi.stamps && tid == 0 && tile_seq < 64) epi.stamps[((int64_t)blockIdx.x * 64 + tile_seq) * 8 + (k)] = __builtin_amdgcn_s_memtime()
    for (int id = valid_from(blockIdx.x); id < total_ids;) {
    ORR_STAMP(0);
    const int next_id = valid_from(id + gridDim.x);
    const bool has_next = next_id < total_ids;
    const int mt = (id >> 3) % n_mtiles, nt = ((id >> 3) / n_mtiles) * 8 + (id & 7);
    const int64_t n0 = row_first + (int64_t)nt * kScBN;
    const int b0 = mt * kScBM;

    typename std::conditional<I8, i32x16v, f32x16>::type acc[LI][2];
#pragma unroll
    for (int i = 0; i < LI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;

    struct Frag { bf16x8 a[LI], b[2]; };
    int c_a = ring_a, c_b = ring_b;                                         // ring stages of the K-tile whose fragments are read next
    auto read_frag = [&](Frag &f, int ks) {
        // fragment addresses: row = tile row + (lane & 31), 16-byte slot = chunk ^ ((row >> 2) & 3) with chunk = 2 ks + (lane >> 5);
        // ks = 1 flips bit 5 of the address.  (Recomputed from the lane id every time, behind an opaque copy: kept in a
        // register across the K loop it is the value the compiler spills, and the reload comes with an s_waitcnt vmcnt(0)
        // that drains the rings.)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int frag = ((ln & 31) * 64 + (((ln >> 5) ^ ((ln >> 2) & 3)) << 4)) ^ (ks * 32);
        const unsigned char *sa = lds_a + c_a * kScImage + wr * 128 * 64 + frag;
        const unsigned char *sb = lds_b + c_b * kScImage + wc * 64 * 64 + frag;
#pragma unroll
        for (int i = 0; i < LI; ++i) f.a[i] = *reinterpret_cast<const bf16x8 *>(sa + i * 2048);
#pragma unroll
        for (int j = 0; j < 2; ++j) f.b[j] = *reinterpret_cast<const bf16x8 *>(sb + j * 2048);
    };

    auto next_stage = [&]() { c_a = c_a + 1 == kScNA ? 0 : c_a + 1; c_b = c_b + 1 == kScNB ? 0 : c_b + 1; };

#define ORR_SB __builtin_amdgcn_sched_barrier(0)
#define ORR_MM(f, i) \
    if constexpr ((i) >= LI) { \
    } else if constexpr (I8) { \
    acc[i][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4v, f.a[i]), __builtin_bit_cast(i32x4v, f.b[0]), acc[i][0], 0, 0, 0); \
    acc[i][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4v, f.a[i]), __builtin_bit_cast(i32x4v, f.b[1]), acc[i][1], 0, 0, 0); \
    } else { \
    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i], f.b[0], acc[i][0], 0, 0, 0); \
    acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i], f.b[1], acc[i][1], 0, 0, 0); } ORR_SB
    // One K-tile: CUR holds the fragments of tile t (read an iteration ago), NXT receives those of tile t+1.
    // At the top a wave's own pieces of tile t+1 have landed (counted wait: behind them its ring's later tiles stay in
    // flight) and its own fragment reads of tile t are complete (lgkmcnt: they were issued half an iteration ago); the
    // barrier then says both for every wave, so tile t+1 is readable and the stages of tile t are free: they take
    // tiles t + kScNA / t + kScNB.  Issue order is pinned, and the first MFMAs go out before the next fragments are
    // requested.  One LDS-DMA piece per pair of MFMAs in the first half of the tile.  (Measured alternatives, 1M x 3072
    // rows x 256 queries, round 1: all pieces right after the barrier, or the SIMD's two waves taking the request half
    // and the multiply half of the period in opposite order: both 5-8 % slower; s_setprio(1) around every MFMA pair:
    // no gain.)
    // (Measured and rejected here, 1M x 3072 rows x 256 queries: the row requesters placing their four requests in the
    // SECOND half of the K-tile so that SIMD partners do not request at the same moment: 0.91 ms against 0.87.)
#define ORR_TILE(CUR0, CUR1, NXT0, NXT1, t) \
    await_own(false); \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
    issue_piece(0); ORR_SB; \
    ORR_MM(CUR0, 0); \
    read_frag(NXT0, 0); ORR_SB; \
    ORR_MM(CUR0, 1); issue_piece(1); ORR_SB; \
    ORR_MM(CUR0, 2); issue_piece(2); ORR_SB; \
    ORR_MM(CUR0, 3); issue_piece(3); advance_stream(); ORR_SB; \
    ORR_MM(CUR1, 0); \
    read_frag(NXT1, 1); next_stage(); ORR_SB; \
    ORR_MM(CUR1, 1); ORR_MM(CUR1, 2); ORR_MM(CUR1, 3)

    // prologue (first output tile of the workgroup only): every stage of this wave's ring requested
    if (first) {
        for (int t = 0; t < ring_n; ++t) {
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) issue_piece(sl);
            advance_stream();
        }
    }
    first = false;
    // tile 0 awaited by its requesters, then visible to everybody
    await_own(true);
    asm volatile("s_barrier" ::: "memory");
    if (active) {
        Frag fa0, fa1, fb0, fb1;
        read_frag(fa0, 0);
        read_frag(fa1, 1);
        next_stage();
        for (int t = 0; t < T; t += 2) {                                    // T is even
            ORR_TILE(fa0, fa1, fb0, fb1, t);
            ORR_TILE(fb0, fb1, fa0, fa1, t + 1);
        }
    } else {
        // waves without queries: the same barriers, their row pieces, nothing else
        for (int t = 0; t < T; ++t) {
            await_own(false);
            asm volatile("s_barrier" ::: "memory");
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) issue_piece(sl);
            advance_stream();
        }
    }
#undef ORR_TILE
#undef ORR_MM
#undef ORR_SB
    if (!has_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the spare pieces of the last tiles
    ORR_STAMP(1);

    if (!FUSED) {
        if (active)
#pragma unroll
        for (int i = 0; i < LI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int64_t col = n0 + wc * 64 + j * 32 + fr;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = b0 + wr * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                    if (row < B && col < n_rows) {
                        if (I8 && (flags & 2)) reinterpret_cast<int *>(S)[(int64_t)row * s_stride + col] = (int)acc[i][j][e];   // raw accumulator (diagnostic)
                        else S[(int64_t)row * s_stride + col] = (float)acc[i][j][e];
                    }
                }
            }
    } else {
        // The epilogue's inputs are laundered once per output tile: otherwise everything it derives from them is
        // hoisted out of the persistent loop and stays live across the K loop, whose registers are all spoken for
        // (the compiler then spilt two values INSIDE it, with an s_waitcnt vmcnt(0) per K-tile to get them back).
        FusedEpilogue ep = epi;
        asm volatile("" : "+s"(ep.rowc), "+s"(ep.qc), "+s"(ep.tau), "+s"(ep.qf), "+s"(ep.count_planes), "+s"(ep.plane_stride),
                          "+s"(ep.i8_rowf), "+s"(ep.i8_qs1), "+s"(ep.cnt), "+s"(ep.buf));
        asm volatile("" : "+s"(ep.kw.bitmaps), "+s"(ep.kw.words_per_term), "+s"(ep.kw.q_term_idx), "+s"(ep.kw.q_term_off));
        uint32_t salt = 0;
        asm volatile("" : "+s"(salt));
        // ONE trip to global memory per tile, requested here and awaited behind the barrier: this thread's 8 bytes of the
        // tile's query constants (256 x float4 = 4 KiB, staged in LDS for every wave) and, for the multiplying waves, the
        // constants and count words of their rows (orr_epilogue.h).  The wait for them is also the wait for the next
        // tile's first K-tiles, which were requested before.
        float2 qf_part;
        {
            const int q = b0 + (tid >> 1);
            qf_part = load_global(reinterpret_cast<const float2 *>(ep.qf), (uint32_t)(2 * (q < B ? q : B - 1) + (tid & 1)));
            if (q >= B) qf_part = (tid & 1) ? make_float2(0.f, 0.f) : make_float2(0.f, __builtin_huge_valf());   // no query: {0, floor +inf, 0, 0}
        }
        EpiTileLoads<LI, 2> pre;
        if (active) epilogue_issue_loads(pre, b0 + wr * 128, n0 + wc * 64, B, n_rows, ep, lane);
        // the epilogue's own 32 KiB: the parking queue and the query constants (the previous tile's epilogue is long over:
        // every wave has passed this tile's barriers since)
        EpiParked *queue = reinterpret_cast<EpiParked *>(lds_epi) + tid;
        float4 *qf_lds = reinterpret_cast<float4 *>(lds_epi + kScEpiQueue * 8 * 512);
        reinterpret_cast<float2 *>(qf_lds)[tid] = qf_part;
        // a bare barrier: __syncthreads() would also wait for the next output tile's first K-tiles
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (active) {
            unsigned long long *st = (epi.stamps && wave == 0 && tile_seq < 64) ? epi.stamps + ((int64_t)blockIdx.x * 64 + tile_seq) * 8 : nullptr;
            // (integer dots go in as they are: |I| <= 3072 * 127^2, the conversion inside costs at most 2^-24)
            fused_epilogue<LI, 2, true, typename std::conditional<I8, i32x16v, f32x16>::type, true, kScEpiQueue>(
                acc, b0 + wr * 128, n0 + wc * 64, B, n_rows, ep, lane, queue, 512, salt, st, &pre, qf_lds + wr * 128);
        }
    }
    ORR_STAMP(2);
    if (FUSED && epi.stamps && tid == 0 && tile_seq < 64) epi.stamps[((int64_t)blockIdx.x * 64 + tile_seq) * 8 + 3] = __builtin_amdgcn_s_memrealtime();
    ++tile_seq;
    ring_a = (ring_a + T) % kScNA;
    ring_b = (ring_b + T) % kScNB;
    id = next_id;
    }
#undef ORR_STAMP
}

// ---------------------------------------------------------------------------
// The same tile with FOUR waves, one per SIMD (screen_tile4_kernel).  LIVE = 8: waves as 2 x 2, each 128 queries x 128
// rows = 4 x 4 accumulator tiles (256 accumulator registers); LIVE <= 4: 1 x 4, each (32 LIVE) x 64.  Per K-tile and
// wave 32 MFMAs (1,024 cycles) against 16 fragment reads and 8 requests, all of it one instruction stream with nobody to
// share the SIMD with: the fragment reads of the NEXT half K-tile and the requests go out between the MFMAs of the
// current one, and the one barrier per K-tile sits between the two halves (by then every wave has read the tile's
// second half, so its stage is free, and the next tile has landed).  Fragments are double-buffered per HALF K-tile
// (64 registers), the rings are 3 query + 6 row stages, the epilogue keeps 16 KiB.
// ---------------------------------------------------------------------------
constexpr int kS4NA = 3, kS4NB = 6;
constexpr int kS4Queue = 6;
constexpr int kS16Queue = 5;                            // the 16 x 16 x 64 form parks five: the sixth plane's first word is its ticket mailbox
constexpr int kS4EpiBytes = kS4Queue * 8 * 256 + 4096;
constexpr int kS4Lds = (kS4NA + kS4NB) * kScImage + kS4EpiBytes;
static_assert(kS4Lds <= 160 * 1024, "the rings and the epilogue's region must fit a CU's LDS");

template <bool FUSED, bool I8, int LIVE = 8, bool NT = false>
__global__ __launch_bounds__(256, 1) void screen_tile4_kernel(const __bf16 *__restrict__ Qh, int32_t B,
                                                              const __bf16 *__restrict__ Eh, int64_t row_first, int64_t n_rows,
                                                              int32_t D, float *__restrict__ S, int64_t s_stride,
                                                              int32_t n_ntiles, int32_t n_mtiles, int32_t flags, FusedEpilogue epi)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char *const lds_a = lds, *const lds_b = lds + kS4NA * kScImage, *const lds_epi = lds + (kS4NA + kS4NB) * kScImage;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NI = LIVE >= 4 ? 4 : LIVE;                               // query tiles of a wave
    constexpr int NJ = LIVE == 8 ? 4 : 2;                                  // row tiles of a wave
    const int wr = LIVE == 8 ? wave >> 1 : 0, wc = LIVE == 8 ? wave & 1 : wave;
    const bool row_loader = wave >= 2;                                      // waves 2, 3 request rows, waves 0, 1 queries
    constexpr int kPA = LIVE == 8 ? 8 : LIVE == 4 ? 4 : 2;                  // query pieces per K-tile and requesting wave
    constexpr int kPB = 8;                                                  // row pieces per K-tile and requesting wave
    const int total_ids = ((n_ntiles + 7) / 8) * 8 * n_mtiles;              // (walk of the output tiles: as in screen_bf16_kernel)
    auto valid_from = [&](int id) {
        while (id < total_ids && ((id >> 3) / n_mtiles) * 8 + (id & 7) >= n_ntiles) id += gridDim.x;
        return id;
    };
    const int T = I8 ? D / 64 : D / kScBK;
    const int fr = lane & 31, fh = lane >> 5;
    const uint32_t lane_off = (uint32_t)(lane * 16);
    const uint32_t part_off = row_loader ? (uint32_t)((wave & 1) * kPB * 1024) : (uint32_t)(wave * kPA * 1024);
    auto stream_src = [&](int sid) -> const unsigned char * {
        const int smt = (sid >> 3) % n_mtiles, snt = ((sid >> 3) / n_mtiles) * 8 + (sid & 7);
        return row_loader ? reinterpret_cast<const unsigned char *>(Eh) + ((int64_t)(row_first / kScBN + snt) * T) * kScImage + part_off
                          : reinterpret_cast<const unsigned char *>(Qh) + ((int64_t)smt * T) * kScImage + part_off;
    };
    int s_id = valid_from(blockIdx.x), s_k = 0, s_stage = 0;
    const unsigned char *s_src = s_id < total_ids ? stream_src(s_id) : nullptr;
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(row_loader ? lds_b : lds_a) + part_off;
    const int ring_n = row_loader ? kS4NB : kS4NA;
    // NT: the rows are streamed once (the whole batch fits one query tile): requested non-temporal, so they do not push the
    // query images (re-read by every workgroup) out of L2.  (A template parameter, and the piece index a constant at every
    // call, and the K loop below exists once per role: a request is M0, a wait state and the load, no branch.)
    auto issue_piece = [&](auto role, int slot) __attribute__((always_inline)) {
        constexpr bool ROWS = decltype(role)::value;
        if constexpr (!ROWS && kPA < kPB) { if (slot >= kPA) return; }
        // (the immediate offset, applied to both addresses, has 12 bits: pieces 4..7 go through bases 4 KiB further on)
        const uint32_t m0v = ring_lds + (uint32_t)s_stage * kScImage + (uint32_t)(slot >> 2) * 4096u;
        const uint64_t src = (uint64_t)(uintptr_t)s_src + (uint64_t)(slot >> 2) * 4096u;
#define ORR_GLDS(OFF, POLICY) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:" #OFF POLICY \
                                           :: "v"(lane_off), "s"(src), "s"(m0v) : "memory")
        if constexpr (NT && ROWS) {
            switch (slot & 3) { case 0: ORR_GLDS(0, " nt"); break; case 1: ORR_GLDS(1024, " nt"); break; case 2: ORR_GLDS(2048, " nt"); break; default: ORR_GLDS(3072, " nt"); }
        } else {
            switch (slot & 3) { case 0: ORR_GLDS(0, ""); break; case 1: ORR_GLDS(1024, ""); break; case 2: ORR_GLDS(2048, ""); break; default: ORR_GLDS(3072, ""); }
        }
#undef ORR_GLDS
    };
    auto advance_stream = [&]() {
        s_stage = s_stage + 1 == ring_n ? 0 : s_stage + 1;
        if (s_k + 1 < T) { ++s_k; s_src += kScImage; return; }
        const int nid = valid_from(s_id + gridDim.x);
        if (nid < total_ids) { s_id = nid; s_k = 0; s_src = stream_src(nid); }
    };
    int ring_a = 0, ring_b = 0;
    bool first = true;
    int tile_seq = 0;
#define ORR_STAMP(k) if (FUSED && epi.stamps && tid == 0 && tile_seq < 64) epi.stamps[((int64_t)blockIdx.x * 64 + tile_seq) * 8 + (k)] = __builtin_amdgcn_s_memtime()
    for (int id = valid_from(blockIdx.x); id < total_ids;) {
    ORR_STAMP(0);
    const int next_id = valid_from(id + gridDim.x);
    const bool has_next = next_id < total_ids;
    const int mt = (id >> 3) % n_mtiles, nt = ((id >> 3) / n_mtiles) * 8 + (id & 7);
    const int64_t n0 = row_first + (int64_t)nt * kScBN;
    const int b0 = mt * kScBM;

    typename std::conditional<I8, i32x16v, f32x16>::type acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;

    struct Frag { bf16x8 a[NI], b[NJ]; };
    int c_a = ring_a, c_b = ring_b;                                         // ring stages of the K-tile whose fragments are read next
    // fragment addresses: row = tile row + (lane & 31), 16-byte slot = chunk ^ ((row >> 2) & 3) with chunk = 2 ks + (lane >> 5);
    // ks = 1 flips bit 5 of the address (one wave per SIMD has the registers to keep both)
    const int frag0 = (lane & 31) * 64 + (((lane >> 5) ^ ((lane >> 2) & 3)) << 4), frag1 = frag0 ^ 32;
    auto a_at = [&](int ks) { return lds_a + c_a * kScImage + wr * 128 * 64 + (ks ? frag1 : frag0); };
    auto b_at = [&](int ks) { return lds_b + c_b * kScImage + wc * (NJ * 32) * 64 + (ks ? frag1 : frag0); };
    auto next_stage = [&]() { c_a = c_a + 1 == kS4NA ? 0 : c_a + 1; c_b = c_b + 1 == kS4NB ? 0 : c_b + 1; };

    // One instruction stream per SIMD: a 32 x 32 x 32 MFMA holds the matrix pipe for 32 cycles, and whatever the wave issues
    // behind it in that time is free.  So the stream is 32 slots per K-tile, each one MFMA followed by AT MOST one fragment
    // read and one request -- nothing is issued in clumps (with the reads and requests grouped after every four MFMAs the
    // K-tile took 1,940 cycles for 1,024 of MFMA).
#define ORR_SB __builtin_amdgcn_sched_barrier(0)
#define ORR_MM(f, i, j) \
    if constexpr ((i) < NI && (j) < NJ) { \
        if constexpr (I8) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4v, f.a[i]), __builtin_bit_cast(i32x4v, f.b[j]), acc[i][j], 0, 0, 0); \
        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i], f.b[j], acc[i][j], 0, 0, 0); } ORR_SB
#define ORR_RA(f, p, i) if constexpr ((i) < NI) f.a[i] = *reinterpret_cast<const bf16x8 *>((p) + (i) * 2048); ORR_SB
#define ORR_RB(f, p, j) if constexpr ((j) < NJ) f.b[j] = *reinterpret_cast<const bf16x8 *>((p) + (j) * 2048); ORR_SB
    // The epilogue's one trip to global memory (this thread's query constants for the LDS copy, its rows' constants and
    // count words, orr_epilogue.h) is requested from INSIDE the K loop, behind the barrier of the last K-tile but one:
    // a wave's loads return in order, so behind the requests of the next output tile's first K-tiles they would arrive
    // when that whole ring has (10,000 cycles per tile by the stamps).  Hence also: the last kS4NA / kS4NB iterations
    // request nothing (their counted waits shrink with what is left in flight), and the ring is refilled for the next
    // output tile only when the epilogue's inputs are in registers -- it fills while the epilogue computes.
    FusedEpilogue ep = epi;
    EpiTileLoads<NI, NJ> pre;
    float4 qf_mine = make_float4(0.f, 0.f, 0.f, 0.f);
    auto epilogue_requests = [&]() __attribute__((always_inline)) {
        if constexpr (FUSED) {
            // laundered once per output tile: otherwise everything derived from these is hoisted out of the persistent loop
            asm volatile("" : "+s"(ep.rowc), "+s"(ep.qc), "+s"(ep.tau), "+s"(ep.qf), "+s"(ep.count_planes), "+s"(ep.plane_stride),
                              "+s"(ep.i8_rowf), "+s"(ep.i8_qs1), "+s"(ep.cnt), "+s"(ep.buf));
            asm volatile("" : "+s"(ep.kw.bitmaps), "+s"(ep.kw.words_per_term), "+s"(ep.kw.q_term_idx), "+s"(ep.kw.q_term_off));
            const int q = b0 + tid;
            qf_mine = load_global(ep.qf, (uint32_t)(q < B ? q : B - 1));
            epilogue_issue_loads(pre, b0 + wr * 128, n0 + wc * (NJ * 32), B, n_rows, ep, lane);
        }
    };
    auto refill = [&](auto role) __attribute__((always_inline)) {          // every stage of this wave's ring requested
        for (int t = 0; t < ring_n; ++t) {
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) issue_piece(role, sl);
            advance_stream();
        }
    };
    // (one copy of the loop per requesting role: the counted waits and the request policy are then immediates)
    auto k_loop = [&](auto role) __attribute__((always_inline)) {
    constexpr bool ROWS = decltype(role)::value;
    constexpr int RING = ROWS ? kS4NB : kS4NA, KP = ROWS ? kPB : kPA;
    if (first) refill(role);
    first = false;
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(KP * (RING - 1)) : "memory");     // tile 0 awaited by its requesters, then visible to everybody
    Frag f0, f1;
    {
        const unsigned char *pa = a_at(0), *pb = b_at(0);
        ORR_RA(f0, pa, 0); ORR_RA(f0, pa, 1); ORR_RA(f0, pa, 2); ORR_RA(f0, pa, 3);
        ORR_RB(f0, pb, 0); ORR_RB(f0, pb, 1); ORR_RB(f0, pb, 2); ORR_RB(f0, pb, 3);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // one K-tile.  REQ: its stages are re-requested (K-tile t + RING of this output tile); WAITN: what may stay in flight
    // when this wave's pieces of K-tile t + 1 have landed; HOOK: the epilogue's requests go out behind the barrier
    auto tile_iter = [&](auto req_c, auto waitn_c, auto hook_c) __attribute__((always_inline)) {
        constexpr bool REQ = decltype(req_c)::value, HOOK = decltype(hook_c)::value;
        constexpr int WAITN = decltype(waitn_c)::value;
        // first half: tile t, first half of its k from f0; the reads of the second half go out behind the first MFMAs
        const unsigned char *pa = a_at(1), *pb = b_at(1);
        ORR_SB;
        ORR_MM(f0, 0, 0); ORR_RA(f1, pa, 0);
        ORR_MM(f0, 0, 1); ORR_RA(f1, pa, 1);
        ORR_MM(f0, 0, 2); ORR_RA(f1, pa, 2);
        ORR_MM(f0, 0, 3); ORR_RA(f1, pa, 3);
        ORR_MM(f0, 1, 0); ORR_RB(f1, pb, 0);
        ORR_MM(f0, 1, 1); ORR_RB(f1, pb, 1);
        ORR_MM(f0, 1, 2); ORR_RB(f1, pb, 2);
        ORR_MM(f0, 1, 3); ORR_RB(f1, pb, 3);
        ORR_MM(f0, 2, 0); ORR_MM(f0, 2, 1); ORR_MM(f0, 2, 2); ORR_MM(f0, 2, 3);
        ORR_MM(f0, 3, 0); ORR_MM(f0, 3, 1); ORR_MM(f0, 3, 2); ORR_MM(f0, 3, 3);
        // every wave has read tile t completely and this wave's pieces of tile t + 1 have landed: after the barrier tile
        // t + 1 is readable and tile t's stages are free
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" :: "n"(WAITN) : "memory");
        ORR_SB;
        next_stage();
        pa = a_at(0); pb = b_at(0);
        if constexpr (HOOK) epilogue_requests();
        ORR_SB;
#define ORR_P(k) if constexpr (REQ) { issue_piece(role, k); } ORR_SB
        ORR_MM(f1, 0, 0); ORR_RA(f0, pa, 0); ORR_P(0);
        ORR_MM(f1, 0, 1); ORR_RA(f0, pa, 1);
        ORR_MM(f1, 0, 2); ORR_RA(f0, pa, 2); ORR_P(1);
        ORR_MM(f1, 0, 3); ORR_RA(f0, pa, 3);
        ORR_MM(f1, 1, 0); ORR_RB(f0, pb, 0); ORR_P(2);
        ORR_MM(f1, 1, 1); ORR_RB(f0, pb, 1);
        ORR_MM(f1, 1, 2); ORR_RB(f0, pb, 2); ORR_P(3);
        ORR_MM(f1, 1, 3); ORR_RB(f0, pb, 3);
        ORR_MM(f1, 2, 0); ORR_P(4);
        ORR_MM(f1, 2, 1);
        ORR_MM(f1, 2, 2); ORR_P(5);
        ORR_MM(f1, 2, 3);
        ORR_MM(f1, 3, 0); ORR_P(6);
        ORR_MM(f1, 3, 1);
        ORR_MM(f1, 3, 2); ORR_P(7);
        ORR_MM(f1, 3, 3);
#undef ORR_P
        if constexpr (REQ) advance_stream();
        ORR_SB;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    for (int t = 0; t < T - RING; ++t)
        tile_iter(std::true_type{}, std::integral_constant<int, KP * (RING - 2)>{}, std::false_type{});
    ORR_STAMP(7);
    // the last RING K-tiles: r-th of them leaves RING - 2 - r tiles in flight (the last one waits for nothing: 63)
    auto tail = [&](auto self, auto r_c) __attribute__((always_inline)) {
        constexpr int r = decltype(r_c)::value;
        if constexpr (r < RING) {
            tile_iter(std::false_type{}, std::integral_constant<int, (r <= RING - 2 ? KP * (RING - 2 - r) : 63)>{},
                      std::integral_constant<bool, r == RING - 2>{});
            self(self, std::integral_constant<int, r + 1>{});
        }
    };
    tail(tail, std::integral_constant<int, 0>{});
    };
    if (row_loader) k_loop(std::true_type{}); else k_loop(std::false_type{});
    ORR_STAMP(1);

    if (!FUSED) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int64_t col = n0 + wc * (NJ * 32) + j * 32 + fr;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = b0 + wr * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                    if (row < B && col < n_rows) {
                        if (I8 && (flags & 2)) reinterpret_cast<int *>(S)[(int64_t)row * s_stride + col] = (int)acc[i][j][e];   // raw accumulator (diagnostic)
                        else S[(int64_t)row * s_stride + col] = (float)acc[i][j][e];
                    }
                }
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // (the stores: the counted waits of the next tile's K loop count requests only)
        if (has_next) { if (row_loader) refill(std::true_type{}); else refill(std::false_type{}); }
    } else {
        uint32_t salt = 0;
        asm volatile("" : "+s"(salt));
        if (b0 + tid >= B) qf_mine = make_float4(0.f, __builtin_huge_valf(), 0.f, 0.f);    // no query: floor +inf
        EpiParked *queue = reinterpret_cast<EpiParked *>(lds_epi) + tid;
        float4 *qf_lds = reinterpret_cast<float4 *>(lds_epi + kS4Queue * 8 * 256);
        qf_lds[tid] = qf_mine;
        // every loaded register is touched here: the compiler's wait for the loads lands HERE, in front of the refill (further
        // down it would wait for the refill's requests as well, which are younger)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            asm volatile("" : "+v"(pre.rc[j].x), "+v"(pre.rc[j].y));
            asm volatile("" : "+v"(pre.rf[j].x), "+v"(pre.rf[j].y), "+v"(pre.rf[j].z), "+v"(pre.rf[j].w));
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int pl = 0; pl < kCountPlanes; ++pl) asm volatile("" : "+v"(pre.w[i][j][pl]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // (bare: __syncthreads() would also wait for outstanding requests)
        unsigned long long *st = (epi.stamps && wave == 0 && tile_seq < 64) ? epi.stamps + ((int64_t)blockIdx.x * 64 + tile_seq) * 8 : nullptr;
        // the ring refill for the next output tile rides on the epilogue: a quarter of this wave's requests in front of each
        // block of 32 queries (NI blocks; the last one takes what is left)
        auto refill_part = [&](int blk) __attribute__((always_inline)) {
            if (!has_next) return;
            auto part = [&](auto role) __attribute__((always_inline)) {
                constexpr int RING = decltype(role)::value ? kS4NB : kS4NA;
                const int t0 = blk * RING / NI, t1 = blk + 1 == NI ? RING : (blk + 1) * RING / NI;
                for (int t = t0; t < t1; ++t) {
#pragma unroll
                    for (int sl = 0; sl < 8; ++sl) issue_piece(role, sl);
                    advance_stream();
                }
            };
            if (row_loader) part(std::true_type{}); else part(std::false_type{});
        };
        fused_epilogue<NI, NJ, true, typename std::conditional<I8, i32x16v, f32x16>::type, true, kS4Queue, LIVE == 8>(
            acc, b0 + wr * 128, n0 + wc * (NJ * 32), B, n_rows, ep, lane, queue, 256, salt, st, &pre, qf_lds + wr * 128, refill_part);
    }
#undef ORR_RB
#undef ORR_RA
#undef ORR_MM
#undef ORR_SB
    ORR_STAMP(2);
    if (FUSED && epi.stamps && tid == 0 && tile_seq < 64) epi.stamps[((int64_t)blockIdx.x * 64 + tile_seq) * 8 + 3] = __builtin_amdgcn_s_memrealtime();
    ++tile_seq;
    ring_a = (ring_a + T) % kS4NA;
    ring_b = (ring_b + T) % kS4NB;
    id = next_id;
    }
#undef ORR_STAMP
}

// ---------------------------------------------------------------------------
// The four-wave tile on v_mfma_i32_16x16x64_i8 (screen_tile16_kernel; 129+ queries on the int8 shadow).  Same rings,
// requests, barrier protocol, tail, refill and epilogue inputs as screen_tile4_kernel; the 128 x 128 wave tile is 8 x 8
// accumulator tiles of four registers.  Why: tools/mfma_rate.hip -- with every matrix core busy the chip holds 1.88 GHz on
// this shape against 1.60 GHz on 32 x 32 x 32 at the same operations per cycle (profiles/r02_mfma_rate_microbench.txt).
// The stored swizzle was made for 32-row fragments; a fragment here takes its 16 rows in the order tile16_row, which makes
// the 16-row x 4-chunk reads conflict-free on it.  The epilogue for this accumulator layout: fused_epilogue16 (orr_epilogue.h).
// ---------------------------------------------------------------------------
// DOTS (diagnostic, orr_index_screen_i8_dots): no scoring epilogue -- the raw int32 accumulators go to S[query][row].
// BITS2: the count words hold two bits per (query,row) (FusedEpilogue::count_bits == 2).
template <bool NT, bool DOTS = false, bool BITS2 = false>
__global__ __launch_bounds__(256, 1) void screen_tile16_kernel(const __bf16 *__restrict__ Qh, int32_t B,
                                                              const __bf16 *__restrict__ Eh, int64_t row_first, int64_t n_rows,
                                                              int32_t D, float *__restrict__ S, int64_t s_stride,
                                                              int32_t n_ntiles, int32_t n_mtiles, int32_t flags, FusedEpilogue epi)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char *const lds_a = lds, *const lds_b = lds + kS4NA * kScImage, *const lds_epi = lds + (kS4NA + kS4NB) * kScImage;
    constexpr bool FUSED = true, I8 = true;
    constexpr int LIVE = 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = LIVE == 8 ? wave >> 1 : 0, wc = LIVE == 8 ? wave & 1 : wave;
    const bool row_loader = wave >= 2;                                      // waves 2, 3 request rows, waves 0, 1 queries
    constexpr int kPA = LIVE == 8 ? 8 : LIVE == 4 ? 4 : 2;                  // query pieces per K-tile and requesting wave
    constexpr int kPB = 8;                                                  // row pieces per K-tile and requesting wave
    const int total_ids = ((n_ntiles + 7) / 8) * 8 * n_mtiles;              // (walk of the output tiles: as in screen_bf16_kernel)
    auto valid_from = [&](int id) {
        while (id < total_ids && ((id >> 3) / n_mtiles) * 8 + (id & 7) >= n_ntiles) id += gridDim.x;
        return id;
    };
    const int T = I8 ? D / 64 : D / kScBK;
    const uint32_t lane_off = (uint32_t)(lane * 16);
    const uint32_t part_off = row_loader ? (uint32_t)((wave & 1) * kPB * 1024) : (uint32_t)(wave * kPA * 1024);
    auto stream_src = [&](int sid) -> const unsigned char * {
        const int smt = (sid >> 3) % n_mtiles, snt = ((sid >> 3) / n_mtiles) * 8 + (sid & 7);
        return row_loader ? reinterpret_cast<const unsigned char *>(Eh) + ((int64_t)(row_first / kScBN + snt) * T) * kScImage + part_off
                          : reinterpret_cast<const unsigned char *>(Qh) + ((int64_t)smt * T) * kScImage + part_off;
    };
    int s_id = valid_from(blockIdx.x), s_k = 0, s_stage = 0;
    const unsigned char *s_src = s_id < total_ids ? stream_src(s_id) : nullptr;
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(row_loader ? lds_b : lds_a) + part_off;
    const int ring_n = row_loader ? kS4NB : kS4NA;
    // NT: the rows are streamed once (the whole batch fits one query tile): requested non-temporal, so they do not push the
    // query images (re-read by every workgroup) out of L2.  (A template parameter, and the piece index a constant at every
    // call, and the K loop below exists once per role: a request is M0, a wait state and the load, no branch.)
    auto issue_piece = [&](auto role, int slot) __attribute__((always_inline)) {
        constexpr bool ROWS = decltype(role)::value;
        if constexpr (!ROWS && kPA < kPB) { if (slot >= kPA) return; }
        // (the immediate offset, applied to both addresses, has 12 bits: pieces 4..7 go through bases 4 KiB further on)
        const uint32_t m0v = ring_lds + (uint32_t)s_stage * kScImage + (uint32_t)(slot >> 2) * 4096u;
        const uint64_t src = (uint64_t)(uintptr_t)s_src + (uint64_t)(slot >> 2) * 4096u;
#define ORR_GLDS(OFF, POLICY) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:" #OFF POLICY \
                                           :: "v"(lane_off), "s"(src), "s"(m0v) : "memory")
        if constexpr (NT && ROWS) {
            switch (slot & 3) { case 0: ORR_GLDS(0, " nt"); break; case 1: ORR_GLDS(1024, " nt"); break; case 2: ORR_GLDS(2048, " nt"); break; default: ORR_GLDS(3072, " nt"); }
        } else {
            switch (slot & 3) { case 0: ORR_GLDS(0, ""); break; case 1: ORR_GLDS(1024, ""); break; case 2: ORR_GLDS(2048, ""); break; default: ORR_GLDS(3072, ""); }
        }
#undef ORR_GLDS
    };
    auto advance_stream = [&]() {
        s_stage = s_stage + 1 == ring_n ? 0 : s_stage + 1;
        if (s_k + 1 < T) { ++s_k; s_src += kScImage; return; }
        const int nid = valid_from(s_id + gridDim.x);
        if (nid < total_ids) { s_id = nid; s_k = 0; s_src = stream_src(nid); }
    };
    int ring_a = 0, ring_b = 0;
    bool first = true;
    int tile_seq = 0;
    // ---- TICKETS (epi.tickets != nullptr): a workgroup's first two output tiles are assigned statically (ids blockIdx.x and
    // blockIdx.x + gridDim.x), every later one is drawn from a counter when the workgroup gets there -- workgroups that run
    // ahead (a CU whose tiles had nothing to park, an XCD with the shorter way to the rows) take more tiles instead of idling
    // at the end of the launch (static assignment: workgroups ended 5-12 % apart by the stamps).  One counter for the launch
    // when the batch is one query tile (every row tile is read once, by whoever); one per XCD otherwise (the query tiles of a
    // row tile stay on the XCD whose L2 holds the rows).  The valid ids are a dense prefix in both forms -- id < n_ntiles, or
    // slot id >> 3 < the XCD's slot count -- so "none left" is final.  The ticket for the tile AFTER NEXT is drawn where the
    // epilogue's loads go out (in the tail of the K loop, where this wave has nothing in flight: its value comes back with
    // them) and handed to the other waves through LDS behind the barrier in front of the epilogue: the request stream needs
    // the next tile's id long before the multiplication gets there.
    const bool dyn = epi.tickets != nullptr;
    const int xcd = blockIdx.x & 7;
    const int lim_slots = n_mtiles * ((n_ntiles - xcd + 7) >> 3);
    auto dyn_checked = [&](int id) { return (n_mtiles == 1 ? id < n_ntiles : (id >> 3) < lim_slots) ? id : total_ids; };
    auto ticket_id = [&](uint32_t t) {
        return n_mtiles == 1 ? (int)t + 2 * (int)gridDim.x : ((((int)t + 2 * (int)(gridDim.x >> 3)) << 3) | xcd);
    };
    volatile int *const mailbox = reinterpret_cast<volatile int *>(lds_epi + kS16Queue * 8 * 256);     // (the LDS of the parking queue's sixth plane)
    int id_after = dyn ? dyn_checked((int)blockIdx.x + (int)gridDim.x) : 0;    // ticket mode: the id after the current one
    if (dyn) {                                                                  // (the request stream of the first tile: as assigned here)
        s_id = dyn_checked(blockIdx.x);
        s_src = s_id < total_ids ? stream_src(s_id) : nullptr;
    }
    uint32_t ticket = 0u;
#define ORR_STAMP(k) if (FUSED && epi.stamps && tid == 0 && tile_seq < 64) epi.stamps[((int64_t)blockIdx.x * 64 + tile_seq) * 8 + (k)] = __builtin_amdgcn_s_memtime()
    for (int id = dyn ? dyn_checked(blockIdx.x) : valid_from(blockIdx.x); id < total_ids;) {
    ORR_STAMP(0);
    // (laundered per output tile: what hangs on the lane number -- fragment offsets, the epilogue's row and query numbers --
    // is made again for every tile instead of living, spilled, across the epilogue)
    int lane_t = lane;
    asm volatile("" : "+v"(lane_t));
    const int next_id = dyn ? id_after : valid_from(id + gridDim.x);
    const bool has_next = next_id < total_ids;
    const int mt = (id >> 3) % n_mtiles, nt = ((id >> 3) / n_mtiles) * 8 + (id & 7);
    const int64_t n0 = row_first + (int64_t)nt * kScBN;
    const int b0 = mt * kScBM;

    // the accumulators: a[0:255], tile (query tile i, row tile j) = a[4 (8 i + j) .. + 3] -- no C++ object, see the K loop
    // (never zeroed: the first K-tile's MFMAs multiply onto the constant 0)
    int acc_token = 0;
    int c_a = ring_a, c_b = ring_b;                                         // ring stages of the K-tile whose fragments are read next
    // 16 x 16 x 64 fragments: lane = (row l & 15, 16-byte chunk l >> 4 of the row's 64 bytes = the whole K-tile); the stored
    // slot of chunk c in row r is c ^ ((r >> 2) & 3), and (r >> 2) & 3 = (l >> 2) & 3 for every tile of 16 rows
    // -- read for the fragment rows in the order tile16_row (orr_epilogue.h), which is what makes the reads conflict-free
    const int frow = tile16_row(lane_t & 15);
    const int fo = frow * 64 + (((lane_t >> 4) ^ ((frow >> 2) & 3)) << 4);
    auto a_at = [&]() { return lds_a + c_a * kScImage + wr * 128 * 64 + fo; };
    auto b_at = [&]() { return lds_b + c_b * kScImage + wc * 128 * 64 + fo; };
    // 64 MFMAs per K-tile (16.3 cycles each), 16 fragment reads, 8 requests.  First half: query tiles 0..3 against all
    // eight row tiles (row tile varies fastest), the fragments of query tiles 4..7 arriving meanwhile; the barrier; second
    // half column by column (row tile j against query tiles 4..7), so that row fragment j is free after its column and is
    // re-read for the NEXT K-tile two columns later -- one register set for the row fragments -- and the next K-tile's query
    // fragments 0..3 take the registers the first half no longer needs.
    // The whole K-tile is ONE piece of assembler text (orr_screen_tile16_asm.inc, generated by tools/gen_tile16_asm.py) on
    // fixed registers -- accumulators a[0:255], fragments v[192:255] (pinned operands of the statement).  With the compiler's
    // MFMA builtin and all 256 accumulation registers taken, most MFMAs got a copy of their tile in front (100 v_accvgpr_mov
    // and 61 s_nop per K-tile); with one asm statement per MFMA the allocator moved tiles between code regions right behind
    // MFMAs whose latency it cannot see (wrong results) or kept copies in vector registers (1,062 spills).  Hazards the text
    // has to respect itself: a fragment register is overwritten at the earliest 4 MFMAs after its last use (plus the LDS
    // latency), every fragment read is waited for (counted lgkmcnt) before the half that uses it, and the accumulators are
    // not read before the s_nops behind the K loop.  The K-tile's bookkeeping (ring stages, fragment addresses, request
    // offsets) is part of the text too, in the shadow of the MFMAs: the loop around it is a counter and a branch.
#define ORR_SB __builtin_amdgcn_sched_barrier(0)
#define ORR_RD(dst, p, i) dst = *reinterpret_cast<const i32x4v *>((p) + (i) * 1024); ORR_SB
    // The epilogue's one trip to global memory (this thread's query constants for the LDS copy, its rows' constants and
    // count words, orr_epilogue.h) is requested from INSIDE the K loop, behind the barrier of the last K-tile but one:
    // a wave's loads return in order, so behind the requests of the next output tile's first K-tiles they would arrive
    // when that whole ring has (10,000 cycles per tile by the stamps).  Hence also: the last kS4NA / kS4NB iterations
    // request nothing (their counted waits shrink with what is left in flight), and the ring is refilled for the next
    // output tile only when the epilogue's inputs are in registers -- it fills while the epilogue computes.
    FusedEpilogue ep = epi;
    EpiTileLoads16 pre;
    float4 qf_mine = make_float4(0.f, 0.f, 0.f, 0.f);
    auto epilogue_requests = [&]() __attribute__((always_inline)) {
        if (dyn && tid == 0) ticket = atomicAdd(&epi.tickets[n_mtiles == 1 ? 0 : xcd], 1u);      // (the oldest of what goes out here)
        if constexpr (FUSED && !DOTS) {
            // laundered once per output tile: otherwise everything derived from these is hoisted out of the persistent loop
            asm volatile("" : "+s"(ep.rowc), "+s"(ep.qc), "+s"(ep.tau), "+s"(ep.qf), "+s"(ep.count_planes), "+s"(ep.plane_stride),
                              "+s"(ep.i8_rowf), "+s"(ep.i8_qs1), "+s"(ep.cnt), "+s"(ep.buf));
            asm volatile("" : "+s"(ep.kw.bitmaps), "+s"(ep.kw.words_per_term), "+s"(ep.kw.q_term_idx), "+s"(ep.kw.q_term_off));
            const int q = b0 + tid;
            qf_mine = load_global(ep.qf16, (uint32_t)(q < B ? q : B - 1));      // (the NaN-safe constants with the batch's QW in .w)
            epilogue_issue_loads16<BITS2 ? 2 : 4>(pre, b0 + wr * 128, n0 + wc * 128, B, n_rows, ep, lane_t);
        }
    };
    auto refill = [&](auto role) __attribute__((always_inline)) {          // every stage of this wave's ring requested
        for (int t = 0; t < ring_n; ++t) {
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) issue_piece(role, sl);
            advance_stream();
        }
    };
    // (one copy of the loop per requesting role: the counted waits and the request policy are then immediates)
    auto k_loop = [&](auto role) __attribute__((always_inline)) {
    constexpr bool ROWS = decltype(role)::value;
    constexpr int RING = ROWS ? kS4NB : kS4NA, KP = ROWS ? kPB : kPA;
    if (first) refill(role);
    first = false;
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(KP * (RING - 1)) : "memory");     // tile 0 awaited by its requesters, then visible to everybody
    i32x4v fa0, fa1, fa2, fa3, fa4, fa5, fa6, fa7, fb0, fb1, fb2, fb3, fb4, fb5, fb6, fb7;     // query fragments 0..7, row fragments 0..7
    {
        const unsigned char *pa = a_at(), *pb = b_at();
        ORR_RD(fa0, pa, 0); ORR_RD(fa1, pa, 1); ORR_RD(fa2, pa, 2); ORR_RD(fa3, pa, 3);
        ORR_RD(fb0, pb, 0); ORR_RD(fb1, pb, 1); ORR_RD(fb2, pb, 2); ORR_RD(fb3, pb, 3);
        ORR_RD(fb4, pb, 4); ORR_RD(fb5, pb, 5); ORR_RD(fb6, pb, 6); ORR_RD(fb7, pb, 7);
        fa4 = fa5 = fa6 = fa7 = fa0;                                           // (defined; read in the first K-tile's first half)
    }
    // The compiler's own wait for these reads has to land HERE: it does not see the waits inside the assembler text, and with
    // the reads still pending in its books at the loop's entry it puts an s_waitcnt lgkmcnt(0) in front of every K-tile.
    asm volatile("" : "+v"(fa0), "+v"(fa1), "+v"(fa2), "+v"(fa3), "+v"(fb0), "+v"(fb1), "+v"(fb2), "+v"(fb3),
                      "+v"(fb4), "+v"(fb5), "+v"(fb6), "+v"(fb7));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // The state the K-tiles' text keeps for itself (tools/gen_tile16_asm.py: ring stages, LDS addresses, request offsets, in
    // vector registers -- as compiler-made code between the texts, twenty dependent scalar instructions, it cost 230 idle
    // cycles per K-tile):
    const uint32_t pab = (uint32_t)(uintptr_t)lds_a + (uint32_t)(wr * 128 * 64 + fo), pbb = (uint32_t)(uintptr_t)lds_b + (uint32_t)(wc * 128 * 64 + fo);
    uint32_t pa = pab + (uint32_t)c_a * kScImage, pan = pab + (uint32_t)(c_a + 1 == kS4NA ? 0 : c_a + 1) * kScImage;
    uint32_t pbn = pbb + (uint32_t)(c_b + 1 == kS4NB ? 0 : c_b + 1) * kScImage, pat = 0, pbt = 0;
    const uint32_t pae = pab + (uint32_t)kS4NA * kScImage, pbe = pbb + (uint32_t)kS4NB * kScImage;
    const uint32_t m0e = ring_lds + (uint32_t)RING * kScImage;
    uint32_t m0s = ring_lds, m0v = ring_lds + (uint32_t)(s_stage == 0 ? RING - 1 : s_stage - 1) * kScImage;    // (advanced in front of the requests)
    asm volatile("" : "+v"(m0s), "+v"(m0v));                                                           // (vector registers, the same in every lane)
    uint32_t vo = lane_off - (uint32_t)kScImage, vo2 = vo + 4096u;                                    // (likewise)
    const uint64_t src = (uint64_t)(uintptr_t)s_src;
    const int n_main = T - RING;
    // one K-tile.  FIRST: it multiplies onto 0 (no zeroing of the accumulators); REQ: its stages are re-requested (K-tile
    // t + RING of this output tile); WAITN: what may stay in flight when this wave's pieces of K-tile t + 1 have landed;
    // HOOK: the epilogue's requests go out in front of it
    auto tile_iter = [&](auto first_c, auto req_c, auto waitn_c, auto hook_c) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value, REQ = decltype(req_c)::value, HOOK = decltype(hook_c)::value;
        constexpr int WAITN = decltype(waitn_c)::value;
        uint32_t m0_scratch;
#define ORR_T16_OPERANDS \
            : "+{v[192:195]}"(fa0), "+{v[196:199]}"(fa1), "+{v[200:203]}"(fa2), "+{v[204:207]}"(fa3), \
              "+{v[208:211]}"(fa4), "+{v[212:215]}"(fa5), "+{v[216:219]}"(fa6), "+{v[220:223]}"(fa7), \
              "+{v[224:227]}"(fb0), "+{v[228:231]}"(fb1), "+{v[232:235]}"(fb2), "+{v[236:239]}"(fb3), \
              "+{v[240:243]}"(fb4), "+{v[244:247]}"(fb5), "+{v[248:251]}"(fb6), "+{v[252:255]}"(fb7), \
              [pa] "+v"(pa), [pan] "+v"(pan), [pbn] "+v"(pbn), [pat] "+v"(pat), [pbt] "+v"(pbt), [vo] "+v"(vo), [vo2] "+v"(vo2), [m0v] "+v"(m0v), [st] "=&s"(m0_scratch) \
            : [pab] "v"(pab), [pbb] "v"(pbb), [pae] "v"(pae), [pbe] "v"(pbe), [m0s] "v"(m0s), [src] "s"(src), [m0e] "s"(m0e), [wn] "n"(WAITN) \
            : ORR_T16_ACC_CLOBBERS, "vcc", "scc", "memory"
        if constexpr (HOOK) {
            // the epilogue's loads go out BEHIND this K-tile's counted wait (a vmcnt(0) here: it would wait for them too, 2,000
            // cycles with the matrix core idle), i.e. between the two halves of its text
            static_assert(!REQ && !FIRST, "the hook rides on a K-tile of the tail");
            asm volatile(ORR_T16_KTILE_NOREQ_H1 ORR_T16_OPERANDS);
            epilogue_requests();
            asm volatile(ORR_T16_KTILE_NOREQ_H2 ORR_T16_OPERANDS);
        } else if constexpr (FIRST) {
            if constexpr (!REQ) asm volatile(ORR_T16_KTILE_FIRST_NOREQ ORR_T16_OPERANDS);
            else if constexpr (NT && ROWS) asm volatile(ORR_T16_KTILE_FIRST_REQ_NT ORR_T16_OPERANDS);
            else asm volatile(ORR_T16_KTILE_FIRST_REQ ORR_T16_OPERANDS);
        } else {
            if constexpr (!REQ) asm volatile(ORR_T16_KTILE_NOREQ ORR_T16_OPERANDS);
            else if constexpr (NT && ROWS) asm volatile(ORR_T16_KTILE_REQ_NT ORR_T16_OPERANDS);
            else asm volatile(ORR_T16_KTILE_REQ ORR_T16_OPERANDS);
        }
#undef ORR_T16_OPERANDS
    };
    using WaitMain = std::integral_constant<int, KP * (RING - 2)>;
    {                                                                      // (n_main >= 1: the launcher sends D / 64 <= kS4NB elsewhere)
        tile_iter(std::true_type{}, std::true_type{}, WaitMain{}, std::false_type{});
        for (int t = 1; t < n_main; ++t) {
            tile_iter(std::false_type{}, std::true_type{}, WaitMain{}, std::false_type{});
        }
        s_stage = (s_stage + n_main) % RING;
        // every K-tile of this output tile is requested now: the stream moves on to the workgroup's next one (the refill
        // behind the epilogue starts there)
        const int nid = next_id;                                           // (= valid_from(s_id + gridDim.x) without tickets: s_id == id here)
        if (nid < total_ids) { s_id = nid; s_k = 0; s_src = stream_src(nid); }
        else { s_k = T; s_src += (int64_t)n_main * kScImage; }
    }
    ORR_STAMP(7);
    // the last RING K-tiles: r-th of them leaves RING - 2 - r tiles in flight (the last one waits for nothing: 63).
    // The epilogue's loads (kEpiLoads vector-memory instructions per wave: 8 row constants x 2, 16 count words, the query
    // constants) go out in the tail K-tile kHookR, FOUR K-tiles before the end for the row requesters, three for the query
    // requesters (round 2: two): under a C3 launch the rows' constants took longer than two K-tiles to arrive and the epilogue
    // began with 2,000 cycles of waiting for them.  They are younger than every ring request still in flight, and a wave's
    // loads return in order, so the counted waits behind the hook simply allow kEpiLoads more to be outstanding.
    constexpr int kEpiLoads = 33, kHookR = RING >= 4 ? RING - 4 : 0;
    auto tail = [&](auto self, auto r_c) __attribute__((always_inline)) {
        constexpr int r = decltype(r_c)::value;
        if constexpr (r < RING) {
            constexpr int ring_left = r <= RING - 2 ? KP * (RING - 2 - r) : 63;
            constexpr int allowed = r > kHookR && !DOTS ? ring_left + kEpiLoads : ring_left;
            using WaitTail = std::integral_constant<int, (allowed > 63 ? 63 : allowed)>;
            using Hook = std::integral_constant<bool, r == kHookR>;
            tile_iter(std::false_type{}, std::false_type{}, WaitTail{}, Hook{});
            self(self, std::integral_constant<int, r + 1>{});
        }
    };
    tail(tail, std::integral_constant<int, 0>{});
    int tok_;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\tv_mov_b32 %0, 0" : "=v"(tok_) :: "memory");
    acc_token = tok_;     // the last fragment reads (into registers about to be reused), the last MFMAs' results; the token orders the epilogue's reads behind this
    };
    if (row_loader) k_loop(std::true_type{}); else k_loop(std::false_type{});
    ORR_STAMP(1);

    if constexpr (DOTS) {
        // element e of lane (c, g) of accumulator tile (i, j) = a[4 (8 i + j) + e] belongs to query 16 i + 4 g + e and row 16 j + c
        // of the wave's 128 x 128 tile (fused_epilogue16)
        const int c = tile16_c_of(lane_t), g = tile16_g_of(lane_t);
        int *Si = reinterpret_cast<int *>(S);
        static_for<8>([&](auto i_c) {
            static_for<8>([&](auto j_c) {
                static_for<4>([&](auto e_c) {
                    constexpr int i = decltype(i_c)::value, j = decltype(j_c)::value, e = decltype(e_c)::value;
                    const int qi = b0 + wr * 128 + 16 * i + 4 * g + e;
                    const int64_t col = n0 + wc * 128 + 16 * j + c;
                    const int v = acc16_as_int_here<4 * (8 * i + j) + e>(acc_token);
                    if (qi < B && col < n_rows) Si[(int64_t)qi * s_stride + col] = v;
                });
            });
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // (the stores: the counted waits of the next tile's K loop count requests only)
        if (dyn) {
            if (tid == 0) *mailbox = dyn_checked(ticket_id(ticket));
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (has_next) { if (row_loader) refill(std::true_type{}); else refill(std::false_type{}); }
    } else {
        uint32_t salt = 0;
        asm volatile("" : "+s"(salt));
        if (b0 + tid >= B) qf_mine = make_float4(0.f, __builtin_huge_valf(), 0.f, qf_mine.w);    // no query: floor +inf (.w: the batch's QW, as in every entry)
        EpiParked *queue = reinterpret_cast<EpiParked *>(lds_epi) + tid;
        float4 *qf_lds = reinterpret_cast<float4 *>(lds_epi + kS4Queue * 8 * 256);
        qf_lds[tid] = qf_mine;
        // every loaded register is touched here: the compiler's wait for the loads lands HERE, in front of the refill (further
        // down it would wait for the refill's requests as well, which are younger)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            asm volatile("" : "+v"(pre.rc[j].x), "+v"(pre.rc[j].y));
            asm volatile("" : "+v"(pre.rf[j].x), "+v"(pre.rf[j].y), "+v"(pre.rf[j].z), "+v"(pre.rf[j].w));
            asm volatile("" : "+v"(pre.w[0][j][0]), "+v"(pre.w[0][j][1]));
        }
        if (dyn) {                                                         // the ticket came back with the loads above: the id after next, for every wave
            asm volatile("" : "+v"(ticket));
            if (tid == 0) *mailbox = dyn_checked(ticket_id(ticket));
        }
        epilogue_issue_later_words16<BITS2 ? 2 : 4>(pre, b0 + wr * 128, n0 + wc * 128, B, n_rows, ep, lane_t);    // (they land under the first block's tests)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // (bare: __syncthreads() would also wait for outstanding requests)
        unsigned long long *st = (epi.stamps && wave == 0 && tile_seq < 64) ? epi.stamps + ((int64_t)blockIdx.x * 64 + tile_seq) * 8 : nullptr;
        // the ring refill for the next output tile rides on the epilogue: a quarter of this wave's requests in front of each
        // block of 32 queries (the last one takes what is left)
        auto refill_part = [&](int blk) __attribute__((always_inline)) {
            if (!has_next) return;
            auto part = [&](auto role) __attribute__((always_inline)) {
                constexpr int RING = decltype(role)::value ? kS4NB : kS4NA;
                const int t0 = blk * RING / 4, t1 = blk + 1 == 4 ? RING : (blk + 1) * RING / 4;
                for (int t = t0; t < t1; ++t) {
#pragma unroll
                    for (int sl = 0; sl < 8; ++sl) issue_piece(role, sl);
                    advance_stream();
                }
            };
            if (row_loader) part(std::true_type{}); else part(std::false_type{});
        };
        fused_epilogue16<kS16Queue, decltype(refill_part), BITS2 ? 2 : 4>(acc_token, b0 + wr * 128, n0 + wc * 128, B, n_rows, ep, lane_t, queue, 256, salt, st, pre, qf_lds + wr * 128, refill_part);
    }
#undef ORR_RD
#undef ORR_SB
    ORR_STAMP(2);
    if (FUSED && epi.stamps && tid == 0 && tile_seq < 64) epi.stamps[((int64_t)blockIdx.x * 64 + tile_seq) * 8 + 3] = __builtin_amdgcn_s_memrealtime();
    ++tile_seq;
    ring_a = (ring_a + T) % kS4NA;
    ring_b = (ring_b + T) % kS4NB;
    id = next_id;
    if (dyn) id_after = __builtin_amdgcn_readfirstlane(*mailbox);          // (written in front of this tile's epilogue barrier; rewritten behind the next tile's K loop)
    }
#undef ORR_STAMP
}

// K2g: the same screening pass for 1..8 queries -- no matrix core, a pure stream over the tiled shadow.
// Work unit = half a row tile (128 rows x D): per K-tile its 8 KiB are eight 1 KiB wave loads, lane l
// always holding chunk c = (l & 3) ^ ((l >> 4) & 3) of rows 16 j + (l >> 2), j = 0..7 (the swizzle of the
// stored layout does not depend on j), so one 16-byte read of the query's hi half from LDS serves all
// eight rows.  v_dot2c_f32_bf16 accumulates in fp32 (exact products, 3072 additions: the bound of the
// GEMM form holds, its order is not assumed).  Two register stages keep 16 KiB per wave in flight.
// Epilogue: quad reduction, then lane (quad, s) scores rows j = s and j = s + 4 in fp64 against the floor.
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));

template <int NQ>
__global__ __launch_bounds__(256, 2) void screen_gemv_bf16_kernel(const __bf16 *__restrict__ q_hi, int32_t D,
                                                                  const __bf16 *__restrict__ Eh, int64_t n_units,
                                                                  int64_t n_rows, FusedEpilogue epi)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lq[];     // [NQ][D] bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid * 8; i < NQ * D; i += 256 * 8)
        *reinterpret_cast<bf16x8 *>(lq + (size_t)i * 2) = *reinterpret_cast<const bf16x8 *>(q_hi + i);
    __syncthreads();
    const int c = (lane & 3) ^ ((lane >> 4) & 3);
    const int KT = D / kScBK;
    for (int64_t u = (int64_t)blockIdx.x * 4 + wave; u < n_units; u += (int64_t)gridDim.x * 4) {
        const __bf16 *base = Eh + ((u >> 1) * KT) * (int64_t)(kScImage / 2) + (u & 1) * (kScImage / 4) + lane * 8;
        float acc[NQ][8];
#pragma unroll
        for (int b = 0; b < NQ; ++b)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[b][j] = 0.f;
        bf16x8 s0[8], s1[8];
        auto load = [&](bf16x8 (&st)[8], int kt) {
            const __bf16 *p = base + (int64_t)kt * (kScImage / 2);
#pragma unroll
            for (int j = 0; j < 8; ++j) st[j] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(p + j * 512));
        };
        auto compute = [&](const bf16x8 (&st)[8], int kt) {
#pragma unroll
            for (int b = 0; b < NQ; ++b) {
                const bf16x8 qv = *reinterpret_cast<const bf16x8 *>(lq + ((size_t)b * D + kt * kScBK + c * 8) * 2);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bf16x2v ev = {st[j][2 * e], st[j][2 * e + 1]};
                        const bf16x2v qq = {qv[2 * e], qv[2 * e + 1]};
                        acc[b][j] = __builtin_amdgcn_fdot2_f32_bf16(ev, qq, acc[b][j], false);
                    }
                }
            }
        };
        load(s0, 0);
        for (int kt = 0; kt < KT; kt += 2) {                                // KT = D / 32 is even
            load(s1, kt + 1);
            compute(s0, kt);
            load(s0, kt + 2 < KT ? kt + 2 : KT - 1);                         // clamped, never branched around
            compute(s1, kt + 1);
        }
        // quad reduction: the four lanes of a quad hold the four chunks of the same rows
#pragma unroll
        for (int b = 0; b < NQ; ++b)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[b][j] += __shfl_xor(acc[b][j], 1, 64);
                acc[b][j] += __shfl_xor(acc[b][j], 2, 64);
            }
        const int sl = lane & 3;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int64_t row = (u >> 1) * kScBN + (u & 1) * 128 + 16 * (jj * 4 + sl) + (lane >> 2);
            if (row >= n_rows) continue;
            const double2 rc = epi.rowc[row];
#pragma unroll
            for (int b = 0; b < NQ; ++b) {
                const float a = sl == 0 ? acc[b][jj * 4] : sl == 1 ? acc[b][jj * 4 + 1] : sl == 2 ? acc[b][jj * 4 + 2] : acc[b][jj * 4 + 3];
                const QueryConst qc = epi.qc[b];
                const uint32_t mm = qc.n_terms > 0 ? kw_matches(epi.kw, b, (uint32_t)row) : 0u;
                unsigned long long key = score_key(fused_score_fast((double)a, rc.x, rc.y, mm, qc));
                if (!(__builtin_fabsf(a) <= 3.4028234663852886e38f)) key = ~0ull;   // never dropped: re-scored exactly later
                if (key > epi.tau[b]) {
                    const uint32_t slot = atomicAdd(&epi.cnt[b], 1u);
                    if (slot < epi.cap) {
                        SelEntry en;
                        en.key = key; en.pos = (uint32_t)row; en.pad = 0;
                        epi.buf[(int64_t)b * epi.cap + slot] = en;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// K2i: the streaming screen on an INT8 shadow (1..8 queries): a quarter of the fp32 bytes.
//
// Row r is stored as ie[r][k] = rint(e[r][k] / se_r), se_r = max_k |e[r][k]| / 127, in the same tiled
// layout (64 bytes = 64 k per image row, chunk c of 16 bytes in slot c ^ ((row >> 2) & 3)).  A query is
// split into TWO int8 components, q^ = s1 (iq1 + iq2 / 254) with iq1 = rint(q / s1), s1 = max|q| / 127,
// iq2 = rint((q - s1 iq1) / (s1 / 254)), so its quantisation error is ~254 times smaller than the row's.
// v_dot4c_i32_i8 sums exactly (|sum| <= 3072 * 127^2 < 2^31), hence  dot^ = s1 se_r (I1 + I2 / 254)  is
// the exact dot of the two quantised vectors and
//     |q.e - dot^|  <=  |q| |e - e^|  +  |q - q^| |e^|                       (Cauchy-Schwarz)
// with BOTH norms of differences computed when the vectors are quantised (fp64, rounded up): the bound is
// per (query,row) and data dependent -- about 0.8 % of |q||e| for Gaussian rows, everything for a row
// whose quantisation is poor (such a row simply always survives).  LOWER = true subtracts the bound
// (keys of the sampled prefix: their k-th best is a lower bound of the k-th best exact score), false adds
// it (screen: a pair is dropped only if even its upper bound stays below the floor).
// ---------------------------------------------------------------------------
struct I8Rows {
    const int8_t *tiled;        // [tiles][D/64][256][64]
    const float *scale;         // se_r
    const float *rel_err;       // |e - e^| / sqrt(normB), rounded up (0 where normB <= 0)
    const float *rel_hat;       // |e^| / sqrt(normB), rounded up
    const double *norm_b;       // not null: the epilogue forms the row's scoring constants itself (row_consts_of) from these
    const int64_t *created;
    int64_t now_ticks;
};
struct I8Queries {
    const int8_t *q1, *q2;      // [B][D]
    const float *s1;            // [B]
    const double *err2;         // [B] |q - q^|^2
};

// JR: 16-row groups per unit of work (one wave streams a unit's D columns in stages of eight 16-byte loads per lane:
// 8/JR k-tiles of JR groups).  8 for the pass over all rows (long sequential streams, few epilogues); 1 for the sampled
// prefix, whose few thousand rows then spread over hundreds of waves instead of a few dozen (2 x 4096 rows x 3072:
// 34 -> 12 us, and that launch is on the critical path of a one-query search).
// LISTS (with JR = 1 and floor keys of 0: the sampled prefix): a workgroup's 64 rows per query leave as one sorted list
// buf[b][64 l .. 64 l + 63], l = unit / 4, what buffer_to_lists would make of them -- no counters, no second launch.
template <int NQ, bool LOWER, int JR, bool LISTS = false>
__global__ __launch_bounds__(256, 2) void screen_gemv_i8_kernel(I8Queries Q, int32_t D, I8Rows R, int64_t n_units, int64_t n_rows,
                                                                FusedEpilogue epi)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lq[];     // [2][NQ][D] int8
    __shared__ SelEntry own[LISTS ? NQ : 1][kSelWidth];
    static_assert(!LISTS || JR == 1, "LISTS: one 16-row group per wave");
    constexpr int KS = 8 / JR;                     // k-tiles per stage
    constexpr int UPT = 16 / JR;                   // units per 256-row tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid * 16; i < NQ * D; i += 256 * 16) {
        *reinterpret_cast<uint4 *>(lq + i) = *reinterpret_cast<const uint4 *>(Q.q1 + i);
        *reinterpret_cast<uint4 *>(lq + NQ * D + i) = *reinterpret_cast<const uint4 *>(Q.q2 + i);
    }
    __syncthreads();
    const int c = (lane & 3) ^ ((lane >> 4) & 3);
    const int KT = D / 64;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    for (int64_t u = (int64_t)blockIdx.x * 4 + wave; u < n_units; u += (int64_t)gridDim.x * 4) {
        const int8_t *base = R.tiled + ((u / UPT) * KT) * (int64_t)(256 * 64) + (u % UPT) * (JR * 1024) + lane * 16;
        int a1[NQ][JR], a2[NQ][JR];
#pragma unroll
        for (int b = 0; b < NQ; ++b)
#pragma unroll
            for (int j = 0; j < JR; ++j) { a1[b][j] = 0; a2[b][j] = 0; }
        i32x4 s0[8], s1[8];
#define ORR_LOAD8(st, kt) { \
        _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) { \
            const int8_t *p_ = base + (int64_t)((kt) + ks) * (256 * 64); \
            _Pragma("unroll") for (int j = 0; j < JR; ++j) \
                st[ks * JR + j] = __builtin_nontemporal_load(reinterpret_cast<const i32x4 *>(p_ + j * 1024)); } }
#define ORR_DOT8(st, kt) { \
        _Pragma("unroll") for (int b = 0; b < NQ; ++b) { \
            _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) { \
                const i32x4 qa = *reinterpret_cast<const i32x4 *>(lq + (size_t)b * D + ((kt) + ks) * 64 + c * 16); \
                const i32x4 qb = *reinterpret_cast<const i32x4 *>(lq + (size_t)(NQ + b) * D + ((kt) + ks) * 64 + c * 16); \
                _Pragma("unroll") for (int j = 0; j < JR; ++j) { \
                    _Pragma("unroll") for (int w = 0; w < 4; ++w) { \
                        a1[b][j] = __builtin_amdgcn_sdot4(st[ks * JR + j][w], qa[w], a1[b][j], false); \
                        a2[b][j] = __builtin_amdgcn_sdot4(st[ks * JR + j][w], qb[w], a2[b][j], false); } } } } }
        ORR_LOAD8(s0, 0)
        for (int kt = 0; kt < KT; kt += 2 * KS) {                           // KT % (2 KS) == 0 (checked by the launcher)
            ORR_LOAD8(s1, kt + KS)
            ORR_DOT8(s0, kt)
            ORR_LOAD8(s0, (kt + 2 * KS < KT ? kt + 2 * KS : KT - KS))        // clamped, never branched around
            ORR_DOT8(s1, kt + KS)
        }
#undef ORR_DOT8
#undef ORR_LOAD8
#pragma unroll
        for (int b = 0; b < NQ; ++b)
#pragma unroll
            for (int j = 0; j < JR; ++j) {
                a1[b][j] += __shfl_xor(a1[b][j], 1, 64); a1[b][j] += __shfl_xor(a1[b][j], 2, 64);
                a2[b][j] += __shfl_xor(a2[b][j], 1, 64); a2[b][j] += __shfl_xor(a2[b][j], 2, 64);
            }
        const int sl = lane & 3;
#pragma unroll
        for (int jj = 0; jj < (JR + 3) / 4; ++jj) {
            if (jj * 4 + sl >= JR) continue;
            const int64_t row = (u / UPT) * kScBN + (u % UPT) * (16 * JR) + 16 * (jj * 4 + sl) + (lane >> 2);
            if (row >= n_rows) {
                if (LISTS) {
#pragma unroll
                    for (int b = 0; b < NQ; ++b) { own[b][wave * 16 + (lane >> 2)].key = 0ull; own[b][wave * 16 + (lane >> 2)].pos = 0xFFFFFFFFu; }
                }
                continue;
            }
            const double2 rc = R.norm_b ? row_consts_of(R.norm_b[row], R.created[row], R.now_ticks) : epi.rowc[row];
            const double se = (double)R.scale[row], re = (double)R.rel_err[row], rh = (double)R.rel_hat[row];
#pragma unroll
            for (int b = 0; b < NQ; ++b) {
                int i1 = a1[b][jj * 4], i2 = a2[b][jj * 4];
#pragma unroll
                for (int t = 1; t < 4; ++t)
                    if (jj * 4 + t < JR && sl == t) { i1 = a1[b][jj * 4 + t < JR ? jj * 4 + t : 0]; i2 = a2[b][jj * 4 + t < JR ? jj * 4 + t : 0]; }
                const QueryConst qc = epi.qc[b];
                const double dot = se * (double)Q.s1[b] * ((double)i1 + (double)i2 * (1.0 / 254.0));
                // |cos error| <= (|q|/sqrt(normA)) re + (|q - q^|/sqrt(normA)) rh ; the first factor is 1 up to 2^-23
                double err = 0.0;
                // ... plus 2^-23: the reference sums fp32-ROUNDED products (RecallSearchService.cs:79), the bound above is
                // about the real dot
                if (qc.use_cos) err = 0.7 * 1.000001 * (re * 1.0000003 + sqrt(Q.err2[b]) * qc.inv_sqrt_na * rh + 1.2e-7) + 1e-9;
                const uint32_t mm = qc.n_terms > 0 ? kw_matches(epi.kw, b, (uint32_t)row) : 0u;
                const double sc = fused_score_fast(dot, rc.x, rc.y, mm, qc) + (LOWER ? -err : err);
                unsigned long long key = score_key(sc);
                if (!(fabs(sc) <= 1.7976931348623157e308)) key = LOWER ? 1ull : ~0ull;     // non-finite: no floor from it / never dropped
                if (LISTS) {
                    own[b][wave * 16 + (lane >> 2)].key = key;
                    own[b][wave * 16 + (lane >> 2)].pos = (uint32_t)row;
                } else if (key > epi.tau[b]) {
                    const uint32_t slot = atomicAdd(&epi.cnt[b], 1u);
                    if (slot < epi.cap) {
                        SelEntry en;
                        en.key = key; en.pos = (uint32_t)row; en.pad = 0;
                        epi.buf[(int64_t)b * epi.cap + slot] = en;
                    }
                }
            }
        }
        if (LISTS) {                               // (n_units is a multiple of 4: a workgroup's waves leave the loop together)
            __syncthreads();
            if (wave < NQ) {
                unsigned long long k = own[wave][lane].key;
                uint32_t p = own[wave][lane].pos;
                wave_sort(k, p, lane);
                SelEntry o;
                o.key = k; o.pos = p; o.pad = 0;
                epi.buf[(int64_t)wave * epi.cap + (u / 4) * kSelWidth + lane] = o;
            }
            __syncthreads();
        }
    }
}

// One wave per row: scale, int8 image (tiled, swizzled), and the two relative norms the bound needs.
__global__ __launch_bounds__(256) void i8_shadow_kernel(const float *__restrict__ E, const double *__restrict__ norm_b, int64_t n_rows,
                                                        int64_t rows_padded, int32_t D, int8_t *__restrict__ out,
                                                        float *__restrict__ scale, float *__restrict__ rel_err, float *__restrict__ rel_hat)
{
    const int lane = threadIdx.x & 63;
    const int KT = D / 64;
    for (int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; row < rows_padded; row += ((int64_t)gridDim.x * blockDim.x) >> 6) {
        const int64_t tile = row >> 8;
        const int rr = (int)(row & 255);
        int8_t *dst = out + (tile * KT) * (int64_t)(256 * 64) + rr * 64;
        if (row >= n_rows) {                                                // padding rows of the last tile
            for (int k = lane * 4; k < D; k += 256)
                *reinterpret_cast<uint32_t *>(dst + (int64_t)(k >> 6) * (256 * 64) + ((((k & 63) >> 4) ^ ((rr >> 2) & 3)) << 4) + (k & 15)) = 0u;
            continue;
        }
        const float *src = E + row * (int64_t)D;
        float mx = 0.f;
        bool bad = false;
        for (int k = lane * 4; k < D; k += 256) {
            const float4 v = *reinterpret_cast<const float4 *>(src + k);
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
            bad = bad || !(fabsf(v.x) <= 3.4028234663852886e38f) || !(fabsf(v.y) <= 3.4028234663852886e38f) ||
                  !(fabsf(v.z) <= 3.4028234663852886e38f) || !(fabsf(v.w) <= 3.4028234663852886e38f);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
        bad = __any(bad);
        const float se = (bad || mx == 0.f) ? 0.f : mx / 127.f;
        const float inv = se > 0.f ? 1.f / se : 0.f;
        double d2 = 0.0, h2 = 0.0;
        for (int k = lane * 4; k < D; k += 256) {
            const float4 v = *reinterpret_cast<const float4 *>(src + k);
            const float x[4] = {v.x, v.y, v.z, v.w};
            uint32_t packed = 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int q = se > 0.f ? __float2int_rn(x[e] * inv) : 0;
                q = q > 127 ? 127 : (q < -127 ? -127 : q);
                const double hat = (double)se * (double)q;
                const double dl = (double)x[e] - hat;
                d2 += dl * dl;
                h2 += hat * hat;
                packed |= ((uint32_t)(uint8_t)(int8_t)q) << (8 * e);
            }
            *reinterpret_cast<uint32_t *>(dst + (int64_t)(k >> 6) * (256 * 64) + ((((k & 63) >> 4) ^ ((rr >> 2) & 3)) << 4) + (k & 15)) = packed;
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { d2 += __shfl_xor(d2, d, 64); h2 += __shfl_xor(h2, d, 64); }
        if (lane == 0) {
            const double nb = norm_b[row];
            scale[row] = se;
            if (!(nb > 0.0)) { rel_err[row] = 0.f; rel_hat[row] = 0.f; }     // cosine is 0 for this row whatever the dot (:84)
            else if (bad) { rel_err[row] = __builtin_huge_valf(); rel_hat[row] = 0.f; }   // never screened out
            else {
                rel_err[row] = __double2float_ru(sqrt(d2 / nb) * 1.000001);
                rel_hat[row] = __double2float_ru(sqrt(h2 / nb) * 1.000001);
            }
        }
    }
}

// Two-level int8 image of up to 8 queries (one workgroup each) and |q - q^|^2.
__global__ __launch_bounds__(256) void i8_queries_kernel(const float *__restrict__ Qf, int32_t D, int8_t *__restrict__ q1, int8_t *__restrict__ q2,
                                                         float *__restrict__ s1_out, double *__restrict__ err2_out,
                                                         double *__restrict__ err2_l1_out, uint32_t *__restrict__ zero, int32_t n_zero)
{
    // (the pass's counters, cleared by the first kernel of the call instead of a memset of their own)
    if (blockIdx.x == 0) for (int i = threadIdx.x; i < n_zero; i += 256) zero[i] = 0u;
    __shared__ float red_f[4];
    __shared__ double red_d[4], red_d1[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *q = Qf + (int64_t)b * D;
    float mx = 0.f;
    bool bad = false;
    for (int k = tid; k < D; k += 256) { const float v = q[k]; mx = fmaxf(mx, fabsf(v)); bad = bad || !(fabsf(v) <= 3.4028234663852886e38f); }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    if (lane == 0) red_f[wave] = (__any(bad) ? __builtin_huge_valf() : mx);
    bad = __any(bad);
    __syncthreads();
    mx = fmaxf(fmaxf(red_f[0], red_f[1]), fmaxf(red_f[2], red_f[3]));
    const bool usable = mx > 0.f && mx <= 3.4028234663852886e38f;
    const float s1 = usable ? mx / 127.f : 0.f, s2 = s1 / 254.f;
    double e2 = 0.0, e1 = 0.0;
    for (int k = tid; k < D; k += 256) {
        const float v = q[k];
        int a = usable ? __float2int_rn(v / s1) : 0;
        a = a > 127 ? 127 : (a < -127 ? -127 : a);
        const double r = (double)v - (double)s1 * (double)a;
        e1 += usable ? r * r : (double)v * (double)v;                           // one level only: q^ = s1 a (int8 GEMM)
        int c2 = (usable && s2 > 0.f) ? __double2int_rn(r / (double)s2) : 0;
        c2 = c2 > 127 ? 127 : (c2 < -127 ? -127 : c2);
        const double dl = r - (double)s1 * ((double)c2 / 254.0);               // q^ = s1 (a + c2 / 254)
        e2 += usable ? dl * dl : (double)v * (double)v;
        q1[(int64_t)b * D + k] = (int8_t)a;
        q2[(int64_t)b * D + k] = (int8_t)c2;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { e2 += __shfl_xor(e2, d, 64); e1 += __shfl_xor(e1, d, 64); }
    if (lane == 0) { red_d[wave] = e2; red_d1[wave] = e1; }
    __syncthreads();
    if (tid == 0) {
        s1_out[b] = s1;
        const double tot = red_d[0] + red_d[1] + red_d[2] + red_d[3];
        const double tot1 = red_d1[0] + red_d1[1] + red_d1[2] + red_d1[3];
        err2_out[b] = (mx <= 3.4028234663852886e38f) ? tot * 1.000001 : __builtin_huge_val();
        if (err2_l1_out) err2_l1_out[b] = (mx <= 3.4028234663852886e38f) ? tot1 * 1.000001 : __builtin_huge_val();
    }
}

// Tiled, pre-swizzled bf16 image of a row-major fp32 matrix X[n_rows][D] (see the header): output
// chunk o (16 bytes) = 8 consecutive k of one row.  rows_padded = n_tiles * 256.
__global__ __launch_bounds__(256) void bf16_tiled_kernel(const float *__restrict__ X, int64_t n_rows, int32_t D, int64_t n_chunks,
                                                         __bf16 *__restrict__ out)
{
    const int KT = D / kScBK;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < n_chunks; o += (int64_t)gridDim.x * blockDim.x) {
        const int slot = (int)(o & 3);
        const int rr = (int)((o >> 2) & 255);
        const int64_t tk = o >> 10;                                        // tile * KT + kt
        const int kt = (int)(tk % KT);
        const int64_t row = (tk / KT) * kScBN + rr;
        const int c = slot ^ ((rr >> 2) & 3);
        bf16x8 h;
        if (row < n_rows) {
            const float *src = X + row * (int64_t)D + kt * kScBK + c * 8;
            const float4 v0 = *reinterpret_cast<const float4 *>(src), v1 = *reinterpret_cast<const float4 *>(src + 4);
            h[0] = (__bf16)v0.x; h[1] = (__bf16)v0.y; h[2] = (__bf16)v0.z; h[3] = (__bf16)v0.w;
            h[4] = (__bf16)v1.x; h[5] = (__bf16)v1.y; h[6] = (__bf16)v1.z; h[7] = (__bf16)v1.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = (__bf16)0.f;
        }
        *reinterpret_cast<bf16x8 *>(out + o * 8) = h;
    }
}

}  // namespace

// Bytes of the tiled bf16 image of an [n_rows][D] matrix.
size_t bf16_tiled_bytes(int64_t n_rows, int32_t D)
{
    const int64_t tiles = (n_rows + kScBN - 1) / kScBN;
    return (size_t)tiles * kScBN * (size_t)D * sizeof(uint16_t);
}

// out = tiled, pre-swizzled bf16(X) (round to nearest even); D % 64 == 0.
hipError_t launch_bf16_tiled(const float *X, int64_t n_rows, int32_t D, void *out, hipStream_t s)
{
    if (n_rows <= 0) return hipSuccess;
    if (D <= 0 || D % 64 != 0) return hipErrorInvalidValue;
    const int64_t n_chunks = (int64_t)(bf16_tiled_bytes(n_rows, D) / 16);
    const int64_t blocks = std::min<int64_t>((n_chunks + 255) / 256, 65536);
    hipLaunchKernelGGL(bf16_tiled_kernel, dim3((unsigned)blocks), dim3(256), 0, s, X, n_rows, D, n_chunks, static_cast<__bf16 *>(out));
    return hipGetLastError();
}

// rowf[r] = {se_r, 0.7 (rel_err + 2^-22) rounded up, rel_hat, 0}: the row constants of the int8 GEMM's epilogue.
__global__ __launch_bounds__(256) void i8_rowf_kernel(const float *__restrict__ scale, const float *__restrict__ rel_err,
                                                      const float *__restrict__ rel_hat, int64_t n, float4 *__restrict__ out)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        float4 o;
        o.x = scale[r];
        o.y = __double2float_ru(0.7 * 1.000002 * ((double)rel_err[r] * 1.0000003 + 2.4e-7) + 1e-9);
        o.z = __double2float_ru((double)rel_hat[r] * 1.000001);
        o.w = 0.f;
        out[r] = o;
    }
}

hipError_t launch_i8_rowf(const float *scale, const float *rel_err, const float *rel_hat, int64_t n_rows, float4 *rowf, hipStream_t s)
{
    if (n_rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(i8_rowf_kernel, dim3((unsigned)std::min<int64_t>((n_rows + 255) / 256, 4096)), dim3(256), 0, s, scale, rel_err,
                       rel_hat, n_rows, rowf);
    return hipGetLastError();
}

// int8 queries [B][D] -> tiled, swizzled image [query tile][D/64][256][64] (rows past B are zero).
__global__ __launch_bounds__(256) void i8_tile_queries_kernel(const int8_t *__restrict__ q, int32_t B, int32_t D, int64_t n_chunks,
                                                              int8_t *__restrict__ out)
{
    const int KT = D / 64;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < n_chunks; o += (int64_t)gridDim.x * blockDim.x) {
        const int slot = (int)(o & 3);
        const int rr = (int)((o >> 2) & 255);
        const int64_t tk = o >> 10;
        const int kt = (int)(tk % KT);
        const int64_t row = (tk / KT) * kScBM + rr;
        const int c = slot ^ ((rr >> 2) & 3);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (row < B) v = *reinterpret_cast<const uint4 *>(q + row * (int64_t)D + kt * 64 + c * 16);
        *reinterpret_cast<uint4 *>(out + o * 16) = v;
    }
}

// Workgroups to start for the screening GEMM: one per CU (it holds all of a CU's LDS), each walking its share of
// the output tiles; a multiple of 8 keeps an id's XCD fixed along the walk.
// k_tiles = K-tiles per output tile (D / 64 for int8, D / 32 for bf16).
static int64_t screen_grid(int64_t tiles, int32_t k_tiles)
{
    if (k_tiles < kScNB) return tiles;        // the requests of a tile reach kScNB K-tiles ahead: never past the next output tile
    const int n_cu = device_cu_count();                                     // (of the CURRENT device: clusters launch on several)
    const int64_t cus = n_cu >= 8 ? (int64_t)(n_cu / 8 * 8) : 0;
    int64_t per_launch = cus;
    // ORR_SCREEN_GRID=n (diagnostic, read at every launch): n persistent workgroups instead of one per CU -- how the K loop's
    // cycles depend on how many CUs multiply at once, and the test that results do not depend on which
    // workgroup gets which output tiles
    if (const char *g = getenv("ORR_SCREEN_GRID")) { const int n = atoi(g); if (n >= 8 && n <= cus) per_launch = n / 8 * 8; }
    return per_launch > 0 && tiles > per_launch ? per_launch : tiles;
}

hipError_t launch_i8_tile_queries(const void *q1_linear, int32_t B, int32_t D, void *tiled, hipStream_t s)
{
    if (B <= 0) return hipSuccess;
    if (D <= 0 || D % 128 != 0) return hipErrorInvalidValue;
    const int64_t n_chunks = (int64_t)(i8_tiled_bytes(B, D) / 16);
    hipLaunchKernelGGL(i8_tile_queries_kernel, dim3((unsigned)std::min<int64_t>((n_chunks + 255) / 256, 65536)), dim3(256), 0, s,
                       static_cast<const int8_t *>(q1_linear), B, D, n_chunks, static_cast<int8_t *>(tiled));
    return hipGetLastError();
}

// The int8 screening GEMM over all rows with the fused epilogue (epi.i8_rowf / epi.i8_qs1 set).
hipError_t launch_screen_i8(const void *q_tiled, int32_t B, const void *e_tiled, int64_t n_rows, int32_t D,
                            const FusedEpilogue &epi, hipStream_t s, int64_t row_first)
{
    if (B <= 0 || n_rows <= row_first) return hipSuccess;
    if (D <= 0 || D % 128 != 0 || !epi.i8_rowf || !epi.i8_qs1 || row_first % kScBN != 0 || row_first < 0) return hipErrorInvalidValue;
    const int64_t n_ntiles = (n_rows - row_first + kScBN - 1) / kScBN;
    const int32_t n_mtiles = (B + kScBM - 1) / kScBM;
    const int64_t blocks = ((n_ntiles + 7) / 8) * 8 * n_mtiles;
    if (blocks > 0x7FFFFFFFll) return hipErrorInvalidValue;
    constexpr int flags = 1;                                               // rows of single-query-tile launches are requested non-temporal
    // ORR_SCREEN_STAMPS=file (diagnostic): every launch appends its workgroups' per-tile phase stamps to the file
    static const char *stamp_path = getenv("ORR_SCREEN_STAMPS");
    static std::atomic<unsigned long long *> d_stamps_of[256] = {};         // (one buffer per device: kernels write it where they run)
    constexpr size_t kStampWords = 256 * 64 * 8;
    FusedEpilogue epi_st = epi;
    unsigned long long *d_stamps = nullptr;
    if (stamp_path) {
        std::atomic<unsigned long long *> &slot = d_stamps_of[current_device_ordinal() & 255];
        d_stamps = slot.load();
        if (!d_stamps) {
            if (hipMalloc(reinterpret_cast<void **>(&d_stamps), kStampWords * 8) != hipSuccess) return hipErrorOutOfMemory;
            unsigned long long *expected = nullptr;
            if (!slot.compare_exchange_strong(expected, d_stamps)) { (void)hipFree(d_stamps); d_stamps = expected; }
        }
        (void)hipMemsetAsync(d_stamps, 0, kStampWords * 8, s);
        epi_st.stamps = d_stamps;
    }
    // ORR_SCREEN_TICKETS=0 (diagnostic, read at every launch): static assignment of the output tiles, for A/B runs
    if (const char *t = getenv("ORR_SCREEN_TICKETS")) { if (atoi(t) == 0) epi_st.tickets = nullptr; }
    const FusedEpilogue &epi_use = epi_st;
#define ORR_LAUNCH_I8(L) do { \
        const hipError_t attr = ensure_max_dynamic_lds<screen_bf16_kernel<true, true, L>>(kScLds); \
        if (attr != hipSuccess) return attr; \
        hipLaunchKernelGGL((screen_bf16_kernel<true, true, L>), dim3((unsigned)screen_grid(blocks, D / 64)), dim3(512), kScLds, s, \
                           static_cast<const __bf16 *>(q_tiled), B, static_cast<const __bf16 *>(e_tiled), row_first, n_rows, D, \
                           static_cast<float *>(nullptr), (int64_t)0, (int32_t)n_ntiles, n_mtiles, flags, epi_use); } while (0)
#define ORR_LAUNCH_I8W4(L, NT) do { \
        const hipError_t attr = ensure_max_dynamic_lds<screen_tile4_kernel<true, true, L, NT>>(kS4Lds); \
        if (attr != hipSuccess) return attr; \
        hipLaunchKernelGGL((screen_tile4_kernel<true, true, L, NT>), dim3((unsigned)screen_grid(blocks, D / 64)), dim3(256), kS4Lds, s, \
                           static_cast<const __bf16 *>(q_tiled), B, static_cast<const __bf16 *>(e_tiled), row_first, n_rows, D, \
                           static_cast<float *>(nullptr), (int64_t)0, (int32_t)n_ntiles, n_mtiles, flags, epi_use); } while (0)
    // From 65 queries up (two or more live 32-query tiles per wave... the matrix cores carry the launch) the four-wave form
    // of the tile; below, the eight-wave form (HBM-bound there, and its two waves per SIMD hide the epilogue's latencies).
    // Measured, 1M x 3072 rows: 128 queries 0.756 -> 0.619 ms, 256: 0.862 -> 0.842, 1024: 3.42 -> 3.30; 64: 0.485 vs 0.506.
    if (B > 64 && D / 64 >= kS4NB) {
#define ORR_LAUNCH_I8W16(NT) do { if (epi.count_bits == 2) ORR_LAUNCH_I8W16B(NT, true); else ORR_LAUNCH_I8W16B(NT, false); } while (0)
#define ORR_LAUNCH_I8W16B(NT, B2) do { \
        const hipError_t attr = ensure_max_dynamic_lds<screen_tile16_kernel<NT, false, B2>>(kS4Lds); \
        if (attr != hipSuccess) return attr; \
        hipLaunchKernelGGL((screen_tile16_kernel<NT, false, B2>), dim3((unsigned)screen_grid(blocks, D / 64)), dim3(256), kS4Lds, s, \
                           static_cast<const __bf16 *>(q_tiled), B, static_cast<const __bf16 *>(e_tiled), row_first, n_rows, D, \
                           static_cast<float *>(nullptr), (int64_t)0, (int32_t)n_ntiles, n_mtiles, flags, epi_use); } while (0)
        // 129+ queries: the tile on 16 x 16 x 64 MFMAs, unless the shard is so large that its epilogue's 32-bit word offsets
        // inside a pair of count planes would not do (then the 32 x 32 x 32 form).  Measured on one box, eight-wave form long
        // gone: 1M x 3072 rows x 256 queries 0.89 -> 0.85 ms, x 1024: 3.23 -> 2.92 ms; C3 (4 launches) 2.23 -> 2.13 ms each.
        const bool tile16 = screen_i8_uses_tile16(B, n_rows, D, epi.plane_stride) && epi.qf16 != nullptr;
        if (epi.count_bits == 2 && !tile16) return hipErrorInvalidValue;   // (only the 16 x 16 x 64 form reads two-bit count words)
        if (tile16 && B > 256) ORR_LAUNCH_I8W16(false);
        else if (tile16 && B > 128) ORR_LAUNCH_I8W16(true);
        else if (B > 256) ORR_LAUNCH_I8W4(8, false);
        else if (B > 128) ORR_LAUNCH_I8W4(8, true);
        else ORR_LAUNCH_I8W4(4, true);
#undef ORR_LAUNCH_I8W16B
#undef ORR_LAUNCH_I8W16
    }
    else if (B > 128) ORR_LAUNCH_I8(8);
    else if (B > 64) ORR_LAUNCH_I8(4);
    else if (B > 32) ORR_LAUNCH_I8(2);
    else ORR_LAUNCH_I8(1);
#undef ORR_LAUNCH_I8W4
#undef ORR_LAUNCH_I8
    if (stamp_path) {
        std::vector<unsigned long long> h(kStampWords);
        if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(h.data(), d_stamps, kStampWords * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE *f = fopen(stamp_path, "ab")) { fwrite(h.data(), 8, kStampWords, f); fclose(f); }
        }
    }
    return hipGetLastError();
}

bool screen_i8_uses_tile16(int32_t B, int64_t n_rows, int32_t D, int64_t plane_stride)
{
    return B > 128 && D > 0 && D % 128 == 0 && D / 64 > kS4NB && (int64_t)((B + 31) / 32) * plane_stride + n_rows < ((int64_t)1 << 30) &&
           n_rows < ((int64_t)1 << 28);
}

hipError_t launch_screen_i8_dots(const void *q_tiled, int32_t B, const void *e_tiled, int64_t n_rows, int32_t D, float *S,
                                 int64_t s_stride, hipStream_t s)
{
    if (B <= 0 || n_rows <= 0) return hipSuccess;
    if (D <= 0 || D % 128 != 0 || !S) return hipErrorInvalidValue;
    const int64_t n_ntiles = (n_rows + kScBN - 1) / kScBN;
    const int32_t n_mtiles = (B + kScBM - 1) / kScBM;
    const int64_t blocks = ((n_ntiles + 7) / 8) * 8 * n_mtiles;
    if (blocks > 0x7FFFFFFFll) return hipErrorInvalidValue;
    const FusedEpilogue none{};
#define ORR_LAUNCH_I8D(L) do { \
        const hipError_t attr = ensure_max_dynamic_lds<screen_bf16_kernel<false, true, L>>(kScLds); \
        if (attr != hipSuccess) return attr; \
        hipLaunchKernelGGL((screen_bf16_kernel<false, true, L>), dim3((unsigned)screen_grid(blocks, D / 64)), dim3(512), kScLds, s, \
                           static_cast<const __bf16 *>(q_tiled), B, static_cast<const __bf16 *>(e_tiled), (int64_t)0, n_rows, D, \
                           S, s_stride, (int32_t)n_ntiles, n_mtiles, 0, none); } while (0)
    if (B > 128) ORR_LAUNCH_I8D(8);
    else if (B > 64) ORR_LAUNCH_I8D(4);
    else if (B > 32) ORR_LAUNCH_I8D(2);
    else ORR_LAUNCH_I8D(1);
#undef ORR_LAUNCH_I8D
    return hipGetLastError();
}

// Diagnostic (orr_index_screen_i8_dots): S[b][r] = the RAW int32 accumulator of the int8 screening GEMM over rows [0, n_rows),
// computed by the given form of the kernel -- 0: eight-wave 32 x 32 x 32 (LIVE query tiles from B, as launch_screen_i8 picks
// them), 1: four-wave 32 x 32 x 32, 2: four-wave 16 x 16 x 64 -- with the SAME K loop, rings, requests and persistent walk of the
// output tiles as the fused launches (only the epilogue differs), so that the integer work is checkable bit for bit.
hipError_t launch_screen_i8_dots_raw(const void *q_tiled, int32_t B, const void *e_tiled, int64_t n_rows, int32_t D, int32_t *S,
                                     int64_t s_stride, int32_t form, bool nt_rows, hipStream_t s, uint32_t *tickets)
{
    if (B <= 0 || n_rows <= 0) return hipSuccess;
    if (D <= 0 || D % 128 != 0 || !S || form < 0 || form > 2) return hipErrorInvalidValue;
    const int64_t n_ntiles = (n_rows + kScBN - 1) / kScBN;
    const int32_t n_mtiles = (B + kScBM - 1) / kScBM;
    const int64_t blocks = ((n_ntiles + 7) / 8) * 8 * n_mtiles;
    if (blocks > 0x7FFFFFFFll) return hipErrorInvalidValue;
    FusedEpilogue none{};
    none.tickets = tickets;                                                // (form 2 draws its output tiles from them, as the fused launches do)
    if (const char *t = getenv("ORR_SCREEN_TICKETS")) { if (atoi(t) == 0) none.tickets = nullptr; }
    float *Sf = reinterpret_cast<float *>(S);
    constexpr int flags = 2;                                               // raw accumulators
    if (form == 0) {
#define ORR_LAUNCH_I8R(L) do { \
        const hipError_t attr = ensure_max_dynamic_lds<screen_bf16_kernel<false, true, L>>(kScLds); \
        if (attr != hipSuccess) return attr; \
        hipLaunchKernelGGL((screen_bf16_kernel<false, true, L>), dim3((unsigned)screen_grid(blocks, D / 64)), dim3(512), kScLds, s, \
                           static_cast<const __bf16 *>(q_tiled), B, static_cast<const __bf16 *>(e_tiled), (int64_t)0, n_rows, D, \
                           Sf, s_stride, (int32_t)n_ntiles, n_mtiles, flags | (nt_rows ? 1 : 0), none); } while (0)
        if (B > 128) ORR_LAUNCH_I8R(8);
        else if (B > 64) ORR_LAUNCH_I8R(4);
        else if (B > 32) ORR_LAUNCH_I8R(2);
        else ORR_LAUNCH_I8R(1);
#undef ORR_LAUNCH_I8R
        return hipGetLastError();
    }
    if (D / 64 <= kS4NB) return hipErrorInvalidValue;                      // (the four-wave forms need more K-tiles than their row ring holds)
#define ORR_LAUNCH_I8R4(L, NT) do { \
        const hipError_t attr = ensure_max_dynamic_lds<screen_tile4_kernel<false, true, L, NT>>(kS4Lds); \
        if (attr != hipSuccess) return attr; \
        hipLaunchKernelGGL((screen_tile4_kernel<false, true, L, NT>), dim3((unsigned)screen_grid(blocks, D / 64)), dim3(256), kS4Lds, s, \
                           static_cast<const __bf16 *>(q_tiled), B, static_cast<const __bf16 *>(e_tiled), (int64_t)0, n_rows, D, \
                           Sf, s_stride, (int32_t)n_ntiles, n_mtiles, flags, none); } while (0)
#define ORR_LAUNCH_I8R16(NT) do { \
        const hipError_t attr = ensure_max_dynamic_lds<screen_tile16_kernel<NT, true>>(kS4Lds); \
        if (attr != hipSuccess) return attr; \
        hipLaunchKernelGGL((screen_tile16_kernel<NT, true>), dim3((unsigned)screen_grid(blocks, D / 64)), dim3(256), kS4Lds, s, \
                           static_cast<const __bf16 *>(q_tiled), B, static_cast<const __bf16 *>(e_tiled), (int64_t)0, n_rows, D, \
                           Sf, s_stride, (int32_t)n_ntiles, n_mtiles, flags, none); } while (0)
    if (form == 1) {
        if (B <= 128) ORR_LAUNCH_I8R4(4, true);                            // (65..128 queries in the product; any B <= 128 here)
        else if (nt_rows) ORR_LAUNCH_I8R4(8, true);
        else ORR_LAUNCH_I8R4(8, false);
    } else {
        if (nt_rows) ORR_LAUNCH_I8R16(true);
        else ORR_LAUNCH_I8R16(false);
    }
#undef ORR_LAUNCH_I8R16
#undef ORR_LAUNCH_I8R4
    return hipGetLastError();
}

// linear [n_rows][D] int8 from the tiled, swizzled image (diagnostic)
__global__ __launch_bounds__(256) void i8_untile_kernel(const int8_t *__restrict__ tiled, int64_t n_rows, int32_t D, int64_t n_chunks,
                                                        int8_t *__restrict__ out)
{
    const int KT = D / 64;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < n_chunks; o += (int64_t)gridDim.x * blockDim.x) {
        const int slot = (int)(o & 3);
        const int rr = (int)((o >> 2) & 255);
        const int64_t tk = o >> 10;
        const int kt = (int)(tk % KT);
        const int64_t row = (tk / KT) * kScBN + rr;
        const int c = slot ^ ((rr >> 2) & 3);
        if (row < n_rows) *reinterpret_cast<uint4 *>(out + row * (int64_t)D + kt * 64 + c * 16) = *reinterpret_cast<const uint4 *>(tiled + o * 16);
    }
}

hipError_t launch_i8_untile(const void *tiled, int64_t n_rows, int32_t D, void *out_linear, hipStream_t s)
{
    if (n_rows <= 0) return hipSuccess;
    if (D <= 0 || D % 128 != 0) return hipErrorInvalidValue;
    const int64_t n_chunks = (int64_t)(i8_tiled_bytes(n_rows, D) / 16);
    hipLaunchKernelGGL(i8_untile_kernel, dim3((unsigned)std::min<int64_t>((n_chunks + 255) / 256, 65536)), dim3(256), 0, s,
                       static_cast<const int8_t *>(tiled), n_rows, D, n_chunks, static_cast<int8_t *>(out_linear));
    return hipGetLastError();
}

size_t i8_tiled_bytes(int64_t n_rows, int32_t D)
{
    return (size_t)((n_rows + kScBN - 1) / kScBN) * kScBN * (size_t)D;
}

// Int8 shadow of the sealed rows (K2i): image + per-row scale and relative norms.  D % 128 == 0.
hipError_t launch_i8_shadow(const float *E, const double *norm_b, int64_t n_rows, int32_t D, void *tiled, float *scale,
                            float *rel_err, float *rel_hat, hipStream_t s)
{
    if (n_rows <= 0) return hipSuccess;
    if (D <= 0 || D % 128 != 0) return hipErrorInvalidValue;
    const int64_t rows_padded = (n_rows + kScBN - 1) / kScBN * kScBN;
    const int64_t blocks = std::min<int64_t>((rows_padded + 3) / 4, 65536);
    hipLaunchKernelGGL(i8_shadow_kernel, dim3((unsigned)blocks), dim3(256), 0, s, E, norm_b, n_rows, rows_padded, D,
                       static_cast<int8_t *>(tiled), scale, rel_err, rel_hat);
    return hipGetLastError();
}

// ws: [q1 B*D][q2 B*D] int8, s1 [B] float, err2 [B] double -- see I8Queries.
hipError_t launch_i8_queries(const float *Q, int32_t B, int32_t D, void *q12, float *s1, double *err2, hipStream_t s, double *err2_level1,
                             uint32_t *zero, int32_t n_zero)
{
    if (B <= 0) return hipSuccess;
    int8_t *q1 = static_cast<int8_t *>(q12);
    hipLaunchKernelGGL(i8_queries_kernel, dim3((unsigned)B), dim3(256), 0, s, Q, D, q1, q1 + (size_t)B * D, s1, err2, err2_level1, zero, zero ? n_zero : 0);
    return hipGetLastError();
}

// The streaming screen over the int8 shadow for up to kMaxGemvScreenQ queries, kMaxI8ScreenQ per launch (two
// int32 accumulators per query and row).  lower_bound = true: keys are score - bound (prefix floor).
bool screen_gemv_i8_prefix_makes_lists(int32_t D) { return D % 1024 == 0; }

hipError_t launch_screen_gemv_i8(const void *q12, const float *s1, const double *err2, int32_t B, const void *tiled,
                                 const float *scale, const float *rel_err, const float *rel_hat, const double *norm_b,
                                 const int64_t *created, int64_t now_ticks, int64_t n_rows, int32_t D,
                                 const FusedEpilogue &epi, bool lower_bound, hipStream_t s)
{
    if (B <= 0 || n_rows <= 0) return hipSuccess;
    if (B > kMaxGemvScreenQ || D <= 0 || D % 128 != 0) return hipErrorInvalidValue;
    // the sampled prefix (lower_bound) is a few thousand rows on the critical path of the call: 16-row units spread it over
    // hundreds of waves; the pass over all rows streams 128-row units
    const bool fine = lower_bound && screen_gemv_i8_prefix_makes_lists(D);    // sorted lists of 64 into epi.buf (floor keys must be 0)
    const int64_t tiles = (n_rows + kScBN - 1) / kScBN;
    const int64_t n_units = tiles * (fine ? 16 : 2);
    const int64_t blocks = std::min<int64_t>((n_units + 3) / 4, 512);
    const int8_t *q1 = static_cast<const int8_t *>(q12), *q2 = q1 + (size_t)B * D;
    I8Rows Rd{static_cast<const int8_t *>(tiled), scale, rel_err, rel_hat, norm_b, created, now_ticks};
    for (int32_t b0 = 0; b0 < B; b0 += kMaxI8ScreenQ) {
        const int32_t nb = std::min<int32_t>(kMaxI8ScreenQ, B - b0);
        const size_t lds = 2 * (size_t)nb * (size_t)D;
        if (lds > 65536) return hipErrorInvalidValue;
        I8Queries Qd{q1 + (size_t)b0 * D, q2 + (size_t)b0 * D, s1 + b0, err2 + b0};
        FusedEpilogue e2 = epi;                              // everything the kernel indexes by query, shifted to b0
        e2.qc += b0; e2.tau += b0; if (e2.cnt) e2.cnt += b0; e2.buf += (size_t)b0 * epi.cap;
        if (e2.kw.q_term_off) e2.kw.q_term_off += b0;
#define ORR_I8(NQ) do { if (fine) hipLaunchKernelGGL((screen_gemv_i8_kernel<NQ, true, 1, true>), dim3((unsigned)blocks), dim3(256), lds, s, Qd, D, Rd, n_units, n_rows, e2); \
                        else if (lower_bound) hipLaunchKernelGGL((screen_gemv_i8_kernel<NQ, true, 8>), dim3((unsigned)blocks), dim3(256), lds, s, Qd, D, Rd, n_units, n_rows, e2); \
                        else hipLaunchKernelGGL((screen_gemv_i8_kernel<NQ, false, 8>), dim3((unsigned)blocks), dim3(256), lds, s, Qd, D, Rd, n_units, n_rows, e2); } while (0)
        switch (nb) {
        case 1: ORR_I8(1); break;
        case 2: ORR_I8(2); break;
        case 3: ORR_I8(3); break;
        default: ORR_I8(4); break;
        }
#undef ORR_I8
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// The screening pass for B <= kMaxGemvScreenQ queries: q_hi is the linear [B][D] bf16 image (hi halves of
// launch_split_queries), the fused epilogue is mandatory.
hipError_t launch_screen_gemv_bf16(const void *q_hi, int32_t B, const void *e_shadow, int64_t n_rows, int32_t D,
                                   const FusedEpilogue &epi, hipStream_t s)
{
    if (B <= 0 || n_rows <= 0) return hipSuccess;
    if (B > kMaxGemvScreenQ || D <= 0 || D % 64 != 0) return hipErrorInvalidValue;
    const int64_t n_units = ((n_rows + kScBN - 1) / kScBN) * 2;
    const int64_t blocks = std::min<int64_t>((n_units + 3) / 4, 512);
    const size_t lds = sizeof(uint16_t) * (size_t)B * (size_t)D;
    if (lds > 65536) return hipErrorInvalidValue;
    const __bf16 *qh = static_cast<const __bf16 *>(q_hi), *eh = static_cast<const __bf16 *>(e_shadow);
#define ORR_GEMV(NQ) hipLaunchKernelGGL(screen_gemv_bf16_kernel<NQ>, dim3((unsigned)blocks), dim3(256), lds, s, qh, D, eh, n_units, n_rows, epi)
    switch (B) {
    case 1: ORR_GEMV(1); break;
    case 2: ORR_GEMV(2); break;
    case 3: ORR_GEMV(3); break;
    case 4: ORR_GEMV(4); break;
    case 5: ORR_GEMV(5); break;
    case 6: ORR_GEMV(6); break;
    case 7: ORR_GEMV(7); break;
    default: ORR_GEMV(8); break;
    }
#undef ORR_GEMV
    return hipGetLastError();
}

// Rows [row_first, n_rows) of the tiled shadow against the tiled hi halves of the queries
// (launch_bf16_tiled of both).  epi == nullptr: dots to S; otherwise the fused epilogue.
// D % 64 == 0, row_first % 256 == 0.
hipError_t launch_screen_bf16(const void *q_tiled, int32_t B, const void *e_shadow, int64_t row_first, int64_t n_rows,
                              int32_t D, float *S, int64_t s_stride, const FusedEpilogue *epi, hipStream_t s)
{
    if (B <= 0 || n_rows <= row_first) return hipSuccess;
    if (D % 64 != 0 || D <= 0 || row_first % kScBN != 0) return hipErrorInvalidValue;
    const int64_t n_ntiles = (n_rows - row_first + kScBN - 1) / kScBN;
    const int32_t n_mtiles = (B + kScBM - 1) / kScBM;
    const int64_t blocks = ((n_ntiles + 7) / 8) * 8 * n_mtiles;
    if (blocks > 0x7FFFFFFFll) return hipErrorInvalidValue;
    const __bf16 *q_hi = static_cast<const __bf16 *>(q_tiled), *eh = static_cast<const __bf16 *>(e_shadow);
    const FusedEpilogue none{};
    constexpr int flags = 1;                                               // rows of single-query-tile launches are requested non-temporal
#define ORR_LAUNCH_LIVE(F, L, E) do { \
        const hipError_t attr = ensure_max_dynamic_lds<screen_bf16_kernel<F, false, L>>(kScLds); \
        if (attr != hipSuccess) return attr; \
        hipLaunchKernelGGL((screen_bf16_kernel<F, false, L>), dim3((unsigned)screen_grid(blocks, D / kScBK)), dim3(512), kScLds, s, q_hi, B, eh, row_first, n_rows, D, \
                           S, s_stride, (int32_t)n_ntiles, n_mtiles, flags, E); } while (0)
    if (epi) {
        if (B > 128) ORR_LAUNCH_LIVE(true, 8, *epi);
        else if (B > 64) ORR_LAUNCH_LIVE(true, 4, *epi);
        else if (B > 32) ORR_LAUNCH_LIVE(true, 2, *epi);
        else ORR_LAUNCH_LIVE(true, 1, *epi);
    } else {
        ORR_LAUNCH_LIVE(false, 8, none);
    }
#undef ORR_LAUNCH_LIVE
    return hipGetLastError();
}

}  // namespace orr
