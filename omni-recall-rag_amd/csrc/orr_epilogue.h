// orr_epilogue.h -- scoring epilogue of the batched MFMA kernels (orr_gemm.hip, orr_screen.hip).
#pragma once

#include "orr_kernels.h"
#include "orr_device.h"

namespace orr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// Scoring epilogue shared by the batched kernels.  acc[i][j] are 32 x 32 accumulator tiles of
// v_mfma_f32_32x32x16_bf16: element e of lane (fr = lane & 31, fh = lane >> 5) belongs to query
// qbase + 32 i + (e & 3) + 8 (e >> 2) + 4 fh and to row colbase + 32 j + fr.
//
// Pass 1 (straight-line, fp32): the score with the exact keyword term -- match counts come
// bit-sliced from the count planes, one 32-bit word per (plane, 32 queries, row) -- against the
// query's floor minus a margin.  The few elements that pass are parked in a per-thread queue in
// LDS (the operand images are dead by now; the caller has put a workgroup barrier in between).
// Pass 2 (one copy of the code): fp64 score from the term bitmaps, compare with the floor key,
// atomic append to the query's buffer.  A non-finite accumulator (fp32 overflow) is never
// filtered.  A thread whose queue is full (a "hot" tile: e.g. the newest rows pass for every query)
// appends further elements straight to the buffers when RESCORED says that every buffer entry gets its
// exact key afterwards anyway (two-stage pass); otherwise it bumps the query's counter past the buffer
// capacity, which the host treats like any other overflow (the batch is repeated unfused).
constexpr int kEpiQueue = 8;            // parked elements per thread
struct EpiParked { float a; uint32_t idx; };

template <int NI, int NJ, bool RESCORED>
__device__ __forceinline__ void fused_epilogue(const f32x16 (&acc)[NI][NJ], int qbase, int64_t colbase, int32_t B,
                                               int64_t n_rows, const FusedEpilogue &epi, int lane, EpiParked *queue,
                                               int queue_stride, uint32_t idx_salt = 0)
{
    // idx_salt: zero, but opaque to the compiler when the caller runs this inside a loop over output tiles -- it
    // keeps the 128 constant tags of the parked entries from being hoisted out of that loop into registers.
    const int fr = lane & 31, fh = lane >> 5;
    float rb[NJ], rr[NJ], ea[NJ], eb[NJ];
    int64_t cols[NJ];
    bool ok[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        cols[j] = colbase + j * 32 + fr;                                   // this lane's row of E
        ok[j] = cols[j] < n_rows;
        const double2 rc = epi.rowc[ok[j] ? cols[j] : n_rows - 1];
        rb[j] = (float)rc.x;
        rr[j] = ok[j] ? (float)rc.y : -__builtin_huge_valf();              // rows past the end never pass
        ea[j] = 0.f; eb[j] = 0.f;
        if (epi.i8_rowf) {                                                 // int8 GEMM: acc is the integer dot
            const float4 rf = epi.i8_rowf[ok[j] ? cols[j] : n_rows - 1];
            rb[j] *= rf.x;
            ea[j] = rf.y; eb[j] = rf.z;
        }
    }
    const int32_t n_qg = (B + 31) >> 5;
    int parked = 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int qg = (qbase + i * 32) >> 5;
        const int qgc = qg < n_qg ? qg : n_qg - 1;                         // clamped, never branched around
        uint32_t w[NJ][kCountPlanes];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int p = 0; p < kCountPlanes; ++p)
                w[j][p] = epi.count_planes ? epi.count_planes[((int64_t)p * n_qg + qgc) * epi.plane_stride + (ok[j] ? cols[j] : n_rows - 1)] : 0u;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int bit = (e & 3) + 8 * (e >> 2) + 4 * fh;
            const int qi = qbase + i * 32 + bit;
            const float4 qf = epi.qf[qi < B ? qi : B - 1];                 // {0.7/sqrt(normA), floor - margin, 0.2/terms, -}
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                uint32_t m = 0;
#pragma unroll
                for (int p = 0; p < kCountPlanes; ++p) m |= ((w[j][p] >> bit) & 1u) << p;
                const float a = acc[i][j][e];
                const float upper = a * (qf.x * rb[j]) + rr[j] + (float)m * qf.z + (ea[j] + qf.w * eb[j]);   // the last term is 0 outside the int8 GEMM
                // NaN and overflowed sums are never dropped here
                const bool drop = (upper < qf.y && __builtin_fabsf(a) <= 3.4028234663852886e38f) || qi >= B || !ok[j];
                if (!drop) {
                    if (parked < kEpiQueue) {
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
    }
    if (parked > kEpiQueue) parked = kEpiQueue;
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

}  // namespace orr
