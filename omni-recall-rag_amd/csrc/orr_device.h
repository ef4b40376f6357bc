// orr_device.h -- device-side scoring helpers shared by the kernels (orr_kernels.hip, orr_gemm.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "orr_kernels.h"

namespace orr {

// ---------------------------------------------------------------------------
// score <-> sortable key.  Larger key = ranks earlier.  double.CompareTo puts
// NaN below every number and treats -0 == +0.
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long score_key(double s)
{
    if (s != s) return 1ull;
    s = s + 0.0;                                   // -0 -> +0
    unsigned long long u = (unsigned long long)__double_as_longlong(s);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

__device__ __forceinline__ double key_score(unsigned long long k)
{
    if (k <= 1ull) return __longlong_as_double(0x7FF8000000000000ll);
    unsigned long long u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)u);
}

// RecallSearchService.cs:59-67 with the per-chunk pieces already reduced.
__device__ __forceinline__ double fused_score(double dot, double norm_b, int64_t created, uint32_t matches,
                                              const QueryConst &qc, int64_t now_ticks)
{
    double cosv = 0.0;
    if (qc.use_cos) {
        if (qc.norm_a <= 0.0 || norm_b <= 0.0)                               // :84-85
            cosv = 0.0;
        else
            cosv = dot / (sqrt(qc.norm_a) * sqrt(norm_b));                   // :87
    }
    double kw = qc.n_terms > 0 ? (double)matches / (double)qc.n_terms : 0.0;  // :112
    double total_days = (double)(now_ticks - created) / 864000000000.0;      // TimeSpan.TotalDays
    double age_days = total_days > 0.0 ? total_days : 0.0;                   // :117
    double rec = exp(-age_days / 30.0);                                      // :118
    return (cosv * 0.7) + (kw * 0.2) + (rec * 0.1);                          // :66
}

// Selection-only form of the fused score for batches: per-row pieces (recency * 0.1 and
// 1/sqrt(normB)) are computed once per batch (row_consts_kernel), per-query 1/sqrt(normA) on
// the host.  It differs from fused_score by a few ulp (reciprocal-multiply instead of divide),
// far inside the certificate's slack; the survivors are always re-scored exactly on the host.
__device__ __forceinline__ double fused_score_fast(double dot, double inv_sqrt_nb, double rec01, uint32_t matches,
                                                   const QueryConst &qc)
{
    double cosv = 0.0;
    if (qc.use_cos) cosv = (inv_sqrt_nb == 0.0) ? 0.0 : dot * (qc.inv_sqrt_na * inv_sqrt_nb);   // normB <= 0 -> 0 (:84)
    const double kw = (double)matches * qc.inv_n_terms;
    return (cosv * 0.7) + (kw * 0.2) + rec01;
}

// matches of RecallSearchService.cs:111 for (query b, row): how many of the query's
// distinct terms have their bit set in the per-term row bitmaps (see expand_hits_kernel).
__device__ __forceinline__ uint32_t kw_matches(const KwView &kw, int b, uint32_t row)
{
    uint32_t m = 0;
    const uint32_t t0 = kw.q_term_off[b], t1 = kw.q_term_off[b + 1];
    for (uint32_t i = t0; i < t1; ++i) {
        const uint32_t word = kw.bitmaps[(int64_t)kw.q_term_idx[i] * kw.words_per_term + (row >> 5)];
        m += (word >> (row & 31)) & 1u;
    }
    return m;
}

// ---------------------------------------------------------------------------
// Wave-resident top-64 list: lane i holds the i-th best entry (best first).
// "better" = larger key, then smaller candidate position (stable order).
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool better(unsigned long long ka, uint32_t pa, unsigned long long kb, uint32_t pb)
{
    return ka > kb || (ka == kb && pa < pb);
}

__device__ __forceinline__ void cmp_exchange(unsigned long long &k, uint32_t &p, int j, bool keep_better)
{
    const unsigned long long ok = __shfl_xor(k, j, 64);
    const uint32_t op = __shfl_xor(p, j, 64);
    const bool other_better = better(ok, op, k, p);
    if (other_better == keep_better) { k = ok; p = op; }
}

// Full bitonic sort of one entry per lane, best first.
__device__ __forceinline__ void wave_sort(unsigned long long &k, uint32_t &p, int lane)
{
#pragma unroll
    for (int kk = 2; kk <= 64; kk <<= 1) {
#pragma unroll
        for (int j = kk >> 1; j > 0; j >>= 1) {
            const bool best_first = (lane & kk) == 0;
            const bool lower = (lane & j) == 0;
            cmp_exchange(k, p, j, best_first == lower);
        }
    }
}

// Sorts a bitonic sequence (one entry per lane) best first.
__device__ __forceinline__ void wave_bitonic_merge(unsigned long long &k, uint32_t &p, int lane)
{
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) cmp_exchange(k, p, j, (lane & j) == 0);
}

// list (sorted best first) <- best 64 of list U other (sorted best first).
__device__ __forceinline__ void wave_merge_sorted(unsigned long long &k, uint32_t &p, unsigned long long ok,
                                                  uint32_t op, int lane)
{
    const unsigned long long rk = __shfl(ok, 63 - lane, 64);
    const uint32_t rp = __shfl(op, 63 - lane, 64);
    if (better(rk, rp, k, p)) { k = rk; p = rp; }
    wave_bitonic_merge(k, p, lane);
}

}  // namespace orr
