// orr_device.h -- device-side scoring helpers shared by the kernels (orr_kernels.hip, orr_gemm.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "orr_kernels.h"

namespace orr {

// A pointer the compiler has lost track of (it went through an opaque asm, or sits in a struct copied by value inside a
// persistent loop) is a GENERIC pointer to it: flat_load, which occupies the LDS path as well, counts on both waitcnt
// counters and takes a 64-bit address per lane.  Saying where it points restores global_load with a scalar base.
template <typename T> using gptr = const __attribute__((address_space(1))) T *;
template <typename T> __device__ __forceinline__ gptr<T> as_global(const T *p) { return (gptr<T>)p; }
// (HIP's float4 / double2 classes have no constructor from an address-space-qualified object: loads go through the
// native vector types of the same layout)
typedef float gf32x4 __attribute__((ext_vector_type(4)));
typedef float gf32x2 __attribute__((ext_vector_type(2)));
typedef double gf64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 load_global(const float4 *p, uint32_t i) { const gf32x4 v = as_global(reinterpret_cast<const gf32x4 *>(p))[i]; return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float2 load_global(const float2 *p, uint32_t i) { const gf32x2 v = as_global(reinterpret_cast<const gf32x2 *>(p))[i]; return make_float2(v.x, v.y); }
// The same from a wave-uniform base and a 32-bit BYTE offset per lane (the caller guarantees it fits): one load with a scalar
// base and a vector offset, no 64-bit address arithmetic per lane.
template <typename V>
__device__ __forceinline__ V load_global_at(const void *base, uint32_t byte_off)
{
    return *reinterpret_cast<const __attribute__((address_space(1))) V *>((const __attribute__((address_space(1))) char *)base + byte_off);
}
__device__ __forceinline__ float4 load_global_at_f4(const float4 *p, uint32_t byte_off) { const gf32x4 v = load_global_at<gf32x4>(p, byte_off); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ double2 load_global_at_d2(const double2 *p, uint32_t byte_off) { const gf64x2 v = load_global_at<gf64x2>(p, byte_off); double2 r; r.x = v.x; r.y = v.y; return r; }
__device__ __forceinline__ double2 load_global(const double2 *p, uint32_t i) { const gf64x2 v = as_global(reinterpret_cast<const gf64x2 *>(p))[i]; double2 r; r.x = v.x; r.y = v.y; return r; }

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

// Two-stage pass, floor: with s_k = the k-th best sampled score (key tau_k),
//   L = s_k - eps3           a lower bound of the exact k-th best score of the prefix, hence of the corpus
//   F = L - eps1 - margin    a row whose screening score is below F has an exact score below L
// floor_key = key(F) - 1 so that "key > floor_key" means "score >= F".
__device__ __forceinline__ void two_stage_floor_of(unsigned long long tau_k, double eps3, double eps1, unsigned long long *floor_key,
                                                   double *L_out)
{
    if (tau_k <= 1ull) {                    // fewer than k rows in the prefix, or a NaN: keep everything (-> overflow -> retry)
        *floor_key = 0ull;
        *L_out = -__builtin_huge_val();
        return;
    }
    const double sk = key_score(tau_k);
    const double L = sk - eps3;
    const double F = L - eps1 - 1e-9 * (1.0 + fabs(sk));
    const unsigned long long fk = score_key(F);
    *floor_key = fk > 2ull ? fk - 1ull : 0ull;
    *L_out = L;
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
// The per-row pieces of fused_score_fast: {1/sqrt(normB) or 0, recency * 0.1} (row_consts_kernel stores them once per
// batch; the streaming screen of 1..4 queries forms them where it needs them).
__device__ __forceinline__ double2 row_consts_of(double nb, int64_t created, int64_t now_ticks)
{
    const double total_days = (double)(now_ticks - created) / 864000000000.0;
    const double age_days = total_days > 0.0 ? total_days : 0.0;
    double2 o;
    o.x = nb <= 0.0 ? 0.0 : 1.0 / sqrt(nb);              // NaN stays NaN
    o.y = exp(-age_days / 30.0) * 0.1;
    return o;
}

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
        const uint32_t word = kw.bitmaps[kw_term_base(kw, kw.q_term_idx[i]) + (row >> 5)];
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

// One record of a query's result (orr_candidate), or the empty record where the selection ran out of rows.
__device__ __forceinline__ void write_record(orr_candidate *o, unsigned long long key, uint32_t pos, int b,
                                             int64_t row_base, const double *dot, const float *dotf, int64_t dot_stride,
                                             const double *norm_b, const int64_t *created, const int64_t *row_ids,
                                             const KwView &kw, int32_t dot_exact)
{
    orr_candidate c;
    if (key == 0ull) {
        c.approx_score = 0.0; c.dot = 0.0; c.norm_b = 0.0; c.created_ticks = 0;
        c.row_id = -1; c.order_key = -1; c.matches = 0; c.flags = 0;
    } else {
        c.approx_score = key_score(key);
        c.dot = dot ? dot[(int64_t)b * dot_stride + pos] : (dotf ? (double)dotf[(int64_t)b * dot_stride + pos] : 0.0);
        c.norm_b = norm_b[pos];
        c.created_ticks = created[pos];
        c.row_id = row_ids[pos];
        c.order_key = row_base + (int64_t)pos;
        c.matches = kw.bitmaps ? (int32_t)kw_matches(kw, b, pos) : 0;
        c.flags = dot_exact ? ORR_CAND_DOT_EXACT : 0;
    }
    *o = c;
}

}  // namespace orr
