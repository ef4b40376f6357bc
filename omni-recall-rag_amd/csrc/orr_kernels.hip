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

#include <hipcub/hipcub.hpp>

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

// ---------------------------------------------------------------------------
// K0 / K1e  exact dot: one row per lane, rows staged through a wave-private
// LDS tile so that HBM reads stay coalesced (4 rows x 256 B per wave
// instruction) while each lane walks its own row in index order.
//
// Tile: [64 rows][64 floats] = 16 KiB per wave, 16-byte chunks XOR-swizzled by
// (row & 15) so that the 16-lane groups of ds_read_b128 hit 16 distinct slots.
// ---------------------------------------------------------------------------
template <int NQ, bool SELF>
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

        for (int c0 = 0; c0 < D; c0 += 64) {
            float4 stage[16];
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int r = it * 4 + ld_row;
                const int64_t row = row0 + r;
                stage[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < n_rows)
                    stage[it] = *reinterpret_cast<const float4 *>(E + row * (int64_t)D + c0 + ld_ch * 4);
            }
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int r = it * 4 + ld_row;
                *reinterpret_cast<float4 *>(tile + r * 64 + ((ld_ch ^ (r & 15)) << 2)) = stage[it];
            }
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
    if (blocks > 256 * 8) blocks = 256 * 8;       // grid-stride beyond 8 workgroups per CU
    dim3 grid((unsigned)blocks), block(256);
    if (self_norm) {
        hipLaunchKernelGGL((dot_exact_tiled<1, true>), grid, block, 0, s, E, n_rows, D, Q, out, out_stride);
    } else {
        switch (nq) {
        case 1: hipLaunchKernelGGL((dot_exact_tiled<1, false>), grid, block, 0, s, E, n_rows, D, Q, out, out_stride); break;
        case 2: hipLaunchKernelGGL((dot_exact_tiled<2, false>), grid, block, 0, s, E, n_rows, D, Q, out, out_stride); break;
        case 3: hipLaunchKernelGGL((dot_exact_tiled<3, false>), grid, block, 0, s, E, n_rows, D, Q, out, out_stride); break;
        default: hipLaunchKernelGGL((dot_exact_tiled<4, false>), grid, block, 0, s, E, n_rows, D, Q, out, out_stride); break;
        }
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K3  keyword scan: one wave per row; each lane owns 16 consecutive content
// bytes per step and tests the 16 start positions in registers against the
// 4-byte prefix of every term (v_alignbyte windows); longer terms are verified
// byte-wise only on a prefix hit.  The content pool is over-allocated by 64
// bytes so the 20-byte reads never leave the allocation.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void keyword_scan_kernel(const uint8_t *__restrict__ pool,
                                                           const uint64_t *__restrict__ off, int64_t n_rows,
                                                           const uint8_t *__restrict__ term_pool,
                                                           const ScanTerm *__restrict__ terms, int32_t n_terms,
                                                           const uint32_t *__restrict__ q_term_off, int32_t B,
                                                           uint16_t *__restrict__ matches, int64_t matches_stride,
                                                           int32_t accumulate)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;

    for (int64_t row = wave_id; row < n_rows; row += n_waves) {
        const uint64_t start = off[row];
        const int64_t len = (int64_t)(off[row + 1] - start);
        const uint64_t abase = start & ~(uint64_t)15;
        const int64_t lead = (int64_t)(start - abase);        // bytes in front of the row in the first chunk
        unsigned long long found = 0ull;                      // wave-uniform bit per term

        for (int64_t cb = 0; cb < lead + len; cb += 1024) {
            const uint8_t *p = pool + abase + cb + lane * 16;
            const uint4 w = *reinterpret_cast<const uint4 *>(p);
            const uint32_t w4 = *reinterpret_cast<const uint32_t *>(p + 16);
            const uint32_t d[5] = {w.x, w.y, w.z, w.w, w4};
            const int64_t pos0 = cb + lane * 16 - lead;       // row-relative position of this lane's byte 0

            for (int t = 0; t < n_terms; ++t) {
                if ((found >> t) & 1ull) continue;
                const ScanTerm tm = terms[t];
                const int64_t last = len - (int64_t)tm.len;   // last valid start position
                bool hit = false;
                if (tm.len > 0 && last >= 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const uint32_t win = (i & 3) ? __builtin_amdgcn_alignbyte(d[(i >> 2) + 1], d[i >> 2], i & 3)
                                                     : d[i >> 2];
                        const int64_t pos = pos0 + i;
                        if (((win & tm.mask) == tm.prefix) && pos >= 0 && pos <= last) {
                            bool ok = true;
                            for (uint32_t k = 4; k < tm.len; ++k)
                                ok = ok && (pool[start + pos + k] == term_pool[tm.off + k]);
                            hit = hit || ok;
                        }
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

hipError_t launch_keyword_scan(const uint8_t *pool, const uint64_t *off, int64_t n_rows,
                               const uint8_t *term_pool, const ScanTerm *terms, int32_t n_terms,
                               const uint32_t *q_term_off, int32_t B, uint16_t *matches,
                               int64_t matches_stride, int32_t accumulate, hipStream_t s)
{
    if (n_rows <= 0 || B <= 0) return hipSuccess;
    if (n_terms > kMaxScanTerms || B > 64) return hipErrorInvalidValue;
    int64_t blocks = (n_rows + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(keyword_scan_kernel, dim3((unsigned)blocks), dim3(256), 0, s, pool, off, n_rows, term_pool,
                       terms, n_terms, q_term_off, B, matches, matches_stride, accumulate);
    return hipGetLastError();
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

// ---------------------------------------------------------------------------
// K4 + K5a  fused score and per-workgroup selection.  One workgroup scans
// kSelSegRows rows of one query: each wave walks a quarter of them 64 at a time
// (lane = row, coalesced 8-byte reads), skips batches that cannot enter its
// list, and the four lists are merged through LDS at the end.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fuse_select_kernel(const double *__restrict__ dot, int64_t dot_stride,
                                                          const double *__restrict__ norm_b,
                                                          const int64_t *__restrict__ created,
                                                          const uint16_t *__restrict__ matches,
                                                          int64_t matches_stride, const QueryConst *__restrict__ qcs,
                                                          int64_t now_ticks, int64_t n_rows,
                                                          SelEntry *__restrict__ out_sel)
{
    __shared__ SelEntry lists[4][kSelWidth];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const QueryConst qc = qcs[b];
    const int64_t seg0 = (int64_t)blockIdx.x * kSelSegRows;
    const int64_t seg1 = (seg0 + kSelSegRows < n_rows) ? seg0 + kSelSegRows : n_rows;

    unsigned long long k = 0ull;
    uint32_t p = 0xFFFFFFFFu;
    for (int64_t base = seg0 + wave * 64; base < seg1; base += 256) {
        const int64_t r = base + lane;
        unsigned long long nk = 0ull;
        uint32_t np = 0xFFFFFFFFu;
        if (r < seg1) {
            const double d = qc.use_cos ? dot[(int64_t)b * dot_stride + r] : 0.0;
            const uint32_t m = qc.n_terms > 0 ? matches[(int64_t)b * matches_stride + r] : 0u;
            nk = score_key(fused_score(d, norm_b[r], created[r], m, qc, now_ticks));
            np = (uint32_t)r;
        }
        const unsigned long long tk = __shfl(k, 63, 64);
        const uint32_t tp = __shfl(p, 63, 64);
        if (!__any(better(nk, np, tk, tp))) continue;
        wave_sort(nk, np, lane);
        wave_merge_sorted(k, p, nk, np, lane);
    }
    lists[wave][lane].key = k;
    lists[wave][lane].pos = p;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) wave_merge_sorted(k, p, lists[w][lane].key, lists[w][lane].pos, lane);
        SelEntry e;
        e.key = k; e.pos = p; e.pad = 0;
        out_sel[((int64_t)b * gridDim.x + blockIdx.x) * kSelWidth + lane] = e;
    }
}

hipError_t launch_fuse_select(const double *dot, int64_t dot_stride, const double *norm_b,
                              const int64_t *created, const uint16_t *matches, int64_t matches_stride,
                              const QueryConst *qc, int64_t now_ticks, int64_t n_rows, int32_t B,
                              SelEntry *out_sel, hipStream_t s)
{
    if (n_rows <= 0 || B <= 0) return hipSuccess;
    const int64_t n_seg = (n_rows + kSelSegRows - 1) / kSelSegRows;
    hipLaunchKernelGGL(fuse_select_kernel, dim3((unsigned)n_seg, (unsigned)B), dim3(256), 0, s, dot, dot_stride,
                       norm_b, created, matches, matches_stride, qc, now_ticks, n_rows, out_sel);
    return hipGetLastError();
}

__device__ __forceinline__ void write_record(orr_candidate *o, unsigned long long key, uint32_t pos, int b,
                                             int64_t row_base, const double *dot, int64_t dot_stride,
                                             const double *norm_b, const int64_t *created, const int64_t *row_ids,
                                             const uint16_t *matches, int64_t matches_stride, int32_t dot_exact)
{
    orr_candidate c;
    if (key == 0ull) {
        c.approx_score = 0.0; c.dot = 0.0; c.norm_b = 0.0; c.created_ticks = 0;
        c.row_id = -1; c.order_key = -1; c.matches = 0; c.flags = 0;
    } else {
        c.approx_score = key_score(key);
        c.dot = dot ? dot[(int64_t)b * dot_stride + pos] : 0.0;
        c.norm_b = norm_b[pos];
        c.created_ticks = created[pos];
        c.row_id = row_ids[pos];
        c.order_key = row_base + (int64_t)pos;
        c.matches = matches ? (int32_t)matches[(int64_t)b * matches_stride + pos] : 0;
        c.flags = dot_exact ? ORR_CAND_DOT_EXACT : 0;
    }
    *o = c;
}

// ---------------------------------------------------------------------------
// K5b  final merge per query: 16 waves fold the n_seg sorted lists, then a
// 4-level tree through LDS; wave 0 writes kprime records and the trailer.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void select_final_kernel(const SelEntry *__restrict__ sel, int32_t n_seg,
                                                            int32_t kprime, int64_t n_rows, int64_t row_base,
                                                            const double *__restrict__ dot, int64_t dot_stride,
                                                            const double *__restrict__ norm_b,
                                                            const int64_t *__restrict__ created,
                                                            const int64_t *__restrict__ row_ids,
                                                            const uint16_t *__restrict__ matches,
                                                            int64_t matches_stride, int32_t dot_exact,
                                                            orr_candidate *__restrict__ out)
{
    __shared__ SelEntry lists[16][kSelWidth];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int b = blockIdx.x;
    const SelEntry *mine = sel + (int64_t)b * n_seg * kSelWidth;

    unsigned long long k = 0ull;
    uint32_t p = 0xFFFFFFFFu;
    for (int sgi = wave; sgi < n_seg; sgi += 16) {
        const SelEntry e = mine[(int64_t)sgi * kSelWidth + lane];
        wave_merge_sorted(k, p, e.key, e.pos, lane);
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
        orr_candidate *o = out + (int64_t)b * (kprime + 1);
        if (lane < kprime)
            write_record(o + lane, k, p, b, row_base, dot, dot_stride, norm_b, created, row_ids, matches,
                         matches_stride, dot_exact);
        const unsigned long long valid_mask = __ballot(k != 0ull && lane < kprime);
        const int n_valid = __popcll(valid_mask);
        const unsigned long long worst_key = __shfl(k, (n_valid > 0 ? n_valid - 1 : 0), 64);
        if (lane == 0) {
            orr_candidate t;
            const bool kept_all = n_rows <= (int64_t)kprime;
            t.approx_score = (kept_all || n_valid == 0) ? -__builtin_huge_val() : key_score(worst_key);
            t.dot = 0.0; t.norm_b = 0.0; t.created_ticks = 0;
            t.row_id = -1; t.order_key = n_rows; t.matches = n_valid; t.flags = ORR_CAND_TRAILER;
            o[kprime] = t;
        }
    }
}

hipError_t launch_select_final(const SelEntry *sel, int32_t n_seg, int32_t B, int32_t kprime,
                               int64_t n_rows, int64_t row_base, const double *dot, int64_t dot_stride,
                               const double *norm_b, const int64_t *created, const int64_t *row_ids,
                               const uint16_t *matches, int64_t matches_stride, int32_t dot_exact,
                               orr_candidate *out, hipStream_t s)
{
    if (B <= 0) return hipSuccess;
    if (kprime < 1 || kprime > kSelWidth) return hipErrorInvalidValue;
    hipLaunchKernelGGL(select_final_kernel, dim3((unsigned)B), dim3(1024), 0, s, sel, n_seg, kprime, n_rows, row_base,
                       dot, dot_stride, norm_b, created, row_ids, matches, matches_stride, dot_exact, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Generic large-k path: all keys -> stable descending radix sort -> records.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void score_keys_kernel(const double *__restrict__ dot,
                                                         const double *__restrict__ norm_b,
                                                         const int64_t *__restrict__ created,
                                                         const uint16_t *__restrict__ matches, QueryConst qc,
                                                         int64_t now_ticks, int64_t n_rows,
                                                         unsigned long long *__restrict__ keys,
                                                         uint32_t *__restrict__ vals)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x) {
        const double d = qc.use_cos ? dot[r] : 0.0;
        const uint32_t m = qc.n_terms > 0 ? matches[r] : 0u;
        keys[r] = score_key(fused_score(d, norm_b[r], created[r], m, qc, now_ticks));
        vals[r] = (uint32_t)r;
    }
}

hipError_t launch_score_keys(const double *dot, const double *norm_b, const int64_t *created,
                             const uint16_t *matches, QueryConst qc, int64_t now_ticks, int64_t n_rows,
                             unsigned long long *keys, uint32_t *vals, hipStream_t s)
{
    if (n_rows <= 0) return hipSuccess;
    int64_t blocks = (n_rows + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(score_keys_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dot, norm_b, created, matches, qc,
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
                                                                  const double *__restrict__ dot,
                                                                  const double *__restrict__ norm_b,
                                                                  const int64_t *__restrict__ created,
                                                                  const int64_t *__restrict__ row_ids,
                                                                  const uint16_t *__restrict__ matches,
                                                                  int32_t dot_exact, orr_candidate *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K) {
        if ((int64_t)i < n_rows)
            write_record(out + i, keys[i], vals[i], 0, row_base, dot, 0, norm_b, created, row_ids, matches, 0, dot_exact);
        else
            write_record(out + i, 0ull, 0u, 0, row_base, dot, 0, norm_b, created, row_ids, matches, 0, dot_exact);
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
                                      int64_t n_rows, int64_t row_base, const double *dot,
                                      const double *norm_b, const int64_t *created, const int64_t *row_ids,
                                      const uint16_t *matches, int32_t dot_exact, orr_candidate *out,
                                      hipStream_t s)
{
    const int blocks = (K + 1 + 255) / 256;
    hipLaunchKernelGGL(records_from_sorted_kernel, dim3(blocks), dim3(256), 0, s, keys, vals, K, n_rows, row_base, dot,
                       norm_b, created, row_ids, matches, dot_exact, out);
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

__global__ __launch_bounds__(256) void gather_content_kernel(const uint8_t *__restrict__ src_pool,
                                                             const uint64_t *__restrict__ src_off,
                                                             uint8_t *__restrict__ dst_pool,
                                                             const uint64_t *__restrict__ dst_off,
                                                             const int64_t *__restrict__ perm, int64_t n)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave_id; r < n; r += n_waves) {
        const int64_t sr = perm[r];
        const uint64_t s0 = src_off[sr], len = src_off[sr + 1] - s0, d0 = dst_off[r];
        for (uint64_t i = lane; i < len; i += 64) dst_pool[d0 + i] = src_pool[s0 + i];
    }
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

hipError_t launch_gather_content(const uint8_t *src_pool, const uint64_t *src_off, uint8_t *dst_pool,
                                 const uint64_t *dst_off, const int64_t *perm, int64_t n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_content_kernel, dim3(capped_blocks(n, 4)), dim3(256), 0, s, src_pool, src_off, dst_pool,
                       dst_off, perm, n);
    return hipGetLastError();
}

hipError_t launch_iota_i64(int64_t *dst, int64_t n, int64_t base, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(iota_i64_kernel, dim3(capped_blocks(n, 256)), dim3(256), 0, s, dst, n, base);
    return hipGetLastError();
}

}  // namespace orr
