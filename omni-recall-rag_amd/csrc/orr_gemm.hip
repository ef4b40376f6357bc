// orr_gemm.hip -- batched candidate pass (K2) and exact re-score of the survivors (K6).
//
// K2b / K2s: approximate dots S[b][r] = sum_k Q[b][k] * E[r][k] for a whole batch on the matrix cores -- three bf16
// products of hi/lo splits (K2b, `gemm_dot_bf16x3`) or the f32-input MFMA in a streaming form for up to 32 queries per
// launch (K2s, `gemv_mfma`); each comes with a bound on |S - reference dot| (orr_api.hip).  They only have to be good
// enough to pick k' >= k candidates per query; K6 then recomputes the survivors' dots in the reference's own
// arithmetic (RecallSearchService.cs:77-82) and the host certifies the result against the cut-off (orr_api.hip).
// (The round-1 f32 MFMA GEMMs, 111 TFLOP/s at 256 queries, were superseded by the split-bf16 and int8 passes and
// are gone from this file; profiles/r01_pmc_mfma_gemm_b256_v1.json keeps their measurement.)
#include "orr_kernels.h"
#include "orr_device.h"
#include "orr_epilogue.h"

#include <algorithm>
#include <cstdio>
#include <vector>
#include <cstdlib>

namespace orr {

typedef float f32x4v __attribute__((ext_vector_type(4)));

// K depth of one step of the bf16x3 GEMM and the streaming MFMA kernels below
constexpr int kGemmBK = 64;


// ---------------------------------------------------------------------------
// K2b  split-bf16 candidate pass: every fp32 operand x is split on the fly into
// hi = bf16(x) and lo = bf16(x - hi); the dot is accumulated from the three
// products hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 (16x the f32 MFMA
// rate per instruction, 3 instructions -> 5.3x), reading the fp32 master copy
// once.  bf16 x bf16 products are exact in the fp32 accumulator; with u = 2^-8 the unit roundoff
// of one bf16 rounding,
//   |S - sum q_k e_k| <= [ 3.1 u^2        (dropped lo*lo and the second-order residuals of the split)
//                         + 3.06 D 2^-23  (3 D fp32 additions, one full ulp each: the matrix core's
//                                          internal order and rounding are not documented) ] * sum|q_k e_k|
// which is the epsilon the certificate is given (orr_api.hip).
//
// LDS: hi and lo images of both operand tiles, [rows][64 k] bf16 with a 144-byte row
// stride (conflict-free ds_read_b128 of the 8 consecutive k a lane's A/B fragment
// holds).  Workgroups that score the same rows against different query tiles (B > 256)
// are placed on one XCD back to back, so later reads of the row tile come from that
// XCD's L2 (default cache policy here).
// ---------------------------------------------------------------------------
constexpr int kBfLd = 72;        // bf16 elements per LDS row (64 + 8 pad)

__device__ __forceinline__ void split_write(__bf16 *hi_img, __bf16 *lo_img, int r, int c4, const float4 &v)
{
    const __bf16 h0 = (__bf16)v.x, h1 = (__bf16)v.y, h2 = (__bf16)v.z, h3 = (__bf16)v.w;
    bf16x4 h, l;
    h[0] = h0; h[1] = h1; h[2] = h2; h[3] = h3;
    l[0] = (__bf16)(v.x - (float)h0);
    l[1] = (__bf16)(v.y - (float)h1);
    l[2] = (__bf16)(v.z - (float)h2);
    l[3] = (__bf16)(v.w - (float)h3);
    *reinterpret_cast<bf16x4 *>(hi_img + r * kBfLd + 4 * c4) = h;
    *reinterpret_cast<bf16x4 *>(lo_img + r * kBfLd + 4 * c4) = l;
}

// Queries are split once per batch (split_queries_kernel); the row tile is split by the
// workgroup that stages it.  Tile: 256 queries x 128 rows x 64 k, 8 waves as 4 x 2, each a
// 64 x 64 sub-tile, so a row piece is converted once per 256 queries.
__device__ __forceinline__ void hi_write(__bf16 *hi_img, int r, int c4, const float4 &v)
{
    bf16x4 h;
    h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
    *reinterpret_cast<bf16x4 *>(hi_img + r * kBfLd + 4 * c4) = h;
}

__global__ __launch_bounds__(256) void split_queries_kernel(const float *__restrict__ Q, int64_t n,
                                                            __bf16 *__restrict__ q_hi, __bf16 *__restrict__ q_lo)
{
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
        const float4 v = *reinterpret_cast<const float4 *>(Q + i);
        const __bf16 h0 = (__bf16)v.x, h1 = (__bf16)v.y, h2 = (__bf16)v.z, h3 = (__bf16)v.w;
        bf16x4 h, l;
        h[0] = h0; h[1] = h1; h[2] = h2; h[3] = h3;
        l[0] = (__bf16)(v.x - (float)h0); l[1] = (__bf16)(v.y - (float)h1);
        l[2] = (__bf16)(v.z - (float)h2); l[3] = (__bf16)(v.w - (float)h3);
        *reinterpret_cast<bf16x4 *>(q_hi + i) = h;
        *reinterpret_cast<bf16x4 *>(q_lo + i) = l;
    }
}

constexpr int kBfBM = 256;

// NT = 32-column MFMA tiles per wave along the rows of E: the workgroup tile is 256 queries x
// (64 NT) rows.  NT = 2 (256 x 128, two-step prefetch) is the default; NT = 4 (256 x 256,
// ORR_BF16_TILE=256) halves the staging work per MFMA and measures the same 6.2 ms at B=256:
// both sit at 42 % MfmaUtil, the ceiling of a two-barriers-per-K-step structure
// (cdna_hip_programming.md, "the step-3 structure"); the next step is the 8-phase interleave.
// FUSED: instead of storing the dots, the epilogue scores every (query,row) of the tile
// (fused_score_fast: cosine from the accumulator, keyword bits, per-row recency) and appends
// the pairs that beat the query's floor key to that query's candidate buffer -- the scores
// never leave the CU.  The floor is the k'-th best key of an already scanned prefix, so about
// k' * rows / prefix entries per query get through.
// PROD = 3: hi*hi + hi*lo + lo*hi (the split pass).  PROD = 1: hi*hi only -- a plain bf16 pass
// with |S - sum q_k e_k| <= [2^-7 (1 + 2^-9) + 1.02 D 2^-23] sum|q_k e_k|, a third of the
// matrix work and half of the staging; only used as the wide first stage of the two-stage pass.
template <int NT, bool TWO_STAGE, bool FUSED, int PROD>
__global__ __launch_bounds__(512, 2) void gemm_dot_bf16x3_kernel(const __bf16 *__restrict__ Qh, const __bf16 *__restrict__ Ql,
                                                                 int32_t B, const float *__restrict__ E, int64_t row_first,
                                                                 int64_t n_rows, int32_t D, float *__restrict__ S,
                                                                 int64_t s_stride, int32_t n_ntiles, int32_t n_mtiles,
                                                                 FusedEpilogue epi)
{
    constexpr int BN = 64 * NT;
    constexpr int NB = BN / 32;                         // float4 row pieces of the E tile per thread
    // A_hi, A_lo: [256][72] bf16; B_hi, B_lo: [BN][72] bf16
    extern __shared__ __attribute__((aligned(16))) __bf16 img[];
    __bf16 *a_hi = img, *a_lo = img + kBfBM * kBfLd, *b_hi = img + 2 * kBfBM * kBfLd, *b_lo = b_hi + BN * kBfLd;
    if (PROD == 1) { b_hi = img + kBfBM * kBfLd; a_lo = nullptr; b_lo = nullptr; }     // hi images only
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware order: ids that differ by 8 share an XCD; the query tiles of one row tile are consecutive there
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int mt = slot % n_mtiles, nt = (slot / n_mtiles) * 8 + xcd;
    if (nt >= n_ntiles) return;
    const int64_t n0 = row_first + (int64_t)nt * BN;
    const int b0 = mt * kBfBM;

    // staging: A pieces are 16-byte (8 bf16) copies, 4 per image per thread; B pieces are float4
    const int la_r = tid >> 3, la_c = tid & 7;          // A: rows it*64 + la_r, 8-element chunk la_c
    const int lb_r = tid >> 4, lb_c = tid & 15;         // B: rows it*32 + lb_r, float4 chunk lb_c
    struct Stage { bf16x8 ah[4], al[4]; float4 b[NB]; };
    Stage st0, st1;
    auto load_stage = [&](Stage &st, int k0) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {                // clamped, never branched around
            const int r = it * 64 + la_r;
            const int qr = (b0 + r < B) ? b0 + r : B - 1;
            st.ah[it] = *reinterpret_cast<const bf16x8 *>(Qh + (int64_t)qr * D + k0 + la_c * 8);
            if (PROD == 3) st.al[it] = *reinterpret_cast<const bf16x8 *>(Ql + (int64_t)qr * D + k0 + la_c * 8);
        }
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int rr = it * 32 + lb_r;
            const int64_t er = (n0 + rr < n_rows) ? n0 + rr : n_rows - 1;
            st.b[it] = *reinterpret_cast<const float4 *>(E + er * (int64_t)D + k0 + lb_c * 4);
        }
    };
    auto store_stage = [&](const Stage &st) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = it * 64 + la_r;
            *reinterpret_cast<bf16x8 *>(a_hi + r * kBfLd + la_c * 8) = st.ah[it];
            if (PROD == 3) *reinterpret_cast<bf16x8 *>(a_lo + r * kBfLd + la_c * 8) = st.al[it];
        }
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            if (PROD == 3) split_write(b_hi, b_lo, it * 32 + lb_r, lb_c, st.b[it]);
            else hi_write(b_hi, it * 32 + lb_r, lb_c, st.b[it]);
        }
    };
    f32x16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;
    const int a_off = (wm * 64 + fr) * kBfLd + fh * 8;
    const int b_off = (wn * 32 * NT + fr) * kBfLd + fh * 8;

    auto mfma_step = [&]() {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {                            // 16 k per MFMA
            bf16x8 ah[2], al[2], bh[NT], bl[NT];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8 *>(a_hi + a_off + i * 32 * kBfLd + ks * 16);
                if (PROD == 3) al[i] = *reinterpret_cast<const bf16x8 *>(a_lo + a_off + i * 32 * kBfLd + ks * 16);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                bh[j] = *reinterpret_cast<const bf16x8 *>(b_hi + b_off + j * 32 * kBfLd + ks * 16);
                if (PROD == 3) bl[j] = *reinterpret_cast<const bf16x8 *>(b_lo + b_off + j * 32 * kBfLd + ks * 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            if (PROD == 3) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            }
        }
    };

    const int n_steps = D / kGemmBK;
    if (TWO_STAGE) {
        // two register sets: the loads of K-step t+2 are issued while step t is multiplied
        load_stage(st0, 0);
        if (n_steps > 1) load_stage(st1, kGemmBK);
        for (int t = 0; t < n_steps; t += 2) {
            __syncthreads();
            store_stage(st0);
            if (t + 2 < n_steps) load_stage(st0, (t + 2) * kGemmBK);
            __builtin_amdgcn_sched_barrier(0);                      // loads stay above the MFMA block
            __syncthreads();
            mfma_step();
            if (t + 1 < n_steps) {
                __syncthreads();
                store_stage(st1);
                if (t + 3 < n_steps) load_stage(st1, (t + 3) * kGemmBK);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                mfma_step();
            }
        }
    } else {
        load_stage(st0, 0);
        for (int t = 0; t < n_steps; ++t) {
            __syncthreads();
            store_stage(st0);
            if (t + 1 < n_steps) load_stage(st0, (t + 1) * kGemmBK);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            mfma_step();
        }
    }
    if (!FUSED) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int64_t col = n0 + wn * 32 * NT + j * 32 + fr;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = b0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                    if (row < B && col < n_rows) S[(int64_t)row * s_stride + col] = acc[i][j][e];
                }
            }
    } else {
        __syncthreads();                                                    // every wave is done with the operand images
        fused_epilogue<2, NT, PROD == 1>(acc, b0 + wm * 64, n0 + wn * 32 * NT, B, n_rows, epi, lane, reinterpret_cast<EpiParked *>(img) + tid, 512);
    }
}

// Count words: out[k][g][r], nibble n = the keyword match count of query 32 g + 8 k + n in row r
// (RecallSearchService.cs:111), so a lane of the scoring epilogue gets the counts of its 32
// queries for one row from kCountPlanes words, one v_bfe each.  Queries with more than 15 terms store 15
// where any term occurs (the pre-filter then grants them the full keyword credit, an upper bound).
// The counts are formed bit-sliced (plane p = bit p of the count) and re-packed into nibbles at the end.
// One wave handles 256 rows x 32 queries at a time: lane (q, h) adds the row bitmaps of query q's
// terms for the four bitmap words 8 pr + 4 h .. + 3 (one 16-byte load per term) bit-sliced -- 32 rows
// per word, ripple carry through the planes -- then each plane of each word, a 32 x 32 bit matrix with
// queries down the lanes and rows along the bits, is transposed across the lanes with five exchange
// steps, after which lane (r, h) holds the word of row r.  words_per_term % 4 == 0.
// BITS = 2: counts of 0..3 (the launcher's caller guarantees at most three terms per query): two planes, and two output words per
// (32 queries, row) -- word kk = queries 16 kk .. 16 kk + 15, two bits each.
template <int BITS>
__global__ __launch_bounds__(256) void query_count_planes_kernel(KwView kw, int32_t B, int64_t n_rows, int64_t plane_stride,
                                                                uint32_t *__restrict__ planes, int64_t oct_first, int64_t oct_end)
{
    constexpr int kCountPlanes = BITS;                                     // (shadows orr::kCountPlanes: planes of THIS form)
    const int g = blockIdx.y;
    const int32_t n_qg = gridDim.y;
    const int lane = threadIdx.x & 63, q = lane & 31, h = lane >> 5;
    const int b = g * 32 + q;
    uint32_t t0 = 0, t1 = 0;
    if (b < B) { t0 = kw.q_term_off[b]; t1 = kw.q_term_off[b + 1]; }
    const bool saturate = BITS == 4 && t1 - t0 > 15u;
    const int64_t n_quads = kw.words_per_term >> 2;
    const int64_t wave_id = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t pr = oct_first + wave_id; pr < oct_end; pr += n_waves) {
        const int64_t Q = pr * 2 + h;                                      // this half-wave's group of four words
        uint32_t c[4][kCountPlanes];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int p = 0; p < kCountPlanes; ++p) c[k][p] = 0u;
        if (Q < n_quads) {
            for (uint32_t i = t0; i < t1; ++i) {
                const uint4 w4 = *reinterpret_cast<const uint4 *>(kw.bitmaps + kw_term_base(kw, kw.q_term_idx[i]) + Q * 4);
                const uint32_t wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint32_t carry = wv[k];
                    if (saturate) { c[k][0] |= carry; continue; }
#pragma unroll
                    for (int p = 0; p < kCountPlanes; ++p) {
                        const uint32_t t = c[k][p] & carry;
                        c[k][p] ^= carry;
                        carry = t;
                    }
                }
            }
            if (saturate) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int p = 1; p < kCountPlanes; ++p) c[k][p] = c[k][0];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int p = 0; p < kCountPlanes; ++p) {
                uint32_t x = c[k][p];
#pragma unroll
                for (int s = 4; s >= 0; --s) {
                    const int jj = 1 << s;
                    const uint32_t low = s == 4 ? 0x0000FFFFu : s == 3 ? 0x00FF00FFu : s == 2 ? 0x0F0F0F0Fu : s == 1 ? 0x33333333u : 0x55555555u;
                    const uint32_t y = (uint32_t)__shfl_xor((int)x, jj, 64);
                    x = (q & jj) ? (((y & ~low) >> jj) | (x & ~low)) : ((x & low) | ((y & low) << jj));
                }
                c[k][p] = x;
            }
            const int64_t row = (Q * 4 + k) * 32 + q;                      // after the transpose this lane holds row q of the word
            if (BITS == 2) {
                if (row < n_rows) {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        uint32_t wv = 0u;
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            uint32_t x = (c[k][p] >> (16 * kk)) & 0xFFFFu;   // 16 queries' bits of plane p
                            x = (x | (x << 8)) & 0x00FF00FFu;
                            x = (x | (x << 4)) & 0x0F0F0F0Fu;
                            x = (x | (x << 2)) & 0x33333333u;
                            x = (x | (x << 1)) & 0x55555555u;              // bit n -> bit 2 n
                            wv |= x << p;
                        }
                        planes[((int64_t)kk * n_qg + g) * plane_stride + row] = wv;
                    }
                }
            } else if (row < n_rows) {
                // planes -> NIBBLES: word kk holds the counts of queries 8 kk .. 8 kk + 7 of the group, four bits each (bit p of a
                // nibble = plane p), so that the epilogue gets a pair's count with one v_bfe_u32 (orr_epilogue.h) instead of
                // gathering four plane bits
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    uint32_t wv = 0u;
#pragma unroll
                    for (int p = 0; p < kCountPlanes; ++p) {
                        uint32_t x = (c[k][p] >> (8 * kk)) & 0xFFu;          // 8 queries' bits of plane p
                        x = (x | (x << 12)) & 0x000F000Fu;
                        x = (x | (x << 6)) & 0x03030303u;
                        x = (x | (x << 3)) & 0x11111111u;                  // bit n -> bit 4 n
                        wv |= x << p;
                    }
                    planes[((int64_t)kk * n_qg + g) * plane_stride + row] = wv;
                }
            }
        }
    }
}

hipError_t launch_query_count_planes(KwView kw, int32_t B, int64_t n_rows, int64_t plane_stride, uint32_t *planes, hipStream_t s,
                                     int64_t row_first, int64_t row_end, int32_t bits)
{
    if (B <= 0 || !kw.bitmaps || n_rows <= 0) return hipSuccess;
    if (kw.words_per_term % 4 != 0 || row_first % 256 != 0) return hipErrorInvalidValue;
    const int64_t n_oct = (kw.words_per_term / 4 + 1) / 2;                 // 256 rows each
    const int64_t oct_first = row_first / 256;
    const int64_t oct_end = row_end < 0 ? n_oct : std::min<int64_t>(n_oct, (row_end + 255) / 256);
    if (oct_end <= oct_first) return hipSuccess;
    int64_t bx = (oct_end - oct_first + 3) / 4;                            // 4 waves per workgroup
    if (bx > 2048) bx = 2048;
    if (bits == 2)
        hipLaunchKernelGGL(query_count_planes_kernel<2>, dim3((unsigned)bx, (unsigned)((B + 31) / 32)), dim3(256), 0, s, kw, B, n_rows,
                           plane_stride, planes, oct_first, oct_end);
    else
        hipLaunchKernelGGL(query_count_planes_kernel<4>, dim3((unsigned)bx, (unsigned)((B + 31) / 32)), dim3(256), 0, s, kw, B, n_rows,
                           plane_stride, planes, oct_first, oct_end);
    return hipGetLastError();
}

// qf[b] = {0.7 / sqrt(normA) in fp32 (0 without cosine), floor score - margin, keyword credit per match, 0}
__device__ __forceinline__ float4 fused_query_const_of(const QueryConst &c, unsigned long long tau_b, const float *i8_qs1, const double *i8_qerr2, int b)
{
    float4 o;
    o.x = c.use_cos ? (float)(c.inv_sqrt_na * 0.7) : 0.f;
    if (tau_b <= 1ull) {
        o.y = -__builtin_huge_valf();                       // no floor (or a NaN floor): everything is tested exactly
    } else {
        const double floor_score = key_score(tau_b);
        // fp32 evaluation of cos*0.7 + kw*0.2 + rec*0.1 is off by < 1e-6 for scores of magnitude <= 1; 1e-5
        // margin, scaled up for larger magnitudes, and rounded down
        const double margin = 1e-5 * (1.0 + fabs(floor_score));
        o.y = __double2float_rd(floor_score - margin);
    }
    // the counts saturate at 15: beyond that a row with any match gets the full credit (15 * 0.2/15)
    o.z = c.n_terms > 0 ? __double2float_ru(0.2 / (double)(c.n_terms > 15 ? 15 : c.n_terms)) : 0.f;
    o.w = 0.f;
    if (i8_qs1) {                                           // int8 screening GEMM: the accumulator is an integer dot
        o.x *= i8_qs1[b];
        o.w = c.use_cos ? __double2float_ru(0.7 * 1.000001 * sqrt(i8_qerr2[b]) * c.inv_sqrt_na) : 0.f;
    }
    return o;
}

// ONE workgroup: the batch's largest (finite) query bound term is part of qf16.
__global__ __launch_bounds__(256) void fused_query_consts_kernel(const QueryConst *__restrict__ qc,
                                                                 const unsigned long long *__restrict__ tau, int32_t B,
                                                                 float4 *__restrict__ qf, const float *__restrict__ i8_qs1,
                                                                 const double *__restrict__ i8_qerr2, float4 *__restrict__ qf16,
                                                                 uint32_t *__restrict__ zero_a, int32_t n_a, uint32_t *__restrict__ zero_b,
                                                                 int32_t n_b)
{
    __shared__ float red[4];
    // (the pass's counters and the screening launches' tickets are cleared here: two memsets less on a chain whose launch
    // boundaries are what a mid-sized batch spends its time on)
    for (int i = threadIdx.x; i < n_a; i += 256) zero_a[i] = 0u;
    for (int i = threadIdx.x; i < n_b; i += 256) zero_b[i] = 0u;
    float wmax = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float4 o = fused_query_const_of(qc[b], tau[b], i8_qs1, i8_qerr2, b);
        qf[b] = o;
        const bool fin = o.x >= 0.f && o.x <= 1.0f && o.w >= 0.f && o.w <= 1e30f && o.z >= 0.f && o.z <= 1.f && o.y == o.y;
        if (fin) wmax = fmaxf(wmax, o.w);
    }
    if (!qf16) return;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, d, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = wmax;
    __syncthreads();
    wmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    for (int b = threadIdx.x; b < B; b += 256) {
        float4 o = fused_query_const_of(qc[b], tau[b], i8_qs1, i8_qerr2, b);
        const bool fin = o.x >= 0.f && o.x <= 1.0f && o.w >= 0.f && o.w <= 1e30f && o.z >= 0.f && o.z <= 1.f && o.y == o.y;
        if (!fin) { o.x = 0.f; o.y = -__builtin_huge_valf(); o.z = 0.f; }      // every pair of this query passes the pre-filter (tested exactly in pass 2)
        o.w = wmax;
        qf16[b] = o;
    }
}

hipError_t launch_fused_query_consts(const QueryConst *qc, const unsigned long long *tau, int32_t B, float4 *qf, hipStream_t s,
                                     const float *i8_qs1, const double *i8_qerr2, float4 *qf16, uint32_t *zero_a, int32_t n_a,
                                     uint32_t *zero_b, int32_t n_b)
{
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(fused_query_consts_kernel, dim3(1), dim3(256), 0, s, qc, tau, B, qf, i8_qs1, i8_qerr2, qf16, zero_a, zero_a ? n_a : 0,
                       zero_b, zero_b ? n_b : 0);
    return hipGetLastError();
}

template <int NT, bool TWO_STAGE, bool FUSED, int PROD>
static hipError_t launch_bf16x3_variant(const __bf16 *q_hi, const __bf16 *q_lo, int32_t B, const float *E, int64_t row_first,
                                        int64_t n_rows, int32_t D, float *S, int64_t s_stride, const FusedEpilogue &epi,
                                        hipStream_t s)
{
    constexpr int BN = 64 * NT;
    const int64_t n_ntiles = (n_rows - row_first + BN - 1) / BN;
    if (n_ntiles <= 0) return hipSuccess;
    const int32_t n_mtiles = (B + kBfBM - 1) / kBfBM;
    const int64_t blocks = ((n_ntiles + 7) / 8) * 8 * n_mtiles;
    if (blocks > 0x7FFFFFFFll) return hipErrorInvalidValue;
    constexpr size_t lds_bytes = sizeof(__bf16) * (PROD == 3 ? 2 : 1) * (kBfBM + BN) * kBfLd;
    const hipError_t attr = ensure_max_dynamic_lds<gemm_dot_bf16x3_kernel<NT, TWO_STAGE, FUSED, PROD>>((int)lds_bytes);     // per device
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((gemm_dot_bf16x3_kernel<NT, TWO_STAGE, FUSED, PROD>), dim3((unsigned)blocks), dim3(512), lds_bytes, s, q_hi, q_lo, B,
                       E, row_first, n_rows, D, S, s_stride, (int32_t)n_ntiles, n_mtiles, epi);
    return hipGetLastError();
}

hipError_t launch_split_queries(const float *Q, int32_t B, int32_t D, void *q_split_ws, hipStream_t s)
{
    if (B <= 0 || D <= 0) return hipSuccess;
    __bf16 *q_hi = static_cast<__bf16 *>(q_split_ws), *q_lo = q_hi + (size_t)B * D;
    const int64_t nq = (int64_t)B * D;
    hipLaunchKernelGGL(split_queries_kernel, dim3((unsigned)std::min<int64_t>((nq / 4 + 255) / 256, 2048)), dim3(256), 0, s,
                       Q, nq, q_hi, q_lo);
    return hipGetLastError();
}

// Rows [row_first, n_rows) of E against the pre-split queries.  epi == nullptr: dots to S;
// otherwise the fused scoring/filter epilogue (nothing is written to S).
hipError_t launch_gemm_dot_bf16x3(const void *q_split_ws, int32_t B, const float *E, int64_t row_first, int64_t n_rows, int32_t D,
                                  float *S, int64_t s_stride, const FusedEpilogue *epi, int32_t products, hipStream_t s)
{
    if (B <= 0 || n_rows <= row_first) return hipSuccess;
    if (D % kGemmBK != 0) return hipErrorInvalidValue;
    const __bf16 *q_hi = static_cast<const __bf16 *>(q_split_ws), *q_lo = q_hi + (size_t)B * D;
    const FusedEpilogue none{};
    if (epi && products == 1) return launch_bf16x3_variant<2, true, true, 1>(q_hi, q_lo, B, E, row_first, n_rows, D, S, s_stride, *epi, s);
    if (epi) return launch_bf16x3_variant<2, true, true, 3>(q_hi, q_lo, B, E, row_first, n_rows, D, S, s_stride, *epi, s);
    return launch_bf16x3_variant<2, true, false, 3>(q_hi, q_lo, B, E, row_first, n_rows, D, S, s_stride, none, s);
}

// Reference-order fp64 dots (RecallSearchService.cs:77-82) of one query against up to 64 gathered rows,
// one row per lane.  The rows are scattered, so there is nothing to coalesce: every lane reads its own row
// straight into registers, 64 columns (sixteen 16-byte loads, eight per 128-byte line) at a time, the next
// 64 already requested while the current ones go through the serial fp64 chain.  (The first version staged
// the pieces through a wave-private LDS tile as the contiguous-row kernel does: 100 us per wave at D = 3072
// against the chain's own ~15 us; ORR_RESCORE_LDS=1 keeps it for comparison.)
// rows[] (LDS, 64 entries, -1 = none) must be visible to the wave.  D % 64 == 0.
template <bool VIA_LDS>
__device__ __forceinline__ double exact_dot_of_gathered_rows(const float *__restrict__ E, int32_t D, const float *__restrict__ q,
                                                             const int64_t *rows, float *tile, int lane)
{
    double acc = 0.0;
    if (!VIA_LDS) {
        const int64_t row = rows[lane];
        const float *src = E + (row >= 0 ? row : 0) * (int64_t)D;              // lanes without a row read row 0 and are ignored
        float4 cur[16], nxt[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) cur[j] = *reinterpret_cast<const float4 *>(src + j * 4);
        for (int c0 = 0; c0 < D; c0 += 64) {
            const int cn = c0 + 64 < D ? c0 + 64 : c0;                         // clamped, never branched around
#pragma unroll
            for (int j = 0; j < 16; ++j) nxt[j] = *reinterpret_cast<const float4 *>(src + cn + j * 4);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float *qp = q + c0 + j * 4;
                float p0 = qp[0] * cur[j].x;
                acc += (double)p0;
                float p1 = qp[1] * cur[j].y;
                acc += (double)p1;
                float p2 = qp[2] * cur[j].z;
                acc += (double)p2;
                float p3 = qp[3] * cur[j].w;
                acc += (double)p3;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) cur[j] = nxt[j];
        }
        return acc;
    }
    const int ld_row = lane >> 4, ld_ch = lane & 15;
    float4 stage[16];
    auto load_stage = [&](int c0) {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int64_t row = rows[it * 4 + ld_row];
            stage[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row >= 0) stage[it] = *reinterpret_cast<const float4 *>(E + row * (int64_t)D + c0 + ld_ch * 4);
        }
    };
    load_stage(0);
    for (int c0 = 0; c0 < D; c0 += 64) {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int r = it * 4 + ld_row;
            *reinterpret_cast<float4 *>(tile + r * 64 + ((ld_ch ^ (r & 15)) << 2)) = stage[it];
        }
        if (c0 + 64 < D) load_stage(c0 + 64);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float4 e = *reinterpret_cast<const float4 *>(tile + lane * 64 + ((j ^ (lane & 15)) << 2));
            const float *qp = q + c0 + j * 4;
            float p0 = qp[0] * e.x;
            acc += (double)p0;
            float p1 = qp[1] * e.y;
            acc += (double)p1;
            float p2 = qp[2] * e.z;
            acc += (double)p2;
            float p3 = qp[3] * e.w;
            acc += (double)p3;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    return acc;
}

// Two-stage pass, second stage: every buffered (query,row) pair gets its score again from the
// reference-order fp64 dot (same wave-private swizzled tile walk as K6) and the exact fused
// formula; the entry's key is overwritten with it.  One wave per 64 buffer entries of a query.
template <bool VIA_LDS>
__global__ __launch_bounds__(64) void rescore_buffer_exact_kernel(const float *__restrict__ E, int32_t D,
                                                                  const float *__restrict__ Q,
                                                                  const double *__restrict__ norm_b,
                                                                  const int64_t *__restrict__ created, KwView kw,
                                                                  const QueryConst *__restrict__ qcs, int64_t now_ticks,
                                                                  const uint32_t *__restrict__ cnt, uint32_t cap,
                                                                  SelEntry *__restrict__ buf, double *__restrict__ buf_dot)
{
    __shared__ __attribute__((aligned(16))) float tile[64 * 64];
    __shared__ int64_t rows[64];
    const int lane = threadIdx.x, b = blockIdx.x;
    const uint32_t n = cnt[b] < cap ? cnt[b] : cap;
    const uint32_t first = blockIdx.y * 64u;
    if (first >= n) return;
    SelEntry *mine = buf + (int64_t)b * cap + first;
    const bool live = first + lane < n;
    const int64_t my_row = live ? (int64_t)mine[lane].pos : -1;
    rows[lane] = my_row;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double acc = exact_dot_of_gathered_rows<VIA_LDS>(E, D, Q + (int64_t)b * D, rows, tile, lane);
    if (live) {
        const QueryConst qc = qcs[b];
        const uint32_t m = qc.n_terms > 0 ? kw_matches(kw, b, (uint32_t)my_row) : 0u;
        QueryConst exact = qc;
        exact.use_cos = 1;                                   // this path only runs with cosine; guards are inside fused_score
        mine[lane].key = score_key(fused_score(acc, norm_b[my_row], created[my_row], m, exact, now_ticks));
        buf_dot[(int64_t)b * cap + first + lane] = acc;      // the records take it from here (records_dot_from_buffer)
    }
}

// Two-stage pass: the records' dots were already computed exactly for the whole buffer; one workgroup per
// query looks every record's row up in the query's buffer and copies the dot (replaces K6 for this pass).
__global__ __launch_bounds__(256) void records_dot_from_buffer_kernel(const SelEntry *__restrict__ buf, const double *__restrict__ buf_dot,
                                                                     const uint32_t *__restrict__ cnt, uint32_t cap, int32_t kprime,
                                                                     int64_t row_base, orr_candidate *__restrict__ recs)
{
    __shared__ uint32_t want[64];
    const int b = blockIdx.x;
    orr_candidate *mine = recs + (int64_t)b * (kprime + 1);
    if (threadIdx.x < 64) {
        uint32_t w = 0xFFFFFFFFu;
        if ((int)threadIdx.x < kprime && mine[threadIdx.x].row_id >= 0 && !(mine[threadIdx.x].flags & ORR_CAND_TRAILER))
            w = (uint32_t)(mine[threadIdx.x].order_key - row_base);
        want[threadIdx.x] = w;
    }
    __syncthreads();
    const uint32_t n = cnt[b] < cap ? cnt[b] : cap;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t pos = buf[(int64_t)b * cap + i].pos;
        for (int c = 0; c < kprime; ++c)
            if (want[c] == pos) {                              // a row occurs once in a buffer
                mine[c].dot = buf_dot[(int64_t)b * cap + i];
                mine[c].flags |= ORR_CAND_DOT_EXACT;
            }
    }
}

hipError_t launch_records_dot_from_buffer(const SelEntry *buf, const double *buf_dot, const uint32_t *cnt, uint32_t cap, int32_t B,
                                          int32_t kprime, int64_t row_base, orr_candidate *recs, hipStream_t s)
{
    if (B <= 0) return hipSuccess;
    if (kprime < 1 || kprime > 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(records_dot_from_buffer_kernel, dim3((unsigned)B), dim3(256), 0, s, buf, buf_dot, cnt, cap, kprime, row_base, recs);
    return hipGetLastError();
}

// Small batches, where the re-score's latency is on the critical path of the call (finish_survivors_kernel): FOUR lanes per
// survivor.  The sum's order is fixed (3072 dependent fp64 additions), but the row's bytes need not arrive in
// that rhythm, and the products (fp32 multiply, widen) are not part of the chain: lane c of a quad holds columns
// [256 r + 64 c, +64) of round r and forms its 64 products while the other three do the same; then the four
// lanes add theirs one after the other -- 64 dependent v_add_f64 each, nothing else in the chain -- the running
// sum handed on by a quad broadcast (DPP).  16 survivors per wave.  D % 256 == 0.
template <int S>
__device__ __forceinline__ double quad_broadcast(double v)
{
    constexpr int ctrl = S | (S << 2) | (S << 4) | (S << 6);              // quad_perm [S,S,S,S]
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, ctrl, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, ctrl, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

template <int S, int N>
__device__ __forceinline__ double quad_step(double acc, const double (&prod)[N], int c)
{
    if (c == S) {
#pragma unroll
        for (int i = 0; i < N; ++i) acc += prod[i];                        // the reference's order: N dependent additions
    }
    return quad_broadcast<S>(acc);
}

// ---- The reference's sum, out of order where that provably changes nothing --------------------------------------------------
// RecallSearchService.cs:77-82 adds the 3072 products one after the other into a double; re-doing that chain is 3072 dependent
// additions per survivor (quad_step above).  Rounding is the only thing that makes the order matter, and whether ANY order of a
// set of terms can round is decidable from two reductions: every product p_j is a float, i.e. an integer multiple of
// 2^(e_j - 23) (e_j its exponent, -126 for subnormals); with g = min_j (e_j - 23) every sum of every subset of the terms is a
// multiple of 2^g, and no such sum exceeds M = sum_j |p_j| in magnitude.  A multiple of 2^g below 2^(g + 53) is a double.  So if
// M < 2^(g + 53), every intermediate sum of EVERY order of additions is exact -- the reference's order included -- and all orders
// return the same bits: the exact sum.  (A running sum s carried in joins the terms with g_s = the exponent of its lowest set
// bit and |s| added to M.)  Unit-norm embeddings pass for ~94 % of rows (the products of a row span ~2^25); for the others the
// row is walked in slabs of 256 columns, each slab tested the same way against the running sum, and a slab that fails is added
// in the reference's order, one addition after the other (the only place a rounding can happen) -- so the result is the
// reference's for any input, infinities and NaNs included (their M is not below any limit).
// mn = min over the terms of (bits of |p| as a float) - 1 (0xFFFFFFFF: every term is zero); M = sum |p| (any order, rounded).
__device__ __forceinline__ bool any_order_is_exact(double s, double M, uint32_t mn)
{
    constexpr int kNone = 1 << 20;
    int g = kNone;
    if (mn != 0xFFFFFFFFu) {
        int field = (int)((mn + 1u) >> 23);
        field = field < 1 ? 1 : field;
        g = field - 127 - 23;
    }
    const unsigned long long sb = (unsigned long long)__double_as_longlong(s) & 0x7FFFFFFFFFFFFFFFull;
    if (sb != 0ull) {
        const int ef = (int)(sb >> 52);
        const unsigned long long m = (sb & 0xFFFFFFFFFFFFFull) | (ef ? 1ull << 52 : 0ull);
        const int gs = (ef ? ef : 1) - 1023 - 52 + (m ? __builtin_ctzll(m) : 0);
        g = gs < g ? gs : g;
    }
    if (g == kNone) return true;                                   // nothing but zeros
    int e = g + 53;
    if (e > 1023) e = 1023;
    if (e < -1000) return false;
    // 2^e (1 - 2^-20): the margin covers the roundings of M itself (a few dozen additions, 2^-53 each)
    const double lim = __longlong_as_double((long long)((unsigned long long)(e + 1023) << 52)) * 0.99999904632568359375;
    return __builtin_fabs(s) + M <= lim;                           // (false for NaN and for infinities)
}

// Three reductions at once (their shuffles overlap): sum, sum, min over the lanes whose index differs in the bits of `mask`
// (63: the wave; 15: each row of 16 lanes).
__device__ __forceinline__ void lanes_reduce(double &t, double &a, uint32_t &m, int mask)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        if (!(mask & off)) continue;
        const double to = __shfl_xor(t, off, 64), ao = __shfl_xor(a, off, 64);
        const uint32_t mo = (uint32_t)__shfl_xor((int)m, off, 64);
        t += to;
        a += ao;
        m = mo < m ? mo : m;
    }
}

__device__ __forceinline__ double uniform_f64(double v)
{
    // (lanes add in different orders: identical when the sums are exact, and made identical for the comparison otherwise)
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

__device__ __forceinline__ double lane_f64(double v, int lane_index)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane_index), __builtin_amdgcn_readlane(__double2loint(v), lane_index));
}

// One wave, one row; lane l holds columns [256 k + 4 l, +4) of slab k (coalesced 1 KB loads), the query's D floats are in LDS.
// CH slabs of a row are in registers at a time (CH divides D / 256: 12 = a whole row of 3072; no predicates, so the loads of the
// next chunk stay in flight behind counted waits while this one is added).
template <int CH>
__device__ __forceinline__ void load_chunk(float4 (&dst)[CH], const float *row, int chunk, int lane)
{
    const float4 *r4 = reinterpret_cast<const float4 *>(row) + lane + (int64_t)chunk * (CH * 64);
#pragma unroll
    for (int u = 0; u < CH; ++u) dst[u] = r4[u * 64];
}

// T: this lane's sum (double, additions in any order), M: its sum of magnitudes in fp32 (rounded: the caller widens the limit by
// 2^-16), mn2: min over its nonzero products of (bits << 1) - 1.
template <int CH>
__device__ __forceinline__ void add_chunk(const float4 (&cur)[CH], const float *q_lds, int chunk, int lane, double &T, float &M,
                                          uint32_t &mn2)
{
    const float4 *q4 = reinterpret_cast<const float4 *>(q_lds) + lane + chunk * (CH * 64);
#pragma unroll
    for (int u = 0; u < CH; ++u) {
        const float4 qv = q4[u * 64];
        const float p0 = qv.x * cur[u].x, p1 = qv.y * cur[u].y, p2 = qv.z * cur[u].z, p3 = qv.w * cur[u].w;
        T += (double)p0; T += (double)p1; T += (double)p2; T += (double)p3;
        M += __builtin_fabsf(p0); M += __builtin_fabsf(p1); M += __builtin_fabsf(p2); M += __builtin_fabsf(p3);
        const uint32_t b0 = (__float_as_uint(p0) << 1) - 1u, b1 = (__float_as_uint(p1) << 1) - 1u;
        const uint32_t b2 = (__float_as_uint(p2) << 1) - 1u, b3 = (__float_as_uint(p3) << 1) - 1u;
        const uint32_t m01 = b0 < b1 ? b0 : b1, m23 = b2 < b3 ? b2 : b3, m4 = m01 < m23 ? m01 : m23;
        mn2 = m4 < mn2 ? m4 : mn2;
    }
}

// Sum / min over the wave with DPP rotations inside each row of 16 lanes, the four rows joined through scalar registers; every
// lane gets the result.
template <int ROR>
__device__ __forceinline__ uint32_t row_ror_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + ROR, 0xF, 0xF, false);
}

__device__ __forceinline__ double wave_total_f64(double v)
{
#define ORR_STEP(R)                                                                                              \
    v += __hiloint2double((int)row_ror_u32<R>((uint32_t)__double2hiint(v)), (int)row_ror_u32<R>((uint32_t)__double2loint(v)))
    ORR_STEP(8); ORR_STEP(4); ORR_STEP(2); ORR_STEP(1);
#undef ORR_STEP
    return (lane_f64(v, 0) + lane_f64(v, 16)) + (lane_f64(v, 32) + lane_f64(v, 48));
}

__device__ __forceinline__ float wave_total_f32(float v)
{
    v += __uint_as_float(row_ror_u32<8>(__float_as_uint(v)));
    v += __uint_as_float(row_ror_u32<4>(__float_as_uint(v)));
    v += __uint_as_float(row_ror_u32<2>(__float_as_uint(v)));
    v += __uint_as_float(row_ror_u32<1>(__float_as_uint(v)));
    const float a = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 0));
    const float b = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 16));
    const float c = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 32));
    const float d = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 48));
    return (a + b) + (c + d);
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
    uint32_t o;
    o = row_ror_u32<8>(v); v = o < v ? o : v;
    o = row_ror_u32<4>(v); v = o < v ? o : v;
    o = row_ror_u32<2>(v); v = o < v ? o : v;
    o = row_ror_u32<1>(v); v = o < v ? o : v;
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    const uint32_t ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}

// The rows that do not pass as a whole: slab by slab (256 columns) against the running sum, a slab that does not pass in four
// blocks of 64 columns (the lanes of one row of 16), a block that does not pass in the reference's order, one addition after
// the other (the products handed over by readlane) -- the only place a rounding can happen.  CH slabs are loaded at a time.
// Every lane returns the same value.
template <int CH>
__device__ __noinline__ double exact_dot_by_slabs(const float *__restrict__ row, const float *q_lds, int32_t D, int lane)
{
    const int chunks = (D >> 8) / CH;
    const float4 *q4 = reinterpret_cast<const float4 *>(q_lds) + lane;
    double s = 0.0;
    for (int c = 0; c < chunks; ++c) {
        float4 cv[CH];
        load_chunk<CH>(cv, row, c, lane);
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const float4 qv = q4[(c * CH + u) * 64];
            const float p0 = qv.x * cv[u].x, p1 = qv.y * cv[u].y, p2 = qv.z * cv[u].z, p3 = qv.w * cv[u].w;
            const double d0 = (double)p0, d1 = (double)p1, d2 = (double)p2, d3 = (double)p3;
            double t = 0.0;
            t += d0; t += d1; t += d2; t += d3;
            double a = __builtin_fabs(d0);
            a += __builtin_fabs(d1); a += __builtin_fabs(d2); a += __builtin_fabs(d3);
            const uint32_t b0 = (__float_as_uint(p0) & 0x7FFFFFFFu) - 1u, b1 = (__float_as_uint(p1) & 0x7FFFFFFFu) - 1u;
            const uint32_t b2 = (__float_as_uint(p2) & 0x7FFFFFFFu) - 1u, b3 = (__float_as_uint(p3) & 0x7FFFFFFFu) - 1u;
            const uint32_t m01 = b0 < b1 ? b0 : b1, m23 = b2 < b3 ? b2 : b3;
            uint32_t m = m01 < m23 ? m01 : m23;
            lanes_reduce(t, a, m, 15);                             // per block of 64 columns
            double tb[4], ab[4];
            uint32_t mb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                tb[q] = lane_f64(t, 16 * q);
                ab[q] = lane_f64(a, 16 * q);
                mb[q] = (uint32_t)__builtin_amdgcn_readlane((int)m, 16 * q);
            }
            const double ts = (tb[0] + tb[1]) + (tb[2] + tb[3]), as = (ab[0] + ab[1]) + (ab[2] + ab[3]);
            const uint32_t m0 = mb[0] < mb[1] ? mb[0] : mb[1], m1 = mb[2] < mb[3] ? mb[2] : mb[3], ms = m0 < m1 ? m0 : m1;
            if (any_order_is_exact(s, as, ms)) {
                s += ts;                                           // exact, like every step of the reference's 256 additions
                continue;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (any_order_is_exact(s, ab[q], mb[q])) {
                    s += tb[q];
                    continue;
                }
                // the reference's order (every lane runs the same chain: the value stays wave-uniform)
#pragma unroll
                for (int l = 0; l < 16; ++l) {
                    s += lane_f64(d0, 16 * q + l);
                    s += lane_f64(d1, 16 * q + l);
                    s += lane_f64(d2, 16 * q + l);
                    s += lane_f64(d3, 16 * q + l);
                }
            }
        }
    }
    return s;
}

// The whole tail of a two-stage pass (each launch boundary on this chain costs a one-query search about 10 us of idle GPU,
// so small batches get it in ONE launch): a workgroup of four waves takes groups of survivors of one query,
//   1. re-scores them (below) and overwrites their keys, dots to buf_dot;
//   2. sorts the group's (key, row) pairs into lists[b][group];
//   3. takes a ticket on done[b]; the workgroup that draws the last one merges the query's lists (four waves, then a
//      tree through LDS), writes the k' records and the trailer (what select_final_kernel writes), and copies the
//      records' exact dots out of the buffer (records_dot_from_buffer_kernel).
// recs may be pinned host memory (the records are final when written); cnt_host (optional, pinned) receives cnt[b].
// WG = 0: four lanes per survivor (quad_step above), groups of 64; NJ float4s per lane and round (a quad walks 16 NJ columns a
// round), AH rounds of the row in flight, QLDS: the query from LDS instead of registers -- measured from 16 to 256 columns per
// round, one to eight rounds ahead, both query sources (10M rows x 256 queries 185-255 us, one query 32-49 us): the launch
// uses 64 columns, two rounds ahead, registers.  WG = 16 / 4: one WAVE per survivor (sums out of order where provably exact,
// above), groups of WG survivors (small groups spread a small batch's survivors over the chip; lists then hold WG entries of
// their 64), CH slabs of 256 columns per load (CH divides D / 256).
template <int WG, int CH, int NJ, int AH, bool QLDS>
__global__ __launch_bounds__(256) void finish_survivors_kernel(const float *__restrict__ E, int32_t D, const float *__restrict__ Q,
                                                               const double *__restrict__ norm_b, const int64_t *__restrict__ created,
                                                               const int64_t *__restrict__ row_ids, KwView kw,
                                                               const QueryConst *__restrict__ qcs, int64_t now_ticks,
                                                               const uint32_t *__restrict__ cnt, uint32_t *__restrict__ done, uint32_t cap,
                                                               SelEntry *__restrict__ buf, double *__restrict__ buf_dot,
                                                               SelEntry *__restrict__ lists, int32_t kprime, int64_t n_rows,
                                                               int64_t row_base, const double *__restrict__ two_stage_L,
                                                               orr_candidate *__restrict__ recs, uint32_t *__restrict__ cnt_host,
                                                               unsigned long long *__restrict__ stamps, int phase)
{
    // phase 0: all of the above in this launch.  Large batches split it at the tickets (phase 1: the groups, every list
    // written; phase 2: one workgroup per query merges and writes the records): the launch boundary orders the two halves,
    // where the one-launch form pays a device-wide fence per group and one per query (L2 write-backs across the XCDs) --
    // tens of us per workgroup by the stamps, against a few us between two launches.
#define ORR_FSTAMP(k) if (stamps && tid == 0) stamps[((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * 8 + (k)] = __builtin_amdgcn_s_memrealtime()
    __shared__ SelEntry sh[4][kSelWidth];
    __shared__ uint32_t want[kSelWidth];
    __shared__ uint32_t ticket;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x, c = lane & 3, r16 = lane >> 2;
    const uint32_t total = cnt[b];
    const uint32_t n = total < cap ? total : cap;
    constexpr uint32_t G = WG ? WG : 64;                        // survivors per group
    const uint32_t active = n ? (n + G - 1u) / G : 1u;          // a query without survivors still gets its (empty) records
    if (blockIdx.y >= (phase == 2 ? 1u : active)) return;
    ORR_FSTAMP(0);
    extern __shared__ __attribute__((aligned(16))) float q_lds[];  // QLDS / WG: the query's D floats, staged once per workgroup
    if ((QLDS || WG) && n && phase != 2) {
        for (int i = tid * 4; i < D; i += 1024)
            *reinterpret_cast<float4 *>(q_lds + i) = *reinterpret_cast<const float4 *>(Q + (int64_t)b * D + i);
        __syncthreads();
    }
    const int64_t n_lists = cap / G;
    unsigned long long k = 0ull;                                   // this wave's running best 64 (finisher)
    uint32_t p = 0xFFFFFFFFu;
    // (a workgroup walks the query's groups of 64 survivors y, y + gridDim.y, ...: the grid is a few workgroups per query, not
    // cap / 64 of which nearly all would exit at once -- 32,768 launches for 256 queries with the default buffers)
    bool finisher = false;
    for (uint32_t yb = blockIdx.y; yb < active && !finisher && phase != 2; yb += gridDim.y) {
    const uint32_t first = yb * G;
    k = 0ull;
    p = 0xFFFFFFFFu;
    if (WG && n) {
        // ---- one wave per survivor, G / 4 survivors per wave; the next survivor's row (and what its score needs) is fetched
        // while this one's is added
        constexpr int kPerWave = (int)G / 4;
        if (G < 64u && wave == 0 && lane >= (int)G) { sh[0][lane].key = 0ull; sh[0][lane].pos = 0xFFFFFFFFu; }
        const QueryConst qc0 = qcs[b];
        const int chunks = (D >> 8) / CH;
        const uint32_t wave_first = first + (uint32_t)(wave * kPerWave);
        const int live = wave_first < n ? (int)(n - wave_first < (uint32_t)kPerWave ? n - wave_first : (uint32_t)kPerWave) : 0;
        SelEntry *mine = buf + (int64_t)b * cap + wave_first;
        // lane i: the row of this wave's i-th survivor and what its score needs besides the dot (one round of dependent loads
        // for all of them, before the rows; readlane hands them out)
        const bool mine_live = lane < live;
        const uint32_t my_pos = mine_live ? mine[lane].pos : 0u;
        const double my_nb = mine_live ? norm_b[my_pos] : 0.0;
        const int64_t my_cr = mine_live ? created[my_pos] : 0;
        const uint32_t my_m = (mine_live && qc0.n_terms > 0) ? kw_matches(kw, b, my_pos) : 0u;
        ORR_FSTAMP(1);
        float4 buf_a[CH], buf_b[CH];                                  // this chunk and the next, swapping roles (no copies: a
                                                                      // copy would wait for the next chunk's loads)
        if (live > 0) load_chunk<CH>(buf_a, E + (int64_t)(uint32_t)__builtin_amdgcn_readlane((int)my_pos, 0) * (int64_t)D, 0, lane);
        double T = 0.0;                                              // (+0.0: the reference's sum starts there and can never be -0)
        float M = 0.0f;
        uint32_t mn2 = 0xFFFFFFFFu;
        double row_T = 0.0;                                          // lane i: what the wave found for its i-th survivor
        float row_M = 0.0f;
        uint32_t row_mn2 = 0xFFFFFFFFu;
        int i = 0, ch = 0;
        auto step = [&](const float4 (&cur)[CH], float4 (&nxt)[CH]) {
            const int nch = ch + 1 == chunks ? 0 : ch + 1, ni = nch ? i : i + 1;
            // the next chunk, unconditionally (behind the last one: that one again) -- a branch here would make every wait
            // below a wait for these loads as well
            const int ni_c = ni < live ? ni : live - 1;
            const int64_t row_nxt = (int64_t)(uint32_t)__builtin_amdgcn_readlane((int)my_pos, ni_c);
            load_chunk<CH>(nxt, E + row_nxt * (int64_t)D, nch, lane);
            add_chunk<CH>(cur, q_lds, ch, lane, T, M, mn2);
            if (nch == 0) {                                          // the row is complete: its totals go to lane i
                const double t = wave_total_f64(T);
                const float a = wave_total_f32(M);
                const uint32_t m = wave_min_u32(mn2);
                if (lane == i) { row_T = t; row_M = a; row_mn2 = m; }
                T = 0.0; M = 0.0f; mn2 = 0xFFFFFFFFu;
            }
            i = ni;
            ch = nch;
        };
        while (i < live) {
            step(buf_a, buf_b);
            if (i >= live) break;
            step(buf_b, buf_a);
        }
        ORR_FSTAMP(2);
        // ---- lane i finishes survivor i: rows whose sum may depend on the order are added again, the reference's way
        const uint32_t row_mn = ((row_mn2 + 1u) >> 1) - 1u;          // (bits << 1) - 1  ->  bits - 1
        const bool any_order = any_order_is_exact(0.0, (double)row_M * 1.0000152587890625, row_mn);    // (1 + 2^-16: M's fp32 roundings)
        unsigned long long redo = __ballot(mine_live && !any_order);
        while (redo) {
            const int r = __builtin_ctzll(redo);
            redo &= redo - 1ull;
            const int64_t row_r = (int64_t)(uint32_t)__builtin_amdgcn_readlane((int)my_pos, r);
            const double acc_r = exact_dot_by_slabs<CH>(E + row_r * (int64_t)D, q_lds, D, lane);
            if (lane == r) row_T = acc_r;
        }
        ORR_FSTAMP(3);
        if (mine_live) {
            QueryConst exact = qc0;
            exact.use_cos = 1;                                       // this path only runs with cosine; guards are inside fused_score
            const unsigned long long key = score_key(fused_score(row_T, my_nb, my_cr, my_m, exact, now_ticks));
            mine[lane].key = key;
            mine[lane].pad = my_m;                                   // the record's matches (below)
            buf_dot[(int64_t)b * cap + wave_first + lane] = row_T;   // ... and its exact dot
            sh[0][wave * kPerWave + lane].key = key;
            sh[0][wave * kPerWave + lane].pos = my_pos;
        }
        if (lane >= live && lane < kPerWave) { sh[0][wave * kPerWave + lane].key = 0ull; sh[0][wave * kPerWave + lane].pos = 0xFFFFFFFFu; }
    } else if (n) {
        SelEntry *mine = buf + (int64_t)b * cap + first + wave * 16;
        const bool live = first + wave * 16 + r16 < n;
        unsigned long long key = 0ull;
        uint32_t pos = 0xFFFFFFFFu;
        if (first + wave * 16 < n) {                            // (whole waves beyond the last survivor skip the dot)
            const int64_t my_row = live ? (int64_t)mine[r16].pos : 0;          // quads without a survivor read row 0 and are ignored
            constexpr int kCols = NJ * 16;                                     // columns a quad walks per round
            const float *src = E + my_row * (int64_t)D + c * (NJ * 4);
            const float *qsrc = Q + (int64_t)b * D + c * (NJ * 4);
            float4 cur[AH][NJ], qc[QLDS ? 1 : AH][NJ];                       // AH rounds of the row in flight
#pragma unroll
            for (int u = 0; u < AH; ++u)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    cur[u][j] = *reinterpret_cast<const float4 *>(src + u * kCols + j * 4);
                    if (!QLDS) qc[u][j] = *reinterpret_cast<const float4 *>(qsrc + u * kCols + j * 4);
                }
            // what the score needs besides the dot, fetched before the 3072-step chain instead of behind it
            const QueryConst qc0 = qcs[b];
            const double nb_row = norm_b[my_row];
            const int64_t cr_row = created[my_row];
            const uint32_t m = (qc0.n_terms > 0 && live && c == 0) ? kw_matches(kw, b, (uint32_t)my_row) : 0u;
            double acc = 0.0;
            for (int c0 = 0; c0 < D; c0 += kCols * AH) {
#pragma unroll
                for (int u = 0; u < AH; ++u) {
                    const int col = c0 + u * kCols;
                    // the products are not part of the chain: all four lanes of the quad round and widen theirs at once
                    double prod[NJ * 4];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const float4 qv = QLDS ? *reinterpret_cast<const float4 *>(q_lds + col + c * (NJ * 4) + j * 4) : qc[QLDS ? 0 : u][j];
                        float p0 = qv.x * cur[u][j].x;
                        float p1 = qv.y * cur[u][j].y;
                        float p2 = qv.z * cur[u][j].z;
                        float p3 = qv.w * cur[u][j].w;
                        prod[4 * j + 0] = (double)p0; prod[4 * j + 1] = (double)p1; prod[4 * j + 2] = (double)p2; prod[4 * j + 3] = (double)p3;
                    }
                    const int cn = col + kCols * AH < D ? col + kCols * AH : col;   // clamped, never branched around
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        cur[u][j] = *reinterpret_cast<const float4 *>(src + cn + j * 4);
                        if (!QLDS) qc[u][j] = *reinterpret_cast<const float4 *>(qsrc + cn + j * 4);
                    }
                    acc = quad_step<0>(acc, prod, c);
                    acc = quad_step<1>(acc, prod, c);
                    acc = quad_step<2>(acc, prod, c);
                    acc = quad_step<3>(acc, prod, c);
                }
            }
            if (live && c == 0) {
                QueryConst exact = qc0;
                exact.use_cos = 1;                               // this path only runs with cosine; guards are inside fused_score
                key = score_key(fused_score(acc, nb_row, cr_row, m, exact, now_ticks));
                pos = (uint32_t)my_row;
                mine[r16].key = key;
                mine[r16].pad = m;                               // the record's matches (below)
                buf_dot[(int64_t)b * cap + first + wave * 16 + r16] = acc;   // ... and its exact dot
            }
        }
        if (c == 0) { sh[0][wave * 16 + r16].key = key; sh[0][wave * 16 + r16].pos = pos; }
    }
    ORR_FSTAMP(4);
    if (n) {
        __syncthreads();
        if (wave == 0) {
            k = sh[0][lane].key;
            p = sh[0][lane].pos;
            wave_sort(k, p, lane);
            if (active > 1u || phase == 1) {
                SelEntry o;
                o.key = k; o.pos = p; o.pad = 0;
                lists[((int64_t)b * n_lists + yb) * kSelWidth + lane] = o;
            }
        }
    }
    if (phase == 1) {
        __syncthreads();                                           // (sh[0] is rewritten by the next group)
    } else if (active > 1u) {
        // ---- the workgroup that finishes the query's LAST group of survivors finishes the query (every thread's writes are
        // visible device-wide before the ticket); a query with at most 64 survivors is finished by its only workgroup from the
        // list it holds
        __threadfence();
        __syncthreads();
        if (tid == 0) ticket = atomicAdd(&done[b], 1u);
        __syncthreads();
        finisher = ticket == active - 1u;
        __syncthreads();                                           // (`ticket` is rewritten by the next group)
    } else {
        finisher = true;
    }
    }
    ORR_FSTAMP(5);
    if (phase == 1 || (!finisher && phase == 0)) return;
    if (active > 1u || (phase == 2 && n)) {
        if (phase == 0) __threadfence();
        k = 0ull;
        p = 0xFFFFFFFFu;
        for (uint32_t l = (uint32_t)wave; l < active; l += 4u) {
            const SelEntry e = lists[((int64_t)b * n_lists + l) * kSelWidth + lane];
            wave_merge_sorted(k, p, e.key, e.pos, lane);
        }
    }
    __syncthreads();                                               // (sh[0] was read by wave 0 above)
    sh[wave][lane].key = k;
    sh[wave][lane].pos = p;
    __syncthreads();
#pragma unroll
    for (int stride = 2; stride > 0; stride >>= 1) {
        if (wave < stride) {
            wave_merge_sorted(k, p, sh[wave + stride][lane].key, sh[wave + stride][lane].pos, lane);
            sh[wave][lane].key = k;
            sh[wave][lane].pos = p;
        }
        __syncthreads();
    }
    orr_candidate *o = recs + (int64_t)b * (kprime + 1);
    if (wave == 0) {
        if (lane < kprime)
            write_record(o + lane, k, p, b, row_base, nullptr, nullptr, 0, norm_b, created, row_ids, KwView{nullptr, 0, nullptr, nullptr}, 0);   // matches: below
        want[lane] = (lane < kprime && k != 0ull) ? p : 0xFFFFFFFFu;
        const unsigned long long valid_mask = __ballot(k != 0ull && lane < kprime);
        const int n_valid = __popcll(valid_mask);
        const unsigned long long worst_key = __shfl(k, (n_valid > 0 ? n_valid - 1 : 0), 64);
        if (lane == 0) {
            orr_candidate t;
            // every buffered row became a record -> the only rows left out are below L
            const bool kept_all = n_rows <= (int64_t)kprime || total <= (uint32_t)kprime;
            t.approx_score = (kept_all || n_valid == 0) ? -__builtin_huge_val() : key_score(worst_key);
            t.dot = 0.0; t.norm_b = two_stage_L[b]; t.created_ticks = 0;
            t.row_id = -1; t.order_key = n_rows; t.matches = n_valid; t.flags = ORR_CAND_TRAILER | ORR_CAND_TWO_STAGE;
            if (total > cap) t.flags |= ORR_CAND_OVERFLOW;
            o[kprime] = t;
            if (cnt_host) cnt_host[b] = total;
        }
    }
    __syncthreads();
    for (uint32_t i = (uint32_t)tid; i < n; i += 256u) {
        const SelEntry e = buf[(int64_t)b * cap + i];
        for (int r = 0; r < kprime; ++r)
            if (want[r] == e.pos) {                                // a row occurs once in a buffer
                o[r].dot = buf_dot[(int64_t)b * cap + i];
                o[r].matches = (int32_t)e.pad;
                o[r].flags = ORR_CAND_DOT_EXACT;
            }
    }
    ORR_FSTAMP(6);
#undef ORR_FSTAMP
}

hipError_t launch_finish_survivors(const float *E, int32_t D, const float *Q, int32_t B, const double *norm_b, const int64_t *created,
                                   const int64_t *row_ids, KwView kw, const QueryConst *qc, int64_t now_ticks, const uint32_t *cnt,
                                   uint32_t *done, uint32_t cap, SelEntry *buf, double *buf_dot, SelEntry *lists, int32_t kprime,
                                   int64_t n_rows, int64_t row_base, const double *two_stage_L, orr_candidate *recs, uint32_t *cnt_host,
                                   hipStream_t s)
{
    if (B <= 0) return hipSuccess;
    if (D % 256 != 0 || cap % 64 != 0 || kprime < 1 || kprime > kSelWidth || !two_stage_L) return hipErrorInvalidValue;
    // (a few workgroups per query, each walking the query's groups of survivors; more for small batches, whose few queries
    // would otherwise leave the chip empty)
    const int32_t group = finish_survivors_group(B, D);
    unsigned per_query = std::min<unsigned>(cap / (group ? group : 64), B >= 128 ? 4u : B >= 16 ? 16u : 64u);
#define ORR_LAUNCH_FINISH(WG, CH, LDS_BYTES)                                                                                          \
    hipLaunchKernelGGL((finish_survivors_kernel<WG, CH, 4, 2, false>), dim3((unsigned)B, per_query), dim3(256), LDS_BYTES, s, E, D, Q, \
                       norm_b, created, row_ids, kw, qc, now_ticks, cnt, done, cap, buf, buf_dot, lists, kprime, n_rows, row_base,      \
                       two_stage_L, recs, cnt_host, stamps_arg, phase)
#define ORR_LAUNCH_FINISH_WAVES(WG)                                                                                                    \
    do {                                                                                                                                \
        if (slabs % 12 == 0) ORR_LAUNCH_FINISH(WG, 12, wave_lds);                                                                       \
        else if (slabs % 4 == 0) ORR_LAUNCH_FINISH(WG, 4, wave_lds);                                                                    \
        else if (slabs % 3 == 0) ORR_LAUNCH_FINISH(WG, 3, wave_lds);                                                                    \
        else ORR_LAUNCH_FINISH(WG, 1, wave_lds);                                                                                        \
    } while (0)
    // ORR_FINISH_STAMPS=file (diagnostic): every launch appends its workgroups' phase stamps (100 MHz) to the file
    static const char *stamps_path = getenv("ORR_FINISH_STAMPS");
    unsigned long long *d_stamps = nullptr;
    const size_t stamp_words = (size_t)B * (per_query + 1) * 8;      // (the second launch of a split one stamps behind the first)
    if (stamps_path) {
        if (hipMalloc(reinterpret_cast<void **>(&d_stamps), stamp_words * 8) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemsetAsync(d_stamps, 0, stamp_words * 8, s);
    }
    const size_t wave_lds = (size_t)D * 4;                                     // the query
    const int slabs = D / 256;
    // one launch (tickets) for small batches, whose call pays every launch boundary; two for large ones (ORR_FINISH_SPLIT=0|1
    // forces either)
    static const int forced_split = [] { const char *e = getenv("ORR_FINISH_SPLIT"); return e ? atoi(e) : -1; }();
    const bool split = forced_split >= 0 ? forced_split != 0 : B >= 64;
    for (int phase = split ? 1 : 0; phase <= (split ? 2 : 0); ++phase) {
        unsigned long long *stamps_arg = d_stamps ? d_stamps + (phase == 2 ? (size_t)B * per_query * 8 : 0) : nullptr;
        if (phase == 2) per_query = 1;
        if (group == 16) ORR_LAUNCH_FINISH_WAVES(16);
        else if (group == 4) ORR_LAUNCH_FINISH_WAVES(4);
        else ORR_LAUNCH_FINISH(0, 1, 0);
    }
#undef ORR_LAUNCH_FINISH_WAVES
    if (d_stamps) {
        std::vector<unsigned long long> h(stamp_words);
        if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(h.data(), d_stamps, stamp_words * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE *f = fopen(stamps_path, "a")) {
                fprintf(f, "launch B=%d per_query=%u group=%d\n", B, per_query, group);
                for (size_t w = 0; w < stamp_words; w += 8)
                    if (h[w]) fprintf(f, "%zu %llu %llu %llu %llu %llu %llu %llu\n", w / 8, h[w], h[w + 1], h[w + 2], h[w + 3], h[w + 4], h[w + 5], h[w + 6]);
                fclose(f);
            }
        }
        (void)hipFree(d_stamps);
    }
#undef ORR_LAUNCH_FINISH
    return hipGetLastError();
}

// Survivors per group of finish_survivors for a batch of B (its lists: cap / group per query, kSelWidth entries each); 0: the quad
// chain (groups of 64).  One wave per survivor for the smallest batches only (1M x 3072 rows, one MI355X, us per launch):
//     queries   1    8    32   64   256 (10M rows)
//     chain     31   35   40   41   103      (two launches from 64 queries on; 220 in one)
//     waves     23   32   47   87   162-234  (groups of 4 / 4 / 16 / 16 / 16-64)
// -- with many survivors per wave the rows that are added twice (6 % of them, 10-30 us each) are a tail the chain does not have.
// ORR_FINISH_CHAIN=1 forces the chain, ORR_FINISH_GROUP=16|4 the group (experiments).
int32_t finish_survivors_group(int32_t B, int32_t D)
{
    static const bool chain = [] { const char *e = getenv("ORR_FINISH_CHAIN"); return e && atoi(e) != 0; }();
    static const int forced = [] { const char *e = getenv("ORR_FINISH_GROUP"); return e ? atoi(e) : 0; }();
    if (chain || D > 8192) return 0;
    if (forced == 16 || forced == 4) return forced;
    return B < 8 ? 4 : 0;
}

hipError_t launch_rescore_buffer_exact(const float *E, int32_t D, const float *Q, int32_t B, const double *norm_b,
                                       const int64_t *created, KwView kw, const QueryConst *qc, int64_t now_ticks,
                                       const uint32_t *cnt, uint32_t cap, SelEntry *buf, double *buf_dot, hipStream_t s)
{
    if (B <= 0) return hipSuccess;
    if (D % 64 != 0) return hipErrorInvalidValue;
    // one row per lane (large batches; up to 256 queries go through launch_finish_survivors, four lanes per survivor)
    hipLaunchKernelGGL(rescore_buffer_exact_kernel<false>, dim3((unsigned)B, cap / 64), dim3(64), 0, s, E, D, Q, norm_b, created, kw, qc,
                       now_ticks, cnt, cap, buf, buf_dot);
    return hipGetLastError();
}

// Candidate buffers of the fused pass -> sorted 64-entry lists appended behind the prefix's
// lists (out_sel[b][seg_first + l]); empty lists where a query has fewer entries.
__global__ __launch_bounds__(64) void buffer_to_lists_kernel(const SelEntry *__restrict__ buf, const uint32_t *__restrict__ cnt,
                                                             uint32_t cap, int32_t seg_first, int32_t n_seg_total,
                                                             SelEntry *__restrict__ out_sel)
{
    const int lane = threadIdx.x, b = blockIdx.x, l = blockIdx.y;
    const uint32_t n = cnt[b] < cap ? cnt[b] : cap;
    const uint32_t idx = (uint32_t)l * 64u + lane;
    unsigned long long k = 0ull;
    uint32_t p = 0xFFFFFFFFu;
    if (idx < n) { const SelEntry e = buf[(int64_t)b * cap + idx]; k = e.key; p = e.pos; }
    if ((uint32_t)l * 64u < n) wave_sort(k, p, lane);
    SelEntry o;
    o.key = k; o.pos = p; o.pad = 0;
    out_sel[((int64_t)b * n_seg_total + seg_first + l) * kSelWidth + lane] = o;
}

hipError_t launch_buffer_to_lists(const SelEntry *buf, const uint32_t *cnt, uint32_t cap, int32_t B, int32_t seg_first,
                                  int32_t n_seg_total, SelEntry *out_sel, hipStream_t s)
{
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(buffer_to_lists_kernel, dim3((unsigned)B, cap / 64), dim3(64), 0, s, buf, cnt, cap, seg_first, n_seg_total,
                       out_sel);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------
// K2s  up to 32 queries per launch, HBM-bound: the streaming structure of
// dot_exact_tiled (each wave owns 64 rows, 64x64-float pieces staged through a
// swizzled wave-private LDS tile, next piece prefetched with nontemporal loads,
// next piece prefetched with nontemporal loads) with the arithmetic on the matrix
// cores: per 4 floats of a row, two 32x32x2 MFMAs per 32-row half against the 32
// queries.  Lane l takes dwords k = 4j,4j+2 (l < 32) or 4j+1,4j+3 (l >= 32) of each
// 16-byte row piece (ds_read2_b32), which is exactly the A/B operand layout.  The matching 32 x 64 query piece is shared by the workgroup's
// four waves through a double-buffered LDS tile (one barrier per piece).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void gemv_mfma_kernel(const float *__restrict__ Q, int32_t B,
                                                        const float *__restrict__ E, int64_t n_rows, int32_t D,
                                                        float *__restrict__ S, int64_t s_stride)
{
    // 4 x 16 KiB wave-private row tiles + 2 x 8 KiB shared query tiles = 80 KiB: two workgroups per CU
    __shared__ __attribute__((aligned(16))) float tile_all[4][64 * 64];
    __shared__ __attribute__((aligned(16))) float qtile[2][32 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *tile = tile_all[wave];
    const int fr = lane & 31, fh = lane >> 5;
    const int ld_row = lane >> 4, ld_ch = lane & 15;
    const int64_t n_blocks = (n_rows + 255) >> 8;           // 256 rows per workgroup step

    for (int64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const int64_t row0 = (blk << 8) + wave * 64;
        f32x16 acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
        float4 stage[16], qreg[2];
        auto load_stage = [&](int c0) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int r = it * 4 + ld_row;
                const int64_t row = row0 + r;
                stage[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < n_rows) {
                    const f32x4v v = __builtin_nontemporal_load(reinterpret_cast<const f32x4v *>(E + row * (int64_t)D + c0 + ld_ch * 4));
                    stage[it] = make_float4(v.x, v.y, v.z, v.w);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {                   // the workgroup's 32 x 64 query piece: 2 float4 per thread
                const int id = tid + 256 * u, qi = id >> 4, qj = id & 15;
                qreg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (qi < B) qreg[u] = *reinterpret_cast<const float4 *>(Q + (int64_t)qi * D + c0 + qj * 4);
            }
        };
        load_stage(0);
        int buf = 0;
        for (int c0 = 0; c0 < D; c0 += 64, buf ^= 1) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int r = it * 4 + ld_row;
                *reinterpret_cast<float4 *>(tile + r * 64 + ((ld_ch ^ (r & 15)) << 2)) = stage[it];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int id = tid + 256 * u, qi = id >> 4, qj = id & 15;
                *reinterpret_cast<float4 *>(qtile[buf] + qi * 64 + ((qj ^ (qi & 15)) << 2)) = qreg[u];
            }
            if (c0 + 64 < D) load_stage(c0 + 64);           // next pieces in flight during the MFMAs
            __builtin_amdgcn_sched_barrier(0);              // keep the loads ABOVE the MFMA block
            __syncthreads();                                // one barrier per piece (query tile is double-buffered)
            const float *qt = qtile[buf];
#pragma unroll 4
            for (int j = 0; j < 16; ++j) {
                const int sw = ((j ^ (fr & 15)) << 2) + fh;  // dwords fh and fh+2 of the 16-byte piece: no select
                const float a_lo = qt[fr * 64 + sw], a_hi = qt[fr * 64 + sw + 2];
                const float b0_lo = tile[fr * 64 + sw], b0_hi = tile[fr * 64 + sw + 2];
                const float b1_lo = tile[(32 + fr) * 64 + sw], b1_hi = tile[(32 + fr) * 64 + sw + 2];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_lo, b0_lo, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_lo, b1_lo, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_hi, b0_hi, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_hi, b1_hi, acc1, 0, 0, 0);
            }
        }
        __syncthreads();                                    // before the next row block reuses qtile[0]
        // C/D layout: column (E row) = lane & 31, query = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int qi = (e & 3) + 8 * (e >> 2) + 4 * fh;
            if (qi < B) {
                if (row0 + fr < n_rows) S[(int64_t)qi * s_stride + row0 + fr] = acc0[e];
                if (row0 + 32 + fr < n_rows) S[(int64_t)qi * s_stride + row0 + 32 + fr] = acc1[e];
            }
        }
    }
}

// K2s for B <= 16: same structure on v_mfma_f32_16x16x4_f32 (A[i = lane & 15][k = lane >> 4]).
// A 16-byte piece of a row holds exactly one MFMA's four k values: lane (i, kq) reads the
// dword kq of it (plain ds_read_b32, no select).  64 rows = 4 column tiles of 16, 16 accumulator VGPRs, half the matrix work of
// the 32-query form, so a batch of up to 16 queries stays HBM-bound.
__global__ __launch_bounds__(256, 2) void gemv_mfma16_kernel(const float *__restrict__ Q, int32_t B,
                                                             const float *__restrict__ E, int64_t n_rows, int32_t D,
                                                             float *__restrict__ S, int64_t s_stride)
{
    __shared__ __attribute__((aligned(16))) float tile_all[4][64 * 64];
    __shared__ __attribute__((aligned(16))) float qtile[2][16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *tile = tile_all[wave];
    const int fr = lane & 15, kq = lane >> 4;
    const int ld_row = lane >> 4, ld_ch = lane & 15;
    const int64_t n_blocks = (n_rows + 255) >> 8;
    const int qi_ld = tid >> 4, qj_ld = tid & 15;           // the 16 x 64 query piece: one float4 per thread

    for (int64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const int64_t row0 = (blk << 8) + wave * 64;
        f32x4v acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        float4 stage[16], qreg;
        auto load_stage = [&](int c0) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int r = it * 4 + ld_row;
                const int64_t row = row0 + r;
                stage[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < n_rows) {
                    const f32x4v v = __builtin_nontemporal_load(reinterpret_cast<const f32x4v *>(E + row * (int64_t)D + c0 + ld_ch * 4));
                    stage[it] = make_float4(v.x, v.y, v.z, v.w);
                }
            }
            qreg = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qi_ld < B) qreg = *reinterpret_cast<const float4 *>(Q + (int64_t)qi_ld * D + c0 + qj_ld * 4);
        };
        load_stage(0);
        int buf = 0;
        for (int c0 = 0; c0 < D; c0 += 64, buf ^= 1) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int r = it * 4 + ld_row;
                *reinterpret_cast<float4 *>(tile + r * 64 + ((ld_ch ^ (r & 15)) << 2)) = stage[it];
            }
            *reinterpret_cast<float4 *>(qtile[buf] + qi_ld * 64 + ((qj_ld ^ (qi_ld & 15)) << 2)) = qreg;
            if (c0 + 64 < D) load_stage(c0 + 64);
            __builtin_amdgcn_sched_barrier(0);              // keep the loads ABOVE the MFMA block
            __syncthreads();
            const float *qt = qtile[buf];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int sw = ((j ^ fr) << 2) + kq;        // rows 16t + fr: (row & 15) == fr
                const float a = qt[fr * 64 + sw];
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, tile[(16 * t + fr) * 64 + sw], acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
        // C/D layout of the 16x16 MFMA: column (E row) = lane & 15, query = (lane >> 4) * 4 + reg
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int qi = kq * 4 + e;
                const int64_t row = row0 + 16 * t + fr;
                if (qi < B && row < n_rows) S[(int64_t)qi * s_stride + row] = acc[t][e];
            }
    }
}

hipError_t launch_gemv_mfma(const float *Q, int32_t B, const float *E, int64_t n_rows, int32_t D, float *S,
                            int64_t s_stride, hipStream_t s)
{
    if (B <= 0 || n_rows <= 0) return hipSuccess;
    if (B > 32 || D % 64 != 0) return hipErrorInvalidValue;
    int64_t blocks = (n_rows + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (B <= 16) {
        hipLaunchKernelGGL(gemv_mfma16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, Q, B, E, n_rows, D, S, s_stride);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(gemv_mfma_kernel, dim3((unsigned)blocks), dim3(256), 0, s, Q, B, E, n_rows, D, S, s_stride);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K6  exact re-score: one wave per query, lane c = the query's c-th candidate record.
// Same structure as dot_exact_tiled (rows staged through a swizzled wave-private LDS
// tile, each lane then walks its own row in index order), with the 64 rows GATHERED by
// the records' candidate positions.  Writes the reference-order fp64 dot into the record
// and marks it ORR_CAND_DOT_EXACT.
// ---------------------------------------------------------------------------
template <bool VIA_LDS>
__global__ __launch_bounds__(64) void rescore_exact_kernel(const float *__restrict__ E, int32_t D,
                                                           const float *__restrict__ Q, int32_t B, int32_t kprime,
                                                           int64_t row_base, orr_candidate *__restrict__ recs)
{
    __shared__ __attribute__((aligned(16))) float tile[64 * 64];
    __shared__ int64_t rows[64];
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    orr_candidate *mine = recs + (int64_t)b * (kprime + 1);
    int64_t my_row = -1;
    if (lane < kprime && mine[lane].row_id >= 0 && !(mine[lane].flags & ORR_CAND_TRAILER))
        my_row = mine[lane].order_key - row_base;                     // position in this shard
    rows[lane] = my_row;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double acc = exact_dot_of_gathered_rows<VIA_LDS>(E, D, Q + (int64_t)b * D, rows, tile, lane);
    if (my_row >= 0) {
        mine[lane].dot = acc;
        mine[lane].flags |= ORR_CAND_DOT_EXACT;
    }
}

// any D: one thread per candidate, plain loads
__global__ __launch_bounds__(256) void rescore_exact_generic(const float *__restrict__ E, int32_t D,
                                                             const float *__restrict__ Q, int32_t B, int32_t kprime,
                                                             int64_t row_base, orr_candidate *__restrict__ recs)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)B * kprime) return;
    const int b = (int)(t / kprime), c = (int)(t % kprime);
    orr_candidate *r = recs + (int64_t)b * (kprime + 1) + c;
    if (r->row_id < 0) return;
    const float *e = E + (r->order_key - row_base) * (int64_t)D;
    const float *q = Q + (int64_t)b * D;
    double acc = 0.0;
    for (int i = 0; i < D; ++i) {
        float p = q[i] * e[i];
        acc += (double)p;
    }
    r->dot = acc;
    r->flags |= ORR_CAND_DOT_EXACT;
}

hipError_t launch_rescore_exact(const float *E, int32_t D, const float *Q, int32_t B, int32_t kprime, int64_t row_base,
                                orr_candidate *recs, hipStream_t s)
{
    if (B <= 0 || D <= 0) return hipSuccess;
    if (kprime <= 64 && D % 64 == 0 && (reinterpret_cast<uintptr_t>(E) & 15) == 0) {
        hipLaunchKernelGGL(rescore_exact_kernel<false>, dim3((unsigned)B), dim3(64), 0, s, E, D, Q, B, kprime, row_base, recs);
    } else {
        const int64_t threads = (int64_t)B * kprime;
        hipLaunchKernelGGL(rescore_exact_generic, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, E, D, Q, B, kprime,
                           row_base, recs);
    }
    return hipGetLastError();
}

}  // namespace orr
