// gather_rate.hip -- what the memory system gives a kernel that reads whole rows of a large table at random: the survivors'
// re-score (finish_survivors, rescore_buffer_exact) reads ~100 rows of 12 KB per query from anywhere in a 123 GB table.
// One wave reads ROWS rows of `row_bytes` each (coalesced 1 KB loads, `depth` rows in flight), rows drawn at random from the
// first `span` GB of the table; prints GB/s for a few (rows, span, waves per CU).  Diagnostic for DESIGN.md; not part of
// the library.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_rate.hip -o /tmp/gather_rate && /tmp/gather_rate [table GB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int KB>
__global__ __launch_bounds__(256) void gather_kernel(const float4 *__restrict__ table, const uint32_t *__restrict__ rows, int rows_per_wave,
                                                     float *sink)
{
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t *mine = rows + (size_t)wave * rows_per_wave;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    float4 a[KB], b[KB];
    const float4 *r = table + (size_t)mine[0] * (KB * 64) + lane;
#pragma unroll
    for (int u = 0; u < KB; ++u) a[u] = r[u * 64];
    for (int i = 0; i < rows_per_wave; i += 2) {
        const float4 *r1 = table + (size_t)mine[i + 1 < rows_per_wave ? i + 1 : i] * (KB * 64) + lane;
#pragma unroll
        for (int u = 0; u < KB; ++u) b[u] = r1[u * 64];
#pragma unroll
        for (int u = 0; u < KB; ++u) { acc.x += a[u].x; acc.y += a[u].y; acc.z += a[u].z; acc.w += a[u].w; }
        const float4 *r2 = table + (size_t)mine[i + 2 < rows_per_wave ? i + 2 : i] * (KB * 64) + lane;
#pragma unroll
        for (int u = 0; u < KB; ++u) a[u] = r2[u * 64];
#pragma unroll
        for (int u = 0; u < KB; ++u) { acc.x += b[u].x; acc.y += b[u].y; acc.z += b[u].z; acc.w += b[u].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

int main(int argc, char **argv)
{
    const double table_gb = argc > 1 ? atof(argv[1]) : 120.0;
    constexpr int KB = 12;                                            // a row of 3072 floats
    const size_t row_bytes = (size_t)KB * 1024, n_rows = (size_t)(table_gb * 1e9 / row_bytes);
    float4 *table = nullptr;
    CHECK(hipMalloc(&table, n_rows * row_bytes));
    CHECK(hipMemset(table, 0, n_rows * row_bytes));
    float *sink = nullptr;
    CHECK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("table %.1f GB, %zu rows of %zu bytes\n", n_rows * row_bytes / 1e9, n_rows, row_bytes);
    printf("%10s %10s %8s %12s %10s %10s\n", "rows", "span GB", "waves", "rows/wave", "us", "GB/s");
    const double spans[] = {table_gb, 8.0, 0.2};
    const int totals[] = {24576, 98304, 393216};
    const int per_wave[] = {1, 4, 16};
    for (double span : spans)
        for (int total : totals)
            for (int rpw : per_wave) {
                const int waves = total / rpw, blocks = waves / 4;
                const size_t span_rows = (size_t)(span * 1e9 / row_bytes) < n_rows ? (size_t)(span * 1e9 / row_bytes) : n_rows;
                std::vector<uint32_t> rows(total);
                uint64_t x = 88172645463325252ull + total + rpw;
                for (int i = 0; i < total; ++i) {
                    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                    rows[i] = (uint32_t)(x % span_rows);
                }
                uint32_t *d_rows = nullptr;
                CHECK(hipMalloc(&d_rows, total * 4));
                CHECK(hipMemcpy(d_rows, rows.data(), total * 4, hipMemcpyHostToDevice));
                float best = 1e30f;
                for (int rep = 0; rep < 4; ++rep) {
                    // (a different draw of rows per repetition would be fairer to the caches; the spans here are far beyond them
                    // except the 0.2 GB one, which is the cache-resident reference)
                    CHECK(hipEventRecord(e0));
                    hipLaunchKernelGGL(gather_kernel<KB>, dim3(blocks), dim3(256), 0, 0, table, d_rows, rpw, sink);
                    CHECK(hipEventRecord(e1));
                    CHECK(hipEventSynchronize(e1));
                    float ms = 0.f;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep > 0 && ms < best) best = ms;
                }
                printf("%10d %10.1f %8d %12d %10.1f %10.0f\n", total, span, waves, rpw, best * 1e3, total * (double)row_bytes / (best * 1e-3) / 1e9);
                CHECK(hipFree(d_rows));
            }
    return 0;
}
