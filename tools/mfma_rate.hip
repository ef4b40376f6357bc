// mfma_rate.hip -- what one wave per SIMD can issue: v_mfma_i32_32x32x32_i8 back to back (16 independent accumulator tiles,
// the screening kernel's shape), alone or with the kernel's fragment reads (8 ds_read_b128 per 16 MFMAs into the other
// register set), and v_mfma_i32_16x16x64_i8 for comparison, on every CU at once.  Prints cycles per MFMA
// (s_memtime) and the clock (s_memtime / s_memrealtime).  Diagnostic for DESIGN.md's K-loop bound; not part of the library.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void rate_kernel(int iters, unsigned long long *out, int *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<int *>(lds)[i] = i * 2654435761u;
    __syncthreads();
    i32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
    i32x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = *reinterpret_cast<const i32x4 *>(lds + lane * 16 + i * 2048); b[i] = *reinterpret_cast<const i32x4 *>(lds + 8192 + lane * 16 + i * 2048); }
    const unsigned char *p = lds + lane * 16;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    i32x4 a2[4], b2[4];                              // MODE 1: the other fragment set (read while this one is multiplied)
#pragma unroll
    for (int i = 0; i < 4; ++i) { a2[i] = a[i]; b2[i] = b[i]; }
    for (int it = 0; it < iters; it += 2) {
        // 16 MFMAs on (a, b) with the 8 reads of (a2, b2) behind the first 8, then the same the other way round: what the
        // screening kernel's half K-tiles do (no MFMA waits for a read issued less than 8 MFMAs earlier)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (MODE == 1 && i < 2) {
                    if (i == 0) a2[j] = *reinterpret_cast<const i32x4 *>(p + ((it + j) & 7) * 2048);
                    else b2[j] = *reinterpret_cast<const i32x4 *>(p + 16384 + ((it + j) & 7) * 2048);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2[i], b2[j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (MODE == 1 && i < 2) {
                    if (i == 0) a[j] = *reinterpret_cast<const i32x4 *>(p + ((it + j + 1) & 7) * 2048);
                    else b[j] = *reinterpret_cast<const i32x4 *>(p + 16384 + ((it + j + 1) & 7) * 2048);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    if (s == 0x12345678) *sink = s;
    if (lane == 0) {
        const int w = blockIdx.x * WAVES + (threadIdx.x >> 6);
        out[2 * w] = t1 - t0;
        out[2 * w + 1] = r1 - r0;
    }
}

// The same with v_mfma_i32_16x16x64_i8: 64 independent accumulator tiles of 4 registers (a 128 x 128 wave tile), half the
// operations per instruction.
__global__ __launch_bounds__(256, 1) void rate16_kernel(int iters, unsigned long long *out, int *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<int *>(lds)[i] = i * 2654435761u;
    __syncthreads();
    i32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0;
    i32x4 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = *reinterpret_cast<const i32x4 *>(lds + lane * 16 + i * 1024); b[i] = *reinterpret_cast<const i32x4 *>(lds + 8192 + lane * 16 + i * 1024); }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], b[j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    if (s == 0x12345678) *sink = s;
    if (lane == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        out[2 * w] = t1 - t0;
        out[2 * w + 1] = r1 - r0;
    }
}

// Block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 (gfx950): FMT 0 = fp8 e4m3, 2 = fp6 e2m3, 4 = fp4 e2m1 for both operands; the
// same 8 x 8 accumulator tiles (f32 here), 128 k per instruction = 65,536 operations each, twice the int8 form's.  Operands are
// eight registers per fragment whatever the format (fp6 uses six of them, fp4 four).  Unit scales (E8M0 127 in every byte).
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int FMT>
__global__ __launch_bounds__(256, 1) void rate_f8f6f4_kernel(int iters, unsigned long long *out, float *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    // random bit patterns with the exponent fields kept small enough that nothing overflows: fp8 bytes masked to |x| < 2,
    // fp6 / fp4 patterns are all finite anyway
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) {
        unsigned v = i * 2654435761u;
        if (FMT == 0) v &= 0xBFBFBFBFu;                        // e4m3: clear the top exponent bit of every byte
        reinterpret_cast<unsigned *>(lds)[i] = v;
    }
    __syncthreads();
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    i32x8 a[8], b[4];                                          // (8 + 4 fragments of 8 registers: 96; the b fragments are reused for j and j + 4)
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const i32x8 *>(lds + lane * 32 + i * 2048);
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const i32x8 *>(lds + 16384 + lane * 32 + j * 2048);
    __syncthreads();
    const int unit = 0x7F7F7F7F;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j & 3], acc[i][j], FMT, FMT, 0, unit, 0, unit);
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    if (s == 0.123456f) *sink = s;
    if (lane == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        out[2 * w] = t1 - t0;
        out[2 * w + 1] = r1 - r0;
    }
}

template <int FMT>
static void run_f8f6f4(const char *what, int blocks, int iters)
{
    unsigned long long *d = nullptr;
    float *sink = nullptr;
    hipMalloc(reinterpret_cast<void **>(&d), sizeof(unsigned long long) * 2 * blocks * 4);
    hipMalloc(reinterpret_cast<void **>(&sink), 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(rate_f8f6f4_kernel<FMT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((rate_f8f6f4_kernel<FMT>), dim3(blocks), dim3(256), 160 * 1024, 0, iters, d, sink);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(2 * blocks * 4);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (int w = 0; w < blocks * 4; ++w) { cyc.push_back((double)h[2 * w] / (64.0 * iters)); clk.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 100e6); }
    std::sort(cyc.begin(), cyc.end());
    std::sort(clk.begin(), clk.end());
    const double c = cyc[cyc.size() / 2], f = clk[clk.size() / 2];
    printf("%-44s 1 wave(s)/SIMD x %d workgroups: %.1f cycles per MFMA per wave (median), clock %.2f GHz  [65536 ops each: %.2f POP/s on %d SIMDs]\n",
           what, blocks, c, f / 1e9, 65536.0 * blocks * 4 / (c / f) / 1e15, blocks * 4);
    hipFree(d);
    hipFree(sink);
}

static void run16(int blocks, int iters)
{
    unsigned long long *d = nullptr;
    int *sink = nullptr;
    hipMalloc(reinterpret_cast<void **>(&d), sizeof(unsigned long long) * 2 * blocks * 4);
    hipMalloc(reinterpret_cast<void **>(&sink), 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(rate16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(rate16_kernel, dim3(blocks), dim3(256), 160 * 1024, 0, iters, d, sink);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(2 * blocks * 4);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (int w = 0; w < blocks * 4; ++w) { cyc.push_back((double)h[2 * w] / (64.0 * iters)); clk.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 100e6); }
    std::sort(cyc.begin(), cyc.end());
    std::sort(clk.begin(), clk.end());
    printf("%-44s 1 wave(s)/SIMD x %d workgroups: %.1f cycles per MFMA per wave (median), clock %.2f GHz  [32768 ops each]\n",
           "v_mfma_i32_16x16x64_i8 only", blocks, cyc[cyc.size() / 2], clk[clk.size() / 2] / 1e9);
    hipFree(d);
    hipFree(sink);
}

template <int MODE, int WAVES>
static void run(const char *what, int blocks, int iters)
{
    unsigned long long *d = nullptr;
    int *sink = nullptr;
    hipMalloc(reinterpret_cast<void **>(&d), sizeof(unsigned long long) * 2 * blocks * WAVES);
    hipMalloc(reinterpret_cast<void **>(&sink), 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(rate_kernel<MODE, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((rate_kernel<MODE, WAVES>), dim3(blocks), dim3(WAVES * 64), 160 * 1024, 0, iters, d, sink);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(2 * blocks * WAVES);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (int w = 0; w < blocks * WAVES; ++w) { cyc.push_back((double)h[2 * w] / (16.0 * iters)); clk.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 100e6); }
    std::sort(cyc.begin(), cyc.end());
    std::sort(clk.begin(), clk.end());
    printf("%-44s %d wave(s)/SIMD x %d workgroups: %.1f cycles per MFMA per wave (median), clock %.2f GHz\n", what, WAVES / 4, blocks,
           cyc[cyc.size() / 2], clk[clk.size() / 2] / 1e9);
    hipFree(d);
    hipFree(sink);
}

int main()
{
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int iters = 20000;          // 320k MFMAs per wave: ~5 ms
    run<0, 4>("MFMA only", 1, iters);
    run<0, 4>("MFMA only", cus, iters);
    run<1, 4>("MFMA + 8 ds_read_b128 per 16 (double-buffered)", cus, iters);
    run16(1, iters / 2);
    run16(cus, iters / 2);
    // round 3: the block-scaled forms (is the 2x of FP4 / FP6 over int8 still there with every matrix core busy?)
    run_f8f6f4<0>("v_mfma_scale_f32_16x16x128_f8f6f4 fp8 e4m3", 1, iters / 2);
    run_f8f6f4<0>("v_mfma_scale_f32_16x16x128_f8f6f4 fp8 e4m3", cus, iters / 2);
    run_f8f6f4<2>("v_mfma_scale_f32_16x16x128_f8f6f4 fp6 e2m3", 1, iters / 2);
    run_f8f6f4<2>("v_mfma_scale_f32_16x16x128_f8f6f4 fp6 e2m3", cus, iters / 2);
    run_f8f6f4<4>("v_mfma_scale_f32_16x16x128_f8f6f4 fp4 e2m1", 1, iters / 2);
    run_f8f6f4<4>("v_mfma_scale_f32_16x16x128_f8f6f4 fp4 e2m1", cus, iters / 2);
    return 0;
}
