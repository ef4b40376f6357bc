// What do the cycles per MFMA of screen_tile16_kernel's K-tile depend on?  Runs the generated K-tile (tools/gen_tile16_asm.py,
// the no-request form) and ablated variants of it (tools/gen_ktile16_variants.py) in a loop, four waves per workgroup as in the
// kernel, on `blocks` CUs.  Build: python tools/gen_ktile16_variants.py && hipcc --offload-arch=gfx950 -O3 tools/ktile16_rate.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#include "ktile16_variants.inc"
#include "../omni-recall-rag_amd/csrc/orr_screen_tile16_asm.inc"

#define FRAG_CLOBBERS "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", \
    "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", \
    "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", \
    "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"

#define KERNEL(NAME, TEXT) \
__global__ __launch_bounds__(256, 1) void NAME(int iters, unsigned long long *out) \
{ \
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[]; \
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6; \
    for (int i = threadIdx.x; i < 36864; i += blockDim.x) reinterpret_cast<int *>(lds)[i] = i * 2654435761u; \
    __syncthreads(); \
    const int frow = ((((0 - ((lane & 15) >> 2)) & 3) << 2) | (lane & 3)); \
    const unsigned fo = frow * 64 + (((lane >> 4) ^ ((frow >> 2) & 3)) << 4); \
    const unsigned pab = (unsigned)(uintptr_t)lds + (wave >> 1) * 8192 + fo, pbb = (unsigned)(uintptr_t)lds + 49152 + (wave & 1) * 8192 + fo; \
    const unsigned pae = pab + 3 * 16384, pbe = pbb + 6 * 16384; \
    unsigned pa = pab, pan = pab + 16384, pbn = pbb + 16384, pat = 0, pbt = 0; \
    asm volatile(ORR_T16_ZERO ::: ORR_T16_ACC_CLOBBERS); \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
    for (int it = 0; it < iters; ++it) \
        asm volatile(TEXT : [pa] "+v"(pa), [pan] "+v"(pan), [pbn] "+v"(pbn), [pat] "+v"(pat), [pbt] "+v"(pbt) \
                     : [pab] "v"(pab), [pbb] "v"(pbb), [pae] "v"(pae), [pbe] "v"(pbe), [wn] "n"(0) : ORR_T16_ACC_CLOBBERS, FRAG_CLOBBERS, "vcc", "scc", "memory"); \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory"); \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
    if (lane == 0) { const int w = blockIdx.x * 4 + wave; out[2 * w] = t1 - t0; out[2 * w + 1] = r1 - r0; } \
}

KERNEL(k_full, KT16_FULL)
KERNEL(k_nobarrier, KT16_NOBARRIER)
KERNEL(k_nowaits, KT16_NOWAITS)
KERNEL(k_mfma_only, KT16_MFMA_ONLY)
KERNEL(k_mfma_only_a2, KT16_MFMA_ONLY_A2)
KERNEL(k_mfma_rowmajor, KT16_MFMA_ROWMAJOR)
KERNEL(k_reads_a2, KT16_READS_A2)
#ifdef WITH_A1
KERNEL(k_mfma_only_a1, KT16_MFMA_ONLY_A1)
#endif

template <typename K>
static void run(const char *what, K kernel, int blocks, int iters)
{
    unsigned long long *d = nullptr;
    hipMalloc(reinterpret_cast<void **>(&d), sizeof(unsigned long long) * 2 * blocks * 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 160 * 1024, 0, iters, d);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(2 * blocks * 4);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (int w = 0; w < blocks * 4; ++w) { cyc.push_back((double)h[2 * w] / iters); clk.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 100e6); }
    std::sort(cyc.begin(), cyc.end());
    std::sort(clk.begin(), clk.end());
    printf("%-34s %3d workgroups: %7.1f cycles per K-tile of 64 MFMAs (%.2f per MFMA), clock %.2f GHz\n", what, blocks, cyc[cyc.size() / 2],
           cyc[cyc.size() / 2] / 64.0, clk[clk.size() / 2] / 1e9);
    hipFree(d);
}

int main()
{
    for (int blocks : {8, 256}) {
        const int iters = blocks == 8 ? 20000 : 200000;
        run("K-tile as in the kernel (no requests)", k_full, blocks, iters);
        run("  without the barrier", k_nobarrier, blocks, iters);
        run("  without barrier and waits", k_nowaits, blocks, iters);
        run("  MFMAs only, same order/registers", k_mfma_only, blocks, iters);
        run("  MFMAs only, A fragments at v[190..]", k_mfma_only_a2, blocks, iters);
#ifdef WITH_A1
        run("  MFMAs only, A fragments at v[191..]", k_mfma_only_a1, blocks, iters);
#endif
        run("  MFMAs only, accumulators in order", k_mfma_rowmajor, blocks, iters);
        run("  full, A fragments at v[190..]", k_reads_a2, blocks, iters);
    }
    return 0;
}
