// valu_issue.hip -- diagnostics, not part of the product: how many integer VALU / SALU / LDS instructions a gfx950
// SIMD issues per cycle at 1, 2, 4, 8 waves per SIMD (the classification kernels are issue bound, so this is the
// number their instruction budgets are priced with).   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITERS = 2048;

// kind 0: 8 independent v_add_u32 chains; 1: v_cmp + v_cndmask pairs; 2: mixed VALU + SALU (s_add between);
// 3: ds_read_b32 dependent on a VALU result (latency chain); 4: v_and_or / v_lshl_add 3-operand ops
template <int KIND>
__global__ __launch_bounds__(256) void k_issue(uint32_t *out, unsigned long long *cyc, uint32_t seed)
{
    __shared__ uint32_t lds[1024];
    lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 256] = threadIdx.x * 3u; lds[threadIdx.x + 512] = 7u; lds[threadIdx.x + 768] = 1u;
    __syncthreads();
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3u, a2 = a0 ^ 5u, a3 = a0 + 7u, a4 = a0 + 11u, a5 = a0 + 13u, a6 = a0 + 17u, a7 = a0 + 19u;
    uint32_t s0 = seed;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < ITERS; ++i) {
        if (KIND == 0) {
            asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                         "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));
        } else if (KIND == 1) {
            asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_u32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n"
                         "v_cmp_lt_u32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc\n v_cmp_lt_u32 vcc, %5, %4\n v_cndmask_b32 %7, %7, %6, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");
        } else if (KIND == 2) {
            asm volatile("v_add_u32 %0, %0, %9\n s_add_u32 %8, %8, 1\n v_add_u32 %1, %1, %9\n s_add_u32 %8, %8, 1\n v_add_u32 %2, %2, %9\n s_add_u32 %8, %8, 1\n v_add_u32 %3, %3, %9\n s_add_u32 %8, %8, 1\n"
                         "v_add_u32 %4, %4, %9\n s_add_u32 %8, %8, 1\n v_add_u32 %5, %5, %9\n s_add_u32 %8, %8, 1\n v_add_u32 %6, %6, %9\n s_add_u32 %8, %8, 1\n v_add_u32 %7, %7, %9\n s_add_u32 %8, %8, 1\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0) : "v"(seed) : "scc");
        } else if (KIND == 3) {
            // dependent LDS chain: address from the last value, 8 links per iteration
#pragma unroll
            for (int u = 0; u < 8; ++u) a0 = lds[(a0 + a1) & 1023u];
        } else {
            asm volatile("v_and_or_b32 %0, %0, %8, %1\n v_lshl_add_u32 %1, %1, 1, %2\n v_and_or_b32 %2, %2, %8, %3\n v_lshl_add_u32 %3, %3, 1, %4\n"
                         "v_and_or_b32 %4, %4, %8, %5\n v_lshl_add_u32 %5, %5, 1, %6\n v_and_or_b32 %6, %6, %8, %7\n v_lshl_add_u32 %7, %7, 1, %0\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + s0;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char *name, int per_iter)
{
    int dev = 0; hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, dev));
    const int n_cu = prop.multiProcessorCount;
    uint32_t *out; unsigned long long *cyc;
    CHECK(hipMalloc(&out, (size_t)n_cu * 8 * 256 * 4)); CHECK(hipMalloc(&cyc, (size_t)n_cu * 8 * 8));
    for (int k : {1, 2, 3, 4, 6, 8}) {
        const int grid = n_cu * k;
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_issue<KIND>, dim3(grid), dim3(256), 0, 0, out, cyc, 1u);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_issue<KIND>, dim3(grid), dim3(256), 0, 0, out, cyc, 1u);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long *h = (unsigned long long *)malloc((size_t)grid * 8);
        CHECK(hipMemcpy(h, cyc, (size_t)grid * 8, hipMemcpyDeviceToHost));
        double mean = 0; for (int i = 0; i < grid; ++i) mean += (double)h[i]; mean /= grid;
        free(h);
        const double insts = (double)ITERS * per_iter;         // per wave
        printf("%-28s waves/SIMD %d: %.2f ms, %.0f cycles per wave (s_memtime), %.2f cycles per wave-instruction, SIMD issues one per %.2f cycles\n",
               name, k, ms, mean, mean / insts, mean / insts / k);
    }
    CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

int main()
{
    setvbuf(stdout, NULL, _IOLBF, 0);
    run<0>("v_add_u32 x8 independent", 8);
    run<1>("v_cmp + v_cndmask x4", 8);
    run<2>("v_add_u32 / s_add_u32 x8", 16);
    run<4>("v_and_or / v_lshl_add x8", 8);
    run<3>("ds_read_b32 dependent x8", 8);
    return 0;
}
