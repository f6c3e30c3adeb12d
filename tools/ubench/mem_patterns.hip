// mem_patterns.hip -- diagnostics, not part of the product: what the two "one lane per read" memory patterns of the
// classification kernel cost on gfx950 against their coalesced forms.
//   stores: every lane writes the run of its read's exons (n ~ 8 consecutive elements of two int32 arrays and one
//           byte array at the read's prefix offset)            vs   the wave writes the same bytes 64 elements at a time
//   loads:  every lane reads its read's CIGAR words (c ~ 15 consecutive dwords, 4-byte aligned, as 16-byte loads)
//           vs   the wave reads the same range 64 x 16 bytes at a time
// hipcc --offload-arch=gfx950 -O3 -o mem_patterns mem_patterns.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u4v __attribute__((ext_vector_type(4), aligned(4)));

// ---- stores
__global__ __launch_bounds__(256) void k_store_lane(const uint32_t *__restrict__ off, int64_t n_reads, int32_t *__restrict__ xs, int32_t *__restrict__ xe, uint8_t *__restrict__ xf)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_reads; r += (int64_t)gridDim.x * 256) {
        const uint32_t a = off[r], b = off[r + 1];
        for (uint32_t i = a; i < b; ++i) { xs[i] = (int32_t)i; xe[i] = (int32_t)(i + 7u); xf[i] = (uint8_t)i; }
    }
}
// the wave's 64 reads own a contiguous run [off[r0], off[r0 + 64]): written 64 elements per instruction
__global__ __launch_bounds__(256) void k_store_wave(const uint32_t *__restrict__ off, int64_t n_reads, int32_t *__restrict__ xs, int32_t *__restrict__ xe, uint8_t *__restrict__ xf)
{
    const int lane = threadIdx.x & 63;
    for (int64_t r0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64; r0 < n_reads; r0 += (int64_t)gridDim.x * 256) {
        const int64_t r1 = r0 + 64 < n_reads ? r0 + 64 : n_reads;
        const uint32_t a = off[r0], b = off[r1];
        for (uint32_t i = a + lane; i < b; i += 64) { xs[i] = (int32_t)i; xe[i] = (int32_t)(i + 7u); xf[i] = (uint8_t)i; }
    }
}
// ---- loads
__global__ __launch_bounds__(256) void k_load_lane(const uint32_t *__restrict__ coff, int64_t n_reads, const uint32_t *__restrict__ cig, uint32_t *__restrict__ out)
{
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_reads; r += (int64_t)gridDim.x * 256) {
        const uint32_t a = coff[r], b = coff[r + 1];
        uint32_t acc = 0;
        for (uint32_t i = a; i < b; i += 4) {                     // (the array is padded: whole vectors may be read)
            const u4v v = *reinterpret_cast<const u4v *>(cig + i);
            acc += v.x + (i + 1 < b ? v.y : 0u) + (i + 2 < b ? v.z : 0u) + (i + 3 < b ? v.w : 0u);
        }
        out[r] = acc;
    }
}
__global__ __launch_bounds__(256) void k_load_wave(const uint32_t *__restrict__ coff, int64_t n_reads, const uint32_t *__restrict__ cig, uint32_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    for (int64_t r0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64; r0 < n_reads; r0 += (int64_t)gridDim.x * 256) {
        const int64_t r1 = r0 + 64 < n_reads ? r0 + 64 : n_reads;
        const uint32_t a = coff[r0] & ~3u, b = coff[r1];
        uint32_t acc = 0;
        for (uint32_t i = a + 4u * lane; i < b; i += 256) {
            const uint4 v = *reinterpret_cast<const uint4 *>(cig + i);
            acc += v.x + v.y + v.z + v.w;
        }
        out[r0 + lane < n_reads ? r0 + lane : r0] = acc;
    }
}

template <typename F>
static float time_ms(F launch, int reps)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main()
{
    setvbuf(stdout, NULL, _IOLBF, 0);
    const int64_t N = 10000000;
    std::vector<uint32_t> off(N + 1), coff(N + 1);
    uint64_t s = 88172645463325252ull; uint32_t x = 0, c = 0;
    for (int64_t i = 0; i < N; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const uint32_t n = 3 + (uint32_t)(s % 11);               // 3..13 exons, mean 8
        off[i] = x; coff[i] = c; x += n; c += 2 * n - 1;
    }
    off[N] = x; coff[N] = c;
    uint32_t *d_off, *d_coff, *d_cig, *d_out; int32_t *xs, *xe; uint8_t *xf;
    CHECK(hipMalloc(&d_off, (N + 1) * 4)); CHECK(hipMalloc(&d_coff, (N + 1) * 4)); CHECK(hipMalloc(&d_cig, ((size_t)c + 1024) * 4)); CHECK(hipMalloc(&d_out, N * 4));
    CHECK(hipMalloc(&xs, (size_t)x * 4)); CHECK(hipMalloc(&xe, (size_t)x * 4)); CHECK(hipMalloc(&xf, (size_t)x));
    CHECK(hipMemcpy(d_off, off.data(), (N + 1) * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_coff, coff.data(), (N + 1) * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_cig, 1, ((size_t)c + 1024) * 4));
    const double sb = 9.0 * x + 4.0 * N, lb = 4.0 * c + 8.0 * N;
    for (int wg : {2048, 4096, 8192}) {
        float a = time_ms([&] { hipLaunchKernelGGL(k_store_lane, dim3(wg), dim3(256), 0, 0, d_off, N, xs, xe, xf); }, 5);
        float b = time_ms([&] { hipLaunchKernelGGL(k_store_wave, dim3(wg), dim3(256), 0, 0, d_off, N, xs, xe, xf); }, 5);
        printf("stores (%.2f GB), grid %d: one run per lane %.3f ms = %.0f GB/s | wave-coalesced %.3f ms = %.0f GB/s\n", sb / 1e9, wg, a, sb / a / 1e6, b, sb / b / 1e6);
        float l1 = time_ms([&] { hipLaunchKernelGGL(k_load_lane, dim3(wg), dim3(256), 0, 0, d_coff, N, d_cig, d_out); }, 5);
        float l2 = time_ms([&] { hipLaunchKernelGGL(k_load_wave, dim3(wg), dim3(256), 0, 0, d_coff, N, d_cig, d_out); }, 5);
        printf("loads  (%.2f GB), grid %d: one run per lane %.3f ms = %.0f GB/s | wave-coalesced %.3f ms = %.0f GB/s\n", lb / 1e9, wg, l1, lb / l1 / 1e6, l2, lb / l2 / 1e6);
    }
    return 0;
}
