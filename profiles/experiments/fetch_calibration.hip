// fetch_calibration.hip -- what rocprofv3's FETCH_SIZE reports on this box for the access kinds of the rasterizer's kernels
// (VERDICT r5 item 4).  Every kernel reads a known number of bytes exactly once from arrays far larger than the 256 MiB
// Infinity Cache; profiles/fetch_calibration.sh runs it under `rocprofv3 --pmc FETCH_SIZE` and divides.
//   k_stream16      16 B per lane, consecutive lanes consecutive addresses (preprocess / Adam / gradient writes' reads)
//   k_stream4        4 B per lane, consecutive (id lists, flags)
//   k_gather32      one 32-byte record per lane at a scattered index (two 16-byte loads): rec_a OR rec_b of the blends
//   k_gather32x2    two 32-byte records per lane from two arrays at the same scattered index: the blends' rec_a AND rec_b
//   k_gather64      one 64-byte record per lane at a scattered index (four 16-byte loads): the merged-record layout
//   k_gather_rows   a 192-byte row per lane at a scattered index (12 x 16 B): the appearance kernel's SH rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void sink(float s, float* out) { if (s == 12345.678f) out[0] = s; }
// a bijection of [0, n), n a power of two: consecutive i land far apart
__device__ __forceinline__ uint32_t scatter(uint32_t i, uint32_t n) { return (i * 2654435761u + 12345u) & (n - 1u); }

__global__ __launch_bounds__(256) void k_stream16(const float4* __restrict__ a, size_t n, float* out)
{
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    sink(s, out);
}
__global__ __launch_bounds__(256) void k_stream4(const float* __restrict__ a, size_t n, float* out)
{
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += a[i];
    sink(s, out);
}
template <int F4>
__global__ __launch_bounds__(256) void k_gather(const float4* __restrict__ a, uint32_t n, float* out)
{
    float s = 0.f;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const size_t id = scatter(i, n);
#pragma unroll
        for (int k = 0; k < F4; k++) { const float4 v = a[id * F4 + k]; s += v.x + v.y + v.z + v.w; }
    }
    sink(s, out);
}
__global__ __launch_bounds__(256) void k_gather32x2(const float4* __restrict__ a, const float4* __restrict__ b, uint32_t n, float* out)
{
    float s = 0.f;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const size_t id = scatter(i, n);
        const float4 v0 = a[2 * id], v1 = a[2 * id + 1], w0 = b[2 * id], w1 = b[2 * id + 1];
        s += v0.x + v1.y + w0.z + w1.w;
    }
    sink(s, out);
}

int main()
{
    const uint32_t n = 1u << 24;                       // 16 M records
    const size_t bytes = (size_t)n * 192;              // 3 GB: every kernel reads a prefix of it
    float4 *a, *b; float* out;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, (size_t)n * 32)); CHECK(hipMalloc(&out, 4));
    CHECK(hipMemset(a, 0, bytes)); CHECK(hipMemset(b, 0, (size_t)n * 32));
    const int blocks = 8192;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_stream16, dim3(blocks), dim3(256), 0, 0, a, (size_t)n * 4, out);             // 1 GiB
        hipLaunchKernelGGL(k_stream4, dim3(blocks), dim3(256), 0, 0, (const float*)a, (size_t)n * 16, out);  // 1 GiB
        hipLaunchKernelGGL(k_gather<2>, dim3(blocks), dim3(256), 0, 0, a, n, out);                          // 16 M x 32 B = 512 MiB
        hipLaunchKernelGGL(k_gather32x2, dim3(blocks), dim3(256), 0, 0, a, b, n, out);                      // 2 x 512 MiB
        hipLaunchKernelGGL(k_gather<4>, dim3(blocks), dim3(256), 0, 0, a, n, out);                          // 16 M x 64 B = 1 GiB
        hipLaunchKernelGGL(k_gather<12>, dim3(blocks), dim3(256), 0, 0, a, n, out);                         // 16 M x 192 B = 3 GiB
    }
    CHECK(hipDeviceSynchronize());
    printf("bytes_read stream16 %zu stream4 %zu gather32 %zu gather32x2 %zu gather64 %zu gather_rows %zu\n", (size_t)n * 64, (size_t)n * 64,
           (size_t)n * 32, (size_t)n * 64, (size_t)n * 64, (size_t)n * 192);
    return 0;
}
