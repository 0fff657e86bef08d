#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ a, size_t n, float* out)
{
    float4 s = make_float4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 v = a[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (s.x + s.y + s.z + s.w == 12345.f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void k_write(float4* __restrict__ a, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] = make_float4(1, 2, 3, 4);
}
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
int main()
{
    const size_t bytes = (size_t)2 << 30, n = bytes / 16;
    float4 *a, *b; float* out; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&out, 4); hipMemset(a, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {2048, 8192, 32768}) {
        float ms;
        hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, n, out);
        hipEventRecord(e0); for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, n, out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("blocks %6d read  %.2f TB/s\n", blocks, 5.0 * bytes / ms / 1e9);
        hipEventRecord(e0); for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, b, n); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("blocks %6d write %.2f TB/s\n", blocks, 5.0 * bytes / ms / 1e9);
        hipEventRecord(e0); for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("blocks %6d copy  %.2f TB/s (read + write)\n", blocks, 10.0 * bytes / ms / 1e9);
    }
    return 0;
}
