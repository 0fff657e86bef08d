// Rate of s_memtime under three loads: scalar spin only, VALU busy, MFMA busy (one wave per SIMD on every CU).
// hipcc --offload-arch=gfx950 -O2 memtime_rate.hip -o /tmp/memtime_rate && /tmp/memtime_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void spin(int mode, long iters, unsigned long long* ticks, float* sink)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    f32x16 acc = {0}, acc2 = {0};
    f16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)0.5f; }
    float x = threadIdx.x;
    for (long i = 0; i < iters; i++) {
        if (mode == 1) {
#pragma unroll
            for (int j = 0; j < 32; j++) x = __builtin_fmaf(x, 1.0001f, 0.5f);
        } else if (mode == 2) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc2, 0, 0, 0);
            }
        } else {
            __builtin_amdgcn_s_sleep(8);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    if (x == 12345.f || acc[0] + acc2[0] == 12345.f) sink[0] = x;
}
int main()
{
    unsigned long long* ticks; float* sink;
    (void)hipMalloc(&ticks, 256 * 8); (void)hipMalloc(&sink, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[] = {"s_sleep only", "VALU fma chain, 4 waves per CU", "MFMA 32x32x16 f16 back to back, 4 waves per CU"};
    const long iters[] = {200000, 400000, 400000};
    for (int rep = 0; rep < 2; rep++)
        for (int mode = 0; mode < 3; mode++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, 0, mode, iters[mode], ticks, sink);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[256]; (void)hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
            double mean = 0; for (int i = 0; i < 256; i++) mean += h[i]; mean /= 256;
            printf("%-50s %8.2f ms  %12.0f ticks  -> %.3f GHz", names[mode], ms, mean, mean / (ms * 1e6));
            if (mode == 2) printf("  (%.0f TFLOP/s of fp16 MFMA)", 256.0 * 4 * iters[mode] * 8 * 32768 / (ms * 1e-3) / 1e12);
            printf("\n");
        }
    return 0;
}
