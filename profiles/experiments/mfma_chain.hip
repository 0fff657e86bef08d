// Cycles per v_mfma_f32_32x32x16_f16 for one wave per SIMD as a function of the number of independent accumulators the
// wave rotates through (1 = every multiply waits for the one before it), and for two waves per SIMD.
// hipcc --offload-arch=gfx950 -O2 mfma_chain.hip -o /tmp/mfma_chain && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int D>
__global__ void chain(long iters, unsigned long long* ticks, float* sink)
{
    f32x16 acc[D];
    for (int d = 0; d < D; d++) acc[d] = f32x16{0};
    f16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)0.5f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (long i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 16 / D; j++)
#pragma unroll
            for (int d = 0; d < D; d++) acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[d], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    float s = 0;
    for (int d = 0; d < D; d++) s += acc[d][0];
    if (s == 12345.f) sink[0] = s;
}
template <int D>
void run(int threads, unsigned long long* ticks, float* sink)
{
    const long iters = 20000;
    hipLaunchKernelGGL(chain<D>, dim3(256), dim3(threads), 0, 0, iters, ticks, sink);
    (void)hipDeviceSynchronize();
    unsigned long long h[256 * 8]; (void)hipMemcpy(h, ticks, sizeof(unsigned long long) * 256 * (threads / 64), hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < 256 * (threads / 64); i++) mean += h[i]; mean /= 256 * (threads / 64);
    printf("accumulators %d, waves per SIMD %d: %.1f ticks per MFMA per wave = %.1f per SIMD\n", D, threads / 256, mean / (iters * 16.0), mean / (iters * 16.0) / (threads / 256));
}
int main()
{
    unsigned long long* ticks; float* sink;
    (void)hipMalloc(&ticks, 256 * 8 * 8); (void)hipMalloc(&sink, 4);
    for (int threads = 256; threads <= 512; threads += 256) {
        run<1>(threads, ticks, sink); run<2>(threads, ticks, sink); run<4>(threads, ticks, sink); run<8>(threads, ticks, sink);
    }
    return 0;
}
