#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// MODE 0: constant operands; MODE 1: random per-lane operands, 8 different pairs
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters, long long* clk)
{
    f32x16 acc[4];
    for (int i = 0; i < 4; i++) for (int q = 0; q < 16; q++) acc[i][q] = 0.f;
    float a[8], b[8];
    for (int u = 0; u < 8; u++) {
        a[u] = MODE ? in[(threadIdx.x * 8 + u) % 4096] : 1.0f;
        b[u] = MODE ? in[(threadIdx.x * 8 + u + 2048) % 4096] : 2.0f;
    }
    long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + i) & 7], acc[i], 0, 0, 0);
    }
    long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 4; i++) for (int q = 0; q < 16; q++) s += acc[i][q];
    if (s == 12345.f) out[0] = s;
    if (blockIdx.x == 7 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
template <int MODE>
void run(int blocks, int iters, const char* name, const float* in)
{
    float* out; hipMalloc(&out, 4);
    long long* clk; hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, in, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        double flop = (double)blocks * 4 * iters * 32 * 4096.0;
        printf("%s blocks=%d: %.3f ms %.1f TF  clock64/wall = %.3f (x100MHz = %.0f MHz)\n", name, blocks, ms, flop / ms / 1e9,
               (double)h[0] / h[1], 100.0 * h[0] / h[1]);
    }
}
int main()
{
    float* in; hipMalloc(&in, 4096 * 4);
    float h[4096]; srand(1); for (int i = 0; i < 4096; i++) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>(512, 2000, "const ", in);
    run<1>(512, 2000, "random", in);
    run<1>(512, 8000, "random long", in);
    run<0>(512, 8000, "const long", in);
    return 0;
}
