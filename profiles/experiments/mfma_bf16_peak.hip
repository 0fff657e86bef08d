// Probe: what does v_mfma_f32_32x32x16_bf16 sustain with 1, 2 and 4 waves per SIMD?  (The deformation network's three
// big kernels run ONE wave per SIMD -- 256 accumulator registers or 97 KB of LDS per workgroup -- and their counters show
// the matrix pipe 36 % busy.)  Operands fixed in registers, NACC independent accumulators per wave, no memory traffic.
// build: hipcc --offload-arch=gfx950 -O3 mfma_bf16_peak.hip -o mfma_bf16_peak ; run: ./mfma_bf16_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int WAVES_PER_BLOCK>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void k(float* out, int iters)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++)
        for (int q = 0; q < 16; q++) acc[i][q] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (__bf16)(1.0f + threadIdx.x * 1e-3f + j); b[j] = (__bf16)(0.5f + j); }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 96 / NACC; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; i++)
        for (int q = 0; q < 16; q++) s += acc[i][q];
    if (s == 12345.f) out[0] = s;
}

template <int NACC, int WPB>
void run(int blocks, int iters, const char* name)
{
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC, WPB>), dim3(blocks), dim3(64 * WPB), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double mfma = (double)blocks * WPB * iters * (96 / NACC * NACC);
        printf("%-44s %.3f ms  %.0f TFLOP/s  (%.1f cycles per MFMA and SIMD at 2.4 GHz)\n", name, ms, mfma * 32768.0 / ms / 1e9,
               ms * 1e-3 * 2.4e9 / (mfma / 1024.0));
    }
    hipFree(out);
}

int main()
{
    // 256 CUs; WPB waves per workgroup, one workgroup per CU unless noted
    run<16, 4>(256, 2000, "1 wave/SIMD, 16 independent accumulators");
    run<4, 4>(256, 2000, "1 wave/SIMD, 4 independent accumulators");
    run<1, 4>(256, 2000, "1 wave/SIMD, 1 accumulator (dependent chain)");
    run<4, 8>(256, 2000, "2 waves/SIMD, 4 accumulators each");
    run<4, 16>(256, 2000, "4 waves/SIMD, 4 accumulators each");
    run<16, 4>(512, 2000, "2 workgroups/CU x 1 wave/SIMD, 16 acc (if they fit)");
    return 0;
}
