// Does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs on gfx950, and does v_cvt_pk_f16_f32 produce them?
// hipcc --offload-arch=gfx950 -O2 mfma_f16_subnormal.hip -o mfma_f16_subnormal && ./mfma_f16_subnormal
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float a_val, float b_val, float* out, unsigned* bits)
{
    f16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) {
        out[0] = acc[0];
        unsigned short u;
        _Float16 h = (_Float16)a_val;
        __builtin_memcpy(&u, &h, 2);
        bits[0] = u;
    }
}
int main()
{
    float* out; unsigned* bits;
    hipMalloc(&out, 4); hipMalloc(&bits, 4);
    const float cases[][2] = {{1e-5f, 1.0f}, {1e-6f, 1024.0f}, {3e-7f, 1.0f}, {6e-5f, 1.0f}, {1.0f, 1e-5f}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, c[0], c[1], out, bits);
        float h; unsigned u;
        hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost); hipMemcpy(&u, bits, 4, hipMemcpyDeviceToHost);
        printf("a=%g b=%g  fp16(a) bits=0x%04x  mfma sum over k=16: %.9g  (exact with subnormals kept: about %.9g)\n", c[0], c[1], u, h, 16.0 * c[0] * c[1]);
    }
    return 0;
}
