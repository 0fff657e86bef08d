// Prototype: fp32-like GEMM from three bf16 planes per operand (hi/mid/lo), six bf16 MFMAs per product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
static inline uint16_t f2bf(float x) { uint32_t u; memcpy(&u, &x, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static inline float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float x; memcpy(&x, &u, 4); return x; }
// A planes: [3][rows][256] bf16 row-major;  W planes: [3][256/8][256][8] bf16 (k-interleaved by 8)
template <int NTERMS>
__global__ __launch_bounds__(256) void k(const uint16_t* __restrict__ Ap, const uint16_t* __restrict__ Wp, float* __restrict__ C, int tiles, int iters)
{
    __shared__ uint4 sA[3][64][33];          // 64 rows x 256 bf16 (= 32 uint4) + pad
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    for (int it = 0; it < iters; it++) {
        const int tile = (blockIdx.x + it * gridDim.x) % tiles;
        __syncthreads();
        for (int q = tid; q < 3 * 64 * 32; q += 256) {
            const int p = q / (64 * 32), rem = q % (64 * 32), row = rem / 32, c = rem % 32;
            sA[p][row][c] = reinterpret_cast<const uint4*>(Ap)[((size_t)p * tiles * 64 + (size_t)tile * 64 + row) * 32 + c];
        }
        __syncthreads();
        f32x16 acc[2][2];
        for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) for (int q = 0; q < 16; q++) acc[a][b][q] = 0.f;
        const int n0 = wave * 64;
        for (int ks = 0; ks < 16; ks++) {          // 16 k per step
            bf16x8 a[3][2], w[3][2];
            for (int p = 0; p < 3; p++)
                for (int t = 0; t < 2; t++) {
                    const uint4 av = sA[p][32 * t + r][2 * ks + h];
                    memcpy(&a[p][t], &av, 16);
                    const uint4 wv = reinterpret_cast<const uint4*>(Wp)[((size_t)p * 32 + 2 * ks + h) * 256 + n0 + 32 * t + r];
                    memcpy(&w[p][t], &wv, 16);
                }
            // terms kept: (hi,hi) (hi,mid) (mid,hi) (hi,lo) (lo,hi) (mid,mid)
            const int pa[6] = {0, 0, 1, 0, 2, 1}, pw[6] = {0, 1, 0, 2, 0, 1};
#pragma unroll
            for (int term = 0; term < NTERMS; term++)
#pragma unroll
                for (int rt = 0; rt < 2; rt++)
#pragma unroll
                    for (int ct = 0; ct < 2; ct++)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[pw[term]][ct], a[pa[term]][rt], acc[rt][ct], 0, 0, 0);
        }
        if (it == 0 && blockIdx.x < (unsigned)tiles)
            for (int rt = 0; rt < 2; rt++) for (int ct = 0; ct < 2; ct++) for (int q = 0; q < 16; q++) {
                const int col = n0 + 32 * ct + (q & 3) + 8 * (q >> 2) + 4 * h, row = 32 * rt + r;     // transposed tile: rows of D = columns of W
                C[((size_t)tile * 64 + row) * 256 + col] = acc[rt][ct][q];
            }
    }
}
int main()
{
    const int tiles = 512, K = 256, N = 256, M = tiles * 64;
    std::vector<float> A((size_t)M * K), W((size_t)K * N);
    srand(3);
    for (auto& v : A) v = (float)rand() / RAND_MAX;                  // activations >= 0 (post-ReLU like)
    for (auto& v : W) v = ((float)rand() / RAND_MAX - 0.5f) * 0.2f;
    std::vector<uint16_t> Ap((size_t)3 * M * K), Wp((size_t)3 * K * N);
    for (size_t i = 0; i < A.size(); i++) {
        float x = A[i]; uint16_t hi = f2bf(x); float r1 = x - bf2f(hi); uint16_t mid = f2bf(r1); float r2 = r1 - bf2f(mid); uint16_t lo = f2bf(r2);
        Ap[i] = hi; Ap[(size_t)M * K + i] = mid; Ap[(size_t)2 * M * K + i] = lo;
    }
    for (int k = 0; k < K; k++) for (int n = 0; n < N; n++) {
        float x = W[(size_t)k * N + n]; uint16_t hi = f2bf(x); float r1 = x - bf2f(hi); uint16_t mid = f2bf(r1); float r2 = r1 - bf2f(mid); uint16_t lo = f2bf(r2);
        const size_t o = ((size_t)(k / 8) * N + n) * 8 + (k % 8);
        Wp[o] = hi; Wp[(size_t)K * N + o] = mid; Wp[(size_t)2 * K * N + o] = lo;
    }
    uint16_t *dA, *dW; float* dC;
    hipMalloc(&dA, Ap.size() * 2); hipMalloc(&dW, Wp.size() * 2); hipMalloc(&dC, (size_t)M * N * 4);
    hipMemcpy(dA, Ap.data(), Ap.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dW, Wp.data(), Wp.size() * 2, hipMemcpyHostToDevice);
    auto run = [&](auto kern, int nterms, const char* name) {
        hipMemset(dC, 0, (size_t)M * N * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 40;
        hipLaunchKernelGGL(kern, dim3(512), dim3(256), 0, 0, dA, dW, dC, tiles, 2);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(512), dim3(256), 0, 0, dA, dW, dC, tiles, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<float> C((size_t)64 * N);
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0, norm = 0, worst32 = 0;
        for (int i = 0; i < 64; i++) for (int n = 0; n < N; n++) {
            double ref = 0; float f32 = 0.f;
            for (int k2 = 0; k2 < K; k2++) { ref += (double)A[(size_t)i * K + k2] * W[(size_t)k2 * N + n]; f32 = fmaf(A[(size_t)i * K + k2], W[(size_t)k2 * N + n], f32); }
            worst = fmax(worst, fabs(C[(size_t)i * N + n] - ref)); worst32 = fmax(worst32, fabs((double)f32 - ref)); norm = fmax(norm, fabs(ref));
        }
        const double flop = 2.0 * 512 * iters * 64.0 * K * N;
        printf("%s: %.3f ms  %.1f TFLOP/s (fp32-equivalent)  max err / max|ref| = %.2e   (plain fp32 fma chain: %.2e)\n", name, ms, flop / ms / 1e9, worst / norm, worst32 / norm);
    };
    run(k<6>, 6, "bf16 x 6 terms");
    run(k<3>, 3, "bf16 x 3 terms (hi*hi, hi*mid, mid*hi)");
    run(k<1>, 1, "bf16 x 1 term");
    return 0;
}
