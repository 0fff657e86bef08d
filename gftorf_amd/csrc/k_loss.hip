// k_loss.hip -- SSIM + squared error of two images, forward and backward, one launch each (include/gftorf_loss.h;
// utils/loss_utils.py:51-53, 76-123).  One 16x16 tile of one channel per workgroup: the 26x26 input patch (window 11, zero
// padding outside the image) goes through LDS, the window is applied as its two 1-D factors (rows, then columns).
#include "gft_internal.h"
#include "gftorf_loss.h"

namespace {

constexpr int TS = GFT_SSIM_TILE, WN = GFT_SSIM_WINDOW, HALO = WN / 2, PS = TS + 2 * HALO;      // 16, 11, 5, 26

struct LossArgs {
    int C, H, W, tiles_x, tiles_y;
    const float* __restrict__ a;        // img1
    const float* __restrict__ b;        // img2
    float* maps;                        // [3][C][H][W]
    float* partials;                    // [blocks][2]
    const float* __restrict__ g_ssim; const float* __restrict__ g_l2;
    float scale_ssim, scale_l2;
    float* grad;
    float w[WN];
};

// rows pass: Q quantities of a [PS][PS] patch -> [PS][TS]; columns pass by the caller
template <int Q>
__device__ __forceinline__ void rows_pass(const float (*src)[PS][PS + 1], float (*dst)[PS][TS + 1], const float* w, int tid)
{
    for (int i = tid; i < PS * TS; i += TS * TS) {
        const int r = i / TS, c = i % TS;
#pragma unroll
        for (int q = 0; q < Q; q++) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < WN; k++) s = fmaf(w[k], src[q][r][c + k], s);
            dst[q][r][c] = s;
        }
    }
}

__global__ __launch_bounds__(TS * TS) void k_ssim_l2_fwd(LossArgs p)
{
    __shared__ float sIn[5][PS][PS + 1];          // a, b, a a, b b, a b
    __shared__ float sRow[5][PS][TS + 1];
    __shared__ float sRed[2][TS * TS / 64];
    const int tid = threadIdx.x;
    const int ch = blockIdx.x / (p.tiles_x * p.tiles_y), t = blockIdx.x % (p.tiles_x * p.tiles_y);
    const int x0 = (t % p.tiles_x) * TS, y0 = (t / p.tiles_x) * TS;
    const size_t plane = (size_t)ch * p.H * p.W;
    for (int i = tid; i < PS * PS; i += TS * TS) {
        const int r = i / PS, c = i % PS, y = y0 + r - HALO, x = x0 + c - HALO;
        float va = 0.f, vb = 0.f;
        if (y >= 0 && y < p.H && x >= 0 && x < p.W) { va = p.a[plane + (size_t)y * p.W + x]; vb = p.b[plane + (size_t)y * p.W + x]; }
        sIn[0][r][c] = va; sIn[1][r][c] = vb; sIn[2][r][c] = va * va; sIn[3][r][c] = vb * vb; sIn[4][r][c] = va * vb;
    }
    __syncthreads();
    rows_pass<5>(sIn, sRow, p.w, tid);
    __syncthreads();
    const int ty = tid / TS, tx = tid % TS, y = y0 + ty, x = x0 + tx;
    float v[5];
#pragma unroll
    for (int q = 0; q < 5; q++) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < WN; k++) s = fmaf(p.w[k], sRow[q][ty + k][tx], s);
        v[q] = s;
    }
    float ssim = 0.f, sq = 0.f;
    if (y < p.H && x < p.W) {
        // loss_utils.py:101-117
        const float mu1 = v[0], mu2 = v[1];
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float s1 = v[2] - mu1_sq, s2 = v[3] - mu2_sq, s12 = v[4] - mu12;
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float A1 = 2.f * mu12 + C1, A2 = 2.f * s12 + C2, B1 = mu1_sq + mu2_sq + C1, B2 = s1 + s2 + C2;
        const float rB1 = 1.f / B1, rB2 = 1.f / B2;
        ssim = (A1 * A2) * (rB1 * rB2);
        const float va = sIn[0][ty + HALO][tx + HALO], vb = sIn[1][ty + HALO][tx + HALO];
        const float d = va - vb;
        sq = d * d;
        if (p.maps) {
            // d ssim / d (mu1, sigma1^2, sigma12) with the three treated as independent ...
            const float dmu1 = 2.f * mu2 * A2 * rB1 * rB2 - 2.f * mu1 * ssim * rB1;
            const float ds1 = -ssim * rB2;
            const float ds12 = 2.f * A1 * rB1 * rB2;
            // ... and sigma1^2 = E[a a] - mu1^2, sigma12 = E[a b] - mu1 mu2 folded into the mu1 map
            const size_t o = plane + (size_t)y * p.W + x, N = (size_t)p.C * p.H * p.W;
            p.maps[o] = dmu1 - 2.f * mu1 * ds1 - mu2 * ds12;
            p.maps[N + o] = ds1;
            p.maps[2 * N + o] = ds12;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { ssim += __shfl_xor(ssim, o); sq += __shfl_xor(sq, o); }
    if ((tid & 63) == 0) { sRed[0][tid >> 6] = ssim; sRed[1][tid >> 6] = sq; }
    __syncthreads();
    if (tid == 0) {
        float s0 = 0.f, s1 = 0.f;
        for (int k = 0; k < TS * TS / 64; k++) { s0 += sRed[0][k]; s1 += sRed[1][k]; }
        p.partials[2 * (size_t)blockIdx.x] = s0;
        p.partials[2 * (size_t)blockIdx.x + 1] = s1;
    }
}

__global__ __launch_bounds__(TS * TS) void k_ssim_l2_bwd(LossArgs p)
{
    __shared__ float sIn[3][PS][PS + 1];
    __shared__ float sRow[3][PS][TS + 1];
    const int tid = threadIdx.x;
    const int ch = blockIdx.x / (p.tiles_x * p.tiles_y), t = blockIdx.x % (p.tiles_x * p.tiles_y);
    const int x0 = (t % p.tiles_x) * TS, y0 = (t / p.tiles_x) * TS;
    const size_t plane = (size_t)ch * p.H * p.W, N = (size_t)p.C * p.H * p.W;
    for (int i = tid; i < PS * PS; i += TS * TS) {
        const int r = i / PS, c = i % PS, y = y0 + r - HALO, x = x0 + c - HALO;
        const bool in = y >= 0 && y < p.H && x >= 0 && x < p.W;
        const size_t o = plane + (size_t)(in ? y : 0) * p.W + (in ? x : 0);
#pragma unroll
        for (int q = 0; q < 3; q++) sIn[q][r][c] = in ? p.maps[q * N + o] : 0.f;
    }
    __syncthreads();
    rows_pass<3>(sIn, sRow, p.w, tid);
    __syncthreads();
    const int ty = tid / TS, tx = tid % TS, y = y0 + ty, x = x0 + tx;
    if (y >= p.H || x >= p.W) return;
    float v[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < WN; k++) s = fmaf(p.w[k], sRow[q][ty + k][tx], s);
        v[q] = s;
    }
    const size_t o = plane + (size_t)y * p.W + x;
    const float va = p.a[o], vb = p.b[o];
    // the window is symmetric: the adjoint of the zero-padded blur is the same blur
    const float gs = p.g_ssim ? *p.g_ssim * p.scale_ssim : 0.f, gl = p.g_l2 ? *p.g_l2 * p.scale_l2 : 0.f;
    p.grad[o] = gs * (v[0] + 2.f * va * v[1] + vb * v[2]) + gl * 2.f * (va - vb);
}

int fill(LossArgs& p, int32_t C, int32_t H, int32_t W, const float* a, const float* b, const float* window, const char* who)
{
    if (C <= 0 || H <= 0 || W <= 0) return gft_fail("%s: bad sizes C=%d H=%d W=%d", who, C, H, W);
    if (!a || !b || !window) return gft_fail("%s: NULL argument", who);
    p.C = C; p.H = H; p.W = W;
    p.tiles_x = (W + TS - 1) / TS; p.tiles_y = (H + TS - 1) / TS;
    p.a = a; p.b = b;
    for (int k = 0; k < WN; k++) p.w[k] = window[k];
    return 0;
}

}  // namespace

extern "C" int64_t gft_ssim_blocks(int32_t C, int32_t H, int32_t W)
{
    if (C <= 0 || H <= 0 || W <= 0) return 0;
    return (int64_t)C * ((W + TS - 1) / TS) * ((H + TS - 1) / TS);
}

extern "C" int gft_ssim_l2_forward(void* hip_stream, int32_t C, int32_t H, int32_t W, const float* img1, const float* img2,
                                   const float* window, float* maps, float* partials)
{
    LossArgs p = {};
    if (fill(p, C, H, W, img1, img2, window, "gft_ssim_l2_forward")) return 1;
    if (!partials) return gft_fail("gft_ssim_l2_forward: partials is NULL");
    p.maps = maps; p.partials = partials;
    const int64_t blocks = gft_ssim_blocks(C, H, W);
    if (blocks > 0x7fffffffll) return gft_fail("gft_ssim_l2_forward: image too large");
    hipLaunchKernelGGL(k_ssim_l2_fwd, dim3((unsigned)blocks), dim3(TS * TS), 0, (hipStream_t)hip_stream, p);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : gft_fail("gft_ssim_l2_forward: %s", hipGetErrorString(e));
}

extern "C" int gft_ssim_l2_backward(void* hip_stream, int32_t C, int32_t H, int32_t W, const float* img1, const float* img2,
                                    const float* window, const float* maps, const float* g_ssim, const float* g_l2,
                                    float scale_ssim, float scale_l2, float* grad_img1)
{
    LossArgs p = {};
    if (fill(p, C, H, W, img1, img2, window, "gft_ssim_l2_backward")) return 1;
    if (!maps || !grad_img1) return gft_fail("gft_ssim_l2_backward: NULL argument");
    p.maps = const_cast<float*>(maps); p.g_ssim = g_ssim; p.g_l2 = g_l2; p.scale_ssim = scale_ssim; p.scale_l2 = scale_l2;
    p.grad = grad_img1;
    const int64_t blocks = gft_ssim_blocks(C, H, W);
    hipLaunchKernelGGL(k_ssim_l2_bwd, dim3((unsigned)blocks), dim3(TS * TS), 0, (hipStream_t)hip_stream, p);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : gft_fail("gft_ssim_l2_backward: %s", hipGetErrorString(e));
}
