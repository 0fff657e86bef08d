// k_binning.hip -- instance binning (gfx950): scan of tiles_touched, duplicate-with-
// keys, key sort, per-tile ranges.  Integer/byte work, HBM-bound; results are
// bit-identical to the reference's CUB pipeline (RAST/cuda_rasterizer/
// rasterizer_impl.cu:307-348): keys = (tile << 32) | float_bits(view z), stable
// order, ranges[tile] = [first, last+1).
#include "gft_internal.h"

#include <cstring>
#include <rocprim/rocprim.hpp>

namespace {

// ---- scan: level 2 (block sums -> exclusive block offsets + total) ---------
// One workgroup; nblocks <= a few thousand (P / 256).
__global__ __launch_bounds__(1024) void k_scan_block_sums(uint32_t* __restrict__ scan_tmp, int nblocks)
{
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    uint32_t* sums = scan_tmp + GFT_SCAN_BLOCKS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < nblocks; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = (i < nblocks) ? sums[i] : 0u;
        // inclusive scan inside the wave (Hillis-Steele over shuffles)
        uint32_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wtot[wave] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; w++) woff += wtot[w];
        const uint32_t carry = carry_s;
        if (i < nblocks) sums[i] = carry + woff + x - v;  // exclusive offset of block i
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) scan_tmp[GFT_SCAN_TOTAL] = carry_s;
}

// ---- scan: level 3 (inclusive offsets per Gaussian) -------------------------
__global__ __launch_bounds__(GFT_BLOCK) void k_scan_final(int P, const uint32_t* __restrict__ tiles,
                                                          const uint32_t* __restrict__ scan_tmp,
                                                          uint32_t* __restrict__ offsets)
{
    __shared__ uint32_t wtot[GFT_BLOCK / 64];
    const int idx = blockIdx.x * GFT_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t v = (idx < P) ? tiles[idx] : 0u;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) wtot[wave] = x;
    __syncthreads();
    uint32_t woff = scan_tmp[GFT_SCAN_BLOCKS + blockIdx.x];
    for (int w = 0; w < wave; w++) woff += wtot[w];
    if (idx < P) offsets[idx] = woff + x;
}

// ---- duplicate with keys (reference K3) ---------------------------------------
__global__ __launch_bounds__(GFT_BLOCK) void k_duplicate(int P, int gx, int gy, const float4* __restrict__ rec_a,
                                                         const float* __restrict__ depth,
                                                         const uint32_t* __restrict__ offsets,
                                                         const uint32_t* __restrict__ tiles,
                                                         const int32_t* __restrict__ radii,
                                                         uint64_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const int idx = blockIdx.x * GFT_BLOCK + threadIdx.x;
    if (idx >= P) return;
    const int radius = radii[idx];
    if (radius > 0) {
        uint32_t off = offsets[idx] - tiles[idx];
        const float4 a0 = rec_a[2 * idx];
        int x0, y0, x1, y1;
        gft_get_rect(a0.x, a0.y, radius, gx, gy, x0, y0, x1, y1);
        const uint64_t dbits = (uint64_t)__float_as_uint(depth[idx]);
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                keys[off] = ((uint64_t)(uint32_t)(y * gx + x) << 32) | dbits;
                vals[off] = (uint32_t)idx;
                off++;
            }
    }
}

// ---- tile ranges (reference K5) -------------------------------------------------
__global__ __launch_bounds__(GFT_BLOCK) void k_tile_ranges(uint32_t R, const uint64_t* __restrict__ keys,
                                                           uint2* __restrict__ ranges)
{
    const uint32_t idx = blockIdx.x * GFT_BLOCK + threadIdx.x;
    if (idx >= R) return;
    const uint32_t cur = (uint32_t)(keys[idx] >> 32);
    if (idx == 0)
        ranges[cur].x = 0;
    else {
        const uint32_t prev = (uint32_t)(keys[idx - 1] >> 32);
        if (cur != prev) {
            ranges[prev].y = idx;
            ranges[cur].x = idx;
        }
    }
    if (idx == R - 1) ranges[cur].y = R;
}

}  // namespace

uint32_t gft_higher_msb(uint32_t n)
{
    // smallest b with (n >> b) == 0, found by bisection from bit 16 (matches the
    // reference getHigherMsb for every n, incl. powers of two)
    uint32_t msb = 16, step = 16;
    while (step > 1) {
        step >>= 1;
        msb = (n >> msb) ? msb + step : msb - step;
    }
    if (n >> msb) msb++;
    return msb;
}

size_t gft_sort_tmp_bytes(int64_t R)
{
    if (R <= 0) return 0;
    size_t bytes = 0;
    uint64_t* k = nullptr;
    uint32_t* v = nullptr;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)R, 0u, 64u, (hipStream_t)0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        bytes = (size_t)R * 12 + (4u << 20);  // no device visible (size query on a CPU-only host)
    }
    return bytes;
}

hipError_t gft_launch_scan(hipStream_t s, int32_t P, const GeomView& g)
{
    const int blocks = (P + GFT_BLOCK - 1) / GFT_BLOCK;
    hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(1024), 0, s, g.scan_tmp, blocks);
    hipLaunchKernelGGL(k_scan_final, dim3(blocks), dim3(GFT_BLOCK), 0, s, P, g.tiles, g.scan_tmp, g.offsets);
    return hipGetLastError();
}

hipError_t gft_launch_duplicate(hipStream_t s, const gft_config& c, const GeomView& g, const int32_t* radii,
                                const BinView& b)
{
    const int blocks = (c.P + GFT_BLOCK - 1) / GFT_BLOCK;
    const int gx = (c.W + GFT_TILE_X - 1) / GFT_TILE_X, gy = (c.H + GFT_TILE_Y - 1) / GFT_TILE_Y;
    hipLaunchKernelGGL(k_duplicate, dim3(blocks), dim3(GFT_BLOCK), 0, s, c.P, gx, gy, g.rec_a, g.depth, g.offsets,
                       g.tiles, radii, b.keys_unsorted, b.vals_unsorted);
    return hipGetLastError();
}

hipError_t gft_launch_sort(hipStream_t s, int64_t R, int end_bit, const BinView& b)
{
    if (R <= 0) return hipSuccess;
    size_t bytes = b.sort_tmp_bytes;
    return rocprim::radix_sort_pairs(b.sort_tmp, bytes, b.keys_unsorted, b.keys, b.vals_unsorted, b.point_list,
                                     (size_t)R, 0u, (unsigned)end_bit, s);
}

hipError_t gft_launch_ranges(hipStream_t s, int64_t R, int T, const BinView& b, const ImgView& im)
{
    hipError_t e = hipMemsetAsync(im.ranges, 0, (size_t)T * sizeof(uint2), s);
    if (e != hipSuccess) return e;
    if (R > 0) {
        const int blocks = (int)((R + GFT_BLOCK - 1) / GFT_BLOCK);
        hipLaunchKernelGGL(k_tile_ranges, dim3(blocks), dim3(GFT_BLOCK), 0, s, (uint32_t)R, b.keys, im.ranges);
    }
    return hipGetLastError();
}
