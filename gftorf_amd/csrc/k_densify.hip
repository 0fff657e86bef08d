// k_densify.hip -- per-Gaussian bookkeeping around the rasterizer (include/gftorf_densify.h): the
// statistics update of every iteration and the order-preserving row compaction of `t[mask]`.
// HBM-bound byte work: coalesced reads, one pass.
#include "gft_internal.h"
#include "gftorf_densify.h"

namespace {

constexpr int DN_BLOCK = 256;
constexpr int RK_ROWS = 4096;        // rows per rank workgroup (256 threads x 16 mask bytes)

// ---- statistics (scene/gaussian_model.py:648-654, train.py:443) ---------------------------------
__global__ __launch_bounds__(DN_BLOCK) void k_densify_stats(int64_t P, const float* __restrict__ grad, const float* __restrict__ pixels,
                                                            const int32_t* __restrict__ radii, const uint8_t* __restrict__ upd,
                                                            const uint8_t* __restrict__ apply, float* accum, float* denom, float* maxr)
{
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * DN_BLOCK + threadIdx.x;
    if (i >= P || !upd[i]) return;
    if (maxr) maxr[i] = fmaxf(maxr[i], (float)radii[i]);                 // torch.max(float, int32) promotes to float
    if (apply && !apply[i]) return;
    const float px = pixels[i];
    if (accum) {
        // torch.norm(g[:, :2], dim=-1): the reduction `acc + v * v` of torch's norm kernels contracts to an fma,
        // on the host as on the device: sqrt(fma(y, y, x * x))
        const float gx = grad[3 * i], gy = grad[3 * i + 1];
        accum[i] += sqrtf(fmaf(gy, gy, gx * gx)) * px;
    }
    if (denom) denom[i] += px;
}

// ---- rank: per-workgroup counts, the last workgroup scans them ----------------------------------
struct RankArgs {
    int64_t P;
    const uint8_t* mask;
    int32_t* rank;
    uint32_t* block_sum;     // [blocks]
    uint32_t* total;         // 1 word
};

__device__ __forceinline__ uint32_t nz_bytes(uint32_t w)
{
    // number of non-zero bytes of a word (torch.bool holds 0 / 1, any non-zero byte counts)
    return ((w & 0xffu) != 0) + ((w & 0xff00u) != 0) + ((w & 0xff0000u) != 0) + ((w & 0xff000000u) != 0);
}

__global__ __launch_bounds__(DN_BLOCK) void k_rows_count(RankArgs a)
{
    __shared__ uint32_t wsum[DN_BLOCK / 64];
    const int64_t r0 = (int64_t)blockIdx.x * RK_ROWS + threadIdx.x * 16;
    uint32_t c = 0;
    if (r0 + 16 <= a.P && ((uintptr_t)(a.mask + r0) & 15) == 0) {
        const uint4 m = *reinterpret_cast<const uint4*>(a.mask + r0);
        c = nz_bytes(m.x) + nz_bytes(m.y) + nz_bytes(m.z) + nz_bytes(m.w);
    } else {
        for (int k = 0; k < 16; k++)
            if (r0 + k < a.P && a.mask[r0 + k]) c++;
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor((int)c, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) a.block_sum[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// one workgroup: exclusive scan of the block sums in place, total
__global__ __launch_bounds__(1024) void k_rows_scan(uint32_t* block_sum, int nblocks, uint32_t* total)
{
    __shared__ uint32_t part[1024];
    const int per = (nblocks + 1023) / 1024;
    const int b0 = threadIdx.x * per;
    uint32_t s = 0;
    for (int k = 0; k < per; k++)
        if (b0 + k < nblocks) s += block_sum[b0 + k];
    part[threadIdx.x] = s;
    __syncthreads();
    // Hillis-Steele over 1024 partials
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;            // exclusive prefix of this thread's first block
    for (int k = 0; k < per; k++)
        if (b0 + k < nblocks) {
            const uint32_t v = block_sum[b0 + k];
            block_sum[b0 + k] = run;
            run += v;
        }
    if (threadIdx.x == 1023) *total = part[1023];
}

__global__ __launch_bounds__(DN_BLOCK) void k_rows_rank(RankArgs a)
{
    __shared__ uint32_t wsum[DN_BLOCK / 64];
    const int64_t r0 = (int64_t)blockIdx.x * RK_ROWS + threadIdx.x * 16;
    uint8_t m[16];
    uint32_t c = 0;
    for (int k = 0; k < 16; k++) {
        m[k] = (r0 + k < a.P) ? a.mask[r0 + k] : 0;
        c += m[k] != 0;
    }
    // exclusive prefix of c over the workgroup
    uint32_t inc = c;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)inc, o);
        if (lane >= o) inc += v;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t base = a.block_sum[blockIdx.x];
    for (int w = 0; w < wave; w++) base += wsum[w];
    uint32_t run = base + inc - c;
    for (int k = 0; k < 16; k++)
        if (r0 + k < a.P) {
            a.rank[r0 + k] = (int32_t)run;
            run += m[k] != 0;
        }
}

// ---- gather: dst[rank[i]] = src[i] where mask[i]; PIECE = bytes per thread access -------------------
template <typename T>
__global__ __launch_bounds__(DN_BLOCK) void k_rows_gather(int64_t P, int pieces /* per row */, const uint8_t* __restrict__ mask,
                                                          const int32_t* __restrict__ rank, const T* __restrict__ src, T* __restrict__ dst)
{
    const int64_t e = (int64_t)blockIdx.x * DN_BLOCK + threadIdx.x;
    const int64_t row = e / pieces;
    if (row >= P || !mask[row]) return;
    const int piece = (int)(e - row * pieces);
    dst[(int64_t)rank[row] * pieces + piece] = src[e];
}

}  // namespace

extern "C" int gft_densify_stats(void* hip_stream, int64_t P, const float* viewspace_grad, const float* pixels, const int32_t* radii,
                                 const uint8_t* update_filter, const uint8_t* apply_mask, float* xyz_gradient_accum, float* denom,
                                 float* max_radii2D)
{
    if (P < 0) return gft_fail("gft_densify_stats: P < 0");
    if (P == 0) return 0;
    if (!update_filter) return gft_fail("gft_densify_stats: update_filter is NULL");
    if ((xyz_gradient_accum && !viewspace_grad) || ((xyz_gradient_accum || denom) && !pixels) || (max_radii2D && !radii))
        return gft_fail("gft_densify_stats: an input of a requested statistic is NULL");
    hipLaunchKernelGGL(k_densify_stats, dim3((unsigned)((P + DN_BLOCK - 1) / DN_BLOCK)), dim3(DN_BLOCK), 0, (hipStream_t)hip_stream, P,
                       viewspace_grad, pixels, radii, update_filter, apply_mask, xyz_gradient_accum, denom, max_radii2D);
    GFT_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" size_t gft_rows_rank_scratch_bytes(int64_t P)
{
    const int64_t blocks = P <= 0 ? 0 : (P + RK_ROWS - 1) / RK_ROWS;
    return (size_t)(blocks + 64) * sizeof(uint32_t);
}

// the three launches of a ranking; the count stays on the device (*total_dev; NULL: behind the block sums in scratch)
static int rows_rank_launch(hipStream_t s, int64_t P, const uint8_t* mask, int32_t* rank, void* scratch, uint32_t* total_dev,
                            uint32_t** total_at)
{
    const int blocks = (int)((P + RK_ROWS - 1) / RK_ROWS);
    RankArgs a;
    a.P = P; a.mask = mask; a.rank = rank;
    a.block_sum = (uint32_t*)scratch;
    a.total = total_dev ? total_dev : a.block_sum + blocks;
    hipLaunchKernelGGL(k_rows_count, dim3(blocks), dim3(DN_BLOCK), 0, s, a);
    hipLaunchKernelGGL(k_rows_scan, dim3(1), dim3(1024), 0, s, a.block_sum, blocks, a.total);
    hipLaunchKernelGGL(k_rows_rank, dim3(blocks), dim3(DN_BLOCK), 0, s, a);
    GFT_CHECK_HIP(hipGetLastError());
    if (total_at) *total_at = a.total;
    return 0;
}

extern "C" int gft_rows_rank(void* hip_stream, int64_t P, const uint8_t* mask, int32_t* rank, void* scratch, int64_t* count)
{
    if (!count) return gft_fail("gft_rows_rank: count is NULL");
    *count = 0;
    if (P < 0 || P > 0x7fffffffll) return gft_fail("gft_rows_rank: bad row count");
    if (P == 0) return 0;
    if (!mask || !rank || !scratch) return gft_fail("gft_rows_rank: NULL argument");
    hipStream_t s = (hipStream_t)hip_stream;
    uint32_t* total = nullptr;
    if (rows_rank_launch(s, P, mask, rank, scratch, nullptr, &total)) return 1;
    uint32_t host = 0;
    GFT_CHECK_HIP(hipMemcpyAsync(&host, total, sizeof(host), hipMemcpyDeviceToHost, s));
    GFT_CHECK_HIP(hipStreamSynchronize(s));
    *count = host;
    return 0;
}

extern "C" int gft_rows_rank_dev(void* hip_stream, int64_t P, const uint8_t* mask, int32_t* rank, void* scratch, uint32_t* count_dev)
{
    if (!count_dev) return gft_fail("gft_rows_rank_dev: count_dev is NULL");
    if (P <= 0 || P > 0x7fffffffll) return gft_fail("gft_rows_rank_dev: bad row count");
    if (!mask || !rank || !scratch) return gft_fail("gft_rows_rank_dev: NULL argument");
    return rows_rank_launch((hipStream_t)hip_stream, P, mask, rank, scratch, count_dev, nullptr);
}

// mask[i] = 1 if row i of `a` ([P, ra] floats) or of `b` ([P, rb] floats) holds a value that is not zero (NaN counts,
// -0 does not): four lanes per row, 16-byte pieces when a row is a whole number of them
namespace {
__device__ __forceinline__ bool row_part_nonzero(const float* __restrict__ t, int64_t row, int r, int q)
{
    bool any = false;
    if (!t) return false;
    const float* p = t + row * r;
    if ((r & 3) == 0 && ((uintptr_t)t & 15) == 0) {
        for (int v = q; v < (r >> 2); v += 4) {
            const float4 x = reinterpret_cast<const float4*>(p)[v];
            any |= (x.x != 0.f) | (x.y != 0.f) | (x.z != 0.f) | (x.w != 0.f);
        }
    } else {
        for (int c = q; c < r; c += 4) any |= p[c] != 0.f;
    }
    return any;
}

__global__ __launch_bounds__(256) void k_rows_any_nonzero(int64_t P, int ra, const float* __restrict__ a, int rb,
                                                          const float* __restrict__ b, uint8_t* __restrict__ mask)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = t >> 2;
    const int q = (int)(t & 3);
    bool any = false;
    if (row < P) any = row_part_nonzero(a, row, ra, q) | row_part_nonzero(b, row, rb, q);
    int v = any ? 1 : 0;
    v |= __shfl_xor(v, 1);
    v |= __shfl_xor(v, 2);
    if (row < P && q == 0) mask[row] = (uint8_t)v;
}
}  // namespace

extern "C" int gft_rows_any_nonzero(void* hip_stream, int64_t P, int32_t row_floats_a, const float* a, int32_t row_floats_b,
                                    const float* b, uint8_t* mask)
{
    if (P < 0 || row_floats_a < 0 || row_floats_b < 0) return gft_fail("gft_rows_any_nonzero: negative size");
    if (P == 0) return 0;
    if (!mask) return gft_fail("gft_rows_any_nonzero: mask is NULL");
    if (P > ((int64_t)1 << 40)) return gft_fail("gft_rows_any_nonzero: P too large");
    hipLaunchKernelGGL(k_rows_any_nonzero, dim3((unsigned)((4 * P + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, P,
                       (int)row_floats_a, a, (int)row_floats_b, b, mask);
    GFT_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int gft_rows_gather(void* hip_stream, int64_t P, const uint8_t* mask, const int32_t* rank, const void* src, void* dst,
                               int64_t row_bytes)
{
    if (P < 0 || row_bytes < 0) return gft_fail("gft_rows_gather: negative size");
    if (P == 0 || row_bytes == 0) return 0;
    if (row_bytes % 4) return gft_fail("gft_rows_gather: row_bytes must be a multiple of 4");
    if (!mask || !rank || !src || !dst) return gft_fail("gft_rows_gather: NULL argument");
    if (((uintptr_t)src | (uintptr_t)dst) & 3) return gft_fail("gft_rows_gather: pointers must be 4-byte aligned");
    hipStream_t s = (hipStream_t)hip_stream;
    const bool wide = row_bytes % 16 == 0 && ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0);
    const int64_t pieces = row_bytes / (wide ? 16 : 4);
    const int64_t total = P * pieces;
    if (pieces > 0x7fffffffll || (total + DN_BLOCK - 1) / DN_BLOCK > 0x7fffffffll) return gft_fail("gft_rows_gather: too large");
    const dim3 grid((unsigned)((total + DN_BLOCK - 1) / DN_BLOCK));
    if (wide) hipLaunchKernelGGL(k_rows_gather<uint4>, grid, dim3(DN_BLOCK), 0, s, P, (int)pieces, mask, rank, (const uint4*)src, (uint4*)dst);
    else hipLaunchKernelGGL(k_rows_gather<uint32_t>, grid, dim3(DN_BLOCK), 0, s, P, (int)pieces, mask, rank, (const uint32_t*)src, (uint32_t*)dst);
    GFT_CHECK_HIP(hipGetLastError());
    return 0;
}
