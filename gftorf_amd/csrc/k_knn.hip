// k_knn.hip -- mean squared distance to the 3 nearest neighbours of every point (gfx950).
//
// Drop-in arithmetic of the reference's second native extension, simple_knn._C.distCUDA2
// (submodules/simple-knn/simple_knn.cu:185-220, spatial.cu:14-25), which
// scene/gaussian_model.py:194-199 calls once to initialise the Gaussian scales: exact 3-NN
// (self excluded by index, duplicates count with distance 0), result (d0 + d1 + d2) / 3.
// The reference groups points into boxes of 1024 along a Morton curve (CUB radix sort) and
// prunes boxes by their distance to the query; any grouping gives the same result, only the
// amount of pruning changes.  MI355X form, everything on the device, no host round trip:
//
//   k_knn_bounds  : bounding box (block partials, last workgroup reduces; like the reference the
//                   reduction starts from 0, so the box contains the origin, simple_knn.cu:190-198)
//   k_knn_codes   : 30-bit Morton code per point
//   k_radix_*     : LSD radix sort of (code, index) pairs, 8-bit digits, 4 passes: per-block digit
//                   histograms, one scan, stable scatter (ranks from wave ballots + LDS row offsets)
//   k_knn_boxes   : min/max of every run of 1024 points along the curve
//   k_knn_search  : one workgroup per 256 consecutive (spatially close) points: the box table is
//                   walked once per workgroup, a box that any lane still needs is staged in LDS
//                   (12 KB) and scanned with broadcast reads
#include "gft_internal.h"
#include "gftorf_knn.h"

#include <cfloat>

namespace {

#define KNN_BOX 1024
#define KNN_BLOCK 256

struct Bounds { float mn[3], mx[3]; };

__global__ __launch_bounds__(KNN_BLOCK) void k_knn_bounds(int P, const float* __restrict__ pts, Bounds* partial,
                                                          uint32_t* ticket, Bounds* __restrict__ out)
{
    __shared__ float s[6][KNN_BLOCK / 64];
    __shared__ uint32_t s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};               // the reference's reduction starts from (0,0,0)
    for (int i = blockIdx.x * KNN_BLOCK + tid; i < P; i += gridDim.x * KNN_BLOCK) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float x = pts[3 * (size_t)i + c];
            v[c] = fminf(v[c], x);
            v[3 + c] = fmaxf(v[3 + c], x);
        }
    }
#pragma unroll
    for (int c = 0; c < 6; c++) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const float o = __shfl_xor(v[c], d, 64);
            v[c] = c < 3 ? fminf(v[c], o) : fmaxf(v[c], o);
        }
        if (lane == 0) s[c][wave] = v[c];
    }
    __syncthreads();
    if (tid == 0) {
        Bounds b;
        for (int c = 0; c < 3; c++) {
            b.mn[c] = s[c][0]; b.mx[c] = s[3 + c][0];
            for (int w = 1; w < KNN_BLOCK / 64; w++) { b.mn[c] = fminf(b.mn[c], s[c][w]); b.mx[c] = fmaxf(b.mx[c], s[3 + c][w]); }
        }
        for (int c = 0; c < 3; c++) {
            __hip_atomic_store(&partial[blockIdx.x].mn[c], b.mn[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&partial[blockIdx.x].mx[c], b.mx[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last || tid != 0) return;
    Bounds b;
    for (int c = 0; c < 3; c++) { b.mn[c] = 0.f; b.mx[c] = 0.f; }
    for (uint32_t k = 0; k < gridDim.x; k++)
        for (int c = 0; c < 3; c++) {
            b.mn[c] = fminf(b.mn[c], __hip_atomic_load(&partial[k].mn[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            b.mx[c] = fmaxf(b.mx[c], __hip_atomic_load(&partial[k].mx[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
    *out = b;
    *ticket = 0;
}

__device__ __forceinline__ uint32_t spread10(uint32_t x)       // simple_knn.cu:45-52
{
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}

__global__ __launch_bounds__(KNN_BLOCK) void k_knn_codes(int P, const float* __restrict__ pts,
                                                         const Bounds* __restrict__ bounds, uint32_t* __restrict__ codes,
                                                         uint32_t* __restrict__ index)
{
    const int i = blockIdx.x * KNN_BLOCK + threadIdx.x;
    if (i >= P) return;
    const Bounds b = *bounds;
    uint32_t q[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        // simple_knn.cu:56-58: ((x - min) / (max - min)) * 1023, truncated; a flat axis gives 0/0 there,
        // any code is as good here
        const float ext = b.mx[c] - b.mn[c];
        const float t = ext > 0.f ? ((pts[3 * (size_t)i + c] - b.mn[c]) / ext) * 1023.0f : 0.f;
        q[c] = (uint32_t)fminf(fmaxf(t, 0.f), 1023.0f);
    }
    codes[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
    index[i] = (uint32_t)i;
}

// ---- LSD radix sort of (key, value) pairs, 8-bit digits ------------------------------------
#define RS_THREADS 256
#define RS_ITEMS 8
#define RS_CHUNK (RS_THREADS * RS_ITEMS)      // 2048 pairs per workgroup
#define RS_ROWS (RS_CHUNK / 64)               // 32 rows of 64 consecutive pairs

// hist[digit * nblocks + block]
__global__ __launch_bounds__(RS_THREADS) void k_radix_hist(int n, const uint32_t* __restrict__ keys, int shift,
                                                           uint32_t* __restrict__ hist)
{
    __shared__ uint32_t s_h[256];
    const int tid = threadIdx.x;
    s_h[tid] = 0;
    __syncthreads();
    const int base = blockIdx.x * RS_CHUNK;
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        const int i = base + k * RS_THREADS + tid;
        if (i < n) atomicAdd(&s_h[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)tid * gridDim.x + blockIdx.x] = s_h[tid];
}

// exclusive scan of `count` words in place by one workgroup
__global__ __launch_bounds__(1024) void k_radix_scan(uint32_t* data, int count)
{
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < count; base += 1024) {
        const int i = base + tid;
        const uint32_t v = i < count ? data[i] : 0u;
        uint32_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        uint32_t off = s_carry;
        for (int w = 0; w < wave; w++) off += s_w[w];
        if (i < count) data[i] = off + x - v;
        __syncthreads();
        if (tid == 1023) s_carry = off + x;
        __syncthreads();
    }
}

// Stable scatter: row r of a workgroup = 64 consecutive pairs; the rank of a pair among the pairs of
// its row with the same digit comes from 8 ballots, the rows before it from an LDS table.
__global__ __launch_bounds__(RS_THREADS) void k_radix_scatter(int n, const uint32_t* __restrict__ keys,
                                                              const uint32_t* __restrict__ vals, int shift,
                                                              const uint32_t* __restrict__ hist,
                                                              uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out)
{
    __shared__ uint32_t s_row[RS_ROWS][256];          // pairs of digit d in row r, then offsets
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < RS_ROWS * 256; k += RS_THREADS) (&s_row[0][0])[k] = 0;
    __syncthreads();
    const int base = blockIdx.x * RS_CHUNK;
    uint32_t key[RS_ITEMS], val[RS_ITEMS], rank[RS_ITEMS];
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        const int row = k * (RS_THREADS / 64) + wave;              // rows in pair order
        const int i = base + row * 64 + lane;
        const bool in = i < n;
        key[k] = in ? keys[i] : 0xffffffffu;
        val[k] = in ? vals[i] : 0u;
        const uint32_t d = in ? (key[k] >> shift) & 255u : 256u;
        unsigned long long same = __builtin_amdgcn_ballot_w64(in);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(bit);
            same &= bit ? m : ~m;
        }
        rank[k] = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        if (in && rank[k] == 0) s_row[row][d] = (uint32_t)__popcll(same);
    }
    __syncthreads();
    {
        // digit `tid`: running offset over the rows, starting at this workgroup's slot of the scan
        uint32_t off = hist[(size_t)tid * gridDim.x + blockIdx.x];
        for (int r = 0; r < RS_ROWS; r++) {
            const uint32_t c = s_row[r][tid];
            s_row[r][tid] = off;
            off += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RS_ITEMS; k++) {
        const int row = k * (RS_THREADS / 64) + wave;
        const int i = base + row * 64 + lane;
        if (i < n) {
            const uint32_t pos = s_row[row][(key[k] >> shift) & 255u] + rank[k];
            keys_out[pos] = key[k];
            vals_out[pos] = val[k];
        }
    }
}

// sorts (keys, vals) by the low `bits` bits of the key; result in (keys, vals); (tk, tv) temporaries
hipError_t radix_sort_pairs(hipStream_t s, int n, uint32_t* keys, uint32_t* vals, uint32_t* tk, uint32_t* tv,
                            uint32_t* hist, int bits)
{
    const int nblocks = (n + RS_CHUNK - 1) / RS_CHUNK;
    uint32_t *ki = keys, *vi = vals, *ko = tk, *vo = tv;
    int passes = 0;
    for (int shift = 0; shift < bits; shift += 8, passes++) {
        hipLaunchKernelGGL(k_radix_hist, dim3(nblocks), dim3(RS_THREADS), 0, s, n, ki, shift, hist);
        hipLaunchKernelGGL(k_radix_scan, dim3(1), dim3(1024), 0, s, hist, 256 * nblocks);
        hipLaunchKernelGGL(k_radix_scatter, dim3(nblocks), dim3(RS_THREADS), 0, s, n, ki, vi, shift, hist, ko, vo);
        uint32_t* t = ki; ki = ko; ko = t;
        t = vi; vi = vo; vo = t;
    }
    if (passes & 1) {       // odd number of passes: the result sits in the temporaries
        hipError_t e = hipMemcpyAsync(keys, ki, (size_t)n * 4, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return e;
        e = hipMemcpyAsync(vals, vi, (size_t)n * 4, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return e;
    }
    return hipGetLastError();
}

// placed coordinates (so the search reads them contiguously) and box bounds
__global__ __launch_bounds__(KNN_BOX) void k_knn_boxes(int P, const float* __restrict__ pts, const uint32_t* __restrict__ order,
                                                       float* __restrict__ placed, Bounds* __restrict__ boxes)
{
    __shared__ float s[6][KNN_BOX / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = blockIdx.x * KNN_BOX + tid;
    float v[6] = {FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (j < P) {
        const uint32_t i = order[j];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float x = pts[3 * (size_t)i + c];
            placed[3 * (size_t)j + c] = x;
            v[c] = x; v[3 + c] = x;
        }
    }
#pragma unroll
    for (int c = 0; c < 6; c++) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const float o = __shfl_xor(v[c], d, 64);
            v[c] = c < 3 ? fminf(v[c], o) : fmaxf(v[c], o);
        }
        if (lane == 0) s[c][wave] = v[c];
    }
    __syncthreads();
    if (tid == 0) {
        Bounds b;
        for (int c = 0; c < 3; c++) {
            b.mn[c] = s[c][0]; b.mx[c] = s[3 + c][0];
            for (int w = 1; w < KNN_BOX / 64; w++) { b.mn[c] = fminf(b.mn[c], s[c][w]); b.mx[c] = fmaxf(b.mx[c], s[3 + c][w]); }
        }
        boxes[blockIdx.x] = b;
    }
}

// simple_knn.cu:119-129
__device__ __forceinline__ float box_point_dist2(const Bounds& b, float px, float py, float pz)
{
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (px < b.mn[0] || px > b.mx[0]) dx = fminf(fabsf(px - b.mn[0]), fabsf(px - b.mx[0]));
    if (py < b.mn[1] || py > b.mx[1]) dy = fminf(fabsf(py - b.mn[1]), fabsf(py - b.mx[1]));
    if (pz < b.mn[2] || pz > b.mx[2]) dz = fminf(fabsf(pz - b.mn[2]), fabsf(pz - b.mx[2]));
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

// simple_knn.cu:131-146 (insertion into the three smallest); the squared distance is the
// expression d.x*d.x + d.y*d.y + d.z*d.z as nvcc contracts it by default
__device__ __forceinline__ void keep3(float px, float py, float pz, float qx, float qy, float qz, float* best)
{
    const float dx = qx - px, dy = qy - py, dz = qz - pz;
    float dist = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
#pragma unroll
    for (int j = 0; j < 3; j++) {
        if (best[j] > dist) {
            const float t = best[j];
            best[j] = dist;
            dist = t;
        }
    }
}

__global__ __launch_bounds__(KNN_BLOCK) void k_knn_search(int P, const float* __restrict__ placed,
                                                          const uint32_t* __restrict__ order,
                                                          const Bounds* __restrict__ boxes, float* __restrict__ out)
{
    __shared__ float s_pts[KNN_BOX * 3];
    __shared__ uint32_t s_any;
    const int tid = threadIdx.x;
    const int j = blockIdx.x * KNN_BLOCK + tid;
    const bool in = j < P;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (in) { px = placed[3 * (size_t)j]; py = placed[3 * (size_t)j + 1]; pz = placed[3 * (size_t)j + 2]; }
    float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    // first bound from the neighbours along the placement order (simple_knn.cu:156-161)
    if (in) {
        const int lo = max(0, j - 3), hi = min(P - 1, j + 3);
        for (int i = lo; i <= hi; i++)
            if (i != j) keep3(px, py, pz, placed[3 * (size_t)i], placed[3 * (size_t)i + 1], placed[3 * (size_t)i + 2], best);
    }
    const float reject = best[2];
    best[0] = best[1] = best[2] = FLT_MAX;
    const int nboxes = (P + KNN_BOX - 1) / KNN_BOX;
    for (int b = 0; b < nboxes; b++) {
        bool want = false;
        if (in) {
            const float d = box_point_dist2(boxes[b], px, py, pz);
            want = !(d > reject || d > best[2]);                 // simple_knn.cu:171-173
        }
        if (tid == 0) s_any = 0;
        __syncthreads();
        if (want) s_any = 1;                                     // benign race: every writer stores 1
        __syncthreads();
        if (s_any) {                                             // uniform per workgroup
            const int first = b * KNN_BOX, cnt = min(KNN_BOX, P - first);
            for (int k = tid; k < cnt * 3; k += KNN_BLOCK) s_pts[k] = placed[3 * (size_t)first + k];
            __syncthreads();
            if (want) {
                for (int k = 0; k < cnt; k++)
                    if (first + k != j) keep3(px, py, pz, s_pts[3 * k], s_pts[3 * k + 1], s_pts[3 * k + 2], best);
            }
        }
        __syncthreads();
    }
    if (in) out[order[j]] = (best[0] + best[1] + best[2]) / 3.0f;
}

}  // namespace

extern "C" size_t gft_knn_scratch_bytes(int32_t P)
{
    const size_t p = (size_t)(P > 0 ? P : 0);
    const size_t nboxes = (p + KNN_BOX - 1) / KNN_BOX;
    const size_t nrs = (p + RS_CHUNK - 1) / RS_CHUNK;
    // codes, order, two sort temporaries u32[P] | placed f32[3P] | boxes | radix histograms |
    // bounds partials (256) + result + ticket
    return p * 4 * 4 + p * 12 + nboxes * sizeof(Bounds) + 256 * nrs * 4 + 258 * sizeof(Bounds) + 2048;
}

extern "C" int gft_knn_mean_dist2(void* hip_stream, int32_t P, const float* points, float* mean_dist2, void* scratch)
{
    if (P < 0) return gft_fail("gft_knn_mean_dist2: P < 0");
    if (P == 0) return 0;
    if (!points || !mean_dist2 || !scratch) return gft_fail("gft_knn_mean_dist2: NULL pointer");
    hipStream_t s = (hipStream_t)hip_stream;
    const size_t p = (size_t)P;
    const int nboxes = (P + KNN_BOX - 1) / KNN_BOX;
    const size_t nrs = (p + RS_CHUNK - 1) / RS_CHUNK;
    char* b = (char*)scratch;
    uint32_t* codes = (uint32_t*)b;               b += p * 4;
    uint32_t* order = (uint32_t*)b;               b += p * 4;
    uint32_t* tk = (uint32_t*)b;                  b += p * 4;
    uint32_t* tv = (uint32_t*)b;                  b += p * 4;
    float* placed = (float*)b;                    b += p * 12;
    b = (char*)(((uintptr_t)b + 255) & ~(uintptr_t)255);
    Bounds* boxes = (Bounds*)b;                   b += (size_t)nboxes * sizeof(Bounds);
    uint32_t* hist = (uint32_t*)b;                b += 256 * nrs * 4;
    Bounds* partial = (Bounds*)b;                 b += 256 * sizeof(Bounds);
    Bounds* bounds = (Bounds*)b;                  b += sizeof(Bounds);
    uint32_t* ticket = (uint32_t*)b;
    GFT_CHECK_HIP(gft_zero_async(ticket, 4, s));
    const int blocks = (P + KNN_BLOCK - 1) / KNN_BLOCK;
    hipLaunchKernelGGL(k_knn_bounds, dim3(blocks < 256 ? blocks : 256), dim3(KNN_BLOCK), 0, s, P, points, partial, ticket, bounds);
    hipLaunchKernelGGL(k_knn_codes, dim3(blocks), dim3(KNN_BLOCK), 0, s, P, points, bounds, codes, order);
    hipError_t e = radix_sort_pairs(s, P, codes, order, tk, tv, hist, 30);
    if (e != hipSuccess) return gft_fail("gft_knn_mean_dist2: sort: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k_knn_boxes, dim3(nboxes), dim3(KNN_BOX), 0, s, P, points, order, placed, boxes);
    hipLaunchKernelGGL(k_knn_search, dim3(blocks), dim3(KNN_BLOCK), 0, s, P, placed, order, boxes, mean_dist2);
    e = hipGetLastError();
    if (e != hipSuccess) return gft_fail("gft_knn_mean_dist2: %s", hipGetErrorString(e));
    return 0;
}
