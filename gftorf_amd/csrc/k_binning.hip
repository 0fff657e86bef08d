// k_binning.hip -- whole-frame instance binning (gfx950): per-tile counting, scatter of
// (depth, id) keys into tile segments, per-tile sort in LDS.  The structure of the reference
// (every instance counted, keyed and sorted); the production forward bins by tile pull instead
// (k_pull.hip), this path is its bit-for-bit comparator (gft_set_binning_mode(0), GFT_LAZY_BIN=0)
// and the fallback for tile grids the supertile tables do not fit.
//
// The reference duplicates every Gaussian into (tile << 32 | depth bits) keys and runs
// one global 64-bit radix sort over all R instances (RAST/cuda_rasterizer/
// rasterizer_impl.cu:72-140,307-348: ~6 passes x 24 B x R of HBM traffic).  The
// result it needs is only: for every tile, the ids of the Gaussians whose rectangle
// covers it, ascending by (depth bits, id) -- that is what a stable sort of keys
// emitted in id order produces.  MI355X form (integer work, HBM-bound, no global sort):
//
//   k_tile_count   : LDS histogram of tile hits per 4096-Gaussian block, one global
//                    atomic per (block, tile)                           [8 B/Gaussian];
//                    the last workgroup to finish scans the counters:
//                    ranges[tile] = [first,last), R, longest list -> ctrl + host mailbox
//   k_tile_scatter : each block reserves one chunk per tile with a single global atomic,
//                    groups its (depth bits << 32 | id) keys by tile in LDS and writes every
//                    chunk with one instruction                          [8 B/instance]
//   k_tile_front   : one workgroup per tile: the nearest ~940 keys (256-bin depth histogram) are
//                    sorted in LDS into the head of the id list, the other ids follow unsorted;
//                    k_tile_tail sorts the tail of a tile whose quadrant walked past the head
//   k_tile_sort_*  : GFT_LAZY_SORT=0: whole lists (bitonic networks in LDS)
//
// The order inside a tile segment after the scatter is arbitrary (atomic cursors); the
// sort makes the final lists deterministic and bit-identical to the reference's.
#include "gft_internal.h"
#include "gft_sort.h"

#include <cstdlib>

namespace {

// BIN_THREADS / BIN_ITEMS / BIN_CHUNK: gft_internal.h (the geom layout depends on them)
#define BIN_LDS_MAX_TILES 16384            // LDS histogram limit (2 x 64 KB in the scatter)
#define BIN_STAGE_LDS_BYTES (156 * 1024)   // dynamic LDS of the staged scatter (160 KB per CU minus static)

// Every workgroup walks the tile table from a different start so that the ~P/4096 workgroups
// do not queue up on the same counter at the same time.
__device__ __forceinline__ int rotated_tile(int i, int T)
{
    const int rot = (int)(((uint64_t)blockIdx.x * (uint64_t)T) / gridDim.x);
    const int t = i + rot;
    return t >= T ? t - T : t;
}

// Exclusive scan of tile_cnt by one workgroup (the last one of k_tile_count): ranges (offset by `base`), zeroed
// cursors; returns the total and the longest list to thread 0.
__device__ void tile_scan_block(int T, const uint32_t* tile_cnt, uint2* __restrict__ ranges, uint32_t* __restrict__ cursor,
                                uint32_t base, uint32_t& total_out, uint32_t& max_out)
{
    constexpr int NW = BIN_THREADS / 64;
    __shared__ uint32_t wtot[NW];
    __shared__ uint32_t wmax[NW];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t vmax = 0;
    for (int b0 = 0; b0 < T; b0 += BIN_THREADS) {
        const int i = b0 + threadIdx.x;
        // counters were accumulated by atomics of other workgroups (other XCDs): device-scope load
        const uint32_t v = (i < T) ? __hip_atomic_load(&tile_cnt[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        vmax = max(vmax, v);
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
        if (i < T) {
            const uint32_t first = base + carry + woff + x - v;
            ranges[i] = v ? make_uint2(first, first + v) : make_uint2(0u, 0u);   // untouched tiles: (0,0) like the reference
            cursor[i] = 0;
        }
        __syncthreads();
        if (threadIdx.x == BIN_THREADS - 1) carry_s = carry + woff + x;
        __syncthreads();
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) vmax = max(vmax, (uint32_t)__shfl_xor((int)vmax, d, 64));
    if (lane == 0) wmax[wave] = vmax;
    __syncthreads();
    uint32_t m = 0;
    if (threadIdx.x == 0)
        for (int w = 0; w < NW; w++) m = max(m, wmax[w]);
    total_out = carry_s;
    max_out = m;
}

struct CountArgs {
    int P, gx, T;
    const ushort4* __restrict__ rect;
    uint32_t* tile_cnt;
    uint2* __restrict__ ranges;
    uint32_t* __restrict__ cursor;
    uint32_t* ctrl;
    uint32_t* mail; uint32_t seq;
    uint32_t post;                  // `mail` is a status block of gft_forward_enqueue: its binning_instances + 1 (0: a mailbox slot)
    uint16_t* __restrict__ blockhist;
};

template <bool USE_LDS>
__global__ __launch_bounds__(BIN_THREADS) void k_tile_count(CountArgs a)
{
    extern __shared__ uint32_t hist[];
    __shared__ uint32_t s_last;
    const int tid = threadIdx.x;
    const int T = a.T, gx = a.gx;
    if (USE_LDS) {
        for (int i = tid; i < T; i += BIN_THREADS) hist[i] = 0;
    }
    __syncthreads();
    const int base = blockIdx.x * BIN_CHUNK;
    // four Gaussians per trip, their rectangle loads issued together
    ushort4 r4[BIN_ITEMS];
#pragma unroll
    for (int u = 0; u < BIN_ITEMS; u++) {
        const int idx = base + u * BIN_THREADS + tid;
        r4[u] = idx < a.P ? a.rect[idx] : make_ushort4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < BIN_ITEMS; u++) {
        const ushort4 r = r4[u];
        for (int y = r.y; y < r.w; y++)
            for (int x = r.x; x < r.z; x++) {
                const int t = y * gx + x;
                if (USE_LDS) atomicAdd(&hist[t], 1u);
                else atomicAdd(&a.tile_cnt[t], 1u);
            }
    }
    __syncthreads();
    if (USE_LDS) {
        for (int i = tid; i < T; i += BIN_THREADS) {
            const int t = rotated_tile(i, T);
            const uint32_t h = hist[t];
            if (h) atomicAdd(&a.tile_cnt[t], h);
            // kept for the scatter, which would otherwise walk the rectangles a second time to count
            if (a.blockhist) a.blockhist[(size_t)blockIdx.x * GFT_BLOCKHIST_TILES + t] = (uint16_t)h;
        }
    }
    // The workgroup that draws the last ticket scans.  Every counter update above is a
    // device-scope atomic, complete once vmcnt drains, and the scan reads the counters with
    // device-scope loads: no cache write-back / invalidate (__threadfence would flush the
    // L2 lines the preprocess kernel just wrote, ~50 us) is needed for that hand-over.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(&a.ctrl[GFT_CTRL_DONE], 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    uint32_t total = 0, longest = 0;
    tile_scan_block(T, a.tile_cnt, a.ranges, a.cursor, 0u, total, longest);
    if (tid == 0) {
        a.ctrl[GFT_CTRL_TOTAL] = total;
        a.ctrl[GFT_CTRL_MAXCNT] = longest;
        if (a.mail) {
            a.mail[GFT_CTRL_TOTAL] = total;          // (GFT_CTRL_FLAGS of the slot belongs to the preprocess kernel)
            a.mail[GFT_CTRL_MAXCNT] = longest;
            gft_status_sticky(a.mail, a.post, total);
            __hip_atomic_store(&a.mail[GFT_CTRL_SEQ], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// MODE 0: global cursors only (tile table too large for LDS)
// MODE 1: LDS counters, one reserved chunk per (workgroup, tile), keys written straight to HBM
// MODE 2: as 1, but the workgroup's keys are first grouped by tile in LDS (`stage_cap` keys)
//         and every chunk is then written by one instruction.  With direct 8-byte writes a
//         chunk's line is touched a dozen times over the workgroup's lifetime and the ~30 MB
//         key array does not stay in the 4 MB L2s: 100 MB of write-backs for 29 MB of keys
//         (rocprofv3 WRITE_SIZE).  A workgroup with more instances than `stage_cap` writes
//         directly (MODE 1 behaviour).
struct ScatterArgs {
    int P, gx, T;
    const ushort4* __restrict__ rect;
    const float* __restrict__ depth;
    const uint2* __restrict__ ranges;
    uint32_t* __restrict__ cursor;
    uint64_t* __restrict__ keys;
    const uint32_t* __restrict__ ctrl;
    uint32_t cap, stage_cap;
    const uint16_t* __restrict__ blockhist;
};

template <int MODE>
__global__ __launch_bounds__(BIN_THREADS) void k_tile_scatter(ScatterArgs a)
{
    extern __shared__ uint32_t sh[];
    if (a.ctrl[GFT_CTRL_TOTAL] > a.cap) return;      // binning buffer too small: the host re-runs stage 2
    const int T = a.T, gx = a.gx, P = a.P;
    uint32_t* cnt = sh;          // [T] instances of this block per tile, then running slot
    uint32_t* first = sh + T;    // [T] global position of this block's chunk in the tile segment
    uint32_t* lstart = sh + 2 * T;                                   // [T] MODE 2: chunk start in the LDS stage
    uint64_t* stage = reinterpret_cast<uint64_t*>(sh + 3 * T + (T & 1));   // MODE 2: stage_cap keys, 8-B aligned
    __shared__ uint32_t s_wave_tot[BIN_THREADS / 64];
    __shared__ uint32_t s_block_tot;
    const int tid = threadIdx.x;
    const int base = blockIdx.x * BIN_CHUNK;
    bool staged = false;
    if (MODE >= 1) {
        if (a.blockhist) {
            // the count kernel kept this workgroup's histogram
            for (int i = tid; i < T; i += BIN_THREADS) cnt[i] = a.blockhist[(size_t)blockIdx.x * GFT_BLOCKHIST_TILES + i];
        } else {
            for (int i = tid; i < T; i += BIN_THREADS) cnt[i] = 0;
            __syncthreads();
#pragma unroll 4
            for (int k = 0; k < BIN_ITEMS; k++) {
                const int idx = base + k * BIN_THREADS + tid;
                if (idx < P) {
                    const ushort4 r = a.rect[idx];
                    if (!(r.z > r.x && r.w > r.y)) continue;
                    for (int y = r.y; y < r.w; y++)
                        for (int x = r.x; x < r.z; x++) atomicAdd(&cnt[y * gx + x], 1u);
                }
            }
        }
        __syncthreads();
        if (MODE == 2) {
            // exclusive scan of cnt over the tile table: K consecutive tiles per thread
            const int K = (T + BIN_THREADS - 1) / BIN_THREADS;
            const int t0 = tid * K;
            uint32_t mine = 0;
            for (int k = 0; k < K; k++)
                if (t0 + k < T) mine += cnt[t0 + k];
            uint32_t x = mine;
            const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_up(x, d, 64);
                if (lane >= d) x += y;
            }
            if (lane == 63) s_wave_tot[wave] = x;
            __syncthreads();
            uint32_t woff = 0;
            for (int w = 0; w < wave; w++) woff += s_wave_tot[w];
            if (tid == BIN_THREADS - 1) s_block_tot = woff + x;
            uint32_t run = woff + x - mine;
            for (int k = 0; k < K; k++)
                if (t0 + k < T) { lstart[t0 + k] = run; run += cnt[t0 + k]; }
            __syncthreads();
            staged = s_block_tot <= a.stage_cap;
        }
        for (int i = tid; i < T; i += BIN_THREADS) {
            const int t = rotated_tile(i, T);
            const uint32_t c = cnt[t];
            if (c) {
                first[t] = a.ranges[t].x + atomicAdd(&a.cursor[t], c);
                cnt[t] = 0;
            }
        }
        __syncthreads();
    }
    for (int k0 = 0; k0 < BIN_ITEMS; k0 += 2) {
        ushort4 r2[2];
        uint32_t d2[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int idx = base + (k0 + u) * BIN_THREADS + tid;
            const bool in = idx < P;
            r2[u] = in ? a.rect[idx] : make_ushort4(0, 0, 0, 0);
            d2[u] = in ? __float_as_uint(a.depth[idx]) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const ushort4 r = r2[u];
            if (!(r.z > r.x && r.w > r.y)) continue;
            const uint32_t d = d2[u];
            const uint32_t idx = (uint32_t)(base + (k0 + u) * BIN_THREADS + tid);
            const uint64_t key = ((uint64_t)d << 32) | idx;
            for (int y = r.y; y < r.w; y++)
                for (int x = r.x; x < r.z; x++) {
                    const int t = y * gx + x;
                    if (MODE == 2 && staged) {
                        stage[lstart[t] + atomicAdd(&cnt[t], 1u)] = key;
                    } else {
                        uint32_t pos;
                        if (MODE >= 1) pos = first[t] + atomicAdd(&cnt[t], 1u);
                        else pos = a.ranges[t].x + atomicAdd(&a.cursor[t], 1u);
                        a.keys[pos] = key;
                    }
                }
        }
    }
    if (MODE == 2 && staged) {
        __syncthreads();
        // one wave per tile chunk: contiguous LDS run -> contiguous run of the tile segment
        const int lane = tid & 63;
        for (int t = tid >> 6; t < T; t += BIN_THREADS / 64) {
            const uint32_t c = cnt[t];
            if (c == 0) continue;
            const uint32_t src = lstart[t], dst = first[t];
            for (uint32_t i = lane; i < c; i += 64) a.keys[dst + i] = stage[src + i];
        }
    }
}

// Sorts keys[first, first + n), n <= 4096, into point_list[first, first + n) (ids only) through
// the 33.8 KB LDS buffer `sk`.  Called by all 256 threads of a workgroup with uniform arguments.
__device__ __forceinline__ void sort_range_small(uint32_t first, uint32_t n, const uint64_t* __restrict__ keys,
                                                 uint32_t* __restrict__ point_list, uint64_t* sk, int tid)
{
    if (n == 0) return;
    if (n == 1) {
        if (tid == 0) point_list[first] = (uint32_t)keys[first];
        return;
    }
    if (n <= 1024u) {
        const uint32_t npad = next_pow2(n);
        for (uint32_t i = tid; i < n; i += GFT_BLOCK) sk[i] = keys[first + i];
        __syncthreads();
        bitonic_ascending<GFT_BLOCK>(n, npad, tid, [&](uint32_t i) { return sk[i]; }, [&](uint32_t i, uint64_t v) { sk[i] = v; },
                                     [] { __syncthreads(); });
        for (uint32_t i = tid; i < n; i += GFT_BLOCK) point_list[first + i] = (uint32_t)sk[i];
        return;
    }
    const uint32_t npad = n <= 2048u ? 2048u : 4096u;
    for (uint32_t i = tid; i < npad; i += GFT_BLOCK) sk[sort_slot(i)] = i < n ? keys[first + i] : ~0ull;
    __syncthreads();
    if (npad == 2048u) bitonic_blocked<3, 8>(sk, tid);
    else bitonic_blocked<4, 8>(sk, tid);
    for (uint32_t i = tid; i < n; i += GFT_BLOCK) point_list[first + i] = (uint32_t)sk[sort_slot(i)];
}

// Sort class A: tile lists of up to 4096 keys, 33.8 KB of LDS.
__global__ __launch_bounds__(GFT_BLOCK) void k_tile_sort_small(const uint2* __restrict__ ranges,
                                                               const uint64_t* __restrict__ keys,
                                                               uint32_t* __restrict__ point_list,
                                                               const uint32_t* __restrict__ ctrl, uint32_t cap,
                                                               float4* __restrict__ clear, size_t clear_vec4)
{
    __shared__ uint64_t sk[SORT_SLOTS(4096)];
    if (ctrl[GFT_CTRL_TOTAL] > cap) return;
    // fire-and-forget zero fill of the backward's accumulator (see gft_api.hip enqueue_stage2):
    // the stores drain while this workgroup sorts in LDS
    if (clear) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (size_t i = (size_t)blockIdx.x * GFT_BLOCK + threadIdx.x; i < clear_vec4; i += (size_t)gridDim.x * GFT_BLOCK)
            clear[i] = z;
    }
    const uint2 r = ranges[blockIdx.x];
    const uint32_t n = r.y - r.x;
    if (n > 4096u) {
        // long list: sorted by k_tile_sort_big.  The id list is filled in scatter order here so that
        // it holds valid ids even when that kernel is only launched after a first render
        // (gft_forward with a wrong list-length guess).
        for (uint32_t i = threadIdx.x; i < n; i += GFT_BLOCK) point_list[r.x + i] = (uint32_t)keys[r.x + i];
        return;
    }
    sort_range_small(r.x, n, keys, point_list, sk, threadIdx.x);
}

// ---- lazy sort: sorted front + unsorted tail ----------------------------------------------
// A pixel stops consuming its tile list when its transmittance drops below 1e-4, so in a dense
// frame most of a long list is never read (1 M metric frame: 3031 entries per tile, 430 walked).
// Lists longer than FRONT_DIRECT are therefore split: the nearest ~FRONT_TARGET keys (every key
// <= a splitter picked from 256 sorted samples) are sorted into the head of the id list, the
// other ids follow unsorted.  The render kernel walks the head; a quadrant that reaches its end
// with unsaturated pixels raises a flag and saves its state, k_tile_tail then sorts the rest of
// that tile's list and the render kernel resumes those quadrants (same arithmetic sequence as a
// single pass, bit-identical results).
#define FRONT_DIRECT 1024u       // lists up to this length are sorted whole
#define FRONT_TARGET 940u        // wanted length of the sorted head of a longer list
#define FRONT_MAX 1024u          // capacity of the head sort (8 KB of LDS: many workgroups per CU)
#define FRONT_BINS 256
#define FRONT_UNROLL 4

// The head = every key whose depth falls into the first bins of a 256-bin histogram (linear in the
// depth value between the list's nearest and farthest key) up to the bin where the running count
// reaches FRONT_TARGET; if that bin overshoots FRONT_MAX the bins before it are taken.
__global__ __launch_bounds__(GFT_BLOCK) void k_tile_front(const uint2* __restrict__ ranges,
                                                          const uint64_t* __restrict__ keys,
                                                          uint32_t* __restrict__ point_list,
                                                          uint32_t* __restrict__ front_len, uint32_t* __restrict__ unit_flag,
                                                          const uint32_t* __restrict__ ctrl, uint32_t cap,
                                                          float4* __restrict__ clear, size_t clear_vec4)
{
    __shared__ uint64_t sk[FRONT_MAX];
    __shared__ uint32_t s_hist[FRONT_BINS];
    __shared__ uint32_t s_min[GFT_BLOCK / 64], s_max[GFT_BLOCK / 64];
    __shared__ uint32_t s_cut, s_kf, s_nf, s_nt;
    if (ctrl[GFT_CTRL_TOTAL] > cap) return;
    // fire-and-forget zero fill of the backward's accumulator (see gft_api.hip enqueue_stage2):
    // the stores drain while this workgroup sorts in LDS
    if (clear) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (size_t i = (size_t)blockIdx.x * GFT_BLOCK + threadIdx.x; i < clear_vec4; i += (size_t)gridDim.x * GFT_BLOCK)
            clear[i] = z;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const uint2 r = ranges[tile];
    const uint32_t n = r.y - r.x;
    if (tid < 4) unit_flag[4 * tile + tid] = 0;
    if (n <= FRONT_DIRECT) {
        if (n == 1u) {
            if (tid == 0) point_list[r.x] = (uint32_t)keys[r.x];
        } else if (n > 1u) {
            const uint32_t npad = next_pow2(n);
            for (uint32_t i = tid; i < npad; i += GFT_BLOCK) sk[i] = i < n ? keys[r.x + i] : ~0ull;
            __syncthreads();
            head_sort_and_store(sk, n, npad, tid, point_list + r.x);
        }
        if (tid == 0) front_len[tile] = n;
        return;
    }
    // pass 1: depth range (depth bits of a positive float order like the value)
    uint32_t dmin = 0xffffffffu, dmax = 0u;
    // (every pass keeps FRONT_UNROLL loads per lane in flight: the loops are latency-bound otherwise)
    for (uint32_t i0 = 0; i0 < n; i0 += FRONT_UNROLL * GFT_BLOCK) {
        uint64_t k4[FRONT_UNROLL];
#pragma unroll
        for (int u = 0; u < FRONT_UNROLL; u++) {
            const uint32_t i = i0 + u * GFT_BLOCK + tid;
            k4[u] = i < n ? keys[r.x + i] : keys[r.x];
        }
#pragma unroll
        for (int u = 0; u < FRONT_UNROLL; u++) {
            const uint32_t d = (uint32_t)(k4[u] >> 32);
            dmin = min(dmin, d);
            dmax = max(dmax, d);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        dmin = min(dmin, (uint32_t)__shfl_xor((int)dmin, o, 64));
        dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, o, 64));
    }
    if (lane == 0) { s_min[wave] = dmin; s_max[wave] = dmax; }
    s_hist[tid] = 0;                                         // FRONT_BINS == GFT_BLOCK
    __syncthreads();
    for (int w = 0; w < GFT_BLOCK / 64; w++) { dmin = min(dmin, s_min[w]); dmax = max(dmax, s_max[w]); }
    const float zmin = __uint_as_float(dmin);
    const float zspan = __uint_as_float(dmax) - zmin;
    const float scale = zspan > 0.f ? (float)FRONT_BINS / zspan : 0.f;
    auto bin_of = [&](uint64_t k) {
        const uint32_t b = (uint32_t)((__uint_as_float((uint32_t)(k >> 32)) - zmin) * scale);   // monotone in the depth
        return b < FRONT_BINS ? b : FRONT_BINS - 1u;
    };
    // pass 2: histogram
    for (uint32_t i0 = 0; i0 < n; i0 += FRONT_UNROLL * GFT_BLOCK) {
        uint64_t k4[FRONT_UNROLL];
#pragma unroll
        for (int u = 0; u < FRONT_UNROLL; u++) {
            const uint32_t i = i0 + u * GFT_BLOCK + tid;
            k4[u] = i < n ? keys[r.x + i] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < FRONT_UNROLL; u++)
            if (i0 + u * GFT_BLOCK + tid < n) atomicAdd(&s_hist[bin_of(k4[u])], 1u);
    }
    __syncthreads();
    if (wave == 0) {
        // inclusive scan of the 256 bins by one wave (4 bins per lane), then the cut
        uint32_t h[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { h[k] = s_hist[4 * lane + k]; sum += h[k]; }
        uint32_t x = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        uint32_t run = x - sum;
        // first bin whose inclusive count reaches the target
        uint32_t cut = 0xffffffffu, kf_at = 0, kf_before = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t before = run;
            run += h[k];
            if (cut == 0xffffffffu && run >= FRONT_TARGET) { cut = 4 * lane + k; kf_at = run; kf_before = before; }
        }
        // lowest lane that found one
        const unsigned long long found = __builtin_amdgcn_ballot_w64(cut != 0xffffffffu);
        const int src = found ? (int)__builtin_ctzll(found) : 0;
        cut = (uint32_t)__shfl((int)cut, src, 64);
        kf_at = (uint32_t)__shfl((int)kf_at, src, 64);
        kf_before = (uint32_t)__shfl((int)kf_before, src, 64);
        if (lane == 0) {
            // bins [0, cut] if they fit, else [0, cut): cut_excl = number of bins taken
            if (kf_at <= FRONT_MAX) { s_cut = cut + 1u; s_kf = kf_at; }
            else { s_cut = cut; s_kf = kf_before; }            // (one bin alone overshoots: shorter head)
            s_nf = 0; s_nt = 0;
        }
    }
    __syncthreads();
    const uint32_t nbins = s_cut, kf = s_kf;                  // kf <= FRONT_MAX by construction
    const uint32_t npad = next_pow2(kf < 2u ? 2u : kf);
    for (uint32_t i = tid + kf; i < npad; i += GFT_BLOCK) sk[i] = ~0ull;
    // pass 3: head keys to LDS, tail ids behind the head in the id list
    for (uint32_t i0 = 0; i0 < n; i0 += FRONT_UNROLL * GFT_BLOCK) {
        uint64_t k4[FRONT_UNROLL];
#pragma unroll
        for (int u = 0; u < FRONT_UNROLL; u++) {
            const uint32_t i = i0 + u * GFT_BLOCK + tid;
            k4[u] = i < n ? keys[r.x + i] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < FRONT_UNROLL; u++) {
            const bool in = i0 + u * GFT_BLOCK + tid < n;
            const uint64_t k = k4[u];
            const bool head = in && bin_of(k) < nbins;
            const bool tail = in && !head;
            // one LDS atomic per wave and destination
            const unsigned long long hm = __builtin_amdgcn_ballot_w64(head), tm = __builtin_amdgcn_ballot_w64(tail);
            uint32_t hb = 0, tb = 0;
            if (lane == 0) {
                if (hm) hb = atomicAdd(&s_nf, (uint32_t)__popcll(hm));
                if (tm) tb = atomicAdd(&s_nt, (uint32_t)__popcll(tm));
            }
            hb = (uint32_t)__builtin_amdgcn_readfirstlane((int)hb);
            tb = (uint32_t)__builtin_amdgcn_readfirstlane((int)tb);
            const unsigned long long lt = (1ull << lane) - 1ull;
            if (head) sk[hb + (uint32_t)__popcll(hm & lt)] = k;
            else if (tail) point_list[r.x + kf + tb + (uint32_t)__popcll(tm & lt)] = (uint32_t)k;
        }
    }
    __syncthreads();
    head_sort_and_store(sk, kf, npad, tid, point_list + r.x);
    if (tid == 0) front_len[tile] = kf;
}

// Long lists: 4097..16384 keys are sorted by 1024 threads with the register-blocked network in
// 132 KB of dynamic LDS (8 or 16 keys per thread), longer ones in place in global memory with the
// plain network.  Up to one workgroup per CU strides over the tile table.
// (Tried and dropped: splitting a long list into 2..32 depth buckets -- sample-sort splitters in
// a 1024-thread partition kernel -- and sorting the buckets as short lists.  5 M @ 1080p, 9150
// keys per tile: 3.0-6.0 ms against 3.6 ms for this whole-list network; centre-heavy 1 M frame:
// 272 vs 239 us.  The 4096-key unit has a 60 us latency; long lists need a different sort.)
#define SORT_BIG_THREADS 1024
#define SORT_LDS_LARGE_KEYS 16384u
__global__ __launch_bounds__(SORT_BIG_THREADS) void k_tile_sort_big(int T, const uint2* __restrict__ ranges,
                                                                    uint64_t* keys, uint32_t* __restrict__ point_list,
                                                                    uint32_t lo, uint32_t hi,
                                                                    const uint32_t* __restrict__ ctrl, uint32_t cap)
{
    extern __shared__ uint64_t sk_dyn[];
    uint64_t* sk = sk_dyn;
    if (ctrl[GFT_CTRL_TOTAL] > cap) return;
    if (ctrl[GFT_CTRL_MAXCNT] <= lo) return;
    const int tid = threadIdx.x;
    for (int tile = blockIdx.x; tile < T; tile += gridDim.x) {
        const uint2 r = ranges[tile];
        const uint32_t n = r.y - r.x;
        if (n <= lo) continue;                       // uniform per workgroup
        if (n <= hi) {
            const uint32_t npad = n <= 8192u ? 8192u : 16384u;
            for (uint32_t i = tid; i < npad; i += SORT_BIG_THREADS) sk[sort_slot(i)] = i < n ? keys[r.x + i] : ~0ull;
            __syncthreads();
            if (npad == 8192u) bitonic_blocked<3, 10>(sk, tid);
            else bitonic_blocked<4, 10>(sk, tid);
            for (uint32_t i = tid; i < n; i += SORT_BIG_THREADS) point_list[r.x + i] = (uint32_t)sk[sort_slot(i)];
            __syncthreads();
        } else {
            uint64_t* seg = keys + r.x;
            const uint32_t npad = next_pow2(n);
            bitonic_ascending<SORT_BIG_THREADS>(
                n, npad, tid,
                [&](uint32_t i) { return __hip_atomic_load(&seg[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); },
                [&](uint32_t i, uint64_t v) { __hip_atomic_store(&seg[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); },
                [] { __threadfence_block(); __syncthreads(); });
            for (uint32_t i = tid; i < n; i += SORT_BIG_THREADS) point_list[r.x + i] = (uint32_t)seg[i];
        }
    }
}

// Sorts one segment of nt list entries starting at `first` into point_list, by 1024 threads with the register-blocked
// network in LDS (up to 16384 keys) or in place in the key array beyond.  FROM_IDS: the segment holds unsorted ids in
// point_list (the tail k_tile_front left) and the keys are rebuilt by a depth gather; else it holds scattered keys.
template <bool FROM_IDS>
__device__ __forceinline__ void tail_sort_segment(uint32_t first, uint32_t nt, uint64_t* keys, uint32_t* __restrict__ point_list,
                                                  const float* __restrict__ depth, uint64_t* sk, int tid)
{
    if (nt == 0u) return;
    auto key_of = [&](uint32_t i) -> uint64_t {
        if (!FROM_IDS) return keys[first + i];
        const uint32_t id = point_list[first + i];
        return ((uint64_t)__float_as_uint(depth[id]) << 32) | id;
    };
    if (nt <= SORT_LDS_LARGE_KEYS) {
        if (nt <= 1024u) {
            const uint32_t npad = next_pow2(nt < 2u ? 2u : nt);
            for (uint32_t i = tid; i < nt; i += SORT_BIG_THREADS) sk[i] = key_of(i);
            __syncthreads();
            bitonic_ascending<SORT_BIG_THREADS>(nt, npad, tid, [&](uint32_t i) { return sk[i]; },
                                                [&](uint32_t i, uint64_t v) { sk[i] = v; }, [] { __syncthreads(); });
            for (uint32_t i = tid; i < nt; i += SORT_BIG_THREADS) point_list[first + i] = (uint32_t)sk[i];
        } else {
            const uint32_t npad = nt <= 4096u ? 4096u : (nt <= 8192u ? 8192u : 16384u);
            // all keys are built before any id is overwritten (the ids live in point_list)
            for (uint32_t i = tid; i < npad; i += SORT_BIG_THREADS) sk[sort_slot(i)] = i < nt ? key_of(i) : ~0ull;
            __syncthreads();
            if (npad == 4096u) bitonic_blocked<2, 10>(sk, tid);
            else if (npad == 8192u) bitonic_blocked<3, 10>(sk, tid);
            else bitonic_blocked<4, 10>(sk, tid);
            for (uint32_t i = tid; i < nt; i += SORT_BIG_THREADS) point_list[first + i] = (uint32_t)sk[sort_slot(i)];
        }
        __syncthreads();
    } else {
        uint64_t* seg = keys + first;                       // (FROM_IDS: the scattered keys are no longer needed)
        if (FROM_IDS) {
            for (uint32_t i = tid; i < nt; i += SORT_BIG_THREADS) seg[i] = key_of(i);
        }
        __threadfence_block();
        __syncthreads();
        const uint32_t npad = next_pow2(nt);
        bitonic_ascending<SORT_BIG_THREADS>(
            nt, npad, tid,
            [&](uint32_t i) { return __hip_atomic_load(&seg[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); },
            [&](uint32_t i, uint64_t v) { __hip_atomic_store(&seg[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); },
            [] { __threadfence_block(); __syncthreads(); });
        for (uint32_t i = tid; i < nt; i += SORT_BIG_THREADS) point_list[first + i] = (uint32_t)seg[i];
        __syncthreads();
    }
}

// For every tile that has a flagged quadrant: sorts the unsorted tail of its id list (lazy sort, see k_tile_front).
__global__ __launch_bounds__(SORT_BIG_THREADS) void k_tile_tail(int T, const uint2* __restrict__ ranges, uint64_t* keys,
                                                                uint32_t* __restrict__ point_list,
                                                                const float* __restrict__ depth,
                                                                const uint32_t* __restrict__ front_len,
                                                                const uint32_t* __restrict__ unit_flag,
                                                                uint32_t* ctrl, uint32_t cap,
                                                                const uint32_t* __restrict__ quad_max, uint32_t* __restrict__ order)
{
    extern __shared__ uint64_t sk_dyn[];
    uint64_t* sk = sk_dyn;
    if (ctrl[GFT_CTRL_TOTAL] > cap) return;
    // The backward's heavy-first tile order rides on this launch (no separate launch in the backward): by the deepest
    // contributors of the first render pass; a tile with a flagged quadrant, whose walk goes on, counts as heaviest.
    if (order && blockIdx.x == 0) {
        gft_tile_order_block(T, quad_max, order, ctrl[GFT_CTRL_NFLAG] ? unit_flag : nullptr);
        if (threadIdx.x == 0) ctrl[GFT_CTRL_ORDER_OK] = 1u;
    }
    if (ctrl[GFT_CTRL_NFLAG] == 0u) return;
    const int tid = threadIdx.x;
    for (int tile = blockIdx.x; tile < T; tile += gridDim.x) {
        const uint4 f = reinterpret_cast<const uint4*>(unit_flag)[tile];
        if ((f.x | f.y | f.z | f.w) == 0u) continue;           // uniform per workgroup
        const uint2 r = ranges[tile];
        const uint32_t kf = front_len[tile];
        tail_sort_segment<true>(r.x + kf, (r.y - r.x) - kf, keys, point_list, depth, sk, tid);
    }
}

}  // namespace

#define SORT_LDS_SMALL 4096u      // 32 KB of LDS
#define SORT_LDS_LARGE 16384u     // 128 KB of LDS (+ 4 KB of bank padding)
#define SORT_LDS_LARGE_BYTES (SORT_SLOTS(SORT_LDS_LARGE) * 8)

hipError_t gft_launch_tile_count(hipStream_t s, const gft_config& c, const GeomView& g, const ImgView& im,
                                 uint32_t* mail, uint32_t seq, int64_t status_cap)
{
    const int gx = (c.W + GFT_TILE_X - 1) / GFT_TILE_X, gy = (c.H + GFT_TILE_Y - 1) / GFT_TILE_Y;
    const int T = gx * gy;
    CountArgs a;
    const int blocks = (c.P + BIN_CHUNK - 1) / BIN_CHUNK;
    a.P = c.P; a.gx = gx; a.T = T;
    a.rect = g.rect;
    a.tile_cnt = im.tile_cnt;
    a.ranges = im.ranges;
    a.cursor = im.tile_cursor; a.ctrl = im.ctrl; a.mail = mail; a.seq = seq;
    a.post = (status_cap >= 0 && mail) ? (uint32_t)status_cap + 1u : 0u;
    a.blockhist = (T <= BIN_LDS_MAX_TILES && T <= GFT_BLOCKHIST_TILES) ? g.blockhist : nullptr;
    if (T <= BIN_LDS_MAX_TILES) hipLaunchKernelGGL((k_tile_count<true>), dim3(blocks), dim3(BIN_THREADS), (size_t)T * 4, s, a);
    else hipLaunchKernelGGL((k_tile_count<false>), dim3(blocks), dim3(BIN_THREADS), 0, s, a);
    return hipGetLastError();
}

hipError_t gft_launch_tile_scatter(hipStream_t s, const gft_config& c, const GeomView& g, const ImgView& im,
                                   const BinView& b, uint32_t cap, int64_t expect_total)
{
    const int gx = (c.W + GFT_TILE_X - 1) / GFT_TILE_X, gy = (c.H + GFT_TILE_Y - 1) / GFT_TILE_Y;
    const int T = gx * gy;
    ScatterArgs a;
    const int blocks = (c.P + BIN_CHUNK - 1) / BIN_CHUNK;
    a.P = c.P; a.gx = gx; a.T = T;
    a.rect = g.rect; a.depth = g.depth;
    a.ranges = im.ranges;
    a.cursor = im.tile_cursor; a.keys = b.keys; a.ctrl = im.ctrl; a.cap = cap; a.stage_cap = 0;
    a.blockhist = nullptr;
    if (T > BIN_LDS_MAX_TILES) {
        hipLaunchKernelGGL((k_tile_scatter<0>), dim3(blocks), dim3(BIN_THREADS), 0, s, a);
        return hipGetLastError();
    }
    {
        static std::atomic<uint64_t> done[2];
        hipError_t e = gft_lds_opt_in(reinterpret_cast<const void*>(&k_tile_scatter<1>), 2 * BIN_LDS_MAX_TILES * 4, done[0]);
        if (e == hipSuccess) e = gft_lds_opt_in(reinterpret_cast<const void*>(&k_tile_scatter<2>), BIN_STAGE_LDS_BYTES, done[1]);
        if (e != hipSuccess) return e;
    }
    // Staging pays when a workgroup's instances (about expect_total / blocks: the caller's estimate of the
    // instances this pass scatters) fit the LDS that the three per-tile tables leave free; workgroups that exceed
    // it write directly.  It also pins one workgroup per CU, so it is not used when most would not fit.
    const size_t tables = ((size_t)3 * T + (T & 1)) * 4;
    const size_t stage_cap = tables + 4096 * 8 <= BIN_STAGE_LDS_BYTES ? (BIN_STAGE_LDS_BYTES - tables) / 8 : 0;
    const size_t expect = blocks > 0 ? (size_t)(expect_total > 0 ? expect_total : 0) / (size_t)blocks : 0;
    a.blockhist = T <= GFT_BLOCKHIST_TILES ? g.blockhist : nullptr;
    if (stage_cap > 0 && expect <= stage_cap + stage_cap / 2) {
        a.stage_cap = (uint32_t)stage_cap;
        hipLaunchKernelGGL((k_tile_scatter<2>), dim3(blocks), dim3(BIN_THREADS), tables + stage_cap * 8, s, a);
    } else {
        hipLaunchKernelGGL((k_tile_scatter<1>), dim3(blocks), dim3(BIN_THREADS), (size_t)T * 8, s, a);
    }
    return hipGetLastError();
}

hipError_t gft_launch_tile_sort(hipStream_t s, const gft_config& c, int64_t max_tile_list, const ImgView& im,
                                const BinView& b, uint32_t cap, float* clear, size_t clear_bytes)
{
    const int gx = (c.W + GFT_TILE_X - 1) / GFT_TILE_X, gy = (c.H + GFT_TILE_Y - 1) / GFT_TILE_Y;
    const int T = gx * gy;
    hipLaunchKernelGGL(k_tile_sort_small, dim3(T), dim3(GFT_BLOCK), 0, s, im.ranges, b.keys, b.point_list, im.ctrl, cap,
                       reinterpret_cast<float4*>(clear), clear_bytes / 16);
    // the longest list (known to the host in the two-stage flow, a guess of the caller in the
    // one-call flow, <= 0 = unknown) tells whether any tile needs the long-list path
    if (max_tile_list <= 0 || max_tile_list > (int64_t)SORT_LDS_SMALL) return gft_launch_tile_sort_long(s, c, im, b, cap);
    return hipGetLastError();
}

hipError_t gft_launch_tile_front(hipStream_t s, const gft_config& c, const ImgView& im, const BinView& b, uint32_t cap,
                                 float* clear, size_t clear_bytes)
{
    const int gx = (c.W + GFT_TILE_X - 1) / GFT_TILE_X, gy = (c.H + GFT_TILE_Y - 1) / GFT_TILE_Y;
    const int T = gx * gy;
    hipLaunchKernelGGL(k_tile_front, dim3(T), dim3(GFT_BLOCK), 0, s, im.ranges, b.keys, b.point_list, im.front_len,
                       im.unit_flag, im.ctrl, cap, reinterpret_cast<float4*>(clear), clear_bytes / 16);
    return hipGetLastError();
}

hipError_t gft_launch_tile_tail(hipStream_t s, const gft_config& c, const GeomView& g, const ImgView& im,
                                const BinView& b, uint32_t cap, bool want_order)
{
    const int gx = (c.W + GFT_TILE_X - 1) / GFT_TILE_X, gy = (c.H + GFT_TILE_Y - 1) / GFT_TILE_Y;
    const int T = gx * gy;
    {
        static std::atomic<uint64_t> done{0};
        const hipError_t e = gft_lds_opt_in(reinterpret_cast<const void*>(&k_tile_tail), SORT_LDS_LARGE_BYTES, done);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_tile_tail, dim3(T < 256 ? T : 256), dim3(SORT_BIG_THREADS), (size_t)SORT_LDS_LARGE_BYTES, s, T,
                       im.ranges, b.keys, b.point_list, g.depth, im.front_len, im.unit_flag, im.ctrl, cap,
                       im.tile_max, want_order ? im.tile_order : nullptr);
    return hipGetLastError();
}

hipError_t gft_launch_tile_sort_long(hipStream_t s, const gft_config& c, const ImgView& im, const BinView& b,
                                     uint32_t cap)
{
    const int gx = (c.W + GFT_TILE_X - 1) / GFT_TILE_X, gy = (c.H + GFT_TILE_Y - 1) / GFT_TILE_Y;
    const int T = gx * gy;
    {
        static std::atomic<uint64_t> done{0};
        const hipError_t e = gft_lds_opt_in(reinterpret_cast<const void*>(&k_tile_sort_big), SORT_LDS_LARGE_BYTES, done);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_tile_sort_big, dim3(T < 256 ? T : 256), dim3(SORT_BIG_THREADS), (size_t)SORT_LDS_LARGE_BYTES, s, T,
                       im.ranges, b.keys, b.point_list, SORT_LDS_SMALL, SORT_LDS_LARGE, im.ctrl, cap);
    return hipGetLastError();
}
