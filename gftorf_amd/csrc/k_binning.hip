// k_binning.hip -- instance binning (gfx950): per-tile counting, scatter of
// (depth, id) keys into tile segments, per-tile sort in LDS.
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
//   k_tile_sort    : one workgroup per tile: bitonic sort of the tile's keys in LDS
//                    (all comparators ascending, so the power-of-two padding never
//                    moves), writes the id list               [8 B read + 4 B written]
//
// The order inside a tile segment after the scatter is arbitrary (atomic cursors); the
// sort makes the final lists deterministic and bit-identical to the reference's.
#include "gft_internal.h"

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

// ---- lazy binning ---------------------------------------------------------------------------
// A pixel stops reading its tile list once its transmittance is below 1e-4, so in a dense frame most
// instances are never read (metric frame: 3.6 M instances, 0.6 M up to the deepest contributor of every
// tile; 5 M @ 1080p: 75 M and 2 M).  With a depth cut only the NEAR slab (view z <= cut) is counted,
// scattered and sorted up front; a quadrant that runs out of near-slab entries with unsaturated pixels is
// flagged (the flag of the lazy sort), and only then is the FAR slab binned, for the tiles that have a
// flagged quadrant (count pass 1, scatter pass 1, k_tile_tail), and those quadrants resume.  Every key of
// the near slab is smaller than every key of the far slab, so a tile's list is its sorted near segment
// followed by its sorted far segment: the same order, the same arithmetic, bit-identical results.  The cut
// is any float; a good one comes from the previous frame: pass 0 also histograms the instances over 256
// log-spaced depth bins and suggests the cut that puts GFT_NEAR_SLAB_PER_TILE instances per tile into the
// near slab (no cut when the frame has less than 3 x that).
struct DepthBins {
    float inv_near;      // 1 / near_n
    float scale;         // GFT_DHIST_BINS / log2(far_n / near_n)
    float near_n;
};

// Does a depth cut pay for a frame of R instances?  `target` = wanted near-slab instances (caller's slab width x tiles),
// `target_min` = the same at the default width.  The frame must be dense (R >= 3 x the default slab: in a sparser one
// too many quadrants outlive any near slab and the second pass costs more than the first one saved), and the slab the
// caller asks for -- possibly widened after frames with flagged quadrants -- must still leave out half of the frame.
__device__ __forceinline__ bool cut_pays(uint32_t R, uint32_t target, uint32_t target_min)
{
    return target > 0u && R / 3u >= target_min && R / 2u >= target;
}

__device__ __forceinline__ int depth_bin(uint32_t dbits, const DepthBins& db)
{
    const float b = __log2f(__uint_as_float(dbits) * db.inv_near) * db.scale;
    const int i = (int)b;
    return i < 0 ? 0 : (i >= GFT_DHIST_BINS ? GFT_DHIST_BINS - 1 : i);
}

// Total of the depth histogram h[GFT_DHIST_BINS] (LDS) and the first bin at which the running count reaches `target`
// (GFT_DHIST_BINS when it never does).  Called by all threads of a workgroup of >= GFT_DHIST_BINS threads.
__device__ void dhist_scan(const uint32_t* h, uint32_t target, uint32_t& total, uint32_t& cut_bin)
{
    __shared__ uint32_t s_w[GFT_DHIST_BINS / 64];
    __shared__ uint32_t s_cut;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t v = 0, x = 0;
    if (tid < GFT_DHIST_BINS) {
        v = h[tid];
        x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) s_w[wave] = x;
    }
    if (tid == 0) s_cut = GFT_DHIST_BINS;
    __syncthreads();
    uint32_t tot = 0;
    for (int w = 0; w < GFT_DHIST_BINS / 64; w++) tot += s_w[w];
    if (tid < GFT_DHIST_BINS) {
        uint32_t incl = x;
        for (int w = 0; w < wave; w++) incl += s_w[w];
        if (incl >= target && incl - v < target) s_cut = (uint32_t)tid;       // the one bin where the count crosses
    }
    __syncthreads();
    total = tot;
    cut_bin = s_cut;
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
    int items;                      // Gaussians per thread (BIN_ITEMS, more when a depth cut leaves few of them to bin)
    const ushort4* __restrict__ rect;
    const float* __restrict__ depth;
    uint32_t cut_bits;              // near slab: depth bits <= cut_bits
    uint32_t* tile_cnt;             // pass 0: near-slab counters; pass 1: far-slab counters
    uint2* __restrict__ ranges;     // pass 0: ranges; pass 1: ranges1
    uint32_t* __restrict__ cursor;
    uint32_t* ctrl;
    uint32_t* mail; uint32_t seq;   // pass 0 only
    uint16_t* __restrict__ blockhist;
    uint32_t* dhist;                // pass 0 only
    DepthBins db;
    uint32_t target;                // wanted near-slab instances of the next frame
    uint32_t target_min;            // the same at the default slab width (cut_pays)
    const uint32_t* __restrict__ unit_flag;   // pass 1 only
    uint32_t cap;
};

// tiles with a flagged quadrant as a bit table in LDS (pass 1 of count and scatter)
__device__ __forceinline__ void load_flagged_tiles(uint32_t* bits, int T, const uint32_t* __restrict__ unit_flag)
{
    const int words = (T + 31) >> 5;
    for (int w = threadIdx.x; w < words; w += BIN_THREADS) bits[w] = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += BIN_THREADS) {
        const uint4 f = reinterpret_cast<const uint4*>(unit_flag)[t];
        if (f.x | f.y | f.z | f.w) atomicOr(&bits[t >> 5], 1u << (t & 31));
    }
    __syncthreads();
}

template <bool USE_LDS, int PASS>
__global__ __launch_bounds__(BIN_THREADS) void k_tile_count(CountArgs a)
{
    extern __shared__ uint32_t hist[];
    __shared__ uint32_t s_last;
    __shared__ uint32_t s_dh[GFT_DHIST_BINS];
    __shared__ uint32_t s_flag[BIN_LDS_MAX_TILES / 32];
    const int tid = threadIdx.x;
    const int T = a.T, gx = a.gx;
    if (PASS == 1) {
        // nothing was flagged, the binning buffer is too small, or the far slab is empty: every workgroup leaves
        if (a.ctrl[GFT_CTRL_NFLAG] == 0u || a.ctrl[GFT_CTRL_TOTAL] > a.cap || a.ctrl[GFT_CTRL_TOTAL] == a.ctrl[GFT_CTRL_TOTAL0])
            return;
        if (USE_LDS) load_flagged_tiles(s_flag, T, a.unit_flag);
    }
    if (USE_LDS) {
        for (int i = tid; i < T; i += BIN_THREADS) hist[i] = 0;
    }
    if (PASS == 0 && tid < GFT_DHIST_BINS) s_dh[tid] = 0;
    __syncthreads();
    const int base = blockIdx.x * (BIN_THREADS * a.items);
    // four Gaussians per trip, their rectangle and depth loads issued together (a.items is a multiple of 4)
    for (int k0 = 0; k0 < a.items; k0 += 4) {
        ushort4 r4[4];
        uint32_t d4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int idx = base + (k0 + u) * BIN_THREADS + tid;
            const bool in = idx < a.P;
            r4[u] = in ? a.rect[idx] : make_ushort4(0, 0, 0, 0);
            d4[u] = in ? __float_as_uint(a.depth[idx]) : 0u;       // (not written for culled Gaussians, not used for them either)
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const ushort4 r = r4[u];
            const uint32_t tiles = (uint32_t)(r.z - r.x) * (uint32_t)(r.w - r.y);
            if (tiles == 0u) continue;
            const uint32_t d = d4[u];
            if (PASS == 0) atomicAdd(&s_dh[depth_bin(d, a.db)], tiles);
            if ((PASS == 0) != (d <= a.cut_bits)) continue;          // pass 0: near slab, pass 1: far slab
            for (int y = r.y; y < r.w; y++)
                for (int x = r.x; x < r.z; x++) {
                    const int t = y * gx + x;
                    if (PASS == 1) {
                        const bool fl = USE_LDS ? ((s_flag[t >> 5] >> (t & 31)) & 1u) != 0u
                                                : (a.unit_flag[4 * t] | a.unit_flag[4 * t + 1] | a.unit_flag[4 * t + 2] | a.unit_flag[4 * t + 3]) != 0u;
                        if (!fl) continue;
                    }
                    if (USE_LDS) atomicAdd(&hist[t], 1u);
                    else atomicAdd(&a.tile_cnt[t], 1u);
                }
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
    if (PASS == 0 && tid < GFT_DHIST_BINS) {
        const uint32_t h = s_dh[tid];
        if (h) atomicAdd(&a.dhist[tid], h);
    }
    // The workgroup that draws the last ticket scans.  Every counter update above is a
    // device-scope atomic, complete once vmcnt drains, and the scan reads the counters with
    // device-scope loads: no cache write-back / invalidate (__threadfence would flush the
    // L2 lines the preprocess kernel just wrote, ~50 us) is needed for that hand-over.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(&a.ctrl[PASS == 0 ? GFT_CTRL_DONE : GFT_CTRL_DONE1], 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    uint32_t total = 0, longest = 0;
    if (PASS == 1) {
        // far-slab segments follow the near slab in the key / id arrays
        tile_scan_block(T, a.tile_cnt, a.ranges, a.cursor, a.ctrl[GFT_CTRL_TOTAL0], total, longest);
        if (tid == 0) a.ctrl[GFT_CTRL_TOTAL1] = total;
        return;
    }
    tile_scan_block(T, a.tile_cnt, a.ranges, a.cursor, 0u, total, longest);
    // depth histogram: R = all instances; the cut for the next frame = upper edge of the first bin at which the
    // running count reaches the target
    __shared__ uint32_t s_cum[GFT_DHIST_BINS];
    if (tid < GFT_DHIST_BINS) s_cum[tid] = __hip_atomic_load(&a.dhist[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    uint32_t R, cut_bin;
    dhist_scan(s_cum, a.target, R, cut_bin);
    if (tid == 0) {
        uint32_t cut_next = GFT_NO_CUT;
        // (a cut pays when the near slab is a small part of the frame: with R below 3 x the target too many
        // quadrants outlive the near slab and the second pass costs more than the first one saved)
        if (cut_pays(R, a.target, a.target_min) && cut_bin + 1u < GFT_DHIST_BINS)
            cut_next = __float_as_uint(a.db.near_n * exp2f((float)(cut_bin + 1u) / a.db.scale));
        a.ctrl[GFT_CTRL_TOTAL] = R;
        a.ctrl[GFT_CTRL_TOTAL0] = total;
        a.ctrl[GFT_CTRL_MAXCNT] = longest;
        a.ctrl[GFT_CTRL_CUTNEXT] = cut_next;
        if (a.mail) {
            a.mail[GFT_CTRL_TOTAL] = R;          // (GFT_CTRL_FLAGS of the slot belongs to the preprocess kernel)
            a.mail[GFT_CTRL_MAXCNT] = longest;
            a.mail[GFT_CTRL_TOTAL0] = total;
            a.mail[GFT_CTRL_CUTNEXT] = cut_next;
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
// PASS 0: near slab (depth bits <= cut); PASS 1: far slab of the flagged tiles (see lazy binning above)
struct ScatterArgs {
    int P, gx, T;
    int items;                      // as in the count pass (the per-workgroup histograms it kept are reused)
    const ushort4* __restrict__ rect;
    const float* __restrict__ depth;
    const uint2* __restrict__ ranges;
    uint32_t* __restrict__ cursor;
    uint64_t* __restrict__ keys;
    const uint32_t* __restrict__ ctrl;
    uint32_t cap, stage_cap;
    const uint16_t* __restrict__ blockhist;
    uint32_t cut_bits;
    const uint32_t* __restrict__ unit_flag;
};

template <int MODE, int PASS>
__global__ __launch_bounds__(BIN_THREADS) void k_tile_scatter(ScatterArgs a)
{
    extern __shared__ uint32_t sh[];
    __shared__ uint32_t s_flag[BIN_LDS_MAX_TILES / 32];
    if (a.ctrl[GFT_CTRL_TOTAL] > a.cap) return;      // binning buffer too small: the host re-runs stage 2
    const int T = a.T, gx = a.gx, P = a.P;
    if (PASS == 1) {
        if (a.ctrl[GFT_CTRL_NFLAG] == 0u || a.ctrl[GFT_CTRL_TOTAL] == a.ctrl[GFT_CTRL_TOTAL0] || a.ctrl[GFT_CTRL_TOTAL1] == 0u) return;
        if (MODE >= 1) load_flagged_tiles(s_flag, T, a.unit_flag);
    }
    auto flagged = [&](int t) -> bool {
        if (MODE >= 1) return ((s_flag[t >> 5] >> (t & 31)) & 1u) != 0u;
        return (a.unit_flag[4 * t] | a.unit_flag[4 * t + 1] | a.unit_flag[4 * t + 2] | a.unit_flag[4 * t + 3]) != 0u;
    };
    uint32_t* cnt = sh;          // [T] instances of this block per tile, then running slot
    uint32_t* first = sh + T;    // [T] global position of this block's chunk in the tile segment
    uint32_t* lstart = sh + 2 * T;                                   // [T] MODE 2: chunk start in the LDS stage
    uint64_t* stage = reinterpret_cast<uint64_t*>(sh + 3 * T + (T & 1));   // MODE 2: stage_cap keys, 8-B aligned
    __shared__ uint32_t s_wave_tot[BIN_THREADS / 64];
    __shared__ uint32_t s_block_tot;
    const int tid = threadIdx.x;
    const int base = blockIdx.x * (BIN_THREADS * a.items);
    bool staged = false;
    if (MODE >= 1) {
        if (a.blockhist) {
            // the count kernel kept this workgroup's histogram
            for (int i = tid; i < T; i += BIN_THREADS) cnt[i] = a.blockhist[(size_t)blockIdx.x * GFT_BLOCKHIST_TILES + i];
        } else {
            for (int i = tid; i < T; i += BIN_THREADS) cnt[i] = 0;
            __syncthreads();
#pragma unroll 4
            for (int k = 0; k < a.items; k++) {
                const int idx = base + k * BIN_THREADS + tid;
                if (idx < P) {
                    const ushort4 r = a.rect[idx];
                    if (!(r.z > r.x && r.w > r.y)) continue;
                    if ((PASS == 0) != (__float_as_uint(a.depth[idx]) <= a.cut_bits)) continue;
                    for (int y = r.y; y < r.w; y++)
                        for (int x = r.x; x < r.z; x++) {
                            if (PASS == 1 && !flagged(y * gx + x)) continue;
                            atomicAdd(&cnt[y * gx + x], 1u);
                        }
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
    for (int k0 = 0; k0 < a.items; k0 += 2) {
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
            if ((PASS == 0) != (d <= a.cut_bits)) continue;
            const uint32_t idx = (uint32_t)(base + (k0 + u) * BIN_THREADS + tid);
            const uint64_t key = ((uint64_t)d << 32) | idx;
            for (int y = r.y; y < r.w; y++)
                for (int x = r.x; x < r.z; x++) {
                    const int t = y * gx + x;
                    if (PASS == 1 && !flagged(t)) continue;
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

// Bitonic network with ascending comparators only: merge step k starts with the
// mirror stage (i <-> i ^ (k-1)), then half-cleaners at distances k/4 .. 1.
// Elements at positions >= n are +inf by construction and never move.
template <int THREADS, typename Ld, typename St, typename Sync>
__device__ __forceinline__ void bitonic_ascending(uint32_t n, uint32_t npad, int tid, Ld ld, St st, Sync sync,
                                                  uint32_t k_first = 2)
{
    // k_first > 2: runs of k_first / 2 keys are already ascending
    for (uint32_t k = k_first; k <= npad; k <<= 1) {
        const uint32_t half = k >> 1;
        for (uint32_t c = tid; c < (npad >> 1); c += THREADS) {
            const uint32_t blk = c / half, off = c - blk * half;
            const uint32_t i = blk * k + off, l = blk * k + (k - 1 - off);
            if (l < n) {
                const uint64_t a = ld(i), b = ld(l);
                if (a > b) { st(i, b); st(l, a); }
            }
        }
        sync();
        for (uint32_t j = k >> 2; j > 0; j >>= 1) {
            for (uint32_t c = tid; c < (npad >> 1); c += THREADS) {
                const uint32_t i = ((c & ~(j - 1)) << 1) | (c & (j - 1)), l = i + j;
                if (l < n) {
                    const uint64_t a = ld(i), b = ld(l);
                    if (a > b) { st(i, b); st(l, a); }
                }
            }
            sync();
        }
    }
}

__device__ __forceinline__ uint32_t next_pow2(uint32_t n)
{
    uint32_t p = 2;
    while (p < n) p <<= 1;
    return p;
}

// ---- register-blocked bitonic sort (lists of 1025..4096 keys) -------------------
// Standard bitonic network on npad = (1 << LOG_T) << LOG_E keys (+inf padded), 1 << LOG_T threads.  A thread owns
// E = 2^LOG_E keys in registers; which keys depends on the layout b: the thread's register
// index supplies key-index bits [b, b+LOG_E), the thread id supplies the rest.  All stages
// whose distance bit falls inside [b, b+LOG_E) are compare-exchanges between registers; the
// keys travel through LDS only when the layout changes (about 20 round trips instead of 78
// LDS stages for 4096 keys).  LDS slot of key i: sort_slot(i) (bank spreading).
// XOR swizzle: conflict-free ds_read_b64 / ds_write_b64 in all three layouts of the 16-keys-per-
// thread network (an i + (i >> 5) padding costs 1.33x there and 1 KB per 4096 keys, which keeps
// a fifth workgroup off the CU)
// (measured: 76.2 -> 72.1 us for 1200 lists of ~3000 keys)
__device__ __forceinline__ uint32_t sort_slot(uint32_t i) { return i ^ ((i >> 4) & 31u); }
#define SORT_SLOTS(n) (n)

template <int LOG_E>
__device__ __forceinline__ uint32_t key_index(int t, int r, int b)
{
    return ((uint32_t)(t >> b) << (b + LOG_E)) | ((uint32_t)r << b) | ((uint32_t)t & ((1u << b) - 1u));
}

template <int LOG_E>
__device__ __forceinline__ void regs_from_lds(uint64_t* v, const uint64_t* sk, int t, int b)
{
#pragma unroll
    for (int r = 0; r < (1 << LOG_E); r++) v[r] = sk[sort_slot(key_index<LOG_E>(t, r, b))];
}

template <int LOG_E>
__device__ __forceinline__ void regs_to_lds(const uint64_t* v, uint64_t* sk, int t, int b)
{
#pragma unroll
    for (int r = 0; r < (1 << LOG_E); r++) sk[sort_slot(key_index<LOG_E>(t, r, b))] = v[r];
}

// one bitonic stage between registers: distance bit S (register-index bit), merge bit m
template <int LOG_E, int S>
__device__ __forceinline__ void reg_stage(uint64_t* v, int t, int b, int m, int LG)
{
#pragma unroll
    for (int r = 0; r < (1 << LOG_E); r++) {
        if (r & (1 << S)) continue;
        const int r2 = r | (1 << S);
        const uint32_t i = key_index<LOG_E>(t, r, b);
        const bool up = (m >= LG) || (((i >> m) & 1u) == 0u);
        const uint64_t x = v[r], y = v[r2];
        const bool sw = (x > y) == up;
        v[r] = sw ? y : x;
        v[r2] = sw ? x : y;
    }
}

template <int LOG_E>
__device__ __forceinline__ void reg_stage_dyn(uint64_t* v, int s_local, int t, int b, int m, int LG)
{
    switch (s_local) {
    case 0: reg_stage<LOG_E, 0>(v, t, b, m, LG); break;
    case 1: reg_stage<LOG_E, 1>(v, t, b, m, LG); break;
    case 2: reg_stage<LOG_E, 2>(v, t, b, m, LG); break;
    case 3:
        if (LOG_E > 3) reg_stage<LOG_E, (LOG_E > 3 ? 3 : 0)>(v, t, b, m, LG);
        break;
    default:
        if (LOG_E > 4) reg_stage<LOG_E, (LOG_E > 4 ? 4 : 0)>(v, t, b, m, LG);
        break;
    }
}

struct BlockSync { __device__ __forceinline__ void operator()() const { __syncthreads(); } };
// the keys of the network belong to one wave: LDS operations of a wave complete in order
struct WaveSync {
    __device__ __forceinline__ void operator()() const
    {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
};

template <int LOG_E, int LOG_T, typename Sync = BlockSync>
__device__ __forceinline__ void bitonic_blocked(uint64_t* sk, int tid, Sync sync = Sync())
{
    constexpr int E = 1 << LOG_E;
    constexpr int LG = LOG_T + LOG_E;      // log2(npad)
    uint64_t v[E];
    int b = 0;                             // current layout (compile-time after unrolling)
    regs_from_lds<LOG_E>(v, sk, tid, 0);
    // the whole schedule is unrolled: every layout, distance and register pair is a constant,
    // which keeps the E keys in registers
#pragma unroll
    for (int m = 1; m <= LG; m++) {        // merge size 2^m
#pragma unroll
        for (int s = m - 1; s >= 0; s--) { // distance bit
            // layout that holds bit s: chunks of LOG_E bits from the bottom, top chunk clipped
            int nb = (s / LOG_E) * LOG_E;
            if (nb > LG - LOG_E) nb = LG - LOG_E;
            if (nb != b) {
                sync();
                regs_to_lds<LOG_E>(v, sk, tid, b);
                sync();
                b = nb;
                regs_from_lds<LOG_E>(v, sk, tid, b);
            }
            reg_stage_dyn<LOG_E>(v, s - b, tid, b, m, LG);
        }
    }
    // b == 0 here (the last stages of every merge are in the natural layout)
    sync();
    regs_to_lds<LOG_E>(v, sk, tid, 0);
    sync();
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

// Sorts the n keys in sk[0, npad) (pads = ~0, npad a power of two <= 1024) with 256 threads and
// writes the ids.  1024-key class: every wave first sorts its own quarter of 256 keys in registers
// (no workgroup barrier: the keys of a quarter belong to one wave), then the two last merge steps
// (19 stages) run across the workgroup -- 19 barriers instead of the 55 of the plain network.
__device__ __forceinline__ void head_sort_and_store(uint64_t* sk, uint32_t n, uint32_t npad, int tid, uint32_t* __restrict__ ids)
{
    if (n <= 1u) {
        if (n == 1u && tid == 0) ids[0] = (uint32_t)sk[0];
        return;
    }
    if (npad == 1024u) {
        const int lane = tid & 63, wave = tid >> 6;
        bitonic_blocked<2, 6, WaveSync>(sk + 256 * wave, lane, WaveSync());
        __syncthreads();
        // key i of a sorted quarter sits at its swizzled slot
        auto slot = [](uint32_t i) { return (i & ~255u) | sort_slot(i & 255u); };
        bitonic_ascending<GFT_BLOCK>(n, npad, tid, [&](uint32_t i) { return sk[slot(i)]; },
                                     [&](uint32_t i, uint64_t v) { sk[slot(i)] = v; }, [] { __syncthreads(); }, 512u);
        for (uint32_t i = tid; i < n; i += GFT_BLOCK) ids[i] = (uint32_t)sk[slot(i)];
        return;
    }
    bitonic_ascending<GFT_BLOCK>(n, npad, tid, [&](uint32_t i) { return sk[i]; }, [&](uint32_t i, uint64_t v) { sk[i] = v; },
                                 [] { __syncthreads(); });
    for (uint32_t i = tid; i < n; i += GFT_BLOCK) ids[i] = (uint32_t)sk[i];
}

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

// ---- tile-pull binning of the near slab --------------------------------------------------------------------
// With a depth cut the near slab is small (about 900 instances per tile by construction), and the count / scatter
// kernels above spend their time on per-workgroup passes over the T-entry tile tables, not on the instances.  The
// near slab is therefore binned the other way round: the near Gaussians are first dealt to SUPERTILES of S x S tiles
// (k_super_count, k_super_scatter: two light passes over rect / depth with a <= 256-entry table), then one workgroup
// per tile pulls its entries out of its supertile's list (a few thousand ids, read from L2), tests the rectangles,
// collects the (depth, id) keys in LDS, reserves its segment of the id list with one atomic, sorts and writes it
// (k_tile_pull).  Segments are placed in the order the tiles finish, which nothing depends on.  No per-tile
// counters, no scan over the tiles, no key array, no separate sort launch.  A tile with more than TPULL_KEYS near
// entries writes its ids unsorted and leaves an empty sorted head: its quadrants raise the lazy-sort flag at once
// and k_tile_tail sorts the segment (rare by construction of the cut).
#define TPULL_KEYS 2048u           // 16 KB of LDS: eight workgroups per CU, a frame's tiles in one round
#define SUPER_MAX GFT_SUPER_MAX     // supertiles at most (table in LDS and in the image buffer)

struct SuperArgs {
    int P, gx, gy, T;
    int sshift, sgx, NS;            // log2 of the supertile side in tiles, supertiles per row, supertile count
    const ushort4* __restrict__ rect;
    const float* __restrict__ depth;
    uint32_t cut_bits;
    uint32_t* st_cnt;               // [NS] entries per supertile
    uint32_t* st_start;             // [NS] first entry of every supertile's list
    uint32_t* st_cursor;            // [NS]
    uint32_t* st_inst;              // [NS] (Gaussian, tile) instances inside the supertile, then the first id-list slot of it
    uint32_t* st_icur;              // [NS] slots of the supertile's id-list region taken by its tiles
    uint64_t* sl_ent;               // entries grouped by supertile: id | rectangle relative to the supertile (4 x 5 bits) << 32
    uint32_t* ctrl;
    uint32_t* mail; uint32_t seq;
    uint32_t* dhist;
    DepthBins db;
    uint32_t target, target_min;
    uint32_t cap;
};

template <int PASS>      // 0: count (+ depth histogram, totals, mailbox), 1: scatter
__global__ __launch_bounds__(BIN_THREADS) void k_super_bin(SuperArgs a)
{
    __shared__ uint32_t s_cnt[SUPER_MAX];
    __shared__ uint32_t s_first[SUPER_MAX];         // pass 0: instances per supertile; pass 1: first entry of this workgroup's chunk
    __shared__ uint32_t s_dh[GFT_DHIST_BINS];
    __shared__ uint32_t s_last, s_near;
    const int tid = threadIdx.x;
    if (PASS == 1 && a.ctrl[GFT_CTRL_TOTAL] > a.cap) return;        // binning buffer too small: the host re-runs stage 2
    if (tid == 0) s_near = 0;
    if (tid < SUPER_MAX) { s_cnt[tid] = 0; s_first[tid] = 0; }
    if (PASS == 0 && tid < GFT_DHIST_BINS) s_dh[tid] = 0;
    __syncthreads();
    const int base = blockIdx.x * BIN_CHUNK;
    ushort4 r4[BIN_ITEMS];
    bool near[BIN_ITEMS];
#pragma unroll
    for (int u = 0; u < BIN_ITEMS; u++) {
        const int idx = base + u * BIN_THREADS + tid;
        const bool in = idx < a.P;
        r4[u] = in ? a.rect[idx] : make_ushort4(0, 0, 0, 0);
        const uint32_t d = in ? __float_as_uint(a.depth[idx]) : 0u;
        const uint32_t tiles = (uint32_t)(r4[u].z - r4[u].x) * (uint32_t)(r4[u].w - r4[u].y);
        near[u] = tiles != 0u && d <= a.cut_bits;
        if (PASS == 0 && tiles != 0u) atomicAdd(&s_dh[depth_bin(d, a.db)], tiles);
        if (PASS == 0 && near[u]) atomicAdd(&s_near, tiles);
    }
#pragma unroll
    for (int u = 0; u < BIN_ITEMS; u++) {
        if (!near[u]) continue;
        const int sx0 = r4[u].x >> a.sshift, sx1 = (r4[u].z - 1) >> a.sshift, sy0 = r4[u].y >> a.sshift, sy1 = (r4[u].w - 1) >> a.sshift;
        for (int sy = sy0; sy <= sy1; sy++)
            for (int sx = sx0; sx <= sx1; sx++) {
                atomicAdd(&s_cnt[sy * a.sgx + sx], 1u);
                if (PASS == 0) {
                    // tiles of the rectangle inside this supertile: its tiles reserve their id-list segments in a region
                    // of exactly that size (no frame-wide counter for 1200 workgroups to queue on)
                    const int S = 1 << a.sshift;
                    const int nx = min((int)r4[u].z, (sx + 1) * S) - max((int)r4[u].x, sx * S);
                    const int ny = min((int)r4[u].w, (sy + 1) * S) - max((int)r4[u].y, sy * S);
                    atomicAdd(&s_first[sy * a.sgx + sx], (uint32_t)(nx * ny));
                }
            }
    }
    __syncthreads();
    if (PASS == 0) {
        if (tid < a.NS) {
            const uint32_t c = s_cnt[tid];
            if (c) atomicAdd(&a.st_cnt[tid], c);
            const uint32_t ci = s_first[tid];
            if (ci) atomicAdd(&a.st_inst[tid], ci);
        }
        static_assert(SUPER_MAX <= BIN_THREADS, "one thread per supertile");
        if (tid < GFT_DHIST_BINS) {
            const uint32_t h = s_dh[tid];
            if (h) atomicAdd(&a.dhist[tid], h);
        }
        if (tid == 0 && s_near) atomicAdd(&a.ctrl[GFT_CTRL_NEARSUM], s_near);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) s_last = atomicAdd(&a.ctrl[GFT_CTRL_DONE], 1u) == gridDim.x - 1 ? 1u : 0u;
        __syncthreads();
        if (!s_last) return;
        // last workgroup: supertile list offsets, frame totals, next cut, mailbox
        if (tid < GFT_DHIST_BINS) s_dh[tid] = __hip_atomic_load(&a.dhist[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        {
            // exclusive scans of the supertile entry counts and instance counts: one thread per supertile
            __shared__ uint32_t s_wt[2][BIN_THREADS / 64];
            const int lane = tid & 63, wave = tid >> 6;
            const uint32_t v = tid < a.NS ? __hip_atomic_load(&a.st_cnt[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            const uint32_t vi = tid < a.NS ? __hip_atomic_load(&a.st_inst[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            uint32_t x = v, xi = vi;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_up(x, d, 64), yi = __shfl_up(xi, d, 64);
                if (lane >= d) { x += y; xi += yi; }
            }
            if (lane == 63) { s_wt[0][wave] = x; s_wt[1][wave] = xi; }
            __syncthreads();
            uint32_t woff = 0, woffi = 0;
            for (int w = 0; w < wave; w++) { woff += s_wt[0][w]; woffi += s_wt[1][w]; }
            if (tid < a.NS) {
                a.st_start[tid] = woff + x - v;
                a.st_cursor[tid] = 0;
                a.st_inst[tid] = woffi + xi - vi;       // from here on: first id-list slot of the supertile's region
                a.st_icur[tid] = 0;
            }
        }
        uint32_t R, cut_bin;
        dhist_scan(s_dh, a.target, R, cut_bin);
        if (tid == 0) {
            uint32_t cut_next = GFT_NO_CUT;
            if (cut_pays(R, a.target, a.target_min) && cut_bin + 1u < GFT_DHIST_BINS)
                cut_next = __float_as_uint(a.db.near_n * exp2f((float)(cut_bin + 1u) / a.db.scale));
            const uint32_t near_total = __hip_atomic_load(&a.ctrl[GFT_CTRL_NEARSUM], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.ctrl[GFT_CTRL_TOTAL] = R;
            a.ctrl[GFT_CTRL_TOTAL0] = near_total;      // instances of the near slab = sum of the supertile regions
            a.ctrl[GFT_CTRL_CUTNEXT] = cut_next;
            if (a.mail) {
                a.mail[GFT_CTRL_TOTAL] = R;
                a.mail[GFT_CTRL_MAXCNT] = 0u;
                a.mail[GFT_CTRL_TOTAL0] = near_total;
                a.mail[GFT_CTRL_CUTNEXT] = cut_next;
                __hip_atomic_store(&a.mail[GFT_CTRL_SEQ], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        return;
    }
    // scatter: one chunk per (workgroup, supertile)
    if (tid < a.NS) {
        const uint32_t c = s_cnt[tid];
        s_first[tid] = c ? a.st_start[tid] + atomicAdd(&a.st_cursor[tid], c) : 0u;
        s_cnt[tid] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < BIN_ITEMS; u++) {
        if (!near[u]) continue;
        const uint32_t idx = (uint32_t)(base + u * BIN_THREADS + tid);
        const int sx0 = r4[u].x >> a.sshift, sx1 = (r4[u].z - 1) >> a.sshift, sy0 = r4[u].y >> a.sshift, sy1 = (r4[u].w - 1) >> a.sshift;
        for (int sy = sy0; sy <= sy1; sy++)
            for (int sx = sx0; sx <= sx1; sx++) {
                const int q = sy * a.sgx + sx;
                // the rectangle clipped to this supertile, in tiles from its corner: x0, y0 in [0, S), x1, y1 in (0, S]
                const int ox = sx << a.sshift, oy = sy << a.sshift, S = 1 << a.sshift;
                const uint32_t x0 = (uint32_t)max((int)r4[u].x - ox, 0), x1 = (uint32_t)min((int)r4[u].z - ox, S);
                const uint32_t y0 = (uint32_t)max((int)r4[u].y - oy, 0), y1 = (uint32_t)min((int)r4[u].w - oy, S);
                const uint32_t rel = x0 | (y0 << 5) | (x1 << 10) | (y1 << 15);
                a.sl_ent[s_first[q] + atomicAdd(&s_cnt[q], 1u)] = ((uint64_t)rel << 32) | idx;
            }
    }
}

// 513 .. 1024 keys in sk[0, 1024) (pads = ~0): every wave sorts its quarter of 256 keys in registers, then the four
// sorted runs are merged by rank: a key's place in the list = its place in its own run + the number of smaller keys in
// each of the three other runs (binary searches in LDS; keys are distinct -- the id is their low half).  No merge
// network: 24 LDS reads per key instead of 19 workgroup-wide compare-exchange stages with a barrier each.
__device__ __forceinline__ void sort1024_by_rank_and_store(uint64_t* sk, uint32_t n, int tid, uint32_t* __restrict__ ids)
{
    const int lane = tid & 63, wave = tid >> 6;
    bitonic_blocked<2, 6, WaveSync>(sk + 256 * wave, lane, WaveSync());
    __syncthreads();
    auto at = [&](int run, uint32_t i) { return sk[256 * run + sort_slot(i)]; };     // key i of a sorted run
#pragma unroll
    for (int run = 0; run < 4; run++) {
        const uint64_t key = at(run, (uint32_t)tid);
        if (key == ~0ull) continue;                      // padding
        uint32_t rank = (uint32_t)tid;
#pragma unroll
        for (int o = 0; o < 4; o++) {
            if (o == run) continue;
            // number of keys of run o below `key`
            uint32_t lo = 0;
#pragma unroll
            for (uint32_t step = 128; step > 0; step >>= 1)
                if (at(o, lo + step - 1) < key) lo += step;
            if (at(o, lo) < key) lo++;                   // (lo <= 255 here)
            rank += lo;
        }
        ids[rank] = (uint32_t)key;
    }
    (void)n;
}

struct PullArgs {
    int gx, sshift, sgx;
    const ushort4* __restrict__ rect;
    const float* __restrict__ depth;
    const uint32_t* __restrict__ st_cnt;
    const uint32_t* __restrict__ st_start;
    const uint32_t* __restrict__ st_inst;       // first id-list slot of every supertile's region
    uint32_t* st_icur;
    const uint64_t* __restrict__ sl_ent;
    uint2* __restrict__ ranges;
    uint32_t* __restrict__ point_list;
    uint32_t* __restrict__ front_len;
    uint32_t* __restrict__ unit_flag;
    uint32_t* ctrl;
    uint32_t cap;
    float4* __restrict__ clear; size_t clear_vec4;
};

__global__ __launch_bounds__(GFT_BLOCK) void k_tile_pull(PullArgs a)
{
    __shared__ uint64_t sk[SORT_SLOTS(TPULL_KEYS)];
    __shared__ uint32_t s_n, s_start;
    if (a.ctrl[GFT_CTRL_TOTAL] > a.cap) return;
    // fire-and-forget zero fill of the backward's accumulator (as k_tile_front does)
    if (a.clear) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (size_t i = (size_t)blockIdx.x * GFT_BLOCK + threadIdx.x; i < a.clear_vec4; i += (size_t)gridDim.x * GFT_BLOCK)
            a.clear[i] = z;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int tile = blockIdx.x;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int q = (ty >> a.sshift) * a.sgx + (tx >> a.sshift);
    const uint32_t ln = a.st_cnt[q];
    const uint64_t* __restrict__ list = a.sl_ent + a.st_start[q];
    const uint32_t lx = (uint32_t)(tx & ((1 << a.sshift) - 1)), ly = (uint32_t)(ty & ((1 << a.sshift) - 1));
    if (tid == 0) s_n = 0;
    if (tid < 4) a.unit_flag[4 * tile + tid] = 0;
    __syncthreads();
    // pass over the supertile's list: keys of the Gaussians whose rectangle covers this tile
    auto scan = [&](bool store_ids, uint32_t seg) {
        for (uint32_t i0 = 0; i0 < ln; i0 += 4 * GFT_BLOCK) {
            uint32_t id4[4];
            uint64_t e4[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = i0 + u * GFT_BLOCK + tid;
                e4[u] = i < ln ? list[i] : 0ull;                     // (an all-zero rectangle covers no tile)
            }
            // hits of the four entries: one LDS atomic reserves the slots of all of them, the depth gathers of the
            // hits are issued together
            bool hit[4];
            unsigned long long hm[4];
            uint32_t off[4], cnt = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                id4[u] = (uint32_t)e4[u];
                const uint32_t rel = (uint32_t)(e4[u] >> 32);
                hit[u] = lx >= (rel & 31u) && lx < ((rel >> 10) & 31u) && ly >= ((rel >> 5) & 31u) && ly < ((rel >> 15) & 31u);
                hm[u] = __builtin_amdgcn_ballot_w64(hit[u]);
                off[u] = cnt;
                cnt += (uint32_t)__popcll(hm[u]);
            }
            if (cnt == 0u) continue;                                 // wave-uniform
            uint32_t hb = 0;
            if (lane == 0) hb = atomicAdd(&s_n, cnt);
            hb = (uint32_t)__builtin_amdgcn_readfirstlane((int)hb);
            uint32_t d4[4];
#pragma unroll
            for (int u = 0; u < 4; u++) d4[u] = (hit[u] && !store_ids) ? __float_as_uint(a.depth[id4[u]]) : 0u;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (!hit[u]) continue;
                const uint32_t pos = hb + off[u] + (uint32_t)__popcll(hm[u] & ((1ull << lane) - 1ull));
                if (store_ids) a.point_list[seg + pos] = id4[u];
                else if (pos < TPULL_KEYS) sk[pos] = ((uint64_t)d4[u] << 32) | id4[u];
            }
        }
    };
    scan(false, 0u);
    __syncthreads();
    const uint32_t n = s_n;
    __syncthreads();
    if (tid == 0) {
        // the tile's segment: inside its supertile's region, in the order its (at most S x S) tiles arrive
        s_start = n ? a.st_inst[q] + atomicAdd(&a.st_icur[q], n) : 0u;
        s_n = 0;
    }
    __syncthreads();
    const uint32_t start = s_start;
    if (tid == 0) {
        a.ranges[tile] = n ? make_uint2(start, start + n) : make_uint2(0u, 0u);
        a.front_len[tile] = n <= TPULL_KEYS ? n : 0u;         // longer: nothing sorted here, the tail sorter takes the segment
    }
    if (n == 0u) return;
    if (n > TPULL_KEYS) {
        scan(true, start);                                    // ids in arrival order
        return;
    }
    uint32_t* ids = a.point_list + start;
    if (n <= 1024u) {
        const uint32_t npad = next_pow2(n < 2u ? 2u : n);
        for (uint32_t i = tid + n; i < npad; i += GFT_BLOCK) sk[i] = ~0ull;
        __syncthreads();
        if (npad == 1024u) sort1024_by_rank_and_store(sk, n, tid, ids);
        else head_sort_and_store(sk, n, npad, tid, ids);
        return;
    }
    // 1025 .. 2048 keys: the register-blocked network wants them at their swizzled slots
    const uint32_t npad = 2048u;
    static_assert(TPULL_KEYS == 2048u, "k_tile_pull sorts at most 2048 keys");
    uint64_t mine[TPULL_KEYS / GFT_BLOCK];
#pragma unroll
    for (int k = 0; k < (int)(TPULL_KEYS / GFT_BLOCK); k++) {
        const uint32_t i = tid + k * GFT_BLOCK;
        mine[k] = i < n ? sk[i] : ~0ull;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < (int)(TPULL_KEYS / GFT_BLOCK); k++) {
        const uint32_t i = tid + k * GFT_BLOCK;
        if (i < npad) sk[sort_slot(i)] = mine[k];
    }
    __syncthreads();
    bitonic_blocked<3, 8>(sk, tid);
    for (uint32_t i = tid; i < n; i += GFT_BLOCK) ids[i] = (uint32_t)sk[sort_slot(i)];
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

// For every tile that has a flagged quadrant: sorts the unsorted tail of its near-slab id list (lazy sort, see
// k_tile_front) and, when the far slab was binned for it (lazy binning), its far-slab segment.
__global__ __launch_bounds__(SORT_BIG_THREADS) void k_tile_tail(int T, const uint2* __restrict__ ranges,
                                                                const uint2* __restrict__ ranges1, uint64_t* keys,
                                                                uint32_t* __restrict__ point_list,
                                                                const float* __restrict__ depth,
                                                                const uint32_t* __restrict__ front_len,
                                                                const uint32_t* __restrict__ unit_flag,
                                                                uint32_t* ctrl, uint32_t cap,
                                                                uint32_t* late_mail, uint32_t seq,
                                                                const uint32_t* __restrict__ quad_max, uint32_t* __restrict__ order)
{
    extern __shared__ uint64_t sk_dyn[];
    uint64_t* sk = sk_dyn;
    if (ctrl[GFT_CTRL_TOTAL] > cap) return;
    // No quadrant was flagged (the common case): the deepest contributors of the first render pass are final, and this
    // otherwise idle launch sorts the tiles by backward weight for k_render_bwd (no separate launch in the backward).
    if (order && blockIdx.x == 0 && ctrl[GFT_CTRL_NFLAG] == 0u) {
        gft_tile_order_block(T, quad_max, order);
        if (threadIdx.x == 0) ctrl[GFT_CTRL_ORDER_OK] = 1u;
    }
    // late report to the host mailbox (read at the caller's next forward, never waited for): how many quadrants outlived
    // what was sorted / binned up front -- the caller widens the near slab of the next frame when that happens
    if (late_mail && blockIdx.x == 0 && threadIdx.x == 0) {
        late_mail[GFT_CTRL_NFLAG] = ctrl[GFT_CTRL_NFLAG];
        __hip_atomic_store(&late_mail[GFT_CTRL_SEQ2], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (ctrl[GFT_CTRL_NFLAG] == 0u) return;
    const bool far = ctrl[GFT_CTRL_TOTAL1] != 0u;       // count pass 1 ran and found far-slab instances
    const int tid = threadIdx.x;
    for (int tile = blockIdx.x; tile < T; tile += gridDim.x) {
        const uint4 f = reinterpret_cast<const uint4*>(unit_flag)[tile];
        if ((f.x | f.y | f.z | f.w) == 0u) continue;           // uniform per workgroup
        const uint2 r = ranges[tile];
        const uint32_t kf = front_len[tile];
        tail_sort_segment<true>(r.x + kf, (r.y - r.x) - kf, keys, point_list, depth, sk, tid);
        if (far) {
            const uint2 r1 = ranges1[tile];
            tail_sort_segment<false>(r1.x, r1.y - r1.x, keys, point_list, depth, sk, tid);
        }
    }
}

}  // namespace

#define SORT_LDS_SMALL 4096u      // 32 KB of LDS
#define SORT_LDS_LARGE 16384u     // 128 KB of LDS (+ 4 KB of bank padding)
#define SORT_LDS_LARGE_BYTES (SORT_SLOTS(SORT_LDS_LARGE) * 8)

// wanted instances of the near slab (GFT_NEAR_SLAB_PER_TILE per tile; the environment variable of that name overrides
// it for tuning runs)
static uint32_t near_slab_target(int T, int per_tile_hint)
{
    static const uint32_t env_per_tile = [] {
        const char* e = getenv("GFT_NEAR_SLAB_PER_TILE");
        const long v = e ? atol(e) : 0;
        return v > 0 ? (uint32_t)v : 0u;
    }();
    const uint32_t per_tile = env_per_tile ? env_per_tile : (per_tile_hint > 0 ? (uint32_t)per_tile_hint : GFT_NEAR_SLAB_PER_TILE);
    const uint64_t t = (uint64_t)T * per_tile;
    return t > 0xfffffffeull ? 0xfffffffeu : (uint32_t)t;
}

// Gaussians per thread of the count / scatter workgroups.  Every workgroup pays for its passes over the tile table
// (zeroing, scan, one chunk reservation per tile); with a depth cut most Gaussians are skipped, so a workgroup takes
// more of them while about 200 workgroups remain (5 M Gaussians @ 1080p: 1221 -> 204 workgroups of 8160-entry tables).
static int bin_items(int P, uint32_t cut_bits, int pass)
{
    if (cut_bits == GFT_NO_CUT) return BIN_ITEMS;
    // (far pass: ~200 workgroups as well -- with 48 big ones an idle launch cost the same and a frame with flagged
    // quadrants 53 + 49 us instead of 18 + 23)
    const int by_blocks = P / (BIN_THREADS * 200);
    (void)pass;
    const int it = by_blocks < BIN_ITEMS ? BIN_ITEMS : (by_blocks > 60 ? 60 : by_blocks);      // (u16 per-tile counts per workgroup)
    return it / BIN_ITEMS * BIN_ITEMS;
}

hipError_t gft_launch_tile_count(hipStream_t s, const gft_config& c, const GeomView& g, const ImgView& im,
                                 uint32_t* mail, uint32_t seq, uint32_t cut_bits, int pass, uint32_t cap, int per_tile)
{
    const int gx = (c.W + GFT_TILE_X - 1) / GFT_TILE_X, gy = (c.H + GFT_TILE_Y - 1) / GFT_TILE_Y;
    const int T = gx * gy;
    CountArgs a;
    a.items = bin_items(c.P, cut_bits, pass);
    const int blocks = (c.P + BIN_THREADS * a.items - 1) / (BIN_THREADS * a.items);
    a.P = c.P; a.gx = gx; a.T = T;
    a.rect = g.rect; a.depth = g.depth; a.cut_bits = cut_bits;
    a.tile_cnt = pass == 0 ? im.tile_cnt : im.tile_cnt1;
    a.ranges = pass == 0 ? im.ranges : im.ranges1;
    a.cursor = im.tile_cursor; a.ctrl = im.ctrl; a.mail = mail; a.seq = seq;
    a.blockhist = (T <= BIN_LDS_MAX_TILES && T <= GFT_BLOCKHIST_TILES) ? g.blockhist : nullptr;
    a.dhist = im.dhist;
    // log-spaced depth bins between the camera's near and far planes (every visible Gaussian lies between them)
    const float nr = c.near_n > 1e-6f ? c.near_n : 1e-6f;
    const float fr = c.far_n > 2.0f * nr ? c.far_n : 2.0f * nr;
    a.db.near_n = nr; a.db.inv_near = 1.0f / nr; a.db.scale = (float)GFT_DHIST_BINS / log2f(fr / nr);
    a.target = near_slab_target(T, per_tile);
    a.target_min = near_slab_target(T, 0) < a.target ? near_slab_target(T, 0) : a.target;
    a.unit_flag = im.unit_flag; a.cap = cap;
    if (T <= BIN_LDS_MAX_TILES) {
        if (pass == 0) hipLaunchKernelGGL((k_tile_count<true, 0>), dim3(blocks), dim3(BIN_THREADS), (size_t)T * 4, s, a);
        else hipLaunchKernelGGL((k_tile_count<true, 1>), dim3(blocks), dim3(BIN_THREADS), (size_t)T * 4, s, a);
    } else {
        if (pass == 0) hipLaunchKernelGGL((k_tile_count<false, 0>), dim3(blocks), dim3(BIN_THREADS), 0, s, a);
        else hipLaunchKernelGGL((k_tile_count<false, 1>), dim3(blocks), dim3(BIN_THREADS), 0, s, a);
    }
    return hipGetLastError();
}

hipError_t gft_launch_tile_scatter(hipStream_t s, const gft_config& c, const GeomView& g, const ImgView& im,
                                   const BinView& b, uint32_t cap, uint32_t cut_bits, int pass, int64_t expect_total)
{
    const int gx = (c.W + GFT_TILE_X - 1) / GFT_TILE_X, gy = (c.H + GFT_TILE_Y - 1) / GFT_TILE_Y;
    const int T = gx * gy;
    ScatterArgs a;
    a.items = bin_items(c.P, cut_bits, pass);
    const int blocks = (c.P + BIN_THREADS * a.items - 1) / (BIN_THREADS * a.items);
    a.P = c.P; a.gx = gx; a.T = T;
    a.rect = g.rect; a.depth = g.depth;
    a.ranges = pass == 0 ? im.ranges : im.ranges1;
    a.cursor = im.tile_cursor; a.keys = b.keys; a.ctrl = im.ctrl; a.cap = cap; a.stage_cap = 0;
    a.blockhist = nullptr; a.cut_bits = cut_bits; a.unit_flag = im.unit_flag;
    if (T > BIN_LDS_MAX_TILES) {
        if (pass == 0) hipLaunchKernelGGL((k_tile_scatter<0, 0>), dim3(blocks), dim3(BIN_THREADS), 0, s, a);
        else hipLaunchKernelGGL((k_tile_scatter<0, 1>), dim3(blocks), dim3(BIN_THREADS), 0, s, a);
        return hipGetLastError();
    }
    {
        static std::atomic<uint64_t> done[4];
        hipError_t e = gft_lds_opt_in(reinterpret_cast<const void*>(&k_tile_scatter<1, 0>), 2 * BIN_LDS_MAX_TILES * 4, done[0]);
        if (e == hipSuccess) e = gft_lds_opt_in(reinterpret_cast<const void*>(&k_tile_scatter<2, 0>), BIN_STAGE_LDS_BYTES, done[1]);
        if (e == hipSuccess) e = gft_lds_opt_in(reinterpret_cast<const void*>(&k_tile_scatter<1, 1>), 2 * BIN_LDS_MAX_TILES * 4, done[2]);
        if (e == hipSuccess) e = gft_lds_opt_in(reinterpret_cast<const void*>(&k_tile_scatter<2, 1>), BIN_STAGE_LDS_BYTES, done[3]);
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
        if (pass == 0) hipLaunchKernelGGL((k_tile_scatter<2, 0>), dim3(blocks), dim3(BIN_THREADS), tables + stage_cap * 8, s, a);
        else hipLaunchKernelGGL((k_tile_scatter<2, 1>), dim3(blocks), dim3(BIN_THREADS), tables + stage_cap * 8, s, a);
    } else {
        if (pass == 0) hipLaunchKernelGGL((k_tile_scatter<1, 0>), dim3(blocks), dim3(BIN_THREADS), (size_t)T * 8, s, a);
        else hipLaunchKernelGGL((k_tile_scatter<1, 1>), dim3(blocks), dim3(BIN_THREADS), (size_t)T * 8, s, a);
    }
    return hipGetLastError();
}

// supertile side: the smallest power of two >= 2 tiles that leaves at most SUPER_MAX supertiles
static void super_shape(int gx, int gy, int& sshift, int& sgx, int& NS)
{
    sshift = 1;
    for (;;) {
        const int S = 1 << sshift;
        sgx = (gx + S - 1) / S;
        NS = sgx * ((gy + S - 1) / S);
        if (NS <= SUPER_MAX) return;
        sshift++;
    }
}

// tile-pull binning packs a rectangle relative to its supertile into 4 x 5 bits: supertiles of at most 16 x 16 tiles
bool gft_tile_pull_ok(const gft_config& c)
{
    const int gx = (c.W + GFT_TILE_X - 1) / GFT_TILE_X, gy = (c.H + GFT_TILE_Y - 1) / GFT_TILE_Y;
    int sshift, sgx, NS;
    super_shape(gx, gy, sshift, sgx, NS);
    return sshift <= 4;
}

// pass 0: count (+ depth histogram, totals, mailbox); pass 1: scatter of the near Gaussians' ids to their supertiles
hipError_t gft_launch_super_bin(hipStream_t s, const gft_config& c, const GeomView& g, const ImgView& im, const BinView& b,
                                uint32_t* mail, uint32_t seq, uint32_t cut_bits, int pass, uint32_t cap, int per_tile)
{
    const int gx = (c.W + GFT_TILE_X - 1) / GFT_TILE_X, gy = (c.H + GFT_TILE_Y - 1) / GFT_TILE_Y;
    SuperArgs a;
    a.P = c.P; a.gx = gx; a.gy = gy; a.T = gx * gy;
    super_shape(gx, gy, a.sshift, a.sgx, a.NS);
    a.rect = g.rect; a.depth = g.depth; a.cut_bits = cut_bits;
    a.st_cnt = im.super_tab; a.st_start = im.super_tab + SUPER_MAX; a.st_cursor = im.super_tab + 2 * SUPER_MAX;
    a.st_inst = im.super_tab + 3 * SUPER_MAX; a.st_icur = im.super_tab + 4 * SUPER_MAX;
    // the supertile lists live in the key array, which this path does not use otherwise (`cap` 8-byte entries; there
    // are at most as many (Gaussian, supertile) pairs as (Gaussian, tile) instances); the far pass reuses it later
    a.sl_ent = pass == 1 ? b.keys : nullptr;
    a.ctrl = im.ctrl; a.mail = mail; a.seq = seq; a.dhist = im.dhist;
    const float nr = c.near_n > 1e-6f ? c.near_n : 1e-6f;
    const float fr = c.far_n > 2.0f * nr ? c.far_n : 2.0f * nr;
    a.db.near_n = nr; a.db.inv_near = 1.0f / nr; a.db.scale = (float)GFT_DHIST_BINS / log2f(fr / nr);
    a.target = near_slab_target(a.T, per_tile);
    a.target_min = near_slab_target(a.T, 0) < a.target ? near_slab_target(a.T, 0) : a.target;
    a.cap = cap;
    const int blocks = (c.P + BIN_CHUNK - 1) / BIN_CHUNK;
    if (pass == 0) hipLaunchKernelGGL(k_super_bin<0>, dim3(blocks), dim3(BIN_THREADS), 0, s, a);
    else hipLaunchKernelGGL(k_super_bin<1>, dim3(blocks), dim3(BIN_THREADS), 0, s, a);
    return hipGetLastError();
}

hipError_t gft_launch_tile_pull(hipStream_t s, const gft_config& c, const GeomView& g, const ImgView& im, const BinView& b,
                                uint32_t cap, float* clear, size_t clear_bytes)
{
    const int gx = (c.W + GFT_TILE_X - 1) / GFT_TILE_X, gy = (c.H + GFT_TILE_Y - 1) / GFT_TILE_Y;
    PullArgs a;
    int NS;
    a.gx = gx;
    super_shape(gx, gy, a.sshift, a.sgx, NS);
    a.rect = g.rect; a.depth = g.depth;
    a.st_cnt = im.super_tab; a.st_start = im.super_tab + SUPER_MAX;
    a.st_inst = im.super_tab + 3 * SUPER_MAX; a.st_icur = im.super_tab + 4 * SUPER_MAX;
    a.sl_ent = b.keys;
    a.ranges = im.ranges; a.point_list = b.point_list; a.front_len = im.front_len; a.unit_flag = im.unit_flag;
    a.ctrl = im.ctrl; a.cap = cap;
    a.clear = reinterpret_cast<float4*>(clear); a.clear_vec4 = clear_bytes / 16;
    hipLaunchKernelGGL(k_tile_pull, dim3(gx * gy), dim3(GFT_BLOCK), 0, s, a);
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
                                const BinView& b, uint32_t cap, uint32_t* late_mail, uint32_t seq, bool want_order)
{
    const int gx = (c.W + GFT_TILE_X - 1) / GFT_TILE_X, gy = (c.H + GFT_TILE_Y - 1) / GFT_TILE_Y;
    const int T = gx * gy;
    {
        static std::atomic<uint64_t> done{0};
        const hipError_t e = gft_lds_opt_in(reinterpret_cast<const void*>(&k_tile_tail), SORT_LDS_LARGE_BYTES, done);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_tile_tail, dim3(T < 256 ? T : 256), dim3(SORT_BIG_THREADS), (size_t)SORT_LDS_LARGE_BYTES, s, T,
                       im.ranges, im.ranges1, b.keys, b.point_list, g.depth, im.front_len, im.unit_flag, im.ctrl, cap, late_mail, seq,
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
