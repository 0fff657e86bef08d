// k_pull.hip -- tile-pull instance binning (gfx950): what is binned, sorted and given an appearance is what
// can be seen, decided from THIS frame (a training loop draws another camera every iteration, train.py:155-163).
// What a camera's earlier frame may leave with the caller are SCHEDULES -- which tiles sort their whole list up front,
// in which order the blend's waves are dealt, where the supertile lists start -- that change when and where work is done,
// never a value: any contents of those words give the same results (gft_forward_io.tile_hints / tile_weights / cell_sched).
//
// A pixel stops reading its tile list once its transmittance is below 1e-4 (reference forward.cu:560-565), so
// in a dense frame most (Gaussian, tile) instances are never read (1 M metric frame: 3.6 M instances, 0.6 M up
// to the deepest contributor of every tile).  The reference duplicates every Gaussian into 64-bit (tile, depth)
// keys and radix-sorts all R of them (rasterizer_impl.cu:72-140,307-348).  Here:
//
//   k_super_bin<0/1> : every visible Gaussian is dealt to the SUPERTILES (S x S tiles) its rectangle touches: one
//                      8-byte entry `id | rectangle relative to the supertile (4 x 5 bits) | depth bin (12 bits)`
//                      per (Gaussian, supertile); count pass (+ R, mailbox), then scatter with one reserved chunk
//                      per (workgroup, supertile, depth slab)                      [8 B per entry, ~0.5 per instance]
//   k_super_bin<2>   : the scatter alone, appending to lists whose starts and capacities the camera's last frame left
//                      (+ R, mailbox, the checks that the lists fitted, the next frame's schedule); big tile grids group
//                      a workgroup's entries by list in LDS first (k_super_bin<1 / 2, true>)
//   k_tile_pull      : one workgroup per tile scans its supertile's entries (L2), histograms its hits over the
//                      depth bins, takes the nearest ~940 (whole bins) as the HEAD of its list: gathers their
//                      depths, places the (depth bits, id) keys in LDS GROUPED BY DEPTH BIN (the histogram's running
//                      count gives every bin its place: the bins' order is the sort's order) and finds a key's
//                      place inside its bin by looking through the bin's few keys -- a sorting network only when
//                      a bin is crowded --, writes the ids into the tile's head slot
//                      and marks those Gaussians as needed (appearance on demand, k_preprocess.hip; deriving the
//                      marks in that kernel from the tiles' cuts instead cost it more than the byte stores cost
//                      here: -6 / +11 us).  The rest of the list is never formed unless somebody walks that far.
//   k_tail_build     : for the tiles with a FLAGGED quadrant (walked its whole head with unsaturated pixels --
//                      the silhouette quadrants of a scene, whose pixels never saturate): pulls the rest of the
//                      list, CULLS it against the unsaturated pixels of the flagged quadrants (their bounding box
//                      rides in the flag word; most entries of a tile do not reach a given 8x8 quadrant, fewer
//                      still the band of pixels along a silhouette), gives the survivors their appearance, sorts
//                      them and writes head + culled tail as one list into the pool; the render kernel's resume pass and the backward walk
//                      that list.  Non-reaching entries contribute nothing to any pixel of those quadrants, so
//                      the blend is the same arithmetic sequence as over the reference's full list: images,
//                      pixel counts and gradients are bit-identical to whole-frame binning (k_binning.hip).
//
// Depth bins are a monotone integer function of the depth bits (no transcendental): keys of a lower bin are
// smaller than keys of a higher bin, so "all entries of the first bins, sorted" is a prefix of the reference's
// sorted list.
#include "gft_internal.h"
#include "gft_sort.h"
#include "gft_render_walk.h"     // (in front of the next one: the blend walk is compiled with contraction on, as in k_render.hip)
#include "gft_appearance.h"      // (floating-point contraction is off from here on)

#include <cstdlib>
#include <cstring>

// Timing experiments on the phases of k_tile_pull / k_tail_build (profiles/pull_phases.py): only in an experiment build
// (python -m gftorf_amd.build --tag phases -DGFT_PHASE_DBG=1; GFT_PULL_DBG / GFT_TAIL_DBG in the environment then truncate the
// kernels -- results are NOT valid).  The product library has no such switch: the expressions below are the constant 0.
#ifdef GFT_PHASE_DBG
#define PHASE_DBG(a) ((a).dbg)
#else
#define PHASE_DBG(a) 0
#endif

namespace {

#define TPULL_KEYS GFT_HEAD_SLOT    // keys of a head / a chunk that may have to go through a sorting network (16 KB of LDS)
#define TPULL_KEYS_BIG 3328u        // keys of a chunk of a whole list whose bins are all small (placed by cursors, never by a network): 26 KB
                                    // (five workgroups per CU is what the kernel's registers allow: 5 x 30.8 KB of LDS fit the CU's 160 KB)
#define HEAD_DIRECT GFT_HEAD_DIRECT  // lists (scanned hits) up to this length are sorted whole: a sparse frame (the reference's
                                    // 100 k Gaussians at 320x240: ~1450 instances per tile, low opacities) saturates nowhere, every
                                    // quadrant of a tile with a tail would flag and every list be completed in a second pass
#define PULL_GROUP_MAX 31u          // largest depth bin of a head whose keys are still ordered by looking through their bin (5-bit count)
#define PULL_WINDOW 2048u           // depth bins, from the tile's first occupied one, that have a cursor
#define HEAD_TARGET GFT_HEAD_TARGET  // wanted length of the sorted head of a longer list
#define TAIL_LDS_KEYS 4096u         // culled tails are sorted in LDS in runs of at most this many keys (whole depth bins)
#define TAIL_ITEMS 4                  // entries per thread and scan trip (8: no faster, 196 registers)
#define TAIL_THREADS 512              // (1024 threads leave 128 registers per lane: the appearance evaluation then spills)

// entry = id | rel << 32 | bin << 52;  rel = x0 | y0 << 5 | x1 << 10 | y1 << 15 (tiles from the supertile's corner)
__device__ __forceinline__ bool entry_hits(uint64_t e, uint32_t lx, uint32_t ly)
{
    const uint32_t rel = (uint32_t)(e >> 32);
    return lx >= (rel & 31u) && lx < ((rel >> 10) & 31u) && ly >= ((rel >> 5) & 31u) && ly < ((rel >> 15) & 31u);
}
__device__ __forceinline__ uint32_t entry_bin(uint64_t e) { return (uint32_t)(e >> 52); }

struct SuperArgs {
    int P;
    SuperShape sh;
    const ushort4* __restrict__ rect;
    const float* __restrict__ depth;
    // Every (supertile, slab) counter exists in `copies` copies (power of two, <= 8): workgroup b adds to copy b & (copies - 1),
    // so a counter takes an eighth of the same-address atomics -- 1000 workgroups x 300 cells queued up on 300 addresses,
    // and the scatter pass's reservations are RETURNING atomics, each waiting for the ones in front of it.
    uint32_t* cnt_copy;             // [copies][NS * K] entries per (copy, supertile, slab)         (zero at kernel start)
    uint32_t* cur_copy;             // [copies][NS * K] next free entry of the copy's part of the list (absolute)
    uint32_t* st_cnt;               // [NS * K] entries per (supertile, slab)
    uint32_t* st_start;             // [NS * K] first entry of every list
    int copies;
    uint64_t* sl_ent;               // entries grouped by (supertile, slab)
    uint32_t* ctrl;
    uint32_t* mail; uint32_t seq;
    uint32_t cap;
    uint32_t post;                  // `mail` is a status block of gft_forward_enqueue: its binning_instances + 1 (0: a mailbox slot)
    const uint32_t* __restrict__ hints;      // pass 0: the caller's per-tile schedule (may be NULL): its non-zero words are counted for the host
    // pass 1 on big tile grids: the workgroup's entries are grouped by cell in LDS first (`stage_cap` of them: 8 bytes + a
    // 16-bit cell each behind the three cell tables) and leave as runs of consecutive lanes.  0: every entry is written
    // from where it was made.
    uint32_t stage_cap;
    // The caller's list schedule of this camera (gft_forward_io.cell_sched; may be NULL): start[cells] | capacity[cells] |
    // {cells, valid}.  Pass 0 (and pass 2) leave the next frame's there: every list's count of this frame plus a quarter
    // plus 64, starts = their prefix sums.  Pass 2 = the scatter WITHOUT a count pass in front of it: entries are appended to
    // the lists where the schedule puts them; a list that outgrew its capacity (or a schedule that is not one) makes the
    // frame's binning invalid -- ctrl[TOTAL] = 0xffffffff, every later kernel of the frame returns, the host runs the
    // counted flow.  Whatever the words hold, nothing is written outside the entry array and nothing wrong is rendered.
    uint32_t* sched;
};

// next frame's schedule from this frame's list counts (all threads of one workgroup; counts read with device-scope loads)
__device__ inline void super_write_sched(uint32_t* sched, const uint32_t* counts, int cells, uint32_t* s_wt, uint32_t* s_carry)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PER = 4;
    if (tid == 0) *s_carry = 0;
    __syncthreads();
    for (int c0 = 0; c0 < cells; c0 += PER * BIN_THREADS) {
        const int first = c0 + tid * PER;
        uint32_t cap[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const uint32_t c = first + k < cells ? __hip_atomic_load(&counts[first + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            cap[k] = first + k < cells ? c + (c >> 2) + 64u : 0u;
            sum += cap[k];
        }
        uint32_t x = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) s_wt[wave] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; w++) woff += s_wt[w];
        const uint32_t carry = *s_carry;
        uint32_t run = carry + woff + x - sum;
#pragma unroll
        for (int k = 0; k < PER; k++)
            if (first + k < cells) { sched[first + k] = run; sched[cells + first + k] = cap[k]; run += cap[k]; }
        __syncthreads();
        if (tid == BIN_THREADS - 1) *s_carry = carry + woff + x;
        __syncthreads();
    }
    if (tid == 0) { sched[2 * cells] = (uint32_t)cells; sched[2 * cells + 1] = 1u; }
}

template <int PASS, bool STAGED = false>      // 0: count (+ R, mailbox), 1: scatter (STAGED: through LDS, big tile grids), 2: scatter by the caller's schedule, no count pass
__global__ __launch_bounds__(BIN_THREADS) void k_super_bin(SuperArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t sb_dyn[];
    const int cells = a.sh.NS * a.sh.K;
    uint32_t* s_cnt = sb_dyn;                 // [cells]
    uint32_t* s_first = sb_dyn + cells;       // [cells] pass 1: first entry of this workgroup's chunk
    uint32_t* s_loc = sb_dyn + 2 * cells;     // [cells] pass 1, staged: first staging slot of the cell
    __shared__ uint32_t s_last, s_sum, s_carry;
    __shared__ uint32_t s_wt[BIN_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (PASS == 1 && a.ctrl[GFT_CTRL_TOTAL] > a.cap) return;        // binning buffer too small: the host re-runs stage 2
    for (int i = tid; i < cells; i += BIN_THREADS) s_cnt[i] = 0;
    if (tid == 0) s_sum = 0;
    __syncthreads();
    if (PASS == 2 && blockIdx.x == 0 && a.hints) {
        // (pass 2, on the side and up front: the schedule's marked tiles for the host -- the workgroup that closes the pass has
        // the frame's critical path behind it)
        uint32_t nh = 0;
        for (int t = tid; t < a.sh.T; t += BIN_THREADS) nh += a.hints[t] != 0u ? 1u : 0u;
        nh = gft_wave_sum_u32_to_lane63(nh);
        if (lane == 63 && nh) atomicAdd(&a.ctrl[GFT_CTRL_NHINT], nh);
    }
    const int base = blockIdx.x * BIN_CHUNK;
    const int K = a.sh.K, sgx = a.sh.sgx, ss = a.sh.sshift;
    ushort4 r4[BIN_ITEMS];
    uint32_t bin[BIN_ITEMS];
    uint32_t mine = 0;
    // All loads of the thread's items first, unconditionally (index clamped to the last Gaussian), then the selects: a load
    // under a lane condition ends in a wait for everything outstanding where its branch joins, so `in ? load : 0` per
    // item made this prologue eight memory round trips in a row instead of one.
    uint32_t dbits[BIN_ITEMS];
#pragma unroll
    for (int u = 0; u < BIN_ITEMS; u++) {
        const int idx = min(base + u * BIN_THREADS + tid, a.P - 1);
        r4[u] = a.rect[idx];
        dbits[u] = __float_as_uint(a.depth[idx]);      // (not written for culled Gaussians and not used for them either)
    }
#pragma unroll
    for (int u = 0; u < BIN_ITEMS; u++) {
        const bool in = base + u * BIN_THREADS + tid < a.P;
        if (!in) r4[u] = make_ushort4(0, 0, 0, 0);
        const uint32_t tiles = (uint32_t)(r4[u].z - r4[u].x) * (uint32_t)(r4[u].w - r4[u].y);
        bin[u] = tiles ? gft_depth_bin(dbits[u], a.sh.near_bits, a.sh.bin_shift) : 0u;
        mine += tiles;
    }
    if (PASS != 1) {
        mine = gft_wave_sum_u32_to_lane63(mine);
        if (lane == 63 && mine) atomicAdd(&s_sum, mine);
    }
#pragma unroll
    for (int u = 0; u < BIN_ITEMS; u++) {
        if (!(r4[u].z > r4[u].x && r4[u].w > r4[u].y)) continue;
        const int sx0 = r4[u].x >> ss, sx1 = (r4[u].z - 1) >> ss, sy0 = r4[u].y >> ss, sy1 = (r4[u].w - 1) >> ss;
        const int slab = (int)(bin[u] >> a.sh.kshift);
        for (int sy = sy0; sy <= sy1; sy++)
            for (int sx = sx0; sx <= sx1; sx++) atomicAdd(&s_cnt[(sy * sgx + sx) * K + slab], 1u);
    }
    __syncthreads();
    if (PASS == 0) {
        // (every workgroup walks the table from another start: they do not queue up on the same counters)
        const int rot = (int)(((uint64_t)blockIdx.x * (uint64_t)cells) / gridDim.x);
        for (int i = tid; i < cells; i += BIN_THREADS) {
            const int cidx = i + rot >= cells ? i + rot - cells : i + rot;
            const uint32_t c = s_cnt[cidx];
            if (c) atomicAdd(&a.cnt_copy[(size_t)(blockIdx.x & (a.copies - 1)) * cells + cidx], c);
        }
        if (tid == 0 && s_sum) atomicAdd(&a.ctrl[GFT_CTRL_RSUM], s_sum);
        // The workgroup that draws the last ticket scans.  Every counter update above is a device-scope atomic, complete
        // once vmcnt drains, and the scan reads the counters with device-scope loads: no cache write-back is needed.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            // two-level ticket (GFT_TICKET_WORDS): the last workgroup of a group draws from the second level
            const uint32_t G = min((uint32_t)GFT_TICKET_WORDS, gridDim.x), grp = blockIdx.x % G;
            const uint32_t members = (gridDim.x - grp + G - 1u) / G;
            uint32_t last = 0u;
            if (atomicAdd(&a.ctrl[GFT_CTRL_WORDS + grp], 1u) == members - 1u)
                last = atomicAdd(&a.ctrl[GFT_CTRL_DONE], 1u) == G - 1u ? 1u : 0u;
            s_last = last;
        }
        __syncthreads();
        if (!s_last) return;
        // last workgroup: exclusive scan of the entry counts -> list starts (and, inside a list, the start of every copy's
        // part); frame totals -> ctrl + host mailbox.  A thread takes PER consecutive cells; all its counter loads
        // (copies x PER) are issued together: one memory round trip per 16 cells and thread.
        if (tid == 0) s_carry = 0;
        __syncthreads();
        const int C = a.copies;
        constexpr int PER = 4;                               // consecutive cells per thread and trip
        for (int c0 = 0; c0 < cells; c0 += PER * BIN_THREADS) {
            const int first = c0 + tid * PER;
            uint32_t v[8][PER];                              // [copy][cell]: all loads of the trip in flight together
#pragma unroll
            for (int cp = 0; cp < 8; cp++)
#pragma unroll
                for (int k = 0; k < PER; k++)
                    v[cp][k] = __hip_atomic_load(&a.cnt_copy[(size_t)min(cp, C - 1) * cells + min(first + k, cells - 1)],
                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t tot[PER], mine_sum = 0;
#pragma unroll
            for (int k = 0; k < PER; k++) {
                tot[k] = 0;
#pragma unroll
                for (int cp = 0; cp < 8; cp++) {
                    if (cp >= C || first + k >= cells) v[cp][k] = 0u;
                    tot[k] += v[cp][k];
                }
                mine_sum += tot[k];
            }
            uint32_t x = mine_sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_up(x, d, 64);
                if (lane >= d) x += y;
            }
            if (lane == 63) s_wt[wave] = x;
            __syncthreads();
            uint32_t woff = 0;
            for (int w = 0; w < wave; w++) woff += s_wt[w];
            const uint32_t carry = s_carry;
            uint32_t run = carry + woff + x - mine_sum;       // first entry of this thread's first cell
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const int cell = first + k;
                if (cell < cells) {
                    a.st_cnt[cell] = tot[k];
                    a.st_start[cell] = run;
                    uint32_t inner = run;
#pragma unroll
                    for (int cp = 0; cp < 8; cp++) {
                        if (cp < C) a.cur_copy[(size_t)cp * cells + cell] = inner;
                        inner += v[cp][k];
                    }
                }
                run += tot[k];
            }
            __syncthreads();
            if (tid == BIN_THREADS - 1) s_carry = carry + woff + x;
            __syncthreads();
        }
        const uint32_t entries_total = s_carry;
        if (a.sched) super_write_sched(a.sched, a.st_cnt, cells, s_wt, &s_carry);
        // tiles the schedule marks (as the previous frame of the shape left it): the host picks the next frame's build of
        // k_tile_pull by it
        uint32_t nh = 0;
        if (a.hints) {
            for (int t = tid; t < a.sh.T; t += BIN_THREADS) nh += a.hints[t] != 0u ? 1u : 0u;
            nh = gft_wave_sum_u32_to_lane63(nh);
            if (tid == 0) s_sum = 0;
            __syncthreads();
            if (lane == 63 && nh) atomicAdd(&s_sum, nh);
            __syncthreads();
        }
        if (tid == 0) {
            const uint32_t R = __hip_atomic_load(&a.ctrl[GFT_CTRL_RSUM], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.ctrl[GFT_CTRL_TOTAL] = R;
            a.ctrl[GFT_CTRL_ENTRIES] = entries_total;
            a.ctrl[GFT_CTRL_MAXCNT] = 0u;              // (the tile lists are never formed)
            a.ctrl[GFT_CTRL_NHINT] = a.hints ? s_sum : 0u;
            if (a.mail) {
                a.mail[GFT_CTRL_TOTAL] = R;            // (GFT_CTRL_FLAGS of the slot belongs to the preprocess kernel)
                a.mail[GFT_CTRL_MAXCNT] = 0u;
                a.mail[GFT_CTRL_ENTRIES] = entries_total;
                a.mail[GFT_CTRL_NHINT] = a.hints ? s_sum : 0u;
                gft_status_sticky(a.mail, a.post, R);
                __hip_atomic_store(&a.mail[GFT_CTRL_SEQ], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        return;
    }
    // scatter: one chunk per (workgroup, supertile, slab)
    // Staged (big tile grids: 2040 cells at 1080p, five or six entries per workgroup and cell): made where its Gaussian
    // sits, an entry is an 8-byte store of its own -- 64 lanes, 64 cache lines, each leaving the L2 as a 32-byte sector
    // (5 M @ 1080p: 441 MB written for 107 MB of entries, the kernel stalled on its store queue 0.65 of the time).
    // Grouped by cell in LDS first, the entries of a cell leave as one run of consecutive lanes.
    uint32_t* s_stage_cell_base = sb_dyn + 3 * cells;                  // staging area: entries, then their cells
    uint64_t* stage_ent = reinterpret_cast<uint64_t*>(s_stage_cell_base + (cells & 1));
    uint16_t* stage_cell = reinterpret_cast<uint16_t*>(stage_ent + a.stage_cap);
    uint32_t n_wg = 0;
    if (STAGED) {
        // exclusive scan of the cell counts (PERC consecutive cells per thread)
        const int PERC = (cells + BIN_THREADS - 1) / BIN_THREADS;
        const int c0 = tid * PERC;
        uint32_t sum = 0;
        for (int k = 0; k < PERC; k++) sum += c0 + k < cells ? s_cnt[c0 + k] : 0u;
        uint32_t x = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) s_wt[wave] = x;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (int w = 0; w < BIN_THREADS / 64; w++) {
            if (w < wave) woff += s_wt[w];
            tot += s_wt[w];
        }
        uint32_t run = woff + x - sum;
        for (int k = 0; k < PERC; k++)
            if (c0 + k < cells) { s_loc[c0 + k] = run; run += s_cnt[c0 + k]; }
        n_wg = min(tot, a.stage_cap);                                // (entries past the staging area go the direct way)
        __syncthreads();
    }
    for (int i = tid; i < cells; i += BIN_THREADS) {
        const uint32_t c = s_cnt[i];
        if (PASS == 2) {
            // by the caller's schedule: the list of cell i starts at sched[i] and holds sched[cells + i] entries; this
            // workgroup's chunk follows what the others have taken so far (plane 0 of the counters = the lists' lengths of
            // this frame).  A chunk that does not fit is not written at all: the frame is rendered by the counted flow then.
            uint32_t first = 0xffffffffu;
            if (c) {
                const uint32_t rel = atomicAdd(&a.cnt_copy[i], c);
                const uint64_t st = a.sched[i], room = a.sched[cells + i];
                if ((uint64_t)rel + c <= room && st + rel + c <= (uint64_t)a.cap) first = (uint32_t)(st + rel);
            }
            s_first[i] = first;
        } else {
            s_first[i] = c ? atomicAdd(&a.cur_copy[(size_t)(blockIdx.x & (a.copies - 1)) * cells + i], c) : 0u;
        }
        s_cnt[i] = 0;
    }
    if (PASS == 2) {
        // The ticket is drawn HERE, not behind the entries: what the closing workgroup needs -- every list's length, the
        // instance sum -- is final once every workgroup has reserved (the returning atomics above have returned), and its
        // serial tail then runs beside the other workgroups' stores instead of behind them.
        if (tid == 0 && s_sum) atomicAdd(&a.ctrl[GFT_CTRL_RSUM], s_sum);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const uint32_t G = min((uint32_t)GFT_TICKET_WORDS, gridDim.x), grp = blockIdx.x % G;
            const uint32_t members = (gridDim.x - grp + G - 1u) / G;
            uint32_t last = 0u;
            if (atomicAdd(&a.ctrl[GFT_CTRL_WORDS + grp], 1u) == members - 1u)
                last = atomicAdd(&a.ctrl[GFT_CTRL_DONE], 1u) == G - 1u ? 1u : 0u;
            s_last = last;
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < BIN_ITEMS; u++) {
        if (!(r4[u].z > r4[u].x && r4[u].w > r4[u].y)) continue;
        const uint32_t idx = (uint32_t)(base + u * BIN_THREADS + tid);
        const int sx0 = r4[u].x >> ss, sx1 = (r4[u].z - 1) >> ss, sy0 = r4[u].y >> ss, sy1 = (r4[u].w - 1) >> ss;
        const int slab = (int)(bin[u] >> a.sh.kshift);
        for (int sy = sy0; sy <= sy1; sy++)
            for (int sx = sx0; sx <= sx1; sx++) {
                const int cell = (sy * sgx + sx) * K + slab;
                // the rectangle clipped to this supertile, in tiles from its corner: x0, y0 in [0, S), x1, y1 in (0, S]
                const int ox = sx << ss, oy = sy << ss, S = 1 << ss;
                const uint32_t x0 = (uint32_t)max((int)r4[u].x - ox, 0), x1 = (uint32_t)min((int)r4[u].z - ox, S);
                const uint32_t y0 = (uint32_t)max((int)r4[u].y - oy, 0), y1 = (uint32_t)min((int)r4[u].w - oy, S);
                const uint32_t hi = x0 | (y0 << 5) | (x1 << 10) | (y1 << 15) | (bin[u] << 20);
                const uint32_t rank = atomicAdd(&s_cnt[cell], 1u);
                const uint64_t e = ((uint64_t)hi << 32) | idx;
                const uint32_t pos = STAGED ? s_loc[cell] + rank : ~0u;
                if (STAGED && pos < a.stage_cap) {
                    stage_ent[pos] = e;
                    stage_cell[pos] = (uint16_t)cell;
                } else if (PASS != 2 || s_first[cell] != 0xffffffffu) {
                    a.sl_ent[s_first[cell] + rank] = e;
                }
            }
    }
    if (STAGED) {
        __syncthreads();
        for (uint32_t i = (uint32_t)tid; i < n_wg; i += BIN_THREADS) {
            const uint32_t cell = stage_cell[i];
            if (PASS != 2 || s_first[cell] != 0xffffffffu) a.sl_ent[s_first[cell] + (i - s_loc[cell])] = stage_ent[i];
        }
    }
    if (PASS != 2) return;
    // ---- pass 2: the workgroup that drew the last ticket closes the frame's binning front end (what the count pass's does)
    if (!s_last) return;
    __syncthreads();
    if (tid == 0) {
        s_sum = 0;                 // from here: the frame's entries
        s_last = 1;                // ... and "the schedule was one and every list fitted" (1: so far)
        s_carry = 0;               // ... and the running start of the next frame's lists
    }
    __syncthreads();
    // One walk over the lists (a thread takes PER consecutive ones, all its loads together): this frame's list tables, the
    // checks, and the next frame's schedule written over this one -- a list's old words are read by its own thread and (its
    // start) by the thread in front of it, both before the barrier that precedes the writes.
    {
        constexpr int PER = 4;
        bool ok = a.sched[2 * cells] == (uint32_t)cells && a.sched[2 * cells + 1] == 1u;
        uint32_t ent = 0;
        for (int c0 = 0; c0 < cells; c0 += PER * BIN_THREADS) {
            const int first = c0 + tid * PER;
            uint32_t c[PER], ncap[PER], sum = 0;
            uint64_t st[PER + 1], room[PER];
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const int cell = min(first + k, cells - 1);
                c[k] = __hip_atomic_load(&a.cnt_copy[cell], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                st[k] = a.sched[cell];
                room[k] = a.sched[cells + cell];
            }
            st[PER] = first + PER < cells ? (uint64_t)a.sched[first + PER] : (uint64_t)a.cap;
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const bool in = first + k < cells;
                const uint64_t nxt = first + k + 1 < cells ? st[k + 1] : (uint64_t)a.cap;
                if (in) {
                    ok = ok && c[k] <= room[k] && st[k] + room[k] <= nxt;
                    a.st_cnt[first + k] = c[k];
                    a.st_start[first + k] = (uint32_t)st[k];
                    ent += c[k];
                }
                ncap[k] = in ? c[k] + (c[k] >> 2) + 64u : 0u;
                sum += ncap[k];
            }
            uint32_t x = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_up(x, d, 64);
                if (lane >= d) x += y;
            }
            if (lane == 63) s_wt[wave] = x;
            __syncthreads();
            uint32_t woff = 0;
            for (int w = 0; w < wave; w++) woff += s_wt[w];
            const uint32_t carry = s_carry;
            uint32_t run = carry + woff + x - sum;
#pragma unroll
            for (int k = 0; k < PER; k++)
                if (first + k < cells) { a.sched[first + k] = run; a.sched[cells + first + k] = ncap[k]; run += ncap[k]; }
            __syncthreads();
            if (tid == BIN_THREADS - 1) s_carry = carry + woff + x;
            __syncthreads();
        }
        ent = gft_wave_sum_u32_to_lane63(ent);
        if (lane == 63 && ent) atomicAdd(&s_sum, ent);
        if (!ok) s_last = 0;       // (any lane)
        if (tid == 0) { a.sched[2 * cells] = (uint32_t)cells; a.sched[2 * cells + 1] = 1u; }
        __syncthreads();
    }
    const uint32_t entries_total = s_sum;
    const bool fits = s_last != 0u;
    if (tid == 0) {
        const uint32_t R = __hip_atomic_load(&a.ctrl[GFT_CTRL_RSUM], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (workgroup 0 added them before it drew its ticket)
        const uint32_t nhint = __hip_atomic_load(&a.ctrl[GFT_CTRL_NHINT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.ctrl[GFT_CTRL_TOTAL] = fits ? R : 0xffffffffu;       // (not binned: every later kernel of the frame returns on it)
        a.ctrl[GFT_CTRL_ENTRIES] = entries_total;
        a.ctrl[GFT_CTRL_MAXCNT] = 0u;
        if (a.mail) {
            a.mail[GFT_CTRL_TOTAL] = R;
            a.mail[GFT_CTRL_MAXCNT] = 0u;
            a.mail[GFT_CTRL_ENTRIES] = entries_total;
            a.mail[GFT_CTRL_NHINT] = nhint;
            if (!fits) __hip_atomic_fetch_or(&a.mail[GFT_CTRL_FLAGS], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&a.mail[GFT_CTRL_SEQ], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

struct PullArgs {
    SuperShape sh;
    const float* __restrict__ depth;
    const uint32_t* __restrict__ st_cnt;
    const uint32_t* __restrict__ st_start;
    const uint64_t* __restrict__ sl_ent;
    uint2* __restrict__ ranges;
    uint32_t* __restrict__ heads;            // point_list: head slot of tile t at [t * GFT_HEAD_SLOT, ...)
    uint32_t* __restrict__ front_len;
    uint32_t* __restrict__ unit_flag;
    uint32_t* __restrict__ tile_cnt;
    uint32_t* __restrict__ tile_cut;
    uint8_t* __restrict__ need;
    uint32_t* ctrl;
    uint32_t cap;
    float4* __restrict__ clear; size_t clear_vec4;
    // Schedule from the caller's previous frame of this shape (gft_forward_io.tile_hints; NULL: none): a non-zero word = some
    // quadrant of the tile walked past where a sorted head would have ended.  Such a tile sorts its WHOLE list here, in
    // chunks of whole depth bins, instead of a head now and the rest through flag -> k_tail_build -> resume pass.  What
    // the hint says never changes a result: the blend walks the same entries in the same order either way.
    const uint32_t* __restrict__ hints;
    uint32_t pool_base;                      // first pool slot inside point_list (= T * GFT_HEAD_SLOT)
    int dbg;
};

// block -> tile: blocks are dealt round-robin to the 8 XCDs; every XCD gets one contiguous run of the tiles in
// supertile-major order, so the S x S tiles that scan the same entry lists run on one XCD (one L2) at about the same time
__device__ __forceinline__ bool pull_tile_of_block(const SuperShape& sh, int b, int& tile, int& q, uint32_t& lx, uint32_t& ly)
{
    const int S2 = 1 << (2 * sh.sshift);
    const int Np = sh.NS * S2;
    const int chunk = (Np + 7) >> 3;
    const int pos = (b & 7) * chunk + (b >> 3);
    if (pos >= Np) return false;
    q = pos >> (2 * sh.sshift);
    const int local = pos & (S2 - 1);
    lx = (uint32_t)(local & ((1 << sh.sshift) - 1));
    ly = (uint32_t)(local >> sh.sshift);
    const int tx = ((q % sh.sgx) << sh.sshift) + (int)lx, ty = ((q / sh.sgx) << sh.sshift) + (int)ly;
    if (tx >= sh.gx || ty >= sh.gy) return false;
    tile = ty * sh.gx + tx;
    return true;
}

// Two builds of one kernel.  WHOLE = false: heads only (no schedule is read): 20.5 KB of LDS and few registers, seven
// workgroups per CU -- what a frame without hinted tiles runs (the 8160 tiles of a 1080p frame take 184 us with it, 220 with
// the other build).  WHOLE = true: hinted tiles sort their whole lists, in chunks: 30.8 KB, 96 registers (five waves per
// SIMD keep a 1200-tile frame resident in one round; left alone the allocator takes 97: four).  Which one a frame runs is
// the caller's schedule (gft_forward_hints.whole_lists, from the count of hinted tiles the previous frame reported).
template <bool WHOLE>
__global__ __launch_bounds__(GFT_BLOCK) __attribute__((amdgpu_waves_per_eu(WHOLE ? 5 : 7, 8))) void k_tile_pull(PullArgs a)
{
    // the depth-bin histogram of pass A and the keys of pass B share 16 KB of LDS (the histogram is done with once the head
    // is chosen): nine workgroups per CU instead of four -- a frame's tiles are resident in one round
    __shared__ uint64_t sk[SORT_SLOTS(WHOLE ? TPULL_KEYS_BIG : TPULL_KEYS)];
    static_assert(sizeof(uint64_t) * SORT_SLOTS(TPULL_KEYS) >= sizeof(uint32_t) * GFT_DEPTH_BINS, "histogram fits the key buffer");
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(sk);
    __shared__ uint32_t s_n, s_cut, s_kf, s_gmax, s_gmax2, s_bmin, s_bmax, s_pool;
    __shared__ uint32_t s_wt[GFT_BLOCK / 64];
    // 16-bit cursors, one per depth bin of a window of PULL_WINDOW bins from the first occupied one (two per word):
    // place of the bin's next key (11 bits; 12 in the big chunks of a whole list) | keys in the bin (5 bits; 4).  Pass B
    // places the keys grouped by
    // bin: the order of the bins IS the order of the sort, what is left is the order inside a bin.
    __shared__ uint32_t s_cur[PULL_WINDOW / 2];
    if (a.ctrl[GFT_CTRL_TOTAL] > a.cap) return;
    // Fire-and-forget zero fill of the backward's accumulator (64 B per Gaussian).  Issued where only LDS work and stores
    // follow (in front of the sort): loads and stores count down one in-order counter, so a load behind these stores
    // waits until HBM has taken all of them -- at the top of the kernel that stalled the first list read of every
    // workgroup behind 64 MB of writes.
    auto clear_slice = [&]() {
        if (!a.clear) return;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (size_t i = (size_t)blockIdx.x * GFT_BLOCK + threadIdx.x; i < a.clear_vec4; i += (size_t)gridDim.x * GFT_BLOCK)
            a.clear[i] = z;
    };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int tile, q;
    uint32_t lx, ly;
    if (!pull_tile_of_block(a.sh, (int)blockIdx.x, tile, q, lx, ly)) { clear_slice(); return; }
    const int K = a.sh.K;
    // (both builds read the schedule: the whole-list build sorts a hinted tile's list whole, the heads-only build gives it the
    // longest head one placement holds -- 2047 keys instead of ~940: most quadrants that walk past a normal head end inside it)
    const bool hinted = (PHASE_DBG(a) & 16) || (a.hints != nullptr && a.hints[tile] != 0u);      // uniform over the workgroup
    for (int i = tid; i < GFT_DEPTH_BINS; i += GFT_BLOCK) s_hist[i] = 0;
    if (tid == 0) s_n = 0;
    __syncthreads();

    uint32_t win_base = 0;                               // first depth bin with a cursor (set per placement)
    uint32_t cshift = 0;                                 // a cursor serves 1 << cshift depth bins (a whole list may span more bins than there are cursors)
    uint32_t pbits = 11, pmask = 0x7ffu;                 // bits of a cursor's place
    // pass over one entry list of the supertile.  MODE 0: count the tile's hits and histogram them over the depth bins;
    // MODE 1: keys (gathered depth bits, id) of the hits with depth bin in [blo, bhi) -> LDS in the order they come;
    // MODE 2: the same keys, each to the next free place of its depth bin (s_cur)
    auto scan = [&](const int tid, const uint64_t* __restrict__ list, uint32_t ln, int mode, uint32_t blo, uint32_t bhi) {
        const int lane = tid & 63;
        // (loads unconditionally -- index clamped --, selects afterwards: a load under a lane condition is waited for where
        // its branch joins, which made the four loads two or four round trips in a row; and one trip ahead: a trip is a
        // memory round trip and a little work)
        uint64_t nx[4];
#pragma unroll
        for (int u = 0; u < 4; u++) nx[u] = list[min((uint32_t)(u * GFT_BLOCK + tid), ln - 1u)];
        for (uint32_t i0 = 0; i0 < ln; i0 += 4 * GFT_BLOCK) {
            uint64_t e4[4];
#pragma unroll
            for (int u = 0; u < 4; u++) e4[u] = nx[u];
#pragma unroll
            for (int u = 0; u < 4; u++) nx[u] = list[min(i0 + (uint32_t)((4 + u) * GFT_BLOCK + tid), ln - 1u)];
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (i0 + u * GFT_BLOCK + tid >= ln) e4[u] = 0ull;    // (an all-zero rectangle covers no tile)
            bool hit[4];
            unsigned long long hm[4];
            uint32_t off[4], cnt = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t eb = entry_bin(e4[u]);
                hit[u] = entry_hits(e4[u], lx, ly) && (mode == 0 || (eb >= blo && eb < bhi));
                hm[u] = __builtin_amdgcn_ballot_w64(hit[u]);
                off[u] = cnt;
                cnt += (uint32_t)__popcll(hm[u]);
            }
            if (cnt == 0u) continue;                                 // wave-uniform
            if (mode == 2) {
                // grouped placement: a key goes to the next free place of its depth bin
                uint32_t d4[4];
#pragma unroll
                for (int u = 0; u < 4; u++) d4[u] = (PHASE_DBG(a) & 32) ? (uint32_t)e4[u] : __float_as_uint(a.depth[hit[u] ? (uint32_t)e4[u] : 0u]);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (!hit[u]) continue;
                    const uint32_t b = (entry_bin(e4[u]) - win_base) >> cshift;
                    const uint32_t w = atomicAdd(&s_cur[b >> 1], (b & 1u) ? 0x10000u : 1u);
                    const uint32_t pos = ((b & 1u) ? (w >> 16) : w) & pmask;
                    if (pos < (WHOLE ? TPULL_KEYS_BIG : TPULL_KEYS)) sk[pos] = ((uint64_t)d4[u] << 32) | (uint32_t)e4[u];
                }
                continue;
            }
            // one LDS atomic per wave reserves the slots of all four entries
            uint32_t hb = 0;
            if (lane == 0) hb = atomicAdd(&s_n, cnt);
            if (mode == 0) {
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (hit[u]) atomicAdd(&s_hist[entry_bin(e4[u])], 1u);
                continue;
            }
            hb = (uint32_t)__builtin_amdgcn_readfirstlane((int)hb);
            // the depth gathers of the hits are issued together (a lane without a hit reads Gaussian 0's)
            uint32_t d4[4];
#pragma unroll
            for (int u = 0; u < 4; u++) d4[u] = __float_as_uint(a.depth[hit[u] ? (uint32_t)e4[u] : 0u]);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (!hit[u]) continue;
                const uint32_t id = (uint32_t)e4[u];
                const uint32_t pos = hb + off[u] + (uint32_t)__popcll(hm[u] & ((1ull << lane) - 1ull));
                if (pos < TPULL_KEYS) sk[pos] = ((uint64_t)d4[u] << 32) | id;
                // (the Gaussians are marked as needed behind the scan, from the keys: a store in this loop was
                // waited for by the next entry's LDS write -- one HBM write acknowledgement per entry and trip)
            }
        }
    };

    // pass A: slabs front to back until the head is covered (one slab = the whole list when K == 1); a hinted tile,
    // which may sort its whole list, looks at all of them
    int kstop = K - 1;
    uint32_t n = 0;                                      // hits in the slabs that were scanned
    for (int k = 0; k < K; k++) {
        const uint32_t ln = a.st_cnt[q * K + k];
        if (ln) scan(tid, a.sl_ent + a.st_start[q * K + k], ln, 0, 0u, 0u);
        __syncthreads();
        n = s_n;
        __syncthreads();                                 // (everybody has read the count before the next slab adds to it)
        if (n >= (hinted ? (WHOLE ? 0xffffffffu : HEAD_DIRECT + 1u) : HEAD_TARGET)) { kstop = k; break; }
    }
    if ((PHASE_DBG(a) & 15) == 1) { if (tid < 4) a.unit_flag[4 * tile + tid] = 0; if (tid == 0) { a.ranges[tile] = make_uint2(0u, 0u); a.front_len[tile] = 0; a.tile_cut[tile] = GFT_NO_TAIL; } return; }
    bool more_slabs = false;
    for (int k = kstop + 1; k < K; k++) more_slabs |= a.st_cnt[q * K + k] != 0u;
    // the histogram, 16 bins per thread, and its running count: run0 = hits in the bins in front of this thread's sixteen,
    // xin = hits up to and including them
    uint32_t h[16], sum = 0, run0, xin;
    {
#pragma unroll
        for (int k = 0; k < 16; k++) { h[k] = s_hist[16 * tid + k]; sum += h[k]; }
        uint32_t x = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) s_wt[wave] = x;
        if (tid == 0) { s_gmax = 0; s_cut = GFT_DEPTH_BINS; s_kf = 0; }
        __syncthreads();
        for (int w = 0; w < wave; w++) x += s_wt[w];
        xin = x;
        run0 = x - sum;
    }
    // A hinted tile with a list beyond one placement sorts it whole, in chunks of whole depth bins of fewer than TPULL_KEYS
    // keys each (a chunk is one placement: cursors of 11 bits) -- unless a single bin holds more than 255 keys (the bin
    // counts are then carried as bytes): such a tile takes the lazy route, whose tail builder sorts anything.
    bool whole = WHOLE && hinted && n > HEAD_DIRECT;
    if (WHOLE && whole) {
        uint32_t gm = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) gm = max(gm, h[k]);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) gm = max(gm, (uint32_t)__shfl_xor((int)gm, d, 64));
        if (lane == 0 && gm) atomicMax(&s_gmax, gm);
        __syncthreads();
        whole = s_gmax <= 255u;
        // every bin small enough for a 4-bit count, pairs of bins included: chunks of up to TPULL_KEYS_BIG - 1 keys with 12-bit
        // places -- a 3000-key list in one round instead of two
        if (whole && s_gmax <= 7u) { pbits = 12; pmask = 0xfffu; }
        __syncthreads();                                 // (s_gmax is set anew by every placement)
    }
    // the head (a tile that does not sort its whole list): whole bins up to the one where the running count reaches
    // HEAD_TARGET.  That bin is taken if the head then still sorts as one 1024-key unit; a bin that overshoots is left out
    // unless the head would otherwise be shorter than 512 and the bin fits the 2048-key sorter.  first_tail = first bin
    // outside the head.  A hinted tile: whole bins while they fit one placement (fewer than 2048 keys).
    uint32_t first_tail = (uint32_t)(kstop + 1) << a.sh.kshift, kf = n;
    if (!whole && n > HEAD_DIRECT) {
        const uint32_t tgt = hinted ? HEAD_DIRECT : HEAD_TARGET;
        uint32_t run = run0;
        if (run < tgt && xin >= tgt) {                           // exactly one thread: the crossing lies in its 16 bins
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t before = run;
                run += h[k];
                if (before < tgt && run >= tgt) {
                    const bool take = !hinted && (run <= 1024u || (before < 512u && run <= TPULL_KEYS));
                    s_cut = (uint32_t)(16 * tid + k) + (take ? 1u : 0u);
                    s_kf = take ? run : before;
                }
            }
        }
        __syncthreads();
        first_tail = s_cut;
        kf = s_kf;
        __syncthreads();
    }
    // From here on the sixteen bin counts of a thread travel as bytes (saturated at 255): what follows needs them exactly
    // only where they are small -- cursors of bins with at most PULL_GROUP_MAX keys, chunks of a whole list (all bins
    // <= 255: checked above) -- and a whole list keeps them over every chunk's sort: four registers instead of sixteen.
    uint32_t hp[4];
#pragma unroll
    for (int w = 0; w < 4; w++)
        hp[w] = min(h[4 * w], 255u) | (min(h[4 * w + 1], 255u) << 8) | (min(h[4 * w + 2], 255u) << 16) | (min(h[4 * w + 3], 255u) << 24);
    auto hcnt = [&](int k) -> uint32_t { return (hp[k >> 2] >> (8 * (k & 3))) & 255u; };

    if ((PHASE_DBG(a) & 15) == 2) { if (tid < 4) a.unit_flag[4 * tile + tid] = 0; if (tid == 0) { a.ranges[tile] = make_uint2(0u, 0u); a.front_len[tile] = 0; a.tile_cut[tile] = GFT_NO_TAIL; } return; }
    // the tile's bookkeeping words (stored by its first placement, behind the scans: a store in front of them holds up
    // the wave's loads) and where its ids go: the tile's head slot, or -- a whole list -- n ids of the pool (over all
    // tiles the pool holds R >= every list it can be asked for)
    uint32_t start = (uint32_t)tile * GFT_HEAD_SLOT;
    if (WHOLE && whole) {
        if (tid == 0) {
            // (ONE returning atomic: the frame's tiles arrive here together and queue on the counter's cache line, ~10 ns each --
            // a second counter beside it, kept for statistics, doubled the 10 us the last of 1200 tiles waits)
            if (PHASE_DBG(a) & 256) s_pool = 0;      // (timing experiment, with the sorters cut off: nobody writes or reads the ids)
            else s_pool = atomicAdd(&a.ctrl[GFT_CTRL_POOLCUR], n);
        }
        __syncthreads();
        start = a.pool_base + s_pool;
        kf = n;
    }
    const bool has_tail = !whole && (kf < n || more_slabs);
    const uint2 bk_range = kf ? make_uint2(start, start + kf) : make_uint2(0u, 0u);   // (empty: (0,0) like the reference)
    const uint32_t bk_cut = has_tail ? first_tail : GFT_NO_TAIL;
    const int k_last = whole ? K - 1 : kstop;
    uint32_t* const list = a.heads + start;

    // One placement per round: the tile's keys with depth bin in [lo, hi) -- c of them, `done` keys in the bins in front of
    // lo -- are collected in LDS, sorted, and their ids written to list[done, done + c); their Gaussians are marked for an
    // appearance.  A head is one round; a whole list takes rounds of whole bins with fewer than TPULL_KEYS - 1 keys.
    // The bins' start places (running count of the histogram) become cursors; the largest bin of the range and the span of
    // its bins decide whether the order inside the bins is found by looking through a key's bin (the usual case: ~940
    // keys spread over hundreds of bins) or the keys are sorted as a whole.
    uint32_t done = 0, lo = 0;
    for (;;) {
        uint32_t hi = first_tail, c = kf;
        // (WHOLE: the thread id and the packed bin counts behind an opaque move: everything the loop body derives from them -- bin
        // numbers, unpacked counts, dozens of LDS addresses of the sorters -- would otherwise be computed once in front of the
        // loop and kept in registers across it: 205 of them, two waves per SIMD)
        int tl = tid;
        if (WHOLE) asm volatile("" : "+v"(tl), "+v"(hp[0]), "+v"(hp[1]), "+v"(hp[2]), "+v"(hp[3]));
        const int ll = tl & 63;
        if (WHOLE && whole) {
            // the thread whose sixteen bins hold the crossing finds the bin (every bin alone fits)
            if (tl == 0) { s_cut = GFT_DEPTH_BINS; s_kf = n - done; }
            __syncthreads();
            const uint32_t lim = done + ((pbits == 12u ? TPULL_KEYS_BIG : TPULL_KEYS) - 1u);
            if (run0 < lim && xin >= lim) {                          // exactly one thread, or none (the rest fits)
                uint32_t run = run0;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const uint32_t before = run;
                    run += hcnt(k);
                    if (before < lim && run >= lim) { s_cut = (uint32_t)(16 * tl + k); s_kf = before - done; }
                }
            }
            __syncthreads();
            hi = s_cut;
            c = s_kf;
        }
        if (tl == 0) { s_gmax = 0; s_gmax2 = 0; s_bmin = GFT_DEPTH_BINS; s_bmax = 0; s_n = 0; }
        __syncthreads();
        {
            uint32_t gm = 0, gm2 = 0, first = GFT_DEPTH_BINS, last1 = 0;     // gm2: largest pair of bins; last1 = last occupied bin + 1
#pragma unroll
            for (int k = 15; k >= 0; k--) {
                const uint32_t b = (uint32_t)(16 * tl + k);
                const uint32_t hb = (b >= lo && b < hi) ? hcnt(k) : 0u;
                gm = max(gm, hb);
                if (!(k & 1)) gm2 = max(gm2, hb + ((b + 1u >= lo && b + 1u < hi) ? hcnt(k + 1) : 0u));
                if (hb) { first = b; last1 = max(last1, b + 1u); }
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                gm = max(gm, (uint32_t)__shfl_xor((int)gm, d, 64));
                gm2 = max(gm2, (uint32_t)__shfl_xor((int)gm2, d, 64));
                first = min(first, (uint32_t)__shfl_xor((int)first, d, 64));
                last1 = max(last1, (uint32_t)__shfl_xor((int)last1, d, 64));
            }
            if (ll == 0) {
                if (gm) atomicMax(&s_gmax, gm);
                if (gm2) atomicMax(&s_gmax2, gm2);
                atomicMin(&s_bmin, first);
                atomicMax(&s_bmax, last1);
            }
        }
        __syncthreads();
        win_base = s_bmin & ~15u;                                        // (a thread's sixteen bins lie inside or outside the window together)
        // (the range's occupied bins span at most twice the cursors)
        cshift = s_bmax > win_base + PULL_WINDOW ? 1u : 0u;
        const uint32_t cmax = 0xffffu >> pbits;                                          // largest bin a cursor can count
        const bool grouped = (cshift ? s_gmax2 : s_gmax) <= cmax && c <= pmask;
        if (grouped && (uint32_t)(16 * tl) >= win_base && (uint32_t)(16 * tl) < win_base + (PULL_WINDOW << cshift)) {
            // keys of [lo, hi) in front of this thread's bins (a thread whose bins straddle lo starts at place 0)
            uint32_t r = (uint32_t)(16 * tl) >= lo ? run0 - done : 0u;
            auto hb = [&](int k) -> uint32_t {
                const uint32_t b = (uint32_t)(16 * tl + k);
                return (b >= lo && b < hi) ? hcnt(k) : 0u;
            };
            if (cshift == 0u) {
#pragma unroll
                for (int k = 0; k < 16; k += 2) {
                    const uint32_t b = (uint32_t)(16 * tl + k) - win_base;
                    const uint32_t c0 = (r & pmask) | (min(hb(k), cmax) << pbits);
                    r += hb(k);
                    const uint32_t c1 = (r & pmask) | (min(hb(k + 1), cmax) << pbits);
                    r += hb(k + 1);
                    s_cur[b >> 1] = c0 | (c1 << 16);                     // (places beyond the range are never used)
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; k += 4) {
                    const uint32_t cw = ((uint32_t)(16 * tl + k) - win_base) >> 1;      // cursor of the bins k, k + 1; the next one of k + 2, k + 3
                    const uint32_t n0 = hb(k) + hb(k + 1), n1 = hb(k + 2) + hb(k + 3);
                    const uint32_t c0 = (r & pmask) | (min(n0, cmax) << pbits);
                    r += n0;
                    const uint32_t c1 = (r & pmask) | (min(n1, cmax) << pbits);
                    r += n1;
                    s_cur[cw >> 1] = c0 | (c1 << 16);
                }
            }
        }
        __syncthreads();
        // pass B: the keys of the range
        if (c) {
            for (int k = 0; k <= k_last; k++) {
                if (((uint32_t)k << a.sh.kshift) >= hi || (((uint32_t)k + 1u) << a.sh.kshift) <= lo) continue;   // slab outside [lo, hi)
                const uint32_t ln = a.st_cnt[q * K + k];
                if (ln) scan(tl, a.sl_ent + a.st_start[q * K + k], ln, grouped ? 2 : 1, lo, hi);
            }
        }
        if (done == 0u) {
            if (tl == 0) {
                const bool cut_short = (PHASE_DBG(a) & 15) == 5;        // (timing experiment without the sorters: nobody may read the ids)
                a.ranges[tile] = cut_short ? make_uint2(0u, 0u) : bk_range;
                a.front_len[tile] = cut_short ? 0u : kf;
                a.tile_cnt[tile] = n;
                a.tile_cut[tile] = cut_short ? GFT_NO_TAIL : bk_cut;
            }
            if (tl < 4) a.unit_flag[4 * tile + tl] = 0;
        }
        __syncthreads();
        // these Gaussians get an appearance (k_appearance): marked here, where only LDS work and stores follow
        if (!(PHASE_DBG(a) & 64)) for (uint32_t i = tl; i < c; i += GFT_BLOCK) a.need[(uint32_t)sk[i]] = 1;
        if (done == 0u && !(PHASE_DBG(a) & 128)) clear_slice();
        uint32_t* __restrict__ ids = list + done;
        if (c == 0u || (PHASE_DBG(a) & 15) == 5) {
        } else if (grouped) {
            // the keys stand grouped by depth bin, the bins in ascending order: a key's place = start of its bin's group + the
            // number of smaller keys in the group (keys are distinct: the id is their low half).  The bin's cursor now holds
            // the end of its group and its size.
            for (uint32_t p = (uint32_t)tl; p < c; p += GFT_BLOCK) {
                const uint64_t key = sk[p];
                const uint32_t b = (gft_depth_bin((uint32_t)(key >> 32), a.sh.near_bits, a.sh.bin_shift) - win_base) >> cshift;
                const uint32_t w = s_cur[b >> 1];
                const uint32_t half = (b & 1u) ? (w >> 16) : (w & 0xffffu);
                const uint32_t end = half & pmask, g = half >> pbits, l0 = end - g;
                uint32_t below = 0;
                for (uint32_t qd = 0; qd < g; qd++) below += sk[l0 + qd] < key ? 1u : 0u;
                ids[l0 + below] = (uint32_t)key;
            }
        } else if (c <= 1024u) {
            const uint32_t npad = next_pow2(c < 2u ? 2u : c);
            for (uint32_t i = (uint32_t)tl + c; i < npad; i += GFT_BLOCK) sk[i] = ~0ull;
            __syncthreads();
            if (npad == 1024u) sort1024_by_rank_and_store(sk, c, tl, ids);
            else head_sort_and_store(sk, c, npad, tl, ids);
        } else {
            // 1025 .. 2048 keys: the register-blocked network wants them at their swizzled slots
            static_assert(TPULL_KEYS == 2048u, "k_tile_pull sorts at most 2048 keys at a time");
            uint64_t mine[TPULL_KEYS / GFT_BLOCK];
#pragma unroll
            for (int k = 0; k < (int)(TPULL_KEYS / GFT_BLOCK); k++) {
                const uint32_t i = (uint32_t)tl + k * GFT_BLOCK;
                mine[k] = i < c ? sk[i] : ~0ull;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < (int)(TPULL_KEYS / GFT_BLOCK); k++) sk[sort_slot((uint32_t)tl + k * GFT_BLOCK)] = mine[k];
            __syncthreads();
            bitonic_blocked<3, 8>(sk, tl);
            for (uint32_t i = (uint32_t)tl; i < c; i += GFT_BLOCK) ids[i] = (uint32_t)sk[sort_slot(i)];
        }
        if (!WHOLE || !whole) break;
        done += c;
        lo = hi;
        if (done >= n) break;
        __syncthreads();                                 // (the ids are out of LDS before the next chunk's keys come in)
    }
}

// ---- lists completed on demand --------------------------------------------------------------------------------
struct TailArgs {
    PreFwdArgs pre;                          // inputs of the appearance evaluation
    SuperShape sh;
    const uint32_t* __restrict__ st_cnt;
    const uint32_t* __restrict__ st_start;
    const uint64_t* __restrict__ sl_ent;
    uint2* __restrict__ ranges;
    uint32_t* point_list;
    uint32_t pool_base;                      // first pool slot inside point_list (= T * GFT_HEAD_SLOT)
    const uint32_t* __restrict__ front_len;
    const uint32_t* __restrict__ unit_flag;
    const uint32_t* __restrict__ tile_cut;
    uint32_t* ctrl;
    uint32_t cap;
    const uint32_t* __restrict__ quad_max;
    uint32_t* __restrict__ order;
    RenderFwdArgs render;                    // the resume pass of the forward blend, run by this kernel's workgroups
    int resume_here;
    int dbg;
};

__device__ __forceinline__ void tail_appearance(const TailArgs& a, uint32_t id)
{
    // 1 = it has its appearance from k_appearance.  3 = some tile's tail builder is giving (or has given) it one in THIS launch:
    // that does not help the tile at hand -- its resumed walk reads the records a moment from now and cannot wait for another
    // workgroup's stores -- so it evaluates the Gaussian as well (the same bits are written).
    if (a.pre.g.need[id] == 1) return;
    a.pre.g.need[id] = 3;
    const float px = a.pre.io.means3D[3 * id], py = a.pre.io.means3D[3 * id + 1], pz = a.pre.io.means3D[3 * id + 2];
    const Mat16 V = load_mat(a.pre.io.viewmatrix);
    // the same expressions as in k_preprocess_fwd / k_appearance: the same distance bit for bit
    const float vz = V.m[2] * px + V.m[6] * py + V.m[10] * pz + V.m[14];
    const float vx = V.m[0] * px + V.m[4] * py + V.m[8] * pz + V.m[12];
    const float vy = V.m[1] * px + V.m[5] * py + V.m[9] * pz + V.m[13];
    appearance_fwd(a.pre, (int)id, threadIdx.x & 63, nullptr, nullptr, px, py, pz, vx, vy, vz);
}

__global__ __launch_bounds__(TAIL_THREADS) void k_tail_build(TailArgs a)
{
    extern __shared__ uint64_t sk_dyn[];
    uint64_t* sk = sk_dyn;                                   // TAIL_LDS_KEYS keys at their swizzled slots
    __shared__ uint32_t s_hist[GFT_DEPTH_BINS];
    __shared__ uint32_t s_m, s_pool, s_hi, s_chunk;
    __shared__ unsigned long long s_nearest;
    if (a.ctrl[GFT_CTRL_TOTAL] > a.cap) return;
    const uint32_t nflag = a.ctrl[GFT_CTRL_NFLAG];
    const int T = a.sh.T;
    // The backward's heavy-first tile order rides on this launch (no separate launch in the backward): by the deepest
    // contributors of the first render pass; a tile with a flagged quadrant, whose walk goes on, counts as heaviest.
    // (not where the frame has the forward's order -- by the walk lengths of the camera's previous frame --: the backward
    // deals by that one, and this launch has nothing to do on a frame that flagged nothing)
    if (a.order && blockIdx.x == 0 && a.ctrl[GFT_CTRL_FWDORDER] == 0u) {
        gft_tile_order_block(T, a.quad_max, a.order, nflag ? a.unit_flag : nullptr);
        if (threadIdx.x == 0) a.ctrl[GFT_CTRL_ORDER_OK] = 1u;
    }
    if (nflag == 0u) return;                                 // the common case: every quadrant saturated inside its head
    const int tid = threadIdx.x, lane = tid & 63;
    const int K = a.sh.K;
    // one workgroup per tile (three are resident per CU): the flagged tiles of a frame -- a few dozen silhouette tiles --
    // are completed side by side; with a tile loop over 256 workgroups two or three of them met in one workgroup on most
    // frames and the kernel took that chain (54 -> 2x us)
    {
        const int tile = (int)blockIdx.x;
        const uint4 f = reinterpret_cast<const uint4*>(a.unit_flag)[tile];
        if ((f.x | f.y | f.z | f.w) == 0u) return;           // uniform per workgroup
        const uint32_t first_tail = a.tile_cut[tile];
        if (first_tail == GFT_NO_TAIL) return;               // (its head is its whole list: nothing to complete)
        const uint32_t kf = a.front_len[tile];
        const int tx = tile % a.sh.gx, ty = tile / a.sh.gx;
        const int q = (ty >> a.sh.sshift) * a.sh.sgx + (tx >> a.sh.sshift);
        const uint32_t lx = (uint32_t)(tx & ((1 << a.sh.sshift) - 1)), ly = (uint32_t)(ty & ((1 << a.sh.sshift) - 1));
        const float qx = (float)(tx * GFT_TILE_X), qy = (float)(ty * GFT_TILE_Y);
        for (int i = tid; i < GFT_DEPTH_BINS; i += TAIL_THREADS) s_hist[i] = 0;
        if (tid == 0) { s_m = 0; s_nearest = ~0ull; }
        __syncthreads();
        // An empty head (every key of the tile in one depth bin too large to sort up front): list position 0, whose depth
        // the render kernels take as the reference of their depth sums, must be the tile's nearest Gaussian as in every
        // other flow -- it is kept whether it reaches a flagged quadrant or not.
        uint32_t keep_id = 0xffffffffu;
        if (kf == 0u) {
            for (int k = 0; k < K; k++) {
                const uint32_t ln = a.st_cnt[q * K + k];
                const uint64_t* __restrict__ list = a.sl_ent + a.st_start[q * K + k];
                for (uint32_t i = tid; i < ln; i += TAIL_THREADS) {
                    const uint64_t e = list[i];
                    if (entry_hits(e, lx, ly) && entry_bin(e) >= first_tail)
                        atomicMin(&s_nearest, ((unsigned long long)__float_as_uint(a.pre.g.depth[(uint32_t)e]) << 32) | (uint32_t)e);
                }
            }
            __syncthreads();
            keep_id = (uint32_t)s_nearest;               // (0xffffffff when the tile has no entry at all)
        }
        // Entries of the supertile behind the head that cover this tile AND reach one of its flagged quadrants.
        // MODE 0: count them, histogram them over the bins, keep the keys of the first TAIL_LDS_KEYS in LDS;
        // MODE 1: keys of those with bin in [lo, hi] -> LDS;  MODE 2: their ids -> dst[0, ...) (one bin too large for LDS)
        auto scan = [&](int mode, uint32_t lo, uint32_t hi, uint32_t* dst) {
            for (int k = 0; k < K; k++) {
                if ((((uint32_t)k + 1u) << a.sh.kshift) <= lo || ((uint32_t)k << a.sh.kshift) > hi) continue;   // slab outside [lo, hi]
                const uint32_t ln = a.st_cnt[q * K + k];
                const uint64_t* __restrict__ list = a.sl_ent + a.st_start[q * K + k];
                // TAIL_ITEMS entries per thread and trip: their loads, then the geometry records of the hits, are in flight together
                // (a trip is two dependent memory round trips)
                for (uint32_t i0 = 0; i0 < ln; i0 += TAIL_ITEMS * TAIL_THREADS) {
                    // (every load unconditional, index clamped / Gaussian 0 for a lane without a hit, selects afterwards: a
                    // load under a lane condition is waited for where its branch joins -- the gathers below were twelve
                    // memory round trips in a row per trip)
                    uint64_t e4[TAIL_ITEMS];
#pragma unroll
                    for (int u = 0; u < TAIL_ITEMS; u++) e4[u] = list[min(i0 + u * TAIL_THREADS + tid, ln - 1u)];
#pragma unroll
                    for (int u = 0; u < TAIL_ITEMS; u++)
                        if (i0 + u * TAIL_THREADS + tid >= ln) e4[u] = 0ull;
                    bool s4[TAIL_ITEMS];
                    float4 ra[TAIL_ITEMS], rb[TAIL_ITEMS];
                    uint32_t dz[TAIL_ITEMS];                             // depth bits of the hits: asked for with their geometry records
                                                                 // (inside the loop below each would be a round trip of its own)
#pragma unroll
                    for (int u = 0; u < TAIL_ITEMS; u++) {
                        const uint32_t b = entry_bin(e4[u]);
                        s4[u] = entry_hits(e4[u], lx, ly) && b >= lo && b <= hi;
                        const uint32_t id = s4[u] ? (uint32_t)e4[u] : 0u;
                        ra[u] = a.pre.g.rec_a[2 * id];
                        rb[u] = a.pre.g.rec_a[2 * id + 1];
                        dz[u] = __float_as_uint(a.pre.g.depth[id]);
                    }
#pragma unroll
                    for (int u = 0; u < TAIL_ITEMS; u++) {
                        const uint32_t id = (uint32_t)e4[u], b = entry_bin(e4[u]);
                        bool sv = s4[u];
                        if (sv)
                            sv = gft_splat_reaches_flagged(f.x, ra[u], rb[u], qx, qy) || gft_splat_reaches_flagged(f.y, ra[u], rb[u], qx + 8.f, qy) ||
                                 gft_splat_reaches_flagged(f.z, ra[u], rb[u], qx, qy + 8.f) || gft_splat_reaches_flagged(f.w, ra[u], rb[u], qx + 8.f, qy + 8.f) ||
                                 id == keep_id;
                        const unsigned long long sm = __builtin_amdgcn_ballot_w64(sv);
                        if (sm == 0ull) continue;                // wave-uniform
                        uint32_t hb = 0;
                        if (lane == 0) hb = atomicAdd(&s_m, (uint32_t)__popcll(sm));
                        hb = (uint32_t)__builtin_amdgcn_readfirstlane((int)hb);
                        if (!sv) continue;
                        const uint32_t pos = hb + (uint32_t)__popcll(sm & ((1ull << lane) - 1ull));
                        if (mode == 0) atomicAdd(&s_hist[b], 1u);
                        if (mode == 2) dst[pos] = id;
                        else if (pos < TAIL_LDS_KEYS) sk[sort_slot(pos)] = ((uint64_t)dz[u] << 32) | id;
                    }
                }
            }
        };
        // sorts the n keys in LDS (n <= TAIL_LDS_KEYS), gives their Gaussians an appearance, writes the ids
        auto finish_lds = [&](uint32_t n, uint32_t* dst) {
            __syncthreads();
            if (PHASE_DBG(a) != 3) for (uint32_t i = tid; i < n; i += TAIL_THREADS) tail_appearance(a, (uint32_t)sk[sort_slot(i)]);
            if (PHASE_DBG(a) == 4) return;
            if (n <= 1024u) {
                // a short tail (the usual case: a few hundred survivors): runs of 256 keys are sorted by one wave each in
                // registers (no workgroup barrier), the runs are merged by rank -- a key's place = its place in its own run +
                // the number of smaller keys in the other runs (binary searches; the keys are distinct: the id is their
                // low half).  sort_slot() permutes inside 32-key blocks, so run r occupies slots [256 r, 256 r + 256).
                const uint32_t runs = (n + 255u) >> 8;
                for (uint32_t i = tid + n; i < 256u * runs; i += TAIL_THREADS) sk[sort_slot(i)] = ~0ull;
                __syncthreads();
                const int wave = tid >> 6;
                if ((uint32_t)wave < runs) bitonic_blocked<2, 6, WaveSync>(sk + 256 * wave, lane, WaveSync());
                __syncthreads();
                for (uint32_t i = tid; i < 256u * runs; i += TAIL_THREADS) {
                    const uint32_t run = i >> 8, pos = i & 255u;
                    const uint64_t key = sk[256u * run + sort_slot(pos)];
                    if (key == ~0ull) continue;                  // padding
                    uint32_t rank = pos;
                    for (uint32_t o = 0; o < runs; o++) {
                        if (o == run) continue;
                        const uint64_t* r = sk + 256u * o;
                        uint32_t lo = 0;                         // number of keys of run o below `key`
#pragma unroll
                        for (uint32_t step = 128; step > 0; step >>= 1)
                            if (r[sort_slot(lo + step - 1)] < key) lo += step;
                        if (r[sort_slot(lo)] < key) lo++;        // (lo <= 255 here)
                        rank += lo;
                    }
                    dst[rank] = (uint32_t)key;
                }
                __syncthreads();
                return;
            } else {
                // (larger register-blocked networks -- 16 or 32 keys per thread -- would push this kernel into scratch memory,
                // which costs every launch, also the idle ones, tens of microseconds: longer tails go in runs of bins)
                static_assert(TAIL_LDS_KEYS == 4096u && TAIL_THREADS == 512, "one 4096-key network: 8 keys per thread");
                for (uint32_t i = tid + n; i < TAIL_LDS_KEYS; i += TAIL_THREADS) sk[sort_slot(i)] = ~0ull;
                __syncthreads();
                bitonic_blocked<3, 9>(sk, tid);
            }
            for (uint32_t i = tid; i < n; i += TAIL_THREADS) dst[i] = (uint32_t)sk[sort_slot(i)];
            __syncthreads();
        };
        if (PHASE_DBG(a) == 1) return;
        scan(0, first_tail, GFT_DEPTH_BINS - 1u, nullptr);
        __syncthreads();
        const uint32_t m = s_m;
        if (PHASE_DBG(a) == 2) return;
        // the completed list (head copy + culled tail) takes kf + m pool slots; over all tiles that is at most R <= cap
        if (tid == 0) s_pool = atomicAdd(&a.ctrl[GFT_CTRL_POOLCUR], kf + m);
        __syncthreads();
        uint32_t* list = a.point_list + a.pool_base + s_pool;
        const uint32_t* head = a.point_list + a.ranges[tile].x;        // (where k_tile_pull put the sorted head)
        for (uint32_t i = tid; i < kf; i += TAIL_THREADS) list[i] = head[i];
        if (m <= TAIL_LDS_KEYS) {
            finish_lds(m, list + kf);
        } else {
            // more survivors than LDS holds: bin ranges of at most TAIL_LDS_KEYS survivors, front to back
            uint32_t done = 0, lo = first_tail;
            while (done < m) {
                if (tid == 0) {
                    uint32_t hi = lo, c = s_hist[lo];
                    while (hi + 1u < GFT_DEPTH_BINS && c + s_hist[hi + 1u] <= TAIL_LDS_KEYS) c += s_hist[++hi];
                    s_hi = hi; s_chunk = c; s_m = 0;
                }
                __syncthreads();
                const uint32_t hi = s_hi, c = s_chunk;
                uint32_t* dst = list + kf + done;
                if (c <= TAIL_LDS_KEYS) {
                    if (c) { scan(1, lo, hi, nullptr); finish_lds(c, dst); }
                } else {
                    // one depth bin with more survivors than LDS holds (thousands of Gaussians within 1e-3 of one depth, all
                    // on one tile): ids to the list, sorted in place by (gathered depth bits, id) -- slow, never seen in practice
                    scan(2, lo, hi, dst);
                    __threadfence_block();
                    __syncthreads();
                    for (uint32_t i = tid; i < c; i += TAIL_THREADS) tail_appearance(a, dst[i]);
                    auto ld = [&](uint32_t i) {
                        const uint32_t id = __hip_atomic_load(&dst[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        return ((uint64_t)__float_as_uint(a.pre.g.depth[id]) << 32) | id;
                    };
                    bitonic_ascending<TAIL_THREADS>(c, next_pow2(c), tid, ld,
                        [&](uint32_t i, uint64_t v) { __hip_atomic_store(&dst[i], (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); },
                        [] { __threadfence_block(); __syncthreads(); });
                }
                done += c;
                lo = hi + 1u;
                __syncthreads();
            }
        }
        if (tid == 0) a.ranges[tile] = make_uint2(a.pool_base + s_pool, a.pool_base + s_pool + kf + m);
        // The list is complete: the tile's flagged quadrants go on from where the first pass parked them, here, a wave each
        // -- the resume pass without a launch of its own (3.5 us on every frame, flagged or not).  What this workgroup has
        // written (list, range, the newcomers' appearance) is made visible first; the key buffer is free for the staging.
        __threadfence();
        __syncthreads();
        if (a.resume_here) {
            const int w = tid >> 6;
            const uint32_t fl = w == 0 ? f.x : w == 1 ? f.y : w == 2 ? f.z : f.w;
            if (w < 4 && fl != 0u) {
                float4* stage = reinterpret_cast<float4*>(sk_dyn) + (size_t)w * (RB * 4);
                render_fwd_walk(a.render, 4 * tile + w, lane, stage, stage + RB * 2);
            }
        }
    }
}

}  // namespace

// does k_tail_build run the forward blend's resume pass itself (GFT_TAIL_RESUME=0: a launch of k_render_fwd behind it, as with
// whole-frame binning)?
bool gft_tail_resumes()
{
    static const bool on = [] { const char* e = getenv("GFT_TAIL_RESUME"); return e ? atoi(e) != 0 : true; }();
    return on;
}

// supertile side: the smallest power of two >= 2 tiles that leaves at most GFT_SUPER_MAX supertiles; depth slabs per
// supertile list: 4 on big tile grids (S >= 4: sixteen tiles share a list several times their own length; a tile then
// scans the slabs front to back only until its head is covered), one otherwise.  Measured at 5 M @ 1080p, count +
// scatter + pull: 1 slab 48 + 100 + 476 us, 4 slabs 53 + 157 + 245, 16 slabs 70 + 194 + 197 (the scatter's chunks per
// (workgroup, supertile, slab) shrink to single 8-byte writes)
SuperShape gft_super_shape(const gft_config& c)
{
    static const int env_slabs = [] { const char* e = getenv("GFT_SLABS"); return e ? atoi(e) : 0; }();
    SuperShape sh;
    sh.gx = (c.W + GFT_TILE_X - 1) / GFT_TILE_X;
    sh.gy = (c.H + GFT_TILE_Y - 1) / GFT_TILE_Y;
    sh.T = sh.gx * sh.gy;
    sh.sshift = 1;
    for (;;) {
        const int S = 1 << sh.sshift;
        sh.sgx = (sh.gx + S - 1) / S;
        sh.sgy = (sh.gy + S - 1) / S;
        sh.NS = sh.sgx * sh.sgy;
        if (sh.NS <= GFT_SUPER_MAX) break;
        sh.sshift++;
    }
    int K = sh.sshift >= 2 ? 4 : 1;
    if (env_slabs == 1 || env_slabs == 2 || env_slabs == 4 || env_slabs == 8 || env_slabs == 16) K = env_slabs;
    sh.K = K;
    sh.kshift = 12;
    for (int k = K; k > 1; k >>= 1) sh.kshift--;
    // depth bins: 4096 equal steps of the float bits between the near and the far plane (log-spaced inside an octave's
    // mantissa steps, i.e. monotone in the depth bits: positive floats order like their bits)
    const float nr = c.near_n > 1e-6f ? c.near_n : 1e-6f;
    const float fr = c.far_n > nr ? c.far_n : 2.0f * nr;
    uint32_t nb, fb;
    memcpy(&nb, &nr, 4);
    memcpy(&fb, &fr, 4);
    sh.near_bits = nb;
    sh.bin_shift = 0;
    while (((fb - nb) >> sh.bin_shift) >= (uint32_t)GFT_DEPTH_BINS - 1u) sh.bin_shift++;
    return sh;
}

// the rectangle of an entry relative to its supertile is packed into 4 x 5 bits: supertiles of at most 16 x 16 tiles
bool gft_tile_pull_ok(const gft_config& c) { return gft_super_shape(c).sshift <= 4; }

// dynamic LDS the scatter pass may ask for: the CU's 160 KB less the kernel's few static words
constexpr size_t SUPER_SCATTER_LDS = 160 * 1024 - 256;
static_assert(SUPER_SCATTER_LDS >= (size_t)GFT_SUPER_CELLS * 8, "the direct scatter's two cell tables");

// words of a camera's list schedule (gft_forward_io.cell_sched) at this image size: start | capacity per cell, {cells, valid, -, -}
size_t gft_cell_sched_words_of(const gft_config& c) { const SuperShape sh = gft_super_shape(c); return 2 * (size_t)sh.NS * sh.K + 4; }

// pass 0: count (+ R, mailbox); pass 1: scatter of the entries to their (supertile, slab) lists; pass 2: the scatter by the
// caller's schedule, without a count pass in front (+ R, mailbox)
hipError_t gft_launch_super_bin(hipStream_t s, const gft_config& c, const GeomView& g, const ImgView& im, const BinView& b,
                                uint32_t* mail, uint32_t seq, int pass, uint32_t cap, const uint32_t* hints, uint32_t* sched,
                                int64_t status_cap)
{
    SuperArgs a;
    a.post = (status_cap >= 0 && mail) ? (uint32_t)status_cap + 1u : 0u;
    a.hints = pass != 1 ? hints : nullptr;
    a.sched = pass != 1 ? sched : nullptr;
    if (pass == 2 && !sched) return hipErrorInvalidValue;
    a.P = c.P;
    a.sh = gft_super_shape(c);
    a.rect = g.rect; a.depth = g.depth;
    a.cnt_copy = im.super_tab; a.cur_copy = im.super_tab + GFT_SUPER_CELLS;
    a.st_cnt = im.super_tab + 2 * GFT_SUPER_CELLS; a.st_start = im.super_tab + 3 * GFT_SUPER_CELLS;
    static const int env_copies = [] { const char* e = getenv("GFT_SUPER_COPIES"); return e ? atoi(e) : 0; }();
    a.copies = (env_copies == 1 || env_copies == 2 || env_copies == 4) ? env_copies : 8;
    while (a.copies > 1 && (size_t)a.copies * a.sh.NS * a.sh.K > (size_t)GFT_SUPER_CELLS) a.copies >>= 1;
    // the entry lists live in the key array (`cap` 8-byte slots: there are at most as many (Gaussian, supertile) pairs as
    // (Gaussian, tile) instances)
    a.sl_ent = pass != 0 ? b.keys : nullptr;
    a.ctrl = im.ctrl; a.mail = mail; a.seq = seq; a.cap = cap;
    const int blocks = (c.P + BIN_CHUNK - 1) / BIN_CHUNK;
    // staging (scatter pass, from 1024 cells on): what is left of the CU's LDS behind the three cell tables, 10 bytes per entry
    const int cells = a.sh.NS * a.sh.K;
    a.stage_cap = 0;
    if (pass != 0 && cells >= 1024 && cells <= 4096) {
        const size_t room = SUPER_SCATTER_LDS - ((size_t)cells * 12 + 8);
        a.stage_cap = (uint32_t)((room / 10) & ~(size_t)3);
    }
    const size_t lds = a.stage_cap ? (size_t)cells * 12 + 8 + (size_t)a.stage_cap * 10
                                   : (size_t)cells * 2 * sizeof(uint32_t);
    {
        static std::atomic<uint64_t> done[5];
        hipError_t e = gft_lds_opt_in(reinterpret_cast<const void*>(&k_super_bin<0>), (size_t)GFT_SUPER_CELLS * 8, done[0]);
        if (e == hipSuccess) e = gft_lds_opt_in(reinterpret_cast<const void*>(&k_super_bin<1>), (size_t)GFT_SUPER_CELLS * 8, done[1]);
        if (e == hipSuccess) e = gft_lds_opt_in(reinterpret_cast<const void*>(&k_super_bin<1, true>), SUPER_SCATTER_LDS, done[2]);
        if (e == hipSuccess) e = gft_lds_opt_in(reinterpret_cast<const void*>(&k_super_bin<2>), (size_t)GFT_SUPER_CELLS * 8, done[3]);
        if (e == hipSuccess) e = gft_lds_opt_in(reinterpret_cast<const void*>(&k_super_bin<2, true>), SUPER_SCATTER_LDS, done[4]);
        if (e != hipSuccess) return e;
    }
    if (pass == 0) hipLaunchKernelGGL(k_super_bin<0>, dim3(blocks), dim3(BIN_THREADS), lds, s, a);
    else if (pass == 2 && a.stage_cap) hipLaunchKernelGGL((k_super_bin<2, true>), dim3(blocks), dim3(BIN_THREADS), lds, s, a);
    else if (pass == 2) hipLaunchKernelGGL(k_super_bin<2>, dim3(blocks), dim3(BIN_THREADS), lds, s, a);
    else if (a.stage_cap) hipLaunchKernelGGL((k_super_bin<1, true>), dim3(blocks), dim3(BIN_THREADS), lds, s, a);
    else hipLaunchKernelGGL(k_super_bin<1>, dim3(blocks), dim3(BIN_THREADS), lds, s, a);
    return hipGetLastError();
}

hipError_t gft_launch_tile_pull(hipStream_t s, const gft_config& c, const GeomView& g, const ImgView& im, const BinView& b,
                                uint32_t cap, float* clear, size_t clear_bytes, const uint32_t* hints, bool whole_lists)
{
    PullArgs a;
    a.sh = gft_super_shape(c);
    a.depth = g.depth;
    a.st_cnt = im.super_tab + 2 * GFT_SUPER_CELLS; a.st_start = im.super_tab + 3 * GFT_SUPER_CELLS;
    a.sl_ent = b.keys;
    a.ranges = im.ranges; a.heads = b.point_list; a.front_len = im.front_len; a.unit_flag = im.unit_flag;
    a.tile_cnt = im.tile_cnt; a.tile_cut = im.tile_cut; a.need = g.need;
    a.ctrl = im.ctrl; a.cap = cap;
    a.clear = reinterpret_cast<float4*>(clear); a.clear_vec4 = clear_bytes / 16;
    a.hints = hints; a.pool_base = (uint32_t)a.sh.T * GFT_HEAD_SLOT;
    a.dbg = 0;
#ifdef GFT_PHASE_DBG
    static const int dbg = [] { const char* e = getenv("GFT_PULL_DBG"); return e ? atoi(e) : 0; }();
    a.dbg = dbg;
    if ((dbg & 15) == 4) { a.clear = nullptr; a.dbg = dbg & 16; }     // (+16: every tile counts as hinted)
#endif
    const int Np = a.sh.NS << (2 * a.sh.sshift);
    if (whole_lists || (PHASE_DBG(a) & 16)) hipLaunchKernelGGL(k_tile_pull<true>, dim3(8 * ((Np + 7) / 8)), dim3(GFT_BLOCK), 0, s, a);
    else hipLaunchKernelGGL(k_tile_pull<false>, dim3(8 * ((Np + 7) / 8)), dim3(GFT_BLOCK), 0, s, a);
    return hipGetLastError();
}

hipError_t gft_launch_tail_build(hipStream_t s, const gft_config& c, const gft_forward_io& io, const GeomView& g,
                                 const ImgView& im, const BinView& b, uint32_t cap, bool want_order)
{
    TailArgs a;
    a.pre = gft_pre_fwd_args(c, io, g, im, nullptr, true);
    a.sh = gft_super_shape(c);
    a.st_cnt = im.super_tab + 2 * GFT_SUPER_CELLS; a.st_start = im.super_tab + 3 * GFT_SUPER_CELLS;
    a.sl_ent = b.keys;
    a.ranges = im.ranges; a.point_list = b.point_list; a.pool_base = (uint32_t)a.sh.T * GFT_HEAD_SLOT;
    a.front_len = im.front_len; a.unit_flag = im.unit_flag; a.tile_cut = im.tile_cut;
    a.ctrl = im.ctrl; a.cap = cap;
    a.quad_max = im.tile_max; a.order = want_order ? im.tile_order : nullptr;
    a.resume_here = gft_tail_resumes() ? 1 : 0;
    a.render = gft_render_fwd_args(c, io, g, im, b, true, cap, 2, true);
    a.dbg = 0;
#ifdef GFT_PHASE_DBG
    static const int tdbg = [] { const char* e = getenv("GFT_TAIL_DBG"); return e ? atoi(e) : 0; }();
    a.dbg = tdbg;
#endif
    const size_t lds = (size_t)SORT_SLOTS(TAIL_LDS_KEYS) * 8;
    {
        static std::atomic<uint64_t> done{0};
        const hipError_t e = gft_lds_opt_in(reinterpret_cast<const void*>(&k_tail_build), lds, done);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_tail_build, dim3(a.sh.T), dim3(TAIL_THREADS), lds, s, a);
    return hipGetLastError();
}
