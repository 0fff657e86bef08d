// gft_sort.h -- sorting networks on 64-bit (depth bits << 32 | id) keys in LDS (gfx950), shared by k_binning.hip and
// k_tail.hip: plain ascending bitonic network, register-blocked network (a thread owns 4..16 keys between layout
// changes), wave-local quarter sorts, merge by rank.
#pragma once
#include "gft_internal.h"

namespace {

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

}  // namespace
