// k_render.hip -- per-tile alpha blending, forward and backward (gfx950, wave64).
//
// Work unit = one wavefront = one 8x8 pixel quadrant of a 16x16 tile (64-thread
// workgroups, 4 per tile).  A quadrant walks its tile's depth-sorted splat list on its
// own: no workgroup barriers, it stops as soon as ITS 64 pixels are saturated, and the
// backward starts at ITS deepest contributor.  The four quadrants of a tile get
// consecutive unit ids inside one XCD's run of units, so their (identical) list reads
// hit the same L2.
//
// The list is staged through LDS in batches of 64 packed records (64 B per splat:
// rec_a 32 B + rec_b 32 B, gathered with 16-byte loads, one splat per lane).  While
// staging, the splat's alpha >= 1/255 ellipse is bounded by a box and tested against the
// quadrant; one ballot gives the 64-bit mask of splats that can reach it, walked with
// scalar bit scans, and their records are read back as LDS broadcasts.
//
// forward  (reference K6, RAST/cuda_rasterizer/forward.cu:424-676): front-to-back
//   blend of colour(3, w = a*T), ToF phasor(7, w = a*T^2), distance, accumulation,
//   depth distortion and the first-hit triple.  The per-Gaussian `pixels` counter is a
//   wave popcount -> LDS -> one global atomic per (quadrant, splat).
// backward (reference K7, backward.cu:609-889): back-to-front.  The 18 per-(pixel,
//   splat) float atomics of the reference become 15 sums (the 7 phasor planes are linear in
//   3 per-splat values), a v_permlane32_swap / v_permlane16_swap / DPP reduction tree
//   (15 values -> 4 registers), one 16-byte LDS store per row of lanes (each splat is visited
//   once per batch by its only wave), and one 64-byte-row global atomic burst per
//   (quadrant, splat) that received a contribution.
#include "gft_internal.h"
#include "gft_render_walk.h"

namespace {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// unit v = 4*tile + quadrant; workgroups are dealt round-robin over the 8 XCDs, so give
// every XCD one contiguous run of units
__device__ __forceinline__ int unit_of_block(int b, int V)
{
    const int chunk = (V + 7) >> 3;
    return (b & 7) * chunk + (b >> 3);
}

__global__ __launch_bounds__(64) void k_render_fwd(RenderFwdArgs a)
{
    __shared__ float4 sA[RB * 2];
    __shared__ float4 sB[RB * 2];

    // binning buffer smaller than the instance count: nothing was binned, the host re-runs stage 2
    if (a.ctrl && a.ctrl[GFT_CTRL_TOTAL] > a.cap) return;
    if (a.resume && *a.nflag == 0u) return;               // no quadrant asked for its tail
    const int V = a.T * 4;
    int v;
    if (a.fwd_order && *a.fwd_order_ok) {
        // dealt as the backward's are: block b on XCD b & 7 takes quadrant (slot & 3) of the tile of weight rank
        // 8 (slot >> 2) + xcd -- heaviest tiles first, by the walk lengths of this camera's previous frame (k_appearance)
        const int xcd = (int)blockIdx.x & 7, qslot = (int)blockIdx.x >> 3;
        const int rank = 8 * (qslot >> 2) + xcd;
        if (rank >= a.T) return;
        v = (int)a.fwd_order[rank] * 4 + (qslot & 3);
    } else {
        if (((int)blockIdx.x >> 3) >= ((V + 7) >> 3)) return;      // (the grid is sized for the ordered dealing: a few blocks more)
        v = unit_of_block(blockIdx.x, V);
    }
    if (v >= V) return;
    render_fwd_walk(a, v, (int)threadIdx.x, sA, sB);
}

// ---------------------------------------------------------------------------
// Segment-parallel forward: one workgroup of up to FSEG_WAVES waves per 8x8 quadrant, wave s walks list entries
// [256 s, 256 (s + 1)) (the last one to the end of the head).  For a frame in which nothing saturates early -- the regime
// the reference trains in: 100 k Gaussians, opacities from 0.1, every list walked whole -- one wave per quadrant is one
// serial chain of ~1700 entries on 15 % of the chip's wave slots; the blend, however, is a composition of per-entry maps
// that can be cut anywhere once the transmittance in front of the cut is known:
//   phase 1  every wave multiplies up the transmittance factor of ITS segment from T = 1 (alpha test only: record,
//            power, exp, one multiply per reaching entry -- no sums, no counters);
//   exchange T in front of segment s = product of the factors of segments 0 .. s - 1 (LDS, one barrier); a pixel is
//            done in front of s exactly when that product is below 1e-4 (the serial walk's test_T is the running
//            product, and it is monotone);
//   phase 2  every wave blends its segment with the reference's arithmetic and stop rule from the TRUE transmittance
//            (forward.cu:536-631: power / alpha skips, test_T < 1e-4 ends the pixel without blending the entry, pixel
//            counts, first hit, contributor numbers);
//   combine  sums and deepest contributors add over the segments; T_final is that of the first segment that ended
//            done; the first-hit triple that of the first segment with a hit; the prefix sums in front of every cut ARE
//            the blend-state snapshots the backward's segments start from.
// Differences to the serial walk: the transmittance in front of a cut is (T0 T1 ..) instead of ((((1 a)(1 b)) ..): fp32
// association, ~1e-7 relative; a pixel whose running product stands within that of 1e-4 at a cut may end one entry
// earlier or later -- the entry in question is never blended either way (any alpha >= 1/255 takes it below the
// threshold).  Used for the frames whose previous frame of the same shape walked most of its lists (api.py decides
// from the forward's late report); bit-identical results from call to call, like the serial kernel.
#define FSEG_WAVES_MAX 8
#define FSEG_MIN_LEN 128
#define FSEG_SPEC_MAX 512     // longest segment that is blended speculatively (its pixel counts wait in LDS)

template <int FSEG_WAVES>      // waves per workgroup = segments per quadrant at most
__global__ __launch_bounds__(64 * FSEG_WAVES) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_render_fwd_seg(RenderFwdArgs a)
{
    __shared__ float4 sStage[FSEG_WAVES][RB * 4];        // per wave: rec_a | rec_b of a batch; afterwards the segment's result
    __shared__ float sT[FSEG_WAVES][64];                 // transmittance factor of every segment, per pixel
    __shared__ uint8_t sCnt[FSEG_WAVES][FSEG_SPEC_MAX];  // speculative pass: pixels that blended every entry of the segment

    if (a.ctrl && a.ctrl[GFT_CTRL_TOTAL] > a.cap) return;
    const int V = a.T * 4;
    const int v = unit_of_block(blockIdx.x, V);
    if (v >= V) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float4* sA = &sStage[wave][0];
    float4* sB = &sStage[wave][RB * 2];
    const int tile = v >> 2, quad = v & 3;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int qx0 = tx * GFT_TILE_X + (quad & 1) * 8, qy0 = ty * GFT_TILE_Y + (quad >> 1) * 8;
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = a.ranges[tile];
    const int full = (int)(range.y - range.x);
    const int head = a.front_len ? (int)a.front_len[tile] : full;
    const bool more = a.tile_cut != nullptr && a.tile_cut[tile] != GFT_NO_TAIL;
    const int total = head;
    // Segments of equal length, whole 64-entry batches, at least FSEG_MIN_LEN entries (uniform over the workgroup)
    const int L = max(FSEG_MIN_LEN, RB * ((total + RB * FSEG_WAVES - 1) / (RB * FSEG_WAVES)));
    const int S = max(1, (total + L - 1) / L);
    // (waves without a segment leave at once: a barrier waits for the surviving waves of its workgroup only)
    if (wave >= S) return;
    const bool active = true;
    const int seg_begin = wave * L;
    const int seg_end = min(total, seg_begin + L);
    const float zref = full > 0 ? a.rec_a[2 * a.point_list[range.x] + 1].z : 0.0f;
    const float4 qbox = make_float4((float)qx0, (float)qy0, 7.f, 7.f);
    // (LDS accesses of one wave are served in the order they were issued: inside a wave's own staging buffer a
    // scheduling barrier is all that stands between the lanes' writes and the broadcast reads)
    auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };

    // ---- pass A ----------------------------------------------------------------------------------------------------
    // Segments of up to FSEG_SPEC_MAX entries are blended SPECULATIVELY from T = 1, without the stop rule: colour-like
    // sums come out short by the factor T_in, phasor sums by T_in^2 (w = alpha T, w_p = alpha T^2), the pixel counts wait
    // in LDS.  If, once T_in is known, no pixel of the quadrant stands below 1e-4 (+ a margin for the rounding of the
    // products) at the end of the segment, no pixel has ended anywhere in it -- the running product is monotone -- and
    // the scaled sums ARE the serial walk's: nothing is evaluated twice (a frame in which nothing saturates, the regime
    // the reference trains in, runs this path only).  Otherwise the segment is blended again with the stop rule, from the
    // true transmittance.  Longer segments (lists beyond 8 x 512 entries, whole-frame binning without the lazy sort) take
    // the transmittance factor alone first (alpha test only) and always blend exactly.
    const bool speculate = S > 1 && L <= FSEG_SPEC_MAX;           // uniform over the workgroup
    float P = 1.0f;                          // transmittance in front of this wave's segment
    bool exact = true;                       // the segment (still) has to be blended with the stop rule
    float T = 1.0f;
    uint32_t last_contributor = 0;
    float C0 = 0, C1 = 0, C2 = 0;
    float PR = 0, PI = 0, PA = 0;
    float Dd = 0, A = 0, DD_D = 0, DD_D2 = 0;
    float WD0 = 0, WD1 = 0, WD2 = 0;
    uint32_t inner_cuts = 0;                 // bit k: the snapshot of list position 256 k was parked by this wave (uniform)
    bool cuts_local = false;                 // ... in the units of the speculative pass (scaled when the prefix is added)
    const unsigned long long in_m = wave_ballot(inside);
    bool alive_start = inside;
    unsigned long long done_m = ~in_m;
    // The first segment's transmittance in front IS known (1): its wave blends exactly straight away -- never twice, and it
    // stops early where the quadrant saturates inside the segment (a frame that saturates early has most of its quadrants
    // end there).  For the waves behind it a pixel that ended here counts as transmittance 0.
    const bool lead_exact = S > 1 && wave == 0;
    // ---- phase 2: the exact blend of the segment from the true transmittance ---------------------------------------
    auto run_exact = [&] {
    alive_start = inside && !(P < 0.0001f);
    done_m = ~wave_ballot(exact ? alive_start : inside);
    if (exact) {
        T = P;
        last_contributor = 0;
        C0 = C1 = C2 = PR = PI = PA = Dd = A = DD_D = DD_D2 = 0.f;
        WD0 = WD1 = WD2 = 0.f;
        inner_cuts = 0;
    }
    if (exact) {
        asm volatile("" : : "v"(zref), "v"(T));
        uint32_t id_next = (seg_begin + lane < seg_end) ? a.point_list[range.x + (uint32_t)(seg_begin + lane)] : 0u;
        for (int base = seg_begin; base < seg_end; base += RB) {
            if (done_m == ~0ull) break;
            // a cut of the backward (list position 256 k) INSIDE this segment: the sums blended by this wave so far and the
            // true transmittance are parked in the cut's snapshot slot; the sums of the segments in front are added below
            if (a.snaps && base > seg_begin && (base & (GFT_SEG_LEN - 1)) == 0 && base / GFT_SEG_LEN <= a.nsnap) {
                float4* sp = a.snaps + ((size_t)v * a.nsnap + (base / GFT_SEG_LEN - 1)) * (GFT_SNAP_F4 * 64) + lane;
                sp[0] = make_float4(T, C0, C1, C2);
                sp[64] = make_float4(PR, PI, PA, Dd);
                sp[128] = make_float4(A, DD_D, DD_D2, 0.f);
                inner_cuts |= 1u << (base / GFT_SEG_LEN);
            }
            const int n = min(RB, seg_end - base);
            bool reach = false;
            uint32_t my_id = 0;
            uint32_t cnt = 0;
            const float4 box = __popcll(~done_m) <= 24 ? box_of_mask(~done_m, qx0, qy0) : qbox;
            wave_sync();
            {
                const uint32_t id = id_next;
                if (base + RB + lane < seg_end) id_next = a.point_list[range.x + (uint32_t)(base + RB + lane)];
                if (lane < n) {
                    my_id = id;
                    reach = stage_splat(id, lane, a.rec_a, a.rec_b, sA, sB, box);
                }
            }
            uint64_t m = to_sgpr(wave_ballot(reach));
            wave_sync();
            auto blend = [&](const int j, const float4& a0, const float4& a1, const float4& b0, const float4& b1) {
                const float dx = a0.x - pxf, dy = a0.y - pyf;
                const float power = -0.5f * (a0.z * dx * dx + a1.x * dy * dy) - a0.w * dx * dy;
                const float alpha = fminf(0.99f, a1.y * gft_exp(power));
                const unsigned long long vm = wave_ballot(!(power > 0.0f)) & wave_ballot(!(alpha < 1.0f / 255.0f)) & ~done_m;
                if (vm == 0ull) return;
                const float test_T = T * (1 - alpha);
                const unsigned long long tm = vm & wave_ballot(test_T < 0.0001f);
                const unsigned long long cm = vm & ~tm;
                done_m |= tm;
                if (cm != 0ull) {
                    const float al = sel_mask(cm, alpha, 0.f);
                    const float w = al * T;
                    const float w_p = w * T;
                    C0 += b0.x * w; C1 += b0.y * w; C2 += b0.z * w;
                    PR += b0.w * w_p; PI += b1.x * w_p; PA += b1.y * w_p;
                    const float dist = a1.w;
                    Dd += dist * w;
                    const unsigned long long fm = cm & wave_ballot(last_contributor == 0u);
                    WD0 = sel_mask(fm, alpha, WD0);
                    WD1 = sel_mask(fm, dist, WD1);
                    WD2 = sel_mask(fm, b1.y, WD2);
                    const float z = a1.z - zref;
                    const float wz = w * z;
                    DD_D += wz;
                    DD_D2 = fmaf(wz, z, DD_D2);
                    A += w;
                    T = sel_mask(cm, test_T, T);
                    last_contributor = sel_mask(cm, (uint32_t)(base + j + 1), last_contributor);
                    {
                        const uint32_t pc = (uint32_t)__popcll(cm);
                        uint32_t m0_keep;
                        asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
                                     : "+v"(cnt), "=&s"(m0_keep) : "s"(pc), "s"(j));
                    }
                }
            };
            if (m) {
                int j0 = (int)__builtin_ctzll(m);
                m &= m - 1;
                float4 p0 = sA[2 * j0], p1 = sA[2 * j0 + 1], q0 = sB[2 * j0], q1 = sB[2 * j0 + 1];
                for (;;) {
                    const bool more1 = m != 0;
                    int j1 = j0;
                    if (more1) { j1 = (int)__builtin_ctzll(m); m &= m - 1; }
                    const float4 r0 = sA[2 * j1], r1 = sA[2 * j1 + 1], t0 = sB[2 * j1], t1 = sB[2 * j1 + 1];
                    blend(j0, p0, p1, q0, q1);
                    if (!more1 || done_m == ~0ull) break;
                    const bool more0 = m != 0;
                    j0 = j1;
                    if (more0) { j0 = (int)__builtin_ctzll(m); m &= m - 1; }
                    p0 = sA[2 * j0]; p1 = sA[2 * j0 + 1]; q0 = sB[2 * j0]; q1 = sB[2 * j0 + 1];
                    blend(j1, r0, r1, t0, t1);
                    if (!more0 || done_m == ~0ull) break;
                }
            }
            if (cnt) atomicAdd(&a.pixels[my_id], (float)cnt);
        }
    }
    };
    if (S > 1) {
        float Tl = 1.0f;
        uint32_t id_next = (seg_begin + lane < seg_end && !lead_exact) ? a.point_list[range.x + (uint32_t)(seg_begin + lane)] : 0u;
        if (lead_exact) {
            run_exact();                                         // (P = 1, exact = true)
            Tl = ((done_m >> lane) & 1ull) ? 0.0f : T;           // a pixel that ended here: nothing behind it blends
        } else if (speculate) {
            unsigned long long hit_m = 0ull;         // pixels that have blended something
            asm volatile("" : : "v"(zref));
            for (int base = seg_begin; base < seg_end; base += RB) {
                if (a.snaps && base > seg_begin && (base & (GFT_SEG_LEN - 1)) == 0 && base / GFT_SEG_LEN <= a.nsnap) {
                    float4* sp = a.snaps + ((size_t)v * a.nsnap + (base / GFT_SEG_LEN - 1)) * (GFT_SNAP_F4 * 64) + lane;
                    sp[0] = make_float4(Tl, C0, C1, C2);
                    sp[64] = make_float4(PR, PI, PA, Dd);
                    sp[128] = make_float4(A, DD_D, DD_D2, 0.f);
                    inner_cuts |= 1u << (base / GFT_SEG_LEN);
                }
                const int n = min(RB, seg_end - base);
                bool reach = false;
                uint32_t cnt = 0;
                wave_sync();
                {
                    const uint32_t id = id_next;
                    if (base + RB + lane < seg_end) id_next = a.point_list[range.x + (uint32_t)(base + RB + lane)];
                    if (lane < n) reach = stage_splat(id, lane, a.rec_a, a.rec_b, sA, sB, qbox);
                }
                uint64_t m = to_sgpr(wave_ballot(reach));
                wave_sync();
                auto blend = [&](const int j, const float4& a0, const float4& a1, const float4& b0, const float4& b1) {
                    const float dx = a0.x - pxf, dy = a0.y - pyf;
                    const float power = -0.5f * (a0.z * dx * dx + a1.x * dy * dy) - a0.w * dx * dy;
                    const float alpha = fminf(0.99f, a1.y * gft_exp(power));
                    const unsigned long long vm = wave_ballot(!(power > 0.0f)) & wave_ballot(!(alpha < 1.0f / 255.0f)) & in_m;
                    if (vm == 0ull) return;
                    const float al = sel_mask(vm, alpha, 0.f);
                    const float w = al * Tl;
                    const float w_p = w * Tl;
                    C0 += b0.x * w; C1 += b0.y * w; C2 += b0.z * w;
                    PR += b0.w * w_p; PI += b1.x * w_p; PA += b1.y * w_p;
                    const float dist = a1.w;
                    Dd += dist * w;
                    const unsigned long long fm = vm & ~hit_m;
                    hit_m |= vm;
                    WD0 = sel_mask(fm, alpha, WD0);
                    WD1 = sel_mask(fm, dist, WD1);
                    WD2 = sel_mask(fm, b1.y, WD2);
                    const float z = a1.z - zref;
                    const float wz = w * z;
                    DD_D += wz;
                    DD_D2 = fmaf(wz, z, DD_D2);
                    A += w;
                    Tl = Tl * (1 - al);                  // (al = 0 leaves it exactly)
                    last_contributor = sel_mask(vm, (uint32_t)(base + j + 1), last_contributor);
                    {
                        const uint32_t pc = (uint32_t)__popcll(vm);
                        uint32_t m0_keep;
                        asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
                                     : "+v"(cnt), "=&s"(m0_keep) : "s"(pc), "s"(j));
                    }
                };
                if (m) {
                    int j0 = (int)__builtin_ctzll(m);
                    m &= m - 1;
                    float4 p0 = sA[2 * j0], p1 = sA[2 * j0 + 1], q0 = sB[2 * j0], q1 = sB[2 * j0 + 1];
                    for (;;) {
                        const bool more1 = m != 0;
                        int j1 = j0;
                        if (more1) { j1 = (int)__builtin_ctzll(m); m &= m - 1; }
                        const float4 r0 = sA[2 * j1], r1 = sA[2 * j1 + 1], t0 = sB[2 * j1], t1 = sB[2 * j1 + 1];
                        blend(j0, p0, p1, q0, q1);
                        if (!more1) break;
                        const bool more0 = m != 0;
                        j0 = j1;
                        if (more0) { j0 = (int)__builtin_ctzll(m); m &= m - 1; }
                        p0 = sA[2 * j0]; p1 = sA[2 * j0 + 1]; q0 = sB[2 * j0]; q1 = sB[2 * j0 + 1];
                        blend(j1, r0, r1, t0, t1);
                        if (!more0) break;
                    }
                }
                if (lane < n) sCnt[wave][base - seg_begin + lane] = (uint8_t)cnt;       // (at most 64 pixels per entry)
            }
        } else {
            // the segment's transmittance factor alone
            for (int base = seg_begin; base < seg_end; base += RB) {
                const int n = min(RB, seg_end - base);
                bool reach = false;
                wave_sync();
                {
                    const uint32_t id = id_next;
                    if (base + RB + lane < seg_end) id_next = a.point_list[range.x + (uint32_t)(base + RB + lane)];
                    if (lane < n) {
                        const float4 a0 = a.rec_a[2 * id], a1 = a.rec_a[2 * id + 1];
                        sA[2 * lane] = a0;
                        sA[2 * lane + 1] = a1;
                        reach = gft_splat_reaches_box(a0, a1, qbox.x, qbox.y, qbox.z, qbox.w);
                    }
                }
                uint64_t m = to_sgpr(wave_ballot(reach));
                wave_sync();
                // (the next entry's record is read from LDS while the current one is evaluated, as in the blend loops)
                if (m) {
                    int j = (int)__builtin_ctzll(m);
                    m &= m - 1;
                    float4 a0 = sA[2 * j], a1 = sA[2 * j + 1];
                    for (;;) {
                        const bool more_e = m != 0;
                        float4 n0 = a0, n1 = a1;
                        if (more_e) { j = (int)__builtin_ctzll(m); m &= m - 1; n0 = sA[2 * j]; n1 = sA[2 * j + 1]; }
                        const float dx = a0.x - pxf, dy = a0.y - pyf;
                        const float power = -0.5f * (a0.z * dx * dx + a1.x * dy * dy) - a0.w * dx * dy;
                        const float alpha = fminf(0.99f, a1.y * gft_exp(power));
                        const bool takes = !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
                        Tl = takes ? Tl * (1 - alpha) : Tl;
                        if (!more_e) break;
                        a0 = n0; a1 = n1;
                    }
                }
            }
        }
        sT[wave][lane] = Tl;
        __syncthreads();
        for (int k = 0; k < wave; k++) P *= sT[k][lane];
        if (speculate && !lead_exact) {
            const float P_end = P * Tl;
            // some pixel of the quadrant at (or within rounding of) the stop threshold by the end of this segment?
            exact = wave_ballot(inside && P_end < 1.0001e-4f) != 0ull;
            if (!exact) {
                const float P2 = P * P;
                T = P_end;
                C0 *= P; C1 *= P; C2 *= P; Dd *= P; A *= P; DD_D *= P; DD_D2 *= P;
                PR *= P2; PI *= P2; PA *= P2;
                cuts_local = true;
                // the pixel counts of the segment: one atomic per entry some pixel blended
                for (int i = lane; i < seg_end - seg_begin; i += 64) {
                    const uint32_t c = sCnt[wave][i];
                    if (c) atomicAdd(&a.pixels[a.point_list[range.x + (uint32_t)(seg_begin + i)]], (float)c);
                }
            }
        }
    }

    if (!lead_exact) run_exact();

    // ---- combine ----------------------------------------------------------------------------------------------------
    if (S > 1) {
        wave_sync();
        if (active) {
            const uint32_t fl = (((done_m >> lane) & 1ull) ? 1u : 0u) | ((exact ? alive_start : inside) ? 2u : 0u);
            float4* res = &sStage[wave][0];
            res[lane] = make_float4(T, C0, C1, C2);
            res[64 + lane] = make_float4(PR, PI, PA, Dd);
            res[128 + lane] = make_float4(A, DD_D, DD_D2, WD0);
            res[192 + lane] = make_float4(WD1, WD2, __uint_as_float(last_contributor), __uint_as_float(fl));
        }
        __syncthreads();
        // state in front of segment `upto` (wave 0: behind the last one = the quadrant's result)
        const int upto = wave == 0 ? S : wave;
        T = 1.0f; C0 = C1 = C2 = PR = PI = PA = Dd = A = DD_D = DD_D2 = 0.f; WD0 = WD1 = WD2 = 0.f;
        last_contributor = 0;
        bool ended = !inside, hit = false;
        for (int k = 0; k < upto; k++) {
            const float4 r0 = sStage[k][lane], r1 = sStage[k][64 + lane], r2 = sStage[k][128 + lane], r3 = sStage[k][192 + lane];
            const uint32_t fl = __float_as_uint(r3.w), lc = __float_as_uint(r3.z);
            // (a segment that started with the pixel done has blended nothing: zeros, T untouched)
            C0 += r0.y; C1 += r0.z; C2 += r0.w; PR += r1.x; PI += r1.y; PA += r1.z; Dd += r1.w;
            A += r2.x; DD_D += r2.y; DD_D2 += r2.z;
            if (!ended && (fl & 2u)) T = r0.x;
            if (!hit && lc != 0u) { WD0 = r2.w; WD1 = r3.x; WD2 = r3.y; hit = true; }
            last_contributor = max(last_contributor, lc);
            ended = ended || (fl & 1u);
        }
        if (wave != 0) {
            if (a.snaps) {
                // blend state in front of this wave's segment, if the backward cuts the quadrant's walk there
                if ((seg_begin & (GFT_SEG_LEN - 1)) == 0 && seg_begin / GFT_SEG_LEN <= a.nsnap) {
                    float4* sp = a.snaps + ((size_t)v * a.nsnap + (seg_begin / GFT_SEG_LEN - 1)) * (GFT_SNAP_F4 * 64) + lane;
                    sp[0] = make_float4(T, C0, C1, C2);
                    sp[64] = make_float4(PR, PI, PA, Dd);
                    sp[128] = make_float4(A, DD_D, DD_D2, 0.f);
                }
                // ... and the cuts inside the segment: the parked sums (this lane's own stores, in front of the barrier
                // above) + the sums of the segments in front; the transmittance parked there is already the true one
                while (inner_cuts) {
                    const int k = __builtin_ctz(inner_cuts);
                    inner_cuts &= inner_cuts - 1;
                    float4* sp = a.snaps + ((size_t)v * a.nsnap + (k - 1)) * (GFT_SNAP_F4 * 64) + lane;
                    float4 s0 = sp[0], s1 = sp[64], s2 = sp[128];
                    if (cuts_local) {
                        // parked by the speculative pass: in its units
                        const float P2 = P * P;
                        s0.x *= P; s0.y *= P; s0.z *= P; s0.w *= P;
                        s1.x *= P2; s1.y *= P2; s1.z *= P2; s1.w *= P;
                        s2.x *= P; s2.y *= P; s2.z *= P;
                    }
                    // (a pixel that was done in front of this segment -- or lies outside the image --: its transmittance
                    // stands where it ended, not at the running product this wave started from, which may be 0)
                    if (!alive_start) s0.x = T;
                    s0.y += C0; s0.z += C1; s0.w += C2;
                    s1.x += PR; s1.y += PI; s1.z += PA; s1.w += Dd;
                    s2.x += A; s2.y += DD_D; s2.z += DD_D2;
                    sp[0] = s0; sp[64] = s1; sp[128] = s2;
                }
            }
            return;
        }
        done_m = wave_ballot(ended);
    }

    // wave 0 holds the quadrant's result: the rest is the serial kernel's epilogue
    const size_t pix_i = inside ? (size_t)a.W * py + px : 0;
    if ((head < full || more) && done_m != ~0ull) {
        if (inside) {
            const bool is_done = (done_m >> lane) & 1ull;
            a.resume_state[4 * pix_i] = make_float4(T, C0, C1, C2);
            a.resume_state[4 * pix_i + 1] = make_float4(PR, PI, PA, Dd);
            a.resume_state[4 * pix_i + 2] = make_float4(A, DD_D, DD_D2, WD0);
            a.resume_state[4 * pix_i + 3] = make_float4(WD1, WD2, __uint_as_float(last_contributor), is_done ? 1.f : 0.f);
        }
        if (lane == 0) {
            a.unit_flag[v] = gft_flag_word(~done_m);
            atomicAdd(a.nflag, 1u);
        }
    }
    if (inside) {
        const size_t HW = (size_t)a.H * a.W;
        const size_t pix = pix_i;
        a.pix_state[pix] = make_float4(T, __uint_as_float(last_contributor), DD_D, DD_D2);
        a.pix_sums[2 * pix] = make_float4(C0, C1, C2, PR);
        a.pix_sums[2 * pix + 1] = make_float4(PI, PA, Dd, A);
        const float* bgp = a.bg + (int64_t)py * a.bsy + (int64_t)px * a.bsx;
        const float g0 = bgp[0], g1 = bgp[a.bsc], g2 = bgp[2 * a.bsc], g3 = bgp[3 * a.bsc];
        const float g4 = bgp[4 * a.bsc], g5 = bgp[5 * a.bsc], g6 = bgp[6 * a.bsc];
        a.out_color[pix] = C0 + T * g0;
        a.out_color[HW + pix] = C1 + T * g1;
        a.out_color[2 * HW + pix] = C2 + T * g2;
        const float dcA = a.dc_offset * PA;
        a.out_phasor[pix] = PR + T * g0;
        a.out_phasor[HW + pix] = PI + T * g1;
        a.out_phasor[2 * HW + pix] = PA + T * g2;
        a.out_phasor[3 * HW + pix] = (PR + dcA) + T * g3;
        a.out_phasor[4 * HW + pix] = (dcA - PR) + T * g4;
        a.out_phasor[5 * HW + pix] = (PI + dcA) + T * g5;
        a.out_phasor[6 * HW + pix] = (dcA - PI) + T * g6;
        a.out_depth[pix] = Dd;
        a.out_acc[pix] = A;
        a.out_dd[pix] = fmaf(A, DD_D2, -DD_D * DD_D);
        a.out_distribution[pix] = WD0;
        a.out_distribution[HW + pix] = WD1;
        a.out_distribution[2 * HW + pix] = WD2;
        a.out_normal[pix] = 0.f; a.out_normal[HW + pix] = 0.f; a.out_normal[2 * HW + pix] = 0.f;
        a.out_entropy[pix] = 0.f;
        a.out_ad[pix] = 0.f;
    }
    uint32_t mx = last_contributor;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
    if (lane == 0) a.quad_max[v] = mx;
    if (a.hint_out && lane == 0) {
        const bool flagged = (head < full || more) && done_m != ~0ull;
        a.hint_out[v] = (flagged || (head > 1024 && mx > GFT_HEAD_TARGET)) ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------
struct RenderBwdArgs {
    int W, H, gx, T;
    const uint2* __restrict__ ranges;
    const uint32_t* __restrict__ point_list;
    const float4* __restrict__ rec_a;
    const float4* __restrict__ rec_b;
    const float* __restrict__ bg;
    int64_t bsc, bsy, bsx;
    float dc_offset;
    const float4* __restrict__ pix_state;
    const float4* __restrict__ pix_sums;
    int split;                 // 1: deep quadrants are shared by several waves
    int nseg;                  // segments (waves) per quadrant at most; snapshots per quadrant = nseg - 1
    const float4* __restrict__ snaps;
    const uint32_t* __restrict__ front_len;   // lazy sort / tile-pull binning: entries the forward's first pass could walk (NULL: all)
    const uint32_t* __restrict__ quad_max;
    const uint32_t* __restrict__ order;     // tiles, heaviest first
    const uint32_t* __restrict__ order_ok;  // ctrl word: the forward computed `order` (NULL: it is valid)
    // the order the forward blend dealt its own waves in (by the walk lengths of this camera's previous frame: k_appearance);
    // where the frame has one, the backward deals by it too and k_tail_build derives none
    const uint32_t* __restrict__ order_fwd;
    const uint32_t* __restrict__ order_fwd_ok;
    const float* __restrict__ g_color; const float* __restrict__ g_phasor; const float* __restrict__ g_depth;
    const float* __restrict__ g_acc; const float* __restrict__ g_dd;
    const uint32_t* __restrict__ ctrl;   // the forward's ctrl words
    uint32_t cap;                        // instances the binning buffer holds: a forward that counted more has binned nothing
    float* acc;   // [P][GFT_ACC_STRIDE]
    float* det;   // deterministic mode: [binning instance][quadrant][16] partial rows instead of atomics (NULL: atomics)
};

__device__ __forceinline__ float swap32_add(float x, float y)
{
    // lanes 0-31: x[l] + x[l+32]; lanes 32-63: y[l-32] + y[l]
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}

__device__ __forceinline__ float swap16_add(float x, float y)
{
    // row0: x.r0 + x.r1; row1: y.r0 + y.r1; row2: x.r2 + x.r3; row3: y.r2 + y.r3
    const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}

// Last stage of the wave reduction: four registers, each holding in every 16-lane row the 16
// partials of one value, are folded into one register.  Folding two registers with a lane
// exchange costs two DPP adds (each writes one half of the lanes via bank_mask) and halves
// the partial count of both, so 4 registers x 16 partials need 4 + 2 + 2 instructions instead
// of the 4 x 4 of a per-register scan.  Afterwards quad q of row r holds, in all four lanes,
// the total of the value that register q had in row r.  (VALU write -> DPP read of the same
// VGPR needs two wait states: s_nop where the schedule does not provide them.)
__device__ __forceinline__ float row_fold4(float t0, float t1, float t2, float t3)
{
    asm volatile(
        "s_nop 1\n\t"
        // distance 8: t0 <- {t0 | t2}, t1 <- {t1 | t3}   (lanes 0-7 | lanes 8-15 of every row)
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "s_nop 0\n\t"
        // distance 4: quads {t0, t1, t2, t3}
        "v_add_f32_dpp %0, %0, %0 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
        "s_nop 1\n\t"
        // inside the quads
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(t0), "+v"(t1)
        : "v"(t2), "v"(t3));
    return t0;
}

typedef float v2f __attribute__((ext_vector_type(2)));

// swap-and-add on register pairs: the permlane swaps work in place on both operands, the two
// sums are one v_pk_add_f32
__device__ __forceinline__ v2f swap32_add2(v2f x, v2f y)
{
    const u32x2 a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x.x), __float_as_uint(y.x), false, false);
    const u32x2 b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x.y), __float_as_uint(y.y), false, false);
    const v2f p = {__uint_as_float(a.x), __uint_as_float(b.x)};
    const v2f q = {__uint_as_float(a.y), __uint_as_float(b.y)};
    return p + q;
}

__device__ __forceinline__ v2f swap16_add2(v2f x, v2f y)
{
    const u32x2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x.x), __float_as_uint(y.x), false, false);
    const u32x2 b = __builtin_amdgcn_permlane16_swap(__float_as_uint(x.y), __float_as_uint(y.y), false, false);
    const v2f p = {__uint_as_float(a.x), __uint_as_float(b.x)};
    const v2f q = {__uint_as_float(a.y), __uint_as_float(b.y)};
    return p + q;
}

// Sum 16 per-lane values (8 "low" L[0..7], 8 "high" H[0..7], as register pairs) over the 64 lanes of the
// wave; position k = {L[0..3], L[4..7], H[0..3], H[4..7]}[k] ends up in lane 4 k (other lanes: unspecified).
// Halving tree: 16 -> 8 registers (lane halves: L[i] | H[i]), 8 -> 4 (16-lane rows: L[i], L[i+4],
// H[i], H[i+4]), then row_fold4: quad q of row r ends up with the value of position 4 r + q.
__device__ __forceinline__ float wave_reduce16(v2f L01, v2f L23, v2f L45, v2f L67, v2f H01, v2f H23, v2f H45, v2f H67)
{
    const v2f s01 = swap32_add2(L01, H01), s23 = swap32_add2(L23, H23);
    const v2f s45 = swap32_add2(L45, H45), s67 = swap32_add2(L67, H67);
    const v2f t01 = swap16_add2(s01, s45), t23 = swap16_add2(s23, s67);
    return row_fold4(t01.x, t01.y, t23.x, t23.y);
}

// Heavy-first launch order for the backward (gft_tile_order_block, gft_internal.h): computed by the forward's
// k_tile_tail launch when no quadrant was flagged, else (and without the lazy sort) by this kernel in the backward.
__global__ __launch_bounds__(1024) void k_tile_order(int T, const uint32_t* __restrict__ quad_max,
                                                     uint32_t* __restrict__ order)
{
    gft_tile_order_block(T, quad_max, order);
}

__global__ __launch_bounds__(64) void k_render_bwd(RenderBwdArgs a)
{
    __shared__ float4 sA[RB * 2];
    __shared__ float4 sB[RB * 2];
    __shared__ uint32_t sId[RB];

    // block b of a segment runs on XCD b & 7; slot b >> 3 of that XCD takes quadrant (slot & 3) of the tile of
    // weight rank 8 * (slot >> 2) + xcd: heavy tiles first, a tile's quadrants on one XCD.
    // Every quadrant wave is one serial chain over its list and all of them are resident from the start, so the deepest
    // lists would set the kernel's duration while the SIMDs run empty.  A deep list [0, tmax) is therefore cut every
    // GFT_SEG_LEN entries: the wave of segment s walks [256 s, 256 (s + 1)) back to front, starting from the blend
    // state the forward saved in front of entry 256 (s + 1) -- what lies behind it is (final sums - sums up to there) --;
    // the last segment walks from the final state as a single wave would.  The same arithmetic per (pixel, splat) as
    // one wave, other summation order of the atomics only.  Workgroups of the last segment first, then segment 0, 1, ...
    // (a forward that was queued without a host read and did not fit its buffer has left no lists: nothing to walk -- its
    // caller learns of it from the status block, gft_forward_enqueue)
    if (a.ctrl[GFT_CTRL_TOTAL] > a.cap) return;
    const int per_seg = (int)gridDim.x / a.nseg;
    const int sgroup = (int)blockIdx.x / per_seg;          // launch group: the workgroups of group 0 start first
    const int bid = (int)blockIdx.x - sgroup * per_seg;
    const int xcd = bid & 7, qslot = bid >> 3;
    const int rank = 8 * (qslot >> 2) + xcd;
    if (rank >= a.T) return;
    const uint32_t* order = a.order;
    bool ordered = order && (a.order_ok == nullptr || *a.order_ok != 0u);
    if (order && a.order_fwd && *a.order_fwd_ok != 0u) { order = a.order_fwd; ordered = true; }
    const int v_unit = (int)(ordered ? order[rank] : (uint32_t)rank) * 4 + (qslot & 3);
    const int tmax = (int)a.quad_max[v_unit];
    if (tmax == 0) return;
    // cuts: at multiples of GFT_SEG_LEN in front of tmax that the forward's first pass walked over (it saved no state
    // past the sorted head of a lazily sorted list)
    int ncut = 0;
    if (a.split) {
        const int head = a.front_len ? (int)a.front_len[v_unit >> 2] : tmax;
        const int lim = min(tmax, head);
        ncut = lim > 0 ? (lim - 1) / GFT_SEG_LEN : 0;                       // cuts at 256, ..., 256 ncut < lim
        if (ncut > a.nseg - 1) ncut = a.nseg - 1;
    }
    // Group 0 walks the LAST segment of its quadrant, group g > 0 segment g - 1: the last segment is the one that is not cut
    // to GFT_SEG_LEN entries -- behind the last snapshot of a list that was sorted whole it is 1200 entries of a
    // 3000-entry walk, and a chain of that length must not be the one that starts last.
    // ... unless it is a SHORT one on a frame whose waves make many rounds over the chip (more than 4096 tiles): there the
    // front segment, with every pixel still open, is the heavier of the two and goes first, group 1 takes the last one
    // (5 M @ 1080p: 538 -> 505 us).  Where nearly all waves are resident at once the same swap costs (metric frame: 153.6 ->
    // 164 us): the light segments between the heavy ones are what keeps the SIMDs mixed.
    const bool front_first = a.T > GFT_FWD_ORDER_MAX_TILES && ncut >= 1 && tmax - ncut * GFT_SEG_LEN <= GFT_SEG_LEN;
    int seg;
    if (sgroup == 0) seg = front_first ? 0 : ncut;
    else if (sgroup == 1) { if (ncut < 1) return; seg = front_first ? ncut : 0; }
    else { seg = sgroup - 1; if (seg >= ncut) return; }
    const int lo_last = seg * GFT_SEG_LEN;                                  // list range [lo_last, hi_first) of this segment
    const int hi_first = seg == ncut ? tmax : lo_last + GFT_SEG_LEN;
    const int tile = v_unit >> 2, quad = v_unit & 3;
    const int lane = threadIdx.x;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int qx0 = tx * GFT_TILE_X + (quad & 1) * 8, qy0 = ty * GFT_TILE_Y + (quad >> 1) * 8;
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const float pxf = (float)px, pyf = (float)py;
    // list position c -> entry of the id list
    const uint32_t r0 = a.ranges[tile].x;
    auto phys = [&](uint32_t c) -> uint32_t { return r0 + c; };
    const float zref = a.rec_a[2 * a.point_list[phys(0u)] + 1].z;                 // the forward's shift of the depth sums (tmax > 0)
    const size_t HW = (size_t)a.H * a.W;
    const size_t pix = inside ? (size_t)a.W * py + px : 0;

    float T_final = 0.f, wz_tot = 0.f, wz2_tot = 0.f;
    int n_contrib = 0;
    float gc0 = 0, gc1 = 0, gc2 = 0, gp0 = 0, gp1 = 0, gp2 = 0, gp3 = 0, gp4 = 0, gp5 = 0, gp6 = 0;
    float gd = 0, ga = 0, gdd = 0;
    float bg_dot = 0.f, bg_dot_p = 0.f;
    if (inside) {
        const float4 st = a.pix_state[pix];
        T_final = st.x; n_contrib = (int)__float_as_uint(st.y); wz_tot = st.z; wz2_tot = st.w;
        if (a.g_color) { gc0 = a.g_color[pix]; gc1 = a.g_color[HW + pix]; gc2 = a.g_color[2 * HW + pix]; }
        if (a.g_phasor) {
            gp0 = a.g_phasor[pix]; gp1 = a.g_phasor[HW + pix]; gp2 = a.g_phasor[2 * HW + pix];
            gp3 = a.g_phasor[3 * HW + pix]; gp4 = a.g_phasor[4 * HW + pix]; gp5 = a.g_phasor[5 * HW + pix];
            gp6 = a.g_phasor[6 * HW + pix];
        }
        if (a.g_depth) gd = a.g_depth[pix];
        if (a.g_acc) ga = a.g_acc[pix];
        if (a.g_dd) gdd = a.g_dd[pix];
        const float* bgp = a.bg + (int64_t)py * a.bsy + (int64_t)px * a.bsx;
        const float b0 = bgp[0], b1 = bgp[a.bsc], b2 = bgp[2 * a.bsc], b3 = bgp[3 * a.bsc];
        const float b4 = bgp[4 * a.bsc], b5 = bgp[5 * a.bsc], b6 = bgp[6 * a.bsc];
        // same summation order as the reference loops (backward.cu:850-857)
        bg_dot = 0.f + b0 * gc0; bg_dot += b1 * gc1; bg_dot += b2 * gc2;
        bg_dot_p = 0.f + b0 * gp0; bg_dot_p += b1 * gp1; bg_dot_p += b2 * gp2; bg_dot_p += b3 * gp3;
        bg_dot_p += b4 * gp4; bg_dot_p += b5 * gp5; bg_dot_p += b6 * gp6;
    }
    // depth-distortion weight gradient dL_dw(z) = gdd*(z^2 (1-Tf) - 2 z wz + wz2) = (A2 z + B2) z + C2
    // (the constant g_acc rides on C2: dL_dw only ever appears as g_acc + dL_dw)
    const float A2 = gdd * (1 - T_final), B2 = -2.0f * gdd * wz_tot, C2 = gdd * wz2_tot + ga;
    const float bg_k = -T_final * (bg_dot + bg_dot_p);       // background term of dL_dalpha, times 1/(1-alpha)
    // upstream phasor gradients folded onto the per-splat basis (R, I, Am):
    // sum_k p_k g_k = R*GR + I*GI + Am*GA; K9 needs sum w_p*{GR, GI, g2, GQ}
    const float GR = gp0 + gp3 - gp4, GI = gp1 + gp5 - gp6, GQ = (gp3 + gp4) + (gp5 + gp6);
    const float GA = gp2 + a.dc_offset * GQ;
    // per-pixel factors of the eight sums that are (pixel constant) x (w_c or w_p), as register pairs
    const v2f gA01 = {gc0, gc1}, gA23 = {gc2, gd}, gB01 = {GR, GI}, gB23 = {gp2, GQ};

    // Back-to-front recurrences.  The reference keeps one "accumulated behind" value per
    // channel (accum_rec[3], accum_rec_p[7], _d, _a, _dd; backward.cu:776-833); only their
    // gradient-weighted sums enter dL_dalpha and all channels of a group share one linear
    // recurrence, so two scalars per group are enough:
    //   S1 <- a*D1 + (1-a)  *S1,  D1 = sum_k c_k g_k + dist*g_d + g_a + dL_dw   (colour, dist, acc, dd)
    //   Sp <- a*Dp + (1-a)^2*Sp,  Dp = sum_k p_k gp_k                           (ToF phasor)
    // updated right after a splat has used them (the reference does the same update lazily at the
    // next splat with last_alpha / last_color: same operands, same value).  A lane that does not
    // blend the splat has a = 0 and keeps S exactly, so no select is needed.
    float T = T_final;
    float S1 = 0.f, Sp = 0.f;

    if (seg < ncut) {
        // blend state in front of entry `cut`, as the forward left it
        const float4* sp = a.snaps + ((size_t)v_unit * (a.nseg - 1) + seg) * (GFT_SNAP_F4 * 64) + lane;
        const float4 s0 = sp[0], s1 = sp[64], s2 = sp[128];
        const float fT = s0.x, qC0 = s0.y, qC1 = s0.z, qC2 = s0.w, qPR = s1.x, qPI = s1.y, qPA = s1.z, qDd = s1.w;
        const float qA = s2.x, qDD_D = s2.y, qDD_D2 = s2.z;
        // what the serial walk holds when it arrives at `cut`:
        //   S1 = sum_{k >= cut} w_k D1_k / T_cut,  Sp = sum_{k >= cut} w_k T_k Dp_k / T_cut^2,
        // the sums taken as (whole list) - (entries before cut); a pixel whose list ended before `cut` has equal
        // sums on both sides and gets exact zeros
        float4 f0 = make_float4(0.f, 0.f, 0.f, 0.f), f1 = f0;
        if (inside) { f0 = a.pix_sums[2 * pix]; f1 = a.pix_sums[2 * pix + 1]; }
        float R1 = (f0.x - qC0) * gc0;
        R1 = fmaf(f0.y - qC1, gc1, R1); R1 = fmaf(f0.z - qC2, gc2, R1); R1 = fmaf(f1.z - qDd, gd, R1);
        R1 = fmaf(A2, wz2_tot - qDD_D2, R1); R1 = fmaf(B2, wz_tot - qDD_D, R1); R1 = fmaf(C2, f1.w - qA, R1);
        float Rp = (f0.w - qPR) * GR;
        Rp = fmaf(f1.x - qPI, GI, Rp); Rp = fmaf(f1.y - qPA, GA, Rp);
        const float rT = __builtin_amdgcn_rcpf(fT);
        T = fT;
        S1 = R1 * rT;
        Sp = Rp * rT * rT;
    }

    for (int hi = hi_first; hi > lo_last; hi -= RB) {        // list indices [hi-n, hi), descending
        const int n = min(RB, hi - lo_last);
        bool reach = false;
        const float4 box = make_float4((float)qx0, (float)qy0, 7.f, 7.f);
        __syncthreads();                           // previous batch's flush has read LDS
        if (lane < n) {
            const uint32_t id = a.point_list[phys((uint32_t)(hi - 1 - lane))];
            sId[lane] = id;
            reach = stage_splat(id, lane, a.rec_a, a.rec_b, sA, sB, box);
        }
        uint64_t m = to_sgpr(wave_ballot(reach));
        __syncthreads();

        // alpha of splat j for this pixel and whether the pixel blended it in the forward
        auto eval = [&](int j, const float4& a0, const float4& a1, float& dx, float& dy, float& G, float& alpha) -> bool {
            const int c = hi - 1 - j;               // list position of this splat
            dx = a0.x - pxf; dy = a0.y - pyf;
            const float power = -0.5f * (a0.z * dx * dx + a1.x * dy * dy) - a0.w * dx * dy;
            G = gft_exp(power);
            alpha = fminf(0.99f, a1.y * G);
            return (c < n_contrib) && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
        };
        // Gradient contributions of splat j, reduced over the 64 pixels into `row`.  Every lane runs
        // the same arithmetic; lanes that do not blend this splat use alpha = G = 0, which leaves T and
        // the two recurrences unchanged (rcp(1) == 1) and makes all 15 partials exactly zero.
        // accumulator row = {dcolor[3], ddist | sum E dx, sum E dy, dconic.xy | XR, XI, X2, XQ | dconic.w, dopacity, dndc, -}
        auto blend = [&](int j, uint32_t gid, const float4& a0, const float4& a1, const float4& b0, const float4& b1, float dx,
                         float dy, float G, float alpha, bool contrib) {
            v2f L01, L23, L45, L67, H01, H23, H45, H67;
            const float al = contrib ? alpha : 0.f;
            const float Gm = contrib ? G : 0.f;
            const float one_m_a = 1.f - al;
            const float rcp_1ma = __builtin_amdgcn_rcpf(one_m_a);
            T = T * rcp_1ma;
            const float wc = al * T;             // dchannel_dcolor == dchannel_ddepth
            const float wp = wc * T;             // dchannel_dphasor = alpha*T*T
            const float dist = a1.w, z = a1.z - zref;
            const float t2 = fmaf(A2, z, B2);    // A2 z + B2
            const float dL_dw_ga = fmaf(t2, z, C2);   // g_acc + dL_dw

            float D1 = b0.x * gc0;
            D1 = fmaf(b0.y, gc1, D1); D1 = fmaf(b0.z, gc2, D1); D1 = fmaf(dist, gd, D1);
            D1 += dL_dw_ga;

            float Dp = b0.w * GR;
            Dp = fmaf(b1.x, GI, Dp); Dp = fmaf(b1.y, GA, Dp);

            // alpha also scales what is left for the background (reference :850-858): -T_final/(1-alpha) * bg . g
            const float dL_dalpha = fmaf(D1 - S1, T, fmaf(fmaf(-2.f * one_m_a, Sp, Dp), T * T, bg_k * rcp_1ma));

            S1 = fmaf(al, D1, one_m_a * S1);
            Sp = fmaf(al, Dp, one_m_a * one_m_a * Sp);

            L01 = gA01 * wc;                      // w_c * (g_c0, g_c1)
            L23 = gA23 * wc;                      // w_c * (g_c2, g_dist)
            H01 = gB01 * wp;                      // w_p * (GR, GI)
            H23 = gB23 * wp;                      // w_p * (g_p2, GQ)

            // E = G dL/dalpha; the per-splat factors of the five geometric sums (opacity, -1/2,
            // 0.5 W, 0.5 H) are applied once per Gaussian in k_preprocess_bwd:
            //   dL/dmean2D.x = -o 0.5W (a sum E dx + b sum E dy),  dL/dconic.x = -o/2 sum E dx^2, ...
            const float E = Gm * dL_dalpha;
            const float Edx = E * dx, Edy = E * dy;
            // dL/dmean2D = -o (0.5 W, 0.5 H) * conic * (sum E dx, sum E dy): the conic is the splat's, so the two plain
            // sums are reduced and k_preprocess_bwd multiplies (6 VALU less per hit than sum E (a dx + b dy), sum E (b dx + c dy))
            L45.x = Edx;
            L45.y = Edy;
            L67.x = Edx * dx;                         // dconic.x / (-o/2)
            L67.y = Edx * dy;                         // dconic.y / (-o/2)
            H45.x = Edy * dy;                         // dconic.w / (-o/2)
            H45.y = E;                                // dopacity
            H67.x = wc * (t2 + A2 * z);           // gdd*2*alpha*T*(z(1-Tf) - wz)
            H67.y = 0.f;
            // 64 pixels -> one partial per value in every fourth lane: 15 of them go straight to the Gaussian's
            // accumulator row (one 64-byte line of float atomics, fire and forget)
            const float tot = wave_reduce16(L01, L23, L45, L67, H01, H23, H45, H67);
            if (a.det) {
                // deterministic mode: the row of this (list entry, quadrant) is stored; k_acc_reduce_det adds the rows
                // in a fixed order
                if ((lane & 3) == 0 && lane < 4 * GFT_NUM_ACC)
                    a.det[((size_t)phys((uint32_t)(hi - 1 - j)) * 4 + quad) * GFT_ACC_STRIDE + (lane >> 2)] = tot;
            } else if ((lane & 3) == 0 && lane < 4 * GFT_NUM_ACC && tot != 0.f)
                atomicAdd(&a.acc[(size_t)gid * GFT_ACC_STRIDE + (lane >> 2)], tot);
        };

        // (as in k_render_fwd: the next entry's LDS records are read while the current one is worked on)
        // (the Gaussian's id for the accumulator row's address comes along: read where it is needed, every entry waits
        // for that LDS read in front of its atomics)
        auto entry = [&](int j, uint32_t gid, const float4& a0, const float4& a1, const float4& b0, const float4& b1) {
            float dx, dy, G, alpha;
            const bool contrib = eval(j, a0, a1, dx, dy, G, alpha);
            if (wave_ballot(contrib) == 0ull) return;     // wave-uniform skip
            blend(j, gid, a0, a1, b0, b1, dx, dy, G, alpha, contrib);
        };
        if (m) {
            int j0 = (int)__builtin_ctzll(m);
            m &= m - 1;
            float4 p0 = sA[2 * j0], p1 = sA[2 * j0 + 1], q0 = sB[2 * j0], q1 = sB[2 * j0 + 1];
            uint32_t g0 = sId[j0];
            for (;;) {
                const bool more1 = m != 0;
                int j1 = j0;
                if (more1) { j1 = (int)__builtin_ctzll(m); m &= m - 1; }
                const float4 r0 = sA[2 * j1], r1 = sA[2 * j1 + 1], t0 = sB[2 * j1], t1 = sB[2 * j1 + 1];
                const uint32_t g1 = sId[j1];
                entry(j0, g0, p0, p1, q0, q1);
                if (!more1) break;
                const bool more0 = m != 0;
                j0 = j1;
                if (more0) { j0 = (int)__builtin_ctzll(m); m &= m - 1; }
                p0 = sA[2 * j0]; p1 = sA[2 * j0 + 1]; q0 = sB[2 * j0]; q1 = sB[2 * j0 + 1];
                g0 = sId[j0];
                entry(j1, g1, r0, r1, t0, t1);
                if (!more0) break;
            }
        }
        // (issuing two splats per iteration in one basic block so that the scheduler can overlap their
        // exp / rcp / DPP latencies was measured: 214 vs 198 us, dropped)
    }
}

// Deterministic mode: adds the partial rows k_render_bwd stored, tile by tile; inside a tile every list entry is another
// Gaussian (no two threads touch one accumulator row), its four quadrant rows are added in the order 0..3, and a
// barrier separates the tiles: every accumulator value is a sum in one fixed order.  One workgroup (a test mode).
__global__ __launch_bounds__(1024) void k_acc_reduce_det(int T, const uint2* __restrict__ ranges,
                                                         const uint32_t* __restrict__ quad_max,
                                                         const uint32_t* __restrict__ point_list,
                                                         const float* __restrict__ det, float* acc,
                                                         const uint32_t* __restrict__ ctrl, uint32_t cap)
{
    if (ctrl[GFT_CTRL_TOTAL] > cap) return;
    const int tid = threadIdx.x;
    for (int tile = 0; tile < T; tile++) {
        const uint4 qm = reinterpret_cast<const uint4*>(quad_max)[tile];
        const uint32_t walked = max(max(qm.x, qm.y), max(qm.z, qm.w));      // entries some quadrant of the tile walked
        const uint2 rg = ranges[tile];
        for (uint32_t i = (uint32_t)tid; i < walked * GFT_ACC_STRIDE; i += 1024u) {
            const uint32_t c = i / GFT_ACC_STRIDE, k = i % GFT_ACC_STRIDE;
            if (k >= GFT_NUM_ACC) continue;
            const uint32_t p = rg.x + c;
            const float* row = det + (size_t)p * 4 * GFT_ACC_STRIDE + k;
            float s = row[0];
            s += row[GFT_ACC_STRIDE];
            s += row[2 * GFT_ACC_STRIDE];
            s += row[3 * GFT_ACC_STRIDE];
            if (s != 0.f) {
                float* dst = acc + (size_t)point_list[p] * GFT_ACC_STRIDE + k;
                __hip_atomic_store(dst, __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + s,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __threadfence();
        __syncthreads();
    }
}

}  // namespace

RenderFwdArgs gft_render_fwd_args(const gft_config& c, const gft_forward_io& io, const GeomView& g, const ImgView& im,
                                  const BinView& b, bool check_cap, uint32_t cap, int lazy, bool pull)
{
    RenderFwdArgs a;
    a.gx = (c.W + GFT_TILE_X - 1) / GFT_TILE_X;
    a.T = a.gx * ((c.H + GFT_TILE_Y - 1) / GFT_TILE_Y);            // (first: the choices below depend on it)
    a.nsnap = gft_bwd_segments((size_t)((c.W + GFT_TILE_X - 1) / GFT_TILE_X) * (size_t)((c.H + GFT_TILE_Y - 1) / GFT_TILE_Y)) - 1;
    a.snaps = (c.want_backward && a.nsnap > 0) ? im.snaps : nullptr;
    a.ctrl = check_cap ? im.ctrl : nullptr;
    a.cap = cap;
    a.tile_cut = pull ? im.tile_cut : nullptr;
    // (the segment-parallel kernel cuts a list by the length of its sorted part -- what the schedule changes: its sums would
    // depend on it in the last bit.  Frames that run it keep no schedule; their lists rarely exceed one placement anyway)
    a.hint_out = (pull && lazy == 1 && !gft_fwd_segmented(a.T)) ? reinterpret_cast<uint8_t*>(io.tile_hints) : nullptr;
    a.W = c.W; a.H = c.H;
    a.gx = (c.W + GFT_TILE_X - 1) / GFT_TILE_X;
    const int gy = (c.H + GFT_TILE_Y - 1) / GFT_TILE_Y;
    a.T = a.gx * gy;
    a.ranges = im.ranges; a.point_list = b.point_list; a.rec_a = g.rec_a; a.rec_b = g.rec_b;
    a.bg = io.bg; a.bsc = c.bg_stride_c; a.bsy = c.bg_stride_y; a.bsx = c.bg_stride_x;
    a.dc_offset = c.dc_offset;
    a.pix_state = im.pix_state; a.quad_max = im.tile_max;
    a.pix_sums = im.pix_sums;
    a.out_color = io.out_color; a.out_phasor = io.out_phasor; a.out_depth = io.out_depth;
    a.out_normal = io.out_normal; a.out_acc = io.out_acc; a.out_entropy = io.out_entropy;
    a.out_dd = io.out_depth_distortion; a.out_ad = io.out_amp_distortion;
    a.out_distribution = io.out_distribution; a.pixels = io.pixels;
    // lazy: 0 = lists sorted whole, 1 = first pass over the sorted heads, 2 = resume pass
    a.front_len = lazy ? im.front_len : nullptr;
    a.unit_flag = im.unit_flag;
    a.nflag = im.ctrl + GFT_CTRL_NFLAG;
    a.resume_state = im.resume_state;
    a.resume = lazy == 2;
    const bool ordered = pull && lazy == 1 && io.tile_weights != nullptr && gft_fwd_ordered(a.T);
    a.fwd_order = ordered ? im.tile_cursor : nullptr;
    a.fwd_order_ok = im.ctrl + GFT_CTRL_FWDORDER;
    a.weights_out = (pull && lazy != 0 && gft_fwd_ordered(a.T)) ? io.tile_weights : nullptr;
    return a;
}

hipError_t gft_launch_render_fwd(hipStream_t s, const gft_config& c, const gft_forward_io& io, const GeomView& g,
                                 const ImgView& im, const BinView& b, bool check_cap, uint32_t cap, int lazy, bool pull,
                                 bool segmented)
{
    const RenderFwdArgs a = gft_render_fwd_args(c, io, g, im, b, check_cap, cap, lazy, pull);
    const bool ordered = a.fwd_order != nullptr;
    const int blocks = ordered ? 32 * ((a.T + 7) / 8) : 8 * ((a.T * 4 + 7) / 8);
    // segmented: up to FSEG_WAVES waves per quadrant (first pass only; the resume pass of flagged quadrants stays one
    // wave per quadrant).  GFT_FWD_SEG=0 / 1 in the environment forces one of the two kernels for every frame.
    const bool seg = segmented && !a.resume && gft_fwd_segmented(a.T);
    if (seg) {
        switch (gft_fwd_seg_waves(a.T)) {
        case 2: hipLaunchKernelGGL(k_render_fwd_seg<2>, dim3(blocks), dim3(64 * 2), 0, s, a); break;
        case 3: hipLaunchKernelGGL(k_render_fwd_seg<3>, dim3(blocks), dim3(64 * 3), 0, s, a); break;
        case 4: hipLaunchKernelGGL(k_render_fwd_seg<4>, dim3(blocks), dim3(64 * 4), 0, s, a); break;
        case 5: hipLaunchKernelGGL(k_render_fwd_seg<5>, dim3(blocks), dim3(64 * 5), 0, s, a); break;
        case 6: hipLaunchKernelGGL(k_render_fwd_seg<6>, dim3(blocks), dim3(64 * 6), 0, s, a); break;
        case 7: hipLaunchKernelGGL(k_render_fwd_seg<7>, dim3(blocks), dim3(64 * 7), 0, s, a); break;
        default: hipLaunchKernelGGL(k_render_fwd_seg<8>, dim3(blocks), dim3(64 * 8), 0, s, a); break;
        }
    }
    else hipLaunchKernelGGL(k_render_fwd, dim3(blocks), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t gft_launch_render_bwd(hipStream_t s, const gft_config& c, const gft_backward_io& io, const GeomView& g,
                                 const ImgView& im, const BinView& b, bool lazy, uint32_t cap)
{
    RenderBwdArgs a;
    a.ctrl = im.ctrl; a.cap = cap;
    a.W = c.W; a.H = c.H;
    a.gx = (c.W + GFT_TILE_X - 1) / GFT_TILE_X;
    const int gy = (c.H + GFT_TILE_Y - 1) / GFT_TILE_Y;
    a.T = a.gx * gy;
    a.ranges = im.ranges; a.point_list = b.point_list; a.rec_a = g.rec_a; a.rec_b = g.rec_b;
    a.bg = io.bg; a.bsc = c.bg_stride_c; a.bsy = c.bg_stride_y; a.bsx = c.bg_stride_x;
    a.dc_offset = c.dc_offset;
    a.pix_state = im.pix_state; a.quad_max = im.tile_max;
    a.g_color = io.dL_dout_color; a.g_phasor = io.dL_dout_phasor; a.g_depth = io.dL_dout_depth;
    a.g_acc = io.dL_dout_acc; a.g_dd = io.dL_dout_depth_distortion;
    a.acc = io.acc;
    a.det = io.det_partials;
    static const int order_on = [] { const char* e = getenv("GFT_BWD_ORDER"); return e ? atoi(e) != 0 : 1; }();
    a.order = order_on ? im.tile_order : nullptr;
    // with the lazy sort the forward's k_tile_tail launch has computed the order (unless a quadrant was flagged: the
    // deepest contributors were not final then; those frames walk in tile order)
    a.order_ok = lazy ? im.ctrl + GFT_CTRL_ORDER_OK : nullptr;
    a.order_fwd = lazy ? im.tile_cursor : nullptr;          // (tile-pull binning: see gft_launch_appearance)
    a.order_fwd_ok = im.ctrl + GFT_CTRL_FWDORDER;
    if (order_on && !lazy) hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, s, a.T, im.tile_max, im.tile_order);
    a.pix_sums = im.pix_sums;
    static const int split_on = [] { const char* e = getenv("GFT_BWD_SPLIT"); return e ? atoi(e) != 0 : 1; }();
    // segments need the forward's snapshots (written when a backward was announced) and, with a lazily sorted list, the
    // length of its sorted head
    a.nseg = gft_bwd_segments((size_t)a.T);
    a.split = split_on && a.nseg > 1 && c.want_backward;
    if (!a.split) a.nseg = 1;
    a.snaps = im.snaps;
    a.front_len = lazy ? im.front_len : nullptr;
    const int blocks = a.nseg * 32 * ((a.T + 7) / 8);
    hipLaunchKernelGGL(k_render_bwd, dim3(blocks), dim3(64), 0, s, a);
    if (a.det)
        hipLaunchKernelGGL(k_acc_reduce_det, dim3(1), dim3(1024), 0, s, a.T, im.ranges, im.tile_max, b.point_list,
                           a.det, io.acc, im.ctrl, cap);
    return hipGetLastError();
}
