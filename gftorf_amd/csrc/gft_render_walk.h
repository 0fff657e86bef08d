// gft_render_walk.h -- the forward blend's per-quadrant walk and its helpers (reference K6,
// RAST/cuda_rasterizer/forward.cu:424-676), shared by k_render.hip and k_pull.hip.  See k_render.hip for the design notes.
#pragma once
#include "gft_internal.h"

struct RenderFwdArgs {
    int W, H, gx, T;
    const uint2* __restrict__ ranges;
    const uint32_t* __restrict__ point_list;
    const float4* __restrict__ rec_a;
    const float4* __restrict__ rec_b;
    const float* __restrict__ bg;
    int64_t bsc, bsy, bsx;
    float dc_offset;
    float4* __restrict__ pix_state;
    float4* __restrict__ pix_sums;
    uint32_t* __restrict__ quad_max;
    float* out_color; float* out_phasor; float* out_depth; float* out_normal; float* out_acc;
    float* out_entropy; float* out_dd; float* out_ad; float* out_distribution;
    float* pixels;
    const uint32_t* __restrict__ ctrl;   // NULL: no instance-count check
    uint32_t cap;
    // tile-pull binning (k_pull.hip): `ranges` holds the sorted head of the tile's list; tile_cut[tile] != GFT_NO_TAIL: the
    // list goes on behind it (completed by k_tail_build for the tiles with a flagged quadrant).  NULL: whole-frame binning
    const uint32_t* __restrict__ tile_cut;
    float4* __restrict__ snaps;               // blend-state snapshots for the backward (NULL: no backward follows)
    int nsnap;                                // snapshots per quadrant (list positions 256, 512, ...)
    // lazy sort (k_binning.hip, k_tile_front): only the head of every id list is sorted
    const uint32_t* __restrict__ front_len;   // NULL: lists are sorted whole
    uint32_t* __restrict__ unit_flag;
    uint32_t* nflag;
    float4* __restrict__ resume_state;
    int resume;                               // second pass: continue the flagged quadrants behind the head
    // tile-pull binning with a caller-kept schedule (gft_forward_io.tile_hints): byte v = did quadrant v walk past where a
    // normal sorted head ends -- it flagged, or (a hinted tile's long head or whole list: more than 1024 sorted entries) its
    // deepest contributor lies beyond GFT_HEAD_TARGET
    uint8_t* __restrict__ hint_out;
    // first pass with a caller-kept schedule: the heavy-first tile order k_appearance derived for this frame (NULL: none)
    const uint32_t* __restrict__ fwd_order;
    const uint32_t* __restrict__ fwd_order_ok;
    // the caller's tile_weights (may be NULL): every quadrant leaves its walk length there for the camera's next frame (the
    // first pass: a flagged quadrant the longest there is, until its resume pass writes the final one), word 4 T = "valid"
    uint32_t* __restrict__ weights_out;
};

// the arguments of one forward blend pass (lazy: 0 = lists sorted whole, 1 = first pass over the sorted heads, 2 = resume pass);
// k_render.hip
RenderFwdArgs gft_render_fwd_args(const gft_config& c, const gft_forward_io& io, const GeomView& g, const ImgView& im,
                                  const BinView& b, bool check_cap, uint32_t cap, int lazy, bool pull);

namespace {

#define RB 64                // splats per staged batch = lanes per wave

// exp(x) for x <= 0 through the hardware exp2: |rel err| < 1e-6 on the range that
// can pass the 1/255 alpha threshold; forward and backward use the same function.
__device__ __forceinline__ float gft_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// ballot straight from the compare (HIP's __ballot(int) costs a v_cndmask + v_cmp per call)
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// select by a wave-uniform 64-bit lane mask held in SGPRs (bit set -> a): one v_cndmask, no compare
__device__ __forceinline__ float sel_mask(unsigned long long m, float a, float b)
{
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
__device__ __forceinline__ uint32_t sel_mask(unsigned long long m, uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}

__device__ __forceinline__ uint64_t to_sgpr(unsigned long long m)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)m), hi = __builtin_amdgcn_readfirstlane((uint32_t)(m >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// Bounding box {x0, y0, x1 - x0, y1 - y0} (pixel centres) of the pixels of mask m (lane = 8 y + x, m != 0, wave-uniform:
// scalar work) inside the quadrant whose first pixel is (qx0, qy0): the pixels that can still take a splat.  A batch of
// the list is culled against it -- late in a quadrant's walk a few open pixels keep the wave going, and most entries
// that reach the quadrant do not reach them.
__device__ __forceinline__ float4 box_of_mask(unsigned long long m, int qx0, int qy0)
{
    uint32_t c = (uint32_t)(m | (m >> 32));
    c |= c >> 16;
    c |= c >> 8;
    c &= 0xffu;
    const int x0 = __builtin_ctz(c), x1 = 31 - __builtin_clz(c);
    const int y0 = (int)(__builtin_ctzll(m) >> 3), y1 = (int)((63 - __builtin_clzll(m)) >> 3);
    return make_float4((float)(qx0 + x0), (float)(qy0 + y0), (float)(x1 - x0), (float)(y1 - y0));
}

// Stage splat `id` into LDS slot `slot`; returns whether it can reach a pixel centre of the rectangle `box`.
__device__ __forceinline__ bool stage_splat(uint32_t id, int slot, const float4* __restrict__ rec_a,
                                            const float4* __restrict__ rec_b, float4* sA, float4* sB,
                                            const float4& box)
{
    const float4 a0 = rec_a[2 * id], a1 = rec_a[2 * id + 1];
    sA[2 * slot] = a0;
    sA[2 * slot + 1] = a1;
    sB[2 * slot] = rec_b[2 * id];
    sB[2 * slot + 1] = rec_b[2 * id + 1];
    return gft_splat_reaches_box(a0, a1, box.x, box.y, box.z, box.w);
}


// The forward blend's walk of one 8x8 quadrant (one wave): list positions [0, head) in the first pass, [head, full) of a
// flagged quadrant in the resume pass.  A function of its own, in a header, because two kernels run it: k_render_fwd
// (k_render.hip: both passes of whole-frame binning, the first pass of tile-pull binning) and k_tail_build (k_pull.hip: the
// tail builder's workgroup resumes its tile's flagged quadrants itself as soon as the list is complete -- no launch in
// between).  Both translation units compile it with the same contraction setting (k_pull.hip includes this header in
// front of the one that switches contraction off) and without the SLP vectoriser: the same arithmetic, bit for bit.
// `sA`, `sB`: RB * 2 float4 of LDS each, this wave's own.
__device__ __forceinline__ void render_fwd_walk(const RenderFwdArgs& a, const int v, const int lane, float4* sA, float4* sB)
{
    auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    const int tile = v >> 2, quad = v & 3;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int qx0 = tx * GFT_TILE_X + (quad & 1) * 8, qy0 = ty * GFT_TILE_Y + (quad >> 1) * 8;
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const float pxf = (float)px, pyf = (float)py;
    // (resume pass inside k_tail_build: the tile's range was rewritten by this very workgroup a moment ago -- read it past
    // the L1, which may still hold the line as the workgroup read it before)
    uint2 range;
    if (a.resume) {
        range.x = __hip_atomic_load(&a.ranges[tile].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        range.y = __hip_atomic_load(&a.ranges[tile].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        range = a.ranges[tile];
    }
    const int full = (int)(range.y - range.x);
    const int head = a.front_len ? (int)a.front_len[tile] : full;
    // first pass: list positions [0, head); resume pass: [head, full) of the flagged quadrants (whole-frame binning: the
    // tail k_tile_tail has sorted; tile-pull binning: the culled tail k_tail_build has appended, `ranges` now names the
    // completed list)
    if (a.resume && a.unit_flag[v] == 0u) return;
    const bool more = a.tile_cut != nullptr && a.tile_cut[tile] != GFT_NO_TAIL;   // the list goes on behind what `ranges` holds
    const int begin = a.resume ? head : 0;
    const int total = a.resume ? full : head;
    // Depth distortion is formed from the sums of w (z - zref) and w (z - zref)^2, zref = NDC depth of the tile's nearest
    // Gaussian: A D2 - D^2 does not depend on the shift, but its two terms cancel to (depth spread / depth)^2 of their
    // size -- around zref they are small to begin with (a scene in a narrow depth range kept 1 significant digit
    // of the plane without the shift, and could go negative).  The backward uses the same shift.
    // (list position 0 = the tile's nearest Gaussian in every flow: the same zref, bit-identical sums)
    const float zref = full > 0 ? a.rec_a[2 * a.point_list[range.x] + 1].z : 0.0f;
    const size_t pix_i = inside ? (size_t)a.W * py + px : 0;

    // Predicates are wave-uniform 64-bit lane masks: every ballot below takes a single compare,
    // the combinations are scalar ALU work and the selects read the masks from SGPRs.
    unsigned long long done_m = ~wave_ballot(inside);     // pixels outside the image never blend
    float T = 1.0f;
    uint32_t last_contributor = 0;
    float C0 = 0, C1 = 0, C2 = 0;
    float PR = 0, PI = 0, PA = 0;     // ToF phasor on its (R, I, Am) basis
    float Dd = 0, A = 0, DD_D = 0, DD_D2 = 0;
    float WD0 = 0, WD1 = 0, WD2 = 0;
    if (a.resume) {
        bool was_done = true;
        if (inside) {
            const float4 s0 = a.resume_state[4 * pix_i], s1 = a.resume_state[4 * pix_i + 1];
            const float4 s2 = a.resume_state[4 * pix_i + 2], s3 = a.resume_state[4 * pix_i + 3];
            T = s0.x; C0 = s0.y; C1 = s0.z; C2 = s0.w;
            PR = s1.x; PI = s1.y; PA = s1.z; Dd = s1.w;
            A = s2.x; DD_D = s2.y; DD_D2 = s2.z; WD0 = s2.w;
            WD1 = s3.x; WD2 = s3.y; last_contributor = __float_as_uint(s3.z);
            was_done = s3.w != 0.f;
        }
        done_m = wave_ballot(was_done);
    }

    // (zref and a resumed blend state are waited for HERE: left to their first use the waits sit inside the entry loop,
    // where the in-order counter of outstanding memory operations makes them wait for the previous batch's pixel-count
    // atomics and snapshot stores as well)
    asm volatile("" : : "v"(zref), "v"(T), "v"(C0), "v"(WD1));
    // (the ids of a batch are asked for one batch ahead: a quadrant on the scene's silhouette walks thousands of entries
    // of which few reach it -- its batches are two dependent memory round trips and little else, this removes one)
    uint32_t id_next = (begin + lane < total) ? a.point_list[range.x + (uint32_t)(begin + lane)] : 0u;
    for (int base = begin; base < total; base += RB) {
        // all 64 pixels finished -> the rest of the list is never used
        if (done_m == ~0ull) break;
        // Blend state in front of list entries 256, 512, ...: the backward cuts a deep quadrant's walk there and gives
        // every segment to a wave of its own (first pass only: its batches start at multiples of 64 from entry 0)
        if (a.snaps && !a.resume && base > 0 && (base & (GFT_SEG_LEN - 1)) == 0 && base / GFT_SEG_LEN <= a.nsnap) {
            float4* sp = a.snaps + ((size_t)v * a.nsnap + (base / GFT_SEG_LEN - 1)) * (GFT_SNAP_F4 * 64) + lane;
            sp[0] = make_float4(T, C0, C1, C2);
            sp[64] = make_float4(PR, PI, PA, Dd);
            sp[128] = make_float4(A, DD_D, DD_D2, 0.f);
        }
        const int n = min(RB, total - base);
        bool reach = false;
        uint32_t my_id = 0;
        uint32_t cnt = 0;                        // lane j: pixels of this quadrant that blend splat j of the batch
        // the pixels that are still open (not all are done: checked above); while most are, the box is the quadrant
        const float4 box = __popcll(~done_m) <= 24 ? box_of_mask(~done_m, qx0, qy0) : make_float4((float)qx0, (float)qy0, 7.f, 7.f);
        wave_sync();                             // previous batch has read LDS
        {
            const uint32_t id = id_next;
            if (base + RB + lane < total) id_next = a.point_list[range.x + (uint32_t)(base + RB + lane)];
            if (lane < n) {
                my_id = id;
                reach = stage_splat(id, lane, a.rec_a, a.rec_b, sA, sB, box);
            }
        }
        uint64_t m = to_sgpr(wave_ballot(reach));
        wave_sync();

        // one list entry (splat j of the batch, its two LDS records already in registers)
        auto blend = [&](const int j, const float4& a0, const float4& a1, const float4& b0, const float4& b1) {
            const float dx = a0.x - pxf, dy = a0.y - pyf;
            const float power = -0.5f * (a0.z * dx * dx + a1.x * dy * dy) - a0.w * dx * dy;
            const float alpha = fminf(0.99f, a1.y * gft_exp(power));
            const unsigned long long vm = wave_ballot(!(power > 0.0f)) & wave_ballot(!(alpha < 1.0f / 255.0f)) & ~done_m;
            if (vm == 0ull) return;                          // wave-uniform skip
            const float test_T = T * (1 - alpha);
            const unsigned long long tm = vm & wave_ballot(test_T < 0.0001f);   // saturated here: splat not blended
            const unsigned long long cm = vm & ~tm;
            done_m |= tm;
            if (cm != 0ull) {
                // Branch-free blend: lanes that do not take this splat use alpha = 0, which adds
                // exact zeros and leaves T unchanged.
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
                // depth distortion: the reference adds w (z^2 A - 2 z D + D2) per splat (A, D, D2 = sums
                // over the splats in front, forward.cu:604-611), which telescopes to
                // sum_{i>j} w_i w_j (z_i - z_j)^2 = A D2 - D^2 of the final sums: formed once per pixel
                const float z = a1.z - zref;
                const float wz = w * z;
                DD_D += wz;
                DD_D2 = fmaf(wz, z, DD_D2);
                A += w;
                T = sel_mask(cm, test_T, T);         // T (1 - alpha) for the lanes that blend
                last_contributor = sel_mask(cm, (uint32_t)(base + j + 1), last_contributor);
                // pixels[id] += 1 for every contributing pixel: wave popcount, parked in lane j
                {
                    const uint32_t pc = (uint32_t)__popcll(cm);
                    // (gfx9: one SGPR per VALU op on the constant bus, the lane select goes through m0)
                    uint32_t m0_keep;
                    asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
                                 : "+v"(cnt), "=&s"(m0_keep) : "s"(pc), "s"(j));
                }
            }
        };
        // The records of the NEXT entry are read from LDS while the current one is blended (a wave's time per entry is its
        // dependent chain: LDS read -> alpha -> test -> LDS read -> sums; two entries alternate between two register sets)
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

    // The sorted head is used up, pixels are still unsaturated and the list goes on: park the blend
    // state and ask for the tail (k_tile_tail sorts it, the resume pass continues from here).
    if (!a.resume && (head < full || more) && done_m != ~0ull) {
        if (inside) {
            const bool is_done = (done_m >> lane) & 1ull;
            a.resume_state[4 * pix_i] = make_float4(T, C0, C1, C2);
            a.resume_state[4 * pix_i + 1] = make_float4(PR, PI, PA, Dd);
            a.resume_state[4 * pix_i + 2] = make_float4(A, DD_D, DD_D2, WD0);
            a.resume_state[4 * pix_i + 3] = make_float4(WD1, WD2, __uint_as_float(last_contributor), is_done ? 1.f : 0.f);
        }
        if (lane == 0) {
            a.unit_flag[v] = gft_flag_word(~done_m);     // (pixels outside the image count as done)
            atomicAdd(a.nflag, 1u);
        }
    }

    if (inside) {
        const size_t HW = (size_t)a.H * a.W;
        const size_t pix = (size_t)a.W * py + px;
        a.pix_state[pix] = make_float4(T, __uint_as_float(last_contributor), DD_D, DD_D2);
        a.pix_sums[2 * pix] = make_float4(C0, C1, C2, PR);
        a.pix_sums[2 * pix + 1] = make_float4(PI, PA, Dd, A);
        const float* bgp = a.bg + (int64_t)py * a.bsy + (int64_t)px * a.bsx;
        const float g0 = bgp[0], g1 = bgp[a.bsc], g2 = bgp[2 * a.bsc], g3 = bgp[3 * a.bsc];
        const float g4 = bgp[4 * a.bsc], g5 = bgp[5 * a.bsc], g6 = bgp[6 * a.bsc];
        a.out_color[pix] = C0 + T * g0;
        a.out_color[HW + pix] = C1 + T * g1;
        a.out_color[2 * HW + pix] = C2 + T * g2;
        // phasor planes share background planes 0..6, weighted by T (not T^2)
        // planes 3..6 = (+-cos + dc, +-sin + dc) A/d^2 blended = +-PR + dc PA, +-PI + dc PA
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
        // planes the reference allocates zero-filled and never writes
        a.out_normal[pix] = 0.f; a.out_normal[HW + pix] = 0.f; a.out_normal[2 * HW + pix] = 0.f;
        a.out_entropy[pix] = 0.f;
        a.out_ad[pix] = 0.f;
    }
    // deepest contributor of the quadrant: where its backward starts
    uint32_t mx = last_contributor;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
    if (lane == 0) a.quad_max[v] = mx;
    if (a.weights_out && lane == 0) {
        const bool open = !a.resume && (head < full || more) && done_m != ~0ull;
        a.weights_out[v] = open ? 0xfffffffeu : mx;
        if (v == 0 && !a.resume) a.weights_out[4 * a.T] = 1u;
    }
    if (a.hint_out && !a.resume && lane == 0) {
        const bool flagged = (head < full || more) && done_m != ~0ull;
        a.hint_out[v] = (flagged || (head > 1024 && mx > GFT_HEAD_TARGET)) ? 1 : 0;
    }
}

}  // namespace
