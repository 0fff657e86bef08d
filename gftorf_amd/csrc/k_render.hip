// k_render.hip -- per-tile alpha blending, forward and backward (gfx950, wave64).
//
// One 256-thread workgroup (4 waves) per 16x16 tile; each wave owns an 8x8 pixel
// quadrant so that a splat's footprint is wave-coherent.  The tile's depth-sorted
// splat list is staged through LDS in batches of 256 packed records (80 B per
// splat: rec_a 32 B + rec_b 48 B, gathered with 16-byte loads).  While staging,
// every splat's alpha >= 1/255 ellipse is bounded by a box and tested against the
// four quadrants; the four ballots per staging wave give each wave a 256-bit mask
// of the splats that can reach it, walked with scalar bit scans, so a wave only
// touches splats that may contribute to its quadrant and reads their records as
// LDS broadcasts.
//
// forward  (reference K6, RAST/cuda_rasterizer/forward.cu:424-676): front-to-back
//   blend of colour(3, w = a*T), ToF phasor(7, w = a*T^2), distance, accumulation,
//   depth distortion and the first-hit triple.  The per-Gaussian `pixels` counter
//   is reduced wave -> LDS -> one global atomic per (tile, splat) instead of one
//   per (pixel, splat).
// backward (reference K7, backward.cu:609-889): back-to-front, starting at the
//   tile's deepest contributor (stored by the forward) instead of the list end.
//   The 18 per-(pixel, splat) float atomics of the reference become: a
//   v_permlane32_swap / v_permlane16_swap / DPP reduction tree (18 values -> 5
//   registers), 5 ds_add_f32 into a per-batch LDS table, and one coalesced global
//   atomic burst per (tile, splat).
#include "gft_internal.h"

namespace {

#define GFT_BATCH 256
#define ACC_LDS_STRIDE 20

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int tile_of_block(int b, int T)
{
    // blocks are dealt round-robin over the 8 XCDs: give every XCD one contiguous
    // run of tiles so neighbouring tiles (which share splats) share an L2
    const int chunk = (T + 7) >> 3;
    return (b & 7) * chunk + (b >> 3);
}

// exp(x) for x <= 0 through the hardware exp2: |rel err| < 1e-6 on the range that
// can pass the 1/255 alpha threshold; forward and backward use the same function.
__device__ __forceinline__ float gft_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

__device__ __forceinline__ uint64_t uniform_mask(const uint64_t* p)
{
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(v.x), hi = __builtin_amdgcn_readfirstlane(v.y);
    return ((uint64_t)hi << 32) | lo;
}

// Stage splat `id` into LDS slot `slot` and return its quadrant-reach bits.
__device__ __forceinline__ uint32_t stage_splat(uint32_t id, int slot, const float4* __restrict__ rec_a,
                                                const float4* __restrict__ rec_b, float4* sA, float4* sB,
                                                float tile_x0, float tile_y0)
{
    const float4 a0 = rec_a[2 * id], a1 = rec_a[2 * id + 1];
    sA[2 * slot] = a0;
    sA[2 * slot + 1] = a1;
    sB[3 * slot] = rec_b[3 * id];
    sB[3 * slot + 1] = rec_b[3 * id + 1];
    sB[3 * slot + 2] = rec_b[3 * id + 2];
    // alpha = min(0.99, o*exp(power)) >= 1/255  <=>  power >= -tau, tau = ln(255 o):
    // the pixels that can blend this splat lie in the ellipse q(d) <= 2 tau, whose bounding
    // box has half extents sqrt(2 tau cov_xx), sqrt(2 tau cov_yy) with cov = conic^-1.
    const float ca = a0.z, cb = a0.w, cc = a1.x, op = a1.y;
    const float det = ca * cc - cb * cb;
    const float tau = __logf(255.0f * op);
    uint32_t bits = 0xfu;
    if (!(tau > 0.0f)) {
        bits = 0;                       // opacity <= 1/255: can never pass the alpha test
    } else if (det > 0.0f && ca > 0.0f && cc > 0.0f) {
        const float inv = 2.0f * tau / det;
        const float ex = sqrtf(inv * cc) * 1.0005f + 0.02f;   // margins keep the box conservative
        const float ey = sqrtf(inv * ca) * 1.0005f + 0.02f;
        const float lx = a0.x - ex - tile_x0, hx = a0.x + ex - tile_x0;
        const float ly = a0.y - ey - tile_y0, hy = a0.y + ey - tile_y0;
        const bool x_lo = hx >= 0.0f && lx <= 7.0f, x_hi = hx >= 8.0f && lx <= 15.0f;
        const bool y_lo = hy >= 0.0f && ly <= 7.0f, y_hi = hy >= 8.0f && ly <= 15.0f;
        bits = (x_lo && y_lo ? 1u : 0u) | (x_hi && y_lo ? 2u : 0u) | (x_lo && y_hi ? 4u : 0u) | (x_hi && y_hi ? 8u : 0u);
    }
    return bits;
}

// masks[q*4 + w] = ballot over staging wave w of "splat reaches quadrant q"
__device__ __forceinline__ void publish_masks(uint32_t bits, int wave, int lane, uint64_t* sMask)
{
    const unsigned long long m0 = __ballot(bits & 1u), m1 = __ballot(bits & 2u);
    const unsigned long long m2 = __ballot(bits & 4u), m3 = __ballot(bits & 8u);
    if (lane == 0) {
        sMask[0 * 4 + wave] = m0;
        sMask[1 * 4 + wave] = m1;
        sMask[2 * 4 + wave] = m2;
        sMask[3 * 4 + wave] = m3;
    }
}

struct RenderFwdArgs {
    int W, H, gx, T;
    const uint2* __restrict__ ranges;
    const uint32_t* __restrict__ point_list;
    const float4* __restrict__ rec_a;
    const float4* __restrict__ rec_b;
    const float* __restrict__ bg;
    int64_t bsc, bsy, bsx;
    float4* __restrict__ pix_state;
    uint32_t* __restrict__ tile_max;
    float* out_color; float* out_phasor; float* out_depth; float* out_normal; float* out_acc;
    float* out_entropy; float* out_dd; float* out_ad; float* out_distribution;
    float* pixels;
};

__global__ __launch_bounds__(GFT_BLOCK) void k_render_fwd(RenderFwdArgs a)
{
    __shared__ float4 sA[GFT_BATCH * 2];
    __shared__ float4 sB[GFT_BATCH * 3];
    __shared__ uint32_t sId[GFT_BATCH];
    __shared__ uint32_t sCnt[GFT_BATCH];
    __shared__ uint64_t sMask[16];
    __shared__ uint32_t sMax;

    const int tile = tile_of_block(blockIdx.x, a.T);
    if (tile >= a.T) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int px = tx * GFT_TILE_X + (wave & 1) * 8 + (lane & 7);
    const int py = ty * GFT_TILE_Y + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const float pxf = (float)px, pyf = (float)py;
    const float tile_x0 = (float)(tx * GFT_TILE_X), tile_y0 = (float)(ty * GFT_TILE_Y);
    const uint2 range = a.ranges[tile];
    const int total = (int)(range.y - range.x);
    const int rounds = (total + GFT_BATCH - 1) / GFT_BATCH;
    if (tid == 0) sMax = 0;

    bool done = !inside;
    float T = 1.0f;
    uint32_t last_contributor = 0;
    float C0 = 0, C1 = 0, C2 = 0;
    float P0 = 0, P1 = 0, P2 = 0, P3 = 0, P4 = 0, P5 = 0, P6 = 0;
    float Dd = 0, A = 0, DD = 0, DD_D = 0, DD_D2 = 0;
    float WD0 = 0, WD1 = 0, WD2 = 0;
    bool first = true;

    int todo = total;
    for (int i = 0; i < rounds; i++, todo -= GFT_BATCH) {
        // all pixels of the tile finished -> the rest of the list is never used
        if (__syncthreads_and(done)) break;
        const int n = min(GFT_BATCH, todo);
        uint32_t bits = 0;
        if (tid < n) {
            const uint32_t id = a.point_list[range.x + i * GFT_BATCH + tid];
            sId[tid] = id;
            sCnt[tid] = 0;
            bits = stage_splat(id, tid, a.rec_a, a.rec_b, sA, sB, tile_x0, tile_y0);
        }
        publish_masks(bits, wave, lane, sMask);
        __syncthreads();

        bool wave_live = __ballot(!done) != 0ull;
        for (int s = 0; s < 4 && wave_live; s++) {
            uint64_t m = uniform_mask(&sMask[wave * 4 + s]);
            while (m) {
                const int j = s * 64 + (int)__builtin_ctzll(m);
                m &= m - 1;
                bool contrib = false;
                if (!done) {
                    const float4 a0 = sA[2 * j], a1 = sA[2 * j + 1];
                    const float dx = a0.x - pxf, dy = a0.y - pyf;
                    const float power = -0.5f * (a0.z * dx * dx + a1.x * dy * dy) - a0.w * dx * dy;
                    const float alpha = fminf(0.99f, a1.y * gft_exp(power));
                    if (!(power > 0.0f) && !(alpha < 1.0f / 255.0f)) {
                        const float test_T = T * (1 - alpha);
                        if (test_T < 0.0001f) {
                            done = true;
                        } else {
                            contrib = true;
                            const float4 b0 = sB[3 * j], b1 = sB[3 * j + 1], b2 = sB[3 * j + 2];
                            const float w = alpha * T;
                            const float w_p = alpha * T * T;
                            C0 += b0.x * w; C1 += b0.y * w; C2 += b0.z * w;
                            P0 += b0.w * w_p; P1 += b1.x * w_p; P2 += b1.y * w_p; P3 += b1.z * w_p;
                            P4 += b1.w * w_p; P5 += b2.x * w_p; P6 += b2.y * w_p;
                            const float dist = a1.w;
                            Dd += dist * w;
                            if (first) {
                                WD0 = alpha; WD1 = dist; WD2 = b1.y;
                                first = false;
                            }
                            const float z = a1.z;
                            DD += w * (z * z * A - 2.0f * z * DD_D + DD_D2);
                            DD_D += w * z;
                            DD_D2 += w * z * z;
                            A += alpha * T;
                            T = test_T;
                            last_contributor = (uint32_t)(i * GFT_BATCH + j + 1);
                        }
                    }
                }
                // pixels[id] += 1 for every contributing pixel: wave popcount -> LDS
                const unsigned long long cm = __ballot(contrib);
                if (cm != 0ull && lane == 0) atomicAdd(&sCnt[j], (uint32_t)__popcll(cm));
                if (__ballot(!done) == 0ull) {
                    wave_live = false;
                    break;
                }
            }
        }
        __syncthreads();
        if (tid < n) {
            const uint32_t cnt = sCnt[tid];
            if (cnt) atomicAdd(&a.pixels[sId[tid]], (float)cnt);
        }
    }

    if (inside) {
        const size_t HW = (size_t)a.H * a.W;
        const size_t pix = (size_t)a.W * py + px;
        a.pix_state[pix] = make_float4(T, __uint_as_float(last_contributor), DD_D, DD_D2);
        const float* bgp = a.bg + (int64_t)py * a.bsy + (int64_t)px * a.bsx;
        const float g0 = bgp[0], g1 = bgp[a.bsc], g2 = bgp[2 * a.bsc], g3 = bgp[3 * a.bsc];
        const float g4 = bgp[4 * a.bsc], g5 = bgp[5 * a.bsc], g6 = bgp[6 * a.bsc];
        a.out_color[pix] = C0 + T * g0;
        a.out_color[HW + pix] = C1 + T * g1;
        a.out_color[2 * HW + pix] = C2 + T * g2;
        // phasor planes share background planes 0..6, weighted by T (not T^2)
        a.out_phasor[pix] = P0 + T * g0;
        a.out_phasor[HW + pix] = P1 + T * g1;
        a.out_phasor[2 * HW + pix] = P2 + T * g2;
        a.out_phasor[3 * HW + pix] = P3 + T * g3;
        a.out_phasor[4 * HW + pix] = P4 + T * g4;
        a.out_phasor[5 * HW + pix] = P5 + T * g5;
        a.out_phasor[6 * HW + pix] = P6 + T * g6;
        a.out_depth[pix] = Dd;
        a.out_acc[pix] = A;
        a.out_dd[pix] = DD;
        a.out_distribution[pix] = WD0;
        a.out_distribution[HW + pix] = WD1;
        a.out_distribution[2 * HW + pix] = WD2;
        // planes the reference allocates zero-filled and never writes
        a.out_normal[pix] = 0.f; a.out_normal[HW + pix] = 0.f; a.out_normal[2 * HW + pix] = 0.f;
        a.out_entropy[pix] = 0.f;
        a.out_ad[pix] = 0.f;
    }
    // deepest contributor of the tile: where the backward starts
    if (last_contributor) atomicMax(&sMax, last_contributor);
    __syncthreads();
    if (tid == 0) a.tile_max[tile] = sMax;
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
    const float4* __restrict__ pix_state;
    const uint32_t* __restrict__ tile_max;
    const float* __restrict__ g_color; const float* __restrict__ g_phasor; const float* __restrict__ g_depth;
    const float* __restrict__ g_acc; const float* __restrict__ g_dd;
    float* acc;   // [P][GFT_ACC_STRIDE]
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

__device__ __forceinline__ float row_sum_to_lane15(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), GFT_DPP_ROW_SHR(1), 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), GFT_DPP_ROW_SHR(2), 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), GFT_DPP_ROW_SHR(4), 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), GFT_DPP_ROW_SHR(8), 0xf, 0xf, false));
    return v;
}

// Sum 18 per-lane values over the 64 lanes of the wave and add the 18 totals to
// row[0..17] in LDS.  Halving tree: 18 -> 9 registers (lane halves), 9 -> 5 (16-lane
// rows), then a 4-step DPP scan inside each row; lanes 15/31/47/63 own the totals.
__device__ __forceinline__ void wave_reduce18_add(const float* v, float* row, int lane)
{
    float s[9];
#pragma unroll
    for (int i = 0; i < 9; i++) s[i] = swap32_add(v[i], v[i + 9]);      // lo half: value i, hi half: value i+9
    float t[5];
#pragma unroll
    for (int i = 0; i < 4; i++) t[i] = swap16_add(s[i], s[i + 4]);      // rows: i, i+4, i+9, i+13
    t[4] = swap16_add(s[8], s[8]);                                      // rows 0,1: value 8; rows 2,3: value 17
#pragma unroll
    for (int i = 0; i < 5; i++) t[i] = row_sum_to_lane15(t[i]);
    if ((lane & 15) == 15) {
        const int r = lane >> 4;
        const int off = (r & 1) * 4 + (r >> 1) * 9;
#pragma unroll
        for (int i = 0; i < 4; i++) atomicAdd(&row[i + off], t[i]);
        if (!(r & 1)) atomicAdd(&row[8 + (r >> 1) * 9], t[4]);
    }
}

__global__ __launch_bounds__(GFT_BLOCK) void k_render_bwd(RenderBwdArgs a)
{
    __shared__ float4 sA[GFT_BATCH * 2];
    __shared__ float4 sB[GFT_BATCH * 3];
    __shared__ uint32_t sId[GFT_BATCH];
    __shared__ float sAcc[GFT_BATCH * ACC_LDS_STRIDE];
    __shared__ uint32_t sTouched[GFT_BATCH];
    __shared__ uint64_t sMask[16];

    const int tile = tile_of_block(blockIdx.x, a.T);
    if (tile >= a.T) return;
    const int tmax = (int)a.tile_max[tile];
    if (tmax == 0) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tx = tile % a.gx, ty = tile / a.gx;
    const int px = tx * GFT_TILE_X + (wave & 1) * 8 + (lane & 7);
    const int py = ty * GFT_TILE_Y + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const float pxf = (float)px, pyf = (float)py;
    const float tile_x0 = (float)(tx * GFT_TILE_X), tile_y0 = (float)(ty * GFT_TILE_Y);
    const uint32_t r0 = a.ranges[tile].x;
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
    const float ddelx_dx = 0.5f * a.W, ddely_dy = 0.5f * a.H;
    const float one_m_Tf = 1 - T_final;

    float T = T_final;
    float ar0 = 0, ar1 = 0, ar2 = 0;                                    // accum_rec colour
    float ap0 = 0, ap1 = 0, ap2 = 0, ap3 = 0, ap4 = 0, ap5 = 0, ap6 = 0;  // accum_rec phasor
    float ar_d = 0, ar_a = 0, ar_dd = 0;
    float last_alpha = 0;
    float lc0 = 0, lc1 = 0, lc2 = 0, lp0 = 0, lp1 = 0, lp2 = 0, lp3 = 0, lp4 = 0, lp5 = 0, lp6 = 0;
    float last_dist = 0, last_dL_dw = 0;

    const int rounds = (tmax + GFT_BATCH - 1) / GFT_BATCH;
    for (int i = 0; i < rounds; i++) {
        const int hi = tmax - i * GFT_BATCH;       // list indices [hi-n, hi) in descending order
        const int n = min(GFT_BATCH, hi);
        __syncthreads();                           // previous batch's flush has read LDS
        uint32_t bits = 0;
        if (tid < n) {
            const uint32_t id = a.point_list[r0 + (uint32_t)(hi - 1 - tid)];
            sId[tid] = id;
            sTouched[tid] = 0;
            bits = stage_splat(id, tid, a.rec_a, a.rec_b, sA, sB, tile_x0, tile_y0);
        }
        publish_masks(bits, wave, lane, sMask);
        {
            float4* z = reinterpret_cast<float4*>(sAcc) + tid * (ACC_LDS_STRIDE / 4);
#pragma unroll
            for (int k = 0; k < ACC_LDS_STRIDE / 4; k++) z[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();

        for (int s = 0; s < 4; s++) {
            uint64_t m = uniform_mask(&sMask[wave * 4 + s]);
            while (m) {
                const int j = s * 64 + (int)__builtin_ctzll(m);
                m &= m - 1;
                const int c = hi - 1 - j;               // list position of this splat
                const float4 a0 = sA[2 * j], a1 = sA[2 * j + 1];
                const float dx = a0.x - pxf, dy = a0.y - pyf;
                const float power = -0.5f * (a0.z * dx * dx + a1.x * dy * dy) - a0.w * dx * dy;
                const float G = gft_exp(power);
                const float alpha = fminf(0.99f, a1.y * G);
                const bool contrib = (c < n_contrib) && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
                if (__ballot(contrib) == 0ull) continue;   // wave-uniform skip

                float v[GFT_NUM_ACC];
#pragma unroll
                for (int k = 0; k < GFT_NUM_ACC; k++) v[k] = 0.f;
                if (contrib) {
                    const float4 b0 = sB[3 * j], b1 = sB[3 * j + 1], b2 = sB[3 * j + 2];
                    const float rcp_1ma = __builtin_amdgcn_rcpf(1.f - alpha);
                    T = T * rcp_1ma;
                    const float wc = alpha * T;          // dchannel_dcolor == dchannel_ddepth
                    const float wp = wc * T;             // dchannel_dphasor = alpha*T*T
                    const float one_m_la = 1.f - last_alpha;
                    float dL_dalpha = 0.f;

                    // colour
                    float dac = 0.f;
                    ar0 = last_alpha * lc0 + one_m_la * ar0; lc0 = b0.x; dac += (b0.x - ar0) * gc0; v[6] = wc * gc0;
                    ar1 = last_alpha * lc1 + one_m_la * ar1; lc1 = b0.y; dac += (b0.y - ar1) * gc1; v[7] = wc * gc1;
                    ar2 = last_alpha * lc2 + one_m_la * ar2; lc2 = b0.z; dac += (b0.z - ar2) * gc2; v[8] = wc * gc2;
                    dac *= T;

                    // ToF phasor (weight alpha*T^2)
                    float dap = 0.f;
                    const float one_m_la2 = one_m_la * one_m_la;
                    const float two_1ma = 2.f * (1.f - alpha);
                    ap0 = last_alpha * lp0 + one_m_la2 * ap0; lp0 = b0.w; dap += (b0.w - two_1ma * ap0) * gp0; v[9] = wp * gp0;
                    ap1 = last_alpha * lp1 + one_m_la2 * ap1; lp1 = b1.x; dap += (b1.x - two_1ma * ap1) * gp1; v[10] = wp * gp1;
                    ap2 = last_alpha * lp2 + one_m_la2 * ap2; lp2 = b1.y; dap += (b1.y - two_1ma * ap2) * gp2; v[11] = wp * gp2;
                    ap3 = last_alpha * lp3 + one_m_la2 * ap3; lp3 = b1.z; dap += (b1.z - two_1ma * ap3) * gp3; v[12] = wp * gp3;
                    ap4 = last_alpha * lp4 + one_m_la2 * ap4; lp4 = b1.w; dap += (b1.w - two_1ma * ap4) * gp4; v[13] = wp * gp4;
                    ap5 = last_alpha * lp5 + one_m_la2 * ap5; lp5 = b2.x; dap += (b2.x - two_1ma * ap5) * gp5; v[14] = wp * gp5;
                    ap6 = last_alpha * lp6 + one_m_la2 * ap6; lp6 = b2.y; dap += (b2.y - two_1ma * ap6) * gp6; v[15] = wp * gp6;
                    dap *= T * T;

                    // distance
                    const float dist = a1.w;
                    ar_d = last_alpha * last_dist + one_m_la * ar_d;
                    last_dist = dist;
                    float dad = (dist - ar_d) * gd;
                    v[16] = wc * gd;
                    dad *= T;

                    // accumulation
                    ar_a = last_alpha + one_m_la * ar_a;
                    float daa = (1.f - ar_a) * ga;
                    daa *= T;

                    // depth distortion
                    const float z = a1.z;
                    const float dL_dw = gdd * (z * z * one_m_Tf - 2.0f * z * wz_tot + wz2_tot);
                    ar_dd = last_alpha * last_dL_dw + one_m_la * ar_dd;
                    last_dL_dw = dL_dw;
                    float dadd = dL_dw - ar_dd;
                    v[17] = gdd * 2.0f * wc * (z * one_m_Tf - wz_tot);
                    dadd *= T;

                    last_alpha = alpha;

                    const float bgf = -T_final * rcp_1ma;
                    dL_dalpha += bgf * bg_dot;
                    dap += bgf * bg_dot_p;
                    dL_dalpha += dac;
                    dL_dalpha += dap;
                    dL_dalpha += dad;
                    dL_dalpha += daa;
                    dL_dalpha += dadd;

                    const float dL_dG = a1.y * dL_dalpha;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * a0.z - gdy * a0.w;
                    const float dG_ddely = -gdy * a1.x - gdx * a0.w;
                    v[0] = dL_dG * dG_ddelx * ddelx_dx;
                    v[1] = dL_dG * dG_ddely * ddely_dy;
                    v[2] = -0.5f * gdx * dx * dL_dG;
                    v[3] = -0.5f * gdx * dy * dL_dG;
                    v[4] = -0.5f * gdy * dy * dL_dG;
                    v[5] = G * dL_dalpha;
                }
                // 64 pixels -> one partial per value, accumulated in the batch's LDS table
                wave_reduce18_add(v, &sAcc[j * ACC_LDS_STRIDE], lane);
                if (lane == 0) sTouched[j] = 1;
            }
        }
        __syncthreads();

        // flush: 32 lanes per splat row, 18 of them active -> contiguous 72-byte bursts
        const int k = tid & 31;
        if (k < GFT_NUM_ACC) {
            for (int g = tid >> 5; g < n; g += GFT_BLOCK / 32) {
                if (sTouched[g]) {
                    const float val = sAcc[g * ACC_LDS_STRIDE + k];
                    if (val != 0.f) atomicAdd(&a.acc[(size_t)sId[g] * GFT_ACC_STRIDE + k], val);
                }
            }
        }
    }
}

}  // namespace

hipError_t gft_launch_render_fwd(hipStream_t s, const gft_config& c, const gft_forward_io& io, const GeomView& g,
                                 const ImgView& im, const BinView& b)
{
    RenderFwdArgs a;
    a.W = c.W; a.H = c.H;
    a.gx = (c.W + GFT_TILE_X - 1) / GFT_TILE_X;
    const int gy = (c.H + GFT_TILE_Y - 1) / GFT_TILE_Y;
    a.T = a.gx * gy;
    a.ranges = im.ranges; a.point_list = b.point_list; a.rec_a = g.rec_a; a.rec_b = g.rec_b;
    a.bg = io.bg; a.bsc = c.bg_stride_c; a.bsy = c.bg_stride_y; a.bsx = c.bg_stride_x;
    a.pix_state = im.pix_state; a.tile_max = im.tile_max;
    a.out_color = io.out_color; a.out_phasor = io.out_phasor; a.out_depth = io.out_depth;
    a.out_normal = io.out_normal; a.out_acc = io.out_acc; a.out_entropy = io.out_entropy;
    a.out_dd = io.out_depth_distortion; a.out_ad = io.out_amp_distortion;
    a.out_distribution = io.out_distribution; a.pixels = io.pixels;
    const int blocks = 8 * ((a.T + 7) / 8);
    hipLaunchKernelGGL(k_render_fwd, dim3(blocks), dim3(GFT_BLOCK), 0, s, a);
    return hipGetLastError();
}

hipError_t gft_launch_render_bwd(hipStream_t s, const gft_config& c, const gft_backward_io& io, const GeomView& g,
                                 const ImgView& im, const BinView& b)
{
    RenderBwdArgs a;
    a.W = c.W; a.H = c.H;
    a.gx = (c.W + GFT_TILE_X - 1) / GFT_TILE_X;
    const int gy = (c.H + GFT_TILE_Y - 1) / GFT_TILE_Y;
    a.T = a.gx * gy;
    a.ranges = im.ranges; a.point_list = b.point_list; a.rec_a = g.rec_a; a.rec_b = g.rec_b;
    a.bg = io.bg; a.bsc = c.bg_stride_c; a.bsy = c.bg_stride_y; a.bsx = c.bg_stride_x;
    a.pix_state = im.pix_state; a.tile_max = im.tile_max;
    a.g_color = io.dL_dout_color; a.g_phasor = io.dL_dout_phasor; a.g_depth = io.dL_dout_depth;
    a.g_acc = io.dL_dout_acc; a.g_dd = io.dL_dout_depth_distortion;
    a.acc = io.acc;
    const int blocks = 8 * ((a.T + 7) / 8);
    hipLaunchKernelGGL(k_render_bwd, dim3(blocks), dim3(GFT_BLOCK), 0, s, a);
    return hipGetLastError();
}
