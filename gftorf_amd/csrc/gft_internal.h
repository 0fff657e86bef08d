// gft_internal.h -- shared declarations of libgftorf_rast.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "gftorf_rast.h"

#include <atomic>
#include <cstdlib>

#define GFT_ALIGN 256

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device only: the opt-in is made once per
// (kernel, device) -- `done` holds one bit per device ordinal -- its error is returned, and two threads that make
// their first call together both set it (idempotent).
inline hipError_t gft_lds_opt_in(const void* kernel, size_t bytes, std::atomic<uint64_t>& done)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;
    if (bit && (done.load(std::memory_order_acquire) & bit)) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && bit) done.fetch_or(bit, std::memory_order_release);
    return e;
}
#define GFT_BLOCK 256            // threads per workgroup = one 16x16 tile = 4 waves
#define GFT_NUM_ACC 15           // accumulators actually used per Gaussian

// ---- scratch views -------------------------------------------------------
struct GeomView {
    float4* rec_a;      // [P][2]  {x,y,ca,cb} {cc,opacity,dist_ndc,dist}: every visible Gaussian
    float4* rec_b;      // [P][2]  {r,g,b,R} {I,Am,phase_sh,amp}: the Gaussians marked in `need` (all visible ones in the whole-frame binning mode)
    float* depth;       // [P]
    uint32_t* tiles;    // [P]
    ushort4* rect;      // [P] tile rectangle {x0,y0,x1,y1}; all zero when culled
    float4* dirgrad;    // [P][4] d(rgb)/d(dir) 9, d(phase,amp)/d(dir) 6, pad (forward with want_backward; as rec_b)
    uint8_t* clamped;   // [P]   (as rec_b)
    uint8_t* need;      // [P]   tile-pull binning: 1 = the Gaussian is in the sorted part of some tile list, its appearance records exist
    uint16_t* blockhist; // [ceil(P/4096)][GFT_BLOCKHIST_TILES] tile hits of every 4096-Gaussian block (T <= GFT_BLOCKHIST_TILES)
};

struct ImgView {
    float4* pix_state;    // [N] {final_T, n_contrib bits, w_z, w_z2}
    float4* pix_sums;     // [N][2] {C0, C1, C2, R}, {I, Am, dist, A}: final sums of the blend, without background
    float4* snaps;        // [4T][segments - 1][3][64] blend state of every quadrant in front of list entries 256, 512, ...
    uint2* ranges;        // [T]  the tile's id list [first, last) inside point_list
    uint32_t* tile_max;   // [T][4] deepest contributor per 8x8 quadrant
    uint32_t* ctrl;       // [GFT_CTRL_WORDS], directly followed by tile_cnt, tile_cut, super_tab (cleared together by k_preprocess_fwd)
    uint32_t* tile_cnt;   // [T]  instances per tile
    uint32_t* tile_cut;   // [T]  tile-pull binning: last depth bin inside the sorted head; GFT_NO_TAIL when the head is the whole list
    uint32_t* super_tab;  // [4][GFT_SUPER_CELLS] tile-pull binning: per (copy, supertile, depth slab) entry counters and scatter cursors; per (supertile, slab) entry count and list start
    uint32_t* tile_cursor;// [T]
    uint32_t* tile_order; // [T] tiles by backward weight, heaviest first
    uint32_t* front_len;  // [T] length of the sorted head of the tile's id list
    uint32_t* unit_flag;  // [4T] non-zero: the quadrant reached the end of the head unsaturated (gft_flag_word: with the box of those pixels)
    float4* resume_state; // [N][4] blend state of the pixels of flagged quadrants
};

struct BinView {
    uint64_t* keys;        // [cap] whole-frame binning: (depth bits << 32 | id) grouped by tile; tile-pull binning: the supertile entry lists
    uint32_t* point_list;  // [T * GFT_HEAD_SLOT + cap] whole-frame binning: ids per tile in [0, R); tile-pull binning: one head slot per
                           // tile, then the pool of the lists that were completed on demand (head copy + culled tail)
};

// ctrl words (uint32)
#define GFT_CTRL_TOTAL 0     // R = number of (Gaussian, tile) instances
#define GFT_CTRL_FLAGS 1     // bit0: prefiltered point culled
#define GFT_CTRL_MAXCNT 2    // longest tile list (whole-frame binning; 0 with tile-pull binning, which never forms the lists)
#define GFT_CTRL_NFLAG 4     // number of flagged quadrants (ran out of sorted entries before saturating)
#define GFT_CTRL_DONE 3      // finished workgroups of the count kernel (ticket for its fused scan)
#define GFT_CTRL_SEQ 3       // host mailbox only: sequence number, written last
#define GFT_CTRL_ENTRIES 5   // tile-pull binning: (Gaussian, supertile) entries
#define GFT_CTRL_POOLCUR 6   // tile-pull binning: ids taken from the pool of completed lists
#define GFT_CTRL_DONE2 7     // finished workgroups of k_tail_build (ticket for the backward's tile order)
#define GFT_CTRL_NHINT 8     // tile-pull binning: non-zero words of the caller's per-tile schedule (gft_forward_io.tile_hints), also in the mailbox
#define GFT_CTRL_RSUM 9      // tile-pull binning: R as summed by the supertile count pass
#define GFT_CTRL_FWDORDER 10 // this frame's forward has a heavy-first tile order of its own (from the caller's tile_weights), in tile_cursor
#define GFT_CTRL_ORDER_OK 11 // the forward computed the backward's heavy-first tile order
#define GFT_CTRL_WORDS 16
// status block of gft_forward_enqueue (a mailbox slot the caller owns): the sticky words, kept by the kernel that posts R
// (`post` = binning_instances + 1 of the frame, 0 = `mail` is no status block)
__device__ __forceinline__ void gft_status_sticky(uint32_t* mail, uint32_t post, uint32_t R)
{
    if (!post) return;
    mail[GFT_STATUS_CAP] = post - 1u;
    if (R > post - 1u) {
        __hip_atomic_fetch_add(&mail[GFT_STATUS_OVERFLOWS], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_fetch_max(&mail[GFT_STATUS_MAX_R], R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// Behind the ctrl words: first-level ticket counters of the count pass.  Its ~1000 workgroups are resident together and all
// draw a ticket at about the same time; returning atomics on ONE address are served one after the other (~10 ns each):
// the workgroup that draws the last ticket -- the one that scans -- waited ~10 us for its answer.  Workgroup b draws from
// counter b % GFT_TICKET_WORDS, the last of every group from the second-level counter ctrl[GFT_CTRL_DONE].
#define GFT_TICKET_WORDS 32
#define GFT_SUPER_MAX 1024   // supertiles (groups of S x S tiles) of the tile-pull binning
#define GFT_SLAB_MAX 16      // depth slabs per supertile list
#define GFT_SUPER_CELLS (GFT_SUPER_MAX * GFT_SLAB_MAX)
#define GFT_DEPTH_BINS 4096  // depth bins a supertile entry carries (12 bits); slab = top bits of the bin
#define GFT_HEAD_SLOT 2048u  // ids per tile head slot = longest sorted head
#define GFT_HEAD_DIRECT 2048u // tile-pull binning: lists up to this length are sorted whole in the first place
#define GFT_HEAD_TARGET 940u  // ... wanted length of the sorted head of a longer list
#define GFT_NO_TAIL 0xffffffffu
#define GFT_BLOCKHIST_TILES 2048
// binning workgroup shape (k_binning.hip): 16 waves per workgroup keep one CU busy on their own
#ifndef BIN_THREADS
#define BIN_THREADS 1024
#endif
#ifndef BIN_ITEMS
#define BIN_ITEMS 4                          // Gaussians per thread in count / scatter
#endif
#define BIN_CHUNK (BIN_THREADS * BIN_ITEMS)  // Gaussians per workgroup
#define GFT_SHORT_LIST_MAX 4096   // tile lists up to this length are sorted by one 256-thread workgroup

// Waves that may share the backward walk of one quadrant: its list is cut every GFT_SEG_LEN entries, where the forward
// left a snapshot of the blend state (up to 4 segments; fewer on frames with more than 16384 quadrants, where one wave
// per quadrant already fills the chip).
#ifndef GFT_SEG_LEN
#define GFT_SEG_LEN 256
#endif
inline int gft_bwd_segments(size_t T)
{
    // (2 ... 8 segments measure alike on the repeated metric frame, whose walks end inside the ~940-entry heads -- the
    // kernel is bound by VALU issue there, not by its chains.  Other views of the same scene have quadrants that walk
    // whole 2048-entry lists: with 4 segments their last one is 1280 entries long and sets the kernel's time
    // (views at the ends of the bench's arc: 180-190 us against 155-170 with 8 segments; over the 30 views 166 -> 155).
    // GFT_BWD_NSEG overrides for tuning.)
    static const int cap = [] { const char* e = getenv("GFT_BWD_NSEG"); const int v = e ? atoi(e) : 0; return v > 0 ? (v > 8 ? 8 : v) : 8; }();
    const size_t v = 4 * (T ? T : 1);
    const size_t n = 65536 / v;
    return n < 1 ? 1 : (n > (size_t)cap ? cap : (int)n);
}
// Waves that may share the FORWARD walk of one quadrant (k_render_fwd_seg): as many as keep all quadrants of the frame
// resident together -- 1024 SIMDs x 6 waves over 4 T quadrants; from 768 tiles on, one wave per quadrant fills the chip
// and the serial kernel runs.  GFT_FWD_SEG_WAVES overrides for tuning.
inline int gft_fwd_seg_waves(int T)
{
    static const int force = [] { const char* e = getenv("GFT_FWD_SEG_WAVES"); return e ? atoi(e) : 0; }();
    if (force > 0) return force > 8 ? 8 : force;
    const int n = 6144 / (4 * (T > 0 ? T : 1));
    return n < 1 ? 1 : (n > 8 ? 8 : n);
}
int gft_render_mode();      // -1: default (segments on under-filled frames), 0: one wave per quadrant, 1: segments (gft_set_render_mode / GFT_FWD_SEG)
// does the first pass of a T-tile frame run the segment-parallel kernel?
inline bool gft_fwd_segmented(int T) { return gft_render_mode() != 0 && gft_fwd_seg_waves(T) > 1; }
// The forward's heavy-first dealing (gft_forward_io.tile_weights) pays where the quadrant waves make few rounds over the chip's
// wave slots: measured -5..-9 us at 1200 tiles (the metric frame, 500 k Gaussians; -8 us per view over 30 cameras), +-0 on the fog frame, +65 us at 8160 tiles (5 M Gaussians at 1080p: 16
// rounds balance by themselves, and neighbouring tiles no longer share an XCD's L2).  Off above 4096 tiles.
#define GFT_FWD_ORDER_MAX_TILES 4096
inline bool gft_fwd_ordered(int T) { return !gft_fwd_segmented(T) && T <= GFT_FWD_ORDER_MAX_TILES; }
#define GFT_SNAP_F4 3      // float4 per pixel and snapshot: {T, C0, C1, C2} {PR, PI, PA, Dd} {A, DD_D, DD_D2, -}

void gft_compute_layout(int32_t P, int32_t W, int32_t H, int64_t R, gft_layout* L);
GeomView gft_geom_view(void* base, const gft_layout& L);
ImgView gft_img_view(void* base, const gft_layout& L);
BinView gft_bin_view(void* base, const gft_layout& L);

int gft_fail(const char* fmt, ...);
// zero fill by a kernel (gft_api.hip: hipMemsetAsync nodes misbehave in replayed graphs on this platform)
hipError_t gft_zero_async(void* ptr, size_t bytes, hipStream_t s);
#define GFT_CHECK_HIP(expr)                                                          \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess)                                                        \
            return gft_fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                            __FILE__, __LINE__);                                     \
    } while (0)

// ---- stage launchers (each enqueues on `s`, returns hipError_t) -----------
// defer_appearance: geometry only (rectangle, depth, radius, geometry record); the appearance records follow from
// gft_launch_appearance for the Gaussians the tile-pull binning marks in `need`
hipError_t gft_launch_preprocess_fwd(hipStream_t s, const gft_config& c, const gft_forward_io& io,
                                     const GeomView& g, const ImgView& im, uint32_t* mail, bool defer_appearance);
hipError_t gft_launch_appearance(hipStream_t s, const gft_config& c, const gft_forward_io& io, const GeomView& g,
                                 const ImgView& im, uint32_t cap);
// whole-frame binning (reference structure: every instance counted, scattered and sorted)
// (status_cap >= 0: `mail` is a status block of gft_forward_enqueue, the binning buffer holds that many instances: the poster
// also keeps the block's sticky words -- include/gftorf_rast.h, GFT_STATUS_OVERFLOWS)
hipError_t gft_launch_tile_count(hipStream_t s, const gft_config& c, const GeomView& g, const ImgView& im,
                                 uint32_t* mail, uint32_t seq, int64_t status_cap = -1);
hipError_t gft_launch_tile_scatter(hipStream_t s, const gft_config& c, const GeomView& g, const ImgView& im,
                                   const BinView& b, uint32_t cap, int64_t expect);
hipError_t gft_launch_tile_sort(hipStream_t s, const gft_config& c, int64_t max_tile_list, const ImgView& im,
                                const BinView& b, uint32_t cap, float* clear, size_t clear_bytes);
hipError_t gft_launch_tile_front(hipStream_t s, const gft_config& c, const ImgView& im, const BinView& b, uint32_t cap,
                                 float* clear, size_t clear_bytes);
hipError_t gft_launch_tile_tail(hipStream_t s, const gft_config& c, const GeomView& g, const ImgView& im,
                                const BinView& b, uint32_t cap, bool want_order);
hipError_t gft_launch_tile_sort_long(hipStream_t s, const gft_config& c, const ImgView& im, const BinView& b,
                                     uint32_t cap);
// tile-pull binning (k_binning.hip, k_tail.hip): ids to supertiles, every tile pulls and sorts the head of its list,
// lists are completed on demand for the tiles with a flagged quadrant
bool gft_tile_pull_ok(const gft_config& c);      // the frame's tile grid fits the supertile tables
bool gft_tail_resumes();                         // k_tail_build resumes its tiles' flagged quadrants itself (no resume launch)
struct SuperShape { int gx, gy, T, sshift, sgx, sgy, NS, K, kshift; uint32_t near_bits; int bin_shift; };
SuperShape gft_super_shape(const gft_config& c);
hipError_t gft_launch_super_bin(hipStream_t s, const gft_config& c, const GeomView& g, const ImgView& im, const BinView& b,
                                uint32_t* mail, uint32_t seq, int pass, uint32_t cap, const uint32_t* hints = nullptr,
                                uint32_t* sched = nullptr, int64_t status_cap = -1);
size_t gft_cell_sched_words_of(const gft_config& c);
hipError_t gft_launch_tile_pull(hipStream_t s, const gft_config& c, const GeomView& g, const ImgView& im, const BinView& b,
                                uint32_t cap, float* clear, size_t clear_bytes, const uint32_t* hints, bool whole_lists);
hipError_t gft_launch_tail_build(hipStream_t s, const gft_config& c, const gft_forward_io& io, const GeomView& g,
                                 const ImgView& im, const BinView& b, uint32_t cap, bool want_order);
// lazy: 0 = lists sorted whole, 1 = first pass over the sorted heads, 2 = resume pass of the flagged quadrants
hipError_t gft_launch_render_fwd(hipStream_t s, const gft_config& c, const gft_forward_io& io,
                                 const GeomView& g, const ImgView& im, const BinView& b, bool check_cap, uint32_t cap,
                                 int lazy, bool pull, bool segmented = true);
hipError_t gft_launch_render_bwd(hipStream_t s, const gft_config& c, const gft_backward_io& io,
                                 const GeomView& g, const ImgView& im, const BinView& b, bool lazy, uint32_t cap);
hipError_t gft_launch_preprocess_bwd(hipStream_t s, const gft_config& c, const gft_backward_io& io,
                                     const GeomView& g);
hipError_t gft_launch_grads_rezero(hipStream_t s, const gft_config& c, const gft_backward_io& io);
hipError_t gft_launch_mark_visible(hipStream_t s, int32_t P, const float* means3D,
                                   const float* view, float near_n, float far_n, uint8_t* present);


// ---- device helpers ---------------------------------------------------------
#define GFT_DPP_ROW_SHR(n) (0x110 + (n))
#define GFT_DPP_ROW_BCAST15 0x142
#define GFT_DPP_ROW_BCAST31 0x143

// Sum over the 64 lanes of a wave; the total is valid in lane 63.
__device__ __forceinline__ float gft_wave_sum_to_lane63(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), GFT_DPP_ROW_SHR(1), 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), GFT_DPP_ROW_SHR(2), 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), GFT_DPP_ROW_SHR(4), 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), GFT_DPP_ROW_SHR(8), 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), GFT_DPP_ROW_BCAST15, 0xa, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), GFT_DPP_ROW_BCAST31, 0xc, 0xf, false));
    return v;
}

__device__ __forceinline__ uint32_t gft_wave_sum_u32_to_lane63(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, GFT_DPP_ROW_SHR(1), 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, GFT_DPP_ROW_SHR(2), 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, GFT_DPP_ROW_SHR(4), 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, GFT_DPP_ROW_SHR(8), 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, GFT_DPP_ROW_BCAST15, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, GFT_DPP_ROW_BCAST31, 0xc, 0xf, false);
    return v;
}

// Tile rectangle of a splat: same float expression order and truncation as the
// reference getRect (auxiliary.h:49-59) so preprocess and duplication agree.
__device__ __forceinline__ void gft_get_rect(float px, float py, int radius, int gx, int gy,
                                             int& x0, int& y0, int& x1, int& y1)
{
    const float r = (float)radius;
    x0 = min(gx, max(0, (int)((px - r) / 16.0f)));
    y0 = min(gy, max(0, (int)((py - r) / 16.0f)));
    x1 = min(gx, max(0, (int)((((px + r) + 16.0f) - 1.0f) / 16.0f)));
    y1 = min(gy, max(0, (int)((((py + r) + 16.0f) - 1.0f) / 16.0f)));
}

// Depth bin of the tile-pull binning: GFT_DEPTH_BINS equal steps of the depth's float bits from the near plane on -- a
// monotone integer function of the depth bits (positive floats order like their bits), no transcendental: keys of a
// lower bin are smaller than keys of a higher bin.
__device__ __forceinline__ uint32_t gft_depth_bin(uint32_t dbits, uint32_t near_bits, int shift)
{
    const uint32_t d = dbits > near_bits ? dbits - near_bits : 0u;
    const uint32_t b = d >> shift;
    return b < GFT_DEPTH_BINS ? b : GFT_DEPTH_BINS - 1u;
}

// Can a splat reach the 8x8 pixel quadrant whose first pixel centre is (qx0, qy0)?  a0 = {x, y, conic a, b},
// a1 = {conic c, opacity, ..}.  alpha = min(0.99, o*exp(power)) >= 1/255  <=>  power >= -tau, tau = ln(255 o), i.e. the
// pixels that can blend this splat lie in the ellipse q(u) = a ux^2 + 2 b ux uy + c uy^2 <= 2 tau around the centre.
// The quadrant's pixel centres span a rectangle; q is convex with its minimum at the centre, so its minimum over the
// rectangle sits on an edge facing the centre: two 1-D clamped minimisations.  (A bounding-box test passes 195 of 429
// walked splats per quadrant on the metric frame, this one 154; 144 really touch a pixel.)  Conservative: a splat that
// fails can blend into no pixel of the quadrant (forward.cu:536-548).
// (general form: the rectangle of pixel centres [bx0, bx0 + bw] x [by0, by0 + bh])
__device__ __forceinline__ bool gft_splat_reaches_box(const float4& a0, const float4& a1, float bx0, float by0, float bw, float bh)
{
    const float ca = a0.z, cb = a0.w, cc = a1.x, op = a1.y;
    const float det = ca * cc - cb * cb;
    const float tau = __logf(255.0f * op);
    if (!(tau > 0.0f)) return false;            // opacity <= 1/255: can never pass the alpha test
    if (!(det > 0.0f && ca > 0.0f && cc > 0.0f)) return true;   // degenerate conic: let the pixel test decide
    const float ux0 = bx0 - a0.x, ux1 = ux0 + bw;
    const float uy0 = by0 - a0.y, uy1 = uy0 + bh;
    const float X = fminf(fmaxf(0.0f, ux0), ux1), Y = fminf(fmaxf(0.0f, uy0), uy1);   // rectangle point nearest the centre, per axis
    const float ys = fminf(fmaxf(-cb * X * __frcp_rn(cc), uy0), uy1);
    const float xs = fminf(fmaxf(-cb * Y * __frcp_rn(ca), ux0), ux1);
    const float q1 = ca * X * X + 2.0f * cb * X * ys + cc * ys * ys;
    const float q2 = ca * xs * xs + 2.0f * cb * xs * Y + cc * Y * Y;
    return fminf(q1, q2) <= 2.0f * tau * 1.0005f + 0.01f;       // margins keep the test conservative
}
__device__ __forceinline__ bool gft_splat_reaches_quadrant(const float4& a0, const float4& a1, float qx0, float qy0)
{
    return gft_splat_reaches_box(a0, a1, qx0, qy0, 7.0f, 7.0f);
}

// Flag word of a quadrant that walked its whole head with unsaturated pixels (ImgView::unit_flag): non-zero, and it
// carries the bounding box of those pixels inside the quadrant (x0 | y0 << 3 | x1 << 6 | y1 << 9, pixels 0..7): the rest
// of the tile's list is culled against that box -- saturated pixels blend nothing more, and a silhouette quadrant's
// unsaturated pixels are a band along one edge, not the quadrant.
#define GFT_FLAG_SET 0x80000000u
__device__ __forceinline__ uint32_t gft_flag_word(unsigned long long unsaturated)      // (non-zero mask, lane = 8 y + x)
{
    uint32_t c = (uint32_t)(unsaturated | (unsaturated >> 32));
    c |= c >> 16;
    c |= c >> 8;
    c &= 0xffu;
    const uint32_t x0 = (uint32_t)__builtin_ctz(c), x1 = 31u - (uint32_t)__builtin_clz(c);
    const uint32_t y0 = (uint32_t)__builtin_ctzll(unsaturated) >> 3, y1 = (63u - (uint32_t)__builtin_clzll(unsaturated)) >> 3;
    return GFT_FLAG_SET | x0 | (y0 << 3) | (x1 << 6) | (y1 << 9);
}
// can the splat reach an unsaturated pixel of the flagged quadrant whose first pixel centre is (qx0, qy0)?
__device__ __forceinline__ bool gft_splat_reaches_flagged(uint32_t flag, const float4& a0, const float4& a1, float qx0, float qy0)
{
    if (!flag) return false;
    const uint32_t x0 = flag & 7u, y0 = (flag >> 3) & 7u, x1 = (flag >> 6) & 7u, y1 = (flag >> 9) & 7u;
    return gft_splat_reaches_box(a0, a1, qx0 + (float)x0, qy0 + (float)y0, (float)(x1 - x0), (float)(y1 - y0));
}

// Heavy-first launch order for the backward: the work of a quadrant is proportional to its
// deepest contributor (known from the forward) and varies by an order of magnitude, so tiles
// are bucket-sorted by that weight (64 buckets, descending) and dealt to the XCDs round-robin;
// the hardware dispatcher then fills free wave slots with the longest remaining units first.
// One workgroup; the order inside a bucket is arbitrary and only affects scheduling.
// (called by all threads of one workgroup).  `flag` (may be NULL): quadrants that will walk further than quad_max
// says (their lists are completed after this order is taken): their tiles go first.
__device__ inline void gft_tile_order_block(int T, const uint32_t* __restrict__ quad_max, uint32_t* __restrict__ order,
                                            const uint32_t* __restrict__ flag = nullptr)
{
    // One pass over global memory: a thread keeps the weights of its tiles (t = tid + k blockDim) in registers; the
    // histogram over the 64 weight buckets is kept per wave (the heavy tiles of a dense frame crowd a few buckets: one
    // shared counter per bucket serialised the block's LDS atomics), the bucket starts are one wave's prefix sum.
    // (The order inside a bucket is whatever the atomics give: any order of equally heavy tiles serves.)
    constexpr int PER = 16;                         // tiles per thread kept in registers; more (T > 16 blockDim) are re-read
    __shared__ uint32_t s_max;
    __shared__ uint32_t cnt[16][64];                // [wave][bucket]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_max = 0;
    for (int i = tid; i < 16 * 64; i += (int)blockDim.x) (&cnt[0][0])[i] = 0;
    // weight of a tile: its deepest quadrant walk; a flagged quadrant's walk is not over: the tile counts as the heaviest
    auto weight = [&](int t) -> uint32_t {
        const uint4 q = reinterpret_cast<const uint4*>(quad_max)[t];
        uint32_t w = max(max(q.x, q.y), max(q.z, q.w));
        if (flag) {
            const uint4 f = reinterpret_cast<const uint4*>(flag)[t];
            if ((f.x | f.y | f.z | f.w) != 0u) w = 0xffffffffu;
        }
        return w;
    };
    // (the PER loads of a thread unconditionally and together -- tile index clamped --, then the selects: under `t < T`
    // every load was a memory round trip of its own)
    uint32_t w[PER];
    uint32_t m = 0;
    {
        uint4 q[PER];
#pragma unroll
        for (int k = 0; k < PER; k++) q[k] = reinterpret_cast<const uint4*>(quad_max)[min(tid + k * (int)blockDim.x, T - 1)];
#pragma unroll
        for (int k = 0; k < PER; k++) w[k] = max(max(q[k].x, q[k].y), max(q[k].z, q[k].w));
        if (flag) {
#pragma unroll
            for (int k = 0; k < PER; k++) q[k] = reinterpret_cast<const uint4*>(flag)[min(tid + k * (int)blockDim.x, T - 1)];
#pragma unroll
            for (int k = 0; k < PER; k++)
                if ((q[k].x | q[k].y | q[k].z | q[k].w) != 0u) w[k] = 0xffffffffu;
        }
    }
#pragma unroll
    for (int k = 0; k < PER; k++) {
        if (tid + k * (int)blockDim.x >= T) w[k] = 0u;
        if (w[k] != 0xffffffffu) m = max(m, w[k]);
    }
    for (int t = tid + PER * (int)blockDim.x; t < T; t += (int)blockDim.x) {
        const uint32_t x = weight(t);
        if (x != 0xffffffffu) m = max(m, x);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
    __syncthreads();
    if (lane == 0 && m) atomicMax(&s_max, m);
    __syncthreads();
    // (any monotone map onto 64 buckets serves: a float multiply instead of an integer division per tile)
    const float scale = 64.0f / (float)(s_max + 1u);
    auto bucket = [&](uint32_t x) -> uint32_t { return x == 0xffffffffu ? 0u : 63u - min(63u, (uint32_t)((float)x * scale)); };
#pragma unroll
    for (int k = 0; k < PER; k++)
        if (tid + k * (int)blockDim.x < T) atomicAdd(&cnt[wave & 15][bucket(w[k])], 1u);
    for (int t = tid + PER * (int)blockDim.x; t < T; t += (int)blockDim.x) atomicAdd(&cnt[wave & 15][bucket(weight(t))], 1u);
    __syncthreads();
    if (wave == 0) {
        // bucket b of wave v starts at sum(buckets < b, all waves) + sum(bucket b, waves < v): cnt becomes those starts
        uint32_t tot = 0;
        for (int v = 0; v < 16; v++) tot += cnt[v][lane];
        uint32_t x = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        uint32_t run = x - tot;
        for (int v = 0; v < 16; v++) {
            const uint32_t c = cnt[v][lane];
            cnt[v][lane] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int t = tid + k * (int)blockDim.x;
        if (t < T) order[atomicAdd(&cnt[wave & 15][bucket(w[k])], 1u)] = (uint32_t)t;
    }
    for (int t = tid + PER * (int)blockDim.x; t < T; t += (int)blockDim.x)
        order[atomicAdd(&cnt[wave & 15][bucket(weight(t))], 1u)] = (uint32_t)t;
}
