// gft_internal.h -- shared declarations of libgftorf_rast.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "gftorf_rast.h"

#define GFT_ALIGN 256
#define GFT_BLOCK 256            // threads per workgroup = one 16x16 tile = 4 waves
#define GFT_NUM_ACC 18           // accumulators actually used per Gaussian

// ---- scratch views -------------------------------------------------------
struct GeomView {
    float4* rec_a;      // [P][2]  {x,y,ca,cb} {cc,opacity,dist_ndc,dist}
    float4* rec_b;      // [P][3]  {r,g,b,p0} {p1,p2,p3,p4} {p5,p6,phase_sh,amp}
    float* depth;       // [P]
    uint32_t* tiles;    // [P]
    uint32_t* offsets;  // [P] inclusive
    uint8_t* clamped;   // [P]
    uint32_t* scan_tmp; // block sums + total + flags
};

struct ImgView {
    float4* pix_state;  // [N] {final_T, n_contrib bits, w_z, w_z2}
    uint2* ranges;      // [T]
    uint32_t* tile_max; // [T]
};

struct BinView {
    uint64_t* keys_unsorted;
    uint64_t* keys;
    uint32_t* vals_unsorted;
    uint32_t* point_list;
    void* sort_tmp;
    size_t sort_tmp_bytes;
};

// scan_tmp layout (uint32): [0] = R (total), [1] = flags (bit0: prefiltered
// point culled), [2..] = per-block sums / scanned block offsets
#define GFT_SCAN_TOTAL 0
#define GFT_SCAN_FLAGS 1
#define GFT_SCAN_BLOCKS 2

void gft_compute_layout(int32_t P, int32_t W, int32_t H, int64_t R, gft_layout* L);
GeomView gft_geom_view(void* base, const gft_layout& L);
ImgView gft_img_view(void* base, const gft_layout& L);
BinView gft_bin_view(void* base, const gft_layout& L);
size_t gft_sort_tmp_bytes(int64_t R);

int gft_fail(const char* fmt, ...);
#define GFT_CHECK_HIP(expr)                                                          \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess)                                                        \
            return gft_fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                            __FILE__, __LINE__);                                     \
    } while (0)

// ---- stage launchers (each enqueues on `s`, returns hipError_t) -----------
hipError_t gft_launch_preprocess_fwd(hipStream_t s, const gft_config& c, const gft_forward_io& io,
                                     const GeomView& g);
hipError_t gft_launch_scan(hipStream_t s, int32_t P, const GeomView& g);
hipError_t gft_launch_duplicate(hipStream_t s, const gft_config& c, const GeomView& g,
                                const int32_t* radii, const BinView& b);
hipError_t gft_launch_sort(hipStream_t s, int64_t R, int end_bit, const BinView& b);
hipError_t gft_launch_ranges(hipStream_t s, int64_t R, int T, const BinView& b, const ImgView& im);
hipError_t gft_launch_render_fwd(hipStream_t s, const gft_config& c, const gft_forward_io& io,
                                 const GeomView& g, const ImgView& im, const BinView& b);
hipError_t gft_launch_render_bwd(hipStream_t s, const gft_config& c, const gft_backward_io& io,
                                 const GeomView& g, const ImgView& im, const BinView& b);
hipError_t gft_launch_preprocess_bwd(hipStream_t s, const gft_config& c, const gft_backward_io& io,
                                     const GeomView& g);
hipError_t gft_launch_mark_visible(hipStream_t s, int32_t P, const float* means3D,
                                   const float* view, float near_n, float far_n, uint8_t* present);

// reference rasterizer_impl.cu:35-50 (bit count of the tile id in the sort key)
uint32_t gft_higher_msb(uint32_t n);

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
