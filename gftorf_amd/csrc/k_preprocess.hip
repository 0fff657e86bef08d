// k_preprocess.hip -- per-Gaussian stages (gfx950).
//
//  k_preprocess_fwd : cull, project, cov3D/cov2D, conic, radius, tile rect,
//                     SH -> RGB, SH_p -> (phase, amplitude), ToF phasor[7]
//                     (reference K1, RAST/cuda_rasterizer/forward.cu:251-419)
//  k_preprocess_bwd : conic -> cov2D -> cov3D/mean chain, projection chain, SH
//                     and SH_p chains, ToF-phasor chain, distance chain,
//                     cov3D -> scale/rotation (reference K8 + K9 fused,
//                     backward.cu:265-395 and :467-606).
//
// Memory-bound: one lane per Gaussian, records packed as 16-byte vectors.
// Floating-point contraction is disabled in this file so that the integer
// decisions derived here (radius, tile rectangle, sort-key depth bits) are the
// same on the device as in the fp32 CPU oracle: "bit-exact tile/key indexing".
#include "gft_internal.h"
#include "gft_appearance.h"
#include <mutex>

#pragma clang fp contract(off)

namespace {

// SH backward: writes dL_dsh[k*NC+c] for the active coefficients, zero for the
// inactive ones up to M, and returns the gradient w.r.t. the unit direction.
// `sh` and `dsh` may be the same array (every coefficient is read before its slot is
// overwritten), which lets the staged path work in place on one register row.
template <int NC>
__device__ __forceinline__ void sh_backward(int deg, int M, float x, float y, float z, const float* sh,
                                            const float* dres, float* dsh, float* ddir)
{
    float ddx[NC], ddy[NC], ddz[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        ddx[c] = 0.f; ddy[c] = 0.f; ddz[c] = 0.f;
        dsh[c] = SH_C0 * dres[c];
    }
    int written = 1;
    if (deg > 0) {
        const float d1 = -SH_C1 * y, d2 = SH_C1 * z, d3 = -SH_C1 * x;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const float s1 = sh[1 * NC + c], s2 = sh[2 * NC + c], s3 = sh[3 * NC + c];
            ddx[c] = -SH_C1 * s3;
            ddy[c] = -SH_C1 * s1;
            ddz[c] = SH_C1 * s2;
            dsh[1 * NC + c] = d1 * dres[c];
            dsh[2 * NC + c] = d2 * dres[c];
            dsh[3 * NC + c] = d3 * dres[c];
        }
        written = 4;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            const float d4 = SH_C2[0] * xy, d5 = SH_C2[1] * yz, d6 = SH_C2[2] * (2.f * zz - xx - yy);
            const float d7 = SH_C2[3] * xz, d8 = SH_C2[4] * (xx - yy);
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const float s4 = sh[4 * NC + c], s5 = sh[5 * NC + c], s6 = sh[6 * NC + c], s7 = sh[7 * NC + c],
                            s8 = sh[8 * NC + c];
                ddx[c] += SH_C2[0] * y * s4 + SH_C2[2] * 2.f * -x * s6 + SH_C2[3] * z * s7 + SH_C2[4] * 2.f * x * s8;
                ddy[c] += SH_C2[0] * x * s4 + SH_C2[1] * z * s5 + SH_C2[2] * 2.f * -y * s6 + SH_C2[4] * 2.f * -y * s8;
                ddz[c] += SH_C2[1] * y * s5 + SH_C2[2] * 2.f * 2.f * z * s6 + SH_C2[3] * x * s7;
                dsh[4 * NC + c] = d4 * dres[c];
                dsh[5 * NC + c] = d5 * dres[c];
                dsh[6 * NC + c] = d6 * dres[c];
                dsh[7 * NC + c] = d7 * dres[c];
                dsh[8 * NC + c] = d8 * dres[c];
            }
            written = 9;
            if (deg > 2) {
                const float d9 = SH_C3[0] * y * (3.f * xx - yy);
                const float d10 = SH_C3[1] * xy * z;
                const float d11 = SH_C3[2] * y * (4.f * zz - xx - yy);
                const float d12 = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                const float d13 = SH_C3[4] * x * (4.f * zz - xx - yy);
                const float d14 = SH_C3[5] * z * (xx - yy);
                const float d15 = SH_C3[6] * x * (xx - 3.f * yy);
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const float s9 = sh[9 * NC + c], s10 = sh[10 * NC + c], s11 = sh[11 * NC + c],
                                s12 = sh[12 * NC + c], s13 = sh[13 * NC + c], s14 = sh[14 * NC + c],
                                s15 = sh[15 * NC + c];
                    ddx[c] += (SH_C3[0] * s9 * 3.f * 2.f * xy + SH_C3[1] * s10 * yz +
                               SH_C3[2] * s11 * -2.f * xy + SH_C3[3] * s12 * -3.f * 2.f * xz +
                               SH_C3[4] * s13 * (-3.f * xx + 4.f * zz - yy) + SH_C3[5] * s14 * 2.f * xz +
                               SH_C3[6] * s15 * 3.f * (xx - yy));
                    ddy[c] += (SH_C3[0] * s9 * 3.f * (xx - yy) + SH_C3[1] * s10 * xz +
                               SH_C3[2] * s11 * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * s12 * -3.f * 2.f * yz +
                               SH_C3[4] * s13 * -2.f * xy + SH_C3[5] * s14 * -2.f * yz +
                               SH_C3[6] * s15 * -3.f * 2.f * xy);
                    ddz[c] += (SH_C3[1] * s10 * xy + SH_C3[2] * s11 * 4.f * 2.f * yz +
                               SH_C3[3] * s12 * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * s13 * 4.f * 2.f * xz +
                               SH_C3[5] * s14 * (xx - yy));
                    dsh[9 * NC + c] = d9 * dres[c];
                    dsh[10 * NC + c] = d10 * dres[c];
                    dsh[11 * NC + c] = d11 * dres[c];
                    dsh[12 * NC + c] = d12 * dres[c];
                    dsh[13 * NC + c] = d13 * dres[c];
                    dsh[14 * NC + c] = d14 * dres[c];
                    dsh[15 * NC + c] = d15 * dres[c];
                }
                written = 16;
            }
        }
    }
    // inactive coefficients: constant trip count keeps register-resident rows out of scratch
#pragma unroll
    for (int k = NC; k < 16 * NC; k++)
        if (k >= written * NC && k < M * NC) dsh[k] = 0.f;
    float sx = ddx[0] * dres[0], sy = ddy[0] * dres[0], sz = ddz[0] * dres[0];
#pragma unroll
    for (int c = 1; c < NC; c++) {
        sx = sx + ddx[c] * dres[c];
        sy = sy + ddy[c] * dres[c];
        sz = sz + ddz[c] * dres[c];
    }
    ddir[0] = sx; ddir[1] = sy; ddir[2] = sz;
}

// dL/dsh[k][c] = basis_k(dir) * dres[c] for the active coefficients, zero up to M (no SH data needed)
template <int NC>
__device__ __forceinline__ void sh_backward_basis(int deg, int M, float x, float y, float z, const float* dres,
                                                  float* dsh, bool accumulate = false)
{
    float d[16];
    d[0] = SH_C0;
#pragma unroll
    for (int k = 1; k < 16; k++) d[k] = 0.f;
    if (deg > 0) {
        d[1] = -SH_C1 * y; d[2] = SH_C1 * z; d[3] = -SH_C1 * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            d[4] = SH_C2[0] * xy; d[5] = SH_C2[1] * yz; d[6] = SH_C2[2] * (2.f * zz - xx - yy);
            d[7] = SH_C2[3] * xz; d[8] = SH_C2[4] * (xx - yy);
            if (deg > 2) {
                d[9] = SH_C3[0] * y * (3.f * xx - yy);
                d[10] = SH_C3[1] * xy * z;
                d[11] = SH_C3[2] * y * (4.f * zz - xx - yy);
                d[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                d[13] = SH_C3[4] * x * (4.f * zz - xx - yy);
                d[14] = SH_C3[5] * z * (xx - yy);
                d[15] = SH_C3[6] * x * (xx - 3.f * yy);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 16; k++)
#pragma unroll
        for (int c = 0; c < NC; c++)
            if (k < M) {
                // (accumulate: the row already holds another view's gradient, cfg.grads_accumulate)
                const float v = d[k] * dres[c];
                dsh[k * NC + c] = accumulate ? dsh[k * NC + c] + v : v;
            }
}

template <int ROW_F4>
__device__ __forceinline__ int padded_slot(int e)
{
    // element e of the wave's linear block -> slot in the padded LDS image
    const int row = ROW_F4 == 12 ? (e * 43691) >> 19 : e >> 3;     // e / 12 for e < 768, e / 8
    return row * (ROW_F4 + 1) + (e - row * ROW_F4);
}

typedef float gft_v4f __attribute__((ext_vector_type(4)));
// streaming store for data written once and not read again by this library (SH gradients):
// measured -3.7 us on k_preprocess_bwd.  (Streaming *loads* of the SH rows are a loss: the
// twelve 16-byte loads of a row rely on the cache keeping its lines, 126 -> 208 us.)
__device__ __forceinline__ void store_stream(float4* p, float4 v)
{
    const gft_v4f t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<gft_v4f*>(p));
}

__device__ __forceinline__ void dirgrad_load(const float4* base, int idx, float* v16)
{
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float4 t = base[4 * (size_t)idx + q];
        v16[4 * q] = t.x; v16[4 * q + 1] = t.y; v16[4 * q + 2] = t.z; v16[4 * q + 3] = t.w;
    }
}

template <int ROW_F4>
__device__ __forceinline__ void wave_rows_to_lds(float4* dst, const float4* __restrict__ src, size_t first_gaussian,
                                                 size_t P, int lane)
{
    const size_t lim = P * ROW_F4;
    const size_t g0 = first_gaussian * ROW_F4;
#pragma unroll
    for (int q = 0; q < ROW_F4; q++) {
        const size_t i = g0 + (size_t)(q * 64 + lane);
        if (i < lim) dst[padded_slot<ROW_F4>(q * 64 + lane)] = src[i];
    }
}

template <int ROW_F4>
__device__ __forceinline__ void wave_rows_from_lds(float4* __restrict__ dst, const float4* src, size_t first_gaussian,
                                                   size_t P, int lane)
{
    const size_t lim = P * ROW_F4;
    const size_t g0 = first_gaussian * ROW_F4;
#pragma unroll
    for (int q = 0; q < ROW_F4; q++) {
        const size_t i = g0 + (size_t)(q * 64 + lane);
        if (i < lim) store_stream(&dst[i], src[padded_slot<ROW_F4>(q * 64 + lane)]);
    }
}

// cfg.grads_accumulate: the staged rows of the Gaussians in `rows` (bit r = Gaussian first_gaussian + r) are ADDED to
// what `dst` holds (another view's gradients), as coalesced 16-byte read-modify-writes; the other rows are not touched
template <int ROW_F4>
__device__ __forceinline__ void wave_rows_add_from_lds(float4* __restrict__ dst, const float4* src, size_t first_gaussian,
                                                       size_t P, int lane, unsigned long long rows)
{
    const size_t lim = P * ROW_F4;
    const size_t g0 = first_gaussian * ROW_F4;
    if (rows == 0ull) return;
    // all loads of the wave first, then the adds and stores: one memory round trip per tensor, not one per piece
    float4 o[ROW_F4];
    bool on[ROW_F4];
#pragma unroll
    for (int q = 0; q < ROW_F4; q++) {
        const int e = q * 64 + lane;
        const int row = ROW_F4 == 12 ? (e * 43691) >> 19 : e >> 3;
        on[q] = g0 + (size_t)e < lim && ((rows >> row) & 1ull);
        o[q] = on[q] ? dst[g0 + (size_t)e] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < ROW_F4; q++) {
        const int e = q * 64 + lane;
        if (on[q]) {
            const float4 v = src[padded_slot<ROW_F4>(e)];
            dst[g0 + (size_t)e] = make_float4(o[q].x + v.x, o[q].y + v.y, o[q].z + v.z, o[q].w + v.w);
        }
    }
}

template <int ROW_F4>
__device__ __forceinline__ void lds_row_store(float4* row, const float* v)
{
#pragma unroll
    for (int q = 0; q < ROW_F4; q++) row[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

// d normalize(v) / dv applied to dv (reference auxiliary.h:110-120)
__device__ __forceinline__ void dnorm_dv(float vx, float vy, float vz, const float* dv, float* o)
{
    const float sum2 = vx * vx + vy * vy + vz * vz;
    const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    o[0] = ((+sum2 - vx * vx) * dv[0] - vy * vx * dv[1] - vz * vx * dv[2]) * invsum32;
    o[1] = (-vx * vy * dv[0] + (sum2 - vy * vy) * dv[1] - vz * vy * dv[2]) * invsum32;
    o[2] = (-vx * vz * dv[0] - vy * vz * dv[1] + (sum2 - vz * vz) * dv[2]) * invsum32;
}

__global__ __launch_bounds__(PRE_BLOCK) void k_preprocess_fwd(PreFwdArgs a)
{
    extern __shared__ float4 lds_rows[];
    const int idx = blockIdx.x * PRE_BLOCK + threadIdx.x;
    const int P = a.c.P;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float4* sh_l = lds_rows + wave * ((a.stage_sh ? 64 * SH_ROW_PAD : 0) + (a.stage_shp ? 64 * SHP_ROW_PAD : 0));
    float4* shp_l = sh_l + (a.stage_sh ? 64 * SH_ROW_PAD : 0);
    if (a.stage_sh | a.stage_shp) {
        const size_t g0 = (size_t)blockIdx.x * PRE_BLOCK + (size_t)wave * 64;
        if (a.stage_sh) wave_rows_to_lds<SH_ROW_F4>(sh_l, reinterpret_cast<const float4*>(a.io.shs), g0, (size_t)P, lane);
        if (a.stage_shp) wave_rows_to_lds<SHP_ROW_F4>(shp_l, reinterpret_cast<const float4*>(a.io.shs_p), g0, (size_t)P, lane);
        __syncthreads();
    }
    if (idx < P) {
        uint32_t tiles = 0;
        int radius = 0;
        ushort4 rect = make_ushort4(0, 0, 0, 0);
        const float px = a.io.means3D[3 * idx], py = a.io.means3D[3 * idx + 1], pz = a.io.means3D[3 * idx + 2];
        const Mat16 V = load_mat(a.io.viewmatrix);
        const float vz = V.m[2] * px + V.m[6] * py + V.m[10] * pz + V.m[14];
        if (vz < a.c.near_n || vz > a.c.far_n) {
            if (a.c.prefiltered && a.mail) __hip_atomic_fetch_or(&a.mail[GFT_CTRL_FLAGS], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            const float vx = V.m[0] * px + V.m[4] * py + V.m[8] * pz + V.m[12];
            const float vy = V.m[1] * px + V.m[5] * py + V.m[9] * pz + V.m[13];
            const ScreenGeom sg = screen_geometry(a, idx, px, py, pz, V);
            if (sg.ok) {
                int x0, y0, x1, y1;
                gft_get_rect(sg.pix_x, sg.pix_y, (int)sg.my_radius, a.gx, a.gy, x0, y0, x1, y1);
                const uint32_t area = (uint32_t)(x1 - x0) * (uint32_t)(y1 - y0);
                if (area != 0) {
                    a.g.depth[idx] = vz;
                    store_rec_a(a, idx, sg, vx, vy, vz);
                    // Tile-pull binning: the appearance (320 B of SH coefficients read, 161 B written) follows in k_appearance
                    // for the Gaussians that come to stand in the sorted part of a tile list -- in a dense frame most never do.
                    if (!a.defer_appearance) appearance_fwd(a, idx, lane, sh_l, shp_l, px, py, pz, vx, vy, vz);
                    radius = (int)sg.my_radius;
                    tiles = area;
                    rect = make_ushort4((unsigned short)x0, (unsigned short)y0, (unsigned short)x1, (unsigned short)y1);
                }
            }
        }
        a.io.radii[idx] = radius;
        a.io.pixels[idx] = 0.0f;
        a.g.tiles[idx] = tiles;
        a.g.rect[idx] = rect;
        a.g.need[idx] = 0;
    }
    // the binning kernels' counters (ctrl words, tile counters, tile cuts, supertile table): zeroed here instead of by a
    // separate fill launch; nothing in this kernel reads or writes them.  (Behind the kernel's loads: in front, the first
    // workgroups' loads waited for these stores.)
    for (uint32_t w = (uint32_t)idx; w < a.clear_words; w += gridDim.x * PRE_BLOCK) a.ctrl[w] = 0u;
}

// Appearance on demand (tile-pull binning, k_pull.hip): the Gaussians k_tile_pull marked -- those in the sorted head of
// some tile list -- get their SH colour, SH (phase, amplitude), phasor basis, direction gradients and clamp flags.  The
// 320 bytes of SH coefficients of every other Gaussian are never read.  The marked Gaussians are scattered (one in
// seven on the metric frame): a workgroup first compacts the ids of its 1024 Gaussians into LDS, so that the lanes of
// its waves have a Gaussian each to evaluate (a lane per Gaussian over all P runs with a sixth of its lanes: 36 us).
#define APP_THREADS 256
#define APP_CHUNK_OF(words) ((words) * 4 * APP_THREADS)        // Gaussians per workgroup: 1024 -- with one in seven wanted, a workgroup's
                                                               // compacted list fills one round of its lanes --, 4096 on big scenes (5 M
                                                               // Gaussians @ 1080p: one in thirty wanted)
template <int WORDS>      // 4-byte words of flags per thread
__global__ __launch_bounds__(APP_THREADS) void k_appearance(PreFwdArgs a, uint32_t cap)
{
    constexpr int APP_CHUNK = APP_CHUNK_OF(WORDS);
    __shared__ uint32_t s_ids[APP_CHUNK];
    __shared__ uint32_t s_n;
    if (a.ctrl[GFT_CTRL_TOTAL] > cap) return;
    const int tid = threadIdx.x, lane = tid & 63;
    // Workgroup 0, on the side: the heavy-first order in which the forward blend deals its quadrant waves, from the walk
    // lengths the previous frame of this camera left with the caller (any contents give a permutation of the tiles; no
    // lengths yet: tiles stay in image order, whose runs per XCD share their L2).  The blend kernel is the next launch.
    if (blockIdx.x == 0 && a.prev_w != nullptr && a.prev_w[4 * a.T] == 1u) {       // (uniform over the workgroup)
        gft_tile_order_block(a.T, a.prev_w, a.fwd_order);
        __syncthreads();
        if (tid == 0) a.ctrl[GFT_CTRL_FWDORDER] = 1u;
    }
    if (tid == 0) s_n = 0;
    __syncthreads();
    const int base = blockIdx.x * APP_CHUNK;
    // 4 * WORDS flags per thread as one load (P is padded by the layout's alignment: reads past P stay inside geom)
    {
        constexpr int PER = 4 * WORDS;
        const int i0 = base + tid * PER;
        uint32_t w[WORDS];
        if (WORDS == 4) {
            const uint4 q = *reinterpret_cast<const uint4*>(a.g.need + min(i0, (a.c.P - 1) & ~15));
            w[0] = q.x; w[WORDS > 1 ? 1 : 0] = q.y; w[WORDS > 2 ? 2 : 0] = q.z; w[WORDS > 3 ? 3 : 0] = q.w;
        } else {
            w[0] = *reinterpret_cast<const uint32_t*>(a.g.need + min(i0, (a.c.P - 1) & ~3));
        }
        auto wanted = [&](int k) { return i0 + k < a.c.P && ((w[k >> 2] >> (8 * (k & 3))) & 0xffu); };
        uint32_t mine = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) mine += wanted(k) ? 1u : 0u;
        // slots of this thread's ids: wave prefix + one LDS atomic per wave
        uint32_t x = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        uint32_t wb = 0;
        if (lane == 63 && x) wb = atomicAdd(&s_n, x);
        wb = (uint32_t)__shfl((int)wb, 63, 64);
        uint32_t pos = wb + x - mine;
#pragma unroll
        for (int k = 0; k < PER; k++)
            if (wanted(k)) s_ids[pos++] = (uint32_t)(i0 + k);
    }
    __syncthreads();
    const uint32_t n = s_n;
    const Mat16 V = load_mat(a.io.viewmatrix);
    for (uint32_t i = (uint32_t)tid; i < n; i += APP_THREADS) {
        const int idx = (int)s_ids[i];
        const float px = a.io.means3D[3 * idx], py = a.io.means3D[3 * idx + 1], pz = a.io.means3D[3 * idx + 2];
        // the same expressions as in k_preprocess_fwd: the same distance bit for bit
        const float vz = V.m[2] * px + V.m[6] * py + V.m[10] * pz + V.m[14];
        const float vx = V.m[0] * px + V.m[4] * py + V.m[8] * pz + V.m[12];
        const float vy = V.m[1] * px + V.m[5] * py + V.m[9] * pz + V.m[13];
        appearance_fwd(a, idx, lane, nullptr, nullptr, px, py, pz, vx, vy, vz);
    }
}

struct PreBwdArgs {
    gft_config c;
    gft_backward_io io;
    GeomView g;
    float focal_x, focal_y, dist2phase;
    int stage_sh, stage_shp;
    uint32_t* rows_report;      // rows-only backward: device pointer of the caller's pinned word (NULL: no report)
};

// Is Gaussian idx one that some pixel blended (non-zero accumulators, hence possibly non-zero gradients)?
__device__ __forceinline__ bool gaussian_blended(const PreBwdArgs& a, int idx)
{
    // A Gaussian that no pixel blended has all-zero accumulators and therefore all-zero gradients: they are written
    // without reading its records (with lazy binning most of the frame's Gaussians are like that, and those behind
    // the depth cut may not even have an appearance record).  The forward's per-Gaussian pixel count tells; without
    // it (pybind-level backward) the accumulator row does.
    if (!(a.io.radii[idx] > 0)) return false;
    if (a.io.pixels != nullptr) return a.io.pixels[idx] != 0.f;
    const float4* ap = reinterpret_cast<const float4*>(a.io.acc + (size_t)idx * GFT_ACC_STRIDE);
    const float4 a0 = ap[0], a1 = ap[1], a2 = ap[2], a3 = ap[3];
    return (a0.x != 0.f) | (a0.y != 0.f) | (a0.z != 0.f) | (a0.w != 0.f) | (a1.x != 0.f) | (a1.y != 0.f) |
           (a1.z != 0.f) | (a1.w != 0.f) | (a2.x != 0.f) | (a2.y != 0.f) | (a2.z != 0.f) | (a2.w != 0.f) |
           (a3.x != 0.f) | (a3.y != 0.f) | (a3.z != 0.f);
}

// COMMON = the call a training loop makes (SH colour and SH phasor of 16 coefficients, scales + rotations, the forward's
// direction-gradient record, whole gradient tensors written): the switches below are constants there and the other paths
// are not compiled in; every other combination runs the general kernel.
// ROWS = the kernel over the compacted rows of blended Gaussians (k_preprocess_bwd_rows: the gradient tensors are zero
// but for the rows this call writes): `row_idx` = this lane's Gaussian (-1: none), only its rows are written, straight
// from the lane, and its offset-gradient terms are handed back instead of being reduced here.
template <bool COMMON, bool ROWS = false>
__device__ __forceinline__ void preprocess_bwd_body(PreBwdArgs a, int row_idx = -1, float* row_phase = nullptr, float* row_dc = nullptr)
{
    if (COMMON) {
        a.stage_sh = ROWS ? 0 : 1; a.stage_shp = ROWS ? 0 : 1;
        a.c.want_backward = 1; a.c.grads_accumulate = 0; a.c.grads_zeroed = ROWS ? 1 : 0;
        a.c.M = 16; a.c.M_p = 16;
        a.io.cov3D_precomp = nullptr; a.io.dL_dcolors = nullptr; a.io.dL_dcov3D = nullptr;
        __builtin_assume(a.io.shs != nullptr);
        __builtin_assume(a.io.shs_p != nullptr);
        __builtin_assume(a.io.scales != nullptr);
    }
    extern __shared__ float4 lds_rows[];
    const int idx = ROWS ? row_idx : (int)(blockIdx.x * PRE_BLOCK + threadIdx.x);
    const int P = a.c.P;
    const int M = a.c.M, M_p = a.c.M_p;
    float sum_phase = 0.f, sum_dc = 0.f;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float4* sh_l = lds_rows + wave * ((a.stage_sh ? 64 * SH_ROW_PAD : 0) + (a.stage_shp ? 64 * SHP_ROW_PAD : 0));
    float4* shp_l = sh_l + (a.stage_sh ? 64 * SH_ROW_PAD : 0);
    const size_t g0 = (size_t)blockIdx.x * PRE_BLOCK + (size_t)wave * 64;
    // with the forward's direction-gradient record no SH coefficient is read here; the LDS rows
    // then only serve to turn the per-lane gradient rows into coalesced 16-byte stores
    const bool have_dg = a.c.want_backward != 0;
    const bool accum = a.c.grads_accumulate != 0;      // rows of blended Gaussians are added to, the others left alone
    if ((a.stage_sh | a.stage_shp) && !have_dg) {
        if (a.stage_sh) wave_rows_to_lds<SH_ROW_F4>(sh_l, reinterpret_cast<const float4*>(a.io.shs), g0, (size_t)P, lane);
        if (a.stage_shp) wave_rows_to_lds<SHP_ROW_F4>(shp_l, reinterpret_cast<const float4*>(a.io.shs_p), g0, (size_t)P, lane);
        __syncthreads();
    }

    bool blended = false;      // this lane's Gaussian got a non-zero gradient row (cfg.grads_accumulate: the rows that are added)
    if (ROWS ? idx >= 0 : idx < P) {
        const bool visible = ROWS ? true : gaussian_blended(a, idx);
        blended = visible;
        // Row marks for a caller that keeps its gradient tensors (gft_backward_io.dirty_rows): a full write leaves exactly
        // the blended rows non-zero; an accumulating call (second view of a pair) adds its rows to the marks; the rows
        // kernel writes the marks of its 1024 Gaussians in one go.
        if (!ROWS && a.io.dirty_rows) {
            if (!a.c.grads_accumulate) a.io.dirty_rows[idx] = visible ? 1 : 0;
            else if (visible) a.io.dirty_rows[idx] = 1;
        }
        float dmean[3] = {0.f, 0.f, 0.f};
        float dmean2d[2] = {0.f, 0.f};
        float dopac = 0.f;
        float dcolor[3] = {0.f, 0.f, 0.f};
        float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float dscale[3] = {0.f, 0.f, 0.f};
        float drot[4] = {0.f, 0.f, 0.f, 0.f};
        float* dsh = (a.io.shs && !a.stage_sh) ? a.io.dL_dsh + (size_t)idx * M * 3 : nullptr;
        float* dsh_p = (a.io.shs_p && !a.stage_shp) ? a.io.dL_dsh_p + (size_t)idx * M_p * 2 : nullptr;

        if (visible) {
            // accumulators written by the render backward
            const float4* ap = reinterpret_cast<const float4*>(a.io.acc + (size_t)idx * GFT_ACC_STRIDE);
            // row = {dcolor[3], ddist | sum E dx, sum E dy, dconic.xy' | XR, XI, X2, XQ | dconic.w', dopacity, dndc, -}
            // (the order the render backward's pairwise wave reduction produces, k_render.hip)
            const float4 a0 = ap[0], a1 = ap[1], a2 = ap[2], a3 = ap[3];
            // cfg.acc_zeroed = 2: the caller keeps the accumulator; the rows of blended Gaussians are its only non-zero
            // ones and each is read exactly once, here: zeroed behind the read, the buffer is all zero again when this
            // kernel ends and the next forward has nothing to clear
            if (a.c.acc_zeroed == 2) {
                float4* zp = reinterpret_cast<float4*>(a.io.acc + (size_t)idx * GFT_ACC_STRIDE);
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                zp[0] = z; zp[1] = z; zp[2] = z; zp[3] = z;
            }
            // (the forward's records for the chains further down, asked for together with the accumulator row: the wave's
            // time is its dependent memory round trips, two waves per SIMD do not hide a third one)
            const uint32_t clamp_bits = a.g.clamped[idx];
            float dg[16];
            if (have_dg) dirgrad_load(a.g.dirgrad, idx, dg);
            const float4 b2 = a.g.rec_b[2 * idx + 1];     // {I, Am, phase_sh, amplitude}
            dcolor[0] = a0.x; dcolor[1] = a0.y; dcolor[2] = a0.z;
            const float ddist_in = a0.w;
            // the five geometric sums arrive without their per-Gaussian factors (k_render_bwd):
            // dL/dmean2D.xy = -o (0.5 W, 0.5 H) * sum,  dL/dconic = -o/2 * sum
            // (the pybind-level backward of the reference has no opacity argument, rasterize_points.h:55-88: the value
            // the forward stored in the geometry record is the same float)
            const float4 ra0 = a.g.rec_a[2 * idx], ra1 = a.g.rec_a[2 * idx + 1];     // {x, y, conic a, b} {conic c, opacity, ..}
            const float nop = -(a.io.opacities ? a.io.opacities[idx] : ra1.y);
            // the render backward reduced sum E dx and sum E dy; the conic is per Gaussian
            dmean2d[0] = (ra0.z * a1.x + ra0.w * a1.y) * (nop * 0.5f * (float)a.c.W);
            dmean2d[1] = (ra0.w * a1.x + ra1.x * a1.y) * (nop * 0.5f * (float)a.c.H);
            const float dconx = a1.z * (0.5f * nop), dcony = a1.w * (0.5f * nop), dconw = a3.x * (0.5f * nop);
            dopac = a3.y;
            const float dndc_in = a3.z;
            // phasor-plane gradients arrive already folded onto the (R, I, Am) basis:
            // XR = sum w_p (g0+g3-g4), XI = sum w_p (g1+g5-g6), X2 = sum w_p g2, XQ = sum w_p (g3+g4+g5+g6)
            const float XR = a2.x, XI = a2.y, X2 = a2.z, XQ = a2.w;

            const float px = a.io.means3D[3 * idx], py = a.io.means3D[3 * idx + 1], pz = a.io.means3D[3 * idx + 2];
            const Mat16 V = load_mat(a.io.viewmatrix);
            const Mat16 PV = load_mat(a.io.projmatrix);

            // ---- conic -> cov2D -> cov3D, mean (reference K8) ----
            float cov[6];
            float sxs = 0.f, sys = 0.f, szs = 0.f;
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.io.cov3D_precomp != nullptr) {
#pragma unroll
                for (int i = 0; i < 6; i++) cov[i] = a.io.cov3D_precomp[6 * idx + i];
            } else {
                const float mod = a.c.scale_modifier;
                q = reinterpret_cast<const float4*>(a.io.rotations)[idx];
                sxs = mod * a.io.scales[3 * idx];
                sys = mod * a.io.scales[3 * idx + 1];
                szs = mod * a.io.scales[3 * idx + 2];
                cov3d_from_scale_rot(sxs, sys, szs, q, cov);
            }
            {
                const float h_x = a.focal_x, h_y = a.focal_y;
                const Ewa e = ewa_project(px, py, pz, V, h_x, h_y, a.c.tanfovx, a.c.tanfovy, cov);
                const float limx = 1.3f * a.c.tanfovx, limy = 1.3f * a.c.tanfovy;
                const float x_grad_mul = (e.txtz < -limx || e.txtz > limx) ? 0.f : 1.f;
                const float y_grad_mul = (e.tytz < -limy || e.tytz > limy) ? 0.f : 1.f;
                const float ca = e.a + 0.3f, cb = e.b, cc = e.c + 0.3f;
                const float denom = ca * cc - cb * cb;
                float dL_da = 0, dL_db = 0, dL_dc = 0;
                const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
                if (denom2inv != 0) {
                    dL_da = denom2inv * (-cc * cc * dconx + 2 * cb * cc * dcony + (denom - ca * cc) * dconw);
                    dL_dc = denom2inv * (-ca * ca * dconw + 2 * ca * cb * dcony + (denom - ca * cc) * dconx);
                    dL_db = denom2inv * 2 * (cb * cc * dconx - (denom + 2 * cb * cb) * dcony + ca * cb * dconw);
                    dcov[0] = (e.T00 * e.T00 * dL_da + e.T00 * e.T10 * dL_db + e.T10 * e.T10 * dL_dc);
                    dcov[3] = (e.T01 * e.T01 * dL_da + e.T01 * e.T11 * dL_db + e.T11 * e.T11 * dL_dc);
                    dcov[5] = (e.T02 * e.T02 * dL_da + e.T02 * e.T12 * dL_db + e.T12 * e.T12 * dL_dc);
                    dcov[1] = 2 * e.T00 * e.T01 * dL_da + (e.T00 * e.T11 + e.T01 * e.T10) * dL_db + 2 * e.T10 * e.T11 * dL_dc;
                    dcov[2] = 2 * e.T00 * e.T02 * dL_da + (e.T00 * e.T12 + e.T02 * e.T10) * dL_db + 2 * e.T10 * e.T12 * dL_dc;
                    dcov[4] = 2 * e.T02 * e.T01 * dL_da + (e.T01 * e.T12 + e.T02 * e.T11) * dL_db + 2 * e.T11 * e.T12 * dL_dc;
                }
                // dL/dT (rows 0,1); Vrk[i][j] symmetric
                const float v00 = cov[0], v01 = cov[1], v02 = cov[2], v11 = cov[3], v12 = cov[4], v22 = cov[5];
                const float dT00 = 2 * (e.T00 * v00 + e.T01 * v01 + e.T02 * v02) * dL_da + (e.T10 * v00 + e.T11 * v01 + e.T12 * v02) * dL_db;
                const float dT01 = 2 * (e.T00 * v01 + e.T01 * v11 + e.T02 * v12) * dL_da + (e.T10 * v01 + e.T11 * v11 + e.T12 * v12) * dL_db;
                const float dT02 = 2 * (e.T00 * v02 + e.T01 * v12 + e.T02 * v22) * dL_da + (e.T10 * v02 + e.T11 * v12 + e.T12 * v22) * dL_db;
                const float dT10 = 2 * (e.T10 * v00 + e.T11 * v01 + e.T12 * v02) * dL_dc + (e.T00 * v00 + e.T01 * v01 + e.T02 * v02) * dL_db;
                const float dT11 = 2 * (e.T10 * v01 + e.T11 * v11 + e.T12 * v12) * dL_dc + (e.T00 * v01 + e.T01 * v11 + e.T02 * v12) * dL_db;
                const float dT12 = 2 * (e.T10 * v02 + e.T11 * v12 + e.T12 * v22) * dL_dc + (e.T00 * v02 + e.T01 * v12 + e.T02 * v22) * dL_db;
                // W[0][r] = (v0,v4,v8), W[1][r] = (v1,v5,v9), W[2][r] = (v2,v6,v10)
                const float dJ00 = V.m[0] * dT00 + V.m[4] * dT01 + V.m[8] * dT02;
                const float dJ02 = V.m[2] * dT00 + V.m[6] * dT01 + V.m[10] * dT02;
                const float dJ11 = V.m[1] * dT10 + V.m[5] * dT11 + V.m[9] * dT12;
                const float dJ12 = V.m[2] * dT10 + V.m[6] * dT11 + V.m[10] * dT12;
                const float tz = 1.f / e.t2, tz2 = tz * tz, tz3 = tz2 * tz;
                const float dtx = x_grad_mul * -h_x * tz2 * dJ02;
                const float dty = y_grad_mul * -h_y * tz2 * dJ12;
                const float dtz = -h_x * tz2 * dJ00 - h_y * tz2 * dJ11 + (2 * h_x * e.t0) * tz3 * dJ02 + (2 * h_y * e.t1) * tz3 * dJ12;
                dmean[0] = V.m[0] * dtx + V.m[1] * dty + V.m[2] * dtz;
                dmean[1] = V.m[4] * dtx + V.m[5] * dty + V.m[6] * dtz;
                dmean[2] = V.m[8] * dtx + V.m[9] * dty + V.m[10] * dtz;
            }

            // ---- screen-space mean -> 3D mean (reference K9, :498-519) ----
            const float hw = PV.m[3] * px + PV.m[7] * py + PV.m[11] * pz + PV.m[15];
            const float m_w = 1.0f / (hw + 0.0000001f);
            const float mvx = V.m[0] * px + V.m[4] * py + V.m[8] * pz + V.m[12];
            const float mvy = V.m[1] * px + V.m[5] * py + V.m[9] * pz + V.m[13];
            const float mvz = V.m[2] * px + V.m[6] * py + V.m[10] * pz + V.m[14];
            {
                const float mul1 = (PV.m[0] * px + PV.m[4] * py + PV.m[8] * pz + PV.m[12]) * m_w * m_w;
                const float mul2 = (PV.m[1] * px + PV.m[5] * py + PV.m[9] * pz + PV.m[13]) * m_w * m_w;
                const float gx2 = dmean2d[0], gy2 = dmean2d[1];
                dmean[0] += (PV.m[0] * m_w - PV.m[3] * mul1) * gx2 + (PV.m[1] * m_w - PV.m[3] * mul2) * gy2;
                dmean[1] += (PV.m[4] * m_w - PV.m[7] * mul1) * gx2 + (PV.m[5] * m_w - PV.m[7] * mul2) * gy2;
                dmean[2] += (PV.m[8] * m_w - PV.m[11] * mul1) * gx2 + (PV.m[9] * m_w - PV.m[11] * mul2) * gy2;
            }

            const float dox = px - a.io.campos[0], doy = py - a.io.campos[1], doz = pz - a.io.campos[2];
            const float dlen = sqrtf(dox * dox + doy * doy + doz * doz);
            const float dx = dox / dlen, dy = doy / dlen, dz = doz / dlen;

            // ---- colour SH (reference backward.cu:20-139) ----
            if (a.io.shs != nullptr) {
                float dres[3], ddir[3], dm[3];
#pragma unroll
                for (int c = 0; c < 3; c++) dres[c] = dcolor[c] * (((clamp_bits >> c) & 1u) ? 0.f : 1.f);
                if (have_dg) {
                    ddir[0] = dg[0] * dres[0]; ddir[0] = ddir[0] + dg[1] * dres[1]; ddir[0] = ddir[0] + dg[2] * dres[2];
                    ddir[1] = dg[3] * dres[0]; ddir[1] = ddir[1] + dg[4] * dres[1]; ddir[1] = ddir[1] + dg[5] * dres[2];
                    ddir[2] = dg[6] * dres[0]; ddir[2] = ddir[2] + dg[7] * dres[1]; ddir[2] = ddir[2] + dg[8] * dres[2];
                    if (a.stage_sh) {
                        float v[4 * SH_ROW_F4];
                        sh_backward_basis<3>(a.c.D, 16, dx, dy, dz, dres, v);
                        lds_row_store<SH_ROW_F4>(sh_l + lane * SH_ROW_PAD, v);
                    } else {
                        sh_backward_basis<3>(a.c.D, M, dx, dy, dz, dres, dsh, accum);
                    }
                } else if (a.stage_sh) {
                    float v[4 * SH_ROW_F4];
                    lds_row_load<SH_ROW_F4>(v, sh_l + lane * SH_ROW_PAD);
                    sh_backward<3>(a.c.D, M, dx, dy, dz, v, dres, v, ddir);     // in place
                    lds_row_store<SH_ROW_F4>(sh_l + lane * SH_ROW_PAD, v);
                } else {
                    sh_backward<3>(a.c.D, M, dx, dy, dz, a.io.shs + (size_t)idx * M * 3, dres, dsh, ddir);
                }
                dnorm_dv(dox, doy, doz, ddir, dm);
                dmean[0] += dm[0]; dmean[1] += dm[1]; dmean[2] += dm[2];
            }

            // ---- ToF phasor chain (reference backward.cu:527-587) ----
            // distance to the camera: the forward's expression on the same view-space position (bit-identical;
            // re-reading it from rec_a would cost a 32-byte sector per Gaussian for 4 bytes)
            const float dist = sqrtf(mvx * mvx + mvy * mvy + mvz * mvz);
            if (a.io.shs_p != nullptr) {
                float phase = dist * a.dist2phase + a.c.phase_offset;
                if (a.c.use_view_dependent_phase) phase += b2.z;
                const float amplitude = b2.w;
                const float factor = 1.0f / (dist * dist);
                const float sin_p = sinf(phase), cos_p = cosf(phase), dc = a.c.dc_offset;
                // reference backward.cu:551-577 with dL_dR + dL_dq1 - dL_dq2 = XR etc.
                const float XA = X2 + dc * XQ;
                const float S = cos_p * XI - sin_p * XR;
                const float Camp = cos_p * XR + sin_p * XI + XA;
                float dCW[2] = {0.f, 0.f};
                if (a.c.use_view_dependent_phase) dCW[0] = S * amplitude * factor;
                sum_phase = S * amplitude * factor;
                dCW[1] = Camp * factor;
                sum_dc = XQ * amplitude * factor;
                const float coeff = S * a.dist2phase * amplitude * factor / dist - Camp * 2.0f * amplitude * factor * factor;
                const float dxv = mvx * coeff, dyv = mvy * coeff, dzv = mvz * coeff;
                dmean[0] += dxv * V.m[0] + dyv * V.m[1] + dzv * V.m[2];
                dmean[1] += dxv * V.m[4] + dyv * V.m[5] + dzv * V.m[6];
                dmean[2] += dxv * V.m[8] + dyv * V.m[9] + dzv * V.m[10];

                float dres[2], ddir[3], dm[3];
                dres[0] = dCW[0];
                dres[1] = dCW[1] * ((clamp_bits & 8u) ? 0.f : 1.f);
                if (have_dg) {
                    ddir[0] = dg[9] * dres[0]; ddir[0] = ddir[0] + dg[10] * dres[1];
                    ddir[1] = dg[11] * dres[0]; ddir[1] = ddir[1] + dg[12] * dres[1];
                    ddir[2] = dg[13] * dres[0]; ddir[2] = ddir[2] + dg[14] * dres[1];
                    if (a.stage_shp) {
                        float v[4 * SHP_ROW_F4];
                        sh_backward_basis<2>(a.c.D, 16, dx, dy, dz, dres, v);
                        lds_row_store<SHP_ROW_F4>(shp_l + lane * SHP_ROW_PAD, v);
                    } else {
                        sh_backward_basis<2>(a.c.D, M_p, dx, dy, dz, dres, dsh_p, accum);
                    }
                } else if (a.stage_shp) {
                    float v[4 * SHP_ROW_F4];
                    lds_row_load<SHP_ROW_F4>(v, shp_l + lane * SHP_ROW_PAD);
                    sh_backward<2>(a.c.D, M_p, dx, dy, dz, v, dres, v, ddir);   // in place
                    lds_row_store<SHP_ROW_F4>(shp_l + lane * SHP_ROW_PAD, v);
                } else {
                    sh_backward<2>(a.c.D, M_p, dx, dy, dz, a.io.shs_p + (size_t)idx * M_p * 2, dres, dsh_p, ddir);
                }
                dnorm_dv(dox, doy, doz, ddir, dm);
                dmean[0] += dm[0]; dmean[1] += dm[1]; dmean[2] += dm[2];
            }

            // ---- distance chain (reference backward.cu:589-601) ----
            {
                const float dndc_ddist = (a.c.far_n * a.c.near_n) / ((a.c.far_n - a.c.near_n) * dist * dist);
                const float ddist = dndc_in * dndc_ddist + ddist_in;
                const float dxv = ddist * mvx / dist, dyv = ddist * mvy / dist, dzv = ddist * mvz / dist;
                dmean[0] += dxv * V.m[0] + dyv * V.m[1] + dzv * V.m[2];
                dmean[1] += dxv * V.m[4] + dyv * V.m[5] + dzv * V.m[6];
                dmean[2] += dxv * V.m[8] + dyv * V.m[9] + dzv * V.m[10];
            }

            // ---- cov3D -> scale, rotation (reference backward.cu:399-462) ----
            if (a.io.scales != nullptr) {
                const float r = q.x, x = q.y, y = q.z, z = q.w;
                // R[c][k] (GLM columns) and M[c][k] = s_k R[c][k]
                const float R00 = 1.f - 2.f * (y * y + z * z), R01 = 2.f * (x * y - r * z), R02 = 2.f * (x * z + r * y);
                const float R10 = 2.f * (x * y + r * z), R11 = 1.f - 2.f * (x * x + z * z), R12 = 2.f * (y * z - r * x);
                const float R20 = 2.f * (x * z - r * y), R21 = 2.f * (y * z + r * x), R22 = 1.f - 2.f * (x * x + y * y);
                const float s[3] = {sxs, sys, szs};
                const float Rm[3][3] = {{R00, R01, R02}, {R10, R11, R12}, {R20, R21, R22}};
                // dL_dSigma columns
                const float dS[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
                                        {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                                        {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
                // dL_dM = (2 M) * dL_dSigma ; dMt[i][j] = dL_dM[j][i]
                float dMt[3][3];
#pragma unroll
                for (int c = 0; c < 3; c++)
#pragma unroll
                    for (int rr = 0; rr < 3; rr++) {
                        const float m0 = 2.0f * (s[rr] * Rm[0][rr]), m1 = 2.0f * (s[rr] * Rm[1][rr]), m2 = 2.0f * (s[rr] * Rm[2][rr]);
                        dMt[rr][c] = m0 * dS[c][0] + m1 * dS[c][1] + m2 * dS[c][2];
                    }
                // Rt[i][j] = R[j][i]; dscale_i = dot(Rt[i], dMt[i])
#pragma unroll
                for (int i = 0; i < 3; i++)
                    dscale[i] = Rm[0][i] * dMt[i][0] + Rm[1][i] * dMt[i][1] + Rm[2][i] * dMt[i][2];
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                    for (int j = 0; j < 3; j++) dMt[i][j] *= s[i];
                drot[0] = 2 * z * (dMt[0][1] - dMt[1][0]) + 2 * y * (dMt[2][0] - dMt[0][2]) + 2 * x * (dMt[1][2] - dMt[2][1]);
                drot[1] = 2 * y * (dMt[1][0] + dMt[0][1]) + 2 * z * (dMt[2][0] + dMt[0][2]) + 2 * r * (dMt[1][2] - dMt[2][1]) - 4 * x * (dMt[2][2] + dMt[1][1]);
                drot[2] = 2 * x * (dMt[1][0] + dMt[0][1]) + 2 * r * (dMt[2][0] - dMt[0][2]) + 2 * z * (dMt[1][2] + dMt[2][1]) - 4 * y * (dMt[2][2] + dMt[0][0]);
                drot[3] = 2 * r * (dMt[0][1] - dMt[1][0]) + 2 * x * (dMt[2][0] + dMt[0][2]) + 2 * y * (dMt[1][2] + dMt[2][1]) - 4 * z * (dMt[1][1] + dMt[0][0]);
                // reference applies scale_modifier inside s but returns dL/dscale without it
                // (backward.cu:443-446 dot(Rt, dL_dMt) before the s multiply): keep as is.
            }
        } else if (!a.c.grads_zeroed && !accum) {
            // culled or never blended: every returned gradient row is zero (already so with grads_zeroed)
            if (dsh) for (int k = 0; k < M * 3; k++) dsh[k] = 0.f;
            if (dsh_p) for (int k = 0; k < M_p * 2; k++) dsh_p[k] = 0.f;
            if (a.stage_sh) {
#pragma unroll
                for (int q = 0; q < SH_ROW_F4; q++) sh_l[lane * SH_ROW_PAD + q] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (a.stage_shp) {
#pragma unroll
                for (int q = 0; q < SHP_ROW_F4; q++) shp_l[lane * SHP_ROW_PAD + q] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }

        if (accum) {
            if (visible) {
                // loads first, stores after: one round trip for the row's small tensors
                float* m3 = a.io.dL_dmeans3D + 3 * (size_t)idx;
                float* m2 = a.io.dL_dmeans2D + 3 * (size_t)idx;
                float* opp = a.io.dL_dopacity + idx;
                const float o0 = m3[0], o1 = m3[1], o2 = m3[2], o3 = m2[0], o4 = m2[1], o5 = opp[0];
                float c0 = 0.f, c1 = 0.f, c2 = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f, v6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.io.dL_dcolors) { c0 = a.io.dL_dcolors[3 * idx]; c1 = a.io.dL_dcolors[3 * idx + 1]; c2 = a.io.dL_dcolors[3 * idx + 2]; }
                if (a.io.dL_dcov3D) {
#pragma unroll
                    for (int i = 0; i < 6; i++) v6[i] = a.io.dL_dcov3D[6 * idx + i];
                }
                if (a.io.scales != nullptr) {
                    s0 = a.io.dL_dscales[3 * idx]; s1 = a.io.dL_dscales[3 * idx + 1]; s2 = a.io.dL_dscales[3 * idx + 2];
                    r4 = reinterpret_cast<const float4*>(a.io.dL_drotations)[idx];
                }
                m3[0] = o0 + dmean[0]; m3[1] = o1 + dmean[1]; m3[2] = o2 + dmean[2];
                m2[0] = o3 + dmean2d[0]; m2[1] = o4 + dmean2d[1];
                opp[0] = o5 + dopac;
                if (a.io.dL_dcolors) {
                    a.io.dL_dcolors[3 * idx] = c0 + dcolor[0];
                    a.io.dL_dcolors[3 * idx + 1] = c1 + dcolor[1];
                    a.io.dL_dcolors[3 * idx + 2] = c2 + dcolor[2];
                }
                if (a.io.dL_dcov3D) {
#pragma unroll
                    for (int i = 0; i < 6; i++) a.io.dL_dcov3D[6 * idx + i] = v6[i] + dcov[i];
                }
                if (a.io.scales != nullptr) {
                    a.io.dL_dscales[3 * idx] = s0 + dscale[0];
                    a.io.dL_dscales[3 * idx + 1] = s1 + dscale[1];
                    a.io.dL_dscales[3 * idx + 2] = s2 + dscale[2];
                    reinterpret_cast<float4*>(a.io.dL_drotations)[idx] = make_float4(r4.x + drot[0], r4.y + drot[1], r4.z + drot[2], r4.w + drot[3]);
                }
            }
        } else if (visible || !a.c.grads_zeroed) {
        a.io.dL_dmeans3D[3 * idx] = dmean[0];
        a.io.dL_dmeans3D[3 * idx + 1] = dmean[1];
        a.io.dL_dmeans3D[3 * idx + 2] = dmean[2];
        a.io.dL_dmeans2D[3 * idx] = dmean2d[0];
        a.io.dL_dmeans2D[3 * idx + 1] = dmean2d[1];
        a.io.dL_dmeans2D[3 * idx + 2] = 0.f;
        a.io.dL_dopacity[idx] = dopac;
        if (a.io.dL_dcolors) {
            a.io.dL_dcolors[3 * idx] = dcolor[0];
            a.io.dL_dcolors[3 * idx + 1] = dcolor[1];
            a.io.dL_dcolors[3 * idx + 2] = dcolor[2];
        }
        if (a.io.dL_dcov3D) {
#pragma unroll
            for (int i = 0; i < 6; i++) a.io.dL_dcov3D[6 * idx + i] = dcov[i];
        }
        if (a.io.scales != nullptr) {
            a.io.dL_dscales[3 * idx] = dscale[0];
            a.io.dL_dscales[3 * idx + 1] = dscale[1];
            a.io.dL_dscales[3 * idx + 2] = dscale[2];
            reinterpret_cast<float4*>(a.io.dL_drotations)[idx] = make_float4(drot[0], drot[1], drot[2], drot[3]);
        }
        }
    }

    // SH gradient rows leave through LDS as coalesced 16-byte stores
    if (a.stage_sh | a.stage_shp) {
        __syncthreads();
        if (accum) {
            const unsigned long long rows = __builtin_amdgcn_ballot_w64(blended);
            if (a.stage_sh) wave_rows_add_from_lds<SH_ROW_F4>(reinterpret_cast<float4*>(a.io.dL_dsh), sh_l, g0, (size_t)P, lane, rows);
            if (a.stage_shp) wave_rows_add_from_lds<SHP_ROW_F4>(reinterpret_cast<float4*>(a.io.dL_dsh_p), shp_l, g0, (size_t)P, lane, rows);
        } else {
            if (a.stage_sh) wave_rows_from_lds<SH_ROW_F4>(reinterpret_cast<float4*>(a.io.dL_dsh), sh_l, g0, (size_t)P, lane);
            if (a.stage_shp) wave_rows_from_lds<SHP_ROW_F4>(reinterpret_cast<float4*>(a.io.dL_dsh_p), shp_l, g0, (size_t)P, lane);
        }
    }

    // phase/dc offset gradients (the reference issues two same-address atomics per Gaussian,
    // backward.cu:556-567): wave sums -> the tail of the acc scratch, two floats per wave -> fixed-order
    // reduction in k_offset_reduce (deterministic).
    if (ROWS) {
        *row_phase += sum_phase;
        *row_dc += sum_dc;
        return;
    }
    const float sp = gft_wave_sum_to_lane63(sum_phase);
    const float sd = gft_wave_sum_to_lane63(sum_dc);
    if (lane == 63 && a.io.shs_p != nullptr) {
        float* part = a.io.acc + (size_t)P * GFT_ACC_STRIDE + 2 * ((size_t)blockIdx.x * (PRE_BLOCK / 64) + wave);     // one per wave
        part[0] = sp;
        part[1] = sd;
    }
}

__global__ __launch_bounds__(PRE_BLOCK) void k_preprocess_bwd(PreBwdArgs a) { preprocess_bwd_body<false>(a); }
__global__ __launch_bounds__(PRE_BLOCK) void k_preprocess_bwd_common(PreBwdArgs a) { preprocess_bwd_body<true>(a); }

// The backward over the rows of blended Gaussians only (cfg.grads_zeroed = 2: the caller's gradient tensors are zero but
// for what this call writes).  A dense frame blends a fraction of its Gaussians (metric frame: 66 k of 1 M) and they are
// scattered: a lane per Gaussian runs the chains with 4 of 64 lanes.  A workgroup compacts the blended ids of its 1024
// Gaussians in LDS (in a fixed order: the partial sums of the two offset gradients then group the same way in every run),
// then every lane has a Gaussian; its rows go straight from the lane to the tensors and are marked dirty.
#define ROWS_THREADS 256
#define ROWS_SUBS_MAX 4
#define ROWS_CHUNK_OF(subs) ((subs) * 4 * ROWS_THREADS)      // Gaussians per workgroup: 2048 or, on big scenes, 4096
// zero rows of Gaussian `id` in every gradient tensor (one wave)
__device__ __forceinline__ void zero_gradient_rows(const gft_backward_io& io, int M, int M_p, size_t id, int lane)
{
    if (lane < 3) {
        io.dL_dmeans3D[3 * id + lane] = 0.f;
        io.dL_dmeans2D[3 * id + lane] = 0.f;
        if (io.dL_dcolors) io.dL_dcolors[3 * id + lane] = 0.f;
        if (io.dL_dscales) io.dL_dscales[3 * id + lane] = 0.f;
    }
    if (lane < 4 && io.dL_drotations) io.dL_drotations[4 * id + lane] = 0.f;
    if (lane < 6 && io.dL_dcov3D) io.dL_dcov3D[6 * id + lane] = 0.f;
    if (lane == 0) io.dL_dopacity[id] = 0.f;
    if (io.dL_dsh) for (int k = lane; k < 3 * M; k += 64) io.dL_dsh[id * (size_t)(3 * M) + k] = 0.f;
    if (io.dL_dsh_p) for (int k = lane; k < 2 * M_p; k += 64) io.dL_dsh_p[id * (size_t)(2 * M_p) + k] = 0.f;
}

template <bool COMMON, int ROWS_SUBS>
__global__ __launch_bounds__(ROWS_THREADS) void k_preprocess_bwd_rows(PreBwdArgs a)
{
    // A workgroup takes ROWS_SUBS x 1024 Gaussians in steps of 4 per thread: with 1024 per workgroup a dense frame left it
    // ~66 blended rows -- one wave of four had work, and the ~1000 workgroups needed two rounds at two waves per SIMD
    // (measured, 1 M / 5 M Gaussians: 1024 per workgroup 28 / 79 us, 2048: 26 / 51, 4096: 32.5 / 41).
    constexpr int ROWS_CHUNK = ROWS_CHUNK_OF(ROWS_SUBS);
    __shared__ uint32_t s_ids[ROWS_CHUNK];
    __shared__ uint32_t s_stale[ROWS_CHUNK];
    __shared__ uint32_t s_wt[ROWS_THREADS / 64];
    __shared__ uint32_t s_nstale;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = a.c.P;
    if (tid == 0) s_nstale = 0;
    // cfg.grads_zeroed = 3: the tensors still hold the rows the previous backward wrote (marked in dirty_rows): those of
    // this workgroup's Gaussians that this backward does not rewrite are zeroed here, the marks become this backward's
    const bool rezero = a.c.grads_zeroed == 3;
    // (all loads of the thread's Gaussians unconditionally and together -- index clamped --, then the tests: `a && load &&
    // load` per item was a memory round trip each, in a row, at the head of every workgroup)
    int rad[ROWS_SUBS][4];
    float pix[ROWS_SUBS][4];
    uint32_t old[ROWS_SUBS];           // the previous backward's marks of the four Gaussians (used when cfg.grads_zeroed = 3)
#pragma unroll
    for (int g = 0; g < ROWS_SUBS; g++) {
        const int i0 = blockIdx.x * ROWS_CHUNK + g * (4 * ROWS_THREADS) + tid * 4;
        old[g] = *reinterpret_cast<const uint32_t*>(a.io.dirty_rows + min(i0, (P - 1) & ~3));
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = min(i0 + k, P - 1);
            rad[g][k] = a.io.radii[i];
            pix[g][k] = a.io.pixels[i];
        }
    }
    uint32_t n = 0;                    // blended Gaussians of the workgroup so far: ids in s_ids, in a fixed order
#pragma unroll
    for (int g = 0; g < ROWS_SUBS; g++) {
        const int i0 = blockIdx.x * ROWS_CHUNK + g * (4 * ROWS_THREADS) + tid * 4;
        bool on[4];
        uint32_t mine = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            on[k] = i0 + k < P && rad[g][k] > 0 && pix[g][k] != 0.f;
            mine += on[k] ? 1u : 0u;
        }
        const uint32_t was = (rezero && i0 < P) ? old[g] : 0u;
        uint32_t x = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        __syncthreads();               // (the previous step has read s_wt)
        if (lane == 63) s_wt[wave] = x;
        __syncthreads();
        uint32_t pos = n + x - mine, total = 0;
        for (int w = 0; w < ROWS_THREADS / 64; w++) {
            if (w < wave) pos += s_wt[w];
            total += s_wt[w];
        }
        n += total;
        uint32_t now = 0u;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (on[k]) { s_ids[pos++] = (uint32_t)(i0 + k); now |= 1u << (8 * k); }
            else if ((was >> (8 * k)) & 0xffu) s_stale[atomicAdd(&s_nstale, 1u)] = (uint32_t)(i0 + k);
        }
        if (i0 < P && (now != was || !rezero)) *reinterpret_cast<uint32_t*>(a.io.dirty_rows + i0) = now;
    }
    __syncthreads();
    float sum_phase = 0.f, sum_dc = 0.f;
    for (uint32_t r0 = 0; r0 < n; r0 += ROWS_THREADS) {
        const uint32_t r = r0 + (uint32_t)tid;
        preprocess_bwd_body<COMMON, true>(a, r < n ? (int)s_ids[r] : -1, &sum_phase, &sum_dc);
    }
    // (the stale rows are zeroed BEHIND the rows' work: loads and stores count down one in-order counter, so in front of
    // it every load of the chains above waited until HBM had taken these stores)
    for (uint32_t r = (uint32_t)wave; r < s_nstale; r += ROWS_THREADS / 64) zero_gradient_rows(a.io, a.c.M, a.c.M_p, s_stale[r], lane);
    const float sp = gft_wave_sum_to_lane63(sum_phase);
    const float sd = gft_wave_sum_to_lane63(sum_dc);
    if (lane == 63 && a.io.shs_p != nullptr) {
        float* part = a.io.acc + (size_t)P * GFT_ACC_STRIDE + 2 * ((size_t)blockIdx.x * (ROWS_THREADS / 64) + wave);     // one per wave
        part[0] = sp;
        part[1] = sd;
    }
    if (a.rows_report && tid == 0) {
        // rows this backward writes, for the caller (gft_backward_io.rows_report): summed behind dirty_rows, the workgroup
        // that draws the last ticket stores the total to the host and leaves the two words zero for the next backward
        uint32_t* cnt = reinterpret_cast<uint32_t*>(a.io.dirty_rows + (((size_t)P + 3) & ~(size_t)3));
        if (n) atomicAdd(&cnt[0], n);
        // (a device-scope atomic is complete once vmcnt drains, and the last workgroup reads the sum with a device-scope load: no
        // cache write-back is needed -- a __threadfence here cost the kernel 17 us)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // two-level ticket (as in the supertile count pass: hundreds of returning atomics on one address are served one
        // after the other): workgroup b draws from first-level counter b % 32, the last of every group from cnt[1]
        const uint32_t G = min(32u, gridDim.x), grp = blockIdx.x % G;
        const uint32_t members = (gridDim.x - grp + G - 1u) / G;
        bool last = false;
        if (atomicAdd(&cnt[2 + grp], 1u) == members - 1u) {
            __hip_atomic_store(&cnt[2 + grp], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = atomicAdd(&cnt[1], 1u) == G - 1u;
        }
        if (last) {
            const uint32_t total = __hip_atomic_load(&cnt[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.rows_report, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&cnt[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&cnt[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Zeroes the gradient rows the previous backward into these tensors wrote (dirty[id] != 0) and clears the marks: the
// tensors are then all zero again.  66 k rows of 376 B instead of 1 M.
struct RezeroArgs {
    int P, M, M_p;
    uint8_t* dirty;
    gft_backward_io io;
};
__global__ __launch_bounds__(256) void k_grads_rezero(RezeroArgs a)
{
    __shared__ uint32_t s_ids[1024];
    __shared__ uint32_t s_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_n = 0;
    __syncthreads();
    const int i0 = blockIdx.x * 1024 + tid * 4;
    uint32_t w = 0u;
    if (i0 < a.P) w = *reinterpret_cast<const uint32_t*>(a.dirty + i0);        // (P is padded by the caller's allocation to a multiple of 4)
    if (w) {
        *reinterpret_cast<uint32_t*>(a.dirty + i0) = 0u;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (((w >> (8 * k)) & 0xffu) && i0 + k < a.P) s_ids[atomicAdd(&s_n, 1u)] = (uint32_t)(i0 + k);
    }
    __syncthreads();
    const uint32_t n = s_n;
    for (uint32_t r = (uint32_t)wave; r < n; r += 4) zero_gradient_rows(a.io, a.M, a.M_p, s_ids[r], lane);
}

__global__ __launch_bounds__(1024) void k_offset_reduce(int nblocks, const float2* __restrict__ part,
                                                        float* __restrict__ out_phase, float* __restrict__ out_dc)
{
    // fixed summation order (deterministic): thread-strided partial sums, butterfly inside the wave,
    // then the 16 wave sums in order
    __shared__ float s0[16], s1[16];
    float p = 0.f, d = 0.f;
    // two partials per 16-byte load
    const float4* part4 = reinterpret_cast<const float4*>(part);
    const int n4 = nblocks >> 1;
    for (int i = threadIdx.x; i < n4; i += 1024) {
        const float4 v = part4[i];
        p += v.x + v.z;
        d += v.y + v.w;
    }
    if ((nblocks & 1) && threadIdx.x == 0) {
        const float2 v = part[nblocks - 1];
        p += v.x;
        d += v.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        p += __shfl_xor(p, o, 64);
        d += __shfl_xor(d, o, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s0[wave] = p; s1[wave] = d; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
        for (int w = 0; w < 16; w++) { a += s0[w]; b += s1[w]; }
        *out_phase = a;
        *out_dc = b;
    }
}

__global__ __launch_bounds__(GFT_BLOCK) void k_mark_visible(int P, const float* __restrict__ means3D,
                                                            const float* __restrict__ view, float near_n,
                                                            float far_n, uint8_t* __restrict__ present)
{
    const int idx = blockIdx.x * GFT_BLOCK + threadIdx.x;
    if (idx >= P) return;
    const float px = means3D[3 * idx], py = means3D[3 * idx + 1], pz = means3D[3 * idx + 2];
    const float vz = view[2] * px + view[6] * py + view[10] * pz + view[14];
    present[idx] = (vz < near_n || vz > far_n) ? 0 : 1;
}

}  // namespace

hipError_t gft_launch_appearance(hipStream_t s, const gft_config& c, const gft_forward_io& io, const GeomView& g,
                                 const ImgView& im, uint32_t cap)
{
    PreFwdArgs a = gft_pre_fwd_args(c, io, g, im, nullptr, true);
    // (frames whose forward blend is segment-parallel keep no schedule)
    a.prev_w = gft_fwd_ordered(a.T) ? io.tile_weights : nullptr;
    a.fwd_order = im.tile_cursor;          // (tile-pull binning has no other use for these T words)
    if (c.P >= 3000000) {
        hipLaunchKernelGGL(k_appearance<4>, dim3((c.P + APP_CHUNK_OF(4) - 1) / APP_CHUNK_OF(4)), dim3(APP_THREADS), 0, s, a, cap);
    } else {
        hipLaunchKernelGGL(k_appearance<1>, dim3((c.P + APP_CHUNK_OF(1) - 1) / APP_CHUNK_OF(1)), dim3(APP_THREADS), 0, s, a, cap);
    }
    return hipGetLastError();
}

hipError_t gft_launch_preprocess_fwd(hipStream_t s, const gft_config& c, const gft_forward_io& io, const GeomView& g,
                                     const ImgView& im, uint32_t* mail, bool defer_appearance)
{
    PreFwdArgs a = gft_pre_fwd_args(c, io, g, im, mail, defer_appearance);
    a.stage_sh = (io.shs != nullptr && c.M == 16) ? 1 : 0;
    a.stage_shp = (io.shs_p != nullptr && c.M_p == 16) ? 1 : 0;
    // Measured on MI355X (1 M Gaussians): staging the forward's SH rows through LDS costs more
    // in occupancy than the strided reads cost in TA cycles (0.155 vs 0.128 ms); the backward,
    // which also writes 320 B of SH gradients per Gaussian, gains from it (0.32 -> 0.245 ms).
    // (re-measured with the final kernel: staging shs_p only 113 vs 111 us, shs only 128, both 178)
    a.stage_sh = a.stage_shp = 0;
    const size_t lds = (size_t)(PRE_BLOCK / 64) * 64 * 16 * ((a.stage_sh ? SH_ROW_PAD : 0) + (a.stage_shp ? SHP_ROW_PAD : 0));
    {
        static std::atomic<uint64_t> done_f{0}, done_b{0};
        hipError_t e = gft_lds_opt_in(reinterpret_cast<const void*>(&k_preprocess_fwd), 4 * 64 * 16 * (SH_ROW_PAD + SHP_ROW_PAD), done_f);
        if (e == hipSuccess)
            e = gft_lds_opt_in(reinterpret_cast<const void*>(&k_preprocess_bwd), 4 * 64 * 16 * (SH_ROW_PAD + SHP_ROW_PAD), done_b);
        if (e != hipSuccess) return e;
    }
    const int blocks = (c.P + PRE_BLOCK - 1) / PRE_BLOCK;
    hipLaunchKernelGGL(k_preprocess_fwd, dim3(blocks), dim3(PRE_BLOCK), lds, s, a);
    return hipGetLastError();
}

hipError_t gft_launch_preprocess_bwd(hipStream_t s, const gft_config& c, const gft_backward_io& io, const GeomView& g)
{
    PreBwdArgs a;
    a.c = c;
    a.io = io;
    a.g = g;
    a.focal_y = c.H / (2.0f * c.tanfovy);
    a.focal_x = c.W / (2.0f * c.tanfovx);
    a.dist2phase = 4.0f * 3.14159265358979323846f / c.depth_range;
    // (grads_zeroed: only the rows of blended Gaussians are written, straight from the lanes; else whole 64-row blocks
    // leave through LDS as coalesced stores)
    // (grads_zeroed: only the rows of blended Gaussians are written, straight from the lanes; else whole 64-row blocks
    // leave through LDS as coalesced stores -- with grads_accumulate the write-out adds the blended rows to what the
    // tensors hold)
    a.stage_sh = (io.shs != nullptr && c.M == 16 && !c.grads_zeroed) ? 1 : 0;
    a.stage_shp = (io.shs_p != nullptr && c.M_p == 16 && !c.grads_zeroed) ? 1 : 0;
    a.rows_report = nullptr;
    if (io.rows_report) {
        // (pinned host memory: the device pointer is asked for once per address; a pointer that is not mapped gets no report)
        static std::mutex mu;
        static const void* last_host = nullptr;
        static uint32_t* last_dev = nullptr;
        std::lock_guard<std::mutex> lk(mu);
        if (last_host != (const void*)io.rows_report) {
            void* d = nullptr;
            if (hipHostGetDevicePointer(&d, (void*)io.rows_report, 0) == hipSuccess) { last_host = io.rows_report; last_dev = (uint32_t*)d; }
            else { (void)hipGetLastError(); last_host = io.rows_report; last_dev = nullptr; }
        }
        a.rows_report = last_dev;
    }
    const size_t lds = (size_t)(PRE_BLOCK / 64) * 64 * 16 * ((a.stage_sh ? SH_ROW_PAD : 0) + (a.stage_shp ? SHP_ROW_PAD : 0));
    const int blocks = (c.P + PRE_BLOCK - 1) / PRE_BLOCK;          // = waves = partial sums of the offset gradients
    const bool common_shape = c.want_backward && io.shs && c.M == 16 && io.shs_p && c.M_p == 16 &&
                              !io.cov3D_precomp && io.scales && io.rotations && !io.dL_dcolors && !io.dL_dcov3D;
    const bool common = common_shape && !c.grads_accumulate && !c.grads_zeroed;
    int partials = blocks;
    if ((c.grads_zeroed == 2 || c.grads_zeroed == 3) && io.pixels && io.dirty_rows && c.want_backward) {
        // rows of blended Gaussians only, compacted onto full waves
        const int subs = c.P >= 3000000 ? 4 : 2;
        const int wgs = (c.P + ROWS_CHUNK_OF(subs) - 1) / ROWS_CHUNK_OF(subs);
        partials = wgs * (ROWS_THREADS / 64);
        const bool fast = common_shape && !c.grads_accumulate;
        if (fast && subs == 4) hipLaunchKernelGGL((k_preprocess_bwd_rows<true, 4>), dim3(wgs), dim3(ROWS_THREADS), 0, s, a);
        else if (fast) hipLaunchKernelGGL((k_preprocess_bwd_rows<true, 2>), dim3(wgs), dim3(ROWS_THREADS), 0, s, a);
        else if (subs == 4) hipLaunchKernelGGL((k_preprocess_bwd_rows<false, 4>), dim3(wgs), dim3(ROWS_THREADS), 0, s, a);
        else hipLaunchKernelGGL((k_preprocess_bwd_rows<false, 2>), dim3(wgs), dim3(ROWS_THREADS), 0, s, a);
    } else if (common) hipLaunchKernelGGL(k_preprocess_bwd_common, dim3(blocks), dim3(PRE_BLOCK), lds, s, a);
    else hipLaunchKernelGGL(k_preprocess_bwd, dim3(blocks), dim3(PRE_BLOCK), lds, s, a);
    // (the two scalar gradients are only reduced when the caller wants them: optimize_phase_offset / optimize_dc_offset)
    if (io.shs_p != nullptr && io.dL_dphase_offset != nullptr && io.dL_ddc_offset != nullptr)
        hipLaunchKernelGGL(k_offset_reduce, dim3(1), dim3(1024), 0, s, partials,
                           reinterpret_cast<const float2*>(io.acc + (size_t)c.P * GFT_ACC_STRIDE), io.dL_dphase_offset, io.dL_ddc_offset);
    return hipGetLastError();
}

hipError_t gft_launch_mark_visible(hipStream_t s, int32_t P, const float* means3D, const float* view,
                                   float near_n, float far_n, uint8_t* present)
{
    const int blocks = (P + GFT_BLOCK - 1) / GFT_BLOCK;
    hipLaunchKernelGGL(k_mark_visible, dim3(blocks), dim3(GFT_BLOCK), 0, s, P, means3D, view, near_n, far_n, present);
    return hipGetLastError();
}

hipError_t gft_launch_grads_rezero(hipStream_t s, const gft_config& c, const gft_backward_io& io)
{
    RezeroArgs a;
    a.P = c.P; a.M = c.M; a.M_p = c.M_p;
    a.dirty = io.dirty_rows;
    a.io = io;
    hipLaunchKernelGGL(k_grads_rezero, dim3((c.P + 1023) / 1024), dim3(256), 0, s, a);
    return hipGetLastError();
}
