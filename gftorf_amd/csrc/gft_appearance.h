// gft_appearance.h -- per-Gaussian forward maths shared by k_preprocess.hip (preprocess / appearance kernels) and
// k_tail.hip (appearance of the Gaussians a flagged tile pulls in late): EWA projection, SH colour, SH (phase,
// amplitude), ToF phasor, the direction-gradient record.  Reference: RAST/cuda_rasterizer/forward.cu:20-125,128-206,
// 303-407.  Contraction is off from here to the end of the including file: the integer decisions and the SH colour
// derived from these expressions are bit-identical to the fp32 CPU oracle.
#pragma once
#include "gft_internal.h"

#pragma clang fp contract(off)

struct PreFwdArgs {
    gft_config c;
    gft_forward_io io;
    GeomView g;
    uint32_t* ctrl;            // ctrl words + the counters behind them: cleared here (clear_words), used from the next kernel on
    uint32_t clear_words;
    uint32_t* mail;            // host mailbox slot: the "prefiltered point culled" flag goes straight there
    int defer_appearance;      // tile-pull binning: geometry only; the appearance follows for the Gaussians marked in g.need
    float focal_x, focal_y, dist2phase;
    int gx, gy;
    int stage_sh, stage_shp;   // SH rows staged through LDS (M == 16)
    // k_appearance, workgroup 0: the forward blend's heavy-first tile order from the caller's quadrant walk lengths of the
    // previous frame of this camera (gft_forward_io.tile_weights; NULL: none)
    const uint32_t* prev_w;
    uint32_t* fwd_order;
    int T;
};

inline PreFwdArgs gft_pre_fwd_args(const gft_config& c, const gft_forward_io& io, const GeomView& g, const ImgView& im,
                                   uint32_t* mail, bool defer_appearance)
{
    PreFwdArgs a;
    a.c = c;
    a.io = io;
    a.g = g;
    a.ctrl = im.ctrl;
    a.mail = mail;
    a.defer_appearance = defer_appearance ? 1 : 0;
    {
        // ctrl | tickets | tile_cnt[T] | tile_cut[T] | super_tab are contiguous (gft_compute_layout)
        const size_t T = (size_t)((c.W + GFT_TILE_X - 1) / GFT_TILE_X) * (size_t)((c.H + GFT_TILE_Y - 1) / GFT_TILE_Y);
        a.clear_words = (uint32_t)(GFT_CTRL_WORDS + GFT_TICKET_WORDS + 2 * T + GFT_SUPER_CELLS);      // (of super_tab only the counters' plane)
    }
    // reference rasterizer_impl.cu:249-250, forward.cu:752
    a.focal_y = c.H / (2.0f * c.tanfovy);
    a.focal_x = c.W / (2.0f * c.tanfovx);
    a.dist2phase = 4.0f * 3.14159265358979323846f / c.depth_range;
    a.gx = (c.W + GFT_TILE_X - 1) / GFT_TILE_X;
    a.gy = (c.H + GFT_TILE_Y - 1) / GFT_TILE_Y;
    a.stage_sh = a.stage_shp = 0;
    a.prev_w = nullptr; a.fwd_order = nullptr; a.T = a.gx * a.gy;
    return a;
}

namespace {

__device__ const float SH_C0 = 0.28209479177387814f;
__device__ const float SH_C1 = 0.4886025119029199f;
__device__ const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                   -1.0925484305920792f, 0.5462742152960396f};
__device__ const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                   0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                   -0.5900435899266435f};

struct Mat16 { float m[16]; };

__device__ __forceinline__ Mat16 load_mat(const float* __restrict__ p)
{
    Mat16 r;
#pragma unroll
    for (int i = 0; i < 16; i++) r.m[i] = p[i];
    return r;
}

// intermediate values of the EWA projection shared by forward and backward
struct Ewa {
    float t0, t1, t2;      // view-space mean with the frustum clamp applied to x,y
    float txtz, tytz;      // unclamped ratios
    float T00, T01, T02;   // T[0][r]
    float T10, T11, T12;   // T[1][r]
    float a, b, c;         // cov2D[0][0], [0][1], [1][1] before the +0.3 low-pass
};

// cov3D: 6 unique entries of Sigma = (S R)^T (S R) in the reference's GLM order
__device__ __forceinline__ void cov3d_from_scale_rot(float sx, float sy, float sz, float4 q, float* cov)
{
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    // M[c][k] = s_k * R[c][k]
    const float M00 = sx * (1.f - 2.f * (y * y + z * z)), M01 = sy * (2.f * (x * y - r * z)), M02 = sz * (2.f * (x * z + r * y));
    const float M10 = sx * (2.f * (x * y + r * z)), M11 = sy * (1.f - 2.f * (x * x + z * z)), M12 = sz * (2.f * (y * z - r * x));
    const float M20 = sx * (2.f * (x * z - r * y)), M21 = sy * (2.f * (y * z + r * x)), M22 = sz * (1.f - 2.f * (x * x + y * y));
    cov[0] = M00 * M00 + M01 * M01 + M02 * M02;
    cov[1] = M10 * M00 + M11 * M01 + M12 * M02;
    cov[2] = M20 * M00 + M21 * M01 + M22 * M02;
    cov[3] = M10 * M10 + M11 * M11 + M12 * M12;
    cov[4] = M20 * M10 + M21 * M11 + M22 * M12;
    cov[5] = M20 * M20 + M21 * M21 + M22 * M22;
}

__device__ __forceinline__ Ewa ewa_project(float px, float py, float pz, const Mat16& V, float fx, float fy,
                                           float tanx, float tany, const float* cov)
{
    Ewa e;
    e.t0 = V.m[0] * px + V.m[4] * py + V.m[8] * pz + V.m[12];
    e.t1 = V.m[1] * px + V.m[5] * py + V.m[9] * pz + V.m[13];
    e.t2 = V.m[2] * px + V.m[6] * py + V.m[10] * pz + V.m[14];
    const float limx = 1.3f * tanx, limy = 1.3f * tany;
    e.txtz = e.t0 / e.t2;
    e.tytz = e.t1 / e.t2;
    e.t0 = fminf(limx, fmaxf(-limx, e.txtz)) * e.t2;
    e.t1 = fminf(limy, fmaxf(-limy, e.tytz)) * e.t2;
    const float J00 = fx / e.t2, J02 = -(fx * e.t0) / (e.t2 * e.t2);
    const float J11 = fy / e.t2, J12 = -(fy * e.t1) / (e.t2 * e.t2);
    // T = W * J with W[0][r] = (v0,v4,v8), W[1][r] = (v1,v5,v9), W[2][r] = (v2,v6,v10)
    e.T00 = V.m[0] * J00 + V.m[2] * J02;
    e.T01 = V.m[4] * J00 + V.m[6] * J02;
    e.T02 = V.m[8] * J00 + V.m[10] * J02;
    e.T10 = V.m[1] * J11 + V.m[2] * J12;
    e.T11 = V.m[5] * J11 + V.m[6] * J12;
    e.T12 = V.m[9] * J11 + V.m[10] * J12;
    // X = T^T * Vrk^T ; cov = X * T  (Vrk[i][j] symmetric: c0 c1 c2 / c1 c3 c4 / c2 c4 c5)
    const float X00 = e.T00 * cov[0] + e.T01 * cov[1] + e.T02 * cov[2];
    const float X10 = e.T00 * cov[1] + e.T01 * cov[3] + e.T02 * cov[4];
    const float X20 = e.T00 * cov[2] + e.T01 * cov[4] + e.T02 * cov[5];
    const float X01 = e.T10 * cov[0] + e.T11 * cov[1] + e.T12 * cov[2];
    const float X11 = e.T10 * cov[1] + e.T11 * cov[3] + e.T12 * cov[4];
    const float X21 = e.T10 * cov[2] + e.T11 * cov[4] + e.T12 * cov[5];
    e.a = X00 * e.T00 + X10 * e.T01 + X20 * e.T02;
    e.b = X01 * e.T00 + X11 * e.T01 + X21 * e.T02;
    e.c = X01 * e.T10 + X11 * e.T11 + X21 * e.T12;
    return e;
}

// SH basis evaluation: result[c] = sum_k basis_k(dir) * sh[k*NC + c]
template <int NC>
__device__ __forceinline__ void sh_eval(int deg, float x, float y, float z, const float* __restrict__ sh, float* res)
{
#pragma unroll
    for (int c = 0; c < NC; c++) res[c] = SH_C0 * sh[c];
    if (deg > 0) {
#pragma unroll
        for (int c = 0; c < NC; c++)
            res[c] = res[c] - SH_C1 * y * sh[1 * NC + c] + SH_C1 * z * sh[2 * NC + c] - SH_C1 * x * sh[3 * NC + c];
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
#pragma unroll
            for (int c = 0; c < NC; c++)
                res[c] = res[c] + SH_C2[0] * xy * sh[4 * NC + c] + SH_C2[1] * yz * sh[5 * NC + c] +
                         SH_C2[2] * (2.0f * zz - xx - yy) * sh[6 * NC + c] + SH_C2[3] * xz * sh[7 * NC + c] +
                         SH_C2[4] * (xx - yy) * sh[8 * NC + c];
            if (deg > 2) {
#pragma unroll
                for (int c = 0; c < NC; c++)
                    res[c] = res[c] + SH_C3[0] * y * (3.0f * xx - yy) * sh[9 * NC + c] +
                             SH_C3[1] * xy * z * sh[10 * NC + c] +
                             SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[11 * NC + c] +
                             SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[12 * NC + c] +
                             SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[13 * NC + c] +
                             SH_C3[5] * z * (xx - yy) * sh[14 * NC + c] +
                             SH_C3[6] * x * (xx - 3.0f * yy) * sh[15 * NC + c];
            }
        }
    }
}

// d(SH polynomial)/d(unit direction) per channel: the same sums the reference backward forms
// from the coefficients (backward.cu:58-60,78-80,99-122).  Evaluated in the forward (which
// has the row in registers) when a backward will follow, so the backward needs no SH reads.
template <int NC>
__device__ __forceinline__ void sh_dir_grad(int deg, float x, float y, float z, const float* sh, float* ddx,
                                            float* ddy, float* ddz)
{
#pragma unroll
    for (int c = 0; c < NC; c++) { ddx[c] = 0.f; ddy[c] = 0.f; ddz[c] = 0.f; }
    if (deg > 0) {
#pragma unroll
        for (int c = 0; c < NC; c++) {
            ddx[c] = -SH_C1 * sh[3 * NC + c];
            ddy[c] = -SH_C1 * sh[1 * NC + c];
            ddz[c] = SH_C1 * sh[2 * NC + c];
        }
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const float s4 = sh[4 * NC + c], s5 = sh[5 * NC + c], s6 = sh[6 * NC + c], s7 = sh[7 * NC + c],
                            s8 = sh[8 * NC + c];
                ddx[c] += SH_C2[0] * y * s4 + SH_C2[2] * 2.f * -x * s6 + SH_C2[3] * z * s7 + SH_C2[4] * 2.f * x * s8;
                ddy[c] += SH_C2[0] * x * s4 + SH_C2[1] * z * s5 + SH_C2[2] * 2.f * -y * s6 + SH_C2[4] * 2.f * -y * s8;
                ddz[c] += SH_C2[1] * y * s5 + SH_C2[2] * 2.f * 2.f * z * s6 + SH_C2[3] * x * s7;
            }
            if (deg > 2) {
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
                }
            }
        }
    }
}

// ---- coalesced SH rows through LDS ---------------------------------------------
// A wave owns 64 consecutive Gaussians whose SH rows form one contiguous block of
// 64*M*NC floats: moved as 16-byte vectors with consecutive lanes on consecutive
// addresses (the per-lane row walk of the straightforward kernel touches 64 cache
// lines per load instruction).  Used when M == 16 (rows are whole float4s).
#define SH_ROW_F4 12     // 16 coefficients x 3 channels
#define SHP_ROW_F4 8     // 16 coefficients x 2 channels
// LDS rows are padded by one float4 (odd stride in 16-byte slots): the per-lane ds_read/write_b128
// row walks are then conflict free (12- and 8-slot strides gave 4- and 8-way conflicts, measured
// 39 M of 50 M LDS cycles in the backward).
#define SH_ROW_PAD (SH_ROW_F4 + 1)
#define SHP_ROW_PAD (SHP_ROW_F4 + 1)

// direction-gradient record: 15 fp32 values in 64 B.  (An fp16 record, 32 B, passes every parity
// test and saves 9 us per step; kept fp32 so that the whole path stays in one arithmetic type.)
__device__ __forceinline__ void dirgrad_store(float4* base, int idx, const float* c9, const float* p6)
{
    base[4 * (size_t)idx] = make_float4(c9[0], c9[1], c9[2], c9[3]);
    base[4 * (size_t)idx + 1] = make_float4(c9[4], c9[5], c9[6], c9[7]);
    base[4 * (size_t)idx + 2] = make_float4(c9[8], p6[0], p6[1], p6[2]);
    base[4 * (size_t)idx + 3] = make_float4(p6[3], p6[4], p6[5], 0.f);
}
template <int ROW_F4>
__device__ __forceinline__ void lds_row_load(float* v, const float4* row)
{
#pragma unroll
    for (int q = 0; q < ROW_F4; q++) {
        const float4 t = row[q];
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
}

// One wave per workgroup: the SH rows of a wave go through a private LDS region, barriers are
// wave-local, and occupancy is not quantised by a 4-wave LDS footprint.
#define PRE_BLOCK 64


// Appearance of one visible Gaussian (reference forward.cu:346-407): SH colour, SH (phase, amplitude), the ToF phasor
// on its (R, I, Am) basis, the direction gradients for the backward and the clamp flags -> rec_b, dirgrad, clamped.
// Called by k_preprocess_fwd for the Gaussians of the near slab (all of them without a depth cut) and by
// k_appearance_far for the others when a quadrant asks for the far slab: most far Gaussians are never blended, and
// their 320 bytes of SH coefficients are then never read.
__device__ __forceinline__ void appearance_fwd(const PreFwdArgs& a, int idx, int lane, const float4* sh_l, const float4* shp_l,
                           float px, float py, float pz, float vx, float vy, float vz)
{
    const float3 cam = make_float3(a.io.campos[0], a.io.campos[1], a.io.campos[2]);
    const float dox = px - cam.x, doy = py - cam.y, doz = pz - cam.z;
    const float dlen = sqrtf(dox * dox + doy * doy + doz * doz);
    const float dx = dox / dlen, dy = doy / dlen, dz = doz / dlen;

    float rgb[3] = {0.f, 0.f, 0.f};
    uint32_t clamp_bits = 0;
    float dgc[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // d rgb / d dir
    float dgp[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                  // d (phase, amp) / d dir
    if (a.io.colors_precomp != nullptr) {
        rgb[0] = a.io.colors_precomp[3 * idx];
        rgb[1] = a.io.colors_precomp[3 * idx + 1];
        rgb[2] = a.io.colors_precomp[3 * idx + 2];
    }
    if (a.io.shs != nullptr) {
        float res[3];
        if (a.stage_sh) {
            float v[4 * SH_ROW_F4];
            lds_row_load<SH_ROW_F4>(v, sh_l + lane * SH_ROW_PAD);
            sh_eval<3>(a.c.D, dx, dy, dz, v, res);
            if (a.c.want_backward) sh_dir_grad<3>(a.c.D, dx, dy, dz, v, dgc, dgc + 3, dgc + 6);
        } else if (a.c.M == 16) {
            // whole 192-byte row as twelve 16-byte loads (4x fewer TA requests than dwords)
            float v[4 * SH_ROW_F4];
            const float4* r4 = reinterpret_cast<const float4*>(a.io.shs) + (size_t)idx * SH_ROW_F4;
#pragma unroll
            for (int q = 0; q < SH_ROW_F4; q++) {
                const float4 t = r4[q];
                v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
            }
            sh_eval<3>(a.c.D, dx, dy, dz, v, res);
            if (a.c.want_backward) sh_dir_grad<3>(a.c.D, dx, dy, dz, v, dgc, dgc + 3, dgc + 6);
        } else {
            const float* sp3 = a.io.shs + (size_t)idx * a.c.M * 3;
            sh_eval<3>(a.c.D, dx, dy, dz, sp3, res);
            if (a.c.want_backward) sh_dir_grad<3>(a.c.D, dx, dy, dz, sp3, dgc, dgc + 3, dgc + 6);
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            res[c] += 0.5f;
            if (res[c] < 0) clamp_bits |= (1u << c);
            rgb[c] = fmaxf(res[c], 0.0f);
        }
    }

    const float dist = sqrtf(vx * vx + vy * vy + vz * vz);
    const float dist_ndc = a.c.far_n / (a.c.far_n - a.c.near_n) * (1 - a.c.near_n / dist);
    const float factor = 1.0f / (dist * dist);

    // ToF phasor; undefined in the reference when neither input is given -> zeros
    // The seven ToF planes are linear in three per-splat values (reference
    // forward.cu:399-406): R = cos(phi) A/d^2, I = sin(phi) A/d^2, Am = A/d^2;
    // planes 3..6 are (+-R + dc Am), (+-I + dc Am) and are formed by the render kernels.
    float ph[3] = {0.f, 0.f, 0.f};
    float phase_sh = 0.f, amplitude = 0.f;
    bool have_phasor = false;
    float phase = 0.f;
    if (a.io.phasors_precomp != nullptr) {
        phase = dist * a.dist2phase;
        phase_sh = a.io.phasors_precomp[2 * idx];
        amplitude = a.io.phasors_precomp[2 * idx + 1];
        if (a.c.use_view_dependent_phase) phase += phase_sh;
        have_phasor = true;
    }
    if (a.io.shs_p != nullptr) {
        float res[2];
        float sp0;
        if (a.stage_shp) {
            float v[4 * SHP_ROW_F4];
            lds_row_load<SHP_ROW_F4>(v, shp_l + lane * SHP_ROW_PAD);
            sh_eval<2>(a.c.D, dx, dy, dz, v, res);
            if (a.c.want_backward) sh_dir_grad<2>(a.c.D, dx, dy, dz, v, dgp, dgp + 2, dgp + 4);
            sp0 = v[0];
        } else if (a.c.M_p == 16) {
            float v[4 * SHP_ROW_F4];
            const float4* r4 = reinterpret_cast<const float4*>(a.io.shs_p) + (size_t)idx * SHP_ROW_F4;
#pragma unroll
            for (int q = 0; q < SHP_ROW_F4; q++) {
                const float4 t = r4[q];
                v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
            }
            sh_eval<2>(a.c.D, dx, dy, dz, v, res);
            if (a.c.want_backward) sh_dir_grad<2>(a.c.D, dx, dy, dz, v, dgp, dgp + 2, dgp + 4);
            sp0 = v[0];
        } else {
            const float* sp = a.io.shs_p + (size_t)idx * a.c.M_p * 2;
            sh_eval<2>(a.c.D, dx, dy, dz, sp, res);
            if (a.c.want_backward) sh_dir_grad<2>(a.c.D, dx, dy, dz, sp, dgp, dgp + 2, dgp + 4);
            sp0 = sp[0];
        }
        res[0] += 0.5f;
        res[1] += 0.5f;
        res[0] = res[0] - 0.5f - SH_C0 * sp0;
        if (res[1] < 0) {
            clamp_bits |= 8u;
            res[1] = 0.0f;
        }
        phase_sh = res[0];
        amplitude = res[1];
        phase = dist * a.dist2phase + a.c.phase_offset;
        if (a.c.use_view_dependent_phase) phase += phase_sh;
        have_phasor = true;
    }
    if (have_phasor) {
        const float cp = cosf(phase), sn = sinf(phase);
        ph[0] = cp * amplitude * factor;
        ph[1] = sn * amplitude * factor;
        ph[2] = amplitude * factor;
    }

    (void)dist_ndc;
    a.g.rec_b[2 * idx] = make_float4(rgb[0], rgb[1], rgb[2], ph[0]);
    a.g.rec_b[2 * idx + 1] = make_float4(ph[1], ph[2], phase_sh, amplitude);
    if (a.c.want_backward) {
        dirgrad_store(a.g.dirgrad, idx, dgc, dgp);
    }
    a.g.clamped[idx] = (uint8_t)clamp_bits;
}

// Screen-space geometry of one Gaussian that passed the depth test (reference forward.cu:303-345): pixel centre, conic,
// radius.  One function for the preprocess kernel and for the far pass, which writes the geometry record of a far
// Gaussian only when a flagged tile needs it: the same expressions, the same bits.
struct ScreenGeom {
    float pix_x, pix_y, conx, cony, conz, my_radius;
    bool ok;             // false: degenerate 2D covariance (det == 0), the Gaussian is invisible
};
__device__ __forceinline__ ScreenGeom screen_geometry(const PreFwdArgs& a, int idx, float px, float py, float pz, const Mat16& V)
{
    ScreenGeom o;
    o.ok = false;
    o.pix_x = o.pix_y = o.conx = o.cony = o.conz = o.my_radius = 0.f;
    const Mat16 PV = load_mat(a.io.projmatrix);
    const float hx = PV.m[0] * px + PV.m[4] * py + PV.m[8] * pz + PV.m[12];
    const float hy = PV.m[1] * px + PV.m[5] * py + PV.m[9] * pz + PV.m[13];
    const float hw = PV.m[3] * px + PV.m[7] * py + PV.m[11] * pz + PV.m[15];
    const float p_w = 1.0f / (hw + 0.0000001f);
    const float ndc_x = hx * p_w, ndc_y = hy * p_w;

    float cov[6];
    if (a.io.cov3D_precomp != nullptr) {
#pragma unroll
        for (int i = 0; i < 6; i++) cov[i] = a.io.cov3D_precomp[6 * idx + i];
    } else {
        const float mod = a.c.scale_modifier;
        const float4 q = reinterpret_cast<const float4*>(a.io.rotations)[idx];
        cov3d_from_scale_rot(mod * a.io.scales[3 * idx], mod * a.io.scales[3 * idx + 1],
                             mod * a.io.scales[3 * idx + 2], q, cov);
    }
    const Ewa e = ewa_project(px, py, pz, V, a.focal_x, a.focal_y, a.c.tanfovx, a.c.tanfovy, cov);
    const float ca = e.a + 0.3f, cb = e.b, cc = e.c + 0.3f;
    const float det = ca * cc - cb * cb;
    if (det != 0.0f) {
        const float det_inv = 1.f / det;
        o.conx = cc * det_inv; o.cony = -cb * det_inv; o.conz = ca * det_inv;
        const float mid = 0.5f * (ca + cc);
        const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        o.my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        // ndc2Pix is evaluated in double in the reference (auxiliary.h:44-47)
        o.pix_x = (float)(((ndc_x + 1.0) * a.c.W - 1.0) * 0.5);
        o.pix_y = (float)(((ndc_y + 1.0) * a.c.H - 1.0) * 0.5);
        o.ok = true;
    }
    return o;
}

// the 32-byte geometry record the render kernels read: {x, y, conic a, b} {conic c, opacity, NDC distance, distance}
__device__ __forceinline__ void store_rec_a(const PreFwdArgs& a, int idx, const ScreenGeom& sg, float vx, float vy, float vz)
{
    const float dist = sqrtf(vx * vx + vy * vy + vz * vz);
    const float dist_ndc = a.c.far_n / (a.c.far_n - a.c.near_n) * (1 - a.c.near_n / dist);
    a.g.rec_a[2 * idx] = make_float4(sg.pix_x, sg.pix_y, sg.conx, sg.cony);
    a.g.rec_a[2 * idx + 1] = make_float4(sg.conz, a.io.opacities[idx], dist_ndc, dist);
}

}  // namespace
