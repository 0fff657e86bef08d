/*
 * gft_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See gft_oracle.h for scope and pinning status.  RAST/ =
 * submodules/diff-gaussian-rasterization-w-tof/ of the reference.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -fPIC -shared (oracle/Makefile).
 * fp32 everywhere the reference is fp32; operation order follows the
 * reference expression by expression (GLM 0.9.9.9 operator order restated
 * from RAST/third_party/glm/glm/detail/type_mat3x3.inl:486-520).
 */
#include "gft_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- constants: RAST/cuda_rasterizer/auxiliary.h:23-42 ------------------- */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f,
                               0.31539156525252005f, -1.0925484305920792f,
                               0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f,
                               -0.4570457994644658f, 0.3731763325901154f,
                               -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};
static const float PI_F = 3.14159265358979323846f;

int gfto_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* threads of the following calls (bench.py times the oracle on all host cores and on one); returns the count in effect */
int gfto_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* ---- tiny column-major 3x3 (GLM semantics: m[c][r]) ---------------------- */
typedef struct { float m[3][3]; } m3;

static m3 m3_cols(float a, float b, float c, float d, float e, float f,
                  float g, float h, float i)
{
    m3 r;
    r.m[0][0] = a; r.m[0][1] = b; r.m[0][2] = c;
    r.m[1][0] = d; r.m[1][1] = e; r.m[1][2] = f;
    r.m[2][0] = g; r.m[2][1] = h; r.m[2][2] = i;
    return r;
}

/* type_mat3x3.inl:486-520 */
static m3 m3_mul(const m3* A, const m3* B)
{
    m3 R;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++)
            R.m[c][r] = A->m[0][r] * B->m[c][0] + A->m[1][r] * B->m[c][1] +
                        A->m[2][r] * B->m[c][2];
    return R;
}

static m3 m3_transpose(const m3* A)
{
    m3 R;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++)
            R.m[c][r] = A->m[r][c];
    return R;
}

static inline float fminf_(float a, float b) { return a < b ? a : b; }
static inline float fmaxf_(float a, float b) { return a > b ? a : b; }
static inline int imin_(int a, int b) { return a < b ? a : b; }
static inline int imax_(int a, int b) { return a > b ? a : b; }

/* ---- auxiliary.h:44-59 --------------------------------------------------- */
static float ndc2pix(float v, int S)
{
    return (float)(((v + 1.0) * S - 1.0) * 0.5);
}

static void get_rect(float px, float py, int max_radius, int gx, int gy,
                     uint32_t* rmin, uint32_t* rmax)
{
    rmin[0] = (uint32_t)imin_(gx, imax_(0, (int)((px - max_radius) / GFTO_BLOCK_X)));
    rmin[1] = (uint32_t)imin_(gy, imax_(0, (int)((py - max_radius) / GFTO_BLOCK_Y)));
    rmax[0] = (uint32_t)imin_(gx, imax_(0, (int)((px + max_radius + GFTO_BLOCK_X - 1) / GFTO_BLOCK_X)));
    rmax[1] = (uint32_t)imin_(gy, imax_(0, (int)((py + max_radius + GFTO_BLOCK_Y - 1) / GFTO_BLOCK_Y)));
}

/* auxiliary.h:61-80 */
static void transform_point_4x3(const float* p, const float* M, float* o)
{
    o[0] = M[0] * p[0] + M[4] * p[1] + M[8] * p[2] + M[12];
    o[1] = M[1] * p[0] + M[5] * p[1] + M[9] * p[2] + M[13];
    o[2] = M[2] * p[0] + M[6] * p[1] + M[10] * p[2] + M[14];
}

static void transform_point_4x4(const float* p, const float* M, float* o)
{
    o[0] = M[0] * p[0] + M[4] * p[1] + M[8] * p[2] + M[12];
    o[1] = M[1] * p[0] + M[5] * p[1] + M[9] * p[2] + M[13];
    o[2] = M[2] * p[0] + M[6] * p[1] + M[10] * p[2] + M[14];
    o[3] = M[3] * p[0] + M[7] * p[1] + M[11] * p[2] + M[15];
}

/* auxiliary.h:110-120 */
static void dnormvdv3(const float* v, const float* dv, float* o)
{
    float sum2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    o[0] = ((+sum2 - v[0] * v[0]) * dv[0] - v[1] * v[0] * dv[1] - v[2] * v[0] * dv[2]) * invsum32;
    o[1] = (-v[0] * v[1] * dv[0] + (sum2 - v[1] * v[1]) * dv[1] - v[2] * v[1] * dv[2]) * invsum32;
    o[2] = (-v[0] * v[2] * dv[0] - v[1] * v[2] * dv[1] + (sum2 - v[2] * v[2]) * dv[2]) * invsum32;
}

/* rasterizer_impl.cu:35-50 */
uint32_t gfto_get_higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb)
            msb += step;
        else
            msb -= step;
    }
    if (n >> msb)
        msb++;
    return msb;
}

/* auxiliary.h:152-179: depth-range test only, no lateral cull */
static int in_frustum(const float* p_orig, const float* view, float near_n,
                      float far_n, float* p_view)
{
    transform_point_4x3(p_orig, view, p_view);
    if (p_view[2] < near_n || p_view[2] > far_n)
        return 0;
    return 1;
}

void gfto_mark_visible(int P, const float* means3D, const float* viewmatrix,
                       const float* projmatrix, float near_n, float far_n,
                       uint8_t* present)
{
    (void)projmatrix;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        float pv[3];
        present[i] = (uint8_t)in_frustum(means3D + 3 * i, viewmatrix, near_n, far_n, pv);
    }
}

/* ---- SH direction basis shared by colour/phasor, forward.cu:20-125 -------
 * Evaluates result = sum_k basis_k * sh[k] for NC channels in the reference's
 * left-to-right order. */
static void sh_eval(int deg, const float* dir, const float* sh, int NC, float* result)
{
    float x = dir[0], y = dir[1], z = dir[2];
    for (int c = 0; c < NC; c++)
        result[c] = SH_C0 * sh[0 * NC + c];
    if (deg > 0) {
        for (int c = 0; c < NC; c++)
            result[c] = result[c] - SH_C1 * y * sh[1 * NC + c] + SH_C1 * z * sh[2 * NC + c] -
                        SH_C1 * x * sh[3 * NC + c];
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            for (int c = 0; c < NC; c++)
                result[c] = result[c] + SH_C2[0] * xy * sh[4 * NC + c] +
                            SH_C2[1] * yz * sh[5 * NC + c] +
                            SH_C2[2] * (2.0f * zz - xx - yy) * sh[6 * NC + c] +
                            SH_C2[3] * xz * sh[7 * NC + c] +
                            SH_C2[4] * (xx - yy) * sh[8 * NC + c];
            if (deg > 2) {
                for (int c = 0; c < NC; c++)
                    result[c] = result[c] + SH_C3[0] * y * (3.0f * xx - yy) * sh[9 * NC + c] +
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

static void view_dir(const float* pos, const float* campos, float* dir_orig, float* dir)
{
    dir_orig[0] = pos[0] - campos[0];
    dir_orig[1] = pos[1] - campos[1];
    dir_orig[2] = pos[2] - campos[2];
    /* glm::length = sqrt(dot(v,v)); dot = x*x + y*y + z*z */
    float len = sqrtf(dir_orig[0] * dir_orig[0] + dir_orig[1] * dir_orig[1] + dir_orig[2] * dir_orig[2]);
    dir[0] = dir_orig[0] / len;
    dir[1] = dir_orig[1] / len;
    dir[2] = dir_orig[2] / len;
}

/* forward.cu:172-206 */
static void compute_cov3d(const float* scale, float mod, const float* rot, float* cov3D)
{
    m3 S = m3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S.m[0][0] = mod * scale[0];
    S.m[1][1] = mod * scale[1];
    S.m[2][2] = mod * scale[2];
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    m3 R = m3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                   2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                   2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    m3 M = m3_mul(&S, &R);
    m3 Mt = m3_transpose(&M);
    m3 Sigma = m3_mul(&Mt, &M);
    cov3D[0] = Sigma.m[0][0];
    cov3D[1] = Sigma.m[0][1];
    cov3D[2] = Sigma.m[0][2];
    cov3D[3] = Sigma.m[1][1];
    cov3D[4] = Sigma.m[1][2];
    cov3D[5] = Sigma.m[2][2];
}

/* forward.cu:128-167 / backward.cu:287-315 shared recomputation */
static void cov2d_terms(const float* mean, float focal_x, float focal_y,
                        float tan_fovx, float tan_fovy, const float* cov3D,
                        const float* view, float* t, float* txtz_o, float* tytz_o,
                        m3* W_o, m3* T_o, m3* Vrk_o, m3* cov_o)
{
    transform_point_4x3(mean, view, t);
    const float limx = 1.3f * tan_fovx;
    const float limy = 1.3f * tan_fovy;
    const float txtz = t[0] / t[2];
    const float tytz = t[1] / t[2];
    t[0] = fminf_(limx, fmaxf_(-limx, txtz)) * t[2];
    t[1] = fminf_(limy, fmaxf_(-limy, tytz)) * t[2];
    *txtz_o = txtz;
    *tytz_o = tytz;

    m3 J = m3_cols(focal_x / t[2], 0.0f, -(focal_x * t[0]) / (t[2] * t[2]),
                   0.0f, focal_y / t[2], -(focal_y * t[1]) / (t[2] * t[2]),
                   0, 0, 0);
    m3 W = m3_cols(view[0], view[4], view[8],
                   view[1], view[5], view[9],
                   view[2], view[6], view[10]);
    m3 T = m3_mul(&W, &J);
    m3 Vrk = m3_cols(cov3D[0], cov3D[1], cov3D[2],
                     cov3D[1], cov3D[3], cov3D[4],
                     cov3D[2], cov3D[4], cov3D[5]);
    m3 Tt = m3_transpose(&T);
    m3 Vt = m3_transpose(&Vrk);
    m3 TV = m3_mul(&Tt, &Vt);
    m3 cov = m3_mul(&TV, &T);
    *W_o = W;
    *T_o = T;
    *Vrk_o = Vrk;
    *cov_o = cov;
}

/* ---- K1: forward.cu:251-419 ---------------------------------------------- */
int gfto_preprocess_fwd(const gfto_config* cfg,
                        const float* means3D, const float* scales,
                        const float* rotations, const float* opacities,
                        const float* shs, const float* shs_p,
                        const float* cov3D_precomp, const float* colors_precomp,
                        const float* phasors_precomp,
                        const float* viewmatrix, const float* projmatrix,
                        const float* campos,
                        int32_t* radii, float* means2D, float* depths,
                        float* dists_ndc, float* cov3Ds, float* conic_opacity,
                        float* rgb, float* phasor7, float* dists,
                        float* phase_amp, uint8_t* clamped, uint8_t* clamped_p,
                        uint32_t* tiles_touched)
{
    const int P = cfg->P, D = cfg->D, M = cfg->M, M_p = cfg->M_p;
    const int W = cfg->W, H = cfg->H;
    /* rasterizer_impl.cu:249-250,261 */
    const float focal_y = H / (2.0f * cfg->tanfovy);
    const float focal_x = W / (2.0f * cfg->tanfovx);
    const int gx = (W + GFTO_BLOCK_X - 1) / GFTO_BLOCK_X;
    const int gy = (H + GFTO_BLOCK_Y - 1) / GFTO_BLOCK_Y;
    /* forward.cu:752 */
    const float dist2phase = 4.0f * PI_F / cfg->depth_range;
    const float near_n = cfg->near_n, far_n = cfg->far_n;
    const float phase_offset = cfg->phase_offset, dc_offset = cfg->dc_offset;
    int trap = 0;

#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        radii[idx] = 0;
        tiles_touched[idx] = 0;

        const float* p_orig = means3D + 3 * idx;
        float p_view[3];
        if (!in_frustum(p_orig, viewmatrix, near_n, far_n, p_view)) {
            if (cfg->prefiltered)
                trap = 1;
            continue;
        }

        float p_hom[4];
        transform_point_4x4(p_orig, projmatrix, p_hom);
        float p_w = 1.0f / (p_hom[3] + 0.0000001f);
        float p_proj[3] = {p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w};

        const float* cov3D;
        if (cov3D_precomp != NULL) {
            cov3D = cov3D_precomp + idx * 6;
        } else {
            compute_cov3d(scales + 3 * idx, cfg->scale_modifier, rotations + 4 * idx, cov3Ds + idx * 6);
            cov3D = cov3Ds + idx * 6;
        }

        float t[3], txtz, tytz;
        m3 Wm, T, Vrk, cov2;
        cov2d_terms(p_orig, focal_x, focal_y, cfg->tanfovx, cfg->tanfovy, cov3D, viewmatrix,
                    t, &txtz, &tytz, &Wm, &T, &Vrk, &cov2);
        cov2.m[0][0] += 0.3f;
        cov2.m[1][1] += 0.3f;
        float cx = cov2.m[0][0], cy = cov2.m[0][1], cz = cov2.m[1][1];

        float det = (cx * cz - cy * cy);
        if (det == 0.0f)
            continue;
        float det_inv = 1.f / det;
        float conic[3] = {cz * det_inv, -cy * det_inv, cx * det_inv};

        float mid = 0.5f * (cx + cz);
        float lambda1 = mid + sqrtf(fmaxf_(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmaxf_(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmaxf_(lambda1, lambda2)));
        float pix[2] = {ndc2pix(p_proj[0], W), ndc2pix(p_proj[1], H)};
        uint32_t rmin[2], rmax[2];
        get_rect(pix[0], pix[1], (int)my_radius, gx, gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0)
            continue;

        if (colors_precomp != NULL) {
            rgb[idx * 3 + 0] = colors_precomp[idx * 3 + 0];
            rgb[idx * 3 + 1] = colors_precomp[idx * 3 + 1];
            rgb[idx * 3 + 2] = colors_precomp[idx * 3 + 2];
        }
        if (shs != NULL) {
            /* forward.cu:20-71 */
            float dir_orig[3], dir[3], res[3];
            view_dir(p_orig, campos, dir_orig, dir);
            sh_eval(D, dir, shs + (size_t)idx * M * 3, 3, res);
            for (int c = 0; c < 3; c++) {
                res[c] += 0.5f;
                clamped[3 * idx + c] = (res[c] < 0);
                rgb[idx * 3 + c] = fmaxf_(res[c], 0.0f);
            }
        }

        float dist = sqrtf(p_view[0] * p_view[0] + p_view[1] * p_view[1] + p_view[2] * p_view[2]);
        float dist_ndc = far_n / (far_n - near_n) * (1 - near_n / dist);
        float factor = 1.0f / (dist * dist);

        float* ph = phasor7 + (size_t)idx * 7;
        if (phasors_precomp != NULL) {
            /* forward.cu:365-387: no phase_offset on this path */
            float phase = dist * dist2phase;
            float phase_sh = phasors_precomp[idx * 2 + 0];
            float amplitude = phasors_precomp[idx * 2 + 1];
            phase_amp[idx * 2 + 0] = phase_sh;
            phase_amp[idx * 2 + 1] = amplitude;
            if (cfg->use_view_dependent_phase)
                phase += phase_sh;
            ph[0] = cosf(phase) * amplitude * factor;
            ph[1] = sinf(phase) * amplitude * factor;
            ph[2] = amplitude * factor;
            ph[3] = (cosf(phase) + dc_offset) * amplitude * factor;
            ph[4] = (-cosf(phase) + dc_offset) * amplitude * factor;
            ph[5] = (sinf(phase) + dc_offset) * amplitude * factor;
            ph[6] = (-sinf(phase) + dc_offset) * amplitude * factor;
        }
        if (shs_p != NULL) {
            /* forward.cu:73-125 */
            float dir_orig[3], dir[3], res[2];
            const float* sp = shs_p + (size_t)idx * M_p * 2;
            view_dir(p_orig, campos, dir_orig, dir);
            sh_eval(D, dir, sp, 2, res);
            res[0] += 0.5f;
            res[1] += 0.5f;
            res[0] = res[0] - 0.5f - SH_C0 * sp[0];
            clamped_p[idx] = (res[1] < 0);
            if (res[1] < 0)
                res[1] = 0.0f;
            /* forward.cu:392-406 */
            float phase = dist * dist2phase + phase_offset;
            phase_amp[idx * 2 + 0] = res[0];
            phase_amp[idx * 2 + 1] = res[1];
            if (cfg->use_view_dependent_phase)
                phase += res[0];
            ph[0] = cosf(phase) * res[1] * factor;
            ph[1] = sinf(phase) * res[1] * factor;
            ph[2] = res[1] * factor;
            ph[3] = (cosf(phase) + dc_offset) * res[1] * factor;
            ph[4] = (-cosf(phase) + dc_offset) * res[1] * factor;
            ph[5] = (sinf(phase) + dc_offset) * res[1] * factor;
            ph[6] = (-sinf(phase) + dc_offset) * res[1] * factor;
        }

        dists[idx] = dist;
        depths[idx] = p_view[2];
        dists_ndc[idx] = dist_ndc;
        radii[idx] = (int32_t)my_radius;
        means2D[2 * idx + 0] = pix[0];
        means2D[2 * idx + 1] = pix[1];
        conic_opacity[4 * idx + 0] = conic[0];
        conic_opacity[4 * idx + 1] = conic[1];
        conic_opacity[4 * idx + 2] = conic[2];
        conic_opacity[4 * idx + 3] = opacities[idx];
        tiles_touched[idx] = (rmax[1] - rmin[1]) * (rmax[0] - rmin[0]);
    }
    return trap ? -1 : 0;
}

/* ---- K2: rasterizer_impl.cu:307 (cub::DeviceScan::InclusiveSum) -----------
 * Integer sums: any grouping gives the same bits.  Blocked two-pass scan so
 * that the CPU baseline uses its cores. */
uint32_t gfto_scan(int P, const uint32_t* tiles_touched, uint32_t* offsets)
{
    if (P <= 0)
        return 0;
    int nb = gfto_num_threads();
    if (nb > 256) nb = 256;
    if (P < 65536) nb = 1;
    uint32_t sums[257];
    const int chunk = (P + nb - 1) / nb;
#pragma omp parallel for schedule(static, 1) num_threads(nb)
    for (int b = 0; b < nb; b++) {
        const int i0 = b * chunk, i1 = imin_(P, i0 + chunk);
        uint32_t s = 0;
        for (int i = i0; i < i1; i++)
            s += tiles_touched[i];
        sums[b + 1] = s;
    }
    sums[0] = 0;
    for (int b = 0; b < nb; b++)
        sums[b + 1] += sums[b];
#pragma omp parallel for schedule(static, 1) num_threads(nb)
    for (int b = 0; b < nb; b++) {
        const int i0 = b * chunk, i1 = imin_(P, i0 + chunk);
        uint32_t s = sums[b];
        for (int i = i0; i < i1; i++) {
            s += tiles_touched[i];
            offsets[i] = s;
        }
    }
    return sums[nb];
}

/* ---- K3: rasterizer_impl.cu:72-113 --------------------------------------- */
void gfto_duplicate_with_keys(int P, int W, int H, const float* means2D,
                              const float* depths, const uint32_t* offsets,
                              const int32_t* radii, uint64_t* keys,
                              uint32_t* values)
{
    const int gx = (W + GFTO_BLOCK_X - 1) / GFTO_BLOCK_X;
    const int gy = (H + GFTO_BLOCK_Y - 1) / GFTO_BLOCK_Y;
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : offsets[idx - 1];
            uint32_t rmin[2], rmax[2];
            get_rect(means2D[2 * idx], means2D[2 * idx + 1], radii[idx], gx, gy, rmin, rmax);
            for (uint32_t y = rmin[1]; y < rmax[1]; y++) {
                for (uint32_t x = rmin[0]; x < rmax[0]; x++) {
                    uint64_t key = (uint64_t)(y * (uint32_t)gx + x);
                    key <<= 32;
                    uint32_t dbits;
                    memcpy(&dbits, &depths[idx], 4);
                    key |= dbits;
                    keys[off] = key;
                    values[off] = (uint32_t)idx;
                    off++;
                }
            }
        }
    }
}

/* ---- K4: cub::DeviceRadixSort::SortPairs semantics (stable LSD) ----------
 * The plain restatement: one thread, 8-bit digits over the low `end_bit` bits.
 * Kept as the definition the parallel version below is tested against
 * (tests/test_oracle_properties.py). */
void gfto_sort_pairs_serial(uint32_t R, const uint64_t* keys_in, const uint32_t* vals_in,
                            uint64_t* keys_out, uint32_t* vals_out, int end_bit)
{
    uint64_t* ka = (uint64_t*)malloc((size_t)R * 8 + 8);
    uint64_t* kb = (uint64_t*)malloc((size_t)R * 8 + 8);
    uint32_t* va = (uint32_t*)malloc((size_t)R * 4 + 4);
    uint32_t* vb = (uint32_t*)malloc((size_t)R * 4 + 4);
    memcpy(ka, keys_in, (size_t)R * 8);
    memcpy(va, vals_in, (size_t)R * 4);
    for (int shift = 0; shift < end_bit; shift += 8) {
        int bits = end_bit - shift < 8 ? end_bit - shift : 8;
        uint32_t mask = (1u << bits) - 1u;
        size_t count[257];
        memset(count, 0, sizeof(count));
        for (uint32_t i = 0; i < R; i++)
            count[((ka[i] >> shift) & mask) + 1]++;
        for (int b = 0; b < 256; b++)
            count[b + 1] += count[b];
        for (uint32_t i = 0; i < R; i++) {
            size_t d = count[(ka[i] >> shift) & mask]++;
            kb[d] = ka[i];
            vb[d] = va[i];
        }
        uint64_t* tk = ka; ka = kb; kb = tk;
        uint32_t* tv = va; va = vb; vb = tv;
    }
    memcpy(keys_out, ka, (size_t)R * 8);
    memcpy(vals_out, va, (size_t)R * 4);
    free(ka); free(kb); free(va); free(vb);
}

/* One bucket of the parallel sort: stable LSD radix over the low 32 key bits
 * (the depth bits), ping-pong between (k0,v0) = the bucket's place in the
 * output and thread-local scratch; four passes end in (k0,v0). */
static void sort_bucket_low32(uint64_t* k0, uint32_t* v0, uint64_t* k1, uint32_t* v1, size_t n)
{
    if (n < 2)
        return;
    if (n <= 32) { /* stable insertion sort */
        for (size_t i = 1; i < n; i++) {
            const uint64_t k = k0[i];
            const uint32_t v = v0[i], lo = (uint32_t)k;
            size_t j = i;
            while (j > 0 && (uint32_t)k0[j - 1] > lo) {
                k0[j] = k0[j - 1];
                v0[j] = v0[j - 1];
                j--;
            }
            k0[j] = k;
            v0[j] = v;
        }
        return;
    }
    uint64_t* ka = k0; uint32_t* va = v0;
    uint64_t* kb = k1; uint32_t* vb = v1;
    for (int shift = 0; shift < 32; shift += 8) {
        size_t count[257];
        memset(count, 0, sizeof(count));
        for (size_t i = 0; i < n; i++)
            count[((ka[i] >> shift) & 255u) + 1]++;
        for (int b = 0; b < 256; b++)
            count[b + 1] += count[b];
        for (size_t i = 0; i < n; i++) {
            size_t d = count[(ka[i] >> shift) & 255u]++;
            kb[d] = ka[i];
            vb[d] = va[i];
        }
        uint64_t* tk = ka; ka = kb; kb = tk;
        uint32_t* tv = va; va = vb; vb = tv;
    }
}

/* The same result on all cores.  A stable sort on the low `end_bit` bits of
 * keys `tile << 32 | depth bits` (rasterizer_impl.cu:331-339: end_bit = 32 +
 * getHigherMsb(tiles)) is: group by the `end_bit - 32` tile bits keeping the
 * input order (per-thread histograms over contiguous input chunks, chunks
 * placed in order), then a stable sort of every group on the low 32 bits.
 * The order is unique, so the output equals gfto_sort_pairs_serial's bit for
 * bit. */
void gfto_sort_pairs(uint32_t R, const uint64_t* keys_in, const uint32_t* vals_in,
                     uint64_t* keys_out, uint32_t* vals_out, int end_bit)
{
    const int hb = end_bit - 32;
    if (R < 32768u || hb < 1 || hb > 20) {
        gfto_sort_pairs_serial(R, keys_in, vals_in, keys_out, vals_out, end_bit);
        return;
    }
    const size_t NB = (size_t)1 << hb;
    const uint32_t hmask = (uint32_t)(NB - 1);
    int nt = gfto_num_threads();
    if (nt > 256) nt = 256;
    const size_t chunk = ((size_t)R + nt - 1) / nt;
    uint32_t* hist = (uint32_t*)calloc((size_t)nt * NB, 4);
    uint32_t* base = (uint32_t*)malloc((NB + 1) * 4);
#pragma omp parallel for schedule(static, 1) num_threads(nt)
    for (int t = 0; t < nt; t++) {
        uint32_t* h = hist + (size_t)t * NB;
        size_t i0 = (size_t)t * chunk, i1 = i0 + chunk < R ? i0 + chunk : R;
        for (size_t i = i0; i < i1; i++)
            h[(uint32_t)(keys_in[i] >> 32) & hmask]++;
    }
    /* start of (thread, bucket) = start of the bucket + what earlier threads hold of it */
    size_t maxlen = 0;
    {
        uint32_t s = 0;
        for (size_t b = 0; b < NB; b++) {
            base[b] = s;
            uint32_t run = s;
            for (int t = 0; t < nt; t++) {
                uint32_t c = hist[(size_t)t * NB + b];
                hist[(size_t)t * NB + b] = run;
                run += c;
            }
            if (run - s > maxlen) maxlen = run - s;
            s = run;
        }
        base[NB] = s;
    }
#pragma omp parallel for schedule(static, 1) num_threads(nt)
    for (int t = 0; t < nt; t++) {
        uint32_t* h = hist + (size_t)t * NB;
        size_t i0 = (size_t)t * chunk, i1 = i0 + chunk < R ? i0 + chunk : R;
        for (size_t i = i0; i < i1; i++) {
            const uint64_t k = keys_in[i];
            const uint32_t d = h[(uint32_t)(k >> 32) & hmask]++;
            keys_out[d] = k;
            vals_out[d] = vals_in[i];
        }
    }
#pragma omp parallel num_threads(nt)
    {
        uint64_t* k1 = (uint64_t*)malloc(maxlen * 8 + 8);
        uint32_t* v1 = (uint32_t*)malloc(maxlen * 4 + 4);
#pragma omp for schedule(dynamic, 4)
        for (size_t b = 0; b < NB; b++)
            sort_bucket_low32(keys_out + base[b], vals_out + base[b], k1, v1, (size_t)(base[b + 1] - base[b]));
        free(k1);
        free(v1);
    }
    free(hist);
    free(base);
}

/* ---- K5: rasterizer_impl.cu:118-140 + memset :341 ------------------------
 * One "thread" per sorted instance, as in the reference: every boundary is
 * written by exactly one index. */
void gfto_tile_ranges(uint32_t R, const uint64_t* keys, int T, uint32_t* ranges)
{
    memset(ranges, 0, (size_t)T * 8);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)R; i++) {
        const uint32_t idx = (uint32_t)i;
        uint32_t currtile = (uint32_t)(keys[idx] >> 32);
        if (idx == 0)
            ranges[2 * currtile + 0] = 0;
        else {
            uint32_t prevtile = (uint32_t)(keys[idx - 1] >> 32);
            if (currtile != prevtile) {
                ranges[2 * prevtile + 1] = idx;
                ranges[2 * currtile + 0] = idx;
            }
        }
        if (idx == R - 1)
            ranges[2 * currtile + 1] = R;
    }
}

/* parallel zero fill for the driver's kept output arrays (oracle.py: reuse_buffers) */
void gfto_zero(void* p, size_t bytes)
{
    const size_t step = (size_t)1 << 20;
    const int64_t n = (int64_t)((bytes + step - 1) / step);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        size_t o = (size_t)i * step;
        memset((char*)p + o, 0, bytes - o < step ? bytes - o : step);
    }
}

/* ---- K6: forward.cu:424-676 ---------------------------------------------- */
void gfto_render_fwd(int W, int H, const uint32_t* ranges,
                     const uint32_t* point_list, const float* means2D,
                     const float* rgb, const float* phasor7, const float* dists,
                     const float* conic_opacity, const float* dists_ndc,
                     const float* bg,
                     float* final_T, uint32_t* n_contrib, float* w_z_total,
                     float* w_z2_total,
                     float* out_color, float* out_phasor, float* out_depth,
                     float* out_acc, float* out_depth_distortion,
                     float* out_distribution, float* pixels)
{
    const int gx = (W + GFTO_BLOCK_X - 1) / GFTO_BLOCK_X;
    const int gy = (H + GFTO_BLOCK_Y - 1) / GFTO_BLOCK_Y;
    const size_t HW = (size_t)H * W;

#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int ly = 0; ly < GFTO_BLOCK_Y; ly++) {
            for (int lx = 0; lx < GFTO_BLOCK_X; lx++) {
                const uint32_t px = (uint32_t)(tx * GFTO_BLOCK_X + lx);
                const uint32_t py = (uint32_t)(ty * GFTO_BLOCK_Y + ly);
                if (!(px < (uint32_t)W && py < (uint32_t)H))
                    continue;
                const size_t pix_id = (size_t)W * py + px;
                const float pixf[2] = {(float)px, (float)py};

                float T = 1.0f;
                uint32_t contributor = 0, last_contributor = 0;
                float C[3] = {0, 0, 0};
                float Pp[7] = {0, 0, 0, 0, 0, 0, 0};
                float Dd = 0, A = 0, DD = 0, DD_D = 0, DD_D2 = 0;
                float WD[3] = {0, 0, 0};
                int gs_idx = 0;

                /* A pixel thread processes list entries until it is `done`
                 * (forward.cu:517,539-543); block-level early exit only skips
                 * entries no thread would use. */
                for (uint32_t k = r0; k < r1; k++) {
                    contributor++;
                    const uint32_t id = point_list[k];
                    const float dx = means2D[2 * id] - pixf[0];
                    const float dy = means2D[2 * id + 1] - pixf[1];
                    const float* con_o = conic_opacity + 4 * (size_t)id;
                    float power = -0.5f * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;
                    if (power > 0.0f)
                        continue;
                    float alpha = fminf_(0.99f, con_o[3] * expf(power));
                    if (alpha < 1.0f / 255.0f)
                        continue;
                    float test_T = T * (1 - alpha);
                    if (test_T < 0.0001f)
                        break; /* done = true */

                    float w = alpha * T;
                    float w_p = alpha * T * T;
                    for (int ch = 0; ch < 3; ch++)
                        C[ch] += rgb[id * 3 + ch] * w;
                    for (int ch = 0; ch < 7; ch++)
                        Pp[ch] += phasor7[(size_t)id * 7 + ch] * w_p;
                    Dd += dists[id] * w;
                    if (gs_idx < 1) {
                        WD[0] = alpha;
                        WD[1] = dists[id];
                        WD[2] = phasor7[(size_t)id * 7 + 2];
                    }
                    gs_idx += 1;

                    float z = dists_ndc[id];
                    DD += w * (z * z * A - 2.0f * z * DD_D + DD_D2);
                    DD_D += w * z;
                    DD_D2 += w * z * z;

                    A += alpha * T;
                    T = test_T;
                    last_contributor = contributor;
#pragma omp atomic
                    pixels[id] += 1.0f;
                }

                final_T[pix_id] = T;
                n_contrib[pix_id] = last_contributor;
                for (int ch = 0; ch < 3; ch++)
                    out_color[ch * HW + pix_id] = C[ch] + T * bg[ch * HW + pix_id];
                for (int ch = 0; ch < 7; ch++)
                    out_phasor[ch * HW + pix_id] = Pp[ch] + T * bg[ch * HW + pix_id];
                out_depth[pix_id] = Dd;
                out_acc[pix_id] = A;
                w_z_total[pix_id] = DD_D;
                w_z2_total[pix_id] = DD_D2;
                out_depth_distortion[pix_id] = DD;
                out_distribution[0 * HW + pix_id] = WD[0];
                out_distribution[1 * HW + pix_id] = WD[1];
                out_distribution[2 * HW + pix_id] = WD[2];
            }
        }
    }
}

/* ---- K7: backward.cu:609-889 --------------------------------------------- */
void gfto_render_bwd(int W, int H, const uint32_t* ranges,
                     const uint32_t* point_list, int P, const float* bg,
                     const float* means2D, const float* conic_opacity,
                     const float* rgb, const float* phasor7, const float* dists,
                     const float* dists_ndc, const float* final_T,
                     const float* w_z_total, const float* w_z2_total,
                     const uint32_t* n_contrib,
                     const float* dL_dcolor, const float* dL_dphasor,
                     const float* dL_ddepth, const float* dL_dacc,
                     const float* dL_ddd, float* acc)
{
    const int gx = (W + GFTO_BLOCK_X - 1) / GFTO_BLOCK_X;
    const int gy = (H + GFTO_BLOCK_Y - 1) / GFTO_BLOCK_Y;
    const size_t HW = (size_t)H * W;
    /* double accumulators, kept between calls and all zero outside this function (the fold below
     * re-zeroes what it reads): one caller at a time, as everywhere in the oracle */
    static double* dacc_keep = NULL;
    static size_t dacc_n = 0;
    const size_t nacc = (size_t)P * GFTO_NUM_ACC;
    if (nacc > dacc_n) {
        free(dacc_keep);
        dacc_keep = (double*)malloc(nacc * sizeof(double));
        dacc_n = nacc;
        gfto_zero(dacc_keep, nacc * sizeof(double));
    }
    double* dacc = dacc_keep;

    /* backward.cu:708-709 */
    const float ddelx_dx = (float)(0.5 * W);
    const float ddely_dy = (float)(0.5 * H);

#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        const uint32_t len = r1 - r0;
        if (len == 0)
            continue;
        /* per-tile partial sums (double) per list entry, then one add per
         * (tile, Gaussian) into the global double accumulators */
        uint32_t max_contrib = 0;
        for (int ly = 0; ly < GFTO_BLOCK_Y; ly++)
            for (int lx = 0; lx < GFTO_BLOCK_X; lx++) {
                const uint32_t px = (uint32_t)(tx * GFTO_BLOCK_X + lx);
                const uint32_t py = (uint32_t)(ty * GFTO_BLOCK_Y + ly);
                if (px < (uint32_t)W && py < (uint32_t)H) {
                    uint32_t nc = n_contrib[(size_t)W * py + px];
                    if (nc > max_contrib)
                        max_contrib = nc;
                }
            }
        if (max_contrib == 0)
            continue;
        double* part = (double*)calloc((size_t)max_contrib * GFTO_NUM_ACC, sizeof(double));

        for (int ly = 0; ly < GFTO_BLOCK_Y; ly++) {
            for (int lx = 0; lx < GFTO_BLOCK_X; lx++) {
                const uint32_t px = (uint32_t)(tx * GFTO_BLOCK_X + lx);
                const uint32_t py = (uint32_t)(ty * GFTO_BLOCK_Y + ly);
                if (!(px < (uint32_t)W && py < (uint32_t)H))
                    continue;
                const size_t pix_id = (size_t)W * py + px;
                const float pixf[2] = {(float)px, (float)py};

                const float T_final = final_T[pix_id];
                float T = T_final;
                const uint32_t last_contributor = n_contrib[pix_id];
                const float wz_tot = w_z_total[pix_id];
                const float wz2_tot = w_z2_total[pix_id];

                float accum_rec[3] = {0, 0, 0}, accum_rec_p[7] = {0, 0, 0, 0, 0, 0, 0};
                float accum_rec_d = 0, accum_rec_a = 0, accum_rec_dd = 0;
                float dpix[3], dpix_p[7];
                for (int i = 0; i < 3; i++)
                    dpix[i] = dL_dcolor[i * HW + pix_id];
                for (int i = 0; i < 7; i++)
                    dpix_p[i] = dL_dphasor[i * HW + pix_id];
                const float dpix_d = dL_ddepth[pix_id];
                const float dpix_a = dL_dacc[pix_id];
                const float dpix_dd = dL_ddd[pix_id];

                float last_alpha = 0;
                float last_color[3] = {0, 0, 0}, last_phasor[7] = {0, 0, 0, 0, 0, 0, 0};
                float last_dist = 0, last_dL_dw = 0;

                /* entries with contributor >= last_contributor are skipped
                 * (backward.cu:739-741): start directly at last_contributor-1 */
                for (uint32_t c = last_contributor; c-- > 0;) {
                    const uint32_t id = point_list[r0 + c];
                    const float dx = means2D[2 * id] - pixf[0];
                    const float dy = means2D[2 * id + 1] - pixf[1];
                    const float* con_o = conic_opacity + 4 * (size_t)id;
                    const float power = -0.5f * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;
                    if (power > 0.0f)
                        continue;
                    const float G = expf(power);
                    const float alpha = fminf_(0.99f, con_o[3] * G);
                    if (alpha < 1.0f / 255.0f)
                        continue;

                    T = T / (1.f - alpha);
                    const float dchannel_dcolor = alpha * T;
                    const float dchannel_dphasor = alpha * T * T;
                    const float dchannel_ddepth = alpha * T;

                    double* pa = part + (size_t)c * GFTO_NUM_ACC;
                    float dL_dalpha = 0.0f, dL_dalpha_c = 0.0f, dL_dalpha_p = 0.0f;
                    float dL_dalpha_d = 0.0f, dL_dalpha_a = 0.0f, dL_dalpha_dd = 0.0f;

                    for (int ch = 0; ch < 3; ch++) {
                        const float cc = rgb[id * 3 + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                        last_color[ch] = cc;
                        const float dL_dchannel = dpix[ch];
                        dL_dalpha_c += (cc - accum_rec[ch]) * dL_dchannel;
                        pa[6 + ch] += (double)(dchannel_dcolor * dL_dchannel);
                    }
                    dL_dalpha_c *= T;

                    for (int ch = 0; ch < 7; ch++) {
                        const float p = phasor7[(size_t)id * 7 + ch];
                        accum_rec_p[ch] = last_alpha * last_phasor[ch] +
                                          (1.f - last_alpha) * (1.f - last_alpha) * accum_rec_p[ch];
                        last_phasor[ch] = p;
                        const float dL_dchannel_p = dpix_p[ch];
                        dL_dalpha_p += (p - 2.f * (1.f - alpha) * accum_rec_p[ch]) * dL_dchannel_p;
                        pa[9 + ch] += (double)(dchannel_dphasor * dL_dchannel_p);
                    }
                    dL_dalpha_p *= T * T;

                    const float dist = dists[id];
                    accum_rec_d = last_alpha * last_dist + (1.f - last_alpha) * accum_rec_d;
                    last_dist = dist;
                    dL_dalpha_d += (dist - accum_rec_d) * dpix_d;
                    pa[16] += (double)(dchannel_ddepth * dpix_d);
                    dL_dalpha_d *= T;

                    accum_rec_a = last_alpha + (1.f - last_alpha) * accum_rec_a;
                    dL_dalpha_a += (1.f - accum_rec_a) * dpix_a;
                    dL_dalpha_a *= T;

                    const float z = dists_ndc[id];
                    float dL_dw = dpix_dd * (z * z * (1 - T_final) - 2.0f * z * wz_tot + wz2_tot);
                    accum_rec_dd = last_alpha * last_dL_dw + (1.f - last_alpha) * accum_rec_dd;
                    last_dL_dw = dL_dw;
                    dL_dalpha_dd += dL_dw - accum_rec_dd;
                    pa[17] += (double)(dpix_dd * 2.0f * alpha * T * (z * (1 - T_final) - wz_tot));
                    dL_dalpha_dd *= T;

                    last_alpha = alpha;

                    float bg_dot_dpixel = 0;
                    for (int i = 0; i < 3; i++)
                        bg_dot_dpixel += bg[i * HW + pix_id] * dpix[i];
                    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;

                    float bg_dot_dpixel_p = 0;
                    for (int i = 0; i < 7; i++)
                        bg_dot_dpixel_p += bg[i * HW + pix_id] * dpix_p[i];
                    dL_dalpha_p += (-T_final / (1.f - alpha)) * bg_dot_dpixel_p;

                    dL_dalpha += dL_dalpha_c;
                    dL_dalpha += dL_dalpha_p;
                    dL_dalpha += dL_dalpha_d;
                    dL_dalpha += dL_dalpha_a;
                    dL_dalpha += dL_dalpha_dd;

                    const float dL_dG = con_o[3] * dL_dalpha;
                    const float gdx = G * dx;
                    const float gdy = G * dy;
                    const float dG_ddelx = -gdx * con_o[0] - gdy * con_o[1];
                    const float dG_ddely = -gdy * con_o[2] - gdx * con_o[1];

                    pa[0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
                    pa[1] += (double)(dL_dG * dG_ddely * ddely_dy);
                    pa[2] += (double)(-0.5f * gdx * dx * dL_dG);
                    pa[3] += (double)(-0.5f * gdx * dy * dL_dG);
                    pa[4] += (double)(-0.5f * gdy * dy * dL_dG);
                    pa[5] += (double)(G * dL_dalpha);
                }
            }
        }
        for (uint32_t c = 0; c < max_contrib; c++) {
            const uint32_t id = point_list[r0 + c];
            for (int k = 0; k < GFTO_NUM_ACC; k++) {
                double v = part[(size_t)c * GFTO_NUM_ACC + k];
                if (v != 0.0) {
#pragma omp atomic
                    dacc[(size_t)id * GFTO_NUM_ACC + k] += v;
                }
            }
        }
        free(part);
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)nacc; i++) {
        if (dacc[i] != 0.0) {
            acc[i] += (float)dacc[i];
            dacc[i] = 0.0;
        }
    }
}

/* The 18 sums per Gaussian as the arrays the reference's backward keeps apart (rasterize_points.cu:222-236):
 * plain copies, on all cores. */
void gfto_unpack_acc(int P, const float* acc, float* dL_dmeans2D /*[P,3]*/, float* dL_dconic /*[P,4]*/,
                     float* dL_dopacity, float* dL_dcolors /*[P,3]*/, float* dL_dphasors /*[P,7]*/,
                     float* dL_ddist, float* dL_dndc)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        const float* a = acc + (size_t)i * GFTO_NUM_ACC;
        dL_dmeans2D[3 * (size_t)i + 0] = a[0];
        dL_dmeans2D[3 * (size_t)i + 1] = a[1];
        dL_dconic[4 * (size_t)i + 0] = a[2];
        dL_dconic[4 * (size_t)i + 1] = a[3];
        dL_dconic[4 * (size_t)i + 3] = a[4];
        dL_dopacity[i] = a[5];
        for (int k = 0; k < 3; k++) dL_dcolors[3 * (size_t)i + k] = a[6 + k];
        for (int k = 0; k < 7; k++) dL_dphasors[7 * (size_t)i + k] = a[9 + k];
        dL_ddist[i] = a[16];
        dL_dndc[i] = a[17];
    }
}

/* ---- SH backward shared by colour/phasor: backward.cu:20-139,143-260 ------
 * dL_dres[NC] is the (already clamp-masked) gradient of the SH polynomial
 * value; writes dL_dsh[k*NC+c] for k < (deg+1)^2 and returns dL_ddir. */
static void sh_bwd(int deg, const float* dir, const float* sh, int NC,
                   const float* dL_dres, float* dL_dsh, float* dL_ddir)
{
    float x = dir[0], y = dir[1], z = dir[2];
    float ddx[3] = {0, 0, 0}, ddy[3] = {0, 0, 0}, ddz[3] = {0, 0, 0}; /* per channel, NC <= 3 */

    float d0 = SH_C0;
    for (int c = 0; c < NC; c++)
        dL_dsh[0 * NC + c] = d0 * dL_dres[c];
    if (deg > 0) {
        float d1 = -SH_C1 * y, d2 = SH_C1 * z, d3 = -SH_C1 * x;
        for (int c = 0; c < NC; c++) {
            dL_dsh[1 * NC + c] = d1 * dL_dres[c];
            dL_dsh[2 * NC + c] = d2 * dL_dres[c];
            dL_dsh[3 * NC + c] = d3 * dL_dres[c];
            ddx[c] = -SH_C1 * sh[3 * NC + c];
            ddy[c] = -SH_C1 * sh[1 * NC + c];
            ddz[c] = SH_C1 * sh[2 * NC + c];
        }
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            float d4 = SH_C2[0] * xy, d5 = SH_C2[1] * yz, d6 = SH_C2[2] * (2.f * zz - xx - yy);
            float d7 = SH_C2[3] * xz, d8 = SH_C2[4] * (xx - yy);
            for (int c = 0; c < NC; c++) {
                const float* s = sh + c;
                dL_dsh[4 * NC + c] = d4 * dL_dres[c];
                dL_dsh[5 * NC + c] = d5 * dL_dres[c];
                dL_dsh[6 * NC + c] = d6 * dL_dres[c];
                dL_dsh[7 * NC + c] = d7 * dL_dres[c];
                dL_dsh[8 * NC + c] = d8 * dL_dres[c];
                /* backward.cu:78-80; (a*b*c) chains are left-to-right */
                ddx[c] += SH_C2[0] * y * s[4 * NC] + SH_C2[2] * 2.f * -x * s[6 * NC] +
                          SH_C2[3] * z * s[7 * NC] + SH_C2[4] * 2.f * x * s[8 * NC];
                ddy[c] += SH_C2[0] * x * s[4 * NC] + SH_C2[1] * z * s[5 * NC] +
                          SH_C2[2] * 2.f * -y * s[6 * NC] + SH_C2[4] * 2.f * -y * s[8 * NC];
                ddz[c] += SH_C2[1] * y * s[5 * NC] + SH_C2[2] * 2.f * 2.f * z * s[6 * NC] +
                          SH_C2[3] * x * s[7 * NC];
            }
            if (deg > 2) {
                float d9 = SH_C3[0] * y * (3.f * xx - yy);
                float d10 = SH_C3[1] * xy * z;
                float d11 = SH_C3[2] * y * (4.f * zz - xx - yy);
                float d12 = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                float d13 = SH_C3[4] * x * (4.f * zz - xx - yy);
                float d14 = SH_C3[5] * z * (xx - yy);
                float d15 = SH_C3[6] * x * (xx - 3.f * yy);
                for (int c = 0; c < NC; c++) {
                    const float* s = sh + c;
                    dL_dsh[9 * NC + c] = d9 * dL_dres[c];
                    dL_dsh[10 * NC + c] = d10 * dL_dres[c];
                    dL_dsh[11 * NC + c] = d11 * dL_dres[c];
                    dL_dsh[12 * NC + c] = d12 * dL_dres[c];
                    dL_dsh[13 * NC + c] = d13 * dL_dres[c];
                    dL_dsh[14 * NC + c] = d14 * dL_dres[c];
                    dL_dsh[15 * NC + c] = d15 * dL_dres[c];
                    /* backward.cu:99-122: "C * sh * k * v" = ((C*sh)*k)*v */
                    ddx[c] += (SH_C3[0] * s[9 * NC] * 3.f * 2.f * xy +
                               SH_C3[1] * s[10 * NC] * yz +
                               SH_C3[2] * s[11 * NC] * -2.f * xy +
                               SH_C3[3] * s[12 * NC] * -3.f * 2.f * xz +
                               SH_C3[4] * s[13 * NC] * (-3.f * xx + 4.f * zz - yy) +
                               SH_C3[5] * s[14 * NC] * 2.f * xz +
                               SH_C3[6] * s[15 * NC] * 3.f * (xx - yy));
                    ddy[c] += (SH_C3[0] * s[9 * NC] * 3.f * (xx - yy) +
                               SH_C3[1] * s[10 * NC] * xz +
                               SH_C3[2] * s[11 * NC] * (-3.f * yy + 4.f * zz - xx) +
                               SH_C3[3] * s[12 * NC] * -3.f * 2.f * yz +
                               SH_C3[4] * s[13 * NC] * -2.f * xy +
                               SH_C3[5] * s[14 * NC] * -2.f * yz +
                               SH_C3[6] * s[15 * NC] * -3.f * 2.f * xy);
                    ddz[c] += (SH_C3[1] * s[10 * NC] * xy +
                               SH_C3[2] * s[11 * NC] * 4.f * 2.f * yz +
                               SH_C3[3] * s[12 * NC] * 3.f * (2.f * zz - xx - yy) +
                               SH_C3[4] * s[13 * NC] * 4.f * 2.f * xz +
                               SH_C3[5] * s[14 * NC] * (xx - yy));
                }
            }
        }
    }
    /* glm::dot: x*x' + y*y' (+ z*z') left to right */
    float sx = 0, sy = 0, sz = 0;
    for (int c = 0; c < NC; c++) {
        sx = (c == 0) ? ddx[c] * dL_dres[c] : sx + ddx[c] * dL_dres[c];
        sy = (c == 0) ? ddy[c] * dL_dres[c] : sy + ddy[c] * dL_dres[c];
        sz = (c == 0) ? ddz[c] * dL_dres[c] : sz + ddz[c] * dL_dres[c];
    }
    dL_ddir[0] = sx;
    dL_ddir[1] = sy;
    dL_ddir[2] = sz;
}

/* backward.cu:399-462 */
static void compute_cov3d_bwd(const float* scale, float mod, const float* rot,
                              const float* dL_dcov3D, float* dL_dscale, float* dL_drot)
{
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    m3 R = m3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                   2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                   2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    m3 S = m3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    S.m[0][0] = s[0];
    S.m[1][1] = s[1];
    S.m[2][2] = s[2];
    m3 M = m3_mul(&S, &R);

    m3 dL_dSigma = m3_cols(dL_dcov3D[0], 0.5f * dL_dcov3D[1], 0.5f * dL_dcov3D[2],
                           0.5f * dL_dcov3D[1], dL_dcov3D[3], 0.5f * dL_dcov3D[4],
                           0.5f * dL_dcov3D[2], 0.5f * dL_dcov3D[4], dL_dcov3D[5]);
    /* 2.0f * M * dL_dSigma = (2.0f * M) * dL_dSigma */
    m3 M2;
    for (int c = 0; c < 3; c++)
        for (int rr = 0; rr < 3; rr++)
            M2.m[c][rr] = 2.0f * M.m[c][rr];
    m3 dL_dM = m3_mul(&M2, &dL_dSigma);
    m3 Rt = m3_transpose(&R);
    m3 dL_dMt = m3_transpose(&dL_dM);

    for (int i = 0; i < 3; i++)
        dL_dscale[i] = Rt.m[i][0] * dL_dMt.m[i][0] + Rt.m[i][1] * dL_dMt.m[i][1] +
                       Rt.m[i][2] * dL_dMt.m[i][2];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            dL_dMt.m[i][j] *= s[i];

#define DM(i, j) dL_dMt.m[i][j]
    dL_drot[0] = 2 * z * (DM(0, 1) - DM(1, 0)) + 2 * y * (DM(2, 0) - DM(0, 2)) + 2 * x * (DM(1, 2) - DM(2, 1));
    dL_drot[1] = 2 * y * (DM(1, 0) + DM(0, 1)) + 2 * z * (DM(2, 0) + DM(0, 2)) + 2 * r * (DM(1, 2) - DM(2, 1)) - 4 * x * (DM(2, 2) + DM(1, 1));
    dL_drot[2] = 2 * x * (DM(1, 0) + DM(0, 1)) + 2 * r * (DM(2, 0) - DM(0, 2)) + 2 * z * (DM(1, 2) + DM(2, 1)) - 4 * y * (DM(2, 2) + DM(0, 0));
    dL_drot[3] = 2 * r * (DM(0, 1) - DM(1, 0)) + 2 * x * (DM(2, 0) + DM(0, 2)) + 2 * y * (DM(1, 2) + DM(2, 1)) - 4 * z * (DM(1, 1) + DM(0, 0));
#undef DM
}

/* ---- K8 + K9: backward.cu:265-395, 467-606 ------------------------------- */
void gfto_preprocess_bwd(const gfto_config* cfg,
                         const float* means3D, const int32_t* radii,
                         const float* shs, const float* shs_p,
                         const uint8_t* clamped, const uint8_t* clamped_p,
                         const float* scales, const float* rotations,
                         const float* cov3Ds,
                         const float* view, const float* proj,
                         const float* campos,
                         const float* dL_dmean2D, const float* dL_dconics,
                         const float* dL_dcolor, const float* dL_dphasor,
                         const float* dL_ddist_in, const float* dL_ddist_ndc,
                         const float* phase_amp, const float* dists,
                         float* dL_dmeans, float* dL_dcov, float* dL_dsh,
                         float* dL_dsh_p, float* dL_dscales, float* dL_drots,
                         float* dL_dphase_offset, float* dL_ddc_offset)
{
    const int P = cfg->P, D = cfg->D, M = cfg->M, M_p = cfg->M_p;
    const float h_y = cfg->H / (2.0f * cfg->tanfovy);
    const float h_x = cfg->W / (2.0f * cfg->tanfovx);
    const float dist2phase = 4.0f * PI_F / cfg->depth_range;
    const float near_n = cfg->near_n, far_n = cfg->far_n;
    const float phase_offset = cfg->phase_offset, dc_offset = cfg->dc_offset;
    double sum_phase = 0.0, sum_dc = 0.0;

#pragma omp parallel for schedule(static) reduction(+ : sum_phase, sum_dc)
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0))
            continue;

        /* ===== K8 computeCov2DCUDA ===== */
        {
            const float* cov3D = cov3Ds + 6 * idx;
            const float* mean = means3D + 3 * idx;
            float dL_dconic[3] = {dL_dconics[4 * idx], dL_dconics[4 * idx + 1], dL_dconics[4 * idx + 3]};
            float t[3], txtz, tytz;
            m3 Wm, T, Vrk, cov2D;
            cov2d_terms(mean, h_x, h_y, cfg->tanfovx, cfg->tanfovy, cov3D, view,
                        t, &txtz, &tytz, &Wm, &T, &Vrk, &cov2D);
            const float limx = 1.3f * cfg->tanfovx;
            const float limy = 1.3f * cfg->tanfovy;
            const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0 : 1;
            const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0 : 1;

            float a = cov2D.m[0][0] += 0.3f;
            float b = cov2D.m[0][1];
            float c = cov2D.m[1][1] += 0.3f;

            float denom = a * c - b * b;
            float dL_da = 0, dL_db = 0, dL_dc = 0;
            float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);

#define TT(i, j) T.m[i][j]
#define VV(i, j) Vrk.m[i][j]
            if (denom2inv != 0) {
                dL_da = denom2inv * (-c * c * dL_dconic[0] + 2 * b * c * dL_dconic[1] + (denom - a * c) * dL_dconic[2]);
                dL_dc = denom2inv * (-a * a * dL_dconic[2] + 2 * a * b * dL_dconic[1] + (denom - a * c) * dL_dconic[0]);
                dL_db = denom2inv * 2 * (b * c * dL_dconic[0] - (denom + 2 * b * b) * dL_dconic[1] + a * b * dL_dconic[2]);

                dL_dcov[6 * idx + 0] = (TT(0, 0) * TT(0, 0) * dL_da + TT(0, 0) * TT(1, 0) * dL_db + TT(1, 0) * TT(1, 0) * dL_dc);
                dL_dcov[6 * idx + 3] = (TT(0, 1) * TT(0, 1) * dL_da + TT(0, 1) * TT(1, 1) * dL_db + TT(1, 1) * TT(1, 1) * dL_dc);
                dL_dcov[6 * idx + 5] = (TT(0, 2) * TT(0, 2) * dL_da + TT(0, 2) * TT(1, 2) * dL_db + TT(1, 2) * TT(1, 2) * dL_dc);
                dL_dcov[6 * idx + 1] = 2 * TT(0, 0) * TT(0, 1) * dL_da + (TT(0, 0) * TT(1, 1) + TT(0, 1) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 1) * dL_dc;
                dL_dcov[6 * idx + 2] = 2 * TT(0, 0) * TT(0, 2) * dL_da + (TT(0, 0) * TT(1, 2) + TT(0, 2) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 2) * dL_dc;
                dL_dcov[6 * idx + 4] = 2 * TT(0, 2) * TT(0, 1) * dL_da + (TT(0, 1) * TT(1, 2) + TT(0, 2) * TT(1, 1)) * dL_db + 2 * TT(1, 1) * TT(1, 2) * dL_dc;
            } else {
                for (int i = 0; i < 6; i++)
                    dL_dcov[6 * idx + i] = 0;
            }

            float dL_dT00 = 2 * (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_da +
                            (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_db;
            float dL_dT01 = 2 * (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_da +
                            (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_db;
            float dL_dT02 = 2 * (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_da +
                            (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_db;
            float dL_dT10 = 2 * (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_dc +
                            (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_db;
            float dL_dT11 = 2 * (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_dc +
                            (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_db;
            float dL_dT12 = 2 * (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_dc +
                            (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_db;
#undef TT
#undef VV
#define WW(i, j) Wm.m[i][j]
            float dL_dJ00 = WW(0, 0) * dL_dT00 + WW(0, 1) * dL_dT01 + WW(0, 2) * dL_dT02;
            float dL_dJ02 = WW(2, 0) * dL_dT00 + WW(2, 1) * dL_dT01 + WW(2, 2) * dL_dT02;
            float dL_dJ11 = WW(1, 0) * dL_dT10 + WW(1, 1) * dL_dT11 + WW(1, 2) * dL_dT12;
            float dL_dJ12 = WW(2, 0) * dL_dT10 + WW(2, 1) * dL_dT11 + WW(2, 2) * dL_dT12;
#undef WW
            float tz = 1.f / t[2];
            float tz2 = tz * tz;
            float tz3 = tz2 * tz;

            float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
            float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
            float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t[0]) * tz3 * dL_dJ02 + (2 * h_y * t[1]) * tz3 * dL_dJ12;

            /* transformVec4x3Transpose, auxiliary.h:92-100; assignment (:394) */
            dL_dmeans[3 * idx + 0] = view[0] * dL_dtx + view[1] * dL_dty + view[2] * dL_dtz;
            dL_dmeans[3 * idx + 1] = view[4] * dL_dtx + view[5] * dL_dty + view[6] * dL_dtz;
            dL_dmeans[3 * idx + 2] = view[8] * dL_dtx + view[9] * dL_dty + view[10] * dL_dtz;
        }

        /* ===== K9 preprocessCUDA (backward) ===== */
        const float* m = means3D + 3 * idx;
        float m_hom[4], m_view[3];
        transform_point_4x4(m, proj, m_hom);
        float m_w = 1.0f / (m_hom[3] + 0.0000001f);
        transform_point_4x3(m, view, m_view);

        float mul1 = (proj[0] * m[0] + proj[4] * m[1] + proj[8] * m[2] + proj[12]) * m_w * m_w;
        float mul2 = (proj[1] * m[0] + proj[5] * m[1] + proj[9] * m[2] + proj[13]) * m_w * m_w;
        const float g2x = dL_dmean2D[3 * idx + 0], g2y = dL_dmean2D[3 * idx + 1];
        float dm[3];
        dm[0] = (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
        dm[1] = (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
        dm[2] = (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
        dL_dmeans[3 * idx + 0] += dm[0];
        dL_dmeans[3 * idx + 1] += dm[1];
        dL_dmeans[3 * idx + 2] += dm[2];

        if (shs != NULL) {
            float dir_orig[3], dir[3], dres[3], dL_ddir[3], dmean[3];
            view_dir(m, campos, dir_orig, dir);
            for (int c = 0; c < 3; c++) {
                dres[c] = dL_dcolor[3 * idx + c];
                dres[c] *= clamped[3 * idx + c] ? 0 : 1;
            }
            sh_bwd(D, dir, shs + (size_t)idx * M * 3, 3, dres, dL_dsh + (size_t)idx * M * 3, dL_ddir);
            dnormvdv3(dir_orig, dL_ddir, dmean);
            dL_dmeans[3 * idx + 0] += dmean[0];
            dL_dmeans[3 * idx + 1] += dmean[1];
            dL_dmeans[3 * idx + 2] += dmean[2];
        }

        float dist_to_light = dists[idx];
        if (shs_p != NULL) {
            float dL_dCW[2] = {0, 0};
            float phase = dist_to_light * dist2phase + phase_offset;
            if (cfg->use_view_dependent_phase)
                phase += phase_amp[idx * 2 + 0];
            float amplitude = phase_amp[idx * 2 + 1];
            float factor = 1.0f / (dist_to_light * dist_to_light);

            const float* g = dL_dphasor + (size_t)idx * 7;
            float dL_dR = g[0], dL_dI = g[1], dL_dA = g[2];
            float dL_dq1 = g[3], dL_dq2 = g[4], dL_dq3 = g[5], dL_dq4 = g[6];
            float sin_p = sinf(phase);
            float cos_p = cosf(phase);

            if (cfg->use_view_dependent_phase) {
                dL_dCW[0] = (dL_dR * -sin_p + dL_dI * cos_p +
                             dL_dq1 * -sin_p + dL_dq2 * sin_p + dL_dq3 * cos_p + dL_dq4 * -cos_p) *
                            amplitude * factor;
            }
            sum_phase += (double)((dL_dR * -sin_p + dL_dI * cos_p +
                                   dL_dq1 * -sin_p + dL_dq2 * sin_p + dL_dq3 * cos_p + dL_dq4 * -cos_p) *
                                  amplitude * factor);

            dL_dCW[1] = (dL_dR * cos_p + dL_dI * sin_p + dL_dA +
                         dL_dq1 * (cos_p + dc_offset) + dL_dq2 * (-cos_p + dc_offset) +
                         dL_dq3 * (sin_p + dc_offset) + dL_dq4 * (-sin_p + dc_offset)) *
                        factor;
            sum_dc += (double)((dL_dq1 + dL_dq2 + dL_dq3 + dL_dq4) * amplitude * factor);

            float coeff = (dL_dR * -sin_p + dL_dI * cos_p +
                           dL_dq1 * -sin_p + dL_dq2 * sin_p + dL_dq3 * cos_p + dL_dq4 * -cos_p) *
                              dist2phase * amplitude * factor / dist_to_light +
                          (dL_dR * -cos_p + dL_dI * -sin_p - dL_dA +
                           dL_dq1 * -(cos_p + dc_offset) + dL_dq2 * (cos_p - dc_offset) +
                           dL_dq3 * -(sin_p + dc_offset) + dL_dq4 * (sin_p - dc_offset)) *
                              2.0f * amplitude * factor * factor;
            float dxv = m_view[0] * coeff, dyv = m_view[1] * coeff, dzv = m_view[2] * coeff;
            float dLx = dxv * view[0] + dyv * view[1] + dzv * view[2];
            float dLy = dxv * view[4] + dyv * view[5] + dzv * view[6];
            float dLz = dxv * view[8] + dyv * view[9] + dzv * view[10];
            dL_dmeans[3 * idx + 0] += dLx;
            dL_dmeans[3 * idx + 1] += dLy;
            dL_dmeans[3 * idx + 2] += dLz;

            /* backward.cu:143-260 */
            float dir_orig[3], dir[3], dres[2], dL_ddir[3], dmean[3];
            view_dir(m, campos, dir_orig, dir);
            dres[0] = dL_dCW[0];
            dres[1] = dL_dCW[1];
            dres[1] *= clamped_p[idx] ? 0 : 1;
            sh_bwd(D, dir, shs_p + (size_t)idx * M_p * 2, 2, dres, dL_dsh_p + (size_t)idx * M_p * 2, dL_ddir);
            dnormvdv3(dir_orig, dL_ddir, dmean);
            dL_dmeans[3 * idx + 0] += dmean[0];
            dL_dmeans[3 * idx + 1] += dmean[1];
            dL_dmeans[3 * idx + 2] += dmean[2];
        }

        /* backward.cu:589-601 */
        float dndc_dist_ddist = (far_n * near_n) / ((far_n - near_n) * dist_to_light * dist_to_light);
        float dL_ddist = dL_ddist_ndc[idx] * dndc_dist_ddist + dL_ddist_in[idx];
        float dxv = dL_ddist * m_view[0] / dist_to_light;
        float dyv = dL_ddist * m_view[1] / dist_to_light;
        float dzv = dL_ddist * m_view[2] / dist_to_light;
        float dLx = dxv * view[0] + dyv * view[1] + dzv * view[2];
        float dLy = dxv * view[4] + dyv * view[5] + dzv * view[6];
        float dLz = dxv * view[8] + dyv * view[9] + dzv * view[10];
        dL_dmeans[3 * idx + 0] += dLx;
        dL_dmeans[3 * idx + 1] += dLy;
        dL_dmeans[3 * idx + 2] += dLz;

        if (scales != NULL)
            compute_cov3d_bwd(scales + 3 * idx, cfg->scale_modifier, rotations + 4 * idx,
                              dL_dcov + 6 * idx, dL_dscales + 3 * idx, dL_drots + 4 * idx);
    }
    *dL_dphase_offset += (float)sum_phase;
    *dL_ddc_offset += (float)sum_dc;
}
