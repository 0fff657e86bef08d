/* group_walk_stats.c -- the statistic behind round 6's sub-group walk of the blend kernels (not part of the library or of the
 * oracle; built and run by group_walk_stats.py).
 *
 * k_render_bwd / k_render_fwd walk a tile's list one 8x8 quadrant per wave, in batches of B staged entries; an entry that
 * some pixel of the quadrant blends costs the WHOLE wave one pass (evaluate + blend + reduce), whatever the number of
 * pixels it touches.  If the wave's 64 lanes are split into G groups of 64 / G pixels that each walk their OWN hits of the
 * batch, a batch costs max-over-groups(hits of the group) passes instead of hits-with-any-lane.  G = 64 is the
 * lane-private walk.  For every quadrant the list is walked back to front from its deepest contributor, as the backward
 * does, in batches of B in {64, 128, 256}; a (group, entry) hit is counted two ways:
 *    live  -- some pixel of the group blended the entry (the floor: what a perfect cull would leave),
 *    cull  -- the entry's alpha >= 1/255 ellipse reaches the group's rectangle of pixel centres (what the kernel can know
 *             before it has evaluated the pixels: the staging pass's test, gft_splat_reaches_box restated).
 * out layout: out[0] = quadrants with a walk, out[1] = walked entries, out[2] = wave hits (live), out[3] = wave hits (cull),
 * out[4] = live (pixel, entry) pairs; then for b in 0..2 (B = 64, 128, 256), for g in 0..NG-1: passes_live, passes_cull,
 * group_hits_cull (sum over groups: how many group records a pass-major walk evaluates). */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#define NG 6
static const int GROUPS[NG] = {1, 2, 4, 8, 16, 64};   /* groups per quadrant: 8x8, 8x4, 4x4, 4x2, 2x2, 1x1 pixels */
static const int GW[NG] = {8, 8, 4, 4, 2, 1};
static const int GH[NG] = {8, 4, 4, 2, 2, 1};

static inline float fminf_(float a, float b) { return a < b ? a : b; }
static inline float fmaxf_(float a, float b) { return a > b ? a : b; }

/* gft_internal.h: gft_splat_reaches_box, restated */
static int reaches_box(float mx, float my, float ca, float cb, float cc, float op, float bx0, float by0, float bw, float bh)
{
    const float det = ca * cc - cb * cb;
    const float tau = logf(255.0f * op);
    if (!(tau > 0.0f)) return 0;
    if (!(det > 0.0f && ca > 0.0f && cc > 0.0f)) return 1;
    const float ux0 = bx0 - mx, ux1 = ux0 + bw;
    const float uy0 = by0 - my, uy1 = uy0 + bh;
    const float X = fminf_(fmaxf_(0.0f, ux0), ux1), Y = fminf_(fmaxf_(0.0f, uy0), uy1);
    const float ys = fminf_(fmaxf_(-cb * X / cc, uy0), uy1);
    const float xs = fminf_(fmaxf_(-cb * Y / ca, ux0), ux1);
    const float q1 = ca * X * X + 2.0f * cb * X * ys + cc * ys * ys;
    const float q2 = ca * xs * xs + 2.0f * cb * xs * Y + cc * Y * Y;
    return fminf_(q1, q2) <= 2.0f * tau * 1.0005f + 0.01f;
}

void gfto_group_walk_stats(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* means2D,
                           const float* conic_opacity, const uint32_t* n_contrib, double* out)
{
    const int gx = (W + 15) / 16, gy = (H + 15) / 16;
    enum { NOUT = 5 + 3 * NG * 3 };
    double acc[NOUT];
    for (int i = 0; i < NOUT; i++) acc[i] = 0;
#pragma omp parallel
    {
        double loc[NOUT];
        for (int i = 0; i < NOUT; i++) loc[i] = 0;
#pragma omp for schedule(dynamic, 4)
        for (int unit = 0; unit < gx * gy * 4; unit++) {
            const int tile = unit >> 2, quad = unit & 3;
            const int tx = tile % gx, ty = tile / gx;
            const int qx0 = tx * 16 + (quad & 1) * 8, qy0 = ty * 16 + (quad >> 1) * 8;
            const uint32_t r0 = ranges[2 * tile];
            uint32_t tmax = 0;
            for (int l = 0; l < 64; l++) {
                const int px = qx0 + (l & 7), py = qy0 + (l >> 3);
                if (px < W && py < H && n_contrib[(size_t)W * py + px] > tmax) tmax = n_contrib[(size_t)W * py + px];
            }
            if (!tmax) continue;
            loc[0] += 1; loc[1] += tmax;
            /* per-batch counters: [b][g][group] for live and cull */
            int cl[3][NG][64], cc_[3][NG][64];
            for (int b = 0; b < 3; b++) for (int g = 0; g < NG; g++) for (int k = 0; k < GROUPS[g]; k++) { cl[b][g][k] = 0; cc_[b][g][k] = 0; }
            uint32_t walked_in[3] = {0, 0, 0};
            const int BS[3] = {64, 128, 256};
            for (uint32_t c = tmax; c-- > 0;) {
                const uint32_t id = point_list[r0 + c];
                const float* co = conic_opacity + 4 * (size_t)id;
                const float mx = means2D[2 * id], my = means2D[2 * id + 1];
                uint64_t mask = 0;
                for (int l = 0; l < 64; l++) {
                    const int px = qx0 + (l & 7), py = qy0 + (l >> 3);
                    if (!(px < W && py < H) || c >= n_contrib[(size_t)W * py + px]) continue;
                    const float dx = mx - (float)px, dy = my - (float)py;
                    const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.0f) continue;
                    const float alpha = fminf_(0.99f, co[3] * expf(power));
                    if (alpha < 1.0f / 255.0f) continue;
                    mask |= 1ull << l;
                }
                if (mask) { loc[2] += 1; loc[4] += __builtin_popcountll(mask); }
                const int wave_cull = reaches_box(mx, my, co[0], co[1], co[2], co[3], (float)qx0, (float)qy0, 7.f, 7.f);
                if (wave_cull) loc[3] += 1;
                for (int g = 0; g < NG; g++) {
                    const int gw = GW[g], gh = GH[g], per_row = 8 / gw;
                    for (int k = 0; k < GROUPS[g]; k++) {
                        const int x0 = (k % per_row) * gw, y0 = (k / per_row) * gh;
                        uint64_t gm = 0;
                        for (int yy = 0; yy < gh; yy++) gm |= ((uint64_t)((1u << gw) - 1u) << x0) << (8 * (y0 + yy));
                        const int live = (mask & gm) != 0;
                        /* (an entry the wave's cull drops is never looked at by a group) */
                        const int cull = wave_cull && (g == 0 ? 1 : reaches_box(mx, my, co[0], co[1], co[2], co[3], (float)(qx0 + x0), (float)(qy0 + y0), (float)(gw - 1), (float)(gh - 1)));
                        for (int b = 0; b < 3; b++) { cl[b][g][k] += live; cc_[b][g][k] += cull || live; }
                    }
                }
                for (int b = 0; b < 3; b++) {
                    walked_in[b]++;
                    if (walked_in[b] == (uint32_t)BS[b] || c == 0) {
                        for (int g = 0; g < NG; g++) {
                            int ml = 0, mc = 0, sc = 0;
                            for (int k = 0; k < GROUPS[g]; k++) {
                                if (cl[b][g][k] > ml) ml = cl[b][g][k];
                                if (cc_[b][g][k] > mc) mc = cc_[b][g][k];
                                sc += cc_[b][g][k];
                                cl[b][g][k] = 0; cc_[b][g][k] = 0;
                            }
                            loc[5 + (b * NG + g) * 3 + 0] += ml;
                            loc[5 + (b * NG + g) * 3 + 1] += mc;
                            loc[5 + (b * NG + g) * 3 + 2] += sc;
                        }
                        walked_in[b] = 0;
                    }
                }
            }
        }
#pragma omp critical
        for (int i = 0; i < NOUT; i++) acc[i] += loc[i];
    }
    for (int i = 0; i < NOUT; i++) out[i] = acc[i];
}
