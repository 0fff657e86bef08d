/* pair_stats.c -- statistics behind DESIGN.md's bound for k_render_bwd (not part of the library or of the oracle):
 * could (quadrant, splat) hits of the backward's walk share a pass?  Built and run by pair_stats.py. */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#define GFTO_BLOCK_X 16
#define GFTO_BLOCK_Y 16
static inline float fminf_(float a, float b) { return a < b ? a : b; }

/* How many (quadrant, splat) hits of the render backward's walk
 * could share a pass.  For every 8x8 quadrant the list is walked back to front from its deepest contributor as
 * k_render_bwd does; a hit = an entry some pixel of the quadrant blended, with the 64-bit mask of those pixels.  Hits
 * whose masks are disjoint touch no common pixel and commute.  out[0] = hits, out[1] = contributing lanes,
 * out[2] = passes when up to two CONSECUTIVE disjoint hits share a pass, out[3] = up to four consecutive,
 * out[4] = passes when a hit may also join one of the last `window` open passes it commutes with (pairs),
 * out[5] = the same with up to four hits per pass, out[6] = entries walked. */
void gfto_pair_stats(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* means2D,
                     const float* conic_opacity, const uint32_t* n_contrib, int window, double* out)
{
    const int gx = (W + GFTO_BLOCK_X - 1) / GFTO_BLOCK_X;
    const int gy = (H + GFTO_BLOCK_Y - 1) / GFTO_BLOCK_Y;
    double hits = 0, lanes = 0, p2 = 0, p4 = 0, w2 = 0, w4 = 0, walked = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : hits, lanes, p2, p4, w2, w4, walked)
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
        walked += tmax;
        uint64_t u2 = 0, u4 = 0; int n2 = 0, n4 = 0;
        uint64_t open2[16], open4[16]; int cnt2[16], cnt4[16]; int no2 = 0, no4 = 0;
        for (uint32_t c = tmax; c-- > 0;) {
            const uint32_t id = point_list[r0 + c];
            const float* co = conic_opacity + 4 * (size_t)id;
            uint64_t mask = 0;
            for (int l = 0; l < 64; l++) {
                const int px = qx0 + (l & 7), py = qy0 + (l >> 3);
                if (!(px < W && py < H) || c >= n_contrib[(size_t)W * py + px]) continue;
                const float dx = means2D[2 * id] - (float)px, dy = means2D[2 * id + 1] - (float)py;
                const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.0f) continue;
                const float alpha = fminf_(0.99f, co[3] * expf(power));
                if (alpha < 1.0f / 255.0f) continue;
                mask |= 1ull << l;
            }
            if (!mask) continue;
            hits += 1; lanes += __builtin_popcountll(mask);
            if (n2 > 0 && n2 < 2 && !(u2 & mask)) { u2 |= mask; n2++; } else { p2 += 1; u2 = mask; n2 = 1; }
            if (n4 > 0 && n4 < 4 && !(u4 & mask)) { u4 |= mask; n4++; } else { p4 += 1; u4 = mask; n4 = 1; }
            /* windowed: join the OLDEST open pass it commutes with such that every younger open pass is disjoint too */
            for (int v = 0; v < 2; v++) {
                uint64_t* op = v ? open4 : open2; int* cn = v ? cnt4 : cnt2; int* no = v ? &no4 : &no2; const int cap = v ? 4 : 2;
                int join = -1;
                for (int k = *no - 1; k >= 0; k--) {        /* youngest first */
                    if (op[k] & mask) break;                 /* cannot move in front of this pass */
                    if (cn[k] < cap) join = k;
                }
                if (join >= 0) { op[join] |= mask; cn[join]++; }
                else {
                    if (*no == window) { for (int k = 1; k < *no; k++) { op[k - 1] = op[k]; cn[k - 1] = cn[k]; } (*no)--; }
                    op[*no] = mask; cn[*no] = 1; (*no)++;
                    if (v) w4 += 1; else w2 += 1;
                }
            }
        }
    }
    out[0] = hits; out[1] = lanes; out[2] = p2; out[3] = p4; out[4] = w2; out[5] = w4; out[6] = walked;
}
