/*
 * gftorf_densify.h -- C ABI of the per-Gaussian bookkeeping kernels of libgftorf_rast.so (gfx950).
 *
 * SURVEY section 8(f) row 4 (bookkeeping part; the optimizer part is gftorf_optim.h): the tensor
 * surgery around the rasterizer in the reference's training loop, which at 1 M Gaussians costs
 * more than a rasterizer step in eager PyTorch because every boolean-mask index is a
 * nonzero() with a host synchronisation plus a gather or a scatter:
 *
 *   every iteration (train.py:441-449, scene/gaussian_model.py:648-654):
 *       max_radii2D[vis] = max(max_radii2D[vis], radii[vis])
 *       xyz_gradient_accum[vis] += ||viewspace_grad[vis, :2]|| * pixels[vis]
 *       denom[vis] += pixels[vis]
 *     -> gft_densify_stats: one pass, in place.
 *
 *   every densification / pruning step (scene/gaussian_model.py:473-514, 571-631): `t[mask]` of
 *   the 11 parameter tensors, their two Adam moments and 3 statistics tensors
 *     -> gft_rows_rank once per mask (row -> output row, number of kept rows), then
 *        gft_rows_gather per tensor: order-preserving compaction, pure byte movement
 *        (bit-identical to `t[mask]`).
 *
 * Device pointers; every function returns 0 on success (gft_last_error()).
 */
#ifndef GFTORF_DENSIFY_H
#define GFTORF_DENSIFY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* In-place statistics update for the rows where update_filter (and apply_mask, when given: the
 * reference's `apply_mask` branch, valid there only when update_filter is a subset of it) is set.
 * viewspace_grad [P,3] (xy = pixel-space gradient of means2D), pixels [P] or [P,1], radii int32 [P];
 * update_filter / apply_mask: one byte per Gaussian (torch.bool), apply_mask may be NULL;
 * xyz_gradient_accum [P,1], denom [P,1], max_radii2D [P] (any of the three may be NULL = skip). */
int gft_densify_stats(void* hip_stream, int64_t P, const float* viewspace_grad, const float* pixels, const int32_t* radii,
                      const uint8_t* update_filter, const uint8_t* apply_mask, float* xyz_gradient_accum, float* denom,
                      float* max_radii2D);

/* rank[i] = number of set mask bytes before row i (int32 [P]); *count (host) = number of set bytes.
 * Blocks until the count has reached the host (the reference's `t[mask]` blocks the same way).
 * scratch: gft_rows_rank_scratch_bytes(P) bytes. */
size_t gft_rows_rank_scratch_bytes(int64_t P);
int gft_rows_rank(void* hip_stream, int64_t P, const uint8_t* mask, int32_t* rank, void* scratch, int64_t* count);
/* The same ranking with the count left on the device (*count_dev, a uint32) and nothing read back: the call does not
 * block and can be captured in a HIP graph.  P > 0. */
int gft_rows_rank_dev(void* hip_stream, int64_t P, const uint8_t* mask, int32_t* rank, void* scratch, uint32_t* count_dev);

/* mask[i] = 1 if row i of a ([P, row_floats_a]) or of b ([P, row_floats_b]) holds a value != 0 (a NaN counts, -0 does
 * not), else 0; either tensor may be NULL.  The rows of a backward that have an upstream gradient -- what
 * `~((a.abs().amax(1).maximum(b.abs().amax(1))) == 0)` is in eight eager launches and two full-size temporaries. */
int gft_rows_any_nonzero(void* hip_stream, int64_t P, int32_t row_floats_a, const float* a, int32_t row_floats_b,
                         const float* b, uint8_t* mask);

/* dst[rank[i]] = src[i] for every row i with mask[i] != 0; rows are row_bytes long (a multiple of 4),
 * src and dst 4-byte aligned (16-byte accesses are used when rows and both pointers allow). */
int gft_rows_gather(void* hip_stream, int64_t P, const uint8_t* mask, const int32_t* rank, const void* src, void* dst,
                    int64_t row_bytes);

#ifdef __cplusplus
}
#endif
#endif
