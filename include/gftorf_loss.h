/*
 * gftorf_loss.h -- C ABI of the image terms of the training loss (libgftorf_rast.so, gfx950).
 *
 * The reference drives every rasterizer backward with  (1 - lambda_dssim) * L_pixel + lambda_dssim * (1 - ssim(img, gt))
 * (train.py:209-231 for the ToF phasor planes, :196-206 for colour), ssim = utils/loss_utils.py:76-123: five grouped 11x11
 * convolutions (Gaussian window, sigma 1.5, zero padding) of img, gt, img^2, gt^2, img * gt, an elementwise map, its mean.
 * In eager PyTorch that is 8 convolution launches forward + backward and ~25 elementwise ones per iteration -- at the
 * reference's own image size (320x240) 0.55 ms of a 2.2 - 3.5 ms iteration, more than the two rasterizer calls together.
 * Here: one launch forward, one backward (separable window through LDS, one 16x16 tile per workgroup).
 *
 * Device pointers, fp32, images [C, H, W] contiguous; returns 0 on success (gft_last_error()).
 */
#ifndef GFTORF_LOSS_H
#define GFTORF_LOSS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GFT_SSIM_WINDOW 11
#define GFT_SSIM_TILE 16

/* workgroups of a launch = rows of `partials` */
int64_t gft_ssim_blocks(int32_t C, int32_t H, int32_t W);

/* partials[b] = {sum of ssim_map, sum of (img1 - img2)^2} over workgroup b's pixels (the caller adds the rows and divides by
 * C*H*W: utils/loss_utils.py:51-53 l2_loss, :119-120 ssim_map.mean()).  window: the 11 weights of the 1-D Gaussian (host;
 * loss_utils.py:76-78: the 2-D window is their outer product).  maps: 3*C*H*W floats kept for the backward -- per pixel
 * the derivatives of ssim_map by mu1 (with the sigma terms' dependence on mu1 folded in), sigma1^2 and sigma12 -- or NULL. */
int gft_ssim_l2_forward(void* hip_stream, int32_t C, int32_t H, int32_t W, const float* img1, const float* img2,
                        const float* window /*host, GFT_SSIM_WINDOW floats*/, float* maps, float* partials);

/* grad_img1 = g_ssim * d(sum ssim_map)/d img1 * scale_ssim + g_l2 * d(sum (img1 - img2)^2)/d img1 * scale_l2, where g_ssim and
 * g_l2 are read on the DEVICE (one float each: the upstream gradients of the two means; NULL = 0) and the scales are the
 * caller's 1 / (C*H*W).  Nothing is read by the host: capturable in a graph. */
int gft_ssim_l2_backward(void* hip_stream, int32_t C, int32_t H, int32_t W, const float* img1, const float* img2,
                         const float* window /*host*/, const float* maps, const float* g_ssim, const float* g_l2,
                         float scale_ssim, float scale_l2, float* grad_img1);

#ifdef __cplusplus
}
#endif
#endif
