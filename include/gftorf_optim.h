/*
 * gftorf_optim.h -- C ABI of the fused Adam step of libgftorf_rast.so (gfx950).
 *
 * SURVEY section 8(f) row 4 (optimizer part).  The reference updates its ~91 floats per Gaussian
 * with `torch.optim.Adam(l, lr=0.0, eps=1e-15)` (scene/gaussian_model.py:274, stepped at
 * train.py:470), whose default multi-tensor path runs one elementwise kernel per arithmetic
 * operation.  This is the same update (torch/optim/adam.py `_single_tensor_adam`, no amsgrad, no
 * maximize) in one pass: 16 B read + 12 B written per element.
 *
 *   g' = g + weight_decay * p
 *   m  = m + (1 - beta1) * (g' - m)                       (lerp)
 *   v  = beta2 * v + (1 - beta2) * g' * g'
 *   p  = p - (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
 *
 * Device pointers, fp32; returns 0 on success (gft_last_error()).
 */
#ifndef GFTORF_OPTIM_H
#define GFTORF_OPTIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Hyper-parameters as torch holds them (Python floats = double); `step` = t, the 1-based count of
 * this update.  Derived factors (1 - beta, lr / (1 - beta1^t), sqrt(1 - beta2^t)) are formed in
 * double precision and rounded to fp32 once, as torch does. */
int gft_adam_step(void* hip_stream, int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                  double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step);

/* The same update for several tensors in one launch per GFT_ADAM_MAX_TENSORS tensors: every tensor has its own
 * learning rate and step count (the reference keeps one tensor per parameter group, each with its own lr,
 * scene/gaussian_model.py:247-272), betas / eps / weight decay are shared. */
#define GFT_ADAM_MAX_TENSORS 40
typedef struct gft_adam_tensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t n;
    double lr;
    int64_t step;      /* 1-based count of this update */
} gft_adam_tensor;

int gft_adam_step_multi(void* hip_stream, int32_t count, const gft_adam_tensor* tensors /*host*/, double beta1,
                        double beta2, double eps, double weight_decay);

/* Opt-in, NOT the reference's optimizer (which is dense): the same update restricted to the rows of a mask -- SURVEY
 * section 8(f) row 4, "sparse Adam on visible Gaussians".  Every tensor has `rows` rows (n = rows x floats per row);
 * row r takes the step when row_mask[r] != 0 (device pointer, one byte per row: e.g. `radii > 0` of the iteration's
 * render, reference train.py:181 `visibility_filter`); the other rows' parameter and moments are left as they are
 * (their moments do not decay).  16-byte groups without a masked-in element are neither read nor written. */
int gft_adam_step_rows(void* hip_stream, int32_t count, const gft_adam_tensor* tensors /*host*/, int64_t rows,
                       const uint8_t* row_mask, double beta1, double beta2, double eps, double weight_decay);

/* The same launch with the learning rates and step counts ON THE DEVICE: nothing the update depends on is baked into the
 * call, so it can be captured in a HIP graph (torch.cuda.graphs around a whole training iteration) and replayed while the
 * schedule moves the learning rates.  tensors[c].lr / .step are ignored; lr[c] (device, double: the Python float the
 * reference's scheduler computes, scene/gaussian_model.py:294-310) and step[c] (device, fp32 count of updates DONE, as
 * torch's capturable Adam keeps it) are read by a one-workgroup kernel in front of the update, which adds 1 to every
 * step[c] and derives lr / (1 - beta1^t) and sqrt(1 - beta2^t) in double precision, rounded to fp32 once -- the factors
 * gft_adam_step_multi forms on the host, to the rounding of the device's double pow -- into factors[2 c], factors[2 c + 1]
 * (device scratch, 2 * count floats).
 * lr and step are given per tensor (arrays of device pointers, host) because optimizer state is edited tensor by tensor
 * (scene/gaussian_model.py:456-540). */
int gft_adam_step_multi_dev(void* hip_stream, int32_t count, const gft_adam_tensor* tensors /*host*/,
                            const double* const* lr /*host array of device pointers*/, float* const* step /*host array of device pointers*/,
                            float* factors /*device, 2 * count floats*/, double beta1, double beta2, double eps, double weight_decay);

#ifdef __cplusplus
}
#endif
#endif
