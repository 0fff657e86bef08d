/*
 * gftorf_deform.h -- C ABI of the deformation network of libgftorf_rast.so (gfx950).
 *
 * `DeformNetwork` of the reference (utils/time_utils.py:56-127),
 * queried 1-4 times per training iteration for the dynamic Gaussians
 * (scene/gaussian_model.py:170-174, train.py:164-177): positional encoding of (xyz, t)
 * (xyz_multires and t_multires octaves: 10 and 10 = 84 inputs as the reference constructs it,
 * scene/deform_model.py:9-16 from arguments/__init__.py:66-69 and configs/{torf,ftorf}.json;
 * 10 and 6 = 76 inputs are the class defaults, time_utils.py:57), 8 x (Linear + ReLU) of width
 * 256 with the encoding concatenated in front of the activations after layer 4
 * (linear.5.weight is [256, in + 256]), heads `xyz_warp` (3) and
 * `r`/`g`/`b` (16 each).  The reference returns zeros for the rotation and phasor offsets
 * (time_utils.py:127) and never uses its `rot` / `a` heads; those are not computed here.
 *
 * This is the one GEMM-shaped piece of the path and it runs on the matrix cores with fp32 results: every fp32
 * operand is split exactly into three bf16 numbers (hi + mid + lo = 24 mantissa bits) and a product is the sum of
 * six `v_mfma_f32_32x32x16_bf16` terms (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid; the three dropped terms are
 * below 2^-23 of the product), accumulated in fp32: the error against a float64 product is that of an fp32 fma
 * chain (measured 5.4e-7 vs 5.5e-7 of the max-norm), at up to 2.7 x the rate of the fp32-operand MFMA.
 * GFT_DEFORM_BF16X3=0 in the environment selects `v_mfma_f32_32x32x2_f32` (fp32 operands) instead.
 * The forward walk goes one step further: its operands are split into two fp16 numbers, x = hi + lo (hi = fp16(x),
 * lo = fp16(x - hi): 22 mantissa bits), and a product is three `v_mfma_f32_32x32x16_f16` terms (hi*hi, hi*lo, lo*hi;
 * lo*lo is below 2^-22 of the product and dropped) into one fp32 accumulator.  fp16 has five exponent bits, so weights
 * are stored times 2^10 and activations times 2^4 (the accumulator is scaled back once per tile): a weight of 1e-4 .. 64
 * and an activation of 8e-3 .. 4094 keep all 22 bits, smaller ones have a subnormal lo that still resolves 4e-9 of an
 * activation (the matrix pipe honours fp16 subnormals).  Measured against float64 the outputs are as close as the
 * six-term bf16 walk's (1.7e-7 of the max-norm at 300 k points) at 0.75 x its time.  A weight, activation or input beyond
 * that range does not fit the planes: the pack kernel flags such weights, the walk notices such values, and the bf16 walk
 * (fp32 range), launched behind the fp16 one in every call, then redoes the call -- its workgroups return at once
 * otherwise.  GFT_DEFORM_FP16X2=0 runs the bf16 walk alone.
 * Since round 6 the backward's two heavy kernels multiply on two fp16 planes as well (GFT_DEFORM_BWD_FP16=0: three bf16
 * planes).  Gradients have no fixed range, so they carry power-of-two scales chosen from the data: the backward walk
 * scales every point's gradient row by its own (the accumulator of a point is a lane of the MFMA's output), the
 * weight-gradient kernel -- whose sums run over the points -- takes one per workgroup from the row maxima the walk leaves
 * in the scratch buffer; scaling by a power of two is exact, so any loss scale gives the same bits times that scale
 * (tests/test_deform.py::test_backward_carries_any_gradient_magnitude).  Activations go in times 2^4 as in the forward; one
 * beyond 4094 (or a weight beyond 64) hands the kernel's work to its bf16 twin launched behind it, as in the forward.
 * Against float64 the parameter gradients are at 2.3e-6 of the max-norm (bf16 kernels: 3.7e-6; numpy in fp32: 2.0e-6).
 *
 *   forward : one workgroup per 64 points walks all layers with the activations in LDS
 *             (weights streamed from L2), saving the post-ReLU activations for the backward
 *   backward: the same walk in reverse for the activation gradients, then the weight / bias
 *             gradients as point-split GEMMs (partial sums + one reduction: deterministic)
 *
 * Built for D = 8, W = 256, sh_degree = 3 (arguments/__init__.py:66-67, both shipped configs) and any
 * (xyz_multires, t_multires) whose encoding has at most GFT_DEFORM_MAX_INPUTS columns -- the
 * octave counts are run-time arguments of every call (gft_deform_inputs() gives the column count, -1
 * when it does not fit).  All pointers are device pointers to fp32; every function returns 0 on
 * success (gft_last_error()).  Inputs are not differentiated (the reference detaches them,
 * scene/gaussian_model.py:172).
 */
#ifndef GFTORF_DEFORM_H
#define GFTORF_DEFORM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GFT_DEFORM_LAYERS 8
#define GFT_DEFORM_WIDTH 256
#define GFT_DEFORM_MAX_INPUTS 96 /* encoded inputs the kernels hold: 3 + 6 xyz_multires + 1 + 2 t_multires <= 96 */
#define GFT_DEFORM_NUM_SHS 16

/* Parameters in torch's layout (`nn.Linear.weight` = [out, in] row-major), state_dict names in
 * the comments.  With in = gft_deform_inputs(xyz_multires, t_multires): linear_w[0] is [256,in]
 * ([256,84] for the reference's configs), linear_w[5] is [256,in+256] (encoding first; [256,340]),
 * the others [256,256]. */
typedef struct gft_deform_params {
    const float* linear_w[GFT_DEFORM_LAYERS];   /* linear.{i}.weight */
    const float* linear_b[GFT_DEFORM_LAYERS];   /* linear.{i}.bias   [256] */
    const float* xyz_w;                         /* xyz_warp.weight   [3,256] */
    const float* xyz_b;                         /* xyz_warp.bias     [3] */
    const float* r_w; const float* r_b;         /* r.weight [16,256], r.bias [16] */
    const float* g_w; const float* g_b;
    const float* b_w; const float* b_b;
} gft_deform_params;

/* Gradients, same shapes; every tensor is written in full (nothing has to be zeroed). */
typedef struct gft_deform_grads {
    float* linear_w[GFT_DEFORM_LAYERS];
    float* linear_b[GFT_DEFORM_LAYERS];
    float* xyz_w; float* xyz_b;
    float* r_w; float* r_b;
    float* g_w; float* g_b;
    float* b_w; float* b_b;
} gft_deform_grads;

/* 3 + 6 xyz_multires + 1 + 2 t_multires (time_utils.py:8-53), or -1 when that exceeds
 * GFT_DEFORM_MAX_INPUTS or an octave count is negative. */
int gft_deform_inputs(int xyz_multires, int t_multires);

/* Sizes of the caller-owned buffers (the same for every supported encoding). */
size_t gft_deform_packed_bytes(void);            /* weights re-laid for the kernels */
size_t gft_deform_saved_bytes(int64_t n);        /* forward -> backward hand-off (encoding + 8 activations) */
size_t gft_deform_scratch_bytes(int64_t n);      /* backward scratch */

/* Re-lays the parameters for the matrix-core kernels (k-interleaved, both directions, biases).
 * Call after every parameter update, before forward / backward. */
int gft_deform_pack(void* hip_stream, int xyz_multires, int t_multires, const gft_deform_params* params, void* packed);

/* d_xyz[n,3], d_sh[n,16,3] for xyz[n,3] and t (t_stride = 1: one value per point, [n,1];
 * t_stride = 0: one value for all points, as scene/gaussian_model.py:171 expands it).
 * saved = NULL: inference, nothing is kept for a backward. */
int gft_deform_forward(void* hip_stream, int xyz_multires, int t_multires, int64_t n, const float* xyz, const float* t,
                       int64_t t_stride, const void* packed, void* saved, float* d_xyz, float* d_sh);

/* Gradients of the parameters for upstream gradients g_d_xyz[n,3] and g_d_sh[n,16,3]
 * (either may be NULL = zeros).  `packed` and `saved` as the forward left them. */
int gft_deform_backward(void* hip_stream, int xyz_multires, int t_multires, int64_t n, const void* packed, const void* saved,
                        const float* g_d_xyz, const float* g_d_sh, void* scratch, const gft_deform_grads* grads);

/* The backward over the rows that count: a point whose upstream gradient row (g_d_xyz, g_d_sh) is all zero has dz = 0
 * in every layer and adds nothing to any parameter gradient -- in a training iteration that is every queried point no
 * pixel blended (86-94 % on the metric scene).  Compacts the saved activations and the upstream gradients of the k
 * selected rows (mask[n] != 0, rank[row] = output row of a selected row: gft_rows_rank of gftorf_densify.h) into
 * `saved_c` (gft_deform_saved_bytes(k); padded rows zeroed) and g_d_xyz_c[k,3] / g_d_sh_c[k,16,3];
 * gft_deform_backward(n = k, saved_c, ...) then gives the gradients of the dense call up to summation order.
 * idx: scratch of k int32. */
int gft_deform_compact(void* hip_stream, int64_t n, int64_t k, const uint8_t* mask, const int32_t* rank, const void* saved,
                       const float* g_d_xyz, const float* g_d_sh, int32_t* idx, void* saved_c, float* g_d_xyz_c,
                       float* g_d_sh_c);

/* The backward over the rows that count with NOTHING read back from the device: the rows whose upstream gradient row
 * (g_d_xyz, g_d_sh) holds a value != 0 are marked, ranked and counted by kernels, their inputs and gradients gathered
 * in row order, their activations computed by a saving forward over them alone (a point's activations do not depend on
 * the batch it is in) and the backward run over them -- every launch sized for n, the capacity, with the extents taken from
 * a plan a one-thread kernel writes from the count (surplus workgroups return at once).  The call neither blocks nor
 * allocates, so it can be captured in a HIP graph and replayed on other gradients: the replay adapts to THEIR rows.
 * For a forward that kept nothing (gft_deform_forward with saved = NULL): xyz [n,3], t and t_stride as the forward had
 * them.  Gradients equal gft_deform_compact + gft_deform_backward over the same rows bit for bit (the same kernels on
 * the same compacted rows; only the buffers' plane stride differs), i.e. the dense backward's up to summation order.
 * work: gft_deform_rows_work_bytes(n) bytes, 256-byte aligned, contents irrelevant before and after.
 * rows_out: device uint32 that receives the number of rows processed (may be NULL). */
size_t gft_deform_rows_work_bytes(int64_t n);
int gft_deform_backward_rows(void* hip_stream, int xyz_multires, int t_multires, int64_t n, const void* packed,
                             const float* xyz, const float* t, int64_t t_stride, const float* g_d_xyz, const float* g_d_sh,
                             void* work, const gft_deform_grads* grads, uint32_t* rows_out);

#ifdef __cplusplus
}
#endif
#endif
