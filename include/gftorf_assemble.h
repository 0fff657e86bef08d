/*
 * gftorf_assemble.h -- C ABI of the fused input assembly of libgftorf_rast.so (gfx950).
 *
 * SURVEY section 8(f) row 1.  Replaces the eager-PyTorch glue that builds the rasterizer's
 * seven per-Gaussian inputs before every render in the reference
 * (gaussian_renderer/__init__.py:81-105 of brownvc/gftorf):
 *
 *     means3D = zeros; means2D = zeros; opacity = zeros; ...            (:81-87)
 *     if "static"  in render_regions: X[~motion_mask] = src[~motion_mask]          (:89-96)
 *     if "dynamic" in render_regions: means3D[motion_mask] = xyz[motion_mask] + d_xyz
 *                                     rotations[motion_mask] = normalize(_rotation[motion_mask] + d_rot)
 *                                     shs[motion_mask] = features_color[motion_mask] + d_sh ...  (:97-104)
 *
 * and its autograd backward.  The d_* offsets are the deformation network's outputs for the
 * dynamic Gaussians, one row per True of motion_mask in order (scene/gaussian_model.py:170-174),
 * or the Python float 0.0 when there is no deformation (train.py:164).
 *
 * Plain device pointers and sizes, no torch types; every entry point returns 0 on success
 * (message from gft_last_error()).  The caller owns all memory; outputs and gradients are
 * written in full, so they can be allocated uninitialised.
 */
#ifndef GFTORF_ASSEMBLE_H
#define GFTORF_ASSEMBLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gft_assemble_io {
    /* per-Gaussian sources (what the reference reads through pc.get_*), all fp32 contiguous */
    const float* xyz;            /* [P,3]   pc.get_xyz */
    const float* screenspace;    /* [P,3]   screenspace_points (zeros; gradient sink) */
    const float* opacity;        /* [P,1]   pc.get_opacity   (activated) */
    const float* scaling;        /* [P,3]   pc.get_scaling   (activated) */
    const float* rotation;       /* [P,4]   pc.get_rotation  (normalised; used for static rows), or NULL: the static rows
                                  *          are normalize(rotation_raw) computed here -- no [P,4] tensor and no eager
                                  *          normalize / backward (eight launches) for the caller */
    const float* rotation_raw;   /* [P,4]   pc._rotation     (used for dynamic rows) */
    const float* feat_color;     /* [P,M,3] pc.get_features_color,  NULL iff M == 0 */
    const float* feat_phasor;    /* [P,M_p,2] pc.get_features_phasor, NULL iff M_p == 0 */
    const uint8_t* motion_mask;  /* [P]     torch.bool */
    /* offsets of the dynamic rows: [Nd,3], [Nd,4], [Nd,M,3], [Nd,M_p,2]; NULL = use the scalar */
    const float* d_xyz;
    const float* d_rot;
    const float* d_sh;
    const float* d_sh_p;
    float d_xyz_scalar, d_rot_scalar, d_sh_scalar, d_sh_p_scalar;
    /* scratch of gft_assemble_scratch_bytes(P): the rank of every dynamic row (forward ->
     * backward hand-off) and the block sums of the rank scan */
    void* scratch;
    /* outputs = the rasterizer's inputs */
    float* out_means3D;          /* [P,3] */
    float* out_means2D;          /* [P,3] */
    float* out_opacity;          /* [P,1] */
    float* out_scales;           /* [P,3] */
    float* out_rotations;        /* [P,4] */
    float* out_shs;              /* [P,M,3]; NULL = not wanted (a scalar offset of 0 with both regions rendered: the */
    float* out_shs_p;            /* [P,M_p,2]  output would be the feature tensor, which the caller then uses itself)  */
    /* Nd: rows of the d_* tensors that are given (all share it; 0 <= Nd <= P; ignored when all four are scalars).
     * A dynamic Gaussian whose rank among the dynamic ones is >= Nd has no offset row -- the reference's masked
     * assignment raises for such shapes -- : nothing is read or written out of bounds, that Gaussian's outputs are
     * NaN and its gradients zero.  gft_assemble_num_dynamic() gives the exact count for a host-side check. */
    int64_t num_offset_rows;
    /* ---- (round 6) the model's own tensors as sources: what pc.get_* computes from them before render() reads it
     * (scene/gaussian_model.py:123-153) is then done here, forward and backward -- no eager exp / sigmoid / cat and no
     * activated copies for the caller.  All zero / NULL: the fields above hold the activated tensors, as before. ---- */
    int32_t opacity_is_raw;        /* 1: `opacity` holds pc._opacity; used: sigmoid(opacity) = 1 / (1 + exp(-x)) */
    int32_t scaling_is_raw;        /* 1: `scaling` holds pc._scaling; used: exp(scaling) */
    const float* feat_dc_color;    /* [P,1,3]     } with feat_color == NULL and M > 0: pc.get_features_color =           */
    const float* feat_rest_color;  /* [P,M-1,3]   }   cat((dc, rest), dim=1), read from its two parts                     */
    const float* phase_dc;         /* [P,1,1]     } with feat_phasor == NULL and M_p > 0: pc.get_features_phasor =        */
    const float* phase_rest;       /* [P,M_p-1,1] }   cat((cat((phase_dc, phase_rest), 1), cat((amp_dc, amp_rest), 1)), 2) */
    const float* amp_dc;           /* [P,1,1]     }   read from its four parts                                            */
    const float* amp_rest;         /* [P,M_p-1,1] }                                                                       */
} gft_assemble_io;

typedef struct gft_assemble_bwd_io {
    const void* scratch;         /* as written by the forward */
    const float* rotation_raw;   /* [P,4] */
    const float* d_rot;          /* [Nd,4] or NULL (scalar) */
    float d_rot_scalar;
    /* upstream gradients of the seven outputs; NULL = zeros */
    const float* g_means3D;
    const float* g_means2D;
    const float* g_opacity;
    const float* g_scales;
    const float* g_rotations;
    const float* g_shs;
    const float* g_shs_p;
    /* gradients of the sources, written in full; NULL = not wanted */
    float* g_xyz;                /* [P,3] */
    float* g_screenspace;        /* [P,3] */
    float* g_opacity_in;         /* [P,1] */
    float* g_scaling;            /* [P,3] */
    float* g_rotation;           /* [P,4]  static rows, zeros elsewhere */
    float* g_rotation_raw;       /* [P,4]  dynamic rows (through the normalisation), zeros elsewhere */
    /* shs = features (+ d_sh on the dynamic rows): with both regions rendered the features' gradient is g_shs itself, row
     * for row.  A caller that hands g_shs on as that gradient passes NULL here and saves the copy (320 bytes per Gaussian
     * read and written at M = 16); only the dynamic rows are then read, for g_d_sh / g_d_sh_p. */
    float* g_feat_color;         /* [P,M,3] */
    float* g_feat_phasor;        /* [P,M_p,2] */
    float* g_d_xyz;              /* [Nd,3] */
    float* g_d_rot;              /* [Nd,4] */
    float* g_d_sh;               /* [Nd,M,3] */
    float* g_d_sh_p;             /* [Nd,M_p,2] */
    /* 1 = the forward ran with io.rotation == NULL: the static rows' gradient goes through the normalisation into
     * g_rotation_raw as well (g_rotation, if given, is zeros) */
    int32_t static_from_raw;
    /* ---- (round 6) a forward over the model's own tensors: the gradients go back through the activations ---- */
    const float* opacity_raw;      /* [P,1] pc._opacity (the forward ran with opacity_is_raw): g_opacity_in = g sigmoid'(x); NULL: plain */
    const float* scaling_raw;      /* [P,3] pc._scaling (scaling_is_raw): g_scaling = g exp(x); NULL: plain */
    float* g_feat_dc_color;        /* [P,1,3]     } the parts' gradients (give them instead of g_feat_color;   */
    float* g_feat_rest_color;      /* [P,M-1,3]   }  NULL = not wanted)                                         */
    float* g_phase_dc;             /* [P,1,1]     } (instead of g_feat_phasor)                                  */
    float* g_phase_rest;           /* [P,M_p-1,1] }                                                             */
    float* g_amp_dc;               /* [P,1,1]     }                                                             */
    float* g_amp_rest;             /* [P,M_p-1,1] }                                                             */
} gft_assemble_bwd_io;

size_t gft_assemble_scratch_bytes(int32_t P);

/* render_static / render_dynamic: "static" / "dynamic" in render_regions */
int gft_assemble_forward(void* hip_stream, int32_t P, int32_t M, int32_t M_p, int32_t render_static,
                         int32_t render_dynamic, const gft_assemble_io* io);

/* number of True entries of motion_mask as counted by the forward (blocking read; the
 * reference's masked assignment raises when d_* has another row count) */
int gft_assemble_num_dynamic(void* hip_stream, int32_t P, const void* scratch, int64_t* num_dynamic /*host*/);

int gft_assemble_backward(void* hip_stream, int32_t P, int32_t M, int32_t M_p, int32_t render_static,
                          int32_t render_dynamic, const gft_assemble_bwd_io* io);

#ifdef __cplusplus
}
#endif
#endif
