"""ctypes binding of libgftorf_rast.so (include/gftorf_rast.h).

The product path has no CPU fallback: if the HIP library is missing this module
raises, it never routes anywhere else.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgftorf_rast.so")
ABI_VERSION = 14
DEFORM_MAX_INPUTS = 96          # GFT_DEFORM_MAX_INPUTS (include/gftorf_deform.h)
ACC_STRIDE = 16

_lib = None

_fp = C.c_void_p  # every tensor pointer travels as void*


class Config(C.Structure):
    _fields_ = [
        ("P", C.c_int32), ("D", C.c_int32), ("M", C.c_int32), ("M_p", C.c_int32),
        ("W", C.c_int32), ("H", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("near_n", C.c_float), ("far_n", C.c_float), ("depth_range", C.c_float),
        ("phase_offset", C.c_float), ("dc_offset", C.c_float),
        ("use_view_dependent_phase", C.c_int32), ("prefiltered", C.c_int32), ("debug", C.c_int32),
        ("want_backward", C.c_int32),
        ("acc_zeroed", C.c_int32),
        ("grads_zeroed", C.c_int32),
        ("grads_accumulate", C.c_int32),
        ("bg_stride_c", C.c_int64), ("bg_stride_y", C.c_int64), ("bg_stride_x", C.c_int64),
    ]


FORWARD_FIELDS = [
    "bg", "means3D", "colors_precomp", "phasors_precomp", "opacities", "scales", "rotations",
    "cov3D_precomp", "viewmatrix", "projmatrix", "campos", "shs", "shs_p",
    "geom", "img", "binning",
    "out_color", "out_phasor", "out_depth", "out_normal", "out_acc", "out_entropy",
    "out_depth_distortion", "out_amp_distortion", "pixels", "out_distribution", "radii", "acc",
]

BACKWARD_FIELDS = [
    "bg", "means3D", "radii", "scales", "rotations", "cov3D_precomp", "viewmatrix", "projmatrix",
    "campos", "shs", "shs_p", "opacities", "pixels",
    "dL_dout_color", "dL_dout_phasor", "dL_dout_depth", "dL_dout_acc", "dL_dout_depth_distortion",
    "geom", "img", "binning", "acc",
    "dL_dmeans3D", "dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dcov3D", "dL_dsh", "dL_dsh_p",
    "dL_dscales", "dL_drotations", "dL_dphase_offset", "dL_ddc_offset", "det_partials", "dirty_rows", "rows_report",
]

LAYOUT_FIELDS = [
    "geom_rec_a", "geom_rec_b", "geom_depth", "geom_tiles", "geom_rect", "geom_dirgrad", "geom_clamped", "geom_need",
    "geom_blockhist", "geom_total",
    "img_pix_state", "img_ranges", "img_tile_max", "img_ctrl", "img_tile_cnt", "img_tile_cut", "img_super_tab", "img_tile_cursor",
    "img_tile_order", "img_front_len", "img_unit_flag", "img_resume_state", "img_pix_sums", "img_snaps", "img_total",
    "bin_keys", "bin_point_list", "bin_total",
]

PROFILE_FIELDS = ["preprocess_fwd_ms", "tile_count_ms", "tile_scatter_ms", "tile_sort_ms",
                  "render_fwd_ms", "render_bwd_ms", "preprocess_bwd_ms", "memset_ms"]


class ForwardIO(C.Structure):
    _fields_ = [(n, _fp) for n in FORWARD_FIELDS] + [("grads_zero", _fp), ("grads_zero_bytes", C.c_size_t), ("tile_hints", _fp), ("tile_weights", _fp),
                                                           ("cell_sched", _fp)]


class BackwardIO(C.Structure):
    _fields_ = [(n, _fp) for n in BACKWARD_FIELDS]


class Layout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in LAYOUT_FIELDS]


class ForwardHints(C.Structure):
    """gft_forward_hints"""
    _fields_ = [("binning_instances", C.c_int64), ("max_tile_list", C.c_int64), ("whole_lists", C.c_int64),
                ("use_cell_sched", C.c_int64)]


class ForwardReport(C.Structure):
    """gft_forward_report"""
    _fields_ = [("num_rendered", C.c_int64), ("max_tile_list", C.c_int64), ("list_entries", C.c_int64), ("hinted_tiles", C.c_int64),
                ("sched_misses", C.c_int64)]


class Profile(C.Structure):
    _fields_ = [(n, C.c_double) for n in PROFILE_FIELDS] + [("forward_calls", C.c_int64),
                                                           ("backward_calls", C.c_int64)]


# ---- fused input assembly (include/gftorf_assemble.h) -------------------------------------------
ASSEMBLE_PTRS_IN = ["xyz", "screenspace", "opacity", "scaling", "rotation", "rotation_raw", "feat_color",
                    "feat_phasor", "motion_mask", "d_xyz", "d_rot", "d_sh", "d_sh_p"]
ASSEMBLE_SCALARS = ["d_xyz_scalar", "d_rot_scalar", "d_sh_scalar", "d_sh_p_scalar"]
ASSEMBLE_PTRS_OUT = ["scratch", "out_means3D", "out_means2D", "out_opacity", "out_scales", "out_rotations",
                     "out_shs", "out_shs_p"]
ASSEMBLE_RAW_FLAGS = ["opacity_is_raw", "scaling_is_raw"]          # (round 6: the model's own tensors as sources)
ASSEMBLE_PARTS = ["feat_dc_color", "feat_rest_color", "phase_dc", "phase_rest", "amp_dc", "amp_rest"]
ASSEMBLE_FIELDS = ASSEMBLE_PTRS_IN + ASSEMBLE_SCALARS + ASSEMBLE_PTRS_OUT + ["num_offset_rows"] + ASSEMBLE_RAW_FLAGS + ASSEMBLE_PARTS
ASSEMBLE_BWD_HEAD = ["scratch", "rotation_raw", "d_rot"]
ASSEMBLE_BWD_TAIL = ["g_means3D", "g_means2D", "g_opacity", "g_scales", "g_rotations", "g_shs", "g_shs_p",
                     "g_xyz", "g_screenspace", "g_opacity_in", "g_scaling", "g_rotation", "g_rotation_raw",
                     "g_feat_color", "g_feat_phasor", "g_d_xyz", "g_d_rot", "g_d_sh", "g_d_sh_p"]
ASSEMBLE_BWD_RAW = ["opacity_raw", "scaling_raw", "g_feat_dc_color", "g_feat_rest_color", "g_phase_dc", "g_phase_rest", "g_amp_dc", "g_amp_rest"]
ASSEMBLE_BWD_FIELDS = ASSEMBLE_BWD_HEAD + ["d_rot_scalar"] + ASSEMBLE_BWD_TAIL + ["static_from_raw"] + ASSEMBLE_BWD_RAW


class AssembleIO(C.Structure):
    _fields_ = ([(n, _fp) for n in ASSEMBLE_PTRS_IN] + [(n, C.c_float) for n in ASSEMBLE_SCALARS] +
                [(n, _fp) for n in ASSEMBLE_PTRS_OUT] + [("num_offset_rows", C.c_int64)] +
                [(n, C.c_int32) for n in ASSEMBLE_RAW_FLAGS] + [(n, _fp) for n in ASSEMBLE_PARTS])


class AssembleBwdIO(C.Structure):
    _fields_ = ([(n, _fp) for n in ASSEMBLE_BWD_HEAD] + [("d_rot_scalar", C.c_float)] +
                [(n, _fp) for n in ASSEMBLE_BWD_TAIL] + [("static_from_raw", C.c_int32)] + [(n, _fp) for n in ASSEMBLE_BWD_RAW])


class DeformParams(C.Structure):
    """gft_deform_params / gft_deform_grads (include/gftorf_deform.h): same field order."""
    _fields_ = [("linear_w", _fp * 8), ("linear_b", _fp * 8), ("xyz_w", _fp), ("xyz_b", _fp), ("r_w", _fp), ("r_b", _fp),
                ("g_w", _fp), ("g_b", _fp), ("b_w", _fp), ("b_b", _fp)]


DEFORM_FIELDS = [f[0] for f in DeformParams._fields_]


class AdamTensor(C.Structure):
    """gft_adam_tensor (include/gftorf_optim.h)"""
    _fields_ = [("param", _fp), ("grad", _fp), ("exp_avg", _fp), ("exp_avg_sq", _fp), ("n", C.c_int64), ("lr", C.c_double),
                ("step", C.c_int64)]

EXPORTS = [
    "gft_abi_version", "gft_lazy_sort", "gft_last_error", "gft_geom_bytes", "gft_image_bytes", "gft_cell_sched_words", "gft_binning_bytes", "gft_acc_bytes",
    "gft_det_partials_bytes", "gft_get_layout", "gft_binning_capacity", "gft_set_binning_mode", "gft_binning_mode", "gft_set_render_mode", "gft_forward_preprocess", "gft_forward_render", "gft_forward", "gft_forward_enqueue", "gft_backward", "gft_grads_rezero",
    "gft_mark_visible", "gft_profile_enable", "gft_profile_reset", "gft_profile_read",
    "gft_assemble_scratch_bytes", "gft_assemble_forward", "gft_assemble_num_dynamic", "gft_assemble_backward",
    "gft_knn_scratch_bytes", "gft_knn_mean_dist2", "gft_adam_step", "gft_adam_step_multi", "gft_adam_step_rows", "gft_adam_step_multi_dev",
    "gft_deform_inputs", "gft_deform_packed_bytes", "gft_deform_saved_bytes", "gft_deform_scratch_bytes", "gft_deform_pack",
    "gft_deform_forward", "gft_deform_backward", "gft_deform_compact", "gft_deform_rows_work_bytes", "gft_deform_backward_rows",
    "gft_ssim_blocks", "gft_ssim_l2_forward", "gft_ssim_l2_backward",
    "gft_densify_stats", "gft_rows_rank_scratch_bytes", "gft_rows_rank", "gft_rows_rank_dev", "gft_rows_gather", "gft_rows_any_nonzero",
]


def load():
    """dlopen libgftorf_rast.so and declare prototypes.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "gftorf_amd: %s is missing. Build it with `python -m gftorf_amd.build` "
            "(hipcc, gfx950). There is no CPU fallback for the rasterizer." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.gft_abi_version.restype = C.c_int
    if lib.gft_abi_version() != ABI_VERSION:
        raise RuntimeError("gftorf_amd: libgftorf_rast.so ABI %d != expected %d, rebuild"
                           % (lib.gft_abi_version(), ABI_VERSION))
    lib.gft_last_error.restype = C.c_char_p
    lib.gft_geom_bytes.restype = C.c_size_t
    lib.gft_geom_bytes.argtypes = [C.c_int32]
    lib.gft_image_bytes.restype = C.c_size_t
    lib.gft_image_bytes.argtypes = [C.c_int32, C.c_int32]
    lib.gft_cell_sched_words.restype = C.c_size_t
    lib.gft_cell_sched_words.argtypes = [C.c_int32, C.c_int32]
    lib.gft_binning_bytes.restype = C.c_size_t
    lib.gft_binning_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int32]
    lib.gft_acc_bytes.restype = C.c_size_t
    lib.gft_acc_bytes.argtypes = [C.c_int32]
    lib.gft_det_partials_bytes.restype = C.c_size_t
    lib.gft_det_partials_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int32]
    lib.gft_binning_capacity.restype = C.c_int64
    lib.gft_binning_capacity.argtypes = [C.c_size_t, C.c_int32, C.c_int32]
    lib.gft_set_binning_mode.restype = C.c_int
    lib.gft_set_binning_mode.argtypes = [C.c_int]
    lib.gft_set_render_mode.restype = C.c_int
    lib.gft_set_render_mode.argtypes = [C.c_int]
    lib.gft_binning_mode.restype = C.c_int
    lib.gft_binning_mode.argtypes = [C.POINTER(Config)]
    lib.gft_get_layout.restype = C.c_int
    lib.gft_get_layout.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.POINTER(Layout)]
    lib.gft_forward_preprocess.restype = C.c_int
    lib.gft_forward_preprocess.argtypes = [C.c_void_p, C.POINTER(Config), C.POINTER(ForwardIO),
                                           C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.gft_forward_render.restype = C.c_int
    lib.gft_adam_step.restype = C.c_int
    lib.gft_adam_step.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64]
    lib.gft_adam_step_multi.restype = C.c_int
    lib.gft_adam_step_multi.argtypes = [C.c_void_p, C.c_int32, C.POINTER(AdamTensor), C.c_double, C.c_double, C.c_double,
                                        C.c_double]
    lib.gft_adam_step_multi_dev.restype = C.c_int
    lib.gft_adam_step_multi_dev.argtypes = [C.c_void_p, C.c_int32, C.POINTER(AdamTensor), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                            C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double]
    lib.gft_adam_step_rows.restype = C.c_int
    lib.gft_adam_step_rows.argtypes = [C.c_void_p, C.c_int32, C.POINTER(AdamTensor), C.c_int64, C.c_void_p, C.c_double, C.c_double,
                                       C.c_double, C.c_double]
    lib.gft_deform_packed_bytes.restype = C.c_size_t
    lib.gft_deform_packed_bytes.argtypes = []
    lib.gft_deform_saved_bytes.restype = C.c_size_t
    lib.gft_deform_saved_bytes.argtypes = [C.c_int64]
    lib.gft_deform_scratch_bytes.restype = C.c_size_t
    lib.gft_deform_scratch_bytes.argtypes = [C.c_int64]
    lib.gft_deform_pack.restype = C.c_int
    lib.gft_deform_inputs.restype = C.c_int
    lib.gft_deform_inputs.argtypes = [C.c_int, C.c_int]
    lib.gft_deform_pack.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(DeformParams), C.c_void_p]
    lib.gft_deform_forward.restype = C.c_int
    lib.gft_deform_forward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]
    lib.gft_deform_backward.restype = C.c_int
    lib.gft_deform_backward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(DeformParams)]
    lib.gft_deform_compact.restype = C.c_int
    lib.gft_deform_compact.argtypes = [C.c_void_p, C.c_int64, C.c_int64] + [C.c_void_p] * 9
    lib.gft_deform_rows_work_bytes.restype = C.c_size_t
    lib.gft_deform_rows_work_bytes.argtypes = [C.c_int64]
    lib.gft_deform_backward_rows.restype = C.c_int
    lib.gft_deform_backward_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.POINTER(DeformParams), C.c_void_p]
    lib.gft_ssim_blocks.restype = C.c_int64
    lib.gft_ssim_blocks.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.gft_ssim_l2_forward.restype = C.c_int
    lib.gft_ssim_l2_forward.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_float),
                                        C.c_void_p, C.c_void_p]
    lib.gft_ssim_l2_backward.restype = C.c_int
    lib.gft_ssim_l2_backward.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_float),
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
    lib.gft_densify_stats.restype = C.c_int
    lib.gft_densify_stats.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 8
    lib.gft_rows_rank_scratch_bytes.restype = C.c_size_t
    lib.gft_rows_rank_scratch_bytes.argtypes = [C.c_int64]
    lib.gft_rows_rank.restype = C.c_int
    lib.gft_rows_rank.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
    lib.gft_rows_rank_dev.restype = C.c_int
    lib.gft_rows_rank_dev.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gft_rows_any_nonzero.restype = C.c_int
    lib.gft_rows_any_nonzero.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.gft_rows_gather.restype = C.c_int
    lib.gft_rows_gather.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    lib.gft_knn_scratch_bytes.restype = C.c_size_t
    lib.gft_knn_scratch_bytes.argtypes = [C.c_int32]
    lib.gft_knn_mean_dist2.restype = C.c_int
    lib.gft_knn_mean_dist2.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gft_assemble_scratch_bytes.restype = C.c_size_t
    lib.gft_assemble_scratch_bytes.argtypes = [C.c_int32]
    lib.gft_assemble_forward.restype = C.c_int
    lib.gft_assemble_forward.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.POINTER(AssembleIO)]
    lib.gft_assemble_backward.restype = C.c_int
    lib.gft_assemble_backward.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                          C.POINTER(AssembleBwdIO)]
    lib.gft_assemble_num_dynamic.restype = C.c_int
    lib.gft_assemble_num_dynamic.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_int64)]
    lib.gft_forward.restype = C.c_int
    lib.gft_forward.argtypes = [C.c_void_p, C.POINTER(Config), C.POINTER(ForwardIO), C.POINTER(ForwardHints),
                                C.POINTER(ForwardReport)]
    lib.gft_forward_render.argtypes = [C.c_void_p, C.POINTER(Config), C.POINTER(ForwardIO), C.c_int64, C.c_int64]
    lib.gft_forward_enqueue.restype = C.c_int
    lib.gft_forward_enqueue.argtypes = [C.c_void_p, C.POINTER(Config), C.POINTER(ForwardIO), C.POINTER(ForwardHints), C.c_void_p]
    lib.gft_backward.restype = C.c_int
    lib.gft_backward.argtypes = [C.c_void_p, C.POINTER(Config), C.POINTER(BackwardIO), C.c_int64]
    lib.gft_grads_rezero.restype = C.c_int
    lib.gft_grads_rezero.argtypes = [C.c_void_p, C.POINTER(Config), C.POINTER(BackwardIO)]
    lib.gft_mark_visible.restype = C.c_int
    lib.gft_mark_visible.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_float, C.c_float, C.c_void_p]
    lib.gft_profile_enable.restype = C.c_int
    lib.gft_profile_enable.argtypes = [C.c_int]
    lib.gft_profile_reset.restype = C.c_int
    lib.gft_profile_read.restype = C.c_int
    lib.gft_profile_read.argtypes = [C.POINTER(Profile)]
    _lib = lib
    return lib


def raw_stream(dev):
    """hipStream_t of torch's current stream on `dev` as an integer (the C ABI takes void*).  The private fast path
    costs ~0.3 us, torch.cuda.current_stream(dev).cuda_stream ~4 us: a training iteration asks ~40 times."""
    import torch
    try:
        return torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device())
    except Exception:
        return torch.cuda.current_stream(dev).cuda_stream


class on_device:
    """`with torch.cuda.device(dev)` only when dev is not already the current device (the context manager costs
    ~10 us, the check ~1 us)."""

    def __init__(self, dev):
        import torch
        self.ctx = None
        idx = dev.index
        if idx is not None and idx != torch.cuda.current_device():
            self.ctx = torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


def last_error():
    return load().gft_last_error().decode("utf-8", "replace")


def check(rc):
    if rc != 0:
        raise RuntimeError(last_error())


def get_layout(P, W, H, R=0):
    L = Layout()
    check(load().gft_get_layout(P, W, H, R, C.byref(L)))
    return L


def profile_enable(on=True):
    load().gft_profile_enable(1 if on else 0)


def profile_reset():
    load().gft_profile_reset()


def profile_read():
    p = Profile()
    check(load().gft_profile_read(C.byref(p)))
    d = {n: getattr(p, n) for n in PROFILE_FIELDS}
    d["forward_calls"] = p.forward_calls
    d["backward_calls"] = p.backward_calls
    return d
