"""CPU ORACLE driver (test infrastructure, NOT product code).

ctypes/numpy front-end of ``libgft_oracle.so`` (oracle/gft_oracle.c), sequencing
the stages exactly as the reference's ``CudaRasterizer::Rasterizer::forward`` /
``::backward`` do (RAST/cuda_rasterizer/rasterizer_impl.cu:215-378, 382-499) and
allocating outputs as ``RasterizeGaussiansCUDA`` / ``...BackwardCUDA`` do
(RAST/rasterize_points.cu:80-92, 222-236).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may
import this module.  ``gftorf_amd`` never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgft_oracle.so")
_lib = None


class Config(C.Structure):
    _fields_ = [
        ("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("M_p", C.c_int),
        ("W", C.c_int), ("H", C.c_int),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float),
        ("scale_modifier", C.c_float),
        ("near_n", C.c_float), ("far_n", C.c_float),
        ("depth_range", C.c_float),
        ("phase_offset", C.c_float), ("dc_offset", C.c_float),
        ("use_view_dependent_phase", C.c_int), ("prefiltered", C.c_int),
    ]


def build(force=False):
    """Compile the C restatement (gcc is in the image and on the GPU box)."""
    src = os.path.join(_HERE, "gft_oracle.c")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.gfto_get_higher_msb.restype = C.c_uint32
        _lib.gfto_get_higher_msb.argtypes = [C.c_uint32]
        _lib.gfto_scan.restype = C.c_uint32
        _lib.gfto_preprocess_fwd.restype = C.c_int
        _lib.gfto_num_threads.restype = C.c_int
        _lib.gfto_set_num_threads.restype = C.c_int
        _lib.gfto_set_num_threads.argtypes = [C.c_int]
        _lib.gfto_set_num_threads(C.c_int(min(int(_lib.gfto_num_threads()), cpu_budget())))
    return _lib


# Kept arrays for timing loops (bench.py's cpu_baseline): with reuse_buffers(True) the big outputs and scratch arrays
# are allocated once per (name, shape) and zero-filled on all cores for every call -- like the reference, which memsets
# its outputs on the device -- instead of being page-faulted in and unmapped again by every iteration (single-threaded
# kernel work that dominated the 128-core timing).  Results of an earlier call are overwritten by the next one: tests
# leave it off.
_keep = None


def reuse_buffers(on=True):
    global _keep
    _keep = {} if on else None


def _zeros(tag, shape, dtype):
    if _keep is None:
        return np.zeros(shape, dtype)
    shape = (shape,) if isinstance(shape, int) else tuple(shape)
    key = (tag, shape, np.dtype(dtype).str)
    a = _keep.get(key)
    if a is None:
        a = _keep[key] = np.empty(shape, dtype)
    if a.nbytes:
        lib().gfto_zero(C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes))
    return a


def cpu_budget():
    """CPUs this process may really use: the smallest of the logical CPU count, the affinity mask and the cgroup's CPU
    quota (the GPU boxes show 256 logical CPUs to a container whose cgroup allows 16 CPUs' worth of time: 128 OpenMP
    threads there are throttled to a crawl -- 0.69 s per frame against 0.28 s with 16)."""
    import math
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                   # cgroup v2
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())                    # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, math.ceil(quota / period)))
        except (OSError, ValueError):
            pass
    return n


def _p(a):
    """numpy array (or None) -> void* ; None == tensor absent (NULL)."""
    if a is None:
        return C.c_void_p(0)
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be contiguous"
    return C.c_void_p(a.ctypes.data)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def get_higher_msb(n):
    return int(lib().gfto_get_higher_msb(int(n)))


def num_threads():
    return int(lib().gfto_num_threads())


def set_num_threads(n):
    """OpenMP threads of the following oracle calls; returns the count in effect."""
    return int(lib().gfto_set_num_threads(int(n)))


def mark_visible(means3D, viewmatrix, projmatrix, near_n, far_n):
    means3D = _f32(means3D)
    P = means3D.shape[0]
    out = np.zeros(P, dtype=np.uint8)
    lib().gfto_mark_visible(C.c_int(P), _p(means3D), _p(_f32(viewmatrix).reshape(-1)),
                            _p(_f32(projmatrix).reshape(-1)), C.c_float(near_n),
                            C.c_float(far_n), _p(out))
    return out.astype(bool)


def make_config(P, D, M, M_p, W, H, tanfovx, tanfovy, scale_modifier=1.0,
                near_n=0.01, far_n=100.0, depth_range=100.0, phase_offset=0.0,
                dc_offset=0.0, use_view_dependent_phase=False, prefiltered=False):
    return Config(P, D, M, M_p, W, H, tanfovx, tanfovy, scale_modifier, near_n,
                  far_n, depth_range, phase_offset, dc_offset,
                  int(bool(use_view_dependent_phase)), int(bool(prefiltered)))


# ---------------------------------------------------------------------------
# individual stages (numpy in / numpy out) -- used by the GPU parity tests to
# check integer stages bit-exactly on identical float inputs.
# ---------------------------------------------------------------------------
def preprocess_fwd(cfg, means3D, scales, rotations, opacities, shs, shs_p,
                   cov3D_precomp, colors_precomp, phasors_precomp, viewmatrix,
                   projmatrix, campos):
    P = cfg.P
    z = lambda n, shape, dt: _zeros("g." + n, shape, dt)
    g = dict(
        radii=z("radii", P, np.int32), means2D=z("means2D", (P, 2), np.float32),
        depths=z("depths", P, np.float32), dists_ndc=z("dists_ndc", P, np.float32),
        cov3D=z("cov3D", (P, 6), np.float32), conic_opacity=z("conic_opacity", (P, 4), np.float32),
        rgb=z("rgb", (P, 3), np.float32), phasor7=z("phasor7", (P, 7), np.float32),
        dists=z("dists", P, np.float32), phase_amp=z("phase_amp", (P, 2), np.float32),
        clamped=z("clamped", (P, 3), np.uint8), clamped_p=z("clamped_p", P, np.uint8),
        tiles_touched=z("tiles_touched", P, np.uint32),
    )
    rc = lib().gfto_preprocess_fwd(
        C.byref(cfg), _p(means3D), _p(scales), _p(rotations), _p(opacities),
        _p(shs), _p(shs_p), _p(cov3D_precomp), _p(colors_precomp),
        _p(phasors_precomp), _p(viewmatrix), _p(projmatrix), _p(campos),
        _p(g["radii"]), _p(g["means2D"]), _p(g["depths"]), _p(g["dists_ndc"]),
        _p(g["cov3D"]), _p(g["conic_opacity"]), _p(g["rgb"]), _p(g["phasor7"]),
        _p(g["dists"]), _p(g["phase_amp"]), _p(g["clamped"]), _p(g["clamped_p"]),
        _p(g["tiles_touched"]))
    if rc != 0:
        raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")
    return g


def bin_and_sort(W, H, means2D, radii, depths, tiles_touched=None):
    """K2..K5 given preprocess outputs -> (R, offsets, keys_sorted, point_list, ranges)."""
    P = radii.shape[0]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    T = gx * gy
    if tiles_touched is None:
        tiles_touched = tiles_from_rect(W, H, means2D, radii)
    offsets = _zeros("offsets", P, np.uint32)
    R = int(lib().gfto_scan(C.c_int(P), _p(np.ascontiguousarray(tiles_touched, np.uint32)), _p(offsets))) if P else 0
    if _keep is None:
        keys = np.zeros(max(R, 1), np.uint64)
        vals = np.zeros(max(R, 1), np.uint32)
        keys_s = np.zeros(max(R, 1), np.uint64)
        vals_s = np.zeros(max(R, 1), np.uint32)
    else:       # (every slot in [0, R) is written by the duplication / the sort: kept arrays need no fill)
        cap = _keep.get("cap", 0)
        if max(R, 1) > cap:
            cap = _keep["cap"] = int(max(R, 1) * 1.25)
            _keep["kv"] = [np.empty(cap, np.uint64), np.empty(cap, np.uint32), np.empty(cap, np.uint64), np.empty(cap, np.uint32)]
        keys, vals, keys_s, vals_s = _keep["kv"]
    ranges = np.zeros((T, 2), np.uint32)
    if P:
        lib().gfto_duplicate_with_keys(C.c_int(P), C.c_int(W), C.c_int(H), _p(means2D),
                                       _p(depths), _p(offsets), _p(radii), _p(keys), _p(vals))
    bit = get_higher_msb(T)
    lib().gfto_sort_pairs(C.c_uint32(R), _p(keys), _p(vals), _p(keys_s), _p(vals_s),
                          C.c_int(32 + bit))
    lib().gfto_tile_ranges(C.c_uint32(R), _p(keys_s), C.c_int(T), _p(ranges))
    return R, offsets, keys_s[:R], vals_s[:R], ranges


def tiles_from_rect(W, H, means2D, radii):
    """tiles_touched recomputed from (means2D, radii) with the reference getRect
    (RAST/cuda_rasterizer/auxiliary.h:49-59); numpy fp32 restatement."""
    gx, gy = (W + 15) // 16, (H + 15) // 16
    px = means2D[:, 0].astype(np.float32)
    py = means2D[:, 1].astype(np.float32)
    r = radii.astype(np.int32).astype(np.float32)
    f16, f1 = np.float32(16), np.float32(1)
    x0 = np.clip(np.trunc((px - r) / f16).astype(np.int64), 0, gx)
    y0 = np.clip(np.trunc((py - r) / f16).astype(np.int64), 0, gy)
    x1 = np.clip(np.trunc((((px + r) + f16) - f1) / f16).astype(np.int64), 0, gx)
    y1 = np.clip(np.trunc((((py + r) + f16) - f1) / f16).astype(np.int64), 0, gy)
    t = (x1 - x0) * (y1 - y0)
    t[radii <= 0] = 0
    return t.astype(np.uint32)


def render_fwd(W, H, ranges, point_list, g, bg, P):
    N = W * H
    z = lambda n, shape, dt: _zeros("img." + n, shape, dt)
    o = dict(
        final_T=z("final_T", N, np.float32), n_contrib=z("n_contrib", N, np.uint32),
        w_z_total=z("w_z_total", N, np.float32), w_z2_total=z("w_z2_total", N, np.float32),
        color=z("color", (3, H, W), np.float32), phasor=z("phasor", (7, H, W), np.float32),
        depth=z("depth", (1, H, W), np.float32), acc=z("acc", (1, H, W), np.float32),
        depth_distortion=z("depth_distortion", (1, H, W), np.float32),
        distribution=z("distribution", (3, H, W), np.float32),
        pixels=z("pixels", (P, 1), np.float32),
    )
    pl = np.ascontiguousarray(point_list, np.uint32)
    if pl.size == 0:
        pl = np.zeros(1, np.uint32)
    lib().gfto_render_fwd(
        C.c_int(W), C.c_int(H), _p(ranges), _p(pl), _p(g["means2D"]), _p(g["rgb"]),
        _p(g["phasor7"]), _p(g["dists"]), _p(g["conic_opacity"]), _p(g["dists_ndc"]),
        _p(bg), _p(o["final_T"]), _p(o["n_contrib"]), _p(o["w_z_total"]),
        _p(o["w_z2_total"]), _p(o["color"]), _p(o["phasor"]), _p(o["depth"]),
        _p(o["acc"]), _p(o["depth_distortion"]), _p(o["distribution"]), _p(o["pixels"]))
    return o


def render_bwd(W, H, ranges, point_list, P, bg, g, img, dL_dcolor, dL_dphasor,
               dL_ddepth, dL_dacc, dL_ddd):
    acc = _zeros("bwd.acc", (P, 18), np.float32)
    pl = np.ascontiguousarray(point_list, np.uint32)
    if pl.size == 0:
        pl = np.zeros(1, np.uint32)
    lib().gfto_render_bwd(
        C.c_int(W), C.c_int(H), _p(ranges), _p(pl), C.c_int(P), _p(bg),
        _p(g["means2D"]), _p(g["conic_opacity"]), _p(g["rgb"]), _p(g["phasor7"]),
        _p(g["dists"]), _p(g["dists_ndc"]), _p(img["final_T"]), _p(img["w_z_total"]),
        _p(img["w_z2_total"]), _p(img["n_contrib"]), _p(dL_dcolor), _p(dL_dphasor),
        _p(dL_ddepth), _p(dL_dacc), _p(dL_ddd), _p(acc))
    return acc


# ---------------------------------------------------------------------------
# whole forward / backward with the reference's operator semantics
# ---------------------------------------------------------------------------
class ForwardResult(dict):
    __getattr__ = dict.__getitem__


def forward(means3D, opacities, *, shs=None, shs_p=None, colors_precomp=None,
            phasors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
            bg, viewmatrix, projmatrix, campos, image_height, image_width,
            tanfovx, tanfovy, sh_degree, scale_modifier=1.0, prefiltered=False,
            near_n=0.01, far_n=100.0, depth_range=100.0,
            use_view_dependent_phase=False, phase_offset=0.0, dc_offset=0.0):
    """Equivalent of _C.rasterize_gaussians (RAST/rasterize_points.cu:42-165).
    bg must be a [7,H,W] array (any broadcastable view is materialised, like the
    reference's .contiguous())."""
    means3D = _f32(means3D)
    P = means3D.shape[0]
    H, W = int(image_height), int(image_width)
    shs, shs_p = _f32(shs), _f32(shs_p)
    M = shs.shape[1] if (shs is not None and shs.shape[0] != 0) else 0
    M_p = shs_p.shape[1] if (shs_p is not None and shs_p.shape[0] != 0) else 0
    if shs is not None and shs.size == 0:
        shs = None
    if shs_p is not None and shs_p.size == 0:
        shs_p = None
    cfg = make_config(P, sh_degree, M, M_p, W, H, tanfovx, tanfovy, scale_modifier,
                      near_n, far_n, depth_range, phase_offset, dc_offset,
                      use_view_dependent_phase, prefiltered)
    bg = np.ascontiguousarray(np.broadcast_to(_f32(bg), (7, H, W)))
    view = _f32(viewmatrix).reshape(-1)
    proj = _f32(projmatrix).reshape(-1)
    campos = _f32(campos).reshape(-1)
    inputs = dict(means3D=means3D, scales=_f32(scales), rotations=_f32(rotations),
                  opacities=_f32(opacities).reshape(-1), shs=shs, shs_p=shs_p,
                  cov3D_precomp=_f32(cov3D_precomp), colors_precomp=_f32(colors_precomp),
                  phasors_precomp=_f32(phasors_precomp), view=view, proj=proj,
                  campos=campos, bg=bg)
    res = ForwardResult(cfg=cfg, inputs=inputs, P=P, W=W, H=H)
    zeros = lambda c: np.zeros((c, H, W), np.float32)
    res.update(normal=zeros(3), entropy=zeros(1), amp_distortion=zeros(1))
    if P == 0:
        res.update(color=zeros(3), phasor=zeros(7), depth=zeros(1), acc=zeros(1),
                   depth_distortion=zeros(1), distribution=zeros(3),
                   pixels=np.zeros((0, 1), np.float32), radii=np.zeros(0, np.int32),
                   num_rendered=0)
        return res
    g = preprocess_fwd(cfg, means3D, inputs["scales"], inputs["rotations"],
                       inputs["opacities"], shs, shs_p, inputs["cov3D_precomp"],
                       inputs["colors_precomp"], inputs["phasors_precomp"], view,
                       proj, campos)
    R, offsets, keys_s, point_list, ranges = bin_and_sort(
        W, H, g["means2D"], g["radii"], g["depths"], g["tiles_touched"])
    img = render_fwd(W, H, ranges, point_list, g, bg, P)
    res.update(geom=g, num_rendered=R, offsets=offsets, keys_sorted=keys_s,
               point_list=point_list, ranges=ranges, img=img,
               color=img["color"], phasor=img["phasor"], depth=img["depth"],
               acc=img["acc"], depth_distortion=img["depth_distortion"],
               distribution=img["distribution"], pixels=img["pixels"],
               radii=g["radii"])
    return res


def backward(fwd, dL_dcolor, dL_dphasor, dL_ddepth, dL_dacc, dL_ddepth_distortion):
    """Equivalent of _C.rasterize_gaussians_backward
    (RAST/rasterize_points.cu:167-281).  Returns a dict of gradients named like
    the reference's return tuple."""
    cfg, inp = fwd.cfg, fwd.inputs
    P, W, H = fwd.P, fwd.W, fwd.H
    M, M_p = cfg.M, cfg.M_p
    z = lambda n, shape: _zeros("bwd." + n, shape, np.float32)
    out = dict(
        dL_dmeans3D=z("means3D", (P, 3)), dL_dmeans2D=z("means2D", (P, 3)),
        dL_dcolors=z("colors", (P, 3)), dL_dphasors=z("phasors", (P, 7)),
        dL_dopacity=z("opacity", (P, 1)), dL_dcov3D=z("cov3D", (P, 6)),
        dL_dsh=z("sh", (P, M, 3)), dL_dsh_p=z("sh_p", (P, M_p, 2)),
        dL_dscales=z("scales", (P, 3)), dL_drotations=z("rotations", (P, 4)),
        dL_dphase_offset=np.zeros(1, np.float32), dL_ddc_offset=np.zeros(1, np.float32),
    )
    if P == 0:
        return out
    g = fwd.geom
    f = lambda a, c: np.ascontiguousarray(np.broadcast_to(_f32(a), (c, H, W)))
    acc = render_bwd(W, H, fwd.ranges, fwd.point_list, P, inp["bg"], g, fwd.img,
                     f(dL_dcolor, 3), f(dL_dphasor, 7), f(dL_ddepth, 1), f(dL_dacc, 1),
                     f(dL_ddepth_distortion, 1))
    out["acc"] = acc
    # the 18 per-Gaussian sums of the render backward -> the arrays the reference keeps separately
    # (rasterize_points.cu:222-236; dL_dconic is float4 with .z unused, backward.cu:883-885)
    dL_dconic = z("conic", (P, 4))
    dL_ddist = z("ddist", P)
    dL_dndc = z("dndc", P)
    lib().gfto_unpack_acc(C.c_int(P), _p(acc), _p(out["dL_dmeans2D"]), _p(dL_dconic), _p(out["dL_dopacity"]),
                          _p(out["dL_dcolors"]), _p(out["dL_dphasors"]), _p(dL_ddist), _p(dL_dndc))
    out["dL_dconic"] = dL_dconic
    cov3D = inp["cov3D_precomp"] if inp["cov3D_precomp"] is not None else g["cov3D"]
    lib().gfto_preprocess_bwd(
        C.byref(cfg), _p(inp["means3D"]), _p(g["radii"]), _p(inp["shs"]), _p(inp["shs_p"]),
        _p(g["clamped"]), _p(g["clamped_p"]), _p(inp["scales"]), _p(inp["rotations"]),
        _p(np.ascontiguousarray(cov3D)), _p(inp["view"]), _p(inp["proj"]), _p(inp["campos"]),
        _p(out["dL_dmeans2D"]), _p(dL_dconic), _p(out["dL_dcolors"]), _p(out["dL_dphasors"]),
        _p(dL_ddist), _p(dL_dndc), _p(g["phase_amp"]), _p(g["dists"]),
        _p(out["dL_dmeans3D"]), _p(out["dL_dcov3D"]), _p(out["dL_dsh"]), _p(out["dL_dsh_p"]),
        _p(out["dL_dscales"]), _p(out["dL_drotations"]), _p(out["dL_dphase_offset"]),
        _p(out["dL_ddc_offset"]))
    return out
