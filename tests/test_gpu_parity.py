"""GPU parity: HIP path (public API -> C ABI -> gfx950 kernels) vs the CPU oracle on
identical seeded inputs.  Integer/index stages bit-exact, fp32 stages within the
tolerances written below (north star: rendered L1 < 1e-5, bit-exact tile/key
indexing)."""
import contextlib
import ctypes as C
import os

import numpy as np
import pytest
import torch

import helpers as Hh

pytestmark = pytest.mark.gpu

# fp32 tolerances (relative to the max-norm of the reference tensor)
IMG_L1 = 1e-5          # north star: mean abs error of rendered images
IMG_MAX = 2e-4         # max-norm, allows a handful of borderline alpha/termination flips
GRAD_RTOL = 3e-4       # gradients: fp32 atomics/reduction order + T recovered by division


# ---------------------------------------------------------------------------
def raw_forward(scene, dev, inputs=None, mode=None, hints=None):
    """Drive the C ABI directly so the scratch buffers can be inspected.  mode: 1 = tile-pull binning, 0 = whole-frame
    binning (gft_set_binning_mode), None = the library's default.  hints: int32[T] device tensor handed over as
    gft_forward_io.tile_hints (the forward overwrites it; returned as st["hints"])."""
    from gftorf_amd import _lib
    lib = _lib.load()
    g = dict(scene["gaussians"])
    if inputs:
        g.update(inputs)
    cfgd, cam = scene["cfg"], scene["cam"]
    P, W, H = g["means3D"].shape[0], cfgd["W"], cfgd["H"]
    t = lambda a: None if a is None else torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
    T = {k: t(v) for k, v in g.items()}
    view, proj, campos, bg = t(cam["viewmatrix"]), t(cam["projmatrix"]), t(cam["campos"]), t(scene["bg"])
    c = _lib.Config()
    c.P, c.D, c.W, c.H = P, cfgd["D"], W, H
    c.M = g["shs"].shape[1] if g.get("shs") is not None else 0
    c.M_p = g["shs_p"].shape[1] if g.get("shs_p") is not None else 0
    c.tanfovx, c.tanfovy, c.scale_modifier = cam["tanfovx"], cam["tanfovy"], 1.0
    c.near_n, c.far_n, c.depth_range = cam["znear"], cam["zfar"], scene["depth_range"]
    c.phase_offset, c.dc_offset = scene["phase_offset"], scene["dc_offset"]
    c.use_view_dependent_phase = int(scene["use_view_dependent_phase"])
    c.bg_stride_c, c.bg_stride_y, c.bg_stride_x = H * W, W, 1
    planes = torch.full((21, H, W), float("nan"), device=dev)
    radii = torch.full((P,), -7, device=dev, dtype=torch.int32)
    pixels = torch.full((P, 1), float("nan"), device=dev)
    geom = torch.zeros(lib.gft_geom_bytes(P), device=dev, dtype=torch.uint8)
    img = torch.zeros(lib.gft_image_bytes(W, H), device=dev, dtype=torch.uint8)
    io = _lib.ForwardIO()
    p = lambda x: None if x is None else x.data_ptr()
    io.bg, io.means3D, io.opacities = p(bg), p(T["means3D"]), p(T["opacities"])
    io.colors_precomp, io.phasors_precomp = p(T.get("colors_precomp")), p(T.get("phasors_precomp"))
    io.scales, io.rotations, io.cov3D_precomp = p(T.get("scales")), p(T.get("rotations")), p(T.get("cov3D_precomp"))
    io.viewmatrix, io.projmatrix, io.campos = p(view), p(proj), p(campos)
    io.shs, io.shs_p = p(T.get("shs")), p(T.get("shs_p"))
    io.geom, io.img = p(geom), p(img)
    sl = [0, 3, 10, 11, 14, 15, 16, 17, 18, 21]
    names = ["out_color", "out_phasor", "out_depth", "out_normal", "out_acc", "out_entropy",
             "out_depth_distortion", "out_amp_distortion", "out_distribution"]
    for n, a in zip(names, sl[:-1]):
        setattr(io, n, planes[a].data_ptr())
    io.pixels, io.radii = p(pixels), p(radii)
    io.tile_hints = p(hints)
    R = C.c_int64(0)
    MX = C.c_int64(0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    if mode is not None:
        lib.gft_set_binning_mode(int(mode))
    try:
        pull = bool(lib.gft_binning_mode(C.byref(c)))
        _lib.check(lib.gft_forward_preprocess(stream, C.byref(c), C.byref(io), C.byref(R), C.byref(MX)))
        R = int(R.value)
        binning = torch.zeros(lib.gft_binning_bytes(R, W, H), device=dev, dtype=torch.uint8)
        io.binning = p(binning)
        _lib.check(lib.gft_forward_render(stream, C.byref(c), C.byref(io), R, int(MX.value)))
        torch.cuda.synchronize()
    finally:
        if mode is not None:
            lib.gft_set_binning_mode(-1)
    L = _lib.get_layout(P, W, H, R)
    keep = (T, view, proj, campos, bg)

    def view_of(buf, off, count, dtype):
        nbytes = count * torch.empty((), dtype=dtype).element_size()
        return buf[off:off + nbytes].view(dtype).cpu().numpy()

    Tn = ((W + 15) // 16) * ((H + 15) // 16)
    st = dict(
        R=R, radii=radii.cpu().numpy(), pixels=pixels.cpu().numpy(), planes=planes.cpu().numpy(), pull=pull,
        rec_a=view_of(geom, L.geom_rec_a, P * 8, torch.float32).reshape(P, 8),
        rec_b=view_of(geom, L.geom_rec_b, P * 8, torch.float32).reshape(P, 8),
        depth=view_of(geom, L.geom_depth, P, torch.float32),
        tiles=view_of(geom, L.geom_tiles, P, torch.int32).astype(np.uint32),
        rect=view_of(geom, L.geom_rect, P * 4, torch.int16).astype(np.uint16).reshape(P, 4),
        clamped=view_of(geom, L.geom_clamped, P, torch.uint8),
        need=view_of(geom, L.geom_need, P, torch.uint8),
        ctrl=view_of(img, L.img_ctrl, 16, torch.int32).astype(np.uint32),
        pix_state=view_of(img, L.img_pix_state, W * H * 4, torch.float32).reshape(H * W, 4),
        ranges=view_of(img, L.img_ranges, Tn * 2, torch.int32).astype(np.uint32).reshape(Tn, 2),
        tile_max=view_of(img, L.img_tile_max, Tn * 4, torch.int32).astype(np.uint32).reshape(Tn, 4),
        tile_cnt=view_of(img, L.img_tile_cnt, Tn, torch.int32).astype(np.uint32),
        tile_cut=view_of(img, L.img_tile_cut, Tn, torch.int32).astype(np.uint32),
        front_len=view_of(img, L.img_front_len, Tn, torch.int32).astype(np.uint32),
        unit_flag=view_of(img, L.img_unit_flag, Tn * 4, torch.int32).astype(np.uint32).reshape(Tn, 4),
        lazy=bool(lib.gft_lazy_sort()),
        hints=None if hints is None else hints.cpu().numpy().view(np.uint8).reshape(Tn, 4),
        point_list=view_of(binning, L.bin_point_list, Tn * 2048 + R, torch.int32).astype(np.uint32) if R else np.zeros(0, np.uint32),
    )
    del keep
    return st


def is_subsequence(sub, full):
    """every element of `sub` occurs in `full`, in the same order"""
    it = iter(full.tolist())
    return all(any(x == y for y in it) for x in sub.tolist())


SCENES = {
    "base": dict(),
    "odd_size": dict(W=50, H=37, P=300),
    "deep_lists": dict(P=1500, W=48, H=32, scale_lo=0.03, scale_hi=0.2),   # > 256 splats per tile
    "wide_spread": dict(spread=1.7, P=500),                               # frustum clamp of t.x/t.z
    "identity_cam": dict(w2c=None),
    "tiny_splats": dict(scale_lo=0.001, scale_hi=0.004, P=600),
    "opaque_early_exit": dict(P=2500, W=64, H=48, scale_lo=0.05, scale_hi=0.25, opacity=0.95),  # T < 1e-4 termination
    "long_lists_lds128k": dict(P=9000, W=32, H=32, scale_lo=0.05, scale_hi=0.3),    # 4096 < tile list <= 16384: depth-bucket split
    "long_lists_flat": dict(P=9000, W=32, H=32, scale_lo=0.05, scale_hi=0.3, w2c=None, z_lo=3.0, z_hi=3.0),   # all depths equal: splitters and order decided by the id half of the keys
    "long_lists_global": dict(P=24000, W=32, H=16, scale_lo=0.05, scale_hi=0.3),    # tile list > 16384: global-memory sort
    "scatter_direct": dict(P=9000, W=64, H=64, scale_lo=0.25, scale_hi=0.6),         # 30 k instances per 4096-Gaussian workgroup: more than the LDS stage takes, keys are written directly
    # tile-pull binning places a head's keys by depth bin (k_tile_pull): whole lists over a depth range of eight octaves span
    # more bins than there are cursors (one cursor per two bins) ...
    "wide_depth_range": dict(P=2500, W=64, H=48, scale_lo=0.02, scale_hi=0.2, z_lo=0.25, z_hi=70.0),
    # ... and a few hundred Gaussians per tile inside 2 % of one depth crowd single bins (beyond the 5-bit group count: the
    # keys are sorted as a whole)
    "crowded_depth_bins": dict(P=4000, W=48, H=32, scale_lo=0.03, scale_hi=0.2, z_lo=3.0, z_hi=3.05),
}


@pytest.mark.parametrize("mode", [1, 0], ids=["tile_pull", "whole_frame"])
@pytest.mark.parametrize("name", list(SCENES))
def test_preprocess_and_binning_bit_exact(name, mode, oracle, gpu):
    scene = Hh.small_scene(**SCENES[name])
    f, _ = Hh.run_oracle(oracle, scene, backward=False)
    st = raw_forward(scene, gpu, mode=mode)
    og = f.geom
    vis = og["radii"] > 0
    pull = st["pull"]
    assert pull == (mode == 1 and st["lazy"])
    # integer decisions
    np.testing.assert_array_equal(st["radii"], og["radii"])
    np.testing.assert_array_equal(st["tiles"], og["tiles_touched"])
    assert st["R"] == f.num_rendered == int(st["ctrl"][0])
    lens = f.ranges[:, 1] - f.ranges[:, 0]
    if os.environ.get("GFT_SLABS", "1") in ("", "0", "1"):
        np.testing.assert_array_equal(st["tile_cnt"], lens)      # (small frames: one depth slab, every tile scans its whole list)
    if not pull:
        assert int(st["ctrl"][2]) == int(lens.max())
    W, H = scene["cfg"]["W"], scene["cfg"]["H"]
    np.testing.assert_array_equal(
        (st["rect"][:, 2].astype(np.int64) - st["rect"][:, 0]) * (st["rect"][:, 3].astype(np.int64) - st["rect"][:, 1]),
        og["tiles_touched"])
    # IEEE-exact float stages (contraction disabled on both sides)
    np.testing.assert_array_equal(st["rec_a"][vis, 0:2], og["means2D"][vis])
    np.testing.assert_array_equal(st["depth"][vis].view(np.uint32), og["depths"][vis].view(np.uint32))
    np.testing.assert_array_equal(st["rec_a"][vis, 2:4], og["conic_opacity"][vis, 0:2])
    np.testing.assert_array_equal(st["rec_a"][vis, 4:6], og["conic_opacity"][vis, 2:4])
    np.testing.assert_array_equal(st["rec_a"][vis, 6], og["dists_ndc"][vis])
    np.testing.assert_array_equal(st["rec_a"][vis, 7], og["dists"][vis])
    # appearance: of every visible Gaussian (whole-frame binning), of those that stand in the sorted part of a list
    # (tile-pull binning: the Gaussians marked in `need`)
    app = vis & (st["need"] != 0) if pull else vis
    if pull:
        assert not (st["need"][~vis]).any()
    np.testing.assert_array_equal(st["rec_b"][app, 0:3], og["rgb"][app])
    cl = og["clamped"][app, 0] | (og["clamped"][app, 1] << 1) | (og["clamped"][app, 2] << 2) | (og["clamped_p"][app] << 3)
    np.testing.assert_array_equal(st["clamped"][app], cl)
    # transcendental stage (sinf/cosf differ by ulps between glibc and the device library)
    # device keeps the phasor on its (R, I, Am) basis; planes 3..6 are +-R + dc*Am, +-I + dc*Am
    Hh.assert_close("phasor RIA", og["phasor7"][app, 0:3], st["rec_b"][app, 3:6], rtol_max=2e-6, atol=1e-9)
    dc = scene["dc_offset"]
    R_, I_, A_ = st["rec_b"][app, 3], st["rec_b"][app, 4], st["rec_b"][app, 5]
    quad = np.stack([R_ + dc * A_, -R_ + dc * A_, I_ + dc * A_, -I_ + dc * A_], 1)
    Hh.assert_close("phasor quads", og["phasor7"][app, 3:7], quad, rtol_max=3e-6, atol=1e-9)
    Hh.assert_close("phase_amp", og["phase_amp"][app], st["rec_b"][app, 6:8], rtol_max=1e-6, atol=1e-9)
    pl = st["point_list"]
    if pull:
        # Tile-pull binning: the sorted HEAD of every list (whole depth bins, about 940 ids; the whole list when it is
        # short) = the first ids of the reference's sorted list; a tile with a flagged quadrant has its list completed:
        # head + the ids of the tail that can reach a flagged quadrant, in the reference's order.
        listed = np.zeros(st["need"].shape, bool)
        for t in range(lens.size):
            a, b = int(f.ranges[t, 0]), int(f.ranges[t, 1])
            ref = f.point_list[a:b]
            k = int(st["front_len"][t])
            r0, r1 = int(st["ranges"][t, 0]), int(st["ranges"][t, 1])
            assert k <= b - a and k <= 2048
            assert (int(st["tile_cut"][t]) == 0xffffffff) == (k == b - a), "tile %d" % t
            one_slab = os.environ.get("GFT_SLABS", "1") in ("", "0", "1")      # (small frames: one depth slab unless forced)
            if b - a <= 2048 and one_slab:
                assert k == b - a       # the tile has seen its whole list (with several slabs it stops at the one that covers the head)
            elif k < b - a:
                # whole bins up to the one where the count reaches 940; a bin that overshoots 1024 is left out unless
                # the head would be shorter than 512
                assert k <= 1024 or k <= 2048
            got = pl[r0:r1]
            listed[got] = True
            np.testing.assert_array_equal(got[:k], ref[:k], err_msg="head of tile %d" % t)
            if st["unit_flag"][t].any() and k < b - a:
                assert r0 >= lens.size * 2048, "a completed list lives in the pool"
                assert is_subsequence(got[k:], ref[k:]), "culled tail of tile %d" % t
            else:
                assert r1 - r0 == k and (k == 0 or r0 == t * 2048)
        np.testing.assert_array_equal(st["need"] != 0, listed)
        if int(st["front_len"].sum()):
            heads = np.concatenate([pl[int(st["ranges"][t, 0]):int(st["ranges"][t, 0]) + int(st["front_len"][t])] for t in range(lens.size)])
            tile_of = np.repeat(np.arange(lens.size, dtype=np.uint64), st["front_len"].astype(np.int64))
            keys = (tile_of << np.uint64(32)) | st["depth"][heads].view(np.uint32).astype(np.uint64)
            ref_keys = np.concatenate([f.keys_sorted[int(f.ranges[t, 0]):int(f.ranges[t, 0]) + int(st["front_len"][t])] for t in range(lens.size)])
            np.testing.assert_array_equal(keys, ref_keys)
        return
    # whole-frame binning: sorted list / ranges bit-identical; the reference's 64-bit keys follow from them
    np.testing.assert_array_equal(st["ranges"], f.ranges)
    pl = pl[:st["R"]].copy()
    if st["lazy"]:
        # lazy sort: the head of every list is sorted, the tail is sorted only where a quadrant
        # needed it; everywhere the list holds exactly the tile's instances
        for t in range(lens.size):
            a, b = int(f.ranges[t, 0]), int(f.ranges[t, 1])
            k = int(st["front_len"][t])
            assert k <= b - a
            if st["unit_flag"][t].any() or k == b - a:
                np.testing.assert_array_equal(pl[a:b], f.point_list[a:b], err_msg="tile %d" % t)
            else:
                np.testing.assert_array_equal(pl[a:a + k], f.point_list[a:a + k], err_msg="head of tile %d" % t)
                np.testing.assert_array_equal(np.sort(pl[a + k:b]), np.sort(f.point_list[a + k:b]), err_msg="tail of tile %d" % t)
                pl[a + k:b] = f.point_list[a + k:b]
    np.testing.assert_array_equal(pl, f.point_list)
    tile_of = np.repeat(np.arange(lens.size, dtype=np.uint64), lens)
    keys = (tile_of << np.uint64(32)) | st["depth"][pl].view(np.uint32).astype(np.uint64)
    np.testing.assert_array_equal(keys, f.keys_sorted)


@pytest.mark.parametrize("name", list(SCENES))
def test_forward_backward_vs_oracle(name, oracle, gpu):
    scene = Hh.small_scene(**SCENES[name])
    f, b = Hh.run_oracle(oracle, scene)
    out, grads, _ = Hh.run_gpu(scene, gpu, optimize_offsets=True)
    check_outputs(f, out)
    check_grads(b, grads, scene)


def check_outputs(f, out):
    for k in ["color", "phasor", "depth", "acc", "depth_distortion", "distribution"]:
        l1 = np.abs(out[k].astype(np.float64) - f[k]).mean()
        scale = max(1.0, np.abs(f[k]).max())
        assert l1 < IMG_L1 * scale, "%s: L1 %.3g" % (k, l1)
        Hh.assert_close(k, f[k], out[k], rtol_max=IMG_MAX, atol=1e-6, frac_bad=1e-3)
    for k in ["normal", "entropy", "amp_distortion"]:
        assert not out[k].any(), k
    np.testing.assert_array_equal(out["radii"], f.radii)
    # integer-valued counters: exact unless an alpha sits on the 1/255 or T=1e-4 edge
    mism = (out["pixels"] != f.pixels).mean() if f.pixels.size else 0.0
    assert mism <= 2e-3, "pixels mismatch fraction %.3g" % mism
    assert abs(out["pixels"].sum() - f.pixels.sum()) <= 1e-4 * max(1.0, f.pixels.sum())


def check_grads(b, grads, scene, rtol=GRAD_RTOL):
    g = scene["gaussians"]
    Hh.assert_close("dL_dmeans3D", b["dL_dmeans3D"], grads["means3D"], rtol_max=rtol)
    Hh.assert_close("dL_dmeans2D", b["dL_dmeans2D"], grads["means2D"], rtol_max=rtol)
    Hh.assert_close("dL_dopacity", b["dL_dopacity"].reshape(g["opacities"].shape), grads["opacities"], rtol_max=rtol)
    if g.get("shs") is not None:
        Hh.assert_close("dL_dsh", b["dL_dsh"], grads["shs"], rtol_max=rtol)
    if g.get("shs_p") is not None:
        Hh.assert_close("dL_dsh_p", b["dL_dsh_p"], grads["shs_p"], rtol_max=rtol)
        if "phase_offset" in grads:
            Hh.assert_close("dL_dphase_offset", b["dL_dphase_offset"], grads["phase_offset"], rtol_max=rtol, atol=1e-5)
            Hh.assert_close("dL_ddc_offset", b["dL_ddc_offset"], grads["dc_offset"], rtol_max=rtol, atol=1e-5)
    if g.get("scales") is not None:
        Hh.assert_close("dL_dscales", b["dL_dscales"], grads["scales"], rtol_max=rtol)
        Hh.assert_close("dL_drotations", b["dL_drotations"], grads["rotations"], rtol_max=rtol)
    if g.get("colors_precomp") is not None:
        Hh.assert_close("dL_dcolors", b["dL_dcolors"], grads["colors_precomp"], rtol_max=rtol)
    if g.get("cov3D_precomp") is not None:
        Hh.assert_close("dL_dcov3D", b["dL_dcov3D"], grads["cov3D_precomp"], rtol_max=rtol)


# ---- operator variants (argument patterns of gaussian_renderer/__init__.py) -----
@pytest.mark.parametrize("D,M", [(0, 16), (1, 16), (2, 16), (3, 16), (0, 1), (1, 4), (2, 9)])
def test_sh_degrees(D, M, oracle, gpu):
    scene = Hh.small_scene(D=D, sh_coeffs=M, P=250)
    f, b = Hh.run_oracle(oracle, scene)
    out, grads, _ = Hh.run_gpu(scene, gpu, optimize_offsets=True)
    check_outputs(f, out)
    check_grads(b, grads, scene)
    if M > (D + 1) ** 2:   # inactive coefficients receive exact zeros
        assert not grads["shs"][:, (D + 1) ** 2:, :].any()
        assert not grads["shs_p"][:, (D + 1) ** 2:, :].any()


def test_colors_precomp_and_no_tof(oracle, gpu):
    """render_flow pattern (gaussian_renderer/__init__.py:194-202): precomputed colours,
    no shs_p / phasors -> phasor planes are background only."""
    scene = Hh.small_scene(P=300, tof=False)
    rng = np.random.default_rng(5)
    inputs = dict(shs=None, colors_precomp=rng.random((300, 3)).astype(np.float32))
    f, b = Hh.run_oracle(oracle, scene, inputs=inputs)
    out, grads, _ = Hh.run_gpu(scene, gpu, inputs=inputs)
    # the reference leaves real_img_amp uninitialised here; both sides define it as zero
    for k in ["color", "depth", "acc", "depth_distortion"]:
        Hh.assert_close(k, f[k], out[k], rtol_max=IMG_MAX, atol=1e-6, frac_bad=1e-3)
    s2 = dict(scene)
    s2["gaussians"] = dict(scene["gaussians"], **inputs)
    check_grads(b, grads, s2)


def test_cov3d_precomp(oracle, gpu):
    scene = Hh.small_scene(P=300)
    f0, _ = Hh.run_oracle(oracle, scene, backward=False)
    inputs = dict(scales=None, rotations=None, cov3D_precomp=f0.geom["cov3D"].copy())
    # culled rows were never written by the oracle: give them a valid covariance
    inputs["cov3D_precomp"][f0.radii <= 0] = np.array([1e-4, 0, 0, 1e-4, 0, 1e-4], np.float32)
    f, b = Hh.run_oracle(oracle, scene, inputs=inputs)
    out, grads, _ = Hh.run_gpu(scene, gpu, inputs=inputs, optimize_offsets=True)
    check_outputs(f, out)
    s2 = dict(scene)
    s2["gaussians"] = dict(scene["gaussians"], **inputs)
    check_grads(b, grads, s2)


def test_phasors_precomp(oracle, gpu):
    scene = Hh.small_scene(P=300)
    rng = np.random.default_rng(9)
    pp = np.stack([rng.uniform(-0.5, 0.5, 300), rng.uniform(0.05, 0.5, 300)], 1).astype(np.float32)
    inputs = dict(shs_p=None, phasors_precomp=pp)
    f, b = Hh.run_oracle(oracle, scene, inputs=inputs)
    out, grads, _ = Hh.run_gpu(scene, gpu, inputs=inputs)
    check_outputs(f, out)


@pytest.mark.parametrize("vdp", [False, True])
def test_view_dependent_phase_flag(vdp, oracle, gpu):
    scene = Hh.small_scene(P=300)
    scene["use_view_dependent_phase"] = vdp
    f, b = Hh.run_oracle(oracle, scene)
    out, grads, _ = Hh.run_gpu(scene, gpu, optimize_offsets=True)
    check_outputs(f, out)
    check_grads(b, grads, scene)


def test_expanded_background_and_float_offsets(oracle, gpu):
    """train.py:127 builds bg as a 7-vector .view(7,1,1).expand(7,H,W): consumed via strides."""
    scene = Hh.small_scene(P=300)
    W, H = scene["cfg"]["W"], scene["cfg"]["H"]
    vec = np.array([0.2, -0.4, 0.6, 0.1, -0.3, 0.5, -0.7], np.float32)
    scene["bg"] = np.ascontiguousarray(np.broadcast_to(vec[:, None, None], (7, H, W)))
    f, b = Hh.run_oracle(oracle, scene)
    bg_t = torch.tensor(vec, device=gpu).view(7, 1, 1).expand(7, H, W)
    assert not bg_t.is_contiguous()
    out, grads, _ = Hh.run_gpu(scene, gpu, bg=bg_t)
    check_outputs(f, out)
    check_grads(b, grads, scene)


def test_edge_cases(oracle, gpu):
    from gftorf_amd import GaussianRasterizer
    scene = Hh.small_scene(P=64)
    # all culled (behind the far plane): background only, zero gradients
    far = dict(scene)
    far["gaussians"] = dict(scene["gaussians"])
    far["gaussians"]["means3D"] = scene["gaussians"]["means3D"] + np.array([0, 0, 100], np.float32)
    f, b = Hh.run_oracle(oracle, far)
    out, grads, _ = Hh.run_gpu(far, gpu)
    assert f.num_rendered == 0 and not out["radii"].any()
    np.testing.assert_allclose(out["color"], scene["bg"][:3], rtol=0, atol=0)
    np.testing.assert_allclose(out["phasor"], scene["bg"][:7], rtol=0, atol=0)
    for k in ["means3D", "shs", "shs_p", "scales", "rotations", "opacities", "means2D"]:
        assert not grads[k].any(), k
    # prefiltered=True promises that no point is culled; a culled one is an error (device trap in the reference,
    # auxiliary.h:171-175), in the two-stage flow and in the one-call flow (the flag travels through the host mailbox)
    from gftorf_amd import api
    key = (gpu.index, 64, scene["cfg"]["W"], scene["cfg"]["H"])
    for flow in ("two-stage", "one-call"):
        api._instance_hint.pop(key, None)
        if flow == "one-call":
            api._instance_hint[key] = (1000, 0)
        with pytest.raises(RuntimeError, match="prefiltered"):
            Hh.run_gpu(far, gpu, backward=False, prefiltered=True)
    Hh.run_gpu(scene, gpu, backward=False, prefiltered=True)      # nothing culled: fine
    # P == 0: the reference returns zero images (rasterize_points.cu:104)
    empty = dict(scene)
    empty["gaussians"] = {k: (v[:0] if v is not None else None) for k, v in scene["gaussians"].items()}
    out, grads, _ = Hh.run_gpu(empty, gpu)
    assert out["color"].shape == (3, scene["cfg"]["H"], scene["cfg"]["W"]) and not out["color"].any()
    assert out["radii"].shape == (0,) and grads["means3D"].shape == (0, 3)
    # single Gaussian
    one = dict(scene)
    one["gaussians"] = {k: (v[:1] if v is not None else None) for k, v in scene["gaussians"].items()}
    f, b = Hh.run_oracle(oracle, one)
    out, grads, _ = Hh.run_gpu(one, gpu)
    check_outputs(f, out)
    check_grads(b, grads, one)
    # argument validation mirrors the reference messages
    st = Hh.gpu_settings(scene, gpu)
    r = GaussianRasterizer(st)
    m = torch.zeros(4, 3, device=gpu)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(means3D=m, means2D=m, opacities=m[:, :1], scales=m, rotations=torch.zeros(4, 4, device=gpu))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
        r(means3D=m, means2D=m, opacities=m[:, :1], shs=torch.zeros(4, 16, 3, device=gpu))
    vis = r.markVisible(torch.tensor(scene["gaussians"]["means3D"], device=gpu))
    ref = oracle.mark_visible(scene["gaussians"]["means3D"], scene["cam"]["viewmatrix"],
                              scene["cam"]["projmatrix"], scene["cam"]["znear"], scene["cam"]["zfar"])
    np.testing.assert_array_equal(vis.cpu().numpy(), ref)


def test_two_calls_accumulate_on_shared_means2d(oracle, gpu):
    """gaussian_renderer/__init__.py:107-128 calls the rasterizer twice on the same
    screenspace_points: gradients accumulate; eval runs under no_grad."""
    from gftorf_amd import GaussianRasterizer
    scene = Hh.small_scene(P=200)
    st = Hh.gpu_settings(scene, gpu)
    g = {k: torch.tensor(v, device=gpu, requires_grad=True) for k, v in scene["gaussians"].items()}
    m2d = torch.zeros(200, 3, device=gpu, requires_grad=True)
    r = GaussianRasterizer(st)
    kw = dict(means3D=g["means3D"], means2D=m2d, opacities=g["opacities"], shs=g["shs"], shs_p=g["shs_p"],
              scales=g["scales"], rotations=g["rotations"], phase_offset=0.1, dc_offset=0.05)
    o1 = r(**kw)
    o2 = r(**kw)
    (o1[0].sum() + o2[1].sum()).backward()
    a = m2d.grad.clone()
    m2d.grad = None
    for v in g.values():
        v.grad = None
    o1 = r(**kw)
    o1[0].sum().backward()
    b1 = m2d.grad.clone()
    m2d.grad = None
    o2 = r(**kw)
    o2[1].sum().backward()
    torch.testing.assert_close(a, b1 + m2d.grad, rtol=1e-4, atol=1e-5)
    with torch.no_grad():
        o3 = r(**kw)
    torch.testing.assert_close(o3[0], o1[0].detach(), rtol=0, atol=0)


def test_one_call_forward_matches_two_stage(oracle, gpu):
    """gft_forward (binning buffer sized from the previous frame, no host round trip) must give
    the results of the two-stage flow bit for bit: with headroom, with an exact fit, and when the
    guess is too small (stage-2 kernels skip themselves, the host re-runs stage 2)."""
    from gftorf_amd import api
    scene = Hh.small_scene(P=3000, seed=11)
    api._instance_hint.clear()
    ref_out, ref_grads, _ = Hh.run_gpu(scene, gpu)               # first frame: two-stage flow
    R = api.last_call_stats["num_rendered"]
    assert R > 0 and api.last_call_stats["binning_instances"] == (R + 63) // 64 * 64 and not api.last_call_stats["restarted"]
    key = next(iter(api._instance_hint))
    # (instance guess, restart expected)
    for hint, restarted in ((None, False), (R, False), (1, True), (0, True), (3 * R, False)):
        if hint is not None:
            api._instance_hint[key] = (hint, api._instance_hint[key][1])
        out, grads, _ = Hh.run_gpu(scene, gpu)
        st = api.last_call_stats
        assert st["num_rendered"] == R and st["restarted"] == restarted
        assert st["binning_instances"] >= R
        for k in ref_out:
            np.testing.assert_array_equal(out[k], ref_out[k], err_msg=k)
        for k in ref_grads:
            if ref_grads[k] is not None:
                # atomics: summation order differs between runs
                Hh.assert_close(k, ref_grads[k], grads[k], rtol_max=1e-5)
    # a frame with far fewer instances than the guess
    small = Hh.small_scene(P=3000, seed=11)
    small["gaussians"]["opacities"] = np.full_like(small["gaussians"]["opacities"], 1e-4)   # all culled (alpha < 1/255)
    f, b = Hh.run_oracle(oracle, small)
    out, grads, _ = Hh.run_gpu(small, gpu)
    check_outputs(f, out)


def test_backward_twice_through_one_forward(gpu):
    """The forward clears the backward's accumulator (side stream); a second backward through the
    same forward (retain_graph) must clear its own and give the same gradients."""
    from gftorf_amd import GaussianRasterizer
    scene = Hh.small_scene(P=400, seed=5)
    st = Hh.gpu_settings(scene, gpu)
    g = {k: torch.tensor(v, device=gpu, requires_grad=True) for k, v in scene["gaussians"].items()}
    m2d = torch.zeros(400, 3, device=gpu, requires_grad=True)
    out = GaussianRasterizer(st)(means3D=g["means3D"], means2D=m2d, opacities=g["opacities"], shs=g["shs"],
                                 shs_p=g["shs_p"], scales=g["scales"], rotations=g["rotations"],
                                 phase_offset=0.1, dc_offset=0.05)
    loss = out[0].sum() + out[1].sum()
    loss.backward(retain_graph=True)
    first = {k: v.grad.clone() for k, v in g.items()}
    for v in g.values():
        v.grad = None
    loss.backward()
    for k, v in g.items():
        torch.testing.assert_close(v.grad, first[k], rtol=1e-4, atol=1e-6)


def test_one_call_forward_with_wrong_list_guess(oracle, gpu):
    """One-call flow on a frame whose longest tile list is over the short-sort limit while the
    caller's guess (previous frame) said it would not be: the library sorts the long lists and
    renders again; outputs, counters and gradients equal the two-stage flow."""
    from gftorf_amd import _lib, api
    scene = Hh.small_scene(**SCENES["long_lists_lds128k"])
    _lib.load().gft_set_binning_mode(0)           # whole-frame binning (the long-list sort belongs to it); reset below
    api._instance_hint.clear()
    ref_out, ref_grads, _ = Hh.run_gpu(scene, gpu)               # two-stage flow
    R, longest = api.last_call_stats["num_rendered"], api.last_call_stats["max_tile_list"]
    assert longest > 4096
    key = next(iter(api._instance_hint))
    for list_guess in (100, longest):                             # wrong guess, right guess
        api._instance_hint[key] = (R, list_guess)
        out, grads, _ = Hh.run_gpu(scene, gpu)
        assert api.last_call_stats["num_rendered"] == R and not api.last_call_stats["restarted"]
        for k in ref_out:
            np.testing.assert_array_equal(out[k], ref_out[k], err_msg=k)
        for k in ref_grads:
            if ref_grads[k] is not None:
                Hh.assert_close(k, ref_grads[k], grads[k], rtol_max=1e-5)
    _lib.load().gft_set_binning_mode(-1)
    f, b = Hh.run_oracle(oracle, scene)
    check_outputs(f, ref_out)


BIN_CASES = {
    # frames where no / every / some quadrant walks past the sorted head of its list
    "base": dict(),
    "deep_lists": SCENES["deep_lists"],
    "opaque_early_exit": SCENES["opaque_early_exit"],              # most quadrants saturate inside the head
    "long_lists": SCENES["long_lists_lds128k"],                    # lists of thousands of keys, heads of ~940
    "thin_fog": dict(P=40000, W=64, H=64, scale_lo=0.01, scale_hi=0.05, opacity=0.02),   # nothing saturates: every list is completed
    "flat_depth": SCENES["long_lists_flat"],                       # every key in one depth bin: empty heads, the lists are built on demand
    "global_tail": SCENES["long_lists_global"],                    # a culled tail larger than the LDS sorter: bin ranges
    "silhouette": dict(P=30000, W=96, H=64, scale_lo=0.005, scale_hi=0.05, spread=0.55),   # cloud edge inside the image: quadrants that never saturate beside dense ones
    "wide_depth_range": SCENES["wide_depth_range"],                # whole lists over more depth bins than the pull kernel has cursors
    "crowded_depth_bins": SCENES["crowded_depth_bins"],            # depth bins with more keys than a cursor counts: sorted as a whole
    # thin lists of ~1500 entries over an image that ends inside its last quadrants: segments with the backward's cuts
    # inside them, blended speculatively, with lanes outside the image (their snapshots must not carry a zero transmittance)
    "ragged_thin_deep": dict(P=6000, W=50, H=37, scale_lo=0.03, scale_hi=0.2, opacity=0.04),
    # a tile grid of 2040 (supertile, slab) cells: the scatter pass groups a workgroup's entries by cell in LDS -- all of them
    # (small splats: a few thousand entries per workgroup) or the first ~13.9 k with the rest written directly (large splats:
    # ~45 supertiles per Gaussian)
    "big_grid_small_splats": dict(P=20000, W=1920, H=1080, scale_lo=0.003, scale_hi=0.02),
    "big_grid_large_splats": dict(P=6000, W=1920, H=1080, scale_lo=0.02, scale_hi=0.12, opacity=0.2),
}


@contextlib.contextmanager
def binning_mode(mode):
    from gftorf_amd import _lib
    lib = _lib.load()
    lib.gft_set_binning_mode(int(mode))
    try:
        yield
    finally:
        lib.gft_set_binning_mode(-1)


@contextlib.contextmanager
def render_mode(mode):
    """0: one wave per quadrant, 1: segment-parallel forward wherever the tile count allows, -1: the library's default"""
    from gftorf_amd import _lib
    lib = _lib.load()
    lib.gft_set_render_mode(int(mode))
    try:
        yield
    finally:
        lib.gft_set_render_mode(-1)


@pytest.mark.parametrize("name", list(BIN_CASES))
def test_segmented_forward_matches_one_wave_per_quadrant(name, oracle, gpu):
    """The segment-parallel forward (k_render_fwd_seg: several waves per quadrant, transmittance factors exchanged, exact
    blend from the true transmittance) against the serial walk of one wave per quadrant: discrete results equal (radii,
    contributor counts up to pixels that stand within rounding of the 1e-4 stop), images to fp32 rounding of the
    transmittance products, gradients -- which start from the snapshots the forward leaves at the backward's cuts -- to
    the order of their sums.  Frames where no / every / some quadrant walks past its sorted head."""
    from gftorf_amd import api
    scene = Hh.small_scene(seed=23, **BIN_CASES[name])
    res = {}
    for mode in (0, 1):
        with render_mode(mode):
            api._instance_hint.clear()
            for frame in range(2):                 # two-stage flow, then one call
                res[mode] = Hh.run_gpu(scene, gpu)[:2]
    (o0, g0), (o1, g1) = res[0], res[1]
    np.testing.assert_array_equal(o0["radii"], o1["radii"])
    assert float((o0["pixels"] != o1["pixels"]).mean()) < 1e-4
    for k in ["color", "phasor", "depth", "acc", "depth_distortion", "distribution"]:
        Hh.assert_close(k, o0[k], o1[k], rtol_max=2e-6, atol=1e-7, frac_bad=2e-4, rtol_elem=2e-6)
    for k in g0:
        if g0[k] is not None:
            Hh.assert_close(k, g0[k], g1[k], rtol_max=2e-5, atol=1e-7, frac_bad=1e-4, rtol_elem=2e-5)
    f, b = Hh.run_oracle(oracle, scene)
    check_outputs(f, o1)
    check_grads(b, g1, scene)


@pytest.mark.parametrize("name", list(BIN_CASES))
def test_tile_pull_matches_whole_frame_binning(name, oracle, gpu):
    """Tile-pull binning (heads pulled per tile, appearance on demand, lists completed and culled for flagged quadrants)
    against whole-frame binning (every instance counted, keyed and sorted): every output bit-identical, in the first
    frame of a shape (two-stage flow) and in the following ones (one call, buffer sized from the previous frame);
    gradients equal up to the order of the atomic sums; and the whole thing against the oracle."""
    from gftorf_amd import _lib, api
    if not _lib.load().gft_lazy_sort():
        pytest.skip("GFT_LAZY_SORT=0: whole-frame binning only")
    scene = Hh.small_scene(seed=23, **BIN_CASES[name])
    # (one wave per quadrant in both: the segment-parallel forward cuts a list by its length, and the sorted heads of the
    # two binning modes differ in length -- the sums would agree to rounding, not bit for bit)
    with render_mode(0):
        with binning_mode(0):
            api._instance_hint.clear()
            ref_out, ref_grads, _ = Hh.run_gpu(scene, gpu)
            R = api.last_call_stats["num_rendered"]
        api._instance_hint.clear()
        for frame in range(3):
            out, grads, _ = Hh.run_gpu(scene, gpu)
            st = api.last_call_stats
            assert st["num_rendered"] == R and not st["restarted"]
            for k in ref_out:
                np.testing.assert_array_equal(out[k], ref_out[k], err_msg="%s frame %d" % (k, frame))
            for k in ref_grads:
                if ref_grads[k] is not None:
                    # (sums in another order of the atomics: the 1080p cases' rotation rows were seen at 1.4e-5)
                    Hh.assert_close(k, ref_grads[k], grads[k], rtol_max=3e-5 if name.startswith("big_grid") else 1e-5)
    f, b = Hh.run_oracle(oracle, scene)
    check_outputs(f, ref_out)
    check_grads(b, grads, scene)


def test_tile_pull_structure(gpu, oracle):
    """What the tile-pull forward leaves behind on frames that exercise its branches: short lists sorted whole and never
    flagged; long lists with a head of about 940 ids; quadrants that walk past their head flagged and their tile's list
    completed in the pool; Gaussians outside every sorted part without an appearance."""
    from gftorf_amd import _lib
    if not _lib.load().gft_lazy_sort():
        pytest.skip("GFT_LAZY_SORT=0: whole-frame binning only")
    fog = Hh.small_scene(seed=21, P=40000, W=64, H=64, scale_lo=0.01, scale_hi=0.05, opacity=0.02)
    st = raw_forward(fog, gpu, mode=1)
    f, _ = Hh.run_oracle(oracle, fog, backward=False)
    lens = f.ranges[:, 1] - f.ranges[:, 0]
    long_tiles = lens > 2048
    assert long_tiles.any() and st["pull"]
    assert ((st["front_len"][long_tiles] >= 512) & (st["front_len"][long_tiles] <= 2048)).all()
    flagged = st["unit_flag"].any(1)
    assert flagged[long_tiles].all() and int(st["ctrl"][4]) == int((st["unit_flag"] != 0).sum())      # nothing saturates in fog
    assert (st["ranges"][long_tiles, 0] >= lens.size * 2048).all()                                # completed lists live in the pool
    assert int(st["ctrl"][6]) == int((st["ranges"][flagged, 1] - st["ranges"][flagged, 0]).sum())  # pool slots taken
    opaque = Hh.small_scene(seed=21, P=30000, W=64, H=48, scale_lo=0.02, scale_hi=0.1, opacity=0.9)
    st = raw_forward(opaque, gpu, mode=1)
    f, _ = Hh.run_oracle(oracle, opaque, backward=False)
    lens = f.ranges[:, 1] - f.ranges[:, 0]
    assert (lens > 2048).any() and not st["unit_flag"].any() and int(st["ctrl"][6]) == 0          # all saturate early: no list is completed
    vis = f.geom["radii"] > 0
    assert 0 < (st["need"] != 0).sum() < 0.6 * vis.sum()                                          # most Gaussians never get an appearance
    assert (st["ranges"][:, 1] - st["ranges"][:, 0] == st["front_len"]).all()


def test_hinted_tiles_sort_their_whole_list(gpu, oracle):
    """gft_forward_io.tile_hints: a tile whose word is non-zero and whose list is longer than one placement sorts it WHOLE
    in k_tile_pull (chunks of whole depth bins, into the pool): the list is the oracle's, no quadrant flags, nothing is
    completed on demand; the forward leaves the next frame's words -- per quadrant, did it walk past where a head ends."""
    from gftorf_amd import _lib
    if not _lib.load().gft_lazy_sort():
        pytest.skip("GFT_LAZY_SORT=0: whole-frame binning only")
    # (one wave per quadrant throughout: the segment-parallel forward cuts a list by the length of its sorted part, which is
    # what the schedule changes -- its sums would agree to rounding, not bit for bit)
    with render_mode(0):
        fog = Hh.small_scene(seed=21, P=40000, W=64, H=64, scale_lo=0.01, scale_hi=0.05, opacity=0.02)
        f, _ = Hh.run_oracle(oracle, fog, backward=False)
        lens = (f.ranges[:, 1] - f.ranges[:, 0]).astype(np.int64)
        T = lens.size
        long_tiles = lens > 2048
        assert long_tiles.any()
        # no schedule yet (all zero): the lazy route, and the words it leaves say "every long tile walked past its head"
        h = torch.zeros((T,), device=gpu, dtype=torch.int32)
        st0 = raw_forward(fog, gpu, mode=1, hints=h)
        assert st0["pull"] and st0["unit_flag"].any(1)[long_tiles].all()
        assert (st0["hints"][long_tiles] != 0).all() and not st0["hints"][~long_tiles].any()
        # with that schedule
        st = raw_forward(fog, gpu, mode=1, hints=h)
        assert not st["unit_flag"].any() and int(st["ctrl"][4]) == 0                   # nobody flags
        assert (st["tile_cut"][long_tiles] == 0xffffffff).all()
        np.testing.assert_array_equal(st["front_len"], lens)                           # every list complete after the pull kernel
        assert (st["ranges"][long_tiles, 0] >= T * 2048).all()                         # whole lists live in the pool
        assert int(st["ctrl"][6]) == int(lens[long_tiles].sum())                       # pool slots taken = the whole lists
        for t in range(T):
            a, e = int(st["ranges"][t, 0]), int(st["ranges"][t, 1])
            np.testing.assert_array_equal(st["point_list"][a:e], f.point_list[int(f.ranges[t, 0]):int(f.ranges[t, 1])], err_msg="tile %d" % t)
        assert (st["hints"][long_tiles] != 0).any(1).all()                             # still nothing saturates: the schedule stays
        np.testing.assert_array_equal(st["planes"], st0["planes"])
        np.testing.assert_array_equal(st["pixels"], st0["pixels"])
        # a frame that saturates early under the same (now wrong) schedule: whole lists all the same, and the words go back to 0
        opaque = Hh.small_scene(seed=21, P=30000, W=64, H=64, scale_lo=0.02, scale_hi=0.1, opacity=0.9)
        fo, _ = Hh.run_oracle(oracle, opaque, backward=False)
        lo = (fo.ranges[:, 1] - fo.ranges[:, 0]).astype(np.int64)
        assert (lo > 2048).any()
        h.fill_(0x01010101)
        so = raw_forward(opaque, gpu, mode=1, hints=h)
        np.testing.assert_array_equal(so["front_len"], lo)
        assert not so["hints"].any() and not so["unit_flag"].any()
        ref = raw_forward(opaque, gpu, mode=1)
        np.testing.assert_array_equal(so["planes"], ref["planes"])
        np.testing.assert_array_equal(so["pixels"], ref["pixels"])
        # a single depth bin with more keys than the bin counts carry (bytes): the hinted tile takes the lazy route
        flat = Hh.small_scene(seed=23, **SCENES["long_lists_flat"])
        h2 = torch.full((4,), 0x01010101, device=gpu, dtype=torch.int32)
        sf = raw_forward(flat, gpu, mode=1, hints=h2)
        ref = raw_forward(flat, gpu, mode=1)
        np.testing.assert_array_equal(sf["front_len"], ref["front_len"])
        np.testing.assert_array_equal(sf["planes"], ref["planes"])


@pytest.mark.parametrize("name", list(BIN_CASES))
def test_tile_hints_do_not_change_results(name, oracle, gpu):
    """The schedules the operator keeps per camera from frame to frame (api.state.cameras -> gft_forward_io.tile_hints, ...) against
    no schedule, an all-ones one and a random one: every output bit-identical (the blend walks the same entries in the
    same order whether a list was sorted whole up front or head first / rest on demand), gradients equal up to the order
    of the atomic sums."""
    from gftorf_amd import _lib, api
    if not _lib.load().gft_lazy_sort():
        pytest.skip("GFT_LAZY_SORT=0: whole-frame binning only")
    scene = Hh.small_scene(seed=23, **BIN_CASES[name])
    rng = np.random.default_rng(5)
    with render_mode(0):
        keep = api._TILE_HINTS
        try:
            api._TILE_HINTS = False
            api._instance_hint.clear()
            Hh.run_gpu(scene, gpu)
            ref_out, ref_grads, _ = Hh.run_gpu(scene, gpu)
        finally:
            api._TILE_HINTS = keep
        api.state.reset_schedules()
        for frame in range(7):
            if frame >= 3:
                # frames 0 - 2: the schedule and the choice of the pull kernel's build as the operator makes them; then forced:
                # all ones, random, random with the whole-list build, random with the heads-only build (which ignores it)
                cams = list(api.state.cameras.values())
                for hbuf in (c.tile_hints for c in cams if c.tile_hints is not None):
                    if frame == 3:
                        hbuf.fill_(0x01010101)
                    else:
                        hbuf.copy_(torch.tensor(rng.integers(0, 2, hbuf.numel()), dtype=torch.int32))
                # ... and whatever walk lengths the forward's heavy-first dealing is derived from
                for wbuf in (c.tile_weights for c in cams if c.tile_weights is not None):
                    wbuf.copy_(torch.tensor(rng.integers(0, 3000 if frame != 4 else 2 ** 31 - 1, wbuf.numel()), dtype=torch.int32))
                    wbuf[-4] = 1
                api._force_whole_lists = frame != 6
                # ... and the camera's list schedule (where the binning's scatter pass appends without a count pass in front):
                # garbage, zeros, capacities of 1 -- found on the device, the counted flow runs in the same call
                for sbuf in (c.cell_sched for c in cams):
                    if sbuf is False or sbuf is None:
                        continue
                    if frame == 4:
                        sbuf.copy_(torch.tensor(rng.integers(0, 2 ** 31 - 1, sbuf.numel()), dtype=torch.int32))
                    elif frame == 5:
                        sbuf.zero_()
                    elif frame == 6:
                        cells = (sbuf.numel() - 4) // 2
                        sbuf[:cells] = torch.arange(cells, dtype=torch.int32, device=sbuf.device)
                        sbuf[cells:2 * cells] = 1
                        sbuf[2 * cells] = cells
                        sbuf[2 * cells + 1] = 1
                api._force_cell_sched = True if frame in (4, 5, 6) else None
            misses = api.last_call_stats.get("sched_misses", 0)
            out, grads, _ = Hh.run_gpu(scene, gpu)
            scheds = [c.cell_sched for c in api.state.cameras.values()]
            if scheds and not any(v is False or v is None for v in scheds) and api._CELL_SCHED:
                if frame in (4, 5, 6):
                    assert api.last_call_stats.get("sched_misses", 0) > misses, frame
                elif frame in (1, 2, 3):
                    assert api.last_call_stats.get("sched_misses", 0) == misses, frame
            for k in ref_out:
                np.testing.assert_array_equal(out[k], ref_out[k], err_msg="%s frame %d" % (k, frame))
            for k in ref_grads:
                if ref_grads[k] is not None:
                    # (the same terms in another order of the atomics: 1e-5 of the max-norm on the small frames; the 1080p
                    # cases' rotation rows -- long sums that cancel -- were seen at 1.4e-5)
                    Hh.assert_close(k, ref_grads[k], grads[k], rtol_max=3e-5 if name.startswith("big_grid") else 1e-5)
    api._force_whole_lists = None
    api._force_cell_sched = None
    f, b = Hh.run_oracle(oracle, scene)
    check_outputs(f, out)
    check_grads(b, grads, scene)


def test_list_schedule_fits_small_scenes(oracle, gpu):
    """A scene of few, small Gaussians on a grid of many (supertile, slab) lists: every list's reserve of `count + count / 4 +
    64` entries adds up to more than the binning buffer's 4096 spare instances -- the operator sizes the buffer for the
    schedule's slack too, so the camera's second and later frames bin by the schedule instead of missing it every time."""
    from gftorf_amd import _lib, api
    if not _lib.load().gft_lazy_sort() or not api._CELL_SCHED:
        pytest.skip("no list schedule in this configuration")
    scene = Hh.small_scene(P=2500, W=640, H=480, seed=9, scale_lo=0.002, scale_hi=0.004)
    f, b = Hh.run_oracle(oracle, scene)
    keep = api._TILE_HINTS_PER_CAMERA
    api._TILE_HINTS_PER_CAMERA = False                 # (run_gpu builds new camera tensors per call: one schedule per image size)
    api._instance_hint.clear()
    api.state.reset_schedules()
    try:
        with render_mode(0):
            for frame in range(5):
                misses = api.last_call_stats.get("sched_misses", 0)
                out, grads, _ = Hh.run_gpu(scene, gpu)
                check_outputs(f, out)
                if frame >= 2:
                    assert api.last_call_stats.get("sched_misses", 0) == misses, frame
            cam = next(iter(api.state.cameras.values()))
            assert cam.sched_seen and cam.cell_sched is not None and cam.cell_sched is not False and cam.sched_off_until == 0
        check_grads(b, grads, scene)
    finally:
        api._TILE_HINTS_PER_CAMERA = keep
        api._instance_hint.clear()
        api.state.reset_schedules()


def test_schedules_are_kept_per_camera(oracle, gpu):
    """Two cameras on the same Gaussians, drawn alternately (a training loop's access pattern): each keeps its own per-tile
    schedule, keyed by the address of its view matrix -- the silhouette tiles of one view are not the other's.  A thin cloud
    whose edge crosses the image: some quadrants walk past a normal head.  With the schedule, such a tile gets a long head
    (or its whole list) at the camera's next visit and flags less; results equal the schedule-free ones bit for bit."""
    from gftorf_amd import GaussianRasterizer, _lib, api, synth
    if not _lib.load().gft_lazy_sort():
        pytest.skip("GFT_LAZY_SORT=0: whole-frame binning only")
    kw = dict(P=30000, W=96, H=64, scale_lo=0.005, scale_hi=0.05, spread=0.55)
    cams = [synth.look_at_w2c(0.15, -0.1, 0.05, (0.1, -0.05, 0.2)), synth.look_at_w2c(-0.2, 0.12, -0.04, (-0.15, 0.05, 0.25))]
    scenes = [Hh.small_scene(seed=23, w2c=c, **kw) for c in cams]
    g = scenes[0]["gaussians"]
    leaf = {k: torch.tensor(v, dtype=torch.float32, device=gpu, requires_grad=True) for k, v in g.items() if v is not None}
    m2 = torch.zeros((kw["P"], 3), device=gpu, requires_grad=True)
    rasts = [GaussianRasterizer(raster_settings=Hh.gpu_settings(sc, gpu)) for sc in scenes]       # persistent camera tensors

    def render(i):
        for v in leaf.values():
            v.grad = None
        sc = scenes[i]
        o = rasts[i](means3D=leaf["means3D"], means2D=m2, opacities=leaf["opacities"], shs=leaf["shs"], shs_p=leaf["shs_p"],
                     scales=leaf["scales"], rotations=leaf["rotations"], phase_offset=sc["phase_offset"], dc_offset=sc["dc_offset"])
        gr = sc["grads"]
        sum((o[j] * torch.tensor(gr[k], device=gpu)).sum() for j, k in ((0, "color"), (1, "phasor"), (2, "depth"), (4, "acc"), (6, "depth_distortion"))).backward()
        torch.cuda.synchronize()
        bufs = api.last_call_buffers
        L = _lib.get_layout(bufs["P"], bufs["W"], bufs["H"], bufs["cap"])
        flagged = int(bufs["img"][L.img_ctrl:L.img_ctrl + 64].view(torch.int32)[4].item())
        return [t.detach().cpu().numpy() for t in o], leaf["means3D"].grad.cpu().numpy(), flagged

    keep = (api._TILE_HINTS, api._TILE_HINTS_PER_CAMERA, api.keep_last_buffers)
    api.keep_last_buffers = True
    try:
        with render_mode(0):
            api._TILE_HINTS = False
            api._instance_hint.clear()
            render(0), render(1)
            ref = [render(0), render(1)]
            assert ref[0][2] > 0 and ref[1][2] > 0, "the case must have quadrants that walk past their heads"
            api._TILE_HINTS, api._TILE_HINTS_PER_CAMERA = True, True
            api.state.reset_schedules()
            flagged = []
            for visit in range(3):
                for i in (0, 1):
                    o, gm, fl = render(i)
                    flagged.append(fl)
                    for a, e in zip(o, ref[i][0]):
                        np.testing.assert_array_equal(a, e)
                    Hh.assert_close("means3D", ref[i][1], gm, rtol_max=1e-5)
            keys = [k for k in api.state.cameras if k[1:3] == (96, 64)]
            assert len(keys) == 2 and len({k[4] for k in keys}) == 2                  # one schedule per camera
            a, b = (api.state.cameras[k].tile_hints.cpu().numpy() for k in keys)
            assert (a != 0).any() and (b != 0).any() and ((a != 0) != (b != 0)).any()    # ... and they differ
            # first visits: no schedule yet; later visits: the marked tiles got long heads and flag less
            assert flagged[0] == ref[0][2] and flagged[1] == ref[1][2]
            assert flagged[2] < flagged[0] and flagged[3] < flagged[1], flagged
    finally:
        api._TILE_HINTS, api._TILE_HINTS_PER_CAMERA, api.keep_last_buffers = keep
        api.last_call_buffers.clear()
        api._instance_hint.clear()
    f, b = Hh.run_oracle(oracle, dict(scenes[1], gaussians=g), backward=False)      # (camera 1 on the shared Gaussians)
    check_outputs(f, dict(zip(Hh.OUT_NAMES, o)))


LAZY_CASES = {
    # lists of 1.2k..5k keys per tile: the sorted head (~940 keys) is not enough for the far pixels
    "thin_fog": dict(P=40000, W=64, H=64, scale_lo=0.01, scale_hi=0.05, opacity=0.02),            # nothing saturates: every quadrant resumes
    "mixed": dict(P=30000, W=96, H=64, scale_lo=0.005, scale_hi=0.08),                             # some quadrants saturate in the head
    "dense_opaque": dict(P=30000, W=64, H=48, scale_lo=0.02, scale_hi=0.1, opacity=0.9),           # all saturate early: no tail is ever sorted
    "one_long_tile": dict(P=6000, W=16, H=16, scale_lo=0.02, scale_hi=0.06, opacity=0.05, spread=0.3),
}


@pytest.mark.parametrize("name", list(LAZY_CASES))
def test_lazy_sort_resume_paths(name, oracle, gpu):
    """Head-first tile sort (k_tile_front / k_tile_tail / resumed quadrants): images, counters and
    gradients against the oracle, and the structure of the id lists, on frames that exercise
    'never resumes', 'always resumes' and the mix."""
    scene = Hh.small_scene(seed=21, **LAZY_CASES[name])
    f, b = Hh.run_oracle(oracle, scene)
    st = raw_forward(scene, gpu, mode=0)                      # whole-frame binning: k_tile_front / k_tile_tail
    lens = f.ranges[:, 1] - f.ranges[:, 0]
    if st["lazy"]:
        long_tiles = lens > 1024
        assert long_tiles.any(), "case does not reach the lazy path"
        flagged = st["unit_flag"].any(1)
        assert int(st["ctrl"][4]) == int((st["unit_flag"] != 0).sum())
        if name == "thin_fog":
            assert flagged[long_tiles].all()
        if name == "dense_opaque":
            assert not flagged.any()
        for t in np.nonzero(long_tiles)[0]:
            a, e = int(f.ranges[t, 0]), int(f.ranges[t, 1])
            k = int(st["front_len"][t])
            assert 0 < k <= 1024
            np.testing.assert_array_equal(st["point_list"][a:a + k], f.point_list[a:a + k])
            if flagged[t]:
                np.testing.assert_array_equal(st["point_list"][a:e], f.point_list[a:e])
    for mode in (0, 1):
        with binning_mode(mode):
            out, grads, _ = Hh.run_gpu(scene, gpu)
        check_outputs(f, out)
        check_grads(b, grads, scene)


@pytest.mark.parametrize("name", ["thin_fog", "mixed"])
def test_scratch_buffers_may_start_as_anything(name, oracle, gpu):
    """No kernel reads a field of the scratch buffers that no kernel of ITS frame wrote: with the three buffers poisoned
    (api._POISON, GFT_POISON_SCRATCH=1: 0x7f bytes -- 3.4e38 as a float, an out-of-range index as an integer -- instead of
    the allocator's block, which usually holds the previous frame's plausible values) the lazy paths, both binning modes
    and both forward blend kernels still meet the oracle.  (The whole `-m gpu` suite and a soak pass with the switch set:
    profiles/r06_soak_raster_poisoned.json.)"""
    from gftorf_amd import api
    scene = Hh.small_scene(seed=21, **LAZY_CASES[name])
    f, b = Hh.run_oracle(oracle, scene)
    keep = api._POISON
    api._POISON = True
    try:
        for mode in (0, 1):
            for rm in (0, -1):
                with binning_mode(mode), render_mode(rm):
                    api._instance_hint.clear()                        # (the first frame's two-stage flow and the one-call flow)
                    for _ in range(2):
                        out, grads, _ = Hh.run_gpu(scene, gpu)
                        check_outputs(f, out)
                        check_grads(b, grads, scene)
    finally:
        api._POISON = keep
        api._instance_hint.clear()


@pytest.mark.parametrize("seed", list(range(12)))
def test_random_configurations(seed, oracle, gpu):
    """Seeded sweep over frame shapes, densities, SH degrees, camera poses and scale modifiers:
    images, counters and every gradient against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    W = int(rng.integers(17, 130))
    H = int(rng.integers(9, 90))
    P = int(rng.integers(50, 6000))
    D = int(rng.integers(0, 4))
    M = int(rng.choice([c for c in (1, 4, 9, 16) if c >= (D + 1) ** 2]))
    lo = float(10 ** rng.uniform(-3.0, -1.3))
    hi = lo * float(10 ** rng.uniform(0.2, 1.3))
    w2c = synth_pose(rng)
    scene = Hh.small_scene(P=P, W=W, H=H, seed=int(rng.integers(1 << 30)), D=D, sh_coeffs=M, scale_lo=lo, scale_hi=hi,
                           w2c=w2c, spread=float(rng.uniform(0.6, 1.8)),
                           opacity=None if rng.random() < 0.6 else float(rng.uniform(0.02, 0.98)))
    scene["use_view_dependent_phase"] = bool(rng.integers(0, 2))
    scene["phase_offset"] = float(rng.uniform(-0.5, 0.5))
    scene["dc_offset"] = float(rng.uniform(0.0, 0.2))
    over = dict(scale_modifier=float(rng.choice([1.0, 0.7, 1.3])))
    f, b = Hh.run_oracle(oracle, scene, **over)
    out, grads, _ = Hh.run_gpu(scene, gpu, optimize_offsets=True, **over)
    check_outputs(f, out)
    check_grads(b, grads, scene)


def synth_pose(rng):
    from gftorf_amd import synth
    return synth.look_at_w2c(float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.2, 0.2)), float(rng.uniform(-0.2, 0.2)),
                             (float(rng.uniform(-0.2, 0.2)), float(rng.uniform(-0.2, 0.2)), float(rng.uniform(-0.1, 0.4))))


@pytest.mark.gpu
@pytest.mark.parametrize("scene_kw", [SCENES["deep_lists"], SCENES["long_lists_lds128k"],
                                      dict(P=40000, W=64, H=64, scale_lo=0.01, scale_hi=0.05, opacity=0.02)],
                         ids=["deep_lists", "thousands_per_tile", "thin_fog_past_the_sorted_head"])
def test_split_backward_matches_one_wave_per_quadrant(tmp_path, scene_kw):
    """Deep quadrants are walked by up to 8 waves, one per 256 list entries, each starting from the blend state the
    forward saved there; GFT_BWD_SPLIT=0 keeps the serial walk.  Both must give the same gradients
    up to the rounding of the later waves' starting states.  Cases: a few segments; eight segments; lists whose sorted
    head ends before the deepest contributor (no state is saved past it: the cuts stop there)."""
    import subprocess
    import sys
    if os.environ.get("GFT_BWD_SPLIT", "1") == "0":
        pytest.skip("GFT_BWD_SPLIT=0: this process runs the serial walk itself, there is nothing to compare")
    child = tmp_path / "serial.py"
    out = tmp_path / "serial.npz"
    child.write_text(
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import helpers\n"
        "scene = helpers.small_scene(**%r)\n"
        "o, g, t = helpers.run_gpu(scene, torch.device('cuda:0'))\n"
        "np.savez(%r, **{k: v for k, v in g.items() if v is not None})\n"
        % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)),
           scene_kw, str(out)))
    subprocess.check_call([sys.executable, str(child)], env=dict(os.environ, GFT_BWD_SPLIT="0"), timeout=300)
    serial = np.load(out)
    scene = Hh.small_scene(**scene_kw)
    _, grads, _ = Hh.run_gpu(scene, torch.device("cuda:0"))
    worst = 0.0
    for k in serial.files:
        den = np.abs(serial[k]).max() + 1e-30
        worst = max(worst, float(np.abs(serial[k] - grads[k]).max() / den))
    assert worst < 2e-5, worst
    # and the split really was in use here: the scene has quadrants deeper than three batches
    assert os.environ.get("GFT_BWD_SPLIT", "1") != "0"


@pytest.mark.gpu
@pytest.mark.parametrize("scene_kw", [SCENES["deep_lists"], SCENES["long_lists_lds128k"],
                                      dict(P=40000, W=64, H=64, scale_lo=0.01, scale_hi=0.05, opacity=0.02)],
                         ids=["deep_lists", "thousands_per_tile", "thin_fog_far_segments"])
def test_deterministic_backward_mode(tmp_path, scene_kw, oracle):
    """GFT_BWD_DETERMINISTIC=1 (gft_backward_io.det_partials): every (list entry, quadrant) stores its partial row and one
    workgroup adds the rows in a fixed order instead of the float atomics (reference backward.cu:795-886, whose sums
    have no defined order).  Two runs must agree bit for bit; the result must agree with the atomic mode to
    summation-order rounding and with the oracle to the usual tolerance.  Cases: segments of the split walk, lists of
    thousands of entries, quadrants that go on into lazily sorted / far-slab segments."""
    import subprocess
    import sys
    child = tmp_path / "det.py"
    out = tmp_path / "det.npz"
    child.write_text(
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import helpers\n"
        "scene = helpers.small_scene(**%r)\n"
        "res = {}\n"
        "for run in (0, 1):\n"
        "    o, g, t = helpers.run_gpu(scene, torch.device('cuda:0'), optimize_offsets=True)\n"
        "    res.update({'%%d_%%s' %% (run, k): v for k, v in g.items() if v is not None})\n"
        "np.savez(%r, **res)\n"
        % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)),
           scene_kw, str(out)))
    subprocess.check_call([sys.executable, str(child)], env=dict(os.environ, GFT_BWD_DETERMINISTIC="1"), timeout=300)
    det = np.load(out)
    keys = sorted(k[2:] for k in det.files if k.startswith("0_"))
    assert "means3D" in keys and "shs" in keys
    for k in keys:
        np.testing.assert_array_equal(det["0_" + k], det["1_" + k], err_msg=k)      # bit-reproducible
    scene = Hh.small_scene(**scene_kw)
    _, grads, _ = Hh.run_gpu(scene, torch.device("cuda:0"), optimize_offsets=True)   # float atomics
    worst = 0.0
    for k in keys:
        den = np.abs(grads[k]).max() + 1e-30
        worst = max(worst, float(np.abs(det["0_" + k] - grads[k]).max() / den))
    assert worst < 2e-5, worst
    _, b = Hh.run_oracle(oracle, scene)
    check_grads(b, {k: det["0_" + k] for k in keys}, scene)


@pytest.mark.gpu
@pytest.mark.parametrize("z_lo,z_hi", [(3.0, 3.05), (4.0, 4.004), (1.0, 5.5)])
def test_depth_distortion_in_a_narrow_depth_range(z_lo, z_hi, oracle, gpu):
    """depth_distortion = A D2 - D^2 of the final sums; its two terms cancel to (depth spread / depth)^2 of their size, so
    the kernels accumulate around the tile's nearest depth.  Against the float64 re-derivation (tests/torch_ref.py) the
    plane must be as accurate as the oracle's reference-order fp32 sums (relative to the plane's own size, not to 1),
    and not negative beyond rounding."""
    import torch_ref
    sc = Hh.small_scene(P=1500, W=64, H=48, seed=5, scale_lo=0.03, scale_hi=0.15, z_lo=z_lo, z_hi=z_hi, w2c=None)
    f, _ = Hh.run_oracle(oracle, sc, backward=False)
    out, grads, _ = Hh.run_gpu(sc, gpu)
    dt = torch.float64
    params = {k: torch.tensor(v, dtype=dt, requires_grad=True) for k, v in sc["gaussians"].items() if v is not None}
    params["phase_offset"] = torch.tensor(sc["phase_offset"], dtype=dt, requires_grad=True)
    params["dc_offset"] = torch.tensor(sc["dc_offset"], dtype=dt, requires_grad=True)
    ref = torch_ref.render(params, f, Hh.oracle_kwargs(sc))["depth_distortion"].detach().numpy()
    size = float(np.abs(ref).max())
    err_oracle = float(np.abs(f["depth_distortion"] - ref).max())
    err_dev = float(np.abs(out["depth_distortion"] - ref).max())
    assert err_dev <= max(2.0 * err_oracle, 2e-3 * size), (err_dev, err_oracle, size)
    assert float(out["depth_distortion"].min()) >= -1e-3 * size
    # and the backward uses the same shifted sums: gradients against the oracle as everywhere else
    _, b = Hh.run_oracle(oracle, sc)
    check_grads(b, grads, sc)


def test_gradient_tensors_are_reused_only_when_nobody_holds_or_modified_them(oracle, gpu):
    """The operator keeps a set of gradient tensors whose rows it re-zeroes instead of writing 376 B of zeros per Gaussian
    in every backward (api.py: _grad_pool).  Reuse must never be observable: gradients of alternating scenes of one shape
    equal the oracle's every time; a caller that still holds a gradient, or wrote to one in place, gets fresh tensors and
    its old ones stay as they were."""
    from gftorf_amd import api
    if not api._GRADS_REUSE:
        pytest.skip("gradient-tensor reuse is off")
    a = Hh.small_scene(P=3000, seed=31)
    b = Hh.small_scene(P=3000, seed=32, opacity=0.7)
    fa, ba = Hh.run_oracle(oracle, a)
    fb, bb = Hh.run_oracle(oracle, b)
    api._grad_pool.clear()
    reused = []
    for it in range(6):
        sc, f, bw = (a, fa, ba) if it % 2 == 0 else (b, fb, bb)
        out, grads, t = Hh.run_gpu(sc, gpu)
        reused.append(api.last_call_stats["grads_reused"])
        check_outputs(f, out)
        check_grads(bw, grads, sc)
        del t, grads, out
    assert reused[0] is False and all(reused[1:]), reused
    # a caller that keeps a gradient tensor: the next call must not touch it
    out, grads, t = Hh.run_gpu(a, gpu)
    held = t["leaf"]["means3D"].grad
    snapshot = held.clone()
    out2, grads2, t2 = Hh.run_gpu(b, gpu)
    assert api.last_call_stats["grads_reused"] is False
    assert torch.equal(held, snapshot)
    check_grads(bb, grads2, b)
    del t, t2, held, grads, grads2, out, out2
    # ... and one that modified a gradient in place (clipping, accumulation over two backwards): not trusted again
    out, grads, t = Hh.run_gpu(a, gpu)
    t["leaf"]["shs"].grad.mul_(0.5)
    del t, grads, out
    pool = api._grad_pool[next(iter(api._grad_pool))]
    edited = [e["buf"].data_ptr() for e in pool if e["buf"]._version != e["version"]]
    assert len(edited) == 1                                   # the edit is visible in the buffer's version counter ...
    out, grads, t = Hh.run_gpu(b, gpu)
    check_grads(bb, grads, b)
    pool = api._grad_pool[next(iter(api._grad_pool))]
    assert edited[0] not in [e["buf"].data_ptr() for e in pool]     # ... and that buffer is forgotten, not reused
    del out, grads, t
    # a forward under grad whose backward never runs (a render for logging, a loss skipped by a NaN guard) takes a set of
    # tensors that nobody ever writes: the next call may take the same buffer, but must write it in full
    api._grad_pool.clear()
    ga = dict(a["gaussians"])
    leaf = torch.tensor(ga["means3D"], dtype=torch.float32, device=gpu, requires_grad=True)
    from gftorf_amd import GaussianRasterizer
    rast = GaussianRasterizer(raster_settings=Hh.gpu_settings(a, gpu))
    kw = {k: (torch.tensor(v, dtype=torch.float32, device=gpu) if v is not None else None) for k, v in ga.items() if k != "means3D"}
    outs = rast(means3D=leaf, means2D=torch.zeros((3000, 3), device=gpu), opacities=kw["opacities"], shs=kw.get("shs"),
                shs_p=kw.get("shs_p"), colors_precomp=kw.get("colors_precomp"), phasors_precomp=kw.get("phasors_precomp"),
                scales=kw.get("scales"), rotations=kw.get("rotations"), cov3D_precomp=kw.get("cov3D_precomp"),
                phase_offset=a["phase_offset"], dc_offset=a["dc_offset"])
    pool = api._grad_pool[next(iter(api._grad_pool))]
    assert len(pool) == 1 and pool[0]["valid"] is False
    pool[0]["buf"].fill_(float("nan"))          # what uninitialised memory may hold (through the pool's own alias: the
    pool[0]["version"] = pool[0]["buf"]._version   # version counter is the test's doing, not a caller's edit)
    del outs, leaf
    import gc
    gc.collect()
    for sc, bw in ((b, bb), (a, ba)):
        out, grads, t = Hh.run_gpu(sc, gpu)
        check_grads(bw, grads, sc)
        del out, grads, t
    assert api.last_call_stats["grads_reused"] is True
    assert len(api._grad_pool[next(iter(api._grad_pool))]) == 1        # the same buffer all along
    # the contract of the reuse: gradients are written through tensors only.  A write past the version counter (`.data`)
    # is invisible to the operator; the debug mode GFT_GRADS_REUSE_CHECK=1 finds it before the next reuse.
    api._GRADS_CHECK = True
    try:
        out, grads, t = Hh.run_gpu(b, gpu)                              # checked reuse of an untouched buffer: fine
        check_grads(bb, grads, b)
        t["leaf"]["means3D"].grad.data.add_(1.0)                        # old-style manual weight decay
        del out, grads, t
        with pytest.raises(RuntimeError, match="past the version counter"):
            Hh.run_gpu(a, gpu)
    finally:
        api._GRADS_CHECK = False
        api._grad_pool.clear()


def test_gradient_tensors_are_reused_without_the_private_use_count(oracle, gpu):
    """`torch._C._storage_Use_Count` is a private counter.  Without it (another torch version; GFT_GRADS_LIFETIME=dlpack)
    the pool hands out DLPack aliases of its memory, whose deleter tells when nobody references them any more (public
    API), and keeps autograd from taking the tensors over as `.grad` by holding them until the next forward.  Reuse works
    the same: alternating scenes against the oracle with reuse from the second call on; a gradient the caller keeps
    stays intact (it is autograd's copy on this route); a forward whose backward never ran leaves a buffer that is
    written in full next time."""
    from gftorf_amd import api
    if not api._GRADS_REUSE:
        pytest.skip("gradient-tensor reuse is off")
    a = Hh.small_scene(P=3000, seed=61)
    b = Hh.small_scene(P=3000, seed=62, opacity=0.7)
    fa, ba = Hh.run_oracle(oracle, a)
    fb, bb = Hh.run_oracle(oracle, b)
    keep = api._USE_COUNT_API
    api._USE_COUNT_API = False
    api._grad_pool.clear()
    try:
        reused, held = [], None
        for it in range(6):
            sc, bw = (a, ba) if it % 2 == 0 else (b, bb)
            out, grads, t = Hh.run_gpu(sc, gpu)
            reused.append(api.last_call_stats["grads_reused"])
            check_grads(bw, grads, sc)
            if it == 2:
                held = t["leaf"]["means3D"].grad
                snapshot = held.clone()
            del t, grads, out
        assert reused[0] is False and all(reused[1:]), reused
        assert torch.equal(held, snapshot)
        pool = api._grad_pool[next(iter(api._grad_pool))]
        assert len(pool) == 1 and "mem" in pool[0]                         # one buffer went round, on the DLPack route
        # a forward under grad whose backward never runs
        api._grad_pool.clear()
        from gftorf_amd import GaussianRasterizer
        ga = dict(a["gaussians"])
        leaf = torch.tensor(ga["means3D"], dtype=torch.float32, device=gpu, requires_grad=True)
        kw = {k: (torch.tensor(v, dtype=torch.float32, device=gpu) if v is not None else None) for k, v in ga.items() if k != "means3D"}
        outs = GaussianRasterizer(raster_settings=Hh.gpu_settings(a, gpu))(
            means3D=leaf, means2D=torch.zeros((3000, 3), device=gpu), opacities=kw["opacities"], shs=kw.get("shs"), shs_p=kw.get("shs_p"),
            scales=kw.get("scales"), rotations=kw.get("rotations"), phase_offset=a["phase_offset"], dc_offset=a["dc_offset"])
        pool = api._grad_pool[next(iter(api._grad_pool))]
        pool[0]["mem"].fill_(float("nan"))
        del outs, leaf
        import gc
        gc.collect()
        for sc, bw in ((b, bb), (a, ba)):
            out, grads, t = Hh.run_gpu(sc, gpu)
            check_grads(bw, grads, sc)
            del out, grads, t
        assert api.last_call_stats["grads_reused"] is True
    finally:
        api._USE_COUNT_API = keep
        api._grad_pool.clear()


def test_two_pending_backwards_on_the_dlpack_route(oracle, gpu):
    """DLPack route (no private use count): forward A, forward B of one shape, then A.backward(), B.backward() into the SAME
    leaves.  The second forward lets go of A's gradient tensors before autograd has seen them, so they become the leaves'
    `.grad` and B's gradient is added into A's pool buffer in place -- rows A's marks do not cover.  The entry must not be
    taken as "zero but for its marked rows" afterwards: the next renders' gradients are the oracle's."""
    from gftorf_amd import api, GaussianRasterizer
    if not api._GRADS_REUSE:
        pytest.skip("gradient-tensor reuse is off")
    a = Hh.small_scene(P=3000, seed=61)
    b = Hh.small_scene(P=3000, seed=62, opacity=0.7)
    c = Hh.small_scene(P=3000, seed=63, opacity=0.9, scale_lo=0.005, scale_hi=0.02)      # blends few Gaussians: most rows must be zero
    fa, ba = Hh.run_oracle(oracle, a)
    fb, bb = Hh.run_oracle(oracle, b)
    fc, bc = Hh.run_oracle(oracle, c)
    keep = api._USE_COUNT_API
    api._USE_COUNT_API = False
    api._grad_pool.clear()
    try:
        # shared leaves: the Gaussians of scene a rendered from the cameras of a and b
        g = dict(a["gaussians"])
        leaf = {k: torch.tensor(v, dtype=torch.float32, device=gpu, requires_grad=True) for k, v in g.items() if v is not None}
        m2 = torch.zeros((3000, 3), device=gpu, requires_grad=True)

        def render(scene):
            o = GaussianRasterizer(raster_settings=Hh.gpu_settings(scene, gpu))(
                means3D=leaf["means3D"], means2D=m2, opacities=leaf["opacities"], shs=leaf["shs"], shs_p=leaf["shs_p"],
                scales=leaf["scales"], rotations=leaf["rotations"], phase_offset=scene["phase_offset"], dc_offset=scene["dc_offset"])
            gr = scene["grads"]
            return sum((o[i] * torch.tensor(gr[k], device=gpu)).sum() for i, k in ((0, "color"), (1, "phasor"), (2, "depth"), (4, "acc"), (6, "depth_distortion")))
        b_on_a = dict(b, gaussians=a["gaussians"])
        fba, bba = Hh.run_oracle(oracle, b_on_a)
        la = render(a)
        lb = render(b_on_a)
        la.backward()
        lb.backward()
        want = np.asarray(ba["dL_dmeans3D"], np.float64) + np.asarray(bba["dL_dmeans3D"], np.float64)
        Hh.assert_close("sum of the two calls", want, leaf["means3D"].grad.cpu().numpy(), rtol_max=GRAD_RTOL)
        for v in leaf.values():
            v.grad = None
        m2.grad = None
        del la, lb
        # whatever buffers come round now: every gradient the oracle's
        for sc, bw in ((c, bc), (a, ba), (c, bc), (b, bb), (c, bc)):
            out, grads, t = Hh.run_gpu(sc, gpu)
            check_grads(bw, grads, sc)
            del out, grads, t
    finally:
        api._USE_COUNT_API = keep
        api._grad_pool.clear()


def test_kept_gradient_tensors_switch_between_rows_and_full_writes(oracle, gpu):
    """A kept set of gradient tensors is rewritten row by row while few Gaussians are blended and in full (coalesced, zeros
    included, every row marked) when most are: the rows backward leaves its row count in pinned host memory
    (gft_backward_io.rows_report) and api.py switches on it.  Not observable in the values: a frame that blends nearly
    everything and one that blends little, repeated and alternated, give the oracle's gradients in every mode and across
    every switch."""
    from gftorf_amd import api
    if not api._GRADS_REUSE:
        pytest.skip("gradient-tensor reuse is off")
    dense = Hh.small_scene(P=3000, seed=51, opacity=0.03)                       # thin: every visible Gaussian is blended
    sparse = Hh.small_scene(P=3000, seed=52, opacity=0.97, scale_lo=0.05, scale_hi=0.25)   # opaque: a fraction is
    ref = {id(sc): Hh.run_oracle(oracle, sc) for sc in (dense, sparse)}
    share = lambda sc: float((ref[id(sc)][0].pixels > 0).mean())
    assert share(dense) > 0.6 and share(sparse) < 0.3, (share(dense), share(sparse))
    api._grad_pool.clear()
    keep = api._DENSE_RUN
    api._DENSE_RUN = 2
    modes = []
    try:
        for sc in [dense] * 7 + [sparse] * 7 + [dense, sparse] * 4:
            out, grads, t = Hh.run_gpu(sc, gpu)
            torch.cuda.synchronize()
            modes.append((api.last_call_stats["grads_reused"], api.last_call_stats["grads_rows_only"], sc is dense))
            check_grads(ref[id(sc)][1], grads, sc)
            del out, grads, t
    finally:
        api._DENSE_RUN = keep
        api._grad_pool.clear()
    assert all(m[0] for m in modes[1:])                                # one buffer went round
    assert any(m[0] and not m[1] for m in modes[:7])                   # the dense frame came to be written in full ...
    assert any(m[1] for m in modes[7:14])                              # ... and the sparse one by rows again


def test_accumulator_is_kept_and_left_zero(oracle, gpu):
    """The backward's accumulator (64 B per Gaussian) is kept between calls (api.py: _AccLease): the preprocess backward
    zeroes the rows it has read (cfg.acc_zeroed = 2) and the next forward clears nothing.  Not observable: gradients of
    alternating scenes equal the oracle's every time, the buffer in the pool is all zero after every backward, a forward
    whose backward never runs hands its buffer back untouched, and a second backward through one forward
    (retain_graph) takes a buffer of its own."""
    from gftorf_amd import api
    if not api._ACC_REUSE:
        pytest.skip("accumulator reuse is off")
    a = Hh.small_scene(P=3000, seed=41)
    b = Hh.small_scene(P=3000, seed=42, opacity=0.7)
    fa, ba = Hh.run_oracle(oracle, a)
    fb, bb = Hh.run_oracle(oracle, b)
    api._acc_pool.clear()
    key = (torch.device(gpu).index, 3000)

    def pool_is_zero():
        torch.cuda.synchronize()
        bufs = api._acc_pool.get(key, [])
        assert len(bufs) >= 1
        return all(float(buf.abs().max()) == 0.0 or float(buf[:3000 * 16].abs().max()) == 0.0 for buf, _ in bufs)

    ptrs = []
    for it in range(5):
        sc, f, bw = (a, fa, ba) if it % 2 == 0 else (b, fb, bb)
        out, grads, t = Hh.run_gpu(sc, gpu)
        check_outputs(f, out)
        check_grads(bw, grads, sc)
        del t, grads, out
        assert pool_is_zero()
        ptrs.append(api._acc_pool[key][-1][0].data_ptr())
    assert len(set(ptrs)) == 1, ptrs                      # one buffer went round
    # a forward under grad whose backward never runs: its buffer comes back as it was
    g = dict(a["gaussians"])
    leaf = torch.tensor(g["means3D"], dtype=torch.float32, device=gpu, requires_grad=True)
    from gftorf_amd import GaussianRasterizer
    rast = GaussianRasterizer(raster_settings=Hh.gpu_settings(a, gpu))
    kw = {k: (torch.tensor(v, dtype=torch.float32, device=gpu) if v is not None else None) for k, v in g.items() if k != "means3D"}
    outs = rast(means3D=leaf, means2D=torch.zeros((3000, 3), device=gpu), opacities=kw["opacities"], shs=kw.get("shs"),
                shs_p=kw.get("shs_p"), colors_precomp=kw.get("colors_precomp"), phasors_precomp=kw.get("phasors_precomp"),
                scales=kw.get("scales"), rotations=kw.get("rotations"), cov3D_precomp=kw.get("cov3D_precomp"),
                phase_offset=a["phase_offset"], dc_offset=a["dc_offset"])
    assert len(api._acc_pool.get(key, [])) == 0           # (the forward holds the buffer)
    del outs
    import gc
    gc.collect()
    assert pool_is_zero() and api._acc_pool[key][-1][0].data_ptr() == ptrs[0]
    out, grads, t = Hh.run_gpu(b, gpu)
    check_grads(bb, grads, b)
    del out, grads, t
    # two backwards through one forward: the second takes its own accumulator, both give the oracle's gradients
    leaf = {k: torch.tensor(v, dtype=torch.float32, device=gpu, requires_grad=True) for k, v in a["gaussians"].items() if v is not None}
    m2 = torch.zeros((3000, 3), device=gpu, requires_grad=True)
    outs = rast(means3D=leaf["means3D"], means2D=m2, opacities=leaf["opacities"], shs=leaf.get("shs"), shs_p=leaf.get("shs_p"),
                colors_precomp=leaf.get("colors_precomp"), phasors_precomp=leaf.get("phasors_precomp"),
                scales=leaf.get("scales"), rotations=leaf.get("rotations"), cov3D_precomp=leaf.get("cov3D_precomp"),
                phase_offset=a["phase_offset"], dc_offset=a["dc_offset"])
    o = dict(zip(Hh.OUT_NAMES, outs))
    loss = sum((o[k] * torch.tensor(a["grads"][k], device=gpu)).sum() for k in Hh.GRAD_KEYS)
    for rep in range(2):
        for v in list(leaf.values()) + [m2]:
            v.grad = None
        loss.backward(retain_graph=(rep == 0))
        grads = {k: (v.grad.detach().cpu().numpy() if v.grad is not None else None) for k, v in leaf.items()}
        grads["means2D"] = m2.grad.detach().cpu().numpy()
        check_grads(ba, grads, a)
        assert pool_is_zero()
