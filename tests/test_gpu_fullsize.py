"""Full-size GPU checks (BASELINE.json configs): parity against the oracle where the oracle
finishes in seconds on the GPU box's host cores, and size-independent properties at the
largest shapes."""
import ctypes as C

import numpy as np
import pytest
import torch

import helpers as Hh
from gftorf_amd import synth

pytestmark = pytest.mark.gpu


def _scene(name, P=None):
    sc = synth.make_scene(name, seed=1234, P=P)
    return sc


def _head_ids(P, W, H):
    """ids sorted into tile-list heads by the most recent forward, and ids in lists that were completed on demand"""
    from gftorf_amd import _lib, api
    b = api.last_call_buffers
    L = _lib.get_layout(P, W, H, b["cap"])
    T = ((W + 15) // 16) * ((H + 15) // 16)
    heads = int(b["img"][L.img_front_len:L.img_front_len + 4 * T].view(torch.int32).sum().item())
    ctrl = b["img"][L.img_ctrl:L.img_ctrl + 64].view(torch.int32).cpu().tolist()
    return heads, ctrl[6], ctrl[4]


def test_metric_config_1m_vs_oracle(oracle, gpu):
    """1 M Gaussians @ 640x480, SH deg 3, RGB + ToF, forward + backward: the bench workload, in the flow of the first
    frame of a shape (two stages, one blocking read) AND in the flow every later frame runs (one call, buffer sized from
    the previous frame) -- the frames bench.py times.  Both through tile-pull binning: about a third of the frame's
    instances are ever sorted, the rest is never formed."""
    from gftorf_amd import _lib, api
    sc = _scene("metric")
    f, b = Hh.run_oracle(oracle, sc)
    api._instance_hint.clear()
    api.keep_last_buffers = True
    try:
        for frame in range(3):
            out, grads, _ = Hh.run_gpu(sc, gpu, optimize_offsets=True)
            st = api.last_call_stats
            assert st["num_rendered"] == f.num_rendered and not st["restarted"]
            if _lib.load().gft_lazy_sort():
                heads, completed, flagged = _head_ids(sc["cfg"]["P"], sc["cfg"]["W"], sc["cfg"]["H"])
                assert 0 < heads + completed < f.num_rendered // 2, (heads, completed, flagged)
            # integer decisions made per Gaussian: bit-exact
            np.testing.assert_array_equal(out["radii"], f.radii)
            for k in ["color", "phasor", "depth", "acc", "depth_distortion"]:
                l1 = float(np.abs(out[k].astype(np.float64) - f[k]).mean())
                assert l1 < 1e-5 * max(1.0, float(np.abs(f[k]).max())), (k, l1, frame)
            mism = float((out["pixels"] != f.pixels).mean())
            assert mism < 2e-3, mism
            for name, ref, got in [("means3D", b["dL_dmeans3D"], grads["means3D"]), ("means2D", b["dL_dmeans2D"], grads["means2D"]),
                                   ("opacity", b["dL_dopacity"], grads["opacities"]), ("sh", b["dL_dsh"], grads["shs"]),
                                   ("sh_p", b["dL_dsh_p"], grads["shs_p"]), ("scales", b["dL_dscales"], grads["scales"]),
                                   ("rot", b["dL_drotations"], grads["rotations"])]:
                # element-wise band: a handful of splats sit on the 1/255 alpha edge of one pixel
                Hh.assert_close(name, ref, got, rtol_max=5e-4, atol=1e-6, frac_bad=1e-5)
            Hh.assert_close("phase_offset", b["dL_dphase_offset"], grads["phase_offset"], rtol_max=1e-3, atol=1e-3)
            Hh.assert_close("dc_offset", b["dL_ddc_offset"], grads["dc_offset"], rtol_max=1e-3, atol=1e-3)
    finally:
        api.keep_last_buffers = False
        api.last_call_buffers.clear()


def test_c2_500k_forward_backward_vs_oracle(oracle, gpu):
    sc = _scene("C2")
    f, b = Hh.run_oracle(oracle, sc)
    out, grads, _ = Hh.run_gpu(sc, gpu)
    np.testing.assert_array_equal(out["radii"], f.radii)
    for k in ["color", "phasor", "depth", "acc", "depth_distortion"]:
        l1 = float(np.abs(out[k].astype(np.float64) - f[k]).mean())
        assert l1 < 1e-5 * max(1.0, float(np.abs(f[k]).max())), (k, l1)
    mism = float((out["pixels"] != f.pixels).mean())
    assert mism < 2e-3, mism
    # every gradient the operator returns (BASELINE.json config 2: forward + backward)
    for name, ref, got in [("means3D", b["dL_dmeans3D"], grads["means3D"]), ("means2D", b["dL_dmeans2D"], grads["means2D"]),
                           ("opacity", b["dL_dopacity"], grads["opacities"]), ("sh", b["dL_dsh"], grads["shs"]),
                           ("sh_p", b["dL_dsh_p"], grads["shs_p"]), ("scales", b["dL_dscales"], grads["scales"]),
                           ("rot", b["dL_drotations"], grads["rotations"])]:
        Hh.assert_close(name, ref, got, rtol_max=5e-4, atol=1e-6, frac_bad=1e-5)


def test_c1_10k_256_deg0_rgb_forward_vs_oracle(oracle, gpu):
    """BASELINE.json config 1: 10 k Gaussians, 256x256, SH degree 0 (one coefficient), RGB only (no shs_p: the
    phasor planes are background only), forward."""
    sc = _scene("C1")
    assert sc["cfg"] == dict(P=10_000, W=256, H=256, D=0, sh_coeffs=1, tof=False) and sc["gaussians"]["shs_p"] is None
    f, _ = Hh.run_oracle(oracle, sc, backward=False)
    out, _, _ = Hh.run_gpu(sc, gpu, backward=False)
    np.testing.assert_array_equal(out["radii"], f.radii)
    for k in ["color", "phasor", "depth", "acc", "depth_distortion", "distribution"]:
        l1 = float(np.abs(out[k].astype(np.float64) - f[k]).mean())
        assert l1 < 1e-5 * max(1.0, float(np.abs(f[k]).max())), (k, l1)
        Hh.assert_close(k, f[k], out[k], rtol_max=2e-4, atol=1e-6, frac_bad=1e-3)
    assert float((out["pixels"] != f.pixels).mean()) < 2e-3
    for k in ["normal", "entropy", "amp_distortion"]:
        assert not out[k].any()


def test_fog_1m_vs_oracle(oracle, gpu):
    """The metric frame in the regime the reference's scenes start in (opacity 0.05-0.1, arguments/__init__.py:99): no pixel
    saturates, every tile list (~3000 entries) is walked whole by forward and backward, most visible Gaussians are blended
    -- lists behind the sorted heads are completed for every tile.  Forward + backward against the oracle at full size, over
    four frames: the first two complete every list on demand (no schedule yet; then one the heads-only build of the pull kernel
    ignores), from the third on every tile sorts its whole list up front (gft_forward_io.tile_hints + the whole-list build)
    and no quadrant flags."""
    from gftorf_amd import _lib, api
    sc = _scene("fog")
    assert float(sc["gaussians"]["opacities"].max()) <= 0.1
    f, b = Hh.run_oracle(oracle, sc)
    vis = f.radii > 0
    assert float((f.pixels[vis] > 0).mean()) > 0.8                 # most visible Gaussians are blended by some pixel
    assert float(f.img["final_T"].min()) > 1e-4                     # ... and nothing saturates
    api._instance_hint.clear()
    api.state.reset_schedules()
    keep = api.keep_last_buffers
    api.keep_last_buffers = True
    flagged = []
    for frame in range(4):
        out, grads, _ = Hh.run_gpu(sc, gpu, optimize_offsets=True)
        assert api.last_call_stats["num_rendered"] == f.num_rendered
        bufs = api.last_call_buffers
        L = _lib.get_layout(bufs["P"], bufs["W"], bufs["H"], bufs["cap"])
        flagged.append(int(bufs["img"][L.img_ctrl:L.img_ctrl + 64].view(torch.int32)[4].item()))
        np.testing.assert_array_equal(out["radii"], f.radii)
        for k in ["color", "phasor", "depth", "acc", "depth_distortion"]:
            l1 = float(np.abs(out[k].astype(np.float64) - f[k]).mean())
            assert l1 < 1e-5 * max(1.0, float(np.abs(f[k]).max())), (k, l1, frame)
        Hh.assert_close("distribution", f["distribution"], out["distribution"], rtol_max=2e-4, atol=1e-6, frac_bad=1e-3)
        mism = float((out["pixels"] != f.pixels).mean())
        assert mism < 2e-3, mism
        for name, ref, got in [("means3D", b["dL_dmeans3D"], grads["means3D"]), ("means2D", b["dL_dmeans2D"], grads["means2D"]),
                               ("opacity", b["dL_dopacity"], grads["opacities"]), ("sh", b["dL_dsh"], grads["shs"]),
                               ("sh_p", b["dL_dsh_p"], grads["shs_p"]), ("scales", b["dL_dscales"], grads["scales"]),
                               ("rot", b["dL_drotations"], grads["rotations"])]:
            Hh.assert_close(name, ref, got, rtol_max=5e-4, atol=1e-6, frac_bad=1e-5)
        Hh.assert_close("phase_offset", b["dL_dphase_offset"], grads["phase_offset"], rtol_max=1e-3, atol=1e-3)
        Hh.assert_close("dc_offset", b["dL_ddc_offset"], grads["dc_offset"], rtol_max=1e-3, atol=1e-3)
    api.keep_last_buffers = keep
    api.last_call_buffers.clear()
    if _lib.load().gft_lazy_sort() and _lib.load().gft_binning_mode(C.byref(_lib.Config(P=1, W=640, H=480))):
        assert flagged[0] > 4000 and flagged[1] > 4000 and flagged[2] == 0 and flagged[3] == 0, flagged


def test_c5_5m_1080p_vs_oracle(oracle, gpu):
    """BASELINE.json config 5 at its full size against the oracle: 5 M Gaussians @ 1920x1080, SH degree 3, RGB + ToF,
    forward + backward (74.7 M instances on 8160 tiles; the oracle's binning runs on all host cores).  Two frames: the
    two-stage flow of a shape's first frame, then the one-call flow with the buffer sized from it.  `radii` bit-exact,
    images L1 < 1e-5, pixel counts, every gradient in the band of the 1 M test."""
    from gftorf_amd import api
    sc = _scene("C5")
    f, b = Hh.run_oracle(oracle, sc)
    assert f.num_rendered > 70_000_000
    api._instance_hint.clear()
    for frame in range(2):
        out, grads, _ = Hh.run_gpu(sc, gpu, optimize_offsets=True)
        st = api.last_call_stats
        assert st["num_rendered"] == f.num_rendered and not st["restarted"]
        np.testing.assert_array_equal(out["radii"], f.radii)
        for k in ["color", "phasor", "depth", "acc", "depth_distortion"]:
            l1 = float(np.abs(out[k].astype(np.float64) - f[k]).mean())
            assert l1 < 1e-5 * max(1.0, float(np.abs(f[k]).max())), (k, l1, frame)
        Hh.assert_close("distribution", f["distribution"], out["distribution"], rtol_max=2e-4, atol=1e-6, frac_bad=1e-3)
        mism = float((out["pixels"] != f.pixels).mean())
        assert mism < 2e-3, mism
        for name, ref, got in [("means3D", b["dL_dmeans3D"], grads["means3D"]), ("means2D", b["dL_dmeans2D"], grads["means2D"]),
                               ("opacity", b["dL_dopacity"], grads["opacities"]), ("sh", b["dL_dsh"], grads["shs"]),
                               ("sh_p", b["dL_dsh_p"], grads["shs_p"]), ("scales", b["dL_dscales"], grads["scales"]),
                               ("rot", b["dL_drotations"], grads["rotations"])]:
            Hh.assert_close(name, ref, got, rtol_max=5e-4, atol=1e-6, frac_bad=1e-5)
        Hh.assert_close("phase_offset", b["dL_dphase_offset"], grads["phase_offset"], rtol_max=1e-3, atol=1e-3)
        Hh.assert_close("dc_offset", b["dL_ddc_offset"], grads["dc_offset"], rtol_max=1e-3, atol=1e-3)
        del out, grads


def test_c5_fog_whole_lists_equal_lists_completed_on_demand(gpu):
    """5 M Gaussians @ 1920x1080 at opacity 0.05-0.1: 8160 tiles with lists of ~9000 entries of which no pixel saturates.
    The per-tile schedule's two routes at that size -- every list completed on demand (heads-only build of the pull kernel,
    4 depth slabs per supertile of 4 x 4 tiles, `k_tail_build`, resume pass) and every list sorted whole up front (the
    whole-list build: chunks of whole depth bins over all slabs, the pool filled by 8160 tiles) -- give the same frame bit
    for bit and the same gradients up to the order of their sums."""
    from gftorf_amd import _lib, api
    if not _lib.load().gft_lazy_sort():
        pytest.skip("GFT_LAZY_SORT=0: whole-frame binning only")
    cfg = dict(synth.CONFIGS["C5"], opacity_range=(0.05, 0.1))
    sc = synth.make_scene(cfg, seed=1234)
    W, H = cfg["W"], cfg["H"]
    T = ((W + 15) // 16) * ((H + 15) // 16)
    api._instance_hint.clear()
    api.state.reset_schedules()
    keep = api.keep_last_buffers
    api.keep_last_buffers = True
    res = {}
    try:
        for route in ("first", "on_demand", "whole", "whole_again"):
            api._force_whole_lists = None if route == "first" else route != "on_demand"
            out, grads, _ = Hh.run_gpu(sc, gpu)
            bufs = api.last_call_buffers
            L = _lib.get_layout(bufs["P"], bufs["W"], bufs["H"], bufs["cap"])
            ctrl = bufs["img"][L.img_ctrl:L.img_ctrl + 64].view(torch.int32).cpu().tolist()
            fl = bufs["img"][L.img_front_len:L.img_front_len + 4 * T].view(torch.int32)
            res[route] = (out, grads, ctrl[4], int(fl.sum().item()), int(api.last_call_stats["num_rendered"]))
            del out, grads
    finally:
        api._force_whole_lists = None
        api.keep_last_buffers = keep
        api.last_call_buffers.clear()
    R = res["first"][4]
    assert R > 70_000_000
    # flagged quadrants: most of them after the first frame's heads of ~940 entries; with the long heads the schedule gives the
    # heads-only build (2047 entries: at this density part of the pixels saturate inside them) still thousands; none with whole lists
    assert res["first"][2] > 3 * T and res["on_demand"][2] > T // 2 and res["whole"][2] == 0 and res["whole_again"][2] == 0
    assert res["whole"][3] == R and res["whole_again"][3] == R                                    # every instance sorted up front
    ref_out, ref_grads = res["on_demand"][:2]
    for route in ("first", "whole", "whole_again"):
        out, grads = res[route][:2]
        for k in ref_out:
            np.testing.assert_array_equal(out[k], ref_out[k], err_msg="%s %s" % (route, k))
        for k in ref_grads:
            if ref_grads[k] is not None:
                # (the same terms added by atomics in another order: fp32 sums of up to 1e5 terms per row; the rotation rows,
                # whose terms cancel most, were seen at 2.04e-5 of the max-norm)
                Hh.assert_close("%s %s" % (route, k), ref_grads[k], grads[k], rtol_max=5e-5, atol=1e-7)


def test_c5_5m_1080p_with_deform_offsets_tile_pull_vs_whole_frame(gpu):
    """BASELINE.json config 5 as it is named: 5 M Gaussians @ 1920x1080, ToF + dynamic deform -- d_xyz / d_sh of the
    deformation network (the architecture the reference constructs) for the 30 % dynamic Gaussians, composed in by the
    fused input assembly, forward + backward through all three pieces.  Size-independent properties: the frame rendered
    with tile-pull binning (the production path: depth slabs, list heads, appearance on demand) equals the frame with
    every instance binned bit for bit, gradients reach the network, and the raster gradients agree."""
    from gftorf_amd import GaussianRasterizer, _lib, api, assemble_inputs, reference_network
    from oracle import deform_ref
    sc = _scene("C5")
    P, W, H = sc["cfg"]["P"], sc["cfg"]["W"], sc["cfg"]["H"]
    g = sc["gaussians"]
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=gpu)
    rng = np.random.default_rng(5)
    mask = torch.tensor(rng.random(P) < 0.3, device=gpu)
    net = reference_network()
    net.load_state_dict({k: torch.tensor(v) for k, v in deform_ref.random_params(9, head_std=2e-3).items()})
    net = net.to(gpu)
    leaf = dict(xyz=t(g["means3D"]), opacity=t(g["opacities"]).reshape(P, 1), scaling=t(g["scales"]), rotation_raw=t(g["rotations"]),
                fc=t(g["shs"]), fp=t(g["shs_p"]))
    for v in leaf.values():
        v.requires_grad_(True)
    x_n = leaf["xyz"].detach()[mask]
    x_n = (x_n - x_n.min(0).values) / (x_n.max(0).values - x_n.min(0).values)
    rast = GaussianRasterizer(Hh.gpu_settings(sc, gpu))
    gr = {k: t(v) for k, v in sc["grads"].items()}
    api._instance_hint.clear()

    def frame():
        for v in leaf.values():
            v.grad = None
        net.zero_grad(set_to_none=True)
        d = net(x_n, torch.full((1, 1), 0.4, device=gpu).expand(x_n.size(0), -1))
        ssp = torch.zeros((P, 3), device=gpu, requires_grad=True)
        rot = torch.nn.functional.normalize(leaf["rotation_raw"])
        m3, m2, op, sc_, ro, shs, shp = assemble_inputs(leaf["xyz"], ssp, leaf["opacity"], leaf["scaling"], rot, leaf["rotation_raw"],
                                                        leaf["fc"], leaf["fp"], mask, *d)
        out = rast(means3D=m3, means2D=m2, opacities=op, shs=shs, shs_p=shp, scales=sc_, rotations=ro,
                   phase_offset=sc["phase_offset"], dc_offset=sc["dc_offset"])
        torch.autograd.backward([out[0], out[1], out[2], out[4], out[6]],
                                [gr["color"], gr["phasor"], gr["depth"], gr["acc"], gr["depth_distortion"]])
        st = dict(api.last_call_stats)
        return [o.detach().clone() for o in out], {k: v.grad.clone() for k, v in leaf.items()}, \
            {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}, ssp.grad.clone(), st

    _lib.load().gft_set_binning_mode(0)
    try:
        o1, g1, n1, s1, st1 = frame()      # whole-frame binning (two-stage flow): every instance counted, keyed, sorted
    finally:
        _lib.load().gft_set_binning_mode(-1)
    api._instance_hint.clear()
    o2, g2, n2, s2, st2 = frame()          # tile-pull binning, two-stage flow (first frame of the shape)
    o3, g3, n3, s3, st3 = frame()          # tile-pull binning, one call (buffer sized from the previous frame)
    R = st1["num_rendered"]
    assert st2["num_rendered"] == R and st3["num_rendered"] == R and not st3["restarted"]
    for a, b, c in zip(o1, o2, o3):
        assert torch.equal(a, b) and torch.equal(a, c)
    assert all(torch.isfinite(v).all() for v in o1)
    # gradients: atomic order only
    for k in g1:
        Hh.assert_close(k, g1[k].cpu().numpy(), g3[k].cpu().numpy(), rtol_max=1e-5)
    Hh.assert_close("screenspace", s1.cpu().numpy(), s3.cpu().numpy(), rtol_max=1e-5)
    assert len(n1) == 24 and all(torch.isfinite(v).all() and v.abs().max() > 0 for v in n1.values())
    for k in n1:
        Hh.assert_close(k, n1[k].cpu().numpy(), n3[k].cpu().numpy(), rtol_max=2e-4)
    # static Gaussians get no offset gradient path: d(loss)/d(xyz) of a static row equals the rasterizer's means3D gradient
    assert g1["xyz"].abs().max() > 0 and torch.isfinite(g1["xyz"]).all()


def test_c5_5m_1080p_properties(gpu):
    """5 M Gaussians @ 1920x1080 (HBM stress): size-independent properties, no oracle."""
    from gftorf_amd import GaussianRasterizer
    sc = _scene("C5")
    W, H = sc["cfg"]["W"], sc["cfg"]["H"]
    out, grads, t = Hh.run_gpu(sc, gpu)
    assert all(np.isfinite(v).all() for v in out.values())
    assert all(np.isfinite(v).all() for v in grads.values() if v is not None)
    # sum of blend weights telescopes: acc = 1 - T_final; background enters as T_final * bg
    zero_bg = torch.zeros(7, H, W, device=gpu)
    out0, _, _ = Hh.run_gpu(sc, gpu, backward=False, bg=zero_bg)
    T = 1.0 - out0["acc"][0]
    np.testing.assert_allclose(out["color"] - out0["color"], T[None] * sc["bg"][:3], atol=3e-6)
    np.testing.assert_allclose(out["phasor"] - out0["phasor"], T[None] * sc["bg"][:7], atol=3e-6)
    np.testing.assert_array_equal(out["depth"], out0["depth"])
    np.testing.assert_array_equal(out["radii"], out0["radii"])
    # quad planes of the phasor are +-real + dc*amp, +-imag + dc*amp
    dc = sc["dc_offset"]
    ph = out0["phasor"]
    np.testing.assert_allclose(ph[3], ph[0] + dc * ph[2], atol=2e-6)
    np.testing.assert_allclose(ph[4], -ph[0] + dc * ph[2], atol=2e-6)
    np.testing.assert_allclose(ph[5], ph[1] + dc * ph[2], atol=2e-6)
    np.testing.assert_allclose(ph[6], -ph[1] + dc * ph[2], atol=2e-6)
    # counters: integers, zero for culled Gaussians, and a culled Gaussian has zero gradients
    assert (out["pixels"] == np.round(out["pixels"])).all()
    culled = out["radii"] <= 0
    assert not out["pixels"][culled].any()
    for k in ["means3D", "shs", "shs_p", "scales", "rotations", "opacities", "means2D"]:
        assert not grads[k][culled].any(), k
    # determinism of the forward
    out1, _, _ = Hh.run_gpu(sc, gpu, backward=False)
    for k in ["color", "phasor", "depth", "acc", "radii", "pixels", "distribution", "depth_distortion"]:
        np.testing.assert_array_equal(out[k], out1[k])


def test_metric_config_deterministic_backward_is_bit_reproducible(tmp_path, gpu):
    """The bench workload through GFT_BWD_DETERMINISTIC=1 (fixed-order sums instead of float atomics, 1.1 GB of partial
    rows): two runs give bit-identical gradients, and they agree with the atomic mode to summation-order rounding --
    the element-wise band of test_metric_config_1m_vs_oracle is for splats on the 1/255 alpha edge, not for the atomics."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    child, out = tmp_path / "det_full.py", tmp_path / "det_full.npz"
    child.write_text(
        "import sys, zlib, numpy as np, torch\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import helpers\n"
        "from gftorf_amd import synth\n"
        "sc = synth.make_scene('metric', seed=1234)\n"
        "res = {}\n"
        "for run in (0, 1):\n"
        "    o, g, t = helpers.run_gpu(sc, torch.device('cuda:0'))\n"
        "    for k, v in g.items():\n"
        "        if v is not None:\n"
        "            res['crc%%d_%%s' %% (run, k)] = np.array(zlib.crc32(np.ascontiguousarray(v).tobytes()), np.uint32)\n"
        "for k in ('means3D', 'means2D', 'opacities', 'scales', 'rotations'):\n"
        "    res['g_' + k] = g[k]\n"
        "res['g_shs'] = g['shs'][::16]; res['g_shs_p'] = g['shs_p'][::16]\n"
        "np.savez(%r, **res)\n" % (os.path.dirname(here), here, str(out)))
    subprocess.check_call([sys.executable, str(child)], env=dict(os.environ, GFT_BWD_DETERMINISTIC="1"), timeout=900)
    det = np.load(out)
    names = sorted(k[5:] for k in det.files if k.startswith("crc0_"))
    assert len(names) >= 7
    for k in names:
        assert int(det["crc0_" + k]) == int(det["crc1_" + k]), k
    sc = _scene("metric")
    _, grads, _ = Hh.run_gpu(sc, gpu)
    for k in ("means3D", "means2D", "opacities", "scales", "rotations"):
        Hh.assert_close(k, det["g_" + k], grads[k], rtol_max=2e-5, atol=1e-7)
    Hh.assert_close("shs", det["g_shs"], grads["shs"][::16], rtol_max=2e-5, atol=1e-7)
    Hh.assert_close("shs_p", det["g_shs_p"], grads["shs_p"][::16], rtol_max=2e-5, atol=1e-7)


def test_c3_shaped_view_sequence_vs_oracle(oracle, gpu):
    """BASELINE.json config 3 at its named shape (configs/torf.json:15-23: 100 k Gaussians, 320x240; train.py:118-279): a
    SEQUENCE of different views -- 8 of the 30 arc views in shuffled order, each as the colour-camera call and the
    ToF-camera call of one iteration (gaussian_renderer/__init__.py:107-128), the active SH degree raised 0 -> 3 along the
    way (train.py:143-145) -- through two long-lived GaussianRasterizer modules, so that every call but the first of
    each kind runs the one-call flow with a buffer sized by another view's frame.  Every frame's images, counters and
    gradients against the oracle."""
    import math
    import random
    from gftorf_amd import GaussianRasterizer, api
    P, W, H, V = 100_000, 320, 240, 30
    base = synth.make_camera(W, H)
    g = synth.make_gaussians(P, base, 1236, sh_coeffs=16, scale_lo=0.004, scale_hi=0.04)
    bg, grads = synth.make_background(W, H, 5), synth.make_pixel_grads(W, H, 5)
    order = list(range(V))
    random.Random(3).shuffle(order)
    api._instance_hint.clear()
    leaf = {k: torch.tensor(v, dtype=torch.float32, device=gpu, requires_grad=True) for k, v in g.items()}
    m2 = torch.zeros((P, 3), device=gpu, requires_grad=True)
    up = {k: torch.tensor(v, device=gpu) for k, v in grads.items()}
    restarts = 0
    for it, v in enumerate(order[:8]):
        a = (v / (V - 1) - 0.5) * 0.30
        pose = dict(yaw=a, pitch=0.04 * math.sin(3 * a))
        degree = min(3, it // 2)
        for tof in (False, True):
            # the ToF sensor sits beside the colour camera (its own pose, scene/cameras.py:121-146)
            w2c = synth.look_at_w2c(t=(-3.2 * math.sin(a) + (0.03 if tof else 0.0), 0.0, 3.2 * (1 - math.cos(a))), **pose)
            sc = dict(cfg=dict(P=P, W=W, H=H, D=degree, sh_coeffs=16, tof=True), cam=synth.make_camera(W, H, w2c=w2c), gaussians=g,
                      bg=bg, grads=grads, depth_range=10.0 if tof else 100.0, phase_offset=0.1 if tof else 0.0,
                      dc_offset=0.02 if tof else 0.0, use_view_dependent_phase=tof)
            f, b = Hh.run_oracle(oracle, sc)
            for x in list(leaf.values()) + [m2]:
                x.grad = None
            out = GaussianRasterizer(Hh.gpu_settings(sc, gpu))(
                means3D=leaf["means3D"], means2D=m2, opacities=leaf["opacities"], shs=leaf["shs"], shs_p=leaf["shs_p"],
                scales=leaf["scales"], rotations=leaf["rotations"], phase_offset=sc["phase_offset"], dc_offset=sc["dc_offset"])
            o = dict(zip(Hh.OUT_NAMES, out))
            sum((o[k] * up[k]).sum() for k in Hh.GRAD_KEYS).backward()
            st = api.last_call_stats
            restarts += int(st["restarted"])
            assert st["num_rendered"] == f.num_rendered, (it, tof)
            np.testing.assert_array_equal(o["radii"].cpu().numpy(), f.radii)
            for k in Hh.GRAD_KEYS + ["distribution"]:
                got = o[k].detach().cpu().numpy()
                l1 = float(np.abs(got.astype(np.float64) - f[k]).mean())
                assert l1 < 1e-5 * max(1.0, float(np.abs(f[k]).max())), (k, l1, it, tof)
                Hh.assert_close(k, f[k], got, rtol_max=2e-4, atol=1e-6, frac_bad=1e-3)
            assert float((o["pixels"].detach().cpu().numpy() != f.pixels).mean()) < 2e-3
            for name, ref, got in [("means3D", b["dL_dmeans3D"], leaf["means3D"].grad), ("means2D", b["dL_dmeans2D"], m2.grad),
                                   ("opacity", b["dL_dopacity"], leaf["opacities"].grad), ("sh", b["dL_dsh"], leaf["shs"].grad),
                                   ("sh_p", b["dL_dsh_p"], leaf["shs_p"].grad), ("scales", b["dL_dscales"], leaf["scales"].grad),
                                   ("rot", b["dL_drotations"], leaf["rotations"].grad)]:
                Hh.assert_close("%s (iteration %d, %s call)" % (name, it, "ToF" if tof else "colour"), ref, got.cpu().numpy(),
                                rtol_max=5e-4, atol=1e-6, frac_bad=1e-5)
    assert restarts <= 2          # the buffer guess (decaying maximum of the recent frames + 25 %) rarely misses


def test_metric_views_repeat_bit_identically(gpu):
    """Soak over changing views at the metric size: 8 views of an arc in shuffled order, three rounds, forward + backward
    each.  Everything that is kept between calls (gradient tensors and their row marks, the accumulator the backward leaves
    zero, the buffer-size guess) and everything decided per frame (heads, flagged tiles, completed lists) must leave no
    trace: a view's outputs are bit-identical every time it comes round, its gradient sums equal up to the order of the
    float atomics."""
    import random
    from gftorf_amd import GaussianRasterizationSettings, GaussianRasterizer
    sc = _scene("metric")
    cfg, g = sc["cfg"], sc["gaussians"]
    P, W, H = cfg["P"], cfg["W"], cfg["H"]
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=gpu)
    bg = t(sc["bg"])
    rasts = []
    for v in range(8):
        a = (v / 7 - 0.5) * 0.30
        cam = synth.make_camera(W, H, w2c=synth.look_at_w2c(yaw=a, pitch=0.04 * np.sin(3 * a), t=(-3.2 * np.sin(a), 0.0, 3.2 * (1 - np.cos(a)))))
        rasts.append(GaussianRasterizer(GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg, scale_modifier=1.0,
            viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), sh_degree=cfg["D"], campos=t(cam["campos"]),
            prefiltered=False, debug=False, near_n=cam["znear"], far_n=cam["zfar"], depth_range=sc["depth_range"],
            use_view_dependent_phase=sc["use_view_dependent_phase"])))
    leaf = {k: t(v).requires_grad_(True) for k, v in g.items() if v is not None}
    m2 = torch.zeros((P, 3), device=gpu, requires_grad=True)
    ups = [t(sc["grads"][k]) for k in ("color", "phasor", "depth", "acc", "depth_distortion")]
    ref, rng = {}, random.Random(5)
    for rd in range(3):
        for v in rng.sample(range(8), 8):
            for x in leaf.values():
                x.grad = None
            m2.grad = None
            o = rasts[v](means3D=leaf["means3D"], means2D=m2, opacities=leaf["opacities"], shs=leaf["shs"], shs_p=leaf["shs_p"],
                         scales=leaf["scales"], rotations=leaf["rotations"], phase_offset=sc["phase_offset"], dc_offset=sc["dc_offset"])
            torch.autograd.backward([o[0], o[1], o[2], o[4], o[6]], ups)
            img = torch.cat([o[i].reshape(-1) for i in (0, 1, 2, 4, 6, 8, 9)])
            gs = torch.stack([leaf[k].grad.double().abs().sum() for k in ("means3D", "opacities", "shs", "shs_p", "scales", "rotations")]
                             + [m2.grad.double().abs().sum()])
            if v not in ref:
                ref[v] = (img.clone(), gs.clone())
                continue
            assert torch.equal(img, ref[v][0]), "view %d, round %d: forward outputs differ" % (v, rd)
            rel = float(((gs - ref[v][1]).abs() / (ref[v][1].abs() + 1e-30)).max())
            assert rel < 1e-5, (v, rd, rel)
