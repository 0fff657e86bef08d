"""Full-size GPU checks (BASELINE.json configs): parity against the oracle where the oracle
finishes in seconds on the GPU box's host cores, and size-independent properties at the
largest shapes."""
import numpy as np
import pytest
import torch

import helpers as Hh
from gftorf_amd import synth

pytestmark = pytest.mark.gpu


def _scene(name, P=None):
    sc = synth.make_scene(name, seed=1234, P=P)
    return sc


def test_metric_config_1m_vs_oracle(oracle, gpu):
    """1 M Gaussians @ 640x480, SH deg 3, RGB + ToF, forward + backward: the bench workload."""
    sc = _scene("metric")
    f, b = Hh.run_oracle(oracle, sc)
    out, grads, _ = Hh.run_gpu(sc, gpu, optimize_offsets=True)
    # integer decisions made per Gaussian: bit-exact
    np.testing.assert_array_equal(out["radii"], f.radii)
    for k in ["color", "phasor", "depth", "acc", "depth_distortion"]:
        l1 = float(np.abs(out[k].astype(np.float64) - f[k]).mean())
        assert l1 < 1e-5 * max(1.0, float(np.abs(f[k]).max())), (k, l1)
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


def test_c5_5m_1080p_with_deform_offsets_and_lazy_binning(gpu):
    """BASELINE.json config 5 as it is named: 5 M Gaussians @ 1920x1080, ToF + dynamic deform -- d_xyz / d_sh of the
    deformation network (the architecture the reference constructs) for the 30 % dynamic Gaussians, composed in by the
    fused input assembly, forward + backward through all three pieces.  Size-independent properties: the frame rendered
    with the near-slab binning (depth cut suggested by the previous frame) equals the frame with every instance binned
    bit for bit, gradients reach the network, and the raster gradients equal those of feeding the assembled tensors."""
    from gftorf_amd import GaussianRasterizer, api, assemble_inputs, reference_network
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

    o1, g1, n1, s1, st1 = frame()          # two-stage flow: every instance binned
    o2, g2, n2, s2, st2 = frame()          # one-call flow, no cut yet: measures the depth histogram
    o3, g3, n3, s3, st3 = frame()          # near-slab binning with the suggested cut
    R = st1["num_rendered"]
    assert st1["near_instances"] == R and st2["near_instances"] == R and st3["num_rendered"] == R
    assert st3["depth_cut"] > 0 and st3["near_instances"] < R // 3, st3
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
