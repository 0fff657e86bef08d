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
    Hh.assert_close("means3D", b["dL_dmeans3D"], grads["means3D"], rtol_max=5e-4, atol=1e-6, frac_bad=1e-5)
    Hh.assert_close("sh_p", b["dL_dsh_p"], grads["shs_p"], rtol_max=5e-4, atol=1e-6, frac_bad=1e-5)


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
