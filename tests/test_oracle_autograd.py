"""The oracle's hand-written backward (restating RAST/cuda_rasterizer/backward.cu) must be
the gradient of its forward: compare with float64 PyTorch autograd of an independent dense
re-derivation (tests/torch_ref.py), over the operator's argument patterns."""
import numpy as np
import pytest
import torch

import helpers as Hh
import torch_ref

RTOL = 5e-5   # fp32 oracle vs float64 autograd, relative to the max-norm


def _compare(oracle, scene, inputs=None, check=("means3D", "opacities", "scales", "rotations", "shs", "shs_p")):
    f, b = Hh.run_oracle(oracle, scene, inputs=inputs)
    g = dict(scene["gaussians"])
    if inputs:
        g.update(inputs)
    dt = torch.float64
    params = {k: torch.tensor(v, dtype=dt, requires_grad=True) for k, v in g.items() if v is not None}
    for k in ("shs", "shs_p", "colors_precomp", "phasors_precomp", "cov3D_precomp", "scales", "rotations"):
        params.setdefault(k, None)
    params["phase_offset"] = torch.tensor(scene["phase_offset"], dtype=dt, requires_grad=True)
    params["dc_offset"] = torch.tensor(scene["dc_offset"], dtype=dt, requires_grad=True)
    settings = Hh.oracle_kwargs(scene)
    out = torch_ref.render(params, f, settings)
    for k in Hh.GRAD_KEYS:
        Hh.assert_close("fwd " + k, out[k].detach().numpy(), f[k], rtol_max=2e-5, atol=1e-6)
    loss = sum((out[k] * torch.tensor(scene["grads"][k], dtype=dt)).sum() for k in Hh.GRAD_KEYS)
    loss.backward()
    names = dict(means3D="dL_dmeans3D", opacities="dL_dopacity", scales="dL_dscales", rotations="dL_drotations",
                 shs="dL_dsh", shs_p="dL_dsh_p", colors_precomp="dL_dcolors", cov3D_precomp="dL_dcov3D")
    for k in check:
        if params.get(k) is None:
            continue
        ref = params[k].grad.numpy()
        Hh.assert_close(k, ref, b[names[k]].reshape(ref.shape), rtol_max=RTOL, atol=1e-7)
    Hh.assert_close("means2D", out["ndc"].grad.numpy(), b["dL_dmeans2D"][:, :2], rtol_max=RTOL, atol=1e-7)
    assert not b["dL_dmeans2D"][:, 2].any()
    if g.get("shs_p") is not None:
        Hh.assert_close("phase_offset", params["phase_offset"].grad.numpy(), b["dL_dphase_offset"][0], rtol_max=RTOL, atol=1e-6)
        Hh.assert_close("dc_offset", params["dc_offset"].grad.numpy(), b["dL_ddc_offset"][0], rtol_max=RTOL, atol=1e-6)
    return f, b


@pytest.mark.parametrize("D,M", [(3, 16), (0, 16), (1, 4), (2, 9)])
def test_sh_paths(D, M, oracle):
    _compare(oracle, Hh.small_scene(P=120, W=48, H=32, D=D, sh_coeffs=M, scale_lo=0.02, scale_hi=0.15))


def test_view_dependent_phase_off(oracle):
    sc = Hh.small_scene(P=120, W=48, H=32, scale_lo=0.02, scale_hi=0.15)
    sc["use_view_dependent_phase"] = False
    _compare(oracle, sc)


def test_colors_and_cov_precomp(oracle):
    sc = Hh.small_scene(P=120, W=48, H=32, scale_lo=0.02, scale_hi=0.15)
    f0, _ = Hh.run_oracle(oracle, sc, backward=False)
    rng = np.random.default_rng(2)
    cov = f0.geom["cov3D"].copy()
    cov[f0.radii <= 0] = np.array([1e-4, 0, 0, 1e-4, 0, 1e-4], np.float32)
    inputs = dict(shs=None, colors_precomp=rng.random((120, 3)).astype(np.float32), scales=None, rotations=None,
                  cov3D_precomp=cov)
    _compare(oracle, sc, inputs=inputs, check=("means3D", "opacities", "shs_p", "colors_precomp", "cov3D_precomp"))


def test_early_termination_and_frustum_clamp(oracle):
    # opaque, overlapping splats (pixels terminate at T < 1e-4) and centres far outside the
    # frustum (t.x/t.z clamp with x_grad_mul = 0)
    sc = Hh.small_scene(P=2000, W=32, H=32, scale_lo=0.05, scale_hi=0.3, opacity=0.95, spread=1.6)
    f, b = _compare(oracle, sc)
    ln = f.ranges[:, 1] - f.ranges[:, 0]
    assert f.img["n_contrib"].max() < ln.max(), "scene does not exercise early termination"
    g = f.geom
    tx = np.abs((g["means2D"][:, 0] + 0.5) / 32 * 2 - 1)
    assert ((f.radii > 0) & (tx > 1.3)).any(), "scene does not exercise the frustum clamp"
