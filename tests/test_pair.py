"""GaussianRasterizerPair (gftorf_amd/pair.py): the colour-camera and ToF-camera calls of one iteration
(gaussian_renderer/__init__.py:107-128) as one autograd node -- forward on two streams, one set of gradient tensors."""
import numpy as np
import pytest
import torch

import helpers as Hh
from gftorf_amd import synth

pytestmark = pytest.mark.gpu


def _two_views(P=3000, W=96, H=64, seed=5, **kw):
    """The same Gaussians seen by two cameras a small baseline apart (colour camera / ToF sensor)."""
    a = Hh.small_scene(P=P, W=W, H=H, seed=seed, w2c=synth.look_at_w2c(0.10, -0.05, 0.02, (0.05, 0.0, 0.1)), **kw)
    b = Hh.small_scene(P=P, W=W, H=H, seed=seed, w2c=synth.look_at_w2c(0.12, -0.05, 0.02, (0.09, 0.0, 0.1)), **kw)
    b["gaussians"] = a["gaussians"]
    b["grads"] = synth.make_pixel_grads(W, H, seed + 100)
    b["bg"] = synth.make_background(W, H, seed + 100)
    b["phase_offset"], b["dc_offset"] = 0.25, 0.02
    a["use_view_dependent_phase"] = False
    return a, b


def _leaves(scene, dev):
    g = scene["gaussians"]
    leaf = {k: torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=True) for k, v in g.items() if v is not None}
    m2 = torch.zeros((g["means3D"].shape[0], 3), device=dev, requires_grad=True)
    return leaf, m2


def _loss(outs, scene, dev, keys=Hh.GRAD_KEYS):
    o = dict(zip(Hh.OUT_NAMES, outs))
    return sum((o[k] * torch.tensor(scene["grads"][k], device=dev)).sum() for k in keys)


def _run_single(a, b, dev, use=(True, True), offsets=False):
    from gftorf_amd import GaussianRasterizer
    leaf, m2 = _leaves(a, dev)
    offs, outs, loss = [], [], 0.0
    for sc, on in ((a, use[0]), (b, use[1])):
        ph = torch.tensor([sc["phase_offset"]], device=dev, requires_grad=True) if offsets else sc["phase_offset"]
        dc = torch.tensor([sc["dc_offset"]], device=dev, requires_grad=True) if offsets else sc["dc_offset"]
        offs.append((ph, dc))
        o = GaussianRasterizer(Hh.gpu_settings(sc, dev, optimize_offsets=offsets))(
            means3D=leaf["means3D"], means2D=m2, opacities=leaf["opacities"], shs=leaf["shs"], shs_p=leaf["shs_p"],
            scales=leaf["scales"], rotations=leaf["rotations"], phase_offset=ph, dc_offset=dc)
        outs.append(o)
        if on:
            loss = loss + _loss(o, sc, dev)
    loss.backward()
    torch.cuda.synchronize()
    return outs, leaf, m2, offs


def _run_pair(a, b, dev, use=(True, True), offsets=False):
    from gftorf_amd import GaussianRasterizerPair
    leaf, m2 = _leaves(a, dev)
    mk = lambda sc, k: torch.tensor([sc[k]], device=dev, requires_grad=True) if offsets else sc[k]
    offs = [(mk(a, "phase_offset"), mk(a, "dc_offset")), (mk(b, "phase_offset"), mk(b, "dc_offset"))]
    oa, ob = GaussianRasterizerPair(Hh.gpu_settings(a, dev, optimize_offsets=offsets), Hh.gpu_settings(b, dev, optimize_offsets=offsets))(
        means3D=leaf["means3D"], means2D=m2, opacities=leaf["opacities"], shs=leaf["shs"], shs_p=leaf["shs_p"],
        scales=leaf["scales"], rotations=leaf["rotations"], phase_offset=(offs[0][0], offs[1][0]),
        dc_offset=(offs[0][1], offs[1][1]))
    assert len(oa) == 11 and len(ob) == 11
    loss = 0.0
    if use[0]:
        loss = loss + _loss(oa, a, dev)
    if use[1]:
        loss = loss + _loss(ob, b, dev)
    loss.backward()
    torch.cuda.synchronize()
    return (oa, ob), leaf, m2, offs


def _same_grads(l1, m1, l2, m2, tol):
    for k in l1:
        x, y = l1[k].grad.cpu().numpy(), l2[k].grad.cpu().numpy()
        assert np.abs(x - y).max() <= tol * (np.abs(x).max() + 1e-30), k
    x, y = m1.grad.cpu().numpy(), m2.grad.cpu().numpy()
    assert np.abs(x - y).max() <= tol * (np.abs(x).max() + 1e-30)
    assert not y[:, 2].any()


@pytest.mark.parametrize("kw", [dict(), dict(P=20000, W=64, H=48, scale_lo=0.03, scale_hi=0.15)], ids=["base", "deep_lists"])
def test_pair_equals_two_single_calls(kw, gpu):
    a, b = _two_views(**kw)
    for rep in range(3):         # first frames size their buffers after a blocking read, later ones from hints
        so, sl, sm, _ = _run_single(a, b, gpu)
        po, pl, pm, _ = _run_pair(a, b, gpu)
        for v in range(2):
            for name, x, y in zip(Hh.OUT_NAMES, so[v], po[v]):
                np.testing.assert_array_equal(x.detach().cpu().numpy(), y.detach().cpu().numpy(), err_msg="%s view %d" % (name, v))
        _same_grads(sl, sm, pl, pm, 2e-5)      # float atomics in the render backward: summation order only


def test_pair_against_the_oracle(oracle, gpu):
    a, b = _two_views()
    (fa, ba), (fb, bb) = Hh.run_oracle(oracle, a), Hh.run_oracle(oracle, b)
    (oa, ob), leaf, m2, offs = _run_pair(a, b, gpu, offsets=True)
    for f, o in ((fa, oa), (fb, ob)):
        o = dict(zip(Hh.OUT_NAMES, o))
        np.testing.assert_array_equal(o["radii"].cpu().numpy(), f.radii)
        for k in ["color", "phasor", "depth", "acc", "depth_distortion"]:
            Hh.assert_close(k, f[k], o[k].detach().cpu().numpy(), rtol_max=2e-4, atol=1e-6, frac_bad=1e-3)
    for key, name in [("means3D", "dL_dmeans3D"), ("shs", "dL_dsh"), ("shs_p", "dL_dsh_p"), ("scales", "dL_dscales"),
                      ("rotations", "dL_drotations")]:
        Hh.assert_close(name, ba[name] + bb[name], leaf[key].grad.cpu().numpy(), rtol_max=3e-4)
    Hh.assert_close("dL_dopacity", (ba["dL_dopacity"] + bb["dL_dopacity"]).reshape(-1), leaf["opacities"].grad.cpu().numpy().reshape(-1),
                    rtol_max=3e-4)
    Hh.assert_close("dL_dmeans2D", ba["dL_dmeans2D"] + bb["dL_dmeans2D"], m2.grad.cpu().numpy(), rtol_max=3e-4)
    # each view's own scalar offsets
    for v, bk in ((0, ba), (1, bb)):
        Hh.assert_close("dL_dphase_offset", bk["dL_dphase_offset"], offs[v][0].grad.cpu().numpy(), rtol_max=3e-4, atol=1e-5)
        Hh.assert_close("dL_ddc_offset", bk["dL_ddc_offset"], offs[v][1].grad.cpu().numpy(), rtol_max=3e-4, atol=1e-5)


@pytest.mark.parametrize("use", [(True, False), (False, True)], ids=["only_view_a", "only_view_b"])
def test_pair_with_one_view_in_the_loss(use, gpu):
    """torf.json has lambda_color = 0: the colour-camera call reaches no loss and autograd never runs its backward
    (train.py:206).  The pair then runs only the other view's backward, which writes (not adds) the gradients."""
    a, b = _two_views()
    _, sl, sm, _ = _run_single(a, b, gpu, use=use)
    _, pl, pm, _ = _run_pair(a, b, gpu, use=use)
    _same_grads(sl, sm, pl, pm, 2e-5)


def test_pair_empty_and_forward_only(gpu):
    from gftorf_amd import GaussianRasterizerPair
    a, b = _two_views(P=50)
    z = lambda *s: torch.zeros(s, device=gpu)
    oa, ob = GaussianRasterizerPair(Hh.gpu_settings(a, gpu), Hh.gpu_settings(b, gpu))(
        means3D=z(0, 3), means2D=z(0, 3), opacities=z(0, 1), shs=z(0, 16, 3), shs_p=z(0, 16, 2), scales=z(0, 3), rotations=z(0, 4))
    assert oa[0].shape == (3, 64, 96) and not oa[0].any() and ob[10].numel() == 0
    with torch.no_grad():
        (po, _, _, _) = _run_pair_nograd(a, b, gpu)
    so = [Hh.run_gpu(sc, gpu, backward=False)[0] for sc in (a, b)]
    for v in range(2):
        for name, y in zip(Hh.OUT_NAMES, po[v]):
            np.testing.assert_array_equal(so[v][name], y.cpu().numpy(), err_msg=name)


def _run_pair_nograd(a, b, dev):
    from gftorf_amd import GaussianRasterizerPair
    t = lambda v: torch.tensor(v, dtype=torch.float32, device=dev)
    g = a["gaussians"]
    out = GaussianRasterizerPair(Hh.gpu_settings(a, dev), Hh.gpu_settings(b, dev))(
        means3D=t(g["means3D"]), means2D=torch.zeros((g["means3D"].shape[0], 3), device=dev), opacities=t(g["opacities"]),
        shs=t(g["shs"]), shs_p=t(g["shs_p"]), scales=t(g["scales"]), rotations=t(g["rotations"]),
        phase_offset=(a["phase_offset"], b["phase_offset"]), dc_offset=(a["dc_offset"], b["dc_offset"]))
    torch.cuda.synchronize()
    return out, None, None, None


@pytest.mark.parametrize("variant", ["colors_cov3d_precomp", "sh_degree1_4coeff"])
def test_pair_operator_variants(variant, oracle, gpu):
    """The second view's backward adds to the first one's tensors on every gradient the operator returns: precomputed
    colours and 3D covariances (dL_dcolors, dL_dcov3D rows), and SH rows that are not whole 16-coefficient rows (the
    unstaged row path of k_preprocess_bwd)."""
    from gftorf_amd import GaussianRasterizer, GaussianRasterizerPair
    if variant == "colors_cov3d_precomp":
        a, b = _two_views(P=600, tof=False)
        f0, _ = Hh.run_oracle(oracle, a, backward=False)
        rng = np.random.default_rng(3)
        cov = f0.geom["cov3D"].copy()
        cov[f0.radii <= 0] = np.array([1e-4, 0, 0, 1e-4, 0, 1e-4], np.float32)
        g = dict(a["gaussians"], shs=None, shs_p=None, scales=None, rotations=None,
                 colors_precomp=rng.random((600, 3)).astype(np.float32), cov3D_precomp=cov)
    else:
        a, b = _two_views(P=600, D=1, sh_coeffs=4)
        g = dict(a["gaussians"])
    a["gaussians"] = b["gaussians"] = g
    keys = [k for k, v in g.items() if v is not None]

    def run(pair):
        leaf = {k: torch.tensor(g[k], dtype=torch.float32, device=gpu, requires_grad=True) for k in keys}
        m2 = torch.zeros((600, 3), device=gpu, requires_grad=True)
        kw = dict(means3D=leaf["means3D"], means2D=m2, opacities=leaf["opacities"], shs=leaf.get("shs"), shs_p=leaf.get("shs_p"),
                  colors_precomp=leaf.get("colors_precomp"), scales=leaf.get("scales"), rotations=leaf.get("rotations"),
                  cov3D_precomp=leaf.get("cov3D_precomp"))
        sa, sb = Hh.gpu_settings(a, gpu), Hh.gpu_settings(b, gpu)
        if pair:
            oa, ob = GaussianRasterizerPair(sa, sb)(phase_offset=(a["phase_offset"], b["phase_offset"]),
                                                    dc_offset=(a["dc_offset"], b["dc_offset"]), **kw)
        else:
            oa = GaussianRasterizer(sa)(phase_offset=a["phase_offset"], dc_offset=a["dc_offset"], **kw)
            ob = GaussianRasterizer(sb)(phase_offset=b["phase_offset"], dc_offset=b["dc_offset"], **kw)
        (_loss(oa, a, gpu) + _loss(ob, b, gpu)).backward()
        torch.cuda.synchronize()
        return (oa, ob), leaf, m2

    so, sl, sm = run(False)
    po, pl, pm = run(True)
    for v in range(2):
        for name, x, y in zip(Hh.OUT_NAMES, so[v], po[v]):
            np.testing.assert_array_equal(x.detach().cpu().numpy(), y.detach().cpu().numpy(), err_msg="%s view %d" % (name, v))
    _same_grads(sl, sm, pl, pm, 2e-5)
    # and against the oracle: the sum of both views' gradients
    (_, ba), (_, bb) = Hh.run_oracle(oracle, a), Hh.run_oracle(oracle, b)
    names = dict(means3D="dL_dmeans3D", shs="dL_dsh", shs_p="dL_dsh_p", scales="dL_dscales", rotations="dL_drotations",
                 colors_precomp="dL_dcolors", cov3D_precomp="dL_dcov3D")
    for k in keys:
        if k in names:
            ref = (ba[names[k]] + bb[names[k]]).reshape(g[k].shape)
            Hh.assert_close(names[k], ref, pl[k].grad.cpu().numpy(), rtol_max=3e-4)


def test_pair_with_strided_camera_matrices_and_a_noncontiguous_input(gpu):
    """The reference builds its camera matrices as `.transpose(0, 1)` views (scene/cameras.py:121-129), which the
    forward copies on every call, on torch's current stream: view B's kernels (side stream) must be ordered behind those
    copies, not only behind what was queued before the pair call.  Large inputs make the window wide enough to see."""
    from gftorf_amd import GaussianRasterizer, GaussianRasterizerPair
    a, b = _two_views(P=60000, W=128, H=96, scale_lo=0.01, scale_hi=0.05)
    dev = gpu

    def strided(sc):
        s = Hh.gpu_settings(sc, dev)
        # the same values with strides (1, 4): what `torch.tensor(M).transpose(0, 1).cuda()` keeps
        return s._replace(viewmatrix=s.viewmatrix.t().contiguous().t(), projmatrix=s.projmatrix.t().contiguous().t())
    sa, sb = strided(a), strided(b)
    assert not sa.viewmatrix.is_contiguous() and not sb.projmatrix.is_contiguous()
    g = a["gaussians"]
    t = lambda v: torch.tensor(v, dtype=torch.float32, device=dev)
    # opacities as a column of a wider tensor, SH rows as a slice of a longer tensor: both are copied by the forward
    wide = torch.cat([t(g["opacities"]), torch.zeros((g["opacities"].shape[0], 3), device=dev)], 1)
    kw = dict(means3D=t(g["means3D"]), means2D=torch.zeros((g["means3D"].shape[0], 3), device=dev), opacities=wide[:, 0:1],
              shs=t(np.concatenate([g["shs"], g["shs"]], 1))[:, :16], shs_p=t(g["shs_p"]), scales=t(g["scales"]),
              rotations=t(g["rotations"]))
    assert not kw["opacities"].is_contiguous() and not kw["shs"].is_contiguous()
    for rep in range(4):
        with torch.no_grad():
            # keep the main stream busy in front of the pair call so that the copies sit behind real work
            junk = torch.randn(4096, 4096, device=dev) @ torch.randn(4096, 4096, device=dev)
            oa, ob = GaussianRasterizerPair(sa, sb)(phase_offset=(a["phase_offset"], b["phase_offset"]),
                                                    dc_offset=(a["dc_offset"], b["dc_offset"]), **kw)
            ra = GaussianRasterizer(sa)(phase_offset=a["phase_offset"], dc_offset=a["dc_offset"], **kw)
            rb = GaussianRasterizer(sb)(phase_offset=b["phase_offset"], dc_offset=b["dc_offset"], **kw)
        torch.cuda.synchronize()
        del junk
        for name, x, y in zip(Hh.OUT_NAMES, tuple(ra) + tuple(rb), tuple(oa) + tuple(ob)):
            np.testing.assert_array_equal(x.cpu().numpy(), y.cpu().numpy(), err_msg="%s (repetition %d)" % (name, rep))
