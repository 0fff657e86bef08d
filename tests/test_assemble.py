"""Fused input assembly: oracle pins on CPU, HIP parity on the GPU."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import assemble_ref as R   # noqa: E402


def make_case(P, M=16, M_p=16, frac=0.3, seed=0, offsets="tensor", mask=None):
    rng = np.random.default_rng(seed)
    f = lambda *s: rng.normal(0, 1, s).astype(np.float32)
    m = (rng.random(P) < frac) if mask is None else mask
    nd = int(m.sum())
    raw = f(P, 4)
    c = dict(xyz=f(P, 3), ssp=f(P, 3), opacity=rng.random((P, 1)).astype(np.float32), scaling=np.exp(f(P, 3) * 0.3),
             rotation=raw / np.linalg.norm(raw, axis=1, keepdims=True), rotation_raw=raw, fc=f(P, M, 3), fp=f(P, M_p, 2),
             mask=m)
    if offsets == "tensor":
        c.update(d_xyz=f(nd, 3) * 0.1, d_rot=f(nd, 4) * 0.1, d_sh=f(nd, M, 3) * 0.1, d_sh_p=f(nd, M_p, 2) * 0.1)
    elif offsets == "float":
        c.update(d_xyz=0.0, d_rot=0.0, d_sh=0.0, d_sh_p=0.0)
    else:   # the deformation network returns zeros for d_rot / d_sh_p (utils/time_utils.py:127)
        c.update(d_xyz=f(nd, 3) * 0.1, d_rot=0.0, d_sh=f(nd, M, 3) * 0.1, d_sh_p=0.0)
    return c


ORDER = ["xyz", "ssp", "opacity", "scaling", "rotation", "rotation_raw", "fc", "fp", "mask", "d_xyz", "d_rot", "d_sh", "d_sh_p"]
OUTS = ["means3D", "means2D", "opacity", "scales", "rotations", "shs", "shs_p"]


def tensors(c, dev, grad):
    out = []
    for k in ORDER:
        v = c[k]
        if isinstance(v, np.ndarray):
            t = torch.tensor(v, device=dev)
            if grad and t.dtype == torch.float32:
                t.requires_grad_(True)
            out.append(t)
        else:
            out.append(v)
    return out


def run(fn, c, dev, regions, grad=True, seed=7, **kw):
    args = tensors(c, dev, grad)
    outs = fn(*args, render_regions=regions, **kw)
    grads = None
    if grad:
        g = np.random.default_rng(seed)
        loss = sum((o * torch.tensor(g.normal(0, 1, tuple(o.shape)).astype(np.float32), device=dev)).sum() for o in outs)
        loss.backward()
        grads = {k: (a.grad.cpu().numpy() if isinstance(a, torch.Tensor) and a.grad is not None else None)
                 for k, a in zip(ORDER, args)}
    return [o.detach().cpu().numpy() for o in outs], grads


@pytest.mark.parametrize("regions", [("static", "dynamic"), ("static",), ("dynamic",), ()])
@pytest.mark.parametrize("offsets", ["tensor", "float"])
def test_eager_restatement_equals_loops(regions, offsets):
    c = make_case(257, M=4, M_p=4, seed=3, offsets=offsets)
    got, _ = run(R.assemble_eager, c, "cpu", regions, grad=False)
    ref = R.assemble_loops(*[c[k] for k in ORDER], render_regions=regions)
    for name, a, b in zip(OUTS, got, ref):
        np.testing.assert_allclose(a, b, rtol=2e-7, atol=0, err_msg=name)


def test_eager_gradients_reach_every_source():
    c = make_case(100, M=4, M_p=4, seed=5)
    _, g = run(R.assemble_eager, c, "cpu", ("static", "dynamic"))
    m = c["mask"]
    assert np.abs(g["rotation"][m]).max() == 0 and np.abs(g["rotation"][~m]).max() > 0
    assert np.abs(g["rotation_raw"][~m]).max() == 0 and np.abs(g["rotation_raw"][m]).max() > 0
    for k in ("d_xyz", "d_rot", "d_sh", "d_sh_p", "xyz", "ssp", "opacity", "scaling", "fc", "fp"):
        assert g[k] is not None and np.abs(g[k]).max() > 0, k


def test_product_raises_on_cpu_tensors():
    from gftorf_amd import assemble_inputs
    c = make_case(8, M=4, M_p=4)
    with pytest.raises(RuntimeError, match="HIP device only"):
        assemble_inputs(*tensors(c, "cpu", False))


CASES = {
    "metric_shape": dict(P=5000, M=16, M_p=16),
    "odd_rows": dict(P=1025, M=9, M_p=9),                 # 27 / 18 floats per row: scalar path
    "one_coeff": dict(P=300, M=1, M_p=1),
    "all_static": dict(P=2049, M=16, M_p=16, frac=0.0),
    "all_dynamic": dict(P=2047, M=16, M_p=16, frac=1.0),
    "mlp_zeros": dict(P=3000, M=16, M_p=16, offsets="mlp"),
    "float_offsets": dict(P=3000, M=16, M_p=16, offsets="float"),
}


@pytest.mark.gpu
@pytest.mark.parametrize("regions", [("static", "dynamic"), ("static",), ("dynamic",), ()])
@pytest.mark.parametrize("name", list(CASES))
def test_hip_assembly_vs_oracle(name, regions, gpu):
    from gftorf_amd import assemble_inputs
    c = make_case(seed=11, **CASES[name])
    # with nothing rendered the eager outputs are constants (no graph): every gradient is zero
    ref, rg = run(R.assemble_eager, c, "cpu", regions, grad=bool(regions))
    got, gg = run(assemble_inputs, c, gpu, regions, validate=True)
    if not regions:
        rg = {k: None for k in ORDER}
    for n, a, b in zip(OUTS, ref, got):
        if n == "rotations":
            np.testing.assert_allclose(b, a, rtol=3e-7, atol=1e-7, err_msg=n)     # sqrt / divide rounding
        else:
            np.testing.assert_array_equal(b, a, err_msg=n)                       # copies and single adds
    for k in ORDER:
        if rg[k] is None:
            assert gg[k] is None or gg[k].size == 0 or np.abs(gg[k]).max() == 0, k
            continue
        if k == "rotation_raw" or k == "d_rot":
            np.testing.assert_allclose(gg[k], rg[k], rtol=2e-5, atol=2e-6, err_msg=k)
        else:
            np.testing.assert_array_equal(gg[k], rg[k], err_msg=k)


@pytest.mark.gpu
def test_hip_assembly_edge_cases(gpu):
    from gftorf_amd import assemble_inputs
    # no Gaussians
    c = make_case(0, M=16, M_p=16)
    outs = assemble_inputs(*tensors(c, gpu, False))
    assert [tuple(o.shape) for o in outs] == [(0, 3), (0, 3), (0, 1), (0, 3), (0, 4), (0, 16, 3), (0, 16, 2)]
    # wrong row count of the offsets is reported like the reference's masked assignment would
    c = make_case(500, M=4, M_p=4, seed=2)
    c["d_xyz"] = c["d_xyz"][:-1]
    with pytest.raises(RuntimeError, match="shape mismatch"):
        assemble_inputs(*tensors(c, gpu, False), validate=True)
    # ... and without the blocking check: every d_* one row short of what the mask selects (the default path of a hot
    # loop).  Nothing is read or written past the tensors; the Gaussian without an offset row comes out as NaN in
    # every output an offset enters, its gradients are zero, everybody else is unchanged.
    c = make_case(500, M=4, M_p=4, seed=2)
    full, _ = run(assemble_inputs, c, gpu, ("static", "dynamic"), grad=False)
    nd = int(c["mask"].sum())
    last = int(np.nonzero(c["mask"])[0][-1])
    short = dict(c)
    for k in ("d_xyz", "d_rot", "d_sh", "d_sh_p"):
        short[k] = c[k][:nd - 1]
    ts = tensors(short, gpu, True)
    outs = assemble_inputs(*ts)
    for i, (o, f) in enumerate(zip(outs, full)):
        o = o.detach().cpu().numpy()
        keep = np.ones(500, bool)
        keep[last] = False
        np.testing.assert_array_equal(o[keep], f[keep])
        if i in (0, 4, 5, 6):          # means3D, rotations, shs, shs_p take an offset
            assert np.isnan(o[last]).all(), i
    torch.autograd.backward(outs, [torch.ones_like(o) for o in outs])
    torch.cuda.synchronize()
    assert all(t.grad is None or torch.isfinite(t.grad).all() for t in ts if isinstance(t, torch.Tensor) and t.is_floating_point())
    # offset tensors that disagree on the row count are refused on the host
    bad = dict(c)
    bad["d_rot"] = c["d_rot"][:-3]
    with pytest.raises(RuntimeError, match="disagree"):
        assemble_inputs(*tensors(bad, gpu, False))
    # a degenerate dynamic quaternion takes the clamped branch of normalize
    c = make_case(64, M=4, M_p=4, seed=4, offsets="float", mask=np.ones(64, bool))
    c["rotation_raw"][:] = 0
    ref, _ = run(R.assemble_eager, c, "cpu", ("dynamic",), grad=False)
    got, _ = run(assemble_inputs, c, gpu, ("dynamic",), grad=False)
    np.testing.assert_array_equal(got[4], ref[4])


@pytest.mark.gpu
@pytest.mark.parametrize("regions", [("static", "dynamic"), ("static",), ("dynamic",)])
def test_hip_assembly_normalises_the_static_rotations_itself(regions, gpu):
    """rotation=None: the static rows are normalize(rotation_raw) (pc.get_rotation) inside the kernels, forward and
    backward -- against the eager composition normalize -> assembly on the CPU."""
    from gftorf_amd import assemble_inputs
    c = make_case(seed=19, **CASES[list(CASES)[0]])
    c["rotation_raw"][5] = 0.0                                   # a zero quaternion: the clamped denominator
    rng = np.random.default_rng(23)
    weights = None

    def eager(*args, **kw):
        a = list(args)
        a[4] = torch.nn.functional.normalize(a[5])               # pc.get_rotation
        return R.assemble_eager(*a, **kw)

    def fused(*args, **kw):
        a = list(args)
        a[4] = None
        return assemble_inputs(*a, **kw)

    c_ref = dict(c)
    ref, rg = run(eager, c_ref, "cpu", regions)
    got, gg = run(fused, c, gpu, regions, validate=True)
    for n, a, b in zip(OUTS, ref, got):
        if n == "rotations":
            np.testing.assert_allclose(b, a, rtol=3e-7, atol=1e-7, err_msg=n)
        else:
            np.testing.assert_array_equal(b, a, err_msg=n)
    assert gg["rotation"] is None                                 # nothing flows to the tensor that was not given
    for k in ORDER:
        if k == "rotation" or rg[k] is None:
            continue
        if k in ("rotation_raw", "d_rot"):
            # (row 5: g / 1e-12 in both)
            np.testing.assert_allclose(gg[k], rg[k], rtol=2e-5, atol=2e-6 * max(1.0, float(np.abs(rg[k]).max()) * 1e-6), err_msg=k)
        else:
            np.testing.assert_array_equal(gg[k], rg[k], err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("regions", [("static", "dynamic"), ("static",)])
def test_sh_gradients_are_handed_on_without_a_copy(regions, gpu):
    """shs = features (+ d_sh on the dynamic rows): with both regions rendered the features' gradient is the incoming
    gradient itself, and the backward returns THAT tensor (no [P, M, 3] copy); the dynamic rows are gathered for d_sh.
    With a region left out its rows must be zero whatever arrives: a tensor of its own."""
    from gftorf_amd import assemble_inputs
    P, M = 5000, 16
    gen = torch.Generator().manual_seed(3)
    rnd = lambda *s: torch.randn(*s, generator=gen).to(gpu)
    mask = (torch.rand(P, generator=gen) < 0.3).to(gpu)
    nd = int(mask.sum())
    fc, fp = rnd(P, M, 3).requires_grad_(), rnd(P, M, 2).requires_grad_()
    d_sh, d_sh_p = rnd(nd, M, 3).requires_grad_(), rnd(nd, M, 2).requires_grad_()
    xyz, ssp, op, sc, raw = rnd(P, 3), torch.zeros(P, 3, device=gpu), rnd(P, 1), rnd(P, 3), rnd(P, 4)
    outs = assemble_inputs(xyz, ssp, op, sc, None, raw, fc, fp, mask, 0.0, 0.0, d_sh, d_sh_p, render_regions=regions)
    g_shs, g_shp = rnd(P, M, 3), rnd(P, M, 2)
    g_fc, g_fp, g_d, g_dp = torch.autograd.grad([outs[5], outs[6]], [fc, fp, d_sh, d_sh_p], [g_shs, g_shp])
    both = len(regions) == 2
    assert (g_fc.data_ptr() == g_shs.data_ptr()) == both and (g_fp.data_ptr() == g_shp.data_ptr()) == both
    on = torch.ones(P, dtype=torch.bool, device=gpu) if both else ~mask
    assert torch.equal(g_fc, g_shs * on[:, None, None]) and torch.equal(g_fp, g_shp * on[:, None, None])
    assert torch.equal(g_d, g_shs[mask] * float(both)) and torch.equal(g_dp, g_shp[mask] * float(both))


@pytest.mark.gpu
def test_zero_scalar_offsets_return_the_feature_tensors_themselves(gpu):
    """d_sh_p is always the scalar 0.0 from this package's network (the reference's network returns zeros there,
    time_utils.py:127), d_sh too before warm-up ends (train.py:164): with both regions rendered shs / shs_p are then the
    feature tensors, handed back as they are -- and their gradient arrives in the features' .grad."""
    from gftorf_amd import assemble_inputs
    P, M = 4000, 16
    gen = torch.Generator().manual_seed(5)
    rnd = lambda *s: torch.randn(*s, generator=gen).to(gpu)
    mask = (torch.rand(P, generator=gen) < 0.3).to(gpu)
    nd = int(mask.sum())
    fc, fp = rnd(P, M, 3).requires_grad_(), rnd(P, M, 2).requires_grad_()
    d_sh = rnd(nd, M, 3).requires_grad_()
    xyz, ssp, op, sc, raw = rnd(P, 3), torch.zeros(P, 3, device=gpu), rnd(P, 1), rnd(P, 3), rnd(P, 4)
    outs = assemble_inputs(xyz, ssp, op, sc, None, raw, fc, fp, mask, 0.0, 0.0, d_sh, 0.0)
    assert outs[6].data_ptr() == fp.data_ptr() and outs[5].data_ptr() != fc.data_ptr()
    want = fc.detach().clone()
    want[mask] += d_sh.detach()
    assert torch.equal(outs[5], want)
    g_shs, g_shp = rnd(P, M, 3), rnd(P, M, 2)
    torch.autograd.backward([outs[5], outs[6]], [g_shs, g_shp])
    assert torch.equal(fp.grad, g_shp) and torch.equal(fc.grad, g_shs) and torch.equal(d_sh.grad, g_shs[mask])
    # a region left out: zeros there, so a tensor of its own
    outs = assemble_inputs(xyz, ssp, op, sc, None, raw, fc, fp, mask, 0.0, 0.0, 0.0, 0.0, render_regions=("static",))
    assert outs[6].data_ptr() != fp.data_ptr() and not outs[6][mask].any() and torch.equal(outs[6][~mask], fp.detach()[~mask])


@pytest.mark.gpu
@pytest.mark.parametrize("regions", [("static", "dynamic"), ("dynamic",)])
@pytest.mark.parametrize("offsets", ["tensors", "scalars"])
def test_assembly_over_the_models_own_tensors(regions, offsets, gpu):
    """assemble_parameters = assemble_inputs behind pc.get_* (scene/gaussian_model.py:123-153: sigmoid, exp, normalize, the two
    concatenations), forward and backward, in the assembly's own kernels: the same outputs as the eager statements feeding
    assemble_inputs (copies and single adds bit for bit, the activations to an ulp) and the same gradients in the raw tensors."""
    from gftorf_amd import assemble_inputs, assemble_parameters
    P, M = 3001, 16
    gen = torch.Generator().manual_seed(9)
    rnd = lambda *s: torch.randn(*s, generator=gen).to(gpu)
    mask = (torch.rand(P, generator=gen) < 0.4).to(gpu)
    nd = int(mask.sum())
    raw = dict(xyz=rnd(P, 3), opacity=rnd(P, 1), scaling=rnd(P, 3) - 2.0, rotation=rnd(P, 4), f_dc=rnd(P, 1, 3), f_rest=rnd(P, M - 1, 3),
               phase_dc=rnd(P, 1, 1), phase_rest=rnd(P, M - 1, 1), amp_dc=rnd(P, 1, 1), amp_rest=rnd(P, M - 1, 1))
    d = [rnd(nd, 3), rnd(nd, 4), rnd(nd, M, 3), rnd(nd, M, 2)] if offsets == "tensors" else [0.0, 0.0, 0.0, 0.0]
    gout = [rnd(P, 3), rnd(P, 3), rnd(P, 1), rnd(P, 3), rnd(P, 4), rnd(P, M, 3), rnd(P, M, 2)]

    def run(fused):
        leaves = {k: v.clone().requires_grad_() for k, v in raw.items()}
        ssp = torch.zeros(P, 3, device=gpu, requires_grad=True)
        dd = [t.clone().requires_grad_() if torch.is_tensor(t) else t for t in d]
        if fused:
            outs = assemble_parameters(leaves["xyz"], ssp, leaves["opacity"], leaves["scaling"], leaves["rotation"], leaves["f_dc"],
                                       leaves["f_rest"], leaves["phase_dc"], leaves["phase_rest"], leaves["amp_dc"], leaves["amp_rest"],
                                       mask, *dd, render_regions=regions)
        else:
            fc = torch.cat((leaves["f_dc"], leaves["f_rest"]), dim=1)
            fp = torch.cat((torch.cat((leaves["phase_dc"], leaves["phase_rest"]), dim=1),
                            torch.cat((leaves["amp_dc"], leaves["amp_rest"]), dim=1)), dim=2)
            outs = assemble_inputs(leaves["xyz"], ssp, torch.sigmoid(leaves["opacity"]), torch.exp(leaves["scaling"]), None,
                                   leaves["rotation"], fc, fp, mask, *dd, render_regions=regions)
        torch.autograd.backward(list(outs), gout)
        grads = {k: v.grad for k, v in leaves.items()}
        grads["ssp"] = ssp.grad
        for i, t in enumerate(dd):
            if torch.is_tensor(t):
                grads["d%d" % i] = t.grad
        return [o.detach() for o in outs], grads

    o_ref, g_ref = run(False)
    o_got, g_got = run(True)
    for i, (a, b) in enumerate(zip(o_ref, o_got)):
        if i in (2, 3):            # opacity, scales: sigmoid / exp (the same expressions as torch's kernels; an ulp for the library calls)
            torch.testing.assert_close(b, a, rtol=3e-7, atol=0)
        else:
            assert torch.equal(b, a), i
    assert set(g_ref) == set(g_got)
    for k in g_ref:
        if g_ref[k] is None:
            assert g_got[k] is None or not g_got[k].any(), k
        elif k in ("opacity", "scaling"):
            torch.testing.assert_close(g_got[k], g_ref[k], rtol=1e-6, atol=1e-30)
        else:
            assert torch.equal(g_got[k], g_ref[k]), k
