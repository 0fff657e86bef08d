"""Deformation network: oracle vs the reference's golden vectors (CPU),
HIP path vs oracle (GPU, through the C ABI in include/gftorf_deform.h)."""
import os

import numpy as np
import pytest
import torch

from oracle import deform_ref

# deform.npz: the network as the reference constructs it (scene/deform_model.py:9-16, t_multires = 10: 84 inputs);
# deform_t6.npz: the class signature's defaults (t_multires = 6: 76 inputs)
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "deform.npz")
GOLDEN_T6 = os.path.join(os.path.dirname(__file__), "golden", "deform_t6.npz")
GOLDENS = {10: GOLDEN, 6: GOLDEN_T6}
ARCHS = pytest.mark.parametrize("tm", [10, 6])


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-30))


# ---------------------------------------------------------------------------------------------
# CPU: the oracle is pinned by outputs and autograd gradients of the reference's own module
# ---------------------------------------------------------------------------------------------
@ARCHS
def test_oracle_matches_reference_module_forward(tm):
    g = np.load(GOLDENS[tm])
    assert int(g["kwargs"][3]) == tm
    params = deform_ref.random_params(int(g["seed"]), t_multires=tm)
    d_xyz, d_rot, d_sh, d_sh_p = deform_ref.forward(params, g["x"], g["t"])
    assert d_xyz.shape == g["d_xyz"].shape and d_sh.shape == g["d_sh"].shape == (g["x"].shape[0], 16, 3)
    assert _rel(d_xyz, g["d_xyz"]) < 2e-6
    assert _rel(d_sh, g["d_sh"]) < 2e-6
    # the reference returns zeros for the rotation and phasor offsets (time_utils.py:127)
    assert d_rot.shape == g["d_rot"].shape and not d_rot.any() and not g["d_rot"].any()
    assert d_sh_p.shape == g["d_sh_p"].shape and not d_sh_p.any() and not g["d_sh_p"].any()


@ARCHS
def test_oracle_matches_reference_module_backward(tm):
    g = np.load(GOLDENS[tm])
    params = deform_ref.random_params(int(g["seed"]), t_multires=tm)
    grads = deform_ref.backward(params, g["x"], g["t"], g["g_dxyz"], g["g_dsh"])
    assert sorted(n for n, v in grads.items() if v is None) == sorted(g["grad_none"].tolist())
    seen = 0
    for key in g.files:
        if key.startswith("grad:"):
            assert _rel(grads[key[5:]], g[key]) < 5e-6, key
            seen += 1
        elif key.startswith("grad_s:"):
            assert _rel(grads[key[7:]][::8, ::4], g[key]) < 5e-6, key
            seen += 1
    assert seen == 2 * 8 + 2 * 4


def test_oracle_float32_close_to_float64():
    params = deform_ref.random_params(5)
    rng = np.random.default_rng(6)
    x, t = rng.random((33, 3)).astype(np.float32), rng.random((33, 1)).astype(np.float32)
    a = deform_ref.forward(params, x, t)
    b = deform_ref.forward(params, x, t, dtype=np.float64)
    assert _rel(a[0], b[0]) < 5e-6 and _rel(a[2], b[2]) < 5e-6


def test_eager_torch_restatement_matches_numpy_oracle():
    params = deform_ref.random_params(21)
    rng = np.random.default_rng(22)
    x, t = rng.random((40, 3)).astype(np.float32), rng.random((40, 1)).astype(np.float32)
    pt = {k: torch.tensor(v, requires_grad=True) for k, v in params.items()}
    g_dxyz, g_dsh = rng.normal(size=(40, 3)).astype(np.float32), rng.normal(size=(40, 16, 3)).astype(np.float32)
    d_xyz, d_rot, d_sh, d_sh_p = deform_ref.deform_eager(pt, torch.tensor(x), torch.tensor(t))
    ref = deform_ref.forward(params, x, t, dtype=np.float64)
    assert _rel(d_xyz.detach().numpy(), ref[0]) < 3e-6 and _rel(d_sh.detach().numpy(), ref[2]) < 3e-6
    assert not d_rot.any() and not d_sh_p.any() and d_sh_p.shape == (40, 16, 2)
    ((d_xyz * torch.tensor(g_dxyz)).sum() + (d_sh * torch.tensor(g_dsh)).sum()).backward()
    gref = deform_ref.backward(params, x, t, g_dxyz, g_dsh, dtype=np.float64)
    for k, v in gref.items():
        if v is None:
            assert pt[k].grad is None
        else:
            assert _rel(pt[k].grad.numpy(), v) < 2e-5, k


def test_embedding_layout():
    x = np.array([[0.1, 0.2, 0.3]], np.float32)
    t = np.array([[0.5]], np.float32)
    e10 = deform_ref.embed(x, t)
    assert e10.shape == (1, 84) and deform_ref.IN_CH == 84
    np.testing.assert_array_equal(e10[0, 82:84], [np.sin(t[0, 0] * np.float32(512)), np.cos(t[0, 0] * np.float32(512))])
    e = deform_ref.embed(x, t, t_multires=6)
    assert e.shape == (1, 76)
    np.testing.assert_array_equal(e10[0, :76], e[0])
    np.testing.assert_array_equal(e[0, :3], x[0])
    np.testing.assert_array_equal(e[0, 3:6], np.sin(x[0]))            # frequency 1: sin of all dims, then cos
    np.testing.assert_array_equal(e[0, 6:9], np.cos(x[0]))
    np.testing.assert_array_equal(e[0, 57:60], np.sin(x[0] * np.float32(512)))
    assert e[0, 63] == t[0, 0]
    np.testing.assert_array_equal(e[0, 64:66], [np.sin(t[0, 0]), np.cos(t[0, 0])])
    np.testing.assert_array_equal(e[0, 74:76], [np.sin(t[0, 0] * np.float32(32)), np.cos(t[0, 0] * np.float32(32))])


def test_module_mirrors_reference_state_dict():
    """The constructor call of scene/deform_model.py:9-16 with the values of arguments/__init__.py:66-69 and
    configs/{torf,ftorf}.json: linear.0.weight [256,84], linear.5.weight [256,340], 522 055 parameters."""
    from gftorf_amd.deform import DeformNetwork, REFERENCE_ARCH, reference_network
    g = np.load(GOLDEN)
    assert REFERENCE_ARCH == dict(zip(("D", "W", "xyz_multires", "t_multires", "sh_degree"), g["kwargs"].tolist()))
    net = DeformNetwork(D=8, W=256, xyz_multires=10, t_multires=10, sh_degree=3)
    sd = net.state_dict()
    assert list(sd.keys()) == list(g["param_names"])      # names and ORDER of the reference's module
    assert [";".join(map(str, v.shape)) for v in sd.values()] == list(g["param_shapes"])
    assert {k: tuple(v.shape) for k, v in sd.items()} == deform_ref.param_shapes()
    assert tuple(sd["linear.0.weight"].shape) == (256, 84) and tuple(sd["linear.5.weight"].shape) == (256, 340)
    assert sum(v.numel() for v in sd.values()) == int(g["num_params"]) == 522055
    # a state_dict of the reference's shapes loads (the seeded parameters the fixture's module was given)
    net.load_state_dict({k: torch.tensor(v) for k, v in deform_ref.random_params(int(g["seed"])).items()})
    assert {k: tuple(v.shape) for k, v in reference_network().state_dict().items()} == deform_ref.param_shapes()
    # the class signature's own defaults (t_multires = 6, time_utils.py:57)
    g6 = np.load(GOLDEN_T6)
    sd6 = DeformNetwork().state_dict()
    assert list(sd6.keys()) == list(g6["param_names"])
    assert [";".join(map(str, v.shape)) for v in sd6.values()] == list(g6["param_shapes"])
    assert sum(v.numel() for v in sd6.values()) == int(g6["num_params"]) == 517959
    with pytest.raises(RuntimeError):
        net.load_state_dict(sd6)                            # [256,76] does not fit [256,84]
    with pytest.raises(NotImplementedError):
        DeformNetwork(W=128)
    with pytest.raises(NotImplementedError):
        DeformNetwork(xyz_multires=10, t_multires=17)       # 98 encoded inputs > 96
    DeformNetwork(xyz_multires=10, t_multires=16)           # 96: the most the kernels hold


def test_product_fails_loudly_without_a_device():
    from gftorf_amd.deform import DeformNetwork
    net = DeformNetwork()
    with pytest.raises(RuntimeError, match="HIP device only"):
        net(torch.zeros(4, 3), torch.zeros(4, 1))


# ---------------------------------------------------------------------------------------------
# GPU: the HIP path (through the C ABI) against the oracle
# ---------------------------------------------------------------------------------------------
def _net(seed, dev, tm=10):
    from gftorf_amd.deform import DeformNetwork
    params = deform_ref.random_params(seed, t_multires=tm)
    net = DeformNetwork(D=8, W=256, xyz_multires=10, t_multires=tm, sh_degree=3)
    net.load_state_dict({k: torch.tensor(v) for k, v in params.items()})
    return net.to(dev), params


def _inputs(n, seed, shared_t):
    rng = np.random.default_rng(seed)
    x = rng.random((n, 3)).astype(np.float32)
    t = np.full((n, 1), rng.random(), np.float32) if shared_t else rng.random((n, 1)).astype(np.float32)
    return x, t


FWD_TOL = 3e-6      # of the max-norm, fp32 sums of 84..340 products per layer against float64
BWD_TOL = 2e-5      # weight gradients are sums over all points as well


@pytest.mark.gpu
@ARCHS
def test_forward_matches_reference_golden_vectors(tm):
    dev = torch.device("cuda:0")
    g = np.load(GOLDENS[tm])
    net, _ = _net(int(g["seed"]), dev, tm)
    with torch.no_grad():
        d_xyz, d_rot, d_sh, d_sh_p = net(torch.tensor(g["x"], device=dev), torch.tensor(g["t"], device=dev))
    assert _rel(d_xyz.cpu().numpy(), g["d_xyz"]) < FWD_TOL
    assert _rel(d_sh.cpu().numpy(), g["d_sh"]) < FWD_TOL
    assert d_rot.shape == g["d_rot"].shape and not d_rot.any()
    assert d_sh_p.shape == g["d_sh_p"].shape and not d_sh_p.any()


@pytest.mark.gpu
@ARCHS
def test_backward_matches_reference_golden_vectors(tm):
    dev = torch.device("cuda:0")
    g = np.load(GOLDENS[tm])
    net, _ = _net(int(g["seed"]), dev, tm)
    d_xyz, _, d_sh, _ = net(torch.tensor(g["x"], device=dev), torch.tensor(g["t"], device=dev))
    ((d_xyz * torch.tensor(g["g_dxyz"], device=dev)).sum() + (d_sh * torch.tensor(g["g_dsh"], device=dev)).sum()).backward()
    grads = {k: p.grad for k, p in net.named_parameters()}
    assert sorted(k for k, v in grads.items() if v is None) == sorted(g["grad_none"].tolist())
    for key in g.files:
        if key.startswith("grad:"):
            assert _rel(grads[key[5:]].cpu().numpy(), g[key]) < BWD_TOL, key
        elif key.startswith("grad_s:"):
            assert _rel(grads[key[7:]].cpu().numpy()[::8, ::4], g[key]) < BWD_TOL, key


@pytest.mark.gpu
@pytest.mark.parametrize("n,shared_t,tm", [(1, False, 10), (63, True, 10), (64, False, 10), (65, True, 10), (1000, False, 10),
                                           (5000, True, 10), (65, False, 6), (1000, True, 6), (333, False, 0), (333, True, 16)])
def test_forward_against_oracle(n, shared_t, tm):
    dev = torch.device("cuda:0")
    net, params = _net(11, dev, tm)
    x, t = _inputs(n, 100 + n, shared_t)
    tt = torch.tensor(t[:1], device=dev).expand(n, -1) if shared_t else torch.tensor(t, device=dev)   # gaussian_model.py:171
    with torch.no_grad():
        d_xyz, d_rot, d_sh, d_sh_p = net(torch.tensor(x, device=dev), tt)
    ref = deform_ref.forward(params, x, t, dtype=np.float64)
    assert d_xyz.shape == (n, 3) and d_sh.shape == (n, 16, 3) and d_rot.shape == (n, 4) and d_sh_p.shape == (n, 16, 2)
    assert _rel(d_xyz.cpu().numpy(), ref[0]) < FWD_TOL
    assert _rel(d_sh.cpu().numpy(), ref[2]) < FWD_TOL
    # the float32 oracle is no closer to float64 than the device is, within a small factor
    o32 = deform_ref.forward(params, x, t)
    assert _rel(d_sh.cpu().numpy(), ref[2]) < 4 * max(_rel(o32[2], ref[2]), 2e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("n,tm", [(48, 10), (333, 10), (5000, 10), (333, 6), (200, 16)])
def test_backward_against_oracle(n, tm):
    dev = torch.device("cuda:0")
    net, params = _net(12, dev, tm)
    x, t = _inputs(n + n // 8, 200 + n, False)
    # a ReLU whose input is within rounding of zero may switch differently in fp32 and in float64 and
    # changes the gradient by a finite amount: keep the n points that are furthest from such an edge
    keep = np.sort(np.argsort(-deform_ref.relu_margin(params, x, t))[:n])
    assert deform_ref.relu_margin(params, x, t)[keep].min() > 1e-6
    x, t = x[keep], t[keep]
    rng = np.random.default_rng(n)
    g_dxyz, g_dsh = rng.normal(size=(n, 3)).astype(np.float32), rng.normal(size=(n, 16, 3)).astype(np.float32)
    d_xyz, _, d_sh, _ = net(torch.tensor(x, device=dev), torch.tensor(t, device=dev))
    ((d_xyz * torch.tensor(g_dxyz, device=dev)).sum() + (d_sh * torch.tensor(g_dsh, device=dev)).sum()).backward()
    ref = deform_ref.backward(params, x, t, g_dxyz, g_dsh, dtype=np.float64)
    for name, p in net.named_parameters():
        if ref[name] is None:
            assert p.grad is None, name
        else:
            assert p.grad.shape == p.shape
            assert _rel(p.grad.cpu().numpy(), ref[name]) < BWD_TOL, name


@pytest.mark.gpu
def test_backward_with_one_output_unused_and_accumulation():
    dev = torch.device("cuda:0")
    net, params = _net(13, dev)
    n = 200
    x, t = _inputs(n, 7, True)
    xs, ts = torch.tensor(x, device=dev), torch.tensor(t, device=dev)
    g_dxyz = np.random.default_rng(1).normal(size=(n, 3)).astype(np.float32)
    # only d_xyz reaches the loss (train.py:171-176 queries discard the other results)
    d_xyz, _, _, _ = net(xs, ts)
    (d_xyz * torch.tensor(g_dxyz, device=dev)).sum().backward()
    ref = deform_ref.backward(params, x, t, g_dxyz, np.zeros((n, 16, 3), np.float32), dtype=np.float64)
    assert _rel(net.linear[3].weight.grad.cpu().numpy(), ref["linear.3.weight"]) < BWD_TOL
    assert _rel(net.xyz_warp.weight.grad.cpu().numpy(), ref["xyz_warp.weight"]) < BWD_TOL
    assert not net.r.weight.grad.any() and not net.b.bias.grad.any()
    # a second query accumulates into .grad like any autograd op (2-4 queries per iteration)
    first = net.linear[3].weight.grad.clone()
    d_xyz, _, _, _ = net(xs, ts)
    (d_xyz * torch.tensor(g_dxyz, device=dev)).sum().backward()
    torch.testing.assert_close(net.linear[3].weight.grad, 2 * first, rtol=1e-6, atol=0)


@pytest.mark.gpu
def test_inference_path_and_determinism():
    dev = torch.device("cuda:0")
    net, _ = _net(14, dev)
    x, t = _inputs(777, 3, False)
    xs, ts = torch.tensor(x, device=dev), torch.tensor(t, device=dev)
    with torch.no_grad():
        a = net(xs, ts)
    b = net(xs, ts)
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])      # saving activations changes no result
    g = torch.ones_like(b[2])
    b[2].backward(g)
    g1 = [p.grad.clone() for p in net.parameters() if p.grad is not None]
    net.zero_grad()
    c = net(xs, ts)
    c[2].backward(g)
    g2 = [p.grad for p in net.parameters() if p.grad is not None]
    assert all(torch.equal(u, v) for u, v in zip(g1, g2))           # split sums + one reduction: no atomics


@pytest.mark.gpu
def test_zero_points():
    dev = torch.device("cuda:0")
    net, _ = _net(15, dev)
    d_xyz, d_rot, d_sh, d_sh_p = net(torch.zeros((0, 3), device=dev), torch.zeros((0, 1), device=dev))
    assert d_xyz.shape == (0, 3) and d_sh.shape == (0, 16, 3) and d_rot.shape == (0, 4) and d_sh_p.shape == (0, 16, 2)
    (d_xyz.sum() + d_sh.sum()).backward()
    assert not net.linear[0].weight.grad.any() and not net.r.bias.grad.any()


@pytest.mark.gpu
def test_full_size_properties():
    """1 M points (every Gaussian of the metric frame dynamic): size-independent properties instead of the oracle."""
    dev = torch.device("cuda:0")
    net, params = _net(16, dev)
    n = 1_000_000
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.rand((n, 3), generator=g).to(dev)
    t = torch.rand((n, 1), generator=g).to(dev)
    with torch.no_grad():
        full = net(x, t)
    # (1) a point's result does not depend on the batch it is in or on its place in it (bit-exact)
    idx = torch.randperm(n, generator=g)[:5000].to(dev)
    with torch.no_grad():
        part = net(x[idx], t[idx])
    assert torch.equal(part[0], full[0][idx]) and torch.equal(part[2], full[2][idx])
    # ... and a sample of it agrees with the oracle
    sel = idx[:256].cpu().numpy()
    ref = deform_ref.forward(params, x.cpu().numpy()[sel], t.cpu().numpy()[sel], dtype=np.float64)
    assert _rel(full[0][idx[:256]].cpu().numpy(), ref[0]) < FWD_TOL and _rel(full[2][idx[:256]].cpu().numpy(), ref[2]) < FWD_TOL

    def grads(sl, scale):
        net.zero_grad(set_to_none=True)
        d_xyz, _, d_sh, _ = net(x[sl], t[sl])
        torch.autograd.backward([d_xyz, d_sh], [g_dxyz[sl] * scale, g_dsh[sl] * scale])
        return {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}

    g_dxyz = torch.randn((n, 3), generator=g).to(dev)
    g_dsh = torch.randn((n, 16, 3), generator=g).to(dev)
    whole = grads(slice(0, n), 1.0)
    # (2) the backward is linear in the upstream gradient: a power-of-two scale is exact
    twice = grads(slice(0, n), 2.0)
    assert all(torch.equal(twice[k], 2 * whole[k]) for k in whole)
    # (3) ... and additive over a partition of the points (different split sums: fp32 tolerance)
    h = 437_123
    a, b = grads(slice(0, h), 1.0), grads(slice(h, n), 1.0)
    for k in whole:
        assert _rel((a[k] + b[k]).cpu().numpy(), whole[k].cpu().numpy()) < 2e-5, k
    assert len(whole) == 24


@pytest.mark.gpu
def test_fp32_mfma_walks_still_match():
    """GFT_DEFORM_BF16X3=0 selects the walks on v_mfma_f32_32x32x2_f32 (the default multiplies three bf16 planes per
    operand, six MFMAs per product).  The switch is read once per process: the oracle comparisons run again in a child."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_deform.py"), "-q", "-m", "gpu", "-x",
                        "-k", "against_oracle or golden_vectors or determinism"],
                       env=dict(os.environ, GFT_DEFORM_BF16X3="0"), cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.gpu
def test_bf16_backward_walk_still_matches():
    """GFT_DEFORM_BWD_FP16=0 keeps the backward walk on three bf16 planes (six MFMAs per product; the default since round 6
    multiplies two fp16 planes, three MFMAs): it is also what runs when a weight does not fit the fp16 planes."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_deform.py"), "-q", "-m", "gpu", "-x",
                        "-k", "backward_against_oracle or backward_matches_reference or gradient_only"],
                       env=dict(os.environ, GFT_DEFORM_BWD_FP16="0"), cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.gpu
def test_backward_carries_any_gradient_magnitude():
    """The fp16 planes of the backward walk hold each point's gradient row times its own power of two (k_deform_bwd_h), so
    the upstream gradient's magnitude is free: times 2^-60 or 2^40 the parameter gradients are the SAME BITS times that
    power (every scaling in the walk is exact), and rows of very different magnitude in one tile keep their own digits."""
    dev = torch.device("cuda:0")
    net, params = _net(31, dev)
    n = 3000
    x, t = _inputs(n + 400, 17, False)
    far = np.sort(np.argsort(-deform_ref.relu_margin(params, x, t))[:n])
    x, t = x[far], t[far]
    rng = np.random.default_rng(8)
    g_dxyz, g_dsh = rng.normal(size=(n, 3)).astype(np.float32), rng.normal(size=(n, 16, 3)).astype(np.float32)
    xs, ts = torch.tensor(x, device=dev), torch.tensor(t, device=dev)

    def grads(gx, gs):
        net.zero_grad(set_to_none=True)
        d_xyz, _, d_sh, _ = net(xs, ts)
        torch.autograd.backward([d_xyz, d_sh], [torch.tensor(gx, device=dev), torch.tensor(gs, device=dev)])
        return {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}

    base = grads(g_dxyz, g_dsh)
    for k in (-60, 40):
        f = np.float32(2.0 ** k)
        scaled = grads(g_dxyz * f, g_dsh * f)
        for name, g in base.items():
            assert torch.equal(scaled[name], g * float(f)), (k, name)
    # rows of one 64-point tile 2^36 apart: against float64, and the small rows alone (the large ones zero) as exactly as before
    row = (np.float32(2.0) ** rng.integers(-18, 19, size=n)).astype(np.float32)
    mixed = grads(g_dxyz * row[:, None], g_dsh * row[:, None, None])
    ref = deform_ref.backward(params, x, t, g_dxyz * row[:, None], g_dsh * row[:, None, None], dtype=np.float64)
    for name, g in mixed.items():
        assert _rel(g.cpu().numpy(), ref[name]) < BWD_TOL, name
    small = np.where(row < 2.0 ** -6, row, 0).astype(np.float32)
    part = grads(g_dxyz * small[:, None], g_dsh * small[:, None, None])
    ref = deform_ref.backward(params, x, t, g_dxyz * small[:, None], g_dsh * small[:, None, None], dtype=np.float64)
    for name, g in part.items():
        assert _rel(g.cpu().numpy(), ref[name]) < BWD_TOL, name


@pytest.mark.gpu
@pytest.mark.parametrize("frac,n", [(0.1, 40_000), (0.0, 20_000), (0.5, 12_345), (0.9, 20_000)])
def test_backward_over_rows_with_a_gradient_only(frac, n):
    """In a training iteration only the Gaussians some pixel blended hand a non-zero (d_xyz, d_sh) gradient row to the
    network (6-14 % of the queried points); a row whose upstream gradient is zero has dz = 0 in every layer.  The
    backward compacts the rows that count and runs on those: weight gradients equal the dense backward's up to
    summation order (2e-5 of the max-norm), for no / few / half / most rows active (the last one keeps the dense path)."""
    from gftorf_amd import deform as D
    dev = torch.device("cuda:0")
    net, _ = _net(21, dev)
    g = torch.Generator(device="cpu").manual_seed(9)
    x, t = torch.rand((n, 3), generator=g).to(dev), torch.rand((n, 1), generator=g).to(dev)
    keep = (torch.rand(n, generator=g) < frac).to(dev)
    g_dxyz = torch.randn((n, 3), generator=g).to(dev) * keep[:, None]
    g_dsh = torch.randn((n, 16, 3), generator=g).to(dev) * keep[:, None, None]
    # rows with a gradient in only one of the two outputs count as well
    if frac > 0:
        only_xyz = torch.nonzero(keep)[:5, 0]
        g_dsh[only_xyz] = 0

    def grads(sparse):
        # (the blocking selection: rows counted by a host read; the rows counted on the device have their own tests below)
        D.sparse_backward, old = sparse, D.device_row_count
        D.device_row_count = False
        try:
            net.zero_grad(set_to_none=True)
            d_xyz, _, d_sh, _ = net(x, t)
            torch.autograd.backward([d_xyz, d_sh], [g_dxyz, g_dsh])
            return {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}, dict(D.last_backward_stats)
        finally:
            D.sparse_backward, D.device_row_count = True, old
    dense, st_d = grads(False)
    sparse, st_s = grads(True)
    assert st_d == {"points": n, "points_processed": n, "recomputed": False}
    k = int(keep.sum())
    assert st_s["points"] == n and st_s["points_processed"] == (k if k <= 0.6 * n else n)
    assert len(sparse) == len(dense) == 24
    for name in dense:
        if k == 0:
            assert not sparse[name].any() and not dense[name].any(), name
        else:
            assert _rel(sparse[name].cpu().numpy(), dense[name].cpu().numpy()) < 2e-5, name


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["weight", "activation", "input"])
def test_values_beyond_the_fp16_planes_fall_back_to_the_fp32_range(case):
    """The forward walk multiplies on fp16 planes that hold |weight| < 64 and |activation| < 4094.  The reference is
    fp32: a network or an input outside that range must come out as the oracle has it (the bf16 walk takes over inside
    the same call), forward and backward."""
    dev = torch.device("cuda:0")
    params = deform_ref.random_params(21)
    if case == "weight":
        params["linear.3.weight"][5, 7] = 100.0            # times the plane scale: beyond 65504
    elif case == "activation":
        params["linear.2.bias"][11] = 6000.0               # an activation of about 6000
    from gftorf_amd.deform import DeformNetwork
    net = DeformNetwork(D=8, W=256, xyz_multires=10, t_multires=10, sh_degree=3)
    net.load_state_dict({k: torch.tensor(v) for k, v in params.items()})
    net = net.to(dev)
    n = 700
    x, t = _inputs(n + 200, 3, shared_t=False)
    # (gradients against float64: keep the points that are furthest from a ReLU edge, as test_backward_matches_... does)
    far = np.sort(np.argsort(-deform_ref.relu_margin(params, x, t))[:n])
    x, t = x[far], t[far]
    if case == "input":
        x[13, 1] = 5000.0                                   # an unnormalised coordinate
    ref = deform_ref.forward(params, x.astype(np.float64), t.astype(np.float64), dtype=np.float64)
    assert np.isfinite(ref[0]).all() and np.isfinite(ref[2]).all()
    with torch.no_grad():
        out = net(torch.tensor(x, device=dev), torch.tensor(t, device=dev))
    if case == "input":
        # sin / cos of 2^9 * 5000 in fp32: the encoding itself differs from float64 in the high octaves, for the reference
        # too; what is checked is that the row is finite and the other rows are exact
        keep = np.arange(n) != 13
        assert np.isfinite(out[0].cpu().numpy()).all() and np.isfinite(out[2].cpu().numpy()).all()
        assert _rel(out[0].cpu().numpy()[keep], ref[0][keep]) < FWD_TOL and _rel(out[2].cpu().numpy()[keep], ref[2][keep]) < FWD_TOL
        return
    assert _rel(out[0].cpu().numpy(), ref[0]) < FWD_TOL and _rel(out[2].cpu().numpy(), ref[2]) < FWD_TOL
    # the saving forward and the backward behind it
    g_dxyz, g_dsh = np.random.default_rng(4).standard_normal((n, 3)), np.random.default_rng(5).standard_normal((n, 16, 3))
    d_xyz, _, d_sh, _ = net(torch.tensor(x, device=dev), torch.tensor(t, device=dev))
    assert _rel(d_xyz.detach().cpu().numpy(), ref[0]) < FWD_TOL
    torch.autograd.backward([d_xyz, d_sh], [torch.tensor(g_dxyz, device=dev, dtype=torch.float32), torch.tensor(g_dsh, device=dev, dtype=torch.float32)])
    gref = deform_ref.backward(params, x.astype(np.float64), t.astype(np.float64), g_dxyz, g_dsh, dtype=np.float64)
    for name, p in net.named_parameters():
        if p.grad is not None and name in gref:
            assert _rel(p.grad.cpu().numpy(), gref[name]) < BWD_TOL, name


@pytest.mark.gpu
def test_gradients_are_views_of_the_data_parallel_bucket():
    """The backward leaves the parameter gradients as consecutive views of one buffer: flat_grad_bucket returns that
    memory itself (no gather, no scatter back), and what is reduced in it is what the optimizer reads."""
    from gftorf_amd.deform import flat_grad_bucket
    dev = torch.device("cuda:0")
    net, _ = _net(4, dev)
    x, t = _inputs(300, 6, shared_t=True)
    d_xyz, _, d_sh, _ = net(torch.tensor(x, device=dev), torch.tensor(t, device=dev))
    (d_xyz.sum() + d_sh.sum()).backward()
    flat, scatter_back = flat_grad_bucket(net)
    ps = [p for p in net.parameters() if p.grad is not None]
    assert flat.untyped_storage().data_ptr() == ps[0].grad.untyped_storage().data_ptr()
    assert sum(p.numel() for p in ps) <= flat.numel() < sum(p.numel() for p in ps) + 4 * len(ps)
    before = [p.grad.clone() for p in ps]
    flat *= 0.5                                   # stands for the all-reduce + average
    scatter_back(flat)
    for p, b in zip(ps, before):
        assert torch.equal(p.grad, 0.5 * b)
    # a gradient that is NOT in the bucket (set by hand) falls back to the gathered copy
    ps[3].grad = ps[3].grad.clone()
    flat2, scatter2 = flat_grad_bucket(net)
    assert flat2.untyped_storage().data_ptr() != ps[0].grad.untyped_storage().data_ptr() and flat2.numel() == sum(p.numel() for p in ps)
    flat2 *= 2.0
    scatter2(flat2)
    for p, b in zip(ps, before):
        assert torch.equal(p.grad, b)


@pytest.mark.gpu
def test_activations_on_demand_give_the_saved_forwards_gradients():
    """lazy_save: after a backward that used few rows the next forward keeps nothing and its backward recomputes the rows
    with a gradient -- bit-identical to the forward that saved everything; a backward that meets many rows after all
    recomputes all of them and the next forward saves again."""
    from gftorf_amd import deform as D
    dev = torch.device("cuda:0")
    n = 20_000
    x, t = _inputs(n, 8, shared_t=True)
    xt, tt = torch.tensor(x, device=dev), torch.tensor(t, device=dev)
    g = torch.Generator().manual_seed(3)
    g_dxyz = torch.randn((n, 3), generator=g).to(dev)
    g_dsh = torch.randn((n, 16, 3), generator=g).to(dev)
    few = (torch.rand((n,), generator=g) < 0.07).to(dev)
    sparse = (g_dxyz * few[:, None], g_dsh * few[:, None, None])

    def step(net, up):
        net.zero_grad(set_to_none=True)
        d_xyz, _, d_sh, _ = net(xt, tt)
        torch.autograd.backward([d_xyz, d_sh], list(up))
        return {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}, dict(D.last_backward_stats)

    old, old_dev = D.lazy_save, D.device_row_count
    try:
        D.device_row_count = "capture"                       # (eagerly the blocking selection: this test's subject)
        D.lazy_save = False
        ref_net, _ = _net(12, dev)
        ref_sparse, st = step(ref_net, sparse)
        assert not st["recomputed"] and st["points_processed"] == int(few.sum())
        ref_dense, _ = step(ref_net, (g_dxyz, g_dsh))
        D.lazy_save = True
        net, _ = _net(12, dev)
        first, st1 = step(net, sparse)                       # nothing known yet: saves
        assert not st1["recomputed"]
        second, st2 = step(net, sparse)                      # few rows last time: keeps nothing, recomputes them
        assert st2["recomputed"] and st2["points_processed"] == int(few.sum())
        for k in ref_sparse:
            assert torch.equal(first[k], ref_sparse[k]) and torch.equal(second[k], ref_sparse[k]), k
        third, st3 = step(net, (g_dxyz, g_dsh))              # lazy forward, dense gradient: all rows again
        assert st3["recomputed"] and st3["points_processed"] == n
        for k in ref_dense:
            assert torch.equal(third[k], ref_dense[k]), k
        _, st4 = step(net, (g_dxyz, g_dsh))                  # ... and the next forward saves
        assert not st4["recomputed"]
    finally:
        D.lazy_save, D.device_row_count = old, old_dev


@pytest.mark.gpu
@pytest.mark.parametrize("frac,n,shared_t", [(0.07, 20_000, True), (0.5, 12_345, False), (0.0, 9_000, True), (1.0, 10_000, True)])
def test_rows_counted_on_the_device_give_the_blocking_selections_gradients(frac, n, shared_t):
    """gft_deform_backward_rows: the rows with a gradient marked, ranked and counted by kernels, nothing read back, every
    launch sized for the capacity with its extents from the plan a kernel writes.  Same kernels on the same compacted rows
    as the blocking selection (gft_rows_rank's count on the host, then buffers of exactly that size): bit-identical
    gradients, for few / half / no / all rows, one time for all points or one per point."""
    from gftorf_amd import deform as D
    dev = torch.device("cuda:0")
    x, t = _inputs(n, 8, shared_t=shared_t)
    xt, tt = torch.tensor(x, device=dev), torch.tensor(t, device=dev)
    if shared_t:
        tt = tt[:1].expand(n, -1)                                 # one time for all points (stride 0), as gaussian_model.py:171 expands it
    g = torch.Generator().manual_seed(5)
    keep = (torch.rand((n,), generator=g) < frac).to(dev)
    g_dxyz = torch.randn((n, 3), generator=g).to(dev) * keep[:, None]
    g_dsh = torch.randn((n, 16, 3), generator=g).to(dev) * keep[:, None, None]
    if 0 < frac < 1:
        g_dsh[torch.nonzero(keep)[:5, 0]] = 0                     # rows with a gradient in one of the two outputs only

    def step(net):
        net.zero_grad(set_to_none=True)
        d_xyz, _, d_sh, _ = net(xt, tt)
        torch.autograd.backward([d_xyz, d_sh], [g_dxyz, g_dsh])
        return {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}, dict(D.last_backward_stats)

    old = (D.device_row_count, D.lazy_save, D._SPARSE_MAX_FRACTION)
    try:
        # the blocking path on exactly the rows with a gradient: a forward that keeps nothing, rows recomputed
        D.device_row_count, D.lazy_save, D._SPARSE_MAX_FRACTION = False, True, 2.0
        ref_net, _ = _net(12, dev)
        ref_net._save_state = {"fraction": 0.0}
        ref, st = step(ref_net)
        assert st["recomputed"] and st["points_processed"] == int(keep.sum())
        D.device_row_count = True
        net, _ = _net(12, dev)
        got, st = step(net)
        assert st["recomputed"] and int(st["rows_on_device"].item()) == int(keep.sum())
        assert D.backward_stats()["points_processed"] == int(keep.sum())
        assert len(got) == len(ref) == 24
        for k in ref:
            assert torch.equal(got[k], ref[k]), k
            if frac == 0.0:
                assert not got[k].any(), k
    finally:
        D.device_row_count, D.lazy_save, D._SPARSE_MAX_FRACTION = old


@pytest.mark.gpu
def test_a_captured_backward_follows_the_rows_of_every_replay():
    """Forward + backward captured in a HIP graph once (the row count stays on the device), replayed on upstream gradients
    with other rows set: every replay's gradients are those of an eager call on the same gradients, bit for bit, and the
    row count it leaves is that replay's."""
    from gftorf_amd import deform as D
    dev = torch.device("cuda:0")
    n = 16_000
    x, t = _inputs(n, 8, shared_t=True)
    xt, tt = torch.tensor(x, device=dev), torch.tensor(t, device=dev)
    g = torch.Generator().manual_seed(11)
    full_xyz, full_sh = torch.randn((n, 3), generator=g).to(dev), torch.randn((n, 16, 3), generator=g).to(dev)
    masks = [(torch.rand((n,), generator=g) < f).to(dev) for f in (0.3, 0.05, 0.0, 0.8)]
    s_xyz, s_sh = torch.zeros_like(full_xyz), torch.zeros_like(full_sh)
    if os.environ.get("GFT_DEFORM_DEVICE_ROWS", "") in ("", "auto"):
        assert D.device_row_count == "auto"                       # (the default: under capture always)
    net, _ = _net(12, dev)
    params = [p for p in net.parameters()]

    def run():
        d_xyz, _, d_sh, _ = net(xt, tt)
        return torch.autograd.grad([d_xyz, d_sh], [p for p in params if p.requires_grad], [s_xyz, s_sh], allow_unused=True)

    def set_mask(m):
        s_xyz.copy_(full_xyz * m[:, None])
        s_sh.copy_(full_sh * m[:, None, None])
    set_mask(masks[0])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()                                                     # (warm-up outside the capture: allocations, LDS opt-ins)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = run()
        rows = D.last_backward_stats["rows_on_device"]
    old = D.device_row_count
    try:
        for m in masks + [masks[0]]:
            set_mask(m)
            graph.replay()
            got = [None if c is None else c.clone() for c in captured]
            assert int(rows.item()) == int(m.sum())
            D.device_row_count = True
            ref = run()
            D.device_row_count = old
            for a, b in zip(got, ref):
                assert (a is None) == (b is None)
                if a is not None:
                    assert torch.equal(a, b)
    finally:
        D.device_row_count = old


@pytest.mark.gpu
def test_the_eager_loop_learns_the_share_of_rows_without_a_host_read():
    """device_row_count = "auto" (the default): the first forward saves and its backward runs dense and only counts; once a
    count has reached pinned memory and says few rows carry a gradient the forward keeps nothing and the backward recomputes
    those rows; when the gradients turn dense again the next forward (after the count arrived) saves again.  Gradients
    equal the blocking selection's in every phase (dense phase: the dense backward's, bit for bit)."""
    from gftorf_amd import deform as D
    dev = torch.device("cuda:0")
    n = 20_000
    x, t = _inputs(n, 8, shared_t=True)
    xt, tt = torch.tensor(x, device=dev), torch.tensor(t, device=dev)
    g = torch.Generator().manual_seed(3)
    g_dxyz = torch.randn((n, 3), generator=g).to(dev)
    g_dsh = torch.randn((n, 16, 3), generator=g).to(dev)
    few = (torch.rand((n,), generator=g) < 0.1).to(dev)
    sparse = (g_dxyz * few[:, None], g_dsh * few[:, None, None])
    dense = (g_dxyz, g_dsh)

    def step(net, up):
        net.zero_grad(set_to_none=True)
        d_xyz, _, d_sh, _ = net(xt, tt)
        torch.autograd.backward([d_xyz, d_sh], list(up))
        torch.cuda.synchronize()                              # (so that the count HAS arrived when the next forward looks)
        return {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}, D.backward_stats()

    old = D.device_row_count
    try:
        D.device_row_count = False
        ref_net, _ = _net(12, dev)
        D.sparse_backward = False
        ref_sparse_dense, _ = step(ref_net, sparse)           # dense backward on the sparse gradients
        ref_dense, _ = step(ref_net, dense)
        D.sparse_backward = True
        D.device_row_count = True
        ref_rows, _ = step(ref_net, sparse)                   # rows counted on the device (equal to the blocking selection: test above)
        D.device_row_count = "auto"
        net, _ = _net(12, dev)
        a, st = step(net, sparse)                             # share unknown: saves, dense, counts
        assert not st["recomputed"] and st["points_processed"] == n
        assert net._save_state["pending"] is not None         # (the count is on its way; the next forward reads it)
        b, st = step(net, sparse)                             # few rows: keeps nothing, recomputes them
        assert st["recomputed"] and st["points_processed"] == int(few.sum())
        c, st = step(net, dense)                              # lazy forward, dense gradients: all rows recomputed
        assert st["recomputed"] and st["points_processed"] == n
        d, st = step(net, dense)                              # the count said "all": saves again
        assert not st["recomputed"] and st["points_processed"] == n
        for k in ref_dense:
            assert torch.equal(a[k], ref_sparse_dense[k]), k
            assert torch.equal(b[k], ref_rows[k]), k
            assert torch.equal(d[k], ref_dense[k]), k
            assert float((c[k] - ref_dense[k]).abs().max()) <= 2e-5 * float(ref_dense[k].abs().max()) + 1e-30, k
    finally:
        D.device_row_count, D.sparse_backward = old, True


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["xyz_only", "sh_only", "one_row", "all_but_one"])
def test_rows_counted_on_the_device_edge_cases(which):
    """One of the two outputs without an upstream gradient (NULL in the C call), a single row, all rows but one: the rows
    counted on the device against the blocking selection, bit for bit."""
    from gftorf_amd import deform as D
    dev = torch.device("cuda:0")
    n = 8_195                                                     # (not a multiple of any tile)
    x, t = _inputs(n, 4, shared_t=False)
    xt, tt = torch.tensor(x, device=dev), torch.tensor(t, device=dev)
    g = torch.Generator().manual_seed(17)
    keep = torch.rand((n,), generator=g) < 0.3
    if which == "one_row":
        keep = torch.zeros((n,), dtype=torch.bool)
        keep[n - 2] = True
    elif which == "all_but_one":
        keep = torch.ones((n,), dtype=torch.bool)
        keep[77] = False
    keep = keep.to(dev)
    g_dxyz = torch.randn((n, 3), generator=g).to(dev) * keep[:, None]
    g_dsh = torch.randn((n, 16, 3), generator=g).to(dev) * keep[:, None, None]

    def step(net):
        net.zero_grad(set_to_none=True)
        d_xyz, _, d_sh, _ = net(xt, tt)
        if which == "xyz_only":
            torch.autograd.backward([d_xyz], [g_dxyz])
        elif which == "sh_only":
            torch.autograd.backward([d_sh], [g_dsh])
        else:
            torch.autograd.backward([d_xyz, d_sh], [g_dxyz, g_dsh])
        return {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}, D.backward_stats()

    old = (D.device_row_count, D.lazy_save, D._SPARSE_MAX_FRACTION)
    try:
        D.device_row_count, D.lazy_save, D._SPARSE_MAX_FRACTION = False, True, 2.0
        ref_net, _ = _net(12, dev)
        ref_net._save_state = {"fraction": 0.0}
        ref, st_ref = step(ref_net)
        D.device_row_count = True
        net, _ = _net(12, dev)
        got, st = step(net)
        assert st["points_processed"] == st_ref["points_processed"] == int(keep.sum())
        assert set(got) == set(ref)
        for k in ref:
            assert torch.equal(got[k], ref[k]), k
    finally:
        D.device_row_count, D.lazy_save, D._SPARSE_MAX_FRACTION = old
