"""Per-Gaussian bookkeeping: statistics update and row compaction.
CPU: the eager restatement against a per-Gaussian loop.  GPU: the HIP path (C ABI of
include/gftorf_densify.h) against the eager statements on CPU tensors, bit for bit."""
import numpy as np
import pytest
import torch

from oracle import densify_ref

# the reference's optimizer groups (scene/gaussian_model.py:247-272): name -> row shape
GROUPS = {"xyz": (3,), "f_dc_color": (1, 3), "f_rest_color": (15, 3), "phase_f_dc": (1, 1), "phase_f_rest": (15, 1),
          "amp_f_dc": (1, 1), "amp_f_rest": (15, 1), "opacity": (1,), "scaling": (3,), "rotation": (4,), "f_seg_color": (3,)}


def _stats_inputs(P, seed, subset_apply=True):
    rng = np.random.default_rng(seed)
    d = dict(accum=rng.random((P, 1)).astype(np.float32), denom=rng.random((P, 1)).astype(np.float32) * 10,
             maxr=rng.integers(0, 30, P).astype(np.float32), grad=rng.normal(0, 1e-3, (P, 3)).astype(np.float32),
             upd=rng.random(P) < 0.6, pixels=rng.integers(0, 400, (P, 1)).astype(np.float32),
             radii=rng.integers(0, 60, P).astype(np.int32))
    d["radii"][~d["upd"]] = 0
    d["apply"] = np.logical_or(d["upd"], rng.random(P) < 0.3) if subset_apply else None
    return d


def _run_eager(d, with_apply, dev="cpu"):
    t = {k: (torch.tensor(v, device=dev) if v is not None else None) for k, v in d.items()}
    densify_ref.add_densification_stats_eager(t["accum"], t["denom"], t["maxr"], t["grad"], t["upd"], t["pixels"], t["radii"],
                                              apply_mask=t["apply"] if with_apply else None)
    return t["accum"], t["denom"], t["maxr"]


@pytest.mark.parametrize("with_apply", [False, True])
def test_eager_statements_match_the_loop(with_apply):
    d = _stats_inputs(777, 1)
    a, dn, m = _run_eager(d, with_apply)
    la, ldn, lm = densify_ref.stats_loops(d["accum"], d["denom"], d["maxr"], d["grad"], d["upd"], d["pixels"], d["radii"],
                                          d["apply"] if with_apply else None)
    np.testing.assert_array_equal(a.numpy(), la)
    np.testing.assert_array_equal(dn.numpy(), ldn)
    np.testing.assert_array_equal(m.numpy(), lm)


def test_product_fails_loudly_without_a_device():
    from gftorf_amd import densify
    with pytest.raises(RuntimeError, match="HIP device only"):
        densify.select_rows(torch.ones(4, dtype=torch.bool), torch.zeros(4, 3))


@pytest.mark.gpu
@pytest.mark.parametrize("P,with_apply", [(1, False), (1000, False), (4097, True), (100003, False), (100003, True)])
def test_stats_bit_exact(P, with_apply):
    from gftorf_amd import densify
    dev = torch.device("cuda:0")
    d = _stats_inputs(P, 10 + P)
    ref = _run_eager(d, with_apply)
    t = {k: (torch.tensor(v, device=dev) if v is not None else None) for k, v in d.items()}
    densify.add_densification_stats(t["accum"], t["denom"], t["maxr"], t["grad"], t["upd"], t["pixels"], t["radii"],
                                    apply_mask=t["apply"] if with_apply else None)
    for got, want in zip((t["accum"], t["denom"], t["maxr"]), ref):
        assert torch.equal(got.cpu(), want)


@pytest.mark.gpu
def test_stats_from_a_real_render():
    """The statistics of one rasterizer call, as train.py:441-449 takes them."""
    import helpers
    from gftorf_amd import densify
    dev = torch.device("cuda:0")
    scene = helpers.small_scene(P=3000, W=128, H=96, seed=9)
    out, grads, tens = helpers.run_gpu(scene, dev)
    P = 3000
    vis = tens["outs"]["radii"] > 0
    acc = torch.zeros((P, 1), device=dev); den = torch.zeros((P, 1), device=dev); mr = torch.zeros((P,), device=dev)
    densify.add_densification_stats(acc, den, mr, tens["means2D"].grad, vis, tens["outs"]["pixels"], tens["outs"]["radii"])
    c = lambda x: x.detach().cpu()
    acc_c, den_c, mr_c = torch.zeros((P, 1)), torch.zeros((P, 1)), torch.zeros((P,))
    densify_ref.add_densification_stats_eager(acc_c, den_c, mr_c, c(tens["means2D"].grad), c(vis), c(tens["outs"]["pixels"]),
                                              c(tens["outs"]["radii"]))
    assert torch.equal(acc.cpu(), acc_c) and torch.equal(den.cpu(), den_c) and torch.equal(mr.cpu(), mr_c)
    assert acc.abs().sum() > 0 and int(vis.sum()) > 100


@pytest.mark.gpu
@pytest.mark.parametrize("P,frac", [(0, 0.5), (1, 1.0), (15, 0.5), (4096, 0.0), (4096, 1.0), (70001, 0.37), (1_000_003, 0.9)])
def test_select_rows_bit_exact(P, frac):
    from gftorf_amd import densify
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(P + 1)
    mask = torch.rand(P, generator=g) < frac
    ts = [torch.randn((P,) + s, generator=g) for s in ((3,), (15, 3), (1, 1), (4,), (16, 2), ())]
    ts.append(torch.randint(0, 1000, (P,), generator=g, dtype=torch.int32))
    ts.append(torch.rand(P, generator=g) < 0.5)                          # 1-byte rows: fallback path
    got = densify.select_rows(mask.to(dev), *[t.to(dev) for t in ts])
    for t, gt in zip(ts, got):
        want = t[mask]
        assert gt.shape == want.shape and gt.dtype == want.dtype
        assert torch.equal(gt.cpu(), want)
    sel = densify.RowSelection(mask.to(dev))
    assert sel.count == int(mask.sum())
    if P:
        # rank = exclusive prefix count of the mask
        assert torch.equal(sel.rank.cpu().long(), torch.cumsum(mask.long(), 0) - mask.long())


def _optimizer(P, dev, seed):
    g = torch.Generator().manual_seed(seed)
    groups = [{"params": [torch.nn.Parameter(torch.randn((P,) + s, generator=g).to(dev))], "lr": 1e-3, "name": n}
              for n, s in GROUPS.items()]
    groups.append({"params": [torch.nn.Parameter(torch.zeros(1, device=dev))], "lr": 1e-3, "name": "phase_offset"})
    opt = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
    for grp in opt.param_groups:                                        # one step so that the moments exist
        p = grp["params"][0]
        p.grad = torch.randn(p.shape, generator=g).to(dev)
    opt.step()
    return opt


@pytest.mark.gpu
def test_prune_and_cat_optimizer_like_the_reference():
    from gftorf_amd import densify
    dev = torch.device("cuda:0")
    P = 50_001
    b = _optimizer(P, "cpu", 3)
    # the same parameters and moments on the device (a device Adam step is not bit-equal to a host one)
    a = torch.optim.Adam([{"params": [torch.nn.Parameter(g_["params"][0].detach().to(dev))], "lr": g_["lr"], "name": g_["name"]}
                          for g_ in b.param_groups], lr=0.0, eps=1e-15)
    a.load_state_dict(b.state_dict())
    g = torch.Generator().manual_seed(4)
    keep = torch.rand(P, generator=g) < 0.8
    new_a = densify.prune_optimizer(a, keep.to(dev))          # {group name: Parameter}, like the reference
    new_b = densify_ref.prune_optimizer_eager(b, keep)
    assert list(new_a) == list(new_b) == list(GROUPS)
    for ga, gb in zip(a.param_groups, b.param_groups):
        pa, pb = ga["params"][0], gb["params"][0]
        assert torch.equal(pa.detach().cpu(), pb.detach()) and pa.requires_grad
        if ga["name"] in GROUPS:
            assert pa is new_a[ga["name"]] and set(a.state[pa]) == set(b.state[pb])
            for k in ("exp_avg", "exp_avg_sq"):
                assert torch.equal(a.state[pa][k].cpu(), b.state[pb][k])
    assert len(a.state) == len(b.state)
    # clone: the selected rows appended (scene/gaussian_model.py:600-621)
    n = new_a["xyz"].size(0)
    pick = torch.rand(n, generator=g) < 0.1
    ext_b = {k: b.param_groups[i]["params"][0].detach()[pick] for i, k in enumerate(GROUPS)}
    sel2 = densify.RowSelection(pick.to(dev))
    ext_a = {k: sel2.take(a.param_groups[i]["params"][0]) for i, k in enumerate(GROUPS)}
    ca, cb = densify.cat_tensors_to_optimizer(a, ext_a), densify_ref.cat_tensors_to_optimizer_eager(b, ext_b)
    for k in GROUPS:
        assert torch.equal(ca[k].detach().cpu(), cb[k].detach())
        assert ca[k].size(0) == n + int(pick.sum())
    for ga, gb in zip(a.param_groups, b.param_groups):
        if ga["name"] in GROUPS:
            for k in ("exp_avg", "exp_avg_sq"):
                assert torch.equal(a.state[ga["params"][0]][k].cpu(), b.state[gb["params"][0]][k])
    for grp in a.param_groups:                                          # the optimizer is still consistent
        grp["params"][0].grad = torch.zeros_like(grp["params"][0])
    a.step()


@pytest.mark.gpu
@pytest.mark.parametrize("P,max_screen_size", [(20_000, 20), (3_001, None)])
def test_densify_composites_like_the_reference(P, max_screen_size):
    """densify_and_clone / densify_and_split / prune_points / densify_and_prune (scene/gaussian_model.py:494-646): the
    product's functions on a model object against the reference's eager statements on a twin, same device, same torch
    generator state (the split draws torch.normal samples): every parameter, Adam moment and statistic bit-identical."""
    from gftorf_amd import FusedAdam, densify
    dev = torch.device("cuda:0")
    a = densify_ref.EagerGaussians(P, dev, seed=5)             # driven by gftorf_amd.densify
    b = densify_ref.EagerGaussians(P, dev, seed=5)             # driven by its own (the reference's) statements
    for k, v in a.snapshot().items():
        assert torch.equal(v, b.snapshot()[k]), k
    extent = 2.0
    for rnd in range(2):
        torch.manual_seed(100 + rnd)
        densify.densify_and_prune(a, 0.0002, 0.005, extent, max_screen_size)
        torch.manual_seed(100 + rnd)
        b.densify_and_prune(0.0002, 0.005, extent, max_screen_size)
        sa, sb = a.snapshot(), b.snapshot()
        assert sa["_xyz"].shape[0] == sb["_xyz"].shape[0] != P
        assert set(sa) == set(sb)
        for k in sa:
            assert torch.equal(sa[k], sb[k]), (rnd, k)
        assert [g["name"] for g in a.optimizer.param_groups] == [g["name"] for g in b.optimizer.param_groups]
        assert a.optimizer.param_groups[0]["params"][0] is a._xyz
        # statistics for the next round (densification_postfix zeroed them)
        g = torch.Generator().manual_seed(7 + rnd)
        n = a._xyz.shape[0]
        acc, den = (torch.rand((n, 1), generator=g) * 0.03).to(dev), torch.randint(0, 60, (n, 1), generator=g).float().to(dev)
        rad = (torch.rand(n, generator=g) * 30).to(dev)
        for m in (a, b):
            m.xyz_gradient_accum, m.denom, m.max_radii2D = acc.clone(), den.clone(), rad.clone()
    # plain pruning (gaussian_model.py:642-646) and one more optimizer step on the result
    densify.prune(a, 0.3)
    b.prune_points((b.get_opacity < 0.3).squeeze())
    sa, sb = a.snapshot(), b.snapshot()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    for grp in a.optimizer.param_groups:
        grp["params"][0].grad = torch.zeros_like(grp["params"][0])
    a.optimizer.step()


@pytest.mark.gpu
@pytest.mark.parametrize("P", [0, 1, 63, 1000, 40_003])
def test_rows_with_a_nonzero_value_match_the_eager_expression(P):
    """gft_rows_any_nonzero against `~(max(a.abs().amax(1), b.abs().amax(1)) == 0)`: zero rows, -0, NaN, either tensor
    absent, rows that are / are not a whole number of 16-byte pieces."""
    from gftorf_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(P + 1)
    a = torch.randn((P, 3), generator=g)
    b = torch.randn((P, 16, 3), generator=g)
    keep = torch.rand((P,), generator=g) < 0.1
    a[~keep] = 0
    b[~keep] = 0
    if P > 10:
        a[3] = 0; b[3] = 0; b[3, 7, 1] = float("nan")       # NaN counts
        a[4] = 0; b[4] = 0; a[4, 2] = -0.0                      # -0 does not
        a[5] = 0; b[5] = 0; b[5, 15, 2] = 1e-38                 # the last value of a row
    for use_a, use_b in ((True, True), (True, False), (False, True)):
        ta, tb = a.to(dev), b.to(dev)
        m = None
        for t, use in ((ta, use_a), (tb, use_b)):
            if use:
                r = t.reshape(P, -1).abs().amax(dim=1) if P else t.new_zeros((0,))
                m = r if m is None else torch.maximum(m, r)
        want = ~(m == 0)
        mask = torch.full((P,), 7, device=dev, dtype=torch.uint8)
        with _lib.on_device(dev):
            _lib.check(lib.gft_rows_any_nonzero(_lib.raw_stream(dev), P, 3 if use_a else 0, ta.data_ptr() if (use_a and P) else None,
                                                48 if use_b else 0, tb.data_ptr() if (use_b and P) else None, mask.data_ptr() if P else None))
        assert torch.equal(mask.bool(), want) and (P == 0 or int(mask.max()) <= 1)
