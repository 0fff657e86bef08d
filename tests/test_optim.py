"""FusedAdam (optimizer part) against torch.optim.Adam, the optimizer the
reference uses (scene/gaussian_model.py:274)."""
import numpy as np
import pytest
import torch


def groups(dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    mk = lambda *s: torch.nn.Parameter(torch.randn(*s, generator=g).to(dev))
    # shapes of the reference's parameter groups (xyz, f_dc, f_rest, opacity, scaling, rotation, an odd one)
    ps = [mk(1001, 3), mk(1001, 1, 3), mk(1001, 15, 3), mk(1001, 1), mk(1001, 3), mk(1001, 4), mk(7)]
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 0.05, 5e-3, 1e-3, 0.0]
    return [{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(ps, lrs))]


def run(opt_cls, dev, steps=5, wd=0.0, **kw):
    gs = groups(dev)
    opt = opt_cls(gs, lr=0.0, eps=1e-15, weight_decay=wd, **kw)
    gen = torch.Generator().manual_seed(123)
    for it in range(steps):
        for grp in opt.param_groups:
            p = grp["params"][0]
            if it == 2 and grp["name"] == "3":
                p.grad = None                       # a parameter without gradient is skipped
            else:
                p.grad = torch.randn(p.shape, generator=gen).to(dev) * (10.0 ** (it - 2))
        opt.step()
    return opt


def test_reference_optimizer_is_deterministic_on_cpu():
    a, b = run(torch.optim.Adam, "cpu"), run(torch.optim.Adam, "cpu")
    for ga, gb in zip(a.param_groups, b.param_groups):
        assert torch.equal(ga["params"][0], gb["params"][0])


def test_fused_adam_rejects_cpu_and_unsupported_modes():
    from gftorf_amd import FusedAdam
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError, match="HIP device only"):
        FusedAdam([p]).step()
    with pytest.raises(NotImplementedError):
        FusedAdam([p], amsgrad=True)


@pytest.mark.gpu
@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_fused_adam_vs_torch_adam(wd, gpu):
    from gftorf_amd import FusedAdam
    ref = run(torch.optim.Adam, "cpu", wd=wd)
    got = run(FusedAdam, gpu, wd=wd)
    for gr, gg in zip(ref.param_groups, got.param_groups):
        pr, pg = gr["params"][0], gg["params"][0]
        sr, sg = ref.state[pr], got.state[pg]
        assert float(sr["step"]) == float(sg["step"])
        for name, a, b in (("param", pr, pg), ("exp_avg", sr["exp_avg"], sg["exp_avg"]),
                           ("exp_avg_sq", sr["exp_avg_sq"], sg["exp_avg_sq"])):
            ref_np = a.detach().numpy()
            # one rounding per operation may differ (fused multiply-adds on either side); parameters
            # near zero are the difference of larger updates, hence the absolute term
            np.testing.assert_allclose(b.detach().cpu().numpy(), ref_np, rtol=3e-6, atol=3e-7 * float(np.abs(ref_np).max()),
                                       err_msg="%s of group %s" % (name, gr["name"]))


@pytest.mark.gpu
def test_fused_adam_state_is_torch_compatible(gpu):
    """The reference's densification edits optimizer.state in place and reloads state_dicts."""
    from gftorf_amd import FusedAdam
    opt = run(FusedAdam, gpu, steps=2)
    sd = opt.state_dict()
    other = torch.optim.Adam(groups(gpu), lr=0.0, eps=1e-15)
    other.load_state_dict(sd)
    assert set(other.state[other.param_groups[0]["params"][0]].keys()) == {"step", "exp_avg", "exp_avg_sq"}


@pytest.mark.gpu
@pytest.mark.parametrize("frac", [0.0, 0.14, 1.0])
def test_step_on_visible_rows_only(frac, gpu):
    """Opt-in `step(visibility=mask)` (sparse Adam on visible Gaussians): rows of the mask take exactly
    the dense step (bit for bit), the other rows keep parameter and moments; parameters that are not per-Gaussian (the
    odd 7-element one) take the dense step."""
    from gftorf_amd import FusedAdam
    gen = torch.Generator().manual_seed(99)
    vis = (torch.rand(1001, generator=gen) < frac).to(gpu)
    dense, rows = FusedAdam(groups(gpu), lr=0.0, eps=1e-15), FusedAdam(groups(gpu), lr=0.0, eps=1e-15)
    ggen = torch.Generator().manual_seed(5)
    for it in range(4):
        for gd, gr in zip(dense.param_groups, rows.param_groups):
            pd, pr = gd["params"][0], gr["params"][0]
            pd.grad = torch.randn(pd.shape, generator=ggen).to(gpu)
            pr.grad = pd.grad.clone()
            # the dense optimizer starts every step from the row-wise one's state: one step is compared at a time
            pd.data.copy_(pr.data)
            if pr in rows.state:
                if pd not in dense.state:
                    dense.state[pd] = {"step": rows.state[pr]["step"].clone(), "exp_avg": rows.state[pr]["exp_avg"].clone(),
                                       "exp_avg_sq": rows.state[pr]["exp_avg_sq"].clone()}
                for k in ("step", "exp_avg", "exp_avg_sq"):
                    dense.state[pd][k].copy_(rows.state[pr][k])
        old = [(g["params"][0].detach().clone(),
                rows.state[g["params"][0]]["exp_avg"].clone() if g["params"][0] in rows.state else None,
                rows.state[g["params"][0]]["exp_avg_sq"].clone() if g["params"][0] in rows.state else None) for g in rows.param_groups]
        dense.step()
        rows.step(visibility=vis if it % 2 == 0 else vis.to(torch.uint8))
        for (p0, m0, v0), gd, gr in zip(old, dense.param_groups, rows.param_groups):
            pd, pr = gd["params"][0], gr["params"][0]
            sd, sr = dense.state[pd], rows.state[pr]
            assert float(sd["step"]) == float(sr["step"]) == it + 1
            m0 = torch.zeros_like(p0) if m0 is None else m0
            v0 = torch.zeros_like(p0) if v0 is None else v0
            if pr.shape[0] == 1001:
                sel = vis.view(-1, *([1] * (pr.dim() - 1)))
                want = (torch.where(sel, pd.detach(), p0), torch.where(sel, sd["exp_avg"], m0), torch.where(sel, sd["exp_avg_sq"], v0))
            else:
                want = (pd.detach(), sd["exp_avg"], sd["exp_avg_sq"])
            for name, w, g in zip(("param", "exp_avg", "exp_avg_sq"), want, (pr.detach(), sr["exp_avg"], sr["exp_avg_sq"])):
                assert torch.equal(w, g), "%s of group %s, step %d" % (name, gr["name"], it)
    with pytest.raises(RuntimeError, match="visibility"):
        rows.step(visibility=torch.zeros(1001, device=gpu))


@pytest.mark.gpu
def test_gradient_view_at_an_odd_offset_and_row_params(gpu):
    """The rasterizer returns the dc_offset gradient as element 1 of a two-float tensor (api.py: `g["offsets"][1:2]`): a
    contiguous view 4 bytes into its allocation, which autograd may adopt as `.grad`.  The step copies such a gradient
    instead of rejecting the whole table, and step counters only advance with a step that was taken.  `row_params` names
    the tensors a visibility mask applies to."""
    from gftorf_amd import FusedAdam
    p = torch.nn.Parameter(torch.tensor([0.5], device=gpu))
    q = torch.nn.Parameter(torch.tensor([0.5], device=gpu))
    two = torch.tensor([9.0, 0.25], device=gpu)
    p.grad = two[1:2]
    q.grad = torch.tensor([0.25], device=gpu)
    assert p.grad.data_ptr() % 16 == 4
    a, b = FusedAdam([p], lr=1e-2), FusedAdam([q], lr=1e-2)
    for _ in range(3):
        a.step()
        b.step()
    assert torch.equal(p.detach(), q.detach()) and float(a.state[p]["step"]) == 3.0
    # a 7-row weight is not a per-Gaussian tensor although a 7-entry mask fits it: named row parameters only
    w = torch.nn.Parameter(torch.ones(7, 3, device=gpu))
    x = torch.nn.Parameter(torch.ones(7, 3, device=gpu))
    w.grad, x.grad = torch.ones_like(w), torch.ones_like(x)
    opt = FusedAdam([w, x], lr=1e-2)
    vis = torch.tensor([1, 0, 0, 0, 0, 0, 1], dtype=torch.bool, device=gpu)
    opt.step(visibility=vis, row_params=[x])
    assert (w.detach() != 1.0).all()                                  # dense step
    assert (x.detach()[1:6] == 1.0).all() and (x.detach()[0] != 1.0).all()


@pytest.mark.gpu
def test_capturable_adam_equals_the_eager_one_and_replays(gpu):
    """FusedAdam(capturable=True): step counts and learning rates on the device.  Eager steps equal the non-capturable ones
    bit for bit (the bias corrections are the same doubles rounded once); a step captured in a graph and replayed follows
    the learning rates the scheduler writes between replays (refresh_lr) and new gradient VALUES in the static tensors."""
    from gftorf_amd import FusedAdam
    ref = run(FusedAdam, gpu, steps=5)
    got = run(FusedAdam, gpu, steps=5, capturable=True)
    for gr, gg in zip(ref.param_groups, got.param_groups):
        pr, pg = gr["params"][0], gg["params"][0]
        sr, sg = ref.state[pr], got.state[pg]
        assert sg["step"].is_cuda and float(sr["step"]) == float(sg["step"])
        for name, a, b in (("param", pr, pg), ("exp_avg", sr["exp_avg"], sg["exp_avg"]), ("exp_avg_sq", sr["exp_avg_sq"], sg["exp_avg_sq"])):
            assert torch.equal(a, b), "%s of group %s" % (name, gr["name"])
    # ---- under a graph: static gradient tensors, learning rates that move between replays
    gen = torch.Generator().manual_seed(7)
    sched = lambda it, base: base * (0.9 ** it)

    def make(capturable):
        gs = groups(gpu, seed=3)
        opt = FusedAdam(gs, lr=0.0, eps=1e-15, capturable=capturable)
        for g in opt.param_groups:
            g["base_lr"] = g["lr"]
            g["params"][0].grad = torch.zeros_like(g["params"][0])
        return opt
    eager, cap = make(False), make(True)
    grads = [[torch.randn(g["params"][0].shape, generator=gen).to(gpu) for g in eager.param_groups] for _ in range(6)]

    def load(opt, it):
        for g, gr in zip(opt.param_groups, grads[it]):
            g["params"][0].grad.copy_(gr)
            g["lr"] = sched(it, g["base_lr"])
    load(eager, 0), load(cap, 0)
    eager.step(), cap.step()                          # the state is created outside the graph
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        load(cap, 1)
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            cap.step()
    torch.cuda.current_stream().wait_stream(side)
    # (the capture itself ran nothing: iteration 1 is the first replay)
    for it in range(1, 6):
        load(eager, it)
        eager.step()
        load(cap, it)
        cap.refresh_lr()
        graph.replay()
    torch.cuda.synchronize()
    for ge, gc in zip(eager.param_groups, cap.param_groups):
        pe, pc = ge["params"][0], gc["params"][0]
        assert float(eager.state[pe]["step"]) == float(cap.state[pc]["step"]) == 6.0
        assert torch.equal(pe, pc), ge["name"]
        assert torch.equal(eager.state[pe]["exp_avg_sq"], cap.state[pc]["exp_avg_sq"])
