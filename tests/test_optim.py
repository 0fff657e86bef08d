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
