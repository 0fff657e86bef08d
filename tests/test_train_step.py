"""One iteration of the reference's training-loop shape through every HIP component at once
(train.py:164-177 + gaussian_renderer/__init__.py:81-117): deformation network query -> fused input
assembly -> rasterizer -> loss -> backward into the network's weights and the Gaussian parameters.

Each component has its own parity test; this one checks the autograd plumbing BETWEEN the custom ops
(gradient routing, None gradients, accumulation into shared leaves) against the composition of their
oracles with the chain rule written out: C oracle of the rasterizer -> eager assembly (CPU autograd) ->
numpy adjoint of the network."""
import numpy as np
import pytest
import torch

from oracle import assemble_ref, deform_ref, oracle
from tests import helpers


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.gpu
def test_training_iteration_chain_matches_composed_oracles():
    from gftorf_amd import reference_network, GaussianRasterizer, assemble_inputs
    dev = torch.device("cuda:0")
    scene = helpers.small_scene(P=700, W=96, H=64, seed=21)
    g = scene["gaussians"]
    P = g["means3D"].shape[0]
    rng = np.random.default_rng(8)
    mask = rng.random(P) < 0.35
    lo, hi = g["means3D"].min(0), g["means3D"].max(0)
    x_in = ((g["means3D"] - lo) / (hi - lo)).astype(np.float32)[mask]          # get_xyz_normalized, detached
    t_in = np.full((x_in.shape[0], 1), 0.3, np.float32)
    params = deform_ref.random_params(31, head_std=0.01)                        # small offsets
    rot_raw = (g["rotations"] * rng.uniform(0.5, 2.0, (P, 1))).astype(np.float32)
    opac = g["opacities"].reshape(P, 1)

    # ---- HIP chain -------------------------------------------------------------------------------------
    net = reference_network()
    net.load_state_dict({k: torch.tensor(v) for k, v in params.items()})
    net = net.to(dev)
    leaf = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in
            dict(xyz=g["means3D"], opacity=opac, scaling=g["scales"], rotation=g["rotations"], rotation_raw=rot_raw,
                 fc=g["shs"], fp=g["shs_p"]).items()}
    ssp = torch.zeros((P, 3), device=dev, requires_grad=True)
    t_dev = torch.tensor(t_in[:1], device=dev).expand(x_in.shape[0], -1)
    d_xyz, d_rot, d_sh, d_sh_p = net(torch.tensor(x_in, device=dev), t_dev)
    m3, m2, op, sc, rot, shs, shs_p = assemble_inputs(leaf["xyz"], ssp, leaf["opacity"], leaf["scaling"], leaf["rotation"],
                                                      leaf["rotation_raw"], leaf["fc"], leaf["fp"],
                                                      torch.tensor(mask, device=dev), d_xyz, d_rot, d_sh, d_sh_p)
    rast = GaussianRasterizer(raster_settings=helpers.gpu_settings(scene, dev))
    outs = dict(zip(helpers.OUT_NAMES, rast(means3D=m3, means2D=m2, opacities=op, shs=shs, shs_p=shs_p, scales=sc,
                                            rotations=rot, phase_offset=scene["phase_offset"], dc_offset=scene["dc_offset"])))
    sum((outs[k] * torch.tensor(scene["grads"][k], device=dev)).sum() for k in helpers.GRAD_KEYS).backward()

    # ---- composed oracles ------------------------------------------------------------------------------
    o_dxyz, _, o_dsh, _ = deform_ref.forward(params, x_in, t_in)
    cl = {k: torch.tensor(v.detach().cpu().numpy(), requires_grad=True) for k, v in leaf.items()}
    c_ssp = torch.zeros((P, 3), requires_grad=True)
    c_dxyz, c_dsh = torch.tensor(o_dxyz, requires_grad=True), torch.tensor(o_dsh, requires_grad=True)
    n_dyn = x_in.shape[0]
    a = assemble_ref.assemble_eager(cl["xyz"], c_ssp, cl["opacity"], cl["scaling"], cl["rotation"], cl["rotation_raw"],
                                    cl["fc"], cl["fp"], torch.tensor(mask), c_dxyz, torch.zeros((n_dyn, 4)), c_dsh,
                                    torch.zeros((n_dyn, 16, 2)))
    inputs = dict(means3D=a[0].detach().numpy(), opacities=a[2].detach().numpy(), scales=a[3].detach().numpy(),
                  rotations=a[4].detach().numpy(), shs=a[5].detach().numpy(), shs_p=a[6].detach().numpy())
    f, b = helpers.run_oracle(oracle, scene, inputs=inputs)
    up = [b["dL_dmeans3D"], b["dL_dmeans2D"], b["dL_dopacity"], b["dL_dscales"], b["dL_drotations"], b["dL_dsh"], b["dL_dsh_p"]]
    torch.autograd.backward(list(a), [torch.tensor(np.asarray(u, np.float32).reshape(tuple(o.shape))) for u, o in zip(up, a)])
    o_net = deform_ref.backward(params, x_in, t_in, c_dxyz.grad.numpy(), c_dsh.grad.numpy(), dtype=np.float64)

    # the rendered images agree first (same tolerance as the rasterizer's own parity tests)
    assert np.abs(outs["color"].detach().cpu().numpy() - f["color"]).mean() < 1e-5
    # Gaussian parameters: the rasterizer's gradients routed back through the assembly
    for k in ("xyz", "opacity", "scaling", "rotation_raw", "rotation", "fc", "fp"):
        assert _rel(leaf[k].grad.cpu().numpy(), cl[k].grad.numpy()) < 1e-3, k
    assert _rel(ssp.grad.cpu().numpy(), c_ssp.grad.numpy()) < 1e-3
    # network weights: rasterizer -> assembly -> network
    for name, p in net.named_parameters():
        if o_net[name] is None:
            assert p.grad is None, name
        else:
            assert _rel(p.grad.cpu().numpy(), o_net[name]) < 2e-3, name


@pytest.mark.gpu
def test_optimisation_loop_reduces_the_loss():
    """The reference's loop shape for 60 iterations on a small scene: network query, assembly, colour + ToF
    render, L1 losses against renders of the unperturbed scene, backward, FusedAdam on the Gaussians, Adam on
    the network, densification statistics.  The loss must fall: gradients are useful, not only equal."""
    from gftorf_amd import reference_network, FusedAdam, GaussianRasterizer, assemble_inputs, densify
    dev = torch.device("cuda:0")
    scene = helpers.small_scene(P=1500, W=96, H=64, seed=33)
    g = scene["gaussians"]
    P = g["means3D"].shape[0]
    rng = np.random.default_rng(3)
    mask = torch.tensor(rng.random(P) < 0.3, device=dev)
    t32 = lambda a: torch.tensor(np.asarray(a, np.float32), device=dev)
    rast = GaussianRasterizer(raster_settings=helpers.gpu_settings(scene, dev))

    def render(xyz, opac, scal, rot, fc, fp):
        ssp = torch.zeros((P, 3), device=dev, requires_grad=True)
        out = rast(means3D=xyz, means2D=ssp, opacities=opac, shs=fc, shs_p=fp, scales=scal, rotations=rot,
                   phase_offset=scene["phase_offset"], dc_offset=scene["dc_offset"])
        return dict(zip(helpers.OUT_NAMES, out)), ssp

    with torch.no_grad():
        target, _ = render(t32(g["means3D"]), t32(g["opacities"]).reshape(P, 1), t32(g["scales"]), t32(g["rotations"]),
                           t32(g["shs"]), t32(g["shs_p"]))
    # start from a perturbed copy; the dynamic third is additionally displaced through the network
    leaf = dict(xyz=t32(g["means3D"] + rng.normal(0, 0.01, (P, 3))), opacity=t32(g["opacities"]).reshape(P, 1) * 0.8,
                scaling=t32(g["scales"] * 1.15), rotation_raw=t32(g["rotations"]), fc=t32(g["shs"] * 0.8), fp=t32(g["shs_p"] * 0.9))
    for v in leaf.values():
        v.requires_grad_(True)
    net = reference_network().to(dev)
    torch.manual_seed(0)
    for n_, p_ in net.named_parameters():
        torch.nn.init.normal_(p_, 0.0, 0.03 if n_.startswith("linear") and n_.endswith("weight") else 1e-3)
    opt = FusedAdam([{"params": [v], "lr": lr, "name": k} for (k, v), lr in
                     zip(leaf.items(), (2e-4, 1e-2, 2e-3, 1e-3, 2e-3, 2e-3))], lr=0.0, eps=1e-15)
    opt_net = torch.optim.Adam(net.parameters(), lr=1e-4, eps=1e-15)
    x_n = ((leaf["xyz"].detach() - leaf["xyz"].detach().min(0).values) /
           (leaf["xyz"].detach().max(0).values - leaf["xyz"].detach().min(0).values))[mask]
    acc_s, den_s, maxr = torch.zeros((P, 1), device=dev), torch.zeros((P, 1), device=dev), torch.zeros(P, device=dev)
    losses = []
    for it in range(60):
        d_xyz, d_rot, d_sh, d_sh_p = net(x_n, torch.full((1, 1), 0.5, device=dev).expand(x_n.size(0), -1))
        ssp0 = torch.zeros((P, 3), device=dev)
        m3, _, op, sc, rot, shs, shs_p = assemble_inputs(leaf["xyz"], ssp0, leaf["opacity"], leaf["scaling"],
                                                         torch.nn.functional.normalize(leaf["rotation_raw"]), leaf["rotation_raw"],
                                                         leaf["fc"], leaf["fp"], mask, d_xyz, d_rot, d_sh, d_sh_p)
        out, ssp = render(m3, op, sc, rot, shs, shs_p)
        loss = (out["color"] - target["color"]).abs().mean() + (out["phasor"] - target["phasor"]).abs().mean() * 0.5
        loss.backward()
        densify.add_densification_stats(acc_s, den_s, maxr, ssp.grad, out["radii"] > 0, out["pixels"], out["radii"])
        opt.step(); opt_net.step()
        opt.zero_grad(set_to_none=True); opt_net.zero_grad(set_to_none=True)
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses))
    assert np.mean(losses[-5:]) < 0.6 * np.mean(losses[:5]), (losses[:5], losses[-5:])
    assert den_s.sum() > 0 and maxr.max() > 0
