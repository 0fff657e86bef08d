"""ORACLE (test infrastructure only) of the per-Gaussian bookkeeping -- SURVEY section 8(f) row 4.

The reference's statements on torch tensors (they are device-agnostic; the tests run them on CPU):

* :func:`add_densification_stats_eager` -- train.py:443 and GaussianModel.add_densification_stats
  (scene/gaussian_model.py:648-654), both branches.
* :func:`stats_loops` -- the same, one Gaussian at a time in numpy float32 (pins the eager statements).
* :func:`prune_optimizer_eager`, :func:`cat_tensors_to_optimizer_eager` -- GaussianModel._prune_optimizer /
  cat_tensors_to_optimizer (scene/gaussian_model.py:473-492, 516-537).

Only tests/, __graft_entry__.smoke() and bench.py's baseline leg may import this module.
"""
import numpy as np
import torch
from torch import nn

SKIP = ("phase_offset", "dc_offset")


def add_densification_stats_eager(xyz_gradient_accum, denom, max_radii2D, viewspace_grad, update_filter, pixels, radii,
                                  apply_mask=None):
    if max_radii2D is not None:
        max_radii2D[update_filter] = torch.max(max_radii2D[update_filter], radii[update_filter])          # train.py:443
    if apply_mask is None:                                                                                # :649-651
        xyz_gradient_accum[update_filter] += torch.norm(viewspace_grad[update_filter, :2], dim=-1, keepdim=True) * pixels[update_filter]
        denom[update_filter] += pixels[update_filter]
    else:                                                                                                 # :652-654
        both = torch.logical_and(apply_mask, update_filter)
        xyz_gradient_accum[both] += torch.norm(viewspace_grad[both, :2], dim=-1, keepdim=True) * pixels[update_filter]
        denom[both] += pixels[update_filter]


def stats_loops(accum, denom, maxr, grad, upd, pixels, radii, apply=None):
    f = np.float32
    accum, denom, maxr = accum.copy(), denom.copy(), maxr.copy()
    for i in range(accum.shape[0]):
        if not upd[i]:
            continue
        maxr[i] = max(maxr[i], f(radii[i]))
        if apply is not None and not apply[i]:
            continue
        # torch's norm reduction is `acc = fma(v, v, acc)`: the first square is rounded, the second fused
        x2 = f(grad[i, 0] * grad[i, 0])
        n = np.sqrt(f(np.float64(grad[i, 1]) * np.float64(grad[i, 1]) + np.float64(x2)))
        accum[i, 0] = f(accum[i, 0] + f(n * pixels[i, 0]))
        denom[i, 0] = f(denom[i, 0] + pixels[i, 0])
    return accum, denom, maxr


def prune_optimizer_eager(optimizer, mask):
    out = {}
    for group in optimizer.param_groups:
        if group["name"] in SKIP:
            continue
        p = group["params"][0]
        st = optimizer.state.get(p, None)
        new_p = nn.Parameter(p[mask].requires_grad_(True))
        if st is not None:
            st["exp_avg"] = st["exp_avg"][mask]
            st["exp_avg_sq"] = st["exp_avg_sq"][mask]
            del optimizer.state[p]
            optimizer.state[new_p] = st
        group["params"][0] = new_p
        out[group["name"]] = new_p
    return out


def cat_tensors_to_optimizer_eager(optimizer, tensors_dict):
    out = {}
    for group in optimizer.param_groups:
        if group["name"] in SKIP:
            continue
        ext = tensors_dict[group["name"]]
        p = group["params"][0]
        st = optimizer.state.get(p, None)
        new_p = nn.Parameter(torch.cat((p, ext), dim=0).requires_grad_(True))
        if st is not None:
            st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext)), dim=0)
            st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
            del optimizer.state[p]
            optimizer.state[new_p] = st
        group["params"][0] = new_p
        out[group["name"]] = new_p
    return out


# ---- the composites of scene/gaussian_model.py:494-646, restated on a minimal model object ------------------------
class EagerGaussians:
    """The attributes of the reference's ``GaussianModel`` that densification touches (scene/gaussian_model.py:55-153),
    holding the reference's eager statements as methods (``device="cuda"`` replaced by the tensors' own device).  Test
    infrastructure: the product's functions (gftorf_amd/densify.py) take any object with these attributes."""

    GROUPS = (("xyz", "_xyz", (3,)), ("f_dc_color", "_features_dc_color", (1, 3)), ("f_rest_color", "_features_rest_color", (15, 3)),
              ("phase_f_dc", "_features_dc_phase", (1, 1)), ("phase_f_rest", "_features_rest_phase", (15, 1)),
              ("amp_f_dc", "_features_dc_amp", (1, 1)), ("amp_f_rest", "_features_rest_amp", (15, 1)), ("opacity", "_opacity", (1,)),
              ("scaling", "_scaling", (3,)), ("rotation", "_rotation", (4,)), ("f_seg_color", "_features_seg_color", (3,)))

    def __init__(self, P, dev, seed, optimizer_cls=torch.optim.Adam):
        g = torch.Generator().manual_seed(seed)
        for name, attr, shape in self.GROUPS:
            t = torch.randn((P,) + shape, generator=g)
            if name == "scaling":
                t = torch.log(torch.exp(t * 0.7) * 0.02)          # world-space extents around 0.02
            if name == "opacity":
                t = t * 2.0                                       # sigmoid(.) spread over (0, 1)
            setattr(self, attr, nn.Parameter(t.to(dev).requires_grad_(True)))
        self._phase_offset = nn.Parameter(torch.zeros(1, device=dev))
        l = [{"params": [getattr(self, attr)], "lr": 1e-3, "name": name} for name, attr, _ in self.GROUPS]
        l.append({"params": [self._phase_offset], "lr": 0.0, "name": "phase_offset"})
        self.optimizer = optimizer_cls(l, lr=0.0, eps=1e-15)
        for grp in self.optimizer.param_groups:                  # one step so that the moments exist
            p = grp["params"][0]
            p.grad = torch.randn(p.shape, generator=g).to(dev) * 1e-3
        self.optimizer.step()
        self.optimizer.zero_grad(set_to_none=True)
        self.percent_dense = 0.01
        self.isotropic = False
        self.xyz_gradient_accum = (torch.rand((P, 1), generator=g) * 4e-4 * 50).to(dev)
        self.denom = torch.randint(0, 100, (P, 1), generator=g).float().to(dev)            # zeros -> nan gradients
        self.max_radii2D = (torch.rand(P, generator=g) * 30).to(dev)
        self.scaling_activation, self.scaling_inverse_activation = torch.exp, torch.log

    get_xyz = property(lambda self: self._xyz)
    get_scaling = property(lambda self: self.scaling_activation(self._scaling))
    get_opacity = property(lambda self: torch.sigmoid(self._opacity))

    def snapshot(self):
        d = {attr: getattr(self, attr).detach().cpu() for _, attr, _ in self.GROUPS}
        for k in ("xyz_gradient_accum", "denom", "max_radii2D"):
            d[k] = getattr(self, k).cpu()
        for grp in self.optimizer.param_groups:
            st = self.optimizer.state.get(grp["params"][0], None)
            if st is not None and "exp_avg" in st:
                d["m:" + grp["name"]], d["v:" + grp["name"]] = st["exp_avg"].cpu(), st["exp_avg_sq"].cpu()
        return d

    # scene/gaussian_model.py:494-514
    def prune_points(self, mask):
        valid_points_mask = ~mask
        t = prune_optimizer_eager(self.optimizer, valid_points_mask)
        for name, attr, _ in self.GROUPS:
            setattr(self, attr, t[name])
        self.xyz_gradient_accum = self.xyz_gradient_accum[valid_points_mask]
        self.denom = self.denom[valid_points_mask]
        self.max_radii2D = self.max_radii2D[valid_points_mask]

    # :539-569
    def densification_postfix(self, d):
        t = cat_tensors_to_optimizer_eager(self.optimizer, d)
        for name, attr, _ in self.GROUPS:
            setattr(self, attr, t[name])
        dev = self.get_xyz.device
        self.xyz_gradient_accum = torch.zeros((self.get_xyz.shape[0], 1), device=dev)
        self.denom = torch.zeros((self.get_xyz.shape[0], 1), device=dev)
        self.max_radii2D = torch.zeros((self.get_xyz.shape[0]), device=dev)

    # :571-601 (build_rotation: utils/general_utils.py:91-112)
    def densify_and_split(self, grads, grad_threshold, scene_extent, N=2):
        n_init_points = self.get_xyz.shape[0]
        dev = self.get_xyz.device
        padded_grad = torch.zeros((n_init_points), device=dev)
        padded_grad[:grads.shape[0]] = grads.squeeze()
        selected_pts_mask = torch.where(padded_grad >= grad_threshold, True, False)
        selected_pts_mask = torch.logical_and(selected_pts_mask,
                                              torch.max(self.get_scaling, dim=1).values > self.percent_dense * scene_extent)
        stds = self.get_scaling[selected_pts_mask].repeat(N, 1)
        means = torch.zeros((stds.size(0), 3), device=dev)
        samples = torch.normal(mean=means, std=stds)
        r = self._rotation[selected_pts_mask]
        q = r / torch.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1] + r[:, 2] * r[:, 2] + r[:, 3] * r[:, 3])[:, None]
        R = torch.zeros((q.size(0), 3, 3), device=dev)
        qr, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - qr * z); R[:, 0, 2] = 2 * (x * z + qr * y)
        R[:, 1, 0] = 2 * (x * y + qr * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - qr * x)
        R[:, 2, 0] = 2 * (x * z - qr * y); R[:, 2, 1] = 2 * (y * z + qr * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
        rots = R.repeat(N, 1, 1)
        d = {"xyz": torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + self.get_xyz[selected_pts_mask].repeat(N, 1)}
        d["scaling"] = self.scaling_inverse_activation(self.get_scaling[selected_pts_mask].repeat(N, 1) / (0.8 * N))
        d["rotation"] = self._rotation[selected_pts_mask].repeat(N, 1)
        for name, attr in (("f_dc_color", "_features_dc_color"), ("f_rest_color", "_features_rest_color"),
                           ("phase_f_dc", "_features_dc_phase"), ("phase_f_rest", "_features_rest_phase"),
                           ("amp_f_dc", "_features_dc_amp"), ("amp_f_rest", "_features_rest_amp")):
            d[name] = getattr(self, attr)[selected_pts_mask].repeat(N, 1, 1)
        d["opacity"] = self._opacity[selected_pts_mask].repeat(N, 1)
        d["f_seg_color"] = self._features_seg_color[selected_pts_mask].repeat(N, 1)
        self.densification_postfix(d)
        prune_filter = torch.cat((selected_pts_mask, torch.zeros(N * selected_pts_mask.sum(), device=dev, dtype=bool)))
        self.prune_points(prune_filter)

    # :603-622
    def densify_and_clone(self, grads, grad_threshold, scene_extent):
        selected_pts_mask = torch.where(torch.norm(grads, dim=-1) >= grad_threshold, True, False)
        selected_pts_mask = torch.logical_and(selected_pts_mask,
                                              torch.max(self.get_scaling, dim=1).values <= self.percent_dense * scene_extent)
        self.densification_postfix({name: getattr(self, attr)[selected_pts_mask] for name, attr, _ in self.GROUPS})

    # :624-640
    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size=20):
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        self.densify_and_clone(grads, max_grad, extent)
        self.densify_and_split(grads, max_grad, extent)
        prune_mask = (self.get_opacity < min_opacity).squeeze()
        if max_screen_size:
            big_points_vs = self.max_radii2D > max_screen_size
            big_points_ws = self.get_scaling.max(dim=1).values > 0.05 * extent
            small_points_ws = self.get_scaling.max(dim=1).values < 0.001 * extent
            prune_mask = torch.logical_or(torch.logical_or(torch.logical_or(prune_mask, big_points_vs), big_points_ws), small_points_ws)
        self.prune_points(prune_mask)
