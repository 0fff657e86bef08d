"""Per-Gaussian bookkeeping around the rasterizer (bookkeeping part).

Fused replacements of the boolean-mask tensor surgery of the reference's training loop -- in eager
PyTorch every ``t[mask]`` / ``t[mask] += ...`` is a ``nonzero()`` with a host synchronisation plus a
gather or scatter, which at 1 M Gaussians costs more per iteration than the rasterizer does:

* :func:`add_densification_stats` -- ``train.py:443`` + ``GaussianModel.add_densification_stats``
  (``scene/gaussian_model.py:648-654``) in one in-place pass.
* :func:`select_rows` -- ``[t[mask] for t in tensors]`` with one rank computation for all tensors.
* :func:`prune_optimizer`, :func:`cat_tensors_to_optimizer` -- ``GaussianModel._prune_optimizer`` /
  ``cat_tensors_to_optimizer`` (``scene/gaussian_model.py:473-492, 516-537``) on top of it: same
  ``param_groups`` / ``state`` surgery, same results (bit-identical: pure data movement).
* :func:`prune_points`, :func:`densification_postfix`, :func:`densify_and_clone`, :func:`densify_and_split`,
  :func:`densify_and_prune`, :func:`prune` -- the composites of ``scene/gaussian_model.py:494-514, 539-646`` as
  functions of the model object (``pc``: anything with the reference ``GaussianModel``'s attributes): the method
  bodies a maintainer replaces by one call each.  Selection masks are the reference's own torch expressions, the
  split's ``torch.normal`` draw is torch's (same generator, same samples); every ``t[mask]`` is a row compaction
  with one rank computation per mask.

Kernels: ``csrc/k_densify.hip`` behind ``include/gftorf_densify.h``; no CPU path.
"""
import ctypes as C

import torch
from torch import nn

from . import _lib

_SKIP = ("phase_offset", "dc_offset")          # groups the reference leaves alone (gaussian_model.py:459,476,519)


def _dev_check(t, who):
    if t.device.type != "cuda":
        raise RuntimeError("gftorf_amd.densify.%s runs on a HIP device only (got a tensor on %s); there is no CPU path"
                           % (who, t.device))


def _mask_u8(mask, P, dev, who):
    if mask.dtype != torch.bool or mask.numel() != P:
        raise RuntimeError("%s: the mask must be a bool tensor with one entry per row (%d), got %s %s"
                           % (who, P, mask.dtype, tuple(mask.shape)))
    return mask.to(dev).contiguous().view(torch.uint8)


def add_densification_stats(xyz_gradient_accum, denom, max_radii2D, viewspace_grad, update_filter, pixels, radii,
                            apply_mask=None):
    """In place, for the rows of ``update_filter`` (``visibility_filter = radii > 0``):

        max_radii2D[f] = max(max_radii2D[f], radii[f])                                   (train.py:443)
        xyz_gradient_accum[f'] += norm(viewspace_grad[f', :2], dim=-1, keepdim=True) * pixels[f']
        denom[f'] += pixels[f']                                          (gaussian_model.py:648-654)

    with ``f' = f`` or, with ``apply_mask``, ``f & apply_mask`` (the reference's second branch; it only runs
    there when ``f`` is a subset of ``apply_mask``).  ``viewspace_grad`` is ``viewspace_point_tensor.grad``.
    ``max_radii2D`` / ``radii`` may be None to leave the radii alone."""
    lib = _lib.load()
    _dev_check(xyz_gradient_accum, "add_densification_stats")
    dev = xyz_gradient_accum.device
    P = xyz_gradient_accum.size(0)
    for t, n in ((xyz_gradient_accum, "xyz_gradient_accum"), (denom, "denom"), (max_radii2D, "max_radii2D")):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != P or t.device != dev):
            raise RuntimeError("%s must be a contiguous float32 tensor with %d entries on %s" % (n, P, dev))
    g = viewspace_grad.detach()
    if g.dtype != torch.float32 or tuple(g.shape) != (P, 3):
        raise RuntimeError("viewspace_grad must be float32 [%d, 3]" % P)
    g = g.contiguous()
    px = pixels.detach().float().contiguous()
    if px.numel() != P:
        raise RuntimeError("pixels must hold one value per Gaussian")
    uf = _mask_u8(update_filter, P, dev, "add_densification_stats")
    am = _mask_u8(apply_mask, P, dev, "add_densification_stats") if apply_mask is not None else None
    rd = None
    if max_radii2D is not None:
        if radii is None or radii.dtype != torch.int32 or radii.numel() != P:
            raise RuntimeError("radii must be the rasterizer's int32 [P] tensor")
        rd = radii.contiguous()
    ptr = lambda t: t.data_ptr() if t is not None and t.numel() else None
    with _lib.on_device(dev):
        _lib.check(lib.gft_densify_stats(_lib.raw_stream(dev), P, ptr(g), ptr(px), ptr(rd), ptr(uf), ptr(am),
                                         ptr(xyz_gradient_accum), ptr(denom), ptr(max_radii2D)))


class RowSelection:
    """``mask`` turned into an output row per kept row, once; :meth:`take` then compacts any tensor whose
    first dimension is the mask's (``t[mask]``)."""

    def __init__(self, mask):
        lib = _lib.load()
        _dev_check(mask, "RowSelection")
        self.dev = mask.device
        self.P = mask.numel()
        self.mask = _mask_u8(mask.reshape(-1), self.P, self.dev, "RowSelection")
        self.rank = torch.empty((self.P,), device=self.dev, dtype=torch.int32)
        scratch = torch.empty((lib.gft_rows_rank_scratch_bytes(self.P),), device=self.dev, dtype=torch.uint8)
        n = C.c_int64(0)
        with _lib.on_device(self.dev):
            _lib.check(lib.gft_rows_rank(_lib.raw_stream(self.dev), self.P,
                                         self.mask.data_ptr() if self.P else None, self.rank.data_ptr() if self.P else None,
                                         scratch.data_ptr() if self.P else None, C.byref(n)))
        self.count = int(n.value)

    def take(self, t, out=None):
        """``t[mask]`` (into ``out[:count]`` when given)."""
        lib = _lib.load()
        if t.size(0) != self.P or t.device != self.dev:
            raise RuntimeError("RowSelection.take: tensor with %d rows on %s, mask has %d rows on %s"
                               % (t.size(0), t.device, self.P, self.dev))
        src = t.detach().contiguous()
        row_bytes = (src.numel() // max(self.P, 1)) * src.element_size()
        if row_bytes % 4:
            return t.detach()[self.mask.view(torch.bool)] if out is None else out[:self.count].copy_(t.detach()[self.mask.view(torch.bool)])
        dst = torch.empty((self.count,) + tuple(src.shape[1:]), device=self.dev, dtype=src.dtype) if out is None else out
        if out is not None and (not out.is_contiguous() or out.dtype != src.dtype or tuple(out.shape[1:]) != tuple(src.shape[1:])
                                or out.size(0) < self.count):
            raise RuntimeError("RowSelection.take: unsuitable output tensor")
        if self.count and row_bytes:
            with _lib.on_device(self.dev):
                _lib.check(lib.gft_rows_gather(_lib.raw_stream(self.dev), self.P, self.mask.data_ptr(),
                                               self.rank.data_ptr(), src.data_ptr(), dst.data_ptr(), row_bytes))
        return dst


    def take_many(self, tensors):
        """``[t[mask] for t in tensors]`` (one launch per tensor: measured faster than one launch over a
        table of tensors, whose workgroups first have to find their tensor)."""
        return [self.take(t) for t in tensors]


def select_rows(mask, *tensors):
    """``[t[mask] for t in tensors]``."""
    return RowSelection(mask).take_many(list(tensors))


def prune_optimizer(optimizer, mask, skip=_SKIP, return_selection=False):
    """``GaussianModel._prune_optimizer`` (scene/gaussian_model.py:473-492): keeps the rows of ``mask`` in every
    group's parameter and Adam moments; returns ``{group name: new nn.Parameter}`` like the reference.  With
    ``return_selection`` the pair ``(dict, RowSelection)``: the selection compacts further per-Gaussian tensors
    (the densification statistics) without ranking the mask again."""
    sel = RowSelection(mask)
    groups = [g for g in optimizer.param_groups if g["name"] not in skip]
    todo = []
    for group in groups:
        p = group["params"][0]
        state = optimizer.state.get(p, None)
        todo.append(p)
        if state is not None:
            todo += [state["exp_avg"], state["exp_avg_sq"]]
    kept = iter(sel.take_many(todo))
    out = {}
    for group in groups:
        p = group["params"][0]
        state = optimizer.state.get(p, None)
        new_p = nn.Parameter(next(kept).requires_grad_(True))
        if state is not None:
            state["exp_avg"] = next(kept)
            state["exp_avg_sq"] = next(kept)
            del optimizer.state[p]
            group["params"][0] = new_p
            optimizer.state[new_p] = state
        else:
            group["params"][0] = new_p
        out[group["name"]] = new_p
    return (out, sel) if return_selection else out


def cat_tensors_to_optimizer(optimizer, tensors_dict, skip=_SKIP):
    """``GaussianModel.cat_tensors_to_optimizer`` (scene/gaussian_model.py:516-537): appends the new rows to
    every group's parameter, zero rows to its Adam moments."""
    out = {}
    for group in optimizer.param_groups:
        if group["name"] in skip:
            continue
        assert len(group["params"]) == 1
        ext = tensors_dict[group["name"]]
        p = group["params"][0]
        state = optimizer.state.get(p, None)
        new_p = nn.Parameter(torch.cat((p.detach(), ext), dim=0).requires_grad_(True))
        if state is not None:
            for k in ("exp_avg", "exp_avg_sq"):
                grown = torch.zeros((p.size(0) + ext.size(0),) + tuple(p.shape[1:]), device=p.device, dtype=p.dtype)
                grown[:p.size(0)] = state[k]
                state[k] = grown
            del optimizer.state[p]
            group["params"][0] = new_p
            optimizer.state[new_p] = state
        else:
            group["params"][0] = new_p
        out[group["name"]] = new_p
    return out


# ---- the composites of scene/gaussian_model.py:494-646 ------------------------------------------------------------
# optimizer group name -> attribute of the model (gaussian_model.py:498-508)
GROUP_ATTR = {"xyz": "_xyz", "f_dc_color": "_features_dc_color", "f_rest_color": "_features_rest_color",
              "phase_f_dc": "_features_dc_phase", "phase_f_rest": "_features_rest_phase", "amp_f_dc": "_features_dc_amp",
              "amp_f_rest": "_features_rest_amp", "opacity": "_opacity", "scaling": "_scaling", "rotation": "_rotation",
              "f_seg_color": "_features_seg_color"}


def _adopt(pc, tensors):
    for name, attr in GROUP_ATTR.items():
        setattr(pc, attr, tensors[name])


def prune_points(pc, mask):
    """``GaussianModel.prune_points`` (gaussian_model.py:494-514): removes the Gaussians of ``mask``."""
    valid = ~mask
    tensors, sel = prune_optimizer(pc.optimizer, valid, return_selection=True)
    _adopt(pc, tensors)
    pc.xyz_gradient_accum = sel.take(pc.xyz_gradient_accum)
    pc.denom = sel.take(pc.denom)
    pc.max_radii2D = sel.take(pc.max_radii2D)


def densification_postfix(pc, new):
    """``GaussianModel.densification_postfix`` (gaussian_model.py:539-569); ``new``: group name -> rows to append."""
    _adopt(pc, cat_tensors_to_optimizer(pc.optimizer, new))
    n, dev = pc.get_xyz.shape[0], pc.get_xyz.device
    pc.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
    pc.denom = torch.zeros((n, 1), device=dev)
    pc.max_radii2D = torch.zeros((n,), device=dev)


def _rotation_matrices(r):
    """``build_rotation`` (utils/general_utils.py:91-112): (r, x, y, z) quaternions, normalised here, to 3x3."""
    q = r / torch.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1] + r[:, 2] * r[:, 2] + r[:, 3] * r[:, 3])[:, None]
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.zeros((q.size(0), 3, 3), device=r.device)
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - w * z)
    R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y)
    R[:, 2, 1] = 2 * (y * z + w * x)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def densify_and_clone(pc, grads, grad_threshold, scene_extent):
    """``GaussianModel.densify_and_clone`` (gaussian_model.py:603-622): small Gaussians with a large view-space
    gradient are duplicated."""
    mask = torch.where(torch.norm(grads, dim=-1) >= grad_threshold, True, False)
    mask = torch.logical_and(mask, torch.max(pc.get_scaling, dim=1).values <= pc.percent_dense * scene_extent)
    sel = RowSelection(mask)
    densification_postfix(pc, {name: sel.take(getattr(pc, attr)) for name, attr in GROUP_ATTR.items()})


def densify_and_split(pc, grads, grad_threshold, scene_extent, N=2):
    """``GaussianModel.densify_and_split`` (gaussian_model.py:571-601): large Gaussians with a large gradient are
    replaced by N smaller ones sampled inside them (torch.normal: torch's generator decides the samples)."""
    n_init = pc.get_xyz.shape[0]
    dev = pc.get_xyz.device
    padded = torch.zeros((n_init,), device=dev)
    padded[:grads.shape[0]] = grads.squeeze()
    mask = torch.where(padded >= grad_threshold, True, False)
    mask = torch.logical_and(mask, torch.max(pc.get_scaling, dim=1).values > pc.percent_dense * scene_extent)
    sel = RowSelection(mask)
    scaling_sel = sel.take(pc.get_scaling)
    stds = scaling_sel.repeat(N, 1)
    means = torch.zeros((stds.size(0), 3), device=dev)
    samples = torch.normal(mean=means, std=stds)
    rotation_sel = sel.take(pc._rotation)
    rots = _rotation_matrices(rotation_sel).repeat(N, 1, 1)
    new = {"xyz": torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + sel.take(pc.get_xyz).repeat(N, 1)}
    if getattr(pc, "isotropic", False):
        new["scaling"] = pc.scaling_inverse_activation(pc.scaling_activation(sel.take(pc._scaling)).repeat(N, 1) / (0.8 * N))
    else:
        new["scaling"] = pc.scaling_inverse_activation(scaling_sel.repeat(N, 1) / (0.8 * N))
    new["rotation"] = rotation_sel.repeat(N, 1)
    for name in ("f_dc_color", "f_rest_color", "phase_f_dc", "phase_f_rest", "amp_f_dc", "amp_f_rest"):
        new[name] = sel.take(getattr(pc, GROUP_ATTR[name])).repeat(N, 1, 1)
    new["opacity"] = sel.take(pc._opacity).repeat(N, 1)
    new["f_seg_color"] = sel.take(pc._features_seg_color).repeat(N, 1)
    densification_postfix(pc, new)
    prune_filter = torch.cat((mask, torch.zeros(N * sel.count, device=dev, dtype=torch.bool)))
    prune_points(pc, prune_filter)


def densify_and_prune(pc, max_grad, min_opacity, extent, max_screen_size=20):
    """``GaussianModel.densify_and_prune`` (gaussian_model.py:624-640)."""
    grads = pc.xyz_gradient_accum / pc.denom
    grads[grads.isnan()] = 0.0
    densify_and_clone(pc, grads, max_grad, extent)
    densify_and_split(pc, grads, max_grad, extent)
    prune_mask = (pc.get_opacity < min_opacity).squeeze()
    if max_screen_size:
        big_points_vs = pc.max_radii2D > max_screen_size
        big_points_ws = pc.get_scaling.max(dim=1).values > 0.05 * extent
        small_points_ws = pc.get_scaling.max(dim=1).values < 0.001 * extent
        prune_mask = torch.logical_or(torch.logical_or(torch.logical_or(prune_mask, big_points_vs), big_points_ws), small_points_ws)
    prune_points(pc, prune_mask)


def prune(pc, min_opacity):
    """``GaussianModel.prune`` (gaussian_model.py:642-646)."""
    prune_points(pc, (pc.get_opacity < min_opacity).squeeze())
