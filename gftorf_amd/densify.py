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
    with torch.cuda.device(dev):
        _lib.check(lib.gft_densify_stats(torch.cuda.current_stream(dev).cuda_stream, P, ptr(g), ptr(px), ptr(rd), ptr(uf), ptr(am),
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
        with torch.cuda.device(self.dev):
            _lib.check(lib.gft_rows_rank(torch.cuda.current_stream(self.dev).cuda_stream, self.P,
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
            with torch.cuda.device(self.dev):
                _lib.check(lib.gft_rows_gather(torch.cuda.current_stream(self.dev).cuda_stream, self.P, self.mask.data_ptr(),
                                               self.rank.data_ptr(), src.data_ptr(), dst.data_ptr(), row_bytes))
        return dst


    def take_many(self, tensors):
        """``[t[mask] for t in tensors]`` (one launch per tensor: measured faster than one launch over a
        table of tensors, whose workgroups first have to find their tensor)."""
        return [self.take(t) for t in tensors]


def select_rows(mask, *tensors):
    """``[t[mask] for t in tensors]``."""
    return RowSelection(mask).take_many(list(tensors))


def prune_optimizer(optimizer, mask, skip=_SKIP):
    """``GaussianModel._prune_optimizer`` (scene/gaussian_model.py:473-492): keeps the rows of ``mask`` in every
    group's parameter and Adam moments; returns ``{group name: new nn.Parameter}``."""
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
    return out, sel


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
