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
