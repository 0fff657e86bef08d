"""Fused Adam for the Gaussian parameters (optimizer part).

``FusedAdam`` is a ``torch.optim.Adam`` whose ``step()`` runs one hand-written gfx950 kernel per
call -- all tensors of all groups that share betas / eps / weight decay in one launch, each with its own learning
rate and step count (``csrc/k_adam.hip``, ``include/gftorf_optim.h``) --
instead of torch's multi-tensor path (one kernel per arithmetic operation); constructor, ``param_groups``, ``state`` (``step`` / ``exp_avg`` / ``exp_avg_sq``),
``state_dict`` and ``zero_grad`` are torch's own, so the reference's densification code, which
edits the optimizer state in place (``scene/gaussian_model.py:456-540``), works unchanged.
Use: replace ``torch.optim.Adam(l, lr=0.0, eps=1e-15)`` at ``scene/gaussian_model.py:274`` by
``gftorf_amd.FusedAdam(l, lr=0.0, eps=1e-15)``.

Opt-in, not the reference's behaviour: ``step(visibility=mask)`` ("sparse Adam on visible
Gaussians") updates only the rows ``mask`` selects in every parameter whose first dimension is ``mask.numel()`` -- with
``visibility_filter`` of the iteration's render (``train.py:181``) that is the 10-20 % of the Gaussians that received a
gradient at all.  The other rows keep parameter AND moments (a dense Adam lets their moments decay and still moves
them by the decaying first moment); parameters of other shapes (the deformation network) take the dense step.
"""
import torch

from . import _lib


class FusedAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        if amsgrad or kw.get("maximize") or kw.get("capturable") or kw.get("differentiable"):
            raise NotImplementedError("gftorf_amd.FusedAdam: amsgrad / maximize / capturable / differentiable are not supported")
        kw.pop("foreach", None)
        kw.pop("fused", None)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, **kw)

    @torch.no_grad()
    def step(self, closure=None, visibility=None, row_params=None):
        """``visibility``: opt-in row mask (see the module docstring).  ``row_params``: the parameters the mask applies to
        (an iterable of tensors; default: every parameter whose first dimension equals ``visibility.numel()`` -- name
        them when another parameter, e.g. a network weight, could have that many rows by coincidence).  A row that is
        skipped keeps its moments, and the bias correction uses the tensor's one step count: a row that was skipped
        k times is corrected as if it had taken those k steps (dense Adam differs there as well as in the decay)."""
        loss = None
        rows = None
        if row_params is not None:
            row_params = {id(t) for t in row_params}
        if visibility is not None:
            if visibility.dim() != 1 or visibility.dtype not in (torch.bool, torch.uint8):
                raise RuntimeError("gftorf_amd.FusedAdam: visibility must be a 1-D bool / uint8 tensor (one entry per Gaussian)")
            rows = visibility.numel()
            visibility = visibility.contiguous()
            mask_u8 = visibility.view(torch.uint8) if visibility.dtype == torch.bool else visibility
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        # (parameter, gradient, moments, lr, step tensor) of every tensor that takes a step, bucketed by the settings
        # one launch shares: (device, betas, eps, weight decay)
        buckets = {}
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            lr, eps, wd = group["lr"], group["eps"], group["weight_decay"]
            if isinstance(lr, torch.Tensor):
                lr = float(lr)
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.device.type != "cuda":
                    raise RuntimeError("gftorf_amd.FusedAdam runs on a HIP device only (parameter on %s); there is no CPU path" % (p.device,))
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("gftorf_amd.FusedAdam: parameters must be contiguous float32 tensors")
                state = self.state[p]
                if len(state) == 0:
                    # same state as torch.optim.Adam._init_group
                    state["step"] = torch.tensor(0.0, dtype=torch.float32)
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                m, v = state["exp_avg"], state["exp_avg_sq"]
                if not (m.is_contiguous() and v.is_contiguous()):
                    raise RuntimeError("gftorf_amd.FusedAdam: optimizer state must be contiguous")
                grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                # the kernels move 16-byte pieces: a gradient that is a view at an odd offset (e.g. the rasterizer's
                # dc_offset gradient, element 1 of a two-float tensor) is copied once; parameter and moments are the
                # caller's own allocations and must be aligned as torch allocates them
                if grad.data_ptr() % 16:
                    grad = grad.clone()
                if p.data_ptr() % 16 or m.data_ptr() % 16 or v.data_ptr() % 16:
                    raise RuntimeError("gftorf_amd.FusedAdam: parameters and optimizer state must be 16-byte aligned")
                by_rows = rows is not None and p.dim() >= 1 and p.shape[0] == rows and rows > 0 and (
                    row_params is None or id(p) in row_params)
                if by_rows and visibility.device != p.device:
                    raise RuntimeError("gftorf_amd.FusedAdam: visibility is on %s, the parameter on %s" % (visibility.device, p.device))
                buckets.setdefault((p.device, float(beta1), float(beta2), float(eps), float(wd), by_rows), []).append(
                    (p, grad, m, v, float(lr), state["step"]))
        for (dev, beta1, beta2, eps, wd, by_rows), items in buckets.items():
            tab = (_lib.AdamTensor * len(items))()
            for e, (p, g, m, v, lr, st) in zip(tab, items):
                e.param, e.grad, e.exp_avg, e.exp_avg_sq, e.n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
                e.lr, e.step = lr, int(st) + 1
            with _lib.on_device(dev):
                if by_rows:
                    _lib.check(lib.gft_adam_step_rows(_lib.raw_stream(dev), len(items), tab, rows, mask_u8.data_ptr(), beta1, beta2,
                                                      eps, wd))
                else:
                    _lib.check(lib.gft_adam_step_multi(_lib.raw_stream(dev), len(items), tab, beta1, beta2, eps, wd))
            # the step counters advance only once the launch was accepted (a rejected table leaves every tensor of the
            # bucket and its counter as they were)
            torch._foreach_add_([it[5] for it in items], 1)
        return loss
