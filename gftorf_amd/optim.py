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

``FusedAdam(..., capturable=True)``: the step inside a captured iteration (``torch.cuda.graph``).  Nothing the update
depends on is baked into the launch: ``state["step"]`` is a 0-dim fp32 DEVICE tensor (as torch's capturable Adam keeps it),
the learning rates are read on the device from a small buffer that every ``step()`` refreshes from ``param_groups`` through
pinned host memory -- a copy node when captured, so a replay sees whatever ``refresh_lr()`` (or any eager bookkeeping
that ends in it: the reference's ``update_learning_rate``, scene/gaussian_model.py:294-310, then ``refresh_lr()``) wrote
there since.  Bias corrections are formed in double precision on the device: N replays leave the parameters N eager
non-capturable steps would (to the rounding of one double ``pow``).  ``visibility`` is not available in this mode.
"""
import torch

from . import _lib


class FusedAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        if amsgrad or kw.get("maximize") or kw.get("differentiable"):
            raise NotImplementedError("gftorf_amd.FusedAdam: amsgrad / maximize / differentiable are not supported")
        kw.pop("foreach", None)
        kw.pop("fused", None)
        self._gft_capturable = bool(kw.pop("capturable", False))
        self._gft_dev = {}              # device -> dict(step, lr, lr_host, factors, used): the buffers of the capturable mode
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, **kw)

    # ---- capturable mode ------------------------------------------------------------------------------------------
    _SLOTS = 512

    def _buffers(self, dev):
        b = self._gft_dev.get(dev)
        if b is None:
            b = self._gft_dev[dev] = dict(
                step=torch.zeros((self._SLOTS,), device=dev, dtype=torch.float32),
                lr=torch.zeros((self._SLOTS,), device=dev, dtype=torch.float64),
                lr_host=torch.zeros((self._SLOTS,), dtype=torch.float64).pin_memory(),
                factors=torch.zeros((2 * self._SLOTS,), device=dev, dtype=torch.float32), used=0, slot_of={})
            b["lr_np"] = b["lr_host"].numpy()
        return b

    def _slot(self, p, state):
        """The parameter's slot in the device buffers; its ``state["step"]`` becomes (or stays) the 0-dim view of the slot's
        count.  A count that came from elsewhere -- a state_dict, steps taken before the mode was switched on, the
        reference's densification code re-keying the state (scene/gaussian_model.py:456-540 keeps the dict) -- is copied in."""
        b = self._buffers(p.device)
        st = state.get("step")
        base, end = b["step"].data_ptr(), b["step"].data_ptr() + 4 * self._SLOTS
        if isinstance(st, torch.Tensor) and st.is_cuda and base <= st.data_ptr() < end and st.dtype == torch.float32:
            return (st.data_ptr() - base) // 4, b
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("gftorf_amd.FusedAdam(capturable=True): take one eager step() before capturing (the optimizer "
                               "state of a parameter is created, or adopted, outside the graph)")
        if b["used"] >= self._SLOTS:
            # (slots of parameters that no longer exist -- densification replaces the tensors -- are reclaimed)
            alive = {id(q) for g in self.param_groups for q in g["params"]}
            b["slot_of"] = {k: v for k, v in b["slot_of"].items() if k in alive}
            free = sorted(set(range(self._SLOTS)) - set(b["slot_of"].values()))
            if not free:
                raise RuntimeError("gftorf_amd.FusedAdam(capturable=True): more than %d parameter tensors" % self._SLOTS)
            slot = free[0]
        else:
            slot = b["used"]
            b["used"] += 1
        b["slot_of"][id(p)] = slot
        view = b["step"][slot]
        view.fill_(float(st) if st is not None else 0.0)
        state["step"] = view
        return slot, b

    def refresh_lr(self):
        """Writes the groups' current learning rates where a captured step reads them (pinned host memory: no device work,
        no synchronisation).  Call it after the scheduler has set ``param_groups[...]["lr"]`` and before the replay."""
        for group in self.param_groups:
            lr = float(group["lr"])
            for p in group["params"]:
                b = self._gft_dev.get(p.device)
                if b is not None:
                    slot = b["slot_of"].get(id(p))
                    if slot is not None:
                        b["lr_np"][slot] = lr

    def _step_capturable(self, lib):
        import ctypes as C
        buckets = {}
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            eps, wd = group["eps"], group["weight_decay"]
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
                if "exp_avg" not in state:
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                slot, b = self._slot(p, state)
                b["lr_np"][slot] = float(group["lr"])
                m, v = state["exp_avg"], state["exp_avg_sq"]
                grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if grad.data_ptr() % 16:
                    grad = grad.clone()
                if p.data_ptr() % 16 or m.data_ptr() % 16 or v.data_ptr() % 16 or not (m.is_contiguous() and v.is_contiguous()):
                    raise RuntimeError("gftorf_amd.FusedAdam: parameters and optimizer state must be contiguous and 16-byte aligned")
                buckets.setdefault((p.device, float(beta1), float(beta2), float(eps), float(wd)), []).append((p, grad, m, v, slot))
        for (dev, beta1, beta2, eps, wd), items in buckets.items():
            b = self._buffers(dev)
            # the learning rates of this step: host values -> device, a copy node under capture (a replay re-reads the pinned
            # buffer: refresh_lr)
            b["lr"].copy_(b["lr_host"], non_blocking=True)
            n = len(items)
            tab = (_lib.AdamTensor * n)()
            lrs, steps = (C.c_void_p * n)(), (C.c_void_p * n)()
            lr0, st0 = b["lr"].data_ptr(), b["step"].data_ptr()
            for i, (e, (p, g, m, v, slot)) in enumerate(zip(tab, items)):
                e.param, e.grad, e.exp_avg, e.exp_avg_sq, e.n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
                e.lr, e.step = 0.0, 0
                lrs[i], steps[i] = lr0 + 8 * slot, st0 + 4 * slot
            with _lib.on_device(dev):
                _lib.check(lib.gft_adam_step_multi_dev(_lib.raw_stream(dev), n, tab, lrs, steps, b["factors"].data_ptr(),
                                                       beta1, beta2, eps, wd))

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
        if self._gft_capturable:
            if visibility is not None:
                raise NotImplementedError("gftorf_amd.FusedAdam(capturable=True): step(visibility=...) is not available")
            self._step_capturable(lib)
            return loss
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
