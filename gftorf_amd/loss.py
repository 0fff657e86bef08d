"""The image terms of the training loss as one launch forward and one backward (``csrc/k_loss.hip``,
``include/gftorf_loss.h``): ``ssim`` (``utils/loss_utils.py:76-123``) and ``l2_loss`` (``:51-53``) of an image against its
ground truth, which the reference combines as ``(1 - lambda_dssim) * L + lambda_dssim * (1 - ssim(image, gt))``
(``train.py:196-231``).  In eager PyTorch the pair is eight grouped 11x11 convolutions and ~25 elementwise launches per
iteration -- at the reference's image size more device time than both rasterizer calls.  Same window (the reference's fp32
weights), same zero padding, same formula; gradients flow to the first image only (the ground truth has none in the
reference either).  There is no CPU path.
"""
import ctypes as C
from math import exp

import torch

from . import _lib

_WINDOW = 11
# utils/loss_utils.py:76-78 gaussian(window_size, 1.5): fp32 weights, normalised in fp32
_g = torch.tensor([exp(-(x - _WINDOW // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(_WINDOW)], dtype=torch.float32)
_WEIGHTS = (C.c_float * _WINDOW)(*[float(v) for v in (_g / _g.sum())])


def _img(t, name):
    if t.device.type != "cuda":
        raise RuntimeError("gftorf_amd.loss: %s is on %s; the loss kernels run on a HIP device only, there is no CPU path" % (name, t.device))
    if t.dim() == 4 and t.size(0) == 1:
        t = t[0]
    if t.dim() != 3:
        raise RuntimeError("gftorf_amd.loss: %s must be [C, H, W] (or [1, C, H, W]), got %s" % (name, tuple(t.shape)))
    return t.float().contiguous()


class _SsimL2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2):
        lib = _lib.load()
        a, b = _img(img1, "img1"), _img(img2.detach(), "img2")
        if a.shape != b.shape or a.device != b.device:
            raise RuntimeError("gftorf_amd.loss: images differ in shape or device: %s, %s" % (tuple(a.shape), tuple(b.shape)))
        Cn, H, W = (int(s) for s in a.shape)
        need_bw = ctx.needs_input_grad[0]
        blocks = int(lib.gft_ssim_blocks(Cn, H, W))
        partials = torch.empty((blocks, 2), device=a.device, dtype=torch.float32)
        maps = torch.empty((3, Cn, H, W), device=a.device, dtype=torch.float32) if need_bw else None
        a_d = a.detach()
        with _lib.on_device(a.device):
            _lib.check(lib.gft_ssim_l2_forward(_lib.raw_stream(a.device), Cn, H, W, a_d.data_ptr(), b.data_ptr(), _WEIGHTS,
                                               maps.data_ptr() if maps is not None else None, partials.data_ptr()))
        sums = partials.sum(0) / float(Cn * H * W)
        ctx.shape = (Cn, H, W, tuple(img1.shape))
        if need_bw:
            ctx.save_for_backward(a_d, b, maps)
        ctx.set_materialize_grads(False)
        return sums[0], sums[1]

    @staticmethod
    def backward(ctx, g_ssim, g_l2):
        lib = _lib.load()
        a, b, maps = ctx.saved_tensors
        Cn, H, W, in_shape = ctx.shape
        grad = torch.empty_like(a)
        f = lambda g: None if g is None else g.detach().float().reshape(1).contiguous()
        gs, gl = f(g_ssim), f(g_l2)
        with _lib.on_device(a.device):
            _lib.check(lib.gft_ssim_l2_backward(_lib.raw_stream(a.device), Cn, H, W, a.data_ptr(), b.data_ptr(), _WEIGHTS,
                                                maps.data_ptr(), gs.data_ptr() if gs is not None else None,
                                                gl.data_ptr() if gl is not None else None, 1.0 / (Cn * H * W), 1.0 / (Cn * H * W),
                                                grad.data_ptr()))
        return grad.view(in_shape), None


class _WeightedLoss(torch.autograd.Function):
    """w_l2 * l2_loss + w_dssim * (1 - ssim) as ONE tensor: the kernels of _SsimL2, the weights folded into the partial sums'
    combination (one dot product) and into the backward kernel's scales -- none of the eager scalar arithmetic around the two
    terms (train.py:196-231: five launches forward, six backward)."""

    @staticmethod
    def forward(ctx, img1, img2, w_l2, w_dssim):
        lib = _lib.load()
        a, b = _img(img1, "img1"), _img(img2.detach(), "img2")
        if a.shape != b.shape or a.device != b.device:
            raise RuntimeError("gftorf_amd.loss: images differ in shape or device: %s, %s" % (tuple(a.shape), tuple(b.shape)))
        Cn, H, W = (int(s) for s in a.shape)
        need_bw = ctx.needs_input_grad[0]
        blocks = int(lib.gft_ssim_blocks(Cn, H, W))
        partials = torch.empty((blocks, 2), device=a.device, dtype=torch.float32)
        maps = torch.empty((3, Cn, H, W), device=a.device, dtype=torch.float32) if need_bw else None
        a_d = a.detach()
        with _lib.on_device(a.device):
            _lib.check(lib.gft_ssim_l2_forward(_lib.raw_stream(a.device), Cn, H, W, a_d.data_ptr(), b.data_ptr(), _WEIGHTS,
                                               maps.data_ptr() if maps is not None else None, partials.data_ptr()))
        n = float(Cn * H * W)
        key = (a.device, blocks, float(w_l2), float(w_dssim), n)
        wv = _WeightedLoss._weights.get(key)
        if wv is None:          # (per block: [ssim sum, l2 sum] -> -w_dssim / n, w_l2 / n)
            if len(_WeightedLoss._weights) > 16:
                _WeightedLoss._weights.clear()
            wv = _WeightedLoss._weights[key] = torch.tensor([-float(w_dssim) / n, float(w_l2) / n], device=a.device).repeat(blocks)
        loss = torch.dot(partials.view(-1), wv) + float(w_dssim)
        ctx.shape = (Cn, H, W, tuple(img1.shape), float(w_l2), float(w_dssim))
        if need_bw:
            ctx.save_for_backward(a_d, b, maps)
        return loss

    _weights = {}

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        a, b, maps = ctx.saved_tensors
        Cn, H, W, in_shape, w_l2, w_dssim = ctx.shape
        grad = torch.empty_like(a)
        gp = g.detach().float().reshape(1).contiguous()
        n = float(Cn * H * W)
        with _lib.on_device(a.device):
            _lib.check(lib.gft_ssim_l2_backward(_lib.raw_stream(a.device), Cn, H, W, a.data_ptr(), b.data_ptr(), _WEIGHTS,
                                                maps.data_ptr(), gp.data_ptr(), gp.data_ptr(), -w_dssim / n, w_l2 / n, grad.data_ptr()))
        return grad.view(in_shape), None, None, None


def weighted_loss(img1, img2, w_l2, w_dssim):
    """``w_l2 * l2_loss(img1, img2) + w_dssim * (1 - ssim(img1, img2))`` -- the combination train.py:196-231 forms from the two
    terms (``w_l2 = lambda * (1 - lambda_dssim)``, ``w_dssim = lambda * lambda_dssim``) -- as one 0-dim tensor: one launch
    forward + one dot product, one launch backward."""
    if isinstance(img2, torch.Tensor) and img2.requires_grad:
        raise NotImplementedError("gftorf_amd.loss: gradients flow to the first image only")
    return _WeightedLoss.apply(img1, img2, float(w_l2), float(w_dssim))


def ssim_l2(img1, img2):
    """``(ssim(img1, img2), l2_loss(img1, img2))`` of utils/loss_utils.py as two 0-dim tensors from one launch."""
    if isinstance(img2, torch.Tensor) and img2.requires_grad:
        raise NotImplementedError("gftorf_amd.loss: gradients flow to the first image only")
    return _SsimL2.apply(img1, img2)


def ssim(img1, img2, window_size=11, size_average=True):
    """Drop-in for ``utils.loss_utils.ssim`` (window 11, ``size_average=True``: what train.py passes)."""
    if window_size != _WINDOW or not size_average:
        raise NotImplementedError("gftorf_amd.loss.ssim: window_size=11 and size_average=True only")
    return ssim_l2(img1, img2)[0]


def l2_loss(network_output, gt):
    """Drop-in for ``utils.loss_utils.l2_loss``."""
    return ssim_l2(network_output, gt)[1]
