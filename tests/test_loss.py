"""gftorf_amd.loss.ssim_l2 (one launch forward, one backward) against the reference's formulas restated with torch
convolutions in float64 on the CPU (utils/loss_utils.py:51-53 l2_loss, :76-123 ssim: Gaussian window 11 / sigma 1.5, zero
padding, groups = channels) -- values and the gradient with respect to the first image."""
from math import exp

import numpy as np
import pytest
import torch
import torch.nn.functional as F


def ref_ssim_l2(a, b, size=11, sigma=1.5):
    g = torch.tensor([exp(-(x - size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(size)], dtype=torch.float32)
    g = (g / g.sum()).double().unsqueeze(1)
    C = a.shape[0]
    win = (g @ g.t()).expand(C, 1, size, size).contiguous()
    a4, b4 = a.unsqueeze(0), b.unsqueeze(0)
    blur = lambda x: F.conv2d(x, win, padding=size // 2, groups=C)
    mu1, mu2 = blur(a4), blur(b4)
    s1, s2, s12 = blur(a4 * a4) - mu1 * mu1, blur(b4 * b4) - mu2 * mu2, blur(a4 * b4) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))
    return m.mean(), ((a - b) ** 2).mean()


def test_loss_rejects_cpu_tensors():
    from gftorf_amd import loss
    with pytest.raises(RuntimeError, match="HIP device only"):
        loss.ssim_l2(torch.zeros(2, 8, 8), torch.zeros(2, 8, 8))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 240, 320), (3, 37, 53), (1, 16, 16), (7, 5, 70)])
def test_ssim_l2_matches_the_reference_formulas(shape, gpu):
    from gftorf_amd import loss
    rng = np.random.default_rng(sum(shape))
    # a smooth image pair with structure (pure noise has ssim ~ 0 and says little) plus noise
    yy, xx = np.meshgrid(np.linspace(0, 3, shape[1]), np.linspace(0, 4, shape[2]), indexing="ij")
    base = np.stack([np.sin(yy * (c + 1)) * np.cos(xx * (c + 2)) for c in range(shape[0])])
    a_np = (0.5 * base + 0.1 * rng.normal(size=shape)).astype(np.float32)
    b_np = (0.5 * base + 0.05 * rng.normal(size=shape) + 0.02).astype(np.float32)
    a64 = torch.tensor(a_np, dtype=torch.float64, requires_grad=True)
    s_ref, l_ref = ref_ssim_l2(a64, torch.tensor(b_np, dtype=torch.float64))
    (0.3 * s_ref + 1.7 * l_ref).backward()
    a = torch.tensor(a_np, device=gpu, requires_grad=True)
    s, l = loss.ssim_l2(a, torch.tensor(b_np, device=gpu))
    (0.3 * s + 1.7 * l).backward()
    assert abs(float(s) - float(s_ref)) < 2e-6 and abs(float(l) - float(l_ref)) < 1e-6 * max(float(l_ref), 1e-3)
    g, g_ref = a.grad.cpu().double().numpy(), a64.grad.numpy()
    assert np.abs(g - g_ref).max() < 2e-5 * np.abs(g_ref).max()
    # the drop-in names, and a [1, C, H, W] input
    assert float(loss.ssim(a.detach()[None], torch.tensor(b_np, device=gpu)[None])) == float(s)
    assert float(loss.l2_loss(a.detach(), torch.tensor(b_np, device=gpu))) == float(l)
    # only one of the two means is used: the other's gradient is None
    a2 = torch.tensor(a_np, device=gpu, requires_grad=True)
    loss.ssim_l2(a2, torch.tensor(b_np, device=gpu))[1].backward()
    np.testing.assert_allclose(a2.grad.cpu().numpy(), 2.0 * (a_np - b_np) / a_np.size, rtol=1e-6, atol=1e-12)


@pytest.mark.gpu
def test_weighted_loss_is_the_combination_of_the_two_terms(gpu):
    """weighted_loss(img, gt, w_l2, w_dssim) = w_l2 * l2 + w_dssim * (1 - ssim) (train.py:196-231's combination) as one tensor:
    the value and the image gradient of the two-term form, and of the float64 formulas."""
    from gftorf_amd import loss
    gen = torch.Generator().manual_seed(21)
    shape = (2, 96, 128)
    gt = torch.rand(shape, generator=gen).to(gpu)
    img0 = (gt.cpu() + 0.1 * torch.randn(shape, generator=gen)).to(gpu)
    w_l2, w_d = 1.0 * (1.0 - 0.2), 1.0 * 0.2
    a = img0.clone().requires_grad_()
    s_val, l2 = loss.ssim_l2(a, gt)
    two = w_l2 * l2 + w_d * (1.0 - s_val)
    two.backward()
    b = img0.clone().requires_grad_()
    one = loss.weighted_loss(b, gt, w_l2, w_d)
    assert one.dim() == 0
    (3.0 * one).backward()          # (an upstream gradient other than 1)
    torch.testing.assert_close(one, two.detach(), rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(b.grad, 3.0 * a.grad, rtol=2e-5, atol=1e-9)
    rs, rl = ref_ssim_l2(img0.double().cpu(), gt.double().cpu())
    assert abs(float(one) - float(w_l2 * rl + w_d * (1 - rs))) < 2e-6
