# -*- coding: utf-8 -*-
"""Stock-torch compositions for argument combinations the HIP kernels do not cover (such combinations "may fall back
to stock torch ops but must not change results").  Written from the maths of SURVEY Appendix A.4 -- valid / reflect-padded
correlation with the Gaussian window, clamped variances, the SSIM and contrast maps -- for

  * `use_padding=True` of the SSIM family (reference core/loss.py:42-49: F.pad(img, k // 2, 'reflect') before every window pass),
  * `size_average=False` (per-pixel maps instead of per-sample means, core/loss.py:103-108),
  * images smaller than the window (the reference shrinks the window to min(win, h, w), core/loss.py:67-71),
  * core.metric.calc_ssim with other windows / `full=True` (core/metric.py:316-364).

Everything here is differentiable through autograd like the reference's own code; nothing here is on the hot path (train.py and
test.py use the defaults, which run on csrc/loss.hip).  Pinned by golden F16 (tests/golden/make_golden.py, tests/test_host_cpu.py).
"""
from math import exp

import torch
import torch.nn.functional as F

eps = 1e-7
MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def window(win_size, loss_side=True):
    """[1,1,k,k] outer product of the normalised 1-D Gaussian (fp32, NOT re-normalised).  The loss module ties sigma to the window
    size for k != 11 (core/loss.py:32-39); the metric always uses 1.5 (core/metric.py:299-303)."""
    sigma = 1.5 if (win_size == 11 or not loss_side) else 0.15 * (win_size - 1)
    taps = torch.tensor([exp(-(i - win_size // 2) ** 2 / (2.0 * sigma ** 2)) for i in range(win_size)], dtype=torch.float32)
    taps = (taps / taps.sum()).unsqueeze(1)
    return torch.mm(taps, taps.t())[None, None]


def _blur(img, win, use_padding):
    if use_padding:
        p = win.shape[-1] // 2
        img = F.pad(img, (p, p, p, p), 'reflect')
    return F.conv2d(img, win, groups=img.shape[1])


def ssim_maps(img1, img2, win, data_range, use_padding):
    """(ssim map, cs map, clamped sigma1^2 map) of Appendix A.4"""
    win = win.to(img1)
    mu1, mu2 = _blur(img1, win, use_padding), _blur(img2, win, use_padding)
    var1 = (_blur(img1 * img1, win, use_padding) - mu1 * mu1).clamp(min=0)
    var2 = (_blur(img2 * img2, win, use_padding) - mu2 * mu2).clamp(min=0)
    cov = _blur(img1 * img2, win, use_padding) - mu1 * mu2
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    v1, v2 = 2.0 * cov + c2, var1 + var2 + c2
    cs = v1 / v2
    ssim = (2.0 * mu1 * mu2 + c1) * v1 / ((mu1 * mu1 + mu2 * mu2 + c1) * v2)
    return ssim, cs, var1.clamp(min=1e-4)


def ssim_terms(img1, img2, win_size=11, data_range=1.0, use_padding=False, size_average=True, win=None):
    """the dict core.loss.SSIM returns (core/loss.py:52-110)"""
    if win is None:
        win = window(min(win_size, img1.shape[-2], img1.shape[-1]))
    ssim, cs, sigma = ssim_maps(img1, img2, win, data_range, use_padding)
    if size_average:
        ssim, cs, sigma = (t.mean(dim=(1, 2, 3)) for t in (ssim, cs, sigma))
    return {'ssim': ssim, 'cs': cs, 'sigma': sigma}


def msssim(img1, img2, win_size=11, data_range=1.0, use_padding=False, size_average=True):
    """five-level product of cs (levels 0-3) and ssim (level 4) on a 2x2 average-pool pyramid, odd sizes reflect-padded by one
    (core/loss.py:113-160)"""
    win = window(min(win_size, img1.shape[-2], img1.shape[-1]))
    weights = torch.tensor(MS_WEIGHTS, dtype=torch.float32).to(img1)
    a, b, vals = img1, img2, []
    for lvl in range(len(MS_WEIGHTS)):
        out = ssim_terms(a, b, win_size, data_range, use_padding, size_average, win)
        if lvl == len(MS_WEIGHTS) - 1:
            vals.append(out['ssim'])
            break
        vals.append(out['cs'])
        ph, pw = a.shape[-2] % 2, a.shape[-1] % 2
        a = F.avg_pool2d(F.pad(a, (0, pw, 0, ph), 'reflect'), 2, 2)
        b = F.avg_pool2d(F.pad(b, (0, pw, 0, ph), 'reflect'), 2, 2)
    vals = torch.stack(vals, dim=0).clamp(min=eps)
    return torch.prod(vals ** weights.unsqueeze(1), dim=0)


def weighted_pair(out1, out2):
    """sigma-weighted mean of two SSIM dicts (core/loss.py:230-234, :262-267)"""
    gamma = out1['sigma'] / (out1['sigma'] + out2['sigma']).clamp(min=eps)
    return (gamma * out1['ssim']).mean() + ((1.0 - gamma) * out2['ssim']).mean()


def mswssim(img1, img2, imgf, win_sizes=(11, 9, 7, 5, 3), data_range=1.0, use_padding=False, size_average=False):
    total = 0.0
    for k in win_sizes:
        o1 = ssim_terms(img1, imgf, k, data_range, use_padding, size_average, window(k))
        o2 = ssim_terms(img2, imgf, k, data_range, use_padding, size_average, window(k))
        total = total + weighted_pair(o1, o2)
    return total / len(win_sizes)


def ssim_loss(mode, img1, img2, imgf, data_range, use_padding, weight):
    """SSIMLoss.forward (core/loss.py:252-284) for any mode"""
    if mode == 'ssim':
        s = 0.5 * (ssim_terms(img1, imgf, 11, data_range, use_padding)['ssim'].mean()
                   + ssim_terms(img2, imgf, 11, data_range, use_padding)['ssim'].mean())
    elif mode == 'w-ssim':
        s = weighted_pair(ssim_terms(img1, imgf, 11, data_range, use_padding), ssim_terms(img2, imgf, 11, data_range, use_padding))
    elif mode == 'ms-ssim':
        s = 0.5 * (msssim(img1, imgf, 11, data_range, use_padding).mean() + msssim(img2, imgf, 11, data_range, use_padding).mean())
    elif mode == 'msw-ssim':
        s = mswssim(img1, img2, imgf, (11, 9, 7, 5, 3), data_range, use_padding)
    else:
        raise ValueError("only supported ['ssim', 'w-ssim', 'ms-ssim', 'msw-ssim'] mode")
    return weight * (1.0 - s)


def metric_ssim(img1, img2, win_size=11, data_range=255.0, use_padding=False, size_average=True, full=False):
    """core.metric.calc_ssim (core/metric.py:316-364): scalar means over the whole batch, sigma 1.5 for every window size"""
    win = window(min(win_size, img1.shape[-2], img1.shape[-1]), loss_side=False)
    ssim, cs, _ = ssim_maps(img1, img2, win, data_range, use_padding)
    if size_average:
        ssim, cs = ssim.mean(), cs.mean()
    return (ssim, cs) if full else ssim


def nuclear_pooling(tensor):
    """channel_pooling(mode='nuclear') (core/fusion.py:127-134): [1,C,1,1] nuclear norms (sum of singular values) of the FIRST
    sample's channel maps clamped at eps -- batch entries beyond the first are ignored by the reference too"""
    c = tensor.shape[1]
    s = torch.linalg.svdvals(tensor[0].clamp(min=eps))      # [C, min(h, w)]
    return s.sum(dim=1).reshape(1, c, 1, 1).to(tensor.device)
