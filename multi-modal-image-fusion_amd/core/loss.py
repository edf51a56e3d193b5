# -*- coding: utf-8 -*-
"""Fusion losses -- API mirror of the reference's core/loss.py (SSIM :163-185, SSIMLoss :240-284,
PixelLoss :287-304, GradLoss :307-344, NormLoss :361-385) on the fused HIP loss kernels: each call
computes the loss AND d(loss)/d(imgf) in one pass; backward() only scales that gradient.
All loss arithmetic is fp32 regardless of the feature-map dtype.
"""
import ctypes as C

import torch
import torch.nn as nn

from mmif import tensor as T
from mmif._lib import check, lib

from . import _stock

__all__ = ['SSIM', 'MS_SSIM', 'MSW_SSIM', 'SSIMLoss', 'PixelLoss', 'GradLoss', 'TVLoss', 'NormLoss', 'FusionLoss']

eps = 1e-7


def _prep(img1, img2, imgf):
    T.require_device(imgf, "loss input")
    if imgf.dim() != 4 or imgf.shape[1] != 1:
        raise ValueError(f"fusion losses take single-channel images [B,1,H,W]; got {tuple(imgf.shape)}")
    if img1.shape != imgf.shape or img2.shape != imgf.shape:
        raise ValueError("img1, img2 and imgf must have the same shape")
    return img1.detach().contiguous().float(), img2.detach().contiguous().float(), imgf.detach().contiguous().float()


_ws_cache = {}


def _workspace(n, h, w, device):
    key = (n, h, w, device)
    ws = _ws_cache.get(key)
    if ws is None:
        ws = torch.empty(lib.mmif_loss_workspace(n, h, w) // 4 + 1, dtype=torch.float32, device=device)
        _ws_cache[key] = ws
    return ws


class _LossFn(torch.autograd.Function):
    """kind: 0 ssim, 1 pixel, 2 grad.  Returns a 0-dim loss; saves dloss/dimgf computed in the same launch."""

    @staticmethod
    def forward(ctx, imgf, img1, img2, kind, weight, a, b):
        i1, i2, f = _prep(img1, img2, imgf)
        n, _, h, w = f.shape
        need = ctx.needs_input_grad[0]
        out = torch.empty(1, dtype=torch.float32, device=f.device)
        grad = torch.empty_like(f) if need else None
        ws = _workspace(n, h, w, f.device)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        nbytes = ws.numel() * 4
        if kind == 0:
            check(lib.mmif_ssim_loss(p(i1), p(i2), p(f), n, h, w, weight, a, p(out), p(grad), p(ws), nbytes, T.stream_ptr()), "ssim_loss")
        elif kind == 1:
            check(lib.mmif_pixel_loss(p(i1), p(i2), p(f), n, h, w, weight, int(a), int(b), p(out), p(grad), p(ws), nbytes, T.stream_ptr()), "pixel_loss")
        else:
            check(lib.mmif_grad_loss(p(i1), p(i2), p(f), n, h, w, weight, int(a), int(b), p(out), p(grad), p(ws), nbytes, T.stream_ptr()), "grad_loss")
        ctx.grad = grad
        return out[0]

    @staticmethod
    def backward(ctx, g):
        if ctx.grad is None:
            return (None,) * 7
        return ctx.grad * g, None, None, None, None, None, None


_SSIM_MODES = {'w-ssim': 1, 'ms-ssim': 2, 'msw-ssim': 3}


class _ModeLossFn(torch.autograd.Function):
    """SSIMLoss 'w-ssim' | 'ms-ssim' | 'msw-ssim' (reference core/loss.py:259-277) on csrc/loss_modes.hip."""

    @staticmethod
    def forward(ctx, imgf, img1, img2, mode, weight, data_range):
        i1, i2, f = _prep(img1, img2, imgf)
        n, _, h, w = f.shape
        need = ctx.needs_input_grad[0]
        out = torch.empty(1, dtype=torch.float32, device=f.device)
        grad = torch.empty_like(f) if need else None
        nbytes = lib.mmif_ssim_loss_mode_workspace(n, h, w, mode)
        ws = torch.empty(nbytes // 4 + 1, dtype=torch.float32, device=f.device)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        check(lib.mmif_ssim_loss_mode(p(i1), p(i2), p(f), n, h, w, weight, data_range, mode, p(out), p(grad), p(ws), ws.numel() * 4,
                                      T.stream_ptr()), "ssim_loss_mode")
        ctx.grad = grad
        return out[0]

    @staticmethod
    def backward(ctx, g):
        if ctx.grad is None:
            return (None,) * 6
        return ctx.grad * g, None, None, None, None, None


class _TVFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, l2, weight):
        T.require_device(x, "TVLoss input")
        if x.dim() < 2:
            raise ValueError("TVLoss needs at least 2 dimensions")
        xf = x.detach().contiguous().float()
        h, w = xf.shape[-2:]
        n = xf.numel() // (h * w)
        need = ctx.needs_input_grad[0]
        out = torch.empty(1, dtype=torch.float32, device=xf.device)
        grad = torch.empty_like(xf) if need else None
        ws = torch.empty(lib.mmif_tv_loss_workspace() // 4 + 1, dtype=torch.float32, device=xf.device)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        check(lib.mmif_tv_loss(p(xf), n, h, w, weight, int(l2), p(out), p(grad), p(ws), ws.numel() * 4, T.stream_ptr()), "tv_loss")
        ctx.grad = grad
        return out[0]

    @staticmethod
    def backward(ctx, g):
        if ctx.grad is None:
            return None, None, None
        return ctx.grad * g, None, None


class SSIM(nn.Module):
    """Structural similarity (reference core/loss.py:163-185): {'ssim', 'cs', 'sigma'} per-sample means of (img1, img2) for a
    window of 3 / 5 / 7 / 9 / 11 taps.  'ssim' with the default 11x11 window is differentiable w.r.t. img2 (the fused image in
    every use of the reference); 'cs' and 'sigma' are values."""

    def __init__(self, win_size=11, data_range=1.0, use_padding=False, size_average=True):
        super(SSIM, self).__init__()
        self.win_size, self.data_range, self.use_padding, self.size_average = win_size, data_range, use_padding, size_average
        # argument combinations outside the HIP kernels (reflect-padded windows, per-pixel maps, other window sizes) run as stock torch
        # ops with the reference's results (core/_stock.py)
        self._stock = win_size not in (3, 5, 7, 9, 11) or bool(use_padding) or not size_average

    def forward(self, img1, img2):
        if self._stock or min(img1.shape[-2:]) < self.win_size:
            return _stock.ssim_terms(img1, img2, self.win_size, self.data_range, self.use_padding, self.size_average, _stock.window(self.win_size))
        i1, i2, _ = _prep(img1, img2, img2)
        n, _, h, w = i1.shape
        out = torch.empty((3, n), dtype=torch.float32, device=i1.device)
        ws = torch.empty(lib.mmif_ssim_loss_mode_workspace(n, h, w, 1) // 4 + 1, dtype=torch.float32, device=i1.device)
        p = lambda t: C.c_void_p(t.data_ptr())
        check(lib.mmif_ssim_terms(p(i1), p(i2), n, h, w, self.win_size, float(self.data_range), p(out), p(ws), ws.numel() * 4,
                                  T.stream_ptr()), "ssim_terms")
        res = {'ssim': out[0], 'cs': out[1], 'sigma': out[2]}
        if self.win_size == 11 and torch.is_grad_enabled() and img2.requires_grad:
            # differentiable path: the batch-1 loss kernel per sample (loss = 1 - S with both sources = img1)
            res['ssim'] = torch.stack([1.0 - _LossFn.apply(img2[i:i + 1], img1[i:i + 1], img1[i:i + 1], 0, 1.0, float(self.data_range), 0)
                                       for i in range(n)])
        return res


class MS_SSIM(nn.Module):
    """Multi-scale SSIM (reference core/loss.py:188-208, calc_msssim :113-160): per-sample values [N] of (img1, img2).
    Runs on csrc/loss_modes.hip; the gradient flows to the SECOND argument (the fused image in every use of the reference)."""

    def __init__(self, win_size=11, data_range=1.0, use_padding=False, size_average=True):
        super(MS_SSIM, self).__init__()
        self.win_size, self.data_range, self.use_padding, self.size_average = win_size, data_range, use_padding, size_average
        self._stock = win_size != 11 or bool(use_padding) or not size_average      # (core/_stock.py: stock torch ops, same results)

    def forward(self, img1, img2):
        if self._stock:
            return _stock.msssim(img1, img2, self.win_size, self.data_range, self.use_padding, self.size_average)
        vals = [1.0 - _ModeLossFn.apply(img2[i:i + 1], img1[i:i + 1], img1[i:i + 1], _SSIM_MODES['ms-ssim'], 1.0, float(self.data_range))
                for i in range(img1.shape[0])]
        return torch.stack(vals)


class MSW_SSIM(nn.Module):
    """Multi-scale (windows 11/9/7/5/3) sigma-weighted SSIM of a fused image against its two sources
    (reference core/loss.py:211-237): a 0-dim tensor, differentiable w.r.t. imgf."""

    def __init__(self, win_sizes=(11, 9, 7, 5, 3), data_range=1.0, use_padding=False, size_average=False):
        super(MSW_SSIM, self).__init__()
        self.win_sizes, self.data_range, self.use_padding, self.size_average = tuple(win_sizes), data_range, use_padding, size_average
        self._stock = tuple(win_sizes) != (11, 9, 7, 5, 3) or bool(use_padding) or bool(size_average)   # (core/_stock.py)

    def forward(self, img1, img2, imgf):
        if self._stock:
            return _stock.mswssim(img1, img2, imgf, self.win_sizes, self.data_range, self.use_padding, self.size_average)
        return 1.0 - _ModeLossFn.apply(imgf, img1, img2, _SSIM_MODES['msw-ssim'], 1.0, float(self.data_range))


class SSIMLoss(nn.Module):
    def __init__(self, mode='ssim', data_range=1.0, use_padding=False, weight=1.0):
        super(SSIMLoss, self).__init__()
        self.mode, self.data_range, self.use_padding, self.weight = mode, data_range, use_padding, weight

    def forward(self, img1, img2, imgf):
        if self.use_padding and (self.mode == 'ssim' or self.mode in _SSIM_MODES):
            # reflect-padded windows (core/loss.py:42-49) are not a HIP kernel: stock torch ops, the reference's results (core/_stock.py)
            return _stock.ssim_loss(self.mode, img1, img2, imgf, self.data_range, True, self.weight)
        if self.mode == 'ssim':
            return _LossFn.apply(imgf, img1, img2, 0, float(self.weight), float(self.data_range), 0)
        if self.mode in _SSIM_MODES:
            return _ModeLossFn.apply(imgf, img1, img2, _SSIM_MODES[self.mode], float(self.weight), float(self.data_range))
        raise ValueError("only supported ['ssim', 'w-ssim', 'ms-ssim', 'msw-ssim'] mode")


def _norm_code(mode):
    if mode == 'l1':
        return 0
    if mode == 'l2':
        return 1
    raise ValueError("only supported ['l1', 'l2'] mode")


class NormLoss(nn.Module):
    """reference core/loss.py:361-385: weight * mean|x| or weight * mean x^2"""

    def __init__(self, mode='l1', weight=1.0):
        super(NormLoss, self).__init__()
        self.mode, self.weight = mode, weight

    def forward(self, x):
        _norm_code(self.mode)
        return self.weight * (torch.abs(x).mean() if self.mode == 'l1' else torch.pow(x, 2).mean())


class TVLoss(nn.Module):
    """reference core/loss.py:347-358: NormLoss(mode, weight) of the vertical plus of the horizontal first differences."""

    def __init__(self, mode='l1', weight=1.0):
        super(TVLoss, self).__init__()
        self.mode, self.weight = mode, weight
        self.loss_fn = NormLoss(mode, weight)

    def forward(self, x):
        return _TVFn.apply(x, _norm_code(self.mode), float(self.weight))


class PixelLoss(nn.Module):
    def __init__(self, mode='l1', weight=1.0):
        super(PixelLoss, self).__init__()
        self.mode, self.weight = mode, weight
        self.loss_fn = NormLoss(mode, weight)

    def forward(self, img1, img2, imgf, mode='avg'):
        if mode not in ('avg', 'max'):
            return None  # the reference has no else branch (core/loss.py:294-304)
        return _LossFn.apply(imgf, img1, img2, 1, float(self.weight), mode == 'max', _norm_code(self.mode))


class GradLoss(nn.Module):
    def __init__(self, mode='l1', weight=1.0):
        super(GradLoss, self).__init__()
        self.mode, self.weight = mode, weight
        self.loss_fn = NormLoss(mode, weight)
        self.register_buffer('x_sobel', torch.FloatTensor([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]]).reshape(1, 1, 3, 3))
        self.register_buffer('y_sobel', torch.FloatTensor([[-1, -2, -1], [0, 0, 0], [1, 2, 1]]).reshape(1, 1, 3, 3))

    def forward(self, img1, img2, imgf, mode='avg'):
        if mode not in ('avg', 'max'):
            return None  # as the reference (core/loss.py:335-344)
        return _LossFn.apply(imgf, img1, img2, 2, float(self.weight), mode == 'max', _norm_code(self.mode))


class _FusionLossFn(torch.autograd.Function):
    """mmif_fusion_loss: {total, ssim, pixel, grad} and d(total)/d(imgf) of the train step's three terms in one call."""

    @staticmethod
    def forward(ctx, imgf, img1, img2, cfg):
        i1, i2, f = _prep(img1, img2, imgf)
        n, _, h, w = f.shape
        need = ctx.needs_input_grad[0]
        vals = torch.empty(5, dtype=torch.float32, device=f.device)
        grad = torch.empty_like(f) if need else None
        key = ("fusion", n, h, w, f.device)
        ws = _ws_cache.get(key)
        if ws is None:
            ws = _ws_cache[key] = torch.empty(lib.mmif_fusion_loss_workspace(n, h, w) // 4 + 1, dtype=torch.float32, device=f.device)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        w_ssim, data_range, w_pixel, pixel_max, pixel_l2, w_grad, grad_max, grad_l2 = cfg
        check(lib.mmif_fusion_loss(p(i1), p(i2), p(f), n, h, w, w_ssim, data_range, w_pixel, pixel_max, pixel_l2, w_grad, grad_max, grad_l2,
                                   p(vals), p(grad), p(ws), ws.numel() * 4, T.stream_ptr()), "fusion_loss")
        ctx.grad = grad
        total, parts = vals[4], vals[:4]
        ctx.mark_non_differentiable(parts)
        ctx.set_materialize_grads(False)   # (no zeros tensor for the parts' gradient slot on every backward)
        return total, parts

    @staticmethod
    def backward(ctx, g, _gparts):
        if ctx.grad is None or g is None:
            return None, None, None, None
        u = _UNIT.get((g.device, g.dtype))
        if u is not None and g.data_ptr() == u.data_ptr():   # total.backward(unit_gradient(total)): d(total)/d(total) = 1, nothing to scale
            return ctx.grad, None, None, None
        return ctx.grad * g, None, None, None


_UNIT = {}


def unit_gradient(total):
    """The root gradient of `total.backward()` -- a tensor of ones of total's shape -- as ONE cached tensor per (device, dtype):
    `total.backward(unit_gradient(total))` is `total.backward()` without the fill kernel autograd launches for its own ones_like, and
    FusionLoss recognises the cached tensor by address and hands its stored d(total)/d(imgf) on without the multiply-by-one pass.  The
    tensor must never be written to."""
    key = (total.device, total.dtype)
    u = _UNIT.get(key)
    if u is None:
        u = _UNIT[key] = torch.ones((), dtype=total.dtype, device=total.device)
    return u if total.dim() == 0 else u.expand(total.shape)


class FusionLoss(nn.Module):
    """loss_fn1(img1, img2, imgf) + loss_fn2(img1, img2, imgf, mode=pixel_mode) + loss_fn3(img1, img2, imgf, mode=grad_mode) of the
    reference's train step (train.py:64-69) as one device call instead of three modules, two additions and autograd's two gradient
    additions.  Takes the three modules the reference builds (train.py:171-173); SSIMLoss must be in mode 'ssim'.  forward returns the
    total (what .backward() is called on); `.values` = the detached device vector [total, l1, l2, l3] of the last call (for logging and
    for the data-parallel scalar reduce)."""

    def __init__(self, loss_fn1, loss_fn2, loss_fn3, pixel_mode='max', grad_mode='max'):
        super().__init__()
        if getattr(loss_fn1, 'mode', 'ssim') != 'ssim':
            raise ValueError("FusionLoss fuses SSIMLoss('ssim') only; use the modules separately for the other SSIM modes")
        for m in (pixel_mode, grad_mode):
            if m not in ('max', 'avg'):
                raise ValueError("only supports 'max' and 'avg' modes.")
        self.fn1, self.fn2, self.fn3 = loss_fn1, loss_fn2, loss_fn3
        self.pixel_mode, self.grad_mode = pixel_mode, grad_mode
        self.values = None

    def forward(self, img1, img2, imgf):
        cfg = (float(self.fn1.weight), float(self.fn1.data_range), float(self.fn2.weight), int(self.pixel_mode == 'max'), _norm_code(self.fn2.mode),
               float(self.fn3.weight), int(self.grad_mode == 'max'), _norm_code(self.fn3.mode))
        total, self.values = _FusionLossFn.apply(imgf, img1, img2, cfg)
        return total
