# -*- coding: utf-8 -*-
"""Evaluation metric on the test.py path -- API mirror of the reference's core/metric.py:316-364 `calc_ssim`
(the SSIM that test.py:49-52 prints per fused image).  It is the step right after the hot path and shares its
maths with SSIMLoss (11x11 Gaussian window, valid correlation), so it runs on the same fused HIP kernel
(csrc/loss.hip: separable window through LDS, block sums, fixed-order second stage) -- one launch pair per call,
no intermediate maps in HBM.  The reference's other (offline, CPU-side) metrics of eval.py are out of scope.
"""
import ctypes as C

import torch

from mmif import tensor as T
from mmif._lib import check, lib

__all__ = ['calc_ssim']


def calc_ssim(img1, img2, win_size=11, data_range=255.0, use_padding=False, size_average=True, full=False):
    """Mean SSIM of two single-channel image batches [B,1,H,W] (reference default data_range=255; test.py passes 1.0).
    Returns a 0-dim tensor.  The accelerated configuration is the one test.py uses; every other argument combination runs as stock
    torch ops with the reference's results (core/_stock.py)."""
    if img1.shape != img2.shape:
        raise ValueError("img1 and img2 must have the same shape")
    if img1.dim() != 4 or img1.shape[1] != 1:
        # the reference's window is [1,1,k,k] with groups=channel: it also only works for C == 1
        raise RuntimeError(f"calc_ssim takes single-channel images [B,1,H,W]; got {tuple(img1.shape)}")
    n, _, h, w = img1.shape
    if win_size != 11 or min(h, w) < 11 or use_padding or not size_average or full:
        from . import _stock
        return _stock.metric_ssim(img1.float(), img2.float(), win_size, data_range, use_padding, size_average, full)
    T.require_device(img1, "img1")
    T.require_device(img2, "img2")
    a = img1.detach().contiguous().float()
    b = img2.detach().contiguous().float()
    # one evaluation of SSIM(a, b): the per-sample means of mmif_ssim_terms (csrc/loss_modes.hip, the kernel behind core.loss.SSIM);
    # equal-sized samples => mean of the per-sample means = the reference's mean over the whole batch.  (Rounds 1-4 went through the
    # two-source loss kernel with both sources = a, i.e. computed the same map twice.)
    out = torch.empty((3, n), dtype=torch.float32, device=a.device)
    ws = torch.empty(lib.mmif_ssim_loss_mode_workspace(n, h, w, 1) // 4 + 1, dtype=torch.float32, device=a.device)
    p = lambda t: C.c_void_p(t.data_ptr())
    check(lib.mmif_ssim_terms(p(a), p(b), n, h, w, 11, float(data_range), p(out), p(ws), ws.numel() * 4, T.stream_ptr()), "calc_ssim")
    return out[0].mean()
