# -*- coding: utf-8 -*-
"""Feature-fusion functions -- API mirror of the reference's core/fusion.py:21-153.

element_fusion runs on the HIP element-fusion kernels (mmif_fuse_elem_{fwd,bwd}); inside the models
concat_fusion is zero-copy (channel-block views) and DenseFuse's 'sum' runs on blocked buffers.
attention_fusion 'sa' / 'ca' / 'sca' with the reference's default pooling (spatial 'l1', channel 'avg') runs on
the HIP attention kernels (mmif_fuse_attn_{fwd,bwd}); the remaining pooling modes are tensor-level compositions
kept for API completeness.
"""
import torch

from mmif import _lib
from mmif import engine as E
from mmif import tensor as T
from mmif.tensor import BT

__all__ = ['element_fusion', 'weighted_fusion', 'concat_fusion', 'attention_fusion', 'spatial_fusion', 'channel_fusion',
           'spatial_pooling', 'channel_pooling']

eps = 1e-7

_ELEM_MODES = {'sum': _lib.FUSE_SUM, 'mean': _lib.FUSE_MEAN, 'max': _lib.FUSE_MAX}


class _ElemFusionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, mode):
        dtype = E.compute_dtype()
        ab, bb = BT.from_nchw(a.detach(), dtype), BT.from_nchw(b.detach(), dtype)
        ob = BT.alloc(ab.n, a.shape[1], ab.h, ab.w, dtype, a.device)
        T.fuse_elem_fwd(ab, bb, ob, mode)
        ctx.saved = (ab, bb, mode, a.shape[1])
        return ob.to_nchw(a.shape[1])

    @staticmethod
    def backward(ctx, g):
        ab, bb, mode, c = ctx.saved
        gb = BT.from_nchw(g.contiguous(), ab.dtype)
        ga = BT.alloc(ab.n, c, ab.h, ab.w, ab.dtype, g.device)
        gbb = BT.alloc(ab.n, c, ab.h, ab.w, ab.dtype, g.device)
        T.fuse_elem_bwd(ab, bb, gb, ga, gbb, mode, False)
        return ga.to_nchw(c), gbb.to_nchw(c), None


class _AttnFusionFn(torch.autograd.Function):
    """attention_fusion('sa'|'ca'|'sca') with spatial 'l1' / channel 'avg' pooling on the HIP kernels."""

    @staticmethod
    def forward(ctx, a, b, mode):
        dtype = E.compute_dtype()
        ab, bb = BT.from_nchw(a.detach(), dtype), BT.from_nchw(b.detach(), dtype)
        c = a.shape[1]
        ob = BT.alloc(ab.n, c, ab.h, ab.w, dtype, a.device)
        ws = T.attn_workspace(ab.n, c, a.device)
        T.attn_fwd(ab, bb, ob, mode, ws)
        ctx.saved = (ab, bb, mode, c, ws)
        return ob.to_nchw(c)

    @staticmethod
    def backward(ctx, g):
        ab, bb, mode, c, ws = ctx.saved
        gb = BT.from_nchw(g.contiguous(), ab.dtype)
        ga = BT.alloc(ab.n, c, ab.h, ab.w, ab.dtype, g.device)
        gbb = BT.alloc(ab.n, c, ab.h, ab.w, ab.dtype, g.device)
        T.attn_bwd(ab, bb, gb, ga, gbb, mode, False, ws)
        return ga.to_nchw(c), gbb.to_nchw(c), None


def element_fusion(tensor1, tensor2, mode='sum'):
    if mode not in _ELEM_MODES:
        raise ValueError("only supported ['sum', 'mean', 'max'] mode")
    if tensor1.is_cuda and tensor1.dim() == 4 and tensor1.shape == tensor2.shape:
        return _ElemFusionFn.apply(tensor1, tensor2, _ELEM_MODES[mode])
    T.require_device(tensor1, "element_fusion input")
    raise ValueError("element_fusion expects two [B,C,H,W] tensors of equal shape")


def weighted_fusion(tensor1, tensor2, w1, w2):
    w = w1 / (w1 + w2).clamp_(min=eps)
    return w * tensor1 + (1.0 - w) * tensor2


def concat_fusion(tensors, dim=1):
    return torch.cat(tensors, dim)


def _nonlocal(t, spatial):
    """Res2Fusion's non-local attention maps (reference core/fusion.py:96-113 spatial, :137-150 channel): softmax over a min-max
    normalised energy, applied to the features, plus the identity.  Tensor-level composition (two batched matmuls)."""
    b, c, h, w = t.shape
    flat = t.reshape(b, c, -1)
    if spatial:   # queries: every pixel; keys / values: the 8x8 average-pooled map
        pooled = torch.nn.functional.avg_pool2d(t, 8, 8).reshape(b, c, -1)
        q, k, v = flat.permute(0, 2, 1), pooled, pooled.permute(0, 2, 1)
    else:         # channel x channel
        q, k, v = flat, flat.permute(0, 2, 1), flat
    energy = q @ k
    lo, hi = torch.min(energy), torch.max(energy)
    attn = torch.softmax((energy - lo) / (hi - lo), dim=-1) @ v
    return (attn.permute(0, 2, 1) if spatial else attn).reshape(b, c, h, w) + t


def spatial_pooling(tensor, mode='l1'):
    if mode == 'sum':
        return tensor.sum(dim=1, keepdim=True)
    if mode == 'mean':
        return tensor.mean(dim=1, keepdim=True)
    if mode == 'l1':
        return tensor.norm(p=1, dim=1, keepdim=True)
    if mode == 'l2':
        return tensor.norm(p=2, dim=1, keepdim=True)
    if mode == 'linf':
        return tensor.max(dim=1, keepdim=True)[0]
    if mode == 'nl':
        return _nonlocal(tensor, spatial=True)
    raise ValueError("only supported ['sum', 'mean', 'l1', 'l2', 'linf', 'nl'] mode")


def channel_pooling(tensor, mode='avg'):
    if mode == 'avg':
        return tensor.mean(dim=(2, 3), keepdim=True)
    if mode == 'max':
        return tensor.amax(dim=(2, 3), keepdim=True)
    if mode == 'nl':
        return _nonlocal(tensor, spatial=False)
    if mode == 'nuclear':
        from ._stock import nuclear_pooling
        return nuclear_pooling(tensor)
    raise ValueError("only supported ['avg', 'max', 'nuclear', 'nl'] mode")


def spatial_fusion(tensor1, tensor2, mode='l1', softmax=True):
    s1, s2 = spatial_pooling(tensor1, mode), spatial_pooling(tensor2, mode)
    if softmax:
        s1, s2 = torch.exp(s1), torch.exp(s2)
    return weighted_fusion(tensor1, tensor2, s1, s2)


def channel_fusion(tensor1, tensor2, mode='avg', softmax=True):
    c1, c2 = channel_pooling(tensor1, mode), channel_pooling(tensor2, mode)
    if softmax:
        c1, c2 = torch.exp(c1), torch.exp(c2)
    return weighted_fusion(tensor1, tensor2, c1, c2)


def attention_fusion(tensor1, tensor2, mode='sca', spatial_mode='l1', channel_mode='avg'):
    if mode not in ('sa', 'ca', 'sca', 'wavg'):
        raise ValueError("only supported ['sa', 'ca', 'sca', 'wavg'] mode")
    if (tensor1.is_cuda and mode in T.ATTN_MODES and spatial_mode == 'l1' and channel_mode == 'avg' and tensor1.dim() == 4
            and tensor1.shape == tensor2.shape):
        return _AttnFusionFn.apply(tensor1, tensor2, T.ATTN_MODES[mode])
    f_spatial = spatial_fusion(tensor1, tensor2, spatial_mode, softmax=False)
    f_channel = channel_fusion(tensor1, tensor2, channel_mode, softmax=False)
    if mode == 'sa':
        return f_spatial
    if mode == 'ca':
        return f_channel
    if mode == 'sca':
        return (f_spatial + f_channel) / 2.0
    return weighted_fusion(f_spatial, f_channel, f_spatial, f_channel)
