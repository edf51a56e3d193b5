"""TEST / MEASUREMENT INFRASTRUCTURE -- never on the product path.

torch-CPU restatement of the reference's train step (train.py:54-78) for the two models the benchmark configurations name, PFNetv1
(core/model.py:69-111) and DenseFuse (:165-186), written from the maths of SURVEY.md Appendix A with stock torch CPU ops:
reflect-pad + conv2d + ReLU (A.1, core/block.py:98-99), DenseBlock concat (A.2, :147-151), SSIM / max-pixel / Sobel-gradient losses
(A.4 / A.5, core/loss.py:52-110, 287-344), clip_grad_norm_(5) + Adam (A.6, train.py:72-75, 319) through torch.autograd.

Used by bench.py's `cpu_baseline` leg (the CPU path timed on the GPU box's host cores with the intra-op thread count torch picks,
SURVEY 8(d)(ii)) and pinned against the reference's golden vectors F5 / F6 by tests/test_torch_cpu_step.py.  Only tests/ and
bench.py's cpu_baseline may import this module (the numpy oracle next to it stays the parity checker)."""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F


def _conv(x, w, b, relu=True):
    """ConvLayer: nn.Conv2d(padding = k // 2, padding_mode = 'reflect') + ReLU (core/block.py:56-66, 98-99)"""
    p = w.shape[2] // 2
    y = F.conv2d(F.pad(x, (p, p, p, p), mode="reflect") if p else x, w, b)
    return torch.relu(y) if relu else y


def _pfnet_shapes(prefixes):
    sh = OrderedDict()
    for pre in prefixes:   # ConvLayer(1, 16) + DenseBlock(16, 16): core/model.py:73-80
        sh[f"{pre}.0.layers.0.weight"], sh[f"{pre}.0.layers.0.bias"] = (16, 1, 3, 3), (16,)
        for i in range(3):
            sh[f"{pre}.1.layers.{i}.layers.0.weight"], sh[f"{pre}.1.layers.{i}.layers.0.bias"] = (16, 16 * (i + 1), 3, 3), (16,)
    return sh


def _decoder_shapes(sh, chans):
    for i, (ci, co) in enumerate(zip(chans[:-1], chans[1:])):
        sh[f"decode.{i}.layers.0.weight"], sh[f"decode.{i}.layers.0.bias"] = (co, ci, 3, 3), (co,)
    return sh


class TorchCpuModel:
    """functional PFNetv1 / DenseFuse on a parameter dict with the reference's state_dict keys"""

    def __init__(self, name):
        assert name in ("PFNetv1", "DenseFuse"), name
        self.name = name
        if name == "PFNetv1":
            self.shapes = _decoder_shapes(_pfnet_shapes(("encode1", "encode2")), (128, 128, 64, 32, 16, 1))
        else:
            self.shapes = _decoder_shapes(_pfnet_shapes(("encode",)), (64, 64, 32, 16, 1))

    def init_params(self, seed=0):
        """the reference's init (core/block.py:101-118): kaiming-normal (fan_in, gain sqrt 2) for ReLU layers, torch's Conv2d default
        for the last layer (act=None), zero biases; seeded generator"""
        g = torch.Generator().manual_seed(seed)
        P = OrderedDict()
        last = [k for k in self.shapes if k.endswith("weight")][-1]
        for k, shp in self.shapes.items():
            if k.endswith("bias"):
                P[k] = torch.zeros(shp)
            else:
                fan_in = shp[1] * shp[2] * shp[3]
                if k == last:
                    bound = 1.0 / math.sqrt(fan_in)      # kaiming_uniform(a = sqrt 5)
                    P[k] = (torch.rand(shp, generator=g) * 2 - 1) * bound
                else:
                    P[k] = torch.randn(shp, generator=g) * math.sqrt(2.0 / fan_in)
        for v in P.values():
            v.requires_grad_(True)
        return P

    @staticmethod
    def _encode(P, pre, img):
        x = _conv(img, P[f"{pre}.0.layers.0.weight"], P[f"{pre}.0.layers.0.bias"])
        for i in range(3):   # DenseBlock: x = cat(x, conv_i(x))
            x = torch.cat((x, _conv(x, P[f"{pre}.1.layers.{i}.layers.0.weight"], P[f"{pre}.1.layers.{i}.layers.0.bias"])), dim=1)
        return x

    def forward(self, P, img1, img2):
        if self.name == "PFNetv1":
            x = torch.cat((self._encode(P, "encode1", img1), self._encode(P, "encode2", img2)), dim=1)   # concat_fusion
            n = 5
        else:
            x = self._encode(P, "encode", img1) + self._encode(P, "encode", img2)                          # element_fusion 'sum'
            n = 4
        for i in range(n):
            x = _conv(x, P[f"decode.{i}.layers.0.weight"], P[f"decode.{i}.layers.0.bias"], relu=i < n - 1)
        return x


def _window():
    """core/loss.py:24-39: 1-D taps in Python floats -> fp32, normalised by their fp32 sum, outer product (not renormalised)"""
    g = torch.tensor([math.exp(-(x - 5) ** 2 / (2.0 * 1.5 ** 2)) for x in range(11)], dtype=torch.float32)
    g = g / g.sum()
    return torch.outer(g, g)[None, None]


def _ssim_mean(x, f, win):
    """calc_ssim (A.4), data_range 1, valid correlation, per-sample mean"""
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    mux, muf = F.conv2d(x, win), F.conv2d(f, win)
    sx = (F.conv2d(x * x, win) - mux * mux).clamp(min=0)
    sf = (F.conv2d(f * f, win) - muf * muf).clamp(min=0)
    sxf = F.conv2d(x * f, win) - mux * muf
    S = (2 * mux * muf + C1) * (2 * sxf + C2) / ((mux * mux + muf * muf + C1) * (sx + sf + C2))
    return S.flatten(1).mean(1)


def _sobel(z):
    kx = torch.tensor([[-1., 0., 1.], [-2., 0., 2.], [-1., 0., 1.]])[None, None]
    zp = F.pad(z, (1, 1, 1, 1), mode="reflect")
    return F.conv2d(zp, kx).abs() + F.conv2d(zp, kx.transpose(2, 3).contiguous()).abs()


def fusion_losses(img1, img2, f):
    """train.py:64-69 with the weights of train.py:304-308: SSIMLoss('ssim', 1.0), PixelLoss('l1', 0.01)(mode='max'),
    GradLoss('l1', 0.1)(mode='max').  Returns (l_ssim, l_pixel, l_grad, total)."""
    win = _window()
    l1 = 1.0 * (1.0 - 0.5 * (_ssim_mean(img1, f, win).mean() + _ssim_mean(img2, f, win).mean()))
    l2 = 0.01 * (f - torch.max(img1, img2)).abs().mean()
    l3 = 0.1 * (_sobel(f) - torch.max(_sobel(img1), _sobel(img2))).abs().mean()
    return l1, l2, l3, l1 + l2 + l3


def make_optimizer(P, lr=1e-4):
    return torch.optim.Adam(list(P.values()), lr=lr, betas=(0.9, 0.999), eps=1e-8)


def train_step(model, P, opt, img1, img2, clip=5.0):
    """one iteration of train.py:61-75: zero_grad, forward, three losses, backward, clip_grad_norm_(5), Adam"""
    opt.zero_grad(set_to_none=True)
    f = model.forward(P, img1, img2)
    l1, l2, l3, tot = fusion_losses(img1, img2, f)
    tot.backward()
    norm = torch.nn.utils.clip_grad_norm_(list(P.values()), clip) if clip else None
    grads = OrderedDict((k, v.grad.detach().clone()) for k, v in P.items())   # (clipped in place, as the reference's are)
    opt.step()
    return dict(losses=[float(l1.detach()), float(l2.detach()), float(l3.detach()), float(tot.detach())], grad_norm=float(norm) if norm is not None else None, imgf=f.detach(), grads=grads)
