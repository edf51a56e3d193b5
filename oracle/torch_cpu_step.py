"""TEST / MEASUREMENT INFRASTRUCTURE -- never on the product path.

torch-CPU restatement of the reference's train step (train.py:54-78) for the models the benchmark configurations name -- PFNetv1
(core/model.py:69-111), DenseFuse (:165-186) and, since round 5, NestFuse (:319-363) / RFN-Nest (:366-384) with ConvBlock, RFN,
NestDecoder, nearest Upsample + reflect pad and 2x2 max-pool (core/block.py:708-759, 836-867, 965-991) and the 'sca' attention fusion
(core/fusion.py:42-153) --, written from the maths of SURVEY.md Appendix A with stock torch CPU ops:
reflect-pad + conv2d + ReLU (A.1, core/block.py:98-99), DenseBlock concat (A.2, :147-151), SSIM / max-pixel / Sobel-gradient losses
(A.4 / A.5, core/loss.py:52-110, 287-344), clip_grad_norm_(5) + Adam (A.6, train.py:72-75, 319) through torch.autograd.

The nested nets exist here so that the full-size parity tests (tests/test_gpu_fullsize.py: 1 x 512 x 512 forward + every parameter
gradient) have a CPU reference that finishes in seconds where the numpy oracle needs minutes; they are pinned to the LIVE golden F5 cases
(1 x 32 x 32 and the odd pyramid 2 x 36 x 44) by tests/test_torch_cpu_step.py.

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


NEST_CH = (64, 112, 160, 208)


def _nest_shapes(rfn):
    """state_dict order of NestFuse / RFNNest: NestDecoder's six ConvBlocks, conv_in, the four encoder ConvBlocks, conv_out, then the
    four RFNs -- the order of golden F5's manifest, which tests/test_torch_cpu_step.py compares against (the reference's
    state_dict lists `decode` first: _FusionModel-style attribute order is not registration order there, core/model.py:321-344)"""
    c = NEST_CH
    sh = OrderedDict()

    def block(pre, cin, cout, k1=3, k2=1):   # ConvBlock(cin, cout): cin -> cin // 2 (3x3) -> cout (1x1), core/block.py:708-722
        hid = cin // 2
        sh[f"{pre}.layers.0.layers.0.weight"], sh[f"{pre}.layers.0.layers.0.bias"] = (hid, cin, k1, k1), (hid,)
        sh[f"{pre}.layers.1.layers.0.weight"], sh[f"{pre}.layers.1.layers.0.bias"] = (cout, hid, k2, k2), (cout,)

    def layer(pre, cin, cout, k=3):
        sh[f"{pre}.layers.0.weight"], sh[f"{pre}.layers.0.bias"] = (cout, cin, k, k), (cout,)

    for name, cin, cout in (("DB1_1", c[0] + c[1], c[0]), ("DB2_1", c[1] + c[2], c[1]), ("DB3_1", c[2] + c[3], c[2]),
                            ("DB1_2", 2 * c[0] + c[1], c[0]), ("DB2_2", 2 * c[1] + c[2], c[1]), ("DB1_3", 3 * c[0] + c[1], c[0])):
        block("decode." + name, cin, cout)
    layer("conv_in", 1, 16, 1)
    for i, (cin, cout) in enumerate(zip((16,) + c[:3], c)):
        block(f"CB{i + 1}_0", cin, cout)
    layer("conv_out", c[0], 1, 1)
    if rfn:
        for i, n in enumerate(c):   # RFN(n): core/block.py:737-759
            pre = f"RFN{i + 1}"
            layer(pre + ".res", 2 * n, n)
            layer(pre + ".conv1", n, n)
            layer(pre + ".conv2", n, n)
            layer(pre + ".layers.0", 2 * n, n, 1)
            layer(pre + ".layers.1", n, n)
            layer(pre + ".layers.2", n, n)
    return sh


def _weighted(a, b, w1, w2):
    """weighted_fusion (A.3, core/fusion.py:32-35): w = w1 / max(w1 + w2, 1e-7)"""
    w = w1 / (w1 + w2).clamp(min=1e-7)
    return w * a + (1.0 - w) * b


def _attention_sca(a, b):
    """attention_fusion(mode='sca', spatial 'l1', channel 'avg', no softmax) (core/fusion.py:42-81, 89-90, 123-124)"""
    f_sp = _weighted(a, b, a.abs().sum(dim=1, keepdim=True), b.abs().sum(dim=1, keepdim=True))
    f_ch = _weighted(a, b, a.mean(dim=(2, 3), keepdim=True), b.mean(dim=(2, 3), keepdim=True))
    return (f_sp + f_ch) / 2.0


def _up_to(x, like):
    """Upsample('nearest', 2).forward(feat, shape) (core/block.py:965-991): x2 nearest, then reflect-pad up to the skip's size"""
    y = F.interpolate(x, scale_factor=2, mode="nearest")
    ph, pw = like.shape[-2] - y.shape[-2], like.shape[-1] - y.shape[-1]
    if ph or pw:
        y = F.pad(y, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2), mode="reflect")
    return y


class TorchCpuModel:
    """functional PFNetv1 / DenseFuse / NestFuse / RFNNest on a parameter dict with the reference's state_dict keys"""

    def __init__(self, name):
        assert name in ("PFNetv1", "DenseFuse", "NestFuse", "RFNNest"), name
        self.name = name
        if name == "PFNetv1":
            self.shapes = _decoder_shapes(_pfnet_shapes(("encode1", "encode2")), (128, 128, 64, 32, 16, 1))
        elif name == "DenseFuse":
            self.shapes = _decoder_shapes(_pfnet_shapes(("encode",)), (64, 64, 32, 16, 1))
        else:
            self.shapes = _nest_shapes(name == "RFNNest")

    def init_params(self, seed=0):
        """the reference's init (core/block.py:101-118): kaiming-normal (fan_in, gain sqrt 2) for ReLU layers, torch's Conv2d default
        for the last layer (act=None), zero biases; seeded generator"""
        g = torch.Generator().manual_seed(seed)
        P = OrderedDict()
        # (the nested nets end in conv_out with the default ReLU: every layer of theirs is kaiming-normal)
        last = [k for k in self.shapes if k.endswith("weight")][-1] if self.name in ("PFNetv1", "DenseFuse") else None
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

    # ---- NestFuse / RFN-Nest
    @staticmethod
    def _layer(P, pre, x, relu=True):
        return _conv(x, P[pre + ".layers.0.weight"], P[pre + ".layers.0.bias"], relu)

    @classmethod
    def _block(cls, P, pre, x):
        return cls._layer(P, pre + ".layers.1", cls._layer(P, pre + ".layers.0", x))

    def _nest_encode(self, P, img):
        x = self._block(P, "CB1_0", self._layer(P, "conv_in", img))
        feats = [x]
        for i in (2, 3, 4):
            x = self._block(P, f"CB{i}_0", F.max_pool2d(x, 2, 2))
            feats.append(x)
        return feats

    def _rfn(self, P, pre, a, b):
        res = self._layer(P, pre + ".res", torch.cat((a, b), dim=1))
        x = torch.cat((self._layer(P, pre + ".conv1", a), self._layer(P, pre + ".conv2", b)), dim=1)
        for i in range(3):
            x = self._layer(P, f"{pre}.layers.{i}", x)
        return x + res

    def _nest_forward(self, P, img1, img2):
        e1, e2 = self._nest_encode(P, img1), self._nest_encode(P, img2)
        if self.name == "RFNNest":
            f = [self._rfn(P, f"RFN{i + 1}", a, b) for i, (a, b) in enumerate(zip(e1, e2))]
        else:
            f = [_attention_sca(a, b) for a, b in zip(e1, e2)]
        B = lambda name, *xs: self._block(P, "decode." + name, torch.cat(xs, dim=1))
        x1_1 = B("DB1_1", f[0], _up_to(f[1], f[0]))
        x2_1 = B("DB2_1", f[1], _up_to(f[2], f[1]))
        x3_1 = B("DB3_1", f[2], _up_to(f[3], f[2]))
        x1_2 = B("DB1_2", f[0], x1_1, _up_to(x2_1, x1_1))
        x2_2 = B("DB2_2", f[1], x2_1, _up_to(x3_1, x2_1))
        x1_3 = B("DB1_3", f[0], x1_1, x1_2, _up_to(x2_2, x1_2))
        return self._layer(P, "conv_out", x1_3)      # (default act: ReLU, core/model.py:344)

    def forward(self, P, img1, img2):
        if self.name in ("NestFuse", "RFNNest"):
            return self._nest_forward(P, img1, img2)
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
