# -*- coding: utf-8 -*-
"""Network blocks -- API mirror of the reference's core/block.py for the hot-path blocks
(ConvLayer :26-118, DenseBlock :137-151, ConvBlock :708-722, RFN :737-759, NestDecoder :836-867,
Upsample/Downsample :941-991), executed by hand-written HIP kernels.

Called standalone (NCHW tensors in / out) every block converts at its boundary; inside the models
(core/model.py) the whole network runs on blocked-NHWC buffers without these conversions.
"""
import torch
import torch.nn as nn

from mmif import _lib
from mmif import dist as D
from mmif import engine as E
from mmif import tensor as T
from mmif.tensor import BT

from .fusion import concat_fusion, element_fusion

__all__ = ['ConvLayer', 'ResBlock', 'DenseBlock', 'SepConvBlock', 'Res2ConvBlock', 'ConvBlock', 'ECB', 'DCB', 'RFN', 'NestEncoder', 'NestDecoder', 'FSDecoder', 'Downsample',
           'Upsample', 'MaxPool2d']


class _ConvLayerFn(torch.autograd.Function):
    """One ConvLayer on NCHW boundary tensors: reflect-pad conv + bias + ReLU through the C ABI."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        T.require_device(x, "ConvLayer input")
        cout, cin, k, _ = weight.shape
        dtype, impl = E.compute_dtype(), E.conv_impl()
        n, _, h, w = x.shape
        dev = x.device
        wd, bd = weight.detach(), bias.detach() if bias is not None else None
        packed = None
        if E.wants_packed(dtype, impl, standalone=True) and cin > 1 and cout > 1:
            packed = T.PackedWeights(cout, cin, k, dev, _lib.BF16 if dtype == torch.bfloat16 else _lib.F32)
            packed.pack(wd)
        if cin == 1:
            xin = x.detach().contiguous().float()
            yb = BT.alloc(n, cout, h, w, dtype, dev)
            T.image_in_fwd(xin, wd, bd, yb, cout, k, relu)
            y = yb.to_nchw(cout)
            ctx.saved = ("in", xin, yb)
        elif cout == 1:
            xb = BT.from_nchw(x.detach(), dtype)
            y = torch.empty((n, 1, h, w), dtype=torch.float32, device=dev)
            T.image_out_fwd(xb, wd, bd, y, cin, k, relu)
            ctx.saved = ("out", xb, y if relu else None)
        else:
            xb = BT.from_nchw(x.detach(), dtype)
            yb = BT.alloc(n, cout, h, w, dtype, dev)
            T.conv_fwd(xb, wd, bd, yb, cin, cout, k, relu, packed, impl)
            y = yb.to_nchw(cout)
            ctx.saved = ("mid", xb, yb)
        ctx.meta = (cin, cout, k, relu, dtype, impl, packed, wd)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        cin, cout, k, relu, dtype, impl, packed, wd = ctx.meta
        kind, a, b = ctx.saved
        gy = gy.contiguous().float()
        n, _, h, w = gy.shape
        dev = gy.device
        dw = torch.empty_like(wd)
        db = torch.empty(cout, dtype=torch.float32, device=dev)
        gx = None
        if kind == "out":
            xb, yimg = a, b
            ws = torch.empty(T.image_wgrad_workspace_bytes(cin, k) // 4 + 1, dtype=torch.float32, device=dev)
            T.image_out_wgrad(xb, gy, yimg, dw, db, cin, k, ws)
            if ctx.needs_input_grad[0]:
                gxb = BT.alloc(n, cin, h, w, dtype, dev, halo=1)
                T.image_out_dgrad(gy, yimg, wd, None, gxb, cin, k, 0, 0)
                gx = gxb.to_nchw(cin)
        else:
            gb = BT.from_nchw(gy, dtype)
            yb = b
            if relu:  # g * [y > 0]
                gm = BT.alloc(n, cout, h, w, dtype, dev)
                scratch = BT.alloc(n, cout, h, w, dtype, dev)
                T.fuse_elem_bwd(yb, yb, gb, gm, scratch, _lib.FUSE_SUM, True)
                gb = gm
            if kind == "in":
                ws = torch.empty(T.image_wgrad_workspace_bytes(cout, k) // 4 + 1, dtype=torch.float32, device=dev)
                T.image_in_wgrad(a, gb, dw, db, cout, k, ws)
                if ctx.needs_input_grad[0]:
                    raise NotImplementedError("mmif: gradient w.r.t. a single-channel input image is not provided")
            else:
                xb = a
                ws = torch.empty(T.wgrad_workspace_bytes(cin, cout, k) // 4 + 1, dtype=torch.float32, device=dev)
                # (fp32 without an x3 image = this layer stays on the exact fp32 FMA kernels: its weight gradient too)
                wimpl = _lib.IMPL_VALU if (dtype == torch.float32 and packed is None) else impl
                T.conv_wgrad(xb, gb, dw, db, cin, cout, k, ws, False, wimpl)
                if ctx.needs_input_grad[0]:
                    gxb = BT.alloc(n, cin, h, w, dtype, dev, halo=1)
                    T.conv_dgrad(gb, wd, None, gxb, cin, cout, k, 0, 0, packed, impl)
                    gx = gxb.to_nchw(cin)
        return gx, dw, (db if ctx.has_bias else None), None


class _GConvFn(torch.autograd.Function):
    """nn.Conv2d with k in 1/3/5/7, stride 1/2, reflect | zero padding (+ ReLU) on the general HIP kernels (csrc/conv_general.hip):
    the layers of DeepFuse (k = 5/7), DBNet / NestFuse down_mode='stride' (stride 2) -- NCHW fp32 boundary tensors."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, reflect, relu):
        T.require_device(x, "ConvLayer input")
        xd, wd = x.detach().contiguous().float(), weight.detach().contiguous().float()
        bd = bias.detach().contiguous().float() if bias is not None else None
        y = T.gconv_fwd(xd, wd, bd, stride, padding, reflect, relu)
        ctx.saved = (xd, wd, y if relu else None)
        ctx.meta = (stride, padding, reflect, relu, bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        xd, wd, y = ctx.saved
        stride, padding, reflect, relu, has_bias = ctx.meta
        gy = gy.contiguous().float()
        if relu:
            gy = T.relu_bwd(gy, y)
        dw, db = T.gconv_wgrad(xd, gy, wd.shape[2], stride, padding, reflect, has_bias)
        dx = T.gconv_dgrad(gy, wd, tuple(xd.shape), stride, padding, reflect) if ctx.needs_input_grad[0] else None
        return dx, dw, db, None, None, None, None


class _GConvTFn(torch.autograd.Function):
    """nn.ConvTranspose2d(k, stride, padding, output_padding) (+ ReLU) on the general HIP kernels (reference core/block.py:67-76)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, output_padding, relu):
        T.require_device(x, "ConvLayer input")
        xd, wd = x.detach().contiguous().float(), weight.detach().contiguous().float()
        bd = bias.detach().contiguous().float() if bias is not None else None
        y = T.gconvt_fwd(xd, wd, bd, stride, padding, output_padding, relu)
        ctx.saved = (xd, wd, y if relu else None)
        ctx.meta = (stride, padding, output_padding, relu, bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        xd, wd, y = ctx.saved
        stride, padding, output_padding, relu, has_bias = ctx.meta
        gy = gy.contiguous().float()
        if relu:
            gy = T.relu_bwd(gy, y)
        dw = T.gconvt_wgrad(xd, gy, wd.shape[2], stride, padding, output_padding)
        db = T.channel_sum(gy) if has_bias else None
        dx = T.gconvt_dgrad(gy, wd, tuple(xd.shape), stride, padding, output_padding) if ctx.needs_input_grad[0] else None
        return dx, dw, db, None, None, None, None


_ACT_CODE = {None: T.ACT_NONE, nn.ReLU: T.ACT_RELU, nn.LeakyReLU: T.ACT_LEAKY, nn.Tanh: T.ACT_TANH, nn.ReLU6: T.ACT_RELU6}


class _DwConvFn(torch.autograd.Function):
    """depth-wise nn.Conv2d (groups == channels, k in {1, 3}, stride 1) on csrc/conv_general.hip -- Res2ConvBlock.dwconvs"""

    @staticmethod
    def forward(ctx, x, weight, bias, reflect):
        T.require_device(x, "ConvLayer input")
        xd, wd = x.detach().contiguous().float(), weight.detach().contiguous().float()
        ctx.saved, ctx.meta = (xd, wd), (reflect, bias is not None)
        return T.dwconv_fwd(xd, wd, bias.detach().contiguous().float() if bias is not None else None, reflect)

    @staticmethod
    def backward(ctx, gy):
        xd, wd = ctx.saved
        reflect, has_bias = ctx.meta
        gy = gy.contiguous().float()
        dw, db = T.dwconv_wgrad(xd, gy, wd.shape[2], reflect, has_bias)
        dx = T.dwconv_dgrad(gy, wd, reflect) if ctx.needs_input_grad[0] else None
        return dx, dw, db, None


class _NormActFn(torch.autograd.Function):
    """norm + activation epilogue of a ConvLayer (reference core/block.py:78-92) on the HIP kernels (csrc/norm.hip):
    nn.BatchNorm2d (batch statistics + running-buffer update when training, running buffers in eval) or nn.GroupNorm(c, c),
    then ReLU / LeakyReLU(0.2) / Tanh / nothing.  The nn module only owns the parameters and buffers."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mod, act):
        T.require_device(x, "ConvLayer input")
        xd = x.detach().contiguous().float()
        if isinstance(mod, nn.BatchNorm2d):
            kind = T.NORM_BN_TRAIN if (mod.training or not mod.track_running_stats) else T.NORM_BN_EVAL
            rm, rv = (mod.running_mean, mod.running_var) if mod.track_running_stats else (None, None)
            momentum = mod.momentum if mod.momentum is not None else 0.1
            if kind == T.NORM_BN_TRAIN and mod.track_running_stats:
                mod.num_batches_tracked.add_(1)
        else:
            kind, rm, rv, momentum = T.NORM_GN, None, None, 0.0
        g = gamma.detach() if gamma is not None else None
        b = beta.detach() if beta is not None else None
        count = None
        if kind == T.NORM_BN_TRAIN and D.sync_bn_active():
            # nn.SyncBatchNorm semantics (reference train.py:296): statistics over the GLOBAL batch -- per-channel (sum, sum^2, count)
            # in fp64, one small all-reduce, then normalise; the count stays on the device for the backward pass
            chan = D.allreduce_sums(T.bn_moments(xd))
            y, stats = T.bn_apply_fwd(xd, chan, g, b, rm, rv, mod.eps, momentum, act)
            count = chan[-1:]
        else:
            y, stats = T.norm_act_fwd(xd, g, b, rm, rv, kind, mod.eps, momentum, act)
        ctx.saved = (xd, y, stats, g, count)
        ctx.meta = (kind, act, gamma is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        xd, y, stats, gamma, count = ctx.saved
        kind, act, affine = ctx.meta
        gy = gy.contiguous().float()
        if count is not None:
            # dgamma / dbeta stay rank-local (they are summed by the gradient all-reduce like every parameter gradient);
            # dx needs the global (sum dz, sum dz xhat)
            chan, dg, db = T.bn_bwd_sums(xd, y, gy, stats, act, want_affine=affine)
            dx = T.bn_apply_bwd(xd, y, gy, stats, gamma, D.allreduce_sums(chan), count, act)
            return dx, dg, db, None, None
        dx, dg, db = T.norm_act_bwd(xd, y, gy, stats, gamma, kind, act, want_affine=affine)
        return dx, dg, db, None, None


class _ActFn(torch.autograd.Function):
    """LeakyReLU(0.2) / Tanh after a conv without norm (PMGI's decode layer, reference core/model.py:579)."""

    @staticmethod
    def forward(ctx, x, act):
        T.require_device(x, "ConvLayer input")
        y = T.act_fwd(x.detach().contiguous().float(), act)
        ctx.saved, ctx.act = y, act
        return y

    @staticmethod
    def backward(ctx, gy):
        return T.act_bwd(gy.contiguous().float(), ctx.saved, ctx.act), None


class ConvLayer(nn.Module):
    """reference core/block.py:26-118 -- same constructor signature, same sub-module layout
    (`layers.0` = the nn.Conv2d that owns weight/bias, so state_dict keys are identical), same
    initialisation (kaiming-normal for ReLU-family acts, zero bias, torch default otherwise)."""

    def __init__(self, in_ch, out_ch, ksize=3, stride=1, padding=None, dilation=1, groups=1, bias=None, norm=None,
                 pre_norm=None, layer=nn.Conv2d, act=nn.ReLU, padding_mode='reflect'):
        super(ConvLayer, self).__init__()
        if padding is None:
            padding = ksize // 2
        if bias is None:
            bias = (norm is not nn.BatchNorm2d) or (pre_norm is not nn.BatchNorm2d)
        mods = []
        if pre_norm is not None:
            mods.append(pre_norm(out_ch, out_ch) if pre_norm is nn.GroupNorm else pre_norm(out_ch))
        if layer is nn.Conv2d:
            mods.append(nn.Conv2d(in_ch, out_ch, ksize, stride, padding, dilation, groups, bias=bias, padding_mode=padding_mode))
        elif layer is nn.ConvTranspose2d:
            mods.append(nn.ConvTranspose2d(in_ch, out_ch, ksize, stride, padding, output_padding=1, bias=bias, padding_mode='zeros'))
        if norm is not None:
            mods.append(norm(out_ch, out_ch) if norm is nn.GroupNorm else norm(out_ch))
        if act is not None:
            if act in (nn.ReLU, nn.ReLU6, nn.Hardswish):
                mods.append(act(inplace=True))
            elif act is nn.LeakyReLU:
                mods.append(act(0.2, inplace=True))
            else:
                mods.append(act())
        self.layers = nn.Sequential(*mods)
        self.norm, self.pre_norm, self.act = norm, pre_norm, act
        # the HIP kernels cover exactly what the hot-path models use
        # what follows the conv: nothing / ReLU fused into the conv kernels, or a norm (+ act) / other activation epilogue kernel
        depthwise = groups > 1 and groups == in_ch == out_ch and layer is nn.Conv2d and stride == 1 and ksize in (1, 3) and padding == ksize // 2
        epi_ok = (pre_norm is None and norm in (None, nn.BatchNorm2d, nn.GroupNorm) and act in _ACT_CODE and dilation == 1
                  and (groups == 1 or depthwise))
        self._epilogue = epi_ok and (norm is not None or act in (nn.LeakyReLU, nn.Tanh, nn.ReLU6) or depthwise)
        self._depthwise = depthwise
        plain = epi_ok and not self._epilogue
        geom_hot = (layer is nn.Conv2d and stride == 1 and ksize in (1, 3) and padding == ksize // 2 and not depthwise
                    and (padding_mode == 'reflect' or ksize == 1) and bias
                    and max(in_ch, out_ch) <= 512)   # (a blocked-layout view holds at most 64 channel blocks: wider layers -- UNFusion's
                                                     # 1024-channel level, MAFusion's 960-channel concats -- take the general kernels)
        # the general kernels (csrc/conv_general.hip): k = 5 / 7, stride 2, zero padding, ConvTranspose2d -- fp32 NCHW
        geom_gen = (ksize in (1, 3, 5, 7) and stride in (1, 2) and 0 <= padding <= ksize // 2
                    and ((layer is nn.Conv2d and padding_mode in ('reflect', 'zeros')) or layer is nn.ConvTranspose2d))
        self._hip = plain and geom_hot                       # hot-path kernels, ReLU fused
        self._gen = plain and not geom_hot and geom_gen      # general kernels, ReLU fused
        self._conv_hot = geom_hot                            # (which conv kernels an epilogue layer uses)
        self._epilogue = self._epilogue and (geom_hot or geom_gen or depthwise)
        self._geom = (stride, padding, padding_mode == 'reflect' and padding > 0)
        self._init_weights()

    def _conv(self, x, relu):
        conv = self.layers[0]
        if self._depthwise:
            return _DwConvFn.apply(x, conv.weight, conv.bias, self._geom[2])
        if self._conv_hot:
            return _ConvLayerFn.apply(x, conv.weight, conv.bias, relu)
        stride, padding, reflect = self._geom
        if isinstance(conv, nn.ConvTranspose2d):
            return _GConvTFn.apply(x, conv.weight, conv.bias, stride, padding, 1, relu)
        return _GConvFn.apply(x, conv.weight, conv.bias, stride, padding, reflect, relu)

    def forward(self, x):
        if self._hip or self._gen:
            return self._conv(x, self.act is not None)
        if self._epilogue:
            z = self._conv(x, False)
            if self.norm is None:
                return z if self.act is None else _ActFn.apply(z, _ACT_CODE[self.act])
            mod = self.layers[1]
            return _NormActFn.apply(z, mod.weight, mod.bias, mod, _ACT_CODE[self.act])
        # argument combinations outside these (pre_norm, other norms / activations, dilation, groups)
        # are not re-implemented: they run as the stock torch modules they are
        return self.layers(x)

    def _init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                if self.act in (nn.ReLU, nn.ReLU6, nn.Hardswish, nn.SiLU, nn.GELU):
                    nn.init.kaiming_normal_(m.weight)
                elif self.act is nn.LeakyReLU:
                    nn.init.kaiming_normal_(m.weight, a=0.2)
                elif self.act is nn.Tanh:
                    nn.init.xavier_normal_(m.weight, gain=nn.init.calculate_gain('tanh'))
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)


class ResBlock(nn.Module):
    """reference core/block.py:121-134 (SEDRFuse, DIFNet): two ConvLayers with optional norms, identity shortcut."""

    def __init__(self, in_ch, out_ch, norm1=None, norm2=None):
        super(ResBlock, self).__init__()
        self.in_ch, self.out_ch = in_ch, out_ch
        self.layers = nn.Sequential(ConvLayer(in_ch, out_ch, norm=norm1), ConvLayer(out_ch, out_ch, norm=norm2, act=None))

    def forward(self, x):
        return element_fusion(self.layers(x), x, 'sum')


class SepConvBlock(nn.Module):
    """reference core/block.py:153-226: point-wise expand (x scale) -> depth-wise k x k -> point-wise project, optional point-wise
    attention gate and (projected) residual, activation at the end.  Base of Res2ConvBlock."""

    def __init__(self, in_ch, out_ch, scale=4, ksize=3, bias=False, norm=None, act=nn.ReLU6, residual=True, attention=False):
        super(SepConvBlock, self).__init__()
        self.norm, self.act, self.residual, self.attention = norm, act(), residual, attention
        self._act_code = _ACT_CODE.get(act)
        self.in_ch, self.out_ch, self.scale = in_ch, out_ch, scale
        hid_ch = in_ch * scale
        self.pwconv1 = ConvLayer(in_ch, hid_ch, ksize=1, bias=bias, norm=norm, act=act)
        self.dwconv = ConvLayer(hid_ch, hid_ch, ksize=ksize, groups=hid_ch, bias=bias, norm=norm, act=None)
        self.pwconv2 = ConvLayer(hid_ch, out_ch, ksize=1, bias=bias, norm=norm, act=None)
        if attention:
            self.pwconv = ConvLayer(in_ch, hid_ch, ksize=1, bias=bias, norm=norm, act=act)
        if residual:
            self.shortcut = ConvLayer(in_ch, out_ch, ksize=1, bias=bias, norm=norm, act=None) if in_ch != out_ch else nn.Identity()

    def _mix(self, x):
        return self.dwconv(self.pwconv1(x))

    def _finish(self, out):
        return _ActFn.apply(out, self._act_code) if (self._act_code is not None and out.is_cuda) else self.act(out)

    def forward(self, x):
        out = self._mix(x)
        if self.attention:
            out = out * self.pwconv(x)
        out = self.pwconv2(out)
        if self.residual:
            out = element_fusion(out, self.shortcut(x), 'sum')
        return self._finish(out)


class Res2ConvBlock(SepConvBlock):
    """reference core/block.py:286-350 (Res2Fusion): the expanded features are split into `scale` groups of in_ch channels that go
    through depth-wise convs hierarchically -- group i adds the previous group's output (from the third group on) before its own
    depth-wise conv (1x1 for the first group, 3x3 after) -- and are concatenated again."""

    def __init__(self, in_ch, out_ch, scale=4, bias=False, norm=None, act=nn.ReLU6, residual=True, attention=False):
        super(Res2ConvBlock, self).__init__(in_ch, out_ch, scale=scale, bias=bias, norm=norm, act=act, residual=residual, attention=attention)
        width = in_ch
        self.dwconvs = nn.ModuleList([ConvLayer(width, width, ksize=3 if i > 0 else 1, groups=width, bias=bias, norm=norm, act=None)
                                      for i in range(scale)])

    def _mix(self, x):
        xs = torch.chunk(self.pwconv1(x), self.scale, dim=1)
        if self.scale == 1:
            return self.dwconvs[0](xs[0])
        outs, y = [], None
        for i in range(self.scale):
            y = element_fusion(y, xs[i].contiguous(), 'sum') if i > 1 else xs[i]   # (reference: `y + xs[i] if i > 1 else xs[i]`)
            y = self.dwconvs[i](y)
            outs.append(y)
        return concat_fusion(outs)


class DenseBlock(nn.Module):
    """reference core/block.py:137-151"""

    def __init__(self, in_ch, out_ch, num_convs=3):
        super(DenseBlock, self).__init__()
        self.in_ch, self.out_ch = in_ch, out_ch
        self.layers = nn.ModuleList([ConvLayer(in_ch + i * out_ch, out_ch) for i in range(num_convs)])

    def forward(self, x):
        for layer in self.layers:
            x = concat_fusion((x, layer(x)))
        return x


class ConvBlock(nn.Module):
    """reference core/block.py:708-722"""

    def __init__(self, in_ch, out_ch, ksize1=3, ksize2=1):
        super(ConvBlock, self).__init__()
        self.in_ch, self.out_ch = in_ch, out_ch
        hid_ch = in_ch // 2
        self.layers = nn.Sequential(ConvLayer(in_ch, hid_ch, ksize=ksize1), ConvLayer(hid_ch, out_ch, ksize=ksize2))

    def forward(self, x):
        return self.layers(x)


class ECB(ConvBlock):
    """UNFusion's encoder block (reference core/block.py:725-728): 1x1 then 3x3"""

    def __init__(self, in_ch, out_ch, ksize1=1, ksize2=3):
        super(ECB, self).__init__(in_ch, out_ch, ksize1=ksize1, ksize2=ksize2)


class DCB(ConvBlock):
    """UNFusion's decoder block (reference core/block.py:731-734): 3x3 then 3x3"""

    def __init__(self, in_ch, out_ch, ksize1=3, ksize2=3):
        super(DCB, self).__init__(in_ch, out_ch, ksize1=ksize1, ksize2=ksize2)


class RFN(nn.Module):
    """reference core/block.py:737-759"""

    def __init__(self, num_ch):
        super(RFN, self).__init__()
        self.res = ConvLayer(num_ch * 2, num_ch)
        self.conv1 = ConvLayer(num_ch, num_ch)
        self.conv2 = ConvLayer(num_ch, num_ch)
        self.layers = nn.Sequential(ConvLayer(num_ch * 2, num_ch, ksize=1), ConvLayer(num_ch, num_ch), ConvLayer(num_ch, num_ch))

    def forward(self, x1, x2):
        f_res = self.res(concat_fusion((x1, x2)))
        f_out = self.layers(concat_fusion((self.conv1(x1), self.conv2(x2))))
        return f_out + f_res


class _MaxPoolFn(torch.autograd.Function):
    """nn.MaxPool2d(k, k) on csrc/resample.hip (first-maximum tie rule in the backward, like ATen)"""

    @staticmethod
    def forward(ctx, x, k):
        y, idx = T.maxpool_nchw_fwd(x.detach().contiguous().float(), k)
        ctx.saved, ctx.meta = idx, ((x.shape[2], x.shape[3]), k)
        return y

    @staticmethod
    def backward(ctx, g):
        return T.maxpool_nchw_bwd(g.contiguous().float(), ctx.saved, *ctx.meta), None


class MaxPool2d(nn.MaxPool2d):
    """nn.MaxPool2d whose forward runs on the HIP kernel for the configurations the models use (kernel == stride, no padding /
    dilation, floor mode, 4-D GPU input); anything else is the stock module."""

    def forward(self, x):
        k = self.kernel_size if isinstance(self.kernel_size, int) else None
        if (x.is_cuda and x.dim() == 4 and k is not None and self.stride == k and self.padding == 0 and self.dilation == 1
                and not self.ceil_mode and not self.return_indices and x.shape[2] >= k and x.shape[3] >= k):
            return _MaxPoolFn.apply(x, k)
        return super(MaxPool2d, self).forward(x)


class _NearestUpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return T.nearest_up_fwd(x.detach().contiguous().float(), scale)

    @staticmethod
    def backward(ctx, g):
        return T.nearest_up_bwd(g.contiguous().float(), ctx.scale), None


class _ReflectPadFn(torch.autograd.Function):
    """nn.ReflectionPad2d((left, right, top, bottom)); negative amounts crop"""

    @staticmethod
    def forward(ctx, x, pads):
        ctx.meta = ((x.shape[2], x.shape[3]), pads)
        return T.reflect_pad_fwd(x.detach().contiguous().float(), pads)

    @staticmethod
    def backward(ctx, g):
        return T.reflect_pad_bwd(g.contiguous().float(), *ctx.meta), None


class _Resample(nn.Module):
    """shape-matching tail shared by Upsample / Downsample (reference core/block.py:953-961,981-991)"""

    @staticmethod
    def _pad(feat, shape):
        pad_h, pad_w = shape[-2] - feat.shape[-2], shape[-1] - feat.shape[-1]
        top, left = pad_h // 2, pad_w // 2
        pads = (left, pad_w - left, top, pad_h - top)
        if feat.is_cuda and max(pads[0], pads[1]) < feat.shape[-1] and max(pads[2], pads[3]) < feat.shape[-2]:
            return _ReflectPadFn.apply(feat, pads)
        return nn.ReflectionPad2d(pads)(feat)

    def forward(self, feat, shape):
        out = self.op(feat)
        if out.shape != shape:
            out = self._pad(out, shape)
        return out


class Downsample(_Resample):
    """reference core/block.py:941-962"""

    def __init__(self, kernel_size=2, stride=2):
        super(Downsample, self).__init__()
        self.down = MaxPool2d(kernel_size, stride)
        self.op = self.down


class _BilinearUpFn(torch.autograd.Function):
    """nn.Upsample(scale_factor, mode='bilinear', align_corners=True) on the HIP kernels (csrc/resample.hip)."""

    @staticmethod
    def forward(ctx, x, scale):
        T.require_device(x, "Upsample input")
        ctx.in_hw = (x.shape[2], x.shape[3])
        return T.bilinear_up_fwd(x.detach().contiguous().float(), scale)

    @staticmethod
    def backward(ctx, g):
        return T.bilinear_up_bwd(g.contiguous().float(), ctx.in_hw), None


class Upsample(_Resample):
    """reference core/block.py:965-991"""

    def __init__(self, mode, scale_factor=2):
        super(Upsample, self).__init__()
        if mode == 'nearest':
            self.up = nn.Upsample(scale_factor=scale_factor, mode=mode)
        else:
            self.up = nn.Upsample(scale_factor=scale_factor, mode=mode, align_corners=True)
        whole = int(scale_factor) == scale_factor
        self.op = self._bilinear_op if (mode == 'bilinear' and whole) else (self._nearest_op if (mode == 'nearest' and whole) else self.up)

    def _bilinear_op(self, feat):
        return _BilinearUpFn.apply(feat, int(self.up.scale_factor))

    def _nearest_op(self, feat):
        return _NearestUpFn.apply(feat, int(self.up.scale_factor)) if feat.is_cuda else self.up(feat)


class NestDecoder(nn.Module):
    """reference core/block.py:836-867 (UNet++ nested decoder)"""

    def __init__(self, block, num_ch, up_mode='bilinear'):
        super(NestDecoder, self).__init__()
        c = num_ch
        self.DB1_1 = block(c[0] + c[1], c[0])
        self.DB2_1 = block(c[1] + c[2], c[1])
        self.DB3_1 = block(c[2] + c[3], c[2])
        self.DB1_2 = block(c[0] * 2 + c[1], c[0])
        self.DB2_2 = block(c[1] * 2 + c[2], c[1])
        self.DB1_3 = block(c[0] * 3 + c[1], c[0])
        self.up = Upsample(up_mode, 2)

    def forward(self, feats):
        f, up = feats, self.up
        x1_1 = self.DB1_1(concat_fusion((f[0], up(f[1], f[0].shape))))
        x2_1 = self.DB2_1(concat_fusion((f[1], up(f[2], f[1].shape))))
        x3_1 = self.DB3_1(concat_fusion((f[2], up(f[3], f[2].shape))))
        x1_2 = self.DB1_2(concat_fusion((f[0], x1_1, up(x2_1, x1_1.shape))))
        x2_2 = self.DB2_2(concat_fusion((f[1], x2_1, up(x3_1, x2_1.shape))))
        return self.DB1_3(concat_fusion((f[0], x1_1, x1_2, up(x2_2, x1_2.shape))))


class NestEncoder(nn.Module):
    """UNFusion's densely nested encoder (reference core/block.py:762-797): every level re-reads the levels above it through
    stride-2 ConvLayers (or max-pooling).  feats = (x1, (x2, d1), (x3, d2), (x4, d3)) with d_i the down-sampled level above."""

    def __init__(self, block, in_ch, out_ch, down_mode='stride'):
        super(NestEncoder, self).__init__()
        i, o = in_ch, out_ch
        self.EB2_1 = block(i[1] + i[0], o[1])
        self.EB3_1 = block(i[2] + i[1], i[2] * 2)
        self.EB4_1 = block(i[3] + i[2], i[3] * 2)
        self.EB3_2 = block(i[2] * 3 + o[1], o[2])
        self.EB4_2 = block(i[3] * 3 + i[2] * 2, i[3] * 4 + i[2])
        self.EB4_3 = block(i[3] * 7 + i[2] + o[2], o[3])
        if down_mode == 'maxpool':
            self.down1, self.down2, self.down3 = MaxPool2d(2, 2), MaxPool2d(2, 2), MaxPool2d(2, 2)
        elif down_mode == 'stride':
            self.down1 = ConvLayer(o[1], o[1], stride=2)
            self.down2 = ConvLayer(i[2] * 2, i[2] * 2, stride=2)
            self.down3 = ConvLayer(o[2], o[2], stride=2)

    def forward(self, feats):
        x2_1 = self.EB2_1(concat_fusion(feats[1]))
        x3_1 = self.EB3_1(concat_fusion(feats[2]))
        x4_1 = self.EB4_1(concat_fusion(feats[3]))
        x3_2 = self.EB3_2(concat_fusion((feats[2][0], x3_1, self.down1(x2_1))))
        x4_2 = self.EB4_2(concat_fusion((feats[3][0], x4_1, self.down2(x3_1))))
        x4_3 = self.EB4_3(concat_fusion((feats[3][0], x4_1, x4_2, self.down3(x3_2))))
        return feats[0], x2_1, x3_2, x4_3


class FSDecoder(nn.Module):
    """MAFusion's full-scale skip decoder (U-Net 3+; reference core/block.py:870-939): every decoder level concatenates all four
    scales, max-pooled down or bilinearly up-sampled (x2 / x4 / x8) to its own size."""

    def __init__(self, block, num_ch, up_mode='bilinear'):
        super(FSDecoder, self).__init__()
        cat_ch = sum(num_ch[:4])
        self.DB1, self.DB2, self.DB3 = block(cat_ch, num_ch[0]), block(cat_ch, num_ch[1]), block(cat_ch, num_ch[2])
        self.down1, self.down2 = Downsample(2, 2), Downsample(4, 4)
        self.up1, self.up2, self.up3 = Upsample(up_mode, 2), Upsample(up_mode, 4), Upsample(up_mode, 8)

    def forward(self, feats):
        f = feats
        y3 = self.DB3(concat_fusion((self.down2(f[0], f[2].shape), self.down1(f[1], f[2].shape), f[2], self.up1(f[3], f[2].shape))))
        y2 = self.DB2(concat_fusion((self.down1(f[0], f[1].shape), f[1], self.up1(y3, f[1].shape), self.up2(f[3], f[1].shape))))
        return self.DB1(concat_fusion((f[0], self.up1(y2, f[0].shape), self.up2(y3, f[0].shape), self.up3(f[3], f[0].shape))))
