# -*- coding: utf-8 -*-
"""The reference's other classic fusion models (core/model.py sections 2 and 3) behind the same class names, constructor
arguments and state_dict layouts: DeepFuse, DBNet, SEDRFuse, UNFusion, Res2Fusion, MAFusion, IFCNN, DIFNet, PMGI.

Scope row n4: none of them has a fused engine -- every ConvLayer / norm / activation / resampling step is its own HIP kernel
launch through autograd (csrc/conv_general.hip, norm.hip, resample.hip, and the hot-path conv kernels where a layer's
geometry allows).  Parity against golden vectors from the reference: tests/test_gpu_general_conv.py (F12 - F15).
"""
import torch
import torch.nn as nn

from .block import *
from .fusion import *

__all__ = ['DeepFuse', 'DBNet', 'SEDRFuse', 'UNFusion', 'Res2Fusion', 'MAFusion', 'IFCNN', 'DIFNet', 'PMGI']


def _bases():
    from . import model   # (model.py imports this module at its end: the bases exist by then)
    return model._FusionModel, model.NestFuse


_FusionModel, NestFuse = _bases()


class DeepFuse(_FusionModel):
    '''DeepFuse (reference core/model.py:146-162): 5x5 / 7x7 ConvLayers, element-wise fusion.  Runs layer by layer on the general
    HIP conv kernels (csrc/conv_general.hip, fp32) -- row n4 of the scope table, no fused engine.'''

    def __init__(self):
        super(DeepFuse, self).__init__()
        self.encode = nn.Sequential(ConvLayer(1, 16, ksize=5), ConvLayer(16, 32, ksize=7))
        self.decode = nn.Sequential(ConvLayer(32, 32, ksize=7), ConvLayer(32, 16, ksize=5), ConvLayer(16, 1, ksize=5, act=None))

    def fusion(self, feat1, feat2, mode='sum'):
        return element_fusion(feat1, feat2, mode)


class DBNet(_FusionModel):
    '''A Dual-Branch Network for Infrared and Visible Image Fusion (reference core/model.py:208-245): a detail branch
    (ConvLayer + DenseBlock) and a semantic branch (three stride-2 ConvLayers, bilinear x8 back to full size), concatenated.
    Layer by layer: the 3x3 stride-1 layers on the hot-path kernels, stride 2 on the general kernels, csrc/resample.hip.'''

    def __init__(self):
        super(DBNet, self).__init__()
        self.encode = ConvLayer(1, 32)
        self.detail = nn.Sequential(ConvLayer(32, 16), DenseBlock(16, 16))
        self.semantic = nn.Sequential(ConvLayer(32, 64, stride=2), ConvLayer(64, 128, stride=2), ConvLayer(128, 64, stride=2))
        self.up = Upsample(mode='bilinear', scale_factor=8)
        self.decode = nn.Sequential(ConvLayer(128, 64), ConvLayer(64, 32), ConvLayer(32, 16), ConvLayer(16, 1, act=None))

    def encoder(self, img):
        feat = self.encode(img)
        return concat_fusion((self.detail(feat), self.up(self.semantic(feat), feat.shape)))

    def fusion(self, feat1, feat2, mode='sum'):
        if mode == 'sum':
            return element_fusion(feat1, feat2, mode)
        elif mode == 'avg':
            return attention_fusion(feat1, feat2, 'ca', channel_mode=mode)
        raise ValueError("only supported ['sum', 'avg'] mode")


def _relu_sum(a, b):
    '''relu(a + b) on the HIP element-wise kernels'''
    from .block import _ActFn
    from mmif import tensor as T
    return _ActFn.apply(element_fusion(a, b, 'sum'), T.ACT_RELU)


class SEDRFuse(nn.Module):
    '''SEDRFuse (reference core/model.py:247-312): symmetric encoder-decoder with a residual block; GroupNorm(c, c) after every
    conv, two stride-2 convs down, two ConvTranspose2d up, skip connections relu(f_conv + f_deconv).  Layer by layer on the general
    conv kernels + the norm epilogue kernels (fp32).'''

    def __init__(self, norm=nn.GroupNorm):
        super(SEDRFuse, self).__init__()
        self.encode = nn.ModuleList([ConvLayer(1, 64, norm=norm), ConvLayer(64, 128, stride=2, norm=norm),
                                     ConvLayer(128, 256, stride=2, norm=norm), ResBlock(256, 256, norm1=norm, norm2=norm)])
        self.decode = nn.ModuleList([ConvLayer(256, 128, stride=2, norm=norm, layer=nn.ConvTranspose2d),
                                     ConvLayer(128, 64, stride=2, norm=norm, layer=nn.ConvTranspose2d), ConvLayer(64, 1)])

    def encoder(self, img):
        c1 = self.encode[0](img)
        c2 = self.encode[1](c1)
        return c1, c2, self.encode[3](self.encode[2](c2))

    def fusion(self, feat1, feat2):
        # channel-softmax-weighted L1 activity of each source -> per-pixel weights (tensor-level composition)
        a1, a2 = torch.abs(feat1), torch.abs(feat2)
        s1 = spatial_pooling(torch.softmax(a1, dim=1) * a1, mode='sum')
        s2 = spatial_pooling(torch.softmax(a2, dim=1) * a2, mode='sum')
        return weighted_fusion(feat1, feat2, s1, s2)

    def decoder(self, f_conv1, f_conv2, f_res):
        f1 = _relu_sum(f_conv2, self.decode[0](f_res))
        f2 = _relu_sum(f_conv1, self.decode[1](f1))
        return self.decode[2](f2)

    def forward(self, img1, img2=None):
        if img2 is None:
            return self.decoder(*self.encoder(img1))
        a, b = self.encoder(img1), self.encoder(img2)
        return self.decoder(element_fusion(a[0], b[0], mode='max'), element_fusion(a[1], b[1], mode='max'), self.fusion(a[2], b[2]))


class UNFusion(_FusionModel):
    '''UNFusion (reference core/model.py:386-436): four single-conv levels joined by stride-2 ConvLayers, the densely nested
    NestEncoder (ECB blocks), 'wavg' attention fusion per level, the UNet++ NestDecoder (DCB blocks, bilinear up-sampling).
    Layer by layer on the HIP conv / resample kernels.'''

    def __init__(self, down_mode='stride', up_mode='bilinear'):
        super(UNFusion, self).__init__()
        enc_ch, dec_ch = [16, 32, 48, 64], [16, 64, 256, 1024]
        self.CB1_0, self.CB2_0 = ConvLayer(1, enc_ch[0]), ConvLayer(enc_ch[0], enc_ch[1])
        self.CB3_0, self.CB4_0 = ConvLayer(enc_ch[1], enc_ch[2]), ConvLayer(enc_ch[2], enc_ch[3])
        if down_mode == 'maxpool':
            self.down1, self.down2, self.down3 = MaxPool2d(2, 2), MaxPool2d(2, 2), MaxPool2d(2, 2)
        elif down_mode == 'stride':
            self.down1 = ConvLayer(enc_ch[0], enc_ch[0], stride=2)
            self.down2 = ConvLayer(enc_ch[1], enc_ch[1], stride=2)
            self.down3 = ConvLayer(enc_ch[2], enc_ch[2], stride=2)
        self.encode = NestEncoder(ECB, enc_ch, dec_ch, down_mode)
        self.decode = NestDecoder(DCB, dec_ch, up_mode)
        self.conv_out = ConvLayer(dec_ch[0], 1, ksize=1)

    def encoder(self, img):
        x1 = self.CB1_0(img)
        d1 = self.down1(x1)
        x2 = self.CB2_0(d1)
        d2 = self.down2(x2)
        x3 = self.CB3_0(d2)
        d3 = self.down3(x3)
        return self.encode((x1, (x2, d1), (x3, d2), (self.CB4_0(d3), d3)))

    def fusion(self, feats1, feats2, mode='wavg'):
        return tuple(attention_fusion(a, b, mode) for a, b in zip(feats1, feats2))

    def decoder(self, feats):
        return self.conv_out(self.decode(feats))


class Res2Fusion(_FusionModel):
    '''Res2Fusion (reference core/model.py:439-470): a dense encoder of Res2ConvBlocks (point-wise + hierarchical depth-wise convs,
    ReLU6) and double non-local attention fusion.  Convs, depth-wise convs and activations on the HIP kernels; the non-local
    attention maps are tensor-level compositions (batched matmuls).'''

    def __init__(self):
        super(Res2Fusion, self).__init__()
        self.conv_in = ConvLayer(1, 16)
        self.RB1 = Res2ConvBlock(16, 32, 4)
        self.RB2 = Res2ConvBlock(48, 64, 8)
        self.decode = nn.Sequential(ConvLayer(112, 64), ConvLayer(64, 32), ConvLayer(32, 16), ConvLayer(16, 1))

    def encoder(self, img):
        x = self.conv_in(img)
        x = concat_fusion((x, self.RB1(x)))
        return concat_fusion((x, self.RB2(x)))

    def fusion(self, feat1, feat2, mode='attn', spatial='nl', channel='nl'):
        if mode == 'elem':
            return element_fusion(feat1, feat2, 'mean')
        elif mode == 'attn':
            return attention_fusion(feat1, feat2, 'sca', spatial, channel)
        raise ValueError("only supported ['elem', 'attn'] mode")


class MAFusion(NestFuse):
    '''MAFusion (reference core/model.py:473-508): NestFuse's encoder with wider levels and the full-scale skip FSDecoder
    (bilinear x2 / x4 / x8, max-pool /2 and /4).'''

    def __init__(self, down_mode='maxpool', up_mode='bilinear'):
        super(MAFusion, self).__init__(down_mode, up_mode)
        num_ch = [64, 128, 256, 512]
        self.conv_in = ConvLayer(1, 16, ksize=1)
        self.CB1_0 = ConvBlock(16, num_ch[0])
        self.CB2_0 = ConvBlock(num_ch[0], num_ch[1])
        self.CB3_0 = ConvBlock(num_ch[1], num_ch[2])
        self.CB4_0 = ConvBlock(num_ch[2], num_ch[3])
        if down_mode == 'maxpool':
            self.down1, self.down2, self.down3 = MaxPool2d(2, 2), MaxPool2d(2, 2), MaxPool2d(2, 2)
        elif down_mode == 'stride':
            self.down1 = ConvLayer(num_ch[0], num_ch[0], stride=2)
            self.down2 = ConvLayer(num_ch[1], num_ch[1], stride=2)
            self.down3 = ConvLayer(num_ch[2], num_ch[2], stride=2)
        self.decode = FSDecoder(ConvBlock, num_ch, up_mode)
        self.conv_out = ConvLayer(num_ch[0], 1, ksize=1)

    def _make_engine(self):
        self._engine_single = False
        return None   # (the fused NestEngine is NestFuse / RFN-Nest's; MAFusion runs block by block)


class IFCNN(_FusionModel):
    '''IFCNN (reference core/model.py:514-530): 7x7 conv without activation, BatchNorm ConvLayers, element-wise max fusion.'''

    def __init__(self, norm=nn.BatchNorm2d):
        super(IFCNN, self).__init__()
        self.encode = nn.Sequential(ConvLayer(1, 64, ksize=7, act=None), ConvLayer(64, 64, norm=norm))
        self.decode = nn.Sequential(ConvLayer(64, 64, norm=norm), ConvLayer(64, 1, ksize=1, act=None))

    def fusion(self, feat1, feat2, mode='max'):
        return element_fusion(feat1, feat2, mode)


class DIFNet(_FusionModel):
    '''DIFNet (reference core/model.py:533-553): residual blocks with BatchNorm, a 3x3 ConvLayer fusing the concatenated features.'''

    def __init__(self, norm=nn.BatchNorm2d):
        super(DIFNet, self).__init__()
        self.encode = nn.Sequential(ConvLayer(1, 16), ResBlock(16, 16, norm1=norm), ResBlock(16, 16, norm1=norm))
        self.fuse = ConvLayer(32, 16, act=None)
        self.decode = nn.Sequential(ResBlock(16, 16, norm1=norm), ResBlock(16, 16, norm1=norm), ResBlock(16, 16, norm1=norm),
                                    ConvLayer(16, 1, act=None))

    def fusion(self, feat1, feat2):
        return self.fuse(concat_fusion((feat1, feat2)))


class PMGI(nn.Module):
    '''PMGI (reference core/model.py:556-624): a gradient and an intensity path of BatchNorm + LeakyReLU ConvLayers that exchange
    1x1 "transfer" features, all eight feature maps concatenated into a 1x1 Tanh ConvLayer; output tanh / 2 + 0.5.'''

    def __init__(self, norm=nn.BatchNorm2d, act=nn.LeakyReLU):
        super(PMGI, self).__init__()

        def path():
            return nn.ModuleList([ConvLayer(3, 16, ksize=5, norm=norm, act=act), ConvLayer(16, 16, norm=norm, act=act),
                                  ConvLayer(48, 16, norm=norm, act=act), ConvLayer(64, 16, norm=norm, act=act)])

        def transfer():
            return nn.ModuleList([ConvLayer(32, 16, ksize=1, norm=norm, act=act), ConvLayer(32, 16, ksize=1, norm=norm, act=act)])
        self.gradient, self.intensity = path(), path()
        self.transfer1, self.transfer2 = transfer(), transfer()
        self.decode = ConvLayer(128, 1, ksize=1, act=nn.Tanh)

    def encoder(self, img1, img2):
        g0 = self.gradient[0](concat_fusion((img1, img1, img2)))
        i0 = self.intensity[0](concat_fusion((img2, img2, img1)))
        g1, i1 = self.gradient[1](g0), self.intensity[1](i0)
        t = concat_fusion((g1, i1))
        # (the reference routes the intensity path's first exchange through transfer2[1]; transfer1[1] is never called -- kept)
        g2 = self.gradient[2](concat_fusion((g0, g1, self.transfer1[0](t))))
        i2 = self.intensity[2](concat_fusion((i0, i1, self.transfer2[1](t))))
        t = concat_fusion((g2, i2))
        g3 = self.gradient[3](concat_fusion((g0, g1, g2, self.transfer2[0](t))))
        i3 = self.intensity[3](concat_fusion((i0, i1, i2, self.transfer2[1](t))))
        return g0, i0, g1, i1, g2, i2, g3, i3

    def fusion(self, feats):
        return concat_fusion(feats)

    def decoder(self, feat):
        return self.decode(feat)

    def forward(self, img1, img2):
        return self.decoder(self.fusion(self.encoder(img1, img2))) / 2.0 + 0.5
