# -*- coding: utf-8 -*-
"""Fusion models -- API mirror of the reference's core/model.py for the hot-path nets: same class
names, zero-argument constructors, forward(img1, img2) -> [B,1,H,W], the same sub-module
registration order and therefore identical state_dict keys / shapes (checkpoints interchange).
forward() hands the whole network to mmif.engine (one autograd node per model call).
"""
import torch
import torch.nn as nn

from mmif import engine as E

from .block import *
from .fusion import *

__all__ = ['PFNetv1', 'PFNetv2', 'DenseFuse', 'VIFNet', 'NestFuse', 'RFNNest',                       # the hot-path nets (fused engines)
           'DeepFuse', 'DBNet', 'SEDRFuse', 'UNFusion', 'Res2Fusion', 'MAFusion', 'IFCNN', 'DIFNet', 'PMGI']   # core/zoo.py


class _FusionModel(nn.Module):
    '''Base class for siamese-style fusion models (reference core/model.py:27-63).'''

    def __init__(self):
        super(_FusionModel, self).__init__()
        self.encode = nn.Sequential()
        self.decode = nn.Sequential()
        self._engine = None
        self._engine_single = True   # engine also serves the auto-encoder call forward(img1)

    def encoder(self, img):
        return self.encode(img)

    def fusion(self, feat1, feat2):
        raise NotImplementedError

    def decoder(self, feat):
        return self.decode(feat)

    def _make_engine(self):
        return None

    def forward(self, img1, img2=None):
        if self._engine is None:
            self._engine = self._make_engine()
        if self._engine is not None and (img2 is not None or self._engine_single):
            return self._engine.run(img1, img2)
        if img2 is None:
            return self.decoder(self.encoder(img1))
        return self.decoder(self.fusion(self.encoder(img1), self.encoder(img2)))


class PFNetv1(nn.Module):
    '''PFNet: An Unsupervised Deep Network for Polarization Image Fusion (reference core/model.py:69-111)'''

    def __init__(self):
        super(PFNetv1, self).__init__()
        self.encode1 = nn.Sequential(ConvLayer(1, 16), DenseBlock(16, 16))
        self.encode2 = nn.Sequential(ConvLayer(1, 16), DenseBlock(16, 16))
        self.decode = nn.Sequential(ConvLayer(128, 128), ConvLayer(128, 64), ConvLayer(64, 32), ConvLayer(32, 16),
                                    ConvLayer(16, 1, act=None))
        self._engine = None

    # block-level API of the reference, kept for callers that use the pieces
    def encoder(self, img1, img2):
        return self.encode1(img1), self.encode2(img2)

    def fusion(self, feats):
        return concat_fusion(feats)

    def decoder(self, feat):
        return self.decode(feat)

    def forward(self, img1, img2):
        if self._engine is None:
            self._engine = E.PFNetv1Engine(self)
        return self._engine.run(img1, img2)


class PFNetv2(_FusionModel):
    '''Polarization Image Fusion with Self-Learned Fusion Strategy (reference core/model.py:114-141).
    The reference's 64-iteration per-channel Python loop over the shared `fuse` stack (192 tiny conv
    launches per forward) is one batched call on [B*64, 2, H, W] -- same weights, same result.
    forward() runs on mmif.engine.PFNetv2Engine (one pair-conv launch per fuse layer, csrc/pair.hip); the
    stand-alone fusion() below keeps the block-level API on plain tensors.'''

    def __init__(self):
        super(PFNetv2, self).__init__()
        self.encode = nn.Sequential(ConvLayer(1, 16), DenseBlock(16, 16))
        self.fuse = nn.Sequential(ConvLayer(2, 2), ConvLayer(2, 2), ConvLayer(2, 1, act=None))
        self.decode = nn.Sequential(ConvLayer(64, 64), ConvLayer(64, 32), ConvLayer(32, 16), ConvLayer(16, 1, act=None))

    def fusion(self, feat1, feat2):
        b, c, h, w = feat1.shape
        pairs = torch.stack((feat1, feat2), dim=2).reshape(b * c, 2, h, w)
        return self.fuse(pairs).reshape(b, c, h, w) + feat1 + feat2

    def _make_engine(self):
        return E.PFNetv2Engine(self)


class DenseFuse(_FusionModel):
    '''DenseFuse: A Fusion Approach to Infrared and Visible Images (reference core/model.py:165-186)'''

    def __init__(self):
        super(DenseFuse, self).__init__()
        self.encode = nn.Sequential(ConvLayer(1, 16), DenseBlock(16, 16))
        self.decode = nn.Sequential(ConvLayer(64, 64), ConvLayer(64, 32), ConvLayer(32, 16), ConvLayer(16, 1, act=None))

    def fusion(self, feat1, feat2, mode='sum'):
        if mode == 'sum':
            return element_fusion(feat1, feat2, mode)
        elif mode == 'l1':
            return attention_fusion(feat1, feat2, 'sa', spatial_mode=mode)
        raise ValueError("only supported ['sum', 'l1'] mode")

    def _make_engine(self):
        return E.DenseFuseEngine(self)


class VIFNet(_FusionModel):
    '''VIF-Net: An Unsupervised Framework for Infrared and Visible Image Fusion (reference core/model.py:189-206):
    DenseFuse's shared encoder, concat fusion, PFNetv1's decoder.'''

    def __init__(self):
        super(VIFNet, self).__init__()
        self.encode = nn.Sequential(ConvLayer(1, 16), DenseBlock(16, 16))
        self.decode = nn.Sequential(ConvLayer(128, 128), ConvLayer(128, 64), ConvLayer(64, 32), ConvLayer(32, 16),
                                    ConvLayer(16, 1, act=None))
        self._engine_single = False   # forward(img1) alone has no meaning here (the decoder takes 128 channels)

    def fusion(self, feat1, feat2):
        return concat_fusion((feat1, feat2))

    def _make_engine(self):
        return E.VIFNetEngine(self)


class NestFuse(_FusionModel):
    '''NestFuse (reference core/model.py:319-363): 1x1 conv_in, four ConvBlock levels with 2x2 max-pool,
    spatial/channel attention fusion per level, UNet++ NestDecoder, 1x1 conv_out.  One autograd node per call
    (mmif/nest_engine.py): every ConvLayer (k = 1 and 3), the 2x2 max-pools, the nearest-x2 up-sampling with its reflect
    pad and the attention fusion run as HIP kernels on blocked buffers (csrc/nest.hip, csrc/conv_mfma.hip, csrc/conv1x1.hip).'''

    def __init__(self, down_mode='maxpool', up_mode='nearest'):
        super(NestFuse, self).__init__()
        num_ch = [64, 112, 160, 208]
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
        self.decode = NestDecoder(ConvBlock, num_ch, up_mode)
        self.conv_out = ConvLayer(num_ch[0], 1, ksize=1)

    def encoder(self, img):
        x1_0 = self.CB1_0(self.conv_in(img))
        x2_0 = self.CB2_0(self.down1(x1_0))
        x3_0 = self.CB3_0(self.down2(x2_0))
        x4_0 = self.CB4_0(self.down3(x3_0))
        return x1_0, x2_0, x3_0, x4_0

    def fusion(self, feats1, feats2, mode='sca'):
        return tuple(attention_fusion(a, b, mode) for a, b in zip(feats1, feats2))

    def _make_engine(self):
        # fused engine for the reference's default configuration; other down/up modes run block by block
        self._engine_single = False
        if isinstance(self.down1, nn.MaxPool2d) and isinstance(self.decode.up.up, nn.Upsample) and self.decode.up.up.mode == 'nearest':
            from mmif.nest_engine import NestEngine
            return NestEngine(self, rfn=hasattr(self, 'RFN1'))
        return None

    def decoder(self, feats):
        return self.conv_out(self.decode(feats))


class RFNNest(NestFuse):
    '''RFN-Nest (reference core/model.py:366-384): NestFuse with a learnable RFN per pyramid level.'''

    def __init__(self, down_mode='maxpool', up_mode='nearest'):
        super(RFNNest, self).__init__(down_mode, up_mode)
        num_ch = [64, 112, 160, 208]
        self.RFN1, self.RFN2, self.RFN3, self.RFN4 = RFN(num_ch[0]), RFN(num_ch[1]), RFN(num_ch[2]), RFN(num_ch[3])

    def fusion(self, feats1, feats2):
        return (self.RFN1(feats1[0], feats2[0]), self.RFN2(feats1[1], feats2[1]), self.RFN3(feats1[2], feats2[2]),
                self.RFN4(feats1[3], feats2[3]))


# the other classic models (scope row n4: layer by layer on the general HIP kernels) live in core/zoo.py; they subclass the bases above
from .zoo import *  # noqa: E402,F401,F403
