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

__all__ = ['PFNetv1', 'DenseFuse']


class _FusionModel(nn.Module):
    '''Base class for siamese-style fusion models (reference core/model.py:27-63).'''

    def __init__(self):
        super(_FusionModel, self).__init__()
        self.encode = nn.Sequential()
        self.decode = nn.Sequential()
        self._engine = None

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
        if self._engine is not None:
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
