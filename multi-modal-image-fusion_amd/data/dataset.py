# -*- coding: utf-8 -*-
"""Whole-image datasets -- API mirror of the reference's data/dataset.py: FusionDataset (:29-119) and AEDataset (:122-196)
with the reference's constructor signatures, directory convention (<root>/[set_name/]vis + .../ir|po), 80/20 sklearn
split (random_state=0) and per-sample semantics (norm -> one of TWO augmentations, np.random.choice(2) -> stack ->
random 256-crop, or crop-to-square + resize when the image is smaller).  The hot path starts after this: the pair
goes to the HIP engine as two [1,H,W] fp32 tensors.

cv2 / natsort / torchvision are not required: files are read through data/_io.py (cv2 when present), and
torchvision's RandomCrop / Resize on tensors are restated with the same torch RNG draws (`torch.randint` for the top and
the left offset, in that order) and `F.interpolate(bilinear, antialias=True)`, which is what tf.Resize runs on a tensor.
"""
import random
from functools import partial

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import Dataset

from ._io import IMG_EXT, imread_gray, list_pairs, natural_sorted
from .transform import norm, transform

img_size = 256

__all__ = ['FusionDataset', 'AEDataset', 'random_crop', 'resize']


def random_crop(img, size):
    """torchvision.transforms.RandomCrop(size) on a [..., H, W] tensor (get_params: i then j from torch.randint)."""
    h, w = img.shape[-2:]
    if h < size or w < size:
        raise ValueError(f'Required crop size {(size, size)} is larger than input image size {(h, w)}')
    if h == size and w == size:
        return img
    i = torch.randint(0, h - size + 1, size=(1,)).item()
    j = torch.randint(0, w - size + 1, size=(1,)).item()
    return img[..., i:i + size, j:j + size]


def resize(img, size):
    """torchvision.transforms.Resize(int) on a [C, H, W] float tensor: smaller edge -> size, bilinear, antialias."""
    h, w = img.shape[-2:]
    if h <= w:
        nh, nw = size, int(size * w / h)
    else:
        nh, nw = int(size * h / w), size
    if (nh, nw) == (h, w):
        return img
    return F.interpolate(img.unsqueeze(0), size=(nh, nw), mode='bilinear', align_corners=False, antialias=True).squeeze(0)


def _fix_size(img):
    min_size = min(img.shape[-2:])
    if min_size < img_size:
        return resize(random_crop(img, min_size), img_size)
    return random_crop(img, img_size)


class FusionDataset(Dataset):
    def __init__(self, root_dir, set_name=None, set_type='train', img_type='ir', norm=None, transform=False, fix_size=False):
        super(FusionDataset, self).__init__()
        assert set_type in ('train', 'valid', 'test')
        assert img_type in ('ir', 'po')
        self.root_dir, self.set_name, self.set_type, self.img_type = root_dir, set_name, set_type, img_type
        self.norm, self.transform, self.fix_size = norm, transform, fix_size
        self.data_info, self.train_data_info, self.valid_data_info = [], [], []
        self._get_data_info()
        if set_type == 'train':
            self.data_info = self.train_data_info
        elif set_type == 'valid':
            self.data_info = self.valid_data_info

    def __getitem__(self, index):
        p1, p2 = self.data_info[index]
        pair = (imread_gray(p1).astype(np.float32), imread_gray(p2).astype(np.float32))
        pair = tuple(map(partial(norm, mode=self.norm), pair))
        if self.transform:
            idx = np.random.choice(2)
            pair = tuple(map(partial(transform, mode=idx), pair))
        pair = torch.stack(tuple(torch.from_numpy(p.copy()).float() for p in pair), dim=0)
        if self.fix_size:
            pair = _fix_size(pair)
        return torch.chunk(pair, 2, dim=0)

    def __len__(self):
        assert len(self.data_info) > 0
        return len(self.data_info)

    def _get_data_info(self):
        info1, info2 = list_pairs(self.root_dir, self.set_name, self.img_type)
        if self.set_type in ('train', 'valid'):
            from sklearn.model_selection import train_test_split
            tr1, va1, tr2, va2 = train_test_split(info1, info2, test_size=0.2, random_state=0)
            self.train_data_info = list(zip(tr1, tr2))
            self.valid_data_info = list(zip(va1, va2))
        else:
            self.data_info = list(zip(info1, info2))


class AEDataset(Dataset):
    """Single images of both modalities, shuffled once (auto-encoder training: model(img))."""

    def __init__(self, root_dir, set_name=None, img_type='ir', norm=None, transform=False, fix_size=False):
        super(AEDataset, self).__init__()
        assert img_type in ('ir', 'po')
        self.root_dir, self.set_name, self.img_type = root_dir, set_name, img_type
        self.norm, self.transform, self.fix_size = norm, transform, fix_size
        self.data_info = []
        self._get_data_info()

    def __getitem__(self, index):
        img = norm(imread_gray(self.data_info[index]).astype(np.float32), mode=self.norm)
        if self.transform:
            img = transform(img, mode=np.random.choice(2))
        img = torch.from_numpy(img.copy()).float().unsqueeze(0)
        if self.fix_size:
            img = _fix_size(img)
        return img

    def __len__(self):
        assert len(self.data_info) > 0
        return len(self.data_info)

    def _get_data_info(self):
        import os
        dir1 = os.path.join(self.root_dir, 'vis') if self.set_name is None else os.path.join(self.root_dir, self.set_name, 'vis')
        dir2 = dir1.replace('vis', self.img_type)
        for d in (dir1, dir2):
            for name in natural_sorted(os.listdir(d)):
                if name.endswith(IMG_EXT):
                    self.data_info.append(os.path.join(d, name))
        random.shuffle(self.data_info)
