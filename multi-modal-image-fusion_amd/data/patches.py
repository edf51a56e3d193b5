# -*- coding: utf-8 -*-
"""Patch dataset -- API mirror of the reference's data/patches.py FusionPatches (:29-123): image pairs under
<root>/[set_name/]vis and .../ir|po, 80/20 train/valid split (sklearn, random_state=0), 64x64 patches with step 64,
shuffled once.  Differences, both on purpose:
  * patches are kept as uint8 (the reference keeps float32 copies of the same integers) and `device_feed()` uploads
    the whole bank to HBM once; batches are then produced on the device (mmif.feed.DevicePatchFeed);
  * images are read through data/_io.py (cv2 when importable, else PIL with OpenCV's luma) and cut with numpy strides
    (cv2 / patchify / natsort are not needed).
`__getitem__` keeps the reference's per-sample host semantics for code that indexes the dataset directly.
"""
import os
import random
from functools import partial

import numpy as np
import torch
from torch.utils.data import Dataset

from ._io import imread_gray, list_pairs
from .transform import norm, transform

patch_size = 64
patch_step = 64

__all__ = ['FusionPatches', 'extract_patches']


def extract_patches(img, size=patch_size, step=patch_step):
    """All size x size windows of a 2-D array at stride `step`, row-major -> [n, size, size] (patchify + reshape)."""
    h, w = img.shape
    ny, nx = (h - size) // step + 1, (w - size) // step + 1
    if ny <= 0 or nx <= 0:
        return np.empty((0, size, size), dtype=img.dtype)
    s0, s1 = img.strides
    view = np.lib.stride_tricks.as_strided(img, shape=(ny, nx, size, size), strides=(s0 * step, s1 * step, s0, s1), writeable=False)
    return np.ascontiguousarray(view).reshape(-1, size, size)


class FusionPatches(Dataset):
    def __init__(self, root_dir, set_name=None, set_type='train', img_type='ir', norm=None, transform=False):
        super(FusionPatches, self).__init__()
        assert set_type in ('train', 'valid', 'test')
        assert img_type in ('ir', 'po')
        self.root_dir, self.set_name, self.set_type, self.img_type = root_dir, set_name, set_type, img_type
        self.norm, self.transform = norm, transform
        self.data_info = []
        self._get_data_info()
        self.patches1 = np.empty((0, patch_size, patch_size), np.uint8)
        self.patches2 = np.empty((0, patch_size, patch_size), np.uint8)
        self._gen_patch_pairs()

    def __getitem__(self, index):
        pair = (self.patches1[index].astype(np.float32), self.patches2[index].astype(np.float32))
        pair = tuple(map(partial(norm, mode=self.norm), pair))
        if self.transform:
            idx = np.random.choice(8)
            pair = tuple(map(partial(transform, mode=idx), pair))
        return tuple(torch.from_numpy(np.ascontiguousarray(p)).float().unsqueeze(0) for p in pair)

    def __len__(self):
        assert len(self.patches1) > 0
        return len(self.patches1)

    def _get_data_info(self):
        info1, info2 = list_pairs(self.root_dir, self.set_name, self.img_type)
        if self.set_type in ('train', 'valid'):
            from sklearn.model_selection import train_test_split
            tr1, va1, tr2, va2 = train_test_split(info1, info2, test_size=0.2, random_state=0)
            self.data_info = list(zip(tr1, tr2)) if self.set_type == 'train' else list(zip(va1, va2))
        else:
            self.data_info = list(zip(info1, info2))

    def _gen_patch_pairs(self):
        p1, p2 = [], []
        for a, b in self.data_info:
            i1, i2 = imread_gray(a), imread_gray(b)
            p1.append(extract_patches(i1))
            p2.append(extract_patches(i2))
        if p1:
            self.patches1, self.patches2 = np.concatenate(p1), np.concatenate(p2)
            order = list(range(len(self.patches1)))
            random.shuffle(order)
            self.patches1, self.patches2 = self.patches1[order], self.patches2[order]

    def device_feed(self, batch_size, device, shuffle=True, seed=0, rank=0, world_size=1, drop_last=False):
        """The on-device replacement of DataLoader(self, batch_size, shuffle, sampler=DistributedSampler(...))."""
        from mmif.feed import DevicePatchFeed
        return DevicePatchFeed(self.patches1, self.patches2, batch_size, device, norm=self.norm, transform=self.transform,
                               shuffle=shuffle, seed=seed, rank=rank, world_size=world_size, drop_last=drop_last)
