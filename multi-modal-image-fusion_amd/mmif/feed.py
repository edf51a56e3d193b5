"""On-device batch feed for patch training: the uint8 patch banks of both modalities live in HBM; every
batch is gathered by index with /255 (or min-max / z-score) normalisation and one of the 8 dihedral augmentations applied by
csrc/feed.hip -- replacing FusionPatches.__getitem__ (data/patches.py:61-74), data/transform.py and the DataLoader's worker
processes, collate, pin_memory and H2D copy (train.py:207-222).  Sampling follows DataLoader(shuffle=True) /
DistributedSampler semantics: one permutation per epoch from (seed + epoch), rank r takes indices r, r + world, ... of the
permutation padded to a multiple of world_size; the augmentation mode is drawn per sample (np.random.choice(8) in the reference)
and is shared by the two images of a pair."""
import ctypes as C

import numpy as np
import torch

from ._lib import check, lib
from .tensor import stream_ptr

NORM_MODES = {None: 0, 'min-max': 1, 'z-score': 2}


class DevicePatchFeed:
    def __init__(self, patches1, patches2, batch_size, device, norm=None, transform=False, shuffle=True, seed=0, rank=0,
                 world_size=1, drop_last=False):
        if norm not in NORM_MODES:
            raise ValueError("only supported ['min-max', 'z-score'] mode")
        p1 = torch.as_tensor(np.ascontiguousarray(patches1)) if not torch.is_tensor(patches1) else patches1
        p2 = torch.as_tensor(np.ascontiguousarray(patches2)) if not torch.is_tensor(patches2) else patches2
        if p1.dtype != torch.uint8 or p2.dtype != torch.uint8:
            raise TypeError("patch banks must be uint8 [n, P, P]")
        if p1.shape != p2.shape or p1.dim() != 3 or p1.shape[1] != p1.shape[2]:
            raise ValueError(f"patch banks must both be [n, P, P]; got {tuple(p1.shape)} and {tuple(p2.shape)}")
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError("mmif: the patch feed runs on the GPU (no CPU path)")
        self.bank1, self.bank2 = p1.contiguous().to(self.device), p2.contiguous().to(self.device)
        self.n, self.P = int(p1.shape[0]), int(p1.shape[1])
        self.batch_size, self.norm_mode, self.transform = int(batch_size), NORM_MODES[norm], bool(transform)
        self.shuffle, self.seed, self.rank, self.world, self.drop_last = shuffle, int(seed), int(rank), int(world_size), drop_last
        self.epoch = 0

    def set_epoch(self, epoch):
        """DistributedSampler.set_epoch (train.py:339)."""
        self.epoch = int(epoch)

    def indices(self):
        """This rank's sample indices for the current epoch (host tensor, int32)."""
        g = torch.Generator().manual_seed(self.seed + self.epoch)
        order = torch.randperm(self.n, generator=g) if self.shuffle else torch.arange(self.n)
        if self.world > 1:
            total = (self.n + self.world - 1) // self.world * self.world
            if total > self.n:
                order = torch.cat((order, order[:total - self.n]))
            order = order[self.rank::self.world]
        return order.to(torch.int32)

    def __len__(self):
        m = (self.n + self.world - 1) // self.world if self.world > 1 else self.n
        return m // self.batch_size if self.drop_last else (m + self.batch_size - 1) // self.batch_size

    def gather(self, idx, modes=None):
        """One batch: idx int32 [B] (device), modes int32 [B] in 0..7 or None -> (img1, img2) fp32 [B,1,P,P]."""
        b = int(idx.numel())
        out1 = torch.empty((b, 1, self.P, self.P), dtype=torch.float32, device=self.device)
        out2 = torch.empty_like(out1)
        pm = C.c_void_p(modes.data_ptr()) if modes is not None else None
        for bank, out in ((self.bank1, out1), (self.bank2, out2)):
            check(lib.mmif_patch_feed(C.c_void_p(bank.data_ptr()), self.n, self.P, C.c_void_p(idx.data_ptr()), pm, b, self.norm_mode,
                                      C.c_void_p(out.data_ptr()), stream_ptr()), "patch_feed")
        return out1, out2

    def __iter__(self):
        order = self.indices().to(self.device)
        g = torch.Generator().manual_seed((self.seed + self.epoch) * 8191 + self.rank)
        modes = torch.randint(0, 8, (order.numel(),), generator=g, dtype=torch.int32).to(self.device) if self.transform else None
        for i in range(len(self)):
            sl = slice(i * self.batch_size, min((i + 1) * self.batch_size, order.numel()))
            yield self.gather(order[sl].contiguous(), modes[sl].contiguous() if modes is not None else None)
