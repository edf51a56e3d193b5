# -*- coding: utf-8 -*-
"""Image transforms -- API mirror of the reference's data/transform.py (norm :15-29, denorm :32-35,
transform :38-66) for host-side numpy arrays.  The training path does not call these per sample: the same
arithmetic runs on the device over a whole batch (mmif.feed.DevicePatchFeed -> csrc/feed.hip); they exist so
that code written against the reference's data package keeps working, and as the host view of the same table.
"""
import numpy as np

eps = 1e-7

__all__ = ['norm', 'denorm', 'transform', 'DIHEDRAL']

# mode -> (transpose first?, flip rows?, flip cols?) describing out = flips(transpose?(img)):
# 0 original, 1 fliplr, 2 rot180, 3 flipud, 4 rot90, 5 rot90+flipud, 6 rot270, 7 rot270+flipud
DIHEDRAL = {
    0: (False, False, False),
    1: (False, False, True),
    2: (False, True, True),
    3: (False, True, False),
    4: (True, True, False),
    5: (True, False, False),
    6: (True, False, True),
    7: (True, True, True),
}


def norm(img, mode=None):
    if mode is None:
        return img / 255.0
    if mode == 'min-max':
        lo, hi = img.min(), img.max()
        return (img - lo) / (hi - lo).clip(eps)
    if mode == 'z-score':
        return (img - img.mean()) / img.std().clip(eps)
    raise ValueError("only supported ['min-max', 'z-score'] mode")


def denorm(img):
    """[C,H,W] tensor in [0,1] -> uint8 [H,W,C] array."""
    im = img.detach().cpu().numpy().clip(0, 1) * 255.0
    return im.transpose((1, 2, 0)).astype(np.uint8)


def transform(img, mode=0):
    """One of the 8 dihedral variants of a 2-D image (modes outside 0..7 return the image unchanged, as in the reference)."""
    if mode not in DIHEDRAL or mode == 0:
        return img
    t, fr, fc = DIHEDRAL[mode]
    out = img.swapaxes(0, 1) if t else img
    if fr:
        out = out[::-1]
    if fc:
        out = out[:, ::-1]
    return out
