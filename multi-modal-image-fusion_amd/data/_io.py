# -*- coding: utf-8 -*-
"""Image file access shared by data/dataset.py and data/patches.py: what the reference does with
`cv2.imread(path, cv2.IMREAD_GRAYSCALE)` (data/dataset.py:60-61, data/patches.py:107-108) and `natsorted(os.listdir())`
(:103, :82) without requiring cv2 / natsort.  cv2 is used when it is importable (bit-identical to the reference for every
file type); otherwise PIL decodes and colour files are reduced with OpenCV's 8-bit fixed-point luma
(9798 R + 19235 G + 3735 B + 2^14) >> 15, which is also what PIL's own 'L' conversion rounds to within one level."""
import os
import re

import numpy as np

IMG_EXT = ('.bmp', '.jpg', '.png')

__all__ = ['imread_gray', 'imwrite', 'natural_sorted', 'list_pairs', 'IMG_EXT']


def natural_sorted(names):
    """natsort.natsorted for plain file names: digit runs compare as integers."""
    return sorted(names, key=lambda s: [int(t) if t.isdigit() else t.lower() for t in re.split(r'(\d+)', s)])


def imread_gray(path):
    """uint8 [H, W] luminance of an image file."""
    try:
        import cv2
    except ImportError:
        cv2 = None
    if cv2 is not None:
        img = cv2.imread(path, cv2.IMREAD_GRAYSCALE)
        assert img is not None, f'cannot read {path}'
        return img
    from PIL import Image
    im = Image.open(path)
    if im.mode not in ('RGB', 'RGBA', 'P', 'CMYK', 'YCbCr'):   # single-channel files: no colour reduction involved
        return np.asarray(im.convert('L'), dtype=np.uint8)
    rgb = np.asarray(im.convert('RGB'), dtype=np.int64)
    gray = (rgb[..., 0] * 9798 + rgb[..., 1] * 19235 + rgb[..., 2] * 3735 + (1 << 14)) >> 15
    return gray.astype(np.uint8)


def imwrite(path, img):
    """cv2.imwrite for uint8 [H,W] / [H,W,1] / [H,W,3] arrays (cv2 when importable, else PIL; format from the extension)."""
    try:
        import cv2
        return bool(cv2.imwrite(path, img))
    except ImportError:
        from PIL import Image
        arr = np.asarray(img)
        if arr.ndim == 3 and arr.shape[2] == 1:
            arr = arr[:, :, 0]
        elif arr.ndim == 3:
            arr = arr[:, :, ::-1]   # cv2 arrays are BGR
        Image.fromarray(np.ascontiguousarray(arr)).save(path)
        return True


def list_pairs(root_dir, set_name, img_type):
    """(vis path, ir|po path) of every image under <root>/[set_name/]vis that has a partner (reference
    data/dataset.py:95-111): the partner's path is the vis path with EVERY 'vis' replaced, as the reference's
    str.replace does."""
    img_dir = os.path.join(root_dir, 'vis') if set_name is None else os.path.join(root_dir, set_name, 'vis')
    info1, info2 = [], []
    for name in natural_sorted(os.listdir(img_dir)):
        if name.endswith(IMG_EXT):
            p1 = os.path.join(img_dir, name)
            p2 = p1.replace('vis', img_type)
            if os.path.isfile(p2):
                info1.append(p1)
                info2.append(p2)
    return info1, info2
