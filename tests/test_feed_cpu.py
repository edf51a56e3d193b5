"""Data feed, CPU side: the oracle's and the host mirror's norm / dihedral transforms against
the reference's data/transform.py outputs (golden F8), and the numpy patch extraction."""
import os
import sys

import numpy as np
import pytest

from oracle import fusion_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")
NORMS = ((None, "none"), ("min-max", "minmax"), ("z-score", "zscore"))


def feed_patch(P, seed):
    y, x = np.mgrid[0:P, 0:P]
    return ((y * 37 + x * 11 + (y * x) * 5 + seed * 13) % 256).astype(np.uint8)


@pytest.mark.parametrize("P", [5, 6])
def test_oracle_feed_vs_golden(P):
    ref = np.load(os.path.join(G, "f8_feed.npz"))
    patch = feed_patch(P, P)
    for nm, tag in NORMS:
        n = O.feed_norm(patch.astype(np.float32), nm)
        for mode in range(8):
            got = O.feed_transform(n, mode)
            want = ref[f"P{P}_{tag}_m{mode}"]
            if nm is None:
                assert np.array_equal(got, want), (tag, mode)      # /255.0 and a permutation: bit exact
            else:
                assert np.abs(got - want).max() <= 1e-6, (tag, mode)


@pytest.mark.parametrize("P", [5, 6])
def test_host_transform_mirror_vs_golden(P):
    from data.transform import norm, transform
    ref = np.load(os.path.join(G, "f8_feed.npz"))
    patch = feed_patch(P, P).astype(np.float32)
    for nm, tag in NORMS:
        n = norm(patch.copy(), nm)
        for mode in range(8):
            got = np.ascontiguousarray(transform(n, mode)).astype(np.float32)
            assert np.abs(got - ref[f"P{P}_{tag}_m{mode}"]).max() <= 1e-6, (tag, mode)
    assert transform(patch, 9) is patch                         # unknown modes fall through unchanged (data/transform.py:38-66)
    with pytest.raises(ValueError, match="min-max"):
        norm(patch, "l2")


def test_extract_patches_matches_sliding_windows():
    from data.patches import extract_patches
    img = np.arange(7 * 9, dtype=np.uint8).reshape(7, 9)
    p = extract_patches(img, size=3, step=2)
    want = [img[y:y + 3, x:x + 3] for y in range(0, 5, 2) for x in range(0, 7, 2)]
    assert p.shape == (len(want), 3, 3) and all(np.array_equal(a, b) for a, b in zip(p, want))
    assert extract_patches(img[:2], size=3, step=2).shape == (0, 3, 3)
