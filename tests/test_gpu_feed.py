"""Data feed on the device (csrc/feed.hip through the C ABI) vs the CPU oracle and the reference's values (golden F8)."""
import os

import numpy as np
import pytest
import torch

from oracle import fusion_oracle as O
from gpu_util import G
from test_feed_cpu import NORMS, feed_patch

pytestmark = pytest.mark.gpu


def test_patch_feed_all_modes_vs_golden():
    from mmif.feed import DevicePatchFeed
    ref = np.load(os.path.join(G, "f8_feed.npz"))
    for P in (5, 6):
        bank = np.stack([feed_patch(P, P), feed_patch(P, 3)])
        for nm, tag in NORMS:
            feed = DevicePatchFeed(bank, bank[::-1].copy(), 8, "cuda:0", norm=nm, transform=True)
            idx = torch.zeros(8, dtype=torch.int32, device="cuda:0")
            modes = torch.arange(8, dtype=torch.int32, device="cuda:0")
            a, b = feed.gather(idx, modes)
            assert a.shape == (8, 1, P, P) and a.dtype == torch.float32
            for m in range(8):
                want = ref[f"P{P}_{tag}_m{m}"]
                got = a[m, 0].cpu().numpy()
                if nm is None:
                    assert np.array_equal(got, want), (P, tag, m)     # bit exact: IEEE division by 255 + a permutation
                else:
                    assert np.abs(got - want).max() <= 2e-6, (P, tag, m)
            # second bank, no augmentation
            a0, b0 = feed.gather(idx, None)
            assert np.abs(b0[0, 0].cpu().numpy() - O.feed_norm(bank[1].astype(np.float32), nm)).max() <= 2e-6


def test_patch_feed_batches_vs_oracle_and_sampler_semantics():
    from mmif.feed import DevicePatchFeed
    rng = np.random.default_rng(0)
    n, P, B = 37, 64, 8
    b1 = rng.integers(0, 256, size=(n, P, P), dtype=np.uint8)
    b2 = rng.integers(0, 256, size=(n, P, P), dtype=np.uint8)
    feeds = [DevicePatchFeed(b1, b2, B, "cuda:0", transform=True, seed=5, rank=r, world_size=2) for r in range(2)]
    seen = []
    for f in feeds:
        f.set_epoch(3)
        idx = f.indices().numpy()
        seen.append(idx)
        assert len(f) == (len(idx) + B - 1) // B
        g = torch.Generator().manual_seed((5 + 3) * 8191 + f.rank)
        modes = torch.randint(0, 8, (len(idx),), generator=g, dtype=torch.int32).numpy()
        k = 0
        for i1, i2 in f:
            nb = i1.shape[0]
            want1 = O.patch_batch(b1, idx[k:k + nb], modes[k:k + nb])
            want2 = O.patch_batch(b2, idx[k:k + nb], modes[k:k + nb])
            assert np.array_equal(i1.cpu().numpy(), want1) and np.array_equal(i2.cpu().numpy(), want2)
            k += nb
        assert k == len(idx)
    # DistributedSampler semantics: the two ranks together cover every patch (padded by wrap-around to a multiple of 2)
    both = np.concatenate(seen)
    assert len(both) == 38 and set(both.tolist()) == set(range(n))
    f0 = DevicePatchFeed(b1, b2, B, "cuda:0", seed=5)
    assert sorted(f0.indices().tolist()) == list(range(n))
    f0.set_epoch(1)
    e1 = f0.indices().tolist()
    f0.set_epoch(2)
    assert e1 != f0.indices().tolist()


def test_patch_feed_errors():
    from mmif.feed import DevicePatchFeed
    b = np.zeros((2, 8, 8), np.uint8)
    with pytest.raises(ValueError, match="min-max"):
        DevicePatchFeed(b, b, 2, "cuda:0", norm="l2")
    with pytest.raises(TypeError):
        DevicePatchFeed(b.astype(np.float32), b.astype(np.float32), 2, "cuda:0")
    with pytest.raises(RuntimeError):
        DevicePatchFeed(b, b, 2, "cpu")
