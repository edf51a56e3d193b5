"""Argument combinations outside the HIP kernels run as stock torch ops and must reproduce the reference (round-4 verdict
"boundary holes"): SSIMLoss(use_padding=True) in its four modes (reference core/loss.py:42-49, :252-284), SSIM(size_average=False) and
reflect-padded 7-tap SSIM (:52-110, :163-185), MS_SSIM / MSW_SSIM with non-default arguments (:113-160, :211-237), the metric-side
calc_ssim with other windows / full=True / images smaller than the window (core/metric.py:316-364) and channel_pooling('nuclear')
(core/fusion.py:127-134).  Golden F16 was captured from the reference by tests/golden/make_golden.py; these run on CPU tensors (the
fallbacks are device-agnostic; the defaults -- the hot path -- still refuse CPU tensors)."""
import os

import numpy as np
import pytest
import torch

from oracle.fusion_oracle import closed_form_image, closed_form_signed

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
T = torch.from_numpy


@pytest.fixture(scope="module")
def f16():
    return np.load(os.path.join(G, "f16_stock_fallbacks.npz"))


def close(a, ref, tol=2e-5):
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    assert np.abs(ref).max() > 0
    err = np.abs(a - ref).max() / np.abs(ref).max()
    assert err <= tol, err


@pytest.mark.parametrize("mode,shape", [("ssim", (2, 1, 40, 52)), ("w-ssim", (2, 1, 40, 52)), ("msw-ssim", (2, 1, 40, 52)),
                                        ("ms-ssim", (1, 1, 192, 208))])
def test_ssim_loss_with_padding(f16, mode, shape):
    from core.loss import SSIMLoss
    a, b = T(closed_form_image(shape, 0.3)), T(closed_form_image(shape, 1.7))
    f = T(closed_form_image(shape, 2.9)).requires_grad_(True)
    loss = SSIMLoss(mode, use_padding=True, weight=0.7)(a, b, f)
    loss.backward()
    assert abs(loss.item() - float(f16[f"pad_{mode}__loss"])) <= 2e-6
    close(f.grad.numpy(), f16[f"pad_{mode}__grad"], 1e-4)


def test_ssim_module_maps_and_padded_window(f16):
    from core.loss import MS_SSIM, MSW_SSIM, SSIM
    shape = (2, 1, 40, 52)
    i1, i2, f = (T(closed_form_image(shape, p)) for p in (0.3, 1.7, 2.9))
    for tag, mod in (("ssim_maps", SSIM(11, 1.0, False, False)), ("ssim_pad7", SSIM(7, 1.0, True, True))):
        res = mod(i1, f)
        for k in ("ssim", "cs", "sigma"):
            assert tuple(res[k].shape) == tuple(f16[f"{tag}__{k}"].shape)
            close(res[k].numpy(), f16[f"{tag}__{k}"])
    big = (1, 1, 192, 208)
    close(MS_SSIM(11, 1.0, True, True)(T(closed_form_image(big, 0.3)), T(closed_form_image(big, 2.9))).numpy(), f16["msssim_pad"])
    assert abs(MSW_SSIM((11, 7, 3), 1.0, False, True)(i1, i2, f).item() - float(f16["mswssim_avg"])) <= 2e-6


def test_metric_calc_ssim_fallbacks(f16):
    from core.metric import calc_ssim
    shape = (2, 1, 40, 52)
    i1, f = T(closed_form_image(shape, 0.3)), T(closed_form_image(shape, 2.9))
    s, c = calc_ssim(i1, f, win_size=7, data_range=1.0, use_padding=True, full=True)
    close([s.item(), c.item()], f16["metric_w7_pad_full"])
    assert abs(calc_ssim(i1[:, :, :9, :20], f[:, :, :9, :20], data_range=1.0).item() - float(f16["metric_small"])) <= 2e-6
    close(calc_ssim(i1, f, data_range=1.0, size_average=False).numpy(), f16["metric_maps"])


def test_nuclear_channel_pooling(f16):
    from core.fusion import channel_pooling
    t = T(closed_form_signed((2, 6, 9, 13), 0.9)).requires_grad_(True)
    v = channel_pooling(t, 'nuclear')
    assert tuple(v.shape) == (1, 6, 1, 1)
    (v * torch.arange(1, 7, dtype=torch.float32).reshape(1, 6, 1, 1)).sum().backward()
    close(v.detach().numpy(), f16["nuclear__y"])
    close(t.grad.numpy(), f16["nuclear__dx"], 1e-4)
    with pytest.raises(ValueError):
        channel_pooling(t, 'bogus')
