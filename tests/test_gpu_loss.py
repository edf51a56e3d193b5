"""Fusion-loss kernels (SSIM / pixel / Sobel-gradient: value and d/dimgf in one launch) against the
reference's known answers (core/loss.py:388-423 recipe) and autograd gradients (golden F1 / F2)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import fusion_oracle as O
from gpu_util import G, close, tg
from test_oracle_golden import f2_inputs

pytestmark = pytest.mark.gpu


def _losses():
    from core.loss import GradLoss, PixelLoss, SSIMLoss
    return SSIMLoss('ssim', weight=1.0), PixelLoss('l1', weight=0.01), GradLoss('l1', weight=0.1).to("cuda:0")


def test_known_answer_losses():
    ref = json.load(open(os.path.join(G, "f1_loss_known_answer.json")))
    torch.manual_seed(0)
    x1 = torch.rand(2, 1, 256, 256).to("cuda:0")
    x2 = torch.rand(2, 1, 256, 256).to("cuda:0")
    y = torch.rand(2, 1, 256, 256).to("cuda:0")
    l1, l2, l3 = _losses()
    a = l1(x1, x2, y).item()
    assert abs(a - ref["ssim"]) < 5e-6 and abs(a - 0.9944541454) < 5e-6
    assert abs(l2(x1, x2, y).item() - ref["pixel_avg"]) < 1e-7
    assert abs(l3(x1, x2, y).item() - ref["grad_avg"]) < 1e-6
    b = l2(x1, x2, y, mode='max').item()
    c = l3(x1, x2, y, mode='max').item()
    assert abs(b - ref["pixel_max"]) < 1e-7 and abs(c - ref["grad_max"]) < 1e-6
    assert abs((a + b + c) - ref["total_max"]) < 6e-6
    from core.loss import SSIM
    s = SSIM(11, 1.0, False)(x1, y)['ssim'].cpu().numpy()
    np.testing.assert_allclose(s, ref["ssim_per_sample_x1_y"], atol=5e-7)
    assert l2(x1, x2, y, mode='nope') is None and l3(x1, x2, y, mode='nope') is None


@pytest.mark.parametrize("case", list("abcd"))
def test_loss_grads_vs_reference_autograd(case):
    ref = np.load(os.path.join(G, "f2_loss_grads.npz"))
    i1, i2, f = f2_inputs(case)
    l1, l2, l3 = _losses()
    t1, t2 = tg(i1), tg(i2)
    tf = tg(f).requires_grad_(True)
    a, b, c = l1(t1, t2, tf), l2(t1, t2, tf, mode='max'), l3(t1, t2, tf, mode='max')
    assert abs(a.item() - ref[f"{case}_l_ssim"]) < 5e-6
    assert abs(b.item() - ref[f"{case}_l_pixel"]) < 1e-7
    assert abs(c.item() - ref[f"{case}_l_grad"]) < 1e-6
    rt = 5e-3 if case == "c" else 3e-4  # case c: sigma_f^2 == 0, the reference's own fp32 gradient is ill-conditioned
    for l, key, tol in ((a, "g_ssim", rt), (b, "g_pixel", 1e-5), (c, "g_grad", 1e-5)):
        g, = torch.autograd.grad(l, tf, retain_graph=True)
        # (case c, g_grad: the Sobel gradient of a constant fused image is zero by design -- the engine's must be exactly zero too)
        close(g.cpu().numpy(), ref[f"{case}_{key}"], tol, key, allow_zero=(case == "c" and key == "g_grad"))
    g, = torch.autograd.grad(a + b + c, tf)
    close(g.cpu().numpy(), ref[f"{case}_g_total"], rt, "g_total")
    assert abs(l2(t1, t2, tf, mode='avg').item() - ref[f"{case}_l_pixel_avg"]) < 1e-7
    assert abs(l3(t1, t2, tf, mode='avg').item() - ref[f"{case}_l_grad_avg"]) < 1e-6


def test_loss_errors():
    from core.loss import NormLoss, SSIMLoss
    with pytest.raises(ValueError):
        SSIMLoss('nope')(torch.zeros(1, 1, 16, 16, device="cuda:0"), torch.zeros(1, 1, 16, 16, device="cuda:0"), torch.zeros(1, 1, 16, 16, device="cuda:0"))
    with pytest.raises(ValueError):
        NormLoss('l3')(torch.zeros(4, device="cuda:0"))
    with pytest.raises(RuntimeError):
        SSIMLoss()(torch.zeros(1, 1, 16, 16), torch.zeros(1, 1, 16, 16), torch.zeros(1, 1, 16, 16))  # CPU tensors: no silent fallback


def test_metric_calc_ssim_vs_golden():
    """core.metric.calc_ssim (test.py:49-52) on the fused HIP loss kernel vs the reference's values (golden F7)."""
    import json
    from core.metric import calc_ssim
    ref = json.load(open(os.path.join(G, "f7_metric_ssim.json")))
    for tag, r in ref.items():
        shape = tuple(r["shape"])
        a = tg(O.closed_form_image(shape, 0.37) * np.float32(r["scale"]))
        b = tg(O.closed_form_image(shape, 1.91) * np.float32(r["scale"]))
        s = calc_ssim(a, b, **r["kwargs"])
        assert s.dim() == 0
        assert abs(float(s) - r["ssim"]) <= 1e-4, (tag, float(s), r["ssim"])
        assert abs(float(calc_ssim(a, a, **r["kwargs"])) - r["ssim_self"]) <= 1e-5
    # argument combinations outside the HIP kernel run as stock torch ops (core/_stock.py; pinned to the reference by golden F16 in
    # tests/test_stock_fallbacks_cpu.py): full=True returns (ssim, cs) and the ssim agrees with the kernel's
    s_full, cs_full = calc_ssim(a, b, full=True, **r["kwargs"])
    assert abs(float(s_full) - float(s)) <= 1e-5 and cs_full.dim() == 0
    with pytest.raises(RuntimeError):
        calc_ssim(a.repeat(1, 3, 1, 1), b.repeat(1, 3, 1, 1))


F9_SSIM = [("w-ssim", (2, 1, 40, 52)), ("w-ssim", (3, 1, 33, 47)), ("msw-ssim", (2, 1, 40, 52)), ("msw-ssim", (1, 1, 33, 47)),
           ("ms-ssim", (1, 1, 192, 208)), ("ms-ssim", (2, 1, 193, 211))]


@pytest.mark.parametrize("mode,shape", F9_SSIM, ids=[f"{m}-{s[0]}x{s[2]}x{s[3]}" for m, s in F9_SSIM])
def test_ssim_modes_vs_golden(mode, shape):
    """SSIMLoss 'w-ssim' / 'ms-ssim' / 'msw-ssim' on csrc/loss_modes.hip: value + d/dimgf vs the reference's autograd (golden F9)."""
    from core.loss import SSIMLoss
    ref = np.load(os.path.join(G, "f9_ssim_modes.npz"))
    tag = f"{mode}_{shape[0]}x{shape[2]}x{shape[3]}"
    i1, i2 = tg(O.closed_form_image(shape, 0.3)), tg(O.closed_form_image(shape, 1.7))
    f = tg(O.closed_form_image(shape, 2.9)).requires_grad_(True)
    loss = SSIMLoss(mode, weight=0.7)(i1, i2, f)
    loss.backward()
    assert abs(float(loss.detach()) - float(ref[tag + "__loss"])) <= 5e-5
    close(f.grad.cpu().numpy(), ref[tag + "__grad"], 5e-4, tag)
    with torch.no_grad():   # value-only call (no gradient buffers)
        assert abs(float(SSIMLoss(mode, weight=0.7)(i1, i2, f)) - float(ref[tag + "__loss"])) <= 5e-5


@pytest.mark.parametrize("mode", ["w-ssim", "msw-ssim"])
def test_ssim_modes_flat_source_clamps(mode):
    from core.loss import SSIMLoss
    ref = np.load(os.path.join(G, "f9_ssim_modes.npz"))
    shape = (2, 1, 24, 24)
    i1, i2 = tg(np.full(shape, 0.4, np.float32)), tg(O.closed_form_image(shape, 1.1))
    f = tg(O.closed_form_image(shape, 2.2)).requires_grad_(True)
    loss = SSIMLoss(mode)(i1, i2, f)
    loss.backward()
    assert abs(float(loss.detach()) - float(ref[f"{mode}_flat__loss"])) <= 5e-5
    close(f.grad.cpu().numpy(), ref[f"{mode}_flat__grad"], 5e-4, mode)


@pytest.mark.parametrize("mode", ["l1", "l2"])
def test_tv_loss_vs_golden(mode):
    from core.loss import TVLoss
    ref = np.load(os.path.join(G, "f9_ssim_modes.npz"))
    x = tg(O.closed_form_image((2, 1, 21, 34), 0.77)).requires_grad_(True)
    loss = TVLoss(mode, weight=0.3)(x)
    loss.backward()
    assert abs(float(loss.detach()) - float(ref[f"tv_{mode}__loss"])) <= 1e-6
    close(x.grad.cpu().numpy(), ref[f"tv_{mode}__grad"], 1e-5, mode)


def test_ssim_mode_errors():
    from core.loss import SSIMLoss, TVLoss
    z = tg(np.zeros((1, 1, 32, 32), np.float32))
    with pytest.raises(ValueError, match="only supported"):
        SSIMLoss("psnr")(z, z, z)
    with pytest.raises(ValueError, match="only supported"):
        TVLoss("l3")(z)
    with pytest.raises(Exception, match="161x161"):
        SSIMLoss("ms-ssim")(z, z, z)


def test_ms_ssim_and_msw_ssim_classes():
    """core.loss.MS_SSIM / MSW_SSIM (reference __all__): values vs the oracle / golden F9, gradient w.r.t. the fused image."""
    from core.loss import MS_SSIM, MSW_SSIM
    ref = np.load(os.path.join(G, "f9_ssim_modes.npz"))
    shape = (2, 1, 40, 52)
    i1n, i2n, fn = O.closed_form_image(shape, 0.3), O.closed_form_image(shape, 1.7), O.closed_form_image(shape, 2.9)
    f = tg(fn).requires_grad_(True)
    v = MSW_SSIM()(tg(i1n), tg(i2n), f)
    want = 1.0 - float(ref["msw-ssim_2x40x52__loss"]) / 0.7        # golden loss was taken with weight 0.7
    assert abs(float(v.detach()) - want) <= 1e-4
    (0.7 * (1.0 - v)).backward()
    close(f.grad.cpu().numpy(), ref["msw-ssim_2x40x52__grad"], 5e-4, "msw grad")
    shape = (2, 1, 193, 211)
    an, bn = O.closed_form_image(shape, 0.3), O.closed_form_image(shape, 2.9)
    ms = MS_SSIM()(tg(an), tg(bn)).cpu().numpy()
    for i in range(2):       # oracle: loss(a, a, b) = 1 - ms(a, b) for one sample
        l, _ = O.ssim_mode_loss(an[i:i + 1], an[i:i + 1], bn[i:i + 1], "ms-ssim", need_grad=False)
        assert abs(ms[i] - (1.0 - float(l))) <= 1e-4, (i, ms[i], 1.0 - float(l))
    # other window lists: stock torch ops with the reference's result (core/_stock.py, golden F16) -- same value as the kernel for the default list
    an, bn, fn_ = (O.closed_form_image((2, 1, 40, 52), p) for p in (0.3, 1.7, 2.9))
    v_k = MSW_SSIM()(tg(an), tg(bn), tg(fn_)).item()
    v_s = MSW_SSIM(win_sizes=(11, 9, 7, 5, 3), use_padding=False, size_average=False)._stock
    assert v_s is False
    import core._stock as S
    assert abs(S.mswssim(tg(an), tg(bn), tg(fn_)).item() - v_k) <= 2e-5
    assert MSW_SSIM(win_sizes=(11, 7))(tg(an), tg(bn), tg(fn_)).dim() == 0


@pytest.mark.parametrize("win", [11, 7, 3])
def test_ssim_module_terms_vs_oracle(win):
    """core.loss.SSIM: {'ssim', 'cs', 'sigma'} per-sample means (reference core/loss.py:163-185) vs the oracle's maps."""
    from core.loss import SSIM
    shape = (3, 1, 33, 47)
    an, bn = O.closed_form_image(shape, 0.3), O.closed_form_image(shape, 2.9)
    out = SSIM(win)(tg(an), tg(bn))
    t = O.ssim_full_terms(an, bn, O.create_window(win))
    for key, ref in (("ssim", t["S"]), ("cs", t["cs"]), ("sigma", t["sigma"])):
        want = ref.mean(axis=(1, 2, 3))
        got = out[key].detach().cpu().numpy()
        assert got.shape == (3,) and np.abs(got - want).max() <= 1e-4 * max(1.0, np.abs(want).max()), (key, got, want)
    if win == 11:
        f = tg(bn).requires_grad_(True)
        s = SSIM()(tg(an), f)['ssim'].sum()
        s.backward()
        assert float(f.grad.abs().max()) > 0


@pytest.mark.parametrize("pm,gm,shape", [("max", "max", (2, 1, 64, 80)), ("avg", "max", (3, 1, 33, 47)), ("max", "avg", (1, 1, 256, 256))])
def test_fusion_loss_equals_the_three_modules(pm, gm, shape):
    """core.loss.FusionLoss (ONE device call for the three terms of train.py:64-69) against the three modules + torch's additions:
    the SSIM and Sobel terms bit for bit (same kernels, same partial sums), the pixel term to 1e-6 (other partials), the total to one fp32 rounding, d(total)/d(imgf) to the rounding of a
    different summation order of the three contributions; and the known answers of golden F1 through the fused call."""
    from core.loss import FusionLoss
    torch.manual_seed(5)
    x1, x2 = torch.rand(shape).to("cuda:0"), torch.rand(shape).to("cuda:0")
    l1, l2, l3 = _losses()
    y = torch.rand(shape).to("cuda:0").requires_grad_(True)
    a, b, c = l1(x1, x2, y), l2(x1, x2, y, mode=pm), l3(x1, x2, y, mode=gm)
    (a + b + c).backward()
    g_sep = y.grad.clone()
    y.grad = None
    fl = FusionLoss(l1, l2, l3, pm, gm)
    tot = fl(x1, x2, y)
    tot.backward()
    v = fl.values.cpu().numpy()
    # (round 6: the pixel term rides in the Sobel kernel of the fused call -- the same arithmetic per pixel, its value summed from that kernel's
    # 16 x 16-tile partials instead of mmif_pixel_loss's grid-stride ones: equal to fp32 summation order; SSIM and Sobel terms: same partials)
    assert v[1] == a.item() and abs(v[2] - b.item()) <= 1e-6 * abs(b.item()) and v[3] == c.item()
    assert tot.item() == v[0] and abs(v[0] - (a + b + c).item()) <= 1.2e-7 * abs(v[0])
    close(y.grad.cpu().numpy(), g_sep.cpu().numpy(), 2e-6, "d total / d imgf")
    assert not fl.values.requires_grad and tot.requires_grad
    # scaled upstream gradient, and no gradient requested
    y.grad = None
    (3.0 * fl(x1, x2, y)).backward()
    close(y.grad.cpu().numpy(), 3.0 * g_sep.cpu().numpy(), 2e-6, "3 x")
    with torch.no_grad():
        assert fl(x1, x2, y).item() == v[0]
    with pytest.raises(ValueError):
        FusionLoss(l1, l2, l3, 'nope', 'max')


def test_fusion_loss_known_answers():
    ref = json.load(open(os.path.join(G, "f1_loss_known_answer.json")))
    from core.loss import FusionLoss
    torch.manual_seed(0)
    x1 = torch.rand(2, 1, 256, 256).to("cuda:0")
    x2 = torch.rand(2, 1, 256, 256).to("cuda:0")
    y = torch.rand(2, 1, 256, 256).to("cuda:0")
    fl = FusionLoss(*_losses(), 'max', 'max')
    tot = fl(x1, x2, y).item()
    v = fl.values.cpu().numpy()
    assert abs(v[1] - ref["ssim"]) < 5e-6 and abs(v[2] - ref["pixel_max"]) < 1e-7 and abs(v[3] - ref["grad_max"]) < 1e-6
    assert abs(tot - ref["total_max"]) < 6e-6


def test_unit_gradient_backward_equals_plain_backward():
    """core.loss.unit_gradient: total.backward(unit_gradient(total)) is total.backward() -- FusionLoss hands its stored gradient on
    without the multiply-by-one pass when it recognises the cached ones tensor, autograd launches no fill kernel for the root -- and a
    different upstream gradient still scales: bit-identical d(total)/d(imgf) in the first two cases, exactly 2.5 x in the third."""
    from core.loss import FusionLoss, GradLoss, PixelLoss, SSIMLoss, unit_gradient
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(3)
    a, b = torch.rand(2, 1, 40, 56, generator=g).to(dev), torch.rand(2, 1, 40, 56, generator=g).to(dev)
    fl = FusionLoss(SSIMLoss('ssim', weight=1.0), PixelLoss('l1', weight=0.01), GradLoss('l1', weight=0.1).to(dev), 'max', 'max')
    grads = []
    for kind in ("plain", "unit", "scaled"):
        f = torch.rand(2, 1, 40, 56, generator=torch.Generator().manual_seed(4)).to(dev).requires_grad_(True)
        tot = fl(a, b, f)
        if kind == "plain":
            tot.backward()
        elif kind == "unit":
            tot.backward(unit_gradient(tot))
        else:
            tot.backward(torch.full_like(tot, 2.5))
        grads.append(f.grad.clone())
    assert float(grads[0].abs().max()) > 0
    assert torch.equal(grads[0], grads[1])
    assert torch.equal(grads[2], grads[0] * 2.5)
    assert float(unit_gradient(tot)) == 1.0
