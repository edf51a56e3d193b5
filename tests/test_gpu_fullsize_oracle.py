"""DIRECT oracle comparisons at BASELINE.json's sizes (round-4 verdict, "what's weak" 1: the full-size evidence was property-only).

The torch-CPU restatement (oracle/torch_cpu_step.py -- pinned to the reference's goldens F5 / F6 by tests/test_torch_cpu_step.py, incl.
the LIVE NestFuse / RFN-Nest cases) runs these sizes in seconds on the GPU box's host cores, so the tiling, the XCD walk, the
persistent block schedules and the 64-bit index paths that only exist at full size are compared against the reference's arithmetic
itself:

  configs 2 / 3   PFNetv1 and DenseFuse, ONE whole train step at B = 4, 256 x 256 (reference train.py:61-75): fused image, the three
                  loss terms, the pre-clip gradient norm, every parameter gradient, the weights after clip + Adam -- fp32 path at the
                  north star's 1e-3, bf16 path against the same fp32 run within the documented bf16-storage bars
  config 4        NestFuse / RFN-Nest, 1 x 512 x 512 (core/model.py:319-384): fused image + every parameter gradient of the train
                  step's loss, fp32 and bf16
  config 5        PFNetv1, the whole 1 x 1024 x 1224 frame (test.py:41-48), forward, fp32 and bf16

Bars.  fp32: 1e-3 of max|reference| (north star) on images, gradients (relative L2 per parameter as well) and losses.  bf16: the
fused image within 3e-2 of max|reference| for the PFNet family (input rounding through ten layers, tests/test_gpu_models.py) and 9e-2
for the nested nets (their last layer sums 64 cancelling terms); losses within 1 %, the gradient
norm within 3 %; per-parameter gradients by relative L2 (median and worst stated per test) and cosine.
"""
import time

import numpy as np
import pytest
import torch

from oracle import fusion_oracle as O
from oracle import torch_cpu_step as TC
from gpu_util import dtype_ctx, load_closed_form, load_live

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cpu_params(model, name, seed=1):
    P = model.init_params(0)
    with torch.no_grad():
        for i, (k, v) in enumerate(P.items()):
            src = O.live_param(name, i, k, tuple(v.shape)) if name in O.LIVE_PARAMS else O.closed_form_param(i, k, tuple(v.shape), seed)
            v.copy_(torch.from_numpy(src))
    return P


def _hip_model(name, seed=1):
    import core.model as M
    m = getattr(M, name)()
    return (load_live(m, name) if name in O.LIVE_PARAMS else load_closed_form(m, seed)).to(DEV)


def _rel_max(a, b):
    a, b = a.double(), b.double()
    assert float(b.abs().max()) > 0, "all-zero reference"
    return float((a - b).abs().max() / b.abs().max())


def _rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    assert float(b.norm()) > 0, "all-zero reference"
    return float((a - b).norm() / b.norm())


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm()))


def _hip_step(name, dtype, i1, i2, step=True):
    """forward + 3 losses + backward (+ clip 5 + Adam 1e-4) on the HIP engine; everything comes back on the CPU"""
    from core.loss import GradLoss, PixelLoss, SSIMLoss
    from mmif.optim import FusedClipAdam
    with dtype_ctx(dtype):
        m = _hip_model(name)
        opt = FusedClipAdam(m.parameters(), lr=1e-4, max_norm=5.0)
        a, b = i1.to(DEV), i2.to(DEV)
        f = m(a, b)
        l1 = SSIMLoss('ssim', weight=1.0)(a, b, f)
        l2 = PixelLoss('l1', weight=0.01)(a, b, f, mode='max')
        l3 = GradLoss('l1', weight=0.1).to(DEV)(a, b, f, mode='max')
        tot = l1 + l2 + l3
        tot.backward()
        grads = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
        out = dict(imgf=f.detach().cpu(), losses=[float(l1), float(l2), float(l3), float(tot)], grads=grads)
        if step:
            opt.step()
            out["grad_norm"] = float(opt.grad_norm.item())
            out["weights"] = {k: p.detach().cpu().clone() for k, p in m.named_parameters()}
        torch.cuda.synchronize()
    return out


def _cpu_step(name, i1, i2, step=True):
    m = TC.TorchCpuModel(name)
    P = _cpu_params(m, name)
    t0 = time.time()
    if step:
        r = TC.train_step(m, P, TC.make_optimizer(P), i1, i2)
        # train_step returns the gradients AFTER clip_grad_norm_ scaled them in place (as the reference's are when Adam reads them);
        # the HIP engine's .grad holds the raw gradients and clips inside the fused Adam kernel: undo the scale for the comparison
        coef = min(1.0, 5.0 / (r["grad_norm"] + 1e-6))
        r["grads"] = {k: v / coef for k, v in r["grads"].items()}
        r["weights"] = {k: v.detach().clone() for k, v in P.items()}
    else:
        f = m.forward(P, i1, i2)
        l1, l2, l3, tot = TC.fusion_losses(i1, i2, f)
        tot.backward()
        r = dict(imgf=f.detach(), losses=[float(l1), float(l2), float(l3), float(tot)], grads={k: v.grad.detach().clone() for k, v in P.items()})
    r["cpu_seconds"] = time.time() - t0
    return r


def _grad_report(hip, ref, what):
    rows = []
    for k, g in ref["grads"].items():
        assert float(g.abs().max()) > 0, f"{k}: the reference gradient is all-zero"
        rows.append((k, _rel_l2(hip["grads"][k], g), _rel_max(hip["grads"][k], g), _cos(hip["grads"][k], g)))
    l2s = np.array([r[1] for r in rows])
    worst = max(rows, key=lambda r: r[1])
    print(f"{what}: per-parameter gradient rel-L2 median {np.median(l2s):.2e}, worst {worst[1]:.2e} ({worst[0]}), worst max-rel "
          f"{max(r[2] for r in rows):.2e}, min cosine {min(r[3] for r in rows):.6f}; torch-CPU reference took {ref['cpu_seconds']:.1f} s")
    return rows


def _check_adam_weights(hip, ref, lr=1e-4):
    """Weights after clip + Adam.  The first Adam step moves every element by lr * g / (|g| + eps) = +-lr: where |g| is within rounding
    of zero the two implementations may disagree on the SIGN (2 lr apart) although both are right; such elements must be rare and sit
    on reference gradients that are tiny next to the layer's."""
    bad = total = 0
    for k, w in ref["weights"].items():
        d = (hip["weights"][k].double() - w.double()).abs()
        off = d > 0.05 * lr
        total += d.numel()
        bad += int(off.sum())
        assert float(d.max()) <= 2.0 * lr * 1.01, (k, float(d.max()))           # never more than a sign flip of one step
        if bool(off.any()):
            g = ref["grads"][k].double().abs()
            assert float(g[off].max()) <= 2e-2 * float(g.max()), (k, float(g[off].max()), float(g.max()))
    frac = bad / total
    print(f"post-Adam weights: {bad} of {total} elements ({frac:.2e}) differ by a first-step sign flip on a near-zero gradient")
    assert frac <= 2e-3, frac


@pytest.mark.parametrize("name", ["PFNetv1", "DenseFuse"])
def test_train_step_b4_256_vs_torch_cpu_reference(name):
    """configs 2 / 3 at their real spatial size: one whole train step, B = 4, 256 x 256"""
    g = torch.Generator().manual_seed(11)
    i1, i2 = torch.rand(4, 1, 256, 256, generator=g), torch.rand(4, 1, 256, 256, generator=g)
    ref = _cpu_step(name, i1, i2)
    if ref["cpu_seconds"] > 60:
        print(f"note: the torch-CPU reference took {ref['cpu_seconds']:.1f} s on this box")
    # ---- fp32 path: the north star's 1e-3
    hip = _hip_step(name, "fp32", i1, i2)
    e_img = _rel_max(hip["imgf"], ref["imgf"])
    print(f"{name} fp32: fused image {e_img:.2e} of max|ref|, losses {hip['losses']} vs {ref['losses']}, grad norm {hip['grad_norm']:.6f} vs {ref['grad_norm']:.6f}")
    assert e_img <= 1e-3
    np.testing.assert_allclose(hip["losses"], ref["losses"], rtol=1e-3, atol=1e-6)
    assert abs(hip["grad_norm"] - ref["grad_norm"]) <= 1e-3 * ref["grad_norm"]
    for k, l2, mx, cs in _grad_report(hip, ref, f"{name} fp32"):
        assert l2 <= 1e-3 and mx <= 1e-3, (k, l2, mx)
    _check_adam_weights(hip, ref)
    # ---- bf16 storage (the throughput path) against the SAME fp32 reference run
    hip = _hip_step(name, "bf16", i1, i2)
    e_img = _rel_max(hip["imgf"], ref["imgf"])
    print(f"{name} bf16: fused image {e_img:.2e} of max|ref|, losses {hip['losses']} vs {ref['losses']}, grad norm {hip['grad_norm']:.6f} vs {ref['grad_norm']:.6f}")
    assert e_img <= 3e-2
    np.testing.assert_allclose(hip["losses"], ref["losses"], rtol=1e-2, atol=1e-5)
    assert abs(hip["grad_norm"] - ref["grad_norm"]) <= 3e-2 * ref["grad_norm"]
    rows = _grad_report(hip, ref, f"{name} bf16")
    assert np.median([r[1] for r in rows]) <= 3e-2
    for k, l2, mx, cs in rows:
        assert l2 <= 1e-1 and cs >= 0.995, (k, l2, cs)


@pytest.mark.parametrize("name", ["NestFuse", "RFNNest"])
def test_nest_1x512x512_vs_torch_cpu_reference(name):
    """config 4 at its real size: fused image + every parameter gradient of the train step's loss, 1 x 512 x 512, live parameters"""
    g = torch.Generator().manual_seed(12)
    i1, i2 = torch.rand(1, 1, 512, 512, generator=g), torch.rand(1, 1, 512, 512, generator=g)
    ref = _cpu_step(name, i1, i2, step=False)
    if ref["cpu_seconds"] > 60:
        print(f"note: the torch-CPU reference took {ref['cpu_seconds']:.1f} s on this box")
    frac = float((ref["imgf"] > 0).float().mean())
    assert 0.2 <= frac <= 0.8, f"{frac:.3f} of the reference's fused pixels pass the final ReLU: the case would be vacuous"
    hip = _hip_step(name, "fp32", i1, i2, step=False)
    e_img = _rel_max(hip["imgf"], ref["imgf"])
    print(f"{name} fp32: fused image {e_img:.2e} of max|ref| ({frac:.2f} of the pixels live), losses {hip['losses']} vs {ref['losses']}")
    assert e_img <= 1e-3
    np.testing.assert_allclose(hip["losses"], ref["losses"], rtol=1e-3, atol=1e-6)
    for k, l2, mx, cs in _grad_report(hip, ref, f"{name} fp32"):
        assert l2 <= 1e-3 and mx <= 2e-3, (k, l2, mx)
    hip = _hip_step(name, "bf16", i1, i2, step=False)
    e_img = _rel_max(hip["imgf"], ref["imgf"])
    print(f"{name} bf16: fused image {e_img:.2e} of max|ref|, losses {hip['losses']} vs {ref['losses']}")
    assert e_img <= 9e-2
    np.testing.assert_allclose(hip["losses"], ref["losses"], rtol=2e-2, atol=1e-5)
    # bf16 storage through ~30 layers against an fp32 run: measured on the first round-5 box, NestFuse median 7.0e-2 / worst 1.04e-1
    # (cosine 0.9955) -- the same order as the fused image's 5.6e-2, i.e. the storage rounding itself (the kernels add none: every one of
    # them equals the oracle on identical bf16 operands, tests/test_gpu_nest.py, and the bf16 engine sits 1e-3 .. 5e-3 from the
    # bf16-storage emulation at 36 x 44)
    rows = _grad_report(hip, ref, f"{name} bf16")
    assert np.median([r[1] for r in rows]) <= 1e-1
    for k, l2, mx, cs in rows:
        assert l2 <= 2e-1 and cs >= 0.98, (k, l2, cs)


def test_pfnetv1_fullres_frame_vs_torch_cpu_reference():
    """config 5: the WHOLE 1 x 1024 x 1224 frame (no crop), forward, against the torch-CPU reference"""
    H, W = 1024, 1224
    g = torch.Generator().manual_seed(5)
    i1, i2 = torch.rand(1, 1, H, W, generator=g), torch.rand(1, 1, H, W, generator=g)
    m = TC.TorchCpuModel("PFNetv1")
    P = _cpu_params(m, "PFNetv1")
    t0 = time.time()
    with torch.no_grad():
        ref = m.forward(P, i1, i2)
    cpu_s = time.time() - t0
    if cpu_s > 60:
        print(f"note: the torch-CPU reference took {cpu_s:.1f} s on this box")
    for dtype, tol in (("fp32", 1e-3), ("bf16", 3e-2)):
        with dtype_ctx(dtype), torch.no_grad():
            y = _hip_model("PFNetv1")(i1.to(DEV), i2.to(DEV)).cpu()
        e = _rel_max(y, ref)
        print(f"PFNetv1 1x{H}x{W} {dtype}: fused frame {e:.2e} of max|ref| (torch-CPU reference {cpu_s:.1f} s)")
        assert y.shape == ref.shape and e <= tol, (dtype, e)
