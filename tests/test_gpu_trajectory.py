"""Does the bf16 headline path TRAIN like the reference beyond three steps?  (round-5 verdict, "missing" 6 / task 5.)

The reference's loop is epochs of train.py:54-78 (zero_grad, forward, SSIM + max-pixel + max-gradient losses, backward,
clip_grad_norm_(5), Adam 1e-4; train.py:319-321); golden F6 pins three steps of it at 64 x 64.  Here: PFNetv1, B = 4, 64 x 64, 300
steps over a fixed cycle of 8 synthetic batches, from the reference's own initialisation (kaiming-normal, seed 0), run three times --
oracle/torch_cpu_step.py (stock torch CPU ops, fp32: the reference's arithmetic), the HIP engine on its fp32 parity path and the HIP
engine on bf16 storage (the headline path) -- and compared as TRAJECTORIES:

  fp32 path   the total loss of each of the first 20 steps within 1e-3 (relative) of the reference's, the cycle-smoothed loss within
              1 % throughout (after some dozens of Adam steps two fp32 implementations drift apart by themselves: Adam's first steps
              are sign-like), the final smoothed loss within 0.5 %
  bf16 path   the cycle-smoothed loss (mean over 8 consecutive steps = one pass over the batches) within 2 % of the reference's
              throughout, the final smoothed loss within 1 %
  both        the loss must have FALLEN (the run trains), and the relative L2 distance of the final weights to the reference's is
              printed (DESIGN section 5 quotes it) and bounded loosely
"""
import time

import numpy as np
import pytest
import torch

from oracle import torch_cpu_step as TC
from gpu_util import dtype_ctx

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
STEPS, NB, B, S = 300, 8, 4, 64


def _batches():
    g = torch.Generator().manual_seed(2024)
    out = []
    for k in range(NB):
        # smooth structure + noise, in [0, 1): two different "modalities" per pair (a pure-noise pair has no SSIM structure to learn)
        y, x = torch.meshgrid(torch.arange(S, dtype=torch.float32), torch.arange(S, dtype=torch.float32), indexing="ij")
        base1 = 0.5 + 0.25 * torch.sin(0.13 * x + 0.7 * k) * torch.cos(0.09 * y + 0.3 * k)
        base2 = 0.5 + 0.25 * torch.cos(0.07 * x - 0.5 * k) * torch.sin(0.11 * y + 0.9 * k)
        i1 = (base1[None, None] + 0.2 * (torch.rand(B, 1, S, S, generator=g) - 0.5)).clamp(0, 0.999)
        i2 = (base2[None, None] + 0.2 * (torch.rand(B, 1, S, S, generator=g) - 0.5)).clamp(0, 0.999)
        out.append((i1.contiguous(), i2.contiguous()))
    return out


def _cpu_run(batches):
    # (8 intra-op threads: on the 256-CPU GPU box torch's default pool made this 64 x 64 run take 275 s instead of ~35)
    prev = torch.get_num_threads()
    torch.set_num_threads(min(8, prev))
    try:
        return _cpu_run_(batches)
    finally:
        torch.set_num_threads(prev)


def _cpu_run_(batches):
    m = TC.TorchCpuModel("PFNetv1")
    P = m.init_params(0)
    opt = TC.make_optimizer(P)
    losses = []
    t0 = time.time()
    for s in range(STEPS):
        i1, i2 = batches[s % NB]
        losses.append(TC.train_step(m, P, opt, i1, i2)["losses"][3])
    return np.array(losses), {k: v.detach().clone() for k, v in P.items()}, time.time() - t0


def _hip_run(batches, dtype):
    import core.model as M
    from core.loss import FusionLoss, GradLoss, PixelLoss, SSIMLoss, unit_gradient
    from mmif.optim import FusedClipAdam
    with dtype_ctx(dtype):
        m = M.PFNetv1()
        P0 = TC.TorchCpuModel("PFNetv1").init_params(0)
        assert list(P0) == list(m.state_dict()), "state_dict keys differ from the reference's"
        m.load_state_dict({k: v.detach().clone() for k, v in P0.items()})
        m = m.to(DEV)
        opt = FusedClipAdam(m.parameters(), lr=1e-4, betas=(0.9, 0.999), max_norm=5.0)
        l_all = FusionLoss(SSIMLoss('ssim', weight=1.0), PixelLoss('l1', weight=0.01), GradLoss('l1', weight=0.1).to(DEV), 'max', 'max')
        dev_b = [(a.to(DEV), b.to(DEV)) for a, b in batches]
        vals = []
        for s in range(STEPS):
            a, b = dev_b[s % NB]
            opt.zero_grad(set_to_none=True)
            f = m(a, b)
            tot = l_all(a, b, f)
            tot.backward(unit_gradient(tot))
            opt.step()
            vals.append(tot.detach())
        torch.cuda.synchronize()
        losses = torch.stack(vals).double().cpu().numpy()
        W = {k: p.detach().cpu().clone() for k, p in m.named_parameters()}
    return losses, W


def _smooth(x):
    c = np.cumsum(np.concatenate(([0.0], x)))
    return (c[NB:] - c[:-NB]) / NB        # mean over one pass of the batch cycle


def _wdist(W, R):
    num = sum(float((W[k].double() - R[k].double()).pow(2).sum()) for k in R)
    den = sum(float(R[k].double().pow(2).sum()) for k in R)
    return (num / den) ** 0.5


def test_pfnetv1_300_steps_bf16_and_fp32_train_like_the_torch_cpu_reference():
    batches = _batches()
    ref, Wr, cpu_s = _cpu_run(batches)
    W0 = {k: v.detach().clone() for k, v in TC.TorchCpuModel("PFNetv1").init_params(0).items()}
    moved = _wdist(Wr, W0)
    sr = _smooth(ref)
    print(f"reference (torch CPU, {cpu_s:.1f} s): loss {ref[0]:.6f} -> smoothed {sr[0]:.6f} -> {sr[-1]:.6f} over {STEPS} steps; the weights moved {moved:.3e} (relative L2) from their initial values")
    assert sr[-1] < 0.9 * sr[0], "the reference run does not train: the test would be vacuous"
    rep = {}
    for dtype in ("fp32", "bf16"):
        los, W = _hip_run(batches, dtype)
        assert np.isfinite(los).all()
        sm = _smooth(los)
        per_step = np.abs(los - ref) / np.abs(ref)
        smoothed = np.abs(sm - sr) / np.abs(sr)
        wd = _wdist(W, Wr)
        rep[dtype] = (per_step, smoothed, wd)
        print(f"{dtype}: per-step loss rel. err first 20 steps max {per_step[:20].max():.2e}, all steps max {per_step.max():.2e}; smoothed (8-step) max {smoothed.max():.2e}, "
              f"final smoothed {sm[-1]:.6f} vs {sr[-1]:.6f} ({smoothed[-1]:.2e}); final weights {wd:.3e} relative L2 from the reference's "
              f"(= {wd / moved:.3f} of the distance the reference's weights travelled)")
        assert sm[-1] < 0.9 * sm[0]
    per_step, smoothed, wd = rep["fp32"]
    assert per_step[:20].max() <= 1e-3
    assert smoothed.max() <= 1e-2 and smoothed[-1] <= 5e-3
    assert wd <= 0.5 * moved
    per_step, smoothed, wd = rep["bf16"]
    assert smoothed.max() <= 2e-2 and smoothed[-1] <= 1e-2
    assert wd <= 1.0 * moved
