#!/usr/bin/env python3
"""Where does an occasional slow default bench run (7.0-7.5 k instead of 8.6-8.9 k pairs/s) lose its ~80 ms?  The headline leg's loop with
host-side time stamps per step (no sync inside) and the garbage collector on / off (argv[1] = gc | nogc)."""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "multi-modal-image-fusion_amd")):
    sys.path.insert(0, p)
import torch
import core.model as M
from core.loss import FusionLoss, GradLoss, PixelLoss, SSIMLoss, unit_gradient
from mmif import engine as E
from mmif.optim import FusedClipAdam
mode = sys.argv[1] if len(sys.argv) > 1 else "gc"
dev = torch.device("cuda", 0)
E.set_compute_dtype("bf16")
torch.manual_seed(0)
model = M.PFNetv1().to(dev)
opt = FusedClipAdam(model.parameters(), lr=1e-4, betas=(0.9, 0.999), max_norm=5.0)
l_all = FusionLoss(SSIMLoss('ssim', weight=1.0), PixelLoss('l1', weight=0.01), GradLoss('l1', weight=0.1).to(dev), 'max', 'max')
g = torch.Generator(device="cpu").manual_seed(0)
a = torch.rand(32, 1, 256, 256, generator=g).to(dev); b = torch.rand(32, 1, 256, 256, generator=g).to(dev)
def step():
    opt.zero_grad(set_to_none=True)
    f = model(a, b)
    tot = l_all(a, b, f)
    opt.stage_scalars(l_all.values)
    tot.backward(unit_gradient(tot))
    opt.step(scalars=l_all.values)
for _ in range(30): step()
for rep in range(4):
    if mode == "nogc":
        gc.collect(); gc.disable()
    torch.cuda.synchronize()
    ts = [time.perf_counter()]
    for i in range(100):
        step(); ts.append(time.perf_counter())
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if mode == "nogc": gc.enable()
    d = [ts[i + 1] - ts[i] for i in range(100)]
    worst = sorted(range(100), key=lambda i: -d[i])[:3]
    print(f"{mode} rep {rep}: {32 * 100 / (t1 - ts[0]):.1f} pairs/s, {(t1 - ts[0]) * 10:.3f} ms/step; host per-step median {sorted(d)[50] * 1e3:.2f} ms, "
          f"worst {[(i, round(d[i] * 1e3, 1)) for i in worst]}, gc counts {gc.get_count()}", flush=True)
