#!/usr/bin/env python3
"""bench.py's headline loop with one event at the end of every step (GPU-side per-step durations) and host stamps: where do slow runs lose time?
argv[1]: 'sample' = set PROFILE_TAGS on every 20th step as bench.py does, 'plain' = never."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "multi-modal-image-fusion_amd")):
    sys.path.insert(0, p)
import torch
import core.model as M
from core.loss import FusionLoss, GradLoss, PixelLoss, SSIMLoss, unit_gradient
from mmif import engine as E
from mmif import tensor as T
from mmif.optim import FusedClipAdam
mode = sys.argv[1] if len(sys.argv) > 1 else "sample"
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
    return tot
for _ in range(30): step()
tags = {"decode.0:fwd", "decode.0:dgrad", "decode.0:wgrad", "encode:fwd", "encode:bwd"}
ev = [torch.cuda.Event(enable_timing=True) for _ in range(101)]
for e in ev: e.record()
T.prealloc_events(tags, 8)
torch.cuda.synchronize()
for rep in range(3):
    T.PROFILE_EVENTS.clear()
    torch.cuda.synchronize()
    ts = [time.perf_counter()]
    ev[0].record()
    for i in range(100):
        T.PROFILE_TAGS = tags if (mode == "sample" and i % 20 == 0) else set()
        step(); ev[i + 1].record(); ts.append(time.perf_counter())
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    T.PROFILE_TAGS = set()
    gd = [ev[i].elapsed_time(ev[i + 1]) for i in range(100)]
    hd = [(ts[i + 1] - ts[i]) * 1e3 for i in range(100)]
    sg = sorted(gd)
    big = [(i, round(gd[i], 2)) for i in range(100) if gd[i] > 1.3 * sg[50]]
    print(f"{mode} rep {rep}: {3200 / (t1 - ts[0]):.1f} pairs/s; GPU step median {sg[50]:.3f} p90 {sg[90]:.3f} max {sg[-1]:.3f} ms, sum {sum(gd):.1f} ms; slow steps {big[:12]}; host worst {sorted([(round(h, 1), i) for i, h in enumerate(hd)])[-3:]}", flush=True)
    for evs in T.PROFILE_EVENTS.values():
        for pr in evs: T.PROFILE_EVENT_POOL.setdefault("decode.0:fwd", []).append(pr)
