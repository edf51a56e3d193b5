#!/usr/bin/env python3
"""Per-layer device time of one train step (HIP events around every tagged engine op, mmif/tensor.py PROFILE_TAGS = {"*"}).

    python tools/layer_times.py --model NestFuse --batch 4 --size 512 [--steps 10] [--dtype bf16]

Prints, per op tag, the average launch time, the conv shape, its algorithmic TFLOP/s (2·pixels·Cin·Cout·k²) and GB/s
(pixels·(Cin+Cout)·sizeof), sorted by time; the events serialise nothing (same stream) but add two event records per op.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multi-modal-image-fusion_amd"))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="PFNetv1")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--top", type=int, default=60)
    args = ap.parse_args()
    from core import model as M, zoo  # noqa: F401
    from core.loss import SSIMLoss, PixelLoss, GradLoss
    from mmif import engine as E, tensor as T
    from mmif.optim import FusedClipAdam
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    E.set_compute_dtype(args.dtype)
    torch.manual_seed(0)
    net = getattr(M, args.model)().to(dev)
    net.train()
    losses = (SSIMLoss(weight=1.0).to(dev), PixelLoss(weight=0.01).to(dev), GradLoss(weight=0.1).to(dev))
    opt = FusedClipAdam(net.parameters(), lr=1e-4, max_norm=5.0)
    a = torch.rand(args.batch, 1, args.size, args.size, device=dev)
    b = torch.rand(args.batch, 1, args.size, args.size, device=dev)

    def step():
        opt.zero_grad()
        f = net(a, b)
        l = losses[0](a, b, f) + losses[1](a, b, f, mode="max") + losses[2](a, b, f, mode="max")
        l.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    T.PROFILE_TAGS = {"*"}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    total = e0.elapsed_time(e1) / args.steps
    rows = []
    esz = 2 if args.dtype == "bf16" else 4
    for tag, evs in T.PROFILE_EVENTS.items():
        ms = sum(x.elapsed_time(y) for x, y in evs) / args.steps
        calls = len(evs) / args.steps
        sh = T.PROFILE_SHAPES.get(tag)
        tf = gb = float("nan")
        desc = ""
        if sh is not None and ms > 0:
            n, h, w, cin, cout, k = sh
            px = n * h * w
            mult = 2 if tag.endswith(":bwd") else 1
            tf = mult * calls * 2.0 * px * cin * cout * k * k / (ms * 1e-3) / 1e12
            gb = mult * calls * px * (cin + cout) * esz / (ms * 1e-3) / 1e9
            desc = f"{cin}->{cout} k{k} @{n}x{h}x{w}"
        rows.append((ms, tag, calls, desc, tf, gb))
    rows.sort(reverse=True)
    tagged = sum(r[0] for r in rows)
    print(f"# {args.model} B={args.batch} {args.size}^2 {args.dtype}: {total:.3f} ms/step ({args.batch / total * 1e3:.0f} pairs/s), tagged ops {tagged:.3f} ms")
    print(f"{'tag':34s} {'calls':>5s} {'ms/step':>8s} {'%':>5s}  {'shape':28s} {'TFLOP/s':>8s} {'GB/s':>7s}")
    for ms, tag, calls, desc, tf, gb in rows[:args.top]:
        print(f"{tag:34s} {calls:5.1f} {ms:8.3f} {100 * ms / total:5.1f}  {desc:28s} {tf:8.1f} {gb:7.0f}")


if __name__ == "__main__":
    main()
