#!/usr/bin/env python3
"""Stand-alone bandwidth of the NestFuse glue kernels (csrc/nest.hip) at one shape:  python tools/bench_elem.py [n c h w]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-modal-image-fusion_amd")]
import torch
from mmif import tensor as T
n, c, h, w = [int(a) for a in (sys.argv[1:5] + ["4", "64", "512", "512"][len(sys.argv) - 1:])]
dev = "cuda:0"
bf = torch.bfloat16
a = T.BT.alloc(n, c, h, w, bf, dev); a.buf.normal_()
b = T.BT.alloc(n, c, h, w, bf, dev); b.buf.normal_()
o = T.BT.alloc(n, c, h, w, bf, dev)
g = T.BT.alloc(n, c, h, w, bf, dev, halo=1, zero=True); g.buf[:, :, 1:-1, 1:-1].normal_(); g = g.as_folded()
ga = T.BT.alloc(n, c, h, w, bf, dev, halo=1, zero=True)
gb = T.BT.alloc(n, c, h, w, bf, dev, halo=1, zero=True)
ws = T.attn_workspace(n, c, dev)
plane = n * c * h * w * 2 / 1e6   # MB of one tensor
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
cases = [
    ("attn_fwd sca", lambda: T.attn_fwd(a, b, o, T.ATTN_MODES["sca"], ws), 5 * plane),      # plane sums read a, b; main reads a, b, writes o
    ("attn_fwd sa ", lambda: T.attn_fwd(a, b, o, T.ATTN_MODES["sa"], ws), 3 * plane),
    ("attn_bwd sca", lambda: T.attn_bwd(a, b, g, ga, gb, T.ATTN_MODES["sca"], False, ws), 12 * plane),
    ("relu_mask   ", lambda: T.relu_mask_(a, ga), 3 * plane),
]
print(f"# n={n} c={c} {h}x{w}: one tensor = {plane:.1f} MB")
for name, fn, mb in cases:
    us = timeit(fn)
    print(f"{name}: {us:8.1f} us   {mb / us * 1e3:7.0f} GB/s (algorithmic {mb:.0f} MB)")
