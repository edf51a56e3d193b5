#!/usr/bin/env python3
"""Time one 3x3 layer's fwd / dgrad / wgrad on fp32 tensors: split-bf16 matrix-pipe kernels (x3) vs the fp32 FMA kernels.
usage: bench_x3.py cin cout B S [iters] [valu]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-modal-image-fusion_amd")]
import torch
from mmif import tensor as T
from mmif._lib import F32, IMPL_VALU, IMPL_X3
cin, cout, B, S = [int(a) for a in (sys.argv[1:5] + ["128", "128", "32", "256"][len(sys.argv) - 1:])]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
with_valu = len(sys.argv) > 6
dev = "cuda:0"
torch.manual_seed(0)
x = T.BT.alloc(B, cin, S, S, torch.float32, dev); x.buf.normal_()
y = T.BT.alloc(B, cout, S, S, torch.float32, dev)
gy = T.BT.alloc(B, cout, S, S, torch.float32, dev, halo=1, zero=True); gy.buf[:, :, 1:-1, 1:-1].normal_()
gy = gy.as_folded()
gx = T.BT.alloc(B, cin, S, S, torch.float32, dev, halo=1, zero=True)
w = torch.randn(cout, cin, 3, 3, device=dev) * 0.03; b = torch.randn(cout, device=dev)
pk = T.PackedWeights(cout, cin, 3, dev, F32); pk.pack(w)
flops = 2.0 * B * S * S * cin * cout * 9
dw = torch.zeros(cout, cin, 3, 3, device=dev); db = torch.zeros(cout, device=dev)
ws = torch.empty(T.wgrad_workspace_bytes(cin, cout, 3) // 4 + 1, dtype=torch.float32, device=dev)
MASK = 0 if os.environ.get("BENCH_NOMASK") else (1 << gx.cb) - 1
def run(kind, impl):
    if kind == "fwd": T.conv_fwd(x, w, b, y, cin, cout, 3, True, pk, impl)
    elif kind == "dgrad": T.conv_dgrad(gy, w, x, gx, cin, cout, 3, MASK, 0, pk, impl, fold=True)
    elif kind == "wide": T.conv_bwd_wide(gy, x, gx, dw, db, cin, cout, 3, pk, MASK, ws, signs)   # wgrad (+ sign map) + dgrad masking with it + fold
    else: T.conv_wgrad(x, gy, dw, db, cin, cout, 3, ws, False, impl)
signs = torch.empty(T.bwd_wide_signs_bytes(B, cin, S, S), dtype=torch.uint8, device=dev)
def timeit(kind, impl, iters):
    for _ in range(2): run(kind, impl)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run(kind, impl)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for name, impl, it in (("x3", IMPL_X3, iters),) + ((("valu", IMPL_VALU, 2),) if with_valu else ()):
    for kind in ("fwd", "dgrad", "wgrad") + (("wide",) if name == "x3" else ()):
        ms = timeit(kind, impl, it)
        print(f"{name} {kind} {cin}->{cout} B={B} {S}x{S}: {ms:.3f} ms  {flops / ms / 1e9:.0f} TFLOP/s fp32-equivalent ({3 * flops / ms / 1e9:.0f} bf16 MFMA)", flush=True)
