#!/usr/bin/env python3
"""decode.0's bf16 kernels (conv_dma forward, folded dgrad, wgrad_dma) on different OPERAND DATA at identical shapes and instruction
streams: Gaussian, constant, all-zero activations.  The time differences are the chip's clock (power) response to the data, not code."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-modal-image-fusion_amd")]
import torch
from mmif import tensor as T
from mmif._lib import IMPL_MFMA
cin = cout = 128; B, S = 32, 256
dev = "cuda:0"
torch.manual_seed(0)
x = T.BT.alloc(B, cin, S, S, torch.bfloat16, dev)
y = T.BT.alloc(B, cout, S, S, torch.bfloat16, dev)
gy = T.BT.alloc(B, cout, S, S, torch.bfloat16, dev, halo=1, zero=True)
gyf = gy.as_folded()
gx = T.BT.alloc(B, cin, S, S, torch.bfloat16, dev, halo=1, zero=True)
w = torch.randn(cout, cin, 3, 3, device=dev) * 0.03; b = torch.randn(cout, device=dev)
pk = T.PackedWeights(cout, cin, 3, dev); pk.pack(w)
dw = torch.zeros(cout, cin, 3, 3, device=dev); db = torch.zeros(cout, device=dev)
ws = torch.empty(T.wgrad_workspace_bytes(cin, cout, 3) // 4 + 1, dtype=torch.float32, device=dev)
flops = 2.0 * B * S * S * cin * cout * 9
def run(kind):
    if kind == "fwd": T.conv_fwd(x, w, b, y, cin, cout, 3, True, pk, IMPL_MFMA)
    elif kind == "dgrad": T.conv_dgrad(gyf, w, x, gx, cin, cout, 3, (1 << 16) - 1, 0, pk, IMPL_MFMA, fold=True)
    else: T.conv_wgrad(x, gyf, dw, db, cin, cout, 3, ws, False, IMPL_MFMA)
def timeit(kind, iters=40):
    for _ in range(20): run(kind)      # sustained: the first launches of a process run ~15 % slower (clock ramp)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run(kind)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for name in ("gaussian", "constant", "zero"):
    if name == "gaussian":
        x.buf.normal_(); gy.buf[:, :, 1:-1, 1:-1].normal_()
    elif name == "constant":
        x.buf.fill_(0.5); gy.buf[:, :, 1:-1, 1:-1].fill_(0.5)
    else:
        x.buf.zero_(); gy.buf.zero_()
    for kind in ("fwd", "dgrad", "wgrad"):
        ms = timeit(kind)
        print(f"{name:9s} {kind:5s}: {ms:.3f} ms  {flops / ms / 1e9:.0f} TFLOP/s ({flops / ms / 1e9 / 2500:.3f} of the 2.5 PFLOP/s peak)", flush=True)
