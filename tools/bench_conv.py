#!/usr/bin/env python3
"""Time one conv layer's fwd / dgrad (bf16 MFMA kernels) and check the DMA-staged kernel against the register-staged one.
usage: bench_conv.py cin cout B S [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-modal-image-fusion_amd")]
import torch
from mmif import tensor as T
from mmif._lib import IMPL_MFMA
cin, cout, B, S = [int(a) for a in (sys.argv[1:5] + ["128", "128", "32", "256"][len(sys.argv) - 1:])]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = "cuda:0"
torch.manual_seed(0)
x = T.BT.alloc(B, cin, S, S, torch.bfloat16, dev); x.buf.normal_()
y = T.BT.alloc(B, cout, S, S, torch.bfloat16, dev)
gy = T.BT.alloc(B, cout, S, S, torch.bfloat16, dev, halo=1, zero=True); gy.buf[:, :, 1:-1, 1:-1].normal_()
gy = gy.as_folded()
gx = T.BT.alloc(B, cin, S, S, torch.bfloat16, dev, halo=1)
w = torch.randn(cout, cin, 3, 3, device=dev) * 0.03; b = torch.randn(cout, device=dev)
pk = T.PackedWeights(cout, cin, 3, dev); pk.pack(w)
flops = 2.0 * B * S * S * cin * cout * 9
MASK = int(os.environ.get("BENCH_MASK", str((1 << gx.cb) - 1)), 0)   # dgrad ReLU-mask bits (default: all channel blocks)
dw = torch.zeros(cout, cin, 3, 3, device=dev); db = torch.zeros(cout, device=dev)
ws = torch.empty(T.wgrad_workspace_bytes(cin, cout, 3) // 4 + 1, dtype=torch.float32, device=dev)
def run(kind):
    if kind == "fwd": T.conv_fwd(x, w, b, y, cin, cout, 3, True, pk, IMPL_MFMA)
    elif kind == "dgrad": T.conv_dgrad(gy, w, x, gx, cin, cout, 3, MASK, 0, pk, IMPL_MFMA, fold=os.environ.get("BENCH_FOLD", "0") == "1")
    else: T.conv_wgrad(x, gy, dw, db, cin, cout, 3, ws, False, IMPL_MFMA)
def timeit(kind):
    for _ in range(3): run(kind)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run(kind)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
mode = os.environ.get("MMIF_CONV_DMA", "1")
for kind in ("fwd", "dgrad", "wgrad"):
    ms = timeit(kind)
    print(f"MMIF_CONV_DMA={mode} {kind} {cin}->{cout} B={B} {S}x{S}: {ms:.3f} ms  {flops / ms / 1e9:.0f} TFLOP/s")
torch.save({"y": y.buf.cpu(), "gx": gx.buf.cpu(), "dw": dw.cpu(), "db": db.cpu()}, f"/tmp/conv_out_{mode}.pt")
other = f"/tmp/conv_out_{'0' if mode != '0' else '1'}.pt"
if os.path.isfile(other):
    o = torch.load(other)
    print("max |y diff| vs other kernel:", float((o["y"].float() - y.buf.cpu().float()).abs().max()),
          " max |gx diff|:", float((o["gx"].float()[:, :, 1:-1, 1:-1] - gx.buf.cpu().float()[:, :, 1:-1, 1:-1]).abs().max()),
          " full gx diff:", float((o["gx"].float() - gx.buf.cpu().float()).abs().max()),
          " dw rel diff:", float((o["dw"] - dw.cpu()).abs().max() / o["dw"].abs().max()), " db rel diff:", float((o["db"] - db.cpu()).abs().max() / o["db"].abs().max()))
