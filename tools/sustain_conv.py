#!/usr/bin/env python3
"""Does conv_dma_kernel hold its rate when it runs back to back for seconds, and what shader clock does it see?
    python tools/sustain_conv.py [cin cout B S] [fwd|dgrad] [data: normal|relu|const] [seconds]
Prints TFLOP/s of the first and the last 20 launches and the s_memtime rate of a traced launch right after the sustained run."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-modal-image-fusion_amd")]
import torch
from mmif import tensor as T
from mmif._lib import lib, IMPL_MFMA
cin, cout, B, S = [int(a) for a in (sys.argv[1:5] + ["128", "128", "32", "256"][len(sys.argv) - 1:])]
kind = sys.argv[5] if len(sys.argv) > 5 else "fwd"
data = sys.argv[6] if len(sys.argv) > 6 else "relu"
secs = float(sys.argv[7]) if len(sys.argv) > 7 else 2.0
dev = "cuda:0"
def fill(t):
    if data == "normal": t.normal_()
    elif data == "relu": t.normal_().clamp_(min=0)
    else: t.fill_(0.5)
x = T.BT.alloc(B, cin, S, S, torch.bfloat16, dev); fill(x.buf)
y = T.BT.alloc(B, cout, S, S, torch.bfloat16, dev)
w = torch.randn(cout, cin, 3, 3, device=dev) * 0.03; b = torch.zeros(cout, device=dev)
pk = T.PackedWeights(cout, cin, 3, dev); pk.pack(w)
gy = T.BT.alloc(B, cout, S, S, torch.bfloat16, dev, halo=1, zero=True); fill(gy.buf[:, :, 1:-1, 1:-1]); gy = gy.as_folded()
gx = T.BT.alloc(B, cin, S, S, torch.bfloat16, dev, halo=1)
def launch():
    if kind == "fwd": T.conv_fwd(x, w, b, y, cin, cout, 3, True, pk, IMPL_MFMA)
    else: T.conv_dgrad(gy, w, x, gx, cin, cout, 3, (1 << gx.cb) - 1, 0, pk, IMPL_MFMA)
for _ in range(3): launch()
torch.cuda.synchronize()
flops = 2.0 * B * S * S * cin * cout * 9
def timed(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): launch()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
first = timed(20)
tot = 0.0
while tot < secs * 1e3:
    tot += timed(100) * 100
last = timed(20)
tr = torch.zeros(1024, 64, dtype=torch.int64, device=dev)
lib.mmif_debug_set_trace(C.c_void_p(tr.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); launch(); e1.record(); torch.cuda.synchronize(); lib.mmif_debug_set_trace(None)
ms = e0.elapsed_time(e1)
t = tr.cpu().numpy().reshape(128, 8, 64).astype(np.float64)
n = int(t[0, 0, 63]) if t[0, 0, 63] > 0 else 62
span = t[:, 0, n - 1] - t[:, 0, 0]
nq = (n - 1) // 5
tops = t[:, :, [1 + 5 * q for q in range(1, nq)]]
if "conv=256" in os.environ.get("MMIF_ABLATE", ""):
    tp = t[:, :, 1:n]                       # tops of chunks 0, 3, 6, ...
    per = np.median(np.diff(tp, axis=2), axis=0) / 3.0
    print("  chunk period over the launch (ticks, wave 0 / wave 4):")
    print("   w0", per[0].round(0))
    print("   w4", per[4].round(0))
whole = np.median(t[:, 0, 62] - t[:, 0, 0])
print(f"  whole block: {whole:.0f} ticks in {ms:.3f} ms -> shader clock ~{whole / (ms * 1e3):.0f} MHz (only the first {nq} chunks carry stamps)")
print(f"conv_dma {kind} {cin}->{cout} B={B} {S}^2 data={data} abl={os.environ.get("MMIF_ABLATE", "0")}: first {first:.3f} ms ({flops / first / 1e9:.0f} TFLOP/s)  "
      f"after {secs:.0f} s {last:.3f} ms ({flops / last / 1e9:.0f})  traced {ms:.3f} ms, chunk period {np.median(np.diff(tops, axis=2)):.0f} ticks, "
      f"{np.median(span) / (ms * 1e3) * (1.0):.0f} ticks/us x (stamped fraction of the launch)")
