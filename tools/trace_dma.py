#!/usr/bin/env python3
"""Phase breakdown of conv_dma_kernel (s_memtime stamps of thread 0 of every block): per chunk
[top -> own DMAs landed -> barrier passed -> next DMAs issued -> pending epilogue done -> (k-loop) -> next top]."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-modal-image-fusion_amd")]
import torch
from mmif import tensor as T
from mmif._lib import lib, IMPL_MFMA
cin, cout, B, S = [int(a) for a in (sys.argv[1:5] + ["128", "128", "32", "256"][len(sys.argv) - 1:])]
kind = sys.argv[5] if len(sys.argv) > 5 else "fwd"   # fwd | dgrad
dev = "cuda:0"
x = T.BT.alloc(B, cin, S, S, torch.bfloat16, dev); x.buf.normal_()
y = T.BT.alloc(B, cout, S, S, torch.bfloat16, dev)
w = torch.randn(cout, cin, 3, 3, device=dev) * 0.03; b = torch.zeros(cout, device=dev)
pk = T.PackedWeights(cout, cin, 3, dev); pk.pack(w)
gy = T.BT.alloc(B, cout, S, S, torch.bfloat16, dev, halo=1, zero=True); gy.buf[:, :, 1:-1, 1:-1].normal_(); gy = gy.as_folded()
gx = T.BT.alloc(B, cin, S, S, torch.bfloat16, dev, halo=1)
def launch():
    if kind == "fwd": T.conv_fwd(x, w, b, y, cin, cout, 3, True, pk, IMPL_MFMA)
    else: T.conv_dgrad(gy, w, x, gx, cin, cout, 3, (1 << gx.cb) - 1, 0, pk, IMPL_MFMA)
for _ in range(3): launch()
tr = torch.zeros(1024, 64, dtype=torch.int64, device=dev)
lib.mmif_debug_set_trace(C.c_void_p(tr.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
launch()
e1.record()
torch.cuda.synchronize(); lib.mmif_debug_set_trace(None)
traced_ms = e0.elapsed_time(e1)
t = tr.cpu().numpy().reshape(128, 8, 64).astype(np.float64)
n = int(t[0, 0, 63])
nq = (n - 1) // 5
# stamps: [0] kernel start; chunk q: 1+5q chunk top, 2+5q barrier passed, 3+5q pending epilogue done, 4+5q k-loop done, 5+5q tile epilogue done
names = ["wait at barrier", "pending epilogue (waves 4-7)", "k-loop", "tile epilogue (waves 0-3)", "loop overhead"]
print(f"conv_dma {kind} {cin}->{cout} B={B} {S}x{S}: {n} stamps/wave; median cycles over 128 blocks, chunks 1..{nq - 1}; one column per consumer wave")
for k, nm in enumerate(names):
    lo = [1 + 5 * q + k for q in range(1, nq - 1)]
    hi = [2 + 5 * q + k for q in range(1, nq - 1)]
    d = t[:, :, hi] - t[:, :, lo]
    print(f"  {nm:30s}", np.median(d, axis=(0, 2)).round(0))
nch = (cin if kind == "fwd" else cout) // 32
print(f"  per chunk position c = q % {nch} (median over blocks; wave 0 | wave 4): barrier wait, pending epilogue, k-loop, tile epilogue, overhead, period")
for q in range(1, nq - 1):
    row = []
    for wv in (0, 4):
        d = [np.median(t[:, wv, 2 + 5 * q + k] - t[:, wv, 1 + 5 * q + k]) for k in range(5)]
        d.append(np.median(t[:, wv, 1 + 5 * (q + 1)] - t[:, wv, 1 + 5 * q]))
        row.append(" ".join(f"{v:6.0f}" for v in d))
    print(f"   q={q:2d} c={q % nch}: {row[0]}  |  {row[1]}")
tops = t[:, :, [1 + 5 * q for q in range(1, nq)]]
print("  chunk period (top -> top)     ", np.median(np.diff(tops, axis=2), axis=(0, 2)).round(0))
arr = t[:, :, [1 + 5 * q for q in range(2, nq)]]
span = (t[:, 0, n - 1] - t[:, 0, 0])
print(f"  traced launch: {traced_ms:.3f} ms; first-to-last stamp of wave 0: median {np.median(span):.0f} ticks -> {np.median(span) / (traced_ms * 1e3):.0f} ticks/us "
      f"(s_memtime is a constant-rate counter if this is ~100, the shader clock in MHz otherwise; stamps cover {nq} of the block's chunks)")
print("  arrival at barrier vs first   ", np.median(arr - arr.min(axis=1, keepdims=True), axis=(0, 2)).round(0))
