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
dev = "cuda:0"
x = T.BT.alloc(B, cin, S, S, torch.bfloat16, dev); x.buf.normal_()
y = T.BT.alloc(B, cout, S, S, torch.bfloat16, dev)
w = torch.randn(cout, cin, 3, 3, device=dev) * 0.03; b = torch.zeros(cout, device=dev)
pk = T.PackedWeights(cout, cin, 3, dev); pk.pack(w)
for _ in range(3): T.conv_fwd(x, w, b, y, cin, cout, 3, True, pk, IMPL_MFMA)
tr = torch.zeros(1024, 64, dtype=torch.int64, device=dev)
lib.mmif_debug_set_trace(C.c_void_p(tr.data_ptr()))
T.conv_fwd(x, w, b, y, cin, cout, 3, True, pk, IMPL_MFMA)
torch.cuda.synchronize(); lib.mmif_debug_set_trace(None)
t = tr.cpu().numpy().reshape(128, 8, 64)
n = int(t[0, 0, 63])
# per-wave view: when does each wave reach the end of its k-loop relative to wave 0 of its block (chunk 1..3)
ends = t[:, :, [1 + 5 * q + 5 for q in range(1, 4)]].astype(np.float64)      # stamp "next top" of chunks 1..3
tops = t[:, :, [1 + 5 * q + 2 for q in range(1, 4)]].astype(np.float64)      # barrier passed
print("k-loop+epilogue span per wave (barrier passed -> next chunk top), median over blocks/chunks:")
print("  ", np.median((ends - tops), axis=(0, 2)).round(0))
print("arrival at the next barrier relative to the block's first arrival, median:")
rel = ends - ends.min(axis=1, keepdims=True)
print("  ", np.median(rel, axis=(0, 2)).round(0))
t = t[:, 0, :]
d = np.diff(t[:, :n], axis=1).astype(np.float64)
names = ["prologue (desc + first DMA issue)"]
per = ["wait own DMAs (vmcnt 0)", "barrier", "issue next DMAs", "pending epilogue", "k-loop"]
i = 0
while len(names) < d.shape[1]:
    names.append(f"q{i // 5} {per[i % 5]}"); i += 1
print(f"conv_dma {cin}->{cout} B={B} {S}x{S}: {n} stamps, median / mean cycles over 128 blocks (wave 0)")
for i, nm in enumerate(names[:d.shape[1]]): print(f"  {nm:36s} {np.median(d[:, i]):9.0f} {d[:, i].mean():9.0f}")
