#!/usr/bin/env python3
"""Phase breakdown of conv_mfma_kernel on one layer shape (s_memtime stamps of wave 0 in 1024 blocks)."""
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
t = tr.cpu().numpy(); n = int(t[0, 63]); d = np.diff(t[:, :n], axis=1).astype(np.float64)
names = ["prologue", "first prefetch issue"]
nch = (n - 4) // 5
for c in range(nch): names += [f"c{c} wait barrier A", f"c{c} vmcnt wait + LDS store", f"c{c} barrier B", f"c{c} next prefetch issue", f"c{c} k-loop"]
names += ["epilogue"]
print(f"conv {cin}->{cout} B={B} {S}x{S}: {n} stamps; s_memtime ticks (100 MHz const clock? see total), median over 1024 blocks")
for i, nm in enumerate(names[:d.shape[1]]): print(f"  {nm:32s} median {np.median(d[:, i]):9.0f}  mean {d[:, i].mean():9.0f}")
print("  total per block median", np.median(t[:, n - 1] - t[:, 0]))
