#!/usr/bin/env python3
"""Effective shader clock of conv_dma_kernel: s_memtime stamps at every third chunk top over the WHOLE block (MMIF_ABLATE conv=256[+1]) against
the launch's wall time, sustained (50 launches with the trace on).  Run:  for a in 256 257; do MMIF_ABLATE=conv=$a python tools/conv_clock.py; done"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-modal-image-fusion_amd")]
import torch
from mmif import tensor as T
from mmif._lib import lib, IMPL_MFMA
cin, cout, B, S = 128, 128, 32, 256
dev = "cuda:0"
x = T.BT.alloc(B, cin, S, S, torch.bfloat16, dev); x.buf.normal_()
y = T.BT.alloc(B, cout, S, S, torch.bfloat16, dev)
w = torch.randn(cout, cin, 3, 3, device=dev) * 0.03; b = torch.zeros(cout, device=dev)
pk = T.PackedWeights(cout, cin, 3, dev); pk.pack(w)
def launch(): T.conv_fwd(x, w, b, y, cin, cout, 3, True, pk, IMPL_MFMA)
tr = torch.zeros(1024, 64, dtype=torch.int64, device=dev)
lib.mmif_debug_set_trace(C.c_void_p(tr.data_ptr()))
for _ in range(60): launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): launch()
e1.record(); torch.cuda.synchronize(); lib.mmif_debug_set_trace(None)
ms = e0.elapsed_time(e1) / 50
t = tr.cpu().numpy().reshape(128, 8, 64).astype(np.float64)
n = int(t[0, 0, 63])
span = t[:, 0, n - 1] - t[:, 0, 0]
per = np.diff(t[:, 0, 1:n], axis=1)
print(f"MMIF_ABLATE={os.environ.get('MMIF_ABLATE','')}: {ms*1e3:.1f} us per launch; {n} stamps; block span median {np.median(span):.0f} cycles (min {span.min():.0f} max {span.max():.0f}) "
      f"-> {np.median(span)/(ms*1e3):.0f} cycles/us if the span is the launch; cycles per 3 chunks: median {np.median(per):.0f} (first 5: {np.median(per[:, :5]):.0f}, last 5: {np.median(per[:, -5:]):.0f})")
