#!/usr/bin/env python3
"""What does the L2 -> LDS staging of conv_dma_kernel cost, and what could sharing it buy?  decode.0 (128 -> 128) and decode.1 (128 -> 64)
forward / folded dgrad at B = 32, 256 x 256 under $MMIF_ABLATE conv= (read once per process: run one process per value, tools/sweep_staging.sh):
  0  the kernel as it ships            2  no WEIGHT pieces on every second item (upper bound of "one weight chunk serves two pixel tiles")
  8  no INPUT pieces on every second item (upper bound of "one input tile serves both M-blocks")   10  both      1  no staging at all
Results of the ablated runs are garbage; the data stays Gaussian (zeros would measure the clock)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-modal-image-fusion_amd")]
import torch
from mmif import tensor as T
from mmif._lib import IMPL_MFMA
B, S = 32, 256
dev = "cuda:0"
torch.manual_seed(0)
abl = os.environ.get("MMIF_ABLATE", "0")
for cin, cout in ((128, 128), (128, 64)):
    x = T.BT.alloc(B, cin, S, S, torch.bfloat16, dev); x.buf.normal_()
    y = T.BT.alloc(B, cout, S, S, torch.bfloat16, dev)
    gy = T.BT.alloc(B, cout, S, S, torch.bfloat16, dev, halo=1, zero=True); gy.buf[:, :, 1:-1, 1:-1].normal_()
    gyf = gy.as_folded()
    gx = T.BT.alloc(B, cin, S, S, torch.bfloat16, dev, halo=1, zero=True)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.03; b = torch.randn(cout, device=dev)
    pk = T.PackedWeights(cout, cin, 3, dev); pk.pack(w)
    flops = 2.0 * B * S * S * cin * cout * 9
    def run(kind):
        if kind == "fwd": T.conv_fwd(x, w, b, y, cin, cout, 3, True, pk, IMPL_MFMA)
        else: T.conv_dgrad(gyf, w, x, gx, cin, cout, 3, (1 << 16) - 1, 0, pk, IMPL_MFMA, fold=True)
    for kind in ("fwd", "dgrad"):
        for _ in range(30): run(kind)      # sustained (clock ramp)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): run(kind)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 40
        print(f"ablate={abl:>2s}  {cin:3d}->{cout:3d} {kind:5s}: {ms:.3f} ms  {flops / ms / 1e9:.0f} TFLOP/s ({flops / ms / 1e9 / 2500:.3f} of peak)", flush=True)
    del x, y, gy, gx
