#!/usr/bin/env python3
"""Ablation of the DenseBlock backward-chain dgrad (thin_conv_async_kernel<1, true>, 16 output channels): time one launch with the
epilogue's ReLU mask and / or accumulate operand switched off, for the three input widths of the chain.
usage: bench_chain.py [B S iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-modal-image-fusion_amd")]
import torch
from mmif import tensor as T
from mmif._lib import IMPL_MFMA
B, S, iters = [int(a) for a in (sys.argv[1:4] + ["32", "256", "20"][len(sys.argv) - 1:])]
dev = "cuda:0"
torch.manual_seed(0)
x = T.BT.alloc(B, 16, S, S, torch.bfloat16, dev); x.buf.normal_()
for cout in (16, 32, 48):
    gy = T.BT.alloc(B, cout, S, S, torch.bfloat16, dev, halo=1, zero=True); gy.buf[:, :, 1:-1, 1:-1].normal_(); gy = gy.as_folded()
    gx = T.BT.alloc(B, 16, S, S, torch.bfloat16, dev, halo=1, zero=True)
    w = torch.randn(cout, 16, 3, 3, device=dev) * 0.03
    pk = T.PackedWeights(cout, 16, 3, dev); pk.pack(w)
    for mask, acc in ((3, 3), (0, 3), (3, 0), (0, 0)):
        def run():
            T.conv_dgrad(gy, None, x, gx, 16, cout, 3, mask, acc, pk, IMPL_MFMA, fold=True)
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        chp = cout * 1.27 + (16 if mask else 0) + (16 if acc else 0) + 16   # channel passes incl. the 18/16-squared halo of gy
        print(f"{cout:2d} -> 16  mask={'on ' if mask else 'off'} accumulate={'on ' if acc else 'off'}: {us:6.1f} us   {chp * B * S * S * 2 / us / 1e6:5.2f} TB/s of real traffic")
