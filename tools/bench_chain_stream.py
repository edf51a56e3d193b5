#!/usr/bin/env python3
"""enc_chain_bwd_kernel (streaming DenseBlock backward chain) at B = 32, 256 x 256, one and two branches per launch, under $MMIF_ABLATE ec=
(read once per process): 0 as shipped, 1 no operand requests, 2 no output stores, 4 no k-loops, 7 all three (what is left is the row
bookkeeping + epilogues).  tools/sweep_chain.sh runs one process per value."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-modal-image-fusion_amd")]
import torch
from mmif import tensor as T
B, S = 32, 256
dev = "cuda:0"
torch.manual_seed(0)
F = T.BT.alloc(B, 128, S, S, torch.bfloat16, dev); F.buf.normal_(); F.buf.relu_()
G = T.BT.alloc(B, 128, S, S, torch.bfloat16, dev, halo=1, zero=True); G.buf[:, :, 1:-1, 1:-1].normal_()
G = G.as_folded()
OUT = T.BT.alloc(B, 128, S, S, torch.bfloat16, dev)
ws = [torch.randn(16, 16 * (i + 1), 3, 3, device=dev) * 0.1 for i in range(3)]
pk = T.pack_dense_chain(*ws, dev)
br = [(G.view(6, 2), G.view(0, 6), F.view(0, 6), pk, OUT.view(0, 8)), (G.view(14, 2), G.view(8, 6), F.view(8, 6), pk, OUT.view(8, 8))]
abl = os.environ.get("MMIF_ABLATE", "0")
for nb in (1, 2):
    for _ in range(10): T.dense_encoder_chain(br[:nb])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): T.dense_encoder_chain(br[:nb])
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    moved = nb * (176 / 16) * B * S * S * 16 * 2
    print(f"ablate={abl}  branches={nb}: {ms * 1e3:.1f} us per launch  ({ms * 1e3 / nb:.1f} per branch; {moved / ms / 1e9:.2f} TB/s of its 176 planes per branch)", flush=True)
