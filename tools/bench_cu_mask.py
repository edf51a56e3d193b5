#!/usr/bin/env python3
"""Experiment: MFMA-bound conv on a CU-masked stream concurrently with an HBM-bound kernel on the complementary mask."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multi-modal-image-fusion_amd")]
import torch
from mmif import tensor as T
from mmif._lib import IMPL_MFMA
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xffffffff for i in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)
dev = "cuda:0"
torch.manual_seed(0)
B, S = 32, 256
def mk(cin, cout):
    x = T.BT.alloc(B, cin, S, S, torch.bfloat16, dev); x.buf.normal_()
    y = T.BT.alloc(B, cout, S, S, torch.bfloat16, dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.03; b = torch.randn(cout, device=dev)
    pk = T.PackedWeights(cout, cin, 3, dev); pk.pack(w)
    return lambda: T.conv_fwd(x, w, b, y, cin, cout, 3, True, pk, IMPL_MFMA)
big = mk(128, 128)
thin = mk(16, 16)
src = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_(); dst = torch.empty_like(src)   # 1 GiB each
copy = lambda: dst.copy_(src)
def t_stream(fn, st, iters=30):
    with torch.cuda.stream(st):
        for _ in range(10): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
full = torch.cuda.current_stream()
print("full chip: big %.3f ms  thin %.3f ms  copy(2 GiB moved) %.3f ms" % (t_stream(big, full), t_stream(thin, full), t_stream(copy, full)))
ALL = (1 << 256) - 1
masks = {"low128": (1 << 128) - 1, "even": int("01" * 128, 2), "low192": (1 << 192) - 1, "3of4": int("0111" * 64, 2), "low64": (1<<64)-1, "1of4": int("0001"*64, 2)}
for name, m in masks.items():
    sm, sh = masked_stream(m), masked_stream(ALL & ~m)
    tb, tt, tc = t_stream(big, sm), t_stream(thin, sh), t_stream(copy, sh)
    # concurrent: nb big convs on sm, enough thin / copy work on sh to cover them
    for oname, other, to in (("thin", thin, tt), ("copy", copy, tc)):
        nb = 20; no = max(1, int(nb * tb / to))
        for _ in range(2):
            torch.cuda.synchronize()
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(full)
            sm.wait_stream(full); sh.wait_stream(full)
            with torch.cuda.stream(sm):
                for _ in range(nb): big()
                e1.record()
            with torch.cuda.stream(sh):
                for _ in range(no): other()
                e2.record()
            torch.cuda.synchronize()
        print(f"{name:7s}: alone big {tb:.3f}  thin {tt:.3f}  copy {tc:.3f} | with {oname}: big {e0.elapsed_time(e1) / nb:.3f} ms/conv, {oname} {e0.elapsed_time(e2) / no:.3f} ms/call  (serial full-chip cost of the same work: {nb} x big + {no} x {oname})", flush=True)
