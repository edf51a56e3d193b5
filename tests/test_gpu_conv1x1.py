"""The streaming 1x1 kernel (csrc/conv1x1.hip: NestFuse / RFN-Nest's ConvBlock second layers, reference core/block.py:708-722, and RFN's
2C -> C layer, :749) against the register-staged conv_mfma_kernel<1, ...> it replaces -- BIT FOR BIT on identical bf16 operands (same
k-group order, same rounding points) -- and against the CPU oracle on the same rounded operands.  Shapes are NestFuse's own channel
pairs, ragged everything: pixel counts that are not multiples of 32, input chunks that are not multiples of 32 channels, output
fragments past 64 / 128 / 192 rows, channel-slot views of wider buffers (cb_off != 0), halo-1 gradients walked over the padded plane."""
import numpy as np
import pytest
import torch

from oracle import fusion_oracle as O
from gpu_util import bf16_round, close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

#          cin cout  n   h   w      (NestFuse: 8->64, 32->112, 56->160, 80->208, 88->64, 136->112, 184->160, 120->64, 192->112, 152->64, 64->1 is image-side)
SHAPES = [(8, 64, 2, 37, 53), (32, 112, 1, 40, 24), (56, 160, 2, 19, 21), (80, 208, 1, 9, 11), (88, 64, 2, 36, 44), (136, 112, 1, 33, 47),
          (184, 160, 2, 18, 22), (120, 64, 1, 64, 64), (192, 112, 1, 25, 31), (152, 64, 3, 16, 16), (128, 64, 1, 5, 7), (416, 208, 1, 12, 10),
          (16, 16, 2, 8, 9), (40, 24, 1, 31, 33)]


def _views(n, c, h, w, halo, lead, tail):
    """a view of `c` channels inside a wider allocation (lead / tail extra channel blocks around it)"""
    from mmif import tensor as T
    big = T.BT.alloc(n, c + 8 * (lead + tail), h, w, torch.bfloat16, DEV, halo=halo, zero=True)
    return big, big.view(lead, (c + 7) // 8)


@pytest.mark.parametrize("cin,cout,n,h,w", SHAPES, ids=[f"{a}-{b}-{n}x{h}x{w}" for a, b, n, h, w in SHAPES])
def test_streaming_1x1_equals_register_staged_and_oracle(cin, cout, n, h, w):
    from mmif import tensor as T
    from mmif._lib import IMPL_MFMA, lib
    torch.manual_seed(cin * 3 + cout)
    xn = torch.randn(n, cin, h, w)
    xn[xn.abs() < 0.4] = 0.0                                  # post-ReLU style: exact zeros for the mask
    gn = torch.randn(n, cout, h, w)
    wt = (torch.randn(cout, cin, 1, 1) * 0.1).to(DEV)
    b = torch.randn(cout).to(DEV)
    pk = T.PackedWeights(cout, cin, 1, DEV)
    pk.pack(wt)
    xb = T.BT.from_nchw(xn.to(DEV), torch.bfloat16)
    gy = T.BT.from_nchw(gn.to(DEV), torch.bfloat16, halo=1).as_folded()
    cbx = xb.cb
    mask = 0x2d2d2d2d2d2d2d & ((1 << cbx) - 1)
    res = {}
    try:
        for mode in (0, 1):
            lib.mmif_debug_set_conv1x1_stream(mode)
            ybig, y = _views(n, cout, h, w, 0, 3, 1)             # output slot inside a wider buffer
            ybig.buf.fill_(7.0)
            T.conv_fwd(xb, wt, b, y, cin, cout, 1, True, pk, IMPL_MFMA)
            y2 = T.BT.alloc(n, cout, h, w, torch.bfloat16, DEV)
            T.conv_fwd(xb, wt, None if False else b, y2, cin, cout, 1, False, pk, IMPL_MFMA)      # no activation
            gbig, gx = _views(n, cin, h, w, 1, 2, 2)             # halo-1 gradient slot
            T.conv_dgrad(gy, wt, xb, gx, cin, cout, 1, mask, 0, pk, IMPL_MFMA, fold=True)
            gx_all = T.BT.alloc(n, cin, h, w, torch.bfloat16, DEV, halo=1, zero=True)
            T.conv_dgrad(gy, wt, xb, gx_all, cin, cout, 1, (1 << cbx) - 1, 0, pk, IMPL_MFMA, fold=True)
            gx_none = T.BT.alloc(n, cin, h, w, torch.bfloat16, DEV, halo=1, zero=True)
            T.conv_dgrad(gy, wt, None, gx_none, cin, cout, 1, 0, 0, pk, IMPL_MFMA, fold=True)
            torch.cuda.synchronize()
            res[mode] = [t.buf.view(torch.int16).clone() for t in (ybig, y2, gbig, gx_all, gx_none)] + [y.to_nchw(cout), gx_all.to_nchw(cin), y2.to_nchw(cout)]
    finally:
        lib.mmif_debug_set_conv1x1_stream(1)
    for i, what in enumerate(("fwd (slot view)", "fwd no-ReLU", "dgrad partial mask (slot view)", "dgrad full mask", "dgrad no mask")):
        assert torch.equal(res[0][i], res[1][i]), what
    # neighbours of the slots untouched, gradient ring zero
    ybuf = res[1][0].view(torch.bfloat16).float()
    assert float((ybuf[:, :3] - 7.0).abs().max()) == 0.0 and float((ybuf[:, 3 + (cout + 7) // 8:] - 7.0).abs().max()) == 0.0
    gbuf = res[1][2].view(torch.bfloat16).float()
    assert float(gbuf[:, :2].abs().max()) == 0.0 and float(gbuf[:, 2 + cbx:].abs().max()) == 0.0
    assert float(gbuf[:, :, 0].abs().max()) == 0.0 and float(gbuf[:, :, -1].abs().max()) == 0.0 and float(gbuf[:, :, :, 0].abs().max()) == 0.0
    # ... and the values are right: oracle on the rounded operands, one output rounding
    xq, gq, wq = bf16_round(xn.numpy()), bf16_round(gn.numpy()), bf16_round(wt.cpu().numpy())
    y_or = O.conv2d_reflect_fwd(xq, wq, b.cpu().numpy(), True)
    close(res[1][5].cpu().numpy(), bf16_round(y_or), 6e-3, "y vs oracle")
    y_lin = O.conv2d_reflect_fwd(xq, wq, b.cpu().numpy(), False)
    close(res[1][7].cpu().numpy(), bf16_round(y_lin), 6e-3, "y (no ReLU) vs oracle")
    gx_or = np.einsum("oc,nohw->nchw", wq[:, :, 0, 0].astype(np.float64), gq.astype(np.float64)) * (xq > 0)
    close(res[1][6].cpu().numpy(), bf16_round(gx_or.astype(np.float32)), 6e-3, "gx vs oracle")


def test_streaming_1x1_is_the_kernel_that_runs_and_accumulate_falls_back():
    """accum_bits != 0 (no such 1x1 dgrad on the NestFuse path) stays on the register-staged kernel and still gives the right sum"""
    from mmif import tensor as T
    from mmif._lib import IMPL_MFMA
    cin, cout, n, h, w = 24, 40, 1, 9, 13
    torch.manual_seed(3)
    xn, gn = torch.randn(n, cin, h, w), torch.randn(n, cout, h, w)
    wt = (torch.randn(cout, cin, 1, 1) * 0.1).to(DEV)
    pk = T.PackedWeights(cout, cin, 1, DEV)
    pk.pack(wt)
    xb = T.BT.from_nchw(xn.to(DEV), torch.bfloat16)
    gy = T.BT.from_nchw(gn.to(DEV), torch.bfloat16, halo=1).as_folded()
    old = torch.randn(n, cin, h, w)
    gx = T.BT.from_nchw(old.to(DEV), torch.bfloat16, halo=1)
    T.conv_dgrad(gy, wt, xb, gx, cin, cout, 1, 0, (1 << xb.cb) - 1, pk, IMPL_MFMA, fold=True)
    torch.cuda.synchronize()
    want = np.einsum("oc,nohw->nchw", bf16_round(wt.cpu().numpy())[:, :, 0, 0].astype(np.float64), bf16_round(gn.numpy()).astype(np.float64)) + bf16_round(old.numpy())
    close(gx.to_nchw(cin).cpu().numpy(), want.astype(np.float32), 8e-3, "accumulating 1x1 dgrad")


@pytest.mark.parametrize("cin,cout,n,h,w", SHAPES, ids=[f"{a}-{b}-{n}x{h}x{w}" for a, b, n, h, w in SHAPES])
def test_1x1_weight_gradient_wide_blocks_vs_fp64(cin, cout, n, h, w):
    """1x1 weight gradients (wgrad_mfma_kernel<1, 4, 2, ICF>: a block owns up to 64 input channels per staged 64-channel gradient tile,
    round 4) against the fp64 definition on the same bf16-rounded operands: dW[o, c] = sum_p g[o, p] x[c, p], db[o] = sum_p g[o, p];
    fp32 accumulation over up to 4 k pixels per block partial -> 2e-4 of max|dW|.  Accumulate flag on top of existing values."""
    from mmif import tensor as T
    from mmif._lib import IMPL_MFMA
    torch.manual_seed(cin + 5 * cout)
    xn, gn = torch.randn(n, cin, h, w), torch.randn(n, cout, h, w)
    xb = T.BT.from_nchw(xn.to(DEV), torch.bfloat16)
    gy = T.BT.from_nchw(gn.to(DEV), torch.bfloat16, halo=1).as_folded()
    ws = torch.empty(T.wgrad_workspace_bytes(cin, cout, 1) // 4 + 1, dtype=torch.float32, device=DEV)
    dw, db = torch.full((cout, cin, 1, 1), 0.5, device=DEV), torch.full((cout,), -0.25, device=DEV)
    T.conv_wgrad(xb, gy, dw, db, cin, cout, 1, ws, True, IMPL_MFMA)
    dw2, db2 = torch.empty(cout, cin, 1, 1, device=DEV), torch.empty(cout, device=DEV)
    T.conv_wgrad(xb, gy, dw2, db2, cin, cout, 1, ws, False, IMPL_MFMA)
    torch.cuda.synchronize()
    xq, gq = bf16_round(xn.numpy()).astype(np.float64), bf16_round(gn.numpy()).astype(np.float64)
    want = np.einsum("nohw,nchw->oc", gq, xq)[:, :, None, None]
    wantb = gq.sum(axis=(0, 2, 3))
    close(dw2.cpu().numpy(), want, 2e-4, "dW")
    close(db2.cpu().numpy(), wantb, 2e-4, "db")
    close(dw.cpu().numpy(), want + 0.5, 2e-4, "dW accumulate")
    close(db.cpu().numpy(), wantb - 0.25, 2e-4, "db accumulate")
