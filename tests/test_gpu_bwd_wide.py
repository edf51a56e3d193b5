"""Backward of one wide 3x3 layer through ReLU sign bytes (csrc/conv_mfma.hip bwd_wide, mmif_conv2d_reflect_bwd_wide): wgrad_dma_kernel
leaves one sign byte per pixel and 8 channels, conv_dma_kernel<true, 2> masks with them.  Everything must be BIT-IDENTICAL to
mmif_conv2d_reflect_wgrad + mmif_conv2d_reflect_dgrad_folded reading the activations: same kernels, same order, the mask is the same
predicate (bf16 > 0).  Ragged tiles, partial masks, -0.0 / tiny activations, the sign bytes themselves."""
import os

import pytest
import torch

from gpu_util import DEV, dtype_ctx

pytestmark = pytest.mark.gpu

SHAPES = [(1, 4, 4), (2, 16, 16), (1, 5, 37), (2, 33, 18), (1, 40, 56), (2, 64, 64), (1, 70, 33), (2, 128, 128)]


def _run(cin, cout, n, h, w, mask_bits, seed):
    from mmif import tensor as T
    from mmif._lib import IMPL_MFMA
    g = torch.Generator().manual_seed(seed)
    xv = torch.relu(torch.randn(n, cin, h, w, generator=g))
    xv[:, ::3, ::2, 1::3] = -0.0                      # negative zero and tiny values must count as "not > 0" / "> 0" exactly as bf16 says
    xv[:, 1::5, 1::2, ::4] = 1e-40
    x = T.BT.from_nchw(xv.to(DEV), torch.bfloat16)
    gy = T.BT.from_nchw(torch.randn(n, cout, h, w, generator=g).to(DEV), torch.bfloat16, halo=1).as_folded()
    wgt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(DEV)
    pk = T.PackedWeights(cout, cin, 3, DEV)
    pk.pack(wgt)
    ws = torch.empty(T.wgrad_workspace_bytes(cin, cout, 3) // 4 + 1, dtype=torch.float32, device=DEV)
    signs = torch.full((T.bwd_wide_signs_bytes(n, cin, h, w),), 0xA5, dtype=torch.uint8, device=DEV)
    gx_a = T.BT.alloc(n, cin, h, w, torch.bfloat16, DEV, halo=1, zero=True)
    dw_a, db_a = torch.zeros(cout, cin, 3, 3, device=DEV), torch.zeros(cout, device=DEV)
    T.conv_bwd_wide(gy, x, gx_a, dw_a, db_a, cin, cout, 3, pk, mask_bits, ws, signs)
    gx_b = T.BT.alloc(n, cin, h, w, torch.bfloat16, DEV, halo=1, zero=True)
    dw_b, db_b = torch.zeros_like(dw_a), torch.zeros_like(db_a)
    T.conv_wgrad(x, gy, dw_b, db_b, cin, cout, 3, ws, False, IMPL_MFMA)
    T.conv_dgrad(gy, wgt, x, gx_b, cin, cout, 3, mask_bits, 0, pk, IMPL_MFMA, fold=True)
    torch.cuda.synchronize()
    return x, gx_a, gx_b, (dw_a, db_a), (dw_b, db_b), signs


@pytest.mark.parametrize("cin,cout", [(128, 64), (128, 128), (64, 64)])
@pytest.mark.parametrize("n,h,w", SHAPES, ids=[f"{n}x{h}x{w}" for n, h, w in SHAPES])
def test_bwd_wide_vs_separate_kernels(cin, cout, n, h, w):
    from mmif import tensor as T
    with dtype_ctx("bf16"):
        assert T.bwd_wide_supported(cin, cout, 3)
        x, gx_a, gx_b, (dw_a, db_a), (dw_b, db_b), signs = _run(cin, cout, n, h, w, (1 << (cin // 8)) - 1, h * 131 + w + cin)
        a, b = gx_a.buf.view(torch.int16), gx_b.buf.view(torch.int16)
        if not torch.equal(a, b):
            d = (a != b).nonzero()
            raise AssertionError(f"gx: {d.shape[0]} of {a.numel()} elements differ; first at [n, cb, ys, xs, e] = {d[0].tolist()}")
        assert torch.equal(dw_a, dw_b) and torch.equal(db_a, db_b)
        # the sign bytes: [n][cb][h + 2][pitch], byte of padded column X at X + 15, bit i = channel 8 cb + i > 0
        pitch = ((w + 15) // 16 + 2) * 16
        sg = signs.view(n, cin // 8, h + 2, pitch)[:, :, 1:h + 1, 16:16 + w]
        pos = (x.buf.float() > 0)                                        # [n, cb, h, w, 8]
        want = (pos.to(torch.int32) << torch.arange(8, device=pos.device, dtype=torch.int32)).sum(-1).to(torch.uint8)
        assert torch.equal(sg, want)


def test_bwd_wide_partial_mask_and_accumulate():
    from mmif import tensor as T
    with dtype_ctx("bf16"):
        cin, cout, n, h, w = 128, 128, 2, 40, 56
        mb = (1 << 6) | (1 << 7) | (1 << 14) | (1 << 15)                 # PFNetv1's decode.0: only the encoders' last blocks are masked
        x, gx_a, gx_b, (dw_a, db_a), (dw_b, db_b), _ = _run(cin, cout, n, h, w, mb, 77)
        assert torch.equal(gx_a.buf.view(torch.int16), gx_b.buf.view(torch.int16))
        assert torch.equal(dw_a, dw_b) and torch.equal(db_a, db_b)
        x, gx_a, gx_b, (dw_a, db_a), (dw_b, db_b), _ = _run(cin, cout, n, h, w, 0, 78)   # no mask at all: no sign bytes written or read
        assert torch.equal(gx_a.buf.view(torch.int16), gx_b.buf.view(torch.int16))


def test_bwd_wide_rejects():
    from mmif import tensor as T
    with dtype_ctx("bf16"):
        assert not T.bwd_wide_supported(64, 32, 3) and not T.bwd_wide_supported(128, 64, 1) and not T.bwd_wide_supported(48, 64, 3)
        n, h, w, cin, cout = 1, 16, 16, 128, 64
        x = T.BT.alloc(n, cin, h, w, torch.bfloat16, DEV)
        gy = T.BT.alloc(n, cout, h, w, torch.bfloat16, DEV, halo=1, zero=True).as_folded()
        gx = T.BT.alloc(n, cin, h, w, torch.bfloat16, DEV, halo=1, zero=True)
        pk = T.PackedWeights(cout, cin, 3, DEV)
        pk.pack(torch.zeros(cout, cin, 3, 3, device=DEV))
        ws = torch.empty(T.wgrad_workspace_bytes(cin, cout, 3) // 4 + 1, dtype=torch.float32, device=DEV)
        dw, db = torch.zeros(cout, cin, 3, 3, device=DEV), torch.zeros(cout, device=DEV)
        with pytest.raises(RuntimeError, match="sign-byte buffer too small"):
            T.conv_bwd_wide(gy, x, gx, dw, db, cin, cout, 3, pk, 0xffff, ws, torch.empty(16, dtype=torch.uint8, device=DEV))


def test_models_with_and_without_bwd_wide():
    import core.model as M
    with dtype_ctx("bf16"):
        for name in ("PFNetv1", "VIFNet"):
            torch.manual_seed(5)
            m = getattr(M, name)().to(DEV)
            g = torch.Generator().manual_seed(9)
            i1, i2 = torch.rand(2, 1, 45, 70, generator=g).to(DEV), torch.rand(2, 1, 45, 70, generator=g).to(DEV)
            res = []
            for flag in ("0", "1"):
                os.environ["MMIF_BWD_WIDE"] = flag
                __import__("mmif.engine").engine.reload_switches()
                try:
                    m.zero_grad(set_to_none=True)
                    m(i1, i2).square().mean().backward()
                    torch.cuda.synchronize()
                    res.append({k: p.grad.clone() for k, p in m.named_parameters()})
                finally:
                    os.environ.pop("MMIF_BWD_WIDE", None)
                    __import__("mmif.engine").engine.reload_switches()
            for k in res[0]:
                assert torch.equal(res[0][k], res[1][k]), f"{name} {k}: the sign-byte path changed a gradient"
