"""Backward of one PFNetv2 pair-conv layer in one pass (csrc/pair.hip pairconv_bwd_kernel, mmif_pairconv_bwd): the operand gradients
must be BIT-IDENTICAL to mmif_pairconv_dgrad (same tap order, residual add and ReLU mask), dW / db within fp32 summation-order
noise of mmif_pairconv_wgrad (bf16 storage multiplies with v_dot2c_f32_bf16: the products are exact either way); ragged tiles,
image borders, one and two outputs, both storage types; the whole model trains to the same numbers with the fused pass on and off."""
import os

import pytest
import torch

from gpu_util import DEV, dtype_ctx, load_closed_form

pytestmark = pytest.mark.gpu

SHAPES = [(1, 2, 2), (1, 4, 4), (2, 16, 16), (1, 5, 37), (2, 33, 18), (1, 40, 56), (2, 64, 64), (4, 128, 128)]


@pytest.mark.parametrize("dt", ["bf16", "fp32"])
@pytest.mark.parametrize("nout,masked,with_add", [(1, True, False), (2, True, False), (2, False, True), (2, True, True)])
@pytest.mark.parametrize("n,h,w", SHAPES, ids=[f"{n}x{h}x{w}" for n, h, w in SHAPES])
def test_pairconv_bwd_vs_separate_kernels(dt, nout, masked, with_add, n, h, w):
    from mmif import tensor as T
    tdt = torch.bfloat16 if dt == "bf16" else torch.float32
    with dtype_ctx(dt):
        gen = torch.Generator().manual_seed(h * 131 + w + nout)
        ch = 16
        X = T.BT.from_nchw(torch.relu(torch.randn(n, 2 * ch, h, w, generator=gen)).to(DEV), tdt)
        G = T.BT.from_nchw(torch.randn(n, 2 * ch, h, w, generator=gen).to(DEV), tdt, halo=1).as_folded()
        # the residual gradient as the engine hands it over: a folded halo-1 tensor (odd seeds) or a plain halo-0 one
        add = None
        if with_add:
            add = T.BT.from_nchw(torch.randn(n, ch, h, w, generator=gen).to(DEV), tdt, halo=(h + w) % 2)
            if add.halo:
                add = add.as_folded()
        cb = ch // 8
        xa, xb, ga, gb = X.view(0, cb), X.view(cb, cb), G.view(0, cb), (G.view(cb, cb) if nout == 2 else None)
        wgt = (torch.randn(nout, 2, 3, 3, generator=gen) * 0.3).to(DEV)
        bits = 0b10 if masked else 0
        ws = torch.empty(T.pairconv_wgrad_workspace_bytes() // 4, dtype=torch.float32, device=DEV)
        GA = T.BT.alloc(n, 2 * ch, h, w, tdt, DEV, halo=1, zero=True)
        GB = T.BT.alloc(n, 2 * ch, h, w, tdt, DEV, halo=1, zero=True)
        dw_a, db_a = torch.zeros(nout, 2, 3, 3, device=DEV), torch.zeros(nout, device=DEV)
        dw_b, db_b = torch.zeros_like(dw_a), torch.zeros_like(db_a)
        T.pairconv_bwd(ga, gb, wgt, nout, xa, xb, GA.view(0, cb), GA.view(cb, cb), dw_a, db_a, ws, bits, add=add)
        T.pairconv_wgrad(xa, xb, ga, gb, nout, dw_b, db_b, ws)
        T.pairconv_dgrad(ga, gb, wgt, nout, xa, xb, GB.view(0, cb), GB.view(cb, cb), bits, add=add)
        torch.cuda.synchronize()
        ia, ib = (GA.buf.view(torch.int16), GB.buf.view(torch.int16)) if dt == "bf16" else (GA.buf.view(torch.int32), GB.buf.view(torch.int32))
        if not torch.equal(ia, ib):
            d = (ia != ib).nonzero()
            raise AssertionError(f"gx: {d.shape[0]} of {ia.numel()} elements differ; first at [n, cb, ys, xs, e] = {d[0].tolist()}")
        assert float(GB.buf.float().abs().max()) > 0
        tol = 1e-4 if dt == "fp32" else 2e-4
        assert float((dw_a - dw_b).abs().max()) <= tol * max(1e-6, float(dw_b.abs().max()))
        assert float((db_a - db_b).abs().max()) <= tol * max(1e-6, float(db_b.abs().max()))
        T.pairconv_bwd(ga, gb, wgt, nout, xa, xb, GA.view(0, cb), GA.view(cb, cb), dw_a, db_a, ws, bits, add=add, accumulate=True)
        torch.cuda.synchronize()
        assert float((dw_a - 2 * dw_b).abs().max()) <= 2 * tol * max(1e-6, float(dw_b.abs().max()))
        assert float((db_a - 2 * db_b).abs().max()) <= 2 * tol * max(1e-6, float(db_b.abs().max()))


def test_pfnetv2_trains_the_same_with_and_without_the_fused_pass():
    import core.model as M
    res = {}
    for mode in ("1", "0"):
        os.environ["MMIF_PAIR_BWD"] = mode
        __import__("mmif.engine").engine.reload_switches()
        try:
            with dtype_ctx("bf16"):
                m = load_closed_form(M.PFNetv2(), 3).to(DEV)
                gen = torch.Generator().manual_seed(5)
                a, b = torch.rand(2, 1, 48, 40, generator=gen).to(DEV), torch.rand(2, 1, 48, 40, generator=gen).to(DEV)
                out = m(a, b)
                out.backward(torch.ones_like(out) / out.numel())
                torch.cuda.synchronize()
                res[mode] = (out.detach().float().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()})
        finally:
            os.environ.pop("MMIF_PAIR_BWD", None)
            __import__("mmif.engine").engine.reload_switches()
    assert torch.equal(res["1"][0], res["0"][0])
    for k, ga in res["1"][1].items():
        gb = res["0"][1][k]
        if k.startswith("fuse."):
            assert float((ga - gb).abs().max()) <= 2e-4 * max(1e-6, float(gb.abs().max())), k
        else:   # everything upstream of the fusion sees bit-identical feature gradients
            assert torch.equal(ga, gb), k


FWD_SHAPES = [(1, 2, 2), (1, 4, 4), (2, 16, 16), (1, 5, 37), (2, 33, 18), (1, 40, 56), (2, 64, 64), (1, 70, 33), (4, 128, 128)]


@pytest.mark.parametrize("nout,relu,res", [(2, True, None), (2, False, None), (1, False, "operands"), (1, False, "other"), (1, True, None)])
@pytest.mark.parametrize("n,h,w", FWD_SHAPES, ids=[f"{n}x{h}x{w}" for n, h, w in FWD_SHAPES])
def test_pairconv_fwd_strip_kernel_is_bit_identical(nout, relu, res, n, h, w):
    """bf16 forward with 1 x 4 output strips on 32 x 32 tiles (pairconv_fwd_strip_kernel) vs the one-output-per-thread kernel
    ($MMIF_PAIR_STRIP=0): same tap order per output, so every bit must agree -- ragged tiles, borders, residual, both widths."""
    from mmif import tensor as T
    with dtype_ctx("bf16"):
        gen = torch.Generator().manual_seed(h * 17 + w + nout)
        ch = 24
        X = T.BT.from_nchw(torch.randn(n, 2 * ch, h, w, generator=gen).to(DEV), torch.bfloat16)
        cb = ch // 8
        a, b = X.view(0, cb), X.view(cb, cb)
        wgt = (torch.randn(nout, 2, 3, 3, generator=gen) * 0.3).to(DEV)
        bias = torch.randn(nout, generator=gen).to(DEV)
        outs = {}
        for mode in ("1", "0"):
            os.environ["MMIF_PAIR_STRIP"] = mode
            __import__("mmif.engine").engine.reload_switches()
            try:
                O = T.BT.alloc(n, 2 * ch, h, w, torch.bfloat16, DEV, zero=True)
                r1, r2 = {None: (None, None), "operands": (a, b), "other": (b, a)}[res]   # "operands": taken from the LDS windows
                T.pairconv_fwd(a, b, wgt, bias, nout, O.view(0, cb), O.view(cb, cb) if nout == 2 else None, relu, r1, r2)
                torch.cuda.synchronize()
                outs[mode] = O.buf.view(torch.int16).clone()
            finally:
                os.environ.pop("MMIF_PAIR_STRIP", None)
                __import__("mmif.engine").engine.reload_switches()
        assert float(outs["0"].float().abs().max()) > 0
        if not torch.equal(outs["1"], outs["0"]):
            d = (outs["1"] != outs["0"]).nonzero()
            raise AssertionError(f"{d.shape[0]} of {outs['0'].numel()} elements differ; first at [n, cb, y, x, e] = {d[0].tolist()}")
