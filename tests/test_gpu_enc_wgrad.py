"""Fused encoder weight gradients (csrc/enc_wgrad.hip, mmif_dense_encoder_wgrad): dW / db of ConvLayer(1,16) + DenseBlock(16,16)
(reference core/model.py:73-80, core/block.py:137-151) in one pass must agree with the four layer-wise kernels -- same bf16
operands, fp32 accumulation, only the summation order differs -- and with the fp64 definition evaluated on the same operands;
the first layer (fp32 image, exact fp32 matrix path) to fp32 accuracy.  Ragged tiles, one-tile images, accumulate mode."""
import os

import numpy as np
import pytest
import torch

from gpu_util import DEV, dtype_ctx

pytestmark = pytest.mark.gpu

SHAPES = [(1, 2, 2), (2, 5, 7), (1, 16, 16), (2, 32, 32), (1, 37, 53), (2, 64, 64), (1, 70, 33), (2, 256, 256)]


def _setup(n, h, w, seed):
    import core.model as M
    from mmif import engine as E
    from mmif import tensor as T
    torch.manual_seed(seed)
    m = M.PFNetv1().to(DEV)
    eng = E.PFNetv1Engine(m)
    g = torch.Generator().manual_seed(seed + 1)
    img = torch.rand(n, 1, h, w, generator=g).to(DEV)
    x = torch.relu(torch.randn(n, 64, h, w, generator=g) * 0.7).to(DEV)             # stand-in activations (ReLU outputs: half zeros)
    gz = (torch.randn(n, 64, h, w, generator=g) * (torch.rand(n, 64, h, w, generator=g) > 0.4)).to(DEV)   # masked gradients
    F = T.BT.from_nchw(x, torch.bfloat16)
    GF = T.BT.from_nchw(gz, torch.bfloat16, halo=1).as_folded()
    (img,), *_ = eng.prepare((img,))
    return eng, T, img, F, GF


def _grads(eng):
    return [(torch.zeros_like(s.conv.weight), torch.zeros_like(s.conv.bias)) for s in eng.enc[0]]


@pytest.mark.parametrize("n,h,w", SHAPES, ids=[f"{n}x{h}x{w}" for n, h, w in SHAPES])
def test_fused_wgrad_vs_layerwise_and_definition(n, h, w):
    with dtype_ctx("bf16"):
        eng, T, img, F, GF = _setup(n, h, w, 7 + h)
        ws = eng.workspace(torch.device(DEV))
        specs = eng.enc[0]
        fused = _grads(eng)
        T.dense_encoder_wgrad(img, F.view(0, 6), GF.view(0, 8), fused, ws)
        lw = _grads(eng)
        for k, nin in ((3, 6), (2, 4), (1, 2)):
            T.conv_wgrad(F.view(0, nin), GF.view(2 * k, 2), lw[k][0], lw[k][1], specs[k].cin, 16, 3, ws, False)
        T.image_in_wgrad(img, GF.view(0, 2), lw[0][0], lw[0][1], 16, 3, ws, False)
        torch.cuda.synchronize()
        # fp64 definition on the SAME (bf16-rounded) operands
        xr = F.to_nchw(64).double().cpu()
        gr = GF.to_nchw(64).double().cpu()
        im = img.double().cpu()
        for k in range(4):
            xin = im if k == 0 else xr[:, :16 * k]
            gk = gr[:, 16 * k:16 * k + 16]
            xp = torch.nn.functional.pad(xin, (1, 1, 1, 1), mode="reflect")
            dw = torch.zeros(16, xin.shape[1], 3, 3, dtype=torch.float64)
            for u in range(3):
                for v in range(3):
                    dw[:, :, u, v] = torch.einsum("nohw,nchw->oc", gk, xp[:, :, u:u + h, v:v + w])
            db = gk.sum(dim=(0, 2, 3))
            scale = max(1e-6, float(dw.abs().max()))
            e_f = float((fused[k][0].double().cpu() - dw).abs().max()) / scale
            e_l = float((lw[k][0].double().cpu() - dw).abs().max()) / scale
            tol = 2e-5 if k > 0 else 2e-6        # bf16 products are exact in fp32: only the accumulation differs
            assert e_f <= tol * max(1.0, (n * h * w) ** 0.5 / 16), (k, "fused dW", e_f, "layer-wise", e_l)
            assert e_f <= max(4 * e_l, 1e-6), (k, "fused dW much worse than layer-wise", e_f, e_l)
            sb = max(1e-6, float(db.abs().max()))
            assert float((fused[k][1].double().cpu() - db).abs().max()) / sb <= 2e-5 * max(1.0, (n * h * w) ** 0.5 / 16), (k, "db")


def test_accumulate_and_second_branch_offsets():
    """accumulate adds to the destinations (shared encoders); views at a channel-block offset (the second branch's half)."""
    with dtype_ctx("bf16"):
        eng, T, img, F, GF = _setup(2, 40, 24, 5)
        ws = eng.workspace(torch.device(DEV))
        a = _grads(eng)
        T.dense_encoder_wgrad(img, F.view(0, 6), GF.view(0, 8), a, ws)
        b = [(w.clone(), bb.clone()) for w, bb in a]
        T.dense_encoder_wgrad(img, F.view(0, 6), GF.view(0, 8), b, ws, accumulate=True)
        torch.cuda.synchronize()
        for (w1, b1), (w2, b2) in zip(a, b):
            assert torch.equal(w2, 2 * w1) and torch.equal(b2, 2 * b1)
        # the same data placed in the upper half of 128-channel buffers
        F2 = T.BT.alloc(2, 128, 40, 24, torch.bfloat16, DEV)
        G2 = T.BT.alloc(2, 128, 40, 24, torch.bfloat16, DEV, halo=1, zero=True)
        F2.buf.zero_()
        F2.buf[:, 8:] = F.buf
        G2.buf[:, 8:] = GF.buf
        c = _grads(eng)
        T.dense_encoder_wgrad(img, F2.view(8, 6), G2.as_folded().view(8, 8), c, ws)
        torch.cuda.synchronize()
        for (w1, b1), (w3, b3) in zip(a, c):
            assert torch.equal(w1, w3) and torch.equal(b1, b3)
        # deterministic
        d = _grads(eng)
        T.dense_encoder_wgrad(img, F.view(0, 6), GF.view(0, 8), d, ws)
        torch.cuda.synchronize()
        assert all(torch.equal(x1, x2) and torch.equal(y1, y2) for (x1, y1), (x2, y2) in zip(a, d))


def test_models_train_step_with_fused_wgrad_matches_layerwise():
    import core.model as M
    with dtype_ctx("bf16"):
        for name in ("PFNetv1", "DenseFuse", "VIFNet"):
            torch.manual_seed(5)
            m = getattr(M, name)().to(DEV)
            g = torch.Generator().manual_seed(9)
            i1, i2 = torch.rand(2, 1, 45, 70, generator=g).to(DEV), torch.rand(2, 1, 45, 70, generator=g).to(DEV)
            res = []
            for flag in ("0", "1"):
                os.environ["MMIF_ENC_WGRAD"] = flag
                __import__("mmif.engine").engine.reload_switches()
                os.environ["MMIF_ENC_CHAIN"] = "0"      # same gradient chain in both runs: only the weight-gradient kernels differ
                __import__("mmif.engine").engine.reload_switches()
                try:
                    m.zero_grad(set_to_none=True)
                    m(i1, i2).square().mean().backward()
                    torch.cuda.synchronize()
                    res.append({k: p.grad.clone() for k, p in m.named_parameters()})
                finally:
                    os.environ.pop("MMIF_ENC_WGRAD")
                    __import__("mmif.engine").engine.reload_switches()
                    os.environ.pop("MMIF_ENC_CHAIN")
                    __import__("mmif.engine").engine.reload_switches()
            for k in res[0]:
                a, b = res[0][k].double(), res[1][k].double()
                if "encode" not in k:
                    assert torch.equal(a, b), (name, k)
                else:
                    assert float((a - b).abs().max()) <= 1e-4 * max(1e-6, float(a.abs().max())), (name, k)


def test_gather_form_dgrad_chain_vs_definition_and_scatter_form():
    """DenseBlock backward chain per destination (virtual stacked layers, mmif_pack_dense_chain + one folded dgrad per x_k) against
    the fp64 definition on the same bf16 operands -- g(x_k) = [x_k > 0] * (G_k + sum_{L>k} dgrad_L(g_L)|x_k), reflect-padding adjoint
    included -- and against the layer-by-layer (scatter, read-modify-write) form it replaces."""
    import core.model as M
    from mmif import engine as E
    from mmif import tensor as T
    with dtype_ctx("bf16"):
        for (n, h, w) in ((2, 32, 32), (1, 37, 53), (1, 64, 80)):
            torch.manual_seed(3)
            m = M.PFNetv1().to(DEV)
            eng = E.PFNetv1Engine(m)
            g = torch.Generator().manual_seed(11)
            img = torch.rand(n, 1, h, w, generator=g).to(DEV)
            (img,), _, _, _, dtype, impl = eng.prepare((img,))
            specs = eng.enc[0]
            x = torch.relu(torch.randn(n, 64, h, w, generator=g)).to(DEV)
            G = torch.randn(n, 64, h, w, generator=g).to(DEV)
            F = T.BT.from_nchw(x, torch.bfloat16)
            res = {}
            for mode in ("1", "0"):
                GF = T.BT.from_nchw(G, torch.bfloat16, halo=1).as_folded()
                os.environ["MMIF_ENC_CHAIN"] = mode
                os.environ["MMIF_ENC_CHAIN_STREAM"] = "0"     # (the in-place launches; the streaming kernel: tests/test_gpu_enc_chain.py)
                __import__("mmif.engine").engine.reload_switches()
                try:
                    eng._assign_grad_views(torch.device(DEV))
                    eng.enc_bwd(specs, img, F, GF, 0, 0, eng.workspace(torch.device(DEV)), impl)
                finally:
                    os.environ.pop("MMIF_ENC_CHAIN")
                    os.environ.pop("MMIF_ENC_CHAIN_STREAM")
                    __import__("mmif.engine").engine.reload_switches()
                torch.cuda.synchronize()
                assert float(GF.buf[:, :, 0].float().abs().max()) == 0.0 and float(GF.buf[:, :, :, 0].float().abs().max()) == 0.0   # ring stays zero
                res[mode] = GF.to_nchw(64).double().cpu()
            # fp64 definition on the bf16-rounded operands
            xr = F.to_nchw(64).double().cpu()
            Gr = T.BT.from_nchw(G, torch.bfloat16).to_nchw(64).double().cpu()
            W = [s.conv.weight.detach().bfloat16().double().cpu() for s in specs[1:]]
            gz = [None, None, None, Gr[:, 48:64]]
            for k in (2, 1, 0):
                t = Gr[:, 16 * k:16 * k + 16].clone()
                for L in range(k + 1, 4):
                    xin = xr[:, :16 * L].clone().requires_grad_(True)
                    y = torch.nn.functional.conv2d(torch.nn.functional.pad(xin, (1, 1, 1, 1), mode="reflect"), W[L - 1])
                    y.backward(gz[L])
                    t = t + xin.grad[:, 16 * k:16 * k + 16]
                gz[k] = t * (xr[:, 16 * k:16 * k + 16] > 0)
                # (later layers see the bf16-ROUNDED g_k, as the kernels do)
                gz[k] = gz[k].float().bfloat16().double()
            want = torch.cat(gz, dim=1)
            scale = float(want.abs().max())
            e_gather = float((res["1"] - want).abs().max()) / scale
            e_scatter = float((res["0"] - want).abs().max()) / scale
            assert e_gather <= 1.2e-2, (n, h, w, e_gather)          # one bf16 rounding of the sum (+ its propagation through <= 2 layers)
            assert e_gather <= e_scatter + 1e-3, (e_gather, e_scatter)   # never worse than the form that rounds after every contribution


def test_wgrad_kernels_geometry_fuzz():
    """seeded random shapes: the fused encoder wgrad and the tap-row single-layer wgrad (64->32, 32->16, 48->16) against the
    per-input-group kernels they replace ($MMIF_WGRAD_TAPROW=0 is read once per process, so the reference here is the VALU family)"""
    import random
    from mmif import _lib
    rnd = random.Random(4321)
    shapes = [(rnd.randint(1, 3), rnd.randint(2, 70), rnd.randint(2, 90)) for _ in range(12)]
    with dtype_ctx("bf16"):
        for n, h, w in shapes:
            eng, T, img, F, GF = _setup(n, h, w, 50 + h * w)
            ws = eng.workspace(torch.device(DEV))
            fused = _grads(eng)
            T.dense_encoder_wgrad(img, F.view(0, 6), GF.view(0, 8), fused, ws)
            for k, nin in ((3, 6), (2, 4), (1, 2)):
                dw, db = torch.zeros_like(fused[k][0]), torch.zeros_like(fused[k][1])
                T.conv_wgrad(F.view(0, nin), GF.view(2 * k, 2), dw, db, 16 * k, 16, 3, ws, False, _lib.IMPL_VALU)
                torch.cuda.synchronize()
                scale = max(1e-6, float(dw.abs().max()))
                assert float((fused[k][0] - dw).abs().max()) / scale < 1e-4, (n, h, w, k)
                assert float((fused[k][1] - db).abs().max()) / max(1e-6, float(db.abs().max())) < 1e-4, (n, h, w, k)
            # tap-row kernel (MFMA dispatch) vs the VALU kernel on a 64 -> 32 and a 32 -> 16 layer
            for cin, cout in ((64, 32), (32, 16)):
                g = torch.Generator().manual_seed(h * 7 + w)
                x = T.BT.from_nchw(torch.relu(torch.randn(n, cin, h, w, generator=g)).to(DEV), torch.bfloat16)
                gy = T.BT.from_nchw(torch.randn(n, cout, h, w, generator=g).to(DEV), torch.bfloat16, halo=1).as_folded()
                wsl = torch.empty(T.wgrad_workspace_bytes(cin, cout, 3) // 4 + 1, dtype=torch.float32, device=DEV)
                out = {}
                for impl in (_lib.IMPL_MFMA, _lib.IMPL_VALU):
                    dw, db = torch.zeros(cout, cin, 3, 3, device=DEV), torch.zeros(cout, device=DEV)
                    T.conv_wgrad(x, gy, dw, db, cin, cout, 3, wsl, False, impl)
                    out[impl] = (dw, db)
                torch.cuda.synchronize()
                a, b = out[_lib.IMPL_MFMA], out[_lib.IMPL_VALU]
                assert float((a[0] - b[0]).abs().max()) / max(1e-6, float(b[0].abs().max())) < 1e-4, (n, h, w, cin, cout)
                assert float((a[1] - b[1]).abs().max()) / max(1e-6, float(b[1].abs().max())) < 1e-4, (n, h, w, cin, cout)


def test_fused_encoder_paths_vs_layerwise_all_models_and_modes():
    """every engine that owns a DenseBlock encoder -- PFNetv1, VIFNet (shared encoder, accumulating second branch), DenseFuse (two
    inputs and the auto-encoder call forward(img)), PFNetv2 -- trained for one step with the three fused encoder passes on
    (streaming forward, gather chain, fused weight gradients) and all off: output and parameter gradients within the bf16 rounding
    differences of the streaming forward and the gradient chain."""
    import core.model as M
    flags = ("MMIF_ENC_STREAM", "MMIF_ENC_CHAIN", "MMIF_ENC_WGRAD")
    with dtype_ctx("bf16"):
        for name, single in (("PFNetv1", False), ("VIFNet", False), ("DenseFuse", False), ("DenseFuse", True), ("PFNetv2", False)):
            torch.manual_seed(7)
            m = getattr(M, name)().to(DEV)
            g = torch.Generator().manual_seed(3)
            i1, i2 = torch.rand(2, 1, 40, 56, generator=g).to(DEV), torch.rand(2, 1, 40, 56, generator=g).to(DEV)
            res = []
            for on in ("0", "1"):
                for f in flags:
                    os.environ[f] = on
                try:
                    m.zero_grad(set_to_none=True)
                    y = m(i1) if single else m(i1, i2)
                    (y * y).mean().backward()
                    torch.cuda.synchronize()
                    res.append((y.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}))
                finally:
                    for f in flags:
                        os.environ.pop(f)
            # (round 5: the streaming forward accumulates in another order than the layer-wise kernels -- one bf16 rounding of difference
            # per encoder stage, csrc/enc_stream2.hip --, so the fused image and the decoder's gradients agree to rounding noise, not bits)
            ya, yb = res[0][0].double(), res[1][0].double()
            assert float((ya - yb).abs().max()) <= 2e-2 * float(ya.abs().max()), (name, single)
            for k in res[0][1]:
                a, b = res[0][1][k].double(), res[1][1][k].double()
                scale = max(1e-7, float(a.abs().max()))
                assert float((a - b).abs().max()) <= 2e-2 * scale, (name, single, k, float((a - b).abs().max()) / scale)


def test_densefuse_shared_fused_gradient_is_bit_identical():
    """'sum' fusion: the per-branch copies of d(f1 + f2) are replaced by the chain reading its accumulate operand from the shared
    gradient (mmif_conv2d_reflect_dgrad_folded_onto); $MMIF_FUSE_SHARE=0 restores the copies -- same bits either way"""
    import os
    import core.model as M
    from gpu_util import DEV, dtype_ctx
    with dtype_ctx("bf16"):
        torch.manual_seed(3)
        m = M.DenseFuse().to(DEV)
        g = torch.Generator().manual_seed(4)
        for shape in ((8, 1, 128, 128), (2, 1, 45, 70)):     # the second one is too small for the asynchronous kernel: falls back by itself
            i1, i2 = torch.rand(*shape, generator=g).to(DEV), torch.rand(*shape, generator=g).to(DEV)
            res = []
            for flag in ("0", "1"):
                os.environ["MMIF_FUSE_SHARE"] = flag
                __import__("mmif.engine").engine.reload_switches()
                try:
                    m.zero_grad(set_to_none=True)
                    m(i1, i2).square().mean().backward()
                    torch.cuda.synchronize()
                    res.append({k: p.grad.clone() for k, p in m.named_parameters()})
                finally:
                    os.environ.pop("MMIF_FUSE_SHARE", None)
                    __import__("mmif.engine").engine.reload_switches()
            for k in res[0]:
                assert torch.equal(res[0][k], res[1][k]), f"{shape} {k}"


@pytest.mark.parametrize("n,h,w", [(1, 2, 2), (2, 5, 7), (1, 37, 53), (2, 64, 64), (1, 70, 33), (4, 128, 128)], ids=lambda v: str(v))
def test_fused_wgrad_fp32_vs_layerwise_split_operand_kernels(n, h, w):
    """fp32 tensors: mmif_dense_encoder_wgrad = image_in_wgrad + ONE split-operand pass over [x0 | x1 | x2] / [g1 | g2 | g3]
    (csrc/conv_x3.hip, wgrad_x3_dense_kernel) against the three layer-wise split-operand weight gradients (same two-piece products, only
    the grouping of the pixel sums differs: 2e-6) and against the fp32 FMA kernels (3e-5); accumulate mode; views at a channel-block offset."""
    import core.model as M
    from mmif import engine as E
    from mmif import tensor as T
    from mmif._lib import IMPL_VALU, IMPL_X3
    with dtype_ctx("fp32"):
        torch.manual_seed(3 + h)
        eng = E.PFNetv1Engine(M.PFNetv1().to(DEV))
        g = torch.Generator().manual_seed(h + w)
        img = torch.rand(n, 1, h, w, generator=g).to(DEV)
        x = torch.relu(torch.randn(n, 128, h, w, generator=g) * 0.7).to(DEV)
        gz = (torch.randn(n, 128, h, w, generator=g) * (torch.rand(n, 128, h, w, generator=g) > 0.4)).to(DEV)
        F = T.BT.from_nchw(x, torch.float32)
        GF = T.BT.from_nchw(gz, torch.float32, halo=1).as_folded()
        (img,), *_ = eng.prepare((img,))
        ws = eng.workspace(torch.device(DEV))
        specs = eng.enc[0]
        for base in (0, 8):                               # first / second encoder's half of the 128-channel buffers
            fused = _grads(eng)
            T.dense_encoder_wgrad(img, F.view(base, 6), GF.view(base, 8), fused, ws)
            res = {}
            for impl in (IMPL_X3, IMPL_VALU):
                lw = _grads(eng)
                for k, nin in ((3, 6), (2, 4), (1, 2)):
                    T.conv_wgrad(F.view(base, nin), GF.view(base + 2 * k, 2), lw[k][0], lw[k][1], specs[k].cin, 16, 3, ws, False, impl)
                T.image_in_wgrad(img, GF.view(base, 2), lw[0][0], lw[0][1], 16, 3, ws, False)
                res[impl] = lw
            torch.cuda.synchronize()
            for k in range(4):
                for t in (0, 1):
                    a = fused[k][t].double().cpu()
                    for impl, tol in ((IMPL_X3, 2e-6), (IMPL_VALU, 3e-5)):
                        r = res[impl][k][t].double().cpu()
                        e = float((a - r).abs().max()) / max(1e-6, float(r.abs().max()))
                        assert e <= tol, (base, k, "dW" if t == 0 else "db", impl, e)
            twice = [(a.clone(), b.clone()) for a, b in fused]
            T.dense_encoder_wgrad(img, F.view(base, 6), GF.view(base, 8), twice, ws, accumulate=True)
            torch.cuda.synchronize()
            for (w1, b1), (w2, b2) in zip(fused, twice):
                assert torch.equal(w2, 2 * w1) and torch.equal(b2, 2 * b1)
