"""Streaming DenseBlock encoder (mmif_dense_encoder_fwd): ONE line-buffer launch for ConvLayer(1,16) + DenseBlock(16,16) (reference
core/model.py:73-80, core/block.py:137-151), two kernel generations:

* csrc/enc_stream2.hip (round 5, the default): 64-column strips, input-stationary accumulation.  Same rounding points as the layer-wise
  kernels but another fp32 accumulation ORDER, so it is held to the fp64 DEFINITION of every stage on the kernel's own bf16 inputs at
  one bf16 rounding (as tests/test_gpu_enc_chain.py holds the backward chain) -- x0 included: it runs on the exact-fp32 matrix path --,
  and to the layer-wise launches within one bf16 rounding per stage; both strip widths (32 pixels / eight waves per CU, the default, and
  64 pixels / four waves);
* csrc/enc_stream.hip (round 2; mmif_debug_set_enc_stream2(0) / $MMIF_ENC_STREAM2=0): BIT-IDENTICAL to the four layer-wise launches.

Both within the bf16 bar of the fp32 oracle; shapes cover one strip, two strips (both ghosts' edge strips), interior strips, ragged
strips / segments, 2-row and 2-column images (every row / column is a reflect source) and the BASELINE size."""
import os

import numpy as np
import pytest
import torch

from oracle import fusion_oracle as O
from gpu_util import DEV, bf16_round, dtype_ctx, rel_err, tg

pytestmark = pytest.mark.gpu
ULP = 2.0 ** -8


class gen:
    """with gen(m): ... -- select the streaming kernel for the block: 0 = round 2's (csrc/enc_stream.hip), 1 = round 5's with 64-pixel strips
    (four waves per CU), 2 = round 5's with 32-pixel strips (eight waves per CU; the default)"""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        from mmif._lib import lib
        lib.mmif_debug_set_enc_stream2(self.mode)

    def __exit__(self, *a):
        from mmif._lib import lib
        lib.mmif_debug_set_enc_stream2(2)


SHAPES = [(1, 2, 2), (2, 5, 7), (1, 3, 40), (2, 32, 32), (1, 37, 53), (3, 64, 64), (1, 70, 33), (2, 129, 200), (2, 256, 256), (1, 300, 331)]


def _engine_and_buffers(n, h, w, seed):
    import core.model as M
    from mmif import engine as E
    from mmif import tensor as T
    torch.manual_seed(seed)
    m = M.PFNetv1().to(DEV)
    with torch.no_grad():     # non-zero biases: the epilogue's bias path must be exercised
        for p in m.parameters():
            if p.dim() == 1:
                p.copy_(torch.randn_like(p) * 0.1)
    eng = E.PFNetv1Engine(m)
    g = torch.Generator().manual_seed(seed + 1)
    i1, i2 = torch.rand(n, 1, h, w, generator=g).to(DEV), torch.rand(n, 1, h, w, generator=g).to(DEV)
    (i1, i2), _, _, _, dtype, impl = eng.prepare((i1, i2))
    return eng, T, i1, i2, dtype, impl


@pytest.mark.parametrize("n,h,w", SHAPES, ids=[f"{n}x{h}x{w}" for n, h, w in SHAPES])
def test_stream_equals_layerwise_bit_for_bit(n, h, w):
    """round-2 kernel"""
    with dtype_ctx("bf16"), gen(0):
        eng, T, i1, i2, dtype, impl = _engine_and_buffers(n, h, w, 3 + h)
        Fa = T.BT.alloc(n, 128, h, w, dtype, DEV)
        Fb = T.BT.alloc(n, 128, h, w, dtype, DEV)
        Fa.buf.fill_(7.0)
        Fb.buf.fill_(7.0)
        br = [(eng.enc[0], i1, 0), (eng.enc[1], i2, 8)]
        os.environ["MMIF_ENC_STREAM"] = "0"
        __import__("mmif.engine").engine.reload_switches()
        try:
            eng.enc_fwd_all(br, Fa, dtype, impl)
        finally:
            os.environ.pop("MMIF_ENC_STREAM")
            __import__("mmif.engine").engine.reload_switches()
        eng.enc_fwd_all(br, Fb, dtype, impl)
        torch.cuda.synchronize()
        a, b = Fa.buf.view(torch.int16), Fb.buf.view(torch.int16)
        if not torch.equal(a, b):
            d = (a != b).nonzero()
            raise AssertionError(f"{d.shape[0]} of {a.numel()} elements differ; first at [n, cb, y, x, e] = {d[0].tolist()}, "
                                 f"per channel block: {[(a[:, c] != b[:, c]).sum().item() for c in range(16)]}")
        # one branch alone (the auto-encoder call / DenseFuse's single mode), into the upper half of another buffer
        Fc = T.BT.alloc(n, 128, h, w, dtype, DEV)
        Fc.buf.zero_()
        eng.enc_fwd_all([(eng.enc[1], i2, 8)], Fc, dtype, impl)
        torch.cuda.synchronize()
        assert torch.equal(Fc.buf[:, 8:].view(torch.int16), a[:, 8:]) and float(Fc.buf[:, :8].float().abs().max()) == 0.0


SHAPES2 = SHAPES + [(1, 2, 70), (1, 70, 2), (2, 9, 62), (1, 8, 63), (1, 5, 64), (1, 6, 120), (1, 4, 121), (1, 7, 178), (1, 5, 179), (1, 12, 512)]


def _fp64_stage(inp, wgt, bias):
    return O.conv2d_reflect_fwd(inp.astype(np.float64), bf16_round(wgt).astype(np.float64), bias.astype(np.float64), relu=True)


@pytest.mark.parametrize("mode", [2, 1], ids=["32px", "64px"])
@pytest.mark.parametrize("n,h,w", SHAPES2, ids=[f"{n}x{h}x{w}" for n, h, w in SHAPES2])
def test_stream2_vs_fp64_definition_and_layerwise(n, h, w, mode):
    """round-5 kernel: every stage = bf16(relu(b + conv(reflect_pad(its own stored inputs)))) in fp64 within one rounding; x0 bit-identical
    to the layer-wise first layer; x1..x3 within one bf16 rounding per stage of the layer-wise launches"""
    with dtype_ctx("bf16"), gen(mode):
        eng, T, i1, i2, dtype, impl = _engine_and_buffers(n, h, w, 3 + h)
        Fa = T.BT.alloc(n, 128, h, w, dtype, DEV)
        Fb = T.BT.alloc(n, 128, h, w, dtype, DEV)
        Fa.buf.fill_(7.0)
        Fb.buf.fill_(7.0)
        br = [(eng.enc[0], i1, 0), (eng.enc[1], i2, 8)]
        os.environ["MMIF_ENC_STREAM"] = "0"
        __import__("mmif.engine").engine.reload_switches()
        try:
            eng.enc_fwd_all(br, Fa, dtype, impl)
        finally:
            os.environ.pop("MMIF_ENC_STREAM")
            __import__("mmif.engine").engine.reload_switches()
        eng.enc_fwd_all(br, Fb, dtype, impl)
        torch.cuda.synchronize()
        lay, got = Fa.to_nchw(128).cpu().numpy(), Fb.to_nchw(128).cpu().numpy()
        assert np.isfinite(got).all()
        for e, img in ((0, i1), (1, i2)):
            x = img.cpu().numpy()
            mine = got[:, 64 * e:64 * e + 64]
            # x0 runs on the exact-fp32 matrix path (v_mfma_f32_16x16x4_f32): same products as the layer-wise FMAs, another summation order
            # inside the instruction -- the rounded values agree except where the fp32 sums straddle a bf16 rounding boundary
            d0 = mine[:, :16] != lay[:, 64 * e:64 * e + 16]
            assert d0.mean() < 0.02, f"branch {e}: x0 differs from the layer-wise first layer in {d0.mean():.4f} of its elements"
            for k, sp in enumerate(eng.enc[e]):
                wgt, b = sp.w.detach().cpu().numpy(), sp.b.detach().cpu().numpy()
                if k == 0:
                    ref = O.conv2d_reflect_fwd(x.astype(np.float64), wgt.astype(np.float64), b.astype(np.float64), relu=True)   # fp32 weights, fp32 image
                else:
                    ref = _fp64_stage(mine[:, :16 * k], wgt, b)          # the kernel's own stored inputs
                assert np.abs(ref).max() > 0
                a = mine[:, 16 * k:16 * k + 16]
                err = np.abs(a - ref) / np.maximum(np.abs(ref), 1e-2 * np.abs(ref).max())
                assert err.max() <= 1.01 * ULP, f"branch {e} x{k}: {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)} (got {a.flat[err.argmax()]}, want {ref.flat[err.argmax()]})"
                if k > 0:   # vs the layer-wise launch: its inputs may already differ by one rounding per earlier stage
                    d = np.abs(a - lay[:, 64 * e + 16 * k:64 * e + 16 * k + 16]) / np.abs(ref).max()
                    assert d.max() <= k * 2.0 ** -6, f"branch {e} x{k} vs layer-wise: {d.max():.3e}"
        # one branch alone (the auto-encoder call / DenseFuse's single mode), into the upper half of another buffer: same bits as in the pair
        Fc = T.BT.alloc(n, 128, h, w, dtype, DEV)
        Fc.buf.zero_()
        eng.enc_fwd_all([(eng.enc[1], i2, 8)], Fc, dtype, impl)
        torch.cuda.synchronize()
        assert torch.equal(Fc.buf[:, 8:].view(torch.int16), Fb.buf[:, 8:].view(torch.int16)) and float(Fc.buf[:, :8].float().abs().max()) == 0.0


@pytest.mark.parametrize("mode", [2, 1], ids=["32px", "64px"])
def test_stream2_geometry_fuzz(mode):
    """seeded random shapes through the round-5 kernel's strip / segment / ghost geometry: x0..x3 against the fp64 definition"""
    import random
    rnd = random.Random(4321)
    shapes = [(rnd.randint(1, 3), rnd.randint(2, 90), rnd.randint(2, 260)) for _ in range(20)] + [(1, 2, 2), (1, 3, 61), (1, 2, 65), (4, 31, 119)]
    with dtype_ctx("bf16"), gen(mode):
        for n, h, w in shapes:
            eng, T, i1, i2, dtype, impl = _engine_and_buffers(n, h, w, 100 + h * w)
            F = T.BT.alloc(n, 128, h, w, dtype, DEV)
            F.buf.fill_(7.0)
            eng.enc_fwd_all([(eng.enc[0], i1, 0), (eng.enc[1], i2, 8)], F, dtype, impl)
            torch.cuda.synchronize()
            got = F.to_nchw(128).cpu().numpy()
            for e, img in ((0, i1), (1, i2)):
                mine = got[:, 64 * e:64 * e + 64]
                for k, sp in enumerate(eng.enc[e]):
                    wgt, b = sp.w.detach().cpu().numpy(), sp.b.detach().cpu().numpy()
                    ref = (O.conv2d_reflect_fwd(img.cpu().numpy().astype(np.float64), wgt.astype(np.float64), b.astype(np.float64), relu=True) if k == 0
                           else _fp64_stage(mine[:, :16 * k], wgt, b))
                    err = np.abs(mine[:, 16 * k:16 * k + 16] - ref) / np.maximum(np.abs(ref), 1e-2 * np.abs(ref).max())
                    assert err.max() <= 1.01 * ULP, f"{(n, h, w)} branch {e} x{k}: {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"


@pytest.mark.parametrize("generation", [0, 1, 2])
def test_stream_vs_fp32_oracle(generation):
    """against the numpy oracle of the four layers (fp32): the bf16 storage bar of the layer-wise path (3e-2 of max|.|)"""
    n, h, w = 2, 37, 53
    with dtype_ctx("bf16"), gen(generation):
        eng, T, i1, i2, dtype, impl = _engine_and_buffers(n, h, w, 11)
        F = T.BT.alloc(n, 128, h, w, dtype, DEV)
        eng.enc_fwd_all([(eng.enc[0], i1, 0), (eng.enc[1], i2, 8)], F, dtype, impl)
        got = F.to_nchw(128).float().cpu().numpy()
        for e, img in ((0, i1), (1, i2)):
            x = img.cpu().numpy()
            feats = []
            for k, s in enumerate(eng.enc[e]):
                wgt, b = s.w.detach().cpu().numpy(), s.b.detach().cpu().numpy()
                inp = x if k == 0 else np.concatenate(feats, axis=1)
                feats.append(O.conv2d_reflect_fwd(inp, wgt, b, relu=True))
            want = np.concatenate(feats, axis=1)
            assert rel_err(got[:, 64 * e:64 * e + 64], want) < 3e-2


@pytest.mark.parametrize("generation", [0, 1, 2])
def test_models_use_the_streaming_encoder_and_match_layerwise(generation):
    """whole models (PFNetv1, DenseFuse incl. auto-encoder mode, VIFNet, PFNetv2): forward output and every parameter gradient with the
    streaming encoder on and off -- bit-identical for the round-2 kernel, within the bf16 rounding noise of three stages for round 5's"""
    import core.model as M
    from mmif import tensor as T

    def same(a, b, what):
        if generation == 0:
            assert torch.equal(a, b), what
        else:
            assert float((a - b).abs().max()) <= 2e-2 * float(b.abs().max()) + 1e-12, (what, float((a - b).abs().max()), float(b.abs().max()))

    with dtype_ctx("bf16"), gen(generation):
        for name in ("PFNetv1", "DenseFuse", "VIFNet", "PFNetv2"):
            torch.manual_seed(5)
            m = getattr(M, name)().to(DEV)
            g = torch.Generator().manual_seed(9)
            i1, i2 = torch.rand(2, 1, 45, 70, generator=g).to(DEV), torch.rand(2, 1, 45, 70, generator=g).to(DEV)
            res = []
            for flag in ("0", "1"):
                os.environ["MMIF_ENC_STREAM"] = flag
                __import__("mmif.engine").engine.reload_switches()
                try:
                    T.PROFILE_TAGS.add("encode:fwd")
                    T.PROFILE_EVENTS.pop("encode:fwd", None)
                    m.zero_grad(set_to_none=True)
                    y = m(i1, i2)
                    y.square().mean().backward()
                    torch.cuda.synchronize()
                    used = len(T.PROFILE_EVENTS.get("encode:fwd", []))
                    res.append((y.detach().clone(), [p.grad.clone() for p in m.parameters()], used))
                finally:
                    os.environ.pop("MMIF_ENC_STREAM")
                    __import__("mmif.engine").engine.reload_switches()
                    T.PROFILE_TAGS.discard("encode:fwd")
            assert res[0][2] == 0 and res[1][2] == 1, "the streaming launch must run exactly when enabled"
            same(res[1][0], res[0][0], name)
            for a, b in zip(res[1][1], res[0][1]):
                same(a, b, name)
            if name == "DenseFuse":
                with torch.no_grad():
                    os.environ["MMIF_ENC_STREAM"] = "0"
                    __import__("mmif.engine").engine.reload_switches()
                    y0 = m(i1)
                    os.environ["MMIF_ENC_STREAM"] = "1"
                    __import__("mmif.engine").engine.reload_switches()
                    y1 = m(i1)
                    os.environ.pop("MMIF_ENC_STREAM")
                    __import__("mmif.engine").engine.reload_switches()
                same(y1, y0, "auto-encoder mode")


def test_argument_validation():
    import ctypes as C
    from mmif import _lib
    e = _lib.MmifDenseEncoder()
    t = _lib.MmifTensor(None, 1, 1, 8, 8, 0, 8, 0, 8, 0)
    assert _lib.lib.mmif_dense_encoder_fwd(C.byref(e), C.byref(t), None, None, None) != 0
    assert b"dense_encoder_fwd" in _lib.lib.mmif_last_error()


def test_stream_geometry_fuzz():
    """seeded random shapes (strip / segment / ragged-edge geometry of the round-2 streaming kernel): bit-identical to the layer-wise path"""
    import random
    rnd = random.Random(1234)
    shapes = [(rnd.randint(1, 3), rnd.randint(2, 90), rnd.randint(2, 140)) for _ in range(24)] + [(1, 2, 33), (1, 33, 2), (1, 3, 58), (5, 31, 59)]
    with dtype_ctx("bf16"), gen(0):
        for n, h, w in shapes:
            eng, T, i1, i2, dtype, impl = _engine_and_buffers(n, h, w, 100 + h * w)
            Fa, Fb = T.BT.alloc(n, 128, h, w, dtype, DEV), T.BT.alloc(n, 128, h, w, dtype, DEV)
            br = [(eng.enc[0], i1, 0), (eng.enc[1], i2, 8)]
            os.environ["MMIF_ENC_STREAM"] = "0"
            __import__("mmif.engine").engine.reload_switches()
            try:
                eng.enc_fwd_all(br, Fa, dtype, impl)
            finally:
                os.environ.pop("MMIF_ENC_STREAM")
                __import__("mmif.engine").engine.reload_switches()
            eng.enc_fwd_all(br, Fb, dtype, impl)
            torch.cuda.synchronize()
            assert torch.equal(Fa.buf.view(torch.int16), Fb.buf.view(torch.int16)), (n, h, w)


SHAPES_SUM = [(1, 2, 2), (2, 5, 7), (1, 3, 30), (2, 32, 32), (1, 37, 53), (3, 64, 64), (1, 70, 33), (2, 129, 200), (2, 256, 256), (1, 300, 331), (1, 6, 58),
              (1, 9, 31), (1, 4, 29), (2, 7, 84), (1, 12, 512)]


@pytest.mark.parametrize("n,h,w", SHAPES_SUM, ids=[f"{n}x{h}x{w}" for n, h, w in SHAPES_SUM])
def test_shared_encoder_plus_sum_in_one_launch(n, h, w):
    """Round 6, DenseFuse (reference core/model.py:165-186, core/fusion.py:21-29): mmif_dense_encoder_fwd_sum -- both images of a pair through
    ONE shared encoder and f1 + f2, one launch (csrc/enc_stream2.hip's dual form: a wave carries the same 32-column strip of both images).
    Against the two-branch encoder launch (32-pixel strips: the same strips, the same MFMA sequence per tile) + mmif_fuse_elem_fwd: the two
    branches' 64 channels and the sum bit for bit; nothing outside the three views is written."""
    import core.model as M
    from mmif import engine as E
    from mmif import tensor as T
    from mmif._lib import FUSE_SUM
    with dtype_ctx("bf16"), gen(2):
        torch.manual_seed(11 + h)
        m = M.DenseFuse().to(DEV)
        with torch.no_grad():
            for p in m.parameters():
                if p.dim() == 1:
                    p.copy_(torch.randn_like(p) * 0.1)
        eng = E.DenseFuseEngine(m)
        g = torch.Generator().manual_seed(h * 7 + w)
        i1, i2 = torch.rand(n, 1, h, w, generator=g).to(DEV), torch.rand(n, 1, h, w, generator=g).to(DEV)
        (i1, i2), _, _, _, dtype, impl = eng.prepare((i1, i2))
        Fa, Fb = T.BT.alloc(n, 128, h, w, dtype, DEV), T.BT.alloc(n, 128 + 16, h, w, dtype, DEV)
        Sa, Sb = T.BT.alloc(n, 64, h, w, dtype, DEV), T.BT.alloc(n, 64 + 16, h, w, dtype, DEV)
        for t in (Fa, Fb, Sa, Sb):
            t.buf.fill_(7.0)
        eng.enc_fwd_all([(eng.enc, i1, 0), (eng.enc, i2, 8)], Fa, dtype, impl)
        T.fuse_elem_fwd(Fa.view(0, 8), Fa.view(8, 8), Sa, FUSE_SUM)
        # the fused launch into views with a guard block on either side
        assert eng.enc_fwd_sum(i1, i2, Fb.view(1, 16), Sb.view(1, 8), dtype, impl)
        torch.cuda.synchronize()
        a, b = Fa.buf.view(torch.int16), Fb.buf.view(torch.int16)
        for name, x, y in (("branch a", a[:, :8], b[:, 1:9]), ("branch b", a[:, 8:], b[:, 9:17]), ("sum", Sa.buf.view(torch.int16), Sb.buf.view(torch.int16)[:, 1:9])):
            if not torch.equal(x, y):
                d = (x != y).nonzero()
                raise AssertionError(f"{name}: {d.shape[0]} of {x.numel()} elements differ; first at [n, cb, y, x, e] = {d[0].tolist()}, "
                                     f"per channel block: {[(x[:, c] != y[:, c]).sum().item() for c in range(x.shape[1])]}")
        assert float(Sa.buf.float().abs().max()) > 0
        for guard in (Fb.buf[:, 0], Fb.buf[:, 17], Sb.buf[:, 0], Sb.buf[:, 9]):
            assert bool((guard.float() == 7.0).all()), "a guard block was written"


@pytest.mark.parametrize("var", ["MMIF_ENC_SUM", "MMIF_DGRAD_DUP"])
def test_densefuse_step_with_and_without_the_fused_sum(monkeypatch, var):
    """DenseFuse train step (bf16), $MMIF_ENC_SUM = 1 / 0 (the sum inside the encoder launch) and $MMIF_DGRAD_DUP = 1 / 0 (the masked copies of
    decode.0's input gradient inside its dgrad): bit-identical fused image, losses and gradients"""
    import core.model as M
    from core.loss import FusionLoss, GradLoss, PixelLoss, SSIMLoss, unit_gradient
    from gpu_util import reload_switches
    g = torch.Generator().manual_seed(4)
    a, b = torch.rand(3, 1, 72, 104, generator=g).to(DEV), torch.rand(3, 1, 72, 104, generator=g).to(DEV)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv(var, mode)
        reload_switches()
        with dtype_ctx("bf16"):
            torch.manual_seed(0)
            m = M.DenseFuse().to(DEV)
            l_all = FusionLoss(SSIMLoss('ssim', weight=1.0), PixelLoss('l1', weight=0.01), GradLoss('l1', weight=0.1).to(DEV), 'max', 'max')
            f = m(a, b)
            tot = l_all(a, b, f)
            tot.backward(unit_gradient(tot))
            torch.cuda.synchronize()
            res[mode] = (f.detach().cpu().clone(), tot.detach().item(), {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()})
    monkeypatch.delenv(var)
    reload_switches()
    assert torch.equal(res["1"][0], res["0"][0]) and res["1"][1] == res["0"][1]
    for k, g1 in res["1"][2].items():
        assert float(g1.abs().max()) > 0 and torch.equal(g1, res["0"][2][k]), k
