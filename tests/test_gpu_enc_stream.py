"""Streaming DenseBlock encoder (csrc/enc_stream.hip, mmif_dense_encoder_fwd): ONE line-buffer launch for ConvLayer(1,16) +
DenseBlock(16,16) (reference core/model.py:73-80, core/block.py:137-151) must be BIT-IDENTICAL to the four layer-wise launches
(same operand images, k-group order, rounding points) and within the bf16 bar of the fp32 oracle; shapes cover one strip (w < 32),
ragged strips / segments, odd sizes and the BASELINE size."""
import os

import numpy as np
import pytest
import torch

from oracle import fusion_oracle as O
from gpu_util import DEV, dtype_ctx, rel_err, tg

pytestmark = pytest.mark.gpu

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
    with dtype_ctx("bf16"):
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


def test_stream_vs_fp32_oracle():
    """against the numpy oracle of the four layers (fp32): the bf16 storage bar of the layer-wise path (3e-2 of max|.|)"""
    n, h, w = 2, 37, 53
    with dtype_ctx("bf16"):
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


def test_models_use_the_streaming_encoder_and_match_layerwise():
    """whole models (PFNetv1, DenseFuse incl. auto-encoder mode, VIFNet, PFNetv2): forward output and every parameter gradient are
    bit-identical with the streaming encoder on and off"""
    import core.model as M
    from mmif import tensor as T
    with dtype_ctx("bf16"):
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
            assert torch.equal(res[0][0], res[1][0]), name
            for a, b in zip(res[0][1], res[1][1]):
                assert torch.equal(a, b), name
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
                assert torch.equal(y0, y1)


def test_argument_validation():
    import ctypes as C
    from mmif import _lib
    e = _lib.MmifDenseEncoder()
    t = _lib.MmifTensor(None, 1, 1, 8, 8, 0, 8, 0, 8, 0)
    assert _lib.lib.mmif_dense_encoder_fwd(C.byref(e), C.byref(t), None, None, None) != 0
    assert b"dense_encoder_fwd" in _lib.lib.mmif_last_error()


def test_stream_geometry_fuzz():
    """seeded random shapes (strip / segment / ragged-edge geometry of the streaming kernel): bit-identical to the layer-wise path"""
    import random
    rnd = random.Random(1234)
    shapes = [(rnd.randint(1, 3), rnd.randint(2, 90), rnd.randint(2, 140)) for _ in range(24)] + [(1, 2, 33), (1, 33, 2), (1, 3, 58), (5, 31, 59)]
    with dtype_ctx("bf16"):
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
