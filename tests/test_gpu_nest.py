"""Row a5 (NestFuse / RFN-Nest, reference core/model.py:319-384, core/block.py:708-759,836-867,965-991) -- parity that cannot
pass on zeros.

* the blocked-layout kernels of csrc/nest.hip that only NestEngine launches (2x2 max-pool, nearest-x2 up-sampling + reflect pad to the
  skip's shape, ReLU masks riding in their backward, attention fusion, RFN's residual add), fp32 AND bf16, against golden F4 (the
  reference's outputs) and against the oracle on the same bf16-rounded operands;
* the stand-alone blocks ConvBlock / RFN / NestDecoder (odd pyramid 37x53 -> 18x26 -> 9x13 -> 4x6) against golden F4;
* the two models end to end on the LIVE closed-form parameter set (oracle.LIVE_PARAMS: the final ReLU passes 30-60 % of the pixels;
  with closed-form seed 1 it is dead everywhere and every comparison is 0 == 0): fp32 against golden F5 at 1x32x32 and 2x36x44, bf16
  forward + every parameter gradient against the oracle's bf16-storage emulation at 2x36x44 and 1x64x64.
Every reference tensor is asserted alive before it is compared (gpu_util.close refuses an all-zero reference)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import fusion_oracle as O
from gpu_util import G, bf16_round, close, close_digest, dtype_ctx, load_closed_form, load_live, tg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TD = {"fp32": torch.float32, "bf16": torch.bfloat16}
ULP = 2.0 ** -8      # one bf16 rounding of a value (relative)


def _q(a, dtype):
    return bf16_round(a) if dtype == "bf16" else np.asarray(a, np.float32)


def _bt(a, dtype, halo=0):
    from mmif.tensor import BT
    return BT.from_nchw(tg(a), TD[dtype], halo)


def _alloc(shape, dtype, halo=0):
    from mmif.tensor import BT
    n, c, h, w = shape
    return BT.alloc(n, c, h, w, TD[dtype], DEV, halo, zero=True)


def _np(bt, c=None):
    torch.cuda.synchronize()
    return bt.to_nchw(c).cpu().numpy()


# ------------------------------------------------------------------------------------------------ nest.hip kernels
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_maxpool_blocked_vs_golden(dtype):
    """mmif_maxpool2x2_fwd / _bwd / _bwd_relu on blocked buffers (nn.MaxPool2d(2, 2) at core/model.py:332-335).  Selection and routing
    move values without arithmetic: exact in both dtypes (bf16: on the rounded operands)."""
    from mmif import tensor as T
    ref = np.load(os.path.join(G, "f4_blocks.npz"))
    xs = (1, 8, 10, 14)
    x, gy = O.closed_form_signed(xs, 0.77), O.closed_form_signed((1, 8, 5, 7), 0.31)
    xb = _bt(x, dtype)
    y = _alloc((1, 8, 5, 7), dtype)
    T.maxpool_fwd(xb, y)
    yq, idx = O.maxpool2x2_fwd(_q(x, dtype))
    assert np.array_equal(_np(y), yq)
    want = O.maxpool2x2_bwd(_q(gy, dtype), idx, xs)
    if dtype == "fp32":
        assert np.array_equal(_np(y), ref["maxpool__y"])
        assert np.array_equal(want, ref["maxpool__dx"])       # (the oracle is the reference here, bit for bit)
    gx = _alloc(xs, dtype, halo=1)
    T.maxpool_bwd(xb, _bt(gy, dtype, halo=1), gx, False)
    assert np.array_equal(_np(gx), want) and np.abs(want).max() > 0
    # accumulate onto an earlier contribution, and the ReLU mask of x riding in the last contribution
    old = O.closed_form_signed(xs, 0.11)
    for relu in (False, True):
        gx = _bt(old, dtype, halo=1)
        T.maxpool_bwd(xb, _bt(gy, dtype, halo=1), gx, True, relu=relu)
        w2 = _q(_q(old, dtype) + want, dtype)
        if relu:
            w2 = w2 * (_q(x, dtype) > 0)
        close(_np(gx), w2, 0 if dtype == "fp32" else ULP, f"accumulate relu={relu}")
    # odd input size: the last row / column is never pooled and gets no gradient
    xo = O.closed_form_signed((2, 16, 9, 11), 0.5)
    yo = _alloc((2, 16, 4, 5), dtype)
    T.maxpool_fwd(_bt(xo, dtype), yo)
    yq, idx = O.maxpool2x2_fwd(_q(xo, dtype))
    assert np.array_equal(_np(yo), yq)
    go = O.closed_form_signed((2, 16, 4, 5), 0.9)
    gxo = _alloc((2, 16, 9, 11), dtype, halo=1)
    T.maxpool_bwd(_bt(xo, dtype), _bt(go, dtype, halo=1), gxo, False)
    assert np.array_equal(_np(gxo), O.maxpool2x2_bwd(_q(go, dtype), idx, xo.shape))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_upsample_blocked_vs_golden(dtype):
    """mmif_upsample2x_fwd / _bwd / _bwd_relu: nearest x2 then reflect pad to the skip's shape (Upsample.forward / _pad,
    core/block.py:975-991); 4x6 -> 9x13 pads one row at the bottom and one column on the right.  Forward copies (exact); backward sums
    up to nine gradient values per source pixel in fp32 and stores once."""
    from mmif import tensor as T
    ref = np.load(os.path.join(G, "f4_blocks.npz"))
    xs, ys = (1, 8, 4, 6), (1, 8, 9, 13)
    x, gy = O.closed_form_signed(xs, 0.57), O.closed_form_signed(ys, 0.41)
    y = _alloc(ys, dtype)
    T.upsample_fwd(_bt(x, dtype), y)
    assert np.array_equal(_np(y), O.upsample_nearest2x_fwd(_q(x, dtype), ys[2:]))
    if dtype == "fp32":
        assert np.array_equal(_np(y), ref["upsample__y"])
    want = O.upsample_nearest2x_bwd(_q(gy, dtype), xs)
    gx = _alloc(xs, dtype, halo=1)
    T.upsample_bwd(_bt(gy, dtype, halo=1), gx, False)
    if dtype == "fp32":
        close(_np(gx), ref["upsample__dx"], 1e-6, "dx vs golden")
    close(_np(gx), _q(want, dtype), 1e-6 if dtype == "fp32" else ULP, "dx")
    act = O.closed_form_signed(xs, 0.23)            # a "ReLU output" with zeros: the mask of the tensor whose gradient this is
    act = np.maximum(act, 0)
    old = O.closed_form_signed(xs, 0.19)
    gx = _bt(old, dtype, halo=1)
    T.upsample_bwd(_bt(gy, dtype, halo=1), gx, True, relu_of=_bt(act, dtype))
    w2 = (_q(old, dtype) + want) * (act > 0)
    assert (act > 0).any() and (act == 0).any()
    close(_np(gx), _q(w2, dtype), 1e-6 if dtype == "fp32" else ULP, "accumulate + mask")
    # even target (no pad) and a two-sample, 16-channel case
    x2, g2 = O.closed_form_signed((2, 16, 5, 7), 0.3), O.closed_form_signed((2, 16, 10, 14), 0.7)
    y2 = _alloc((2, 16, 10, 14), dtype)
    T.upsample_fwd(_bt(x2, dtype), y2)
    assert np.array_equal(_np(y2), O.upsample_nearest2x_fwd(_q(x2, dtype), (10, 14)))
    gx2 = _alloc((2, 16, 5, 7), dtype, halo=1)
    T.upsample_bwd(_bt(g2, dtype, halo=1), gx2, False)
    close(_np(gx2), _q(O.upsample_nearest2x_bwd(_q(g2, dtype), x2.shape), dtype), 1e-6 if dtype == "fp32" else ULP, "dx even")


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_relu_mask_and_residual_add_blocked(dtype):
    """mmif_relu_mask (level-3 encoder gradient) and RFN's residual add y = t2 + res with both addends' ReLU masks in its backward
    (core/block.py:753-759)."""
    from mmif import _lib
    from mmif import tensor as T
    s = (2, 24, 6, 10)
    a, b = np.maximum(O.closed_form_signed(s, 0.15), 0), np.maximum(O.closed_form_signed(s, 1.25), 0)
    g = O.closed_form_signed(s, 2.35)
    gb = _bt(g, dtype, halo=1)
    T.relu_mask_(_bt(a, dtype), gb)
    assert np.array_equal(_np(gb), _q(g, dtype) * (a > 0)) and (a == 0).any()
    out = _alloc(s, dtype)
    T.fuse_elem_fwd(_bt(a, dtype), _bt(b, dtype), out, _lib.FUSE_SUM)
    close(_np(out), _q(_q(a, dtype) + _q(b, dtype), dtype), 0 if dtype == "fp32" else ULP, "sum")
    ga, gb2 = _alloc(s, dtype, halo=1), _alloc(s, dtype, halo=1)
    T.fuse_elem_bwd(_bt(a, dtype), _bt(b, dtype), _bt(g, dtype, halo=1), ga, gb2, _lib.FUSE_SUM, True)
    assert np.array_equal(_np(ga), _q(g, dtype) * (a > 0)) and np.array_equal(_np(gb2), _q(g, dtype) * (b > 0))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("mode", ["sa", "ca", "sca"])
def test_attention_fusion_blocked_vs_golden(mode, dtype):
    """mmif_fuse_attn_fwd / _bwd / _bwd_cached on blocked buffers (attention_fusion, core/fusion.py:42-59, with 'l1' spatial and 'avg'
    channel pooling, :84-153): fp32 against golden F4 (signed and post-ReLU features), bf16 against the oracle on the rounded operands
    -- one output rounding."""
    from mmif import tensor as T
    ref = np.load(os.path.join(G, "f4_blocks.npz"))
    s = (2, 16, 6, 10)
    a0, b0, gy = O.closed_form_signed(s, 0.15), O.closed_form_signed(s, 1.25), O.closed_form_signed(s, 2.35)
    cases = [("attn_" + mode, a0, b0)] + ([("attn_relu", np.maximum(a0, 0), np.maximum(b0, 0))] if mode == "sca" else [])
    for tag, a, b in cases:
        aq, bq, gq = _q(a, dtype), _q(b, dtype), _q(gy, dtype)
        y_or = O.attention_fusion(aq, bq, mode)
        da_or, db_or = O.attention_fusion_bwd(aq, bq, gq, mode)
        ab, bb = _bt(a, dtype), _bt(b, dtype)
        out = _alloc(s, dtype)
        ws = T.attn_workspace(s[0], s[1], DEV)
        T.attn_fwd(ab, bb, out, T.ATTN_MODES[mode], ws)
        ytol, gtol = (1e-5, 5e-5) if dtype == "fp32" else (ULP, 2 * ULP)
        close(_np(out), y_or, ytol, tag + " y vs oracle")
        for cached in (True, False):
            ga, gb = _alloc(s, dtype, halo=1), _alloc(s, dtype, halo=1)
            T.attn_bwd(ab, bb, _bt(gy, dtype, halo=1), ga, gb, T.ATTN_MODES[mode], False, ws, cached=cached)
            close(_np(ga), da_or, gtol, f"{tag} da cached={cached}")
            close(_np(gb), db_or, gtol, f"{tag} db cached={cached}")
            if dtype == "fp32":
                close(_np(out), ref[tag + "__y"], 1e-4, tag + " y vs golden")
                close(_np(ga), ref[tag + "__da"], 2e-4, tag + " da vs golden")
                close(_np(gb), ref[tag + "__db"], 2e-4, tag + " db vs golden")
        # accumulate onto an earlier contribution
        old = O.closed_form_signed(s, 0.66)
        ga, gb = _bt(old, dtype, halo=1), _bt(old, dtype, halo=1)
        T.attn_bwd(ab, bb, _bt(gy, dtype, halo=1), ga, gb, T.ATTN_MODES[mode], True, ws, cached=False)
        close(_np(ga), _q(old, dtype) + da_or, gtol if dtype == "fp32" else 2 * ULP, tag + " da accumulate")


# ------------------------------------------------------------------------------------------------ stand-alone blocks vs golden F4
def _run_block(mod, inputs, gphase):
    xs = [tg(a).requires_grad_(True) for a in inputs]
    y = mod(*xs)
    y.backward(tg(O.closed_form_signed(tuple(y.shape), gphase, 1.0)))
    torch.cuda.synchronize()
    return y.detach().cpu().numpy(), [x.grad.cpu().numpy() for x in xs]


def _check_block(ref, prefix, mod, y, dxs):
    close(y, ref[prefix + "__y"], 1e-4, prefix + " y")
    for i, dx in enumerate(dxs):
        close(dx, ref[f"{prefix}__dx{i}"], 2e-4, f"{prefix} dx{i}")
    keys = [k for k in ref.files if k.startswith(prefix + "__dp_")]
    P = dict(mod.named_parameters())
    assert keys and len(keys) == len(P)
    for k in keys:
        close_digest(P[k[len(prefix) + 5:]].grad.cpu().numpy(), ref[k], 2e-4, k)


@pytest.mark.parametrize("impl", ["valu", "auto"], ids=["fp32-fma", "x3"])
def test_convblock_rfn_nestdecoder_vs_golden(impl):
    """ConvBlock (3x3 -> 1x1, core/block.py:708-722), RFN (:737-759) and NestDecoder with nearest up-sampling on the odd pyramid
    37x53 / 18x26 / 9x13 / 4x6 (:836-867, Upsample._pad :981-991) on the HIP kernels, layer by layer, against the reference's outputs,
    input gradients and parameter-gradient digests (golden F4) -- fp32 FMA kernels and the split-operand matrix-pipe kernels."""
    import core.block as B
    ref = np.load(os.path.join(G, "f4_blocks.npz"))
    with dtype_ctx("fp32", impl):
        m = load_closed_form(B.ConvBlock(16, 64), 5).to(DEV)
        _check_block(ref, "convblock", m, *_run_block(m, [O.closed_form_signed((1, 16, 9, 11), 0.4)], 0.9))
        m = load_closed_form(B.RFN(16), 6).to(DEV)
        _check_block(ref, "rfn", m, *_run_block(m, [O.closed_form_signed((1, 16, 8, 10), 0.6), O.closed_form_signed((1, 16, 8, 10), 1.6)], 1.0))
        ch = [8, 16, 24, 32]
        dec = load_closed_form(B.NestDecoder(B.ConvBlock, ch, "nearest"), 7).to(DEV)
        sizes = [(37, 53), (18, 26), (9, 13), (4, 6)]
        feats = [tg(O.closed_form_signed((1, c, h, w), 0.2 * i + 0.1)).requires_grad_(True) for i, (c, (h, w)) in enumerate(zip(ch, sizes))]
        y = dec(feats)
        y.backward(tg(O.closed_form_signed(tuple(y.shape), 1.1)))
        torch.cuda.synchronize()
        close(y.detach().cpu().numpy(), ref["nestdec__y"], 1e-4, "nestdec y")
        for i, f in enumerate(feats):
            close(f.grad.cpu().numpy(), ref[f"nestdec__dx{i}"], 2e-4, f"nestdec dx{i}")
        for k, p in dec.named_parameters():
            close_digest(p.grad.cpu().numpy(), ref["nestdec__dp_" + k], 2e-4, k)


# ------------------------------------------------------------------------------------------------ the models, live parameters
LIVE_CASES = [("NestFuse", (1, 1, 32, 32)), ("RFNNest", (1, 1, 32, 32)), ("NestFuse", (2, 1, 36, 44)), ("RFNNest", (2, 1, 36, 44))]


def _live_model(name):
    import core.model as M
    return load_live(getattr(M, name)(), name).to(DEV)


@pytest.mark.parametrize("impl", ["valu", "auto"], ids=["fp32-fma", "x3"])
@pytest.mark.parametrize("name,shape", LIVE_CASES, ids=[f"{n}-{s[0]}x{s[2]}x{s[3]}" for n, s in LIVE_CASES])
def test_nest_models_fp32_vs_golden_live(name, shape, impl):
    """NestEngine (fused: blocked buffers, HIP pool / up-sample / attention / RFN adds, zero-copy concats) against the REFERENCE's fused
    image and all 44 / 92 parameter-gradient digests (golden F5, live parameter set: 31-56 % of the output pixels pass the final ReLU;
    smooth positive upstream gradient, see make_golden.py).  2x36x44 walks the odd pyramid 36x44 -> 18x22 -> 9x11 -> 4x5.  Gradient bar
    1e-3 for both kernel families: what is left above summation order (1e-6 ... 6e-5 measured) is a ReLU / max-pool decision taken the
    other way on a value within rounding of a tie, which moves a sum over ~3000 positive terms by 1 / 3000."""
    ref = np.load(os.path.join(G, "f5_models.npz"))
    man = json.load(open(os.path.join(G, "f5_manifest.json")))
    tag = f"{name}_{shape[0]}x{shape[2]}x{shape[3]}"
    O.assert_alive(ref[tag + "__y"], tag, 0.3, 0.7)
    with dtype_ctx("fp32", impl):
        m = _live_model(name)
        assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == man[name]
        assert m._make_engine() is not None
        y = m(tg(O.closed_form_image(shape, 0.3)), tg(O.closed_form_image(shape, 1.7)))
        y.backward(tg(O.closed_form_image(shape, 0.9)))
        torch.cuda.synchronize()
        close(y.detach().cpu().numpy(), ref[tag + "__y"], 2e-4, "imgf")
        for k, p in m.named_parameters():
            close_digest(p.grad.cpu().numpy(), ref[f"{tag}__dp_{k}"], 1e-3, k)


BF16_CASES = [("NestFuse", (2, 1, 36, 44)), ("RFNNest", (2, 1, 36, 44)), ("NestFuse", (1, 1, 64, 64)), ("RFNNest", (1, 1, 64, 64))]


def _rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-300)))


def _cos(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float((a * b).sum() / max(np.sqrt((a * a).sum() * (b * b).sum()), 1e-300))


@pytest.mark.parametrize("name,shape", BF16_CASES, ids=[f"{n}-{s[0]}x{s[2]}x{s[3]}" for n, s in BF16_CASES])
def test_nest_models_bf16_vs_bf16_storage_oracle(name, shape):
    """The bf16 NestEngine -- what config 4's pairs/s is measured on -- forward AND backward against the oracle, live parameters, smooth
    positive upstream gradient.  These nets are NOISY under bf16 storage: the oracle's own bf16-storage emulation (every feature map,
    the attention / RFN fusion outputs, every complete activation gradient and the matrix-pipe layers' weights rounded where the engine
    stores them) sits 4.4-8e-2 of max|y| from its fp32 run on the fused image and 1.6-3 % (median over parameters, relative L2; up to
    17 % on RFN-Nest's first layers) on the gradients -- up to 20 layers of 0.4 % roundings, three max-pools and a ReLU per layer that
    turn a rounding into a decision -- and two bf16 runs with different fp32 summation orders (engine, emulation) decorrelate the same
    way on the fused image (its last layer sums 64 cancelling terms: a 0.4 % rounding of x1_3 is 4 % of max|y|).  The GRADIENTS are the
    sharp check: against the emulation the engine measures 1e-3 ... 5e-3 median relative L2 over the parameters (worst 2.1e-2, cosine
    >= 0.9998) -- held to median 1.5e-2, every parameter relative L2 <= 6e-2 and cosine >= 0.995; against the fp32 oracle to the
    emulation's own distance (every parameter cosine >= 0.95 and relative L2 <= 0.35, median <= 8e-2).  Fused image: max error 9e-2 /
    relative L2 8e-2 of both (measured 2.8-5.6e-2 / 0.9-4.9e-2 vs the emulation), ReLU mask of the output equal on >= 96 % of the
    pixels.  A plumbing error (wrong slot, view, permuted weight, missing contribution) is O(1) on all of these; the kernels themselves
    are pinned bit-tight above and in tests/test_gpu_conv.py."""
    om = O.MODELS[name]()
    P = om.init_params_live()
    i1n, i2n, gn = O.closed_form_image(shape, 0.3), O.closed_form_image(shape, 1.7), O.closed_form_image(shape, 0.9)
    y_fp32 = om.forward(P, i1n, i2n)
    G_fp32 = om.backward(P, gn)
    with O.bf16_storage():
        y_or = om.forward(P, i1n, i2n)
        G_or = om.backward(P, gn)
    O.assert_alive(y_or, name, 0.25, 0.75)
    with dtype_ctx("bf16", "mfma"):
        m = _live_model(name)
        assert m._make_engine() is not None
        y = m(tg(i1n), tg(i2n))
        y.backward(tg(gn))
        torch.cuda.synchronize()
        yn = y.detach().cpu().numpy()
        grads = {k: p.grad.cpu().numpy() for k, p in m.named_parameters()}
    rep = []
    for what, yr, Gr in (("bf16-storage oracle", y_or, G_or), ("fp32 oracle", y_fp32, G_fp32)):
        ey = close(yn, yr, 9e-2, f"imgf vs {what}")
        l2y = _rel_l2(yn, yr)
        assert l2y <= 8e-2, (what, l2y)
        assert ((yn > 0) != (yr > 0)).mean() < 0.04, what
        l2s, worst_cos = [], ("", 1.0)
        for k, g in grads.items():
            O.assert_alive(Gr[k], k)
            c, l2 = _cos(g, Gr[k]), _rel_l2(g, Gr[k])
            emu = what.startswith("bf16")
            assert c >= (0.995 if emu else 0.95) and l2 <= (6e-2 if emu else 0.35), f"{k} vs {what}: cosine {c:.4f}, relative L2 {l2:.3e}"
            l2s.append(l2)
            worst_cos = min(worst_cos, (k, c), key=lambda t: t[1])
        assert np.median(l2s) <= (1.5e-2 if what.startswith("bf16") else 8e-2), (what, np.median(l2s))
        rep.append(f"vs {what}: imgf max {ey:.2e} l2 {l2y:.2e}; gradients median l2 {np.median(l2s):.2e} max l2 {max(l2s):.2e} min cos {worst_cos[1]:.4f} ({worst_cos[0]})")
    print(f"{name} {shape}: " + " | ".join(rep))
