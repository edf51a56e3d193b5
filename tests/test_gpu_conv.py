"""Per-layer parity of the HIP ConvLayer (forward, dgrad, wgrad) -- through the C ABI, via the
reference-shaped core.block.ConvLayer -- against the golden vectors (fp32) and against the CPU
oracle evaluated on bf16-rounded operands (bf16, VALU and MFMA kernel families)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import fusion_oracle as O
from gpu_util import G, bf16_round, close, dtype_ctx, load_closed_form, tg
from test_oracle_golden import F3_CASES, f3_tensors

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

import os


def _run_layer(case):
    from core.block import ConvLayer
    name, cin, cout, k, relu, N, H, W = case
    x, w, b, gy = f3_tensors(case)
    layer = load_closed_form(ConvLayer(cin, cout, ksize=k, act=nn.ReLU if relu else None), 3).to("cuda:0")
    xt = tg(x).requires_grad_(cin > 1)
    y = layer(xt)
    y.backward(tg(gy))
    torch.cuda.synchronize()
    dx = xt.grad.cpu().numpy() if cin > 1 else None
    return (y.detach().cpu().numpy(), dx, layer.layers[0].weight.grad.cpu().numpy(), layer.layers[0].bias.grad.cpu().numpy())


@pytest.mark.parametrize("case", F3_CASES, ids=[c[0] for c in F3_CASES])
def test_conv_fp32_vs_golden(case):
    """fp32 storage, VALU kernels: the north-star parity bar is 1e-3 relative; we hold 1e-4."""
    ref = np.load(os.path.join(G, "f3_conv.npz"))
    name = case[0]
    with dtype_ctx("fp32"):
        y, dx, dw, db = _run_layer(case)
    close(y, ref[name + "_y"], 1e-4, "y")
    if dx is not None:
        close(dx, ref[name + "_dx"], 1e-4, "dx")
    close(dw, ref[name + "_dw"], 1e-4, "dw")
    close(db, ref[name + "_db"], 1e-4, "db")


@pytest.mark.parametrize("impl", ["valu", "mfma"])
@pytest.mark.parametrize("case", F3_CASES, ids=[c[0] for c in F3_CASES])
def test_conv_bf16_vs_oracle_on_rounded_operands(case, impl):
    """bf16 storage, fp32 accumulate: must equal the oracle applied to the SAME bf16-rounded
    operands up to one output rounding (2^-8) -- i.e. the kernel adds no error of its own."""
    name, cin, cout, k, relu, N, H, W = case
    x, w, b, gy = f3_tensors(case)
    xq = x if cin == 1 else bf16_round(x)         # images stay fp32 (image-side kernels)
    wq = w if (cin == 1 or cout == 1) else bf16_round(w)
    y_ref = O.conv2d_reflect_fwd(xq, wq, b, relu)
    with dtype_ctx("bf16", impl):
        y, dx, dw, db = _run_layer(case)
    close(y, y_ref if cout == 1 else bf16_round(y_ref), 6e-3, "y")
    # backward reference: upstream gradient rounded to bf16 where it enters a bf16 tensor, ReLU mask
    # from the kernel's own (rounded) output
    gq = gy if cout == 1 else bf16_round(gy)
    y_mask = y if relu else y_ref
    gx_ref, gw_ref, gb_ref = O.conv2d_reflect_bwd(xq, wq, y_mask, gq, relu, need_gx=cin > 1)
    close(dw, gw_ref, 2e-3, "dw")
    close(db, gb_ref, 2e-3, "db")
    if dx is not None:
        close(dx, gx_ref, 1.2e-2, "dx")   # dgrad output is stored in bf16 (padded domain) and folded in fp32


DMA_SHAPES = [(128, 128, 2, 37, 53), (176, 112, 1, 64, 48), (72, 64, 2, 33, 40), (64, 184, 1, 40, 72), (96, 96, 3, 16, 16)]
# thin layers (<= 48 channels either side) take the asynchronous loader/consumer kernel once a persistent block owns at least
# two tiles: >= 512 tiles of 16x16
DMA_SHAPES += [(16, 16, 3, 180, 200), (48, 16, 2, 250, 270), (16, 48, 3, 178, 190), (32, 16, 2, 256, 256), (24, 40, 2, 257, 300),
               (48, 48, 2, 241, 275), (8, 16, 2, 256, 300)]
# NestFuse's decoder / encoder 3x3 layers (reference core/block.py:836-867: channel counts that are not multiples of 64 -- ragged last
# M-block in forward (Cout) and dgrad (Cin), ragged 64-channel groups in the weight gradient, which skips the staging of their padded planes)
DMA_SHAPES += [(304, 152, 1, 48, 40), (176, 88, 2, 33, 48), (272, 136, 1, 40, 56), (240, 120, 1, 64, 32), (384, 192, 1, 32, 48), (368, 184, 1, 36, 44),
               (112, 56, 2, 40, 40), (160, 80, 1, 64, 64), (64, 72, 1, 70, 50), (80, 200, 1, 34, 34)]
_rng = np.random.default_rng(20240)
DMA_SHAPES += [(int(_rng.integers(7, 40)) * 8, int(_rng.integers(7, 32)) * 8, int(_rng.integers(1, 4)), int(_rng.integers(8, 90)),
                int(_rng.integers(8, 90))) for _ in range(10)]   # seeded random shapes: ragged everything


@pytest.mark.parametrize("cin,cout,n,h,w", DMA_SHAPES, ids=[f"{a}-{b}-{n}x{h}x{w}" for a, b, n, h, w in DMA_SHAPES])
def test_dma_staged_kernels_equal_register_staged(cin, cout, n, h, w):
    """The two kernel generations (conv_dma / wgrad_dma vs conv_mfma / wgrad_mfma) on identical bf16 operands: ragged
    tiles, ragged K chunks (cin % 32 != 0), ragged 64-channel groups, partial mask / accumulate bit sets.  fwd and dgrad
    use the same MFMA order -> bit identical; wgrad sums tiles in a different order -> 2e-5."""
    from mmif import tensor as T
    from mmif._lib import IMPL_MFMA, lib
    dev = "cuda:0"
    torch.manual_seed(cin * 7 + cout)
    x = T.BT.alloc(n, cin, h, w, torch.bfloat16, dev); x.buf.normal_()
    gy = T.BT.alloc(n, cout, h, w, torch.bfloat16, dev, halo=1, zero=True); gy.buf[:, :, 1:-1, 1:-1].normal_()
    gy = gy.as_folded()
    wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    pk = T.PackedWeights(cout, cin, 3, dev); pk.pack(wt)
    ws = torch.empty(T.wgrad_workspace_bytes(cin, cout, 3) // 4 + 1, dtype=torch.float32, device=dev)
    mask = 0x5a5a5a5a5a5a & ((1 << x.cb) - 1)
    acc_bits = 0x333333333333 & ((1 << x.cb) - 1)
    res = {}
    try:
        for mode in (0, 1):
            lib.mmif_debug_set_conv_dma(mode)
            y = T.BT.alloc(n, cout, h, w, torch.bfloat16, dev)
            gx = T.BT.alloc(n, cin, h, w, torch.bfloat16, dev, halo=1, zero=True)
            gx.buf.fill_(0.25)
            dw, db = torch.zeros_like(wt), torch.zeros_like(b)
            T.conv_fwd(x, wt, b, y, cin, cout, 3, True, pk, IMPL_MFMA)
            T.conv_dgrad(gy, wt, x, gx, cin, cout, 3, mask, acc_bits, pk, IMPL_MFMA)
            T.conv_wgrad(x, gy, dw, db, cin, cout, 3, ws, False, IMPL_MFMA)
            torch.cuda.synchronize()
            res[mode] = (y.buf.clone(), gx.buf.clone(), dw, db)
    finally:
        lib.mmif_debug_set_conv_dma(1)
    assert torch.equal(res[0][0], res[1][0]), "fwd differs"
    assert torch.equal(res[0][1], res[1][1]), "dgrad differs"
    close(res[1][2].cpu().numpy(), res[0][2].cpu().numpy(), 2e-5, "dw")
    close(res[1][3].cpu().numpy(), res[0][3].cpu().numpy(), 2e-5, "db")


RAGGED_SHAPES = [(304, 152, 2, 96, 80), (176, 88, 1, 128, 128), (272, 136, 2, 64, 72), (384, 192, 1, 96, 96), (120, 72, 1, 130, 70), (64, 136, 2, 50, 66)]


@pytest.mark.parametrize("cin,cout,n,h,w", RAGGED_SHAPES, ids=[f"{a}-{b}-{n}x{h}x{w}" for a, b, n, h, w in RAGGED_SHAPES])
def test_ragged_channel_groups_on_equals_off(cin, cout, n, h, w):
    """Round 6: wgrad_dma_kernel skips the staging of channel-block planes past the tensor in a ragged last 64-channel group
    (mmif_debug_set_ragged): the consumers then multiply stale LDS contents for the padded fragments, which must not reach any real output --
    dW / db identical to the run that stages them, forward and input gradient untouched (with ReLU masks / accumulation in the ragged block as
    well) -- and the weight gradient against its fp64 definition on the bf16 operands."""
    from mmif import tensor as T
    from mmif._lib import IMPL_MFMA, lib
    dev = "cuda:0"
    torch.manual_seed(cin * 3 + cout)
    x = T.BT.alloc(n, cin, h, w, torch.bfloat16, dev); x.buf.normal_()
    gy = T.BT.alloc(n, cout, h, w, torch.bfloat16, dev, halo=1, zero=True); gy.buf[:, :, 1:-1, 1:-1].normal_()
    gy = gy.as_folded()
    wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    pk = T.PackedWeights(cout, cin, 3, dev); pk.pack(wt)
    ws = torch.empty(T.wgrad_workspace_bytes(cin, cout, 3) // 4 + 1, dtype=torch.float32, device=dev)
    mask = 0xa5a5a5a5a5a5a5 & ((1 << x.cb) - 1) | (1 << (x.cb - 1))      # (the ragged last block's channel blocks masked AND accumulated)
    acc_bits = 0xcccccccccccccc & ((1 << x.cb) - 1) | (1 << (x.cb - 1))
    res = {}
    try:
        for mode in (0, 1):
            lib.mmif_debug_set_ragged(mode)
            y = T.BT.alloc(n, cout, h, w, torch.bfloat16, dev)
            y.buf.fill_(7.0)          # (a fragment past the tensor must not be stored anywhere)
            gx = T.BT.alloc(n, cin, h, w, torch.bfloat16, dev, halo=1, zero=True)
            gx.buf[:, :, 1:-1, 1:-1].fill_(0.25)
            dw, db = torch.zeros_like(wt), torch.zeros_like(b)
            T.conv_fwd(x, wt, b, y, cin, cout, 3, True, pk, IMPL_MFMA)
            T.conv_dgrad(gy, wt, x, gx, cin, cout, 3, mask, acc_bits, pk, IMPL_MFMA, fold=True)
            T.conv_wgrad(x, gy, dw, db, cin, cout, 3, ws, False, IMPL_MFMA)
            torch.cuda.synchronize()
            res[mode] = (y.buf.clone(), gx.buf.clone(), dw, db)
    finally:
        lib.mmif_debug_set_ragged(1)
    assert torch.equal(res[0][0], res[1][0]), "fwd differs"
    assert torch.equal(res[0][1], res[1][1]), "dgrad differs"
    close(res[1][2].cpu().numpy(), res[0][2].cpu().numpy(), 2e-5, "dw")
    close(res[1][3].cpu().numpy(), res[0][3].cpu().numpy(), 2e-5, "db")
    # the weight gradient against its fp64 definition on the same bf16 operands (reflect-padded x, interior of gy)
    xs = x.to_nchw().double()
    gs = gy.to_nchw().double()
    xp = torch.nn.functional.pad(xs, (1, 1, 1, 1), mode="reflect")
    ref = torch.nn.grad.conv2d_weight(xp, wt.shape, gs)
    close(res[1][2].cpu().numpy(), ref.cpu().numpy(), 1e-4, "dw vs fp64")
    close(res[1][3].cpu().numpy(), gs.sum(dim=(0, 2, 3)).cpu().numpy(), 1e-4, "db vs fp64")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,c,h,w", [(2, 16, 9, 12), (1, 24, 4, 4), (1, 8, 3, 5), (2, 8, 2, 2)])
def test_fold_halo_is_reflect_pad_adjoint_and_zeroes_halo(dtype, n, c, h, w):
    """mmif_fold_halo on a padded-domain gradient: interior = adjoint of F.pad(mode='reflect', 1) applied to the stored
    [h+2][w+2] map, halo ring = 0 afterwards (the DMA-staged dgrad reads that ring as its zero fill)."""
    from mmif import tensor as T
    torch.manual_seed(h * 31 + w)
    g = T.BT.alloc(n, c, h, w, dtype, "cuda:0", halo=1)
    g.buf.normal_()
    full = g.buf.float().cpu().numpy().copy()            # [n][cb][h+2][w+2][8]
    want = full[:, :, 1:-1, 1:-1].copy()
    def R(t, L):
        t = -t if t < 0 else t
        return 2 * (L - 1) - t if t >= L else t
    for ys in range(h + 2):
        for xs in range(w + 2):
            if 1 <= ys <= h and 1 <= xs <= w:
                continue
            want[:, :, R(ys - 1, h), R(xs - 1, w)] += full[:, :, ys, xs]
    folded = g.fold_halo_()
    torch.cuda.synchronize()
    got = folded.buf.float().cpu().numpy()
    tol = 1e-6 if dtype == torch.float32 else 2e-2
    assert np.abs(got[:, :, 1:-1, 1:-1] - want).max() <= tol * max(1.0, np.abs(want).max())
    ring = got.copy()
    ring[:, :, 1:-1, 1:-1] = 0
    assert np.abs(ring).max() == 0.0, "halo ring must be zero after the fold"


@pytest.mark.gpu
def test_batched_weight_packing_equals_per_layer_packing():
    """mmif_pack_weights_multi (one launch for every operand image of a model) writes exactly what per-layer mmif_pack_weights does,
    including more images than one launch's table holds (64)."""
    from mmif import tensor as T
    dev = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(7)
    shapes = [(16, 16, 3), (16, 48, 3), (128, 128, 3), (64, 128, 3), (64, 88, 1), (8, 64, 1), (40, 24, 3)] * 5   # 70 images
    ws = [torch.randn(co, ci, k, k, generator=g).to(dev) for co, ci, k in shapes]
    single = [T.PackedWeights(co, ci, k, dev) for co, ci, k in shapes]
    multi = [T.PackedWeights(co, ci, k, dev) for co, ci, k in shapes]
    for pk in single + multi:   # the buffers hold max(fwd, dgrad) bytes: bytes past an image's end are not written by either
        pk.fwd.fill_(0xAB), pk.dgrad.fill_(0xCD)
    for pk, w in zip(single, ws):
        pk.pack(w)
    T.pack_many(list(zip(multi, ws)))
    torch.cuda.synchronize()
    for a, b in zip(single, multi):
        assert torch.equal(a.fwd, b.fwd) and torch.equal(a.dgrad, b.dgrad)


FOLD_SHAPES = DMA_SHAPES + [(128, 64, 1, 4, 4), (64, 64, 2, 5, 7), (16, 16, 40, 64, 64), (48, 16, 36, 66, 62), (128, 128, 1, 70, 35),
                            (64, 32, 2, 256, 256)]


@pytest.mark.parametrize("cin,cout,n,h,w", FOLD_SHAPES, ids=[f"{a}-{b}-{n}x{h}x{w}" for a, b, n, h, w in FOLD_SHAPES])
def test_dgrad_with_fused_fold_equals_dgrad_then_fold(cin, cout, n, h, w):
    """mmif_conv2d_reflect_dgrad_folded: the DMA-staged kernels tile the interior only and fold the reflect halo inside the
    border tiles (extra MFMA steps); reference = the register-staged kernel on the padded domain + mmif_fold_halo.  Pixels
    that are no fold target are bit-identical; the targets (logical rows 1 / h-2, cols 1 / w-2) differ only by the bf16
    rounding of the halo values the fused form never stores; the halo ring stays zero; partial mask / accumulate bits."""
    from mmif import tensor as T
    from mmif._lib import IMPL_MFMA, lib
    dev = "cuda:0"
    torch.manual_seed(cin * 11 + cout + h)
    x = T.BT.alloc(n, cin, h, w, torch.bfloat16, dev); x.buf.normal_()
    gy = T.BT.alloc(n, cout, h, w, torch.bfloat16, dev, halo=1, zero=True); gy.buf[:, :, 1:-1, 1:-1].normal_()
    gy = gy.as_folded()
    wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    pk = T.PackedWeights(cout, cin, 3, dev); pk.pack(wt)
    mask = 0x5a5a5a5a5a5a & ((1 << x.cb) - 1)
    acc_bits = 0x333333333333 & ((1 << x.cb) - 1)
    old = torch.randn(n, x.cb, h, w, 8, device=dev).to(torch.bfloat16)
    res = {}
    try:
        for mode in (0, 1):
            lib.mmif_debug_set_conv_dma(mode)   # 0: register-staged dgrad + fold kernel; 1: fused where the DMA kernels apply
            gx = T.BT.alloc(n, cin, h, w, torch.bfloat16, dev, halo=1, zero=True)
            gx.buf[:, :, 1:-1, 1:-1] = old      # accumulate target: folded gradient, zero ring
            out = T.conv_dgrad(gy, wt, x, gx, cin, cout, 3, mask, acc_bits, pk, IMPL_MFMA, fold=True)
            torch.cuda.synchronize()
            assert out.flags & 1
            res[mode] = gx.buf.float().clone()
    finally:
        lib.mmif_debug_set_conv_dma(1)
    ref, got = res[0], res[1]
    for r in (ref, got):   # ring is zero in both forms
        assert float(r[:, :, 0].abs().max()) == 0 and float(r[:, :, -1].abs().max()) == 0
        assert float(r[:, :, :, 0].abs().max()) == 0 and float(r[:, :, :, -1].abs().max()) == 0
    tgt = torch.zeros(h + 2, w + 2, dtype=torch.bool, device=dev)
    tgt[[2, h - 1], :] = True
    tgt[:, [2, w - 1]] = True
    assert torch.equal(ref[:, :, ~tgt], got[:, :, ~tgt]), "non-target pixels differ"
    a, b = ref[:, :, tgt], got[:, :, tgt]
    err = (a - b).abs()
    tol = 2.0 ** -6 * torch.maximum(a.abs(), b.abs()) + 3e-2   # two bf16 roundings of O(1) halo values
    assert bool((err <= tol).all()), f"fold targets differ by up to {float(err.max())}"


@pytest.mark.parametrize("cin,cout,n,h,w", [(64, 32, 2, 256, 256), (64, 32, 9, 128, 128), (56, 32, 3, 250, 180), (64, 24, 1, 512, 384), (64, 32, 5, 100, 333)],
                         ids=lambda v: str(v))
def test_thin_wide_forward_equals_register_staged_kernel(cin, cout, n, h, w):
    """round 4: the 64 -> 32 forward (decode.2 of the PFNet / DenseFuse decoders, core/model.py:84) on thin_conv_async_kernel's two-group /
    three-slot geometry against the register-staged conv_mfma_kernel<3, 2> it replaces: bit for bit (same operand image, k-group order,
    bias / ReLU / rounding points), ragged tiles on both axes, 56 input channels (7 blocks: a ragged last chunk), output slot inside a wider
    buffer with untouched neighbours -- and against the CPU oracle on the rounded operands."""
    from mmif import tensor as T
    from mmif._lib import IMPL_MFMA, lib
    from gpu_util import bf16_round
    torch.manual_seed(cin + h)
    xn = torch.relu(torch.randn(n, cin, h, w))
    wt = (torch.randn(cout, cin, 3, 3) * 0.05).to(DEV)
    b = torch.randn(cout).to(DEV)
    pk = T.PackedWeights(cout, cin, 3, DEV)
    pk.pack(wt)
    xb = T.BT.from_nchw(xn.to(DEV), torch.bfloat16)
    res = {}
    try:
        for mode in (0, 1):
            lib.mmif_debug_set_thin_wide(mode)
            big = T.BT.alloc(n, 32 + 16, h, w, torch.bfloat16, DEV)
            big.buf.fill_(3.0)
            y = big.view(1, (cout + 7) // 8)
            T.conv_fwd(xb, wt, b, y, cin, cout, 3, True, pk, IMPL_MFMA)
            torch.cuda.synchronize()
            res[mode] = (big.buf.view(torch.int16).clone(), y.to_nchw(cout).cpu().numpy())
    finally:
        lib.mmif_debug_set_thin_wide(1)
    assert torch.equal(res[0][0], res[1][0])
    bb = res[1][0].view(torch.bfloat16).float()
    assert float((bb[:, 0] - 3).abs().max()) == 0 and float((bb[:, 1 + (cout + 7) // 8:] - 3).abs().max()) == 0, "neighbours of the output slot"
    if n * h * w <= 2 * 256 * 256:
        want = O.conv2d_reflect_fwd(bf16_round(xn.numpy()), bf16_round(wt.cpu().numpy()), b.cpu().numpy(), True)
        close(res[1][1], bf16_round(want), 6e-3, "y vs oracle")


@pytest.mark.parametrize("n,h,w", [(1, 40, 56), (2, 37, 53), (1, 4, 4), (2, 128, 96), (1, 33, 250)], ids=lambda v: str(v))
def test_dgrad_with_masked_copies_equals_dgrad_plus_fuse_elem_bwd(n, h, w):
    """Round 6 (DenseFuse, reference core/model.py:165-186): mmif_conv2d_reflect_dgrad_folded_dup -- decode.0's input gradient (64 -> 64, nothing
    masked on its own output) also leaves blocks 6, 7 masked by x3 of encoder branch a (blocks 6, 7 of F) and by x3 of branch b (blocks 14, 15)
    in GF.  Against mmif_conv2d_reflect_dgrad_folded + mmif_fuse_elem_bwd(sum, relu mask): bit for bit; the other blocks of GF and its ring
    untouched."""
    from mmif import tensor as T
    from mmif._lib import FUSE_SUM, IMPL_MFMA
    dev = "cuda:0"
    torch.manual_seed(h * 3 + w)
    cin = cout = 64
    gy = T.BT.alloc(n, cout, h, w, torch.bfloat16, dev, halo=1, zero=True); gy.buf[:, :, 1:-1, 1:-1].normal_()
    gy = gy.as_folded()
    F = T.BT.alloc(n, 128, h, w, torch.bfloat16, dev); F.buf.normal_()       # [x(img1) | x(img2)]: about half of the values > 0
    wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    pk = T.PackedWeights(cout, cin, 3, dev); pk.pack(wt)
    gx_a = T.BT.alloc(n, cin, h, w, torch.bfloat16, dev, halo=1, zero=True)
    gx_b = T.BT.alloc(n, cin, h, w, torch.bfloat16, dev, halo=1, zero=True)
    GF_a = T.BT.alloc(n, 128, h, w, torch.bfloat16, dev, halo=1, zero=True)
    GF_b = T.BT.alloc(n, 128, h, w, torch.bfloat16, dev, halo=1, zero=True)
    assert T.conv_dgrad_dup_supported(gy, gx_b, cin, cout, 3) == (h >= 4 and w >= 4)
    ga = T.conv_dgrad(gy, wt, None, gx_a, cin, cout, 3, 0, 0, pk, IMPL_MFMA, fold=True)
    T.fuse_elem_bwd(F.view(6, 2), F.view(14, 2), ga.view(6, 2), GF_a.view(6, 2), GF_a.view(14, 2), FUSE_SUM, True)
    gb = T.conv_dgrad_dup(gy, gx_b, cin, cout, 3, pk, GF_b, F, 3)
    torch.cuda.synchronize()
    assert torch.equal(gx_a.buf, gx_b.buf), "the dgrad's own output differs"
    assert torch.equal(GF_a.buf, GF_b.buf), "the masked copies differ"
    assert float(GF_b.buf[:, 6:8].float().abs().max()) > 0 and float(GF_b.buf[:, 14:16].float().abs().max()) > 0
    assert float(GF_b.buf[:, :6].float().abs().max()) == 0 and float(GF_b.buf[:, 8:14].float().abs().max()) == 0
    assert gb.flags == ga.flags
