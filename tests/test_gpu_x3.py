"""The split-bf16 ("x3") kernels that run the 3x3 ConvLayers of FP32 tensors on the matrix pipe (csrc/conv_x3.hip): operand images,
forward, input gradient (mask / accumulate bits, halo-0 and folded halo-1 upstream gradients) and weight gradient against the fp32 FMA
kernels on identical operands and against the fp64 definition.  The reference computes these layers in fp32 (core/block.py:56-66,
98-99); the north-star bar is 1e-3 relative, the split products are held to 3e-5 here (2^-16 per product, fp32 accumulate)."""
import numpy as np
import pytest
import torch

from gpu_util import close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

X3_TOL = 3e-5


def _split(w, pieces):
    """torch model of the split: piece p = bf16 (RNE) of what the earlier pieces left; pieces == 16: two fp16 pieces of 2^10 w"""
    t = torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32))
    if pieces == 16:
        t = (t * 1024.0).clamp(-65000.0, 65000.0)
        hi = t.half()
        return [hi.view(torch.bfloat16), (t - hi.float()).half().view(torch.bfloat16)]   # (bit patterns, compared as int16 below)
    out = []
    for _ in range(pieces):
        q = t.bfloat16()
        out.append(q)
        t = t - q.float()
    return out


@pytest.fixture(params=[16, 3, 2], ids=["fwd-f16x2", "fwd6", "fwd3"])
def fwd_pieces(request):
    from mmif import engine as E
    from mmif._lib import lib
    prev = lib.mmif_get_x3_forward_pieces()
    E.set_x3_forward_pieces(request.param)
    yield request.param
    E.set_x3_forward_pieces(prev)


@pytest.mark.parametrize("ks", [3, 1])
@pytest.mark.parametrize("cout,cin", [(128, 128), (16, 48), (64, 128), (40, 24), (72, 136)])
def test_x3_operand_images_bit_exact(cout, cin, ks, fwd_pieces):
    """mmif_pack_weights_x3: [m-block][chunk][piece][tap][2 channel blocks][32*MB out][8 in] -- every element against the layout
    formula, forward (2 or 3 pieces) and (flipped, transposed, 2 pieces) dgrad images"""
    from mmif import tensor as T
    from mmif._lib import F32
    torch.manual_seed(cout * 131 + cin)
    w = (torch.randn(cout, cin, ks, ks) * 0.1)
    pk = T.PackedWeights(cout, cin, ks, DEV, F32)
    pk.pack(w.to(DEV))
    torch.cuda.synchronize()
    wn = w.numpy()
    for dgrad, img in ((0, pk.fwd), (1, pk.dgrad)):
        pieces = 2 if dgrad else fwd_pieces
        n_out, n_in = (cin, cout) if dgrad else (cout, cin)
        mb = 2 if n_out > 32 else 1
        mbw = 32 * mb
        nmb, nch = -(-n_out // mbw), -(-(-(-n_in // 8)) // 2)
        wk = np.zeros((nmb * mbw, nch * 16, ks, ks), np.float32)   # [out][in][u][v] in the kernel's view
        if dgrad:
            wk[:n_out, :n_in] = wn.transpose(1, 0, 2, 3)[:, :, ::-1, ::-1]
        else:
            wk[:n_out, :n_in] = wn
        want = wk.reshape(nmb, mbw, nch, 2, 8, ks * ks).transpose(0, 2, 5, 3, 1, 4)   # [mb][ch][tap][cbl][ocl][e]
        both = torch.stack(_split(want, pieces), dim=2).contiguous()            # [mb][ch][piece][tap][cbl][ocl][e]
        got = img.cpu().view(torch.bfloat16)[:both.numel()].view(both.shape)
        assert torch.equal(got.view(torch.int16), both.view(torch.int16)), f"dgrad={dgrad}"


SHAPES = [(128, 128, 2, 37, 53), (16, 16, 2, 40, 70), (48, 16, 1, 33, 64), (16, 48, 2, 31, 45), (64, 32, 1, 64, 48), (32, 16, 2, 16, 32),
          (128, 64, 1, 19, 33), (24, 40, 2, 9, 100), (136, 88, 1, 20, 36), (8, 8, 1, 2, 2), (72, 152, 1, 17, 34), (64, 64, 3, 8, 16)]


SHAPES_1x1 = [(88, 64, 2, 37, 53), (152, 304, 1, 20, 36), (32, 64, 2, 16, 32), (8, 16, 1, 2, 2), (136, 72, 1, 9, 100), (64, 32, 1, 40, 24)]


@pytest.mark.parametrize("cin,cout,n,h,w,ks", [s + (3,) for s in SHAPES] + [s + (1,) for s in SHAPES_1x1],
                         ids=[f"{a}-{b}-{n}x{h}x{w}" for a, b, n, h, w in SHAPES] + [f"1x1-{a}-{b}-{n}x{h}x{w}" for a, b, n, h, w in SHAPES_1x1])
@pytest.mark.parametrize("ghalo", [0, 1])
def test_x3_kernels_vs_fp32_fma_kernels(cin, cout, n, h, w, ks, ghalo, fwd_pieces):
    """forward (bias + ReLU), dgrad (partial mask / accumulate bit sets; upstream gradient and old values vary per element) and wgrad:
    IMPL_X3 vs IMPL_VALU on the same fp32 tensors -- ragged tiles, ragged 16-channel chunks (cin % 16 = 8), ragged 32 / 64-channel
    groups.  The 3-piece forward (six products) must agree to fp32 rounding (2e-6), everything with 2 pieces to 3e-5."""
    from mmif import tensor as T
    from mmif._lib import F32, IMPL_VALU, IMPL_X3
    torch.manual_seed(cin * 7 + cout + h)
    x = T.BT.alloc(n, cin, h, w, torch.float32, DEV); x.buf.normal_()
    gy = T.BT.alloc(n, cout, h, w, torch.float32, DEV, halo=ghalo, zero=True)
    if ghalo:
        gy.buf[:, :, 1:-1, 1:-1].normal_()
        gy = gy.as_folded()
    else:
        gy.buf.normal_()
    wt = torch.randn(cout, cin, ks, ks, device=DEV) * 0.05
    b = torch.randn(cout, device=DEV)
    pk = T.PackedWeights(cout, cin, ks, DEV, F32); pk.pack(wt)
    ws = torch.empty(T.wgrad_workspace_bytes(cin, cout, ks) // 4 + 1, dtype=torch.float32, device=DEV)
    mask = 0x5a5a5a5a5a5a & ((1 << x.cb) - 1)
    acc_bits = 0x333333333333 & ((1 << x.cb) - 1)
    res = {}
    for impl in (IMPL_VALU, IMPL_X3):
        y = T.BT.alloc(n, cout, h, w, torch.float32, DEV)
        gx = T.BT.alloc(n, cin, h, w, torch.float32, DEV, halo=1, zero=True)
        gx.buf.copy_(torch.sin(torch.arange(gx.buf.numel(), device=DEV, dtype=torch.float32)).view_as(gx.buf))   # old values differ per element
        dw, db = torch.full_like(wt, 0.5), torch.full_like(b, -0.5)
        T.conv_fwd(x, wt, b, y, cin, cout, ks, True, pk, impl)
        T.conv_dgrad(gy, wt, x, gx, cin, cout, ks, mask, acc_bits, pk, impl)
        T.conv_wgrad(x, gy, dw, db, cin, cout, ks, ws, True, impl)
        torch.cuda.synchronize()
        res[impl] = [t.cpu().numpy() for t in (y.buf, gx.buf, dw, db)]
    for (a, r, what) in zip(res[IMPL_X3], res[IMPL_VALU], ("y", "gx", "dw", "db")):
        close(a, r, 2e-6 if (what == "y" and fwd_pieces in (3, 16)) else X3_TOL, what)


@pytest.mark.parametrize("cin,cout,n,h,w,ks", [s + (3,) for s in SHAPES] + [s + (1,) for s in SHAPES_1x1[:3]],
                         ids=[f"{a}-{b}-{n}x{h}x{w}" for a, b, n, h, w in SHAPES] + [f"1x1-{a}-{b}-{n}x{h}x{w}" for a, b, n, h, w in SHAPES_1x1[:3]])
def test_bwd_wide_fp32_equals_wgrad_plus_folded_dgrad(cin, cout, n, h, w, ks):
    """mmif_conv2d_reflect_bwd_wide on fp32 tensors (the split-operand weight gradient leaves the ReLU sign map of x, the split-operand
    dgrad masks with it) == mmif_conv2d_reflect_wgrad followed by mmif_conv2d_reflect_dgrad_folded, BIT FOR BIT: same kernels, and the
    map holds exactly [x > 0] -- zeros and negative values of x included, partial mask-bit sets, ragged tiles and channel groups."""
    from mmif import tensor as T
    from mmif._lib import F32, IMPL_X3
    torch.manual_seed(cin + 3 * cout + w)
    x = T.BT.alloc(n, cin, h, w, torch.float32, DEV); x.buf.normal_()
    x.buf[x.buf.abs() < 0.3] = 0.0                     # plenty of exact zeros (post-ReLU activations have them)
    gy = T.BT.alloc(n, cout, h, w, torch.float32, DEV, halo=1, zero=True)
    gy.buf[:, :, 1:-1, 1:-1].normal_()
    gy = gy.as_folded()
    wt = torch.randn(cout, cin, ks, ks, device=DEV) * 0.05
    pk = T.PackedWeights(cout, cin, ks, DEV, F32); pk.pack(wt)
    ws = torch.empty(T.wgrad_workspace_bytes(cin, cout, ks) // 4 + 1, dtype=torch.float32, device=DEV)
    assert T.bwd_wide_supported(cin, cout, ks, torch.float32)
    for mask in ((1 << x.cb) - 1, 0x5a5a5a5a5a5a & ((1 << x.cb) - 1), 0):
        gx_a = T.BT.alloc(n, cin, h, w, torch.float32, DEV, halo=1, zero=True)
        gx_b = T.BT.alloc(n, cin, h, w, torch.float32, DEV, halo=1, zero=True)
        dw_a, db_a = torch.zeros_like(wt), torch.zeros(cout, device=DEV)
        dw_b, db_b = torch.zeros_like(wt), torch.zeros(cout, device=DEV)
        T.conv_wgrad(x, gy, dw_a, db_a, cin, cout, ks, ws, False, IMPL_X3)
        T.conv_dgrad(gy, wt, x, gx_a, cin, cout, ks, mask, 0, pk, IMPL_X3, fold=True)
        signs = torch.full((T.bwd_wide_signs_bytes(n, cin, h, w),), 0xa5, dtype=torch.uint8, device=DEV)   # stale scratch
        T.conv_bwd_wide(gy, x, gx_b, dw_b, db_b, cin, cout, ks, pk, mask, ws, signs)
        torch.cuda.synchronize()
        assert torch.equal(dw_a, dw_b) and torch.equal(db_a, db_b), f"mask {mask:#x}: weight gradient"
        assert torch.equal(gx_a.buf, gx_b.buf), f"mask {mask:#x}: input gradient"


@pytest.mark.parametrize("cin,cout", [(72, 64), (68, 32), (136, 64)])
def test_bwd_wide_fp32_exact_sign_buffer(cin, cout):
    """the fp32 bwd_wide dgrad read the sign map one 32-channel group too far for cin % 64 in (0, 32] (72, 136, ...; 68 also
    has cin % 8 != 0), past a buffer sized exactly like the x3 layout ([n][ceil(cb / 4)][h][w] dwords).  The group index is clamped now:
    the call accepts a buffer of exactly that size (the last bytes of an allocation), gives the bits of the two separate calls, and
    leaves the bytes behind the map untouched."""
    from mmif import tensor as T
    from mmif._lib import F32, IMPL_X3
    n, h, w, ks = 2, 20, 36, 3
    torch.manual_seed(cin)
    x = T.BT.alloc(n, cin, h, w, torch.float32, DEV); x.buf.normal_()
    x.buf[x.buf.abs() < 0.3] = 0.0
    if cin % 8:
        x.buf.view(n, -1, h, w, 8)[:, -1, :, :, cin % 8:] = 0.0
    gy = T.BT.alloc(n, cout, h, w, torch.float32, DEV, halo=1, zero=True)
    gy.buf[:, :, 1:-1, 1:-1].normal_()
    gy = gy.as_folded()
    wt = torch.randn(cout, cin, ks, ks, device=DEV) * 0.05
    pk = T.PackedWeights(cout, cin, ks, DEV, F32); pk.pack(wt)
    ws = torch.empty(T.wgrad_workspace_bytes(cin, cout, ks) // 4 + 1, dtype=torch.float32, device=DEV)
    if not T.bwd_wide_supported(cin, cout, ks, torch.float32):
        pytest.skip("shape not on the sign-map path")
    exact = n * ((x.cb + 3) // 4) * h * w * 4
    guard = 4096
    store = torch.full((exact + guard,), 0xa5, dtype=torch.uint8, device=DEV)
    mask = (1 << x.cb) - 1
    gx_a = T.BT.alloc(n, cin, h, w, torch.float32, DEV, halo=1, zero=True)
    gx_b = T.BT.alloc(n, cin, h, w, torch.float32, DEV, halo=1, zero=True)
    dw_a, db_a = torch.zeros_like(wt), torch.zeros(cout, device=DEV)
    dw_b, db_b = torch.zeros_like(wt), torch.zeros(cout, device=DEV)
    T.conv_wgrad(x, gy, dw_a, db_a, cin, cout, ks, ws, False, IMPL_X3)
    T.conv_dgrad(gy, wt, x, gx_a, cin, cout, ks, mask, 0, pk, IMPL_X3, fold=True)
    T.conv_bwd_wide(gy, x, gx_b, dw_b, db_b, cin, cout, ks, pk, mask, ws, store[:exact])
    torch.cuda.synchronize()
    assert torch.equal(dw_a, dw_b) and torch.equal(db_a, db_b)
    assert torch.equal(gx_a.buf, gx_b.buf) and float(gx_a.buf.abs().max()) > 0
    assert bool((store[exact:] == 0xa5).all()), "bytes behind the sign map were written"
    with pytest.raises(RuntimeError):      # one byte short is refused
        T.conv_bwd_wide(gy, x, gx_b, dw_b, db_b, cin, cout, ks, pk, mask, ws, store[:exact - 1])


def test_x3_forward_vs_fp64_definition(fwd_pieces):
    """x3 forward on a 128 -> 128 layer against torch's fp64 conv on the CPU (reflect padding): the error of the split (2 pieces: 2^-17
    relative per product; 3 pieces: fp32 accumulation only) next to the fp32 FMA kernel's own rounding"""
    import torch.nn.functional as F
    from mmif import tensor as T
    from mmif._lib import F32, IMPL_VALU, IMPL_X3
    torch.manual_seed(5)
    n, c, h, w = 1, 128, 24, 40
    xn = torch.randn(n, c, h, w)
    wt = torch.randn(c, c, 3, 3) * 0.03
    b = torch.randn(c)
    ref = F.conv2d(F.pad(xn.double(), (1, 1, 1, 1), mode="reflect"), wt.double(), b.double()).clamp_min(0).numpy()
    x = T.BT.from_nchw(xn.to(DEV), torch.float32)
    pk = T.PackedWeights(c, c, 3, DEV, F32); pk.pack(wt.to(DEV))
    errs = {}
    for impl in (IMPL_VALU, IMPL_X3):
        y = T.BT.alloc(n, c, h, w, torch.float32, DEV)
        T.conv_fwd(x, wt.to(DEV), b.to(DEV), y, c, c, 3, True, pk, impl)
        errs[impl] = np.abs(y.to_nchw(c).cpu().numpy() - ref).max() / np.abs(ref).max()
    print("fp32 FMA err", errs[IMPL_VALU], "x3 err", errs[IMPL_X3])
    assert errs[IMPL_X3] < (2e-5 if fwd_pieces == 2 else 3 * max(errs[IMPL_VALU], 2e-7)), errs
    print(fwd_pieces, errs)


def test_x3_is_the_default_for_fp32_and_can_be_forced_off():
    """AUTO on fp32 tensors takes the x3 kernels when the layer's x3 image is passed (bit-identical to IMPL_X3), the fp32 FMA kernels
    without it (bit-identical to IMPL_VALU); asking for IMPL_X3 on bf16 tensors or without the image is an error, not a fallback"""
    from mmif import tensor as T
    from mmif._lib import BF16, F32, IMPL_AUTO, IMPL_VALU, IMPL_X3, MmifError
    torch.manual_seed(9)
    n, cin, cout, h, w = 1, 32, 32, 20, 33
    x = T.BT.alloc(n, cin, h, w, torch.float32, DEV); x.buf.normal_()
    wt = torch.randn(cout, cin, 3, 3, device=DEV) * 0.05
    b = torch.randn(cout, device=DEV)
    pk = T.PackedWeights(cout, cin, 3, DEV, F32); pk.pack(wt)
    out = {}
    for key, impl, p in (("auto", IMPL_AUTO, pk), ("x3", IMPL_X3, pk), ("auto_nopk", IMPL_AUTO, None), ("valu", IMPL_VALU, pk)):
        y = T.BT.alloc(n, cout, h, w, torch.float32, DEV)
        T.conv_fwd(x, wt, b, y, cin, cout, 3, True, p, impl)
        out[key] = y.buf.clone()
    assert torch.equal(out["auto"], out["x3"]) and torch.equal(out["auto_nopk"], out["valu"])
    assert not torch.equal(out["x3"], out["valu"])
    y = T.BT.alloc(n, cout, h, w, torch.float32, DEV)
    with pytest.raises(MmifError):
        T.conv_fwd(x, wt, b, y, cin, cout, 3, True, None, IMPL_X3)
    pkb = T.PackedWeights(cout, cin, 3, DEV, BF16); pkb.pack(wt)
    with pytest.raises(MmifError):    # a bf16-format image is not an x3 image: never reinterpreted
        T.conv_fwd(x, wt, b, y, cin, cout, 3, True, pkb, IMPL_X3)


def test_forward_relu_decisions_agree_with_fp32_kernels():
    """Why the forward pass uses three pieces (six products): the ReLU decision [pre-activation > 0] of decode.0 (128 -> 128) at
    2 x 256 x 256 on the x3 kernels against the fp32 FMA kernels, on post-ReLU-like inputs.  With six products the two agree as two fp32
    implementations do (a handful of 16.8 M decisions, all on pre-activations at fp32 rounding noise); with three products the count is
    an order of magnitude larger -- each such flip moves the parameter gradients of every layer below by O(1 / sqrt(pixels))."""
    from mmif import engine as E
    from mmif import tensor as T
    from mmif._lib import F32, IMPL_VALU, IMPL_X3, lib
    torch.manual_seed(77)
    n, c, S = 2, 128, 256
    x = T.BT.alloc(n, c, S, S, torch.float32, DEV)
    x.buf.normal_().clamp_(min=0)                     # what a ReLU layer hands on
    wt = torch.randn(c, c, 3, 3, device=DEV) * (2.0 / (c * 9)) ** 0.5
    b = torch.randn(c, device=DEV) * 0.05
    y0 = T.BT.alloc(n, c, S, S, torch.float32, DEV)
    T.conv_fwd(x, wt, b, y0, c, c, 3, True, None, IMPL_VALU)
    ref = y0.buf > 0
    prev = lib.mmif_get_x3_forward_pieces()
    flips = {}
    try:
        for pieces in (16, 3, 2):
            E.set_x3_forward_pieces(pieces)
            pk = T.PackedWeights(c, c, 3, DEV, F32); pk.pack(wt)
            y = T.BT.alloc(n, c, S, S, torch.float32, DEV)
            T.conv_fwd(x, wt, b, y, c, c, 3, True, pk, IMPL_X3)
            torch.cuda.synchronize()
            diff = (y.buf > 0) != ref
            flips[pieces] = int(diff.sum())
            # a flipped element is one whose value is at the kernels' own error level on BOTH sides
            worst = max(float((y.buf.abs() * diff).max()), float((y0.buf.abs() * diff).max()))
            scale = float(y0.buf.max())
            assert worst <= (5e-5 if pieces == 2 else 1e-6) * scale, (pieces, worst, scale)
    finally:
        E.set_x3_forward_pieces(prev)
    total = ref.numel()
    print(f"ReLU decisions that differ from the fp32 FMA kernels, of {total}: scaled fp16 pieces (3 products) {flips[16]}, "
          f"three bf16 pieces (6 products) {flips[3]}, two bf16 pieces (3 products) {flips[2]}")
    assert flips[3] <= 32 and flips[16] <= 32, flips                    # fp32 summation-order noise
    assert max(flips[3], flips[16]) * 4 <= max(flips[2], 8), flips      # ... and several times rarer than with two bf16 pieces


def test_fp16_forward_range_window():
    """The fp16 forward (mode 16) scales every staged tile by the power of two that puts its maximum below 2^15, so the result is exact
    to 2^-23 of the largest operand of the sum at ANY magnitude -- whole tensors at 1e-30 or 1e20, a few channels 1e8 larger than the
    rest (the accumulator rescale between chunks), a bright region next to a dark one -- and weights saturate at |w| >= 64."""
    import torch.nn.functional as F
    from mmif import engine as E
    from mmif import tensor as T
    from mmif._lib import F32, IMPL_X3, lib
    torch.manual_seed(3)
    n, c, h, w = 1, 96, 48, 72                          # 96 channels = 6 chunks of 16: several LDS chunks per item
    wt = torch.randn(c, c, 3, 3) * 0.05
    b = torch.zeros(c)
    prev = lib.mmif_get_x3_forward_pieces()

    def run(xn, wt_, mode):
        E.set_x3_forward_pieces(mode)
        ref = F.conv2d(F.pad(xn.double(), (1, 1, 1, 1), mode="reflect"), wt_.double(), b.double()).clamp_min(0).numpy()
        x = T.BT.from_nchw(xn.to(DEV), torch.float32)
        pk = T.PackedWeights(c, c, 3, DEV, F32); pk.pack(wt_.to(DEV))
        y = T.BT.alloc(n, c, h, w, torch.float32, DEV)
        T.conv_fwd(x, wt_.to(DEV), b.to(DEV), y, c, c, 3, True, pk, IMPL_X3)
        return y.to_nchw(c).cpu().numpy(), ref

    try:
        for scale in (1e-30, 1e-6, 1e-4, 1.0, 3000.0, 1e5, 1e20):
            for mode in (16, 3):
                got, ref = run(torch.randn(n, c, h, w) * scale, wt, mode)
                close(got, ref, 2e-6, f"scale {scale} mode {mode}")
        # channel groups of very different magnitude, in both orders (big first: later chunks keep the exponent; big last: accumulators
        # are rescaled down), and a bright rectangle in a dark image
        for order in (0, 1):
            xn = torch.randn(n, c, h, w) * 1e-3
            sl = slice(0, 16) if order == 0 else slice(80, 96)
            xn[:, sl] *= 1e8
            got, ref = run(xn, wt, 16)
            close(got, ref, 2e-6, f"mixed channel magnitudes, order {order}")
        xn = torch.randn(n, c, h, w) * 1e-3
        xn[:, :, 4:9, 30:50] *= 1e6
        got, ref = run(xn, wt, 16)
        close(got, ref, 2e-6, "bright region")
        dark = (np.abs(ref) < 1.0) & (ref > 0)           # tiles away from the bright region keep THEIR OWN precision
        assert dark.sum() > 1000
        # (tiles that see the bright region carry its absolute error; the rest are exact to 2^-23 of the dark magnitude)
        far = np.zeros_like(dark); far[:, :, 32:, :] = True   # tiles are at most 32 rows: none of these staged a bright pixel
        e_far = np.abs(got - ref)[dark & far].max() / np.abs(ref)[dark & far].max()
        assert e_far <= 2e-6, e_far
        # weights beyond the window saturate: finite, bounded by the exact result
        got, ref = run(torch.randn(n, c, h, w), wt * 1e4, 16)
        assert np.isfinite(got).all() and np.abs(got).max() <= np.abs(ref).max()
    finally:
        E.set_x3_forward_pieces(prev)


def test_fp32_encoder_backward_gather_chain_and_fused_wgrad_vs_layerwise():
    """The fp32 encoder backward in its fused forms -- the dgrad chain per DESTINATION on the stacked virtual layers
    (mmif_pack_dense_chain_x3 + the 16-wide thin dgrad kernel) and the one-pass weight gradient (wgrad_x3_dense_kernel) -- against the
    layer-wise scatter form ($MMIF_ENC_CHAIN=0 $MMIF_ENC_WGRAD=0): every parameter gradient of PFNetv1 to the rounding of a different
    summation order (the gather form adds all contributions of an x_k in one fp32 accumulator before the mask)."""
    import os
    import core.model as M
    from core.loss import FusionLoss, GradLoss, PixelLoss, SSIMLoss
    from gpu_util import dtype_ctx, load_closed_form, reload_switches
    from oracle import fusion_oracle as O
    shape = (2, 1, 72, 88)
    i1 = torch.from_numpy(O.closed_form_image(shape, 0.3)).to(DEV)
    i2 = torch.from_numpy(O.closed_form_image(shape, 1.7)).to(DEV)
    res = {}
    prev = {k: os.environ.get(k) for k in ("MMIF_ENC_CHAIN", "MMIF_ENC_WGRAD")}
    try:
        for mode in ("0", "1"):
            os.environ["MMIF_ENC_CHAIN"] = os.environ["MMIF_ENC_WGRAD"] = mode
            reload_switches()
            with dtype_ctx("fp32"):
                m = load_closed_form(M.PFNetv1(), 1).to(DEV)
                m.train()
                fl = FusionLoss(SSIMLoss('ssim', weight=1.0), PixelLoss('l1', weight=0.01), GradLoss('l1', weight=0.1).to(DEV), 'max', 'max')
                fl(i1, i2, m(i1, i2)).backward()
                torch.cuda.synchronize()
                res[mode] = {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters()}
    finally:
        for k, v in prev.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        reload_switches()
    for k, g in res["0"].items():
        close(res["1"][k], g, 2e-5, k)


def test_out_of_range_weights_switch_the_forward_to_bf16_pieces():
    """the scaled-fp16 forward images clamp |w| >= ~63.5 -- that used to be silent.  The pack kernel now counts clamped values
    (mmif_x3_pack_saturations) and the engine, on its first pack, switches the process to three bf16 pieces per operand with a warning and
    re-packs before anything runs on the clamped images: the fused image then agrees with the fp32 FMA kernels as usual."""
    import warnings
    import core.model as M
    from mmif import engine as E
    from mmif._lib import lib
    from oracle import fusion_oracle as O
    from gpu_util import dtype_ctx, load_closed_form, tg
    shape = (1, 1, 24, 40)
    i1, i2 = tg(O.closed_form_image(shape, 0.3)), tg(O.closed_form_image(shape, 1.7))
    assert lib.mmif_get_x3_forward_pieces() == 16
    lib.mmif_x3_pack_saturations(1)
    res = {}
    try:
        for impl in ("valu", "auto"):
            with dtype_ctx("fp32", impl):
                m = load_closed_form(M.PFNetv1(), 1)
                with torch.no_grad():
                    m.decode[1].layers[0].weight[3, 5, 1, 1] = 100.0      # far outside the 2^10-scaled fp16 range
                    m.decode[1].layers[0].weight[7, 2, 0, 2] = -80.0
                m = m.to(DEV)
                with warnings.catch_warnings(record=True) as rec:
                    warnings.simplefilter("always")
                    with torch.no_grad():
                        res[impl] = m(i1, i2).cpu().numpy()
                if impl == "auto":
                    assert any("scaled-fp16" in str(w.message) for w in rec), [str(w.message) for w in rec]
                    assert lib.mmif_get_x3_forward_pieces() == 3
        close(res["auto"], res["valu"], 1e-5, "fused image with out-of-range weights")
        assert lib.mmif_x3_pack_saturations(0) == 0        # (three bf16 pieces never clamp)
    finally:
        E.set_x3_forward_pieces(16)
        lib.mmif_x3_pack_saturations(1)


def test_forward_dispatches_on_the_format_the_image_was_packed_in():
    """mmif_set_x3_forward_pieces used to change how EVERY later forward read its operand image, also images packed in
    the old format (fp16 pieces read as bf16 pieces = garbage).  The library now remembers each forward image's format at pack time and the
    launch follows the image: a mode change without a re-pack leaves the result bit-identical; after a re-pack the new format is used
    (a different kernel: equal within the formats' common 1e-6)."""
    from mmif import tensor as T
    from mmif._lib import F32, IMPL_X3, lib
    torch.manual_seed(11)
    cin, cout, n, h, w = 32, 64, 1, 20, 36
    x = T.BT.from_nchw(torch.randn(n, cin, h, w).to(DEV), torch.float32)
    wt = (torch.randn(cout, cin, 3, 3) * 0.1).to(DEV)
    b = torch.randn(cout).to(DEV)
    assert lib.mmif_get_x3_forward_pieces() == 16
    try:
        pk = T.PackedWeights(cout, cin, 3, DEV, fmt=F32)
        pk.pack(wt)
        outs = []
        for mode in (16, 3, 2, 16):
            lib.mmif_set_x3_forward_pieces(mode)                 # (the raw entry point: no epoch bump, no re-pack)
            y = T.BT.alloc(n, cout, h, w, torch.float32, DEV)
            T.conv_fwd(x, wt, b, y, cin, cout, 3, True, pk, IMPL_X3)
            torch.cuda.synchronize()
            outs.append(y.to_nchw(cout).clone())
        for o in outs[1:]:
            assert torch.equal(o, outs[0]), "a mode change without a re-pack must not change how the image is read"
        lib.mmif_set_x3_forward_pieces(3)
        pk.pack(wt)
        y = T.BT.alloc(n, cout, h, w, torch.float32, DEV)
        T.conv_fwd(x, wt, b, y, cin, cout, 3, True, pk, IMPL_X3)
        torch.cuda.synchronize()
        close(y.to_nchw(cout).cpu().numpy(), outs[0].cpu().numpy(), 2e-6, "three bf16 pieces vs scaled fp16 pieces")
    finally:
        lib.mmif_set_x3_forward_pieces(16)
