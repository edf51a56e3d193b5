"""BASELINE.json's full-size configurations, checked through size-independent properties (the CPU
oracle cannot run them in seconds):

  config 5  full-res 1224(W) x 1024(H) inference: a window of the full-res fused image must equal the
            fused image of the (receptive-field padded) crop -- i.e. tiling / block order / XCD mapping
            leave no trace -- and the crop itself is held to the oracle.
  config 2/3  B=32 256x256 train step: the gradient of the batch-mean loss equals the mean of the
            gradients of its sub-batches (linearity; "checksum of checksums" over the batch).
  config 4  NestFuse 512x512: samples of a batch do not interact (no cross-sample coupling on the path,
            SURVEY 8e) although channel attention pools over whole planes.
"""
import numpy as np
import pytest
import torch

from oracle import fusion_oracle as O
from gpu_util import close, dtype_ctx, load_closed_form, load_live, tg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

RF = 8  # PFNetv1: 4 encoder + 4 decoder 3x3 convs -> receptive-field radius 8


def _model(name, seed=1):
    """closed-form seed 1 for the nets without an output activation; NestFuse / RFN-Nest end in a ReLU that seed 1 kills on every
    pixel (the round-3 full-size tests compared zero images and zero gradients): they get the live set (oracle.LIVE_PARAMS: 53 % /
    46 % of the pixels of a 512 x 512 uniform-random pair pass the final ReLU, measured on the reference)"""
    import core.model as M
    m = getattr(M, name)()
    return (load_live(m, name) if name in O.LIVE_PARAMS else load_closed_form(m, seed)).to("cuda:0")


def _assert_alive_image(y, what):
    frac = float((y > 0).float().mean())
    assert float(y.abs().max()) > 0 and 0.2 <= frac <= 0.8, f"{what}: {frac:.3f} of the fused pixels are positive -- the test would be vacuous"


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_fullres_inference_window_equals_crop(dtype):
    H, W = 1024, 1224
    g = torch.Generator().manual_seed(5)
    i1, i2 = torch.rand(1, 1, H, W, generator=g).cuda(), torch.rand(1, 1, H, W, generator=g).cuda()
    with dtype_ctx(dtype), torch.no_grad():
        m = _model("PFNetv1")
        full = m(i1, i2)
        assert full.shape == (1, 1, H, W) and bool(torch.isfinite(full).all())
        # interior window at an unaligned offset, the last (ragged: 1224 = 76*16 + 8) columns, and the corners
        for (y0, x0, h, w) in [(301, 517, 64, 64), (H - 48, W - 56, 48, 56), (0, 0, 40, 40), (511, W - 40, 33, 40)]:
            ya, xa = max(y0 - 2 * RF, 0), max(x0 - 2 * RF, 0)
            yb, xb = min(y0 + h + 2 * RF, H), min(x0 + w + 2 * RF, W)
            crop = m(i1[:, :, ya:yb, xa:xb].contiguous(), i2[:, :, ya:yb, xa:xb].contiguous())
            a = full[:, :, y0:y0 + h, x0:x0 + w]
            b = crop[:, :, y0 - ya:y0 - ya + h, x0 - xa:x0 - xa + w]
            assert torch.equal(a, b), f"window {(y0, x0, h, w)}: max diff {float((a - b).abs().max()):.3e}"
        # and the crop is right: oracle on a 56x72 corner crop
        c1, c2 = i1[:, :, :56, W - 72:].contiguous(), i2[:, :, :56, W - 72:].contiguous()
        y = m(c1, c2).cpu().numpy()
    mo = O.MODELS["PFNetv1"]()
    y_or = mo.forward(mo.init_params(seed=1), c1.cpu().numpy(), c2.cpu().numpy())
    close(y, y_or, 1e-4 if dtype == "fp32" else 3e-2, "crop vs oracle")


def _loss_grads(m, i1, i2):
    from core.loss import GradLoss, PixelLoss, SSIMLoss
    l1, l2, l3 = SSIMLoss('ssim', weight=1.0), PixelLoss('l1', weight=0.01), GradLoss('l1', weight=0.1).to("cuda:0")
    for p in m.parameters():
        p.grad = None
    f = m(i1, i2)
    tot = l1(i1, i2, f) + l2(i1, i2, f, mode='max') + l3(i1, i2, f, mode='max')
    tot.backward()
    return float(tot.detach()), {k: p.grad.detach().clone() for k, p in m.named_parameters()}


@pytest.mark.parametrize("name,dtype,tol", [("PFNetv1", "fp32", 1e-4), ("PFNetv1", "bf16", 2e-3), ("DenseFuse", "bf16", 2e-3)])
def test_full_batch_gradient_is_mean_of_subbatch_gradients(name, dtype, tol):
    """B=32 256x256 (configs 2 and 3 per GPU).  Every loss is a batch mean, so grad(B=32) = mean of the four
    B=8 gradients.  bf16: activations are identical per sample either way; only the fp32 wgrad summation
    order over the batch differs."""
    B, S = 32, 256
    g = torch.Generator().manual_seed(7)
    i1, i2 = torch.rand(B, 1, S, S, generator=g).cuda(), torch.rand(B, 1, S, S, generator=g).cuda()
    with dtype_ctx(dtype):
        m = _model(name)
        tot, full = _loss_grads(m, i1, i2)
        acc, tots = None, []
        for c in range(4):
            t, gr = _loss_grads(m, i1[8 * c:8 * c + 8].contiguous(), i2[8 * c:8 * c + 8].contiguous())
            tots.append(t)
            acc = gr if acc is None else {k: acc[k] + v for k, v in gr.items()}
    assert abs(tot - np.mean(tots)) <= 1e-5 * abs(tot)
    for k in full:
        close(full[k].cpu().numpy(), (acc[k] / 4).cpu().numpy(), tol, k)


@pytest.mark.parametrize("name", ["NestFuse", "RFNNest"])
def test_nest_512_samples_do_not_interact(name):
    S = 512
    g = torch.Generator().manual_seed(9)
    i1, i2 = torch.rand(2, 1, S, S, generator=g).cuda(), torch.rand(2, 1, S, S, generator=g).cuda()
    with dtype_ctx("bf16"), torch.no_grad():
        m = _model(name)
        both = m(i1, i2)
        assert bool(torch.isfinite(both).all())
        _assert_alive_image(both, name)
        for b in range(2):
            one = m(i1[b:b + 1].contiguous(), i2[b:b + 1].contiguous())
            assert torch.equal(both[b:b + 1], one), f"sample {b}: max diff {float((both[b:b + 1] - one).abs().max()):.3e}"


@pytest.mark.parametrize("name", ["NestFuse", "RFNNest"])
def test_nest_512_backward_is_mean_of_per_sample_gradients(name):
    """Config 4 (NestFuse / RFN-Nest at 512 x 512), BACKWARD at full size: every loss is a batch mean and no layer couples samples, so
    the parameter gradients of a B = 2 step are the mean of the two B = 1 steps' -- a size-independent property that needs no CPU
    reference (the 36 x 44 oracle case is test_gpu_models.py).  bf16 engine: per-sample activations are identical either way; only the
    fp32 summation order of the weight gradients over the batch differs."""
    S = 512
    g = torch.Generator().manual_seed(9)
    i1, i2 = torch.rand(2, 1, S, S, generator=g).cuda(), torch.rand(2, 1, S, S, generator=g).cuda()
    with dtype_ctx("bf16"):
        m = _model(name)
        tot, full = _loss_grads(m, i1, i2)
        acc, tots = None, []
        for b in range(2):
            t, gr = _loss_grads(m, i1[b:b + 1].contiguous(), i2[b:b + 1].contiguous())
            tots.append(t)
            acc = gr if acc is None else {k: acc[k] + v for k, v in gr.items()}
        with torch.no_grad():
            _assert_alive_image(m(i1, i2), name)
    assert np.isfinite(tot) and abs(tot - np.mean(tots)) <= 1e-5 * abs(tot)
    gnorm = float(torch.sqrt(sum((g.double() ** 2).sum() for g in full.values())))
    assert gnorm > 1e-3, f"gradient norm {gnorm}: nothing flows"
    for k in full:
        assert bool(torch.isfinite(full[k]).all()) and float(full[k].abs().max()) > 0, k       # every parameter gets a gradient
        close(full[k].cpu().numpy(), (acc[k] / 2).cpu().numpy(), 2e-3, k)


def test_decode0_dma_kernels_vs_valu_family_256():
    """decode.0 of PFNetv1 (128 -> 128, 3 x 3) at B = 2, 256 x 256: the DMA-staged MFMA kernels (conv_dma_kernel forward / folded
    dgrad with ReLU mask, wgrad_dma_kernel) against the independent fp32-FMA kernel family on IDENTICAL bf16 operands -- 64 full
    32 x 16 tiles per image row band instead of the 48 x 40 cross-check of test_gpu_models.py.  Both families accumulate in fp32 and
    store bf16: outputs may differ by one bf16 rounding (2^-8 of the value; fold targets two)."""
    from mmif import tensor as T
    from mmif._lib import IMPL_MFMA, IMPL_VALU
    dev = "cuda:0"
    n, c, S = 2, 128, 256
    torch.manual_seed(21)
    x = T.BT.alloc(n, c, S, S, torch.bfloat16, dev); x.buf.normal_()
    gy = T.BT.alloc(n, c, S, S, torch.bfloat16, dev, halo=1, zero=True); gy.buf[:, :, 1:-1, 1:-1].normal_()
    gy = gy.as_folded()
    wt = torch.randn(c, c, 3, 3, device=dev) * 0.03
    b = torch.randn(c, device=dev)
    pk = T.PackedWeights(c, c, 3, dev); pk.pack(wt)
    wq = wt.bfloat16().float()          # the MFMA operand images are bf16: give the FMA family the same rounded weights
    ws = torch.empty(T.wgrad_workspace_bytes(c, c, 3) // 4 + 1, dtype=torch.float32, device=dev)
    res = {}
    for impl, w_ in ((IMPL_VALU, wq), (IMPL_MFMA, wt)):
        y = T.BT.alloc(n, c, S, S, torch.bfloat16, dev)
        gx = T.BT.alloc(n, c, S, S, torch.bfloat16, dev, halo=1, zero=True)
        dw, db = torch.zeros_like(wt), torch.zeros_like(b)
        T.conv_fwd(x, w_, b, y, c, c, 3, True, pk, impl)
        T.conv_dgrad(gy, w_, x, gx, c, c, 3, (1 << 16) - 1, 0, pk, impl, fold=True)
        T.conv_wgrad(x, gy, dw, db, c, c, 3, ws, False, impl)
        torch.cuda.synchronize()
        res[impl] = (y.buf.float().cpu().numpy(), gx.buf.float().cpu().numpy(), dw.cpu().numpy(), db.cpu().numpy())
    a, r = res[IMPL_MFMA], res[IMPL_VALU]
    ey = np.abs(a[0] - r[0]) / np.maximum(np.abs(r[0]), 1e-2 * np.abs(r[0]).max())
    assert ey.max() <= 2.0 ** -7, ("fwd", ey.max())               # one bf16 rounding of the value either way
    assert (ey > 0).mean() < 0.05                                  # ... and only where fp32 sums straddle a rounding boundary
    # dgrad: the fold targets (logical rows / columns 1 and h-2 / w-2 = stored 2 and h-1 / w-1) are sums of an interior and a halo value
    # -- one fp32 sum and ONE rounding in the DMA kernel's in-tile fold, two roundings + a bf16 add in "dgrad + fold kernel": where the two
    # parts cancel the difference is large RELATIVE to the small sum, so those pixels are held in absolute terms
    scale = np.abs(r[1]).max()
    tgt = np.zeros(r[1].shape[2:4], dtype=bool)
    tgt[[2, S - 1], :] = True
    tgt[:, [2, S - 1]] = True
    eg = np.abs(a[1] - r[1]) / np.maximum(np.abs(r[1]), 1e-2 * scale)
    inner = eg[:, :, ~tgt]
    assert inner.max() <= 2.0 ** -7 and (inner > 0).mean() < 0.05, ("dgrad", inner.max(), (inner > 0).mean())
    assert np.abs(a[1] - r[1])[:, :, tgt].max() <= 2.0 ** -6 * scale, "dgrad fold targets"
    assert np.abs(a[1][:, :, [0, S + 1], :]).max() == 0.0 and np.abs(a[1][:, :, :, [0, S + 1]]).max() == 0.0, "halo ring must stay zero"
    close(a[2], r[2], 1e-4, "dw")
    close(a[3], r[3], 1e-4, "db")


def test_encoder_round2_kernels_at_full_size():
    """B=32 256x256 (config 2): the fused encoder passes against the layer-wise kernels they replace, through properties that
    need no CPU reference: streaming forward == four layers BIT for bit; fused weight gradients exactly linear under a
    power-of-two scaling of the gradient and within fp32 summation-order noise of the layer-wise kernels; gather-form gradient
    chain within one bf16 rounding per accumulated contribution of the scatter form."""
    import os
    import core.model as M
    from mmif import engine as E
    from mmif import tensor as T
    B, S = 32, 256
    with dtype_ctx("bf16"):
        torch.manual_seed(3)
        m = M.PFNetv1().cuda()
        eng = E.PFNetv1Engine(m)
        g = torch.Generator().manual_seed(11)
        i1, i2 = torch.rand(B, 1, S, S, generator=g).cuda(), torch.rand(B, 1, S, S, generator=g).cuda()
        (i1, i2), _, _, _, dtype, impl = eng.prepare((i1, i2))
        dev = torch.device("cuda", 0)
        Fa, Fb = T.BT.alloc(B, 128, S, S, dtype, dev), T.BT.alloc(B, 128, S, S, dtype, dev)
        br = [(eng.enc[0], i1, 0), (eng.enc[1], i2, 8)]
        os.environ["MMIF_ENC_STREAM"] = "0"
        __import__("mmif.engine").engine.reload_switches()
        try:
            eng.enc_fwd_all(br, Fa, dtype, impl)
        finally:
            os.environ.pop("MMIF_ENC_STREAM")
            __import__("mmif.engine").engine.reload_switches()
        from mmif._lib import lib
        lib.mmif_debug_set_enc_stream2(0)          # round 2's streaming kernel: bit-identical to the layer-wise launches
        try:
            eng.enc_fwd_all(br, Fb, dtype, impl)
            torch.cuda.synchronize()
            assert torch.equal(Fa.buf.view(torch.int16), Fb.buf.view(torch.int16))
        finally:
            lib.mmif_debug_set_enc_stream2(2)
        # round 5's (the default; another accumulation order, csrc/enc_stream2.hip): x0 .. x3 within one bf16 rounding per stage
        Fb.buf.fill_(7.0)
        eng.enc_fwd_all(br, Fb, dtype, impl)
        torch.cuda.synchronize()
        for e in (0, 8):
            assert float((Fa.buf[:, e:e + 2].view(torch.int16) != Fb.buf[:, e:e + 2].view(torch.int16)).float().mean()) < 0.02, "x0"
            for k in (0, 1, 2, 3):
                a, b = Fa.buf[:, e + 2 * k:e + 2 * k + 2].float(), Fb.buf[:, e + 2 * k:e + 2 * k + 2].float()
                d = (a - b).abs()
                assert float(d.max()) <= (k + 1) * 2.0 ** -6 * float(a.abs().max()), (e, k, float(d.max()), float(a.abs().max()))
                assert float((d > 0).float().mean()) < 0.25, (e, k, float((d > 0).float().mean()))
        # weight gradients
        ws = eng.workspace(dev)
        G = T.BT.alloc(B, 64, S, S, dtype, dev, halo=1, zero=True)
        G.buf[:, :, 1:-1, 1:-1].copy_((torch.randn(B, 8, S, S, 8, generator=g) * (torch.rand(B, 8, S, S, 8, generator=g) > 0.5)).to(dev))
        G2 = T.BT.alloc(B, 64, S, S, dtype, dev, halo=1, zero=True)
        G2.buf.copy_(G.buf * 2)
        def grads():
            return [(torch.zeros_like(s.conv.weight), torch.zeros_like(s.conv.bias)) for s in eng.enc[0]]
        a, b, c = grads(), grads(), grads()
        T.dense_encoder_wgrad(i1, Fb.view(0, 6), G.as_folded().view(0, 8), a, ws)
        T.dense_encoder_wgrad(i1, Fb.view(0, 6), G2.as_folded().view(0, 8), b, ws)
        for k, nin in ((3, 6), (2, 4), (1, 2)):
            T.conv_wgrad(Fb.view(0, nin), G.as_folded().view(2 * k, 2), c[k][0], c[k][1], 16 * k, 16, 3, ws, False)
        T.image_in_wgrad(i1, G.as_folded().view(0, 2), c[0][0], c[0][1], 16, 3, ws, False)
        torch.cuda.synchronize()
        for k in range(4):
            assert torch.equal(b[k][0], 2 * a[k][0]) and torch.equal(b[k][1], 2 * a[k][1]), k       # exact: scaling by 2 commutes with every rounding
            close(a[k][0].cpu().numpy(), c[k][0].cpu().numpy(), 2e-4, f"dW{k}")
            close(a[k][1].cpu().numpy(), c[k][1].cpu().numpy(), 2e-4, f"db{k}")
        # gradient chain: gather vs scatter form on the same inputs
        res = {}
        for mode in ("1", "0"):
            GF = T.BT.alloc(B, 64, S, S, dtype, dev, halo=1, zero=True)
            GF.buf.copy_(G.buf)
            os.environ["MMIF_ENC_CHAIN"] = mode
            os.environ["MMIF_ENC_CHAIN_STREAM"] = "0"          # (the in-place gather-form launches; the streaming kernel is checked below)
            __import__("mmif.engine").engine.reload_switches()
            try:
                eng._assign_grad_views(dev)
                eng.enc_bwd(eng.enc[0], i1, Fb, GF.as_folded(), 0, 0, ws, impl)
            finally:
                os.environ.pop("MMIF_ENC_CHAIN")
                os.environ.pop("MMIF_ENC_CHAIN_STREAM")
                __import__("mmif.engine").engine.reload_switches()
            torch.cuda.synchronize()
            assert float(GF.buf[:, :, 0].float().abs().max()) == 0.0       # the halo ring stays zero (folded convention)
            res[mode] = GF.buf.float()
        assert torch.equal(res["1"][:, 6:], res["0"][:, 6:])               # g3 is an input
        d = (res["1"] - res["0"]).abs().max().item() / res["0"].abs().max().item()
        assert d < 3e-2, d
        # round 4: the chain as ONE streaming launch (csrc/enc_chain.hip) at full size: [g0 | g1 | g2 | g3] in its own buffer, within one
        # bf16 rounding per stage of the gather-form launches (10 strips x row segments x 32 images; same operand images)
        GF = T.BT.alloc(B, 64, S, S, dtype, dev, halo=1, zero=True)
        GF.buf.copy_(G.buf)
        eng.enc_bwd(eng.enc[0], i1, Fb, GF.as_folded(), 0, 0, ws, impl)
        torch.cuda.synchronize()
        assert torch.equal(GF.buf, G.buf)                                  # not in place: the inputs are untouched
        gz = eng.chain_out(eng.enc[0][0], Fb).buf.float()      # (the buffer the streaming chain just wrote: one per shape and branch slot)
        ref = res["1"][:, :, 1:-1, 1:-1]
        assert torch.equal(gz[:, 6:], ref[:, 6:])
        err = (gz - ref).abs() / ref.abs().max()
        assert float(err.max()) <= 2.0 ** -6 and float((err > 0).float().mean()) < 0.2, (float(err.max()), float((err > 0).float().mean()))


def test_decoder_round2_kernels_at_full_size():
    """B=32 256x256 (config 2), the wide decoder layers' backward through sign bytes and the thin layers' fused backward, against the
    separate kernels they replace -- properties that need no CPU reference: decode.1 (128 -> 64, every block masked) and decode.0
    (128 -> 128, PFNetv1's partial mask) give the SAME bits for gx / dW / db as mmif_conv2d_reflect_wgrad + _dgrad_folded reading the
    activations; the sign bytes equal [x > 0] of the whole tensor; weight gradients are exactly linear under a power-of-two scaling of
    the gradient; the DenseFuse chain's accumulate-onto-another-tensor form equals copy + accumulate bit for bit."""
    from mmif import tensor as T
    from mmif._lib import IMPL_MFMA
    B, S = 32, 256
    dev = torch.device("cuda", 0)
    with dtype_ctx("bf16"):
        g = torch.Generator().manual_seed(21)
        for cin, cout, mb in ((128, 64, (1 << 16) - 1), (128, 128, (1 << 6) | (1 << 7) | (1 << 14) | (1 << 15))):
            x = T.BT.alloc(B, cin, S, S, torch.bfloat16, dev)
            x.buf.copy_(torch.relu(torch.randn(x.buf.shape, generator=g)).to(dev))
            gy = T.BT.alloc(B, cout, S, S, torch.bfloat16, dev, halo=1, zero=True)
            gy.buf[:, :, 1:-1, 1:-1].copy_(torch.randn(B, cout // 8, S, S, 8, generator=g).to(dev))
            gy = gy.as_folded()
            wgt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.03).to(dev)
            pk = T.PackedWeights(cout, cin, 3, dev)
            pk.pack(wgt)
            ws = torch.empty(T.wgrad_workspace_bytes(cin, cout, 3) // 4 + 1, dtype=torch.float32, device=dev)
            signs = torch.empty(T.bwd_wide_signs_bytes(B, cin, S, S), dtype=torch.uint8, device=dev)
            gx_a = T.BT.alloc(B, cin, S, S, torch.bfloat16, dev, halo=1, zero=True)
            dw_a, db_a = torch.zeros(cout, cin, 3, 3, device=dev), torch.zeros(cout, device=dev)
            T.conv_bwd_wide(gy, x, gx_a, dw_a, db_a, cin, cout, 3, pk, mb, ws, signs)
            gx_b = T.BT.alloc(B, cin, S, S, torch.bfloat16, dev, halo=1, zero=True)
            dw_b, db_b = torch.zeros_like(dw_a), torch.zeros_like(db_a)
            T.conv_wgrad(x, gy, dw_b, db_b, cin, cout, 3, ws, False, IMPL_MFMA)
            T.conv_dgrad(gy, wgt, x, gx_b, cin, cout, 3, mb, 0, pk, IMPL_MFMA, fold=True)
            torch.cuda.synchronize()
            assert torch.equal(gx_a.buf.view(torch.int16), gx_b.buf.view(torch.int16)), (cin, cout)
            assert torch.equal(dw_a, dw_b) and torch.equal(db_a, db_b)
            pitch = (S // 16 + 2) * 16
            sg = signs.view(B, cin // 8, S + 2, pitch)[:, :, 1:S + 1, 16:16 + S]
            want = ((x.buf.float() > 0).to(torch.int32) << torch.arange(8, device=dev, dtype=torch.int32)).sum(-1).to(torch.uint8)
            assert torch.equal(sg, want)
            gy2 = T.BT.alloc(B, cout, S, S, torch.bfloat16, dev, halo=1, zero=True)          # linearity: g -> 4 g  =>  dW -> 4 dW exactly
            gy2.buf.copy_(gy.buf * 4)
            dw_c, db_c = torch.zeros_like(dw_a), torch.zeros_like(db_a)
            T.conv_bwd_wide(gy2.as_folded(), x, gx_a, dw_c, db_c, cin, cout, 3, pk, mb, ws, signs)
            torch.cuda.synchronize()
            assert torch.equal(dw_c, dw_a * 4) and torch.equal(db_c, db_a * 4)
            del x, gy, gy2, gx_a, gx_b, signs
        # accumulate-onto-another-tensor (the DenseFuse chain's form) == copy + accumulate, bit for bit
        gsrc = T.BT.alloc(B, 48, S, S, torch.bfloat16, dev, halo=1, zero=True)
        gsrc.buf[:, :, 1:-1, 1:-1].copy_(torch.randn(B, 6, S, S, 8, generator=g).to(dev))
        gsrc = gsrc.as_folded()
        xk = T.BT.alloc(B, 16, S, S, torch.bfloat16, dev)
        xk.buf.copy_(torch.relu(torch.randn(xk.buf.shape, generator=g)).to(dev))
        old = T.BT.alloc(B, 16, S, S, torch.bfloat16, dev, halo=1, zero=True)
        old.buf[:, :, 1:-1, 1:-1].copy_(torch.randn(B, 2, S, S, 8, generator=g).to(dev))
        old = old.as_folded()
        wv = (torch.randn(48, 16, 3, 3, generator=g) * 0.05).to(dev)      # virtual stacked layer 16 -> 48 (its dgrad maps 48 -> 16)
        pkv = T.PackedWeights(48, 16, 3, dev)
        pkv.pack(wv)
        dst_a = T.BT.alloc(B, 16, S, S, torch.bfloat16, dev, halo=1, zero=True)
        assert T.dgrad_onto_supported(gsrc, dst_a, 16, 48, 3)
        T.conv_dgrad_onto(gsrc, xk, old, dst_a, 16, 48, 3, 0b11, 0b11, pkv)
        dst_b = T.BT.alloc(B, 16, S, S, torch.bfloat16, dev, halo=1, zero=True)
        dst_b.buf.copy_(old.buf)
        T.conv_dgrad(gsrc, None, xk, dst_b.as_folded(), 16, 48, 3, 0b11, 0b11, pkv, IMPL_MFMA, fold=True)
        torch.cuda.synchronize()
        assert torch.equal(dst_a.buf.view(torch.int16), dst_b.buf.view(torch.int16))


def test_pfnetv2_pair_kernels_at_full_size():
    """PFNetv2's pair-conv kernels at B=32 256x256 (64 channel pairs): the strip forward is bit-identical to the
    one-output-per-thread kernel, the one-pass backward's operand gradients are bit-identical to pairconv_dgrad and its weight
    gradients agree with pairconv_wgrad and are exactly linear in g (x4 scaling is exact in bf16 / fp32)."""
    import os
    from mmif import tensor as T
    dev = "cuda:0"
    n, ch, h, w = 32, 64, 256, 256
    cb = ch // 8
    with dtype_ctx("bf16"):
        gen = torch.Generator().manual_seed(11)
        X = T.BT.alloc(n, 2 * ch, h, w, torch.bfloat16, dev)
        X.buf.copy_(torch.relu(torch.randn(X.buf.shape, generator=gen)).to(dev))
        G = T.BT.alloc(n, 2 * ch, h, w, torch.bfloat16, dev, halo=1, zero=True)
        G.buf[:, :, 1:-1, 1:-1].copy_(torch.randn(n, 2 * cb, h, w, 8, generator=gen).to(dev))
        G = G.as_folded()
        a, b, ga, gb = X.view(0, cb), X.view(cb, cb), G.view(0, cb), G.view(cb, cb)
        wgt = (torch.randn(2, 2, 3, 3, generator=gen) * 0.3).to(dev)
        bias = torch.randn(2, generator=gen).to(dev)
        outs = {}
        for mode in ("1", "0"):
            os.environ["MMIF_PAIR_STRIP"] = mode
            __import__("mmif.engine").engine.reload_switches()
            try:
                Y = T.BT.alloc(n, 2 * ch, h, w, torch.bfloat16, dev, zero=True)
                T.pairconv_fwd(a, b, wgt, bias, 2, Y.view(0, cb), Y.view(cb, cb), True)
                torch.cuda.synchronize()
                outs[mode] = Y.buf.view(torch.int16).clone()
            finally:
                os.environ.pop("MMIF_PAIR_STRIP", None)
                __import__("mmif.engine").engine.reload_switches()
        assert torch.equal(outs["1"], outs["0"]) and float(outs["0"].float().abs().max()) > 0
        ws = torch.empty(T.pairconv_wgrad_workspace_bytes() // 4, dtype=torch.float32, device=dev)
        GA = T.BT.alloc(n, 2 * ch, h, w, torch.bfloat16, dev, halo=1, zero=True)
        GB = T.BT.alloc(n, 2 * ch, h, w, torch.bfloat16, dev, halo=1, zero=True)
        dw1, db1 = torch.zeros(2, 2, 3, 3, device=dev), torch.zeros(2, device=dev)
        dw2, db2 = torch.zeros_like(dw1), torch.zeros_like(db1)
        bits = (1 << cb) - 1
        T.pairconv_bwd(ga, gb, wgt, 2, a, b, GA.view(0, cb), GA.view(cb, cb), dw1, db1, ws, bits)
        T.pairconv_dgrad(ga, gb, wgt, 2, a, b, GB.view(0, cb), GB.view(cb, cb), bits)
        T.pairconv_wgrad(a, b, ga, gb, 2, dw2, db2, ws)
        torch.cuda.synchronize()
        assert torch.equal(GA.buf.view(torch.int16), GB.buf.view(torch.int16))
        assert float((dw1 - dw2).abs().max()) <= 2e-4 * float(dw2.abs().max())
        assert float((db1 - db2).abs().max()) <= 2e-4 * max(1.0, float(db2.abs().max()))
        G.buf.mul_(4.0)
        dw4, db4 = torch.zeros_like(dw1), torch.zeros_like(db1)
        T.pairconv_bwd(ga, gb, wgt, 2, a, b, GA.view(0, cb), GA.view(cb, cb), dw4, db4, ws, bits)
        torch.cuda.synchronize()
        assert torch.equal(dw4, 4 * dw1) and torch.equal(db4, 4 * db1)


@pytest.mark.parametrize("name", ["PFNetv1", "DenseFuse"])
def test_parity_path_step_vs_fp32_fma_family_256(name):
    """The whole fp32 train step at the bench's resolution (4 pairs of 256 x 256: multi-tile launches of every split-operand kernel --
    the 64-channel forward / dgrad / wgrad, the four-wave thin forward / dgrad with 16-wide tiles, the thin weight gradient, the sign-map
    backward) against the SAME step on the fp32 FMA kernels (`MMIF_CONV_IMPL=valu`): fused image, the four loss values and every
    parameter gradient inside BASELINE's north-star tolerance (1e-3 relative) -- in fact ~1e-5 -- at a size the CPU oracle cannot run."""
    from core.loss import FusionLoss, GradLoss, PixelLoss, SSIMLoss
    shape = (4, 1, 256, 256)
    i1 = torch.from_numpy(O.closed_form_image(shape, 0.3)).cuda()
    i2 = torch.from_numpy(O.closed_form_image(shape, 1.7)).cuda()
    res = {}
    for impl in ("valu", "auto"):
        with dtype_ctx("fp32", impl):
            m = _model(name)
            m.train()
            fl = FusionLoss(SSIMLoss('ssim', weight=1.0), PixelLoss('l1', weight=0.01), GradLoss('l1', weight=0.1).cuda(), 'max', 'max')
            f = m(i1, i2)
            fl(i1, i2, f).backward()
            torch.cuda.synchronize()
            res[impl] = (f.detach().cpu().numpy(), fl.values.cpu().numpy(), {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters()})
    close(res["auto"][0], res["valu"][0], 1e-5, "fused image")
    assert np.abs(res["auto"][1] - res["valu"][1]).max() <= 1e-5, (res["auto"][1], res["valu"][1])
    worst = 0.0
    for k, g in res["valu"][2].items():
        worst = max(worst, close(res["auto"][2][k], g, 1e-3, f"d loss / d {k}"))
    assert worst <= 2e-4, worst   # (measured ~2e-5: the backward's two-piece products; a ReLU-decision flip would show as ~1e-3)


@pytest.mark.parametrize("name", ["PFNetv1", "DenseFuse"])
def test_round4_kernel_variants_are_bit_identical_at_the_headline_size(name):
    """BASELINE config 2 (batch 32, 256 x 256, bf16): one train step's fused image and EVERY parameter gradient with round 4's DMA-staged pair
    backward (bwd_pair_dma_kernel) and wide thin forward (thin_conv_async_kernel, two groups / three slots) against the register-staged
    kernels they replace -- bit for bit (same operand images, k-group orders, rounding points, fixed-order reductions)."""
    import core.model as M
    from mmif._lib import lib
    from gpu_util import dtype_ctx
    g = torch.Generator().manual_seed(21)
    i1, i2 = torch.rand(32, 1, 256, 256, generator=g).to(DEV), torch.rand(32, 1, 256, 256, generator=g).to(DEV)
    gy = torch.rand(32, 1, 256, 256, generator=g).to(DEV)
    res = {}
    try:
        for mode in (0, 1):
            lib.mmif_debug_set_bwd_pair_dma(mode)
            lib.mmif_debug_set_thin_wide(mode)
            with dtype_ctx("bf16"):
                torch.manual_seed(5)
                m = getattr(M, name)().to(DEV)
                y = m(i1, i2)
                y.backward(gy)
                torch.cuda.synchronize()
                res[mode] = (y.detach().clone(), [p.grad.detach().clone() for p in m.parameters()])
    finally:
        lib.mmif_debug_set_bwd_pair_dma(1)
        lib.mmif_debug_set_thin_wide(1)
    assert float(res[0][0].abs().max()) > 0
    assert torch.equal(res[0][0], res[1][0]), "fused image"
    for k, (a, b) in enumerate(zip(res[0][1], res[1][1])):
        assert float(a.abs().max()) > 0
        assert torch.equal(a, b), f"parameter gradient {k}"


@pytest.mark.parametrize("name", ["PFNetv1", "DenseFuse"])
def test_round5_encoder_kernels_at_the_headline_size(name):
    """BASELINE config 2 (batch 32, 256 x 256, bf16): one train step with round 5's encoder kernels against the kernels they replace.
    (a) the fused encoder backward (csrc/enc_bwd.hip: gradient chain + weight gradients of all four layers in one launch, a block per CU walking
    slices of rows) against enc_chain_bwd_kernel + enc_wgrad_kernel: identical fused image and decoder gradients (nothing upstream of them
    changed), encoder gradients within the summation-order noise (2e-3 of their maximum; same rounding points);
    (b) the streaming encoder forward's three forms (csrc/enc_stream2.hip 32- / 64-pixel strips, csrc/enc_stream.hip): another accumulation
    order inside a layer, same rounding points -- fused image within two bf16 steps of its range, every gradient within 2e-2 of its maximum."""
    import os
    import core.model as M
    from mmif import engine as E
    from mmif._lib import lib
    from gpu_util import dtype_ctx
    g = torch.Generator().manual_seed(23)
    i1, i2 = torch.rand(32, 1, 256, 256, generator=g).to(DEV), torch.rand(32, 1, 256, 256, generator=g).to(DEV)
    gy = torch.rand(32, 1, 256, 256, generator=g).to(DEV)

    def step():
        with dtype_ctx("bf16"):
            torch.manual_seed(5)
            m = getattr(M, name)().to(DEV)
            y = m(i1, i2)
            y.backward(gy)
            torch.cuda.synchronize()
            return y.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}

    ref = step()
    assert float(ref[0].abs().max()) > 0
    # (a)
    os.environ["MMIF_ENC_BWD_FUSED"] = "0"
    E.reload_switches()
    try:
        two = step()
    finally:
        os.environ.pop("MMIF_ENC_BWD_FUSED")
        E.reload_switches()
    assert torch.equal(ref[0], two[0]), "fused image"
    for k in ref[1]:
        a, b = ref[1][k].double(), two[1][k].double()
        assert float(b.abs().max()) > 0, k
        if "encode" in k:
            assert float((a - b).abs().max()) <= 2e-3 * float(b.abs().max()), (k, float((a - b).abs().max()), float(b.abs().max()))
        else:
            assert torch.equal(ref[1][k], two[1][k]), k
    # (b)
    try:
        for mode in (1, 0):
            lib.mmif_debug_set_enc_stream2(mode)
            alt = step()
            assert float((ref[0].double() - alt[0].double()).abs().max()) <= 2 * 2.0 ** -8 * float(ref[0].abs().max()), mode
            for k in ref[1]:
                a, b = ref[1][k].double(), alt[1][k].double()
                assert float((a - b).abs().max()) <= 2e-2 * float(b.abs().max()), (mode, k, float((a - b).abs().max()), float(b.abs().max()))
    finally:
        lib.mmif_debug_set_enc_stream2(2)


@pytest.mark.parametrize("name", ["PFNetv1", "DenseFuse", "VIFNet"])
def test_deferred_weight_gradient_reduces_are_bit_identical_at_the_headline_size(name):
    """csrc/reduce_defer.hip: the decoder's five weight-gradient reduce launches queued and run as ONE launch ($MMIF_DEFER_REDUCE=1; off by default: no faster)
    against a reduce right after every producer: the same sums in the same order -- every parameter gradient bit for bit (batch 32, 256 x 256,
    bf16), nothing left queued after the backward."""
    import os
    import core.model as M
    from mmif import engine as E
    from mmif._lib import lib
    from gpu_util import dtype_ctx
    g = torch.Generator().manual_seed(29)
    i1, i2 = torch.rand(32, 1, 256, 256, generator=g).to(DEV), torch.rand(32, 1, 256, 256, generator=g).to(DEV)
    gy = torch.rand(32, 1, 256, 256, generator=g).to(DEV)
    res = {}
    try:
        for mode in ("1", "0"):
            os.environ["MMIF_DEFER_REDUCE"] = mode
            E.reload_switches()
            with dtype_ctx("bf16"):
                torch.manual_seed(5)
                m = getattr(M, name)().to(DEV)
                for _ in range(2):      # (twice: the second backward re-uses the arena)
                    m.zero_grad(set_to_none=True)
                    y = m(i1, i2)
                    y.backward(gy)
                torch.cuda.synchronize()
                assert lib.mmif_reduce_defer_pending() == 0
                res[mode] = (y.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()})
    finally:
        os.environ.pop("MMIF_DEFER_REDUCE", None)
        E.reload_switches()
    assert torch.equal(res["0"][0], res["1"][0])
    for k in res["0"][1]:
        assert float(res["0"][1][k].abs().max()) > 0, k
        assert torch.equal(res["0"][1][k], res["1"][1][k]), k


def test_deferred_reduces_fall_back_when_the_arena_is_too_small():
    """mmif_reduce_defer_begin with an arena that holds no layer's partial sums: every reduce runs right behind its producer as without
    deferral (nothing queued), gradients bit-identical; an arena that holds only the small layers queues those and reduces the rest at once."""
    import os
    import core.model as M
    from mmif import engine as E
    from mmif import tensor as T
    from mmif._lib import lib, check
    from gpu_util import dtype_ctx
    g = torch.Generator().manual_seed(31)
    i1, i2 = torch.rand(2, 1, 64, 80, generator=g).to(DEV), torch.rand(2, 1, 64, 80, generator=g).to(DEV)
    gy = torch.rand(2, 1, 64, 80, generator=g).to(DEV)
    with dtype_ctx("bf16"):
        torch.manual_seed(5)
        m = M.PFNetv1().to(DEV)
        y = m(i1, i2)
        y.backward(gy)
        torch.cuda.synchronize()
        ref = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        for arena_bytes in (256, 4 << 20):
            arena = torch.empty(arena_bytes // 4, dtype=torch.float32, device=DEV)
            # (the engine itself only defers with $MMIF_DEFER_REDUCE=1: drive the queue from outside around a plain backward)
            m.zero_grad(set_to_none=True)
            y = m(i1, i2)
            check(lib.mmif_reduce_defer_begin(arena.data_ptr(), arena.numel() * 4), "begin")
            y.backward(gy)
            queued = lib.mmif_reduce_defer_pending()
            check(lib.mmif_reduce_defer_flush(0, T.stream_ptr()), "flush")
            torch.cuda.synchronize()
            assert lib.mmif_reduce_defer_pending() == 0
            assert (queued == 0) if arena_bytes == 256 else (queued > 0), (arena_bytes, queued)
            for k, p in m.named_parameters():
                assert torch.equal(p.grad, ref[k]), (arena_bytes, k)
