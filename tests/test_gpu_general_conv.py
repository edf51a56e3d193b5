"""Row n4, first part: the general ConvLayer kernels (csrc/conv_general.hip) behind the reference's ConvLayer signature --
kernel sizes 5/7, stride 2, zero padding, ConvTranspose2d -- against the golden vectors captured from the reference
(tests/golden/f11_general_conv.npz) and against the numpy oracle on larger seeded shapes."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import fusion_oracle as O
from gpu_util import G, close, load_closed_form
from test_oracle_golden import F11_CASES, f11_tensors

pytestmark = pytest.mark.gpu


def _layer(cin, cout, k, stride, transposed, pmode, relu):
    from core.block import ConvLayer
    return ConvLayer(cin, cout, ksize=k, stride=stride, act=nn.ReLU if relu else None, layer=nn.ConvTranspose2d if transposed else nn.Conv2d,
                     padding_mode=pmode)


@pytest.mark.parametrize("case", F11_CASES, ids=[c[0] for c in F11_CASES])
def test_general_conv_layer_vs_golden(case):
    name, cin, cout, k, stride, transposed, pmode, relu, N, H, W = case
    g = np.load(os.path.join(G, "f11_general_conv.npz"))
    layer = load_closed_form(_layer(cin, cout, k, stride, transposed, pmode, relu), 11).cuda()
    assert layer._gen and not layer._hip
    _, _, x = f11_tensors(case)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    y = layer(xt)
    assert tuple(y.shape) == g[name + "_y"].shape
    y.backward(torch.from_numpy(O.closed_form_signed(tuple(y.shape), 1.5, 1.0)).cuda())
    conv = layer.layers[0]
    close(y.detach().cpu().numpy(), g[name + "_y"], 1e-5, "y")
    close(xt.grad.cpu().numpy(), g[name + "_dx"], 1e-5, "dx")
    close(conv.weight.grad.cpu().numpy(), g[name + "_dw"], 1e-5, "dw")
    close(conv.bias.grad.cpu().numpy(), g[name + "_db"], 1e-5, "db")


BIG = [(16, 32, 7, 1, False, "reflect", 3, 70, 45), (32, 16, 5, 1, False, "reflect", 2, 33, 97), (32, 64, 3, 2, False, "reflect", 3, 65, 50),
       (64, 128, 3, 2, False, "reflect", 2, 32, 32), (20, 12, 5, 2, False, "zeros", 2, 41, 38), (3, 16, 5, 1, False, "reflect", 2, 50, 50),
       (128, 64, 3, 2, True, "zeros", 2, 17, 23), (256, 128, 3, 2, True, "zeros", 1, 16, 16), (16, 1, 5, 1, False, "reflect", 2, 64, 64)]


@pytest.mark.parametrize("cin,cout,k,stride,transposed,pmode,N,H,W", BIG, ids=[f"{a}-{b}-k{k}s{s}{'T' if t else ''}-{n}x{h}x{w}" for a, b, k, s, t, _, n, h, w in BIG])
def test_general_conv_kernels_vs_oracle_on_ragged_shapes(cin, cout, k, stride, transposed, pmode, N, H, W):
    """tiles that straddle the image, channel counts that are no multiple of the kernels' groups (8 / 4), many tile groups in
    the weight-gradient reduction."""
    from mmif import tensor as T
    rng = np.random.default_rng(cin * 131 + cout * 7 + k + H)
    x = rng.standard_normal((N, cin, H, W), dtype=np.float32)
    w = (rng.standard_normal((cin, cout, k, k) if transposed else (cout, cin, k, k), dtype=np.float32) * 0.1).astype(np.float32)
    b = rng.standard_normal(cout, dtype=np.float32)
    p = k // 2
    dev = "cuda:0"
    xt, wt, bt = (torch.from_numpy(a).to(dev) for a in (x, w, b))
    if transposed:
        y_ref = O.conv_transpose2d_fwd(x, w, b, stride, p, 1, True)
        y = T.gconvt_fwd(xt, wt, bt, stride, p, 1, True)
    else:
        y_ref = O.conv2d_general_fwd(x, w, b, stride, p, pmode == "reflect", True)
        y = T.gconv_fwd(xt, wt, bt, stride, p, pmode == "reflect", True)
    close(y.cpu().numpy(), y_ref, 2e-5, "y")
    gy = rng.standard_normal(y_ref.shape, dtype=np.float32)
    gm = T.relu_bwd(torch.from_numpy(gy).to(dev), y)
    if transposed:
        dx_ref, dw_ref, db_ref = O.conv_transpose2d_bwd(x, w, y_ref, gy, stride, p, 1, True)
        dx, dw, db = T.gconvt_dgrad(gm, wt, tuple(x.shape), stride, p, 1), T.gconvt_wgrad(xt, gm, k, stride, p, 1), T.channel_sum(gm)
    else:
        dx_ref, dw_ref, db_ref = O.conv2d_general_bwd(x, w, y_ref, gy, stride, p, pmode == "reflect", True)
        dx = T.gconv_dgrad(gm, wt, tuple(x.shape), stride, p, pmode == "reflect")
        dw, db = T.gconv_wgrad(xt, gm, k, stride, p, pmode == "reflect")
    close(dx.cpu().numpy(), dx_ref, 3e-5, "dx")
    close(dw.cpu().numpy(), dw_ref, 3e-5, "dw")
    close(db.cpu().numpy(), db_ref, 3e-5, "db")


@pytest.mark.parametrize("tag,scale,shape", [("bl_x2", 2, (2, 3, 5, 7)), ("bl_x8", 8, (1, 4, 4, 3)), ("bl_x2_row", 2, (1, 2, 1, 6))])
def test_bilinear_upsample_vs_golden(tag, scale, shape):
    """Upsample('bilinear', scale) of the mirror == the reference's (nn.Upsample align_corners=True), forward and input gradient."""
    from core.block import Upsample
    g = np.load(os.path.join(G, "f12_n4_models.npz"))
    up = Upsample('bilinear', scale)
    x = torch.from_numpy(O.closed_form_signed(shape, 0.4, 1.0)).cuda().requires_grad_(True)
    tgt = (shape[0], shape[1], shape[2] * scale, shape[3] * scale)
    y = up(x, torch.Size(tgt))
    y.backward(torch.from_numpy(O.closed_form_signed(tgt, 1.3, 1.0)).cuda())
    close(y.detach().cpu().numpy(), g[tag + "_y"], 1e-5, "y")
    close(x.grad.cpu().numpy(), g[tag + "_dx"], 2e-5, "dx")


@pytest.mark.parametrize("scale,shape", [(2, (2, 5, 33, 47)), (8, (1, 3, 9, 6)), (4, (1, 2, 1, 1)), (3, (2, 1, 17, 2))])
def test_bilinear_kernels_vs_oracle(scale, shape):
    from mmif import tensor as T
    rng = np.random.default_rng(scale * 100 + shape[2])
    x = rng.standard_normal(shape, dtype=np.float32)
    y = T.bilinear_up_fwd(torch.from_numpy(x).cuda(), scale)
    close(y.cpu().numpy(), O.bilinear_up_fwd(x, scale), 1e-5, "y")
    gy = rng.standard_normal(tuple(y.shape), dtype=np.float32)
    dx = T.bilinear_up_bwd(torch.from_numpy(gy).cuda(), shape[2:])
    close(dx.cpu().numpy(), O.bilinear_up_bwd(gy, shape[2:]), 2e-5, "dx")


N4_MODELS = [("DeepFuse", (2, 1, 32, 32)), ("DeepFuse", (1, 1, 21, 30)), ("DBNet", (2, 1, 32, 32)), ("DBNet", (1, 1, 40, 24)), ("DBNet", (1, 1, 37, 53))]


@pytest.mark.parametrize("name,shape", N4_MODELS, ids=[f"{n}-{s[0]}x{s[2]}x{s[3]}" for n, s in N4_MODELS])
def test_n4_models_vs_golden(name, shape):
    """DeepFuse (k 5/7) and DBNet (stride 2 + bilinear x8, incl. the crop to an odd target size) layer by layer on the HIP
    kernels, fp32: fused image and every parameter gradient against the reference's (closed-form weights and inputs)."""
    import core.model as M
    from gpu_util import close_digest, dtype_ctx
    g = np.load(os.path.join(G, "f12_n4_models.npz"))
    tag = f"{name}_{shape[0]}x{shape[2]}x{shape[3]}"
    with dtype_ctx("fp32"):
        model = load_closed_form(getattr(M, name)(), 2).cuda()
        i1, i2 = (torch.from_numpy(O.closed_form_image(shape, p)).cuda() for p in (0.3, 1.7))
        y = model(i1, i2)
        y.backward(torch.from_numpy(O.closed_form_signed(shape, 0.9, 1.0)).cuda())
    close(y.detach().cpu().numpy(), g[tag + "__y"], 1e-4, "fused image")
    for k, p in model.named_parameters():
        close_digest(p.grad.cpu().numpy(), g[f"{tag}__dp_{k}"], 2e-4, k)


# ------------------------------------------------------------------ norm / activation epilogues (golden F13)
from test_oracle_golden import F13_CASES, f13_params  # noqa: E402

_NORMS = {"bn": nn.BatchNorm2d, "gn": nn.GroupNorm, None: None}
_ACTS = {"relu": nn.ReLU, "leaky": nn.LeakyReLU, "tanh": nn.Tanh, None: None}


@pytest.mark.parametrize("case", F13_CASES, ids=[c[0] for c in F13_CASES])
def test_norm_act_conv_layer_vs_golden(case):
    """ConvLayer(norm=BatchNorm2d | GroupNorm, act=ReLU | LeakyReLU | Tanh | None) of the mirror: HIP conv + csrc/norm.hip epilogue
    against the reference: y, dx, every parameter gradient and the BatchNorm buffers after the step (train and eval mode)."""
    from core.block import ConvLayer
    name, cin, cout, k, stride, transposed, norm, act, train, N, H, W = case
    g = np.load(os.path.join(G, "f13_n4_norm.npz"))
    layer = ConvLayer(cin, cout, ksize=k, stride=stride, norm=_NORMS[norm], act=_ACTS[act], layer=nn.ConvTranspose2d if transposed else nn.Conv2d)
    assert layer._epilogue and not layer._hip and not layer._gen
    P = f13_params(case)
    sd = layer.state_dict()
    layer.load_state_dict({kk: torch.from_numpy(np.asarray(P[kk])).to(sd[kk].dtype) if kk != "layers.1.num_batches_tracked" else torch.zeros((), dtype=torch.long)
                           for kk in sd})
    layer = layer.cuda().train(train)
    want_dx = not (cin == 1 and layer._conv_hot)   # (the hot-path image-input conv gives no gradient w.r.t. the image)
    x = torch.from_numpy(O.closed_form_signed((N, cin, H, W), 0.5, 1.0)).cuda().requires_grad_(want_dx)
    y = layer(x)
    y.backward(torch.from_numpy(O.closed_form_signed(tuple(y.shape), 1.5, 1.0)).cuda())
    close(y.detach().cpu().numpy(), g[name + "_y"], 2e-5, "y")
    if want_dx:
        close(x.grad.cpu().numpy(), g[name + "_dx"], 5e-5, "dx")
    for kname, p in layer.named_parameters():
        ref = g[f"{name}_dp_{kname}"]
        if norm and kname == "layers.0.bias":   # mathematically zero (the norm removes the mean): rounding noise on both sides
            assert np.abs(p.grad.cpu().numpy() - ref).max() <= 1e-4
        else:
            close(p.grad.cpu().numpy(), ref, 5e-5, kname)
    for kname, b in layer.named_buffers():
        close(b.detach().cpu().numpy().astype(np.float32), g[f"{name}_buf_{kname}"], 2e-5, kname, allow_zero=kname.endswith("num_batches_tracked"))


N4B_MODELS = [("SEDRFuse", (2, 1, 32, 32)), ("IFCNN", (2, 1, 32, 32)), ("IFCNN", (1, 1, 21, 30)), ("DIFNet", (2, 1, 32, 32)), ("PMGI", (2, 1, 32, 32)),
              ("PMGI", (1, 1, 19, 26))]


@pytest.mark.parametrize("name,shape", N4B_MODELS, ids=[f"{n}-{s[0]}x{s[2]}x{s[3]}" for n, s in N4B_MODELS])
def test_n4_norm_models_vs_golden(name, shape):
    """SEDRFuse (GroupNorm, stride 2, ConvTranspose2d, ResBlock), IFCNN / DIFNet (BatchNorm, k = 7), PMGI (BatchNorm + LeakyReLU,
    k = 5, Tanh) in training mode on the HIP kernels: fused image, every parameter gradient and every BatchNorm buffer."""
    import core.model as M
    from gpu_util import close_digest, dtype_ctx
    g = np.load(os.path.join(G, "f13_n4_norm.npz"))
    tag = f"{name}_{shape[0]}x{shape[2]}x{shape[3]}"
    with dtype_ctx("fp32"):
        model = load_closed_form(getattr(M, name)(), 2)
        for mod in model.modules():
            if isinstance(mod, nn.BatchNorm2d):
                mod.running_var.abs_().add_(0.5)
                mod.num_batches_tracked.zero_()
        model = model.cuda().train()
        i1, i2 = (torch.from_numpy(O.closed_form_image(shape, p)).cuda() for p in (0.3, 1.7))
        y = model(i1, i2)
        y.backward(torch.from_numpy(O.closed_form_signed(tuple(y.shape), 0.9, 1.0)).cuda())
    close(y.detach().cpu().numpy(), g[tag + "__y"], 2e-4, "fused image")
    for k, p in model.named_parameters():
        if f"{tag}__dp_{k}" not in g.files:
            assert p.grad is None, k   # PMGI's unused transfer1[1]
            continue
        dg = g[f"{tag}__dp_{k}"]
        if k.endswith("bias") and dg[1] < 1e-5 * p.numel() * max(1.0, abs(float(y.numel()))) ** 0.5:
            # a conv bias whose effect the following norm removes: the exact gradient is zero, both sides hold rounding noise
            assert float(p.grad.abs().sum()) < 1e-5 * p.numel() * float(y.numel()) ** 0.5, k
            continue
        close_digest(p.grad.cpu().numpy(), dg, 2e-3, k)   # (fp32 through several BatchNorms; max-fusion ties)
    for k, b in model.named_buffers():
        close_digest(b.detach().cpu().numpy().astype(np.float32), g[f"{tag}__buf_{k}"], 2e-4, k, allow_zero=k.endswith("num_batches_tracked"))


N4C_MODELS = [("UNFusion", (1, 1, 32, 32)), ("UNFusion", (1, 1, 37, 53)), ("MAFusion", (1, 1, 32, 32)), ("MAFusion", (1, 1, 40, 24))]


@pytest.mark.parametrize("name,shape", N4C_MODELS, ids=[f"{n}-{s[0]}x{s[2]}x{s[3]}" for n, s in N4C_MODELS])
def test_n4_nested_models_vs_golden(name, shape):
    """UNFusion (stride-2 ConvLayers, NestEncoder, 'wavg' fusion, bilinear NestDecoder) and MAFusion (FSDecoder: bilinear x2/x4/x8,
    max-pool /2 /4) -- compositions of the n4 primitives -- against the reference, fp32, odd sizes included."""
    import core.model as M
    from gpu_util import close_digest, dtype_ctx
    g = np.load(os.path.join(G, "f14_n4_nested.npz"))
    tag = f"{name}_{shape[0]}x{shape[2]}x{shape[3]}"
    with dtype_ctx("fp32"):
        model = load_closed_form(getattr(M, name)(), 2).cuda()
        i1, i2 = (torch.from_numpy(O.closed_form_image(shape, p)).cuda() for p in (0.3, 1.7))
        y = model(i1, i2)
        y.backward(torch.from_numpy(O.closed_form_signed(tuple(y.shape), 0.9, 1.0)).cuda())
    close(y.detach().cpu().numpy(), g[tag + "__y"], 2e-4, "fused image")
    for k, p in model.named_parameters():
        close_digest(p.grad.cpu().numpy(), g[f"{tag}__dp_{k}"], 2e-3, k)


# ------------------------------------------------------------------ depth-wise ConvLayer, ReLU6, Res2ConvBlock, Res2Fusion (golden F15)
from test_oracle_golden import F15_LAYERS  # noqa: E402


@pytest.mark.parametrize("case", F15_LAYERS, ids=[c[0] for c in F15_LAYERS])
def test_depthwise_and_relu6_layers_vs_golden(case):
    from core.block import ConvLayer
    name, kw, shape = case
    kw = dict(kw)
    if kw.get("act") == "relu6":
        kw["act"] = nn.ReLU6
    g = np.load(os.path.join(G, "f15_n4_res2.npz"))
    layer = load_closed_form(ConvLayer(**kw), 15).cuda()
    assert layer._epilogue and (layer._depthwise or kw["act"] is nn.ReLU6)
    x = torch.from_numpy(O.closed_form_signed(shape, 0.5, 8.0 if name == "relu6_k1" else 1.0)).cuda().requires_grad_(True)
    y = layer(x)
    y.backward(torch.from_numpy(O.closed_form_signed(tuple(y.shape), 1.5, 1.0)).cuda())
    close(y.detach().cpu().numpy(), g[name + "_y"], 1e-5, "y")
    close(x.grad.cpu().numpy(), g[name + "_dx"], 2e-5, "dx")
    for kname, p in layer.named_parameters():
        close(p.grad.cpu().numpy(), g[f"{name}_dp_{kname}"], 3e-5, kname)


def test_res2_conv_block_vs_golden():
    """Res2ConvBlock(16, 32, 4): point-wise expand + ReLU6, hierarchical depth-wise convs, point-wise project, projected residual."""
    from core.block import Res2ConvBlock
    g = np.load(os.path.join(G, "f15_n4_res2.npz"))
    blk = load_closed_form(Res2ConvBlock(16, 32, 4), 15).cuda()
    x = torch.from_numpy(O.closed_form_signed((2, 16, 10, 12), 0.5, 1.0)).cuda().requires_grad_(True)
    y = blk(x)
    y.backward(torch.from_numpy(O.closed_form_signed(tuple(y.shape), 1.5, 1.0)).cuda())
    close(y.detach().cpu().numpy(), g["res2block_y"], 2e-5, "y")
    close(x.grad.cpu().numpy(), g["res2block_dx"], 5e-5, "dx")
    for kname, p in blk.named_parameters():
        if f"res2block_dp_{kname}" in g.files:
            close(p.grad.cpu().numpy(), g[f"res2block_dp_{kname}"], 1e-4, kname)
        else:
            assert p.grad is None, kname   # the inherited, never-called SepConvBlock.dwconv


@pytest.mark.parametrize("shape", [(1, 1, 32, 32), (2, 1, 24, 40)], ids=["1x32x32", "2x24x40"])
def test_res2fusion_vs_golden(shape):
    import core.model as M
    from gpu_util import close_digest, dtype_ctx
    g = np.load(os.path.join(G, "f15_n4_res2.npz"))
    tag = f"Res2Fusion_{shape[0]}x{shape[2]}x{shape[3]}"
    with dtype_ctx("fp32"):
        model = load_closed_form(M.Res2Fusion(), 2).cuda()
        i1, i2 = (torch.from_numpy(O.closed_form_image(shape, p)).cuda() for p in (0.3, 1.7))
        y = model(i1, i2)
        y.backward(torch.from_numpy(O.closed_form_signed(tuple(y.shape), 0.9, 1.0)).cuda())
    close(y.detach().cpu().numpy(), g[tag + "__y"], 2e-4, "fused image")
    for k, p in model.named_parameters():
        if f"{tag}__dp_{k}" in g.files:
            close_digest(p.grad.cpu().numpy(), g[f"{tag}__dp_{k}"], 2e-3, k)
        else:
            assert p.grad is None, k


# ------------------------------------------------------------------ resampling glue (max-pool, nearest up-sampling, reflect pad / crop)
@pytest.mark.parametrize("k,shape", [(2, (2, 5, 13, 18)), (4, (1, 3, 17, 16)), (2, (1, 1, 2, 2))])
def test_maxpool_nchw_vs_numpy(k, shape):
    from core.block import MaxPool2d
    rng = np.random.default_rng(k + shape[2])
    x = rng.integers(-3, 4, size=shape).astype(np.float32)          # small integers: plenty of ties (first maximum wins)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    y = MaxPool2d(k, k)(xt)
    n, c, h, w = shape
    ho, wo = h // k, w // k
    win = x[:, :, :ho * k, :wo * k].reshape(n, c, ho, k, wo, k).transpose(0, 1, 2, 4, 3, 5).reshape(n, c, ho, wo, k * k)
    assert np.array_equal(y.detach().cpu().numpy(), win.max(-1))
    g = rng.standard_normal((n, c, ho, wo)).astype(np.float32)
    y.backward(torch.from_numpy(g).cuda())
    dx = np.zeros((n, c, ho, wo, k * k), dtype=np.float32)
    np.put_along_axis(dx, win.argmax(-1)[..., None], g[..., None], axis=-1)      # argmax = first maximum
    ref = np.zeros(shape, dtype=np.float32)
    ref[:, :, :ho * k, :wo * k] = dx.reshape(n, c, ho, wo, k, k).transpose(0, 1, 2, 4, 3, 5).reshape(n, c, ho * k, wo * k)
    assert np.array_equal(xt.grad.cpu().numpy(), ref)


@pytest.mark.parametrize("scale,shape", [(2, (2, 3, 5, 7)), (4, (1, 2, 3, 2))])
def test_nearest_upsample_vs_numpy(scale, shape):
    from core.block import Upsample
    rng = np.random.default_rng(scale)
    x = rng.standard_normal(shape).astype(np.float32)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    tgt = (shape[0], shape[1], shape[2] * scale, shape[3] * scale)
    y = Upsample('nearest', scale)(xt, torch.Size(tgt))
    assert np.array_equal(y.detach().cpu().numpy(), x.repeat(scale, axis=2).repeat(scale, axis=3))
    g = rng.standard_normal(tgt).astype(np.float32)
    y.backward(torch.from_numpy(g).cuda())
    ref = g.reshape(shape[0], shape[1], shape[2], scale, shape[3], scale).sum(axis=(3, 5))
    close(xt.grad.cpu().numpy(), ref, 1e-6, "dx")


@pytest.mark.parametrize("pads,shape", [((1, 2, 0, 1), (2, 3, 6, 7)), ((-1, -2, -1, 0), (1, 2, 9, 8)), ((2, -1, 3, 3), (1, 2, 5, 6)), ((0, 0, 1, 0), (1, 1, 2, 1))])
def test_reflect_pad_and_crop_vs_numpy(pads, shape):
    """the shape-matching tail of Upsample / Downsample: nn.ReflectionPad2d((l, r, t, b)), negative amounts crop; adjoint by scatter."""
    from core.block import _ReflectPadFn
    l, r, t, b = pads
    rng = np.random.default_rng(abs(l) + shape[2])
    x = rng.standard_normal(shape).astype(np.float32)
    n, c, h, w = shape
    H, W = h + t + b, w + l + r
    R = lambda i, L: np.where(np.abs(i) >= L, 2 * (L - 1) - np.abs(i), np.abs(i))
    sy, sx = R(np.arange(H) - t, h), R(np.arange(W) - l, w)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    y = _ReflectPadFn.apply(xt, pads)
    assert np.array_equal(y.detach().cpu().numpy(), x[:, :, sy][:, :, :, sx])
    g = rng.standard_normal((n, c, H, W)).astype(np.float32)
    y.backward(torch.from_numpy(g).cuda())
    ref = np.zeros(shape, dtype=np.float64)
    np.add.at(ref, (slice(None), slice(None), sy[:, None], sx[None, :]), g)
    close(xt.grad.cpu().numpy(), ref.astype(np.float32), 1e-6, "dx")
