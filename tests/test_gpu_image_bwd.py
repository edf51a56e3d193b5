"""Round 6: the whole backward of the decoders' last layer, ConvLayer(16, 1, 3x3, reflect) (reference core/model.py:86,178;
core/block.py:26-99), as ONE launch (csrc/image_bwd.hip, mmif_conv2d_image_out_bwd): the input gradient with the reflect-padding adjoint
applied and the ReLU mask of the layer's input, the weight gradient and the bias gradient.

Checked against (i) the fp64 definition through torch autograd on the same bf16 activations (F.pad(mode='reflect') + conv2d, the
reference's own operators): dW / db to 1e-5, dL/dx to one bf16 rounding; (ii) the three launches it replaces (weight gradient, input
gradient on the padded domain, fold of the halo): identical weight gradients up to the order of the per-block partial sums, input
gradients equal up to the second rounding the fold kernel adds on the fold targets; the halo ring stays zero."""
import numpy as np
import pytest
import torch

from gpu_util import close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [(2, 37, 53), (1, 4, 4), (3, 64, 96), (1, 8, 33), (2, 9, 32), (1, 5, 70), (4, 256, 256), (1, 130, 258)]


def _case(n, h, w, relu_out, seed):
    from mmif import tensor as T
    g = torch.Generator().manual_seed(seed)
    x = torch.relu(torch.randn(n, 16, h, w, generator=g)).bfloat16().float()        # the previous layer's ReLU output, bf16 values
    wt = torch.randn(1, 16, 3, 3, generator=g) * 0.2
    gimg = torch.randn(n, 1, h, w, generator=g)
    # fp64 definition
    xd = x.double().requires_grad_(True)
    wd = wt.double().requires_grad_(True)
    bd = torch.zeros(1, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(torch.nn.functional.pad(xd, (1, 1, 1, 1), mode="reflect"), wd, bd)
    yimg = None
    if relu_out:
        yimg = torch.relu(y.detach()).float()
        y = torch.relu(y)
    (y * gimg.double()).sum().backward()
    ref_dx = (xd.grad * (x > 0)).float()       # the ReLU of the previous layer rides in this layer's input gradient (mask x > 0)
    xb = T.BT.from_nchw(x.to(DEV), torch.bfloat16)
    return xb, x, wt.to(DEV), gimg.to(DEV).contiguous(), (yimg.to(DEV).contiguous() if yimg is not None else None), ref_dx, wd.grad.float(), bd.grad.float()


@pytest.mark.parametrize("relu_out", [False, True])
@pytest.mark.parametrize("n,h,w", SHAPES, ids=[f"{n}x{h}x{w}" for n, h, w in SHAPES])
def test_image_out_bwd_equals_definition_and_the_three_launches(n, h, w, relu_out):
    from mmif import tensor as T
    xb, x, wt, gimg, yimg, ref_dx, ref_dw, ref_db = _case(n, h, w, relu_out, 100 * h + w)
    assert T.image_out_bwd_supported(xb, 16, 3)
    ws = torch.empty(T.image_wgrad_workspace_bytes(16, 3) // 4 + 1, dtype=torch.float32, device=DEV)
    # ---- the fused launch
    gx = T.BT.alloc(n, 16, h, w, torch.bfloat16, DEV, halo=1, zero=True)
    dw, db = torch.full((1, 16, 3, 3), 7.0, device=DEV), torch.full((1,), 7.0, device=DEV)
    out = T.image_out_bwd(xb, gimg, yimg, wt, gx, dw, db, 16, 3, ws)
    torch.cuda.synchronize()
    ring = gx.buf.clone()
    ring[:, :, 1:-1, 1:-1] = 0
    assert float(ring.abs().max()) == 0.0, "the halo ring must stay zero"
    dx = out.to_nchw().cpu()
    close(dw.cpu().numpy(), ref_dw.numpy(), 1e-5, "dW vs fp64")
    close(db.cpu().numpy(), ref_db.numpy(), 1e-5, "db vs fp64")
    # one bf16 rounding of the exact value (2^-9 relative per element; measured against max|ref| like every other dgrad test)
    err = (dx - ref_dx).abs()
    assert float((err - ref_dx.abs() * 2.0 ** -8).max()) <= 1e-6 * float(ref_dx.abs().max()), float(err.max())
    # ---- the three launches it replaces
    gx2 = T.BT.alloc(n, 16, h, w, torch.bfloat16, DEV, halo=1, zero=True)
    dw2, db2 = torch.zeros_like(dw), torch.zeros_like(db)
    T.image_out_wgrad(xb, gimg, yimg, dw2, db2, 16, 3, ws)
    T.image_out_dgrad(gimg, yimg, wt, xb, gx2, 16, 3, 3, 0)
    dx2 = gx2.fold_halo_().to_nchw().cpu()
    torch.cuda.synchronize()
    close(dw.cpu().numpy(), dw2.cpu().numpy(), 2e-5, "dW vs image_out_wgrad")
    close(db.cpu().numpy(), db2.cpu().numpy(), 2e-5, "db vs image_out_wgrad")
    close(dx.numpy(), dx2.numpy(), 1.2e-2, "dx vs dgrad + fold")
    # away from the fold targets (rows / columns 1 and h-2 / w-2) the two are the same arithmetic rounded once
    if h > 6 and w > 6:
        a, b = dx[:, :, 3:-3, 3:-3], dx2[:, :, 3:-3, 3:-3]
        assert float((a - b).abs().max()) <= 2.0 ** -7 * float(b.abs().max())
    # accumulate = True adds onto dW / db
    dw3, db3 = dw.clone(), db.clone()
    gx3 = T.BT.alloc(n, 16, h, w, torch.bfloat16, DEV, halo=1, zero=True)
    T.image_out_bwd(xb, gimg, yimg, wt, gx3, dw3, db3, 16, 3, ws, accumulate=True)
    torch.cuda.synchronize()
    close(dw3.cpu().numpy(), 2 * dw.cpu().numpy(), 1e-6, "accumulate")
    assert torch.equal(gx3.buf, gx.buf), "the launch is deterministic"


def test_engine_step_with_and_without_the_fused_image_layer_backward(monkeypatch):
    """PFNetv1 and DenseFuse train steps (bf16) with $MMIF_IMAGE_BWD = 1 / 0: same losses, gradients within the bf16 path's own noise"""
    import core.model as M
    from core.loss import FusionLoss, GradLoss, PixelLoss, SSIMLoss, unit_gradient
    from gpu_util import dtype_ctx, reload_switches
    g = torch.Generator().manual_seed(3)
    a, b = torch.rand(2, 1, 72, 104, generator=g).to(DEV), torch.rand(2, 1, 72, 104, generator=g).to(DEV)
    for name in ("PFNetv1", "DenseFuse"):
        res = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("MMIF_IMAGE_BWD", mode)
            reload_switches()
            with dtype_ctx("bf16"):
                torch.manual_seed(0)
                m = getattr(M, name)().to(DEV)
                l_all = FusionLoss(SSIMLoss('ssim', weight=1.0), PixelLoss('l1', weight=0.01), GradLoss('l1', weight=0.1).to(DEV), 'max', 'max')
                f = m(a, b)
                tot = l_all(a, b, f)
                tot.backward(unit_gradient(tot))
                torch.cuda.synchronize()
                res[mode] = (float(tot), {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()})
        monkeypatch.delenv("MMIF_IMAGE_BWD")
        reload_switches()
        assert res["1"][0] == res["0"][0]
        for k, g1 in res["1"][1].items():
            g0 = res["0"][1][k]
            assert float(g0.abs().max()) > 0
            rel = float((g1 - g0).norm() / g0.norm())
            assert rel <= 2e-2, (name, k, rel)
