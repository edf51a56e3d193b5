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
