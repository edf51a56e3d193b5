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
