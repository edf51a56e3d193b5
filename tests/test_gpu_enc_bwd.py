"""The fused encoder backward (csrc/enc_bwd.hip, mmif_dense_encoder_bwd; reference: autograd of core/model.py:73-80 + core/block.py:137-151
as train.py:71 runs it): the DenseBlock's gradient chain AND dW, db of all four encoder layers in ONE streaming launch -- g0..g2 never
reach HBM.

* against the two kernels it replaces on identical inputs (mmif_dense_encoder_chain -> mmif_dense_encoder_wgrad; each of those is held to
  its fp64 definition / the oracle by tests/test_gpu_enc_chain.py and tests/test_gpu_enc_wgrad.py): same rounding points (g0..g2 are
  bf16 in both), another summation order -- every dW / db within 2e-3 of its maximum;
* against the fp64 DEFINITION directly: g2, g1, g0 by the oracle's conv backward on the bf16 operands (rounded to bf16 per stage, as both
  kernels store them), then dW_L, db_L = the oracle's weight gradients of conv L in fp64 -- within 4e-3 (one bf16 rounding of a g value
  that sits on a rounding boundary may differ between the two);
* shapes: one strip, several strips with the clamped right edge strip, several row segments, h = w = 4, two branches in one launch,
  accumulation onto existing gradients (shared encoders), the gradient of x3 in one tensor and G0..G2 in another (DenseFuse)."""
import numpy as np
import pytest
import torch

from oracle import fusion_oracle as O
from gpu_util import bf16_round, close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [(2, 37, 53), (1, 16, 16), (1, 4, 4), (2, 5, 31), (1, 9, 30), (3, 64, 29), (1, 33, 32), (1, 20, 57), (2, 40, 256), (1, 300, 64), (1, 7, 4), (2, 128, 130),
          # one block per CU walks a slice of the (image, strip) rows in pieces: many short columns per slice (h small, n large), a slice that
          # starts and ends inside a column, the reference's test frame (1024 x 1224: 47 strips), widths whose right fold crosses a 16-lane row
          (32, 8, 40), (5, 64, 64), (1, 515, 1224), (3, 100, 15), (2, 61, 16), (7, 19, 83)]


def _setup(n, h, w, seed):
    from mmif import tensor as T
    g = torch.Generator().manual_seed(seed)
    xs = torch.randn(n, 64, h, w, generator=g)
    xs[xs.abs() < 0.5] = 0.0                                   # ReLU-style activations: zeros and negatives mask the gradient
    xs = xs.abs() * (torch.rand(n, 64, h, w, generator=g) > 0.3)
    G = torch.randn(n, 64, h, w, generator=g)
    ws = [torch.randn(16, 16 * (i + 1), 3, 3, generator=g) * (0.25 / (i + 1)) for i in range(3)]
    img = torch.rand(n, 1, h, w, generator=g)
    F = T.BT.from_nchw(xs.to(DEV), torch.bfloat16)
    GF = T.BT.from_nchw(G.to(DEV), torch.bfloat16, halo=1).as_folded()
    pk = T.pack_dense_chain(*[t.to(DEV) for t in ws], DEV)
    return xs, G, ws, img.to(DEV), F, GF, pk


def _grads(val=None):
    shapes = [((16, 1, 3, 3), (16,)), ((16, 16, 3, 3), (16,)), ((16, 32, 3, 3), (16,)), ((16, 48, 3, 3), (16,))]
    mk = (lambda s: torch.zeros(s, device=DEV)) if val is None else (lambda s: torch.full(s, val, device=DEV))
    return [(mk(a), mk(b)) for a, b in shapes]


def _two_kernels(T, img, F, GF, pk, grads, accumulate=False, g3=None, glow=None):
    n, h, w = F.n, F.h, F.w
    out = T.BT.alloc(n, 64, h, w, torch.bfloat16, DEV)
    T.dense_encoder_chain([(g3 if g3 is not None else GF.view(6, 2), glow if glow is not None else GF.view(0, 6), F.view(0, 6), pk, out)])
    ws = torch.empty(T.dense_encoder_wgrad_workspace_bytes() // 4 + 1, dtype=torch.float32, device=DEV)
    T.dense_encoder_wgrad(img, F.view(0, 6), out, grads, ws, accumulate)
    return out


@pytest.mark.parametrize("n,h,w", SHAPES, ids=[f"{n}x{h}x{w}" for n, h, w in SHAPES])
def test_fused_backward_vs_chain_plus_wgrad_kernels(n, h, w):
    from mmif import tensor as T
    xs, G, ws, img, F, GF, pk = _setup(n, h, w, 100 * h + w)
    ref, got = _grads(), _grads(7.0)          # (the fused call must overwrite, not add onto, what the buffers hold)
    _two_kernels(T, img, F, GF, pk, ref)
    wsf = torch.empty(T.dense_encoder_bwd_workspace_bytes() // 4 + 1, dtype=torch.float32, device=DEV)
    T.dense_encoder_bwd([(GF.view(6, 2), GF.view(0, 6), F.view(0, 6), pk, img, got, False)], wsf)
    torch.cuda.synchronize()
    for L, ((dw, db), (rw, rb)) in enumerate(zip(got, ref)):
        close(dw.cpu().numpy(), rw.cpu().numpy(), 2e-3, f"dW{L}")
        close(db.cpu().numpy(), rb.cpu().numpy(), 2e-3, f"db{L}")


@pytest.mark.parametrize("n,h,w", [(2, 21, 45), (1, 12, 70), (1, 4, 4)], ids=["2x21x45", "1x12x70", "1x4x4"])
def test_fused_backward_vs_fp64_definition(n, h, w):
    from mmif import tensor as T
    xs, G, ws, img, F, GF, pk = _setup(n, h, w, 31 * h + w)
    got = _grads()
    wsf = torch.empty(T.dense_encoder_bwd_workspace_bytes() // 4 + 1, dtype=torch.float32, device=DEV)
    T.dense_encoder_bwd([(GF.view(6, 2), GF.view(0, 6), F.view(0, 6), pk, img, got, False)], wsf)
    torch.cuda.synchronize()
    xq = bf16_round(xs.numpy()).astype(np.float64)
    Gq = bf16_round(G.numpy()).astype(np.float64)
    wq = [bf16_round(t.numpy()).astype(np.float64) for t in ws]
    g = {3: Gq[:, 48:64]}
    for k in (2, 1, 0):                               # the chain, stage by stage, each stage rounded to bf16 as both kernel forms store it
        acc = Gq[:, 16 * k:16 * k + 16].copy()
        for l in range(k + 1, 4):
            gx, _, _ = O.conv2d_reflect_bwd(xq[:, :16 * l], wq[l - 1], None, g[l], relu=False, need_gx=True)
            acc += gx[:, 16 * k:16 * k + 16]
        g[k] = bf16_round((acc * (xq[:, 16 * k:16 * k + 16] > 0)).astype(np.float32)).astype(np.float64)
    imq = img.cpu().numpy().astype(np.float64)
    for L in range(4):                                # conv L: input = image (L = 0) or [x0 .. x(L-1)], output gradient g_L
        xin = imq if L == 0 else xq[:, :16 * L]
        w_ = np.zeros((16, xin.shape[1], 3, 3))
        _, gw, gb = O.conv2d_reflect_bwd(xin, w_, None, g[L], relu=False, need_gx=False)
        close(got[L][0].cpu().numpy(), gw, 4e-3, f"dW{L} vs fp64")
        close(got[L][1].cpu().numpy(), gb, 4e-3, f"db{L} vs fp64")


def test_fused_backward_two_branches_accumulate_and_split_gradient_tensors():
    """PFNetv1: two branches with their own weights in one launch; shared encoders (DenseFuse / VIFNet): the second branch accumulates onto
    the first's gradients; DenseFuse: G0..G2 from the ONE gradient of f1 + f2, g3 from the branch's own buffer."""
    from mmif import tensor as T
    n, h, w = 2, 45, 70
    xa, Ga, wa, ia, Fa, GFa, pka = _setup(n, h, w, 7)
    xb, Gb, wb, ib, Fb, GFb, pkb = _setup(n, h, w, 8)
    wsf = torch.empty(T.dense_encoder_bwd_workspace_bytes() // 4 + 1, dtype=torch.float32, device=DEV)
    # (1) two independent branches in one launch = two single launches
    ra, rb, ga, gb = _grads(), _grads(), _grads(), _grads()
    T.dense_encoder_bwd([(GFa.view(6, 2), GFa.view(0, 6), Fa.view(0, 6), pka, ia, ra, False)], wsf)
    T.dense_encoder_bwd([(GFb.view(6, 2), GFb.view(0, 6), Fb.view(0, 6), pkb, ib, rb, False)], wsf)
    T.dense_encoder_bwd([(GFa.view(6, 2), GFa.view(0, 6), Fa.view(0, 6), pka, ia, ga, False), (GFb.view(6, 2), GFb.view(0, 6), Fb.view(0, 6), pkb, ib, gb, False)], wsf)
    torch.cuda.synchronize()
    for (a, b), (c, d) in zip(ra + rb, ga + gb):
        assert torch.equal(a, c) and torch.equal(b, d)
    # (2) shared weights, glow from ANOTHER tensor, second branch accumulating: against the two-kernel path driven the same way
    ref, got = _grads(), _grads()
    _two_kernels(T, ia, Fa, GFa, pka, ref, False, g3=GFa.view(6, 2), glow=GFa.view(0, 6))
    _two_kernels(T, ib, Fb, GFb, pka, ref, True, g3=GFb.view(6, 2), glow=GFa.view(0, 6))
    T.dense_encoder_bwd([(GFa.view(6, 2), GFa.view(0, 6), Fa.view(0, 6), pka, ia, got, False), (GFb.view(6, 2), GFa.view(0, 6), Fb.view(0, 6), pka, ib, got, True)], wsf)
    torch.cuda.synchronize()
    for L, ((dw, db), (rw, rb_)) in enumerate(zip(got, ref)):
        close(dw.cpu().numpy(), rw.cpu().numpy(), 2e-3, f"dW{L}")
        close(db.cpu().numpy(), rb_.cpu().numpy(), 2e-3, f"db{L}")


@pytest.mark.parametrize("name,single", [("PFNetv1", False), ("VIFNet", False), ("DenseFuse", False), ("DenseFuse", True), ("PFNetv2", False)])
def test_models_with_the_fused_encoder_backward_match_the_two_kernel_path(name, single):
    """every engine that owns a DenseBlock encoder, one backward with the fused kernel ($MMIF_ENC_BWD_FUSED, the default) and with the
    chain + weight-gradient launches it replaces: identical fused image (the forward is the same), every parameter gradient within the
    summation-order noise of the encoder's weight gradients (the decoder's are bit-identical: nothing upstream of them changed)"""
    import os
    import core.model as M
    from mmif import engine as E
    from mmif import tensor as T
    from gpu_util import dtype_ctx
    with dtype_ctx("bf16"):
        torch.manual_seed(7)
        m = getattr(M, name)().to(DEV)
        g = torch.Generator().manual_seed(3)
        i1, i2 = torch.rand(2, 1, 40, 56, generator=g).to(DEV), torch.rand(2, 1, 40, 56, generator=g).to(DEV)
        res = []
        for on in ("0", "1"):
            os.environ["MMIF_ENC_BWD_FUSED"] = on
            E.reload_switches()
            try:
                T.PROFILE_TAGS.add("encode:bwd")
                T.PROFILE_EVENTS.pop("encode:bwd", None)
                m.zero_grad(set_to_none=True)
                y = m(i1) if single else m(i1, i2)
                (y * y).mean().backward()
                torch.cuda.synchronize()
                res.append((y.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}, len(T.PROFILE_EVENTS.get("encode:bwd", []))))
            finally:
                os.environ.pop("MMIF_ENC_BWD_FUSED")
                E.reload_switches()
                T.PROFILE_TAGS.discard("encode:bwd")
        assert res[0][2] == 0 and res[1][2] == 1, "the fused launch must run exactly when enabled"
        assert torch.equal(res[0][0], res[1][0])
        for k in res[0][1]:
            a, b = res[0][1][k].double(), res[1][1][k].double()
            assert float(a.abs().max()) > 0, k
            if "encode" in k:
                assert float((a - b).abs().max()) <= 2e-3 * float(a.abs().max()), (name, single, k, float((a - b).abs().max()) / float(a.abs().max()))
            else:
                assert torch.equal(a, b), (name, single, k)


def test_fused_backward_argument_validation():
    import ctypes as C
    from mmif import _lib
    e = _lib.MmifDenseChain()
    assert _lib.lib.mmif_dense_encoder_bwd(C.byref(e), None, None, 0, None, None, None, 0, None, 0, None) != 0
    assert b"dense_encoder_bwd" in _lib.lib.mmif_last_error()
