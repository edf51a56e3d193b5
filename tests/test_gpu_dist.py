"""Data parallel on the device: two processes (both on cuda:0, gloo transport -- RCCL refuses two ranks on one GPU) run
the REAL train step (HIP engine + losses + FusedClipAdam with its [gradients | loss scalars] all-reduce: one collective on the first
step, the early decoder bucket + late encoder bucket of mmif/dist.py afterwards) on the two halves of a batch; parameters and reduced
loss scalars after two steps must match one process on the whole batch.
The 8-GPU RCCL run itself is the driver's; this pins the code path it takes."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPE = (4, 1, 48, 40)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, out, model_name="PFNetv1"):
    for p in (ROOT, os.path.join(ROOT, "multi-modal-image-fusion_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import core.model as M
    from core.loss import GradLoss, PixelLoss, SSIMLoss
    from mmif import engine as E
    from mmif.dist import broadcast_parameters, shard_batch
    from mmif.optim import FusedClipAdam
    from oracle import fusion_oracle as O
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    E.set_compute_dtype("fp32")
    torch.manual_seed(10 + rank)                      # different initial weights per rank: the broadcast must fix that
    model = getattr(M, model_name)().to(dev)
    if world > 1:
        broadcast_parameters(model, 0)
    else:
        torch.manual_seed(10)
        model = getattr(M, model_name)().to(dev)
    model.train()
    opt = FusedClipAdam(model.parameters(), lr=1e-3, betas=(0.9, 0.999), max_norm=5.0)
    l1, l2, l3 = SSIMLoss('ssim', weight=1.0), PixelLoss('l1', weight=0.01), GradLoss('l1', weight=0.1).to(dev)
    from core.loss import FusionLoss
    fused = FusionLoss(l1, l2, l3, 'max', 'max')
    lo, hi = shard_batch(SHAPE[0], rank, world)
    scal = None
    for step in range(2):      # step 0: one all-reduce; step 1: the two-bucket early reduce (armed by the first distributed step)
        i1 = torch.from_numpy(O.closed_form_image(SHAPE, 0.3 + step)).to(dev)[lo:hi].contiguous()
        i2 = torch.from_numpy(O.closed_form_image(SHAPE, 1.7 + step)).to(dev)[lo:hi].contiguous()
        opt.zero_grad(set_to_none=True)
        f = model(i1, i2)
        # bench.py's / train.py's path: one fused loss call, its ready 4-vector as the scalars (_run_seq below passes a list of 0-dim
        # tensors).  Both world sizes take the same path: Adam's first steps turn a last-bit difference of a near-zero gradient into a
        # visible parameter difference, so the comparison wants identical arithmetic up to the batch split
        tot = fused(i1, i2, f)
        vals = fused.values
        opt.stage_scalars(vals)         # (a no-op on step 0: the flat buffer is not known yet)
        tot.backward()
        opt.step(scalars=vals)
        scal = opt.reduced_scalars.detach().cpu().numpy()
    torch.cuda.synchronize()
    # a conv bias in front of a BatchNorm has an exactly-zero true gradient (the norm removes the mean): what the kernels produce for
    # it is rounding noise, which Adam's g / sqrt(v) turns into +-lr steps -- such parameters cannot be compared between runs
    import torch.nn as nn
    from core.block import ConvLayer
    noise = [f"{n}.layers.0.bias" for n, mod in model.named_modules()
             if isinstance(mod, ConvLayer) and mod.norm is nn.BatchNorm2d and mod.layers[0].bias is not None]
    from mmif import dist as D
    out[(model_name, world, rank)] = dict(noise=noise, P={k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}, scal=scal,
                                          early=D.early_reduce_count())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_two_rank_step_equals_single_process_full_batch():
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_run, args=(2, port, out), nprocs=2, join=True)
        mp.spawn(_run, args=(1, port, out), nprocs=1, join=True)
        res = {k: dict(v) for k, v in out.items()}
    single, r0, r1 = res[("PFNetv1", 1, 0)], res[("PFNetv1", 2, 0)], res[("PFNetv1", 2, 1)]
    for k in single["P"]:
        assert np.array_equal(r0["P"][k], r1["P"][k]), f"ranks diverged on {k}"
        ref = single["P"][k]
        assert np.abs(r0["P"][k] - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), k   # (identical with $MMIF_EARLY_REDUCE=0)
    assert np.allclose(r0["scal"], r1["scal"]) and np.allclose(r0["scal"], single["scal"], rtol=2e-5, atol=1e-6)
    want = 0 if os.environ.get("MMIF_EARLY_REDUCE", "1") == "0" else 1
    assert r0["early"] == want and r1["early"] == want and single["early"] == 0, "step 1 must take the two-bucket path"


def _run_seq(rank, world, port, out, backend, key):
    """eager step -> backward WITHOUT a step (a skipped iteration) -> eager step -> GraphedStep (its warm-up backwards have no step
    either) -> two graphed steps; with a process group the early / late gradient buckets of mmif/dist.py are live throughout"""
    for p in (ROOT, os.path.join(ROOT, "multi-modal-image-fusion_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import core.model as M
    from core.loss import GradLoss, PixelLoss, SSIMLoss
    from mmif import dist as D
    from mmif import engine as E
    from mmif.dist import broadcast_parameters, shard_batch
    from mmif.graph import GraphedStep
    from mmif.optim import FusedClipAdam
    from oracle import fusion_oracle as O
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if backend is not None:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    E.set_compute_dtype("fp32")
    torch.manual_seed(10)
    model = M.PFNetv1().to(dev)
    if backend is not None:
        broadcast_parameters(model, 0)
    opt = FusedClipAdam(model.parameters(), lr=1e-3, betas=(0.9, 0.999), max_norm=5.0)
    l1, l2, l3 = SSIMLoss('ssim', weight=1.0), PixelLoss('l1', weight=0.01), GradLoss('l1', weight=0.1).to(dev)
    lo, hi = shard_batch(SHAPE[0], rank, world)

    def batch(step):
        return tuple(torch.from_numpy(O.closed_form_image(SHAPE, ph + step)).to(dev)[lo:hi].contiguous() for ph in (0.3, 1.7))

    def losses(i1, i2, f):
        a, b, c = l1(i1, i2, f), l2(i1, i2, f, mode='max'), l3(i1, i2, f, mode='max')
        return a + b + c, a, b, c

    def eager(step, do_step=True):
        i1, i2 = batch(step)
        opt.zero_grad(set_to_none=True)
        outs = losses(i1, i2, model(i1, i2))
        opt.stage_scalars(list(outs))
        outs[0].backward()
        if do_step:
            opt.step(scalars=list(outs))
    eager(0)
    eager(1, do_step=False)        # backward with no step: its early all-reduce (if any) must not leak into the next step
    eager(2)
    i1, i2 = batch(3)
    g = GraphedStep(model, losses, opt, i1, i2)
    g(i1, i2)
    g(*batch(4))
    eager(5)                       # and an eager step after the graph: the early path re-arms
    torch.cuda.synchronize()
    out[key + (rank,)] = dict(P={k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}, scal=opt.reduced_scalars.detach().cpu().numpy(),
                              early=D.early_reduce_count(), pending=D.pending_early())
    if backend is not None:
        dist.barrier()
        dist.destroy_process_group()


def test_two_rank_skipped_step_and_graphed_steps_equal_single_process():
    """a backward without a matching step() -- a skipped iteration, GraphedStep's warm-up -- used to leave an early
    all-reduce handle behind that the next step() tripped over or trusted for the wrong gradients.  Two ranks through eager + skipped +
    graphed steps == one process on the full batch; nothing left pending."""
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_run_seq, args=(2, port, out, "gloo", ("seq", 2)), nprocs=2, join=True)
        mp.spawn(_run_seq, args=(1, port, out, None, ("seq", 1)), nprocs=1, join=True)
        res = {k: dict(v) for k, v in out.items()}
    single, r0, r1 = res[("seq", 1, 0)], res[("seq", 2, 0)], res[("seq", 2, 1)]
    for k in single["P"]:
        assert np.array_equal(r0["P"][k], r1["P"][k]), f"ranks diverged on {k}"
        ref = single["P"][k]
        assert np.abs(r0["P"][k] - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max()), k
    assert np.allclose(r0["scal"], r1["scal"]) and np.allclose(r0["scal"], single["scal"], rtol=5e-5, atol=1e-6)
    assert r0["pending"] == r1["pending"] == 0
    if os.environ.get("MMIF_EARLY_REDUCE", "1") != "0":
        assert r0["early"] >= 2 and r1["early"] >= 2, "the eager steps after the first must take the two-bucket path"


@pytest.mark.parametrize("early", ["1", "0"])
def test_single_rank_nccl_path(early):
    """The RCCL code path itself (every other multi-rank test runs on gloo: RCCL refuses two ranks on one GPU): world_size 1, backend
    'nccl', a fresh child process: init -> one flat parameter broadcast -> eager / skipped / graphed steps with the asynchronous decoder
    bucket on RCCL's stream while the encoder backward runs on the compute stream, the encoder bucket + loss scalars in step().  With one
    rank every collective is the identity, so parameters and reduced scalars must equal the no-process-group run BIT FOR BIT -- with the
    two-bucket path and with $MMIF_EARLY_REDUCE=0 (a wrong stream order or a stale scratch copy shows up as a different bit)."""
    port = _free_port()
    prev = os.environ.get("MMIF_EARLY_REDUCE")
    os.environ["MMIF_EARLY_REDUCE"] = early
    try:
        with mp.Manager() as mgr:
            out = mgr.dict()
            mp.spawn(_run_seq, args=(1, port, out, "nccl", ("nccl", 1)), nprocs=1, join=True)
            mp.spawn(_run_seq, args=(1, port, out, None, ("plain", 1)), nprocs=1, join=True)
            res = {k: dict(v) for k, v in out.items()}
    finally:
        if prev is None:
            os.environ.pop("MMIF_EARLY_REDUCE", None)
        else:
            os.environ["MMIF_EARLY_REDUCE"] = prev
    a, b = res[("nccl", 1, 0)], res[("plain", 1, 0)]
    for k in b["P"]:
        assert np.array_equal(a["P"][k], b["P"][k]), f"RCCL path differs from the plain path on {k}"
    assert np.array_equal(a["scal"], b["scal"])
    assert a["pending"] == 0 and b["early"] == 0
    assert (a["early"] >= 2) if early == "1" else (a["early"] == 0)


def test_two_rank_batchnorm_model_equals_single_process_full_batch():
    """A BatchNorm net (DIFNet, reference core/model.py) under data parallel: the reference converts to nn.SyncBatchNorm
    (train.py:296), i.e. statistics over the GLOBAL batch -- the two-rank run must reproduce the one-process full-batch run, running
    buffers included, and both ranks must hold identical buffers (only rank 0's reach the checkpoint)."""
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_run, args=(2, port, out, "DIFNet"), nprocs=2, join=True)
        mp.spawn(_run, args=(1, port, out, "DIFNet"), nprocs=1, join=True)
        res = {k: dict(v) for k, v in out.items()}
    single, r0, r1 = res[("DIFNet", 1, 0)], res[("DIFNet", 2, 0)], res[("DIFNet", 2, 1)]
    assert any("running_mean" in k for k in single["P"]) and len(single["noise"]) >= 5
    for k in single["P"]:
        assert np.array_equal(r0["P"][k], r1["P"][k]), f"ranks diverged on {k}"
        if k in single["noise"]:
            continue
        ref = single["P"][k]
        assert np.abs(r0["P"][k].astype(np.float64) - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), k
    assert np.allclose(r0["scal"], r1["scal"]) and np.allclose(r0["scal"], single["scal"], rtol=1e-4, atol=1e-6)


def test_staged_batchnorm_calls_equal_the_fused_ones():
    """mmif_bn_moments -> mmif_bn_apply_fwd and mmif_bn_bwd_sums -> mmif_bn_apply_bwd without an all-reduce in between == kind 0 of
    mmif_norm_act_fwd / _bwd (include/mmif.h), bit for bit, running buffers included."""
    import sys
    for p in (ROOT, os.path.join(ROOT, "multi-modal-image-fusion_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from mmif import tensor as T
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(3, 5, 9, 7, generator=g) * 1.7 + 0.4).to(dev)
    gy = torch.randn(3, 5, 9, 7, generator=g).to(dev)
    gamma, beta = torch.randn(5, generator=g).to(dev), torch.randn(5, generator=g).to(dev)
    for act in (0, 1, 2, 3):
        rm1, rv1 = torch.zeros(5, device=dev), torch.ones(5, device=dev)
        rm2, rv2 = torch.zeros(5, device=dev), torch.ones(5, device=dev)
        y1, st1 = T.norm_act_fwd(x, gamma, beta, rm1, rv1, T.NORM_BN_TRAIN, 1e-5, 0.1, act)
        chan = T.bn_moments(x)
        assert float(chan[-1]) == 3 * 9 * 7
        y2, st2 = T.bn_apply_fwd(x, chan, gamma, beta, rm2, rv2, 1e-5, 0.1, act)
        assert torch.equal(y1, y2) and torch.equal(st1, st2) and torch.equal(rm1, rm2) and torch.equal(rv1, rv2)
        dx1, dg1, db1 = T.norm_act_bwd(x, y1, gy, st1, gamma, T.NORM_BN_TRAIN, act)
        ch2, dg2, db2 = T.bn_bwd_sums(x, y2, gy, st2, act)
        dx2 = T.bn_apply_bwd(x, y2, gy, st2, gamma, ch2, chan[-1:], act)
        assert torch.equal(dx1, dx2) and torch.equal(dg1, dg2) and torch.equal(db1, db2)
    # against torch's own BatchNorm2d (CPU, fp64)
    bn = torch.nn.BatchNorm2d(5).double()
    with torch.no_grad():
        bn.weight.copy_(gamma.cpu())
        bn.bias.copy_(beta.cpu())
    xf = x.cpu().double().requires_grad_(True)
    yr = torch.relu(bn(xf))
    (yr * gy.cpu().double()).sum().backward()
    y2, st2 = T.bn_apply_fwd(x, T.bn_moments(x), gamma, beta, None, None, 1e-5, 0.1, 1)
    ch2, dg2, db2 = T.bn_bwd_sums(x, y2, gy, st2, 1)
    dx2 = T.bn_apply_bwd(x, y2, gy, st2, gamma, ch2, chan[-1:], 1)
    assert (y2.cpu().double() - yr).abs().max() < 1e-5 and (dx2.cpu().double() - xf.grad).abs().max() < 1e-4
    assert (dg2.cpu().double() - bn.weight.grad).abs().max() < 1e-3 and (db2.cpu().double() - bn.bias.grad).abs().max() < 1e-3
