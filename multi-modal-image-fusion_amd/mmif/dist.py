"""Data-parallel plumbing (reference common.py:96-113, train.py:151-158,285-297): one process per
GPU, RCCL ('nccl' backend on ROCm) over xGMI; parameters are synchronised with ONE broadcast of the
flat parameter buffer instead of the reference's temp-file + DDP-constructor dance."""
import os

import torch
import torch.distributed as dist


def setup_dist(rank=None, world_size=None, backend=None):
    """env:// rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if rank is not None:
        os.environ["RANK"] = str(rank)
    if world_size is not None:
        os.environ["WORLD_SIZE"] = str(world_size)
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        dist.init_process_group(backend)
    return dist.get_rank(), dist.get_world_size()


def shard_batch(n, rank, world):
    """Rank r takes [r*n/world, (r+1)*n/world) -- per-rank batch = bs // world (train.py:209)."""
    per = n // world
    return rank * per, (rank + 1) * per


@torch.no_grad()
def broadcast_parameters(module, src=0):
    """ONE broadcast of all parameters AND floating-point buffers (BatchNorm running statistics) from `src`, flattened -- what the
    reference gets from init_weights.pth + the DDP constructor (train.py:285-297); integer buffers (num_batches_tracked) follow in a
    second, tiny broadcast."""
    ts = [p.data for p in module.parameters()] + [b.data for b in module.buffers() if b.is_floating_point()]
    if ts:
        flat = torch.cat([t.reshape(-1).float() for t in ts])
        dist.broadcast(flat, src)
        off = 0
        for t in ts:
            n = t.numel()
            t.copy_(flat[off:off + n].view(t.shape))
            off += n
    ints = [b.data for b in module.buffers() if not b.is_floating_point()]
    if ints:
        flat = torch.cat([t.reshape(-1).to(torch.int64) for t in ints])
        dist.broadcast(flat, src)
        off = 0
        for t in ints:
            n = t.numel()
            t.copy_(flat[off:off + n].view(t.shape))
            off += n


# ---- cross-rank BatchNorm -------------------------------------------------------------------------------------------------------
_SYNC_BN = [True]


def set_sync_batchnorm(enabled):
    """The reference converts every BatchNorm to nn.SyncBatchNorm when it trains distributed (train.py:296); this engine does the same
    by default: in training mode, with an initialised process group of more than one rank, BatchNorm statistics (forward) and their
    gradient sums (backward) are all-reduced.  set_sync_batchnorm(False) restores per-rank statistics."""
    _SYNC_BN[0] = bool(enabled)


def sync_bn_active():
    return _SYNC_BN[0] and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def allreduce_sums(t):
    """SUM all-reduce of a small fp64 statistics buffer, in place (RCCL on the current stream; gloo in the CPU tests)."""
    dist.all_reduce(t)
    return t


@torch.no_grad()
def allreduce_flat(flat, scalars=None):
    """SUM all-reduce of [gradients | scalars]; returns (flat, mean scalars).  `flat` must have
    len(scalars) spare slots at its end (mmif.engine.GRAD_TAIL)."""
    world = dist.get_world_size()
    k = len(scalars) if scalars else 0
    if k:
        flat[-k:] = torch.stack([s.detach().float().reshape(()) for s in scalars])
    dist.all_reduce(flat)
    return flat, (flat[-k:] / world if k else None)


# ---- gradient all-reduce in two buckets: the decoder's gradients leave while the encoder's backward still runs -----------------------
# A fusion net's backward finishes its decoder first (PFNetv1: 98 % of the 272 k parameters) and then spends ~1 ms of a 4 ms step in
# the encoder chains.  With one process per GPU the all-reduce of the flat gradient buffer was the only thing on the device between the
# last backward kernel and clip + Adam (RCCL launch + two cross-stream hand-offs: 85 us even with ONE rank).  Protocol:
#   optimizer   : FusedClipAdam.stage_scalars([...]) before backward (optional) parks the loss values for the tail slots of the buffer;
#   engine      : every backward starts with begin_backward() (a new generation); after the decoder's gradients are written,
#                 early_allreduce(flat, lo, hi) COPIES flat[lo:hi] (+ the staged scalars) into a scratch buffer and starts an ASYNC
#                 all-reduce of the copy on RCCL's stream; the compute stream goes on with the encoder backward;
#   optimizer   : step() -> take_early(flat): when the handle belongs to the LATEST backward, it is waited for and the reduced copy is
#                 written back over flat[lo:hi]; step() then all-reduces only flat[0:lo] (encoder gradients).  Otherwise -- no handle, or
#                 a handle of an earlier backward (a skipped optimizer step, gradient accumulation, hipGraph warm-up backwards) -- the
#                 whole buffer is reduced as before.
# The early collective is SPECULATIVE: it never modifies the gradient buffer until step() accepts it, so a backward without a matching
# step() cannot leave half-reduced gradients behind, and a stale handle can only be discarded (round 2 reduced in place: a backward
# with no step() left a handle that the next step() either tripped over or trusted for the wrong gradients).  (Gradients edited by
# hand between backward and step() would be overwritten for the decoder range: use $MMIF_EARLY_REDUCE=0 for such a flow.)
# Off ($MMIF_EARLY_REDUCE=0, or never armed) nothing changes.  Only armed by FusedClipAdam when a process group with a collective
# backend is up; never inside a hipGraph capture; only for a backward that finds every .grad None.
_EARLY = {"armed": False, "pending": {}, "tail": None, "count": 0, "gen": 0, "scratch": {}}


def arm_early_reduce(on=True):
    """armed by the optimizer after a step that consumed the engine's flat gradient buffer zero-copy"""
    _EARLY["armed"] = bool(on) and os.environ.get("MMIF_EARLY_REDUCE", "1") != "0"


def begin_backward():
    """every engine backward: a new generation -- handles of earlier backwards are stale from here on (their reduced copies are
    simply dropped by take_early / the next early_allreduce)"""
    _EARLY["gen"] += 1
    return _EARLY["gen"]


def pending_early():
    return len(_EARLY["pending"])


def stage_tail(values):
    """the optimizer parks this step's loss scalars (a small 1-D device tensor) here before backward; the early all-reduce of that
    backward carries them; step() clears whatever is still parked"""
    _EARLY["tail"] = values


def staged_tail():
    return _EARLY["tail"]


def early_reduce_count():
    """how many early all-reduces this process has started (tests)"""
    return _EARLY["count"]


def early_reduce_armed():
    return _EARLY["armed"] and dist.is_available() and dist.is_initialized()


@torch.no_grad()
def early_allreduce(flat, lo, hi, tail_at=None):
    """async SUM all-reduce of a COPY of flat[lo:hi]; the handle is kept until take_early(flat).  tail_at: index of the buffer's scalar
    slots when hi reaches them -- staged scalars (stage_tail) ride behind the gradients and the accepted range grows to cover them."""
    drain_early()                                      # handles of earlier backwards (never consumed): wait and drop, ALL buffers
    if not early_reduce_armed() or hi <= lo:
        return False
    if flat.is_cuda and torch.cuda.is_current_stream_capturing():
        return False
    vals = _EARLY["tail"]
    _EARLY["tail"] = None
    ntail = 0
    if tail_at is not None and hi == tail_at and vals is not None and vals.numel() <= flat.numel() - tail_at:
        ntail = vals.numel()
    n = hi - lo
    key = flat.data_ptr()
    scratch = _EARLY["scratch"].get(key)
    if scratch is None or scratch.numel() < n + ntail or scratch.device != flat.device:
        scratch = _EARLY["scratch"][key] = torch.empty(n + 8, dtype=flat.dtype, device=flat.device)
    scratch[:n].copy_(flat[lo:hi])
    if ntail:
        scratch[n:n + ntail].copy_(vals)
    work = dist.all_reduce(scratch[:n + ntail], async_op=True)
    _EARLY["count"] += 1
    _EARLY["pending"][key] = (work, lo, hi, ntail, _EARLY["gen"], scratch)
    return True


def take_early(flat):
    """-> (lo, hi) of the range of `flat` that now holds all-reduced values (hi includes the scalar slots that rode along), or None:
    no handle for this buffer, or its handle is from an earlier backward than the latest one (the buffer may have changed since)."""
    ent = _EARLY["pending"].pop(flat.data_ptr(), None)
    if ent is None:
        return None
    work, lo, hi, ntail, gen, scratch = ent
    work.wait()                                        # (orders the current stream after the collective)
    if gen != _EARLY["gen"]:
        return None
    n = hi - lo
    flat[lo:hi + ntail].copy_(scratch[:n + ntail])
    return lo, hi + ntail


def drain_early(flat=None):
    """wait for and DROP pending handles (their reduced copies are never written back)"""
    keys = [flat.data_ptr()] if flat is not None else list(_EARLY["pending"].keys())
    for k in keys:
        ent = _EARLY["pending"].pop(k, None)
        if ent is not None:
            ent[0].wait()
