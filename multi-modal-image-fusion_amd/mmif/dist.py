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
