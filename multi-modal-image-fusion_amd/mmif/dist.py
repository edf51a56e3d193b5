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
    """ONE broadcast of all parameters (flattened) from `src`."""
    ps = list(module.parameters())
    if not ps:
        return
    flat = torch.cat([p.data.reshape(-1) for p in ps])
    dist.broadcast(flat, src)
    off = 0
    for p in ps:
        n = p.numel()
        p.data.copy_(flat[off:off + n].view(p.shape))
        off += n


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
