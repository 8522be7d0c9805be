"""Seed-shard data parallelism over torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU).

The reference is single process (no torch.distributed import anywhere).  The
path shards by independent units -- each seed's 2-hop embedding depends only on the replicated graph,
the replicated feature table and the weights -- so:

  forward   no collective: every rank embeds its own shard of the mini-batch;
  training  ONE all-reduce per step over the flattened weight gradients (~0.3 MB: latency bound on
            xGMI, so a single flat buffer and a single call, never one call per tensor);
  timing    barrier + MAX over ranks (bench.py).

One process per GPU, launched by torch.distributed.run; rendezvous on 127.0.0.1.
"""
import os

import torch
import torch.distributed as td


def init_from_env(backend=None, device=None):
    """-> (rank, world_size, local_rank).  No-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not td.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kwargs["device_id"] = device if device is not None else torch.device("cuda", local_rank)
        td.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local_rank


def world_size():
    return td.get_world_size() if td.is_available() and td.is_initialized() else 1


def shard_bounds(n, rank, world):
    """Contiguous, near-equal split of n units: ranks [0, n % world) get one extra."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(nodes, rank=None, world=None):
    """This rank's contiguous slice of a mini-batch of seed nodes."""
    if world is None:
        world = world_size()
    if rank is None:
        rank = td.get_rank() if world > 1 else 0
    lo, hi = shard_bounds(len(nodes), rank, world)
    return nodes[lo:hi]


def all_reduce_grads(params, average=False):
    """SUM the gradients of `params` over all ranks with ONE collective on one flat buffer.
    Ranks scale their local loss by 1/global_batch beforehand, so SUM gives the full-batch
    gradient exactly (average=True divides by the world size instead)."""
    world = world_size()
    params = [p for p in params if p.requires_grad]
    if world == 1 or not params:
        return
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    if flat.device.type == "cpu" and td.get_backend() == "nccl":
        flat = flat.cuda()
    td.all_reduce(flat, op=td.ReduceOp.SUM)
    if average:
        flat /= world
    off = 0
    for p in params:
        k = p.numel()
        p.grad.copy_(flat[off:off + k].view_as(p.grad))
        off += k


def broadcast_params(params, src=0):
    if world_size() == 1:
        return
    for p in params:
        t = p.data
        if t.device.type == "cpu" and td.get_backend() == "nccl":
            buf = t.cuda()
            td.broadcast(buf, src)
            t.copy_(buf)
        else:
            td.broadcast(t, src)


def max_over_ranks(value, device="cpu"):
    if world_size() == 1:
        return float(value)
    if td.get_backend() == "nccl":
        device = torch.device("cuda", torch.cuda.current_device())
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if world_size() > 1:
        td.barrier()
