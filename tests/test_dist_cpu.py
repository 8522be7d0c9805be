"""world_size-2 gloo tests of the N > 1 path (runs on CPU): shard bookkeeping, the single flat
gradient all-reduce, max-over-ranks timing, and that seed-sharded sum/global-batch losses reproduce
the full-batch gradient -- the invariant bench.py / train.py rely on with RCCL on the GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from sage355 import dist


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, tmp):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = dist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dist.world_size() == world
    # 1. shards tile the batch exactly once
    batch = list(range(1003))
    mine = dist.shard_batch(batch, r, w)
    gathered = [None] * w
    torch.distributed.all_gather_object(gathered, mine)
    assert sum(gathered, []) == batch
    # 2. sharded (sum / global batch) losses + ONE flat all-reduce == full-batch gradient
    torch.manual_seed(0)
    x = torch.randn(1003, 16)
    y = torch.randint(0, 5, (1003,))
    model = torch.nn.Sequential(torch.nn.Linear(16, 8), torch.nn.ReLU(), torch.nn.Linear(8, 5))
    params = list(model.parameters())
    dist.broadcast_params(params)
    full = torch.nn.functional.cross_entropy(model(x), y)
    want = torch.autograd.grad(full, params)
    model.zero_grad()
    idx = torch.tensor(mine)
    loss = torch.nn.functional.cross_entropy(model(x[idx]), y[idx], reduction="sum") / len(batch)
    loss.backward()
    dist.all_reduce_grads(params)
    for p, g in zip(params, want):
        assert torch.allclose(p.grad, g, atol=1e-6), (p.grad - g).abs().max()
    # 3. a parameter that got no gradient on this rank still takes part in the collective
    extra = torch.nn.Parameter(torch.ones(3))
    if r == 0:
        extra.grad = torch.full((3,), 2.0)
    dist.all_reduce_grads([extra])
    assert torch.equal(extra.grad, torch.full((3,), 2.0))
    # 4. timing reduction
    assert dist.max_over_ranks(1.0 + r) == float(w)
    dist.barrier()
    open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_data_parallel_path(tmp_path):
    port = free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


class _RandomHungryModel(torch.nn.Module):
    """CPU stand-in for SupervisedGraphSage: like the neighbour samplers it draws from Python's GLOBAL `random`,
    by an amount that depends on the shard it was handed (so ranks de-synchronise that stream at once)."""

    def __init__(self, num_nodes, classes):
        super().__init__()
        self.emb = torch.nn.Embedding(num_nodes, classes)

    def forward(self, nodes):
        import random
        for _ in range(int(sum(int(n) for n in nodes) % 17) + 1):
            random.getrandbits(64)
        return self.emb(torch.as_tensor(np.asarray(nodes), dtype=torch.int64))


def _train_worker(rank, world, port, tmp):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    dist.init_from_env(backend="gloo")
    from sage355.train import run_training
    n, classes = 203, 3
    torch.manual_seed(0)
    model = _RandomHungryModel(n, classes)
    labels = np.random.default_rng(0).integers(0, classes, (n, 1))
    seen = []
    # per-rank sampler streams differ (tests/test_gpu_train.py seeds them per rank); the shuffle must not care
    run_training(np.zeros((n, 4), np.float32), labels, None, classes, seed=3, epochs=3, batch_size=32, model=model, verbose=False,
                 sample_seed=50 + rank, on_batch=lambda batch, mine: seen.append(([int(x) for x in batch], [int(x) for x in mine])))
    everyone = [None] * world
    torch.distributed.all_gather_object(everyone, seen)
    assert len(seen) >= 3 * 5
    for step in range(len(seen)):
        batches = [everyone[r][step][0] for r in range(world)]
        shards = [everyone[r][step][1] for r in range(world)]
        assert all(b == batches[0] for b in batches), f"step {step}: ranks hold different permutations of the train list"
        assert sum(shards, []) == batches[0], f"step {step}: shards do not tile the global batch"
        assert len(set(sum(shards, []))) == len(batches[0])
    # the permutation really changes from epoch to epoch (it is a shuffle, not a fixed order)
    per_epoch = len(seen) // 3
    assert everyone[0][0][0] != everyone[0][per_epoch][0]
    open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    torch.distributed.destroy_process_group()


def test_two_rank_training_shards_tile_every_global_batch(tmp_path):
    """the epoch shuffle used Python's global `random`, which the samplers consume by a shard-dependent
    amount -> ranks held different permutations.  Three epochs, two ranks, different sampler seeds per rank."""
    port = free_port()
    mp.spawn(_train_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_shard_bounds_cover_every_unit_once():
    for n in (0, 1, 7, 4096, 4097):
        for world in (1, 2, 3, 8):
            spans = [dist.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_helpers_are_noops():
    p = torch.nn.Parameter(torch.ones(2))
    p.grad = torch.ones(2)
    dist.all_reduce_grads([p])
    assert torch.equal(p.grad, torch.ones(2))
    assert dist.max_over_ranks(3.5) == 3.5 and dist.world_size() == 1
    assert dist.shard_batch([1, 2, 3]) == [1, 2, 3]
