"""Engine-level training step: sage355.train.EngineTrainer -- device sampler at both hops, forward through
TwoHopEngine, backward through its saved intermediates with the C-ABI backward kernels, SGD in place, no host
synchronisation inside a step.  Gradients against fp64 autograd on the very same sampled sets; F1 against the REFERENCE's
F1 on the stand-in datasets (tests/golden/reference_f1_*_standin.json); step time against the 1 ms target."""
import json
import os

import numpy as np
import pytest
import torch

from sage355.datasets import standin_citation
from sage355.graph import CSRGraph, rmat_graph
from sage355.train import EngineTrainer, run_engine_training
from util import GOLDEN_DIR

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _topology(name):
    z = np.load(os.path.join(GOLDEN_DIR, name))
    return CSRGraph(z["rowptr"], z["col"], len(z["rowptr"]) - 1)


def _autograd_reference(tr, seeds, labels):
    """fp64 torch autograd of the reference's expression (aggregators.py:54-74 mean, encoders.py:49-62 concat + W.x + relu,
    model.py:59-69 classifier + CrossEntropy) on the sets the engine sampled."""
    e = tr.engine
    it = e.intermediates()
    first = it["first_frontier_row"]
    nbr2, cnt2, row2 = (it[k].cpu().long() for k in ("nbr2", "cnt2", "row2"))
    nbr1, cnt1 = it["nbr1"].cpu().long(), it["cnt1"].cpu().long()
    s1 = it["s1_nodes"].cpu().long()
    table = e.table[:, :e.d0].cpu().double()
    w1 = tr.w1.detach().cpu().double().requires_grad_(True)
    w2 = tr.w2.detach().cpu().double().requires_grad_(True)
    wc = tr.w_cls.detach().cpu().double().requires_grad_(True)

    def mean_rows(src, idx, cnt):
        m = (torch.arange(idx.shape[1])[None, :] < cnt[:, None]).double()
        return (src[idx.clamp_min(0)] * m[:, :, None]).sum(1) / cnt[:, None].double()

    agg1 = mean_rows(table, nbr1, cnt1)
    x1 = torch.cat([table[s1], agg1], 1) if tr.concat else agg1
    h1 = torch.relu(x1 @ w1.t())
    agg2 = mean_rows(h1, row2, cnt2)
    x2 = torch.cat([h1[:len(seeds)], agg2], 1) if tr.concat else agg2
    out = torch.relu(x2 @ w2.t())
    loss = torch.nn.functional.cross_entropy(out @ wc.t(), labels.cpu())
    g = torch.autograd.grad(loss, (w1, w2, wc))
    return loss.item(), g


@pytest.mark.parametrize("relabel", [None, "degree"])
@pytest.mark.parametrize("gcn,d0,h1", [(True, 256, 128), (False, 100, 52), (True, 1433, 50), (False, 66, 30)])
def test_engine_gradients_match_fp64_autograd_on_the_same_sets(gcn, d0, h1, relabel):
    graph = rmat_graph(13, 150_000, seed=4, accel=None)
    gen = torch.Generator().manual_seed(1)
    table = torch.randn(graph.num_nodes, d0, generator=gen).to(DEV)
    rowptr, col = graph.to(DEV)
    torch.manual_seed(3)
    tr = EngineTrainer(rowptr, col, table, 5, hidden1=h1, hidden2=64, num_sample1=7, num_sample2=9, gcn=gcn, max_batch=300, relabel=relabel)
    seeds = np.random.default_rng(2).choice(np.nonzero(graph.degrees() > 0)[0], 300, replace=False)
    labels = torch.from_numpy(np.random.default_rng(3).integers(0, 5, 300)).to(DEV)
    loss, grads = tr.grads(torch.from_numpy(seeds.astype(np.int32)).to(DEV), labels, key=11)
    ref_loss, ref = _autograd_reference(tr, seeds, labels)
    assert abs(loss.item() - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss))
    for name, g, r in zip(("w1", "w2", "w_cls"), grads, ref):
        scale = r.abs().max().item()
        err = (g.cpu().double() - r).abs().max().item() / scale
        assert err <= 2e-5, f"grad {name}: max |g - ref| / max|ref| = {err:.2e}"


def test_engine_step_at_config3_size_gradients_and_step_time():
    """BASELINE configs[2] dimensions (R-MAT 2^20 / 16 M edges, D0 = 256, H = 128/128, fanout 15/25, 4096 seeds, degree layout):
    the step's gradients against fp64 autograd on the ~23.5 k-row layer the engine sampled, and a step in under 1.5 ms
    (measured 0.39 ms = 1e7 seeds/s, experiments/train_big.py)."""
    import time
    graph = rmat_graph(20, 16_000_000, seed=0, cache_dir=os.environ.get("SAGE_CACHE", "/tmp/sage_cache"))
    gen = torch.Generator(device=DEV).manual_seed(0)
    table = torch.randn(graph.num_nodes, 256, generator=gen, device=DEV)
    rowptr, col = graph.to(DEV)
    torch.manual_seed(0)
    tr = EngineTrainer(rowptr, col, table, 16, hidden1=128, hidden2=128, num_sample1=15, num_sample2=25, gcn=True, lr=0.05,
                       max_batch=4096, relabel="degree")
    rs = np.random.default_rng(1)
    cand = np.nonzero(graph.degrees() > 0)[0]
    batches = [rs.choice(cand, 4096, replace=False) for _ in range(24)]
    labels = torch.from_numpy(rs.integers(0, 16, graph.num_nodes)).to(DEV)
    s0 = torch.from_numpy(batches[0].astype(np.int32)).to(DEV)
    loss, grads = tr.grads(s0, labels[s0.long()], key=7)
    assert tr.engine.intermediates()["n_s1"] > 15000
    ref_loss, ref = _autograd_reference(tr, batches[0], labels[s0.long()])
    assert abs(loss.item() - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss))
    for name, g, r in zip(("w1", "w2", "w_cls"), grads, ref):
        err = (g.cpu().double() - r).abs().max().item() / r.abs().max().item()
        assert err <= 2e-5, f"grad {name}: max |g - ref| / max|ref| = {err:.2e}"
    dev_batches = [torch.from_numpy(x.astype(np.int32)).to(DEV) for x in batches]
    for i in range(4):
        tr.step(dev_batches[i], labels[dev_batches[i].long()], key=100 + i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(4, 24):
        tr.step(dev_batches[i], labels[dev_batches[i].long()], key=100 + i)
    torch.cuda.synchronize()
    per_step = (time.perf_counter() - t0) / 20
    assert per_step < 1.5e-3, f"{per_step * 1e3:.2f} ms per 4096-seed step"
    assert all(torch.isfinite(w).all() for w in tr.parameters())


@pytest.mark.parametrize("gcn,relabel,hidden1", [(True, None, 64), (False, "degree", 64), (True, None, 30)])
def test_captured_step_trains_like_the_eager_step(gcn, relabel, hidden1):
    """EngineTrainer.capture_step: forward + loss + backward + SGD of the batch at the queue cursor as ONE hipGraph.  Replaying it
    over a ring of mini-batches must leave the weights an eager loop over the same batches and keys leaves -- bit for bit since the
    end of round 3 (no float atomics on the path, fills as kernels: a captured hipMemsetAsync went wrong from its second replay on) --
    wrap around the ring, and report the same losses."""
    graph = rmat_graph(13, 150_000, seed=4, accel=None)
    gen = torch.Generator().manual_seed(1)
    table = torch.randn(graph.num_nodes, 128, generator=gen).to(DEV)
    rowptr, col = graph.to(DEV)
    labels_by_node = torch.from_numpy(np.random.default_rng(3).integers(0, 5, graph.num_nodes)).to(DEV)
    cand = np.nonzero(graph.degrees() > 0)[0]
    ring = torch.from_numpy(np.stack([np.random.default_rng(10 + i).choice(cand, 256, replace=False) for i in range(4)]).astype(np.int32)).to(DEV)
    keys = [101, 102, 103, 104]

    def make():
        torch.manual_seed(5)
        return EngineTrainer(rowptr, col, table, 5, hidden1=hidden1, hidden2=32, num_sample1=7, num_sample2=9, gcn=gcn, lr=0.3, max_batch=256,
                             relabel=relabel)

    eager = make()
    eager_losses = []
    for i in range(6):                                           # 6 steps over a ring of 4: wraps around
        j = i % 4
        eager_losses.append(float(eager.step(ring[j], labels_by_node[ring[j].long()], keys[j])))
    cap = make()
    for a, b in zip(cap.parameters(), make().parameters()):
        assert torch.equal(a, b)
    loss = cap.capture_step(ring, keys, labels_by_node)
    for a, b in zip(cap.parameters(), make().parameters()):
        assert torch.equal(a, b)                                 # capture (and its warm-up step) left the weights alone
    cap_losses = []
    for i in range(6):
        cap.replay_step()
        cap_losses.append(float(loss))
    assert cap_losses == eager_losses, (cap_losses, eager_losses)
    assert all(np.isfinite(cap_losses)) and (hidden1 != 64 or cap_losses[-1] < cap_losses[0])
    for name, a, b in zip(("w1", "w2", "w_cls"), cap.parameters(), eager.parameters()):
        assert torch.equal(a, b), f"{name}: captured vs eager differ by {(a - b).abs().max().item():.3e} at {int((a != b).sum())} elements"
    # a replay moves the weights but not their version counters; an EAGER forward after it (validation) must run on
    # W_t, not on padded copies / bf16 planes of W_{t-1}: bit-identical to a fresh engine built from the trainer's weights
    from sage355.engine import TwoHopEngine
    val = torch.from_numpy(np.random.default_rng(77).choice(cand, 200, replace=False).astype(np.int32)).to(DEV)
    fresh = TwoHopEngine(rowptr, col, table, cap.w1.clone(), cap.w2.clone(), 7, 9, concat=not gcn, max_batch=256, relabel=relabel)
    assert torch.equal(cap.embed(val, key=5), fresh.forward(val, seed=5)), "eager forward after replays ran on stale weight caches"
    sc, se = cap.scores(val, key=5), eager.scores(val, key=5)
    assert (sc - se).abs().max().item() <= 2e-3 * se.abs().max().item()


def test_engine_training_step_on_standin_cora_takes_under_a_millisecond():
    """Plain 256-seed steps through EngineTrainer: target <= 1 ms per step (the reference 140-180 ms on a CPU).  F1 against
    the reference's distribution over sampling streams -- Cora and Pubmed, this path and the module-level ones -- is
    tests/test_gpu_train.py::test_f1_distribution_over_sampling_streams_matches_the_reference."""
    g = _topology("cora_topology.npz")
    feats, labels = standin_citation(g, num_classes=7, feat_dim=1433, seed=0)
    torch.manual_seed(0)
    res = run_engine_training(g, feats, labels, 7, seed=1, epochs=1, batch_size=256)
    res = run_engine_training(g, feats, labels, 7, seed=1, epochs=4, batch_size=256)
    print(f"engine step (256 seeds, 1433 -> 50 -> 128, fanout 10/10): {res['mean_step_time'] * 1e3:.3f} ms")
    assert res["mean_step_time"] <= 1.0e-3, res["mean_step_time"]
    assert res["f1_micro"] > 0.85


def _dp_rank(rank, world, port, tmp, d0=64, h1=32):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from sage355 import dist
    dist.init_from_env(backend="gloo")          # both ranks share cuda:0 here; on a node it is nccl (RCCL), one GPU per rank
    graph = rmat_graph(13, 150_000, seed=4, accel=None)
    gen = torch.Generator().manual_seed(1)
    table = torch.randn(graph.num_nodes, d0, generator=gen).to(DEV)
    rowptr, col = graph.to(DEV)
    torch.manual_seed(10 + rank)                # different initial weights per rank: the broadcast must make them equal
    tr = EngineTrainer(rowptr, col, table, 4, hidden1=h1, hidden2=16, num_sample1=5, num_sample2=5, gcn=True, max_batch=128)
    rs = np.random.default_rng(0)
    cand = np.nonzero(graph.degrees() > 0)[0]
    labels_all = torch.from_numpy(rs.integers(0, 4, graph.num_nodes)).to(DEV)
    losses = []
    for step in range(6):
        batch = rs.choice(cand, 256, replace=False)               # same global batch on every rank (same generator state)
        mine = dist.shard_batch(list(batch), rank, world)
        ids = torch.as_tensor(np.asarray(mine, dtype=np.int32)).to(DEV)
        losses.append(float(tr.step(ids, labels_all[ids.long()], key=1000 * rank + step, global_batch=len(batch))))
        if step == 0:
            w_after_first = torch.cat([p.reshape(-1).cpu() for p in tr.parameters()])
    w = torch.cat([p.reshape(-1).cpu() for p in tr.parameters()])
    torch.save({"w": w, "losses": losses, "w_after_first": w_after_first}, os.path.join(tmp, f"r{rank}.pt"))
    dist.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("d0,h1", [(64, 32), (66, 30)])       # (66, 30): zero-padded widths -- cached copies of the weights inside the engine
def test_engine_trainer_data_parallel_one_flat_all_reduce_keeps_replicas_identical(tmp_path, d0, h1):
    """seeds shard across ranks, graph / table / weights are replicated, ONE all-reduce of the weight gradients per
    step.  Two ranks (gloo, sharing this box's GPU; RCCL on a node): started from different weights, they must be bit-identical
    after the broadcast and stay so through six steps, each with its own sampler stream."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_dp_rank, args=(2, port, str(tmp_path), d0, h1), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(a["w"], b["w"]), "replicas diverged"
    assert all(np.isfinite(a["losses"])) and all(np.isfinite(b["losses"]))
    # identical replicas and finite losses are what the all-reduce guarantees even when a rank's FIRST step ran on caches
    # of its own pre-broadcast weights.  The first step must be the single-process step from rank 0's weights: the sum over the
    # two shards' gradients (each with its rank's sampler key), applied once.
    from sage355 import dist
    graph = rmat_graph(13, 150_000, seed=4, accel=None)
    gen = torch.Generator().manual_seed(1)
    table = torch.randn(graph.num_nodes, d0, generator=gen).to(DEV)
    rowptr, col = graph.to(DEV)
    torch.manual_seed(10)                       # rank 0's initial weights
    tr = EngineTrainer(rowptr, col, table, 4, hidden1=h1, hidden2=16, num_sample1=5, num_sample2=5, gcn=True, max_batch=128)
    rs = np.random.default_rng(0)
    cand = np.nonzero(graph.degrees() > 0)[0]
    labels_all = torch.from_numpy(rs.integers(0, 4, graph.num_nodes)).to(DEV)
    batch = rs.choice(cand, 256, replace=False)
    total = [torch.zeros_like(p) for p in tr.parameters()]
    for r in range(2):
        ids = torch.as_tensor(np.asarray(dist.shard_batch(list(batch), r, 2), dtype=np.int32)).to(DEV)
        _, g = tr.grads(ids, labels_all[ids.long()], key=1000 * r, global_batch=256)
        for t, gi in zip(total, g):
            t += gi
    want = torch.cat([(p - tr.lr * t).reshape(-1).cpu() for p, t in zip(tr.parameters(), total)])
    err = (a["w_after_first"] - want).abs().max().item() / want.abs().max().item()
    assert err <= 1e-5, f"first data-parallel step differs from the single-process step from rank 0's weights: {err:.2e}"
