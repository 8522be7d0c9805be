"""Training / evaluation harness: this build's counterpart of graphsage/model.py:52-69
(SupervisedGraphSage) and model.py:184-259 (run_model) -- SURVEY.md section 8 row f-1.

Kept from the reference: the classifier (weight [C, embed_dim], xavier, scores = (W . embeds)^T,
CrossEntropyLoss), the 10 / 10 / 80 test / val / train split of np.random.permutation
(model.py:229-234), SGD lr = 0.7 (model.py:237), random.shuffle of the train list per epoch,
micro/macro F1 on the validation split (model.py:256-258), mean batch time (model.py:259).
Single process: the per-epoch shuffle is `random.shuffle(train)` on Python's GLOBAL stream, the one the strict
path's neighbour sampling also draws from -- as model.py:243 -- so a strict run consumes `random` call for
call like the reference.  Data parallel (world_size > 1): the shuffle draws from a generator of its OWN
(`random.Random(seed)`), because the samplers consume the global stream by a shard-dependent amount and ranks
sharing it would hold different permutations of `train`: `shard_batch` would slice overlapping / incomplete
shards.  With the private generator every rank holds the same permutation and the shards tile the global batch.
Deliberately different: batches are plain `batch_size` slices by default; `ref_batching=True`
reproduces the reference's `train[batch:max(train_num, batch+batch_size)]` descending batches
(model.py:244, a `max` where `min` was meant).  Everything stays on the GPU; with world_size > 1
each rank embeds its shard of the batch and the weight gradients are summed with one RCCL
all-reduce per step (dist.py).
"""
import random
import time

import numpy as np
import torch
import torch.nn as nn
from torch.nn import init

from . import dist
from .aggregators import MeanAggregator
from .encoders import Encoder


class SupervisedGraphSage(nn.Module):
    """model.py:52-69."""

    def __init__(self, num_classes, enc):
        super(SupervisedGraphSage, self).__init__()
        self.enc = enc
        self.xent = nn.CrossEntropyLoss()
        self.weight = nn.Parameter(torch.FloatTensor(num_classes, enc.embed_dim))
        init.xavier_uniform_(self.weight)

    def forward(self, nodes):
        embeds = self.enc(nodes)
        scores = self.weight.to(embeds.device).mm(embeds)
        return scores.t()

    def loss(self, nodes, labels):
        scores = self.forward(nodes)
        return self.xent(scores, labels.squeeze().to(scores.device))


def build_model(feat_data, adj_lists, num_classes, hidden1=50, hidden2=128, num_sample1=10, num_sample2=10, gcn=True,
                cuda=True):
    """model.py:214-227 with this package's classes (num_sample defaults to 10/10: what the reference
    actually uses, its `enc.num_samples = ...` assignments being no-ops)."""
    n, d = feat_data.shape
    features = nn.Embedding(n, d)
    features.weight = nn.Parameter(torch.as_tensor(feat_data, dtype=torch.float32), requires_grad=False)
    agg1 = MeanAggregator(features, cuda=cuda)
    enc1 = Encoder(features, d, hidden1, adj_lists, agg1, num_sample=num_sample1, gcn=gcn, cuda=cuda)
    agg2 = MeanAggregator(lambda nodes: enc1(nodes).t(), cuda=cuda)
    enc2 = Encoder(lambda nodes: enc1(nodes).t(), enc1.embed_dim, hidden2, adj_lists, agg2, num_sample=num_sample2,
                   base_model=enc1, gcn=gcn, cuda=cuda)
    return SupervisedGraphSage(num_classes, enc2)


def run_training(feat_data, labels, adj_lists, num_classes, seed=1, epochs=1, batch_size=128, ref_batching=False, lr=0.7,
                 model=None, verbose=True, sample_seed=None, return_model=False, on_batch=None, **model_kwargs):
    """-> dict(f1_micro, f1_macro, mean_batch_time, losses).  Mirrors run_model (model.py:184-259).
    `seed` seeds numpy (the split) and Python's random (shuffles + neighbour sampling) as model.py:192-193
    does; `sample_seed` reseeds only Python's random, to vary the sampling stream on a fixed split.  With
    world_size > 1 the shuffles come from a private `random.Random(seed)` instead (module docstring)."""
    from sklearn.metrics import f1_score
    np.random.seed(seed)
    random.seed(seed if sample_seed is None else sample_seed)
    rank, world = (torch.distributed.get_rank(), dist.world_size()) if dist.world_size() > 1 else (0, 1)
    num_nodes = feat_data.shape[0]
    if model is None:
        model = build_model(feat_data, adj_lists, num_classes, **model_kwargs)
    params = [p for p in model.parameters() if p.requires_grad]
    dist.broadcast_params(params)
    rand_indices = np.random.permutation(num_nodes)
    test = rand_indices[:int(0.1 * num_nodes)]
    val = rand_indices[int(0.1 * num_nodes):int(0.2 * num_nodes)]
    train = list(rand_indices[int(0.2 * num_nodes):])
    optimizer = torch.optim.SGD(params, lr=lr)
    # world 1: the global stream, as model.py:243; world > 1: a generator identical on every rank, untouched by neighbour sampling
    shuffler = random if world == 1 else random.Random(seed)
    labels_t = torch.as_tensor(labels, dtype=torch.int64).squeeze(-1)
    times, losses = [], []
    for _ in range(epochs):
        shuffler.shuffle(train)
        for batch in range(0, len(train), batch_size):
            hi = max(len(train), batch + batch_size) if ref_batching else min(len(train), batch + batch_size)
            batch_nodes = train[batch:hi]
            mine = dist.shard_batch(batch_nodes, rank, world)
            if on_batch is not None:
                on_batch(batch_nodes, mine)       # test hook: which nodes this rank was given
            start = time.time()
            optimizer.zero_grad()
            if len(mine):
                scores = model(mine)
                tgt = labels_t[np.asarray(mine)].to(scores.device)
                # sum of per-sample losses / GLOBAL batch size: the all-reduce SUM is then the full-batch gradient
                loss = nn.functional.cross_entropy(scores, tgt, reduction="sum") / len(batch_nodes)
                loss.backward()
                losses.append(float(loss.detach()) * len(batch_nodes) / len(mine))
            dist.all_reduce_grads(params)
            optimizer.step()
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            times.append(time.time() - start)
    with torch.no_grad():
        val_output = model(val)
    pred = val_output.detach().cpu().numpy().argmax(axis=1)
    truth = np.asarray(labels)[val].reshape(-1)
    res = {"f1_micro": float(f1_score(truth, pred, average="micro")), "f1_macro": float(f1_score(truth, pred, average="macro")),
           "mean_batch_time": float(np.mean(times)) if times else 0.0, "losses": losses, "test_nodes": test}
    if return_model:
        res["model"] = model
    if verbose and rank == 0:
        print("Validation F1 micro:", res["f1_micro"])
        print("Validation F1 macro:", res["f1_macro"])
        print("Average batch time:", res["mean_batch_time"])
    return res


def _sum_over_batch(g_scores, emb, piece=256):
    """g_scores^T . emb = [C, B] x [B, H] with the reduction over B in a FIXED order: 256-row pieces as one batched product, their
    sum over the piece index by torch.sum (one thread per output element adds them in index order), a ragged tail last."""
    b = g_scores.shape[0]
    whole = (b // piece) * piece
    if whole <= piece:
        return g_scores.t() @ emb
    n = whole // piece
    acc = torch.bmm(g_scores[:whole].view(n, piece, -1).transpose(1, 2), emb[:whole].view(n, piece, -1)).sum(0)
    if whole < b:
        acc = acc + g_scores[whole:].t() @ emb[whole:]
    return acc


class EngineTrainer:
    """One SGD step of SupervisedGraphSage (model.py:52-69, 240-252: forward, CrossEntropy, backward, SGD lr 0.7) with NOTHING
    on the host but enqueues: device sampler at both hops and both layers through TwoHopEngine (sage_forward2), the classifier and
    the loss as a handful of torch ops on the same stream, the backward through the engine's own intermediates (nbr / cnt / row2 / h1
    stay in its workspace) with the C-ABI backward kernels, and in-place SGD updates.  No host synchronisation inside a step: the
    frontier size never leaves the device (every kernel takes its row count from the workspace).  run_training above is the
    module-level path (the reference's class surface: since round 3 it runs the same engine forward and backward as ONE autograd
    node, ~1.2 ms per 256-seed step with the host-side classifier of model.py); this is the throughput path: 0.42 ms per 256-seed
    step on stand-in Cora (0.24 ms as a captured hipGraph), 0.31 ms per 4096-seed step at config-3 size, against 140-180 ms for
    the reference on a CPU.

    table is frozen (model.py:214-215), so layer 1 needs no input gradient.  Gradients follow torch autograd of the reference's
    expression: d relu, d sigmoid, d mean = 1/|set| per member -- TwoHopEngine.backward_weights: sage_linear_act_backward_ws for
    layer 2, sage_two_hop_grad_w1 (the layer-1 weight gradient summed over the outer samples) for layer 1; no float atomics, two
    runs of one schedule give the same bits."""

    def __init__(self, rowptr, col, table, num_classes, hidden1=50, hidden2=128, num_sample1=10, num_sample2=10, gcn=True, lr=0.7,
                 max_batch=256, agg_self_loop=False, relabel=None):
        from . import native, ops
        from .engine import TwoHopEngine
        dev = table.device
        d0 = table.shape[1]
        m = 1 if gcn else 2
        self.w1 = torch.empty(hidden1, m * d0, device=dev)
        self.w2 = torch.empty(hidden2, m * hidden1, device=dev)
        self.w_cls = torch.empty(num_classes, hidden2, device=dev)
        for w in (self.w1, self.w2, self.w_cls):
            init.xavier_uniform_(w)
        self.lr = float(lr)
        self.concat = not gcn
        # data parallel: every rank starts from rank 0's weights.  BEFORE the engine is built: its constructor prepares the bf16
        # planes of W1 (and zero-padded copies of odd widths) and caches them under (data_ptr, _version), which a broadcast
        # through `.data` does not move -- ranks != 0 would run their first step on planes of their own initial W1
        dist.broadcast_params(self.parameters())
        self.engine = TwoHopEngine(rowptr, col, table, self.w1, self.w2, num_sample1, num_sample2, concat=self.concat,
                                   agg_self_loop=agg_self_loop, max_batch=max_batch, relabel=relabel)
        self.engine.invalidate_weights()
        self._native, self._ops = native, ops
        self._out_q = None
        self._step_graph = None

    def parameters(self):
        return [self.w1, self.w2, self.w_cls]

    def embed(self, seeds, key=0):
        """[B, hidden2] embeddings (no grad): the evaluation forward."""
        return self.engine.forward(seeds, seed=key)

    def scores(self, seeds, key=0):
        return self.embed(seeds, key) @ self.w_cls.t()

    def grads(self, seeds, labels, key, global_batch=None):
        """loss (device scalar) and the gradients of (w1, w2, w_cls) for one batch; nothing is updated.
        global_batch: data parallel -- this rank holds a shard of a mini-batch of that many seeds; its loss is the SUM over its
        shard / global_batch, so that the SUM of the ranks' gradients is the full-batch gradient."""
        e = self.engine
        if seeds is None:                                                  # the batch at the engine's queue cursor (capturable step)
            b = e._queue_batch
            if self._out_q is None or self._out_q.shape[0] != b:
                self._out_q = torch.empty(b, e.h2, device=e.device)
            out = e.forward_queued(self._out_q)
        else:
            b = seeds.shape[0]
            out = e.forward(seeds, seed=key)                               # sample, frontier, sample, layer 1, layer 2
        # classifier + loss + their gradients: stock torch on the same stream (model.py:59-69)
        # The classifier's weight gradient is a [C, B] x [B, H2] product whose reduction runs over the BATCH: for B >= 1024 the BLAS behind
        # torch.mm may split it and add the pieces with atomics, and two runs of one schedule then differ in the last bits of w_cls (seen
        # once in four runs of the GPU suite, 1024-seed case only).  So autograd stops at the scores and the two small products are
        # spelled out; the one over the batch is cut into 256-row pieces (one batched product) that are added in a fixed order.
        # (Accumulating it in fp64 instead costs 0.2 ms per step: 0.31 -> 0.53 ms captured at config-3 size.)
        emb = out.detach()
        scores = (emb @ self.w_cls.detach().t()).requires_grad_(True)
        if global_batch is None:
            loss = nn.functional.cross_entropy(scores, labels)
        else:
            loss = nn.functional.cross_entropy(scores, labels, reduction="sum") / float(global_batch)
        (g_scores,) = torch.autograd.grad(loss, (scores,))
        g_out = g_scores @ self.w_cls.detach()                              # [B, C] x [C, H2]: reduction over the classes
        g_cls = _sum_over_batch(g_scores, emb)                               # [C, B] x [B, H2]: reduction over the batch
        # both layers' weight gradients from the intermediates this forward left in the engine's workspace
        g_w1, g_w2 = e.backward_weights(out, g_out)
        return loss.detach(), (g_w1, g_w2, g_cls)

    def step(self, seeds, labels, key, global_batch=None):
        """forward + backward + SGD; -> loss as a device scalar (reading it is the caller's only synchronisation).
        With torch.distributed initialised (one process per GPU, backend "nccl" = RCCL over xGMI) the three weight gradients
        travel in ONE flat all-reduce (SUM) per step, ~0.3 MB: latency bound, so one buffer and one call."""
        loss, (g1, g2, gc) = self.grads(seeds, labels, key, global_batch)
        if dist.world_size() > 1:
            flat = torch.cat([g1.reshape(-1), g2.reshape(-1), gc.reshape(-1)])
            torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.SUM)
            n1, n2 = g1.numel(), g2.numel()
            g1, g2, gc = flat[:n1].view_as(g1), flat[n1:n1 + n2].view_as(g2), flat[n1 + n2:].view_as(gc)
        self.w1.add_(g1, alpha=-self.lr)
        self.w2.add_(g2, alpha=-self.lr)
        self.w_cls.add_(gc, alpha=-self.lr)
        return loss


    # ---- the whole step as ONE hipGraph ("graph-capturable fwd+bwd") ------------------------------------------------
    def capture_step(self, seeds_ring, keys, labels_by_node):
        """seeds_ring: int32 device tensor [S, B] of mini-batches, keys: S sampler keys, labels_by_node: int64 device tensor [N].
        Captures forward + loss + backward + SGD of the batch at the engine's queue cursor into one torch.cuda.CUDAGraph:
        replay_step() then trains on batch after batch of the ring (wrapping around) with a single graph launch each, nothing
        read from the host: the seeds and the sampler key come from the device-side ring (sage_batch_t), the labels are
        gathered by the ids at the cursor, the weight planes are re-prepared inside the graph after every update.
        -> the static device scalar that holds the last replayed step's loss."""
        e = self.engine
        if dist.world_size() > 1:
            raise self._native.SageError("capture_step: single-process only (the data-parallel step has a collective per step)")
        e.set_queue(seeds_ring, keys)
        ring, nring = e._queue_seeds, seeds_ring.shape[0]
        labels_by_node = labels_by_node.to(e.device)
        self._loss_static = torch.zeros((), device=e.device)

        def one():
            cur = torch.remainder(e._cursor.to(torch.int64), nring)       # read BEFORE the forward's last kernel advances the cursor
            ids = ring.index_select(0, cur)[0]
            labels = labels_by_node.index_select(0, ids.to(torch.int64))
            self._loss_static.copy_(self.step(None, labels, None))

        saved = [w.clone() for w in self.parameters()]
        side = torch.cuda.Stream(device=e.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                                     # warm-up outside capture: a real step, undone below
            one()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            one()
        with torch.no_grad():
            for w, w0 in zip(self.parameters(), saved):
                w.copy_(w0)
        e._model()                                                        # planes of the restored W1
        e.rewind(0)
        torch.cuda.synchronize()
        self._step_graph = g
        return self._loss_static

    def replay_step(self):
        self._step_graph.replay()
        # the replay updated the weights without moving their version counters: an eager forward after it (validation, a
        # plain step) must not find its cached planes / padded copies of W_{t-1} "up to date" 
        self.engine.invalidate_weights()
        return self._loss_static


def run_engine_training(graph, feat_data, labels, num_classes, seed=1, epochs=1, batch_size=128, ref_batching=False, lr=0.7,
                        sample_seed=0, hidden1=50, hidden2=128, num_sample1=10, num_sample2=10, gcn=True):
    """run_model (model.py:184-259) on the engine path: same split, shuffles, batching and optimiser as run_training above, every
    step through EngineTrainer.  -> dict(f1_micro, f1_macro, mean_step_time, losses, trainer)."""
    from sklearn.metrics import f1_score
    dev = torch.device("cuda")
    np.random.seed(seed)
    rowptr, col = graph.to(dev)
    table = torch.as_tensor(feat_data, dtype=torch.float32).to(dev)
    n = graph.num_nodes
    rand_indices = np.random.permutation(n)
    val = rand_indices[int(0.1 * n):int(0.2 * n)]
    train = list(rand_indices[int(0.2 * n):])
    top = len(train) if ref_batching else batch_size
    tr = EngineTrainer(rowptr, col, table, num_classes, hidden1, hidden2, num_sample1, num_sample2, gcn=gcn, lr=lr, max_batch=max(top, len(val)))
    labels_dev = torch.as_tensor(np.asarray(labels).reshape(-1), dtype=torch.int64).to(dev)
    shuffler = random.Random(seed)
    losses, key = [], int(sample_seed) << 20
    torch.cuda.synchronize()
    t0 = time.time()
    steps = 0
    for _ in range(epochs):
        shuffler.shuffle(train)
        order = torch.as_tensor(np.asarray(train), dtype=torch.int32).to(dev)
        for lo in range(0, len(train), batch_size):
            hi = max(len(train), lo + batch_size) if ref_batching else min(len(train), lo + batch_size)
            ids = order[lo:hi].contiguous()
            losses.append(tr.step(ids, labels_dev[ids.long()], key))
            key += 1
            steps += 1
    torch.cuda.synchronize()
    elapsed = time.time() - t0
    with torch.no_grad():
        pred = tr.scores(torch.as_tensor(val.astype(np.int32)).to(dev), key=key).argmax(1).cpu().numpy()
    truth = np.asarray(labels)[val].reshape(-1)
    return {"f1_micro": float(f1_score(truth, pred, average="micro")), "f1_macro": float(f1_score(truth, pred, average="macro")),
            "mean_step_time": elapsed / max(steps, 1), "losses": [float(x) for x in losses], "trainer": tr}
