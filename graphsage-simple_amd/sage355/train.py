"""Training / evaluation harness: this build's counterpart of graphsage/model.py:52-69
(SupervisedGraphSage) and model.py:184-259 (run_model) -- SURVEY.md section 8 row f-1.

Kept from the reference: the classifier (weight [C, embed_dim], xavier, scores = (W . embeds)^T,
CrossEntropyLoss), the 10 / 10 / 80 test / val / train split of np.random.permutation
(model.py:229-234), SGD lr = 0.7 (model.py:237), random.shuffle of the train list per epoch,
micro/macro F1 on the validation split (model.py:256-258), mean batch time (model.py:259).
The per-epoch shuffle draws from a generator of its OWN (`random.Random(seed)`), not from Python's
global `random`: the neighbour samplers consume the global stream by a shard-dependent amount, so
data-parallel ranks sharing it would hold different permutations of `train` and `shard_batch` would
slice overlapping / incomplete shards.  With a private generator every rank holds the same
permutation and the union of the shards is exactly the global batch.
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
    does; `sample_seed` reseeds only Python's random, to vary the sampling stream on a fixed split."""
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
    shuffler = random.Random(seed)        # identical on every rank, untouched by neighbour sampling
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
