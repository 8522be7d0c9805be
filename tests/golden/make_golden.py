#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING THE REFERENCE (runs only in the build
container, where /root/reference is mounted; the reference never travels).

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Each fixture holds inputs and the reference's own outputs for one case of the
hot path (graphsage/aggregators.py:34-76, graphsage/encoders.py:40-62):
pre-sampled neighbour sets are injected through the reference's
``num_sample=None`` switch (aggregators.py:47-48), so the arithmetic is pinned
independently of any random stream.  One case also runs the reference with its
sampler ON after ``random.seed`` to pin oracle/ref_dense.sample_sets' use of
Python's ``random`` stream.

Only data is written: node ids, neighbour lists, the touched feature rows,
weights, outputs, gradients.
"""
import io
import os
import random
import sys
import warnings
from contextlib import redirect_stdout

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REF)   # the reference's `graphsage` must win over this repo's drop-in shim
sys.path.insert(1, os.path.join(REPO, "graphsage-simple_amd"))
sys.path.insert(2, os.path.join(REPO, "tests"))
warnings.filterwarnings("ignore")

from graphsage.aggregators import MeanAggregator  # noqa: E402  (the reference)
from graphsage.encoders import Encoder            # noqa: E402  (the reference)
from sage355 import graph as G                    # noqa: E402  (host-only ingestion)
from util import TABLE_KINDS, synth_table         # noqa: E402  (tests/util.py: the batch-size fixtures' table generator)


def quiet(fn, *a, **k):
    with redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def pad_sets(nodes, sets, k):
    nbr = np.full((len(nodes), max(k, 1)), -1, dtype=np.int64)
    cnt = np.zeros(len(nodes), dtype=np.int64)
    for r, n in enumerate(nodes):
        s = sorted(sets[int(n)])
        cnt[r] = len(s)
        nbr[r, :len(s)] = s
    return nbr, cnt


def presample(adj, nodes, k, rng):
    out = {}
    for n in nodes:
        neigh = sorted(adj[int(n)])
        out[int(n)] = set(rng.sample(neigh, k)) if len(neigh) >= k else set(neigh)
    return out


def xavier(shape, gen):
    fan_out, fan_in = shape
    bound = float(np.sqrt(6.0 / (fan_in + fan_out)))
    return (torch.rand(shape, generator=gen) * 2 - 1) * bound


def build_reference_stack(table, sets1, sets2, w1, w2, gcn, init1="None", init2="None"):
    """graphsage/model.py:214-222 wiring with injected, pre-sampled adjacency."""
    n, d0 = table.shape
    features = torch.nn.Embedding(n, d0)
    features.weight = torch.nn.Parameter(table.clone(), requires_grad=False)
    agg1 = MeanAggregator(features, cuda=False)
    enc1 = quiet(Encoder, features, d0, w1.shape[0], sets1, agg1, num_sample=None,
                 gcn=gcn, cuda=False, initializer=init1)
    agg2 = MeanAggregator(lambda nodes: enc1(nodes).t(), cuda=False)
    enc2 = quiet(Encoder, lambda nodes: enc1(nodes).t(), enc1.embed_dim, w2.shape[0], sets2, agg2,
                 num_sample=None, base_model=enc1, gcn=gcn, cuda=False, initializer=init2)
    with torch.no_grad():
        enc1.weight.copy_(w1)
        enc2.weight.copy_(w2)
    return features, agg1, enc1, agg2, enc2


def two_layer_case(name, graph, table, seeds, k1, k2, h1, h2, gcn, seed, init1="None", init2="None", table_spec=None):
    """table_spec = (kind code, generator seed): a BATCH-SIZE fixture -- `table` came from tests/util.synth_table(kind, N, D0, seed) and
    is NOT stored (its sha256 is; the tests regenerate it), and of the [|S1|, D0] layer-1 aggregator output only 16 rows are."""
    rng = random.Random(seed)
    gen = torch.Generator().manual_seed(seed)
    adj = graph.to_adj_lists()
    seeds = [int(s) for s in seeds]
    sets2 = presample(adj, seeds, k2, rng)
    u2 = sorted(set().union(*sets2.values())) if sets2 else []
    layer1_nodes = sorted(set(u2) | (set() if gcn else set(seeds)))
    sets1 = presample(adj, layer1_nodes, k1, rng)
    d0 = table.shape[1]
    w1 = xavier((h1, d0 if gcn else 2 * d0), gen)
    w2 = xavier((h2, h1 if gcn else 2 * h1), gen)
    features, agg1, enc1, agg2, enc2 = build_reference_stack(table, sets1, sets2, w1, w2, gcn, init1, init2)

    with torch.no_grad():
        agg1_out = agg1.forward(layer1_nodes, [sets1[u] for u in layer1_nodes], None)
        enc1_out = enc1(torch.LongTensor(layer1_nodes))
        agg2_out = agg2.forward(seeds, [sets2[s] for s in seeds], None)
        enc2_out = enc2(seeds)
    cot = torch.randn(enc2_out.shape, generator=gen)
    out = enc2(seeds)
    (out * cot).sum().backward()

    nbr2, cnt2 = pad_sets(seeds, sets2, k2)
    nbr1, cnt1 = pad_sets(layer1_nodes, sets1, k1)
    touched = sorted(set(layer1_nodes) | set(int(x) for x in nbr1[nbr1 >= 0]))
    common = dict(
        num_nodes=np.int64(table.shape[0]), d0=np.int64(d0), k1=np.int64(k1), k2=np.int64(k2),
        gcn=np.int64(gcn), sigmoid1=np.int64(init1 in ("node_degree", "shared", "pagerank")),
        sigmoid2=np.int64(init2 in ("node_degree", "shared", "pagerank")),
        seeds=np.array(seeds, dtype=np.int64), nbr2=nbr2, cnt2=cnt2,
        layer1_nodes=np.array(layer1_nodes, dtype=np.int64), nbr1=nbr1, cnt1=cnt1,
        w1=w1.numpy(), w2=w2.numpy(), enc1_out=enc1_out.numpy(), agg2_out=agg2_out.numpy(), enc2_out=enc2_out.numpy(),
        cotangent=cot.numpy(), grad_w1=enc1.weight.grad.numpy(), grad_w2=enc2.weight.grad.numpy())
    if table_spec is None:
        np.savez_compressed(os.path.join(HERE, name + ".npz"), feat_ids=np.array(touched, dtype=np.int64), feat_rows=table[touched].numpy(),
                            agg1_out=agg1_out.numpy(), **common)
    else:
        import hashlib
        kind, tseed = table_spec
        rows = np.sort(np.random.default_rng(seed).choice(len(layer1_nodes), min(16, len(layer1_nodes)), replace=False)).astype(np.int64)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), table_kind=np.int64(kind), table_seed=np.int64(tseed),
                            table_sha256=np.frombuffer(hashlib.sha256(table.numpy().tobytes()).digest(), dtype=np.uint8),
                            agg1_rows=rows, agg1_out_rows=agg1_out.numpy()[rows], **common)
    print(f"{name}: B={len(seeds)} |S1|={len(layer1_nodes)} E1={int(cnt1.sum())} touched={len(touched)} "
          f"enc2_out {tuple(enc2_out.shape)} max|out|={float(enc2_out.abs().max()):.4f}")


def empty_set_cases():
    """aggregators.py:60-61: 0/0 = NaN for an empty set inside a mixed batch;
    an all-empty batch has zero columns and yields zeros."""
    gen = torch.Generator().manual_seed(3)
    table = torch.randn(6, 4, generator=gen)
    features = torch.nn.Embedding(6, 4)
    features.weight = torch.nn.Parameter(table.clone(), requires_grad=False)
    agg = MeanAggregator(features, cuda=False)
    mixed_sets = [{1, 2}, set(), {0}, set()]
    with torch.no_grad():
        mixed = agg.forward([0, 3, 4, 5], mixed_sets, None)
        allempty = agg.forward([3, 5], [set(), set()], None)
        w = xavier((3, 4), gen)
        enc = quiet(Encoder, features, 4, 3, {0: {1, 2}, 3: set(), 4: {0}, 5: set()}, agg,
                    num_sample=None, gcn=True, cuda=False)
        enc.weight.copy_(w)
        enc_mixed = enc([0, 3, 4, 5])
    nbr, cnt = pad_sets([0, 1, 2, 3], {0: {1, 2}, 1: set(), 2: {0}, 3: set()}, 2)
    np.savez_compressed(os.path.join(HERE, "empty_sets.npz"), table=table.numpy(), nodes=np.array([0, 3, 4, 5]),
                        nbr=nbr, cnt=cnt, agg_mixed=mixed.numpy(), agg_all_empty=allempty.numpy(),
                        w=w.numpy(), enc_mixed=enc_mixed.numpy())
    print("empty_sets: mixed NaN rows", np.isnan(mixed.numpy()).all(1).tolist(),
          "all-empty max", float(allempty.abs().max()))


def sampler_stream_case(graph, table):
    """Reference run with its own sampler ON (aggregators.py:42-46) after
    random.seed: pins the restatement's consumption of Python's RNG stream."""
    adj = graph.to_adj_lists()
    gen = torch.Generator().manual_seed(11)
    n, d0 = table.shape
    features = torch.nn.Embedding(n, d0)
    features.weight = torch.nn.Parameter(table.clone(), requires_grad=False)
    w1 = xavier((16, d0), gen)
    w2 = xavier((8, 16), gen)
    agg1 = MeanAggregator(features, cuda=False)
    enc1 = quiet(Encoder, features, d0, 16, adj, agg1, num_sample=5, gcn=True, cuda=False)
    agg2 = MeanAggregator(lambda nodes: enc1(nodes).t(), cuda=False)
    enc2 = quiet(Encoder, lambda nodes: enc1(nodes).t(), 16, 8, adj, agg2, num_sample=4,
                 base_model=enc1, gcn=True, cuda=False)
    with torch.no_grad():
        enc1.weight.copy_(w1)
        enc2.weight.copy_(w2)
        seeds = list(range(0, 40, 3))
        random.seed(2024)
        out = enc2(seeds)
    np.savez_compressed(os.path.join(HERE, "sampler_stream.npz"), rowptr=graph.rowptr, col=graph.col,
                        table=table.numpy(), w1=w1.numpy(), w2=w2.numpy(), seeds=np.array(seeds),
                        k1=np.int64(5), k2=np.int64(4), py_seed=np.int64(2024), enc2_out=out.numpy())
    print("sampler_stream: out", tuple(out.shape))


def tiny_graph():
    """12 nodes: a hub, a chain, two isolated nodes (10, 11), a self loop on 7."""
    edges = [(0, 1), (0, 2), (0, 3), (0, 4), (0, 5), (0, 6), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6),
             (6, 7), (7, 7), (7, 8), (8, 9), (1, 9), (2, 9)]
    src, dst = zip(*edges)
    return G.csr_from_edges(src, dst, 12, symmetric=True)


def cora_embeddings(names):
    idx = {str(n): i for i, n in enumerate(names)}
    with open(os.path.join(REF, "cora/cora.embeddings")) as fp:
        n, d = map(int, fp.readline().split())
        out = np.zeros((len(names), d), dtype=np.float32)
        for line in fp:
            parts = line.split()
            out[idx[parts[0]]] = np.array(parts[1:], dtype=np.float32)
    return torch.from_numpy(out)


def reference_f1_fixture(g_cora):
    """Train the REFERENCE model on the stand-in Cora data (real cora.cites topology + synthesised
    content, sage355/datasets.py), following run_model (model.py:214-259) line for line: seeds,
    10/10/80 split, SGD lr 0.7, 5 epochs (CLI default), the descending `max` batches of model.py:244,
    num_sample 10/10 (the reference's effective fanout).  torch is unseeded in the reference
    (model.py:192-193), so F1 is recorded for five torch seeds."""
    import json
    import time
    from graphsage.model import SupervisedGraphSage
    from sage355.datasets import standin_citation
    from sklearn.metrics import f1_score
    feat_data, labels = standin_citation(g_cora, num_classes=7, feat_dim=1433, seed=0)
    adj_lists = g_cora.to_adj_lists()
    runs = []
    for torch_seed in range(5):
        torch.manual_seed(torch_seed)
        np.random.seed(1)
        random.seed(1)
        features = torch.nn.Embedding(2708, 1433)
        features.weight = torch.nn.Parameter(torch.FloatTensor(feat_data), requires_grad=False)
        agg1 = MeanAggregator(features, cuda=True, feature_dim=100, num_nodes=2708, initializer="None")
        enc1 = quiet(Encoder, features, 1433, 50, adj_lists, agg1, gcn=True, cuda=False, initializer="None")
        agg2 = MeanAggregator(lambda nodes: enc1(nodes).t(), 2708, cuda=False)
        enc2 = quiet(Encoder, lambda nodes: enc1(nodes).t(), enc1.embed_dim, 128, adj_lists, agg2, base_model=enc1, gcn=True, cuda=False)
        model = SupervisedGraphSage(7, enc2)
        rand_indices = np.random.permutation(2708)
        val = rand_indices[270:541]
        train = list(rand_indices[541:])
        opt = torch.optim.SGD(filter(lambda p: p.requires_grad, model.parameters()), lr=0.7)
        times = []
        for _ in range(5):
            random.shuffle(train)
            for batch in range(0, len(train), 128):
                batch_nodes = train[batch:max(len(train), batch + 128)]
                t0 = time.time()
                opt.zero_grad()
                loss = model.loss(batch_nodes, torch.LongTensor(labels[np.array(batch_nodes)]))
                loss.backward()
                opt.step()
                times.append(time.time() - t0)
        out = model.forward(val)
        pred = out.data.numpy().argmax(axis=1)
        runs.append({"torch_seed": torch_seed, "f1_micro": float(f1_score(labels[val], pred, average="micro")),
                     "f1_macro": float(f1_score(labels[val], pred, average="macro")), "mean_batch_time": float(np.mean(times))})
        print("reference F1 run", runs[-1])
    fixture = {"dataset": "cora.cites topology + sage355.datasets.standin_citation(num_classes=7, feat_dim=1433, seed=0)",
               "config": {"epochs": 5, "batch_size": 128, "ref_batching": True, "lr": 0.7, "seed": 1, "hidden": [50, 128],
                          "num_sample": [10, 10], "gcn": True},
               "runs": runs,
               "f1_micro_mean": float(np.mean([r["f1_micro"] for r in runs])), "f1_micro_std": float(np.std([r["f1_micro"] for r in runs])),
               "f1_macro_mean": float(np.mean([r["f1_macro"] for r in runs])), "f1_macro_std": float(np.std([r["f1_macro"] for r in runs]))}
    with open(os.path.join(HERE, "reference_f1_cora_standin.json"), "w") as fp:
        json.dump(fixture, fp, indent=1)
    print("reference F1 micro %.4f +- %.4f, macro %.4f +- %.4f" % (fixture["f1_micro_mean"], fixture["f1_micro_std"],
                                                                  fixture["f1_macro_mean"], fixture["f1_macro_std"]))


def reference_f1_fixture_pubmed(g_pub):
    """The same for Pubmed (run_model("pubmed", ...), model.py:184-259): real Pubmed-Diabetes topology (19717 nodes, 3 classes)
    + synthesised 500-dim content (the .NODE.paper.tab file is not in the checkout), hidden 50 / 128, gcn encoders, and the
    reference's EFFECTIVE fanout 10 / 10 (its `enc.num_samples = 10 / 25` assignments are no-ops: Encoder reads `num_sample`,
    model.py:223-224).  Bounded so that it finishes on CPU: ONE epoch of the reference's descending `max` batches with
    batch_size 1024 (16 steps over 15774 .. 414 nodes), three torch seeds."""
    import json
    import time
    from graphsage.model import SupervisedGraphSage
    from sage355.datasets import standin_citation
    from sklearn.metrics import f1_score
    n = g_pub.num_nodes
    feat_data, labels = standin_citation(g_pub, num_classes=3, feat_dim=500, seed=0)
    adj_lists = g_pub.to_adj_lists()
    runs = []
    for torch_seed in range(3):
        torch.manual_seed(torch_seed)
        np.random.seed(1)
        random.seed(1)
        features = torch.nn.Embedding(n, 500)
        features.weight = torch.nn.Parameter(torch.FloatTensor(feat_data), requires_grad=False)
        agg1 = MeanAggregator(features, cuda=True, feature_dim=100, num_nodes=n, initializer="None")
        enc1 = quiet(Encoder, features, 500, 50, adj_lists, agg1, gcn=True, cuda=False, initializer="None")
        agg2 = MeanAggregator(lambda nodes: enc1(nodes).t(), n, cuda=False)
        enc2 = quiet(Encoder, lambda nodes: enc1(nodes).t(), enc1.embed_dim, 128, adj_lists, agg2, base_model=enc1, gcn=True, cuda=False)
        model = SupervisedGraphSage(3, enc2)
        rand_indices = np.random.permutation(n)
        val = rand_indices[int(0.1 * n):int(0.2 * n)]
        train = list(rand_indices[int(0.2 * n):])
        opt = torch.optim.SGD(filter(lambda p: p.requires_grad, model.parameters()), lr=0.7)
        times = []
        for _ in range(1):
            random.shuffle(train)
            for batch in range(0, len(train), 1024):
                batch_nodes = train[batch:max(len(train), batch + 1024)]
                t0 = time.time()
                opt.zero_grad()
                loss = model.loss(batch_nodes, torch.LongTensor(labels[np.array(batch_nodes)]))
                loss.backward()
                opt.step()
                times.append(time.time() - t0)
        with torch.no_grad():
            out = model.forward(val)
        pred = out.data.numpy().argmax(axis=1)
        runs.append({"torch_seed": torch_seed, "f1_micro": float(f1_score(labels[val], pred, average="micro")),
                     "f1_macro": float(f1_score(labels[val], pred, average="macro")), "mean_batch_time": float(np.mean(times))})
        print("reference Pubmed F1 run", runs[-1], flush=True)
    fixture = {"dataset": "Pubmed-Diabetes.DIRECTED.cites.tab topology + sage355.datasets.standin_citation(num_classes=3, feat_dim=500, seed=0)",
               "config": {"epochs": 1, "batch_size": 1024, "ref_batching": True, "lr": 0.7, "seed": 1, "hidden": [50, 128],
                          "num_sample": [10, 10], "gcn": True},
               "runs": runs,
               "f1_micro_mean": float(np.mean([r["f1_micro"] for r in runs])), "f1_micro_std": float(np.std([r["f1_micro"] for r in runs])),
               "f1_macro_mean": float(np.mean([r["f1_macro"] for r in runs])), "f1_macro_std": float(np.std([r["f1_macro"] for r in runs]))}
    with open(os.path.join(HERE, "reference_f1_pubmed_standin.json"), "w") as fp:
        json.dump(fixture, fp, indent=1)
    print("reference Pubmed F1 micro %.4f +- %.4f, macro %.4f +- %.4f" % (fixture["f1_micro_mean"], fixture["f1_micro_std"],
                                                                         fixture["f1_macro_mean"], fixture["f1_macro_std"]))


def reference_f1_streams(g, name, num_classes, feat_dim, epochs, batch_size, runs=24):
    """the reference's F1 over MANY (sampling stream, weight initialisation) pairs on ONE split -- what this build's
    F1 test varies too (the device sampler cannot replay Python's `random`, so only distributions can be compared).  run_model
    (model.py:192-259) line for line as in reference_f1_fixture, with np.random.seed(1) (the split), random.seed(1000 + i) (shuffles +
    neighbour sampling) and torch.manual_seed(i) (weights) for run i."""
    import json
    import time
    from graphsage.model import SupervisedGraphSage
    from sage355.datasets import standin_citation
    from sklearn.metrics import f1_score
    n = g.num_nodes
    feat_data, labels = standin_citation(g, num_classes=num_classes, feat_dim=feat_dim, seed=0)
    adj_lists = g.to_adj_lists()
    out = []
    for i in range(runs):
        torch.manual_seed(i)
        np.random.seed(1)
        random.seed(1000 + i)
        features = torch.nn.Embedding(n, feat_dim)
        features.weight = torch.nn.Parameter(torch.FloatTensor(feat_data), requires_grad=False)
        agg1 = MeanAggregator(features, cuda=True, feature_dim=100, num_nodes=n, initializer="None")
        enc1 = quiet(Encoder, features, feat_dim, 50, adj_lists, agg1, gcn=True, cuda=False, initializer="None")
        agg2 = MeanAggregator(lambda nodes: enc1(nodes).t(), n, cuda=False)
        enc2 = quiet(Encoder, lambda nodes: enc1(nodes).t(), enc1.embed_dim, 128, adj_lists, agg2, base_model=enc1, gcn=True, cuda=False)
        model = SupervisedGraphSage(num_classes, enc2)
        rand_indices = np.random.permutation(n)
        val = rand_indices[int(0.1 * n):int(0.2 * n)]
        train = list(rand_indices[int(0.2 * n):])
        opt = torch.optim.SGD(filter(lambda p: p.requires_grad, model.parameters()), lr=0.7)
        t0 = time.time()
        for _ in range(epochs):
            random.shuffle(train)
            for batch in range(0, len(train), batch_size):
                batch_nodes = train[batch:max(len(train), batch + batch_size)]
                opt.zero_grad()
                loss = model.loss(batch_nodes, torch.LongTensor(labels[np.array(batch_nodes)]))
                loss.backward()
                opt.step()
        with torch.no_grad():
            pred = model.forward(val).data.numpy().argmax(axis=1)
        out.append({"run": i, "f1_micro": float(f1_score(labels[val], pred, average="micro")),
                    "f1_macro": float(f1_score(labels[val], pred, average="macro")), "seconds": round(time.time() - t0, 1)})
        print(name, "reference F1 run", out[-1], flush=True)
    mi, ma = np.array([r["f1_micro"] for r in out]), np.array([r["f1_macro"] for r in out])
    fixture = {"dataset": f"{name} topology + sage355.datasets.standin_citation(num_classes={num_classes}, feat_dim={feat_dim}, seed=0)",
               "config": {"epochs": epochs, "batch_size": batch_size, "ref_batching": True, "lr": 0.7, "split_seed": 1,
                          "sample_seed": "1000 + run", "torch_seed": "run", "hidden": [50, 128], "num_sample": [10, 10], "gcn": True},
               "runs": out,
               "f1_micro_mean": float(mi.mean()), "f1_micro_std": float(mi.std(ddof=1)), "f1_micro_se": float(mi.std(ddof=1) / np.sqrt(len(mi))),
               "f1_macro_mean": float(ma.mean()), "f1_macro_std": float(ma.std(ddof=1)), "f1_macro_se": float(ma.std(ddof=1) / np.sqrt(len(ma))),
               "f1_macro_min": float(ma.min()), "f1_macro_max": float(ma.max())}
    with open(os.path.join(HERE, f"reference_f1_{name}_standin_streams.json"), "w") as fp:
        json.dump(fixture, fp, indent=1)
    print(name, "reference F1 over %d (stream, init) pairs: micro %.4f +- %.4f (SE %.4f), macro %.4f +- %.4f" %
          (len(out), fixture["f1_micro_mean"], fixture["f1_micro_std"], fixture["f1_micro_se"], fixture["f1_macro_mean"], fixture["f1_macro_std"]))


def batch_size_cases(g_cora, g_pub):
    """BASELINE configs[0] / configs[1] at their batch size: B = 256 seeds, Cora 1433 -> 50 -> 128 with the fanout the
    reference really runs (10 / 10: model.py:223-224 writes `num_samples`, encoders.py:23 reads `num_sample`) and the 5 / 5 the config
    names; Pubmed 500 -> 50 -> 128, fanout 10 / 25; both encoder modes.  Generators of their own: the small cases' streams stay as they were."""
    inv = {v: k for k, v in TABLE_KINDS.items()}
    bow = torch.from_numpy(synth_table("bow", 2708, 1433, 20241))
    for i, (k1, k2, gcn) in enumerate([(10, 10, True), (10, 10, False), (5, 5, True), (5, 5, False)]):
        seeds = np.random.default_rng(100 + i).choice(2708, 256, replace=False)
        two_layer_case(f"cora_{'gcn' if gcn else 'concat'}_{k1}_{k2}_b256", g_cora, bow, seeds, k1, k2, 50, 128, gcn, 20 + i,
                       table_spec=(inv["bow"], 20241))
    tfidf = torch.from_numpy(synth_table("tfidf", 19717, 500, 20242))
    for i, gcn in enumerate([True, False]):
        seeds = np.random.default_rng(200 + i).choice(19717, 256, replace=False)
        two_layer_case(f"pubmed_{'gcn' if gcn else 'concat'}_10_25_b256", g_pub, tfidf, seeds, 10, 25, 50, 128, gcn, 30 + i,
                       table_spec=(inv["tfidf"], 20242))


def main():
    if "--b256-only" in sys.argv:
        g_cora, _ = G.read_edge_list(os.path.join(REF, "cora/cora.cites"))
        g_pub, _ = G.read_edge_list(os.path.join(REF, "pubmed-data/Pubmed-Diabetes.DIRECTED.cites.tab"), fmt="pubmed")
        batch_size_cases(g_cora, g_pub)
        return
    if "--f1-streams-cora" in sys.argv:
        g_cora, _ = G.read_edge_list(os.path.join(REF, "cora/cora.cites"))
        reference_f1_streams(g_cora, "cora", 7, 1433, epochs=5, batch_size=128)
        return
    if "--f1-streams-pubmed" in sys.argv:
        g_pub, _ = G.read_edge_list(os.path.join(REF, "pubmed-data/Pubmed-Diabetes.DIRECTED.cites.tab"), fmt="pubmed")
        reference_f1_streams(g_pub, "pubmed", 3, 500, epochs=1, batch_size=1024, runs=12)
        return
    if "--f1-pubmed-only" in sys.argv:
        g_pub, _ = G.read_edge_list(os.path.join(REF, "pubmed-data/Pubmed-Diabetes.DIRECTED.cites.tab"), fmt="pubmed")
        reference_f1_fixture_pubmed(g_pub)
        return
    g_tiny = tiny_graph()
    gen = torch.Generator().manual_seed(0)
    t_tiny = torch.randn(12, 8, generator=gen)
    conn = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9]
    two_layer_case("tiny_gcn", g_tiny, t_tiny, conn, 3, 2, 6, 5, True, 1)
    two_layer_case("tiny_concat", g_tiny, t_tiny, conn, 3, 2, 6, 5, False, 2)
    two_layer_case("tiny_sigmoid", g_tiny, t_tiny, conn, 3, 2, 6, 5, True, 3, init1="shared", init2="pagerank")
    empty_set_cases()

    g_cora, names = G.read_edge_list(os.path.join(REF, "cora/cora.cites"))
    assert g_cora.num_nodes == 2708
    emb = cora_embeddings(names)
    rs = np.random.default_rng(5)
    two_layer_case("cora_emb_gcn_5_5", g_cora, emb, rs.choice(2708, 64, replace=False), 5, 5, 50, 128, True, 4)
    two_layer_case("cora_emb_concat_10_10", g_cora, emb, rs.choice(2708, 48, replace=False), 10, 10, 50, 128, False, 5)
    bow = torch.from_numpy((rs.random((2708, 1433)) < 18.0 / 1433).astype(np.float32))
    two_layer_case("cora_bow_gcn_5_5", g_cora, bow, rs.choice(2708, 32, replace=False), 5, 5, 50, 128, True, 6)
    two_layer_case("cora_bow_concat_5_5", g_cora, bow, rs.choice(2708, 16, replace=False), 5, 5, 50, 128, False, 7)
    sampler_stream_case(g_cora, emb[:, :16].contiguous())
    np.savez_compressed(os.path.join(HERE, "cora_topology.npz"), rowptr=g_cora.rowptr, col=g_cora.col)
    if "--f1" in sys.argv or not os.path.exists(os.path.join(HERE, "reference_f1_cora_standin.json")):
        reference_f1_fixture(g_cora)

    g_pub, _ = G.read_edge_list(os.path.join(REF, "pubmed-data/Pubmed-Diabetes.DIRECTED.cites.tab"), fmt="pubmed")
    assert g_pub.num_nodes == 19717, g_pub.num_nodes
    np.savez_compressed(os.path.join(HERE, "pubmed_topology.npz"), rowptr=g_pub.rowptr, col=g_pub.col)
    tfidf = torch.from_numpy((rs.random((19717, 500)) * (rs.random((19717, 500)) < 0.1)).astype(np.float32))
    two_layer_case("pubmed_gcn_10_25", g_pub, tfidf, rs.choice(19717, 6, replace=False), 10, 25, 50, 128, True, 8)
    two_layer_case("pubmed_concat_10_25", g_pub, tfidf, rs.choice(19717, 4, replace=False), 10, 25, 50, 128, False, 9)
    if "--f1" in sys.argv or not os.path.exists(os.path.join(HERE, "reference_f1_pubmed_standin.json")):
        reference_f1_fixture_pubmed(g_pub)
    batch_size_cases(g_cora, g_pub)


if __name__ == "__main__":
    main()
