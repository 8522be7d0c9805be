"""Training harness on the MI355X vs the reference's F1 on the same stand-in datasets
(tests/golden/reference_f1_<dataset>_standin_streams.json: the REFERENCE trained here on the real cora.cites / Pubmed
topology + synthesised content over 24 / 12 (sampling stream, weight initialisation) pairs).  BASELINE.json: "Cora/Pubmed
F1 within +-0.5 of the CPU reference" -- the real content files are not available offline, so
this is the closest pinned statement."""
import json
import os
import random

import numpy as np
import pytest
import torch

from sage355.datasets import standin_citation
from sage355.graph import CSRGraph
from sage355.train import run_training
from util import GOLDEN_DIR

pytestmark = pytest.mark.gpu


def cora():
    z = np.load(os.path.join(GOLDEN_DIR, "cora_topology.npz"))
    g = CSRGraph(z["rowptr"], z["col"], len(z["rowptr"]) - 1)
    feats, labels = standin_citation(g, num_classes=7, feat_dim=1433, seed=0)
    return g, feats, labels


def _f1_runs(path, graph, feats, labels, num_classes, runs, epochs, batch_size):
    """path "engine": EngineTrainer (everything on the device); "modules": the reference's class surface on the device (run_training:
    Encoder / MeanAggregator / SupervisedGraphSage, the two-hop forward and backward as one autograd node over the same engine);
    "dropin": the reference's loop over the shim classes with cuda=False (host classifier / loss / SGD, as model.py runs)."""
    from sage355.train import run_engine_training
    micro, macro = [], []
    adj = graph.to_adj_lists() if path != "engine" else None
    for i in range(runs):
        torch.manual_seed(i)                          # weight initialisation, as the fixture's run i
        if path == "engine":
            res = run_engine_training(graph, feats, labels, num_classes, seed=1, epochs=epochs, batch_size=batch_size, ref_batching=True,
                                      sample_seed=1000 + i, hidden1=50, hidden2=128, num_sample1=10, num_sample2=10, gcn=True)
        elif path == "dropin":
            from test_gpu_round3 import _reference_loop       # model.py:192-259 restated over the shim classes, cuda=False
            f1, _, losses, _, f1_macro = _reference_loop(feats, labels, adj, num_classes, 1, 1000 + i, epochs, batch_size, True, lr=0.7, with_macro=True)
            res = {"f1_micro": f1, "f1_macro": f1_macro, "losses": losses}
        else:
            res = run_training(feats, labels, adj, num_classes, seed=1, sample_seed=1000 + i, epochs=epochs, batch_size=batch_size,
                               ref_batching=True, lr=0.7, verbose=False, hidden1=50, hidden2=128, num_sample1=10, num_sample2=10, gcn=True)
        assert res["losses"][-1] < 0.6 * res["losses"][0]
        micro.append(res["f1_micro"])
        macro.append(res["f1_macro"])
    return np.array(micro), np.array(macro)


@pytest.mark.parametrize("dataset,path", [("cora", "engine"), ("cora", "modules"), ("cora", "dropin"), ("pubmed", "engine")])
def test_f1_distribution_over_sampling_streams_matches_the_reference(dataset, path):
    """"F1 within +-0.5 of the CPU reference", settled over MANY sampling streams on both sides.  The device sampler
    cannot replay Python's `random`, so what can be compared is the F1 DISTRIBUTION over (sampling stream, weight initialisation)
    pairs on one split: the reference's own (tests/golden/reference_f1_<dataset>_standin_streams.json: run_model line for line,
    np.random.seed(1), random.seed(1000 + i), torch.manual_seed(i); 24 runs on Cora, 12 on Pubmed) against EngineTrainer with the same
    split, epochs, descending batches (model.py:244), lr, fanout 10 / 10 and as many streams.
    Bars: |mean micro F1 - reference mean| <= 0.5 points + 2 standard errors of the difference's SMALLER side; mean macro F1 (model.py:258
    prints it too) within one standard deviation of the reference's single runs; spreads of the same order.
    (Round 3 compared six streams with a fixture of five runs that shared ONE stream, random.seed(1): 0.9373 +- 0.0033.  Over 24 streams
    the reference itself reads 0.9311 +- 0.0069 on stand-in Cora: that fixture sat on a lucky stream, there was no offset to explain.)"""
    ref = json.load(open(os.path.join(GOLDEN_DIR, f"reference_f1_{dataset}_standin_streams.json")))
    if dataset == "cora":
        g, feats, labels = cora()
        classes = 7
    else:
        z = np.load(os.path.join(GOLDEN_DIR, "pubmed_topology.npz"))
        g = CSRGraph(z["rowptr"], z["col"], len(z["rowptr"]) - 1)
        feats, labels = standin_citation(g, num_classes=3, feat_dim=500, seed=0)
        classes = 3
    cfg = ref["config"]
    runs = len(ref["runs"])
    micro, macro = _f1_runs(path, g, feats, labels, classes, runs, cfg["epochs"], cfg["batch_size"])
    se_mine = float(micro.std(ddof=1) / np.sqrt(runs))
    tol = 0.005 + 2 * min(se_mine, ref["f1_micro_se"])
    d_micro = float(micro.mean()) - ref["f1_micro_mean"]
    d_macro = float(macro.mean()) - ref["f1_macro_mean"]
    print(f"{dataset} ({path}): F1 micro {micro.mean():.4f} +- {micro.std(ddof=1):.4f} (SE {se_mine:.4f}) vs reference {ref['f1_micro_mean']:.4f} +- "
          f"{ref['f1_micro_std']:.4f} (SE {ref['f1_micro_se']:.4f}): delta {d_micro:+.4f}, bar {tol:.4f}; macro {macro.mean():.4f} +- {macro.std(ddof=1):.4f} vs "
          f"{ref['f1_macro_mean']:.4f} +- {ref['f1_macro_std']:.4f}: delta {d_macro:+.4f}, {runs} streams each")
    assert abs(d_micro) <= tol, (d_micro, tol)
    assert abs(d_macro) <= ref["f1_macro_std"], (d_macro, ref["f1_macro_std"])
    assert micro.std(ddof=1) <= 2.0 * ref["f1_micro_std"] + 0.002


def test_plain_batching_and_eval_fast_path():
    """128-node batches (no model.py:244 quirk), device sampler at layer 1, fused 2-hop engine for the
    validation forward; loss must fall and F1 must clear the majority-class rate by a wide margin."""
    g, feats, labels = cora()
    torch.manual_seed(0)
    res = run_training(feats, labels, g.to_adj_lists(), 7, seed=1, epochs=3, batch_size=128, ref_batching=False, verbose=False)
    assert np.mean(res["losses"][-5:]) < 0.6 * np.mean(res["losses"][:5])
    assert res["f1_micro"] > 0.85


def _train_rank(rank, world, port, tmp):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import io
    from contextlib import redirect_stdout
    from sage355 import dist
    dist.init_from_env(backend="gloo")          # both ranks share cuda:0 here; on a node it is nccl, one GPU per rank
    g, feats, labels = cora()
    torch.manual_seed(0)                        # same initial weights; broadcast_params makes that explicit anyway
    with redirect_stdout(io.StringIO()):
        res = run_training(feats, labels, g.to_adj_lists(), 7, seed=1, sample_seed=50 + rank, epochs=2, batch_size=256,
                           verbose=False, return_model=True)
    w = torch.cat([p.detach().reshape(-1).cpu() for p in res["model"].parameters() if p.requires_grad])
    torch.save({"w": w, "f1": res["f1_micro"], "first": res["losses"][0], "last": res["losses"][-1]}, os.path.join(tmp, f"r{rank}.pt"))
    dist.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_seed_sharded_training_keeps_replicas_identical(tmp_path):
    """Each rank embeds its shard of every mini-batch (own sampler stream) and the weight gradients are summed
    with one all-reduce per step: replicas must stay bit-identical and still learn."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_train_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(a["w"], b["w"]), "replicas diverged"
    assert a["last"] < 0.6 * a["first"] and a["f1"] > 0.8
