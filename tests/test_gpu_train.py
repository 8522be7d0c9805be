"""Training harness on the MI355X vs the reference's F1 on the same stand-in dataset
(tests/golden/reference_f1_cora_standin.json: the REFERENCE trained here on cora.cites topology +
synthesised content, five torch seeds).  BASELINE.json: "Cora F1 within +-0.5 of the CPU reference"
-- the real content files are not available offline, so this is the
closest pinned statement; the tolerance below is 0.5 points plus twice the reference's own
seed-to-seed spread."""
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


def test_training_reaches_reference_f1_on_standin_cora():
    """Same split, optimiser, epochs and batching as the reference run; six different sampling streams /
    weight initialisations.  The device sampler cannot replay Python's `random`, so the comparison is
    between MEANS: |mean F1 - reference mean| <= 0.5 points + 2 standard errors (measured spread over
    sampling streams is ~0.9 points, the reference's five runs share ONE stream and spread 0.3)."""
    ref = json.load(open(os.path.join(GOLDEN_DIR, "reference_f1_cora_standin.json")))
    g, feats, labels = cora()
    adj = g.to_adj_lists()
    cfg = ref["config"]
    micro = []
    for run in range(6):
        torch.manual_seed(run)
        res = run_training(feats, labels, adj, 7, seed=cfg["seed"], sample_seed=1000 + run, epochs=cfg["epochs"],
                           batch_size=cfg["batch_size"], ref_batching=True, lr=cfg["lr"], verbose=False, hidden1=50,
                           hidden2=128, num_sample1=10, num_sample2=10, gcn=True)
        assert res["losses"][-1] < 0.5 * res["losses"][0]
        micro.append(res["f1_micro"])
    mine, spread = float(np.mean(micro)), float(np.std(micro))
    tol = 0.005 + 2 * float(np.sqrt(spread ** 2 / len(micro) + ref["f1_micro_std"] ** 2 / len(ref["runs"])))
    print(f"F1 micro: this build {mine:.4f} +- {spread:.4f} (runs {[round(m, 4) for m in micro]}), "
          f"reference {ref['f1_micro_mean']:.4f} +- {ref['f1_micro_std']:.4f}, tolerance {tol:.4f}")
    assert abs(mine - ref["f1_micro_mean"]) <= tol, (mine, ref["f1_micro_mean"], tol)


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
