"""Training step time of the harness (sage355.train.run_training) on the stand-in Cora, next to the reference's 0.14-0.18 s
per step on this container's CPU."""
import os, sys, json
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "graphsage-simple_amd")); sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import numpy as np, torch
from sage355.datasets import standin_citation
from sage355.graph import CSRGraph
from sage355.train import run_training
z = np.load(os.path.join(HERE, "..", "tests", "golden", "cora_topology.npz"))
g = CSRGraph(z["rowptr"], z["col"], len(z["rowptr"]) - 1)
feats, labels = standin_citation(g, num_classes=7, feat_dim=1433, seed=0)
adj = g.to_adj_lists()
for bs, refb in ((256, True), (128, False), (512, False)):
    torch.manual_seed(0)
    res = run_training(feats, labels, adj, 7, seed=1, epochs=2, batch_size=bs, ref_batching=refb, verbose=False, hidden1=50, hidden2=128,
                       num_sample1=10, num_sample2=10, gcn=True)
    t = np.array(res.get("times", [])) if "times" in res else None
    print(f"batch {bs} ref_batching={refb}: mean step {1e3 * res['mean_batch_time']:.2f} ms, F1 micro {res['f1_micro']:.3f}, steps {len(res['losses'])}")
