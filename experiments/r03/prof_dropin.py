"""Where does a step of the reference-shaped training loop with the drop-in classes go?  (cProfile, stand-in Cora, 256-seed steps)"""
import cProfile, os, pstats, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (REPO, os.path.join(REPO, "graphsage-simple_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from test_gpu_round3 import _reference_loop
from sage355.datasets import standin_citation
from sage355.graph import CSRGraph
z = np.load(os.path.join(REPO, "tests", "golden", "cora_topology.npz"))
g = CSRGraph(z["rowptr"], z["col"], len(z["rowptr"]) - 1)
feats, labels = standin_citation(g, num_classes=7, feat_dim=1433, seed=0)
adj = g.to_adj_lists()
torch.manual_seed(0)
_reference_loop(feats, labels, adj, 7, 1, 1, 1, 256, False)           # warm
pr = cProfile.Profile()
pr.enable()
f1, times, losses, _ = _reference_loop(feats, labels, adj, 7, 1, 1, 3, 256, False)
pr.disable()
print("mean step %.3f ms, median %.3f ms, steps %d" % (np.mean(times) * 1e3, np.median(times) * 1e3, len(times)))
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
