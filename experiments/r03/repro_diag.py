"""Which piece of the training step is not bitwise reproducible?  Same weights, same batch, same key, several calls."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "graphsage-simple_amd")]
import numpy as np, torch
from sage355.graph import rmat_graph
from sage355.train import EngineTrainer
DEV = "cuda"
graph = rmat_graph(14, 300_000, seed=4, accel=None)
rowptr, col = graph.to(DEV)
cand = np.nonzero(graph.degrees() > 0)[0]
labels_by_node = torch.from_numpy(np.random.default_rng(3).integers(0, 5, graph.num_nodes)).to(DEV)
for gcn, relabel, hidden1, d0 in [(True, None, 64, 128), (False, "degree", 64, 128), (True, "degree", 30, 66), (True, None, 128, 256)]:
    table = torch.randn(graph.num_nodes, d0, generator=torch.Generator().manual_seed(1)).to(DEV)
    b = 1024 if d0 == 256 else 256
    torch.manual_seed(5)
    tr = EngineTrainer(rowptr, col, table, 5, hidden1=hidden1, hidden2=32, num_sample1=7, num_sample2=15 if d0 == 256 else 9, gcn=gcn, lr=0.3, max_batch=b, relabel=relabel)
    for trial in range(3):
        ids = torch.from_numpy(np.random.default_rng(10 + trial).choice(cand, b, replace=False).astype(np.int32)).to(DEV)
        res = []
        for rep in range(4):
            loss, (g1, g2, gc) = tr.grads(ids, labels_by_node[ids.long()], 101 + trial)
            e = tr.engine
            nl = int(e._bwd["nlive"])
            gh = e._bwd["grad_h1"][:nl].clone()
            L = e.layout
            h1 = e._view(L.h1, L.max_s1 * e.h1p, torch.float32).view(L.max_s1, e.h1p)[:nl].clone()
            a1 = (e._view(L.agg1, L.max_s1 * e.d0p, torch.float32).view(L.max_s1, e.d0p) if L.layer1_split else e._bwd["agg1"])[:nl].clone()
            out = tr.engine.forward(ids, seed=101 + trial).clone()
            res.append((loss.clone(), g1.clone(), g2.clone(), gc.clone(), out, gh, h1, a1))
        eq = [[bool(torch.equal(res[0][j], res[r][j])) for j in range(8)] for r in range(1, 4)]
        print(f"gcn={gcn} relabel={relabel} h1={hidden1} d0={d0} nlive={nl} trial {trial}: (loss, g_w1, g_w2, g_cls, out, grad_h1, h1, agg1) equal to call 0: {eq[0]} {eq[1]}", flush=True)
