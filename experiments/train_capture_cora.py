"""EngineTrainer on the stand-in Cora (1433 -> 50 -> 128, fanout 10/10, 256 seeds): eager step vs the step captured as one hipGraph."""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "graphsage-simple_amd"))
import numpy as np, torch
from sage355.datasets import standin_citation
from sage355.graph import CSRGraph
from sage355.train import EngineTrainer
dev = "cuda"
z = np.load(os.path.join(HERE, "..", "tests", "golden", "cora_topology.npz"))
g = CSRGraph(z["rowptr"], z["col"], len(z["rowptr"]) - 1)
feats, labels = standin_citation(g, num_classes=7, feat_dim=1433, seed=0)
table = torch.from_numpy(feats).to(dev)
labels_by_node = torch.from_numpy(labels.reshape(-1)).to(dev)
rowptr, col = g.to(dev)
rs = np.random.default_rng(0)
for b in (256, 1024):
    ring = torch.from_numpy(np.stack([rs.choice(g.num_nodes, b, replace=False) for _ in range(16)]).astype(np.int32)).to(dev)
    keys = list(range(16))
    torch.manual_seed(0)
    tr = EngineTrainer(rowptr, col, table, 7, hidden1=50, hidden2=128, num_sample1=10, num_sample2=10, gcn=True, lr=0.7, max_batch=b)
    for i in range(8):
        tr.step(ring[i], labels_by_node[ring[i].long()], keys[i])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(64):
        tr.step(ring[i % 16], labels_by_node[ring[i % 16].long()], keys[i % 16])
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 64
    loss = tr.capture_step(ring, keys, labels_by_node)
    for _ in range(8): tr.replay_step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(64): tr.replay_step()
    torch.cuda.synchronize(); cap = (time.perf_counter() - t0) / 64
    print(f"batch {b}: eager step {eager * 1e3:.3f} ms, captured step {cap * 1e3:.3f} ms ({eager / cap:.2f}x); loss {float(loss):.4f}", flush=True)
