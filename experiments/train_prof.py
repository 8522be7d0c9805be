"""Kernel breakdown of one EngineTrainer step at config-3 size: run under rocprofv3 --kernel-trace --stats."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "graphsage-simple_amd")]
import numpy as np, torch
from sage355.train import EngineTrainer
from sage355.graph import rmat_graph
dev = torch.device("cuda", 0)
g = rmat_graph(20, 16_000_000, seed=0, cache_dir="/tmp/sage_cache")
gen = torch.Generator(device=dev).manual_seed(0)
table = torch.randn(g.num_nodes, 256, generator=gen, device=dev)
labels = torch.randint(0, 16, (g.num_nodes,), device=dev)
rowptr, col = g.to(dev)
cand = np.nonzero(g.degrees() > 0)[0]
rs = np.random.default_rng(1)
tr = EngineTrainer(rowptr, col, table, 16, hidden1=128, hidden2=128, num_sample1=15, num_sample2=25, gcn=True, lr=0.05, max_batch=4096, relabel="degree")
seeds = [torch.from_numpy(rs.choice(cand, 4096, replace=False).astype(np.int32)).to(dev) for _ in range(40)]
for i in range(40):
    tr.step(seeds[i], labels[seeds[i].long()], key=i)
torch.cuda.synchronize()
