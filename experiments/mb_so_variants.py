"""kernel-only durations of the outer sampler variants: run under rocprofv3 --kernel-trace and read the trace."""
import sys, os
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "graphsage-simple_amd"))
import numpy as np, torch
from sage355 import ops
from sage355.graph import rmat_graph
dev = "cuda"
g = rmat_graph(20, 16_000_000, cache_dir="/tmp/sage_cache")
rowptr, col = g.to(dev)
deg = g.degrees(); cands = np.nonzero(deg > 0)[0]
seeds = torch.from_numpy(np.random.default_rng(0).choice(cands, 4096, replace=False).astype(np.int32)).to(dev)
flag = torch.zeros(1, dtype=torch.int32, device=dev)
fr = ops.Frontier(4096 * 26, dev)
for rep in range(6):
    for k in (5, 25):
        ops.sample_neighbors(rowptr, col, seeds, k, seed=rep, tag=2)                                    # no frontier
        ops.sample_neighbors(rowptr, col, seeds, k, seed=rep, tag=2, any_nonempty=flag)               # + flag
        fr.reset(0)
        ops.sample_neighbors(rowptr, col, seeds, k, seed=rep, tag=2, frontier=fr, any_nonempty=flag)  # + frontier
    torch.cuda.synchronize()
