"""Debug: the pre-transformed-table pipeline with the express lane (bench.py's variant produced NaN).  Per batch: lane mix, NaN count, equality
with the single-stream forward."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "graphsage-simple_amd"))
import numpy as np, torch
from sage355.engine import TwoHopEngine, RolePipeline, pretransform_table
from sage355.graph import rmat_graph
DEV = torch.device("cuda:0")
scale, edges = int(sys.argv[1]), int(sys.argv[2])
threads = sys.argv[3] == "1"
graph = rmat_graph(scale, edges, seed=0)
gen = torch.Generator().manual_seed(0)
d0, h1, h2, b, k1, k2, n = 256, 128, 128, 4096, 15, 25, 30
table = torch.randn(graph.num_nodes, d0, generator=gen).to(DEV)
w1 = ((torch.rand(h1, d0, generator=gen) * 2 - 1) * np.sqrt(6.0 / (h1 + d0))).to(DEV)
w2 = ((torch.rand(h2, h1, generator=gen) * 2 - 1) * np.sqrt(6.0 / (h2 + h1))).to(DEV)
rowptr, col = graph.to(DEV)
cand = np.nonzero(graph.degrees() > 0)[0]
rs = np.random.default_rng(1)
seeds = torch.from_numpy(np.stack([rs.choice(cand, b, replace=False) for _ in range(n)]).astype(np.int32)).to(DEV)
keys = [77 + i for i in range(n)]
for name, (tab, wa) in {"plain": (table, w1), "pretransformed": pretransform_table(table, w1)}.items():
    single = TwoHopEngine(rowptr, col, tab, wa, w2, k1, k2, max_batch=b)
    want = torch.stack([single.forward(seeds[i], seed=keys[i]).clone() for i in range(n)])
    torch.cuda.synchronize()
    pipe = RolePipeline(rowptr, col, tab, wa, w2, k1, k2, batch=b, depth=4, threads=threads)
    out = torch.zeros(n, b, h2, device=DEV)
    for rnd in range(3):
        out.zero_()
        c0 = pipe.express_count
        if rnd == 0:
            pipe.submit_many(seeds, keys, out)
        else:
            for i in range(n):
                pipe.submit(seeds[i], keys[i], out[i])
        pipe.synchronize()
        bad = [(i, int(torch.isnan(out[i]).sum()), bool(torch.equal(out[i], want[i]))) for i in range(n)]
        print(name, "threads", threads, "round", rnd, "express", pipe.express_count - c0, "nan in want", int(torch.isnan(want).sum()),
              "mismatching batches", [(i, nn) for i, nn, eq in bad if not eq])
