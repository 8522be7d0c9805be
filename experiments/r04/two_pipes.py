"""Two role pipelines on eight streams, alternate batches: do one pipeline's kernels fill the other's hand-off gaps?
    python experiments/r04/two_pipes.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(R, "graphsage-simple_amd")); sys.path.insert(0, R)
import numpy as np, torch
from sage355.engine import RolePipeline
from sage355.graph import rmat_graph
dev = torch.device("cuda", 0)
graph = rmat_graph(20, 16_000_000, seed=0)
gen = torch.Generator(device=dev).manual_seed(0)
n, d0, h1, h2, b, k1, k2 = graph.num_nodes, 256, 128, 128, 4096, 15, 25
table = torch.randn(n, d0, generator=gen, device=dev)
wg = torch.Generator().manual_seed(0)
w1 = ((torch.rand(h1, d0, generator=wg) * 2 - 1) * np.sqrt(6.0 / (h1 + d0))).to(dev)
w2 = ((torch.rand(h2, h1, generator=wg) * 2 - 1) * np.sqrt(6.0 / (h2 + h1))).to(dev)
rowptr, col = graph.to(dev)
cand = np.nonzero(graph.degrees() > 0)[0]
rs = np.random.default_rng(1)
NB = 350
seeds = torch.from_numpy(np.stack([rs.choice(cand, b, replace=False) for _ in range(NB)]).astype(np.int32)).to(dev)
keys = list(range(900, 900 + NB))
out = torch.empty(8, b, h2, device=dev)
pa = RolePipeline(rowptr, col, table, w1, w2, k1, k2, batch=b, depth=4, threads=True, relabel="degree")
pb = RolePipeline(rowptr, col, table, w1, w2, k1, k2, batch=b, depth=4, threads=True, relabel="degree", streams="new")


def run(pipes, lo, hi):
    for p in pipes: p.flush()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(lo, hi):
        pipes[i % len(pipes)].submit(seeds[i], keys[i], out[i % 8])
    for p in pipes: p.flush()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (hi - lo) * 1e6


for rep in range(3):
    for name, pipes in (("one pipeline ", [pa]), ("two pipelines", [pa, pb]), ("the second alone", [pb])):
        run(pipes, 0, 50)
        long_ = run(pipes, 50, 350)
        short = run(pipes, 0, 20)
        print(f"rep {rep} {name:17s} 300 batches {long_:6.2f} us per forward   20 batches {short:6.2f}", flush=True)
