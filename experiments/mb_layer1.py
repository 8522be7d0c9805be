import sys, os
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "graphsage-simple_amd"))
import numpy as np, torch
from sage355 import ops
from sage355.engine import TwoHopEngine
from sage355.graph import rmat_graph
dev = "cuda"
g = rmat_graph(20, 16_000_000, cache_dir="/tmp/sage_cache")
table = torch.randn(g.num_nodes, 256, device=dev)
w1 = torch.randn(128, 256, device=dev) / 16
rowptr, col = g.to(dev)
eng = TwoHopEngine(rowptr, col, table, w1, torch.randn(128, 128, device=dev) / 11, 15, 25, max_batch=4096)
deg = g.degrees(); cands = np.nonzero(deg > 0)[0]
batches = []
for i in range(6):
    seeds = torch.from_numpy(np.random.default_rng(i).choice(cands, 4096, replace=False).astype(np.int32)).to(dev)
    eng.forward(seeds, seed=i)
    it = eng.intermediates()
    batches.append((it["nbr1"].clone(), it["cnt1"].clone()))
out = torch.empty(110000, 128, device=dev)
def once():
    for nbr, cnt in batches:
        ops.layer_forward(table, nbr, cnt, w1, out=out[: nbr.shape[0]])
for _ in range(2): once()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5): once()
e.record(); torch.cuda.synchronize()
print("variant", os.environ.get("SAGE_FUSED_VARIANT", "0"), "layer1 fused: %.1f us" % (s.elapsed_time(e) / 5 / len(batches) * 1e3), flush=True)
