import sys, os
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "graphsage-simple_amd"))
import numpy as np, torch
from sage355.engine import TwoHopEngine
from sage355.graph import rmat_graph, relabel_by_degree
dev = "cuda"
g = rmat_graph(20, 16_000_000, cache_dir="/tmp/sage_cache")
if os.environ.get("SAGE_NODE_ORDER", "degree") == "degree": g = relabel_by_degree(g)[0]
table = torch.randn(g.num_nodes, 256, device=dev)
rowptr, col = g.to(dev)
eng = TwoHopEngine(rowptr, col, table, torch.randn(128, 256, device=dev) / 16, torch.randn(128, 128, device=dev) / 11, 15, 25, max_batch=4096, nan_empty=os.environ.get('SAGE_NAN', '1') == '1')
deg = g.degrees(); cands = np.nonzero(deg > 0)[0]
S = 40
seeds = torch.from_numpy(np.stack([np.random.default_rng(i).choice(cands, 4096, replace=False) for i in range(S)]).astype(np.int32)).to(dev)
eng.set_queue(seeds, list(range(S)))
eng.capture()
for _ in range(10): eng.replay()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(100): eng.replay()
e.record(); torch.cuda.synchronize()
print("SO_THREADS", os.environ.get("SAGE_SO_THREADS", "1024"), "graph forward single stream: %.1f us" % (s.elapsed_time(e) / 100 * 1e3), flush=True)
