"""Where a persistent block of dense_bf16x3_kernel spends its time (diagnostic library experiments/ab/libsage355_stamps.so,
built with -DSAGE_DENSE_STAMPS).  SAGE355_LIB=experiments/ab/libsage355_stamps.so python experiments/dense_stamps.py"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "graphsage-simple_amd")]
import numpy as np, torch
from sage355 import native
from sage355.engine import TwoHopEngine
from sage355.graph import rmat_graph, relabel_by_degree
dev = torch.device("cuda", 0)
g = relabel_by_degree(rmat_graph(20, 16_000_000, seed=0, cache_dir="/tmp/sage_cache"))[0]
n, d0, h1, h2, k1, k2, b = g.num_nodes, 256, 128, 128, 15, 25, 4096
gen = torch.Generator(device=dev).manual_seed(0)
table = torch.randn(n, d0, generator=gen, device=dev)
w1 = torch.randn(h1, d0, device=dev) / 16; w2 = torch.randn(h2, h1, device=dev) / 11
rowptr, col = g.to(dev)
cand = np.nonzero(g.degrees() > 0)[0]
rs = np.random.default_rng(1)
eng = TwoHopEngine(rowptr, col, table, w1, w2, k1, k2, max_batch=b)
for i in range(5):
    seeds = torch.from_numpy(rs.choice(cand, b, replace=False).astype(np.int32)).to(dev)
    eng.forward(seeds, seed=i)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (512 * 40))()
assert native.lib().sage_debug_dense_stamps(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 40).astype(np.int64)
used = a[:, 0] > 0
a = a[used]
print("blocks", len(a))
t0 = a[:, 0]
print("kernel span (first start -> last end): %.1f us at 100 MHz ticks?" % ((a[:, 39].max() - t0.min()) / 100.0))
rel = (a - t0[:, None])
def show(name, col):
    v = rel[:, col]; v = v[a[:, col] > 0]
    print("  %-34s n=%3d  median %8.0f  min %8.0f  max %8.0f ticks" % (name, len(v), np.median(v), v.min(), v.max()))
show("W loaded + split (stamp 1)", 1)
for t in range(3):
    show(f"tile {t}: staged + barrier", 2 + 3 * t)
    show(f"tile {t}: MFMA loop done", 3 + 3 * t)
    show(f"tile {t}: epilogue done", 4 + 3 * t)
show("block done (stamp 39)", 39)
print("start skew across blocks: %.0f ticks" % (t0.max() - t0.min()))
