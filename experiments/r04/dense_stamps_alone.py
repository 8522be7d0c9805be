"""Stamps of the pipelined contraction alone: per block, time between consecutive stamps (0 start, 1 W requested, then barrier A / barrier B per iteration, 39 done)."""
import ctypes, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(R, "graphsage-simple_amd")); sys.path.insert(0, R)
import numpy as np, torch
from sage355 import native
from sage355.engine import TwoHopEngine
from sage355.graph import rmat_graph
dev = torch.device("cuda", 0)
graph = rmat_graph(20, 16_000_000, seed=0)
gen = torch.Generator(device=dev).manual_seed(0)
n, d0, h1, h2, b, k1, k2 = graph.num_nodes, 256, 128, 128, 4096, 15, 25
table = torch.randn(n, d0, generator=gen, device=dev)
w1 = torch.randn(h1, d0, device=dev) / 16; w2 = torch.randn(h2, h1, device=dev) / 11
rowptr, col = graph.to(dev)
cand = np.nonzero(graph.degrees() > 0)[0]
rs = np.random.default_rng(1)
eng = TwoHopEngine(rowptr, col, table, w1, w2, k1, k2, max_batch=b, relabel="degree")
L = native.lib()
L.sage_debug_dense_select.argtypes = [ctypes.c_int]; L.sage_debug_dense_stamps.argtypes = [ctypes.c_void_p]
for i in range(5):
    eng.forward(torch.from_numpy(rs.choice(cand, b, replace=False).astype(np.int32)).to(dev), seed=i)
torch.cuda.synchronize()
assert L.sage_debug_dense_select(-1) == 0
eng.forward(torch.from_numpy(rs.choice(cand, b, replace=False).astype(np.int32)).to(dev), seed=9)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (512 * 40))()
assert L.sage_debug_dense_stamps(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 40).astype(np.int64)
a = a[a[:, 0] > 0]
print("blocks", len(a), "span %.1f us" % ((a[:, 39].max() - a[:, 0].min()) / 100.0))
prev = a[:, 0]
for c in list(range(1, 12)) + [39]:
    ok = a[:, c] > 0
    if ok.sum() == 0: continue
    last = np.where(a[:, 1:c] > 0, a[:, 1:c], 0).max(axis=1) if c > 1 else a[:, 0]
    last = np.maximum(last, a[:, 0])
    print("  stamp %2d: %3d blocks, median %.2f us after the previous stamp, %.2f after the start" % (c, ok.sum(), np.median((a[ok, c] - last[ok]) / 100.0), np.median((a[ok, c] - a[ok, 0]) / 100.0)))
