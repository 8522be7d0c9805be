"""Where a block of dense_pc_kernel spends its time (diagnostic build -DSAGE_DENSE_STAMPS -> experiments/ab/pc_stamps.so).
SAGE355_LIB=experiments/ab/pc_stamps.so python experiments/r03/pc_stamps.py [concat]"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "graphsage-simple_amd")]
import numpy as np, torch
from sage355 import native
from sage355.engine import TwoHopEngine
from sage355.graph import rmat_graph
concat = len(sys.argv) > 1 and sys.argv[1] == "concat"
dev = torch.device("cuda", 0)
g = rmat_graph(20, 16_000_000, seed=0, cache_dir="/tmp/sage_cache")
n, d0, h1, h2, k1, k2, b = g.num_nodes, 256, 128, 128, 15, 25, 4096
gen = torch.Generator(device=dev).manual_seed(0)
table = torch.randn(n, d0, generator=gen, device=dev)
m = 2 if concat else 1
w1 = torch.randn(h1, m * d0, device=dev) / 16; w2 = torch.randn(h2, m * h1, device=dev) / 11
rowptr, col = g.to(dev)
cand = np.nonzero(g.degrees() > 0)[0]
rs = np.random.default_rng(1)
eng = TwoHopEngine(rowptr, col, table, w1, w2, k1, k2, max_batch=b, concat=concat, relabel="degree")
for i in range(5):
    seeds = torch.from_numpy(rs.choice(cand, b, replace=False).astype(np.int32)).to(dev)
    eng.forward(seeds, seed=i)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (512 * 40))()
assert native.lib().sage_debug_dense_stamps(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 40).astype(np.int64)
a = a[a[:, 0] > 0]
print("blocks", len(a), "(stamps of the LAST launch of the kernel: with concat that is the means' chunk, epi 2)")
t0 = a[:, 0]
rel = a - t0[:, None]
def show(name, c):
    ok = a[:, c] > 0
    v = rel[ok, c]
    if len(v): print("  %-44s n=%3d  median %8.0f  min %8.0f  max %8.0f ticks" % (name, len(v), np.median(v), v.min(), v.max()))
print("consumer wave 0 (s_memtime ticks = 100 MHz?  -> see span)")
show("W requested, before barrier 0 (1)", 1); show("after barrier 0 (2)", 2)
for j in range(4):
    show(f"tile {j}: MFMA + ds_add done", 3 + 2 * j); show(f"tile {j}: after barrier", 4 + 2 * j)
show("all waves past the main loop (39)", 39)
print("producer wave 4")
show("start (20)", 20); show("tile 0 staged (21)", 21); show("after barrier 0 (22)", 22)
for p in range(1, 4):
    show(f"phase {p}: loads landed", 22 + 4 * (p - 1) + 1); show(f"phase {p}: staged", 22 + 4 * (p - 1) + 2)
    show(f"phase {p}: requested + finished", 22 + 4 * (p - 1) + 3); show(f"phase {p}: after barrier", 22 + 4 * (p - 1) + 4)
show("last tile finished (38)", 38)
