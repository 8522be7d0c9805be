"""Where does the layer-1 contraction's time go INSIDE the running pipeline?  In-kernel real-time stamps (diagnostic library built with
-DSAGE_DENSE_STAMPS: experiments/ab_build.sh stamps -DSAGE_DENSE_STAMPS) of ONE launch in steady state, against a launch alone.
    SAGE355_LIB=$PWD/experiments/ab/stamps.so python experiments/r04/dense_stamps_pipe.py"""
import ctypes, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(R, "graphsage-simple_amd")); sys.path.insert(0, R)
import numpy as np, torch
from sage355 import native
from sage355.engine import TwoHopEngine, RolePipeline
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
NB = 64
seeds = torch.from_numpy(np.stack([rs.choice(cand, b, replace=False) for _ in range(NB)]).astype(np.int32)).to(dev)
keys = list(range(500, 500 + NB))
L = native.lib()
L.sage_debug_dense_select.argtypes = [ctypes.c_int]
L.sage_debug_dense_stamps.argtypes = [ctypes.c_void_p]


def read(title):
    buf = (ctypes.c_ulonglong * (512 * 40))()
    assert L.sage_debug_dense_stamps(buf) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 40).astype(np.int64)
    a = a[a[:, 0] > 0]
    us = lambda t: t / 100.0
    t0 = a[:, 0].min()
    start = us(a[:, 0] - t0)
    end = us(a[:, 39] - t0)
    life = end - start
    print(f"== {title}: {len(a)} blocks; kernel span (first block start -> last block end) {end.max():.1f} us")
    q = lambda v: "min %.1f  p25 %.1f  median %.1f  p75 %.1f  max %.1f" % (v.min(), *np.percentile(v, [25, 50, 75]), v.max())
    print("   block start after the first block's:   " + q(start))
    print("   block lifetime (start -> done):        " + q(life))
    print("   block end after the first block start: " + q(end))
    tiles = []
    for t in range(5):
        c = 2 + 3 * t
        ok = a[:, c + 2] > 0
        if ok.sum() == 0:
            break
        prev = a[ok, c - 1] if t > 0 else a[ok, 0]
        tiles.append((t, int(ok.sum()), np.median(us(a[ok, c] - prev)), np.median(us(a[ok, c + 1] - a[ok, c])), np.median(us(a[ok, c + 2] - a[ok, c + 1]))))
    for t, cnt, st_, mf, ep in tiles:
        print(f"   tile {t} ({cnt:3d} blocks): {'start -> staged (incl. W prologue)' if t == 0 else 'previous epilogue -> staged      '} {st_:5.2f}  MFMA loop {mf:5.2f}  epilogue {ep:5.2f} us (medians)")
    return end.max()


# alone: single forwards on one stream
eng = TwoHopEngine(rowptr, col, table, w1, w2, k1, k2, max_batch=b, relabel="degree")
for i in range(4):
    eng.forward(seeds[i], seed=keys[i])
torch.cuda.synchronize()
assert L.sage_debug_dense_select(-1) == 0
eng.forward(seeds[5], seed=keys[5])
torch.cuda.synchronize()
read("alone (one forward on one stream)")
# in the pipeline: launch number `target` of a run of NB batches
pipe = RolePipeline(rowptr, col, table, w1, w2, k1, k2, batch=b, depth=4, threads=True, relabel="degree")
out = torch.empty(4, b, h2, device=dev)
pipe.submit_many(seeds, keys, out); pipe.synchronize()
for target in (30, 40):
    assert L.sage_debug_dense_select(target) == 0
    torch.cuda.synchronize()
    pipe.submit_many(seeds, keys, out); pipe.synchronize()
    read(f"in the running pipeline, launch {target} of {NB}")
