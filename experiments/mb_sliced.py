import sys, os, ctypes
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "graphsage-simple_amd"))
import numpy as np, torch
from sage355 import ops
from sage355.engine import TwoHopEngine
from sage355.graph import rmat_graph
lib = ctypes.CDLL(os.path.join(HERE, "mb_sliced.so"))
dev = "cuda"
g = rmat_graph(20, 16_000_000, cache_dir="/tmp/sage_cache")
table = torch.randn(g.num_nodes, 256, device=dev)
rowptr, col = g.to(dev)
eng = TwoHopEngine(rowptr, col, table, torch.randn(128, 256, device=dev) / 16, torch.randn(128, 128, device=dev) / 11, 15, 25, max_batch=4096)
deg = g.degrees(); cands = np.nonzero(deg > 0)[0]
batches = []
for i in range(6):
    seeds = torch.from_numpy(np.random.default_rng(i).choice(cands, 4096, replace=False).astype(np.int32)).to(dev)
    eng.forward(seeds, seed=i)
    it = eng.intermediates()
    batches.append((it["nbr1"].clone(), it["cnt1"].clone()))
out = torch.empty(110000, 256, device=dev)
ref = ops.gather_mean(table, batches[0][0], batches[0][1])
def run(sl, blocks):
    st = torch.cuda.current_stream().cuda_stream
    for nbr, cnt in batches:
        lib.run_sliced(sl, blocks, ctypes.c_void_p(table.data_ptr()), ctypes.c_void_p(nbr.data_ptr()), ctypes.c_void_p(cnt.data_ptr()), 15, nbr.shape[0],
                       ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(st))
def timeit(sl, blocks):
    run(sl, blocks); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): run(sl, blocks)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 5 / len(batches) * 1e3
for sl in (16, 32):
    st = torch.cuda.current_stream().cuda_stream
    nbr, cnt = batches[0]
    lib.run_sliced(sl, 2048, ctypes.c_void_p(table.data_ptr()), ctypes.c_void_p(nbr.data_ptr()), ctypes.c_void_p(cnt.data_ptr()), 15, nbr.shape[0], ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(st))
    torch.cuda.synchronize()
    err = (out[: nbr.shape[0]] - ref).abs().max().item()
    for blocks in (2048, 4096):
        print(f"slice lanes={sl} ({sl*16} B) blocks={blocks}: {timeit(sl, blocks):6.1f} us  (max err {err:.1e})", flush=True)
def base():
    for nbr, cnt in batches: ops.gather_mean(table, nbr, cnt, out=out[: nbr.shape[0]])
base(); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5): base()
e.record(); torch.cuda.synchronize()
print("row-per-wave gather_mean: %.1f us" % (s.elapsed_time(e) / 5 / len(batches) * 1e3))
