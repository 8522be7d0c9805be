"""Sweep gather (mb_sweep.hip) vs the product's layer-1 gather on the engine's own sampled lists (config 3, degree layout)."""
import sys, os, ctypes
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "graphsage-simple_amd"))
import numpy as np, torch
from sage355 import ops
from sage355.engine import TwoHopEngine
from sage355.graph import rmat_graph
lib = ctypes.CDLL(os.path.join(HERE, os.environ.get("MB_LIB", "mb_sweep.so")))
dev = "cuda"
NB = 8
scale, edges = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (20, 16_000_000)
g = rmat_graph(scale, edges, cache_dir="/tmp/sage_cache")
table = torch.randn(g.num_nodes, 256, device=dev)
rowptr, col = g.to(dev)
eng = TwoHopEngine(rowptr, col, table, torch.randn(128, 256, device=dev) / 16, torch.randn(128, 128, device=dev) / 11, 15, 25,
                   max_batch=4096, relabel="degree")
tab = eng.table                                                  # rows by descending degree
rp = eng.rowptr1
nnz = int(rp[-1])
bounds = torch.searchsorted(rp, torch.tensor([nnz * p // NB for p in range(1, NB)], device=dev, dtype=rp.dtype)).to(torch.int32).cpu().numpy()
uni = np.array([g.num_nodes * p // NB for p in range(1, NB)], dtype=np.int32)
if os.environ.get("MB_BOUNDS") == "uni": bounds = uni
print("bucket bounds (equal edge mass):", bounds.tolist(), flush=True)
cands = np.nonzero(g.degrees() > 0)[0]
batches = []
for i in range(12):
    seeds = torch.from_numpy(np.random.default_rng(i).choice(cands, 4096, replace=False).astype(np.int32)).to(dev)
    eng.forward(seeds, seed=i)
    it = eng.intermediates()
    batches.append((it["nbr1"].clone(), it["cnt1"].clone()))
out = torch.empty(110000, 256, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
lib.run_sweep.restype = ctypes.c_int
def sweep(mode, blocks, bnd, nbr, cnt):
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.run_sweep(mode, blocks, P(tab), ctypes.c_int64(256), P(nbr), P(cnt), 15, nbr.shape[0], bnd.ctypes.data_as(ctypes.c_void_p), P(out),
                       ctypes.c_int64(256), st)
    assert rc == 0, rc
def timeit(fn):
    for b in batches: fn(*b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        for b in batches: fn(*b)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 3 / len(batches) * 1e3
nbr0, cnt0 = batches[0]
ref = ops.gather_mean(tab, nbr0, cnt0)
ref64 = None
print(f"rows {nbr0.shape[0]}, edges {int(cnt0.sum())}", flush=True)
print(f"product gather (ops.gather_mean): {timeit(lambda n_, c_: ops.gather_mean(tab, n_, c_)):.1f} us", flush=True)
if os.environ.get("MB_ONLY"):
    modes = [tuple(int(x) for x in os.environ["MB_ONLY"].split(","))]
else:
    modes = [(0, 256), (1, 512), (2, 256), (3, 512), (5, 512), (4, 768), (4, 1024)]
for mode, blocks in modes:
    out.zero_()
    sweep(mode, blocks, bounds, nbr0, cnt0); torch.cuda.synchronize()
    err = ((out[: ref.shape[0]] - ref).abs().max() / ref.abs().max()).item()
    a = out[: ref.shape[0]].clone()
    sweep(mode, blocks, bounds, nbr0, cnt0); torch.cuda.synchronize()
    same = torch.equal(a, out[: ref.shape[0]])
    t_b = timeit(lambda n_, c_: sweep(mode, blocks, bounds, n_, c_))
    t_u = timeit(lambda n_, c_: sweep(mode, blocks, uni, n_, c_))
    print(f"sweep mode {mode} blocks {blocks}: {t_b:6.1f} us with equal-mass buckets, {t_u:6.1f} us with uniform id ranges (= no sweep in this layout)"
          f"   max rel err vs product {err:.1e}, run-to-run identical {same}", flush=True)
