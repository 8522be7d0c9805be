import sys, os
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "graphsage-simple_amd"))
import numpy as np, torch
from sage355 import ops, native
from sage355.graph import rmat_graph
dev = "cuda"
g = rmat_graph(20, 16_000_000, cache_dir="/tmp/sage_cache")
rowptr, col = g.to(dev)
deg = g.degrees()
cands = np.nonzero(deg > 0)[0]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
seeds = torch.from_numpy(np.random.default_rng(0).choice(cands, 4096, replace=False).astype(np.int32)).to(dev)
nbr, cnt, _, _ = ops.sample_neighbors(rowptr, col, seeds, 25, 1, 2)
s1 = torch.unique(nbr[nbr >= 0]).to(torch.int32)
print("S1", s1.numel())
L = native.lib()
def raw_sample(nodes, k, n_upper, n_dev, any_flag, frontier=None):
    n = nodes.shape[0] if n_upper is None else n_upper
    nb = torch.empty((n, k), dtype=torch.int32, device=dev); ct = torch.empty(n, dtype=torch.int32, device=dev)
    sl = torch.empty((n, k), dtype=torch.int32, device=dev)
    def f():
        if frontier is not None: frontier.reset(0)
        native.check(L.sage_sample_neighbors(native.ptr(rowptr), native.ptr(col), rowptr.shape[0]-1, native.ptr(nodes), n, native.ptr(n_dev), k, 1, 2,
            native.ptr(nb), native.ptr(ct), native.ptr(any_flag), frontier.c if frontier is not None else None, 0, native.ptr(sl) if frontier is not None else None, None, native.stream_handle()))
    return f
flag = torch.zeros(1, dtype=torch.int32, device=dev)
for k in (5, 15, 25):
    print(f"inner-like k={k}: n=|S1| exact: {timeit(raw_sample(s1, k, None, None, None)):.1f} us; with flag: {timeit(raw_sample(s1, k, None, None, flag)):.1f} us")
pad = torch.zeros(106496, dtype=torch.int32, device=dev); pad[: s1.numel()] = s1
nd = torch.tensor([s1.numel()], dtype=torch.int32, device=dev)
print(f"inner k=15 n_upper=106496 with n_dev: {timeit(raw_sample(pad, 15, 106496, nd, flag)):.1f} us")
fr = ops.Frontier(4096 * 26, dev)
print(f"outer k=25 frontier (incl reset kernel): {timeit(raw_sample(seeds, 25, None, None, flag, fr)):.1f} us; reset alone: {timeit(lambda: fr.reset(0)):.1f} us")
print(f"outer k=25 no frontier: {timeit(raw_sample(seeds, 25, None, None, flag)):.1f} us")
print(f"empty-ish launch (n=1): {timeit(raw_sample(seeds[:1].contiguous(), 25, None, None, None)):.1f} us")
