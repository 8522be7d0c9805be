"""Two-phase (hot sources first) column-sliced gather vs the single-phase one, config-3 layer 1."""
import sys, os, ctypes
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "graphsage-simple_amd"))
import numpy as np, torch
from sage355 import ops
from sage355.engine import TwoHopEngine
from sage355.graph import rmat_graph
lib = ctypes.CDLL(os.path.join(HERE, "mb_phase.so"))
dev = "cuda"
g = rmat_graph(20, 16_000_000, cache_dir="/tmp/sage_cache")
table = torch.randn(g.num_nodes, 256, device=dev)
rowptr, col = g.to(dev)
eng = TwoHopEngine(rowptr, col, table, torch.randn(128, 256, device=dev) / 16, torch.randn(128, 128, device=dev) / 11, 15, 25, max_batch=4096)
deg = g.degrees(); cands = np.nonzero(deg > 0)[0]
rank = np.empty(g.num_nodes, dtype=np.int64); rank[np.argsort(-deg, kind="stable")] = np.arange(g.num_nodes)
rank_d = torch.from_numpy(rank).to(dev)
batches = []
for i in range(6):
    seeds = torch.from_numpy(np.random.default_rng(i).choice(cands, 4096, replace=False).astype(np.int32)).to(dev)
    eng.forward(seeds, seed=i)
    it = eng.intermediates()
    batches.append((it["nbr1"].clone(), it["cnt1"].clone()))      # only the live rows
out = torch.empty(110000, 256, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
def mark(nbr, hot_k):
    r = rank_d[nbr.clamp(min=0).long()]
    return torch.where((r >= hot_k) & (nbr >= 0), nbr | (-2147483648), nbr).to(torch.int32).contiguous()
def run(mode, marked):
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for (nbr, cnt), m in zip(batches, marked):
        lib.run_phase(mode, 2048, P(table), P(m), P(cnt), 15, nbr.shape[0], P(out), st)
def timeit(mode, marked):
    run(mode, marked); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): run(mode, marked)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 5 / len(batches) * 1e3
ref = ops.gather_mean(table, batches[0][0], batches[0][1])
for hot_k in (0, 2048, 4096, 8192, 16384, 32768):
    marked = [mark(nbr, hot_k) for nbr, _ in batches]
    for mode in (0, 1):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        lib.run_phase(mode, 2048, P(table), P(marked[0]), P(batches[0][1]), 15, batches[0][0].shape[0], P(out), st); torch.cuda.synchronize()
        err = ((out[: ref.shape[0]] - ref).abs().max() / ref.abs().max()).item()
        print(f"hot = top {hot_k:6d} by degree  {'two-phase' if mode else 'single   '}: {timeit(mode, marked):6.1f} us  (rows {ref.shape[0]}, max rel err {err:.1e})", flush=True)
