import sys, os, ctypes
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "graphsage-simple_amd"))
import numpy as np, torch
from sage355.engine import TwoHopEngine
from sage355.graph import rmat_graph
lib = ctypes.CDLL(os.path.join(HERE, "mb_gather2.so"))
dev = "cuda"
g = rmat_graph(20, 16_000_000, cache_dir="/tmp/sage_cache")
table = torch.randn(g.num_nodes, 256, device=dev)
w = torch.randn(128, 256, device=dev) / 16
rowptr, col = g.to(dev)
eng = TwoHopEngine(rowptr, col, table, w, torch.randn(128, 128, device=dev), 15, 25, max_batch=4096)
deg = g.degrees()
cands = np.nonzero(deg > 0)[0]
batches = []
for i in range(6):
    seeds = torch.from_numpy(np.random.default_rng(i).choice(cands, 4096, replace=False).astype(np.int32)).to(dev)
    eng.forward(seeds, seed=i)
    it = eng.intermediates()
    batches.append((it["nbr1"].clone(), it["cnt1"].clone()))
order = np.argsort(-deg)
def bitmap(top):
    bits = np.zeros(g.num_nodes // 32, dtype=np.uint32)
    ids = order[:top]
    np.bitwise_or.at(bits, ids >> 5, (np.uint32(1) << (ids & 31).astype(np.uint32)))
    return torch.from_numpy(bits.view(np.int32)).to(dev)
out = torch.empty(110000, 256, device=dev)
def timeit(policy, unroll, blocks, hot):
    st = torch.cuda.current_stream().cuda_stream
    def once():
        for nbr, cnt in batches:
            lib.run(policy, unroll, blocks, ctypes.c_void_p(table.data_ptr()), ctypes.c_void_p(nbr.data_ptr()), ctypes.c_void_p(cnt.data_ptr()),
                    15, nbr.shape[0], ctypes.c_void_p(hot.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(st))
    once(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): once()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 5 / len(batches) * 1e3
hot0 = bitmap(1)
for blocks in (1024, 2048, 4096):
    for unroll in (8, 16):
        print(f"blocks={blocks} unroll={unroll}: default {timeit(0, unroll, blocks, hot0):6.1f} us   nt-all {timeit(1, unroll, blocks, hot0):6.1f} us", flush=True)
for top in (512, 1024, 2048, 3072, 4096, 8192, 16384):
    hb = bitmap(top)
    print(f"hot top-{top:5d} by degree: nt for the rest, unroll 16, 2048 blocks: {timeit(2, 16, 2048, hb):6.1f} us   unroll 8: {timeit(2, 8, 2048, hb):6.1f}", flush=True)
