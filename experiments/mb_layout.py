"""Row-major vs slice-major feature table under the column-sliced gather (config-3 layer 1, degree-sorted node ids)."""
import sys, os, ctypes
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "graphsage-simple_amd"))
import numpy as np, torch
from sage355.engine import TwoHopEngine
from sage355.graph import rmat_graph, relabel_by_degree
lib = ctypes.CDLL(os.path.join(HERE, "mb_layout.so"))
dev = "cuda"
for order in ("original", "degree"):
    g = rmat_graph(20, 16_000_000, cache_dir="/tmp/sage_cache")
    if order == "degree": g = relabel_by_degree(g)[0]
    n = g.num_nodes
    table = torch.randn(n, 256, device=dev)
    sliced = table.view(n, 4, 64).permute(1, 0, 2).contiguous()          # [4][N][64]
    rowptr, col = g.to(dev)
    eng = TwoHopEngine(rowptr, col, table, torch.randn(128, 256, device=dev) / 16, torch.randn(128, 128, device=dev) / 11, 15, 25, max_batch=4096)
    cands = np.nonzero(g.degrees() > 0)[0]
    batches = []
    for i in range(40):
        seeds = torch.from_numpy(np.random.default_rng(i).choice(cands, 4096, replace=False).astype(np.int32)).to(dev)
        eng.forward(seeds, seed=i)
        it = eng.intermediates()
        batches.append((it["nbr1"].clone(), it["cnt1"].clone()))
    out = torch.empty(110000, 256, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    def run(tab, ss, ld, hot_k=1 << 30):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for nbr, cnt in batches:
            lib.run_layout(P(tab), ss, ld, P(nbr), P(cnt), 15, nbr.shape[0], P(out), st, hot_k)
    def timeit(tab, ss, ld, hot_k=1 << 30):
        run(tab, ss, ld, hot_k); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(2): run(tab, ss, ld, hot_k)
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / 2 / len(batches) * 1e3
    run(table, 64, 256); torch.cuda.synchronize(); a = out[:1000].clone()
    run(sliced, n * 64, 64); torch.cuda.synchronize()
    assert torch.equal(a, out[:1000])
    for rep in range(2):
        print(f"node order {order}: row-major {timeit(table, 64, 256):.1f} us, slice-major {timeit(sliced, n * 64, 64):.1f} us", flush=True)
    if order == "degree":
        for hk in (4096, 16384, 65536, 262144):
            print(f"   rows with id >= {hk} loaded nt: row-major {timeit(table, 64, 256, hk):.1f} us, slice-major {timeit(sliced, n * 64, 64, hk):.1f} us", flush=True)
