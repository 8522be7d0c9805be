"""Where do the 20 steps of the driver's form (--steps 20 --warmup 5) spend their time?  Per-batch END times of the four roles
(timing events recorded behind every submit on the role streams, caller-thread submission) relative to the start of the region.
python experiments/r03/timeline20.py [steps] [warmup] [depth]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "graphsage-simple_amd"))
from sage355.engine import RolePipeline
from sage355.graph import rmat_graph

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 5
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = "cuda"
graph = rmat_graph(20, 16_000_000, seed=0, cache_dir=os.environ.get("SAGE_CACHE", "/tmp/sage_cache"))
gen = torch.Generator(device=dev).manual_seed(0)
d0, h1, h2, k1, k2, b = 256, 128, 128, 15, 25, 4096
table = torch.randn(graph.num_nodes, d0, generator=gen, device=dev)
w1 = (torch.rand(h1, d0, device=dev) * 2 - 1) * np.sqrt(6.0 / (h1 + d0))
w2 = (torch.rand(h2, h1, device=dev) * 2 - 1) * np.sqrt(6.0 / (h2 + h1))
cand = np.nonzero(graph.degrees() > 0)[0]
rs = np.random.default_rng(1)
total = warm + steps
seeds = torch.from_numpy(np.stack([rs.choice(cand, b, replace=False) for _ in range(total)]).astype(np.int32)).to(dev)
keys = [0x5A6E355 + i for i in range(total)]
rowptr, col = graph.to(dev)
pipe = RolePipeline(rowptr, col, table, w1, w2, k1, k2, batch=b, depth=depth, roles="SGDL", relabel="degree", threads=False)
out = torch.empty(8, b, h2, device=dev)
S, G, D, L = pipe.role_streams
t_ph = time.perf_counter()
while time.perf_counter() - t_ph < 0.5:
    pipe.submit_many(seeds[:8], keys[:8], out)
    torch.cuda.synchronize()
for rep in range(3):
    for i in range(warm):
        pipe.submit(seeds[i], keys[i], out[i % 8])
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)]
    host = []
    t0 = time.perf_counter()
    ev0.record(S)
    for j, i in enumerate(range(warm, total)):
        pipe.submit(seeds[i], keys[i], out[i % 8])
        for r, st in enumerate((S, G, D, L)):
            evs[j][r].record(st)
        host.append((time.perf_counter() - t0) * 1e6)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) * 1e6
    print(f"rep {rep}: region {el:.0f} us = {el / steps:.1f} us per step (events add 5 HIP calls per batch)")
    if rep == 2:
        prev = 0.0
        print(" batch  host-submitted   S-end   G-end   D-end   L-end   L-end delta")
        for j in range(steps):
            t = [ev0.elapsed_time(evs[j][r]) * 1e3 for r in range(4)]
            print(f"  {j:3d}   {host[j]:8.0f}      {t[0]:7.0f} {t[1]:7.0f} {t[2]:7.0f} {t[3]:7.0f}   {t[3] - prev:7.1f}")
            prev = t[3]
