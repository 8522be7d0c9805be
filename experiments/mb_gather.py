"""micro-benchmark: gather_mean rate vs duplicate pattern / table size (GPU box only)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "graphsage-simple_amd"))
import numpy as np, torch
from sage355 import ops
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us
D, k, n = 256, 15, 23555
for rows in (1 << 20, 1 << 17, 1 << 15, 4096):
    table = torch.randn(rows, D, device=dev)
    out = torch.empty(n, D, device=dev)
    cnt = torch.full((n,), k, dtype=torch.int32, device=dev)
    for name, nbr in (
        ("unique", torch.randperm(rows, device=dev)[: n * k].to(torch.int32).view(n, k) if rows >= n * k else None),
        ("uniform", torch.randint(0, rows, (n, k), device=dev, dtype=torch.int32)),
        ("zipf", (torch.rand(n, k, device=dev) ** 6 * rows).to(torch.int32).clamp(max=rows - 1)),
    ):
        if nbr is None: continue
        nbr = nbr.contiguous()
        uniq = torch.unique(nbr).numel()
        us = timeit(lambda: ops.gather_mean(table, nbr, cnt, out=out))
        print(f"rows={rows:8d} ({rows*D*4/2**20:7.1f} MiB) {name:8s} unique={uniq:7d} edges={n*k} : {us:7.1f} us  per-edge {n*k*D*4/us/1e6:6.2f} TB/s  unique {uniq*D*4/us/1e6:6.2f} TB/s")
