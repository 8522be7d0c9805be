"""EngineTrainer at config-3 size (R-MAT 2^20 / 16 M edges, D0 = 256, H = 128/128, fanout 15/25, B = 4096, 16 classes):
time per SGD step, loss trend, and the step's gradients against torch autograd (fp64) on the sets the engine sampled."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "graphsage-simple_amd")]
import numpy as np, torch
from sage355.train import EngineTrainer
from sage355.graph import rmat_graph
dev = torch.device("cuda", 0)
g = rmat_graph(20, 16_000_000, seed=0, cache_dir="/tmp/sage_cache")
n, d0, b, nc = g.num_nodes, 256, 4096, 16
gen = torch.Generator(device=dev).manual_seed(0)
table = torch.randn(n, d0, generator=gen, device=dev)
proj = torch.randn(d0, nc, generator=gen, device=dev)
labels = (table @ proj).argmax(1)                     # learnable from the node's own features (and so from neighbourhood means, weakly)
rowptr, col = g.to(dev)
cand = np.nonzero(g.degrees() > 0)[0]
rs = np.random.default_rng(1)
for relabel in (None, "degree"):
    torch.manual_seed(0)
    tr = EngineTrainer(rowptr, col, table, nc, hidden1=128, hidden2=128, num_sample1=15, num_sample2=25, gcn=True, lr=0.05, max_batch=b, relabel=relabel)
    steps = 60
    seeds = [torch.from_numpy(rs.choice(cand, b, replace=False).astype(np.int32)).to(dev) for _ in range(steps)]
    # gradient check of one batch against fp64 autograd on the engine's own sets
    loss, (g1, g2, gc) = tr.grads(seeds[0], labels[seeds[0].long()], key=7)
    it = tr.engine.intermediates()
    n1 = it["n_s1"]
    T = tr.engine.table.double()
    nbr1, cnt1 = it["nbr1"].long(), it["cnt1"].long()
    mask1 = (torch.arange(nbr1.shape[1], device=dev)[None, :] < cnt1[:, None]).double()
    agg1 = (T[nbr1.clamp(min=0)] * mask1[:, :, None]).sum(1) / cnt1.clamp(min=1)[:, None].double()
    w1 = tr.w1.double().requires_grad_(True); w2 = tr.w2.double().requires_grad_(True); wc = tr.w_cls.double().requires_grad_(True)
    h1 = torch.relu(agg1 @ w1.t())
    row2, cnt2 = it["row2"].long(), it["cnt2"].long()
    mask2 = (torch.arange(row2.shape[1], device=dev)[None, :] < cnt2[:, None]).double()
    agg2 = (h1[row2.clamp(min=0)] * mask2[:, :, None]).sum(1) / cnt2.clamp(min=1)[:, None].double()
    out = torch.relu(agg2 @ w2.t())
    l64 = torch.nn.functional.cross_entropy(out @ wc.t(), labels[seeds[0].long()])
    r1, r2, rc = torch.autograd.grad(l64, (w1, w2, wc))
    err = lambda a, r: float((a.double() - r).abs().max() / r.abs().max())
    print(f"relabel={relabel}: rows of layer 1 {n1}; loss {float(loss):.5f} vs fp64 {float(l64):.5f}; grad max rel err w1 {err(g1, r1):.1e} w2 {err(g2, r2):.1e} w_cls {err(gc, rc):.1e}", flush=True)
    losses = []
    for i in range(10):
        tr.step(seeds[i], labels[seeds[i].long()], key=100 + i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(10, steps):
        losses.append(tr.step(seeds[i], labels[seeds[i].long()], key=100 + i))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (steps - 10)
    ls = [float(x) for x in losses]
    print(f"   {dt * 1e3:.3f} ms per SGD step of {b} seeds = {b / dt:.3g} seeds/s; loss {ls[0]:.4f} -> {ls[-1]:.4f}", flush=True)
    ring = torch.stack(seeds[:32])
    loss_t = tr.capture_step(ring, [500 + i for i in range(32)], labels)
    for _ in range(8):
        tr.replay_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(64):
        tr.replay_step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 64
    print(f"   captured step (one hipGraph per step): {dt * 1e3:.3f} ms = {b / dt:.3g} seeds/s; loss {float(loss_t):.4f}", flush=True)
    del tr
