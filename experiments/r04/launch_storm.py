"""Does a kernel START on another stream cost the running gather its L2 contents?  (gfx942/gfx950: the L2s of the eight XCDs are not coherent with
each other; a dispatch's acquire fence may invalidate them.)  The engine's own forward at config-3 size on stream A with stage events (the gather
launch's own start / stop), alone and beside a second stream that does nothing but start tiny kernels / record events / wait for them.
    python experiments/r04/launch_storm.py"""
import ctypes, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(R, "graphsage-simple_amd")); sys.path.insert(0, R)
import numpy as np, torch
from sage355.engine import TwoHopEngine
from sage355.graph import rmat_graph
from bench import HipEvents
dev = torch.device("cuda", 0)
graph = rmat_graph(20, 16_000_000, seed=0)
gen = torch.Generator(device=dev).manual_seed(0)
n, d0, h1, h2, b, k1, k2 = graph.num_nodes, 256, 128, 128, 4096, 15, 25
table = torch.randn(n, d0, generator=gen, device=dev)
wg = torch.Generator().manual_seed(0)
w1 = ((torch.rand(h1, d0, generator=wg) * 2 - 1) * np.sqrt(6.0 / (h1 + d0))).to(dev)
w2 = ((torch.rand(h2, h1, generator=wg) * 2 - 1) * np.sqrt(6.0 / (h2 + h1))).to(dev)
rowptr, col = graph.to(dev)
eng = TwoHopEngine(rowptr, col, table, w1, w2, k1, k2, max_batch=b, relabel="degree")
cand = np.nonzero(graph.degrees() > 0)[0]
rs = np.random.default_rng(1)
NB = 40
seeds = torch.from_numpy(np.stack([rs.choice(cand, b, replace=False) for _ in range(NB)]).astype(np.int32)).to(dev)
he = HipEvents()
sA, sB, sC = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
tiny = torch.zeros(64, device=dev)
big = torch.zeros(1 << 20, device=dev)


def forwards():
    """NB forwards back to back on stream A -> median per-stage microseconds (outer, inner, gather, contraction, layer 2)"""
    evs = []
    with torch.cuda.stream(sA):
        for i in range(NB):
            arr = (ctypes.c_void_p * 10)(*[he.create() for _ in range(10)])
            eng.forward(seeds[i], seed=100 + i, stage_events=arr)
            evs.append(arr)
    sA.synchronize()
    st = np.array([[he.elapsed_ms(a[2 * s], a[2 * s + 1]) * 1e3 for s in range(5)] for a in evs])
    for a in evs:
        for j in range(10):
            he.destroy(a[j])
    return np.median(st[5:], axis=0)


def storm(kind, count):
    """enqueue `count` operations on the side stream(s); returns at once"""
    if kind == "tiny kernels":
        with torch.cuda.stream(sB):
            for _ in range(count):
                tiny.add_(1.0)
    elif kind == "4 MB kernels (256 blocks)":
        with torch.cuda.stream(sB):
            for _ in range(count):
                big.add_(1.0)
    elif kind == "event record + wait ping-pong, no kernels":
        e = [torch.cuda.Event(), torch.cuda.Event()]
        for i in range(count):
            a, c = (sB, sC) if i % 2 == 0 else (sC, sB)
            e[i % 2].record(a)
            c.wait_event(e[i % 2])
    elif kind == "tiny kernel + event hand-off ping-pong":
        e = [torch.cuda.Event(), torch.cuda.Event()]
        for i in range(count):
            a, c = (sB, sC) if i % 2 == 0 else (sC, sB)
            with torch.cuda.stream(a):
                tiny.add_(1.0)
            e[i % 2].record(a)
            c.wait_event(e[i % 2])


torch.cuda.synchronize()
forwards()
base = forwards()
print("alone:                                              outer %5.1f inner %5.1f GATHER %5.1f contraction %5.1f layer2 %5.1f us" % tuple(base), flush=True)
for kind, count in (("tiny kernels", 6000), ("4 MB kernels (256 blocks)", 3000), ("event record + wait ping-pong, no kernels", 4000),
                    ("tiny kernel + event hand-off ping-pong", 3000)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    storm(kind, count)
    t_enq = time.perf_counter() - t0
    st = forwards()                       # runs while the storm is still draining (checked below)
    still = not (sB.query() and sC.query())
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("beside %-44s outer %5.1f inner %5.1f GATHER %5.1f contraction %5.1f layer2 %5.1f us   (storm: %d ops enqueued in %.1f ms, drained after %.1f ms, %s)"
          % ((kind + ":",) + tuple(st) + (count, t_enq * 1e3, t_all * 1e3, "still running when the forwards ended" if still else "ENDED EARLY: not covered")), flush=True)
