"""Loaded latency: what ONE dependent global load costs on an idle chip, beside a streaming copy, beside a random-row gather, and beside the
library's own layer-1 gather / whole forward running back to back on another stream.  The mechanism behind DESIGN.md section 4: the
latency-bound kernels of the pipeline (samplers, layer 2, the contraction's tile staging) are chains of such loads.
    python experiments/r04/loaded_latency.py"""
import ctypes, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "graphsage-simple_amd")]
import numpy as np, torch
from sage355.engine import TwoHopEngine
from sage355.graph import rmat_graph

dev = torch.device("cuda", 0)
torch.cuda.init()
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "loaded_latency.so"))
lib.chase_launch.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
n = 1 << 28                                           # 2^28 x 4 B = 1 GiB
perm = torch.randperm(n, device=dev, dtype=torch.int32 if False else torch.int64)
nxt = torch.empty(n, dtype=torch.int32, device=dev)
nxt[perm] = torch.roll(perm, -1).to(torch.int32)      # one cycle through all entries
del perm
ticks = torch.zeros(1, dtype=torch.int64, device=dev)
end = torch.zeros(1, dtype=torch.int32, device=dev)
s_chase, s_load = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
HOPS = 4000


_start = [12345]


def chase():
    _start[0] = (_start[0] * 1103515245 + 12345) % n      # a new stretch of the cycle every time: nothing of it is cached
    lib.chase_launch(nxt.data_ptr(), _start[0], HOPS, ticks.data_ptr(), end.data_ptr(), s_chase.cuda_stream)
    s_chase.synchronize()
    return ticks.item() * 10.0 / HOPS                 # ns per hop (100 MHz counter)


def beside(name, work, seconds=0.05):
    """`work()` enqueues ~a millisecond of load on s_load; keep it fed from this thread while the chase runs."""
    with torch.cuda.stream(s_load):
        for _ in range(8):
            work()
    res = []
    for rep in range(3):
        with torch.cuda.stream(s_load):
            for _ in range(40):
                work()
        res.append(chase())
        torch.cuda.synchronize()
    print(f"{name:58s} {min(res):7.0f} ns per dependent load   (reps {', '.join('%.0f' % r for r in res)})", flush=True)


torch.cuda.synchronize()                              # the cycle was built on the default stream
idle = [chase() for _ in range(5)]
print(f"{'idle chip':58s} {min(idle):7.0f} ns per dependent load   (reps {', '.join('%.0f' % r for r in idle)})", flush=True)
a = torch.empty(1 << 28, dtype=torch.float32, device=dev)           # 1 GiB
b = torch.empty_like(a)
beside("beside a streaming copy (1 GiB -> 1 GiB, torch copy_)", lambda: b.copy_(a))
table = torch.randn(1 << 20, 256, device=dev)
idx = torch.randint(0, 1 << 20, (400_000,), device=dev)
out = torch.empty(400_000, 256, device=dev)
beside("beside a random-row gather (torch.index_select, 1 KiB rows)", lambda: torch.index_select(table, 0, idx, out=out))
g = rmat_graph(20, 16_000_000, seed=0, cache_dir="/tmp/sage_cache")
wgen = torch.Generator().manual_seed(0)
w1 = ((torch.rand(128, 256, generator=wgen) * 2 - 1) * 0.1).to(dev)
w2 = ((torch.rand(128, 128, generator=wgen) * 2 - 1) * 0.1).to(dev)
rowptr, col = g.to(dev)
eng = TwoHopEngine(rowptr, col, table, w1, w2, 15, 25, max_batch=4096, relabel="degree")
cand = np.nonzero(g.degrees() > 0)[0]
seeds = torch.from_numpy(np.random.default_rng(1).choice(cand, 4096, replace=False).astype(np.int32)).to(dev)
o = torch.empty(4096, 128, device=dev)
key = [0]
def fwd():
    key[0] += 1
    eng.forward(seeds, seed=key[0], out=o)
beside("beside the library's 2-hop forward, one stream, back to back", fwd)
from sage355.engine import RolePipeline
pipe = RolePipeline(rowptr, col, table, w1, w2, 15, 25, batch=4096, depth=4, relabel="degree", threads=True)
nb = 64
sd = torch.from_numpy(np.stack([np.random.default_rng(7 + i).choice(cand, 4096, replace=False) for i in range(nb)]).astype(np.int32)).to(dev)
po = torch.empty(8, 4096, 128, device=dev)
pipe.submit_many(sd, list(range(nb)), po); pipe.synchronize()
res = []
for rep in range(3):
    for _ in range(5):
        pipe.submit_many(sd, list(range(nb)), po)         # 320 batches = ~19 ms of pipeline
    pipe.flush()
    time.sleep(0.002)                                     # let the pipeline reach its steady state
    res.append(chase())
    pipe.synchronize()
print(f"{'beside the role pipeline in steady state (four streams)':58s} {min(res):7.0f} ns per dependent load   (reps {', '.join('%.0f' % r for r in res)})", flush=True)
