"""Role pipeline with the role streams bound to CU subsets (hipExtStreamCreateWithCUMask), config 3, degree layout.

Question: the gather (G, 6 blocks x 64 VGPRs per SIMD) and the contraction (D, 368 of a SIMD's 512 VGPRs, 135 KB LDS) cannot
share a CU, so the pipeline's period is G + D.  G is bound by the fabric, D by its own CUs: with G on 3/4 of every XCD's CUs and
D on the remaining quarter they could run side by side.

    python experiments/cu_mask.py "S=4,G=3,D=-1,L=4" ...     role=m: the role's stream gets m quarters of every XCD's CUs
                                                            (m = 4: all, no mask); role=-m: the LAST m quarters
"""
import ctypes, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "graphsage-simple_amd")]
import numpy as np, torch
from sage355.engine import RolePipeline, TwoHopEngine
from sage355.graph import rmat_graph

steps, warmup = int(os.environ.get("STEPS", 300)), 40
dev = torch.device("cuda", 0)
torch.cuda.init()
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int


def masked_stream(m):
    """m quarters of every XCD's CUs: bit i stands for (XCD i % 8, CU i / 8) or (XCD i / 32, CU i % 32) depending on the driver;
    ((i % 8) + (i / 8)) % 4 picks the same number of CUs per XCD under both readings."""
    if m == 4:
        return torch.cuda.Stream(device=dev)
    words = (ctypes.c_uint32 * 8)()
    n = 0
    for i in range(256):
        q = ((i % 8) + (i // 8)) % 4
        on = q < m if m > 0 else q >= 4 + m
        if on:
            words[i // 32] |= 1 << (i % 32)
            n += 1
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask rc={rc}"
    s = torch.cuda.ExternalStream(h.value, device=dev)
    s.n_cus = n
    return s


g = rmat_graph(20, 16_000_000, seed=0, cache_dir="/tmp/sage_cache")
n, d0, h1, h2, k1, k2, b = g.num_nodes, 256, 128, 128, 15, 25, 4096
gen = torch.Generator(device=dev).manual_seed(0)
table = torch.randn(n, d0, generator=gen, device=dev)
wgen = torch.Generator().manual_seed(0)
w1 = ((torch.rand(h1, d0, generator=wgen) * 2 - 1) * np.sqrt(6.0 / (h1 + d0))).to(dev)
w2 = ((torch.rand(h2, h1, generator=wgen) * 2 - 1) * np.sqrt(6.0 / (h2 + h1))).to(dev)
rowptr, col = g.to(dev)
cand = np.nonzero(g.degrees() > 0)[0]
total = warmup + steps
rs = np.random.default_rng(1)
seeds = torch.from_numpy(np.stack([rs.choice(cand, b, replace=False) for _ in range(total)]).astype(np.int32)).to(dev)
keys = [0x5A6E355 + i for i in range(total)]
ref = TwoHopEngine(rowptr, col, table, w1, w2, k1, k2, max_batch=b, relabel="degree")
want = [ref.forward(seeds[i], seed=keys[i]).clone() for i in range(4)]
out = torch.empty(8, b, h2, device=dev)
for spec in sys.argv[1:]:
    m = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in spec.split(",")}
    streams = [masked_stream(m.get(r, 4)) for r in "SGDL"]
    pipe = RolePipeline(rowptr, col, table, w1, w2, k1, k2, batch=b, depth=4, roles="SGDL", streams=streams, relabel="degree")
    torch.cuda.synchronize()
    for i in range(4):
        pipe.submit(seeds[i], keys[i], out[i])
    pipe.synchronize()
    same = all(bool(torch.equal(out[i], want[i])) for i in range(4))
    for i in range(warmup):
        pipe.submit(seeds[i], keys[i], out[i % 8])
    torch.cuda.synchronize()
    reps = []
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(warmup, total):
            pipe.submit(seeds[i], keys[i], out[i % 8])
        torch.cuda.synchronize()
        reps.append((time.perf_counter() - t0) / steps * 1e6)
    print(f"{spec:28s} CUs " + "/".join(str(getattr(s, 'n_cus', 256)) for s in streams) + f": {min(reps):6.1f} us/forward (reps {', '.join('%.1f' % r for r in reps)})  identical={same}", flush=True)
    del pipe
