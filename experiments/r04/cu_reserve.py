"""reserved CUs for the latency-bound roles.  Role streams bound to CU subsets (hipExtStreamCreateWithCUMask), host enqueue
threads on, config 3, degree layout, both forms of the bench (300 steps after 40; 20 steps after 5, mean of 8 regions).

    python experiments/r04/cu_reserve.py "S=0,L=0,G=1-7,D=1-7" "S=0" ...      role=<set of eighths>: a-b ranges and a+b unions; absent = all 256 CUs

An "eighth" q = 32 CUs, 4 per XCD under both readings of the mask's bit order (bit i = (XCD i % 8, CU i / 8) or (XCD i / 32, CU i % 32)).
"""
import ctypes, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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


def parse_set(s):
    out = set()
    for part in s.split("+"):
        if "-" in part:
            a, b = part.split("-")
            out |= set(range(int(a), int(b) + 1))
        else:
            out.add(int(part))
    return out


def masked_stream(eighths):
    if eighths is None:
        s = torch.cuda.Stream(device=dev)
        s.n_cus = 256
        return s
    words = (ctypes.c_uint32 * 8)()
    n = 0
    for i in range(256):
        if ((i % 8) + (i // 8)) % 8 in eighths:
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
depth = int(os.environ.get("DEPTH", 4))
specs = sys.argv[1:]
hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
ph = RolePipeline(rowptr, col, table, w1, w2, k1, k2, batch=b, depth=depth, roles="SGDL", relabel="degree", threads=True)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    ph.submit_many(seeds[:32], keys[:32], out)
    ph.synchronize()
del ph
for spec in specs:
    m = {} if spec.startswith("base") else {kv.split("=")[0]: parse_set(kv.split("=")[1]) for kv in spec.split(",")}
    streams = [masked_stream(m.get(r)) for r in "SGDL"]
    pipe = RolePipeline(rowptr, col, table, w1, w2, k1, k2, batch=b, depth=depth, roles="SGDL", streams=streams, relabel="degree", threads=True)
    torch.cuda.synchronize()
    for i in range(4):
        pipe.submit(seeds[i], keys[i], out[i])
    pipe.synchronize()
    same = all(bool(torch.equal(out[i], want[i])) for i in range(4))
    r = {"long": [], "short": []}
    for rep in range(int(os.environ.get("REPS", 3))):
        for i in range(warmup):
            pipe.submit(seeds[i], keys[i], out[i % 8])
        pipe.synchronize()
        t0 = time.perf_counter()
        for i in range(warmup, total):
            pipe.submit(seeds[i], keys[i], out[i % 8])
        pipe.synchronize()
        r["long"].append((time.perf_counter() - t0) / steps * 1e6)
        sh = []
        for r8 in range(8):
            for i in range(5):
                pipe.submit(seeds[i], keys[i], out[i % 8])
            pipe.synchronize()
            t0 = time.perf_counter()
            for i in range(5, 25):
                pipe.submit(seeds[i], keys[i], out[i % 8])
            pipe.synchronize()
            sh.append((time.perf_counter() - t0) / 20 * 1e6)
        r["short"].append(sum(sh) / len(sh))
    print(f"{spec:34s} CUs " + "/".join(str(s.n_cus) for s in streams) + f": {steps}-step {min(r['long']):6.1f} ({', '.join('%.1f' % x for x in r['long'])})   "
          f"20-step {min(r['short']):6.1f} ({', '.join('%.1f' % x for x in r['short'])})  identical={same}", flush=True)
    del pipe
    torch.cuda.synchronize()
    for s_ in streams:
        if isinstance(s_, torch.cuda.ExternalStream):
            hip.hipStreamDestroy(ctypes.c_void_p(s_.cuda_stream))
