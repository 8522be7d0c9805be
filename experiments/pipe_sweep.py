"""Role-pipeline sweep on config 3 (run on an MI355X):  python experiments/pipe_sweep.py [--steps 200] [--configs ...]
Each config = depth:roles:priorities (e.g. 4:SGDL:S-1,D-1,L-1).  Prints us/forward per config, plus the two-stream
graph-replay baseline, and checks the pipe against TwoHopEngine bit for bit."""
import argparse, os, sys, time, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "graphsage-simple_amd")]
import numpy as np, torch
from sage355.engine import RolePipeline, TwoHopEngine
from sage355.graph import rmat_graph, relabel_by_degree

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--warmup", type=int, default=20)
ap.add_argument("--order", default="original")
ap.add_argument("--configs", nargs="*", default=["4:SGDL:", "4:SGDL:S-1,D-1,L-1", "3:SGDD:", "4:SGDD:S-1,D-1", "2:SSSS:", "6:SGDL:S-1,D-1,L-1"])
ap.add_argument("--baseline", type=int, default=1)
ap.add_argument("--tag", default="")
ap.add_argument("--bstreams", type=int, nargs="*", default=[1, 2])
args = ap.parse_args()
dev = torch.device("cuda", 0)
g = rmat_graph(20, 16_000_000, seed=0, cache_dir="/tmp/sage_cache")
if args.order == "degree":
    g = relabel_by_degree(g)[0]
n, d0, h1, h2, k1, k2, b = g.num_nodes, 256, 128, 128, 15, 25, 4096
gen = torch.Generator(device=dev).manual_seed(0)
table = torch.randn(n, d0, generator=gen, device=dev)
wgen = torch.Generator().manual_seed(0)
w1 = ((torch.rand(h1, d0, generator=wgen) * 2 - 1) * np.sqrt(6.0 / (h1 + d0))).to(dev)
w2 = ((torch.rand(h2, h1, generator=wgen) * 2 - 1) * np.sqrt(6.0 / (h2 + h1))).to(dev)
rowptr, col = g.to(dev)
cand = np.nonzero(g.degrees() > 0)[0]
total = args.warmup + args.steps
rs = np.random.default_rng(1)
seeds = torch.from_numpy(np.stack([rs.choice(cand, b, replace=False) for _ in range(total)]).astype(np.int32)).to(dev)
keys = [0x5A6E355 + i for i in range(total)]
res = {"tag": args.tag, "order": args.order, "env": {k: v for k, v in os.environ.items() if k.startswith("SAGE_")}}

def timed(fn_warm, fn_run):
    fn_warm(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn_run(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.steps * 1e6

if args.baseline:
    for ns in args.bstreams:
        engs = [TwoHopEngine(rowptr, col, table, w1, w2, k1, k2, max_batch=b) for _ in range(ns)]
        sts = [torch.cuda.Stream() for _ in range(ns)]
        outs = [torch.empty(b, h2, device=dev) for _ in range(ns)]
        for s in range(ns):
            mine = torch.arange(s, total, ns, device=dev)
            engs[s].set_queue(seeds[mine].contiguous(), [keys[i] for i in range(s, total, ns)])
            with torch.cuda.stream(sts[s]):
                engs[s].capture(out=outs[s])
        torch.cuda.synchronize()
        def run(lo, hi):
            for i in range(lo, hi):
                with torch.cuda.stream(sts[i % ns]):
                    engs[i % ns].replay()
        us = timed(lambda: run(0, args.warmup), lambda: run(args.warmup, total))
        res[f"graph_replay_{ns}stream"] = round(us, 2)
        print(f"graph replay, {ns} stream(s): {us:.2f} us/forward", flush=True)
        del engs

ref_eng = TwoHopEngine(rowptr, col, table, w1, w2, k1, k2, max_batch=b)
for cfg in args.configs:
    depth, roles, pr = cfg.split(":")
    prio = {x[0]: int(x[1:]) for x in pr.split(",") if x}
    pipe = RolePipeline(rowptr, col, table, w1, w2, k1, k2, batch=b, depth=int(depth), roles=roles, priorities=prio)
    out = torch.empty(max(int(depth), 8), b, h2, device=dev)
    torch.cuda.synchronize()
    # parity vs the plain engine (bit-identical), batches 0..5
    pipe.submit_many(seeds[:6], keys[:6], out[:8], segment_start=True)
    pipe.synchronize()
    ok = True
    for i in range(6):
        ok &= bool(torch.equal(out[i], ref_eng.forward(seeds[i], seed=keys[i])))
    us_many = timed(lambda: pipe.submit_many(seeds[:args.warmup], keys[:args.warmup], out),
                    lambda: pipe.submit_many(seeds[args.warmup:], keys[args.warmup:], out))
    def each(lo, hi):
        for i in range(lo, hi):
            pipe.submit(seeds[i], keys[i], out[i % out.shape[0]])
    us_each = timed(lambda: each(0, args.warmup), lambda: each(args.warmup, total))
    # host enqueue cost alone: time the submit loop without waiting
    torch.cuda.synchronize(); t0 = time.perf_counter(); pipe.submit_many(seeds[:40], keys[:40], out); host_us = (time.perf_counter() - t0) / 40 * 1e6
    torch.cuda.synchronize()
    # the same K batches captured as ONE hipGraph over the four role streams (fork -> submit_many -> join), launched once
    us_graph = float("nan")
    if os.environ.get("SWEEP_GRAPH"):
        cap = torch.cuda.Stream()
        def build(lo, hi):
            g_ = torch.cuda.CUDAGraph()
            with torch.cuda.stream(cap):
                torch.cuda.synchronize()
                with torch.cuda.graph(g_, stream=cap):
                    pipe.fork(cap)
                    pipe.submit_many(seeds[lo:hi], keys[lo:hi], out, segment_start=True)
                    pipe.join(cap)
            return g_
        try:
            gw, gt = build(0, args.warmup), build(args.warmup, total)
            torch.cuda.synchronize()
            with torch.cuda.stream(cap):
                gw.replay(); torch.cuda.synchronize()
                t0 = time.perf_counter(); gt.replay(); torch.cuda.synchronize()
                us_graph = (time.perf_counter() - t0) / args.steps * 1e6
            ok_g = all(bool(torch.equal(out[i % out.shape[0]], ref_eng.forward(seeds[i], seed=keys[i]))) for i in range(total - 3, total))
            print(f"     one hipGraph over the role streams: {us_graph:.2f} us/forward  identical={ok_g}", flush=True)
        except Exception as ex:
            print("     graph capture failed:", repr(ex)[:300], flush=True)
    res[cfg] = {"us_graph": us_graph, "us_submit_many": round(us_many, 2), "us_submit_each": round(us_each, 2), "host_enqueue_us": round(host_us, 2), "bit_identical": ok}
    print(f"pipe {cfg:28s}: {us_many:7.2f} us/forward (submit_many)  {us_each:7.2f} (submit each)  host enqueue {host_us:.1f} us  identical={ok}", flush=True)
    del pipe
print(json.dumps(res))
