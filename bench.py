#!/usr/bin/env python3
"""bench.py -- node-embeddings/sec of the GraphSAGE 2-hop mean-aggregate forward on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one 2-hop forward (sample both hops + frontier dedupe + layer 1 on the frontier +
layer 2 on the seeds) over one batch of B seed nodes, through the C ABI (sage_forward2).
Workload at every N: BASELINE.json configs[2] -- synthetic R-MAT 2^20 nodes / 16 M edges,
256-dim fp32 features, fanout 15/25, H = 128/128, B = 4096 seeds per GPU -- the configuration the
roofline target is quoted on.  N > 1 (launched by torch.distributed.run, one rank per GPU): the
graph, table and weights are replicated, every rank draws its own seed batches (seed-node shard,
weak scaling), and the forward needs no collective; barrier + max-over-ranks timing only.

Inputs are resident in HBM before the timed region; outputs stay on the device.  Rank 0 prints
ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import random
import sys
import time

# One hardware queue per role stream: ROCm's default is 4 HW queues per process, shared round-robin by all HIP streams, and two
# role streams that land on one queue serialise (measured: role pipeline 82.6 us/forward with 4, 69.1 us with 6 or more).
# Read by the HIP runtime when it initialises, so it must be set before the first HIP call.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, "graphsage-simple_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK = 8.0e12          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
CACHE_DIR = os.environ.get("SAGE_CACHE", "/tmp/sage_cache")


PRESETS = {
    2: dict(dim=500, hidden1=50, hidden=128, k1=10, k2=25, batch=256),
    3: dict(scale=20, edges=16_000_000, dim=256, hidden1=128, hidden=128, k1=15, k2=25, batch=4096),
    4: dict(scale=23, edges=128_000_000, dim=256, hidden1=128, hidden=128, k1=15, k2=25, batch=4096),
    5: dict(scale=22, edges=62_000_000, dim=100, hidden1=128, hidden=128, k1=20, k2=25, batch=4096, truncate=2_400_000),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("SAGE_STREAMS", "2")),
                    help="mini-batches in flight (each on its own HIP stream with its own workspace)")
    ap.add_argument("--mode", choices=["gcn", "concat"], default="gcn",
                    help="encoder mode: gcn = model.py:219,222 (gcn=True, no concat); concat = encoders.py:49-54")
    ap.add_argument("--unfused", action="store_true", help="two-launch layers (gather_mean + linear_act)")
    ap.add_argument("--no-graph", action="store_true", help="same as --exec direct")
    ap.add_argument("--node-order", choices=["degree", "original"], default=os.environ.get("SAGE_NODE_ORDER", "original"),
                    help="original (default): the generator's node ids, the BASELINE workload as generated; degree: VARIANT in which "
                         "the dataset itself is renumbered by descending degree at ingestion (sage355.graph.relabel_by_degree)")
    ap.add_argument("--engine-layout", choices=["degree", "input"], default=os.environ.get("SAGE_ENGINE_LAYOUT", "degree"),
                    help="device layout the engine builds from the caller's graph: degree = rows sorted by descending degree "
                         "(TwoHopEngine(relabel='degree'): the most-gathered feature rows are neighbours in HBM; seeds keep the "
                         "caller's ids and are translated by the outer-hop kernel inside every forward); input = the caller's order")
    ap.add_argument("--exec", choices=["pipe", "pipegraph", "replay", "direct"], default=os.environ.get("SAGE_EXEC", "pipe"),
                    help="pipe: RolePipeline (stages on role streams), one host call per batch; pipegraph: the same role pipeline with the "
                         "W warm-up batches and the K timed batches each captured as ONE hipGraph over the four role streams (one graph "
                         "launch per region); replay: hipGraph replay from a device batch queue, --streams forwards in flight; direct: "
                         "host-enqueued forwards")
    ap.add_argument("--preheat-seconds", type=float, default=float(os.environ.get("SAGE_PREHEAT", "0.5")),
                    help="untimed forwards on throw-away batches BEFORE the W warm-up steps, until this much time has passed: a GPU that "
                         "sat idle while the host built the inputs needs tens of ms of load to reach its sustained clocks (measured: the "
                         "first 250 forwards of a process run 20 %% slower than the next 250); 0 = none.  Declared in config.preheat")
    ap.add_argument("--no-variant", action="store_true",
                    help="skip the second timed run with the engine in the caller's node order (reported as config.variants)")
    ap.add_argument("--depth", type=int, default=int(os.environ.get("SAGE_DEPTH", "4")),
                    help="pipe: batches in flight (workspaces).  Depth 4 ... 8 are indistinguishable (20-step form, eight interleaved runs on "
                         "one box: 71.2 vs 71.5 us; experiments/r03/call19.sh); under rocprofv3's tracer depth 8 is host-bound (89 vs 70 us)")
    ap.add_argument("--host-threads", type=int, choices=[0, 1], default=int(os.environ.get("SAGE_BENCH_THREADS", "1")),
                    help="pipe: 1 (default) = one host enqueue thread per role stream (sage_pipe_set_threads: a submit only posts the batch, "
                         "the four streams are fed in parallel); 0 = the submitting thread makes all thirteen HIP calls of a batch itself")
    ap.add_argument("--window", type=int, default=int(os.environ.get("SAGE_PIPE_WINDOW", "0")),
                    help="pipe with host threads: role S enqueues batch b only once batch b - window has left the GPU (0 = unbounded)")
    ap.add_argument("--high-priority", default=os.environ.get("SAGE_PIPE_PRIO", ""),
                    help="pipe: role letters whose streams get HIP's high stream priority, e.g. DL (default: none)")
    ap.add_argument("--roles", default=os.environ.get("SAGE_ROLES", "SGDL"), help="pipe: roles S,G,D,L -> streams, e.g. SGDL, SGDD")
    ap.add_argument("--batches-per-replay", type=int, default=0,
                    help="queued batches embedded in one hipGraph replay (0 = preset: 1, or 10 for the 256-seed Pubmed configuration, "
                         "whose 20 us of GPU work per batch is less than one graph launch costs the host)")
    ap.add_argument("--scale", type=int, default=20)
    ap.add_argument("--edges", type=int, default=16_000_000)
    ap.add_argument("--dim", type=int, default=256)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--k1", type=int, default=15)
    ap.add_argument("--k2", type=int, default=25)
    ap.add_argument("--self-loop", action="store_true", help="GCN-variant aggregator: the node joins its own neighbour set (aggregators.py:50-51)")
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 4, 5],
                    help="BASELINE.json configs[] preset: 3 = the bench line (default); 2 = Pubmed topology D0=500 H=50/128 fanout 10/25 "
                         "B=256; 4 = R-MAT 2^23 / 128 M edges; 5 = ogbn-products-shaped 2.4 M nodes / 62 M edges D0=100 fanout 20/25")
    ap.add_argument("--scale-variant", choices=["auto", "on", "off"], default=os.environ.get("SAGE_SCALE_VARIANT", "auto"),
                    help="auto (default): with the headline on configs[2], ALSO run BASELINE configs[3] (R-MAT 2^23 / 128 M edges, the workload "
                         "BASELINE names for the 1/2/4/8 scaling curve) through the same measurement code at every N and report it as "
                         "config.variants.configs3_rmat23 (own parity gate, timed-path check, counted bytes, live kernel timing); `value` stays "
                         "on configs[2].  on: whatever --config is; off: never")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (rehearsing N > 1 on a 1-GPU box)")
    ap.add_argument("--share-device", action="store_true", help="rehearsal only: every rank uses cuda:0 (more than two ranks: the configs[3] "
                                                                  "variant is off unless --scale-variant on: 45 GB per rank)")
    args = ap.parse_args()
    given = {a.split("=")[0].lstrip("-").replace("-", "_") for a in sys.argv[1:] if a.startswith("--")}
    args.exec_given = "exec" in given or "SAGE_EXEC" in os.environ
    if args.no_graph:
        args.exec, args.exec_given = "direct", True
    if "streams" in given and not args.exec_given:
        args.exec, args.exec_given = "replay", True         # --streams N selects the forwards-in-flight mode it belongs to
    args.hidden1 = args.hidden
    args.truncate = 0
    for key, val in PRESETS[args.config].items():       # a preset fills what the command line left at its default
        if key not in given:
            setattr(args, key, val)
    return args


class HipEvents:
    """hipEvent_t pairs recorded by sage_forward2_profiled on the kernel's own stream."""

    def __init__(self):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]

    def create(self):
        ev = ctypes.c_void_p()
        assert self.hip.hipEventCreate(ctypes.byref(ev)) == 0
        return ev

    def elapsed_ms(self, a, b):
        ms = ctypes.c_float()
        rc = self.hip.hipEventElapsedTime(ctypes.byref(ms), a, b)
        assert rc == 0, rc
        return ms.value

    def destroy(self, ev):
        self.hip.hipEventDestroy(ev)


def algorithmic_bytes(d0, h1, h2, k1dim, k2dim, b, n_s1, e1, e2, n_r1):
    """SURVEY.md 8(d) / BASELINE.md section 4: compulsory traffic of one forward, and the share of
    its dominant kernel (layer 1: raw rows in, ids in, h1 out, W1)."""
    layer1 = 4 * d0 * n_r1 + 4 * h1 * n_s1 + 4 * e1 + 4 * n_s1 + 4 * h1 * k1dim
    gather1 = 4 * d0 * n_r1 + 4 * d0 * n_s1 + 4 * e1 + 4 * n_s1       # split layer 1: rows in, ids in, [|S1|, D0] means out
    total = (4 * d0 * n_r1 + 2 * 4 * h1 * n_s1 + 4 * h2 * b + 4 * (e1 + e2) + 16 * (n_s1 + b)
             + 4 * (h1 * k1dim + h2 * k2dim))
    return total, layer1, gather1


def self_launch(n, argv, script=None, timeout=None):
    """`python bench.py --gpus N` from a plain shell: start N fresh rank processes (one per GPU) with
    torch.distributed.run and relay rank 0's JSON line.  Called BEFORE this process touches the GPU (a process
    that has initialised HIP must never exec or fork GPU work); the parent only waits.  -> exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)
    import tempfile
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL over dmabuf IPC (see the environment notes)
    env.setdefault("OMP_NUM_THREADS", "4")
    # rank 0 also leaves its line in a file of its own: N ranks share the stdout pipe, and a write longer than PIPE_BUF (the
    # line is ~5 KB) is not atomic there -- another rank's output could land in the middle of it
    fd, result_file = tempfile.mkstemp(prefix="sage_bench_", suffix=".json")
    os.close(fd)
    env["SAGE_BENCH_RESULT_FILE"] = result_file
    try:
        res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True, timeout=timeout)
        with open(result_file) as fh:
            from_file = [ln for ln in fh.read().splitlines() if ln.startswith("{") and '"metric"' in ln]
    finally:
        try:
            os.unlink(result_file)
        except OSError:
            pass
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in res.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    lines = from_file or lines
    if res.returncode != 0 or not lines:
        print(f"bench.py: the {n}-rank launch failed (rc={res.returncode}, {len(lines)} result lines)", file=sys.stderr)
        return res.returncode or 1
    print(lines[-1], flush=True)
    return 0


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.share_device:
        local_rank = 0
        if args.scale_variant == "auto" and world > 2:
            # rehearsal on ONE device: configs[3] needs ~45 GB per rank (8.6 GB table, its degree-ordered copy and slice-major copy, twice:
            # gate engine + pipeline) plus the sort temporaries of the relabelling; four ranks of it oversubscribe the 288 GB and the
            # driver starts evicting whole processes -- a four-rank rehearsal sat in torch.sort for minutes (experiments/r04/call34.sh)
            args.scale_variant = "off"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # SAGE_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, barriers, MAX all-reduce) with ONE rank -- the only way to run the
    # RCCL calls of this file on a one-GPU box (tests/test_gpu_round4.py); two ranks on one device are refused by RCCL
    if world > 1 or os.environ.get("SAGE_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # RCCL; only barriers + one MAX all-reduce of the time
        else:
            # one node, rendezvous on 127.0.0.1: keep gloo's pairs on the loopback interface too (the container's hostname may not resolve:
            # a four-rank rehearsal once sat in gloo's full-mesh connect for seven minutes)
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            dist.init_process_group(args.dist_backend)

    from sage355 import native
    native.lib()
    host = host_plan(args, world)

    def fence(p=None):
        if p is not None:
            p.flush()                             # host enqueue threads: everything submitted is on the streams
        torch.cuda.synchronize()
        if dist is not None:                      # one process: nothing can have been enqueued since the synchronize above
            dist.barrier()
            torch.cuda.synchronize()

    # ---- the headline: BASELINE configs[2] (or --config N) ----
    head = run_workload(args, rank, world, dev, dist, fence, host, side_variants=not args.no_variant, with_cpu=args.cpu_seconds > 0)
    variants = head.pop("variants", None) if head is not None else None

    # ---- BASELINE configs[3] (R-MAT 2^23 / 128 M edges: the workload BASELINE names for the 1/2/4/8 curve) beside it, at EVERY N since round 4
    #      (VERDICT r3 #2): the same function, hence the same K / W, fences, max-over-ranks, oracle gate, timed-path check, counted bytes and
    #      live kernel timing as the headline; never substituted for `value` ----
    if args.scale_variant == "on" or (args.scale_variant == "auto" and args.config == 3):
        import copy
        a4 = copy.copy(args)
        for key, val in PRESETS[4].items():
            setattr(a4, key, val)
        a4.config, a4.truncate = 4, 0
        torch.cuda.empty_cache()
        sv = run_workload(a4, rank, world, dev, dist, fence, host, side_variants=False, with_cpu=False)
        if rank == 0:
            ro = sv["roofline"] or {}
            variants = dict(variants or {}, configs3_rmat23={
                "value": sv["value"], "unit": "embeddings/s", "ms_per_step": sv["ms_per_step"], "n_gpus": world, "workload": sv["workload"],
                "execution": sv["execution"], "parity_max_err_vs_fp64_oracle": sv["parity_err"], "timed_path_check": sv["timed_check"],
                "forward_bytes": ro.get("forward_bytes"), "forward_GBps": ro.get("forward_GBps"), "forward_frac": ro.get("forward_frac"),
                "roofline": {k: ro.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "bytes_per_launch", "kernel_ms", "kernel_ms_is",
                                                     "kernel_ms_alone", "frac_alone", "stage_ms_alone", "mean_sizes", "per_edge_gather_bytes")},
                "setup_seconds": sv["setup_seconds"],
                "note": "the workload BASELINE names for the 1/2/4/8 curve, measured by the same code path as the headline (steps / warm-up / "
                        "fences / max-over-ranks / oracle gate on the last timed batch / bit-for-bit check of the timed path's output / bytes "
                        "counted per batch / the gather launch's own start-stop events); `value` stays on configs[2]"})

    if rank == 0:
        line = {
            "metric": "node-embeddings/sec (2-hop forward)", "value": head["value"], "unit": "embeddings/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": dict(head["config"], variants=variants, host=host,
                           parallelism=f"seed-shard x{world}, replicated graph+features, no forward collective"),
            "parity_max_err_vs_fp64_oracle": head["parity_err"], "timed_path_check": head["timed_check"],
            "roofline": head["roofline"], "cpu_baseline": head["cpu_baseline"],
        }
        print(json.dumps(line), flush=True)
        if os.environ.get("SAGE_BENCH_RESULT_FILE"):          # the self-launching parent reads it from here (see self_launch)
            with open(os.environ["SAGE_BENCH_RESULT_FILE"], "w") as fh:
                fh.write(json.dumps(line) + "\n")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def usable_host_cores():
    """Host cores this process may actually use: the affinity mask / cgroup quota, not the machine's core count."""
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            usable = max(1, min(usable, int(int(quota) / int(period))))
    except Exception:
        pass
    return usable


def host_plan(args, world):
    """How the role pipeline is fed.  One host enqueue thread per role stream (four spinning threads + the submitter
    per rank) only where every rank of this node has at least five usable cores to itself; otherwise the submitting thread makes all
    of a batch's HIP calls itself.  Reported in config.host of every line."""
    usable = usable_host_cores()
    per_rank = usable // max(1, world)
    want = bool(args.host_threads) and len(set(args.roles)) == 4
    threads = want and per_rank >= 5
    return {"host_cores_usable": usable, "ranks_on_node": world, "cores_per_rank": per_rank,
            "enqueue_mode": "one host thread per role stream" if threads else "submitting thread only",
            "why": ("--host-threads 0 / fewer than four role streams" if not want else
                    "cores_per_rank >= 5" if threads else "cores_per_rank < 5: five busy host threads per rank would oversubscribe the node"),
            "role_threads": bool(threads)}


_PROGRESS = {"fh": None, "t0": time.perf_counter()}


def mark(rank, msg):
    """SAGE_BENCH_PROGRESS=<dir>: every rank appends `seconds message` lines to <dir>/rank<r>.log (where a silent multi-rank run stands;
    SAGE_BENCH_STACKS_AFTER=<seconds> adds a dump of every thread's Python stack there, once, if the run is still going by then)."""
    d = os.environ.get("SAGE_BENCH_PROGRESS")
    if not d:
        return
    if _PROGRESS["fh"] is None:
        os.makedirs(d, exist_ok=True)
        _PROGRESS["fh"] = open(os.path.join(d, f"rank{rank}.log"), "a", buffering=1)
        after = float(os.environ.get("SAGE_BENCH_STACKS_AFTER", "0") or 0)
        if after > 0:
            import faulthandler
            faulthandler.dump_traceback_later(after, exit=False, file=_PROGRESS["fh"])
    _PROGRESS["fh"].write(f"{time.perf_counter() - _PROGRESS['t0']:8.2f} {msg}\n")


def run_workload(args, rank, world, dev, dist, fence, host, side_variants, with_cpu):
    """One workload (args.config and the sizes its preset filled in) through the parity gate, the timed region, the timed-path check and
    the kernel / byte accounting.  Every rank runs it; rank 0 returns the dict the JSON line is made of, the others None."""
    t_setup = time.perf_counter()
    from sage355.engine import RolePipeline, TwoHopEngine
    from sage355.graph import rmat_graph
    mark(rank, f"workload config {args.config}: start")

    # ---- synthetic inputs: rank 0 generates, the others load its cache ----
    def make_graph():
        if args.config == 2:        # real Pubmed topology (edge list shipped as a fixture; features are synthetic)
            from sage355.graph import CSRGraph
            z = np.load(os.path.join(REPO, "tests", "golden", "pubmed_topology.npz"))
            return CSRGraph(z["rowptr"], z["col"], len(z["rowptr"]) - 1)
        g_ = rmat_graph(args.scale, args.edges, seed=0, cache_dir=CACHE_DIR)
        if args.truncate:
            from sage355.graph import truncate_nodes
            g_ = truncate_nodes(g_, args.truncate)
        if args.node_order == "degree":       # VARIANT: the dataset itself renumbered by descending degree at ingestion
            from sage355.graph import relabel_by_degree
            g_ = relabel_by_degree(g_)[0]
        return g_

    if rank == 0:
        graph = make_graph()
    if dist is not None:
        dist.barrier()
    if rank != 0:
        graph = make_graph()
    mark(rank, "graph ready")
    n = graph.num_nodes
    concat = args.mode == "concat"
    mult = 2 if concat else 1
    d0, h1, h2, k1, k2, b = args.dim, args.hidden1, args.hidden, args.k1, args.k2, args.batch
    gen = torch.Generator(device=dev).manual_seed(0)
    table = torch.randn(n, d0, generator=gen, device=dev)
    if args.config == 2:            # Pubmed features rand * Bernoulli(0.1)
        table = torch.rand(n, d0, generator=gen, device=dev) * (torch.rand(n, d0, generator=gen, device=dev) < 0.1)
    wgen = torch.Generator().manual_seed(0)
    w1 = ((torch.rand(h1, mult * d0, generator=wgen) * 2 - 1) * np.sqrt(6.0 / (h1 + mult * d0))).to(dev)
    w2 = ((torch.rand(h2, mult * h1, generator=wgen) * 2 - 1) * np.sqrt(6.0 / (h2 + mult * h1))).to(dev)
    rowptr, col = graph.to(dev)
    deg = graph.degrees()
    candidates = np.nonzero(deg > 0)[0]
    total_steps = args.warmup + args.steps
    rs = np.random.default_rng(1 + 7919 * rank)
    seeds_host = np.stack([rs.choice(candidates, b, replace=False) for _ in range(total_steps)]).astype(np.int32)
    seeds_dev = torch.from_numpy(seeds_host).to(dev)
    sampler_seed = [0x5A6E355 + 1000003 * rank + i for i in range(total_steps)]

    # The engine keeps graph and table in a layout of its own: relabel="degree" = rows sorted by descending degree (built once
    # in the constructor, like the dict-of-sets -> CSR conversion); seeds arrive in the GENERATOR's ids and are translated by
    # the outer-hop kernel inside every timed forward; outputs are in the caller's seed order.
    relabel = "degree" if (args.engine_layout == "degree" and args.config != 2 and args.node_order != "degree") else None
    ekw = dict(concat=concat, agg_self_loop=args.self_loop, fused=not args.unfused, relabel=relabel)
    base = TwoHopEngine(rowptr, col, table, w1, w2, k1, k2, max_batch=b, **ekw)
    mark(rank, "engine built")
    base_split = base.layout.layer1_split
    nstreams = max(1, args.streams)

    # ---- parity gate: one batch against the fp64 oracle on the GPU's own sampled sets.  The batch is the LAST one of the timed
    #      region: after the timed region the output the role pipeline left for that very batch is compared with this one bit
    #      for bit (gate the path that is timed, at the size it is timed) ----
    parity_err = None
    gate_i = total_steps - 1
    gate_out = None
    if not args.no_parity and rank == 0:
        from oracle import ref_sparse
        gate_out = base.forward(seeds_dev[gate_i], seed=sampler_seed[gate_i]).clone()
        o = gate_out.cpu()
        it = base.intermediates()            # ids below are the engine's INTERNAL ones: index its own table copy with them
        first = it["first_frontier_row"]
        s1, nbr1, cnt1 = it["s1_nodes"].cpu().numpy(), it["nbr1"].cpu().numpy(), it["cnt1"].cpu().numpy()
        seeds_int = seeds_host[gate_i] if base._new_of_old is None else base._new_of_old[torch.from_numpy(seeds_host[gate_i]).long().to(dev)].cpu().numpy()
        ref = ref_sparse.two_hop_forward(base.table.cpu(), w1.cpu(), w2.cpu(), seeds_int, it["nbr2"].cpu().numpy(),
                                         it["cnt2"].cpu().numpy(), s1[first:], nbr1[first:], cnt1[first:], gcn=not concat,
                                         agg_gcn=args.self_loop,
                                         seed_nbr1=nbr1[:first] if concat else None, seed_cnt1=cnt1[:first] if concat else None)
        scale = ref.abs().amax(1, keepdim=True).clamp_min(1e-30)
        parity_err = ((o.double() - ref).abs() / scale).max().item()
        if not parity_err <= 1e-5:
            raise SystemExit(f"parity gate failed: max |gpu-oracle|/rowmax = {parity_err:.3e}")

    mark(rank, "parity gate done")
    # ---- execution modes of the timed region ----
    #  pipe   (default): RolePipeline -- stages S / G / D / L on role streams, `depth` batches in flight, one host call per batch
    #  replay          : hipGraph replay from a device batch queue, `streams` independent forwards in flight (round 1's mode)
    #  direct          : host-enqueued sage_forward2 calls
    exec_mode = args.exec
    bpr = args.batches_per_replay if args.batches_per_replay > 0 else (10 if args.config == 2 else 1)
    if args.config == 2 and args.exec == "pipe" and not args.exec_given:
        exec_mode = "replay"        # 256-seed batches: 20 us of GPU work per batch, less than the host's enqueue per batch
    pipe = None
    engines, streams, outs = [], [], []
    pipe_graphs = None
    if exec_mode in ("pipe", "pipegraph"):
        # (every RolePipeline of a process runs on its first pipeline's role streams -- sage355.engine: configs[3] as the second workload
        #  read 97.8 us per forward on fresh streams that shared a hardware queue, 82.4 on the first pipeline's)
        pipe = RolePipeline(rowptr, col, table, w1, w2, k1, k2, batch=b, depth=args.depth, roles=args.roles,
                            priorities={ch: -1 for ch in args.high_priority},
                            threads=host["role_threads"] and exec_mode == "pipe", window=args.window, **ekw)
        pipe_out = torch.empty(max(args.depth, 4), b, h2, device=dev)
        torch.cuda.synchronize()
        if exec_mode == "pipegraph":
            # fork -> K submits -> join captured once per region; the graphs embed the seeds pointers and keys of their batches
            gw, cap_stream = pipe.capture(seeds_dev[:args.warmup], sampler_seed[:args.warmup], pipe_out) if args.warmup > 0 else (None, None)
            gt, cap_stream = pipe.capture(seeds_dev[args.warmup:], sampler_seed[args.warmup:], pipe_out, stream=cap_stream)
            pipe_graphs = (gw, gt, cap_stream)
            torch.cuda.synchronize()
    else:
        engines = [base] + [base.sibling() for _ in range(nstreams - 1)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
        outs = [torch.empty(b, h2, device=dev) for _ in range(nstreams)]
        use_graph = exec_mode == "replay"
        if use_graph:
            if bpr > 1 and (args.warmup % (bpr * nstreams) or args.steps % (bpr * nstreams)):
                bpr = 1                                   # the step counts must split into whole replays per stream
            group = np.arange(total_steps) // bpr
            for s in range(nstreams):
                mine = torch.from_numpy(np.nonzero(group % nstreams == s)[0]).to(dev)
                engines[s].set_queue(seeds_dev[mine].contiguous(), [sampler_seed[int(i)] for i in mine.cpu()])
                with torch.cuda.stream(streams[s]):
                    if bpr == 1:
                        engines[s].capture(out=outs[s])
                    else:
                        engines[s].capture(batches=bpr)
            torch.cuda.synchronize()

    pipe_threaded = bool(pipe is not None and pipe.threads)

    def run(step_range):
        if pipe_graphs is not None:
            g_ = pipe_graphs[0] if step_range.start == 0 else pipe_graphs[1]
            if g_ is not None:
                with torch.cuda.stream(pipe_graphs[2]):
                    g_.replay()
            return
        if pipe is not None:
            # one host call per batch (11 enqueues each): measured 2-3 us per forward FASTER than handing all K batches to the
            # C loop at once (sage_pipe_submit_many), whose only difference is that the host runs further ahead of the GPU
            if os.environ.get("SAGE_SUBMIT_MANY") == "1":      # A/B knob: the whole region handed to the C loop at once
                pipe.submit_many(seeds_dev[step_range.start:step_range.stop], sampler_seed[step_range.start:step_range.stop], pipe_out)
                return
            for i in step_range:
                pipe.submit(seeds_dev[i], sampler_seed[i], pipe_out[i % pipe_out.shape[0]])
            return
        if exec_mode == "replay" and bpr > 1:
            for j in range(step_range.start // bpr, step_range.stop // bpr):  # one replay = steps j*bpr .. j*bpr + bpr - 1
                s = j % nstreams
                with torch.cuda.stream(streams[s]):
                    engines[s].replay()
            return
        for i in step_range:
            s = i % nstreams
            with torch.cuda.stream(streams[s]):
                if exec_mode == "replay":
                    engines[s].replay()
                else:
                    engines[s].forward(seeds_dev[i], seed=sampler_seed[i], out=outs[s])

    setup_seconds = time.perf_counter() - t_setup
    # ---- preheat (untimed, throw-away batches, same execution mode), then W warm-up steps, then the timed region ----
    preheat_forwards = 0
    if args.preheat_seconds > 0:
        ph_rs = np.random.default_rng(777 + rank)
        ph_n = 32
        ph_seeds = torch.from_numpy(np.stack([ph_rs.choice(candidates, b, replace=False) for _ in range(ph_n)]).astype(np.int32)).to(dev)
        ph_keys = [0xBEEF + i for i in range(ph_n)]
        ph_eng = base if pipe is None else None
        t_ph = time.perf_counter()
        while time.perf_counter() - t_ph < args.preheat_seconds:
            if pipe is not None:
                pipe.submit_many(ph_seeds, ph_keys, pipe_out)
                pipe.flush()
            else:
                for i in range(ph_n):
                    ph_eng.forward(ph_seeds[i], seed=ph_keys[i])
            torch.cuda.synchronize()
            preheat_forwards += ph_n
        if pipe is None and exec_mode == "replay":
            for e_ in engines:                    # the preheat used engine 0 directly: queue cursors are untouched, nothing to rewind
                pass
    mark(rank, "preheat done")
    # ---- warm-up, then the timed region: exactly K steps between two barrier+synchronize fences ----
    run(range(args.warmup))
    fence(pipe)
    mark(rank, "warm-up done, opening fence passed")
    t0 = time.perf_counter()
    run(range(args.warmup, total_steps))
    host_submit_ms = (time.perf_counter() - t0) / args.steps * 1e3       # the submitting thread, per step
    if pipe is not None:
        pipe.flush()
    host_enqueue_ms = (time.perf_counter() - t0) / args.steps * 1e3      # until the last HIP call of the region was made (it must stay below ms_per_step)
    torch.cuda.synchronize()
    own_done = time.perf_counter() - t0                                  # this rank's K steps have left the GPU (N > 1: reported, never `value`)
    fence(pipe)
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed, own_done], device=dev if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, own_done = float(t[0].item()), float(t[1].item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * b * args.steps / elapsed
    mark(rank, f"timed region done: {ms_per_step * 1e3:.1f} us per step")

    # ---- the timed path's own output, checked AFTER the clock stopped: what the timed execution mode left for the gate batch
    #      (the last timed step) against the oracle-gated single forward of the same (seeds, key), bit for bit ----
    timed_check = None
    if gate_out is not None:
        if pipe_graphs is not None:
            got = pipe_out[(gate_i - args.warmup) % pipe_out.shape[0]]      # the timed graph numbers its batches from 0
        elif pipe is not None:
            got = pipe_out[gate_i % pipe_out.shape[0]]
        elif exec_mode == "replay" and bpr > 1:
            got = engines[(gate_i // bpr) % nstreams]._graph_out[gate_i % bpr]
        else:
            got = outs[gate_i % nstreams]
        same = bool(torch.equal(got, gate_out))
        timed_check = {"batch": gate_i, "bit_identical_to_oracle_gated_forward": same,
                       "what": "output slot the timed region wrote for its last step vs TwoHopEngine.forward(same seeds, same key), "
                               "which passed the fp64-oracle gate above"}
        if not same:
            raise SystemExit(f"timed-path check failed: the {exec_mode} execution's output for step {gate_i} differs from the single forward "
                             f"(max abs diff {(got - gate_out).abs().max().item():.3e})")

    workload = (f"BASELINE configs[{args.config - 1}]: "
                + ("Pubmed topology (19717 nodes)" if args.config == 2 else
                   f"R-MAT 2^{args.scale} / {args.edges} edges" + (f" truncated to {n} nodes" if args.truncate else f" ({n} nodes)"))
                + (", node ids renumbered by degree at ingestion" if (args.node_order == "degree" and args.config != 2) else "")
                + f", {graph.nnz} directed nnz, {d0}-dim fp32 features, 2-layer GraphSAGE-mean {args.mode} encoder"
                + (" + self-loop (GCN-variant) aggregator [intended semantics of aggregators.py:50-51; the reference line raises TypeError, so this variant is parity-unpinned]" if args.self_loop else "")
                + f" H={h1}/{h2}, fanout {k1}/{k2}, batch {b} seeds per GPU")

    mark(rank, "timed-path check done")
    # ---- dominant kernel (layer-1 gather) duration, measured live with HIP events on the kernel's own stream:
    #      (a) IN SITU: the same K steps again in the same execution mode (the pipeline running, events around the gather on
    #          stream G) -- what the timed region's kernel launches took, overlap stretch included;
    #      (b) ALONE: host-enqueued forwards, one batch in flight -- the kernel's own speed.
    #      Both intervals are between the gather launch's OWN start / stop events (hipExtLaunchKernelGGL; csrc/sage_gather.hip's measurement
    #      hook), i.e. the kernel's execution as rocprofv3 reports it -- until round 3 they were marker packets around the launch, which in
    #      a busy queue read 5-9 us longer.  The other stages' figures in stage_ms_alone are still marker intervals.
    #      The data-determined set sizes of every batch are counted by torch reductions after pass (b). ----
    roofline = None
    if rank == 0:
        he = HipEvents()
        split = bool(base.layout.layer1_split)      # layer 1 ran as column-sliced gather + dense contraction
        insitu_ms = None
        if pipe is not None and split:
            pairs = []
            for i in range(args.warmup, total_steps):
                arr = (ctypes.c_void_p * 2)(he.create(), he.create())
                pairs.append(arr)
            torch.cuda.synchronize()
            for i in range(args.warmup):
                pipe.submit(seeds_dev[i], sampler_seed[i], pipe_out[i % pipe_out.shape[0]])
            for j, i in enumerate(range(args.warmup, total_steps)):
                pipe.submit_profiled(seeds_dev[i], sampler_seed[i], pipe_out[i % pipe_out.shape[0]], pairs[j])
            pipe.synchronize()
            torch.cuda.synchronize()
            insitu_ms = sum(he.elapsed_ms(a_[0], a_[1]) for a_ in pairs) / len(pairs)
            for a_ in pairs:
                he.destroy(a_[0]); he.destroy(a_[1])
        evs = {}
        for i in range(args.warmup, total_steps):
            arr = (ctypes.c_void_p * 10)()
            for j in range(10):
                arr[j] = he.create()
            evs[i] = arr
        stats = torch.zeros(total_steps, 4, dtype=torch.int64, device=dev)       # E2, |S1|, E1, |R1| per step
        flag = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        col_idx = torch.arange(k1, device=dev)[None, :]
        e = base
        L = e.layout
        first = b if concat else 0
        torch.cuda.synchronize()
        for i in range(args.warmup, total_steps):
            e.forward(seeds_dev[i], seed=sampler_seed[i], stage_events=evs[i])
            n_s1 = e._view(L.counters, 16, torch.int32)[8].long() + first          # device scalar
            cnt2 = e._view(L.cnt2, b, torch.int32)
            cnt1 = e._view(L.cnt1, L.max_s1, torch.int32)
            nbr1 = e._view(L.nbr1, L.max_s1 * k1, torch.int32).view(L.max_s1, k1)
            live = torch.arange(L.max_s1, device=dev) < n_s1
            valid = live[:, None] & (col_idx < cnt1[:, None])
            ids = torch.where(valid, nbr1, n).long()
            flag.zero_()
            flag.scatter_(0, ids.reshape(-1), 1)
            if concat or args.self_loop:
                s1 = e._view(L.s1_nodes, L.max_s1, torch.int32)
                flag.scatter_(0, torch.where(live, s1, n).long(), 1)
            stats[i, 0] = cnt2.sum()
            stats[i, 1] = n_s1
            stats[i, 2] = torch.where(live, cnt1, 0).sum()
            stats[i, 3] = flag[:n].sum()
        torch.cuda.synchronize()
        stage = np.zeros(5)
        gap = 0.0
        for i, arr in evs.items():
            for sidx in range(5):
                stage[sidx] += he.elapsed_ms(arr[2 * sidx], arr[2 * sidx + 1])
            gap += 0.5 * (he.elapsed_ms(arr[1], arr[2]) + he.elapsed_ms(arr[7], arr[8]))   # an event pair with nothing between
            for j in range(10):
                he.destroy(arr[j])
        stage /= args.steps
        gap /= args.steps
        if not split:
            stage[2] = 0.0
        alone_ms = float(stage[2]) if split else float(stage[3])
        kernel_ms = insitu_ms if insitu_ms is not None else alone_ms
        st = stats[args.warmup:].cpu().numpy().astype(np.float64)
        tot = l1 = own = 0.0
        for e2, n_s1, e1, n_r1 in st:
            t_, l_, g_ = algorithmic_bytes(d0, h1, h2, mult * d0, mult * h1, b, n_s1, e1, e2, n_r1)
            tot += t_
            # SURVEY 8(d) bytes of the work this kernel does: unique raw rows in + sampled ids in (+ counts); the [|S1|, D0]
            # means it writes for the contraction are a round trip the split design itself adds -- reported separately
            l1 += (g_ - 4 * d0 * n_s1) if split else l_
            own += g_ if split else l_
        tot /= args.steps
        l1 /= args.steps
        own /= args.steps
        sizes = st.mean(0)
        per_edge = 4 * d0 * sizes[2] + 4 * h1 * sizes[0]
        traffic = None
        traffic_source = None
        tfile = os.path.join(REPO, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))          # PMC passes are separate runs (profiles/collect.sh): only valid
                if tj.get("workload", workload) == workload and tj.get("engine_layout") == (relabel or "input") and split:
                    traffic = tj.get("layer1_hbm_bytes_per_launch")
                    traffic_source = ("profiles/traffic.json <- " + str(tj.get("source")) + " (builder's rocprofv3 --pmc pass of this "
                                      "command on an MI355X, committed; NOT collected by this run: PMC needs the profiler)")
            except Exception:
                traffic = None
        achieved = l1 / (kernel_ms * 1e-3) / 1e9
        roofline = {
            "bound": "hbm", "kernel": "gather_mean_rows_kernel / gather_mean_sliced* (layer-1 gather-mean)" if split
            else "layer_fused_kernel (layer 1: gather-mean + W1 contraction)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(achieved * 1e9 / HBM_PEAK, 4),
            "traffic": traffic, "traffic_source": traffic_source, "bytes_per_launch": round(l1),
            "bytes_note": "SURVEY 8(d) compulsory bytes of the kernel's work (unique raw rows + sampled ids + counts); its own write of "
                          "the [|S1|, D0] means (a round trip the split layer adds) is excluded here and included in bytes_per_launch_own",
            "bytes_per_launch_own": round(own),
            "kernel_ms": round(kernel_ms, 5),
            "kernel_ms_is": ("the launch's own start / stop HIP events (hipExtLaunchKernelGGL on the kernel's stream), measured in the running "
                             "role pipeline (overlap with the other stages' kernels included)" if insitu_ms is not None else
                             "the launch's own start / stop HIP events, one batch in flight"),
            "kernel_ms_alone": round(alone_ms, 5), "frac_alone": round(l1 / (alone_ms * 1e-3) / HBM_PEAK, 4),
            # the basis of round 1's figure (kernel alone, its own bytes incl. the write of the means): 0.388 there
            "frac_alone_own_bytes": round(own / (alone_ms * 1e-3) / HBM_PEAK, 4),
            "frac_note": ("`frac` / `kernel_ms` are measured with the other stages' kernels of three more batches co-resident (role "
                          "pipeline): they say what share of the chip the kernel gets there, not how good it is; the kernel alone: "
                          "`frac_alone` (SURVEY-8(d) bytes) and `frac_alone_own_bytes`" if insitu_ms is not None else ""),
            "empty_event_pair_ms": round(gap, 5),
            "stage_ms_alone": {"sample_outer": round(float(stage[0]), 5), "sample_inner": round(float(stage[1]), 5),
                               "layer1_gather": round(float(stage[2]), 5), "layer1_contract": round(float(stage[3]), 5),
                               "layer2": round(float(stage[4]), 5)},
            "forward_bytes": round(tot), "forward_GBps": round(tot / (ms_per_step * 1e-3) / 1e9, 1),
            "forward_frac": round(tot / (ms_per_step * 1e-3) / HBM_PEAK, 4),
            # "also quote against the 6.29 TB/s measured-copy ceiling"
            "copy_ceiling_GBps": 6290.0, "frac_vs_copy_ceiling": round(achieved / 6290.0, 4),
            "forward_frac_vs_copy_ceiling": round(tot / (ms_per_step * 1e-3) / 6.29e12, 4),
            "per_edge_gather_bytes": round(float(per_edge)),
            "mean_sizes": {"E2": round(float(sizes[0]), 1), "S1": round(float(sizes[1]), 1), "E1": round(float(sizes[2]), 1),
                           "R1": round(float(sizes[3]), 1)},
        }
        if args.config == 2:
            roofline["note"] = ("the 39 MB feature table of this configuration lives in the 256 MB Infinity Cache and a forward is ~20 us of "
                                "dependent launches: `frac` here is NOT an HBM-roofline figure, only bytes over time")
        mfile = os.path.join(REPO, "profiles", "mfma.json")
        if os.path.exists(mfile):
            try:
                mj = json.load(open(mfile))
                if mj.get("workload", workload) == workload:
                    roofline["mfma"] = mj.get("kernels")
                    roofline["mfma_source"] = ("profiles/mfma.json <- " + str(mj.get("source")) + " (builder's rocprofv3 --pmc pass, "
                                               "committed; not collected by this run)")
            except Exception:
                pass

    # ---- variant, reported beside the headline and never substituted for it: the same K steps with the engine keeping the
    #      CALLER's node order (no degree-sorted internal layout), same execution mode ----
    variants = None
    if rank == 0 and world == 1 and pipe is not None and relabel is not None and side_variants:
        role_streams = pipe.distinct_streams()       # the variant runs on the SAME role streams (hence hardware queues)
        pipe_threads = pipe.threads
        del pipe
        torch.cuda.synchronize()
        ekw_in = dict(ekw, relabel=None)
        pipe_in = RolePipeline(rowptr, col, table, w1, w2, k1, k2, batch=b, depth=args.depth, roles=args.roles, streams=role_streams,
                               threads=pipe_threads, window=args.window, **ekw_in)
        t_ph = time.perf_counter()
        while time.perf_counter() - t_ph < min(args.preheat_seconds, 0.2):
            pipe_in.submit_many(ph_seeds, ph_keys, pipe_out) if args.preheat_seconds > 0 else None
            pipe_in.flush()
            torch.cuda.synchronize()
        for i in range(args.warmup):
            pipe_in.submit(seeds_dev[i], sampler_seed[i], pipe_out[i % pipe_out.shape[0]])
        pipe_in.flush()
        torch.cuda.synchronize()
        t0v = time.perf_counter()
        for i in range(args.warmup, total_steps):
            pipe_in.submit(seeds_dev[i], sampler_seed[i], pipe_out[i % pipe_out.shape[0]])
        pipe_in.flush()
        torch.cuda.synchronize()
        el_v = time.perf_counter() - t0v
        variants = {"engine_layout_input": {"value": round(b * args.steps / el_v, 1), "ms_per_step": round(el_v / args.steps * 1e3, 5),
                                            "note": "engine keeps the caller's node order (TwoHopEngine(relabel=None)); same steps, same execution mode"}}
        del pipe_in
        # ---- a second variant, never substituted for `value` either: INFERENCE on a pre-transformed table (gcn encoder, fixed weights):
        #      Y = X . W1^T once (timed apart: it is paid per weight update, not per batch), then the unchanged engine on (Y, identity, W2)
        #      -- the same embeddings up to fp32 rounding (checked against the oracle-gated forward below) ----
        if not concat and not args.unfused and d0 % 4 == 0 and h1 % 4 == 0:
            from sage355.engine import pretransform_table
            torch.cuda.synchronize()
            t_pt = time.perf_counter()
            y_tab, eye1 = pretransform_table(table, w1)
            torch.cuda.synchronize()
            prep_ms = (time.perf_counter() - t_pt) * 1e3
            pipe_pt = RolePipeline(rowptr, col, y_tab, eye1, w2, k1, k2, batch=b, depth=args.depth, roles=args.roles, streams=role_streams,
                                   threads=pipe_threads, window=args.window, **ekw)
            t_ph = time.perf_counter()
            while time.perf_counter() - t_ph < min(args.preheat_seconds, 0.2):
                pipe_pt.submit_many(ph_seeds, ph_keys, pipe_out) if args.preheat_seconds > 0 else None
                pipe_pt.flush()
                torch.cuda.synchronize()
            for i in range(args.warmup):
                pipe_pt.submit(seeds_dev[i], sampler_seed[i], pipe_out[i % pipe_out.shape[0]])
            pipe_pt.flush()
            torch.cuda.synchronize()
            t0v = time.perf_counter()
            for i in range(args.warmup, total_steps):
                pipe_pt.submit(seeds_dev[i], sampler_seed[i], pipe_out[i % pipe_out.shape[0]])
            pipe_pt.flush()
            torch.cuda.synchronize()
            el_p = time.perf_counter() - t0v
            diff = None
            if gate_out is not None:                      # the last timed batch is the gate batch: same seeds, same key, same sampled sets
                got_p = pipe_out[gate_i % pipe_out.shape[0]]
                diff = float(((got_p - gate_out).abs() / gate_out.abs().amax(1, keepdim=True).clamp_min(1e-30)).max().item())
                if not diff <= 1e-5:
                    raise SystemExit(f"pre-transformed variant: max |out - default forward| / rowmax = {diff:.3e}")
            variants["pretransformed_table_inference"] = {
                "value": round(b * args.steps / el_p, 1), "ms_per_step": round(el_p / args.steps * 1e3, 5),
                "prepare_ms_per_weight_update": round(prep_ms, 3), "max_rel_diff_vs_oracle_gated_forward": diff,
                "note": "INFERENCE at fixed weights, NOT the BASELINE forward's work per batch: layer 1's contraction is done once per weight "
                        f"update for the whole table (Y = X . W1^T, [{n}, {h1}]; sage355.engine.pretransform_table) and the timed forward gathers "
                        f"{h1}-float rows of Y instead of {d0}-float rows of X; same sampled sets, same embeddings to fp32 rounding"}
            del pipe_pt, y_tab
        pipe = None
        was_pipe = True
    else:
        was_pipe = pipe is not None

    result = None
    mark(rank, "kernel accounting / variants done")
    # ---- CPU side by side: the reference-faithful restatement on this box's host cores ----
    cpu_baseline = None
    if rank == 0 and world == 1 and with_cpu:
        cpu_baseline = cpu_port_baseline(graph, table.cpu(), w1.cpu(), w2.cpu(), candidates, k1, k2, concat, args.cpu_seconds)

    if rank == 0:
        if was_pipe:
            execution = f"role pipeline {args.roles} (stages S/G/D/L on HIP streams, hipEvent hand-offs), {args.depth} batches in flight"
            if pipe_threaded:
                execution += ", one host enqueue thread per role stream" + (f", host run-ahead window {args.window}" if args.window else "")
            if exec_mode == "pipegraph":
                execution += f"; the {args.steps} timed batches captured as ONE hipGraph over the role streams (one graph launch per timed region)"
        elif exec_mode == "replay":
            execution = f"hipGraph replay from a device batch queue, {nstreams} forwards in flight"
        else:
            execution = f"host-enqueued sage_forward2, {nstreams} forwards in flight"
        result = {
            "value": round(value, 1), "ms_per_step": round(ms_per_step, 5), "workload": workload, "execution": execution,
            "parity_err": parity_err, "timed_check": timed_check, "roofline": roofline, "cpu_baseline": cpu_baseline, "variants": variants,
            "setup_seconds": round(setup_seconds, 1),
            "config": {"workload": workload,
                       "batch_per_gpu": b, "global_batch": b * world, "fanout": [k1, k2], "encoder_mode": args.mode,
                       "execution": execution, "host_enqueue_ms_per_step": round(host_enqueue_ms, 5),
                       "host_submit_ms_per_step": round(host_submit_ms, 5), "fused_layers": not args.unfused,
                       # N > 1: what the closing barrier itself adds to the K-step region (`value` includes it, as the contract's bracket does)
                       "closing_fence": None if dist is None else {
                           "slowest_rank_done_ms_per_step": round(own_done / args.steps * 1e3, 5),
                           "barrier_us_in_timed_region": round((elapsed - own_done) * 1e6, 1),
                           "note": "ms_per_step / value are taken AFTER synchronize + dist.barrier + synchronize on every rank (max over ranks); "
                                   "slowest_rank_done is the max over ranks of the time at which a rank's own K steps had completed"},
                       "preheat": f"{preheat_forwards} untimed forwards on throw-away batches before the {args.warmup} warm-up steps (GPU clock ramp)",
                       "node_order": args.node_order if args.config != 2 else "original",
                       "engine_layout": (relabel or "input") + (" (internal: rows by descending degree; seeds arrive in the generator's ids and are "
                                                                "translated inside every timed forward)" if relabel else ""),
                       "table_layout": (f"engine-internal: row-major [N, {d0}] + a slice-major copy [{d0 // 32}][N][32] read by the column-sliced "
                                        "layer-1 gather (128-byte slices, one per XCD)" if (not concat and d0 % 32 == 0 and d0 >= 64 and
                                        os.environ.get("SAGE_TABLE_SLICED", "1") != "0" and bool(base_split)) else "row-major"),
                       "contraction": "bf16x3-split MFMA (fp32-accurate: x.w from the three bf16 terms of x and of w; weight planes prepared "
                                      "once per weight update by sage_prepare_weights)",
                       "batches_per_replay": bpr if exec_mode == "replay" else 1},
        }
    # this workload's device objects go before the next one is built (configs[3]: 8.6 GB table + its slice-major copy)
    pipe = base = e = None
    engines, outs = [], []
    del table, rowptr, col, seeds_dev
    torch.cuda.empty_cache()
    return result


def cpu_port_baseline(graph, table, w1, w2, candidates, k1, k2, concat, budget_s):
    """oracle/ref_dense.py (Python set sampling + dense mask + div + mm: the reference's algorithm,
    pinned to it by tests/golden) timed on the host cores.  Bounded sample: B = 256 forwards until
    `budget_s` seconds of CPU work (the dense mask of a B = 4096 batch is 10 GB)."""
    from oracle import ref_dense

    class LazyAdj(dict):          # adjacency sets built on first touch; warmed before timing
        def __missing__(self, v):
            s = set(int(x) for x in graph.neighbors(v))
            self[v] = s
            return s

    adj = LazyAdj()
    bs = 256
    rs = np.random.default_rng(12345)
    batches = [[int(x) for x in rs.choice(candidates, bs, replace=False)] for _ in range(64)]

    def forward(i):
        random.seed(1000 + i)
        with torch.no_grad():
            return ref_dense.two_hop_forward(batches[i], adj, adj, table, w1, w2, k1, k2, not concat)

    # host cores actually usable: the affinity mask / cgroup quota, not the machine's core count
    # (256 intra-op threads on a 16-core share ran 20x slower than 16)
    usable = usable_host_cores()
    forward(0)                       # builds the sets this batch touches
    best = None
    for threads in sorted({min(usable, 8), min(usable, 16), min(usable, 32), usable}):
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        forward(0)
        dt = time.perf_counter() - t0
        if best is None or dt < best[1]:
            best = (threads, dt)
    torch.set_num_threads(best[0])
    reps, dt = 0, 0.0
    while reps < len(batches) and dt < budget_s:
        forward(reps)              # untimed: builds the adjacency sets this batch touches (the reference has them prebuilt)
        t0 = time.perf_counter()
        forward(reps)              # timed: same seed -> same sets, now cached
        dt += time.perf_counter() - t0
        reps += 1
    ratio = None
    try:        # how far the port is from the real reference's speed, measured where the reference can be imported
        ratio = json.load(open(os.path.join(REPO, "tests", "golden", "cpu_port_vs_reference.json")))["port_over_reference_time"]
    except Exception:
        pass
    return {"value": round(bs * reps / dt, 1), "unit": "embeddings/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": cpu_model(), "host_cores_usable": usable,
            "sample": f"{reps} forwards of B={bs} seeds on the same graph/features/weights/fanout, "
                      f"{dt:.1f} s of CPU work, oracle/ref_dense.py (dense-mask algorithm of the reference)",
            "ms_per_forward": round(dt / reps * 1e3, 2),
            "port_over_reference_time": ratio,
            "port_over_reference_source": "tests/golden/cpu_port_vs_reference.json (build container, reference imported; "
                                          "Pubmed + R-MAT-131k workloads of SURVEY.md section 6, outputs bit-identical)"}


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
