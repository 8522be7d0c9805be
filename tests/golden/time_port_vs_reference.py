#!/usr/bin/env python3
"""Time oracle/ref_dense.py next to the IMPORTED reference on the workloads of SURVEY.md section 6
(build container only: /root/reference does not travel).  Writes tests/golden/cpu_port_vs_reference.json,
which bench.py's `cpu_baseline` carries as `port_over_reference_time` -- the GPU box can only time the
port, and this ratio says how far the port is from the reference's own speed (target 1.0 +- 0.1).

    python tests/golden/time_port_vs_reference.py

Same graph, features, weights, fanout, seed batches and `random.seed` for both sides; forward only,
torch.no_grad(), all host threads; the two sides are interleaved batch by batch so that machine drift
hits both equally; outputs are compared bit for bit (same sets, same arithmetic).
"""
import io
import json
import os
import platform
import random
import sys
import time
import warnings
from contextlib import redirect_stdout

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")          # the reference's `graphsage`
sys.path.insert(1, os.path.join(REPO, "graphsage-simple_amd"))
sys.path.insert(2, REPO)
warnings.filterwarnings("ignore")

from graphsage.aggregators import MeanAggregator  # noqa: E402  (the reference)
from graphsage.encoders import Encoder            # noqa: E402  (the reference)
from oracle import ref_dense                      # noqa: E402
from sage355 import graph as G                    # noqa: E402


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def reference_stack(table, adj, w1, w2, k1, k2, gcn):
    n, d0 = table.shape
    feats = torch.nn.Embedding(n, d0)
    feats.weight = torch.nn.Parameter(table, requires_grad=False)
    with redirect_stdout(io.StringIO()):
        agg1 = MeanAggregator(feats, cuda=False)
        enc1 = Encoder(feats, d0, w1.shape[0], adj, agg1, num_sample=k1, gcn=gcn, cuda=False)
        agg2 = MeanAggregator(lambda nodes: enc1(nodes).t(), cuda=False)
        enc2 = Encoder(lambda nodes: enc1(nodes).t(), enc1.embed_dim, w2.shape[0], adj, agg2, num_sample=k2, base_model=enc1,
                       gcn=gcn, cuda=False)
    with torch.no_grad():
        enc1.weight.copy_(w1)
        enc2.weight.copy_(w2)
    return enc2


def time_case(name, graph, d0, h1, h2, k1, k2, gcn, batch, reps, feature_kind):
    gen = torch.Generator().manual_seed(0)
    n = graph.num_nodes
    if feature_kind == "pubmed":
        table = torch.rand(n, d0, generator=gen) * (torch.rand(n, d0, generator=gen) < 0.1)
    else:
        table = torch.randn(n, d0, generator=gen)
    mult = 1 if gcn else 2
    w1 = (torch.rand(h1, mult * d0, generator=gen) * 2 - 1) * float(np.sqrt(6.0 / (h1 + mult * d0)))
    w2 = (torch.rand(h2, mult * h1, generator=gen) * 2 - 1) * float(np.sqrt(6.0 / (h2 + mult * h1)))
    adj = graph.to_adj_lists()
    deg = graph.degrees()
    cand = np.nonzero(deg > 0)[0]
    rs = np.random.default_rng(7)
    batches = [[int(x) for x in rs.choice(cand, batch, replace=False)] for _ in range(reps + 1)]
    enc2 = reference_stack(table, adj, w1, w2, k1, k2, gcn)
    t_ref = t_port = 0.0
    same = True
    for i, nodes in enumerate(batches):
        random.seed(100 + i)
        t0 = time.perf_counter()
        with torch.no_grad():
            a = enc2(nodes)
        ta = time.perf_counter() - t0
        random.seed(100 + i)
        t0 = time.perf_counter()
        with torch.no_grad():
            b = ref_dense.two_hop_forward(nodes, adj, adj, table, w1, w2, k1, k2, gcn)
        tb = time.perf_counter() - t0
        same = same and bool(torch.equal(a, b))
        if i:                     # first pass warms both
            t_ref += ta
            t_port += tb
    res = {"case": name, "batch": batch, "forwards": reps, "reference_ms": round(t_ref / reps * 1e3, 2),
           "port_ms": round(t_port / reps * 1e3, 2), "port_over_reference_time": round(t_port / t_ref, 3),
           "outputs_bit_identical": same}
    print(res, flush=True)
    return res


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    z = np.load(os.path.join(HERE, "pubmed_topology.npz"))
    pubmed = G.CSRGraph(z["rowptr"], z["col"], len(z["rowptr"]) - 1)
    rmat = G.rmat_graph(17, 2_000_000, seed=0)           # R-MAT 131 k nodes / 3.7 M directed nnz
    cases = [
        time_case("pubmed gcn 10/25 D0=500 H=50/128", pubmed, 500, 50, 128, 10, 25, True, 256, 12, "pubmed"),
        time_case("pubmed concat 10/25 D0=500 H=50/128", pubmed, 500, 50, 128, 10, 25, False, 256, 8, "pubmed"),
        time_case("rmat-131k gcn 15/25 D0=256 H=128/128", rmat, 256, 128, 128, 15, 25, True, 256, 8, "randn"),
        time_case("rmat-131k concat 15/25 D0=256 H=128/128", rmat, 256, 128, 128, 15, 25, False, 256, 6, "randn"),
    ]
    tot_ref = sum(c["reference_ms"] * c["forwards"] for c in cases)
    tot_port = sum(c["port_ms"] * c["forwards"] for c in cases)
    out = {"what": "oracle/ref_dense.py timed next to the imported reference (graphsage/encoders.py + aggregators.py), forward only",
           "cpu_model": cpu_model(), "threads": torch.get_num_threads(), "torch": torch.__version__,
           "python": platform.python_version(), "cases": cases,
           "port_over_reference_time": round(tot_port / tot_ref, 3),
           "generated_by": "tests/golden/time_port_vs_reference.py"}
    json.dump(out, open(os.path.join(HERE, "cpu_port_vs_reference.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
