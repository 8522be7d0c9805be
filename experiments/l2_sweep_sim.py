"""How many row-slice fetches must leave an XCD's L2 in the layer-1 gather, by the ORDER the edges are walked in?

CPU simulation (numpy), BASELINE configs[2] shape: R-MAT 2^20 / 16 M edges, degree layout, B = 4096, fanout 15/25.
An XCD owns one 256-B column slice (4 slices, two XCDs per slice, each with half of the destination rows) and 4 MiB of
L2 = 16 Ki slices.  Compared per XCD:
  * today's order: destination rows in S1 order, all neighbours of a row together (LRU);
  * sweeps: every lane group owns a few destination rows and walks ITS edges in P buckets of ascending source id
    (P = 1: today; P = inf: sorted), the groups advancing in lock step with a jitter of +-J of their list;
  * the floor: unique slices per XCD (an infinite L2), and unique rows on the chip.

    python experiments/l2_sweep_sim.py [--scale 20] [--edges 16000000]
"""
import argparse
import os
import sys
from collections import OrderedDict

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "graphsage-simple_amd"))
from sage355.graph import rmat_graph, relabel_by_degree  # noqa: E402


def sample_rows(g, nodes, k, rng):
    """k distinct neighbours (all when degree <= k) of every node -> list of arrays."""
    out = []
    for v in nodes:
        a, b = g.rowptr[v], g.rowptr[v + 1]
        d = b - a
        if d <= k:
            out.append(g.col[a:b].astype(np.int64))
        else:
            out.append(g.col[a + rng.choice(d, size=k, replace=False)].astype(np.int64))
    return out


def lru_misses(stream, capacity):
    cache = OrderedDict()
    miss = 0
    for x in stream:
        if x in cache:
            cache.move_to_end(x)
        else:
            miss += 1
            cache[x] = True
            if len(cache) > capacity:
                cache.popitem(last=False)
    return miss


def sweep_stream(lists, order_key, rng, jitter):
    """lists: per lane group, the source ids of its edges.  Every group walks its list in `order_key` order; groups
    advance in lock step (edge i of every group at step i) with a per-group offset of up to `jitter` of the list."""
    walks = []
    for src in lists:
        walks.append(src[np.argsort(order_key(src), kind="stable")])
    n = max(len(w) for w in walks)
    steps = []
    for w in walks:
        off = int(rng.integers(-int(jitter * n), int(jitter * n) + 1)) if jitter > 0 else 0
        steps.append(np.arange(len(w)) * (n / max(len(w), 1)) + off)
    t = np.concatenate(steps)
    s = np.concatenate(walks)
    return s[np.argsort(t, kind="stable")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=20)
    ap.add_argument("--edges", type=int, default=16_000_000)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--k1", type=int, default=15)
    ap.add_argument("--k2", type=int, default=25)
    ap.add_argument("--l2-slices", type=int, default=16384, help="4 MiB / 256 B")
    ap.add_argument("--groups", type=int, default=32 * 8 * 4 // 1, help="lane groups per XCD (32 CUs x 8 waves x 4 groups)")
    args = ap.parse_args()
    g = rmat_graph(args.scale, args.edges, seed=0, cache_dir="/tmp/sage_cache", accel=None)
    g = relabel_by_degree(g)[0]
    rng = np.random.default_rng(1)
    seeds = rng.choice(g.num_nodes, size=args.batch, replace=False)
    hop2 = sample_rows(g, seeds, args.k2, rng)
    s1 = np.unique(np.concatenate([seeds] + hop2))
    rng.shuffle(s1)                                    # S1's order is the hash table's: arbitrary
    hop1 = sample_rows(g, s1, args.k1, rng)
    e1 = sum(len(x) for x in hop1)
    allsrc = np.concatenate(hop1)
    uniq = len(np.unique(allsrc))
    print(f"|S1| = {len(s1)}, E1 = {e1}, unique source rows = {uniq} ({uniq * 1024 / 1e6:.1f} MB), per-edge {e1 * 1024 / 1e6:.1f} MB")
    # one XCD: half of the destination rows (the other half and the other 3 slices behave alike)
    half = hop1[0::2]
    eh = sum(len(x) for x in half)
    uh = len(np.unique(np.concatenate(half)))
    scale_mb = 8 * 256 / 1e6                           # misses of one XCD -> MB past L2 on the chip

    def report(name, miss):
        print(f"  {name:58s} misses {miss:7d} of {eh}  hit {1 - miss / eh:5.1%}  -> {miss * scale_mb:6.1f} MB past L2")

    report("floor: unique slices of the XCD (infinite L2)", uh)
    today = np.concatenate(half)
    report("today: rows in S1 order, LRU 16 Ki slices", lru_misses(today, args.l2_slices))
    report("today, LRU 12 Ki (ids / stores / other kernels share L2)", lru_misses(today, args.l2_slices * 3 // 4))
    ng = args.groups
    lists = [np.concatenate(half[i::ng]) if len(half[i::ng]) else np.zeros(0, np.int64) for i in range(ng)]
    lists = [x for x in lists if len(x)]
    print(f"  sweeps: {len(lists)} lane groups, {np.mean([len(x) for x in lists]):.0f} edges each")
    n = g.num_nodes
    for P in (2, 4, 8, 16, 64, 0):
        for jit in (0.0, 0.05, 0.15):
            if P == 0:
                key = lambda s: s
                name = "sorted by source id"
            else:
                # buckets of equal edge mass: boundaries at the quantiles of this batch's sources (a fixed power-law
                # spacing in the degree-sorted id would do the same)
                qs = np.quantile(allsrc, np.linspace(0, 1, P + 1)[1:-1])
                key = lambda s, qs=qs: np.searchsorted(qs, s, side="right")
                name = f"{P} buckets of ascending id"
            st = sweep_stream(lists, key, rng, jit)
            for cap in (args.l2_slices, args.l2_slices * 3 // 4):
                report(f"{name}, jitter {jit:.2f}, LRU {cap // 1024} Ki", lru_misses(st, cap))


if __name__ == "__main__":
    main()
