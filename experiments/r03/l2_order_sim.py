"""Round 3, slice-major table (one XCD sees EVERY row's 128-byte slice; its L2 holds 32 Ki of them): would an ORDER of the layer-1
destination rows cut the gather's reads (measured 171 MB = 52 % hits; compulsory 108 MB)?  Ideal-LRU replay of one XCD's stream for
candidate orders of S1.  CPU only.    python experiments/r03/l2_order_sim.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "graphsage-simple_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sage355.graph import rmat_graph, relabel_by_degree  # noqa: E402
from l2_caps_sim import sample_rows, lru_misses           # noqa: E402


def stream_of(rows, order, round_rows=0):
    """rows[i] = neighbour ids of destination i; order = permutation.  round_rows > 0: the kernel's rounds -- that many rows are in
    flight together, their loads interleaved trip by trip (load j of every row of the round before load j + 1)."""
    if not round_rows:
        return np.concatenate([rows[i] for i in order])
    out = []
    for lo in range(0, len(order), round_rows):
        grp = [rows[i] for i in order[lo:lo + round_rows]]
        k = max(len(x) for x in grp)
        for j in range(k):
            out.append(np.array([x[j] for x in grp if len(x) > j], dtype=np.int64))
    return np.concatenate(out)


def main():
    g = relabel_by_degree(rmat_graph(20, 16_000_000, seed=0, cache_dir="/tmp/sage_cache", accel=None))[0]
    rng = np.random.default_rng(1)
    cand = np.nonzero(g.degrees() > 0)[0]
    cap = 32768
    mb = 8 * 128 / 1e6
    prev_tail = None
    for trial in range(2):
        seeds = rng.choice(cand, size=4096, replace=False)
        s1 = np.unique(np.concatenate(sample_rows(g, seeds, 25, rng)))
        rng.shuffle(s1)
        rows = sample_rows(g, s1, 15, rng)
        e1 = sum(len(x) for x in rows)
        uniq = len(np.unique(np.concatenate(rows)))
        print(f"trial {trial}: |S1| = {len(s1)}, E1 = {e1}, unique = {uniq} -> compulsory {uniq * mb:.1f} MB, per edge {e1 * mb:.1f} MB")
        n = len(rows)
        key_min = np.array([x.min() if len(x) else 0 for x in rows])
        key_max = np.array([x.max() if len(x) else 0 for x in rows])
        key_2max = np.array([np.sort(x)[-2] if len(x) > 1 else 0 for x in rows])
        h = (np.arange(g.num_nodes, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(40)
        key_minhash = np.array([h[x].min() if len(x) else 0 for x in rows])
        orders = {
            "frontier order (random)": np.arange(n),
            "by node id of the destination": np.argsort(s1, kind="stable"),
            "by hottest neighbour (min id)": np.argsort(key_min, kind="stable"),
            "by coldest neighbour (max id)": np.argsort(key_max, kind="stable"),
            "by second coldest neighbour": np.argsort(key_2max, kind="stable"),
            "by MinHash of the neighbour set": np.argsort(key_minhash, kind="stable"),
        }
        for name, order in orders.items():
            for rr in (0, 6144):
                st = stream_of(rows, order, rr)
                warm = lru_misses(np.concatenate([prev_tail, st]), cap) - lru_misses(prev_tail, cap) if prev_tail is not None else lru_misses(st, cap)
                print(f"  {name:34s} {'rounds of 6144 rows' if rr else 'row after row':20s}: {warm * mb:6.1f} MB ({1 - warm / e1:5.1%} hits)")
        prev_tail = stream_of(rows, np.arange(n))[-200000:]      # the next batch's gather starts on this batch's L2 contents


if __name__ == "__main__":
    main()
