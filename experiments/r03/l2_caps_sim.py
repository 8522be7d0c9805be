"""Round 3: which effective L2 capacity reproduces the MEASURED 41 % hit / 214 MB past L2 of the layer-1 gather, and what
would store pollution, an LFU-ideal cache and candidate S1 orders do?  CPU only (numpy), config-3 shape, degree layout.

    python experiments/r03/l2_caps_sim.py
"""
import os, sys
from collections import OrderedDict
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "graphsage-simple_amd"))
from sage355.graph import rmat_graph, relabel_by_degree  # noqa: E402


def sample_rows(g, nodes, k, rng):
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


def main():
    g = rmat_graph(20, 16_000_000, seed=0, cache_dir="/tmp/sage_cache", accel=None)
    g = relabel_by_degree(g)[0]
    deg = g.degrees()
    rng = np.random.default_rng(1)
    cand = np.nonzero(deg > 0)[0]
    res = {}
    for trial in range(2):
        seeds = rng.choice(cand, size=4096, replace=False)
        hop2 = sample_rows(g, seeds, 25, rng)
        s1 = np.unique(np.concatenate(hop2))
        rng.shuffle(s1)
        hop1 = sample_rows(g, s1, 15, rng)
        e1 = sum(len(x) for x in hop1)
        allsrc = np.concatenate(hop1)
        uniq = len(np.unique(allsrc))
        print(f"trial {trial}: |S1| = {len(s1)}, E1 = {e1}, unique = {uniq}")
        # XCD halves: chunks of 4 rows alternate between the two XCDs of a slice
        idx = np.arange(len(s1))
        mine = (idx // 4) % 2 == 0
        half = [hop1[i] for i in idx[mine]]
        eh = sum(len(x) for x in half)
        today = np.concatenate(half)
        uh = len(np.unique(today))
        mb = 8 * 256 / 1e6
        print(f"  one XCD: {eh} slice reads, {uh} unique -> floor {uh * mb:.1f} MB; per-edge {eh * mb:.1f} MB")
        for cap in (1024, 2048, 4096, 6144, 8192, 12288, 16384, 32768):
            m = lru_misses(today, cap)
            print(f"  LRU {cap:6d} slices ({cap * 256 // 1024:5d} KiB): hit {1 - m / eh:5.1%} -> {m * mb:6.1f} MB")
        # store pollution: every destination row's 256-B output slice is allocated in L2 after its reads (nt / plain stores keep the line)
        st = []
        for i, x in enumerate(half):
            st.extend(x.tolist())
            st.append(-1 - i)
        for cap in (4096, 8192, 16384):
            m = lru_misses(st, cap) - len(half)
            print(f"  LRU {cap:6d} + output lines allocated: hit {1 - m / eh:5.1%} -> {m * mb:6.1f} MB")
        # LFU-ideal: the top-K ids by degree pinned (degree layout: id = popularity rank), everything else streams through
        for K in (4096, 8192, 12288, 16384):
            hot = today < K
            m = len(np.unique(today[hot])) + int((~hot).sum())
            m2 = len(np.unique(today[hot])) + lru_misses(today[~hot], 16384 - K) if K < 16384 else m
            print(f"  pinned top {K:6d}: cold uncached -> {m * mb:6.1f} MB; cold through LRU of the rest -> {m2 * mb:6.1f} MB")
        # 128-B slices: one XCD sees every row
        allrows = np.concatenate(hop1)
        for cap in (8192, 16384, 32768):
            m = lru_misses(allrows, cap)
            print(f"  128-B slices, LRU {cap} lines-pairs: hit {1 - m / e1:5.1%} -> {m * 8 * 128 / 1e6:6.1f} MB")


if __name__ == "__main__":
    main()
