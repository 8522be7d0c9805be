"""CPU tests of the host-side logic around the path: graph ingestion, the stand-in dataset,
the bench's byte accounting, and the harness pieces that do not need a GPU."""
import os

import numpy as np
import pytest
import torch

from sage355 import graph as G
from sage355.datasets import standin_citation
from util import GOLDEN_DIR


def test_adj_lists_csr_round_trip_and_canonical_form():
    adj = {0: {3, 1, 2}, 1: {0}, 2: {0, 2}, 3: {0}, 5: set()}      # node 4 missing, 5 empty, self loop on 2
    g = G.csr_from_adj_lists(adj)
    assert g.num_nodes == 6 and g.nnz == 7
    assert g.neighbors(0).tolist() == [1, 2, 3] and g.neighbors(2).tolist() == [0, 2]      # sorted rows, self loop kept
    assert g.neighbors(4).size == 0 and g.neighbors(5).size == 0
    back = g.to_adj_lists()
    for v in range(6):
        assert back[v] == adj.get(v, set())
    assert g.rowptr.dtype == np.int64 and g.col.dtype == np.int32


def test_edges_are_symmetrised_deduplicated_and_self_loops_kept():
    g = G.csr_from_edges([0, 0, 1, 2, 2], [1, 1, 0, 2, 3], 4)          # duplicate edge, reversed duplicate, self loop
    assert g.to_adj_lists() == {0: {1}, 1: {0}, 2: {2, 3}, 3: {2}}
    g2 = G.csr_from_edges([0, 2], [1, 2], 3, drop_self_loops=True)
    assert g2.to_adj_lists() == {0: {1}, 1: {0}, 2: set()}


def test_edge_list_readers(tmp_path):
    """model.py:303-310 (tab/space pairs) and model.py:449-458 (Pubmed .tab) formats; ids by first appearance."""
    p = tmp_path / "x.cites"
    p.write_text("35\t1033\n35\t103482\n1033 103482\n9 9\n")
    g, names = G.read_edge_list(str(p))
    assert names.tolist() == ["35", "1033", "103482", "9"]
    assert g.to_adj_lists() == {0: {1, 2}, 1: {0, 2}, 2: {0, 1}, 3: {3}}
    q = tmp_path / "p.tab"
    q.write_text("DIRECTED\tcites\nNO_FEATURES\n1\tpaper:19127292\t|\tpaper:17363749\n2\tpaper:19668377\t|\tpaper:19127292\n")
    g, names = G.read_edge_list(str(q), fmt="pubmed")
    assert names.tolist() == ["19127292", "17363749", "19668377"]
    assert g.to_adj_lists() == {0: {1, 2}, 1: {0}, 2: {0}}


def test_topology_fixtures_match_the_survey_statistics():
    """Cora 2708 nodes / sum of degrees 10556 / max 168; Pubmed 19717 / 88651 / 171."""
    for name, n, nnz, dmax in (("cora", 2708, 10556, 168), ("pubmed", 19717, 88651, 171)):
        z = np.load(os.path.join(GOLDEN_DIR, f"{name}_topology.npz"))
        g = G.CSRGraph(z["rowptr"], z["col"], len(z["rowptr"]) - 1)
        deg = g.degrees()
        assert (g.num_nodes, g.nnz, int(deg.max()), int(deg.min())) == (n, nnz, dmax, 1)
        src = np.repeat(np.arange(n), deg)
        fwd = set(zip(src.tolist(), g.col.tolist()))
        assert all((b, a) in fwd for a, b in list(fwd)[:2000])      # symmetric


def test_rmat_generator_is_deterministic_symmetric_and_loop_free(tmp_path):
    a = G.rmat_graph(12, 40_000, seed=3)
    b = G.rmat_graph(12, 40_000, seed=3, cache_dir=str(tmp_path))
    c = G.rmat_graph(12, 40_000, seed=3, cache_dir=str(tmp_path))     # from the cache
    for g in (b, c):
        assert np.array_equal(a.rowptr, g.rowptr) and np.array_equal(a.col, g.col)
    src = np.repeat(np.arange(a.num_nodes), a.degrees())
    assert not (src == a.col).any()
    key = set((src * a.num_nodes + a.col).tolist())
    assert all((int(d) * a.num_nodes + int(s)) in key for s, d in zip(src[:3000], a.col[:3000]))
    assert a.degrees().max() > 50 * max(1, np.median(a.degrees()))      # power-law-ish skew
    t = G.truncate_nodes(a, 3000)
    assert t.num_nodes == 3000 and t.col.max() < 3000 and t.nnz < a.nnz


def test_standin_dataset_is_deterministic_and_learnable_in_principle():
    z = np.load(os.path.join(GOLDEN_DIR, "cora_topology.npz"))
    g = G.CSRGraph(z["rowptr"], z["col"], len(z["rowptr"]) - 1)
    f1, l1 = standin_citation(g, seed=0)
    f2, l2 = standin_citation(g, seed=0)
    assert np.array_equal(f1, f2) and np.array_equal(l1, l2)
    assert f1.shape == (2708, 1433) and l1.shape == (2708, 1) and set(np.unique(l1)) == set(range(7))
    assert set(np.unique(f1)) == {0.0, 1.0} and 10 < f1.sum(1).mean() < 18
    # labels follow the topology: most edges join same-label nodes
    src = np.repeat(np.arange(g.num_nodes), g.degrees())
    assert (l1[src, 0] == l1[g.col, 0]).mean() > 0.8


def test_bench_byte_accounting_matches_the_survey_worked_example():
    """B=4096, E2=39095, |S1|=23657, E1=339183, |R1|=106339 -> 136.9 MB per forward."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(GOLDEN_DIR), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    total, layer1, gather1 = bench.algorithmic_bytes(256, 128, 128, 256, 128, 4096, 23657, 339183, 39095, 106339)
    assert abs(total - 136.9e6) < 0.01 * 136.9e6        # the formula, term for term; the survey rounds its example
    assert layer1 < total and gather1 == 4 * 256 * 106339 + 4 * 256 * 23657 + 4 * 339183 + 4 * 23657


def test_training_split_and_batching_follow_the_reference():
    """model.py:229-234 (10/10/80 split of a seeded permutation) and model.py:244 (the `max` batches)."""
    np.random.seed(1)
    perm = np.random.permutation(2708)
    assert len(perm[:270]) == 270 and len(perm[270:541]) == 271 and len(perm[541:]) == 2167
    train_num, bs = 2167, 128
    ref_sizes = [len(list(range(train_num))[b:max(train_num, b + bs)]) for b in range(0, train_num, bs)]
    assert ref_sizes[:3] == [2167, 2039, 1911] and ref_sizes[-1] == 119 and len(ref_sizes) == 17
    plain = [min(train_num, b + bs) - b for b in range(0, train_num, bs)]
    assert plain[0] == 128 and plain[-1] == 119 and sum(plain) == train_num


def test_relabel_by_degree_is_an_isomorphism_in_descending_degree_order():
    """Ingestion option behind bench.py --node-order degree / TwoHopEngine(relabel="degree")."""
    from sage355.graph import relabel_by_degree, rmat_graph
    g = rmat_graph(11, 30_000, seed=3)
    h, new_of_old = relabel_by_degree(g)
    assert h.num_nodes == g.num_nodes and h.nnz == g.nnz
    assert sorted(new_of_old.tolist()) == list(range(g.num_nodes))
    deg = h.degrees()
    assert (np.diff(deg) <= 0).all() and np.array_equal(deg[new_of_old], g.degrees())
    for v in (0, 5, 77, g.num_nodes - 1):
        assert set(new_of_old[g.neighbors(v)].tolist()) == set(h.neighbors(int(new_of_old[v])).tolist())
        assert (np.diff(h.neighbors(int(new_of_old[v]))) > 0).all()      # rows stay sorted


def test_host_ids_are_range_checked_before_the_int32_cast():
    """a negative / too large / wrapping id must raise on the host (the reference raises IndexError from
    the nn.Embedding lookup), never reach a kernel that walks rowptr[] with it."""
    from sage355 import native, ops
    ok = np.array([0, 5, 99], dtype=np.int64)
    ops.check_id_range(ok, 100)
    ops.check_id_range(np.zeros(0, np.int64), 100)
    ops.check_id_range(ok, None)
    for bad in ([-1, 3], [100], [2 ** 32 + 5]):          # the last one would wrap to 5 under a blind int32 cast
        with pytest.raises(native.SageError):
            ops.check_id_range(np.asarray(bad, dtype=np.int64), 100)
    with pytest.raises(native.SageError):
        ops.as_ids([3, 2 ** 31], "cpu", 10)
    assert ops.as_ids([3, 9], "cpu", 10).dtype == torch.int32


def test_adjacency_cache_detects_in_place_mutation_and_is_bounded(monkeypatch):
    """the device-CSR cache was keyed by id(adj_lists) only (stale after an in-place edit, never evicted)."""
    from collections import defaultdict
    from sage355 import encoders
    calls = []

    class FakeGraph:
        def __init__(self, adj):
            self.num_nodes = (max(adj) + 1) if adj else 0
            self.rowptr = np.zeros(self.num_nodes + 1, np.int64)

        def to(self, device):
            return torch.zeros(self.num_nodes + 1, dtype=torch.int64), torch.zeros(1, dtype=torch.int32)

    def fake_csr(adj, n):
        calls.append(sum(len(s) for s in adj.values()))
        return FakeGraph(adj)

    monkeypatch.setattr(encoders, "csr_from_adj_lists", fake_csr)
    monkeypatch.setattr(encoders, "_CSR_RECHECK_EVERY", 1)       # full fingerprint on every call for this part
    encoders.invalidate_adjacency()
    adj = defaultdict(set, {0: {1}, 1: {0, 2}, 2: {1}})
    encoders._device_csr(adj, 3, "cpu")
    encoders._device_csr(adj, 3, "cpu")
    assert calls == [4]                          # second call is a cache hit
    adj[7]                                       # the reference's defaultdict inserts an empty set on a miss: same graph
    encoders._device_csr(adj, 3, "cpu")
    assert calls == [4]
    adj[0].add(2)
    adj[2].add(0)                                # edited in place: must be rebuilt
    encoders._device_csr(adj, 3, "cpu")
    assert calls == [4, 6]
    encoders.invalidate_adjacency(adj)
    encoders._device_csr(adj, 3, "cpu")
    assert calls == [4, 6, 6]
    # the O(N) fingerprint runs only every _CSR_RECHECK_EVERY-th call (a new key triggers it at once); an edit that
    # keeps the key count is seen at the next periodic check at the latest, and a MOVED edge (same counts) is seen too
    monkeypatch.setattr(encoders, "_CSR_RECHECK_EVERY", 8)
    encoders.invalidate_adjacency()
    del calls[:]
    adj = defaultdict(set, {0: {1}, 1: {0, 2}, 2: {1}, 3: set()})
    encoders._device_csr(adj, 4, "cpu")
    adj[0].discard(1); adj[1].discard(0); adj[0].add(3); adj[3].add(0)      # moved: same non-empty count? no -- 3 becomes non-empty
    adj[3].discard(0); adj[0].discard(3); adj[0].add(2); adj[2].add(0)      # now 0-2 instead of 0-1: same counts, same degree total
    for _ in range(6):
        encoders._device_csr(adj, 4, "cpu")
    assert calls == [4]                          # not yet re-fingerprinted
    for _ in range(3):
        encoders._device_csr(adj, 4, "cpu")
    assert calls == [4, 4]                       # the periodic check saw the moved edge
    adj[9].add(0)                                # a new key: checked at once
    encoders._device_csr(adj, 4, "cpu")
    assert calls == [4, 4, 5]
    keep = [defaultdict(set, {0: {i + 1}}) for i in range(encoders._CSR_CACHE_MAX + 3)]
    for a in keep:
        encoders._device_csr(a, 0, "cpu")
    assert len(encoders._csr_cache) == encoders._CSR_CACHE_MAX


def test_adjacency_cache_probes_the_batch_own_rows_every_call(monkeypatch):
    """between two periodic fingerprints an in-place edit of an existing node's set was served from the stale CSR.  Every call
    now compares the set sizes of (up to 64 of) the batch's own nodes with the cached degrees; SAGE_ADJ_STRICT=1 fingerprints every call."""
    from collections import defaultdict
    from sage355 import encoders
    from sage355.graph import csr_from_adj_lists
    builds = []
    real = csr_from_adj_lists

    def counting_csr(adj, n):
        builds.append(sum(len(s) for s in adj.values()))
        return real(adj, n)

    monkeypatch.setattr(encoders, "csr_from_adj_lists", counting_csr)
    monkeypatch.setattr(encoders, "_CSR_RECHECK_EVERY", 1000)
    monkeypatch.delenv("SAGE_ADJ_STRICT", raising=False)
    encoders.invalidate_adjacency()
    adj = defaultdict(set, {0: {1, 2}, 1: {0}, 2: {0}, 3: set()})
    encoders._device_csr(adj, 4, "cpu", [0, 1])
    encoders._device_csr(adj, 4, "cpu", [0, 1])
    assert builds == [4]
    adj[1].add(2); adj[2].add(1)                       # node 1 grows: a batch that contains it sees the change at once ...
    encoders._device_csr(adj, 4, "cpu", [0, 3])        # ... one that does not is still served from the cache (the documented freeze)
    assert builds == [4]
    rp, col = encoders._device_csr(adj, 4, "cpu", [1])
    assert builds == [4, 6] and rp.tolist() == [0, 2, 4, 6, 6]
    n_keys = len(adj)
    encoders._device_csr(adj, 4, "cpu", [3, 2, 1, 0])
    assert builds == [4, 6] and len(adj) == n_keys     # the probe never inserts keys into a defaultdict
    adj[0].discard(2); adj[0].add(3); adj[2].discard(0); adj[3].add(0)     # degrees of 0 kept, 2 shrinks, 3 grows; batch = [0] only: sizes equal
    encoders._device_csr(adj, 4, "cpu", [0])
    assert builds == [4, 6]
    monkeypatch.setenv("SAGE_ADJ_STRICT", "1")         # strict mode: the full fingerprint on every call
    encoders._device_csr(adj, 4, "cpu", [0])
    assert builds == [4, 6, 6]
    encoders.invalidate_adjacency()


def test_strict_drop_in_mode_caps_torch_threads_to_the_usable_cores(monkeypatch):
    """with cuda=False the caller's classifier / loss / SGD run in torch on the host; on a box whose cgroup share is smaller
    than its core count the intra-op pool is capped once (SAGE_KEEP_TORCH_THREADS=1 opts out)."""
    import torch
    from sage355 import encoders
    from util import usable_cores
    before = torch.get_num_threads()
    try:
        monkeypatch.setattr(encoders, "_threads_capped", False)
        monkeypatch.setenv("SAGE_KEEP_TORCH_THREADS", "1")
        torch.set_num_threads(usable_cores() + 3)
        encoders._cap_host_threads_once()
        assert torch.get_num_threads() == usable_cores() + 3          # opted out
        monkeypatch.setattr(encoders, "_threads_capped", False)
        monkeypatch.delenv("SAGE_KEEP_TORCH_THREADS")
        encoders._cap_host_threads_once()
        assert torch.get_num_threads() == usable_cores()
        torch.set_num_threads(usable_cores() + 3)
        encoders._cap_host_threads_once()                              # once per process: a caller's later choice is respected
        assert torch.get_num_threads() == usable_cores() + 3
    finally:
        torch.set_num_threads(before)


def test_rmat_accelerated_integer_path_gives_the_same_csr():
    """rmat_graph(accel=...) runs the integer work on torch; the uniforms still come from numpy's stream, so the CSR is identical."""
    a = G.rmat_graph(13, 150_000, seed=3, accel=None, chunk=1 << 16)
    b = G.rmat_graph(13, 150_000, seed=3, accel="cpu", chunk=1 << 16)
    assert np.array_equal(a.rowptr, b.rowptr) and np.array_equal(a.col, b.col) and a.col.dtype == b.col.dtype == np.int32


def test_classifier_gradient_is_summed_over_the_batch_in_a_fixed_order():
    """train._sum_over_batch: g_scores^T . emb cut into 256-row pieces (one batched product) added in index order, a ragged tail
    last -- the same value as the plain product up to fp32 rounding, for every batch size, and the plain product itself for
    batches of at most 256 rows (model.py:247-250: the classifier's weight gradient of loss.backward())."""
    import torch
    from sage355.train import _sum_over_batch
    gen = torch.Generator().manual_seed(0)
    for b, c, h in [(4, 3, 8), (256, 7, 128), (257, 7, 128), (1024, 5, 32), (1100, 5, 32), (4096, 7, 128)]:
        g, e = torch.randn(b, c, generator=gen), torch.randn(b, h, generator=gen)
        ref = g.double().t() @ e.double()
        got = _sum_over_batch(g, e)
        assert got.shape == (c, h)
        assert (got.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
        if b <= 511:
            assert torch.equal(got, g.t() @ e)
        # piece by piece, by hand
        if b % 256 == 0 and b > 256:
            want = sum((g[i:i + 256].t() @ e[i:i + 256]).double() for i in range(0, b, 256))
            assert (got.double() - want).abs().max().item() <= 1e-5 * ref.abs().max().item()


def test_import_leaves_the_environment_alone_when_asked(tmp_path):
    """`import sage355` defaults GPU_MAX_HW_QUEUES=8 (role streams need a hardware queue each) unless the user chose a value or set
    SAGE355_KEEP_ENV=1 (a side effect a host application may not expect)."""
    import os, subprocess, sys
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "graphsage-simple_amd")
    code = "import os, sys; sys.path.insert(0, %r); import sage355; print(os.environ.get('GPU_MAX_HW_QUEUES'))" % pkg
    base = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "SAGE355_KEEP_ENV")}
    run = lambda env: subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120).stdout.strip()
    assert run(base) == "8"
    assert run(dict(base, GPU_MAX_HW_QUEUES="5")) == "5"
    assert run(dict(base, SAGE355_KEEP_ENV="1")) == "None"
