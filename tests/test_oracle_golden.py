"""Pin the CPU oracle against the reference's own outputs (tests/golden/*.npz,
written by tests/golden/make_golden.py from the imported reference).

CPU only.  These tests are what makes every later "HIP == oracle" statement a
"HIP == reference" statement.
"""
import random

import numpy as np
import pytest
import torch

from oracle import ref_dense, ref_sparse
from sage355.graph import CSRGraph
from util import TWO_LAYER_CASES, assert_agg1_close, assert_close_rowmax, full_table, load_golden, sets_from_padded


def _acts(g):
    return ("sigmoid" if int(g["sigmoid1"]) else "relu", "sigmoid" if int(g["sigmoid2"]) else "relu")


def _inits(g):
    return ("shared" if int(g["sigmoid1"]) else "None", "shared" if int(g["sigmoid2"]) else "None")


@pytest.mark.parametrize("name", TWO_LAYER_CASES)
def test_dense_restatement_matches_reference(name):
    g = load_golden(name)
    table = full_table(g)
    gcn = bool(g["gcn"])
    sets1 = sets_from_padded(g["layer1_nodes"], g["nbr1"], g["cnt1"])
    sets2 = sets_from_padded(g["seeds"], g["nbr2"], g["cnt2"])
    w1, w2 = torch.from_numpy(g["w1"]), torch.from_numpy(g["w2"])
    i1, i2 = _inits(g)
    l1 = [int(x) for x in g["layer1_nodes"]]

    agg1, _ = ref_dense.mean_aggregate(l1, [sets1[u] for u in l1], lambda ids: table[ids])
    assert_agg1_close(agg1, g, what="agg1")
    enc1 = ref_dense.encoder_forward(l1, sets1, lambda ids: table[ids], w1, None, gcn, initializer=i1)
    assert_close_rowmax(enc1, g["enc1_out"], rows_dim=1, what="enc1")
    out = ref_dense.two_hop_forward([int(s) for s in g["seeds"]], sets1, sets2, table, w1, w2, None, None, gcn,
                                    initializer1=i1, initializer2=i2)
    assert_close_rowmax(out, g["enc2_out"], rows_dim=1, what="enc2")


@pytest.mark.parametrize("name", TWO_LAYER_CASES)
def test_sparse_restatement_matches_reference(name):
    g = load_golden(name)
    table = full_table(g)
    gcn = bool(g["gcn"])
    a1, a2 = _acts(g)
    l1 = g["layer1_nodes"]
    agg1 = ref_sparse.gather_mean(table, g["nbr1"], g["cnt1"])
    assert_agg1_close(agg1, g, what="agg1")
    pos = {int(v): i for i, v in enumerate(l1)}
    seed_rows = [pos[int(s)] for s in g["seeds"]] if not gcn else None
    out = ref_sparse.two_hop_forward(
        table, g["w1"], g["w2"], g["seeds"], g["nbr2"], g["cnt2"], l1, g["nbr1"], g["cnt1"], gcn,
        act1=a1, act2=a2,
        seed_nbr1=None if gcn else g["nbr1"][seed_rows], seed_cnt1=None if gcn else g["cnt1"][seed_rows])
    assert_close_rowmax(out.t(), g["enc2_out"], rows_dim=1, what="enc2")


def test_empty_neighbour_sets_follow_reference():
    """aggregators.py:60-61: NaN row inside a mixed batch, zeros for an all-empty batch."""
    g = load_golden("empty_sets")
    table = torch.from_numpy(g["table"])
    sets = [{1, 2}, set(), {0}, set()]
    mixed, _ = ref_dense.mean_aggregate([0, 3, 4, 5], sets, lambda ids: table[ids])
    assert_close_rowmax(mixed, g["agg_mixed"], what="mixed")
    allempty, _ = ref_dense.mean_aggregate([3, 5], [set(), set()], lambda ids: table[ids])
    assert_close_rowmax(allempty, g["agg_all_empty"], what="all-empty")
    sp = ref_sparse.gather_mean(table, g["nbr"], g["cnt"])
    assert_close_rowmax(sp, g["agg_mixed"], what="sparse mixed")
    sp0 = ref_sparse.gather_mean(table, g["nbr"][[1, 3]], g["cnt"][[1, 3]])
    assert_close_rowmax(sp0, g["agg_all_empty"], what="sparse all-empty")
    enc = ref_sparse.linear_act(None, sp, g["w"])
    assert_close_rowmax(enc.t(), g["enc_mixed"], rows_dim=1, what="enc mixed (NaN propagates through relu)")


def test_sampler_consumes_python_random_like_reference():
    """Reference run with sampling ON after random.seed(s): the restatement must
    draw the same sets from the same stream (aggregators.py:42-46)."""
    g = load_golden("sampler_stream")
    graph = CSRGraph(g["rowptr"], g["col"], len(g["rowptr"]) - 1)
    adj = graph.to_adj_lists()
    table = torch.from_numpy(g["table"])
    random.seed(int(g["py_seed"]))
    out = ref_dense.two_hop_forward([int(s) for s in g["seeds"]], adj, adj, table, torch.from_numpy(g["w1"]),
                                    torch.from_numpy(g["w2"]), int(g["k1"]), int(g["k2"]), True)
    assert_close_rowmax(out, g["enc2_out"], rows_dim=1, what="sampled forward")


def test_sample_sets_contract():
    rng = random.Random(0)
    sets = [set(range(20)), set(range(3)), set(), {7}]
    out = ref_dense.sample_sets(sets, 5, rng)
    assert len(out[0]) == 5 and out[0] <= sets[0]
    assert out[1] == sets[1] and out[2] == set() and out[3] == {7}
    assert ref_dense.sample_sets(sets, None) == sets
    exact = ref_dense.sample_sets([set(range(5))], 5, rng)
    assert exact[0] == set(range(5))


def test_gcn_aggregator_variant_intended_semantics():
    """aggregators.py:50-51 raises TypeError in the reference (set + set); the build
    implements the intended self-loop union.  Parity unpinned by the reference: dense
    and sparse restatements must agree with each other."""
    gen = torch.Generator().manual_seed(0)
    table = torch.randn(8, 5, generator=gen)
    nodes = [0, 1, 2, 3]
    sets = [{1, 2}, {1, 4}, set(), {5, 6, 7}]
    dense, _ = ref_dense.mean_aggregate(nodes, sets, lambda ids: table[ids], gcn=True)
    nbr = np.array([[1, 2, -1], [1, 4, -1], [-1, -1, -1], [5, 6, 7]])
    cnt = np.array([2, 2, 0, 3])
    sparse = ref_sparse.gather_mean(table, nbr, cnt, self_idx=np.array(nodes))
    assert_close_rowmax(sparse, dense, what="gcn aggregator")
    assert torch.allclose(dense[1], table[[1, 4]].mean(0), atol=1e-6)  # self already sampled: not counted twice
    assert torch.allclose(dense[2], table[2], atol=1e-6)               # isolated node: its own row
