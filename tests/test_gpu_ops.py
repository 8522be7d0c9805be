"""GPU parity of each C-ABI operator against the CPU oracle (run with -m gpu on an MI355X).

Integer / index work is compared bit-exact; floating point within 1e-5 of the row maximum
(util.RTOL, BASELINE.json north_star + SURVEY.md section 7).
"""
import numpy as np
import pytest
import torch

from oracle import ref_sparse, sampler_ref
from sage355 import native, ops
from sage355.graph import rmat_graph
from util import assert_close_rowmax

pytestmark = pytest.mark.gpu
DEV = "cuda"


def small_rmat(scale=13, edges=200_000, seed=3):
    return rmat_graph(scale, edges, seed=seed)


@pytest.fixture(scope="module")
def graph():
    g = small_rmat()
    return g, torch.from_numpy(g.rowptr).to(DEV), torch.from_numpy(g.col).to(DEV)


# ------------------------------------------------------------------ sampler (a1, a2)
@pytest.mark.parametrize("k", [1, 5, 15, 25, 64])
def test_sampler_bit_exact_vs_c_oracle(graph, k):
    g, rowptr, col = graph
    rs = np.random.default_rng(k)
    nodes = rs.integers(0, g.num_nodes, size=5000).astype(np.int32)
    nodes[:50] = np.argsort(-g.degrees())[:50]          # the hubs
    nbr, cnt, _, _ = ops.sample_neighbors(rowptr, col, torch.from_numpy(nodes).to(DEV), k, seed=0xC0FFEE1234, tag=ops.TAG_INNER)
    ref_nbr, ref_cnt = sampler_ref.sample_neighbors(g.rowptr, g.col, nodes, k, 0xC0FFEE1234, ops.TAG_INNER)
    assert np.array_equal(cnt.cpu().numpy(), ref_cnt)
    assert np.array_equal(nbr.cpu().numpy(), ref_nbr)


def test_sampler_contract_distinct_members_takeall(graph):
    """aggregators.py:42-46: deg >= k -> k distinct neighbours; deg < k -> all of them."""
    g, rowptr, col = graph
    k = 10
    nodes = np.arange(g.num_nodes, dtype=np.int32)
    nbr, cnt, _, _ = ops.sample_neighbors(rowptr, col, torch.from_numpy(nodes).to(DEV), k, seed=7, tag=ops.TAG_OUTER)
    nbr, cnt = nbr.cpu().numpy(), cnt.cpu().numpy()
    deg = g.degrees()
    assert np.array_equal(cnt, np.minimum(deg, k))
    for v in np.concatenate([np.argsort(-deg)[:200], np.random.default_rng(0).integers(0, g.num_nodes, 2000)]):
        row = nbr[v, :cnt[v]]
        assert len(set(row.tolist())) == cnt[v]
        assert set(row.tolist()) <= set(g.neighbors(v).tolist())
        if deg[v] <= k:
            assert np.array_equal(row, g.neighbors(v))
        assert (nbr[v, cnt[v]:] == -1).all()


def test_sampler_is_uniform_without_replacement():
    """60 000 nodes that all have the same 40 neighbours, k=8: every neighbour must be chosen
    with probability k/deg and every ordered pair with k(k-1)/(deg(deg-1)) (6-sigma bands);
    streams keyed by different node ids must be independent enough for that to hold."""
    deg, k, trials = 40, 8, 60000
    rowptr = torch.arange(trials + deg + 1, dtype=torch.int64).clamp(max=trials) * deg
    col = (trials + torch.arange(deg, dtype=torch.int32)).repeat(trials)
    nodes = torch.arange(trials, dtype=torch.int32)
    nbr, cnt, _, _ = ops.sample_neighbors(rowptr.to(DEV), col.to(DEV), nodes.to(DEV), k, seed=2718281828, tag=5)
    rows = nbr.cpu().numpy() - trials
    assert (cnt.cpu().numpy() == k).all() and rows.min() >= 0 and rows.max() < deg
    srt = np.sort(rows, 1)
    assert (srt[:, 1:] != srt[:, :-1]).all()
    hits = np.bincount(rows.ravel(), minlength=deg)
    p = k / deg
    assert np.abs(hits - trials * p).max() < 6 * np.sqrt(trials * p * (1 - p)), hits
    pair = np.zeros((deg, deg))
    for a in range(k):
        for b in range(k):
            if a != b:
                np.add.at(pair, (rows[:, a], rows[:, b]), 1)
    pp = k * (k - 1) / (deg * (deg - 1))
    off = pair[~np.eye(deg, dtype=bool)]
    assert np.abs(off - trials * pp).max() < 6 * np.sqrt(trials * pp * (1 - pp))
    # a different seed gives different sets for (almost) every node
    nbr2, _, _, _ = ops.sample_neighbors(rowptr.to(DEV), col.to(DEV), nodes.to(DEV), k, seed=31415, tag=5)
    assert (np.sort(nbr2.cpu().numpy() - trials, 1) != srt).any(1).mean() > 0.99


# ------------------------------------------------------------------ frontier (a4)
@pytest.mark.parametrize("insert_self", [False, True])
def test_frontier_is_the_set_union_with_a_consistent_row_map(graph, insert_self):
    g, rowptr, col = graph
    k = 25
    rs = np.random.default_rng(11)
    deg = g.degrees()
    nodes = rs.choice(np.nonzero(deg > 0)[0], 4096, replace=False).astype(np.int32)
    nodes_d = torch.from_numpy(nodes).to(DEV)
    first_row = 17
    fr = ops.Frontier(nodes.size * (k + 1), DEV, first_row=first_row)
    nbr, cnt, slot, self_slot = ops.sample_neighbors(rowptr, col, nodes_d, k, seed=99, tag=ops.TAG_OUTER, frontier=fr,
                                                     insert_self=insert_self)
    torch.cuda.synchronize()
    total = fr.size()
    ids = fr.nodes[first_row:total].cpu().numpy()
    nbr_h, cnt_h, slot_h = nbr.cpu().numpy(), cnt.cpu().numpy(), slot.cpu().numpy()
    valid = np.arange(k)[None, :] < cnt_h[:, None]
    expect = set(nbr_h[valid].tolist()) | (set(nodes.tolist()) if insert_self else set())
    assert len(ids) == len(set(ids.tolist())), "frontier holds a duplicate"
    assert set(ids.tolist()) == expect                                    # aggregators.py:52
    rows = fr.rows.cpu().numpy()
    keys = fr.keys.cpu().numpy()
    listed = fr.nodes.cpu().numpy()
    assert (slot_h[~valid] == -1).all()
    assert np.array_equal(keys[slot_h[valid]], nbr_h[valid])
    assert np.array_equal(listed[rows[slot_h[valid]]], nbr_h[valid])        # aggregators.py:53,55
    assert rows[slot_h[valid]].min() >= first_row
    if insert_self:
        ss = self_slot.cpu().numpy()
        assert np.array_equal(listed[rows[ss]], nodes)
    # frontier_insert on the same lists reproduces the same set in a fresh table
    fr2 = ops.Frontier(nodes.size * (k + 1), DEV)
    slot2, _ = ops.frontier_insert(nbr, cnt, fr2, self_nodes=nodes_d if insert_self else None)
    torch.cuda.synchronize()
    assert set(fr2.node_list().cpu().numpy().tolist()) == expect
    fr.reset(0)
    torch.cuda.synchronize()
    assert fr.size() == 0 and bool((fr.keys == -1).all())


# ------------------------------------------------------------------ gather_mean (a5-a8)
@pytest.mark.parametrize("dim,k,n", [(256, 15, 3000), (128, 25, 1000), (1433, 5, 300), (50, 25, 700), (64, 100, 200), (4, 1, 65),
                                     (256, 15, 9000), (128, 25, 8200), (192, 40, 8192), (256, 64, 8300), (256, 70, 8300)])  # >= 8192 rows: column-sliced kernel
def test_gather_mean_matches_oracle(dim, k, n):
    rs = np.random.default_rng(dim + k)
    rows = 5000
    table = torch.randn(rows, dim, generator=torch.Generator().manual_seed(dim))
    cnt = rs.integers(1, k + 1, size=n).astype(np.int32)
    cnt[:3] = k
    nbr = rs.integers(0, rows, size=(n, k)).astype(np.int32)
    out = ops.gather_mean(table.to(DEV), torch.from_numpy(nbr).to(DEV), torch.from_numpy(cnt).to(DEV))
    ref = ref_sparse.gather_mean(table, nbr, cnt)
    assert_close_rowmax(out.cpu(), ref, what=f"gather_mean dim={dim} k={k}")


@pytest.mark.parametrize("rows,dim,k,n", [(900, 96, 9, 500), (3000, 256, 15, 8500)])   # row-per-wave and column-sliced kernels
def test_gather_mean_strided_table_slot_rows_and_self_row(rows, dim, k, n):
    rs = np.random.default_rng(5)
    big = torch.randn(rows, dim + 32, generator=torch.Generator().manual_seed(1)).to(DEV)
    table = big[:, :dim]                                   # ld = dim + 32
    perm = rs.permutation(4096).astype(np.int32)           # slot -> row indirection
    slot_rows = np.full(4096, -1, dtype=np.int32)
    slot_rows[perm[:rows]] = np.arange(rows, dtype=np.int32)
    cnt = rs.integers(0, k + 1, size=n).astype(np.int32)
    nbr_row = rs.integers(0, rows, size=(n, k)).astype(np.int32)
    self_rowid = rs.integers(0, rows, size=n).astype(np.int32)
    self_rowid[:50] = nbr_row[:50, 0]                      # self already sampled (when cnt > 0)
    nbr_slot, self_slot = perm[nbr_row], perm[self_rowid]
    out = ops.gather_mean(table, torch.from_numpy(nbr_slot).to(DEV), torch.from_numpy(cnt).to(DEV),
                          slot_rows=torch.from_numpy(slot_rows).to(DEV), self_row=torch.from_numpy(self_slot).to(DEV))
    ref = ref_sparse.gather_mean(table.cpu(), nbr_row, cnt, self_idx=self_rowid)
    assert_close_rowmax(out.cpu(), ref, what="slot_rows + self_row")


def test_gather_mean_empty_set_rule_follows_reference():
    """aggregators.py:60-61 via tests/golden/empty_sets.npz (reference outputs)."""
    from util import load_golden
    g = load_golden("empty_sets")
    table = torch.from_numpy(g["table"]).to(DEV)
    nbr = torch.from_numpy(g["nbr"].astype(np.int32)).to(DEV)
    cnt = torch.from_numpy(g["cnt"].astype(np.int32)).to(DEV)
    one = torch.ones(1, dtype=torch.int32, device=DEV)
    zero = torch.zeros(1, dtype=torch.int32, device=DEV)
    mixed = ops.gather_mean(table, nbr, cnt, any_nonempty=one)
    assert_close_rowmax(mixed.cpu(), g["agg_mixed"], what="mixed batch: NaN rows")
    allempty = ops.gather_mean(table, nbr[[1, 3]].contiguous(), cnt[[1, 3]].contiguous(), any_nonempty=zero)
    assert_close_rowmax(allempty.cpu(), g["agg_all_empty"], what="all-empty batch: zeros")
    enc = ops.linear_act(mixed, torch.from_numpy(g["w"]).to(DEV))
    assert_close_rowmax(enc.cpu().t(), g["enc_mixed"], rows_dim=1, what="relu keeps NaN (torch.relu semantics)")


# ------------------------------------------------------------------ linear_act (a10, a11)
@pytest.mark.parametrize("n,dim,h,concat,act", [
    (1000, 256, 128, False, "relu"), (1000, 256, 128, True, "relu"), (77, 1433, 50, False, "relu"),
    (300, 500, 50, True, "sigmoid"), (129, 50, 128, False, "relu"), (65, 7, 3, True, "none"), (4096, 128, 300, False, "relu")])
def test_linear_act_matches_oracle(n, dim, h, concat, act):
    gen = torch.Generator().manual_seed(n + dim)
    agg = torch.randn(n, dim, generator=gen)
    w = torch.randn(h, dim * (2 if concat else 1), generator=gen) / np.sqrt(dim)
    code = {"relu": ops.ACT_RELU, "sigmoid": ops.ACT_SIGMOID, "none": ops.ACT_NONE}[act]
    self_tab = self_index = self_feats = None
    if concat:
        self_tab = torch.randn(2 * n, dim, generator=gen)
        self_index = torch.randperm(2 * n, generator=gen)[:n].to(torch.int32)
        self_feats = self_tab[self_index.long()]
    out = ops.linear_act(agg.to(DEV), w.to(DEV), act=code, self_tab=None if self_tab is None else self_tab.to(DEV),
                         self_index=None if self_index is None else self_index.to(DEV))
    ref = ref_sparse.linear_act(self_feats, agg, w, act)
    # pre-activation scale is what the 1e-5 is relative to (post-relu rows can be all ~0)
    pre = ref_sparse.linear_act(self_feats, agg, w, "none").abs().amax(1, keepdim=True)
    err = ((out.cpu().double() - ref).abs() / pre.clamp_min(1e-30)).max().item()
    assert err <= 1e-5, err


def test_bad_arguments_raise_and_do_not_launch():
    t = torch.zeros(4, 4, device=DEV)
    with pytest.raises(native.SageError):
        ops.gather_mean(t, torch.zeros(2, 2, dtype=torch.int64, device=DEV), torch.zeros(2, dtype=torch.int32, device=DEV))
    with pytest.raises(native.SageError):
        ops.linear_act(t, torch.zeros(3, 5, device=DEV))
    with pytest.raises(native.SageError):
        ops.sample_neighbors(torch.zeros(3, dtype=torch.int64, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV),
                             torch.zeros(2, dtype=torch.int32, device=DEV), 65, seed=0)
