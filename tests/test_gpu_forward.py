"""GPU parity of the 2-hop forward and of the drop-in nn.Modules.

* against the reference's own outputs (tests/golden/*.npz) on identical, injected
  neighbour sets -- the fixtures' sets become the rows of a CSR and k >= the longest
  row, which is the reference's num_sample=None switch (aggregators.py:47-48);
* against the fp64 oracle on the device sampler's own sets, copied back, at sizes up
  to BASELINE config 3 (B = 4096, fanout 15/25, D0 = 256, H = 128/128).
"""
import random

import numpy as np
import pytest
import torch

from oracle import ref_sparse, sampler_ref
from sage355 import ops
from sage355.aggregators import MeanAggregator
from sage355.encoders import Encoder
from sage355.engine import TwoHopEngine
from sage355.graph import CSRGraph, rmat_graph
from util import TWO_LAYER_CASES, assert_agg1_close, assert_close_rowmax, full_table, load_golden, sets_from_padded

pytestmark = pytest.mark.gpu
DEV = "cuda"


def csr_of_sets(nodes, nbr, cnt, num_nodes):
    rowptr = np.zeros(num_nodes + 1, dtype=np.int64)
    deg = np.zeros(num_nodes, dtype=np.int64)
    deg[nodes] = cnt
    np.cumsum(deg, out=rowptr[1:])
    col = np.zeros(max(int(rowptr[-1]), 1), dtype=np.int32)
    for r, v in enumerate(nodes):
        col[rowptr[v]:rowptr[v] + cnt[r]] = nbr[r, :cnt[r]]
    return torch.from_numpy(rowptr).to(DEV), torch.from_numpy(col).to(DEV)


def engine_from_golden(g, fused):
    n = int(g["num_nodes"])
    gcn = bool(g["gcn"])
    rp1, c1 = csr_of_sets(g["layer1_nodes"], g["nbr1"], g["cnt1"], n)
    rp2, c2 = csr_of_sets(g["seeds"], g["nbr2"], g["cnt2"], n)
    act1 = ops.ACT_SIGMOID if int(g["sigmoid1"]) else ops.ACT_RELU
    act2 = ops.ACT_SIGMOID if int(g["sigmoid2"]) else ops.ACT_RELU
    return TwoHopEngine(rp1, c1, full_table(g).to(DEV), torch.from_numpy(g["w1"]).to(DEV), torch.from_numpy(g["w2"]).to(DEV),
                        k1=max(int(g["cnt1"].max()), 1), k2=max(int(g["cnt2"].max()), 1), concat=not gcn, act1=act1, act2=act2,
                        fused=fused, max_batch=len(g["seeds"]), rowptr_outer=rp2, col_outer=c2)


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("name", TWO_LAYER_CASES)
def test_two_hop_forward_matches_reference_golden(name, fused):
    g = load_golden(name)
    eng = engine_from_golden(g, fused)
    out = eng.forward(torch.from_numpy(g["seeds"].astype(np.int32)).to(DEV), seed=1)
    assert_close_rowmax(out.cpu().t(), g["enc2_out"], rows_dim=1, what=f"{name} enc2_out")
    inter = eng.intermediates()
    # layer-1 embeddings of the frontier rows == the reference's enc1 output for those ids
    pos = {int(v): i for i, v in enumerate(g["layer1_nodes"])}
    s1 = inter["s1_nodes"].cpu().numpy()
    ref_h1 = torch.from_numpy(g["enc1_out"]).t()[[pos[int(v)] for v in s1]]
    assert_close_rowmax(inter["h1"].cpu(), ref_h1, what=f"{name} enc1_out on the frontier")
    first = inter["first_frontier_row"]
    assert len(set(s1[first:].tolist())) == len(s1) - first
    valid = np.arange(g["nbr2"].shape[1])[None, :] < g["cnt2"][:, None]
    assert set(s1[first:].tolist()) == set(g["nbr2"][valid].tolist())


def build_modules(g, num_sample1, num_sample2, cuda, adj1, adj2):
    """model.py:214-222 wiring with this package's classes."""
    table = full_table(g)
    gcn = bool(g["gcn"])
    features = torch.nn.Embedding(*table.shape)
    features.weight = torch.nn.Parameter(table, requires_grad=False)
    i1 = "shared" if int(g["sigmoid1"]) else "None"
    i2 = "shared" if int(g["sigmoid2"]) else "None"
    agg1 = MeanAggregator(features, cuda=cuda)
    enc1 = Encoder(features, table.shape[1], g["w1"].shape[0], adj1, agg1, num_sample=num_sample1, gcn=gcn, cuda=cuda, initializer=i1)
    agg2 = MeanAggregator(lambda nodes: enc1(nodes).t(), cuda=cuda)
    enc2 = Encoder(lambda nodes: enc1(nodes).t(), enc1.embed_dim, g["w2"].shape[0], adj2, agg2, num_sample=num_sample2,
                   base_model=enc1, gcn=gcn, cuda=cuda, initializer=i2)
    with torch.no_grad():
        enc1.weight.copy_(torch.from_numpy(g["w1"]))
        enc2.weight.copy_(torch.from_numpy(g["w2"]))
    return enc1, enc2


@pytest.mark.parametrize("name", TWO_LAYER_CASES)
def test_modules_strict_path_num_sample_none(name):
    """Encoder / MeanAggregator used exactly as the reference's own golden run used its classes:
    pre-sampled dicts, num_sample=None, cuda=False -> CPU tensors [embed_dim, B]."""
    g = load_golden(name)
    sets1 = sets_from_padded(g["layer1_nodes"], g["nbr1"], g["cnt1"])
    sets2 = sets_from_padded(g["seeds"], g["nbr2"], g["cnt2"])
    enc1, enc2 = build_modules(g, None, None, False, sets1, sets2)
    seeds = [int(s) for s in g["seeds"]]
    with torch.no_grad():
        l1 = [int(x) for x in g["layer1_nodes"]]
        agg1 = enc1.aggregator.forward(l1, [sets1[u] for u in l1], None)
        out1 = enc1(torch.LongTensor(l1))
        out2 = enc2(seeds)
    assert not out2.is_cuda and out2.shape == g["enc2_out"].shape
    assert_agg1_close(agg1, g, what="agg1_out")
    assert_close_rowmax(out1, g["enc1_out"], rows_dim=1, what="enc1_out")
    assert_close_rowmax(out2, g["enc2_out"], rows_dim=1, what="enc2_out")


@pytest.mark.parametrize("name", ["tiny_gcn", "cora_emb_gcn_5_5", "cora_emb_concat_10_10", "pubmed_gcn_10_25"])
def test_modules_fast_path_takes_whole_sets(name):
    """Same fixtures through the device-sampler path: with num_sample >= every set size the sampler's
    take-all branch reproduces the injected sets, so the fused 2-hop engine must hit the golden too."""
    g = load_golden(name)
    sets1 = sets_from_padded(g["layer1_nodes"], g["nbr1"], g["cnt1"])
    sets2 = sets_from_padded(g["seeds"], g["nbr2"], g["cnt2"])
    enc1, enc2 = build_modules(g, int(g["k1"]), int(g["k2"]), True, sets1, sets2)
    assert enc2._can_fuse_two_hop()
    with torch.no_grad():
        out2 = enc2(g["seeds"])                     # numpy ids, as model.py:256 passes `val`
    assert out2.is_cuda
    assert_close_rowmax(out2.cpu(), g["enc2_out"], rows_dim=1, what="enc2_out (fast path)")


def test_strict_path_consumes_python_random_like_the_reference():
    """sampler_stream.npz: the reference ran with ITS sampler on after random.seed(2024).  The
    generic module path samples with the same calls on the same stream, so it must land on the
    same output (aggregators.py:42-46)."""
    g = load_golden("sampler_stream")
    graph = CSRGraph(g["rowptr"], g["col"], len(g["rowptr"]) - 1)
    adj = graph.to_adj_lists()
    table = torch.from_numpy(g["table"])
    features = torch.nn.Embedding(*table.shape)
    features.weight = torch.nn.Parameter(table, requires_grad=False)
    agg1 = MeanAggregator(features)
    enc1 = Encoder(features, table.shape[1], 16, adj, agg1, num_sample=int(g["k1"]), gcn=True, fuse_base_model=False)
    enc1._is_table = lambda: False               # force the reference call sequence at layer 1 too
    agg2 = MeanAggregator(lambda nodes: enc1(nodes).t())
    enc2 = Encoder(lambda nodes: enc1(nodes).t(), 16, 8, adj, agg2, num_sample=int(g["k2"]), base_model=enc1, gcn=True,
                   fuse_base_model=False)
    with torch.no_grad():
        enc1.weight.copy_(torch.from_numpy(g["w1"]))
        enc2.weight.copy_(torch.from_numpy(g["w2"]))
        random.seed(int(g["py_seed"]))
        out = enc2([int(s) for s in g["seeds"]])
    assert_close_rowmax(out, g["enc2_out"], rows_dim=1, what="strict path with live sampling")


# ---------------------------------------------------------------- device-sampled sets vs the fp64 oracle
def check_engine_against_oracle(graph, table, w1, w2, seeds, k1, k2, concat, self_loop, fused, seed=1234):
    rowptr, col = torch.from_numpy(graph.rowptr).to(DEV), torch.from_numpy(graph.col).to(DEV)
    eng = TwoHopEngine(rowptr, col, table.to(DEV), w1.to(DEV), w2.to(DEV), k1, k2, concat=concat, agg_self_loop=self_loop,
                       fused=fused, max_batch=len(seeds))
    out = eng.forward(torch.from_numpy(seeds.astype(np.int32)).to(DEV), seed=seed).cpu()
    it = eng.intermediates()
    s1 = it["s1_nodes"].cpu().numpy()
    nbr2, cnt2 = it["nbr2"].cpu().numpy(), it["cnt2"].cpu().numpy()
    nbr1, cnt1 = it["nbr1"].cpu().numpy(), it["cnt1"].cpu().numpy()
    first = it["first_frontier_row"]
    # integer work, bit exact: the sampled lists are what the C restatement draws for these ids
    r2, c2 = sampler_ref.sample_neighbors(graph.rowptr, graph.col, seeds, k2, seed, ops.TAG_OUTER)
    assert np.array_equal(nbr2, r2) and np.array_equal(cnt2, c2)
    r1, c1 = sampler_ref.sample_neighbors(graph.rowptr, graph.col, s1[first:], k1, seed, ops.TAG_INNER)
    assert np.array_equal(nbr1[first:], r1) and np.array_equal(cnt1[first:], c1)
    if concat:
        assert np.array_equal(s1[:first], seeds)
        rs, cs = sampler_ref.sample_neighbors(graph.rowptr, graph.col, seeds, k1, seed, ops.TAG_INNER_SELF)
        assert np.array_equal(nbr1[:first], rs) and np.array_equal(cnt1[:first], cs)
    valid2 = np.arange(k2)[None, :] < cnt2[:, None]
    expect = set(nbr2[valid2].tolist()) | (set(seeds.tolist()) if self_loop else set())
    assert set(s1[first:].tolist()) == expect and len(s1) - first == len(expect)
    # floating point: fp64 oracle on the very same sets
    ref = ref_sparse.two_hop_forward(table, w1, w2, seeds, nbr2, cnt2, s1[first:], nbr1[first:], cnt1[first:], gcn=not concat,
                                     agg_gcn=self_loop, seed_nbr1=nbr1[:first] if concat else None,
                                     seed_cnt1=cnt1[:first] if concat else None)
    return assert_close_rowmax(out, ref, what=f"2-hop concat={concat} self_loop={self_loop} fused={fused}")


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("d0", [100, 128, 256])      # 128, 256: layer 1 takes the column-sliced gather + dense contraction
@pytest.mark.parametrize("concat,self_loop", [(False, False), (True, False), (False, True), (True, True)])
def test_two_hop_small_rmat_all_variants(concat, self_loop, fused, d0):
    graph = rmat_graph(14, 300_000, seed=2)
    gen = torch.Generator().manual_seed(0)
    h1, h2 = 64, 32
    table = torch.randn(graph.num_nodes, d0, generator=gen)
    m = 2 if concat else 1
    w1 = torch.randn(h1, m * d0, generator=gen) / np.sqrt(m * d0)
    w2 = torch.randn(h2, m * h1, generator=gen) / np.sqrt(m * h1)
    deg = graph.degrees()
    seeds = np.random.default_rng(1).choice(np.nonzero(deg > 0)[0], 777, replace=False)
    check_engine_against_oracle(graph, table, w1, w2, seeds, 10, 20, concat, self_loop, fused)


@pytest.mark.parametrize("concat", [False, True])
@pytest.mark.parametrize("d0,h1", [(320, 128), (500, 50), (1433, 50), (66, 30)])
def test_two_hop_wide_and_odd_widths(d0, h1, concat):
    """Rows wider than 256 floats (Pubmed's 500: multi-pass contraction) and widths that are not multiples of 4
    (Cora's 1433 raw features, the reference's 50-wide layer 1, model.py:543: the engine zero-pads them) at a size where
    layer 1 has ~15 k rows, against the fp64 oracle; the weights are changed in place afterwards to check that the
    padded copies follow."""
    graph = rmat_graph(14, 300_000, seed=2)
    gen = torch.Generator().manual_seed(6)
    m = 2 if concat else 1
    table = torch.randn(graph.num_nodes, d0, generator=gen)
    w1 = torch.randn(h1, m * d0, generator=gen) / np.sqrt(m * d0)
    w2 = torch.randn(40, m * h1, generator=gen) / np.sqrt(m * h1)
    seeds = np.random.default_rng(2).choice(np.nonzero(graph.degrees() > 0)[0], 777, replace=False)
    check_engine_against_oracle(graph, table, w1, w2, seeds, 10, 20, concat, False, True, seed=21)
    rowptr, col = torch.from_numpy(graph.rowptr).to(DEV), torch.from_numpy(graph.col).to(DEV)
    w1d, w2d = w1.to(DEV), w2.to(DEV)
    eng = TwoHopEngine(rowptr, col, table.to(DEV), w1d, w2d, 10, 20, concat=concat, max_batch=len(seeds))
    sd = torch.from_numpy(seeds.astype(np.int32)).to(DEV)
    before = eng.forward(sd, seed=21).clone()
    w1d.mul_(0.5)
    w2d.add_(0.01)
    after = eng.forward(sd, seed=21)
    fresh = TwoHopEngine(rowptr, col, table.to(DEV), w1d, w2d, 10, 20, concat=concat, max_batch=len(seeds)).forward(sd, seed=21)
    assert torch.equal(after, fresh) and not torch.equal(after, before)


@pytest.mark.parametrize("concat", [False, True])
@pytest.mark.parametrize("b,k1,k2", [(1, 15, 25), (17, 3, 40), (300, 64, 5), (64, 8, 64)])
def test_two_hop_odd_batches_and_fanouts(b, k1, k2, concat):
    """One seed, a batch that is no multiple of any tile, fanouts on both sides of every group width (3, 5, 8 | 40, 64:
    lane groups of 8, 16 and 64) -- sets bit-exact against the C sampler, values against the fp64 oracle."""
    graph = rmat_graph(13, 150_000, seed=4)
    gen = torch.Generator().manual_seed(8)
    m = 2 if concat else 1
    table = torch.randn(graph.num_nodes, 128, generator=gen)
    w1 = torch.randn(64, m * 128, generator=gen) / np.sqrt(m * 128)
    w2 = torch.randn(32, m * 64, generator=gen) / np.sqrt(m * 64)
    seeds = np.random.default_rng(b).choice(np.nonzero(graph.degrees() > 0)[0], b, replace=False)
    check_engine_against_oracle(graph, table, w1, w2, seeds, k1, k2, concat, False, True, seed=5)


@pytest.mark.parametrize("concat,self_loop", [(False, False), (True, False), (False, True)])
def test_two_hop_batch_with_repeated_seeds(concat, self_loop):
    """A node may appear more than once in a batch (encoders.py:47 builds one neighbour list per POSITION of `nodes`): every
    occurrence gets its own row of the output, the frontier still holds each node once, and the whole forward matches the
    oracle on the sets the device sampled."""
    graph = rmat_graph(13, 150_000, seed=4)
    gen = torch.Generator().manual_seed(9)
    m = 2 if concat else 1
    table = torch.randn(graph.num_nodes, 128, generator=gen)
    w1 = torch.randn(64, m * 128, generator=gen) / np.sqrt(m * 128)
    w2 = torch.randn(32, m * 64, generator=gen) / np.sqrt(m * 64)
    base = np.random.default_rng(3).choice(np.nonzero(graph.degrees() > 0)[0], 60, replace=False)
    seeds = np.concatenate([base, base[:25], base[:5], base[:5]])        # 95 positions, 60 distinct nodes
    np.random.default_rng(4).shuffle(seeds)
    check_engine_against_oracle(graph, table, w1, w2, seeds, 7, 9, concat, self_loop, True, seed=6)


@pytest.mark.parametrize("concat,self_loop", [(False, False), (True, True)])
def test_relabel_by_degree_is_the_same_engine_on_the_renumbered_graph(concat, self_loop):
    """TwoHopEngine(relabel="degree") = the plain engine on (graph, table, seeds) renumbered by descending degree on the
    host (sage355.graph.relabel_by_degree), bit for bit -- direct calls and graph replay -- and correct against the oracle
    on that renumbered graph."""
    from sage355.graph import relabel_by_degree
    graph = rmat_graph(14, 300_000, seed=2)
    gen = torch.Generator().manual_seed(12)
    m = 2 if concat else 1
    table = torch.randn(graph.num_nodes, 128, generator=gen)
    w1 = (torch.randn(64, m * 128, generator=gen) / 12).to(DEV)
    w2 = (torch.randn(32, m * 64, generator=gen) / 8).to(DEV)
    seeds = np.stack([np.random.default_rng(i).choice(graph.num_nodes, 700, replace=False) for i in range(3)])
    rowptr, col = graph.to(DEV)
    eng = TwoHopEngine(rowptr, col, table.to(DEV), w1, w2, 10, 20, concat=concat, agg_self_loop=self_loop, max_batch=700, relabel="degree")
    g2, new_of_old = relabel_by_degree(graph)
    assert np.array_equal(eng.node_order.cpu().numpy(), np.argsort(new_of_old))
    table2 = torch.empty_like(table)
    table2[torch.from_numpy(new_of_old)] = table
    rp2, c2 = g2.to(DEV)
    plain = TwoHopEngine(rp2, c2, table2.to(DEV), w1, w2, 10, 20, concat=concat, agg_self_loop=self_loop, max_batch=700)
    sd = torch.from_numpy(seeds.astype(np.int32)).to(DEV)
    sd2 = torch.from_numpy(new_of_old[seeds].astype(np.int32)).to(DEV)
    want = [plain.forward(sd2[i], seed=40 + i).clone() for i in range(3)]
    for i in range(3):
        assert torch.equal(torch.nan_to_num(eng.forward(sd[i], seed=40 + i), nan=-7.0), torch.nan_to_num(want[i], nan=-7.0))
    eng.set_queue(sd, [40, 41, 42])
    out = eng.capture()
    for i in range(3):
        eng.replay()
        torch.cuda.synchronize()
        assert torch.equal(torch.nan_to_num(out, nan=-7.0), torch.nan_to_num(want[i], nan=-7.0))
    check_engine_against_oracle(g2, table2, w1.cpu(), w2.cpu(), new_of_old[seeds[0]], 10, 20, concat, self_loop, True, seed=40)


@pytest.mark.parametrize("concat", [False, True])
def test_two_hop_all_isolated_batch_gives_zeros_not_nan(concat):
    """aggregators.py:60-61 when NO node of the call has a neighbour: the mask has zero columns, 0/0 never happens and
    the reference returns zeros (relu(W.0) = 0) -- also through the engine, where the frontier stays empty."""
    graph = rmat_graph(13, 150_000, seed=4)
    lonely = np.nonzero(graph.degrees() == 0)[0][:50]
    assert len(lonely) == 50
    gen = torch.Generator().manual_seed(9)
    m = 2 if concat else 1
    table = torch.randn(graph.num_nodes, 128, generator=gen)
    w1 = torch.randn(64, m * 128, generator=gen) / 16
    w2 = torch.randn(32, m * 64, generator=gen) / 8
    rowptr, col = torch.from_numpy(graph.rowptr).to(DEV), torch.from_numpy(graph.col).to(DEV)
    eng = TwoHopEngine(rowptr, col, table.to(DEV), w1.to(DEV), w2.to(DEV), 10, 25, concat=concat, max_batch=50)
    out = eng.forward(torch.from_numpy(lonely.astype(np.int32)).to(DEV), seed=3).cpu()
    assert not torch.isnan(out).any()
    it = eng.intermediates()
    assert not it["cnt2"].cpu().numpy().any() and it["n_s1"] == (50 if concat else 0)
    # every aggregate of the call is the all-empty batch's zeros: gcn -> relu(W.0) = 0; concat -> only the nodes' own rows count
    if concat:
        x = table[torch.from_numpy(lonely)].double()
        h1_self = torch.relu(x @ w1[:, :128].double().t())
        ref = torch.relu(h1_self @ w2[:, :64].double().t())
    else:
        ref = torch.zeros(50, 32, dtype=torch.float64)
    assert_close_rowmax(out, ref, what="all-isolated batch")
    # and the next, ordinary batch on the same engine is unaffected (the workspace cleaned itself)
    seeds = np.random.default_rng(1).choice(np.nonzero(graph.degrees() > 0)[0], 50, replace=False)
    a = eng.forward(torch.from_numpy(seeds.astype(np.int32)).to(DEV), seed=4).clone()
    fresh = TwoHopEngine(rowptr, col, table.to(DEV), w1.to(DEV), w2.to(DEV), 10, 25, concat=concat, max_batch=50)
    assert torch.equal(a, fresh.forward(torch.from_numpy(seeds.astype(np.int32)).to(DEV), seed=4))


@pytest.mark.parametrize("d0", [128, 256])
@pytest.mark.parametrize("concat", [False, True])
def test_two_hop_isolated_seeds_take_the_nan_rule_through_the_split_layer(concat, d0):
    """Seeds drawn from ALL nodes, isolated ones included: their rows are the reference's 0/0 = NaN
    (aggregators.py:60-61) at layer 2 and, for the concat encoder, also at layer 1 (they head S1), where the
    NaN has to survive the split-bf16 contraction of the dense kernel."""
    graph = rmat_graph(14, 300_000, seed=2)
    gen = torch.Generator().manual_seed(4)
    m = 2 if concat else 1
    table = torch.randn(graph.num_nodes, d0, generator=gen)
    w1 = torch.randn(128, m * d0, generator=gen) / np.sqrt(m * d0)
    w2 = torch.randn(64, m * 128, generator=gen) / np.sqrt(m * 128)
    seeds = np.random.default_rng(3).choice(graph.num_nodes, 900, replace=False)
    assert (graph.degrees()[seeds] == 0).sum() > 20
    rowptr, col = torch.from_numpy(graph.rowptr).to(DEV), torch.from_numpy(graph.col).to(DEV)
    eng = TwoHopEngine(rowptr, col, table.to(DEV), w1.to(DEV), w2.to(DEV), 10, 20, concat=concat, max_batch=len(seeds))
    out = eng.forward(torch.from_numpy(seeds.astype(np.int32)).to(DEV), seed=9).cpu()
    assert eng.layout.layer1_split == 1
    assert torch.isnan(out[torch.from_numpy(graph.degrees()[seeds] == 0)]).all()
    check_engine_against_oracle(graph, table, w1, w2, seeds, 10, 20, concat, False, True, seed=9)


@pytest.mark.parametrize("fused", [False, True])
def test_two_hop_baseline_config3_shape(fused):
    """BASELINE.json configs[2]: R-MAT 2^20 nodes / 16 M edges, D0 = 256, H = 128/128, fanout 15/25, B = 4096."""
    graph = rmat_graph(20, 16_000_000, seed=0, cache_dir="/tmp/sage_cache")
    gen = torch.Generator().manual_seed(0)
    table = torch.randn(graph.num_nodes, 256, generator=gen)
    w1 = (torch.rand(128, 256, generator=gen) * 2 - 1) * np.sqrt(6.0 / (128 + 256))
    w2 = (torch.rand(128, 128, generator=gen) * 2 - 1) * np.sqrt(6.0 / (128 + 128))
    deg = graph.degrees()
    seeds = np.random.default_rng(1).choice(np.nonzero(deg > 0)[0], 4096, replace=False)
    err = check_engine_against_oracle(graph, table, w1, w2, seeds, 15, 25, False, False, fused)
    print(f"config-3 batch: max |gpu-oracle| / rowmax = {err:.2e}")


def test_graph_replay_of_queued_batches_equals_direct_calls():
    """Device-side batch queue + hipGraph replay (what bench.py times) must give, batch for batch,
    the bits of the host-enqueued forward with the same seeds and sampler key."""
    graph = rmat_graph(14, 300_000, seed=2)
    gen = torch.Generator().manual_seed(3)
    table = torch.randn(graph.num_nodes, 256, generator=gen).to(DEV)
    w1 = (torch.randn(128, 256, generator=gen) / 16).to(DEV)
    w2 = (torch.randn(128, 128, generator=gen) / 11).to(DEV)
    rowptr, col = graph.to(DEV)
    deg = graph.degrees()
    rs = np.random.default_rng(5)
    # 600 seeds x fanout 25 -> layer 1 is large enough to take the split (column-sliced gather + dense contraction) path
    seeds = torch.from_numpy(np.stack([rs.choice(np.nonzero(deg > 0)[0], 600, replace=False) for _ in range(5)]).astype(np.int32)).to(DEV)
    keys = [11, 2**63 + 5, 13, 2**64 - 1, 17]
    for concat in (False, True):
        m = 2 if concat else 1
        w1c = w1.repeat(1, m).contiguous()
        w2c = w2.repeat(1, m).contiguous()
        direct = TwoHopEngine(rowptr, col, table, w1c, w2c, 15, 25, concat=concat, max_batch=600)
        want = [direct.forward(seeds[i], seed=keys[i]).clone() for i in range(5)]
        eng = TwoHopEngine(rowptr, col, table, w1c, w2c, 15, 25, concat=concat, max_batch=600)
        eng.set_queue(seeds, keys)
        out = eng.capture()
        assert eng.layout.layer1_split == 1
        for rnd in range(2):                       # second round: the ring wraps around
            for i in range(5):
                eng.replay()
                torch.cuda.synchronize()
                assert torch.equal(out, want[i]), f"concat={concat} round {rnd} batch {i}"
        eng.rewind(3)                              # jump inside the ring
        for i in (3, 4, 0):
            eng.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, want[i]), f"concat={concat} after rewind, batch {i}"
        # several batches per replay (one graph launch embeds three consecutive ring entries)
        multi = TwoHopEngine(rowptr, col, table, w1c, w2c, 15, 25, concat=concat, max_batch=600)
        multi.set_queue(seeds, keys)
        out3 = multi.capture(batches=3)
        for first in (0, 3, 1):                    # 0-2, 3-4-0 (wraps), 1-3
            multi.replay()
            torch.cuda.synchronize()
            for j in range(3):
                assert torch.equal(out3[j], want[(first + j) % 5]), f"concat={concat} multi-batch replay, batch {(first + j) % 5}"




def test_integration_md_ctypes_stub():
    """The ctypes stub INTEGRATION.md shows a maintainer (section B), verbatim in spirit: raw CDLL, raw pointers."""
    import ctypes
    from sage355 import native
    L = ctypes.CDLL(native.LIB_PATH)
    P, I32, I64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    L.sage_last_error.restype = ctypes.c_char_p
    L.sage_sample_neighbors.argtypes = [P, P, I64, P, I32, P, I32, ctypes.c_uint64, ctypes.c_uint32, P, P, P, P, I32, P, P, P]
    L.sage_layer_forward.argtypes = [P, I64, I64, I32, P, P, I32, I32, P, P, P, P, I32, P, P, I64, I32, I32, P, I64, P]

    def encoder_forward(table, rowptr, col, nodes_i32, W, k, seed, gcn=True):
        n, D, H = nodes_i32.numel(), table.shape[1], W.shape[0]
        nbr = torch.empty(n, k, dtype=torch.int32, device="cuda")
        cnt = torch.empty(n, dtype=torch.int32, device="cuda")
        out = torch.empty(n, H, device="cuda")
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())   # noqa: E731
        rc = L.sage_sample_neighbors(p(rowptr), p(col), rowptr.numel() - 1, p(nodes_i32), n, None, k, seed, 1,
                                     p(nbr), p(cnt), None, None, 0, None, None, st)
        assert rc == 0, L.sage_last_error()
        rc = L.sage_layer_forward(p(table), table.shape[0], table.stride(0), D, p(nbr), p(cnt), k, n, None, None, None, None,
                                  0 if gcn else 1, None if gcn else p(nodes_i32), p(W), W.stride(0), H, 0, p(out), H, st)
        assert rc == 0, L.sage_last_error()
        return out.t(), nbr, cnt

    graph = rmat_graph(13, 150_000, seed=4)
    gen = torch.Generator().manual_seed(1)
    table = torch.randn(graph.num_nodes, 128, generator=gen)
    rowptr, col = graph.to(DEV)
    nodes = torch.from_numpy(np.random.default_rng(2).choice(np.nonzero(graph.degrees() > 0)[0], 600, replace=False).astype(np.int32))
    for gcn in (True, False):
        W = torch.randn(64, 128 if gcn else 256, generator=gen) / 12
        out, nbr, cnt = encoder_forward(table.to(DEV), rowptr, col, nodes.to(DEV), W.to(DEV), 10, 77, gcn)
        agg = ref_sparse.gather_mean(table, nbr.cpu().numpy(), cnt.cpu().numpy())
        ref = ref_sparse.linear_act(None if gcn else table[nodes.long()], agg, W)
        assert_close_rowmax(out.t().cpu(), ref, what=f"INTEGRATION.md stub gcn={gcn}")
