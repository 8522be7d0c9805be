"""Round-2 GPU tests: the role pipeline, the prepared-weight contraction, the engine's internal degree layout, the kernel
variants behind the launch tunables, BASELINE configs[3] / configs[4] at full size, Pubmed topology, and `bench.py --gpus N`
from a plain shell.  All through the C ABI (sage355 -> libsage355.so), checked against oracle/ (fp64 restatement on the
GPU's own sampled sets, C restatement of the sampler)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import ref_sparse, sampler_ref
from sage355 import native, ops
from sage355.engine import RolePipeline, TwoHopEngine
from sage355.graph import CSRGraph, relabel_by_degree, rmat_graph, truncate_nodes
from test_gpu_forward import check_engine_against_oracle
from util import GOLDEN_DIR, assert_close_rowmax

pytestmark = pytest.mark.gpu
DEV = "cuda"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CACHE = os.environ.get("SAGE_CACHE", "/tmp/sage_cache")


def _problem(scale=15, edges=600_000, d0=256, h1=128, h2=64, concat=False, seed=2):
    graph = rmat_graph(scale, edges, seed=seed, accel=None)
    gen = torch.Generator().manual_seed(0)
    m = 2 if concat else 1
    table = torch.randn(graph.num_nodes, d0, generator=gen)
    w1 = torch.randn(h1, m * d0, generator=gen) / np.sqrt(m * d0)
    w2 = torch.randn(h2, m * h1, generator=gen) / np.sqrt(m * h1)
    return graph, table, w1, w2


def _eq(a, b):
    return torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))


@pytest.mark.parametrize("roles,depth", [("SGDL", 4), ("SGDD", 3), ("SSSS", 1), ("SGDL", 2)])
@pytest.mark.parametrize("concat,self_loop", [(False, False), (True, False), (False, True)])
def test_role_pipeline_is_bit_identical_to_single_forwards(concat, self_loop, roles, depth):
    """sage_pipe_*: the stages of consecutive batches run on role streams over `depth` workspaces; every batch must come out
    exactly as TwoHopEngine.forward (one stream, one workspace) computes it -- including when a workspace is reused."""
    graph, table, w1, w2 = _problem(concat=concat)
    rowptr, col = graph.to(DEV)
    b, k1, k2 = 1024, 15, 25
    cand = np.nonzero(graph.degrees() > 0)[0]
    rs = np.random.default_rng(5)
    nb = 3 * depth + 2
    seeds = torch.from_numpy(np.stack([rs.choice(cand, b, replace=False) for _ in range(nb)]).astype(np.int32)).to(DEV)
    keys = [1000 + i for i in range(nb)]
    kw = dict(concat=concat, agg_self_loop=self_loop)
    eng = TwoHopEngine(rowptr, col, table.to(DEV), w1.to(DEV), w2.to(DEV), k1, k2, max_batch=b, **kw)
    want = [eng.forward(seeds[i], seed=keys[i]).clone() for i in range(nb)]
    pipe = RolePipeline(rowptr, col, table.to(DEV), w1.to(DEV), w2.to(DEV), k1, k2, batch=b, depth=depth, roles=roles, **kw)
    out = torch.empty(nb, b, w2.shape[0], device=DEV)
    torch.cuda.synchronize()
    pipe.submit_many(seeds[:nb - 2], keys[:nb - 2], out)
    for i in (nb - 2, nb - 1):
        pipe.submit(seeds[i], keys[i], out[i])
    pipe.synchronize()
    for i in range(nb):
        assert _eq(out[i], want[i]), f"batch {i} of the pipeline differs from the single forward"


def test_role_pipeline_argument_checks():
    graph, table, w1, w2 = _problem(scale=12, edges=40_000, d0=64, h1=32, h2=16)
    rowptr, col = graph.to(DEV)
    with pytest.raises(native.SageError):
        RolePipeline(rowptr, col, table.to(DEV), w1.to(DEV), w2.to(DEV), 5, 5, batch=64, depth=native.PIPE_MAX_DEPTH + 1)
    pipe = RolePipeline(rowptr, col, table.to(DEV), w1.to(DEV), w2.to(DEV), 5, 5, batch=64, depth=2)
    with pytest.raises(native.SageError):
        pipe.submit(torch.zeros(63, dtype=torch.int32, device=DEV), 1, torch.empty(64, 16, device=DEV))
    with pytest.raises(native.SageError):
        pipe.submit_many(torch.zeros(3, 64, dtype=torch.int32, device=DEV), [1, 2], torch.empty(4, 64, 16, device=DEV))


@pytest.mark.parametrize("concat", [False, True])
@pytest.mark.parametrize("d0,h1", [(256, 128), (128, 64), (64, 128), (100, 52), (500, 50), (1433, 128)])
def test_prepared_weight_planes_do_not_change_a_bit(d0, h1, concat):
    """sage_prepare_weights moves the bf16 split of W out of the contraction kernel; same planes, same MFMAs, same sums -- for the
    one-pass layer, for the concat encoder's [self | agg] weight staged as one pass (2 x dim <= 256) and as two (dim = 256), and for
    rows wider than 256 floats (Pubmed 500, Cora 1433 padded to 1436: ceil(dim / 256) passes per chunk)."""
    graph, table, w1, w2 = _problem(d0=d0, h1=h1, concat=concat)
    rowptr, col = graph.to(DEV)
    seeds = torch.from_numpy(np.random.default_rng(3).choice(np.nonzero(graph.degrees() > 0)[0], 2048, replace=False).astype(np.int32)).to(DEV)
    outs = []
    for prep in (True, False):
        eng = TwoHopEngine(rowptr, col, table.to(DEV), w1.to(DEV), w2.to(DEV), 15, 25, concat=concat, max_batch=2048, prepare_weights=prep)
        assert bool(eng.layout.layer1_split)
        assert (eng._model().w1_prepared is not None) == prep
        outs.append(eng.forward(seeds, seed=9).clone())
    assert _eq(outs[0], outs[1])
    # an in-place weight update is followed (version counter), as an optimizer step does it
    eng = TwoHopEngine(rowptr, col, table.to(DEV), w1.to(DEV).clone(), w2.to(DEV), 15, 25, concat=concat, max_batch=2048)
    a = eng.forward(seeds, seed=9).clone()
    eng.w1.mul_(-0.5)
    bb = eng.forward(seeds, seed=9).clone()
    ref = TwoHopEngine(rowptr, col, table.to(DEV), eng.w1.clone(), w2.to(DEV), 15, 25, concat=concat, max_batch=2048, prepare_weights=False).forward(seeds, seed=9)
    assert not _eq(a, bb) and _eq(bb, ref)


@pytest.mark.parametrize("concat,self_loop", [(False, False), (True, True)])
def test_engine_degree_layout_keeps_the_callers_ids(concat, self_loop):
    """TwoHopEngine(relabel="degree"): the engine's own device layout.  Seeds arrive in the CALLER's ids and are translated
    inside the outer-hop kernel; the result equals the plain engine run on the host-renumbered graph, row for row."""
    graph, table, w1, w2 = _problem(concat=concat)
    g2, new_of_old = relabel_by_degree(graph)
    order = np.argsort(new_of_old)
    cand = np.nonzero(graph.degrees() > 0)[0]
    seeds = np.random.default_rng(8).choice(cand, 1500, replace=False)
    kw = dict(concat=concat, agg_self_loop=self_loop, max_batch=1500)
    rp, cl = graph.to(DEV)
    a = TwoHopEngine(rp, cl, table.to(DEV), w1.to(DEV), w2.to(DEV), 10, 20, relabel="degree", **kw)
    out_a = a.forward(torch.from_numpy(seeds.astype(np.int32)).to(DEV), seed=77)
    rp2, cl2 = g2.to(DEV)
    b = TwoHopEngine(rp2, cl2, table[torch.from_numpy(order)].to(DEV), w1.to(DEV), w2.to(DEV), 10, 20, **kw)
    out_b = b.forward(torch.from_numpy(new_of_old[seeds].astype(np.int32)).to(DEV), seed=77)
    assert _eq(out_a, out_b)
    assert np.array_equal(a.node_order.cpu().numpy(), order)
    # and against the oracle, in the caller's ids: translate the engine's internal ids back
    it = a.intermediates()
    back = a.node_order.cpu().numpy()
    first = it["first_frontier_row"]
    nbr2, cnt2 = it["nbr2"].cpu().numpy(), it["cnt2"].cpu().numpy()
    nbr1, cnt1 = it["nbr1"].cpu().numpy(), it["cnt1"].cpu().numpy()
    s1 = back[it["s1_nodes"].cpu().numpy()]
    tr = lambda x: np.where(x >= 0, back[np.maximum(x, 0)], -1)
    ref = ref_sparse.two_hop_forward(table, w1, w2, seeds, tr(nbr2), cnt2, s1[first:], tr(nbr1[first:]), cnt1[first:], gcn=not concat,
                                     agg_gcn=self_loop, seed_nbr1=tr(nbr1[:first]) if concat else None,
                                     seed_cnt1=cnt1[:first] if concat else None)
    assert_close_rowmax(out_a.cpu(), ref, what="degree layout vs oracle in caller ids")


def test_out_of_range_seed_ids_are_isolated_nodes_on_the_device_and_errors_on_the_host():
    """a bad id must never walk rowptr[].  Host lists raise; device-resident ids sample nothing."""
    graph, table, w1, w2 = _problem(scale=12, edges=40_000, d0=64, h1=32, h2=16)
    rp, cl = graph.to(DEV)
    eng = TwoHopEngine(rp, cl, table.to(DEV), w1.to(DEV), w2.to(DEV), 5, 5, max_batch=8, nan_empty=False)
    with pytest.raises(native.SageError):
        eng.forward([0, 1, graph.num_nodes], seed=1)
    with pytest.raises(native.SageError):
        eng.forward([-1, 2], seed=1)
    good = np.nonzero(graph.degrees() > 0)[0][:6].astype(np.int32)
    ids = torch.from_numpy(np.concatenate([good, np.array([graph.num_nodes + 5, -3], np.int32)])).to(DEV)
    out = eng.forward(ids, seed=1)
    cnt2 = eng.intermediates()["cnt2"].cpu().numpy()
    assert (cnt2[:6] > 0).all() and (cnt2[6:] == 0).all()
    assert torch.isfinite(out[:6]).all()
    with pytest.raises(native.SageError):
        eng.set_queue(ids[None, :].contiguous(), [1])


VARIANT_SCRIPT = r"""
import sys
sys.path[:0] = [{repo!r}, {repo!r} + "/graphsage-simple_amd", {repo!r} + "/tests"]
import numpy as np, torch
from sage355.graph import rmat_graph
from test_gpu_forward import check_engine_against_oracle
graph = rmat_graph(15, 600_000, seed=2, accel=None)
gen = torch.Generator().manual_seed(0)
for d0, concat, self_loop in ((256, False, False), (256, True, False), (100, False, True), (128, False, False)):
    m = 2 if concat else 1
    table = torch.randn(graph.num_nodes, d0, generator=gen)
    w1 = torch.randn(128, m * d0, generator=gen) / np.sqrt(m * d0)
    w2 = torch.randn(64, m * 128, generator=gen) / np.sqrt(m * 128)
    seeds = np.random.default_rng(1).choice(np.nonzero(graph.degrees() > 0)[0], 2048, replace=False)
    err = check_engine_against_oracle(graph, table, w1, w2, seeds, 15, 25, concat, self_loop, True)
    print("ok", d0, concat, self_loop, err)
print("VARIANT_OK")
"""

VARIANTS = [
    {"SAGE_G_VARIANT": "0", "SAGE_G_PER_CU": "8"},
    {"SAGE_G_VARIANT": "1", "SAGE_G_ROWS": "2", "SAGE_G_PER_CU": "4"},
    {"SAGE_G_VARIANT": "1", "SAGE_G_ROWS": "4", "SAGE_G_SLICE_LANES": "8"},
    {"SAGE_G_VARIANT": "2", "SAGE_G_TRIP": "8", "SAGE_G_PER_CU": "2"},
    {"SAGE_G_VARIANT": "2", "SAGE_G_TRIP": "16", "SAGE_G_SLICE_LANES": "8", "SAGE_G_PER_CU": "3"},
    {"SAGE_G_VARIANT": "2", "SAGE_G_SLICE_LANES": "32"},
    {"SAGE_G_VARIANT": "1", "SAGE_G_SLICE_LANES": "64"},
    {"SAGE_G_VARIANT": "2", "SAGE_G_SLICE_LANES": "64"},
    {"SAGE_DENSE_BLOCKS": "512", "SAGE_T16_WAVES": "8", "SAGE_SO_THREADS": "256"},
    {"SAGE_DENSE_BLOCKS": "96", "SAGE_SO_THREADS": "512", "SAGE_T16_GRID": "128"},
    {"SAGE_SAMPLE_FUSED": "1", "SAGE_T16_WAVES": "16"},
    {"SAGE_SAMPLE_FUSED": "1", "SAGE_SO_THREADS": "512", "SAGE_T16_WAVES": "8"},
]


@pytest.mark.parametrize("env", VARIANTS, ids=lambda e: ",".join(f"{k[5:]}={v}" for k, v in e.items()))
def test_kernel_variants_behind_the_launch_tunables(env, tmp_path):
    """Every kernel variant selectable by a SAGE_* tunable (read once per process, csrc/sage_api.hip) against the oracle:
    samplers bit-exact, values within 1e-5 of the row maximum.  One child process per setting, one at a time."""
    script = tmp_path / "variant.py"
    script.write_text(VARIANT_SCRIPT.format(repo=REPO))
    e = dict(os.environ)
    e.update(env)
    res = subprocess.run([sys.executable, str(script)], env=e, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "VARIANT_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]


# ------------------------------------------------------------------------------------------ size-independent properties at full size
@pytest.mark.parametrize("relabel", [None, "degree"])
def test_full_size_properties_homogeneity_permutation_sub_batch(relabel):
    """BASELINE configs[2] at full size (2^20 nodes, 16 M edges, D0 = 256, 4096 seeds), properties that need no oracle and hold BIT
    FOR BIT: (1) both layers are positively homogeneous, and a factor of two is exact in every step (the three-term bf16 split
    included): out(2 X) = 2 out(X), out(4 W1, W2 / 2) = 2 out; (2) permuting the seeds permutes the rows (a node's draws are a
    function of (key, node), and no sum depends on the frontier's order); (3) a sub-batch gives the same rows as the batch it
    was cut from; (4) the same call twice gives the same bits."""
    graph = rmat_graph(20, 16_000_000, seed=0, cache_dir=CACHE)
    gen = torch.Generator(device=DEV).manual_seed(0)
    table = torch.randn(graph.num_nodes, 256, generator=gen, device=DEV)
    w1 = torch.randn(128, 256, generator=gen, device=DEV) / 16
    w2 = torch.randn(128, 128, generator=gen, device=DEV) / 11
    rowptr, col = graph.to(DEV)
    cand = np.nonzero(graph.degrees() > 0)[0]
    seeds = torch.from_numpy(np.random.default_rng(1).choice(cand, 4096, replace=False).astype(np.int32)).to(DEV)
    kw = dict(max_batch=4096, relabel=relabel)
    eng = TwoHopEngine(rowptr, col, table, w1, w2, 15, 25, **kw)
    a = eng.forward(seeds, seed=5).clone()
    assert torch.isfinite(a).all() and float(a.abs().max()) > 0
    assert torch.equal(eng.forward(seeds, seed=5), a)
    assert torch.equal(TwoHopEngine(rowptr, col, table * 2, w1, w2, 15, 25, **kw).forward(seeds, seed=5), a * 2)
    assert torch.equal(TwoHopEngine(rowptr, col, table, w1 * 4, w2 * 0.5, 15, 25, **kw).forward(seeds, seed=5), a * 2)
    perm = torch.randperm(4096, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    assert torch.equal(eng.forward(seeds[perm].contiguous(), seed=5), a[perm])
    assert torch.equal(eng.forward(seeds[:1000].contiguous(), seed=5), a[:1000])
    assert not torch.equal(eng.forward(seeds, seed=6), a)                # another key: other draws


# ------------------------------------------------------------------------------------------ BASELINE configs at full size
def _full_size_check(graph, d0, k1, k2, concat, self_loop, h1=128, h2=128, b=4096, relabel=None):
    gen = torch.Generator().manual_seed(0)
    m = 2 if concat else 1
    table = torch.randn(graph.num_nodes, d0, generator=gen)
    w1 = (torch.rand(h1, m * d0, generator=gen) * 2 - 1) * np.sqrt(6.0 / (h1 + m * d0))
    w2 = (torch.rand(h2, m * h1, generator=gen) * 2 - 1) * np.sqrt(6.0 / (h2 + m * h1))
    seeds = np.random.default_rng(1).choice(np.nonzero(graph.degrees() > 0)[0], b, replace=False)
    return check_engine_against_oracle(graph, table, w1, w2, seeds, k1, k2, concat, self_loop, True)


def test_config4_rmat_8m_nodes_128m_edges_full_batch():
    """BASELINE configs[3]: R-MAT 2^23 nodes / 128 M generated edges, 256-dim features, fanout 15/25, B = 4096, gcn encoder.
    Sampled sets bit-exact against oracle/sampler_ref.c, frontier = the set union, values against the fp64 oracle."""
    graph = rmat_graph(23, 128_000_000, seed=0, cache_dir=CACHE)
    assert graph.num_nodes == 1 << 23 and graph.nnz > 200_000_000
    _full_size_check(graph, 256, 15, 25, False, False)


@pytest.mark.parametrize("concat,self_loop", [(True, False), (False, True)])
def test_config5_products_shaped_fanout_20_25(concat, self_loop):
    """BASELINE configs[4]: 2.4 M nodes / 62 M generated edges (scale-22 R-MAT truncated), 100-dim features, fanout 20/25,
    B = 4096: the concat encoder, and the GCN-variant aggregator (self-loop union; intended semantics of aggregators.py:50-51 --
    the reference line raises TypeError, so that half is parity-unpinned and checked against the restatement only)."""
    graph = truncate_nodes(rmat_graph(22, 62_000_000, seed=0, cache_dir=CACHE), 2_400_000)
    assert graph.num_nodes == 2_400_000
    _full_size_check(graph, 100, 20, 25, concat, self_loop)


@pytest.mark.parametrize("concat", [False, True])
def test_config2_pubmed_topology_batch_256(concat):
    """BASELINE configs[1]: Pubmed topology (19717 nodes), 500-dim features, H = 50 / 128, fanout 10 / 25, batch 256."""
    z = np.load(os.path.join(GOLDEN_DIR, "pubmed_topology.npz"))
    graph = CSRGraph(z["rowptr"], z["col"], len(z["rowptr"]) - 1)
    gen = torch.Generator().manual_seed(0)
    table = torch.rand(graph.num_nodes, 500, generator=gen) * (torch.rand(graph.num_nodes, 500, generator=gen) < 0.1)
    m = 2 if concat else 1
    w1 = (torch.rand(50, m * 500, generator=gen) * 2 - 1) * np.sqrt(6.0 / (50 + m * 500))
    w2 = (torch.rand(128, m * 50, generator=gen) * 2 - 1) * np.sqrt(6.0 / (128 + m * 50))
    seeds = np.random.default_rng(1).choice(np.nonzero(graph.degrees() > 0)[0], 256, replace=False)
    check_engine_against_oracle(graph, table, w1, w2, seeds, 10, 25, concat, False, True)


def test_bench_gpus_2_from_a_plain_shell():
    """`python bench.py --gpus 2` must start its own ranks.  Two ranks share this box's one GPU (gloo), Pubmed-sized
    configuration so that it takes seconds; the line must say n_gpus = 2 and carry a whole-job value."""
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--share-device", "--dist-backend", "gloo", "--config", "2",
           "--steps", "20", "--warmup", "10", "--cpu-seconds", "0", "--preheat-seconds", "0.05"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 20 and line["value"] > 0 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 2 * line["config"]["batch_per_gpu"]


def test_bench_two_ranks_say_how_the_pipeline_was_fed():
    """every rank of the role pipeline starts four spinning host threads beside its submitter, so bench.py turns them on only
    where each rank of the node has five usable cores to itself -- and the line says which mode ran (config.host).  Two ranks share this
    box's one GPU (gloo), the headline workload, no side variants."""
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--share-device", "--dist-backend", "gloo", "--steps", "20", "--warmup", "5",
           "--cpu-seconds", "0", "--preheat-seconds", "0.05", "--no-variant", "--scale-variant", "off"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    h = line["config"]["host"]
    assert line["n_gpus"] == 2 and h["ranks_on_node"] == 2 and h["cores_per_rank"] == h["host_cores_usable"] // 2
    assert h["role_threads"] == (h["cores_per_rank"] >= 5) and h["enqueue_mode"] in ("one host thread per role stream", "submitting thread only")
    assert ("host enqueue thread" in line["config"]["execution"]) == h["role_threads"]
    assert line["timed_path_check"]["bit_identical_to_oracle_gated_forward"] is True
    # N > 1: the line says what the closing barrier added to the K-step region; `value` keeps the contract's bracket (after the barrier)
    cf = line["config"]["closing_fence"]
    assert 0 < cf["slowest_rank_done_ms_per_step"] <= line["ms_per_step"] + 1e-5 and cf["barrier_us_in_timed_region"] >= 0
    assert abs(line["value"] - 2 * 4096 / (line["ms_per_step"] * 1e-3)) <= 2e-3 * line["value"]


# ------------------------------------------------------------------------------------------ the split-bf16 contraction, adversarially
def _contract_through_the_engine(x_rows, w1, d0=256, prepare=True):
    """h1 = x . W1^T as the ENGINE's layer-1 contraction computes it (dense_bf16x3_kernel): a perfect-matching graph
    (every node has exactly one neighbour) with fanout 1 makes the layer-1 mean of frontier node p(v) exactly table[v]."""
    b = x_rows.shape[0]
    assert b == 4096                                   # 2 B = 8192 layer-1 rows: the split (sliced gather + dense contraction) path
    n = 2 * b
    rowptr = torch.arange(n + 1, dtype=torch.int64, device=DEV)
    col = torch.cat([torch.arange(b, n), torch.arange(0, b)]).to(torch.int32).to(DEV)      # partner(i) = i +- b
    table = torch.zeros(n, d0)
    table[:b] = x_rows
    w2 = torch.zeros(8, w1.shape[0])
    eng = TwoHopEngine(rowptr, col, table.to(DEV), w1.to(DEV), w2.to(DEV), 1, 1, act1=ops.ACT_NONE, act2=ops.ACT_NONE, max_batch=b,
                       nan_empty=False, prepare_weights=prepare)
    assert bool(eng.layout.layer1_split)
    eng.forward(torch.arange(b, dtype=torch.int32, device=DEV), seed=1)
    it = eng.intermediates()
    s1 = it["s1_nodes"].cpu().numpy()
    assert len(s1) == b and set(s1.tolist()) == set(range(b, n))
    h1 = torch.empty(b, w1.shape[0])
    h1[torch.from_numpy(s1 - b)] = it["h1"].cpu()      # frontier node b + j carries x_rows[j]
    return h1


def _norm_err(got, x, w):
    ref = x.double() @ w.double().t()
    scale = x.double().abs() @ w.double().abs().t()            # sum_k |x_k| |w_k| per output element
    return ((got.double() - ref).abs() / (2.0 ** -23 * scale).clamp_min(1e-300)).max().item(), ref


def test_bf16x3_contraction_on_adversarial_finite_data():
    """VERDICT r1 #6.  x.w from the three bf16 terms of x and w must be as good as an fp32 dot product on data that is NOT
    benign: magnitudes from 1e-6 to 1e+6 inside one row, cancelling pairs, values next to the fp32 maximum, denormals.
    Bar per ELEMENT (not per row maximum): |err| <= 4 * 2^-23 * sum_k |x_k||w_k| (+ the denormal allowance below), and no
    worse than twice what torch's own fp32 mm leaves on the same data."""
    gen = torch.Generator().manual_seed(11)
    b, d0, h1 = 4096, 256, 128
    w = torch.randn(h1, d0, generator=gen) / 16
    sign = lambda shape: (torch.randint(0, 2, shape, generator=gen) * 2 - 1).float()
    x = torch.randn(b, d0, generator=gen)
    # rows 0..1023: magnitudes 10^U(-6, 6)
    x[:1024] = sign((1024, d0)) * 10.0 ** (torch.rand(1024, d0, generator=gen) * 12 - 6)
    # rows 1024..2047: cancelling pairs against (nearly) equal weights -> results tiny against sum |x||w|
    base = torch.randn(1024, d0 // 2, generator=gen) * 1000
    x[1024:2048, 0::2], x[1024:2048, 1::2] = base, -base
    wc = w.clone()
    wc[:, 1::2] = wc[:, 0::2] * (1 + 2.0 ** -20)
    # rows 2048..3071: around 2^126 (below the kernel's exact-path threshold 2^127) with a few huge ones that take the exact path
    x[2048:3072] = sign((1024, d0)) * (2.0 ** 125) * (1 + torch.rand(1024, d0, generator=gen))
    x[2048:2560, 7] = 3.0e38
    x[2560:2600, 100] = -3.4e38
    # rows 3072..: fp32 denormals and values just above them
    x[3072:] = sign((1024, d0)) * torch.rand(1024, d0, generator=gen) * 2.0e-38
    for weights, rows, small_w in ((w, slice(0, 1024), False), (wc, slice(1024, 2048), False), (w * 1e-3, slice(2048, 3072), True),
                                   (w, slice(3072, 4096), False)):
        wt = weights
        got = _contract_through_the_engine(x, wt)[rows]
        xs = x[rows]
        e_ours, ref = _norm_err(got, xs, wt)
        e_torch, _ = _norm_err(xs @ wt.t(), xs, wt)
        # denormal allowance: the matrix pipe may flush bf16 denormal terms (|term| < 2^-126): at most K * 2^-126 * max|w| absolute
        allow = d0 * 2.0 ** -126 * wt.abs().max().item()
        bad = ((got.double() - ref).abs() - allow).clamp_min(0) / (2.0 ** -23 * (xs.double().abs() @ wt.double().abs().t())).clamp_min(1e-300)
        assert torch.isfinite(got).all()
        assert bad.max().item() <= 4.0, f"rows {rows}: normalised error {bad.max().item():.2f} (torch fp32 mm: {e_torch:.2f})"
        assert bad.max().item() <= max(2.0 * e_torch, 1.0), f"rows {rows}: {bad.max().item():.2f} vs torch fp32 {e_torch:.2f}"


@pytest.mark.parametrize("prepare", [True, False])
def test_bf16x3_contraction_propagates_inf_and_nan_like_torch_mm(prepare):
    """+-Inf and NaN inputs / weights: same class (NaN, +Inf, -Inf, finite) in every output element as torch.mm in fp32."""
    gen = torch.Generator().manual_seed(12)
    b, d0, h1 = 4096, 256, 128
    w = torch.randn(h1, d0, generator=gen) / 16
    w[w == 0] = 0.01
    x = torch.randn(b, d0, generator=gen)
    inf = float("inf")
    x[5, 17] = inf                       # one +Inf: +-Inf by the sign of the weight
    x[40, 3], x[40, 200] = inf, -inf     # both: NaN or Inf depending on the signs of the two weights
    x[77, 255] = float("nan")
    x[100, 0] = -inf
    x[4095, 128] = inf                   # last row, second K half
    x[2048:2080, 64] = inf               # a whole 32-row tile
    for weights in (w, None):
        wt = w.clone()
        if weights is None:              # non-finite WEIGHTS: every tile takes the exact path
            wt[3, 9], wt[64, 200] = inf, float("nan")
        got = _contract_through_the_engine(x, wt, prepare=prepare)
        ref = x @ wt.t()
        assert torch.equal(torch.isnan(got), torch.isnan(ref))
        assert torch.equal(torch.isposinf(got), torch.isposinf(ref)) and torch.equal(torch.isneginf(got), torch.isneginf(ref))
        fin = torch.isfinite(ref)
        rows_fin = fin.all(1)
        assert_close_rowmax(got[rows_fin], (x.double() @ wt.double().t())[rows_fin], what="rows without a non-finite value")


@pytest.mark.parametrize("d0", [256, 128])
@pytest.mark.parametrize("prepare", [True, False])
def test_bf16x3_concat_contraction_finite_and_non_finite(prepare, d0):
    """The concat encoder's 2 x dim-deep contraction (two K passes at dim = 256, one at 128), W as prepared planes or as it is:
    h1 = W_self . y + W_agg . x on a perfect-matching graph (frontier node b + j has the own row y_j and the neighbour mean x_j);
    finite rows within 6 x 2^-23 sum|x||w| of fp64 (or twice torch's fp32 mm), Inf / NaN classes as torch.mm, huge weights through the exact path."""
    gen = torch.Generator().manual_seed(21)
    b, h1 = 4096, 128
    n = 2 * b
    x = torch.randn(b, d0, generator=gen)
    y = torch.randn(b, d0, generator=gen) * torch.logspace(-3, 3, b)[:, None]
    inf = float("inf")
    x[9, 5], y[9, 6] = inf, 1.0
    y[300, d0 - 1] = -inf
    x[301, 0], y[301, 0] = inf, -inf
    y[4000, 17] = float("nan")
    rowptr = torch.arange(n + 1, dtype=torch.int64, device=DEV)
    col = torch.cat([torch.arange(b, n), torch.arange(0, b)]).to(torch.int32).to(DEV)
    table = torch.cat([x, y])                                  # node j < b: x_j; node b + j: y_j
    w = torch.randn(h1, 2 * d0, generator=gen) / 20
    w[w == 0] = 0.01
    for huge_w in (False, True):
        wt = w.clone()
        if huge_w:
            wt[7, 3], wt[90, d0 + 11] = 3e38, inf             # one in each half
        eng = TwoHopEngine(rowptr, col, table.to(DEV), wt.to(DEV), torch.zeros(8, 2 * h1, device=DEV), 1, 1, concat=True, act1=ops.ACT_NONE,
                           act2=ops.ACT_NONE, max_batch=b, nan_empty=False, prepare_weights=prepare)
        assert bool(eng.layout.layer1_split) and (eng._model().w1_prepared is not None) == prepare
        eng.forward(torch.arange(b, dtype=torch.int32, device=DEV), seed=1)
        it = eng.intermediates()
        first = it["first_frontier_row"]
        s1 = it["s1_nodes"].cpu().numpy()[first:]
        assert first == b and set(s1.tolist()) == set(range(b, n))
        got = torch.empty(b, h1)
        got[torch.from_numpy(s1 - b)] = it["h1"].cpu()[first:]
        comb = torch.cat([y, x], 1)                            # encoders.py:54: [self | agg]
        ref = comb @ wt.t()
        assert torch.equal(torch.isnan(got), torch.isnan(ref))
        assert torch.equal(torch.isposinf(got), torch.isposinf(ref)) and torch.equal(torch.isneginf(got), torch.isneginf(ref))
        cols_fin = torch.isfinite(wt).all(1) & (wt.abs().amax(1) < 1e30)      # outputs whose weights are ordinary numbers
        rows_fin = torch.isfinite(ref[:, cols_fin]).all(1) & torch.isfinite(comb).all(1)
        assert rows_fin.sum() > 4000 and cols_fin.sum() >= h1 - 2
        err, _ = _norm_err(got[rows_fin][:, cols_fin], comb[rows_fin], wt[cols_fin])
        e_torch, _ = _norm_err((comb[rows_fin] @ wt[cols_fin].t()), comb[rows_fin], wt[cols_fin])
        assert err <= max(6.0, 2.0 * e_torch), f"concat contraction: {err:.2f} x 2^-23 sum|x||w| (torch fp32 mm: {e_torch:.2f})"


def test_sigmoid_activation_matches_torch_over_the_fp32_range():
    """the sigmoid switch (encoders.py:58-59) was pinned only by one 10-seed fixture.  linear_act with an identity
    weight is act(x): compare with torch.sigmoid from -100 to 100 (saturation, the steep part, tiny arguments)."""
    x = torch.cat([torch.linspace(-100, 100, 4001), torch.tensor([0.0, -0.0, 1e-8, -1e-8, 88.0, -88.0, 20.0, -20.0])])
    n = x.numel()
    pad = (-n) % 4
    x = torch.cat([x, torch.zeros(pad)])
    agg = x.view(-1, 4).contiguous().to(DEV)
    w = torch.eye(4, device=DEV)
    got = ops.linear_act(agg, w, act=ops.ACT_SIGMOID).cpu().view(-1)
    want = torch.sigmoid(x.double())
    assert (got.double() - want).abs().max().item() <= 2e-7


@pytest.mark.parametrize("concat,self_loop", [(False, False), (True, True)])
def test_fused_sampler_leaves_the_same_sets_and_rows(concat, self_loop, tmp_path):
    """SAGE_SAMPLE_FUSED=1 (both hops in one launch, slots resolved by the layer-2 kernel): the sampled sets are a function of
    (key, node, hop) only, so outputs must equal the two-launch sampler's up to the frontier's arbitrary row order -- compared
    here through what does not depend on that order: per-seed outputs (bit for bit: every kernel downstream works per row)."""
    script = tmp_path / "fused.py"
    script.write_text(f"""
import sys
sys.path[:0] = [{REPO!r}, {REPO!r} + "/graphsage-simple_amd", {REPO!r} + "/tests"]
import numpy as np, torch
from sage355.engine import TwoHopEngine
from sage355.graph import rmat_graph
graph = rmat_graph(15, 600_000, seed=2, accel=None)
gen = torch.Generator().manual_seed(0)
concat, self_loop = {concat}, {self_loop}
m = 2 if concat else 1
table = torch.randn(graph.num_nodes, 128, generator=gen).cuda()
w1 = (torch.randn(128, m * 128, generator=gen) / 16).cuda(); w2 = (torch.randn(64, m * 128, generator=gen) / 11).cuda()
rp, cl = graph.to("cuda")
seeds = torch.from_numpy(np.random.default_rng(1).choice(np.nonzero(graph.degrees() > 0)[0], 2048, replace=False).astype(np.int32)).cuda()
eng = TwoHopEngine(rp, cl, table, w1, w2, 15, 25, concat=concat, agg_self_loop=self_loop, max_batch=2048)
out = eng.forward(seeds, seed=5)
it = eng.intermediates()
torch.save(dict(out=out.cpu(), nbr2=it["nbr2"].cpu(), cnt2=it["cnt2"].cpu(), s1=torch.sort(it["s1_nodes"].cpu()).values,
                rows_ok=bool(((it["row2"] >= 0) == (it["nbr2"] >= 0)).all())), sys.argv[1])
""")
    res = {}
    for fused in ("0", "1"):
        e = dict(os.environ)
        e["SAGE_SAMPLE_FUSED"] = fused
        path = str(tmp_path / f"r{fused}.pt")
        r = subprocess.run([sys.executable, str(script), path], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        res[fused] = torch.load(path, weights_only=True)
    a, b = res["0"], res["1"]
    assert torch.equal(a["nbr2"], b["nbr2"]) and torch.equal(a["cnt2"], b["cnt2"]) and torch.equal(a["s1"], b["s1"])
    assert a["rows_ok"] and b["rows_ok"]
    assert _eq(a["out"], b["out"])


def test_graph_replay_from_a_batch_queue_with_the_degree_layout():
    """Device batch queue + hipGraph replay (sage_model_t.queue) with the engine's internal degree layout: the queued seeds are in
    the caller's ids and the outer-hop kernel translates them (model.seed_map) -- replays equal direct forwards bit for bit."""
    graph, table, w1, w2 = _problem()
    rp, cl = graph.to(DEV)
    cand = np.nonzero(graph.degrees() > 0)[0]
    rs = np.random.default_rng(4)
    seeds = torch.from_numpy(np.stack([rs.choice(cand, 512, replace=False) for _ in range(5)]).astype(np.int32)).to(DEV)
    keys = [31, 32, 33, 34, 35]
    eng = TwoHopEngine(rp, cl, table.to(DEV), w1.to(DEV), w2.to(DEV), 10, 20, max_batch=512, relabel="degree")
    want = [eng.forward(seeds[i], seed=keys[i]).clone() for i in range(5)]
    eng.set_queue(seeds, keys)
    out = eng.capture()
    for i in range(5):
        eng.replay()
        torch.cuda.synchronize()
        assert _eq(out, want[i]), f"replay {i}"
