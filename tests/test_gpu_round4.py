"""Round-4 GPU tests: the trainable feature table, the in-place refresh of the slice-major table copy."""
import numpy as np
import pytest
import torch

from oracle import ref_sparse, sampler_ref
from sage355 import native, ops
from sage355.engine import RolePipeline, TwoHopEngine
from test_gpu_forward import build_modules
from util import assert_close_rowmax, full_table, load_golden, sets_from_padded

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _torch_two_hop(table, w1, w2, g):
    """The reference's expression (aggregators.py:54-74, encoders.py:49-62) on the fixture's injected sets, differentiable, fp64."""
    gcn = bool(g["gcn"])
    l1 = torch.from_numpy(g["layer1_nodes"])
    pos = {int(v): i for i, v in enumerate(g["layer1_nodes"])}

    def mean_rows(src, nbr, cnt, index_of):
        rows = []
        for r in range(nbr.shape[0]):
            ids = [index_of(int(x)) for x in nbr[r, :int(cnt[r])]]
            rows.append(src[ids].mean(0))
        return torch.stack(rows)

    agg1 = mean_rows(table, g["nbr1"], g["cnt1"], lambda x: x)
    x1 = agg1 if gcn else torch.cat([table[l1], agg1], 1)
    h1 = torch.relu(x1 @ w1.t())
    agg2 = mean_rows(h1, g["nbr2"], g["cnt2"], lambda x: pos[x])
    x2 = agg2 if gcn else torch.cat([h1[[pos[int(s)] for s in g["seeds"]]], agg2], 1)
    return torch.relu(x2 @ w2.t())                     # [B, H2]


@pytest.mark.parametrize("table_on_device", [True, False])
@pytest.mark.parametrize("name", ["tiny_gcn", "tiny_concat", "cora_emb_gcn_5_5", "cora_emb_concat_10_10"])
def test_trainable_feature_table_gets_its_gradient(name, table_on_device):
    """under grad mode the fused two-hop node returns gradients for the two `weight` Parameters only; a features table
    with requires_grad=True (nn.Embedding's default -- model.py:215 freezes it explicitly) silently got none.  Such a model now takes the
    per-operator path, whose gather_mean / linear_act backward reach the table, as the reference's autograd does."""
    g = load_golden(name)
    if int(g["cnt1"].min()) == 0 or int(g["cnt2"].min()) == 0:
        pytest.skip("fixture with empty sets (NaN rows): gradients are NaN by the reference's own rule")
    sets1 = sets_from_padded(g["layer1_nodes"], g["nbr1"], g["cnt1"])
    sets2 = sets_from_padded(g["seeds"], g["nbr2"], g["cnt2"])
    enc1, enc2 = build_modules(g, int(g["k1"]), int(g["k2"]), True, sets1, sets2)
    if table_on_device:
        enc2.to("cuda")                       # nn.Module.to: table and both weights on the GPU; otherwise everything stays on the host
    enc1.features.weight.requires_grad_(True)
    assert enc2._can_fuse_two_hop()
    out = enc2([int(s) for s in g["seeds"]])
    assert out.requires_grad and (enc2._engine is None or enc2._engine.generation == 0)      # NOT the fused node
    assert_close_rowmax(out.detach().cpu(), g["enc2_out"], rows_dim=1, what="forward with a trainable table")
    cot = torch.from_numpy(g["cotangent"])
    (out * cot.to(out.device)).sum().backward()
    t64 = full_table(g).double().requires_grad_()
    w1 = torch.from_numpy(g["w1"]).double().requires_grad_()
    w2 = torch.from_numpy(g["w2"]).double().requires_grad_()
    (_torch_two_hop(t64, w1, w2, g) * cot.double().t()).sum().backward()
    tg = enc1.features.weight.grad
    assert tg is not None and tg.is_cuda == table_on_device and float(tg.abs().max()) > 0
    for got, want, what in ((tg, t64.grad, "grad_table"), (enc1.weight.grad, w1.grad, "grad_w1"), (enc2.weight.grad, w2.grad, "grad_w2")):
        err = ((got.double().cpu() - want).abs().max() / want.abs().max()).item()
        assert err < 5e-5, f"{name} {what}: {err:.2e}"
    # the frozen-table model (the reference's own configuration) still runs as ONE autograd node over the engine
    enc1.features.weight.requires_grad_(False)
    enc1.features.weight.grad = None
    out = enc2([int(s) for s in g["seeds"]])
    assert enc2._engine is not None and enc2._engine.generation == 1


def _rmat_problem(scale=15, edges=600_000, d0=256, h1=128, h2=64, seed=3):
    from sage355.graph import rmat_graph
    graph = rmat_graph(scale, edges, seed=seed)
    gen = torch.Generator().manual_seed(seed)
    table = torch.randn(graph.num_nodes, d0, generator=gen)
    w1 = (torch.rand(h1, d0, generator=gen) * 2 - 1) * np.sqrt(6.0 / (h1 + d0))
    w2 = (torch.rand(h2, h1, generator=gen) * 2 - 1) * np.sqrt(6.0 / (h2 + h1))
    return graph, table, w1, w2


def test_slice_major_copy_is_refreshed_in_place_under_a_live_pipe():
    """the slice-major table copy used to be a NEW tensor on every rebuild while a RolePipeline held the old pointer.  It is
    refreshed in place now: after an in-place table update (version bump) or refresh_table() (a `.data` write), pipe and engine read
    the new values through the pointers they already hold."""
    graph, table, w1, w2 = _rmat_problem(scale=14, edges=300_000)
    rowptr, col = graph.to(DEV)
    b, k1, k2 = 2048, 15, 25
    tdev = table.to(DEV)
    cand = np.nonzero(graph.degrees() > 0)[0]
    seeds = torch.from_numpy(np.random.default_rng(2).choice(cand, b, replace=False).astype(np.int32)).to(DEV)
    pipe = RolePipeline(rowptr, col, tdev, w1.to(DEV), w2.to(DEV), k1, k2, batch=b, depth=2)
    e0 = pipe.engines[0]
    if e0._table_sliced is None:
        pytest.skip("this shape does not use the slice-major copy")
    ptr = e0._table_sliced.data_ptr()
    out = torch.empty(2, b, 64, device=DEV)
    pipe.submit(seeds, 5, out[0]); pipe.synchronize()
    before = out[0].clone()
    tdev.mul_(2.0)                                       # in place: the version counter moves
    fresh = TwoHopEngine(rowptr, col, tdev.clone(), w1.to(DEV), w2.to(DEV), k1, k2, max_batch=b)
    want = fresh.forward(seeds, seed=5)
    pipe.submit(seeds, 5, out[1]); pipe.synchronize()    # the submit notices the version: refresh in place, role streams forked behind it
    assert e0._table_sliced.data_ptr() == ptr
    assert torch.equal(out[1], want) and not torch.equal(out[1], before)
    tdev.data.mul_(0.5)                                  # `.data`: no version bump -> explicit refresh
    pipe.refresh_table()
    assert e0._table_sliced.data_ptr() == ptr
    pipe.submit(seeds, 5, out[1]); pipe.synchronize()
    assert torch.equal(out[1], before)


@pytest.mark.parametrize("threads", [False, True])
def test_an_idle_pipes_batch_takes_the_express_lane_and_changes_no_bit(threads):
    """Round 4: a batch submitted to an IDLE RolePipeline is enqueued whole on stream L (no stream-to-stream hand-offs on an empty GPU);
    batches that find earlier ones in flight go through the role streams.  Both are the single forward's kernels on the same workspace
    layout: every output equals TwoHopEngine.forward on the same (seeds, key) bit for bit, whichever lane it took."""
    graph, table, w1, w2 = _rmat_problem(scale=14, edges=300_000)
    rowptr, col = graph.to(DEV)
    b, k1, k2, n = 2048, 15, 25, 12
    cand = np.nonzero(graph.degrees() > 0)[0]
    rs = np.random.default_rng(11)
    seeds = torch.from_numpy(np.stack([rs.choice(cand, b, replace=False) for _ in range(n)]).astype(np.int32)).to(DEV)
    keys = [1000 + i for i in range(n)]
    args = (rowptr, col, table.to(DEV), w1.to(DEV), w2.to(DEV), k1, k2)
    single = TwoHopEngine(*args, max_batch=b)
    want = torch.stack([single.forward(seeds[i], seed=keys[i]).clone() for i in range(n)])
    pipe = RolePipeline(*args, batch=b, depth=4, threads=threads)
    assert pipe.express_count == 0
    # (a) one batch at a time, the pipe drained in between: every batch finds it idle
    out = torch.zeros(n, b, w2.shape[0], device=DEV)
    for i in range(n):
        pipe.submit(seeds[i], keys[i], out[i])
        pipe.synchronize()
    assert pipe.express_count == n
    assert torch.equal(out, want)
    # (b) all at once: the first finds the pipe idle, the ones behind it (submitted within microseconds) find it busy
    out.zero_()
    pipe.submit_many(seeds, keys, out)
    pipe.synchronize()
    took = pipe.express_count - n
    assert 1 <= took < n, took
    assert torch.equal(out, want)
    # (c) and again after the drain, interleaved with single submits: whatever mix of lanes, the same bits
    out.zero_()
    for i in range(n):
        pipe.submit(seeds[i], keys[i], out[i])
        if i % 5 == 4:
            pipe.synchronize()
    pipe.synchronize()
    assert torch.equal(out, want)


def test_bench_rccl_calls_run_with_one_rank():
    """The N > 1 code path of bench.py -- init_process_group("nccl", device_id), the fences' barriers, the MAX all-reduce of the elapsed
    times on a DEVICE tensor, destroy -- has never met hardware with two GPUs (this pool hands out one).  With SAGE_BENCH_FORCE_DIST=1 a
    single rank takes that path through RCCL: the calls themselves are exercised, and the line carries config.closing_fence."""
    import json, os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SAGE_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--dist-backend", "nccl", "--steps", "20", "--warmup", "5",
                        "--cpu-seconds", "0", "--preheat-seconds", "0.05", "--no-variant", "--scale-variant", "off"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    cf = line["config"]["closing_fence"]
    assert line["n_gpus"] == 1 and cf is not None and cf["barrier_us_in_timed_region"] >= 0
    assert 0 < cf["slowest_rank_done_ms_per_step"] <= line["ms_per_step"] + 1e-5
    assert line["timed_path_check"]["bit_identical_to_oracle_gated_forward"] is True
