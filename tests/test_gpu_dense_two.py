"""The 512-deep concat contraction as TWO launches of the lock-step kernel (csrc/sage_dense.hip: dense_bf16x3_kernel's
SRC / EPI forms; sage_set_option("dense_two", 1) / SAGE_DENSE_TWO=1): the nodes' own rows' chunk -> partial sums in h1 (the role
pipeline runs it on stream D BESIDE the gather: it needs the sampling only), then h1 = act(h1 + means' chunk).  This file runs the concat
contraction's tests with the option on -- values against the fp64 oracle, the Inf / NaN classes of torch.mm (encoders.py:58-61), the
pipeline against single forwards -- and compares the two forms with each other."""
import numpy as np
import pytest
import torch

import test_gpu_round2 as r2
from sage355 import native
from sage355.engine import RolePipeline, TwoHopEngine

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture
def two_on():
    native.check(native.lib().sage_set_option(b"dense_two", 1), "set_option")
    yield
    native.check(native.lib().sage_set_option(b"dense_two", -1), "set_option")


def test_two_launch_form_against_the_two_pass_kernel():
    """Same six bf16 products per k-step; the sums are ASSOCIATED differently -- two launches: (self h0 + self h1) + (agg h0 + agg h1),
    two passes: (self h0 + agg h0) + (self h1 + agg h1) -- so the outputs agree to a few fp32 roundings of the row maximum, not bit for bit;
    both are inside the 1e-5 bar of the fp64 oracle (the tests below and tests/test_gpu_round2.py)."""
    graph, table, w1, w2 = r2._problem(d0=256, h1=128, concat=True)
    rowptr, col = graph.to(DEV)
    seeds = torch.from_numpy(np.random.default_rng(3).choice(np.nonzero(graph.degrees() > 0)[0], 2048, replace=False).astype(np.int32)).to(DEV)
    eng = TwoHopEngine(rowptr, col, table.to(DEV), w1.to(DEV), w2.to(DEV), 15, 25, concat=True, max_batch=2048)
    assert bool(eng.layout.layer1_split) and eng._model().w1_prepared is not None
    one = eng.forward(seeds, seed=9).clone()
    native.check(native.lib().sage_set_option(b"dense_two", 1), "set_option")
    try:
        two = eng.forward(seeds, seed=9).clone()
        again = eng.forward(seeds, seed=9).clone()
    finally:
        native.lib().sage_set_option(b"dense_two", -1)
    scale = one.abs().amax(1, keepdim=True).clamp_min(1e-30)
    assert ((two - one).abs() / scale).max().item() < 2e-6
    assert torch.equal(two, again) and not torch.equal(two, one)


def test_set_option_knows_dense_two():
    L = native.lib()
    assert L.sage_set_option(b"dense_two", 7) != 0 and L.sage_set_option(b"dense_two", 1) == 0 and L.sage_set_option(b"dense_two", -1) == 0


def test_concat_contraction_finite_and_non_finite(two_on):
    r2.test_bf16x3_concat_contraction_finite_and_non_finite(True, 256)


@pytest.mark.parametrize("roles,depth", [("SGDL", 4), ("SGDD", 3), ("SSSS", 2)])
def test_role_pipeline_is_bit_identical_to_single_forwards(two_on, roles, depth):
    """The self chunk's launch runs on stream D beside the gather (SAGE_STAGE_CONTRACT1_SELF); a single forward makes the same two launches
    one after the other: the same bits."""
    r2.test_role_pipeline_is_bit_identical_to_single_forwards(True, False, roles, depth)


def test_role_pipeline_with_host_threads(two_on):
    """Role D waits for S, launches the self chunk, waits for G, launches the means' chunk -- from its own host thread."""
    graph, table, w1, w2 = r2._problem(d0=256, h1=128, concat=True)
    rowptr, col = graph.to(DEV)
    cand = np.nonzero(graph.degrees() > 0)[0]
    rs = np.random.default_rng(4)
    nb, b = 24, 1024
    seeds = torch.from_numpy(np.stack([rs.choice(cand, b, replace=False) for _ in range(nb)]).astype(np.int32)).to(DEV)
    keys = [5 + i for i in range(nb)]
    tdev, w1d, w2d = table.to(DEV), w1.to(DEV), w2.to(DEV)
    pipe = RolePipeline(rowptr, col, tdev, w1d, w2d, 15, 25, batch=b, depth=4, concat=True, threads=True)
    eng = TwoHopEngine(rowptr, col, tdev, w1d, w2d, 15, 25, concat=True, max_batch=b)
    out = torch.empty(nb, b, w2.shape[0], device=DEV)
    torch.cuda.synchronize()
    for i in range(nb):
        pipe.submit(seeds[i], keys[i], out[i])
    pipe.synchronize()
    for i in range(nb):
        assert torch.equal(out[i], eng.forward(seeds[i], seed=keys[i])), f"batch {i}"


def test_concat_forward_against_the_oracle_at_config3_size(two_on):
    """R-MAT 2^20 / 16 M edges, D0 = 256 (the 512-deep layer), B = 4096, fanout 15 / 25, concat encoder: sampled sets bit-exact against
    oracle/sampler_ref.c, values within 1e-5 of the row maximum of the fp64 oracle on those sets."""
    from sage355.graph import rmat_graph
    graph = rmat_graph(20, 16_000_000, seed=0, cache_dir=r2.CACHE)
    r2._full_size_check(graph, 256, 15, 25, True, False)
