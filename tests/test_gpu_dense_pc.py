"""The producer / consumer form of the split-bf16 contraction (csrc/sage_dense.hip: dense_pc_kernel; sage_set_option("dense_pc", 1)).
Not the default (it is faster alone and slower inside the role pipeline), so the other GPU tests run the lock-step
kernel; this file runs the contraction's own tests again with the option on -- values against the fp64 oracle / torch, the Inf / NaN
classes of torch.mm (encoders.py:58-61), the pipeline against single forwards -- and compares the two kernels with each other."""
import numpy as np
import pytest
import torch

import test_gpu_round2 as r2
from sage355 import native
from sage355.engine import RolePipeline, TwoHopEngine

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture
def pc_on():
    native.check(native.lib().sage_set_option(b"dense_pc", 1), "set_option")
    yield
    native.check(native.lib().sage_set_option(b"dense_pc", -1), "set_option")


def test_set_option_checks_its_arguments():
    L = native.lib()
    assert L.sage_set_option(b"no_such_option", 1) != 0 and b"unknown option" in L.sage_last_error()
    assert L.sage_set_option(b"dense_pc", 7) != 0
    assert L.sage_set_option(None, 1) != 0
    assert L.sage_set_option(b"dense_pc", -1) == 0


@pytest.mark.parametrize("concat", [False, True])
@pytest.mark.parametrize("d0,h1", [(256, 128), (128, 64), (64, 128), (100, 52)])
def test_producer_consumer_kernel_against_the_lock_step_kernel(d0, h1, concat):
    """Same six bf16 products per k-step in the same order, one accumulator per K half added at the end exactly as the lock-step
    kernel's two wave groups are: bit for bit the same 2-hop output.  The 512-deep concat layer (dim = 256) is the one shape whose
    sums are ASSOCIATED differently: two launches, (self h0 + self h1) + (agg h0 + agg h1), against two passes,
    (self h0 + agg h0) + (self h1 + agg h1) -- equal to a few fp32 roundings of the row maximum."""
    graph, table, w1, w2 = r2._problem(d0=d0, h1=h1, concat=concat)
    rowptr, col = graph.to(DEV)
    seeds = torch.from_numpy(np.random.default_rng(3).choice(np.nonzero(graph.degrees() > 0)[0], 2048, replace=False).astype(np.int32)).to(DEV)
    eng = TwoHopEngine(rowptr, col, table.to(DEV), w1.to(DEV), w2.to(DEV), 15, 25, concat=concat, max_batch=2048)
    assert bool(eng.layout.layer1_split) and eng._model().w1_prepared is not None
    lock = eng.forward(seeds, seed=9).clone()
    native.check(native.lib().sage_set_option(b"dense_pc", 1), "set_option")
    try:
        pc = eng.forward(seeds, seed=9).clone()
    finally:
        native.lib().sage_set_option(b"dense_pc", -1)
    if concat and d0 == 256:
        scale = lock.abs().amax(1, keepdim=True).clamp_min(1e-30)
        assert ((pc - lock).abs() / scale).max().item() < 2e-6
        assert not torch.equal(pc, lock)            # (if this ever holds, the docstring above is out of date)
    else:
        assert torch.equal(pc, lock)


def test_adversarial_finite_data(pc_on):
    r2.test_bf16x3_contraction_on_adversarial_finite_data()


def test_inf_and_nan_like_torch_mm(pc_on):
    r2.test_bf16x3_contraction_propagates_inf_and_nan_like_torch_mm(True)


@pytest.mark.parametrize("d0", [256, 128])
def test_concat_contraction_finite_and_non_finite(pc_on, d0):
    r2.test_bf16x3_concat_contraction_finite_and_non_finite(True, d0)


@pytest.mark.parametrize("concat,self_loop", [(False, False), (True, False), (False, True)])
def test_role_pipeline_is_bit_identical_to_single_forwards(pc_on, concat, self_loop):
    """Concat with dim = 256: the self chunk's launch runs on stream D beside the gather (SAGE_STAGE_CONTRACT1_SELF)."""
    r2.test_role_pipeline_is_bit_identical_to_single_forwards(concat, self_loop, "SGDL", 4)
    r2.test_role_pipeline_is_bit_identical_to_single_forwards(concat, self_loop, "SGDD", 3)


def test_role_pipeline_with_host_threads_and_the_two_launch_contraction(pc_on):
    """dim = 256, concat: role D waits for S, launches the self chunk, waits for G, launches the means' chunk -- from its own host thread."""
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


def test_config5_and_full_size_oracle_checks(pc_on):
    r2.test_config5_products_shaped_fanout_20_25(True, False)
    r2.test_full_size_properties_homogeneity_permutation_sub_batch("degree")
