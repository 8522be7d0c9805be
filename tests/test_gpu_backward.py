"""GPU parity of the backward operators (sage_linear_act_backward, sage_gather_mean_backward) and of
the modules' autograd path against (a) torch autograd in fp64 on the CPU and (b) the reference's own
weight gradients stored in tests/golden/*.npz (model.py:249 loss.backward() through the reference stack)."""
import numpy as np
import pytest
import torch

from sage355 import autograd, ops
from test_gpu_forward import build_modules
from util import TWO_LAYER_CASES, assert_close_rowmax, load_golden, sets_from_padded

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("n,dim,h,concat,act", [(700, 256, 128, False, "relu"), (300, 100, 50, True, "relu"),
                                                 (129, 50, 7, False, "sigmoid"), (1000, 64, 128, True, "none"), (5000, 128, 128, False, "relu")])
def test_linear_act_backward_matches_autograd(n, dim, h, concat, act):
    gen = torch.Generator().manual_seed(n)
    agg = torch.randn(n, dim, generator=gen)
    w = torch.randn(h, dim * (2 if concat else 1), generator=gen) / np.sqrt(dim)
    self_tab = torch.randn(n + 50, dim, generator=gen) if concat else None
    self_index = torch.randperm(n + 50, generator=gen)[:n].to(torch.int32) if concat else None
    cot = torch.randn(n, h, generator=gen)
    code = {"relu": ops.ACT_RELU, "sigmoid": ops.ACT_SIGMOID, "none": ops.ACT_NONE}[act]
    # fp64 autograd reference
    a64 = agg.double().requires_grad_()
    w64 = w.double().requires_grad_()
    s64 = self_tab.double().requires_grad_() if concat else None
    x = torch.cat([s64[self_index.long()], a64], 1) if concat else a64
    pre = x.mm(w64.t())
    y = torch.relu(pre) if act == "relu" else torch.sigmoid(pre) if act == "sigmoid" else pre
    (y * cot.double()).sum().backward()
    # HIP
    ad = agg.to(DEV).requires_grad_()
    wd = w.to(DEV).requires_grad_()
    sd = self_tab.to(DEV).requires_grad_() if concat else None
    out = autograd.linear_act(ad, wd, code, sd, self_index.to(DEV) if concat else None)
    (out * cot.to(DEV)).sum().backward()
    assert rel_err(wd.grad, w64.grad) < 2e-5
    assert rel_err(ad.grad, a64.grad) < 2e-5
    if concat:
        assert rel_err(sd.grad, s64.grad) < 2e-5


def test_gather_mean_backward_matches_autograd():
    rs = np.random.default_rng(0)
    rows, dim, k, n = 400, 96, 9, 300
    table = torch.randn(rows, dim, generator=torch.Generator().manual_seed(1))
    cnt = rs.integers(0, k + 1, size=n).astype(np.int32)
    nbr = rs.integers(0, rows, size=(n, k)).astype(np.int32)
    cot = torch.randn(n, dim, generator=torch.Generator().manual_seed(2))
    t64 = table.double().requires_grad_()
    mask = torch.from_numpy(np.arange(k)[None, :] < cnt[:, None])
    gathered = t64[torch.from_numpy(nbr).long()] * mask.unsqueeze(-1)
    mean = gathered.sum(1) / torch.from_numpy(cnt).clamp(min=1).unsqueeze(1)
    (mean * cot.double()).sum().backward()
    td = table.to(DEV).requires_grad_()
    out = autograd.gather_mean(td, torch.from_numpy(nbr).to(DEV), torch.from_numpy(cnt).to(DEV))
    (out * cot.to(DEV)).sum().backward()
    assert rel_err(td.grad, t64.grad) < 2e-5


@pytest.mark.parametrize("name", TWO_LAYER_CASES)
def test_module_weight_gradients_match_reference_golden(name):
    """grad of (enc2(seeds) * cotangent).sum() w.r.t. enc2.weight and enc1.weight, as the reference's
    autograd produced them on the same injected neighbour sets."""
    g = load_golden(name)
    sets1 = sets_from_padded(g["layer1_nodes"], g["nbr1"], g["cnt1"])
    sets2 = sets_from_padded(g["seeds"], g["nbr2"], g["cnt2"])
    enc1, enc2 = build_modules(g, None, None, False, sets1, sets2)
    out = enc2([int(s) for s in g["seeds"]])
    assert out.requires_grad and not out.is_cuda
    assert_close_rowmax(out.detach(), g["enc2_out"], rows_dim=1, what="forward in grad mode")
    (out * torch.from_numpy(g["cotangent"])).sum().backward()
    assert enc2.weight.grad is not None and enc1.weight.grad is not None
    for got, want, what in ((enc2.weight.grad, g["grad_w2"], "grad_w2"), (enc1.weight.grad, g["grad_w1"], "grad_w1")):
        err = rel_err(got, torch.from_numpy(want))
        assert err < 5e-5, f"{name} {what}: {err:.2e}"
