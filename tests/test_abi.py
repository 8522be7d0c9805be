"""CPU checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and
exports every symbol include/sage355.h declares; the host mirror keeps the reference's
constructor / attribute surface.  No compute calls (no GPU here)."""
import inspect
import os
import re

import pytest
import torch

from sage355 import native

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(REPO, "include", "sage355.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sage_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    if not os.path.exists(native.LIB_PATH):
        native.build()
    L = native.lib()
    names = header_symbols()
    assert len(names) >= 12
    for name in names:
        assert hasattr(L, name), f"{name} declared in include/sage355.h but not exported"
    assert sorted(native.SYMBOLS) == names, "native.SYMBOLS out of sync with the header"
    assert L.sage_abi_version() == native.ABI_VERSION
    assert L.sage_build_arch() == b"gfx950"


def test_code_object_is_gfx950():
    """Every device code object bundled in the library targets gfx950 and nothing else (no second arch, no PTX).  The check reads
    the offload bundle's target triples; a bare substring search would trip over rocPRIM's HOST-side table of architecture names
    (the radix sort of the reproducible mean backward, csrc/sage_backward_det.hip, carries it), which is data, not code."""
    import re
    blob = open(native.LIB_PATH, "rb").read()
    targets = set(re.findall(rb"hip[v0-9]*-amdgcn-amd-amdhsa--([a-z0-9]+)", blob))
    assert targets == {b"gfx950"}, targets
    assert b"nvptx" not in blob and b".target sm_" not in blob


def test_host_argument_validation_rejects_bad_calls_before_any_launch():
    L = native.lib()
    # NULL arrays / bad fanout are refused on the host: nothing is launched, so this is safe without a GPU
    rc = L.sage_sample_neighbors(None, None, 10, None, 4, None, 5, 0, 1, None, None, None, None, 0, None, None, None)
    assert rc == -1 and b"NULL" in L.sage_last_error()
    rc = L.sage_gather_mean(None, 1, 1, 1, None, None, 1, 1, None, None, None, None, None, 1, None)
    assert rc == -1
    m = native.Model()
    lay = native.WsLayout()
    assert L.sage_forward2_layout(m, 16, lay) == -1          # zeroed model: num_nodes = 0
    m.num_nodes, m.d0, m.h1, m.h2, m.k1, m.k2, m.table_ld = 100, 8, 4, 4, 3, 99, 8
    assert L.sage_forward2_layout(m, 16, lay) == -1 and b"fanout" in L.sage_last_error()
    m.k2 = 5
    assert L.sage_forward2_layout(m, 16, lay) == 0
    assert lay.max_s1 == 16 * 5 + 16 and lay.hash_capacity >= 2 * 16 * 6 and lay.total_bytes > 0
    assert lay.hash_capacity & (lay.hash_capacity - 1) == 0
    # ABI 3 (round 3): the reproducible backward entry points validate on the host too, and their size queries are host arithmetic
    assert L.sage_linear_act_backward_ws(None, 0, None, None, 4, 4, None, 4, 4, 0, None, 4, None, 4, 8, None, None, 4, None, 0, None, None, 0, None) == -1
    assert L.sage_two_hop_grad_w1(None, 4, None, None, 5, None, 8, None, 4, 4, 0, None, 4, 4, 0, None, 4, None, None, 4, None, 0, None) == -1
    assert b"NULL" in L.sage_last_error()
    assert L.sage_gather_mean_backward_ws(None, 4, 4, None, None, 3, 8, None, None, None, None, 10, None, 4, None, 0, None) == -1
    assert L.sage_row_order(None, 8, None, 0, None, None, 0, None) == -1
    assert L.sage_pipe_reset(None) == -1
    need = L.sage_linear_act_backward_workspace_bytes(23_000, 256, 0, 128)
    assert need >= 128 * 256 * 4 and need % 4 == 0                      # at least one partial tile
    assert L.sage_two_hop_grad_w1_workspace_bytes(4096, 25, 256, 0, 128) >= 128 * 256 * 4
    assert L.sage_two_hop_grad_w1_workspace_bytes(4096, 25, 256, 1, 128) >= 2 * L.sage_two_hop_grad_w1_workspace_bytes(4096, 25, 256, 0, 128) - 64


def test_module_surface_matches_reference_signatures():
    """encoders.py:12-16, aggregators.py:16,34 -- same positional order and defaults."""
    from sage355.aggregators import MeanAggregator
    from sage355.encoders import Encoder
    a = inspect.signature(MeanAggregator.__init__)
    assert list(a.parameters)[1:] == ["features", "initializer", "cuda", "gcn", "feature_dim", "num_nodes"]
    assert a.parameters["initializer"].default == "None" and a.parameters["feature_dim"].default == 100
    f = inspect.signature(MeanAggregator.forward)
    assert list(f.parameters)[1:] == ["nodes", "to_neighs", "num_sample", "initializer"]
    assert f.parameters["num_sample"].default == 10
    e = inspect.signature(Encoder.__init__)
    assert list(e.parameters)[1:12] == ["features", "feature_dim", "embed_dim", "adj_lists", "aggregator", "num_sample",
                                        "initializer", "base_model", "gcn", "cuda", "feature_transform"]
    assert e.parameters["num_sample"].default == 10 and e.parameters["gcn"].default is False


def test_model_py_wiring_and_state_dict_names(capsys):
    """model.py:214-224 builds the stack like this; state-dict names must match the reference's
    (weight, base_model.weight, base_model.features.weight ...)."""
    from graphsage.aggregators import MeanAggregator   # the drop-in shim package
    from graphsage.encoders import Encoder
    features = torch.nn.Embedding(20, 6)
    features.weight = torch.nn.Parameter(torch.randn(20, 6), requires_grad=False)
    adj = {i: {(i + 1) % 20, (i + 7) % 20} for i in range(20)}
    agg1 = MeanAggregator(features, cuda=True, feature_dim=6, num_nodes=20, initializer="None")
    enc1 = Encoder(features, 6, 5, adj, agg1, gcn=True, cuda=False, initializer="None")
    agg2 = MeanAggregator(lambda nodes: enc1(nodes).t(), 20, cuda=False)       # num_nodes lands in `initializer`
    enc2 = Encoder(lambda nodes: enc1(nodes).t(), enc1.embed_dim, 8, adj, agg2, base_model=enc1, gcn=True, cuda=False)
    enc1.num_samples = 5                                                         # model.py:223-224 (a no-op there too)
    assert "feat dim: 6 embed_dim: 5" in capsys.readouterr().out                # encoders.py:38
    assert agg1.cuda is False                                                    # encoders.py:30 overwrote it
    assert enc2.embed_dim == 8 and enc2.num_sample == 10 and enc1.weight.shape == (5, 6)
    assert Encoder(features, 6, 5, adj, MeanAggregator(features), gcn=False).weight.shape == (5, 12)
    names = dict(enc2.named_parameters())
    assert set(names) >= {"weight", "base_model.weight", "base_model.features.weight"}
    assert enc2._can_fuse_two_hop()
    if not torch.cuda.is_available():
        with pytest.raises(native.SageError):
            enc2([0, 1, 2])      # no GPU in this container: must fail loudly, never fall back
