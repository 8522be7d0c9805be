"""Tensor-level wrappers of the C ABI operators (include/sage355.h).

Each function checks device / dtype / contiguity on the host -- a kernel that
walks a bad pointer can take the whole GPU node down -- then passes raw device
pointers to libsage355.  All work is enqueued on torch's current stream.
"""
import torch

from . import native
from .native import ACT_NONE, ACT_RELU, ACT_SIGMOID, TAG_INNER, TAG_INNER_SELF, TAG_OUTER  # noqa: F401


def _need_gpu():
    if not torch.cuda.is_available():
        raise native.SageError("sage355 needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")


def _chk(t, dtype, name, dims=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise native.SageError(f"{name}: expected a device tensor")
    if t.dtype != dtype:
        raise native.SageError(f"{name}: dtype {t.dtype}, expected {dtype}")
    if not t.is_contiguous():
        raise native.SageError(f"{name}: must be contiguous")
    if dims is not None and t.dim() != dims:
        raise native.SageError(f"{name}: {t.dim()}-d, expected {dims}-d")
    return t


def _row_major(t, name):
    """2-d fp32, unit inner stride; returns (tensor, leading dimension)."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2:
        raise native.SageError(f"{name}: expected a 2-d fp32 device tensor")
    if t.stride(1) != 1 and t.shape[1] > 1:
        raise native.SageError(f"{name}: inner stride must be 1")
    return t, (t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1))


def check_id_range(ids64, num_nodes, what="node ids"):
    """Host-side range check of ids that arrived from the host (numpy int64 array).  The reference fails safely for a
    bad id (IndexError from the nn.Embedding lookup, encoders.py:50 / aggregators.py:65); a device kernel would read
    rowptr[] out of bounds, so the check happens here, BEFORE the int32 cast can wrap a large id into range."""
    if num_nodes is None or ids64.size == 0:
        return
    lo, hi = int(ids64.min()), int(ids64.max())
    if lo < 0 or hi >= int(num_nodes):
        raise native.SageError(f"{what}: id {lo if lo < 0 else hi} outside [0, {int(num_nodes)})")


def as_ids(nodes, device, num_nodes=None):
    """list / numpy / tensor of node ids -> int32 device tensor (encoders.py:40-47 accepts all three).
    num_nodes: ids that come from the HOST are range-checked against it (SageError); device-resident ids are not
    (that would be a host sync per call) -- the sampler kernels treat an out-of-range id as an isolated node instead."""
    import numpy as np
    if isinstance(nodes, torch.Tensor):
        if not nodes.is_cuda:
            check_id_range(nodes.detach().to(torch.int64).numpy(), num_nodes)
        return nodes.to(device=device, dtype=torch.int32).contiguous()
    ids64 = np.asarray(nodes, dtype=np.int64)
    check_id_range(ids64, num_nodes)
    return torch.from_numpy(np.ascontiguousarray(ids64.astype(np.int32))).to(device)


def next_pow2(x):
    p = 4
    while p < x:
        p <<= 1
    return p


class Frontier:
    """Device hash set of distinct ids + id -> row map (aggregators.py:52-53)."""

    def __init__(self, max_ids, device, first_row=0):
        _need_gpu()
        self.capacity = next_pow2(2 * max(int(max_ids), 1))
        self.max_nodes = int(max_ids) + int(first_row)
        self.keys = torch.empty(self.capacity, dtype=torch.int32, device=device)
        self.rows = torch.empty(self.capacity, dtype=torch.int32, device=device)
        self.nodes = torch.empty(self.max_nodes, dtype=torch.int32, device=device)
        self.count = torch.zeros(1, dtype=torch.int32, device=device)
        self.c = native.Frontier(self.keys.data_ptr(), self.rows.data_ptr(), self.capacity, self.nodes.data_ptr(),
                                 self.count.data_ptr(), self.max_nodes)
        self.reset(first_row)

    def reset(self, first_row=0):
        native.check(native.lib().sage_frontier_reset(self.c, int(first_row), native.stream_handle()), "frontier_reset")

    def size(self):
        return int(self.count.item())   # host sync: tests / generic path only

    def node_list(self):
        return self.nodes[: self.size()]


def sample_neighbors(rowptr, col, nodes, k, seed, tag=TAG_OUTER, n_dev=None, frontier=None, insert_self=False,
                     any_nonempty=None, out_nbr=None, out_cnt=None):
    """encoders.py:47 + aggregators.py:42-48 (+52-53 with a frontier).
    -> (nbr int32 [n,k], cnt int32 [n], nbr_slot or None, self_slot or None)."""
    _need_gpu()
    _chk(rowptr, torch.int64, "rowptr", 1)
    _chk(col, torch.int32, "col", 1)
    _chk(nodes, torch.int32, "nodes", 1)
    n = nodes.shape[0]
    dev = nodes.device
    nbr = torch.empty((n, k), dtype=torch.int32, device=dev) if out_nbr is None else _chk(out_nbr, torch.int32, "out_nbr")
    cnt = torch.empty(n, dtype=torch.int32, device=dev) if out_cnt is None else _chk(out_cnt, torch.int32, "out_cnt")
    if nbr.numel() < n * k or cnt.numel() < n:
        raise native.SageError("sample_neighbors: output buffers too small")
    nbr_slot = torch.empty((n, k), dtype=torch.int32, device=dev) if frontier is not None else None
    self_slot = torch.empty(n, dtype=torch.int32, device=dev) if (frontier is not None and insert_self) else None
    rc = native.lib().sage_sample_neighbors(
        native.ptr(rowptr), native.ptr(col), rowptr.shape[0] - 1, native.ptr(nodes), n, native.ptr(n_dev), int(k),
        int(seed) & 0xFFFFFFFFFFFFFFFF, int(tag), native.ptr(nbr), native.ptr(cnt), native.ptr(any_nonempty),
        frontier.c if frontier is not None else None, 1 if insert_self else 0, native.ptr(nbr_slot),
        native.ptr(self_slot), native.stream_handle())
    native.check(rc, "sample_neighbors")
    return nbr, cnt, nbr_slot, self_slot


def frontier_insert(nbr, cnt, frontier, self_nodes=None, n_dev=None):
    _need_gpu()
    _chk(nbr, torch.int32, "nbr", 2)
    _chk(cnt, torch.int32, "cnt", 1)
    n, k = nbr.shape
    nbr_slot = torch.empty_like(nbr)
    self_slot = torch.empty(n, dtype=torch.int32, device=nbr.device) if self_nodes is not None else None
    rc = native.lib().sage_frontier_insert(native.ptr(nbr), native.ptr(cnt), k, native.ptr(self_nodes), n,
                                           native.ptr(n_dev), frontier.c, native.ptr(nbr_slot), native.ptr(self_slot),
                                           native.stream_handle())
    native.check(rc, "frontier_insert")
    return nbr_slot, self_slot


def gather_mean(table, nbr, cnt, slot_rows=None, self_row=None, any_nonempty=None, n_dev=None, out=None):
    """aggregators.py:54-74 -> [n, dim] mean of the gathered rows."""
    _need_gpu()
    table, ld = _row_major(table, "table")
    _chk(nbr, torch.int32, "nbr", 2)
    _chk(cnt, torch.int32, "cnt", 1)
    n, k = nbr.shape
    dim = table.shape[1]
    if out is None:
        out = torch.empty((n, dim), dtype=torch.float32, device=table.device)
    out, ldo = _row_major(out, "out")
    rc = native.lib().sage_gather_mean(native.ptr(table), table.shape[0], ld, dim, native.ptr(nbr), native.ptr(cnt), k, n,
                                       native.ptr(n_dev), native.ptr(slot_rows), native.ptr(self_row),
                                       native.ptr(any_nonempty), native.ptr(out), ldo, native.stream_handle())
    native.check(rc, "gather_mean")
    return out


def linear_act(agg, weight, act=ACT_RELU, self_tab=None, self_index=None, n_dev=None, out=None):
    """encoders.py:49-62 -> [n, out_dim] (the module hands out the transpose view)."""
    _need_gpu()
    agg, ld_agg = _row_major(agg, "agg")
    weight, ldw = _row_major(weight, "weight")
    n, dim = agg.shape
    out_dim = weight.shape[0]
    ld_self = 0
    if self_tab is not None:
        self_tab, ld_self = _row_major(self_tab, "self_tab")
        if self_tab.shape[1] != dim:
            raise native.SageError("linear_act: self_tab width != agg width")
    if weight.shape[1] != dim * (2 if self_tab is not None else 1):
        raise native.SageError(f"linear_act: weight is {tuple(weight.shape)}, inputs are {dim} wide")
    if out is None:
        out = torch.empty((n, out_dim), dtype=torch.float32, device=agg.device)
    out, ldo = _row_major(out, "out")
    rc = native.lib().sage_linear_act(native.ptr(self_tab), ld_self, native.ptr(self_index), native.ptr(agg), ld_agg, dim,
                                      native.ptr(weight), ldw, out_dim, int(act), n, native.ptr(n_dev), native.ptr(out),
                                      ldo, native.stream_handle())
    native.check(rc, "linear_act")
    return out


def layer_forward_supported(dim, out_dim, concat):
    return bool(native.lib().sage_layer_forward_supported(int(dim), int(out_dim), 1 if concat else 0))


def layer_forward(table, nbr, cnt, weight, act=ACT_RELU, concat=False, self_index=None, slot_rows=None, self_row=None,
                  any_nonempty=None, n_dev=None, out=None):
    """One Encoder.forward (encoders.py:47-62) in one launch."""
    _need_gpu()
    table, ld = _row_major(table, "table")
    weight, ldw = _row_major(weight, "weight")
    _chk(nbr, torch.int32, "nbr", 2)
    _chk(cnt, torch.int32, "cnt", 1)
    n, k = nbr.shape
    dim = table.shape[1]
    out_dim = weight.shape[0]
    if weight.shape[1] != dim * (2 if concat else 1):
        raise native.SageError(f"layer_forward: weight is {tuple(weight.shape)}, table is {dim} wide")
    if out is None:
        out = torch.empty((n, out_dim), dtype=torch.float32, device=table.device)
    out, ldo = _row_major(out, "out")
    rc = native.lib().sage_layer_forward(native.ptr(table), table.shape[0], ld, dim, native.ptr(nbr), native.ptr(cnt), k, n,
                                         native.ptr(n_dev), native.ptr(slot_rows), native.ptr(self_row),
                                         native.ptr(any_nonempty), 1 if concat else 0, native.ptr(self_index),
                                         native.ptr(weight), ldw, out_dim, int(act), native.ptr(out), ldo,
                                         native.stream_handle())
    native.check(rc, "layer_forward")
    return out
