"""Encoder -- host-side mirror of graphsage/encoders.py:8-62.

Same constructor, attributes, parameter name (``weight``) and ``forward(nodes)
-> [embed_dim, len(nodes)]`` as the reference class; ``model.py`` can build its
two-layer stack from it unchanged (model.py:218-222).  Underneath:

* the adjacency dict-of-sets is converted ONCE to CSR in HBM (graph.py);
* ``features`` being an ``nn.Embedding`` means "raw feature table": its weight
  is kept resident in HBM;
* ``features`` being ``lambda nodes: enc1(nodes).t()`` with ``base_model=enc1``
  (exactly how model.py:220-222 wires layer 2) means "the layer below": the
  whole 2-hop forward then runs as one C call (engine.TwoHopEngine ->
  sage_forward2) -- sample, frontier dedupe, sample, layer 1, layer 2 -- with
  no Python in between.  This assumes ``features(ids) == base_model(ids).t()``;
  pass ``fuse_base_model=False`` for a base model wired differently.
* anything else is treated as an opaque feature function: the frontier ids are
  handed to it (as the reference does, aggregators.py:62-65) and the returned
  rows are aggregated on the GPU.

Adjacency freeze semantics.  The reference re-reads ``adj_lists`` on every call
(encoders.py:47); here the device CSR of an adjacency object is CACHED.  It is
revalidated (a) on every call by the number of keys and by the set sizes of up
to 64 of the batch's own nodes against the cached degrees, (b) every 64th call
by a fingerprint over all set sizes and the contents of ~256 rows (adjacencies
up to 200 000 keys), (c) on every call when ``SAGE_ADJ_STRICT=1``.  An in-place
edit that keeps every probed size can therefore be served from the stale CSR
for up to 63 forwards: call ``invalidate_adjacency(adj_lists)`` after editing an
adjacency an Encoder has already seen.  The feature table is treated as frozen
too (model.py:214-215): in-place writes that bump the tensor's version counter
are picked up, ``.data`` writes need ``TwoHopEngine.refresh_table()``.

Sampling: the device sampler draws k distinct uniform neighbours per node
(all of them when deg < k), the reference's rule (aggregators.py:42-46), from a
counter-based generator.  The two device-sampler paths (``_forward_two_hop``,
``_forward_table``) take ONE 64-bit value from Python's global ``random`` per
``forward`` as the key, so ``random.seed(s)`` (model.py:193) still makes a run
reproducible, but (a) the sets differ from the reference's for the same seed and
(b) the global stream is advanced by one ``getrandbits(64)`` per call instead of
one ``random.sample`` per node -- a program that interleaves its own draws from
``random`` sees different values than under the reference.  Only the strict path
(``MeanAggregator.forward`` / ``_forward_generic``) consumes the stream call for
call as the reference does.
"""
import collections
import random

import numpy as np
import torch
import torch.nn as nn
from torch.nn import init

from . import autograd, native, ops
from .aggregators import MeanAggregator
from .engine import TwoHopEngine
from .graph import csr_from_adj_lists

SIGMOID_INITIALIZERS = ("node_degree", "shared", "pagerank")   # encoders.py:58

_CSR_CACHE_MAX = 8            # adjacency objects kept (LRU): both layers of a model share one, a process holds a few models
_CSR_CHECK_MAX_NODES = 200_000  # adjacencies up to this many keys are re-fingerprinted (periodically, below); larger ones are frozen
_csr_cache = collections.OrderedDict()


_CSR_RECHECK_EVERY = 64       # forwards between two full fingerprints of a cached adjacency


def _fingerprint(adj_lists):
    """(non-empty sets, total degree, content hash of a sample of rows): detects an adjacency mutated in place after first
    use, including an edge MOVED between two sampled rows.  Empty sets that the reference's defaultdict inserts on a miss
    (encoders.py:47) do not change it.  O(N) in Python, so it is recomputed only every _CSR_RECHECK_EVERY-th call per
    adjacency, and never for a large one, which is FROZEN at first use: call invalidate_adjacency(adj_lists) after editing
    an adjacency that a sage355 Encoder has already seen."""
    if len(adj_lists) > _CSR_CHECK_MAX_NODES:
        return None
    nonempty = total = 0
    for s in adj_lists.values():
        n = len(s)
        total += n
        nonempty += n > 0
    stride, i, h = max(1, nonempty // 256), 0, 0
    for k, s in adj_lists.items():               # ~256 of the NON-EMPTY rows, evenly spaced (empty ones come and go, see above)
        if s:
            if i % stride == 0:
                h = (h * 1000003 + hash(k) + 31 * sum(s)) & 0xFFFFFFFFFFFF
            i += 1
    return nonempty, total, h


def invalidate_adjacency(adj_lists=None):
    """Drop the cached device CSR of `adj_lists` (all of them when None)."""
    if adj_lists is None:
        _csr_cache.clear()
    else:
        _csr_cache.pop(id(adj_lists), None)


def _probe_ok(adj_lists, deg_host, probe_nodes):
    """Cheap per-call signal: the set sizes of up to 64 of the batch's own nodes still equal the cached CSR's degrees."""
    if probe_nodes is None:
        return True
    n = len(deg_host)
    step = max(1, len(probe_nodes) // 64)
    for i in range(0, len(probe_nodes), step):
        v = int(probe_nodes[i])
        if 0 <= v < n:
            s = adj_lists.get(v) if hasattr(adj_lists, "get") else adj_lists[v]      # .get: never insert (a defaultdict would)
            if (len(s) if s is not None else 0) != deg_host[v]:
                return False
    return True


def _device_csr(adj_lists, num_nodes_hint, device, probe_nodes=None):
    """dict-of-sets -> CSR in HBM, cached per adjacency object (both layers share one, model.py:219-222).
    The reference re-reads the dict on every call; here the conversion is cached and revalidated: every call by the key count and
    by the set sizes of a few of the batch's own nodes (`probe_nodes`), every _CSR_RECHECK_EVERY-th call (or every call under
    SAGE_ADJ_STRICT=1) by a fingerprint."""
    import os
    key = id(adj_lists)
    hit = _csr_cache.get(key)
    if hit is not None and hit[0] is adj_lists and hit[4] == num_nodes_hint:
        # every call: O(1) + O(64) -- a changed key count (new nodes; also the empty sets the reference's defaultdict inserts on a miss)
        # or a changed set size among the batch's own nodes triggers the full fingerprint at once; otherwise it is recomputed every
        # _CSR_RECHECK_EVERY-th call (it walks every set in Python: ~0.3 ms at Cora's size, twice per forward, was a third of a
        # 256-seed training step)
        state = hit[5]
        state[0] += 1
        strict = os.environ.get("SAGE_ADJ_STRICT", "0") == "1"
        probe = _probe_ok(adj_lists, hit[6], probe_nodes)
        if probe and not strict and len(adj_lists) == state[1] and state[0] % _CSR_RECHECK_EVERY != 0:
            _csr_cache.move_to_end(key)
            return hit[1], hit[2]
        if probe and hit[3] is not None and hit[3] == _fingerprint(adj_lists):
            state[1] = len(adj_lists)
            _csr_cache.move_to_end(key)
            return hit[1], hit[2]
        if probe and hit[3] is None and not strict:          # too large to fingerprint: frozen at first use (probe and key count aside)
            state[1] = len(adj_lists)
            _csr_cache.move_to_end(key)
            return hit[1], hit[2]
    fp = _fingerprint(adj_lists)
    g = csr_from_adj_lists(adj_lists, None)
    if num_nodes_hint and g.num_nodes < num_nodes_hint:
        pad = np.full(num_nodes_hint - g.num_nodes, g.rowptr[-1], dtype=np.int64)
        g.rowptr = np.concatenate([g.rowptr, pad])
        g.num_nodes = num_nodes_hint
    rowptr, col = g.to(device)
    if col.numel() == 0:
        col = torch.zeros(1, dtype=torch.int32, device=device)
    deg_host = np.diff(g.rowptr)
    _csr_cache[key] = (adj_lists, rowptr, col, fp, num_nodes_hint, [0, len(adj_lists)], deg_host)
    _csr_cache.move_to_end(key)
    while len(_csr_cache) > _CSR_CACHE_MAX:
        _csr_cache.popitem(last=False)
    return rowptr, col


_threads_capped = False


def _cap_host_threads_once():
    """Strict drop-in mode (cuda=False): the caller's classifier, loss and SGD run in torch on the HOST (model.py:237-250).  On a box
    whose cgroup share is smaller than its core count torch starts one intra-op thread per CORE, and three 70 k-element `add_`s of the
    optimizer then take 2 ms each on the oversubscribed share (a 256-seed step: mean 3-10 ms against a median of 0.7-1.2).  Once, at
    the first such forward, torch's intra-op pool is capped to the cores this process may actually use.  SAGE_KEEP_TORCH_THREADS=1 opts out."""
    global _threads_capped
    if _threads_capped:
        return
    _threads_capped = True
    import os
    if os.environ.get("SAGE_KEEP_TORCH_THREADS", "0") == "1":
        return
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            usable = max(1, min(usable, int(int(quota) / int(period))))
    except Exception:
        pass
    if torch.get_num_threads() > usable:
        import sys
        print(f"sage355: torch intra-op threads {torch.get_num_threads()} -> {usable} (the cores this process may use; "
              "SAGE_KEEP_TORCH_THREADS=1 keeps torch's own choice)", file=sys.stderr)
        torch.set_num_threads(usable)


class Encoder(nn.Module):
    """Encodes a node using the 'convolutional' GraphSage approach"""

    def __init__(self, features, feature_dim, embed_dim, adj_lists, aggregator, num_sample=10, initializer="None",
                 base_model=None, gcn=False, cuda=False, feature_transform=False, fuse_base_model=True):
        super(Encoder, self).__init__()
        self.features = features
        self.feat_dim = feature_dim
        self.adj_lists = adj_lists
        self.aggregator = aggregator
        self.num_sample = num_sample
        if base_model != None:   # noqa: E711  (encoders.py:24)
            self.base_model = base_model
        self.gcn = gcn
        self.embed_dim = embed_dim
        self.cuda = cuda                       # shadows nn.Module.cuda, as encoders.py:29 does
        self.aggregator.cuda = cuda            # encoders.py:30
        self.weight = nn.Parameter(torch.FloatTensor(embed_dim, self.feat_dim if self.gcn else 2 * self.feat_dim))
        self.initializer = initializer
        init.xavier_uniform_(self.weight)
        self.fuse_base_model = fuse_base_model
        self._engine = None
        self._engine_key = None
        self._dev_cache = {}
        print("feat dim:", self.feat_dim, "embed_dim:", self.embed_dim)   # encoders.py:38

    # ------------------------------------------------------------------ helpers
    def _act(self):
        return ops.ACT_SIGMOID if self.initializer in SIGMOID_INITIALIZERS else ops.ACT_RELU

    def _on_device(self, t, tag):
        """Device-resident copy of a host tensor, refreshed when the tensor is modified in place."""
        if t.is_cuda:
            return t
        key = (tag, t.data_ptr(), t._version, tuple(t.shape))
        hit = self._dev_cache.get(tag)
        if hit is None or hit[0] != key:
            hit = (key, t.detach().to("cuda", torch.float32).contiguous())
            self._dev_cache[tag] = hit
        return hit[1]

    def _weight_dev(self):
        w = self.weight
        if w.is_cuda:
            return w
        if torch.is_grad_enabled() and w.requires_grad:
            return w.to("cuda")            # differentiable copy: the gradient flows back to the host Parameter
        return self._on_device(w, "weight")

    def _is_table(self):
        return isinstance(self.features, nn.Embedding)

    def _agg_self_loop(self):
        return bool(getattr(self.aggregator, "gcn", False))

    def _can_fuse_two_hop(self):
        base = getattr(self, "base_model", None)
        return (self.fuse_base_model and isinstance(base, Encoder) and base._is_table()
                and isinstance(self.aggregator, MeanAggregator) and isinstance(base.aggregator, MeanAggregator)
                and self.gcn == base.gcn and self._agg_self_loop() == base._agg_self_loop()
                and self.num_sample is not None and base.num_sample is not None
                and 1 <= self.num_sample <= native.MAX_FANOUT and 1 <= base.num_sample <= native.MAX_FANOUT)

    # ------------------------------------------------------------------ forward
    def forward(self, nodes):
        """Generates embeddings for a batch of nodes.  nodes -- list / array / LongTensor of node ids.
        -> FloatTensor [embed_dim, len(nodes)] (encoders.py:62)"""
        if not torch.cuda.is_available():
            raise native.SageError("sage355.Encoder needs an MI355X; there is no CPU path")
        native.lib()
        if not self.cuda:
            _cap_host_threads_once()
        grad = torch.is_grad_enabled()
        training = grad and any(p.requires_grad for p in self.parameters())
        # The fused two-hop node (autograd._TwoHop) returns gradients for the two `weight` Parameters only -- the model.py:214-215 case, a
        # frozen table.  A TRAINABLE table (nn.Embedding's default) under grad mode takes the per-operator path below, whose
        # autograd.gather_mean / linear_act do reach the table, as the reference's autograd does.
        table_trains = grad and self._can_fuse_two_hop() and self.base_model.features.weight.requires_grad
        if self._can_fuse_two_hop() and not table_trains:
            out = self._forward_two_hop(nodes, training)
        elif self._is_table() and self.num_sample is not None and self.num_sample <= native.MAX_FANOUT \
                and isinstance(self.aggregator, MeanAggregator):
            out = self._forward_table(nodes)
        else:
            out = self._forward_generic(nodes)
        out = out.t()
        return out if self.cuda else out.cpu()

    def _engine_weight(self, tag):
        """The device tensor the engine reads this Encoder's weight from: the Parameter's own storage when it lives on the GPU
        (an in-place optimizer step is then seen as it is), else a persistent device buffer refreshed from the host Parameter
        (strict drop-in mode, cuda=False: model.py keeps the model on the host).  -> (tensor, refreshed)"""
        w = self.weight
        hit = self._dev_cache.get(tag)
        if w.is_cuda:
            if hit is None or hit[0] != "device" or hit[1] != w.data_ptr():
                hit = ["device", w.data_ptr(), w.detach()]          # shares storage AND version counter with the Parameter
                self._dev_cache[tag] = hit
            return hit[2], False
        key = (w.data_ptr(), w._version, tuple(w.shape))
        if hit is None or hit[0] != "host" or hit[2].shape != w.shape:
            hit = ["host", None, torch.empty(w.shape, dtype=torch.float32, device="cuda")]
            self._dev_cache[tag] = hit
        refreshed = hit[1] != key or torch.is_grad_enabled()   # while training, never trust the key: `.data` writes do not move it
        if refreshed:
            hit[2].copy_(w.detach(), non_blocking=True)
            hit[1] = key
        return hit[2], refreshed

    def _forward_two_hop(self, nodes, training=False):
        """Both layers as one C call (sage_forward2).  Under grad mode the same call is one autograd node (autograd._TwoHop)
        whose backward runs the C-ABI backward kernels on the engine's intermediates: the reference's training loop
        (model.py:240-252) then runs at the engine's speed through the unchanged class surface."""
        base = self.base_model
        dev = torch.device("cuda")
        table = base._on_device(base.features.weight, "table")
        n = table.shape[0]
        probe = nodes if isinstance(nodes, (list, tuple, np.ndarray)) else None      # a device tensor is not read back for this
        rp1, c1 = _device_csr(base.adj_lists, n, dev, probe)
        rp2, c2 = (rp1, c1) if self.adj_lists is base.adj_lists else _device_csr(self.adj_lists, n, dev, probe)
        (w1, r1), (w2, r2) = base._engine_weight("engine_w"), self._engine_weight("engine_w")
        key = (rp1.data_ptr(), rp2.data_ptr(), table.data_ptr(), w1.data_ptr(), w2.data_ptr(), base.num_sample,
               self.num_sample, self.gcn, self._agg_self_loop(), base._act(), self._act())
        if self._engine is None or self._engine_key != key:
            self._engine = TwoHopEngine(rp1, c1, table, w1, w2, base.num_sample, self.num_sample, concat=not self.gcn,
                                        agg_self_loop=self._agg_self_loop(), act1=base._act(), act2=self._act(),
                                        nan_empty=True, max_batch=max(len(nodes), 256), rowptr_outer=rp2, col_outer=c2)
            self._engine_key = key
        elif r1 or r2:
            self._engine.invalidate_weights()             # the buffers were rewritten in place: planes / padded copies are stale
        sampler_key = random.getrandbits(64)
        if training:
            ids = ops.as_ids(nodes, dev, n)
            return autograd.two_hop(base.weight, self.weight, self._engine, ids, sampler_key)
        return self._engine.forward(nodes, seed=sampler_key)

    def _forward_table(self, nodes):
        """One layer over a raw feature table (encoders.py:47-62 with features = nn.Embedding)."""
        dev = torch.device("cuda")
        tw = self.features.weight
        if torch.is_grad_enabled() and tw.requires_grad and not tw.is_cuda:
            table = tw.to("cuda", torch.float32)             # differentiable copy: the gradient flows back to the host table
        else:
            table = self._on_device(tw, "table")
        rowptr, col = _device_csr(self.adj_lists, table.shape[0], dev, nodes if isinstance(nodes, (list, tuple, np.ndarray)) else None)
        ids = ops.as_ids(nodes, dev, table.shape[0])
        any_nonempty = torch.zeros(1, dtype=torch.int32, device=dev)
        nbr, cnt, _, _ = ops.sample_neighbors(rowptr, col, ids, self.num_sample, random.getrandbits(64), ops.TAG_INNER,
                                              any_nonempty=any_nonempty)
        self_row = ids if self._agg_self_loop() else None
        w = self._weight_dev()
        agg = autograd.gather_mean(table, nbr, cnt, any_nonempty, None, self_row)
        return autograd.linear_act(agg, w, self._act(), None if self.gcn else table, None if self.gcn else ids)

    def _forward_generic(self, nodes):
        """Opaque feature function / foreign aggregator: the reference's own call sequence
        (encoders.py:47-56) with the aggregation and the contraction on the GPU."""
        node_list = [int(n) for n in nodes]
        neigh_feats = self.aggregator.forward(node_list, [self.adj_lists[n] for n in node_list], self.num_sample,
                                              initializer=self.initializer)
        neigh_feats = neigh_feats.to("cuda", torch.float32)
        self_feats = None
        if not self.gcn:
            ids = torch.LongTensor(node_list)
            self_feats = self.features(ids.cuda() if self.cuda else ids).to("cuda", torch.float32).contiguous()
        return autograd.linear_act(neigh_feats.contiguous(), self._weight_dev(), self._act(), self_feats, None)
