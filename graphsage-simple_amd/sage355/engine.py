"""TwoHopEngine: the 2-layer forward of graphsage/model.py:219-222 as ONE C call.

Everything the reference rebuilds per batch in Python (adjacency lookups, set
sampling, set union, id->column dict, dense mask) is device resident here:
CSR adjacency, the fp32 feature table and both weight matrices stay in HBM, and
`forward` enqueues sample -> frontier -> sample -> layer 1 -> layer 2 on the
current stream without a host round trip (sage_forward2, include/sage355.h).
"""
import torch

from . import native
from .ops import ACT_RELU, _chk, _need_gpu, _row_major, as_ids


def pretransform_table(table, w1, rows_per_call=1 << 18):
    """INFERENCE at fixed weights, gcn encoder (encoders.py:56-61 with gcn=True): the mean is linear, so
    relu(W1 . mean(X[nbrs])) = relu(mean(Y[nbrs])) with Y = X . W1^T -- layer 1's contraction moves from every forward to weight-
    preparation time (once per weight update: 2 N D0 H1 flop, ~1 ms for a million 256-wide rows), the gather reads rows of H1 floats
    instead of D0 (half the bytes at 256 -> 128) and the per-batch contraction shrinks to H1 x H1.

        y, eye = pretransform_table(table, w1)                       # [N, H1] fp32 on the table's device, identity [H1, H1]
        eng = TwoHopEngine(rowptr, col, y, eye, w2, k1, k2)          # or RolePipeline(rowptr, col, y, eye, w2, ...)

    The engine is used UNCHANGED: with W1 := I the split-bf16 contraction returns its operand exactly (x = hi + mid + lo times 1.0); and
    since the returned identity is MARKED as such, the split layer skips that contraction altogether (the gather applies the activation
    and writes h1) -- the same bits, one launch and one round trip of the means less.
    Same sampled sets as the plain engine for the same keys; values equal to a few fp32 roundings (the sums are associated
    differently: mean of products instead of product of the mean) -- inside the 1e-5 parity bar, not bit-identical.  Differences by
    construction: non-finite FEATURES meet W1 before the mean (Inf - Inf cases of torch.mm land elsewhere); training needs the plain
    engine (Y is stale after every optimizer step; call this again after an update of W1).  Not for the concat encoder (two weight
    halves meet different rows there).  Y is computed by this library's own fp32-accurate contraction (sage_linear_act)."""
    from . import ops
    _need_gpu()
    if w1.dim() != 2 or table.dim() != 2 or w1.shape[1] != table.shape[1]:
        raise native.SageError(f"pretransform_table: W1 {tuple(w1.shape)} does not fit a {table.shape[1]}-wide table (gcn encoder only)")
    n, h1 = table.shape[0], w1.shape[0]
    w = w1.detach().to(table.device, torch.float32).contiguous()
    y = torch.empty((n, h1), dtype=torch.float32, device=table.device)
    for lo in range(0, n, rows_per_call):
        hi = min(n, lo + rows_per_call)
        ops.linear_act(table[lo:hi], w, act=ops.ACT_NONE, out=y[lo:hi])
    eye = torch.eye(h1, dtype=torch.float32, device=table.device)
    eye._sage_identity = True          # TwoHopEngine then declares it to the library (sage_model_t.w1_is_identity): layer 1's contraction is
    return y, eye                      # skipped and the column-sliced gather applies the activation and writes h1 itself


class TwoHopEngine:
    def __init__(self, rowptr, col, table, w1, w2, k1, k2, concat=False, agg_self_loop=False, act1=ACT_RELU,
                 act2=ACT_RELU, nan_empty=True, fused=True, max_batch=4096, rowptr_outer=None, col_outer=None, relabel=None,
                 prepare_weights=True, slice_major="auto", _shared_sliced=None):
        """rowptr/col: CSR of enc1.adj_lists (inner hop); rowptr_outer/col_outer: CSR of
        enc2.adj_lists when it differs (injected pre-sampled sets), default the same.
        w1 [h1, d0 | 2*d0], w2 [h2, h1 | 2*h1]: the Encoders' `weight` Parameters
        (referenced, not copied: an optimizer step is seen by the next forward)."""
        _need_gpu()
        self._ctor = dict(rowptr=rowptr, col=col, table=table, w1=w1, w2=w2, k1=k1, k2=k2, concat=concat, agg_self_loop=agg_self_loop,
                          act1=act1, act2=act2, nan_empty=nan_empty, fused=fused, max_batch=max_batch, rowptr_outer=rowptr_outer,
                          col_outer=col_outer, relabel=relabel, prepare_weights=prepare_weights, slice_major=slice_major)
        self.rowptr1 = _chk(rowptr, torch.int64, "rowptr", 1)
        self.col1 = _chk(col, torch.int32, "col", 1)
        self.rowptr2 = self.rowptr1 if rowptr_outer is None else _chk(rowptr_outer, torch.int64, "rowptr_outer", 1)
        self.col2 = self.col1 if col_outer is None else _chk(col_outer, torch.int32, "col_outer", 1)
        if self.rowptr2.shape[0] != self.rowptr1.shape[0]:
            raise native.SageError("inner and outer CSR must cover the same node ids")
        self.table, self.table_ld = _row_major(table, "table")
        # relabel="degree": work on a copy of graph and table renumbered by descending degree, so that the rows gathered
        # most often are neighbours in memory (config 3: gather 46 -> 40 us).  The outer-hop kernel translates the seeds
        # (model.seed_map), so the caller keeps its ids and outputs stay in the caller's seed order; ids in intermediates()
        # are the INTERNAL ones (self.node_order[i] = caller's id of internal node i).
        # The device sampler is keyed by node id, so the sampled sets differ from the unrelabelled engine's (same law).
        self.node_order = self._new_of_old = None
        if relabel not in (None, "degree"):
            raise native.SageError("relabel must be None or 'degree'")
        if relabel == "degree":
            self._relabel_by_degree()
        self.num_nodes = self.rowptr1.shape[0] - 1
        if self.table.shape[0] < self.num_nodes:
            raise native.SageError(f"table has {self.table.shape[0]} rows for {self.num_nodes} nodes")
        self.w1, self.w2 = w1, w2
        self.d0 = self.table.shape[1]
        self.h1, self.h2 = w1.shape[0], w2.shape[0]
        mult = 2 if concat else 1
        if tuple(w1.shape) != (self.h1, mult * self.d0) or tuple(w2.shape) != (self.h2, mult * self.h1):
            raise native.SageError(f"weight shapes {tuple(w1.shape)}, {tuple(w2.shape)} do not fit d0={self.d0}, concat={concat}")
        # The 16-B-per-lane kernels want row widths that are multiples of 4 floats.  Cora's 1433 raw features and the
        # reference's default 50-wide layer 1 (model.py:543) are not: the engine then works on zero-padded copies
        # (table once -- it is frozen, model.py:214-215 -- and the weights whenever their version counter moves).
        # Zero weight rows / columns make the pad inert: relu(0) = 0, and sigmoid's 0.5 meets a zero column of W2.
        self.d0p, self.h1p = -(-self.d0 // 4) * 4, -(-self.h1 // 4) * 4
        self._padded = (self.d0p != self.d0) or (self.h1p != self.h1)
        if self.d0p != self.d0 or self.table_ld % 4 != 0 or self.table.data_ptr() % 16 != 0:
            padded = torch.zeros((self.table.shape[0], self.d0p), dtype=torch.float32, device=self.table.device)
            padded[:, :self.d0] = self.table
            self.table, self.table_ld = padded, self.d0p
        # Slice-major second copy of the table for the column-sliced layer-1 gather: float[d0 / W][N][W] (W = 32 floats = 128-byte
        # slices by default), so that the XCD that owns a slice reads ONE contiguous array -- consecutive hub rows' slices share DRAM
        # pages and L2 sets -- instead of 128 bytes out of every KiB.  With it the 128-byte slices that round 2 measured SLOWER on the
        # row-major table (fewer bytes past L2, 171 vs 214 MB, but 3.7 TB/s granules) are faster: same-box A/B at config 3, gcn encoder:
        # row-major 65.6 us per forward, slice-major 256-B slices 62.8, 128-B slices + one destination row per lane group 60.6
        # (gather alone 42.5 -> 37.5 us); config 4 (2^23 nodes) 96.0 -> 86.2; caller's node order 73.3 -> 61-64.  The concat encoder
        # gets SLOWER with it (90 -> 96 us: its pacemaker is the two-pass contraction, which a more aggressive gather beside it
        # slows down), so "auto" = gcn encoder only.  Costs a second copy of the table in HBM (1 GB of 288 at config 3); the row-major
        # one stays for whole-row consumers (the concat encoder's own rows, the backward, read-back).
        # slice_major: "auto" / True / False; SAGE_TABLE_SLICED=0 disables, SAGE_TABLE_SLICE_FLOATS = 32 / 64 / 128 picks W.
        import os
        self._table_sliced = _shared_sliced
        self._table_sliced_version = self.table._version
        _sl = os.environ.get("SAGE_TABLE_SLICED", "1")              # 0: never, 1: "auto" as above, 2: "auto" includes the concat encoder (A/B)
        self._want_sliced = slice_major is True or (slice_major == "auto" and (_sl == "2" or (not concat and _sl != "0")))
        self._wpad_key = None
        self._w1p = self._w2p = None
        self._w1prep = self._w1prep_key = None
        self.prepare_weights = bool(prepare_weights)
        self.k1, self.k2 = int(k1), int(k2)
        self.concat, self.agg_self_loop = bool(concat), bool(agg_self_loop)
        self.act1, self.act2 = int(act1), int(act2)
        self.nan_empty, self.fused = bool(nan_empty), bool(fused)
        self.device = self.table.device
        self.max_batch = 0
        self.workspace = None
        self.layout = native.WsLayout()
        self._model_key = None
        self._model_c = None
        self._model_q = None
        self._queue = None
        self._cursor = None
        self._graph = None
        self._last_batch = 0
        self.generation = 0                  # forwards run so far: a backward checks that the workspace still holds ITS forward
        self._bwd = None                     # scratch of backward_weights (allocated on first use)
        self._reserve(max_batch)

    def _relabel_by_degree(self):
        n = self.rowptr1.shape[0] - 1
        deg = self.rowptr1[1:] - self.rowptr1[:-1]
        order = torch.sort(deg, descending=True, stable=True).indices           # internal -> caller's id
        new_of_old = torch.empty_like(order)
        new_of_old[order] = torch.arange(n, device=order.device)

        def renumber(rowptr, col):
            d = rowptr[1:] - rowptr[:-1]
            src = new_of_old[torch.repeat_interleave(torch.arange(n, device=col.device), d)]
            key = torch.sort(src * n + new_of_old[col.long()]).values
            rp = torch.zeros(n + 1, dtype=torch.int64, device=col.device)
            rp[1:] = torch.cumsum(d[order], 0)
            return rp, (key % n).to(torch.int32)

        same = self.rowptr2 is self.rowptr1 and self.col2 is self.col1
        rp1, c1 = renumber(self.rowptr1, self.col1)
        if same:
            rp2, c2 = rp1, c1
        else:
            rp2, c2 = renumber(self.rowptr2, self.col2)
        self.rowptr1, self.col1, self.rowptr2, self.col2 = rp1, c1, rp2, c2
        self.table = self.table[order].contiguous()
        self.table_ld = self.table.shape[1]
        self.node_order, self._new_of_old = order, new_of_old.to(torch.int32)

    def sibling(self):
        """Another engine over the SAME device graph / table / weights (shared, not copied) with a workspace of its own:
        what every additional mini-batch in flight needs."""
        if self._padded:                      # zero-padded copies of table / weights (small graphs): just build another
            return TwoHopEngine(**self._ctor)
        e = TwoHopEngine(self.rowptr1, self.col1, self.table, self.w1, self.w2, self.k1, self.k2, concat=self.concat,
                         agg_self_loop=self.agg_self_loop, act1=self.act1, act2=self.act2, nan_empty=self.nan_empty, fused=self.fused,
                         max_batch=self.max_batch, rowptr_outer=self.rowptr2, col_outer=self.col2, relabel=None,
                         prepare_weights=self.prepare_weights, slice_major=self._table_sliced is not None, _shared_sliced=self._table_sliced)
        e.node_order, e._new_of_old = self.node_order, self._new_of_old
        e._model_key = None
        return e

    def _seeds_in(self, seeds):
        """Seeds cross the boundary in the CALLER's ids; with relabel="degree" the outer-hop kernel translates them
        (model.seed_map), inside the forward."""
        return seeds

    def _weights(self):
        """The weight tensors the kernels read: the caller's own, or zero-padded copies kept in step with them."""
        if not self._padded:
            return self.w1, self.w2
        key = (self.w1.data_ptr(), self.w1._version, self.w2.data_ptr(), self.w2._version)
        if key != self._wpad_key:
            m = 2 if self.concat else 1
            if self._w1p is None:
                self._w1p = torch.zeros((self.h1p, m * self.d0p), dtype=torch.float32, device=self.device)
                self._w2p = torch.zeros((self.h2, m * self.h1p), dtype=torch.float32, device=self.device)
            with torch.no_grad():
                for c in range(m):
                    self._w1p[:self.h1, c * self.d0p: c * self.d0p + self.d0] = self.w1[:, c * self.d0: (c + 1) * self.d0]
                    self._w2p[:, c * self.h1p: c * self.h1p + self.h1] = self.w2[:, c * self.h1: (c + 1) * self.h1]
            self._wpad_key = key
        return self._w1p, self._w2p

    def invalidate_weights(self):
        """Forget every cached form of the weights (zero-padded copies, bf16 planes, the C model struct): the next forward
        rebuilds them from the tensors as they are NOW.  The caches are keyed on (data_ptr, tensor._version), which does not move
        when the weights are written through `.data`, by a collective, or by a replayed hipGraph."""
        self._w1prep_key = self._wpad_key = self._model_key = None

    def refresh_weights(self):
        """Re-read the caller's weights into the padded copies (only needed between replays of a captured graph when
        the widths are padded; `forward` does it by itself)."""
        self._model(queued=self._queue is not None)

    def _prepare_w1(self, w1):
        """enc1.weight split into bf16 planes in the contraction kernel's register order (sage_prepare_weights), redone
        whenever the weight tensor or its version counter changes; None when this layer shape has no prepared form."""
        L = native.lib()
        need = L.sage_prepared_weight_bytes(self.d0p, self.h1p, int(self.concat))
        if need == 0 or not self.prepare_weights:
            return None
        key = (w1.data_ptr(), w1._version)
        if self._w1prep is None or self._w1prep.numel() != need:
            self._w1prep = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._w1prep_key = None
        if self._w1prep_key != key:
            native.check(L.sage_prepare_weights(w1.data_ptr(), w1.stride(0), self.d0p, self.h1p, int(self.concat), self._w1prep.data_ptr(),
                                                need, torch.cuda.current_stream().cuda_stream), "prepare_weights")
            self._w1prep_key = key
        return self._w1prep

    def _slice_table(self):
        """Build (or refresh, if the table was written in place since) the slice-major copy -- only when layer 1 runs as the
        column-sliced gather with 256-byte slices of whole 64-float pieces."""
        import os
        w = int(os.environ.get("SAGE_TABLE_SLICE_FLOATS", "32"))         # floats per slice: 32 (128 B, default) / 64 / 128
        self._slice_floats = w
        ok = (self._want_sliced and bool(self.layout.layer1_split) and w in (32, 64, 128) and self.d0p % w == 0 and self.d0p >= 2 * w
              and self.table_ld == self.d0p and self.table.shape[0] == self.num_nodes)
        if not ok:
            self._table_sliced = None
            return
        if self._table_sliced is None or self._table_sliced_version != self.table._version:
            n_rows = self.table.shape[0]
            src = self.table.view(n_rows, self.d0p // w, w).permute(1, 0, 2)
            if self._table_sliced is not None and tuple(self._table_sliced.shape) == tuple(src.shape):
                # refreshed IN PLACE: a role pipeline built over this engine copied the buffer's pointer at sage_pipe_create
                # (a new tensor per rebuild left the pipe reading a stale, later a freed, copy)
                self._table_sliced.copy_(src)
            else:
                self._table_sliced = src.contiguous()
                self._model_key = None
            self._table_sliced_version = self.table._version

    def refresh_table(self):
        """The table was written in a way that does not move its version counter (`.data`, a collective, a replayed hipGraph): bring the
        engine's slice-major copy up to date, in place (pointers held by role pipelines stay valid).  The table is otherwise treated
        as frozen (model.py:214-215)."""
        self._table_sliced_version = None
        if self.layout.total_bytes:
            self._slice_table()

    def _model(self, queued=False):
        if self.layout.total_bytes:          # the layout is known (after the first _reserve)
            self._slice_table()
        w1, w2 = self._weights()
        prep = self._prepare_w1(w1.detach())
        key = (w1.data_ptr(), w2.data_ptr(), self._queue.data_ptr() if self._queue is not None else 0, prep.data_ptr() if prep is not None else 0)
        if self._model_key == key:
            return self._model_q if queued else self._model_c
        _row_major(w1.detach(), "w1")
        _row_major(w2.detach(), "w2")
        if not (w1.is_contiguous() and w2.is_contiguous()):
            raise native.SageError("weights must be contiguous")
        self._model_c = native.Model(
            self.rowptr1.data_ptr(), self.col1.data_ptr(), self.rowptr2.data_ptr(), self.col2.data_ptr(), self.num_nodes,
            self.table.data_ptr(), self.table_ld, self.d0p, w1.data_ptr(), self.h1p, w2.data_ptr(), self.h2, self.k1, self.k2,
            int(self.concat), int(self.agg_self_loop), self.act1, self.act2, int(self.nan_empty), int(self.fused),
            int(self.max_batch))
        self._model_c.w1_prepared = prep.data_ptr() if prep is not None else None
        self._model_c.seed_map = self._new_of_old.data_ptr() if self._new_of_old is not None else None
        self._model_c.table_sliced = self._table_sliced.data_ptr() if self._table_sliced is not None else None
        self._model_c.table_slice_floats = getattr(self, "_slice_floats", 64)
        self._model_c.w1_is_identity = 1 if (getattr(self.w1, "_sage_identity", False) and not self.concat and not self._padded) else 0
        self._model_q = None
        if self._queue is not None:
            self._model_q = native.Model.from_buffer_copy(self._model_c)
            self._model_q.queue = self._queue.data_ptr()
            self._model_q.queue_len = self._queue.shape[0]
            self._model_q.queue_cursor = self._cursor.data_ptr()
        self._model_key = key
        return self._model_q if queued else self._model_c

    # ---- device-side batch queue + hipGraph replay (no per-batch host work) ----
    def set_queue(self, seeds, rng_seeds):
        """seeds: int32 device tensor [S, B] (kept alive by the engine); rng_seeds: S sampler keys.
        Builds the ring of sage_batch_t descriptors the kernels read (include/sage355.h)."""
        if not (isinstance(seeds, torch.Tensor) and seeds.is_cuda and seeds.dtype == torch.int32 and seeds.dim() == 2
                and seeds.is_contiguous()):
            raise native.SageError("set_queue: seeds must be a contiguous int32 device tensor [S, B]")
        s, b = seeds.shape
        if len(rng_seeds) != s:
            raise native.SageError("set_queue: one sampler key per batch")
        if seeds.numel() and (int(seeds.min()) < 0 or int(seeds.max()) >= self.num_nodes):     # one sync, outside any timed loop
            raise native.SageError(f"set_queue: seed id outside [0, {self.num_nodes})")
        seeds = self._seeds_in(seeds).contiguous()
        self._reserve(b)
        desc = torch.empty((s, 2), dtype=torch.int64)
        desc[:, 0] = seeds.data_ptr() + torch.arange(s, dtype=torch.int64) * (b * 4)
        keys = [int(x) & 0xFFFFFFFFFFFFFFFF for x in rng_seeds]
        desc[:, 1] = torch.tensor([k - (1 << 64) if k >= (1 << 63) else k for k in keys], dtype=torch.int64)
        self._queue_seeds = seeds
        self._queue = desc.to(self.device)
        self._cursor = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._queue_batch = b
        self._graph = None
        self._model_key = None

    def forward_queued(self, out):
        """One forward on the batch at the queue cursor (advances it).  Capturable."""
        L = native.lib()
        rc = L.sage_forward2(self._model(queued=True), self.workspace.data_ptr(), self.workspace.numel(), None,
                             self._queue_batch, 0, out.data_ptr(), out.stride(0), torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            native.check(rc, "forward2 (queued)")
        self._last_batch = self._queue_batch
        self.generation += 1
        return out

    def capture(self, out=None, batches=1):
        """Capture queued forwards into a hipGraph (torch.cuda.CUDAGraph); replay() then runs batch after batch with a
        single graph launch each.  batches > 1: one replay embeds that many consecutive batches of the ring into
        out[0..batches-1] -- for small batches (Pubmed's 256 seeds: 20 us of GPU work) the host's graph launch is
        otherwise the bottleneck."""
        if self._queue is None:
            raise native.SageError("capture: call set_queue first")
        batches = int(batches)
        shape = (self._queue_batch, self.h2) if batches == 1 else (batches, self._queue_batch, self.h2)
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=self.device)
        elif tuple(out.shape) != shape:
            raise native.SageError(f"capture: `out` must have shape {shape}")
        self._graph_out = out
        outs = [out] if batches == 1 else [out[j] for j in range(batches)]
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # warm-up outside capture (function attributes, lazy init)
            self.forward_queued(outs[0])
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for o in outs:
                self.forward_queued(o)
        self._graph = g
        self._cursor.zero_()
        torch.cuda.synchronize()
        return out

    def replay(self):
        self._graph.replay()
        return self._graph_out

    def rewind(self, position=0):
        self._cursor.fill_(int(position))

    def _reserve(self, batch):
        if batch <= self.max_batch:
            return
        self.max_batch = int(batch)
        self._model_key = None
        model = self._model()
        native.check(native.lib().sage_forward2_layout(model, self.max_batch, self.layout), "forward2_layout")
        self.workspace = torch.zeros(self.layout.total_bytes, dtype=torch.uint8, device=self.device)
        native.check(native.lib().sage_forward2_init(model, self.workspace.data_ptr(), self.workspace.numel(), self.max_batch,
                                                     torch.cuda.current_stream().cuda_stream), "forward2_init")
        self._graph = None

    def forward(self, seeds, seed=0, out=None, stage_events=None):
        """seeds: int32 device tensor (or anything as_ids takes) -> out [B, h2] on device.
        `seed` keys the sampler: the sets are a pure function of (seed, node id, hop).
        stage_events: optional (c_void_p * 8) of hipEvent_t recorded around the four stages."""
        if not (isinstance(seeds, torch.Tensor) and seeds.is_cuda and seeds.dtype == torch.int32 and seeds.is_contiguous()):
            seeds = as_ids(seeds, self.device, self.num_nodes)
        seeds = self._seeds_in(seeds)
        b = seeds.shape[0]
        if b > self.max_batch:
            self._reserve(b)
        if out is None:
            out = torch.empty((b, self.h2), dtype=torch.float32, device=self.device)
        elif out.shape != (b, self.h2) or out.dtype != torch.float32 or not out.is_cuda or out.stride(1) != 1:
            raise native.SageError("forward: `out` must be a [B, h2] fp32 device tensor with unit inner stride")
        L = native.lib()
        args = (self._model(), self.workspace.data_ptr(), self.workspace.numel(), seeds.data_ptr(), b,
                int(seed) & 0xFFFFFFFFFFFFFFFF, out.data_ptr(), out.stride(0), torch.cuda.current_stream().cuda_stream)
        rc = L.sage_forward2(*args) if stage_events is None else L.sage_forward2_profiled(*args, stage_events)
        if rc != 0:
            try:
                native.check(rc, "forward2")
            finally:
                # a forward that stopped between two of its launches leaves sampled sets and frontier keys behind (layer 2's last
                # block is what cleans up): give the next call a clean workspace rather than a full hash table
                L.sage_forward2_init(args[0], self.workspace.data_ptr(), self.workspace.numel(), self.max_batch, args[-1])
        self._last_batch = b
        self.generation += 1
        return out

    # ---- backward of the last forward through the intermediates it left in the workspace (model.py:249) ----
    def backward_weights(self, out, grad_out, need_w1=True):
        """Gradients of (w1, w2), in the caller's shapes, of the LAST forward on this engine: `out` is what it returned
        ([B, h2]), grad_out = d loss / d out.  Autograd of the reference's expression (mm / relu / cat / div) from
        the engine's own intermediates -- nbr / cnt / row2 / h1 (and the layer-1 means when layer 1 ran split) are still in the
        workspace, the layer-2 means are re-gathered -- with the C-ABI backward kernels.  The table is frozen (model.py:214-215):
        no gradient reaches it.  No host synchronisation: the frontier size never leaves the device."""
        from . import ops
        if getattr(self.w1, "_sage_identity", False):
            raise native.SageError("backward_weights: this engine serves a PRE-TRANSFORMED table (W1 is a declared identity): train with the "
                                   "plain engine on the raw features")
        lib = native.lib()
        st = native.stream_handle()
        P = native.ptr
        L, b = self.layout, self._last_batch
        dev = self.device
        if self._bwd is None or self._bwd["max_s1"] != L.max_s1:
            self._bwd = {"max_s1": L.max_s1, "nlive": torch.zeros(1, dtype=torch.int32, device=dev),
                         "agg1": None,
                         "any": torch.ones(1, dtype=torch.int32, device=dev)}
        sc = self._bwd
        first = b if self.concat else 0
        k1, k2, h1p, d0p = self.k1, self.k2, self.h1p, self.d0p
        # rows of layer 1 = first + frontier size, kept on the device (counters[8] is the read-back copy the forward's last block leaves)
        torch.add(self._view(L.counters, 16, torch.int32)[8:9], first, out=sc["nlive"])
        grad_out = grad_out.contiguous()
        w1p, w2p = self._weights()
        h1 = self._view(L.h1, L.max_s1 * h1p, torch.float32).view(L.max_s1, h1p)
        row2 = self._view(L.row2, b * k2, torch.int32).view(b, k2)
        cnt2 = self._view(L.cnt2, b, torch.int32)
        self_row2 = self._view(L.self_row2, b, torch.int32) if self.agg_self_loop else None
        nbr1 = self._view(L.nbr1, L.max_s1 * k1, torch.int32).view(L.max_s1, k1)
        cnt1 = self._view(L.cnt1, L.max_s1, torch.int32)
        s1_nodes = self._view(L.s1_nodes, L.max_s1, torch.int32)
        def scratch(name, nbytes):
            t = sc.get(name)
            if t is None or t.numel() < nbytes:
                t = sc[name] = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)
            return t

        # ---- layer 2 backward: agg2 is recomputed (one small gather), then dW2 and d[h1_self | agg2].  The *_ws entry points are
        #      the reproducible forms (partial tiles added in a fixed order; inverted index instead of atomics): two runs of one
        #      schedule give the same bits, and a captured step equals the eager one bit for bit
        agg2 = ops.gather_mean(h1, row2, cnt2, self_row=self_row2, any_nonempty=sc["any"])
        mult = 2 if self.concat else 1
        g_w2p = torch.zeros_like(w2p)
        g_x2 = torch.empty(b, mult * h1p, device=dev)
        ws2 = scratch("ws_dw2", lib.sage_linear_act_backward_workspace_bytes(b, h1p, int(self.concat), self.h2))
        native.check(lib.sage_linear_act_backward_ws(P(h1) if self.concat else None, h1p, None, P(agg2), agg2.stride(0), h1p, P(w2p),
                                                     w2p.stride(0), self.h2, self.act2, P(out), out.stride(0), P(grad_out), grad_out.stride(0), b, None,
                                                     P(g_w2p), g_w2p.stride(0), P(g_x2) if need_w1 else None, g_x2.stride(0), None,
                                                     P(ws2), ws2.numel(), st), "linear_act_backward (layer 2)")
        g_w1p = None
        if need_w1:
            # ---- layer 1 backward: only dW1 (the table is frozen), summed over the OUTER EDGES (sage_two_hop_grad_w1): the terms
            #      act1'(h1[t]) . g_agg2[r] / c_r (x) X1[t] come in (seed, slot) order, so neither grad_h1 (a scatter) nor an order of the
            #      frontier's arbitrarily placed rows is needed, and the bits do not depend on the layout.  agg1 from the workspace
            #      (split layer) or recomputed on the live rows
            self_row1 = s1_nodes if self.agg_self_loop else None
            if L.layer1_split:
                # the split layer (sliced gather + contraction) left the means of this very forward in the workspace: no second gather
                agg1 = self._view(L.agg1, L.max_s1 * d0p, torch.float32).view(L.max_s1, d0p)
            else:
                if sc["agg1"] is None:
                    sc["agg1"] = torch.zeros(L.max_s1, d0p, device=dev)
                ops.gather_mean(self.table, nbr1, cnt1, self_row=self_row1, any_nonempty=sc["any"], n_dev=sc["nlive"], out=sc["agg1"])
                agg1 = sc["agg1"]
            g_w1p = torch.zeros_like(w1p)
            ws1 = scratch("ws_dw1", lib.sage_two_hop_grad_w1_workspace_bytes(b, k2, d0p, int(self.concat), h1p))
            native.check(lib.sage_two_hop_grad_w1(P(g_x2), g_x2.stride(0), P(row2), P(cnt2), k2, P(self_row2), b, P(h1), h1p, h1p, self.act1,
                                                  P(agg1), agg1.stride(0), d0p, int(self.concat), P(self.table) if self.concat else None,
                                                  self.table_ld, P(s1_nodes) if self.concat else None, P(g_w1p), g_w1p.stride(0),
                                                  P(ws1), ws1.numel(), st), "two_hop_grad_w1")
        # padded widths (Cora 1433 -> 1436, 50 -> 52): gradients of the caller's own shapes
        if self._padded:
            g_w1 = None if g_w1p is None else torch.cat([g_w1p[:self.h1, c * d0p: c * d0p + self.d0] for c in range(mult)], 1)
            g_w2 = torch.cat([g_w2p[:, c * h1p: c * h1p + self.h1] for c in range(mult)], 1)
        else:
            g_w1, g_w2 = g_w1p, g_w2p
        return g_w1, g_w2

    # ---- read-back of the last forward's intermediates (tests, parity gate, byte counting) ----
    def _view(self, off, count, dtype):
        itemsize = torch.empty(0, dtype=dtype).element_size()
        return self.workspace[off: off + count * itemsize].view(dtype)

    def intermediates(self):
        """Sampled sets and layer-1 state of the LAST forward (synchronises)."""
        L, b = self.layout, self._last_batch
        torch.cuda.synchronize()
        counters = self._view(L.counters, 16, torch.int32).cpu()
        first = b if self.concat else 0
        n1 = first + int(counters[8])          # [8..15]: the counters as the last forward left them
        s1 = self._view(L.s1_nodes, L.max_s1, torch.int32)[:n1]
        return {
            "n_s1": n1, "first_frontier_row": first, "s1_nodes": s1,
            "nbr2": self._view(L.nbr2, b * self.k2, torch.int32).view(b, self.k2),
            "cnt2": self._view(L.cnt2, b, torch.int32),
            "row2": self._view(L.row2, b * self.k2, torch.int32).view(b, self.k2),
            "nbr1": self._view(L.nbr1, L.max_s1 * self.k1, torch.int32).view(L.max_s1, self.k1)[:n1],
            "cnt1": self._view(L.cnt1, L.max_s1, torch.int32)[:n1],
            "h1": self._view(L.h1, L.max_s1 * self.h1p, torch.float32).view(L.max_s1, self.h1p)[:n1, :self.h1],
        }



_PROCESS_ROLE_STREAMS = {}          # (device, distinct role letters, priorities) -> the role streams of the process's first such pipeline


class RolePipeline:
    """Consecutive 2-hop forwards software-pipelined over ROLE STREAMS (sage_pipe_*, include/sage355.h).

    Stage S (outer + inner sample), G (layer-1 gather), D (layer-1 contraction) and L (layer 2) each get a HIP
    stream; batch b+1 is gathered while batch b is contracted and batch b+2 is sampled, over `depth` workspaces.
    Bit-identical to TwoHopEngine.forward on the same (seeds, key).  `roles` maps the four roles onto streams:
    "SGDL" = four streams, "SGDD" = D and L share one, "SSSS" = one stream (= sage_forward2's launch order).
    `priorities`: per distinct stream, 0 = default, -1 = high (HIP stream priority; the latency-bound roles).
    The reference has no counterpart (model.py:240-252 is one batch at a time on the host)."""

    def __init__(self, rowptr, col, table, w1, w2, k1, k2, batch, depth=4, roles="SGDL", priorities=None, streams=None, threads=None,
                 window=None, **engine_kwargs):
        import ctypes
        import os
        if depth < 1 or depth > native.PIPE_MAX_DEPTH:
            raise native.SageError(f"RolePipeline: depth must be in [1, {native.PIPE_MAX_DEPTH}]")
        if len(roles) != 4:
            raise native.SageError("RolePipeline: roles is a 4-letter map of S, G, D, L onto streams, e.g. 'SGDL' or 'SGDD'")
        e0 = TwoHopEngine(rowptr, col, table, w1, w2, k1, k2, max_batch=batch, **engine_kwargs)
        self.engines = [e0] + [e0.sibling() for _ in range(depth - 1)]
        self.device, self.batch, self.depth, self.h2 = e0.device, int(batch), int(depth), e0.h2
        names = []
        for ch in roles:
            if ch not in names:
                names.append(ch)
        priorities = priorities or {}
        # Which hardware queue a NEW HIP stream lands on is the runtime's choice, and two role streams on one queue serialise (82-105 us per
        # forward instead of 59-90, seen for the second pipeline of a process).  So every pipeline of a process runs on the role streams
        # of the FIRST one with the same (device, role map, priorities) unless the caller hands in streams of its own or asks for new
        # ones (streams="new"); pipes that share streams and are used at the same time interleave in stream order, which is still correct.
        key = (str(self.device), "".join(names), tuple(int(priorities.get(ch, 0)) for ch in names))
        if streams is None and key in _PROCESS_ROLE_STREAMS:
            streams = _PROCESS_ROLE_STREAMS[key]
        if streams is not None and not isinstance(streams, str):
            if len(streams) != len(names):
                raise native.SageError(f"RolePipeline: {len(names)} distinct role streams needed, {len(streams)} given")
            self._streams = dict(zip(names, streams))
        else:
            self._streams = {ch: torch.cuda.Stream(device=self.device, priority=int(priorities.get(ch, 0))) for ch in names}
            _PROCESS_ROLE_STREAMS.setdefault(key, list(self._streams.values()))
        self.role_streams = [self._streams[ch] for ch in roles]
        ws = (ctypes.c_void_p * depth)(*[e.workspace.data_ptr() for e in self.engines])
        st = (ctypes.c_void_p * 4)(*[s.cuda_stream for s in self.role_streams])
        self._h = ctypes.c_void_p()
        self._wkey = None
        native.check(native.lib().sage_pipe_create(e0._model(), self.batch, depth, ws, e0.workspace.numel(), st, ctypes.byref(self._h)),
                     "pipe_create")
        self._wkey = self._weights_key()
        self._keep = []
        # host enqueue threads (one per role stream; sage_pipe_set_threads): submit() then only posts the batch.  Opt-in (threads=True or
        # SAGE_PIPE_THREADS=1), because a caller that synchronises the device itself must then flush() first; bench.py opts in
        if threads is None:
            threads = os.environ.get("SAGE_PIPE_THREADS", "0") == "1" and len(names) == 4
        if window is None:
            window = int(os.environ.get("SAGE_PIPE_WINDOW", "0"))
        self.threads, self._window = False, int(window)
        if threads:
            self.set_threads(True, window)
        # the workspaces were zeroed and initialised, the slice-major table copy and the weight planes written, on the CURRENT stream; the
        # role streams are non-blocking streams of their own and would otherwise start the first batches beside that work (a sampler
        # filling a frontier table that a memset is still wiping leaves stale keys behind for good)
        self.fork()

    def set_threads(self, on, window=None):
        """Start (or drain and stop) the four host enqueue threads.  `window` > 0: role S enqueues batch b only once batch
        b - window has left the GPU (bounds the host's run-ahead); None keeps the pipe's setting."""
        if window is not None:
            self._window = int(window)
        native.check(native.lib().sage_pipe_set_threads(self._h, 1 if on else 0, self._window), "pipe_set_threads")
        self.threads = bool(on)

    def flush(self):
        """Every submitted batch has been ENQUEUED on the role streams (host enqueue threads; a no-op without them).  Call it
        before synchronising the device or the role streams yourself; join() / synchronize() do."""
        rc = native.lib().sage_pipe_flush(self._h)
        if rc != 0:
            self._broken = rc == native.ELAUNCH
            native.check(rc, "pipe_flush")

    def _check_usable(self):
        # a submit that failed between two of its enqueues leaves a batch half-way through the role streams, its workspace dirty and
        # its hand-off events unrecorded: later batches on that slot would wait for ever or sample into a full frontier table
        if getattr(self, "_broken", False):
            raise native.SageError("RolePipeline: an earlier submit failed half-way; synchronise, drop this pipe and create a new one")

    def _weights_key(self):
        e0 = self.engines[0]
        w1, w2 = e0._weights()
        return (w1.data_ptr(), w2.data_ptr(), w1._version, w2._version, e0.table._version)

    def _sync_weights(self):
        """Weights or table written in place since the last submit (version counters): re-prepare the weight planes / refresh the
        slice-major table copy on the CURRENT stream -- the stream such a write was made on -- and make the role streams wait for it."""
        key = self._weights_key()
        if key != self._wkey:
            self.join()                           # batches still in the pipe read the planes / the copy that are about to be rewritten
            m = self.engines[0]._model()          # re-prepares the weight planes, refreshes the slice-major copy in place: current stream
            native.check(native.lib().sage_pipe_update_weights(self._h, m.w1, m.w2, m.w1_prepared), "pipe_update_weights")
            self.fork()                           # the role streams wait for that
            self._wkey = key

    def refresh_table(self):
        """The table was written without moving its version counter (`.data`, a collective): refresh the engine's slice-major copy in
        place and order the role streams behind it (and behind whatever wrote the table on the current stream)."""
        self.join()
        self.engines[0].refresh_table()
        self.fork()

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                native.lib().sage_pipe_flush(h)
                torch.cuda.synchronize()
                native.lib().sage_pipe_destroy(h)
            except Exception:
                pass
            self._h = None

    def distinct_streams(self):
        """The pipe's distinct role streams, in first-use order of `roles` (pass them to another pipe's `streams=`)."""
        return list(self._streams.values())

    @property
    def express_count(self):
        """Batches of this pipe that found it idle and were enqueued whole on stream L (csrc/sage_pipe.hip, "express lane")."""
        return int(native.lib().sage_pipe_express_count(self._h))

    def fork(self, stream=None):
        """Every role stream waits for `stream` (default: the current one): inputs written there are ready."""
        s = stream or torch.cuda.current_stream()
        native.check(native.lib().sage_pipe_fork(self._h, s.cuda_stream), "pipe_fork")

    def join(self, stream=None):
        """`stream` (default: the current one) waits for everything submitted so far."""
        s = stream or torch.cuda.current_stream()
        native.check(native.lib().sage_pipe_join(self._h, s.cuda_stream), "pipe_join")

    def submit_profiled(self, seeds, key, out, gather_events):
        """submit() with two hipEvent_t (a ctypes c_void_p * 2) recorded on stream G around the layer-1 gather."""
        self._sync_weights()
        self._check_usable()
        rc = native.lib().sage_pipe_submit_profiled(self._h, seeds.data_ptr(), int(key) & 0xFFFFFFFFFFFFFFFF, out.data_ptr(),
                                                    out.stride(0), gather_events)
        if rc != 0:
            self._broken = rc == native.ELAUNCH
            native.check(rc, "pipe_submit_profiled")

    def submit(self, seeds, key, out):
        """One batch: seeds int32 [batch] device tensor, out [batch, h2] fp32 device tensor (both must stay alive
        and unmodified until the batch has left the pipe: join() + synchronize, or an event on stream L)."""
        if not (isinstance(seeds, torch.Tensor) and seeds.is_cuda and seeds.dtype == torch.int32 and seeds.is_contiguous()
                and seeds.numel() == self.batch):
            raise native.SageError("RolePipeline.submit: seeds must be a contiguous int32 device tensor of `batch` ids")
        if out.shape != (self.batch, self.h2) or out.dtype != torch.float32 or not out.is_cuda or out.stride(1) != 1:
            raise native.SageError("RolePipeline.submit: `out` must be a [batch, h2] fp32 device tensor with unit inner stride")
        self._check_usable()
        self._sync_weights()
        rc = native.lib().sage_pipe_submit(self._h, seeds.data_ptr(), int(key) & 0xFFFFFFFFFFFFFFFF, out.data_ptr(), out.stride(0))
        if rc != 0:
            self._broken = rc == native.ELAUNCH      # argument errors are raised before anything is enqueued: the pipe stays usable
            native.check(rc, "pipe_submit")

    def submit_many(self, seeds, keys, out, segment_start=False):
        """seeds int32 [n, batch] (device, contiguous), keys: n sampler keys, out [slots, batch, h2] with slots >= depth
        (batch i lands in out[i % slots]) -- ONE host call enqueues all n batches."""
        import ctypes
        if not (isinstance(seeds, torch.Tensor) and seeds.is_cuda and seeds.dtype == torch.int32 and seeds.is_contiguous()
                and seeds.dim() == 2 and seeds.shape[1] == self.batch):
            raise native.SageError("RolePipeline.submit_many: seeds must be a contiguous int32 device tensor [n, batch]")
        n = seeds.shape[0]
        if len(keys) != n:
            raise native.SageError("RolePipeline.submit_many: one sampler key per batch")
        if (out.dim() != 3 or out.shape[1:] != (self.batch, self.h2) or out.dtype != torch.float32 or not out.is_cuda
                or not out.is_contiguous() or out.shape[0] < self.depth):
            raise native.SageError("RolePipeline.submit_many: `out` must be a contiguous [slots >= depth, batch, h2] fp32 device tensor")
        self._sync_weights()
        karr = (ctypes.c_uint64 * n)(*[int(k) & 0xFFFFFFFFFFFFFFFF for k in keys])
        self._check_usable()
        rc = native.lib().sage_pipe_submit_many(self._h, seeds.data_ptr(), self.batch, karr, n, out.data_ptr(), out.stride(1),
                                                out.stride(0), out.shape[0], 1 if segment_start else 0)
        if rc != 0:
            self._broken = rc == native.ELAUNCH
            native.check(rc, "pipe_submit_many")

    def reset(self):
        """Forget every submit (after synchronising): the next `depth` submits find their workspaces free."""
        native.check(native.lib().sage_pipe_reset(self._h), "pipe_reset")

    def capture(self, seeds, keys, out, stream=None):
        """The n batches seeds[i] / keys[i] -> out[i % slots] through the role streams, captured as ONE hipGraph:
        fork -> submit_many -> join on `stream` (default: a stream of the pipe's own).  -> (torch.cuda.CUDAGraph, stream);
        `graph.replay()` under `torch.cuda.stream(stream)` then runs all n batches with a single host call.  The graph embeds
        these seeds / keys / out tensors (keep them alive and unmodified in place of new data: write new seeds INTO `seeds`).
        Bit-identical to submit() on the same batches.  The pipe must be idle (synchronise first); eager submission afterwards
        needs reset()."""
        self._sync_weights()
        self._check_usable()
        if stream is None:
            stream = getattr(self, "_cap_stream", None) or torch.cuda.Stream(device=self.device)
            self._cap_stream = stream
        self.flush()
        torch.cuda.synchronize()
        was_threaded = self.threads
        if was_threaded:
            self.set_threads(False)          # a capture records the calls of the capturing thread
        self.reset()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(stream):
            with torch.cuda.graph(g, stream=stream):
                self.fork(stream)
                self.submit_many(seeds, keys, out, segment_start=True)
                self.join(stream)
        self.reset()
        if was_threaded:
            self.set_threads(True)
        self._keep.append((seeds, out))
        return g, stream

    def synchronize(self):
        self.flush()
        for s in self._streams.values():
            s.synchronize()
