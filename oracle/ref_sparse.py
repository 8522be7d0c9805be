"""Sparse (padded neighbour list) restatement of the same arithmetic, fp64.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Parity status: PINNED through
ref_dense (tests/test_oracle_golden.py checks ref_sparse == ref_dense ==
reference golden vectors on identical neighbour sets).

The dense-mask form (aggregators.py:54-74) needs a [|S1|, |U1|] fp32 mask --
10.1 GB at BASELINE config 3 -- so full-size batches are checked against this
form instead: it computes the same per-row mean (aggregators.py:60-61,74) and
the same relu(W . combined^T) (encoders.py:49-62) from padded neighbour lists,
accumulating in float64 so that it is the more exact side of any comparison.
"""
import numpy as np
import torch


def gather_mean(table, nbr, cnt, self_idx=None, nan_empty=True):
    """table [N,D] float; nbr [n,k] int (entries >= cnt[r] ignored); cnt [n].
    Row r = mean of table[nbr[r,:cnt[r]]] (aggregators.py:60-61,74).
    self_idx (gcn aggregator, aggregators.py:50-51 intended semantics): also
    average in table[self_idx[r]] unless that id is already among the sampled.
    Zero-count rows: NaN when any row of the batch is non-empty (0/0 in the
    reference's mixed batch), zeros when every row is empty."""
    table = torch.as_tensor(table, dtype=torch.float64)
    nbr = torch.as_tensor(np.asarray(nbr), dtype=torch.int64)
    cnt = torch.as_tensor(np.asarray(cnt), dtype=torch.int64)
    n, k = nbr.shape
    valid = torch.arange(k).unsqueeze(0) < cnt.unsqueeze(1)
    total = torch.empty((n, table.shape[1]), dtype=torch.float64)
    step = max(1, (1 << 25) // max(1, k * table.shape[1]))        # bound the [rows, k, D] temporary
    for a in range(0, n, step):
        b = min(n, a + step)
        total[a:b] = (table[nbr[a:b].clamp(min=0)] * valid[a:b].unsqueeze(-1)).sum(1)
    denom = cnt.clone()
    if self_idx is not None:
        self_idx = torch.as_tensor(np.asarray(self_idx), dtype=torch.int64)
        present = ((nbr == self_idx.unsqueeze(1)) & valid).any(1)
        total = total + table[self_idx] * (~present).unsqueeze(1)
        denom = denom + (~present).to(torch.int64)
    out = total / denom.unsqueeze(1).to(torch.float64)
    empty = denom == 0
    if empty.any():
        out[empty] = float("nan") if (nan_empty and (~empty).any()) else 0.0
    return out


def linear_act(self_feats, agg, weight, act="relu"):
    """encoders.py:49-62: act(W . cat(self, agg)^T), returned as [n, H]
    (the reference returns the transpose, [H, n])."""
    agg = torch.as_tensor(agg, dtype=torch.float64)
    weight = torch.as_tensor(weight, dtype=torch.float64)
    combined = agg if self_feats is None else torch.cat(
        [torch.as_tensor(self_feats, dtype=torch.float64), agg], dim=1)
    pre = combined.mm(weight.t())
    if act == "relu":
        return torch.relu(pre)
    if act == "sigmoid":
        return torch.sigmoid(pre)
    return pre


def two_hop_forward(table, w1, w2, seeds, nbr2, cnt2, s1_nodes, nbr1, cnt1, gcn,
                    agg_gcn=False, act1="relu", act2="relu", seed_nbr1=None, seed_cnt1=None):
    """2-hop forward on explicit sampled sets.

    seeds [B]; nbr2/cnt2: layer-2 samples of the seeds (global ids);
    s1_nodes [S]: the ids layer 1 is evaluated on for the aggregation (the
    frontier, any order; must contain every id in nbr2, and the seeds when
    agg_gcn); nbr1/cnt1 [S,k1]:
    their layer-1 samples.  Concat encoder (gcn=False): the reference calls
    layer 1 a SECOND time on the seeds with fresh samples (encoders.py:49-52
    via model.py:220-221); seed_nbr1/seed_cnt1 are those samples.
    Returns [B, H2] float64.
    """
    table = torch.as_tensor(table, dtype=torch.float64)
    s1 = np.asarray(s1_nodes, dtype=np.int64)

    def layer1(ids, nbr, cnt):
        agg = gather_mean(table, nbr, cnt, self_idx=ids if agg_gcn else None)
        return linear_act(None if gcn else table[torch.as_tensor(ids)], agg, w1, act1)

    h1 = layer1(s1, nbr1, cnt1)
    pos = {int(v): i for i, v in enumerate(s1)}
    nbr2 = np.asarray(nbr2)
    cnt2 = np.asarray(cnt2)
    loc2 = np.zeros_like(nbr2, dtype=np.int64)
    for r in range(nbr2.shape[0]):
        for j in range(int(cnt2[r])):
            loc2[r, j] = pos[int(nbr2[r, j])]
    seeds = np.asarray(seeds, dtype=np.int64)
    if agg_gcn:
        # aggregators.py:50-51 (intended): the seed joins its own sampled set,
        # so it is part of the frontier and its self row is h1[frontier pos].
        self_loc = np.array([pos[int(s)] for s in seeds])
        agg2 = gather_mean(h1, loc2, cnt2, self_idx=self_loc)
    else:
        agg2 = gather_mean(h1, loc2, cnt2)
    if gcn:
        self2 = None
    else:
        self2 = layer1(seeds, seed_nbr1, seed_cnt1)
    return linear_act(self2, agg2, w2, act2)
