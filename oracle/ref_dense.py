"""Reference-faithful CPU restatement of the GraphSAGE mean-aggregate forward.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Parity status: PINNED -- every
function here is checked against golden vectors produced by importing the
reference itself (tests/golden/make_golden.py -> tests/golden/*.npz,
tests/test_oracle_golden.py).

Same algorithm as the reference, step for step, so its CPU timing is a fair
stand-in for "the reference's pure-PyTorch CPU path" on a machine the reference
cannot travel to:

  step                           reference
  -----------------------------  ------------------------------------------
  fixed-fanout set sampling      graphsage/aggregators.py:42-48
  optional self-loop (gcn agg)   graphsage/aggregators.py:50-51 (intended
                                 semantics; the reference line raises TypeError)
  frontier dedupe + column map   graphsage/aggregators.py:52-53
  dense 0/1 mask, row normalise  graphsage/aggregators.py:54-61
  feature fetch for frontier     graphsage/aggregators.py:62-65
  masked mean as a matmul        graphsage/aggregators.py:74
  self features + concat         graphsage/encoders.py:49-56
  weight contraction + act       graphsage/encoders.py:58-62
"""
import random

import torch

SIGMOID_INITIALIZERS = ("node_degree", "shared", "pagerank")  # encoders.py:58


def sample_sets(to_neighs, num_sample, rng=random):
    """aggregators.py:42-48: k distinct uniform neighbours when deg >= k,
    otherwise the whole set; ``num_sample is None`` passes sets through."""
    if num_sample is None:
        return list(to_neighs)
    out = []
    for neigh in to_neighs:
        if len(neigh) >= num_sample:
            # the reference hands the set itself to random.sample, which (Python
            # 3.10) draws from tuple(set); doing that conversion here consumes
            # the identical RNG stream and also runs on Python >= 3.11.
            out.append(set(rng.sample(tuple(neigh), num_sample)))
        else:
            out.append(neigh)
    return out


def mean_aggregate(nodes, samp_neighs, features, gcn=False):
    """aggregators.py:50-74 on already-sampled sets.

    features: callable LongTensor[n] -> FloatTensor[n, D].
    Returns (to_feats [len(nodes), D], unique_nodes_list).
    A zero-degree row in a batch that also holds non-empty rows is 0/0 = NaN,
    an all-empty batch gives zeros -- both as the reference (aggregators.py:60-61).
    """
    if gcn:
        samp_neighs = [set(s) | {int(nodes[i])} for i, s in enumerate(samp_neighs)]
    unique_nodes_list = list(set.union(*samp_neighs))   # same expression => same iteration order
    column_of = {n: c for c, n in enumerate(unique_nodes_list)}
    mask = torch.zeros(len(samp_neighs), len(unique_nodes_list))
    # one comprehension per index list, as the reference builds them (aggregators.py:55-56): an append loop here
    # made this restatement ~1.2x slower than the reference it stands in for (tests/golden/cpu_port_vs_reference.json)
    cols = [column_of[n] for s in samp_neighs for n in s]
    rows = [r for r, s in enumerate(samp_neighs) for _ in s]
    mask[rows, cols] = 1
    num_neigh = mask.sum(1, keepdim=True)
    mask = mask.div(num_neigh)
    embed_matrix = features(torch.LongTensor(unique_nodes_list))
    return mask.mm(embed_matrix), unique_nodes_list


def encoder_forward(nodes, adj_lists, features, weight, num_sample, gcn,
                    agg_gcn=False, initializer="None", rng=random, return_agg=False):
    """encoders.py:47-62 -> [embed_dim, len(nodes)]."""
    to_neighs = [adj_lists[int(n)] for n in nodes]
    samp = sample_sets(to_neighs, num_sample, rng)
    neigh_feats, _ = mean_aggregate(nodes, samp, features, gcn=agg_gcn)
    if not gcn:
        self_feats = features(torch.LongTensor([int(n) for n in nodes]))
        combined = torch.cat([self_feats, neigh_feats], dim=1)
    else:
        combined = neigh_feats
    pre = weight.mm(combined.t())
    out = torch.sigmoid(pre) if initializer in SIGMOID_INITIALIZERS else torch.relu(pre)
    return (out, neigh_feats) if return_agg else out


def two_hop_forward(seeds, adj1, adj2, table, w1, w2, k1, k2, gcn,
                    agg_gcn=False, initializer1="None", initializer2="None", rng=random):
    """The 2-layer stack wired as graphsage/model.py:214-222: layer 2's feature
    function is layer 1 evaluated on whatever ids layer 2 asks for."""
    def raw(ids):
        return table[ids]

    def hidden(ids):
        return encoder_forward(ids, adj1, raw, w1, k1, gcn, agg_gcn, initializer1, rng).t()

    return encoder_forward(seeds, adj2, hidden, w2, k2, gcn, agg_gcn, initializer2, rng)
