"""MeanAggregator -- host-side mirror of graphsage/aggregators.py:12-76.

Same constructor and ``forward`` signature as the reference class, so it can be
imported in its place (``from graphsage.aggregators import MeanAggregator`` via
the shim package next to this one).  What differs is where the work happens:

reference (aggregators.py)                      here
----------------------------------------------  ---------------------------------
42-48  random.sample per node (Python)          same calls, same RNG stream, when the
                                                caller hands in Python sets (this
                                                method); device sampler when an
                                                Encoder drives the fast path
52-57  set union, dict, dense [B,U] mask        ids -> padded [n,k] position lists
60-61  mask.sum / mask.div                      \\
62-65  features(unique ids)                      > sage_gather_mean (HIP): one wave
74     mask.mm(embed_matrix)                    /  per row, no mask, no matmul

``forward`` keeps the reference's exact semantics, including its use of
Python's global ``random`` stream, which makes it the strict drop-in path: for
the same ``random.seed`` it aggregates the SAME neighbour sets as the reference.
The throughput path does not go through this method: ``Encoder.forward`` samples
on the device from CSR (see encoders.py in this package).
"""
import random

import numpy as np
import torch
import torch.nn as nn

from . import autograd, native


class MeanAggregator(nn.Module):
    """Aggregates a node's embeddings using the mean of its (sampled) neighbours' embeddings."""

    def __init__(self, features, initializer="None", cuda=False, gcn=False, feature_dim=100, num_nodes=100):
        """features -- callable LongTensor[n] -> FloatTensor[n, D] (an nn.Embedding or a lambda)
        cuda     -- True: tensors are returned on the MI355X; False: on the CPU, as the reference
                    returns them (the mean itself always runs on the GPU)
        gcn      -- add the node itself to its neighbour set (aggregators.py:50-51; that line raises
                    TypeError in the reference, this implements the intended set union)
        initializer may be a non-string: model.py:220 passes num_nodes in this position."""
        super(MeanAggregator, self).__init__()
        self.features = features
        self.cuda = cuda          # noqa: shadows nn.Module.cuda exactly as aggregators.py:28 does
        self.gcn = gcn
        if initializer in ["1hot", "node_degree"]:
            # aggregators.py:30-31 -- parameter kept for state_dict compatibility; the detour
            # itself (aggregators.py:68-71) is out of scope
            self.embed = nn.Embedding(num_nodes, feature_dim)

    def forward(self, nodes, to_neighs, num_sample=10, initializer="None"):
        """nodes      -- list of nodes in a batch
        to_neighs  -- list of sets, the neighbours of each node
        num_sample -- neighbours to sample; None = use to_neighs as given
        -> FloatTensor [len(nodes), D]"""
        if initializer in ["1hot", "node_degree"]:
            raise NotImplementedError("the 1hot/node_degree embedding detour (aggregators.py:68-71) is out of scope")
        if num_sample is not None:
            _sample = random.sample   # the reference's stream (aggregators.py:43)
            samp_neighs = [set(_sample(tuple(to_neigh), num_sample)) if len(to_neigh) >= num_sample else to_neigh
                           for to_neigh in to_neighs]
        else:
            samp_neighs = to_neighs
        if self.gcn:
            samp_neighs = [set(s) | {int(nodes[i])} for i, s in enumerate(samp_neighs)]
        return self.aggregate(samp_neighs)

    def aggregate(self, samp_neighs):
        """Mean over explicit neighbour sets (aggregators.py:52-74)."""
        n = len(samp_neighs)
        cnt = np.fromiter((len(s) for s in samp_neighs), dtype=np.int64, count=n)
        flat = np.fromiter((x for s in samp_neighs for x in s), dtype=np.int64, count=int(cnt.sum()))
        # aggregators.py:52 -- the same expression, hence the same iteration order: a layer below
        # that samples per id (features = lambda: enc1(...)) then consumes Python's RNG in the
        # reference's order
        unique_list = list(set.union(*samp_neighs)) if n else []
        unique_nodes = np.fromiter(unique_list, dtype=np.int64, count=len(unique_list))
        order = np.argsort(unique_nodes, kind="stable")
        inverse = order[np.searchsorted(unique_nodes[order], flat)] if flat.size else np.zeros(0, dtype=np.int64)
        embed_matrix = self.features(self._ids_for_features(unique_nodes))
        dev = torch.device("cuda")
        embed_dev = embed_matrix.to(dev, torch.float32)
        if embed_dev.dim() != 2 or embed_dev.shape[0] != unique_nodes.shape[0]:
            raise native.SageError(f"features returned {tuple(embed_matrix.shape)} for {unique_nodes.shape[0]} ids")
        k = max(int(cnt.max()) if n else 1, 1)
        pos = np.full((n, k), -1, dtype=np.int32)
        mask = np.arange(k)[None, :] < cnt[:, None]
        pos[mask] = inverse.astype(np.int32)
        nbr = torch.from_numpy(pos).to(dev)
        cnt_d = torch.from_numpy(cnt.astype(np.int32)).to(dev)
        # aggregators.py:60-61: 0/0 = NaN for an empty set unless the whole batch is empty
        any_nonempty = torch.full((1,), 1 if flat.size else 0, dtype=torch.int32, device=dev)
        if embed_dev.shape[0] == 0:
            embed_dev = torch.zeros((1, embed_dev.shape[1]), device=dev)
        to_feats = autograd.gather_mean(embed_dev.contiguous(), nbr, cnt_d, any_nonempty)
        return to_feats if self.cuda else to_feats.cpu()

    def _ids_for_features(self, unique_nodes):
        ids = torch.from_numpy(np.ascontiguousarray(unique_nodes))
        return ids.cuda() if self.cuda else ids
